"""Training entry point.  Same signature, outputs and error behaviour as the reference's ``src/orcAI/train.py:39-242``:
reads ``<data_dir>/{train_dataset,val_dataset}`` (+ optional ``dataset_shapes.json``), builds or reloads the model, fits
with EarlyStopping / ModelCheckpoint / ReduceLROnPlateau, writes ``<output_dir>/<name>/{<name>.weights.npz,
training_history.json, orcai_parameter.json, model_shape.json}``.

One process per GPU: under ``torchrun`` every rank trains on its own batches and the flat gradient bucket is
all-reduced over RCCL each step (synchronous data parallel, BatchNorm statistics per replica -- what the reference's
MirroredStrategy does in hpsearch.py:186-205); rank 0 writes the outputs.
"""

from __future__ import annotations

from importlib.resources import files
from pathlib import Path

import numpy as np

from orcai_amd import parallel
from orcai_amd.architectures import build_model
from orcai_amd.auxiliary import SEED_ID_LOAD_TRAIN_DATA, SEED_ID_LOAD_VAL_DATA, Messenger
from orcai_amd.datasets import load_dataset
from orcai_amd.fit import EarlyStopping, ModelCheckpoint, ReduceLROnPlateau
from orcai_amd.io import load_orcai_model, read_json, write_json

DEFAULT_ORCAI_PARAMETER = files("orcai_amd.defaults").joinpath("default_orcai_parameter.json")


def _count_params(weights: list) -> int:
    return int(np.sum([np.prod(w.shape) for w in weights]))


def train(data_dir: Path | str, output_dir: Path | str, orcai_parameter: (Path | str) | dict = DEFAULT_ORCAI_PARAMETER, data_compression: str | None = "GZIP",
          load_model: bool = False, verbosity: int = 2, msgr: Messenger | None = None) -> None:
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Training model")
    rank, world, _ = parallel.init()
    if rank != 0:
        msgr.verbosity = 0
    msgr.print_platform_info()
    msgr.print_device_info()
    msgr.part("Loading parameter")
    output_dir, data_dir = Path(output_dir), Path(data_dir)
    if not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    model_name = orcai_parameter["name"]
    model_parameter = orcai_parameter["model"]
    label_calls = orcai_parameter["calls"]

    msgr.part(f"Loading training and validation datasets from {data_dir}")
    if data_dir.joinpath("dataset_shapes.json").exists():
        dataset_shape = read_json(data_dir.joinpath("dataset_shapes.json"))
    else:
        msgr.info("Using default OrcAI dataset shapes")
        dataset_shape = {"spectrogram": [736, 171, 1], "labels": [46, 7]}
    # more than one rank (torchrun; the reference trains on one device, train.py:155): "replicate" = batch_size per GPU (throughput mode, the
    # optimiser sees batch_size x world), "split" = batch_size is the global batch as under the reference's MirroredStrategy
    dp_batch = model_parameter.get("dp_batch", "replicate")
    train_dataset = load_dataset(data_dir.joinpath("train_dataset"), model_parameter["batch_size"], compression=data_compression,
                                 seed=[SEED_ID_LOAD_TRAIN_DATA, orcai_parameter["seed"]], rank=rank, world_size=world, dp_batch=dp_batch)
    val_dataset = load_dataset(data_dir.joinpath("val_dataset"), model_parameter["batch_size"], compression=data_compression,
                               seed=[SEED_ID_LOAD_VAL_DATA, orcai_parameter["seed"]], rank=rank, world_size=world, dp_batch=dp_batch)
    if model_parameter.get("call_weights") is not None:
        call_weights = read_json(data_dir.joinpath("call_weights.json"))
        if list(call_weights.keys()) != label_calls:
            raise ValueError("Call weights do not match label calls. Please check the call weights file. Order of calls must be the same as in the orcAI parameter file.")
        call_weights_int = {n: call_weights[key] for n, key in enumerate(call_weights)}
    else:
        call_weights_int = None
    msgr.info(f"Batch size {model_parameter['batch_size']}" + ((f" per GPU x {world} GPUs" if dp_batch == "replicate" else f" split over {world} GPUs") if world > 1 else ""))
    model_dir = output_dir.joinpath(model_name)

    if load_model:
        msgr.part("Loading model")
        model, _, _ = load_orcai_model(model_dir)
    else:
        msgr.part("Building model")
        model = build_model(tuple(dataset_shape["spectrogram"]), orcai_parameter, msgr=msgr)
    msgr.part("Compiling model: " + model_name)
    model.compile(learning_rate=model_parameter["learning_rate"], seed=int(orcai_parameter["seed"] or 0) % (2**31))

    callbacks = [
        EarlyStopping(monitor=model_parameter["monitor"], patience=model_parameter["EarlyStopping_patience"], mode="max", restore_best_weights=True),
        ModelCheckpoint(model_dir.joinpath(model_name + ".keras"), monitor=model_parameter["monitor"], save_best_only=True),
        ReduceLROnPlateau(monitor=model_parameter["monitor"], factor=model_parameter["ReduceLROnPlateau_factor"],
                          patience=model_parameter["ReduceLROnPlateau_patience"], min_lr=model_parameter["ReduceLROnPlateau_min_learning_rate"]),
    ]
    msgr.info("Model size:", indent=1)
    msgr.info(f"Total parameter: {model.count_params()}")
    msgr.info(f"Trainable parameter: {_count_params(model.trainable_weights)}")
    msgr.info(f"Non-trainable parameter: {_count_params(model.non_trainable_weights)}", indent=-1)
    msgr.print_memory_usage()

    msgr.part(f"Fitting model: {model_name}")
    msgr.info(f"Monitoring {model_parameter['monitor']}")
    history = model.fit(train_dataset, validation_data=val_dataset, epochs=model_parameter["epochs"], callbacks=callbacks, class_weight=call_weights_int,
                        verbose=1 if verbosity > 2 else 0)

    if rank == 0:
        msgr.part("Saving Model")
        model_dir.mkdir(parents=True, exist_ok=True)
        model.save(model_dir.joinpath(model_name + ".keras"), include_optimizer=True)
        write_json(history.history, model_dir.joinpath("training_history.json"))
        write_json(orcai_parameter, model_dir.joinpath("orcai_parameter.json"))
        write_json({"input_shape": dataset_shape["spectrogram"], "num_labels": len(label_calls)}, model_dir.joinpath("model_shape.json"))
    msgr.success(f"Training model finished. Model saved to {model_name + '.weights.npz'}")
