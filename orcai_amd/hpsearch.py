"""Hyperparameter search.  Same entry point and outputs as the reference's ``src/orcAI/hpsearch.py:110-257``
(``<output_dir>/hps_logs/{best_hyperparameters.json, all_trials.csv}``).  keras_tuner is absent, so the Hyperband
schedule (max_epochs 10, factor 3, objective = monitor, direction max; hpsearch.py:189-234) is restated as host logic:
successive-halving brackets over seeded random draws from the choice lists of the hps parameter file
(``_hp_model_builder``, hpsearch.py:21-85).  As in keras_tuner's Hyperband, a configuration promoted to the next rung RESUMES from
the state its previous rung ended in (weights, Adam moments, BatchNorm statistics, step count) and only trains the additional
epochs; the best model seen so far (by the monitor) is written to ``<output_dir>/<name>/hps/<name>.weights.npz`` -- the
reference's ModelCheckpoint(save_best_only=True) shared by all trials (hpsearch.py:227-242).  Each trial trains with the HIP
training path; with ``parallel`` every trial is data parallel over all ranks (what MirroredStrategy does in the reference).
"""

from __future__ import annotations

import copy
import math
from importlib.resources import files
from pathlib import Path

import numpy as np
import pandas as pd

from orcai_amd.architectures import build_model
from orcai_amd.auxiliary import SEED_ID_LOAD_TEST_DATA, SEED_ID_LOAD_VAL_DATA, Messenger
from orcai_amd.datasets import load_dataset
from orcai_amd.fit import EarlyStopping, ModelCheckpoint
from orcai_amd.io import read_json, write_json

DEFAULT_ORCAI_PARAMETER = files("orcai_amd.defaults").joinpath("default_orcai_parameter.json")
DEFAULT_HPS_PARAMETER = files("orcai_amd.defaults").joinpath("default_hps_parameter.json")
MAX_EPOCHS, FACTOR = 10, 3


def _draw(rng, hps_parameter: dict, orcai_parameter: dict) -> dict:
    """One configuration: a Choice per hyper-parameter (hpsearch.py:54-83)."""
    hp = {"filters": str(rng.choice(list(hps_parameter["filters"].keys()))), "kernel_size": int(rng.choice(hps_parameter["kernel_size"])),
          "dropout_rate": float(rng.choice(hps_parameter["dropout_rate"])), "batch_size": int(rng.choice(hps_parameter["batch_size"]))}
    if "lstm_units" in orcai_parameter["model"]:
        if "lstm_units" not in hps_parameter:
            raise ValueError("LSTM units not in hyperparameter search parameter. Is the right model specified?")
        hp["lstm_units"] = int(rng.choice(hps_parameter["lstm_units"]))
    elif "lstm_units" in hps_parameter:
        raise ValueError("LSTM units not in model parameter. Is the right model specified?")
    return hp


def _apply(hp: dict, orcai_parameter: dict, hps_parameter: dict) -> dict:
    p = copy.deepcopy(orcai_parameter)
    p["model"]["filters"] = hps_parameter["filters"][hp["filters"]]
    for k in ("kernel_size", "dropout_rate", "batch_size", "lstm_units"):
        if k in hp:
            p["model"][k] = hp[k]
    return p


def hyperband_brackets(max_epochs: int = MAX_EPOCHS, factor: int = FACTOR):
    """[(n_configs, [epochs per rung])] of Hyperband (Li et al.) for R = max_epochs, eta = factor."""
    s_max = int(math.floor(math.log(max_epochs, factor) + 1e-9))
    out = []
    for s in range(s_max, -1, -1):
        n = int(math.ceil((s_max + 1) / (s + 1) * factor**s))
        rungs = [max(1, int(round(max_epochs * factor ** (i - s)))) for i in range(s + 1)]
        out.append((n, rungs))
    return out


def hyperparameter_search(data_dir: Path | str, output_dir: Path | str, orcai_parameter: (Path | str) | dict = DEFAULT_ORCAI_PARAMETER,
                          hps_parameter: (Path | str) | dict = DEFAULT_HPS_PARAMETER, parallel: bool = False, data_compression: str | None = "GZIP",
                          verbosity: int = 2, msgr: Messenger | None = None, max_epochs: int = MAX_EPOCHS) -> None:
    """Same positional order as the reference (hpsearch.py:110-123); ``max_epochs`` (keyword only in practice) is Hyperband's R."""
    use_parallel = bool(parallel)
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Hyperparameter search")
    import orcai_amd.parallel as par

    rank, world, _ = par.init() if use_parallel else (0, 1, 0)
    if rank != 0:
        msgr.verbosity = 0
    data_dir, output_dir = Path(data_dir), Path(output_dir)
    if not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    if not isinstance(hps_parameter, dict):
        hps_parameter = read_json(hps_parameter)
    dataset_shape = read_json(data_dir.joinpath("dataset_shapes.json"))
    monitor = orcai_parameter["model"]["monitor"]
    hps_logs_dir = output_dir.joinpath("hps_logs")
    msgr.part("Searching hyperparameters")
    msgr.info(f"{'Parallel - running on ' + str(world) + ' GPU' if use_parallel else 'Sequential - running on 1 GPU'}")
    rng = np.random.default_rng(orcai_parameter.get("seed") or 0)
    trials = []
    model_name = orcai_parameter["name"]
    msgr.info(f"Saving best model to hps/{model_name}.weights.npz")
    # ONE checkpoint callback for the whole search, like the reference's: it remembers the best monitor value over all trials
    checkpoint = ModelCheckpoint(output_dir.joinpath(model_name, "hps", model_name + ".keras"), monitor=monitor, save_best_only=True)
    seed = int(orcai_parameter.get("seed") or 0) % (2**31)

    def run(hp: dict, epochs: int, resume: dict | None, initial_epoch: int):
        """Train configuration `hp` from epoch `initial_epoch` to `epochs` (keras_tuner: tuner/initial_epoch .. tuner/epochs),
        starting from `resume` (the Trainer state its previous rung ended in) when it was promoted.  -> (score, end state)"""
        p = _apply(hp, orcai_parameter, hps_parameter)
        bs = p["model"]["batch_size"]
        # --parallel is the reference's MirroredStrategy (hpsearch.py:170-205): the batch size of the search space is the GLOBAL batch, split
        # over the ranks (dp_batch "split"; orcai_parameter["model"]["dp_batch"] = "replicate" makes it the per-GPU batch instead)
        dpb = p["model"].get("dp_batch", "split")
        train_ds = load_dataset(data_dir.joinpath("train_dataset"), bs, compression=data_compression, seed=[SEED_ID_LOAD_TEST_DATA, orcai_parameter["seed"]], rank=rank, world_size=world, dp_batch=dpb)
        val_ds = load_dataset(data_dir.joinpath("val_dataset"), bs, compression=data_compression, seed=[SEED_ID_LOAD_VAL_DATA, orcai_parameter["seed"]], rank=rank, world_size=world, dp_batch=dpb)
        # the checkpoint is one weights file for trials of different architectures: the trial that writes it leaves its parameters beside it
        checkpoint.sidecar = {"orcai_parameter.json": p, "model_shape.json": {"input_shape": list(dataset_shape["spectrogram"]), "num_labels": len(p["calls"]), "hyperparameters": hp}}
        model = build_model(tuple(dataset_shape["spectrogram"]), p, msgr=Messenger(verbosity=0))
        model.compile(learning_rate=p["model"]["learning_rate"], seed=seed)  # in a process group the Trainer broadcasts rank 0's initial weights
        trainer = model._loop.trainer
        if resume is not None:
            trainer.load_state_dict(resume)
        hist = model.fit(train_ds, validation_data=val_ds, epochs=max(1, epochs - initial_epoch),
                         callbacks=[EarlyStopping(monitor=monitor, patience=5, mode="max", restore_best_weights=True), checkpoint])
        end_state = trainer.state_dict()
        trainer.release_graph()  # the trial's captured step and its memory pool go now, not whenever the collector finds the trainer
        return float(max(hist.history[monitor])), end_state

    for n, rungs in hyperband_brackets(max_epochs, FACTOR):
        configs = [(_draw(rng, hps_parameter, orcai_parameter), None) for _ in range(n)]  # (hyper-parameters, state after the previous rung)
        for i, epochs in enumerate(rungs):
            scores, states = [], []
            for hp, state in configs:
                score, end_state = run(hp, epochs, state, rungs[i - 1] if i > 0 else 0)
                scores.append(score)
                states.append(end_state)
                trials.append({**hp, "epochs": epochs, "initial_epoch": rungs[i - 1] if i > 0 else 0, "score": score, "status": "COMPLETED", monitor: score})
                msgr.info(f"trial {len(trials)}: {hp} epochs {epochs} -> {monitor} {score:.4f}")
            keep = max(1, len(configs) // FACTOR)
            order = np.argsort(-np.array(scores), kind="stable")[:keep]
            configs = [(configs[j][0], states[j]) for j in order]
            if i == len(rungs) - 1:
                break
    best = max(trials, key=lambda t: t["score"])
    best_hp = {k: best[k] for k in ("filters", "kernel_size", "dropout_rate", "batch_size", "lstm_units") if k in best}
    if rank == 0:
        hps_logs_dir.mkdir(parents=True, exist_ok=True)
        msgr.part("Best Hyperparameters")
        msgr.info(best_hp)
        write_json(best_hp, hps_logs_dir.joinpath("best_hyperparameters.json"))
        pd.DataFrame(trials).to_csv(hps_logs_dir.joinpath("all_trials.csv"), index=False)
        msgr.info(f"Saved trial data to {hps_logs_dir.joinpath('all_trials.csv')}")
    msgr.success("Hyperparameter search completed")
