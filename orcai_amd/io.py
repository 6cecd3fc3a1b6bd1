"""JSON / array / model IO for the hot path.  Mirrors the hot-path part of reference
``src/orcAI/io.py`` (read_json :259-274, write_json :277-293, write_vector_to_json :221-238,
generate_times_from_spectrogram :241-256, load_orcai_model :357-410).

Container formats that belong to third-party libraries (zarr, tf.data snapshots, Keras .keras/HDF5)
are out of scope: arrays are stored as raw ``.npy`` and model weights as ``.npz``
with Keras weight names and layouts.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from orcai_amd.json_encoder import JsonEncoderExt


def read_json(filename: Path | str) -> dict:
    with open(filename, "r") as file:
        return json.load(file)


def write_json(dictionary, filename) -> None:
    with open(filename, "w") as file:
        file.write(json.dumps(dictionary, indent=4, cls=JsonEncoderExt))


def write_vector_to_json(vector, filename: Path | str) -> None:
    """Equally spaced vector in short form {min, max, length} (io.py:221-238)."""
    dictionary = {"min": vector[0], "max": vector[-1], "length": len(vector)}
    with open(filename, "w") as f:
        json.dump(dictionary, f, indent=4, cls=JsonEncoderExt)


def generate_times_from_spectrogram(filename: Path | str) -> np.ndarray:
    with open(filename, "r") as f:
        d = json.load(f)
    return np.linspace(d["min"], d["max"], d["length"])


def save_array(obj: np.ndarray, filename: Path) -> None:
    """Raw float32 ``.npy`` store (stands where the reference writes zarr, io.py:296-331)."""
    np.save(filename, np.ascontiguousarray(obj, dtype=np.float32))


def load_array(filename: Path, mmap: bool = True) -> np.ndarray:
    return np.load(filename, mmap_mode="r" if mmap else None)


WEIGHTS_SUFFIX = ".weights.npz"


def load_orcai_model(model_dir: Path):
    """(model, orcai_parameter, shape) from a model directory (io.py:357-410).

    Looks for ``<name>.weights.npz`` (this package's weight store: Keras variable names and layouts).
    A Keras ``<name>.keras`` / legacy ``model_weights.h5`` (io.py:386-404) is an HDF5 container, which needs h5py to read: it is
    converted once with ``python tools/keras_to_npz.py <model_dir>`` (variable paths -> npz names: ``orcai_amd/keras_layout.py``).
    """
    from orcai_amd.architectures import build_model

    model_dir = Path(model_dir)
    orcai_parameter = read_json(model_dir.joinpath("orcai_parameter.json"))
    shape = read_json(model_dir.joinpath("model_shape.json"))
    name = orcai_parameter["name"]
    wpath = model_dir.joinpath(name + WEIGHTS_SUFFIX)
    if wpath.exists():
        from orcai_amd.auxiliary import Messenger

        model = build_model(tuple(shape["input_shape"]), orcai_parameter, msgr=Messenger(verbosity=0))
        model.load_weights(wpath)
        return model, orcai_parameter, shape
    if model_dir.joinpath(name + ".keras").exists() or model_dir.joinpath("model_weights.h5").exists():
        raise ValueError(
            f"{model_dir} holds Keras weights ({name}.keras / model_weights.h5) but no {name}{WEIGHTS_SUFFIX}; convert them once with "
            f"`python tools/keras_to_npz.py {model_dir}` (needs h5py)"
        )
    raise ValueError(
        f"Couldn't find model weights ({name}{WEIGHTS_SUFFIX}) in {model_dir}.  The trained orcai-V1 weights are a large blob the reference "
        f"ships outside its source tree ({name}.keras); put that file into the directory and convert it once with "
        f"`python tools/keras_to_npz.py {model_dir}` (needs h5py), or -- to exercise the pipeline with UNTRAINED weights of the same "
        f"architecture -- run `orcai init-weights {model_dir} --seed 1`."
    )
