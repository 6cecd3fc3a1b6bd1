#!/usr/bin/env python3
"""Convert a reference-trained orcAI model to this package's weight store.

    python tools/keras_to_npz.py MODEL_DIR            # reads MODEL_DIR/<name>.keras or MODEL_DIR/model_weights.h5
                                                      # writes MODEL_DIR/<name>.weights.npz
    python tools/keras_to_npz.py --layout [MODEL_DIR] # prints every dataset path the converter expects, with its shape (no h5py needed;
                                                      # MODEL_DIR supplies orcai_parameter.json / model_shape.json, default: the shipped orcai-V1)

What is read.  ``<name>.keras`` (Keras 3, reference io.py:386-392) is a zip archive; the ONLY member read is ``model.weights.h5``
(``config.json`` and ``metadata.json`` are ignored: the architecture comes from orcai_parameter.json).  Inside it, one HDF5 dataset
per variable at ``layers/<layer>/vars/<i>`` -- ``layers/<bidirectional>/forward_layer/cell/vars/<i>`` and ``.../backward_layer/cell/vars/<i>``
for the LSTM cells -- with <layer> = Keras' snake-case class name plus creation ordinal (``conv2d``, ``conv2d_1`` .., ``separable_conv2d`` ..,
``batch_normalization`` .., ``bidirectional``, ``bidirectional_1``, ``dense``, ``dense_1``) and <i> the index in ``layer.weights`` order
(Conv2D / Dense: kernel, bias; SeparableConv2D: depthwise_kernel, pointwise_kernel, bias; BatchNormalization: gamma, beta, moving_mean,
moving_variance; LSTM cell: kernel, recurrent_kernel, bias).  ``optimizer/...`` datasets are ignored.  The legacy ``model_weights.h5``
(io.py:394-404) is read through its ``layer_names`` / ``weight_names`` attributes.  ``--layout`` prints the full table for a model.
The converter FAILS -- naming the path -- on a variable the architecture needs and the file lacks, on a ``layers/...`` dataset of the file
that no variable of the architecture maps to, and on any shape that differs from ``ResNetLSTM.variable_spec()``; nothing is written then.

Needs ``h5py`` (NOT present in the build image -- run it where the reference's environment exists; keras itself is not needed:
the ``.keras`` archive is a zip whose ``model.weights.h5`` member is read directly).  What is read, and how variable paths map
to npz names, is defined in ``orcai_amd/keras_layout.py`` (reference ``io.py:386-404``, ``architectures.py:162-241``).  After the
rename every array is checked against ``ResNetLSTM.variable_spec()`` (name set and shapes) before anything is written; optimizer
state in the archive is ignored (``load_orcai_model`` only needs the forward weights; training resumes with fresh Adam moments).
"""

from __future__ import annotations

import io
import json
import sys
import zipfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from orcai_amd import keras_layout  # noqa: E402


def _read_h5_datasets(h5file) -> dict:
    out = {}

    def visit(name, obj):
        if hasattr(obj, "shape") and hasattr(obj, "dtype"):
            out[name] = np.asarray(obj)

    h5file.visititems(visit)
    return out


def _mapped_keras3(arrays: dict, n_blocks: int, architecture: str, what: str) -> dict:
    """from_keras3_paths + the check that every ``layers/...`` dataset of the file was consumed (an unmapped variable means the file is not
    the architecture orcai_parameter.json describes)."""
    try:
        out = keras_layout.from_keras3_paths(arrays, n_blocks, architecture)
    except KeyError as e:
        raise SystemExit(f"{what}: {e.args[0]}") from e
    used = set(keras_layout.expected_keras3_paths(arrays, n_blocks, architecture))
    extra = sorted(p for p in arrays if p.startswith("layers/") and p not in used)
    if extra:
        raise SystemExit(f"{what}: {len(extra)} dataset(s) no variable of the {architecture} architecture maps to: {extra[:8]}{' ...' if len(extra) > 8 else ''}")
    return out


def check_against_spec(weights: dict, model) -> None:
    """Every variable of the architecture present, nothing else, shapes equal to variable_spec(): all problems listed at once."""
    spec = {n: tuple(sh) for n, sh, *_ in model.variable_spec()}
    problems = [f"missing {n} {sh}" for n, sh in spec.items() if n not in weights]
    problems += [f"unexpected {n} {tuple(np.shape(a))}" for n, a in weights.items() if n not in spec]
    problems += [f"shape of {n}: file {tuple(np.shape(weights[n]))}, architecture {sh}" for n, sh in spec.items() if n in weights and tuple(np.shape(weights[n])) != sh]
    if problems:
        raise SystemExit("the weight file does not match orcai_parameter.json / model_shape.json:\n  " + "\n  ".join(problems))


def print_layout(model_dir: Path | None) -> None:
    from orcai_amd.architectures import build_model
    from orcai_amd.auxiliary import Messenger
    from orcai_amd.io import read_json

    if model_dir is None:
        model_dir = Path(__file__).resolve().parents[1] / "orcai_amd" / "models" / "orcai-V1"
    param, shape = read_json(model_dir / "orcai_parameter.json"), read_json(model_dir / "model_shape.json")
    model = build_model(tuple(shape["input_shape"]), param, msgr=Messenger(verbosity=0))
    spec = {n: tuple(sh) for n, sh, *_ in model.variable_spec()}
    print(f"{param['name']}.keras -> zip member model.weights.h5 -> HDF5 datasets ({param['architecture']}, filters {param['model']['filters']}):")
    for path, name in keras_layout.to_keras3_paths({n: n for n in spec}, len(param["model"]["filters"]), param["architecture"]).items():
        print(f"  {path:62s} {str(spec[name]):22s} -> {name}")


def load_keras_arrays(path: Path, n_blocks: int, architecture: str) -> dict:
    try:
        import h5py
    except ImportError as e:  # the build image: say what is missing instead of failing obscurely
        raise SystemExit(f"tools/keras_to_npz.py needs h5py to read {path.name}: {e}") from e
    if path.suffix == ".keras":
        with zipfile.ZipFile(path) as z:
            if "model.weights.h5" not in z.namelist():
                raise SystemExit(f"{path.name}: zip member 'model.weights.h5' not found (members: {z.namelist()})")
            blob = z.read("model.weights.h5")
        with h5py.File(io.BytesIO(blob), "r") as f:
            return _mapped_keras3(_read_h5_datasets(f), n_blocks, architecture, path.name)
    with h5py.File(path, "r") as f:
        if "layers" in f:  # a Keras 3 ``*.weights.h5``
            return _mapped_keras3(_read_h5_datasets(f), n_blocks, architecture, path.name)
        dec = lambda v: v.decode() if isinstance(v, bytes) else str(v)  # noqa: E731
        layer_names = [dec(n) for n in f.attrs["layer_names"]]
        weight_names = {ln: [dec(w) for w in f[ln].attrs["weight_names"]] for ln in layer_names}
        arrays = {f"{ln}/{w}": np.asarray(f[ln][w]) for ln in layer_names for w in weight_names[ln]}
        return keras_layout.from_legacy_h5(arrays, layer_names, weight_names, n_blocks, architecture)


def convert(model_dir: Path) -> Path:
    from orcai_amd.architectures import build_model
    from orcai_amd.auxiliary import Messenger
    from orcai_amd.io import WEIGHTS_SUFFIX, read_json

    model_dir = Path(model_dir)
    param = read_json(model_dir / "orcai_parameter.json")
    shape = read_json(model_dir / "model_shape.json")
    name = param["name"]
    src = model_dir / f"{name}.keras"
    if not src.exists():
        src = model_dir / "model_weights.h5"
    if not src.exists():
        raise SystemExit(f"neither {name}.keras nor model_weights.h5 in {model_dir}")
    weights = load_keras_arrays(src, len(param["model"]["filters"]), param["architecture"])
    model = build_model(tuple(shape["input_shape"]), param, msgr=Messenger(verbosity=0))
    check_against_spec(weights, model)
    model.set_weights_dict(weights)
    out = model_dir / (name + WEIGHTS_SUFFIX)
    model.save_weights(out)
    print(json.dumps({"source": src.name, "written": out.name, "variables": len(weights), "parameters": model.count_params()}))
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or args[0] in ("-h", "--help"):
        print(__doc__)
        raise SystemExit(0 if args else 2)
    if args[0] == "--layout":
        print_layout(Path(args[1]) if len(args) > 1 else None)
    elif len(args) == 1:
        convert(Path(args[0]))
    else:
        raise SystemExit(__doc__)
