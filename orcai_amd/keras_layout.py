"""Where each variable of a reference-trained model lives inside a Keras weight file, and how it maps to this package's
``<name>.weights.npz`` (reference ``io.py:386-404``: ``<name>.keras`` via ``keras.saving.load_model``, legacy
``model_weights.h5`` via ``model.load_weights``).  Pure Python: no keras / h5py needed to BUILD the map, so the map itself is
testable here; reading the HDF5 container is ``tools/keras_to_npz.py``'s job, where h5py exists.

Keras names layers ``<snake_case class>[_<n>]`` in creation order per class.  ``res_net_LSTM_arch`` (architectures.py:162-241)
creates, per class:
  Conv2D              #0 entry conv, #b the 1x1 stride-2 residual conv of block b (b = 1..len(filters))
  SeparableConv2D     #2(b-1) / #2(b-1)+1 the two separable convs of block b, #2*len(filters) the final one (36 filters)
  BatchNormalization  #0 after the entry conv, #2b-1 / #2b after block b's separable convs, then final conv, then Dense-128
  Bidirectional       #0, #1 (forward_layer / backward_layer each hold one LSTM cell: kernel, recurrent_kernel, bias)
  Dense               #0 Dense(128), #1 Dense(num_labels)
Variable order inside a layer (``layer.weights``): Conv2D / Dense kernel, bias; SeparableConv2D depthwise_kernel, pointwise_kernel,
bias; BatchNormalization gamma, beta, moving_mean, moving_variance; LSTM cell kernel, recurrent_kernel, bias.
Array layouts are Keras' own in both files (the npz keeps Keras layouts), so conversion is a rename.
"""

from __future__ import annotations

import re

SNAKE = {"Conv2D": "conv2d", "SeparableConv2D": "separable_conv2d", "BatchNormalization": "batch_normalization", "Bidirectional": "bidirectional",
         "Dense": "dense", "Conv1D": "conv1d"}
_LAYER_VARS = {"Conv2D": ["kernel", "bias"], "Dense": ["kernel", "bias"], "Conv1D": ["kernel", "bias"],
               "SeparableConv2D": ["depthwise", "pointwise", "bias"], "BatchNormalization": ["gamma", "beta", "mean", "var"]}
_KERAS_VAR_NAMES = {"kernel": "kernel", "bias": "bias", "depthwise": "depthwise_kernel", "pointwise": "pointwise_kernel", "gamma": "gamma", "beta": "beta",
                    "mean": "moving_mean", "var": "moving_variance", "recurrent": "recurrent_kernel"}


def layer_roles(n_blocks: int, architecture: str = "ResNetLSTM") -> list[tuple[str, int, str]]:
    """[(keras class, ordinal within the class, npz prefix)] for every weighted layer of the architecture."""
    roles = [("Conv2D", 0, "conv0"), ("BatchNormalization", 0, "bn0")]
    for b in range(1, n_blocks + 1):
        roles += [("SeparableConv2D", 2 * (b - 1), f"b{b}/sep_a"), ("BatchNormalization", 2 * b - 1, f"b{b}/bn_a"),
                  ("SeparableConv2D", 2 * (b - 1) + 1, f"b{b}/sep_b"), ("BatchNormalization", 2 * b, f"b{b}/bn_b"), ("Conv2D", b, f"b{b}/res")]
    roles += [("SeparableConv2D", 2 * n_blocks, "sep_f"), ("BatchNormalization", 2 * n_blocks + 1, "bn_f")]
    if architecture == "ResNetLSTM":
        roles += [("Bidirectional", 0, "lstm1"), ("Bidirectional", 1, "lstm2"), ("Dense", 0, "dense1"), ("BatchNormalization", 2 * n_blocks + 2, "bn_d"),
                  ("Dense", 1, "dense2")]
    elif architecture == "ResNet1DConv":
        roles += [("Conv1D", 0, "conv1d")]
    else:
        raise ValueError(f"Unknown model architecture: {architecture}")
    return roles


def variable_map(n_blocks: int, architecture: str = "ResNetLSTM") -> list[tuple[str, int, tuple, int, str]]:
    """[(keras class, ordinal, sub-path inside the layer, variable index, npz name)].  sub-path is () except for Bidirectional:
    ("forward_layer", "cell") / ("backward_layer", "cell")."""
    out = []
    for cls, k, prefix in layer_roles(n_blocks, architecture):
        if cls == "Bidirectional":
            for sub, d in (("forward_layer", "fwd"), ("backward_layer", "bwd")):
                for i, v in enumerate(("kernel", "recurrent", "bias")):
                    out.append((cls, k, (sub, "cell"), i, f"{prefix}/{d}/{v}"))
        else:
            for i, v in enumerate(_LAYER_VARS[cls]):
                out.append((cls, k, (), i, f"{prefix}/{v}"))
    return out


def _ordinal_of(layer_name: str, cls: str):
    """('conv2d_7', 'Conv2D') -> 7; ('conv2d', 'Conv2D') -> 0; None when the name is not of that class."""
    m = re.fullmatch(re.escape(SNAKE[cls]) + r"(?:_(\d+))?", layer_name)
    return None if m is None else int(m.group(1) or 0)


def _layers_by_class(layer_names) -> dict:
    """{class: [layer names sorted by their numeric suffix]}: the RANK in that list is the creation ordinal, which also holds
    for a model that was not built in a fresh Keras session (names then start at some offset)."""
    out = {}
    for cls in SNAKE:
        found = sorted((o, n) for n in set(layer_names) if (o := _ordinal_of(n, cls)) is not None)
        out[cls] = [n for _, n in found]
    return out


def from_keras3_paths(arrays: dict, n_blocks: int, architecture: str = "ResNetLSTM") -> dict:
    """Keras 3 weight store (``model.weights.h5`` inside ``<name>.keras``, also what ``model.save_weights('x.weights.h5')``
    writes): dataset paths ``layers/<layer>/[forward_layer/cell/]vars/<i>`` -> {npz name: array}.  Optimizer state (``optimizer/``)
    and anything else is ignored."""
    layer_names = [p.split("/")[1] for p in arrays if p.startswith("layers/") and p.count("/") >= 3]
    by_cls = _layers_by_class(layer_names)
    out = {}
    for cls, k, sub, i, name in variable_map(n_blocks, architecture):
        if k >= len(by_cls[cls]):
            raise KeyError(f"weight file has {len(by_cls[cls])} {cls} layers, the architecture needs #{k} ({name})")
        path = "/".join(["layers", by_cls[cls][k], *sub, "vars", str(i)])
        if path not in arrays:
            raise KeyError(f"{path} (-> {name}) not found in the weight file")
        out[name] = arrays[path]
    return out


def expected_keras3_paths(arrays: dict, n_blocks: int, architecture: str = "ResNetLSTM") -> list[str]:
    """The dataset paths from_keras3_paths reads from this file (layer names resolved against the file's own layer list)."""
    layer_names = [p.split("/")[1] for p in arrays if p.startswith("layers/") and p.count("/") >= 3]
    by_cls = _layers_by_class(layer_names)
    return ["/".join(["layers", by_cls[cls][k], *sub, "vars", str(i)]) for cls, k, sub, i, _ in variable_map(n_blocks, architecture) if k < len(by_cls[cls])]


def to_keras3_paths(weights: dict, n_blocks: int, architecture: str = "ResNetLSTM") -> dict:
    """Inverse of from_keras3_paths for a model built in a fresh session (layer names without offset): used by the round-trip test
    and to document the layout."""
    out = {}
    for cls, k, sub, i, name in variable_map(n_blocks, architecture):
        layer = SNAKE[cls] + (f"_{k}" if k else "")
        out["/".join(["layers", layer, *sub, "vars", str(i)])] = weights[name]
    return out


def from_legacy_h5(arrays: dict, layer_names: list, weight_names: dict, n_blocks: int, architecture: str = "ResNetLSTM") -> dict:
    """Legacy ``model_weights.h5`` (tf.keras / Keras 2 ``save_weights``): root attribute ``layer_names`` and per-layer attribute
    ``weight_names`` (e.g. ``conv2d/kernel:0``, ``bidirectional/forward_lstm/lstm_cell/recurrent_kernel:0``) list the datasets in
    ``layer.weights`` order; arrays = {"<layer>/<weight name>": array}."""
    by_cls = _layers_by_class(layer_names)
    out = {}
    for cls, k, sub, i, name in variable_map(n_blocks, architecture):
        if k >= len(by_cls[cls]):
            raise KeyError(f"weight file has {len(by_cls[cls])} {cls} layers, the architecture needs #{k} ({name})")
        layer = by_cls[cls][k]
        names = list(weight_names[layer])
        if cls == "Bidirectional":  # forward cell's three variables first, then the backward cell's
            names = [n for n in names if ("backward" in n) == (sub[0] == "backward_layer")]
        want = _KERAS_VAR_NAMES[name.rsplit("/", 1)[1]]
        hits = [n for n in names if n.split("/")[-1].split(":")[0] == want]
        wn = hits[0] if len(hits) == 1 else names[i]
        out[name] = arrays[f"{layer}/{wn}"]
    return out
