"""CPU restatement (torch, CPU only) of the reference's ResNetLSTM forward pass, loss and metric.
TEST INFRASTRUCTURE.

Follows ``/root/reference/src/orcAI/architectures.py:120-241`` (res_net_LSTM_arch), ``:244-270``
(MaskedBinaryCrossentropy), ``:273-286`` (MaskedBinaryAccuracy).  Keras 3.10 / TensorFlow 2.19 are
absent from this image, so the layer semantics are restated from their documented behaviour:
**parity unpinned at the Keras boundary**; cross-checked by ``model_ref_loops`` (explicit numpy loops).

Weights are a dict of numpy arrays with Keras variable layouts:
  conv kernel (kh,kw,cin,cout) - depthwise kernel (kh,kw,c,1) - pointwise kernel (1,1,cin,cout)
  LSTM kernel (in,4u) / recurrent (u,4u) / bias (4u), gate order i,f,c,o - dense kernel (in,out)
  BN gamma/beta/mean/var (c).
"""

from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # keras.layers.BatchNormalization default epsilon
BN_MOMENTUM = 0.99
L2 = 1e-3  # kernel_regularizer l2(0.001), architectures.py:215,225,235
BCE_EPS = 1e-7  # keras.backend.epsilon() used by binary_crossentropy to clip probabilities


def param_spec(input_shape=(736, 171, 1), num_labels=7, filters=(30, 40, 50, 60), kernel_size=3, lstm_units=128):
    """Ordered list of (name, shape, kind) for the ResNetLSTM variables (architectures.py:162-241)."""
    k = kernel_size
    spec = [("conv0/kernel", (k, k, input_shape[2], 16), "conv"), ("conv0/bias", (16,), "zeros")]
    spec += _bn("bn0", 16)
    h, w, c = input_shape[0], input_shape[1], 16
    for b, s in enumerate(filters, start=1):
        spec += [(f"b{b}/sep_a/depthwise", (k, k, c, 1), "conv"), (f"b{b}/sep_a/pointwise", (1, 1, c, s), "conv"), (f"b{b}/sep_a/bias", (s,), "zeros")]
        spec += _bn(f"b{b}/bn_a", s)
        spec += [(f"b{b}/sep_b/depthwise", (k, k, s, 1), "conv"), (f"b{b}/sep_b/pointwise", (1, 1, s, s), "conv"), (f"b{b}/sep_b/bias", (s,), "zeros")]
        spec += _bn(f"b{b}/bn_b", s)
        spec += [(f"b{b}/res/kernel", (1, 1, c, s), "conv"), (f"b{b}/res/bias", (s,), "zeros")]
        c = s
        h, w = -(-h // 2), -(-w // 2)
    spec += [("sep_f/depthwise", (k, k, c, 1), "conv"), ("sep_f/pointwise", (1, 1, c, 36), "conv"), ("sep_f/bias", (36,), "zeros")]
    spec += _bn("bn_f", 36)
    feat = w * 36
    u = lstm_units
    for layer, fin in ((1, feat), (2, 2 * u)):
        for d in ("fwd", "bwd"):
            spec += [(f"lstm{layer}/{d}/kernel", (fin, 4 * u), "lstm"), (f"lstm{layer}/{d}/recurrent", (u, 4 * u), "orthogonal"), (f"lstm{layer}/{d}/bias", (4 * u,), "lstm_bias")]
    spec += [("dense1/kernel", (2 * u, 128), "conv"), ("dense1/bias", (128,), "zeros")]
    spec += _bn("bn_d", 128)
    spec += [("dense2/kernel", (128, num_labels), "glorot"), ("dense2/bias", (num_labels,), "zeros")]
    return spec


def _bn(name, c):
    return [(f"{name}/gamma", (c,), "ones"), (f"{name}/beta", (c,), "zeros"), (f"{name}/mean", (c,), "bn_mean"), (f"{name}/var", (c,), "bn_var")]


def random_params(seed=1, randomize_bn=True, **arch):
    """Seeded synthetic weights (the trained orcai-v1.keras is absent).  With randomize_bn the BN
    statistics / affine terms and all biases are non-trivial so every term of the forward is exercised."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, kind in param_spec(**arch):
        if kind in ("conv", "lstm", "glorot", "orthogonal"):
            fan_in = int(np.prod(shape[:-1])) if len(shape) > 1 else shape[0]
            if name.endswith("depthwise"):
                fan_in = shape[0] * shape[1]
            w = rng.standard_normal(shape) * math.sqrt(2.0 / fan_in) if kind == "conv" else rng.uniform(-1, 1, shape) * math.sqrt(6.0 / (shape[0] + shape[-1]))
        elif kind == "lstm_bias":
            u = shape[0] // 4
            w = np.zeros(shape)
            w[u : 2 * u] = 1.0  # unit_forget_bias
            if randomize_bn:
                w += 0.1 * rng.standard_normal(shape)
        elif kind == "ones":
            w = 1.0 + (0.2 * rng.standard_normal(shape) if randomize_bn else 0)
        elif kind == "zeros":
            w = 0.1 * rng.standard_normal(shape) if randomize_bn else np.zeros(shape)
        elif kind == "bn_mean":
            w = 0.1 * rng.standard_normal(shape) if randomize_bn else np.zeros(shape)
        elif kind == "bn_var":
            w = rng.uniform(0.5, 1.5, shape) if randomize_bn else np.ones(shape)
        out[name] = np.asarray(w, dtype=np.float32)
    return out


def calibrated_params(seed=1, calib_batch=2, **arch):
    """random_params + BN moving statistics set from the activations of a seeded calibration batch (jittered),
    so that every layer's output is O(1) as in a trained network.  Without this the synthetic network's
    features reach ~1e3 and the LSTM becomes ill-conditioned (fp32 vs fp64 differ by 1e-4 on the CPU too)."""
    global _CALIBRATE
    p = random_params(seed=seed, **arch)
    shape = arch.get("input_shape", (736, 171, 1))
    rng = np.random.default_rng(seed + 1000)
    x = rng.random((calib_batch, *shape), dtype=np.float32)
    _CALIBRATE = rng
    try:
        forward_ref(p, x, dtype=torch.float64)
    finally:
        _CALIBRATE = None
    return p


def same_pad(n, k, s):
    """TF/Keras padding="same": (out, pad_before, pad_after)."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


def _conv_same(x, kernel, bias, stride, dtype, groups=1):
    """x NCHW; kernel Keras (kh,kw,cin,cout) (or depthwise (kh,kw,c,1) with groups=c)."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    _, pt, pb = same_pad(x.shape[2], kh, stride)
    _, pl, pr = same_pad(x.shape[3], kw, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    if groups == 1:
        w = _t(kernel, dtype).permute(3, 2, 0, 1).contiguous()
    else:
        w = _t(kernel, dtype).permute(2, 3, 0, 1).contiguous()  # (c,1,kh,kw)
    b = None if bias is None else _t(bias, dtype)
    return F.conv2d(x, w, b, stride=stride, groups=groups)


_CALIBRATE = None  # set by calibrated_params(): np.random.Generator used to jitter the measured statistics


def _bn_infer(x, p, name, dtype, axis=1):
    """tf.nn.batch_normalization form: inv = gamma*rsqrt(var+eps); x*inv + (beta - mean*inv)."""
    if _CALIBRATE is not None:  # give the BN layer the statistics a trained network would have (activations stay O(1))
        dims = [d for d in range(x.dim()) if d != axis]
        c = x.shape[axis]
        p[name + "/mean"] = (x.mean(dim=dims).numpy() + 0.1 * _CALIBRATE.standard_normal(c)).astype(np.float32)
        p[name + "/var"] = (x.var(dim=dims, unbiased=False).numpy() * _CALIBRATE.uniform(0.8, 1.25, c) + 1e-3).astype(np.float32)
    inv = _t(p[name + "/gamma"], dtype) * torch.rsqrt(_t(p[name + "/var"], dtype) + BN_EPS)
    shift = _t(p[name + "/beta"], dtype) - _t(p[name + "/mean"], dtype) * inv
    shape = [1] * x.dim()
    shape[axis] = -1
    return x * inv.view(shape) + shift.view(shape)


def _sepconv(x, p, name, dtype):
    c = x.shape[1]
    x = _conv_same(x, p[name + "/depthwise"], None, 1, dtype, groups=c)
    return _conv_same(x, p[name + "/pointwise"], p[name + "/bias"], 1, dtype)


def _maxpool_same(x, k=(3, 2), s=2):
    _, pt, pb = same_pad(x.shape[2], k[0], s)
    _, pl, pr = same_pad(x.shape[3], k[1], s)
    x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(x, kernel_size=k, stride=s)


def _lstm_dir(x, kernel, recurrent, bias, dtype, reverse):
    """Keras LSTM (tanh / sigmoid, gate order i,f,c,o), return_sequences=True.  x: (B,T,F)."""
    W, U, b = _t(kernel, dtype), _t(recurrent, dtype), _t(bias, dtype)
    B, T, _ = x.shape
    u = U.shape[0]
    h = torch.zeros(B, u, dtype=dtype)
    c = torch.zeros(B, u, dtype=dtype)
    xz = x @ W + b
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        z = xz[:, t] + h @ U
        i, f, g, o = z[:, :u], z[:, u : 2 * u], z[:, 2 * u : 3 * u], z[:, 3 * u :]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def _bilstm(x, p, name, dtype):
    fwd = _lstm_dir(x, p[name + "/fwd/kernel"], p[name + "/fwd/recurrent"], p[name + "/fwd/bias"], dtype, False)
    bwd = _lstm_dir(x, p[name + "/bwd/kernel"], p[name + "/bwd/recurrent"], p[name + "/bwd/bias"], dtype, True)
    return torch.cat([fwd, bwd], dim=2)


def forward_ref(p: dict, snippets: np.ndarray, n_blocks: int | None = None, dtype=torch.float32, return_intermediates=False):
    """Inference forward (BN moving statistics, dropout off): (B,H,W,1) -> (B,H/2**n,num_labels).
    architectures.py:162-241."""
    if n_blocks is None:
        n_blocks = sum(1 for k in p if k.endswith("/res/kernel"))
    inter = {}
    with torch.no_grad():
        x = _t(snippets, dtype).permute(0, 3, 1, 2)  # NHWC -> NCHW
        x = _conv_same(x, p["conv0/kernel"], p["conv0/bias"], 1, dtype)
        x = torch.relu(_bn_infer(x, p, "bn0", dtype))
        inter["conv0"] = x
        prev = x
        for b in range(1, n_blocks + 1):
            x = torch.relu(x)
            x = _sepconv(x, p, f"b{b}/sep_a", dtype)
            x = torch.relu(_bn_infer(x, p, f"b{b}/bn_a", dtype))
            inter[f"b{b}/a"] = x
            x = _sepconv(x, p, f"b{b}/sep_b", dtype)
            x = _bn_infer(x, p, f"b{b}/bn_b", dtype)
            inter[f"b{b}/b"] = x
            x = _maxpool_same(x)
            res = _conv_same(prev, p[f"b{b}/res/kernel"], p[f"b{b}/res/bias"], 2, dtype)
            x = x + res
            inter[f"b{b}"] = x
            prev = x
        x = _sepconv(x, p, "sep_f", dtype)
        x = torch.relu(_bn_infer(x, p, "bn_f", dtype))
        B, C, H, W = x.shape
        x = x.permute(0, 2, 3, 1).reshape(B, H, W * C)  # Keras Reshape of NHWC: feature = w*C + c
        inter["features"] = x
        x = _bilstm(x, p, "lstm1", dtype)
        inter["lstm1"] = x
        x = _bilstm(x, p, "lstm2", dtype)
        inter["lstm2"] = x
        x = torch.relu(x @ _t(p["dense1/kernel"], dtype) + _t(p["dense1/bias"], dtype))
        x = _bn_infer(x, p, "bn_d", dtype, axis=2)
        x = torch.sigmoid(x @ _t(p["dense2/kernel"], dtype) + _t(p["dense2/bias"], dtype))
    out = x.numpy()
    if return_intermediates:
        return out, {k: v.numpy() for k, v in inter.items()}
    return out


def forward_ref_1dconv(p: dict, snippets: np.ndarray, n_blocks: int | None = None, dtype=torch.float32, return_intermediates=False):
    """ResNet1DConv inference forward (architectures.py:18-117): the same convolutional trunk (Dropout layers are identity at
    inference), ReduceFrequencyMean = mean over the frequency axis (:10-15), Conv1D(num_labels, kernel_size = 36 channels,
    padding "same", sigmoid) over time (:107-115).  Keras/TF "same" with an even kernel pads (K-1)//2 left, K//2 right."""
    if n_blocks is None:
        n_blocks = sum(1 for k in p if k.endswith("/res/kernel"))
    inter = {}
    with torch.no_grad():
        x = _t(snippets, dtype).permute(0, 3, 1, 2)
        x = _conv_same(x, p["conv0/kernel"], p["conv0/bias"], 1, dtype)
        x = torch.relu(_bn_infer(x, p, "bn0", dtype))
        prev = x
        for b in range(1, n_blocks + 1):
            x = torch.relu(x)
            x = torch.relu(_bn_infer(_sepconv(x, p, f"b{b}/sep_a", dtype), p, f"b{b}/bn_a", dtype))
            x = _bn_infer(_sepconv(x, p, f"b{b}/sep_b", dtype), p, f"b{b}/bn_b", dtype)
            x = _maxpool_same(x) + _conv_same(prev, p[f"b{b}/res/kernel"], p[f"b{b}/res/bias"], 2, dtype)
            prev = x
        x = torch.relu(_bn_infer(_sepconv(x, p, "sep_f", dtype), p, "bn_f", dtype))  # (B, 36, T', W')
        x = x.mean(dim=3).permute(0, 2, 1)  # ReduceFrequencyMean: (B, T', 36)
        inter["freq_mean"] = x
        w = _t(p["conv1d/kernel"], dtype)  # (K, C, L)
        K = w.shape[0]
        xp = torch.nn.functional.pad(x.permute(0, 2, 1), ((K - 1) // 2, K // 2))  # (B, C, T' + K - 1)
        y = torch.nn.functional.conv1d(xp, w.permute(2, 1, 0).contiguous(), _t(p["conv1d/bias"], dtype))  # weight (L, C, K)
        x = torch.sigmoid(y.permute(0, 2, 1))
    out = x.numpy()
    if return_intermediates:
        return out, {k: v.numpy() for k, v in inter.items()}
    return out


def masked_bce_ref(y_true: np.ndarray, y_pred: np.ndarray, mask_value=-1.0) -> float:
    """architectures.py:262-270: BCE over unmasked elements (probabilities clipped to [1e-7, 1-1e-7]), mean."""
    m = y_true != mask_value
    t = y_true[m].astype(np.float64)
    q = np.clip(y_pred[m].astype(np.float64), BCE_EPS, 1 - BCE_EPS)
    return float(np.mean(-(t * np.log(q) + (1 - t) * np.log(1 - q))))


def masked_binary_accuracy_ref(y_true: np.ndarray, y_pred: np.ndarray, mask_value=-1.0) -> float:
    """architectures.py:281-286: mean((y_pred > 0.5) == y_true) over unmasked elements."""
    m = y_true != mask_value
    return float(np.mean((y_pred[m] > 0.5).astype(np.float32) == y_true[m]))


def count_params(p: dict) -> tuple[int, int]:
    """(trainable, non-trainable): BN moving statistics are the non-trainable variables."""
    nt = sum(int(v.size) for k, v in p.items() if k.endswith("/mean") or k.endswith("/var"))
    return sum(int(v.size) for v in p.values()) - nt, nt
