"""Second, independent CPU restatement of the ResNetLSTM / ResNet1DConv forward pass: explicit numpy loops over output pixels,
no convolution library.  TEST INFRASTRUCTURE (two independent CPU implementations must agree before the HIP kernels are
compared with either).  Only for tiny shapes.

Conventions restated from the Keras / TensorFlow documentation, independently of ``model_ref`` (architectures.py:120-241):
  * "same" padding: out = ceil(in / stride), total pad = max((out - 1) * stride + k - in, 0), floor(total / 2) before, rest after;
  * Conv2D / SeparableConv2D are cross-correlations (no kernel flip); depth_multiplier 1; bias on the pointwise part only;
  * MaxPooling2D pads with -inf; BatchNormalization (inference): gamma * (x - mean) / sqrt(var + 1e-3) + beta;
  * LSTM gates in the order i, f, c, o; h_t = o * tanh(c_t); the backward direction is returned in input time order.
"""

from __future__ import annotations

import numpy as np

BN_EPS = 1e-3


def _pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def conv2d_same(x, kernel, bias, stride):
    """x (H, W, Cin), kernel (kh, kw, Cin, Cout) -> (Ho, Wo, Cout)."""
    H, W, Cin = x.shape
    kh, kw, _, Cout = kernel.shape
    Ho, pt = _pads(H, kh, stride)
    Wo, pl = _pads(W, kw, stride)
    out = np.zeros((Ho, Wo, Cout))
    for i in range(Ho):
        for j in range(Wo):
            acc = np.zeros(Cout) if bias is None else bias.astype(np.float64).copy()
            for a in range(kh):
                for b in range(kw):
                    y, xx = i * stride - pt + a, j * stride - pl + b
                    if 0 <= y < H and 0 <= xx < W:
                        acc += x[y, xx, :] @ kernel[a, b]
            out[i, j] = acc
    return out


def depthwise_same(x, kernel):
    """x (H, W, C), kernel (kh, kw, C, 1)."""
    H, W, C = x.shape
    kh, kw = kernel.shape[:2]
    _, pt = _pads(H, kh, 1)
    _, pl = _pads(W, kw, 1)
    out = np.zeros((H, W, C))
    for i in range(H):
        for j in range(W):
            for a in range(kh):
                for b in range(kw):
                    y, xx = i - pt + a, j - pl + b
                    if 0 <= y < H and 0 <= xx < W:
                        out[i, j] += x[y, xx] * kernel[a, b, :, 0]
    return out


def sepconv(x, p, name):
    d = depthwise_same(x, p[name + "/depthwise"].astype(np.float64))
    return d @ p[name + "/pointwise"][0, 0].astype(np.float64) + p[name + "/bias"]


def bn(x, p, name):
    g, b, m, v = (p[f"{name}/{s}"].astype(np.float64) for s in ("gamma", "beta", "mean", "var"))
    return g * (x - m) / np.sqrt(v + BN_EPS) + b


def maxpool_3x2_s2_same(x):
    H, W, C = x.shape
    Ho, pt = _pads(H, 3, 2)
    Wo, pl = _pads(W, 2, 2)
    out = np.full((Ho, Wo, C), -np.inf)
    for i in range(Ho):
        for j in range(Wo):
            for a in range(3):
                for b in range(2):
                    y, xx = 2 * i - pt + a, 2 * j - pl + b
                    if 0 <= y < H and 0 <= xx < W:
                        out[i, j] = np.maximum(out[i, j], x[y, xx])
    return out


def _sig(z):
    return 1.0 / (1.0 + np.exp(-z))


def lstm_direction(x, kernel, recurrent, bias, reverse):
    T, u = x.shape[0], recurrent.shape[0]
    h, c = np.zeros(u), np.zeros(u)
    out = np.zeros((T, u))
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = x[t] @ kernel + h @ recurrent + bias
        i, f, g, o = _sig(z[:u]), _sig(z[u : 2 * u]), np.tanh(z[2 * u : 3 * u]), _sig(z[3 * u :])
        c = f * c + i * g
        h = o * np.tanh(c)
        out[t] = h
    return out


def bilstm(x, p, name):
    f = lstm_direction(x, *(p[f"{name}/fwd/{k}"].astype(np.float64) for k in ("kernel", "recurrent", "bias")), reverse=False)
    b = lstm_direction(x, *(p[f"{name}/bwd/{k}"].astype(np.float64) for k in ("kernel", "recurrent", "bias")), reverse=True)
    return np.concatenate([f, b], axis=1)


def trunk(p, snippet, n_blocks):
    """snippet (H, W, 1) -> (H', W', 36) after the final separable conv + BN + ReLU."""
    x = np.maximum(bn(conv2d_same(snippet.astype(np.float64), p["conv0/kernel"].astype(np.float64), p["conv0/bias"], 1), p, "bn0"), 0)
    prev = x
    for b in range(1, n_blocks + 1):
        x = np.maximum(x, 0)
        x = np.maximum(bn(sepconv(x, p, f"b{b}/sep_a"), p, f"b{b}/bn_a"), 0)
        x = bn(sepconv(x, p, f"b{b}/sep_b"), p, f"b{b}/bn_b")
        x = maxpool_3x2_s2_same(x) + conv2d_same(prev, p[f"b{b}/res/kernel"].astype(np.float64), p[f"b{b}/res/bias"], 2)
        prev = x
    return np.maximum(bn(sepconv(x, p, "sep_f"), p, "bn_f"), 0)


def forward_one(p, snippet):
    """ResNetLSTM: snippet (H, W, 1) -> (H / 2**n, labels)."""
    n_blocks = sum(1 for k in p if k.endswith("/res/kernel"))
    x = trunk(p, snippet, n_blocks)
    x = x.reshape(x.shape[0], -1)  # Keras Reshape((-1, W*C)) of (H, W, C): feature = w*C + c
    x = bilstm(x, p, "lstm1")
    x = bilstm(x, p, "lstm2")
    x = np.maximum(x @ p["dense1/kernel"].astype(np.float64) + p["dense1/bias"], 0)
    x = bn(x, p, "bn_d")
    return _sig(x @ p["dense2/kernel"].astype(np.float64) + p["dense2/bias"])


def forward_one_1dconv(p, snippet):
    """ResNet1DConv: frequency mean, Conv1D(k = kernel.shape[0], same) + sigmoid (architectures.py:100-115)."""
    n_blocks = sum(1 for k in p if k.endswith("/res/kernel"))
    x = trunk(p, snippet, n_blocks).mean(axis=1)  # (T', 36)
    w = p["conv1d/kernel"].astype(np.float64)
    K, T = w.shape[0], x.shape[0]
    _, left = _pads(T, K, 1)
    out = np.zeros((T, w.shape[2]))
    for t in range(T):
        acc = p["conv1d/bias"].astype(np.float64).copy()
        for k in range(K):
            tt = t - left + k
            if 0 <= tt < T:
                acc += x[tt] @ w[k]
        out[t] = acc
    return _sig(out)
