"""CPU restatement (torch autograd, CPU only) of one reference training step.  TEST INFRASTRUCTURE.

Follows ``/root/reference/src/orcAI/train.py:155-219`` (compile with Adam(lr), MaskedBinaryCrossentropy,
MaskedBinaryAccuracy; ``model.fit``), ``architectures.py:162-241`` in training mode (BatchNormalization with batch
statistics, Dropout after each BiLSTM and after Dense-128+BN, L2(1e-3) on the four LSTM input kernels and the
Dense-128 kernel) and ``architectures.py:244-286`` (masked loss / metric).  Keras/TF are absent from this image:
**parity unpinned at the Keras boundary**; Keras conventions restated here:
  * BN training: biased batch variance; moving = moving*0.99 + batch*0.01 (mean and the same biased variance);
  * Dropout: keep mask / (1 - rate); masks are passed in so the HIP path and the oracle share them;
  * BCE on probabilities clipped to [1e-7, 1-1e-7], mean over unmasked elements; all-masked batch -> NaN;
  * Adam (Keras 3): m,v EMA; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t * m / (sqrt(v) + 1e-7).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle.model_ref import BCE_EPS, BN_EPS, BN_MOMENTUM, L2, same_pad

L2_KERNELS = ("lstm1/fwd/kernel", "lstm1/bwd/kernel", "lstm2/fwd/kernel", "lstm2/bwd/kernel", "dense1/kernel")


def is_trainable(name: str) -> bool:
    return not (name.endswith("/mean") or name.endswith("/var"))


def _conv_same(x, kernel, bias, stride, groups=1):
    kh, kw = kernel.shape[0], kernel.shape[1]
    _, pt, pb = same_pad(x.shape[2], kh, stride)
    _, pl, pr = same_pad(x.shape[3], kw, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    w = kernel.permute(3, 2, 0, 1) if groups == 1 else kernel.permute(2, 3, 0, 1)
    return F.conv2d(x, w.contiguous(), bias, stride=stride, groups=groups)


def _bn_train(x, p, name, new_stats, axis=1):
    dims = [d for d in range(x.dim()) if d != axis]
    mean = x.mean(dim=dims)
    var = x.var(dim=dims, unbiased=False)
    new_stats[name + "/mean"] = (p[name + "/mean"] * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM))
    new_stats[name + "/var"] = (p[name + "/var"] * BN_MOMENTUM + var.detach() * (1 - BN_MOMENTUM))
    shape = [1] * x.dim()
    shape[axis] = -1
    inv = torch.rsqrt(var + BN_EPS) * p[name + "/gamma"]
    return x * inv.view(shape) + (p[name + "/beta"] - mean * inv).view(shape)


def _sepconv(x, p, name):
    c = x.shape[1]
    x = _conv_same(x, p[name + "/depthwise"], None, 1, groups=c)
    return _conv_same(x, p[name + "/pointwise"], p[name + "/bias"], 1)


def _maxpool_same(x, k=(3, 2), s=2, forced=None, key=None):
    _, pt, pb = same_pad(x.shape[2], k[0], s)
    _, pl, pr = same_pad(x.shape[3], k[1], s)
    x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    if forced is None or (key not in forced and "record" not in forced):
        return F.max_pool2d(x, kernel_size=k, stride=s)
    win = x.unfold(2, k[0], s).unfold(3, k[1], s)  # [B, C, Ho, Wo, kh, kw]
    B, C, Ho, Wo = win.shape[:4]
    win = win.reshape(B, C, Ho, Wo, k[0] * k[1])
    if "record" in forced:  # the branch THIS forward takes: first maximal element in window scan order (what max_pool2d's backward routes to)
        forced["record"][key] = torch.argmax(win.detach(), dim=4)
        return F.max_pool2d(x, kernel_size=k, stride=s)
    # forced branch: element forced[key][b, c, i, j] (0 .. kh*kw-1, window scan order) of every window is taken as its maximum
    return torch.gather(win, 4, forced[key].unsqueeze(-1)).squeeze(-1)


def _relu(x, forced=None, key=None):
    """ReLU, or -- forced[key] given -- the branch another implementation took: x * mask (0/1 tensor of x's shape).  Used to compare a reduced-
    precision path's gradient with the float64 gradient of the SAME piecewise-linear branch (tests/test_half_gpu.py): a pre-activation within
    f16 rounding of zero gets one mask in f64 and the other in f16, and every such flip moves that element's gradient by O(1)."""
    if forced is not None and "record" in forced:
        forced["record"][key] = (x.detach() > 0).to(x.dtype)
        return torch.relu(x)
    if forced is None or key not in forced:
        return torch.relu(x)
    return x * forced[key]


def _lstm_dir(x, W, U, b, reverse):
    B, T, _ = x.shape
    u = U.shape[0]
    h = x.new_zeros(B, u)
    c = x.new_zeros(B, u)
    xz = x @ W + b
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = xz[:, t] + h @ U
        i, f, g, o = z[:, :u], z[:, u : 2 * u], z[:, 2 * u : 3 * u], z[:, 3 * u :]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def _bilstm(x, p, name):
    fwd = _lstm_dir(x, p[name + "/fwd/kernel"], p[name + "/fwd/recurrent"], p[name + "/fwd/bias"], False)
    bwd = _lstm_dir(x, p[name + "/bwd/kernel"], p[name + "/bwd/recurrent"], p[name + "/bwd/bias"], True)
    return torch.cat([fwd, bwd], dim=2)


def forward_train(p: dict, x_nhwc: torch.Tensor, masks: dict | None, rate: float, n_blocks: int, forced: dict | None = None):
    """Training-mode forward.  p: dict of torch tensors (Keras layouts).  masks: {'drop1','drop2','drop3'} of 0/1
    tensors shaped like the tensors they multiply, or None for no dropout.  Returns (probabilities, new BN stats).
    forced: optional branches taken by another implementation -- 'relu/bn0', 'relu/b{i}/in', 'relu/b{i}/bn_a', 'relu/bn_f' (NCHW 0/1 masks),
    'relu/dense1' ([B][T][128]) and 'pool/b{i}' (int64 window element per pooled value); see _relu / _maxpool_same."""
    new_stats = {}
    keep = 1.0 - rate
    x = x_nhwc.permute(0, 3, 1, 2)
    x = _relu(_bn_train(_conv_same(x, p["conv0/kernel"], p["conv0/bias"], 1), p, "bn0", new_stats), forced, "relu/bn0")
    prev = x
    for b in range(1, n_blocks + 1):
        x = _relu(x, forced, f"relu/b{b}/in")
        x = _relu(_bn_train(_sepconv(x, p, f"b{b}/sep_a"), p, f"b{b}/bn_a", new_stats), forced, f"relu/b{b}/bn_a")
        x = _bn_train(_sepconv(x, p, f"b{b}/sep_b"), p, f"b{b}/bn_b", new_stats)
        x = _maxpool_same(x, forced=forced, key=f"pool/b{b}")
        x = x + _conv_same(prev, p[f"b{b}/res/kernel"], p[f"b{b}/res/bias"], 2)
        prev = x
    x = _relu(_bn_train(_sepconv(x, p, "sep_f"), p, "bn_f", new_stats), forced, "relu/bn_f")
    B, C, H, W = x.shape
    x = x.permute(0, 2, 3, 1).reshape(B, H, W * C)
    x = _bilstm(x, p, "lstm1")
    if masks is not None:
        x = x * masks["drop1"] / keep
    x = _bilstm(x, p, "lstm2")
    if masks is not None:
        x = x * masks["drop2"] / keep
    x = _relu(x @ p["dense1/kernel"] + p["dense1/bias"], forced, "relu/dense1")
    x = _bn_train(x, p, "bn_d", new_stats, axis=2)
    if masks is not None:
        x = x * masks["drop3"] / keep
    return torch.sigmoid(x @ p["dense2/kernel"] + p["dense2/bias"]), new_stats


def forward_train_1dconv(p: dict, x_nhwc: torch.Tensor, masks: dict | None, rate: float, n_blocks: int):
    """ResNet1DConv in training mode (architectures.py:54-115): Dropout after every residual block (the residual branch of the
    NEXT block reads the un-dropped tensor, :73-97), after the final BN+ReLU, then frequency mean and Conv1D(k = 36, same) + sigmoid.
    masks: {'block1'.., 'final'} 0/1 tensors shaped like the NCHW tensors they multiply, or None for no dropout."""
    new_stats = {}
    keep = 1.0 - rate
    x = x_nhwc.permute(0, 3, 1, 2)
    x = torch.relu(_bn_train(_conv_same(x, p["conv0/kernel"], p["conv0/bias"], 1), p, "bn0", new_stats))
    prev = x
    for b in range(1, n_blocks + 1):
        x = torch.relu(x)
        x = torch.relu(_bn_train(_sepconv(x, p, f"b{b}/sep_a"), p, f"b{b}/bn_a", new_stats))
        x = _bn_train(_sepconv(x, p, f"b{b}/sep_b"), p, f"b{b}/bn_b", new_stats)
        x = _maxpool_same(x)
        x = x + _conv_same(prev, p[f"b{b}/res/kernel"], p[f"b{b}/res/bias"], 2)
        prev = x
        if masks is not None:
            x = x * masks[f"block{b}"] / keep
    x = torch.relu(_bn_train(_sepconv(x, p, "sep_f"), p, "bn_f", new_stats))
    if masks is not None:
        x = x * masks["final"] / keep
    x = x.mean(dim=3).permute(0, 2, 1)  # (B, T', 36)
    w = p["conv1d/kernel"]
    K = w.shape[0]
    xp = F.pad(x.permute(0, 2, 1), ((K - 1) // 2, K // 2))
    y = F.conv1d(xp, w.permute(2, 1, 0).contiguous(), p["conv1d/bias"])
    return torch.sigmoid(y.permute(0, 2, 1)), new_stats


def loss_and_grads_1dconv(params_np: dict, x: np.ndarray, y: np.ndarray, masks_np: dict | None, rate: float, dtype=torch.float64):
    """One forward + backward of ResNet1DConv (masked BCE; the architecture has no weight regularisers)."""
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=is_trainable(k)) for k, v in params_np.items()}
    n_blocks = sum(1 for k in p if k.endswith("/res/kernel"))
    masks = None if masks_np is None else {k: torch.tensor(v, dtype=dtype) for k, v in masks_np.items()}
    probs, new_stats = forward_train_1dconv(p, torch.tensor(x, dtype=dtype), masks, rate, n_blocks)
    bce = masked_bce(torch.tensor(y, dtype=dtype), probs)
    bce.backward()
    grads = {k: v.grad.numpy() for k, v in p.items() if v.requires_grad}
    return {"loss": float(bce.detach()), "bce": float(bce.detach()), "grads": grads, "probs": probs.detach().numpy(), "new_stats": {k: v.numpy() for k, v in new_stats.items()}}


def masked_bce(y_true: torch.Tensor, y_pred: torch.Tensor, mask_value=-1.0) -> torch.Tensor:
    """architectures.py:262-270."""
    m = y_true != mask_value
    t = y_true[m]
    q = torch.clamp(y_pred[m], BCE_EPS, 1 - BCE_EPS)
    return torch.mean(-(t * torch.log(q) + (1 - t) * torch.log(1 - q)))


def l2_penalty(p: dict) -> torch.Tensor:
    return sum(L2 * torch.sum(p[k] ** 2) for k in L2_KERNELS)


def loss_and_grads(params_np: dict, x: np.ndarray, y: np.ndarray, masks_np: dict | None, rate: float, dtype=torch.float64, forced_np: dict | None = None):
    """One forward + backward.  Returns dict(loss, bce, grads {name: ndarray}, probs, new_stats).  forced_np: see forward_train."""
    p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=is_trainable(k)) for k, v in params_np.items()}
    n_blocks = sum(1 for k in p if k.endswith("/res/kernel"))
    masks = None if masks_np is None else {k: torch.tensor(v, dtype=dtype) for k, v in masks_np.items()}
    forced = None if forced_np is None else {k: (v if k == "record" else torch.tensor(np.asarray(v), dtype=torch.int64) if k.startswith("pool/") else torch.tensor(np.asarray(v), dtype=dtype))
                                             for k, v in forced_np.items()}  # {"record": {}} collects the branches this forward takes (torch tensors)
    probs, new_stats = forward_train(p, torch.tensor(x, dtype=dtype), masks, rate, n_blocks, forced)
    bce = masked_bce(torch.tensor(y, dtype=dtype), probs)
    loss = bce + l2_penalty(p)
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in p.items() if v.requires_grad}
    return {"loss": float(loss.detach()), "bce": float(bce.detach()), "grads": grads, "probs": probs.detach().numpy(), "new_stats": {k: v.numpy() for k, v in new_stats.items()}}


def adam_step_ref(w: np.ndarray, g: np.ndarray, m: np.ndarray, v: np.ndarray, t: int, lr: float, b1=0.9, b2=0.999, eps=1e-7):
    """Keras 3 Adam.update_step; t is the 1-based iteration."""
    m = m + (g - m) * (1 - b1)
    v = v + (g * g - v) * (1 - b2)
    alpha = lr * np.sqrt(1 - b2**t) / (1 - b1**t)
    return w - alpha * m / (np.sqrt(v) + eps), m, v


def reshape_labels_ref(labels: np.ndarray, n_filters: int) -> np.ndarray:
    """io.py:101-126: (T, L) frame labels -> (T / 2**n, L): mean over groups of 2**n rows, tf.round (half to even)."""
    f = 2**n_filters
    if labels.shape[0] % f != 0:
        raise ValueError("The number of rows in 'arr' must be divisible by 2**'n_filters'.")
    avg = labels.reshape(labels.shape[0] // f, f, labels.shape[1]).astype(np.float32).mean(axis=1)
    return np.round(avg).astype(np.float32)  # numpy rounds half to even like tf.round
