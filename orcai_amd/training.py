"""Training engine: forward in training mode, backward, Adam on flat device buffers (reference ``train.py:155-219``,
``architectures.py:162-286``).  All arithmetic runs in HIP kernels behind the C ABI; torch supplies device memory,
tiny re-layouts of weight tensors (index/permute) and, for data parallel runs, the RCCL all-reduce of the flat
gradient bucket.

Parameters live in ONE flat fp32 buffer in Keras variable layouts (the order of ``ResNetLSTM.variable_spec``), the
gradients in a second flat buffer of the same shape (one all-reduce per step), Adam moments in two more.
"""

from __future__ import annotations

import numpy as np
import torch

from orcai_amd import _native as N
from orcai_amd.architectures import BN_EPS, DENSE_UNITS, FINAL_FILTERS, ResNetLSTM, lstm_column_permutation

L2_LAMBDA = 1e-3  # kernel_regularizer=l2(0.001) on the LSTM input kernels and Dense-128 (architectures.py:215,225,235)
BN_MOMENTUM = 0.99
MASK_VALUE = -1.0


class FlatParams:
    """Named views into flat device buffers (weights, gradients, Adam m / v)."""

    def __init__(self, model: ResNetLSTM, device):
        self.spec = [(n, tuple(s), t) for n, s, _, t in model.variable_spec()]
        self.offsets = {}
        off = 0
        for n, s, t in self.spec:
            if t:
                self.offsets[n] = (off, int(np.prod(s)), s)
                off += int(np.prod(s))
        self.n_trainable = off
        self.w = torch.empty(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.stats = {}  # BN moving mean / var (non-trainable)
        for n, s, t in self.spec:
            a = torch.from_numpy(np.ascontiguousarray(model.weights[n])).to(device)
            if t:
                self.W(n).copy_(a)
            else:
                self.stats[n] = a.clone()

    def W(self, name) -> torch.Tensor:
        o, n, s = self.offsets[name]
        return self.w[o : o + n].view(s)

    def G(self, name) -> torch.Tensor:
        o, n, s = self.offsets[name]
        return self.g[o : o + n].view(s)

    def to_model(self, model: ResNetLSTM) -> None:
        weights = {}
        for n, s, t in self.spec:
            weights[n] = (self.W(n) if t else self.stats[n]).detach().cpu().numpy().reshape(s)
        model.set_weights_dict(weights)


def _gemm(lib, A, sam, sak, B, sbk, sbn, C, M, Nn, K, alpha=1.0, accumulate=0, wreg=None, beta_w=0.0):
    N.check(lib.orcai_gemm_strided(A.data_ptr(), sam, sak, B.data_ptr(), sbk, sbn, C.data_ptr(), M, Nn, K, alpha, accumulate,
                                   None if wreg is None else wreg.data_ptr(), beta_w, N.stream_ptr()), "orcai_gemm_strided")


class HeadTrainer:
    """Training forward / backward of everything after the convolutional trunk: BN+ReLU of the final separable conv
    (Keras Reshape layout), two BiLSTM layers with dropout, Dense-128 + BN + dropout, Dense-labels + sigmoid, loss."""

    def __init__(self, model: ResNetLSTM, params: FlatParams):
        self.model, self.P = model, params
        self.lib = N.lib()
        self.u = model.lstm_units
        self.perm = torch.from_numpy(lstm_column_permutation(self.u)).to(params.w.device)
        self.inv_perm = torch.argsort(self.perm)
        self.cache = None

    # -- kernel-layout LSTM weights from the flat master copy (tiny device-side index ops)
    def _lstm_weights(self, layer):
        P = self.P
        Wc = torch.cat([P.W(f"lstm{layer}/fwd/kernel")[:, self.perm], P.W(f"lstm{layer}/bwd/kernel")[:, self.perm]], dim=1).contiguous()
        bc = torch.cat([P.W(f"lstm{layer}/fwd/bias")[self.perm], P.W(f"lstm{layer}/bwd/bias")[self.perm]]).contiguous()
        Uc = torch.stack([P.W(f"lstm{layer}/fwd/recurrent")[:, self.perm], P.W(f"lstm{layer}/bwd/recurrent")[:, self.perm]]).contiguous()
        return Wc, bc, Uc

    def forward(self, featv: torch.Tensor, masks: dict | None, rate: float) -> torch.Tensor:
        """featv: f32 cuda [n][T][W*36] = pre-BN output of the final separable conv.  Returns probabilities [n][T][labels]."""
        lib, P, u = self.lib, self.P, self.u
        st = N.stream_ptr()
        n, T, cols = featv.shape
        M = n * T
        dev = featv.device
        keep = 1.0 - rate
        c = {"featv": featv, "n": n, "T": T, "rate": rate, "masks": masks}
        f32 = dict(dtype=torch.float32, device=dev)
        # BN (batch statistics over snippet, time, frequency) + ReLU
        c["f_mean"], c["f_var"] = torch.empty(FINAL_FILTERS, **f32), torch.empty(FINAL_FILTERS, **f32)
        N.check(lib.orcai_bn_rows_stats(featv.data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(), st), "bn_rows_stats")
        x1 = torch.empty_like(featv)
        N.check(lib.orcai_bn_rows_apply(featv.data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(), P.W("bn_f/gamma").data_ptr(),
                                        P.W("bn_f/beta").data_ptr(), BN_EPS, 1, x1.data_ptr(), st), "bn_rows_apply")
        x, fin = x1, cols
        c["x1"] = x1
        for layer in (1, 2):
            Wc, bc, Uc = self._lstm_weights(layer)
            xz = torch.empty((n, T, 2, 4 * u), **f32)
            N.check(lib.orcai_gemm_bias_act(x.data_ptr(), Wc.data_ptr(), bc.data_ptr(), None, None, xz.data_ptr(), M, 8 * u, fin, 0, st), "gemm")
            h = torch.empty((n, T, 2 * u), **f32)
            gates = torch.empty((n, T, 2, 4 * u), **f32)
            cs = torch.empty((n, T, 2, u), **f32)
            N.check(lib.orcai_lstm_train_fwd(xz.data_ptr(), Uc.data_ptr(), n, T, u, h.data_ptr(), gates.data_ptr(), cs.data_ptr(), st), "lstm_train_fwd")
            c[f"lstm{layer}"] = dict(x=x, fin=fin, Wc=Wc, Uc=Uc, h=h, gates=gates, cs=cs)
            if masks is not None:
                hd = torch.empty_like(h)
                N.check(lib.orcai_mask_scale(h.data_ptr(), masks[f"drop{layer}"].data_ptr(), 1.0 / keep, h.numel(), hd.data_ptr(), st), "mask_scale")
            else:
                hd = h
            x, fin = hd, 2 * u
        c["h2d"] = x
        pre1 = torch.empty((n, T, DENSE_UNITS), **f32)
        N.check(lib.orcai_gemm_bias_act(x.data_ptr(), P.W("dense1/kernel").data_ptr(), P.W("dense1/bias").data_ptr(), None, None, pre1.data_ptr(), M, DENSE_UNITS,
                                        2 * u, 1, st), "gemm")
        c["pre1"] = pre1
        c["d_mean"], c["d_var"] = torch.empty(DENSE_UNITS, **f32), torch.empty(DENSE_UNITS, **f32)
        N.check(lib.orcai_bn_rows_stats(pre1.data_ptr(), M, DENSE_UNITS, DENSE_UNITS, c["d_mean"].data_ptr(), c["d_var"].data_ptr(), st), "bn_rows_stats")
        d1 = torch.empty_like(pre1)
        N.check(lib.orcai_bn_rows_apply(pre1.data_ptr(), M, DENSE_UNITS, DENSE_UNITS, c["d_mean"].data_ptr(), c["d_var"].data_ptr(), P.W("bn_d/gamma").data_ptr(),
                                        P.W("bn_d/beta").data_ptr(), BN_EPS, 0, d1.data_ptr(), st), "bn_rows_apply")
        if masks is not None:
            d1d = torch.empty_like(d1)
            N.check(lib.orcai_mask_scale(d1.data_ptr(), masks["drop3"].data_ptr(), 1.0 / keep, d1.numel(), d1d.data_ptr(), st), "mask_scale")
        else:
            d1d = d1
        c["d1d"] = d1d
        L = self.model.num_labels
        probs = torch.empty((n, T, L), **f32)
        N.check(lib.orcai_dense_sigmoid(d1d.data_ptr(), P.W("dense2/kernel").data_ptr(), P.W("dense2/bias").data_ptr(), M, DENSE_UNITS, L, probs.data_ptr(), st),
                "dense_sigmoid")
        c["probs"] = probs
        self.cache = c
        return probs

    def update_moving_stats(self) -> None:
        """moving = moving * 0.99 + batch * 0.01 (Keras BatchNormalization, biased batch variance)."""
        c, S = self.cache, self.P.stats
        for bn, mean, var in (("bn_f", c["f_mean"], c["f_var"]), ("bn_d", c["d_mean"], c["d_var"])):
            S[bn + "/mean"].mul_(BN_MOMENTUM).add_(mean, alpha=1 - BN_MOMENTUM)
            S[bn + "/var"].mul_(BN_MOMENTUM).add_(var, alpha=1 - BN_MOMENTUM)

    def loss_and_backward(self, labels: torch.Tensor) -> dict:
        """labels: f32 cuda [n][T][L] in {0,1} or -1 (masked).  Fills the head's slices of the flat gradient buffer,
        returns {"loss", "bce", "count", "correct", "dfeatv"} (device scalars stay on the device until .item())."""
        lib, P, u, c = self.lib, self.P, self.u, self.cache
        st = N.stream_ptr()
        n, T = c["n"], c["T"]
        M = n * T
        L = self.model.num_labels
        dev = labels.device
        f32 = dict(dtype=torch.float32, device=dev)
        keep = 1.0 - c["rate"]
        masks = c["masks"]
        acc = torch.zeros(4, dtype=torch.float64, device=dev)  # bce sum, count, correct, l2
        dz2 = torch.empty((M, L), **f32)
        N.check(lib.orcai_masked_bce(c["probs"].data_ptr(), labels.contiguous().data_ptr(), M * L, MASK_VALUE, acc.data_ptr(), dz2.data_ptr(), st), "masked_bce")
        # Dense(labels): dW2 = d1d^T dz2, db2 = colsum(dz2), dd1d = dz2 W2^T
        _gemm(lib, c["d1d"], 1, DENSE_UNITS, dz2, L, 1, P.G("dense2/kernel"), DENSE_UNITS, L, M)
        N.check(lib.orcai_colsum(dz2.data_ptr(), M, L, P.G("dense2/bias").data_ptr(), 0, st), "colsum")
        dd1 = torch.empty((M, DENSE_UNITS), **f32)
        _gemm(lib, dz2, L, 1, P.W("dense2/kernel"), 1, L, dd1, M, DENSE_UNITS, L)
        if masks is not None:
            N.check(lib.orcai_mask_scale(dd1.data_ptr(), masks["drop3"].data_ptr(), 1.0 / keep, dd1.numel(), dd1.data_ptr(), st), "mask_scale")
        # BN_d backward, ReLU backward
        dpre = torch.empty_like(dd1)
        N.check(lib.orcai_bn_rows_bwd(dd1.data_ptr(), c["pre1"].data_ptr(), M, DENSE_UNITS, DENSE_UNITS, c["d_mean"].data_ptr(), c["d_var"].data_ptr(),
                                      P.W("bn_d/gamma").data_ptr(), P.W("bn_d/beta").data_ptr(), BN_EPS, 0, P.G("bn_d/beta").data_ptr(), P.G("bn_d/gamma").data_ptr(),
                                      dpre.data_ptr(), st), "bn_rows_bwd")
        N.check(lib.orcai_relu_bwd(dpre.data_ptr(), c["pre1"].data_ptr(), dpre.numel(), dpre.data_ptr(), st), "relu_bwd")
        # Dense-128: dW1 = h2d^T dpre + 2 lambda W1, db1, dh2d = dpre W1^T
        _gemm(lib, c["h2d"], 1, 2 * u, dpre, DENSE_UNITS, 1, P.G("dense1/kernel"), 2 * u, DENSE_UNITS, M, wreg=P.W("dense1/kernel"), beta_w=2 * L2_LAMBDA)
        N.check(lib.orcai_colsum(dpre.data_ptr(), M, DENSE_UNITS, P.G("dense1/bias").data_ptr(), 0, st), "colsum")
        N.check(lib.orcai_l2_value(P.W("dense1/kernel").data_ptr(), P.W("dense1/kernel").numel(), L2_LAMBDA, acc[3:].data_ptr(), st), "l2_value")
        dh = torch.empty((M, 2 * u), **f32)
        _gemm(lib, dpre, DENSE_UNITS, 1, P.W("dense1/kernel"), 1, DENSE_UNITS, dh, M, 2 * u, DENSE_UNITS)
        for layer in (2, 1):
            lc = c[f"lstm{layer}"]
            if masks is not None:
                N.check(lib.orcai_mask_scale(dh.data_ptr(), masks[f"drop{layer}"].data_ptr(), 1.0 / keep, dh.numel(), dh.data_ptr(), st), "mask_scale")
            dxz = torch.empty((n, T, 2, 4 * u), **f32)
            N.check(lib.orcai_lstm_bwd(dh.data_ptr(), lc["gates"].data_ptr(), lc["cs"].data_ptr(), lc["Uc"].data_ptr(), n, T, u, dxz.data_ptr(), st), "lstm_bwd")
            fin = lc["fin"]
            # input kernels (both directions at once, permuted columns): dWc = x^T dxz
            dWc = torch.empty((fin, 8 * u), **f32)
            _gemm(lib, lc["x"], 1, fin, dxz, 8 * u, 1, dWc, fin, 8 * u, M)
            dbc = torch.empty(8 * u, **f32)
            N.check(lib.orcai_colsum(dxz.data_ptr(), M, 8 * u, dbc.data_ptr(), 0, st), "colsum")
            hp = torch.empty((n, T, 2, u), **f32)
            N.check(lib.orcai_lstm_hprev(lc["h"].data_ptr(), n, T, u, hp.data_ptr(), st), "lstm_hprev")
            for d, name in enumerate(("fwd", "bwd")):
                dU = torch.empty((u, 4 * u), **f32)
                _gemm(lib, hp.view(-1)[d * u :], 1, 2 * u, dxz.view(-1)[d * 4 * u :], 8 * u, 1, dU, u, 4 * u, M)
                P.G(f"lstm{layer}/{name}/recurrent").copy_(dU[:, self.inv_perm])
                Wk = P.W(f"lstm{layer}/{name}/kernel")
                P.G(f"lstm{layer}/{name}/kernel").copy_(dWc[:, d * 4 * u : (d + 1) * 4 * u][:, self.inv_perm] + 2 * L2_LAMBDA * Wk)
                P.G(f"lstm{layer}/{name}/bias").copy_(dbc[d * 4 * u : (d + 1) * 4 * u][self.inv_perm])
                N.check(lib.orcai_l2_value(Wk.data_ptr(), Wk.numel(), L2_LAMBDA, acc[3:].data_ptr(), st), "l2_value")
            dx = torch.empty((M, fin), **f32)
            _gemm(lib, dxz, 8 * u, 1, lc["Wc"], 1, 8 * u, dx, M, fin, 8 * u)
            dh = dx
        # BN_f + ReLU backward on the Keras-Reshape layout
        dfeatv = torch.empty_like(c["featv"])
        cols = c["featv"].shape[2]
        N.check(lib.orcai_bn_rows_bwd(dh.data_ptr(), c["featv"].data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(),
                                      P.W("bn_f/gamma").data_ptr(), P.W("bn_f/beta").data_ptr(), BN_EPS, 1, P.G("bn_f/beta").data_ptr(), P.G("bn_f/gamma").data_ptr(),
                                      dfeatv.data_ptr(), st), "bn_rows_bwd")
        return {"acc": acc, "dfeatv": dfeatv}


def adam_step(params: FlatParams, lr: float, step: int, gscale: float = 1.0, b1=0.9, b2=0.999, eps=1e-7) -> None:
    N.check(N.lib().orcai_adam_step(params.w.data_ptr(), params.g.data_ptr(), params.m.data_ptr(), params.v.data_ptr(), params.n_trainable, lr, b1, b2, eps, step,
                                    gscale, N.stream_ptr()), "orcai_adam_step")
