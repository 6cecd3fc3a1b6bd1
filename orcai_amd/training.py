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
        # BN moving mean / var (non-trainable): views into ONE flat buffer, mirrored by a flat buffer of the current step's batch
        # statistics in the same order, so that the moving-average update of every BatchNorm is one launch (orcai_ema_update)
        self.stat_offsets, soff = {}, 0
        for n, s, t in self.spec:
            if not t:
                self.stat_offsets[n] = (soff, int(np.prod(s)))
                soff += int(np.prod(s))
        self.stats_flat = torch.zeros(max(soff, 1), dtype=torch.float32, device=device)
        self.batch_flat = torch.zeros(max(soff, 1), dtype=torch.float32, device=device)
        self.stats = {n: self.stats_flat[o : o + k] for n, (o, k) in self.stat_offsets.items()}
        for n, s, t in self.spec:
            a = torch.from_numpy(np.ascontiguousarray(model.weights[n])).to(device)
            if t:
                self.W(n).copy_(a)
            else:
                self.stats[n].copy_(a)
        self.batch_flat.copy_(self.stats_flat)  # a BatchNorm that does not run this step leaves its moving statistics unchanged

    def W(self, name) -> torch.Tensor:
        o, n, s = self.offsets[name]
        return self.w[o : o + n].view(s)

    def G(self, name) -> torch.Tensor:
        o, n, s = self.offsets[name]
        return self.g[o : o + n].view(s)

    def B(self, name) -> torch.Tensor:
        """This step's batch statistic of a BatchNorm (`<bn>/mean` or `<bn>/var`): the kernels write it here."""
        o, k = self.stat_offsets[name]
        return self.batch_flat[o : o + k]

    def ema_all(self, momentum: float) -> None:
        """moving = moving * momentum + batch * (1 - momentum) for every BatchNorm at once (Keras BatchNormalization, biased variance)."""
        N.check(N.lib().orcai_ema_update(self.stats_flat.data_ptr(), self.batch_flat.data_ptr(), self.stats_flat.numel(), momentum, N.stream_ptr()), "ema_update")

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

    def __init__(self, model: ResNetLSTM, params: FlatParams, half: bool = False, grad_scale: float = 1.0):
        self.model, self.P = model, params
        self.lib = N.lib()
        self.u = model.lstm_units
        self.perm = torch.from_numpy(lstm_column_permutation(self.u)).to(params.w.device)
        self.inv_perm = torch.argsort(self.perm)
        self.cache = None
        # half: the forward GEMMs (LSTM input projections, Dense-128) run on f16 MFMA with f32 accumulation (orcai_h_gemm_bias_act);
        # the backward GEMMs stay on the f32 kernels.  grad_scale: static loss scale carried by every gradient (undone in Adam).
        self.half, self.grad_scale = bool(half), float(grad_scale)
        self._build_lstm_pack()

    def _build_lstm_pack(self):
        """Kernel-order copies of the LSTM variables (gate columns permuted, both directions side by side), refreshed by ONE
        orcai_pack_lstm launch per step: f32 [Fin][8u] input kernels, [8u] biases, [2][u][4u] recurrent kernels, and for the f16 path the
        transposed, zero-padded f16 [8u][roundup32(Fin)] input kernels orcai_h_gemm_bias_act multiplies with."""
        P, u, dev = self.P, self.u, self.P.w.device
        feat = self.model.stage_shapes()[-1][1] * FINAL_FILTERS
        desc, off32, off16 = [], 0, 0
        self.lstm_views = {}
        for layer, fin in ((1, feat), (2, 2 * u)):
            kp = (fin + 31) // 32 * 32
            v = {"fin": fin, "Wc": (off32, (fin, 8 * u))}
            off32 += fin * 8 * u
            v["bc"] = (off32, (8 * u,))
            off32 += 8 * u
            v["Uc"] = (off32, (2, u, 4 * u))
            off32 += 2 * u * 4 * u
            v["Wt"] = (off16, (8 * u, kp))
            off16 += 8 * u * kp
            for d, name in enumerate(("fwd", "bwd")):
                desc.append([P.offsets[f"lstm{layer}/{name}/kernel"][0], v["Wc"][0], fin, u, 8 * u, d * 4 * u, 0])
                desc.append([P.offsets[f"lstm{layer}/{name}/bias"][0], v["bc"][0], 1, u, 8 * u, d * 4 * u, 0])
                desc.append([P.offsets[f"lstm{layer}/{name}/recurrent"][0], v["Uc"][0] + d * u * 4 * u, u, u, 4 * u, 0, 0])
                if self.half:
                    desc.append([P.offsets[f"lstm{layer}/{name}/kernel"][0], v["Wt"][0], fin, u, kp, d * 4 * u, 1])
            self.lstm_views[layer] = v
        self.lstm_desc = torch.tensor(desc, dtype=torch.int32, device=dev).contiguous()
        self.lstm32 = torch.empty(off32, dtype=torch.float32, device=dev)
        self.lstm16 = torch.zeros(off16 if self.half else 8, dtype=torch.float16, device=dev)  # the k padding stays zero

    def _lv(self, layer, key):
        o, shape = self.lstm_views[layer][key]
        buf = self.lstm16 if key == "Wt" else self.lstm32
        return buf[o : o + int(np.prod(shape))].view(shape)

    def _gemm_fwd(self, x, W, bias, out, M, Nn, K, act, Wt=None):
        """out = act(x W + bias): f32 MFMA, or (half) f16 MFMA on a transposed, zero-padded f16 copy of W made here."""
        st = N.stream_ptr()
        if not self.half:
            N.check(self.lib.orcai_gemm_bias_act(x.data_ptr(), W.data_ptr(), bias.data_ptr(), None, None, out.data_ptr(), M, Nn, K, act, st), "gemm")
            return
        if Wt is None:
            Wt = torch.zeros((Nn, (K + 31) // 32 * 32), dtype=torch.float16, device=W.device)
            Wt[:, :K] = W.t()
        N.check(self.lib.orcai_h_gemm_bias_act(x.data_ptr(), Wt.data_ptr(), bias.data_ptr(), None, None, out.data_ptr(), M, Nn, K, act, st), "h_gemm")

    def _lstm_weights(self, layer):
        """Views of this step's kernel-order copies (made by the orcai_pack_lstm launch at the start of forward)."""
        return self._lv(layer, "Wc"), self._lv(layer, "bc"), self._lv(layer, "Uc")

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
        N.check(lib.orcai_pack_lstm(P.w.data_ptr(), self.lstm_desc.data_ptr(), int(self.lstm_desc.shape[0]), self.lstm32.data_ptr(), self.lstm16.data_ptr(), st), "pack_lstm")
        # BN (batch statistics over snippet, time, frequency) + ReLU
        c["f_mean"], c["f_var"] = P.B("bn_f/mean"), P.B("bn_f/var")
        N.check(lib.orcai_bn_rows_stats(featv.data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(), st), "bn_rows_stats")
        x1 = torch.empty_like(featv)
        N.check(lib.orcai_bn_rows_apply(featv.data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(), P.W("bn_f/gamma").data_ptr(),
                                        P.W("bn_f/beta").data_ptr(), BN_EPS, 1, x1.data_ptr(), st), "bn_rows_apply")
        x, fin = x1, cols
        c["x1"] = x1
        for layer in (1, 2):
            Wc, bc, Uc = self._lstm_weights(layer)
            xz = torch.empty((n, T, 2, 4 * u), **f32)
            self._gemm_fwd(x, Wc, bc, xz, M, 8 * u, fin, 0, Wt=self._lv(layer, "Wt") if self.half else None)
            h = torch.empty((n, T, 2 * u), **f32)
            gates = torch.empty((n, T, 2, 4 * u), **f32)
            cs = torch.empty((n, T, 2, u), **f32)
            fwd = lib.orcai_h_lstm_train_fwd if self.half else lib.orcai_lstm_train_fwd  # f16 path: the recurrent product on f16 MFMA
            N.check(fwd(xz.data_ptr(), Uc.data_ptr(), n, T, u, h.data_ptr(), gates.data_ptr(), cs.data_ptr(), st), "lstm_train_fwd")
            c[f"lstm{layer}"] = dict(x=x, fin=fin, Wc=Wc, Uc=Uc, h=h, gates=gates, cs=cs)
            if masks is not None:
                hd = torch.empty_like(h)
                N.check(lib.orcai_mask_scale(h.data_ptr(), masks[f"drop{layer}"].data_ptr(), 1.0 / keep, h.numel(), hd.data_ptr(), st), "mask_scale")
            else:
                hd = h
            x, fin = hd, 2 * u
        c["h2d"] = x
        pre1 = torch.empty((n, T, DENSE_UNITS), **f32)
        self._gemm_fwd(x, P.W("dense1/kernel"), P.W("dense1/bias"), pre1, M, DENSE_UNITS, 2 * u, 1)
        c["pre1"] = pre1
        c["d_mean"], c["d_var"] = P.B("bn_d/mean"), P.B("bn_d/var")
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

    def loss_and_backward(self, labels: torch.Tensor, loss_weight: torch.Tensor | None = None) -> dict:
        """labels: f32 cuda [n][T][L] in {0,1} or -1 (masked).  Fills the head's slices of the flat gradient buffer,
        returns {"loss", "bce", "count", "correct", "dfeatv"} (device scalars stay on the device until .item()).
        loss_weight: optional device float[1], the batch mean of Keras' class_weight sample weights (orcai_masked_bce_w)."""
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
        N.check(lib.orcai_masked_bce_w(c["probs"].data_ptr(), labels.contiguous().data_ptr(), M * L, MASK_VALUE, acc.data_ptr(), dz2.data_ptr(),
                                       None if loss_weight is None else loss_weight.data_ptr(), self.grad_scale, st), "masked_bce")
        l2g = 2 * L2_LAMBDA * self.grad_scale  # the regularisers' gradients join loss-scaled gradients
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
        _gemm(lib, c["h2d"], 1, 2 * u, dpre, DENSE_UNITS, 1, P.G("dense1/kernel"), 2 * u, DENSE_UNITS, M, wreg=P.W("dense1/kernel"), beta_w=l2g)
        N.check(lib.orcai_colsum(dpre.data_ptr(), M, DENSE_UNITS, P.G("dense1/bias").data_ptr(), 0, st), "colsum")
        # the value of the L2 penalty over the five regularised kernels: one launch over their slices of the flat weight buffer
        l2_names = ["dense1/kernel"] + [f"lstm{layer}/{name}/kernel" for layer in (1, 2) for name in ("fwd", "bwd")]
        offs = (N.c_i64 * len(l2_names))(*[P.offsets[n][0] for n in l2_names])
        cnts = (N.c_i64 * len(l2_names))(*[P.offsets[n][1] for n in l2_names])
        N.check(lib.orcai_l2_values(P.w.data_ptr(), offs, cnts, len(l2_names), L2_LAMBDA, acc[3:].data_ptr(), st), "l2_values")
        unpack, alive = [], []  # kernel-order LSTM gradients -> the Keras-layout gradient buffer: ONE launch for both layers at the end (sources kept alive until then)
        dh = torch.empty((M, 2 * u), **f32)
        _gemm(lib, dpre, DENSE_UNITS, 1, P.W("dense1/kernel"), 1, DENSE_UNITS, dh, M, 2 * u, DENSE_UNITS)
        for layer in (2, 1):
            lc = c[f"lstm{layer}"]
            if masks is not None:
                N.check(lib.orcai_mask_scale(dh.data_ptr(), masks[f"drop{layer}"].data_ptr(), 1.0 / keep, dh.numel(), dh.data_ptr(), st), "mask_scale")
            dxz = torch.empty((n, T, 2, 4 * u), **f32)
            bwd = lib.orcai_h_lstm_bwd if self.half else lib.orcai_lstm_bwd
            N.check(bwd(dh.data_ptr(), lc["gates"].data_ptr(), lc["cs"].data_ptr(), lc["Uc"].data_ptr(), n, T, u, dxz.data_ptr(), st), "lstm_bwd")
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
                Wk = P.W(f"lstm{layer}/{name}/kernel")
                # kernel-order gradients back into the Keras-layout gradient buffer (+ the L2 term of the input kernel)
                unpack.append(N.UnpackDesc(dU.data_ptr(), 4 * u, 0, u, P.G(f"lstm{layer}/{name}/recurrent").data_ptr(), None, 0.0))
                unpack.append(N.UnpackDesc(dWc.data_ptr(), 8 * u, d * 4 * u, fin, P.G(f"lstm{layer}/{name}/kernel").data_ptr(), Wk.data_ptr(), l2g))
                unpack.append(N.UnpackDesc(dbc.data_ptr(), 8 * u, d * 4 * u, 1, P.G(f"lstm{layer}/{name}/bias").data_ptr(), None, 0.0))
                alive += [dU, dWc, dbc]
            dx = torch.empty((M, fin), **f32)
            _gemm(lib, dxz, 8 * u, 1, lc["Wc"], 1, 8 * u, dx, M, fin, 8 * u)
            dh = dx
        N.check(lib.orcai_unpack_lstm_grads((N.UnpackDesc * len(unpack))(*unpack), len(unpack), u, st), "unpack_lstm_grads")
        del alive
        # BN_f + ReLU backward on the Keras-Reshape layout
        dfeatv = torch.empty_like(c["featv"])
        cols = c["featv"].shape[2]
        N.check(lib.orcai_bn_rows_bwd(dh.data_ptr(), c["featv"].data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(),
                                      P.W("bn_f/gamma").data_ptr(), P.W("bn_f/beta").data_ptr(), BN_EPS, 1, P.G("bn_f/beta").data_ptr(), P.G("bn_f/gamma").data_ptr(),
                                      dfeatv.data_ptr(), st), "bn_rows_bwd")
        return {"acc": acc, "dfeatv": dfeatv}


class Conv1DHeadTrainer:
    """ResNet1DConv head in training mode (architectures.py:100-115): BN_f + ReLU of the final separable conv (Keras Reshape
    layout), Dropout, ReduceFrequencyMean, Conv1D(num_labels, k = 36, same) + sigmoid, masked BCE (no weight regularisers)."""

    def __init__(self, model, params: FlatParams):
        self.model, self.P = model, params
        self.lib = N.lib()
        self.cache = None

    def forward(self, featv: torch.Tensor, masks: dict | None, rate: float) -> torch.Tensor:
        lib, P, st = self.lib, self.P, N.stream_ptr()
        n, T, cols = featv.shape
        M, Wd = n * T, cols // FINAL_FILTERS
        f32 = dict(dtype=torch.float32, device=featv.device)
        keep = 1.0 - rate
        c = {"featv": featv, "n": n, "T": T, "rate": rate, "masks": masks, "Wd": Wd}
        c["f_mean"], c["f_var"] = P.B("bn_f/mean"), P.B("bn_f/var")
        N.check(lib.orcai_bn_rows_stats(featv.data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(), st), "bn_rows_stats")
        x1 = torch.empty_like(featv)
        N.check(lib.orcai_bn_rows_apply(featv.data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(), P.W("bn_f/gamma").data_ptr(),
                                        P.W("bn_f/beta").data_ptr(), BN_EPS, 1, x1.data_ptr(), st), "bn_rows_apply")
        if masks is not None:
            N.check(lib.orcai_mask_scale(x1.data_ptr(), masks["final"].data_ptr(), 1.0 / keep, x1.numel(), x1.data_ptr(), st), "mask_scale")
        fm = torch.empty((n, T, FINAL_FILTERS), **f32)
        N.check(lib.orcai_freq_mean(x1.data_ptr(), M, Wd, FINAL_FILTERS, fm.data_ptr(), st), "freq_mean")
        L = self.model.num_labels
        probs = torch.empty((n, T, L), **f32)
        N.check(lib.orcai_conv1d_sigmoid(fm.data_ptr(), P.W("conv1d/kernel").data_ptr(), P.W("conv1d/bias").data_ptr(), n, T, FINAL_FILTERS, FINAL_FILTERS, L,
                                         probs.data_ptr(), st), "conv1d_sigmoid")
        c["fm"], c["probs"] = fm, probs
        self.cache = c
        return probs

    def update_moving_stats(self) -> None:
        c, S = self.cache, self.P.stats
        S["bn_f/mean"].mul_(BN_MOMENTUM).add_(c["f_mean"], alpha=1 - BN_MOMENTUM)
        S["bn_f/var"].mul_(BN_MOMENTUM).add_(c["f_var"], alpha=1 - BN_MOMENTUM)

    def loss_and_backward(self, labels: torch.Tensor, loss_weight: torch.Tensor | None = None) -> dict:
        lib, P, c, st = self.lib, self.P, self.cache, N.stream_ptr()
        n, T, Wd = c["n"], c["T"], c["Wd"]
        M, L = n * T, self.model.num_labels
        f32 = dict(dtype=torch.float32, device=labels.device)
        keep = 1.0 - c["rate"]
        acc = torch.zeros(4, dtype=torch.float64, device=labels.device)  # bce sum, count, correct, l2 (stays 0)
        dz = torch.empty((M, L), **f32)
        N.check(lib.orcai_masked_bce_w(c["probs"].data_ptr(), labels.contiguous().data_ptr(), M * L, MASK_VALUE, acc.data_ptr(), dz.data_ptr(),
                                       None if loss_weight is None else loss_weight.data_ptr(), 1.0, st), "masked_bce")
        dfm = torch.empty((n, T, FINAL_FILTERS), **f32)
        N.check(lib.orcai_conv1d_bwd(c["fm"].data_ptr(), P.W("conv1d/kernel").data_ptr(), dz.data_ptr(), n, T, FINAL_FILTERS, FINAL_FILTERS, L,
                                     P.G("conv1d/kernel").data_ptr(), dfm.data_ptr(), st), "conv1d_bwd")  # the gradient buffer was zeroed at the start of the step
        N.check(lib.orcai_colsum(dz.data_ptr(), M, L, P.G("conv1d/bias").data_ptr(), 0, st), "colsum")
        dx1 = torch.empty_like(c["featv"])
        N.check(lib.orcai_freq_mean_bwd(dfm.data_ptr(), M, Wd, FINAL_FILTERS, dx1.data_ptr(), st), "freq_mean_bwd")
        if c["masks"] is not None:
            N.check(lib.orcai_mask_scale(dx1.data_ptr(), c["masks"]["final"].data_ptr(), 1.0 / keep, dx1.numel(), dx1.data_ptr(), st), "mask_scale")
        dfeatv = torch.empty_like(c["featv"])
        cols = c["featv"].shape[2]
        N.check(lib.orcai_bn_rows_bwd(dx1.data_ptr(), c["featv"].data_ptr(), M, cols, FINAL_FILTERS, c["f_mean"].data_ptr(), c["f_var"].data_ptr(),
                                      P.W("bn_f/gamma").data_ptr(), P.W("bn_f/beta").data_ptr(), BN_EPS, 1, P.G("bn_f/beta").data_ptr(), P.G("bn_f/gamma").data_ptr(),
                                      dfeatv.data_ptr(), st), "bn_rows_bwd")
        return {"acc": acc, "dfeatv": dfeatv}


def adam_step(params: FlatParams, lr: float, step: int, gscale: float = 1.0, b1=0.9, b2=0.999, eps=1e-7) -> None:
    N.check(N.lib().orcai_adam_step(params.w.data_ptr(), params.g.data_ptr(), params.m.data_ptr(), params.v.data_ptr(), params.n_trainable, lr, b1, b2, eps, step,
                                    gscale, N.stream_ptr()), "orcai_adam_step")


class TrunkTrainer:
    """Training forward / backward of the convolutional trunk on padded channel-quad planes.  The forward stores, per
    BatchNorm, the pre-normalisation tensor v and the batch statistics, and materialises y = [relu](BN(v)); the backward
    walks the layers in reverse with the kernels of csrc/train_trunk.hip (correctness first: u = dw(relu(x)) is
    re-materialised for the pointwise weight gradient, and depthwise-only / pointwise-only passes reuse the
    separable-conv kernel with identity factors)."""

    def __init__(self, model: ResNetLSTM, params: FlatParams, half: bool = False):
        self.model, self.P = model, params
        self.lib = N.lib()
        self.k = model.kernel_size
        self.R = self.k // 2
        self.dev = params.w.device
        self.buf = {}
        self.B = None
        self.consts = {}
        # half: activations and activation gradients as f16 channel-octet planes, contractions on f16 MFMA (csrc/half_fwd.hip,
        # half_bwd.hip); the launchers have the f32 path's signatures, only the weight operands differ (f16 copies packed once per step)
        self.half = bool(half)
        self.G = 8 if self.half else 4  # channels per 16-byte pixel vector
        self.adt = torch.float16 if self.half else torch.float32
        if self.half and getattr(model, "architecture", "") != "ResNetLSTM":
            raise NotImplementedError("the f16 path implements ResNetLSTM only")
        self._own_scratch = torch.zeros(8 * 16 * 32, dtype=torch.float64, device=self.dev)  # 8 doubles per channel quad (<= 64 channels) x the 32 accumulator copies of orcai_bn_planes_stats
        self.scratch = self._own_scratch
        # One clear per step instead of one 5-us zero-fill launch per reduction (~30 per step): every launcher that accumulates gets its own 32-KiB slot of
        # an arena that Trainer.forward_backward clears with ONE launch (orcai_scratch_arena); the launchers skip their own fill for a slot nobody has taken
        # since (include/orcai_hip.h).  self.scratch / self.res_scratch always name the slot of the most recent producer: its consumers read it there.
        self.arena_slots = 96
        self.arena = torch.empty(self.arena_slots * 4096, dtype=torch.float64, device=self.dev)
        self._slot = self.arena_slots  # no step in flight: _fresh() hands out the private buffers
        self.use_arena = True  # A/B: tools/ab_flags.py
        # ResNet1DConv drops out the output of every residual block (architectures.py:97); ResNetLSTM has no Dropout in the trunk
        self.block_rate = float(model.dropout_rate) if getattr(model, "architecture", "") == "ResNet1DConv" else 0.0
        self.block_masks = None  # list of 0/1 plane tensors (one per block) for the current step, or None
        self._own_res_scratch = torch.zeros(8 * 16, dtype=torch.float64, device=self.dev)  # planes_sum of the residual bias gradient: pool_bwd_bn's sums stay in self.scratch
        self.res_scratch = self._own_res_scratch
        self.stats_in_epilogue = True  # block-1-shaped separable convs reduce their BatchNorm statistics in the epilogue (A/B: tools/ab_train_order.py)
        self.dgrad_first = True  # order of a separable conv's backward kernels (A/B: tools/ab_train_order.py)
        self.bias_in_pool = True  # residual bias gradients reduced inside the pooling backward (no planes_sum pass over dout)
        self.conv0_two_pass = True  # entry conv: statistics pass + conv/bn0/ReLU pass, v0 never stored, rebuilt in the backward pass (A/B: tools/ab_flags.py)
        self.apply_on_load = True  # bn_a + ReLU applied where sep_b / its depthwise weight gradient load their input: y_a is never written (A/B: tools/ab_flags.py)
        self.dgrad_epilogues = True  # BatchNorm backward sums / ReLU backward in the epilogue of the input-gradient passes (A/B: tools/ab_flags.py)
        self.conv0_stats_from_input = True  # f16 path: bn0's statistics from the snippet (orcai_conv0_stats_march) instead of a pass over the stored v0 (A/B: tools/ab_sweep_flags.py)
        self.conv0_march = True  # entry conv statistics on the marching kernel (A/B: tools/ab_flags.py; its backward twin: orcai_conv0_march)
        self.conv0_in_dgrad = True  # block 1's first conv: y0 rebuilt from the snippet inside the marching depthwise backward, bn0's sums in its epilogue (A/B: tools/ab_flags.py)
        self.bn0_sums_ready = False
        self._resq = None
        self.fused_dw_bwd = True  # input gradient + its epilogue extra + depthwise weight gradient of a k = 3 separable conv in one marching pass (A/B: tools/ab_flags.py)
        self.fused_pw_wgrad = True  # BN backward apply + du + pointwise weight gradient in one pass where the layer is narrow enough (A/B: tools/ab_train.py)
        self.fused_stats_under_capture = True  # the epilogue statistics also inside a captured step (tools/debug_graph_divergence.py)
        self.partials = torch.empty(512 * 64 * 64, dtype=torch.float32, device=self.dev)  # per-workgroup partial weight gradients (outer_reduce)

    # ------------------------------------------------------------- helpers
    def begin_step(self) -> None:
        """Clear the accumulator arena (one launch) and start handing out its slots."""
        if self.use_arena:
            N.check(self.lib.orcai_scratch_arena(self.arena.data_ptr(), self.arena.numel() * 8, N.stream_ptr()), "scratch_arena")
            self._slot = 0

    def end_step(self) -> None:
        """No launcher may skip its zero fill outside a step (the arena's memory may be anybody's by then)."""
        N.lib().orcai_scratch_arena(None, 0, None)
        self._slot = self.arena_slots
        self.scratch, self.res_scratch = self._own_scratch, self._own_res_scratch

    def _slot_view(self, own):
        if self._slot < self.arena_slots:
            v = self.arena[self._slot * 4096 : (self._slot + 1) * 4096]
            self._slot += 1
            return v
        return own

    def _fresh(self):
        """A cleared accumulator for the NEXT producer launch (self.scratch); call it in front of every launcher that accumulates into self.scratch, never
        in front of one that reads sums an earlier launcher left there."""
        self.scratch = self._slot_view(self._own_scratch)

    def _fresh_res(self):
        self.res_scratch = self._slot_view(self._own_res_scratch)

    def _planes(self, B, c, h, w):
        G = self.G
        return torch.zeros((B, (c + G - 1) // G, h + 2 * self.R, self.model.padded_width(w), G), dtype=self.adt, device=self.dev)

    def _fn(self, name):
        """C-ABI launcher `orcai_<name>` (f32 quad planes) or its f16 octet-plane twin `orcai_h_<name>` (same argument list)."""
        if self.half:
            name = {"sepconv_planes_u": "sepconv", "pool_res_add_bn": "pool_res_add"}.get(name, name)
            return getattr(self.lib, "orcai_h_" + name)
        return getattr(self.lib, "orcai_" + name)

    # weight operands of the trunk kernels: f32 master views / packed f32 copies, or packed f16 copies (half)
    def _w_dw(self, name, reverse=False):
        return self._packed(1 if reverse else 0, name + "/depthwise", (-1,))

    def _w_pw(self, name):  # forward contraction operand of a pointwise / residual kernel [Cin][Cout]
        return self._packed(2, name, (-1,)) if self.half else self.P.W(name)

    def _w_pwT(self, name, cin, cout):  # input-gradient operand
        return self._packed(3, name, (-1,)) if self.half else self._packed(2, name, (cout, cin))

    def _w_eye(self, c):
        return self._packed(4, f"eye{c}", (-1,)) if self.half else self._eye(c)

    def _w_ones_dw(self, c):
        return self._packed(5, f"ones{c}", (-1,)) if self.half else self._ones(4 * ((c + 3) // 4))

    def _const(self, key, make):
        if key not in self.consts:
            self.consts[key] = make()
        return self.consts[key]

    def _ones(self, n):
        return self._const(("ones", n), lambda: torch.ones(n, dtype=torch.float32, device=self.dev))

    def _zeros(self, n):
        return self._const(("zeros", n), lambda: torch.zeros(n, dtype=torch.float32, device=self.dev))

    def _eye(self, c):
        return self._const(("eye", c), lambda: torch.eye(c, dtype=torch.float32, device=self.dev).contiguous())

    def _sep(self, x, Cin, H, W, ktap, relu_in, dw, pw, shift, Cout, out, layout=0, H2=0, W2=0, u_out=None):
        N.check(self._fn("sepconv_planes_u")(x.data_ptr(), self.B, Cin, H, W, self.k, ktap, relu_in, dw.data_ptr(), pw.data_ptr(), self._ones(64).data_ptr(),
                                                shift.data_ptr(), Cout, 0, layout, H2, W2, out.data_ptr(), None if u_out is None else u_out.data_ptr(),
                                                N.stream_ptr()), "orcai_sepconv_planes_u")

    def _sep_stats(self, x, Cin, H, W, relu_in, dw, pw, shift, Cout, out, u_out) -> bool:
        """Training forward of a k = 3 separable conv with the batch statistics of its output reduced in the kernel's epilogue (sums into
        self.scratch); False when the shape is not one of the strip-tile kernel's: the caller then runs the two separate launches."""
        if not self.fused_stats_under_capture and torch.cuda.is_current_stream_capturing():
            return False  # A/B switch of tools/debug_graph_divergence.py
        self._fresh()
        rc = (self.lib.orcai_h_sepconv_stats if self.half else self.lib.orcai_sepconv_planes_stats)(x.data_ptr(), self.B, Cin, H, W, relu_in, dw.data_ptr(), pw.data_ptr(), self._ones(64).data_ptr(), shift.data_ptr(), Cout,
                                                 out.data_ptr(), u_out.data_ptr(), self.scratch.data_ptr(), N.stream_ptr())
        if rc == N.E_UNSUPPORTED:
            return False
        N.check(rc, "orcai_sepconv_planes_stats")
        return True

    def _bn_fwd(self, v, bn, C, H, W, relu, y, sums_in_shards=False):
        """Batch statistics of v; y = [relu](BN(v)) is materialised unless y is None (the consumer applies BN on the fly)."""
        lib, P, st = self.lib, self.P, N.stream_ptr()
        mean, var = P.B(bn + "/mean"), P.B(bn + "/var")  # this step's batch statistics live in the flat buffer the EMA update reads
        if sums_in_shards:  # the producing kernel left the sums in self.scratch (orcai_sepconv_planes_stats)
            N.check((lib.orcai_h_bn_finish_sharded if self.half else lib.orcai_bn_finish_sharded)(self.scratch.data_ptr(), self.B, C, H, W, mean.data_ptr(), var.data_ptr(), st), "bn_finish_sharded")
        else:
            self._fresh()
            N.check(self._fn("bn_planes_stats")(v.data_ptr(), self.B, C, H, W, self.k, self.scratch.data_ptr(), mean.data_ptr(), var.data_ptr(), st), "bn_planes_stats")
        self.stats[bn] = (mean, var)
        if y is None:
            return
        N.check(self._fn("bn_planes_apply")(v.data_ptr(), self.B, C, H, W, self.k, mean.data_ptr(), var.data_ptr(), P.W(bn + "/gamma").data_ptr(),
                                            P.W(bn + "/beta").data_ptr(), BN_EPS, relu, y.data_ptr(), st), "bn_planes_apply")
        self.stats[bn] = (mean, var)

    def _bn_apply(self, v, bn, C, H, W, relu, y):
        """y = [relu](BN(v)) with the batch statistics _bn_fwd left in self.stats[bn] (the fallback of the apply-on-load path)."""
        P = self.P
        mean, var = self.stats[bn]
        N.check(self._fn("bn_planes_apply")(v.data_ptr(), self.B, C, H, W, self.k, mean.data_ptr(), var.data_ptr(), P.W(bn + "/gamma").data_ptr(),
                                            P.W(bn + "/beta").data_ptr(), BN_EPS, relu, y.data_ptr(), N.stream_ptr()), "bn_planes_apply")

    def _sep_stats_bn(self, v_in, bn_in, Cin, H, W, dw, pw, shift, Cout, out, u_out) -> bool:
        """_sep_stats whose input is the PRE-normalisation tensor of BatchNorm `bn_in` (+ ReLU): normalised on load, never materialised."""
        P = self.P
        mean, var = self.stats[bn_in]
        self._fresh()
        rc = (self.lib.orcai_h_sepconv_stats_bn if self.half else self.lib.orcai_sepconv_planes_stats_bn)(v_in.data_ptr(), self.B, Cin, H, W, mean.data_ptr(), var.data_ptr(), P.W(bn_in + "/gamma").data_ptr(),
                                                    P.W(bn_in + "/beta").data_ptr(), BN_EPS, dw.data_ptr(), pw.data_ptr(), self._ones(64).data_ptr(), shift.data_ptr(), Cout,
                                                    out.data_ptr(), u_out.data_ptr(), self.scratch.data_ptr(), N.stream_ptr())
        if rc == N.E_UNSUPPORTED:
            return False
        N.check(rc, "orcai_sepconv_planes_stats_bn")
        return True

    def _bn_bwd(self, dy, v, bn, C, H, W, relu, dv):
        lib, P, st = self.lib, self.P, N.stream_ptr()
        mean, var = self.stats[bn]
        self._fresh()
        N.check(lib.orcai_bn_planes_bwd(dy.data_ptr(), v.data_ptr(), self.B, C, H, W, self.k, mean.data_ptr(), var.data_ptr(), P.W(bn + "/gamma").data_ptr(),
                                        P.W(bn + "/beta").data_ptr(), BN_EPS, relu, self.scratch.data_ptr(), P.G(bn + "/beta").data_ptr(),
                                        P.G(bn + "/gamma").data_ptr(), dv.data_ptr(), st), "bn_planes_bwd")

    def _alloc(self, B):
        if self.B == B:
            return
        self.B = B
        m = self.model
        shapes = m.stage_shapes()
        b = {}
        h, w, _ = shapes[0]
        b["v0"], b["y0"] = self._planes(B, 16, h, w), self._planes(B, 16, h, w)
        b["rq1"] = self._planes(B, 16, shapes[1][0], shapes[1][1])  # block 1's residual gradient w.r.t. y0 at the even pixels (compact; see conv0_in_dgrad)
        for i, f in enumerate(m.filters, start=1):
            h, w, cprev = shapes[i - 1]
            for n in ("va", "ya", "vb"):  # y_b = BN_b(v_b) is never materialised (the pooling kernels apply BN on the fly)
                b[f"{n}{i}"] = self._planes(B, f, h, w)
            b[f"u_a{i}"], b[f"du_a{i}"] = self._planes(B, cprev, h, w), self._planes(B, cprev, h, w)  # depthwise output / its gradient (sep_a)
            b[f"u_b{i}"], b[f"du_b{i}"] = self._planes(B, f, h, w), self._planes(B, f, h, w)
            b[f"prev{i}"] = self._planes(B, f, shapes[i][0], shapes[i][1])
            if self.block_rate > 0.0:
                b[f"prevd{i}"] = self._planes(B, f, shapes[i][0], shapes[i][1])
            # gradient planes (only interiors are ever written, so the zero pads persist from step to step)
            b[f"dyb{i}"], b[f"dya{i}"], b[f"dr{i}"] = self._planes(B, f, h, w), self._planes(B, f, h, w), self._planes(B, cprev, h, w)
        h, w, c = shapes[-1]
        b["u_f"], b["du_f"] = self._planes(B, c, h, w), self._planes(B, c, h, w)
        b["dvf"] = self._planes(B, FINAL_FILTERS, h, w)
        b["dprev_f"] = self._planes(B, c, h, w)
        self.buf = b
        self._build_pack_table()

    def _build_pack_table(self):
        """Descriptor table for orcai_pack_weights / orcai_h_pack_weights: kernel-layout copies of every trunk weight as views into
        one packed buffer refreshed once per step.  f32: depthwise taps (forward / reversed) and transposed pointwise / residual
        matrices.  f16 (half): depthwise octets (forward / reversed), A fragments of every pointwise / residual matrix and of its
        transpose, identity fragments and all-ones taps for the depthwise-only / pointwise-only passes."""
        P, m, k = self.P, self.model, self.k
        desc, self.packed_views, off = [], {}, 0
        align = 8 if self.half else 4  # keep every view 16-byte aligned

        def add(kind, key, src_name, C, aux, numel):
            nonlocal off
            desc.append([kind, P.offsets[src_name][0] if src_name else 0, off, C, aux])
            self.packed_views[(kind, key)] = (off, numel)
            off += (numel + align - 1) // align * align

        def frag(ck, cr):  # halves of an A-fragment array: contraction channels ck, row channels cr
            return ((ck + 31) // 32) * ((cr + 15) // 16) * 512

        names = []
        c = 16
        for i, f in enumerate(m.filters, start=1):
            names += [(f"b{i}/sep_a", c, f), (f"b{i}/sep_b", f, f)]
            if self.half:
                add(2, f"b{i}/res/kernel", f"b{i}/res/kernel", c, f, frag(c, f))
                add(3, f"b{i}/res/kernel", f"b{i}/res/kernel", c, f, frag(f, c))
            else:
                add(2, f"b{i}/res/kernel", f"b{i}/res/kernel", c, f, c * f)
            c = f
        names.append(("sep_f", c, FINAL_FILTERS))
        for name, cin, cout in names:
            G = self.G
            cpad = G * ((cin + G - 1) // G)
            add(0, name + "/depthwise", name + "/depthwise", cin, k * k, cpad * k * k)
            add(1, name + "/depthwise", name + "/depthwise", cin, k * k, cpad * k * k)
            if self.half:
                add(2, name + "/pointwise", name + "/pointwise", cin, cout, frag(cin, cout))
                add(3, name + "/pointwise", name + "/pointwise", cin, cout, frag(cout, cin))
            else:
                add(2, name + "/pointwise", name + "/pointwise", cin, cout, cin * cout)
        if self.half:
            for ch in sorted({16, FINAL_FILTERS, *m.filters}):
                add(4, f"eye{ch}", None, ch, 0, frag(ch, ch))
                add(5, f"ones{ch}", None, ch, 0, 8 * ((ch + 7) // 8))
        self.pack_desc = torch.tensor(desc, dtype=torch.int32, device=self.dev).contiguous()
        self.packed = torch.empty(off, dtype=self.adt, device=self.dev)

    def _packed(self, kind, name, shape):
        o, n = self.packed_views[(kind, name)]
        return self.packed[o : o + n].view(shape)

    # ------------------------------------------------------------- forward
    def forward(self, src: torch.Tensor, snippet_stride: int, B: int) -> torch.Tensor:
        """src: flat f32 cuda tensor, snippet i = [H][W] at element offset i*snippet_stride.  Returns featv [B][T][W_last*36]
        (pre-BN output of the final separable conv)."""
        self._alloc(B)
        lib, P, m, b, st = self.lib, self.P, self.model, self.buf, N.stream_ptr()
        self.stats = {}
        self.src, self.snippet_stride = src, snippet_stride
        N.check(self._fn("pack_weights")(P.w.data_ptr(), self.pack_desc.data_ptr(), int(self.pack_desc.shape[0]), self.packed.data_ptr(), st), "pack_weights")
        H, W = m.input_hw
        k = self.k
        shapes = m.stage_shapes()
        self.v0_stored = self.half or not self.conv0_two_pass
        if self.v0_stored:
            stats_first = bool(self.half and self.conv0_stats_from_input and self.conv0_march and k == 3)
            if not stats_first:
                N.check(self._fn("conv0_affine")(src.data_ptr(), snippet_stride, B, H, W, k, P.W("conv0/kernel").data_ptr(), self._ones(16).data_ptr(), P.W("conv0/bias").data_ptr(),
                                                 0, b["v0"].data_ptr(), st), "orcai_conv0_affine")
            if stats_first:
                # bn0's batch statistics from the 1-channel snippet (the f32 path's marching statistics pass: 4 bytes per pixel read) instead of a pass over
                # the stored 16-channel v0 (32 bytes per pixel).  They are the statistics of the conv BEFORE its rounding to f16: the mean moves by < 2^-12 of
                # a standard deviation, the variance by 2^-24 relative -- below what the f16 storage of v0 itself does to the normalised values
                mean0, var0 = P.B("bn0/mean"), P.B("bn0/var")
                self._fresh()
                N.check(lib.orcai_conv0_stats_march(src.data_ptr(), snippet_stride, B, H, W, P.W("conv0/kernel").data_ptr(), self._ones(16).data_ptr(), P.W("conv0/bias").data_ptr(),
                                                    self.scratch.data_ptr(), st), "conv0_stats_march")
                N.check(lib.orcai_bn_finish_sharded(self.scratch.data_ptr(), B, 16, H, W, mean0.data_ptr(), var0.data_ptr(), st), "bn_finish_sharded")
                self.stats["bn0"] = (mean0, var0)
                # v0 (kept for the backward pass) and y0 = relu(bn0(v0)) from one launch: the statistics already exist
                N.check(lib.orcai_h_conv0_affine_bn(src.data_ptr(), snippet_stride, B, H, W, k, P.W("conv0/kernel").data_ptr(), self._ones(16).data_ptr(), P.W("conv0/bias").data_ptr(),
                                                    mean0.data_ptr(), var0.data_ptr(), P.W("bn0/gamma").data_ptr(), P.W("bn0/beta").data_ptr(), BN_EPS, b["v0"].data_ptr(),
                                                    b["y0"].data_ptr(), st), "orcai_h_conv0_affine_bn")
            else:
                self._bn_fwd(b["v0"], "bn0", 16, H, W, 1, b["y0"])
        else:
            # two passes over the 1-channel input instead of three over the 16-channel v0: statistics only, then conv + bn0 + ReLU -> y0; v0 itself
            # is never written (the backward pass rebuilds it from the input taps: orcai_conv0_bn_bwd_x)
            w0, b0, ones = P.W("conv0/kernel").data_ptr(), P.W("conv0/bias").data_ptr(), self._ones(16).data_ptr()
            mean0, var0 = P.B("bn0/mean"), P.B("bn0/var")
            self._fresh()
            if self.conv0_march and k == 3:  # the marching form: no tiles, LDS or barriers (csrc/train_trunk.hip conv0_march_kernel)
                N.check(lib.orcai_conv0_stats_march(src.data_ptr(), snippet_stride, B, H, W, w0, ones, b0, self.scratch.data_ptr(), st), "conv0_stats_march")
            else:
                N.check(lib.orcai_conv0_stats(src.data_ptr(), snippet_stride, B, H, W, k, w0, ones, b0, self.scratch.data_ptr(), st), "conv0_stats")
            N.check(lib.orcai_bn_finish_sharded(self.scratch.data_ptr(), B, 16, H, W, mean0.data_ptr(), var0.data_ptr(), st), "bn_finish_sharded")
            self.stats["bn0"] = (mean0, var0)
            N.check(lib.orcai_conv0_affine_bn(src.data_ptr(), snippet_stride, B, H, W, k, w0, ones, b0, mean0.data_ptr(), var0.data_ptr(), P.W("bn0/gamma").data_ptr(),
                                              P.W("bn0/beta").data_ptr(), BN_EPS, 1, b["y0"].data_ptr(), st), "conv0_affine_bn")
        prev, c = b["y0"], 16
        res_in = prev
        self.dwl = {}
        self.block_in = {}
        self.on_load = {}  # block -> bn_a + ReLU applied on load by sep_b (y_a not materialised this step)
        for i, f in enumerate(m.filters, start=1):
            h, w, _ = shapes[i - 1]
            self.block_in[i] = (prev, res_in)  # (input of sep_a, input of the residual conv): the same tensor without block dropout
            # sep_a -> bn_a (+ ReLU) -> sep_b -> bn_b.  bn_b feeds only the pooling, which applies it on the fly to the maximum (monotone per channel):
            # y_b is never written.  bn_a + ReLU: applied where sep_b (and, in the backward pass, its depthwise weight gradient) load their
            # input -- y_a is not materialised either -- when sep_b runs on the LDS-tile kernels with the statistics epilogue (k = 3).
            na, nb = f"b{i}/sep_a", f"b{i}/sep_b"
            self.dwl[na], self.dwl[nb] = self._w_dw(na), self._w_dw(nb)
            va, ya, vb = b[f"va{i}"], b[f"ya{i}"], b[f"vb{i}"]
            fused = self.stats_in_epilogue and k == 3 and self._sep_stats(prev, c, h, w, 1, self.dwl[na], self._w_pw(na + "/pointwise"), P.W(na + "/bias"), f, va, b[f"u_a{i}"])
            if not fused:
                self._sep(prev, c, h, w, k, 1, self.dwl[na], self._w_pw(na + "/pointwise"), P.W(na + "/bias"), f, va, u_out=b[f"u_a{i}"])
            on_load = self.apply_on_load and self.stats_in_epilogue and k == 3
            self._bn_fwd(va, f"b{i}/bn_a", f, h, w, 1, None if on_load else ya, sums_in_shards=fused)
            if on_load:
                on_load = self._sep_stats_bn(va, f"b{i}/bn_a", f, h, w, self.dwl[nb], self._w_pw(nb + "/pointwise"), P.W(nb + "/bias"), f, vb, b[f"u_b{i}"])
                if not on_load:  # not a shape of the tile kernels: materialise y_a after all
                    self._bn_apply(va, f"b{i}/bn_a", f, h, w, 1, ya)
            self.on_load[i] = on_load
            fused = on_load
            if not on_load:
                fused = self.stats_in_epilogue and k == 3 and self._sep_stats(ya, f, h, w, 0, self.dwl[nb], self._w_pw(nb + "/pointwise"), P.W(nb + "/bias"), f, vb, b[f"u_b{i}"])
                if not fused:
                    self._sep(ya, f, h, w, k, 0, self.dwl[nb], self._w_pw(nb + "/pointwise"), P.W(nb + "/bias"), f, vb, u_out=b[f"u_b{i}"])
            self._bn_fwd(vb, f"b{i}/bn_b", f, h, w, 0, None, sums_in_shards=fused)
            # the residual branch reads the block input BEFORE the previous block's Dropout (architectures.py:88-97)
            bmean, bvar = self.stats[f"b{i}/bn_b"]
            N.check(self._fn("pool_res_add_bn")(b[f"vb{i}"].data_ptr(), res_in.data_ptr(), B, f, c, h, w, k, self._w_pw(f"b{i}/res/kernel").data_ptr(),
                                                P.W(f"b{i}/res/bias").data_ptr(), b[f"prev{i}"].data_ptr(), 0, bmean.data_ptr(), bvar.data_ptr(),
                                                P.W(f"b{i}/bn_b/gamma").data_ptr(), P.W(f"b{i}/bn_b/beta").data_ptr(), BN_EPS, st), "orcai_pool_res_add_bn")
            prev, c = b[f"prev{i}"], f
            res_in = prev
            if self.block_masks is not None:  # ResNet1DConv: Dropout after every block; the dropped tensor feeds the next separable conv only
                dropped = b[f"prevd{i}"]
                N.check(lib.orcai_mask_scale(prev.data_ptr(), self.block_masks[i - 1].data_ptr(), 1.0 / (1.0 - self.block_rate), prev.numel(), dropped.data_ptr(), st),
                        "mask_scale")
                prev = dropped
        h, w, _ = shapes[-1]
        self.dwl["sep_f"] = self._w_dw("sep_f")
        featv = torch.empty((B, h, w * FINAL_FILTERS), dtype=torch.float32, device=self.dev)
        self._sep(prev, c, h, w, k, 0, self.dwl["sep_f"], self._w_pw("sep_f/pointwise"), P.W("sep_f/bias"), FINAL_FILTERS, featv, layout=1, u_out=b["u_f"])
        self.final_in = prev
        return featv

    def update_moving_stats(self) -> None:
        S = self.P.stats
        for bn, (mean, var) in self.stats.items():
            S[bn + "/mean"].mul_(BN_MOMENTUM).add_(mean, alpha=1 - BN_MOMENTUM)
            S[bn + "/var"].mul_(BN_MOMENTUM).add_(var, alpha=1 - BN_MOMENTUM)

    # ------------------------------------------------------------- backward
    def _bn_sep_backward(self, dy, v, bn, relu, name, x, relu_in, Cin, Cout, H, W, u, du, dr, sums_ready=0, epi=None, x_bn=None):
        """BatchNorm backward (in place on dy -> dv) fused with the first step of the separable conv's backward (du = Wpw dv),
        then the rest of _sep_backward.  One pass over (dy, v) replaces BN apply + a pointwise pass that re-reads dv.
        epi: epilogue extra of the input-gradient pass (see _dgrad); returns whether it ran."""
        lib, P, st, k = self.lib, self.P, N.stream_ptr(), self.k
        mean, var = self.stats[bn]
        wt = self._w_pwT(name + "/pointwise", Cin, Cout)  # pointwise^T [Cout][Cin]
        if not sums_ready:  # the launcher reduces the BatchNorm backward sums itself (otherwise it READS them from self.scratch)
            self._fresh()
        if self.fused_pw_wgrad:
            # one pass: dv formed per pixel, du = Wpw dv, AND the pointwise weight gradient u (x) dv -- dv is never written or re-read
            rc = (lib.orcai_h_bn_bwd_pointwise_wgrad if self.half else lib.orcai_bn_bwd_pointwise_wgrad)(dy.data_ptr(), v.data_ptr(), u.data_ptr(), self.B, Cout, H, W, k, mean.data_ptr(), var.data_ptr(), P.W(bn + "/gamma").data_ptr(),
                                                  P.W(bn + "/beta").data_ptr(), BN_EPS, relu, self.scratch.data_ptr(), sums_ready, P.G(bn + "/beta").data_ptr(),
                                                  P.G(bn + "/gamma").data_ptr(), wt.data_ptr(), Cin, du.data_ptr(), P.G(name + "/pointwise").data_ptr(), self.partials.data_ptr(),
                                                  self.partials.numel(), st)
            if rc != N.E_UNSUPPORTED:
                N.check(rc, "bn_bwd_pointwise_wgrad")
                return self._sep_backward(name, x, relu_in, Cin, Cout, H, W, None, u, du, dr, have_du=True, have_pw_wgrad=True, epi=epi, x_bn=x_bn)
        N.check(self._fn("bn_bwd_pointwise")(dy.data_ptr(), v.data_ptr(), self.B, Cout, H, W, k, mean.data_ptr(), var.data_ptr(), P.W(bn + "/gamma").data_ptr(),
                                           P.W(bn + "/beta").data_ptr(), BN_EPS, relu, self.scratch.data_ptr(), sums_ready, P.G(bn + "/beta").data_ptr(),
                                           P.G(bn + "/gamma").data_ptr(), wt.data_ptr(), Cin, dy.data_ptr(), du.data_ptr(), st), "bn_bwd_pointwise")
        return self._sep_backward(name, x, relu_in, Cin, Cout, H, W, dy, u, du, dr, have_du=True, epi=epi, x_bn=x_bn)

    def _dgrad(self, name, du, Cin, H, W, dr, epi=None) -> bool:
        """dr = depthwise conv of du with the flipped taps (identity pointwise factor): the gradient w.r.t. the separable conv's input.
        epi ("bsums", v_ref, bn, relu): dr is the gradient dy of BatchNorm `bn` (pre-normalisation tensor v_ref): its backward sums are reduced
        in the kernel's epilogue and left in self.scratch; ("relu", x_ref): dr is masked by x_ref > 0 (the ReLU in front of the conv).
        Returns True when the epilogue ran (k = 3, f32, a shape of the LDS-tile kernels); otherwise the plain pass ran and the caller
        does the separate launches."""
        k, P = self.k, self.P
        dw, eye, zeros = self._w_dw(name, reverse=True), self._w_eye(Cin), self._zeros(64)
        if epi is not None and self.dgrad_epilogues and k == 3 and not self.half:
            if epi[0] == "bsums":
                _, ref, bn, relu = epi
                mean, var = self.stats[bn]
                self._fresh()
                rc = self.lib.orcai_sepconv_planes_epi(du.data_ptr(), self.B, Cin, H, W, dw.data_ptr(), eye.data_ptr(), self._ones(64).data_ptr(), zeros.data_ptr(), Cin, dr.data_ptr(), 2,
                                                       ref.data_ptr(), mean.data_ptr(), var.data_ptr(), P.W(bn + "/gamma").data_ptr(), P.W(bn + "/beta").data_ptr(), BN_EPS, relu,
                                                       self.scratch.data_ptr(), N.stream_ptr())
            else:
                rc = self.lib.orcai_sepconv_planes_epi(du.data_ptr(), self.B, Cin, H, W, dw.data_ptr(), eye.data_ptr(), self._ones(64).data_ptr(), zeros.data_ptr(), Cin, dr.data_ptr(), 3,
                                                       epi[1].data_ptr(), None, None, None, None, 0.0, 0, None, N.stream_ptr())
            if rc != N.E_UNSUPPORTED:
                N.check(rc, "sepconv_planes_epi")
                return True
        self._sep(du, Cin, H, W, k, 0, dw, eye, zeros, Cin, dr)
        return False

    def _conv0_dgrad_ok(self, x) -> bool:
        """Block 1's first conv may rebuild its input y0 from the snippet (orcai_dw_bwd_fused_conv0): f32, k = 3, the two-pass entry conv (v0 not stored)."""
        return bool(self.conv0_in_dgrad and self.fused_dw_bwd and not self.v0_stored and not self.half and self.k == 3 and x.data_ptr() == self.buf["y0"].data_ptr())

    def _conv0_dgrad_half_ok(self, x) -> bool:
        """f16 path: block 1's first conv takes bn0's sums and the residual gradient into its marching pass (x = the stored v0)."""
        return bool(self.half and self.conv0_in_dgrad and self.fused_dw_bwd and self.v0_stored and self.k == 3 and self.block_masks is None
                    and x.data_ptr() == self.buf["y0"].data_ptr())

    def _dw_bwd_fused(self, name, x, relu_in, Cin, H, W, du, dr, epi, x_bn):
        """orcai_dw_bwd_fused for one separable conv: dr, the depthwise weight gradient and the epilogue extra `epi` of _dgrad in one pass over
        (du, x).  Returns whether the epilogue extra ran (True / False), or None when the launch is not this kernel's (the caller runs the
        separate passes): the extras read the conv's own input, so ("bsums", ref, ...) needs ref to be the pre-normalisation tensor x."""
        P = self.P
        if name == "b1/sep_a" and epi is None and x_bn is None and self._resq is not None and self._conv0_dgrad_half_ok(x):
            # f16 path: the entry conv's stored v0 is the pass's x (y0 formed from it on load), bn0's backward sums over the total gradient in its
            # epilogue, the residual branch's even-pixel gradient added inside (orcai_h_dw_bwd_fused_res + orcai_h_conv0_bn_bwd_ready)
            mean0, var0 = self.stats["bn0"]
            self._fresh()
            rc = self.lib.orcai_h_dw_bwd_fused_res(self.buf["v0"].data_ptr(), du.data_ptr(), self.B, Cin, H, W, self._w_dw(name, reverse=True).data_ptr(), dr.data_ptr(),
                                                   P.G(name + "/depthwise").data_ptr(), mean0.data_ptr(), var0.data_ptr(), P.W("bn0/gamma").data_ptr(), P.W("bn0/beta").data_ptr(),
                                                   BN_EPS, 1, self.scratch.data_ptr(), self._resq.data_ptr(), N.stream_ptr())
            if rc != N.E_UNSUPPORTED:
                N.check(rc, "orcai_h_dw_bwd_fused_res")
                self.bn0_sums_ready = True
                return False
        if name == "b1/sep_a" and epi is None and x_bn is None and self._conv0_dgrad_ok(x):
            # block 1's first conv: y0 rebuilt from the snippet's taps instead of read, bn0's backward sums left in self.scratch for orcai_conv0_bn_bwd_x_ready
            mean0, var0 = self.stats["bn0"]
            self._fresh()
            rc = self.lib.orcai_dw_bwd_fused_conv0(self.src.data_ptr(), self.snippet_stride, du.data_ptr(), self.B, H, W, P.W("conv0/kernel").data_ptr(), P.W("conv0/bias").data_ptr(),
                                                   self._w_dw(name, reverse=True).data_ptr(), dr.data_ptr(), P.G(name + "/depthwise").data_ptr(), mean0.data_ptr(), var0.data_ptr(),
                                                   P.W("bn0/gamma").data_ptr(), P.W("bn0/beta").data_ptr(), BN_EPS, self.scratch.data_ptr(),
                                                   None if self._resq is None else self._resq.data_ptr(), N.stream_ptr())
            if rc != N.E_UNSUPPORTED:
                N.check(rc, "orcai_dw_bwd_fused_conv0")
                self.bn0_sums_ready = True
                return False
        mode, bn, bn_relu = 0, x_bn, 0
        if epi is not None and self.dgrad_epilogues:
            if epi[0] == "bsums":
                if self.half and x_bn is None and relu_in == 0 and epi[3] == 1:
                    # f16 path: the forward materialised x = y_a = f16(relu(BN(v_a))); the kernel forms exactly that value from v_a on load,
                    # so the pass reads the pre-normalisation tensor (which the sums need anyway) instead of y_a
                    x, x_bn = epi[1], epi[2]
                    bn = x_bn
                if x_bn is None or epi[2] != x_bn or epi[1].data_ptr() != x.data_ptr():
                    return None
                mode, bn_relu = 2, epi[3]
            elif epi[0] == "relu":
                if x_bn is not None or not relu_in or epi[1].data_ptr() != x.data_ptr():
                    return None
                mode = 3
        elif epi is not None:
            return None
        bnp = [None] * 4
        if bn is not None:
            mean, var = self.stats[bn]
            bnp = [mean.data_ptr(), var.data_ptr(), P.W(bn + "/gamma").data_ptr(), P.W(bn + "/beta").data_ptr()]
        if mode == 2:
            self._fresh()
        rc = self._fn("dw_bwd_fused")(x.data_ptr(), du.data_ptr(), self.B, Cin, H, W, relu_in, self._w_dw(name, reverse=True).data_ptr(), dr.data_ptr(),
                                         P.G(name + "/depthwise").data_ptr(), mode, *bnp, BN_EPS, bn_relu, self.scratch.data_ptr(), N.stream_ptr())
        if rc == N.E_UNSUPPORTED:
            return None
        N.check(rc, "orcai_dw_bwd_fused")
        return mode != 0

    def _sep_backward(self, name, x, relu_in, Cin, Cout, H, W, dv, u, du, dr, have_du=False, have_pw_wgrad=False, epi=None, x_bn=None):
        """Backward of one separable conv (+bias): fills dW(depthwise), dW(pointwise), dbias; writes dr = gradient w.r.t. the
        (ReLU'd) input into `dr` (planes of Cin channels)."""
        lib, P, st, k = self.lib, self.P, N.stream_ptr(), self.k
        # d loss / d bias = sum_pixels dv, and dv is the gradient through a BatchNormalization of batch statistics: that sum is
        # identically zero (the bias shifts the batch mean, which BN subtracts), so the gradient buffer keeps its zero.
        # u = dw(relu?(x)) was stored by the forward pass.
        if not have_du:  # du = Wpw dv   (pointwise conv with the transposed weights)
            wt = self._w_pwT(name + "/pointwise", Cin, Cout)
            self._sep(dv, Cout, H, W, 1, 0, self._w_ones_dw(Cout), wt, self._zeros(64), Cin, du)
        epi_ran = False
        if self.fused_dw_bwd and k == 3:
            # one marching pass over (du, x): input gradient, its epilogue extra and the depthwise weight gradient (csrc/train_trunk.hip: dw_bwd_march_kernel)
            fused = self._dw_bwd_fused(name, x, relu_in, Cin, H, W, du, dr, epi, x_bn)
            if fused is not None:
                if not have_pw_wgrad:
                    N.check(self._fn("outer_reduce")(u.data_ptr(), Cin, dv.data_ptr(), Cout, self.B, H, W, k, 0, 0, 0, P.G(name + "/pointwise").data_ptr(), self.partials.data_ptr(),
                                                     self.partials.numel(), st), "outer_reduce")
                return fused
        if self.dgrad_first:
            # the input gradient (the only kernel of this layer the next layer waits for) first; the two weight-gradient passes are
            # read-only, and a read-only pass runs faster behind a kernel that wrote ANOTHER tensor (dr) than directly behind the writer
            # of its own input (dv, du) -- DESIGN.md 4.4
            epi_ran = self._dgrad(name, du, Cin, H, W, dr, epi)
        if not have_pw_wgrad:
            N.check(self._fn("outer_reduce")(u.data_ptr(), Cin, dv.data_ptr(), Cout, self.B, H, W, k, 0, 0, 0, P.G(name + "/pointwise").data_ptr(), self.partials.data_ptr(),
                                             self.partials.numel(), st), "outer_reduce")
        # depthwise weight gradient, accumulated straight into the (zeroed) flat gradient buffer in the Keras layout
        if x_bn is not None and self.half:
            # f16 path, a shape the marching pass refused: its plain depthwise weight gradient reads the MATERIALISED y_a -- form it now (the value the
            # forward conv formed on load) in the block's y_a planes
            ya = self.buf["ya" + name[1:name.index("/")]]
            self._bn_apply(x, x_bn, Cin, H, W, 1, ya)
            N.check(self._fn("dw_wgrad")(ya.data_ptr(), du.data_ptr(), self.B, Cin, H, W, k, k, 0, P.G(name + "/depthwise").data_ptr(), st), "dw_wgrad")
        elif x_bn is not None:  # x is the pre-normalisation tensor of BatchNorm x_bn (+ ReLU): normalised on load, as the forward conv did
            mean, var = self.stats[x_bn]
            N.check(lib.orcai_dw_wgrad_bn(x.data_ptr(), du.data_ptr(), self.B, Cin, H, W, mean.data_ptr(), var.data_ptr(), P.W(x_bn + "/gamma").data_ptr(),
                                          P.W(x_bn + "/beta").data_ptr(), BN_EPS, P.G(name + "/depthwise").data_ptr(), st), "dw_wgrad_bn")
        else:
            N.check(self._fn("dw_wgrad")(x.data_ptr(), du.data_ptr(), self.B, Cin, H, W, k, k, relu_in, P.G(name + "/depthwise").data_ptr(), st), "dw_wgrad")
        if not self.dgrad_first:
            epi_ran = self._dgrad(name, du, Cin, H, W, dr, epi)
        return epi_ran

    def backward(self, dfeatv: torch.Tensor) -> None:
        """dfeatv: gradient w.r.t. the pre-BN output of the final separable conv, Keras Reshape layout [B][T][W*36]."""
        lib, P, m, b, st, k, B = self.lib, self.P, self.model, self.buf, N.stream_ptr(), self.k, self.B
        shapes = m.stage_shapes()
        L = len(m.filters)
        h, w, c = shapes[-1]
        self.bn0_sums_ready = False
        N.check(self._fn("feat_to_planes")(dfeatv.data_ptr(), B, FINAL_FILTERS, h, w, k, b["dvf"].data_ptr(), st), "feat_to_planes")
        dprev = b["dprev_f"]
        self._sep_backward("sep_f", self.final_in, 0, c, FINAL_FILTERS, h, w, b["dvf"], b["u_f"], b["du_f"], dprev)
        for i in range(L, 0, -1):
            f = m.filters[i - 1]
            h, w, cprev = shapes[i - 1]
            ho, wo, _ = shapes[i]
            x_in, prev = self.block_in[i]  # input of sep_a (dropped for ResNet1DConv) / input of the residual conv
            if self.block_masks is not None and i == L:  # sep_f read Dropout(prev_L): its input gradient goes back through that Dropout
                N.check(lib.orcai_mask_scale(dprev.data_ptr(), self.block_masks[i - 1].data_ptr(), 1.0 / (1.0 - self.block_rate), dprev.numel(), dprev.data_ptr(), st),
                        "mask_scale")
            dout = dprev  # gradient w.r.t. prev_i (planes of f channels, ho x wo)
            bias_in_pool = self.bias_in_pool
            def residual_wgrad():  # residual 1x1 stride-2 conv: weight / bias gradients (read-only passes over prev and dout)
                N.check(self._fn("outer_reduce")(prev.data_ptr(), cprev, dout.data_ptr(), f, B, ho, wo, k, 1, h, w, P.G(f"b{i}/res/kernel").data_ptr(),
                                                 self.partials.data_ptr(), self.partials.numel(), st), "outer_reduce")
                if not bias_in_pool:
                    self._fresh_res()
                    N.check(self._fn("planes_sum")(dout.data_ptr(), B, f, ho, wo, k, self.res_scratch.data_ptr(), P.G(f"b{i}/res/bias").data_ptr(), 0, st), "planes_sum")

            if not self.dgrad_first:
                residual_wgrad()
            # max-pool branch
            dyb = b[f"dyb{i}"]
            bmean, bvar = self.stats[f"b{i}/bn_b"]  # the pooling backward also accumulates bn_b's backward reductions (sum dy, sum dy*xhat)
            self._fresh()
            if bias_in_pool:  # the residual conv's bias gradient (sum of dout) reduced where the pooling backward reads dout anyway
                self._fresh_res()
                N.check((lib.orcai_h_pool_bwd_bn_bias if self.half else lib.orcai_pool_bwd_bn_bias)(dout.data_ptr(), b[f"vb{i}"].data_ptr(), B, f, h, w, k, dyb.data_ptr(), P.W(f"b{i}/bn_b/gamma").data_ptr(), bmean.data_ptr(),
                                                   bvar.data_ptr(), BN_EPS, self.scratch.data_ptr(), self.res_scratch.data_ptr(), P.G(f"b{i}/res/bias").data_ptr(), st),
                        "pool_bwd_bn_bias")
            else:
                N.check(self._fn("pool_bwd_bn")(dout.data_ptr(), b[f"vb{i}"].data_ptr(), B, f, h, w, k, dyb.data_ptr(), P.W(f"b{i}/bn_b/gamma").data_ptr(), bmean.data_ptr(),
                                                bvar.data_ptr(), BN_EPS, self.scratch.data_ptr(), st), "pool_bwd_bn")
            if self.dgrad_first:  # behind the kernel that wrote dyb, not behind the one that wrote dout (see _sep_backward)
                residual_wgrad()
            dya = b[f"dya{i}"]
            # the input-gradient pass of sep_b writes dy_a = the gradient of bn_a's output: bn_a's backward sums are reduced in its epilogue
            on_load = self.on_load.get(i, False)  # sep_b read bn_a + ReLU of v_a on load: its depthwise weight gradient does the same
            sums_a = self._bn_sep_backward(dyb, b[f"vb{i}"], f"b{i}/bn_b", 0, f"b{i}/sep_b", b[f"va{i}"] if on_load else b[f"ya{i}"], 0, f, f, h, w, b[f"u_b{i}"], b[f"du_b{i}"],
                                           dya, sums_ready=1, epi=("bsums", b[f"va{i}"], f"b{i}/bn_a", 1), x_bn=f"b{i}/bn_a" if on_load else None)
            dr = b[f"dr{i}"]
            # through the ReLU in front of sep_a (folded into the input-gradient pass of sep_a where its kernel has the epilogue), then add the
            # residual branch (scatter-add to the even pixels).  For block 1, x_in = relu(bn0(v0)): its ReLU mask is the one the bn0 backward
            # applies anyway (mask * mask = mask)
            wrt = self._w_pwT(f"b{i}/res/kernel", cprev, f)  # residual weights transposed [f][cprev]
            self._resq = None
            if i == 1 and (self._conv0_dgrad_ok(x_in) or self._conv0_dgrad_half_ok(x_in)):
                # the residual branch's gradient w.r.t. y0 lives on the even pixels only: one plain pointwise pass at the pooled resolution, added inside
                # the marching pass below (where bn0's sums are taken over the TOTAL gradient) instead of scatter-added to dr afterwards
                self._resq = b["rq1"]
                self._sep(dout, f, ho, wo, 1, 0, self._w_ones_dw(f), wrt, self._zeros(64), cprev, self._resq)
            relu_done = self._bn_sep_backward(dya, b[f"va{i}"], f"b{i}/bn_a", 1, f"b{i}/sep_a", x_in, 1, cprev, f, h, w, b[f"u_a{i}"], b[f"du_a{i}"], dr,
                                              sums_ready=1 if sums_a else 0, epi=("relu", x_in) if i > 1 else None)
            if i > 1 and not relu_done:
                N.check(self._fn("planes_relu_bwd")(dr.data_ptr(), x_in.data_ptr(), dr.numel(), dr.data_ptr(), st), "planes_relu_bwd")
            if self.block_masks is not None and i > 1:  # x_in = Dropout(prev_{i-1}): back to the un-dropped tensor before the residual gradient joins
                N.check(lib.orcai_mask_scale(dr.data_ptr(), self.block_masks[i - 2].data_ptr(), 1.0 / (1.0 - self.block_rate), dr.numel(), dr.data_ptr(), st), "mask_scale")
            if not (i == 1 and self._resq is not None and self.bn0_sums_ready):  # (block 1 with the residual gradient already inside dr)
                self._sep(dout, f, ho, wo, 1, 0, self._w_ones_dw(f), wrt, self._zeros(64), cprev, dr, layout=3, H2=h, W2=w)
            self._resq = None
            dprev = dr
        H, W = m.input_hw
        mean0, var0 = self.stats["bn0"]  # bn0 (+ReLU) backward fused into the entry conv's weight gradient: dv0 is never written
        if not self.bn0_sums_ready:
            self._fresh()
        if self.v0_stored:
            c0bwd = self.lib.orcai_h_conv0_bn_bwd_ready if (self.half and self.bn0_sums_ready) else self._fn("conv0_bn_bwd")
            N.check(c0bwd(self.src.data_ptr(), self.snippet_stride, dprev.data_ptr(), b["v0"].data_ptr(), B, H, W, k, mean0.data_ptr(), var0.data_ptr(),
                                             P.W("bn0/gamma").data_ptr(), P.W("bn0/beta").data_ptr(), BN_EPS, self.scratch.data_ptr(), P.G("bn0/beta").data_ptr(),
                                             P.G("bn0/gamma").data_ptr(), P.G("conv0/kernel").data_ptr(), st), "conv0_bn_bwd")
        else:
            N.check((lib.orcai_conv0_bn_bwd_x_ready if self.bn0_sums_ready else lib.orcai_conv0_bn_bwd_x)(self.src.data_ptr(), self.snippet_stride, dprev.data_ptr(), B, H, W, k, P.W("conv0/kernel").data_ptr(), P.W("conv0/bias").data_ptr(),
                                             mean0.data_ptr(), var0.data_ptr(), P.W("bn0/gamma").data_ptr(), P.W("bn0/beta").data_ptr(), BN_EPS, self.scratch.data_ptr(),
                                             P.G("bn0/beta").data_ptr(), P.G("bn0/gamma").data_ptr(), P.G("conv0/kernel").data_ptr(), self.partials.data_ptr(), self.partials.numel(),
                                             st), "conv0_bn_bwd_x")
        # conv0/bias feeds bn0: zero gradient (see _sep_backward)


class Trainer:
    """One optimisation step = forward (training mode) + masked BCE + L2 + backward + (all-reduce) + Adam."""

    def __init__(self, model: ResNetLSTM, learning_rate: float = 1e-4, seed: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("orcai_amd training needs a ROCm GPU: there is no CPU / autograd fallback")
        self.model = model
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.P = FlatParams(model, self.dev)
        # model.precision "f16": activations / activation gradients in f16 octet planes, contractions on f16 MFMA, f32 master weights,
        # f32 gradients and Adam, static loss scale (orcai_amd/half.py; BASELINE configs[4])
        self.half = getattr(model, "precision", "f32") == "f16"
        if self.half:
            from orcai_amd.half import LOSS_SCALE

            self.grad_scale = LOSS_SCALE
        else:
            self.grad_scale = 1.0
        self.trunk = TrunkTrainer(model, self.P, half=self.half)
        self.conv1d = getattr(model, "architecture", "") == "ResNet1DConv"
        self.head = Conv1DHeadTrainer(model, self.P) if self.conv1d else HeadTrainer(model, self.P, half=self.half, grad_scale=self.grad_scale)
        self.skipped = torch.zeros(1, dtype=torch.int64, device=self.dev)  # f16 path: steps voided by a non-finite gradient / batch statistic
        self.ok_dev = torch.ones(1, dtype=torch.int32, device=self.dev)  # this step's verdict (orcai_step_ok), read by the guarded update kernels
        # Step state the kernels read from DEVICE memory (so that a captured hipGraph of the step stays valid from replay to replay):
        # the number of applied steps (dropout seeds, Adam's bias correction) and the learning rate (callbacks change it between steps)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.lr = float(learning_rate)
        self.step_count = 0
        self.seed = int(seed)
        self.rank = 0
        self._graph = None
        self.broadcast_parameters()

    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float) -> None:
        self._lr = float(value)
        self.lr_dev.fill_(self._lr)

    def broadcast_parameters(self, src: int = 0) -> None:
        """Data parallel replicas must START from the same weights (the reference's MirroredStrategy creates the variables once and
        mirrors them, hpsearch.py:186-205): inside an initialised process group, rank `src`'s parameters, Adam moments and BatchNorm
        moving statistics overwrite every other rank's.  Only gradients are exchanged afterwards, so replicas stay identical.  Each
        rank keeps its own dropout stream (the rank is mixed into the mask seed)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        self.rank = dist.get_rank()
        tensors = [self.P.w, self.P.m, self.P.v] + [self.P.stats[k] for k in sorted(self.P.stats)]
        if dist.get_backend() == "nccl":
            for t in tensors:
                dist.broadcast(t, src=src)
        else:  # gloo (CPU tests): stage through the host
            for t in tensors:
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)
        self.model._dev = None  # folded inference copies are stale

    def _masks(self, n, T):
        rate = self.model.dropout_rate
        if rate <= 0.0:
            return None
        lib, st = N.lib(), N.stream_ptr()
        out = {}
        if self.conv1d:  # one mask per residual block (plane layout, drawn over the whole buffer: the pads are zero anyway) + one after BN_f
            shapes = self.model.stage_shapes()
            R = self.model.kernel_size // 2
            todo = [(f"block{i}", (n, (shapes[i][2] + 3) // 4, shapes[i][0] + 2 * R, self.model.padded_width(shapes[i][1]), 4)) for i in range(1, len(self.model.filters) + 1)]
            todo.append(("final", (n, T, shapes[-1][1] * FINAL_FILTERS)))
        else:
            todo = [("drop1", (n, T, 2 * self.model.lstm_units)), ("drop2", (n, T, 2 * self.model.lstm_units)), ("drop3", (n, T, DENSE_UNITS))]
        for idx, (j, shape) in enumerate(todo):
            mk = torch.empty(shape, dtype=torch.float32, device=self.dev)
            seed = (self.seed * 1000003 + self.rank * 0x9E3779B1 + idx + 1) & 0xFFFFFFFFFFFFFFFF  # + the step counter, on the device
            N.check(lib.orcai_dropout_mask_dev(mk.data_ptr(), mk.numel(), self.counter.data_ptr(), seed, 1.0 - rate, st), "dropout_mask")
            out[j] = mk
        return out

    def forward_backward(self, src: torch.Tensor, snippet_stride: int, B: int, labels: torch.Tensor, masks: dict | None = "auto",
                         loss_weight: torch.Tensor | None = None) -> dict:
        """Gradients of (masked BCE + L2) into the flat gradient buffer.  Returns device accumulators {bce sum, count, correct, l2}."""
        self.P.g.zero_()
        if isinstance(masks, str):
            masks = self._masks(B, self.model.out_steps)
        self.trunk.block_masks = [masks[f"block{i}"] for i in range(1, len(self.model.filters) + 1)] if (self.conv1d and masks is not None) else None
        self.trunk.begin_step()  # one clear for every reduction scratch of the step
        try:
            featv = self.trunk.forward(src, snippet_stride, B)
            probs = self.head.forward(featv, masks, self.model.dropout_rate)
            out = self.head.loss_and_backward(labels, loss_weight)
            self.trunk.backward(out["dfeatv"])
        finally:
            self.trunk.end_step()
        return {"acc": out["acc"], "probs": probs}

    def apply(self, world_size: int = 1) -> None:
        """(all-reduce) + Adam + BN moving statistics."""
        if world_size > 1:
            import torch.distributed as dist

            if self.half:
                # one verdict for all replicas: a rank whose batch statistics are not finite poisons the bucket it contributes, so the all-reduced
                # gradient is non-finite everywhere and every rank's orcai_step_ok below voids the step (weights, moments, counters stay identical)
                N.check(N.lib().orcai_poison_if_nonfinite(self.P.batch_flat.data_ptr(), self.P.batch_flat.numel(), self.P.g.data_ptr(), N.stream_ptr()), "poison_if_nonfinite")
            if dist.get_backend() == "nccl":
                dist.all_reduce(self.P.g, op=dist.ReduceOp.SUM)  # one flat 4 MB bucket over RCCL / xGMI
            else:  # gloo (tests): stage through the host
                g = self.P.g.cpu()
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
                self.P.g.copy_(g)
        self.step_count += 1
        P, st, lib = self.P, N.stream_ptr(), N.lib()
        if self.half:
            # f16 under a static loss scale: a non-finite gradient or batch statistic voids the WHOLE step on the device -- no Adam update
            # (not even the momentum term), no moving-statistics update, no step-counter advance -- and is counted (Keras' LossScaleOptimizer
            # skips such a step the same way).  The host mirror of the counter is re-read from the device where it matters (state_dict).
            N.check(lib.orcai_step_ok(P.g.data_ptr(), P.n_trainable, P.batch_flat.data_ptr(), P.batch_flat.numel(), self.ok_dev.data_ptr(), self.skipped.data_ptr(), st), "step_ok")
            N.check(lib.orcai_adam_step_guarded(P.w.data_ptr(), P.g.data_ptr(), P.m.data_ptr(), P.v.data_ptr(), P.n_trainable, self.lr_dev.data_ptr(), 0.9, 0.999, 1e-7,
                                                self.counter.data_ptr(), 1.0 / (world_size * self.grad_scale), self.ok_dev.data_ptr(), st), "adam_step_guarded")
            N.check(lib.orcai_ema_update_guarded(P.stats_flat.data_ptr(), P.batch_flat.data_ptr(), P.stats_flat.numel(), BN_MOMENTUM, self.ok_dev.data_ptr(), st), "ema_update_guarded")
            N.check(lib.orcai_counter_advance_guarded(self.counter.data_ptr(), self.ok_dev.data_ptr(), st), "counter_advance_guarded")
            return
        N.check(lib.orcai_adam_step_dev(P.w.data_ptr(), P.g.data_ptr(), P.m.data_ptr(), P.v.data_ptr(), P.n_trainable, self.lr_dev.data_ptr(), 0.9, 0.999, 1e-7,
                                        self.counter.data_ptr(), 1.0 / (world_size * self.grad_scale), st), "adam_step_dev")
        self.P.ema_all(BN_MOMENTUM)  # every BatchNorm's moving statistics in one launch
        N.check(lib.orcai_counter_advance(self.counter.data_ptr(), st), "counter_advance")

    def sync_model(self) -> None:
        """Copy the flat device parameters (and BN moving statistics) back into the model object (for predict / save)."""
        self.P.to_model(self.model)

    def state_dict(self) -> dict:
        if self.half:  # voided steps do not advance the device counter: the device value is the truth
            self.step_count = int(self.counter.item())
        return {"w": self.P.w.clone(), "m": self.P.m.clone(), "v": self.P.v.clone(), "stats": {k: t.clone() for k, t in self.P.stats.items()},
                "step": self.step_count, "lr": self.lr}

    def load_state_dict(self, s: dict) -> None:
        self.P.w.copy_(s["w"]); self.P.m.copy_(s["m"]); self.P.v.copy_(s["v"])
        for k, t in s["stats"].items():
            self.P.stats[k].copy_(t)
        self.step_count, self.lr = s["step"], s["lr"]
        self.counter.fill_(int(self.step_count))

    def train_step(self, src, snippet_stride, B, labels, world_size: int = 1, loss_weight=None) -> dict:
        out = self.forward_backward(src, snippet_stride, B, labels, loss_weight=loss_weight)
        self.apply(world_size)
        return out

    def release_graph(self) -> None:
        """Drop the captured step and its private memory pool (a hyper-parameter search builds one trainer per trial: hpsearch.py:110-257).  Not
        inside another capture: destroying a graph is a runtime call."""
        if self._graph is not None:
            torch.cuda.synchronize()
            self._graph = None

    def train_step_graphed(self, src: torch.Tensor, snippet_stride: int, B: int, labels: torch.Tensor) -> dict:
        """The whole single-GPU step -- dropout masks, forward, loss, backward, Adam, moving statistics, step counter -- as ONE
        hipGraph: captured on the first call for this batch geometry (after two eager warm-up steps that size every workspace; the
        trainer's state is put back afterwards, so a graphed run is step-for-step the eager run), replayed afterwards on the batch
        copied into the graph's input buffers.  The C ABI's promise (nothing allocates, frees or synchronises; include/orcai_hip.h) is
        what makes the capture legal; what changes per step lives in device memory (self.counter, self.lr_dev).  Every accumulator the
        library clears inside the step is cleared by a KERNEL node: a captured hipMemsetAsync node re-reads its fill pattern at replay
        from kernel-argument memory the graph does not own (HIP 7.0.51831 as bundled with torch 2.10; tools/debug_graph_memset_torch.py).
        Returns the graph's output tensors: valid until the next replay."""
        key = (int(snippet_stride), int(B), tuple(labels.shape))
        if self._graph is None or self._graph["key"] != key:
            x_in = src[: (B - 1) * snippet_stride + self.model.input_hw[0] * self.model.input_hw[1]].clone()
            y_in = labels.clone()
            saved = self.state_dict()
            saved_batch, saved_skipped = self.P.batch_flat.clone(), self.skipped.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):  # warm-up: allocate workspaces, instantiate kernels (hipFuncSetAttribute and friends happen here)
                    self.train_step(x_in, snippet_stride, B, y_in)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread-local capture mode: the dataset's producer thread keeps pinning and uploading the next batches meanwhile (a global-mode
                # capture is invalidated by another thread's hipHostMalloc and the process aborts)
                # ... and no cyclic garbage collection inside the capture: a collection that happens to run there finalises whatever became unreachable
                # earlier in the process (a previous trial's trainer with its own captured graph, pinned batches of the dataset thread) with
                # runtime calls that are illegal while this thread captures -- the process aborted in hpsearch's second trial
                import gc

                gc.collect()
                gc_was_enabled = gc.isenabled()
                gc.disable()
                try:
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        out = self.train_step(x_in, snippet_stride, B, y_in)
                finally:
                    if gc_was_enabled:
                        gc.enable()
            torch.cuda.current_stream().wait_stream(side)
            # the two warm-up steps moved weights, Adam moments, moving statistics and the step counter (the capture pass ran nothing):
            # back to the state the caller handed in
            self.load_state_dict(saved)
            self.P.batch_flat.copy_(saved_batch)
            self.skipped.copy_(saved_skipped)
            self._graph = {"key": key, "graph": g, "x": x_in, "y": y_in, "out": out}
        G = self._graph
        G["x"].copy_(src[: G["x"].numel()])
        G["y"].copy_(labels)
        G["graph"].replay()
        self.step_count += 1
        return G["out"]
