"""Model definitions.  Mirrors reference ``src/orcAI/architectures.py`` for the hot path:
``res_net_LSTM_arch`` (:120-241), ``build_model`` (:316-359), the architecture registry (:307-312).

The returned object is duck-type compatible with what the reference's callers use of a keras.Model
(``predict``, ``count_params``, ``trainable_weights``, ``non_trainable_weights``, ``input_shape``,
``output_shape``, ``save``/``load_weights``); all arithmetic runs in the HIP kernels of
``csrc/model_fwd.hip`` behind the C ABI -- there is no CPU path.

Weights are kept on the host as a dict of numpy arrays with Keras variable layouts (so a converter from
``.keras`` only has to rename), and are folded / permuted into the kernels' layouts when uploaded.
"""

from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import torch

from orcai_amd import _native as N
from orcai_amd.auxiliary import MASK_VALUE, Messenger  # noqa: F401  (re-exported like the reference)

BN_EPS = 1e-3  # keras BatchNormalization default epsilon
ENTRY_FILTERS = 16  # architectures.py:164
FINAL_FILTERS = 36  # architectures.py:199
DENSE_UNITS = 128  # architectures.py:232


def _bn_names(prefix):
    return [f"{prefix}/gamma", f"{prefix}/beta", f"{prefix}/mean", f"{prefix}/var"]


def lstm_column_permutation(units: int) -> np.ndarray:
    """perm[p] = Keras gate column held by kernel column p (see orcai_lstm_recurrent in orcai_hip.h)."""
    p = np.arange(4 * units)
    w, nt, j = p // 32, (p // 16) % 2, p % 16
    return (2 * nt + (j >> 3)) * units + 8 * w + (j & 7)


def depthwise_kernel_layout(dwk: np.ndarray) -> np.ndarray:
    """Keras depthwise kernel (k,k,c,1) -> [ceil(c/4)][k*k][4] (channel innermost inside a quad; zero taps for the channels that
    pad the last quad): the layout the kernels read with scalar loads."""
    k, c = dwk.shape[0], dwk.shape[2]
    cq = (c + 3) // 4
    out = np.zeros((cq * 4, k * k), dtype=np.float32)
    out[:c] = dwk[:, :, :, 0].transpose(2, 0, 1).reshape(c, k * k)
    return np.ascontiguousarray(out.reshape(cq, 4, k * k).transpose(0, 2, 1))


class ResNetLSTM:
    """CNN with residual connections + 2 bidirectional LSTM layers (architectures.py:120-241)."""

    architecture = "ResNetLSTM"

    def __init__(self, input_shape, num_labels, filters, kernel_size, dropout_rate=0.0, lstm_units=128,
                 conv_initializer="he_normal", lstm_initializer="glorot_uniform", seed=None, precision="f32", **unused):
        self.input_hw = (int(input_shape[0]), int(input_shape[1]))
        if int(input_shape[2]) != 1:
            raise ValueError("ResNetLSTM expects a single input channel")
        self.num_labels = int(num_labels)
        self.filters = [int(f) for f in filters]
        self.kernel_size = int(kernel_size)
        self.dropout_rate = float(dropout_rate)
        self.lstm_units = int(lstm_units)
        if self.kernel_size not in (3, 5, 7):
            raise NotImplementedError("HIP kernels implement kernel_size 3, 5 and 7")
        if self.lstm_units not in (64, 128):
            raise NotImplementedError("HIP LSTM kernel implements lstm_units 64 and 128")
        if max(self.filters + [FINAL_FILTERS]) > 64:
            raise NotImplementedError("HIP separable-conv kernel implements up to 64 filters")
        if self.num_labels > 8:
            raise NotImplementedError("HIP head kernel implements up to 8 labels")
        # "f32" (the reference's arithmetic, default) or "f16": f16 octet planes + v_mfma_f32_16x16x32_f16 contractions with f32
        # master weights (orcai_amd/half.py; BASELINE configs[4]).  An extra key of orcai_parameter["model"], swallowed by the
        # reference's **unused like any other (architectures.py:129).
        if precision not in ("f32", "f16"):
            raise ValueError(f"precision must be 'f32' or 'f16', got {precision!r}")
        self.precision = precision
        self.graph_step = bool(unused.get("graph_step", True))  # FitLoop: replay the training step as one hipGraph (single GPU)
        self._half_engine = None
        self.weights: dict[str, np.ndarray] = {}
        self._init_weights(np.random.default_rng(seed))
        self._dev = None  # folded device copies
        self._ws = {}
        import os

        # inference, k = 3: the entry convolution is computed inside the first separable convolution (orcai_conv0_sepconv); the
        # 16-channel entry activation never reaches HBM, block 1's residual branch reads a quarter-size subsample of it
        self.fuse_entry = os.environ.get("ORCAI_FUSE_ENTRY", "1") != "0"
        self.kernel_events = None  # bench hook: {label: [(start_event, end_event), ...]} when not None
        self.kernel_event_labels = None  # bench hook: restrict the event brackets to these labels (an event pair costs ~15 us of queue time)
        # Inference trunk in two phases: blocks < tail_from_block in chunks of `chunk` snippets (their planes are large), the
        # later blocks (planes of a few thousand pixels: a chunk of 128 snippets is < 2 waves per SIMD) over up to tail_chunk snippets.
        self.tail_from_block = int(os.environ.get("ORCAI_TAIL_FROM_BLOCK", "3"))
        self.tail_chunk = int(os.environ.get("ORCAI_TAIL_CHUNK", "2048"))

    # ------------------------------------------------------------------ structure
    @property
    def input_shape(self):
        return (None, self.input_hw[0], self.input_hw[1], 1)

    @property
    def output_shape(self):
        return (None, self.out_steps, self.num_labels)

    @property
    def time_reduction(self) -> int:
        return 2 ** len(self.filters)

    def stage_shapes(self):
        """[(H, W, C)] after the entry conv and after each block (SAME pooling: ceil(n/2))."""
        h, w = self.input_hw
        shapes = [(h, w, ENTRY_FILTERS)]
        for f in self.filters:
            h, w = -(-h // 2), -(-w // 2)
            shapes.append((h, w, f))
        return shapes

    @property
    def out_steps(self) -> int:
        return self.stage_shapes()[-1][0]

    conv_kind = "he"  # he_normal: orcai-V1's conv_initializer (models/orcai-V1/orcai_parameter.json:13)

    def _trunk_spec(self):
        k, ck = self.kernel_size, self.conv_kind
        spec = [("conv0/kernel", (k, k, 1, ENTRY_FILTERS), ck, True), ("conv0/bias", (ENTRY_FILTERS,), "zeros", True)]
        spec += self._bn_spec("bn0", ENTRY_FILTERS)
        c = ENTRY_FILTERS
        for b, s in enumerate(self.filters, start=1):
            spec += [(f"b{b}/sep_a/depthwise", (k, k, c, 1), ck, True), (f"b{b}/sep_a/pointwise", (1, 1, c, s), ck, True), (f"b{b}/sep_a/bias", (s,), "zeros", True)]
            spec += self._bn_spec(f"b{b}/bn_a", s)
            spec += [(f"b{b}/sep_b/depthwise", (k, k, s, 1), ck, True), (f"b{b}/sep_b/pointwise", (1, 1, s, s), ck, True), (f"b{b}/sep_b/bias", (s,), "zeros", True)]
            spec += self._bn_spec(f"b{b}/bn_b", s)
            spec += [(f"b{b}/res/kernel", (1, 1, c, s), ck, True), (f"b{b}/res/bias", (s,), "zeros", True)]
            c = s
        spec += [("sep_f/depthwise", (k, k, c, 1), ck, True), ("sep_f/pointwise", (1, 1, c, FINAL_FILTERS), ck, True), ("sep_f/bias", (FINAL_FILTERS,), "zeros", True)]
        spec += self._bn_spec("bn_f", FINAL_FILTERS)
        return spec

    def _head_spec(self):
        u = self.lstm_units
        spec = []
        feat = self.stage_shapes()[-1][1] * FINAL_FILTERS
        for layer, fin in ((1, feat), (2, 2 * u)):
            for d in ("fwd", "bwd"):
                spec += [(f"lstm{layer}/{d}/kernel", (fin, 4 * u), "glorot", True), (f"lstm{layer}/{d}/recurrent", (u, 4 * u), "orthogonal", True),
                         (f"lstm{layer}/{d}/bias", (4 * u,), "lstm_bias", True)]
        spec += [("dense1/kernel", (2 * u, DENSE_UNITS), "he", True), ("dense1/bias", (DENSE_UNITS,), "zeros", True)]
        spec += self._bn_spec("bn_d", DENSE_UNITS)
        spec += [("dense2/kernel", (DENSE_UNITS, self.num_labels), "glorot", True), ("dense2/bias", (self.num_labels,), "zeros", True)]
        return spec

    def variable_spec(self):
        """Ordered (name, shape, initializer-kind, trainable)."""
        return self._trunk_spec() + self._head_spec()

    @staticmethod
    def _bn_spec(name, c):
        return [(f"{name}/gamma", (c,), "ones", True), (f"{name}/beta", (c,), "zeros", True), (f"{name}/mean", (c,), "zeros", False), (f"{name}/var", (c,), "ones", False)]

    def _init_weights(self, rng):
        """Keras initialisers: he_normal (truncated normal, stddev sqrt(2/fan_in)/.8796), glorot_uniform,
        orthogonal recurrent kernels, unit_forget_bias (architectures.py:126-127, 210-229)."""
        for name, shape, kind, _ in self.variable_spec():
            if kind == "he":
                if len(shape) == 4:
                    fan_in = shape[0] * shape[1] * shape[2]
                else:
                    fan_in = shape[0]
                std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
                w = np.clip(rng.standard_normal(shape), -2.0, 2.0) * std
            elif kind == "glorot":  # keras fans: receptive field x in / out channels for conv kernels
                rf = int(np.prod(shape[:-2]))
                limit = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
                w = rng.uniform(-limit, limit, shape)
            elif kind == "orthogonal":
                a = rng.standard_normal((shape[1], shape[0]))
                q, r = np.linalg.qr(a)
                q = q * np.sign(np.diag(r))
                w = q.T[: shape[0], : shape[1]]
            elif kind == "lstm_bias":
                u = shape[0] // 4
                w = np.zeros(shape)
                w[u : 2 * u] = 1.0
            elif kind == "ones":
                w = np.ones(shape)
            else:
                w = np.zeros(shape)
            self.weights[name] = np.ascontiguousarray(w, dtype=np.float32)

    # ------------------------------------------------------------------ keras-shaped accessors
    @property
    def trainable_weights(self):
        return [self.weights[n] for n, _, _, t in self.variable_spec() if t]

    @property
    def non_trainable_weights(self):
        return [self.weights[n] for n, _, _, t in self.variable_spec() if not t]

    def count_params(self) -> int:
        return int(sum(int(np.prod(s)) for _, s, _, _ in self.variable_spec()))

    def set_weights_dict(self, weights: dict) -> None:
        for name, shape, _, _ in self.variable_spec():
            if name not in weights:
                raise ValueError(f"missing weight {name}")
            w = np.asarray(weights[name], dtype=np.float32)
            if tuple(w.shape) != tuple(shape):
                raise ValueError(f"weight {name}: shape {w.shape} != {shape}")
            self.weights[name] = np.ascontiguousarray(w)
        self._dev = None
        if self._half_engine is not None:
            self._half_engine._dev = None

    def save_weights(self, path) -> None:
        np.savez(path, **self.weights)

    def load_weights(self, path) -> None:
        with np.load(path) as z:
            self.set_weights_dict({k: z[k] for k in z.files})

    def save(self, path, include_optimizer: bool = True) -> None:
        """Stores ``<path minus .keras>.weights.npz`` (Keras' zip/HDF5 container cannot be written here)."""
        path = Path(path)
        if path.suffix == ".keras":
            path = path.with_suffix("")
        from orcai_amd.io import WEIGHTS_SUFFIX

        self.save_weights(str(path) + WEIGHTS_SUFFIX)

    # ------------------------------------------------------------------ device side
    def _upload(self, a) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    def _fold_bn(self, bn, bias=None):
        w = self.weights
        g, b, m, v = (w[n].astype(np.float64) for n in _bn_names(bn))
        scale = g / np.sqrt(v + BN_EPS)
        shift = b - m * scale
        if bias is not None:
            shift = shift + bias.astype(np.float64) * scale
        return self._upload(scale), self._upload(shift)

    def prepare(self) -> dict:
        """Fold BN, transpose / permute weights into kernel layouts, upload.  Cached until weights change."""
        if self._dev is not None:
            return self._dev
        if not torch.cuda.is_available():
            raise RuntimeError("orcai_amd model needs a ROCm GPU: there is no CPU fallback")
        N.lib()
        w = self.weights
        k = self.kernel_size
        d = {}
        d["conv0/w"] = self._upload(w["conv0/kernel"].reshape(k * k, ENTRY_FILTERS))
        d["conv0/scale"], d["conv0/shift"] = self._fold_bn("bn0", w["conv0/bias"])

        def sep(name, bn):
            dwk = w[name + "/depthwise"]  # (k,k,c,1)
            c = dwk.shape[2]
            d[name + "/dw"] = self._upload(depthwise_kernel_layout(dwk))
            d[name + "/pw"] = self._upload(w[name + "/pointwise"][0, 0])
            d[name + "/scale"], d[name + "/shift"] = self._fold_bn(bn, w[name + "/bias"])

        for b in range(1, len(self.filters) + 1):
            sep(f"b{b}/sep_a", f"b{b}/bn_a")
            sep(f"b{b}/sep_b", f"b{b}/bn_b")
            d[f"b{b}/res/w"] = self._upload(w[f"b{b}/res/kernel"][0, 0])
            d[f"b{b}/res/b"] = self._upload(w[f"b{b}/res/bias"])
        sep("sep_f", "bn_f")
        self._prepare_head(d)
        self._dev = d
        return d

    def padded_width(self, w: int) -> int:
        return (w + self.kernel_size // 2 + 3) & ~3

    def _buffers(self, B: int, first: int = 1, last: int | None = None, need_input: bool = True) -> dict:
        """Zero-padded activation planes of residual blocks first..last for a trunk chunk of B snippets (see "Padded plane
        layout" in csrc/model_fwd.hip).  Allocated ZEROED once for the largest chunk seen; the kernels never write the pads.
        `prev{first-1}` (the input of block `first`) is included when need_input."""
        last = len(self.filters) if last is None else last
        key = (first, last)
        have = self._ws.get(key)
        if have is not None and have[0] >= B:  # planes are snippet-major: a smaller chunk uses the head of a larger workspace
            return have[1]
        shapes = self.stage_shapes()
        dev = torch.device("cuda", torch.cuda.current_device())
        R = self.kernel_size // 2

        def planes(c, h, w):  # [B][channel quad][HP][WP][4]
            return torch.zeros((B, (c + 3) // 4, h + 2 * R, self.padded_width(w), 4), dtype=torch.float32, device=dev)

        ws = {}
        if need_input:
            ws[f"prev{first - 1}"] = planes(shapes[first - 1][2], shapes[first - 1][0], shapes[first - 1][1])
            if first == 1:  # compact (2i, 2j) subsample of the entry activation for the fused entry path
                h0, w0, c0 = shapes[0]
                ws["prev0s"] = torch.zeros((B, (c0 + 3) // 4, (h0 + 1) // 2, (w0 + 1) // 2, 4), dtype=torch.float32, device=dev)
        for b in range(first, last + 1):
            f = self.filters[b - 1]
            h, wd, _ = shapes[b - 1]
            ws[f"a{b}"] = planes(f, h, wd)
            wx = (wd + 1) // 2  # x-pooled output of the block's second separable conv: [B][CQ][H][roundup4(ceil(W/2))][4]
            ws[f"b{b}"] = torch.zeros((B, (f + 3) // 4, h, (wx + 3) & ~3, 4), dtype=torch.float32, device=dev)
            ws[f"prev{b}"] = planes(f, shapes[b][0], shapes[b][1])
        self._ws[key] = (B, ws)
        return ws

    def _prepare_head(self, d: dict) -> None:
        w = self.weights
        perm = lstm_column_permutation(self.lstm_units)
        for layer in (1, 2):
            ks = [w[f"lstm{layer}/{dd}/kernel"][:, perm] for dd in ("fwd", "bwd")]
            bs = [w[f"lstm{layer}/{dd}/bias"][perm] for dd in ("fwd", "bwd")]
            us = [w[f"lstm{layer}/{dd}/recurrent"][:, perm] for dd in ("fwd", "bwd")]
            d[f"lstm{layer}/W"] = self._upload(np.concatenate(ks, axis=1))  # [Fin][2*4u]
            d[f"lstm{layer}/b"] = self._upload(np.concatenate(bs))
            d[f"lstm{layer}/U"] = self._upload(np.stack(us))  # [2][u][4u]
        d["dense1/W"] = self._upload(w["dense1/kernel"])
        d["dense1/b"] = self._upload(w["dense1/bias"])
        d["dense1/scale"], d["dense1/shift"] = self._fold_bn("bn_d")
        d["dense2/W"] = self._upload(w["dense2/kernel"])
        d["dense2/b"] = self._upload(w["dense2/bias"])

    def _launch(self, label: str, what: str, fn, *args, may_refuse: bool = False) -> bool:
        """Call one C-ABI launcher; optionally bracket it with HIP events on the launch stream (bench.py).  may_refuse: ORCAI_E_UNSUPPORTED (the
        launcher refused the shape before touching anything) returns False and the caller launches its alternative; otherwise True."""
        ev = self.kernel_events
        if ev is None or (self.kernel_event_labels is not None and label not in self.kernel_event_labels):
            rc = fn(*args)
            if may_refuse and rc == N.E_UNSUPPORTED:
                return False
            N.check(rc, what)
            return True
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        if may_refuse and rc == N.E_UNSUPPORTED:
            return False
        N.check(rc, what)
        e1.record()
        ev.setdefault(label, []).append((e0, e1))
        return True

    def trunk_device(self, src: torch.Tensor, snippet_stride: int, B: int, feat: torch.Tensor, keep: dict | None = None, first: int = 0,
                     last: int | None = None, ws: dict | None = None) -> dict:
        """Convolutional trunk for one chunk of B snippets, stages first..last: stage 0 = entry conv, b = residual block b,
        len(filters)+1 = final separable conv (writes the LSTM input features feat[B][steps][W_last*36]).  Returns the workspace
        (its `prev{last}` planes are the input of stage last+1)."""
        lib = N.lib()
        d = self.prepare()
        nb = len(self.filters)
        last = nb + 1 if last is None else last
        if ws is None:
            ws = self._buffers(B, max(first, 1), min(last, nb))
        st = N.stream_ptr()
        H, W = self.input_hw
        k = self.kernel_size
        shapes = self.stage_shapes()
        fuse_entry = first == 0 and last >= 1 and k == 3 and self.fuse_entry and keep is None and self.filters[0] <= 64
        if first == 0 and not fuse_entry:
            self._launch("conv0", "orcai_conv0_bn_relu", lib.orcai_conv0_bn_relu, src.data_ptr(), snippet_stride, B, H, W, k, N.ptr(d["conv0/w"]),
                         N.ptr(d["conv0/scale"]), N.ptr(d["conv0/shift"]), N.ptr(ws["prev0"]), st)
        for b in range(max(first, 1), min(last, nb) + 1):
            f = self.filters[b - 1]
            c = shapes[b - 1][2]
            h, wd, _ = shapes[b - 1]
            prev, a, bb, nxt = ws[f"prev{b - 1}"], ws[f"a{b}"], ws[f"b{b}"], ws[f"prev{b}"]
            pa, pb = f"b{b}/sep_a", f"b{b}/sep_b"
            entry = fuse_entry and b == 1
            if entry:
                prev = ws["prev0s"]
                self._launch("conv0+b1/sep_a", "orcai_conv0_sepconv", lib.orcai_conv0_sepconv, src.data_ptr(), snippet_stride, B, H, W, N.ptr(d["conv0/w"]),
                             N.ptr(d["conv0/scale"]), N.ptr(d["conv0/shift"]), N.ptr(d[pa + "/dw"]), N.ptr(d[pa + "/pw"]), N.ptr(d[pa + "/scale"]),
                             N.ptr(d[pa + "/shift"]), f, 1, N.ptr(a), N.ptr(prev), st)
            else:
                self._launch(pa, "orcai_sepconv_bn", lib.orcai_sepconv_bn, N.ptr(prev), B, c, h, wd, k, 1, N.ptr(d[pa + "/dw"]), N.ptr(d[pa + "/pw"]),
                             N.ptr(d[pa + "/scale"]), N.ptr(d[pa + "/shift"]), f, 1, 0, N.ptr(a), st)
            # second separable conv + the block's tail (pool, residual conv, add) in one launch where the marching kernel has the shape (orcai-V1
            # block 1): the x-pooled tensor never reaches HBM; otherwise the two launches -- the same bits either way
            if keep is None and self._launch(f"b{b}/sep_b+pool_res", "orcai_sepconv_pool_res", lib.orcai_sepconv_pool_res, N.ptr(a), N.ptr(prev), B, f, f, c, h, wd, k, 0,
                                             N.ptr(d[pb + "/dw"]), N.ptr(d[pb + "/pw"]), N.ptr(d[pb + "/scale"]), N.ptr(d[pb + "/shift"]), 0, N.ptr(d[f"b{b}/res/w"]),
                                             N.ptr(d[f"b{b}/res/b"]), N.ptr(nxt), 1 if entry else 0, st, may_refuse=True):
                continue
            self._launch(pb, "orcai_sepconv_bn", lib.orcai_sepconv_bn, N.ptr(a), B, f, h, wd, k, 0, N.ptr(d[pb + "/dw"]), N.ptr(d[pb + "/pw"]),
                         N.ptr(d[pb + "/scale"]), N.ptr(d[pb + "/shift"]), f, 0, 2, N.ptr(bb), st)
            self._launch(f"b{b}/pool_res", "orcai_pool_res_add", lib.orcai_pool_res_add, N.ptr(bb), N.ptr(prev), B, f, c, h, wd, k, N.ptr(d[f"b{b}/res/w"]),
                         N.ptr(d[f"b{b}/res/b"]), N.ptr(nxt), 3 if entry else 1, st)
        if last == nb + 1:
            h, wd, c = shapes[-1]
            self._launch("sep_f", "orcai_sepconv_bn", lib.orcai_sepconv_bn, N.ptr(ws[f"prev{nb}"]), B, c, h, wd, k, 0, N.ptr(d["sep_f/dw"]), N.ptr(d["sep_f/pw"]),
                         N.ptr(d["sep_f/scale"]), N.ptr(d["sep_f/shift"]), FINAL_FILTERS, 1, 1, feat.data_ptr(), st)
        if keep is not None:  # test hook: planes back to [B][C][H][W]
            R = k // 2
            chans = {"prev0": ENTRY_FILTERS}
            widths = {"prev0": shapes[0][1]}
            for i, f in enumerate(self.filters, start=1):
                chans.update({f"a{i}": f, f"b{i}": f, f"prev{i}": f})
                widths.update({f"a{i}": shapes[i - 1][1], f"b{i}": shapes[i - 1][1], f"prev{i}": shapes[i][1]})
            for name, t in ws.items():
                if name == "prev0s":  # only written by the fused entry path, which the keep hook does not use
                    continue
                if name.startswith("b"):  # x-pooled (unpadded rows): [B][CQ][H][WPx][4] -> [B][C][H][ceil(W/2)]
                    t = t[:B]  # the workspace may be larger than this chunk
                    Bq, CQ, hh, WPx, _ = t.shape
                    full = t.permute(0, 1, 4, 2, 3).reshape(Bq, CQ * 4, hh, WPx)
                    keep[name] = full[:, : chans[name], :, : (widths[name] + 1) // 2].clone()
                    continue
                t = t[:B]
                Bq, CQ, HPp, WPp, _ = t.shape
                hh = HPp - 2 * R
                full = t.permute(0, 1, 4, 2, 3).reshape(Bq, CQ * 4, HPp, WPp)
                keep[name] = full[:, : chans[name], R : R + hh, : widths[name]].clone()
                pads = full.clone()
                pads[:, : chans[name], R : R + hh, : widths[name]] = 0
                keep[name + "/pads"] = pads
        return ws

    def head_device(self, feat: torch.Tensor, out: torch.Tensor, keep: dict | None = None) -> None:
        """Both BiLSTM layers, Dense(128)+BN and Dense(labels)+sigmoid for ALL n snippets at once
        (the recurrence kernel wants >= 128 snippet tiles in flight to fill the chip)."""
        lib = N.lib()
        d = self.prepare()
        st = N.stream_ptr()
        n, h, fin = int(feat.shape[0]), int(feat.shape[1]), int(feat.shape[2])
        u = self.lstm_units
        dev = feat.device
        xz = torch.empty((n, h, 2, 4 * u), dtype=torch.float32, device=dev)
        h1 = torch.empty((n, h, 2 * u), dtype=torch.float32, device=dev)
        h2 = torch.empty((n, h, 2 * u), dtype=torch.float32, device=dev)
        M = n * h
        x = feat
        for layer, hout in ((1, h1), (2, h2)):
            self._launch(f"lstm{layer}/gemm", "orcai_gemm_bias_act", lib.orcai_gemm_bias_act, N.ptr(x), N.ptr(d[f"lstm{layer}/W"]), N.ptr(d[f"lstm{layer}/b"]),
                         None, None, N.ptr(xz), M, 8 * u, fin, 0, st)
            self._launch(f"lstm{layer}/rec", "orcai_lstm_recurrent", lib.orcai_lstm_recurrent, N.ptr(xz), N.ptr(d[f"lstm{layer}/U"]), n, h, u, N.ptr(hout), st)
            x, fin = hout, 2 * u
        d1 = xz.view(-1)[: M * DENSE_UNITS].view(n, h, DENSE_UNITS)  # xz is free again: reuse it for the Dense-128 output
        self._launch("dense1", "orcai_gemm_bias_act", lib.orcai_gemm_bias_act, N.ptr(h2), N.ptr(d["dense1/W"]), N.ptr(d["dense1/b"]), N.ptr(d["dense1/scale"]),
                     N.ptr(d["dense1/shift"]), N.ptr(d1), M, DENSE_UNITS, 2 * u, 1, st)
        self._launch("dense2", "orcai_dense_sigmoid", lib.orcai_dense_sigmoid, N.ptr(d1), N.ptr(d["dense2/W"]), N.ptr(d["dense2/b"]), M, DENSE_UNITS,
                     self.num_labels, out.data_ptr(), st)
        if keep is not None:
            keep.update({"feat": feat.clone(), "h1": h1.clone(), "h2": h2.clone()})

    def forward_device(self, src: torch.Tensor, snippet_stride: int, n: int, out: torch.Tensor, chunk: int = 128, keep: dict | None = None) -> None:
        """n snippets starting at ``src`` (f32 cuda), snippet i at element offset i*snippet_stride, each [H][W] row-major
        (unpadded).  Writes probabilities into out[n][steps][labels].  The trunk runs in chunks of `chunk` snippets
        (bounds activation memory); the recurrent head runs once over all n."""
        if self.precision == "f16":
            return self.half_engine().forward_device(src, snippet_stride, n, out, chunk=chunk, keep=keep)
        steps, wd, _ = self.stage_shapes()[-1]
        feat = torch.empty((n, steps, wd * FINAL_FILTERS), dtype=torch.float32, device=src.device)
        nb = len(self.filters)
        split = self.tail_from_block  # blocks >= split (small planes) run over `tail_chunk` snippets per launch to fill the chip
        if keep is not None or split > nb or n <= chunk:
            for s in range(0, n, chunk):
                B = min(chunk, n - s)
                self.trunk_device(src[s * snippet_stride :], snippet_stride, B, feat[s:], keep=keep if s == 0 else None)
        else:
            big = min(n, self.tail_chunk)
            tail = self._buffers(big, split, nb, need_input=True)  # its prev{split-1} planes receive the head stages' output
            carry = tail[f"prev{split - 1}"]
            for t0 in range(0, n, big):
                nt = min(big, n - t0)
                for s in range(t0, t0 + nt, chunk):
                    B = min(chunk, t0 + nt - s)
                    head = dict(self._buffers(B, 1, split - 1))
                    head[f"prev{split - 1}"] = carry[s - t0 :]  # the last head stage writes straight into the tail's input planes
                    self.trunk_device(src[s * snippet_stride :], snippet_stride, B, None, first=0, last=split - 1, ws=head)
                self.trunk_device(None, snippet_stride, nt, feat[t0:], first=split, last=nb + 1, ws=tail)
        self.head_device(feat, out, keep=keep)

    def half_engine(self):
        if self._half_engine is None:
            from orcai_amd.half import HalfEngine

            self._half_engine = HalfEngine(self)
        return self._half_engine

    def predict_spectrogram(self, spectrogram: torch.Tensor, chunk: int = 128, shard: bool = False) -> torch.Tensor:
        """All 50 %-overlapping snippets of a device spectrogram [T][W] -> f32 cuda [n][steps][labels].
        Snippet i = rows [i*H/2, i*H/2 + H) (predict.py:244-261), read in place (no snippet copy).
        shard=True inside an initialised process group: every rank holds the spectrogram, runs a contiguous block of the
        snippets and the blocks are all-gathered; the result is identical on every rank and to shard=False."""
        H, W = self.input_hw
        assert spectrogram.is_cuda and spectrogram.dtype == torch.float32 and spectrogram.is_contiguous()
        assert spectrogram.shape[1] == W
        shift = H // 2
        n = (spectrogram.shape[0] - H) // shift + 1
        if shard and n > 0:
            import torch.distributed as dist

            from orcai_amd import parallel

            if dist.is_initialized() and dist.get_world_size() > 1:
                i0, i1 = parallel.contiguous_range(n, dist.get_rank(), dist.get_world_size())
                local = torch.empty((i1 - i0, self.out_steps, self.num_labels), dtype=torch.float32, device=spectrogram.device)
                if i1 > i0:
                    self.forward_device(spectrogram.view(-1)[i0 * shift * W :], shift * W, i1 - i0, local, chunk=chunk)
                return parallel.all_gather_rows(local, n)
        out = torch.empty((max(n, 0), self.out_steps, self.num_labels), dtype=torch.float32, device=spectrogram.device)
        if n > 0:
            self.forward_device(spectrogram.view(-1), shift * W, n, out, chunk=chunk)
        return out

    # ------------------------------------------------------------------ keras-shaped training API (train.py:155-219)
    def compile(self, optimizer=None, loss=None, metrics=None, learning_rate: float | None = None, seed: int = 0) -> None:
        """``optimizer`` may be a number (learning rate) or an object with ``learning_rate``; the loss / metric are always
        MaskedBinaryCrossentropy / MaskedBinaryAccuracy (the only ones the reference compiles with)."""
        from orcai_amd.fit import FitLoop
        from orcai_amd.training import Trainer

        lr = learning_rate if learning_rate is not None else (optimizer if isinstance(optimizer, (int, float)) else getattr(optimizer, "learning_rate", 1e-4))
        self._loop = FitLoop(self, Trainer(self, learning_rate=float(lr), seed=seed))

    def _need_loop(self):
        if getattr(self, "_loop", None) is None:
            self.compile()
        return self._loop

    def fit(self, train_dataset, validation_data=None, epochs: int = 1, callbacks=(), class_weight=None, verbose: int = 0):
        return self._need_loop().fit(train_dataset, validation_data, epochs, callbacks, class_weight, verbose)

    def evaluate(self, dataset, return_dict: bool = True, verbose: int = 0):
        logs = self._need_loop().evaluate(dataset)
        return logs if return_dict else [logs["loss"], logs["MBA"]]

    def predict(self, snippets, batch_size: int = 64, verbose: int = 0, **unused) -> np.ndarray:
        """keras.Model.predict for materialised snippets: ndarray [n, H, W, 1] -> ndarray [n, steps, labels] (float32)."""
        x = np.asarray(snippets, dtype=np.float32)
        H, W = self.input_hw
        if x.ndim != 4 or x.shape[1:] != (H, W, 1):
            raise ValueError(f"expected input of shape (n, {H}, {W}, 1), got {x.shape}")
        n = x.shape[0]
        out = np.empty((n, self.out_steps, self.num_labels), dtype=np.float32)
        for s in range(0, n, batch_size):
            B = min(batch_size, n - s)
            xd = torch.from_numpy(np.ascontiguousarray(x[s : s + B, :, :, 0])).cuda()
            od = torch.empty((B, self.out_steps, self.num_labels), dtype=torch.float32, device=xd.device)
            self.forward_device(xd.view(-1), H * W, B, od, chunk=B)
            out[s : s + B] = od.cpu().numpy()
        return out


def res_net_LSTM_arch(input_shape, num_labels, filters, kernel_size, dropout_rate=0.0, lstm_units=128, conv_initializer="he_normal",
                      lstm_initializer="glorot_uniform", **unused) -> ResNetLSTM:
    """architectures.py:120-241.  ``precision`` ("f32" | "f16") may ride along in **unused (an extra key of orcai_parameter["model"])."""
    return ResNetLSTM(input_shape, num_labels, filters, kernel_size, dropout_rate, lstm_units, conv_initializer, lstm_initializer,
                      precision=unused.get("precision", "f32"), seed=unused.get("seed"), graph_step=unused.get("graph_step", True))


class ResNet1DConv(ResNetLSTM):
    """CNN with residual connections + frequency mean + one Conv1D over time (architectures.py:18-117).  The convolutional trunk
    is the ResNetLSTM one (the per-block Dropout layers are the identity at inference); the head is ReduceFrequencyMean
    (:10-15) and Conv1D(num_labels, kernel_size = 36, "same", sigmoid) (:107-115).  Training adds the Dropout after every block and
    after BN_f (orcai_amd.training.Conv1DHeadTrainer, TrunkTrainer.block_masks)."""

    architecture = "ResNet1DConv"
    conv_kind = "glorot"  # conv_initializer default "glorot_uniform" (architectures.py:24)

    def __init__(self, input_shape, num_labels, filters, kernel_size, dropout_rate=0.0, conv_initializer="glorot_uniform", seed=None, **unused):
        if unused.get("precision", "f32") != "f32":
            raise NotImplementedError("the f16 path implements ResNetLSTM only")
        super().__init__(input_shape, num_labels, filters, kernel_size, dropout_rate, lstm_units=128, conv_initializer=conv_initializer, seed=seed)

    def _head_spec(self):
        k1 = FINAL_FILTERS  # k_size = x.shape[2] after the frequency mean = the channel count (architectures.py:108)
        return [("conv1d/kernel", (k1, FINAL_FILTERS, self.num_labels), self.conv_kind, True), ("conv1d/bias", (self.num_labels,), "zeros", True)]

    def _prepare_head(self, d: dict) -> None:
        d["conv1d/W"] = self._upload(self.weights["conv1d/kernel"])
        d["conv1d/b"] = self._upload(self.weights["conv1d/bias"])

    def head_device(self, feat: torch.Tensor, out: torch.Tensor, keep: dict | None = None) -> None:
        lib = N.lib()
        d = self.prepare()
        st = N.stream_ptr()
        n, h, fin = int(feat.shape[0]), int(feat.shape[1]), int(feat.shape[2])
        wd = fin // FINAL_FILTERS
        fm = torch.empty((n, h, FINAL_FILTERS), dtype=torch.float32, device=feat.device)
        self._launch("freq_mean", "orcai_freq_mean", lib.orcai_freq_mean, N.ptr(feat), n * h, wd, FINAL_FILTERS, N.ptr(fm), st)
        self._launch("conv1d", "orcai_conv1d_sigmoid", lib.orcai_conv1d_sigmoid, N.ptr(fm), N.ptr(d["conv1d/W"]), N.ptr(d["conv1d/b"]), n, h, FINAL_FILTERS,
                     FINAL_FILTERS, self.num_labels, out.data_ptr(), st)
        if keep is not None:
            keep.update({"feat": feat.clone(), "freq_mean": fm.clone()})


def res_net_1Dconv_arch(input_shape, num_labels, filters, kernel_size, dropout_rate=0.0, conv_initializer="glorot_uniform", **unused) -> ResNet1DConv:
    """architectures.py:18-117."""
    return ResNet1DConv(input_shape, num_labels, filters, kernel_size, dropout_rate, conv_initializer, **unused)


ORCAI_ARCHITECTURES_FN = {"ResNet1DConv": res_net_1Dconv_arch, "ResNetLSTM": res_net_LSTM_arch}
ORCAI_ARCHITECTURES = list(ORCAI_ARCHITECTURES_FN.keys())


def build_model(input_shape, orcai_parameter: dict, msgr: Messenger = Messenger(verbosity=0)):
    """architectures.py:316-359."""
    num_labels = len(orcai_parameter["calls"])
    if orcai_parameter["architecture"] in ORCAI_ARCHITECTURES:
        model = ORCAI_ARCHITECTURES_FN[orcai_parameter["architecture"]](input_shape, num_labels, **orcai_parameter["model"])
    else:
        raise ValueError(f"Unknown model architecture: {orcai_parameter['architecture']}")
    n_filters = len(orcai_parameter["model"]["filters"])
    output_shape = (input_shape[0] // 2**n_filters, num_labels)
    msgr.part("Building model architecture")
    msgr.info(f"model name:          {orcai_parameter['name']}")
    msgr.info(f"model architecture:  {orcai_parameter['architecture']}")
    msgr.info(f"model input shape:   {model.input_shape}")
    msgr.info(f"model output shape:  {model.output_shape}")
    msgr.info(f"actual input_shape:  {input_shape}")
    msgr.info(f"actual output_shape: {output_shape}")
    msgr.info(f"n_filters:           {n_filters}")
    msgr.info(f"num_labels:          {num_labels}")
    return model
