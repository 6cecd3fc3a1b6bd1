"""The f16 path (BASELINE configs[4], "fp16 MFMA path" of the hyper-parameter sweep; reference strategy hpsearch.py:186-205 over the
width variants of defaults/default_hps_parameter.json:2-25): the same ResNetLSTM (architectures.py:162-241) with
  * activations stored as f16 channel-octet planes [snippet][ceil(C/8)][H + 2R][WP][8] (csrc/half_planes.h),
  * every contraction -- pointwise / residual 1x1 convolutions, LSTM input projections, Dense-128 -- on v_mfma_f32_16x16x32_f16 with
    f32 accumulation, depthwise taps in packed f16, BatchNorm / bias / activations in f32 registers,
  * f32 master weights (this module only packs f16 COPIES in the kernels' fragment layouts), the 46-step LSTM recurrences and the
    final Dense + sigmoid on the exact-f32 kernels of the f32 path.
Selected per model with ``model.precision = "f16"`` (or orcai_parameter["model"]["precision"]); the default stays f32 -- the
reference computes in f32, and the predict headline is measured in f32.  No CPU fallback, like the f32 path.
"""

from __future__ import annotations

import numpy as np
import torch

from orcai_amd import _native as N
from orcai_amd.architectures import DENSE_UNITS, ENTRY_FILTERS, FINAL_FILTERS, lstm_column_permutation

LOSS_SCALE = 1024.0  # static loss scale of the f16 backward pass (gradients of O(1e-4) activations stay normal f16 numbers)


# ------------------------------------------------------------------ host-side packers (f32 master weights -> f16 kernel layouts)
def pack_depthwise_octets(dwk: np.ndarray) -> np.ndarray:
    """Keras depthwise kernel (k,k,C,1) -> f16 [ceil(C/8)][k*k][8] (channel innermost inside an octet, zero taps for padding channels)."""
    k, c = dwk.shape[0], dwk.shape[2]
    co = (c + 7) // 8
    out = np.zeros((co * 8, k * k), dtype=np.float32)
    out[:c] = dwk[:, :, :, 0].transpose(2, 0, 1).reshape(c, k * k)
    return np.ascontiguousarray(out.reshape(co, 8, k * k).transpose(0, 2, 1)).astype(np.float16)


def pack_pointwise_fragments(pw: np.ndarray) -> np.ndarray:
    """W[Cin][Cout] -> A fragments of v_mfma_f32_16x16x32_f16, f16 [KG][MT][64][8]:
    element (kg, m, lane, e) = W[32 kg + 8 (lane >> 4) + e][16 m + (lane & 15)], zero outside the matrix."""
    cin, cout = pw.shape
    kg, mt = (cin + 31) // 32, (cout + 15) // 16
    padded = np.zeros((kg * 32, mt * 16), dtype=np.float32)
    padded[:cin, :cout] = pw
    lane = np.arange(64)
    ci = (np.arange(kg)[:, None, None, None] * 32 + 8 * (lane >> 4)[None, None, :, None] + np.arange(8)[None, None, None, :])
    co = (np.arange(mt)[None, :, None, None] * 16 + (lane & 15)[None, None, :, None])
    return np.ascontiguousarray(padded[ci, co]).astype(np.float16)


def pack_transposed(W: np.ndarray) -> np.ndarray:
    """W[K][N] -> f16 Wt[N][roundup32(K)] (zero padded): both operands of orcai_h_gemm_bias_act are 16-byte runs along k."""
    K, Nn = W.shape
    kp = (K + 31) // 32 * 32
    out = np.zeros((Nn, kp), dtype=np.float16)
    out[:, :K] = W.T.astype(np.float16)
    return out


class HalfEngine:
    """Inference forward of a ResNetLSTM on the f16 path.  Holds the f16 weight copies and the octet-plane workspaces."""

    def __init__(self, model):
        if getattr(model, "architecture", "") != "ResNetLSTM":
            raise NotImplementedError("the f16 path implements the ResNetLSTM architecture (the one the hyper-parameter sweep trains)")
        self.m = model
        self._dev = None
        self._ws = {}

    # ------------------------------------------------------------------ weights
    def _up(self, a, dtype=None):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda() if dtype is None else torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()

    def prepare(self) -> dict:
        m = self.m
        if self._dev is not None and self._dev["_src"] is m.weights:
            return self._dev
        if not torch.cuda.is_available():
            raise RuntimeError("orcai_amd model needs a ROCm GPU: there is no CPU fallback")
        N.lib()
        w, k = m.weights, m.kernel_size
        d = {"_src": w}
        d["conv0/w"] = self._up(w["conv0/kernel"].reshape(k * k, ENTRY_FILTERS), np.float32)
        d["conv0/scale"], d["conv0/shift"] = m._fold_bn("bn0", w["conv0/bias"])

        def sep(name, bn):
            d[name + "/dw"] = self._up(pack_depthwise_octets(w[name + "/depthwise"]))
            d[name + "/pw"] = self._up(pack_pointwise_fragments(w[name + "/pointwise"][0, 0]))
            d[name + "/scale"], d[name + "/shift"] = m._fold_bn(bn, w[name + "/bias"])

        for b in range(1, len(m.filters) + 1):
            sep(f"b{b}/sep_a", f"b{b}/bn_a")
            sep(f"b{b}/sep_b", f"b{b}/bn_b")
            d[f"b{b}/res/w"] = self._up(pack_pointwise_fragments(w[f"b{b}/res/kernel"][0, 0]))
            d[f"b{b}/res/b"] = self._up(w[f"b{b}/res/bias"], np.float32)
        sep("sep_f", "bn_f")
        perm = lstm_column_permutation(m.lstm_units)
        for layer in (1, 2):
            Wc = np.concatenate([w[f"lstm{layer}/{dd}/kernel"][:, perm] for dd in ("fwd", "bwd")], axis=1)  # [Fin][2*4u]
            d[f"lstm{layer}/Wt"] = self._up(pack_transposed(Wc))
            d[f"lstm{layer}/b"] = self._up(np.concatenate([w[f"lstm{layer}/{dd}/bias"][perm] for dd in ("fwd", "bwd")]), np.float32)
            d[f"lstm{layer}/U"] = self._up(np.stack([w[f"lstm{layer}/{dd}/recurrent"][:, perm] for dd in ("fwd", "bwd")]), np.float32)
        d["dense1/Wt"] = self._up(pack_transposed(w["dense1/kernel"]))
        d["dense1/b"] = self._up(w["dense1/bias"], np.float32)
        d["dense1/scale"], d["dense1/shift"] = m._fold_bn("bn_d")
        d["dense2/W"] = self._up(w["dense2/kernel"], np.float32)
        d["dense2/b"] = self._up(w["dense2/bias"], np.float32)
        self._dev = d
        return d

    # ------------------------------------------------------------------ workspaces
    def planes(self, B, c, h, w):
        m = self.m
        R = m.kernel_size // 2
        return torch.zeros((B, (c + 7) // 8, h + 2 * R, m.padded_width(w), 8), dtype=torch.float16, device="cuda")

    def _buffers(self, B: int) -> dict:
        have = self._ws.get("trunk")
        if have is not None and have[0] >= B:
            return have[1]
        m = self.m
        shapes = m.stage_shapes()
        ws = {"prev0": self.planes(B, shapes[0][2], shapes[0][0], shapes[0][1])}
        for b, f in enumerate(m.filters, start=1):
            h, wd, _ = shapes[b - 1]
            ws[f"a{b}"] = self.planes(B, f, h, wd)
            wx = (wd + 1) // 2
            ws[f"b{b}"] = torch.zeros((B, (f + 7) // 8, h, (wx + 3) & ~3, 8), dtype=torch.float16, device="cuda")
            ws[f"prev{b}"] = self.planes(B, f, shapes[b][0], shapes[b][1])
        self._ws["trunk"] = (B, ws)
        return ws

    # ------------------------------------------------------------------ forward
    def trunk(self, src: torch.Tensor, snippet_stride: int, B: int, feat: torch.Tensor, keep: dict | None = None) -> None:
        m, lib, d, st = self.m, N.lib(), self.prepare(), N.stream_ptr()
        ws = self._buffers(B)
        H, W = m.input_hw
        k = m.kernel_size
        shapes = m.stage_shapes()
        launch = m._launch
        launch("h/conv0", "orcai_h_conv0_affine", lib.orcai_h_conv0_affine, src.data_ptr(), snippet_stride, B, H, W, k, N.ptr(d["conv0/w"]), N.ptr(d["conv0/scale"]),
               N.ptr(d["conv0/shift"]), 1, N.ptr(ws["prev0"]), st)
        for b in range(1, len(m.filters) + 1):
            f, c = m.filters[b - 1], shapes[b - 1][2]
            h, wd, _ = shapes[b - 1]
            prev, a, bb, nxt = ws[f"prev{b - 1}"], ws[f"a{b}"], ws[f"b{b}"], ws[f"prev{b}"]
            pa, pb = f"b{b}/sep_a", f"b{b}/sep_b"
            launch("h/" + pa, "orcai_h_sepconv", lib.orcai_h_sepconv, N.ptr(prev), B, c, h, wd, k, k, 1, N.ptr(d[pa + "/dw"]), N.ptr(d[pa + "/pw"]),
                   N.ptr(d[pa + "/scale"]), N.ptr(d[pa + "/shift"]), f, 1, 0, 0, 0, N.ptr(a), None, st)
            launch("h/" + pb, "orcai_h_sepconv", lib.orcai_h_sepconv, N.ptr(a), B, f, h, wd, k, k, 0, N.ptr(d[pb + "/dw"]), N.ptr(d[pb + "/pw"]),
                   N.ptr(d[pb + "/scale"]), N.ptr(d[pb + "/shift"]), f, 0, 2, 0, 0, N.ptr(bb), None, st)
            launch(f"h/b{b}/pool_res", "orcai_h_pool_res_add", lib.orcai_h_pool_res_add, N.ptr(bb), N.ptr(prev), B, f, c, h, wd, k, N.ptr(d[f"b{b}/res/w"]),
                   N.ptr(d[f"b{b}/res/b"]), N.ptr(nxt), 1, None, None, None, None, 0.0, st)
        h, wd, c = shapes[-1]
        launch("h/sep_f", "orcai_h_sepconv", lib.orcai_h_sepconv, N.ptr(ws[f"prev{len(m.filters)}"]), B, c, h, wd, k, k, 0, N.ptr(d["sep_f/dw"]), N.ptr(d["sep_f/pw"]),
               N.ptr(d["sep_f/scale"]), N.ptr(d["sep_f/shift"]), FINAL_FILTERS, 1, 1, 0, 0, feat.data_ptr(), None, st)
        if keep is not None:  # test hook: planes back to f32 [B][C][H][W]
            R = k // 2
            chans = {"prev0": ENTRY_FILTERS}
            widths = {"prev0": shapes[0][1]}
            for i, f in enumerate(m.filters, start=1):
                chans.update({f"a{i}": f, f"b{i}": f, f"prev{i}": f})
                widths.update({f"a{i}": shapes[i - 1][1], f"b{i}": shapes[i - 1][1], f"prev{i}": shapes[i][1]})
            for name, t in ws.items():
                t = t[:B].float()
                if name.startswith("b"):  # x-pooled: [B][CO][H][WPx][8]
                    Bq, CO, hh, WPx, _ = t.shape
                    full = t.permute(0, 1, 4, 2, 3).reshape(Bq, CO * 8, hh, WPx)
                    keep[name] = full[:, : chans[name], :, : (widths[name] + 1) // 2].clone()
                    continue
                Bq, CO, HPp, WPp, _ = t.shape
                hh = HPp - 2 * R
                full = t.permute(0, 1, 4, 2, 3).reshape(Bq, CO * 8, HPp, WPp)
                keep[name] = full[:, : chans[name], R : R + hh, : widths[name]].clone()
                pads = full.clone()
                pads[:, : chans[name], R : R + hh, : widths[name]] = 0
                keep[name + "/pads"] = pads

    def head(self, feat: torch.Tensor, out: torch.Tensor, keep: dict | None = None) -> None:
        m, lib, d, st = self.m, N.lib(), self.prepare(), N.stream_ptr()
        n, h, fin = int(feat.shape[0]), int(feat.shape[1]), int(feat.shape[2])
        u = m.lstm_units
        dev = feat.device
        xz = torch.empty((n, h, 2, 4 * u), dtype=torch.float32, device=dev)
        h1 = torch.empty((n, h, 2 * u), dtype=torch.float32, device=dev)
        h2 = torch.empty((n, h, 2 * u), dtype=torch.float32, device=dev)
        M = n * h
        x = feat
        launch = m._launch
        for layer, hout in ((1, h1), (2, h2)):
            launch(f"h/lstm{layer}/gemm", "orcai_h_gemm_bias_act", lib.orcai_h_gemm_bias_act, N.ptr(x), N.ptr(d[f"lstm{layer}/Wt"]), N.ptr(d[f"lstm{layer}/b"]), None, None,
                   N.ptr(xz), M, 8 * u, fin, 0, st)
            launch(f"h/lstm{layer}/rec", "orcai_lstm_recurrent", lib.orcai_lstm_recurrent, N.ptr(xz), N.ptr(d[f"lstm{layer}/U"]), n, h, u, N.ptr(hout), st)
            x, fin = hout, 2 * u
        d1 = xz.view(-1)[: M * DENSE_UNITS].view(n, h, DENSE_UNITS)
        launch("h/dense1", "orcai_h_gemm_bias_act", lib.orcai_h_gemm_bias_act, N.ptr(h2), N.ptr(d["dense1/Wt"]), N.ptr(d["dense1/b"]), N.ptr(d["dense1/scale"]),
               N.ptr(d["dense1/shift"]), N.ptr(d1), M, DENSE_UNITS, 2 * u, 1, st)
        launch("h/dense2", "orcai_dense_sigmoid", lib.orcai_dense_sigmoid, N.ptr(d1), N.ptr(d["dense2/W"]), N.ptr(d["dense2/b"]), M, DENSE_UNITS, m.num_labels,
               out.data_ptr(), st)
        if keep is not None:
            keep.update({"feat": feat.clone(), "h1": h1.clone(), "h2": h2.clone()})

    def forward_device(self, src: torch.Tensor, snippet_stride: int, n: int, out: torch.Tensor, chunk: int = 128, keep: dict | None = None) -> None:
        """Same contract as ResNetLSTM.forward_device (f32 spectrogram snippets in, f32 probabilities out)."""
        steps, wd, _ = self.m.stage_shapes()[-1]
        feat = torch.empty((n, steps, wd * FINAL_FILTERS), dtype=torch.float32, device=src.device)
        for s in range(0, n, chunk):
            B = min(chunk, n - s)
            self.trunk(src[s * snippet_stride :], snippet_stride, B, feat[s:], keep=keep if s == 0 else None)
        self.head(feat, out, keep=keep)
