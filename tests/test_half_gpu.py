"""The f16 path (BASELINE configs[4]: width variants of the hyper-parameter sweep on f16 MFMA) against the fp32 oracle.

* Layout tests on EXACT integer data (every product and sum is an integer below 2048, exactly representable in f16): a wrong
  fragment / octet / lane map cannot hide behind a tolerance.  Asymmetric weights, channel counts that are not multiples of 8, 16
  or 32, two K groups, all output layouts.
* Forward of the three filter sets of the reference's default_hps_parameter.json (:2-25) against oracle/model_ref.forward_ref in
  fp32: |delta p| <= 5e-3 ("<= 5e-3 fp16 path"), observed error printed.
"""

import json
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import model_ref as M  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
HPS = json.loads((ROOT / "orcai_amd" / "defaults" / "default_hps_parameter.json").read_text())


def to_octet_planes(x, ksize):
    """[B][C][H][W] -> f16 [B][ceil(C/8)][H + 2R][WP][8] with zero pads."""
    B, C, H, W = x.shape
    R = ksize // 2
    WP = (W + R + 3) & ~3
    CO = (C + 7) // 8
    out = np.zeros((B, CO * 8, H + 2 * R, WP), dtype=np.float16)
    out[:, :C, R : R + H, :W] = x
    return np.ascontiguousarray(out.reshape(B, CO, 8, H + 2 * R, WP).transpose(0, 1, 3, 4, 2))


def from_octet_planes(p, C, H, W, ksize):
    B, CO, HP, WP, _ = p.shape
    R = ksize // 2
    full = p.transpose(0, 1, 4, 2, 3).reshape(B, CO * 8, HP, WP)
    pads = full.copy()
    pads[:, :C, R : R + H, :W] = 0
    return full[:, :C, R : R + H, :W], pads


def _sepconv_int_ref(x, dwk, pw, relu_in):
    """x [B][C][H][W] ints, dwk (k,k,C) ints, pw [C][Cout] ints -> [B][Cout][H][W] (same padding)."""
    B, C, H, W = x.shape
    k = dwk.shape[0]
    R = k // 2
    xr = np.maximum(x, 0) if relu_in else x
    xp = np.zeros((B, C, H + 2 * R, W + 2 * R), dtype=np.int64)
    xp[:, :, R : R + H, R : R + W] = xr
    u = np.zeros((B, C, H, W), dtype=np.int64)
    for dy in range(k):
        for dx in range(k):
            u += xp[:, :, dy : dy + H, dx : dx + W] * dwk[dy, dx][None, :, None, None]
    return np.einsum("bchw,cd->bdhw", u, pw), u


@pytest.mark.parametrize("Cin,Cout,k,H,W", [(40, 24, 3, 9, 21), (16, 30, 3, 12, 62), (13, 36, 5, 8, 17), (64, 64, 3, 6, 33), (20, 10, 7, 10, 15)])
def test_sepconv_h_layouts_exact_integers(Cin, Cout, k, H, W):
    from orcai_amd import _native as N
    from orcai_amd.half import pack_depthwise_octets, pack_pointwise_fragments

    lib = N.lib()
    rng = np.random.default_rng(Cin * 100 + Cout)
    B = 2
    x = rng.integers(-2, 3, size=(B, Cin, H, W))
    dwk = rng.integers(-1, 2, size=(k, k, Cin)) if k == 3 else (rng.random((k, k, Cin)) < 0.2).astype(np.int64) * rng.integers(-1, 2, size=(k, k, Cin))
    pw = rng.integers(-1, 3, size=(Cin, Cout)) * (rng.random((Cin, Cout)) < (0.5 if Cin <= 40 else 0.25))  # asymmetric, keeps |sum| < 2048
    for relu_in in (0, 1):
        want, u_want = _sepconv_int_ref(x, dwk, pw, relu_in)
        assert np.abs(want).max() < 2048 and np.abs(u_want).max() < 2048
        xin = torch.from_numpy(to_octet_planes(x.astype(np.float16), k)).cuda()
        dwd = torch.from_numpy(pack_depthwise_octets(dwk[..., None].astype(np.float32))).cuda()
        pwd = torch.from_numpy(pack_pointwise_fragments(pw.astype(np.float32))).cuda()
        ones, zeros = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
        R = k // 2
        WP = (W + R + 3) & ~3
        CO, COo = (Cin + 7) // 8, (Cout + 7) // 8
        # layout 0 (+ the depthwise output u)
        out = torch.zeros((B, COo, H + 2 * R, WP, 8), dtype=torch.float16, device="cuda")
        u = torch.zeros((B, CO, H + 2 * R, WP, 8), dtype=torch.float16, device="cuda")
        N.check(lib.orcai_h_sepconv(N.ptr(xin), B, Cin, H, W, k, k, relu_in, N.ptr(dwd), N.ptr(pwd), N.ptr(ones), N.ptr(zeros), Cout, 0, 0, 0, 0, N.ptr(out), N.ptr(u),
                                    N.stream_ptr()), "h_sepconv")
        got, pads = from_octet_planes(out.cpu().numpy().astype(np.float64), Cout, H, W, k)
        assert np.array_equal(got, want), np.abs(got - want).max()
        assert not pads.any()  # pads and padding channels stay zero
        gu, upads = from_octet_planes(u.cpu().numpy().astype(np.float64), Cin, H, W, k)
        assert np.array_equal(gu, u_want) and not upads.any()
        # layout 2: x-pooled (max over column pairs), with ReLU on the output
        Wx = (W + 1) // 2
        WPx = (Wx + 3) & ~3
        outx = torch.zeros((B, COo, H, WPx, 8), dtype=torch.float16, device="cuda")
        N.check(lib.orcai_h_sepconv(N.ptr(xin), B, Cin, H, W, k, k, relu_in, N.ptr(dwd), N.ptr(pwd), N.ptr(ones), N.ptr(zeros), Cout, 1, 2, 0, 0, N.ptr(outx), None,
                                    N.stream_ptr()), "h_sepconv")
        wr = np.maximum(want, 0)
        wpad = np.full((B, Cout, H, 2 * Wx), -1, dtype=np.int64)
        wpad[..., :W] = wr
        wantx = wpad.reshape(B, Cout, H, Wx, 2).max(axis=-1)
        gx = outx.cpu().numpy().astype(np.float64).transpose(0, 1, 4, 2, 3).reshape(B, COo * 8, H, WPx)[:, :Cout, :, :Wx]
        assert np.array_equal(gx, wantx)
        # layout 1: Keras Reshape, f32
        feat = torch.zeros((B, H, W * Cout), dtype=torch.float32, device="cuda")
        N.check(lib.orcai_h_sepconv(N.ptr(xin), B, Cin, H, W, k, k, relu_in, N.ptr(dwd), N.ptr(pwd), N.ptr(ones), N.ptr(zeros), Cout, 0, 1, 0, 0, N.ptr(feat), None,
                                    N.stream_ptr()), "h_sepconv")
        assert np.array_equal(feat.cpu().numpy().reshape(B, H, W, Cout).transpose(0, 3, 1, 2), want)


@pytest.mark.parametrize("C,Cp,H,W", [(30, 16, 12, 21), (12, 40, 9, 14), (64, 50, 6, 9)])
def test_pool_res_add_h_exact_integers(C, Cp, H, W):
    """MaxPooling2D((3,2), 2, same) + strided 1x1 residual on integer data: x-pooled and plane inputs, odd and even sizes."""
    from orcai_amd import _native as N
    from orcai_amd.half import pack_pointwise_fragments

    lib = N.lib()
    rng = np.random.default_rng(C + 7 * Cp)
    B, k = 2, 3
    s = rng.integers(-50, 50, size=(B, C, H, W))
    prev = rng.integers(-3, 4, size=(B, Cp, H, W))
    wr = rng.integers(-1, 2, size=(Cp, C))
    br = rng.integers(-3, 4, size=C)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    pad_h = max((Ho - 1) * 2 + 3 - H, 0)
    sp = np.full((B, C, 2 * Ho + 1 + 2, 2 * Wo + 2), -10**6, dtype=np.int64)
    top = pad_h // 2
    sp[:, :, top : top + H, :W] = s
    pooled = np.stack([sp[:, :, 2 * i : 2 * i + 3, :][:, :, :, : 2 * Wo].reshape(B, C, 3, Wo, 2).max(axis=(2, 4)) for i in range(Ho)], axis=2)
    res = np.einsum("bchw,cd->bdhw", prev[:, :, ::2, ::2], wr) + br[None, :, None, None]
    want = pooled + res
    assert np.abs(want).max() < 2048
    wrd = torch.from_numpy(pack_pointwise_fragments(wr.astype(np.float32))).cuda()
    brd = torch.from_numpy(br.astype(np.float32)).cuda()
    prevd = torch.from_numpy(to_octet_planes(prev.astype(np.float16), k)).cuda()
    R = 1
    WPo = (Wo + R + 3) & ~3
    CO = (C + 7) // 8
    for xpooled in (0, 1):
        if xpooled:
            if W % 2:  # "same" pads a column on the right for odd W: the x-pooled layout covers it (the pair's second column is ignored)
                pass
            Wx = (W + 1) // 2
            WPx = (Wx + 3) & ~3
            spad = np.full((B, C, H, 2 * Wx), -10**6, dtype=np.int64)
            spad[..., :W] = s
            sx = np.zeros((B, CO * 8, H, WPx), dtype=np.float16)
            sx[:, :C, :, :Wx] = spad.reshape(B, C, H, Wx, 2).max(axis=-1)
            sd = torch.from_numpy(np.ascontiguousarray(sx.reshape(B, CO, 8, H, WPx).transpose(0, 1, 3, 4, 2))).cuda()
        else:
            sd = torch.from_numpy(to_octet_planes(s.astype(np.float16), k)).cuda()
        out = torch.zeros((B, CO, Ho + 2 * R, WPo, 8), dtype=torch.float16, device="cuda")
        N.check(lib.orcai_h_pool_res_add(N.ptr(sd), N.ptr(prevd), B, C, Cp, H, W, k, N.ptr(wrd), N.ptr(brd), N.ptr(out), xpooled, None, None, None, None, 0.0,
                                         N.stream_ptr()), "h_pool_res_add")
        got, pads = from_octet_planes(out.cpu().numpy().astype(np.float64), C, Ho, Wo, k)
        assert np.array_equal(got, want), (xpooled, np.abs(got - want).max())
        assert not pads.any()


def test_gemm_h_exact_integers():
    from orcai_amd import _native as N
    from orcai_amd.half import pack_transposed

    lib = N.lib()
    rng = np.random.default_rng(3)
    for M_, K, Nn, act in ((70, 396, 130, 0), (33, 256, 128, 1), (5, 44, 7, 0)):
        A = rng.integers(-2, 3, size=(M_, K)).astype(np.float32)
        W = (rng.integers(-1, 2, size=(K, Nn)) * (rng.random((K, Nn)) < 0.3)).astype(np.float32)
        bias = rng.integers(-4, 5, size=Nn).astype(np.float32)
        want = A @ W + bias
        if act:
            want = np.maximum(want, 0) * 2.0 + 1.0
        Ad, Wt, bd = torch.from_numpy(A).cuda(), torch.from_numpy(pack_transposed(W)).cuda(), torch.from_numpy(bias).cuda()
        two, one = torch.full((Nn,), 2.0, device="cuda"), torch.ones(Nn, device="cuda")
        C = torch.empty((M_, Nn), dtype=torch.float32, device="cuda")
        N.check(lib.orcai_h_gemm_bias_act(N.ptr(Ad), N.ptr(Wt), N.ptr(bd), N.ptr(two) if act else None, N.ptr(one) if act else None, N.ptr(C), M_, Nn, K, act,
                                          N.stream_ptr()), "h_gemm")
        assert np.array_equal(C.cpu().numpy(), want)


@pytest.mark.parametrize("fset", ["set1", "set2", "set3"])
@pytest.mark.parametrize("kernel_size,units", [(3, 128), (5, 64)])
def test_half_forward_vs_fp32_oracle(fset, kernel_size, units):
    """configs[4]: the three filter sets of default_hps_parameter.json, full orcai-V1 input shape, against the fp32 oracle."""
    from orcai_amd.architectures import ResNetLSTM

    filters = HPS["filters"][fset]
    cfg = dict(input_shape=(736, 171, 1), num_labels=7, filters=tuple(filters), kernel_size=kernel_size, lstm_units=units)
    p = M.calibrated_params(seed=31, calib_batch=1, **cfg)
    model = ResNetLSTM(cfg["input_shape"], 7, list(filters), kernel_size, 0.0, units, precision="f16")
    model.set_weights_dict(p)
    x = np.random.default_rng(6).random((3, 736, 171, 1), dtype=np.float32)
    got = model.predict(x, batch_size=3)
    ref = M.forward_ref(p, x)
    err = float(np.abs(got - ref).max())
    model.precision = "f32"
    err32 = float(np.abs(model.predict(x, batch_size=3) - ref).max())
    print(f"f16 path {fset} k={kernel_size} u={units}: max|dp| vs fp32 oracle = {err:.2e} (f32 path {err32:.1e}); mean|dp| = {float(np.abs(got - ref).mean()):.2e}")
    assert err <= 5e-3, err


def test_half_forward_intermediates():
    """Every intermediate of the f16 trunk against the oracle's, relative to the tensor's scale (f16 storage: 2^-11 per rounding)."""
    from orcai_amd.architectures import ResNetLSTM

    cfg = dict(input_shape=(64, 45, 1), num_labels=5, filters=(12, 30, 44), kernel_size=3, lstm_units=64)
    p = M.calibrated_params(seed=17, **cfg)
    model = ResNetLSTM(cfg["input_shape"], 5, list(cfg["filters"]), 3, 0.0, 64, precision="f16")
    model.set_weights_dict(p)
    x = np.random.default_rng(2).random((4, 64, 45, 1), dtype=np.float32)
    keep = {}
    xd = torch.from_numpy(np.ascontiguousarray(x[..., 0])).cuda()
    out = torch.empty((4, 8, 5), dtype=torch.float32, device="cuda")
    model.forward_device(xd.view(-1), 64 * 45, 4, out, chunk=4, keep=keep)
    ref, inter = M.forward_ref(p, x, return_intermediates=True)
    worst = {}
    for name, rname in (("prev0", "conv0"), ("a1", "b1/a"), ("prev1", "b1"), ("a2", "b2/a"), ("prev2", "b2"), ("a3", "b3/a"), ("prev3", "b3")):
        r = np.asarray(inter[rname])
        g = keep[name].cpu().numpy()
        assert g.shape == r.shape, (name, g.shape, r.shape)
        worst[name] = float(np.abs(g - r).max() / max(1.0, np.abs(r).max()))
        assert not keep[name + "/pads"].any(), name
    print("f16 intermediates, max|delta| / max(1, max|ref|):", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) <= 2e-2, worst
    assert np.abs(out.cpu().numpy() - ref).max() <= 5e-3
