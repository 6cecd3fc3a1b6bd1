"""Round-3 fusions of the f32 training step, each against the launches it replaces and against float64 arithmetic.

orcai_bn_bwd_pointwise_wgrad: BatchNorm backward apply + du = Wpw dv + the pointwise weight gradient u (x) dv in one pass (dv is never
written) against orcai_bn_bwd_pointwise + orcai_outer_reduce."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _quad_planes(x, ksize):
    B, C, H, W = x.shape
    R = ksize // 2
    WP = (W + R + 3) & ~3
    CQ = (C + 3) // 4
    out = np.zeros((B, CQ * 4, H + 2 * R, WP), dtype=np.float32)
    out[:, :C, R : R + H, :W] = x
    return np.ascontiguousarray(out.reshape(B, CQ, 4, H + 2 * R, WP).transpose(0, 1, 3, 4, 2))


def _from_quad(p, C, H, W, ksize):
    B, CQ, HP, WP, _ = p.shape
    R = ksize // 2
    return p.transpose(0, 1, 4, 2, 3).reshape(B, CQ * 4, HP, WP)[:, :C, R : R + H, :W]


@pytest.mark.parametrize("C,Cin,H,W,B,relu,ready", [(30, 16, 12, 21, 2, 1, 0), (30, 30, 37, 171, 3, 0, 1), (10, 12, 9, 14, 5, 1, 0), (20, 30, 8, 70, 2, 0, 0),
                                                   (32, 32, 5, 6, 1, 1, 0), (40, 30, 17, 86, 2, 1, 0), (40, 40, 9, 86, 3, 0, 1), (50, 40, 7, 9, 2, 0, 0)])
def test_bn_bwd_pointwise_wgrad_vs_two_kernels(C, Cin, H, W, B, relu, ready):
    from orcai_amd import _native as N

    lib = N.lib()
    prev_tiles = lib.orcai_pw_wgrad_tiles(6)  # the launcher's default stops at 4 tiles (two-wave workgroups measured slower); the kernel itself is tested to 6
    try:
        _pw_wgrad_case(lib, N, C, Cin, H, W, B, relu, ready)
    finally:
        lib.orcai_pw_wgrad_tiles(prev_tiles)


def _pw_wgrad_case(lib, N, C, Cin, H, W, B, relu, ready):
    rng = np.random.default_rng(C * 7 + Cin + H)
    k = 3
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    dy, v, u = f(B, C, H, W), 2.0 * f(B, C, H, W), f(B, Cin, H, W)
    mean, var = 0.3 * f(C), (0.5 + rng.random(C)).astype(np.float32)
    gamma, beta = 1 + 0.3 * f(C), 0.2 * f(C)
    wt = f(C, Cin) / 4  # pointwise^T [Cout][Cin]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    md, vd, gd, bd, wd = dev(mean), dev(var), dev(gamma), dev(beta), dev(wt)
    dyd, vdv, ud = dev(_quad_planes(dy, k)), dev(_quad_planes(v, k)), dev(_quad_planes(u, k))
    CQi = (Cin + 3) // 4
    # float64 reference of the sums, dv, du, dWpw
    inv = 1.0 / np.sqrt(var.astype(np.float64) + 1e-3)
    xh = (v.astype(np.float64) - mean[None, :, None, None]) * inv[None, :, None, None]
    de = dy.astype(np.float64)
    if relu:
        de = np.where(xh * gamma[None, :, None, None] + beta[None, :, None, None] > 0, de, 0.0)
    n = B * H * W
    dbeta_ref, dgamma_ref = de.sum(axis=(0, 2, 3)), (de * xh).sum(axis=(0, 2, 3))
    dv_ref = (gamma * inv)[None, :, None, None] * (de - dbeta_ref[None, :, None, None] / n - xh * dgamma_ref[None, :, None, None] / n)
    du_ref = np.einsum("bchw,ci->bihw", dv_ref, wt.astype(np.float64))
    dW_ref = np.einsum("bihw,bchw->ic", u.astype(np.float64), dv_ref)

    def sums(scratch):  # what orcai_pool_bwd_bn leaves behind when sums_ready = 1
        s = np.zeros(128)
        CQ = (C + 3) // 4
        s[:C], s[4 * CQ : 4 * CQ + C] = dbeta_ref, dgamma_ref
        scratch.copy_(torch.from_numpy(s))

    out = {}
    for fused in (False, True):
        dv = torch.zeros_like(dyd)
        du = torch.zeros((B, CQi) + tuple(dyd.shape[2:]), dtype=torch.float32, device="cuda")
        scratch = torch.zeros(128, dtype=torch.float64, device="cuda")
        if ready:
            sums(scratch)
        dbeta, dgamma = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dW = torch.full((Cin, C), 0.5, dtype=torch.float32, device="cuda")  # the entry points ACCUMULATE into the gradient buffer
        ws = torch.empty(512 * 64 * 64, dtype=torch.float32, device="cuda")
        st = N.stream_ptr()
        if fused:
            rc = lib.orcai_bn_bwd_pointwise_wgrad(N.ptr(dyd), N.ptr(vdv), N.ptr(ud), B, C, H, W, k, N.ptr(md), N.ptr(vd), N.ptr(gd), N.ptr(bd), 1e-3, relu, N.ptr(scratch), ready,
                                                  N.ptr(dbeta), N.ptr(dgamma), N.ptr(wd), Cin, N.ptr(du), N.ptr(dW), N.ptr(ws), ws.numel(), st)
            if (Cin + 15) // 16 + (C + 15) // 16 > 6:
                assert rc == N.E_UNSUPPORTED and float(du.abs().max()) == 0.0 and float((dW - 0.5).abs().max()) == 0.0  # refused before anything was touched
                return
            N.check(rc, "bn_bwd_pointwise_wgrad")
        else:
            N.check(lib.orcai_bn_bwd_pointwise(N.ptr(dyd), N.ptr(vdv), B, C, H, W, k, N.ptr(md), N.ptr(vd), N.ptr(gd), N.ptr(bd), 1e-3, relu, N.ptr(scratch), ready, N.ptr(dbeta),
                                               N.ptr(dgamma), N.ptr(wd), Cin, N.ptr(dv), N.ptr(du), st), "bn_bwd_pointwise")
            N.check(lib.orcai_outer_reduce(N.ptr(ud), Cin, N.ptr(dv), C, B, H, W, k, 0, 0, 0, N.ptr(dW), N.ptr(ws), ws.numel(), st), "outer_reduce")
        torch.cuda.synchronize()
        out[fused] = (_from_quad(du.cpu().numpy(), Cin, H, W, k), dW.cpu().numpy() - 0.5, dbeta.cpu().numpy(), dgamma.cpu().numpy(), du.cpu().numpy())
    a, b = out[True], out[False]
    assert np.array_equal(a[0], b[0])  # du: the same arithmetic in the same order
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    scale = max(1.0, np.abs(dW_ref).max())
    assert np.abs(a[1] - dW_ref).max() <= 2e-5 * scale * np.sqrt(n / 64 + 1), (np.abs(a[1] - dW_ref).max(), scale)
    assert np.abs(b[1] - dW_ref).max() <= 2e-5 * scale * np.sqrt(n / 64 + 1)
    assert np.abs(a[0] - du_ref).max() <= 1e-4 * max(1.0, np.abs(du_ref).max())
    # pads of du stay untouched (zero): the next kernels rely on them
    pads = a[4].copy()
    R, CQi_ = 1, (Cin + 3) // 4
    pads[:, :, R : R + H, :W, :] = 0
    assert float(np.abs(pads).max()) == 0.0


@pytest.mark.parametrize("C,H,W,B,relu", [(30, 37, 171, 2, 1), (40, 23, 86, 3, 1), (50, 12, 43, 2, 0), (60, 9, 22, 4, 1), (10, 16, 12, 5, 1), (30, 8, 171, 1, 0)])
def test_input_gradient_pass_with_epilogues(C, H, W, B, relu):
    """orcai_sepconv_planes_epi against the plain input-gradient pass (orcai_sepconv_planes_u with flipped taps and the identity pointwise
    factor): the output is bit-identical; epi 2 leaves dbeta | dgamma of the BatchNorm whose gradient the output is (float64 reference from
    the plain output and the reference tensor); epi 3 masks the output by ref > 0."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(C + H)
    k = 3
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    du, ref = f(B, C, H, W), 2.0 * f(B, C, H, W)
    mean, var = 0.3 * f(C), (0.5 + rng.random(C)).astype(np.float32)
    gamma, beta = 1 + 0.3 * f(C), 0.2 * f(C)
    CQ = (C + 3) // 4
    dw = np.zeros((CQ, 9, 4), dtype=np.float32)
    dw.reshape(CQ, 9, 4)[...] = f(CQ, 9, 4)
    dw.transpose(0, 2, 1).reshape(CQ * 4, 9)[C:] = 0  # padding channels carry zero taps
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    dud, refd, dwd = dev(_quad_planes(du, k)), dev(_quad_planes(ref, k)), dev(dw)
    eye, ones, zeros = torch.eye(C, device="cuda").contiguous(), torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    md, vd, gd, bd = dev(mean), dev(var), dev(gamma), dev(beta)
    st = N.stream_ptr()
    plain = torch.zeros_like(dud)
    N.check(lib.orcai_sepconv_planes_u(N.ptr(dud), B, C, H, W, k, k, 0, N.ptr(dwd), N.ptr(eye), N.ptr(ones), N.ptr(zeros), C, 0, 0, 0, 0, N.ptr(plain), None, st), "plain")
    dy = _from_quad(plain.cpu().numpy(), C, H, W, k).astype(np.float64)
    # epi 2
    out2 = torch.zeros_like(dud)
    shards = torch.full((8 * 16 * 32,), 7.0, dtype=torch.float64, device="cuda")
    rc = lib.orcai_sepconv_planes_epi(N.ptr(dud), B, C, H, W, N.ptr(dwd), N.ptr(eye), N.ptr(ones), N.ptr(zeros), C, N.ptr(out2), 2, N.ptr(refd), N.ptr(md), N.ptr(vd), N.ptr(gd),
                                      N.ptr(bd), 1e-3, relu, N.ptr(shards), st)
    N.check(rc, "epi 2")
    assert torch.equal(out2, plain)
    inv = 1.0 / np.sqrt(var.astype(np.float64) + 1e-3)
    xh = (ref.astype(np.float64) - mean[None, :, None, None]) * inv[None, :, None, None]
    g = np.where(xh * gamma[None, :, None, None] + beta[None, :, None, None] > 0, dy, 0.0) if relu else dy
    db_ref, dg_ref = g.sum(axis=(0, 2, 3)), (g * xh).sum(axis=(0, 2, 3))
    got = shards.cpu().numpy()
    n = B * H * W
    tol = 3e-6 * np.sqrt(n) * max(1.0, np.abs(dy).max() * 3)
    assert np.abs(got[:C] - db_ref).max() <= tol and np.abs(got[4 * CQ : 4 * CQ + C] - dg_ref).max() <= tol, (np.abs(got[:C] - db_ref).max(), np.abs(got[4 * CQ : 4 * CQ + C] - dg_ref).max(), tol)
    # the sums feed orcai_bn_bwd_pointwise with sums_ready = 1: same dv as with its own reduction pass
    wt = dev(f(C, 16) / 4)
    res = {}
    for ready in (0, 1):
        scratch = shards.clone() if ready else torch.zeros_like(shards)
        dv, du2 = torch.zeros_like(dud), torch.zeros((B, 4) + tuple(dud.shape[2:]), device="cuda")
        dbeta, dgamma = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        N.check(lib.orcai_bn_bwd_pointwise(N.ptr(plain), N.ptr(refd), B, C, H, W, k, N.ptr(md), N.ptr(vd), N.ptr(gd), N.ptr(bd), 1e-3, relu, N.ptr(scratch), ready, N.ptr(dbeta),
                                           N.ptr(dgamma), N.ptr(wt), 16, N.ptr(dv), N.ptr(du2), st), "bn_bwd_pointwise")
        res[ready] = (dv.cpu().numpy(), dbeta.cpu().numpy(), dgamma.cpu().numpy())
    scale = max(1.0, np.abs(res[0][0]).max())
    assert np.abs(res[1][0] - res[0][0]).max() <= 1e-5 * scale
    assert np.abs(res[1][1] - res[0][1]).max() <= 1e-4 * max(1.0, np.abs(res[0][1]).max()) and np.abs(res[1][2] - res[0][2]).max() <= 1e-4 * max(1.0, np.abs(res[0][2]).max())
    # epi 3
    out3 = torch.zeros_like(dud)
    rc = lib.orcai_sepconv_planes_epi(N.ptr(dud), B, C, H, W, N.ptr(dwd), N.ptr(eye), N.ptr(ones), N.ptr(zeros), C, N.ptr(out3), 3, N.ptr(refd), None, None, None, None, 0.0, 0, None, st)
    N.check(rc, "epi 3")
    assert torch.equal(out3, torch.where(refd > 0, plain, torch.zeros_like(plain)))


@pytest.mark.parametrize("Cin,Cout,H,W,B", [(30, 30, 37, 171, 2), (40, 40, 23, 86, 3), (50, 50, 12, 43, 2), (60, 60, 9, 22, 3), (10, 10, 16, 12, 4), (20, 20, 8, 6, 2), (30, 30, 9, 171, 1)])
def test_batchnorm_applied_on_load_equals_the_materialised_tensor(Cin, Cout, H, W, B):
    """orcai_sepconv_planes_stats_bn / orcai_dw_wgrad_bn (BatchNorm + ReLU of the pre-normalisation tensor formed where the conv loads it,
    zero outside the image) against orcai_bn_planes_apply followed by orcai_sepconv_planes_stats / orcai_dw_wgrad on the materialised
    tensor: conv output, depthwise output and batch-statistic sums bit for bit (the same two roundings per value), the depthwise weight
    gradient to float-atomic reordering."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(Cin + H * 3)
    k = 3
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    v, du = 2.0 * f(B, Cin, H, W), f(B, Cin, H, W)
    mean, var = 0.3 * f(Cin), (0.5 + rng.random(Cin)).astype(np.float32)
    gamma, beta = 1 + 0.3 * f(Cin), 0.4 * f(Cin) + 0.3  # beta - mean * scale != 0: the pads would NOT normalise to zero by themselves
    CQ = (Cin + 3) // 4
    dw = f(CQ, 9, 4)
    dw.transpose(0, 2, 1).reshape(CQ * 4, 9)[Cin:] = 0
    pw = (f(Cin, Cout) / np.sqrt(Cin)).astype(np.float32)  # (a numpy float64 scalar would silently promote the array)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    vd, dud, dwd, pwd = dev(_quad_planes(v, k)), dev(_quad_planes(du, k)), dev(dw), dev(pw)
    md, vard, gd, bd = dev(mean), dev(var), dev(gamma), dev(beta)
    ones, shift = torch.ones(64, device="cuda"), dev(f(64))
    st = N.stream_ptr()
    y = torch.zeros_like(vd)
    N.check(lib.orcai_bn_planes_apply(N.ptr(vd), B, Cin, H, W, k, N.ptr(md), N.ptr(vard), N.ptr(gd), N.ptr(bd), 1e-3, 1, N.ptr(y), st), "apply")
    CQo = (Cout + 3) // 4
    res = {}
    for on_load in (False, True):
        out = torch.zeros((B, CQo) + tuple(vd.shape[2:]), device="cuda")
        u = torch.zeros_like(vd)
        shards = torch.zeros(8 * 16 * 32, dtype=torch.float64, device="cuda")
        if on_load:
            rc = lib.orcai_sepconv_planes_stats_bn(N.ptr(vd), B, Cin, H, W, N.ptr(md), N.ptr(vard), N.ptr(gd), N.ptr(bd), 1e-3, N.ptr(dwd), N.ptr(pwd), N.ptr(ones), N.ptr(shift), Cout,
                                                   N.ptr(out), N.ptr(u), N.ptr(shards), st)
        else:
            rc = lib.orcai_sepconv_planes_stats(N.ptr(y), B, Cin, H, W, 0, N.ptr(dwd), N.ptr(pwd), N.ptr(ones), N.ptr(shift), Cout, N.ptr(out), N.ptr(u), N.ptr(shards), st)
        N.check(rc, "stats")
        dW = torch.zeros((9, Cin), device="cuda")
        if on_load:
            N.check(lib.orcai_dw_wgrad_bn(N.ptr(vd), N.ptr(dud), B, Cin, H, W, N.ptr(md), N.ptr(vard), N.ptr(gd), N.ptr(bd), 1e-3, N.ptr(dW), st), "dw_wgrad_bn")
        else:
            N.check(lib.orcai_dw_wgrad(N.ptr(y), N.ptr(dud), B, Cin, H, W, k, k, 0, N.ptr(dW), st), "dw_wgrad")
        torch.cuda.synchronize()
        res[on_load] = (out, u, shards, dW.cpu().numpy())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert float((res[True][2] - res[False][2]).abs().max()) <= 1e-9 * float(res[False][2].abs().max())  # f64 atomics in another order
    yr = _from_quad(y.cpu().numpy(), Cin, H, W, k).astype(np.float64)
    yp = np.zeros((B, Cin, H + 2, W + 2))
    yp[:, :, 1:-1, 1:-1] = yr
    want = np.stack([[np.einsum("bchw,bchw->c", yp[:, :, dy : dy + H, dx : dx + W], du.astype(np.float64)) for dx in range(3)] for dy in range(3)]).reshape(9, Cin)
    for r in (res[True][3], res[False][3]):
        assert np.abs(r - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("H,W,B,k", [(37, 171, 2, 3), (16, 12, 5, 3), (9, 33, 3, 5), (8, 7, 2, 7)])
def test_entry_conv_in_two_passes_without_v0(H, W, B, k):
    """orcai_conv0_stats + orcai_bn_finish_sharded + orcai_conv0_affine_bn against orcai_conv0_affine + orcai_bn_planes_stats +
    orcai_bn_planes_apply: y0 bit for bit, batch statistics to summation order; orcai_conv0_bn_bwd_x (v0 rebuilt from the input taps)
    against orcai_conv0_bn_bwd on the stored v0: gradients to float-atomic reordering."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(H * 7 + W)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    x = dev(rng.random((B, H, W), dtype=np.float32))
    w0, bias, ones = dev(f(k * k, 16) / k), dev(0.1 * f(16)), torch.ones(16, device="cuda")
    gamma, beta = dev(1 + 0.3 * f(16)), dev(0.2 * f(16))
    R, WP = k // 2, (W + k // 2 + 3) & ~3
    shape = (B, 4, H + 2 * R, WP, 4)
    st = N.stream_ptr()
    # stored path
    v0, y_a = torch.zeros(shape, device="cuda"), torch.zeros(shape, device="cuda")
    mean_a, var_a = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
    scratch = torch.zeros(8 * 16 * 32, dtype=torch.float64, device="cuda")
    N.check(lib.orcai_conv0_affine(N.ptr(x), H * W, B, H, W, k, N.ptr(w0), N.ptr(ones), N.ptr(bias), 0, N.ptr(v0), st), "conv0_affine")
    N.check(lib.orcai_bn_planes_stats(N.ptr(v0), B, 16, H, W, k, N.ptr(scratch), N.ptr(mean_a), N.ptr(var_a), st), "bn_planes_stats")
    N.check(lib.orcai_bn_planes_apply(N.ptr(v0), B, 16, H, W, k, N.ptr(mean_a), N.ptr(var_a), N.ptr(gamma), N.ptr(beta), 1e-3, 1, N.ptr(y_a), st), "apply")
    # two passes
    y_b = torch.zeros(shape, device="cuda")
    mean_b, var_b = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
    N.check(lib.orcai_conv0_stats(N.ptr(x), H * W, B, H, W, k, N.ptr(w0), N.ptr(ones), N.ptr(bias), N.ptr(scratch), st), "conv0_stats")
    N.check(lib.orcai_bn_finish_sharded(N.ptr(scratch), B, 16, H, W, N.ptr(mean_b), N.ptr(var_b), st), "finish")
    assert float((mean_a - mean_b).abs().max()) <= 2e-6 * max(1.0, float(mean_a.abs().max())) and float((var_a - var_b).abs().max()) <= 1e-5 * float(var_a.abs().max())
    # with the SAME statistics the second pass reproduces the materialised tensor bit for bit
    N.check(lib.orcai_conv0_affine_bn(N.ptr(x), H * W, B, H, W, k, N.ptr(w0), N.ptr(ones), N.ptr(bias), N.ptr(mean_a), N.ptr(var_a), N.ptr(gamma), N.ptr(beta), 1e-3, 1,
                                      N.ptr(y_b), st), "conv0_affine_bn")
    assert torch.equal(y_a, y_b)
    # backward
    dy = torch.zeros(shape, device="cuda")
    dy[:, :, R : R + H, :W, :] = dev(f(B, 4, H, W, 4))
    out = {}
    for recompute in (False, True):
        dbeta, dgamma, dW = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda"), torch.zeros((k * k, 16), device="cuda")
        if recompute:
            ws = torch.empty(512 * 64 * 64, device="cuda")
            N.check(lib.orcai_conv0_bn_bwd_x(N.ptr(x), H * W, N.ptr(dy), B, H, W, k, N.ptr(w0), N.ptr(bias), N.ptr(mean_a), N.ptr(var_a), N.ptr(gamma), N.ptr(beta), 1e-3,
                                             N.ptr(scratch), N.ptr(dbeta), N.ptr(dgamma), N.ptr(dW), N.ptr(ws), ws.numel(), st), "conv0_bn_bwd_x")
        else:
            N.check(lib.orcai_conv0_bn_bwd(N.ptr(x), H * W, N.ptr(dy), N.ptr(v0), B, H, W, k, N.ptr(mean_a), N.ptr(var_a), N.ptr(gamma), N.ptr(beta), 1e-3, N.ptr(scratch),
                                           N.ptr(dbeta), N.ptr(dgamma), N.ptr(dW), st), "conv0_bn_bwd")
        torch.cuda.synchronize()
        out[recompute] = [t.cpu().numpy() for t in (dbeta, dgamma, dW)]
    for a, b_, name in zip(out[True], out[False], ("dbeta", "dgamma", "dW")):
        assert np.abs(a - b_).max() <= 2e-5 * max(1.0, np.abs(b_).max()), (name, np.abs(a - b_).max())
    if k == 3:
        # the marching forms (csrc/train_trunk.hip conv0_march_kernel): statistics pass, and the weight-gradient pass behind sums that are already in
        # the scratch (orcai_conv0_bn_bwd_x left dbeta[16] | dgamma[16] doubles there) -- against the tile kernels
        sc2 = torch.zeros(8 * 16 * 32, dtype=torch.float64, device="cuda")
        mean_c, var_c = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
        N.check(lib.orcai_conv0_stats_march(N.ptr(x), H * W, B, H, W, N.ptr(w0), N.ptr(ones), N.ptr(bias), N.ptr(sc2), st), "conv0_stats_march")
        N.check(lib.orcai_bn_finish_sharded(N.ptr(sc2), B, 16, H, W, N.ptr(mean_c), N.ptr(var_c), st), "finish")
        assert float((mean_c - mean_b).abs().max()) <= 2e-6 * max(1.0, float(mean_b.abs().max())) and float((var_c - var_b).abs().max()) <= 1e-5 * float(var_b.abs().max())
        sums = scratch[:32].clone()
        ws = torch.empty(512 * 64 * 64, device="cuda")
        prev = lib.orcai_conv0_march(-1)
        try:
            for march in (0, 1):
                lib.orcai_conv0_march(march)
                sc3 = torch.zeros_like(scratch)
                sc3[:32] = sums
                dbeta, dgamma, dW = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda"), torch.zeros((9, 16), device="cuda")
                N.check(lib.orcai_conv0_bn_bwd_x_ready(N.ptr(x), H * W, N.ptr(dy), B, H, W, k, N.ptr(w0), N.ptr(bias), N.ptr(mean_a), N.ptr(var_a), N.ptr(gamma), N.ptr(beta), 1e-3,
                                                       N.ptr(sc3), N.ptr(dbeta), N.ptr(dgamma), N.ptr(dW), N.ptr(ws), ws.numel(), st), "conv0_bn_bwd_x_ready")
                torch.cuda.synchronize()
                for a, b_, name in zip((dbeta, dgamma, dW), out[True], ("dbeta", "dgamma", "dW")):
                    assert np.abs(a.cpu().numpy() - b_).max() <= 2e-5 * max(1.0, np.abs(b_).max()), (march, name, np.abs(a.cpu().numpy() - b_).max())
        finally:
            lib.orcai_conv0_march(prev)


@pytest.mark.parametrize("U,B,T", [(128, 20, 46), (64, 7, 12)])
def test_lstm_recurrence_on_split_f16_mfma_keeps_f32_accuracy(U, B, T):
    """orcai_lstm_train_fwd with the recurrent product on f16 MFMA, every operand split as hi + lo / 4096 (the default), against the
    v_mfma_f32_16x16x4_f32 kernel and against a float64 recurrence in the kernels' column order: the split kernel is as close to float64 as
    the f32-MFMA kernel is (both limited by f32 accumulation and the fast exp of the gate functions), over all 46 dependent steps."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(U + B)
    xz = rng.standard_normal((B, T, 2, 4 * U)).astype(np.float32)
    Uw = (rng.standard_normal((2, U, 4 * U)) * (0.8 / np.sqrt(U))).astype(np.float32)
    xd, ud = torch.from_numpy(xz).cuda(), torch.from_numpy(Uw).cuda()
    res = {}
    prev = lib.orcai_lstm_split(-1)
    try:
        for split in (0, 1):
            lib.orcai_lstm_split(split)
            h = torch.zeros((B, T, 2 * U), device="cuda")
            g = torch.zeros((B, T, 2, 4 * U), device="cuda")
            c = torch.zeros((B, T, 2, U), device="cuda")
            N.check(lib.orcai_lstm_train_fwd(N.ptr(xd), N.ptr(ud), B, T, U, N.ptr(h), N.ptr(g), N.ptr(c), N.stream_ptr()), "lstm_train_fwd")
            torch.cuda.synchronize()
            res[split] = (h.cpu().numpy(), g.cpu().numpy(), c.cpu().numpy())
    finally:
        lib.orcai_lstm_split(prev)
    # float64 recurrence in the permuted column order: wave w owns columns [32 w, 32 w + 32) = (i, f | g, o) of units 8 w .. 8 w + 7
    sig = lambda a: 1.0 / (1.0 + np.exp(-a))  # noqa: E731
    href = np.zeros((B, T, 2 * U))
    for d in range(2):
        hh, cc = np.zeros((B, U)), np.zeros((B, U))
        W = Uw[d].astype(np.float64)
        for step in range(T):
            t = T - 1 - step if d else step
            z = xz[:, t, d].astype(np.float64) + hh @ W
            zz = z.reshape(B, U // 8, 4, 8)  # [wave][i, f, g, o][unit in wave]
            i_, f_, g_, o_ = sig(zz[:, :, 0]), sig(zz[:, :, 1]), np.tanh(zz[:, :, 2]), sig(zz[:, :, 3])
            cc = (f_ * cc.reshape(B, U // 8, 8) + i_ * g_).reshape(B, U)
            hh = (o_ * np.tanh(cc.reshape(B, U // 8, 8))).reshape(B, U)
            href[:, t, d * U : (d + 1) * U] = hh
    e32, esp = np.abs(res[0][0] - href).max(), np.abs(res[1][0] - href).max()
    print(f"LSTM units {U}: max |h - float64| f32-MFMA kernel {e32:.2e}, split-f16 kernel {esp:.2e}; kernels against each other {np.abs(res[0][0] - res[1][0]).max():.2e}")
    assert esp <= max(2.0 * e32, 2e-6) and esp <= 1e-5
    for a, b_ in zip(res[0], res[1]):
        assert np.abs(a - b_).max() <= 1e-5


@pytest.mark.parametrize("U,B,T,gscale", [(128, 20, 46, 1e-4), (64, 7, 12, 3.0), (128, 3, 9, 1e-9)])
def test_lstm_backward_on_split_f16_mfma_keeps_f32_accuracy(U, B, T, gscale):
    """orcai_lstm_bwd with the recurrent term dz U^T on split-f16 MFMA (every dz scaled by the power of two that brings max |dH| into
    [0.5, 1), then hi + lo / 4096) against the v_mfma_f32_16x16x4_f32 kernel, for incoming gradients of very different magnitudes: the
    two agree to f32 accumulation noise relative to the largest gradient, over all dependent steps."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(U + T)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    gates = (1.0 / (1.0 + np.exp(-f(B, T, 2, 4 * U)))).astype(np.float32)  # i, f, o in (0, 1); the g columns are re-drawn in (-1, 1) below
    gv = gates.reshape(B, T, 2, U // 8, 4, 8)  # [wave][i f g o][unit]... the kernel's permuted order: (i, f | g, o) per 32 columns
    gv[..., 2, :] = np.tanh(f(B, T, 2, U // 8, 8))
    cst = f(B, T, 2, U) * 0.7
    dH = (f(B, T, 2 * U) * gscale).astype(np.float32)
    dH[0, 0, :5] = 0.0
    Uw = (f(2, U, 4 * U) * (0.8 / np.sqrt(U))).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    gd, cd, hd, ud = dev(gates), dev(cst), dev(dH), dev(Uw)
    res = {}
    prev = lib.orcai_lstm_split(-1)
    try:
        for split in (0, 1, 1):
            lib.orcai_lstm_split(split)
            dxz = torch.zeros((B, T, 2, 4 * U), device="cuda")
            N.check(lib.orcai_lstm_bwd(N.ptr(hd), N.ptr(gd), N.ptr(cd), N.ptr(ud), B, T, U, N.ptr(dxz), N.stream_ptr()), "lstm_bwd")
            torch.cuda.synchronize()
            res.setdefault(split, []).append(dxz.cpu().numpy().astype(np.float64))
    finally:
        lib.orcai_lstm_split(prev)
    ref, got = res[0][0], res[1][0]
    assert np.array_equal(res[1][0], res[1][1])  # the scale of a launch does not leak into the next one
    scale = np.abs(ref).max()
    assert scale > 0 and np.isfinite(got).all()
    err = np.abs(got - ref).max() / scale
    print(f"LSTM backward units {U}, |dH| ~ {gscale:g}: max |split - f32| / max |dxz| = {err:.2e}")
    assert err <= 5e-6


@pytest.mark.parametrize("C,H,W,B,relu", [(16, 37, 171, 2, 1), (30, 8, 130, 3, 0), (10, 70, 101, 1, 1), (30, 736, 171, 1, 0)])
def test_marching_depthwise_weight_gradient(C, H, W, B, relu):
    """dw_wgrad_march_kernel (wide planes: a wave walks down a 64-column strip, every x row and gradient row requested once) against the
    flat-window kernel and float64: segments, strips, the ReLU on load and planes whose last strip hangs over the row pitch."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(C + W)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    x, du = f(B, C, H, W), f(B, C, H, W)
    xd, dud = torch.from_numpy(_quad_planes(x, 3)).cuda(), torch.from_numpy(_quad_planes(du, 3)).cuda()
    xr = np.maximum(x, 0) if relu else x
    xp = np.zeros((B, C, H + 2, W + 2))
    xp[:, :, 1:-1, 1:-1] = xr
    want = np.stack([[np.einsum("bchw,bchw->c", xp[:, :, dy : dy + H, dx : dx + W], du.astype(np.float64)) for dx in range(3)] for dy in range(3)]).reshape(9, C)
    got = {}
    prev = lib.orcai_dw_wgrad_march(-1)
    try:
        for march in (0, 1):
            lib.orcai_dw_wgrad_march(march)
            dW = torch.zeros((9, C), device="cuda")
            N.check(lib.orcai_dw_wgrad(N.ptr(xd), N.ptr(dud), B, C, H, W, 3, 3, relu, N.ptr(dW), N.stream_ptr()), "dw_wgrad")
            got[march] = dW.cpu().numpy()
    finally:
        lib.orcai_dw_wgrad_march(prev)
    tol = 2e-5 * np.sqrt(B * H * W) * max(1.0, np.abs(want).max() / np.sqrt(B * H * W))
    for march in (0, 1):
        assert np.abs(got[march] - want).max() <= max(tol, 1e-4 * np.abs(want).max()), (march, np.abs(got[march] - want).max(), np.abs(want).max())


@pytest.mark.parametrize("C,H,W,B,mode", [(30, 37, 171, 2, 2), (16, 40, 171, 2, 0), (30, 23, 86, 3, 3), (40, 25, 86, 2, 2), (50, 31, 43, 2, 2), (40, 12, 43, 3, 3),
                                          (60, 9, 22, 4, 2), (50, 7, 22, 2, 3), (10, 16, 12, 5, 2), (30, 736, 171, 1, 2), (30, 5, 62, 1, 3), (36, 11, 63, 2, 1),
                                          (30, 50, 300, 1, 2), (7, 3, 1, 2, 3)])
def test_depthwise_backward_in_one_marching_pass(C, H, W, B, mode):
    """orcai_dw_bwd_fused (input gradient + epilogue extra + depthwise weight gradient from one pass over (du, x)) against the launches it
    replaces -- orcai_sepconv_planes_u with reversed taps and the identity pointwise factor, orcai_dw_wgrad / orcai_dw_wgrad_bn,
    orcai_bn_bwd_pointwise's own reduction pass -- and float64: mode 0 plain (ReLU on load), 1 plain with BatchNorm + ReLU on load, 2 BatchNorm
    backward sums of the output (x = the pre-normalisation tensor), 3 ReLU mask by x > 0.  Strip widths 64 / 32 / 16, several segments, planes
    whose last strip hangs over the row pitch, a single column; pads of the output stay zero; the gradient buffer is accumulated into."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(C * 3 + W + mode)
    k = 3
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    x, du = 2.0 * f(B, C, H, W), f(B, C, H, W)
    mean, var = 0.3 * f(C), (0.5 + rng.random(C)).astype(np.float32)
    gamma, beta = 1 + 0.3 * f(C), 0.4 * f(C) + 0.3
    CQ = (C + 3) // 4
    taps = f(9, C)  # Keras depthwise kernel (3, 3, C, 1) flattened
    rev = np.zeros((CQ * 4, 9), dtype=np.float32)
    rev[:C] = taps[::-1].T  # reversed taps, [channel][tap]
    rev = np.ascontiguousarray(rev.reshape(CQ, 4, 9).transpose(0, 2, 1))  # [CQ][9][4]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    xd, dud, revd = dev(_quad_planes(x, k)), dev(_quad_planes(du, k)), dev(rev)
    md, vd, gd, bd = dev(mean), dev(var), dev(gamma), dev(beta)
    eye, ones, zeros = torch.eye(C, device="cuda").contiguous(), torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    st = N.stream_ptr()
    bn = mode in (1, 2)
    relu_in = 0 if bn else 1
    # the separate launches
    plain = torch.zeros_like(dud)
    N.check(lib.orcai_sepconv_planes_u(N.ptr(dud), B, C, H, W, k, k, 0, N.ptr(revd), N.ptr(eye), N.ptr(ones), N.ptr(zeros), C, 0, 0, 0, 0, N.ptr(plain), None, st), "plain")
    dW_sep = torch.full((9, C), 0.25, device="cuda")
    if bn:
        N.check(lib.orcai_dw_wgrad_bn(N.ptr(xd), N.ptr(dud), B, C, H, W, N.ptr(md), N.ptr(vd), N.ptr(gd), N.ptr(bd), 1e-3, N.ptr(dW_sep), st), "dw_wgrad_bn")
    else:
        N.check(lib.orcai_dw_wgrad(N.ptr(xd), N.ptr(dud), B, C, H, W, k, k, relu_in, N.ptr(dW_sep), st), "dw_wgrad")
    # fused
    out = torch.zeros_like(dud)
    dW = torch.full((9, C), 0.25, device="cuda")
    shards = torch.full((8 * 16 * 32,), 7.0, dtype=torch.float64, device="cuda")
    epi = {0: 0, 1: 0, 2: 2, 3: 3}[mode]
    bnp = [N.ptr(md), N.ptr(vd), N.ptr(gd), N.ptr(bd)] if bn else [None] * 4
    N.check(lib.orcai_dw_bwd_fused(N.ptr(xd), N.ptr(dud), B, C, H, W, relu_in, N.ptr(revd), N.ptr(out), N.ptr(dW), epi, *bnp, 1e-3, 1, N.ptr(shards), st), "dw_bwd_fused")
    torch.cuda.synchronize()
    want_out = torch.where(xd > 0, plain, torch.zeros_like(plain)) if mode == 3 else plain
    scale = max(1.0, float(plain.abs().max()))
    assert float((out - want_out).abs().max()) <= 2e-6 * scale  # the same nine products, one summation order against another
    pads = out.cpu().numpy().copy()
    pads[:, :, 1 : 1 + H, :W, :] = 0
    assert float(np.abs(pads).max()) == 0.0
    # weight gradient against float64 and the separate kernel
    inv = 1.0 / np.sqrt(var.astype(np.float64) + 1e-3)
    if bn:
        sc = (gamma * (1.0 / np.sqrt(var + np.float32(1e-3))).astype(np.float32)).astype(np.float32)
        xr = np.maximum(x * sc[None, :, None, None] + (beta - mean * sc)[None, :, None, None], 0).astype(np.float64)
    else:
        xr = np.maximum(x, 0).astype(np.float64)
    xp = np.zeros((B, C, H + 2, W + 2))
    xp[:, :, 1:-1, 1:-1] = xr
    want = np.stack([[np.einsum("bchw,bchw->c", xp[:, :, dy : dy + H, dx : dx + W], du.astype(np.float64)) for dx in range(3)] for dy in range(3)]).reshape(9, C)
    n = B * H * W
    tol = max(2e-5 * np.sqrt(n) * max(1.0, np.abs(want).max() / np.sqrt(n)), 1e-4 * np.abs(want).max())
    got, sep = dW.cpu().numpy() - 0.25, dW_sep.cpu().numpy() - 0.25
    assert np.abs(got - want).max() <= tol, (np.abs(got - want).max(), tol)
    assert np.abs(sep - want).max() <= tol
    if mode == 2:  # BatchNorm backward sums of the output, gated by the ReLU
        dy = _from_quad(plain.cpu().numpy(), C, H, W, k).astype(np.float64)
        xh = (x.astype(np.float64) - mean[None, :, None, None]) * inv[None, :, None, None]
        g = np.where(xh * gamma[None, :, None, None] + beta[None, :, None, None] > 0, dy, 0.0)
        db_ref, dg_ref = g.sum(axis=(0, 2, 3)), (g * xh).sum(axis=(0, 2, 3))
        s = shards.cpu().numpy()
        tol_s = 3e-6 * np.sqrt(n) * max(1.0, np.abs(dy).max() * 3)
        assert np.abs(s[:C] - db_ref).max() <= tol_s and np.abs(s[4 * CQ : 4 * CQ + C] - dg_ref).max() <= tol_s, (np.abs(s[:C] - db_ref).max(), np.abs(s[4 * CQ : 4 * CQ + C] - dg_ref).max(), tol_s)


@pytest.mark.parametrize("H,W,B,res", [(37, 171, 2, True), (40, 171, 1, False), (16, 12, 5, True), (9, 33, 3, True), (25, 86, 2, True), (7, 1, 2, True)])
def test_depthwise_backward_rebuilds_the_entry_activation(H, W, B, res):
    """orcai_dw_bwd_fused_conv0 (block 1's first separable conv: y0 = relu(bn0(conv0(snippet))) rebuilt from the snippet's nine taps instead of read,
    the residual branch's even-pixel gradient added in the pass, bn0's backward sums over the TOTAL gradient in the epilogue) against the
    materialised path -- orcai_conv0_affine_bn for y0, orcai_dw_bwd_fused on it, the residual gradient added at the even pixels -- and float64:
    input gradient (the same nine products + one addend), depthwise weight gradient, sums dbeta | dgamma as orcai_conv0_bn_bwd_x's first pass
    defines them (gate and xhat from the recomputed pre-normalisation value)."""
    from orcai_amd import _native as N

    lib = N.lib()
    rng = np.random.default_rng(H * 5 + W)
    k = 3
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    xin = rng.random((B, H, W), dtype=np.float32)
    w0, bias = (f(9, 16) / 3).astype(np.float32), (0.1 * f(16)).astype(np.float32)
    mean, var = (0.2 * f(16)).astype(np.float32), (0.3 + rng.random(16)).astype(np.float32)
    gamma, beta = (1 + 0.3 * f(16)).astype(np.float32), (0.2 * f(16)).astype(np.float32)
    du = f(B, 16, H, W)
    taps = f(9, 16)
    rev = np.ascontiguousarray(taps[::-1].T.reshape(4, 4, 9).transpose(0, 2, 1))  # [CQ][9][4] reversed taps
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    rq = f(B, 16, Ho, Wo)
    xd, w0d, bd, md, vd, gd, btd, dud, revd = dev(xin), dev(w0), dev(bias), dev(mean), dev(var), dev(gamma), dev(beta), dev(_quad_planes(du, k)), dev(rev)
    rqd = dev(_quad_planes(rq, k)) if res else None
    ones = torch.ones(16, device="cuda")
    st = N.stream_ptr()
    # the materialised path
    y0 = torch.zeros_like(dud)
    N.check(lib.orcai_conv0_affine_bn(N.ptr(xd), H * W, B, H, W, k, N.ptr(w0d), N.ptr(ones), N.ptr(bd), N.ptr(md), N.ptr(vd), N.ptr(gd), N.ptr(btd), 1e-3, 1, N.ptr(y0), st), "conv0_affine_bn")
    dr_ref, dW_ref = torch.zeros_like(dud), torch.zeros((9, 16), device="cuda")
    N.check(lib.orcai_dw_bwd_fused(N.ptr(y0), N.ptr(dud), B, 16, H, W, 1, N.ptr(revd), N.ptr(dr_ref), N.ptr(dW_ref), 0, None, None, None, None, 0.0, 0, None, st), "dw_bwd_fused")
    dr_ref = _from_quad(dr_ref.cpu().numpy(), 16, H, W, k).astype(np.float64)
    if res:
        dr_ref[:, :, ::2, ::2] += rq.astype(np.float64)
    # one pass from the snippet
    dr, dW = torch.zeros_like(dud), torch.zeros((9, 16), device="cuda")
    shards = torch.full((8 * 16 * 32,), 7.0, dtype=torch.float64, device="cuda")
    N.check(lib.orcai_dw_bwd_fused_conv0(N.ptr(xd), H * W, N.ptr(dud), B, H, W, N.ptr(w0d), N.ptr(bd), N.ptr(revd), N.ptr(dr), N.ptr(dW), N.ptr(md), N.ptr(vd), N.ptr(gd), N.ptr(btd), 1e-3,
                                         N.ptr(shards), None if rqd is None else N.ptr(rqd), st), "dw_bwd_fused_conv0")
    torch.cuda.synchronize()
    got = _from_quad(dr.cpu().numpy(), 16, H, W, k).astype(np.float64)
    assert np.abs(got - dr_ref).max() <= 2e-6 * max(1.0, np.abs(dr_ref).max())
    pads = dr.cpu().numpy().copy()
    pads[:, :, 1 : 1 + H, :W, :] = 0
    assert float(np.abs(pads).max()) == 0.0
    assert float((dW - dW_ref).abs().max()) <= 1e-4 * max(1.0, float(dW_ref.abs().max()))  # the same y0, another summation order
    # bn0's backward sums over the total gradient (float64)
    xp = np.zeros((B, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = xin
    v0 = sum(w0[dy * 3 + dx].astype(np.float64)[None, :, None, None] * xp[:, None, dy : dy + H, dx : dx + W] for dy in range(3) for dx in range(3)) + bias[None, :, None, None]
    inv = 1.0 / np.sqrt(var.astype(np.float64) + 1e-3)
    xh = (v0 - mean[None, :, None, None]) * inv[None, :, None, None]
    g = np.where(xh * gamma[None, :, None, None] + beta[None, :, None, None] > 0, dr_ref, 0.0)
    s = shards.cpu().numpy()
    n = B * H * W
    tol = 3e-6 * np.sqrt(n) * max(1.0, np.abs(dr_ref).max() * 3) + 2e-3  # + a gate decision or two within f32 rounding of zero
    assert np.abs(s[:16] - g.sum(axis=(0, 2, 3))).max() <= tol and np.abs(s[16:32] - (g * xh).sum(axis=(0, 2, 3))).max() <= tol, (np.abs(s[:16] - g.sum(axis=(0, 2, 3))).max(), np.abs(s[16:32] - (g * xh).sum(axis=(0, 2, 3))).max(), tol)


def test_accumulator_arena_skips_only_fresh_slots():
    """orcai_scratch_arena (one clear per step for every reduction scratch): a launcher's own zero fill is skipped for a 32-KiB slot nobody has taken since the
    arena was cleared, and ONLY then -- a slot used a second time, a pointer outside the arena, a range that straddles two slots and an unregistered arena all get
    the fill.  Seen from outside through orcai_planes_sum, which clears its scratch and then accumulates into it: a stale scratch would double the sum."""
    from orcai_amd import _native as N

    lib, st = N.lib(), N.stream_ptr()
    B, C, H, W, k = 2, 8, 6, 10, 3
    WP = lib.orcai_padded_width(W, k)
    x = torch.zeros((B, 2, H + 2, WP, 4), device="cuda")
    x[:, :, 1:H + 1, :W] = torch.rand((B, 2, H, W, 4), device="cuda")
    want = x.double().sum(dim=(0, 2, 3)).reshape(-1).float()
    arena = torch.full((4 * 4096,), 123.0, dtype=torch.float64, device="cuda")  # four slots of 32 KiB, full of garbage
    out = torch.zeros(C, device="cuda")

    def run(scratch):
        N.check(lib.orcai_planes_sum(N.ptr(x), B, C, H, W, k, scratch.data_ptr(), N.ptr(out), 0, st), "planes_sum")
        torch.cuda.synchronize()
        return out.clone()

    try:
        assert lib.orcai_arena_take(arena.data_ptr(), 64) == 0  # nothing registered
        N.check(lib.orcai_scratch_arena(arena.data_ptr(), arena.numel() * 8, st), "scratch_arena")
        torch.cuda.synchronize()
        assert float(arena.abs().max()) == 0.0  # ONE launch cleared all of it
        assert torch.allclose(run(arena[:4096]), want, rtol=1e-6)  # fresh slot: fill skipped, the arena's zeros are the accumulator's start
        assert float(arena[:32].abs().max()) > 0  # ... and the sums were left in the slot
        assert torch.allclose(run(arena[:4096]), want, rtol=1e-6)  # the same slot again: NOT fresh any more -> cleared by the launcher -> still right
        arena[4096:8192] = 5.0  # a slot somebody scribbled on without the library knowing: still "fresh" to the library ...
        assert lib.orcai_arena_take(arena[4096:].data_ptr(), 32768 + 8) == 0  # ... but a range that straddles two slots is never skipped
        assert lib.orcai_arena_take(arena[2 * 4096:].data_ptr(), 64) == 1 and lib.orcai_arena_take(arena[2 * 4096:].data_ptr(), 64) == 0  # take once
        other = torch.full((4096,), 9.0, dtype=torch.float64, device="cuda")
        assert torch.allclose(run(other), want, rtol=1e-6)  # outside the arena: cleared as ever
    finally:
        lib.orcai_scratch_arena(None, 0, None)
    assert lib.orcai_arena_take(arena[3 * 4096:].data_ptr(), 64) == 0  # unregistered: nothing is skipped
    garbage = torch.full((4096,), 7.0, dtype=torch.float64, device="cuda")
    assert torch.allclose(run(garbage), want, rtol=1e-6)


def test_poison_if_nonfinite_marks_the_gradient_bucket():
    from orcai_amd import _native as N

    lib, st = N.lib(), N.stream_ptr()
    g = torch.ones(1000, device="cuda")
    stats = torch.rand(300, device="cuda")
    N.check(lib.orcai_poison_if_nonfinite(N.ptr(stats), 300, N.ptr(g), st), "poison")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all()) and float(g.sum()) == 1000.0  # finite statistics: nothing written
    for bad in (float("inf"), float("-inf"), float("nan")):
        stats[177] = bad
        g.fill_(1.0)
        N.check(lib.orcai_poison_if_nonfinite(N.ptr(stats), 300, N.ptr(g), st), "poison")
        torch.cuda.synchronize()
        assert bool(torch.isnan(g[0])) and bool(torch.isfinite(g[1:]).all())


def test_lstm_backward_split_saturates_instead_of_overflowing():
    """the split-f16 backward recurrence scales dz by the power of two taken from max|dH| of the INCOMING gradient; with large recurrent weights the
    recurrent term grows past 2^16 times that maximum within a few steps, f16(dz S) would be inf and NaN would reach dxz and the weights.  dz S is saturated
    at 2^15: dxz stays finite (the f32-MFMA kernel's behaviour); with ordinary weights the split kernel still agrees with the f32-MFMA one."""
    from orcai_amd import _native as N

    lib, st = N.lib(), N.stream_ptr()
    Bn, T, u = 16, 46, 128
    g = torch.Generator(device="cpu").manual_seed(3)
    gates = torch.rand((Bn, T, 2, 4 * u), generator=g).cuda() * 0.9 + 0.05
    cs = (torch.rand((Bn, T, 2, u), generator=g).cuda() - 0.5)
    dH = torch.randn((Bn, T, 2 * u), generator=g).cuda() * 1e-3
    before = lib.orcai_lstm_split(-1)
    try:
        for scale, explode in ((0.05, False), (1.0, True)):
            U = (torch.randn((2, u, 4 * u), generator=g) * scale).cuda()
            outs = {}
            for split in (1, 0):
                lib.orcai_lstm_split(split)
                dxz = torch.zeros((Bn, T, 2, 4 * u), device="cuda")
                N.check(lib.orcai_lstm_bwd(N.ptr(dH), N.ptr(gates), N.ptr(cs), N.ptr(U), Bn, T, u, N.ptr(dxz), st), "lstm_bwd")
                torch.cuda.synchronize()
                outs[split] = dxz
            if explode:
                assert bool(torch.isfinite(outs[0]).all()) and float(outs[0].abs().max()) > 65504.0 * float(dH.abs().max()) * 4  # the recurrence really left the f16 range of the scaled operand (and f32 holds it)
                assert bool(torch.isfinite(outs[1]).all())  # no inf / NaN from the split operand
            else:
                assert float((outs[1] - outs[0]).abs().max()) <= 5e-6 * float(outs[0].abs().max())
    finally:
        lib.orcai_lstm_split(before)


def test_batched_lstm_gradient_unpack_and_l2_values_match_the_single_launches():
    """Round 4: orcai_unpack_lstm_grads (12 (source, destination) pairs of a step in one launch) and orcai_l2_values (the five regularised kernels in one
    launch) against one orcai_unpack_lstm_grad / orcai_l2_value launch each: bit-identical gradients, the same penalty to double rounding of the sum."""
    from orcai_amd import _native as N

    lib, st = N.lib(), N.stream_ptr()
    u = 64
    g = torch.Generator(device="cpu").manual_seed(11)
    cases = []  # (src, ld, col_off, rows, W or None, l2g)
    for fin in (256, 2 * u):
        dWc = torch.randn((fin, 8 * u), generator=g).cuda()
        dbc = torch.randn((8 * u,), generator=g).cuda()
        for d in (0, 1):
            dU = torch.randn((u, 4 * u), generator=g).cuda()
            Wk = torch.randn((fin, 4 * u), generator=g).cuda()
            cases += [(dU, 4 * u, 0, u, None, 0.0), (dWc, 8 * u, d * 4 * u, fin, Wk, 2e-3), (dbc, 8 * u, d * 4 * u, 1, None, 0.0)]
    one = [torch.full((rows, 4 * u), 7.0, device="cuda") for _, _, _, rows, _, _ in cases]
    many = [torch.full((rows, 4 * u), -7.0, device="cuda") for _, _, _, rows, _, _ in cases]
    descs = []
    for (src, ld, off, rows, W, l2g), a, b in zip(cases, one, many):
        N.check(lib.orcai_unpack_lstm_grad(N.ptr(src), ld, off, rows, u, N.ptr(a), N.ptr(W) if W is not None else None, l2g, st), "unpack")
        descs.append(N.UnpackDesc(src.data_ptr(), ld, off, rows, b.data_ptr(), W.data_ptr() if W is not None else None, l2g))
    N.check(lib.orcai_unpack_lstm_grads((N.UnpackDesc * len(descs))(*descs), len(descs), u, st), "unpack_lstm_grads")
    torch.cuda.synchronize()
    for a, b in zip(one, many):
        assert torch.equal(a, b)
    assert lib.orcai_unpack_lstm_grads((N.UnpackDesc * 17)(), 17, u, st) == N.E_BADARG
    bad = N.UnpackDesc(cases[0][0].data_ptr(), 4 * u - 1, 0, u, many[0].data_ptr(), None, 0.0)  # a row shorter than the 4 u columns read from it
    assert lib.orcai_unpack_lstm_grads((N.UnpackDesc * 1)(bad), 1, u, st) == N.E_BADARG

    flat = torch.randn((300_000,), generator=g).cuda()
    slices = [(0, 32768), (40_000, 131072), (171_072, 1), (200_000, 99_999)]
    acc = torch.zeros(2, dtype=torch.float64, device="cuda")
    for o, n in slices:
        N.check(lib.orcai_l2_value(flat[o:].data_ptr(), n, 1e-3, acc[0:].data_ptr(), st), "l2_value")
    offs, cnts = (N.c_i64 * len(slices))(*[o for o, _ in slices]), (N.c_i64 * len(slices))(*[n for _, n in slices])
    N.check(lib.orcai_l2_values(N.ptr(flat), offs, cnts, len(slices), 1e-3, acc[1:].data_ptr(), st), "l2_values")
    torch.cuda.synchronize()
    want = float(np.float32(1e-3)) * sum(float((flat[o : o + n].double() ** 2).sum()) for o, n in slices)  # (lambda crosses the C ABI as a float)
    assert abs(float(acc[1]) - want) <= 1e-12 * want and abs(float(acc[0]) - float(acc[1])) <= 1e-12 * want
    assert lib.orcai_l2_values(N.ptr(flat), offs, cnts, 9, 1e-3, acc[1:].data_ptr(), st) == N.E_BADARG
