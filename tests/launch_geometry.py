"""Host-side model of the launch geometry and index expressions of the training-trunk launchers (csrc/train_trunk.hip,
csrc/model_fwd.hip, and their f16 twins in csrc/half_*.hip, which share every expression with G = 8 instead of 4 channels per
16-byte vector).  For a given layer shape each function replays, with numpy, the addresses every lane of every wave of the launch
reads or writes -- the same expressions, guards and clamps as the kernel, in units of one 16-byte pixel vector -- and returns
{operand: (lowest index touched, highest index touched + 1)}.  tests/test_launch_geometry.py compares them with the operand sizes
the trainer allocates (orcai_amd/training.TrunkTrainer._alloc) over a sweep of small, odd shapes: an out-of-bounds expression is a
CPU test failure instead of a GPU fault that depends on what sits next to the buffer.

Test infrastructure: nothing in orcai_amd/ imports this file.
"""

from __future__ import annotations

import numpy as np


def padded_width(W: int, k: int) -> int:
    return (W + k // 2 + 3) & ~3


def plane_size(H: int, W: int, k: int) -> int:
    return (H + 2 * (k // 2)) * padded_width(W, k)


class Touch:
    """Accumulates [lo, hi) of the vector indices touched per operand."""

    def __init__(self):
        self.r = {}

    def add(self, name, idx, mask=None):
        idx = np.asarray(idx, dtype=np.int64)
        if mask is not None:
            idx = idx[np.broadcast_to(np.asarray(mask, dtype=bool), idx.shape)]
        if idx.size == 0:
            return
        lo, hi = int(idx.min()), int(idx.max()) + 1
        if name in self.r:
            lo, hi = min(lo, self.r[name][0]), max(hi, self.r[name][1])
        self.r[name] = (lo, hi)


def _windows(tasks, lanes=64):
    """(task, lane) grids for kernels whose waves own 64-pixel windows."""
    return np.arange(tasks)[:, None], np.arange(lanes)[None, :]


# ------------------------------------------------------------------------------------------------ sepconv_tile_kernel / sepconv_ftile_kernel
def f32_k3_variant(Cin, Cout, H, W, out_layout, u_out, tile_mode=1):
    """Which kernel launch_sepconv_impl<3, MT> picks for an f32 k = 3 launch on R = 1 planes (model_fwd.hip): 'tile' (strip tiles),
    'ftile' (flat-range tiles) or 'window' (sepconv_kernel)."""
    MT, CQ, CQo = -(-Cout // 16), -(-Cin // 4), -(-Cout // 4)
    WP = padded_width(W, 3)
    if not tile_mode or not (out_layout == 0 or (out_layout == 2 and not u_out)):
        return "window"
    if CQo * (H + 2) * WP >= 1 << 27 or CQ * (H + 2) * WP >= 1 << 27:
        return "window"
    if MT == 2 and tile_mode == 1:
        VALt = 60 if out_layout == 2 else 62
        nstrip = -(-W // VALt)
        if CQ <= 8 and nstrip >= 2 and W * 100 >= nstrip * VALt * 85:
            return "tile"
    lo = 2 if out_layout == 2 else 1
    nchunk = (7 * (64 - 2 * lo) + 64 + 2 * WP + 63) // 64
    lds = (2 * nchunk * 256 + CQ * 64 * MT + 2 * MT * 16) * 4
    return "ftile" if nchunk <= 24 and lds <= 64 * 1024 else "window"


def sepconv_tile(B, Cin, H, W, Cout, out_layout=0, u_out=False, TR=8):
    """sepconv_tile_kernel<2, CQ, XP, RELU, 8, UOUT>: 8 image rows x 64 columns per workgroup; global row loads (LDS-DMA sources,
    clamped into the quad plane), LDS slot offsets in KiB rows, output / depthwise-output stores."""
    t = Touch()
    XP = out_layout == 2
    lo = 2 if XP else 1
    VAL = 64 - 2 * lo
    WP = padded_width(W, 3)
    plane = (H + 2) * WP
    CQ, CQo = -(-Cin // 4), -(-Cout // 4)
    nstrip = -(-W // VAL)
    ngroups = -(-H // TR)
    lane = np.arange(64)[None, None, :]
    rg = np.arange(ngroups)[:, None, None]
    strip = np.arange(nstrip)[None, :, None]
    r0, c0 = rg * TR, strip * VAL
    lds_rows = []
    for wave in range(TR):
        for j in [wave] + ([TR + wave] if wave < 2 else []):  # tile rows this wave fetches
            i = np.clip((r0 + j) * WP + c0 - lo + lane, 0, plane - 1)
            for b in (0, B - 1):
                for cq in (0, CQ - 1):
                    t.add("in", (b * CQ + cq) * plane + i)
            lds_rows.append(j)
        for dy in range(3):  # rows read back: wave .. wave + 2 of a slot of TR + 2 rows
            lds_rows.append(wave + dy)
        row = r0 + wave
        x = c0 - lo + lane
        live = (lane >= lo) & (lane < 64 - lo) & (x < W) & (row < H)
        for b in (0, B - 1):
            if u_out:
                for cq in (0, CQ - 1):
                    t.add("u_out", (b * CQ + cq) * plane + (1 + row) * WP + x, live)
            for oq in (0, CQo - 1):
                if XP:
                    Wx = (W + 1) // 2
                    WPx = (Wx + 3) & ~3
                    t.add("out", ((b * CQo + oq) * H + row) * WPx + (x >> 1), live & (x % 2 == 0))
                else:
                    t.add("out", (b * CQo + oq) * plane + (1 + row) * WP + x, live)
    t.add("lds_rows", np.array(lds_rows))
    return t.r, {"grid": (nstrip * ngroups, B), "lds_rows": TR + 2}


def sepconv_ftile(B, Cin, H, W, Cout, out_layout=0, u_out=False, NWV=8):
    """sepconv_ftile_kernel<MT, XP, RELU, UOUT, 8>: 8 consecutive flat windows per workgroup; the LDS-DMA chunks of the contiguous row
    range (clamped sources, chunk slots), the three LDS rows every wave reads (in pixels of a slot), output / depthwise-output stores."""
    t = Touch()
    XP = out_layout == 2
    lo = 2 if XP else 1
    VAL = 64 - 2 * lo
    WP = padded_width(W, 3)
    plane = (H + 2) * WP
    CQ, CQo = -(-Cin // 4), -(-Cout // 4)
    tasks = (H * WP + VAL - 1) // VAL
    ngroups = -(-tasks // NWV)
    nchunk = ((NWV - 1) * VAL + 64 + 2 * WP + 63) // 64
    lane = np.arange(64)[None, :]
    bx = np.arange(ngroups)[:, None]
    s0 = bx * NWV * VAL - lo
    for c in range(nchunk):
        i = np.clip(s0 + 64 * c + lane, 0, plane - 1)
        for b in (0, B - 1):
            for cq in (0, CQ - 1):
                t.add("in", (b * CQ + cq) * plane + i)
    for wave in range(NWV):
        for dy in range(3):
            t.add("lds_pixels", wave * VAL + lane + dy * WP)
        task = bx * NWV + wave
        q = WP + task * VAL - lo + lane
        row = q // WP
        x = q - row * WP
        live = (task < tasks) & (lane >= lo) & (lane < 64 - lo) & (x < W) & (row < 1 + H)
        for b in (0, B - 1):
            if u_out:
                for cq in (0, CQ - 1):
                    t.add("u_out", (b * CQ + cq) * plane + q, live)
            for oq in (0, CQo - 1):
                if XP:
                    Wx = (W + 1) // 2
                    WPx = (Wx + 3) & ~3
                    t.add("out", ((b * CQo + oq) * H + (row - 1)) * WPx + (x >> 1), live & (x % 2 == 0))
                else:
                    t.add("out", (b * CQo + oq) * plane + q, live)
    return t.r, {"grid": (ngroups, B), "lds_pixels": nchunk * 64, "chunks_per_wave": -(-nchunk // NWV)}


def sepconv_f32(B, Cin, H, W, Cout, out_layout=0, u_out=False, tile_mode=1):
    """The f32 k = 3 launch as the launcher would run it (R = 1 planes)."""
    v = f32_k3_variant(Cin, Cout, H, W, out_layout, u_out, tile_mode)
    if v == "tile":
        r, g = sepconv_tile(B, Cin, H, W, Cout, out_layout, u_out)
    elif v == "ftile":
        r, g = sepconv_ftile(B, Cin, H, W, Cout, out_layout, u_out)
    else:
        r, g = sepconv(B, Cin, H, W, 3, 3, Cout, out_layout, u_out=u_out, G=4)
    g["variant"] = v
    return r, g


def conv0_sep_tile(B, H, W, Cout, TRW=10):
    """conv0_sep_tile_kernel<2, TRW>: output-plane stores, the (2i, 2j) subsample stores, LDS rows (the snippet reads go through a
    range-checked buffer resource)."""
    t = Touch()
    TR, VAL = TRW - 2, 62
    WP = padded_width(W, 3)
    plane = (H + 2) * WP
    CQo = -(-Cout // 4)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    nstrip, ngroups = -(-W // VAL), -(-H // TR)
    lane = np.arange(64)[None, None, :]
    r0 = (np.arange(ngroups) * TR)[:, None, None]
    c0 = (np.arange(nstrip) * VAL)[None, :, None]
    x = c0 - 1 + lane
    for wave in range(TRW):
        e = r0 - 1 + wave
        sub = (wave >= 1) & (wave <= TR) & (lane >= 1) & (lane < 63) & (x < W) & (e < H) & (((x | e) & 1) == 0)
        for b in (0, B - 1):
            for cq in (0, 3):
                t.add("prev_sub", (b * 4 + cq) * Ho * Wo + (e >> 1) * Wo + (x >> 1), sub)
        if wave < TR:
            y = r0 + wave
            live = (lane >= 1) & (lane < 63) & (x < W) & (y < H)
            for b in (0, B - 1):
                for oq in (0, CQo - 1):
                    t.add("out", (b * CQo + oq) * plane + (1 + y) * WP + x, live)
            t.add("lds_rows", np.array([wave, wave + 2]))
    return t.r, {"grid": (nstrip * ngroups, B), "lds_rows": TRW}


# ------------------------------------------------------------------------------------------------ sepconv_kernel<KS, MT> / sepconv_h_kernel
def sepconv(B, Cin, H, W, kplanes, ktap, Cout, out_layout=0, H2=0, W2=0, u_out=False, G=4):
    """model_fwd.hip sepconv_kernel (launch_sepconv_impl): lo / VAL / tasks, clamped row indices, the three plane-shaped output layouts
    and the depthwise-output store.  Layout 1 (Keras Reshape) is reported in ELEMENTS of the feature tensor."""
    t = Touch()
    R, WP = kplanes // 2, padded_width(W, kplanes)
    plane = (H + 2 * R) * WP
    CG, CGo = -(-Cin // G), -(-Cout // G)
    lo = ((ktap // 2 + 1) & ~1) if out_layout == 2 else ktap // 2
    VAL = 64 - 2 * lo
    tasks = (H * WP + VAL - 1) // VAL
    task, lane = _windows(tasks)
    qbase = R * WP + task * VAL - lo
    q = qbase + lane
    for b in (0, B - 1):
        for dy in range(ktap):
            i = np.clip(q + (dy - ktap // 2) * WP, 0, plane - 1)
            for cg in (0, CG - 1):
                t.add("in", (b * CG + cg) * plane + i)
        row = q // WP
        x = q - row * WP
        live = (lane >= lo) & (lane < 64 - lo) & (x < W) & (row < R + H)
        if u_out:
            for cg in (0, CG - 1):
                t.add("u_out", (b * CG + cg) * plane + q, live)
        if out_layout == 0:
            for og in (0, CGo - 1):
                t.add("out", (b * CGo + og) * plane + q, live)
        elif out_layout == 2:
            Wx = (W + 1) // 2
            WPx = (Wx + 3) & ~3
            for og in (0, CGo - 1):
                t.add("out", ((b * CGo + og) * H + (row - R)) * WPx + (x >> 1), live & (x % 2 == 0))
        elif out_layout == 3:
            WP2 = padded_width(W2, kplanes)
            plane2 = (H2 + 2 * R) * WP2
            for og in (0, CGo - 1):
                t.add("out", (b * CGo + og) * plane2 + (2 * (row - R) + R) * WP2 + 2 * x, live)
        else:
            t.add("out", ((b * H + (row - R)) * W * Cout + x * Cout + (Cout - 1)), live)
            t.add("out", ((b * H + (row - R)) * W * Cout + x * Cout), live)
    return t.r, {"tasks": tasks, "grid": ((tasks + 3) // 4, B)}


# ------------------------------------------------------------------------------------------------ bn_bwd_pw_kernel<MT>
def bn_bwd_pointwise(B, C, Cin, H, W, k, G=4):
    """train_trunk.hip bn_bwd_pw_kernel: 64-pixel windows over the interior rows (no halo), loads clamped to the plane, stores live-masked."""
    t = Touch()
    R, WP = k // 2, padded_width(W, k)
    plane = (H + 2 * R) * WP
    CG, CGi = -(-C // G), -(-Cin // G)
    tasks = (H * WP + 63) // 64
    task, lane = _windows(tasks)
    q = R * WP + task * 64 + lane
    row = q // WP
    live = ((q - row * WP) < W) & (row < R + H)
    qc = np.minimum(q, plane - 1)
    for b in (0, B - 1):
        for cg in (0, CG - 1):
            t.add("dy", (b * CG + cg) * plane + qc)
            t.add("v", (b * CG + cg) * plane + qc)
            t.add("dv", (b * CG + cg) * plane + qc, live)
        for og in (0, CGi - 1):
            t.add("du", (b * CGi + og) * plane + q, live)
    return t.r, {"tasks": tasks, "grid": ((tasks + 3) // 4, B)}


# ------------------------------------------------------------------------------------------------ pool_bwd_kernel
def pool_bwd(B, C, H, W, k, G=4, PB_ROWS=8):
    """train_trunk.hip pool_bwd_kernel: a thread owns pooled column j of one (snippet, vector group) and marches PB_ROWS pooled rows."""
    t = Touch()
    R, WP = k // 2, padded_width(W, k)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    WPo = padded_width(Wo, k)
    pad_top, pad_left = max((Ho - 1) * 2 + 3 - H, 0) // 2, max((Wo - 1) * 2 + 2 - W, 0) // 2
    CG = -(-C // G)
    nchunk = (Ho + PB_ROWS - 1) // PB_ROWS
    per_bq = nchunk * Wo
    grid_x = (per_bq + 255) // 256
    idx = np.arange(grid_x * 256)
    in_range = idx < per_bq
    j = np.where(in_range, idx % Wo, 0)
    chunk = np.where(in_range, idx // Wo, 0)
    pin, pout = (H + 2 * R) * WP, (Ho + 2 * R) * WPo
    x0 = 2 * j - pad_left
    x1 = x0 + 1
    cx0, cx1 = (x0 >= 0) & (x0 < W), x1 < W
    i0 = chunk * PB_ROWS
    i1 = np.where(in_range, np.minimum(i0 + PB_ROWS, Ho), i0)
    istart = np.where(i0 > 0, i0 - 1, 0)

    def ld(bq, y, x, cx):
        ok = cx & (y >= 0) & (y < H)
        t.add("v", bq * pin + (y + R) * WP + x, ok)

    for bq in (0, B * CG - 1):
        ld(bq, 2 * istart - pad_top, x0, cx0)
        ld(bq, 2 * istart - pad_top, x1, cx1)
        for step in range(PB_ROWS + 1):
            i = istart + step
            run = i < i1
            r0 = 2 * i - pad_top
            for dr in (1, 2):
                ld(bq, np.where(run, r0 + dr, 0), x0, cx0 & run)
                ld(bq, np.where(run, r0 + dr, 0), x1, cx1 & run)
            t.add("dout", bq * pout + (i + R) * WPo + j, run)
            own = run & (i >= i0)
            t.add("dy", bq * pin + (r0 + R) * WP + x0, own & (r0 >= 0) & cx0)
            t.add("dy", bq * pin + (r0 + R) * WP + x1, own & (r0 >= 0) & cx1)
            t.add("dy", bq * pin + (r0 + 1 + R) * WP + x0, own & (r0 + 1 < H) & cx0)
            t.add("dy", bq * pin + (r0 + 1 + R) * WP + x1, own & (r0 + 1 < H) & cx1)
        rl = 2 * i1 - pad_top
        last = in_range & (i1 == Ho) & (rl < H)
        t.add("dy", bq * pin + (rl + R) * WP + x0, last & cx0)
        t.add("dy", bq * pin + (rl + R) * WP + x1, last & cx1)
    return t.r, {"grid": (grid_x, B * CG), "scratch_doubles": 2 * G * CG}


# ------------------------------------------------------------------------------------------------ outer_reduce_kernel
def outer_reduce(B, Ca, Cb, H, W, k, a_stride2=False, Ha=0, Wa=0, workspace_floats=512 * 64 * 64, G=4):
    """train_trunk.hip outer_reduce_kernel<PP> + add_partials: PP-pixel chunks over whole planes (pads are zero; f32: PP = 256 up to 32
    channels per operand, 128 beyond, 768 workgroups then; f16: 256), the stride-2 sampling of A for the residual conv, and the
    per-workgroup partial products in the workspace."""
    t = Touch()
    R, WP = k // 2, padded_width(W, k)
    plane = (H + 2 * R) * WP
    CGa, CGb = -(-Ca // G), -(-Cb // G)
    WPa = padded_width(Wa, k) if a_stride2 else 0
    plane_a = (Ha + 2 * R) * WPa if a_stride2 else plane
    tiles = (Ca + 15) // 16 + (Cb + 15) // 16
    PP = 256 if (G != 4 or tiles <= 4) else 128
    cpp = (plane + PP - 1) // PP
    nchunks = B * cpp
    grid = min(nchunks, 512 if PP == 256 else 768)
    if grid * Ca * Cb > workspace_floats:
        grid = workspace_floats // (Ca * Cb)
    tid = (np.arange(256) % PP)[None, :]  # PP = 128: the two halves of the workgroup stage one operand each, the same pixels
    # every chunk a block can ask for, including the prefetch one stride past the end (must be masked, not dereferenced)
    ch = np.arange(nchunks + grid)[:, None]
    b = ch // cpp
    p = (ch - b * cpp) * PP + tid
    pin = (ch < nchunks) & (p < plane)
    row = np.where(pin, p, 0) // WP
    x, i = p - row * WP, row - R
    if a_stride2:
        ain = pin & (i >= 0) & (i < H) & (x < W)
        pa = np.where(ain, (2 * i + R) * WPa + 2 * x, 0)
    else:
        ain, pa = pin, p
    for cg in (0, CGa - 1):
        t.add("A", (b * CGa + cg) * plane_a + pa, ain)
    for cg in (0, CGb - 1):
        t.add("B", (b * CGb + cg) * plane + p, pin)
    t.add("workspace", np.array([0, grid * Ca * Cb - 1]))
    lds = tiles * 16 * (PP + 2) * 4 if G == 4 else 256 * (tiles * 16 + 16) * 2  # f32: [channel][pixel] image; f16: [pixel][channel] image
    return t.r, {"grid": grid, "lds_bytes": lds}


# ------------------------------------------------------------------------------------------------ dw_wgrad_kernel<KS>
def dw_wgrad(B, C, H, W, kplanes, ktap, G=4):
    t = Touch()
    RP, WP = kplanes // 2, padded_width(W, kplanes)
    R = ktap // 2
    VAL = 64 - 2 * R
    plane = (H + 2 * RP) * WP
    CG = -(-C // G)
    tasks = (H * WP + VAL - 1) // VAL
    tpw = max((tasks + 7) // 8, 8)
    grid_x = ((tasks + tpw - 1) // tpw + 3) // 4
    wv = np.arange(grid_x * 4)[:, None, None]
    step = np.arange(tpw)[None, :, None]
    lane = np.arange(64)[None, None, :]
    task = wv * tpw + step
    run = task < tasks
    q = RP * WP + task * VAL - R + lane
    contributes = (lane >= R) & (lane < 64 - R)
    for bq in (0, B * CG - 1):
        t.add("du", bq * plane + q, run & contributes & (q < plane))
        for dy in range(ktap):
            t.add("x", bq * plane + np.clip(q + (dy - R) * WP, 0, plane - 1), run)
    assert int((run.any(axis=2)).sum()) == tasks  # every window is visited exactly once
    return t.r, {"grid": (grid_x, CG, B), "dW_floats": ktap * ktap * C}


# ------------------------------------------------------------------------------------------------ pool_res_add_kernel (training: planes + BN)
def pool_res_add(B, C, Cp, H, W, k, G=4):
    t = Touch()
    R, WP = k // 2, padded_width(W, k)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    WPo = padded_width(Wo, k)
    pad_top, pad_left = max((Ho - 1) * 2 + 3 - H, 0) // 2, max((Wo - 1) * 2 + 2 - W, 0) // 2
    CG, CGp = -(-C // G), -(-Cp // G)
    plane, plane_o = (H + 2 * R) * WP, (Ho + 2 * R) * WPo
    tasks = (Ho * WPo + 63) // 64
    task, lane = _windows(tasks)
    q = R * WPo + task * 64 + lane
    prow = q // WPo
    pj, pi = q - prow * WPo, prow - R
    valid = (pj < Wo) & (pi < Ho)
    src = np.where(valid, (2 * pi + R) * WP + 2 * pj, 0)
    for b in (0, B - 1):
        for cg in (0, CGp - 1):
            t.add("prev", (b * CGp + cg) * plane + src)
        for og in (0, CG - 1):
            t.add("out", (b * CG + og) * plane_o + q, valid)
            for dy in range(3):
                for dx in range(2):
                    y, x = 2 * pi - pad_top + dy, 2 * pj - pad_left + dx
                    ok = valid & (y >= 0) & (y < H) & (x >= 0) & (x < W)
                    t.add("s", (b * CG + og) * plane + (y + R) * WP + x, ok)
    return t.r, {"tasks": tasks}


# ------------------------------------------------------------------------------------------------ whole-plane / interior kernels
def planes_sums(B, C, H, W, k, G=4):
    """planes_sums_kernel / bn_planes_bwd_sums_kernel: every pixel vector of every plane, grid.x capped at 128 blocks."""
    plane = plane_size(H, W, k)
    CG = -(-C // G)
    gx = min((B * plane + 255) // 256, 128)
    p0 = np.arange(gx * 256)
    reach = p0[p0 < plane]
    t = Touch()
    t.add("x", np.array([0, (B * CG - 1) * plane + int(reach.max())]))
    covered = np.zeros(plane, dtype=bool)
    for start in reach:
        covered[start::gx * 256] = True
    assert covered.all()  # the strided loop visits every pixel
    return t.r, {"grid": (gx, CG), "scratch_doubles": 2 * G * CG}


def interior_kernel(B, C, H, W, k, G=4):
    """bn_planes_apply_kernel / feat_to_planes_kernel: one thread per interior pixel of one (snippet, vector group)."""
    R, WP = k // 2, padded_width(W, k)
    plane = (H + 2 * R) * WP
    CG = -(-C // G)
    pix = np.arange(H * W)
    yy, xx = pix // W, pix % W
    t = Touch()
    for bq in (0, B * CG - 1):
        t.add("planes", bq * plane + (yy + R) * WP + xx)
    return t.r, {"grid_y": B * CG}


def conv0_bn_wgrad(B, H, W, k, G=4):
    R, WP = k // 2, padded_width(W, k)
    plane = (H + 2 * R) * WP
    pix = np.arange(H * W)
    y, x = pix // W, pix % W
    t = Touch()
    CG = 16 // G
    for bq in (0, B * CG - 1):
        t.add("dy", bq * plane + (y + R) * WP + x)
    src = []
    for dy in range(k):
        for dx in range(k):
            yy, xx = y + dy - R, x + dx - R
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            src.append((yy * W + xx)[ok])
    t.add("snippet", np.concatenate(src))
    return t.r, {}
