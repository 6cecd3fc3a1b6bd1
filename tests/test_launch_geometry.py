"""Every address the training-trunk launchers compute stays inside the operand the trainer allocates (CPU only).

A GPU fault from one stray index is intermittent -- it needs the neighbouring pages to be unmapped -- so it is hunted here, on the
host, by replaying the kernels' index expressions (tests/launch_geometry.py) for a whole training step over a sweep of small, odd
layer shapes: the shape of the round-1 abort (input 48 x 21, filters 12 / 30 / 40, batch 2: three blocks, width chain 21 -> 11 -> 6 ->
3, a 3-quad first block), the other shapes of tests/test_train_full_gpu.py, tap sizes 5 and 7, widths around the 64-pixel window
and the 256-pixel chunk, single rows / columns, and orcai-V1 itself; for the f32 quad planes (G = 4) and the f16 octet planes (G = 8).
"""

import itertools

import numpy as np
import pytest

import launch_geometry as LG

SHAPES = [
    ((48, 21), (12, 30, 40), 3, 2),  # the abort of gpurun_out/train_cfg1.log (round 1)
    ((32, 12), (10, 20), 3, 3),
    ((32, 16), (10, 20), 5, 2),
    ((64, 61), (30, 40, 50, 60), 3, 4),
    ((16, 1), (4, 9), 3, 1),  # a single column
    ((8, 63), (5,), 7, 2),
    ((8, 64), (64,), 3, 1),
    ((24, 255), (17, 33), 5, 2),
    ((40, 7), (3, 6, 9), 7, 5),
    ((736, 171), (30, 40, 50, 60), 3, 2),  # orcai-V1 (batch 2 keeps the sweep fast: indices scale with (B - 1))
]


def stage_shapes(hw, filters):
    h, w = hw
    out = [(h, w, 16)]
    for f in filters:
        h, w = -(-h // 2), -(-w // 2)
        out.append((h, w, f))
    return out


def check(touched, sizes, what):
    for name, (lo, hi) in touched.items():
        assert lo >= 0, (what, name, "negative index", lo)
        assert hi <= sizes[name], (what, name, f"touches [{lo}, {hi}) of an operand of {sizes[name]} vectors")


@pytest.mark.parametrize("G", [4, 8])
@pytest.mark.parametrize("hw,filters,k,B", SHAPES)
def test_training_step_addresses_stay_inside_their_operands(hw, filters, k, B, G):
    shapes = stage_shapes(hw, filters)
    planes = lambda c, h, w: B * -(-c // G) * LG.plane_size(h, w, k)  # noqa: E731  (TrunkTrainer._planes)
    H, W = hw
    # entry conv + bn0
    r, _ = LG.interior_kernel(B, 16, H, W, k, G)
    check(r, {"planes": planes(16, H, W)}, "conv0 / bn0 apply")
    r, g = LG.planes_sums(B, 16, H, W, k, G)
    check(r, {"x": planes(16, H, W)}, "bn0 stats")
    assert g["scratch_doubles"] <= 128
    r, _ = LG.conv0_bn_wgrad(B, H, W, k, G)
    check(r, {"dy": planes(16, H, W), "snippet": H * W}, "conv0_bn_wgrad")
    c = 16
    for i, f in enumerate(filters, start=1):
        h, w, _ = shapes[i - 1]
        ho, wo, _ = shapes[i]
        for cin, tag in ((c, "a"), (f, "b")):
            r, _ = LG.sepconv(B, cin, h, w, k, k, f, 0, u_out=True, G=G)
            check(r, {"in": planes(cin, h, w), "out": planes(f, h, w), "u_out": planes(cin, h, w)}, f"b{i}/sep_{tag} forward")
            if G == 4 and k == 3:  # the f32 launcher hands k = 3 plane launches to the LDS-shared-row kernels
                r, _ = LG.sepconv_f32(B, cin, h, w, f, 0, True)
                r.pop("lds_rows", None), r.pop("lds_pixels", None)
                check(r, {"in": planes(cin, h, w), "out": planes(f, h, w), "u_out": planes(cin, h, w)}, f"b{i}/sep_{tag} forward (tile)")
                r, _ = LG.sepconv_f32(B, cin, h, w, cin, 0, False)
                r.pop("lds_rows", None), r.pop("lds_pixels", None)
                check(r, {"in": planes(cin, h, w), "out": planes(cin, h, w)}, f"b{i}/sep_{tag} input gradient (tile)")
            r, g = LG.planes_sums(B, f, h, w, k, G)
            check(r, {"x": planes(f, h, w)}, f"b{i}/bn_{tag} stats")
            assert g["scratch_doubles"] <= 128 and g["grid"][0] >= 1
            r, g = LG.interior_kernel(B, f, h, w, k, G)
            check(r, {"planes": planes(f, h, w)}, f"b{i}/bn_{tag} apply")
            assert g["grid_y"] <= 65535
        r, _ = LG.pool_res_add(B, f, c, h, w, k, G)
        check(r, {"prev": planes(c, h, w), "s": planes(f, h, w), "out": planes(f, ho, wo)}, f"b{i}/pool_res_add")
        # backward of the block
        r, g = LG.outer_reduce(B, c, f, ho, wo, k, True, h, w, G=G)
        check(r, {"A": planes(c, h, w), "B": planes(f, ho, wo), "workspace": 512 * 64 * 64}, f"b{i}/res weight gradient")
        assert g["grid"] >= 1 and g["lds_bytes"] <= 160 * 1024
        r, g = LG.pool_bwd(B, f, h, w, k, G)
        check(r, {"v": planes(f, h, w), "dout": planes(f, ho, wo), "dy": planes(f, h, w)}, f"b{i}/pool_bwd")
        assert g["scratch_doubles"] <= 128 and g["grid"][1] <= 65535
        for cin, tag in ((f, "b"), (c, "a")):
            r, _ = LG.bn_bwd_pointwise(B, f, cin, h, w, k, G)
            check(r, {"dy": planes(f, h, w), "v": planes(f, h, w), "dv": planes(f, h, w), "du": planes(cin, h, w)}, f"b{i}/bn_{tag} backward + pointwise^T")
            r, _ = LG.outer_reduce(B, cin, f, h, w, k, G=G)
            check(r, {"A": planes(cin, h, w), "B": planes(f, h, w), "workspace": 512 * 64 * 64}, f"b{i}/sep_{tag} pointwise weight gradient")
            r, g = LG.dw_wgrad(B, cin, h, w, k, k, G)
            check(r, {"x": planes(cin, h, w), "du": planes(cin, h, w)}, f"b{i}/sep_{tag} depthwise weight gradient")
            assert g["grid"][2] <= 65535
            r, _ = LG.sepconv(B, cin, h, w, k, k, cin, 0, G=G)
            check(r, {"in": planes(cin, h, w), "out": planes(cin, h, w)}, f"b{i}/sep_{tag} input gradient")
        r, _ = LG.sepconv(B, f, ho, wo, k, 1, c, 3, H2=h, W2=w, G=G)
        check(r, {"in": planes(f, ho, wo), "out": planes(c, h, w)}, f"b{i}/res input gradient (scatter-add)")
        c = f
    h, w, _ = shapes[-1]
    r, _ = LG.sepconv(B, c, h, w, k, k, 36, 1, u_out=True, G=G)
    check(r, {"in": planes(c, h, w), "out": B * h * w * 36, "u_out": planes(c, h, w)}, "sep_f forward")
    r, _ = LG.interior_kernel(B, 36, h, w, k, G)
    check(r, {"planes": planes(36, h, w)}, "feat_to_planes")
    r, _ = LG.outer_reduce(B, c, 36, h, w, k, G=G)
    check(r, {"A": planes(c, h, w), "B": planes(36, h, w), "workspace": 512 * 64 * 64}, "sep_f pointwise weight gradient")
    r, _ = LG.sepconv(B, 36, h, w, k, 1, c, 0, G=G)
    check(r, {"in": planes(36, h, w), "out": planes(c, h, w)}, "sep_f pointwise^T")
    r, _ = LG.dw_wgrad(B, c, h, w, k, k, G)
    check(r, {"x": planes(c, h, w), "du": planes(c, h, w)}, "sep_f depthwise weight gradient")


def test_tile_kernels_stay_inside_their_operands_and_lds():
    """The LDS-shared-row kernels of the f32 k = 3 launches (strip tiles, flat-range tiles) and the strip-tile entry kernel over a sweep
    of widths around the strip / window boundaries, one-row planes, ragged channel counts: clamped LDS-DMA sources inside the input
    planes, stores inside the output / depthwise-output planes and only on interior pixels, LDS rows / pixels inside a slot, at
    most three chunks per wave; and the launcher's choice covers all three kernels."""
    seen = set()
    for (H, W), (Cin, Cout), layout, u_out in itertools.product(
            [(1, 3), (5, 61), (7, 62), (9, 63), (8, 106), (13, 118), (16, 171), (33, 124), (3, 250), (2, 500), (4, 520)],
            [(16, 30), (30, 30), (13, 17), (40, 40), (16, 16), (60, 60), (7, 64)], (0, 2), (False, True)):
        if u_out and layout == 2:
            continue
        B = 2
        r, g = LG.sepconv_f32(B, Cin, H, W, Cout, layout, u_out)
        seen.add(g["variant"])
        WP = LG.padded_width(W, 3)
        plane = (H + 2) * WP
        Wx = (W + 1) // 2
        sizes = {"in": B * -(-Cin // 4) * plane, "u_out": B * -(-Cin // 4) * plane,
                 "out": B * -(-Cout // 4) * (plane if layout == 0 else H * ((Wx + 3) & ~3))}
        if g["variant"] == "tile":
            sizes["lds_rows"] = g["lds_rows"]
        if g["variant"] == "ftile":
            sizes["lds_pixels"] = g["lds_pixels"]
            assert g["chunks_per_wave"] <= 3
        check(r, sizes, (g["variant"], H, W, Cin, Cout, layout, u_out))
        if layout == 0:  # stores only on interior pixels: the first / last plane row and the padding columns are never written
            lo_out, hi_out = r["out"]
            assert lo_out >= WP and hi_out <= B * -(-Cout // 4) * plane - WP
    assert seen == {"tile", "ftile", "window"}
    for (H, W), Cout, trw in itertools.product([(1, 106), (9, 118), (17, 171), (8, 124), (30, 250)], (17, 30, 32), (10, 16)):
        r, g = LG.conv0_sep_tile(2, H, W, Cout, trw)
        WP = LG.padded_width(W, 3)
        check(r, {"out": 2 * -(-Cout // 4) * (H + 2) * WP, "prev_sub": 2 * 4 * ((H + 1) // 2) * ((W + 1) // 2), "lds_rows": g["lds_rows"]}, ("conv0_sep_tile", H, W, Cout, trw))


def test_x_pooled_layout_and_window_cover():
    """Inference-side layouts of the same kernel: the x-pooled output (windows start on even pixels) and full coverage of the plane."""
    for (H, W), k, C in itertools.product([(5, 1), (7, 21), (12, 62), (9, 171)], (3, 5, 7), (10, 36)):
        Wx = (W + 1) // 2
        r, _ = LG.sepconv(2, C, H, W, k, k, C, 2)
        assert r["out"][1] <= 2 * -(-C // 4) * H * ((Wx + 3) & ~3)
        R, WP = k // 2, LG.padded_width(W, k)
        lo = k // 2
        VAL = 64 - 2 * lo
        tasks = (H * WP + VAL - 1) // VAL
        covered = np.zeros((H + 2 * R) * WP + 64, dtype=bool)
        for t in range(tasks):
            covered[R * WP + t * VAL : R * WP + (t + 1) * VAL] = True
        assert covered[R * WP : (R + H) * WP].all()  # every interior pixel is some window's output


def test_clamps_and_guards_are_load_bearing():
    """Negative controls: without the kernels' clamp / guard the same shapes DO leave their operands, so the sweep above is not
    vacuous: the first window's top tap row starts before the plane, the last bn_bwd_pw window ends past the interior rows, and
    outer_reduce's prefetch of the chunk one grid-stride past the end addresses a snippet that does not exist."""
    (H, W), k, B = (48, 21), 3, 2
    WP = LG.padded_width(W, k)
    plane = LG.plane_size(H, W, k)
    assert (1 * WP + 0 * 62 - 1) + (0 - 1) * WP < 0  # sepconv: lane 0 of window 0, tap row -1, unclamped
    H2, W2 = 12, 6  # block 3 of that model
    WP2, plane2 = LG.padded_width(W2, k), LG.plane_size(H2, W2, k)
    tasks = (H2 * WP2 + 63) // 64
    assert 1 * WP2 + (tasks - 1) * 64 + 63 >= plane2  # bn_bwd_pw: the last window runs past the plane: loads clamped, stores masked
    assert plane > 0
    cpp = (LG.plane_size(6, 3, k) + 255) >> 8
    assert (B * cpp + 0) // cpp >= B  # outer_reduce: chunk index nchunks belongs to snippet B


def test_kernels_with_dynamic_lds_opt_in_beyond_64_kib():
    """Source lint for the round-1 abort's most probable cause: a kernel that takes `extern __shared__` memory sized by its launcher
    (outer_reduce: 16.5 KB per 16-channel tile row -- over the 64 KiB default as soon as an operand has more than 32 channels, which
    of the round-1 test shapes only the aborting one had) must either opt in with hipFuncSetAttribute(MaxDynamicSharedMemorySize) or
    bound its request at 64 KiB in the launcher."""
    import re
    from pathlib import Path

    seen = 0
    for src in sorted((Path(__file__).resolve().parent.parent / "orcai_amd" / "csrc").glob("*.hip")):
        text = src.read_text()
        for m in re.finditer(r"extern\s+__shared__", text):
            head = text[: m.start()]
            kernel = re.findall(r"void\s+(\w+)\s*\(", head)[-1]  # the nearest function definition above the declaration
            seen += 1
            opted_in = re.search(r"hipFuncSetAttribute\(\(const void\*\)\(?" + kernel + r"\b[^;]*hipFuncAttributeMaxDynamicSharedMemorySize", text)
            # "lds > 64 * 1024) return ..." (refuse) or "... && lds <= 64 * 1024 ..." (fall back to a kernel without dynamic LDS)
            bounded = re.search(r"lds\s*>\s*64\s*\*\s*1024\)\s*return\b|lds\s*<=\s*64\s*\*\s*1024", text)
            assert opted_in or bounded, f"{src.name}: {kernel} uses dynamic LDS without opting in beyond 64 KiB or bounding its size"
    assert seen >= 3
