"""GPU parity of the HIP front end (through the C ABI) against the CPU oracle and the golden
vectors made by the reference's own preprocess_spectrogram.

Tolerance (stated per BASELINE north_star "within a stated fp32 tolerance"):
  * order statistics / clip+normalise of a given dB array: BIT-EXACT;
  * STFT -> dB -> normalised spectrogram from PCM: the reference computes the FFT in float64 and
    rounds to complex64, the kernel computes in float32: |delta| <= 2e-4 on the [0,1] output for
    every element, <= 2e-5 for 99.9 % of them.
"""

import json
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SPEC_PARAM = {"sampling_rate": 48000, "nfft": 512, "n_overlap": 256, "freq_range": [0, 16000], "quantiles": [0.01, 0.999], "duration": 4}
FREQS = np.fft.rfftfreq(512, 1 / 48000)


@pytest.fixture(scope="module")
def fe():
    from orcai_amd.frontend import get_frontend

    return get_frontend()


def _select_case(fe, x, ranks):
    xs = np.sort(x, axis=None)
    d = torch.from_numpy(x).cuda()
    lo, hi = fe.select(d, ranks[0], ranks[1])
    assert np.float32(lo) == xs[ranks[0]] or (np.isnan(lo) and np.isnan(xs[ranks[0]])), (lo, xs[ranks[0]])
    assert np.float32(hi) == xs[ranks[1]], (hi, xs[ranks[1]])


@pytest.mark.parametrize("kind", ["normal_db", "ties", "constant", "near_zero", "wide", "tiny"])
def test_quantile_select_exact(fe, kind):
    rng = np.random.default_rng(5)
    n = 171 * 3001
    if kind == "normal_db":
        x = (-40 + 12 * rng.standard_normal(n)).astype(np.float32)
    elif kind == "ties":
        x = np.round(-50 + 25 * rng.standard_normal(n)).clip(-80, 0).astype(np.float32)
    elif kind == "constant":
        x = np.full(n, -100.0, dtype=np.float32)
    elif kind == "near_zero":  # everything inside the single level-1 bucket around 0 -> two 16-bit rounds
        x = (1e-3 * rng.standard_normal(n)).astype(np.float32)
        x[:1000] = 0.0
        x[1000:2000] = -0.0
    elif kind == "wide":
        x = (rng.standard_normal(n) * np.exp(8 * rng.standard_normal(n))).astype(np.float32)
    else:
        n = 37
        x = rng.standard_normal(n).astype(np.float32)
    for ranks in [(int(0.01 * (n - 1)), int(0.999 * (n - 1))), (0, n - 1), (n // 2, n // 2)]:
        _select_case(fe, x, ranks)


@pytest.mark.parametrize("name", ["smooth_T300", "ties_T200"])
def test_preprocess_matches_reference_golden_bit_exact(golden_dir, name):
    from orcai_amd.spectrogram import preprocess_spectrogram

    g = np.load(golden_dir / f"preprocess_{name}.npz")
    out = preprocess_spectrogram(g["db"], FREQS, SPEC_PARAM)
    assert out.dtype == np.float32 and out.shape == g["out"].shape
    assert np.array_equal(out, g["out"])


def test_preprocess_constant_is_nan_like_reference(golden_dir):
    from orcai_amd.spectrogram import preprocess_spectrogram

    g = np.load(golden_dir / "preprocess_constant_T64.npz")
    out = preprocess_spectrogram(g["db"], FREQS, SPEC_PARAM)
    assert np.array_equal(np.isnan(out), np.isnan(g["out"]))


def test_preprocess_large_golden(golden_dir):
    """T = 11251 (the 60 s recording of BASELINE configs[0]): 1.92 M values through the exact order statistics, against the CRC of
    what the reference's own preprocess_spectrogram returned.  The hashed input is integer arithmetic only, so it is the same
    array on every numpy; the Generator-made one is checked as well and a stream mismatch FAILS (it is never skipped)."""
    import sys

    sys.path.insert(0, str(golden_dir))
    from portable_inputs import hashed_db

    from orcai_amd.spectrogram import preprocess_spectrogram

    metas = json.loads((golden_dir / "preprocess_large.json").read_text())
    meta = metas["hashed_T11251"]
    x = hashed_db(meta["seed"], meta["T"])
    assert (zlib.crc32(x.tobytes()) & 0xFFFFFFFF) == meta["input_crc32"]
    out = preprocess_spectrogram(x, FREQS, SPEC_PARAM)
    assert (zlib.crc32(np.ascontiguousarray(out).tobytes()) & 0xFFFFFFFF) == meta["output_crc32"]
    meta = metas["smooth_T11251"]
    rng = np.random.default_rng(meta["seed"])
    x = np.clip(-40.0 + 12.0 * rng.standard_normal((257, meta["T"])), -80.0, 0.0).astype(np.float32)
    x[7, 3] = 0.0
    assert (zlib.crc32(x.tobytes()) & 0xFFFFFFFF) == meta["input_crc32"], "numpy Generator stream differs from the fixture: regenerate tests/golden"
    out = preprocess_spectrogram(x, FREQS, SPEC_PARAM)
    assert (zlib.crc32(np.ascontiguousarray(out).tobytes()) & 0xFFFFFFFF) == meta["output_crc32"]


def _pcm(seconds, seed=20250620):
    from orcai_amd.synthetic import pcm16_to_float, synth_recording

    return pcm16_to_float(synth_recording(seconds, 48000, seed))


@pytest.mark.parametrize("seconds", [60.0, 7.3, 0.05])
def test_make_spectrogram_vs_oracle(fe, seconds):
    from oracle import frontend_ref as F

    y = _pcm(seconds)
    ref, _, _ = F.make_spectrogram_ref(y, {"spectrogram": SPEC_PARAM})
    out = fe.make_spectrogram(torch.from_numpy(y).cuda(), SPEC_PARAM).cpu().numpy()
    assert out.shape == ref.shape == (1 + len(y) // 256, 171)
    d = np.abs(out - ref)
    assert d.max() <= 2e-4, d.max()
    assert np.quantile(d, 0.999) <= 2e-5, np.quantile(d, 0.999)
    assert out.min() == 0.0 and out.max() == 1.0


@pytest.mark.parametrize("nfft,hop,seconds", [(256, 128, 7.3), (1024, 256, 7.3), (1024, 512, 20.0), (2048, 300, 3.1), (4096, 1024, 7.3), (64, 32, 0.5), (128, 77, 1.0)])
def test_other_transform_sizes_vs_oracle(fe, nfft, hop, seconds):
    """nfft is a field of the parameter file (spectrogram.py:34-39); sizes other than 512 run the plain radix-2 kernel and a separate level-1
    histogram pass: same pipeline, same bars as the 512-point path, any hop (odd ones too), crop taken from the frequency vector."""
    from oracle import frontend_ref as F

    sp = dict(SPEC_PARAM, nfft=nfft, n_overlap=hop)
    y = _pcm(seconds, seed=nfft)
    ref, _, _ = F.make_spectrogram_ref(y, {"spectrogram": sp})
    out = fe.make_spectrogram(torch.from_numpy(y).cuda(), sp).cpu().numpy()
    k = int(np.argwhere(np.fft.rfftfreq(nfft, 1 / 48000) >= 16000)[0][0])
    assert out.shape == ref.shape == (1 + len(y) // hop, k)
    d = np.abs(out - ref)
    assert d.max() <= 2e-4, d.max()
    assert np.quantile(d, 0.999) <= 2e-5, np.quantile(d, 0.999)
    assert out.min() == 0.0 and out.max() == 1.0
    # and the dB matrix of all 1 + nfft/2 bins
    refdb, _, _ = F.calculate_spectrogram_ref(y, sp)
    db = fe.calculate_db(torch.from_numpy(y).cuda(), nfft, hop).cpu().numpy().T
    assert db.shape == refdb.shape and db.max() == 0.0 and db.min() >= -80.0
    live = refdb > -79.0
    assert np.abs(db - refdb)[live].max() <= 5e-3


@pytest.mark.parametrize("nfft,hop,seconds", [(500, 250, 3.0), (1000, 300, 2.0), (16, 8, 0.2), (675, 128, 1.5), (3000, 1024, 4.0), (4095, 2000, 3.0)])
def test_transform_sizes_that_are_not_powers_of_two_vs_oracle(fe, nfft, hop, seconds):
    """Round 4: any nfft from 2 to 4096 (spectrogram.py:34-39 hands the parameter file's value to librosa.stft): sizes the radix-2 kernels do not take
    run a direct float64 transform.  Same pipeline, same bars as test_other_transform_sizes_vs_oracle (odd sizes: 1 + nfft // 2 bins, centre pad nfft // 2)."""
    from oracle import frontend_ref as F

    sp = dict(SPEC_PARAM, nfft=nfft, n_overlap=hop)
    y = _pcm(seconds, seed=nfft)
    ref, _, _ = F.make_spectrogram_ref(y, {"spectrogram": sp})
    out = fe.make_spectrogram(torch.from_numpy(y).cuda(), sp).cpu().numpy()
    assert out.shape == ref.shape and out.shape[0] == 1 + (len(y) - (nfft & 1)) // hop
    d = np.abs(out - ref)
    assert d.max() <= 2e-4, d.max()
    assert np.quantile(d, 0.999) <= 2e-5, np.quantile(d, 0.999)
    assert out.min() == 0.0 and out.max() == 1.0
    refdb, _, _ = F.calculate_spectrogram_ref(y, sp)
    db = fe.calculate_db(torch.from_numpy(y).cuda(), nfft, hop).cpu().numpy().T
    assert db.shape == refdb.shape == (1 + nfft // 2, 1 + (len(y) - (nfft & 1)) // hop) and db.max() == 0.0 and db.min() >= -80.0
    live = refdb > -79.0
    assert np.abs(db - refdb)[live].max() <= 5e-3


def test_unsupported_transform_sizes_say_so(fe):
    for nfft in (1, 8192):
        with pytest.raises(NotImplementedError, match="from 2 to 4096"):
            fe.make_spectrogram(torch.zeros(48000, device="cuda"), dict(SPEC_PARAM, nfft=nfft))


def test_calculate_db_vs_oracle(fe):
    from oracle import frontend_ref as F

    y = _pcm(20.0, seed=3)
    ref, _, _ = F.calculate_spectrogram_ref(y, SPEC_PARAM)  # [257, T]
    out = fe.calculate_db(torch.from_numpy(y).cuda(), 512, 256).cpu().numpy().T
    assert out.shape == ref.shape
    assert out.max() == 0.0 and out.min() >= -80.0
    live = ref > -79.0
    assert np.abs(out - ref)[live].max() <= 5e-3  # dB; f32 FFT vs f64 FFT rounded to complex64


def test_silence_and_edge_inputs(fe):
    # all-zero input: every value is the amin floor -> p_lo == p_hi -> NaN, like the reference (unguarded 0/0)
    z = torch.zeros(48000, device="cuda")
    out = fe.make_spectrogram(z, SPEC_PARAM).cpu().numpy()
    assert out.shape == (188, 171) and np.isnan(out).all()
    # shorter than one hop: a single frame
    y = _pcm(0.004)
    out = fe.make_spectrogram(torch.from_numpy(y).cuda(), SPEC_PARAM).cpu().numpy()
    assert out.shape == (1, 171)


def test_odd_hop_and_other_crop(fe):
    from oracle import frontend_ref as F

    p = dict(SPEC_PARAM, n_overlap=255, freq_range=[0, 24000])
    y = _pcm(3.0, seed=9)
    ref, _, _ = F.make_spectrogram_ref(y, {"spectrogram": p})
    out = fe.make_spectrogram(torch.from_numpy(y).cuda(), p).cpu().numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-4
