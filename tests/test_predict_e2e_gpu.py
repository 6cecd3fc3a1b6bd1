"""End-to-end GPU parity of `orcai predict` (config 1 of BASELINE.json: a mono PCM16 wav, 22.05 kHz -> resampled to
48 kHz, and the 48 kHz parity twin) against the CPU oracle pipeline, plus the resampler on its own.

Tolerances: resampler |delta| <= 2e-6 (f32 taps, f32 accumulate vs f64 oracle accumulate); aggregated probabilities
|delta| <= 2e-4 end to end (front end 2e-4 on [0,1] inputs propagates through a calibrated network to ~1e-5);
label intervals identical unless an averaged probability lies within 1e-3 of the 0.25 threshold.
"""

import gzip
import io
import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
CALLS = ["BR", "BUZZ", "HERDING", "PHS", "SS", "TAILSLAP", "WHISTLE"]


def test_resampler_vs_oracle():
    from oracle.resample_ref import resample_ref
    from orcai_amd.resample import output_length, resample_device
    from orcai_amd.synthetic import pcm16_to_float, synth_recording

    for sr_in, sr_out, secs in [(22050, 48000, 3.0), (44100, 48000, 1.0), (96000, 48000, 1.0), (48000, 48000, 0.5), (8000, 48000, 0.7)]:
        x = pcm16_to_float(synth_recording(secs, sr_in, seed=4))
        y = resample_device(torch.from_numpy(x).cuda(), sr_in, sr_out).cpu().numpy()
        assert len(y) == output_length(len(x), sr_in, sr_out)
        if sr_in == sr_out:
            assert np.array_equal(y, x)
            continue
        ref = resample_ref(x, sr_in, sr_out)
        assert np.abs(y - ref).max() <= 2e-6, (sr_in, np.abs(y - ref).max())
    # band-limited sine: the resampled signal is the same sine at the new rate
    t = np.arange(22050) / 22050
    x = (0.5 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    y = resample_device(torch.from_numpy(x).cuda(), 22050, 48000).cpu().numpy()
    tt = np.arange(len(y)) / 48000
    m = slice(1000, len(y) - 1000)
    assert np.abs(y[m] - 0.5 * np.sin(2 * np.pi * 3000 * tt[m])).max() <= 1e-5


def _model_dir(tmp_path, seed=21):
    from oracle import model_ref as M

    p = M.calibrated_params(seed=seed, calib_batch=1)
    d = tmp_path / "model"
    d.mkdir()
    for n in ("orcai_parameter.json", "model_shape.json"):
        (d / n).write_text((ROOT / "orcai_amd" / "models" / "orcai-V1" / n).read_text())
    np.savez(d / "orcai-v1.weights.npz", **p)
    return d, p


def _oracle_pipeline(y48, p):
    from oracle import frontend_ref as F
    from oracle import model_ref as M
    from oracle import postprocess_ref as P

    param = json.loads((ROOT / "orcai_amd" / "models" / "orcai-V1" / "orcai_parameter.json").read_text())
    spec, _, times = F.make_spectrogram_ref(y48, param)
    pred = M.forward_ref(p, P.slice_snippets(spec, 736))
    agg, cnt = P.aggregate_predictions_ref(pred, spec.shape[0], 736, 4, 7)
    s, e, n = P.compute_binary_predictions_ref(agg, cnt, CALLS)
    labels = P.compute_labels_ref(s, e, n, 16, "*")
    return agg, cnt, P.labels_to_tsv_ref(labels, times[1] - times[0]), times[1] - times[0]


@pytest.mark.parametrize("sr,seconds", [(48000, 14.0), (22050, 60.0)])  # the second one is BASELINE configs[0]: one 60 s 22.05 kHz recording, 29 snippets
def test_predict_wav_end_to_end(tmp_path, sr, seconds):
    from oracle.resample_ref import resample_ref
    from orcai_amd.predict import predict
    from orcai_amd.synthetic import pcm16_to_float, synth_recording
    from orcai_amd.wavio import write_wav_pcm16

    model_dir, p = _model_dir(tmp_path)
    pcm = synth_recording(seconds, sr, seed=9)
    wav = tmp_path / "rec.wav"
    write_wav_pcm16(wav, np.stack([pcm, pcm[::-1]]), sr)  # 2 channels: channel 1 is used
    out = tmp_path / "rec_pred.txt"
    predict(wav, channel=1, model_dir=model_dir, output_path=out, save_probabilities=True, verbosity=0)
    y = pcm16_to_float(pcm)
    y48 = y if sr == 48000 else resample_ref(y, sr, 48000)
    agg_ref, cnt_ref, tsv_ref, delta_t = _oracle_pipeline(y48, p)
    probs = pd.read_csv(io.BytesIO(gzip.decompress((tmp_path / "rec_pred_probabilities.csv.gz").read_bytes())), index_col="time")
    assert list(probs.columns) == CALLS and probs.shape == agg_ref.shape
    assert np.abs(probs.to_numpy() - agg_ref).max() <= 2e-4
    # Label intervals: identical to the oracle's except where an averaged probability lies within tolerance of the
    # threshold -- those (step, label) cells are COUNTED, set to the oracle's side of the threshold in the GPU probabilities, and the
    # intervals extracted from the result must then equal the oracle's text exactly.  Never skipped; few cells may be ambiguous.
    thr = 0.5 / cnt_ref.max()
    ambiguous = np.abs(agg_ref - thr) <= 1e-3
    n_amb = int(ambiguous.sum())
    print(f"e2e sr={sr}: {n_amb} of {agg_ref.size} (step, label) cells within 1e-3 of the threshold {thr}")
    assert n_amb <= max(3, agg_ref.size // 200), n_amb
    from oracle import postprocess_ref as P

    gpu_agg = probs.to_numpy().copy()
    gpu_agg[ambiguous] = agg_ref[ambiguous]
    s_, e_, n_ = P.compute_binary_predictions_ref(gpu_agg, cnt_ref, CALLS)
    assert P.labels_to_tsv_ref(P.compute_labels_ref(s_, e_, n_, 16, "*"), delta_t) == tsv_ref
    if n_amb == 0:
        assert out.read_text() == tsv_ref  # the file the product wrote, byte for byte
    else:  # the product's file differs from the oracle's at most by runs touching an ambiguous cell
        got = [ln.split("\t") for ln in out.read_text().splitlines()[1:]]
        want = [ln.split("\t") for ln in tsv_ref.splitlines()[1:]]
        amb_labels = {CALLS[j] + "*" for j in np.nonzero(ambiguous.any(axis=0))[0]}
        assert [g for g in got if g[2] not in amb_labels] == [w for w in want if w[2] not in amb_labels]
    with pytest.raises(FileExistsError):
        predict(wav, channel=1, model_dir=model_dir, output_path=out, verbosity=0)
    predict(wav, channel=1, model_dir=model_dir, output_path=out, overwrite=True, verbosity=0)


def test_predict_table_mode_and_cli(tmp_path):
    from click.testing import CliRunner

    from orcai_amd.cli import cli
    from orcai_amd.synthetic import synth_recording
    from orcai_amd.wavio import write_wav_pcm16

    model_dir, _ = _model_dir(tmp_path)
    rec = tmp_path / "recs"
    rec.mkdir()
    for name in ("a", "b"):
        write_wav_pcm16(rec / f"{name}.wav", synth_recording(9.0, 48000, seed=ord(name)), 48000)
    # a failing recording BETWEEN two good ones: table mode keeps one recording in flight on the GPU while the previous one's labels are
    # written, and an error in either half must neither lose nor mix up its neighbours
    table = pd.DataFrame({"recording": ["a", "missing", "b"], "base_dir_recording": [str(rec)] * 3, "rel_recording_path": ["a.wav", "nope.wav", "b.wav"], "channel": [1, 1, 1]})
    table.to_csv(tmp_path / "table.csv", index=False)
    outdir = tmp_path / "out"
    outdir.mkdir()
    res = CliRunner().invoke(cli, ["predict", str(tmp_path / "table.csv"), "-md", str(model_dir), "-o", str(outdir), "-v", "1"])
    assert res.exit_code == 0, res.output
    assert (outdir / "a_model_predicted.txt").exists() and (outdir / "b_model_predicted.txt").exists()
    assert not (outdir / "missing_model_predicted.txt").exists()  # per-recording errors are logged, not raised
    first = (outdir / "a_model_predicted.txt").read_text().splitlines()[0]
    assert first == "start\tstop\tlabel"
    # the pipelined table run writes, per recording, exactly the file the one-recording path writes
    from orcai_amd.predict import predict

    for name in ("a", "b"):
        single = tmp_path / f"single_{name}.txt"
        predict(rec / f"{name}.wav", model_dir=model_dir, output_path=single, verbosity=0)
        assert single.read_text() == (outdir / f"{name}_model_predicted.txt").read_text(), name
    res = CliRunner().invoke(cli, ["predict", str(tmp_path / "model" / "model_shape.json")])
    assert res.exit_code != 0  # "Recording file must be a wav or csv file"


def test_create_spectrograms(tmp_path):
    from oracle import frontend_ref as F
    from orcai_amd.io import read_json
    from orcai_amd.spectrogram import create_spectrograms
    from orcai_amd.synthetic import pcm16_to_float, synth_recording
    from orcai_amd.wavio import write_wav_pcm16

    pcm = synth_recording(6.0, 48000, seed=2)
    write_wav_pcm16(tmp_path / "r1.wav", pcm, 48000)
    row = {"recording": "r1", "base_dir_recording": str(tmp_path), "rel_recording_path": "r1.wav", "channel": 1, "base_dir_annotation": "x"}
    row.update({c: True for c in CALLS})
    pd.DataFrame([row]).to_csv(tmp_path / "t.csv", index=False)
    out = tmp_path / "spec"
    create_spectrograms(tmp_path / "t.csv", out, verbosity=0)
    spec = np.load(out / "r1" / "spectrogram" / "spectrogram.npy")
    ref, freqs, times = F.make_spectrogram_ref(pcm16_to_float(pcm), {"spectrogram": read_json(ROOT / "orcai_amd" / "defaults" / "default_orcai_parameter.json")["spectrogram"]})
    assert spec.shape == ref.shape and np.abs(spec - ref).max() <= 2e-4
    t = read_json(out / "r1" / "spectrogram" / "times.json")
    assert t["length"] == len(times) and t["max"] == times[-1]


def _shard_worker(rank, world, port, q):
    import os
    import sys

    sys.path.insert(0, str(ROOT))
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch
    import torch.distributed as dist

    from orcai_amd.architectures import ResNetLSTM

    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = ResNetLSTM((64, 20, 1), 3, [10, 20], 3, 0.0, 64, seed=5)
    model.prepare()
    g = torch.Generator().manual_seed(11)
    spec = torch.rand((64 + 32 * 8 + 5, 20), generator=g).cuda()  # 9 snippets: blocks of 5 and 4
    whole = model.predict_spectrogram(spec)
    sharded = model.predict_spectrogram(spec, shard=True)
    q.put((rank, tuple(whole.shape), bool(torch.equal(whole, sharded))))
    dist.barrier()
    dist.destroy_process_group()


def test_predict_sharded_by_snippet_ranges_two_ranks():
    """one recording, snippets split into contiguous per-rank blocks, one all_gather; bit-identical to one rank."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] == (9, 16, 3) and r[2] for r in results), results
