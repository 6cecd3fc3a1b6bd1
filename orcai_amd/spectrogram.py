"""Front end API.  Same names, arguments, return types and error behaviour as the reference's
``src/orcAI/spectrogram.py``; the arithmetic runs in the HIP kernels of csrc/frontend.hip.

Arrays cross this API as host numpy (the reference's contract).  ``make_spectrogram_device``
is the zero-copy variant the native predict path uses.
"""

from __future__ import annotations

from importlib.resources import files
from pathlib import Path

import numpy as np
import pandas as pd
import torch
from tqdm import tqdm

from orcai_amd import frontend as fe
from orcai_amd.auxiliary import Messenger
from orcai_amd.io import read_json, save_array, write_vector_to_json
from orcai_amd.wavio import read_wav_prefetched

DEFAULT_ORCAI_PARAMETER = files("orcai_amd.defaults").joinpath("default_orcai_parameter.json")


def load_wav(wav_file_path: Path | str, sampling_rate: int, channel: int, msgr: Messenger) -> torch.Tensor:
    """``librosa.load(path, sr=sampling_rate, mono=False)`` + channel pick (spectrogram.py:23-31),
    returning the mono signal as a float32 tensor on the GPU at `sampling_rate`."""
    wav, native_sr = read_wav_prefetched(wav_file_path)  # [channels, frames]; decoded ahead of time in table mode
    if wav.shape[0] > 1:
        msgr.warning(f"Multiple channels found, using channel {channel}")
        mono = wav[channel - 1]
    else:
        mono = wav[0]
    pcm = torch.from_numpy(np.ascontiguousarray(mono)).cuda()
    if native_sr != sampling_rate:
        from orcai_amd.resample import resample_device

        pcm = resample_device(pcm, native_sr, sampling_rate)
    return pcm


def calculate_spectrogram(wav_file_path: Path, channel: int, spectrogram_parameter: dict, msgr: Messenger = Messenger(verbosity=0)):
    """dB spectrogram [257, T] (float32), frequencies [257], times [T] (spectrogram.py:15-55)."""
    sr = spectrogram_parameter["sampling_rate"]
    n_fft = spectrogram_parameter["nfft"]
    hop = spectrogram_parameter["n_overlap"]
    pcm = load_wav(wav_file_path, sr, channel, msgr)
    db_tf = fe.get_frontend().calculate_db(pcm, n_fft, hop)
    spectrogram = db_tf.cpu().numpy().T  # [freq, time] view, as the reference returns
    frequencies = fe.fft_frequencies(sr, n_fft)
    times = fe.frames_to_time(spectrogram.shape[1], sr, hop)
    return spectrogram, frequencies, times


def preprocess_spectrogram(spectrogram: np.ndarray, frequencies: np.ndarray, spectrogram_parameter: dict) -> np.ndarray:
    """Crop to freq_range, clip to the quantiles, normalise to [0, 1], transpose (spectrogram.py:58-87)."""
    f_lo, f_hi = fe.crop_indices(frequencies, spectrogram_parameter["freq_range"])
    x = torch.from_numpy(np.ascontiguousarray(spectrogram, dtype=np.float32)).cuda()
    out = fe.get_frontend().preprocess_db(x, f_lo, f_hi, spectrogram_parameter["quantiles"])
    return out.cpu().numpy()


def make_spectrogram_device(wav_file_path: Path | str, channel: int, orcai_parameter: dict, msgr: Messenger):
    """make_spectrogram with the spectrogram left on the GPU: (f32 cuda tensor [T, K], frequencies, times)."""
    sp = orcai_parameter["spectrogram"]
    pcm = load_wav(wav_file_path, sp["sampling_rate"], channel, msgr)
    spec = fe.get_frontend().make_spectrogram(pcm, sp)
    frequencies = fe.fft_frequencies(sp["sampling_rate"], sp["nfft"])
    times = fe.frames_to_time(spec.shape[0], sp["sampling_rate"], sp["n_overlap"])
    return spec, frequencies, times


def make_spectrogram(
    wav_file_path: Path | str,
    channel: int = 1,
    orcai_parameter: (Path | str) | dict = DEFAULT_ORCAI_PARAMETER,
    verbosity: int = 2,
    msgr: Messenger | None = None,
) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Spectrogram [T, K] in [0, 1], frequencies (uncropped), times (spectrogram.py:90-147)."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Making spectrogram")
    if not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    wav_file_path = Path(wav_file_path)
    sp = orcai_parameter["spectrogram"]
    msgr.part("Calculating power spectrogram by stft")
    msgr.info(f"Loading & resampling (to {sp['sampling_rate'] / 1000:.2f} kHz) wav file: {wav_file_path.stem}")
    spec, frequencies, times = make_spectrogram_device(wav_file_path, channel, orcai_parameter, msgr)
    msgr.info(f"Duration of wav file: {times[-1]:.2f} seconds")
    msgr.info("Extracting frequency range and clipping spectrogram")
    return spec.cpu().numpy(), frequencies, times


def save_spectrogram(spectrogram: np.ndarray, frequencies: np.ndarray, times: np.ndarray, output_dir: Path | str,
                     msgr: Messenger = Messenger(verbosity=0)) -> None:
    """<output_dir>/{spectrogram.npy, frequencies.json, times.json} (spectrogram.py:150-196; raw .npy
    where the reference writes zarr)."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    save_array(spectrogram, output_dir.joinpath("spectrogram.npy"))
    write_vector_to_json(frequencies, output_dir.joinpath("frequencies.json"))
    write_vector_to_json(times, output_dir.joinpath("times.json"))
    msgr.info(f"Spectrogram saved to {output_dir}")


def _make_and_save_spectrogram(recording: pd.Series, output_dir: Path, orcai_parameter: dict, msgr: Messenger) -> None:
    """spectrogram.py:199-223."""
    wav_path = Path(recording["base_dir_recording"]).joinpath(recording["rel_recording_path"])
    spectrogram, frequencies, times = make_spectrogram(wav_path, recording["channel"], orcai_parameter, msgr=msgr)
    save_spectrogram(spectrogram, frequencies, times, Path(output_dir).joinpath(recording["recording"], "spectrogram"), msgr=msgr)


def create_spectrograms(
    recording_table_path: Path | str,
    output_dir: Path | str,
    base_dir_recording: Path | None = None,
    orcai_parameter: (Path | str) | dict = DEFAULT_ORCAI_PARAMETER,
    include_not_annotated: bool = False,
    include_no_possible_annotations: bool = False,
    overwrite: bool = False,
    verbosity: int = 2,
    msgr: Messenger | None = None,
) -> None:
    """Spectrograms for every recording of a recording table (spectrogram.py:226-321)."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Creating spectrograms")
    if not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    output_dir = Path(output_dir)
    recording_table = pd.read_csv(recording_table_path)
    if not include_not_annotated:  # spectrogram.py:279-285
        not_annotated = recording_table["base_dir_annotation"].isna()
        if len(not_annotated) > 0:
            msgr.info(f"Excluded {not_annotated.sum()} recordings because they are not annotated.")
            recording_table = recording_table[~not_annotated]
    if not include_no_possible_annotations:  # spectrogram.py:287-296
        label_calls = orcai_parameter["calls"]
        is_included = recording_table[label_calls].apply(lambda x: x.any(), axis=1)
        if sum(~is_included) > 0:
            msgr.info("Excluded recordings because they lack any possible annotations:", indent=1)
            msgr.info(str(recording_table[~is_included]["recording"].values), indent=-1)
            recording_table = recording_table[is_included]
    if not overwrite:  # spectrogram.py:298-306
        existing = recording_table["recording"].apply(lambda x: output_dir.joinpath(x, "spectrogram").exists())
        if sum(existing) > 0:
            msgr.info(f"Skipping {sum(existing)} recordings because they already have spectrograms.")
            recording_table = recording_table[~existing]
    if base_dir_recording is not None:
        recording_table["base_dir_recording"] = base_dir_recording
    msgr.part(f"Creating {len(recording_table)} spectrograms")
    for i in tqdm(recording_table.index, desc="Making spectrograms", total=len(recording_table), unit="recording", disable=verbosity < 2):
        _make_and_save_spectrogram(recording_table.loc[i], output_dir, orcai_parameter, msgr=Messenger(verbosity=0))
    msgr.success("Spectrograms created.")
