"""Inference driver.  Same names, arguments, return types, output files and error behaviour as the
reference's ``src/orcAI/predict.py``; the arithmetic (front end, snippet forward passes, overlap average)
runs in HIP kernels.  Label extraction (thresholding an [S,7] matrix, run detection, sorting, TSV) is the
reference's thin host logic restated with numpy/pandas.
"""

from __future__ import annotations

from importlib.resources import files
from pathlib import Path

import numpy as np
import pandas as pd
import torch
from tqdm import tqdm

from orcai_amd import _native as N
from orcai_amd.auxiliary import Messenger, find_consecutive_ones
from orcai_amd.io import load_orcai_model, read_json
from orcai_amd.spectrogram import make_spectrogram, make_spectrogram_device

DEFAULT_MODEL_DIR = files("orcai_amd.models").joinpath("orcai-V1")
DEFAULT_CALL_DURATION_LIMITS = files("orcai_amd.defaults").joinpath("default_call_duration_limits.json")


def _check_duration(calls: pd.Series, call_duration_limits: dict, delta_t: float, label_suffix: str = "*") -> str:
    """predict.py:14-66."""
    label = calls["label"].replace(f"{label_suffix}", "")
    if label in call_duration_limits:
        min_duration, max_duration = call_duration_limits[label]
    elif "default" in call_duration_limits:
        min_duration, max_duration = call_duration_limits["default"]
    else:
        min_duration, max_duration = 0, np.inf
    if min_duration is None:
        min_duration = 0
    if max_duration is None:
        max_duration = np.inf
    if calls["duration"] * delta_t < min_duration:
        return "too short"
    if calls["duration"] * delta_t > max_duration:
        return "too long"
    return "keep"


def filter_predictions(predicted_labels: pd.DataFrame, delta_t: float, call_duration_limits: (Path | str) | dict = DEFAULT_CALL_DURATION_LIMITS,
                       label_suffix: str = "*", verbosity: int = 2, msgr: Messenger | None = None) -> pd.DataFrame:
    """Drop predicted calls whose duration is outside the per-label limits (predict.py:69-159)."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Filtering predictions")
    msgr.part("Filtering predictions")
    predicted_labels["duration"] = predicted_labels["stop"] - predicted_labels["start"]
    if not isinstance(call_duration_limits, dict):
        call_duration_limits = read_json(call_duration_limits)
    msgr.part("Filtering calls based on duration")
    if len(predicted_labels) == 0:
        predicted_labels["duration_ok"] = []
        return predicted_labels
    predicted_labels["duration_ok"] = predicted_labels.apply(lambda x: _check_duration(x, call_duration_limits, delta_t, label_suffix), axis=1)
    n_long = int((predicted_labels["duration_ok"] == "too long").sum())
    n_short = int((predicted_labels["duration_ok"] == "too short").sum())
    msgr.info(f"Discarding {n_long + n_short} calls based on duration (too short: {n_short}, too long: {n_long})")
    kept = predicted_labels[predicted_labels["duration_ok"] == "keep"]
    msgr.success("Filtering predictions finished.")
    return kept


def filter_predictions_file(predicted_labels: Path | str, output_file: Path | str = "default", overwrite: bool = False,
                            call_duration_limits: (Path | str) | dict = DEFAULT_CALL_DURATION_LIMITS, label_suffix: str = "*", verbosity: int = 2,
                            msgr: Messenger | None = None):
    """predict.py:162-232."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Filtering predictions file")
    if output_file == "default":
        output_file = Path(predicted_labels).with_name(Path(predicted_labels).stem + "_filtered.txt")
    else:
        output_file = Path(output_file)
    msgr.info(f"Output file: {output_file}")
    if output_file.exists() and not overwrite:
        raise FileExistsError(f"Annotation file already exists: {output_file}")
    table = pd.read_csv(predicted_labels, sep="\t", encoding="utf-8")
    kept = filter_predictions(table, delta_t=1, call_duration_limits=call_duration_limits, label_suffix=label_suffix, verbosity=verbosity, msgr=msgr)
    save_predictions(kept, output_file, delta_t=1, msgr=msgr)


class PendingAggregate:
    """Averaged probabilities of one recording on their way from the GPU to pinned host memory.  `result()` waits for the copy (an
    event on the launch stream, not a device-wide synchronisation) and returns the two host arrays."""

    def __init__(self, agg_host: torch.Tensor, cnt_host: torch.Tensor, event, keep=()):
        self._agg, self._cnt, self._event, self._keep = agg_host, cnt_host, event, keep

    def result(self) -> tuple[np.ndarray, np.ndarray]:
        if self._event is not None:
            self._event.synchronize()
            self._event, self._keep = None, ()
        return self._agg.numpy(), self._cnt.numpy()


def aggregate_predictions_device(predictions: torch.Tensor, n_frames: int, snippet_length: int, n_filters: int, wait: bool = True):
    """Overlap-average of per-snippet predictions on the GPU (predict.py:276-293).
    predictions: f32 cuda [n, P, L].  Returns host (f64 [n_frames // 2**n_filters, L], f64 [...]); with wait=False a PendingAggregate
    whose copy to (pinned) host memory is still in flight, so that the caller can queue the next recording's GPU work first."""
    lib = N.lib()
    tpo = 2**n_filters
    shift = snippet_length // 2
    P = snippet_length // tpo
    S = n_frames // tpo
    n, L = int(predictions.shape[0]), int(predictions.shape[2])
    agg = torch.empty((S, L), dtype=torch.float64, device=predictions.device)
    cnt = torch.empty((S,), dtype=torch.float64, device=predictions.device)
    if S > 0:
        pred = predictions.contiguous()
        N.check(lib.orcai_overlap_average(pred.data_ptr() if n > 0 else agg.data_ptr(), n, P, L, shift // tpo, S, N.ptr(agg), N.ptr(cnt), N.stream_ptr()),
                "orcai_overlap_average")
    if wait:
        return agg.cpu().numpy(), cnt.cpu().numpy()
    agg_host = torch.empty((S, L), dtype=torch.float64, pin_memory=True)
    cnt_host = torch.empty((S,), dtype=torch.float64, pin_memory=True)
    agg_host.copy_(agg, non_blocking=True)
    cnt_host.copy_(cnt, non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    return PendingAggregate(agg_host, cnt_host, event, keep=(agg, cnt, predictions))


def compute_aggregated_predictions(recording_path: Path, spectrogram, model, orcai_parameter: dict, shape: dict,
                                   msgr: Messenger = Messenger(verbosity=0), progressbar: tqdm = None, wait: bool = True):
    """Slice into 50 %-overlapping snippets, predict, overlap-average (predict.py:235-295).

    ``model`` may be any object with ``.predict(ndarray[n,L,F,1], verbose=int) -> ndarray[n,P,labels]`` (the
    reference's duck-typed boundary).  A native ResNetLSTM is run without materialising the snippets; a
    ``spectrogram`` that already lives on the GPU (torch tensor) is used in place.  wait=False returns a PendingAggregate (the
    device-to-host copy of the averaged probabilities still in flight) instead of the two host arrays.
    """
    snippet_length = shape["input_shape"][0]
    shift = snippet_length // 2
    n_filters = len(orcai_parameter["model"]["filters"])
    n_frames = int(spectrogram.shape[0])
    num_snippets = (n_frames - snippet_length) // shift + 1
    if num_snippets <= 0:
        # predict.py:244-268 is unguarded here: the empty snippet array reaches model.predict and Keras raises on its shape.
        # Same outcome (an exception, caught per recording in table mode, predict.py:752-755), with a usable message.
        raise ValueError(f"recording too short: {n_frames} spectrogram frames, one snippet needs {snippet_length}")
    msgr.info(f"slicing into {num_snippets} snippets for prediction")
    msgr.info("Prediction of snippets")
    native = hasattr(model, "predict_spectrogram")
    if native:
        spec_dev = spectrogram if isinstance(spectrogram, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(spectrogram, dtype=np.float32)).cuda()
        # inside a process group the snippets of this recording are split into contiguous per-rank blocks and all-gathered
        predictions = model.predict_spectrogram(spec_dev.contiguous(), shard=bool(orcai_parameter.get("shard_snippets", False)))
    else:
        spec_host = spectrogram.cpu().numpy() if isinstance(spectrogram, torch.Tensor) else spectrogram
        snippets = np.array([spec_host[i * shift : i * shift + snippet_length] for i in range(num_snippets)])
        snippets = snippets[..., np.newaxis]
        predictions = torch.from_numpy(np.asarray(model.predict(snippets, verbose=0 if msgr.verbosity < 2 else 1), dtype=np.float32)).cuda()
    msgr.info("Aggregating predictions")
    if progressbar:
        progressbar.set_description(f"{recording_path.stem} - Aggregating predictions")
        progressbar.refresh()
    if predictions.shape[0] == 0:
        predictions = predictions.reshape(0, snippet_length // 2**n_filters, shape["num_labels"])
    return aggregate_predictions_device(predictions, n_frames, snippet_length, n_filters, wait=wait)


def compute_binary_predictions(aggregated_predictions: np.ndarray, overlap_count: np.ndarray, calls: list[str], threshold: float = 0.5):
    """predict.py:298-317."""
    adjusted_threshold = threshold / np.max(overlap_count)
    binary_prediction = (aggregated_predictions > adjusted_threshold).astype(int)
    row_starts, row_stops, label_names = [], [], []
    for i, label_name in enumerate(calls):
        if binary_prediction[:, i].sum() > 0:  # == the reference's builtin sum(), vectorised
            row_start, row_stop = find_consecutive_ones(binary_prediction[:, i])
            row_starts += list(row_start)
            row_stops += list(row_stop)
            label_names += [label_name] * len(row_start)
    return row_starts, row_stops, label_names


def compute_labels(row_starts, row_stops, label_names, time_steps_per_output_step: int, label_suffix: str | None) -> pd.DataFrame:
    """predict.py:320-340."""
    if (label_suffix is not None) & (label_suffix != ""):
        label_names = [label + label_suffix for label in label_names]
    return (
        pd.DataFrame({"start": np.asarray(row_starts) * time_steps_per_output_step, "stop": np.asarray(row_stops) * time_steps_per_output_step,
                      "label": label_names})
        .sort_values(by=["start", "stop", "label"])
        .reset_index(drop=True)
    )


def _convert_times_to_seconds(predicted_labels: pd.DataFrame, delta_t: float) -> pd.DataFrame:
    """predict.py:343-364."""
    predicted_labels = predicted_labels.copy()
    predicted_labels["start"] = predicted_labels["start"] * delta_t
    predicted_labels["stop"] = predicted_labels["stop"] * delta_t
    return predicted_labels


def predict_wav_launch(recording_path: Path | str, channel: int, model, orcai_parameter: dict, shape: dict, msgr: Messenger = Messenger(verbosity=0),
                       progressbar: tqdm = None) -> dict:
    """First half of predict_wav: spectrogram, snippets through the model and the overlap average are QUEUED on the GPU, the averaged
    probabilities are on their way to pinned host memory.  The returned state goes to predict_wav_finish; in table mode the next
    recording is launched in between, so the host half of one recording runs beside the GPU half of the next."""
    recording_path = Path(recording_path)
    if progressbar:
        progressbar.set_description(f"{recording_path.stem}: Generating spectrogram")
        progressbar.refresh()
    native = hasattr(model, "predict_spectrogram")
    if native:
        spectrogram, _, times = make_spectrogram_device(recording_path, channel, orcai_parameter, msgr)
    else:
        spectrogram, _, times = make_spectrogram(recording_path, channel, orcai_parameter, msgr=msgr)
    delta_t = times[1] - times[0]
    if spectrogram.shape[1] != shape["input_shape"][1]:
        raise ValueError(f"Spectrogram shape ({spectrogram.shape[1]}) for {recording_path.stem} not equal to input shape ({shape['input_shape'][1]})")
    msgr.part(f"Prediction of annotations for wav_file: {recording_path.stem}")
    if progressbar:
        progressbar.set_description(f"{recording_path.stem} - Predicting annotations")
        progressbar.refresh()
    pending = compute_aggregated_predictions(recording_path=recording_path, spectrogram=spectrogram, model=model, orcai_parameter=orcai_parameter, shape=shape,
                                             msgr=msgr, progressbar=progressbar, wait=not native)
    return {"pending": pending, "delta_t": delta_t, "orcai_parameter": orcai_parameter}


def predict_wav_finish(state: dict, label_suffix: str = "*", msgr: Messenger = Messenger(verbosity=0)):
    """Second half of predict_wav (host): threshold, runs of ones, label table (predict.py:298-340)."""
    pending, orcai_parameter = state["pending"], state["orcai_parameter"]
    aggregated_predictions, overlap_count = pending.result() if isinstance(pending, PendingAggregate) else pending
    row_starts, row_stops, label_names = compute_binary_predictions(aggregated_predictions=aggregated_predictions, overlap_count=overlap_count,
                                                                    calls=orcai_parameter["calls"], threshold=0.5)
    msgr.info("converting binary predictions into start and stop frames")
    time_steps_per_output_step = 2 ** len(orcai_parameter["model"]["filters"])
    predicted_labels = compute_labels(row_starts, row_stops, label_names, time_steps_per_output_step=time_steps_per_output_step, label_suffix=label_suffix)
    msgr.info(f"found {len(predicted_labels)} acoustic signals")
    msgr.success("Prediction finished.")
    return predicted_labels, aggregated_predictions, state["delta_t"]


def predict_wav(recording_path: Path | str, channel: int, model, orcai_parameter: dict, shape: dict, label_suffix: str = "*",
                msgr: Messenger = Messenger(verbosity=0), progressbar: tqdm = None):
    """(predicted_labels DataFrame, aggregated_predictions ndarray, delta_t) for one wav file (predict.py:367-471)."""
    state = predict_wav_launch(recording_path, channel, model, orcai_parameter, shape, msgr=msgr, progressbar=progressbar)
    return predict_wav_finish(state, label_suffix=label_suffix, msgr=msgr)


def save_predictions(predicted_labels: pd.DataFrame, output_path: Path | str, delta_t: float, msgr: Messenger = Messenger(verbosity=0)) -> None:
    """Tab-separated start/stop/label with header, seconds rounded to 4 places (predict.py:474-499)."""
    predicted_labels = _convert_times_to_seconds(predicted_labels, delta_t)
    predicted_labels[["start", "stop", "label"]].round(4).to_csv(output_path, sep="\t", index=False)
    msgr.info(f"Predictions saved to {output_path}")


def save_prediction_probabilities(aggregated_predictions: np.ndarray, orcai_parameter: dict, delta_t: float, output_path: Path | str,
                                  msgr: Messenger = Messenger(verbosity=0)) -> None:
    """predict.py:502-531."""
    output_path = Path(output_path)
    predictions_path = output_path.with_name(f"{output_path.stem}_probabilities.csv.gz")
    pd.DataFrame(aggregated_predictions, columns=orcai_parameter["calls"], index=delta_t * range(len(aggregated_predictions))).to_csv(
        predictions_path, index_label="time", compression="gzip")
    msgr.info(f"Prediction probabilities saved to {predictions_path}")


def _launch_recording(recording_path: Path | str, channel: int, model, orcai_parameter: dict, shape: dict, output_path: Path | str = "default",
                      overwrite: bool = False, msgr: Messenger = Messenger(verbosity=0), progressbar: tqdm = None) -> dict:
    """predict.py:534-575: output path and overwrite check, then the GPU half of the recording (queued, not waited for)."""
    recording_path = Path(recording_path)
    if output_path is not None:
        if output_path == "default":
            filename = f"{recording_path.stem}_c{channel}_{orcai_parameter['name']}_predicted.txt"
            output_path = recording_path.with_name(filename)
        else:
            output_path = Path(output_path)
        msgr.info(f"Output file: {output_path}")
        if output_path.exists():
            if overwrite:
                msgr.warning(f"Output file {output_path} already exists. Overwriting.")
            else:
                raise FileExistsError(f"Annotation file already exists: {output_path}")
    state = predict_wav_launch(recording_path=recording_path, channel=channel, model=model, orcai_parameter=orcai_parameter, shape=shape, msgr=msgr,
                               progressbar=progressbar)
    state["output_path"] = output_path
    return state


def _finish_recording(state: dict, save_probabilities: bool = False, call_duration_limits: (Path | str) | dict = None, label_suffix: str = "*",
                      msgr: Messenger = Messenger(verbosity=0)) -> None:
    """predict.py:576-632: the host half -- labels, optional duration filter, files."""
    predicted_labels, aggregated_predictions, delta_t = predict_wav_finish(state, label_suffix=label_suffix, msgr=msgr)
    output_path, orcai_parameter = state["output_path"], state["orcai_parameter"]
    if call_duration_limits is not None:
        predicted_labels = filter_predictions(predicted_labels, delta_t=delta_t, call_duration_limits=call_duration_limits, label_suffix=label_suffix, msgr=msgr)
    save_predictions(predicted_labels=predicted_labels, output_path=output_path, delta_t=delta_t, msgr=msgr)
    if save_probabilities:
        save_prediction_probabilities(aggregated_predictions=aggregated_predictions, orcai_parameter=orcai_parameter, delta_t=delta_t, output_path=output_path, msgr=msgr)


def _predict_and_save(recording_path: Path | str, channel: int, model, orcai_parameter: dict, shape: dict, output_path: Path | str = "default",
                      overwrite: bool = False, save_probabilities: bool = False, call_duration_limits: (Path | str) | dict = None,
                      label_suffix: str = "*", msgr: Messenger = Messenger(verbosity=0), progressbar: tqdm = None) -> None:
    """predict.py:534-632."""
    state = _launch_recording(recording_path, channel, model, orcai_parameter, shape, output_path=output_path, overwrite=overwrite, msgr=msgr,
                              progressbar=progressbar)
    _finish_recording(state, save_probabilities=save_probabilities, call_duration_limits=call_duration_limits, label_suffix=label_suffix, msgr=msgr)


def predict(recording_path: str | Path, channel: int = 1, model_dir: str | Path = DEFAULT_MODEL_DIR, output_path: str | Path = "default",
            overwrite: bool = False, save_probabilities: bool = False, base_dir_recording: str | Path | None = None,
            call_duration_limits: str | Path | None = None, label_suffix: str = "*", verbosity: int = 2, msgr: Messenger | None = None) -> None:
    """Predict calls in a wav file or in every recording of a recording-table CSV (predict.py:635-757)."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Predicting calls")
    model_dir = Path(model_dir)
    recording_path = Path(recording_path)
    msgr.part(f"Loading model: {model_dir.stem}")
    model, orcai_parameter, shape = load_orcai_model(model_dir)
    if recording_path.suffix == ".wav":
        return _predict_and_save(recording_path=recording_path, channel=channel, model=model, orcai_parameter=orcai_parameter, shape=shape,
                                 output_path=output_path, overwrite=overwrite, save_probabilities=save_probabilities,
                                 call_duration_limits=call_duration_limits, label_suffix=label_suffix, msgr=msgr, progressbar=None)
    elif recording_path.suffix == ".csv":
        recording_table = pd.read_csv(recording_path)
    else:
        raise ValueError("Recording file must be a wav or csv file")
    if base_dir_recording is not None:
        recording_table["base_dir_recording"] = base_dir_recording
    if (output_path is not None) & (output_path != "default"):
        recording_table["output_path"] = [Path(output_path).joinpath(recording + "_" + model_dir.stem + "_predicted.txt") for recording in recording_table["recording"]]
    else:
        recording_table["output_path"] = output_path
    # one process per GPU (torchrun): each rank takes its own recordings; there is no data-path collective
    from orcai_amd import parallel

    rank, size, _ = parallel.init()
    if size > 1:
        mine = parallel.shard_indices(len(recording_table), rank, size)
        recording_table = recording_table.iloc[mine]
    msgr.part(f"Predicting annotations for {len(recording_table)} wav files" + (f" (rank {rank} of {size})" if size > 1 else ""))
    progressbar = tqdm(recording_table.index, desc="Starting ...", unit="file", disable=verbosity < 1)
    from orcai_amd import wavio

    # decode the next recordings on background threads while the GPU works on the current one (the GPU needs ~60 ms per hour of audio)
    wavio.set_prefetcher(wavio.WavPrefetcher([Path(recording_table.loc[i, "base_dir_recording"]).joinpath(recording_table.loc[i, "rel_recording_path"])
                                              for i in recording_table.index]))
    # one recording deep: the GPU half of recording i is queued before the host half of recording i - 1 (threshold, label table, files) runs
    quiet = Messenger(verbosity=0)
    in_flight = None  # (table index, state)

    def finish(entry):
        j, state = entry
        try:
            _finish_recording(state, save_probabilities=save_probabilities, call_duration_limits=call_duration_limits, label_suffix=label_suffix, msgr=quiet)
        except Exception as e:  # predict.py:752-755: log and continue with the next recording
            msgr.error(f"Error predicting {recording_table.loc[j, 'recording']}: {e.args[0] if e.args else e}")

    for i in progressbar:
        launched = None
        try:
            launched = (i, _launch_recording(recording_path=Path(recording_table.loc[i, "base_dir_recording"]).joinpath(recording_table.loc[i, "rel_recording_path"]),
                                             channel=recording_table.loc[i, "channel"], model=model, orcai_parameter=orcai_parameter, shape=shape,
                                             output_path=recording_table.loc[i, "output_path"], overwrite=overwrite, msgr=quiet, progressbar=progressbar))
        except Exception as e:  # predict.py:752-755
            msgr.error(f"Error predicting {recording_table.loc[i, 'recording']}: {e.args[0] if e.args else e}")
        if in_flight is not None:
            finish(in_flight)
        in_flight = launched
    if in_flight is not None:
        finish(in_flight)
    wavio.set_prefetcher(None)
    msgr.success("Predictions finished.")
