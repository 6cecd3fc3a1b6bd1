"""Label rasterisation: annotation intervals -> per-frame label arrays.  Implements the contract of reference
``src/orcAI/labels.py`` (``_convert_annotation`` :18-123, ``create_label_arrays`` :126-259) for the arrays the GPU
training-data path reads (``datasets.RecordingStore``); the rasterisation itself is written from its definition as an interval
sweep (``rasterise_intervals``), not as the reference's per-interval mask loop.  Host-side numpy / pandas, pinned by golden vectors from the reference's own
``_convert_annotation`` (``tests/golden/labels_raster.*``).  Label arrays are stored as ``labels.npy`` where the reference
writes ``labels.zarr``.
"""

from __future__ import annotations

from importlib.resources import files
from pathlib import Path

import numpy as np
import pandas as pd

from orcai_amd.auxiliary import MASK_VALUE, Messenger
from orcai_amd.io import generate_times_from_spectrogram, read_json, save_array, write_json


def read_annotation_file(annotation_file_path) -> pd.DataFrame:
    """io.py:334-354: tab-separated start, stop, label without header; adds the recording name."""
    table = pd.read_csv(annotation_file_path, sep="\t", encoding="utf-8", header=None, names=["start", "stop", "origlabel"])
    table["recording"] = Path(annotation_file_path).stem
    return table[["recording", "start", "stop", "origlabel"]]


def rasterise_intervals(t_vec: np.ndarray, starts: np.ndarray, stops: np.ndarray) -> np.ndarray:
    """0/1 per frame: 1 where the frame time t satisfies start <= t <= stop for at least one interval.  Written from the
    definition (labels.py:100-107 tests every frame against every interval) as one sweep: frame times ascend, so an interval
    covers the index range [first t >= start, last t <= stop] -- two binary searches per interval, a +1/-1 difference array and a
    running sum instead of len(intervals) passes over all frames.  Intervals with a NaN bound or stop < start cover nothing,
    exactly as the comparisons of the definition do."""
    t_vec = np.asarray(t_vec, dtype=np.float64)
    starts, stops = np.asarray(starts, dtype=np.float64), np.asarray(stops, dtype=np.float64)
    ok = ~(np.isnan(starts) | np.isnan(stops))
    first = np.searchsorted(t_vec, starts[ok], side="left")
    past = np.maximum(np.searchsorted(t_vec, stops[ok], side="right"), first)
    edges = np.zeros(len(t_vec) + 1, dtype=np.int64)
    np.add.at(edges, first, 1)
    np.add.at(edges, past, -1)
    return (np.cumsum(edges[:-1]) > 0).astype(int)


def _convert_annotation(annotation_file_path: Path, recording_data_dir: Path, label_calls: list, labels_present: list, labels_masked: list,
                        call_equivalences: (Path | str) | dict = None, msgr: Messenger = Messenger(verbosity=0)) -> tuple[pd.DataFrame, dict]:
    """Same contract as the reference's ``_convert_annotation`` (labels.py:18-123): a frame of the recording's spectrogram gets 1
    in a label's column when its time lies inside (bounds included) one of the label's annotated intervals, 0 otherwise, and
    MASK_VALUE in the columns of labels that cannot be annotated in this recording; columns come back in ``label_calls`` order
    together with {label: "present" | "masked"}.  The ``label`` column only exists once call equivalences have been applied (a
    missing mapping is a KeyError there too); a missing ``times.json`` is logged and re-raised as FileNotFoundError."""
    msgr.part("Converting annotation to label array")
    annotation_file_path = Path(annotation_file_path)
    table = read_annotation_file(annotation_file_path)
    if call_equivalences is not None:
        msgr.info("Applying call equivalences")
        mapping = read_json(call_equivalences) if isinstance(call_equivalences, (Path, str)) else call_equivalences
        table["label"] = table["origlabel"].map(mapping)
        unmapped = sorted(set(table["origlabel"]) - set(mapping))
        if unmapped:
            msgr.info(f"labels not in call equivalences: {unmapped}")
    times_file = Path(recording_data_dir).joinpath(annotation_file_path.stem, "spectrogram", "times.json")
    if not times_file.exists():
        msgr.error(f"File not found: {times_file}")
        msgr.error("Did you create the spectrogram?")
        raise FileNotFoundError(times_file)
    t_vec = generate_times_from_spectrogram(times_file)
    by_label = {name: rows for name, rows in table[["start", "stop", "label"]].groupby("label", sort=False)}
    columns, status = {}, {}
    for name in label_calls:
        if name in labels_present:
            rows = by_label.get(name)
            columns[name] = (rasterise_intervals(t_vec, rows["start"].to_numpy(), rows["stop"].to_numpy()) if rows is not None
                             else np.zeros(len(t_vec), dtype=int))
            status[name] = "present"
        elif name in labels_masked:
            columns[name] = np.full(len(t_vec), MASK_VALUE, dtype=int)
            status[name] = "masked"
        else:
            raise KeyError(name)  # the reference's final {k: label_dict[k] for k in label_calls} raises the same
    return pd.DataFrame(columns, columns=list(label_calls)), status


def create_label_arrays(recording_table_path: Path | str, output_dir: Path | str, base_dir_annotation: Path | str = None,
                        orcai_parameter: (Path | str) | dict = files("orcai_amd.defaults").joinpath("default_orcai_parameter.json"),
                        call_equivalences: (Path | str) | dict = None, overwrite: bool = False, verbosity: int = 2, msgr: Messenger | None = None) -> None:
    """labels.py:126-259: ``<output_dir>/<recording>/labels/{labels.npy, label_list.json}`` for every annotated recording."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Making label arrays")
    msgr.part("Reading recordings table")
    output_dir = Path(output_dir)
    table = pd.read_csv(recording_table_path)
    if base_dir_annotation is not None:
        table["base_dir_annotation"] = base_dir_annotation
    not_annotated = table["base_dir_annotation"].isna()
    if any(not_annotated):
        msgr.info(f"Skipping {sum(not_annotated)} because of missing annotation files.")
        table = table[~not_annotated]
    if isinstance(orcai_parameter, (Path, str)) or not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    label_calls = orcai_parameter["calls"]
    if not overwrite:
        existing = table["recording"].apply(lambda x: output_dir.joinpath(x, "labels").exists())
        msgr.info(f"Skipping {sum(existing)} recordings because they already have Labels.")
        table = table[~existing]
    msgr.part("Making label arrays")
    no_labels = []
    for i in table.index:
        flags = table.loc[i, label_calls]
        labels_present = list(flags[flags.astype(bool)].index)
        if len(labels_present) == 0:
            no_labels.append(table.loc[i, "recording"])
            continue
        labels_masked = list(set(label_calls).difference(labels_present))
        array, label_dict = _convert_annotation(Path(table.loc[i, "base_dir_annotation"]).joinpath(table.loc[i, "rel_annotation_path"]), output_dir,
                                                label_calls, labels_present, labels_masked, call_equivalences, Messenger(verbosity=0))
        out = output_dir.joinpath(table.loc[i, "recording"], "labels")
        out.mkdir(parents=True, exist_ok=True)
        save_array(array.to_numpy().astype(np.float32), out.joinpath("labels.npy"))
        write_json(label_dict, out.joinpath("label_list.json"))
    if no_labels:
        msgr.warning(f"No valid labels present in {no_labels}")
    msgr.success("Finished making label arrays")
