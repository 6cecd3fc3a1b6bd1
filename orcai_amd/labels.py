"""Label rasterisation: annotation intervals -> per-frame label arrays.  Mirrors reference ``src/orcAI/labels.py``
(``_convert_annotation`` :18-123, ``create_label_arrays`` :126-259) for the arrays the GPU training-data path reads
(``datasets.RecordingStore``).  Host-side numpy / pandas, pinned by golden vectors from the reference's own
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


def _convert_annotation(annotation_file_path: Path, recording_data_dir: Path, label_calls: list, labels_present: list, labels_masked: list,
                        call_equivalences: (Path | str) | dict = None, msgr: Messenger = Messenger(verbosity=0)) -> tuple[pd.DataFrame, dict]:
    """labels.py:18-123.  1 where a spectrogram frame time lies inside (inclusive) an annotated interval of the label, 0
    elsewhere, MASK_VALUE in the columns of labels that cannot be annotated in this recording.  Like the reference, the label
    column only exists after the call equivalences have been applied (a missing mapping raises KeyError)."""
    msgr.part("Converting annotation to label array")
    annotation_file_path = Path(annotation_file_path)
    recording = annotation_file_path.stem
    annotations = read_annotation_file(annotation_file_path)
    if call_equivalences is not None:
        msgr.info("Applying call equivalences")
        if isinstance(call_equivalences, (Path, str)):
            call_equivalences = read_json(call_equivalences)
        annotations["label"] = annotations["origlabel"].map(call_equivalences)
        missing = set(annotations["origlabel"].unique()).difference(call_equivalences.keys())
        if missing:
            msgr.info(f"labels not in call equivalences: {missing}")
    annotations = annotations[["start", "stop", "label"]]
    spectrogram_dir = Path(recording_data_dir).joinpath(recording, "spectrogram")
    try:
        t_vec = generate_times_from_spectrogram(spectrogram_dir.joinpath("times.json"))
    except FileNotFoundError:
        msgr.error(f"File not found: {spectrogram_dir.joinpath('times.json')}")
        msgr.error("Did you create the spectrogram?")
        raise
    annotations_array = pd.DataFrame({})
    for label in labels_present:
        intervals = annotations[annotations["label"] == label]
        inside = np.zeros(len(t_vec), dtype=bool)
        for start, stop in zip(intervals["start"], intervals["stop"]):
            inside |= (t_vec >= start) & (t_vec <= stop)
        annotations_array[label] = inside.astype(int)
    for label in labels_masked:
        annotations_array[label] = MASK_VALUE * np.ones(len(t_vec), dtype=int)
    annotations_array = annotations_array.reindex(label_calls, axis=1)
    label_dict = dict.fromkeys(labels_present, "present") | dict.fromkeys(labels_masked, "masked")
    return annotations_array, {k: label_dict[k] for k in label_calls}


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
