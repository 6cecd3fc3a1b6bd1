"""DERIVED RESTATEMENT, OUTSIDE THE PRODUCT PACKAGE.  This module follows the reference's ``src/orcAI/snippets.py`` statement by
statement (same branches, random-stream order and messages) because its only purpose is to reproduce that module's tables bit
for bit; SURVEY 2 rows 8-9 mark the component out of scope, so it lives under ``tools/`` as data-preparation / test tooling and
nothing in ``orcai_amd/`` imports it (the hot path only needs the index arithmetic in ``orcai_amd.datasets.snippet_rows``).

Snippet tables: which rows of which recording make up the training / validation / test snippets.  Mirrors reference
``src/orcAI/snippets.py`` (``_make_snippet_table`` :26-163, ``_compute_snippet_stats`` :166-193, ``create_snippet_table``
:196-321, ``_filter_snippet_table`` :324-388, ``create_tvt_snippet_tables`` :391-556, ``create_tvt_data`` :618-744): same
function names, arguments, random streams (``default_rng([SEED_ID, seed])``, draws in the same order) and output files, so a
table made here equals the reference's bit for bit -- pinned by golden vectors from the reference's own functions
(``tests/golden/snippet_tables.*``).  Host-side numpy / pandas index arithmetic; the snippets themselves are never
materialised: ``create_tvt_data`` writes, instead of the reference's TF datasets, a small descriptor that
``datasets.load_dataset`` turns into a ``SnippetTableDataset`` gathering batches on the GPU from HBM-resident recordings
(``csrc/datapath.hip``).  Label arrays are read from ``labels/labels.npy`` (the reference: ``labels.zarr``).
"""

from __future__ import annotations

from importlib.resources import files
from pathlib import Path

import numpy as np
import pandas as pd

from orcai_amd.auxiliary import (SEED_ID_FILTER_SNIPPET_TABLE, SEED_ID_MAKE_SNIPPET_TABLE, SEED_ID_UNFILTERED_TEST_DATA, Messenger, resolve_recording_data_dir,
                                 seconds_to_hms)
from orcai_amd.io import read_json, write_json

DATA_TYPES = ["train", "val", "test"]
DEFAULT_ORCAI_PARAMETER = files("orcai_amd.defaults").joinpath("default_orcai_parameter.json")


def _open_labels(recording_dir: Path):
    """Per-frame label array [T, n_labels] of a recording (memory-mapped); FileNotFoundError if absent."""
    return np.load(Path(recording_dir).joinpath("labels", "labels.npy"), mmap_mode="r")


def _make_snippet_table(recording_dir: Path, orcai_parameter: dict, rng=np.random.default_rng(), msgr: Messenger = Messenger(verbosity=2)):
    """snippets.py:26-163.  Random snippet start times per 'segment' of the recording, the first `train` fraction of each
    block of segments for training, then validation, then test; row_start = (index of the first frame time >= t_start) - 1,
    row_stop = row_start + the snippet length rounded down to a multiple of 2**len(filters); label columns = seconds of each
    call inside the snippet (NaN where the label is masked).  Returns (table | None, duration, n_segments, recording, status)."""
    recording_dir = Path(recording_dir)
    recording = recording_dir.stem
    label_list_path = recording_dir.joinpath("labels", "label_list.json")
    spectrogram_times_path = recording_dir.joinpath("spectrogram", "times.json")
    try:
        spectrogram_times = read_json(spectrogram_times_path)
    except FileNotFoundError:
        msgr.error(f"File not found: {spectrogram_times_path}")
        msgr.error("Did you create the spectrogram?")
        raise
    model_parameter = orcai_parameter["model"]
    sp = orcai_parameter["snippets"]
    recording_duration = spectrogram_times["max"]
    n_segments = int(recording_duration // sp["segment_duration"])
    if n_segments <= 0:
        msgr.warning(f"Duration of recording ({recording_duration}) is shorter than segment length ({sp['segment_duration']}). Skipping recording.")
        return (None, recording_duration, n_segments, recording, "shorter than segment_duration")
    try:
        labels = _open_labels(recording_dir)
    except FileNotFoundError:
        msgr.warning(f"Label file not found: {recording_dir.joinpath('labels', 'labels.npy')}")
        return (None, recording_duration, n_segments, recording, "missing label files")
    try:
        label_list = read_json(label_list_path)
    except FileNotFoundError:
        msgr.warning(f"Label file not found: {label_list_path}")
        return (None, recording_duration, n_segments, recording, "missing label files")
    label_names = list(label_list.keys())

    times = np.linspace(spectrogram_times["min"], spectrogram_times["max"], spectrogram_times["length"])
    delta_t = times[1] - times[0]
    factor = 2 ** len(model_parameter["filters"])
    n_steps = int(factor * ((sp["snippet_duration"] / delta_t) // factor))  # time axis divisible by 2**n_filters
    msgr.info(f"Number of spectrogram snippet timesteps: {n_steps}")
    rows = []
    for i_segment in range(n_segments):
        msgr.info(f"Segment {i_segment + 1} of {n_segments}")
        lo_hi = (0, 0)
        for data_type in DATA_TYPES:
            lo_hi = (lo_hi[1], lo_hi[1] + sp[data_type])
            t_min = (i_segment + lo_hi[0]) * sp["segment_duration"]
            for _ in range(int(sp[data_type] * sp["segment_duration"] * sp["snippets_per_sec"])):
                t_max = (i_segment + lo_hi[1]) * sp["segment_duration"] - sp["snippet_duration"]
                t_start = rng.uniform(low=t_min, high=t_max, size=1)[0]
                row_start = np.searchsorted(times, t_start, side="left") - 1
                row_stop = row_start + n_steps
                duration = np.asarray(labels[row_start:row_stop, :]).sum(axis=0) * delta_t
                duration[duration < 0] = np.nan
                rows.append([recording, str(recording_dir), data_type, row_start, row_stop] + list(duration))
    table = pd.DataFrame(rows, columns=["recording", "recording_data_dir", "data_type", "row_start", "row_stop"] + label_names)
    return (table.drop_duplicates(), recording_duration, n_segments, recording, "success")  # duplicates arise from the random sampling


def _compute_snippet_stats(snippet_table: pd.DataFrame, for_calls: list) -> pd.DataFrame:
    """snippets.py:166-193: seconds per call and data type, their total, and the 'equalizing factors' max/x per column."""
    stats = snippet_table.groupby("data_type")[for_calls].sum().T
    stats = stats.reindex(columns=DATA_TYPES)
    stats["total"] = stats.sum(axis=1)
    factors = stats.apply(lambda x: 1 / x * x.max(), axis=0)
    factors.columns = factors.columns + "_ef"
    return pd.merge(stats, factors, left_index=True, right_index=True)


def create_snippet_table(recording_table_path: Path | str, recording_data_dir: Path | str, output_dir: Path | str = None,
                         orcai_parameter: dict | (Path | str) = DEFAULT_ORCAI_PARAMETER, verbosity: int = 2, msgr: Messenger | None = None) -> None:
    """snippets.py:196-321.  Snippet tables of all annotated recordings of the recording table (one random stream across the
    recordings, in table order) -> ``<output_dir>/all_snippets.csv.gz`` and ``failed_snippets.csv``."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Making snippet table")
    msgr.part("Reading recording table")
    if isinstance(orcai_parameter, (Path, str)) or not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    output_dir = Path(recording_table_path).parent.joinpath("tvt_data") if output_dir is None else Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    recording_data_dir = Path(recording_data_dir)
    recording_table = pd.read_csv(Path(recording_table_path))
    recording_table = recording_table[~recording_table["base_dir_annotation"].isna()]  # recordings without annotation
    recording_table["recording_data_dir"] = recording_table.apply(lambda row: resolve_recording_data_dir(row["recording"], recording_data_dir), axis=1)
    missing = pd.isna(recording_table["recording_data_dir"])
    if any(missing):
        msgr.warning(f"Missing recording data directories for {sum(missing)} recordings. Skipping these recordings.")
        msgr.warning("Did you create the spectrograms & Labels?")
        recording_table = recording_table[~missing]

    lengths, segments, tables, failed, reasons = [], [], [], [], []
    msgr.part("Making snippet tables")
    rng = np.random.default_rng(seed=[SEED_ID_MAKE_SNIPPET_TABLE, orcai_parameter["seed"]])
    for i in recording_table.index:
        table, length, n_segments, recording, result = _make_snippet_table(Path(recording_table.loc[i, "recording_data_dir"]), orcai_parameter, rng=rng,
                                                                           msgr=Messenger(verbosity=0))
        if result == "success":
            tables.append(table)
            lengths.append(length)
            segments.append(n_segments)
        else:
            failed.append(recording)
            reasons.append(result)
    snippet_table = pd.concat(tables).reset_index(drop=True)
    failed_table = pd.DataFrame({"recording": failed, "reason": reasons})
    msgr.info(f"Created snippet table for {len(snippet_table['recording'].unique())} recordings.")
    msgr.info(f"Total recording duration: {seconds_to_hms(np.sum(lengths))}.")
    msgr.info(f"Total number of snippets: {len(snippet_table)}.")
    msgr.info(f"Total number of segments: {np.sum(segments)}")
    msgr.info(f"Creating snippet table failed for {len(failed)} recordings.", indent=1)
    msgr.info(failed_table.groupby("reason").size(), indent=-1)
    msgr.part("Saving snippet table...")
    failed_table.to_csv(output_dir.joinpath("failed_snippets.csv"), index=False)
    snippet_table.to_csv(output_dir.joinpath("all_snippets.csv.gz"), compression="gzip", index=False)
    msgr.success(f"Snippet table saved to {output_dir.joinpath('all_snippets.csv.gz')}")


def _filter_snippet_table(snippet_table: pd.DataFrame, orcai_parameter: dict, rng=np.random.default_rng(), msgr: Messenger = Messenger(verbosity=2)) -> pd.DataFrame:
    """snippets.py:324-388: drop a random `fraction_removal` of the snippets that contain none of the calls."""
    msgr.part("Filtering snippet table")
    calls = orcai_parameter["calls"]
    no_label = snippet_table[snippet_table[calls].sum(axis=1) <= 0.0000001]
    msgr.info(f"Percentage of snippets containing no label before selection: {np.around(100 * len(no_label) / snippet_table.shape[0], 2)} %")
    fraction = orcai_parameter["snippets"]["fraction_removal"]
    msgr.info(f"removing {np.around(fraction * 100, 2)}% of snippets without label")
    drop = rng.choice(no_label.index, size=int(fraction * len(no_label)), replace=False)
    snippet_table = snippet_table.drop(drop, axis=0)
    no_label = snippet_table[snippet_table[calls].sum(axis=1) <= 0.0000001]
    msgr.info(f"Percentage of snippets containing no label after selection: {np.around(100 * len(no_label) / snippet_table.shape[0], 2)} %")
    snippet_table = snippet_table.reset_index(drop=True)
    msgr.info("Number of train, val, test snippets:", indent=1)
    msgr.info(snippet_table.groupby("data_type").size(), indent=-1)
    return snippet_table


def create_tvt_snippet_tables(output_dir: Path | str, snippet_table: (Path | str) | pd.DataFrame | None = None, orcai_parameter: Path | str = DEFAULT_ORCAI_PARAMETER,
                              create_unfiltered_test_snippets: bool = False, n_unfiltered_test_snippets: int | None = None, overwrite: bool = False,
                              verbosity: int = 2, msgr: Messenger | None = None) -> None:
    """snippets.py:391-556.  Filter the snippet table, then sample n_batch_<type> * batch_size snippets per data type (one
    random stream: filter draws, then train, val, test samples) -> ``{train,val,test}.csv.gz`` (+ ``test_unfiltered.csv.gz``),
    ``all_snippet_stats_duration.csv``, ``selected_snippet_stats_duration.csv``.  ValueError if a type has too few snippets."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Creating train, validation and test snippet tables")
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    msgr.part("Reading snippet table")
    if not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    if snippet_table is None:
        snippet_table = output_dir.joinpath("all_snippets.csv.gz")
    if isinstance(snippet_table, (Path, str)):
        snippet_table = pd.read_csv(snippet_table)

    calls = orcai_parameter["calls"]
    all_stats = _compute_snippet_stats(snippet_table, for_calls=calls).filter(regex=".*(?<!_ef)$", axis=1).map(seconds_to_hms)
    msgr.info("Snippet stats [HMS]:", indent=1)
    msgr.info(all_stats, indent=-1)
    all_stats.to_csv(output_dir.joinpath("all_snippet_stats_duration.csv"), index=True)

    rng = np.random.default_rng(seed=[SEED_ID_FILTER_SNIPPET_TABLE, orcai_parameter["seed"]])
    filtered = _filter_snippet_table(snippet_table, orcai_parameter=orcai_parameter, rng=rng, msgr=msgr)
    snippets = []
    for i, itype in enumerate(DATA_TYPES):
        n_snippets = orcai_parameter["model"][f"n_batch_{itype}"] * orcai_parameter["model"]["batch_size"]
        msgr.info(f"Extracting {orcai_parameter['model'][f'n_batch_{itype}']} batches of {orcai_parameter['model']['batch_size']} random {itype} snippets ({n_snippets} snippets)")
        table_i = filtered[filtered["data_type"] == itype]
        if len(table_i) < n_snippets:
            raise ValueError(f"Number of {itype} snippets ({n_snippets}) larger than available snippets ({len(table_i)}).")
        snippets.append(table_i.sample(n=n_snippets, replace=False, random_state=rng))
        path_i = output_dir.joinpath(f"{itype}.csv.gz")
        if path_i.exists() and not overwrite:
            msgr.warning(f"File {path_i} already exists. Skipping. Set overwrite=True to overwrite.")
            continue
        snippets[i][["recording_data_dir", "row_start", "row_stop"]].to_csv(path_i, compression="gzip", index=False)
        msgr.info(f"saved {itype} snippets to disk")

    selected = _compute_snippet_stats(pd.concat(snippets, ignore_index=True), for_calls=calls).filter(regex=".*(?<!_ef)$", axis=1).map(seconds_to_hms)
    msgr.info("Snippet stats for train, val and test datasets [HMS]:", indent=1)
    msgr.info(selected, indent=-1)
    selected.to_csv(output_dir.joinpath("selected_snippet_stats_duration.csv"), index=True)

    if create_unfiltered_test_snippets:
        if n_unfiltered_test_snippets is None:
            n_unfiltered_test_snippets = orcai_parameter["model"]["n_batch_train"] * orcai_parameter["model"]["batch_size"]
        msgr.info(f"Extracting {n_unfiltered_test_snippets} unfiltered test snippets")
        all_test = snippet_table[snippet_table["data_type"] == "test"]
        if len(all_test) < n_unfiltered_test_snippets:
            msgr.warning(f"Number of unfiltered test snippets ({n_unfiltered_test_snippets}) larger than available snippets ({len(all_test)}).")
            msgr.warning("Using all test snippets.")
            n_unfiltered_test_snippets = len(all_test)
        rng = np.random.default_rng(seed=[SEED_ID_UNFILTERED_TEST_DATA, orcai_parameter["seed"]])
        unfiltered = all_test.sample(n=n_unfiltered_test_snippets, replace=False, random_state=rng)
        path_u = output_dir.joinpath("test_unfiltered.csv.gz")
        if path_u.exists() and not overwrite:
            msgr.warning(f"File {path_u} already exists. Skipping. Set overwrite=True to overwrite.")
        else:
            unfiltered.to_csv(path_u, compression="gzip", index=False)
            msgr.info("saved unfiltered test snippets to disk")
    msgr.success("All snippet tables created and saved to disk")


TABLE_DATASET_FILE = "snippet_table_dataset.json"


def create_tvt_data(tvt_dir: Path | str, orcai_parameter: dict | (Path | str) = DEFAULT_ORCAI_PARAMETER, overwrite: bool = False,
                    data_compression: str | None = "GZIP", verbosity: int = 2, msgr: Messenger | None = None) -> dict:
    """snippets.py:618-744 without the materialisation: for each of ``{train,val,test[,test_unfiltered]}.csv.gz`` in `tvt_dir`
    writes ``<type>_dataset/snippet_table_dataset.json`` (table path + n_filters) -- what ``load_dataset`` needs to gather the
    batches on the GPU from the recordings' spectrogram / label arrays -- and ``dataset_shapes.json``.  The 120 GB of
    materialised, gzip-compressed TF datasets the reference writes here do not exist.  `data_compression` is accepted and
    ignored.  Returns {type: descriptor dict}.  FileExistsError semantics as the reference: existing datasets are skipped with
    a warning unless `overwrite`."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Creating train, validation and test datasets")
    tvt_dir = Path(tvt_dir)
    data_types = list(DATA_TYPES)
    if tvt_dir.joinpath("test_unfiltered.csv.gz").exists():
        data_types.append("test_unfiltered")
    msgr.part("Reading in snippet tables")
    if not isinstance(orcai_parameter, dict):
        orcai_parameter = read_json(orcai_parameter)
    if orcai_parameter["model"].get("call_weights") is not None:
        raise NotImplementedError("call_weights (class weights) are not supported by the MI355X training path")
    n_filters = len(orcai_parameter["model"]["filters"])
    out, shapes = {}, None
    for itype in data_types:
        table = pd.read_csv(tvt_dir.joinpath(f"{itype}.csv.gz"))
        rows = (table["row_stop"] - table["row_start"]).unique()
        if len(rows) != 1:
            raise ValueError(f"{itype}: snippets of different lengths {sorted(rows)}")
        if int(rows[0]) % 2**n_filters:
            raise ValueError("The number of rows in 'arr' must be divisible by 2**'n_filters'.")  # io.py:123-126
        first = Path(table["recording_data_dir"].iloc[0])
        width = int(np.load(first.joinpath("spectrogram", "spectrogram.npy"), mmap_mode="r").shape[1])
        n_labels = int(np.load(first.joinpath("labels", "labels.npy"), mmap_mode="r").shape[1])
        if shapes is None:
            shapes = {"spectrogram": [int(rows[0]), width, 1], "labels": [int(rows[0]) // 2**n_filters, n_labels]}
            msgr.info("Data shape:", indent=1)
            msgr.info(f"Input spectrogram batch shape: {tuple(shapes['spectrogram'])}")
            msgr.info(f"Input label batch shape: {tuple(shapes['labels'])}", indent=-1)
        desc = {"snippet_table": f"../{itype}.csv.gz", "n_filters": n_filters, "length": int(len(table))}
        ddir = tvt_dir.joinpath(f"{itype}_dataset")
        if ddir.joinpath(TABLE_DATASET_FILE).exists() and not overwrite:
            msgr.warning(f"File {ddir} already exists. Skipping. Set overwrite=True to overwrite.")
        else:
            ddir.mkdir(parents=True, exist_ok=True)
            write_json(desc, ddir.joinpath(TABLE_DATASET_FILE))
        msgr.info(f"{itype.capitalize()} dataset created. Length {len(table)}.")
        out[itype] = desc
    write_json(shapes, tvt_dir.joinpath("dataset_shapes.json"))
    msgr.success("Train, validation and test datasets created and saved to disk")
    return out
