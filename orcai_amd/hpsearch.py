"""Hyperparameter search entry point.  Mirrors reference ``src/orcAI/hpsearch.py:110-257``.

NOT BUILT YET: depends on the training path."""

from __future__ import annotations

from pathlib import Path

from orcai_amd.auxiliary import Messenger


def hyperparameter_search(data_dir: Path | str, output_dir: Path | str, orcai_parameter=None, hps_parameter=None, parallel: bool = False,
                          data_compression: str | None = "GZIP", verbosity: int = 2, msgr: Messenger | None = None) -> None:
    raise NotImplementedError("orcai_amd.hpsearch: needs the HIP training path, which is not built yet")
