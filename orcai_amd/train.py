"""Training entry point.  Mirrors reference ``src/orcAI/train.py:39-242``.

NOT BUILT YET (round 1 covers the predict path): the backward kernels, Adam and the RCCL gradient
all-reduce are SURVEY 8 rows C1-C6.  The function exists so that the CLI surface is complete and fails loudly.
"""

from __future__ import annotations

from pathlib import Path

from orcai_amd.auxiliary import Messenger


def train(data_dir: Path | str, output_dir: Path | str, orcai_parameter=None, data_compression: str | None = "GZIP", load_model: bool = False,
          verbosity: int = 2, msgr: Messenger | None = None) -> None:
    raise NotImplementedError("orcai_amd.train: the HIP training path (backward kernels, Adam, RCCL data parallel) is not built yet; "
                              "there is deliberately no CPU/PyTorch-autograd fallback")
