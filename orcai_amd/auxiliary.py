"""Host utilities the hot path uses.  Mirrors ``src/orcAI/auxiliary.py`` of the reference
(only the symbols the hot path touches; the console Messenger is re-created minimally --
its cosmetics are out of scope)."""

from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

# seed IDs for the different parts of the pipeline (auxiliary.py:16-23)
SEED_ID_MAKE_SNIPPET_TABLE = 1
SEED_ID_FILTER_SNIPPET_TABLE = 2
SEED_ID_CREATE_DATALOADER = {"train": 3, "val": 4, "test": 5, "unfiltered_test": 6}
SEED_ID_LOAD_TRAIN_DATA = 7
SEED_ID_LOAD_VAL_DATA = 8
SEED_ID_LOAD_TEST_DATA = 9
SEED_ID_UNFILTERED_TEST_DATA = 10
SEED_ID_LOAD_UNFILTERED_TEST_DATA = 11

# value used to mask labels in the dataset (auxiliary.py:26)
MASK_VALUE = -1.0


class Messenger:
    """Levelled console logger: 0 error, 1 warning, 2 info, 3 debug (auxiliary.py:29-200)."""

    def __init__(self, title: str | None = None, n_indent: int = 0, verbosity: int = 2, indent_str: str = "    ",
                 show_part_times: bool = True, file: Path | None = None):
        self.n_indent = n_indent
        self.verbosity = verbosity
        self.indent_str = indent_str
        self.show_part_times = show_part_times
        self.file = file
        self.start_time = time.time()
        self.last_part_time = self.start_time
        if title is not None:
            self.start(title)

    def print(self, message, indent=0, set_indent=None, severity=2, prefix="", **kwargs):
        if set_indent is not None:
            self.n_indent = set_indent
        if severity <= self.verbosity:
            text = str(message)
            pad = self.indent_str * self.n_indent
            out = "\n".join(pad + prefix + line for line in text.split("\n"))
            stream = open(self.file, "a") if self.file else (sys.stderr if severity <= 1 else sys.stdout)
            try:
                print(out, file=stream, **kwargs)
            finally:
                if self.file:
                    stream.close()
        self.n_indent = max(self.n_indent + indent, 0)

    def debug(self, message, indent=0, set_indent=None, severity=3, **kwargs):
        self.print(message, indent, set_indent, severity, **kwargs)

    def info(self, message, indent=0, set_indent=None, severity=2, **kwargs):
        self.print(message, indent, set_indent, severity, **kwargs)

    def start(self, message, indent=0, set_indent=0, severity=2, **kwargs):
        self.start_time = self.last_part_time = time.time()
        self.print(f"== {message} ==", indent, set_indent, severity, **kwargs)

    def part(self, message, indent=1, set_indent=0, severity=2, **kwargs):
        now = time.time()
        suffix = f" [total {now - self.start_time:.1f}s, +{now - self.last_part_time:.1f}s]" if self.show_part_times else ""
        self.last_part_time = now
        self.print(f"-- {message}{suffix}", indent, set_indent, severity, **kwargs)

    def success(self, message, indent=0, set_indent=0, severity=2, **kwargs):
        self.print(f"OK {message} [{time.time() - self.start_time:.1f}s]", indent, set_indent, severity, **kwargs)

    def warning(self, message, indent=0, set_indent=None, severity=1, **kwargs):
        self.print(message, indent, set_indent, severity, prefix="WARNING: ", **kwargs)

    def error(self, message, indent=0, set_indent=None, severity=0, **kwargs):
        self.print(message, indent, set_indent, severity, prefix="ERROR: ", **kwargs)

    def print_platform_info(self, severity=2, **kwargs):
        import platform

        self.info(f"platform: {platform.platform()}  python {platform.python_version()}", severity=severity)

    def print_device_info(self, severity=2, **kwargs):
        import torch

        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(0)
            self.info(f"devices: {torch.cuda.device_count()} x {p.name} ({p.total_memory / 2**30:.0f} GiB)", severity=severity)
        else:
            self.warning("no GPU visible: orcai_amd has no CPU fallback")

    print_tf_device_info = print_device_info  # name the reference uses (auxiliary.py:242)

    def print_memory_usage(self, indent=0, set_indent=None, severity=2, **kwargs):
        try:
            import psutil

            self.info(f"memory usage (RSS): {psutil.Process().memory_info().rss / 2**20:.0f} MiB", indent, set_indent, severity)
        except ImportError:
            pass


def resolve_recording_data_dir(recording: str, recording_data_dir):
    """auxiliary.py:347-365: ``<recording_data_dir>/<recording>`` if it exists, else None."""
    from pathlib import Path

    p = Path(recording_data_dir, recording)
    return p if p.exists() else None


def seconds_to_hms(seconds: int) -> str:
    hours, remainder = divmod(seconds, 3600)
    minutes, seconds = divmod(remainder, 60)
    return f"{int(hours):02}:{int(minutes):02}:{int(seconds):02}"


def find_consecutive_ones(binary_vector: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Start and INCLUSIVE end indices of runs of ones (auxiliary.py:420-440)."""
    diff = np.diff(binary_vector, prepend=0, append=0)
    starts = np.where(diff == 1)[0]
    stops = np.where(diff == -1)[0] - 1
    return starts, stops
