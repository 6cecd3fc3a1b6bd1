"""Data-preparation commands that sit OUTSIDE the hot path (out of scope for the product package): the
snippet-table builders of ``tools/dataprep/snippets.py`` behind the reference's subcommand names and option flags
(reference cli.py:419-627).  Kept as tooling so that the GPU training-data path can be exercised end to end
from recordings to ``orcai train``:  ``python -m tools.dataprep.cli create-snippet-table ...``.
"""

from __future__ import annotations

import sys
from pathlib import Path

import click

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from orcai_amd.cli import DEFAULT_PARAM, EPILOG, DirR, DirW, DirWcreate, FileR  # noqa: E402


@click.group(epilog=EPILOG)
def dataprep():
    """Snippet-table tooling (not part of the orcai_amd product package)."""


@dataprep.command(name="create-snippet-table", short_help="Creates snippet table.", no_args_is_help=True, epilog=EPILOG,
             help="Creates a table of snippets for all files in recording table at RECORDING_TABLE_PATH and writes them to RECORDING_DATA_DIR.")
@click.argument("recording_table_path", type=FileR)
@click.argument("recording_data_dir", type=DirW)
@click.option("--output_dir", "-o", type=DirWcreate, default=None, show_default="None", help="Output directory; None: tvt_data next to the recording table.")
@click.option("--orcai_parameter", "-p", type=FileR, default=DEFAULT_PARAM, show_default="default_orcai_parameter.json", help="Path to the orcAI parameter file.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_create_snippet_table(**kwargs):
    from orcai_amd.auxiliary import Messenger
    from tools.dataprep.snippets import create_snippet_table

    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Creating snippet table")
    create_snippet_table(**kwargs)


@dataprep.command(name="create-tvt-snippet-tables", short_help="Creates TVT snippet tables.", no_args_is_help=True, epilog=EPILOG,
             help="Creates snippet tables for training, validation and test datasets and saves them to OUTPUT_DIR.")
@click.argument("output_dir", type=DirWcreate)
@click.option("--snippet_table", "-st", type=FileR, default=None, show_default="None", help="Path to the snippet table; None: OUTPUT_DIR/all_snippets.csv.gz.")
@click.option("--orcai_parameter", "-p", type=FileR, default=DEFAULT_PARAM, show_default="default_orcai_parameter.json", help="Path to the orcAI parameter file.")
@click.option("--create_unfiltered_test_snippets", "-uts", is_flag=True, help="Also create an unfiltered test snippet table.")
@click.option("--n_unfiltered_test_snippets", "-n_uts", type=int, default=None, show_default="None", help="Number of unfiltered test snippets; None: as many as training snippets.")
@click.option("--overwrite", "-ow", is_flag=True, help="Overwrite existing snippet tables.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_create_tvt_snippet_tables(**kwargs):
    from orcai_amd.auxiliary import Messenger
    from tools.dataprep.snippets import create_tvt_snippet_tables

    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Creating train, validation and test snippet tables")
    create_tvt_snippet_tables(**kwargs)


@dataprep.command(name="create-tvt-data", short_help="Creates TVT datasets.", no_args_is_help=True, epilog=EPILOG,
             help="Creates training, validation and test datasets from snippet tables in TVT_DIR (descriptors for the GPU-side gather; nothing is materialised).")
@click.argument("tvt_dir", type=DirR)
@click.option("--orcai_parameter", "-p", type=FileR, default=DEFAULT_PARAM, show_default="default_orcai_parameter.json", help="Path to the orcAI parameter file.")
@click.option("--overwrite", "-ow", is_flag=True, help="Overwrite existing datasets.")
@click.option("--data_compression", "-dc", type=click.Choice(["GZIP", "NONE"], case_sensitive=False), default="GZIP", show_default=True, help="Accepted for compatibility; ignored.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_create_tvt_data(**kwargs):
    from orcai_amd.auxiliary import Messenger
    from tools.dataprep.snippets import create_tvt_data

    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Creating train, validation and test datasets")
    create_tvt_data(**kwargs)


if __name__ == "__main__":
    dataprep()
