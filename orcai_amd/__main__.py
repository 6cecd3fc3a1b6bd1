from orcai_amd.cli import cli

cli()
