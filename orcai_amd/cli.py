"""``orcai`` command line for the hot path.  Same subcommand names, arguments and option flags as the
reference's ``src/orcAI/cli.py`` for the in-scope commands (predict :93-184, create-spectrograms
:359-416, train :630-677, test :680-729, hpsearch :732-788); the data-preparation subcommands are out of scope.
Workflow modules are imported lazily, like the reference does.
"""

from __future__ import annotations

from importlib.resources import files
from pathlib import Path

import click

from orcai_amd.auxiliary import Messenger

INCLUDED_MODELS = ["orcai-V1"]
FileR = click.Path(exists=True, dir_okay=False, readable=True, resolve_path=True, path_type=Path)
DirR = click.Path(exists=True, file_okay=False, readable=True, resolve_path=True, path_type=Path)
DirW = click.Path(exists=True, file_okay=False, writable=True, resolve_path=True, path_type=Path)
DirWcreate = click.Path(exists=False, file_okay=False, writable=True, resolve_path=True, path_type=Path)
DEFAULT_PARAM = files("orcai_amd.defaults").joinpath("default_orcai_parameter.json")
DEFAULT_HPS = files("orcai_amd.defaults").joinpath("default_hps_parameter.json")
EPILOG = "MI355X-native implementation of the orcAI hot path (https://github.com/ethz-tb/orcAI)"


@click.group(help="orcAI on MI355X: detect acoustic signals in spectrograms generated from audio recordings.", epilog=EPILOG)
@click.version_option(package_name=None, version=__import__("orcai_amd").__version__)
def cli():
    pass


@cli.command(name="predict", short_help="Predicts call annotations.", no_args_is_help=True, epilog=EPILOG,
             help="Predicts call annotations from RECORDING_PATH: a wav file or a recording table (.csv).")
@click.argument("recording_path", type=FileR)
@click.option("--channel", "-c", type=int, default=1, show_default=True, help="Channel to use for prediction if running predictions for a single file.")
@click.option("--model", "-m", type=click.Choice(INCLUDED_MODELS, case_sensitive=False), default="orcai-V1", show_default=True,
              help="Builtin model to use for prediction. Overridden if model_dir is given.")
@click.option("--model_dir", "-md", "model_dir", type=DirR, default=None, show_default="use builtin model", help="Path to a model directory.")
@click.option("--output_path", "-o", default="default", show_default="default",
              help="Path to the output file/folder or 'default' to save next to the wav file. None to not save predictions to disk.")
@click.option("--overwrite", "-ow", is_flag=True, help="Overwrite existing predictions.")
@click.option("--save_probabilities", "-sp", is_flag=True, help="Also save the prediction probabilities.")
@click.option("--base_dir_recording", "-bdr", type=DirW, default=None, show_default="None", help="Alternative base directory containing the recordings.")
@click.option("--call_duration_limits", "-cdl", type=FileR, default=None, show_default="None", help="JSON file with call duration limits; None for no filtering.")
@click.option("--label_suffix", "-ls", default="*", show_default=True, help="Suffix to add to the label names.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_predict(**kwargs):
    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Predicting calls")
    from orcai_amd.predict import predict

    if kwargs["model_dir"] is None:
        kwargs["model_dir"] = files("orcai_amd.models").joinpath(kwargs["model"])
    del kwargs["model"]
    if kwargs["output_path"] == "None":
        kwargs["output_path"] = None
    predict(**kwargs)


@cli.command(name="init-weights", short_help="Writes seeded, UNTRAINED weights into a model directory.", no_args_is_help=True, epilog=EPILOG,
             help="Writes <name>.weights.npz with seeded random weights of the architecture described by MODEL_DIR/orcai_parameter.json and "
                  "model_shape.json (the architecture's initialisers; BatchNorm statistics at their initial values).  For plumbing runs where the "
                  "trained orcai-v1.keras is not available: the predictions of such a model mean nothing.  Not a reference command.")
@click.argument("model_dir", type=DirW)
@click.option("--seed", "-s", type=int, default=1, show_default=True, help="Seed of the weight draw.")
@click.option("--overwrite", "-ow", is_flag=True, help="Overwrite an existing weights file.")
def cli_init_weights(model_dir, seed, overwrite):
    from orcai_amd.architectures import build_model
    from orcai_amd.io import WEIGHTS_SUFFIX, read_json

    param = read_json(model_dir.joinpath("orcai_parameter.json"))
    shape = read_json(model_dir.joinpath("model_shape.json"))
    out = model_dir.joinpath(param["name"] + WEIGHTS_SUFFIX)
    if out.exists() and not overwrite:
        raise click.ClickException(f"{out} exists (use --overwrite)")
    param = {**param, "model": {**param["model"], "seed": seed}}
    model = build_model(tuple(shape["input_shape"]), param, msgr=Messenger(verbosity=0))
    model.save_weights(out)
    click.echo(f"wrote {out} (seed {seed}, untrained)")


@cli.command(name="create-spectrograms", short_help="Creates spectrograms for all files in a recording table.", no_args_is_help=True, epilog=EPILOG,
             help="Creates spectrograms for all files in the recording table at RECORDING_TABLE_PATH and saves them to OUTPUT_DIR.")
@click.argument("recording_table_path", type=FileR)
@click.argument("output_dir", type=DirWcreate)
@click.option("--base_dir_recording", "-bdr", type=DirR, default=None, show_default="None", help="Base directory for the wav files.")
@click.option("--orcai_parameter", "-p", type=FileR, default=DEFAULT_PARAM, show_default="default_orcai_parameter.json", help="Path to the orcAI parameter file.")
@click.option("--include_not_annotated", "-en", is_flag=True, help="Include recordings without annotations.")
@click.option("--include_no_possible_annotations", "-enp", is_flag=True, help="Include recordings without possible annotations.")
@click.option("--overwrite", "-ow", is_flag=True, help="Recreate existing spectrograms.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_create_spectrograms(**kwargs):
    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Creating spectrograms")
    from orcai_amd.spectrogram import create_spectrograms

    create_spectrograms(**kwargs)


@cli.command(name="create-label-arrays", short_help="Creates label arrays.", no_args_is_help=True, epilog=EPILOG,
             help="Creates label arrays for all files in recording table at RECORDING_TABLE_PATH and writes them to OUTPUT_DIR.")
@click.argument("recording_table_path", type=FileR)
@click.argument("output_dir", type=DirW)
@click.option("--base_dir_annotation", "-bda", type=DirR, default=None, show_default="None", help="Base directory for the annotation files; None: from the recording table.")
@click.option("--orcai_parameter", "-p", type=FileR, default=DEFAULT_PARAM, show_default="default_orcai_parameter.json", help="Path to the orcAI parameter file.")
@click.option("--call_equivalences", "-ce", type=FileR, default=None, show_default="None", help="Optional JSON associating original call labels with new call labels.")
@click.option("--overwrite", "-ow", is_flag=True, help="Recreate existing label arrays.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_create_label_arrays(**kwargs):
    from orcai_amd.auxiliary import Messenger
    from orcai_amd.labels import create_label_arrays

    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Creating label arrays")
    create_label_arrays(**kwargs)


@cli.command(name="train", short_help="Trains a model on the training dataset.", no_args_is_help=True, epilog=EPILOG,
             help="Trains a model on the dataset in DATA_DIR and saves it to OUTPUT_DIR.")
@click.argument("data_dir", type=DirR)
@click.argument("output_dir", type=DirW)
@click.option("--orcai_parameter", "-p", type=FileR, default=DEFAULT_PARAM, show_default="default_orcai_parameter.json", help="Path to the orcAI parameter file.")
@click.option("--data_compression", "-dc", type=click.Choice(["GZIP", "NONE"], case_sensitive=False), default="GZIP", show_default=True, help="Compression of the data files.")
@click.option("--load_model", "-lm", is_flag=True, help="Load model from previous training.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_train(**kwargs):
    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Training model")
    from orcai_amd.train import train

    train(**kwargs)


@cli.command(name="hpsearch", short_help="Performs hyperparameter search.", no_args_is_help=True, epilog=EPILOG,
             help="Performs hyperparameter search on the dataset in DATA_DIR and saves the results to OUTPUT_DIR.")
@click.argument("data_dir", type=DirR)
@click.argument("output_dir", type=DirW)
@click.option("--orcai_parameter", "-p", type=FileR, default=DEFAULT_PARAM, show_default="default_orcai_parameter.json", help="Path to the orcAI parameter file.")
@click.option("--hps_parameter", "-hp", type=FileR, default=DEFAULT_HPS, show_default="default_hps_parameter.json", help="Path to the hyperparameter search parameter file.")
@click.option("--parallel", "-pl", is_flag=True, help="Run the search data-parallel on all GPUs of the node.")
@click.option("--data_compression", "-dc", type=click.Choice(["GZIP", "NONE"], case_sensitive=False), default="GZIP", show_default=True, help="Compression of the data files.")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_hpsearch(**kwargs):
    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title="Hyperparameter search")
    from orcai_amd.hpsearch import hyperparameter_search

    hyperparameter_search(**kwargs)


@cli.command(name="test", short_help="Tests a model.", no_args_is_help=True, epilog=EPILOG,
             help="Tests a model at MODEL_DIR on the test dataset in DATA_DIR and saves the results to OUTPUT_DIR.")
@click.argument("model_dir", type=DirR)
@click.argument("data_dir", type=DirR)
@click.option("--test_unfiltered", "-tu", is_flag=True, help="If set, the model is also tested on the unfiltered test dataset.")
@click.option("--output_dir", "-o", type=DirWcreate, default=None, show_default="None", help="Path to the output directory. None to save in the same directory as the model.")
@click.option("--data_compression", "-dc", type=click.Choice(["GZIP", "None"], case_sensitive=False), default="GZIP", show_default=True, help="Data compression of saved datasets")
@click.option("--verbosity", "-v", type=click.IntRange(0, 3), default=2, show_default=True, help="0: Errors only, 1: Warnings, 2: Info, 3: Debug")
def cli_test(**kwargs):
    """cli.py:680-729."""
    kwargs["msgr"] = Messenger(verbosity=kwargs["verbosity"], title=f"Testing model {kwargs['model_dir'].name}")
    if kwargs["data_compression"] == "None":
        kwargs["data_compression"] = None
    from orcai_amd.test import test_model

    test_model(**kwargs)


if __name__ == "__main__":
    cli()
