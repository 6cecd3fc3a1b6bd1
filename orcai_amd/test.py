"""``orcai test``: evaluation of a trained model on the test datasets.  Mirrors reference ``src/orcAI/test.py``:
``compute_confusion_table`` (:160-225), ``compute_misclassification_tables`` (:37-157), ``_test_model_on_dataset`` (:228-287),
``_save_test_results`` (:290-315), ``test_model`` (:318-420).

The tables are host-side numpy / pandas (pinned bit-for-bit by golden vectors from the reference's own functions,
``tests/golden/test_tables.*``); the forward passes behind them run on the GPU through the model's HIP kernels.
"""

from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np
import pandas as pd

from orcai_amd.auxiliary import MASK_VALUE, SEED_ID_LOAD_TEST_DATA, SEED_ID_LOAD_UNFILTERED_TEST_DATA, Messenger
from orcai_amd.datasets import load_dataset
from orcai_amd.io import load_orcai_model


def _stack_batch(batch) -> np.ndarray:
    """test.py:23-26: (B, T, L) -> (B*T, L) int."""
    return np.vstack(batch).astype(int)


def _get_mask_for_rows_with_atmost_one_1(matrix: np.ndarray) -> np.ndarray:
    """test.py:29-34."""
    return np.sum(matrix == 1, axis=1) <= 1


def _compute_misclassification_table(label_matrix_1: np.ndarray, label_matrix_2: np.ndarray, suffix_1: str, suffix_2: str, label_names: list[str]) -> pd.DataFrame:
    """test.py:37-110.  Row r of matrix 1 with exactly one label a (whose column in matrix 2 is not masked) spreads one unit over
    the labels matrix 2 sets in that row (or NOLABEL); a row without a label does the same from the NOLABEL row.  The
    accumulation runs in row order (np.add.at), i.e. the same float64 sums as the reference's loop."""
    num_labels = len(label_names)
    m = np.zeros((num_labels + 1, num_labels + 1))
    ones1 = label_matrix_1 == 1
    ones2 = label_matrix_2 == 1
    n1 = ones1.sum(axis=1)
    n2 = ones2.sum(axis=1)
    if np.any(n1 > 1):
        print("WARNING: more than one 1 in row of matrix y_true_stacked_drop")
    a = np.argmax(ones1, axis=1)  # the single label of the row (valid where n1 == 1)
    rows = np.arange(label_matrix_1.shape[0])
    not_masked = label_matrix_2[rows, a] != -1
    src = np.where(n1 == 1, a, num_labels)  # source row of the table
    use = ((n1 == 1) & not_masked) | (n1 == 0)
    # rows whose matrix-2 side has labels: 1/n2 to each of them, in (row, column) order
    r_idx, c_idx = np.nonzero(ones2 & (use & (n2 > 0))[:, None])
    np.add.at(m, (src[r_idx], c_idx), 1.0 / n2[r_idx])
    none = use & (n2 == 0)
    np.add.at(m, (src[none], np.full(int(none.sum()), num_labels)), 1.0)
    row_sum = np.sum(m, axis=1, keepdims=True)
    with np.errstate(all="ignore"):
        m = np.around(m / row_sum, 3)
        table = pd.DataFrame(m)
        table.columns = [suffix_2 + "_" + x for x in label_names] + [suffix_2 + "_NOLABEL"]
        table.index = [suffix_1 + "_" + x for x in label_names] + [suffix_1 + "_NOLABEL"]
        table["fraction_time"] = np.around(row_sum / sum(row_sum), 5)
    return table


def compute_misclassification_tables(label_matrix_1: np.ndarray, label_matrix_2: np.ndarray, suffix_1: str, suffix_2: str, label_names: list[str]) -> dict:
    """test.py:113-157."""
    mask1 = _get_mask_for_rows_with_atmost_one_1(label_matrix_1)
    mask2 = _get_mask_for_rows_with_atmost_one_1(label_matrix_2)
    return {
        "_".join([suffix_1, suffix_2]): _compute_misclassification_table(label_matrix_1[mask1], label_matrix_2[mask1], suffix_1, suffix_2, label_names),
        "_".join([suffix_2, suffix_1]): _compute_misclassification_table(label_matrix_2[mask2], label_matrix_1[mask2], suffix_2, suffix_1, label_names),
    }


def compute_confusion_table(y_true_batch: np.ndarray, y_pred_batch: np.ndarray, label_names: list[str]) -> pd.DataFrame:
    """test.py:160-225: per label, over the unmasked elements, with predictions thresholded at >= 0.5."""
    y_true_batch = np.array(y_true_batch)
    y_pred_binary = (np.asarray(y_pred_batch) >= 0.5).astype(int)
    assert y_true_batch.shape == y_pred_binary.shape, "Shapes of y_true_batch and y_pred_binary_batch must match"
    table = {}
    for idx, name in enumerate(label_names):
        t = y_true_batch[:, :, idx].flatten()
        p = y_pred_binary[:, :, idx].flatten()
        keep = t != MASK_VALUE
        t, p = t[keep], p[keep]
        tn = int(np.sum((t == 0) & (p == 0)))
        fp = int(np.sum((t == 0) & (p == 1)))
        fn = int(np.sum((t == 1) & (p == 0)))
        tp = int(np.sum((t == 1) & (p == 1)))
        tot = tn + fp + fn + tp
        with np.errstate(all="ignore"):
            table[name] = {
                "TP": float(np.float64(tp) / tot), "FN": float(np.float64(fn) / tot), "FP": float(np.float64(fp) / tot), "TN": float(np.float64(tn) / tot),
                "PR": float(tp / (tp + fp)) if tp + fp > 0 else np.nan,
                "RE": float(tp / (tp + fn)) if tp + fn > 0 else np.nan,
                "F1": float(2 * tp / (2 * tp + fp + fn)) if tp + fp + fn > 0 else np.nan,
                "Total": int(tot),
            }
    return pd.DataFrame.from_dict(table, orient="index").sort_values(by="Total", ascending=False)


def _test_model_on_dataset(model, dataset, label_names: list[str], dataset_name: str, msgr: Messenger) -> dict:
    """test.py:228-287."""
    msgr.part(f"Testing model on {dataset_name}")
    msgr.info(f"Evaluating model on {dataset_name}")
    data_metrics = model.evaluate(dataset, return_dict=True, verbose=0 if msgr.verbosity < 3 else 1)
    msgr.info(data_metrics)
    msgr.part(f"Calculating confusion table for {dataset_name}")
    data_true, data_predicted = [], []
    for spectrogram_batch, label_batch in dataset:
        data_true.append(label_batch.cpu().numpy() if hasattr(label_batch, "cpu") else np.asarray(label_batch))
        data_predicted.append(_predict_batch(model, spectrogram_batch))
    data_true = np.concatenate(data_true, axis=0)
    data_predicted = np.concatenate(data_predicted, axis=0)
    confusion_table = compute_confusion_table(data_true, data_predicted, label_names)
    msgr.info(confusion_table)
    tables = compute_misclassification_tables(_stack_batch(data_true), _stack_batch((data_predicted >= 0.5).astype(int)), "true", "pred", label_names)
    msgr.part("Misclassification tables on dataset:")
    for key, table in tables.items():
        msgr.info("\n" + key, indent=1)
        msgr.info(table, indent=-1)
    return {"dataset": dataset_name, "data_metrics": data_metrics, "confusion_table": confusion_table, "misclassification_tables": tables}


def _predict_batch(model, spectrogram_batch) -> np.ndarray:
    """A device batch [B][H][W] goes straight through the HIP forward; anything else through the keras-shaped ``predict``."""
    import torch

    if isinstance(spectrogram_batch, torch.Tensor) and spectrogram_batch.is_cuda and hasattr(model, "forward_device"):
        x = spectrogram_batch.contiguous()
        B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        out = torch.empty((B, model.out_steps, model.num_labels), dtype=torch.float32, device=x.device)
        model.forward_device(x.view(-1), H * W, B, out, chunk=B)
        return out.cpu().numpy()
    x = np.asarray(spectrogram_batch.cpu() if hasattr(spectrogram_batch, "cpu") else spectrogram_batch, dtype=np.float32)
    return model.predict(x[..., None] if x.ndim == 3 else x, verbose=0)


def _save_test_results(results: dict, save_results_dir: Path, msgr: Messenger) -> None:
    """test.py:290-315."""
    msgr.part("Saving test results")
    name = results["dataset"]
    os.makedirs(save_results_dir, exist_ok=True)
    with open(save_results_dir.joinpath(name + "_metrics.json"), "w") as f:
        json.dump(results["data_metrics"], f)
    results["confusion_table"].to_csv(save_results_dir.joinpath(name + "_confusion_table.csv"), index_label="Label")
    for key, table in results["misclassification_tables"].items():
        table.to_csv(save_results_dir.joinpath(name + "_misclassification_table_" + key + ".csv"), index_label="Label")


def test_model(model_dir: Path | str, data_dir: Path | str, test_unfiltered: bool = True, output_dir: None | Path | str = None,
               data_compression: str | None = "GZIP", verbosity: int = 2, msgr: Messenger | None = None) -> None:
    """test.py:318-420."""
    if msgr is None:
        msgr = Messenger(verbosity=verbosity, title="Testing model")
    data_dir, model_dir = Path(data_dir), Path(model_dir)
    output_dir = model_dir.joinpath("test") if output_dir is None else Path(output_dir)
    msgr.part("Loading model")
    msgr.info(f"Model directory: {model_dir}")
    msgr.info(f"Model data directory: {data_dir}")
    model, orcai_parameter, _ = load_orcai_model(model_dir)
    model_parameter = orcai_parameter["model"]
    trained_calls = orcai_parameter["calls"]
    seed = orcai_parameter.get("seed")
    todo = [("test_dataset", "test_data", SEED_ID_LOAD_TEST_DATA)]
    if test_unfiltered:
        todo.append(("test_unfiltered_dataset", "test_unfiltered_dataset", SEED_ID_LOAD_UNFILTERED_TEST_DATA))
    for folder, name, seed_id in todo:
        dataset = load_dataset(data_dir.joinpath(folder), model_parameter["batch_size"], compression=data_compression,
                               seed=[seed_id, seed] if seed is not None else None)
        results = _test_model_on_dataset(model, dataset, trained_calls, name, msgr)
        _save_test_results(results, output_dir, msgr)
        msgr.info(f"Saved test results to {output_dir}")
    msgr.success("Model testing completed.")


test_model.__test__ = False  # not a pytest test
