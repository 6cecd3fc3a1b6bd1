"""Keras-shaped training loop for ResNetLSTM: compile / fit / evaluate, History, and the three callbacks the reference
uses (train.py:155-219): EarlyStopping, ModelCheckpoint, ReduceLROnPlateau.  Host logic only -- every number comes
from the HIP kernels driven by orcai_amd.training.Trainer."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from orcai_amd import _native as N
from orcai_amd import parallel
from orcai_amd.training import L2_LAMBDA, MASK_VALUE, Trainer


class History:
    def __init__(self):
        self.history: dict[str, list] = {}

    def add(self, logs: dict) -> None:
        for k, v in logs.items():
            self.history.setdefault(k, []).append(float(v))


class Callback:
    def on_train_begin(self, loop): ...
    def on_epoch_end(self, loop, epoch: int, logs: dict): ...
    def on_train_end(self, loop): ...


def _better(a, b, mode, min_delta: float = 0.0):
    """Keras' monitor_op with min_delta: an improvement has to beat the best value by more than min_delta."""
    return a - min_delta > b if mode == "max" else a + min_delta < b


def _world_size() -> int:
    """Ranks that all-reduce gradients with this process: the initialised process group, not the launcher's environment (a
    sequential hpsearch started under torchrun has WORLD_SIZE > 1 but no group, and must not call a collective)."""
    import torch.distributed as dist

    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class EarlyStopping(Callback):
    """keras.callbacks.EarlyStopping(monitor, patience, mode, restore_best_weights) (train.py:165-171)."""

    def __init__(self, monitor="val_MBA", patience=10, mode="max", restore_best_weights=True, verbose=0):
        self.monitor, self.patience, self.mode, self.restore = monitor, patience, mode, restore_best_weights
        self.best, self.wait, self.best_state, self.stopped_epoch = None, 0, None, None

    def on_epoch_end(self, loop, epoch, logs):
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.best is None or _better(cur, self.best, self.mode):
            self.best, self.wait = cur, 0
            if self.restore:
                self.best_state = loop.trainer.state_dict()
        else:
            self.wait += 1
            if self.wait >= self.patience:
                loop.stop_training = True
                self.stopped_epoch = epoch

    def on_train_end(self, loop):
        if self.restore and self.best_state is not None and self.stopped_epoch is not None:
            loop.trainer.load_state_dict(self.best_state)


class ModelCheckpoint(Callback):
    """keras.callbacks.ModelCheckpoint(path, monitor, save_best_only=True) (train.py:172-177).  The reference leaves mode="auto",
    which Keras resolves to "min" for a monitor named val_MBA; the monitor is an accuracy, so "max" is used."""

    def __init__(self, filepath, monitor="val_MBA", save_best_only=True, mode="max", verbose=0):
        self.filepath, self.monitor, self.best_only, self.mode, self.best = Path(filepath), monitor, save_best_only, mode, None
        # JSON files written next to the checkpoint whenever it is saved ({file name: object}).  The reference's .keras file carries the
        # architecture; a weights file does not, so a search over architectures (hpsearch) records the trial's resolved parameters here
        self.sidecar: dict = {}

    def on_epoch_end(self, loop, epoch, logs):
        cur = logs.get(self.monitor)
        if not self.best_only or cur is None or self.best is None or _better(cur, self.best, self.mode):
            self.best = cur if cur is not None else self.best
            if parallel.world()[0] == 0:
                loop.trainer.sync_model()
                self.filepath.parent.mkdir(parents=True, exist_ok=True)
                loop.model.save(self.filepath)
                for name, obj in self.sidecar.items():
                    import json

                    self.filepath.parent.joinpath(name).write_text(json.dumps(obj, indent=1))


class ReduceLROnPlateau(Callback):
    """keras.callbacks.ReduceLROnPlateau(monitor, factor, patience, min_lr) (train.py:178-184), mode "max" (see ModelCheckpoint)."""

    def __init__(self, monitor="val_MBA", factor=0.5, patience=3, min_lr=1e-7, mode="max", min_delta=1e-4, verbose=0):
        self.monitor, self.factor, self.patience, self.min_lr, self.mode = monitor, factor, patience, min_lr, mode
        self.min_delta = float(min_delta)  # keras.callbacks.ReduceLROnPlateau default
        self.best, self.wait = None, 0

    def on_epoch_end(self, loop, epoch, logs):
        cur = logs.get(self.monitor)
        logs["learning_rate"] = loop.trainer.lr
        if cur is None:
            return
        if self.best is None or _better(cur, self.best, self.mode, self.min_delta):
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                loop.trainer.lr = max(loop.trainer.lr * self.factor, self.min_lr)
                self.wait = 0


def _epoch_logs(acc: np.ndarray, prefix: str = "") -> dict:
    """acc = {sum of BCE, unmasked count, correct count, l2}: Keras reports loss = mean BCE + regularisation, MBA = accuracy."""
    n = max(acc[1], 1.0)
    return {prefix + "loss": acc[0] / n + acc[3], prefix + "MBA": acc[2] / n}


class FitLoop:
    def __init__(self, model, trainer: Trainer, graph_step: bool | None = None):
        self.model, self.trainer, self.stop_training = model, trainer, False
        # One GPU, ResNetLSTM, no class weights: the whole step is captured once as a hipGraph and replayed (Trainer.train_step_graphed) --
        # what Keras' compiled train function is to the reference (train.py:201-219); the step then no longer depends on the host's launch
        # rate.  orcai_parameter["model"]["graph_step"] = false or ORCAI_GRAPH_STEP=0 keeps eager launches.
        if graph_step is None:
            import os

            graph_step = bool(getattr(model, "graph_step", True)) and os.environ.get("ORCAI_GRAPH_STEP", "1") != "0"
        self.graph_step = bool(graph_step) and getattr(model, "architecture", "") == "ResNetLSTM"

    def evaluate(self, dataset) -> dict:
        """Inference-mode pass (moving BN statistics, no dropout): mean masked BCE + L2 and masked binary accuracy."""
        self.trainer.sync_model()
        m, lib = self.model, N.lib()
        H, W = m.input_hw
        tot = torch.zeros(4, dtype=torch.float64, device=self.trainer.dev)
        acc = torch.zeros(3, dtype=torch.float64, device=self.trainer.dev)
        for xb, yb in dataset:
            B = xb.shape[0]
            probs = torch.empty((B, m.out_steps, m.num_labels), dtype=torch.float32, device=xb.device)
            m.forward_device(xb.contiguous().view(-1), H * W, B, probs, chunk=B)
            N.check(lib.orcai_masked_bce(probs.data_ptr(), yb.contiguous().data_ptr(), probs.numel(), MASK_VALUE, acc.data_ptr(), None, N.stream_ptr()), "masked_bce")
            tot[:3] += acc
        l2 = sum(float((m.weights[k].astype(np.float64) ** 2).sum()) for k in m.weights if k.endswith("/kernel") and (k.startswith("lstm") or k.startswith("dense1")))
        a = tot.cpu().numpy()
        a[3] = L2_LAMBDA * l2
        if _world_size() > 1:
            parts = parallel.gather_objects(a[:3].tolist())
            a[:3] = np.sum(np.array(parts), axis=0)
        return _epoch_logs(a)

    def fit(self, train_dataset, validation_data=None, epochs=1, callbacks=(), class_weight=None, verbose=0) -> History:
        hist = History()
        m = self.model
        H, W = m.input_hw
        world = _world_size()
        cw = None
        if class_weight is not None:  # {class index: weight} (train.py:125-136); classes without an entry weigh 1, as in Keras
            cw = torch.ones(max(m.num_labels, max(int(k) for k in class_weight) + 1), dtype=torch.float32, device=self.trainer.dev)
            for k, v in class_weight.items():
                cw[int(k)] = float(v)
        for cb in callbacks:
            cb.on_train_begin(self)
        skipped_before = int(self.trainer.skipped.item()) if self.trainer.half else 0
        for epoch in range(epochs):
            tot = torch.zeros(4, dtype=torch.float64, device=self.trainer.dev)
            for xb, yb in train_dataset:
                # Keras: sample weight of (snippet, step) = class_weight[argmax over the label axis of y_true]; a loss that returns a
                # scalar (MaskedBinaryCrossentropy does) is multiplied by the batch mean of those weights
                lw = None if cw is None else cw[yb.argmax(dim=-1)].mean().reshape(1)
                if self.graph_step and world == 1 and lw is None:
                    out = self.trainer.train_step_graphed(xb.contiguous().view(-1), H * W, xb.shape[0], yb)  # outputs valid until the next replay
                else:
                    out = self.trainer.train_step(xb.contiguous().view(-1), H * W, xb.shape[0], yb, world_size=world, loss_weight=lw)
                tot[:3] += out["acc"][:3]
                tot[3] = out["acc"][3]
            logs = _epoch_logs(tot.cpu().numpy())
            if self.trainer.half:  # f16 path: steps the device voided because a gradient / batch statistic was not finite (static loss scale)
                skipped = int(self.trainer.skipped.item())
                logs["skipped_steps"] = skipped - skipped_before
                if logs["skipped_steps"] * 2 > max(1, len(train_dataset)):
                    raise RuntimeError(f"f16 training: {logs['skipped_steps']} of {len(train_dataset)} steps of epoch {epoch + 1} overflowed under the static loss scale; "
                                       "train this model with precision 'f32'")
                skipped_before = skipped
            if validation_data is not None:
                logs.update({"val_" + k: v for k, v in self.evaluate(validation_data).items()})
            for cb in callbacks:
                cb.on_epoch_end(self, epoch, logs)
            hist.add(logs)
            if verbose:
                print(f"epoch {epoch + 1}/{epochs}: " + "  ".join(f"{k} {v:.4f}" for k, v in logs.items()), flush=True)
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end(self)
        self.trainer.sync_model()
        return hist
