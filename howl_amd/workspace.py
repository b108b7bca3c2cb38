"""``Workspace`` (``howl/workspace.py:15-110``): same artefacts -- ``model.pt.bin`` / ``model-best.pt.bin`` (bare
``state_dict``), ``zmuv.pt.bin``, ``settings.json``, ``cmd-args.json`` -- so checkpoints move between stock Howl and this
package in both directions.  tensorboard is not a dependency here: scalars go to ``logs/scalars.jsonl`` through an
object with ``SummaryWriter.add_scalar``'s signature, buffered so that logging never synchronises the device."""
import json
import shutil
from dataclasses import dataclass
from pathlib import Path

import torch
import torch.nn as nn

from howl_amd.settings import KEY_TO_SETTINGS_CLASS, SETTINGS, HowlSettings


class NullWriter:
    """What the non-zero ranks of a data-parallel job log to."""

    def add_scalar(self, tag, value, step=None):
        pass

    def flush(self):
        pass

    def close(self):
        pass


class ScalarWriter:
    def __init__(self, log_dir: Path):
        self.path = Path(log_dir) / "scalars.jsonl"
        self.path.parent.mkdir(parents=True, exist_ok=True)
        self._pending = []

    def add_scalar(self, tag, value, step=None):
        self._pending.append((tag, value, step))   # device tensors stay on the device until flush()
        if len(self._pending) >= 256:
            self.flush()

    def flush(self):
        # device scalars come over in ONE copy (a .item() per value was one synchronising read-back each)
        dev = [(i, v) for i, (_, v, _) in enumerate(self._pending) if torch.is_tensor(v) and v.is_cuda]
        host = {}
        if dev:
            vals = torch.stack([v.detach().reshape(()).float() for _, v in dev]).cpu().tolist()
            host = {i: x for (i, _), x in zip(dev, vals)}
        with self.path.open("a") as f:
            for i, (tag, value, step) in enumerate(self._pending):
                v = host[i] if i in host else (float(value.item()) if torch.is_tensor(value) else float(value))
                f.write(json.dumps({"tag": tag, "value": v, "step": step}) + "\n")
        self._pending = []

    def close(self):
        self.flush()


@dataclass
class Workspace:
    path: Path
    best_quality: float = float("-inf")
    delete_existing: bool = True
    writable: bool = True      # False on the non-zero ranks of a data-parallel job: they read checkpoints, rank 0 writes

    def __post_init__(self):
        self.path = Path(self.path)
        if not self.writable:
            self.summary_writer = NullWriter()
            return
        self.path.mkdir(parents=True, exist_ok=True)
        log_path = self.path / "logs"
        if self.delete_existing:
            shutil.rmtree(str(log_path), ignore_errors=True)
        self.summary_writer = ScalarWriter(log_path)

    def model_path(self, best=False):
        return str(self.path / f'model{"-best" if best else ""}.pt.bin')

    def write_args(self, args):
        if not self.writable:
            return
        with (self.path / "cmd-args.json").open("w") as f:
            json.dump({k: v for k, v in vars(args).items()}, f, indent=2, default=str)

    def increment_model(self, model: nn.Module, quality):
        if quality > self.best_quality:
            self.save_model(model, best=True)
            self.best_quality = quality
        self.save_model(model, best=False)

    def save_model(self, model: nn.Module, best: bool = False):
        if not self.writable:
            return
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, self.model_path(best=best))

    def load_model(self, model: nn.Module, best=True):
        model.load_state_dict(torch.load(self.model_path(best=best), map_location="cpu"))

    def save_settings(self, settings: HowlSettings = SETTINGS):
        if not self.writable:
            return
        with (self.path / "settings.json").open("w") as f:
            out = {k: getattr(settings, k).dict() for k in KEY_TO_SETTINGS_CLASS
                   if k not in ("_dataset", "_raw_dataset", "_resource") and getattr(settings, k) is not None}
            json.dump(out, f, indent=2)

    def load_settings(self, settings: HowlSettings = SETTINGS) -> HowlSettings:
        with (self.path / "settings.json").open("r") as f:
            for key, value in json.load(f).items():
                setattr(settings, key, KEY_TO_SETTINGS_CLASS[key](**value))
        return settings
