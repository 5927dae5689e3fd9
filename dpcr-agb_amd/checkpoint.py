"""Reading and writing the reference's checkpoint files, so that weights trained with StefOe/DPCR-AGB load into this
build and vice versa (torch_points3d/metrics/model_checkpoint.py:23-148 ``Checkpoint``; base_model.py:151-171
``load_state_dict_with_same_shape`` / ``set_pretrained_weights``).

File layout (one ``torch.save`` dict):  ``models`` {"latest": state_dict, "best_<metric>": state_dict, ...},
``optimizer`` (class name, state_dict), ``schedulers`` {name: [opt, state_dict]}, ``grad_scale``, ``stats``
{"train": [...], "test": [...], "val": [...]}, ``run_config``, ``dataset_properties``.
The state_dict keys are those of the whole instance model (``model.blocks.<s>.<i>.conv1.kernel`` ...,
``reg_scale_targets`` ...): this build's modules use the same names.  Two layout differences are bridged when loading:
a ``module.`` prefix left by ``nn.DataParallel`` (trainer.py:149-150), and MinkowskiEngine's kernel shape for
kernel_size 1 (``[Cin, Cout]`` vs ``[1, Cin, Cout]`` depending on the stride).
No GPU work happens here (host-side plumbing of the drop-in).
"""
import glob
import os
from typing import Any, Dict, List, Optional, Tuple

import torch


class Checkpoint:
    _LATEST = "latest"

    def __init__(self, checkpoint_file: str):
        self._check_path = checkpoint_file
        self._filled = False
        self.run_config: Optional[Dict] = None
        self.models: Dict[str, Any] = {}
        self.stats: Dict[str, List[Any]] = {"train": [], "test": [], "val": []}
        self.optimizer: Optional[Tuple[str, Any]] = None
        self.grad_scale: Optional[Any] = None
        self.schedulers: Dict[str, Any] = {}
        self.dataset_properties: Dict = {}

    @property
    def path(self):
        return self._check_path

    @property
    def is_empty(self):
        return not self._filled

    def save_objects(self, models_to_save: Dict[str, Any], stage, current_stat, optimizer, schedulers=None,
                     grad_scale=None, **kwargs):
        """model_checkpoint.py:43-58: same keys, same nesting."""
        self.models = models_to_save
        self.optimizer = (optimizer.__class__.__name__, optimizer.state_dict())
        self.schedulers = {name: [getattr(s, "scheduler_opt", None), s.state_dict()]
                           for name, s in (schedulers or {}).items()}
        self.grad_scale = None if grad_scale is None else grad_scale.state_dict()
        if current_stat is not None:
            self.stats.setdefault(stage, []).append(current_stat)
        to_save = dict(kwargs)
        for key, value in self.__dict__.items():
            if not key.startswith("_"):
                to_save[key] = value
        torch.save(to_save, self.path)
        self._filled = True

    @staticmethod
    def load(checkpoint_dir: str, checkpoint_name: str, run_config: Any = None, strict=False, resume=True):
        """model_checkpoint.py:64-92 without the copy into the run directory."""
        checkpoint_file = os.path.join(checkpoint_dir, checkpoint_name) + ".pt"
        ckp = Checkpoint(checkpoint_file)
        if not os.path.exists(checkpoint_file):
            if strict or resume:
                message = "The provided path {} didn't contain the checkpoint_file {}".format(
                    checkpoint_dir, checkpoint_name + ".pt")
                available = glob.glob(os.path.join(checkpoint_dir, "*.pt"))
                if available:
                    message += "\nDid you mean {}?".format(os.path.basename(available[0]))
                raise ValueError(message)
            ckp.run_config = run_config
            return ckp
        objects = torch.load(checkpoint_file, map_location="cpu", weights_only=False)
        for key, value in objects.items():
            setattr(ckp, key, value)
        ckp._filled = True
        return ckp

    def get_state_dict(self, weight_name=_LATEST):
        """model_checkpoint.py:131-148: ``best_<weight_name>`` if present, else ``latest``."""
        if self.is_empty:
            return None
        models = self.models
        key = "best_{}".format(weight_name)
        if key in models:
            return models[key]
        if Checkpoint._LATEST in models:
            return models[Checkpoint._LATEST]
        raise Exception("This weight name isn't within the checkpoint ")

    def load_optim_sched(self, model, load_state=True):
        """model_checkpoint.py:98-124 for the objects ``init_train_objects`` created."""
        if self.is_empty or not load_state:
            return
        if self.optimizer is not None:
            model.optimizer.load_state_dict(self.optimizer[1])
        sch = self.schedulers.get("lr_scheduler") if self.schedulers else None
        if sch is not None and getattr(model, "_lr_scheduler", None) is not None:
            model._lr_scheduler.load_state_dict(sch[1])


def adapt_state_dict(weights: Dict[str, torch.Tensor], model: torch.nn.Module) -> Tuple[Dict[str, torch.Tensor], List[str]]:
    """Keys / shapes of a reference state_dict mapped onto ``model``; returns (loadable weights, unmatched keys)."""
    target = model.state_dict()
    out, unmatched = {}, []
    for k, v in weights.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k not in target:
            unmatched.append(k)
            continue
        t = target[k]
        if v.shape != t.shape:
            if k.endswith(".kernel") and v.numel() == t.numel() and (v.dim(), t.dim()) in ((2, 3), (3, 2)):
                v = v.reshape(t.shape)       # kernel_size 1: [Cin, Cout] <-> [1, Cin, Cout]
            else:
                unmatched.append(k)
                continue
        out[k] = v
    return out, unmatched


def load_reference_weights(model: torch.nn.Module, path_or_checkpoint, weight_name=Checkpoint._LATEST, strict=False):
    """``BaseModel.set_pretrained_weights`` (base_model.py:161-171): loads ``models[weight_name]`` of a checkpoint file
    (or of a ``Checkpoint``), keeping the weights whose names and shapes match.  Returns the unmatched keys."""
    if isinstance(path_or_checkpoint, Checkpoint):
        weights = path_or_checkpoint.get_state_dict(weight_name)
    else:
        if not os.path.exists(path_or_checkpoint):
            raise FileNotFoundError("The path does not exist, it will not load any model")
        models = torch.load(path_or_checkpoint, map_location="cpu", weights_only=False)["models"]
        weights = models["best_{}".format(weight_name)] if "best_{}".format(weight_name) in models else \
            models[weight_name]
    ok, unmatched = adapt_state_dict(weights, model)
    missing = [k for k in model.state_dict() if k not in ok]
    if strict and (unmatched or missing):
        raise RuntimeError(f"checkpoint does not match the model: unmatched {unmatched[:5]}, missing {missing[:5]}")
    model.load_state_dict(ok, strict=False)
    return unmatched
