"""`.tckpt` checkpoints with a `checkpoints.json` index (reference: rslo/torchplus/train/checkpoint.py:24-218).

On-disk contract (what a run directory written by the reference looks like, and what this module reads/writes):
  <model_dir>/<name>-<global_step>.tckpt     torch.save(model.state_dict())
  <model_dir>/checkpoints.json               {"latest_ckpt": {name: file}, "all_ckpts": {name: [file, ...]}}
`name` is the `.name` attribute of the module / optimizer ("voxelnet", "adam_optimizer").
The index is rewritten atomically (temp file + rename) and SIGINT is deferred while files are written.
"""
import json
import os
import signal
import threading
from pathlib import Path

import torch

INDEX = "checkpoints.json"


class _DeferSigint:
    """Hold back Ctrl-C until a checkpoint is completely on disk."""

    def __enter__(self):
        self.pending = None
        self.armed = threading.current_thread() is threading.main_thread()
        if self.armed:
            self.prev = signal.signal(signal.SIGINT, lambda sig, frame: setattr(self, "pending", (sig, frame)))
        return self

    def __exit__(self, *exc):
        if self.armed:
            signal.signal(signal.SIGINT, self.prev)
            if self.pending is not None and callable(self.prev):
                self.prev(*self.pending)


def _read_index(model_dir):
    p = Path(model_dir) / INDEX
    if not p.is_file():
        return {"latest_ckpt": {}, "all_ckpts": {}}
    return json.loads(p.read_text())


def _write_index(model_dir, index):
    p = Path(model_dir) / INDEX
    tmp = p.with_suffix(".json.tmp")
    tmp.write_text(json.dumps(index, indent=2))
    os.replace(tmp, p)


def _file_name(name, step):
    return "%s-%s.tckpt" % (name, step)


def _step_of(file_name):
    return int(Path(file_name).name.split(".")[0].split("-")[-1])


def latest_checkpoint(model_dir, model_name):
    """Path of the newest checkpoint of `model_name`, or None."""
    latest = _read_index(model_dir)["latest_ckpt"].get(model_name)
    if latest is None:
        return None
    path = Path(model_dir) / latest
    return str(path) if path.is_file() else None


def _to_host(obj):
    """Copy of a (nested) state dict with every tensor on the host; the live objects stay where they are."""
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        out = type(obj)((k, _to_host(v)) for k, v in obj.items())
        meta = getattr(obj, "_metadata", None)      # module.state_dict(): per-module version info, an ATTRIBUTE of the
        if meta is not None:                        # OrderedDict -- the reference saves model.cpu().state_dict() with it
            out._metadata = meta
        return out
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v) for v in obj)
    return obj


def save(model_dir, model, model_name, global_step, max_to_keep=8, keep_latest=True, host_copy=False):
    """Writes `<model_name>-<global_step>.tckpt` and updates the index; beyond `max_to_keep` files the oldest
    entry (keep_latest) or the smallest step is deleted.  host_copy: serialise CPU copies of the state."""
    with _DeferSigint():
        index = _read_index(model_dir)
        fname = _file_name(model_name, global_step)
        state = model.state_dict()
        torch.save(_to_host(state) if host_copy else state, str(Path(model_dir) / fname))
        index["latest_ckpt"][model_name] = fname
        known = index["all_ckpts"].get(model_name, []) + [fname]
        alive = []
        for f in known:                       # drop entries whose file vanished, keep order, no duplicates
            if f not in alive and (Path(model_dir) / f).is_file():
                alive.append(f)
        if len(alive) > max_to_keep:
            victim = alive[0] if keep_latest else min(alive, key=_step_of)
            alive.remove(victim)
            os.remove(str(Path(model_dir) / victim))
        index["all_ckpts"][model_name] = alive
        _write_index(model_dir, index)


def restore(ckpt_path, model, map_func=None, map_location="cpu"):
    if not Path(ckpt_path).is_file():
        raise ValueError("checkpoint {} not exist.".format(ckpt_path))
    # the reference's plain torch.load (checkpoint.py:118-126, torch 1.2): a .tckpt of the OPTIMIZER carries what the
    # schedule wrote into its param groups (numpy scalars from the OneCycle interpolation), which torch >= 2.6's default
    # weights_only unpickler refuses; try the restricted loader first, fall back for the user's own checkpoint files
    # restricted loader with the numpy scalar types the schedule leaves in the optimizer's param groups allow-listed; only
    # when THAT still refuses (a type outside the list: pickle.UnpicklingError) the file is unpickled in full like the
    # reference does -- with a warning, since a full unpickle runs whatever the file says
    import pickle
    import warnings
    import numpy as np
    safe = [np.dtype, np.ndarray, np.float64, np.float32, np.int64, np.int32, np.bool_]
    for name in ("core.multiarray.scalar", "_core.multiarray.scalar"):
        mod, _, attr = ("numpy." + name).rpartition(".")
        try:
            safe.append(getattr(__import__(mod, fromlist=[attr]), attr))
        except (ImportError, AttributeError):
            pass
    safe += [type(np.dtype(t)) for t in ("float64", "float32", "int64", "int32", "bool")]
    try:
        with torch.serialization.safe_globals(safe):
            state = torch.load(ckpt_path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        warnings.warn("checkpoint %s holds objects outside the weights-only allow-list (%s): unpickling it in full, as the "
                      "reference's torch.load does -- only do this with files you trust" % (ckpt_path, str(e).splitlines()[0]))
        state = torch.load(ckpt_path, map_location=map_location, weights_only=False)
    if map_func is not None:
        state = map_func(state)
    model.load_state_dict(state)
    print("Restoring parameters from {}".format(ckpt_path))


def _named(models):
    if isinstance(models, dict):
        return dict(models)
    names = []
    for m in models:
        if not hasattr(m, "name"):
            raise ValueError("models must have name attr")
        names.append(m.name)
    if len(set(names)) != len(names):
        raise ValueError("models must have unique name: {}".format(", ".join(names)))
    return {m.name: m for m in models}


def try_restore_latest_checkpoints(model_dir, models, map_func=None, map_location="cpu"):
    for name, model in _named(models).items():
        path = latest_checkpoint(model_dir, name)
        if path is not None:
            restore(path, model, map_func, map_location)


def restore_latest_checkpoints(model_dir, models, map_func=None, map_location="cpu"):
    for name, model in _named(models).items():
        path = latest_checkpoint(model_dir, name)
        if path is None:
            raise ValueError("model {}'s ckpt isn't exist".format(name))
        restore(path, model, map_func, map_location)


def restore_models(model_dir, models, global_step, map_func=None, map_location="cpu"):
    for name, model in _named(models).items():
        restore(str(Path(model_dir) / _file_name(name, global_step)), model, map_func, map_location)


def save_models(model_dir, models, global_step, max_to_keep=15, keep_latest=True, host_copy=False):
    with _DeferSigint():
        for name, model in _named(models).items():
            save(model_dir, model, name, global_step, max_to_keep, keep_latest, host_copy=host_copy)


def save_models_cpu(model_dir, models, global_step, max_to_keep=15, keep_latest=True):
    """The reference moves network + optimizer state to the host, saves, and moves back
    (checkpoint.py:178-218), so its .tckpt files hold CPU tensors and load anywhere without map_location.  Same
    guarantee here from host COPIES of the state dicts (network and optimizer, incl. device-resident Adam step
    counters); the live model / optimizer are not moved back and forth."""
    save_models(model_dir, models, global_step, max_to_keep, keep_latest, host_copy=True)
