"""Checkpoint save / filtered reload with the reference's file layout and reload rule.

Reference: ``JobRunner.save_model`` / ``update_model_state`` (job_runner.py:333-350) write one ``torch.save`` dict
``{iteration, epoch_n, model_dict, optimizer_dict, scheduler_dict, metric[, amp]}``; ``load_pretrained_model``
(job_runner.py:85-123) reloads the objects named in ``RELOAD_DICT_LIST`` and, for everything but "metric", keeps only
entries whose KEY exists in the live object's state_dict and whose tensor SIZE matches it (silently skipping the rest),
then ``load_state_dict``s the merged dict.  The layers in :mod:`spgnn_amd.nn` carry DGL's state_dict keys, so a GNN
checkpoint written by the reference (DGL 0.6: GATConv without ``bias``; DGL >= 0.7: with it) loads through this rule.
"""
from __future__ import annotations

import logging
from typing import Dict, Iterable, List, Optional, Sequence

import torch

__all__ = ["filter_state", "reload_state", "load_pretrained_model", "make_states", "save_states"]

log = logging.getLogger(__name__)


def filter_state(current: Dict[str, torch.Tensor], saved: Dict[str, torch.Tensor], ignored_keys: Iterable[str] = ()) -> Dict:
    """Entries of ``saved`` that may overwrite ``current``: known key, not ignored, same tensor size."""
    ignored = set(ignored_keys)
    kept = {}
    for k, v in saved.items():
        if k not in current:
            continue
        if k in ignored:
            log.info("ignore key: %s", k)
            continue
        cv = current[k]
        if isinstance(cv, torch.Tensor) and torch.is_tensor(v) and v.size() != cv.size():
            log.info("in %s, saved tensor size %s does not match current tensor size %s", k, tuple(v.size()), tuple(cv.size()))
            continue
        kept[k] = v
    return kept


def reload_state(obj, saved: Dict, overwrite: bool = False, ignored_keys: Iterable[str] = ()) -> List[str]:
    """Merge ``saved`` into ``obj.state_dict()`` (filtered unless ``overwrite``) and load it; returns the keys taken."""
    if hasattr(obj, "bucket") and hasattr(obj, "load_state_dict") and "param_groups" in saved:
        obj.load_state_dict(saved)               # train.TrainStep: an optimizer-style state (its own or torch.optim.SGD's)
        return sorted(saved)
    current = obj.state_dict()
    matched = dict(saved) if overwrite else filter_state(current, saved, ignored_keys)
    current.update(matched)
    obj.load_state_dict(current)
    return sorted(matched)


def _load_file(cpk_path, map_location, trusted: bool):
    """``torch.load`` restricted to tensors and plain containers (``weights_only=True``).  A checkpoint that pickles other
    objects (the reference stores its metric object's ``state_dict``, plain data, but older files may hold more) is only
    unpickled in full when the caller says the file is ``trusted``: unpickling runs arbitrary code from the file."""
    import pickle
    try:
        return torch.load(cpk_path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:      # something outside the weights_only allow-list; a missing or truncated file, a
        if not trusted:                      # bad map_location etc. (OSError / RuntimeError / EOFError) propagate as they are
            raise RuntimeError(f"{cpk_path}: the checkpoint holds objects beyond tensors and plain containers ({e}); pass "
                               "trusted=True to unpickle it in full (only for files you wrote yourself), or allow-list "
                               "its classes with torch.serialization.add_safe_globals") from e
        log.warning("%s: falling back to a full unpickle (trusted=True)", cpk_path)
        return torch.load(cpk_path, map_location=map_location, weights_only=False)


def load_pretrained_model(cpk_path, reload_objects: Sequence, state_keys: Sequence[str], ignored_keys: Iterable[str] = (),
                          device: str = "cuda", trusted: bool = False) -> Dict:
    """Same contract as the reference function of this name: ``reload_objects[n]`` is restored from
    ``saved_states[state_keys[n]]`` when that key exists ("metric" is taken as it is, everything else filtered).
    ``reload_objects`` may hold a :class:`spgnn_amd.train.TrainStep` under "optimizer_dict": it takes a
    ``torch.optim.SGD`` state dict (the reference's) as well as its own.  ``trusted``: see :func:`_load_file`."""
    saved_states = _load_file(cpk_path, "cpu" if device == "cpu" else None, trusted)
    for obj, key in zip(reload_objects, state_keys):
        if key in saved_states:
            reload_state(obj, saved_states[key], overwrite=(key == "metric"), ignored_keys=ignored_keys)
    return saved_states


def make_states(model, optimizer=None, scheduler=None, metric=None, iteration: int = 0, epoch_n: int = 0, **extra) -> Dict:
    """The dict ``update_model_state`` builds (job_runner.py:333-343)."""
    states = {"iteration": iteration, "epoch_n": epoch_n, "model_dict": model.state_dict()}
    if optimizer is not None:
        states["optimizer_dict"] = optimizer.state_dict()
    if scheduler is not None:
        states["scheduler_dict"] = scheduler.state_dict()
    if metric is not None:
        states["metric"] = metric.state_dict()
    states.update(extra)
    return states


def save_states(path, states: Dict) -> None:
    torch.save(states, path)
