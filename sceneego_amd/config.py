"""YAML config surface of the hot path.

Mirrors the reference's ``utils/cfg.py:5-9`` (``load_config(path) -> EasyDict``).  The reference
depends on the third-party ``easydict`` package; it is not in this image, so the attribute-dict
behaviour the model relies on (nested dicts readable as attributes, ``config.model.volume_size``)
is restated here in ~30 lines.  Keys read by the path are listed in SURVEY.md §5 "Config".
"""
from __future__ import annotations

import os

import yaml

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONFIG = os.path.join(os.path.dirname(_PKG_DIR), "experiments", "sceneego", "test", "sceneego.yaml")


class EasyDict(dict):
    """dict whose items are also attributes; nested dicts (also inside lists) are wrapped recursively."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setattr__(self, name, value):
        value = self._wrap(value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__

    def __delattr__(self, name):
        super().__delattr__(name)
        super().__delitem__(name)

    def update(self, e=None, **f):
        d = dict(e or {})
        d.update(f)
        for k, v in d.items():
            setattr(self, k, v)


def load_config(path: str = DEFAULT_CONFIG) -> EasyDict:
    """Same contract as the reference's ``utils/cfg.py:5-9``."""
    with open(path) as fin:
        return EasyDict(yaml.safe_load(fin))


def resolve_calibration_path(path: str) -> str:
    """The reference's YAML holds a path relative to its repo root (``sceneego.yaml:72``).

    Try it as given, then relative to this repo's root, then fall back to the calibration file
    shipped inside the package (same numbers, see ``calibration/README``).
    """
    if os.path.isfile(path):
        return path
    root = os.path.dirname(_PKG_DIR)
    for cand in (os.path.join(root, path), os.path.join(_PKG_DIR, "calibration", os.path.basename(path))):
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError(f"camera calibration file not found: {path}")
