"""Configuration/logging base class with the interface of the reference's ``mpsfm.baseclass.BaseClass``
(reference mpsfm/baseclass.py:8-51): ``default_conf`` merged with the passed conf, ``_init`` hook,
``log(level=...)`` with optional timers.  omegaconf is optional: plain dicts work."""

from __future__ import annotations

from time import time


class Conf(dict):
    """dict with attribute access (stands in for an OmegaConf node; accepts one as input)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_conf(obj) -> Conf:
    if obj is None:
        return Conf()
    if hasattr(obj, "items") and not isinstance(obj, dict):
        try:  # OmegaConf DictConfig
            from omegaconf import OmegaConf

            obj = OmegaConf.to_container(obj, resolve=True)
        except Exception:  # noqa: BLE001
            obj = dict(obj.items())
    out = Conf()
    for k, v in dict(obj).items():
        out[k] = to_conf(v) if isinstance(v, dict) or (hasattr(v, "items") and not isinstance(v, (str, bytes))) else v
    return out


class BaseClass:
    freeze_conf = True
    default_conf = {"verbose": 0}

    def __init__(self, conf=None, *args, **kwargs):
        default = to_conf(self.default_conf)
        passed = to_conf(conf)
        if self.freeze_conf:
            unknown = [k for k in passed if k not in default]
            if unknown:
                raise KeyError(f"unknown configuration key(s) {unknown} for {type(self).__name__}")
        merged = Conf(default)
        merged.update(passed)
        self.conf = merged
        self._assert_configs()
        self._propagate_conf()
        self._init(*args, **kwargs)
        self.tstart = None

    def _init(self, *args, **kwargs):
        pass

    def _assert_configs(self):
        pass

    def _propagate_conf(self):
        pass

    def log(self, *message, level=0, tstart=False, tend=False, **kwargs):
        if self.conf.verbose >= level:
            if tstart:
                self.tstart = time()
            elif tend:
                assert len(message) == 0 and self.tstart is not None
                message = [f"{time() - self.tstart:.3f} s"]
            print(*message, end=" " if tstart else "\n", **kwargs)
