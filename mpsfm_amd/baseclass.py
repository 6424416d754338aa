"""Configuration/logging base class with the interface of the reference's ``mpsfm.baseclass.BaseClass``
(reference mpsfm/baseclass.py:8-51): ``default_conf`` merged with the passed conf, ``_init`` hook,
``log(level=...)`` with optional timers.  omegaconf is optional: plain dicts work."""

from __future__ import annotations

from time import perf_counter


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


class _Stopwatch:
    """One running interval; `log(tstart=True)` arms it, `log(tend=True)` reads it."""

    def __init__(self):
        self.t0 = None

    def arm(self):
        self.t0 = perf_counter()

    def read(self) -> float:
        if self.t0 is None:
            raise AssertionError("log(tend=True) without a preceding log(tstart=True)")
        return perf_counter() - self.t0


class BaseClass:
    """Hook order of the reference's base class (mpsfm/baseclass.py:16-29): merge the configuration, then
    `_assert_configs`, `_propagate_conf`, `_init(*args, **kwargs)`."""

    freeze_conf = True
    default_conf = {"verbose": 0}

    def __init__(self, conf=None, *args, **kwargs):
        self.conf = self._merged_conf(conf)
        self._watch = _Stopwatch()
        for hook in (self._assert_configs, self._propagate_conf):
            hook()
        self._init(*args, **kwargs)

    @classmethod
    def _merged_conf(cls, conf) -> Conf:
        base, given = to_conf(cls.default_conf), to_conf(conf)
        if cls.freeze_conf:
            extra = sorted(set(given) - set(base))
            if extra:
                raise KeyError(f"unknown configuration key(s) {extra} for {cls.__name__}")
        base.update(given)
        return base

    # hooks for subclasses
    def _init(self, *args, **kwargs):
        return None

    def _assert_configs(self):
        return None

    def _propagate_conf(self):
        return None

    @property
    def tstart(self):
        return self._watch.t0

    def log(self, *message, level=0, tstart=False, tend=False, **kwargs):
        """Prints when conf.verbose >= level.  tstart: print without a newline and start the stopwatch;
        tend (no message allowed): print the elapsed seconds."""
        if self.conf.verbose < level:
            return
        if tstart:
            self._watch.arm()
            print(*message, end=" ", **kwargs)
            return
        if tend:
            if message:
                raise AssertionError("log(tend=True) takes no message")
            message = (f"{self._watch.read():.3f} s",)
        print(*message, **kwargs)
