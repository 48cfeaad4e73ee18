"""Reference-import harness (TEST INFRASTRUCTURE, container-only).

Imports the *unmodified* reference envs from ``/root/reference`` so that
(i) the CPU restatement in ``oracle/`` can be validated against them and
(ii) golden vectors can be generated (``tests/golden/make_golden.py``).

The reference never travels to the GPU box: nothing in ``-m gpu`` tests,
``smoke()`` or ``bench.py`` imports this module.  Importing it where
``/root/reference`` is absent raises ``ReferenceUnavailable``.

What it does (description only -- no reference source is
copied here):
  * registers a bare ``finrl`` package object whose ``__path__`` points at the
    reference tree, so ``finrl/__init__.py`` (which pulls broker SDKs) is never
    executed;
  * registers minimal stand-ins for ``gym`` / ``gym.spaces`` /
    ``gym.utils.seeding`` / ``stable_baselines3.common.vec_env`` /
    ``stable_baselines3.common.logger`` (none is installed in this image); the
    env arithmetic itself only uses numpy + pandas;
  * offers ``stable_argsort_patch(module)``: swaps the module-level ``np`` name
    for a proxy whose ``argsort`` defaults to ``kind="stable"`` (O-stable
    oracle) without touching reference files.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("FINRL_REFERENCE_ROOT", "/root/reference")


class ReferenceUnavailable(RuntimeError):
    pass


class _Box:
    """Stand-in for gym.spaces.Box: just remembers its arguments."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = low
        self.high = high
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.dtype = np.dtype(dtype)


class _DummyVecEnv:
    """Behavioural stand-in for SB3 ``DummyVecEnv`` (documented public behaviour):
    float32 obs/reward buffers, auto-reset on done with
    ``info['terminal_observation']``."""

    def __init__(self, env_fns):
        self.envs = [fn() for fn in env_fns]
        self.num_envs = len(self.envs)
        env = self.envs[0]
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        shp = tuple(self.observation_space.shape)
        self.buf_obs = np.zeros((self.num_envs,) + shp, dtype=np.float32)
        self.buf_rews = np.zeros((self.num_envs,), dtype=np.float32)
        self.buf_dones = np.zeros((self.num_envs,), dtype=bool)
        self.buf_infos = [{} for _ in range(self.num_envs)]
        self.actions = None

    def reset(self):
        for i, e in enumerate(self.envs):
            self.buf_obs[i] = np.asarray(e.reset(), dtype=np.float32)
        return self.buf_obs.copy()

    def step_async(self, actions):
        self.actions = actions

    def step_wait(self):
        for i, e in enumerate(self.envs):
            obs, self.buf_rews[i], self.buf_dones[i], info = e.step(self.actions[i])
            info = dict(info) if info else {}
            if self.buf_dones[i]:
                info["terminal_observation"] = np.asarray(obs, dtype=np.float32)
                obs = e.reset()
            self.buf_infos[i] = info
            self.buf_obs[i] = np.asarray(obs, dtype=np.float32)
        return (self.buf_obs.copy(), self.buf_rews.copy(), self.buf_dones.copy(),
                [dict(x) for x in self.buf_infos])

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        # SB3's public signature; the reference calls it with method_name= (models.py:120-121)
        return [getattr(e, method_name)(*method_args, **method_kwargs) for e in self.envs]

    def render(self, mode="human"):
        # SB3 DummyVecEnv.render: a single env renders itself (models.py:321 relies on it)
        return self.envs[0].render(mode=mode)

    def close(self):
        pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Idempotently register the stub modules and the bare ``finrl`` package."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "finrl")):
        raise ReferenceUnavailable(
            f"{REFERENCE_ROOT}/finrl not present: the reference only exists in the "
            "build container; use the committed fixtures under tests/golden/ instead")
    import matplotlib
    matplotlib.use("Agg")

    if "gym" not in sys.modules:
        spaces = _mod("gym.spaces", Box=_Box)
        seeding = _mod("gym.utils.seeding",
                       np_random=lambda seed=None: (np.random.RandomState(seed), seed))
        utils = _mod("gym.utils", seeding=seeding)
        logger = _mod("gym.logger", set_level=lambda *_a, **_k: None)
        _mod("gym", Env=type("Env", (), {}), spaces=spaces, utils=utils, logger=logger)
    if "stable_baselines3" not in sys.modules:
        vec_env = _mod("stable_baselines3.common.vec_env",
                       DummyVecEnv=_DummyVecEnv, SubprocVecEnv=_DummyVecEnv)
        sb_logger = _mod("stable_baselines3.common.logger",
                         record=lambda *_a, **_k: None)
        common = _mod("stable_baselines3.common", vec_env=vec_env, logger=sb_logger)
        _mod("stable_baselines3", common=common)

    pkg = types.ModuleType("finrl")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "finrl")]
    sys.modules["finrl"] = pkg
    _installed = True


def _fresh_import(modname):
    install()
    sys.modules.pop(modname, None)
    return importlib.import_module(modname)


def load_stocktrading():
    """-> module finrl.meta.env_stock_trading.env_stocktrading (unmodified)."""
    return _fresh_import("finrl.meta.env_stock_trading.env_stocktrading")


def load_stocktrading_np():
    return _fresh_import("finrl.meta.env_stock_trading.env_stocktrading_np")


def load_portfolio():
    return _fresh_import("finrl.meta.env_portfolio_allocation.env_portfolio")


def load_multiple_crypto():
    """env_multiple_crypto imports both agent wrappers and DataProcessor at module
    top (env_multiple_crypto.py:3-5); satisfy those three names with dummies."""
    install()
    for name in ("finrl.agents", "finrl.agents.elegantrl", "finrl.agents.elegantrl.models",
                 "finrl.agents.stablebaselines3", "finrl.agents.stablebaselines3.models"):
        if name not in sys.modules:
            _mod(name, DRLAgent=object)
    if "finrl.meta.data_processor" not in sys.modules:
        _mod("finrl.meta.data_processor", DataProcessor=object)
    return _fresh_import("finrl.meta.env_cryptocurrency_trading.env_multiple_crypto")


class _StableNumpy:
    """numpy proxy: identical to numpy except argsort defaults to kind='stable'."""

    def __init__(self, real):
        object.__setattr__(self, "_real", real)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_real"), name)

    def argsort(self, a, *args, **kwargs):
        if len(args) < 2 and "kind" not in kwargs:
            kwargs["kind"] = "stable"
        return object.__getattribute__(self, "_real").argsort(a, *args, **kwargs)


def stable_argsort_patch(module):
    """O-stable oracle: make the env module's ``np.argsort`` stable."""
    module.np = _StableNumpy(np)
    return module


class _LenientPyplot:
    """matplotlib.pyplot proxy whose ``savefig`` drops the ``index=`` keyword the reference passes
    (env_stocktrading.py:284-289): matplotlib releases of the reference's day ignored unknown
    savefig keywords, 3.10 (this image) raises TypeError -- after the three CSV files of the
    terminal branch are written, before ``step`` returns."""

    def __init__(self, real):
        object.__setattr__(self, "_real", real)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_real"), name)

    def savefig(self, *args, **kwargs):
        kwargs.pop("index", None)
        return object.__getattribute__(self, "_real").savefig(*args, **kwargs)


def lenient_savefig_patch(module):
    """Harness-side name substitution (reference files untouched), like stable_argsort_patch."""
    module.plt = _LenientPyplot(module.plt)
    return module


def make_stock_frame(close, tech, risk, tech_names, risk_col="turbulence", tickers=None,
                     dates=None):
    """Build the DataFrame the reference env expects:
    rows sorted by (date, tic); index = day ordinal (preprocessors.py:31-32 contract);
    columns date, tic, close, <tech names>, <risk col>.

    close: [T,N] float64, tech: [T,K,N] float64, risk: [T] float64.
    """
    import pandas as pd
    T, N = close.shape
    K = tech.shape[1]
    if tickers is None:
        tickers = [f"TIC{i:03d}" for i in range(N)]
    if dates is None:
        dates = [f"2000-{1 + (t // 28) % 12:02d}-{1 + t % 28:02d}#{t:05d}" for t in range(T)]
    cols = {
        "date": np.repeat(np.asarray(dates, dtype=object), N),
        "tic": np.tile(np.asarray(tickers, dtype=object), T),
        "close": np.asarray(close, dtype=np.float64).reshape(-1),
    }
    for k, name in enumerate(tech_names):
        cols[name] = np.asarray(tech[:, k, :], dtype=np.float64).reshape(-1)
    cols[risk_col] = np.repeat(np.asarray(risk, dtype=np.float64), N)
    df = pd.DataFrame(cols)
    df.index = np.repeat(np.arange(T), N)
    assert K == len(tech_names)
    return df
