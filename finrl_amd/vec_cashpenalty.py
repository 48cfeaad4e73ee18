"""Device-resident batch of the reference's StockTradingEnvCashpenalty
(finrl/meta/env_stock_trading/env_stocktrading_cashpenalty.py:19-409), one HIP launch per step
through the C ABI (finenv_cashpenalty_*)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat
from .spaces import Box


class CashPenaltyPanel:
    """close [T,N] f64, info [T,N,C] f64 (daily_information_cols per asset, ticker-major as
    get_date_vector builds them, :159-171), turb [T]."""

    def __init__(self, close, info, turb=None, dates=None, assets=None):
        self.close = np.ascontiguousarray(close, dtype=np.float64)
        T, N = self.close.shape
        self.info = np.ascontiguousarray(info, dtype=np.float64).reshape(T, N, -1)
        self.turb = np.ascontiguousarray(np.zeros(T) if turb is None else turb, np.float64)
        self.T, self.N, self.C = T, N, self.info.shape[2]
        self.D = 1 + N + N * self.C
        self.dates = list(dates) if dates is not None else list(range(T))
        self.assets = list(assets) if assets is not None else [f"A{i}" for i in range(N)]

    @classmethod
    def from_dataframe(cls, df, daily_information_cols, date_col_name="date"):
        """assets = df.tic.unique() (order of appearance), dates = sorted unique (:70-71)."""
        assets = list(dict.fromkeys(df["tic"].tolist()))
        dates = sorted(df[date_col_name].unique().tolist())
        T, N = len(dates), len(assets)
        piv = df.set_index([date_col_name, "tic"]).sort_index()
        idx = [(d, a) for d in dates for a in assets]
        sub = piv.loc[idx]
        close = sub["close"].to_numpy(np.float64).reshape(T, N)
        info = sub[list(daily_information_cols)].to_numpy(np.float64).reshape(T, N, -1)
        turb = sub["turbulence"].to_numpy(np.float64).reshape(T, N)[:, 0] \
            if "turbulence" in sub.columns else None
        return cls(close, info, turb, dates, assets)


class VecCashPenaltyEnv:
    env_name = "StockTradingEnvCashpenalty-MI355X"
    if_discrete = False
    _kind = "cashpenalty"                      # finenv_<kind>_* entry points
    _cfg_cls, _panel_cls, _state_cls = nat.CashPenaltyConfig, nat.CashPenaltyPanelPtrs, \
        nat.CashPenaltyStatePtrs
    _f64_fields, _i32_fields = nat.CASHPENALTY_F64_FIELDS, nat.CASHPENALTY_I32_FIELDS
    _books = ("holdings",)                     # [N][E] f64 blocks after the scalar rows

    def _extra_cfg(self, **kw):
        if kw:
            raise TypeError(f"unexpected arguments {sorted(kw)}")
        return ()

    def _fn(self, name):
        return getattr(nat.lib(), f"finenv_{self._kind}_{name}")

    def __init__(self, panel: CashPenaltyPanel, num_envs, *, buy_cost_pct=3e-3, sell_cost_pct=3e-3,
                 hmax=10, discrete_actions=False, shares_increment=1, turbulence_threshold=None,
                 initial_amount=1e6, cash_penalty_proportion=0.1, random_start=True, patient=False,
                 auto_reset=True, device="cuda", seed=0, **extra):
        import torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.FinenvError("finrl_amd has no CPU path: device must be a HIP GPU")
        self.panel = panel
        E, N, Cc, T = int(num_envs), panel.N, panel.C, panel.T
        self.num_envs = self.env_num = E
        self.action_dim = N
        self.state_dim = self.state_space = panel.D
        self.random_start = bool(random_start)
        self.auto_reset = bool(auto_reset)
        self.observation_space = Box(-np.inf, np.inf, (panel.D,), np.float32)
        self.action_space = Box(-1, 1, (N,), np.float32)
        self._seed, self._rs_on_device = int(seed) & (2 ** 63 - 1), False
        L = nat.lib()
        self._cfg = self._cfg_cls(
            E, N, Cc, T, int(discrete_actions), int(shares_increment),
            int(turbulence_threshold is not None), int(patient), float(hmax), float(buy_cost_pct),
            float(sell_cost_pct), float(initial_amount), float(cash_penalty_proportion),
            float(turbulence_threshold if turbulence_threshold is not None else 0.0),
            *self._extra_cfg(**extra))
        self._h = C.c_void_p()
        nat.check(self._fn("create")(C.byref(self._cfg), C.byref(self._h)), None,
                  f"finenv_{self._kind}_create")
        dev = self.device
        self._close = torch.from_numpy(panel.close).to(dev)
        self._info = torch.from_numpy(panel.info.reshape(T, N * Cc).astype(np.float32)).to(dev)
        self._turb = torch.from_numpy(panel.turb).to(dev)
        nf, ni = len(self._f64_fields), len(self._i32_fields)
        self._f64 = torch.zeros(nf + len(self._books) * N, E, dtype=torch.float64, device=dev)
        self._i32 = torch.zeros(ni, E, dtype=torch.int32, device=dev)
        self.state = {k: self._f64[j] for j, k in enumerate(self._f64_fields)}
        self.state.update({k: self._i32[j] for j, k in enumerate(self._i32_fields)})
        for b, k in enumerate(self._books):
            self.state[k] = self._f64[nf + b * N:nf + (b + 1) * N]
        self.state["episode"].fill_(-1)                                         # :98
        pp = self._panel_cls(self._close.data_ptr(), self._info.data_ptr(),
                             self._turb.data_ptr())
        sp = self._state_cls(self._f64.data_ptr(), self._i32.data_ptr())
        nat.check(self._fn("bind")(self._h, C.byref(pp), C.byref(sp)), self._h, "bind",
                  self._kind)
        self.obs = torch.zeros(E, panel.D, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(E, dtype=torch.float32, device=dev)
        self.done = torch.zeros(E, dtype=torch.uint8, device=dev)
        self.term_obs = None
        self.audit = None

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._fn("destroy")(self._h)
                self._h = None
        except Exception:
            pass

    def enable_terminal_obs(self):
        import torch
        if self.term_obs is None:
            self.term_obs = torch.zeros_like(self.obs)
        return self.term_obs

    def enable_audit(self):
        """Per-step harness log rows [E, AUDIT_HEAD + N] f64 (begin cash, asset value, reward,
        reason flags, applied transactions): what the single-env facade appends to the reference's
        account_information / transaction_memory lists.  Off by default (no extra traffic)."""
        import torch
        if getattr(self, "audit", None) is None:
            self.audit = torch.zeros(self.num_envs, nat.AUDIT_HEAD + self.action_dim,
                                     dtype=torch.float64, device=self.device)
            nat.check(self._fn("set_audit")(self._h, C.c_void_p(self.audit.data_ptr())), self._h,
                      "set_audit", self._kind)
        return self.audit

    def set_next_start(self, starts):
        """Starting points the next reset of each env will use (the reference draws
        random.choice(range(int(len(dates) * 0.5))), :134-138)."""
        import torch
        self.state["next_start"].copy_(torch.as_tensor(
            np.broadcast_to(np.asarray(starts, np.int32), (self.num_envs,)).copy()))

    def _draw_starts(self):
        """random_start: resets draw their starting point on the device (set once; no host work
        per step).  `set_next_start()` + random_start=False pins them for reproducible runs."""
        if not self._rs_on_device:
            hi = max(1, int(self.panel.T * 0.5))                                   # :134-138
            nat.check(self._fn("set_random_start")(self._h, hi, int(self._seed)), self._h,
                      "set_random_start", self._kind)
            self._rs_on_device = True

    def reset(self, mask=None):
        import torch
        if self.random_start:
            self._draw_starts()
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        nat.check(self._fn("reset")(
            self._h, mptr, C.c_void_p(self.obs.data_ptr()), self._stream()), self._h, "reset",
            self._kind)
        return self.obs

    def step(self, actions, out=None):
        import torch
        if actions.dtype != torch.float32 or not actions.is_contiguous() or \
                actions.device != self.obs.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if self.random_start and self.auto_reset:
            self._draw_starts()          # fresh starting points for envs that end this step
        obs, rew, done = out if out is not None else (self.obs, self.reward, self.done)
        nat.check(self._fn("step")(
            self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
            C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
            C.c_void_p(self.term_obs.data_ptr()) if self.term_obs is not None else None,
            int(self.auto_reset), self._stream()), self._h, "step", self._kind)
        return obs, rew, done, None

    def as_sb3_vec_env(self):
        """stable-baselines3 VecEnv-shaped view (numpy in / out, auto-reset, terminal_observation)."""
        from .vec_env import SB3VecEnvAdapter
        return SB3VecEnvAdapter(self)

    def episode_return(self):
        """last logged total assets / initial amount per env (the GainLoss figure, :176), f32."""
        import torch
        return (self.state["logged_total"] / float(self._cfg.initial_amount)).to(torch.float32)

    def state_numpy(self):
        out = {k: v.detach().cpu().numpy() for k, v in self.state.items()}
        for k in self._books:
            out[k] = np.ascontiguousarray(out[k].T)
        return out


class VecStopLossEnv(VecCashPenaltyEnv):
    """Device-resident batch of the reference's StockTradingEnvStopLoss
    (finrl/meta/env_stock_trading/env_stocktrading_stoploss.py:19-459); extra kwargs
    stoploss_penalty (:73) and profit_loss_ratio (:74)."""
    env_name = "StockTradingEnvStopLoss-MI355X"
    _kind = "stoploss"
    _cfg_cls, _panel_cls, _state_cls = nat.StopLossConfig, nat.StopLossPanelPtrs, \
        nat.StopLossStatePtrs
    _f64_fields, _i32_fields = nat.STOPLOSS_F64_FIELDS, nat.STOPLOSS_I32_FIELDS
    _books = nat.STOPLOSS_BOOKS

    def _extra_cfg(self, stoploss_penalty=0.9, profit_loss_ratio=2):
        min_profit_penalty = 1 + profit_loss_ratio * (1 - stoploss_penalty)      # :101
        return float(stoploss_penalty), float(min_profit_penalty)
