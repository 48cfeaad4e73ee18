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

    def __init__(self, panel: CashPenaltyPanel, num_envs, *, buy_cost_pct=3e-3, sell_cost_pct=3e-3,
                 hmax=10, discrete_actions=False, shares_increment=1, turbulence_threshold=None,
                 initial_amount=1e6, cash_penalty_proportion=0.1, random_start=True, patient=False,
                 auto_reset=True, device="cuda", seed=0):
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
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        L = nat.lib()
        self._cfg = nat.CashPenaltyConfig(
            E, N, Cc, T, int(discrete_actions), int(shares_increment),
            int(turbulence_threshold is not None), int(patient), float(hmax), float(buy_cost_pct),
            float(sell_cost_pct), float(initial_amount), float(cash_penalty_proportion),
            float(turbulence_threshold if turbulence_threshold is not None else 0.0))
        self._h = C.c_void_p()
        nat.check(L.finenv_cashpenalty_create(C.byref(self._cfg), C.byref(self._h)), None,
                  "finenv_cashpenalty_create")
        dev = self.device
        self._close = torch.from_numpy(panel.close).to(dev)
        self._info = torch.from_numpy(panel.info.reshape(T, N * Cc).astype(np.float32)).to(dev)
        self._turb = torch.from_numpy(panel.turb).to(dev)
        nf, ni = len(nat.CASHPENALTY_F64_FIELDS), len(nat.CASHPENALTY_I32_FIELDS)
        self._f64 = torch.zeros(nf + N, E, dtype=torch.float64, device=dev)
        self._i32 = torch.zeros(ni, E, dtype=torch.int32, device=dev)
        self.state = {k: self._f64[j] for j, k in enumerate(nat.CASHPENALTY_F64_FIELDS)}
        self.state.update({k: self._i32[j] for j, k in enumerate(nat.CASHPENALTY_I32_FIELDS)})
        self.state["holdings"] = self._f64[nf:nf + N]
        self.state["episode"].fill_(-1)                                         # :98
        pp = nat.CashPenaltyPanelPtrs(self._close.data_ptr(), self._info.data_ptr(),
                                      self._turb.data_ptr())
        sp = nat.CashPenaltyStatePtrs(self._f64.data_ptr(), self._i32.data_ptr())
        nat.check(L.finenv_cashpenalty_bind(self._h, C.byref(pp), C.byref(sp)), self._h, "bind",
                  "cashpenalty")
        self.obs = torch.zeros(E, panel.D, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(E, dtype=torch.float32, device=dev)
        self.done = torch.zeros(E, dtype=torch.uint8, device=dev)
        self.term_obs = None

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                nat.lib().finenv_cashpenalty_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def enable_terminal_obs(self):
        import torch
        if self.term_obs is None:
            self.term_obs = torch.zeros_like(self.obs)
        return self.term_obs

    def set_next_start(self, starts):
        """Starting points the next reset of each env will use (the reference draws
        random.choice(range(int(len(dates) * 0.5))), :134-138)."""
        import torch
        self.state["next_start"].copy_(torch.as_tensor(
            np.broadcast_to(np.asarray(starts, np.int32), (self.num_envs,)).copy()))

    def _draw_starts(self):
        import torch
        hi = max(1, int(self.panel.T * 0.5))
        self.state["next_start"].copy_(torch.randint(
            0, hi, (self.num_envs,), generator=self._gen, device=self.device).to(torch.int32))

    def reset(self, mask=None):
        import torch
        if self.random_start:
            self._draw_starts()
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        nat.check(nat.lib().finenv_cashpenalty_reset(
            self._h, mptr, C.c_void_p(self.obs.data_ptr()), self._stream()), self._h, "reset",
            "cashpenalty")
        return self.obs

    def step(self, actions, out=None):
        import torch
        if actions.dtype != torch.float32 or not actions.is_contiguous() or \
                actions.device != self.obs.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if self.random_start and self.auto_reset:
            self._draw_starts()          # fresh starting points for envs that end this step
        obs, rew, done = out if out is not None else (self.obs, self.reward, self.done)
        nat.check(nat.lib().finenv_cashpenalty_step(
            self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
            C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
            C.c_void_p(self.term_obs.data_ptr()) if self.term_obs is not None else None,
            int(self.auto_reset), self._stream()), self._h, "step", "cashpenalty")
        return obs, rew, done, None

    def state_numpy(self):
        out = {k: v.detach().cpu().numpy() for k, v in self.state.items()}
        out["holdings"] = np.ascontiguousarray(out["holdings"].T)
        return out
