"""Drop-in for ``finrl.meta.env_cryptocurrency_trading.env_multiple_crypto.CryptoEnv``
(env_multiple_crypto.py:10-111 in the reference tree): same constructor, attributes and
``reset() / step()`` protocol (``info`` is ``None``, as in the reference, :90), one HIP launch per
step through the C ABI (finenv_crypto_*).  The reference scales the caller's action array in
place (:63-65); so does this facade."""
from __future__ import annotations

import numpy as np

from ...spaces import Box
from ...vec_crypto import VecCryptoEnv, action_norm_vector
from .._single import to_action_tensor


class CryptoEnv:
    def __init__(self, config, lookback=1, initial_capital=1e6, buy_cost_pct=1e-3,
                 sell_cost_pct=1e-3, gamma=0.99, device="cuda"):
        self._vec = VecCryptoEnv(config, 1, lookback=lookback, initial_capital=initial_capital,
                                 buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct,
                                 gamma=gamma, auto_reset=False, device=device)
        v = self._vec
        self.lookback = lookback
        self.initial_total_asset = self.initial_cash = initial_capital
        self.buy_cost_pct, self.sell_cost_pct, self.gamma = buy_cost_pct, sell_cost_pct, gamma
        self.max_stock = 1
        self.price_array, self.tech_array = v.price_array, v.tech_array
        self.action_norm_vector = action_norm_vector(self.price_array[0])           # :103-111
        self.crypto_num = self.price_array.shape[1]
        self.max_step = self.price_array.shape[0] - lookback - 1                    # :24
        self.env_name = "MulticryptoEnv"
        self.state_dim, self.action_dim = v.state_dim, v.action_dim
        self.if_discrete = False
        self.target_return = 10
        self.episode_return = 0.0
        self.observation_space = Box(low=-3000, high=3000, shape=(self.state_dim,), dtype=np.float32)
        self.action_space = Box(low=-1, high=1, shape=(self.action_dim,), dtype=np.float32)
        self._sync()

    @classmethod
    def make_vec(cls, config, num_envs, **kw):
        return VecCryptoEnv(config, num_envs, **kw)

    def _sync(self):
        st = self._vec.state_numpy()
        self.time = int(st["time"][0])
        self.cash = float(st["cash"][0])
        self.stocks = st["stocks"][0]
        self.total_asset = float(st["total_asset"][0])
        self.gamma_return = float(st["gamma_return"][0])
        self.current_price = self.price_array[self.time]
        self.current_tech = self.tech_array[self.time]

    def reset(self):
        obs = self._vec.reset().cpu().numpy()[0]
        self._sync()
        return obs

    def step(self, actions):
        act = to_action_tensor(self._vec, actions)
        obs, rew, done, _ = self._vec.step(act)
        if isinstance(actions, np.ndarray):              # in-place scaling of the caller's array, :63-65
            for i in range(self.action_dim):
                actions[i] = actions[i] * self.action_norm_vector[i]
        self._sync()
        d = bool(done.cpu().numpy()[0])
        if d:
            self.episode_return = float(self._vec.state_numpy()["episode_return"][0])   # :89
        return obs.cpu().numpy()[0], float(rew.cpu().numpy()[0]), d, None

    def close(self):
        pass
