"""The reference's own CALLER loops, restated (test infrastructure): what ``DRLAgent`` does with
an env after training.  The tests run these against the ``finrl_amd.meta`` facades; the golden
generator (tests/golden/make_golden.py, build container only) runs the very same functions against
the unmodified reference envs, so the fixtures pin the facade's harness surface end to end.

  * ``drl_prediction``          <- finrl/agents/stablebaselines3/models.py:110-129
  * ``elegantrl_prediction``    <- finrl/agents/elegantrl/models.py:105-125 (torch tensors replaced
                                   by numpy: the agent network is not part of the env path)
  * ``drl_validation`` / ``get_validation_sharpe`` / ``ensemble_prediction``
                                <- models.py:272-276, :213-230, :278-325: the rolling-window
                                   ensemble's use of the env (CSV dumps of the terminal branch read
                                   back for the Sharpe ratio; ``render()`` -> ``last_state`` ->
                                   ``initial=False, previous_state=last_state`` of the next window);
                                   scalar costs (the fork's env raises on the lists the reference's
                                   own ensemble passes, SURVEY.md headline 5)
A trained policy is replaced by ``ScriptedModel`` / ``scripted_act``: deterministic functions of
(step, observation) whose float32 arithmetic is exact, so both sides see identical actions only if
the observations they return are identical.
"""
from __future__ import annotations

import numpy as np


def drl_prediction(model, environment, deterministic=True):
    """models.py:110-129: roll one episode through ``environment.get_sb_env()`` and fetch the
    account / action memories through ``env_method(method_name=...)`` on the second-to-last day."""
    test_env, test_obs = environment.get_sb_env()
    account_memory = []
    actions_memory = []
    test_env.reset()
    n_days = len(environment.df.index.unique())
    for i in range(n_days):
        action, _states = model.predict(test_obs, deterministic=deterministic)
        test_obs, rewards, dones, info = test_env.step(action)
        if i == (n_days - 2):
            account_memory = test_env.env_method(method_name="save_asset_memory")
            actions_memory = test_env.env_method(method_name="save_action_memory")
        if dones[0]:
            print("hit end!")
            break
    return account_memory[0], actions_memory[0]


def elegantrl_prediction(act, environment):
    """models.py:105-125: step the array-state env with the actor's output and rebuild the
    account value from the attributes the loop reads (amount, price_ary, day, stocks)."""
    environment.env_num = 1
    state = environment.reset()
    episode_returns = []
    episode_total_assets = [environment.initial_total_asset]
    for i in range(environment.max_step):
        s_tensor = np.asarray((state,))
        a_tensor = act(s_tensor)
        action = a_tensor[0]
        state, reward, done, _ = environment.step(action)
        total_asset = environment.amount + \
            (environment.price_ary[environment.day] * environment.stocks).sum()
        episode_total_assets.append(total_asset)
        episode_return = total_asset / environment.initial_total_asset
        episode_returns.append(episode_return)
        if done:
            break
    return episode_total_assets, episode_returns


def drl_validation(model, test_data, test_env, test_obs):
    """models.py:272-276: one pass over the validation window (the env's terminal branch writes
    results/account_value_validation_<model>_<iteration>.csv, env_stocktrading.py:266-292)."""
    for _ in range(len(test_data.index.unique())):
        action, _states = model.predict(test_obs)
        test_obs, rewards, dones, info = test_env.step(action)


def get_validation_sharpe(iteration, model_name):
    """models.py:213-230."""
    import pandas as pd
    df_total_value = pd.read_csv(f"results/account_value_validation_{model_name}_{iteration}.csv")
    if df_total_value["daily_return"].var() == 0:
        return np.inf if df_total_value["daily_return"].mean() > 0 else 0.0
    return (4 ** 0.5) * df_total_value["daily_return"].mean() / df_total_value["daily_return"].std()


def ensemble_prediction(make_vec_env, env_cls, model, trade_data, env_kwargs, name, last_state,
                        iter_num, turbulence_threshold, initial):
    """models.py:278-325: trade one window through ``DummyVecEnv([lambda: StockTradingEnv(...)])``,
    take ``render()`` on the second-to-last day as the state handed to the next window, dump it."""
    import pandas as pd
    trade_env = make_vec_env([lambda: env_cls(
        df=trade_data, turbulence_threshold=turbulence_threshold, initial=initial,
        previous_state=last_state, model_name=name, mode="trade", iteration=iter_num,
        **env_kwargs)])
    trade_obs = trade_env.reset()
    n_days = len(trade_data.index.unique())
    for i in range(n_days):
        action, _states = model.predict(trade_obs)
        trade_obs, rewards, dones, info = trade_env.step(action)
        if i == (n_days - 2):
            last_state = trade_env.render()
    df_last_state = pd.DataFrame({"last_state": last_state})
    df_last_state.to_csv(f"results/last_state_{name}_{i}.csv", index=False)
    return last_state


class ScriptedModel:
    """Stands in for a trained SB3 model: ``predict(obs) -> (action [1, N] f32, None)``.
    action_j = 0.75 * base[step, j] + 0.25 * sign(obs[cols[j]] - obs[cols[j+1]]): exact in
    float32 and sensitive to the observation the VecEnv returned."""

    def __init__(self, base, cols):
        self.base = np.asarray(base, dtype=np.float32)
        self.cols = np.asarray(cols, dtype=np.int64)
        self.step = 0

    def predict(self, obs, deterministic=True):
        o = np.asarray(obs)
        assert o.ndim == 2 and o.shape[0] == 1 and o.dtype == np.float32, (o.shape, o.dtype)
        v = o[0, self.cols]
        sgn = np.sign(v - np.roll(v, -1)).astype(np.float32)
        a = np.float32(0.75) * self.base[self.step % len(self.base)] + np.float32(0.25) * sgn
        self.step += 1
        return a[None].astype(np.float32), None


def scripted_act(base, cols):
    """Stands in for ``agent.act``: state [1, D] -> action [1, N] f32."""
    model = ScriptedModel(base, cols)
    return lambda s: model.predict(np.asarray(s, dtype=np.float32))[0]
