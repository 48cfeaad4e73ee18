"""Drop-in for ``finrl.meta.env_stock_trading.env_stocktrading_stoploss.StockTradingEnvStopLoss``
(env_stocktrading_stoploss.py:19-459 in the reference tree): the cash-penalty facade plus
``stoploss_penalty`` / ``profit_loss_ratio`` and the average-buy-price books
(``avg_buy_price, n_buys, closing_diff_avg_buy, profit_sell_diff_avg_buy, actual_num_trades``);
one HIP launch per step through the C ABI (finenv_stoploss_*)."""
from __future__ import annotations

from ... import _native as nat
from ...vec_cashpenalty import VecStopLossEnv
from .env_stocktrading_cashpenalty import StockTradingEnvCashpenalty


class StockTradingEnvStopLoss(StockTradingEnvCashpenalty):
    _vec_cls = VecStopLossEnv

    def __init__(self, df, buy_cost_pct=3e-3, sell_cost_pct=3e-3, date_col_name="date", hmax=10,
                 discrete_actions=False, shares_increment=1, stoploss_penalty=0.9,
                 profit_loss_ratio=2, turbulence_threshold=None, print_verbosity=10,
                 initial_amount=1e6,
                 daily_information_cols=["open", "close", "high", "low", "volume"],
                 cache_indicator_data=True, cash_penalty_proportion=0.1, random_start=True,
                 patient=False, currency="$", device="cuda"):
        self.stoploss_penalty = stoploss_penalty
        self.profit_loss_ratio = profit_loss_ratio
        self.min_profit_penalty = 1 + profit_loss_ratio * (1 - stoploss_penalty)    # :101
        super().__init__(df, buy_cost_pct, sell_cost_pct, date_col_name, hmax, discrete_actions,
                         shares_increment, turbulence_threshold, print_verbosity, initial_amount,
                         daily_information_cols, cache_indicator_data, cash_penalty_proportion,
                         random_start, patient, currency, device,
                         stoploss_penalty=stoploss_penalty, profit_loss_ratio=profit_loss_ratio)

    def _extra_ctor(self):
        return dict(stoploss_penalty=self.stoploss_penalty, profit_loss_ratio=self.profit_loss_ratio)

    def _record_action(self, actions, closings):
        self.actions_memory.append((actions * self.hmax) * closings)                # :321-324

    def _reasons_before_shortage(self, flags):
        return [("TURBULENCE", flags & nat.AUDIT_F_TURBULENCE),                     # :327-331
                ("STOP LOSS", flags & nat.AUDIT_F_STOP_LOSS)]                       # :359-360

    def _reasons_after_trades(self, flags):                                         # :401-405
        if flags & nat.AUDIT_F_LOW_PROFIT:
            return [("LOW PROFIT", True)]
        return [("HIGH PROFIT", flags & nat.AUDIT_F_HIGH_PROFIT)]

    def _record_extra(self, logger):                                                # :195-198
        logger.record("environment/actual_num_trades", self.actual_num_trades)

    def _sync(self):
        state = super()._sync()
        st = self._st
        self.avg_buy_price = st["avg_buy_price"][0]
        self.n_buys = st["n_buys"][0]
        self.closing_diff_avg_buy = st["closing_diff_avg_buy"][0]
        self.profit_sell_diff_avg_buy = st["profit_sell_diff_avg_buy"][0]
        self.actual_num_trades = float(st["actual_num_trades"][0])
        return state
