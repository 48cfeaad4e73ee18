/*
 * oracle/stoploss_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's StockTradingEnvStopLoss
 * (finrl/meta/env_stock_trading/env_stocktrading_stoploss.py: reset :134-165,
 * get_reward :255-290, step :292-442).
 *
 * Parity status: PINNED by outputs of the unmodified reference run in the build container
 * (tests/golden/stoploss_*.npz).  As in cashpenalty_oracle.c the reference's np.dot calls
 * (BLAS, order unspecified) are summed left to right here: money agrees to ~1e-15 relative
 * (tests: rtol 1e-12); flags, steps, average buy prices and the market part of the
 * observation are exact.  Contract: close > 0, scalar hmax, float32 actions.
 *
 * Quirks kept on purpose (each checked against the goldens):
 *  - get_reward() is called BEFORE this step's account_information entry is appended (:313 vs
 *    :315-318), so the per-step reward uses the PREVIOUS step's logged cash / total assets,
 *    while the cash-shortage terminal (:381-383) calls it again and sees the fresh entry and
 *    the freshly updated closing_diff_avg_buy (:350).
 *  - turbulence sell-off is built in dollars, -(holdings*close) (:330), then divided by close
 *    again (:345) and clipped at -holdings (:348): (h*c)/c may differ from h in the last bit.
 *  - patient mode zeroes the buy transactions but not `buys` (:376 vs :418): n_buys and
 *    avg_buy_price still advance for the buys that were cancelled.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t n_envs, n_assets, n_cols, n_days;
    int32_t discrete_actions, shares_increment, use_turbulence, patient;
    double hmax, buy_cost_pct, sell_cost_pct, initial_amount, cash_penalty_proportion,
           turbulence_threshold, stoploss_penalty, min_profit_penalty;
} sl_cfg;

typedef struct {
    sl_cfg cfg;
    const double *close, *info, *turb;
    double *coh, *turbulence, *sum_trades, *logged_total, *logged_cash, *actual_num_trades;
    double *holdings, *prev_holdings, *cdab, *psdab, *n_buys, *avg_buy;   /* [E][N] each */
    int32_t *date_index, *start, *episode;
} sl_oracle;

int sl_oracle_obs_dim(const sl_oracle *o)
{
    return 1 + o->cfg.n_assets + o->cfg.n_assets * o->cfg.n_cols;
}

sl_oracle *sl_oracle_create(const sl_cfg *cfg, const double *close, const double *info,
                            const double *turb)
{
    sl_oracle *o = (sl_oracle *)calloc(1, sizeof(*o));
    const size_t E = cfg->n_envs, N = cfg->n_assets;
    o->cfg = *cfg;
    o->close = close; o->info = info; o->turb = turb;
    o->coh = (double *)calloc(E, 8); o->turbulence = (double *)calloc(E, 8);
    o->sum_trades = (double *)calloc(E, 8); o->logged_total = (double *)calloc(E, 8);
    o->logged_cash = (double *)calloc(E, 8); o->actual_num_trades = (double *)calloc(E, 8);
    o->holdings = (double *)calloc(E * N, 8); o->prev_holdings = (double *)calloc(E * N, 8);
    o->cdab = (double *)calloc(E * N, 8); o->psdab = (double *)calloc(E * N, 8);
    o->n_buys = (double *)calloc(E * N, 8); o->avg_buy = (double *)calloc(E * N, 8);
    o->date_index = (int32_t *)calloc(E, 4); o->start = (int32_t *)calloc(E, 4);
    o->episode = (int32_t *)calloc(E, 4);
    for (size_t e = 0; e < E; e++) o->episode[e] = -1;
    return o;
}

void sl_oracle_destroy(sl_oracle *o)
{
    if (!o) return;
    free(o->coh); free(o->turbulence); free(o->sum_trades); free(o->logged_total);
    free(o->logged_cash); free(o->actual_num_trades); free(o->holdings); free(o->prev_holdings);
    free(o->cdab); free(o->psdab); free(o->n_buys); free(o->avg_buy); free(o->date_index);
    free(o->start); free(o->episode); free(o);
}

static void write_obs(const sl_oracle *o, int e, double *obs)
{
    const int N = o->cfg.n_assets, C = o->cfg.n_cols;
    obs[0] = o->coh[e];
    memcpy(obs + 1, o->holdings + (size_t)e * N, 8 * N);
    memcpy(obs + 1 + N, o->info + (size_t)o->date_index[e] * N * C, 8 * N * C);
}

void sl_oracle_reset_env(sl_oracle *o, int e, int start, double *obs)            /* :134-165 */
{
    const size_t N = o->cfg.n_assets, b = (size_t)e * N;
    o->sum_trades[e] = 0.0;
    o->actual_num_trades[e] = 0.0;
    memset(o->cdab + b, 0, 8 * N); memset(o->psdab + b, 0, 8 * N);
    memset(o->n_buys + b, 0, 8 * N); memset(o->avg_buy + b, 0, 8 * N);
    o->start[e] = start;
    o->date_index[e] = start;
    o->turbulence[e] = 0.0;
    o->episode[e] += 1;
    o->coh[e] = o->cfg.initial_amount;
    memset(o->holdings + b, 0, 8 * N); memset(o->prev_holdings + b, 0, 8 * N);
    if (obs) write_obs(o, e, obs);
}

static int64_t floordiv_i64(int64_t a, int64_t b)
{
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
    return q;
}

static double floordiv64(double a, double b)
{
    double mod, div, fl;
    if (b == 0.0) return a / b;
    mod = fmod(a, b);
    div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) { mod += b; div -= 1.0; }
    if (div != 0.0) { fl = floor(div); if (div - fl > 0.5) fl += 1.0; }
    else fl = copysign(0.0, a / b);
    return fl;
}

static double get_reward(const sl_oracle *o, int e, int step)                    /* :255-290 */
{
    const sl_cfg *c = &o->cfg;
    const size_t N = c->n_assets, b = (size_t)e * N;
    if (step == 0) return 0.0;
    const double total = o->logged_total[e], cash = o->logged_cash[e];
    const double cash_penalty = fmax(0.0, total * c->cash_penalty_proportion - cash);
    double slp = 0.0, lpp = 0.0, add = 0.0;
    if (step > 1) {
        for (size_t i = 0; i < N; i++) slp += o->prev_holdings[b + i] * fmin(o->cdab[b + i], 0.0);
        slp = -1 * slp;
    }
    for (size_t i = 0; i < N; i++) lpp += o->holdings[b + i] * fmin(o->psdab[b + i], 0.0);
    lpp = -1 * lpp;
    const double total_penalty = cash_penalty + slp + lpp;
    for (size_t i = 0; i < N; i++) add += o->holdings[b + i] * fmax(o->psdab[b + i], 0.0);
    double r = ((total - total_penalty + add) / c->initial_amount) - 1;
    r /= (double)step;
    return r;
}

/* step(), :292-442.  returns 0 none, 1 last date, 2 cash shortage */
int sl_oracle_step_env(sl_oracle *o, int e, const float *act, double *obs, double *reward,
                       uint8_t *done)
{
    const sl_cfg *c = &o->cfg;
    const int N = c->n_assets;
    const size_t b = (size_t)e * N;
    double *h = o->holdings + b, *abp = o->avg_buy + b, *nb = o->n_buys + b;
    double tr[512], sells[512], buys[512];
    for (int i = 0; i < N; i++) o->sum_trades[e] += fabs((double)act[i]);        /* :294 */
    const int step = o->date_index[e] - o->start[e];
    if (o->date_index[e] == c->n_days - 1) {                                      /* :302-304 */
        *reward = get_reward(o, e, step);
        *done = 1;
        if (obs) write_obs(o, e, obs);
        return 1;
    }
    const double *cl = o->close + (size_t)o->date_index[e] * N;
    const double begin_cash = o->coh[e];                                          /* :307 */
    double asset_value = 0.0;
    for (int i = 0; i < N; i++) asset_value += h[i] * cl[i];                      /* :311 */
    const double rew = get_reward(o, e, step);                                    /* :313 (stale log) */
    o->logged_cash[e] = begin_cash;                                               /* :315-317 */
    o->logged_total[e] = begin_cash + asset_value;

    const float hmaxf = (float)c->hmax;
    const int turbulent = c->use_turbulence && o->turbulence[e] >= c->turbulence_threshold;
    for (int i = 0; i < N; i++) {
        volatile float a32 = act[i] * hmaxf;                                      /* :321 float32 */
        double a = cl[i] > 0 ? (double)a32 : 0.0;                                 /* :326 */
        if (turbulent) a = -(h[i] * cl[i]);                                       /* :327-331 */
        if (c->discrete_actions) {                                                /* :333-343 */
            int64_t q = cl[i] > 0 ? (int64_t)floordiv64(a, cl[i]) : 0;
            const int64_t inc = c->shares_increment;
            q = q >= 0 ? floordiv_i64(q, inc) * inc : floordiv_i64(q + inc, inc) * inc;
            tr[i] = (double)q;
        } else {
            tr[i] = cl[i] > 0 ? a / cl[i] : 0.0;                                  /* :345 */
        }
        tr[i] = fmax(tr[i], -h[i]);                                               /* :348 */
        o->cdab[b + i] = cl[i] - (c->stoploss_penalty * abp[i]);                  /* :350-352 */
    }
    if (begin_cash >= c->stoploss_penalty * c->initial_amount)                    /* :353-357 */
        for (int i = 0; i < N; i++) if (o->cdab[b + i] < 0) tr[i] = -h[i];

    double proceeds = 0.0, spend = 0.0;
    for (int i = 0; i < N; i++) {                                                 /* :363-364 */
        sells[i] = tr[i] < 0 ? -tr[i] : 0.0;
        proceeds += sells[i] * cl[i];
    }
    double costs = proceeds * c->sell_cost_pct;
    double coh = begin_cash + proceeds;
    for (int i = 0; i < N; i++) {                                                 /* :368-369 */
        buys[i] = tr[i] > 0 ? tr[i] : 0.0;
        spend += buys[i] * cl[i];
    }
    costs += spend * c->buy_cost_pct;
    if (spend + costs > coh) {                                                    /* :372 */
        if (c->patient) {                                                         /* :373-378 */
            for (int i = 0; i < N; i++) if (tr[i] > 0) tr[i] = 0.0;
            spend = 0.0;
            costs = 0.0;
        } else {                                                                  /* :379-383 */
            *reward = get_reward(o, e, step);
            *done = 1;
            if (obs) write_obs(o, e, obs);
            return 2;
        }
    }
    double ntr = 0.0;
    for (int i = 0; i < N; i++) {                                                 /* :388-399 */
        const double scp = sells[i] > 0 ? cl[i] : 0.0;
        const int profit = scp - abp[i] > 0;
        o->psdab[b + i] = profit ? cl[i] - (c->min_profit_penalty * abp[i]) : 0.0;
        ntr += tr[i] != 0.0 ? 1.0 : 0.0;                                          /* :411 */
    }
    o->actual_num_trades[e] = ntr;
    coh = coh - spend - costs;                                                    /* :414 */
    for (int i = 0; i < N; i++) {
        const double hu = h[i] + tr[i];                                           /* :415 */
        const double sb = buys[i] > 0 ? 1.0 : 0.0;                                /* :418 */
        nb[i] += sb;                                                              /* :419 */
        if (sb > 0) abp[i] = abp[i] + ((cl[i] - abp[i]) / nb[i]);                 /* :420-424 */
        if (!(hu > 0)) { nb[i] = 0.0; abp[i] = 0.0; }                             /* :427-428 */
        o->prev_holdings[b + i] = h[i];
        h[i] = hu;
    }
    o->coh[e] = coh;
    o->date_index[e] += 1;                                                        /* :430 */
    if (c->use_turbulence) o->turbulence[e] = o->turb[o->date_index[e]];          /* :431-434 */
    if (obs) write_obs(o, e, obs);
    *reward = rew;
    *done = 0;
    return 0;
}

void sl_oracle_vec_step(sl_oracle *o, const float *act, double *obs, double *reward,
                        uint8_t *done, double *term_obs, const int32_t *starts, int auto_reset)
{
    const int E = o->cfg.n_envs, N = o->cfg.n_assets, D = sl_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) {
        double *ob = obs ? obs + (size_t)e * D : NULL;
        sl_oracle_step_env(o, e, act + (size_t)e * N, ob, reward + e, done + e);
        if (done[e] && auto_reset) {
            if (term_obs && ob) memcpy(term_obs + (size_t)e * D, ob, D * sizeof(double));
            sl_oracle_reset_env(o, e, starts ? starts[e] : 0, ob);
        }
    }
}

void sl_oracle_reset(sl_oracle *o, const int32_t *starts, double *obs)
{
    const int E = o->cfg.n_envs, D = sl_oracle_obs_dim(o);
    for (int e = 0; e < E; e++)
        sl_oracle_reset_env(o, e, starts ? starts[e] : 0, obs ? obs + (size_t)e * D : NULL);
}

/* scal [8][E]: coh turbulence sum_trades logged_total logged_cash actual_num_trades;
 * vec [6][E][N]: holdings prev_holdings cdab psdab n_buys avg_buy; ints [3][E] */
void sl_oracle_get_state(const sl_oracle *o, double *scal, double *vec, int32_t *ints)
{
    const size_t E = o->cfg.n_envs, N = o->cfg.n_assets;
    const double *s[6] = {o->coh, o->turbulence, o->sum_trades, o->logged_total, o->logged_cash,
                          o->actual_num_trades};
    const double *v[6] = {o->holdings, o->prev_holdings, o->cdab, o->psdab, o->n_buys, o->avg_buy};
    for (int k = 0; k < 6; k++) memcpy(scal + k * E, s[k], E * 8);
    for (int k = 0; k < 6; k++) memcpy(vec + k * E * N, v[k], E * N * 8);
    memcpy(ints, o->date_index, E * 4); memcpy(ints + E, o->start, E * 4);
    memcpy(ints + 2 * E, o->episode, E * 4);
}
