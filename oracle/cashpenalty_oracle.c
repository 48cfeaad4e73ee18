/*
 * oracle/cashpenalty_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's StockTradingEnvCashpenalty
 * (finrl/meta/env_stock_trading/env_stocktrading_cashpenalty.py: reset :131-157,
 * get_reward :237-247, get_transactions :249-289, step :291-372).
 *
 * Parity status: PINNED by outputs of the unmodified reference run in the build container
 * (tests/golden/cashpenalty_*.npz).  The reference evaluates three dot products per step with
 * np.dot (BLAS ddot: SIMD accumulators, order unspecified); this restatement sums left to
 * right, so float64 cash / reward agree to ~1e-15 relative (tests: rtol 1e-12); holdings in
 * the continuous mode are bit-exact, as are done flags, steps and observations' market part.
 * Contract: close > 0 (the reference produces NaN for 0/0), scalar hmax, float32 actions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t n_envs, n_assets, n_cols, n_days;
    int32_t discrete_actions, shares_increment, use_turbulence, patient;
    double hmax, buy_cost_pct, sell_cost_pct, initial_amount, cash_penalty_proportion,
           turbulence_threshold;
} cp_cfg;

typedef struct {
    cp_cfg cfg;
    const double *close;      /* [T][N]                                   */
    const double *info;       /* [T][N*C] date vector, ticker-major, :159-171 */
    const double *turb;       /* [T]                                      */
    double *coh, *holdings, *turbulence, *sum_trades;   /* [E], [E][N], ... */
    double *logged_total, *logged_cash;   /* account_information[...][-1], :312-314 */
    int32_t *date_index, *start, *episode;
} cp_oracle;

int cp_oracle_obs_dim(const cp_oracle *o)
{
    return 1 + o->cfg.n_assets + o->cfg.n_assets * o->cfg.n_cols;
}

cp_oracle *cp_oracle_create(const cp_cfg *cfg, const double *close, const double *info,
                            const double *turb)
{
    cp_oracle *o = (cp_oracle *)calloc(1, sizeof(*o));
    const size_t E = cfg->n_envs, N = cfg->n_assets;
    o->cfg = *cfg;
    o->close = close; o->info = info; o->turb = turb;
    o->coh = (double *)calloc(E, 8); o->holdings = (double *)calloc(E * N, 8);
    o->turbulence = (double *)calloc(E, 8); o->sum_trades = (double *)calloc(E, 8);
    o->logged_total = (double *)calloc(E, 8); o->logged_cash = (double *)calloc(E, 8);
    o->date_index = (int32_t *)calloc(E, 4); o->start = (int32_t *)calloc(E, 4);
    o->episode = (int32_t *)calloc(E, 4);
    for (size_t e = 0; e < E; e++) o->episode[e] = -1;
    return o;
}

void cp_oracle_destroy(cp_oracle *o)
{
    if (!o) return;
    free(o->coh); free(o->holdings); free(o->turbulence); free(o->sum_trades);
    free(o->logged_total); free(o->logged_cash); free(o->date_index); free(o->start); free(o->episode); free(o);
}

static void write_obs(const cp_oracle *o, int e, double *obs)
{
    const int N = o->cfg.n_assets, C = o->cfg.n_cols;
    obs[0] = o->coh[e];
    memcpy(obs + 1, o->holdings + (size_t)e * N, 8 * N);
    memcpy(obs + 1 + N, o->info + (size_t)o->date_index[e] * N * C, 8 * N * C);
}

/* reset(): the starting point is drawn by the caller (the reference uses `random`, :134-138) */
void cp_oracle_reset_env(cp_oracle *o, int e, int start, double *obs)
{
    const int N = o->cfg.n_assets;
    o->start[e] = start;
    o->date_index[e] = start;
    o->turbulence[e] = 0.0;
    o->sum_trades[e] = 0.0;
    o->episode[e] += 1;
    o->coh[e] = o->cfg.initial_amount;
    memset(o->holdings + (size_t)e * N, 0, 8 * N);
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

static double get_reward(const cp_cfg *c, int step, double total, double cash)   /* :237-247 */
{
    if (step == 0) return 0.0;
    const double pen = fmax(0.0, total * c->cash_penalty_proportion - cash);
    double r = ((total - pen) / c->initial_amount) - 1;
    r /= (double)step;
    return r;
}

/* step(), :291-372.  returns the terminal reason: 0 none, 1 last date, 2 cash shortage */
int cp_oracle_step_env(cp_oracle *o, int e, const float *act, double *obs, double *reward,
                       uint8_t *done)
{
    const cp_cfg *c = &o->cfg;
    const int N = c->n_assets;
    double *h = o->holdings + (size_t)e * N;
    double tr[512];
    for (int i = 0; i < N; i++) o->sum_trades[e] += fabs((double)act[i]);        /* :293 */
    const int step = o->date_index[e] - o->start[e];                              /* current_step */
    if (o->date_index[e] == c->n_days - 1) {                                      /* :299-301 */
        *reward = get_reward(c, step, o->logged_total[e], o->logged_cash[e]);
        *done = 1;
        if (obs) write_obs(o, e, obs);
        return 1;
    }
    const double *cl = o->close + (size_t)o->date_index[e] * N;
    const double begin_cash = o->coh[e];                                          /* :308 */
    double asset_value = 0.0;
    for (int i = 0; i < N; i++) asset_value += h[i] * cl[i];                      /* np.dot, :310 */
    o->logged_cash[e] = begin_cash;                                               /* :312-314 */
    o->logged_total[e] = begin_cash + asset_value;
    const double rew = get_reward(c, step, o->logged_total[e], begin_cash);       /* :317 */

    /* get_transactions(), :249-289 */
    const float hmaxf = (float)c->hmax;
    for (int i = 0; i < N; i++) {
        volatile float a32 = act[i] * hmaxf;                   /* actions * hmax, float32 */
        const float a = cl[i] > 0 ? a32 : 0.0f;                /* np.where(closings > 0, ., 0) */
        if (c->discrete_actions) {                                                /* :263-274 */
            int64_t q = (int64_t)floordiv64((double)a, cl[i]);
            const int64_t inc = c->shares_increment;
            q = q >= 0 ? floordiv_i64(q, inc) * inc : floordiv_i64(q + inc, inc) * inc;
            tr[i] = (double)q;
        } else {
            tr[i] = (double)a / cl[i];                                            /* :276 */
        }
        tr[i] = fmax(tr[i], -h[i]);                                               /* :279 */
    }
    if (c->use_turbulence && o->turbulence[e] >= c->turbulence_threshold)         /* :282-287 */
        for (int i = 0; i < N; i++) tr[i] = -h[i];

    double proceeds = 0.0, spend = 0.0;                                           /* :323-331 */
    for (int i = 0; i < N; i++) proceeds += (tr[i] < 0 ? -tr[i] : 0.0) * cl[i];
    double costs = proceeds * c->sell_cost_pct;
    double coh = begin_cash + proceeds;
    for (int i = 0; i < N; i++) spend += (tr[i] > 0 ? tr[i] : 0.0) * cl[i];
    costs += spend * c->buy_cost_pct;
    if (spend + costs > coh) {                                                    /* :333 */
        if (c->patient) {                                                         /* :334-339 */
            for (int i = 0; i < N; i++) if (tr[i] > 0) tr[i] = 0.0;
            spend = 0.0;
            costs = 0.0;
        } else {                                                                  /* :341-344 */
            *reward = rew;
            *done = 1;
            if (obs) write_obs(o, e, obs);
            return 2;
        }
    }
    coh = coh - spend - costs;                                                    /* :351 */
    for (int i = 0; i < N; i++) h[i] = h[i] + tr[i];                              /* :352 */
    o->coh[e] = coh;
    o->date_index[e] += 1;                                                        /* :353 */
    if (c->use_turbulence) o->turbulence[e] = o->turb[o->date_index[e]];          /* :354-357 */
    if (obs) write_obs(o, e, obs);
    *reward = rew;
    *done = 0;
    return 0;
}

/* SB3 DummyVecEnv semantics; starts [E] = starting points to use when an env resets */
void cp_oracle_vec_step(cp_oracle *o, const float *act, double *obs, double *reward,
                        uint8_t *done, double *term_obs, const int32_t *starts, int auto_reset)
{
    const int E = o->cfg.n_envs, N = o->cfg.n_assets, D = cp_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) {
        double *ob = obs ? obs + (size_t)e * D : NULL;
        cp_oracle_step_env(o, e, act + (size_t)e * N, ob, reward + e, done + e);
        if (done[e] && auto_reset) {
            if (term_obs && ob) memcpy(term_obs + (size_t)e * D, ob, D * sizeof(double));
            cp_oracle_reset_env(o, e, starts ? starts[e] : 0, ob);
        }
    }
}

void cp_oracle_reset(cp_oracle *o, const int32_t *starts, double *obs)
{
    const int E = o->cfg.n_envs, D = cp_oracle_obs_dim(o);
    for (int e = 0; e < E; e++)
        cp_oracle_reset_env(o, e, starts ? starts[e] : 0, obs ? obs + (size_t)e * D : NULL);
}

void cp_oracle_get_state(const cp_oracle *o, double *coh, double *holdings, int32_t *date_index,
                         int32_t *start, double *turbulence, double *sum_trades,
                         double *logged_total, double *logged_cash, int32_t *episode)
{
    const size_t E = o->cfg.n_envs, N = o->cfg.n_assets;
    memcpy(coh, o->coh, E * 8); memcpy(holdings, o->holdings, E * N * 8);
    memcpy(date_index, o->date_index, E * 4); memcpy(start, o->start, E * 4);
    memcpy(turbulence, o->turbulence, E * 8); memcpy(sum_trades, o->sum_trades, E * 8);
    memcpy(logged_total, o->logged_total, E * 8); memcpy(logged_cash, o->logged_cash, E * 8);
    memcpy(episode, o->episode, E * 4);
}
