/*
 * oracle/stock_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's StockTradingEnv.step/reset hot path
 * (finrl/meta/env_stock_trading/env_stocktrading.py in the reference tree; every
 * function below cites the reference lines it follows).  It is the checker for
 * the HIP path and the "port" CPU baseline in bench.py; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (finrl_amd/) never links or calls it.
 *
 * Parity status: PINNED.  The reference holds no golden vectors for this path
 * (SURVEY.md 4/8c), so the restatement is pinned by outputs of the unmodified
 * reference run in the build container (tests/golden/make_golden.py ->
 * tests/golden/stock_*.npz; tests/test_oracle_golden.py replays them here).
 *
 * Arithmetic contract (App. A of SURVEY.md):
 *   - all money arithmetic IEEE fp64, each * and + separately rounded, in the
 *     reference's left-to-right order: p*q*(1+-c) == (p*q)*(1+-c);
 *     build with -ffp-contract=off;
 *   - begin/end asset = cash + (sequential sum from 0 of close_i*shares_i)
 *     (Python builtin sum over an fp64 array, :311-314, :344-347);
 *   - `//` is NumPy/CPython floor division (fmod-based, exact), :178-180;
 *   - action scaling is a float32 multiply then truncation toward zero, :304-307;
 *   - canonical order = STABLE ascending argsort (ties: lower index first for
 *     sells, higher index first for buys because the reference reverses the
 *     permutation, :317-319).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t n_envs;
    int32_t n_tickers;          /* N = stock_dim                                     */
    int32_t n_tech;             /* K = len(tech_indicator_list)                      */
    int32_t n_days;             /* T = len(df.index.unique())                        */
    int32_t hmax;               /* :51                                               */
    int32_t use_turbulence;     /* turbulence_threshold is not None, :68             */
    int32_t reset_quirk;        /* 1 = reference stale-row reset (:361 before :380)  */
    int32_t initial;            /* `initial` ctor flag, :70 (asset0 summation order) */
    int32_t single_ticker;      /* len(df.tic.unique()) == 1: the single-stock branches
                                   (:415-422: `[0] * stock_dim` shares when initial)  */
    int32_t reserved0;
    double  buy_cost_pct;       /* scalar in this fork, :54                          */
    double  sell_cost_pct;      /* :55                                               */
    double  reward_scaling;     /* :56                                               */
    double  turbulence_threshold;
} stock_cfg;

typedef struct {
    double  cash;               /* state[0]                                          */
    double  cost;               /* self.cost                                         */
    double  last_reward;        /* self.reward (scaled), survives reset (:359-393)   */
    double  turbulence;         /* self.turbulence                                   */
    double  asset0;             /* asset_memory[0]                                   */
    double  prev_asset;         /* asset_memory[-1]                                  */
    double  ret_sum, ret_sumsq; /* running sums of pct_change(asset_memory)          */
    int32_t n_ret;
    int32_t day;                /* self.day                                          */
    int32_t price_day;          /* day whose row sits in self.state / self.data      */
    int32_t trades;             /* self.trades                                       */
    int32_t episode;            /* self.episode                                      */
    int32_t terminal;
} stock_env;

typedef struct {
    stock_cfg cfg;
    const double *close;        /* [T][N]                                            */
    const double *tech;         /* [T][K][N]  (indicator-major, :460-466)            */
    const double *risk;         /* [T]        (df[risk_indicator_col], :337-341)     */
    double  *cash0;             /* [E]   initial_amount / previous_state[0]          */
    int64_t *shares0;           /* [E][N] num_stock_shares / previous_state shares   */
    stock_env *env;             /* [E]                                               */
    int64_t *shares;            /* [E][N] state[1+N : 1+2N]                          */
    int64_t *scratch;           /* [3N] per-call scratch                             */
} stock_oracle;

/* NumPy npy_divmod / CPython float_floor_div: exact floor division via fmod. */
static double floordiv_exact(double a, double b)
{
    double mod, div, fl;
    if (b == 0.0) return a / b;
    mod = fmod(a, b);
    div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) { mod += b; div -= 1.0; }
    if (div != 0.0) {
        fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
    } else {
        fl = copysign(0.0, a / b);
    }
    return fl;
}

/* np.sum over a contiguous fp64 array of n < 128 (NumPy pairwise_sum, 8 lanes). */
static double np_sum_small(const double *a, int n)
{
    double r[8], res;
    int i, j;
    if (n < 8) {
        res = 0.0;
        for (i = 0; i < n; i++) res += a[i];
        return res;
    }
    for (j = 0; j < 8; j++) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8)
        for (j = 0; j < 8; j++) r[j] += a[i + j];
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* cash + builtin-sum(close*shares): :311-314 / :344-347 / :226-229 */
static double total_asset(const stock_oracle *o, int e, int row)
{
    const int N = o->cfg.n_tickers;
    const double *p = o->close + (size_t)row * N;
    const int64_t *h = o->shares + (size_t)e * N;
    double s = 0.0;
    for (int i = 0; i < N; i++) s = s + p[i] * (double)h[i];
    return o->env[e].cash + s;
}

stock_oracle *stock_oracle_create(const stock_cfg *cfg, const double *close, const double *tech,
                                  const double *risk)
{
    stock_oracle *o = (stock_oracle *)calloc(1, sizeof(*o));
    const size_t E = cfg->n_envs, N = cfg->n_tickers;
    o->cfg = *cfg;
    o->close = close; o->tech = tech; o->risk = risk;
    o->cash0 = (double *)calloc(E, sizeof(double));
    o->shares0 = (int64_t *)calloc(E * N, sizeof(int64_t));
    o->env = (stock_env *)calloc(E, sizeof(stock_env));
    o->shares = (int64_t *)calloc(E * N, sizeof(int64_t));
    o->scratch = (int64_t *)calloc(3 * N + 8, sizeof(int64_t));
    return o;
}

void stock_oracle_destroy(stock_oracle *o)
{
    if (!o) return;
    free(o->cash0); free(o->shares0); free(o->env); free(o->shares); free(o->scratch); free(o);
}

static double initial_asset(const stock_oracle *o, int e, int row)
{
    const int N = o->cfg.n_tickers;
    const double *p = o->close + (size_t)row * N;
    const int64_t *h0 = o->shares0 + (size_t)e * N;
    if (o->cfg.initial) {
        /* initial_amount + np.sum(np.array(shares)*np.array(prices)), :85-91 / :364-370 */
        double prod[128];
        if (N <= 128) {
            for (int i = 0; i < N; i++) prod[i] = (double)h0[i] * p[i];
            return o->cash0[e] + np_sum_small(prod, N);
        }
    }
    /* previous_state[0] + builtin sum(prices*prev_shares), :372-378 */
    double s = 0.0;
    for (int i = 0; i < N; i++) s = s + p[i] * (double)h0[i];
    return o->cash0[e] + s;
}

/* __init__ :48-100 (+ _initiate_state :398-451): all envs start at `day0`. */
void stock_oracle_init(stock_oracle *o, const double *cash0, const int64_t *shares0, int day0)
{
    const size_t E = o->cfg.n_envs, N = o->cfg.n_tickers;
    memcpy(o->cash0, cash0, E * sizeof(double));
    memcpy(o->shares0, shares0, E * N * sizeof(int64_t));
    memcpy(o->shares, shares0, E * N * sizeof(int64_t));
    /* single stock, initial=True: state shares are `[0] * stock_dim` whatever
     * num_stock_shares says (:415-422) -- asset_memory[0] still counts them (:85-91) */
    if (o->cfg.single_ticker && o->cfg.initial) memset(o->shares, 0, E * N * sizeof(int64_t));
    for (size_t e = 0; e < E; e++) {
        stock_env *s = &o->env[e];
        memset(s, 0, sizeof(*s));
        s->cash = cash0[e];
        s->day = day0;
        s->price_day = day0;
        s->asset0 = initial_asset(o, (int)e, day0);
        s->prev_asset = s->asset0;
    }
}

/* OBS(cash, close, shares, tech) = [cash | close | shares | tech_0.. | tech_{K-1}..], :456-467 */
static void write_obs(const stock_oracle *o, int e, int row, double *obs)
{
    const int N = o->cfg.n_tickers, K = o->cfg.n_tech;
    const double *p = o->close + (size_t)row * N;
    const double *t = o->tech + (size_t)row * K * N;
    const int64_t *h = o->shares + (size_t)e * N;
    obs[0] = o->env[e].cash;
    for (int i = 0; i < N; i++) obs[1 + i] = p[i];
    for (int i = 0; i < N; i++) obs[1 + N + i] = (double)h[i];
    for (int j = 0; j < K * N; j++) obs[1 + 2 * N + j] = t[j];
}

int stock_oracle_obs_dim(const stock_oracle *o)
{
    return 1 + 2 * o->cfg.n_tickers + o->cfg.n_tech * o->cfg.n_tickers;
}

/* reset(), :359-393.  obs is built from the row currently held (`price_day`)
 * BEFORE day is rewound (stale-row quirk, App. B-3) unless reset_quirk == 0. */
void stock_oracle_reset_env(stock_oracle *o, int e, double *obs /* [D] or NULL */)
{
    const int N = o->cfg.n_tickers;
    stock_env *s = &o->env[e];
    if (!o->cfg.reset_quirk) s->price_day = 0;
    s->cash = o->cash0[e];
    if (o->cfg.single_ticker && o->cfg.initial)                              /* :415-422 */
        memset(o->shares + (size_t)e * N, 0, N * sizeof(int64_t));
    else
        memcpy(o->shares + (size_t)e * N, o->shares0 + (size_t)e * N, N * sizeof(int64_t));
    s->asset0 = initial_asset(o, e, s->price_day);        /* num_stock_shares, :364-370 */
    s->prev_asset = s->asset0;
    s->ret_sum = 0.0; s->ret_sumsq = 0.0; s->n_ret = 0;
    if (obs) write_obs(o, e, s->price_day, obs);
    s->day = 0;
    s->turbulence = 0.0;
    s->cost = 0.0;
    s->trades = 0;
    s->terminal = 0;
    s->episode += 1;
}

/* Stable ascending argsort of int64 keys (canonical order, App. B-1). */
static void stable_argsort(const int64_t *a, int n, int64_t *order)
{
    for (int i = 0; i < n; i++) {
        int j = i;
        while (j > 0 && a[order[j - 1]] > a[i]) { order[j] = order[j - 1]; j--; }
        order[j] = i;
    }
}

/* step(), :220-357, one env.  realised: [N] traded share counts written back into
 * `actions` by the reference (:324, :330) or NULL. */
void stock_oracle_step_env(stock_oracle *o, int e, const float *act, double *obs, double *reward,
                           uint8_t *done, int64_t *realised)
{
    const stock_cfg *c = &o->cfg;
    const int N = c->n_tickers, T = c->n_days;
    stock_env *s = &o->env[e];
    int64_t *h = o->shares + (size_t)e * N;
    int64_t *a = o->scratch, *order = o->scratch + N;

    s->terminal = s->day >= T - 1;                                           /* :221 */
    if (s->terminal) {                                                       /* :301 */
        if (obs) write_obs(o, e, s->price_day, obs);
        *reward = s->last_reward;
        *done = 1;
        if (realised) memset(realised, 0, N * sizeof(int64_t));
        return;
    }

    const float hmaxf = (float)c->hmax;
    for (int i = 0; i < N; i++) {                                            /* :304-307 */
        volatile float scaled = act[i] * hmaxf;
        a[i] = (int64_t)scaled;
    }
    const int turbulent = c->use_turbulence && s->turbulence >= c->turbulence_threshold;
    if (turbulent)                                                           /* :308-310 */
        for (int i = 0; i < N; i++) a[i] = -(int64_t)c->hmax;

    const double *p = o->close + (size_t)s->price_day * N;     /* prices held in state */
    const double *tech0 = o->tech + (size_t)s->price_day * c->n_tech * N;   /* state[1+2N+i] */
    const double begin = total_asset(o, e, s->price_day);                   /* :311-314 */

    stable_argsort(a, N, order);                                             /* :317 */
    int n_neg = 0, n_pos = 0;
    for (int i = 0; i < N; i++) { n_neg += a[i] < 0; n_pos += a[i] > 0; }

    for (int r = 0; r < n_neg; r++) {                                        /* :321-324 */
        const int i = (int)order[r];
        int64_t q = 0;
        if (turbulent) {                                                     /* :139-163 */
            if (p[i] > 0 && h[i] > 0) {
                q = h[i];
                s->cash += p[i] * (double)q * (1 - c->sell_cost_pct);
                h[i] = 0;
                s->cost += p[i] * (double)q * c->sell_cost_pct;
                s->trades += 1;
            }
        } else if (c->n_tech == 0 || tech0[i] != 1.0) {                      /* :105 (`!= True`) */
            if (h[i] > 0) {                                                  /* :110-129 */
                const int64_t want = a[i] < 0 ? -a[i] : a[i];
                q = want < h[i] ? want : h[i];
                s->cash += p[i] * (double)q * (1 - c->sell_cost_pct);
                h[i] -= q;
                s->cost += p[i] * (double)q * c->sell_cost_pct;
                s->trades += 1;
            }
        }
        a[i] = -q;
    }
    for (int r = 0; r < n_pos; r++) {                                        /* :328-330 */
        const int i = (int)order[N - 1 - r];
        int64_t q = 0;
        if (!turbulent && (c->n_tech == 0 || tech0[i] != 1.0)) {             /* :204-211, :174 */
            const double unit = p[i] * (1 + c->buy_cost_pct);
            if (unit > 0.0) {     /* contract: close <= 0 -> no fill (App. B-9) */
                const double avail = floordiv_exact(s->cash, unit);          /* :178-180 */
                const double qd = avail < (double)a[i] ? avail : (double)a[i];  /* :184 */
                q = (int64_t)qd;
                s->cash -= p[i] * qd * (1 + c->buy_cost_pct);               /* :185-190 */
                h[i] += q;                                                   /* :192 */
                s->cost += p[i] * qd * c->buy_cost_pct;                      /* :194-196 */
                s->trades += 1;                                              /* :197 */
            }
        }
        a[i] = q;
    }
    if (realised) memcpy(realised, a, N * sizeof(int64_t));

    s->day += 1;                                                             /* :335-336 */
    s->price_day = s->day;
    if (c->use_turbulence) s->turbulence = o->risk[s->day];                  /* :337-341 */
    if (obs) write_obs(o, e, s->price_day, obs);                             /* :342 */
    const double end = total_asset(o, e, s->price_day);                      /* :344-347 */
    s->last_reward = (end - begin) * c->reward_scaling;                      /* :350-352 */
    {   /* running form of df_total_value.pct_change(1) mean/std, :243-251 */
        const double ret = end / s->prev_asset - 1.0;
        s->n_ret += 1;
        s->ret_sum = s->ret_sum + ret;
        s->ret_sumsq = s->ret_sumsq + ret * ret;
        s->prev_asset = end;
    }
    *reward = s->last_reward;
    *done = 0;
}

/* Whole batch, gym semantics (no auto-reset). act [E][N], obs [E][D]|NULL. */
void stock_oracle_step(stock_oracle *o, const float *act, double *obs, double *reward,
                       uint8_t *done, int64_t *realised)
{
    const int E = o->cfg.n_envs, N = o->cfg.n_tickers, D = stock_oracle_obs_dim(o);
    for (int e = 0; e < E; e++)
        stock_oracle_step_env(o, e, act + (size_t)e * N, obs ? obs + (size_t)e * D : NULL,
                              reward + e, done + e, realised ? realised + (size_t)e * N : NULL);
}

/* Whole batch, SB3 DummyVecEnv semantics: on done, term_obs <- obs, obs <- reset(). */
void stock_oracle_vec_step(stock_oracle *o, const float *act, double *obs, double *reward,
                           uint8_t *done, double *term_obs)
{
    const int E = o->cfg.n_envs, N = o->cfg.n_tickers, D = stock_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) {
        double *ob = obs ? obs + (size_t)e * D : NULL;
        stock_oracle_step_env(o, e, act + (size_t)e * N, ob, reward + e, done + e, NULL);
        if (done[e]) {
            if (term_obs && ob) memcpy(term_obs + (size_t)e * D, ob, D * sizeof(double));
            stock_oracle_reset_env(o, e, ob);
        }
    }
}

void stock_oracle_reset(stock_oracle *o, double *obs)
{
    const int E = o->cfg.n_envs, D = stock_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) stock_oracle_reset_env(o, e, obs ? obs + (size_t)e * D : NULL);
}

/* State readback (SoA views for the parity tests). */
void stock_oracle_get_state(const stock_oracle *o, double *cash, int64_t *shares, int32_t *day,
                            int32_t *price_day, double *cost, int32_t *trades,
                            double *last_reward, double *turbulence, int32_t *episode)
{
    const size_t E = o->cfg.n_envs, N = o->cfg.n_tickers;
    for (size_t e = 0; e < E; e++) {
        const stock_env *s = &o->env[e];
        if (cash) cash[e] = s->cash;
        if (day) day[e] = s->day;
        if (price_day) price_day[e] = s->price_day;
        if (cost) cost[e] = s->cost;
        if (trades) trades[e] = s->trades;
        if (last_reward) last_reward[e] = s->last_reward;
        if (turbulence) turbulence[e] = s->turbulence;
        if (episode) episode[e] = s->episode;
    }
    if (shares) memcpy(shares, o->shares, E * N * sizeof(int64_t));
}

/* Episode summary the reference prints at the terminal step (:226-264):
 * out[e] = {begin_total_asset, end_total_asset, total_reward, total_cost,
 *           total_trades, sharpe (NaN when std == 0 or < 2 returns)}. */
void stock_oracle_episode_stats(const stock_oracle *o, double *out /* [E][6] */)
{
    const int E = o->cfg.n_envs;
    for (int e = 0; e < E; e++) {
        const stock_env *s = &o->env[e];
        const double end = total_asset(o, e, s->price_day);
        double *r = out + (size_t)e * 6;
        r[0] = s->asset0;
        r[1] = end;
        r[2] = end - s->asset0;
        r[3] = s->cost;
        r[4] = (double)s->trades;
        r[5] = NAN;
        if (s->n_ret >= 2) {    /* sqrt(252) * mean / std(ddof=1) from the running sums */
            const double mean = s->ret_sum / (double)s->n_ret;
            const double var = (s->ret_sumsq - s->ret_sum * mean) / (double)(s->n_ret - 1);
            if (var > 0.0) r[5] = sqrt(252.0) * mean / sqrt(var);
        }
    }
}
