/*
 * oracle/stocknp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's array-state StockTradingEnv
 * (finrl/meta/env_stock_trading/env_stocktrading_np.py: __init__ :9-78, reset :80-101,
 * step :103-147, get_state :149-162, sigmoid_sign :164-169) -- the ElegantRL / RLlib-facing
 * env.
 *
 * Parity status: PINNED, bit-exact, against the unmodified reference run in the build
 * container under NumPy 2.2.6 (tests/golden/stocknp_*.npz).  Under NumPy >= 2 (NEP 50) the
 * reference's money arithmetic is partly FLOAT32: `amount` starts as a Python float, a
 * Python float combined with a np.float32 scalar yields np.float32, np.float32 with np.int64
 * yields float64, so the precision of amount / total_asset / gamma_reward depends on the
 * trade history.  This restatement tracks that dtype per quantity
 * (TAG_PY weak Python float, TAG_F32, TAG_F64) and performs each operation in the dtype NumPy
 * would pick; that is what "identical to the reference on this toolchain" means here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { TAG_PY = 0, TAG_F32 = 1, TAG_F64 = 2 };

typedef struct { double v; int tag; } num;           /* value (f32-representable if TAG_F32) */

typedef struct {
    int32_t n_envs, n_tickers, n_techw, n_days;
    int32_t max_stock_i;        /* int(max_stock) used for the action scale, :104          */
    int32_t min_action;         /* int(max_stock * min_stock_rate), :111                   */
    double max_stock;           /* :39                                                     */
    double buy_cost_pct, sell_cost_pct, reward_scaling, gamma, initial_capital;
    double obs_amount_floor;    /* 0: get_state shows self.amount (env_stocktrading_np.py:150);
                                   1e4: max(self.amount, 1e4), env_nas100_wrds.py:154          */
} np_cfg;

typedef struct {
    np_cfg cfg;
    const float *price;         /* [T][N]   price_ary (f32), :27                           */
    const float *tech;          /* [T][W]   tech_ary * 2^-7 (f32), :31                     */
    const float *turb;          /* [T]      turbulence_ary (sigmoid_sign * 2^-5, f32), :33 */
    const float *turb_bool;     /* [T]      (turbulence > thresh) as f32, :32              */
    num *amount, *total_asset, *gamma_reward, *initial_total_asset;      /* [E]            */
    double *episode_return;     /* [E] */
    int32_t *day;               /* [E] */
    float *stocks, *cool_down;  /* [E][N] */
    float *stocks0;             /* [E][N] initial_stocks (eval) / drawn stocks (train)      */
    num *amount0;               /* [E]    amount right after reset (tag PY eval, F32 train) */
} np_oracle;

static num mk(double v, int tag) { num r; r.v = v; r.tag = tag; return r; }

/* result dtype of (Python-float-or-np scalar) op (np scalar), NumPy 2 promotion */
static int promote(int a, int b)
{
    if (a == TAG_PY) return b;
    if (b == TAG_PY) return a;
    return (a == TAG_F64 || b == TAG_F64) ? TAG_F64 : TAG_F32;
}
static num n_add(num a, num b)
{
    const int t = promote(a.tag, b.tag);
    if (t == TAG_F32) return mk((double)((float)a.v + (float)b.v), t);
    return mk(a.v + b.v, t);
}
static num n_sub(num a, num b)
{
    const int t = promote(a.tag, b.tag);
    if (t == TAG_F32) return mk((double)((float)a.v - (float)b.v), t);
    return mk(a.v - b.v, t);
}
static num n_mul(num a, num b)
{
    const int t = promote(a.tag, b.tag);
    if (t == TAG_F32) return mk((double)((float)a.v * (float)b.v), t);
    return mk(a.v * b.v, t);
}
static num n_div(num a, num b)
{
    const int t = promote(a.tag, b.tag);
    if (t == TAG_F32) return mk((double)((float)a.v / (float)b.v), t);
    return mk(a.v / b.v, t);
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
static float floordiv32(float a, float b)            /* npy_floor_dividef */
{
    float mod, div, fl;
    if (b == 0.0f) return a / b;
    mod = fmodf(a, b);
    div = (a - mod) / b;
    if (mod != 0.0f && ((b < 0) != (mod < 0))) { mod += b; div -= 1.0f; }
    if (div != 0.0f) { fl = floorf(div); if (div - fl > 0.5f) fl += 1.0f; }
    else fl = copysignf(0.0f, a / b);
    return fl;
}
static num n_floordiv(num a, num b)
{
    const int t = promote(a.tag, b.tag);
    if (t == TAG_F32) return mk((double)floordiv32((float)a.v, (float)b.v), t);
    return mk(floordiv64(a.v, b.v), t);
}

static float np_sum_f32(const float *a, int n)
{
    float r[8], res;
    int i, j;
    if (n < 8) { res = 0.f; for (i = 0; i < n; i++) res += a[i]; return res; }
    for (j = 0; j < 8; j++) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8) for (j = 0; j < 8; j++) r[j] += a[i + j];
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

static float holdings_value(const np_oracle *o, int e, const float *price)  /* (stocks*price).sum() */
{
    const int N = o->cfg.n_tickers;
    float prod[512];
    const float *st = o->stocks + (size_t)e * N;
    for (int i = 0; i < N; i++) prod[i] = st[i] * price[i];
    return np_sum_f32(prod, N);
}

int np_oracle_obs_dim(const np_oracle *o) { return 3 + 3 * o->cfg.n_tickers + o->cfg.n_techw; }

np_oracle *np_oracle_create(const np_cfg *cfg, const float *price, const float *tech,
                            const float *turb, const float *turb_bool)
{
    np_oracle *o = (np_oracle *)calloc(1, sizeof(*o));
    const size_t E = cfg->n_envs, N = cfg->n_tickers;
    o->cfg = *cfg;
    o->price = price; o->tech = tech; o->turb = turb; o->turb_bool = turb_bool;
    o->amount = (num *)calloc(E, sizeof(num));
    o->total_asset = (num *)calloc(E, sizeof(num));
    o->gamma_reward = (num *)calloc(E, sizeof(num));
    o->initial_total_asset = (num *)calloc(E, sizeof(num));
    o->amount0 = (num *)calloc(E, sizeof(num));
    o->episode_return = (double *)calloc(E, sizeof(double));
    o->day = (int32_t *)calloc(E, sizeof(int32_t));
    o->stocks = (float *)calloc(E * N, sizeof(float));
    o->cool_down = (float *)calloc(E * N, sizeof(float));
    o->stocks0 = (float *)calloc(E * N, sizeof(float));
    for (size_t e = 0; e < E; e++) o->amount0[e] = mk(cfg->initial_capital, TAG_PY);
    return o;
}

void np_oracle_destroy(np_oracle *o)
{
    if (!o) return;
    free(o->amount); free(o->total_asset); free(o->gamma_reward); free(o->initial_total_asset);
    free(o->amount0); free(o->episode_return); free(o->day); free(o->stocks);
    free(o->cool_down); free(o->stocks0); free(o);
}

/* Per-env start state: eval mode = (initial_stocks, initial_capital as Python float);
 * train mode = (drawn stocks, capital*U - sum as np.float32), :84-96. */
void np_oracle_set_initial(np_oracle *o, const float *stocks0, const double *amount0,
                           const int32_t *amount0_tag)
{
    const size_t E = o->cfg.n_envs, N = o->cfg.n_tickers;
    memcpy(o->stocks0, stocks0, E * N * sizeof(float));
    for (size_t e = 0; e < E; e++) o->amount0[e] = mk(amount0[e], amount0_tag[e]);
}

static void write_obs(const np_oracle *o, int e, const float *price, float *obs)   /* :149-162 */
{
    const int N = o->cfg.n_tickers, W = o->cfg.n_techw, d = o->day[e];
    const float *st = o->stocks + (size_t)e * N, *cd = o->cool_down + (size_t)e * N;
    /* env_nas100_wrds.py:154: Python's max(self.amount, 1e4) returns self.amount (with its NumPy
     * scalar type) unless 1e4 is strictly larger, in which case the Python float 1e4 */
    const num shown = (o->cfg.obs_amount_floor > 0.0 && o->cfg.obs_amount_floor > o->amount[e].v)
                          ? mk(o->cfg.obs_amount_floor, TAG_PY) : o->amount[e];
    const num a = n_mul(shown, mk(0x1p-12, TAG_PY));               /* ... * 2**-12 */
    obs[0] = (float)a.v;
    obs[1] = o->turb[d];
    obs[2] = o->turb_bool[d];
    for (int i = 0; i < N; i++) obs[3 + i] = price[i] * 0x1p-6f;
    for (int i = 0; i < N; i++) obs[3 + N + i] = st[i] * 0x1p-6f;
    for (int i = 0; i < N; i++) obs[3 + 2 * N + i] = cd[i];
    memcpy(obs + 3 + 3 * N, o->tech + (size_t)d * W, sizeof(float) * W);
}

void np_oracle_reset_env(np_oracle *o, int e, float *obs)          /* :80-101 */
{
    const int N = o->cfg.n_tickers;
    o->day[e] = 0;
    const float *price = o->price;
    memcpy(o->stocks + (size_t)e * N, o->stocks0 + (size_t)e * N, sizeof(float) * N);
    memset(o->cool_down + (size_t)e * N, 0, sizeof(float) * N);
    o->amount[e] = o->amount0[e];
    o->total_asset[e] = n_add(o->amount[e], mk((double)holdings_value(o, e, price), TAG_F32));
    o->initial_total_asset[e] = o->total_asset[e];
    o->gamma_reward[e] = mk(0.0, TAG_PY);
    if (obs) write_obs(o, e, price, obs);
}

void np_oracle_step_env(np_oracle *o, int e, const float *act, float *obs, double *reward,
                        int32_t *reward_tag, uint8_t *done)
{
    const np_cfg *c = &o->cfg;
    const int N = c->n_tickers;
    float *st = o->stocks + (size_t)e * N, *cd = o->cool_down + (size_t)e * N;
    int64_t a[512];
    const float ms = (float)c->max_stock;
    for (int i = 0; i < N; i++) { volatile float x = act[i] * ms; a[i] = (int64_t)x; }   /* :104 */
    o->day[e] += 1;                                                               /* :106 */
    const float *price = o->price + (size_t)o->day[e] * N;
    for (int i = 0; i < N; i++) cd[i] += 1.0f;                                    /* :108 */
    const num one_m = mk(1 - c->sell_cost_pct, TAG_PY), one_p = mk(1 + c->buy_cost_pct, TAG_PY);
    if (o->turb_bool[o->day[e]] == 0.0f) {                                        /* :110 */
        for (int i = 0; i < N; i++) {                                             /* :112-119 */
            if (a[i] < -c->min_action && price[i] > 0) {
                const int64_t want = -a[i];
                num sell;            /* min(stocks[i] (f32), -a (int64)) */
                if ((double)want < (double)st[i]) sell = mk((double)want, TAG_F64 + 10);
                else sell = mk((double)st[i], TAG_F32);
                const int is_int = sell.tag > TAG_F64;
                st[i] = (float)((double)st[i] - sell.v);
                /* price(f32) * sell: int64 operand -> float64 result, f32 operand -> f32 */
                num t = is_int ? mk((double)price[i] * sell.v, TAG_F64)
                               : n_mul(mk((double)price[i], TAG_F32), sell);
                t = n_mul(t, one_m);
                o->amount[e] = n_add(o->amount[e], t);
                cd[i] = 0.0f;
            }
        }
        for (int i = 0; i < N; i++) {                                             /* :120-129 */
            if (a[i] > c->min_action && price[i] > 0) {
                const num q = n_floordiv(o->amount[e], mk((double)price[i], TAG_F32));
                num t;
                double buy;
                if ((double)a[i] < q.v) {       /* min(q, a) -> a (np.int64) */
                    buy = (double)a[i];
                    t = mk((double)price[i] * buy, TAG_F64);
                } else {                        /* -> q (f32 or f64) */
                    buy = q.v;
                    t = n_mul(mk((double)price[i], TAG_F32), q);
                }
                st[i] = (float)((double)st[i] + buy);
                t = n_mul(t, one_p);
                o->amount[e] = n_sub(o->amount[e], t);
                cd[i] = 0.0f;
            }
        }
    } else {                                                                      /* :131-134 */
        num t = n_mul(mk((double)holdings_value(o, e, price), TAG_F32), one_m);
        o->amount[e] = n_add(o->amount[e], t);
        for (int i = 0; i < N; i++) { st[i] = 0.0f; cd[i] = 0.0f; }
    }
    if (obs) write_obs(o, e, price, obs);                                         /* :136 */
    const num ta = n_add(o->amount[e], mk((double)holdings_value(o, e, price), TAG_F32));
    num r = n_mul(n_sub(ta, o->total_asset[e]), mk(c->reward_scaling, TAG_PY));   /* :138 */
    o->total_asset[e] = ta;
    o->gamma_reward[e] = n_add(n_mul(o->gamma_reward[e], mk(c->gamma, TAG_PY)), r);   /* :141 */
    *done = o->day[e] == c->n_days - 1;                                           /* :142 */
    if (*done) {
        r = o->gamma_reward[e];
        o->episode_return[e] = n_div(ta, o->initial_total_asset[e]).v;            /* :145 */
    }
    *reward = r.v;
    if (reward_tag) *reward_tag = r.tag;
}

void np_oracle_vec_step(np_oracle *o, const float *act, float *obs, double *reward,
                        uint8_t *done, float *term_obs, int auto_reset)
{
    const int E = o->cfg.n_envs, N = o->cfg.n_tickers, D = np_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) {
        float *ob = obs ? obs + (size_t)e * D : NULL;
        np_oracle_step_env(o, e, act + (size_t)e * N, ob, reward + e, NULL, done + e);
        if (done[e] && auto_reset) {
            if (term_obs && ob) memcpy(term_obs + (size_t)e * D, ob, D * sizeof(float));
            np_oracle_reset_env(o, e, ob);
        }
    }
}

void np_oracle_reset(np_oracle *o, float *obs)
{
    const int E = o->cfg.n_envs, D = np_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) np_oracle_reset_env(o, e, obs ? obs + (size_t)e * D : NULL);
}

void np_oracle_get_state(const np_oracle *o, double *amount, int32_t *amount_tag,
                         double *total_asset, int32_t *ta_tag, double *gamma_reward,
                         int32_t *g_tag, double *episode_return, int32_t *day, float *stocks,
                         float *cool_down)
{
    const size_t E = o->cfg.n_envs, N = o->cfg.n_tickers;
    for (size_t e = 0; e < E; e++) {
        amount[e] = o->amount[e].v; amount_tag[e] = o->amount[e].tag;
        total_asset[e] = o->total_asset[e].v; ta_tag[e] = o->total_asset[e].tag;
        gamma_reward[e] = o->gamma_reward[e].v; g_tag[e] = o->gamma_reward[e].tag;
        episode_return[e] = o->episode_return[e]; day[e] = o->day[e];
    }
    memcpy(stocks, o->stocks, E * N * sizeof(float));
    memcpy(cool_down, o->cool_down, E * N * sizeof(float));
}
