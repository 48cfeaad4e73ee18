/*
 * oracle/crypto_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's CryptoEnv
 * (finrl/meta/env_cryptocurrency_trading/env_multiple_crypto.py: step :59-90, reset :48-57,
 * get_state :92-98, _generate_action_normalizer :103-111).  Contract: price_array and
 * tech_array are float64 (what the reference's data processors produce); stocks are float32
 * (:53), cash and assets float64.
 *
 * Parity status: PINNED, bit-exact, by outputs of the unmodified reference run in the build
 * container (tests/golden/crypto_*.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t n_envs, n_assets, n_tech, n_steps, lookback, reserved;
    double initial_cash, buy_cost_pct, sell_cost_pct, gamma;
} cr_cfg;

typedef struct {
    cr_cfg cfg;
    const double *price;      /* [T][N] */
    const double *tech;       /* [T][W] */
    double *norm;             /* [N] action_norm_vector, :103-111 */
    double *cash, *total_asset, *gamma_return, *episode_return;   /* [E] */
    int32_t *time;            /* [E] */
    float *stocks;            /* [E][N] */
} cr_oracle;

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

static double np_sum_f64(const double *a, int n)     /* NumPy pairwise sum, n < 128 */
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

int cr_oracle_obs_dim(const cr_oracle *o)
{
    return 1 + o->cfg.n_assets + o->cfg.n_tech * o->cfg.lookback;
}

cr_oracle *cr_oracle_create(const cr_cfg *cfg, const double *price, const double *tech)
{
    cr_oracle *o = (cr_oracle *)calloc(1, sizeof(*o));
    const size_t E = cfg->n_envs, N = cfg->n_assets;
    o->cfg = *cfg;
    o->price = price; o->tech = tech;
    o->norm = (double *)calloc(N, sizeof(double));
    for (size_t i = 0; i < N; i++) {                 /* :103-111 */
        const double x = floor(log10(price[i]));     /* math.floor(math.log(price, 10)) */
        o->norm[i] = (1.0 / pow(10.0, x)) * 10000.0;
    }
    o->cash = (double *)calloc(E, sizeof(double));
    o->total_asset = (double *)calloc(E, sizeof(double));
    o->gamma_return = (double *)calloc(E, sizeof(double));
    o->episode_return = (double *)calloc(E, sizeof(double));
    o->time = (int32_t *)calloc(E, sizeof(int32_t));
    o->stocks = (float *)calloc(E * N, sizeof(float));
    for (size_t e = 0; e < E; e++) {                 /* __init__ :26-35 */
        o->time[e] = cfg->lookback - 1;
        o->cash[e] = cfg->initial_cash;
        o->total_asset[e] = cfg->initial_cash;
    }
    return o;
}

void cr_oracle_set_norm(cr_oracle *o, const double *norm)    /* take Python's own vector */
{
    memcpy(o->norm, norm, sizeof(double) * o->cfg.n_assets);
}

void cr_oracle_destroy(cr_oracle *o)
{
    if (!o) return;
    free(o->norm); free(o->cash); free(o->total_asset); free(o->gamma_return);
    free(o->episode_return); free(o->time); free(o->stocks); free(o);
}

static void write_obs(const cr_oracle *o, int e, float *obs)           /* :92-98 */
{
    const int N = o->cfg.n_assets, W = o->cfg.n_tech, L = o->cfg.lookback;
    const float *st = o->stocks + (size_t)e * N;
    obs[0] = (float)(o->cash[e] * 0x1p-18);
    for (int i = 0; i < N; i++) obs[1 + i] = st[i] * 0x1p-3f;
    for (int l = 0; l < L; l++) {
        const double *t = o->tech + (size_t)(o->time[e] - l) * W;
        for (int j = 0; j < W; j++) obs[1 + N + l * W + j] = (float)(t[j] * 0x1p-15);
    }
}

void cr_oracle_reset_env(cr_oracle *o, int e, float *obs)              /* :48-57 */
{
    const int N = o->cfg.n_assets;
    o->time[e] = o->cfg.lookback - 1;
    o->cash[e] = o->cfg.initial_cash;
    memset(o->stocks + (size_t)e * N, 0, sizeof(float) * N);
    o->total_asset[e] = o->cash[e];        /* cash + (zeros * price).sum() */
    if (obs) write_obs(o, e, obs);
}

void cr_oracle_step_env(cr_oracle *o, int e, const float *act, float *obs, double *reward,
                        uint8_t *done)
{
    const cr_cfg *c = &o->cfg;
    const int N = c->n_assets;
    const int max_step = c->n_steps - c->lookback - 1;                 /* :24 */
    float *st = o->stocks + (size_t)e * N;
    float a[256];
    double prod[256];
    o->time[e] += 1;                                                   /* :60 */
    const double *price = o->price + (size_t)o->time[e] * N;
    for (int i = 0; i < N; i++) a[i] = (float)((double)act[i] * o->norm[i]);   /* :63-65 */
    for (int i = 0; i < N; i++) {                                      /* :67-71 */
        if (a[i] < 0 && price[i] > 0) {
            const float want = -a[i];
            const float sell = (want < st[i]) ? want : st[i];          /* min(stocks, -a) */
            st[i] = st[i] - sell;
            o->cash[e] += price[i] * (double)sell * (1 - c->sell_cost_pct);
        }
    }
    for (int i = 0; i < N; i++) {                                      /* :73-77 */
        if (a[i] > 0 && price[i] > 0) {
            const double avail = floordiv_exact(o->cash[e], price[i]);
            const double buy = ((double)a[i] < avail) ? (double)a[i] : avail;   /* min(avail, a) */
            st[i] = (float)((double)st[i] + buy);
            o->cash[e] -= price[i] * buy * (1 + c->buy_cost_pct);
        }
    }
    *done = o->time[e] == max_step;                                    /* :80 */
    if (obs) write_obs(o, e, obs);                                     /* :81 */
    for (int i = 0; i < N; i++) prod[i] = (double)st[i] * price[i];
    const double next = o->cash[e] + np_sum_f64(prod, N);              /* :82 */
    double r = (next - o->total_asset[e]) * 0x1p-16;                   /* :83 */
    o->total_asset[e] = next;
    o->gamma_return[e] = o->gamma_return[e] * c->gamma + r;            /* :85 */
    if (*done) {                                                       /* :87-89 */
        r = o->gamma_return[e];
        o->episode_return[e] = o->total_asset[e] / c->initial_cash;
    }
    *reward = r;
}

void cr_oracle_vec_step(cr_oracle *o, const float *act, float *obs, double *reward,
                        uint8_t *done, float *term_obs, int auto_reset)
{
    const int E = o->cfg.n_envs, N = o->cfg.n_assets, D = cr_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) {
        float *ob = obs ? obs + (size_t)e * D : NULL;
        cr_oracle_step_env(o, e, act + (size_t)e * N, ob, reward + e, done + e);
        if (done[e] && auto_reset) {
            if (term_obs && ob) memcpy(term_obs + (size_t)e * D, ob, D * sizeof(float));
            cr_oracle_reset_env(o, e, ob);
        }
    }
}

void cr_oracle_reset(cr_oracle *o, float *obs)
{
    const int E = o->cfg.n_envs, D = cr_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) cr_oracle_reset_env(o, e, obs ? obs + (size_t)e * D : NULL);
}

void cr_oracle_get_state(const cr_oracle *o, double *cash, double *total_asset,
                         double *gamma_return, double *episode_return, int32_t *time,
                         float *stocks)
{
    const size_t E = o->cfg.n_envs, N = o->cfg.n_assets;
    memcpy(cash, o->cash, E * 8); memcpy(total_asset, o->total_asset, E * 8);
    memcpy(gamma_return, o->gamma_return, E * 8);
    memcpy(episode_return, o->episode_return, E * 8);
    memcpy(time, o->time, E * 4); memcpy(stocks, o->stocks, E * N * 4);
}

void cr_oracle_get_norm(const cr_oracle *o, double *norm)
{
    memcpy(norm, o->norm, sizeof(double) * o->cfg.n_assets);
}
