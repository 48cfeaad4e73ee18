/*
 * oracle/portfolio_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's StockPortfolioEnv.step/reset
 * (finrl/meta/env_portfolio_allocation/env_portfolio.py :125-200, :202-220, softmax
 * :225-229).  Checker for the HIP path and "port" CPU baseline; see stock_oracle.c for the
 * rules on who may load it.
 *
 * Parity status: PINNED by outputs of the unmodified reference run in the build container
 * (tests/golden/portfolio_*.npz).  The only non-bit-reproducible step is exp() in float32:
 * NumPy's SIMD expf and libm/GPU expf may differ in the last ulp, so weights (f32) can differ
 * by 1 ulp and the fp64 portfolio value by ~1e-8 relative; tests use rel 1e-6 there (the
 * north-star bound is 1e-5) and exact equality everywhere else (obs rows, done, day).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t n_envs, n_tickers, n_tech, n_days;
    double initial_amount;
} pf_cfg;

typedef struct {
    pf_cfg cfg;
    const double *close;     /* [T][N]                                         */
    const double *cov;       /* [T][N][N]  df["cov_list"] of the day           */
    const double *tech;      /* [T][K][N]                                      */
    double *value;           /* [E] self.portfolio_value                        */
    double *last_reward;     /* [E] self.reward                                 */
    int32_t *day;            /* [E]                                             */
} pf_oracle;

pf_oracle *pf_oracle_create(const pf_cfg *cfg, const double *close, const double *cov,
                            const double *tech)
{
    pf_oracle *o = (pf_oracle *)calloc(1, sizeof(*o));
    o->cfg = *cfg;
    o->close = close; o->cov = cov; o->tech = tech;
    o->value = (double *)calloc(cfg->n_envs, sizeof(double));
    o->last_reward = (double *)calloc(cfg->n_envs, sizeof(double));
    o->day = (int32_t *)calloc(cfg->n_envs, sizeof(int32_t));
    for (int e = 0; e < cfg->n_envs; e++) o->value[e] = cfg->initial_amount;
    return o;
}

void pf_oracle_destroy(pf_oracle *o)
{
    if (!o) return;
    free(o->value); free(o->last_reward); free(o->day); free(o);
}

int pf_oracle_obs_dim(const pf_oracle *o)
{
    return (o->cfg.n_tickers + o->cfg.n_tech) * o->cfg.n_tickers;
}

/* state = np.append(cov (N x N), tech rows (K x N), axis=0), :172-179 */
static void write_obs(const pf_oracle *o, int day, double *obs)
{
    const int N = o->cfg.n_tickers, K = o->cfg.n_tech;
    memcpy(obs, o->cov + (size_t)day * N * N, sizeof(double) * N * N);
    memcpy(obs + N * N, o->tech + (size_t)day * K * N, sizeof(double) * K * N);
}

void pf_oracle_reset_env(pf_oracle *o, int e, double *obs)      /* :202-220 */
{
    o->day[e] = 0;
    o->value[e] = o->cfg.initial_amount;
    if (obs) write_obs(o, 0, obs);
}

/* np.sum over contiguous float32, n < 128: pairwise with 8 accumulators (as stock_oracle.c) */
static float np_sum_f32(const float *a, int n)
{
    float r[8], res;
    int i, j;
    if (n < 8) {
        res = 0.f;
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

void pf_oracle_step_env(pf_oracle *o, int e, const float *act, double *obs, double *reward,
                        uint8_t *done, float *weights_out)
{
    const int N = o->cfg.n_tickers, T = o->cfg.n_days;
    if (o->day[e] >= T - 1) {                                    /* :127, :156 */
        if (obs) write_obs(o, o->day[e], obs);
        *reward = o->last_reward[e];
        *done = 1;
        return;
    }
    float w[256], ex[256];
    for (int i = 0; i < N; i++) ex[i] = expf(act[i]);           /* np.exp(actions), f32 */
    const float den = np_sum_f32(ex, N);                         /* np.sum(np.exp(actions)) */
    for (int i = 0; i < N; i++) w[i] = ex[i] / den;              /* :228 */
    if (weights_out) memcpy(weights_out, w, sizeof(float) * N);
    const double *c0 = o->close + (size_t)o->day[e] * N;
    o->day[e] += 1;                                              /* :172 */
    const double *c1 = o->close + (size_t)o->day[e] * N;
    double ret = 0.0;                                            /* builtin sum, :183-185 */
    for (int i = 0; i < N; i++) ret = ret + ((c1[i] / c0[i]) - 1) * (double)w[i];
    o->value[e] = o->value[e] * (1 + ret);                       /* :187-188 */
    o->last_reward[e] = o->value[e];                             /* :196 */
    if (obs) write_obs(o, o->day[e], obs);
    *reward = o->value[e];
    *done = 0;
}

void pf_oracle_vec_step(pf_oracle *o, const float *act, double *obs, double *reward,
                        uint8_t *done, double *term_obs, int auto_reset)
{
    const int E = o->cfg.n_envs, N = o->cfg.n_tickers, D = pf_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) {
        double *ob = obs ? obs + (size_t)e * D : NULL;
        pf_oracle_step_env(o, e, act + (size_t)e * N, ob, reward + e, done + e, NULL);
        if (done[e] && auto_reset) {
            if (term_obs && ob) memcpy(term_obs + (size_t)e * D, ob, D * sizeof(double));
            pf_oracle_reset_env(o, e, ob);
        }
    }
}

void pf_oracle_reset(pf_oracle *o, double *obs)
{
    const int E = o->cfg.n_envs, D = pf_oracle_obs_dim(o);
    for (int e = 0; e < E; e++) pf_oracle_reset_env(o, e, obs ? obs + (size_t)e * D : NULL);
}

void pf_oracle_get_state(const pf_oracle *o, double *value, int32_t *day)
{
    memcpy(value, o->value, sizeof(double) * o->cfg.n_envs);
    memcpy(day, o->day, sizeof(int32_t) * o->cfg.n_envs);
}
