// finenv_crypto.hip -- MI355X (gfx950) kernel + C ABI for the batched multi-crypto env.
//
// Replaces finrl/meta/env_cryptocurrency_trading/env_multiple_crypto.py step() :59-90,
// reset() :48-57, get_state() :92-98 for E independent envs per launch.
//
// lane = env, one wave per 64 envs, four independent waves per block (no block barriers).
// Trades run in asset-index order (no sort in this env), serial through cash in fp64 with the
// reference's operation order; `cash // price` is the exact floor (reciprocal + FMA-remainder
// fix-up); stocks are float32 and fractional as in the reference.  Small rows (51 floats at
// 10 pairs x 4 indicators): 385 algorithmic bytes per env-step, HBM-bound.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "finenv.h"
#include "finenv_dev.h"
#include "finenv_host.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxN = FINENV_CRYPTO_MAX_ASSETS;
constexpr int kRow = kMaxN + 1;                 // odd row stride (dwords)
constexpr int kWaves = 4;
constexpr int kLdsPerWave = kWave * kRow + kMaxN * kWave;   // rows + stocks [i][lane]

struct CrParams {
    finenv_crypto_config cfg;
    finenv_crypto_panel panel;
    finenv_crypto_state st;
    const float *actions;
    float *obs;
    float *reward;
    uint8_t *done;
    float *term_obs;
    const uint8_t *mask;
    int32_t auto_reset;
    int32_t D;
    uint32_t magicN;
    uint32_t magicW;
};

#define CF(fld) (*at(p.st.f64, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define CI(fld) (*at(p.st.i32, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define STK(i) (*at(p.st.stocks, (unsigned)(i) * (unsigned)E + (unsigned)e))

__device__ __forceinline__ double cr_floordiv(double a, double d)   // exact floor(a/d), d > 0
{
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    double q = floor(a * x);
    double r = fma(-q, d, a);
    // estimate within 1 of the true floor for |a/d| < 2^40; two fix-up rounds for safety
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double adj = (r < 0.0) ? -1.0 : ((r >= d) ? 1.0 : 0.0);
        q += adj;
        r = fma(-q, d, a);
    }
    return q;
}

// rows[el*kRow + 0] = f32(cash * 2^-18), rows[el*kRow + 1 + i] = stocks_i * 2^-3   (:93)
// columns >= 1 + N: tech_scaled[(t_el - l) * W + j]                              (:94-97)
__device__ __forceinline__ void cr_write_rows(float *__restrict__ dst, const CrParams &p,
                                              int e0, int nenv_w, int t_row,
                                              unsigned long long lane_mask, const float *rows,
                                              int lane)
{
    const int N = p.cfg.n_assets, W = p.cfg.n_tech, D = p.D;
    const unsigned magicW = p.magicW;
    write_obs_rows_generic<8, 16>(
        dst, p.panel.tech_scaled, D, e0, nenv_w, t_row, lane_mask, rows, kRow, lane,
        [=](int t, int col) {                                // lookback row l, indicator j
            const int c2 = col - 1 - N;
            const int l = (W == 1) ? c2 : (int)__umulhi((unsigned)c2, magicW);
            return (t - l) * W + (c2 - l * W);
        },
        [=](int col) { return col < 1 + N ? col : -1; });
}

template <bool RESET_ONLY>
__global__ void __launch_bounds__(kWave *kWaves) crypto_kernel(const CrParams p)
{
    __shared__ float lds_all[kWaves * kLdsPerWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    float *rows = lds_all + wib * kLdsPerWave;              // [env][kRow]: actions, then obs heads
    float *stk = rows + kWave * kRow;                       // [asset][lane]
    const int E = p.cfg.n_envs, N = p.cfg.n_assets;
    const int e0 = (blockIdx.x * kWaves + wib) * kWave;
    if (e0 >= E) return;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;
    float *row = rows + lane * kRow;

    if (RESET_ONLY) {                                       // reset(), :48-57
        const bool sel = valid && (p.mask == nullptr || p.mask[e] != 0);
        const int t = p.cfg.lookback - 1;
        if (sel) {
            CI(FINENV_CI_TIME) = t;
            CF(FINENV_CF_CASH) = p.cfg.initial_cash;
            CF(FINENV_CF_TOTAL_ASSET) = p.cfg.initial_cash;
            for (int i = 0; i < N; ++i) STK(i) = 0.0f;
        }
        if (p.obs == nullptr) return;
        row[0] = (float)(p.cfg.initial_cash * 0x1p-18);
        for (int i = 0; i < N; ++i) row[1 + i] = 0.0f;
        wave_sync();
        cr_write_rows(p.obs, p, e0, nenv_w, t, __ballot(sel), rows, lane);
        return;
    }

    // ---- action tile [nenv_w][N]: coalesced read, transposed through LDS ------------------
    stage_action_tile(rows, kRow, p.actions + (size_t)e0 * N, nenv_w, N, p.magicN, lane);
    double cash = CF(FINENV_CF_CASH);
    const double prev_asset = CF(FINENV_CF_TOTAL_ASSET);
    double gamma_ret = CF(FINENV_CF_GAMMA_RETURN);
    const int time = CI(FINENV_CI_TIME) + 1;                                  // :60
    const int max_step = p.cfg.n_steps - p.cfg.lookback - 1;                  // :24
    const unsigned pb = (unsigned)(time * N);
    wave_sync();

    // normalised actions (f32 <- f64 product, :63-65) and holdings -> LDS
    // (global loads in batches of kPB issued before their first use: a rolled loop exposes one
    //  HBM round trip per asset at few waves per SIMD)
    constexpr int kPB = 16;
    for (int i0 = 0; i0 < N; i0 += kPB) {
        float sv[kPB];
#pragma unroll
        for (int j = 0; j < kPB; ++j) sv[j] = STK(min(i0 + j, N - 1));
#pragma unroll
        for (int j = 0; j < kPB; ++j) pin(sv[j]);
#pragma unroll
        for (int j = 0; j < kPB; ++j) {
            const int i = i0 + j;
            if (i >= N) break;
            row[i] = (float)((double)row[i] * p.panel.norm[i]);
            stk[i * kWave + lane] = sv[j];
        }
    }
    const double one_m_cs = 1 - p.cfg.sell_cost_pct, one_p_cb = 1 + p.cfg.buy_cost_pct;
    for (int i0 = 0; i0 < N; i0 += kPB) {                                     // sells :67-71
        double prb[kPB];
#pragma unroll
        for (int j = 0; j < kPB; ++j) prb[j] = *at(p.panel.price, pb + (unsigned)min(i0 + j, N - 1));
#pragma unroll
        for (int j = 0; j < kPB; ++j) pin(prb[j]);
#pragma unroll
        for (int j = 0; j < kPB; ++j) {
            const int i = i0 + j;
            if (i >= N) break;
            const float a = row[i];
            const double pr = prb[j];
            if (a < 0.0f && pr > 0.0) {
                const float s = stk[i * kWave + lane];
                const float want = -a;
                const float sell = (want < s) ? want : s;                     // min(stocks, -a)
                stk[i * kWave + lane] = s - sell;
                cash += pr * (double)sell * one_m_cs;
            }
        }
    }
    for (int i0 = 0; i0 < N; i0 += kPB) {                                     // buys :73-77
        double prb[kPB];
#pragma unroll
        for (int j = 0; j < kPB; ++j) prb[j] = *at(p.panel.price, pb + (unsigned)min(i0 + j, N - 1));
#pragma unroll
        for (int j = 0; j < kPB; ++j) pin(prb[j]);
#pragma unroll
        for (int j = 0; j < kPB; ++j) {
            const int i = i0 + j;
            if (i >= N) break;
            const float a = row[i];
            const double pr = prb[j];
            if (a > 0.0f && pr > 0.0) {
                const double avail = cr_floordiv(cash, pr);                   // cash // price
                const double buy = ((double)a < avail) ? (double)a : avail;   // min(avail, a)
                const float s = stk[i * kWave + lane];
                stk[i * kWave + lane] = (float)((double)s + buy);
                cash -= pr * buy * one_p_cb;
            }
        }
    }
    const bool done = time == max_step;                                       // :80

    // ---- total asset: cash + np.sum(stocks * price) (NumPy pairwise order), :82 -----------
    auto prod = [&](int i) { return (double)stk[i * kWave + lane] * *at(p.panel.price, pb + (unsigned)i); };
    double sum;
    if (N < 8) {
        sum = 0.0;
        for (int i = 0; i < N; ++i) sum += prod(i);
    } else {
        double r8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[j] = prod(j);
        const int full = N - (N & 7);
        for (int i = 8; i < full; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r8[j] += prod(i + j);
        }
        sum = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
        for (int i = full; i < N; ++i) sum += prod(i);
    }
    const double next = cash + sum;
    double reward = (next - prev_asset) * 0x1p-16;                            // :83
    gamma_ret = gamma_ret * p.cfg.gamma + reward;                             // :85
    if (done) reward = gamma_ret;                                             // :87-88

    // ---- observation heads -> LDS rows; state write-back --------------------------------------
    wave_sync();
    for (int i = 0; i < N; ++i) {
        const float s = stk[i * kWave + lane];
        row[1 + i] = s * 0x1p-3f;
        if (valid) STK(i) = (done && p.auto_reset) ? 0.0f : s;
    }
    row[0] = (float)(cash * 0x1p-18);
    if (valid) {
        *at(p.reward, (unsigned)e) = (float)reward;
        *at(p.done, (unsigned)e) = done ? 1 : 0;
        CF(FINENV_CF_LAST_REWARD) = reward;
        CF(FINENV_CF_GAMMA_RETURN) = gamma_ret;
        if (done) CF(FINENV_CF_EPISODE_RETURN) = next / p.cfg.initial_cash;   // :89
    }
    wave_sync();
    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long done_mask = __ballot(done && valid);
    int t_row = time;
    double cash_out = cash, asset_out = next;
    if (done_mask != 0ull) {
        if (p.term_obs != nullptr)
            cr_write_rows(p.term_obs, p, e0, nenv_w, time, done_mask, rows, lane);
        if (p.auto_reset) {                                                   // reset(), :48-57
            wave_sync();
            if (done) {
                t_row = p.cfg.lookback - 1;
                cash_out = p.cfg.initial_cash;
                asset_out = p.cfg.initial_cash;
                row[0] = (float)(p.cfg.initial_cash * 0x1p-18);
                for (int i = 0; i < N; ++i) row[1 + i] = 0.0f;
            }
            wave_sync();
        }
    }
    cr_write_rows(p.obs, p, e0, nenv_w, t_row, valid_mask, rows, lane);
    if (valid) {
        CF(FINENV_CF_CASH) = cash_out;
        CF(FINENV_CF_TOTAL_ASSET) = asset_out;
        CI(FINENV_CI_TIME) = t_row;
    }
}

}  // namespace

struct finenv_crypto {
    int device;           // HIP device that owns the bound state block (-1 before bind)
    finenv_crypto_config cfg;
    finenv_crypto_panel panel;
    finenv_crypto_state st;
    int bound;
    int D;
    uint32_t magicN, magicW;
    char err[256];
};

namespace {
int cr_fail(finenv_crypto *h, int code, const char *msg)
{
    if (h) snprintf(h->err, sizeof(h->err), "%s", msg);
    return code;
}
int cr_check(finenv_crypto *h, const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        return FINENV_ERR_HIP;
    }
    return FINENV_OK;
}
CrParams cr_params(const finenv_crypto *h)
{
    CrParams p;
    memset(&p, 0, sizeof(p));
    p.cfg = h->cfg;
    p.panel = h->panel;
    p.st = h->st;
    p.D = h->D;
    p.magicN = h->magicN;
    p.magicW = h->magicW;
    return p;
}
uint32_t magic_for(long long n)
{
    return n >= 2 ? (uint32_t)(((1ull << 32) + n - 1) / (unsigned long long)n) : 0u;
}
dim3 cr_grid(int E)
{
    const int waves = (E + kWave - 1) / kWave;
    return dim3((unsigned)((waves + kWaves - 1) / kWaves));
}
}  // namespace

extern "C" {

int finenv_crypto_create(const finenv_crypto_config *cfg, finenv_crypto **out)
{
    if (!cfg || !out) return FINENV_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_envs < 1 || cfg->n_assets < 1 || cfg->n_assets > FINENV_CRYPTO_MAX_ASSETS ||
        cfg->n_tech < 0 || cfg->lookback < 1 || cfg->n_steps < cfg->lookback + 2)
        return FINENV_ERR_INVALID;
    const long long E = cfg->n_envs, N = cfg->n_assets, T = cfg->n_steps, W = cfg->n_tech;
    const long long D = 1 + N + W * cfg->lookback, lim = (1ll << 32) - 1;
    if (E * 8 * FINENV_CRYPTO_F64_FIELDS > lim || E * N * 4 > lim || T * N * 8 > lim ||
        T * W * 4 > lim || 64 * D * 4 > lim || D > 65535)
        return FINENV_ERR_INVALID;
    finenv_crypto *h = new (std::nothrow) finenv_crypto;
    if (!h) return FINENV_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->device = -1;
    h->cfg = *cfg;
    h->D = (int)D;
    h->magicN = magic_for(N);
    h->magicW = magic_for(W);
    *out = h;
    return FINENV_OK;
}

void finenv_crypto_destroy(finenv_crypto *h) { delete h; }
const char *finenv_crypto_last_error(const finenv_crypto *h) { return h ? h->err : "null handle"; }
int finenv_crypto_obs_dim(const finenv_crypto *h) { return h ? h->D : FINENV_ERR_INVALID; }

int finenv_crypto_bind(finenv_crypto *h, const finenv_crypto_panel *panel,
                       const finenv_crypto_state *st)
{
    if (!h || !panel || !st) return FINENV_ERR_INVALID;
    if (!panel->price || (!panel->tech_scaled && h->cfg.n_tech > 0) || !panel->norm || !st->f64 ||
        !st->i32 || !st->stocks)
        return cr_fail(h, FINENV_ERR_INVALID, "bind: null pointer");
    h->panel = *panel;
    h->st = *st;
    h->device = finenv_host::pointer_device(st->f64);
    h->bound = 1;
    return FINENV_OK;
}

int finenv_crypto_reset(finenv_crypto *h, const uint8_t *mask, float *obs_out, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return cr_fail(h, FINENV_ERR_UNBOUND, "reset: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    CrParams p = cr_params(h);
    p.mask = mask;
    p.obs = obs_out;
    hipLaunchKernelGGL((crypto_kernel<true>), cr_grid(h->cfg.n_envs), dim3(kWave * kWaves), 0,
                       (hipStream_t)stream, p);
    return cr_check(h, "crypto_reset");
}

int finenv_crypto_step(finenv_crypto *h, const float *actions, float *obs, float *reward,
                       uint8_t *done, float *term_obs, int32_t auto_reset, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return cr_fail(h, FINENV_ERR_UNBOUND, "step: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    if (!actions || !obs || !reward || !done)
        return cr_fail(h, FINENV_ERR_INVALID, "step: null actions/obs/reward/done");
    CrParams p = cr_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.auto_reset = auto_reset;
    hipLaunchKernelGGL((crypto_kernel<false>), cr_grid(h->cfg.n_envs), dim3(kWave * kWaves), 0,
                       (hipStream_t)stream, p);
    return cr_check(h, "crypto_step");
}

}  // extern "C"
