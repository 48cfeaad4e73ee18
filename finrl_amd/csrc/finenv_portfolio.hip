// finenv_portfolio.hip -- MI355X (gfx950) kernel + C ABI for the batched StockPortfolioEnv.
//
// Replaces finrl/meta/env_portfolio_allocation/env_portfolio.py step() :125-200,
// reset() :202-220, softmax_normalization :225-229 for E independent envs per launch.
//
// The per-env arithmetic is tiny (softmax over N scores, one N-term fp64 dot product, one
// multiply); the step is a pure HBM write stream: every env receives the day's
// (N+K) x N observation block (4560 B at DOW30 x 8), 95 % of all bytes.  Roofline: HBM.
//   * lane = env for the arithmetic; the [64][N] action tile is read coalesced and
//     transposed through LDS (row stride odd: conflict-free);
//   * one 128-thread block per 64 envs: both waves stream observation rows (32 rows each)
//     with 16-byte stores in row-major order; wave 0 also does the arithmetic and the state;
//   * when all 64 envs sit on the same day (always, in lock-step batches) the template row
//     lives in registers and the row loop holds no load.

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
constexpr int kMaxN = FINENV_PORTFOLIO_MAX_TICKERS;
constexpr int kTileStride = kMaxN + 1;                 // odd row stride (dwords)
constexpr int kThreads = 2 * kWave;
constexpr int kMaxVecChunks = 8;                       // float4 chunks of a row kept in VGPRs

struct PfParams {
    finenv_portfolio_config cfg;
    finenv_portfolio_panel panel;
    finenv_portfolio_state st;
    const float *actions;
    float *obs;
    float *reward;
    uint8_t *done;
    float *term_obs;
    float *weights;
    const uint8_t *mask;
    int32_t auto_reset;
    int32_t D;
    int32_t mode;          // aux kernel: 1 = reset
    uint32_t magicN;
};

#define PF(fld) (*at(p.st.f64, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define PI(fld) (*at(p.st.i32, (unsigned)(fld) * (unsigned)E + (unsigned)e))

// Stream rows [el_lo, el_hi) of the block's observation tile.  row_day: per-lane panel row.
__device__ __forceinline__ void pf_write_rows(float *__restrict__ dst,
                                              const float *__restrict__ tmpl, int D, int e0,
                                              int el_lo, int el_hi, int row_day,
                                              unsigned long long lane_mask, int lane)
{
    if (el_lo >= el_hi) return;
    unsigned long long sel = lane_mask;
    if (el_hi < 64) sel &= (1ull << el_hi) - 1ull;
    sel &= ~((1ull << el_lo) - 1ull);
    if (sel == 0ull) return;
    const int first = __builtin_ctzll(sel);
    const int rd0 = __builtin_amdgcn_readlane(row_day, first);
    const bool mine = (sel >> lane) & 1ull;
    const bool uniform_row = __all(!mine || row_day == rd0);
    unsigned long long want = (el_hi >= 64 ? ~0ull : (1ull << el_hi) - 1ull) &
                              ~((1ull << el_lo) - 1ull);
    const bool all_rows = sel == want;
    float *const base = dst + (size_t)e0 * D;
    const int n4 = D >> 2;
    const int nchunk4 = (n4 + kWave - 1) / kWave;

    if ((D & 3) == 0 && uniform_row && all_rows && nchunk4 <= kMaxVecChunks) {
        // fast path: 16-byte stores, row-major, template in registers (rows are 16-B aligned
        // because D % 4 == 0 and the tile base is 256-B aligned)
        float4 t[kMaxVecChunks];
#pragma unroll
        for (int j = 0; j < kMaxVecChunks; ++j) {
            const int c4 = j * kWave + lane;
            t[j] = (c4 < n4) ? *at(reinterpret_cast<const float4 *>(tmpl),
                                   (unsigned)(rd0 * n4 + c4))
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 *const base4 = reinterpret_cast<float4 *>(base);
#pragma unroll 2
        for (int el = el_lo; el < el_hi; ++el) {
#pragma unroll
            for (int j = 0; j < kMaxVecChunks; ++j) {
                const int c4 = j * kWave + lane;
                if (j < nchunk4 && c4 < n4) *at(base4, (unsigned)(el * n4 + c4)) = t[j];
            }
        }
        return;
    }
    // general path: per-row template loads (desynchronised days, odd D, masked rows)
    const int nchunk = (D + kWave - 1) / kWave;
    for (int el = el_lo; el < el_hi; ++el) {
        if (!((sel >> el) & 1ull)) continue;
        const int rd = __builtin_amdgcn_readlane(row_day, el);
        for (int k = 0; k < nchunk; ++k) {
            const int col = k * kWave + lane;
            if (col < D) *at(base, (unsigned)(el * D + col)) = *at(tmpl, (unsigned)(rd * D + col));
        }
    }
}

__global__ void __launch_bounds__(kThreads) portfolio_step_kernel(const PfParams p)
{
    __shared__ float tile[kWave * kTileStride];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    const int E = p.cfg.n_envs, N = p.cfg.n_tickers, D = p.D, T = p.cfg.n_days;
    const int e0 = blockIdx.x * kWave;
    if (e0 >= E) return;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;

    // both waves: which panel rows this step shows (needs only `day`)
    int day = PI(FINENV_PI_DAY);
    const bool term = day >= T - 1;                                           // :127
    const int day_next = term ? day : day + 1;
    const int row_obs = (term && p.auto_reset) ? 0 : day_next;               // reset(): day 0
    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long term_mask = __ballot(term && valid);

    // stage the action tile (coalesced), split between the two waves
    {
        const float *__restrict__ src = p.actions + (size_t)e0 * N;
        const int total = nenv_w * N;
        for (int f = threadIdx.x; f < total; f += kThreads) {
            const int el = (N == 1) ? f : (int)__umulhi((unsigned)f, p.magicN);
            tile[el * kTileStride + (f - el * N)] = *at(src, (unsigned)f);
        }
    }
    __syncthreads();      // also orders every wave's read of `day` before wave 0 rewrites it

    if (wib == 0) {
        double value = PF(FINENV_PF_VALUE);
        double last_reward = PF(FINENV_PF_LAST_REWARD);
        float *row = tile + lane * kTileStride;
        if (!term) {
            // softmax in float32 (:225-229): exp, NumPy pairwise sum order, divide
            float r8[8];
            float den;
            if (N < 8) {
                den = 0.f;
                for (int i = 0; i < N; ++i) {
                    const float ex = expf(row[i]);
                    row[i] = ex;
                    den += ex;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    r8[j] = expf(row[j]);
                    row[j] = r8[j];
                }
                const int full = N - (N & 7);
                for (int i = 8; i < full; i += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float ex = expf(row[i + j]);
                        row[i + j] = ex;
                        r8[j] += ex;
                    }
                }
                den = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
                for (int i = full; i < N; ++i) {
                    const float ex = expf(row[i]);
                    row[i] = ex;
                    den += ex;
                }
            }
            // portfolio_return = builtin sum(((close_new/close_old) - 1) * weights), :183-185
            double ret = 0.0;
            const unsigned gb = (unsigned)(day * N);
            for (int i = 0; i < N; ++i) {
                const float w = row[i] / den;                                 // :228
                row[i] = w;
                ret = ret + *at(p.panel.gross_ret, gb + (unsigned)i) * (double)w;
            }
            value = value * (1 + ret);                                        // :187-188
            last_reward = value;                                              // :196
            day = day_next;
            if (p.weights != nullptr && valid)
                for (int i = 0; i < N; ++i) *at(p.weights, (unsigned)(e * N + i)) = row[i];
        }
        if (valid) {
            *at(p.reward, (unsigned)e) = (float)last_reward;
            *at(p.done, (unsigned)e) = term ? 1 : 0;
        }
        if (term && p.auto_reset) {                                           // :202-220
            day = 0;
            value = p.cfg.initial_amount;
        }
        if (valid) {
            PF(FINENV_PF_VALUE) = value;
            PF(FINENV_PF_LAST_REWARD) = last_reward;
            PI(FINENV_PI_DAY) = day;
        }
    }

    // both waves: observation rows (wave 0: rows [0,32), wave 1: rows [32,64))
    const int el_lo = wib * 32, el_hi = min(nenv_w, el_lo + 32);
    if (term_mask != 0ull && p.term_obs != nullptr)
        pf_write_rows(p.term_obs, p.panel.obs_tmpl, D, e0, el_lo, el_hi, day_next, term_mask, lane);
    pf_write_rows(p.obs, p.panel.obs_tmpl, D, e0, el_lo, el_hi, row_obs, valid_mask, lane);
}

__global__ void __launch_bounds__(kThreads) portfolio_reset_kernel(const PfParams p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    const int E = p.cfg.n_envs, D = p.D;
    const int e0 = blockIdx.x * kWave;
    if (e0 >= E) return;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;
    const bool sel = valid && (p.mask == nullptr || p.mask[e] != 0);
    if (wib == 0 && sel) {
        PF(FINENV_PF_VALUE) = p.cfg.initial_amount;
        PI(FINENV_PI_DAY) = 0;
    }
    if (p.obs == nullptr) return;
    const int el_lo = wib * 32, el_hi = min(nenv_w, el_lo + 32);
    pf_write_rows(p.obs, p.panel.obs_tmpl, D, e0, el_lo, el_hi, 0, __ballot(sel), lane);
}

}  // namespace

struct finenv_portfolio {
    int device;           // HIP device that owns the bound state block (-1 before bind)
    finenv_portfolio_config cfg;
    finenv_portfolio_panel panel;
    finenv_portfolio_state st;
    int bound;
    int D;
    uint32_t magicN;
    char err[256];
};

namespace {
int pf_fail(finenv_portfolio *h, int code, const char *msg)
{
    if (h) snprintf(h->err, sizeof(h->err), "%s", msg);
    return code;
}
int pf_check(finenv_portfolio *h, const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        return FINENV_ERR_HIP;
    }
    return FINENV_OK;
}
PfParams pf_params(const finenv_portfolio *h)
{
    PfParams p;
    memset(&p, 0, sizeof(p));
    p.cfg = h->cfg;
    p.panel = h->panel;
    p.st = h->st;
    p.D = h->D;
    p.magicN = h->magicN;
    return p;
}
}  // namespace

extern "C" {

int finenv_portfolio_create(const finenv_portfolio_config *cfg, finenv_portfolio **out)
{
    if (!cfg || !out) return FINENV_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_envs < 1 || cfg->n_tickers < 1 || cfg->n_tickers > FINENV_PORTFOLIO_MAX_TICKERS ||
        cfg->n_tech < 0 || cfg->n_days < 1)
        return FINENV_ERR_INVALID;
    const long long E = cfg->n_envs, N = cfg->n_tickers, T = cfg->n_days;
    const long long D = (N + cfg->n_tech) * N, lim = (1ll << 32) - 1;
    if (E * 8 * FINENV_PORTFOLIO_F64_FIELDS > lim || T * D * 4 > lim || T * N * 8 > lim ||
        E * N * 4 > lim || 64 * D * 4 > lim)
        return FINENV_ERR_INVALID;
    finenv_portfolio *h = new (std::nothrow) finenv_portfolio;
    if (!h) return FINENV_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->device = -1;
    h->cfg = *cfg;
    h->D = (int)D;
    h->magicN = N >= 2 ? (uint32_t)(((1ull << 32) + N - 1) / (unsigned long long)N) : 0u;
    *out = h;
    return FINENV_OK;
}

void finenv_portfolio_destroy(finenv_portfolio *h) { delete h; }
const char *finenv_portfolio_last_error(const finenv_portfolio *h) { return h ? h->err : "null handle"; }
int finenv_portfolio_obs_dim(const finenv_portfolio *h) { return h ? h->D : FINENV_ERR_INVALID; }

int finenv_portfolio_bind(finenv_portfolio *h, const finenv_portfolio_panel *panel,
                          const finenv_portfolio_state *st)
{
    if (!h || !panel || !st) return FINENV_ERR_INVALID;
    if (!panel->gross_ret || !panel->obs_tmpl || !st->f64 || !st->i32)
        return pf_fail(h, FINENV_ERR_INVALID, "bind: null pointer");
    h->panel = *panel;
    h->st = *st;
    h->device = finenv_host::pointer_device(st->f64);
    h->bound = 1;
    return FINENV_OK;
}

int finenv_portfolio_reset(finenv_portfolio *h, const uint8_t *mask, float *obs_out, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return pf_fail(h, FINENV_ERR_UNBOUND, "reset: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    PfParams p = pf_params(h);
    p.mask = mask;
    p.obs = obs_out;
    hipLaunchKernelGGL(portfolio_reset_kernel, dim3((h->cfg.n_envs + kWave - 1) / kWave),
                       dim3(kThreads), 0, (hipStream_t)stream, p);
    return pf_check(h, "portfolio_reset");
}

int finenv_portfolio_step(finenv_portfolio *h, const float *actions, float *obs, float *reward,
                          uint8_t *done, float *term_obs, float *weights_out,
                          int32_t auto_reset, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return pf_fail(h, FINENV_ERR_UNBOUND, "step: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    if (!actions || !obs || !reward || !done)
        return pf_fail(h, FINENV_ERR_INVALID, "step: null actions/obs/reward/done");
    PfParams p = pf_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.weights = weights_out;
    p.auto_reset = auto_reset;
    hipLaunchKernelGGL(portfolio_step_kernel, dim3((h->cfg.n_envs + kWave - 1) / kWave),
                       dim3(kThreads), 0, (hipStream_t)stream, p);
    return pf_check(h, "portfolio_step");
}

}  // extern "C"
