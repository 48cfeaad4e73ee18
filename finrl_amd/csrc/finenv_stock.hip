// finenv_stock.hip -- MI355X (gfx950) kernels + C ABI for the batched StockTradingEnv.
//
// Replaces the per-timestep work of the reference's
//   finrl/meta/env_stock_trading/env_stocktrading.py  step() :220-357, reset() :359-393,
//   _sell_stock :102-169, _buy_stock :171-213, _update_state :453-478
// for E independent environments in one launch.  Not a translation: the reference is a
// Python list + pandas object per env; here the state is a structure-of-arrays in HBM and
// one wavefront lane owns one environment.
//
// Mapping (see DESIGN.md "stock_step"):
//   * lane = env, wave = 64 envs, block = 4 independent waves (no block barriers);
//   * the wave's [64][N] action tile is read coalesced and transposed through LDS;
//   * (action, ticker) pairs become 32 composite int keys per lane, sorted in VGPRs by a
//     191-compare-exchange Batcher network == the reference's stable argsort order;
//   * sells then buys walk the sorted keys; holdings live in LDS as [ticker][lane]
//     (bank = lane, conflict-free under per-lane dynamic ticker index); the cash chain is
//     fp64 with the reference's operation order (-ffp-contract=off), floor division is
//     exact (reciprocal + FMA-remainder correction);
//   * the [64][D] f32 observation block -- 76 % of all bytes -- is streamed out by the
//     whole wave row by row from a pre-packed f32 panel row (L2-resident), patching in
//     cash/holdings from LDS.
// HBM-bound by design (no MFMA: there is no contraction here).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "finenv.h"

namespace {

constexpr int kWave = 64;
constexpr int kNPad = FINENV_STOCK_MAX_TICKERS;          // 32
constexpr int kRow = 33;                                 // LDS row stride in dwords (odd)
constexpr int kWavesPerBlock = 4;
constexpr int kLdsPerWave = kWave * kRow;                // 2112 dwords >= kNPad * kWave
constexpr int kAMax = 1 << 25;                           // |scaled action| clamp (key packing)

static_assert(kLdsPerWave >= kNPad * kWave, "LDS region must hold [ticker][lane] holdings");
static_assert(kRow >= kNPad + 1, "LDS row must hold cash + N holdings");

struct Params {
    finenv_stock_config cfg;
    finenv_stock_panel panel;
    finenv_stock_state st;
    const float *actions;
    float *obs;
    float *reward;
    uint8_t *done;
    float *term_obs;
    int32_t *realised;
    const uint8_t *mask;
    double *stats_out;
    int32_t auto_reset;
    int32_t D;
    int32_t day0;
    uint32_t magicN;      // ceil(2^32 / N) for N >= 2 (exact f / N for f < 2^16)
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void ce(int &a, int &b)
{
    const int lo = min(a, b);
    const int hi = max(a, b);
    a = lo;
    b = hi;
}

// Batcher network on 32 statically indexed VGPRs.
__device__ __forceinline__ void sort32(int (&k)[kNPad])
{
#define CE(i, j) ce(k[i], k[j]);
#include "sortnet32.inc"
#undef CE
}

// Exact floor(a / d) for d > 0 (what NumPy/CPython `//` returns, env_stocktrading.py:178):
// reciprocal + one Newton step gives a quotient within 1 of the true floor for
// |a/d| < 2^40; the FMA remainder (exact sign) fixes it up.
__device__ __forceinline__ double floordiv_exact(double a, double d)
{
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    double q = floor(a * x);
    const double r = fma(-q, d, a);
    if (r < 0.0) q -= 1.0;
    else if (r >= d) q += 1.0;
    return q;
}

// asset_memory[0] (env_stocktrading.py:364-378): initial=True -> initial_amount +
// np.sum(shares*prices) (NumPy pairwise sum, 8 accumulators for 8 <= n < 128);
// initial=False -> previous cash + builtin sum (sequential from 0).
__device__ __forceinline__ double initial_asset(double cash0, const int (&h)[kNPad],
                                                const double *__restrict__ prow, int N,
                                                bool np_sum)
{
    // Branch-free over the 32 static slots (selects, clamped loads): conditional writes to
    // register arrays would turn them into 32-wide vector PHIs and spill.
#define PROD(i) ((double)h[i] * prow[(i) < N ? (i) : 0])
    double res = 0.0;
    if (!np_sum) {
#pragma unroll
        for (int i = 0; i < kNPad; ++i) {
            const double t = res + prow[i < N ? i : 0] * (double)h[i];
            res = (i < N) ? t : res;
        }
    } else if (N < 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double t = res + PROD(i);
            res = (i < N) ? t : res;
        }
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = PROD(j);
        const int full = N - (N & 7);
#pragma unroll
        for (int i = 8; i < kNPad; ++i) {
            const double t = r[i & 7] + PROD(i);
            r[i & 7] = (i < full) ? t : r[i & 7];
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = 8; i < kNPad; ++i) {
            const double t = res + PROD(i);
            res = (i >= full && i < N) ? t : res;
        }
    }
#undef PROD
    return cash0 + res;
}

// Stream observation rows for the envs selected by `lane_mask` (bit el = env e0+el).
//   rows : per-wave LDS, rows[el*kRow + 0] = f32 cash, rows[el*kRow + 1 + i] = f32 holdings_i
//   row_day (per lane el) = panel row whose prices/indicators go into that env's obs.
// Chunk-outer / env-inner: one template register live, re-loaded only when the panel row
// changes (never, in lock-step batches); every store instruction writes 256 contiguous B.
__device__ __forceinline__ void write_obs_rows(float *__restrict__ dst,
                                               const float *__restrict__ tmpl, int D, int N,
                                               int e0, int nenv_w, int row_day,
                                               unsigned long long lane_mask,
                                               const float *rows, int lane)
{
    const int nchunk = (D + kWave - 1) / kWave;
    for (int k = 0; k < nchunk; ++k) {
        const int col = k * kWave + lane;
        const bool in = col < D;
        const int hidx = col - 1 - N;
        const bool patch = in && (col == 0 || (hidx >= 0 && hidx < N));
        const int sel = (col == 0) ? 0 : (patch ? 1 + hidx : 0);
        const bool any_patch = __any(patch);
        int prev_rd = -1;
        float t = 0.0f;
        float *out = dst + (size_t)e0 * D + col;
        for (int el = 0; el < nenv_w; ++el, out += D) {
            if (!((lane_mask >> el) & 1ull)) continue;
            const int rd = __builtin_amdgcn_readlane(row_day, el);
            if (rd != prev_rd) {
                t = in ? tmpl[(size_t)rd * D + col] : 0.0f;
                prev_rd = rd;
            }
            float v = t;
            if (any_patch) {
                const float pv = rows[el * kRow + sel];
                v = patch ? pv : t;
            }
            if (in) *out = v;
        }
    }
}

// -------------------------------------------------------------------------------------
// step(): env_stocktrading.py:220-357 (+ DummyVecEnv auto-reset when p.auto_reset)
// -------------------------------------------------------------------------------------
template <bool TURB, bool STATS>
__global__ void __launch_bounds__(kWave *kWavesPerBlock)
stock_step_kernel(const Params p)
{
    __shared__ float lds_all[kWavesPerBlock * kLdsPerWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    float *lds = lds_all + wib * kLdsPerWave;
    int *ldsh = reinterpret_cast<int *>(lds);            // [ticker][lane] view

    const int E = p.cfg.n_envs, N = p.cfg.n_tickers, D = p.D, T = p.cfg.n_days;
    const int e0 = (blockIdx.x * kWavesPerBlock + wib) * kWave;
    if (e0 >= E) return;                                  // wave-uniform
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;                 // clamped: tail lanes shadow env e0

    // ---- A. action tile [nenv_w][N] f32: coalesced read, transpose through LDS ----------
    {
        const float *__restrict__ src = p.actions + (size_t)e0 * N;
        const int total = nenv_w * N;
        for (int f = lane; f < total; f += kWave) {
            const int el = (N == 1) ? f : (int)__umulhi((unsigned)f, p.magicN);
            const int i = f - el * N;
            lds[el * kRow + i] = src[f];
        }
    }

    // ---- B. per-env scalars ---------------------------------------------------------------
    int day = p.st.day[e];
    int pd = p.st.price_day[e];
    double cash = p.st.cash[e];
    double cost = p.st.cost[e];
    int trades = p.st.trades[e];
    double turb = 0.0;
    if (TURB) turb = p.st.turbulence[e];
    const bool term = day >= T - 1;                                           // :221
    const bool turbulent = TURB && (turb >= p.cfg.turbulence_threshold);      // :308-310
    const int hmax = p.cfg.hmax;
    const float hmaxf = (float)hmax;

    wave_sync();
    int keys[kNPad];
#pragma unroll
    for (int i = 0; i < kNPad; ++i) {
        const float x = lds[lane * kRow + (i < N ? i : 0)] * hmaxf;           // f32 mul, :304
        int a = (int)x;                                                       // trunc, :305
        a = max(-kAMax, min(kAMax, a));
        a = turbulent ? -hmax : a;
        a = (term || i >= N) ? 0 : a;                                         // no trading
        keys[i] = a * kNPad + i;          // unique; order == stable argsort(actions), :317
    }
    wave_sync();

    // ---- C. holdings -> LDS [ticker][lane]; begin_total_asset (:311-314) ------------------
    const double *__restrict__ prow = p.panel.close + (size_t)pd * N;
    double s = 0.0;
    {
        const int *__restrict__ hp = p.st.holdings + e;
#pragma unroll 6
        for (int i = 0; i < N; ++i) {
            const int h = hp[(size_t)i * E];
            ldsh[i * kWave + lane] = h;
            s = s + prow[i] * (double)h;
        }
    }
    const double begin = cash + s;
    const uint32_t untr = p.panel.untradable[pd];

    // ---- D. canonical order ---------------------------------------------------------------
    sort32(keys);

    const double c_s = p.cfg.sell_cost_pct, c_b = p.cfg.buy_cost_pct;
    const double one_m_cs = 1 - c_s, one_p_cb = 1 + c_b;

    // ---- E. sells: most negative first (:318, :321-324, _sell_stock :102-169) -------------
#pragma unroll
    for (int r = 0; r < kNPad; ++r) {
        const int key = keys[r];
        const bool act = key < 0;
        if (!__any(act)) break;
        const int idx = key & (kNPad - 1);
        const int a = key >> 5;
        const int li = act ? idx : 0;
        const double pr = prow[li];
        const int h = ldsh[li * kWave + lane];
        bool ok;
        int q;
        if (turbulent) {                                  // :139-163 (flag ignored)
            ok = act && pr > 0.0 && h > 0;
            q = h;
        } else {                                          // :105-133
            ok = act && !((untr >> idx) & 1u) && h > 0;
            q = min(-a, h);
        }
        if (ok) {
            const double amt = pr * (double)q;
            cash += amt * one_m_cs;
            cost += amt * c_s;
            trades += 1;
            ldsh[li * kWave + lane] = h - q;
        }
    }

    // ---- F. buys: largest first (:319, :328-330, _buy_stock :171-213) ---------------------
#pragma unroll
    for (int r = kNPad - 1; r >= 0; --r) {
        const int key = keys[r];
        const bool act = key >= kNPad;                    // a >= 1
        if (!__any(act)) break;
        const int idx = key & (kNPad - 1);
        const int a = key >> 5;
        const int li = act ? idx : 0;
        const double pr = prow[li];
        const double unit = pr * one_p_cb;                // :179
        const bool ok = act && !turbulent && !((untr >> idx) & 1u) && unit > 0.0;
        // cash >= a*unit exactly  <=>  cash // unit >= a  (then min(avail, a) == a, :184)
        const bool full = fma(-(double)a, unit, cash) >= 0.0;
        double qd = (double)a;
        if (__any(ok && !full)) {
            const double avail = floordiv_exact(cash, unit);                  // :178-180
            qd = full ? qd : avail;
        }
        if (ok) {
            const double amt = pr * qd;
            cash -= amt * one_p_cb;                                           // :185-190
            cost += amt * c_b;                                                // :194-196
            trades += 1;                                                      // :197
            ldsh[li * kWave + lane] += (int)qd;                               // :192
        }
    }

    // ---- G. day += 1, new row, end_total_asset, reward (:335-352) --------------------------
    double last_reward = p.st.last_reward[e];
    if (!term) {
        day += 1;
        pd = day;
        if (TURB) turb = p.panel.risk[day];
    }
    const double *__restrict__ nrow = p.panel.close + (size_t)pd * N;
    int hf[kNPad];
    s = 0.0;
#pragma unroll
    for (int i = 0; i < kNPad; ++i) {
        const int ii = i < N ? i : 0;
        const int h = ldsh[ii * kWave + lane];
        hf[i] = (i < N) ? h : 0;
        const double t = s + nrow[ii] * (double)h;
        s = (i < N) ? t : s;
    }
    const double end = cash + s;
    if (!term) last_reward = (end - begin) * p.cfg.reward_scaling;
    if (valid) {
        p.reward[e] = (float)last_reward;
        p.done[e] = term ? 1 : 0;
    }
    if (STATS && !term) {                    // running pct_change mean / M2, :243-251
        const double prev = p.st.prev_asset[e];
        const int n = p.st.n_ret[e] + 1;
        double mean = p.st.ret_mean[e], m2 = p.st.ret_m2[e];
        const double ret = end / prev - 1.0;
        const double d1 = ret - mean;
        mean += d1 / (double)n;
        m2 += d1 * (ret - mean);
        if (valid) {
            p.st.prev_asset[e] = end;
            p.st.n_ret[e] = n;
            p.st.ret_mean[e] = mean;
            p.st.ret_m2[e] = m2;
        }
    }
    if (p.realised != nullptr) {             // traded shares == holdings delta (:324, :330)
#pragma unroll
        for (int i = 0; i < kNPad; ++i)
            if (i < N && valid)
                p.realised[(size_t)e * N + i] = hf[i] - p.st.holdings[(size_t)i * E + e];
    }

    // ---- H. observation rows in LDS; terminal obs; auto-reset -------------------------------
    wave_sync();
    lds[lane * kRow] = (float)cash;
#pragma unroll
    for (int i = 0; i < kNPad; ++i)
        if (i < N) lds[lane * kRow + 1 + i] = (float)hf[i];
    wave_sync();

    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long term_mask = __ballot(term && valid);
    int episode_inc = 0;
    if (term_mask != 0ull) {                 // wave-uniform, once per episode
        if (p.term_obs != nullptr)
            write_obs_rows(p.term_obs, p.panel.obs_tmpl, D, N, e0, nenv_w, pd, term_mask, lds,
                           lane);
        if (p.auto_reset) {                  // reset(), :359-393
            wave_sync();
            if (term) {
                if (!p.cfg.reset_quirk) pd = 0;
                cash = p.st.cash0[e];
#pragma unroll
                for (int i = 0; i < kNPad; ++i) {
                    const int v = p.st.shares0[(size_t)(i < N ? i : 0) * E + e];
                    hf[i] = (i < N) ? v : 0;
                }
                const double a0 = initial_asset(cash, hf, p.panel.close + (size_t)pd * N, N,
                                                p.cfg.initial != 0);
                if (valid) {
                    p.st.asset0[e] = a0;
                    p.st.prev_asset[e] = a0;
                    p.st.ret_mean[e] = 0.0;
                    p.st.ret_m2[e] = 0.0;
                    p.st.n_ret[e] = 0;
                }
                day = 0;
                turb = 0.0;
                cost = 0.0;
                trades = 0;
                episode_inc = 1;
                lds[lane * kRow] = (float)cash;
#pragma unroll
                for (int i = 0; i < kNPad; ++i)
                    if (i < N) lds[lane * kRow + 1 + i] = (float)hf[i];
            }
            wave_sync();
        }
    }

    // ---- I. next observation [64][D] f32 (:342 / :453-478) ----------------------------------
    write_obs_rows(p.obs, p.panel.obs_tmpl, D, N, e0, nenv_w, pd, valid_mask, lds, lane);

    // ---- J. state write-back ------------------------------------------------------------------
    if (valid) {
        p.st.cash[e] = cash;
        p.st.cost[e] = cost;
        p.st.trades[e] = trades;
        p.st.day[e] = day;
        p.st.price_day[e] = pd;
        p.st.last_reward[e] = last_reward;
        if (TURB) p.st.turbulence[e] = turb;
        if (episode_inc) p.st.episode[e] += 1;
#pragma unroll
        for (int i = 0; i < kNPad; ++i)
            if (i < N) p.st.holdings[(size_t)i * E + e] = hf[i];
    }
}

// -------------------------------------------------------------------------------------
// reset() :359-393 (masked), __init__ state :64-91, render() :395-396
// mode 0 = init (no obs), 1 = reset (masked, obs for reset envs), 2 = observe only
// -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWave *kWavesPerBlock) stock_aux_kernel(const Params p, int mode)
{
    __shared__ float lds_all[kWavesPerBlock * kLdsPerWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    float *lds = lds_all + wib * kLdsPerWave;
    const int E = p.cfg.n_envs, N = p.cfg.n_tickers, D = p.D;
    const int e0 = (blockIdx.x * kWavesPerBlock + wib) * kWave;
    if (e0 >= E) return;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;

    int hf[kNPad];
    double cash;
    int pd;
    bool sel = valid;
    if (mode == 2) {
        cash = p.st.cash[e];
        pd = p.st.price_day[e];
#pragma unroll
        for (int i = 0; i < kNPad; ++i) {
            const int v = p.st.holdings[(size_t)(i < N ? i : 0) * E + e];
            hf[i] = (i < N) ? v : 0;
        }
    } else {
        if (mode == 1 && p.mask != nullptr) sel = valid && p.mask[e] != 0;
        if (mode == 0) pd = p.day0;
        else pd = p.cfg.reset_quirk ? p.st.price_day[e] : 0;
        cash = p.st.cash0[e];
#pragma unroll
        for (int i = 0; i < kNPad; ++i) {
            const int v = p.st.shares0[(size_t)(i < N ? i : 0) * E + e];
            hf[i] = (i < N) ? v : 0;
        }
        const double a0 =
            initial_asset(cash, hf, p.panel.close + (size_t)pd * N, N, p.cfg.initial != 0);
        if (sel) {
            p.st.cash[e] = cash;
#pragma unroll
            for (int i = 0; i < kNPad; ++i)
                if (i < N) p.st.holdings[(size_t)i * E + e] = hf[i];
            p.st.asset0[e] = a0;
            p.st.prev_asset[e] = a0;
            p.st.ret_mean[e] = 0.0;
            p.st.ret_m2[e] = 0.0;
            p.st.n_ret[e] = 0;
            p.st.day[e] = (mode == 0) ? p.day0 : 0;
            p.st.price_day[e] = pd;
            p.st.turbulence[e] = 0.0;
            p.st.cost[e] = 0.0;
            p.st.trades[e] = 0;
            if (mode == 0) {
                p.st.episode[e] = 0;
                p.st.last_reward[e] = 0.0;
            } else {
                p.st.episode[e] += 1;
            }
        }
    }
    if (mode == 0 || p.obs == nullptr) return;
    lds[lane * kRow] = (float)cash;
#pragma unroll
    for (int i = 0; i < kNPad; ++i)
        if (i < N) lds[lane * kRow + 1 + i] = (float)hf[i];
    wave_sync();
    write_obs_rows(p.obs, p.panel.obs_tmpl, D, N, e0, nenv_w, pd, __ballot(sel), lds, lane);
}

// Terminal summary :226-264 from current state: one lane per env.
__global__ void stock_stats_kernel(const Params p)
{
    const int E = p.cfg.n_envs, N = p.cfg.n_tickers;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const double *prow = p.panel.close + (size_t)p.st.price_day[e] * N;
    double s = 0.0;
    for (int i = 0; i < N; ++i) s = s + prow[i] * (double)p.st.holdings[(size_t)i * E + e];
    const double end = p.st.cash[e] + s;
    const double a0 = p.st.asset0[e];
    double *out = p.stats_out + (size_t)e * 6;
    out[0] = a0;
    out[1] = end;
    out[2] = end - a0;
    out[3] = p.st.cost[e];
    out[4] = (double)p.st.trades[e];
    double sharpe = __builtin_nan("");
    const int n = p.st.n_ret[e];
    if (n >= 2) {
        const double sd = sqrt(p.st.ret_m2[e] / (double)(n - 1));
        if (sd != 0.0) sharpe = sqrt(252.0) * p.st.ret_mean[e] / sd;
    }
    out[5] = sharpe;
}

}  // namespace

// =====================================================================================
// Host side: handle, validation, launches.  No allocation on the device, no sync.
// =====================================================================================
struct finenv_stock {
    finenv_stock_config cfg;
    finenv_stock_panel panel;
    finenv_stock_state st;
    int bound;
    int D;
    uint32_t magicN;
    char err[256];
};

namespace {

int fail(finenv_stock *h, int code, const char *fmt, const char *detail = "")
{
    if (h) snprintf(h->err, sizeof(h->err), fmt, detail);
    return code;
}

int check_launch(finenv_stock *h, const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        return FINENV_ERR_HIP;
    }
    return FINENV_OK;
}

Params make_params(const finenv_stock *h)
{
    Params p;
    memset(&p, 0, sizeof(p));
    p.cfg = h->cfg;
    p.panel = h->panel;
    p.st = h->st;
    p.D = h->D;
    p.magicN = h->magicN;
    return p;
}

dim3 grid_for(int E)
{
    const int waves = (E + kWave - 1) / kWave;
    return dim3((unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock));
}

}  // namespace

extern "C" {

int finenv_abi_version(void) { return FINENV_ABI_VERSION; }

const char *finenv_strerror(int code)
{
    switch (code) {
    case FINENV_OK: return "ok";
    case FINENV_ERR_INVALID: return "invalid argument";
    case FINENV_ERR_UNBOUND: return "panel/state not bound";
    case FINENV_ERR_HIP: return "HIP runtime error";
    case FINENV_ERR_NOMEM: return "out of host memory";
    default: return "unknown error";
    }
}

int finenv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return FINENV_ERR_HIP;
    }
    return n;
}

int finenv_stock_create(const finenv_stock_config *cfg, finenv_stock **out)
{
    if (!cfg || !out) return FINENV_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_envs < 1 || cfg->n_tickers < 1 || cfg->n_tickers > FINENV_STOCK_MAX_TICKERS ||
        cfg->n_tech < 0 || cfg->n_days < 1 || cfg->hmax < 0 || cfg->hmax > (1 << 24))
        return FINENV_ERR_INVALID;
    if ((long long)cfg->n_envs * (1 + 2 * cfg->n_tickers + cfg->n_tech * cfg->n_tickers) >
        (1ll << 40))
        return FINENV_ERR_INVALID;
    finenv_stock *h = new (std::nothrow) finenv_stock;
    if (!h) return FINENV_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->D = 1 + 2 * cfg->n_tickers + cfg->n_tech * cfg->n_tickers;
    h->magicN = cfg->n_tickers >= 2
                    ? (uint32_t)(((1ull << 32) + cfg->n_tickers - 1) / (unsigned)cfg->n_tickers)
                    : 0u;
    *out = h;
    return FINENV_OK;
}

void finenv_stock_destroy(finenv_stock *h) { delete h; }

const char *finenv_stock_last_error(const finenv_stock *h) { return h ? h->err : "null handle"; }

int finenv_stock_obs_dim(const finenv_stock *h) { return h ? h->D : FINENV_ERR_INVALID; }

int finenv_stock_bind(finenv_stock *h, const finenv_stock_panel *panel,
                      const finenv_stock_state *st)
{
    if (!h || !panel || !st) return FINENV_ERR_INVALID;
    if (!panel->close || !panel->obs_tmpl || !panel->untradable ||
        (h->cfg.use_turbulence && !panel->risk))
        return fail(h, FINENV_ERR_INVALID, "bind: null panel pointer%s");
    if (!st->cash || !st->holdings || !st->day || !st->price_day || !st->trades ||
        !st->episode || !st->n_ret || !st->cost || !st->last_reward || !st->turbulence ||
        !st->asset0 || !st->prev_asset || !st->ret_mean || !st->ret_m2 || !st->cash0 ||
        !st->shares0)
        return fail(h, FINENV_ERR_INVALID, "bind: null state pointer%s");
    h->panel = *panel;
    h->st = *st;
    h->bound = 1;
    return FINENV_OK;
}

int finenv_stock_init(finenv_stock *h, int32_t day0, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "init: bind first%s");
    if (day0 < 0 || day0 >= h->cfg.n_days) return fail(h, FINENV_ERR_INVALID, "init: bad day0%s");
    Params p = make_params(h);
    p.day0 = day0;
    hipLaunchKernelGGL(stock_aux_kernel, grid_for(h->cfg.n_envs), dim3(kWave * kWavesPerBlock), 0,
                       (hipStream_t)stream, p, 0);
    return check_launch(h, "stock_init");
}

int finenv_stock_reset(finenv_stock *h, const uint8_t *mask, float *obs_out, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "reset: bind first%s");
    Params p = make_params(h);
    p.mask = mask;
    p.obs = obs_out;
    hipLaunchKernelGGL(stock_aux_kernel, grid_for(h->cfg.n_envs), dim3(kWave * kWavesPerBlock), 0,
                       (hipStream_t)stream, p, 1);
    return check_launch(h, "stock_reset");
}

int finenv_stock_observe(finenv_stock *h, float *obs_out, void *stream)
{
    if (!h || !obs_out) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "observe: bind first%s");
    Params p = make_params(h);
    p.obs = obs_out;
    hipLaunchKernelGGL(stock_aux_kernel, grid_for(h->cfg.n_envs), dim3(kWave * kWavesPerBlock), 0,
                       (hipStream_t)stream, p, 2);
    return check_launch(h, "stock_observe");
}

int finenv_stock_step(finenv_stock *h, const float *actions, float *obs, float *reward,
                      uint8_t *done, float *term_obs, int32_t *realised, int32_t auto_reset,
                      void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "step: bind first%s");
    if (!actions || !obs || !reward || !done)
        return fail(h, FINENV_ERR_INVALID, "step: null actions/obs/reward/done%s");
    Params p = make_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.realised = realised;
    p.auto_reset = auto_reset;
    const dim3 grid = grid_for(h->cfg.n_envs), block(kWave * kWavesPerBlock);
    const hipStream_t s = (hipStream_t)stream;
    const bool turb = h->cfg.use_turbulence != 0, stats = h->cfg.track_stats != 0;
    if (turb && stats) hipLaunchKernelGGL((stock_step_kernel<true, true>), grid, block, 0, s, p);
    else if (turb) hipLaunchKernelGGL((stock_step_kernel<true, false>), grid, block, 0, s, p);
    else if (stats) hipLaunchKernelGGL((stock_step_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((stock_step_kernel<false, false>), grid, block, 0, s, p);
    return check_launch(h, "stock_step");
}

int finenv_stock_episode_stats(finenv_stock *h, double *out, void *stream)
{
    if (!h || !out) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "episode_stats: bind first%s");
    Params p = make_params(h);
    p.stats_out = out;
    const int E = h->cfg.n_envs;
    hipLaunchKernelGGL(stock_stats_kernel, dim3((E + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, p);
    return check_launch(h, "stock_episode_stats");
}

}  // extern "C"
