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
#include <stdlib.h>
#include <string.h>

#include <new>
#include <type_traits>

#include "finenv.h"
#include "finenv_dev.h"

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
    int32_t diag;         // FINENV_DIAG builds only: phase-skip bitmask (timing experiments)
    unsigned long long *dbg;   // FINENV_DIAG builds only: [block][role][16] s_memrealtime stamps
};

#ifdef FINENV_DIAG
#define DIAG(bit) (p.diag & (bit))
// phase stamps (100 MHz wall clock) for tools/phase_times.py; diagnostic build only
#define STAMP(k)                                                                          \
    do {                                                                                  \
        if (p.dbg != nullptr && lane == 0) {                                              \
            __builtin_amdgcn_sched_barrier(0);                                            \
            p.dbg[((size_t)blockIdx.x * 2 + role) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
    } while (0)
#else
#define DIAG(bit) 0
#define STAMP(k) do { } while (0)
#endif

// per-env state fields: [field][env] blocks (include/finenv.h)
#define SF(fld) (*at(p.st.f64, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define SI(fld) (*at(p.st.i32, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define HOLD(i) SI(FINENV_STOCK_I32_FIELDS + (i))
#define SH0(i) SI(FINENV_STOCK_I32_FIELDS + N + (i))

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

// Same, with the starting shares in LDS (hcol[i * kWave] = shares of ticker i for this lane)
// and rolled loops: the in-kernel auto-reset path runs once per episode and must not cost
// registers or code size in the step kernel.
__device__ __forceinline__ double initial_asset_lds(double cash0, const int *hcol,
                                                    const double *__restrict__ close,
                                                    unsigned row_base, int N, bool np_sum)
{
    auto prod = [&](int i) { return (double)hcol[i * kWave] * *at(close, row_base + (unsigned)i); };
    double res = 0.0;
    if (!np_sum) {
        for (int i = 0; i < N; ++i) res = res + *at(close, row_base + (unsigned)i) * (double)hcol[i * kWave];
    } else if (N < 8) {
        for (int i = 0; i < N; ++i) res += prod(i);
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = prod(j);
        const int full = N - (N & 7);
        for (int i = 8; i < full; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += prod(i + j);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = full; i < N; ++i) res += prod(i);
    }
    return cash0 + res;
}

// Stream observation rows for the envs selected by `lane_mask` (bit el = env e0+el).
//   rows : per-wave LDS, rows[el*kRow + 0] = f32 cash, rows[el*kRow + 1 + i] = f32 holdings_i
//   row_day (per lane el) = panel row whose prices/indicators go into that env's obs.
// Chunk-outer / env-inner; every store instruction writes 256 contiguous bytes.
// Fast path (all selected envs on the same panel row -- always, in lock-step batches): the
// template chunk is loaded ONCE per chunk, so the env loop holds no load and its stores are
// fire-and-forget (a load inside that loop makes hipcc wait vmcnt(0) per iteration, which
// also drains every outstanding store: measured 45 us -> see DESIGN.md).
__device__ __forceinline__ void write_obs_rows(float *__restrict__ dst,
                                               const float *__restrict__ tmpl, int D, int N,
                                               int e0, int nenv_w, int row_day,
                                               unsigned long long lane_mask,
                                               const float *rows, int lane, int k_lo = 0,
                                               int k_hi = 1 << 30)
{
    if (lane_mask == 0ull) return;
    const int nchunk = min(k_hi, (D + kWave - 1) / kWave);
    const int first = __builtin_ctzll(lane_mask);
    const int rd0 = __builtin_amdgcn_readlane(row_day, first);
    const bool mine = (lane_mask >> lane) & 1ull;
    const bool uniform_row = __all(!mine || row_day == rd0);
    const unsigned long long full_mask = (nenv_w >= 64) ? ~0ull : ((1ull << nenv_w) - 1ull);
    const bool all_rows = lane_mask == full_mask;
    float *const base = dst + (size_t)e0 * D;

    // Streamer fast path: row-major order (all chunks of a row back to back).  Rows are 1204 B,
    // so a 256-B store chunk straddles 64-B memory segments; writing the neighbouring chunk of
    // the same row immediately lets L2 merge the two halves before they leave for HBM
    // (chunk-major order left them ~64 stores apart: WRITE_SIZE 1.19x the bytes stored).
    if (uniform_row && all_rows && k_lo >= (2 * N) / kWave + 1 && nchunk - k_lo <= 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = (k_lo + j) * kWave + lane;
            t[j] = (col < D) ? *at(tmpl, (unsigned)(rd0 * D + col)) : 0.0f;
        }
#pragma unroll 2
        for (int el = 0; el < nenv_w; ++el) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = (k_lo + j) * kWave + lane;
                if (k_lo + j < nchunk && col < D) *at(base, (unsigned)(el * D + col)) = t[j];
            }
        }
        return;
    }

    for (int k = k_lo; k < nchunk; ++k) {
        const int col = k * kWave + lane;
        const bool in = col < D;
        const int hidx = col - 1 - N;
        const bool patch = in && (col == 0 || (hidx >= 0 && hidx < N));
        const int sel = (col == 0) ? 0 : (patch ? 1 + hidx : 0);
        const bool any_patch = __any(patch);
        if (uniform_row) {
            const float t = in ? *at(tmpl, (unsigned)(rd0 * D + col)) : 0.0f;
            if (all_rows && !any_patch) {
#pragma unroll 8
                for (int el = 0; el < nenv_w; ++el)
                    if (in) *at(base, (unsigned)(el * D + col)) = t;
            } else if (all_rows) {
#pragma unroll 8
                for (int el = 0; el < nenv_w; ++el) {
                    const float pv = rows[el * kRow + sel];
                    if (in) *at(base, (unsigned)(el * D + col)) = patch ? pv : t;
                }
            } else {
                for (int el = 0; el < nenv_w; ++el) {
                    if (!((lane_mask >> el) & 1ull)) continue;
                    const float pv = rows[el * kRow + sel];
                    if (in) *at(base, (unsigned)(el * D + col)) = patch ? pv : t;
                }
            }
        } else {
            for (int el = 0; el < nenv_w; ++el) {
                if (!((lane_mask >> el) & 1ull)) continue;
                const int rd = __builtin_amdgcn_readlane(row_day, el);
                const float t = in ? *at(tmpl, (unsigned)(rd * D + col)) : 0.0f;
                const float pv = rows[el * kRow + sel];
                if (in) *at(base, (unsigned)(el * D + col)) = patch ? pv : t;
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// step(): env_stocktrading.py:220-357 (+ DummyVecEnv auto-reset when p.auto_reset)
//
// One 128-thread block per 64 envs, two specialised waves (lane = env in both):
//   wave 0 "trader"  : owns the env state; sort, sells, buys, assets, reward; writes the
//                      observation chunk(s) that contain cash/holdings, and the state.
//   wave 1 "streamer": stages the action tile and the current price row into LDS for the
//                      trader, then streams the market-data part of the observation rows
//                      (chunks that hold no per-env value: 237 of 301 columns at DOW30x8).
// At 65,536 envs there is exactly one trader wave per SIMD on the chip, i.e. no
// thread-level parallelism to hide its latency; the streamer wave shares the SIMD and keeps
// HBM writing while the trader computes (measured: DESIGN.md "stock_step").
// -------------------------------------------------------------------------------------
constexpr int kStepThreads = 2 * kWave;
// Tuning switches (see tools/sweep_stock.py --variants; defaults = measured best)
#ifndef FINENV_NREG_PIN
#define FINENV_NREG_PIN 0         // next-row price loads: 0 hoistable, 1 after sells, 2 after buys
#endif
constexpr int kR1 = kWave * kRow;                 // dwords: act tile / holdings / obs rows
constexpr int kR2 = kNPad * kWave * 2;            // dwords: f64 prices [ticker][lane]
constexpr int kR3 = kWave * kRow;                 // dwords: sorted keys [rank][lane] / obs rows

template <bool TURB, bool STATS>
__global__ void __launch_bounds__(kStepThreads, 2)
stock_step_kernel(const Params p)
{
    __shared__ __attribute__((aligned(16))) float lds_all[kR1 + kR2 + kR3];
    const int lane = threadIdx.x & (kWave - 1);
    const int role = threadIdx.x >> 6;                    // 0 trader, 1 streamer
    float *lds = lds_all;
    int *ldsh = reinterpret_cast<int *>(lds_all);         // [ticker][lane] view of R1
    double *ldsp = reinterpret_cast<double *>(lds_all + kR1);   // [ticker][lane] f64 prices
    int *ldsk = reinterpret_cast<int *>(lds_all + kR1 + kR2);   // [rank][lane] sorted keys
    float *rows = lds_all + kR1 + kR2;                          // later: obs rows [env][kRow]

    const int E = p.cfg.n_envs, N = p.cfg.n_tickers, D = p.D, T = p.cfg.n_days;
    const int e0 = blockIdx.x * kWave;
    if (e0 >= E) return;                                  // block-uniform
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;                 // clamped: tail lanes shadow env e0

    STAMP(0);
#ifdef FINENV_DIAG
    if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 2 + role) * 16 + 14] = __builtin_amdgcn_s_memtime();
#endif
    // ---- both waves: their half of the action tile [nenv_w][N] f32, issued before anything else
    // (16-B coalesced loads; chunk `it` belongs to wave it & 1) -------------------------------
    const float *__restrict__ act_src = p.actions + (size_t)e0 * N;          // 16-B aligned
    const int act_total = nenv_w * N;
    const int act_n4 = act_total >> 2;
    float4 av[kNPad / 8];
#pragma unroll
    for (int j = 0; j < kNPad / 8; ++j) {
        const int idx4 = (2 * j + role) * kWave + lane;
        av[j] = reinterpret_cast<const float4 *>(act_src)[(idx4 < act_n4 && !DIAG(8)) ? idx4 : 0];
    }
    // ---- both roles: which panel rows this step touches (needs only day / price_day) -----
    int day = SI(FINENV_SI_DAY);
    int pd = SI(FINENV_SI_PRICE_DAY);
    const bool term = day >= T - 1;                                           // :221
    const bool do_reset = term && p.auto_reset != 0;
    const int pd_cur = pd;                            // row held in the current observation
    const int pd_next = term ? pd : day + 1;          // row after the step, before any reset
    const int row_obs = do_reset ? (p.cfg.reset_quirk ? pd : 0) : pd_next;
    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long term_mask = __ballot(term && valid);
    const int kpatch = (2 * N) / kWave + 1;           // chunks holding cash/holdings columns
    STAMP(1);

    // values the trader loads before the barrier (declared here: one barrier call site)
    double cash = 0.0, cost = 0.0, turb = 0.0, last_reward = 0.0;
    double st_prev = 0.0, st_mean = 0.0, st_m2 = 0.0;
    int trades = 0, st_n = 0;
    int hreg[kNPad];

    if (role == 1) {
        // ---- streamer, part 1: stage the current price row in LDS ------------------------------
        double pv[kNPad];
#pragma unroll
        for (int i = 0; i < kNPad; ++i)
            pv[i] = *at(p.panel.close, (unsigned)(pd_cur * N + (i < N ? i : 0)));
#pragma unroll
        for (int i = 0; i < kNPad; ++i) ldsp[i * kWave + lane] = pv[i];
    } else {
        // ---- trader, part 1: every global load it will ever need, issued up front ----------
        cash = SF(FINENV_SF_CASH);
        cost = SF(FINENV_SF_COST);
        trades = SI(FINENV_SI_TRADES);
        last_reward = SF(FINENV_SF_LAST_REWARD);
        if (TURB) turb = SF(FINENV_SF_TURBULENCE);
        if (STATS) {
            st_prev = SF(FINENV_SF_PREV_ASSET);
            st_n = SI(FINENV_SI_N_RET);
            st_mean = SF(FINENV_SF_RET_SUM);
            st_m2 = SF(FINENV_SF_RET_SUMSQ);
        }
#pragma unroll
        for (int i = 0; i < kNPad; ++i) hreg[i] = HOLD(i < N ? i : 0);
    }
    // ---- both waves: transpose their half of the action tile into LDS rows (stride 33) ------
#pragma unroll
    for (int j = 0; j < kNPad / 8; ++j) {
        const int idx4 = (2 * j + role) * kWave + lane;
        if (idx4 < act_n4) {
            const float c[4] = {av[j].x, av[j].y, av[j].z, av[j].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 4 * idx4 + u;
                const int el = (N == 1) ? f : (int)__umulhi((unsigned)f, p.magicN);
                lds[el * kRow + (f - el * N)] = c[u];
            }
        }
    }
    if (role == 1)
        for (int f = 4 * act_n4 + lane; f < act_total; f += kWave) {          // < 4 leftover floats
            const int el = (N == 1) ? f : (int)__umulhi((unsigned)f, p.magicN);
            lds[el * kRow + (f - el * N)] = *at(act_src, (unsigned)f);
        }
    STAMP(2);
    __syncthreads();
    STAMP(3);

    if (role == 1) {
        // ---- streamer, part 2: market-data chunks of the observation rows -------------------
        if (!DIAG(1)) {
            if (term_mask != 0ull && p.term_obs != nullptr)
                write_obs_rows(p.term_obs, p.panel.obs_tmpl, D, N, e0, nenv_w, pd_cur,
                               term_mask, lds, lane, kpatch);
            write_obs_rows(p.obs, p.panel.obs_tmpl, D, N, e0, nenv_w, row_obs, valid_mask, lds,
                           lane, kpatch);
        }
        STAMP(4);
        // ---- streamer, part 3: once the trader has published the (cash, holdings) rows in LDS,
        // write the chunk(s) that contain them; the trader goes on to its state write-back.
        // (Episode-end steps keep that write in the trader: it interleaves with the reset.)
        __syncthreads();
        if (!DIAG(1) && term_mask == 0ull)
            write_obs_rows(p.obs, p.panel.obs_tmpl, D, N, e0, nenv_w, row_obs, valid_mask, rows,
                           lane, 0, kpatch);
        return;
    }

    // =========================== trader wave only below ===================================
    // Code-size note: only the key build, the begin-asset sum and the sorting network are
    // unrolled (they need statically indexed VGPRs).  Everything after the sort is a rolled,
    // software-pipelined loop over LDS-resident data: the fully unrolled form was ~55 KB of
    // straight-line code per kernel, i.e. one pass through the whole instruction cache per wave.
    const bool turbulent = TURB && (turb >= p.cfg.turbulence_threshold);      // :308-310
    const int hmax = p.cfg.hmax;
    const float hmaxf = (float)hmax;
    const uint32_t untr = *at(p.panel.untradable, (unsigned)pd_cur);
    double risk_next = 0.0;
    if (TURB) risk_next = *at(p.panel.risk, (unsigned)pd_next);

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

    // ---- holdings -> LDS [ticker][lane]; begin_total_asset (:311-314) -----------------------
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kNPad; ++i) {
        const double t = s + ldsp[i * kWave + lane] * (double)hreg[i];
        s = (i < N) ? t : s;
        ldsh[i * kWave + lane] = hreg[i];          // slots >= N hold a harmless copy of slot 0
    }
    const double begin = cash + s;
    STAMP(4);

    // ---- canonical order, then park the sorted keys in LDS [rank][lane] ------------------------
    if (!DIAG(4)) sort32(keys);
#pragma unroll
    for (int r = 0; r < kNPad; ++r) ldsk[r * kWave + lane] = keys[r];
    STAMP(5);

    // next-row prices for end_total_asset: issued now (keys are parked, registers are free), so the
    // loads fly during the trade loops instead of stalling the end of the step
    double nreg[kNPad];
#pragma unroll
    for (int i = 0; i < kNPad; ++i)
        nreg[i] = *at(p.panel.close, (unsigned)(pd_next * N + (i < N ? i : 0)));

    const double c_s = p.cfg.sell_cost_pct, c_b = p.cfg.buy_cost_pct;
    const double one_m_cs = 1 - c_s, one_p_cb = 1 + c_b;
    const int *kcol = ldsk + lane;                 // this env's sorted keys, stride kWave
    int *hcol = ldsh + lane;                       // this env's holdings by ticker, stride kWave
    const double *pcol = ldsp + lane;              // this env's prices by ticker, stride kWave

    // ---- sells: most negative first (:317-324, _sell_stock :102-169) ---------------------------
    // A ticker is sold OR bought at most once per step, so its holdings are read once and
    // written once (no read-after-write through LDS).  Two-deep software pipeline: the key of
    // rank r+2 and the (holdings, price) of rank r+1 are in flight while rank r is applied.
    // Ranks that sell nothing add +0.0 to cash / cost, which leaves the fp64 sums bit-identical
    // to the reference skipping them.
    if (!DIAG(2)) {
        int key0 = kcol[0];
        int key1 = kcol[1 * kWave];
        int h0 = hcol[(key0 & (kNPad - 1)) * kWave];
        double q0p = pcol[(key0 & (kNPad - 1)) * kWave];
        int n_sold = 0;
#pragma unroll 2
        for (int r = 0; r < kNPad; ++r) {
            if (!__any(key0 < 0)) break;               // sorted: no sells beyond this rank
            const int key2 = kcol[min(r + 2, kNPad - 1) * kWave];
            const int i1 = key1 & (kNPad - 1);
            const int h1 = hcol[i1 * kWave];
            const double p1 = pcol[i1 * kWave];

            const int idx = key0 & (kNPad - 1);
            const int a = key0 >> 5;
            // turbulent: sell everything, tradable flag ignored (:139-163); else :105-133
            const bool ok = key0 < 0 && h0 > 0 &&
                            (turbulent ? (q0p > 0.0) : !((untr >> idx) & 1u));
            const int q = ok ? (turbulent ? h0 : min(-a, h0)) : 0;
            hcol[idx * kWave] = h0 - q;                                       // :123
            n_sold += ok ? 1 : 0;                                             // :129
            const double amt = q0p * (double)q;
            cash = cash + amt * one_m_cs;                                     // :115-121
            cost = cost + amt * c_s;                                          // :124-128
            key0 = key1; key1 = key2; h0 = h1; q0p = p1;
        }
        trades += n_sold;
    }
    STAMP(6);

    // ---- buys: largest first (:319, :328-330, _buy_stock :171-213) ---------------------------
    // Serial through cash: q = min(a, cash // unit), cash -= p*q*(1+c_b).  Everything that does
    // not depend on cash is pipelined off the chain (key two ranks ahead, price + unit price +
    // refined reciprocal one rank ahead).  On the chain, `cash // unit` is floor(cash*(1/unit))
    // clamped to a, made exact by the sign of one FMA remainder (the estimate is within 1 of the
    // true floor for |cash/unit| < 2^40; once it reaches a the test is the exact cash >= a*unit):
    //   q0 = min(floor(cash*x), a);  rem = cash - q0*unit (exact sign)
    //   rem < 0 -> q0-1;  rem >= unit and q0 < a -> q0+1;  else q0      == min(a, cash // unit)
    if (!DIAG(2)) {
        auto refined_rcp = [](double u) {
            const double x = __builtin_amdgcn_rcp(u);
            return fma(fma(-u, x, 1.0), x, x);
        };
        int key0 = kcol[(kNPad - 1) * kWave];
        int key1 = kcol[(kNPad - 2) * kWave];
        double p0 = pcol[(key0 & (kNPad - 1)) * kWave];
        double u0 = p0 * one_p_cb;                                            // :179
        double x0 = refined_rcp(u0);
#pragma unroll 2
        for (int r = kNPad - 1; r >= 0; --r) {
            if (!__any(key0 >= kNPad)) break;          // sorted: no buys (a >= 1) below this rank
            const int key2 = kcol[max(r - 2, 0) * kWave];
            const double p1 = pcol[(key1 & (kNPad - 1)) * kWave];
            const double u1 = p1 * one_p_cb;
            const double x1 = refined_rcp(u1);

            const int idx = key0 & (kNPad - 1);
            const double ad = (double)(key0 >> 5);
            const bool ok = key0 >= kNPad && !turbulent && !((untr >> idx) & 1u) && u0 > 0.0;
            const double q0 = fmin(floor(cash * x0), ad);                     // :178-184
            const double rem = fma(-q0, u0, cash);
            const double adj = ((rem < 0.0) ? -1.0 : 0.0) + ((rem >= u0 && q0 < ad) ? 1.0 : 0.0);
            const double qd = ok ? q0 + adj : 0.0;     // q = 0 leaves cash / cost bit-identical
            const double amt = p0 * qd;
            cash = cash - amt * one_p_cb;                                     // :185-190
            cost = cost + amt * c_b;                                          // :194-196
            trades += ok ? 1 : 0;                                             // :197
            // ds_add (no return): a plain `+=` is an LDS read-modify-write whose wait stalls
            // every iteration of this serial loop for a full LDS round trip
            __hip_atomic_fetch_add(&hcol[idx * kWave], (int)qd, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WAVEFRONT);              // :192
            key0 = key1; key1 = key2; p0 = p1; u0 = u1; x0 = x1;
        }
    }
    STAMP(7);

    // ---- day += 1, new row, end_total_asset (:335-347); obs rows -> LDS (keys are dead) -------
    if (!term) {
        day += 1;
        pd = day;
        if (TURB) turb = risk_next;
    }
    wave_sync();
    s = 0.0;
#pragma unroll
    for (int i = 0; i < kNPad; ++i) {
        const int h = hcol[i * kWave];
        const double t = s + nreg[i] * (double)h;
        s = (i < N) ? t : s;
        if (i < N) rows[lane * kRow + 1 + i] = (float)h;
    }
    rows[lane * kRow] = (float)cash;
    const double end = cash + s;
    STAMP(8);
    if (!term) last_reward = (end - begin) * p.cfg.reward_scaling;            // :350-352
    if (valid) {
        *at(p.reward, (unsigned)e) = (float)last_reward;
        *at(p.done, (unsigned)e) = term ? 1 : 0;
    }
    if (p.realised != nullptr && valid) {    // traded shares == holdings delta (:324, :330)
        for (int i = 0; i < N; ++i)
            *at(p.realised, (unsigned)(e * N + i)) = hcol[i * kWave] - HOLD(i);
    }
    STAMP(9);
    wave_sync();

    // ---- terminal observation; auto-reset (once per episode, wave-uniform) ---------------------
    int episode_inc = 0;
    if (term_mask != 0ull) {
        if (p.term_obs != nullptr)
            write_obs_rows(p.term_obs, p.panel.obs_tmpl, D, N, e0, nenv_w, pd_cur, term_mask,
                           rows, lane, 0, kpatch);
        if (p.auto_reset) {                  // reset(), :359-393
            wave_sync();
            if (term) {
                pd = row_obs;
                cash = SF(FINENV_SF_CASH0);
                for (int i = 0; i < N; ++i) {
                    const int v = SH0(i);
                    hcol[i * kWave] = v;
                    rows[lane * kRow + 1 + i] = (float)v;
                }
                rows[lane * kRow] = (float)cash;
                const double a0 = initial_asset_lds(cash, hcol, p.panel.close, (unsigned)(pd * N),
                                                    N, p.cfg.initial != 0);
                if (valid) {
                    SF(FINENV_SF_ASSET0) = a0;
                    SF(FINENV_SF_PREV_ASSET) = a0;
                    SF(FINENV_SF_RET_SUM) = 0.0;
                    SF(FINENV_SF_RET_SUMSQ) = 0.0;
                    SI(FINENV_SI_N_RET) = 0;
                }
                day = 0;
                turb = 0.0;
                cost = 0.0;
                trades = 0;
                episode_inc = 1;
            }
            wave_sync();
        }
    }

    // ---- the observation chunk(s) holding cash / holdings (:342 / :453-478): written by the
    // streamer after this barrier (rows are final), except on episode-end steps ---------------
    __syncthreads();
    if (!DIAG(1) && term_mask != 0ull)
        write_obs_rows(p.obs, p.panel.obs_tmpl, D, N, e0, nenv_w, row_obs, valid_mask, rows, lane,
                       0, kpatch);
    STAMP(10);

    // ---- running sums of pct_change(asset_memory) for the terminal Sharpe (:243-251) ------------
    // Placed after the observation stores: nothing waits on it but its own write-back.
    if (STATS && !term) {
        const double ret = end / st_prev - 1.0;
        if (valid) {
            SF(FINENV_SF_PREV_ASSET) = end;
            SI(FINENV_SI_N_RET) = st_n + 1;
            SF(FINENV_SF_RET_SUM) = st_mean + ret;
            SF(FINENV_SF_RET_SUMSQ) = st_m2 + ret * ret;
        }
    }
    // ---- state write-back --------------------------------------------------------------------------
    if (valid) {
        SF(FINENV_SF_CASH) = cash;
        SF(FINENV_SF_COST) = cost;
        SI(FINENV_SI_TRADES) = trades;
        SI(FINENV_SI_DAY) = day;
        SI(FINENV_SI_PRICE_DAY) = pd;
        SF(FINENV_SF_LAST_REWARD) = last_reward;
        if (TURB) SF(FINENV_SF_TURBULENCE) = turb;
        if (episode_inc) SI(FINENV_SI_EPISODE) += 1;
#pragma unroll 6
        for (int i = 0; i < N; ++i) HOLD(i) = hcol[i * kWave];
    }
    STAMP(11);
#ifdef FINENV_DIAG
    if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 2 + role) * 16 + 15] = __builtin_amdgcn_s_memtime();
#endif
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
        cash = SF(FINENV_SF_CASH);
        pd = SI(FINENV_SI_PRICE_DAY);
#pragma unroll
        for (int i = 0; i < kNPad; ++i) {
            const int v = HOLD(i < N ? i : 0);
            hf[i] = (i < N) ? v : 0;
        }
    } else {
        if (mode == 1 && p.mask != nullptr) sel = valid && p.mask[e] != 0;
        if (mode == 0) pd = p.day0;
        else pd = p.cfg.reset_quirk ? SI(FINENV_SI_PRICE_DAY) : 0;
        cash = SF(FINENV_SF_CASH0);
#pragma unroll
        for (int i = 0; i < kNPad; ++i) {
            const int v = SH0(i < N ? i : 0);
            hf[i] = (i < N) ? v : 0;
        }
        const double a0 =
            initial_asset(cash, hf, p.panel.close + (size_t)pd * N, N, p.cfg.initial != 0);
        if (sel) {
            SF(FINENV_SF_CASH) = cash;
#pragma unroll
            for (int i = 0; i < kNPad; ++i)
                if (i < N) HOLD(i) = hf[i];
            SF(FINENV_SF_ASSET0) = a0;
            SF(FINENV_SF_PREV_ASSET) = a0;
            SF(FINENV_SF_RET_SUM) = 0.0;
            SF(FINENV_SF_RET_SUMSQ) = 0.0;
            SI(FINENV_SI_N_RET) = 0;
            SI(FINENV_SI_DAY) = (mode == 0) ? p.day0 : 0;
            SI(FINENV_SI_PRICE_DAY) = pd;
            SF(FINENV_SF_TURBULENCE) = 0.0;
            SF(FINENV_SF_COST) = 0.0;
            SI(FINENV_SI_TRADES) = 0;
            if (mode == 0) {
                SI(FINENV_SI_EPISODE) = 0;
                SF(FINENV_SF_LAST_REWARD) = 0.0;
            } else {
                SI(FINENV_SI_EPISODE) += 1;
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
    const double *prow = p.panel.close + (size_t)SI(FINENV_SI_PRICE_DAY) * N;
    double s = 0.0;
    for (int i = 0; i < N; ++i) s = s + prow[i] * (double)HOLD(i);
    const double end = SF(FINENV_SF_CASH) + s;
    const double a0 = SF(FINENV_SF_ASSET0);
    double *out = p.stats_out + (size_t)e * 6;
    out[0] = a0;
    out[1] = end;
    out[2] = end - a0;
    out[3] = SF(FINENV_SF_COST);
    out[4] = (double)SI(FINENV_SI_TRADES);
    double sharpe = __builtin_nan("");
    const int n = SI(FINENV_SI_N_RET);
    if (n >= 2) {        // sqrt(252) * mean / std(ddof=1) from the running sums
        const double s1 = SF(FINENV_SF_RET_SUM), s2 = SF(FINENV_SF_RET_SUMSQ);
        const double mean = s1 / (double)n;
        const double var = (s2 - s1 * mean) / (double)(n - 1);
        if (var > 0.0) sharpe = sqrt(252.0) * mean / sqrt(var);
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

#ifdef FINENV_DIAG
static unsigned long long *g_dbg = nullptr;
extern "C" void finenv_diag_set_stamp_buffer(void *ptr) { g_dbg = (unsigned long long *)ptr; }
#endif

extern "C" {

int finenv_abi_version(void) { return FINENV_ABI_VERSION; }

int finenv_struct_size(int which)
{
    switch (which) {
    case 0: return (int)sizeof(finenv_stock_config);
    case 1: return (int)sizeof(finenv_stock_panel);
    case 2: return (int)sizeof(finenv_stock_state);
    case 3: return (int)sizeof(finenv_portfolio_config);
    case 4: return (int)sizeof(finenv_portfolio_panel);
    case 5: return (int)sizeof(finenv_portfolio_state);
    case 6: return (int)sizeof(finenv_crypto_config);
    case 7: return (int)sizeof(finenv_crypto_panel);
    case 8: return (int)sizeof(finenv_crypto_state);
    case 9: return (int)sizeof(finenv_stocknp_config);
    case 10: return (int)sizeof(finenv_stocknp_panel);
    case 11: return (int)sizeof(finenv_stocknp_state);
    default: return FINENV_ERR_INVALID;
    }
}

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
    {   // every device offset is a 32-bit byte offset from a uniform base (see at())
        const long long E = cfg->n_envs, N = cfg->n_tickers, T = cfg->n_days;
        const long long D = 1 + 2 * N + (long long)cfg->n_tech * N;
        const long long lim = (1ll << 32) - 1;
        if ((FINENV_STOCK_I32_FIELDS + 2 * N) * E * 4 > lim || FINENV_STOCK_F64_FIELDS * E * 8 > lim ||
            T * D * 4 > lim || T * N * 8 > lim || E * N * 4 > lim)
            return FINENV_ERR_INVALID;
    }
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
    if (!st->f64 || !st->i32)
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
#ifdef FINENV_DIAG
    {
        const char *d = getenv("FINENV_DIAG");
        p.diag = d ? atoi(d) : 0;
        p.dbg = g_dbg;
    }
#endif
    const dim3 grid((unsigned)((h->cfg.n_envs + kWave - 1) / kWave)), block(kStepThreads);
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
