// finenv_crypto.hip -- MI355X (gfx950) kernel + C ABI for the batched multi-crypto env.
//
// Replaces finrl/meta/env_cryptocurrency_trading/env_multiple_crypto.py step() :59-90,
// reset() :48-57, get_state() :92-98 for E independent envs per launch.
//
// lane = env, one wave per 64 envs, independent waves (no block barriers; one wave per block up to
// 2,048 waves so that a 32,768-env shard -- BASELINE configs[4] per GPU -- spreads over every CU).
// Trades run in asset-index order (no sort in this env), serial through cash in fp64 with the
// reference's operation order; `cash // price` is the exact floor (reciprocal + FMA-remainder
// fix-up); stocks are float32 and fractional as in the reference.  Small rows (51 floats at
// 10 pairs x 4 indicators): 385 algorithmic bytes per env-step.  At these sizes the step is
// LATENCY-bound (12.6 MB per launch at 32,768 envs = 2 us of HBM time): what counts is the number
// of DEPENDENT global round trips, each ~1.5-2 us.  The first version had five (state -> prices for
// the sells -> prices again for the buys -> prices for the asset sum -> indicator row of the
// observation); this one has two: everything that does not depend on `time` (state, holdings,
// action tile) is issued at once, and as soon as `time` is known the price row (kept in registers
// for sells, buys and the asset sum) and the observation's indicator values go out together.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "finenv.h"
#include "finenv_dev.h"
#include "finenv_host.h"

#ifdef FINENV_DIAG
extern unsigned long long *g_finenv_dbg;         // finenv_stock.hip (diagnostic builds)
#endif

namespace {

typedef float wide4 __attribute__((ext_vector_type(4)));
constexpr int kWave = 64;
constexpr int kBlkMaxD = 64;                    // observation rows up to this width: block form (below)
// LDS per wave (dwords): [env][NP + 1] action rows, later the observation heads (odd row stride);
// [64] the indicator values of the wave's common observation row (block form); [64] the wave's time
// counters, published by the env wave for its streamer.
__host__ __device__ constexpr int lds_per_wave(int NP) { return kWave * (NP + 1) + 2 * kWave; }
constexpr int kObsChunks = 4;                   // observation chunks preloaded on the fast path

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
    uint32_t magicH;              // ceil(2^32 / (1 + N)): row of a flat index into the [64][1 + N] heads
    uint32_t magicD;              // ceil(2^32 / D): row of a flat index into the wave's [64][D] block
    // finenv_crypto_step_record: blocks >= env_blocks copy the policy's outputs of this step into
    // the rollout tensors (16-byte elements), beside the env blocks
    int32_t env_blocks;
    int32_t rec_na4, rec_nv4;     // float4 counts: actions [E*N/4], values / log-probs [E/4]
    const float *rec_src[3];      // actions (== actions), values, log-probs
    float *rec_dst[3];
    unsigned long long *dbg;      // FINENV_DIAG builds only: [wave][16] s_memrealtime stamps
};

#ifdef FINENV_DIAG
#define CSTAMP(k)                                                                           \
    do {                                                                                    \
        if (p.dbg != nullptr && lane == 0) {                                                \
            __builtin_amdgcn_sched_barrier(0);                                              \
            p.dbg[(size_t)(e0 / kWave) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
#else
#define CSTAMP(k) do { } while (0)
#endif

#define CF(fld) (*at(p.st.f64, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define CI(fld) (*at(p.st.i32, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define STK(i) (*at(p.st.stocks, (unsigned)(i) * (unsigned)E + (unsigned)e))

// rows[el*stride + 0] = f32(cash * 2^-18), rows[el*stride + 1 + i] = stocks_i * 2^-3   (:93)
// columns >= 1 + N: tech_scaled[(t_el - l) * W + j]                              (:94-97)
__device__ __forceinline__ void cr_write_rows(float *__restrict__ dst, const CrParams &p,
                                              int e0, int nenv_w, int t_row,
                                              unsigned long long lane_mask, const float *rows,
                                              int row_stride, int lane)
{
    const int N = p.cfg.n_assets, W = p.cfg.n_tech, D = p.D;
    const unsigned magicW = p.magicW;
    write_obs_rows_generic<8, 16>(
        dst, p.panel.tech_scaled, D, e0, nenv_w, t_row, lane_mask, rows, row_stride, lane,
        [=](int t, int col) {                                // lookback row l, indicator j
            const int c2 = col - 1 - N;
            const int l = (W == 1) ? c2 : (int)__umulhi((unsigned)c2, magicW);
            return (t - l) * W + (c2 - l * W);
        },
        [=](int col) { return col < 1 + N ? col : -1; });
}

// NP = asset count padded to 8 / 12 / 16 / 32: the per-asset state lives in statically indexed
// registers, so the unrolled loops are compiled per padded width (N = 10 runs the 12-wide build:
// <= 128 VGPRs, four env waves per SIMD -- every wave of a 262,144-env batch is resident at once).
// TWO: one 128-thread block per 64 envs, wave 0 the env step ("trader"), wave 1 a "streamer" that writes the
// indicator columns of the 64 next observation rows -- they depend on the time counter alone (:80, :94-97)
// -- while the trader computes; the trader then writes only the 1 + N head columns.  The two write disjoint
// columns.  The streamer never reads the time counter from memory: the env wave, which owns it, publishes
// the values it loaded through LDS (one barrier, a round trip after the launch, where the streamer would
// wait for its own load anyway) and is the only wave that ever touches time[e] -- nothing to order against
// its write-back at the end of the step, the streamer leaves as soon as its stores are issued.
// (At 32,768 envs the env waves occupy half the chip's SIMDs and the step is one serial chain: assembling
// and writing the whole rows was its last 2 us.)
template <bool RESET_ONLY, int kWaves, int NP, bool TWO = false>
__global__ void __launch_bounds__(kWave *(TWO ? 2 * kWaves : kWaves), 1) crypto_kernel(const CrParams p)
{
    constexpr int kRowW = NP + 1;                            // odd row stride of the LDS rows (dwords)
    extern __shared__ __attribute__((aligned(16))) float lds_all[];   // kWaves * lds_per_wave(NP)
    if (!RESET_ONLY && p.env_blocks > 0 && (int)blockIdx.x >= p.env_blocks) {
        // ---- record blocks (finenv_crypto_step_record): the step's policy outputs -> slice t of
        // the rollout tensors.  At 32,768 envs the env blocks occupy half the SIMDs of the chip and
        // the step is a chain of dependent round trips: the copy rides along for free instead of
        // costing a launch of its own (2-3 us with its launch gap, of a 10 us step).
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int nthreads = ((int)gridDim.x - p.env_blocks) * (int)blockDim.x;
        const int tid = ((int)blockIdx.x - p.env_blocks) * (int)blockDim.x + (int)threadIdx.x;
        const int na4 = p.rec_na4, nv4 = p.rec_nv4;
        for (int i = tid; i < na4 + 2 * nv4; i += nthreads) {
            const int k = i < na4 ? 0 : (i < na4 + nv4 ? 1 : 2);
            const int j = i - (k == 0 ? 0 : (k == 1 ? na4 : na4 + nv4));
            reinterpret_cast<f4 *>(p.rec_dst[k])[j] = reinterpret_cast<const f4 *>(p.rec_src[k])[j];
        }
        return;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wib = TWO ? wv % kWaves : wv;                 // env group of this wave within the block
    const int role = TWO ? wv / kWaves : 0;                 // 0: env step, 1: streamer of the same group
    float *rows = lds_all + wib * lds_per_wave(NP);         // [env][kRowW]: actions, then obs heads
    float *ind = rows + kWave * kRowW;                      // [64]: indicator values of the common row
    int *tpub = reinterpret_cast<int *>(ind + kWave);       // [64]: the wave's time counters (TWO)
    const int E = p.cfg.n_envs, N = p.cfg.n_assets;
    const int e0 = (blockIdx.x * kWaves + wib) * kWave;
    if (e0 >= E) return;                                    // (both roles of the group alike)
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;
    float *row = rows + lane * kRowW;
    // full waves whose indicator columns fit one lane-per-column pass: the streamer takes them
    const int n_ind = p.D - 1 - N;
    const bool split = TWO && !RESET_ONLY && nenv_w == kWave && n_ind > 0 && n_ind <= kWave;
    if (TWO && role == 1) {
        lds_barrier();                                       // the env wave has published its time counters
        if (!split) return;
        const int max_step_s = p.cfg.n_steps - p.cfg.lookback - 1;           // :24
        const int time_s = tpub[lane];                                        // time + 1, :60
        const bool done_s = time_s == max_step_s;                             // :80
        const int trow = (done_s && p.auto_reset) ? p.cfg.lookback - 1 : time_s;
        const int W = p.cfg.n_tech;
        const bool act = lane < n_ind;
        const int c2 = act ? lane : 0;
        const int l = (W == 1) ? c2 : (int)__umulhi((unsigned)c2, p.magicW);   // lookback row, :94-97
        const int j = c2 - l * W;
        float *const sb = p.obs + (size_t)e0 * p.D + 1 + N + c2;
        const int t0 = __builtin_amdgcn_readfirstlane(trow);
        if (__all(trow == t0)) {
            const float v = *at(p.panel.tech_scaled, (unsigned)((t0 - l) * W + j));
#pragma unroll 8
            for (int el = 0; el < kWave; ++el)
                if (act) *at(sb, (unsigned)(el * p.D)) = v;
        } else {
            for (int g = 0; g < kWave; g += 32) {
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int rd = __builtin_amdgcn_readlane(trow, g + u);
                    v[u] = *at(p.panel.tech_scaled, (unsigned)((rd - l) * W + j));
                }
#pragma unroll
                for (int u = 0; u < 32; ++u) pin(v[u]);
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (act) *at(sb, (unsigned)((g + u) * p.D)) = v[u];
            }
        }
        return;
    }

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
        cr_write_rows(p.obs, p, e0, nenv_w, t, __ballot(sel), rows, kRowW, lane);
        return;
    }

    CSTAMP(0);
    // ---- round trip 1: everything that does not depend on `time`; the time counter first (loads
    // return in order: it is the one value the next round trip -- and the streamer -- wait for) ------
    const int time = CI(FINENV_CI_TIME) + 1;                                  // :60
    double cash = CF(FINENV_CF_CASH);
    const double prev_asset = CF(FINENV_CF_TOTAL_ASSET);
    double gamma_ret = CF(FINENV_CF_GAMMA_RETURN);
    float sv[NP];
    double nrm[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        sv[i] = STK(min(i, N - 1));
        nrm[i] = p.panel.norm[min(i, N - 1)];
    }
    // action tile [nenv_w][N]: coalesced read (NP loads cover 64 x N values), transposed through LDS
    // once the second round trip is on its way
    const int a_total = nenv_w * N;
    const float *const a_src = p.actions + (size_t)e0 * N;
    float av[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int f = j * kWave + lane;
        av[j] = *at(a_src, (unsigned)(f < a_total ? f : a_total - 1));
    }
    if (TWO) {
        tpub[lane] = time;
        lds_barrier();                    // the streamer takes the time counters from here
    }
    CSTAMP(1);
    const int max_step = p.cfg.n_steps - p.cfg.lookback - 1;                  // :24
    const bool done = time == max_step;                                       // :80

    // ---- round trip 2: the price row of the new time step (registers: sells, buys, asset sum) and
    // the indicator values of the observation row, issued together ---------------------------
    const unsigned pb = (unsigned)(time * N);
    double prc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) prc[i] = *at(p.panel.price, pb + (unsigned)min(i, N - 1));
    const int D = p.D, W = p.cfg.n_tech;
    const int nchunk = (D + kWave - 1) / kWave;
    const int t_first = __builtin_amdgcn_readfirstlane(time);
    // fast path of the observation write: every env of the wave shows the same row (lock-step and
    // nobody resets in this step) and the row fits kObsChunks chunks
    const bool fast_obs = nchunk <= kObsChunks && __all(time == t_first) &&
                          !(p.auto_reset && __any(done && valid));
    float tt[kObsChunks];
    int tsel[kObsChunks];
#pragma unroll
    for (int k = 0; k < kObsChunks; ++k) {
        const int col = k * kWave + lane;
        const bool in = k < nchunk && col < D;
        tsel[k] = in ? (col < 1 + N ? col : -1) : -2;           // >= 0: per-env head, -1: indicator
        int idx = 0;
        if (in && col >= 1 + N && W > 0) {
            const int c2 = col - 1 - N;
            const int l = (W == 1) ? c2 : (int)__umulhi((unsigned)c2, p.magicW);
            idx = (t_first - l) * W + (c2 - l * W);
        }
        tt[k] = 0.0f;
        if (W > 0 && !split) tt[k] = *at(p.panel.tech_scaled, (unsigned)idx);
    }
    // the action tile has landed (its loads are older than the price loads): transpose it
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int f = j * kWave + lane;
        if (f < a_total) {
            const int el = (N == 1) ? f : (int)__umulhi((unsigned)f, p.magicN);
            rows[el * kRowW + (f - el * N)] = av[j];
        }
    }
    wave_sync();
    CSTAMP(2);

    // Everything below runs on statically indexed registers (holdings sv[], actions act[], prices
    // prc[]): the first version kept holdings and actions in LDS and paid an LDS round trip inside
    // every divergent per-asset block (sells + buys: 4.4 us of a 16 us step at 10 assets).
    // normalised actions (f32 <- f64 product, :63-65)
    float act[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) act[i] = row[min(i, N - 1)];
#pragma unroll
    for (int i = 0; i < NP; ++i) act[i] = (float)((double)act[i] * nrm[i]);
    const double one_m_cs = 1 - p.cfg.sell_cost_pct, one_p_cb = 1 + p.cfg.buy_cost_pct;
#pragma unroll
    for (int i = 0; i < NP; ++i) {                                         // sells :67-71
        if (i >= N) continue;             // (continue, not break: keeps the loop fully unrollable)
        const float a = act[i];
        const double pr = prc[i];
        const bool ok = a < 0.0f && pr > 0.0;
        const float want = -a;
        const float sell = ok ? ((want < sv[i]) ? want : sv[i]) : 0.0f;       // min(stocks, -a)
        const float s_new = sv[i] - sell;
        const double cash_new = cash + pr * (double)sell * one_m_cs;
        sv[i] = ok ? s_new : sv[i];
        cash = ok ? cash_new : cash;      // (a skipped sell must not touch cash: -0.0 / rounding)
    }
    CSTAMP(3);
    // buys :73-77, serial through cash.  The refined reciprocal of every price is computed off the
    // chain; on it, `cash // price` = floor(cash * x) fixed up by the exact sign of FMA remainders.
    double xr[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const double x = __builtin_amdgcn_rcp(prc[i]);
        xr[i] = fma(fma(-prc[i], x, 1.0), x, x);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i >= N) continue;
        const float a = act[i];
        const double pr = prc[i];
        const bool ok = a > 0.0f && pr > 0.0;
        // cash // price.  x = refined reciprocal (relative error far below 2^-40), so floor(cash * x)
        // is within 1 of the true floor whenever the quotient is below 2^40, and the EXACT sign of the
        // FMA remainder says which way: one fix-up makes it the true floor.  Above 2^40 the quotient
        // dwarfs any action and min() returns the action whatever the last units of q are.
        double q = floor(cash * xr[i]);
        const double r = fma(-q, pr, cash);
        q += (r < 0.0) ? -1.0 : ((r >= pr) ? 1.0 : 0.0);
        const double buy = ((double)a < q) ? (double)a : q;                   // min(avail, a)
        const float s_new = (float)((double)sv[i] + buy);
        const double cash_new = cash - pr * buy * one_p_cb;
        sv[i] = ok ? s_new : sv[i];
        cash = ok ? cash_new : cash;
    }
    CSTAMP(4);
    // ---- total asset: cash + np.sum(stocks * price) (NumPy pairwise order), :82 -----------
    // (statically indexed over the NP registers; guards are wave-uniform)
    auto prod = [&](int i) { return (double)sv[i] * prc[i]; };
    double sum = 0.0;
    if (N < 8) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (i < N) sum += prod(i);
    } else {
        double r8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[j] = prod(j);
        const int full = N - (N & 7);
#pragma unroll
        for (int i = 8; i < NP; ++i)
            if (i < full) r8[i & 7] += prod(i);
        sum = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
#pragma unroll
        for (int i = 8; i < NP; ++i)
            if (i >= full && i < N) sum += prod(i);
    }
    const double next = cash + sum;
    double reward = (next - prev_asset) * 0x1p-16;                            // :83
    gamma_ret = gamma_ret * p.cfg.gamma + reward;                             // :85
    if (done) reward = gamma_ret;                                             // :87-88

    CSTAMP(5);
    // ---- observation heads -> LDS; state write-back ------------------------------------------
    // Block form (full wave, lock-step, nobody done, D <= 64, no streamer): the wave's 64 observation
    // rows are ONE contiguous [64][D] block of the output, written with 16-B-per-lane coalesced
    // stores: 64*D/256 store instructions instead of 64, every 64-byte segment written once and
    // whole.  (A wave can have at most 64 vector-memory operations outstanding; behind the state
    // stores the 64 row stores of the row-wise form stalled on write acknowledgements: 3 us of a
    // 13 us step.)  Each lane assembles its 16 bytes from the heads [64][kRowW] and the 64 indicator
    // values of the common row, both in LDS: no [64][D] image is built (13 KB per wave at D = 51 --
    // it capped the residency at 12 waves per CU).
    const bool use_blk = !split && fast_obs && nenv_w == kWave && D <= kBlkMaxD && !__any(done && valid);
    wave_sync();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i >= N) continue;
        row[1 + i] = sv[i] * 0x1p-3f;
        if (valid) STK(i) = (done && p.auto_reset) ? 0.0f : sv[i];
    }
    row[0] = (float)(cash * 0x1p-18);
    if (use_blk) ind[lane] = tt[0];                          // lane = column (D <= 64: one chunk)
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
            cr_write_rows(p.term_obs, p, e0, nenv_w, time, done_mask, rows, kRowW, lane);
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
    CSTAMP(6);
    if (split) {
        // the streamer writes the indicator columns; here: the 64 x (1 + N) head values, read back
        // from LDS in flat order (lane = consecutive columns of a row)
        const int H = 1 + N, total = kWave * H;
        float *const hb = p.obs + (size_t)e0 * D;
#pragma unroll 4
        for (int idx = lane; idx < total; idx += kWave) {
            const int r = (int)__umulhi((unsigned)idx, p.magicH);
            const int cc = idx - r * H;
            *at(hb, (unsigned)(r * D + cc)) = rows[r * kRowW + cc];
        }
    } else if (use_blk) {
        // flat float f of the block = row f / D, column f % D: a head value of that env, or the
        // indicator value of that column
        const int H = 1 + N, n4 = kWave * D / 4;
        wide4 *dst4 = reinterpret_cast<wide4 *>(p.obs + (size_t)e0 * D);
#pragma unroll 2
        for (int j = lane; j < n4; j += kWave) {
            wide4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 4 * j + u;
                const int r = (int)__umulhi((unsigned)f, p.magicD);
                const int cc = f - r * D;
                v[u] = rows[cc < H ? r * kRowW + cc : kWave * kRowW + cc];   // (ind = rows + 64 * kRowW)
            }
            dst4[j] = v;
        }
    } else if (fast_obs) {     // store-only: indicator values preloaded, heads from LDS; row-major order
        // (batches of 16 rows: the batch's LDS reads are in flight before its first store; one
        //  row at a time exposed an LDS round trip per row -- 6.1 us for 64 rows)
        float *const base = p.obs + (size_t)e0 * D;
        constexpr int kRB = 16;
#pragma unroll
        for (int k = 0; k < kObsChunks; ++k) {
            if (k >= nchunk) break;
            const bool head = __any(tsel[k] >= 0);
            for (int g = 0; g < nenv_w; g += kRB) {
                float hv[kRB];
                if (head) {
#pragma unroll
                    for (int j = 0; j < kRB; ++j)
                        hv[j] = rows[min(g + j, kWave - 1) * kRowW + (tsel[k] >= 0 ? tsel[k] : 0)];
                }
#pragma unroll
                for (int j = 0; j < kRB; ++j) {
                    const int el = g + j;
                    if (el >= nenv_w) break;
                    const float v = (head && tsel[k] >= 0) ? hv[j] : tt[k];
                    if (tsel[k] > -2) *at(base, (unsigned)(el * D + k * kWave + lane)) = v;
                }
            }
        }
    } else {
        cr_write_rows(p.obs, p, e0, nenv_w, t_row, valid_mask, rows, kRowW, lane);
    }
    CSTAMP(7);
    if (valid) {
        CF(FINENV_CF_CASH) = cash_out;
        CF(FINENV_CF_TOTAL_ASSET) = asset_out;
        CI(FINENV_CI_TIME) = t_row;       // (this wave alone reads and writes time[e])
    }
    CSTAMP(8);
}

}  // namespace

struct finenv_crypto {
    int device;           // HIP device that owns the bound state block (-1 before bind)
    finenv_crypto_config cfg;
    finenv_crypto_panel panel;
    finenv_crypto_state st;
    int bound;
    int D;
    uint32_t magicN, magicW, magicH, magicD;
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
    p.magicH = h->magicH;
    p.magicD = h->magicD;
    return p;
}
uint32_t magic_for(long long n)
{
    return n >= 2 ? (uint32_t)(((1ull << 32) + n - 1) / (unsigned long long)n) : 0u;
}
constexpr int kSmallWaves = 2048;      // up to here: one env wave per block (spread over every CU)
template <bool RESET_ONLY, int NP>
void cr_launch_np(const CrParams &p, hipStream_t stream)
{
    const int waves = (p.cfg.n_envs + kWave - 1) / kWave;
    CrParams q = p;
    const int n4 = p.rec_na4 + 2 * p.rec_nv4;            // record work (0: plain step)
    const size_t lds1 = sizeof(float) * lds_per_wave(NP);
    // (regime boundary checked in one process, same buffers: with streamers 9.05 / 11.0 / 12.7 us at
    //  65,536 / 98,304 / 131,072 envs, without 10.0 / 12.5 / 13.0 us)
    const int small_waves = kSmallWaves;
    if (waves <= small_waves) {
        const int rec_blocks = n4 > 0 ? min(512, (n4 + kWave * 16 - 1) / (kWave * 16)) : 0;
        q.env_blocks = n4 > 0 ? waves : 0;
        if (RESET_ONLY)
            hipLaunchKernelGGL((crypto_kernel<RESET_ONLY, 1, NP, false>), dim3((unsigned)(waves + rec_blocks)),
                               dim3(kWave), lds1, stream, q);
        else        // trader + streamer wave per 64 envs
            hipLaunchKernelGGL((crypto_kernel<RESET_ONLY, 1, NP, true>), dim3((unsigned)(waves + rec_blocks)),
                               dim3(2 * kWave), lds1, stream, q);
    } else {
        const int blocks = (waves + 3) / 4;
        const int rec_blocks = n4 > 0 ? min(256, (n4 + kWave * 32 - 1) / (kWave * 32)) : 0;
        q.env_blocks = n4 > 0 ? blocks : 0;
        // Large batches are bandwidth-, not latency-bound: no streamer waves (a streamer is a wave of
        // the same kernel and would hold a full wave's registers: half the env waves' residency), the
        // env wave writes whole rows in the block form.
        hipLaunchKernelGGL((crypto_kernel<RESET_ONLY, 4, NP, false>), dim3((unsigned)(blocks + rec_blocks)),
                           dim3(kWave * 4), 4 * lds1, stream, q);
    }
}
template <bool RESET_ONLY>
void cr_launch(const CrParams &p, hipStream_t stream)
{
    if (RESET_ONLY) cr_launch_np<RESET_ONLY, 32>(p, stream);      // (no per-asset registers: one build)
    else if (p.cfg.n_assets <= 8) cr_launch_np<RESET_ONLY, 8>(p, stream);
    else if (p.cfg.n_assets <= 12) cr_launch_np<RESET_ONLY, 12>(p, stream);
    else if (p.cfg.n_assets <= 16) cr_launch_np<RESET_ONLY, 16>(p, stream);
    else cr_launch_np<RESET_ONLY, 32>(p, stream);
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
    h->magicH = magic_for(N + 1);
    h->magicD = magic_for(D);
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
    cr_launch<true>(p, (hipStream_t)stream);
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
#ifdef FINENV_DIAG
    p.dbg = g_finenv_dbg;
#endif
    cr_launch<false>(p, (hipStream_t)stream);
    return cr_check(h, "crypto_step");
}

int finenv_crypto_step_record(finenv_crypto *h, const float *actions, float *obs, float *reward,
                              uint8_t *done, float *term_obs, int32_t auto_reset,
                              const float *values, const float *log_probs, float *actions_out,
                              float *values_out, float *log_probs_out, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return cr_fail(h, FINENV_ERR_UNBOUND, "step_record: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    if (!actions || !obs || !reward || !done || !values || !log_probs || !actions_out ||
        !values_out || !log_probs_out)
        return cr_fail(h, FINENV_ERR_INVALID, "step_record: null pointer");
    const long long na = (long long)h->cfg.n_envs * h->cfg.n_assets, nv = h->cfg.n_envs;
    const uintptr_t bits = (uintptr_t)actions | (uintptr_t)values | (uintptr_t)log_probs |
                           (uintptr_t)actions_out | (uintptr_t)values_out | (uintptr_t)log_probs_out;
    if ((bits & 15) != 0 || (na & 3) != 0 || (nv & 3) != 0)
        return cr_fail(h, FINENV_ERR_INVALID,
                       "step_record: 16-byte aligned buffers and n_envs % 4 == 0 required "
                       "(use finenv_crypto_step + finenv_rollout_put otherwise)");
    CrParams p = cr_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.auto_reset = auto_reset;
    p.rec_na4 = (int32_t)(na / 4);
    p.rec_nv4 = (int32_t)(nv / 4);
    p.rec_src[0] = actions;      p.rec_dst[0] = actions_out;
    p.rec_src[1] = values;       p.rec_dst[1] = values_out;
    p.rec_src[2] = log_probs;    p.rec_dst[2] = log_probs_out;
#ifdef FINENV_DIAG
    p.dbg = g_finenv_dbg;
#endif
    cr_launch<false>(p, (hipStream_t)stream);
    return cr_check(h, "crypto_step_record");
}

}  // extern "C"
