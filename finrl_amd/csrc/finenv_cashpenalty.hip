// finenv_cashpenalty.hip -- MI355X (gfx950) kernel + C ABI for the batched cash-penalty env
// (finrl/meta/env_stock_trading/env_stocktrading_cashpenalty.py: step :291-372,
// get_transactions :249-289, get_reward :237-247, reset :131-157).
//
// lane = env, one 128-thread block per 64 envs, two specialised waves:
//   wave 0 "trader"  : owns the env state.  No ordering between tickers in this env: transactions
//                      are computed per ticker (fp64), two dot products give proceeds / spend, one
//                      test decides cash shortage -- a short fp64 loop per lane.
//   wave 1 "streamer": gathers every env's next panel row (date + 1, or the known restart row on
//                      the last date) and writes the market-data chunks of the observation rows
//                      while the trader computes.  When the trader has decided which episodes end
//                      (cash shortage) it publishes every env's row through an LDS flag -- no
//                      barrier, the trader never waits; the streamer then fetches the chunk that
//                      holds cash / holdings for the decided rows plus the market-data chunks of
//                      the (rare) re-decided rows and parks chunk 0 in LDS.  After the one hand-off
//                      barrier both waves store 32 rows of chunk 0 each.
// One wave used to do both: its 192 row stores alone are 10 us of a 27 us wave (a wave issues one
// store per ~45 ns), with one wave per SIMD on the chip and the other half of the queue idle.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "finenv.h"
#include "finenv_dev.h"
#include "finenv_host.h"

#ifdef FINENV_DIAG
extern unsigned long long *g_finenv_dbg;         // finenv_stock.hip (diagnostic builds)
#endif

namespace {

constexpr int kWave = 64;
constexpr int kMaxN = FINENV_CASHPENALTY_MAX_ASSETS;
constexpr int kRow = kMaxN + 1;
constexpr int kWaves = 2;                    // trader + streamer
constexpr int kClStride = kMaxN + 1;           // f64 close rows [env][33]: odd stride, conflict-free
constexpr int kLdsRows = kWave * kRow;         // f32 [el][33]: action tile, then cash / holdings
constexpr int kLdsClose = kClStride * kWave * 2;   // f64 closes [el][i]
constexpr int kLdsPerBlock = kLdsRows + kLdsClose + kWave + 2;  // + the rows the trade decided (i32), flag
constexpr int kFix = 4;                        // re-decided rows per block the streamer patches in registers
static_assert(kWave * kWave <= kLdsClose, "the parked chunk 0 reuses the close rows");

struct CpParams {
    finenv_cashpenalty_config cfg;
    finenv_cashpenalty_panel panel;
    finenv_cashpenalty_state st;
    const float *actions;
    float *obs;
    float *reward;
    uint8_t *done;
    float *term_obs;
    const uint8_t *mask;
    int32_t auto_reset;
    int32_t D;
    uint32_t magicN;
    int32_t rs_hi;                  // random_start: draw in [0, rs_hi) on the device (0 = off)
    unsigned long long rs_seed;
    double *audit;                  // optional [E][FINENV_AUDIT_HEAD + N] per-step log row, or NULL
    unsigned long long *dbg;        // FINENV_DIAG builds only: [wave][16] s_memrealtime stamps
};

#ifdef FINENV_DIAG
#define KSTAMP(k)                                                                           \
    do {                                                                                    \
        if (p.dbg != nullptr && lane == 0) {                                                \
            __builtin_amdgcn_sched_barrier(0);                                              \
            p.dbg[(size_t)(e0 / kWave) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
#else
#define KSTAMP(k) do { } while (0)
#endif

#define KF(fld) (*at(p.st.f64, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define KI(fld) (*at(p.st.i32, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define KH(i) KF(FINENV_CASHPENALTY_F64_FIELDS + (i))

__device__ __forceinline__ double cp_floordiv(double a, double d)       // exact floor(a/d), d > 0
{
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    double q = floor(a * x);
    double r = fma(-q, d, a);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        q += (r < 0.0) ? -1.0 : ((r >= d) ? 1.0 : 0.0);
        r = fma(-q, d, a);
    }
    return q;
}

__device__ __forceinline__ double cp_reward(const finenv_cashpenalty_config &c, int step,
                                            double total, double cash)     // :237-247
{
    if (step == 0) return 0.0;
    const double pen = fmax(0.0, total * c.cash_penalty_proportion - cash);
    double r = ((total - pen) / c.initial_amount) - 1;
    r /= (double)step;
    return r;
}

// rows[el*kRow + 0] = f32 cash, rows[el*kRow + 1 + i] = f32 holdings_i; columns > N: info row
template <bool kCompact = false>
__device__ __forceinline__ void cp_write_rows(float *__restrict__ dst, const CpParams &p, int e0,
                                              int nenv_w, int row_day,
                                              unsigned long long lane_mask, const float *rows,
                                              int lane)
{
    const int N = p.cfg.n_assets, D = p.D, W = D - 1 - N;
    write_obs_rows_generic<8, 32, kCompact>(
        dst, W > 0 ? p.panel.info : nullptr, D, e0, nenv_w, row_day, lane_mask, rows, kRow, lane,
        [=](int day, int col) { return day * W + col - 1 - N; },
        [=](int col) { return col <= N ? col : -1; });
}

// f64 closes of every env's own date into LDS [el][i] (stride kClStride), 64 row loads in flight
__device__ __forceinline__ void cp_gather_closes(double *trl, const CpParams &p, int di, int lane)
{
    const int N = p.cfg.n_assets;
    const int li = min(lane, N - 1);
    double cv[kWave];
#pragma unroll
    for (int j = 0; j < kWave; ++j) {
        const int de = __builtin_amdgcn_readlane(di, j);
        cv[j] = *at(p.panel.close, (unsigned)(de * N + li));
    }
#pragma unroll
    for (int j = 0; j < kWave; ++j)
        if (lane < N) trl[j * kClStride + lane] = cv[j];
}

// Chunk 0 of rows [el_lo, el_hi): market values parked in LDS ([el][64]) with cash / holdings
// patched in from rows[].  Only stores towards HBM (LDS reads run ahead of them).
template <int NCH>
__device__ __forceinline__ void cp_head_store(float *__restrict__ dst, const CpParams &p, int e0,
                                              int nenv_w, unsigned long long lane_mask,
                                              const float *rows, const float *park, int lane,
                                              int el_lo, int el_hi)
{
    const int N = p.cfg.n_assets, D = p.D;
    float *const base = dst + (size_t)e0 * D;
    const bool head = lane <= N, in = NCH > 1 || lane < D;
    const unsigned long long want = ((el_hi - el_lo >= 64) ? ~0ull : ((1ull << (el_hi - el_lo)) - 1ull))
                                    << el_lo;
    if (nenv_w >= el_hi && (lane_mask & want) == want) {       // all rows: LDS reads 8 rows ahead
        for (int g = el_lo; g < el_hi; g += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float hv = rows[(g + j) * kRow + (head ? lane : 0)];
                const float pv = park[(g + j) * kWave + lane];
                v[j] = head ? hv : pv;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (in) *at(base, (unsigned)((g + j) * D + lane)) = v[j];
        }
        return;
    }
    for (int el = el_lo; el < el_hi; ++el) {
        if (el >= nenv_w || !((lane_mask >> el) & 1ull)) continue;
        const float hv = rows[el * kRow + (head ? lane : 0)];
        const float v = head ? hv : park[el * kWave + lane];
        if (in) *at(base, (unsigned)(el * D + lane)) = v;
    }
}

// (launch bounds: four 2-wave blocks per CU = two waves per SIMD, i.e. at most 256 VGPRs -- at
//  65,536 envs every block of the grid is then resident at once; DISCRETE = cfg.discrete_actions as
//  a template flag: with both transaction formulas in one body the step kernel was 66 KB, over the
//  64 KB instruction cache two CUs share)
template <bool RESET_ONLY, int NCH, bool DISCRETE>
__global__ void __launch_bounds__(kWave *kWaves) __attribute__((amdgpu_waves_per_eu(2)))
cashpenalty_kernel(const CpParams p)
{
    __shared__ __attribute__((aligned(16))) float lds_all[kLdsPerBlock];
    const int lane = threadIdx.x & (kWave - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *rows = lds_all;
    double *trl = reinterpret_cast<double *>(lds_all + kLdsRows);
    int *decided = reinterpret_cast<int *>(lds_all + kLdsRows + kLdsClose);
    const int E = p.cfg.n_envs, N = p.cfg.n_assets;
    const int e0 = blockIdx.x * kWave;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;
    float *row = rows + lane * kRow;
    const finenv_cashpenalty_config &c = p.cfg;

    if (RESET_ONLY) {                                                          // :131-157
        if (role != 0) return;
        const bool sel = valid && (p.mask == nullptr || p.mask[e] != 0);
        const int start = p.rs_hi > 0 ? draw_start(p.rs_seed, e, KI(FINENV_KI_EPISODE) + 1, p.rs_hi)
                                      : KI(FINENV_KI_NEXT_START);
        if (sel) {
            KI(FINENV_KI_START) = start;
            KI(FINENV_KI_DATE_INDEX) = start;
            KI(FINENV_KI_EPISODE) += 1;
            KF(FINENV_KF_TURBULENCE) = 0.0;
            KF(FINENV_KF_SUM_TRADES) = 0.0;
            KF(FINENV_KF_COH) = c.initial_amount;
            for (int i = 0; i < N; ++i) KH(i) = 0.0;
        }
        if (p.obs == nullptr) return;
        row[0] = (float)c.initial_amount;
        for (int i = 0; i < N; ++i) row[1 + i] = 0.0f;
        wave_sync();
        cp_write_rows(p.obs, p, e0, nenv_w, start, __ballot(sel), rows, lane);
        return;
    }

    volatile int *const flag = decided + kWave;
    if (NCH > 0 && role == 0 && lane == 0) *flag = 0;         // (visible after the staging barrier)

    if (role != 0) {
        // ---- streamer ---------------------------------------------------------------------------
        if (NCH == 0) return;                 // rows wider than 320 columns: the trader writes them
        KSTAMP(8);
        const int W = p.D - 1 - N;
        const int di_s = KI(FINENV_KI_DATE_INDEX);
        const bool last = di_s == c.n_days - 1;                                   // :299
        int ns = 0;
        if (p.auto_reset && __any(last))
            ns = p.rs_hi > 0 ? draw_start(p.rs_seed, e, KI(FINENV_KI_EPISODE) + 1, p.rs_hi)
                             : KI(FINENV_KI_NEXT_START);
        // the row the next observation shows unless a cash shortage ends the episode here
        const int row_spec = last ? (p.auto_reset ? ns : di_s) : di_s + 1;
        cp_gather_closes(trl, p, di_s, lane);
        lds_barrier();                        // staging barrier: the trader reads its close rows
        const unsigned long long valid_mask = __ballot(valid);
        float *const base = p.obs + (size_t)e0 * p.D;
        // market-data columns [64, D) of every row (NCH == 2: 64 < D <= 320): ONE 16-byte-per-lane load
        // and store per row at 4-byte aligned addresses, the last quad shifted back to end at D (a
        // wave's 64 vector-memory slots retire in order, ~47 ns each: the dword form -- up to four
        // loads and four stores per row -- covered rows up to 192 columns in 256 slots, this one
        // rows up to 320 columns in 128), 32 rows of loads ahead of their stores
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef f4 f4u __attribute__((aligned(4)));
        const int nq = (p.D - kWave + 3) >> 2;                 // quads per row (<= 64)
        const bool qa = NCH > 1 && lane < nq;
        const int qstart = qa ? min(kWave + 4 * lane, p.D - 4) : kWave;   // (> N: market data only)
        auto quad_src = [&](int de) {
            return reinterpret_cast<const f4u *>(reinterpret_cast<const char *>(p.panel.info) +
                                                 (size_t)((unsigned)(de * W + qstart - 1 - N) * 4u));
        };
        auto quad_dst = [&](int el) {
            return reinterpret_cast<f4u *>(reinterpret_cast<char *>(base) +
                                           (size_t)((unsigned)(el * p.D + qstart) * 4u));
        };
        if (NCH > 1) {
            for (int g = 0; g < kWave; g += 32) {
                f4 t[32];
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    t[j] = *quad_src(__builtin_amdgcn_readlane(row_spec, g + j));
#pragma unroll
                for (int j = 0; j < 32; ++j) pin(t[j]);
                if (g == 0) KSTAMP(9);
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (g + j < nenv_w && qa) *quad_dst(g + j) = t[j];
            }
        }
        KSTAMP(10);
        while (*flag == 0) __builtin_amdgcn_s_sleep(2);       // the trader has decided every env's row
        asm volatile("" ::: "memory");
        const int dec = decided[lane];
        const unsigned long long fix = __ballot(valid && dec != row_spec);
        const int nfix = __builtin_popcountll(fix);
        float t0[kWave];
        f4 tf[kFix];
        if (W > 0) {
#pragma unroll
            for (int el = 0; el < kWave; ++el) {
                const int de = __builtin_amdgcn_readlane(dec, el);
                const bool ld = lane > N && (NCH > 1 || lane < p.D);
                t0[el] = *at(p.panel.info, (unsigned)(ld ? de * W + lane - 1 - N : 0));
            }
            if (NCH > 1 && nfix > 0) {
                unsigned long long m = fix;
#pragma unroll
                for (int j = 0; j < kFix; ++j) {
                    const int el = m != 0ull ? __builtin_ctzll(m) : 0;
                    m &= m - 1ull;
                    tf[j] = *quad_src(__builtin_amdgcn_readlane(dec, el));
                }
            }
        } else {
#pragma unroll
            for (int el = 0; el < kWave; ++el) t0[el] = 0.0f;
        }
        float *const park = reinterpret_cast<float *>(trl);      // (the trader is done with the closes)
#pragma unroll
        for (int el = 0; el < kWave; ++el) park[el * kWave + lane] = t0[el];
        KSTAMP(11);
        lds_barrier();                        // cash / holdings are published; chunk 0 is parked
        cp_head_store<NCH>(p.obs, p, e0, nenv_w, valid_mask, rows, park, lane, kWave / 2, kWave);
        if (NCH > 1 && nfix > 0 && W > 0) {
            unsigned long long m = fix;
#pragma unroll
            for (int j = 0; j < kFix; ++j) {
                if (m == 0ull) continue;
                const int el = __builtin_ctzll(m);
                m &= m - 1ull;
                if (qa) *quad_dst(el) = tf[j];
            }
            if (m != 0ull) cp_write_rows<true>(p.obs, p, e0, nenv_w, dec, m, rows, lane);
        }
        KSTAMP(12);
        return;
    }

    KSTAMP(0);
    // ---- round trip 1: per-env scalars (the action tile is issued behind them) ----------------
    int di = KI(FINENV_KI_DATE_INDEX);
    const int start = KI(FINENV_KI_START);
    double coh = KF(FINENV_KF_COH);
    double turb = c.use_turbulence ? KF(FINENV_KF_TURBULENCE) : 0.0;
    double sum_trades = KF(FINENV_KF_SUM_TRADES);
    double logged_total = KF(FINENV_KF_LOGGED_TOTAL), logged_cash = KF(FINENV_KF_LOGGED_CASH);
    // holdings of every asset, kept in registers (issued behind the scalars: one round trip together
    // with the action tile; the first form fetched them in four dependent batches of 8 and read
    // them a second time for the book update: 12 us of a 28 us step)
    double hb[kMaxN], clb[kMaxN];
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) hb[i] = KH(min(i, N - 1));
    stage_action_tile(rows, kRow, p.actions + (size_t)e0 * N, nenv_w, N, p.magicN, lane);
    const int step = di - start;                                                 // current_step
    const bool at_end = di == c.n_days - 1;                                      // :299
    // closes: every env sits on its own date (random starts), so a per-lane load touches 64
    // different rows per instruction.  Row-wise instead: lane i < N loads close[date_el][i] for one
    // env el per instruction (one 8N-byte segment), all 64 rows in flight, parked in LDS [el][i];
    // each lane then reads its own row back.  The STREAMER does this gather (it needs the dates
    // anyway) while the trader's own loads are in flight: the date -> close-row dependency is off
    // the trader's path, which meets the streamer at the staging barrier.
    if (NCH == 0) {
        cp_gather_closes(trl, p, di, lane);
        wave_sync();
    } else {
        lds_barrier();
    }
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) clb[i] = trl[lane * kClStride + min(i, N - 1)];
    KSTAMP(1);
    float act[kMaxN];
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) act[i] = row[min(i, N - 1)];

    double reward;
    bool done = at_end;
    const float hmaxf = (float)c.hmax;
    const bool turbulent = c.use_turbulence && turb >= c.turbulence_threshold;
    double asset_value = 0.0, proceeds = 0.0, spend = 0.0;
#pragma unroll
    for (int i = 0; i < kMaxN; ++i)
        if (i < N) sum_trades += fabs((double)act[i]);                           // :293
    double tr_[kMaxN];
    if (!at_end) {
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            if (i >= N) continue;              // (continue, not break: keeps the loop fully unrollable)
            const double h = hb[i], cl = clb[i];
            asset_value += h * cl;                                               // np.dot, :310
            const float a32 = act[i] * hmaxf;                                    // :257 (float32)
            const float a = cl > 0.0 ? a32 : 0.0f;                               // :260
            double tr;
            if (DISCRETE) {                                                      // :263-274
                // integer-valued doubles instead of int64 arithmetic (exact below 2^53; a software
                // 64-bit division per asset, unrolled, was 30 KB of code)
                const double q = cp_floordiv((double)a, cl);
                const double inc = (double)c.shares_increment;
                const double num = q >= 0.0 ? q : q + inc;
                tr = cp_floordiv(num, inc) * inc;
            } else {
                tr = (double)a / cl;                                             // :276
            }
            // :279 np.maximum(tr, -h) as compare + select (fmax() quiets both operands first: three
            // extra fp64 instructions per asset at 8 cycles each); a NaN tr (close == 0) gives -h like fmax
            tr = !(tr >= -h) ? -h : tr;
            tr = turbulent ? -h : tr;                                            // :282-287
            tr_[i] = tr;
            proceeds += (tr < 0.0 ? -tr : 0.0) * cl;                             // :323-324
            spend += (tr > 0.0 ? tr : 0.0) * cl;                                 // :328-329
        }
        logged_cash = coh;                                                       // :312-314
        logged_total = coh + asset_value;
    }
    KSTAMP(2);
    reward = cp_reward(c, step, logged_total, logged_cash);                      // :317 / :301
    bool keep_buys = true;
    double coh_new = coh;
    if (!at_end) {
        double costs = proceeds * c.sell_cost_pct;                               // :325
        const double coh1 = coh + proceeds;                                      // :326
        costs += spend * c.buy_cost_pct;                                         // :330
        if (spend + costs > coh1) {                                              // :333
            if (c.patient) {                                                     // :334-339
                keep_buys = false;
                spend = 0.0;
                costs = 0.0;
            } else {
                done = true;                                                     // :341-344
            }
        }
        coh_new = coh1 - spend - costs;                                          // :351
    }
    if (p.audit != nullptr && valid) {      // harness log row (account_information / transaction_memory)
        double *au = p.audit + (size_t)e * (size_t)(FINENV_AUDIT_HEAD + N);
        au[FINENV_AUDIT_BEGIN_CASH] = coh;                                       // :312
        au[FINENV_AUDIT_ASSET_VALUE] = asset_value;                              // :313
        au[FINENV_AUDIT_REWARD] = reward;                                        // :317
        au[FINENV_AUDIT_FLAGS] = (double)((at_end ? FINENV_AUDIT_F_LAST_DATE : 0) |
            ((!at_end && (done || !keep_buys)) ? FINENV_AUDIT_F_CASH_SHORTAGE : 0) |
            ((!at_end && turbulent) ? FINENV_AUDIT_F_TURBULENCE : 0));
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            if (i >= N) continue;
            const double tr = at_end ? 0.0 : tr_[i];
            au[FINENV_AUDIT_HEAD + i] = (tr > 0.0 && !keep_buys) ? 0.0 : tr;     // :336 / :345
        }
    }
    const bool advance = !done;
    // ---- the row of the panel each env's next observation shows: date + 1, the unchanged date on a
    // terminal step, or the new starting point on an auto-reset ----------------------------------
    int ns_reset = 0;
    if (p.auto_reset && __any(done))
        ns_reset = p.rs_hi > 0 ? draw_start(p.rs_seed, e, KI(FINENV_KI_EPISODE) + 1, p.rs_hi)
                               : KI(FINENV_KI_NEXT_START);
    const int row_final = done ? (p.auto_reset ? ns_reset : di) : di + 1;
    if (NCH > 0) {                            // publish (LDS is in-order per wave: rows first, then the flag)
        decided[lane] = row_final;
        asm volatile("" ::: "memory");
        *flag = 1;
    }
    if (advance) {
        coh = coh_new;
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            if (i >= N) continue;
            const double tr = tr_[i];
            const double hn = hb[i] + ((tr > 0.0 && !keep_buys) ? 0.0 : tr);     // :352
            if (valid) KH(i) = hn;
            row[1 + i] = (float)hn;
        }
        di += 1;                                                                 // :353
        if (c.use_turbulence) turb = *at(p.panel.turb, (unsigned)di);            // :354-357
    } else {
#pragma unroll
        for (int i = 0; i < kMaxN; ++i)
            if (i < N) row[1 + i] = (float)hb[i];
    }
    KSTAMP(3);
    row[0] = (float)coh;
    if (valid) {
        *at(p.reward, (unsigned)e) = (float)reward;
        *at(p.done, (unsigned)e) = done ? 1 : 0;
        KF(FINENV_KF_SUM_TRADES) = sum_trades;
        KF(FINENV_KF_LOGGED_TOTAL) = logged_total;
        KF(FINENV_KF_LOGGED_CASH) = logged_cash;
    }
    wave_sync();
    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long done_mask = __ballot(done && valid);
    int row_day = di;
    if (done_mask != 0ull) {
        if (p.term_obs != nullptr)
            cp_write_rows<true>(p.term_obs, p, e0, nenv_w, di, done_mask, rows, lane);   // once per episode
        if (p.auto_reset) {                                                      // reset()
            wave_sync();
            if (done) {
                const int ns = ns_reset;
                di = ns;
                row_day = ns;
                coh = c.initial_amount;
                turb = 0.0;
                row[0] = (float)coh;
                for (int i = 0; i < N; ++i) {
                    if (valid) KH(i) = 0.0;
                    row[1 + i] = 0.0f;
                }
                if (valid) {
                    KI(FINENV_KI_START) = ns;
                    KI(FINENV_KI_EPISODE) += 1;
                    KF(FINENV_KF_SUM_TRADES) = 0.0;
                }
            }
            wave_sync();
        }
    }
    KSTAMP(4);
    if (NCH > 0) {
        lds_barrier();                        // chunk 0 is parked: rows 0..31 here, 32..63 by the streamer
        cp_head_store<NCH>(p.obs, p, e0, nenv_w, valid_mask, rows,
                           reinterpret_cast<const float *>(trl), lane, 0, kWave / 2);
    } else {
        cp_write_rows(p.obs, p, e0, nenv_w, row_day, valid_mask, rows, lane);
    }
    KSTAMP(5);
    if (valid) {
        KF(FINENV_KF_COH) = coh;
        KI(FINENV_KI_DATE_INDEX) = di;
        if (c.use_turbulence) KF(FINENV_KF_TURBULENCE) = turb;
    }
}

}  // namespace

struct finenv_cashpenalty {
    int32_t rs_hi;
    unsigned long long rs_seed;
    double *audit;
    int device;           // HIP device that owns the bound state block (-1 before bind)
    finenv_cashpenalty_config cfg;
    finenv_cashpenalty_panel panel;
    finenv_cashpenalty_state st;
    int bound;
    int D;
    uint32_t magicN;
    char err[256];
};

namespace {
int kp_fail(finenv_cashpenalty *h, int code, const char *msg)
{
    if (h) snprintf(h->err, sizeof(h->err), "%s", msg);
    return code;
}
int kp_check(finenv_cashpenalty *h, const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        return FINENV_ERR_HIP;
    }
    return FINENV_OK;
}
CpParams kp_params(const finenv_cashpenalty *h)
{
    CpParams p;
    memset(&p, 0, sizeof(p));
    p.cfg = h->cfg;
    p.panel = h->panel;
    p.st = h->st;
    p.D = h->D;
    p.magicN = h->magicN;
    p.rs_hi = h->rs_hi;
    p.rs_seed = h->rs_seed;
    p.audit = h->audit;
    return p;
}
dim3 kp_grid(int E) { return dim3((unsigned)((E + kWave - 1) / kWave)); }
}  // namespace

extern "C" {

int finenv_cashpenalty_create(const finenv_cashpenalty_config *cfg, finenv_cashpenalty **out)
{
    if (!cfg || !out) return FINENV_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_envs < 1 || cfg->n_assets < 1 || cfg->n_assets > FINENV_CASHPENALTY_MAX_ASSETS ||
        cfg->n_cols < 0 || cfg->n_days < 1 || cfg->shares_increment < 1 || !(cfg->hmax >= 0) ||
        !(cfg->initial_amount > 0))
        return FINENV_ERR_INVALID;
    const long long E = cfg->n_envs, N = cfg->n_assets, T = cfg->n_days;
    const long long D = 1 + N + N * cfg->n_cols, lim = (1ll << 32) - 1;
    if ((FINENV_CASHPENALTY_F64_FIELDS + N) * E * 8 > lim || T * N * cfg->n_cols * 4 > lim ||
        T * N * 8 > lim || 64 * D * 4 > lim || E * N * 4 > lim)
        return FINENV_ERR_INVALID;
    finenv_cashpenalty *h = new (std::nothrow) finenv_cashpenalty;
    if (!h) return FINENV_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->device = -1;
    h->cfg = *cfg;
    h->D = (int)D;
    h->magicN = N >= 2 ? (uint32_t)(((1ull << 32) + N - 1) / (unsigned long long)N) : 0u;
    *out = h;
    return FINENV_OK;
}

void finenv_cashpenalty_destroy(finenv_cashpenalty *h) { delete h; }
const char *finenv_cashpenalty_last_error(const finenv_cashpenalty *h)
{
    return h ? h->err : "null handle";
}
int finenv_cashpenalty_obs_dim(const finenv_cashpenalty *h) { return h ? h->D : FINENV_ERR_INVALID; }

int finenv_cashpenalty_bind(finenv_cashpenalty *h, const finenv_cashpenalty_panel *panel,
                            const finenv_cashpenalty_state *st)
{
    if (!h || !panel || !st) return FINENV_ERR_INVALID;
    if (!panel->close || (!panel->info && h->cfg.n_cols > 0) ||
        (!panel->turb && h->cfg.use_turbulence) || !st->f64 || !st->i32)
        return kp_fail(h, FINENV_ERR_INVALID, "bind: null pointer");
    h->panel = *panel;
    h->st = *st;
    h->device = finenv_host::pointer_device(st->f64);
    h->bound = 1;
    return FINENV_OK;
}

int finenv_cashpenalty_set_random_start(finenv_cashpenalty *h, int32_t hi, uint64_t seed)
{
    if (!h || hi < 0 || hi > h->cfg.n_days) return FINENV_ERR_INVALID;
    h->rs_hi = hi;
    h->rs_seed = seed;
    return FINENV_OK;
}

int finenv_cashpenalty_set_audit(finenv_cashpenalty *h, double *audit)
{
    if (!h) return FINENV_ERR_INVALID;
    h->audit = audit;
    return FINENV_OK;
}

int finenv_cashpenalty_reset(finenv_cashpenalty *h, const uint8_t *mask, float *obs_out,
                             void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return kp_fail(h, FINENV_ERR_UNBOUND, "reset: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    CpParams p = kp_params(h);
    p.mask = mask;
    p.obs = obs_out;
    hipLaunchKernelGGL((cashpenalty_kernel<true, 0, false>), kp_grid(h->cfg.n_envs), dim3(kWave * kWaves),
                       0, (hipStream_t)stream, p);
    return kp_check(h, "cashpenalty_reset");
}

int finenv_cashpenalty_step(finenv_cashpenalty *h, const float *actions, float *obs,
                            float *reward, uint8_t *done, float *term_obs, int32_t auto_reset,
                            void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return kp_fail(h, FINENV_ERR_UNBOUND, "step: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    if (!actions || !obs || !reward || !done)
        return kp_fail(h, FINENV_ERR_INVALID, "step: null actions/obs/reward/done");
    CpParams p = kp_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.auto_reset = auto_reset;
#ifdef FINENV_DIAG
    p.dbg = g_finenv_dbg;
#endif
    const dim3 grid = kp_grid(h->cfg.n_envs), block(kWave * kWaves);
    const int nch = (h->D + kWave - 1) / kWave;      // chunks per observation row
#define CP_LAUNCH(NCH_)                                                                          \
    do {                                                                                         \
        if (h->cfg.discrete_actions)                                                             \
            hipLaunchKernelGGL((cashpenalty_kernel<false, NCH_, true>), grid, block, 0,          \
                               (hipStream_t)stream, p);                                          \
        else                                                                                     \
            hipLaunchKernelGGL((cashpenalty_kernel<false, NCH_, false>), grid, block, 0,         \
                               (hipStream_t)stream, p);                                          \
    } while (0)
    // NCH_: 1 = rows of one chunk, 2 = rows of up to 320 columns (streamer copies the market data as
    // 16-byte quads), 0 = wider rows (one-wave form)
    if (nch == 1) CP_LAUNCH(1);
    else if (h->D <= kWave + 4 * kWave) CP_LAUNCH(2);
    else CP_LAUNCH(0);
#undef CP_LAUNCH
    return kp_check(h, "cashpenalty_step");
}

}  // extern "C"
