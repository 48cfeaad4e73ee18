// finenv_stoploss.hip -- MI355X (gfx950) kernel + C ABI for the batched stop-loss env
// (finrl/meta/env_stock_trading/env_stocktrading_stoploss.py: step :292-442,
// get_reward :255-290, reset :134-165).
//
// lane = env.  Per-asset books (holdings, previous holdings, the two price-vs-average-buy
// differences, buy counts, average buy price) live in HBM as [N][E] f64 rows, so every per-asset
// access of a wave is one coalesced 512-B row segment.  Three short fp64 passes over the assets:
// (1) previous-state reward terms, (2) transactions + stop-loss override + cash test inputs,
// (3) book update.  Sums run in asset order, like the oracle.
//
// Observation rows of up to 320 columns (stoploss_step2_kernel): one 128-thread block per 64 envs,
// two specialised waves --
//   wave 0 "trader"  : passes 1 and 2 over every asset, the cash decision, pass 3 for assets
//                      0..15, scalars, terminal observation / auto-reset.
//   wave 1 "streamer": gathers every env's close row (each env sits on its own date), streams the
//                      market-data chunks of the next observation rows while the trader computes,
//                      loads the books of assets 16.. meanwhile; once the trader has published
//                      its decision (LDS flag, no barrier) it runs pass 3 for those assets from
//                      the transactions the trader parked in LDS, and fetches the observation
//                      chunk that holds cash / holdings.  Two barriers at the end: pass 3 complete,
//                      rows final; both waves then store 32 rows of chunk 0.
// A wave keeps at most 64 vector-memory operations in flight and retires them in order (~47 ns per
// slot under load): the one-wave form (stoploss_kernel, still used for wider rows and for reset)
// pushes ~800 of them per step through one queue.

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
constexpr int kMaxN = FINENV_STOPLOSS_MAX_ASSETS;
constexpr int kRow = kMaxN + 1;
constexpr int kWaves = 2;
constexpr int kB = 8;                        // assets per load batch
constexpr int kLdsPerWave = kWave * kRow + kMaxN * kWave * 2;   // rows + f64 transactions [i][lane]
// two-wave step kernel
constexpr int kHalf = 16;                      // assets [0, kHalf): trader's pass 3; the rest: streamer's
constexpr int kClStride = kMaxN + 1;           // f64 close rows [env][33]: odd stride, conflict-free
constexpr int kL2Rows = kWave * kRow;                          // f32 [el][33]
constexpr int kL2Close = kClStride * kWave * 2;                // f64 closes [el][i]; later: parked chunk 0
constexpr int kL2Trx = (kMaxN - kHalf) * kWave * 2;            // f64 transactions of assets >= kHalf [i][lane]
constexpr int kL2Misc = 6 * kWave + 2;                         // decided, eflags, aflags, ntr (f64), flag
constexpr int kLds2 = kL2Rows + kL2Close + kL2Trx + kL2Misc;
static_assert(kLds2 * 4 * 4 <= 160 * 1024, "four blocks per CU");
constexpr int kFix = 4;                        // re-decided rows per block the streamer patches in registers
static_assert(kWave * kWave <= kL2Close, "the parked chunk 0 reuses the close rows");

struct SlParams {
    finenv_stoploss_config cfg;
    finenv_stoploss_panel panel;
    finenv_stoploss_state st;
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
    unsigned long long *dbg;        // FINENV_DIAG builds only: [block][16] s_memrealtime stamps
};

#ifdef FINENV_DIAG
#define SSTAMP(k)                                                                           \
    do {                                                                                    \
        if (p.dbg != nullptr && lane == 0) {                                                \
            __builtin_amdgcn_sched_barrier(0);                                              \
            p.dbg[(size_t)(e0 / kWave) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
#else
#define SSTAMP(k) do { } while (0)
#endif

#define LF(fld) (*at(p.st.f64, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define LI(fld) (*at(p.st.i32, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define LV(book, i) LF(FINENV_STOPLOSS_F64_FIELDS + (book) * N + (i))

__device__ __forceinline__ double sl_floordiv(double a, double d)       // exact floor(a/d), d > 0
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

// get_reward(), :255-290, from its four sums
__device__ __forceinline__ double sl_reward(const finenv_stoploss_config &c, int step,
                                            double total, double cash, double slp_sum,
                                            double lpp_sum, double add)
{
    if (step == 0) return 0.0;
    const double cash_penalty = fmax(0.0, total * c.cash_penalty_proportion - cash);   // :270
    const double slp = step > 1 ? -slp_sum : 0.0;                                       // :271-277
    const double lpp = -lpp_sum;                                                        // :278-280
    const double total_penalty = cash_penalty + slp + lpp;                              // :281
    double r = ((total - total_penalty + add) / c.initial_amount) - 1;                  // :285-287
    r /= (double)step;                                                                  // :288
    return r;
}

// rows[el*kRow + 0] = f32 cash, rows[el*kRow + 1 + i] = f32 holdings_i; columns > N: info row
template <bool kCompact = false>
__device__ __forceinline__ void sl_write_rows(float *__restrict__ dst, const SlParams &p, int e0,
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

// The one-wave kernel: reset, and steps whose observation rows are wider than 320 columns.
template <bool RESET_ONLY>
__global__ void __launch_bounds__(kWave *kWaves, 1) stoploss_kernel(const SlParams p)
{
    __shared__ __attribute__((aligned(16))) float lds_all[kWaves * kLdsPerWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x >> 6;
    float *rows = lds_all + wib * kLdsPerWave;
    const int E = p.cfg.n_envs, N = p.cfg.n_assets;
    const int e0 = (blockIdx.x * kWaves + wib) * kWave;
    if (e0 >= E) return;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;
    float *row = rows + lane * kRow;
    const finenv_stoploss_config &c = p.cfg;

    if (RESET_ONLY) {                                                          // :134-165
        const bool sel = valid && (p.mask == nullptr || p.mask[e] != 0);
        const int start = p.rs_hi > 0 ? draw_start(p.rs_seed, e, LI(FINENV_LI_EPISODE) + 1, p.rs_hi)
                                      : LI(FINENV_LI_NEXT_START);
        if (sel) {
            LI(FINENV_LI_START) = start;
            LI(FINENV_LI_DATE_INDEX) = start;
            LI(FINENV_LI_EPISODE) += 1;
            LF(FINENV_LF_TURBULENCE) = 0.0;
            LF(FINENV_LF_SUM_TRADES) = 0.0;
            LF(FINENV_LF_ACTUAL_NUM_TRADES) = 0.0;
            LF(FINENV_LF_COH) = c.initial_amount;
            for (int i = 0; i < FINENV_STOPLOSS_BOOKS * N; ++i) LV(0, i) = 0.0;
        }
        if (p.obs == nullptr) return;
        row[0] = (float)c.initial_amount;
        for (int i = 0; i < N; ++i) row[1 + i] = 0.0f;
        wave_sync();
        sl_write_rows(p.obs, p, e0, nenv_w, start, __ballot(sel), rows, lane);
        return;
    }

    // ---- action tile -> LDS rows --------------------------------------------------------------
    stage_action_tile(rows, kRow, p.actions + (size_t)e0 * N, nenv_w, N, p.magicN, lane);
    int di = LI(FINENV_LI_DATE_INDEX);
    const int start = LI(FINENV_LI_START);
    double coh = LF(FINENV_LF_COH);
    double turb = c.use_turbulence ? LF(FINENV_LF_TURBULENCE) : 0.0;
    double sum_trades = LF(FINENV_LF_SUM_TRADES);
    double logged_total = LF(FINENV_LF_LOGGED_TOTAL), logged_cash = LF(FINENV_LF_LOGGED_CASH);
    double actual_num_trades = LF(FINENV_LF_ACTUAL_NUM_TRADES);
    const int step = di - start;                                                 // current_step
    const bool at_end = di == c.n_days - 1;                                      // :302
    const unsigned cb = (unsigned)(di * N);
    wave_sync();

    // ---- pass 1: reward terms of the state as the previous step left it (:313 / :304).  On every
    // step but the last date these sums ride along in pass 2 (same assets, same order), which saves
    // re-reading the holdings and previous-holdings books: the three-pass form moved 1.34x the
    // algorithmic bytes (PMC, profiles/side_traffic.json).
    double slp_sum = 0.0, lpp_sum = 0.0, add = 0.0;
    const double lt_old = logged_total, lc_old = logged_cash;
    // (every per-asset loop below runs in batches of kB assets with the batch's global loads issued
    //  first: a rolled loop exposes one HBM round trip per asset at one wave per SIMD)
    if (at_end) {
        for (int i0 = 0; i0 < N; i0 += kB) {
            double hh[kB], ps[kB], ph[kB], cd[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = min(i0 + j, N - 1);
                hh[j] = LV(FINENV_LV_HOLDINGS, i);
                ps[j] = LV(FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, i);
                ph[j] = LV(FINENV_LV_PREV_HOLDINGS, i);
                cd[j] = LV(FINENV_LV_CLOSING_DIFF_AVG_BUY, i);
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) { pin(hh[j]); pin(ps[j]); pin(ph[j]); pin(cd[j]); }
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                if (i0 + j >= N) continue;
                sum_trades += fabs((double)row[i0 + j]);                         // :294
                slp_sum += ph[j] * fmin(cd[j], 0.0);                             // :262,:273-275
                lpp_sum += hh[j] * fmin(ps[j], 0.0);                             // :263-265,:278-280
                add += hh[j] * fmax(ps[j], 0.0);                                 // :266-268,:283
            }
        }
    }
    double reward = at_end ? sl_reward(c, step, lt_old, lc_old, slp_sum, lpp_sum, add) : 0.0;  // :304
    bool done = at_end;

    // ---- pass 2: transactions (:320-357), proceeds / spend (:363-370) --------------------------
    const float hmaxf = (float)c.hmax;
    const bool turbulent = c.use_turbulence && turb >= c.turbulence_threshold;
    const bool stop_armed = coh >= c.stoploss_penalty * c.initial_amount;        // :353
    double asset_value = 0.0, proceeds = 0.0, spend = 0.0, slp_new = 0.0;
    bool keep_buys = true;
    double coh_new = coh;
    const double coh_begin = coh;
    int audit_flags = at_end ? FINENV_AUDIT_F_LAST_DATE : 0;
    // holdings, closes, average buy prices and transactions stay in statically indexed registers
    // from pass 2 to pass 3 (the batch loops are unrolled over the kMaxN slots): pass 3 used to
    // re-read the first three (1.22x the algorithmic traffic, profiles/side_traffic.json)
    double hS[kMaxN], clS[kMaxN], abS[kMaxN], trS[kMaxN];
    if (!at_end) {
#pragma unroll
        for (int i0 = 0; i0 < kMaxN; i0 += kB) {
            if (i0 >= N) continue;
            double hb[kB], clb[kB], ab[kB], pb[kB], pso[kB], cdo[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = min(i0 + j, N - 1);
                hb[j] = LV(FINENV_LV_HOLDINGS, i);
                clb[j] = *at(p.panel.close, cb + (unsigned)i);
                ab[j] = LV(FINENV_LV_AVG_BUY_PRICE, i);
                pb[j] = LV(FINENV_LV_PREV_HOLDINGS, i);
                pso[j] = LV(FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, i);      // pass-1 terms (old books)
                cdo[j] = LV(FINENV_LV_CLOSING_DIFF_AVG_BUY, i);
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                pin(hb[j]); pin(clb[j]); pin(ab[j]); pin(pb[j]); pin(pso[j]); pin(cdo[j]);
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = i0 + j;
                if (i >= N) continue;          // (continue, not break: keeps the loop fully unrollable)
                const double h = hb[j], cl = clb[j], abp = ab[j];
                sum_trades += fabs((double)row[i]);                              // :294
                slp_sum += pb[j] * fmin(cdo[j], 0.0);                            // :262,:273-275
                lpp_sum += h * fmin(pso[j], 0.0);                                // :263-265,:278-280
                add += h * fmax(pso[j], 0.0);                                    // :266-268,:283
                asset_value += h * cl;                                           // :311
                const float a32 = row[i] * hmaxf;                                // :321 (float32)
                double a = cl > 0.0 ? (double)a32 : 0.0;                         // :326
                a = turbulent ? -(h * cl) : a;                                   // :327-331
                double tr;
                if (c.discrete_actions) {                                        // :333-343
                    // integer-valued doubles instead of int64 arithmetic (exact below 2^53; a
                    // software 64-bit division per asset, unrolled, doubled the kernel's code)
                    const double q = cl > 0.0 ? sl_floordiv(a, cl) : 0.0;
                    const double inc = (double)c.shares_increment;
                    const double num = q >= 0.0 ? q : q + inc;
                    tr = sl_floordiv(num, inc) * inc;
                } else {
                    tr = cl > 0.0 ? a / cl : 0.0;                                // :345
                }
                tr = fmax(tr, -h);                                               // :348
                const double cd = cl - (c.stoploss_penalty * abp);               // :350-352
                if (valid) LV(FINENV_LV_CLOSING_DIFF_AVG_BUY, i) = cd;
                slp_new += pb[j] * fmin(cd, 0.0);
                tr = (stop_armed && cd < 0.0) ? -h : tr;                         // :353-357
                audit_flags |= (stop_armed && cd < 0.0) ? FINENV_AUDIT_F_STOP_LOSS : 0;   // :359-360
                hS[i] = h;
                clS[i] = cl;
                abS[i] = abp;
                trS[i] = tr;
                proceeds += (tr < 0.0 ? -tr : 0.0) * cl;                         // :363-364
                spend += (tr > 0.0 ? tr : 0.0) * cl;                             // :368-369
            }
        }
        reward = sl_reward(c, step, lt_old, lc_old, slp_sum, lpp_sum, add);      // :313 (stale log)
        logged_cash = coh;                                                       // :315-317
        logged_total = coh + asset_value;
        double costs = proceeds * c.sell_cost_pct;                               // :365
        const double coh1 = coh + proceeds;                                      // :366
        costs += spend * c.buy_cost_pct;                                         // :370
        audit_flags |= turbulent ? FINENV_AUDIT_F_TURBULENCE : 0;
        if (spend + costs > coh1) {                                              // :372
            audit_flags |= FINENV_AUDIT_F_CASH_SHORTAGE;
            if (c.patient) {                                                     // :373-378
                keep_buys = false;
                spend = 0.0;
                costs = 0.0;
            } else {                                                             // :379-383
                done = true;
                reward = sl_reward(c, step, logged_total, logged_cash, slp_new, lpp_sum, add);
            }
        }
        coh_new = coh1 - spend - costs;                                          // :414
    }

    // ---- pass 3: book update (:388-428) --------------------------------------------------------
    const bool advance = !done;
    if (advance) {
        coh = coh_new;
        double ntr = 0.0;
        // buy counts: the one book pass 2 did not read -- all of them in one batch, before this
        // pass's first store
        double nbS[kMaxN];
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) nbS[i] = LV(FINENV_LV_N_BUYS, min(i, N - 1));
#pragma unroll
        for (int i0 = 0; i0 < kMaxN; i0 += kB) {
            if (i0 >= N) continue;
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = i0 + j;
                if (i >= N) continue;          // (continue, not break: keeps the loop fully unrollable)
                const double tr0 = trS[i];
                const double cl = clS[i], h = hS[i];
                double abp = abS[i], nb = nbS[i];
                const bool sold = tr0 < 0.0;                                     // sells > 0
                const bool bought = tr0 > 0.0;                                   // buys > 0 (:418)
                const double tr = (bought && !keep_buys) ? 0.0 : tr0;            // :376
                const double scp = sold ? cl : 0.0;                              // :388-390
                const bool profit = scp - abp > 0.0;                             // :391-393
                const double ps = profit ? cl - (c.min_profit_penalty * abp) : 0.0;  // :395-399
                audit_flags |= ps < 0.0 ? FINENV_AUDIT_F_LOW_PROFIT                   // :401-405
                                        : (ps > 0.0 ? FINENV_AUDIT_F_HIGH_PROFIT : 0);
                ntr += tr != 0.0 ? 1.0 : 0.0;                                    // :411
                const double hu = h + tr;                                        // :415
                nb += bought ? 1.0 : 0.0;                                        // :419
                const double abp_new = abp + ((cl - abp) / nb);                  // :420-424
                abp = bought ? abp_new : abp;
                const bool held = hu > 0.0;                                      // :427-428
                nb = held ? nb : 0.0;
                abp = held ? abp : 0.0;
                if (valid) {
                    LV(FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, i) = ps;
                    LV(FINENV_LV_PREV_HOLDINGS, i) = h;
                    LV(FINENV_LV_HOLDINGS, i) = hu;
                    LV(FINENV_LV_N_BUYS, i) = nb;
                    LV(FINENV_LV_AVG_BUY_PRICE, i) = abp;
                }
                row[1 + i] = (float)hu;
            }
        }
        actual_num_trades = ntr;
        di += 1;                                                                 // :430
        if (c.use_turbulence) turb = *at(p.panel.turb, (unsigned)di);            // :431-434
    } else {
        for (int i = 0; i < N; ++i) row[1 + i] = (float)LV(FINENV_LV_HOLDINGS, i);
    }
    if (p.audit != nullptr && valid) {      // harness log row (account_information / transaction_memory)
        double *au = p.audit + (size_t)e * (size_t)(FINENV_AUDIT_HEAD + N);
        au[FINENV_AUDIT_BEGIN_CASH] = coh_begin;                                 // :307
        au[FINENV_AUDIT_ASSET_VALUE] = asset_value;                              // :311
        au[FINENV_AUDIT_REWARD] = reward;
        au[FINENV_AUDIT_FLAGS] = (double)audit_flags;
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            if (i >= N) continue;
            const double tr = at_end ? 0.0 : trS[i];
            au[FINENV_AUDIT_HEAD + i] = (tr > 0.0 && !keep_buys) ? 0.0 : tr;     // :376 / :385
        }
    }
    row[0] = (float)coh;
    if (valid) {
        *at(p.reward, (unsigned)e) = (float)reward;
        *at(p.done, (unsigned)e) = done ? 1 : 0;
        LF(FINENV_LF_SUM_TRADES) = sum_trades;
        LF(FINENV_LF_LOGGED_TOTAL) = logged_total;
        LF(FINENV_LF_LOGGED_CASH) = logged_cash;
        LF(FINENV_LF_ACTUAL_NUM_TRADES) = actual_num_trades;
    }
    wave_sync();
    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long done_mask = __ballot(done && valid);
    int row_day = di;
    if (done_mask != 0ull) {
        if (p.term_obs != nullptr)
            sl_write_rows<true>(p.term_obs, p, e0, nenv_w, di, done_mask, rows, lane);   // once per episode
        if (p.auto_reset) {                                                      // reset()
            wave_sync();
            if (done) {
                const int ns = p.rs_hi > 0 ? draw_start(p.rs_seed, e, LI(FINENV_LI_EPISODE) + 1, p.rs_hi)
                                           : LI(FINENV_LI_NEXT_START);
                di = ns;
                row_day = ns;
                coh = c.initial_amount;
                turb = 0.0;
                row[0] = (float)coh;
                for (int i = 0; i < N; ++i) row[1 + i] = 0.0f;
                if (valid) {
                    for (int i = 0; i < FINENV_STOPLOSS_BOOKS * N; ++i) LV(0, i) = 0.0;
                    LI(FINENV_LI_START) = ns;
                    LI(FINENV_LI_EPISODE) += 1;
                    LF(FINENV_LF_SUM_TRADES) = 0.0;
                    LF(FINENV_LF_ACTUAL_NUM_TRADES) = 0.0;
                }
            }
            wave_sync();
        }
    }
    sl_write_rows(p.obs, p, e0, nenv_w, row_day, valid_mask, rows, lane);
    if (valid) {
        LF(FINENV_LF_COH) = coh;
        LI(FINENV_LI_DATE_INDEX) = di;
        if (c.use_turbulence) LF(FINENV_LF_TURBULENCE) = turb;
    }
}


// f64 closes of every env's own date into LDS [el][i] (stride kClStride), 64 row loads in flight
__device__ __forceinline__ void sl_gather_closes(double *trl, const SlParams &p, int di, int lane)
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
__device__ __forceinline__ void sl_head_store(float *__restrict__ dst, const SlParams &p, int e0,
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

// pass 3 (:388-428) for one asset: new books from (transaction, close, holdings, average buy price,
// buy count).  Returns the new holdings; *ntr counts applied transactions, *flags the audit bits.
struct SlBook { double ps, hu, nb, abp; };
__device__ __forceinline__ SlBook sl_update(const finenv_stoploss_config &c, double tr0, double cl,
                                            double h, double abp, double nb, bool keep_buys,
                                            double &ntr, int &flags)
{
    const bool sold = tr0 < 0.0;                                     // sells > 0
    const bool bought = tr0 > 0.0;                                   // buys > 0 (:418)
    const double tr = (bought && !keep_buys) ? 0.0 : tr0;            // :376
    const double scp = sold ? cl : 0.0;                              // :388-390
    const bool profit = scp - abp > 0.0;                             // :391-393
    SlBook b;
    b.ps = profit ? cl - (c.min_profit_penalty * abp) : 0.0;         // :395-399
    flags |= b.ps < 0.0 ? FINENV_AUDIT_F_LOW_PROFIT                  // :401-405
                        : (b.ps > 0.0 ? FINENV_AUDIT_F_HIGH_PROFIT : 0);
    ntr += tr != 0.0 ? 1.0 : 0.0;                                    // :411
    b.hu = h + tr;                                                   // :415
    nb += bought ? 1.0 : 0.0;                                        // :419
    const double abp_new = abp + ((cl - abp) / nb);                  // :420-424
    abp = bought ? abp_new : abp;
    const bool held = b.hu > 0.0;                                    // :427-428
    b.nb = held ? nb : 0.0;
    b.abp = held ? abp : 0.0;
    return b;
}

template <int NCH, bool DISCRETE>
__global__ void __launch_bounds__(kWave *kWaves) __attribute__((amdgpu_waves_per_eu(2)))
stoploss_step2_kernel(const SlParams p)
{
    __shared__ __attribute__((aligned(16))) float lds_all[kLds2];
    const int lane = threadIdx.x & (kWave - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *rows = lds_all;
    double *trl = reinterpret_cast<double *>(lds_all + kL2Rows);
    double *trx = reinterpret_cast<double *>(lds_all + kL2Rows + kL2Close);
    int *decided = reinterpret_cast<int *>(lds_all + kL2Rows + kL2Close + kL2Trx);
    int *eflags = decided + kWave;             // bit 0 advance, bit 1 keep_buys, bit 2 last date
    int *aflags = eflags + kWave;              // audit bits of the streamer's pass 3
    double *ntrx = reinterpret_cast<double *>(aflags + kWave);   // its count of applied transactions
    volatile int *const flag = reinterpret_cast<int *>(ntrx + kWave);
    const int E = p.cfg.n_envs, N = p.cfg.n_assets;
    const int e0 = blockIdx.x * kWave;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;
    float *row = rows + lane * kRow;
    const finenv_stoploss_config &c = p.cfg;
    const int W = p.D - 1 - N;
    if (role == 0 && lane == 0) *flag = 0;     // (visible after the staging barrier)

    if (role != 0) {
        // ================================ streamer ===============================================
        SSTAMP(8);
        const int di_s = LI(FINENV_LI_DATE_INDEX);
        const bool last = di_s == c.n_days - 1;                                   // :302
        int ns = 0;
        if (p.auto_reset && __any(last))
            ns = p.rs_hi > 0 ? draw_start(p.rs_seed, e, LI(FINENV_LI_EPISODE) + 1, p.rs_hi)
                             : LI(FINENV_LI_NEXT_START);
        // the row the next observation shows unless a cash shortage ends the episode here
        const int row_spec = last ? (p.auto_reset ? ns : di_s) : di_s + 1;
        sl_gather_closes(trl, p, di_s, lane);
        lds_barrier();                        // staging barrier: the trader reads its close rows
        SSTAMP(9);
        const unsigned long long valid_mask = __ballot(valid);
        float *const base = p.obs + (size_t)e0 * p.D;
        // market-data columns [64, D) of every row (NCH == 2: 64 < D <= 320): ONE 16-byte-per-lane load
        // and store per row at 4-byte aligned addresses, the last quad shifted back to end at D; 32
        // rows of loads ahead of their stores (see finenv_cashpenalty.hip)
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
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (g + j < nenv_w && qa) *quad_dst(g + j) = t[j];
            }
        }
        SSTAMP(10);
        // books of its assets (they do not depend on the decision): in flight while it waits
        double hx[kMaxN - kHalf], ax[kMaxN - kHalf], nx[kMaxN - kHalf];
#pragma unroll
        for (int u = 0; u < kMaxN - kHalf; ++u) {
            const int i = min(kHalf + u, N - 1);
            hx[u] = LV(FINENV_LV_HOLDINGS, i);
            ax[u] = LV(FINENV_LV_AVG_BUY_PRICE, i);
            nx[u] = LV(FINENV_LV_N_BUYS, i);
        }
        while (*flag == 0) __builtin_amdgcn_s_sleep(2);       // the trader has decided
        asm volatile("" ::: "memory");
        SSTAMP(11);
        const int dec = decided[lane], ef = eflags[lane];
        const bool adv = (ef & 1) != 0, keepb = (ef & 2) != 0, at_end_s = (ef & 4) != 0;
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
        // ---- pass 3 for assets kHalf.. (:388-428) -------------------------------------------------
        double ntr = 0.0;
        int fl = 0;
        const bool keep_cd = valid && !at_end_s && !(p.auto_reset && !adv);   // (a reset zeroes the books)
#pragma unroll
        for (int u = 0; u < kMaxN - kHalf; ++u) {
            const int i = kHalf + u;
            if (i >= N) continue;              // (continue, not break: keeps the loop fully unrollable)
            const double cl = trl[lane * kClStride + i], h = hx[u], abp = ax[u];
            if (keep_cd) LV(FINENV_LV_CLOSING_DIFF_AVG_BUY, i) = cl - (c.stoploss_penalty * abp);   // :350-352
            const SlBook b = sl_update(c, trx[u * kWave + lane], cl, h, abp, nx[u], keepb, ntr, fl);
            if (valid && adv) {
                LV(FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, i) = b.ps;
                LV(FINENV_LV_PREV_HOLDINGS, i) = h;
                LV(FINENV_LV_HOLDINGS, i) = b.hu;
                LV(FINENV_LV_N_BUYS, i) = b.nb;
                LV(FINENV_LV_AVG_BUY_PRICE, i) = b.abp;
            }
            row[1 + i] = (float)(adv ? b.hu : h);
        }
        ntrx[lane] = adv ? ntr : 0.0;
        aflags[lane] = adv ? fl : 0;
        SSTAMP(12);
        lds_barrier();                        // #1: pass 3 complete, rows[] hold every new holding
        float *const park = reinterpret_cast<float *>(trl);      // (nobody reads the closes any more)
#pragma unroll
        for (int el = 0; el < kWave; ++el) park[el * kWave + lane] = t0[el];
        lds_barrier();                        // #2: terminal observation / reset done, chunk 0 parked
        sl_head_store<NCH>(p.obs, p, e0, nenv_w, valid_mask, rows, park, lane, kWave / 2, kWave);
        if (NCH > 1 && nfix > 0 && W > 0) {
            unsigned long long m = fix;
#pragma unroll
            for (int j = 0; j < kFix; ++j) {
                if (m == 0ull) continue;
                const int el = __builtin_ctzll(m);
                m &= m - 1ull;
                if (qa) *quad_dst(el) = tf[j];
            }
            if (m != 0ull) sl_write_rows<true>(p.obs, p, e0, nenv_w, dec, m, rows, lane);
        }
        SSTAMP(13);
        return;
    }

    // ==================================== trader =================================================
    SSTAMP(0);
    int di = LI(FINENV_LI_DATE_INDEX);
    const int start = LI(FINENV_LI_START);
    double coh = LF(FINENV_LF_COH);
    double turb = c.use_turbulence ? LF(FINENV_LF_TURBULENCE) : 0.0;
    double sum_trades = LF(FINENV_LF_SUM_TRADES);
    double logged_total = LF(FINENV_LF_LOGGED_TOTAL), logged_cash = LF(FINENV_LF_LOGGED_CASH);
    double actual_num_trades = LF(FINENV_LF_ACTUAL_NUM_TRADES);
    stage_action_tile(rows, kRow, p.actions + (size_t)e0 * N, nenv_w, N, p.magicN, lane);
    const int step = di - start;                                                 // current_step
    const bool at_end = di == c.n_days - 1;                                      // :302
    lds_barrier();                            // staging barrier: close rows gathered by the streamer
    SSTAMP(1);

    double slp_sum = 0.0, lpp_sum = 0.0, add = 0.0;
    const double lt_old = logged_total, lc_old = logged_cash;
    double hS[kHalf], abS[kHalf], trS[kHalf];  // own assets, pass 2 -> pass 3 (statically indexed)
#pragma unroll
    for (int i = 0; i < kHalf; ++i) { hS[i] = 0.0; abS[i] = 0.0; trS[i] = 0.0; }
    // ---- pass 1 (last date only): reward terms of the state as the previous step left it (:304) --
    if (at_end) {
#pragma unroll
        for (int i0 = 0; i0 < kMaxN; i0 += kB) {
            if (i0 >= N) continue;
            double hh[kB], ps[kB], ph[kB], cd[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = min(i0 + j, N - 1);
                hh[j] = LV(FINENV_LV_HOLDINGS, i);
                ps[j] = LV(FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, i);
                ph[j] = LV(FINENV_LV_PREV_HOLDINGS, i);
                cd[j] = LV(FINENV_LV_CLOSING_DIFF_AVG_BUY, i);
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) { pin(hh[j]); pin(ps[j]); pin(ph[j]); pin(cd[j]); }
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = i0 + j;
                if (i >= N) continue;
                sum_trades += fabs((double)row[i]);                              // :294
                slp_sum += ph[j] * fmin(cd[j], 0.0);                             // :262,:273-275
                lpp_sum += hh[j] * fmin(ps[j], 0.0);                             // :263-265,:278-280
                add += hh[j] * fmax(ps[j], 0.0);                                 // :266-268,:283
                if (i < kHalf) hS[i < kHalf ? i : 0] = hh[j];
                else trx[(i >= kHalf ? i - kHalf : 0) * kWave + lane] = 0.0;
            }
        }
    }
    double reward = at_end ? sl_reward(c, step, lt_old, lc_old, slp_sum, lpp_sum, add) : 0.0;  // :304
    bool done = at_end;

    // ---- pass 2: transactions (:320-357), proceeds / spend (:363-370) --------------------------
    const float hmaxf = (float)c.hmax;
    const bool turbulent = c.use_turbulence && turb >= c.turbulence_threshold;
    const bool stop_armed = coh >= c.stoploss_penalty * c.initial_amount;        // :353
    double asset_value = 0.0, proceeds = 0.0, spend = 0.0, slp_new = 0.0;
    bool keep_buys = true;
    double coh_new = coh;
    const double coh_begin = coh;
    int audit_flags = at_end ? FINENV_AUDIT_F_LAST_DATE : 0;
    if (!at_end) {
#pragma unroll
        for (int i0 = 0; i0 < kMaxN; i0 += kB) {
            if (i0 >= N) continue;
            double hb[kB], ab[kB], pb[kB], pso[kB], cdo[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = min(i0 + j, N - 1);
                hb[j] = LV(FINENV_LV_HOLDINGS, i);
                ab[j] = LV(FINENV_LV_AVG_BUY_PRICE, i);
                pb[j] = LV(FINENV_LV_PREV_HOLDINGS, i);
                pso[j] = LV(FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, i);      // pass-1 terms (old books)
                cdo[j] = LV(FINENV_LV_CLOSING_DIFF_AVG_BUY, i);
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                pin(hb[j]); pin(ab[j]); pin(pb[j]); pin(pso[j]); pin(cdo[j]);
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int i = i0 + j;
                if (i >= N) continue;          // (continue, not break: keeps the loop fully unrollable)
                const double h = hb[j], cl = trl[lane * kClStride + i], abp = ab[j];
                sum_trades += fabs((double)row[i]);                              // :294
                slp_sum += pb[j] * fmin(cdo[j], 0.0);                            // :262,:273-275
                lpp_sum += h * fmin(pso[j], 0.0);                                // :263-265,:278-280
                add += h * fmax(pso[j], 0.0);                                    // :266-268,:283
                asset_value += h * cl;                                           // :311
                const float a32 = row[i] * hmaxf;                                // :321 (float32)
                double a = cl > 0.0 ? (double)a32 : 0.0;                         // :326
                a = turbulent ? -(h * cl) : a;                                   // :327-331
                double tr;
                if (DISCRETE) {                                                  // :333-343
                    const double q = cl > 0.0 ? sl_floordiv(a, cl) : 0.0;
                    const double inc = (double)c.shares_increment;
                    const double num = q >= 0.0 ? q : q + inc;
                    tr = sl_floordiv(num, inc) * inc;
                } else {
                    tr = cl > 0.0 ? a / cl : 0.0;                                // :345
                }
                tr = fmax(tr, -h);                                               // :348
                const double cd = cl - (c.stoploss_penalty * abp);               // :350-352
                slp_new += pb[j] * fmin(cd, 0.0);
                tr = (stop_armed && cd < 0.0) ? -h : tr;                         // :353-357
                audit_flags |= (stop_armed && cd < 0.0) ? FINENV_AUDIT_F_STOP_LOSS : 0;   // :359-360
                if (i < kHalf) {
                    hS[i < kHalf ? i : 0] = h;
                    abS[i < kHalf ? i : 0] = abp;
                    trS[i < kHalf ? i : 0] = tr;
                } else {
                    trx[(i >= kHalf ? i - kHalf : 0) * kWave + lane] = tr;   // for the streamer's pass 3
                }
                proceeds += (tr < 0.0 ? -tr : 0.0) * cl;                         // :363-364
                spend += (tr > 0.0 ? tr : 0.0) * cl;                             // :368-369
            }
        }
        reward = sl_reward(c, step, lt_old, lc_old, slp_sum, lpp_sum, add);      // :313 (stale log)
        logged_cash = coh;                                                       // :315-317
        logged_total = coh + asset_value;
        double costs = proceeds * c.sell_cost_pct;                               // :365
        const double coh1 = coh + proceeds;                                      // :366
        costs += spend * c.buy_cost_pct;                                         // :370
        audit_flags |= turbulent ? FINENV_AUDIT_F_TURBULENCE : 0;
        if (spend + costs > coh1) {                                              // :372
            audit_flags |= FINENV_AUDIT_F_CASH_SHORTAGE;
            if (c.patient) {                                                     // :373-378
                keep_buys = false;
                spend = 0.0;
                costs = 0.0;
            } else {                                                             // :379-383
                done = true;
                reward = sl_reward(c, step, logged_total, logged_cash, slp_new, lpp_sum, add);
            }
        }
        coh_new = coh1 - spend - costs;                                          // :414
    }
    const bool advance = !done;
    SSTAMP(2);
    // ---- the decision is published (LDS is in-order per wave: data first, then the flag) --------
    int ns_reset = 0;
    if (p.auto_reset && __any(done))
        ns_reset = p.rs_hi > 0 ? draw_start(p.rs_seed, e, LI(FINENV_LI_EPISODE) + 1, p.rs_hi)
                               : LI(FINENV_LI_NEXT_START);
    const int row_final = done ? (p.auto_reset ? ns_reset : di) : di + 1;
    decided[lane] = row_final;
    eflags[lane] = (advance ? 1 : 0) | (keep_buys ? 2 : 0) | (at_end ? 4 : 0);
    asm volatile("" ::: "memory");
    *flag = 1;

    // ---- pass 3 for assets 0..kHalf-1 (:388-428) ------------------------------------------------
    double ntr = 0.0;
    {
        double nbS[kHalf];
#pragma unroll
        for (int i = 0; i < kHalf; ++i) nbS[i] = LV(FINENV_LV_N_BUYS, min(i, N - 1));
        const bool keep_cd = valid && !at_end && !(p.auto_reset && !advance);    // (a reset zeroes the books)
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
            if (i >= N) continue;
            const double cl = trl[lane * kClStride + i], h = hS[i], abp = abS[i];
            if (keep_cd) LV(FINENV_LV_CLOSING_DIFF_AVG_BUY, i) = cl - (c.stoploss_penalty * abp);   // :350-352
            int fl = 0;
            double nt = 0.0;
            const SlBook b = sl_update(c, trS[i], cl, h, abp, nbS[i], keep_buys, nt, fl);
            if (advance) {
                ntr += nt;
                audit_flags |= fl;
                if (valid) {
                    LV(FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, i) = b.ps;
                    LV(FINENV_LV_PREV_HOLDINGS, i) = h;
                    LV(FINENV_LV_HOLDINGS, i) = b.hu;
                    LV(FINENV_LV_N_BUYS, i) = b.nb;
                    LV(FINENV_LV_AVG_BUY_PRICE, i) = b.abp;
                }
            }
            row[1 + i] = (float)(advance ? b.hu : h);
        }
    }
    if (advance) {
        coh = coh_new;
        di += 1;                                                                 // :430
        if (c.use_turbulence) turb = *at(p.panel.turb, (unsigned)di);            // :431-434
    }
    SSTAMP(3);
    lds_barrier();                            // #1: the streamer's pass 3 is complete
    SSTAMP(4);
    if (advance) {
        actual_num_trades = ntr + ntrx[lane];    // (counts: exact in any order)
        audit_flags |= aflags[lane];
    }
    if (p.audit != nullptr && valid) {      // harness log row (account_information / transaction_memory)
        double *au = p.audit + (size_t)e * (size_t)(FINENV_AUDIT_HEAD + N);
        au[FINENV_AUDIT_BEGIN_CASH] = coh_begin;                                 // :307
        au[FINENV_AUDIT_ASSET_VALUE] = asset_value;                              // :311
        au[FINENV_AUDIT_REWARD] = reward;
        au[FINENV_AUDIT_FLAGS] = (double)audit_flags;
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            if (i >= N) continue;
            const double trv = i < kHalf ? trS[i < kHalf ? i : 0]
                                         : trx[(i >= kHalf ? i - kHalf : 0) * kWave + lane];
            const double tr = at_end ? 0.0 : trv;
            au[FINENV_AUDIT_HEAD + i] = (tr > 0.0 && !keep_buys) ? 0.0 : tr;     // :376 / :385
        }
    }
    row[0] = (float)coh;
    if (valid) {
        *at(p.reward, (unsigned)e) = (float)reward;
        *at(p.done, (unsigned)e) = done ? 1 : 0;
        LF(FINENV_LF_SUM_TRADES) = sum_trades;
        LF(FINENV_LF_LOGGED_TOTAL) = logged_total;
        LF(FINENV_LF_LOGGED_CASH) = logged_cash;
        LF(FINENV_LF_ACTUAL_NUM_TRADES) = actual_num_trades;
    }
    wave_sync();
    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long done_mask = __ballot(done && valid);
    if (done_mask != 0ull) {
        if (p.term_obs != nullptr)
            sl_write_rows<true>(p.term_obs, p, e0, nenv_w, di, done_mask, rows, lane);   // once per episode
        if (p.auto_reset) {                                                      // reset()
            wave_sync();
            if (done) {
                di = ns_reset;
                coh = c.initial_amount;
                turb = 0.0;
                row[0] = (float)coh;
                for (int i = 0; i < N; ++i) row[1 + i] = 0.0f;
                if (valid) {
                    for (int i = 0; i < FINENV_STOPLOSS_BOOKS * N; ++i) LV(0, i) = 0.0;
                    LI(FINENV_LI_START) = ns_reset;
                    LI(FINENV_LI_EPISODE) += 1;
                    LF(FINENV_LF_SUM_TRADES) = 0.0;
                    LF(FINENV_LF_ACTUAL_NUM_TRADES) = 0.0;
                }
            }
            wave_sync();
        }
    }
    SSTAMP(5);
    lds_barrier();                            // #2: rows final, chunk 0 parked by the streamer
    sl_head_store<NCH>(p.obs, p, e0, nenv_w, valid_mask, rows, reinterpret_cast<const float *>(trl),
                       lane, 0, kWave / 2);
    SSTAMP(6);
    if (valid) {
        LF(FINENV_LF_COH) = coh;
        LI(FINENV_LI_DATE_INDEX) = di;
        if (c.use_turbulence) LF(FINENV_LF_TURBULENCE) = turb;
    }
}

}  // namespace

struct finenv_stoploss {
    int32_t rs_hi;
    unsigned long long rs_seed;
    double *audit;
    int device;           // HIP device that owns the bound state block (-1 before bind)
    finenv_stoploss_config cfg;
    finenv_stoploss_panel panel;
    finenv_stoploss_state st;
    int bound;
    int D;
    uint32_t magicN;
    char err[256];
};

namespace {
int sl_fail(finenv_stoploss *h, int code, const char *msg)
{
    if (h) snprintf(h->err, sizeof(h->err), "%s", msg);
    return code;
}
int sl_check(finenv_stoploss *h, const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        return FINENV_ERR_HIP;
    }
    return FINENV_OK;
}
SlParams sl_params(const finenv_stoploss *h)
{
    SlParams p;
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
dim3 sl_grid(int E)
{
    const int waves = (E + kWave - 1) / kWave;
    return dim3((unsigned)((waves + kWaves - 1) / kWaves));
}
}  // namespace

extern "C" {

int finenv_stoploss_create(const finenv_stoploss_config *cfg, finenv_stoploss **out)
{
    if (!cfg || !out) return FINENV_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_envs < 1 || cfg->n_assets < 1 || cfg->n_assets > FINENV_STOPLOSS_MAX_ASSETS ||
        cfg->n_cols < 0 || cfg->n_days < 1 || cfg->shares_increment < 1 || !(cfg->hmax >= 0) ||
        !(cfg->initial_amount > 0))
        return FINENV_ERR_INVALID;
    const long long E = cfg->n_envs, N = cfg->n_assets, T = cfg->n_days;
    const long long D = 1 + N + N * cfg->n_cols, lim = (1ll << 32) - 1;
    if ((FINENV_STOPLOSS_F64_FIELDS + FINENV_STOPLOSS_BOOKS * N) * E * 8 > lim ||
        T * N * cfg->n_cols * 4 > lim || T * N * 8 > lim || 64 * D * 4 > lim || E * N * 4 > lim)
        return FINENV_ERR_INVALID;
    finenv_stoploss *h = new (std::nothrow) finenv_stoploss;
    if (!h) return FINENV_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->device = -1;
    h->cfg = *cfg;
    h->D = (int)D;
    h->magicN = N >= 2 ? (uint32_t)(((1ull << 32) + N - 1) / (unsigned long long)N) : 0u;
    *out = h;
    return FINENV_OK;
}

void finenv_stoploss_destroy(finenv_stoploss *h) { delete h; }
const char *finenv_stoploss_last_error(const finenv_stoploss *h)
{
    return h ? h->err : "null handle";
}
int finenv_stoploss_obs_dim(const finenv_stoploss *h) { return h ? h->D : FINENV_ERR_INVALID; }

int finenv_stoploss_bind(finenv_stoploss *h, const finenv_stoploss_panel *panel,
                         const finenv_stoploss_state *st)
{
    if (!h || !panel || !st) return FINENV_ERR_INVALID;
    if (!panel->close || (!panel->info && h->cfg.n_cols > 0) ||
        (!panel->turb && h->cfg.use_turbulence) || !st->f64 || !st->i32)
        return sl_fail(h, FINENV_ERR_INVALID, "bind: null pointer");
    h->panel = *panel;
    h->st = *st;
    h->device = finenv_host::pointer_device(st->f64);
    h->bound = 1;
    return FINENV_OK;
}

int finenv_stoploss_set_random_start(finenv_stoploss *h, int32_t hi, uint64_t seed)
{
    if (!h || hi < 0 || hi > h->cfg.n_days) return FINENV_ERR_INVALID;
    h->rs_hi = hi;
    h->rs_seed = seed;
    return FINENV_OK;
}

int finenv_stoploss_set_audit(finenv_stoploss *h, double *audit)
{
    if (!h) return FINENV_ERR_INVALID;
    h->audit = audit;
    return FINENV_OK;
}

int finenv_stoploss_reset(finenv_stoploss *h, const uint8_t *mask, float *obs_out, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return sl_fail(h, FINENV_ERR_UNBOUND, "reset: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    SlParams p = sl_params(h);
    p.mask = mask;
    p.obs = obs_out;
    hipLaunchKernelGGL((stoploss_kernel<true>), sl_grid(h->cfg.n_envs), dim3(kWave * kWaves), 0,
                       (hipStream_t)stream, p);
    return sl_check(h, "stoploss_reset");
}

int finenv_stoploss_step(finenv_stoploss *h, const float *actions, float *obs, float *reward,
                         uint8_t *done, float *term_obs, int32_t auto_reset, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return sl_fail(h, FINENV_ERR_UNBOUND, "step: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    if (!actions || !obs || !reward || !done)
        return sl_fail(h, FINENV_ERR_INVALID, "step: null actions/obs/reward/done");
    SlParams p = sl_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.auto_reset = auto_reset;
#ifdef FINENV_DIAG
    p.dbg = g_finenv_dbg;
#endif
    const dim3 grid = sl_grid(h->cfg.n_envs), block(kWave * kWaves);
    const dim3 grid2((unsigned)((h->cfg.n_envs + kWave - 1) / kWave));   // one block per 64 envs
#define SL_LAUNCH2(NCH_)                                                                         \
    do {                                                                                         \
        if (h->cfg.discrete_actions)                                                             \
            hipLaunchKernelGGL((stoploss_step2_kernel<NCH_, true>), grid2, block, 0,             \
                               (hipStream_t)stream, p);                                          \
        else                                                                                     \
            hipLaunchKernelGGL((stoploss_step2_kernel<NCH_, false>), grid2, block, 0,            \
                               (hipStream_t)stream, p);                                          \
    } while (0)
    // 1 = rows of one chunk, 2 = rows of up to 320 columns (the streamer copies the market data as
    // 16-byte quads), wider rows: the one-wave kernel
    if (h->D <= kWave) SL_LAUNCH2(1);
    else if (h->D <= kWave + 4 * kWave) SL_LAUNCH2(2);
    else hipLaunchKernelGGL((stoploss_kernel<false>), grid, block, 0, (hipStream_t)stream, p);
#undef SL_LAUNCH2
    return sl_check(h, "stoploss_step");
}

}  // extern "C"
