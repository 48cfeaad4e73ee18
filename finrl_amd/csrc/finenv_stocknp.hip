// finenv_stocknp.hip -- MI355X (gfx950) kernel + C ABI for the batched array-state
// StockTradingEnv (finrl/meta/env_stock_trading/env_stocktrading_np.py).
//
// lane = env, one wave per 64 envs, four independent waves per block.  Trades run in ticker
// index order (this env does not sort), serial through `amount`.  The arithmetic reproduces
// the reference under NumPy >= 2 bit for bit: amount / total_asset / gamma_reward carry a
// per-env dtype tag (python float / float32 / float64) and every operation is evaluated in
// the dtype NumPy's promotion rules pick (see include/finenv.h and oracle/stocknp_oracle.c).
// HBM-bound: 1993 algorithmic bytes per env-step at DOW30 x 8 (obs row 1332 B).

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

constexpr int kWave = 64;
constexpr int kMaxN = FINENV_STOCKNP_MAX_TICKERS;
constexpr int kRowA = kMaxN + 1;                       // action rows, stride 33
constexpr int kRowH = 2 * kMaxN + 1;                   // obs heads [amount|stocks|cool], stride 65
constexpr int kWaves = 4;
// heads/actions + stocks + cool + price row
constexpr int kLdsPerWave = kWave * kRowH + 2 * kMaxN * kWave + kMaxN;

struct NpParams {
    finenv_stocknp_config cfg;
    finenv_stocknp_panel panel;
    finenv_stocknp_state st;
    const float *actions;
    float *obs;
    float *reward;
    uint8_t *done;
    float *term_obs;
    const uint8_t *mask;
    int32_t auto_reset;
    int32_t D;
    int32_t obs_pitch;              // row pitch of obs in floats (>= D; D = packed rows)
    uint32_t magicN;
    unsigned long long *dbg;      // FINENV_DIAG builds only: [wave][16] s_memrealtime stamps
    int32_t diag;                 // FINENV_DIAG builds only: experiment switches (env FINENV_DIAG)
};

#ifdef FINENV_DIAG
#define NSTAMP(k)                                                                           \
    do {                                                                                    \
        if (p.dbg != nullptr && lane == 0) {                                                \
            __builtin_amdgcn_sched_barrier(0);                                              \
            p.dbg[(size_t)(e0 / kWave) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
#define NDIAG(bit) (p.diag & (bit))
#else
#define NSTAMP(k) do { } while (0)
#define NDIAG(bit) 0
#endif

#define NF(fld) (*at(p.st.f64, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define NI(fld) (*at(p.st.i32, (unsigned)(fld) * (unsigned)E + (unsigned)e))
#define NS(k, i) (*at(p.st.f32, ((unsigned)(k) * (unsigned)N + (unsigned)(i)) * (unsigned)E + (unsigned)e))

struct Num { double v; int tag; };
__device__ __forceinline__ Num mk(double v, int tag) { Num r; r.v = v; r.tag = tag; return r; }
__device__ __forceinline__ int promote(int a, int b)
{
    return a == FINENV_NT_PY ? b : (b == FINENV_NT_PY ? a : max(a, b));
}
__device__ __forceinline__ Num n_add(Num a, Num b)
{
    const int t = promote(a.tag, b.tag);
    const double r32 = (double)((float)a.v + (float)b.v), r64 = a.v + b.v;
    return mk(t == FINENV_NT_F32 ? r32 : r64, t);
}
__device__ __forceinline__ Num n_sub(Num a, Num b)
{
    const int t = promote(a.tag, b.tag);
    const double r32 = (double)((float)a.v - (float)b.v), r64 = a.v - b.v;
    return mk(t == FINENV_NT_F32 ? r32 : r64, t);
}
__device__ __forceinline__ Num n_mul(Num a, Num b)
{
    const int t = promote(a.tag, b.tag);
    const double r32 = (double)((float)a.v * (float)b.v), r64 = a.v * b.v;
    return mk(t == FINENV_NT_F32 ? r32 : r64, t);
}
__device__ __forceinline__ Num n_div(Num a, Num b)
{
    const int t = promote(a.tag, b.tag);
    const double r32 = (double)((float)a.v / (float)b.v), r64 = a.v / b.v;
    return mk(t == FINENV_NT_F32 ? r32 : r64, t);
}
// exact floor(a/d) for d > 0 (== numpy floor_divide in float64 and, for float32 operands,
// == npy_floor_dividef: both return the true floor at these magnitudes)
__device__ __forceinline__ double floordiv_true(double a, double d)
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
__device__ __forceinline__ Num n_floordiv(Num a, Num b)      // b > 0
{
    const int t = promote(a.tag, b.tag);
    // float32 case: operands are first rounded to float32, the quotient is exact in fp64
    const double av = (t == FINENV_NT_F32) ? (double)(float)a.v : a.v;
    return mk(floordiv_true(av, b.v), t);
}

// heads[el*kRowH + 0] = amount*2^-12 (f32), [1..N] stocks*2^-6, [1+N..2N] cool_down
template <bool kCompact = false>
__device__ __forceinline__ void np_write_rows(float *__restrict__ dst, const NpParams &p, int e0,
                                              int nenv_w, int row_day,
                                              unsigned long long lane_mask, const float *heads,
                                              int lane, int k_lo = 0, int k_hi = 1 << 30)
{
    const int N = p.cfg.n_tickers, D = p.D;
    const int pitch = dst == p.obs ? p.obs_pitch : D;      // (terminal observations stay packed)
    write_obs_rows_generic<8, 16, kCompact>(
        dst, p.panel.obs_tmpl, D, e0, nenv_w, row_day, lane_mask, heads, kRowH, lane,
        [=](int day, int col) { return day * D + col; },
        [=](int col) {                                       // amount | ... | stocks | cool_down
            const int hidx = col - 3 - N;
            return col == 0 ? 0 : ((hidx >= 0 && hidx < 2 * N) ? 1 + hidx : -1);
        }, k_lo, k_hi, pitch);
}

// (stocks * price).sum() in float32, NumPy pairwise order (8 accumulators, n < 128)
// prow != nullptr: the wave's envs share the day and its price row sits in LDS (one coalesced
// load for the whole step instead of a global round trip per batch of 8 tickers)
__device__ __forceinline__ float holdings_value(const float *scol, const float *__restrict__ price,
                                                unsigned pb, int N, const float *prow = nullptr)
{
    if (prow != nullptr) {                   // LDS row: same sums, no global load
        auto prodl = [&](int i) { return scol[i * kWave] * prow[i]; };
        float suml;
        if (N < 8) {
            suml = 0.0f;
            for (int i = 0; i < N; ++i) suml += prodl(i);
        } else {
            float r8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r8[j] = prodl(j);
            const int full = N - (N & 7);
            for (int i = 8; i < full; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r8[j] += prodl(i + j);
            }
            suml = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
            for (int i = full; i < N; ++i) suml += prodl(i);
        }
        return suml;
    }
    auto prod = [&](int i) { return scol[i * kWave] * *at(price, pb + (unsigned)i); };
    float sum;
    if (N < 8) {
        sum = 0.0f;
        for (int i = 0; i < N; ++i) sum += prod(i);
    } else {
        float r8[8];
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
    return sum;
}

// step launches carry a second set of kWaves "streamer" waves per block (role 1): done and the
// panel row of the next observation depend only on the day counter (:106, :137), so the chunks
// of the observation rows that hold no per-env value (4 of 6 at DOW30x8) are streamed from the
// first microsecond on, beside the trade arithmetic, instead of after it -- 16 bytes per lane, one
// store per row and 256 columns: this wave is slot-bound (a wave keeps 64 vector-memory operations in
// flight, ~60 ns each under load), and four dword stores per row kept it busy for 17 us, longer
// than the trader's whole chain.  Done after ~11 us, it waits at the hand-off barrier for the trader's
// results and then writes the upper half of the rows of the chunks that hold amount / stocks /
// cool-downs (16 bytes per lane: two rows per store at DOW30) and of the books; the trader the
// lower halves and the scalars.
// The trader issues EVERY global load it needs before the one block barrier that releases the
// streamers' stores: the CU's memory pipeline serves requests in order, and loads queued behind
// four streamers' stores came back after 6 us (profiles/r02_stocknp_phase_timeline.txt).
template <bool RESET_ONLY>
__global__ void __launch_bounds__(kWave *kWaves *(RESET_ONLY ? 1 : 2)) stocknp_kernel(const NpParams p)
{
    __shared__ float lds_all[kWaves * kLdsPerWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = (threadIdx.x >> 6) % kWaves;
    const int role = (threadIdx.x >> 6) / kWaves;          // 0 trader, 1 streamer (step only)
    float *heads = lds_all + wib * kLdsPerWave;            // [env][kRowH] (actions use stride kRowA)
    float *stk = heads + kWave * kRowH;                    // [ticker][lane]
    float *cdl = stk + kMaxN * kWave;                      // [ticker][lane]
    const int E = p.cfg.n_envs, N = p.cfg.n_tickers;
    const int e0 = (blockIdx.x * kWaves + wib) * kWave;
    if (e0 >= E) return;
    const int nenv_w = min(kWave, E - e0);
    const bool valid = lane < nenv_w;
    const int e = valid ? e0 + lane : e0;
    float *scol = stk + lane, *ccol = cdl + lane;
    float *head = heads + lane * kRowH;
    float *prow_lds = cdl + kMaxN * kWave;                 // [ticker] shared price row (lock-step days)

    // reset(): day 0, start state, total_asset = amount + (stocks*price[0]).sum()  (:80-101)
    auto do_reset = [&](Num &amount, Num &ta, Num &gr, Num &ita) {
        for (int i = 0; i < N; ++i) {
            scol[i * kWave] = NS(2, i);
            ccol[i * kWave] = 0.0f;
        }
        amount = mk(NF(FINENV_NF_AMOUNT0), NI(FINENV_NI_AMOUNT0_TAG));
        ta = n_add(amount, mk((double)holdings_value(scol, p.panel.price, 0u, N), FINENV_NT_F32));
        ita = ta;
        gr = mk(0.0, FINENV_NT_PY);
    };
    auto fill_head = [&](Num amount) {
        // env_nas100_wrds.py:154: Python's max(self.amount, 1e4) keeps self.amount (and its NumPy
        // scalar type) unless the floor is strictly larger, then it is the Python float
        const double fl = p.cfg.obs_amount_floor;
        const Num shown = (fl > 0.0 && fl > amount.v) ? mk(fl, FINENV_NT_PY) : amount;
        head[0] = (float)n_mul(shown, mk(0x1p-12, FINENV_NT_PY)).v;              // :150
        for (int i = 0; i < N; ++i) {
            head[1 + i] = scol[i * kWave] * 0x1p-6f;
            head[1 + N + i] = ccol[i * kWave];
        }
    };
    // the books (stocks, cool-downs) of tickers [i_lo, i_hi) from LDS back to the state block
    auto store_books = [&](int i_lo, int i_hi) {
        for (int i = i_lo; i < i_hi; ++i) {
            NS(0, i) = scol[i * kWave];
            NS(1, i) = ccol[i * kWave];
        }
    };
    auto store_state = [&](Num amount, Num ta, Num gr, Num ita, int rtag, int day, int i_lo = 0,
                           int i_hi = 1 << 30) {
        NF(FINENV_NF_AMOUNT) = amount.v;
        NF(FINENV_NF_TOTAL_ASSET) = ta.v;
        NF(FINENV_NF_GAMMA_REWARD) = gr.v;
        NF(FINENV_NF_INITIAL_TOTAL_ASSET) = ita.v;
        NI(FINENV_NI_TAGS) = amount.tag | (ta.tag << 2) | (gr.tag << 4) | (ita.tag << 6) | (rtag << 8);
        NI(FINENV_NI_DAY) = day;
        store_books(i_lo, min(i_hi, N));
    };

    if (RESET_ONLY) {
        const bool sel = valid && (p.mask == nullptr || p.mask[e] != 0);
        Num amount, ta, gr, ita;
        do_reset(amount, ta, gr, ita);
        if (sel) store_state(amount, ta, gr, ita, (NI(FINENV_NI_TAGS) >> 8) & 3, 0);
        if (p.obs == nullptr) return;
        fill_head(amount);
        wave_sync();
        np_write_rows(p.obs, p, e0, nenv_w, 0, __ballot(sel), heads, lane);
        return;
    }

    const int kpatch = (2 + 3 * N) / kWave + 1;            // chunks that hold amount / stocks / cool_down
    // column -> index into heads[el * kRowH + .] (amount | stocks | cool_down), or -1 (market data)
    auto head_widx = [N](int col) {
        const int hidx = col - 3 - N;
        return col == 0 ? 0 : ((hidx >= 0 && hidx < 2 * N) ? 1 + hidx : -1);
    };
    // ---- the chunks that hold amount / stocks / cool-downs, quad form: 16 bytes per lane, lane (sub, q)
    // writes columns 4q .. 4q + 3 of row g * rpi + sub (two rows per store at DOW30); per-env values
    // from the heads in LDS, market-data values from the quad `hq` every lane loaded for its columns
    typedef float np_f4 __attribute__((ext_vector_type(4)));
    typedef np_f4 np_f4u __attribute__((aligned(4)));
    const int lpr = 16 * kpatch;                           // lanes per row
    const bool quad_ok = kpatch <= 2 && p.D >= kWave * kpatch;
    auto head_quads = [&](const np_f4 hq, int el_lo, int el_hi) {
        const int rpi = kWave / lpr, sub = lane / lpr, q = lane & (lpr - 1);
        int w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = head_widx(4 * q + u);
        char *const hb = reinterpret_cast<char *>(p.obs + (size_t)e0 * p.obs_pitch);
        constexpr int kB = 8;                 // stores per batch: their LDS reads before their stores
        const int hi = min(nenv_w, el_hi);
        for (int g0 = el_lo / rpi; g0 * rpi < hi; g0 += kB) {
            float pv4[kB][4];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int el = min((g0 + j) * rpi + sub, kWave - 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) pv4[j][u] = heads[el * kRowH + (w[u] >= 0 ? w[u] : 0)];
            }
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int el = (g0 + j) * rpi + sub;
                if (el >= hi) continue;
                np_f4 v;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = w[u] >= 0 ? pv4[j][u] : hq[u];
                *reinterpret_cast<np_f4u *>(hb + (size_t)((unsigned)(el * p.obs_pitch + 4 * q) * 4u)) = v;
            }
        }
    };
    auto head_quad_load = [&](int row) -> np_f4 {
        return *reinterpret_cast<const np_f4u *>(reinterpret_cast<const char *>(p.panel.obs_tmpl) +
                                                 (size_t)((unsigned)(row * p.D + 4 * (lane & (lpr - 1))) * 4u));
    };
    if (role == 0) NSTAMP(0);
    if (role == 1) {
        int day_s = NI(FINENV_NI_DAY) + 1;
        // the day counter is read (and has arrived) before the block-wide barrier; the traders
        // overwrite it only at the very end, long after they passed the same barrier
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(day_s) : : "memory");
        lds_barrier();
        const bool done_s = day_s == p.cfg.n_days - 1;
        const unsigned long long vm = __ballot(valid), dm = __ballot(done_s && valid);
        if (dm != 0ull && p.term_obs != nullptr)
            np_write_rows<true>(p.term_obs, p, e0, nenv_w, day_s, dm, heads, lane, kpatch);
        const int rd = (done_s && p.auto_reset) ? 0 : day_s;
        const int rd0 = __builtin_amdgcn_readfirstlane(rd);
        const bool same_row = __all(!valid || rd == rd0);
        // what follows the streaming is decided from the day counters alone, identically in the trader
        const bool share = quad_ok && dm == 0ull && same_row;
        np_f4 hq = {0.0f, 0.0f, 0.0f, 0.0f};
        if (share) hq = head_quad_load(rd0);           // (a load behind the stores below would wait for them)
        NSTAMP(9);
        // market-data chunks.  Every env of the wave on the same panel row: 16 bytes per lane, ONE store per
        // row and 256 columns, the last quad shifted back to end at D (this wave is slot-bound: four dword
        // stores per row took 17 us for 67 MB, same process / same buffers 26.6 -> 24.4 us per step)
        const int c0 = kpatch * kWave, span = p.D - c0;
        if (same_row && span >= 4) {
            const int nq = (span + 3) >> 2;
            char *const ob = reinterpret_cast<char *>(p.obs + (size_t)e0 * p.obs_pitch);
            for (int q0 = 0; q0 < nq; q0 += kWave) {
                const bool qa = q0 + lane < nq;
                const int start = qa ? min(c0 + 4 * (q0 + lane), p.D - 4) : c0;
                const np_f4 t = *reinterpret_cast<const np_f4u *>(
                    reinterpret_cast<const char *>(p.panel.obs_tmpl) + (size_t)((unsigned)(rd0 * p.D + start) * 4u));
#pragma unroll 4
                for (int el = 0; el < nenv_w; ++el)
                    if (qa) *reinterpret_cast<np_f4u *>(ob + (size_t)((unsigned)(el * p.obs_pitch + start) * 4u)) = t;
            }
        } else {
            np_write_rows(p.obs, p, e0, nenv_w, rd, vm, heads, lane, kpatch);
        }
        NSTAMP(10);
        // hand-off: this wave is done ~3 us before the trader has its results; once they are final in LDS it
        // takes the upper half of the head rows and of the books (the trader the lower halves and the scalars)
        lds_barrier();                                                        // #2
        if (share && !NDIAG(1)) {
            head_quads(hq, kWave / 2, kWave);
            if (valid) store_books(N / 2, N);
        }
        return;
    }
    // ---- trader: all global loads first -- the day counter (the second round trip hangs on it),
    // scalars, stocks / cool-downs, the action tile (flat and coalesced) -- then the barrier that
    // lets the streamers start storing, then the LDS images ---------------------------------------
    const int day = NI(FINENV_NI_DAY) + 1;                                        // :106
    const int tags = NI(FINENV_NI_TAGS);
    Num amount = mk(NF(FINENV_NF_AMOUNT), tags & 3);
    const Num ta_old = mk(NF(FINENV_NF_TOTAL_ASSET), (tags >> 2) & 3);
    Num gr = mk(NF(FINENV_NF_GAMMA_REWARD), (tags >> 4) & 3);
    Num ita = mk(NF(FINENV_NF_INITIAL_TOTAL_ASSET), (tags >> 6) & 3);
    float sv[kMaxN], cv[kMaxN], av[kMaxN];
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) {
        sv[i] = NS(0, min(i, N - 1));
        cv[i] = NS(1, min(i, N - 1));
    }
    // (the action tile staged by the idle streamer instead: 25.5 vs 24.4 us -- staging is the 23.6 MB of
    //  reads at the chip's rate whichever wave issues them, and the streamer then starts 2 us later)
    const int a_total = nenv_w * N;
    const float *const a_src = p.actions + (size_t)e0 * N;
#pragma unroll
    for (int j = 0; j < kMaxN; ++j) {
        const int f = j * kWave + lane;
        av[j] = *at(a_src, (unsigned)(f < a_total ? f : a_total - 1));
    }
    const unsigned pb = (unsigned)(day * N);
    const int day0 = __builtin_amdgcn_readfirstlane(day);
    const bool uni = __all(day == day0);                   // lock-step batch: one price row per wave
    float prv = 0.0f;
    if (uni && lane < kMaxN) prv = *at(p.panel.price, (unsigned)(day0 * N + min(lane, N - 1)));
    const float *prow = uni ? prow_lds : nullptr;
    // market-data values of the head chunks of the NEXT observation (row `day`), 16 bytes per lane:
    // lane (sub, q) holds columns 4q .. 4q + 3 of the 64 * kpatch head columns
    np_f4 head_q = {0.0f, 0.0f, 0.0f, 0.0f};
    if (quad_ok) head_q = head_quad_load(day0);
    const bool calm = *at(p.panel.turb_bool, (unsigned)day) == 0.0f;              // :110
    // (releasing the streamers before the day-dependent loads above instead: 27.0 vs 26.65 us, same box)
    lds_barrier();                           // pairs with the streamers' barrier (see above)
#pragma unroll
    for (int j = 0; j < kMaxN; ++j) {        // action tile -> rows of stride kRowA in the heads region
        const int f = j * kWave + lane;
        if (f < a_total) {
            const int el = (N == 1) ? f : (int)__umulhi((unsigned)f, p.magicN);
            heads[el * kRowA + (f - el * N)] = av[j];
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxN; ++i) {
        if (i >= N) continue;
        cv[i] += 1.0f;                                                            // :108
        scol[i * kWave] = sv[i];
        ccol[i * kWave] = cv[i];
    }
    if (uni && lane < kMaxN) prow_lds[lane] = prv;
    wave_sync();
    NSTAMP(1);
    const float *arow = heads + lane * kRowA;
    const float ms = (float)p.cfg.max_stock;
    const Num one_m = mk(1 - p.cfg.sell_cost_pct, FINENV_NT_PY);
    const Num one_p = mk(1 + p.cfg.buy_cost_pct, FINENV_NT_PY);
    const int min_action = p.cfg.min_action;

    // sells then buys, ticker index order (:112-129); liquidation when turbulent (:131-134)
    constexpr int kPB = 8;                    // price loads per batch
    // Steady state: once an env has traded an integer share count its cash is a np.float64 and
    // stays one (float64 dominates every later promotion).  When that holds for the whole wave the
    // trade loops run on plain doubles -- the per-operation dtype dispatch of Num (both roundings
    // computed, tag-selected) is most of this kernel's arithmetic; any other tag mix takes the
    // generic loops below.  Same operations, same order, same roundings in both.
    bool head_filled = false;
    if (__all(amount.tag == FINENV_NT_F64)) {
        // Statically indexed registers, branch-free per ticker (selects), reciprocals of the prices
        // computed off the cash chain: the first form kept stocks / cool-downs / actions in LDS and
        // paid an LDS round trip inside every divergent per-ticker block (10.4 us of a 34 us step).
        double amt = amount.v;
        float (&sr)[kMaxN] = sv, (&cr)[kMaxN] = cv;          // as loaded (cool-downs already + 1)
        float pr_[kMaxN];
        int ai[kMaxN];
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            const int ic = min(i, N - 1);
            ai[i] = (int)(arow[ic] * ms);                                         // :104
            pr_[i] = 0.0f;
        }
        if (uni) {                           // (a branch, not a select: no global load at all)
#pragma unroll
            for (int i = 0; i < kMaxN; ++i) pr_[i] = prow_lds[min(i, N - 1)];
        } else {
#pragma unroll
            for (int i = 0; i < kMaxN; ++i) pr_[i] = *at(p.panel.price, pb + (unsigned)min(i, N - 1));
        }
        NSTAMP(11);
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {                                         // sells :112-119
            if (i >= N) continue;             // (continue, not break: keeps the loop fully unrollable)
            const int a = ai[i];
            const float pr = pr_[i];
            const bool ok = calm && a < -min_action && pr > 0.0f;
            const float s = sr[i];
            const double want = (double)(-a);
            const bool is_int = want < (double)s;            // min(stocks, -a) -> -a (np.int64)
            const double sell = is_int ? want : (double)s;
            const float s_new = (float)((double)s - sell);
            // int64 share count: float64 product chain; float32 count: float32 chain
            const double t64 = ((double)pr * sell) * one_m.v;
            const double t32 = (double)((pr * (float)sell) * (float)one_m.v);
            const double amt_new = amt + (is_int ? t64 : t32);
            sr[i] = ok ? s_new : s;
            cr[i] = ok ? 0.0f : cr[i];
            amt = ok ? amt_new : amt;
        }
        NSTAMP(7);
        double xr[kMaxN];
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            const double d = (double)pr_[i];
            const double x = __builtin_amdgcn_rcp(d);
            xr[i] = fma(fma(-d, x, 1.0), x, x);
        }
        NSTAMP(12);
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {                                         // buys :120-129
            if (i >= N) continue;
            const int a = ai[i];
            const double d = (double)pr_[i];
            const bool ok = calm && a > min_action && pr_[i] > 0.0f;
            // amount // price (floordiv_true with the reciprocal hoisted).  floor(amt * x) is within 1
            // of the true floor for quotients below 2^40 and the exact sign of the FMA remainder says
            // which way: ONE fix-up round; above 2^40 min() returns the (int32) action regardless.
            double q = floor(amt * xr[i]);
            const double r = fma(-q, d, amt);
            q += (r < 0.0) ? -1.0 : ((r >= d) ? 1.0 : 0.0);
            const double buy = ((double)a < q) ? (double)a : q;                   // min(q, a)
            const float s_new = (float)((double)sr[i] + buy);
            const double amt_new = amt - ((d * buy) * one_p.v);
            sr[i] = ok ? s_new : sr[i];
            cr[i] = ok ? 0.0f : cr[i];
            amt = ok ? amt_new : amt;
        }
        amount = mk(amt, FINENV_NT_F64);
        NSTAMP(8);
        // books back to LDS and, in the same pass, this env's observation head (amount | stocks |
        // cool-downs) over the consumed action rows -- every lane has read its own action row above
        wave_sync();
        {
            const double fl = p.cfg.obs_amount_floor;
            const Num shown = (fl > 0.0 && fl > amount.v) ? mk(fl, FINENV_NT_PY) : amount;
            head[0] = (float)n_mul(shown, mk(0x1p-12, FINENV_NT_PY)).v;          // :150
        }
#pragma unroll
        for (int i = 0; i < kMaxN; ++i) {
            if (i >= N) continue;
            scol[i * kWave] = sr[i];
            ccol[i * kWave] = cr[i];
            head[1 + i] = sr[i] * 0x1p-6f;
            head[1 + N + i] = cr[i];
        }
        head_filled = calm;                  // (a turbulent day still liquidates below)
    } else {
        for (int i0 = 0; i0 < N; i0 += kPB) {
            float prb[kPB];
    #pragma unroll
            for (int j = 0; j < kPB; ++j) prb[j] = 0.0f;
            if (uni) {                       // (a branch, not a select: no global load at all)
#pragma unroll
                for (int j = 0; j < kPB; ++j) prb[j] = prow_lds[min(i0 + j, N - 1)];
            } else {
#pragma unroll
                for (int j = 0; j < kPB; ++j)
                    prb[j] = *at(p.panel.price, pb + (unsigned)min(i0 + j, N - 1));
            }
    #pragma unroll
            for (int j = 0; j < kPB; ++j) pin(prb[j]);
    #pragma unroll
            for (int j = 0; j < kPB; ++j) {
            const int i = i0 + j;
            if (i >= N) break;
            const int a = (int)(arow[i] * ms);                                        // :104
            const float pr = prb[j];
            if (calm && a < -min_action && pr > 0.0f) {
                const float s = scol[i * kWave];
                const double want = (double)(-a);
                const bool is_int = want < (double)s;            // min(stocks, -a) -> -a (np.int64)
                const double sell = is_int ? want : (double)s;
                scol[i * kWave] = (float)((double)s - sell);
                // price(f32) * sell: int64 operand -> float64; float32 operand -> float32
                const Num t0 = is_int ? mk((double)pr * sell, FINENV_NT_F64)
                                      : mk((double)(pr * (float)sell), FINENV_NT_F32);
                amount = n_add(amount, n_mul(t0, one_m));
                ccol[i * kWave] = 0.0f;
            }
            }
        }
        for (int i0 = 0; i0 < N; i0 += kPB) {
            float prb[kPB];
    #pragma unroll
            for (int j = 0; j < kPB; ++j) prb[j] = 0.0f;
            if (uni) {                       // (a branch, not a select: no global load at all)
#pragma unroll
                for (int j = 0; j < kPB; ++j) prb[j] = prow_lds[min(i0 + j, N - 1)];
            } else {
#pragma unroll
                for (int j = 0; j < kPB; ++j)
                    prb[j] = *at(p.panel.price, pb + (unsigned)min(i0 + j, N - 1));
            }
    #pragma unroll
            for (int j = 0; j < kPB; ++j) pin(prb[j]);
    #pragma unroll
            for (int j = 0; j < kPB; ++j) {
            const int i = i0 + j;
            if (i >= N) break;
            const int a = (int)(arow[i] * ms);
            const float pr = prb[j];
            if (calm && a > min_action && pr > 0.0f) {
                const Num q = n_floordiv(amount, mk((double)pr, FINENV_NT_F32));      // amount // price
                const bool is_int = (double)a < q.v;             // min(q, a) -> a (np.int64)
                const double buy = is_int ? (double)a : q.v;
                const Num t0 = is_int ? mk((double)pr * buy, FINENV_NT_F64)
                                      : n_mul(mk((double)pr, FINENV_NT_F32), q);
                const float s = scol[i * kWave];
                scol[i * kWave] = (float)((double)s + buy);
                amount = n_sub(amount, n_mul(t0, one_p));
                ccol[i * kWave] = 0.0f;
            }
            }
        }
    }
    NSTAMP(2);
    if (!calm) {
        const Num t0 = mk((double)holdings_value(scol, p.panel.price, pb, N, prow), FINENV_NT_F32);
        amount = n_add(amount, n_mul(t0, one_m));
        for (int i = 0; i < N; ++i) {
            scol[i * kWave] = 0.0f;
            ccol[i * kWave] = 0.0f;
        }
    }
    // total asset, reward, discounted return (:137-145)
    Num ta = n_add(amount, mk((double)holdings_value(scol, p.panel.price, pb, N, prow), FINENV_NT_F32));
    Num r = n_mul(n_sub(ta, ta_old), mk(p.cfg.reward_scaling, FINENV_NT_PY));
    gr = n_add(n_mul(gr, mk(p.cfg.gamma, FINENV_NT_PY)), r);
    const bool done = day == p.cfg.n_days - 1;
    if (done) {
        r = gr;
        if (valid) NF(FINENV_NF_EPISODE_RETURN) = n_div(ta, ita).v;
    }
    if (valid) {
        *at(p.reward, (unsigned)e) = (float)r.v;
        *at(p.done, (unsigned)e) = done ? 1 : 0;
        NF(FINENV_NF_LAST_REWARD) = r.v;
    }
    NSTAMP(3);
    wave_sync();
    if (!head_filled) fill_head(amount);     // overwrites the (consumed) action rows
    wave_sync();
    const unsigned long long valid_mask = __ballot(valid);
    const unsigned long long done_mask = __ballot(done && valid);
    int row_day = day;
    if (done_mask != 0ull) {
        if (p.term_obs != nullptr)
            np_write_rows<true>(p.term_obs, p, e0, nenv_w, day, done_mask, heads, lane, 0, kpatch);
        if (p.auto_reset) {
            wave_sync();
            if (done) {
                do_reset(amount, ta, gr, ita);
                row_day = 0;
                fill_head(amount);
            }
            wave_sync();
        }
    }
    NSTAMP(4);
    // ---- hand-off (pairs with the streamers' second barrier): heads and books are final in LDS.  With
    // every env on the same row and nobody done, the streamer -- long finished -- writes the upper half of
    // the head rows and of the books; otherwise this wave writes everything.
    const bool share = quad_ok && done_mask == 0ull && __all(!valid || row_day == day0);
    lds_barrier();                                                            // #2
    if (!NDIAG(2)) {
        if (share) head_quads(head_q, 0, kWave / 2);
        else np_write_rows(p.obs, p, e0, nenv_w, row_day, valid_mask, heads, lane, 0, kpatch);
    }
    NSTAMP(5);
    if (valid) store_state(amount, ta, gr, ita, r.tag, row_day, 0, share ? N / 2 : N);
    NSTAMP(6);
}

}  // namespace

struct finenv_stocknp {
    int device;           // HIP device that owns the bound state block (-1 before bind)
    finenv_stocknp_config cfg;
    finenv_stocknp_panel panel;
    finenv_stocknp_state st;
    int bound;
    int D;
    int obs_pitch;        // row pitch of the obs buffers handed to step / reset (floats)
    uint32_t magicN;
    char err[256];
};

namespace {
int np_fail(finenv_stocknp *h, int code, const char *msg)
{
    if (h) snprintf(h->err, sizeof(h->err), "%s", msg);
    return code;
}
int np_check(finenv_stocknp *h, const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        return FINENV_ERR_HIP;
    }
    return FINENV_OK;
}
NpParams np_params(const finenv_stocknp *h)
{
    NpParams p;
    memset(&p, 0, sizeof(p));
    p.cfg = h->cfg;
    p.panel = h->panel;
    p.st = h->st;
    p.D = h->D;
    p.obs_pitch = h->obs_pitch;
    p.magicN = h->magicN;
    return p;
}
dim3 np_grid(int E)
{
    const int waves = (E + kWave - 1) / kWave;
    return dim3((unsigned)((waves + kWaves - 1) / kWaves));
}
}  // namespace

extern "C" {

int finenv_stocknp_create(const finenv_stocknp_config *cfg, finenv_stocknp **out)
{
    if (!cfg || !out) return FINENV_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_envs < 1 || cfg->n_tickers < 1 || cfg->n_tickers > FINENV_STOCKNP_MAX_TICKERS ||
        cfg->n_techw < 0 || cfg->n_days < 2 || cfg->min_action < 0 || cfg->max_stock <= 0 ||
        cfg->max_stock > 1e6)
        return FINENV_ERR_INVALID;
    const long long E = cfg->n_envs, N = cfg->n_tickers, T = cfg->n_days;
    const long long D = 3 + 3 * N + cfg->n_techw, lim = (1ll << 32) - 1;
    if (E * 8 * FINENV_STOCKNP_F64_FIELDS > lim || E * N * 12 > lim || T * D * 4 > lim ||
        64 * D * 4 > lim)
        return FINENV_ERR_INVALID;
    finenv_stocknp *h = new (std::nothrow) finenv_stocknp;
    if (!h) return FINENV_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->device = -1;
    h->cfg = *cfg;
    h->D = (int)D;
    h->obs_pitch = (int)D;
    h->magicN = N >= 2 ? (uint32_t)(((1ull << 32) + N - 1) / (unsigned long long)N) : 0u;
    *out = h;
    return FINENV_OK;
}

void finenv_stocknp_destroy(finenv_stocknp *h) { delete h; }
const char *finenv_stocknp_last_error(const finenv_stocknp *h) { return h ? h->err : "null handle"; }
int finenv_stocknp_obs_dim(const finenv_stocknp *h) { return h ? h->D : FINENV_ERR_INVALID; }

int finenv_stocknp_set_obs_pitch(finenv_stocknp *h, int32_t pitch)
{
    if (!h) return FINENV_ERR_INVALID;
    if (pitch == 0) pitch = h->D;
    if (pitch < h->D || (long long)pitch * 64 * 4 > (1ll << 32) - 1)
        return np_fail(h, FINENV_ERR_INVALID, "set_obs_pitch: pitch must be >= obs_dim");
    h->obs_pitch = pitch;
    return FINENV_OK;
}

int finenv_stocknp_bind(finenv_stocknp *h, const finenv_stocknp_panel *panel,
                        const finenv_stocknp_state *st)
{
    if (!h || !panel || !st) return FINENV_ERR_INVALID;
    if (!panel->price || !panel->obs_tmpl || !panel->turb_bool || !st->f64 || !st->i32 || !st->f32)
        return np_fail(h, FINENV_ERR_INVALID, "bind: null pointer");
    h->panel = *panel;
    h->st = *st;
    h->device = finenv_host::pointer_device(st->f64);
    h->bound = 1;
    return FINENV_OK;
}

int finenv_stocknp_reset(finenv_stocknp *h, const uint8_t *mask, float *obs_out, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return np_fail(h, FINENV_ERR_UNBOUND, "reset: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    NpParams p = np_params(h);
    p.mask = mask;
    p.obs = obs_out;
    hipLaunchKernelGGL((stocknp_kernel<true>), np_grid(h->cfg.n_envs), dim3(kWave * kWaves), 0,
                       (hipStream_t)stream, p);
    return np_check(h, "stocknp_reset");
}

int finenv_stocknp_step(finenv_stocknp *h, const float *actions, float *obs, float *reward,
                        uint8_t *done, float *term_obs, int32_t auto_reset, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return np_fail(h, FINENV_ERR_UNBOUND, "step: bind first");
    const finenv_host::DeviceGuard guard(h->device);
    if (!actions || !obs || !reward || !done)
        return np_fail(h, FINENV_ERR_INVALID, "step: null actions/obs/reward/done");
    NpParams p = np_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.auto_reset = auto_reset;
#ifdef FINENV_DIAG
    p.dbg = g_finenv_dbg;
    {
        const char *d = getenv("FINENV_DIAG");
        p.diag = d ? atoi(d) : 0;
    }
#endif
    hipLaunchKernelGGL((stocknp_kernel<false>), np_grid(h->cfg.n_envs), dim3(kWave * kWaves * 2), 0,
                       (hipStream_t)stream, p);
    return np_check(h, "stocknp_step");
}

}  // extern "C"
