// finenv_dev.h -- device helpers shared by the finenv kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace {

constexpr int kWaveSize = 64;

// base (uniform, SGPR pair) + 32-bit per-lane BYTE offset: lets hipcc use the
// `global_load/store v, v_off, s[base]` addressing form instead of keeping a 64-bit VGPR
// address per array alive across the kernel.  Hosts validate that every offset fits 32 bits.
template <typename T>
__device__ __forceinline__ T *at(T *base, unsigned idx)
{
    return reinterpret_cast<T *>(
        reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(base)) +
        (size_t)(idx * (unsigned)sizeof(T)));
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}



// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic (lgkmcnt) but NOT
// for its outstanding global stores.  __syncthreads() also emits s_waitcnt vmcnt(0), which makes
// a wave that is streaming stores sit at the barrier until every store has completed (~2 us
// under load) -- the streamer must keep its stores in flight across the hand-off barriers.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Counter-based uniform draw in [0, hi): splitmix64 of (seed, env, episode), multiply-shift.
__device__ __forceinline__ int draw_start(unsigned long long seed, int env, int episode, int hi)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(unsigned)env +
                           0xD1B54A32D192ED03ull * (unsigned long long)(unsigned)episode;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (int)(((z >> 32) * (unsigned long long)(unsigned)hi) >> 32);
}

// Keep a just-loaded value in a register HERE.  hipcc sinks a load into the (conditional) block
// that holds its only use, which turns "issue a batch of loads, then consume them" back into
// load / s_waitcnt vmcnt(0) / use, one exposed HBM round trip per element.
template <typename T>
__device__ __forceinline__ void pin(T &x)
{
    asm volatile("" : "+v"(x));
}

// Transpose a [nenv_w][N] f32 action tile (row-major in HBM, what SB3 / ElegantRL hand over) into
// LDS rows of `stride` dwords: rows[el * stride + i] = act[el][i].  kBatch coalesced loads are
// issued before the first LDS write; a plain rolled loop (load, index math, ds_write) exposes one
// HBM round trip per 64 floats at one wave per SIMD.  magicN = ceil(2^32 / N) (N >= 2).
template <int kBatch = 32>
__device__ __forceinline__ void stage_action_tile(float *rows, int stride,
                                                  const float *__restrict__ src, int nenv_w, int N,
                                                  unsigned magicN, int lane)
{
    const int total = nenv_w * N;
    for (int f0 = 0; f0 < total; f0 += kBatch * kWaveSize) {
        float v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int f = f0 + j * kWaveSize + lane;
            v[j] = *at(src, (unsigned)(f < total ? f : total - 1));
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) pin(v[j]);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int f = f0 + j * kWaveSize + lane;
            if (f < total) {
                const int el = (N == 1) ? f : (int)__umulhi((unsigned)f, magicN);
                rows[el * stride + (f - el * N)] = v[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Observation-row writer shared by the sibling env kernels (one wave, lane = column within a
// 64-column chunk).  For each env el selected by lane_mask it writes D floats to dst[el * D ..]:
//   column col comes from heads[el * head_stride + patch_sel(col)] when patch_sel(col) >= 0
//   (per-env values published in LDS), else from tmpl[tmpl_index(day_el, col)] (market data;
//   tmpl_index < 0 = no source, writes 0).
// Memory-op order is what matters here (vmcnt is in-order on gfx950: a load placed between
// stores waits until every older store is acknowledged):
//   * all selected envs on the same panel row (lock-step batches): the template values of up to
//     kMaxChunks chunks are loaded ONCE, then the env loop holds nothing but stores, in
//     row-major order (neighbouring chunks of a row back to back: L2 merges the 64-B segments
//     that a row of odd length straddles);
//   * per-env rows (random starts): kBatch rows' loads are issued before the batch's stores.
// ---------------------------------------------------------------------------------------------
//   * kCompact: plain rolled loops (small code) for paths that run once per episode -- the env
//     kernels must stay inside the 64 KB instruction cache two CUs share: at 79 KB the
//     cash-penalty step kernel ran 2x slower than at 57 KB with strictly less work per wave.
template <int kMaxChunks, int kBatch, bool kCompact = false, typename TmplIndex, typename PatchSel>
__device__ __forceinline__ void write_obs_rows_generic(float *__restrict__ dst,
                                                       const float *__restrict__ tmpl, int D,
                                                       int e0, int nenv_w, int row_day,
                                                       unsigned long long lane_mask,
                                                       const float *heads, int head_stride,
                                                       int lane, TmplIndex tmpl_index,
                                                       PatchSel patch_sel, int k_lo = 0,
                                                       int k_hi = 1 << 30, int pitch = 0)
{
    // [k_lo, k_hi): chunk range to write (a streamer wave takes the pure market-data chunks)
    // pitch: row pitch of dst in floats (0 = packed rows of D floats)
    if (lane_mask == 0ull) return;
    const int P = pitch > 0 ? pitch : D;
    const int first = __builtin_ctzll(lane_mask);
    const int d0 = __builtin_amdgcn_readlane(row_day, first);
    const bool mine = (lane_mask >> lane) & 1ull;
    const bool uniform = __all(!mine || row_day == d0);
    float *const base = dst + (size_t)e0 * P;
    const int nchunk = min(k_hi, (D + kWaveSize - 1) / kWaveSize);
    if (k_lo >= nchunk) return;

    if constexpr (kCompact) {
        for (int k = k_lo; k < nchunk; ++k) {
            const int col = k * kWaveSize + lane;
            const bool in = col < D;
            const int s = in ? patch_sel(col) : -1;
            for (int el = 0; el < nenv_w; ++el) {
                if (!((lane_mask >> el) & 1ull)) continue;
                const int de = __builtin_amdgcn_readlane(row_day, el);
                const int idx = (in && s < 0) ? tmpl_index(de, col) : -1;
                float v = 0.0f;
                if (tmpl != nullptr && idx >= 0) v = *at(tmpl, (unsigned)idx);
                if (s >= 0) v = heads[el * head_stride + s];
                if (in) *at(base, (unsigned)(el * P + col)) = v;
            }
        }
        return;
    }

    if (uniform && nchunk - k_lo <= kMaxChunks) {
        float t[kMaxChunks];
        int sel[kMaxChunks];
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            const int col = (k_lo + k) * kWaveSize + lane;
            const bool in = k_lo + k < nchunk && col < D;
            sel[k] = in ? patch_sel(col) : -1;
            const int idx = (in && sel[k] < 0) ? tmpl_index(d0, col) : -1;
            t[k] = 0.0f;
            if (tmpl != nullptr) {
                const float x = *at(tmpl, (unsigned)(idx >= 0 ? idx : 0));
                t[k] = idx >= 0 ? x : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) pin(t[k]);
        // which chunks carry per-env columns at all (wave-uniform)
        bool anyp[kMaxChunks];
        bool some = false;
#pragma unroll
        for (int k = 0; k < kMaxChunks; ++k) {
            anyp[k] = __any(sel[k] >= 0);
            some = some || anyp[k];
        }
        if (!some) {                         // pure market-data chunks: stores only
            for (int el = 0; el < nenv_w; ++el) {
                if (!((lane_mask >> el) & 1ull)) continue;
#pragma unroll
                for (int k = 0; k < kMaxChunks; ++k) {
                    if (k_lo + k >= nchunk) break;
                    const int col = (k_lo + k) * kWaveSize + lane;
                    if (col < D) *at(base, (unsigned)(el * P + col)) = t[k];
                }
            }
            return;
        }
        // batches of kRB rows: the batch's LDS reads are all in flight before its first store (one
        // row at a time exposes an LDS round trip per row and chunk: 12 us for 64 rows x 2 chunks
        // in the array-state env, 6 us in the crypto env)
        constexpr int kRB = kMaxChunks >= 8 ? 4 : 8;     // (32 reads in flight; code size)
        for (int g = 0; g < nenv_w; g += kRB) {
            float hv[kRB][kMaxChunks];
#pragma unroll
            for (int j = 0; j < kRB; ++j) {
                const int el = min(g + j, kWaveSize - 1);
#pragma unroll
                for (int k = 0; k < kMaxChunks; ++k)
                    if (anyp[k]) hv[j][k] = heads[el * head_stride + (sel[k] >= 0 ? sel[k] : 0)];
            }
#pragma unroll
            for (int j = 0; j < kRB; ++j) {
                const int el = g + j;
                if (el >= nenv_w || !((lane_mask >> el) & 1ull)) continue;
#pragma unroll
                for (int k = 0; k < kMaxChunks; ++k) {
                    if (k_lo + k >= nchunk) break;
                    const int col = (k_lo + k) * kWaveSize + lane;
                    const float v = (anyp[k] && sel[k] >= 0) ? hv[j][k] : t[k];
                    if (col < D) *at(base, (unsigned)(el * P + col)) = v;
                }
            }
        }
        return;
    }

    for (int k = k_lo; k < nchunk; ++k) {
        const int col = k * kWaveSize + lane;
        const bool in = col < D;
        const int s = in ? patch_sel(col) : -1;
        const bool any_patch = __any(s >= 0);
        for (int g = 0; g < nenv_w; g += kBatch) {
            float t[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int de = __builtin_amdgcn_readlane(row_day, min(g + j, nenv_w - 1));
                const int idx = (in && s < 0) ? tmpl_index(de, col) : -1;
                t[j] = 0.0f;
                if (tmpl != nullptr) {
                    const float x = *at(tmpl, (unsigned)(idx >= 0 ? idx : 0));
                    t[j] = idx >= 0 ? x : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) pin(t[j]);
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int el = g + j;
                if (el >= nenv_w || !((lane_mask >> el) & 1ull)) continue;
                float v = t[j];
                if (any_patch) {
                    const float hv = heads[el * head_stride + (s >= 0 ? s : 0)];
                    v = s >= 0 ? hv : v;
                }
                if (in) *at(base, (unsigned)(el * P + col)) = v;
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// The observation chunks that hold cash / holdings columns ("head" chunks, k < KP), lock-step days:
// written after the hand-off barrier, half the rows by each wave.  Their template (market data)
// values must NOT be loaded from global memory at that point: a load issued after stores waits until
// every older store of the wave is acknowledged (vmcnt retires in order; 2-4 us for a streamer with
// a few hundred stores in flight), pinning or `volatile` loads at the start of the streamer cost a
// round trip before its first store (measured: +0.7 / +2.3 us).  So the wave that waits for global
// data anyway at the start of the step (it stages the price row) also fetches these KP values per
// lane and parks them in LDS (head_stage); after the barrier both waves read them back (head_plan).
//   widx(col)     -> index of the per-env value of that column in the caller's LDS image, or -1
//                    (market data: the template value is written)
//   val(el, w)    -> that value of env el as f32
// -------------------------------------------------------------------------------------
template <int KP>
struct HeadPlan {
    float t[KP];
    int w[KP];
    bool in[KP];
    bool uniform;            // every selected env reads the same panel row (else: generic writer)
};

// issue the KP template loads (values for the caller to hold until it calls head_park)
template <int KP>
__device__ __forceinline__ void head_fetch(float (&tt)[KP], const float *__restrict__ tmpl, int D,
                                           int row_first, int lane, int kp)
{
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int col = k * kWaveSize + lane;
        tt[k] = *at(tmpl, (unsigned)(row_first * D + ((k < kp && col < D) ? col : 0)));
    }
}

template <int KP>
__device__ __forceinline__ void head_park(float *ldst, const float (&tt)[KP], int lane)
{
#pragma unroll
    for (int k = 0; k < KP; ++k) ldst[k * kWaveSize + lane] = tt[k];
}

template <int KP, typename WIdx>
__device__ __forceinline__ void head_plan(HeadPlan<KP> &hp, const float *ldst, int D, int row_day,
                                          int row_first, unsigned long long lane_mask, int lane,
                                          int kp, WIdx widx)
{
    const bool mine = (lane_mask >> lane) & 1ull;
    hp.uniform = __all(!mine || row_day == row_first);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int col = k * kWaveSize + lane;
        hp.in[k] = k < kp && col < D;
        hp.w[k] = hp.in[k] ? widx(col) : -1;
        hp.t[k] = ldst[k * kWaveSize + lane];
    }
}

template <int KP, typename Val>
__device__ __forceinline__ void head_store(const HeadPlan<KP> &hp, float *__restrict__ dst, int D,
                                           int P, int e0, int nenv_w, unsigned long long lane_mask,
                                           int lane, int el_lo, int el_hi, Val val)
{
    (void)D;                                  // P: row pitch of dst in floats (>= D)
    float *const base = dst + (size_t)e0 * P;
    const int hi = min(nenv_w, el_hi);
    // batches of kB rows: the batch's LDS reads are all in flight before its first store (one row
    // at a time exposes an LDS round trip per row: ~170 cycles x 64 rows at the tail of every block)
    constexpr int kB = KP <= 2 ? 8 : 4;
    for (int g = el_lo; g < hi; g += kB) {
        float pv[kB][KP];
#pragma unroll
        for (int j = 0; j < kB; ++j)
#pragma unroll
            for (int k = 0; k < KP; ++k) pv[j][k] = val(min(g + j, kWaveSize - 1), hp.w[k] >= 0 ? hp.w[k] : 0);
#pragma unroll
        for (int j = 0; j < kB; ++j) {
            const int el = g + j;
            if (el >= hi || !((lane_mask >> el) & 1ull)) continue;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float v = hp.w[k] >= 0 ? pv[j][k] : hp.t[k];
                if (hp.in[k]) *at(base, (unsigned)(el * P + k * kWaveSize + lane)) = v;
            }
        }
    }
}


}  // namespace
