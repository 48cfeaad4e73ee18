// finenv_stock_common.h -- shared by the translation units of the batched StockTradingEnv:
// finenv_stock.hip (C ABI, dispatch, terminal-summary kernel) and finenv_stock_np{32,64,128}.hip
// (the step / aux kernels compiled per padded ticker count, one file each so `make -j` builds
// them side by side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <type_traits>

#include "finenv.h"
#include "finenv_dev.h"
#include "finenv_host.h"

namespace finenv_stock_impl {
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
    int32_t obs_pitch;    // row pitch of `obs` in floats (>= D; D = packed rows)
    int32_t desync_hint;  // envs may sit on different days (selects the kernel instantiation only)
    int32_t day0;
    uint32_t magicN;      // ceil(2^32 / N) for N >= 2 (exact f / N for f < 2^16)
    int32_t block_base;   // first 64-env group of this launch (batches larger than one resident round of
                          // blocks are stepped as several launches: launch_rounds below)
    int32_t diag;         // FINENV_DIAG builds only: phase-skip bitmask (timing experiments)
    unsigned long long *dbg;   // FINENV_DIAG builds only: [block][role][16] s_memrealtime stamps
};
}  // namespace finenv_stock_impl

namespace {

constexpr int kWave = 64;
constexpr int kStepThreads = 2 * kWave;
#ifndef FINENV_TRADE_UNROLL
#define FINENV_TRADE_UNROLL 2     // unroll factor of the rolled sell / buy loops (tuning switch)
#endif

using finenv_stock_impl::Params;

// The step kernels are built for ONE resident round of blocks (every block in the same phase: traders
// loading while nobody stores yet, streamers streaming while traders compute).  Handed more blocks than
// fit at once, the hardware refills slots as blocks finish, phases mix -- traders' loads queue behind
// other blocks' stores -- and the per-env cost rises by a third (DOW30: 65,536 envs 0.64 of the
// roofline, 262,144 envs 0.47; profiles/r03_placement.md).  So a large batch is stepped as
// ceil(blocks / round) launches of equal size on the caller's stream, each one resident round or less.
// `fn(q, nblocks)` launches nblocks blocks with q.block_base set.
template <typename Kernel, typename Launch>
inline void launch_rounds(const Params &p, Kernel kernel, size_t lds_bytes, Launch fn)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
               prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    static int per_cu = 0;                       // (one static per kernel instantiation)
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kernel), kStepThreads,
                                                         lds_bytes) != hipSuccess || nb < 1)
            nb = 1;
        per_cu = nb;
    }
    const int blocks = (p.cfg.n_envs + kWave - 1) / kWave;
    const int round = per_cu * cus;
    Params q = p;
    if (blocks <= round) {
        q.block_base = 0;
        fn(q, blocks);
        return;
    }
    const int k = (blocks + round - 1) / round, chunk = (blocks + k - 1) / k;
    for (int b = 0; b < blocks; b += chunk) {
        q.block_base = b;
        fn(q, blocks - b < chunk ? blocks - b : chunk);
    }
}

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

}  // namespace

// Launchers exported by the per-width translation units (C++ linkage, library-internal).
namespace finenv_stock_impl {
// step(): returns 0, or -1 when the dynamic-LDS limit could not be raised
int launch_step_np32(const Params &p, bool turb, bool stats, int device, hipStream_t stream);
int launch_step_np64(const Params &p, bool turb, bool stats, int device, hipStream_t stream);
int launch_step_np128(const Params &p, bool turb, bool stats, int device, hipStream_t stream);
// init / reset / observe (mode 0 / 1 / 2)
void launch_aux_np32(const Params &p, int mode, hipStream_t stream);
void launch_aux_np64(const Params &p, int mode, hipStream_t stream);
void launch_aux_np128(const Params &p, int mode, hipStream_t stream);
}  // namespace finenv_stock_impl
