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
