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

}  // namespace
