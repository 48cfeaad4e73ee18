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


}  // namespace
