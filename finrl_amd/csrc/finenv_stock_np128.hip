// finenv_stock_np128.hip -- step / aux kernels of the batched StockTradingEnv for N <= 128 tickers
// (finenv_stock_kernels.inc compiled with FINENV_NP = 128); design notes: finenv_stock.hip.
#include "finenv_stock_common.h"

namespace {
namespace np128 {
#define FINENV_NP 128
#define FINENV_LOG2NP 7
#define FINENV_SORTNET "sortnet128.inc"
#include "finenv_stock_kernels.inc"
#include "finenv_stock_wide.inc"
#undef FINENV_NP
#undef FINENV_LOG2NP
#undef FINENV_SORTNET
}  // namespace np128

// NASDAQ-100 shape (BASELINE configs[3]): compile-time ticker count, 40.7 KB of LDS, 4 blocks per CU
template <bool TURB, bool STATS, int NT>
int launch_step_wide(const Params &p, hipStream_t stream)
{
    const dim3 block(kStepThreads);
    launch_rounds(p, &np128::stock_step_wide_kernel<TURB, STATS, NT>, np128::WideGeom<NT>::kBytes,
                  [&](const Params &q, int nb) {
                      hipLaunchKernelGGL((np128::stock_step_wide_kernel<TURB, STATS, NT>), dim3((unsigned)nb), block,
                                         np128::WideGeom<NT>::kBytes, stream, q);
                  });
    return 0;
}

template <bool TURB, bool STATS>
int launch_step(const Params &p, int device, hipStream_t stream)
{
    if (p.cfg.n_tickers == 100 && p.cfg.hmax <= np128::WideGeom<100>::kMaxHmax)
        return launch_step_wide<TURB, STATS, 100>(p, stream);
    // one 128-thread block per 64 envs, dynamic LDS = kLdsStep
    const dim3 block(kStepThreads);
    constexpr size_t lds = sizeof(float) * np128::kLdsStep;
    // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device (a process may hold
    // handles on several GPUs)
    if (lds > 64 * 1024) {
        static unsigned long long attr_set_mask = 0ull;
        const int dev = device >= 0 && device < 64 ? device : 0;
        if (!((attr_set_mask >> dev) & 1ull)) {
            if (hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&np128::stock_step_kernel<TURB, STATS>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return -1;
            attr_set_mask |= 1ull << dev;
        }
    }
    launch_rounds(p, &np128::stock_step_kernel<TURB, STATS>, lds, [&](const Params &q, int nb) {
        hipLaunchKernelGGL((np128::stock_step_kernel<TURB, STATS>), dim3((unsigned)nb), block, lds, stream, q);
    });
    return 0;
}
}  // namespace

namespace finenv_stock_impl {

int launch_step_np128(const Params &p, bool turb, bool stats, int device, hipStream_t stream)
{
    if (turb && stats) return launch_step<true, true>(p, device, stream);
    if (turb) return launch_step<true, false>(p, device, stream);
    if (stats) return launch_step<false, true>(p, device, stream);
    return launch_step<false, false>(p, device, stream);
}

void launch_aux_np128(const Params &p, int mode, hipStream_t stream)
{
    const int waves = (p.cfg.n_envs + kWave - 1) / kWave;
    const dim3 grid((unsigned)((waves + np128::kAuxWaves - 1) / np128::kAuxWaves));
    hipLaunchKernelGGL(np128::stock_aux_kernel, grid, dim3(kWave * np128::kAuxWaves), 0, stream, p, mode);
}

}  // namespace finenv_stock_impl
