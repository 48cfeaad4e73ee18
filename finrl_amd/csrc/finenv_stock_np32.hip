// finenv_stock_np32.hip -- step / aux kernels of the batched StockTradingEnv for N <= 32 tickers
// (finenv_stock_kernels.inc compiled with FINENV_NP = 32); design notes: finenv_stock.hip.
#include "finenv_stock_common.h"

namespace {
namespace np32 {
#define FINENV_NP 32
#define FINENV_LOG2NP 5
#define FINENV_SORTNET "sortnet32.inc"
#include "finenv_stock_kernels.inc"
#undef FINENV_NP
#undef FINENV_LOG2NP
#undef FINENV_SORTNET
}  // namespace np32

template <bool TURB, bool STATS>
int launch_step(const Params &p, int device, hipStream_t stream)
{
    // one 128-thread block per 64 envs, dynamic LDS = kLdsStep; one resident round of blocks per launch
    const dim3 block(kStepThreads);
    constexpr size_t lds = sizeof(float) * np32::kLdsStep;
    (void)device;
    if (p.desync_hint)      // envs may sit on different days: the instantiation with the per-env fast paths
        launch_rounds(p, &np32::stock_step_kernel<TURB, STATS, true>, lds, [&](const Params &q, int nb) {
            hipLaunchKernelGGL((np32::stock_step_kernel<TURB, STATS, true>), dim3((unsigned)nb), block, lds, stream, q);
        });
    else
        launch_rounds(p, &np32::stock_step_kernel<TURB, STATS, false>, lds, [&](const Params &q, int nb) {
            hipLaunchKernelGGL((np32::stock_step_kernel<TURB, STATS, false>), dim3((unsigned)nb), block, lds, stream, q);
        });
    return 0;
}
}  // namespace

namespace finenv_stock_impl {

int launch_step_np32(const Params &p, bool turb, bool stats, int device, hipStream_t stream)
{
    if (turb && stats) return launch_step<true, true>(p, device, stream);
    if (turb) return launch_step<true, false>(p, device, stream);
    if (stats) return launch_step<false, true>(p, device, stream);
    return launch_step<false, false>(p, device, stream);
}

void launch_aux_np32(const Params &p, int mode, hipStream_t stream)
{
    const int waves = (p.cfg.n_envs + kWave - 1) / kWave;
    const dim3 grid((unsigned)((waves + np32::kAuxWaves - 1) / np32::kAuxWaves));
    hipLaunchKernelGGL(np32::stock_aux_kernel, grid, dim3(kWave * np32::kAuxWaves), 0, stream, p, mode);
}

}  // namespace finenv_stock_impl
