// finenv_stock_np64.hip -- step / aux kernels of the batched StockTradingEnv for N <= 64 tickers
// (finenv_stock_kernels.inc compiled with FINENV_NP = 64); design notes: finenv_stock.hip.
#include "finenv_stock_common.h"

namespace {
namespace np64 {
#define FINENV_NP 64
#define FINENV_LOG2NP 6
#define FINENV_SORTNET "sortnet64.inc"
#include "finenv_stock_kernels.inc"
#undef FINENV_NP
#undef FINENV_LOG2NP
#undef FINENV_SORTNET
}  // namespace np64

template <bool TURB, bool STATS>
int launch_step(const Params &p, int device, hipStream_t stream)
{
    // one 128-thread block per 64 envs, dynamic LDS = kLdsStep
    const dim3 block(kStepThreads);
    constexpr size_t lds = sizeof(float) * np64::kLdsStep;
    (void)device;
    launch_rounds(p, &np64::stock_step_kernel<TURB, STATS>, lds, [&](const Params &q, int nb) {
        hipLaunchKernelGGL((np64::stock_step_kernel<TURB, STATS>), dim3((unsigned)nb), block, lds, stream, q);
    });
    return 0;
}
}  // namespace

namespace finenv_stock_impl {

int launch_step_np64(const Params &p, bool turb, bool stats, int device, hipStream_t stream)
{
    if (turb && stats) return launch_step<true, true>(p, device, stream);
    if (turb) return launch_step<true, false>(p, device, stream);
    if (stats) return launch_step<false, true>(p, device, stream);
    return launch_step<false, false>(p, device, stream);
}

void launch_aux_np64(const Params &p, int mode, hipStream_t stream)
{
    const int waves = (p.cfg.n_envs + kWave - 1) / kWave;
    const dim3 grid((unsigned)((waves + np64::kAuxWaves - 1) / np64::kAuxWaves));
    hipLaunchKernelGGL(np64::stock_aux_kernel, grid, dim3(kWave * np64::kAuxWaves), 0, stream, p, mode);
}

}  // namespace finenv_stock_impl
