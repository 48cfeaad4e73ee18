// finenv_rollout.hip -- GAE time-reverse scan over [n_steps][E] rollout tensors (gfx950).
// One lane per env, fully coalesced row reads; float32 with the operation order of SB3's
// documented RolloutBuffer.compute_returns_and_advantage (see include/finenv.h).  HBM-bound:
// 17 bytes per (step, env).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "finenv.h"
#include "finenv_host.h"
#include "finenv_dev.h"

namespace {
__global__ void __launch_bounds__(256) gae_scan_kernel(const float *__restrict__ rewards,
                                                       const float *__restrict__ values,
                                                       const uint8_t *__restrict__ dones,
                                                       const float *__restrict__ last_values,
                                                       float *__restrict__ adv,
                                                       float *__restrict__ ret, int S, int E,
                                                       float gamma, float lam)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float next_v = last_values[e];
    float gae = 0.0f;
    const float gl = gamma * lam;
#pragma unroll 4
    for (int t = S - 1; t >= 0; --t) {
        const size_t k = (size_t)t * E + e;
        const float nnt = 1.0f - (float)dones[k];
        const float v = values[k];
        const float delta = rewards[k] + gamma * next_v * nnt - v;
        gae = delta + gl * nnt * gae;
        adv[k] = gae;
        ret[k] = gae + v;
        next_v = v;
    }
}
}  // namespace

extern "C" int finenv_gae_scan(const float *rewards, const float *values, const uint8_t *dones,
                               const float *last_values, float *advantages, float *returns,
                               int32_t n_steps, int32_t n_envs, float gamma, float gae_lambda,
                               void *stream)
{
    if (!rewards || !values || !dones || !last_values || !advantages || !returns ||
        n_steps < 1 || n_envs < 1)
        return FINENV_ERR_INVALID;
    const finenv_host::DeviceGuard guard(finenv_host::pointer_device(rewards));
    hipLaunchKernelGGL(gae_scan_kernel, dim3((n_envs + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, rewards, values, dones, last_values, advantages,
                       returns, n_steps, n_envs, gamma, gae_lambda);
    return hipGetLastError() == hipSuccess ? FINENV_OK : FINENV_ERR_HIP;
}
