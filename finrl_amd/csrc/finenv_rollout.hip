// finenv_rollout.hip -- rollout-buffer helpers over [n_steps][E] tensors (gfx950): the GAE
// time-reverse scan and the one-launch store of a policy's outputs into slice t.
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

// actions [E*A], values [E], log-probs [E] -> slice t of the rollout tensors, one launch (three
// separate copy launches cost ~2 us each at 32,768 envs: more than the env step they sit beside)
template <typename V>
__global__ void __launch_bounds__(256) rollout_put_kernel(const V *__restrict__ a_src, V *__restrict__ a_dst,
                                                          const V *__restrict__ v_src, V *__restrict__ v_dst,
                                                          const V *__restrict__ l_src, V *__restrict__ l_dst,
                                                          int na, int nv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) {
        a_dst[i] = a_src[i];
    } else if (i < na + nv) {
        v_dst[i - na] = v_src[i - na];
    } else if (i < na + 2 * nv) {
        l_dst[i - na - nv] = l_src[i - na - nv];
    }
}
}  // namespace

extern "C" int finenv_rollout_put(const float *actions, const float *values, const float *log_probs,
                                  float *actions_out, float *values_out, float *log_probs_out,
                                  int32_t n_envs, int32_t action_dim, void *stream)
{
    if (!actions || !values || !log_probs || !actions_out || !values_out || !log_probs_out ||
        n_envs < 1 || action_dim < 1 || (long long)n_envs * action_dim > (1ll << 30))
        return FINENV_ERR_INVALID;
    const finenv_host::DeviceGuard guard(finenv_host::pointer_device(actions_out));
    const int na = n_envs * action_dim, nv = n_envs;
    const uintptr_t bits = (uintptr_t)actions | (uintptr_t)values | (uintptr_t)log_probs |
                           (uintptr_t)actions_out | (uintptr_t)values_out | (uintptr_t)log_probs_out;
    if ((bits & 15) == 0 && (na & 3) == 0 && (nv & 3) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int n4 = (na + 2 * nv) / 4;
        hipLaunchKernelGGL(rollout_put_kernel<f4>, dim3((n4 + 255) / 256), dim3(256), 0,
                           (hipStream_t)stream, (const f4 *)actions, (f4 *)actions_out,
                           (const f4 *)values, (f4 *)values_out, (const f4 *)log_probs,
                           (f4 *)log_probs_out, na / 4, nv / 4);
    } else {
        const int n = na + 2 * nv;
        hipLaunchKernelGGL(rollout_put_kernel<float>, dim3((n + 255) / 256), dim3(256), 0,
                           (hipStream_t)stream, actions, actions_out, values, values_out, log_probs,
                           log_probs_out, na, nv);
    }
    return hipGetLastError() == hipSuccess ? FINENV_OK : FINENV_ERR_HIP;
}

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
