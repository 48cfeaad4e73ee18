// Microbenchmark: how should a wave write its [64][D] f32 observation block?  (gfx950)
//   A: dword per lane, 256-B chunk stores, row by row (what the streamers did in round 1)
//   B: one UNALIGNED dwordx4 store per row segment (4-B aligned addresses)
//   C: flat 16-B-aligned dwordx4 stores over the whole contiguous block
// One wave per 64 rows, grid = E/64, like the env kernels.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(64) k(float *__restrict__ out, const float *__restrict__ tmpl, int D, int c0)
{
    const int lane = threadIdx.x, e0 = blockIdx.x * 64;
    float *base = out + (size_t)e0 * D;
    if (MODE == 0) {                       // A: cols [c0, D) in 64-float chunks, row-major
        const int nch = (D - c0 + 63) / 64;
        float t[8];
        for (int k2 = 0; k2 < 8; ++k2) { int col = c0 + k2 * 64 + lane; t[k2] = tmpl[col < D ? col : 0]; }
        for (int el = 0; el < 64; ++el)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                int col = c0 + k2 * 64 + lane;
                if (k2 < nch && col < D) base[el * D + col] = t[k2];
            }
    } else if (MODE == 1) {                // B: cols [c0, D) with unaligned x4 stores, row-major
        const int n4 = (D - c0) / 4;       // assume divisible
        const int per = (n4 + 63) / 64;
        f4 t[4];
        for (int s = 0; s < 4; ++s) { int q = s * 64 + lane; int col = c0 + 4 * (q < n4 ? q : 0);
            t[s] = f4{tmpl[col], tmpl[col + 1], tmpl[col + 2], tmpl[col + 3]}; }
        for (int el = 0; el < 64; ++el)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                int q = s * 64 + lane;
                if (s < per && q < n4) *reinterpret_cast<f4 *>(base + el * D + c0 + 4 * q) = t[s];
            }
    } else {                               // C: the whole block flat, aligned x4 (template wraps per row)
        const int n4 = 64 * D / 4;
        for (int q = lane; q < n4; q += 64) {
            int f = 4 * q; int col = f % D;
            f4 v; v.x = tmpl[col]; v.y = tmpl[(col + 1) % D]; v.z = tmpl[(col + 2) % D]; v.w = tmpl[(col + 3) % D];
            reinterpret_cast<f4 *>(base)[q] = v;
        }
    }
}

int main(int argc, char **argv)
{
    const int E = 65536;
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int D = cfg == 0 ? 301 : 901, c0 = cfg == 0 ? 61 : 201;
        float *out, *tmpl;
        hipMalloc(&out, (size_t)E * D * 4); hipMalloc(&tmpl, D * 4 + 64);
        hipMemset(tmpl, 0, D * 4 + 64);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                for (int i = 0; i < 200; ++i) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(E / 64), dim3(64), 0, 0, out, tmpl, D, c0);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(E / 64), dim3(64), 0, 0, out, tmpl, D, c0);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(E / 64), dim3(64), 0, 0, out, tmpl, D, c0);
                }
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double bytes = mode == 2 ? (double)E * D * 4 : (double)E * (D - c0) * 4;
                if (rep == 2) printf("D=%d mode %c: %.2f us/launch, %.2f TB/s (%.1f MB)\n", D, "ABC"[mode], ms * 1000 / 200,
                                     bytes / (ms * 1e-3 / 200) / 1e12, bytes / 1e6);
            }
        }
        hipFree(out); hipFree(tmpl);
    }
    return 0;
}
