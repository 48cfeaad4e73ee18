// finenv_riskpre.hip -- MI355X (gfx950) kernels + C ABI for the risk precompute steps that feed
// the envs' panels:
//   calculate_turbulence  (finrl/meta/preprocessor/preprocessors.py:215-267)
//   cov_list              (tutorials/2-Advance/FinRL_PortfolioAllocation_Explainable_DRL.py:160-172)
//
// One workgroup per output day.  The day's window of returns (<= 252 x N f64, L2-resident and
// shared by neighbouring days) is reduced to the N x N sample covariance directly into LDS; the
// turbulence path then diagonalises that matrix in place with a parallel cyclic Jacobi sweep
// (round-robin pairing: N/2 disjoint rotations per step), carrying the de-meaned return vector
// through the same rotations, so x' * pinv(C) * x = sum_k y_k^2 / lambda_k over the eigenvalues
// above NumPy's pinv cutoff (1e-15 * lambda_max) -- no eigenvector matrix, no MFMA (N <= 128,
// fp64, ~0.1 GFLOP per day: latency-bound, not a GEMM).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "finenv.h"
#include "finenv_dev.h"
#include "finenv_host.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSweeps = 40;

__global__ void __launch_bounds__(kThreads) returns_kernel(const double *__restrict__ close,
                                                           double *__restrict__ ret, int T, int N)
{
    const long long total = (long long)T * N;
    for (long long f = (long long)blockIdx.x * kThreads + threadIdx.x; f < total;
         f += (long long)gridDim.x * kThreads) {
        // DataFrame.pct_change(): p[t] / p[t-1] - 1, first row NaN (:221)
        ret[f] = f < N ? __builtin_nan("") : close[f] / close[f - N] - 1;
    }
}

struct RiskParams {
    const double *ret;   // [T][N]
    double *cov_out;     // [n_out][N][N] or null
    double *quad_out;    // [T] or null
    int T, N, window, shift, first_day;
};

// LDS: mean[N] | x[N] | A[N][N+1] | cs[2 * M/2] | flag   (cov-only launches use mean[] alone)
template <bool QUAD>
__global__ void __launch_bounds__(kThreads) rolling_risk_kernel(const RiskParams p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int N = p.N, S = N + 1, M = (N + 1) & ~1, H = M / 2;
    double *mean = lds;
    double *x = mean + N;
    double *A = x + N;
    double *cs = A + N * S;
    int *flag = reinterpret_cast<int *>(cs + 2 * H);
    const int tid = threadIdx.x;
    const int d = p.first_day + blockIdx.x;
    // turbulence: rows [d-window, d) minus the leading NaN row (:229-237); cov_list: the
    // `window` returns ending at d inclusive (tutorial :163-165)
    const int lo = max(d - p.window + p.shift, 1), hi = d + p.shift;
    const int n = hi - lo;

    for (int j = tid; j < N; j += kThreads) {                 // np.mean(axis=0): row order
        double s = 0.0;
        for (int t = lo; t < hi; ++t) s += p.ret[(size_t)t * N + j];
        const double m = s / (double)n;
        mean[j] = m;
        if (QUAD) x[j] = p.ret[(size_t)d * N + j] - m;        // :239-241
    }
    __syncthreads();
    const double inv = 1.0 / (double)(n - 1);                 // np.cov: c *= 1 / (n - ddof)
    for (int f = tid; f < N * N; f += kThreads) {
        const int a = f / N, b = f - a * N;
        const double ma = mean[a], mb = mean[b];
        double s = 0.0;
        for (int t = lo; t < hi; ++t)
            s += (p.ret[(size_t)t * N + a] - ma) * (p.ret[(size_t)t * N + b] - mb);
        const double c = s * inv;
        if (QUAD) A[a * S + b] = c;
        if (p.cov_out) p.cov_out[(size_t)blockIdx.x * N * N + f] = c;
    }
    if (!QUAD) return;
    __syncthreads();

    // ---- fast path: Cholesky.  With a well-conditioned covariance (every pivot above 1e-6 of the
    // largest diagonal entry: condition number <~ 1e6, relative error of the quadratic form
    // <~ 1e-10) pinv(C) IS the inverse and x' C^-1 x = |L^-1 x|^2 with C = L L'.  N^3/3 flops
    // instead of the ~40 N^3 of the Jacobi sweeps below (148 ms -> a few ms at N = 100 over 2,893
    // days); anything ill-conditioned (e.g. a duplicated ticker: rank deficient, where pinv drops
    // the null space) falls through to the Jacobi diagonalisation, which applies NumPy's cutoff.
    // The factor is built in a COPY below the diagonal bookkeeping: A's upper triangle (incl. the
    // diagonal, saved in mean[]) stays intact for the fallback.
    {
        double dmax = 0.0;
        for (int k = 0; k < N; ++k) dmax = fmax(dmax, A[k * S + k]);
        const double tol = 1e-6 * dmax;
        if (tid == 0) *flag = 0;
        for (int j = tid; j < N; j += kThreads) mean[j] = A[j * S + j];     // keep the diagonal
        __syncthreads();
        bool ill = !(dmax > 0.0);
        for (int k = 0; k < N && !ill; ++k) {
            // (every thread reads the pivot after the previous step's barrier)
            const double dk = A[k * S + k];
            if (!(dk > tol)) { ill = true; break; }                          // uniform: same dk
            const double lkk = sqrt(dk);
            __syncthreads();                                                 // all have read dk
            for (int i = k + 1 + tid; i < N; i += kThreads) A[i * S + k] /= lkk;   // column k of L
            if (tid == 0) {
                A[k * S + k] = lkk;
                x[k] = x[k] / lkk;                                           // forward solve: y_k
            }
            __syncthreads();
            // trailing update of the lower triangle and of the right-hand side
            const double yk = x[k];
            for (int i = k + 1 + (tid >> 4); i < N; i += kThreads >> 4) {
                const double lik = A[i * S + k];
                for (int j = k + 1 + (tid & 15); j <= i; j += 16) A[i * S + j] -= lik * A[j * S + k];
                if ((tid & 15) == 0) x[i] -= lik * yk;
            }
            __syncthreads();
        }
        if (!ill) {
            if (tid == 0) {
                double q = 0.0;
                for (int k = 0; k < N; ++k) q += x[k] * x[k];
                p.quad_out[d] = q;                                           // temp, :244-246
            }
            return;
        }
        // fallback: restore the symmetric matrix (lower triangle + diagonal from the intact upper
        // triangle / saved diagonal) and the de-meaned return, then diagonalise
        __syncthreads();
        for (int f = tid; f < N * N; f += kThreads) {
            const int a = f / N, b = f - a * N;
            if (b < a) A[a * S + b] = A[b * S + a];
            else if (a == b) A[a * S + a] = mean[a];
        }
        for (int j = tid; j < N; j += kThreads) {
            double s = 0.0;
            for (int t = lo; t < hi; ++t) s += p.ret[(size_t)t * N + j];
            x[j] = p.ret[(size_t)d * N + j] - s / (double)n;
        }
        __syncthreads();
    }

    double tr0 = 0.0;
    for (int k = 0; k < N; ++k) tr0 += fabs(A[k * S + k]);
    const double floor_abs = 1e-26 * tr0;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        if (tid == 0) *flag = 0;
        __syncthreads();
        for (int step = 0; step < M - 1; ++step) {
            // round-robin pairing of M players: M-1 stays, the others rotate
            if (tid < H) {
                int pi = tid == 0 ? M - 1 : (step + tid) % (M - 1);
                int qi = tid == 0 ? step : (step - tid + (M - 1)) % (M - 1);
                double c = 1.0, s = 0.0;
                if (pi < N && qi < N) {
                    const double app = A[pi * S + pi], aqq = A[qi * S + qi], apq = A[pi * S + qi];
                    const double mag = fabs(apq);
                    if (mag > 0x1p-53 * sqrt(fabs(app * aqq)) && mag > floor_abs) {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        s = t * c;
                        *flag = 1;
                    }
                }
                cs[2 * tid] = c;
                cs[2 * tid + 1] = s;
            }
            __syncthreads();
            for (int f = tid; f < N * H; f += kThreads) {                 // A <- A J
                const int r = f / H, k = f - r * H;
                const int pi = k == 0 ? M - 1 : (step + k) % (M - 1);
                const int qi = k == 0 ? step : (step - k + (M - 1)) % (M - 1);
                const double c = cs[2 * k], s = cs[2 * k + 1];
                if (pi < N && qi < N && s != 0.0) {
                    const double arp = A[r * S + pi], arq = A[r * S + qi];
                    A[r * S + pi] = c * arp - s * arq;
                    A[r * S + qi] = s * arp + c * arq;
                }
            }
            __syncthreads();
            for (int f = tid; f < H * (N + 1); f += kThreads) {           // A <- J^T A, x <- J^T x
                const int k = f / (N + 1), col = f - k * (N + 1);
                const int pi = k == 0 ? M - 1 : (step + k) % (M - 1);
                const int qi = k == 0 ? step : (step - k + (M - 1)) % (M - 1);
                const double c = cs[2 * k], s = cs[2 * k + 1];
                if (pi < N && qi < N && s != 0.0) {
                    if (col == N) {
                        const double xp = x[pi], xq = x[qi];
                        x[pi] = c * xp - s * xq;
                        x[qi] = s * xp + c * xq;
                    } else {
                        const double bp = A[pi * S + col], bq = A[qi * S + col];
                        const bool zero_p = col == qi, zero_q = col == pi;   // annihilated pair
                        A[pi * S + col] = zero_p ? 0.0 : c * bp - s * bq;
                        A[qi * S + col] = zero_q ? 0.0 : s * bp + c * bq;
                    }
                }
            }
            __syncthreads();
        }
        const int any = *flag;
        __syncthreads();
        if (!any) break;
    }
    if (tid == 0) {
        double lmax = 0.0;
        for (int k = 0; k < N; ++k) lmax = fmax(lmax, fabs(A[k * S + k]));
        const double cutoff = 1e-15 * lmax;                    // np.linalg.pinv default rcond
        double q = 0.0;
        for (int k = 0; k < N; ++k) {
            const double lam = A[k * S + k];
            if (fabs(lam) > cutoff) q += x[k] * x[k] / lam;
        }
        p.quad_out[d] = q;                                     // temp, :244-246
    }
}

// :247-257 -- keep a value only from the third positive one on; one wave scans the days
__global__ void __launch_bounds__(64) turbulence_filter_kernel(const double *__restrict__ quad,
                                                               double *__restrict__ out, int T,
                                                               int window)
{
    const int lane = threadIdx.x;
    int count = 0;
    for (int base = 0; base < T; base += 64) {
        const int i = base + lane;
        const double v = (i < T && i >= window) ? quad[i] : 0.0;
        const bool pos = v > 0.0;
        const unsigned long long m = __ballot(pos);
        const int rank = count + __popcll(m & ((1ull << lane) - 1ull)) + 1;
        if (i < T) out[i] = (pos && rank > 2) ? v : 0.0;
        count += __popcll(m);
    }
}

size_t risk_lds_bytes(int N)
{
    const int M = (N + 1) & ~1;
    return sizeof(double) * ((size_t)N * (N + 1) + 2 * (size_t)N + (size_t)M) + 16;
}

int risk_hip_status()
{
    return hipGetLastError() == hipSuccess ? FINENV_OK : FINENV_ERR_HIP;
}

template <bool QUAD>
int launch_rolling(const RiskParams &p, int n_blocks, hipStream_t stream)
{
    const size_t lds = QUAD ? risk_lds_bytes(p.N) : sizeof(double) * (size_t)p.N;
    // > 64 KiB of dynamic LDS needs an explicit opt-in (rare call: set it on every such launch,
    // it is per device)
    if (QUAD && lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&rolling_risk_kernel<QUAD>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)risk_lds_bytes(FINENV_RISKPRE_MAX_ASSETS)) != hipSuccess)
            return FINENV_ERR_HIP;
    }
    hipLaunchKernelGGL((rolling_risk_kernel<QUAD>), dim3((unsigned)n_blocks), dim3(kThreads), lds,
                       stream, p);
    return risk_hip_status();
}

}  // namespace

extern "C" {

int finenv_riskpre_returns(const double *close, double *returns, int32_t n_days, int32_t n_assets,
                           void *stream)
{
    if (!close || !returns || n_days < 1 || n_assets < 1) return FINENV_ERR_INVALID;
    const finenv_host::DeviceGuard guard(finenv_host::pointer_device(close));
    const long long total = (long long)n_days * n_assets;
    const unsigned blocks = (unsigned)((total + kThreads - 1) / kThreads > 4096
                                           ? 4096 : (total + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(returns_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, close,
                       returns, n_days, n_assets);
    return risk_hip_status();
}

int finenv_riskpre_turbulence(const double *returns, double *quad, double *turbulence,
                              int32_t n_days, int32_t n_assets, int32_t window, void *stream)
{
    if (!returns || !quad || !turbulence || n_assets < 2 ||
        n_assets > FINENV_RISKPRE_MAX_ASSETS || window < 3 || n_days < window)
        return FINENV_ERR_INVALID;
    const finenv_host::DeviceGuard guard(finenv_host::pointer_device(returns));
    if (n_days > window) {
        RiskParams p = {returns, nullptr, quad, n_days, n_assets, window, 0, window};
        const int rc = launch_rolling<true>(p, n_days - window, (hipStream_t)stream);
        if (rc != FINENV_OK) return rc;
    }
    hipLaunchKernelGGL(turbulence_filter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, quad,
                       turbulence, n_days, window);
    return risk_hip_status();
}

int finenv_riskpre_rolling_cov(const double *returns, double *cov_out, int32_t n_days,
                               int32_t n_assets, int32_t lookback, void *stream)
{
    if (!returns || !cov_out || n_assets < 1 || n_assets > FINENV_RISKPRE_MAX_ASSETS ||
        lookback < 2 || n_days <= lookback)
        return FINENV_ERR_INVALID;
    const finenv_host::DeviceGuard guard(finenv_host::pointer_device(returns));
    RiskParams p = {returns, cov_out, nullptr, n_days, n_assets, lookback, 1, lookback};
    return launch_rolling<false>(p, n_days - lookback, (hipStream_t)stream);
}

}  // extern "C"
