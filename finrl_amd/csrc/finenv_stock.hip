// finenv_stock.hip -- MI355X (gfx950) kernels + C ABI for the batched StockTradingEnv.
//
// Replaces the per-timestep work of the reference's
//   finrl/meta/env_stock_trading/env_stocktrading.py  step() :220-357, reset() :359-393,
//   _sell_stock :102-169, _buy_stock :171-213, _update_state :453-478
// for E independent environments in one launch.  Not a translation: the reference is a
// Python list + pandas object per env; here the state is a structure-of-arrays in HBM and
// one wavefront lane owns one environment.
//
// Mapping (see DESIGN.md "stock_step"; kernels in finenv_stock_kernels.inc / finenv_stock_wide.inc,
// compiled per padded ticker count in finenv_stock_np{32,64,128}.hip):
//   * lane = env; one 128-thread block per 64 envs with TWO SPECIALISED WAVES: the "trader" owns
//     the env state (staging, sort, sells, buys, assets, reward, state write-back), the "streamer"
//     writes the market-data part of the observation rows from the first microsecond on; they meet
//     at one hand-off barrier, after which both write the rows' cash / holdings chunk(s);
//   * the wave's [64][N] action tile is read coalesced and transposed through LDS;
//   * (action, ticker) pairs become composite int keys per lane, sorted in VGPRs by a Batcher
//     network == the reference's stable argsort order;
//   * sells then buys walk the sorted keys; holdings live in LDS as [ticker][lane]; the cash chain
//     is fp64 with the reference's operation order (-ffp-contract=off), floor division is exact
//     (reciprocal + FMA-remainder correction);
//   * the [64][D] f32 observation block -- 76 % of all bytes -- comes from a pre-packed f32 panel
//     row (L2-resident); cash / holdings are patched in from LDS.
// HBM-bound by design (no MFMA: there is no contraction here).

#include "finenv_stock_common.h"

namespace {

// Terminal summary :226-264 from current state: one lane per env.
__global__ void stock_stats_kernel(const Params p)
{
    const int E = p.cfg.n_envs, N = p.cfg.n_tickers;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const double *prow = p.panel.close + (size_t)SI(FINENV_SI_PRICE_DAY) * N;
    double s = 0.0;
    for (int i = 0; i < N; ++i) s = s + fabs(prow[i]) * (double)HOLD(i);   // sign bit = flag
    const double end = SF(FINENV_SF_CASH) + s;
    const double a0 = SF(FINENV_SF_ASSET0);
    double *out = p.stats_out + (size_t)e * 6;
    out[0] = a0;
    out[1] = end;
    out[2] = end - a0;
    out[3] = SF(FINENV_SF_COST);
    out[4] = (double)SI(FINENV_SI_TRADES);
    double sharpe = __builtin_nan("");
    const int n = SI(FINENV_SI_DAY) - SI(FINENV_SI_START_DAY);   // daily returns accumulated
    if (n >= 2) {        // sqrt(252) * mean / std(ddof=1) from the running sums
        const double s1 = SF(FINENV_SF_RET_SUM), s2 = SF(FINENV_SF_RET_SUMSQ);
        const double mean = s1 / (double)n;
        const double var = (s2 - s1 * mean) / (double)(n - 1);
        if (var > 0.0) sharpe = sqrt(252.0) * mean / sqrt(var);
    }
    out[5] = sharpe;
}

}  // namespace

// =====================================================================================
// Host side: handle, validation, launches.  No allocation on the device, no sync.
// =====================================================================================
struct finenv_stock {
    int device;           // HIP device that owns the bound state block (-1 before bind)
    finenv_stock_config cfg;
    finenv_stock_panel panel;
    finenv_stock_state st;
    int bound;
    int D;
    int obs_pitch;        // row pitch of the obs buffers handed to step / reset / observe (floats)
    int desync_hint;      // finenv_stock_set_desync_hint
    uint32_t magicN;
    char err[256];
};

namespace {

int fail(finenv_stock *h, int code, const char *fmt, const char *detail = "")
{
    if (h) snprintf(h->err, sizeof(h->err), fmt, detail);
    return code;
}

int check_launch(finenv_stock *h, const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        return FINENV_ERR_HIP;
    }
    return FINENV_OK;
}

Params make_params(const finenv_stock *h)
{
    Params p;
    memset(&p, 0, sizeof(p));
    p.cfg = h->cfg;
    p.panel = h->panel;
    p.st = h->st;
    p.D = h->D;
    p.obs_pitch = h->obs_pitch;
    p.desync_hint = h->desync_hint;
    p.magicN = h->magicN;
    return p;
}

int launch_aux(finenv_stock *h, const Params &p, int mode, hipStream_t stream)
{
    if (h->cfg.n_tickers <= 32) finenv_stock_impl::launch_aux_np32(p, mode, stream);
    else if (h->cfg.n_tickers <= 64) finenv_stock_impl::launch_aux_np64(p, mode, stream);
    else finenv_stock_impl::launch_aux_np128(p, mode, stream);
    return 0;
}

// step kernel: one 128-thread block per 64 envs; 33..64 tickers use the 128-wide code paths at
// half the padding (34 KB of LDS, four blocks per CU instead of two)
int launch_step(finenv_stock *h, const Params &p, bool turb, bool stats, hipStream_t stream)
{
    if (h->cfg.n_tickers <= 32) return finenv_stock_impl::launch_step_np32(p, turb, stats, h->device, stream);
    if (h->cfg.n_tickers <= 64) return finenv_stock_impl::launch_step_np64(p, turb, stats, h->device, stream);
    return finenv_stock_impl::launch_step_np128(p, turb, stats, h->device, stream);
}

}  // namespace

#ifdef FINENV_DIAG
unsigned long long *g_finenv_dbg = nullptr;      // shared with the other kernels' diagnostic builds
#define g_dbg g_finenv_dbg
extern "C" void finenv_diag_set_stamp_buffer(void *ptr) { g_finenv_dbg = (unsigned long long *)ptr; }
#endif

extern "C" {

int finenv_abi_version(void) { return FINENV_ABI_VERSION; }

int finenv_struct_size(int which)
{
    switch (which) {
    case 0: return (int)sizeof(finenv_stock_config);
    case 1: return (int)sizeof(finenv_stock_panel);
    case 2: return (int)sizeof(finenv_stock_state);
    case 3: return (int)sizeof(finenv_portfolio_config);
    case 4: return (int)sizeof(finenv_portfolio_panel);
    case 5: return (int)sizeof(finenv_portfolio_state);
    case 6: return (int)sizeof(finenv_crypto_config);
    case 7: return (int)sizeof(finenv_crypto_panel);
    case 8: return (int)sizeof(finenv_crypto_state);
    case 9: return (int)sizeof(finenv_stocknp_config);
    case 10: return (int)sizeof(finenv_stocknp_panel);
    case 11: return (int)sizeof(finenv_stocknp_state);
    case 12: return (int)sizeof(finenv_cashpenalty_config);
    case 13: return (int)sizeof(finenv_cashpenalty_panel);
    case 14: return (int)sizeof(finenv_cashpenalty_state);
    case 15: return (int)sizeof(finenv_stoploss_config);
    case 16: return (int)sizeof(finenv_stoploss_panel);
    case 17: return (int)sizeof(finenv_stoploss_state);
    default: return FINENV_ERR_INVALID;
    }
}

const char *finenv_strerror(int code)
{
    switch (code) {
    case FINENV_OK: return "ok";
    case FINENV_ERR_INVALID: return "invalid argument";
    case FINENV_ERR_UNBOUND: return "panel/state not bound";
    case FINENV_ERR_HIP: return "HIP runtime error";
    case FINENV_ERR_NOMEM: return "out of host memory";
    default: return "unknown error";
    }
}

int finenv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return FINENV_ERR_HIP;
    }
    return n;
}

int finenv_stock_create(const finenv_stock_config *cfg, finenv_stock **out)
{
    if (!cfg || !out) return FINENV_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_envs < 1 || cfg->n_tickers < 1 || cfg->n_tickers > FINENV_STOCK_MAX_TICKERS ||
        cfg->n_tech < 0 || cfg->n_days < 1 || cfg->hmax < 0 ||
        cfg->hmax > (cfg->n_tickers <= 32 ? (1 << 24) : (1 << 22)))
        return FINENV_ERR_INVALID;
    if (cfg->single_ticker && cfg->n_tickers != 1) return FINENV_ERR_INVALID;
    if ((long long)cfg->n_envs * (1 + 2 * cfg->n_tickers + cfg->n_tech * cfg->n_tickers) >
        (1ll << 40))
        return FINENV_ERR_INVALID;
    {   // every device offset is a 32-bit byte offset from a uniform base (see at())
        const long long E = cfg->n_envs, N = cfg->n_tickers, T = cfg->n_days;
        const long long D = 1 + 2 * N + (long long)cfg->n_tech * N;
        const long long lim = (1ll << 32) - 1;
        if ((FINENV_STOCK_I32_FIELDS + 2 * N) * E * 4 > lim || FINENV_STOCK_F64_FIELDS * E * 8 > lim ||
            T * D * 4 > lim || T * N * 8 > lim || E * N * 4 > lim)
            return FINENV_ERR_INVALID;
    }
    finenv_stock *h = new (std::nothrow) finenv_stock;
    if (!h) return FINENV_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->device = -1;
    h->cfg = *cfg;
    h->D = 1 + 2 * cfg->n_tickers + cfg->n_tech * cfg->n_tickers;
    h->obs_pitch = h->D;
    h->magicN = cfg->n_tickers >= 2
                    ? (uint32_t)(((1ull << 32) + cfg->n_tickers - 1) / (unsigned)cfg->n_tickers)
                    : 0u;
    *out = h;
    return FINENV_OK;
}

void finenv_stock_destroy(finenv_stock *h) { delete h; }

const char *finenv_stock_last_error(const finenv_stock *h) { return h ? h->err : "null handle"; }

int finenv_stock_obs_dim(const finenv_stock *h) { return h ? h->D : FINENV_ERR_INVALID; }

int finenv_stock_set_desync_hint(finenv_stock *h, int32_t on)
{
    if (!h) return FINENV_ERR_INVALID;
    h->desync_hint = on != 0;
    return FINENV_OK;
}

int finenv_stock_set_obs_pitch(finenv_stock *h, int32_t pitch)
{
    if (!h) return FINENV_ERR_INVALID;
    if (pitch == 0) pitch = h->D;
    if (pitch < h->D || (long long)pitch * 64 * 4 > (1ll << 32) - 1)
        return fail(h, FINENV_ERR_INVALID, "set_obs_pitch: pitch must be >= obs_dim");
    h->obs_pitch = pitch;
    return FINENV_OK;
}

int finenv_stock_bind(finenv_stock *h, const finenv_stock_panel *panel,
                      const finenv_stock_state *st)
{
    if (!h || !panel || !st) return FINENV_ERR_INVALID;
    if (!panel->close || !panel->obs_tmpl ||
        (h->cfg.use_turbulence && !panel->risk))
        return fail(h, FINENV_ERR_INVALID, "bind: null panel pointer%s");
    if (!st->f64 || !st->i32)
        return fail(h, FINENV_ERR_INVALID, "bind: null state pointer%s");
    h->panel = *panel;
    h->st = *st;
    h->device = finenv_host::pointer_device(st->f64);
    h->bound = 1;
    return FINENV_OK;
}

int finenv_stock_init(finenv_stock *h, int32_t day0, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "init: bind first%s");
    const finenv_host::DeviceGuard guard(h->device);
    if (day0 < 0 || day0 >= h->cfg.n_days) return fail(h, FINENV_ERR_INVALID, "init: bad day0%s");
    Params p = make_params(h);
    p.day0 = day0;
    launch_aux(h, p, 0, (hipStream_t)stream);
    return check_launch(h, "stock_init");
}

int finenv_stock_reset(finenv_stock *h, const uint8_t *mask, float *obs_out, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "reset: bind first%s");
    const finenv_host::DeviceGuard guard(h->device);
    Params p = make_params(h);
    p.mask = mask;
    p.obs = obs_out;
    launch_aux(h, p, 1, (hipStream_t)stream);
    return check_launch(h, "stock_reset");
}

int finenv_stock_observe(finenv_stock *h, float *obs_out, void *stream)
{
    if (!h || !obs_out) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "observe: bind first%s");
    const finenv_host::DeviceGuard guard(h->device);
    Params p = make_params(h);
    p.obs = obs_out;
    launch_aux(h, p, 2, (hipStream_t)stream);
    return check_launch(h, "stock_observe");
}

int finenv_stock_refresh(finenv_stock *h, void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "refresh: bind first%s");
    const finenv_host::DeviceGuard guard(h->device);
    Params p = make_params(h);
    launch_aux(h, p, 3, (hipStream_t)stream);
    return check_launch(h, "stock_refresh");
}

int finenv_stock_step(finenv_stock *h, const float *actions, float *obs, float *reward,
                      uint8_t *done, float *term_obs, int32_t *realised, int32_t auto_reset,
                      void *stream)
{
    if (!h) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "step: bind first%s");
    const finenv_host::DeviceGuard guard(h->device);
    if (!actions || !obs || !reward || !done)
        return fail(h, FINENV_ERR_INVALID, "step: null actions/obs/reward/done%s");
    Params p = make_params(h);
    p.actions = actions;
    p.obs = obs;
    p.reward = reward;
    p.done = done;
    p.term_obs = term_obs;
    p.realised = realised;
    p.auto_reset = auto_reset;
#ifdef FINENV_DIAG
    {
        const char *d = getenv("FINENV_DIAG");
        p.diag = d ? atoi(d) : 0;
        p.dbg = g_dbg;
    }
#endif
    const hipStream_t s = (hipStream_t)stream;
    const bool turb = h->cfg.use_turbulence != 0, stats = h->cfg.track_stats != 0;
    const int rc = launch_step(h, p, turb, stats, s);
    if (rc) return fail(h, FINENV_ERR_HIP, "step: cannot raise the dynamic LDS limit%s");
    return check_launch(h, "stock_step");
}

int finenv_stock_episode_stats(finenv_stock *h, double *out, void *stream)
{
    if (!h || !out) return FINENV_ERR_INVALID;
    if (!h->bound) return fail(h, FINENV_ERR_UNBOUND, "episode_stats: bind first%s");
    const finenv_host::DeviceGuard guard(h->device);
    Params p = make_params(h);
    p.stats_out = out;
    const int E = h->cfg.n_envs;
    hipLaunchKernelGGL(stock_stats_kernel, dim3((E + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, p);
    return check_launch(h, "stock_episode_stats");
}

}  // extern "C"
