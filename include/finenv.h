/*
 * finenv.h -- C ABI of libfinenv.so: MI355X-native batched trading environments.
 *
 * This is the drop-in boundary for the reference's market-environment hot path
 * (superyuri/FinRL, finrl/meta/env_stock_trading/env_stocktrading.py).  The reference
 * is pure Python, so there is no existing FFI to mirror symbol-for-symbol; each entry
 * point below names the reference method (file:line) whose work it replaces, and
 * INTEGRATION.md shows the ctypes binding a FinRL maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - Every buffer is CALLER-OWNED DEVICE memory (e.g. torch tensor .data_ptr());
 *     the library allocates nothing on the device and never synchronises the stream.
 *   - All launches go to the caller's hipStream_t, passed as void* (0 = null stream).
 *   - Return value: 0 = FINENV_OK, negative = error (finenv_strerror /
 *     finenv_stock_last_error).  No exceptions cross the ABI.
 *   - One handle per (device, stream) user; handles are thread-compatible, not
 *     thread-safe.
 *   - There is NO CPU fallback: without a HIP device every launch entry point fails
 *     with FINENV_ERR_HIP.
 *
 * Layout (E envs, N tickers, K indicators, T days, D = 1 + 2N + K*N)
 *   actions  [E][N] f32 row-major  (what SB3 / ElegantRL hand over)
 *   obs      [E][D] f32 row-major  = [cash | close[N] | holdings[N] | tech[K][N]]
 *                                    (indicator-major, env_stocktrading.py:456-467)
 *   state    structure-of-arrays over envs ([field][E] blocks); holdings is [N][E]
 *            (ticker-major) so that lane e of a wavefront reads holdings[i][e] coalesced.
 *   Limits   every device array must stay below 4 GiB (32-bit lane offsets).
 */
#ifndef FINENV_H
#define FINENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FINENV_ABI_VERSION 3

enum {
    FINENV_OK = 0,
    FINENV_ERR_INVALID = -1,     /* bad argument / unsupported shape            */
    FINENV_ERR_UNBOUND = -2,     /* panel or state not bound yet                */
    FINENV_ERR_HIP = -3,         /* HIP runtime error (see *_last_error)        */
    FINENV_ERR_NOMEM = -4
};

#define FINENV_STOCK_MAX_TICKERS 128  /* two kernel variants: N <= 32 (DOW30), N <= 128
                                         (NASDAQ-100)                               */

/* Constructor arguments of StockTradingEnv that shape the arithmetic
 * (env_stocktrading.py:24-47). */
typedef struct finenv_stock_config {
    int32_t n_envs;               /* E                                                   */
    int32_t n_tickers;            /* stock_dim, :50                                      */
    int32_t n_tech;               /* len(tech_indicator_list), :59                       */
    int32_t n_days;               /* len(df.index.unique()), :221                        */
    int32_t hmax;                 /* :51; |action*hmax| must stay below 2^25 (N <= 32) or
                                     2^23 (N <= 128)                                      */
    int32_t use_turbulence;       /* turbulence_threshold is not None, :68               */
    int32_t reset_quirk;          /* 1: reset() builds obs from the row held before the
                                     rewind (reference behaviour, :361 vs :380-381)      */
    int32_t initial;              /* `initial` flag, :70 -- selects the summation order
                                     of asset_memory[0] (:364-378)                       */
    int32_t track_stats;          /* keep running sums of daily returns (Sharpe,
                                     :243-251) on device                                 */
    int32_t single_ticker;        /* 1 <=> len(df.tic.unique()) == 1 (needs n_tickers == 1):
                                     the reference's single-stock branches -- with `initial`
                                     the state starts from `[0] * stock_dim` shares whatever
                                     num_stock_shares holds (:415-422) while asset_memory[0]
                                     still counts them (:85-91, :364-370).  0 keeps the
                                     multi-stock rule for a 1-wide batch                   */
    double  buy_cost_pct;         /* scalar in this fork, :54                            */
    double  sell_cost_pct;        /* :55                                                 */
    double  reward_scaling;       /* :56                                                 */
    double  turbulence_threshold; /* :68                                                 */
} finenv_stock_config;

/* Read-only market panel (device pointers), packed by finrl_amd.panel.StockPanel from
 * the DataFrame the reference env receives (index = day ordinal, rows sorted by
 * (date, tic): preprocessors.py:24-33). */
typedef struct finenv_stock_panel {
    const double   *close;        /* [T][N]  fp64 closes: the money arithmetic runs on
                                     the same doubles the reference holds in state[1..N].
                                     The SIGN BIT of close[t][i] carries the day's
                                     "untradable" flag of ticker i: set <=> its first
                                     indicator == 1.0 on day t (the fork's `!= True` test,
                                     :105, :174, evaluated on the fp64 values); prices
                                     themselves are >= 0 by contract                     */
    const float    *obs_tmpl;     /* [T][D]  f32 observation rows with the cash and
                                     holdings slots zero: f32(close) and f32(tech) in obs
                                     order (what DummyVecEnv's float32 buffer would hold) */
    const double   *risk;         /* [T]     df[risk_indicator_col], :337-341 (may be
                                     NULL when use_turbulence == 0)                      */
} finenv_stock_panel;

/* Mutable per-env state: two caller-owned device blocks of [field][env] arrays
 * (structure-of-arrays; lane e of a wavefront touches element e of every field, so every
 * access is coalesced, and two base pointers keep the kernels' scalar-register footprint
 * small).  Field order: */
enum {                            /* f64 block: double f64[FINENV_STOCK_F64_FIELDS][E]  */
    FINENV_SF_CASH = 0,           /* state[0]                                            */
    FINENV_SF_COST,               /* self.cost                                           */
    FINENV_SF_LAST_REWARD,        /* self.reward (scaled; survives reset)                */
    FINENV_SF_TURBULENCE,         /* self.turbulence                                     */
    FINENV_SF_ASSET0,             /* asset_memory[0] (written by init / reset only; the
                                     running asset_memory[-1] is never stored: it equals
                                     the next step's begin asset bit for bit)            */
    FINENV_SF_RET_SUM,            /* sum of pct_change(asset_memory) this episode        */
    FINENV_SF_RET_SUMSQ,          /* sum of its squares (Sharpe at the terminal step)    */
    FINENV_SF_CASH0,              /* initial_amount / previous_state[0] (read-only)      */
    FINENV_SF_BEGIN_ASSET,        /* cash + sum(close * shares) of the CURRENT observation,
                                     summed sequentially from ticker 0 as :311-314 does: the
                                     next step's begin_total_asset.  It is the previous step's
                                     end_total_asset bit for bit (same state list, same
                                     expression, :344-347), so step() carries it over instead
                                     of recomputing it; init / reset evaluate it afresh       */
    FINENV_STOCK_F64_FIELDS
};
enum {                            /* i32 block: int32 i32[FINENV_STOCK_I32_FIELDS+2N][E] */
    FINENV_SI_DAY = 0,            /* self.day                                            */
    FINENV_SI_PRICE_DAY,          /* row whose prices/indicators sit in the current
                                     observation (== day except right after a quirk
                                     reset)                                              */
    FINENV_SI_TRADES,             /* self.trades                                         */
    FINENV_SI_EPISODE,            /* self.episode                                        */
    FINENV_SI_START_DAY,          /* day the episode started on (init: day0, reset: 0);
                                     daily returns accumulated so far = day - start_day  */
    FINENV_STOCK_I32_FIELDS       /* followed by holdings[N][E] = state[1+N .. 1+2N) and
                                     shares0[N][E] = num_stock_shares / previous_state
                                     shares (read-only)                                  */
};
typedef struct finenv_stock_state {
    double  *f64;                 /* [FINENV_STOCK_F64_FIELDS][E]                        */
    int32_t *i32;                 /* [FINENV_STOCK_I32_FIELDS + 2N][E]                   */
} finenv_stock_state;

typedef struct finenv_stock finenv_stock;   /* opaque host-side handle */

int         finenv_abi_version(void);
/* sizeof() of the ABI structs as the library was compiled (0 = finenv_stock_config,
 * 1 = finenv_stock_panel, 2 = finenv_stock_state, 3..5 = the finenv_portfolio_* trio, 6..8 = the finenv_crypto_* trio, 9..11 = the finenv_stocknp_* trio,
 * 12..14 = the finenv_cashpenalty_* trio, 15..17 = the finenv_stoploss_* trio): lets a foreign-language binding verify its
 * struct declarations at load time instead of corrupting memory. */
int         finenv_struct_size(int which);
const char *finenv_strerror(int code);
/* Number of HIP devices visible, or a negative FINENV_ERR_HIP. */
int         finenv_device_count(void);

/* StockTradingEnv.__init__ (:24-100), config part.  Validates shapes. */
int  finenv_stock_create(const finenv_stock_config *cfg, finenv_stock **out);
void finenv_stock_destroy(finenv_stock *h);
const char *finenv_stock_last_error(const finenv_stock *h);
int  finenv_stock_obs_dim(const finenv_stock *h);

/* Row pitch, in floats, of the observation buffers later handed to step / reset / observe
 * (terminal observations stay packed).  Default = obs_dim (packed [E][D] rows, what the reference
 * returns); 0 restores it.  A pitch that is a multiple of 16 floats starts every row on a 64-byte
 * boundary: packed rows of 1204 B share their first and last 64-B segment with a neighbour that
 * is written ~13 us earlier or later (two partial HBM writes instead of one; PMC: 1.05x the bytes
 * stored).  The consumer sees a [E][D] view with a row stride. */
int  finenv_stock_set_obs_pitch(finenv_stock *h, int32_t pitch);

/* Performance hint, never needed for correctness: the envs of this batch may sit on different days
 * (per-env start days, episodes that end at different steps).  step() then launches the kernel
 * instantiation whose per-env panel-row paths are tuned (16-byte row copies, row-wise price gather,
 * parked head chunks); with every env on the same day -- the reference's training setup, BASELINE's
 * configs -- leave it off: that instantiation carries none of that code. */
int  finenv_stock_set_desync_hint(finenv_stock *h, int32_t on);

/* Attach the panel and state buffers (replaces self.df / self.state ownership). */
int finenv_stock_bind(finenv_stock *h, const finenv_stock_panel *panel,
                      const finenv_stock_state *state);

/* __init__ state part (:64-91): every env starts on `day0` with cash0/shares0,
 * turbulence = cost = trades = episode = 0; no observation is produced
 * (use finenv_stock_observe). */
int finenv_stock_init(finenv_stock *h, int32_t day0, void *stream);

/* reset() (:359-393) for all envs, or those with mask[e] != 0 (device u8[E], may be
 * NULL).  Writes the reset observation rows into obs_out [E][D] (rows of unmasked envs
 * are left untouched). */
int finenv_stock_reset(finenv_stock *h, const uint8_t *mask, float *obs_out, void *stream);

/* render() / current state (:395-396): writes obs [E][D] without changing state. */
int finenv_stock_observe(finenv_stock *h, float *obs_out, void *stream);

/* After the CALLER has edited cash, holdings or price_day of the bound state block in place
 * (restoring a snapshot, desynchronising a batch): re-evaluate FINENV_SF_BEGIN_ASSET = cash +
 * sum(close[price_day] * holdings) in the reference's order (:311-314).  step() carries that field
 * from one step's end asset to the next step's begin asset and never recomputes it; init / reset
 * write it themselves.  Without this call the first reward after such an edit is computed against
 * a stale begin asset. */
int finenv_stock_refresh(finenv_stock *h, void *stream);

/* step() (:220-357) for all envs in ONE launch.
 *   actions   [E][N] f32 in [-1, 1]
 *   obs       [E][D] f32 (next observation; after auto-reset: the reset observation)
 *   reward    [E]    f32 (float32 cast of the fp64 reward, as DummyVecEnv stores it)
 *   done      [E]    u8
 *   term_obs  [E][D] f32 or NULL: rows of envs with done == 1 receive the terminal
 *             observation (SB3 info["terminal_observation"]); other rows untouched
 *   realised  [E][N] i32 or NULL: shares actually traded (the values the reference
 *             writes back into `actions`, :324/:330 -> actions_memory)
 *   auto_reset != 0: envs that report done are reset inside the same launch
 *             (SB3 DummyVecEnv.step_wait semantics); 0: plain gym semantics (state
 *             unchanged on the terminal step, :301).
 */
int finenv_stock_step(finenv_stock *h, const float *actions, float *obs, float *reward,
                      uint8_t *done, float *term_obs, int32_t *realised, int32_t auto_reset,
                      void *stream);

/* Terminal-branch summary (:226-264) for every env, computed from current state:
 * out [E][6] f64 = {begin_total_asset, end_total_asset, total_reward, total_cost,
 *                   total_trades, sharpe (NaN if undefined)}. */
int finenv_stock_episode_stats(finenv_stock *h, double *out, void *stream);

/* =====================================================================================
 * StockPortfolioEnv (finrl/meta/env_portfolio_allocation/env_portfolio.py:15-261)
 *   actions [E][N] f32 (portfolio scores; softmax-normalised inside, :225-229)
 *   obs     [E][D] f32, D = (N + K) * N: the day's N x N covariance rows then K indicator
 *           rows (:172-179) -- independent of per-env state
 *   reward  = new portfolio value, unscaled (:196-198)
 * Per-env state: portfolio_value, last reward (f64), day (i32).
 * ===================================================================================== */
#define FINENV_PORTFOLIO_MAX_TICKERS 64

typedef struct finenv_portfolio_config {
    int32_t n_envs;
    int32_t n_tickers;            /* stock_dim                                           */
    int32_t n_tech;               /* len(tech_indicator_list)                            */
    int32_t n_days;               /* len(df.index.unique()), :127                        */
    double  initial_amount;       /* :88, restored by reset() (:213)                     */
} finenv_portfolio_config;

typedef struct finenv_portfolio_panel {
    const double *gross_ret;      /* [T][N] f64: row t = close[t+1]/close[t] - 1, evaluated
                                     elementwise in fp64 exactly as :184 (row T-1 unused) */
    const float  *obs_tmpl;       /* [T][D] f32 observation rows                          */
} finenv_portfolio_panel;

enum { FINENV_PF_VALUE = 0, FINENV_PF_LAST_REWARD, FINENV_PORTFOLIO_F64_FIELDS };
enum { FINENV_PI_DAY = 0, FINENV_PORTFOLIO_I32_FIELDS };
typedef struct finenv_portfolio_state {
    double  *f64;                 /* [FINENV_PORTFOLIO_F64_FIELDS][E]                     */
    int32_t *i32;                 /* [FINENV_PORTFOLIO_I32_FIELDS][E]                     */
} finenv_portfolio_state;

typedef struct finenv_portfolio finenv_portfolio;

int  finenv_portfolio_create(const finenv_portfolio_config *cfg, finenv_portfolio **out);
void finenv_portfolio_destroy(finenv_portfolio *h);
const char *finenv_portfolio_last_error(const finenv_portfolio *h);
int  finenv_portfolio_obs_dim(const finenv_portfolio *h);
int  finenv_portfolio_bind(finenv_portfolio *h, const finenv_portfolio_panel *panel,
                           const finenv_portfolio_state *state);
/* reset() (:202-220) for all envs or those with mask[e] != 0; obs rows of reset envs. */
int  finenv_portfolio_reset(finenv_portfolio *h, const uint8_t *mask, float *obs_out,
                            void *stream);
/* step() (:125-200); weights_out [E][N] f32 or NULL receives the softmax weights
 * (actions_memory, :168); term_obs / auto_reset as in finenv_stock_step. */
int  finenv_portfolio_step(finenv_portfolio *h, const float *actions, float *obs, float *reward,
                           uint8_t *done, float *term_obs, float *weights_out,
                           int32_t auto_reset, void *stream);

/* =====================================================================================
 * CryptoEnv (finrl/meta/env_cryptocurrency_trading/env_multiple_crypto.py:10-111)
 *   actions [E][N] f32 in [-1,1], scaled per asset by the action normaliser (:63-65, :103-111)
 *   obs     [E][D] f32, D = 1 + N + W*lookback = [cash*2^-18 | stocks*2^-3 | tech[time-l]*2^-15]
 *   reward  (delta total asset) * 2^-16; on the last step the discounted return (:83-89)
 * Contract: price / tech arrays are float64 (as the reference's processors build them);
 * stocks are float32 (fractional), cash and assets float64.
 * ===================================================================================== */
#define FINENV_CRYPTO_MAX_ASSETS 32

typedef struct finenv_crypto_config {
    int32_t n_envs;
    int32_t n_assets;             /* crypto_num, :23                                     */
    int32_t n_tech;               /* tech_array.shape[1]                                 */
    int32_t n_steps;              /* price_array.shape[0]                                */
    int32_t lookback;             /* :13                                                 */
    int32_t reserved0;
    double  initial_cash;         /* initial_capital, :14-15                             */
    double  buy_cost_pct;         /* :16                                                 */
    double  sell_cost_pct;        /* :17                                                 */
    double  gamma;                /* :19                                                 */
} finenv_crypto_config;

typedef struct finenv_crypto_panel {
    const double *price;          /* [T][N] f64                                          */
    const float  *tech_scaled;    /* [T][W] f32 = float32(tech * 2^-15), :95-97          */
    const double *norm;           /* [N] action_norm_vector, :103-111 (host-evaluated)    */
} finenv_crypto_panel;

enum { FINENV_CF_CASH = 0, FINENV_CF_TOTAL_ASSET, FINENV_CF_GAMMA_RETURN,
       FINENV_CF_EPISODE_RETURN, FINENV_CF_LAST_REWARD, FINENV_CRYPTO_F64_FIELDS };
enum { FINENV_CI_TIME = 0, FINENV_CRYPTO_I32_FIELDS };
typedef struct finenv_crypto_state {
    double  *f64;                 /* [FINENV_CRYPTO_F64_FIELDS][E]                        */
    int32_t *i32;                 /* [FINENV_CRYPTO_I32_FIELDS][E]                        */
    float   *stocks;              /* [N][E] f32 holdings (fractional)                     */
} finenv_crypto_state;

typedef struct finenv_crypto finenv_crypto;

int  finenv_crypto_create(const finenv_crypto_config *cfg, finenv_crypto **out);
void finenv_crypto_destroy(finenv_crypto *h);
const char *finenv_crypto_last_error(const finenv_crypto *h);
int  finenv_crypto_obs_dim(const finenv_crypto *h);
int  finenv_crypto_bind(finenv_crypto *h, const finenv_crypto_panel *panel,
                        const finenv_crypto_state *state);
/* reset() (:48-57); gamma_return is NOT cleared, as in the reference. */
int  finenv_crypto_reset(finenv_crypto *h, const uint8_t *mask, float *obs_out, void *stream);
/* step() (:59-90).  The output pointers may address slice t of rollout tensors
 * [n_steps][E][...]: collecting a rollout needs no extra copy. */
int  finenv_crypto_step(finenv_crypto *h, const float *actions, float *obs, float *reward,
                        uint8_t *done, float *term_obs, int32_t auto_reset, void *stream);
/* step() that also records the policy's outputs of this step into the rollout tensors, in the same
 * launch (extra blocks beside the env blocks): actions -> actions_out [E][N], values -> values_out
 * [E], log_probs -> log_probs_out [E] (what SB3's RolloutBuffer.add copies; obs / reward / done are
 * written in place by the step itself).  All six buffers 16-byte aligned, E % 4 == 0, else
 * FINENV_ERR_INVALID (use finenv_crypto_step + finenv_rollout_put). */
int  finenv_crypto_step_record(finenv_crypto *h, const float *actions, float *obs, float *reward,
                               uint8_t *done, float *term_obs, int32_t auto_reset,
                               const float *values, const float *log_probs, float *actions_out,
                               float *values_out, float *log_probs_out, void *stream);

/* =====================================================================================
 * Rollout helper (caller side of the path): generalized advantage estimation
 * over device-resident rollout tensors [n_steps][E], time-reverse scan, one lane per env.
 * Arithmetic follows stable-baselines3's documented RolloutBuffer.compute_returns_and_advantage
 * in float32 (SB3 is not vendored in the reference: parity unpinned, defined against the
 * documented formula):
 *   nnt_t   = 1 - dones[t]                 (dones[t] = done flag returned by step t)
 *   delta_t = rewards[t] + gamma * V_{t+1} * nnt_t - values[t],  V_{n} = last_values
 *   A_t     = delta_t + gamma * lam * nnt_t * A_{t+1};   returns_t = A_t + values[t]
 * ===================================================================================== */
int finenv_gae_scan(const float *rewards, const float *values, const uint8_t *dones,
                    const float *last_values, float *advantages, float *returns,
                    int32_t n_steps, int32_t n_envs, float gamma, float gae_lambda,
                    void *stream);

/* One launch that stores a policy's outputs for step t into the rollout tensors: actions [E][A],
 * values [E], log-probs [E] (float32, contiguous) -> the three destination slices.  Replaces the
 * three tensor copies of SB3's RolloutBuffer.add (its obs / reward / done parts need no copy: the
 * env kernels write them in place). */
int finenv_rollout_put(const float *actions, const float *values, const float *log_probs,
                       float *actions_out, float *values_out, float *log_probs_out,
                       int32_t n_envs, int32_t action_dim, void *stream);

/* =====================================================================================
 * Array-state StockTradingEnv (finrl/meta/env_stock_trading/env_stocktrading_np.py:8-169),
 * the ElegantRL / RLlib-facing env.
 *   actions [E][N] f32;  obs [E][D] f32, D = 3 + 3N + W =
 *     [amount*2^-12 | turbulence_ary[d] | turbulence_bool[d] | price*2^-6 | stocks*2^-6 |
 *      cool_down | tech_ary[d]]                                               (:149-162)
 *   reward = delta total_asset * reward_scaling; discounted return on the last step (:137-145)
 * Numerics: bit-identical to the reference under NumPy >= 2 (NEP 50), where amount /
 * total_asset / gamma_reward are Python-float, float32 or float64 depending on the trade
 * history; the dtype of each is tracked per env (FINENV_NT_*, 2 bits each in the `tags` word)
 * and every operation is performed in the dtype NumPy would use.
 * ===================================================================================== */
#define FINENV_STOCKNP_MAX_TICKERS 32
enum { FINENV_NT_PY = 0, FINENV_NT_F32 = 1, FINENV_NT_F64 = 2 };

typedef struct finenv_stocknp_config {
    int32_t n_envs;
    int32_t n_tickers;
    int32_t n_techw;              /* tech_ary.shape[1] (= N*K, ticker-major)              */
    int32_t n_days;               /* price_ary.shape[0]; max_step = n_days - 1, :67       */
    int32_t min_action;           /* int(max_stock * min_stock_rate), :111                */
    int32_t reserved0;
    double  max_stock;            /* :39 (action scale, :104)                             */
    double  buy_cost_pct, sell_cost_pct, reward_scaling, gamma;
    double  obs_amount_floor;     /* 0: the observation shows self.amount (:150); > 0: Python's
                                     max(self.amount, floor) as StockEnvNAS100.get_state does
                                     (env_nas100_wrds.py:154, floor = 1e4)                 */
} finenv_stocknp_config;

typedef struct finenv_stocknp_panel {
    const float *price;           /* [T][N] price_ary (f32), :27                          */
    const float *obs_tmpl;        /* [T][D] f32 rows: turbulence / price*2^-6 / tech filled,
                                     amount, stocks and cool_down slots zero              */
    const float *turb_bool;       /* [T]    (turbulence > thresh) as f32, :32             */
} finenv_stocknp_panel;

enum { FINENV_NF_AMOUNT = 0, FINENV_NF_TOTAL_ASSET, FINENV_NF_GAMMA_REWARD,
       FINENV_NF_INITIAL_TOTAL_ASSET, FINENV_NF_EPISODE_RETURN, FINENV_NF_LAST_REWARD,
       FINENV_NF_AMOUNT0, FINENV_STOCKNP_F64_FIELDS };
enum { FINENV_NI_DAY = 0, FINENV_NI_TAGS, FINENV_NI_AMOUNT0_TAG, FINENV_STOCKNP_I32_FIELDS };
/* tags word: bits 0-1 amount, 2-3 total_asset, 4-5 gamma_reward, 6-7 initial_total_asset,
 * 8-9 last reward */
typedef struct finenv_stocknp_state {
    double  *f64;                 /* [FINENV_STOCKNP_F64_FIELDS][E]                       */
    int32_t *i32;                 /* [FINENV_STOCKNP_I32_FIELDS][E]                       */
    float   *f32;                 /* [3N][E]: stocks[N], cool_down[N], stocks0[N]          */
} finenv_stocknp_state;

typedef struct finenv_stocknp finenv_stocknp;

int  finenv_stocknp_create(const finenv_stocknp_config *cfg, finenv_stocknp **out);
void finenv_stocknp_destroy(finenv_stocknp *h);
const char *finenv_stocknp_last_error(const finenv_stocknp *h);
int  finenv_stocknp_obs_dim(const finenv_stocknp *h);
/* row pitch (floats) of the obs buffers of step / reset; see finenv_stock_set_obs_pitch */
int  finenv_stocknp_set_obs_pitch(finenv_stocknp *h, int32_t pitch);
int  finenv_stocknp_bind(finenv_stocknp *h, const finenv_stocknp_panel *panel,
                         const finenv_stocknp_state *state);
/* reset() (:80-101) from the per-env start state (stocks0, amount0, amount0_tag): eval mode =
 * (initial_stocks, initial_capital as FINENV_NT_PY); train mode = caller-drawn values with
 * FINENV_NT_F32 (the reference draws them from the global numpy RNG, :85-92). */
int  finenv_stocknp_reset(finenv_stocknp *h, const uint8_t *mask, float *obs_out, void *stream);
int  finenv_stocknp_step(finenv_stocknp *h, const float *actions, float *obs, float *reward,
                         uint8_t *done, float *term_obs, int32_t auto_reset, void *stream);

/* =====================================================================================
 * StockTradingEnvCashpenalty
 * (finrl/meta/env_stock_trading/env_stocktrading_cashpenalty.py:19-409): continuous (or
 * discretised) share counts from dollar-sized actions, no per-ticker ordering, reward =
 * (assets - cash-shortfall penalty) / initial - 1, per elapsed step (:237-247); episode ends at
 * the last date or on a cash shortage (unless patient) (:333-344).
 *   actions [E][N] f32;  obs [E][D] f32, D = 1 + N + N*C = [cash | holdings | info[N][C]]
 * Contract: close > 0, scalar hmax.  The three dot products per step are summed left to right
 * (the reference uses BLAS ddot, order unspecified: agreement ~1e-15 relative).
 * ===================================================================================== */
#define FINENV_CASHPENALTY_MAX_ASSETS 32

typedef struct finenv_cashpenalty_config {
    int32_t n_envs, n_assets, n_cols, n_days;
    int32_t discrete_actions;     /* :60, :263-274                                        */
    int32_t shares_increment;     /* :61                                                  */
    int32_t use_turbulence;       /* turbulence_threshold is not None, :282               */
    int32_t patient;              /* :68, :334-339                                        */
    double  hmax;                 /* :59 (dollars per trade)                              */
    double  buy_cost_pct, sell_cost_pct, initial_amount, cash_penalty_proportion,
            turbulence_threshold;
} finenv_cashpenalty_config;

typedef struct finenv_cashpenalty_panel {
    const double *close;          /* [T][N] f64                                           */
    const float  *info;           /* [T][N*C] f32 date vectors, ticker-major (:159-171)   */
    const double *turb;           /* [T] f64 (may be NULL when use_turbulence == 0)       */
} finenv_cashpenalty_panel;

enum { FINENV_KF_COH = 0, FINENV_KF_TURBULENCE, FINENV_KF_SUM_TRADES, FINENV_KF_LOGGED_TOTAL,
       FINENV_KF_LOGGED_CASH, FINENV_CASHPENALTY_F64_FIELDS /* then holdings[N][E] */ };
enum { FINENV_KI_DATE_INDEX = 0, FINENV_KI_START, FINENV_KI_EPISODE,
       FINENV_KI_NEXT_START,      /* starting point the next reset() uses (random_start: the
                                     caller refills it; the reference draws it with `random`) */
       FINENV_CASHPENALTY_I32_FIELDS };
typedef struct finenv_cashpenalty_state {
    double  *f64;                 /* [FINENV_CASHPENALTY_F64_FIELDS + N][E]                */
    int32_t *i32;                 /* [FINENV_CASHPENALTY_I32_FIELDS][E]                    */
} finenv_cashpenalty_state;

typedef struct finenv_cashpenalty finenv_cashpenalty;

int  finenv_cashpenalty_create(const finenv_cashpenalty_config *cfg, finenv_cashpenalty **out);
void finenv_cashpenalty_destroy(finenv_cashpenalty *h);
const char *finenv_cashpenalty_last_error(const finenv_cashpenalty *h);
int  finenv_cashpenalty_obs_dim(const finenv_cashpenalty *h);
int  finenv_cashpenalty_bind(finenv_cashpenalty *h, const finenv_cashpenalty_panel *panel,
                             const finenv_cashpenalty_state *state);
int  finenv_cashpenalty_reset(finenv_cashpenalty *h, const uint8_t *mask, float *obs_out,
                              void *stream);
/* random_start (:134-138: `random.choice(range(int(len(dates) * 0.5)))`): with hi > 0 every reset
 * (explicit or inside step) draws its starting point on the device, uniformly in [0, hi), from a
 * counter-based generator keyed by (seed, env, episode) -- no host work per step; hi == 0 (default):
 * resets take FINENV_KI_NEXT_START.  Not bit-reproducible against Python's `random` by construction. */
int  finenv_cashpenalty_set_random_start(finenv_cashpenalty *h, int32_t hi, uint64_t seed);
int  finenv_cashpenalty_step(finenv_cashpenalty *h, const float *actions, float *obs,
                             float *reward, uint8_t *done, float *term_obs, int32_t auto_reset,
                             void *stream);
/* Harness log (the reference's account_information / transaction_memory lists, :149-154, :345-347,
 * read back by save_asset_memory / save_action_memory :382-409): when `audit` is non-NULL every
 * step writes one f64 row [FINENV_AUDIT_HEAD + N] per env:
 *   begin cash (:307/:312), asset value (:310), reward in f64 (:317), reason flags, then the N
 *   transactions that were (or, on a cash-shortage terminal step, would have been) applied.
 * Meant for the single-env facades (back-tests); NULL (default) = no log, no extra traffic. */
enum { FINENV_AUDIT_BEGIN_CASH = 0, FINENV_AUDIT_ASSET_VALUE, FINENV_AUDIT_REWARD,
       FINENV_AUDIT_FLAGS, FINENV_AUDIT_HEAD };
enum { FINENV_AUDIT_F_LAST_DATE = 1, FINENV_AUDIT_F_CASH_SHORTAGE = 2,
       FINENV_AUDIT_F_TURBULENCE = 4, FINENV_AUDIT_F_STOP_LOSS = 8,
       FINENV_AUDIT_F_LOW_PROFIT = 16, FINENV_AUDIT_F_HIGH_PROFIT = 32 };
int  finenv_cashpenalty_set_audit(finenv_cashpenalty *h, double *audit /* [E][HEAD+N] or NULL */);

/* =====================================================================================
 * StockTradingEnvStopLoss
 * (finrl/meta/env_stock_trading/env_stocktrading_stoploss.py:19-459): the cash-penalty env
 * plus an average-buy-price book per asset: positions are force-sold when
 * close < stoploss_penalty * avg_buy_price (and cash >= stoploss_penalty * initial, :353-357);
 * the reward adds a stop-loss penalty, a low-profit penalty and a profit bonus (:255-290).
 * Quirks kept (see oracle/stoploss_oracle.c): per-step reward uses the PREVIOUS step's
 * logged totals (:313 precedes :315-318); the turbulence sell-off goes through
 * (h*close)/close (:330,:345); patient mode still books the cancelled buys (:376 vs :418).
 *   actions [E][N] f32;  obs [E][D] f32, D = 1 + N + N*C  (same layout as the cash-penalty env)
 * ===================================================================================== */
#define FINENV_STOPLOSS_MAX_ASSETS 32

typedef struct finenv_stoploss_config {
    int32_t n_envs, n_assets, n_cols, n_days;
    int32_t discrete_actions;     /* :71, :333-343                                        */
    int32_t shares_increment;     /* :72                                                  */
    int32_t use_turbulence;       /* turbulence_threshold is not None, :327-331           */
    int32_t patient;              /* :373-378                                             */
    double  hmax;                 /* :70 (scalar)                                         */
    double  buy_cost_pct, sell_cost_pct, initial_amount, cash_penalty_proportion,
            turbulence_threshold;
    double  stoploss_penalty;     /* :73                                                  */
    double  min_profit_penalty;   /* 1 + profit_loss_ratio * (1 - stoploss_penalty), :101 */
} finenv_stoploss_config;

typedef struct finenv_stoploss_panel {
    const double *close;          /* [T][N] f64                                           */
    const float  *info;           /* [T][N*C] f32 date vectors, ticker-major (:167-180)   */
    const double *turb;           /* [T] f64 (may be NULL when use_turbulence == 0)       */
} finenv_stoploss_panel;

enum { FINENV_LF_COH = 0, FINENV_LF_TURBULENCE, FINENV_LF_SUM_TRADES, FINENV_LF_LOGGED_TOTAL,
       FINENV_LF_LOGGED_CASH, FINENV_LF_ACTUAL_NUM_TRADES, FINENV_STOPLOSS_F64_FIELDS };
/* after the scalar rows, six [N][E] f64 books in this order */
enum { FINENV_LV_HOLDINGS = 0, FINENV_LV_PREV_HOLDINGS, FINENV_LV_CLOSING_DIFF_AVG_BUY,
       FINENV_LV_PROFIT_SELL_DIFF_AVG_BUY, FINENV_LV_N_BUYS, FINENV_LV_AVG_BUY_PRICE,
       FINENV_STOPLOSS_BOOKS };
enum { FINENV_LI_DATE_INDEX = 0, FINENV_LI_START, FINENV_LI_EPISODE, FINENV_LI_NEXT_START,
       FINENV_STOPLOSS_I32_FIELDS };

typedef struct finenv_stoploss_state {
    double  *f64;                 /* [FINENV_STOPLOSS_F64_FIELDS + FINENV_STOPLOSS_BOOKS*N][E] */
    int32_t *i32;                 /* [FINENV_STOPLOSS_I32_FIELDS][E]                           */
} finenv_stoploss_state;

typedef struct finenv_stoploss finenv_stoploss;

int  finenv_stoploss_create(const finenv_stoploss_config *cfg, finenv_stoploss **out);
void finenv_stoploss_destroy(finenv_stoploss *h);
const char *finenv_stoploss_last_error(const finenv_stoploss *h);
int  finenv_stoploss_obs_dim(const finenv_stoploss *h);
int  finenv_stoploss_bind(finenv_stoploss *h, const finenv_stoploss_panel *panel,
                          const finenv_stoploss_state *state);
/* reset(), :134-165 (mask NULL = all envs; starting points from FINENV_LI_NEXT_START) */
int  finenv_stoploss_reset(finenv_stoploss *h, const uint8_t *mask, float *obs_out, void *stream);
/* step(), :292-442 (+ DummyVecEnv auto-reset when auto_reset != 0) */
/* as finenv_cashpenalty_set_random_start (:142-147) */
int  finenv_stoploss_set_random_start(finenv_stoploss *h, int32_t hi, uint64_t seed);
int  finenv_stoploss_step(finenv_stoploss *h, const float *actions, float *obs, float *reward,
                          uint8_t *done, float *term_obs, int32_t auto_reset, void *stream);
/* as finenv_cashpenalty_set_audit; flags additionally carry STOP_LOSS (:359-360) and
 * LOW_PROFIT / HIGH_PROFIT (:401-405), the reasons the reference logs for this env */
int  finenv_stoploss_set_audit(finenv_stoploss *h, double *audit /* [E][HEAD+N] or NULL */);

/* =====================================================================================
 * Risk precompute that feeds the panels.  Stateless; all buffers are
 * caller-owned device memory; launches go to `stream`; nothing synchronises.
 * Floating point (tolerances in tests/test_gpu_riskpre_parity.py): covariance sums run in day
 * order (NumPy: BLAS), the pseudo-inverse is applied through a Jacobi eigen-decomposition
 * (NumPy: LAPACK SVD) with NumPy's cutoff 1e-15 * largest eigenvalue.
 * Contract: complete panel (every asset on every day, close > 0, no NaN).
 * ===================================================================================== */
#define FINENV_RISKPRE_MAX_ASSETS 128

/* DataFrame.pct_change() of the close pivot (preprocessors.py:219-221):
 * returns[t][j] = close[t][j] / close[t-1][j] - 1, row 0 = NaN.  close, returns: [T][N] f64 */
int finenv_riskpre_returns(const double *close, double *returns, int32_t n_days,
                           int32_t n_assets, void *stream);
/* FeatureEngineer.calculate_turbulence (preprocessors.py:215-267): turbulence[t] for all T days
 * (0 for t < window and for the first two positive values, :247-257).  quad: [T] f64 scratch
 * (the unfiltered quadratic forms, :244-246).  Needs n_days >= window (the reference raises
 * otherwise, :260-266) and 2 <= n_assets <= FINENV_RISKPRE_MAX_ASSETS. */
int finenv_riskpre_turbulence(const double *returns, double *quad, double *turbulence,
                              int32_t n_days, int32_t n_assets, int32_t window, void *stream);
/* cov_list of the portfolio-allocation tutorial
 * (tutorials/2-Advance/FinRL_PortfolioAllocation_Explainable_DRL.py:160-172): cov_out[i-lookback]
 * = sample covariance of the `lookback` returns ending at day i inclusive, for i in
 * [lookback, T).  cov_out: [T-lookback][N][N] f64. */
int finenv_riskpre_rolling_cov(const double *returns, double *cov_out, int32_t n_days,
                               int32_t n_assets, int32_t lookback, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FINENV_H */
