/*
 * wildfire.h - C-ABI of the MI355X-native batched wildfire cellular-automaton step.
 *
 * Drop-in boundary for ONE hot path of dashdeckers/Wildfire-Control-Python: the
 * per-cell fire-spread update and the step()/reset()/get_state() surface around it
 *     Simulation/forest_fire.py:30-49   ForestFire.step
 *     Simulation/forest_fire.py:52-54   ForestFire.reset
 *     Simulation/forest_fire.py:85-106  ForestFire.update (the CA tick)
 *     Simulation/environment.py:99-171  Agent.move / dig / is_dead
 *     Simulation/environment.py:186-402 World.reset / set_fire_to / reduce_fuel /
 *                                       apply_heat_from_to / get_reward / get_state
 *
 * FFI convention = the reference's only existing FFI (pyastar/pyastar.py:7-22 binding
 * pyastar/astar.cpp:41-47): a ctypes-loaded shared library, extern "C", plain
 * pointers and sizes, caller-owned host buffers (C-contiguous numpy arrays), no
 * exceptions across the boundary.  Differences, all forced by the device:
 * state lives in HBM inside an opaque context (one context per GPU; a context is
 * not thread-safe, distinct contexts are independent), every entry point returns
 * 0 or a negative wf_status (never exit()/abort()), and each host-buffer entry
 * point has a *_device twin that takes device pointers and is asynchronous on the
 * context's HIP stream (for learners that keep tensors on the GPU).
 *
 * Boundary data formats
 *   planes  int8 (batch, H, W, 4) C-contiguous; H <-> y, W <-> x (the reference
 *           indexes env[x, y, layer], environment.py:38-50; the transpose happens
 *           only in wf_get_state).  One 32-bit word per cell:
 *             plane 0 type   0 grass 1 fire 2 burnt 3 dirt 4 water (utility.py:128-140);
 *                          5 / 6 only with allow_dig_toggle, 13 only with make_rivers: see WF_DIRT_BURNING below
 *             plane 1 fuel   (utility.py:99; reduce_fuel environment.py:297-307)
 *             plane 2 heat   number of heat quanta received = temp / (wind_speed*heat)
 *                            (environment.py:286-290); WF_HEAT_FORCED marks
 *                            temp = threshold+1 written by set_fire_to (:236-237)
 *             plane 3 agent  agent_pos layer (utility.py:121): 0 or 1 (wf_upload refuses other values)
 *   counts  only with wind (wf_params.wind_mode != 0): uint8 (batch, H, W, 4), see wf_wind_counts_upload
 *   envs    wf_env[batch]: agent and world scalars (Agent.x/.y/.dead/.digging,
 *           World.RUNNING/.fire_at_border, "border deque exhausted")
 *   actions int8: 0..3 = N,S,E,W (environment.py:163-171); 4 = dig toggle when
 *           allow_dig_toggle (forest_fire.py:35-36); anything else = no-op
 *   reward  float64 (exactly the Python float get_reward returns, environment.py:342-390)
 *   done    uint8 = not World.RUNNING (forest_fire.py:48)
 *   obs     int8 (batch, W, H, 3) indexed [b][x][y][k] = World.get_state()
 *           (environment.py:399-402): agent_pos, type==fire, fire_mobility!=inf
 */
#ifndef WILDFIRE_H
#define WILDFIRE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WF_ABI_VERSION 2

enum { WF_P_TYPE = 0, WF_P_FUEL = 1, WF_P_HEAT = 2, WF_P_AGENT = 3 };
enum { WF_GRASS = 0, WF_FIRE = 1, WF_BURNT = 2, WF_DIRT = 3, WF_WATER = 4,
       /* Two states only allow_dig_toggle can produce: toggle_digging() on a BURNING cell runs dig()
        * (environment.py:123-138), which retypes the cell to dirt and blocks it (fire_mobility inf) but leaves
        * it in World.burning_cells: it keeps losing fuel and heating its neighbours, shows as dirt (is_burning()
        * false), and when its fuel is gone reduce_fuel() (:297-307) retypes it to burnt, still blocked.
        * "Burning" in a context created with allow_dig_toggle is (type & 3) == 1, i.e. WF_FIRE or WF_DIRT_BURNING;
        * contexts without it run kernels that know neither code (do not upload them there, and do not switch the
        * option off with wf_set_params while such cells exist). */
       WF_DIRT_BURNING = 5,     /* reference: type dirt, member of burning_cells, fire_mobility inf */
       WF_BURNT_BLOCKED = 6,    /* reference: type burnt, fire_mobility inf (dig() turns it into WF_DIRT) */
       /* One state only make_rivers can produce: reset_map may draw its river THROUGH the fire cell
        * (environment.py:69-95; 8-16 % of the resets do) - that cell is then type water with fire_mobility inf, and
        * set_fire_to() (:233-246) retypes it to fire.  It burns and shows as fire (is_burning() true: an agent that
        * steps on it dies; observation plane 1 = 1), blocks the containment search like dirt (plane 2 = 0; the search
        * still starts FROM it), and burnt out it is WF_BURNT_BLOCKED.  Accepted by contexts created with make_rivers;
        * "burning" there is (type & 3) == 1. */
       WF_FIRE_BLOCKED = 13 };  /* reference: type fire, member of burning_cells, fire_mobility inf */
#define WF_HEAT_FORCED 25
#define WF_HEAT_IGNITED 24   /* wind_mode != 0 only: plane 2 of a cell that accumulated heat ignited */

/* Wind (environment.py:187-193, 260-287).  wf_env.wind = (wx+1) | (wy+1) << 2 | speed index << 4: the wind vector
 * (wx, wy) in {-1,0,1}^2 and an index into wf_params.wind_speed[]. */
#define WF_WIND_CODE(wx, wy, si) ((unsigned)((wx) + 1) | (unsigned)((wy) + 1) << 2 | (unsigned)(si) << 4)
enum { WF_WIND_CALM = 0,     /* wind vector (0,0) for every environment: the integer rule (ignite_count)          */
       WF_WIND_FIXED = 1,    /* METADATA['wind'] = [speed, (wx, wy)]: every reset writes wf_params.wind_code        */
       WF_WIND_RANDOM = 2 }; /* METADATA['wind'] = "random" (environment.py:188-190): every reset draws the code    */

/* wf_env.flags */
enum {
    WF_F_ALIVE       = 1,    /* World.agents non-empty                  environment.py:206-208 */
    WF_F_VISIBLE     = 2,    /* agent_pos plane shows the agent (a rejected move blanks it, :142) */
    WF_F_DIGGING     = 4,    /* Agent.digging                           environment.py:112     */
    WF_F_DEAD        = 8,    /* Agent.dead set by move(), consumed by update()  :154-155, :116-120 */
    WF_F_RUNNING     = 16,   /* World.RUNNING                           forest_fire.py:105-106 */
    WF_F_BORDER      = 32,   /* World.fire_at_border                    environment.py:245-246 */
    WF_F_CONTAINED   = 64,   /* containment bonus already paid (border deque empty, :361-367)  */
    WF_F_REACH_VALID = 128,  /* internal: the context's border-reachability bitboard is current */
    WF_F_REACH_ROWS  = 256   /* internal: the same, for the rows-layout copy of the bitboard      */
};

typedef struct wf_env {
    uint16_t x, y;       /* agent position                                   */
    uint16_t flags;      /* WF_F_*                                           */
    uint16_t wind;       /* WF_WIND_CODE of this environment (wind_mode != 0)  */
    uint32_t episode;    /* resets seen by this environment (keys the stream) */
    uint32_t tick;       /* update() calls since the last reset               */
} wf_env;

typedef struct wf_params {
    /* rule (integer restatement of environment.py:278-294, valid for wind vector (0,0)):
     * a grass cell ignites when it has received >= ignite_count heat quanta */
    int32_t ignite_count;      /* 19 for wind 0.54, heat 0.3, threshold 3 (constants.py:39, utility.py:98-100) */
    int32_t allow_dig_toggle;  /* constants.py:46 */
    int32_t containment;       /* 1: evaluate the "fire contained" test of get_reward (environment.py:345-377) */
    int32_t auto_reset;        /* 1: a step on a finished env resets it instead (reward 0, done 0) */
    double  death_penalty;     /* constants.py:32 */
    double  contained_bonus;   /* constants.py:33 */
    double  default_reward;    /* constants.py:34 */
    /* synthetic reset law for wf_reset / auto_reset (reference default: 20, 20, 0) */
    int32_t  fuel_lo, fuel_hi; /* fuel ~ U{lo..hi} per cell                            */
    uint32_t ignite_ppm;       /* extra pre-ignited grass cells per million             */
    int32_t  make_rivers;      /* constants.py:44: reset draws the river of reset_map (environment.py:69-95) from the
                                * counter-based stream (same law, rivers through the fire cell included:
                                * WF_FIRE_BLOCKED)  */
    uint64_t seed;             /* stream seed for auto_reset                            */
    uint64_t env_id0;          /* global id of env 0 of this context (multi-GPU shards) */
    /* Wind with a direction (fixed at wf_create: WF_WIND_CALM cannot be switched to or from later).  The heat addend
     * then depends on the direction burning cell -> heated cell, wind_speed * heat * (angle + 1)^-1
     * (environment.py:260-287), a cell ignites when the float64 sum of its addends exceeds `threshold`, and the
     * state gains a per-direction count of addends for every grass cell (wf_wind_counts_*); plane 2 then only
     * records how a cell was ignited (0 / WF_HEAT_FORCED / WF_HEAT_IGNITED).  Served by a wind instantiation of the
     * register-resident group-layout kernel for the shapes the wave family serves (10x10, 14x14, ...), by the generic
     * kernel family for every other grid shape. */
    int32_t wind_mode;         /* WF_WIND_*                                             */
    int32_t wind_code;         /* WF_WIND_FIXED: the code of every environment          */
    double  wind_speed[4];     /* speeds a code's index selects; {0, 0.7, 0.85} for "random" (environment.py:189) */
    double  heat, threshold;   /* grass['heat'], grass['threshold'] (utility.py:98,100): 0.3, 3 */
} wf_params;

typedef enum wf_status {
    WF_OK = 0,
    WF_ERR_ARG = -1,          /* NULL / out-of-range argument                          */
    WF_ERR_UNSUPPORTED = -2,  /* grid shape or option not supported by this build      */
    WF_ERR_NO_DEVICE = -3,    /* no usable HIP device                                  */
    WF_ERR_HIP = -4,          /* a HIP call failed; wf_last_hip_error() has the code   */
    WF_ERR_NOMEM = -5
} wf_status;

typedef struct wf_ctx wf_ctx;

int  wf_abi_version(void);
void wf_default_params(wf_params *p);
const char *wf_strerror(int status);
int  wf_last_hip_error(const wf_ctx *ctx);

/* Context: device state for `batch` environments of H x W cells on HIP device `device`.
 * replaces: World.__init__ / create_map (environment.py:38-50, 175-183) */
int wf_create(wf_ctx **out, int device, int batch, int H, int W, const wf_params *p);
int wf_destroy(wf_ctx *ctx);
int wf_set_params(wf_ctx *ctx, const wf_params *p);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own. */
int wf_set_stream(wf_ctx *ctx, void *hip_stream);

/* Host <-> device state exchange (synchronous).  Either pointer may be NULL to skip it.  A record whose agent position
 * lies outside the grid, or a cell type outside 0..4 (allow_dig_toggle: also 5, 6; make_rivers: also 6, 13), is refused
 * (WF_ERR_ARG, nothing uploaded).  With wind, planes and heat counts form ONE state: uploading planes zeroes the counts
 * (a new grid has received no heat); upload the counts AFTER the planes when restoring a snapshot.
 * replaces: direct numpy access to World.env (environment.py:38-50) */
int wf_upload(wf_ctx *ctx, const int8_t *planes, const wf_env *envs);
int wf_download(wf_ctx *ctx, int8_t *planes, wf_env *envs);

/* Wind only (WF_ERR_UNSUPPORTED otherwise): counts uint8 (batch, H, W, 4) = how many heat addends a GRASS cell has
 * received per direction k of the vector heated cell - burning cell, k = 0 (0,-1), 1 (0,+1), 2 (+1,0), 3 (-1,0) (the
 * order of _direction_to_coords, environment.py:163-171).  The reference's temp layer is the float64 sum of those
 * addends (environment.py:290).  Download reports zero for every cell that is not grass. */
int wf_wind_counts_upload(wf_ctx *ctx, const uint8_t *counts);
int wf_wind_counts_download(wf_ctx *ctx, uint8_t *counts);

/* World.reset() (environment.py:186-212) for envs with mask[i] != 0 (NULL = all), spawn
 * drawn from the counter-based stream keyed by (seed, env_id0 + i, episode).  Grids smaller than 7 cells in either
 * direction cannot hold the spawn circle (the reference asserts 10x10, utility.py:68): WF_ERR_UNSUPPORTED, also for
 * auto_reset; such grids can still be uploaded and stepped. */
int wf_reset(wf_ctx *ctx, const uint8_t *mask, uint64_t seed);
int wf_reset_device(wf_ctx *ctx, const uint8_t *d_mask, uint64_t seed);

/* n_steps x ForestFire.step() (forest_fire.py:30-49).
 *   actions int8 (n_steps, batch), NULL = no-op;  reward float64 (n_steps, batch);
 *   done uint8 (n_steps, batch); reward/done may be NULL.
 * update() runs every params-independent a_speed-th step: the caller passes
 * update_mask, bit s = run update() in step s (all ones for a_speed = 1; n_steps <= 64).
 * wf_step is synchronous with host buffers; wf_step_device is asynchronous with device buffers.
 * Ordering of the *_device calls: with a caller stream (wf_set_stream) everything is enqueued on that
 * stream, in order.  With the context's OWN streams (the default) the calls of one context are ordered
 * among themselves, the device buffers handed in must be complete when the call is made, and results
 * are complete after wf_sync / wf_timer_stop (a large single step is issued as two - big grids: three - concurrent
 * launches over shares of the batch on internal streams; every other call joins them first). */
int wf_step(wf_ctx *ctx, const int8_t *actions, int n_steps, uint64_t update_mask,
            double *reward, uint8_t *done);
int wf_step_device(wf_ctx *ctx, const int8_t *d_actions, int n_steps, uint64_t update_mask,
                   double *d_reward, uint8_t *d_done);

/* Pinned host buffers the library lends for wf_step / wf_step_obs calls of up to n_steps steps: actions int8
 * (n_steps, batch), reward float64 (n_steps, batch), done uint8 (n_steps, batch).  A call that is handed these very
 * pointers reads and writes them in place - no host copy in either direction (at 65 536 environments the copies of a
 * step's reward and done are 590 KB, about as long as the step itself).  The pointers stay valid until a call with a
 * larger n_steps, or wf_destroy; their contents until the next wf_step*.  Any of the three may be NULL. */
int wf_host_staging(wf_ctx *ctx, int n_steps, int8_t **actions, double **reward, uint8_t **done);

/* The same steps, and World.get_state() of the resulting state written in the same launch
 * (the grid is in registers anyway): obs int8 (batch, W, H, 3).  This is what
 * `[get_state(), get_reward(), not RUNNING, {}]` of forest_fire.py:46-49 costs per step. */
int wf_step_obs(wf_ctx *ctx, const int8_t *actions, int n_steps, uint64_t update_mask,
                double *reward, uint8_t *done, int8_t *obs);
int wf_step_obs_device(wf_ctx *ctx, const int8_t *d_actions, int n_steps, uint64_t update_mask,
                       double *d_reward, uint8_t *d_done, int8_t *d_obs);

/* wf_step followed by wf_download of the resulting state under ONE stream synchronisation (planes / envs as in
 * wf_download; either may be NULL).  What a host-side view of a small context wants after every step - the
 * single-environment adapter, whose World.env / agents / burning_cells mirror the device state
 * (forest_fire.py:46-49 returns get_state(), which the reference computes from World.env). */
int wf_step_download(wf_ctx *ctx, const int8_t *actions, int n_steps, uint64_t update_mask,
                     double *reward, uint8_t *done, int8_t *planes, wf_env *envs);

/* World.get_state() (environment.py:399-402) for the whole batch. */
int wf_get_state(wf_ctx *ctx, int8_t *obs);
int wf_get_state_device(wf_ctx *ctx, int8_t *d_obs);

int wf_sync(wf_ctx *ctx);

/* Zero-copy access for device-side consumers; pointers stay valid until wf_destroy. */
void *wf_planes_device(wf_ctx *ctx);   /* int8 (batch,H,W,4) current state */
void *wf_envs_device(wf_ctx *ctx);     /* wf_env[batch]                     */

/* HIP-event timing on the context's stream (bench.py's roofline leg). */
int wf_timer_start(wf_ctx *ctx);
int wf_timer_stop(wf_ctx *ctx, float *elapsed_ms);   /* synchronises */

/* Introspection: name of the kernel family serving this context's grid shape; symbol of the kernel
 * behind a single step (n_steps == 1) of this context, and how many concurrent launches of it one
 * step is (large batches on the context's own streams are stepped as two or three launches over shares of the batch). */
const char *wf_kernel_name(const wf_ctx *ctx);
const char *wf_step_kernel_name(const wf_ctx *ctx);
int wf_step_launches(const wf_ctx *ctx);
int wf_call_launches(const wf_ctx *ctx, int n_steps);   /* kernel launches behind one wf_step*(n_steps) call WITH actions */
const char *wf_call_kernel_name(const wf_ctx *ctx, int n_steps);   /* the kernel behind that call (an n-step call without
                                                         * actions of a large batch may run on wf_step_ticks where this one does not) */

/* Introspection without a context (touches no device): how `batch` grids of height x width would be stepped by default.
 * Returns the kernel family as wf_kernel_name names it ("wave", "tiled" or "cells" as a prefix is enough for callers:
 * 1 = wave family, 2 = tiled, 3 = generic).  For the tiled family also: *one_launch = 1 if a step is one launch of one
 * workgroup per environment (0: wf_tiled_prologue -> wf_step_tiled), *rows = rows per strip and *waves = waves per
 * workgroup of that form, *step_loop = 1 if an n-step call runs its steps inside one launch.  Any pointer may be NULL. */
int wf_step_plan(int height, int width, int batch, int *one_launch, int *rows, int *waves, int *step_loop);

/* Experiments and tests only: the WF_* environment switches (kernel variant, split on/off ...) are read once in
 * wf_create; this re-reads them for an existing context (synchronises; WF_SPLIT itself stays as created). */
int wf_reload_env(wf_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
