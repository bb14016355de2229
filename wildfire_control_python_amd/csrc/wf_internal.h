// wf_internal.h - shared between the C-ABI host code (wf_abi.cpp) and the gfx950
// kernels (wf_kernels.hip).  Not part of the public boundary (include/wildfire.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wildfire.h"

typedef unsigned long long u64;

// Agent spawn table: circle_points(r) of reference Simulation/utility.py:8-52 for
// r = 1,2,3 (8 / 12 / 16 points), offsets relative to the fire location.
struct WfSpawn {
    int8_t dx[3][16];
    int8_t dy[3][16];
    int32_t n[3];
    int32_t pad;
};

// Rarely needed parameters (reset law, spawn table).  Lives in device memory and is
// read only inside the rare branches of the kernels, so that it costs no scalar
// registers on the hot path.
struct WfCold {
    wf_params p;
    WfSpawn spawn;
    // wind (wf_params.wind_mode != 0): heat addend [speed index][(wy+1)*4 + (wx+1)][direction k], computed on the host
    // with the reference's own float64 expression (environment.py:260-287)
    double wind_c[4][16][4];
};

enum { WF_HOT_CONTAINMENT = 1, WF_HOT_AUTO_RESET = 2, WF_HOT_DIG_TOGGLE = 4, WF_HOT_RIVERS = 8, WF_HOT_WIND = 16 };
// The two options that are off in the reference's default configuration (constants.py:44,46) live in separate
// instantiations of the step kernels (template parameter RARE), so that they cost the default path neither
// registers nor instructions.
#define WF_HOT_RARE (WF_HOT_DIG_TOGGLE | WF_HOT_RIVERS)

// Everything a step / reset launch needs on its hot path, passed by value (kernarg segment).
struct WfArgs {
    uint32_t *planes;        // (batch, H*W) one dword per cell
    uint32_t *planes_out;    // tiled path: ping-pong target (== planes for the wave path)
    wf_env *envs;            // (batch)
    u64 *reach;              // (batch, 4) border-reachability bitboards (wave path), pointer of this launch's env 0
    u64 *reach_rows;         // rows-layout boards of the whole CONTEXT (indexed by context env index / EPW)
    const int8_t *actions;   // (n_steps, batch) or nullptr
    double *reward;          // (n_steps, batch) or nullptr
    uint8_t *done;           // (n_steps, batch) or nullptr
    const uint8_t *mask;     // reset: per-env mask or nullptr
    const WfCold *cold;      // device copy of the rarely needed parameters
    int8_t *obs;             // optional fused observation output (batch, W, H, 3) of the final state
    uint32_t *wind_cnt;      // wind only: (batch, H*W) per-direction heat counts, one byte per direction
    int32_t batch, H, W, n_steps;
    u64 update_mask;
    u64 reset_seed;          // wf_reset: explicit seed (auto_reset uses cold->p.seed)
    int32_t ignite_count;
    int32_t hot_flags;       // WF_HOT_*
    int32_t group;           // single-step group kernel: environments per wave
    uint32_t env_base;       // index of this launch's env 0 inside the context (split launches)
    int32_t step_stride;     // environments in the whole context = row length of the (n_steps, batch) action / reward / done arrays
    int32_t sweep_rev;       // big-grid families: this step walks the batch from its far end (see WfKnobs::sweep_alt)
    double default_reward, death_penalty, contained_bonus;
};

// Experiment / test switches, read from the WF_* environment variables ONCE per context (wf_create; wf_reload_env
// re-reads them) so that the launch path never calls getenv().
struct WfKnobs {
    int group;       // WF_GROUP: cap on environments per wave of the single-step kernel (0 = automatic)
    int split_min;   // WF_SPLIT_MIN: smallest batch stepped as concurrent part launches (0 = automatic)
    int no_split;    // WF_NO_SPLIT: one launch per step
    int n_parts;     // WF_SPLIT: launches per step of a large batch (0: automatic; wf_create stores the number it settled on)
    int shared;      // WF_SHARED=N: workgroup-shared agent phases (N waves per workgroup; 1 = automatic)
    int rows;        // WF_ROWS: keep the rows layout for group overrides beyond 16
    int no_rows;     // WF_NO_ROWS: 4-cells-per-lane group layout only
    int no_multi;    // WF_NO_MULTI: an n-step call of a large batch = n single-step launches
    int family;       // WF_FAMILY=cells (1) / tiled (2): kernel family override, honoured by wf_create only
    int cells_legacy; // WF_CELLS_LEGACY: generic family as the five-launch thread-per-cell sequence for every batch
    int multi_small; // WF_MULTI_SMALL: n-step calls of batches that are not split also use the rows kernel's step loop
    int host_stage_mode; // WF_HOST_STAGE_MODE: 1 actions through a device mirror, 2 reward / done, 3 both (default of the experiment)
    int ticks_split;     // WF_TICKS_SPLIT (default 2048): smallest batch whose n-step wf_step_ticks calls go out as several launches on the context's streams (0: never)
    int split_explicit;  // WF_SPLIT was given: every split call uses exactly that many parts
    int sweep_alt;       // WF_SWEEP_ALT (default 1): the tiled step kernels walk the batch forwards and backwards in turn, so that a step
                         // starts on the bytes the step before touched last (still in the Infinity Cache); 0 = always forwards
    int fused_legacy;    // WF_FUSED_LEGACY: n-step calls of one environment per wave on wf_step_wave (unpacked cells, agent logic on the vector unit) instead of wf_step_ticks
    int flag_wait;       // WF_FLAG_WAIT: synchronous host-buffer calls wait on a flag a one-lane kernel writes to pinned memory (us of spinning; 0 = hipStreamSynchronize)
    int host_stage_min;  // WF_HOST_STAGE_MIN: host-buffer calls with at least this many actions go through device mirrors (0 = never: measured slower)
    int no_resident; // unless WF_RESIDENT is set: the rows kernel's step loop goes through L2, not through an LDS mirror
    int graph;       // WF_GRAPH: wave family, own streams: every distinct wf_step*_device call is captured into a HIP graph and replayed
    int tiled_rows;  // WF_TILED_ROWS: rows per strip of the one-launch tiled step (default: wf_tiled_solo_rows)
    int no_solo_loop; // WF_NO_SOLO_LOOP: n-step calls of the one-launch tiled step as n launches, never one launch with the step loop
    int solo_loop;    // WF_SOLO_LOOP: ... always the step loop (default: for jobs of fewer than WF_TILED_SOLO_WAVES waves, wf_abi.cpp: solo_loop_serves)
    int no_solo;     // WF_NO_SOLO: tiled grids one workgroup covers step as the larger ones do (wf_tiled_prologue -> wf_step_tiled), not in one launch
    int no_onegen;   // WF_NO_ONEGEN: never the one-generation shape of a single step (4 environments per wave, shared agent phases, 3 parts; wf_abi.cpp: single_step_shape)
    int no_prio;     // WF_NO_PRIO: never set WF_HOT_PROGRESS_PRIO
    int whole_batch; // per call (not an environment switch): this launch steps the whole batch alone (no part launches beside it)
    int multi_rows;  // per call (not an environment switch): n_steps > 1 is served by the rows kernel's step loop
};
void wf_read_knobs(WfKnobs *k);

// kernel families (wf_kernels.hip)
enum WfFamily { WF_FAM_NONE = 0, WF_FAM_WAVE = 1, WF_FAM_TILED = 2, WF_FAM_CELLS = 3 };

// Returns the family able to serve (H, W), and its name.
WfFamily wf_pick_family(int H, int W);
bool wf_tiled_serves(int H, int W);        // the tiled family can run this shape (WF_FAMILY=tiled forces it below its size threshold)
#define WF_TILED_MIN_CELLS 575             // below 24x24 the generic family is the faster one (wf_pick_family)
const char *wf_family_name(WfFamily f, int H, int W);

// Single-step group kernel: environments per wave, sized so that every resident wave slot of
// the chip (32 per CU) gets one group of a `batch`-environment step.
static inline int wf_pick_group(int cells, int num_cus, long long batch)
{
    int G = 1, gmax = 1;
    while (gmax * 2 <= cells / 4 && gmax < 32) gmax *= 2;        // env-lane k must be a cell-owning lane
    while (G < gmax && (long long)G * num_cus * 32 < batch) G *= 2;
    return G;
}

// The same, knowing the grid shape: shapes served by the rows-layout kernel keep G <= 8 - more, shorter waves than
// resident slots beat longer waves at every batch from 131 072 up (75 vs 69 % of roofline at 131 072 x 14x14,
// 80 vs 75 % at 262 144; a wash once the state outgrows the memory-side cache).
int wf_pick_group_for(int W, int H, int num_cus, long long batch, const WfKnobs &k);
int wf_rows_epw(int W, int H, const WfKnobs &k);        // environments per iteration of the shape's rows-layout kernel (0: none)
int wf_split_align(int W, int H, const WfKnobs &k);     // split launches cut the batch at multiples of this

// Launchers; all asynchronous on `s`.  Return hipSuccess or the launch error.
const char *wf_single_step_kernel(WfFamily fam, const WfArgs &a, const WfKnobs &k, int num_cus);
// n_steps > 1 served by ONE launch of the rows kernel (decided ONCE per call from the whole call's shape; the
// verdict travels to the part launches in WfKnobs::multi_rows)
bool wf_multi_step_in_rows(WfFamily fam, const WfArgs &a, const WfKnobs &k, int num_cus);
const char *wf_multi_step_kernel(const WfArgs &a, const WfKnobs &k, int num_cus);
hipError_t wf_launch_step(WfFamily fam, const WfArgs &a, const WfKnobs &k, hipStream_t s, int num_cus);
hipError_t wf_launch_reset(WfFamily fam, const WfArgs &a, hipStream_t s, int num_cus);
// the n-step kernels of one environment per wave (wf_step_ticks; wf_kernels.hip built with -DWF_TU=1)
hipError_t wf_launch_nstep_wave(const WfArgs &a, const WfKnobs &k, hipStream_t s, int blocks);
// big-grid families (wf_tiled.hip, wf_generic.hip): one step = pre / (auto-reset) / tiles / post; planes -> planes_out
// halo != nullptr (tiled family): the step runs IN PLACE (planes_out == planes); halo = u64[batch][gx][8] scratch
// generic family's part of wf_launch_step_big (wf_generic.hip)
hipError_t wf_launch_step_generic(const WfArgs &a, uint8_t *mode, uint32_t *acc, u64 auto_seed, hipStream_t s, bool per_env);
hipError_t wf_launch_step_big(WfFamily fam, const WfArgs &a, uint8_t *mode, uint32_t *acc, u64 *halo, u64 auto_seed, hipStream_t s, bool per_env,
                              int solo_rows);
// rows per strip of the tiled family's one-launch step of this shape (0: the shape needs several workgroups per environment)
int wf_tiled_solo_rows(int H, int W, int want, int batch);
int wf_tiled_solo_waves(int H, int W, int rows);   // waves of such a workgroup
bool wf_tiled_solo_fits(int H, int W, int rows, bool containment, bool obs);   // ... and such a launch's LDS fits (else: the two-launch form)
#define WF_SOLO_MAX_WAVES 16          // waves of a SOLO workgroup
#define WF_TILED_MIN_ROWS 8           // ... never shorter than this for the sake of more waves
#define WF_TILED_SOLO_WAVES 8192      // ... which a job with fewer waves than this (8 per SIMD) gets
#define WF_RIVER_MAX_H 2048       // big-grid families: rows of a grid whose river fits the reset kernel's LDS row table
#define WF_CELLS_FLOOD_MAX 60000   // generic family: cells of a grid whose reachability byte board fits the flood kernel's LDS
// generic family: wf_big_pre -> wf_step_env instead of the five-launch sequence (decided for the context's whole batch)
bool wf_cells_per_env_kernel(int batch, int H, int W, const WfKnobs &k);
int wf_tiled_groups(int H, int W);   // workgroups of wf_step_tiled per environment
bool wf_tiled_fused_obs(int H, int W);   // wf_step_tiled writes World.get_state() itself (a.obs != nullptr) for this shape
// sel == nullptr: all envs; sel_value >= 0: envs with sel[e] == sel_value; sel_value < 0: envs with sel[e] != 0
hipError_t wf_launch_reset_big(const WfArgs &a, const uint8_t *sel, int sel_value, u64 seed, hipStream_t s);
hipError_t wf_launch_invalidate_reach(wf_env *envs, int batch, hipStream_t s);
hipError_t wf_launch_copy_words(const void *src, void *dst, size_t bytes, hipStream_t s);   // bytes: a multiple of 4
hipError_t wf_launch_flag(uint32_t *flag, uint32_t value, hipStream_t s);
hipError_t wf_launch_copy_words2(const void *src0, void *dst0, size_t bytes0, const void *src1, void *dst1, size_t bytes1, hipStream_t s);
hipError_t wf_launch_get_state(const uint32_t *planes, int8_t *obs, int batch, int H, int W,
                               hipStream_t s);
