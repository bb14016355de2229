// wf_abi.cpp - host side of the C-ABI declared in include/wildfire.h.
//
// Owns the device state of a batch of environments (planes, env scalars,
// containment bitboards, staging buffers) and launches the gfx950 kernels of
// wf_kernels.hip on the context's HIP stream.  There is NO CPU fallback: without a
// usable HIP device wf_create fails with WF_ERR_NO_DEVICE.
#include <math.h>
#include <atomic>
#include <chrono>
#include <new>
#include <thread>
#include <vector>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wf_internal.h"

constexpr int WF_MAX_SIDE = 3;      // (more than four launches in flight was measured far slower: 4 096 x 10x10 x 16 ticks 13.3 us as 4 parts, 24.0 as 6)
constexpr size_t WF_ZC_OBS_MAX = (size_t)1 << 20;

struct wf_ctx {
    int device = 0;
    int batch = 0, H = 0, W = 0;
    int num_cus = 256;
    wf_params p;
    WfFamily fam = WF_FAM_NONE;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Split launches: with the context's OWN streams a single step is issued as two independent
    // halves of the batch on two streams, so that one half's latency-bound prologue overlaps the
    // other half's compute (environments never interact).  Joined by events before anything else.
    hipStream_t side[WF_MAX_SIDE] = {};
    hipEvent_t ev_side[WF_MAX_SIDE] = {}, ev_fork = nullptr;
    int n_parts = 2;                  // launches per step (WF_SPLIT overrides, 1 = never split)
    int n_side = 0;                   // side streams that exist (own streams: WF_MAX_SIDE; a call may use fewer than there are)
    int last_parts = 0;               // parts of the last split call: another number of parts maps environments to other streams
    int side_used = 0;                // side streams with work since the last join
    WfKnobs knobs = {};               // WF_* experiment switches, read once (wf_create / wf_reload_env)
    bool sync_call = false;           // inside a synchronous host-buffer call: one launch, no fork / join (see wf_step_obs)
    bool side_pending = false;        // work enqueued on `side` that `stream` has not waited for yet
    bool main_touched = true;         // non-step work enqueued on `stream` since the last fork
    // device state
    uint32_t *planes = nullptr;       // (batch, H*W) dwords
    uint32_t *planes_alt = nullptr;   // generic thread-per-cell family: ping-pong target
    u64 *d_halo = nullptr;            // tiled family (steps in place): masks of the rows between workgroups
    wf_env *envs = nullptr;
    u64 *reach = nullptr;
    uint32_t *wind_cnt = nullptr;     // wind only: per-direction heat counts, one dword per cell
    // staging for the host-buffer entry points (grown on demand): PINNED HOST memory the kernels read and write
    // directly over PCIe.  No DMA operation per array and step - at 20 us per step those, not the bytes, were the cost:
    // 16 against 49-119 us per wf_step of one environment, 19 against 56 at 4 096, 62 against 117 at 65 536.
    int8_t *d_actions = nullptr;
    int8_t *h_obs = nullptr;          // small observations the same way (up to WF_ZC_OBS_MAX bytes; larger ones by DMA)
    uint8_t *h_dl = nullptr;          // wf_download of a small context: planes + records through one copy kernel
    size_t h_dl_bytes = 0;
    uint8_t *d_mask = nullptr;
    int8_t *d_obs = nullptr;
    WfCold *d_cold = nullptr;         // device copy of {params, spawn table} for the kernels' rare paths
    uint8_t *d_mode = nullptr;        // big-grid families: per-env mode of the current step
    uint32_t *d_acc = nullptr;        // big-grid families: per-env reduction words
    int staged_steps = 0;
    // reward and done of the host-buffer entry points: ONE pinned block (p_out), lent to callers by wf_host_staging.
    // (v_actions / v_out: device mirrors of the WF_HOST_STAGE_MIN experiment, see wf_step_obs.)
    int8_t *v_actions = nullptr;
    uint8_t *v_out = nullptr;          // [n doubles of reward][n bytes of done]
    uint8_t *p_out = nullptr;          // pinned host block of the same layout
    uint32_t sweeps = 0;               // big-grid steps issued so far (WF_SWEEP_ALT: odd ones walk the batch backwards)
    uint32_t *p_flag = nullptr;        // pinned completion word of synchronous calls (WF_FLAG_WAIT)
    uint32_t flag_seq = 0;
    size_t v_count = 0;                // n the mirrors were sized for
    void *dbg_timing = nullptr;       // measurement builds (WF_WAVE_TIMING) only
    WfSpawn spawn;
    int last_hip = 0;
    // WF_GRAPH experiment: captured launch sequences, one per distinct argument set of wf_step*_device
    struct GraphEntry { const int8_t *actions; double *reward; uint8_t *done; int8_t *obs; int n_steps; uint64_t mask; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
};

namespace {

int fail(wf_ctx *c, hipError_t e)
{
    if (c) c->last_hip = (int)e;
    return e == hipErrorOutOfMemory ? WF_ERR_NOMEM : WF_ERR_HIP;
}

#define HIP_TRY(ctx, expr)                                 \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return fail((ctx), _e);      \
    } while (0)

// circle_points(r) around the origin, in the reference's generation order
// (midpoint circle algorithm, Simulation/utility.py:8-52).
int circle_points(int r, int8_t *dx, int8_t *dy)
{
    int n = 0;
    auto put = [&](int a, int b) { dx[n] = (int8_t)a; dy[n] = (int8_t)b; ++n; };
    int x = r, y = 0;
    put(x, y);
    if (r > 0) { put(-x, -y); put(y, -x); put(-y, x); }
    int P = 1 - r;
    while (x > y) {
        ++y;
        if (P <= 0) {
            P += 2 * y + 1;
        } else {
            --x;
            P += 2 * y - 2 * x + 1;
        }
        if (x < y) break;
        put(x, y); put(-x, y); put(x, -y); put(-x, -y);
        if (x != y) { put(y, x); put(-y, x); put(y, -x); put(-y, -x); }
    }
    return n;
}

bool params_ok(const wf_params *p)
{
    if (p->ignite_count < 1 || p->ignite_count >= WF_HEAT_FORCED) return false;
    if (p->fuel_lo < 1 || p->fuel_hi < p->fuel_lo || p->fuel_hi > 127) return false;
    if (p->ignite_ppm > 1000000u) return false;
    if (p->wind_mode < WF_WIND_CALM || p->wind_mode > WF_WIND_RANDOM) return false;
    if (p->wind_mode != WF_WIND_CALM) {
        const unsigned code = (unsigned)p->wind_code;
        if ((code & 3u) > 2u || ((code >> 2) & 3u) > 2u || code > 63u) return false;
        for (double v : p->wind_speed)
            if (!(v >= 0.0) || !isfinite(v)) return false;
        if (!(p->heat > 0.0) || !(p->threshold > 0.0) || !isfinite(p->heat) || !isfinite(p->threshold)) return false;
    }
    return true;
}

// Wind with a direction: the kernels decide `temp > threshold` from per-direction COUNTS of addends (one byte each)
// and the float64 sum of count x addend; the reference adds the addends one by one in the iteration order of a Python
// set.  Both decide alike as long as (a) no count a cell can reach before it ignites exceeds 255 and (b) no reachable
// sum comes within rounding distance of the threshold.  Checked here for every wind the context can meet (the fixed
// code, or the 27 winds "random" draws): parameters that fail are refused (WF_ERR_UNSUPPORTED), not approximated.
int wind_params_status(const wf_params *p)
{
    if (p->wind_mode == WF_WIND_CALM) return WF_OK;
    static const int dcx[4] = { 0, 0, 1, -1 }, dcy[4] = { -1, 1, 0, 0 };
    const double T = p->threshold, margin = 1e-9 * (T > 1.0 ? T : 1.0);
    for (int si = 0; si < (p->wind_mode == WF_WIND_RANDOM ? 3 : 1); ++si)
        for (int v = 0; v < (p->wind_mode == WF_WIND_RANDOM ? 9 : 1); ++v) {
            int wxi, wyi, s = si;
            if (p->wind_mode == WF_WIND_RANDOM) { wxi = v % 3 - 1; wyi = v / 3 - 1; }
            else { wxi = (p->wind_code & 3) - 1; wyi = ((p->wind_code >> 2) & 3) - 1; s = (p->wind_code >> 4) & 3; }
            const double speed = p->wind_speed[s];
            if (!(speed > 0.0)) continue;                    // no heat at all: nothing ever ignites
            // distinct addends and how many directions share each (an axis wind: 1 + 2 + 1, a diagonal one: 2 + 2)
            double cv[4];
            int mult[4], nv = 0;
            for (int k = 0; k < 4; ++k) {
                const double wx = wxi, wy = wyi, cx = dcx[k], cy = dcy[k];
                const double angle = fabs(atan2(wx * cy - wy * cx, wx * cx + wy * cy));
                const double ck = speed * p->heat * pow(angle + 1.0, -1.0);
                int j = 0;
                while (j < nv && cv[j] != ck) ++j;
                if (j == nv) { cv[nv] = ck; mult[nv] = 0; ++nv; }
                ++mult[j];
            }
            long long top[4] = { 0, 0, 0, 0 }, combos = 1;
            for (int j = 0; j < nv; ++j) {
                const double per_dir = floor(T / cv[j]) + 1.0;          // addends of this size until the sum exceeds T
                if (!(per_dir <= 255.0)) return WF_ERR_UNSUPPORTED;      // a count byte would saturate first
                top[j] = (long long)per_dir * mult[j];
                combos *= top[j] + 1;
            }
            if (combos > 50000000ll) return WF_ERR_UNSUPPORTED;
            for (long long n0 = 0; n0 <= top[0]; ++n0)
                for (long long n1 = 0; n1 <= top[1]; ++n1)
                    for (long long n2 = 0; n2 <= top[2]; ++n2)
                        for (long long n3 = 0; n3 <= top[3]; ++n3) {
                            const double sum = (double)n0 * cv[0] + (nv > 1 ? (double)n1 * cv[1] : 0.0) +
                                               (nv > 2 ? (double)n2 * cv[2] : 0.0) + (nv > 3 ? (double)n3 * cv[3] : 0.0);
                            if (fabs(sum - T) < margin) return WF_ERR_UNSUPPORTED;
                        }
        }
    return WF_OK;
}

// Every cell type of an upload is one this context's kernels know: bit t of `allowed`.
bool types_ok(const wf_ctx *c, const int8_t *planes)
{
    uint32_t allowed = (1u << WF_GRASS) | (1u << WF_FIRE) | (1u << WF_BURNT) | (1u << WF_DIRT) | (1u << WF_WATER);
    if (c->p.allow_dig_toggle) allowed |= (1u << WF_DIRT_BURNING) | (1u << WF_BURNT_BLOCKED);
    if (c->p.make_rivers) allowed |= (1u << WF_FIRE_BLOCKED) | (1u << WF_BURNT_BLOCKED);
    const size_t n = (size_t)c->batch * c->H * c->W;
    auto scan = [planes, allowed](size_t lo, size_t hi) -> uint32_t {
        uint32_t bad = 0;
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t t = (uint8_t)planes[4 * i];
            bad |= (t > 15u ? 1u : 0u) | (((allowed >> (t & 15u)) & 1u) ^ 1u);
            bad |= (uint8_t)planes[4 * i + 3] > 1u ? 1u : 0u;          // agent_pos is 0 or 1 (utility.py:121; the tiled family's observation tile holds it as a bit)
        }
        return bad;
    };
    const size_t per_thread = (size_t)1 << 22;
    if (n <= 2 * per_thread) return scan(0, n) == 0;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 2) nt = 2;
    if (nt > 16) nt = 16;
    std::atomic<uint32_t> bad(0);
    std::vector<std::thread> pool;
    const size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        if (lo >= hi) break;
        pool.emplace_back([&, lo, hi] { bad.fetch_or(scan(lo, hi)); });
    }
    for (auto &th : pool) th.join();
    return bad.load() == 0;
}

// World.reset() puts the agent on a circle of radius 1..3 around the centre (utility.py:66-78; the reference asserts
// grids of at least 10x10, utility.py:68): a grid must hold that circle.  Smaller grids can still be uploaded and stepped.
bool reset_ok(int H, int W) { return H >= 7 && W >= 7; }

// reset_map's river needs room for its margins (the reference asserts grids of at least 10x10, utility.py:68) and, in
// the big-grid families, a row table in LDS
bool rivers_ok(WfFamily fam, int H, int W)
{
    return H >= 8 && W >= 8 && (fam == WF_FAM_WAVE || H <= WF_RIVER_MAX_H);
}

WfArgs base_args(const wf_ctx *c)
{
    WfArgs a;
    memset(&a, 0, sizeof(a));
    a.planes = c->planes;
    a.planes_out = c->planes_alt ? c->planes_alt : c->planes;
    a.envs = c->envs;
    a.reach = c->reach;
    a.reach_rows = c->fam == WF_FAM_WAVE ? c->reach + (size_t)c->batch * 4 : nullptr;
    a.step_stride = c->batch;
    a.cold = c->d_cold;
    a.wind_cnt = c->wind_cnt;
    a.batch = c->batch;
    a.H = c->H;
    a.W = c->W;
    a.n_steps = 1;
    a.update_mask = ~0ull;
    a.ignite_count = c->p.ignite_count;
    a.hot_flags = (c->p.containment ? WF_HOT_CONTAINMENT : 0) | (c->p.auto_reset ? WF_HOT_AUTO_RESET : 0) |
                  (c->p.allow_dig_toggle ? WF_HOT_DIG_TOGGLE : 0) | (c->p.make_rivers ? WF_HOT_RIVERS : 0) |
                  (c->p.wind_mode != WF_WIND_CALM ? WF_HOT_WIND : 0);
    a.group = c->knobs.group;          // experiment knob: cap on environments per wave of the single-step kernel
#ifdef WF_WAVE_TIMING
    a.mask = static_cast<const uint8_t *>(c->dbg_timing);
#endif
    a.default_reward = c->p.default_reward;
    a.death_penalty = c->p.death_penalty;
    a.contained_bonus = c->p.contained_bonus;
    return a;
}

// make `stream` wait for everything issued on the side stream (no host sync)
int join_side(wf_ctx *c)
{
    if (c->side_pending) {         // recorded lazily: an event per step would serialise the side queues
        for (int i = 0; i < c->side_used; ++i) {
            HIP_TRY(c, hipEventRecord(c->ev_side[i], c->side[i]));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_side[i], 0));
        }
        c->side_pending = false;
        c->side_used = 0;
    }
    c->main_touched = true;
    return WF_OK;
}

#define JOIN(c)                              \
    do {                                     \
        int _rc = join_side(c);              \
        if (_rc != WF_OK) return _rc;        \
    } while (0)

// Start of a call that goes out as `parts` launches on the context's streams (main + parts - 1 side streams).  The sides
// must see earlier reset / upload / get_state work of `stream`; and a call with ANOTHER number of parts than the last one
// cuts the batch elsewhere, so its parts depend on what other streams did before: everything joins `stream` first.
int begin_split(wf_ctx *c, int parts)
{
    if (c->last_parts != 0 && c->last_parts != parts) JOIN(c);      // (sets main_touched: the fork below follows)
    if (c->main_touched) {
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
        for (int i = 0; i < parts - 1; ++i) HIP_TRY(c, hipStreamWaitEvent(c->side[i], c->ev_fork, 0));
        c->main_touched = false;
    }
    c->last_parts = parts;
    if (c->side_used < parts - 1) c->side_used = parts - 1;
    return WF_OK;
}

// A single step of a large batch on the context's OWN streams goes out as n_parts concurrent launches.
bool splits(const wf_ctx *c, int n_steps)
{
    const int min_batch = c->knobs.split_min > 0 ? c->knobs.split_min : 4096 * c->n_parts;   // 8 192 envs: 7.9 vs 9.6 us per step; 4 096: 7.4 vs 6.8
    return c->fam == WF_FAM_WAVE && c->own_stream && c->n_parts > 1 && n_steps == 1 && c->batch >= min_batch &&
           !c->knobs.no_split && !c->sync_call;
}

// the tiled family's one-launch step (wf_step_tiled<.., SOLO>): grids one of its workgroups covers
int tiled_solo(const wf_ctx *c)
{
    return c->fam == WF_FAM_TILED && !c->knobs.no_solo ? wf_tiled_solo_rows(c->H, c->W, c->knobs.tiled_rows, c->batch) : 0;
}

// n-step calls of the tiled family's one-launch step as ONE launch per share, the step loop inside the kernel
// (wf_step_tiled<.., SOLO, MULTI>).  For jobs of fewer than WF_TILED_SOLO_WAVES waves - about what the chip holds at
// once: the loop's workgroups stay where they are and pay no launch boundary per step (us per step, loop / n launches:
// 512 x 64x64 11.1 / 18.3, 2 048 x 64x64 14.9 / 20.5, 512 x 128x128 15.0 / 21.4, 256 x 256x256 26.3 / 34.5) - while a
// job of several generations of waves is better off re-dealt to the CUs every step (4 096 x 64x64 26.2 / 24.4,
// 16 384 x 32x32 47.8 / 41.6).
bool solo_loop_serves(const wf_ctx *c, int n_steps)
{
    const int rows = tiled_solo(c);
    if (n_steps < 2 || c->knobs.no_solo_loop || !wf_tiled_solo_fits(c->H, c->W, rows, c->p.containment != 0, false)) return false;
    return c->knobs.solo_loop || (long long)c->batch * wf_tiled_solo_waves(c->H, c->W, rows) < WF_TILED_SOLO_WAVES;
}

bool splits_big(const wf_ctx *c)
{
    return c->fam != WF_FAM_WAVE && c->own_stream && c->n_parts > 1 && c->batch >= 2 * c->n_parts &&
           (size_t)c->batch * c->H * c->W >= ((size_t)8 << 20) && !c->knobs.no_split && !c->sync_call;
}

// (re)publish the rarely needed parameters to the device
int push_cold(wf_ctx *c)
{
    WfCold h;
    memset(&h, 0, sizeof(h));
    h.p = c->p;
    h.spawn = c->spawn;
    if (c->p.wind_mode != WF_WIND_CALM) {
        // the addend of apply_heat_from_to per wind and direction, in the reference's own float64 expression:
        // angle = abs(atan2(wx*cy - wy*cx, wx*cx + wy*cy)) (environment.py:273), distance 1 (manhattan, radius 1),
        // heat = wind_speed * cell heat * (angle + distance)^-1 (:286-287)
        static const int dcx[4] = { 0, 0, 1, -1 }, dcy[4] = { -1, 1, 0, 0 };   // heated cell - burning cell: N, S, E, W
        for (int si = 0; si < 4; ++si)
            for (int v = 0; v < 16; ++v)
                for (int k = 0; k < 4; ++k) {
                    const double wx = (double)((v & 3) - 1), wy = (double)(((v >> 2) & 3) - 1);
                    const double cx = dcx[k], cy = dcy[k];
                    const double angle = fabs(atan2(wx * cy - wy * cx, wx * cx + wy * cy));
                    h.wind_c[si][v][k] = c->p.wind_speed[si] * c->p.heat * pow(angle + 1.0, -1.0);
                }
    }
    HIP_TRY(c, hipMemcpy(c->d_cold, &h, sizeof(h), hipMemcpyHostToDevice));
    return WF_OK;
}

// environments per part of a split launch
int split_per(const wf_ctx *c, int parts = 0)
{
    if (parts <= 0) parts = c->n_parts;
    const int al = wf_split_align(c->W, c->H, c->knobs);
    return ((c->batch + parts - 1) / parts + al - 1) / al * al;
}

// Parts of an n-step call served by wf_step_ticks (0: one launch).  Measured at 16 ticks per call, no actions
// (profiles/r04_ticks_split.txt; us per call as 1 / 2 / 4 launches): 2 048 x 10x10 12.1 / 10.7 / 12.0, 4 096 17.4 / 13.7 / 13.0,
// 8 192 28.3 / 22.0 / 21.1, 16 384 51.7 / 41.5 / 41.7; below 2 048 environments one launch is fastest.
int ticks_parts(const wf_ctx *c)
{
    if (c->knobs.ticks_split <= 0 || c->batch < c->knobs.ticks_split || !c->own_stream || c->n_parts <= 1 || c->knobs.no_split) return 0;
    if (c->knobs.n_parts > 0 && c->knobs.split_explicit) return c->n_parts;                 // WF_SPLIT given: that many
    const int most = c->n_side + 1;
    return c->batch >= 2 * c->knobs.ticks_split ? (most < 4 ? most : 4) : 2;
}

// Is an n_steps call of this (split) context served by the rows kernel's own step loop?  Probed with the FIRST part's
// shape (the largest one); every part then runs the same kernel.
bool multi_rows_call(const wf_ctx *c, int n_steps)
{
    WfArgs probe = base_args(c);
    probe.n_steps = n_steps;
    probe.batch = split_per(c);
    if (probe.group <= 0) probe.group = wf_pick_group_for(c->W, c->H, c->num_cus, c->batch, c->knobs);
    return wf_multi_step_in_rows(c->fam, probe, c->knobs, c->num_cus);
}

// The same question for a batch that is not split (experiment knob WF_MULTI_SMALL)
bool multi_rows_whole(const wf_ctx *c, int n_steps)
{
    WfArgs probe = base_args(c);
    probe.n_steps = n_steps;
    return wf_multi_step_in_rows(c->fam, probe, c->knobs, c->num_cus);
}

// An n-step call of a SPLIT context as ONE launch of wf_step_ticks (one environment per wave, grid in registers across
// the steps) instead of the rows kernel's step loop (rows_loop) or n single-step launches per part?  Measured at 16 steps
// per call (tools/nstep_kernels.py, profiles/r03_nstep_kernels.txt; us per call, ticks against the alternative): without
// actions 8 192 envs 29-31 against 53-113, 16 384 envs 54-57 against 69-129, 65 536 envs 200-211 against 129-226 where a
// step loop exists and 206 against 336 where none does (16x16); with actions (the agent's gather and its scalar chain double
// the step) 8 192 envs 61-63 against 56-68, 16 384 envs 116-119 against 76-95 - but 62 / 117 against 108 / 130 at 16x16.
bool ticks_serves_split(const wf_ctx *c, bool has_actions, bool rows_loop)
{
    if (c->knobs.no_multi || c->knobs.multi_small) return false;      // (experiment switches ask for the other kernels)
    if (!has_actions) return !rows_loop || c->batch <= 16384;
    return !rows_loop && c->batch <= 16384;
}

int ensure_staging(wf_ctx *c, int n_steps)
{
    if (n_steps <= c->staged_steps) return WF_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->d_actions) (void)hipHostFree(c->d_actions);
    c->d_actions = nullptr;
    c->staged_steps = 0;
    const size_t n = (size_t)n_steps * (size_t)c->batch;
    HIP_TRY(c, hipHostMalloc((void **)&c->d_actions, (n + 15) & ~(size_t)15, hipHostMallocDefault));   // (copied in whole dwords)
    c->staged_steps = n_steps;
    return WF_OK;
}

// the pinned reward / done block for calls of n actions (and, only for the WF_HOST_STAGE_MIN experiment, device mirrors)
int ensure_mirrors(wf_ctx *c, size_t n)
{
    const bool want_dev = c->knobs.host_stage_min > 0;
    if (n <= c->v_count && (!want_dev || c->v_out)) return WF_OK;
    if (n < c->v_count) n = c->v_count;
    if (c->v_actions) (void)hipFree(c->v_actions);
    if (c->v_out) (void)hipFree(c->v_out);
    c->v_actions = nullptr; c->v_out = nullptr;
    const size_t na = (n + 15) & ~(size_t)15, no = (n * 9 + 15) & ~(size_t)15;
    if (n > c->v_count || !c->p_out) {
        if (c->p_out) (void)hipHostFree(c->p_out);
        c->p_out = nullptr;
        c->v_count = 0;
        HIP_TRY(c, hipHostMalloc((void **)&c->p_out, no, hipHostMallocDefault));
    }
    if (want_dev) {
        HIP_TRY(c, hipMalloc((void **)&c->v_actions, na));
        HIP_TRY(c, hipMalloc((void **)&c->v_out, no));
    }
    c->v_count = n;
    return WF_OK;
}

// WF_GRAPH experiment: captured launch sequences hold the launch arguments of the moment they were captured
void drop_graphs(wf_ctx *c)
{
    for (wf_ctx::GraphEntry &g : c->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
}

// How a single step of a split context goes out.  Default: n_parts part launches of the kernel wf_pick_group_for sizes.
// One generation of waves on the reference's 14x14 grid (round 4, profiles/r04_group_shared_sweep.txt and r04_experiments.txt: 48 combinations
// of group size x shared agent phases x parts, twice): 4 environments per wave, phases A and C shared by the 4 waves of a workgroup,
// THREE part launches - 19.4-19.7 against 19.9-20.2 us per 65 536-env step, 13.9 against 15.7 at 40 960, 22.6 against 23.6 at 81 920;
// no gain on 10x10 / 12x12, slower from ~100 000 environments up.  WF_NO_ONEGEN=1 switches it off; any explicit WF_GROUP / WF_SHARED / WF_SPLIT wins.
struct StepShape { int parts, group, shared; };
StepShape single_step_shape(const wf_ctx *c)
{
    StepShape sh = { c->n_parts, c->knobs.group, c->knobs.shared };
    const bool rare = c->p.allow_dig_toggle || c->p.make_rivers;
    if (c->fam == WF_FAM_WAVE && c->W == 14 && c->H == 14 && !rare && c->p.wind_mode == WF_WIND_CALM && c->knobs.group <= 0 && !c->knobs.shared &&
        !c->knobs.split_explicit && !c->knobs.no_rows && !c->knobs.no_onegen && c->n_side >= 2 && c->batch >= 40960 && c->batch <= 90112) {
        sh.parts = 3;
        sh.group = 4;
        sh.shared = 4;
    }
    return sh;
}

// One wf_step*_device call of the wave family: the launches, on the context's stream(s).
int wave_step_issue(wf_ctx *c, const int8_t *d_actions, int n_steps, uint64_t update_mask, double *d_reward, uint8_t *d_done, int8_t *d_obs)
{
    // Large batches: one launch per stream whose waves step their environments n_steps times (shapes with a
    // rows-layout kernel), else n_steps single-step launches; both beat the fused one-environment-per-wave
    // kernel there (4.2e11 cell-updates/s at 65 536 x 14x14), which serves small, launch-bound batches.
    // The verdict is reached ONCE, from the shape of the whole call, and handed to every part launch
    // (WfKnobs::multi_rows): a ragged last part must not fall back to another kernel on its own.
    const bool sp = splits(c, 1);
    const bool calm = c->p.wind_mode == WF_WIND_CALM;
    const bool rows_loop = n_steps > 1 && sp && multi_rows_call(c, n_steps);
    const bool ticks = n_steps > 1 && sp && calm && ticks_serves_split(c, d_actions != nullptr, rows_loop);   // one launch, not split
    const bool in_kernel = n_steps > 1 && !ticks && (sp ? rows_loop
                                              : ((c->knobs.multi_small || !c->knobs.no_resident) && multi_rows_whole(c, n_steps)));
    WfKnobs kn = c->knobs;
    kn.multi_rows = in_kernel ? 1 : 0;
    // (wind with a direction has a single-step kernel only: n single-step launches whatever the batch)
    const bool each = n_steps > 1 && ((sp && !in_kernel && !ticks) || !calm);
    const int n_launch = each ? n_steps : 1;
    const size_t cells = (size_t)c->H * c->W;
    for (int st = 0; st < n_launch; ++st) {
        WfArgs a = base_args(c);
        const size_t soff = (size_t)st * (size_t)c->batch;
        a.obs = st == n_launch - 1 ? d_obs : nullptr;      // observation of the final state, written by the step kernel
        a.actions = d_actions ? d_actions + soff : nullptr;
        a.reward = d_reward ? d_reward + soff : nullptr;
        a.done = d_done ? d_done + soff : nullptr;
        a.n_steps = each ? 1 : n_steps;
        a.update_mask = each ? (update_mask >> st) & 1ull : update_mask;
        // A state past the 256 MB Infinity Cache: single steps walk the batch forwards and backwards in turn, so that a step
        // starts on the bytes the step before touched last (the rows kernel only; 1 M x 14x14: 335 -> 320 us per step in
        // round 3's experiment, which paid for it at every size; now a launch argument set for these sizes only)
        if (c->knobs.sweep_alt && a.n_steps == 1 && cells * (size_t)c->batch * 4 > ((size_t)224 << 20)) a.sweep_rev = (int32_t)(c->sweeps++ & 1u);
        // An n-step call served by wf_step_ticks goes out as several launches on the context's streams (round 4): one part's
        // launch ramp, loads and tail run under the other parts' ticks, and the parts' waves are out of step with each other.
        // 4 096 x 10x10 x 16 ticks: 13.0-13.7 against 17.3 us per call (ticks_parts above).
        const bool ticks_call = n_steps > 1 && calm && !in_kernel && !each;
        const int tparts = ticks_call && !c->sync_call ? ticks_parts(c) : 0;
        const bool split = tparts > 1 || (!ticks && splits(c, in_kernel ? 1 : a.n_steps));
        if (!split) {
            JOIN(c);
            WfKnobs k1 = kn;
            k1.whole_batch = 1;
            HIP_TRY(c, wf_launch_step(c->fam, a, k1, c->stream, c->num_cus));
            continue;
        }
        // independent parts of the batch on separate streams (inputs must be complete when this is
        // called: the context's own streams are not ordered with the caller's streams)
        WfKnobs kp = kn;
        int parts = tparts > 1 ? tparts : c->n_parts;
        if (tparts <= 1 && a.n_steps == 1) {               // a single step (also each step of an n-step call issued as single steps)
            const StepShape sh = single_step_shape(c);
            parts = sh.parts;
            a.group = sh.group;
            kp.shared = sh.shared;
        }
        {
            const int rc = begin_split(c, parts);
            if (rc != WF_OK) return rc;
        }
        const int per = split_per(c, parts);
        if (a.group <= 0) a.group = wf_pick_group_for(c->W, c->H, c->num_cus, c->batch, c->knobs);   // as for the whole batch
        for (int i = 0; i < parts; ++i) {
            const int lo = i * per, hi = lo + per < c->batch ? lo + per : c->batch;
            if (lo >= hi) break;
            WfArgs ai = a;
            ai.batch = hi - lo;
            ai.env_base = (uint32_t)lo;
            ai.planes = a.planes + (size_t)lo * cells;
            ai.planes_out = ai.planes;
            ai.envs = a.envs + lo;
            ai.reach = a.reach + (size_t)lo * 4;
            if (a.wind_cnt) ai.wind_cnt = a.wind_cnt + (size_t)lo * cells;
            if (a.actions) ai.actions = a.actions + lo;
            if (a.reward) ai.reward = a.reward + lo;
            if (a.done) ai.done = a.done + lo;
            if (a.obs) ai.obs = a.obs + (size_t)lo * cells * 3;
            HIP_TRY(c, wf_launch_step(c->fam, ai, kp, i == 0 ? c->stream : c->side[i - 1], c->num_cus));
        }
        c->side_pending = true;
    }
    return WF_OK;
}

// WF_GRAPH=1 (experiment): the launches of a call are captured ONCE per distinct argument set into a HIP graph - fork,
// the part launches on their streams (chained per stream over the steps of an n-step call), join - and replayed with
// one hipGraphLaunch.  What it measures: whether replaying pre-built dispatch packets beats issuing them (it does not
// shorten what the GPU does per launch); a one-step graph also joins the halves after EVERY step, which the plain path
// does not (its two streams run on across steps).
int wave_step_graph(wf_ctx *c, const int8_t *d_actions, int n_steps, uint64_t update_mask, double *d_reward, uint8_t *d_done, int8_t *d_obs)
{
    for (const wf_ctx::GraphEntry &g : c->graphs)
        if (g.actions == d_actions && g.reward == d_reward && g.done == d_done && g.obs == d_obs && g.n_steps == n_steps && g.mask == update_mask) {
            JOIN(c);
            HIP_TRY(c, hipGraphLaunch(g.exec, c->stream));
            return WF_OK;
        }
    if (c->graphs.size() >= 64) {                       // bounded cache: beyond it, plain launches
        return wave_step_issue(c, d_actions, n_steps, update_mask, d_reward, d_done, d_obs);
    }
    JOIN(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    wf_ctx::GraphEntry g = { d_actions, d_reward, d_done, d_obs, n_steps, update_mask, nullptr, nullptr };
    HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->main_touched = true;                             // the fork event belongs inside the capture
    int rc = wave_step_issue(c, d_actions, n_steps, update_mask, d_reward, d_done, d_obs);
    if (rc == WF_OK) rc = join_side(c);                 // the side streams rejoin the capturing stream
    hipError_t e = hipStreamEndCapture(c->stream, &g.graph);
    if (rc != WF_OK) return rc;
    HIP_TRY(c, e);
    HIP_TRY(c, hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    c->graphs.push_back(g);
    HIP_TRY(c, hipGraphLaunch(g.exec, c->stream));
    c->main_touched = true;
    return WF_OK;
}

// wf_download in two halves, so that a host-buffer step can bring the state back under ITS synchronisation
// (wf_step_download): enqueue the device -> host movement on the context's stream; after the stream has been
// synchronised, hand the bytes to the caller.
struct Download { int8_t *planes; wf_env *envs; size_t pb, eb; bool small; };

int download_enqueue(wf_ctx *c, int8_t *planes, wf_env *envs, Download *d)
{
    const size_t cells = (size_t)c->batch * c->H * c->W;
    d->planes = planes;
    d->envs = envs;
    d->pb = planes ? cells * 4 : 0;
    d->eb = envs ? (size_t)c->batch * sizeof(wf_env) : 0;
    d->small = d->pb + d->eb <= WF_ZC_OBS_MAX;
    if (d->small) {
        // small contexts (the single-environment adapter refreshes its view after every step): copy kernels into
        // pinned host memory instead of a DMA operation per array
        if (c->h_dl_bytes < d->pb + d->eb) {
            if (c->h_dl) (void)hipHostFree(c->h_dl);
            c->h_dl = nullptr;
            c->h_dl_bytes = 0;
            HIP_TRY(c, hipHostMalloc((void **)&c->h_dl, d->pb + d->eb, hipHostMallocDefault));
            c->h_dl_bytes = d->pb + d->eb;
        }
        if (d->pb && d->eb) HIP_TRY(c, wf_launch_copy_words2(c->planes, c->h_dl, d->pb, c->envs, c->h_dl + d->pb, d->eb, c->stream));
        else if (d->pb) HIP_TRY(c, wf_launch_copy_words(c->planes, c->h_dl, d->pb, c->stream));
        else if (d->eb) HIP_TRY(c, wf_launch_copy_words(c->envs, c->h_dl + d->pb, d->eb, c->stream));
        return WF_OK;
    }
    if (planes) HIP_TRY(c, hipMemcpyAsync(planes, c->planes, d->pb, hipMemcpyDeviceToHost, c->stream));
    if (envs) HIP_TRY(c, hipMemcpyAsync(envs, c->envs, d->eb, hipMemcpyDeviceToHost, c->stream));
    return WF_OK;
}

void download_finish(wf_ctx *c, const Download &d)
{
    if (!d.small) return;
    if (d.pb) memcpy(d.planes, c->h_dl, d.pb);
    if (d.eb) memcpy(d.envs, c->h_dl + d.pb, d.eb);
}

}  // namespace

extern "C" {

int wf_abi_version(void) { return WF_ABI_VERSION; }

void wf_default_params(wf_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->ignite_count = 19;          // 18 x 0.162 = 2.916 <= 3 < 19 x 0.162 (constants.py:39, utility.py:98-100)
    p->allow_dig_toggle = 0;       // constants.py:46
    p->containment = 1;
    p->auto_reset = 0;
    p->death_penalty = -1000.0;    // constants.py:32
    p->contained_bonus = 1000.0;   // constants.py:33
    p->default_reward = -1.0;      // constants.py:34
    p->fuel_lo = 20;               // utility.py:99
    p->fuel_hi = 20;
    p->ignite_ppm = 0;
    p->seed = 0;
    p->env_id0 = 0;
    p->wind_mode = WF_WIND_CALM;   // constants.py:39: [0.54, (0, 0)]
    p->wind_code = WF_WIND_CODE(0, 0, 0);
    p->wind_speed[0] = 0.0;        // environment.py:189
    p->wind_speed[1] = 0.7;
    p->wind_speed[2] = 0.85;
    p->wind_speed[3] = 0.0;
    p->heat = 0.3;                 // utility.py:98
    p->threshold = 3.0;            // utility.py:100
}

const char *wf_strerror(int status)
{
    switch (status) {
    case WF_OK: return "ok";
    case WF_ERR_ARG: return "invalid argument";
    case WF_ERR_UNSUPPORTED: return "grid shape or option not supported by this build";
    case WF_ERR_NO_DEVICE: return "no usable HIP device (the wildfire step has no CPU fallback)";
    case WF_ERR_HIP: return "HIP runtime error (see wf_last_hip_error)";
    case WF_ERR_NOMEM: return "out of device memory";
    default: return "unknown status";
    }
}

int wf_last_hip_error(const wf_ctx *ctx) { return ctx ? ctx->last_hip : 0; }

int wf_create(wf_ctx **out, int device, int batch, int H, int W, const wf_params *p)
{
    if (!out) return WF_ERR_ARG;
    *out = nullptr;
    if (batch < 1 || H < 1 || W < 1 || H > 65535 || W > 65535) return WF_ERR_ARG;
    wf_params pp;
    if (p) pp = *p; else wf_default_params(&pp);
    if (!params_ok(&pp)) return WF_ERR_ARG;
    if (wind_params_status(&pp) != WF_OK) return WF_ERR_UNSUPPORTED;
    // wind with a direction: the wave family's group-layout kernel for the shapes it serves (round 3), the generic family
    // for every other shape
    // (WF_FAMILY=cells / tiled, experiments and tests: the generic family for a shape another family would serve, the
    // tiled family for a small grid)
    WfKnobs kn0;
    wf_read_knobs(&kn0);
    WfFamily fam = kn0.family == 1 ? WF_FAM_CELLS : wf_pick_family(H, W);
    if (pp.wind_mode != WF_WIND_CALM && fam != WF_FAM_WAVE) fam = WF_FAM_CELLS;
    if (kn0.family == 2 && fam == WF_FAM_CELLS && pp.wind_mode == WF_WIND_CALM && wf_tiled_serves(H, W)) fam = WF_FAM_TILED;
    if (fam == WF_FAM_NONE) return WF_ERR_UNSUPPORTED;
    // the wave kernels address the planes with 32-bit byte offsets
    if (fam == WF_FAM_WAVE && (unsigned long long)batch * H * W * 4ull > 0xFFFFFFFFull) return WF_ERR_UNSUPPORTED;
    // the containment test (get_reward's A* part) is built for the wave and tiled families
    if (fam == WF_FAM_CELLS && pp.containment && (long long)H * W > WF_CELLS_FLOOD_MAX) return WF_ERR_UNSUPPORTED;   // byte board in LDS
    if (fam == WF_FAM_TILED && pp.containment && H > 1024) return WF_ERR_UNSUPPORTED;   // boards live in LDS (64 KiB)
    if (fam != WF_FAM_WAVE && (unsigned long long)H * W > 0xFFFFull * 0xFFFFull) return WF_ERR_UNSUPPORTED;
    if (pp.make_rivers && !rivers_ok(fam, H, W)) return WF_ERR_UNSUPPORTED;
    if (pp.auto_reset && !reset_ok(H, W)) return WF_ERR_UNSUPPORTED;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return WF_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return WF_ERR_ARG;

    wf_ctx *c = new (std::nothrow) wf_ctx();
    if (!c) return WF_ERR_NOMEM;
    c->device = device;
    c->batch = batch;
    c->H = H;
    c->W = W;
    c->p = pp;
    c->fam = fam;
    for (int r = 0; r < 3; ++r) c->spawn.n[r] = circle_points(r + 1, c->spawn.dx[r], c->spawn.dy[r]);

    int rc = WF_OK;
    hipError_t e;
    do {
        if ((e = hipSetDevice(device)) != hipSuccess) break;
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) break;
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) break;
        c->own_stream = true;
        if ((e = hipEventCreate(&c->ev0)) != hipSuccess) break;
        if ((e = hipEventCreate(&c->ev1)) != hipSuccess) break;
        c->knobs = kn0;
        // shares of the batch a large step is issued as, on as many streams.  Wave family: 2 (3 and 4 measured the same
        // 20.1-20.6 us at 65 536 x 14x14).  Big-grid families: 3 - a share's prologue / agent kernel then hides behind
        // two other shares' ticks (1 024 x 256x256: 88.5 against 91.8 us per step with 2, 89.5 with 4)
        const int want = c->knobs.n_parts > 0 ? c->knobs.n_parts : fam == WF_FAM_WAVE ? 2 : 3;
        c->n_parts = want > WF_MAX_SIDE + 1 ? WF_MAX_SIDE + 1 : want;
        c->knobs.n_parts = c->n_parts;
        // (all WF_MAX_SIDE side streams exist whatever n_parts says: n-step calls of wf_step_ticks use up to four parts)
        c->n_side = c->n_parts > 1 ? WF_MAX_SIDE : 0;
        for (int i = 0; i < c->n_side && e == hipSuccess; ++i) {
            if ((e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking)) != hipSuccess) break;
            e = hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess) break;
        const size_t cells = (size_t)batch * H * W;
        if ((e = hipMalloc((void **)&c->planes, cells * 4)) != hipSuccess) break;
        if (fam != WF_FAM_WAVE) {
            if (fam == WF_FAM_TILED) {   // in place + 64 B of halo masks per workgroup
                if ((e = hipMalloc((void **)&c->d_halo, (size_t)batch * wf_tiled_groups(H, W) * 8 * sizeof(u64))) != hipSuccess) break;
            } else {
                if ((e = hipMalloc((void **)&c->planes_alt, cells * 4)) != hipSuccess) break;
            }
            if ((e = hipMalloc((void **)&c->d_mode, (size_t)batch)) != hipSuccess) break;
            if ((e = hipMalloc((void **)&c->d_acc, (size_t)batch * 16)) != hipSuccess) break;   // 4 words per environment
            if ((e = hipMemsetAsync(c->d_mode, 0, (size_t)batch, c->stream)) != hipSuccess) break;
            if ((e = hipMemsetAsync(c->d_acc, 0, (size_t)batch * 16, c->stream)) != hipSuccess) break;
        }
        if (pp.wind_mode != WF_WIND_CALM) {
            if ((e = hipMalloc((void **)&c->wind_cnt, cells * 4)) != hipSuccess) break;
            if ((e = hipMemsetAsync(c->wind_cnt, 0, cells * 4, c->stream)) != hipSuccess) break;
        }
        if ((e = hipMalloc((void **)&c->envs, (size_t)batch * sizeof(wf_env))) != hipSuccess) break;
        // wave family: (batch,4) boards of the 4-cells-per-lane kernels + as much again for the rows layout
        // tiled family: u64[batch][H][4] lane masks; generic family: one byte per cell
        const size_t reach_words = fam == WF_FAM_WAVE ? (size_t)batch * 8 + 64
                                 : fam == WF_FAM_TILED ? (size_t)batch * H * 4 : ((size_t)batch * H * W + 7) / 8;
        if ((e = hipMalloc((void **)&c->reach, reach_words * sizeof(u64))) != hipSuccess) break;
        if ((e = hipMalloc((void **)&c->d_mask, (size_t)batch)) != hipSuccess) break;
        if ((e = hipMalloc((void **)&c->d_cold, sizeof(WfCold))) != hipSuccess) break;
        if ((e = hipMemsetAsync(c->planes, 0, cells * 4, c->stream)) != hipSuccess) break;
        if ((e = hipMemsetAsync(c->envs, 0, (size_t)batch * sizeof(wf_env), c->stream)) != hipSuccess) break;
        if ((e = hipMemsetAsync(c->reach, 0, reach_words * sizeof(u64), c->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess) {
        rc = fail(c, e);
        wf_destroy(c);
        return rc;
    }
    if ((rc = push_cold(c)) != WF_OK) {
        wf_destroy(c);
        return rc;
    }
    *out = c;
    return WF_OK;
}

int wf_destroy(wf_ctx *c)
{
    if (!c) return WF_ERR_ARG;
    (void)hipSetDevice(c->device);
    for (hipStream_t sd : c->side)
        if (sd) (void)hipStreamSynchronize(sd);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graphs(c);
    void *pinned[] = { c->d_actions, c->h_obs, c->h_dl, c->p_out, c->p_flag };
    for (void *b : pinned)
        if (b) (void)hipHostFree(b);
    void *bufs[] = { c->planes, c->planes_alt, c->envs, c->reach, c->d_mask, c->d_obs, c->d_cold, c->d_mode, c->d_acc, c->d_halo, c->wind_cnt,
                     c->v_actions, c->v_out };
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t ev : c->ev_side)
        if (ev) (void)hipEventDestroy(ev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipStream_t sd : c->side)
        if (sd) (void)hipStreamDestroy(sd);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return WF_OK;
}

int wf_set_params(wf_ctx *c, const wf_params *p)
{
    if (!c || !p || !params_ok(p)) return WF_ERR_ARG;
    if (wind_params_status(p) != WF_OK) return WF_ERR_UNSUPPORTED;
    if (c->fam == WF_FAM_CELLS && p->containment && (long long)c->H * c->W > WF_CELLS_FLOOD_MAX) return WF_ERR_UNSUPPORTED;
    if (c->fam == WF_FAM_TILED && p->containment && c->H > 1024) return WF_ERR_UNSUPPORTED;
    if (p->make_rivers && !rivers_ok(c->fam, c->H, c->W)) return WF_ERR_UNSUPPORTED;
    if (p->auto_reset && !reset_ok(c->H, c->W)) return WF_ERR_UNSUPPORTED;
    // the kernel family and the count plane were chosen in wf_create
    if ((p->wind_mode != WF_WIND_CALM) != (c->p.wind_mode != WF_WIND_CALM)) return WF_ERR_UNSUPPORTED;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graphs(c);
    c->p = *p;
    // a changed containment switch makes cached reachability boards untrustworthy
    HIP_TRY(c, wf_launch_invalidate_reach(c->envs, c->batch, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return push_cold(c);
}

int wf_set_stream(wf_ctx *c, void *hip_stream)
{
    if (!c) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    drop_graphs(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return WF_OK;
}

int wf_upload(wf_ctx *c, const int8_t *planes, const wf_env *envs)
{
    if (!c) return WF_ERR_ARG;
    // the kernels address cells through the agent's position: a record that points outside the grid is refused here
    // (like the reference's binding, pyastar.py:27-37, which validates before it calls into native code)
    if (envs)
        for (int i = 0; i < c->batch; ++i)
            if (envs[i].x >= c->W || envs[i].y >= c->H) return WF_ERR_ARG;
    // ... and so is an agent_pos byte other than 0 / 1 or a cell type the context's kernels do not know (codes 5 / 6 exist only with allow_dig_toggle, 13 / 6
    // only with make_rivers, see wildfire.h).  Large uploads are checked by several host threads (the scalar loop cost
    // 0.1 s per upload of the 1 M-environment configuration).
    if (planes && !types_ok(c, planes)) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    const size_t cells = (size_t)c->batch * c->H * c->W;
    if (planes) HIP_TRY(c, hipMemcpyAsync(c->planes, planes, cells * 4, hipMemcpyHostToDevice, c->stream));
    // planes and heat counts form ONE state: a new grid has received no heat until wf_wind_counts_upload says otherwise
    if (planes && c->wind_cnt) HIP_TRY(c, hipMemsetAsync(c->wind_cnt, 0, cells * 4, c->stream));
    if (envs) HIP_TRY(c, hipMemcpyAsync(c->envs, envs, (size_t)c->batch * sizeof(wf_env), hipMemcpyHostToDevice, c->stream));
    if (planes || envs) HIP_TRY(c, wf_launch_invalidate_reach(c->envs, c->batch, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WF_OK;
}

int wf_download(wf_ctx *c, int8_t *planes, wf_env *envs)
{
    if (!c) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    Download d;
    int rc = download_enqueue(c, planes, envs, &d);
    if (rc != WF_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    download_finish(c, d);
    return WF_OK;
}

int wf_wind_counts_upload(wf_ctx *c, const uint8_t *counts)
{
    if (!c || !counts) return WF_ERR_ARG;
    if (!c->wind_cnt) return WF_ERR_UNSUPPORTED;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    HIP_TRY(c, hipMemcpyAsync(c->wind_cnt, counts, (size_t)c->batch * c->H * c->W * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WF_OK;
}

int wf_wind_counts_download(wf_ctx *c, uint8_t *counts)
{
    if (!c || !counts) return WF_ERR_ARG;
    if (!c->wind_cnt) return WF_ERR_UNSUPPORTED;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    const size_t cells = (size_t)c->batch * c->H * c->W;
    uint32_t *types = static_cast<uint32_t *>(malloc(cells * 4));
    if (!types) return WF_ERR_NOMEM;
    hipError_t e = hipMemcpyAsync(counts, c->wind_cnt, cells * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(types, c->planes, cells * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)       // only a grass cell's counts are state (a burning or burnt cell never takes heat again)
        for (size_t i = 0; i < cells; ++i)
            if ((types[i] & 0xffu) != (uint32_t)WF_GRASS) memset(counts + 4 * i, 0, 4);
    free(types);
    HIP_TRY(c, e);
    return WF_OK;
}

int wf_reset_device(wf_ctx *c, const uint8_t *d_mask, uint64_t seed)
{
    if (!c) return WF_ERR_ARG;
    if (!reset_ok(c->H, c->W)) return WF_ERR_UNSUPPORTED;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    WfArgs a = base_args(c);
    a.mask = d_mask;
    a.reset_seed = seed;
    if (c->fam == WF_FAM_WAVE) HIP_TRY(c, wf_launch_reset(c->fam, a, c->stream, c->num_cus));
    else HIP_TRY(c, wf_launch_reset_big(a, d_mask, -1, seed, c->stream));
    return WF_OK;
}

int wf_reset(wf_ctx *c, const uint8_t *mask, uint64_t seed)
{
    if (!c) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (mask) HIP_TRY(c, hipMemcpyAsync(c->d_mask, mask, (size_t)c->batch, hipMemcpyHostToDevice, c->stream));
    int rc = wf_reset_device(c, mask ? c->d_mask : nullptr, seed);
    if (rc != WF_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WF_OK;
}

int wf_step_device(wf_ctx *c, const int8_t *d_actions, int n_steps, uint64_t update_mask,
                   double *d_reward, uint8_t *d_done)
{
    return wf_step_obs_device(c, d_actions, n_steps, update_mask, d_reward, d_done, nullptr);
}

int wf_step_obs_device(wf_ctx *c, const int8_t *d_actions, int n_steps, uint64_t update_mask,
                       double *d_reward, uint8_t *d_done, int8_t *d_obs)
{
    if (!c || n_steps < 1 || n_steps > 64) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->fam == WF_FAM_WAVE)
        return (c->knobs.graph && c->own_stream) ? wave_step_graph(c, d_actions, n_steps, update_mask, d_reward, d_done, d_obs)
                                                 : wave_step_issue(c, d_actions, n_steps, update_mask, d_reward, d_done, d_obs);
    // big-grid families: one launch sequence per step, source planes -> destination planes.  Large jobs on
    // the context's own streams run as n_parts independent shares of the batch on separate streams: the
    // small lanes = environments kernels of one share overlap the streaming kernel of the other.
    const size_t cells = (size_t)c->H * c->W;
    const bool split = splits_big(c);
    const int parts = split ? c->n_parts : 1;
    const int per = (c->batch + parts - 1) / parts;
    if (split) {
        const int rc = begin_split(c, parts);
        if (rc != WF_OK) return rc;
    } else {
        JOIN(c);
    }
    // n-step calls of the tiled family's one-launch step: the steps before the one that writes the observation (all of
    // them without one) are ONE launch per share, the step loop inside the kernel (wf_step_tiled<.., SOLO, MULTI>)
    const int solo_rows = tiled_solo(c);
    int looped = 0;
    if (solo_loop_serves(c, n_steps)) {
        const bool obs_fused = d_obs && wf_tiled_fused_obs(c->H, c->W) &&
                               wf_tiled_solo_fits(c->H, c->W, solo_rows, c->p.containment != 0, true);
        looped = obs_fused ? n_steps - 1 : n_steps;
    }
    if (looped > 1) {
        for (int i = 0; i < parts; ++i) {
            const int lo = i * per, hi = lo + per < c->batch ? lo + per : c->batch;
            if (lo >= hi) break;
            WfArgs a = base_args(c);
            a.batch = hi - lo;
            a.env_base = (uint32_t)lo;
            a.planes += (size_t)lo * cells;
            a.planes_out += (size_t)lo * cells;
            a.envs += lo;
            a.reach += (size_t)lo * c->H * 4;
            a.actions = d_actions ? d_actions + lo : nullptr;
            a.reward = d_reward ? d_reward + lo : nullptr;
            a.done = d_done ? d_done + lo : nullptr;
            a.n_steps = looped;
            a.step_stride = c->batch;
            a.update_mask = update_mask;
            a.sweep_rev = c->knobs.sweep_alt ? (int32_t)(c->sweeps & 1u) : 0;
            a.obs = nullptr;
            hipStream_t si = i == 0 ? c->stream : c->side[i - 1];
            HIP_TRY(c, wf_launch_step_big(c->fam, a, c->d_mode + lo, c->d_acc + (size_t)lo * 4, nullptr, c->p.seed, si, false, solo_rows));
        }
        c->sweeps += (unsigned)looped;
    } else {
        looped = 0;
    }
    for (int st = looped; st < n_steps; ++st) {
        const size_t off = (size_t)st * (size_t)c->batch;
        for (int i = 0; i < parts; ++i) {
            const int lo = i * per, hi = lo + per < c->batch ? lo + per : c->batch;
            if (lo >= hi) break;
            WfArgs a = base_args(c);
            a.batch = hi - lo;
            a.env_base = (uint32_t)lo;
            a.planes += (size_t)lo * cells;
            a.planes_out += (size_t)lo * cells;
            a.envs += lo;
            if (a.wind_cnt) a.wind_cnt += (size_t)lo * cells;
            a.reach = c->fam == WF_FAM_TILED ? a.reach + (size_t)lo * c->H * 4
                                             : reinterpret_cast<u64 *>(reinterpret_cast<uint8_t *>(a.reach) + (size_t)lo * cells);
            a.actions = d_actions ? d_actions + off + lo : nullptr;
            a.reward = d_reward ? d_reward + off + lo : nullptr;
            a.done = d_done ? d_done + off + lo : nullptr;
            a.n_steps = 1;
            a.update_mask = (update_mask >> st) & 1ull;
            a.sweep_rev = c->knobs.sweep_alt ? (int32_t)(c->sweeps & 1u) : 0;
            // World.get_state() of the final state: written by the tiled step kernel itself where the shape allows,
            // else by the observation kernel behind it - on the share's own stream either way (no join for it)
            const bool want_obs = d_obs && st == n_steps - 1;
            const bool fused_obs = want_obs && c->fam == WF_FAM_TILED && wf_tiled_fused_obs(c->H, c->W);
            a.obs = fused_obs ? d_obs + (size_t)lo * cells * 3 : nullptr;
            u64 *halo = c->d_halo ? c->d_halo + (size_t)lo * wf_tiled_groups(c->H, c->W) * 8 : nullptr;
            hipStream_t si = i == 0 ? c->stream : c->side[i - 1];
            HIP_TRY(c, wf_launch_step_big(c->fam, a, c->d_mode + lo, c->d_acc + (size_t)lo * 4, halo, c->p.seed, si,
                                          c->fam == WF_FAM_CELLS && wf_cells_per_env_kernel(c->batch, c->H, c->W, c->knobs), tiled_solo(c)));
            if (want_obs && !fused_obs)
                HIP_TRY(c, wf_launch_get_state(a.planes_out, d_obs + (size_t)lo * cells * 3, hi - lo, c->H, c->W, si));
        }
        if (c->planes_alt) { uint32_t *t = c->planes; c->planes = c->planes_alt; c->planes_alt = t; }   // ping-pong
        ++c->sweeps;
    }
    if (split) c->side_pending = true;
    return WF_OK;
}

int wf_step(wf_ctx *c, const int8_t *actions, int n_steps, uint64_t update_mask,
            double *reward, uint8_t *done)
{
    return wf_step_obs(c, actions, n_steps, update_mask, reward, done, nullptr);
}

// Pinned staging of the host-buffer entry points for calls of up to n_steps steps: ONE pinned block laid out
// [n doubles of reward][n bytes of done] (p_out) and the pinned action block (d_actions), n = n_steps x batch.
int ensure_host_staging(wf_ctx *c, int n_steps)
{
    int rc = ensure_staging(c, n_steps);
    if (rc != WF_OK) return rc;
    return ensure_mirrors(c, (size_t)c->staged_steps * (size_t)c->batch);
}

int wf_host_staging(wf_ctx *c, int n_steps, int8_t **actions, double **reward, uint8_t **done)
{
    if (!c || n_steps < 1 || n_steps > 64) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = ensure_host_staging(c, n_steps);
    if (rc != WF_OK) return rc;
    const size_t n = (size_t)n_steps * (size_t)c->batch;
    if (actions) *actions = c->d_actions;
    if (reward) *reward = reinterpret_cast<double *>(c->p_out);
    if (done) *done = c->p_out + n * sizeof(double);
    return WF_OK;
}

static int host_step(wf_ctx *c, const int8_t *actions, int n_steps, uint64_t update_mask, double *reward, uint8_t *done, int8_t *obs,
                     int8_t *planes, wf_env *envs);

// End of a synchronous host-buffer call.  WF_FLAG_WAIT=us: a one-lane kernel behind everything the call enqueued writes
// the call's sequence number to a pinned word and the host spins on it for at most that long before it falls back to
// hipStreamSynchronize (which is also what reports a failed kernel).
static hipError_t wait_stream(wf_ctx *c)
{
    if (c->knobs.flag_wait > 0) {
        if (!c->p_flag) {
            hipError_t e = hipHostMalloc((void **)&c->p_flag, 64, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            *c->p_flag = 0;
        }
        const uint32_t seq = ++c->flag_seq;
        hipError_t e = wf_launch_flag(c->p_flag, seq, c->stream);
        if (e != hipSuccess) return e;
        const auto t0 = std::chrono::steady_clock::now();
        const auto budget = std::chrono::microseconds(c->knobs.flag_wait);
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(c->p_flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
            __builtin_ia32_pause();
            if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > budget) break;
        }
    }
    return hipStreamSynchronize(c->stream);
}

int wf_step_obs(wf_ctx *c, const int8_t *actions, int n_steps, uint64_t update_mask,
                double *reward, uint8_t *done, int8_t *obs)
{
    return host_step(c, actions, n_steps, update_mask, reward, done, obs, nullptr, nullptr);
}

int wf_step_download(wf_ctx *c, const int8_t *actions, int n_steps, uint64_t update_mask,
                     double *reward, uint8_t *done, int8_t *planes, wf_env *envs)
{
    return host_step(c, actions, n_steps, update_mask, reward, done, nullptr, planes, envs);
}

static int host_step(wf_ctx *c, const int8_t *actions, int n_steps, uint64_t update_mask, double *reward, uint8_t *done, int8_t *obs,
                     int8_t *planes, wf_env *envs)
{
    if (!c || n_steps < 1 || n_steps > 64) return WF_ERR_ARG;
    int rc = ensure_host_staging(c, n_steps);
    if (rc != WF_OK) return rc;
    const size_t n = (size_t)n_steps * (size_t)c->batch;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    const size_t nobs = (size_t)c->batch * c->H * c->W * 3;
    const bool zc_obs = obs && nobs <= WF_ZC_OBS_MAX;
    if (zc_obs && !c->h_obs) HIP_TRY(c, hipHostMalloc((void **)&c->h_obs, nobs, hipHostMallocDefault));
    if (obs && !zc_obs && !c->d_obs) HIP_TRY(c, hipMalloc((void **)&c->d_obs, nobs));
    int8_t *obs_dst = !obs ? nullptr : zc_obs ? c->h_obs : c->d_obs;
    // pinned host views of this call: reward / done inside the one pinned block.  A caller that passes the pointers
    // wf_host_staging() lent it spares the host copies (590 KB per step at 65 536 environments: ~45 us of memcpy).
    double *p_reward = reinterpret_cast<double *>(c->p_out);
    uint8_t *p_done = c->p_out + n * sizeof(double);
    if (actions && actions != c->d_actions) memcpy(c->d_actions, actions, n);
    // A synchronous call gains nothing from the two-stream split of a large step - nothing else is in flight to overlap
    // with - and pays its fork and join (events across streams): 88 -> 77 us per wf_step at 65 536 x 14x14, 67 -> 51 on lent
    // buffers (tools/host_path_breakdown.py: a step + synchronisation with EVERY buffer in device memory is 57 us with the
    // split, of which the step is 19: the kernels' own PCIe accesses add 5).  (Polling hipStreamQuery instead of
    // hipStreamSynchronize was measured too: slower, 59 against 49 us per single-environment adapter step.)
    // (single steps only: an n-step call of a large batch keeps its split and the rows kernel's step loop)
    struct SyncCall { wf_ctx *c; SyncCall(wf_ctx *c_, bool on) : c(c_) { c->sync_call = on; } ~SyncCall() { c->sync_call = false; } } in_sync_call(c, n_steps == 1);
    // Experiment (WF_HOST_STAGE_MIN=n, off by default): calls of at least n actions go through DEVICE MIRRORS - bit 0 of
    // WF_HOST_STAGE_MODE: the actions come over with one coalesced copy kernel before the step; bit 1: reward / done
    // leave through a mirror and one copy kernel after it - instead of the step kernel reading and writing pinned host
    // memory itself.  Measured at 65 536 x 14x14 (us per wf_step, caller-owned arrays / lent buffers): no mirrors
    // 87 / 70, actions 74 lent, results 97 / 77, both 95 / 75.  The extra launches and the join cost more than the
    // kernel's own PCIe accesses; what does help is lending the pinned buffers (wf_host_staging: no host copies).
    const size_t stage_min = c->knobs.host_stage_min > 0 ? (size_t)c->knobs.host_stage_min : ~(size_t)0;
    const int stage_mode = n >= stage_min ? (c->knobs.host_stage_mode > 0 ? c->knobs.host_stage_mode : 3) : 0;
    if (stage_mode) {
        const bool in_m = (stage_mode & 1) != 0, out_m = (stage_mode & 2) != 0;
        if (actions && in_m) HIP_TRY(c, wf_launch_copy_words(c->d_actions, c->v_actions, (n + 3) & ~(size_t)3, c->stream));
        double *v_reward = reinterpret_cast<double *>(c->v_out);
        uint8_t *v_done = c->v_out + n * sizeof(double);
        rc = wf_step_obs_device(c, !actions ? nullptr : in_m ? c->v_actions : c->d_actions, n_steps, update_mask,
                                !reward ? nullptr : out_m ? v_reward : p_reward, !done ? nullptr : out_m ? v_done : p_done, obs_dst);
        if (rc != WF_OK) return rc;
        JOIN(c);
        if ((reward || done) && out_m) {
            const size_t lo = reward ? 0 : n * sizeof(double), hi = done ? n * 9 : n * sizeof(double);
            HIP_TRY(c, wf_launch_copy_words(c->v_out + lo, c->p_out + lo, (hi - lo + 3) & ~(size_t)3, c->stream));
        }
    } else {
        // the kernels read the actions from and write reward / done to pinned host memory themselves
        rc = wf_step_obs_device(c, actions ? c->d_actions : nullptr, n_steps, update_mask, reward ? p_reward : nullptr, done ? p_done : nullptr, obs_dst);
        if (rc != WF_OK) return rc;
        JOIN(c);
    }
    if (obs && !zc_obs) HIP_TRY(c, hipMemcpyAsync(obs, c->d_obs, nobs, hipMemcpyDeviceToHost, c->stream));
    Download dl = { nullptr, nullptr, 0, 0, false };
    if (planes || envs) {           // the resulting state comes back under the same synchronisation (wf_step_download)
        rc = download_enqueue(c, planes, envs, &dl);
        if (rc != WF_OK) return rc;
    }
    HIP_TRY(c, wait_stream(c));
    if (zc_obs) memcpy(obs, c->h_obs, nobs);
    if (reward && reward != p_reward) memcpy(reward, p_reward, n * sizeof(double));
    if (done && done != p_done) memcpy(done, p_done, n);
    download_finish(c, dl);
    return WF_OK;
}

int wf_get_state_device(wf_ctx *c, int8_t *d_obs)
{
    if (!c || !d_obs) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    HIP_TRY(c, wf_launch_get_state(c->planes, d_obs, c->batch, c->H, c->W, c->stream));
    return WF_OK;
}

int wf_get_state(wf_ctx *c, int8_t *obs)
{
    if (!c || !obs) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->batch * c->H * c->W * 3;
    if (!c->d_obs) HIP_TRY(c, hipMalloc((void **)&c->d_obs, n));
    int rc = wf_get_state_device(c, c->d_obs);
    if (rc != WF_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(obs, c->d_obs, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WF_OK;
}

int wf_sync(wf_ctx *c)
{
    if (!c) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WF_OK;
}

void *wf_planes_device(wf_ctx *c) { return c ? (void *)c->planes : nullptr; }
void *wf_envs_device(wf_ctx *c) { return c ? (void *)c->envs : nullptr; }

int wf_timer_start(wf_ctx *c)
{
    if (!c) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    return WF_OK;
}

int wf_timer_stop(wf_ctx *c, float *elapsed_ms)
{
    if (!c || !elapsed_ms) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);                          // the timed region ends when BOTH halves are done
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return WF_OK;
}

const char *wf_kernel_name(const wf_ctx *c)
{
    return c ? wf_family_name(c->fam, c->H, c->W) : "none";
}

const char *wf_step_kernel_name(const wf_ctx *c)
{
    if (!c) return "none";
    WfArgs a = base_args(c);
    WfKnobs kn = c->knobs;
    if (splits(c, 1)) {
        const StepShape sh = single_step_shape(c);
        a.group = sh.group;
        kn.shared = sh.shared;
        if (a.group <= 0) a.group = wf_pick_group_for(c->W, c->H, c->num_cus, c->batch, c->knobs);
        a.batch = split_per(c, sh.parts);
    }
    return wf_single_step_kernel(c->fam, a, kn, c->num_cus);
}

const char *wf_call_kernel_name(const wf_ctx *c, int n_steps)
{
    if (!c || n_steps < 1) return "none";
    if (n_steps == 1 || c->fam != WF_FAM_WAVE || c->p.wind_mode != WF_WIND_CALM) return wf_step_kernel_name(c);
    // (a call WITH actions: one without may go to wf_step_ticks where this one does not, see ticks_serves_split)
    const bool sp = splits(c, 1);
    const bool rows_loop = sp && multi_rows_call(c, n_steps);
    const bool ticks = sp && ticks_serves_split(c, true, rows_loop);
    const bool in_kernel = !ticks && (sp ? rows_loop : ((c->knobs.multi_small || !c->knobs.no_resident) && multi_rows_whole(c, n_steps)));
    if (!in_kernel) return sp && !ticks ? wf_step_kernel_name(c) : (c->knobs.fused_legacy ? "wf_step_wave (fused ticks, one environment per wave)" : "wf_step_ticks (fused ticks, one environment per wave, packed cells)");
    WfArgs a = base_args(c);
    a.n_steps = n_steps;
    if (sp) {
        a.batch = split_per(c);
        if (a.group <= 0) a.group = wf_pick_group_for(c->W, c->H, c->num_cus, c->batch, c->knobs);
    }
    return wf_multi_step_kernel(a, c->knobs, c->num_cus);
}

int wf_call_launches(const wf_ctx *c, int n_steps)
{
    if (!c || n_steps < 1) return 0;
    if (c->fam != WF_FAM_WAVE) {                                                        // launch sequences (a call without the observation)
        return (solo_loop_serves(c, n_steps) ? 1 : n_steps) * (splits_big(c) ? c->n_parts : 1);
    }
    const int per_step = splits(c, 1) ? single_step_shape(c).parts : 1;
    if (n_steps == 1) return per_step;
    if (c->p.wind_mode != WF_WIND_CALM) return per_step * n_steps;                      // wind: single-step launches only
    const int tp = ticks_parts(c) > 1 ? ticks_parts(c) : 1;                             // wf_step_ticks: one launch, or a few parts
    if (!splits(c, 1)) {
        const bool in_rows = (c->knobs.multi_small || !c->knobs.no_resident) && multi_rows_whole(c, n_steps);
        return in_rows ? 1 : tp;
    }
    const bool rows_loop = multi_rows_call(c, n_steps);
    if (ticks_serves_split(c, true, rows_loop)) return tp;                              // (a call with actions, as above)
    return rows_loop ? c->n_parts : per_step * n_steps;
}

int wf_step_plan(int height, int width, int batch, int *one_launch, int *rows, int *waves, int *step_loop)
{
    const WfFamily fam = height > 0 && width > 0 ? wf_pick_family(height, width) : WF_FAM_NONE;
    int r = 0, w = 0;
    if (fam == WF_FAM_TILED && batch > 0) {
        r = wf_tiled_solo_rows(height, width, 0, batch);
        w = wf_tiled_solo_waves(height, width, r);
    }
    if (one_launch) *one_launch = r > 0;
    if (rows) *rows = r;
    if (waves) *waves = w;
    if (step_loop) *step_loop = r > 0 && wf_tiled_solo_fits(height, width, r, true, false) && (long long)batch * w < WF_TILED_SOLO_WAVES;
    return (int)fam;
}

#ifdef WF_WAVE_TIMING
// measurement builds only: device buffer of 5 x u64 per wave of a step launch (see wf_step_rows)
int wf_debug_timing(wf_ctx *c, void *d_buf) { if (!c) return WF_ERR_ARG; c->dbg_timing = d_buf; return WF_OK; }
#endif

int wf_reload_env(wf_ctx *c)
{
    if (!c) return WF_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    JOIN(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int parts = c->n_parts;
    drop_graphs(c);
    wf_read_knobs(&c->knobs);
    c->knobs.n_parts = parts;          // the side streams exist since wf_create: WF_SPLIT is fixed for a context's life
    // boards are kept per kernel layout: a switch of kernels must not meet a board the other layout left stale
    HIP_TRY(c, wf_launch_invalidate_reach(c->envs, c->batch, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WF_OK;
}

int wf_step_launches(const wf_ctx *c)
{
    if (!c) return 0;
    if (splits(c, 1)) return single_step_shape(c).parts;
    return splits_big(c) ? c->n_parts : 1;
}

}  // extern "C"
