// wf_generic.hip - the GENERIC kernel family: any grid shape, wind with a direction, grids of fewer than 1 024 cells.
//
// Same rule as wf_kernels.hip / wf_tiled.hip.  A step is
//   wf_big_pre -> wf_step_env      lanes = environments for the agent logic, then a wave (up to 1 024 cells) or a workgroup
//                                  per environment: reset / reachability flood / tick / World.RUNNING, reward, done
//   (few huge grids, WF_CELLS_LEGACY=1: wf_big_pre -> [wf_big_reset_cells] -> [wf_cells_flood] -> wf_step_cells -> wf_big_post)
// and wf_reset of both big-grid families is wf_big_reset_rec -> wf_big_reset_cells.  Planes are ping-pong buffers here.
#include "wf_device.h"

namespace {

#include "wf_big_common.h"

// generic family: lanes = environments
__global__ __launch_bounds__(256) void wf_big_pre(const WfArgs a, uint8_t *mode)
{
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (uint32_t)a.batch) return;
    uint4 ev;
    bool have_record;
    mode[e] = (uint8_t)big_pre_env(a, e, ev, have_record);
    if (have_record) *reinterpret_cast<uint4 *>(&a.envs[e]) = ev;
}

// generic family: lanes = environments, after the step kernel.  acc: 4 words per environment (flags, healthy, arrivals, -)
__global__ __launch_bounds__(256) void wf_big_post(const WfArgs a, const uint8_t *mode, uint32_t *acc)
{
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (uint32_t)a.batch) return;
    const uint32_t A = acc[4 * e], healthy = acc[4 * e + 1];
    acc[4 * e] = 0;
    acc[4 * e + 1] = 0;
    const uint32_t m = mode[e];
    if ((m & MODE_MASK) != MODE_STEP) return;
    big_post_env(a, e, m, A, healthy);
}

// ---------------------------------------------------------------------------- flood, generic family
// Border-reachability board of one environment as one byte per cell (1 = 4-connected to a passable border cell
// through passable cells), for grid shapes that only the thread-per-cell kernel serves.  One workgroup per flagged
// environment, the byte board lives in LDS (H*W <= WF_CELLS_FLOOD_MAX), Jacobi-free in-place relaxation until a
// sweep changes nothing.
__global__ __launch_bounds__(256) void wf_cells_flood(const WfArgs a, const uint8_t *mode)
{
    extern __shared__ uint8_t board[];               // [cells] 0 impassable, 1 passable, 3 passable + reached
    __shared__ uint32_t changed;
    const uint32_t e = blockIdx.x;
    if (!(mode[e] & MODE_FLOOD)) return;
    const uint32_t W = (uint32_t)a.W, H = (uint32_t)a.H, cells = W * H;
    const uint32_t *se = a.planes + (size_t)e * cells;
    for (uint32_t q = threadIdx.x; q < cells; q += 256u) {
        const uint32_t x = q % W, y = q / W;
        const bool pass = (se[q] & 0xffu) < (uint32_t)WF_DIRT;
        const bool edge = x == 0 || y == 0 || x == W - 1 || y == H - 1;
        board[q] = pass ? (edge ? 3 : 1) : 0;
    }
    __syncthreads();
    for (uint32_t it = 0; it < cells; ++it) {
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        bool ch = false;
        for (uint32_t q = threadIdx.x; q < cells; q += 256u) {
            if (board[q] != 1) continue;
            const uint32_t x = q % W, y = q / W;
            const bool nb = (x > 0 && board[q - 1] == 3) || (x + 1 < W && board[q + 1] == 3) ||
                            (y > 0 && board[q - W] == 3) || (y + 1 < H && board[q + W] == 3);
            if (nb) { board[q] = 3; ch = true; }
        }
        if (ch) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    uint8_t *out = reinterpret_cast<uint8_t *>(a.reach) + (size_t)e * cells;
    for (uint32_t q = threadIdx.x; q < cells; q += 256u) {
        uint8_t v = board[q] >> 1;
        if ((a.hot_flags & WF_HOT_RARE) && blocked_burning(se[q])) {
            // burning dirt / blocked fire: impassable, but the reference's search starts from it -> reached iff a 4-neighbour is
            const uint32_t x = q % W, y = q / W;
            v = ((x > 0 && board[q - 1] == 3) || (x + 1 < W && board[q + 1] == 3) ||
                 (y > 0 && board[q - W] == 3) || (y + 1 < H && board[q + W] == 3)) ? 1 : 0;
        }
        out[q] = v;
    }
    if (threadIdx.x == 0) atomicOr(reinterpret_cast<uint32_t *>(&a.envs[e]) + 1, (uint32_t)WF_F_REACH_VALID);
}

// ---------------------------------------------------------------------------- generic step
// Any grid shape: one thread per cell, neighbours read from the source planes.
// WIND: wind with a direction (wf_params.wind_mode != 0).  The addend a burning cell hands to a neighbour depends on the
// direction between them (environment.py:260-287), so a grass cell keeps one count per direction (a.wind_cnt, a byte
// each) and ignites when the float64 sum of its counted addends exceeds the threshold (:293).  The reference adds in
// the order its burning set iterates and rounds differently in the last bits; no reachable count combination brings
// the sum closer than 1.9e-5 to the threshold for the winds "random" can draw (tests/test_host_cpu.py enumerates
// them), so every order decides alike.
// One cell's tick (forest_fire.py:85-106) from its own word `c` and its four neighbours' words (nS: the cell above,
// nN: the one below, nE: the left one, nW: the right one - named after the direction in which they heat this cell;
// 0 where the grid ends): returns the new word; `ig` = ignited this tick.  wc / threshold / pn: wind only (addends per
// direction, the cell's count word).
template <bool WIND>
__device__ __forceinline__ uint32_t cell_tick(uint32_t c, uint32_t nN, uint32_t nS, uint32_t nE, uint32_t nW, uint32_t ign,
                                              const double *wc, double threshold, uint32_t *pn, uint32_t &ig)
{
    auto still = [&](uint32_t v) -> uint32_t { return ((v & 3u) == 1u && (int32_t)(int8_t)(v >> 8) > 1) ? 1u : 0u; };
    const uint32_t t = c & 0xffu;
    ig = 0;
    if ((t & 3u) == 1u) {                                                  // burning: fire or burning dirt
        // reduce_fuel :297-307 (low half only): fuel - 1, and type + 1 when it burns out (blocked fire 13 -> blocked burnt 6)
        c = (c & 0xFFFF0000u) | ((c + (still(c) ? 0xFF00u : burn_out_step<true>(c))) & 0xFFFFu);
    } else if (t == WF_GRASS) {
        // get_neighbours :311-326.  Direction k of the vector burning cell -> this cell: the cell below heats
        // it northwards (k = 0), the one above southwards (1), the left one eastwards (2), the right one westwards (3)
        const uint32_t bN = still(nN), bS = still(nS), bE = still(nE), bW = still(nW);
        if (WIND) {
            if (bN | bS | bE | bW) {
                const uint32_t n = *pn;
                const uint32_t n0 = min((n & 0xffu) + bN, 255u), n1 = min(((n >> 8) & 0xffu) + bS, 255u);
                const uint32_t n2 = min(((n >> 16) & 0xffu) + bE, 255u), n3 = min((n >> 24) + bW, 255u);
                *pn = n0 | (n1 << 8) | (n2 << 16) | (n3 << 24);
                const double temp = (double)n0 * wc[0] + (double)n1 * wc[1] + (double)n2 * wc[2] + (double)n3 * wc[3];
                ig = temp > threshold ? 1u : 0u;                           // apply_heat_from_to :278-294
                if (ig) c |= (uint32_t)WF_HEAT_IGNITED << 16;
                c += ig;
            }
        } else {
            const uint32_t cnt = bN + bS + bE + bW;
            const uint32_t room = ign - ((c >> 16) & 0xffu);
            ig = cnt >= room ? 1u : 0u;                                    // apply_heat_from_to :278-294
            c += (min(cnt, room) << 16) + ig;
        }
    }
    return c;
}

template <bool WIND>
__global__ __launch_bounds__(256) void wf_step_cells(const WfArgs a, const uint8_t *mode, uint32_t *acc)
{
    const uint32_t W = (uint32_t)a.W, H = (uint32_t)a.H, cells = W * H;
    const size_t total = (size_t)a.batch * cells;
    const uint32_t ign = (uint32_t)a.ignite_count;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t e = (uint32_t)(i / cells), q = (uint32_t)(i % cells), x = q % W, y = q / W;
        const uint32_t *se = a.planes + (size_t)e * cells;
        uint32_t c = se[q];
        const uint32_t m = mode[e] & MODE_MASK;
        if (m == MODE_STEP && (a.update_mask & 1ull)) {
            const uint32_t code = WIND ? (uint32_t)a.envs[e].wind : 0u;
            uint32_t ig, nN = 0, nS = 0, nE = 0, nW = 0;
            if ((c & 0xffu) == WF_GRASS) {                                         // get_neighbours :311-326
                if (y + 1 < H) nN = se[q + W];
                if (y > 0) nS = se[q - W];
                if (x > 0) nE = se[q - 1];
                if (x + 1 < W) nW = se[q + 1];
            }
            c = cell_tick<WIND>(c, nN, nS, nE, nW, ign, WIND ? a.cold->wind_c[(code >> 4) & 3u][code & 15u] : nullptr,
                                WIND ? a.cold->p.threshold : 0.0, WIND ? a.wind_cnt + (size_t)e * cells + q : nullptr, ig);
            if (ig && (x == 0 || y == 0 || x == W - 1 || y == H - 1)) atomicOr(&acc[4 * e], 2u);
        }
        a.planes_out[(size_t)e * cells + q] = c;
        if (m == MODE_STEP) {
            const uint32_t t = c & 0xffu;
            // containment (environment.py:345-377): a burning cell on a border-connected cell
            if ((mode[e] & MODE_PRECHECK) && (t & 3u) == 1u && reinterpret_cast<const uint8_t *>(a.reach)[(size_t)e * cells + q])
                atomicOr(&acc[4 * e], 4u);
            // wave-aggregated when the whole wave sits in one env (the common case)
            const uint32_t e0 = rfl(e);
            if (__ballot(e != e0) == 0) {
                const u64 f = __ballot((t & 3u) == 1u), g = __ballot(t == WF_GRASS);
                if (__lane_id() == (uint32_t)__ffsll((long long)__ballot(true)) - 1u) {
                    if (f) atomicOr(&acc[4 * e], 1u);
                    if (g) atomicAdd(&acc[4 * e + 1], (uint32_t)__popcll(g));
                }
            } else {
                if ((t & 3u) == 1u) atomicOr(&acc[4 * e], 1u);
                if (t == WF_GRASS) atomicAdd(&acc[4 * e + 1], 1u);
            }
        }
    }
}

// ---------------------------------------------------------------------------- generic step, one wave / workgroup per environment
// The generic family's step behind wf_big_pre as ONE launch: NT threads own an environment (NT = 64: a wave, four
// environments per workgroup; NT = 256: a workgroup) and do what its mode asks - the fresh grid of an auto-reset, the
// copy of a frozen one, or [reachability flood ->] tick -> World.RUNNING / reward / done - with the reductions in
// ballots / LDS instead of global atomics.  (The five-launch sequence above costs 340 us per 65 536 x 14x14 step,
// three quarters of it in launches whose workgroups find nothing to do and in same-address atomics; it stays for a few
// huge grids, where one workgroup per environment would leave the chip empty.)
template <int NT>
__device__ __forceinline__ void env_sync()
{
    if constexpr (NT == 64) {      // one wave: LDS is in order, the compiler must not move accesses across this point
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// STAGE: the environment's grid (and its reachability bytes) fit the slot's LDS: they are staged there by batched,
// independent loads - one memory latency per environment instead of one per pass - and flood and tick read LDS.
// BM: words per row of the linear row bitmaps the flood runs on (wf_tiled_prologue's run fill): 4 for the
// workgroup-per-environment grids (W <= 256), 1 for the wave-per-environment ones (W <= 64; the 256-bit row arithmetic
// would cost that kernel half its occupancy); wider grids flood on the byte board (bm_off == 0).
#ifndef WF_ENV_WAVES
#define WF_ENV_WAVES 6          // waves per SIMD of the wave-per-environment instantiations (4 / 6 / 8: 101 / 91 / 93 us per 65 536 x 14x14 wind step)
#endif
template <int NT, bool WIND, bool STAGE, int BM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT == 64 ? WF_ENV_WAVES : 4, 8))) void wf_step_env(const WfArgs a, const uint8_t *mode, uint32_t slot_bytes, uint32_t bm_off)
{
    // per environment slot: [STAGE: the grid, one word per cell] [reachability byte board | river row table]
    // [bm_off != 0: two linear row bitmaps of BM words per row for the flood]
    extern __shared__ __attribute__((aligned(16))) uint8_t env_lds[];
    constexpr uint32_t EPB = 256 / NT;
    __shared__ uint32_t red[EPB][4];                 // flags, healthy, changed (flood), -
    const uint32_t slot = threadIdx.x / NT, t = threadIdx.x % NT;
    const uint32_t W = (uint32_t)a.W, H = (uint32_t)a.H, cells = W * H;
    const uint32_t ign = (uint32_t)a.ignite_count;
    uint32_t *ls = reinterpret_cast<uint32_t *>(env_lds + (size_t)slot * slot_bytes);
    uint8_t *board = env_lds + (size_t)slot * slot_bytes + (STAGE ? (((size_t)cells * 4 + 15) & ~(size_t)15) : 0);
    const WfCold *cold = a.cold;
    for (uint32_t e = blockIdx.x * EPB + slot; e < (uint32_t)a.batch; e += gridDim.x * EPB) {
        // everything this environment needs from memory is requested up front - mode, record, and (STAGE) the whole
        // grid with its reachability bytes: one memory latency per environment
        const uint32_t m = mode[e], mm = m & MODE_MASK;
        const uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);
        const uint32_t *__restrict__ se = a.planes + (size_t)e * cells;
        uint32_t *__restrict__ de = a.planes_out + (size_t)e * cells;
        uint8_t *reach_e = reinterpret_cast<uint8_t *>(a.reach) + (size_t)e * cells;
        if constexpr (STAGE) {
            const bool want_reach = (a.hot_flags & WF_HOT_CONTAINMENT) != 0;
            env_sync<NT>();                           // the slot's last environment is done with it
            constexpr int U = 4;                      // loads in flight per lane and trip (x2 with the reachability bytes)
            for (uint32_t q0 = 0; q0 < cells; q0 += NT * (uint32_t)U) {
                uint32_t v[U], r[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t q = q0 + (uint32_t)k * NT + t;
                    v[k] = q < cells ? se[q] : 0u;
                    r[k] = (want_reach && q < cells) ? (uint32_t)reach_e[q] : 0u;
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t q = q0 + (uint32_t)k * NT + t;
                    if (q < cells) { ls[q] = v[k]; if (want_reach) board[q] = (uint8_t)(r[k] << 2); }
                }
            }
            env_sync<NT>();
        }
        if (mm == MODE_RESET) {
            // World.reset() (environment.py:186-212): the record was written by wf_big_pre (reset_record)
            const EnvKey key(cold->p.seed, cold->p.env_id0 + a.env_base + (u64)e, ev.z - 1u);
            const ResetLaw L = reset_law(a, ev);
            const bool rivers = cold->p.make_rivers != 0;
            uint16_t *river_x = reinterpret_cast<uint16_t *>(board);
            if (rivers) {
                env_sync<NT>();
                for (uint32_t r = t; r < H; r += NT) river_x[r] = 0xffffu;
                env_sync<NT>();
                if (t == 0) river_cells(key, (int)W, (int)H, [&](int x, int y) { river_x[y] = (uint16_t)x; });
                env_sync<NT>();
            }
            bool border_fire = false;
            for (uint32_t q = t; q < cells; q += NT) {
                const uint32_t x = q % W, y = q / W;
                de[q] = reset_cell(L, key, q, x, y, rivers ? (uint32_t)river_x[y] : 0xffffu, border_fire);
                if (WIND) a.wind_cnt[(size_t)e * cells + q] = 0;            // a fresh grid has received no heat
            }
            if (__ballot(border_fire) && __lane_id() == 0)
                atomicOr(reinterpret_cast<uint32_t *>(&a.envs[e]) + 1, (uint32_t)WF_F_BORDER);
            if (rivers) env_sync<NT>();
            continue;
        }
        if (mm != MODE_STEP) {                        // frozen after done: the grid moves to the other buffer unchanged
            if (de != se)
                for (uint32_t q = t; q < cells; q += NT) de[q] = se[q];
            continue;
        }
        const bool upd = (a.update_mask & 1ull) != 0, precheck = (m & MODE_PRECHECK) != 0, flood = (m & MODE_FLOOD) != 0;
        auto src = [&](uint32_t q) -> uint32_t { if constexpr (STAGE) return ls[q]; else return se[q]; };
        uint32_t valid = 0;
        if (flood && bm_off) {
            // border-reachability board through linear row bitmaps (as wf_tiled_prologue): passable cells and seeds per
            // row, then one thread per row takes what the rows above and below reached and fills along the row's runs
            u64 *pass = reinterpret_cast<u64 *>(env_lds + (size_t)slot * slot_bytes + bm_off), *reach = pass + H * (uint32_t)BM;
            for (uint32_t k = t; k < H * (uint32_t)BM; k += NT) {
                const uint32_t r = k / (uint32_t)BM, w = k % (uint32_t)BM, x0 = w * 64u;
                u64 mk = 0;
                if (x0 < W) {
                    const uint32_t n = min(64u, W - x0);
                    for (uint32_t i = 0; i < n; ++i) mk |= (u64)((src(r * W + x0 + i) & 0xffu) < (uint32_t)WF_DIRT ? 1u : 0u) << i;
                }
                pass[k] = mk;
                u64 seeds = (r == 0 || r == H - 1) ? ~0ull : 0ull;
                if (w == 0) seeds |= 1ull;
                if (w == (W - 1u) >> 6) seeds |= 1ull << ((W - 1u) & 63u);
                reach[k] = mk & seeds;
            }
            env_sync<NT>();
            relax_row_bitmaps<BM>(pass, reach, H, cells, t, (uint32_t)NT, &red[slot][2], [] { env_sync<NT>(); });
            auto reached = [&](uint32_t x, uint32_t y) -> uint32_t { return (uint32_t)(reach[y * (uint32_t)BM + (x >> 6)] >> (x & 63u)) & 1u; };
            for (uint32_t q = t; q < cells; q += NT) {
                const uint32_t y = q / W, x = q - y * W;
                uint32_t v = reached(x, y);
                if ((a.hot_flags & WF_HOT_RARE) && blocked_burning(src(q)))     // burning dirt / blocked fire: reached iff a 4-neighbour is
                    v = ((x > 0 && reached(x - 1, y)) || (x + 1 < W && reached(x + 1, y)) || (y > 0 && reached(x, y - 1)) || (y + 1 < H && reached(x, y + 1))) ? 1u : 0u;
                reach_e[q] = (uint8_t)v;
                if constexpr (STAGE) board[q] = (uint8_t)(v << 2);
            }
            env_sync<NT>();
            valid = WF_F_REACH_VALID;
        } else if (flood) {
            // (grids wider than 256 cells, or too large for the bitmaps next to the byte board: cell-at-a-time relaxation)
            // border-reachability board of this environment, bits 0-1: 0 impassable, 1 passable, 3 passable + reached;
            // bit 2: the verdict the containment test reads (burning dirt: reached iff a 4-neighbour is)
            for (uint32_t q = t; q < cells; q += NT) {
                const uint32_t x = q % W, y = q / W;
                const bool pass = (src(q) & 0xffu) < (uint32_t)WF_DIRT;
                const bool edge = x == 0 || y == 0 || x == W - 1 || y == H - 1;
                board[q] = pass ? (edge ? 3 : 1) : 0;
            }
            env_sync<NT>();
            for (uint32_t it = 0; it < cells; ++it) {
                if (t == 0) red[slot][2] = 0;
                env_sync<NT>();
                bool ch = false;
                for (uint32_t q = t; q < cells; q += NT) {
                    if (board[q] != 1) continue;
                    const uint32_t x = q % W, y = q / W;
                    const bool nb = (x > 0 && board[q - 1] == 3) || (x + 1 < W && board[q + 1] == 3) ||
                                    (y > 0 && board[q - W] == 3) || (y + 1 < H && board[q + W] == 3);
                    if (nb) { board[q] = 3; ch = true; }
                }
                if (ch) red[slot][2] = 1;
                env_sync<NT>();
                const uint32_t changed = red[slot][2];
                env_sync<NT>();
                if (!changed) break;
            }
            for (uint32_t q = t; q < cells; q += NT) {
                uint32_t v = (board[q] & 3u) >> 1;
                if ((a.hot_flags & WF_HOT_RARE) && blocked_burning(src(q))) {
                    const uint32_t x = q % W, y = q / W;
                    v = ((x > 0 && (board[q - 1] & 3u) == 3u) || (x + 1 < W && (board[q + 1] & 3u) == 3u) ||
                         (y > 0 && (board[q - W] & 3u) == 3u) || (y + 1 < H && (board[q + W] & 3u) == 3u)) ? 1u : 0u;
                }
                reach_e[q] = (uint8_t)v;
                board[q] |= (uint8_t)(v << 2);        // (own byte only; the neighbours' tests mask bits 0-1)
            }
            env_sync<NT>();
            valid = WF_F_REACH_VALID;
        }
        // ---- the tick, with the environment's reductions in wave-uniform registers
        double wc[4] = { 0.0, 0.0, 0.0, 0.0 }, threshold = 0.0;      // wind: this environment's addends per direction
        if (WIND) {
            const uint32_t code = ev.y >> 16;
            const double *tab = cold->wind_c[(code >> 4) & 3u][code & 15u];
            wc[0] = tab[0]; wc[1] = tab[1]; wc[2] = tab[2]; wc[3] = tab[3];
            threshold = cold->p.threshold;
        }
        uint32_t A = 0, healthy = 0;
        // (x, y) of this lane's cell, advanced from pass to pass without a division
        uint32_t y = t / W, x = t - y * W;
        const uint32_t step_y = (uint32_t)NT / W, step_x = (uint32_t)NT - step_y * W;
        // every lane takes its cell, the four neighbours and the reachability byte up front (clamped addresses, the
        // words masked where the grid ends): nothing in a pass depends on the last one
        for (uint32_t q0 = 0; q0 < cells; q0 += NT) {
            const uint32_t q = q0 + t;
            const bool in = q < cells;
            const uint32_t qc = in ? q : cells - 1u;
            const uint32_t yc = in ? y : H - 1u, xc = in ? x : W - 1u;
            const bool top = yc == 0, bottom = yc == H - 1, left = xc == 0, right = xc == W - 1;
            x += step_x;
            y += step_y;
            if (x >= W) { x -= W; ++y; }
            uint32_t c = src(qc);
            const uint32_t vS = src(top ? qc : qc - W), vN = src(bottom ? qc : qc + W);
            const uint32_t vE = src(left ? qc : qc - 1u), vW = src(right ? qc : qc + 1u);
            uint32_t rb = 0;
            if (precheck) {
                if constexpr (STAGE) rb = board[qc] & 4u;
                else rb = reach_e[qc];
            }
            uint32_t ig = 0;
            if (upd && in) c = cell_tick<WIND>(c, bottom ? 0u : vN, top ? 0u : vS, left ? 0u : vE, right ? 0u : vW, ign, wc, threshold,
                                               WIND ? a.wind_cnt + (size_t)e * cells + qc : nullptr, ig);
            if (in) de[q] = c;
            const uint32_t ty = c & 0xffu;
            const bool fire = in && (ty & 3u) == 1u;
            if (__ballot(fire)) A |= 1u;
            if (__ballot(in && ig && (top | bottom | left | right))) A |= 2u;               // set_fire_to :245-246
            if (__ballot(fire && rb)) A |= 4u;                                               // environment.py:345-377
            healthy += (uint32_t)__popcll(__ballot(in && ty == WF_GRASS));
        }
        if constexpr (NT != 64) {                     // four waves: combine in LDS
            if (t == 0) { red[slot][0] = 0; red[slot][1] = 0; }
            __syncthreads();
            if (__lane_id() == 0) { atomicOr(&red[slot][0], A); atomicAdd(&red[slot][1], healthy); }
            __syncthreads();
            A = red[slot][0];
            healthy = red[slot][1];
            __syncthreads();
        }
        if (t == 0) big_post_env(a, e, m, A, healthy, valid, ev);
    }
}

// ---------------------------------------------------------------------------- reset
// World.reset() (environment.py:186-212) in two kernels: records, then cells.
__global__ __launch_bounds__(256) void wf_big_reset_rec(const WfArgs a, const uint8_t *sel, int sel_value, u64 seed)
{
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (uint32_t)a.batch) return;
    if (sel && sel[e] != (uint8_t)sel_value && !(sel_value < 0 && sel[e])) return;
    reset_record(a, e, seed);
}

// Tasks = (environment, chunk of WF_RESET_CHUNK cells), spread over the workgroups of a bounded grid.  (One workgroup
// per environment made a 256x256 reset a 100 us serial job of 256 threads; with finished environments being reset
// every step of a long 1 024 x 256x256 run that was a quarter of the step time.)  Environments that are not being
// reset cost their tasks one byte load each.
constexpr uint32_t WF_RESET_CHUNK = 8192;

__global__ __launch_bounds__(256) void wf_big_reset_cells(const WfArgs a, const uint8_t *sel, int sel_value, u64 seed)
{
    const uint32_t W = (uint32_t)a.W, H = (uint32_t)a.H, cells = W * H;
    const uint32_t chunks = (cells + WF_RESET_CHUNK - 1u) / WF_RESET_CHUNK, ntasks = (uint32_t)a.batch * chunks;
    // reset_map's river (environment.py:69-95): column of the water cell of every row (0xffff: none)
    __shared__ uint16_t river_x[WF_RIVER_MAX_H];
    const WfCold *cold = a.cold;
    for (uint32_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const uint32_t e = task / chunks, q0 = (task % chunks) * WF_RESET_CHUNK, q1 = min(q0 + WF_RESET_CHUNK, cells);
        if (sel && sel[e] != (uint8_t)sel_value && !(sel_value < 0 && sel[e])) continue;
        const uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);     // written by wf_big_reset_rec / the agent logic
        const EnvKey key(seed, cold->p.env_id0 + a.env_base + (u64)e, ev.z - 1u);
        const ResetLaw L = reset_law(a, ev);
        uint32_t *pe = a.planes + (size_t)e * cells;
        bool border_fire = false;
        const bool rivers = cold->p.make_rivers != 0;
        if (rivers) {                                                      // (every chunk walks the river: it is short)
            __syncthreads();
            for (uint32_t r = threadIdx.x; r < H; r += 256u) river_x[r] = 0xffffu;
            __syncthreads();
            if (threadIdx.x == 0) river_cells(key, (int)W, (int)H, [&](int x, int y) { river_x[y] = (uint16_t)x; });
            __syncthreads();
        }
        for (uint32_t q = q0 + threadIdx.x; q < q1; q += 256u) {
            const uint32_t x = q % W, y = q / W;
            pe[q] = reset_cell(L, key, q, x, y, rivers ? (uint32_t)river_x[y] : 0xffffu, border_fire);
            if (a.wind_cnt) a.wind_cnt[(size_t)e * cells + q] = 0;        // a fresh grid has received no heat
        }
        if (__ballot(border_fire) && __lane_id() == 0)
            atomicOr(reinterpret_cast<uint32_t *>(&a.envs[e]) + 1, (uint32_t)WF_F_BORDER);
    }
}

}  // namespace

static unsigned reset_cells_grid(const WfArgs &a)
{
    const unsigned long long chunks = ((unsigned long long)a.H * a.W + WF_RESET_CHUNK - 1) / WF_RESET_CHUNK;
    const unsigned long long tasks = chunks * (unsigned long long)a.batch;
    return (unsigned)(tasks < 4096 ? tasks : 4096);
}

static hipError_t launch_reset_cells(const WfArgs &a, const uint8_t *sel, int sel_value, u64 seed, hipStream_t s)
{
    hipLaunchKernelGGL(wf_big_reset_cells, dim3(reset_cells_grid(a)), dim3(256), 0, s, a, sel, sel_value, seed);
    return hipGetLastError();
}

hipError_t wf_launch_reset_big(const WfArgs &a, const uint8_t *sel, int sel_value, u64 seed, hipStream_t s)
{
    const unsigned eb = (unsigned)((a.batch + 255) / 256);
    hipLaunchKernelGGL(wf_big_reset_rec, dim3(eb), dim3(256), 0, s, a, sel, sel_value, seed);
    return launch_reset_cells(a, sel, sel_value, seed, s);
}

// Generic family: is a step served by wf_big_pre -> wf_step_env (one wave / workgroup per environment)?  Not for a few
// huge grids, which the thread-per-cell sequence spreads over the whole chip.  WF_CELLS_LEGACY=1 forces that sequence.
bool wf_cells_per_env_kernel(int batch, int H, int W, const WfKnobs &k)
{
    const long long cells = (long long)H * W;
    return !k.cells_legacy && (cells <= 1024 || batch >= 512);
}

hipError_t wf_launch_step_generic(const WfArgs &a, uint8_t *mode, uint32_t *acc, u64 auto_seed, hipStream_t s, bool per_env)
{
    const unsigned eb = (unsigned)((a.batch + 255) / 256);
    hipLaunchKernelGGL(wf_big_pre, dim3(eb), dim3(256), 0, s, a, mode);
    if (per_env) {
        // agent logic, then everything else of the step in one launch: a wave (grids up to 1 024 cells) or a workgroup per
        // environment; grid + reachability bytes staged in LDS where that leaves eight workgroups per CU (20 KiB each:
        // a 48x48 grid per wave with 46 KiB per workgroup ran at 3 waves per SIMD, 257 us against 110)
        const size_t cells = (size_t)a.H * a.W;
        const bool cont = (a.hot_flags & WF_HOT_CONTAINMENT) != 0, wind = (a.hot_flags & WF_HOT_WIND) != 0;
        size_t aux = cont ? cells : 0;                                           // byte board
        if ((a.hot_flags & WF_HOT_RIVERS) && aux < (size_t)a.H * 2) aux = (size_t)a.H * 2;   // river row table
        aux = (aux + 15) & ~(size_t)15;
        const bool wave = cells <= 1024;
        // flood on two row bitmaps: 256 bits per row for the workgroup-per-environment grids, 64 for a wave's
        size_t bm = !cont ? 0 : (!wave && a.W <= 256) ? (size_t)a.H * 64 : (wave && a.W <= 64) ? (size_t)a.H * 16 : 0;
        const unsigned epb = wave ? 4u : 1u;
        const size_t words = (cells * 4 + 15) & ~(size_t)15;                     // the staged grid (16-byte granules, like every part)
        const bool stage = (words + aux + bm) * epb <= 20480;
        if (((stage ? words : 0) + aux + bm) * epb > 65536) bm = 0;
        const size_t slot = (stage ? words : 0) + aux + bm;
        const uint32_t bm_off = bm ? (uint32_t)(slot - bm) : 0u;
        unsigned blocks = (unsigned)((a.batch + epb - 1) / epb);
        if (blocks > 8192) blocks = 8192;
#define WF_ENV_LAUNCH(NT, WIND, STAGE) \
        hipLaunchKernelGGL((wf_step_env<NT, WIND, STAGE, (NT == 256 ? 4 : 1)>), dim3(blocks), dim3(256), slot * epb, s, a, (const uint8_t *)mode, (uint32_t)slot, bm_off)
        if (wave) {
            if (wind) { if (stage) WF_ENV_LAUNCH(64, true, true); else WF_ENV_LAUNCH(64, true, false); }
            else { if (stage) WF_ENV_LAUNCH(64, false, true); else WF_ENV_LAUNCH(64, false, false); }
        } else {
            if (wind) { if (stage) WF_ENV_LAUNCH(256, true, true); else WF_ENV_LAUNCH(256, true, false); }
            else { if (stage) WF_ENV_LAUNCH(256, false, true); else WF_ENV_LAUNCH(256, false, false); }
        }
#undef WF_ENV_LAUNCH
        return hipGetLastError();
    }
    if (a.hot_flags & WF_HOT_AUTO_RESET) {
        hipError_t e = launch_reset_cells(a, mode, MODE_RESET, auto_seed, s);
        if (e != hipSuccess) return e;
    }
    if (a.hot_flags & WF_HOT_CONTAINMENT) {
        const size_t lds_bytes = ((size_t)a.H * a.W + 15) & ~(size_t)15;
        hipLaunchKernelGGL(wf_cells_flood, dim3((unsigned)a.batch), dim3(256), lds_bytes, s, a, (const uint8_t *)mode);
    }
    const size_t total = (size_t)a.batch * a.H * a.W;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (a.hot_flags & WF_HOT_WIND) hipLaunchKernelGGL(wf_step_cells<true>, dim3((unsigned)blocks), dim3(256), 0, s, a, mode, acc);
    else hipLaunchKernelGGL(wf_step_cells<false>, dim3((unsigned)blocks), dim3(256), 0, s, a, mode, acc);
    hipLaunchKernelGGL(wf_big_post, dim3(eb), dim3(256), 0, s, a, mode, acc);
    return hipGetLastError();
}
