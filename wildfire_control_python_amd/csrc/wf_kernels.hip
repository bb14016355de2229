// wf_kernels.hip - gfx950 (MI355X / CDNA4) kernels of the batched wildfire CA step.
//
// Rule implemented (synchronous integer restatement of the reference tick,
// Simulation/forest_fire.py:85-106 + Simulation/environment.py:278-326, proven
// order-independent for wind vector (0,0): all heat addends are equal and >= 0):
//   B     = cells of type fire at tick start
//   fuel[B] -= 1; fuel <= 0 -> burnt;  still = B minus the burnt-out cells
//   grass: heat += #{4-neighbours in still}; heat >= ignite_count -> fire (heat saturates there)
//   cells ignited this tick neither lose fuel nor radiate until the next tick.
//
// WAVE family (H*W <= 256, H*W % 4 == 0): ONE ENVIRONMENT PER 64-LANE WAVEFRONT.
//   * lane l owns cells q = 4l .. 4l+3 (row-major, q = y*W + x): one coalesced
//     global_load_dwordx4 / global_store_dwordx4 per lane moves the whole grid
//     (196 cells = 49 lanes x 16 B, contiguous 784 B per env); the next env of the
//     wave is prefetched while the current one is computed.
//   * "still burning" is turned into 64-bit lane masks with a wavefront ballot
//     (v_cmp -> SGPR pair); the 4-neighbour counts come from those masks with scalar
//     shifts (neighbour q+-1 / q+-W of component j lives at a fixed (lane offset,
//     component) pair known at compile time) and are read back per lane with
//     v_cndmask on the mask.  No LDS allocation, no barrier, no per-cell index math.
//   * work split: the scalar unit (one instruction per clock for the whole CU: a SIMD
//     gets a scalar slot as often as a vector slot) does the 64-bit mask algebra and
//     the rare uniform branches; the per-cell rule runs on the vector unit.  The
//     per-environment logic (agent move/dig/death, RUNNING, reward) runs where it
//     has the most lanes to fill: lanes = environments in the single-step kernels
//     (wf_step_group / wf_step_rows, phases A and C), the scalar unit in the n-step
//     kernel of one environment per wave (wf_step_ticks); round 1's wf_step_wave ran
//     it on the vector unit with 64 identical lanes.  Row-wrap and border tests use
//     per-lane VGPR constants, not 64-bit scalar constants.
//   * rarely needed parameters live in device memory (WfCold) and every rare branch
//     pins its inputs, so the optimiser cannot speculate rare-path work (or its
//     scalar registers) onto the hot path.
//   * n_steps fused steps keep the grid in registers; HBM traffic is one read and
//     one write of the grid per launch.
//
// No MFMA anywhere: the path is HBM-bound byte work, not a contraction.
#include <string.h>
#include <utility>

#include "wf_device.h"

namespace {

using namespace wfdev;

// ----------------------------------------------------------------------------
// compile-time geometry of the wave layout
// ----------------------------------------------------------------------------
constexpr u64 cm_mask(int W, int H, int j, int kind)
{
    u64 m = 0;
    for (int l = 0; l < W * H / 4; ++l) {
        int q = 4 * l + j, x = q % W, y = q / W;
        bool b = kind == 0 ? (x != 0)
               : kind == 1 ? (x != W - 1)
                           : (x == 0 || y == 0 || x == W - 1 || y == H - 1);
        if (b) m |= 1ull << l;
    }
    return m;
}

template <int W, int H, int J>
struct CM {
    static constexpr int CELLS = W * H;
    static constexpr int LANES = CELLS / 4;
    static constexpr u64 VALID = LANES == 64 ? ~0ull : ((1ull << LANES) - 1ull);
    static constexpr u64 NOT_X0 = cm_mask(W, H, J, 0);
    static constexpr u64 NOT_XL = cm_mask(W, H, J, 1);
    static constexpr u64 BORDER = cm_mask(W, H, J, 2);
    // cell q - W  ->  component UP_SRC of lane l - UP_SH
    static constexpr int UP_SRC = (((J - W) % 4) + 4) % 4;
    static constexpr int UP_SH = (UP_SRC - (J - W)) / 4;
    // cell q + W  ->  component DN_SRC of lane l + DN_SH
    static constexpr int DN_SRC = (J + W) % 4;
    static constexpr int DN_SH = (J + W) / 4;
};

// Neighbour masks of a per-component cell set S (bit l of S[j] <-> cell 4l+j).
template <int W, int H, int J>
__device__ __forceinline__ u64 nb_up(const u64 (&S)[4]) { return S[CM<W, H, J>::UP_SRC] << CM<W, H, J>::UP_SH; }
template <int W, int H, int J>
__device__ __forceinline__ u64 nb_dn(const u64 (&S)[4]) { return S[CM<W, H, J>::DN_SRC] >> CM<W, H, J>::DN_SH; }
template <int W, int H, int J>
__device__ __forceinline__ u64 nb_lf(const u64 (&S)[4])
{
    return (J > 0 ? S[(J + 3) & 3] : (S[3] << 1)) & CM<W, H, J>::NOT_X0;
}
template <int W, int H, int J>
__device__ __forceinline__ u64 nb_rt(const u64 (&S)[4])
{
    return (J < 3 ? S[(J + 1) & 3] : (S[0] >> 1)) & CM<W, H, J>::NOT_XL;
}

struct EnvS {        // wave-uniform scalars of one environment
    uint32_t ax, ay, fl, episode, tick;
};


// World.reset() into registers (environment.py:186-212, reset_map :59-67,
// get_agent_location utility.py:66-78) with the counter-based stream.
// t/u/h: type, fuel (unsigned byte), heat quanta of the lane's 4 cells.
template <int W, int H, bool RARE>
__device__ __forceinline__ void reset_regs(const WfCold *cold_g, u64 seed, uint32_t e, uint32_t lane,
                                           uint32_t (&t)[4], uint32_t (&u)[4], uint32_t (&h)[4], EnvS &s,
                                           u64 *reach_e)
{
    constexpr int CELLS = W * H;
    cold_ptr cold = (cold_ptr)(uintptr_t)cold_g;
    const int32_t fuel_lo = cold->p.fuel_lo, fuel_hi = cold->p.fuel_hi;
    const uint32_t ppm = cold->p.ignite_ppm;
    const EnvKey key(seed, cold->p.env_id0 + (u64)e, s.episode);     // e = index inside the context
    const uint32_t r = below(key.draw(0), 3u);                        // radius - 1
    const uint32_t k = below(key.draw(1), (uint32_t)cold->spawn.n[r]);   // point on the circle
    const uint32_t fx = W / 2, fy = H / 2;                            // utility.py:61-64
    const uint32_t sx = fx + (int)cold_g->spawn.dx[r][k], sy = fy + (int)cold_g->spawn.dy[r][k];
    const uint32_t qc = fy * W + fx, qs = sy * W + sx;
    const bool active = lane < (uint32_t)(CELLS / 4);
    const uint32_t span = (uint32_t)(fuel_hi - fuel_lo + 1);
    u64 onborder = 0;
    uint32_t water = 0;                                              // bit j: cell j of this lane is a river cell
    bool rivers = false;
    if constexpr (RARE) {
        rivers = cold->p.make_rivers != 0;                           // reset_map :69-95
        if (rivers)
            river_cells(key, W, H, [&](int x, int y) {
                const uint32_t d = (uint32_t)(y * W + x) - 4u * lane;
                water |= d < 4u ? 1u << d : 0u;
            });
    }
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const uint32_t q = 4u * lane + j;
        int32_t fuel = fuel_lo;
        if (span > 1u) fuel += (int32_t)below(key.draw(16u + q), span);
        uint32_t ty = ((water >> j) & 1u) ? (uint32_t)WF_WATER : (uint32_t)WF_GRASS, heat = 0;
        // set_fire_to :233-246; under a river the cell keeps its infinite fire mobility (:85)
        if (q == qc) { ty = ty == (uint32_t)WF_WATER ? (uint32_t)WF_FIRE_BLOCKED : (uint32_t)WF_FIRE; heat = WF_HEAT_FORCED; }
        if (q == qs) ty = WF_DIRT;                                    // Agent.__init__ digs :111-113
        bool extra = false;
        if (ppm) extra = ty == WF_GRASS && below(key.draw(16u + CELLS + q), 1000000u) < ppm;
        if (extra) { ty = WF_FIRE; heat = WF_HEAT_FORCED; }
        onborder |= __ballot(extra && active) & CM<W, H, j>::BORDER;
        t[j] = active ? ty : (uint32_t)WF_DIRT;                       // idle lanes: inert, impassable
        u[j] = active ? ((uint32_t)fuel & 0xffu) : 0u;
        h[j] = active ? heat : 0u;
    });
    // Border reachability of a fresh grid is known in closed form: every cell except the
    // agent's (dug) spawn cell - removing one cell cannot disconnect a grid - so the first
    // containment test of the episode needs no flood.  (Not with a river: the board stays invalid.)
    if (lane == 0) {
        u64 rb[4];
        static_for<4>([&](auto J) {
            constexpr int j = decltype(J)::value;
            rb[j] = CM<W, H, j>::VALID & ~(((qs & 3u) == (uint32_t)j) ? (1ull << (qs >> 2)) : 0ull);
        });
        ulonglong2 *rp = reinterpret_cast<ulonglong2 *>(reach_e);
        rp[0] = make_ulonglong2(rb[0], rb[1]);
        rp[1] = make_ulonglong2(rb[2], rb[3]);
    }
    s.ax = sx;
    s.ay = sy;
    s.fl = WF_F_ALIVE | WF_F_VISIBLE | WF_F_DIGGING | WF_F_RUNNING | (rivers ? 0u : (uint32_t)WF_F_REACH_VALID) | (onborder ? WF_F_BORDER : 0u);
    // World.reset :187-193: the episode's wind (wind with a direction only; the code rides in the upper half of the flags word)
    const int32_t wmode = cold->p.wind_mode;
    if (wmode != WF_WIND_CALM) {
        uint32_t code = (uint32_t)cold->p.wind_code;
        if (wmode == WF_WIND_RANDOM)                                  // :189-190, three uniform draws
            code = WF_WIND_CODE((int)below(key.draw(3), 3u) - 1, (int)below(key.draw(4), 3u) - 1, below(key.draw(2), 3u));
        s.fl |= code << 16;
    }
    s.episode += 1;
    s.tick = 0;
}

// Cells of P reachable from the seed set R through 4-neighbour steps inside P
// (scalar bitboard flood; wave-uniform loop).
template <int W, int H>
__device__ __forceinline__ void flood(u64 (&R)[4], const u64 (&P)[4])
{
    for (int it = 0; it < W * H; ++it) {
        u64 n[4];
        static_for<4>([&](auto J) {
            constexpr int j = decltype(J)::value;
            n[j] = (nb_up<W, H, j>(R) | nb_dn<W, H, j>(R) | nb_lf<W, H, j>(R) | nb_rt<W, H, j>(R)) & P[j] & ~R[j];
        });
        if (!(n[0] | n[1] | n[2] | n[3])) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) R[j] |= n[j];
    }
}

// A burning cell that dig() retyped to dirt (WF_DIRT_BURNING, allow_dig_toggle only) or that a river was drawn under
// (WF_FIRE_BLOCKED, make_rivers only) is impassable, yet the
// reference's search STARTS from it (the start node's own weight is never charged, pyastar/astar.cpp:41-116):
// it reaches the border iff one of its 4-neighbours does.  Z = those cells; R = the flooded board.
template <int W, int H>
__device__ __forceinline__ void add_blocked_burning(u64 (&R)[4], const u64 (&Z)[4])
{
    u64 n[4];
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        n[j] = (nb_up<W, H, j>(R) | nb_dn<W, H, j>(R) | nb_lf<W, H, j>(R) | nb_rt<W, H, j>(R)) & Z[j];
    });
#pragma unroll
    for (int j = 0; j < 4; ++j) R[j] |= n[j];
}

// ----------------------------------------------------------------------------
// fused observation: World.get_state() (environment.py:399-402) from registers
// ----------------------------------------------------------------------------
// obs[b][x][y][k] = (agent_pos, type == fire, fire_mobility != inf): the grid is transposed
// ([x][y] vs the planes' [y][x]) and 3 bytes per cell, so each lane drops its 4 cells'
// byte triples at their transposed offsets of a per-wave LDS image of the env's 3*W*H
// output bytes, and the image leaves as coalesced 16-byte stores.
template <int W, int H>
struct ObsGeo {
    static constexpr int BYTES = W * H * 3;
    static constexpr int PAD = (BYTES + 15) & ~15;
    static constexpr int FULL = BYTES / 16;          // lanes storing 16 bytes
    static constexpr int TAIL = BYTES % 16;          // bytes of the last partial lane (multiple of 4)
};

template <int W, int H>
__device__ __forceinline__ void obs_offsets(uint32_t lane, uint32_t (&oj)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t q = 4u * lane + (uint32_t)j, x = q % W, y = q / W;
        oj[j] = (x * H + y) * 3u;
    }
    asm volatile("" : "+v"(oj[0]), "+v"(oj[1]), "+v"(oj[2]), "+v"(oj[3]));
}

// c[j] = packed cell words (type in byte 0, agent_pos in byte 3); call with cell-owning lanes only
template <int W, int H>
__device__ __forceinline__ void emit_obs(uint8_t *lds, int8_t *obs_e, const uint32_t (&c)[4],
                                         const uint32_t (&oj)[4], uint32_t lane)
{
    using G = ObsGeo<W, H>;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t t = c[j] & 0xffu;
        lds[oj[j]] = (uint8_t)(c[j] >> 24);                       // agent_pos
        lds[oj[j] + 1u] = (uint8_t)((FIRE_SET >> t) & 1u);        // type == fire (fire, blocked fire)
        lds[oj[j] + 2u] = (uint8_t)((7u >> t) & 1u);              // not dirt / water / blocked
    }
    if (lane < (uint32_t)G::FULL) {
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + lane * 16u);
        // the env stride 3*W*H is only 4-byte aligned: a 16-byte store with 4-byte alignment
        struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
        *reinterpret_cast<U4 *>(obs_e + lane * 16u) = U4{ v.x, v.y, v.z, v.w };
    } else if (G::TAIL && lane == (uint32_t)G::FULL) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(lds + lane * 16u);
        uint32_t *dst = reinterpret_cast<uint32_t *>(obs_e + lane * 16u);
#pragma unroll
        for (int k = 0; k < G::TAIL / 4; ++k) dst[k] = src[k];
    }
}

#ifndef WF_WAVE_SHAPES   // (a kernel experiment may build one shape only: -D'WF_WAVE_SHAPES(X)=X(10,10)')
#define WF_WAVE_SHAPES(X) X(8, 8) X(10, 10) X(12, 12) X(14, 14) X(16, 16) X(16, 8) X(8, 16) X(20, 10) X(10, 20)
#endif

#include "wf_wave_fused.h"   // wf_step_wave: n fused steps, one environment per wave (round-1 formulation, WF_FUSED_LEGACY=1)
#include "wf_wave_ticks.h"   // wf_step_ticks: n fused steps, one environment per wave, packed cells + scalar agent

// This file is built as TWO translation units so that its halves compile in parallel (the Makefile passes -DWF_TU=1 for
// the second): WF_TU == 1 holds the n-step kernels of one environment per wave and their launcher, WF_TU == 0 everything else.
#ifndef WF_TU
#define WF_TU 0
#endif
#if WF_TU == 1

}  // namespace

hipError_t wf_launch_nstep_wave(const WfArgs &a, const WfKnobs &k, hipStream_t s, int blocks)
{
    // dig toggle / rivers (off in the reference's defaults) run in their own instantiations of the step kernels
    const bool rare = (a.hot_flags & WF_HOT_RARE) != 0;
#define X(w, h)                                                                                                          \
    if (a.W == w && a.H == h) {                                                                                          \
        if (k.fused_legacy) {                                                                                            \
            if (rare) hipLaunchKernelGGL((wf_step_wave<w, h, true, true>), dim3(blocks), dim3(256), 0, s, a);            \
            else hipLaunchKernelGGL((wf_step_wave<w, h, true, false>), dim3(blocks), dim3(256), 0, s, a);                \
        } else {                                                                                                         \
            if (rare) hipLaunchKernelGGL((wf_step_ticks<w, h, true>), dim3(blocks), dim3(256), 0, s, a);                 \
            else hipLaunchKernelGGL((wf_step_ticks<w, h, false>), dim3(blocks), dim3(256), 0, s, a);                     \
        }                                                                                                                \
        return hipGetLastError();                                                                                        \
    }
    WF_WAVE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

#else   // WF_TU == 0

#include "wf_wave_group.h"   // wf_step_group + phases A / C
#include "wf_wave_rows.h"    // wf_step_rows

// World.reset() for masked envs (environment.py:186-212)
template <int W, int H>
__global__ __launch_bounds__(256) void wf_reset_wave(const WfArgs a)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    const uint32_t lane = __lane_id();
    const uint32_t wave0 = rfl(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * 4u;
    for (uint32_t e = wave0; e < (uint32_t)a.batch; e += nwaves) {
        if (a.mask && !rfl((uint32_t)a.mask[e])) continue;
        const uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);
        EnvS s;
        s.ax = s.ay = 0;
        s.fl = 0;
        s.episode = rfl(ev.z);
        s.tick = 0;
        uint32_t t[4], u[4], h[4];
        reset_regs<W, H, true>(a.cold, a.reset_seed, e, lane, t, u, h, s, a.reach + (size_t)e * 4);
        const uint32_t qa = s.ay * W + s.ax;
        uint32_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = t[j] | (u[j] << 8) | (h[j] << 16) |
                   ((lane == (qa >> 2) && (qa & 3u) == (uint32_t)j) ? (1u << 24) : 0u);
        if (lane < (uint32_t)LANES) {
            reinterpret_cast<uint4 *>(a.planes + (size_t)e * CELLS)[lane] = make_uint4(r[0], r[1], r[2], r[3]);
            if (a.wind_cnt) reinterpret_cast<uint4 *>(a.wind_cnt + (size_t)e * CELLS)[lane] = make_uint4(0, 0, 0, 0);   // no heat received yet
        }
        if (lane == 0)
            *reinterpret_cast<uint4 *>(&a.envs[e]) = make_uint4(s.ax | (s.ay << 16), s.fl, s.episode, s.tick);
    }
}

// World.get_state() for the wave family: one env per wave iteration, LDS transpose, 16-byte stores.
template <int W, int H>
__global__ __launch_bounds__(256) void wf_get_state_wave(const uint32_t *__restrict__ planes, int8_t *__restrict__ obs, int batch)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    __shared__ __attribute__((aligned(16))) uint8_t sObs[4][ObsGeo<W, H>::PAD];
    uint8_t *obs_lds = sObs[threadIdx.x >> 6];
    const uint32_t lane = __lane_id();
    const uint32_t wave0 = rfl(blockIdx.x * 4u + (threadIdx.x >> 6)), nwaves = gridDim.x * 4u;
    uint32_t oj[4];
    obs_offsets<W, H>(lane, oj);
    if (lane < (uint32_t)LANES) {
        for (uint32_t e = wave0; e < (uint32_t)batch; e += nwaves) {
            const uint4 raw = reinterpret_cast<const uint4 *>(planes)[(size_t)e * LANES + lane];
            const uint32_t c[4] = { raw.x, raw.y, raw.z, raw.w };
            emit_obs<W, H>(obs_lds, obs + (size_t)e * (CELLS * 3), c, oj, lane);
        }
    }
}

// World.get_state() (environment.py:399-402): obs[b][x][y][k], any grid shape: an LDS-tiled transpose.
// obs is planes with [y][x] swapped and 3 bytes per cell instead of 4, so the columns X = b*W + x of the WHOLE batch
// form one linear sequence of 3H-byte runs.  A workgroup takes TXG consecutive columns (a power of two; they may span
// environments):
//   phase 1  lanes run along X: row pieces of the planes come in as coalesced dword loads (a 128-byte line per 32
//            lanes), each cell becomes the 24-bit triple (agent_pos, type == fire, fire_mobility != inf) and goes to
//            LDS column-major, tile[col * Hp + y] with Hp odd (lanes of consecutive columns hit different banks);
//   phase 2  the workgroup's output is ONE contiguous byte range.  A thread packs the triples of 4-5 consecutive cells
//            of the linear (col, y) sequence into 12 output bytes (three funnel shifts by the range's byte misalignment,
//            which is workgroup-uniform) and writes them as one 4-byte aligned dwordx3 store; the few head and tail
//            bytes of the range go out as byte stores.
// 7 B of traffic per cell (4 read + 3 written).  The kernel it replaces ran one thread per output triple with 64-bit
// div / mod and read the planes at a stride of 4W bytes: 0.65 TB/s at 1 024 x 256x256.
#ifndef WF_GS_BUDGET
#define WF_GS_BUDGET 9216    // wf_get_state_cols: dwords of LDS a workgroup's tile may take (36 KiB: four workgroups per CU; 18 and 64 KiB measured slower)
#endif
__device__ __forceinline__ uint32_t obs_triple(uint32_t c)
{
    const uint32_t t = c & 0xffu;
    return (c >> 24) | (((FIRE_SET >> t) & 1u) << 8) | ((t < (uint32_t)WF_DIRT ? 1u : 0u) << 16);
}

__global__ __launch_bounds__(256) void wf_get_state_cols(const uint32_t *__restrict__ planes, int8_t *__restrict__ obs,
                                                        uint32_t ncols, uint32_t H, uint32_t W, uint32_t txg_log2)
{
    extern __shared__ uint32_t tile[];                       // [TXG][Hp] triples
    const uint32_t TXG = 1u << txg_log2, Hp = H | 1u, pad = Hp - H;
    const uint32_t X0 = blockIdx.x << txg_log2;
    const uint32_t nX = min(TXG, ncols - X0);
    // ---- phase 1 (U independent row loads in flight per thread: 8 for tall grids - 1 024 x 256x256 109 -> 99 us -, 4 for
    // short ones, whose columns a thread finishes in one or two trips anyway)
    {
        const uint32_t lx = threadIdx.x & (TXG - 1u), ly = threadIdx.x >> txg_log2, ystep = 256u >> txg_log2;
        if (lx < nX) {
            const uint32_t X = X0 + lx, b = X / W, x = X - b * W;
            const uint32_t *src = planes + (size_t)b * H * W + x;
            uint32_t *dst = tile + lx * Hp;
            auto rows = [&](auto UU) {
                constexpr int U = decltype(UU)::value;
                for (uint32_t y = ly; y < H; y += ystep * (uint32_t)U) {
                    uint32_t v[U];
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const uint32_t yy = y + (uint32_t)k * ystep;
                        v[k] = yy < H ? src[(size_t)yy * W] : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const uint32_t yy = y + (uint32_t)k * ystep;
                        if (yy < H) dst[yy] = obs_triple(v[k]);
                    }
                }
            };
            if (H >= 8u * ystep) rows(std::integral_constant<int, 8>{});
            else rows(std::integral_constant<int, 4>{});
        }
    }
    __syncthreads();
    // ---- phase 2: cells n = col * H + y of the workgroup, linear; LDS address n + (n / H) * pad
    int8_t *out = obs + (size_t)X0 * H * 3u;
    const uint32_t len = nX * H * 3u;
    const uint32_t a0 = min((uint32_t)(-(intptr_t)out) & 3u, len);   // head bytes up to the first aligned dword
    const uint32_t nchunk = (len - a0) / 12u;                        // 12-byte chunks = 4 cells (+ the shift)
    const uint32_t ncells = nX * H;
    // (a chunk's fifth cell may lie past the range: clamped, its bytes are shifted out)
    auto cell = [&](uint32_t n, uint32_t col) -> uint32_t { return tile[min(n, ncells - 1u) + min(col, nX - 1u) * pad]; };
    const uint32_t s8 = (a0 % 3u) * 8u, nbase = a0 / 3u;             // chunk i starts at byte 12 i + a0 = cell 4 i + nbase, byte a0 % 3
    {
        uint32_t i = threadIdx.x;
        uint32_t n = 4u * i + nbase;
        uint32_t col = n / H, y = n - col * H;                       // one division per thread; then incremental
        const uint32_t dcol = 1024u / H, dy = 1024u - dcol * H;
        for (; i < nchunk; i += 256u) {
            uint32_t w[5];
            uint32_t cc = col, yy = y;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                w[k] = cell(n + (uint32_t)k, cc);
                if (++yy == H) { yy = 0; ++cc; }
            }
            const uint32_t B0 = w[0] | (w[1] << 24), B1 = (w[1] >> 8) | (w[2] << 16), B2 = (w[2] >> 16) | (w[3] << 8), B3 = w[4];
            struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };
            *reinterpret_cast<U3 *>(out + a0 + 12u * i) =
                U3{ __funnelshift_r(B0, B1, s8), __funnelshift_r(B1, B2, s8), __funnelshift_r(B2, B3, s8) };
            n += 1024u;
            col += dcol;
            y += dy;
            if (y >= H) { y -= H; ++col; }
        }
    }
    // head and tail bytes (fewer than 16 per workgroup)
    auto byte_at = [&](uint32_t o) -> int8_t {
        const uint32_t n = o / 3u, k = o - n * 3u;
        return (int8_t)(tile[n + (n / H) * pad] >> (8u * k));
    };
    const uint32_t tail0 = a0 + 12u * nchunk;
    if (threadIdx.x < a0) out[threadIdx.x] = byte_at(threadIdx.x);
    if (tail0 + threadIdx.x < len) out[tail0 + threadIdx.x] = byte_at(tail0 + threadIdx.x);
}

// fallback for grids too tall for the tile (H >= 16 000): one thread per output triple
__global__ __launch_bounds__(256) void wf_get_state_kernel(const uint32_t *__restrict__ planes,
                                                          int8_t *__restrict__ obs, int batch, int H, int W)
{
    const size_t cells = (size_t)H * (size_t)W;
    const size_t total = (size_t)batch * cells;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / cells, r = i % cells;
        const uint32_t x = (uint32_t)(r / H), y = (uint32_t)(r % H);
        const uint32_t tr = obs_triple(planes[b * cells + (size_t)y * W + x]);
        int8_t *o = obs + i * 3;
        o[0] = (int8_t)tr;
        o[1] = (int8_t)(tr >> 8);
        o[2] = (int8_t)(tr >> 16);
    }
}

// Host uploads replace planes/scalars behind the context's back: the cached
// border-reachability bitboards are no longer trustworthy.
// small device -> pinned-host copies of wf_download (a kernel costs a launch; a DMA operation per array costs more)
__global__ __launch_bounds__(256) void wf_copy_words_kernel(const uint32_t *src, uint32_t *dst, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
}

// two ranges in one launch (planes + records of wf_download's small path: a launch costs more than the bytes)
__global__ __launch_bounds__(256) void wf_copy_words2_kernel(const uint32_t *src0, uint32_t *dst0, uint32_t n0,
                                                            const uint32_t *src1, uint32_t *dst1, uint32_t n1)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n0 + n1; i += gridDim.x * 256u) {
        if (i < n0) dst0[i] = src0[i];
        else dst1[i - n0] = src1[i - n0];
    }
}

__global__ __launch_bounds__(256) void wf_copy_quads_kernel(const uint4 *src, uint4 *dst, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void wf_invalidate_reach_kernel(wf_env *envs, int batch)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < batch) envs[i].flags &= (uint16_t)~(WF_F_REACH_VALID | WF_F_REACH_ROWS);
}

// Which kernel serves a single step (n_steps == 1) of `a.batch` environments.
//   G     environments per wave, sized so that every resident wave slot of the chip (32 per CU) gets
//         one group (a.group > 0: fixed by a split launch or the WF_GROUP experiment override)
//   rows  layout with several environments per wave iteration, where the shape has an instantiation
//         (WF_NO_ROWS=1 keeps the 4-cells-per-lane group layout)
//   shared (WF_SHARED=1, opt-in) a workgroup shares the agent phases A and C over 64 environments:
//         fewer instructions (-10..-20 %), but the barrier costs as much as it saves once two
//         half-batch launches overlap (measured 23.3 vs 22.4 us per 65536x14x14 step)
enum { STEP_GROUP = 0, STEP_GROUP_SHARED = 1, STEP_ROWS = 2, STEP_ROWS_SHARED = 3 };
struct StepPlan { int G, kind; };

template <int W, int H> struct HasRows { static constexpr bool value = false; static constexpr int S = 1; };
template <> struct HasRows<14, 14> { static constexpr bool value = true; static constexpr int S = 2; };   // 7 cells/lane, 2 envs/iteration
template <> struct HasRows<10, 10> { static constexpr bool value = true; static constexpr int S = 2; };   // 5 cells/lane, 2 envs/iteration
template <> struct HasRows<12, 12> { static constexpr bool value = true; static constexpr int S = 2; };   // 6 cells/lane, 2 envs/iteration
// shapes whose layout puts an ODD number of environments side by side: groups (and split boundaries) are multiples of it
template <> struct HasRows<8, 8> { static constexpr bool value = true; static constexpr int S = 1; };     // 8 cells/lane, 7 envs/iteration
template <> struct HasRows<16, 8> { static constexpr bool value = true; static constexpr int S = 2; };    // 8 cells/lane, 3 envs/iteration
template <> struct HasRows<8, 16> { static constexpr bool value = true; static constexpr int S = 1; };    // 8 cells/lane, 3 envs/iteration
// (16x16 and 20x10 would need 5.3 / 5 cell slots per environment against the group layout's 4: no rows kernel; 10x20 with
// 10 cells per lane and 3 environments per iteration wants 20 mask pairs in scalar registers and measured 24.6 us per
// 65 536-env step against the group layout's 19.6)

template <int W, int H>
StepPlan plan_single(const WfArgs &a, const WfKnobs &k, int num_cus)
{
    StepPlan p;
    p.G = wf_pick_group_for(W, H, num_cus, a.batch, k);
    const int gmax = wf_pick_group(W * H, num_cus, 1ll << 40);
    if (a.group > 0) p.G = a.group < gmax ? a.group : gmax;
    if constexpr (HasRows<W, H>::value) {
        // an n-step call: at least one iteration's worth of environments per wave, so that the rows kernel's step loop
        // can serve it (8 192 x 10x10 x 16 ticks: 48 us with 2 environments per wave against 104 as 16 single steps)
        constexpr int EPW0 = RG<W, H, HasRows<W, H>::S>::EPW;
        if (a.n_steps > 1 && p.G < EPW0) p.G = EPW0;
    }
    // (the workgroup-shared experiment exists for the reference's default options only)
    const bool wind = (a.hot_flags & WF_HOT_WIND) != 0;      // wind with a direction: the group layout's WIND instantiation
    const bool shared = k.shared && (p.G == 2 || p.G == 4 || p.G == 8 || p.G == 16) && a.batch >= 4096 && !(a.hot_flags & WF_HOT_RARE) && !wind;
    p.kind = shared ? STEP_GROUP_SHARED : STEP_GROUP;
    if (wind) return p;
    if constexpr (HasRows<W, H>::value) {
        constexpr int EPW = RG<W, H, HasRows<W, H>::S>::EPW;
        // (wf_pick_group_for keeps G <= 16 for these shapes; an override beyond that needs WF_ROWS=1 to stay on rows)
        if (p.G % EPW == 0 && (p.G <= 16 || (p.G <= 32 && k.rows)) && a.env_base % EPW == 0 && !k.no_rows)
            p.kind = shared ? STEP_ROWS_SHARED : STEP_ROWS;
    }
    return p;
}

template <int W, int H>
hipError_t launch_wave(bool reset, const WfArgs &a, const WfKnobs &k, hipStream_t s, int num_cus)
{
    int blocks = (a.batch + 3) / 4;
    const int cap = num_cus * 8;            // up to 32 resident waves per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    // dig toggle / rivers (off in the reference's defaults) run in their own instantiations of the step kernels
    const bool rare = (a.hot_flags & WF_HOT_RARE) != 0;
    if (reset)
        hipLaunchKernelGGL((wf_reset_wave<W, H>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.n_steps > 1 && !k.multi_rows) {
        if (a.hot_flags & WF_HOT_WIND) return hipErrorInvalidValue;     // (wind: the host issues n single-step launches)
        return wf_launch_nstep_wave(a, k, s, blocks);                   // wf_step_ticks (the file's other translation unit)
    } else {
        const StepPlan p = plan_single<W, H>(a, k, num_cus);
        if (a.n_steps > 1 && p.kind != STEP_ROWS) return hipErrorInvalidValue;   // only the rows kernel has the step loop
        WfArgs g = a;
        g.group = p.G;
        const int waves = (a.batch + p.G - 1) / p.G;
        // a launch that steps the whole batch alone with one generation of waves (at most the chip's 32 per CU, at least half
        // of them): its waves lower their issue priority as they get ahead (set_progress_prio; WF_NO_PRIO=1 switches it off)
        const bool prio = k.whole_batch && !k.no_prio && a.n_steps <= 1 && waves <= num_cus * 32 && waves >= num_cus * 12;
        const dim3 grid_private((waves + 3) / 4), grid_shared((a.batch + 63) / 64), block_shared(64 * (64 / p.G));
        if constexpr (HasRows<W, H>::value) {
            constexpr int S = HasRows<W, H>::S;
            const dim3 grid_rows((waves + WF_ROWS_WPB - 1) / WF_ROWS_WPB), block_rows(64 * WF_ROWS_WPB);
            // step loop: LDS-resident when the wave's environments fit its mirror (WF_NO_RESIDENT: the plain loop)
            const bool res = !k.no_resident && p.G <= WF_RES_MAXG;
            const size_t mirror_bytes = (size_t)WF_ROWS_WPB * p.G * W * H * 4;
            if (p.kind == STEP_ROWS && a.n_steps > 1 && !rare && res) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, true, false, true>), grid_rows, block_rows, mirror_bytes, s, g);
            if (p.kind == STEP_ROWS && a.n_steps > 1 && rare && res) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, true, true, true>), grid_rows, block_rows, mirror_bytes, s, g);
            if (p.kind == STEP_ROWS && a.n_steps > 1 && !rare && !res) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, true, false>), grid_rows, block_rows, 0, s, g);
            if (p.kind == STEP_ROWS && a.n_steps > 1 && rare && !res) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, true, true>), grid_rows, block_rows, 0, s, g);
            const bool rev = a.sweep_rev != 0 && !rare && !prio;
            if (p.kind == STEP_ROWS && a.n_steps <= 1 && !rare && !prio && !rev) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, false, false>), grid_rows, block_rows, 0, s, g);
            if (p.kind == STEP_ROWS && a.n_steps <= 1 && rev) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, false, false, false, false, true>), grid_rows, block_rows, 0, s, g);
            if (p.kind == STEP_ROWS && a.n_steps <= 1 && !rare && prio) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, false, false, false, true>), grid_rows, block_rows, 0, s, g);
            if (p.kind == STEP_ROWS && a.n_steps <= 1 && rare) hipLaunchKernelGGL((wf_step_rows<W, H, S, false, false, true>), grid_rows, block_rows, 0, s, g);
            if (p.kind == STEP_ROWS_SHARED) {
                // WF_SHARED=N with N in {2,4,8}: N waves share the agent phases (N*G <= 64 environments per workgroup)
                int nw = k.shared;
                if (nw < 2 || nw * p.G > 64) nw = 64 / p.G;
                const int per_wg = nw * p.G;
                hipLaunchKernelGGL((wf_step_rows<W, H, S, true, false, false>), dim3((a.batch + per_wg - 1) / per_wg), dim3(64 * nw), 0, s, g);
            }
        }
        const bool wind = (a.hot_flags & WF_HOT_WIND) != 0;
        if (p.kind == STEP_GROUP && !rare && !wind) hipLaunchKernelGGL((wf_step_group<W, H, false, false>), grid_private, dim3(256), 0, s, g);
        if (p.kind == STEP_GROUP && rare && !wind) hipLaunchKernelGGL((wf_step_group<W, H, false, true>), grid_private, dim3(256), 0, s, g);
        if (p.kind == STEP_GROUP && !rare && wind) hipLaunchKernelGGL((wf_step_group<W, H, false, false, true>), grid_private, dim3(256), 0, s, g);
        if (p.kind == STEP_GROUP && rare && wind) hipLaunchKernelGGL((wf_step_group<W, H, false, true, true>), grid_private, dim3(256), 0, s, g);
        if (p.kind == STEP_GROUP_SHARED) hipLaunchKernelGGL((wf_step_group<W, H, true, false>), grid_shared, block_shared, 0, s, g);
    }
    return hipGetLastError();
}


hipError_t dispatch_wave(bool reset, const WfArgs &a, const WfKnobs &k, hipStream_t s, int num_cus)
{
#define X(w, h) if (a.W == w && a.H == h) return launch_wave<w, h>(reset, a, k, s, num_cus);
    WF_WAVE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace

bool wf_tiled_serves(int H, int W) { return W >= 4 && W <= 256 && W % 4 == 0 && H >= 1; }

WfFamily wf_pick_family(int H, int W)
{
#define X(w, h) if (W == w && H == h) return WF_FAM_WAVE;
    WF_WAVE_SHAPES(X)
#undef X
    // a wave of the tiled family has a fixed part worth about five rows of sweep, and narrow grids leave lanes idle:
    // below 24x24 the generic family's wave-per-environment kernel is faster or as fast (us per step, generic against
    // the tiled family's one-launch step with eight strips per wave, round 4: 32 768 x 20x20 49 against 53 - 4 096: 19
    // against 15 -, 32 768 x 24x24 61 against 54 - 4 096: 21 against 15 -, 16 384 x 28x28 44 against 30)
    if (W >= 4 && W <= 256 && W % 4 == 0 && H >= 1 && (long long)H * W > WF_TILED_MIN_CELLS) return WF_FAM_TILED;
    return WF_FAM_CELLS;
}

void wf_read_knobs(WfKnobs *k)
{
    auto num = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
    auto set = [](const char *name) { return getenv(name) != nullptr; };
    k->group = num("WF_GROUP", 0);
    k->split_min = num("WF_SPLIT_MIN", 0);
    k->no_split = set("WF_NO_SPLIT");
    k->n_parts = num("WF_SPLIT", 0);      // 0: automatic (wf_create: 2 for the wave family, 3 for the big-grid families)
    k->shared = set("WF_SHARED") ? (num("WF_SHARED", 1) > 1 ? num("WF_SHARED", 1) : 1) : 0;
    k->rows = set("WF_ROWS");
    k->no_rows = set("WF_NO_ROWS");
    k->no_multi = set("WF_NO_MULTI");
    k->multi_small = set("WF_MULTI_SMALL");
    k->cells_legacy = set("WF_CELLS_LEGACY");
    const char *fam = getenv("WF_FAMILY");
    k->family = !fam ? 0 : !strcmp(fam, "cells") ? 1 : !strcmp(fam, "tiled") ? 2 : 0;
    k->no_resident = !set("WF_RESIDENT");     // opt-in experiment (measured: no gain, profiles/r03_c2_variants.txt)
    k->graph = set("WF_GRAPH");
    k->host_stage_min = num("WF_HOST_STAGE_MIN", 0);
    k->flag_wait = num("WF_FLAG_WAIT", 0);
    k->fused_legacy = set("WF_FUSED_LEGACY");
    k->sweep_alt = num("WF_SWEEP_ALT", 1);     // (1 024 x 256x256: 82 against 86 us per step; tools/sweep_alt.sh)
    k->host_stage_mode = num("WF_HOST_STAGE_MODE", 0);
    k->ticks_split = num("WF_TICKS_SPLIT", 2048);
    k->split_explicit = set("WF_SPLIT") ? 1 : 0;
    k->multi_rows = 0;
    k->whole_batch = 0;
    k->no_prio = set("WF_NO_PRIO");
    k->no_onegen = set("WF_NO_ONEGEN");
    k->no_solo = set("WF_NO_SOLO");
    k->no_solo_loop = set("WF_NO_SOLO_LOOP");
    k->solo_loop = set("WF_SOLO_LOOP");
    k->tiled_rows = num("WF_TILED_ROWS", 0);
}

// environments per wave iteration of the shape's rows-layout kernel (0: the shape has none, or WF_NO_ROWS)
int wf_rows_epw(int W, int H, const WfKnobs &k)
{
    int epw = 0;
#define X(w, h) if (W == w && H == h) { if constexpr (HasRows<w, h>::value) epw = RG<w, h, HasRows<w, h>::S>::EPW; }
    WF_WAVE_SHAPES(X)
#undef X
    return k.no_rows ? 0 : epw;
}

int wf_pick_group_for(int W, int H, int num_cus, long long batch, const WfKnobs &k)
{
    int G = wf_pick_group(W * H, num_cus, batch);          // a power of two
    const int epw = wf_rows_epw(W, H, k);
    if (epw) {
        if (G > 8) G = 8;
        if (G >= epw) G = G / epw * epw;                   // a whole number of iterations: 8 -> 6 (3 per iteration), 8 -> 7
    }
    return G;
}

// Split launches cut the batch at multiples of this many environments: a wave's first environment must start an
// iteration of the rows layout in every part (the boards are indexed per iteration of the whole context).
int wf_split_align(int W, int H, const WfKnobs &k)
{
    const int epw = wf_rows_epw(W, H, k);
    return (epw > 1 && (epw & 1)) ? 64 * epw : 64;
}

const char *wf_family_name(WfFamily f, int, int)
{
    switch (f) {
    case WF_FAM_WAVE: return "wf_step_rows/wf_step_group/wf_step_ticks (whole grids in a wave's registers, CA core on ballot bitboards; env logic lane-parallel or scalar)";
    case WF_FAM_TILED: return "wf_step_tiled (16-row strips per wave, ballot row masks, LDS-staged strip halos)";
    case WF_FAM_CELLS: return "wf_step_env / wf_step_cells (generic: a wave or a workgroup per environment; thread-per-cell for a few huge grids)";
    default: return "none";
    }
}

// Large batches of a shape with a rows-layout kernel: an n_steps call is ONE launch of that kernel, every wave
// stepping its environments n_steps times (small batches: the fused one-environment-per-wave kernel).
bool wf_multi_step_in_rows(WfFamily fam, const WfArgs &a, const WfKnobs &k, int num_cus)
{
    // (below 4 096 environments per launch the one-environment-per-wave kernel has more waves to hide its latency with:
    // 4 096 x 10x10 x 16 ticks 37.7 us against 42.1 on the step loop with 2 environments per wave)
    if (fam != WF_FAM_WAVE || a.step_stride <= 0 || a.batch < 4096 || k.no_multi || (a.hot_flags & WF_HOT_WIND)) return false;
    StepPlan p = { 0, -1 };
#define X(w, h) if (a.W == w && a.H == h) p = plan_single<w, h>(a, k, num_cus);
    WF_WAVE_SHAPES(X)
#undef X
    return p.kind == STEP_ROWS;
}

// symbol of the kernel that would serve a single step of `a` (introspection for bench / profiles)
const char *wf_single_step_kernel(WfFamily fam, const WfArgs &a, const WfKnobs &k, int num_cus)
{
    static thread_local char buf[64];
    if (fam == WF_FAM_TILED) return !k.no_solo && wf_tiled_solo_rows(a.H, a.W, 0, a.batch) ? "wf_step_tiled<solo>" : "wf_step_tiled";
    if (fam == WF_FAM_CELLS) return wf_cells_per_env_kernel(a.batch, a.H, a.W, k) ? "wf_step_env" : "wf_step_cells";
    StepPlan p = { 0, -1 };
    int S = 1;
#define X(w, h) if (a.W == w && a.H == h) { p = plan_single<w, h>(a, k, num_cus); S = HasRows<w, h>::S; }
    WF_WAVE_SHAPES(X)
#undef X
    if (p.kind < 0) return "none";
    if (p.kind >= STEP_ROWS) snprintf(buf, sizeof buf, "wf_step_rows<%d,%d,%d,%s>", a.W, a.H, S, p.kind == STEP_ROWS_SHARED ? "true" : "false");
    else if (a.hot_flags & WF_HOT_WIND) snprintf(buf, sizeof buf, "wf_step_group<%d,%d,wind>", a.W, a.H);
    else snprintf(buf, sizeof buf, "wf_step_group<%d,%d,%s>", a.W, a.H, p.kind == STEP_GROUP_SHARED ? "true" : "false");
    return buf;
}

// symbol of the kernel behind an n_steps call whose verdict (WfKnobs::multi_rows) says "rows step loop"
const char *wf_multi_step_kernel(const WfArgs &a, const WfKnobs &k, int num_cus)
{
    static thread_local char buf[96];
    StepPlan p = { 0, -1 };
    int S = 1;
#define X(w, h) if (a.W == w && a.H == h) { p = plan_single<w, h>(a, k, num_cus); S = HasRows<w, h>::S; }
    WF_WAVE_SHAPES(X)
#undef X
    if (p.kind != STEP_ROWS) return "none";
    snprintf(buf, sizeof buf, "wf_step_rows<%d,%d,%d> step loop, %s, %d environments per wave", a.W, a.H, S,
             (!k.no_resident && p.G <= WF_RES_MAXG) ? "LDS-resident" : "through L2", p.G);
    return buf;
}

hipError_t wf_launch_step(WfFamily fam, const WfArgs &a, const WfKnobs &k, hipStream_t s, int num_cus)
{
    if (fam == WF_FAM_WAVE) return dispatch_wave(false, a, k, s, num_cus);
    return hipErrorInvalidValue;
}

hipError_t wf_launch_reset(WfFamily fam, const WfArgs &a, hipStream_t s, int num_cus)
{
    if (fam == WF_FAM_WAVE) { WfKnobs k = {}; return dispatch_wave(true, a, k, s, num_cus); }
    return hipErrorInvalidValue;
}

hipError_t wf_launch_copy_words(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    if (bytes >= 65536 && bytes % 16 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
        const uint32_t nq = (uint32_t)(bytes / 16);          // large copies: 16 bytes per lane
        unsigned blocks = (nq + 255u) / 256u;
        if (blocks > 1024u) blocks = 1024u;
        hipLaunchKernelGGL(wf_copy_quads_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), nq);
        return hipGetLastError();
    }
    const uint32_t n = (uint32_t)(bytes / 4);
    unsigned blocks = (n + 255u) / 256u;
    if (blocks > 1024u) blocks = 1024u;
    if (n) hipLaunchKernelGGL(wf_copy_words_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint32_t *>(src), static_cast<uint32_t *>(dst), n);
    return hipGetLastError();
}

hipError_t wf_launch_copy_words2(const void *src0, void *dst0, size_t bytes0, const void *src1, void *dst1, size_t bytes1, hipStream_t s)
{
    const uint32_t n0 = (uint32_t)(bytes0 / 4), n1 = (uint32_t)(bytes1 / 4);
    unsigned blocks = (n0 + n1 + 255u) / 256u;
    if (blocks > 1024u) blocks = 1024u;
    if (n0 + n1) hipLaunchKernelGGL(wf_copy_words2_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint32_t *>(src0), static_cast<uint32_t *>(dst0), n0,
                                    static_cast<const uint32_t *>(src1), static_cast<uint32_t *>(dst1), n1);
    return hipGetLastError();
}

// Completion flag of a synchronous host-buffer call: stream order puts it behind the step (and its copy kernels), the
// host spins on the pinned word instead of sleeping on the queue's completion signal.
__global__ void wf_flag_kernel(uint32_t *flag, uint32_t value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t wf_launch_flag(uint32_t *flag, uint32_t value, hipStream_t s)
{
    hipLaunchKernelGGL(wf_flag_kernel, dim3(1), dim3(1), 0, s, flag, value);
    return hipGetLastError();
}

hipError_t wf_launch_invalidate_reach(wf_env *envs, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(wf_invalidate_reach_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, envs, batch);
    return hipGetLastError();
}

hipError_t wf_launch_get_state(const uint32_t *planes, int8_t *obs, int batch, int H, int W, hipStream_t s)
{
#define X(w, h)                                                                                         \
    if (W == w && H == h) {                                                                             \
        int blocks = (batch + 3) / 4;                                                                   \
        if (blocks > 2048) blocks = 2048;                                                               \
        hipLaunchKernelGGL((wf_get_state_wave<w, h>), dim3(blocks), dim3(256), 0, s, planes, obs, batch); \
        return hipGetLastError();                                                                       \
    }
    WF_WAVE_SHAPES(X)
#undef X
    // LDS-tiled transpose: TXG columns of Hp = H | 1 triples per workgroup, up to 36 KiB of LDS (4 workgroups per CU)
    const unsigned long long ncols = (unsigned long long)batch * W;
    const unsigned Hp = (unsigned)H | 1u;
    if (Hp <= 16000u && ncols < (1ull << 31) && (unsigned long long)H * 3ull * 64ull < (1ull << 31)) {
        unsigned lg = 0;
        while (lg < 6u && (2u << lg) * Hp <= (unsigned)WF_GS_BUDGET) ++lg;   // largest TXG = 2^lg <= 64 whose tile fits
        while (lg > 0u && (ncols >> lg) < 1024ull) --lg;             // small jobs: more, smaller workgroups
        const unsigned nblk = (unsigned)((ncols + (1ull << lg) - 1) >> lg);
        hipLaunchKernelGGL(wf_get_state_cols, dim3(nblk), dim3(256), (size_t)(1u << lg) * Hp * 4u, s, planes, obs,
                           (uint32_t)ncols, (uint32_t)H, (uint32_t)W, (uint32_t)lg);
        return hipGetLastError();
    }
    const size_t total = (size_t)batch * H * W;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wf_get_state_kernel, dim3((unsigned)blocks), dim3(256), 0, s, planes, obs, batch, H, W);
    return hipGetLastError();
}

#endif   // WF_TU
