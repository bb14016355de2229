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
//   * work split: the scalar unit (one instruction per clock for the whole CU) does
//     ONLY the 64-bit mask algebra and the rare uniform branches; the per-cell rule
//     and the per-environment flag logic (agent move/dig/death, RUNNING, reward)
//     run on the vector unit, the latter branch-free on values that are identical
//     in all 64 lanes.  Row-wrap and border tests use per-lane VGPR constants, not
//     64-bit scalar constants.
//   * rarely needed parameters live in device memory (WfCold) and every rare branch
//     pins its inputs, so the optimiser cannot speculate rare-path work (or its
//     scalar registers) onto the hot path.
//   * n_steps fused steps keep the grid in registers; HBM traffic is one read and
//     one write of the grid per launch.
//
// No MFMA anywhere: the path is HBM-bound byte work, not a contraction.
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
template <int W, int H>
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
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const uint32_t q = 4u * lane + j;
        int32_t fuel = fuel_lo;
        if (span > 1u) fuel += (int32_t)below(key.draw(16u + q), span);
        uint32_t ty = WF_GRASS, heat = 0;
        if (q == qc) { ty = WF_FIRE; heat = WF_HEAT_FORCED; }         // set_fire_to :233-246
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
    // containment test of the episode needs no flood.
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
    s.fl = WF_F_ALIVE | WF_F_VISIBLE | WF_F_DIGGING | WF_F_RUNNING | WF_F_REACH_VALID | (onborder ? WF_F_BORDER : 0u);
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
        lds[oj[j] + 1u] = (uint8_t)((2u >> t) & 1u);              // type == fire
        lds[oj[j] + 2u] = (uint8_t)((7u >> t) & 1u);              // not dirt / water
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

// ----------------------------------------------------------------------------
// the step kernel: n_steps x ForestFire.step() (forest_fire.py:30-49) per env
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bperm(uint32_t byte_addr, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v);
}
__device__ __forceinline__ uint32_t pack_types(const uint32_t (&t)[4])
{
    return t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
}
// type of cell q (q identical in all lanes but held in a VGPR): owner lane q>>2 (byte
// address of a lane = 4*lane = q & ~3), component q&3
__device__ __forceinline__ uint32_t type_of(uint32_t tp, uint32_t q)
{
    return __builtin_amdgcn_ubfe(bperm(q & ~3u, tp), (q & 3u) << 3, 8u);
}

// Containment shortcut: removing a cell q from the passable set can only change which
// cells are connected to the border when q is itself a border cell, or when its
// passable 4-neighbours are not all connected to each other without q.  The fused
// kernel tests that on the 8-cell ring (passable ring cells in cyclic order
// N,NE,E,SE,S,SW,W,NW form one run; lanes 0..7 each inspect one ring cell); the
// single-step kernel counts the connected components exactly and refines the border case
// (see wf_step_group).

#ifndef WF_LB_ROWS
#define WF_LB_ROWS 8   // rows kernel: 56 VGPRs, 8 waves per SIMD (fewer resident waves measured slower)
#endif
#ifndef WF_SKIP_CLEAN
#define WF_SKIP_CLEAN 1   // single-step kernels store only the row pieces / cell quads that changed (-7.5 % time at config 3)
#endif
#ifndef WF_STORE_WAIT
#define WF_STORE_WAIT 1   // 0: private-phase kernels skip the wait for phase A byte stores (no measurable gain: 22.4-22.7 us either way)
#endif
#ifndef WF_LB
#define WF_LB 6   // minimum waves per SIMD the register allocator must leave room for
#endif
template <int W, int H, bool FUSED>
__global__ __launch_bounds__(256, WF_LB) void wf_step_wave(const WfArgs a)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    const uint32_t lane = __lane_id();
    const uint32_t wave0 = rfl(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * 4u;
    const uint32_t batch = (uint32_t)a.batch;
    const bool active = lane < (uint32_t)LANES;
    const uint32_t ign = (uint32_t)a.ignite_count;
    const char *planes_b = reinterpret_cast<const char *>(a.planes);
    // ring geometry for the containment shortcut: lanes 0..7 <-> N,NE,E,SE,S,SW,W,NW
    const uint32_t rk = lane & 7u;
    const uint32_t rdx = ((0x01A9u >> (2u * rk)) & 3u) - 1u;
    const uint32_t rdy = ((0x1A90u >> (2u * rk)) & 3u) - 1u;
    // per-lane geometry constants (one VGPR each, computed once per wave): for the lane's
    // cell j, nx0[j] = 1 unless x == 0, nxl[j] = 1 unless x == W-1, bmask bit j = border cell.
    // packed in ONE register `geo`: bit j = nx0[j], bit 4+j = nxl[j], bit 8+j = border cell j.
    uint32_t geo = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t q = 4u * lane + (uint32_t)j, x = q % W, y = q / W;
        geo |= (x != 0u ? 1u : 0u) << j;
        geo |= (x != (uint32_t)(W - 1) ? 1u : 0u) << (4 + j);
        geo |= ((x == 0u || y == 0u || x == (uint32_t)(W - 1) || y == (uint32_t)(H - 1)) ? 1u : 0u) << (8 + j);
    }
    // opaque: stop the optimiser from turning this back into 64-bit scalar masks
    asm volatile("" : "+v"(geo));
    const uint32_t lane_off = lane * 16u;
    __shared__ __attribute__((aligned(16))) uint8_t sObs[4][ObsGeo<W, H>::PAD];
    uint8_t *obs_lds = sObs[threadIdx.x >> 6];
    uint32_t oj[4];
    obs_offsets<W, H>(lane, oj);

    uint4 raw_next = make_uint4(0, 0, 0, 0), ev_next = make_uint4(0, 0, 0, 0);
    uint32_t act_pref = 0xffu;
    if (wave0 < batch) {
        if (active) raw_next = *reinterpret_cast<const uint4 *>(planes_b + (wave0 * (uint32_t)(CELLS * 4) + lane_off));
        ev_next = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.envs) + wave0 * 16u);
        if (a.actions) act_pref = (uint32_t)(uint8_t)a.actions[wave0];
    }

    for (uint32_t e = wave0; e < batch; e += nwaves) {
        const uint4 raw = raw_next;
        // env record: same 16 bytes in every lane, kept in VGPRs on purpose
        const uint4 ev = ev_next;
        uint32_t act = act_pref;
        const uint32_t poff = e * (uint32_t)(CELLS * 4) + lane_off;       // byte offset of this lane's 4 cells
        {   // software prefetch of this wave's next environment (grid, record, action)
            const uint32_t en = e + nwaves;
            if (en < batch) {
                if (active) raw_next = *reinterpret_cast<const uint4 *>(planes_b + (en * (uint32_t)(CELLS * 4) + lane_off));
                ev_next = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.envs) + en * 16u);
                if (a.actions) act_pref = (uint32_t)(uint8_t)a.actions[en];
            }
        }
        uint32_t ax = ev.x & 0xffffu, ay = ev.x >> 16, fl = ev.y & 0xffffu, episode = ev.z, tick = ev.w;
        // The compiler knows a load from a wave-uniform address is wave-uniform and would move
        // these to scalar registers (and all the flag logic below onto the scalar unit with
        // them).  Opaque VGPR constraints keep them - deliberately - on the vector side.
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(fl), "+v"(act));
        uint32_t t[4], u[4], h[4];           // type, fuel (unsigned byte), heat quanta
        {
            const uint32_t r[4] = { raw.x, raw.y, raw.z, raw.w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[j] = active ? (r[j] & 0xffu) : (uint32_t)WF_DIRT;   // idle lanes: inert, impassable
                u[j] = (r[j] >> 8) & 0xffu;
                h[j] = (r[j] >> 16) & 0xffu;
            }
        }
        bool dirty = false;
        double reward = 0.0;
        uint32_t done = 1;
        // fused launches keep the env's reachability bitboards in scalar registers across steps:
        // a bitboard written earlier in this launch must not be re-read through the scalar cache
        u64 RBK[4] = { 0, 0, 0, 0 };
        bool rb_have = false;

        const int n_steps = FUSED ? a.n_steps : 1;
        for (int st = 0; st < n_steps; ++st) {
            const uint32_t oi = (uint32_t)st * batch + e;
            uint32_t act_next = 0xffu;
            if (FUSED) {
                if (st + 1 < n_steps && a.actions) act_next = (uint32_t)(uint8_t)a.actions[oi + batch];
            }
            reward = 0.0;
            done = 1;
            if (rfl(fl & WF_F_RUNNING) == 0u) {
                // ---- rare: finished environment
                if (a.hot_flags & WF_HOT_AUTO_RESET) {
                    // gymnasium-style next-step autoreset: this step returns the fresh episode.
                    // Inputs are pinned inside the branch so none of this is speculated above it.
                    EnvS s = { rfl(ax), rfl(ay), rfl(fl), rfl(episode), rfl(tick) };
                    const WfCold *cold = a.cold;
                    uint32_t e_pin = e;
                    asm volatile("" : "+s"(cold), "+s"(e_pin));
                    cold_ptr cc = (cold_ptr)(uintptr_t)cold;
                    reset_regs<W, H>(cold, cc->p.seed, e_pin, lane, t, u, h, s, a.reach + (size_t)e_pin * 4);
                    ax = s.ax; ay = s.ay; fl = s.fl; episode = s.episode; tick = s.tick;
                    {   // the fresh bitboards were written with vector stores: keep them in registers,
                        // never re-read them through the scalar cache inside this launch
                        const uint32_t qs = s.ay * W + s.ax;
                        static_for<4>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            RBK[j] = CM<W, H, j>::VALID & ~(((qs & 3u) == (uint32_t)j) ? (1ull << (qs >> 2)) : 0ull);
                        });
                        rb_have = true;
                    }
                    dirty = true;
                    done = 0;
                }
            } else {
                dirty = true;
                // ---- Agent.move (environment.py:141-155).  Integer 0/1 flags on the vector
                // unit only: combining per-lane booleans would cost 64-bit scalar mask ops.
                const uint32_t ent = MOVE_LUT >> (min(act, 4u) * 6u);                  // act is an unsigned byte
                const uint32_t ism = (ent >> 4) & 1u;                                  // action in N,S,E,W
                const uint32_t nx = ax + (ent & 3u) - 1u;                              // :163-171
                const uint32_t ny = ay + ((ent >> 2) & 3u) - 1u;
                const uint32_t out = (((uint32_t)(W - 1) - nx) | nx | ((uint32_t)(H - 1) - ny) | ny) >> 31;
                const uint32_t inb = ism & (out ^ 1u);                                 // :146 inbounds
                const uint32_t qa = ay * W + ax;
                uint32_t qd = qa + inb * (ny * W + nx - qa);
                const uint32_t tp = pack_types(t);
                const uint32_t td = type_of(tp, qd);
                const uint32_t tcur = type_of(tp, qa);
                const uint32_t moved = inb & min(td ^ (uint32_t)WF_WATER, 1u);         // :146 traversable
                const uint32_t dfire = moved & (2u >> td);                             // destination burning
                uint32_t dug = moved & (fl >> 2) & (5u >> td) & 1u;                    // digging, grass|burnt (:151-152, :125-133)
                fl = (fl & ~(ism << 1)) | (moved << 1) | (dfire << 3);                 // :142, :148-149, :154-155
                ax += moved * (nx - ax);
                ay += moved * (ny - ay);
                uint32_t ta = tcur + moved * (td - tcur);                              // type under the agent
                // ---- Agent.toggle_digging (forest_fire.py:35-36, environment.py:136-138): rare
                if (a.hot_flags & WF_HOT_DIG_TOGGLE) {
                    if (rfl(act == 4u ? 1u : 0u)) {
                        fl ^= WF_F_DIGGING;
                        // divergence (documented): the reference also retypes a BURNING cell to dirt
                        // here while keeping it in burning_cells; this build leaves it burning.
                        dug = (fl >> 2) & (5u >> tcur) & 1u;
                        qd = qa;
                    }
                }
                ta += dug * ((uint32_t)WF_DIRT - ta);
                {
                    // per-lane selector: component to retype on the owning lane, >= 4 elsewhere
                    const uint32_t sel = (qd & 3u) + ((min((qd >> 2) ^ lane, 1u) | (dug ^ 1u)) << 2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = (sel == (uint32_t)j) ? (uint32_t)WF_DIRT : t[j];
                }

                // ---- ForestFire.update (forest_fire.py:85-106)
                const bool do_update = FUSED ? (bool)((a.update_mask >> st) & 1ull) : (bool)(a.update_mask & 1ull);
                if (do_update) {
                    // :87 drop dead agents (Agent.is_dead environment.py:116-120)
                    const uint32_t dies = fl & ((fl >> 3) | (2u >> ta)) & 1u;          // alive & (dead | on fire)
                    fl &= ~(dies * (uint32_t)(WF_F_ALIVE | WF_F_VISIBLE));

                    u64 STILL[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // reduce_fuel (:297-307): a fire cell whose fuel-1 > 0 keeps burning
                        const uint32_t key = (t[j] << 8) + u[j];
                        STILL[j] = __ballot((key - 0x102u) < 0x7Eu);                   // t==1 && 2<=fuel<=127
                        const uint32_t isfire = (2u >> t[j]) & 1u;
                        u[j] = (u[j] - isfire) & 0xffu;
                        t[j] += isfire;                                                // fire -> burnt ...
                    }
                    uint32_t ign_bits = 0;
                    static_for<4>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        using C = CM<W, H, j>;
                        t[j] -= lanebit(STILL[j]) ? 1u : 0u;                           // ... unless still burning
                        // get_neighbours radius 1 (:311-326) + apply_heat_from_to (:278-294): count the
                        // still-burning 4-neighbours.  Vertical: pure mask shifts.  Horizontal: the
                        // row-wrap guard is the per-lane constant, not a 64-bit scalar mask.
                        const u64 LF = j > 0 ? STILL[(j + 3) & 3] : (STILL[3] << 1);
                        const u64 RT = j < 3 ? STILL[(j + 1) & 3] : (STILL[0] >> 1);
                        const uint32_t cl = C::NOT_X0 == C::VALID ? (lanebit(LF) ? 1u : 0u) : (lanebit(LF) ? ((geo >> j) & 1u) : 0u);
                        const uint32_t cr = C::NOT_XL == C::VALID ? (lanebit(RT) ? 1u : 0u) : (lanebit(RT) ? ((geo >> (4 + j)) & 1u) : 0u);
                        const uint32_t hn = h[j] + (lanebit(nb_up<W, H, j>(STILL)) ? 1u : 0u) +
                                                   (lanebit(nb_dn<W, H, j>(STILL)) ? 1u : 0u) + cl + cr;
                        const uint32_t key2 = (t[j] << 8) + hn;
                        const uint32_t ig = ((key2 - ign) < (256u - ign)) ? 1u : 0u;   // grass && hn >= ign
                        h[j] = (t[j] == WF_GRASS) ? min(hn, ign) : h[j];
                        t[j] += ig;                                                    // grass -> fire
                        ign_bits |= ig << j;
                    });
                    // set_fire_to :245-246  a newly ignited border cell
                    fl |= __ballot((ign_bits & (geo >> 8)) != 0u) ? (uint32_t)WF_F_BORDER : 0u;
                    tick += 1;
                }
                u64 FIRE[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) FIRE[j] = __ballot(t[j] == WF_FIRE);
                const uint32_t anyfire = (FIRE[0] | FIRE[1] | FIRE[2] | FIRE[3]) ? 1u : 0u;   // wave-uniform
                if (do_update)
                    fl &= ~(((fl & anyfire) ^ 1u) << 4);                               // :105-106 (!alive || !fire) -> !RUNNING

                // ---- World.get_reward (environment.py:342-390)
                uint32_t contained = 0;
                // :345-346  containment test runs while the fire has not touched the border
                const uint32_t check = (uint32_t)(a.hot_flags & WF_HOT_CONTAINMENT) & anyfire &
                                       (((fl >> 5) | (fl >> 6)) ^ 1u) & 1u;
                if (rfl(check)) {
                    // border-reachability bitboards of this env, through the scalar cache
                    typedef u64 u64x4 __attribute__((ext_vector_type(4)));
                    typedef const u64x4 __attribute__((address_space(4))) *const_u64x4_ptr;
                    u64 RB[4];
                    if (FUSED && rb_have) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) RB[j] = RBK[j];
                    } else {
                        const u64x4 rbv = *(const_u64x4_ptr)(uintptr_t)(a.reach + (size_t)e * 4);
                        RB[0] = rbv.x; RB[1] = rbv.y; RB[2] = rbv.z; RB[3] = rbv.w;
                    }
                    uint32_t recompute = ((fl >> 7) ^ 1u) & 1u;                        // bitboard not valid yet
                    {
                        const uint32_t xd = dug ? ax : 1u, yd = dug ? ay : 1u;         // dug cell == agent cell
                        const uint32_t onb = (((xd - 1u) | (yd - 1u) | ((uint32_t)(W - 2) - xd) | ((uint32_t)(H - 2) - yd)) >> 31);
                        const uint32_t rx = xd + rdx, ry = yd + rdy;
                        const uint32_t rout = (((uint32_t)(W - 1) - rx) | rx | ((uint32_t)(H - 1) - ry) | ry) >> 31;
                        const uint32_t nq = (rout ^ 1u) * (ry * W + rx);
                        const uint32_t ty = type_of(pack_types(t), nq);
                        const uint32_t ring = (uint32_t)__ballot((ty + 4u * rout + 4u * (lane >> 3)) < (uint32_t)WF_DIRT) & 0xffu;
                        const uint32_t rot = ((ring << 1) | (ring >> 7)) & 0xffu;
                        const uint32_t runs = (uint32_t)__popc(ring & ~rot);
                        recompute |= dug & (onb | (runs >> 1) | (runs >> 2));
                    }
                    if (rfl(recompute)) {
                        u64 PASS[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) PASS[j] = ~__ballot(t[j] >= WF_DIRT);
                        static_for<4>([&](auto J) { constexpr int j = decltype(J)::value; RB[j] = CM<W, H, j>::BORDER & PASS[j]; });
                        flood<W, H>(RB, PASS);
                        fl = (fl | WF_F_REACH_VALID) & ~(uint32_t)WF_F_REACH_ROWS;      // the rows-layout board is stale now
                        if (lane == 0) {
                            ulonglong2 *rp = reinterpret_cast<ulonglong2 *>(a.reach + (size_t)e * 4);
                            rp[0] = make_ulonglong2(RB[0], RB[1]);
                            rp[1] = make_ulonglong2(RB[2], RB[3]);
                        }
                    }
                    if (FUSED) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) RBK[j] = RB[j];
                        rb_have = true;
                    }
                    // stale RB bits on cells dug since the last flood are harmless: fire never sits there
                    contained = ((FIRE[0] & RB[0]) | (FIRE[1] & RB[1]) | (FIRE[2] & RB[2]) | (FIRE[3] & RB[3])) ? 0u : 1u;
                    fl |= contained << 6;                                              // :361-367
                } else {
                    fl &= ~(dug << 7);                                                 // bitboard no longer trustworthy
                }
                reward = a.default_reward;                                             // :390
                if (rfl(anyfire ^ 1u)) {                                               // :384-387 burnt out (rare)
                    int healthy = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) healthy += __popcll(__ballot(t[j] == WF_GRASS));
                    reward = a.contained_bonus * ((double)healthy / (double)CELLS);
                }
                reward = (fl & WF_F_ALIVE) ? reward : a.death_penalty;                 // :380-381
                reward = contained ? a.contained_bonus : reward;
                done = ((fl >> 4) & 1u) ^ 1u;                                          // forest_fire.py:48
            }
            if (FUSED) {
                if (lane == 0) {
                    if (a.reward) a.reward[oi] = reward;
                    if (a.done) a.done[oi] = (uint8_t)done;
                }
                act = act_next;
                asm volatile("" : "+v"(act));
            }
        }

        if (dirty) {
            const uint32_t qa = ay * W + ax;
            const uint32_t sel = (qa & 3u) + ((min((qa >> 2) ^ lane, 1u) | (((fl >> 1) & 1u) ^ 1u)) << 2);
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[j] = t[j] | (u[j] << 8) | (h[j] << 16) | (sel == (uint32_t)j ? (1u << 24) : 0u);
            if (active)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.planes) + poff) = make_uint4(r[0], r[1], r[2], r[3]);
            if (a.obs && active) emit_obs<W, H>(obs_lds, a.obs + (size_t)e * (CELLS * 3), r, oj, lane);
        } else if (a.obs && active) {      // frozen env: observation of the unchanged grid
            const uint32_t cf[4] = { raw.x, raw.y, raw.z, raw.w };
            emit_obs<W, H>(obs_lds, a.obs + (size_t)e * (CELLS * 3), cf, oj, lane);
        }
        if (lane == 0) {
            if (dirty)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.envs) + e * 16u) = make_uint4(ax | (ay << 16), fl, episode, tick);
            if (!FUSED) {
                if (a.reward) a.reward[e] = reward;
                if (a.done) a.done[e] = (uint8_t)done;
            }
        }
    }
}


// ----------------------------------------------------------------------------
// single-step kernel (n_steps == 1): GROUP layout
// ----------------------------------------------------------------------------
// A wave owns a contiguous group of G environments and runs three phases:
//   A  lanes = environments.  Lane k does ALL per-environment logic of env k at once
//      for the group: Agent.move / toggle / death (environment.py:116-171), the
//      containment ring test, and applies the two or three touched bytes (dug cell,
//      agent_pos plane) straight to the grid in memory.  One vector instruction
//      serves up to 64 environments instead of one.
//   B  one environment at a time, lanes = cells (4 per lane, packed dwords as they lie
//      in memory, no unpack/pack): the CA tick proper with ballot masks.
//   C  lanes = environments again: flags, reward, done, records.
// Per environment this costs ~75 vector + ~85 scalar instructions against ~310 + ~140
// for the one-env-per-wave formulation (which remains the fused multi-step path).
enum {
    PB_RUNNING = 1, PB_PRECHECK = 2, PB_VALID = 4, PB_RING = 8,       // phase A -> B (per env)
    RB_ANYFIRE = 1, RB_BORDER = 2, RB_CONTAINED = 4, RB_RECOMPUTED = 8, RB_WASRESET = 16, RB_HEALTHY_SHIFT = 8
};

// lane-parallel per-environment registers (lanes = environments)
struct EnvRegs {
    uint32_t ax, ay, fl, episode, tick, Pv;
};

// Phase A for the environment of this lane: Agent.move / toggle / death, the containment ring
// verdict, and the touched bytes applied to the grid in memory.  Returns the env's registers;
// Pv = PB_* bits for phase B (bit 0 = the env was running when the step began).
template <int W, int H, int VB>
__device__ __forceinline__ EnvRegs group_phase_a(const WfArgs &a, const bool lane_env, const uint32_t ek, const bool do_update)
{
    constexpr int CELLS = W * H;
    constexpr uint32_t ENV_BYTES = CELLS * 4;
    char *planes_b = reinterpret_cast<char *>(a.planes);
    const uint32_t pbase = ek * ENV_BYTES;
    const uint4 ev = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.envs) + ek * 16u);
    const uint32_t act = a.actions ? (uint32_t)(uint8_t)a.actions[ek] : 0xffu;
    uint32_t ax = ev.x & 0xffffu, ay = ev.x >> 16, fl = ev.y & 0xffffu, episode = ev.z, tick = ev.w;
    const uint32_t running = (fl >> 4) & 1u & (lane_env ? 1u : 0u);

    // ---- Agent.move (environment.py:141-155), integer 0/1 flags
    const uint32_t ent = MOVE_LUT >> (min(act, 4u) * 6u);
    const uint32_t ism = (ent >> 4) & 1u;
    const uint32_t nx = ax + (ent & 3u) - 1u;                                          // :163-171
    const uint32_t ny = ay + ((ent >> 2) & 3u) - 1u;
    const uint32_t out = (((uint32_t)(W - 1) - nx) | nx | ((uint32_t)(H - 1) - ny) | ny) >> 31;
    const uint32_t inb = ism & (out ^ 1u);                                             // :146 inbounds
    const uint32_t qa = ay * W + ax;
    uint32_t qd = qa + inb * (ny * W + nx - qa);
    // gathers: type byte of the destination / current cell, and of the 8-ring around the
    // cell that would be dug (ring test, see "Containment shortcut" above)
    const uint32_t td = *reinterpret_cast<const uint8_t *>(planes_b + (pbase + qd * 4u));
    const uint32_t tcur = *reinterpret_cast<const uint8_t *>(planes_b + (pbase + qa * 4u));
    const bool toggling = (a.hot_flags & WF_HOT_DIG_TOGGLE) != 0;
    const uint32_t istog = toggling ? (act == 4u ? 1u : 0u) : 0u;
    // cell that may be dug: destination of a move, or the own cell when digging is toggled on
    const uint32_t xd = istog ? ax : (inb ? nx : ax), yd = istog ? ay : (inb ? ny : ay);
    // Containment shortcut: gather the passability of the 8-ring (N,NE,E,SE,S,SW,W,NW) around
    // the cell that may get dug; off-grid = impassable.
    uint32_t ring = 0, bnmask = 0;
    const uint32_t precheck0 = ((a.hot_flags & WF_HOT_CONTAINMENT) ? 1u : 0u) & (((fl >> 5) | (fl >> 6)) ^ 1u) & 1u;
    if (__ballot(precheck0 & running) != 0) {        // wave-uniform: nobody needs the test otherwise
        const uint32_t qx = istog ? qa : qd;
        // edge flags of the centre cell: which ring cells fall off the grid
        const uint32_t x0 = xd == 0u ? 1u : 0u, xl = xd == (uint32_t)(W - 1) ? 1u : 0u;
        const uint32_t y0 = yd == 0u ? 1u : 0u, yl = yd == (uint32_t)(H - 1) ? 1u : 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            constexpr int DX[8] = { 0, 1, 1, 1, 0, -1, -1, -1 }, DY[8] = { -1, -1, 0, 1, 1, 1, 0, -1 };
            const uint32_t off = (DX[r] < 0 ? x0 : DX[r] > 0 ? xl : 0u) | (DY[r] < 0 ? y0 : DY[r] > 0 ? yl : 0u);
            const uint32_t nq = off ? 0u : qx + (uint32_t)(DY[r] * W + DX[r]);
            const uint32_t ty = *reinterpret_cast<const uint8_t *>(planes_b + (pbase + nq * 4u));
            ring |= ((off ^ 1u) & (ty < (uint32_t)WF_DIRT ? 1u : 0u)) << r;
        }
        // which 4-neighbours (ring bits 0,2,4,6 = N,E,S,W) are border cells themselves
        const uint32_t ex = x0 | xl, ey = y0 | yl;
        bnmask = (ex | (yd == 1u ? 1u : 0u)) | ((ey | (xd == (uint32_t)(W - 2) ? 1u : 0u)) << 2) |
                 ((ex | (yd == (uint32_t)(H - 2) ? 1u : 0u)) << 4) | ((ey | (xd == 1u ? 1u : 0u)) << 6);
    }
    const uint32_t moved = inb & min(td ^ (uint32_t)WF_WATER, 1u) & running;           // :146 traversable
    const uint32_t dfire = moved & (2u >> td);                                         // destination burning
    uint32_t dug = moved & (fl >> 2) & (5u >> td) & 1u;                                // :151-152, :125-133
    const uint32_t vis_before = (fl >> 1) & 1u & running;
    fl = (fl & ~((ism & running) << 1)) | (moved << 1) | (dfire << 3);                 // :142, :148-149, :154-155
    ax += moved * (nx - ax);
    ay += moved * (ny - ay);
    uint32_t ta = tcur + moved * (td - tcur);                                          // type under the agent
    {   // Agent.toggle_digging (forest_fire.py:35-36, environment.py:136-138)
        const uint32_t tg = istog & running;
        fl ^= tg << 2;
        // divergence (documented): the reference also retypes a BURNING cell to dirt here
        // while keeping it in burning_cells; this build leaves it burning.
        const uint32_t dugt = (fl >> 2) & (5u >> tcur) & 1u;
        dug = tg ? dugt : dug;
        qd = tg ? qa : qd;
    }
    ta += dug * ((uint32_t)WF_DIRT - ta);
    if (do_update) {   // forest_fire.py:87 drop dead agents (Agent.is_dead environment.py:116-120)
        const uint32_t dies = fl & ((fl >> 3) | (2u >> ta)) & 1u & running;
        fl &= ~(dies * (uint32_t)(WF_F_ALIVE | WF_F_VISIBLE));
    }
    const uint32_t vis_after = (fl >> 1) & 1u;
    const uint32_t qa_new = ay * W + ax;
    // apply the touched bytes to the grid in memory (phase B reloads the whole grid)
    if (vis_before) *reinterpret_cast<uint8_t *>(planes_b + (pbase + qa * 4u + 3u)) = 0;
    if (dug) *reinterpret_cast<uint8_t *>(planes_b + (pbase + qd * 4u)) = (uint8_t)WF_DIRT;
    if (vis_after & running) *reinterpret_cast<uint8_t *>(planes_b + (pbase + qa_new * 4u + 3u)) = 1;
    // Verdict: can this dig change which cells reach the border?  Digging cell X re-routes
    // nothing iff the passable 4-neighbours of X stay connected to each other around the ring
    // (a passable diagonal connects its two even neighbours): components = #passable
    // 4-neighbours - #connecting diagonals (the closed 8-ring gives 0).  A border cell X may in
    // addition have been the only border cell its neighbours reached: harmless iff one of those
    // neighbours is a border cell itself.  Exact; everything else triggers the flood in phase B.
    uint32_t ringflag;
    {
        const uint32_t onb = (((xd - 1u) | (yd - 1u) | ((uint32_t)(W - 2) - xd) | ((uint32_t)(H - 2) - yd)) >> 31);
        const uint32_t even = ring & 0x55u;
        const uint32_t eff = ring & 0xAAu & (even << 1) & ((even >> 1) | (even << 7));
        const uint32_t comps = (uint32_t)__popc(even) - (uint32_t)__popc(eff);
        const uint32_t local_fail = (comps + 0x7FFFFFFEu) >> 31;              // comps >= 2 (comps in 0..4)
        const uint32_t border_fail = onb & min(even, 1u) & (min(ring & bnmask, 1u) ^ 1u);
        ringflag = dug & (local_fail | border_fail) & 1u;
    }
    const uint32_t precheck = ((a.hot_flags & WF_HOT_CONTAINMENT) ? 1u : 0u) & (((fl >> 5) | (fl >> 6)) ^ 1u) & 1u;
    const uint32_t Pv = running | (precheck << 1) | (((fl >> VB) & 1u) << 2) | (ringflag << 3);   // VB: this layout's board-valid flag
    EnvRegs S = { ax, ay, fl, episode, tick, Pv };
    return S;
}

// Phase C for the environment of this lane: flags, reward, done, record (Rv = RB_* bits of phase B).
template <int W, int H, int VB>
__device__ __forceinline__ void group_phase_c(const WfArgs &a, const EnvRegs &S, const uint32_t Rv, const bool lane_env,
                                              const uint32_t ek, const bool do_update)
{
    constexpr int CELLS = W * H;
    uint32_t ax = S.ax, ay = S.ay, fl = S.fl, episode = S.episode, tick = S.tick;
    const uint32_t running = S.Pv & 1u;
    {
        const uint32_t anyfire = Rv & 1u, bhit = (Rv >> 1) & 1u, contained = (Rv >> 2) & 1u;
        const uint32_t recomputed = (Rv >> 3) & 1u, wasreset = (Rv >> 4) & 1u, healthy = Rv >> RB_HEALTHY_SHIFT;
        const uint32_t upd = do_update ? running : 0u;
        // a recomputed board is valid in THIS kernel's layout only (the event that forced it staled the other)
        fl = (fl & ~(recomputed * (uint32_t)(WF_F_REACH_VALID | WF_F_REACH_ROWS))) | (bhit << 5) | (contained << 6) | (recomputed << VB);
        fl &= ~((upd & ((fl & anyfire) ^ 1u)) << 4);                           // forest_fire.py:105-106
        tick += upd;
        double reward = a.default_reward;                                      // environment.py:390
        if (__ballot(running && !anyfire))                                     // :384-387 (rare, wave-uniform branch)
            reward = anyfire ? reward : a.contained_bonus * ((double)healthy / (double)CELLS);
        reward = (fl & WF_F_ALIVE) ? reward : a.death_penalty;                 // :380-381
        reward = contained ? a.contained_bonus : reward;                       // :361-367
        reward = running ? reward : 0.0;                                       // finished (frozen or freshly reset)
        const uint32_t done = wasreset ? 0u : (((fl >> 4) & 1u) ^ 1u);         // forest_fire.py:48
        if (lane_env) {
            if (running | wasreset)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.envs) + ek * 16u) = make_uint4(ax | (ay << 16), fl, episode, tick);
            if (a.reward) a.reward[ek] = reward;
            if (a.done) a.done[ek] = (uint8_t)done;
        }
    }
}

// SHARED = false: every wave runs phases A, B, C for its own group of G environments.
// SHARED = true : a workgroup owns 64 environments.  Wave 0 runs phase A for all 64 at once (all
//   lanes busy instead of G of them), the registers travel through LDS, the 64/G waves each run
//   phase B for G environments, and whichever wave finishes last runs phase C for all 64.  This
//   divides the per-environment instruction cost of phases A and C by 64/G.
template <int W, int H, bool SHARED>
__global__ __launch_bounds__(SHARED ? 512 : 256, SHARED ? 8 : WF_LB) void wf_step_group(const WfArgs a)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    constexpr uint32_t ENV_BYTES = CELLS * 4;
    const uint32_t lane = __lane_id();
    const uint32_t batch = (uint32_t)a.batch;
    const uint32_t G = (uint32_t)a.group;
    const uint32_t wv = rfl(threadIdx.x >> 6);
    uint32_t e0, ng, A0, na;                                // B: group [e0, e0+ng);  A/C: lanes cover [A0, A0+na)
    if constexpr (SHARED) {
        A0 = blockIdx.x * 64u;
        na = min(64u, batch - A0);
        e0 = A0 + wv * G;
        ng = wv * G < na ? min(G, na - wv * G) : 0u;
    } else {
        e0 = rfl(blockIdx.x * 4u + wv) * G;
        if (e0 >= batch) return;
        ng = min(G, batch - e0);                            // environments in this group (wave-uniform)
        A0 = e0;
        na = ng;
    }
    const bool active = lane < (uint32_t)LANES;
    const uint32_t ign = (uint32_t)a.ignite_count;
    const bool do_update = (a.update_mask & 1ull) != 0;
    char *planes_b = reinterpret_cast<char *>(a.planes);
    const uint32_t lane_off = lane * 16u;
    __shared__ uint32_t sS[SHARED ? 7 : 1][64];            // ax|ay<<16, fl, episode, tick, Pv, Rv, counter
    enum { S_POS = 0, S_FL = 1, S_EP = 2, S_TICK = 3, S_P = 4, S_R = 5, S_CNT = 6 };

    // ================= phase A: lanes = environments =================
    const bool lane_env = lane < na;
    const uint32_t ek = A0 + (lane_env ? lane : 0u);        // idle lanes shadow the first env (loads only)
    EnvRegs S = { 0, 0, 0, 0, 0, 0 };
    uint32_t Pv;
    if constexpr (SHARED) {
        if (wv == 0) {
            S = group_phase_a<W, H, 7>(a, lane_env, ek, do_update);
            sS[S_POS][lane] = S.ax | (S.ay << 16);
            sS[S_FL][lane] = S.fl;
            sS[S_EP][lane] = S.episode;
            sS[S_TICK][lane] = S.tick;
            sS[S_P][lane] = S.Pv;
            if (lane == 0) sS[S_CNT][0] = 0;
            __builtin_amdgcn_s_waitcnt(0);                   // the byte stores have landed
        }
        __syncthreads();                                     // workgroup-scope release/acquire: same CU, same L1
        Pv = sS[S_P][wv * G + (lane < G ? lane : 0u)];       // lane k <-> k-th environment of this wave's group
    } else {
        S = group_phase_a<W, H, 7>(a, lane_env, ek, do_update);
        Pv = S.Pv;
        // the byte stores above must have landed before phase B reads those grids back
        __builtin_amdgcn_s_waitcnt(0);                       // vmcnt(0) lgkmcnt(0) expcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    uint32_t Rv = 0;                                         // phase B results, one word per env-lane

    // ================= phase B: one environment at a time, lanes = cells =================
    // The whole loop runs with EXEC = the lanes that own cells: idle lanes never load,
    // store, or contribute ballot bits, so no validity masks are needed anywhere.
    // geo bit j of a lane = its cell j is a border cell (per-lane constant instead of
    // 64-bit scalar constants).
    uint32_t bgeo = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t q = 4u * lane + (uint32_t)j, x = q % W, y = q / W;
        const uint32_t inner = ((x - 1u) | (y - 1u) | ((uint32_t)(W - 2) - x) | ((uint32_t)(H - 2) - y)) >> 31;
        bgeo |= inner << j;
    }
    asm volatile("" : "+v"(bgeo));
    const uint32_t K_FIRE_STEP = 0xFFFFFF01u;                // fuel byte -1, type byte +1 (fire -> burnt)
    __shared__ __attribute__((aligned(16))) uint8_t sObs[SHARED ? 8 : 4][ObsGeo<W, H>::PAD];
    uint8_t *obs_lds = sObs[wv];
    uint32_t oj[4];
    obs_offsets<W, H>(lane, oj);
    if (active) {
        uint4 raw_next = *reinterpret_cast<const uint4 *>(planes_b + (e0 * ENV_BYTES + lane_off));
        for (uint32_t k = 0; k < ng; ++k) {
            const uint32_t e = e0 + k;
            const uint4 raw = raw_next;
            {   // prefetch the next environment of the group (clamped: the last one re-reads itself)
                const uint32_t en = e0 + min(k + 1u, ng - 1u);
                raw_next = *reinterpret_cast<const uint4 *>(planes_b + (en * ENV_BYTES + lane_off));
            }
            const uint32_t P = __builtin_amdgcn_readlane(Pv, k);
            uint32_t R = 0;
            if (!(P & PB_RUNNING)) {
                // ---- rare: finished environment
                if (a.hot_flags & WF_HOT_AUTO_RESET) {
                    uint32_t ep_k;
                    if constexpr (SHARED) ep_k = rfl(sS[S_EP][wv * G + k]);
                    else ep_k = (uint32_t)__builtin_amdgcn_readlane(S.episode, k);
                    EnvS s = { 0, 0, 0, ep_k, 0 };
                    const WfCold *cold = a.cold;
                    uint32_t e_pin = e;
                    asm volatile("" : "+s"(cold), "+s"(e_pin));
                    cold_ptr cc = (cold_ptr)(uintptr_t)cold;
                    uint32_t t[4], u[4], h[4];
                    // the stream is keyed by the env's index in the context, not in this (possibly split) launch
                    reset_regs<W, H>(cold, cc->p.seed, e_pin + a.env_base, lane, t, u, h, s, a.reach + (size_t)e_pin * 4);
                    const uint32_t qs = s.ay * W + s.ax;
                    uint32_t r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r[j] = t[j] | (u[j] << 8) | (h[j] << 16) | ((lane == (qs >> 2) && (qs & 3u) == (uint32_t)j) ? (1u << 24) : 0u);
                    *reinterpret_cast<uint4 *>(planes_b + (e * ENV_BYTES + lane_off)) = make_uint4(r[0], r[1], r[2], r[3]);
                    if (a.obs) emit_obs<W, H>(obs_lds, a.obs + (size_t)e * (CELLS * 3), r, oj, lane);
                    if constexpr (SHARED) {
                        if (lane == 0) {
                            const uint32_t i = wv * G + k;
                            sS[S_POS][i] = s.ax | (s.ay << 16);
                            sS[S_FL][i] = s.fl;
                            sS[S_EP][i] = s.episode;
                            sS[S_TICK][i] = s.tick;
                        }
                    } else {
                        S.ax = (lane == k ? s.ax : S.ax);
                        S.ay = (lane == k ? s.ay : S.ay);
                        S.fl = (lane == k ? s.fl : S.fl);
                        S.episode = (lane == k ? s.episode : S.episode);
                        S.tick = (lane == k ? s.tick : S.tick);
                    }
                    R = RB_WASRESET;
                } else if (a.obs) {      // frozen: the observation of the unchanged grid
                    const uint32_t cf[4] = { raw.x, raw.y, raw.z, raw.w };
                    emit_obs<W, H>(obs_lds, a.obs + (size_t)e * (CELLS * 3), cf, oj, lane);
                }
                Rv = (lane == k ? R : Rv);
                continue;
            }
            uint32_t c[4] = { raw.x, raw.y, raw.z, raw.w };     // packed cells: type | fuel<<8 | heat<<16 | agent<<24
            u64 FIRE[4];                                        // fire cells after the tick
            if (do_update) {
                // ---- ForestFire.update (forest_fire.py:85-106)
                u64 STILL[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // reduce_fuel (environment.py:297-307)
                    const u64 F = __ballot((c[j] & 0xffu) == (uint32_t)WF_FIRE);
                    STILL[j] = F & __ballot((int32_t)(int8_t)(c[j] >> 8) > 1);     // fuel - 1 > 0
                    // fuel -= 1 and fire -> burnt, unless still burning (type stays fire)
                    const uint32_t d = lanebit(STILL[j]) ? (K_FIRE_STEP - 1u) : K_FIRE_STEP;
                    c[j] += lanebit(F) ? d : 0u;
                }
                uint32_t ign_bits = 0;
                static_for<4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    using C = CM<W, H, j>;
                    // get_neighbours radius 1 (:311-326) + apply_heat_from_to (:278-294), branch-free so
                    // that every mask stays a wave-uniform scalar.  Row wrap: compile-time lane masks.
                    const u64 GR = __ballot((c[j] & 0xffu) == (uint32_t)WF_GRASS);
                    u64 LF = j > 0 ? STILL[(j + 3) & 3] : (STILL[3] << 1);
                    u64 RT = j < 3 ? STILL[(j + 1) & 3] : (STILL[0] >> 1);
                    if (C::NOT_X0 != C::VALID) LF &= C::NOT_X0;
                    if (C::NOT_XL != C::VALID) RT &= C::NOT_XL;
                    const uint32_t cnt = (lanebit(nb_up<W, H, j>(STILL)) ? 1u : 0u) + (lanebit(nb_dn<W, H, j>(STILL)) ? 1u : 0u) +
                                         (lanebit(LF) ? 1u : 0u) + (lanebit(RT) ? 1u : 0u);
                    const uint32_t room = ign - ((c[j] >> 16) & 0xffu);            // quanta still missing (>= 1 on grass)
                    const u64 IGN = GR & __ballot(cnt >= room);
                    const uint32_t ig = lanebit(IGN) ? 1u : 0u;
                    const uint32_t cn = c[j] + (min(cnt, room) << 16) + ig;        // heat saturates; grass -> fire
                    c[j] = lanebit(GR) ? cn : c[j];
                    ign_bits |= ig << j;
                    FIRE[j] = STILL[j] | IGN;
                });
                // set_fire_to :245-246: a newly ignited border cell
                R |= __ballot((ign_bits & bgeo) != 0u) ? RB_BORDER : 0u;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) FIRE[j] = __ballot((c[j] & 0xffu) == (uint32_t)WF_FIRE);
            }
            const bool anyfire = (FIRE[0] | FIRE[1] | FIRE[2] | FIRE[3]) != 0;
            R |= anyfire ? RB_ANYFIRE : 0u;
            // ---- containment part of World.get_reward (environment.py:345-377)
            if ((P & PB_PRECHECK) && anyfire && !(R & RB_BORDER)) {
                typedef u64 u64x4 __attribute__((ext_vector_type(4)));
                typedef const u64x4 __attribute__((address_space(4))) *const_u64x4_ptr;
                u64 RB[4];
                if ((P & PB_VALID) && !(P & PB_RING)) {
                    const u64x4 rbv = *(const_u64x4_ptr)(uintptr_t)(a.reach + (size_t)e * 4);
                    RB[0] = rbv.x; RB[1] = rbv.y; RB[2] = rbv.z; RB[3] = rbv.w;
                } else {
                    u64 PASS[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) PASS[j] = __ballot((c[j] & 0xffu) < (uint32_t)WF_DIRT);
                    static_for<4>([&](auto J) { constexpr int j = decltype(J)::value; RB[j] = CM<W, H, j>::BORDER & PASS[j]; });
                    flood<W, H>(RB, PASS);
                    if (lane == 0) {
                        ulonglong2 *rp = reinterpret_cast<ulonglong2 *>(a.reach + (size_t)e * 4);
                        rp[0] = make_ulonglong2(RB[0], RB[1]);
                        rp[1] = make_ulonglong2(RB[2], RB[3]);
                    }
                    R |= RB_RECOMPUTED;
                }
                // stale RB bits on cells dug since the last flood are harmless: fire never sits there
                if (!((FIRE[0] & RB[0]) | (FIRE[1] & RB[1]) | (FIRE[2] & RB[2]) | (FIRE[3] & RB[3]))) R |= RB_CONTAINED;
            }
            if (!anyfire) {                                                        // :384-387 burnt out (rare)
                uint32_t healthy = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) healthy += (uint32_t)__popcll(__ballot((c[j] & 0xffu) == (uint32_t)WF_GRASS));
                R |= healthy << RB_HEALTHY_SHIFT;
            }
#if WF_SKIP_CLEAN
            // most cells do not change in a tick: a lane whose four cells are unchanged writes nothing
            if ((c[0] ^ raw.x) | (c[1] ^ raw.y) | (c[2] ^ raw.z) | (c[3] ^ raw.w))
#endif
                *reinterpret_cast<uint4 *>(planes_b + (e * ENV_BYTES + lane_off)) = make_uint4(c[0], c[1], c[2], c[3]);
            if (a.obs) emit_obs<W, H>(obs_lds, a.obs + (size_t)e * (CELLS * 3), c, oj, lane);
            Rv = (lane == k ? R : Rv);
        }
    }

    // ================= phase C: lanes = environments =================
    if constexpr (SHARED) {
        if (lane < ng) sS[S_R][wv * G + lane] = Rv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(&sS[S_CNT][0], 1u);
        if (rfl(old) != (blockDim.x >> 6) - 1u) return;     // not the last wave of the workgroup
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t pos = sS[S_POS][lane];
        S.ax = pos & 0xffffu;
        S.ay = pos >> 16;
        S.fl = sS[S_FL][lane];
        S.episode = sS[S_EP][lane];
        S.tick = sS[S_TICK][lane];
        S.Pv = sS[S_P][lane];
        Rv = sS[S_R][lane];
    }
    group_phase_c<W, H, 7>(a, S, Rv, lane_env, ek, do_update);
}

// ----------------------------------------------------------------------------
// single-step kernel (n_steps == 1): ROWS layout, several environments per wave iteration
// ----------------------------------------------------------------------------
// A lane owns 1/S of a grid ROW (CPL = W/S consecutive cells = 4*CPL contiguous bytes), an
// environment owns LPE = H*S consecutive lanes, and EPW environments sit side by side in the
// wave, separated by S idle lanes, so that a ballot mask shifted by one grid row (S lanes) can
// never leak into the neighbouring environment.  Compared with the 4-cells-per-lane layout of
// wf_step_group (14x14: one environment on 49 lanes, 4 cell slots):
//   * one pass over the CPL cell slots serves EPW environments (14x14, S=2: 7 slots for 2
//     environments = 3.5 per environment), and so does every scalar mask operation;
//   * left/right neighbours are mostly the lane's own neighbouring slots (a 0/1 register, no
//     mask shift, no select); only the row seam inside a split row (S>1) needs a lane shift;
//   * the EPW environments of an iteration are contiguous in memory (env stride = LPE lanes x
//     4*CPL bytes), so the wave still moves one dense span per iteration.
// Everything per-environment is expressed on lane masks (RUNM = lanes of running environments,
// ...), never on divergent control flow; per-environment scalars are cut out of the masks with
// the constant lane sets hm(h).  Phases A and C are those of wf_step_group; the environment
// registers travel through LDS in both the private and the workgroup-shared variant.
// environments per wave iteration: as many as fit with S idle lanes between them AND one idle lane above the
// last one (the carry-out trick of the result extraction needs that lane: top lane index <= 62)
constexpr int rg_epw(int H, int S) { return (63 + S) / (H * S + S); }

constexpr u64 rg_lanes(int W, int H, int S, int kind, int j)
{
    const int LPE = H * S, PITCH = LPE + S, CPL = W / S, EPW = rg_epw(H, S);
    u64 m = 0;
    for (int h = 0; h < EPW; ++h)
        for (int l = 0; l < LPE; ++l) {
            const int y = l / S, x = (l % S) * CPL + j;
            const bool b = kind == 0 ? true
                         : kind == 1 ? (l % S != 0)
                         : kind == 2 ? (l % S != S - 1)
                         : kind == 3 ? (x == 0 || y == 0 || x == W - 1 || y == H - 1)
                         : kind == 4 ? (y == 0 || y == H - 1)
                                     : (l == LPE - 1);
            if (b) m |= 1ull << (h * PITCH + l);
        }
    return m;
}

template <int W, int H, int S>
struct RG {
    static_assert(W % S == 0, "a row must split evenly over S lanes");
    static constexpr int CELLS = W * H, CPL = W / S, LPE = H * S, PITCH = LPE + S;
    static constexpr int EPW = rg_epw(H, S);
    static_assert(LPE < 64 && EPW >= 1 && EPW * PITCH - S < 64, "environment does not fit the wave in this layout");
    static constexpr u64 ENV0 = (1ull << LPE) - 1ull;
    static constexpr u64 hm(int h) { return ENV0 << (h * PITCH); }          // lanes of environment h
    static constexpr u64 CELL_LANES = rg_lanes(W, H, S, 0, 0);
    static constexpr u64 NOT_ROWSTART = rg_lanes(W, H, S, 1, 0);
    static constexpr u64 NOT_ROWEND = rg_lanes(W, H, S, 2, 0);
    template <int J> static constexpr u64 BORDER = rg_lanes(W, H, S, 3, J);
    static constexpr u64 ROW_EDGE = rg_lanes(W, H, S, 4, 0);                // lanes of the first and last grid row
    static constexpr u64 COL_FIRST = CELL_LANES & ~NOT_ROWSTART;            // lanes whose slot 0 is column 0
    static constexpr u64 COL_LAST = CELL_LANES & ~NOT_ROWEND;               // lanes whose last slot is column W-1
    static constexpr u64 WRITERS = rg_lanes(W, H, S, 5, 0);                 // last lane of every environment
    static constexpr int REACH_WORDS = CPL;                                   // u64 per group of EPW environments
};

template <int W, int H, int S>
using RowMasks = u64[RG<W, H, S>::CPL];          // one lane mask per cell slot

// neighbour sets of a per-slot cell set M (bit L of M[j] <-> slot j of lane L)
template <int W, int H, int S, int J>
__device__ __forceinline__ u64 rg_lf(const RowMasks<W, H, S> &M)
{
    using R = RG<W, H, S>;
    if constexpr (J > 0) return M[J - 1];
    else if constexpr (S > 1) return (M[R::CPL - 1] << 1) & R::NOT_ROWSTART;
    else return 0ull;
}
template <int W, int H, int S, int J>
__device__ __forceinline__ u64 rg_rt(const RowMasks<W, H, S> &M)
{
    using R = RG<W, H, S>;
    if constexpr (J < R::CPL - 1) return M[J + 1];
    else if constexpr (S > 1) return (M[0] >> 1) & R::NOT_ROWEND;
    else return 0ull;
}

// cells of P reachable from the seed set R (in-place relaxation, wave-uniform loop)
template <int W, int H, int S>
__device__ __forceinline__ void flood_rows(RowMasks<W, H, S> &R, const RowMasks<W, H, S> &P)
{
    constexpr int CPL = RG<W, H, S>::CPL;
    for (int it = 0; it < W * H; ++it) {
        u64 any = 0;
        static_for<CPL>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const u64 add = ((R[j] << S) | (R[j] >> S) | rg_lf<W, H, S, j>(R) | rg_rt<W, H, S, j>(R)) & P[j] & ~R[j];
            R[j] |= add;
            any |= add;
        });
        if (!any) break;
    }
}

// N consecutive dwords at a 4-byte aligned address
template <int N>
__device__ __forceinline__ void load_dwords(const char *p, uint32_t *r)
{
    struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
    struct __attribute__((packed, aligned(4))) U3 { uint32_t x, y, z; };
    struct __attribute__((packed, aligned(4))) U2 { uint32_t x, y; };
    if constexpr (N >= 4) {
        const U4 v = *reinterpret_cast<const U4 *>(p);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        if constexpr (N > 4) load_dwords<N - 4>(p + 16, r + 4);
    } else if constexpr (N == 3) {
        const U3 v = *reinterpret_cast<const U3 *>(p);
        r[0] = v.x; r[1] = v.y; r[2] = v.z;
    } else if constexpr (N == 2) {
        const U2 v = *reinterpret_cast<const U2 *>(p);
        r[0] = v.x; r[1] = v.y;
    } else if constexpr (N == 1) {
        r[0] = *reinterpret_cast<const uint32_t *>(p);
    }
}
template <int N>
__device__ __forceinline__ void store_dwords(char *p, const uint32_t *r)
{
    struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
    struct __attribute__((packed, aligned(4))) U3 { uint32_t x, y, z; };
    struct __attribute__((packed, aligned(4))) U2 { uint32_t x, y; };
    if constexpr (N >= 4) {
        *reinterpret_cast<U4 *>(p) = U4{ r[0], r[1], r[2], r[3] };
        if constexpr (N > 4) store_dwords<N - 4>(p + 16, r + 4);
    } else if constexpr (N == 3) {
        *reinterpret_cast<U3 *>(p) = U3{ r[0], r[1], r[2] };
    } else if constexpr (N == 2) {
        *reinterpret_cast<U2 *>(p) = U2{ r[0], r[1] };
    } else if constexpr (N == 1) {
        *reinterpret_cast<uint32_t *>(p) = r[0];
    }
}

// World.reset() of ONE environment of the iteration into the rows layout (same stream and
// same law as reset_regs).  Every lane computes the cells "as if" it belonged to that
// environment (y, xs = its row and first column); the lanes of HM (`mine`) keep them.
// Returns the spawn cell in (s.ax, s.ay); writes the 4-cells-per-lane closed-form board too.
template <int W, int H, int S>
__device__ __forceinline__ void reset_rows(const WfCold *cold_g, u64 seed, uint32_t e_ctx, uint32_t y, uint32_t xs,
                                           uint32_t bgeo, u64 HM, bool mine, uint32_t (&c)[RG<W, H, S>::CPL], EnvS &s,
                                           u64 *reach_e, bool writer)
{
    constexpr int CELLS = W * H, CPL = RG<W, H, S>::CPL;
    cold_ptr cold = (cold_ptr)(uintptr_t)cold_g;
    const int32_t fuel_lo = cold->p.fuel_lo, fuel_hi = cold->p.fuel_hi;
    const uint32_t ppm = cold->p.ignite_ppm;
    const EnvKey key(seed, cold->p.env_id0 + (u64)e_ctx, s.episode);
    const uint32_t r = below(key.draw(0), 3u);                        // radius - 1
    const uint32_t k = below(key.draw(1), (uint32_t)cold->spawn.n[r]);   // point on the circle
    const uint32_t fx = W / 2, fy = H / 2;                            // utility.py:61-64
    const uint32_t sx = fx + (int)cold_g->spawn.dx[r][k], sy = fy + (int)cold_g->spawn.dy[r][k];
    const uint32_t qc = fy * W + fx, qs = sy * W + sx;
    const uint32_t span = (uint32_t)(fuel_hi - fuel_lo + 1);
    const uint32_t q0 = y * W + xs;
    u64 onborder = 0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const uint32_t q = q0 + (uint32_t)j;
        int32_t fuel = fuel_lo;
        if (span > 1u) fuel += (int32_t)below(key.draw(16u + q), span);
        uint32_t ty = WF_GRASS, heat = 0;
        if (q == qc) { ty = WF_FIRE; heat = WF_HEAT_FORCED; }         // set_fire_to :233-246
        if (q == qs) ty = WF_DIRT;                                    // Agent.__init__ digs :111-113
        bool extra = false;
        if (ppm) extra = ty == WF_GRASS && below(key.draw(16u + CELLS + q), 1000000u) < ppm;
        if (extra) { ty = WF_FIRE; heat = WF_HEAT_FORCED; }
        onborder |= __ballot(extra && ((bgeo >> j) & 1u));
        const uint32_t fresh = ty | (((uint32_t)fuel & 0xffu) << 8) | (heat << 16) | (q == qs ? (1u << 24) : 0u);
        c[j] = mine ? fresh : c[j];
    }
    onborder &= HM;
    if (writer) {   // the closed-form board of the 4-cells-per-lane kernels (see reset_regs)
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
    s.fl = WF_F_ALIVE | WF_F_VISIBLE | WF_F_DIGGING | WF_F_RUNNING | WF_F_REACH_VALID | WF_F_REACH_ROWS |
           (onborder ? WF_F_BORDER : 0u);
    s.episode += 1;
    s.tick = 0;
}

template <int W, int H, int S, bool SHARED>
__global__ __launch_bounds__(SHARED ? 512 : 256, WF_LB_ROWS) void wf_step_rows(const WfArgs a)
{
    using R = RG<W, H, S>;
    constexpr int CELLS = W * H, CPL = R::CPL, LPE = R::LPE, PITCH = R::PITCH, EPW = R::EPW;
    constexpr uint32_t ENV_BYTES = CELLS * 4, OBS_BYTES = CELLS * 3;
    const uint32_t lane = __lane_id();
    const uint32_t batch = (uint32_t)a.batch;
    const uint32_t G = (uint32_t)a.group;                   // environments per wave: multiple of EPW, <= 16
    const uint32_t wv = rfl(threadIdx.x >> 6);
    // B: this wave steps [e0, e0+ng);  A/C: lanes cover [A0, A0+na);  gi0: the wave's slice of the LDS arrays
    uint32_t e0, ng, A0, na, gi0;
    if constexpr (SHARED) {
        const uint32_t per_wg = G * (blockDim.x >> 6);      // <= 64
        A0 = blockIdx.x * per_wg;
        na = min(per_wg, batch - A0);
        gi0 = wv * G;
        e0 = A0 + gi0;
        ng = gi0 < na ? min(G, na - gi0) : 0u;
    } else {
        e0 = rfl(blockIdx.x * 4u + wv) * G;
        if (e0 >= batch) return;
        ng = min(G, batch - e0);
        A0 = e0;
        na = ng;
        gi0 = wv * 16u;
    }
    const uint32_t ign = (uint32_t)a.ignite_count;
    const bool do_update = (a.update_mask & 1ull) != 0;
    char *planes_b = reinterpret_cast<char *>(a.planes);
    __shared__ uint32_t sS[7][64];                           // ax|ay<<16, fl, episode, tick, Pv, Rv, counter
    enum { S_POS = 0, S_FL = 1, S_EP = 2, S_TICK = 3, S_P = 4, S_R = 5, S_CNT = 6 };

    // ================= phase A: lanes = environments =================
    const bool lane_env = lane < na;
    const uint32_t ek = A0 + (lane_env ? lane : 0u);
    const uint32_t ai = SHARED ? lane : gi0 + (lane_env ? lane : 0u);   // LDS slot of this env-lane
    if (!SHARED || wv == 0) {
        const EnvRegs S0 = group_phase_a<W, H, 8>(a, lane_env, ek, do_update);
        if (lane_env) {
            sS[S_POS][ai] = S0.ax | (S0.ay << 16);
            sS[S_FL][ai] = S0.fl;
            sS[S_EP][ai] = S0.episode;
            sS[S_TICK][ai] = S0.tick;
            sS[S_P][ai] = S0.Pv;
            sS[S_R][ai] = 0;
        }
        if (SHARED && lane == 0) sS[S_CNT][0] = 0;
        // Other waves of the workgroup read the touched bytes back: the stores must have landed.  The wave's
        // own later loads need no wait: a wave's vector memory instructions reach its L1 in program order
        // (wavefront-scope fences are no-ops in the AMDGPU memory model), the fence below only stops the compiler.
        if (SHARED || WF_STORE_WAIT) __builtin_amdgcn_s_waitcnt(0);
    }
    if constexpr (SHARED) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // ================= phase B: EPW environments at a time, lanes = row pieces =================
    const uint32_t hh = lane / (uint32_t)PITCH, l = lane - hh * (uint32_t)PITCH;
    const bool cell_lane = hh < (uint32_t)EPW && l < (uint32_t)LPE;
    const uint32_t y = l / (uint32_t)S, xs = (l - y * (uint32_t)S) * (uint32_t)CPL;
    // bit j: slot j of this lane is a border cell (needed by the rare reset path only)
    uint32_t bgeo = (y == 0u || y == (uint32_t)(H - 1)) ? (1u << CPL) - 1u
                  : ((xs == 0u ? 1u : 0u) | (xs + (uint32_t)CPL == (uint32_t)W ? 1u << (CPL - 1) : 0u));
    uint32_t ob = hh * OBS_BYTES + (xs * H + y) * 3u;         // obs[x][y][k] image of the iteration's environments
    asm volatile("" : "+v"(bgeo), "+v"(ob));
    const uint32_t K_FIRE_STEP = 0xFFFFFF01u;                // fuel byte -1, type byte +1 (fire -> burnt)
    constexpr int OBS_PAD = (EPW * OBS_BYTES + 15) & ~15;
    __shared__ __attribute__((aligned(16))) uint8_t sObs[SHARED ? 8 : 4][OBS_PAD];
    uint8_t *obs_lds = sObs[wv];
    typedef const u64 __attribute__((address_space(4))) *const_u64_ptr;
    const bool cont = (a.hot_flags & WF_HOT_CONTAINMENT) != 0;

    if (cell_lane && ng) {
        // byte offset of this lane's cells in the current iteration; lanes whose environment does not
        // exist (ragged tail) shadow environment 0 of the iteration for loads and sit out otherwise
        uint32_t off = e0 * ENV_BYTES + l * (uint32_t)(CPL * 4);
        uint32_t nxt[CPL];
        load_dwords<CPL>(planes_b + (off + (hh < ng ? hh * ENV_BYTES : 0u)), nxt);
        for (uint32_t kb = 0; kb < ng; kb += (uint32_t)EPW, off += (uint32_t)EPW * ENV_BYTES) {
            if (kb + hh < ng) {
                const uint32_t eoff = off + hh * ENV_BYTES;
                uint32_t c[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) c[j] = nxt[j];
#if WF_SKIP_CLEAN
                uint32_t orig[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) orig[j] = nxt[j];
#endif
                load_dwords<CPL>(planes_b + (eoff + (kb + (uint32_t)EPW + hh < ng ? (uint32_t)EPW * ENV_BYTES : 0u)), nxt);
                const uint32_t idx = gi0 + kb + hh;          // LDS slot of this lane's environment
                const uint32_t Pl = sS[S_P][idx];
                const u64 VM = __ballot(true);
                const u64 RUNM = __ballot((Pl & PB_RUNNING) != 0u);
                // border-reachability board of this group of EPW environments (context-level group index)
                u64 *board = a.reach_rows + (size_t)((a.env_base + e0 + kb) / (uint32_t)EPW) * CPL;
                u64 RBG[CPL];
                bool rb_dirty = false;
                if (cont) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) RBG[j] = ((const_u64_ptr)(uintptr_t)board)[j];
                } else {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) RBG[j] = 0;
                }
                u64 WASRESET = 0;
                if (RUNM != VM && (a.hot_flags & WF_HOT_AUTO_RESET)) {
                    // ---- rare: finished environments start a new episode instead of stepping
                    WASRESET = VM & ~RUNM;
                    static_for<EPW>([&](auto HH) {
                        constexpr int h = decltype(HH)::value;
                        if (WASRESET & R::hm(h)) {
                            const uint32_t ei = kb + (uint32_t)h;
                            EnvS s = { 0, 0, 0, rfl(sS[S_EP][gi0 + ei]), 0 };
                            const WfCold *cold = a.cold;
                            uint32_t e_pin = e0 + ei;
                            uint32_t y_pin = y, xs_pin = xs, bgeo_pin = bgeo, hh_pin = hh;     // keep the rare path's per-lane
                            asm volatile("" : "+s"(cold), "+s"(e_pin), "+v"(y_pin), "+v"(xs_pin), "+v"(bgeo_pin), "+v"(hh_pin));   // work inside this branch
                            cold_ptr cc = (cold_ptr)(uintptr_t)cold;
                            reset_rows<W, H, S>(cold, cc->p.seed, e_pin + a.env_base, y_pin, xs_pin, bgeo_pin, R::hm(h), hh_pin == (uint32_t)h, c, s,
                                                a.reach + (size_t)e_pin * 4, lane == (uint32_t)(h * PITCH));
                            // rows board in closed form: every cell of the environment but the spawn cell
                            const uint32_t sax = rfl(s.ax), say = rfl(s.ay);      // (loaded through the vector path: keep the boards scalar)
                            const uint32_t sl = (uint32_t)(h * PITCH) + say * (uint32_t)S + sax / (uint32_t)CPL, sj = sax % (uint32_t)CPL;
#pragma unroll
                            for (int j = 0; j < CPL; ++j)
                                RBG[j] = (RBG[j] & ~R::hm(h)) | (R::hm(h) & ~(sj == (uint32_t)j ? (1ull << sl) : 0ull));
                            rb_dirty = true;
                            if (lane == (uint32_t)(h * PITCH)) {
                                const uint32_t i = gi0 + ei;
                                sS[S_POS][i] = s.ax | (s.ay << 16);
                                sS[S_FL][i] = s.fl;
                                sS[S_EP][i] = s.episode;
                                sS[S_TICK][i] = s.tick;
                            }
                        }
                    });
                }
                u64 FIREALL = 0, HITS = 0, BH = 0;                  // fire after the tick / fire on reachable cells / ignition on the border
                if (do_update) {
                    // ---- ForestFire.update (forest_fire.py:85-106)
                    u64 STILL[CPL];
                    uint32_t sv[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        // reduce_fuel (environment.py:297-307); finished environments hold no "fire" here
                        const u64 F = __ballot((c[j] & 0xffu) == (uint32_t)WF_FIRE) & RUNM;
                        STILL[j] = F & __ballot((int32_t)(int8_t)(c[j] >> 8) > 1);     // fuel - 1 > 0
                        const bool st = lanebit(STILL[j]);
                        sv[j] = st ? 1u : 0u;
                        const uint32_t d = st ? (K_FIRE_STEP - 1u) : K_FIRE_STEP;
                        c[j] += lanebit(F) ? d : 0u;
                    }
                    u64 IGNANY = 0, IGN0 = 0, IGNL = 0;
                    static_for<CPL>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        // get_neighbours radius 1 (:311-326) + apply_heat_from_to (:278-294): burning neighbours
                        // above / below come from the row masks, left / right mostly from the lane's own slots
                        const u64 GR = __ballot((c[j] & 0xffu) == (uint32_t)WF_GRASS);
                        uint32_t cnt;
                        if constexpr (j > 0 && j < CPL - 1) cnt = addc(sv[j - 1], sv[j + 1], STILL[j] << S);
                        else if constexpr (CPL == 1) cnt = addc0(addc0(addc0(0u, STILL[j] << S), rg_lf<W, H, S, j>(STILL)), rg_rt<W, H, S, j>(STILL));
                        else if constexpr (j == 0) cnt = S > 1 ? addc0(addc0(sv[j + 1], STILL[j] << S), rg_lf<W, H, S, j>(STILL)) : addc0(sv[j + 1], STILL[j] << S);
                        else cnt = S > 1 ? addc0(addc0(sv[j - 1], STILL[j] << S), rg_rt<W, H, S, j>(STILL)) : addc0(sv[j - 1], STILL[j] << S);
                        cnt = addc0(cnt, STILL[j] >> S);
                        const uint32_t room = ign - ((c[j] >> 16) & 0xffu);            // quanta still missing (>= 1 on grass)
                        const u64 IGN = GR & __ballot(cnt >= room);
                        const uint32_t cn = addc(c[j], min(cnt, room) << 16, IGN);     // heat saturates; grass -> fire (+1)
                        c[j] = lanebit(GR) ? cn : c[j];
                        IGNANY |= IGN;
                        if constexpr (j == 0) IGN0 = IGN;
                        if constexpr (j == CPL - 1) IGNL = IGN;
                        const u64 FJ = STILL[j] | IGN;
                        FIREALL |= FJ;
                        HITS |= FJ & RBG[j];
                    });
                    // set_fire_to :245-246: a newly ignited border cell
                    BH = (IGNANY & R::ROW_EDGE) | (IGN0 & R::COL_FIRST) | (IGNL & R::COL_LAST);
                } else {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const u64 FJ = __ballot((c[j] & 0xffu) == (uint32_t)WF_FIRE) & RUNM;
                        FIREALL |= FJ;
                        HITS |= FJ & RBG[j];
                    }
                }
                // ---- containment part of World.get_reward (environment.py:345-377)
                const u64 NEEDM = __ballot((Pl & PB_PRECHECK) != 0u && ((Pl & PB_VALID) == 0u || (Pl & PB_RING) != 0u)) & RUNM;
                if (NEEDM) {
                    // rare: the board of some environment is unknown or a dig may have cut a region off
                    u64 PASS[CPL], NB[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) PASS[j] = __ballot((c[j] & 0xffu) < (uint32_t)WF_DIRT);
                    static_for<CPL>([&](auto J) { constexpr int j = decltype(J)::value; NB[j] = R::template BORDER<j> & PASS[j]; });
                    flood_rows<W, H, S>(NB, PASS);
                    HITS = 0;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        RBG[j] = (RBG[j] & ~NEEDM) | (NB[j] & NEEDM);
                        HITS |= __ballot((c[j] & 0xffu) == (uint32_t)WF_FIRE) & RUNM & RBG[j];
                    }
                    rb_dirty = true;
                }
                if (rb_dirty && lane == 0) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) board[j] = RBG[j];
                }
                // ---- per-environment results.  "Any lane of my environment" without a loop over the
                // environments: adding the all-ones lane set to a mask carries out of every environment
                // that holds a bit, into the idle lane above it; one lane down sits the environment's
                // last lane, which assembles and publishes the result word.
                const u64 ANYW = (FIREALL + R::CELL_LANES) >> 1;
                const u64 BHW = (BH + R::CELL_LANES) >> 1;
                const u64 HITW = (HITS + R::CELL_LANES) >> 1;
                const uint32_t any_l = lanebit(ANYW) ? 1u : 0u, bh_l = lanebit(BHW) ? 1u : 0u, hit_l = lanebit(HITW) ? 1u : 0u;
                const uint32_t run_l = Pl & 1u, pre_l = (Pl >> 1) & run_l;
                uint32_t Rl = any_l | (bh_l << 1) | ((pre_l & any_l & (bh_l ^ 1u) & (hit_l ^ 1u)) << 2) | (lanebit(NEEDM) ? (uint32_t)RB_RECOMPUTED : 0u);
                const u64 OUTM = __ballot(run_l != 0u && any_l == 0u) & R::WRITERS;     // running, no fire left
                if (OUTM) {                                                          // :384-387 burnt out (rare)
                    u64 GRM[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) GRM[j] = __ballot((c[j] & 0xffu) == (uint32_t)WF_GRASS);
                    static_for<EPW>([&](auto HH) {
                        constexpr int h = decltype(HH)::value;
                        if (OUTM & R::hm(h)) {
                            uint32_t healthy = 0;
#pragma unroll
                            for (int j = 0; j < CPL; ++j) healthy += (uint32_t)__popcll(GRM[j] & R::hm(h));
                            Rl |= hh == (uint32_t)h ? healthy << RB_HEALTHY_SHIFT : 0u;
                        }
                    });
                }
                Rl = lanebit(WASRESET) ? (uint32_t)RB_WASRESET : Rl;
                if (l == (uint32_t)(LPE - 1)) sS[S_R][idx] = Rl;
#if WF_SKIP_CLEAN
                {   // most cells do not change in a tick: a lane whose row piece is unchanged writes nothing
                    uint32_t chg = 0;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) chg |= c[j] ^ orig[j];
                    if (chg) store_dwords<CPL>(planes_b + eoff, c);
                }
#else
                store_dwords<CPL>(planes_b + eoff, c);
#endif
                if (a.obs) {
                    // World.get_state() of the iteration's environments: byte triples into the LDS image at
                    // their transposed offsets, then the image leaves as 16-byte stores
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const uint32_t t = c[j] & 0xffu, o = ob + (uint32_t)(j * H * 3);
                        obs_lds[o] = (uint8_t)(c[j] >> 24);                       // agent_pos
                        obs_lds[o + 1u] = (uint8_t)((2u >> t) & 1u);              // type == fire
                        obs_lds[o + 2u] = (uint8_t)((7u >> t) & 1u);              // not dirt / water
                    }
                    const uint32_t nenv = min((uint32_t)EPW, ng - kb);
                    const uint32_t nbytes = nenv * OBS_BYTES;                     // multiple of 4
                    int8_t *obs_g = a.obs + (size_t)(e0 + kb) * OBS_BYTES;
                    const uint32_t rank = (uint32_t)__popcll(VM & ((1ull << lane) - 1ull));   // dense index among the working lanes
                    const uint32_t nwork = (uint32_t)__popcll(VM);
                    struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
                    for (uint32_t o = rank * 16u; o < nbytes; o += nwork * 16u) {
                        if (o + 16u <= nbytes) {
                            const uint4 v = *reinterpret_cast<const uint4 *>(obs_lds + o);
                            *reinterpret_cast<U4 *>(obs_g + o) = U4{ v.x, v.y, v.z, v.w };
                        } else {
                            for (uint32_t q = o; q < nbytes; q += 4u)
                                *reinterpret_cast<uint32_t *>(obs_g + q) = *reinterpret_cast<const uint32_t *>(obs_lds + q);
                        }
                    }
                }
            }
        }
    }

    // ================= phase C: lanes = environments =================
    if constexpr (SHARED) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t old = 0;
        if (lane == 0) old = atomicAdd(&sS[S_CNT][0], 1u);
        if (rfl(old) != (blockDim.x >> 6) - 1u) return;     // not the last wave of the workgroup
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    {
        EnvRegs S1;
        const uint32_t pos = sS[S_POS][ai];
        S1.ax = pos & 0xffffu;
        S1.ay = pos >> 16;
        S1.fl = sS[S_FL][ai];
        S1.episode = sS[S_EP][ai];
        S1.tick = sS[S_TICK][ai];
        S1.Pv = sS[S_P][ai];
        const uint32_t Rv = sS[S_R][ai];
        group_phase_c<W, H, 8>(a, S1, Rv, lane_env, ek, do_update);
    }
}

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
        reset_regs<W, H>(a.cold, a.reset_seed, e, lane, t, u, h, s, a.reach + (size_t)e * 4);
        const uint32_t qa = s.ay * W + s.ax;
        uint32_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = t[j] | (u[j] << 8) | (h[j] << 16) |
                   ((lane == (qa >> 2) && (qa & 3u) == (uint32_t)j) ? (1u << 24) : 0u);
        if (lane < (uint32_t)LANES)
            reinterpret_cast<uint4 *>(a.planes + (size_t)e * CELLS)[lane] = make_uint4(r[0], r[1], r[2], r[3]);
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

// World.get_state() (environment.py:399-402): obs[b][x][y][k], any grid shape.
__global__ __launch_bounds__(256) void wf_get_state_kernel(const uint32_t *__restrict__ planes,
                                                          int8_t *__restrict__ obs, int batch, int H, int W)
{
    const size_t cells = (size_t)H * (size_t)W;
    const size_t total = (size_t)batch * cells;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / cells, r = i % cells;
        const uint32_t x = (uint32_t)(r / H), y = (uint32_t)(r % H);
        const uint32_t c = planes[b * cells + (size_t)y * W + x];
        const uint32_t t = c & 0xffu;
        int8_t *o = obs + i * 3;
        o[0] = (int8_t)(c >> 24);
        o[1] = (int8_t)(t == WF_FIRE);
        o[2] = (int8_t)(t != WF_DIRT && t != WF_WATER);
    }
}

// Host uploads replace planes/scalars behind the context's back: the cached
// border-reachability bitboards are no longer trustworthy.
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

template <int W, int H>
StepPlan plan_single(const WfArgs &a, int num_cus)
{
    StepPlan p;
    p.G = wf_pick_group(W * H, num_cus, a.batch);
    const int gmax = wf_pick_group(W * H, num_cus, 1ll << 40);
    if (a.group > 0) p.G = a.group < gmax ? a.group : gmax;
    const bool shared = getenv("WF_SHARED") && (p.G == 8 || p.G == 16) && a.batch >= 4096;
    p.kind = shared ? STEP_GROUP_SHARED : STEP_GROUP;
    if constexpr (HasRows<W, H>::value) {
        constexpr int EPW = RG<W, H, HasRows<W, H>::S>::EPW;
        if (p.G % EPW == 0 && p.G <= 16 && a.env_base % EPW == 0 && !getenv("WF_NO_ROWS"))
            p.kind = shared ? STEP_ROWS_SHARED : STEP_ROWS;
    }
    return p;
}

template <int W, int H>
hipError_t launch_wave(bool reset, const WfArgs &a, hipStream_t s, int num_cus)
{
    int blocks = (a.batch + 3) / 4;
    const int cap = num_cus * 8;            // up to 32 resident waves per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (reset)
        hipLaunchKernelGGL((wf_reset_wave<W, H>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.n_steps > 1)
        hipLaunchKernelGGL((wf_step_wave<W, H, true>), dim3(blocks), dim3(256), 0, s, a);
    else {
        const StepPlan p = plan_single<W, H>(a, num_cus);
        WfArgs g = a;
        g.group = p.G;
        const int waves = (a.batch + p.G - 1) / p.G;
        const dim3 grid_private((waves + 3) / 4), grid_shared((a.batch + 63) / 64), block_shared(64 * (64 / p.G));
        if constexpr (HasRows<W, H>::value) {
            constexpr int S = HasRows<W, H>::S;
            if (p.kind == STEP_ROWS) hipLaunchKernelGGL((wf_step_rows<W, H, S, false>), grid_private, dim3(256), 0, s, g);
            if (p.kind == STEP_ROWS_SHARED) {
                // WF_SHARED=N with N in {2,4,8}: N waves share the agent phases (N*G <= 64 environments per workgroup)
                int nw = atoi(getenv("WF_SHARED"));
                if (nw < 2 || nw * p.G > 64) nw = 64 / p.G;
                const int per_wg = nw * p.G;
                hipLaunchKernelGGL((wf_step_rows<W, H, S, true>), dim3((a.batch + per_wg - 1) / per_wg), dim3(64 * nw), 0, s, g);
            }
        }
        if (p.kind == STEP_GROUP) hipLaunchKernelGGL((wf_step_group<W, H, false>), grid_private, dim3(256), 0, s, g);
        if (p.kind == STEP_GROUP_SHARED) hipLaunchKernelGGL((wf_step_group<W, H, true>), grid_shared, block_shared, 0, s, g);
    }
    return hipGetLastError();
}

#define WF_WAVE_SHAPES(X) X(8, 8) X(10, 10) X(12, 12) X(14, 14) X(16, 16) X(16, 8) X(8, 16) X(20, 10) X(10, 20)

hipError_t dispatch_wave(bool reset, const WfArgs &a, hipStream_t s, int num_cus)
{
#define X(w, h) if (a.W == w && a.H == h) return launch_wave<w, h>(reset, a, s, num_cus);
    WF_WAVE_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace

WfFamily wf_pick_family(int H, int W)
{
#define X(w, h) if (W == w && H == h) return WF_FAM_WAVE;
    WF_WAVE_SHAPES(X)
#undef X
    if (W >= 4 && W <= 256 && W % 4 == 0 && H >= 1) return WF_FAM_TILED;
    return WF_FAM_CELLS;
}

const char *wf_family_name(WfFamily f, int, int)
{
    switch (f) {
    case WF_FAM_WAVE: return "wf_step_rows/wf_step_group/wf_step_wave (whole grids in a wave's registers, CA core on ballot bitboards; env logic lane-parallel)";
    case WF_FAM_TILED: return "wf_step_tiled (16-row strips per wave, ballot row masks, LDS-staged strip halos)";
    case WF_FAM_CELLS: return "wf_step_cells (generic thread-per-cell)";
    default: return "none";
    }
}

// symbol of the kernel that would serve a single step of `a` (introspection for bench / profiles)
const char *wf_single_step_kernel(WfFamily fam, const WfArgs &a, int num_cus)
{
    static thread_local char buf[64];
    if (fam == WF_FAM_TILED) return "wf_step_tiled";
    if (fam == WF_FAM_CELLS) return "wf_step_cells";
    StepPlan p = { 0, -1 };
    int S = 1;
#define X(w, h) if (a.W == w && a.H == h) { p = plan_single<w, h>(a, num_cus); S = HasRows<w, h>::S; }
    WF_WAVE_SHAPES(X)
#undef X
    if (p.kind < 0) return "none";
    if (p.kind >= STEP_ROWS) snprintf(buf, sizeof buf, "wf_step_rows<%d,%d,%d,%s>", a.W, a.H, S, p.kind == STEP_ROWS_SHARED ? "true" : "false");
    else snprintf(buf, sizeof buf, "wf_step_group<%d,%d,%s>", a.W, a.H, p.kind == STEP_GROUP_SHARED ? "true" : "false");
    return buf;
}

hipError_t wf_launch_step(WfFamily fam, const WfArgs &a, hipStream_t s, int num_cus)
{
    if (fam == WF_FAM_WAVE) return dispatch_wave(false, a, s, num_cus);
    return hipErrorInvalidValue;
}

hipError_t wf_launch_reset(WfFamily fam, const WfArgs &a, hipStream_t s, int num_cus)
{
    if (fam == WF_FAM_WAVE) return dispatch_wave(true, a, s, num_cus);
    return hipErrorInvalidValue;
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
    const size_t total = (size_t)batch * H * W;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wf_get_state_kernel, dim3((unsigned)blocks), dim3(256), 0, s, planes, obs, batch, H, W);
    return hipGetLastError();
}
