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
//     (196 cells = 49 lanes x 16 B, contiguous 784 B per env).
//   * every per-cell predicate is turned into a 64-bit lane mask with a wavefront
//     ballot (v_cmp -> SGPR pair); the 4-neighbour "still burning" counts are
//     built from those masks with scalar shifts (neighbour q+-1 / q+-W of component
//     j lives at a fixed (lane offset, component) pair known at compile time) and
//     read back per lane with v_cndmask on the mask.  No LDS, no barrier, no
//     per-cell index arithmetic.
//   * everything per-environment (agent move/dig/death, fire_at_border, RUNNING,
//     reward, containment bitboards) is wave-uniform scalar work on the same masks.
//   * n_steps fused steps keep the grid in registers; HBM traffic is one read and
//     one write of the grid per launch.
//
// No MFMA anywhere: the path is HBM-bound byte work, not a contraction.
#include <utility>

#include "wf_internal.h"

namespace {

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

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool lanebit(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ u64 sel4(const u64 (&m)[4], int k)
{
    return k == 0 ? m[0] : k == 1 ? m[1] : k == 2 ? m[2] : m[3];
}

// Neighbour masks of a per-component cell set S (bit l of S[j] <-> cell 4l+j):
// N[j] has bit l set iff some 4-neighbour of cell 4l+j is in S.
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

// ----------------------------------------------------------------------------
// counter-based stream (same arithmetic as oracle/wf_oracle.c, written independently)
// ----------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t below(uint32_t r, uint32_t n) { return __umulhi(r, n); }

struct EnvKey {
    uint32_t k0, k1;
    __device__ __forceinline__ EnvKey(u64 seed, u64 env_id, uint32_t episode)
    {
        u64 k = mix64(mix64(seed ^ (env_id * 0x9E3779B97F4A7C15ull)) + (u64)episode);
        k0 = (uint32_t)k;
        k1 = (uint32_t)(k >> 32);
    }
    __device__ __forceinline__ uint32_t draw(uint32_t d) const { return hash32(k0 ^ hash32(d + k1)); }
};

struct EnvS {        // wave-uniform scalars of one environment
    uint32_t ax, ay, fl, episode, tick;
};

// World.reset() into registers (environment.py:186-212, reset_map :59-67,
// get_agent_location utility.py:66-78) with the counter-based stream.
template <int W, int H>
__device__ __forceinline__ void reset_regs(const WfArgs &a, u64 seed, int e, int lane,
                                           uint32_t (&t)[4], uint32_t (&u)[4], uint32_t (&h)[4], EnvS &s)
{
    constexpr int CELLS = W * H;
    const EnvKey key(seed, a.p.env_id0 + (u64)e, s.episode);
    const uint32_t r = below(key.draw(0), 3u);                       // radius - 1
    const uint32_t k = below(key.draw(1), (uint32_t)a.spawn.n[r]);   // point on the circle
    const uint32_t fx = W / 2, fy = H / 2;                           // utility.py:61-64
    const uint32_t sx = fx + (int)a.spawn.dx[r][k], sy = fy + (int)a.spawn.dy[r][k];
    const uint32_t qc = fy * W + fx, qs = sy * W + sx;
    const bool active = lane < CELLS / 4;
    const uint32_t span = (uint32_t)(a.p.fuel_hi - a.p.fuel_lo + 1);
    u64 onborder = 0;
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const uint32_t q = 4u * (uint32_t)lane + j;
        int32_t fuel = a.p.fuel_lo;
        if (span > 1u) fuel += (int32_t)below(key.draw(16u + q), span);
        uint32_t ty = WF_GRASS, heat = 0;
        if (q == qc) { ty = WF_FIRE; heat = WF_HEAT_FORCED; }        // set_fire_to :233-246
        if (q == qs) ty = WF_DIRT;                                   // Agent.__init__ digs :111-113
        bool extra = false;
        if (a.p.ignite_ppm)
            extra = ty == WF_GRASS && below(key.draw(16u + CELLS + q), 1000000u) < a.p.ignite_ppm;
        if (extra) { ty = WF_FIRE; heat = WF_HEAT_FORCED; }
        onborder |= __ballot(extra && active) & CM<W, H, j>::BORDER;
        t[j] = active ? ty : (uint32_t)WF_DIRT;
        u[j] = active ? ((uint32_t)fuel & 0xffu) : 0u;
        h[j] = active ? heat : 0u;
    });
    s.ax = sx;
    s.ay = sy;
    s.fl = WF_F_ALIVE | WF_F_VISIBLE | WF_F_DIGGING | WF_F_RUNNING | (onborder ? WF_F_BORDER : 0u);
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

// Containment shortcut used by the step kernel: removing a cell q from the passable
// set can only change which cells are connected to the border when q is itself a
// border cell, or when the passable cells on the 8-cell ring around q (cyclic order
// N,NE,E,SE,S,SW,W,NW; off-grid = impassable) form >= 2 separate runs.  Otherwise any
// path through q re-routes along the single run.  Lanes 0..7 each inspect one ring
// cell (ds_bpermute gather of the packed types).

// ----------------------------------------------------------------------------
// the step kernel: n_steps x ForestFire.step() (forest_fire.py:30-49) per env
// ----------------------------------------------------------------------------
// Work split inside a wave (the scalar unit is one instruction per clock for the
// whole CU, the four SIMDs together issue two vector instructions per clock):
//   scalar unit : only the 64-bit mask algebra (neighbour shifts, reachability
//                 bitboards, any/or reductions) and the rare uniform branches;
//   vector unit : the per-cell rule AND the per-environment flag logic, the latter
//                 branch-free on values that are identical in all 64 lanes.
typedef const u64 __attribute__((address_space(4))) *const_u64_ptr;

// move table, 6 bits per action 0..4: (dx+1) | (dy+1)<<2 | is_move<<4; entry 4 = "no move"
// (N=(0,-1) S=(0,+1) E=(+1,0) W=(-1,0): environment.py:163-171)
constexpr uint32_t MOVE_LUT = (1u | 0u << 2 | 16u) | (1u | 2u << 2 | 16u) << 6 | (2u | 1u << 2 | 16u) << 12 |
                              (0u | 1u << 2 | 16u) << 18 | (1u | 1u << 2) << 24;

__device__ __forceinline__ uint32_t bperm(uint32_t src_lane, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
__device__ __forceinline__ uint32_t pack_types(const uint32_t (&t)[4])
{
    return t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
}
// type of cell q, q identical in all lanes but held in a VGPR
__device__ __forceinline__ uint32_t type_of(uint32_t tp, uint32_t q)
{
    return __builtin_amdgcn_ubfe(bperm(q >> 2, tp), (q & 3u) << 3, 8u);
}

template <int W, int H>
__global__ __launch_bounds__(256) void wf_step_wave(const WfArgs a)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    const uint32_t lane = __lane_id();
    const int wave0 = (int)rfl(blockIdx.x * 4u + (threadIdx.x >> 6));
    const int nwaves = (int)gridDim.x * 4;
    const bool active = lane < (uint32_t)LANES;
    const uint32_t ign = (uint32_t)a.p.ignite_count;
    const uint4 *planes4 = reinterpret_cast<const uint4 *>(a.planes);
    // ring geometry for the containment shortcut: lanes 0..7 <-> N,NE,E,SE,S,SW,W,NW
    const uint32_t rk = lane & 7u;
    const int rdx = (int)((0x01A9u >> (2u * rk)) & 3u) - 1;
    const int rdy = (int)((0x1A90u >> (2u * rk)) & 3u) - 1;

    uint4 raw_next = make_uint4(0, 0, 0, 0);
    if (wave0 < a.batch && active) raw_next = planes4[(size_t)wave0 * LANES + lane];

    for (int e = wave0; e < a.batch; e += nwaves) {
        const uint4 raw = raw_next;
        {   // software prefetch of this wave's next environment
            const int en = e + nwaves;
            if (en < a.batch && active) raw_next = planes4[(size_t)en * LANES + lane];
        }
        // env record: same 16 bytes in every lane, kept in VGPRs on purpose
        const uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);
        uint32_t ax = ev.x & 0xffffu, ay = ev.x >> 16, fl = ev.y & 0xffffu, episode = ev.z, tick = ev.w;
        uint32_t act = a.actions ? (uint32_t)(uint8_t)a.actions[e] : 0xffu;
        // border-reachability bitboards through the scalar cache
        u64 RB[4];
        {
            const_u64_ptr rp = (const_u64_ptr)(uintptr_t)(a.reach + (size_t)e * 4);
            RB[0] = rp[0]; RB[1] = rp[1]; RB[2] = rp[2]; RB[3] = rp[3];
        }
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
        bool dirty = false, rb_dirty = false;

        for (int st = 0; st < a.n_steps; ++st) {
            const size_t oi = (size_t)st * (size_t)a.batch + (size_t)e;
            uint32_t act_next = 0xffu;
            if (st + 1 < a.n_steps && a.actions) act_next = (uint32_t)(uint8_t)a.actions[oi + (size_t)a.batch];

            if (rfl(fl & WF_F_RUNNING) == 0u) {
                // ---- rare: finished environment
                bool done = true;
                if (a.p.auto_reset) {
                    // gymnasium-style next-step autoreset: this step returns the fresh episode
                    EnvS s = { rfl(ax), rfl(ay), rfl(fl), rfl(episode), rfl(tick) };
                    reset_regs<W, H>(a, a.p.seed, e, (int)lane, t, u, h, s);
                    ax = s.ax; ay = s.ay; fl = s.fl; episode = s.episode; tick = s.tick;
                    dirty = true;
                    done = false;
                }
                if (lane == 0) {
                    if (a.reward) a.reward[oi] = 0.0;
                    if (a.done) a.done[oi] = done ? 1 : 0;
                }
                act = act_next;
                continue;
            }
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
            if (a.p.allow_dig_toggle) {
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
            u64 NEWFIRE[4];
            const bool do_update = (a.update_mask >> st) & 1ull;
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
                u64 border_hit = 0;
                static_for<4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    t[j] -= lanebit(STILL[j]) ? 1u : 0u;                           // ... unless still burning
                    // get_neighbours radius 1 (:311-326) + apply_heat_from_to (:278-294)
                    const uint32_t hn = h[j] + (lanebit(nb_up<W, H, j>(STILL)) ? 1u : 0u) +
                                               (lanebit(nb_dn<W, H, j>(STILL)) ? 1u : 0u) +
                                               (lanebit(nb_lf<W, H, j>(STILL)) ? 1u : 0u) +
                                               (lanebit(nb_rt<W, H, j>(STILL)) ? 1u : 0u);
                    const uint32_t key2 = (t[j] << 8) + hn;
                    const u64 IGN = __ballot((key2 - ign) < (256u - ign));         // grass && hn >= ign
                    h[j] = (t[j] == WF_GRASS) ? min(hn, ign) : h[j];
                    t[j] += lanebit(IGN) ? 1u : 0u;                                // grass -> fire
                    border_hit |= IGN & CM<W, H, j>::BORDER;                       // set_fire_to :245-246
                    NEWFIRE[j] = STILL[j] | IGN;
                });
                fl |= border_hit ? (uint32_t)WF_F_BORDER : 0u;
                tick += 1;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) NEWFIRE[j] = __ballot(t[j] == WF_FIRE);
            }
            const uint32_t anyfire = (NEWFIRE[0] | NEWFIRE[1] | NEWFIRE[2] | NEWFIRE[3]) ? 1u : 0u;   // wave-uniform
            if (do_update)
                fl &= ~(((fl & anyfire) ^ 1u) << 4);                               // :105-106 (!alive || !fire) -> !RUNNING

            // ---- World.get_reward (environment.py:342-390)
            uint32_t contained = 0;
            // :345-346  containment test runs while the fire has not touched the border
            const uint32_t check = (uint32_t)(a.p.containment != 0) & anyfire & (((fl >> 5) | (fl >> 6)) ^ 1u) & 1u;
            if (rfl(check)) {
                // A dug cell can only cut the fire off from the border when it is a border cell
                // itself or when the passable cells on its 8-ring form >= 2 runs (see above).
                uint32_t recompute = ((fl >> 7) ^ 1u) & 1u;                        // bitboard not valid yet
                {
                    const uint32_t xd = dug ? ax : 1u, yd = dug ? ay : 1u;         // dug cell == agent cell
                    const uint32_t onb = (((xd - 1u) | (yd - 1u) | ((uint32_t)(W - 2) - xd) | ((uint32_t)(H - 2) - yd)) >> 31);
                    const uint32_t rx = xd + (uint32_t)rdx, ry = yd + (uint32_t)rdy;
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
                    fl |= WF_F_REACH_VALID;
                    rb_dirty = true;
                }
                // stale RB bits on cells dug since the last flood are harmless: fire never sits there
                contained = ((NEWFIRE[0] & RB[0]) | (NEWFIRE[1] & RB[1]) | (NEWFIRE[2] & RB[2]) | (NEWFIRE[3] & RB[3])) ? 0u : 1u;
                fl |= contained << 6;                                              // :361-367
            } else {
                fl &= ~(dug << 7);                                                 // bitboard no longer trustworthy
            }
            double reward = a.p.default_reward;                                    // :390
            if (rfl(anyfire ^ 1u)) {                                               // :384-387 burnt out (rare)
                int healthy = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) healthy += __popcll(__ballot(t[j] == WF_GRASS));
                reward = a.p.contained_bonus * ((double)healthy / (double)CELLS);
            }
            reward = (fl & WF_F_ALIVE) ? reward : a.p.death_penalty;               // :380-381
            reward = contained ? a.p.contained_bonus : reward;
            if (lane == 0) {
                if (a.reward) a.reward[oi] = reward;
                if (a.done) a.done[oi] = (uint8_t)(((fl >> 4) & 1u) ^ 1u);          // forest_fire.py:48
            }
            act = act_next;
        }

        if (dirty) {
            const uint32_t qa = ay * W + ax;
            const uint32_t sel = ((fl & WF_F_VISIBLE) && lane == (qa >> 2)) ? (qa & 3u) : 4u;
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[j] = t[j] | (u[j] << 8) | (h[j] << 16) | (sel == (uint32_t)j ? (1u << 24) : 0u);
            if (active) reinterpret_cast<uint4 *>(a.planes)[(size_t)e * LANES + lane] = make_uint4(r[0], r[1], r[2], r[3]);
            if (lane == 0) {
                *reinterpret_cast<uint4 *>(&a.envs[e]) = make_uint4(ax | (ay << 16), fl, episode, tick);
                if (rb_dirty) {
                    ulonglong2 *rp = reinterpret_cast<ulonglong2 *>(a.reach + (size_t)e * 4);
                    rp[0] = make_ulonglong2(RB[0], RB[1]);
                    rp[1] = make_ulonglong2(RB[2], RB[3]);
                }
            }
        }
    }
}


// World.reset() for masked envs (environment.py:186-212)
template <int W, int H>
__global__ __launch_bounds__(256) void wf_reset_wave(const WfArgs a)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    const int lane = threadIdx.x & 63;
    const int wave0 = (int)rfl(blockIdx.x * 4u + (threadIdx.x >> 6));
    const int nwaves = (int)gridDim.x * 4;
    for (int e = wave0; e < a.batch; e += nwaves) {
        if (a.mask && !rfl((uint32_t)a.mask[e])) continue;
        const uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);
        EnvS s;
        s.ax = s.ay = 0;
        s.fl = 0;
        s.episode = rfl(ev.z);
        s.tick = 0;
        uint32_t t[4], u[4], h[4];
        reset_regs<W, H>(a, a.reset_seed, e, lane, t, u, h, s);
        const uint32_t qa = s.ay * W + s.ax;
        uint32_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[j] = t[j] | (u[j] << 8) | (h[j] << 16) |
                   (((uint32_t)lane == (qa >> 2) && (qa & 3u) == (uint32_t)j) ? (1u << 24) : 0u);
        if (lane < LANES)
            reinterpret_cast<uint4 *>(a.planes + (size_t)e * CELLS)[lane] = make_uint4(r[0], r[1], r[2], r[3]);
        if (lane == 0)
            *reinterpret_cast<uint4 *>(&a.envs[e]) = make_uint4(s.ax | (s.ay << 16), s.fl, s.episode, s.tick);
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
    if (i < batch) envs[i].flags &= (uint16_t)~WF_F_REACH_VALID;
}

template <int W, int H>
hipError_t launch_wave(bool reset, const WfArgs &a, hipStream_t s, int num_cus)
{
    int blocks = (a.batch + 3) / 4;
    const int cap = num_cus * 8;            // 32 waves per CU = full occupancy at <= 64 VGPRs
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (reset)
        hipLaunchKernelGGL((wf_reset_wave<W, H>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((wf_step_wave<W, H>), dim3(blocks), dim3(256), 0, s, a);
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
    return WF_FAM_NONE;
}

const char *wf_family_name(WfFamily f, int, int)
{
    switch (f) {
    case WF_FAM_WAVE: return "wf_step_wave (one env per wavefront, ballot bitboards)";
    case WF_FAM_TILED: return "wf_step_tiled (row strips, LDS halo masks)";
    default: return "none";
    }
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
    const size_t total = (size_t)batch * H * W;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wf_get_state_kernel, dim3((unsigned)blocks), dim3(256), 0, s, planes, obs, batch, H, W);
    return hipGetLastError();
}
