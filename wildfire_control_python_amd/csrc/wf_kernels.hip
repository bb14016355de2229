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

// ----------------------------------------------------------------------------
// per-wave environment state held in registers
// ----------------------------------------------------------------------------
struct Cells {
    uint32_t t[4];   // type
    int32_t f[4];    // fuel (signed int8 range)
    uint32_t h[4];   // heat quanta
};

struct EnvS {        // wave-uniform scalars of one environment
    uint32_t ax, ay, fl, episode, tick;
};

template <int W, int H>
__device__ __forceinline__ void unpack(const uint4 &raw, Cells &c)
{
    const uint32_t r[4] = { raw.x, raw.y, raw.z, raw.w };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c.t[j] = r[j] & 0xffu;
        c.f[j] = (int32_t)(int8_t)(r[j] >> 8);
        c.h[j] = (r[j] >> 16) & 0xffu;
    }
}

template <int W, int H>
__device__ __forceinline__ uint4 pack(const Cells &c, const EnvS &s, int lane)
{
    const uint32_t qa = s.ay * W + s.ax;
    const bool here = (s.fl & WF_F_VISIBLE) && (uint32_t)lane == (qa >> 2);
    uint32_t r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t ag = (here && (qa & 3u) == (uint32_t)j) ? (1u << 24) : 0u;
        r[j] = c.t[j] | (((uint32_t)c.f[j] & 0xffu) << 8) | (c.h[j] << 16) | ag;
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

// type of cell q (q wave-uniform)
__device__ __forceinline__ uint32_t type_at(const Cells &c, uint32_t q)
{
    const uint32_t k = q & 3u;
    const uint32_t v = k == 0 ? c.t[0] : k == 1 ? c.t[1] : k == 2 ? c.t[2] : c.t[3];
    return __builtin_amdgcn_readlane(v, q >> 2);
}

__device__ __forceinline__ void set_type_at(Cells &c, uint32_t q, uint32_t ty, int lane)
{
    const bool here = (uint32_t)lane == (q >> 2);
    const uint32_t k = q & 3u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c.t[j] = (here && k == (uint32_t)j) ? ty : c.t[j];
}

// World.reset() into registers (environment.py:186-212, reset_map :59-67,
// get_agent_location utility.py:66-78) with the counter-based stream.
template <int W, int H>
__device__ __forceinline__ void reset_regs(const WfArgs &a, u64 seed, int e, int lane, Cells &c, EnvS &s)
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
        c.t[j] = active ? ty : 0u;
        c.f[j] = active ? fuel : 0;
        c.h[j] = active ? heat : 0u;
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

// Number of maximal runs of passable cells on the 8-cell ring around cell q
// (cyclic order N,NE,E,SE,S,SW,W,NW; off-grid = impassable).  Removing q from the
// passable set can only change connectivity between its neighbours when that
// number is >= 2.  Lanes 0..7 each look at one ring cell (ds_bpermute gather).
template <int W, int H>
__device__ __forceinline__ int ring_runs(const Cells &c, uint32_t q, int lane)
{
    const int x = (int)(q % W), y = (int)(q / W);
    const int k = lane & 7;
    // dx+1 / dy+1 packed 2 bits per ring position
    const int dx = (int)((0x01A9u >> (2 * k)) & 3u) - 1;   // 0,1,1,1,0,-1,-1,-1 -> 1,2,2,2,1,0,0,0
    const int dy = (int)((0x1A90u >> (2 * k)) & 3u) - 1;   // -1,-1,0,1,1,1,0,-1 -> 0,0,1,2,2,2,1,0
    const int nx = x + dx, ny = y + dy;
    const bool in = (unsigned)nx < (unsigned)W && (unsigned)ny < (unsigned)H;
    const uint32_t nq = in ? (uint32_t)(ny * W + nx) : 0u;
    const uint32_t tp = c.t[0] | (c.t[1] << 8) | (c.t[2] << 16) | (c.t[3] << 24);
    const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nq >> 2) << 2), (int)tp);
    const uint32_t ty = (v >> (8u * (nq & 3u))) & 0xffu;
    const uint32_t ring = (uint32_t)(__ballot(in && ty < WF_DIRT && lane < 8)) & 0xffu;
    const uint32_t rot = ((ring << 1) | (ring >> 7)) & 0xffu;
    return __popc(ring & ~rot);
}

// ----------------------------------------------------------------------------
// the step kernel: n_steps x ForestFire.step() (forest_fire.py:30-49) per env
// ----------------------------------------------------------------------------
template <int W, int H>
__global__ __launch_bounds__(256) void wf_step_wave(const WfArgs a)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    constexpr u64 VALID = CM<W, H, 0>::VALID;
    const int lane = threadIdx.x & 63;
    const int wave0 = (int)rfl(blockIdx.x * 4u + (threadIdx.x >> 6));
    const int nwaves = (int)gridDim.x * 4;
    const bool active = lane < LANES;
    const uint32_t ign = (uint32_t)a.p.ignite_count;

    for (int e = wave0; e < a.batch; e += nwaves) {
        // ---- environment scalars (one 16-byte record, same address in every lane)
        const uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);
        EnvS s;
        {
            const uint32_t xy = rfl(ev.x);
            s.ax = xy & 0xffffu;
            s.ay = xy >> 16;
            s.fl = rfl(ev.y) & 0xffffu;
            s.episode = rfl(ev.z);
            s.tick = rfl(ev.w);
        }
        uint4 *gp = reinterpret_cast<uint4 *>(a.planes + (size_t)e * CELLS) + lane;
        Cells c;
        bool dirty = false;
        if (s.fl & WF_F_RUNNING) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (active) raw = *gp;
            unpack<W, H>(raw, c);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { c.t[j] = 0; c.f[j] = 0; c.h[j] = 0; }
        }
        u64 RB[4] = { 0, 0, 0, 0 };          // cells connected to a passable border cell
        bool rb_loaded = false, rb_dirty = false;

        for (int st = 0; st < a.n_steps; ++st) {
            const size_t oi = (size_t)st * (size_t)a.batch + (size_t)e;
            double reward = 0.0;
            bool done;
            if (!(s.fl & WF_F_RUNNING)) {
                if (a.p.auto_reset) {
                    // gymnasium-style next-step autoreset: this step returns the fresh episode
                    reset_regs<W, H>(a, a.p.seed, e, lane, c, s);
                    dirty = true;
                    rb_loaded = true;        // nothing to load: the bitboard is invalid (flag cleared)
                    done = false;
                } else {
                    done = true;             // frozen: the reference is never stepped after done
                }
                if (lane == 0) {
                    if (a.reward) a.reward[oi] = reward;
                    if (a.done) a.done[oi] = done ? 1 : 0;
                }
                continue;
            }
            dirty = true;
            const int act = a.actions ? (int)(int8_t)rfl((uint32_t)a.actions[oi]) : -1;
            const bool do_update = (a.update_mask >> st) & 1ull;

            // ---- Agent.move (environment.py:141-155)
            bool dug = false;
            uint32_t qdug = 0;
            if ((unsigned)act < 4u) {
                s.fl &= ~(uint32_t)WF_F_VISIBLE;                                  // :142
                const int nx = (int)s.ax + (act == 2) - (act == 3);                // :163-171
                const int ny = (int)s.ay + (act == 1) - (act == 0);
                if ((unsigned)nx < (unsigned)W && (unsigned)ny < (unsigned)H) {    // :146 inbounds
                    const uint32_t qd = (uint32_t)(ny * W + nx);
                    const uint32_t td = type_at(c, qd);
                    if (td != WF_WATER) {                                          // :146 traversable
                        s.ax = (uint32_t)nx;                                       // :148-149
                        s.ay = (uint32_t)ny;
                        s.fl |= WF_F_VISIBLE;
                        if (td == WF_FIRE) {
                            s.fl |= WF_F_DEAD;                                     // :154-155
                        } else if ((s.fl & WF_F_DIGGING) && td != WF_DIRT) {       // :151-152, :125-133
                            dug = true;
                            qdug = qd;
                        }
                    }
                }
            }
            // ---- Agent.toggle_digging (forest_fire.py:35-36, environment.py:136-138)
            if (a.p.allow_dig_toggle && act == 4) {
                s.fl ^= WF_F_DIGGING;
                if (s.fl & WF_F_DIGGING) {
                    const uint32_t qa = s.ay * W + s.ax;
                    const uint32_t ta = type_at(c, qa);
                    // divergence (documented): the reference also retypes a BURNING cell to
                    // dirt here while keeping it in burning_cells; this build leaves it burning.
                    if (ta != WF_DIRT && ta != WF_FIRE) { dug = true; qdug = qa; }
                }
            }
            if (dug) set_type_at(c, qdug, WF_DIRT, lane);

            // ---- ForestFire.update (forest_fire.py:85-106)
            u64 FIRE[4], GRASS[4], NEWFIRE[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                FIRE[j] = __ballot(c.t[j] == WF_FIRE);
                GRASS[j] = __ballot(c.t[j] == WF_GRASS) & VALID;
            }
            if (do_update) {
                if (s.fl & WF_F_ALIVE) {                                           // :87, is_dead :116-120
                    const uint32_t qa = s.ay * W + s.ax;
                    const bool onfire = (sel4(FIRE, (int)(qa & 3u)) >> (qa >> 2)) & 1ull;
                    if ((s.fl & WF_F_DEAD) || onfire)
                        s.fl &= ~(uint32_t)(WF_F_ALIVE | WF_F_VISIBLE);
                }
                u64 STILL[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    STILL[j] = FIRE[j] & __ballot(c.f[j] > 1);                    // reduce_fuel :297-307
                    c.f[j] -= lanebit(FIRE[j]) ? 1 : 0;
                }
                u64 border_hit = 0;
                static_for<4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    // get_neighbours radius 1 (:311-326) + apply_heat_from_to (:278-294)
                    const uint32_t cnt = (lanebit(nb_up<W, H, j>(STILL)) ? 1u : 0u) +
                                         (lanebit(nb_dn<W, H, j>(STILL)) ? 1u : 0u) +
                                         (lanebit(nb_lf<W, H, j>(STILL)) ? 1u : 0u) +
                                         (lanebit(nb_rt<W, H, j>(STILL)) ? 1u : 0u);
                    const uint32_t hn = c.h[j] + cnt;
                    const u64 IGN = GRASS[j] & __ballot(hn >= ign);
                    c.h[j] = lanebit(GRASS[j]) ? min(hn, ign) : c.h[j];
                    c.t[j] = lanebit(FIRE[j] & ~STILL[j]) ? (uint32_t)WF_BURNT
                           : lanebit(IGN)                ? (uint32_t)WF_FIRE : c.t[j];
                    border_hit |= IGN & CM<W, H, j>::BORDER;                       // set_fire_to :245-246
                    NEWFIRE[j] = STILL[j] | IGN;
                    GRASS[j] &= ~IGN;
                });
                if (border_hit) s.fl |= WF_F_BORDER;
                s.tick += 1;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) NEWFIRE[j] = FIRE[j];
            }
            const bool anyfire = (NEWFIRE[0] | NEWFIRE[1] | NEWFIRE[2] | NEWFIRE[3]) != 0;
            if (do_update && (!(s.fl & WF_F_ALIVE) || !anyfire))
                s.fl &= ~(uint32_t)WF_F_RUNNING;                                   // :105-106

            // ---- World.get_reward (environment.py:342-390)
            bool paid = false;
            bool rb_updated = false;
            if (a.p.containment && !(s.fl & (WF_F_BORDER | WF_F_CONTAINED)) && anyfire) {   // :345-346
                u64 PASS[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) PASS[j] = VALID & ~__ballot(c.t[j] >= WF_DIRT);
                if (!rb_loaded) {
                    if (s.fl & WF_F_REACH_VALID) {
                        const uint32_t v = lane < 8 ? reinterpret_cast<const uint32_t *>(a.reach)[(size_t)e * 8 + lane] : 0u;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            RB[j] = (u64)__builtin_amdgcn_readlane(v, 2 * j) |
                                    ((u64)__builtin_amdgcn_readlane(v, 2 * j + 1) << 32);
                    }
                    rb_loaded = true;
                }
                bool recompute = !(s.fl & WF_F_REACH_VALID);
                if (!recompute && dug) {
                    const int kd = (int)(qdug & 3u);
                    const u64 bit = 1ull << (qdug >> 2);
                    if (sel4(RB, kd) & bit) {
                        const uint32_t xd = qdug % W, yd = qdug / W;
                        const bool onb = xd == 0 || yd == 0 || xd == W - 1 || yd == H - 1;
                        if (onb || ring_runs<W, H>(c, qdug, lane) >= 2) {
                            recompute = true;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) RB[j] &= ~(kd == j ? bit : 0ull);
                            rb_dirty = true;
                        }
                    }
                }
                if (recompute) {
                    static_for<4>([&](auto J) { constexpr int j = decltype(J)::value; RB[j] = CM<W, H, j>::BORDER & PASS[j]; });
                    flood<W, H>(RB, PASS);
                    s.fl |= WF_F_REACH_VALID;
                    rb_dirty = true;
                }
                rb_updated = true;
                if (!((NEWFIRE[0] & RB[0]) | (NEWFIRE[1] & RB[1]) | (NEWFIRE[2] & RB[2]) | (NEWFIRE[3] & RB[3]))) {
                    s.fl |= WF_F_CONTAINED;                                        // :361-367
                    reward = a.p.contained_bonus;
                    paid = true;
                }
            }
            if (dug && !rb_updated) s.fl &= ~(uint32_t)WF_F_REACH_VALID;
            if (!paid) {
                if (!(s.fl & WF_F_ALIVE)) {
                    reward = a.p.death_penalty;                                    // :380-381
                } else if (!anyfire) {                                             // :384-387
                    const int healthy = __popcll(GRASS[0]) + __popcll(GRASS[1]) + __popcll(GRASS[2]) + __popcll(GRASS[3]);
                    reward = a.p.contained_bonus * ((double)healthy / (double)CELLS);
                } else {
                    reward = a.p.default_reward;                                   // :390
                }
            }
            done = !(s.fl & WF_F_RUNNING);                                         // forest_fire.py:48
            if (lane == 0) {
                if (a.reward) a.reward[oi] = reward;
                if (a.done) a.done[oi] = done ? 1 : 0;
            }
        }

        if (dirty) {
            const uint4 out = pack<W, H>(c, s, lane);
            if (active) *gp = out;
            if (lane == 0) {
                *reinterpret_cast<uint4 *>(&a.envs[e]) = make_uint4(s.ax | (s.ay << 16), s.fl, s.episode, s.tick);
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
        Cells c;
        reset_regs<W, H>(a, a.reset_seed, e, lane, c, s);
        const uint4 out = pack<W, H>(c, s, lane);
        if (lane < LANES) reinterpret_cast<uint4 *>(a.planes + (size_t)e * CELLS)[lane] = out;
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
