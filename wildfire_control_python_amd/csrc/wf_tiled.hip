// wf_tiled.hip - gfx950 kernels for grids that do not fit one wavefront (e.g. 256x256), and the generic family.
//
// Same rule as wf_kernels.hip (synchronous integer restatement of
// Simulation/forest_fire.py:85-106 + Simulation/environment.py:278-326).  One step of the TILED family
// (W % 4 == 0, W <= 256, at least 1 024 cells) is two launches:
//   wf_tiled_prologue   one workgroup per environment: Agent.move / toggle / death (environment.py:116-171) with
//                       the dug cell and the agent_pos bytes written straight into the grid; the border-reachability
//                       flood when a dig may have cut a region off (run fill on linear row bitmaps, relax_row_bitmaps);
//                       the still-burning masks of the rows between the workgroups of the tick (it runs in place)
//   wf_step_tiled       the CA tick in place (+ the observation, when asked for; + the fresh rows of environments that
//                       are being auto-reset); per-env reductions by atomics, and the last of an environment's
//                       workgroups to arrive does World.RUNNING, reward, done, record
// (wf_reset: wf_big_reset_rec -> wf_big_reset_cells)
// and of the GENERIC family (any shape, wind with a direction, small grids; ping-pong planes):
//   wf_big_pre -> wf_step_env      a wave (up to 1 024 cells) or a workgroup per environment: reset / flood / tick / reward
//   (few huge grids, WF_CELLS_LEGACY=1: wf_big_pre -> [wf_big_reset_cells] -> [wf_cells_flood] -> wf_step_cells -> wf_big_post)
//
// wf_step_tiled: a workgroup = up to 4 waves = vertically adjacent strips of TR rows of one
// environment (grids narrower than 128 cells: 2 or 4 strips side by side per wave, struct Pack).
// A wave sweeps its strip top to bottom; lane l owns cells x = 4l..4l+3 of
// the current row (one coalesced dwordx4 per lane, a 256-wide row = one 1 KiB wave load).
// "Still burning" becomes four 64-bit lane masks per row (wavefront ballot).  The vertical
// neighbours of a cell are THE SAME BIT of the masks of the rows above and below, the
// horizontal ones are the adjacent component / a one-lane shift, so a row needs the masks
// of three rows and no per-cell index arithmetic.  The masks of each strip's first and
// last row are staged in LDS so that the strips of a workgroup hand their halo to the
// neighbour strip without touching memory again; only the workgroup's outer halo rows
// are read from global memory.
#include "wf_device.h"

namespace {

using namespace wfdev;

enum { MODE_COPY = 0, MODE_STEP = 1, MODE_RESET = 2, MODE_MASK = 3,
       MODE_PRECHECK = 4,   // containment test may run this step (environment.py:345-346)
       MODE_FLOOD = 8 };    // the reachability board must be rebuilt before the tick
constexpr int TR = 16;   // rows per wave strip

// World.reset(), record part (environment.py:186-212): spawn point from the counter-based stream
__device__ __forceinline__ void reset_record(const WfArgs &a, uint32_t e, u64 seed)
{
    const WfCold *cold = a.cold;
    uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);
    const EnvKey key(seed, cold->p.env_id0 + a.env_base + (u64)e, ev.z);
    const uint32_t r = below(key.draw(0), 3u);
    const uint32_t k = below(key.draw(1), (uint32_t)cold->spawn.n[r]);
    const uint32_t sx = (uint32_t)a.W / 2u + (int)cold->spawn.dx[r][k], sy = (uint32_t)a.H / 2u + (int)cold->spawn.dy[r][k];
    ev.x = sx | (sy << 16);
    ev.y = WF_F_ALIVE | WF_F_VISIBLE | WF_F_DIGGING | WF_F_RUNNING;
    if (a.hot_flags & WF_HOT_WIND) {                                        // World.reset :187-193
        uint32_t code = (uint32_t)cold->p.wind_code;
        if (cold->p.wind_mode == WF_WIND_RANDOM)                            // :189-190, three uniform draws
            code = WF_WIND_CODE((int)below(key.draw(3), 3u) - 1, (int)below(key.draw(4), 3u) - 1, below(key.draw(2), 3u));
        ev.y |= code << 16;
    }
    ev.z += 1;
    ev.w = 0;
    *reinterpret_cast<uint4 *>(&a.envs[e]) = ev;
}

// One cell of a fresh grid (reset_map environment.py:59-95, set_fire_to :233-246, Agent.__init__ :108-113) under the
// counter-based stream; river_col = column of the river's water cell in this row (0xffff: none).
struct ResetLaw { uint32_t W, H, cells, qc, qs, span, ppm; int32_t fuel_lo; };

__device__ __forceinline__ ResetLaw reset_law(const WfArgs &a, const uint4 &ev)
{
    const WfCold *cold = a.cold;
    const uint32_t W = (uint32_t)a.W, H = (uint32_t)a.H;
    ResetLaw L;
    L.W = W; L.H = H; L.cells = W * H;
    L.qc = (H / 2u) * W + W / 2u;
    L.qs = (ev.x >> 16) * W + (ev.x & 0xffffu);
    L.fuel_lo = cold->p.fuel_lo;
    L.span = (uint32_t)(cold->p.fuel_hi - cold->p.fuel_lo + 1);
    L.ppm = cold->p.ignite_ppm;
    return L;
}

__device__ __forceinline__ uint32_t reset_cell(const ResetLaw &L, const EnvKey &key, uint32_t q, uint32_t x, uint32_t y,
                                               uint32_t river_col, bool &border_fire)
{
    uint32_t fuel = (uint32_t)L.fuel_lo;
    if (L.span > 1u) fuel += below(key.draw(16u + q), L.span);
    uint32_t ty = WF_GRASS, heat = 0, ag = 0;
    if (river_col == x) ty = WF_WATER;
    if (q == L.qc) { ty = WF_FIRE; heat = WF_HEAT_FORCED; }                 // set_fire_to :233-246
    if (q == L.qs) { ty = WF_DIRT; ag = 1; }                                // Agent.__init__ :108-113
    if (L.ppm && ty == WF_GRASS && below(key.draw(16u + L.cells + q), 1000000u) < L.ppm) {
        ty = WF_FIRE;
        heat = WF_HEAT_FORCED;
        border_fire |= (x == 0 || y == 0 || x == L.W - 1 || y == L.H - 1);
    }
    return ty | ((fuel & 0xffu) << 8) | (heat << 16) | (ag << 24);
}

// ---------------------------------------------------------------------------- pre
// Agent.move / toggle / death and the containment bookkeeping of ONE environment (run by one lane): applies the touched
// bytes to the grid, returns the step mode; the updated record is left in `ev` for the caller to write.
__device__ __forceinline__ uint32_t big_pre_env(const WfArgs &a, const uint32_t e, uint4 &ev, bool &have_record)
{
    const uint32_t W = (uint32_t)a.W, H = (uint32_t)a.H;
    ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);
    have_record = false;
    uint32_t ax = ev.x & 0xffffu, ay = ev.x >> 16, fl = ev.y & 0xffffu;
    const uint32_t wind = ev.y & 0xffff0000u;
    if (!(fl & WF_F_RUNNING)) {
        const bool rs = (a.hot_flags & WF_HOT_AUTO_RESET) != 0;
        if (a.reward) a.reward[e] = 0.0;
        if (a.done) a.done[e] = rs ? 0 : 1;
        if (rs) reset_record(a, e, a.cold->p.seed);   // the cells follow in wf_big_reset_cells
        return rs ? MODE_RESET : MODE_COPY;           // frozen: the reference is never stepped after done
    }
    const uint32_t act = a.actions ? (uint32_t)(uint8_t)a.actions[e] : 0xffu;
    uint8_t *pe = reinterpret_cast<uint8_t *>(a.planes) + (size_t)e * H * W * 4;
    // ---- Agent.move (environment.py:141-155)
    const uint32_t ent = MOVE_LUT >> (min(act, 4u) * 6u);
    const uint32_t ism = (ent >> 4) & 1u;
    const uint32_t nx = ax + (ent & 3u) - 1u, ny = ay + ((ent >> 2) & 3u) - 1u;      // :163-171
    const uint32_t out = ((W - 1u - nx) | nx | (H - 1u - ny) | ny) >> 31;
    const uint32_t inb = ism & (out ^ 1u);                                            // :146 inbounds
    const uint32_t qa = ay * W + ax;
    uint32_t qd = inb ? ny * W + nx : qa;
    const uint32_t td = pe[(size_t)qd * 4], tcur = pe[(size_t)qa * 4];
    const uint32_t moved = inb & (td != WF_WATER ? 1u : 0u);                          // :146 traversable
    const uint32_t dfire = moved & (td == WF_FIRE ? 1u : 0u);
    uint32_t dug = moved & ((fl >> 2) & 1u) & (DIG_SET >> td) & 1u;                   // :151-152, :125-133
    uint32_t dugtype = WF_DIRT;
    const uint32_t vis_before = (fl >> 1) & 1u;
    fl = (fl & ~(ism << 1)) | (moved << 1) | (dfire << 3);                            // :142, :148-149, :154-155
    if (moved) { ax = nx; ay = ny; }
    uint32_t ta = moved ? td : tcur;
    if ((a.hot_flags & WF_HOT_DIG_TOGGLE) && act == 4u) {                              // forest_fire.py:35-36
        fl ^= WF_F_DIGGING;
        // dig() under the agent: a BURNING cell too, which keeps burning under its new type (WF_DIRT_BURNING)
        dug = ((fl >> 2) & 1u) & ((DIG_SET | 2u) >> tcur) & 1u;
        if (tcur == WF_FIRE) dugtype = WF_DIRT_BURNING;
        qd = qa;
    }
    if (dug) ta = WF_DIRT;
    if (a.update_mask & 1ull) {                                                        // forest_fire.py:87
        const uint32_t dies = (fl & 1u) & (((fl >> 3) & 1u) | (ta == WF_FIRE ? 1u : 0u));
        if (dies) fl &= ~(uint32_t)(WF_F_ALIVE | WF_F_VISIBLE);
    }
    // containment bookkeeping (same exact rule as wf_step_group's phase A): can this dig change
    // which cells reach the border?  If so the board is rebuilt (wf_tiled_prologue / wf_cells_flood) before the tick.
    uint32_t m = MODE_STEP;
    if ((a.hot_flags & WF_HOT_CONTAINMENT) && !(fl & (WF_F_BORDER | WF_F_CONTAINED))) {
        m |= MODE_PRECHECK;
        bool flood = !(fl & WF_F_REACH_VALID);
        if (dug && !flood) {
            const uint32_t xd = qd % W, yd = qd / W;
            uint32_t ring = 0, bn = 0;
            bool zn = false;
            const int DX[8] = { 0, 1, 1, 1, 0, -1, -1, -1 }, DY[8] = { -1, -1, 0, 1, 1, 1, 0, -1 };
            for (int r = 0; r < 8; ++r) {
                const uint32_t rx = xd + (uint32_t)DX[r], ry = yd + (uint32_t)DY[r];
                if (rx < W && ry < H) {
                    const uint32_t ty = pe[(size_t)(ry * W + rx) * 4];
                    if (ty < WF_DIRT) ring |= 1u << r;
                    if (ty == WF_DIRT_BURNING && !(r & 1)) zn = true;      // burning dirt hangs on the board through its neighbours
                    if (!(r & 1) && (rx == 0 || ry == 0 || rx == W - 1 || ry == H - 1)) bn |= 1u << r;
                }
            }
            const uint32_t even = ring & 0x55u;
            const uint32_t eff = ring & 0xAAu & (even << 1) & ((even >> 1) | (even << 7));
            const bool onb = xd == 0 || yd == 0 || xd == W - 1 || yd == H - 1;
            flood = (__popc(even) - __popc(eff) >= 2) || (onb && even && !(ring & bn)) || zn;
        }
        if (flood) m |= MODE_FLOOD;
    }
    if (vis_before) pe[(size_t)qa * 4 + 3] = 0;
    if (dug) pe[(size_t)qd * 4] = (uint8_t)dugtype;
    if (fl & WF_F_VISIBLE) pe[(size_t)(ay * W + ax) * 4 + 3] = 1;
    ev.x = ax | (ay << 16);
    ev.y = fl | wind;
    have_record = true;                               // the caller writes it (after a flood, with its valid flag)
    return m;
}

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

// ---------------------------------------------------------------------------- post
// World.RUNNING, reward, done and the record of ONE stepped environment from its reduction words
// (A: bit 0 any fire left, bit 1 fire at the border, bit 2 fire on a border-connected cell; healthy: grass cells).
__device__ __forceinline__ void big_post_env(const WfArgs &a, const uint32_t e, const uint32_t m, const uint32_t A, const uint32_t healthy,
                                             const uint32_t extra_flags, uint4 ev);

__device__ __forceinline__ void big_post_env(const WfArgs &a, const uint32_t e, const uint32_t m, const uint32_t A, const uint32_t healthy)
{
    big_post_env(a, e, m, A, healthy, 0u, *reinterpret_cast<const uint4 *>(&a.envs[e]));
}

// (ev: the environment's record as it is in memory, loaded by the caller)
__device__ __forceinline__ void big_post_env(const WfArgs &a, const uint32_t e, const uint32_t m, const uint32_t A, const uint32_t healthy,
                                             const uint32_t extra_flags, uint4 ev)
{
    uint32_t fl = (ev.y & 0xffffu) | extra_flags;
    const uint32_t anyfire = A & 1u, bhit = (A >> 1) & 1u, reach_hit = (A >> 2) & 1u;
    const uint32_t upd = (a.update_mask & 1ull) ? 1u : 0u;
    fl |= bhit << 5;                                                                   // set_fire_to :245-246
    if (upd && (!(fl & WF_F_ALIVE) || !anyfire)) fl &= ~(uint32_t)WF_F_RUNNING;        // forest_fire.py:105-106
    ev.w += upd;
    double reward = a.default_reward;                                                  // environment.py:390
    if (!anyfire) reward = a.contained_bonus * ((double)healthy / (double)((uint32_t)a.H * (uint32_t)a.W));   // :384-387
    if (!(fl & WF_F_ALIVE)) reward = a.death_penalty;                                  // :380-381
    // containment (environment.py:345-377): no burning cell is connected to a passable border cell
    if ((m & MODE_PRECHECK) && anyfire && !bhit && !reach_hit) {
        fl |= WF_F_CONTAINED;                                                          // :361-367
        reward = a.contained_bonus;
    }
    ev.y = (ev.y & 0xffff0000u) | fl;
    *reinterpret_cast<uint4 *>(&a.envs[e]) = ev;
    if (a.reward) a.reward[e] = reward;
    if (a.done) a.done[e] = (fl & WF_F_RUNNING) ? 0 : 1;                               // forest_fire.py:48
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

// ---------------------------------------------------------------------------- tiled step
struct RowMasks { u64 m[4]; };

__device__ __forceinline__ void still_masks(const uint4 &r, u64 (&S)[4])
{
    const uint32_t c[4] = { r.x, r.y, r.z, r.w };
#pragma unroll
    for (int j = 0; j < 4; ++j)   // reduce_fuel (environment.py:297-307): fire cell with fuel-1 > 0
        S[j] = __ballot((c[j] & 3u) == 1u) & __ballot((int32_t)(int8_t)(c[j] >> 8) > 1);   // burning = fire or burning dirt
}

// The step runs IN PLACE (planes_out == planes).  The masks of the two rows just outside the workgroup - which
// another workgroup may already have rewritten - come from `outer` (wf_tiled_prologue, written before this launch);
// environments that are not stepped are not touched at all; a lane stores its four cells only when one of them
// changed (most cells do not change in a tick).
//
// The sweep keeps three rows in flight (cells C[3], still-burning masks M[3]) and is unrolled three times, so
// that "previous / current / next row" is a renaming of registers instead of 16 moves per row; the per-launch
// switches (update or not, containment test or not) are template parameters of the sweep.
struct TiledAcc { uint32_t anyfire, bhit, healthy, reach_hit; u64 ign_x0, ign_xl; };

// Destination of the in-place stores of one environment: a buffer descriptor over its cells + this lane's byte offset.
// The stores can carry cache-policy bits (WF_TILED_STORE_AUX 16 = sc1: written through the XCD's L2 instead of left
// dirty for the write-back at the kernel boundary, MI355X_MICROARCH.md row "boundary").  Measured at 1 024 x 256x256:
// 102.8 vs 105.7 us per step with two launch sequences per step, 127.3 vs 126.4 with one: within noise, so plain stores.
#ifndef WF_TILED_STORE_AUX
#define WF_TILED_STORE_AUX 0
#endif
struct TiledDst { __amdgpu_buffer_rsrc_t rsrc; uint32_t lane_off; };
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));

// Fused observation (World.get_state, environment.py:399-402) of the tiled family.  obs[b][x][y][k] is transposed
// against the planes' [y][x] and has 3 bytes per cell, so a workgroup stages the byte triples (agent_pos, type == fire,
// fire_mobility != inf) of its 64 rows row-major in LDS - 12 contiguous bytes per lane and row, conflict-free dword
// writes - and, once its four strips are done, thread x walks down column x: four rows' triples are four unaligned LDS
// dword reads, packed into the 12 contiguous bytes obs[x][4g..4g+3][0..2].  Needs H % 4 == 0 (4-byte aligned runs).
struct ObsTile { uint8_t *base; uint32_t W3, row0; };        // LDS tile [64][W*3] (+4 bytes), first row of the workgroup

__device__ __forceinline__ void stage_obs_row(const ObsTile &ot, uint32_t r, uint32_t lane, const uint32_t (&c)[4])
{
    uint32_t tri[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t t = c[j] & 0xffu;
        tri[j] = (c[j] >> 24) | (t == (uint32_t)WF_FIRE ? 0x100u : 0u) | (t < (uint32_t)WF_DIRT ? 0x10000u : 0u);
    }
    uint32_t *p = reinterpret_cast<uint32_t *>(ot.base + (r - ot.row0) * ot.W3 + lane * 12u);
    p[0] = tri[0] | (tri[1] << 24);
    p[1] = (tri[1] >> 8) | (tri[2] << 16);
    p[2] = (tri[2] >> 16) | (tri[3] << 8);
}

// rows [row0, row0 + nrows) of the tile -> obs_e (this environment's (W, H, 3) block); nrows % 4 == 0
__device__ __forceinline__ void flush_obs_tile(const ObsTile &ot, uint32_t nrows, uint32_t W, uint32_t H, int8_t *obs_e)
{
    struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };
    for (uint32_t x = threadIdx.x; x < W; x += blockDim.x)
    for (uint32_t g = 0; g < nrows; g += 4u) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __builtin_memcpy(&v[k], ot.base + (g + (uint32_t)k) * ot.W3 + x * 3u, 4);      // unaligned LDS dword read
            v[k] &= 0xffffffu;
        }
        *reinterpret_cast<U3 *>(obs_e + ((size_t)x * H + ot.row0 + g) * 3u) =
            U3{ v[0] | (v[1] << 24), (v[1] >> 8) | (v[2] << 16), (v[2] >> 16) | (v[3] << 8) };
    }
}

// Packing (P > 1): a grid narrower than 128 cells leaves most of a wave's lanes without a quad, so the wave takes
// P = 2 (W <= 128) or 4 (W <= 64) of its 16-row strips side by side: lane = (segment, quad), segment s sweeping rows
// r0 + 16 s .. + 15 while segment 0 sweeps r0 .. r0 + 15.  Every mask then holds P rows' bits, 64/P lanes apart: the
// vertical neighbours are still "the same bit of the adjacent step's masks" inside a segment; between segments the
// last row of segment s-1 / the first row of segment s+1 come in by a shift of 64/P bits (tiled_sweep's Sabove / Sbelow);
// the one-lane shifts of the horizontal neighbours must not cross a segment border.
template <int P>
struct Pack {
    static constexpr uint32_t LWP = 64u / (uint32_t)P;                   // lanes per segment
    static constexpr u64 SEG0 = P == 1 ? ~0ull : ((1ull << (LWP % 64u)) - 1ull);
    static constexpr u64 X0 = P == 1 ? 1ull : P == 2 ? 0x0000000100000001ull : 0x0001000100010001ull;   // first lane of every segment
    static constexpr u64 XE = X0 << (LWP - 1u);                          // last lane of every segment
    uint32_t row_off;        // this lane's segment * 16: its row at sweep position r is r + row_off
    uint32_t ql;             // this lane's quad within the row
    bool col_ok;             // the quad exists (ql < W / 4)
    u64 XL;                  // lane of the quad holding x == W-1, in every segment
};

template <bool STEP, bool REACH, bool OBS, int P>
__device__ __forceinline__ void tiled_row(uint32_t r, uint32_t H, uint32_t LW, uint32_t ign, const Pack<P> &pk, const uint4 &cur,
                                          const u64 (&Sp)[4], const u64 (&Sc)[4], const u64 (&Sn)[4], const u64 *reach_e,
                                          const TiledDst &dst, const ObsTile &ot, TiledAcc &A)
{
    const uint32_t row = r + pk.row_off;
    const bool active = pk.col_ok && row < H;
    uint32_t c[4] = { cur.x, cur.y, cur.z, cur.w };
    u64 FIRE[4], GR[4];
    if constexpr (STEP) {
        u64 anyign = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 F = __ballot((c[j] & 3u) == 1u);
            // fuel -= 1 and fire -> burnt (low half + 0xFF01) unless still burning (+ 0xFF00); the halves are added
            // independently, so a burning cell uploaded with fuel 0 borrows nothing from its heat byte
            c[j] = pk_add_u16(c[j], lanebit(F) ? (lanebit(Sc[j]) ? 0xFF00u : 0xFF01u) : 0u);
            const u64 G = __ballot((c[j] & 0xffu) == (uint32_t)WF_GRASS);   // idle lanes hold inert (dirt) cells
            // get_neighbours radius 1 (:311-326): above/below = same bit of the adjacent rows; every
            // neighbour set is counted by one add-with-carry-in
            const u64 LF = j > 0 ? Sc[(j + 3) & 3] : ((Sc[3] << 1) & (P == 1 ? ~0ull : ~Pack<P>::X0));
            const u64 RT = j < 3 ? Sc[(j + 1) & 3] : ((Sc[0] >> 1) & (P == 1 ? ~0ull : ~Pack<P>::XE));
            const uint32_t cnt = addc0(addc0(addc0(lanebit(Sp[j]) ? 1u : 0u, Sn[j]), LF), RT);
            const uint32_t room = ign - ((c[j] >> 16) & 0xffu);
            const u64 IGN = G & __ballot(cnt >= room);                     // apply_heat_from_to :278-294
            const uint32_t cn = addc(c[j], min(cnt, room) << 16, IGN);     // heat saturates; grass -> fire (+1)
            c[j] = lanebit(G) ? cn : c[j];
            anyign |= IGN;
            if (j == 0) A.ign_x0 |= IGN;                                    // cells with x == 0 (lane 0)
            if (j == 3) A.ign_xl |= IGN;                                    // cells with x == W-1 (lane LW-1)
            FIRE[j] = Sc[j] | IGN;
            GR[j] = G & ~IGN;
        }
        // set_fire_to :245-246: first / last row of the grid are border rows
        if constexpr (P == 1) {
            if ((r == 0 || r == H - 1) && anyign) A.bhit = 1;
        } else {
            if (anyign & __ballot(row == 0 || row == H - 1)) A.bhit = 1;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            FIRE[j] = __ballot((c[j] & 3u) == 1u);
            GR[j] = __ballot((c[j] & 0xffu) == (uint32_t)WF_GRASS);
        }
    }
    if constexpr (REACH) {   // does a burning cell sit on a border-connected cell?
        typedef u64 u64x4 __attribute__((ext_vector_type(4)));
        typedef const u64x4 __attribute__((address_space(4))) *const_u64x4_ptr;
        u64x4 rb = *(const_u64x4_ptr)(uintptr_t)(reach_e + (size_t)r * 4);
        if constexpr (P > 1) {      // the board rows of the other segments, moved to their lanes (rows past the grid: no fire bit meets them)
#pragma unroll
            for (int sgm = 1; sgm < P; ++sgm) {
                const u64x4 rs = *(const_u64x4_ptr)(uintptr_t)(reach_e + (size_t)min(r + (uint32_t)sgm * (uint32_t)TR, H - 1u) * 4);
                rb |= rs << (u64)(sgm * (int)Pack<P>::LWP);
            }
        }
        A.reach_hit |= ((FIRE[0] & rb.x) | (FIRE[1] & rb.y) | (FIRE[2] & rb.z) | (FIRE[3] & rb.w)) ? 1u : 0u;
    }
    A.anyfire |= (FIRE[0] | FIRE[1] | FIRE[2] | FIRE[3]) ? 1u : 0u;
    A.healthy += (uint32_t)(__popcll(GR[0]) + __popcll(GR[1]) + __popcll(GR[2]) + __popcll(GR[3]));
    if constexpr (OBS) {
        if (active) stage_obs_row(ot, row, pk.ql, c);
    }
    if constexpr (STEP) {
        if (active && ((c[0] ^ cur.x) | (c[1] ^ cur.y) | (c[2] ^ cur.z) | (c[3] ^ cur.w)))
            __builtin_amdgcn_raw_buffer_store_b128(v4u32{ c[0], c[1], c[2], c[3] }, dst.rsrc, (int)(dst.lane_off + r * LW * 16u), 0, WF_TILED_STORE_AUX);
    }
}

template <bool STEP, bool REACH, bool OBS, int P>
__device__ __forceinline__ void tiled_sweep(uint32_t r0, uint32_t r1, uint32_t H, uint32_t LW, uint32_t ign, const Pack<P> &pk,
                                            const uint4 *src, const TiledDst &dst, const ObsTile &ot, const uint4 &first, const u64 (&Sf)[4],
                                            const u64 (&Sabove)[4], const u64 (&Sbelow)[4], const u64 *reach_e, TiledAcc &A)
{
    const uint4 inert = make_uint4(WF_DIRT, WF_DIRT, WF_DIRT, WF_DIRT);
    uint4 C[3] = { first, inert, inert };
    u64 M[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { M[0][j] = Sf[j]; M[2][j] = Sabove[j]; M[1][j] = 0; }
    if (r0 + 1 < r1 && pk.col_ok && r0 + 1u + pk.row_off < H) C[1] = src[(r0 + 1u) * LW];
    for (uint32_t r = r0; r < r1; r += 3u) {
        static_for<3>([&](auto K) {
            constexpr int k = decltype(K)::value, kn = (k + 1) % 3, kp = (k + 2) % 3;
            const uint32_t rr = r + (uint32_t)k;
            if (rr < r1) {
                // the registers of row rr-1 are free: request row rr+2 into them
                C[kp] = inert;
                if (rr + 2 < r1 && pk.col_ok && rr + 2u + pk.row_off < H) C[kp] = src[(rr + 2u) * LW];
                u64 Sn[4];
                if (rr + 1 < r1) still_masks(C[kn], Sn);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) Sn[j] = Sbelow[j];
                }
                tiled_row<STEP, REACH, OBS, P>(rr, H, LW, ign, pk, C[k], M[kp], M[k], Sn, reach_e, dst, ot, A);
#pragma unroll
                for (int j = 0; j < 4; ++j) M[kn][j] = Sn[j];
            }
        });
    }
}

// Workgroup = nw waves (blockDim.x / 64: 4, fewer for grids that need fewer), wave = P strips of 16 rows side by side.
template <bool OBS, int P>
__global__ __launch_bounds__(256) void wf_step_tiled(const WfArgs a, const uint8_t *mode, uint32_t *acc, const u64 *outer)
{
    __shared__ u64 halo[4][2][4];                    // [wave][first,last row of each of its segments][component]
    __shared__ uint16_t river_tab[4][64];            // auto-reset with rivers: water column of each of a wave's rows
    extern __shared__ __attribute__((aligned(16))) uint8_t obs_lds[];     // OBS: [rows of the workgroup][W*3] (+4) staging tile
    constexpr uint32_t LWP = Pack<P>::LWP, WR = (uint32_t)P * (uint32_t)TR;          // lanes per segment, rows per wave
    const uint32_t lane = __lane_id();
    const uint32_t wv = rfl(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint32_t H = (uint32_t)a.H, W = (uint32_t)a.W, LW = W >> 2;
    const uint32_t gx = (H + nw * WR - 1u) / (nw * WR);          // workgroups per environment
    const uint32_t e = blockIdx.x / gx;
    const uint32_t r0 = ((blockIdx.x % gx) * nw + wv) * WR;      // segment 0's first row; segment s: r0 + 16 s
    const uint32_t r1 = min(r0 + TR, H);                         // end of the sweep (segment 0's rows)
    const bool has_rows = r0 < H;
    Pack<P> pk;
    pk.row_off = (P == 1 ? 0u : lane / LWP) * (uint32_t)TR;
    pk.ql = P == 1 ? lane : lane % LWP;
    pk.col_ok = pk.ql < LW;
    pk.XL = Pack<P>::X0 << (LW - 1u);
    const uint32_t lane_cell0 = pk.row_off * LW + pk.ql;         // this lane's quad index at sweep position 0 (+ r * LW)
    const uint32_t ign = (uint32_t)a.ignite_count;
    const uint32_t md = mode[e];
    const uint32_t wg_row0 = (blockIdx.x % gx) * nw * WR;
    const ObsTile ot = { obs_lds, W * 3u, wg_row0 };
    int8_t *obs_e = OBS ? a.obs + (size_t)e * H * W * 3u : nullptr;
    if ((md & MODE_MASK) != MODE_STEP) {                   // frozen / freshly reset environment: nothing to step
        const bool fresh = (md & MODE_MASK) == MODE_RESET;  // auto-reset: World.reset() of this environment's cells, here
        if (fresh || OBS) {                                // (a frozen one still owes its observation)
            uint4 *rows = reinterpret_cast<uint4 *>(a.planes) + (size_t)e * H * LW + lane_cell0;
            const uint4 ev = *reinterpret_cast<const uint4 *>(&a.envs[e]);          // fresh: written by the prologue
            const EnvKey key(a.cold->p.seed, a.cold->p.env_id0 + a.env_base + (u64)e, ev.z - 1u);
            const ResetLaw L = reset_law(a, ev);
            const bool rivers = fresh && a.cold->p.make_rivers != 0;
            bool border_fire = false;
            // reset_map's river: the wave walks it once and keeps the columns of its own rows (0xffff: none)
            uint16_t *river_col = river_tab[wv];
            if (rivers) {
                river_col[lane] = 0xffffu;
                river_cells(key, (int)W, (int)H, [&](int x, int y) {
                    if ((uint32_t)y - r0 < WR && lane == 0) river_col[(uint32_t)y - r0] = (uint16_t)x;
                });
            }
            for (uint32_t r = r0; r < r1; ++r) {
                const uint32_t row = r + pk.row_off;
                const bool active = pk.col_ok && row < H;
                uint32_t cq[4] = { 0, 0, 0, 0 };
                if (fresh) {
                    if (active) {
                        const uint32_t rcol = rivers ? (uint32_t)river_col[row - r0] : 0xffffu;
#pragma unroll
                        for (int j = 0; j < 4; ++j) cq[j] = reset_cell(L, key, row * W + 4u * pk.ql + (uint32_t)j, 4u * pk.ql + (uint32_t)j, row, rcol, border_fire);
                        rows[r * LW] = make_uint4(cq[0], cq[1], cq[2], cq[3]);
                    }
                } else if (active) {
                    const uint4 q = rows[r * LW];
                    cq[0] = q.x; cq[1] = q.y; cq[2] = q.z; cq[3] = q.w;
                }
                if constexpr (OBS) {
                    if (active) stage_obs_row(ot, row, pk.ql, cq);
                }
            }
            if (__ballot(border_fire) && lane == 0)
                atomicOr(reinterpret_cast<uint32_t *>(&a.envs[e]) + 1, (uint32_t)WF_F_BORDER);
        }
        if constexpr (OBS) {
            __syncthreads();
            flush_obs_tile(ot, min(nw * WR, H - wg_row0), W, H, obs_e);
        }
        return;
    }
    const bool step = (a.update_mask & 1ull) != 0;
    const bool reach_test = (md & MODE_PRECHECK) != 0;
    const u64 *reach_e = a.reach + (size_t)e * H * 4;      // board rows: u64[H][4], bit = quad of the row
    // per-env base pointers (64-bit, once) + 32-bit uint4 indices inside the env (H*W/4 < 2^32)
    const uint4 *src = reinterpret_cast<const uint4 *>(a.planes) + (size_t)e * H * LW + lane_cell0;
    const TiledDst dst = { __builtin_amdgcn_make_buffer_rsrc(a.planes + (size_t)e * H * W, 0, (int)(H * W * 4u), 0x00020000), lane_cell0 * 16u };
    const uint4 inert = make_uint4(WF_DIRT, WF_DIRT, WF_DIRT, WF_DIRT);

    // ---- prologue: masks of every segment's first and last row (original state) -> LDS
    uint4 first = inert, last = inert;
    u64 Sf[4] = { 0, 0, 0, 0 }, Sl[4] = { 0, 0, 0, 0 };
    if (has_rows) {
        if (pk.col_ok && r0 + pk.row_off < H) first = src[r0 * LW];
        if (pk.col_ok && r1 - 1u + pk.row_off < H) last = src[(r1 - 1u) * LW];
        still_masks(first, Sf);
        still_masks(last, Sl);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { halo[wv][0][j] = Sf[j]; halo[wv][1][j] = Sl[j]; }
    }
    __syncthreads();
    TiledAcc A = { 0, 0, 0, 0, 0, 0 };
    if (has_rows) {                                        // (a wave without rows of its own still flushes and arrives below)

    // masks of the row above / below every segment (original state).  Segment 0's row above: the last segment of the
    // wave before (LDS) or of the workgroup before (the saved outer halos); the last segment's row below likewise; the
    // segments in between see each other's last / first rows through a shift of one segment
    u64 Sp[4] = { 0, 0, 0, 0 }, Sb[4] = { 0, 0, 0, 0 };
    if (r0 > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 v = wv > 0 ? (halo[wv - 1][1][j] >> (LWP * (uint32_t)(P - 1))) : outer[(size_t)blockIdx.x * 8 + j];
            Sp[j] = (((u64)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v)) & Pack<P>::SEG0;
        }
    }
    if (r0 + WR < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 v = wv + 1u < nw ? halo[wv + 1][0][j] : outer[(size_t)blockIdx.x * 8 + 4 + j];
            Sb[j] = ((((u64)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v)) & Pack<P>::SEG0) << (LWP * (uint32_t)(P - 1));
        }
    }
    if constexpr (P > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Sp[j] |= Sl[j] << LWP;
            Sb[j] |= Sf[j] >> LWP;
        }
    }

    // ---- sweep
    if (step) {
        if (reach_test) tiled_sweep<true, true, OBS, P>(r0, r1, H, LW, ign, pk, src, dst, ot, first, Sf, Sp, Sb, reach_e, A);
        else tiled_sweep<true, false, OBS, P>(r0, r1, H, LW, ign, pk, src, dst, ot, first, Sf, Sp, Sb, reach_e, A);
    } else {
        if (reach_test) tiled_sweep<false, true, OBS, P>(r0, r1, H, LW, ign, pk, src, dst, ot, first, Sf, Sp, Sb, reach_e, A);
        else tiled_sweep<false, false, OBS, P>(r0, r1, H, LW, ign, pk, src, dst, ot, first, Sf, Sp, Sb, reach_e, A);
    }
    A.bhit |= ((A.ign_x0 & Pack<P>::X0) | (A.ign_xl & pk.XL)) ? 1u : 0u;      // x == 0 / x == W-1 columns
    if (lane == 0) {
        // device-scope atomics are performed at the memory side (the per-XCD L2s are not coherent with each other);
        // the RETURNING forms, consumed, make the wave wait until they have been
        const uint32_t R = A.anyfire | (A.bhit << 1) | (A.reach_hit << 2);
        uint32_t o0 = 0, o1 = 0;
        if (R) o0 = atomicOr(&acc[4 * e], R);
        if (A.healthy) o1 = atomicAdd(&acc[4 * e + 1], A.healthy);
        asm volatile("" :: "v"(o0), "v"(o1));
    }
    }   // has_rows
    __syncthreads();
    if constexpr (OBS) flush_obs_tile(ot, min(nw * WR, H - wg_row0), W, H, obs_e);
    // ---- World.RUNNING / reward / done of the environment: by whichever of its gx workgroups arrives last (every wave
    // of this workgroup has its reduction atomics behind it: the barrier above).  No data but the atomically updated
    // words passes between the workgroups, so no cache write-back fence is needed.
    if (threadIdx.x == 0) {
        const uint32_t ticket = atomicAdd(&acc[4 * e + 2], 1u);
        if (ticket == gx - 1u) {
            const uint32_t Aw = atomicExch(&acc[4 * e], 0u), healthy = atomicExch(&acc[4 * e + 1], 0u);
            atomicExch(&acc[4 * e + 2], 0u);
            big_post_env(a, e, md, Aw, healthy);
        }
    }
}

// bit i of the low 16 bits -> bit 4i; and back
__device__ __forceinline__ u64 spread4(u64 v)
{
    v &= 0xFFFFull;
    v = (v | (v << 24)) & 0x000000FF000000FFull;
    v = (v | (v << 12)) & 0x000F000F000F000Full;
    v = (v | (v << 6)) & 0x0303030303030303ull;
    v = (v | (v << 3)) & 0x1111111111111111ull;
    return v;
}

__device__ __forceinline__ u64 squeeze4(u64 v)
{
    v &= 0x1111111111111111ull;
    v = (v | (v >> 3)) & 0x0303030303030303ull;
    v = (v | (v >> 6)) & 0x000F000F000F000Full;
    v = (v | (v >> 12)) & 0x000000FF000000FFull;
    v = (v | (v >> 24)) & 0xFFFFull;
    return v;
}

// Flood the seeds s (a subset of p) along the runs of ones of the row bitmap p (NW words of 64 bits), towards both ends:
// adding a seed to its run carries through the rest of the run, so (p ^ (p + s)) & p is everything from a seed up to
// the run's end; the same on the bit-reversed row fills downwards.
template <int NW>
__device__ __forceinline__ void fill_runs(const u64 (&p)[NW], u64 (&s)[NW])
{
    u64 up[NW], dn[NW];
    unsigned char carry = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const u64 t = p[w] + s[w], t2 = t + carry;
        const unsigned char c2 = (t < p[w]) | (t2 < t);
        up[w] = ((p[w] ^ t2) & p[w]) | s[w];
        carry = c2;
    }
    carry = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {                      // reversed row: word NW-1 - w, bits reversed
        const u64 pr = __brevll(p[NW - 1 - w]), sr = __brevll(s[NW - 1 - w]);
        const u64 t = pr + sr, t2 = t + carry;
        const unsigned char c2 = (t < pr) | (t2 < t);
        dn[NW - 1 - w] = __brevll(((pr ^ t2) & pr) | sr);
        carry = c2;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) s[w] = up[w] | dn[w];
}

// Relaxation of a border-reachability board kept as linear row bitmaps (NW words per row: pass = passable cells, reach =
// seeds on entry, the flood on return), by `nthreads` threads of which this is thread `t`: one thread per row takes what
// the rows above and below have reached, then fills along the row's passable runs in both directions.  A sweep advances a
// whole horizontal run and one row, so the open field outside a dug ring converges in two or three sweeps; `bound`
// (H*W) sweeps always suffice.  `changed`: an LDS word of the group; `sync`: its barrier.  Returns true when converged.
template <int NW, class Sync>
__device__ __forceinline__ bool relax_row_bitmaps(const u64 *pass, u64 *reach, uint32_t H, uint32_t bound, uint32_t t, uint32_t nthreads,
                                                  uint32_t *changed, Sync sync)
{
    for (uint32_t it = 0; it < bound; ++it) {
        if (t == 0) *changed = 0;
        sync();
        bool ch = false;
        for (uint32_t r = t; r < H; r += nthreads) {
            u64 p4[NW], s4[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                p4[w] = pass[r * NW + w];
                s4[w] = (reach[r * NW + w] | (r > 0 ? reach[(r - 1u) * NW + w] : 0ull) | (r + 1 < H ? reach[(r + 1u) * NW + w] : 0ull)) & p4[w];
            }
            fill_runs<NW>(p4, s4);
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (s4[w] != reach[r * NW + w]) { reach[r * NW + w] = s4[w]; ch = true; }
        }
        if (ch) *changed = 1;
        sync();
        const uint32_t any = *changed;
        sync();
        if (!any) return true;
    }
    return false;
}

// ---------------------------------------------------------------------------- prologue of a tiled step
// One workgroup per environment does everything that has to precede the in-place tick, in ONE launch (it used to be
// three: each kernel boundary of a step costs ~4.5 us that nothing overlaps):
//   1. thread 0: Agent.move / toggle / death + containment bookkeeping (big_pre_env), touched bytes applied to the grid;
//   2. if that flagged it: the border-reachability board of the environment is rebuilt (boards resident in LDS,
//      monotone in-place relaxation until nothing changes): bit l of word [r][j] = cell (row r, x = 4l+j) is
//      4-connected to a passable border cell through passable cells;
//   3. the still-burning masks of the two rows just outside every workgroup of wf_step_tiled, taken from the state
//      BEFORE the tick: halo[(e*gx + g)*8 + 0..3] = row above workgroup g, +4..7 = row below (one wave per row).
__global__ __launch_bounds__(256) void wf_tiled_prologue(const WfArgs a, uint8_t *mode, u64 *halo, uint32_t rows_per_wg)
{
    extern __shared__ u64 lds[];                     // flood: [H*4] pass, [H*4] reach
    __shared__ u64 pm[4][8][4];                      // flood: the interleaved passable masks of a wave's current 8 rows
    __shared__ uint32_t s_mode, changed;
    const uint32_t e = blockIdx.x;
    const uint32_t H = (uint32_t)a.H, W = (uint32_t)a.W, LW = W >> 2, words = H * 4u;
    const uint32_t lane = __lane_id(), wv = threadIdx.x >> 6;
    uint4 ev;
    bool have_record = false;
    if (threadIdx.x == 0) {
        const uint32_t m = big_pre_env(a, e, ev, have_record);
        mode[e] = (uint8_t)m;
        s_mode = m;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the touched bytes, for the other waves of this CU
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t md = s_mode;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.planes) + (size_t)e * H * LW + lane;
    if (md & MODE_FLOOD) {                                           // (only set with the containment test on)
        u64 *pass = lds, *reach = lds + words;
        // passable masks; kept LINEAR for the flood: bit x % 64 of word [r][x / 64] (the ballots give the interleaved
        // layout of the tick - bit l of component j = cell x = 4l + j).  A wave takes 8 consecutive rows per trip and
        // requests them all before it looks at the first: a row per trip made the build one memory latency per row,
        // 32 to 64 of them in a row for a 128- or 256-row grid (40-100 us, with the whole launch waiting)
        for (uint32_t rb = wv * 8u; rb < H; rb += 32u) {
            uint4 cq[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t r = rb + (uint32_t)k;
                cq[k] = make_uint4(WF_DIRT, WF_DIRT, WF_DIRT, WF_DIRT);
                if (lane < LW && r < H) cq[k] = src[r * LW];
            }
            // the four interleaved masks of each of the 8 rows go through LDS, so that lane (k, w) = (row of the trip,
            // linear word) interleaves ONE word - the four 16 -> 64 bit spreads once per trip instead of once per row
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint4 c = cq[k];
                const u64 P[4] = { __ballot((c.x & 0xffu) < WF_DIRT), __ballot((c.y & 0xffu) < WF_DIRT),
                                   __ballot((c.z & 0xffu) < WF_DIRT), __ballot((c.w & 0xffu) < WF_DIRT) };
                if (lane < 4u) pm[wv][k][lane] = lane == 0u ? P[0] : lane == 1u ? P[1] : lane == 2u ? P[2] : P[3];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                const uint32_t k = lane >> 2, w = lane & 3u, r = rb + k;
                if (lane < 32u && r < H) {
                    const u64 lin = spread4(pm[wv][k][0] >> (16u * w)) | (spread4(pm[wv][k][1] >> (16u * w)) << 1) |
                                    (spread4(pm[wv][k][2] >> (16u * w)) << 2) | (spread4(pm[wv][k][3] >> (16u * w)) << 3);
                    pass[r * 4 + w] = lin;
                    // seeds: passable border cells (first / last row; x == 0; x == W-1)
                    const bool edge = r == 0 || r == H - 1;
                    u64 bm = edge ? ~0ull : 0ull;
                    if (w == 0) bm |= 1ull;
                    if (w == (W - 1u) >> 6) bm |= 1ull << ((W - 1u) & 63u);
                    reach[r * 4 + w] = lin & bm;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        const bool converged = relax_row_bitmaps<4>(pass, reach, H, H * W, threadIdx.x, 256u, &changed, [] { __syncthreads(); });
        // back to the tick's interleaved layout, in place: component j of row r = every fourth bit from bit j
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < H; r += 256u) {
            u64 lin[4], il[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int w = 0; w < 4; ++w) lin[w] = reach[r * 4 + w];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int w = 0; w < 4; ++w) il[j] |= squeeze4(lin[w] >> j) << (16 * w);
#pragma unroll
            for (int j = 0; j < 4; ++j) reach[r * 4 + j] = il[j];
        }
        __syncthreads();
        u64 *out = a.reach + (size_t)e * words;
        if (a.hot_flags & WF_HOT_DIG_TOGGLE) {
            // Burning dirt (WF_DIRT_BURNING) is impassable, yet the reference's search starts FROM it: it reaches the
            // border iff one of its 4-neighbours does.  One row per wave iteration: re-read the row, add those cells.
            for (uint32_t r = wv; r < H; r += 4u) {
                uint4 c = make_uint4(WF_DIRT, WF_DIRT, WF_DIRT, WF_DIRT);
                if (lane < LW) c = src[r * LW];
                const u64 Z[4] = { __ballot((c.x & 0xffu) == WF_DIRT_BURNING), __ballot((c.y & 0xffu) == WF_DIRT_BURNING),
                                   __ballot((c.z & 0xffu) == WF_DIRT_BURNING), __ballot((c.w & 0xffu) == WF_DIRT_BURNING) };
                if (lane < 4u) {
                    const uint32_t j = lane, k = r * 4u + j;
                    u64 nb = (r > 0 ? reach[k - 4] : 0ull) | (r + 1 < H ? reach[k + 4] : 0ull);
                    nb |= j > 0 ? reach[k - 1] : (reach[k + 3] << 1);
                    nb |= j < 3 ? reach[k + 1] : (reach[k - 3] >> 1);
                    out[k] = reach[k] | (nb & Z[j]);
                }
            }
        } else {
            for (uint32_t k = threadIdx.x; k < words; k += 256u) out[k] = reach[k];
        }
        // (always true with the H*W bound; a board that did not converge would not be trusted later)
        if (threadIdx.x == 0 && converged) ev.y |= (uint32_t)WF_F_REACH_VALID;
    }
    if (threadIdx.x == 0 && have_record) *reinterpret_cast<uint4 *>(&a.envs[e]) = ev;
    // ---- outer halos
    if ((md & MODE_MASK) != MODE_STEP || !halo) return;
    const uint32_t gx = (H + rows_per_wg - 1u) / rows_per_wg;
    for (uint32_t task = wv; task < gx * 2u; task += 4u) {
        const uint32_t side = task & 1u, g = task >> 1;
        const uint32_t first = g * rows_per_wg, endr = min(first + rows_per_wg, H);
        if (side == 0 ? first == 0 : endr >= H) continue;                  // grid edge: no such row
        const uint32_t r = side == 0 ? first - 1u : endr;
        uint4 c = make_uint4(WF_DIRT, WF_DIRT, WF_DIRT, WF_DIRT);
        if (lane < LW) c = src[r * LW];
        u64 S[4];
        still_masks(c, S);
        if (lane == 0) {
            u64 *o = halo + ((size_t)e * gx + g) * 8 + side * 4u;
            o[0] = S[0]; o[1] = S[1]; o[2] = S[2]; o[3] = S[3];
        }
    }
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
        if ((a.hot_flags & WF_HOT_DIG_TOGGLE) && (se[q] & 0xffu) == (uint32_t)WF_DIRT_BURNING) {
            // burning dirt: impassable, but the reference's search starts from it -> reached iff a 4-neighbour is
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
        c = (c & 0xFFFF0000u) | ((c + 0xFF01u - still(c)) & 0xFFFFu);      // reduce_fuel :297-307 (low half only)
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
                if ((a.hot_flags & WF_HOT_DIG_TOGGLE) && (src(q) & 0xffu) == (uint32_t)WF_DIRT_BURNING)     // reached iff a 4-neighbour is
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
                if ((a.hot_flags & WF_HOT_DIG_TOGGLE) && (src(q) & 0xffu) == (uint32_t)WF_DIRT_BURNING) {
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

// Shape of a wf_step_tiled launch: strips packed side by side per wave, waves per workgroup, workgroups per environment
struct TiledShape { int P, nw, rows_per_wg, gx; };
static TiledShape tiled_shape(int H, int W)
{
    TiledShape t;
    const int LW = W / 4;
    t.P = LW <= 16 ? 4 : LW <= 32 ? 2 : 1;
    const int rows_per_wave = t.P * TR;
    t.nw = (H + rows_per_wave - 1) / rows_per_wave;
    if (t.nw > 4) t.nw = 4;
    t.rows_per_wg = t.nw * rows_per_wave;
    t.gx = (H + t.rows_per_wg - 1) / t.rows_per_wg;
    return t;
}
int wf_tiled_groups(int H, int W) { return tiled_shape(H, W).gx; }

// wf_step_tiled can write the observation itself when the (x, y-run) blocks of obs are 4-byte aligned
bool wf_tiled_fused_obs(int H, int W) { return H % 4 == 0 && W <= 256; }

hipError_t wf_launch_step_big(WfFamily fam, const WfArgs &a, uint8_t *mode, uint32_t *acc, u64 *halo, u64 auto_seed, hipStream_t s, bool per_env)
{
    if (fam == WF_FAM_TILED) {
        // prologue (agent logic, reachability flood, outer halos) -> the in-place tick, which also writes the fresh
        // grids of environments that are being auto-reset and whose last workgroup per environment does
        // World.RUNNING / reward / done
        const TiledShape ts = tiled_shape(a.H, a.W);
        const unsigned gx = (unsigned)ts.gx;
        const size_t flood_lds = (a.hot_flags & WF_HOT_CONTAINMENT) ? (size_t)a.H * 4 * 8 * 2 : 0;
        hipLaunchKernelGGL(wf_tiled_prologue, dim3((unsigned)a.batch), dim3(256), flood_lds, s, a, mode, gx > 1 ? halo : (u64 *)nullptr,
                           (uint32_t)ts.rows_per_wg);
        (void)auto_seed;       // == cold->p.seed: finished environments are reset by the tick kernel itself
        const dim3 grid(gx * (unsigned)a.batch), block(64u * (unsigned)ts.nw);
        const size_t obs_lds = a.obs ? (size_t)ts.rows_per_wg * a.W * 3 + 16 : 0;      // fused World.get_state(): the caller checked wf_tiled_fused_obs(H, W)
#define WF_TILED_LAUNCH(OBS, P) hipLaunchKernelGGL((wf_step_tiled<OBS, P>), grid, block, obs_lds, s, a, mode, acc, (const u64 *)halo)
        if (a.obs) { if (ts.P == 4) WF_TILED_LAUNCH(true, 4); else if (ts.P == 2) WF_TILED_LAUNCH(true, 2); else WF_TILED_LAUNCH(true, 1); }
        else { if (ts.P == 4) WF_TILED_LAUNCH(false, 4); else if (ts.P == 2) WF_TILED_LAUNCH(false, 2); else WF_TILED_LAUNCH(false, 1); }
#undef WF_TILED_LAUNCH
        return hipGetLastError();
    }
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
