// wf_big_common.h - shared by the two big-grid kernel families (wf_tiled.hip, wf_generic.hip): step modes, the reset
// law of one cell, the agent logic and the reward / done logic of one environment (one lane), and the run-fill flood
// on linear row bitmaps.  Included inside each file's anonymous namespace; not a stand-alone header.
#pragma once


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
    // set_fire_to :233-246; under a river the cell keeps its infinite fire mobility (:85)
    if (q == L.qc) { ty = ty == WF_WATER ? (uint32_t)WF_FIRE_BLOCKED : (uint32_t)WF_FIRE; heat = WF_HEAT_FORCED; }
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
    // the eight cells around the one a dig would take (it is `qd` whatever the action turns out to do), for the test
    // below: requested WITH the two cells above, not after them - this runs on one lane with a workgroup waiting for it
    const bool ring_test = (a.hot_flags & WF_HOT_CONTAINMENT) && !(fl & (WF_F_BORDER | WF_F_CONTAINED)) && (fl & WF_F_REACH_VALID);
    const int DX[8] = { 0, 1, 1, 1, 0, -1, -1, -1 }, DY[8] = { -1, -1, 0, 1, 1, 1, 0, -1 };
    uint32_t ring_ty[8];
    const uint32_t xd = inb ? nx : ax, yd = inb ? ny : ay;        // (x, y) of qd
    if (ring_test) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t rx = xd + (uint32_t)DX[r], ry = yd + (uint32_t)DY[r];
            ring_ty[r] = rx < W && ry < H ? (uint32_t)pe[(size_t)(ry * W + rx) * 4] : 0xffu;
        }
    }
    const uint32_t moved = inb & (td != WF_WATER ? 1u : 0u);                          // :146 traversable
    const uint32_t dfire = moved & (FIRE_SET >> td) & 1u;                             // is_burning(): fire, blocked fire
    uint32_t dug = moved & ((fl >> 2) & 1u) & (DIG_SET >> td) & 1u;                   // :151-152, :125-133
    uint32_t dugtype = WF_DIRT;
    const uint32_t vis_before = (fl >> 1) & 1u;
    fl = (fl & ~(ism << 1)) | (moved << 1) | (dfire << 3);                            // :142, :148-149, :154-155
    if (moved) { ax = nx; ay = ny; }
    uint32_t ta = moved ? td : tcur;
    if ((a.hot_flags & WF_HOT_DIG_TOGGLE) && act == 4u) {                              // forest_fire.py:35-36
        fl ^= WF_F_DIGGING;
        // dig() under the agent: a BURNING cell too, which keeps burning under its new type (WF_DIRT_BURNING)
        dug = ((fl >> 2) & 1u) & ((DIG_SET | FIRE_SET) >> tcur) & 1u;
        if ((FIRE_SET >> tcur) & 1u) dugtype = WF_DIRT_BURNING;
        qd = qa;
    }
    if (dug) ta = WF_DIRT;
    if (a.update_mask & 1ull) {                                                        // forest_fire.py:87
        const uint32_t dies = (fl & 1u) & (((fl >> 3) | (FIRE_SET >> ta)) & 1u);
        if (dies) fl &= ~(uint32_t)(WF_F_ALIVE | WF_F_VISIBLE);
    }
    // containment bookkeeping (same exact rule as wf_step_group's phase A): can this dig change
    // which cells reach the border?  If so the board is rebuilt (wf_tiled_prologue / wf_cells_flood) before the tick.
    uint32_t m = MODE_STEP;
    if ((a.hot_flags & WF_HOT_CONTAINMENT) && !(fl & (WF_F_BORDER | WF_F_CONTAINED))) {
        m |= MODE_PRECHECK;
        bool flood = !(fl & WF_F_REACH_VALID);
        if (dug && !flood) {
            uint32_t ring = 0, bn = 0;
            bool zn = false;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t rx = xd + (uint32_t)DX[r], ry = yd + (uint32_t)DY[r];
                if (rx < W && ry < H) {
                    const uint32_t ty = ring_ty[r];
                    if (ty < WF_DIRT) ring |= 1u << r;
                    if (blocked_burning(ty) && !(r & 1)) zn = true;        // burning dirt / blocked fire hangs on the board through its neighbours
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
