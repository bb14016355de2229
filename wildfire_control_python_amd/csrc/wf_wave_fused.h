// wf_wave_fused.h - fused multi-step kernel of the wave family (one environment per wave, grid in registers)
// Included by wf_kernels.hip inside its anonymous namespace (one translation unit: the launchers there
// instantiate these templates); not a stand-alone header.

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

#ifndef WF_LB_ROWS_MULTI
#define WF_LB_ROWS_MULTI 6   // rows kernel with the step loop: a few more live values; 6 waves per SIMD without spills
#endif
#ifndef WF_LB_ROWS
#define WF_LB_ROWS 8   // rows kernel: 56 VGPRs, 8 waves per SIMD (fewer resident waves measured slower)
#endif
#ifndef WF_SKIP_CLEAN
#define WF_SKIP_CLEAN 1   // single-step kernels store only the row pieces / cell quads that changed (-7.5 % time at config 3)
#endif
#ifndef WF_STORE_AUX
#define WF_STORE_AUX 0    // rows kernel: cache-policy bits of the grid stores (0 = plain stores, 16 = sc1 write-through: measured
                          // 20.0 vs 20.5 us per 65 536-env step with two launches per step, but 28.0 vs 25.7 with one, 36.0 vs 30.2 at
                          // 125 000 envs and 318 vs 291 at 1 M - sparse 28-byte pieces written through cost more than the write-back saves)
#endif
#ifndef WF_PROGRESS_PRIO
#define WF_PROGRESS_PRIO 0   // rows kernel: a wave lowers its issue priority as it gets ahead (see set_progress_prio; measured: off is faster)
#endif
#ifndef WF_SHARED_A_PRIO
#define WF_SHARED_A_PRIO 1   // workgroup-shared rows kernel: issue priority of the wave that runs phase A (see wf_step_rows)
#endif
#ifndef WF_ROWS_WAVES_MIN
#define WF_ROWS_WAVES_MIN WF_LB_ROWS   // see wf_step_rows
#endif
#ifndef WF_STORE_WAIT
#define WF_STORE_WAIT 1   // 0: private-phase kernels skip the wait for phase A byte stores (no measurable gain: 22.4-22.7 us either way)
#endif
#ifndef WF_LB
#define WF_LB 6   // minimum waves per SIMD the register allocator must leave room for
#endif
template <int W, int H, bool FUSED, bool RARE>
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
        // (fl: the record's whole second word; its upper half - the wind code of wind contexts, unused here - rides along)
        uint32_t ax = ev.x & 0xffffu, ay = ev.x >> 16, fl = ev.y, episode = ev.z, tick = ev.w;
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
            const uint32_t oi = (uint32_t)st * (uint32_t)a.step_stride + e;       // row st of the (n_steps, context batch) arrays
            uint32_t act_next = 0xffu;
            if (FUSED) {
                if (st + 1 < n_steps && a.actions) act_next = (uint32_t)(uint8_t)a.actions[oi + (uint32_t)a.step_stride];
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
                    reset_regs<W, H, RARE>(cold, cc->p.seed, e_pin + a.env_base, lane, t, u, h, s, a.reach + (size_t)e_pin * 4);   // stream keyed by the context index
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
                constexpr uint32_t ONFIRE = RARE ? FIRE_SET : 2u;                      // is_burning(): type fire (RARE: blocked fire too)
                const uint32_t dfire = moved & (ONFIRE >> td);                         // destination burning
                uint32_t dug = moved & (fl >> 2) & (DIG_SET >> td) & 1u;               // digging, not burning, not dirt yet (:151-152, :125-133)
                uint32_t dugtype = (uint32_t)WF_DIRT;
                fl = (fl & ~(ism << 1)) | (moved << 1) | (dfire << 3);                 // :142, :148-149, :154-155
                ax += moved * (nx - ax);
                ay += moved * (ny - ay);
                uint32_t ta = tcur + moved * (td - tcur);                              // type under the agent
                // ---- Agent.toggle_digging (forest_fire.py:35-36, environment.py:136-138): rare
                if (RARE && (a.hot_flags & WF_HOT_DIG_TOGGLE)) {
                    if (rfl(act == 4u ? 1u : 0u)) {
                        fl ^= WF_F_DIGGING;
                        // dig() under the agent: a BURNING cell too, which stays in burning_cells under its
                        // new type (WF_DIRT_BURNING)
                        dug = (fl >> 2) & ((DIG_SET | ONFIRE) >> tcur) & 1u;
                        dugtype = ((ONFIRE >> tcur) & 1u) ? (uint32_t)WF_DIRT_BURNING : dugtype;
                        qd = qa;
                    }
                }
                ta += dug * ((uint32_t)WF_DIRT - ta);
                {
                    // per-lane selector: component to retype on the owning lane, >= 4 elsewhere
                    const uint32_t sel = (qd & 3u) + ((min((qd >> 2) ^ lane, 1u) | (dug ^ 1u)) << 2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = (sel == (uint32_t)j) ? dugtype : t[j];
                }

                // ---- ForestFire.update (forest_fire.py:85-106)
                const bool do_update = FUSED ? (bool)((a.update_mask >> st) & 1ull) : (bool)(a.update_mask & 1ull);
                if (do_update) {
                    // :87 drop dead agents (Agent.is_dead environment.py:116-120)
                    const uint32_t dies = fl & ((fl >> 3) | (ONFIRE >> ta)) & 1u;      // alive & (dead | on fire)
                    fl &= ~(dies * (uint32_t)(WF_F_ALIVE | WF_F_VISIBLE));

                    u64 STILL[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // reduce_fuel (:297-307): a fire cell whose fuel-1 > 0 keeps burning
                        const uint32_t key = ((t[j] & 3u) << 8) + u[j];               // burning: fire (1), burning dirt (5), blocked fire (13)
                        STILL[j] = __ballot((key - 0x102u) < 0x7Eu);                   // burning && 2<=fuel<=127
                        const uint32_t isfire = ((RARE ? 0x2022u : 0x22u) >> t[j]) & 1u;
                        u[j] = (u[j] - isfire) & 0xffu;
                        t[j] += isfire;                                                // fire -> burnt ...
                    }
                    uint32_t ign_bits = 0;
                    static_for<4>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        using C = CM<W, H, j>;
                        t[j] -= lanebit(STILL[j]) ? 1u : 0u;                           // ... unless still burning
                        if constexpr (RARE) t[j] &= ~((t[j] << 2) & 8u);               // burnt-out blocked fire: 14 -> 6 (blocked burnt)
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
                for (int j = 0; j < 4; ++j) FIRE[j] = __ballot((t[j] & 3u) == 1u);
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
                        if constexpr (RARE)      // digging next to burning dirt / a blocked fire cell may cut its last link (see wf_step_group)
                            recompute |= dug & ((uint32_t)__ballot(blocked_burning(ty) && !rout && lane < 8u && !(lane & 1u)) ? 1u : 0u);
                    }
                    if (rfl(recompute)) {
                        u64 PASS[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) PASS[j] = ~__ballot(t[j] >= WF_DIRT);
                        static_for<4>([&](auto J) { constexpr int j = decltype(J)::value; RB[j] = CM<W, H, j>::BORDER & PASS[j]; });
                        flood<W, H>(RB, PASS);
                        if constexpr (RARE) {
                            u64 Z[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) Z[j] = __ballot(blocked_burning(t[j]));
                            add_blocked_burning<W, H>(RB, Z);
                        }
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


