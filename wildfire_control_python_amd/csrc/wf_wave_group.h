// wf_wave_group.h - single-step kernel, group layout (4 cells per lane), and the agent phases A / C shared with the rows layout
// Included by wf_kernels.hip inside its anonymous namespace (one translation unit: the launchers there
// instantiate these templates); not a stand-alone header.

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
// for round 1's one-environment-per-wave formulation (wf_step_wave; the n-step path has since moved
// to wf_step_ticks, which keeps this phase B and puts the per-environment logic on the scalar unit).
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
// PRE: the record comes in `pre` (a resident step loop keeps it in LDS between its steps) instead of from a.envs.
template <int W, int H, int VB, bool RARE, bool PRE = false>
__device__ __forceinline__ EnvRegs group_phase_a(const WfArgs &a, const bool lane_env, const uint32_t ek, const bool do_update,
                                                 const uint4 pre = make_uint4(0, 0, 0, 0))
{
    constexpr int CELLS = W * H;
    constexpr uint32_t ENV_BYTES = CELLS * 4;
    char *planes_b = reinterpret_cast<char *>(a.planes);
    const uint32_t pbase = ek * ENV_BYTES;
    uint4 ev = pre;
    if constexpr (!PRE) ev = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.envs) + ek * 16u);
    const uint32_t act = a.actions ? (uint32_t)(uint8_t)a.actions[ek] : 0xffu;
    // fl: the record's whole second word = flags | wind code << 16.  The flag logic only ever looks at and changes the
    // low bits, so the environment's wind code (wind with a direction; 0 otherwise) rides along into phase C's record.
    uint32_t ax = ev.x & 0xffffu, ay = ev.x >> 16, fl = ev.y, episode = ev.z, tick = ev.w;
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
    const bool toggling = RARE && (a.hot_flags & WF_HOT_DIG_TOGGLE) != 0;
    const uint32_t istog = toggling ? (act == 4u ? 1u : 0u) : 0u;
    // cell that may be dug: destination of a move, or the own cell when digging is toggled on
    const uint32_t xd = istog ? ax : (inb ? nx : ax), yd = istog ? ay : (inb ? ny : ay);
    // Containment shortcut: gather the passability of the 8-ring (N,NE,E,SE,S,SW,W,NW) around
    // the cell that may get dug; off-grid = impassable.
    uint32_t ring = 0, bnmask = 0, zring = 0;        // zring: ring cells that are burning dirt (toggle only)
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
            if constexpr (RARE) zring |= ((off ^ 1u) & (blocked_burning(ty) ? 1u : 0u)) << r;
        }
        // which 4-neighbours (ring bits 0,2,4,6 = N,E,S,W) are border cells themselves
        const uint32_t ex = x0 | xl, ey = y0 | yl;
        bnmask = (ex | (yd == 1u ? 1u : 0u)) | ((ey | (xd == (uint32_t)(W - 2) ? 1u : 0u)) << 2) |
                 ((ex | (yd == (uint32_t)(H - 2) ? 1u : 0u)) << 4) | ((ey | (xd == 1u ? 1u : 0u)) << 6);
    }
    const uint32_t moved = inb & min(td ^ (uint32_t)WF_WATER, 1u) & running;           // :146 traversable
    constexpr uint32_t ONFIRE = RARE ? FIRE_SET : 2u;                                  // is_burning(): type fire (RARE: blocked fire too)
    const uint32_t dfire = moved & (ONFIRE >> td);                                     // destination burning
    uint32_t dug = moved & (fl >> 2) & (DIG_SET >> td) & 1u;                           // :151-152, :125-133 (not dirt, not burning)
    uint32_t dugtype = (uint32_t)WF_DIRT;
    const uint32_t vis_before = (fl >> 1) & 1u & running;
    fl = (fl & ~((ism & running) << 1)) | (moved << 1) | (dfire << 3);                 // :142, :148-149, :154-155
    ax += moved * (nx - ax);
    ay += moved * (ny - ay);
    uint32_t ta = tcur + moved * (td - tcur);                                          // type under the agent
    {   // Agent.toggle_digging (forest_fire.py:35-36, environment.py:136-138)
        const uint32_t tg = istog & running;
        fl ^= tg << 2;
        // dig() under the agent: any cell that is not dirt - a BURNING one too, which stays in burning_cells
        // under its new type (WF_DIRT_BURNING)
        const uint32_t dugt = (fl >> 2) & ((DIG_SET | ONFIRE) >> tcur) & 1u;
        dug = tg ? dugt : dug;
        qd = tg ? qa : qd;
        if constexpr (RARE) dugtype = (tg & (ONFIRE >> tcur) & 1u) ? (uint32_t)WF_DIRT_BURNING : dugtype;
    }
    ta += dug * ((uint32_t)WF_DIRT - ta);
    if (do_update) {   // forest_fire.py:87 drop dead agents (Agent.is_dead environment.py:116-120)
        const uint32_t dies = fl & ((fl >> 3) | (ONFIRE >> ta)) & 1u & running;
        fl &= ~(dies * (uint32_t)(WF_F_ALIVE | WF_F_VISIBLE));
    }
    const uint32_t vis_after = (fl >> 1) & 1u;
    const uint32_t qa_new = ay * W + ax;
    // apply the touched bytes to the grid in memory (phase B reloads the whole grid)
    if (vis_before) *reinterpret_cast<uint8_t *>(planes_b + (pbase + qa * 4u + 3u)) = 0;
    if (dug) *reinterpret_cast<uint8_t *>(planes_b + (pbase + qd * 4u)) = (uint8_t)dugtype;
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
        // a burning-dirt 4-neighbour hangs on the board through its neighbours only (add_blocked_burning): digging
        // next to it may have been its last link, which the passable-cell argument above does not see
        ringflag = dug & (local_fail | border_fail | min(zring & 0x55u, 1u)) & 1u;
    }
    const uint32_t precheck = ((a.hot_flags & WF_HOT_CONTAINMENT) ? 1u : 0u) & (((fl >> 5) | (fl >> 6)) ^ 1u) & 1u;
    const uint32_t Pv = running | (precheck << 1) | (((fl >> VB) & 1u) << 2) | (ringflag << 3);   // VB: this layout's board-valid flag
    EnvRegs S = { ax, ay, fl, episode, tick, Pv };
    return S;
}

// Phase C for the environment of this lane: flags, reward, done, record (Rv = RB_* bits of phase B).
// Returns the environment's record as written (x | y << 16, flags, episode, tick).
template <int W, int H, int VB>
__device__ __forceinline__ uint4 group_phase_c(const WfArgs &a, const EnvRegs &S, const uint32_t Rv, const bool lane_env,
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
    return make_uint4(ax | (ay << 16), fl, episode, tick);
}

// SHARED = false: every wave runs phases A, B, C for its own group of G environments.
// SHARED = true : a workgroup owns 64 environments.  Wave 0 runs phase A for all 64 at once (all
//   lanes busy instead of G of them), the registers travel through LDS, the 64/G waves each run
//   phase B for G environments, and whichever wave finishes last runs phase C for all 64.  This
//   divides the per-environment instruction cost of phases A and C by 64/G.
// WIND: wind with a direction (wf_params.wind_mode != 0; environment.py:187-193, 260-287).  The heat a burning cell hands
// to a neighbour depends on the direction between them, so a grass cell keeps one count per direction (a.wind_cnt: a
// dword per cell, bytes = addends received from the cell below / above / left / right, i.e. heating it N / S / E / W) and
// ignites when the float64 sum of count x addend exceeds the threshold; the environment's four addends are
// wave-uniform (scalar loads from the table the host computed with the reference's own expression), the four "neighbour
// still burning" bits are the lane bits of the same masks the calm rule counts.  Plane 2 then only records how a cell was
// ignited.  No count a cell can reach brings the sum within rounding distance of the threshold (wf_create checks it), so
// the order in which the reference adds its addends decides nothing.
template <int W, int H, bool SHARED, bool RARE, bool WIND = false>
__global__ __launch_bounds__(SHARED ? 512 : 256, SHARED ? 8 : WF_LB) void wf_step_group(const WfArgs a)
{
    static_assert(!(SHARED && WIND), "wind runs on the private variant");
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
            S = group_phase_a<W, H, 7, RARE>(a, lane_env, ek, do_update);
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
        S = group_phase_a<W, H, 7, RARE>(a, lane_env, ek, do_update);
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
    const uint32_t K_FIRE_STEP = 0x0000FF01u;                // low half: fuel byte -1, type byte +1 (fire -> burnt); pk_add_u16
    __shared__ __attribute__((aligned(16))) uint8_t sObs[SHARED ? 8 : 4][ObsGeo<W, H>::PAD];
    uint8_t *obs_lds = sObs[wv];
    uint32_t oj[4];
    obs_offsets<W, H>(lane, oj);
    if (active) {
        uint4 raw_next = *reinterpret_cast<const uint4 *>(planes_b + (e0 * ENV_BYTES + lane_off));
        uint4 cnt_next = make_uint4(0, 0, 0, 0);
        if constexpr (WIND) cnt_next = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.wind_cnt) + (e0 * ENV_BYTES + lane_off));
        for (uint32_t k = 0; k < ng; ++k) {
            const uint32_t e = e0 + k;
            const uint4 raw = raw_next;
            const uint4 cnt_raw = cnt_next;
            {   // prefetch the next environment of the group (clamped: the last one re-reads itself)
                const uint32_t en = e0 + min(k + 1u, ng - 1u);
                raw_next = *reinterpret_cast<const uint4 *>(planes_b + (en * ENV_BYTES + lane_off));
                if constexpr (WIND) cnt_next = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.wind_cnt) + (en * ENV_BYTES + lane_off));
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
                    reset_regs<W, H, RARE>(cold, cc->p.seed, e_pin + a.env_base, lane, t, u, h, s, a.reach + (size_t)e_pin * 4);
                    const uint32_t qs = s.ay * W + s.ax;
                    uint32_t r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r[j] = t[j] | (u[j] << 8) | (h[j] << 16) | ((lane == (qs >> 2) && (qs & 3u) == (uint32_t)j) ? (1u << 24) : 0u);
                    *reinterpret_cast<uint4 *>(planes_b + (e * ENV_BYTES + lane_off)) = make_uint4(r[0], r[1], r[2], r[3]);
                    if constexpr (WIND)      // a fresh grid has received no heat
                        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.wind_cnt) + (e * ENV_BYTES + lane_off)) = make_uint4(0, 0, 0, 0);
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
                    const u64 F = __ballot(burning<RARE>(c[j]));                         // burning: fire or burning dirt
                    STILL[j] = F & __ballot((int32_t)(int8_t)(c[j] >> 8) > 1);     // fuel - 1 > 0
                    // fuel -= 1 and fire -> burnt, unless still burning (type stays fire)
                    const uint32_t d = lanebit(STILL[j]) ? (K_FIRE_STEP - 1u) : burn_out_step<RARE>(c[j]);
                    c[j] = pk_add_u16(c[j], lanebit(F) ? d : 0u);
                }
                uint32_t ign_bits = 0;
                // wind: this environment's addends per direction and its count words (one dword per cell)
                double wc0 = 0.0, wc1 = 0.0, wc2 = 0.0, wc3 = 0.0, threshold = 0.0;
                uint32_t n[4] = { 0, 0, 0, 0 }, nchg = 0;
                char *cnt_p = nullptr;
                if constexpr (WIND) {
                    const uint32_t code = (uint32_t)__builtin_amdgcn_readlane(S.fl, k) >> 16;
                    cold_ptr cc = (cold_ptr)(uintptr_t)a.cold;
                    const double __attribute__((address_space(4))) *tab = cc->wind_c[(code >> 4) & 3u][code & 15u];
                    wc0 = tab[0]; wc1 = tab[1]; wc2 = tab[2]; wc3 = tab[3];
                    threshold = cc->p.threshold;
                    cnt_p = reinterpret_cast<char *>(a.wind_cnt) + (e * ENV_BYTES + lane_off);
                    n[0] = cnt_raw.x; n[1] = cnt_raw.y; n[2] = cnt_raw.z; n[3] = cnt_raw.w;
                }
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
                    u64 IGN;
                    if constexpr (WIND) {
                        // direction k of the vector burning cell -> this cell: the cell BELOW heats it northwards (k = 0),
                        // the one above southwards (1), the left one eastwards (2), the right one westwards (3)
                        uint32_t inc = (lanebit(nb_dn<W, H, j>(STILL)) ? 1u : 0u) | (lanebit(nb_up<W, H, j>(STILL)) ? 0x100u : 0u) |
                                       (lanebit(LF) ? 0x10000u : 0u) | (lanebit(RT) ? 0x1000000u : 0u);
                        inc = lanebit(GR) ? inc : 0u;
                        // a count byte saturates at 255 (bytes that are 0xFF take no increment)
                        const uint32_t full = ((((n[j] & 0x7f7f7f7fu) + 0x01010101u) & n[j]) & 0x80808080u) >> 7;
                        const uint32_t nn = n[j] + (inc & ~full);
                        const double temp = (double)(nn & 0xffu) * wc0 + (double)((nn >> 8) & 0xffu) * wc1 +
                                            (double)((nn >> 16) & 0xffu) * wc2 + (double)(nn >> 24) * wc3;
                        IGN = GR & __ballot(inc != 0u && temp > threshold);           // a new addend arrived and the sum is over
                        nchg |= nn ^ n[j];
                        n[j] = nn;
                        const uint32_t ig = lanebit(IGN) ? 1u : 0u;
                        c[j] = ig ? (c[j] | ((uint32_t)WF_HEAT_IGNITED << 16)) + 1u : c[j];     // grass -> fire, marked "ignited by heat"
                        ign_bits |= ig << j;
                    } else {
                        const uint32_t cnt = (lanebit(nb_up<W, H, j>(STILL)) ? 1u : 0u) + (lanebit(nb_dn<W, H, j>(STILL)) ? 1u : 0u) +
                                             (lanebit(LF) ? 1u : 0u) + (lanebit(RT) ? 1u : 0u);
                        const uint32_t room = ign - ((c[j] >> 16) & 0xffu);            // quanta still missing (>= 1 on grass)
                        IGN = GR & __ballot(cnt >= room);
                        const uint32_t ig = lanebit(IGN) ? 1u : 0u;
                        const uint32_t cn = c[j] + (min(cnt, room) << 16) + ig;        // heat saturates; grass -> fire
                        c[j] = lanebit(GR) ? cn : c[j];
                        ign_bits |= ig << j;
                    }
                    FIRE[j] = STILL[j] | IGN;
                });
                if constexpr (WIND) {
                    if (nchg) *reinterpret_cast<uint4 *>(cnt_p) = make_uint4(n[0], n[1], n[2], n[3]);
                }
                // set_fire_to :245-246: a newly ignited border cell
                R |= __ballot((ign_bits & bgeo) != 0u) ? RB_BORDER : 0u;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) FIRE[j] = __ballot(burning<RARE>(c[j]));
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
                    if constexpr (RARE) {
                        u64 Z[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) Z[j] = __ballot(blocked_burning(c[j]));
                        add_blocked_burning<W, H>(RB, Z);
                    }
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

