// wf_wave_ticks.h - fused multi-step kernel of the wave family: one environment per wave, the grid PACKED in registers
// (four dwords per lane, as they lie in memory) and everything per-environment on the SCALAR unit.
// Included by wf_kernels.hip inside its anonymous namespace (one translation unit: the launchers there
// instantiate these templates); not a stand-alone header.
//
// n_steps x ForestFire.step() (forest_fire.py:30-49) per environment and launch.  wf_step_wave (wf_wave_fused.h), which
// this kernel replaces on the n-step path, keeps type / fuel / heat of a cell in three registers and runs the
// per-environment logic (Agent.move, flags, reward) on the vector unit with every lane holding the same value: ~250
// vector instructions per environment-tick, of which the tick itself needs ~90.  With one environment per wave the
// agent's position, the flags word, the reward and the reachability board are wave-uniform: they live in scalar
// registers here, the agent reads the few cell types it needs with one ds_bpermute (ring + destination + own cell), and
// the vector unit only does the tick on the packed words (the formulation of wf_step_group's phase B).  BASELINE
// configs[1] (4 096 x 10x10, 16 ticks per call: four waves per SIMD) is bound by instruction issue - 72 vector + 84 scalar
// instructions per tick in the no-action loop, and the scalar unit issues one instruction per cycle for the four SIMDs
// of a CU, i.e. a SIMD gets a scalar slot as often as a vector slot: scalar instructions are not free.
// Round 4: a call of 2 048 environments or more goes out as two or four launches of this kernel on the context's streams
// (wf_abi.cpp: ticks_parts) - 4 096 x 10x10 x 16 ticks 17.3 -> 13.1 us per call; the vector forms used here (compares, mask read-backs,
// carry-in adds) issue at 4.25 cycles per wave64 instruction, the rate of the scalar unit (profiles/r04_issue_rates_*.txt).

// type byte of cell q (wave-uniform q) of the packed grid c
__device__ __forceinline__ uint32_t cell_type_at(const uint32_t (&c)[4], uint32_t q)
{
    const uint32_t k = q & 3u;
    const uint32_t v = k == 0u ? c[0] : k == 1u ? c[1] : k == 2u ? c[2] : c[3];
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(q >> 2)) & 0xffu;
}

__device__ __forceinline__ double rfl_f64(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    return __longlong_as_double((long long)((u64)rfl((uint32_t)b) | ((u64)rfl((uint32_t)(b >> 32)) << 32)));
}

// 1 if the wave mask / the word is not zero, else 0, as scalar ARITHMETIC (the compiler's own "m != 0 ? 1 : 0" on a
// wave-uniform value goes through a vector register and back)
__device__ __forceinline__ uint32_t nz64(u64 m)
{
    uint32_t r;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(m) : "scc");
    return r;
}
__device__ __forceinline__ uint32_t nz32(uint32_t v)
{
    uint32_t r;
    asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(v) : "scc");
    return r;
}

#ifndef WF_TICKS_LB
#define WF_TICKS_LB 6   // waves per SIMD the register allocator must leave room for (8: 80 scalar registers, 145 of them spilled, 22 vector registers in scratch)
#endif
template <int W, int H, bool RARE>
__global__ __launch_bounds__(256, WF_TICKS_LB) void wf_step_ticks(const WfArgs a)
{
    constexpr int CELLS = W * H, LANES = CELLS / 4;
    constexpr uint32_t ENV_BYTES = CELLS * 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(4))) *const_u32x4_ptr;
    typedef u64 u64x4 __attribute__((ext_vector_type(4)));
    typedef const u64x4 __attribute__((address_space(4))) *const_u64x4_ptr;
    const uint32_t lane = __lane_id();
    // EXEC = the lanes that own cells, for the whole kernel: idle lanes never load, store or put bits into a ballot
    if (lane >= (uint32_t)LANES) return;
    const uint32_t wave0 = rfl(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * 4u;
    const uint32_t batch = (uint32_t)a.batch;
    const uint32_t ign = (uint32_t)a.ignite_count;
    char *planes_b = reinterpret_cast<char *>(a.planes);
    const uint32_t lane_off = lane * 16u;
    // bgeo bit j = the lane's cell j is a border cell (per-lane constant instead of 64-bit scalar constants)
    uint32_t bgeo = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t q = 4u * lane + (uint32_t)j, x = q % W, y = q / W;
        const uint32_t inner = ((x - 1u) | (y - 1u) | ((uint32_t)(W - 2) - x) | ((uint32_t)(H - 2) - y)) >> 31;
        bgeo |= inner << j;
    }
    // the agent's gather lanes: 0..7 = the 8-ring N,NE,E,SE,S,SW,W,NW around the cell that may be dug, 8 = that cell
    // (the destination of the move), 9 = the agent's own cell
    const uint32_t rk = lane & 7u;
    uint32_t gdx = lane < 8u ? ((0x01A9u >> (2u * rk)) & 3u) - 1u : 0u;
    uint32_t gdy = lane < 8u ? ((0x1A90u >> (2u * rk)) & 3u) - 1u : 0u;
    asm volatile("" : "+v"(bgeo), "+v"(gdx), "+v"(gdy));
    __shared__ __attribute__((aligned(16))) uint8_t sObs[4][ObsGeo<W, H>::PAD];
    uint8_t *obs_lds = sObs[threadIdx.x >> 6];
    constexpr uint32_t K_FIRE_STEP = 0x0000FF01u;            // low half: fuel byte -1, type byte +1 (fire -> burnt); pk_add_u16
    constexpr uint32_t ONFIRE = RARE ? FIRE_SET : 2u;        // is_burning(): type fire (RARE: blocked fire too)
    const int n_steps = a.n_steps;

    // One environment, all its steps.  The loop is unswitched on two kernel-uniform facts: ACT = the call carries actions
    // (a no-op call - BASELINE configs[1] - runs a body without the agent phase), ALLUPD = every step of the call runs
    // update() (a_speed = 1, the reference's default: no "does this step update" masks).
    auto run_env = [&](auto ACTC, auto UPDC, const uint32_t e) {
        constexpr bool ACT = decltype(ACTC)::value, ALLUPD = decltype(UPDC)::value;
        const uint32_t stride = (uint32_t)a.step_stride;     // row length of the (n_steps, context batch) action / reward / done arrays
        const uint4 raw = *reinterpret_cast<const uint4 *>(planes_b + (e * ENV_BYTES + lane_off));
        const u32x4 ev = *(const_u32x4_ptr)(uintptr_t)(reinterpret_cast<const char *>(a.envs) + e * 16u);
        // the environment's border-reachability board lives in scalar registers for the whole launch (a board written
        // in this launch must not be re-read through the scalar cache)
        u64 RB[4];
        {
            const u64x4 rbv = *(const_u64x4_ptr)(uintptr_t)(a.reach + (size_t)e * 4);
            RB[0] = rbv.x; RB[1] = rbv.y; RB[2] = rbv.z; RB[3] = rbv.w;
        }
        // (fl: the record's whole second word; its upper half - the wind code of wind contexts, unused here - rides along)
        uint32_t ax = ev.x & 0xffffu, ay = ev.x >> 16, fl = ev.y, episode = ev.z, tick = ev.w;
        // packed cells: type | fuel << 8 | heat << 16; the agent_pos byte is rebuilt from the record at the end
        uint32_t c[4] = { raw.x & 0x00ffffffu, raw.y & 0x00ffffffu, raw.z & 0x00ffffffu, raw.w & 0x00ffffffu };
        bool dirty = false;
        // am bit j = cell j of this lane is the agent's cell (a per-lane register: "is the agent standing on fire" is one
        // AND against the lane's burning bits instead of a four-way select on the scalar unit)
        // watch = that cell can burn under the agent at all (grass or fire: an agent that digs stands on dirt, and the test is skipped)
        uint32_t am = 0, watch = 1;
        auto place_agent = [&](const uint32_t type_under) {
            const uint32_t qn = ay * W + ax;
            am = lane == (qn >> 2) ? 1u << (qn & 3u) : 0u;
            watch = RARE ? 1u : ((1u << WF_GRASS | 1u << WF_FIRE) >> min(type_under, 31u)) & 1u;
        };
        place_agent(cell_type_at(c, ay * W + ax));

        // No memory operation per step: the actions of a block of 16 steps come in with one load (lane k = step k of the
        // block), its rewards / done flags collect in lanes 0..15 of three registers and leave with one store each.
        uint32_t vact = 0xffu, vr_lo = 0, vr_hi = 0, vdone = 0;
        auto put = [&](const uint32_t slot, const double reward, const uint32_t done) {
            const u64 rbits = (u64)__double_as_longlong(reward);
            // (two scalar operands of one vector instruction: the lane select goes through m0)
            asm("s_mov_b32 m0, %6\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                : "+v"(vr_lo), "+v"(vr_hi), "+v"(vdone)
                : "s"(rfl((uint32_t)rbits)), "s"(rfl((uint32_t)(rbits >> 32))), "s"(rfl(done)), "s"(rfl(slot))
                : "m0");
        };
        // the steps base .. base + count - 1 leave the three registers.  (Called at the TOP of a block and after the loop,
        // never right before a loop's back edge: a divergent branch that joins at the loop header would make the
        // compiler treat every loop-carried value - the step counter, the flags word - as divergent.)
        auto flush = [&](const uint32_t base, const uint32_t count) {
            if (lane < count) {
                const uint32_t oi = (base + lane) * stride + e;
                if (a.reward) a.reward[oi] = __hiloint2double((int)vr_hi, (int)vr_lo);
                if (a.done) a.done[oi] = (uint8_t)vdone;
            }
        };
        // The environment is finished (World.RUNNING off) when step `slot` of a block of bn begins.  With auto-reset that
        // step returns the fresh episode (gymnasium-style next-step autoreset; reward 0, done 0) and the next one runs
        // on; without, the environment stays frozen: every remaining step of the block reports (0, done).  Returns the
        // number of steps this took.  Rare, and an if-then off the step loop: the loop body itself only ever sees a
        // running environment.
        auto finished = [&](const uint32_t slot, const uint32_t bn) -> uint32_t {
            if (a.hot_flags & WF_HOT_AUTO_RESET) {
                // Inputs are pinned inside the branch so none of this is speculated above it.
                EnvS s = { ax, ay, fl, episode, tick };
                const WfCold *cold = a.cold;
                uint32_t e_pin = e;
                asm volatile("" : "+s"(cold), "+s"(e_pin));
                cold_ptr cc = (cold_ptr)(uintptr_t)cold;
                uint32_t t[4], u[4], h[4];
                reset_regs<W, H, RARE>(cold, cc->p.seed, e_pin + a.env_base, lane, t, u, h, s, a.reach + (size_t)e_pin * 4);   // stream keyed by the context index
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = t[j] | (u[j] << 8) | (h[j] << 16);
                // (every lane computed the same record: back to the scalar unit)
                ax = rfl(s.ax); ay = rfl(s.ay); fl = rfl(s.fl); episode = rfl(s.episode); tick = rfl(s.tick);
                const uint32_t qs = ay * W + ax;             // the board reset_regs wrote (closed form; invalid with a river)
                static_for<4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    RB[j] = CM<W, H, j>::VALID & ~(((qs & 3u) == (uint32_t)j) ? (1ull << (qs >> 2)) : 0ull);
                });
                place_agent((uint32_t)WF_DIRT);              // Agent.__init__ digs its spawn cell (:111-113)
                dirty = true;
                put(slot, 0.0, 0u);
                return 1u;
            }
            const bool rest = lane >= slot;                  // lanes slot .. 15: the block's remaining steps
            vr_lo = rest ? 0u : vr_lo;
            vr_hi = rest ? 0u : vr_hi;
            vdone = rest ? 1u : vdone;
            return bn - slot;
        };
        // containment test off for this context: as if the fire had touched the border already (environment.py:345-346)
        const uint32_t nocont = (a.hot_flags & WF_HOT_CONTAINMENT) ? 0u : (uint32_t)WF_F_BORDER;

        for (uint32_t b0 = 0; b0 < (uint32_t)n_steps; b0 += 16u) {
            const uint32_t bn = min(16u, (uint32_t)n_steps - b0);
            if (b0) flush(b0 - 16u, 16u);
            if constexpr (ACT) {
                vact = 0xffu;
                if (lane < bn) vact = (uint32_t)(uint8_t)a.actions[(b0 + lane) * stride + e];
            }
            {   // every step of the block: the default reward, not done - unless its step says otherwise
                const u64 dbits = (u64)__double_as_longlong(a.default_reward);
                vr_lo = (uint32_t)dbits;
                vr_hi = (uint32_t)(dbits >> 32);
                vdone = 0;
            }
            uint32_t k = 0;
            if (__builtin_expect(!(fl & WF_F_RUNNING), 0)) k = finished(0u, bn);
            for (; k < bn; ++k) {
                dirty = true;
                uint32_t upd = 1u;                           // forest_fire.py:40-43: this step runs update()
                if constexpr (!ALLUPD) upd = (uint32_t)(a.update_mask >> (b0 + k)) & 1u;
                uint32_t ringflag = 0;                       // "this dig may have cut a region off the border"
                if constexpr (ACT) {
                    // ---- Agent.move (environment.py:141-155) on the scalar unit
                    const uint32_t act = (uint32_t)__builtin_amdgcn_readlane((int)vact, (int)k);
                    const uint32_t ent = MOVE_LUT >> (min(act, 4u) * 6u);          // act is an unsigned byte
                    const uint32_t ism = (ent >> 4) & 1u;                          // action in N,S,E,W
                    const uint32_t nx = ax + (ent & 3u) - 1u;                      // :163-171
                    const uint32_t ny = ay + ((ent >> 2) & 3u) - 1u;
                    const uint32_t out = (((uint32_t)(W - 1) - nx) | nx | ((uint32_t)(H - 1) - ny) | ny) >> 31;
                    const uint32_t inb = ism & (out ^ 1u);                         // :146 inbounds
                    const uint32_t qa = ay * W + ax;
                    uint32_t qd = qa + inb * (ny * W + nx - qa);
                    const bool toggling = RARE && (a.hot_flags & WF_HOT_DIG_TOGGLE) != 0;
                    const uint32_t istog = toggling ? (act == 4u ? 1u : 0u) : 0u;
                    // cell that may be dug: destination of a move, or the own cell when digging is toggled on
                    const uint32_t xd = istog ? ax : (inb ? nx : ax), yd = istog ? ay : (inb ? ny : ay);
                    // one gather for every cell type the agent looks at (types BEFORE the dig)
                    uint32_t ring, zring = 0, td, tcur;
                    {
                        const uint32_t gx = (lane == 9u ? ax : xd) + gdx, gy = (lane == 9u ? ay : yd) + gdy;
                        const uint32_t off = (((uint32_t)(W - 1) - gx) | gx | ((uint32_t)(H - 1) - gy) | gy) >> 31;   // off the grid
                        const uint32_t nq = off ? 0u : gy * W + gx;
                        const uint32_t tp = (c[0] & 0xffu) | ((c[1] & 0xffu) << 8) | ((c[2] & 0xffu) << 16) | (c[3] << 24);
                        const uint32_t ty = __builtin_amdgcn_ubfe(bperm(nq & ~3u, tp), (nq & 3u) << 3, 8u);
                        ring = (uint32_t)__ballot(!off && ty < (uint32_t)WF_DIRT) & 0xffu;
                        if constexpr (RARE) zring = (uint32_t)__ballot(!off && blocked_burning(ty)) & 0xffu;
                        td = (uint32_t)__builtin_amdgcn_readlane((int)ty, 8);
                        tcur = (uint32_t)__builtin_amdgcn_readlane((int)ty, 9);
                    }
                    const uint32_t moved = inb & min(td ^ (uint32_t)WF_WATER, 1u); // :146 traversable
                    const uint32_t dfire = moved & (ONFIRE >> td) & 1u;            // destination burning
                    uint32_t dug = moved & (fl >> 2) & (DIG_SET >> td) & 1u;       // :151-152, :125-133 (not dirt, not burning)
                    uint32_t dugtype = (uint32_t)WF_DIRT, dugold = td;
                    fl = (fl & ~(ism << 1)) | (moved << 1) | (dfire << 3);         // :142, :148-149, :154-155
                    ax += moved * (nx - ax);
                    ay += moved * (ny - ay);
                    if constexpr (RARE) {   // Agent.toggle_digging (forest_fire.py:35-36, environment.py:136-138)
                        fl ^= istog << 2;
                        // dig() under the agent: any cell that is not dirt - a BURNING one too, which stays in
                        // burning_cells under its new type (WF_DIRT_BURNING)
                        const uint32_t dugt = (fl >> 2) & ((DIG_SET | ONFIRE) >> tcur) & 1u;
                        dug = istog ? dugt : dug;
                        qd = istog ? qa : qd;
                        dugold = istog ? tcur : dugold;
                        dugtype = (istog & (ONFIRE >> tcur) & 1u) ? (uint32_t)WF_DIRT_BURNING : dugtype;
                    }
                    {   // the dug cell's type byte, in the registers of its lane; the agent's new lane bit
                        const uint32_t delta = dug ? (dugold ^ dugtype) : 0u;
                        const uint32_t dl = lane == (qd >> 2) ? delta : 0u;
                        const uint32_t kk = qd & 3u;
                        c[0] ^= kk == 0u ? dl : 0u;
                        c[1] ^= kk == 1u ? dl : 0u;
                        c[2] ^= kk == 2u ? dl : 0u;
                        c[3] ^= kk == 3u ? dl : 0u;
                        place_agent(dug ? (uint32_t)WF_DIRT : moved ? td : tcur);
                    }
                    // Verdict: can this dig change which cells reach the border?  (exact component count on the ring
                    // and the border refinement: see group_phase_a.)  Nobody asks once the fire has touched the border or
                    // the bonus is paid - the containment test is over for the episode (environment.py:345-346).
                    if (dug & (nz32((fl | nocont) & (uint32_t)(WF_F_BORDER | WF_F_CONTAINED)) ^ 1u)) {
                        const uint32_t x0 = xd == 0u ? 1u : 0u, xl = xd == (uint32_t)(W - 1) ? 1u : 0u;
                        const uint32_t y0 = yd == 0u ? 1u : 0u, yl = yd == (uint32_t)(H - 1) ? 1u : 0u;
                        const uint32_t ex = x0 | xl, ey = y0 | yl;
                        const uint32_t bnmask = (ex | (yd == 1u ? 1u : 0u)) | ((ey | (xd == (uint32_t)(W - 2) ? 1u : 0u)) << 2) |
                                                ((ex | (yd == (uint32_t)(H - 2) ? 1u : 0u)) << 4) | ((ey | (xd == 1u ? 1u : 0u)) << 6);
                        const uint32_t onb = (((xd - 1u) | (yd - 1u) | ((uint32_t)(W - 2) - xd) | ((uint32_t)(H - 2) - yd)) >> 31);
                        const uint32_t even = ring & 0x55u;
                        const uint32_t eff = ring & 0xAAu & (even << 1) & ((even >> 1) | (even << 7));
                        const uint32_t comps = (uint32_t)__popc(even) - (uint32_t)__popc(eff);
                        const uint32_t local_fail = (comps + 0x7FFFFFFEu) >> 31;          // comps >= 2 (comps in 0..4)
                        const uint32_t border_fail = onb & min(even, 1u) & (min(ring & bnmask, 1u) ^ 1u);
                        ringflag = (local_fail | border_fail | min(zring & 0x55u, 1u)) & 1u;
                    }
                }

                // ---- ForestFire.update (forest_fire.py:85-106).  Straight-line: a step that does not update (a_speed > 1,
                // ALLUPD = false) runs it on empty masks, which changes nothing.
                const u64 updm = upd ? ~0ull : 0ull;
                u64 FR[4], STILL[4], FIRE[4];                // burning at the start of the step / still burning / fire afterwards
                uint32_t onfire = 0;                         // the agent stands on a burning cell (is_dead :116-120)
                if constexpr (RARE) onfire = (ONFIRE >> cell_type_at(c, ay * W + ax)) & 1u;   // (burning dirt under it is type dirt: not "burning")
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // reduce_fuel (environment.py:297-307)
                    FR[j] = __ballot(burning<RARE>(c[j]));                         // burning: fire, burning dirt, blocked fire
                    const u64 F = ALLUPD ? FR[j] : FR[j] & updm;
                    STILL[j] = F & __ballot((int32_t)(int8_t)(c[j] >> 8) > 1);     // fuel - 1 > 0
                    // fuel -= 1 and fire -> burnt, unless still burning (type stays fire)
                    const uint32_t d = lanebit(STILL[j]) ? (K_FIRE_STEP - 1u) : burn_out_step<RARE>(c[j]);
                    c[j] = pk_add_u16(c[j], lanebit(F) ? d : 0u);
                }
                {   // :87 drop dead agents (Agent.is_dead environment.py:116-120): dead, or standing on a burning cell
                    if (!RARE && __builtin_expect(watch, 0)) {
                        // without the rare codes "burning" is "type fire": the lane's burning bits (one add-with-carry
                        // per component: x = 2x + bit) against the agent's bit
                        uint32_t fb = lanebit(FR[3]) ? 1u : 0u;
                        fb = addc(fb, fb, FR[2]);
                        fb = addc(fb, fb, FR[1]);
                        fb = addc(fb, fb, FR[0]);                                  // bit j = cell j of the lane burns
                        onfire = nz64(__ballot((fb & am) != 0u));
                    }
                    const uint32_t dies = upd & fl & ((fl >> 3) | onfire) & 1u;    // alive & (dead | on fire)
                    fl &= ~(dies | (dies << 1));                                   // WF_F_ALIVE | WF_F_VISIBLE (no multiply: it would go through the vector unit)
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
                    // the still-burning 4-neighbours: one add-with-carry per neighbour mask
                    const uint32_t cnt = addc0(addc0(addc0(lanebit(nb_up<W, H, j>(STILL)) ? 1u : 0u, nb_dn<W, H, j>(STILL)), LF), RT);
                    const uint32_t room = ign - ((c[j] >> 16) & 0xffu);            // quanta still missing (>= 1 on grass)
                    u64 IGN = GR & __ballot(cnt >= room);
                    if constexpr (!ALLUPD) IGN &= updm;
                    const uint32_t ig = lanebit(IGN) ? 1u : 0u;
                    const uint32_t cn = c[j] + (min(cnt, room) << 16) + ig;        // heat saturates; grass -> fire
                    c[j] = lanebit(GR) ? cn : c[j];
                    ign_bits |= ig << j;
                    FIRE[j] = STILL[j] | IGN;
                    if constexpr (!ALLUPD) FIRE[j] = upd ? FIRE[j] : FR[j];
                });
                // set_fire_to :245-246: a newly ignited border cell
                fl |= nz64(__ballot((ign_bits & bgeo) != 0u)) << 5;                 // WF_F_BORDER
                tick += upd;
                const uint32_t anyfire = nz64(FIRE[0] | FIRE[1] | FIRE[2] | FIRE[3]);
                fl &= ~((upd & ((fl & anyfire) ^ 1u)) << 4);                       // :105-106 (!alive || !fire) -> !RUNNING

                // ---- World.get_reward (environment.py:342-390)
                uint32_t contained = 0;
                // :345-346  the containment test runs while the fire has not touched the border
                const uint32_t check = anyfire & (nz32((fl | nocont) & (uint32_t)(WF_F_BORDER | WF_F_CONTAINED)) ^ 1u);
                fl &= ~((ringflag & (check ^ 1u)) << 7);                           // a dig nobody looked at: board no longer trustworthy
                if (check) {
                    if (__builtin_expect((((fl >> 7) ^ 1u) | ringflag) & 1u, 0)) { // board not valid, or the dig may have cut it
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
                        fl = (fl | WF_F_REACH_VALID) & ~(uint32_t)WF_F_REACH_ROWS;  // the rows-layout board is stale now
                        if (lane == 0) {
                            ulonglong2 *rp = reinterpret_cast<ulonglong2 *>(a.reach + (size_t)e * 4);
                            rp[0] = make_ulonglong2(RB[0], RB[1]);
                            rp[1] = make_ulonglong2(RB[2], RB[3]);
                        }
                    }
                    // stale RB bits on cells dug since the last flood are harmless: fire never sits there
                    const u64 hit = (FIRE[0] & RB[0]) | (FIRE[1] & RB[1]) | (FIRE[2] & RB[2]) | (FIRE[3] & RB[3]);
                    contained = nz64(hit) ^ 1u;
                    fl |= contained << 6;                                          // :361-367
                }
                // anything but "the default reward (:390), not done" - what the block's registers were pre-set to -: burnt out,
                // agent dead, contained (rare)
                if (__builtin_expect(((anyfire & fl) ^ 1u) | contained, 0)) {
                    double reward = a.default_reward;
                    if (!anyfire) {                                                // :384-387 burnt out
                        int healthy = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) healthy += __popcll(__ballot((c[j] & 0xffu) == (uint32_t)WF_GRASS));
                        reward = rfl_f64(a.contained_bonus * ((double)healthy / (double)CELLS));
                    }
                    reward = (fl & WF_F_ALIVE) ? reward : a.death_penalty;         // :380-381
                    reward = rfl_f64(contained ? a.contained_bonus : reward);
                    put(k, reward, ((fl >> 4) & 1u) ^ 1u);                         // done: forest_fire.py:48
                }
                if (__builtin_expect(!(fl & WF_F_RUNNING), 0)) {                   // the episode ended with this step
                    if (k + 1u < bn) k += finished(k + 1u, bn);
                }
            }
        }
        flush(((uint32_t)n_steps - 1u) & ~15u, (((uint32_t)n_steps - 1u) & 15u) + 1u);

        if (dirty) {
            const uint32_t qa = ay * W + ax;
            const uint32_t sel = (qa & 3u) + ((min((qa >> 2) ^ lane, 1u) | (((fl >> 1) & 1u) ^ 1u)) << 2);
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = c[j] | (sel == (uint32_t)j ? (1u << 24) : 0u);
            *reinterpret_cast<uint4 *>(planes_b + (e * ENV_BYTES + lane_off)) = make_uint4(r[0], r[1], r[2], r[3]);
            if (a.obs) {
                uint32_t oj[4];
                obs_offsets<W, H>(lane, oj);
                emit_obs<W, H>(obs_lds, a.obs + (size_t)e * (CELLS * 3), r, oj, lane);
            }
            if (lane == 0)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.envs) + e * 16u) = make_uint4(ax | (ay << 16), fl, episode, tick);
        } else if (a.obs) {                // frozen env: observation of the unchanged grid
            const uint32_t cf[4] = { raw.x, raw.y, raw.z, raw.w };
            uint32_t oj[4];
            obs_offsets<W, H>(lane, oj);
            emit_obs<W, H>(obs_lds, a.obs + (size_t)e * (CELLS * 3), cf, oj, lane);
        }
    };
    const u64 allm = n_steps >= 64 ? ~0ull : (1ull << n_steps) - 1ull;
    const bool allupd = (a.update_mask & allm) == allm;
    if (a.actions) {
        if (allupd) for (uint32_t e = wave0; e < batch; e += nwaves) run_env(std::true_type{}, std::true_type{}, e);
        else for (uint32_t e = wave0; e < batch; e += nwaves) run_env(std::true_type{}, std::false_type{}, e);
    } else {
        if (allupd) for (uint32_t e = wave0; e < batch; e += nwaves) run_env(std::false_type{}, std::true_type{}, e);
        else for (uint32_t e = wave0; e < batch; e += nwaves) run_env(std::false_type{}, std::false_type{}, e);
    }
}
