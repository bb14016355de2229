// wf_wave_rows.h - single-step kernel, rows layout (several environments per wave iteration)
// Included by wf_kernels.hip inside its anonymous namespace (one translation unit: the launchers there
// instantiate these templates); not a stand-alone header.

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

// The same N dwords through a buffer store that carries cache-policy bits (aux 16 = sc1: agent scope, i.e. written
// THROUGH the XCD's L2).  A plain store leaves the line dirty in L2, and everything still dirty when the kernel ends
// is written back at the kernel boundary, where nothing overlaps it (MI355X_MICROARCH.md, row "boundary":
// + B / 6 TB/s for B dirty bytes - ~27 MB per 65 536 x 14x14 step); write-through moves that traffic under the kernel.
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef uint32_t v3u32 __attribute__((ext_vector_type(3)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
template <int N, int AUX>
__device__ __forceinline__ void store_dwords_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t off, const uint32_t *r)
{
    if constexpr (N >= 4) {
        __builtin_amdgcn_raw_buffer_store_b128(v4u32{ r[0], r[1], r[2], r[3] }, rsrc, (int)off, 0, AUX);
        if constexpr (N > 4) store_dwords_buf<N - 4, AUX>(rsrc, off + 16u, r + 4);
    } else if constexpr (N == 3) {
        __builtin_amdgcn_raw_buffer_store_b96(v3u32{ r[0], r[1], r[2] }, rsrc, (int)off, 0, AUX);
    } else if constexpr (N == 2) {
        __builtin_amdgcn_raw_buffer_store_b64(v2u32{ r[0], r[1] }, rsrc, (int)off, 0, AUX);
    } else if constexpr (N == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(r[0], rsrc, (int)off, 0, AUX);
    }
}

// World.reset() of ONE environment of the iteration into the rows layout (same stream and
// same law as reset_regs).  Every lane computes the cells "as if" it belonged to that
// environment (y, xs = its row and first column); the lanes of HM (`mine`) keep them.
// Returns the spawn cell in (s.ax, s.ay); writes the 4-cells-per-lane closed-form board too.
template <int W, int H, int S, bool RARE>
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
    uint32_t water = 0;                                              // bit j: slot j of this lane is a river cell
    bool rivers = false;
    if constexpr (RARE) {
        rivers = cold->p.make_rivers != 0;                           // reset_map :69-95
        if (rivers)
            river_cells(key, W, H, [&](int x, int yy) {
                const uint32_t d = (uint32_t)(yy * W + x) - q0;
                water |= d < (uint32_t)CPL ? 1u << d : 0u;
            });
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const uint32_t q = q0 + (uint32_t)j;
        int32_t fuel = fuel_lo;
        if (span > 1u) fuel += (int32_t)below(key.draw(16u + q), span);
        uint32_t ty = ((water >> j) & 1u) ? (uint32_t)WF_WATER : (uint32_t)WF_GRASS, heat = 0;
        // set_fire_to :233-246; under a river the cell keeps its infinite fire mobility (:85)
        if (q == qc) { ty = ty == (uint32_t)WF_WATER ? (uint32_t)WF_FIRE_BLOCKED : (uint32_t)WF_FIRE; heat = WF_HEAT_FORCED; }
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
    // (with a river the closed-form boards do not hold: the first containment test of the episode floods)
    s.fl = WF_F_ALIVE | WF_F_VISIBLE | WF_F_DIGGING | WF_F_RUNNING | (rivers ? 0u : (uint32_t)(WF_F_REACH_VALID | WF_F_REACH_ROWS)) |
           (onborder ? WF_F_BORDER : 0u);
    s.episode += 1;
    s.tick = 0;
}

// Issue priority by progress.  The SIMD arbitrates VALU issue between its resident waves by priority, then AGE
// (MI355X_MICROARCH.md, "Two waves per SIMD"): at equal priority the oldest wave runs nearly unimpeded and the
// youngest gets the leftovers.  With one generation of waves per launch (65 536 envs = 8 192 waves = every wave slot of
// the chip) that turns the eight waves of a SIMD into a staircase - tools/wave_timeline.py: first wave done after
// 8.5 us, last after 25, the tail of each SIMD running one or two waves deep where a single wave is bound by its own
// dependency chain (~12 cycles per instruction).  Experiment (WF_PROGRESS_PRIO=1): a wave LOWERS its priority as it gets
// ahead (phase A and the first quarter of its iterations at 3, then 2, 1, 0).  It does what it says - all eight waves of
// a SIMD then end within 3 us of each other - but the SIMD is not done earlier: eight waves in the SAME phase compete
// for the same pipe, the staircase had them in different phases.  23.7 vs 25.3 us per step as one launch, but 22.3 vs
// 20.3 with the two half-batch launches, 34.9 vs 30.8 at 125 000 envs, 66.7 vs 60.3 at 262 144: off by default.
template <bool ON>
__device__ __forceinline__ void set_progress_prio(uint32_t done, uint32_t total)
{
    if constexpr (!ON) return;
    const uint32_t q = (done * 4u) / total;                 // wave-uniform
    if (q == 0u) __builtin_amdgcn_s_setprio(3);
    else if (q == 1u) __builtin_amdgcn_s_setprio(2);
    else if (q == 2u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// Waves per workgroup of the private (not workgroup-shared) variant.  The dispatcher hands out workgroups, so fewer,
// larger ones mean fewer dispatches per launch (8 192 waves = 2 048 / 1 024 / 512 workgroups for 4 / 8 / 16).
#ifndef WF_ROWS_WPB
#define WF_ROWS_WPB 4
#endif
// MULTI = true: an n_steps call served by one launch (the step loop below); false: exactly one step, the loop folds away.
// RES (with MULTI): RESIDENT step loop.  The wave copies its environments' cells into an LDS mirror once, runs all its
// steps on the mirror (the agent phase reads and patches bytes there, the tick loads and stores row pieces there, the
// records travel from phase C to the next phase A through LDS) and writes the cells back once at the end.  Per step
// that leaves stores of reward / done / record and one action load - no round trip to L2 on the critical path, where
// the plain step loop has four (record, type bytes, cells, cells again after the agent's byte stores).  MEASURED, and
// therefore opt-in (WF_RESIDENT=1): it is no faster - 4 096 x 10x10 x 16 ticks 42.1 us against 42.5 through L2 (and 37.7 on
// the one-environment-per-wave kernel), 8 192: 48.3 against 49.8, 65 536 x 14x14: 15.6 against 15.0 us per step.  A wave of
// the step loop is bound by instruction issue along its own dependency chain (~450 instructions per step at 2
// environments per wave, 6 cycles each with two waves per SIMD), not by the L2 round trips the mirror removes.
#ifndef WF_RES_MAXG
#define WF_RES_MAXG 8        // environments per wave of the resident loop (LDS mirror: WF_RES_MAXG x W x H x 4 bytes per wave)
#endif
template <int W, int H, int S, bool SHARED, bool MULTI, bool RARE, bool RES = false, bool PRIO = false, bool REV = false>
// Waves per SIMD: minimum = maximum = 8.  (A minimum of 7 gives the register allocator 94 scalar registers instead of
// 80 and removes a third of the scalar spills to VGPR lanes, and the compiler still reports 8 waves per SIMD - but a
// wave's 94 SGPRs are allocated as 112, 800 / 112 = 7: tools/wave_timeline.py showed exactly one workgroup in eight
// starting 10 us late.  Same step time either way; 8 resident waves is what the launch geometry assumes.)
__global__ __launch_bounds__(SHARED ? 512 : 64 * WF_ROWS_WPB)
__attribute__((amdgpu_waves_per_eu(MULTI ? WF_LB_ROWS_MULTI : WF_ROWS_WAVES_MIN, MULTI ? WF_LB_ROWS_MULTI : WF_LB_ROWS), amdgpu_num_sgpr(102))) void wf_step_rows(const WfArgs a0)
{
    using R = RG<W, H, S>;
    constexpr int CELLS = W * H, CPL = R::CPL, LPE = R::LPE, PITCH = R::PITCH, EPW = R::EPW;
    constexpr uint32_t ENV_BYTES = CELLS * 4, OBS_BYTES = CELLS * 3;
    const uint32_t lane = __lane_id();
    const uint32_t batch = (uint32_t)a0.batch;
    const uint32_t G = (uint32_t)a0.group;                   // environments per wave: multiple of EPW, <= 32
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
        // (REV: this launch walks its environments from the far end - states past the Infinity Cache, see wave_step_issue; an
        // instantiation of its own: as a launch argument it cost the headline kernel three more spilled scalar registers, 1.5 %)
        const uint32_t bid = REV ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
        e0 = rfl(bid * (uint32_t)WF_ROWS_WPB + wv) * G;
        if (e0 >= batch) return;
        ng = min(G, batch - e0);
        A0 = e0;
        na = ng;
        gi0 = wv * 32u;                                      // private slice of the LDS arrays (G <= 32)
    }
    // n_steps > 1: the wave simply steps its environments again (each wave owns its environments, so nothing has
    // to be synchronised across waves); this spares a kernel boundary - 2.8 us of dispatch - per step and launch.
    static_assert(!(SHARED && MULTI), "the workgroup-shared variant steps once");
    static_assert(!RES || (MULTI && WF_STORE_AUX == 0), "the resident loop is a step loop with plain stores");
    // ---- RES: the wave's cells -> LDS mirror; from here on "the planes" of this wave are the mirror
    extern __shared__ uint32_t sMirror[];                    // RES: [waves per workgroup][G * CELLS], sized by the launch
    uint32_t *mirror = sMirror + (RES ? wv * G * (uint32_t)CELLS : 0u);
    uint32_t *planes_wave = nullptr;                         // (generic pointer such that planes_wave + e * CELLS is env e's mirror)
    if constexpr (RES) {
        const uint32_t *src = a0.planes + (size_t)e0 * CELLS;
        for (uint32_t i = lane; i < ng * (uint32_t)CELLS; i += 64u) mirror[i] = src[i];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        planes_wave = mirror - (size_t)e0 * CELLS;
    }
    // (round 4: per launch - the PRIO instantiation serves a launch that steps the whole batch alone with one generation of
    // waves: 65 536 envs 24.0 -> 22.6 us, 32 768 16.3 -> 15.6; the part launches of a split step run without it)
    // (an instantiation of its own, PRIO: as a run-time flag it cost the part launches of a split step 1.5 % - one more live
    // wave-uniform value in a kernel that already spills scalar registers)
    constexpr bool prio_on = !SHARED && !MULTI && (WF_PROGRESS_PRIO || PRIO);
    if constexpr (prio_on) __builtin_amdgcn_s_setprio(3);
#ifdef WF_WAVE_TIMING   // measurement build (tools/wave_timeline.py): 100 MHz wall clock at the phase borders of every wave
    const u64 wt0 = wall_clock64();
    u64 wtA = 0, wtB = 0;
    uint32_t wt_reset = 0, wt_flood = 0, wt_flood_it = 0;
    u64 wt_it[4] = { 0, 0, 0, 0 };
#endif
    const uint32_t nst = MULTI ? (uint32_t)a0.n_steps : 1u;
    for (uint32_t st = 0; st < nst; ++st) {
    WfArgs a = a0;
    {
        const size_t so = (size_t)st * (size_t)a0.step_stride;          // row of the (n_steps, context batch) arrays
        if (a.actions) a.actions += so;
        if (a.reward) a.reward += so;
        if (a.done) a.done += so;
        if constexpr (RES) a.planes = planes_wave;
        a.update_mask = a0.update_mask >> st;
        if (st + 1u < nst) a.obs = nullptr;                             // the observation of the final state only
        if (st) {
            // boards this wave rewrote with vector stores in the previous step are read through the scalar cache
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_s_dcache_inv();
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
    const uint32_t ign = (uint32_t)a.ignite_count;
    const bool do_update = (a.update_mask & 1ull) != 0;
    char *planes_b = reinterpret_cast<char *>(a.planes);
    // (this launch's planes; 32-bit byte offsets - wf_create keeps a context's planes below 4 GiB)
    const __amdgpu_buffer_rsrc_t planes_rsrc = __builtin_amdgcn_make_buffer_rsrc(a0.planes, 0, (int)(batch * ENV_BYTES), 0x00020000);   // (WF_STORE_AUX builds only)
    __shared__ uint32_t sS[7][SHARED ? 64 : 32 * WF_ROWS_WPB];   // ax|ay<<16, fl, episode, tick, Pv, Rv, counter
    enum { S_POS = 0, S_FL = 1, S_EP = 2, S_TICK = 3, S_P = 4, S_R = 5, S_CNT = 6 };

    // ================= phase A: lanes = environments =================
    const bool lane_env = lane < na;
    const uint32_t ek = A0 + (lane_env ? lane : 0u);
    const uint32_t ai = SHARED ? lane : gi0 + (lane_env ? lane : 0u);   // LDS slot of this env-lane
    if (!SHARED || wv == 0) {
        // the one wave that runs phase A for its workgroup does so at issue priority 1: the other waves of the workgroup wait
        // for it at the barrier (65 536 x 14x14, three parts: 19.37 against 19.52-19.71 us per step; priority 3: 19.46-19.53)
        if constexpr (SHARED) __builtin_amdgcn_s_setprio(WF_SHARED_A_PRIO);
        EnvRegs S0;
        if (RES && st > 0u) {       // the record phase C of the last step left in LDS
            const uint4 pre = make_uint4(sS[S_POS][ai], sS[S_FL][ai], sS[S_EP][ai], sS[S_TICK][ai]);
            S0 = group_phase_a<W, H, 8, RARE, true>(a, lane_env, ek, do_update, pre);
        } else {
            S0 = group_phase_a<W, H, 8, RARE>(a, lane_env, ek, do_update);
        }
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
        if constexpr (SHARED) __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (SHARED) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

#ifdef WF_WAVE_TIMING
    wtA = wall_clock64();
#endif
    // ================= phase B: EPW environments at a time, lanes = row pieces =================
    const uint32_t hh = lane / (uint32_t)PITCH, l = lane - hh * (uint32_t)PITCH;
    const bool cell_lane = hh < (uint32_t)EPW && l < (uint32_t)LPE;
    const uint32_t y = l / (uint32_t)S, xs = (l - y * (uint32_t)S) * (uint32_t)CPL;
    // bit j: slot j of this lane is a border cell (needed by the rare reset path only)
    uint32_t bgeo = (y == 0u || y == (uint32_t)(H - 1)) ? (1u << CPL) - 1u
                  : ((xs == 0u ? 1u : 0u) | (xs + (uint32_t)CPL == (uint32_t)W ? 1u << (CPL - 1) : 0u));
    uint32_t ob = hh * OBS_BYTES + (xs * H + y) * 3u;         // obs[x][y][k] image of the iteration's environments
    asm volatile("" : "+v"(bgeo), "+v"(ob));
    const uint32_t K_FIRE_STEP = 0x0000FF01u;                // low half: fuel byte -1, type byte +1 (fire -> burnt); pk_add_u16
    constexpr int OBS_PAD = (EPW * OBS_BYTES + 15) & ~15;
    __shared__ __attribute__((aligned(16))) uint8_t sObs[SHARED ? 8 : WF_ROWS_WPB][OBS_PAD];
    uint8_t *obs_lds = sObs[wv];
    typedef const u64 __attribute__((address_space(4))) *const_u64_ptr;

    if (cell_lane && ng) {
        // byte offset of this lane's cells in the current iteration; lanes whose environment does not
        // exist (ragged tail) shadow environment 0 of the iteration for loads and sit out otherwise
        uint32_t off = e0 * ENV_BYTES + l * (uint32_t)(CPL * 4);
        uint32_t nxt[CPL];
        load_dwords<CPL>(planes_b + (off + (hh < ng ? hh * ENV_BYTES : 0u)), nxt);
        // board of the iteration's group (context-level group index), advanced with the loop
        u64 *board = a.reach_rows + (size_t)((a.env_base + e0) / (uint32_t)EPW) * CPL;
        for (uint32_t kb = 0; kb < ng; kb += (uint32_t)EPW, off += (uint32_t)EPW * ENV_BYTES, board += CPL) {
            set_progress_prio<prio_on>(kb, G);
            if (kb + hh < ng) {
                const uint32_t eoff = off + hh * ENV_BYTES;
                uint32_t c[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) c[j] = nxt[j];
                uint32_t chg = 0;                                   // nonzero: this lane's row piece changes in this step
                load_dwords<CPL>(planes_b + (eoff + (kb + (uint32_t)EPW + hh < ng ? (uint32_t)EPW * ENV_BYTES : 0u)), nxt);
                const uint32_t idx = gi0 + kb + hh;          // LDS slot of this lane's environment
                const uint32_t Pl = sS[S_P][idx];
                const u64 VM = __ballot(true);
                const u64 RUNM = __ballot((Pl & PB_RUNNING) != 0u);
                // border-reachability board of this group of EPW environments.  Always loaded: with the containment
                // test off no environment carries the precheck bit, so whatever the board holds is never looked at.
                u64 RBG[CPL];
                bool rb_dirty = false;
#pragma unroll
                for (int j = 0; j < CPL; ++j) RBG[j] = ((const_u64_ptr)(uintptr_t)board)[j];
                u64 WASRESET = 0;
                if (RUNM != VM && (a.hot_flags & WF_HOT_AUTO_RESET)) {
                    // ---- rare: finished environments start a new episode instead of stepping
                    WASRESET = VM & ~RUNM;
#ifdef WF_WAVE_TIMING
                    wt_reset += (uint32_t)__popcll(WASRESET & R::WRITERS);
#endif
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
                            reset_rows<W, H, S, RARE>(cold, cc->p.seed, e_pin + a.env_base, y_pin, xs_pin, bgeo_pin, R::hm(h), hh_pin == (uint32_t)h, c, s,
                                                a.reach + (size_t)e_pin * 4, lane == (uint32_t)(h * PITCH));
                            chg |= hh == (uint32_t)h ? 1u : 0u;
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
                        const u64 F = __ballot(burning<RARE>(c[j])) & RUNM;               // burning: fire or burning dirt
                        STILL[j] = F & __ballot((int32_t)(int8_t)(c[j] >> 8) > 1);     // fuel - 1 > 0
                        const bool st = lanebit(STILL[j]);
                        sv[j] = st ? 1u : 0u;
                        const uint32_t d = st ? (K_FIRE_STEP - 1u) : burn_out_step<RARE>(c[j]);
                        const uint32_t dd = lanebit(F) ? d : 0u;
                        c[j] = pk_add_u16(c[j], dd);
                        chg |= dd;
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
                        const uint32_t add = lanebit(GR) ? min(cnt, room) << 16 : 0u;  // heat saturates (room >= 1 on grass)
                        c[j] = addc(c[j], add, IGN);                                   // grass -> fire: +1 through the carry-in
                        chg |= add;                                                    // (an ignition always comes with heat)
                        IGNANY |= IGN;
#ifdef WF_EXP_EXTRA_SALU     // timing experiment: WF_EXP_EXTRA_SALU extra scalar instructions per cell slot (results unused)
#pragma unroll
                        for (int x = 0; x < WF_EXP_EXTRA_SALU; ++x) { u64 dm; asm volatile("s_and_b64 %0, %1, %1" : "=s"(dm) : "s"(IGNANY) : "scc"); }
#endif
#ifdef WF_EXP_EXTRA_VALU     // timing experiment: WF_EXP_EXTRA_VALU extra vector compares per cell slot (results unused, state unchanged)
#pragma unroll
                        for (int x = 0; x < WF_EXP_EXTRA_VALU; ++x) { u64 dm; asm volatile("v_cmp_eq_u32_e64 %0, %1, %1" : "=s"(dm) : "v"(c[j])); }
#endif
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
                        const u64 FJ = __ballot(burning<RARE>(c[j])) & RUNM;
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
#ifdef WF_WAVE_TIMING
                    wt_flood += 1;
                    {   // count the relaxation sweeps of this flood (same loop as flood_rows, on a copy)
                        u64 T[CPL];
#pragma unroll
                        for (int j = 0; j < CPL; ++j) T[j] = NB[j];
                        for (int it = 0; it < W * H; ++it) {
                            u64 any = 0;
                            static_for<CPL>([&](auto J) {
                                constexpr int j = decltype(J)::value;
                                const u64 add = ((T[j] << S) | (T[j] >> S) | rg_lf<W, H, S, j>(T) | rg_rt<W, H, S, j>(T)) & PASS[j] & ~T[j];
                                T[j] |= add;
                                any |= add;
                            });
                            wt_flood_it += 1;
                            if (!any) break;
                        }
                    }
#endif
                    flood_rows<W, H, S>(NB, PASS);
                    if constexpr (RARE) {
                        // burning dirt / a blocked fire cell reaches the border through its 4-neighbours (see add_blocked_burning)
                        u64 ZB[CPL], ADD[CPL];
#pragma unroll
                        for (int j = 0; j < CPL; ++j) ZB[j] = __ballot(blocked_burning(c[j]));
                        static_for<CPL>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            ADD[j] = ((NB[j] << S) | (NB[j] >> S) | rg_lf<W, H, S, j>(NB) | rg_rt<W, H, S, j>(NB)) & ZB[j];
                        });
#pragma unroll
                        for (int j = 0; j < CPL; ++j) NB[j] |= ADD[j];
                    }
                    HITS = 0;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        RBG[j] = (RBG[j] & ~NEEDM) | (NB[j] & NEEDM);
                        HITS |= __ballot(burning<RARE>(c[j])) & RUNM & RBG[j];
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
                // most cells do not change in a tick: a lane whose row piece is unchanged writes nothing
#ifdef WF_WAVE_TIMING
                if ((kb >> 1) < 4u) wt_it[kb >> 1] = wall_clock64();
#endif
                if (chg || !WF_SKIP_CLEAN) {
                    if constexpr (WF_STORE_AUX != 0) store_dwords_buf<CPL, WF_STORE_AUX>(planes_rsrc, eoff, c);
                    else store_dwords<CPL>(planes_b + eoff, c);
                }
                if (a.obs) {
                    // World.get_state() of the iteration's environments: byte triples into the LDS image at
                    // their transposed offsets, then the image leaves as 16-byte stores
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const uint32_t t = c[j] & 0xffu, o = ob + (uint32_t)(j * H * 3);
                        obs_lds[o] = (uint8_t)(c[j] >> 24);                       // agent_pos
                        obs_lds[o + 1u] = (uint8_t)((FIRE_SET >> t) & 1u);        // type == fire (fire, blocked fire)
                        obs_lds[o + 2u] = (uint8_t)((7u >> t) & 1u);              // not dirt / water / blocked
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

#ifdef WF_WAVE_TIMING
    wtB = wall_clock64();
#endif
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
        const uint4 rec = group_phase_c<W, H, 8>(a, S1, Rv, lane_env, ek, do_update);
        if constexpr (RES) {
            if (lane_env) {
                sS[S_POS][ai] = rec.x;
                sS[S_FL][ai] = rec.y;
                sS[S_EP][ai] = rec.z;
                sS[S_TICK][ai] = rec.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
#ifdef WF_WAVE_TIMING
    if (a0.mask && lane == 0) {      // (the reset mask pointer is unused by step launches: the timing buffer rides on it)
        __builtin_amdgcn_s_waitcnt(0);
        u64 *rec = reinterpret_cast<u64 *>(const_cast<uint8_t *>(a0.mask)) + (size_t)((a0.env_base + e0) / G) * 10;
        rec[6] = wt_it[0]; rec[7] = wt_it[1]; rec[8] = wt_it[2]; rec[9] = wt_it[3];
        uint32_t xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        rec[5] = hwid;
        rec[0] = wt0; rec[1] = wtA; rec[2] = wtB; rec[3] = wall_clock64();
        rec[4] = (u64)(wt_reset & 0xffu) | ((u64)(wt_flood & 0xffu) << 8) | ((u64)(wt_flood_it & 0xffffu) << 16) | ((u64)(xcc & 0xfu) << 32);
    }
#endif
    }   // steps
    if constexpr (RES) {            // the mirror -> the planes
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t *dst = a0.planes + (size_t)e0 * CELLS;
        for (uint32_t i = lane; i < ng * (uint32_t)CELLS; i += 64u) dst[i] = mirror[i];
    }
}
