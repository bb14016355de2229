// wf_tiled.hip - gfx950 kernels for grids that do not fit one wavefront (e.g. 256x256): the TILED family.
//
// Same rule as wf_kernels.hip (synchronous integer restatement of
// Simulation/forest_fire.py:85-106 + Simulation/environment.py:278-326).  One step of the TILED family
// (W % 4 == 0, W <= 256, at least 576 cells) is ONE launch of wf_step_tiled<.., SOLO> - a workgroup per environment
// does what the two kernels below do between them, see there - or, for jobs of thousands of waves in workgroups
// of more than four (1 024 x 256x256) and for observation tiles past a launch's LDS, two launches:
//   wf_tiled_prologue   one workgroup per environment: Agent.move / toggle / death (environment.py:116-171) with
//                       the dug cell and the agent_pos bytes written straight into the grid; the border-reachability
//                       flood when a dig may have cut a region off (run fill on linear row bitmaps, relax_row_bitmaps);
//                       the still-burning masks of the rows between the workgroups of the tick (it runs in place)
//   wf_step_tiled       the CA tick in place (+ the observation, when asked for; + the fresh rows of environments that
//                       are being auto-reset); per-env reductions by atomics, and the last of an environment's
//                       workgroups to arrive does World.RUNNING, reward, done, record
// (wf_reset: wf_big_reset_rec -> wf_big_reset_cells, and the generic family for every other shape: wf_generic.hip;
// what the two share - step modes, reset law, agent and reward logic of one environment, the flood - is wf_big_common.h)
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

#include "wf_big_common.h"

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
#ifndef WF_TILED_LAZY_BOARD
#define WF_TILED_LAZY_BOARD 1
#endif
struct TiledDst { __amdgpu_buffer_rsrc_t rsrc; uint32_t lane_off; };
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));

// Fused observation (World.get_state, environment.py:399-402) of the tiled family.  obs[b][x][y][k] is transposed
// against the planes' [y][x] and has 3 bytes per cell, so a workgroup stages the byte triples (agent_pos, type == fire,
// fire_mobility != inf) of its 64 rows row-major in LDS - 12 contiguous bytes per lane and row, conflict-free dword
// writes - and, once its four strips are done, thread x walks down column x: four rows' triples are four unaligned LDS
// dword reads, packed into the 12 contiguous bytes obs[x][4g..4g+3][0..2].  Needs H % 4 == 0 (4-byte aligned runs).
// The tile holds THREE BITS per cell - agent_pos, type == fire, fire_mobility != inf; the observation's bytes are 0 or 1
// (wf_upload refuses agent_pos bytes other than 0 and 1) - i.e. 12 bits per lane and row, a sixth of the byte triples it
// held until round 4: 128 rows of 128 cells are 8.5 KiB instead of 49, which is what lets a CU hold more than two such
// workgroups (and a 256x256 grid's tile fit a launch's LDS at all).
struct ObsTile { uint8_t *base; uint32_t W3, row0; };        // LDS tile [rows of the workgroup][pitch W3 = W/2 + 4 bytes], first row of the workgroup
#define WF_OBS_PITCH(W) ((W) / 2 + 4)                        // two bytes per quad; + 4: the rows of a column's piece in different banks

__device__ __forceinline__ void stage_obs_row(const ObsTile &ot, uint32_t r, uint32_t lane, const uint32_t (&c)[4])
{
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t t = c[j] & 0xffu;
        bits |= (((c[j] >> 24) & 1u) | (((FIRE_SET >> t) & 1u) << 1) | (t < (uint32_t)WF_DIRT ? 4u : 0u)) << (3 * j);
    }
    *reinterpret_cast<uint16_t *>(ot.base + (r - ot.row0) * ot.W3 + lane * 2u) = (uint16_t)bits;
}

// rows [row0, row0 + nrows) of the tile -> obs_e (this environment's (W, H, 3) block); nrows % 4 == 0.
// A column x of the observation is nrows * 3 CONTIGUOUS bytes: consecutive threads take consecutive 12-byte pieces (four
// rows' triples) of one column, so a wave's store covers a few whole columns instead of 64 pieces of 64 columns
// (4 096 x 64x64 with the observation: 65 us per step with a thread per column).
__device__ __forceinline__ void flush_obs_tile(const ObsTile &ot, uint32_t nrows, uint32_t W, uint32_t H, int8_t *obs_e)
{
    struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };
    const uint32_t G4 = nrows >> 2, items = W * G4;
    const bool pow2 = (G4 & (G4 - 1u)) == 0u;                    // (32, 64, 128, 256 rows: no division per piece)
    const uint32_t sh = 31u - (uint32_t)__builtin_clz(G4 | 1u);
    for (uint32_t i = threadIdx.x; i < items; i += blockDim.x) {
        const uint32_t x = pow2 ? i >> sh : i / G4, g = (i - x * G4) * 4u;
        const uint32_t qoff = (x >> 2) * 2u, bsh = (x & 3u) * 3u;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t b = (uint32_t)*reinterpret_cast<const uint16_t *>(ot.base + (g + (uint32_t)k) * ot.W3 + qoff) >> bsh;
            v[k] = (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14);                 // the cell's three bits -> its three bytes
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
    static constexpr u64 X0 = P == 1 ? 1ull : P == 2 ? 0x0000000100000001ull : P == 4 ? 0x0001000100010001ull : 0x0101010101010101ull;   // first lane of every segment
    static constexpr u64 XE = X0 << (LWP - 1u);                          // last lane of every segment
    uint32_t tr;             // rows per strip (TR; fewer where that gives a small grid's workgroup more waves: tiled_shape)
    uint32_t row_off;        // this lane's segment * tr: its row at sweep position r is r + row_off
    uint32_t ql;             // this lane's quad within the row
    bool col_ok;             // the quad exists (ql < W / 4)
    bool rivers;             // the context was created with make_rivers: WF_FIRE_BLOCKED cells can exist
    bool rare;               // ... or with the dig toggle too: cells that burn without being passable can exist
    u64 XL;                  // lane of the quad holding x == W-1, in every segment
};

// REACH: 0 no containment test this step; 1 burning cells are tested against the environment's reachability board;
// 2 (one workgroup per environment, board to be rebuilt): the row's masks are COLLECTED for the flood that follows the
// sweep - slot[0..3] passable before the tick, [4..7] burning after it, [8..11] burning-but-impassable before it.
enum { REACH_NONE = 0, REACH_TEST = 1, REACH_COLLECT = 2, COLLECT_WORDS = 12 };

template <bool STEP, int REACH, bool OBS, int P>
__device__ __forceinline__ void tiled_row(uint32_t r, uint32_t H, uint32_t LW, uint32_t ign, const Pack<P> &pk, const uint4 &cur,
                                          const u64 (&Sp)[4], const u64 (&Sc)[4], const u64 (&Sn)[4], const u64 *reach_e, u64 *slot,
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
        // make_rivers only (kernel-uniform branch): a blocked fire cell (13) burns out to blocked burnt (6), not 14
        if (pk.rivers) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] &= ~((c[j] << 2) & 8u);
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
    if constexpr (REACH == REACH_COLLECT) {
        const uint32_t t0[4] = { cur.x, cur.y, cur.z, cur.w };      // (idle lanes hold inert dirt: neither passable nor burning)
        u64 PS[4], Z[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int j = 0; j < 4; ++j) PS[j] = __ballot((t0[j] & 0xffu) < (uint32_t)WF_DIRT);
        if (pk.rare) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Z[j] = __ballot(blocked_burning(t0[j]));
        }
        if (__lane_id() == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { slot[j] = PS[j]; slot[4 + j] = FIRE[j]; slot[8 + j] = Z[j]; }
        }
    }
    // does a burning cell sit on a border-connected cell?  Only "one does" is asked for: the board rows (scalar loads the
    // sweep would wait for at every row) are fetched for rows that hold fire, until the wave has its first hit
    if (REACH == REACH_TEST && WF_TILED_LAZY_BOARD ? ((FIRE[0] | FIRE[1] | FIRE[2] | FIRE[3]) != 0 && !A.reach_hit) : REACH == REACH_TEST) {
        typedef u64 u64x4 __attribute__((ext_vector_type(4)));
        typedef const u64x4 __attribute__((address_space(4))) *const_u64x4_ptr;
        u64x4 rb = *(const_u64x4_ptr)(uintptr_t)(reach_e + (size_t)r * 4);
        if constexpr (P > 1) {      // the board rows of the other segments, moved to their lanes (rows past the grid: no fire bit meets them)
#pragma unroll
            for (int sgm = 1; sgm < P; ++sgm) {
                const u64x4 rs = *(const_u64x4_ptr)(uintptr_t)(reach_e + (size_t)min(r + (uint32_t)sgm * pk.tr, H - 1u) * 4);
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

template <bool STEP, int REACH, bool OBS, int P>
__device__ __forceinline__ void tiled_sweep(uint32_t r0, uint32_t r1, uint32_t H, uint32_t LW, uint32_t ign, const Pack<P> &pk,
                                            const uint4 *src, const TiledDst &dst, const ObsTile &ot, const uint4 &first, const uint4 &second,
                                            const u64 (&Sf)[4], const u64 (&Sabove)[4], const u64 (&Sbelow)[4], const u64 *reach_e, u64 *collect,
                                            TiledAcc &A)
{
    const uint4 inert = make_uint4(WF_DIRT, WF_DIRT, WF_DIRT, WF_DIRT);
    uint4 C[3] = { first, second, inert };
    u64 M[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { M[0][j] = Sf[j]; M[2][j] = Sabove[j]; M[1][j] = 0; }
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
                tiled_row<STEP, REACH, OBS, P>(rr, H, LW, ign, pk, C[k], M[kp], M[k], Sn, reach_e, collect + (rr - r0) * (uint32_t)COLLECT_WORDS, dst, ot, A);
#pragma unroll
                for (int j = 0; j < 4; ++j) M[kn][j] = Sn[j];
            }
        });
    }
}

// The border-reachability board of ONE environment, rebuilt by the environment's own (only) workgroup behind its sweep
// from the masks the sweep collected (REACH_COLLECT; `collect`: [wave][16 sweep positions][COLLECT_WORDS]) - no cell is
// read a second time.  `area`: LDS, (H * NW + 4 H) words: the linear reach bitmaps, then the linear passable bitmaps
// which the board in the tick's interleaved layout replaces once the relaxation is through.  Same board, bit for bit,
// as wf_tiled_prologue's: bit l of word [r][j] = cell (row r, x = 4l + j) is 4-connected to a passable border cell
// through passable cells, or burns impassably next to such a cell.  Returns, to every thread: bit 0 a cell that burns
// after the tick is on the board, bit 1 the relaxation converged (the board may be trusted by later steps).
template <int P>
__device__ __forceinline__ uint32_t solo_flood(const WfArgs &a, const uint32_t e, const uint32_t tr, const u64 *collect, u64 *area, uint32_t *changed)
{
    constexpr int NW = P >= 4 ? 1 : 4 / P;                       // words of a linear row bitmap: W <= 64 * NW
    constexpr uint32_t LWP = Pack<P>::LWP;
    const uint32_t WR = (uint32_t)P * tr;
    const uint32_t H = (uint32_t)a.H, W = (uint32_t)a.W, t = threadIdx.x, nt = blockDim.x;
    u64 *reach = area, *pass = area + H * NW, *board = pass;
    // mask `kind` of row r, component j, bit l = quad l of the row (a wave's ballots hold P rows, LWP lanes apart)
    auto row_mask = [&](uint32_t r, uint32_t kind, uint32_t j) -> u64 {
        const uint32_t rr = r % WR;
        const u64 m = collect[((r / WR) * (uint32_t)TR + rr % tr) * (uint32_t)COLLECT_WORDS + kind * 4u + j];
        return P == 1 ? m : (m >> ((rr / tr) * LWP)) & Pack<P>::SEG0;
    };
    // passable cells, LINEAR for the run fill (bit x % 64 of word [r][x / 64]), and the seeds: passable border cells
    for (uint32_t k = t; k < H * NW; k += nt) {
        const uint32_t r = k / NW, w = k % NW;
        const u64 lin = spread4(row_mask(r, 0, 0) >> (16u * w)) | (spread4(row_mask(r, 0, 1) >> (16u * w)) << 1) |
                        (spread4(row_mask(r, 0, 2) >> (16u * w)) << 2) | (spread4(row_mask(r, 0, 3) >> (16u * w)) << 3);
        u64 bm = r == 0 || r == H - 1 ? ~0ull : 0ull;
        if (w == 0) bm |= 1ull;
        if (w == (W - 1u) >> 6) bm |= 1ull << ((W - 1u) & 63u);
        pass[k] = lin;
        reach[k] = lin & bm;
    }
    __syncthreads();
    const bool converged = relax_row_bitmaps<NW>(pass, reach, H, H * W, t, nt, changed, [] { __syncthreads(); });
    // back to the tick's interleaved layout: component j of row r = every fourth bit from bit j
    for (uint32_t r = t; r < H; r += nt) {
        u64 il[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const u64 lin = reach[r * NW + w];
#pragma unroll
            for (int j = 0; j < 4; ++j) il[j] |= squeeze4(lin >> j) << (16 * w);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) board[r * 4 + j] = il[j];
    }
    __syncthreads();
    u64 *out = a.reach + (size_t)e * H * 4;
    const bool rare = (a.hot_flags & WF_HOT_RARE) != 0;
    bool hit = false;
    for (uint32_t k = t; k < H * 4u; k += nt) {
        const uint32_t r = k >> 2, j = k & 3u;
        u64 v = board[k];
        if (rare) {
            // Burning dirt (WF_DIRT_BURNING) and a fire cell over a river (WF_FIRE_BLOCKED) are impassable, yet the
            // reference's search starts FROM them: such a cell reaches the border iff one of its 4-neighbours does
            u64 nb = (r > 0 ? board[k - 4] : 0ull) | (r + 1 < H ? board[k + 4] : 0ull);
            nb |= j > 0 ? board[k - 1] : (board[k + 3] << 1);
            nb |= j < 3 ? board[k + 1] : (board[k - 3] >> 1);
            v |= nb & row_mask(r, 2, j);
        }
        out[k] = v;
        hit |= (v & row_mask(r, 1, j)) != 0;
    }
    return (__syncthreads_or(hit ? 1 : 0) ? 1u : 0u) | (converged ? 2u : 0u);
}

// Workgroup = nw waves (blockDim.x / 64: 4, fewer for grids that need fewer), wave = P strips of 16 rows side by side.
// SOLO (grids one workgroup covers, e.g. 64x64 and 128x128): the workgroup does the WHOLE step of its environment -
// Agent.move / toggle / death by its first thread (big_pre_env, what wf_tiled_prologue does for the larger grids), the
// tick, the reachability flood when the dig called for one (solo_flood, behind the sweep: only that workgroup is late,
// not a launch), World.RUNNING / reward / done from reductions in LDS: one launch per step, nothing but the cells, the
// record and the board touched in memory.
template <bool OBS, int P, bool SOLO, bool MULTI = false>
__global__ __launch_bounds__(SOLO ? WF_SOLO_MAX_WAVES * 64 : 256) void wf_step_tiled(const WfArgs a0, const uint8_t *mode, uint32_t *acc, const u64 *outer, const uint32_t tr)
{
    constexpr int MAXW = SOLO ? WF_SOLO_MAX_WAVES : 4;  // waves per workgroup (SOLO: as many as it takes to cover the grid)
    __shared__ u64 halo[MAXW][2][4];                 // [wave][first,last row of each of its segments][component]
    __shared__ uint16_t river_tab[MAXW][64];         // auto-reset with rivers: water column of each of a wave's rows
    __shared__ uint4 s_ev;                           // SOLO: the environment's record between big_pre_env and big_post_env
    __shared__ uint32_t s_word[2], s_red[MAXW][2];   // SOLO: step mode, the flood's "changed"; per wave: result bits, healthy cells
    // [SOLO with the containment test: the sweep's collected masks, nw * 16 * COLLECT_WORDS words]
    // [OBS: the observation's staging tile, [rows of the workgroup][W/2 + 4 bytes] | solo_flood's area, once the tile has been flushed]
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    constexpr uint32_t LWP = Pack<P>::LWP;                       // lanes per segment
    const uint32_t WR = (uint32_t)P * tr;                        // rows per wave (tr per strip: TR, fewer for small SOLO grids)
    const uint32_t lane = __lane_id();
    const uint32_t wv = rfl(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint32_t H = (uint32_t)a0.H, W = (uint32_t)a0.W, LW = W >> 2;
    const uint32_t gx = SOLO ? 1u : (H + nw * WR - 1u) / (nw * WR);     // workgroups per environment
    // (workgroups are dispatched in blockIdx order: a reversed step walks the batch from its far end)
    const uint32_t bid = a0.sweep_rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    const uint32_t e = bid / gx;
    const uint32_t r0 = ((bid % gx) * nw + wv) * WR;             // segment 0's first row; segment s: r0 + 16 s
    const uint32_t r1 = min(r0 + tr, H);                         // end of the sweep (segment 0's rows)
    const bool has_rows = r0 < H;
    Pack<P> pk;
    pk.tr = tr;
    pk.row_off = (P == 1 ? 0u : lane / LWP) * tr;
    pk.ql = P == 1 ? lane : lane % LWP;
    pk.col_ok = pk.ql < LW;
    pk.rivers = (a0.hot_flags & WF_HOT_RIVERS) != 0;
    pk.rare = (a0.hot_flags & WF_HOT_RARE) != 0;
    pk.XL = Pack<P>::X0 << (LW - 1u);
    const uint32_t lane_cell0 = pk.row_off * LW + pk.ql;         // this lane's quad index at sweep position 0 (+ r * LW)
    const uint32_t ign = (uint32_t)a0.ignite_count;
    u64 *const collect = reinterpret_cast<u64 *>(dyn_lds);
    uint8_t *const obs_lds = dyn_lds + (SOLO && (a0.hot_flags & WF_HOT_CONTAINMENT) ? nw * (uint32_t)(TR * COLLECT_WORDS * 8) : 0u);
    // One step of the environment (SOLO) / of this workgroup's rows; `a`: the launch's arguments as that step sees them
    auto one_step = [&](const WfArgs &a) {
        uint32_t md;
        if constexpr (SOLO) {
            if (threadIdx.x == 0) {
                uint4 ev;
                bool have_record;
                s_word[0] = big_pre_env(a, e, ev, have_record);
                s_word[1] = 0;
                s_ev = ev;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the touched bytes, for the other waves of this CU
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            md = rfl(s_word[0]);
        } else {
            md = mode[e];
        }
        const uint32_t wg_row0 = (bid % gx) * nw * WR;
        const ObsTile ot = { obs_lds, WF_OBS_PITCH(W), wg_row0 };
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
                if (__ballot(border_fire) && lane == 0) {
                    // (SOLO: through LDS and the first thread's plain store - an atomic is performed at the L2 and would leave
                    // this CU's L1 with the old record, which the step loop's next step reads)
                    if constexpr (SOLO) atomicOr(&s_word[1], 1u);
                    else atomicOr(reinterpret_cast<uint32_t *>(&a.envs[e]) + 1, (uint32_t)WF_F_BORDER);
                }
            }
            if constexpr (SOLO) {
                if (fresh) {
                    __syncthreads();
                    if (threadIdx.x == 0 && s_word[1]) reinterpret_cast<uint32_t *>(&a.envs[e])[1] |= (uint32_t)WF_F_BORDER;
                }
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
        uint4 first = inert, last = inert, second = inert;
        u64 Sf[4] = { 0, 0, 0, 0 }, Sl[4] = { 0, 0, 0, 0 };
        if (has_rows) {
            if (pk.col_ok && r0 + pk.row_off < H) first = src[r0 * LW];
            if (pk.col_ok && r1 - 1u + pk.row_off < H) last = src[(r1 - 1u) * LW];
            // (the sweep's second row, requested here: it arrives while the halos are exchanged)
            if (r0 + 1 < r1 && pk.col_ok && r0 + 1u + pk.row_off < H) second = src[(r0 + 1u) * LW];
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
                const u64 v = wv > 0 ? (halo[wv - 1][1][j] >> (LWP * (uint32_t)(P - 1))) : outer[(size_t)bid * 8 + j];
                Sp[j] = (((u64)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v)) & Pack<P>::SEG0;
            }
        }
        if (r0 + WR < H) {
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u64 v = wv + 1u < nw ? halo[wv + 1][0][j] : outer[(size_t)bid * 8 + 4 + j];
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
        u64 *const collect_w = collect + wv * (uint32_t)(TR * COLLECT_WORDS);
        auto sweep = [&](auto ST, auto RM) {
            tiled_sweep<decltype(ST)::value, decltype(RM)::value, OBS, P>(r0, r1, H, LW, ign, pk, src, dst, ot, first, second, Sf, Sp, Sb, reach_e, collect_w, A);
        };
        typedef std::integral_constant<int, REACH_NONE> RNone;
        typedef std::integral_constant<int, REACH_TEST> RTest;
        typedef std::integral_constant<int, SOLO ? REACH_COLLECT : REACH_TEST> RCollect;
        const bool collecting = SOLO && (md & MODE_FLOOD);                  // (MODE_FLOOD: only ever with MODE_PRECHECK)
        if (step) {
            if (collecting) sweep(std::true_type{}, RCollect{});
            else if (reach_test) sweep(std::true_type{}, RTest{});
            else sweep(std::true_type{}, RNone{});
        } else {
            if (collecting) sweep(std::false_type{}, RCollect{});
            else if (reach_test) sweep(std::false_type{}, RTest{});
            else sweep(std::false_type{}, RNone{});
        }
        A.bhit |= ((A.ign_x0 & Pack<P>::X0) | (A.ign_xl & pk.XL)) ? 1u : 0u;      // x == 0 / x == W-1 columns
        if (!SOLO && lane == 0) {
            // device-scope atomics are performed at the memory side (the per-XCD L2s are not coherent with each other);
            // the RETURNING forms, consumed, make the wave wait until they have been
            const uint32_t R = A.anyfire | (A.bhit << 1) | (A.reach_hit << 2);
            uint32_t o0 = 0, o1 = 0;
            if (R) o0 = atomicOr(&acc[4 * e], R);
            if (A.healthy) o1 = atomicAdd(&acc[4 * e + 1], A.healthy);
            asm volatile("" :: "v"(o0), "v"(o1));
        }
        }   // has_rows
        if (SOLO && lane == 0) {
            s_red[wv][0] = A.anyfire | (A.bhit << 1) | (A.reach_hit << 2);
            s_red[wv][1] = A.healthy;
        }
        __syncthreads();
        if constexpr (OBS) flush_obs_tile(ot, min(nw * WR, H - wg_row0), W, H, obs_e);
        if constexpr (SOLO) {
            uint32_t fl = 0;
            if (md & MODE_FLOOD) {
                if constexpr (OBS) __syncthreads();                   // the flood's boards take the observation tile's place
                fl = solo_flood<P>(a, e, tr, collect, reinterpret_cast<u64 *>(obs_lds), &s_word[1]);
            }
            if (threadIdx.x == 0) {
                uint32_t Aw = (fl & 1u) << 2, healthy = 0;
                for (uint32_t w = 0; w < nw; ++w) { Aw |= s_red[w][0]; healthy += s_red[w][1]; }
                // (a board that did not converge - never, with the H*W bound - would not be trusted later)
                big_post_env(a, e, md, Aw, healthy, (fl & 2u) ? (uint32_t)WF_F_REACH_VALID : 0u, s_ev);
            }
            return;
        }
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
    };
    if constexpr (MULTI) {
        // n_steps steps of the environment in ONE launch (SOLO only: the workgroup owns the environment, so a step's stores
        // have to reach no further than this CU, whose waves share one vector L1).  Between two steps: every store of the
        // step issued and acknowledged, the workgroup together, and the scalar cache dropped - the reachability board is
        // written by vector stores and read back through it.  (Dropping the vector L1 as well - buffer_inv sc1, which also
        // invalidates L2 lines - made a step of 4 096 x 64x64 66 us instead of 24.)
        static_assert(SOLO && !OBS, "the step loop is the one-launch form without the observation");
        const uint32_t n = (uint32_t)a0.n_steps, stride = (uint32_t)a0.step_stride;
        for (uint32_t st = 0; st < n; ++st) {
            WfArgs as = a0;
            const size_t off = (size_t)st * stride;
            as.actions = a0.actions ? a0.actions + off : nullptr;
            as.reward = a0.reward ? a0.reward + off : nullptr;
            as.done = a0.done ? a0.done + off : nullptr;
            as.update_mask = (a0.update_mask >> st) & 1ull;
            one_step(as);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_s_dcache_inv();
        }
    } else {
        one_step(a0);
    }
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
#ifdef WF_DBG_PRO
__device__ unsigned long long wf_dbg[16];
#define DBG_T(i) do { if (threadIdx.x == 0) tdbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DBG_T(i)
#endif
__global__ __launch_bounds__(256) void wf_tiled_prologue(const WfArgs a, uint8_t *mode, u64 *halo, uint32_t rows_per_wg)
{
#ifdef WF_DBG_PRO
    unsigned long long tdbg[8];
#endif
    DBG_T(0);
    extern __shared__ u64 lds[];                     // flood: [H*4] pass, [H*4] reach
    __shared__ u64 pm[4][8][4];                      // flood: the interleaved passable masks of a wave's current 8 rows
    __shared__ uint32_t s_mode, changed;
    const uint32_t e = a.sweep_rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
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
    DBG_T(1);
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
        DBG_T(2);
        const bool converged = relax_row_bitmaps<4>(pass, reach, H, H * W, threadIdx.x, 256u, &changed, [] { __syncthreads(); });
        // back to the tick's interleaved layout, in place: component j of row r = every fourth bit from bit j
        __syncthreads();
        DBG_T(3);
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
        if (a.hot_flags & WF_HOT_RARE) {
            // Burning dirt (WF_DIRT_BURNING) and a fire cell over a river (WF_FIRE_BLOCKED) are impassable, yet the
            // reference's search starts FROM them: such a cell reaches the border iff one of its 4-neighbours does.
            // One row per wave iteration: re-read the row, add those cells.
            for (uint32_t r = wv; r < H; r += 4u) {
                uint4 c = make_uint4(WF_DIRT, WF_DIRT, WF_DIRT, WF_DIRT);
                if (lane < LW) c = src[r * LW];
                const u64 Z[4] = { __ballot(blocked_burning(c.x)), __ballot(blocked_burning(c.y)),
                                   __ballot(blocked_burning(c.z)), __ballot(blocked_burning(c.w)) };
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
#ifdef WF_DBG_PRO
        __syncthreads();
        DBG_T(4);
        if (threadIdx.x == 0) {
            for (int i = 0; i < 4; ++i) atomicAdd(&wf_dbg[i], tdbg[i + 1] - tdbg[i]);
            atomicAdd(&wf_dbg[4], 1ull);
            atomicMax(&wf_dbg[5], tdbg[4] - tdbg[0]);
            atomicMax(&wf_dbg[6], tdbg[3] - tdbg[2]);
        }
#endif
    }
#ifdef WF_DBG_PRO
    if (threadIdx.x == 0 && e % 64u == 5u) { atomicAdd(&wf_dbg[8], 64ull); if (!(md & MODE_FLOOD)) atomicAdd(&wf_dbg[9], 64ull * (__builtin_amdgcn_s_memrealtime() - tdbg[0])); }
#endif
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

}  // namespace

#ifdef WF_DBG_PRO
extern "C" __attribute__((visibility("default"))) void wf_debug_read(unsigned long long *out)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_dbg), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = { 0 };
    hipMemcpyToSymbol(HIP_SYMBOL(wf_dbg), z, sizeof z);
}
#endif

// Shape of a wf_step_tiled launch: strips packed side by side per wave, waves per workgroup, workgroups per environment
struct TiledShape { int P, nw, rows_per_wg, gx, tr; };
static TiledShape tiled_shape(int H, int W, int tr = TR, int max_waves = 4)
{
    TiledShape t;
    const int LW = W / 4;
    // (eight strips side by side for grids up to 32 cells wide - every lane of a 32x32 grid's wave busy instead of half of
    // them - in the one-launch form only, with strips of at most 8 rows: a wave's river table holds 64 rows)
    t.P = LW <= 8 && max_waves > 4 ? 8 : LW <= 16 ? 4 : LW <= 32 ? 2 : 1;
    if (t.P == 8 && tr > 8) tr = 8;
    t.tr = tr;
    const int rows_per_wave = t.P * tr;
    t.nw = (H + rows_per_wave - 1) / rows_per_wave;
    if (t.nw > max_waves) t.nw = max_waves;
    t.rows_per_wg = t.nw * rows_per_wave;
    t.gx = (H + t.rows_per_wg - 1) / t.rows_per_wg;
    return t;
}
int wf_tiled_groups(int H, int W) { return tiled_shape(H, W).gx; }

// Rows per strip of the one-launch (SOLO) step of `batch` H x W grids: its workgroup has as many waves as it takes to
// cover the grid (up to WF_SOLO_MAX_WAVES: 16 waves of one 16-row strip for 256x256), 0 if that many do not.  The strips
// are as short as the waves allow (a 32x32 grid: four strips of 8 rows side by side in ONE wave - 8 sweep positions,
// every lane busy - not two of 16; 48x48: four of 12), and a job too small to give every SIMD a few waves takes twice
// the waves at half the strip (4 096 x 64x64: 24.3 instead of 29.9 us per step; a sweep is serial, so with one or two
// waves per SIMD it is their latency that shows) - but only then: a wave's fixed part (halo rows, barriers) is worth
// about five rows (16 384 x 64x64: 83 us per step with 16-row strips, 102 with 8).  The same count decides whether a
// grid that needs MORE than four waves is stepped this way at all: a 16-wave workgroup starts and ends as one - its
// waves wait for the first thread's agent logic and for the slowest strip - which a job with enough waves to fill the
// chip pays for (1 024 x 256x256, no-op actions: 93.5 us per step against 82.1 in the two-launch form whose workgroups
// are four waves; 256 x 256x256: 32 against 54).  `want` > 0: the caller's choice (WF_TILED_ROWS), honoured if a
// workgroup of such strips covers the grid.
int wf_tiled_solo_rows(int H, int W, int want, int batch)
{
    const TiledShape t = tiled_shape(H, W, TR, WF_SOLO_MAX_WAVES);
    if (t.gx != 1) return 0;
    if (want > 0 && want <= TR && tiled_shape(H, W, want, WF_SOLO_MAX_WAVES).gx == 1) return want;
    if (t.nw > 4 && (long long)batch * t.nw >= WF_TILED_SOLO_WAVES) return 0;
    int nw = t.nw;
    auto rows = [&](int n) { return (H + t.P * n - 1) / (t.P * n); };
    while (nw * 2 <= WF_SOLO_MAX_WAVES && (long long)batch * nw < WF_TILED_SOLO_WAVES && rows(nw * 2) >= WF_TILED_MIN_ROWS) nw *= 2;
    return rows(nw);
}

// dynamic LDS of a SOLO launch; it is one while this fits the 64 KiB a launch gets without asking (a 256x256 grid's
// observation tile does not: such a step takes the two-launch form)
static size_t solo_lds(const TiledShape &ts, int H, int W, bool containment, bool obs)
{
    const size_t collect_lds = containment ? (size_t)ts.nw * TR * COLLECT_WORDS * 8 : 0;
    const size_t flood_lds = containment ? ((size_t)H * (ts.P >= 4 ? 1 : 4 / ts.P) + (size_t)H * 4) * 8 : 0;
    const size_t obs_lds = obs ? (size_t)ts.rows_per_wg * WF_OBS_PITCH(W) + 16 : 0;
    return collect_lds + (obs_lds > flood_lds ? obs_lds : flood_lds);
}
int wf_tiled_solo_waves(int H, int W, int rows) { return rows > 0 ? tiled_shape(H, W, rows, WF_SOLO_MAX_WAVES).nw : 0; }
bool wf_tiled_solo_fits(int H, int W, int rows, bool containment, bool obs)
{
    return rows > 0 && solo_lds(tiled_shape(H, W, rows, WF_SOLO_MAX_WAVES), H, W, containment, obs) <= 64 * 1024 - 4096;
}

// wf_step_tiled can write the observation itself when the (x, y-run) blocks of obs are 4-byte aligned
bool wf_tiled_fused_obs(int H, int W) { return H % 4 == 0 && W <= 256; }

// solo_rows > 0 (wf_tiled_solo_rows): one workgroup of wf_step_tiled, strips of that many rows, covers an environment
// and the step is that kernel's SOLO form, one launch
hipError_t wf_launch_step_big(WfFamily fam, const WfArgs &a, uint8_t *mode, uint32_t *acc, u64 *halo, u64 auto_seed, hipStream_t s, bool per_env,
                              int solo_rows)
{
    const bool containment = (a.hot_flags & WF_HOT_CONTAINMENT) != 0;
    if (fam == WF_FAM_TILED && wf_tiled_solo_fits(a.H, a.W, solo_rows, containment, a.obs != nullptr)) {
        const TiledShape ts = tiled_shape(a.H, a.W, solo_rows, WF_SOLO_MAX_WAVES);
        if (ts.gx != 1) return hipErrorInvalidValue;
        const size_t lds = solo_lds(ts, a.H, a.W, containment, a.obs != nullptr);
        const dim3 grid((unsigned)a.batch), block(64u * (unsigned)ts.nw);
#define WF_SOLO_LAUNCH(OBS, P) hipLaunchKernelGGL((wf_step_tiled<OBS, P, true>), grid, block, lds, s, a, (const uint8_t *)nullptr, (uint32_t *)nullptr, (const u64 *)nullptr, (uint32_t)ts.tr)
#define WF_SOLO_LOOP(P) hipLaunchKernelGGL((wf_step_tiled<false, P, true, true>), grid, block, lds, s, a, (const uint8_t *)nullptr, (uint32_t *)nullptr, (const u64 *)nullptr, (uint32_t)ts.tr)
        if (a.n_steps > 1) {        // the step loop (the caller keeps the observation's step out of it)
            if (a.obs) return hipErrorInvalidValue;
            if (ts.P == 8) WF_SOLO_LOOP(8); else if (ts.P == 4) WF_SOLO_LOOP(4); else if (ts.P == 2) WF_SOLO_LOOP(2); else WF_SOLO_LOOP(1);
        } else
        if (a.obs) { if (ts.P == 8) WF_SOLO_LAUNCH(true, 8); else if (ts.P == 4) WF_SOLO_LAUNCH(true, 4); else if (ts.P == 2) WF_SOLO_LAUNCH(true, 2); else WF_SOLO_LAUNCH(true, 1); }
        else { if (ts.P == 8) WF_SOLO_LAUNCH(false, 8); else if (ts.P == 4) WF_SOLO_LAUNCH(false, 4); else if (ts.P == 2) WF_SOLO_LAUNCH(false, 2); else WF_SOLO_LAUNCH(false, 1); }
#undef WF_SOLO_LAUNCH
#undef WF_SOLO_LOOP
        return hipGetLastError();
    }
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
        const size_t obs_lds = a.obs ? (size_t)ts.rows_per_wg * WF_OBS_PITCH(a.W) + 16 : 0;      // fused World.get_state(): the caller checked wf_tiled_fused_obs(H, W)
#define WF_TILED_LAUNCH(OBS, P) hipLaunchKernelGGL((wf_step_tiled<OBS, P, false>), grid, block, obs_lds, s, a, mode, acc, (const u64 *)halo, (uint32_t)TR)
        if (a.obs) { if (ts.P == 4) WF_TILED_LAUNCH(true, 4); else if (ts.P == 2) WF_TILED_LAUNCH(true, 2); else WF_TILED_LAUNCH(true, 1); }
        else { if (ts.P == 4) WF_TILED_LAUNCH(false, 4); else if (ts.P == 2) WF_TILED_LAUNCH(false, 2); else WF_TILED_LAUNCH(false, 1); }
#undef WF_TILED_LAUNCH
        return hipGetLastError();
    }
    return wf_launch_step_generic(a, mode, acc, auto_seed, s, per_env);
}
