// wf_device.h - device helpers shared by the kernel files (wf_kernels.hip, wf_tiled.hip).
#pragma once
#include "wf_internal.h"
#include <utility>

namespace wfdev {

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool lanebit(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// compile-time loop: f(integral_constant<int, 0>), ..., f(integral_constant<int, N-1>)
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

// d = a + b + (lane's bit of the wave mask): one VALU instruction counts a neighbour set
__device__ __forceinline__ uint32_t addc(uint32_t a, uint32_t b, u64 carry)
{
    uint32_t d;
    asm("v_addc_co_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(carry) : "vcc");
    return d;
}
__device__ __forceinline__ uint32_t addc0(uint32_t a, u64 carry)
{
    uint32_t d;
    asm("v_addc_co_u32 %0, vcc, 0, %1, %2" : "=v"(d) : "v"(a), "s"(carry) : "vcc");
    return d;
}

// c + d in two independent 16-bit halves (v_pk_add_u16).  The fire step of a packed cell word adds 0xFF01 / 0xFF00
// (fuel byte -1, type byte +1 unless still burning) to the LOW half only: as a 32-bit add of 0xFFFFFF01 it would borrow a
// heat quantum from the high half whenever the fuel byte is 0 - a state no reset produces (fuel >= 1) but an upload can.
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t c, uint32_t d)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(c), "v"(d));
    return r;
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
// river of reset_map (environment.py:69-95) under the counter-based stream
// ----------------------------------------------------------------------------
// Same law as oracle/wf_oracle.c wfo_river (written independently): start column uniform, start row in {1,2,3}, one
// water cell per row, the column drifting by +-1 under the reference's acceptance test, the walk ending H - {1,2,3}
// rows down.  The walk may pass through the fire cell (the reference's does in 8-16 % of its resets): mark() is called
// for that cell too, and the reset makes it WF_FIRE_BLOCKED.  Every lane of a wave holds the same key, so the walk is
// wave-uniform control flow.  mark(x, y) is called for every river cell.
template <class F>
__device__ __forceinline__ void river_cells(const EnvKey &key, int W, int H, F &&mark)
{
    uint32_t n = 0x80000000u;
    const int fx = W / 2, fy = H / 2;                        // utility.py:61-64
    int x = (int)below(key.draw(n++), (uint32_t)W);          // :75
    int y = 1 + (int)below(key.draw(n++), 3u);               // :77
    while (y < H - (1 + (int)below(key.draw(n++), 3u))) {    // :79
        mark(x, y);                                          // :81-87
        const int ny = y + 1;
        int nx = x, tries = 0;
        bool ok = false;
        do {                                                 // :89-93
            const int s = (key.draw(n++) & 1u) ? 1 : -1;
            const int a = 1 + (int)below(key.draw(n++), 3u), b = 1 + (int)below(key.draw(n++), 3u);
            nx = x + s;
            ok = (a <= nx && nx < W - b) || (nx == fx && ny == fy);
        } while (!ok && ++tries < 64);
        if (!ok) nx = x + 1 <= W - 2 ? x + 1 : x - 1;        // never in practice: (2/3)^64
        x = nx;
        y = ny;
    }
}

// move table, 6 bits per action 0..4: (dx+1) | (dy+1)<<2 | is_move<<4; entry 4 = "no move"
// (N=(0,-1) S=(0,+1) E=(+1,0) W=(-1,0): environment.py:163-171)
constexpr uint32_t MOVE_LUT = (1u | 0u << 2 | 16u) | (1u | 2u << 2 | 16u) << 6 | (2u | 1u << 2 | 16u) << 12 |
                              (0u | 1u << 2 | 16u) << 18 | (1u | 1u << 2) << 24;

// cell types dig() turns into dirt on a MOVE (environment.py:151-152: not burning; :127: not dirt already):
// grass, burnt, blocked burnt.  Bit t of the set = type t.  toggle_digging() digs under the agent without the
// "not burning" guard: DIG_SET | FIRE_SET.
constexpr uint32_t DIG_SET = (1u << WF_GRASS) | (1u << WF_BURNT) | (1u << WF_BURNT_BLOCKED);
// cell types World.is_burning() is true for (type == fire, environment.py:249-251): plain fire and the fire cell a
// river was drawn under (WF_FIRE_BLOCKED).  An agent that steps on one dies; observation plane 1 shows them.
constexpr uint32_t FIRE_SET = (1u << WF_FIRE) | (1u << WF_FIRE_BLOCKED);
// a cell that burns but blocks the containment search (fire_mobility inf): WF_DIRT_BURNING (5) or WF_FIRE_BLOCKED (13).
// The reference's A* still STARTS from such a cell: it reaches the border iff one of its 4-neighbours does.
__device__ __forceinline__ bool blocked_burning(uint32_t type) { return (type & 7u) == 5u; }
// The fire step of a packed cell word: low half + 0xFF00 (fuel - 1) while the cell keeps burning, + 0xFF01 (type + 1:
// fire -> burnt, burning dirt -> blocked burnt) when it burns out.  A blocked fire cell (13) burns out to blocked burnt
// (6, not 14): 8 less.  Only the instantiations that know those codes (RARE) pay the two instructions.
template <bool RARE>
__device__ __forceinline__ uint32_t burn_out_step(uint32_t c)
{
    if constexpr (RARE) return 0x0000FF01u - (c & 8u);
    else return 0x0000FF01u;
}

// "burning" test on a packed cell word.  With the dig toggle / rivers a burning cell is WF_FIRE, WF_DIRT_BURNING or
// WF_FIRE_BLOCKED: (type & 3) == 1.
// Without it (RARE = false) those codes cannot arise and the plain byte compare is one instruction (SDWA) and no
// temporary register.
template <bool RARE>
__device__ __forceinline__ bool burning(uint32_t c)
{
    if constexpr (RARE) return (c & 3u) == 1u;
    else return (c & 0xffu) == (uint32_t)WF_FIRE;
}

typedef const WfCold __attribute__((address_space(4))) *cold_ptr;

}  // namespace wfdev
