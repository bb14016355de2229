/*
 * wf_oracle.c - CPU parity oracle (TEST INFRASTRUCTURE ONLY; see wf_oracle.h).
 *
 * Restates, in plain C and with the reference's own float64 arithmetic, the hot
 * path of dashdeckers/Wildfire-Control-Python:
 *     Simulation/forest_fire.py:30-49   ForestFire.step
 *     Simulation/forest_fire.py:85-106  ForestFire.update  (the CA tick)
 *     Simulation/environment.py:99-171  Agent
 *     Simulation/environment.py:186-402 World (reset, set_fire_to, heat, fuel, reward, state)
 *     Simulation/utility.py:8-78        circle_points / spawn position
 * Every function cites the lines it follows.  Nothing here is copied: the
 * reference is Python operating on a float64 (W,H,9) ndarray; this is C operating
 * on per-layer scratch arrays decoded from / encoded to the int8 boundary format.
 */
#include "wf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ params */

void wfo_default_params(wfo_params *p)
{
    memset(p, 0, sizeof(*p));
    p->wind_speed = 0.54;          /* constants.py:39 */
    p->wind_x = 0.0;
    p->wind_y = 0.0;
    p->heat = 0.3;                 /* utility.py:98  */
    p->threshold = 3.0;            /* utility.py:100 */
    p->death_penalty = -1000.0;    /* constants.py:32 */
    p->contained_bonus = 1000.0;   /* constants.py:33 */
    p->default_reward = -1.0;      /* constants.py:34 */
    p->allow_dig_toggle = 0;       /* constants.py:46 */
    p->containment = 1;
    p->fuel_lo = 20;               /* utility.py:99  */
    p->fuel_hi = 20;
    p->ignite_ppm = 0;
    p->wind_mode = 0;
    p->wind_code = WFO_WIND_CODE(0, 0, 0);
    p->wind_speeds[0] = 0.0;       /* environment.py:189 */
    p->wind_speeds[1] = 0.7;
    p->wind_speeds[2] = 0.85;
    p->wind_speeds[3] = 0.0;
}

/* ------------------------------------------------------- counter-based RNG */
/* Build-defined stream (the reference's numpy legacy RNG is not reproduced on a
 * GPU).  splitmix64 finalizer for the (seed, env, episode) key,
 * two rounds of a 32-bit integer hash per draw.  The HIP reset kernel implements
 * the same arithmetic independently. */

static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

uint32_t wfo_rng_u32(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t draw)
{
    uint64_t k = mix64(mix64(seed ^ (env_id * 0x9E3779B97F4A7C15ull)) + (uint64_t)episode);
    uint32_t k0 = (uint32_t)k, k1 = (uint32_t)(k >> 32);
    return hash32(k0 ^ hash32(draw + k1));
}

/* uniform integer in [0, n) from one 32-bit draw (multiply-high) */
static uint32_t below(uint32_t r, uint32_t n)
{
    return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32);
}

/* ------------------------------------------------------------------ river */

/* reset_map's river (environment.py:69-95).  Draws come from the river domain of the env's stream: index
 * 0x80000000 + n.  The walk may pass through the fire cell (the acceptance test :91-92 even favours it). */
int wfo_river(int H, int W, uint64_t seed, uint64_t id, uint32_t ep, uint8_t *water)
{
    uint32_t n = 0x80000000u;
    int fx = W / 2, fy = H / 2, cells = 0;                   /* utility.py:61-64 */
    int x, y;
#define D() wfo_rng_u32(seed, id, ep, n++)
    memset(water, 0, (size_t)H * (size_t)W);
    x = (int)below(D(), (uint32_t)W);                        /* :75 np.random.choice(range(WIDTH)) */
    y = 1 + (int)below(D(), 3);                              /* :77 np.random.choice(d) */
    while (y < H - (1 + (int)below(D(), 3))) {               /* :79 */
        int nx = x, ny = y + 1, tries = 0, ok = 0;
        water[y * W + x] = 1;                                /* :81-87 */
        cells++;
        do {                                                 /* :89-93 */
            int s = (D() & 1u) ? 1 : -1;                     /* np.random.choice([1, -1]) */
            int a = 1 + (int)below(D(), 3), b = 1 + (int)below(D(), 3);
            nx = x + s;
            ok = (a <= nx && nx < W - b) || (nx == fx && ny == fy);
        } while (!ok && ++tries < 64);
        if (!ok)                                             /* never in practice: (2/3)^64 */
            nx = x + 1 <= W - 2 ? x + 1 : x - 1;
        x = nx;
        y = ny;
    }
#undef D
    return cells;
}

/* ---------------------------------------------------------- circle_points */

/* utility.py:8-52, midpoint circle around (0,0), points in generation order */
int wfo_circle_points(int r, int *dx, int *dy)
{
    int n = 0, x = r, y = 0, P;
#define PUT(a, b) do { dx[n] = (a); dy[n] = (b); ++n; } while (0)
    PUT(x, y);                                   /* :13 */
    if (r > 0) {                                 /* :16-19 */
        PUT(-x, -y);
        PUT(y, -x);
        PUT(-y, x);
    }
    P = 1 - r;                                   /* :22 */
    while (x > y) {                              /* :23 */
        y += 1;
        if (P <= 0) {                            /* :27-28 */
            P = P + 2 * y + 1;
        } else {                                 /* :30-32 */
            x -= 1;
            P = P + 2 * y - 2 * x + 1;
        }
        if (x < y)                               /* :35-36 */
            break;
        PUT(x, y);                               /* :39-42 */
        PUT(-x, y);
        PUT(x, -y);
        PUT(-x, -y);
        if (x != y) {                            /* :46-50 */
            PUT(y, x);
            PUT(-y, x);
            PUT(y, -x);
            PUT(-y, -x);
        }
    }
#undef PUT
    return n;
}

/* ------------------------------------------------------- one world (scratch) */

typedef struct {
    int W, H;
    const wfo_params *p;
    /* layers, indexed [y*W + x] (the reference indexes env[x, y]; the axis swap
     * happens only here, at decode/encode) */
    double *type, *temp, *fuel;
    int8_t *agent_pos;
    /* fire_mobility layer as a flag: 1 = inf (dirt, water, and whatever dig() blocked; environment.py:85,133).
     * Not a function of `type` alone: Agent.dig() on a BURNING cell (toggle_digging, :136-138) retypes it to
     * dirt but leaves it in burning_cells, and reduce_fuel() later retypes it to burnt with the mobility still inf. */
    uint8_t *fmob_inf;
    uint8_t *in_burning;     /* scratch: membership flags of burning_cells (encode) */
    /* World.burning_cells (environment.py:202): list of cell indices */
    int *burning;
    int n_burning;
    /* scratch for the flood fill */
    int *queue;
    uint8_t *seen;
    /* agent + flags */
    int ax, ay;
    int alive, visible, digging, dead, running, fire_at_border, contained;
    double quantum;     /* one heat addend: wind_speed*heat*(0+1)^-1 */
    /* World.wind_speed / wind_vector (environment.py:187-193) */
    double wind_speed, wind_x, wind_y;
    /* wind_mode != 0: addends received per direction, [4*cell + k] (exchange format, see wf_oracle.h) */
    uint8_t *cnt;
} world;

/* direction index of the vector (cx, cy) = heated cell - burning cell, in the order of _direction_to_coords */
static int dir_index(int cx, int cy)
{
    return cy < 0 ? 0 : cy > 0 ? 1 : cx > 0 ? 2 : 3;
}

/* the addend apply_heat_from_to computes for that vector (environment.py:260-287) */
static double heat_addend(const world *w, double cx, double cy, double distance)
{
    double wx = w->wind_x, wy = w->wind_y;
    double angle = fabs(atan2(wx * cy - wy * cx, wx * cx + wy * cy));   /* :273 */
    double env_factor = pow(angle + distance, -1.0);                    /* :286 */
    return w->wind_speed * w->p->heat * env_factor;                     /* :287 */
}

static int inbounds(const world *w, int x, int y)           /* environment.py:225-226 */
{
    return 0 <= x && x < w->W && 0 <= y && y < w->H;
}

static int is_burning(const world *w, int x, int y)          /* environment.py:249-251 */
{
    return w->type[y * w->W + x] == WFO_FIRE;
}

static int is_burnable(const world *w, int x, int y)         /* environment.py:254-257 */
{
    double t = w->type[y * w->W + x];
    return !(t == WFO_FIRE || t == WFO_BURNT || t == WFO_DIRT || t == WFO_WATER);
}

static void set_fire_to(world *w, int x, int y)              /* environment.py:233-246 */
{
    int c = y * w->W + x, i, present = 0;
    if (w->temp[c] < w->p->threshold)                        /* :236-237 */
        w->temp[c] = w->p->threshold + 1;
    w->type[c] = WFO_FIRE;                                   /* :241 */
    for (i = 0; i < w->n_burning; ++i)                       /* :243 set.add */
        if (w->burning[i] == c)
            present = 1;
    if (!present)
        w->burning[w->n_burning++] = c;
    if (x == 0 || x == w->W - 1 || y == 0 || y == w->H - 1)  /* :245-246 */
        w->fire_at_border = 1;
}

static void dig(world *w)                                    /* environment.py:123-133 */
{
    if (w->digging) {
        int c = w->ay * w->W + w->ax;
        if (w->type[c] != WFO_DIRT) {
            w->type[c] = WFO_DIRT;   /* :129 */
            w->fmob_inf[c] = 1;      /* :133 fire_mobility = inf.  burning_cells is NOT touched: a cell that was
                                      * burning keeps burning (fuel, radiation) under its new type */
        }
    }
}

static void agent_move(world *w, int direction)              /* environment.py:141-171 */
{
    int nx = w->ax, ny = w->ay;
    w->agent_pos[w->ay * w->W + w->ax] = 0;                  /* :142 (always) */
    w->visible = 0;
    switch (direction) {                                     /* :163-171 */
    case 0: ny -= 1; break;   /* N */
    case 1: ny += 1; break;   /* S */
    case 2: nx += 1; break;   /* E */
    case 3: nx -= 1; break;   /* W */
    default: return;
    }
    if (inbounds(w, nx, ny) && w->type[ny * w->W + nx] != WFO_WATER) {   /* :146, :229-230 */
        w->ax = nx;                                          /* :148-149 */
        w->ay = ny;
        w->agent_pos[ny * w->W + nx] = 1;
        w->visible = 1;
        if (w->digging && !is_burning(w, nx, ny))            /* :151-152 */
            dig(w);
        if (is_burning(w, nx, ny))                           /* :154-155 */
            w->dead = 1;
    }
}

static int reduce_fuel(world *w, int idx_in_list)            /* environment.py:297-307 */
{
    int c = w->burning[idx_in_list];
    w->fuel[c] -= 1;
    if (w->fuel[c] <= 0) {
        w->type[c] = WFO_BURNT;
        w->burning[idx_in_list] = -1;                        /* set.remove */
        return 0;
    }
    return 1;
}

static void apply_heat_from_to(world *w, int x, int y, int ox, int oy)   /* environment.py:278-294 */
{
    /* get_distance_and_angle, manhattan (:260-275) */
    double cx = ox - x, cy = oy - y;
    double distance = fabs((double)(x - ox)) + fabs((double)(y - oy));
    double calculated_heat = heat_addend(w, cx, cy, distance);
    int o = oy * w->W + ox;
    w->temp[o] += calculated_heat;                           /* :290 */
    if (w->cnt && w->cnt[4 * o + dir_index(ox - x, oy - y)] < 255)
        w->cnt[4 * o + dir_index(ox - x, oy - y)] += 1;
    if (w->temp[o] > w->p->threshold)                        /* :293-294 */
        set_fire_to(w, ox, oy);
}

static void update(world *w)                                 /* forest_fire.py:85-106 */
{
    int i, k, snapshot, m;
    static const int ddx[4] = { 1, -1, 0, 0 }, ddy[4] = { 0, 0, 1, -1 };

    /* :87  drop dead agents; Agent.is_dead environment.py:116-120 */
    if (w->alive && (w->dead || is_burning(w, w->ax, w->ay))) {
        w->agent_pos[w->ay * w->W + w->ax] = 0;
        w->visible = 0;
        w->alive = 0;
    }

    snapshot = w->n_burning;                                 /* :90 list(copy) */
    for (i = 0; i < snapshot; ++i) {                         /* :92 */
        int c = w->burning[i], x = c % w->W, y = c / w->W;
        if (!reduce_fuel(w, i))                              /* :95 */
            continue;
        /* get_neighbours, radius 1 => the four manhattan-1 cells (environment.py:311-326) */
        for (k = 0; k < 4; ++k) {
            int nx = x + ddx[k], ny = y + ddy[k];
            if (inbounds(w, nx, ny) && is_burnable(w, nx, ny))       /* :324 */
                if (is_burnable(w, nx, ny))                          /* forest_fire.py:99 */
                    apply_heat_from_to(w, x, y, nx, ny);             /* :102 */
        }
    }
    /* compact the list (cells removed by reduce_fuel) */
    for (i = 0, m = 0; i < w->n_burning; ++i)
        if (w->burning[i] >= 0)
            w->burning[m++] = w->burning[i];
    w->n_burning = m;

    if (!w->alive || w->n_burning == 0)                      /* :105-106 */
        w->running = 0;
}

/* "no route from any burning cell to any border point" (environment.py:345-377,
 * pyastar/astar.cpp:41-116 with infinite weights on dirt/water).  The reference
 * pops one burning cell and walks a persistent deque of border points with A*;
 * with a 4-connected fire region that is equivalent to this reachability test
 * (pinned by the 'circle' golden episodes).  Returns 1 when a
 * passable border cell is reachable. */
static int fire_reaches_border(world *w)
{
    int W = w->W, H = w->H, head = 0, tail = 0, i, k;
    static const int ddx[4] = { 1, -1, 0, 0 }, ddy[4] = { 0, 0, 1, -1 };
    memset(w->seen, 0, (size_t)(W * H));
    for (i = 0; i < w->n_burning; ++i) {
        w->seen[w->burning[i]] = 1;
        w->queue[tail++] = w->burning[i];
    }
    while (head < tail) {
        int c = w->queue[head++], x = c % W, y = c / W;
        if (x == 0 || y == 0 || x == W - 1 || y == H - 1)
            return 1;
        for (k = 0; k < 4; ++k) {
            int nx = x + ddx[k], ny = y + ddy[k], n;
            if (!inbounds(w, nx, ny))
                continue;
            n = ny * W + nx;
            if (w->seen[n] || w->fmob_inf[n])
                continue;                                    /* fire_mobility == inf (weights of astar_path, :353-356) */
            w->seen[n] = 1;
            w->queue[tail++] = n;
        }
    }
    return 0;
}

static double get_reward(world *w)                           /* environment.py:342-390 */
{
    int i, healthy = 0;
    if (w->p->containment && !w->fire_at_border && !w->contained && w->n_burning) {   /* :345-346 */
        if (!fire_reaches_border(w)) {
            w->contained = 1;                                /* deque left empty, :361 */
            return w->p->contained_bonus;                    /* :367 */
        }
    }
    if (!w->alive)                                           /* :380-381 */
        return w->p->death_penalty;
    if (w->n_burning == 0) {                                 /* :384-387 */
        for (i = 0; i < w->W * w->H; ++i)
            if (w->type[i] == 0)
                healthy++;
        return w->p->contained_bonus * ((double)healthy / (double)(w->W * w->H));
    }
    return w->p->default_reward;                             /* :390 */
}

/* ------------------------------------------------------ decode / encode */

static int world_alloc(world *w, int H, int W, const wfo_params *p)
{
    size_t n = (size_t)H * (size_t)W;
    memset(w, 0, sizeof(*w));
    w->W = W;
    w->H = H;
    w->p = p;
    w->type = malloc(n * sizeof(double));
    w->temp = malloc(n * sizeof(double));
    w->fuel = malloc(n * sizeof(double));
    w->agent_pos = malloc(n);
    w->fmob_inf = malloc(n);
    w->in_burning = malloc(n);
    w->burning = malloc(n * sizeof(int));
    w->queue = malloc(n * sizeof(int));
    w->seen = malloc(n);
    w->quantum = p->wind_speed * p->heat * pow(0.0 + 1.0, -1.0);
    w->wind_speed = p->wind_speed;                           /* environment.py:192-193 */
    w->wind_x = p->wind_x;
    w->wind_y = p->wind_y;
    w->cnt = NULL;
    return w->type && w->temp && w->fuel && w->agent_pos && w->fmob_inf && w->in_burning && w->burning && w->queue && w->seen;
}

static void world_free(world *w)
{
    free(w->type); free(w->temp); free(w->fuel); free(w->agent_pos); free(w->fmob_inf); free(w->in_burning);
    free(w->burning); free(w->queue); free(w->seen);
}

static double temp_from_count(const world *w, int q)
{
    double t = 0.0;
    int i;
    if (q == WFO_HEAT_FORCED)
        return w->p->threshold + 1;
    for (i = 0; i < q; ++i)
        t += w->quantum;            /* the same running sum the reference builds */
    return t;
}

/* wind_mode != 0: the environment's own wind (wf_oracle.h) */
static void take_wind(world *w, const wfo_env *e)
{
    if (!w->p->wind_mode)
        return;
    w->wind_speed = w->p->wind_speeds[(e->wind >> 4) & 3];
    w->wind_x = (double)((int)(e->wind & 3) - 1);
    w->wind_y = (double)((int)((e->wind >> 2) & 3) - 1);
}

/* wind_mode != 0: temperature of a cell from its marker byte and per-direction counts */
static double temp_from_wind_counts(const world *w, int marker, const uint8_t *cnt)
{
    static const int dcx[4] = { 0, 0, 1, -1 }, dcy[4] = { -1, 1, 0, 0 };
    double t = 0.0;
    int k, i;
    if (marker == WFO_HEAT_FORCED)
        return w->p->threshold + 1;
    if (marker == WFO_HEAT_IGNITED)
        return w->p->threshold + 0.5;       /* some value above the threshold; never read again (a cell burns once) */
    for (k = 0; k < 4; ++k) {
        double c = heat_addend(w, dcx[k], dcy[k], 1.0);
        for (i = 0; i < cnt[k]; ++i)
            t += c;
    }
    return t;
}

static void decode(world *w, const int8_t *pl, const wfo_env *e, uint8_t *counts)
{
    int n = w->W * w->H, i;
    w->n_burning = 0;
    w->cnt = w->p->wind_mode ? counts : NULL;
    take_wind(w, e);
    for (i = 0; i < n; ++i) {
        int t = pl[4 * i + WFO_P_TYPE];
        /* boundary codes 5 / 6 (see wf_oracle.h): reference type dirt + member of burning_cells / reference
         * type burnt + fire_mobility inf */
        w->type[i] = t == WFO_DIRT_BURNING ? WFO_DIRT : t == WFO_BURNT_BLOCKED ? WFO_BURNT : t == WFO_FIRE_BLOCKED ? WFO_FIRE : t;
        w->fmob_inf[i] = t >= WFO_DIRT;
        w->fuel[i] = pl[4 * i + WFO_P_FUEL];
        w->temp[i] = w->cnt ? temp_from_wind_counts(w, pl[4 * i + WFO_P_HEAT], w->cnt + 4 * i)
                            : temp_from_count(w, pl[4 * i + WFO_P_HEAT]);
        w->agent_pos[i] = pl[4 * i + WFO_P_AGENT];
        if (t == WFO_FIRE || t == WFO_DIRT_BURNING || t == WFO_FIRE_BLOCKED)
            w->burning[w->n_burning++] = i;
    }
    w->ax = e->x;
    w->ay = e->y;
    w->alive = !!(e->flags & WFO_F_ALIVE);
    w->visible = !!(e->flags & WFO_F_VISIBLE);
    w->digging = !!(e->flags & WFO_F_DIGGING);
    w->dead = !!(e->flags & WFO_F_DEAD);
    w->running = !!(e->flags & WFO_F_RUNNING);
    w->fire_at_border = !!(e->flags & WFO_F_BORDER);
    w->contained = !!(e->flags & WFO_F_CONTAINED);
}

static void encode(const world *w, int8_t *pl, wfo_env *e)
{
    int n = w->W * w->H, i;
    memset(w->in_burning, 0, (size_t)n);
    for (i = 0; i < w->n_burning; ++i)
        if (w->burning[i] >= 0)
            w->in_burning[w->burning[i]] = 1;
    for (i = 0; i < n; ++i) {
        int t = (int)w->type[i];
        if (t == WFO_DIRT && w->in_burning[i]) t = WFO_DIRT_BURNING;
        if (t == WFO_BURNT && w->fmob_inf[i]) t = WFO_BURNT_BLOCKED;
        if (t == WFO_FIRE && w->fmob_inf[i]) t = WFO_FIRE_BLOCKED;     /* a river under the fire cell (:81-85, then :241) */
        pl[4 * i + WFO_P_TYPE] = (int8_t)t;
        pl[4 * i + WFO_P_FUEL] = (int8_t)w->fuel[i];
        if (w->cnt) {
            pl[4 * i + WFO_P_HEAT] = w->temp[i] == w->p->threshold + 1 ? WFO_HEAT_FORCED
                                   : w->temp[i] > w->p->threshold ? WFO_HEAT_IGNITED : 0;
            if (t != WFO_GRASS)
                memset(w->cnt + 4 * i, 0, 4);
        } else {
            pl[4 * i + WFO_P_HEAT] = (int8_t)lrint(w->temp[i] / w->quantum);
        }
        pl[4 * i + WFO_P_AGENT] = w->agent_pos[i];
    }
    e->x = (uint16_t)w->ax;
    e->y = (uint16_t)w->ay;
    e->flags = (uint16_t)((w->alive ? WFO_F_ALIVE : 0) | (w->visible ? WFO_F_VISIBLE : 0) |
                          (w->digging ? WFO_F_DIGGING : 0) | (w->dead ? WFO_F_DEAD : 0) |
                          (w->running ? WFO_F_RUNNING : 0) | (w->fire_at_border ? WFO_F_BORDER : 0) |
                          (w->contained ? WFO_F_CONTAINED : 0));
}

/* ------------------------------------------------------------- public API */

int wfo_step(int batch, int H, int W, int8_t *planes, wfo_env *envs,
             const int8_t *actions, int do_update, const wfo_params *p,
             double *reward, uint8_t *done)
{
    return wfo_step_wind(batch, H, W, planes, envs, NULL, actions, do_update, p, reward, done);
}

int wfo_step_wind(int batch, int H, int W, int8_t *planes, wfo_env *envs, uint8_t *counts,
                  const int8_t *actions, int do_update, const wfo_params *p,
                  double *reward, uint8_t *done)
{
    world w;
    int b;
    size_t stride = (size_t)H * (size_t)W * 4;
    if (batch < 0 || H <= 0 || W <= 0 || !planes || !envs || !p)
        return -1;
    if ((p->wind_mode != 0) != (counts != NULL))
        return -1;
    if (!world_alloc(&w, H, W, p)) {
        world_free(&w);
        return -1;
    }
    for (b = 0; b < batch; ++b) {
        int8_t *pl = planes + stride * (size_t)b;
        wfo_env *e = &envs[b];
        int a = actions ? actions[b] : -1;
        double r;
        if (!(e->flags & WFO_F_RUNNING)) {       /* frozen: the reference is never stepped after done */
            if (reward) reward[b] = 0.0;
            if (done) done[b] = 1;
            continue;
        }
        decode(&w, pl, e, counts ? counts + stride * (size_t)b : NULL);
        /* ForestFire.step, forest_fire.py:30-49 */
        if (a >= 0 && a <= 3)                    /* :32-33 */
            agent_move(&w, a);
        if (p->allow_dig_toggle && a == 4) {     /* :35-36, Agent.toggle_digging environment.py:136-138 */
            w.digging = !w.digging;
            dig(&w);
        }
        if (do_update) {                         /* :40-43 */
            update(&w);
            e->tick += 1;
        }
        r = get_reward(&w);                      /* :47 */
        encode(&w, pl, e);
        if (reward) reward[b] = r;
        if (done) done[b] = (uint8_t)!w.running; /* :48 */
    }
    world_free(&w);
    return 0;
}

int wfo_reset(int batch, int H, int W, int8_t *planes, wfo_env *envs,
              const uint8_t *mask, uint64_t seed, uint64_t env_id0,
              const wfo_params *p)
{
    world w;
    int b, i, n = H * W;
    size_t stride = (size_t)n * 4;
    int cdx[3][32], cdy[3][32], cn[3];
    if (batch < 0 || H <= 0 || W <= 0 || !planes || !envs || !p)
        return -1;
    if (H < 7 || W < 7)          /* the spawn circle of radius 3 must fit (the reference asserts 10x10, utility.py:68) */
        return -1;
    if (!world_alloc(&w, H, W, p)) {
        world_free(&w);
        return -1;
    }
    for (i = 0; i < 3; ++i)
        cn[i] = wfo_circle_points(i + 1, cdx[i], cdy[i]);

    for (b = 0; b < batch; ++b) {
        wfo_env *e = &envs[b];
        uint64_t id = env_id0 + (uint64_t)b;
        uint32_t ep;
        int fx, fy, r, k, sx, sy;
        if (mask && !mask[b])
            continue;
        ep = e->episode;
        /* World.reset environment.py:186-212: wind :187-193 */
        e->wind = 0;
        if (p->wind_mode == 1)
            e->wind = (uint16_t)p->wind_code;
        if (p->wind_mode == 2) {                             /* :189-190, three uniform draws */
            unsigned si = below(wfo_rng_u32(seed, id, ep, 2), 3);
            int wx = (int)below(wfo_rng_u32(seed, id, ep, 3), 3) - 1;
            int wy = (int)below(wfo_rng_u32(seed, id, ep, 4), 3) - 1;
            e->wind = (uint16_t)WFO_WIND_CODE(wx, wy, si);
        }
        w.cnt = NULL;                                        /* a fresh grid has received no heat: the caller zeroes counts */
        /* reset_map :59-67 */
        w.running = 1;                                       /* :196 */
        for (i = 0; i < n; ++i) {
            w.type[i] = 0;
            w.temp[i] = 0;
            w.agent_pos[i] = 0;
            w.fmob_inf[i] = 0;                               /* :64 fire_mobility.fill(1) */
            if (p->fuel_lo == p->fuel_hi)
                w.fuel[i] = p->fuel_lo;                      /* :63 grass['fuel'] */
            else
                w.fuel[i] = p->fuel_lo + (int)below(wfo_rng_u32(seed, id, ep, 16u + (uint32_t)i),
                                                    (uint32_t)(p->fuel_hi - p->fuel_lo + 1));
        }
        if (p->make_rivers) {                                /* :69-95 */
            wfo_river(H, W, seed, id, ep, w.seen);
            for (i = 0; i < n; ++i)
                if (w.seen[i]) {
                    w.type[i] = WFO_WATER;
                    w.fmob_inf[i] = 1;
                }
        }
        w.n_burning = 0;                                     /* :202 */
        w.fire_at_border = 0;
        fx = (int)(W / 2);                                   /* utility.py:61-64 */
        fy = (int)(H / 2);
        set_fire_to(&w, fx, fy);                             /* :203 */
        /* get_agent_location utility.py:66-78 (two uniform draws) */
        r = 1 + (int)below(wfo_rng_u32(seed, id, ep, 0), 3);
        k = (int)below(wfo_rng_u32(seed, id, ep, 1), (uint32_t)cn[r - 1]);
        sx = fx + cdx[r - 1][k];
        sy = fy + cdy[r - 1][k];
        /* Agent.__init__ environment.py:100-113 */
        w.ax = sx;
        w.ay = sy;
        w.agent_pos[sy * W + sx] = 1;
        w.alive = 1;
        w.visible = 1;
        w.dead = 0;
        w.digging = 1;
        dig(&w);
        w.fire_at_border = 0;                                /* :212 */
        w.contained = 0;                                     /* :211 reset_border_points */
        /* synthetic extension: extra ignition points on grass cells */
        if (p->ignite_ppm)
            for (i = 0; i < n; ++i)
                if (is_burnable(&w, i % W, i / W) &&
                    below(wfo_rng_u32(seed, id, ep, 16u + (uint32_t)n + (uint32_t)i), 1000000u) < p->ignite_ppm)
                    set_fire_to(&w, i % W, i / W);
        encode(&w, planes + stride * (size_t)b, e);
        e->episode = ep + 1;
        e->tick = 0;
    }
    world_free(&w);
    return 0;
}

int wfo_get_state(int batch, int H, int W, const int8_t *planes, int8_t *obs)
{
    int b, x, y;
    if (batch < 0 || H <= 0 || W <= 0 || !planes || !obs)
        return -1;
    for (b = 0; b < batch; ++b) {
        const int8_t *pl = planes + (size_t)b * H * W * 4;
        int8_t *o = obs + (size_t)b * W * H * 3;
        for (x = 0; x < W; ++x)
            for (y = 0; y < H; ++y) {
                const int8_t *c = pl + 4 * (y * W + x);
                int8_t *d = o + 3 * (x * H + y);             /* environment.py:399-402, [x][y][k] */
                d[0] = c[WFO_P_AGENT];
                d[1] = (int8_t)(c[WFO_P_TYPE] == WFO_FIRE || c[WFO_P_TYPE] == WFO_FIRE_BLOCKED);
                d[2] = (int8_t)(c[WFO_P_TYPE] < WFO_DIRT);   /* fire_mobility != inf: not dirt / water / codes 5, 6, 13 */
            }
    }
    return 0;
}
