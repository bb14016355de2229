"""Test-side model of the containment search inside the reference's World.get_reward
(Simulation/environment.py:345-377), with the ONE input the reference takes from CPython internals made explicit:
the order in which `burning.pop()` hands out the burning cells (set iteration order).

What the reference does, restated:
  * `border_points` is a deque of 2W + 2H [x, y] pairs, built by reset_border_points (:215-222).  get_reward pops
    candidates from the RIGHT, discards every one the search cannot reach and pushes the one it could reach back on
    the right (:355, :373, :377): the deque is therefore always a PREFIX of its initial order, and its whole state is
    its length L (0 <=> "containment already paid": the test of :345 never runs again).
  * the first burning cell popped is searched against that prefix; if it reaches none of it and other burning cells
    are left, the deque is refilled (:369) and the next cell is searched against ALL border points; and so on.  If no
    cell reaches anything the fire is contained (:362-367) and the deque stays empty.
  * "reaches" = pyastar.astar_path finds a route on the fire_mobility grid (inf on dirt / water / blocked cells): the
    start cell's own weight is never charged (pyastar/astar.cpp:41-116), the goal must be passable, and a route from a
    cell to itself counts as "no route" (pyastar/pyastar.py:52-64 returns an empty array for start == goal).

While all burning cells lie in one region of passable cells the result does not depend on the order (every cell
reaches the same border points, and points dropped earlier stay unreachable because the passable region only ever
shrinks).  When the fire is SPLIT over several regions it does: which cell is searched against the shortened deque,
and hence the deque's next length and - in contrived states - even whether containment is paid.

The build implements the trajectory the reference follows under the order SIGMA_STAR: always pop first a cell of the
region whose highest-indexed reachable border point is highest.  Then the prefix never hides a reachable point, and
"contained" <=> no burning cell reaches any passable border cell - the rule of the kernels and of the C oracle.

Used by tests/test_oracle_golden.py and tests/test_gpu_parity.py.  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np


def border_points(W, H):
    """reset_border_points (environment.py:215-222), in deque order; corners appear twice.  (The reference writes
    HEIGHT - 1 for the right column's x: grids are square.)"""
    pts = []
    for x in range(W):
        pts.append((x, 0))
        pts.append((x, H - 1))
    for y in range(H):
        pts.append((0, y))
        pts.append((H - 1, y))
    return pts


def regions(passable):
    """Label the 4-connected regions of passable cells.  passable: bool [y, x].  Returns int32 labels (0 = impassable)."""
    H, W = passable.shape
    lab = np.zeros((H, W), dtype=np.int32)
    n = 0
    for y0 in range(H):
        for x0 in range(W):
            if passable[y0, x0] and not lab[y0, x0]:
                n += 1
                lab[y0, x0] = n
                stack = [(x0, y0)]
                while stack:
                    x, y = stack.pop()
                    for nx, ny in ((x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)):
                        if 0 <= nx < W and 0 <= ny < H and passable[ny, nx] and not lab[ny, nx]:
                            lab[ny, nx] = n
                            stack.append((nx, ny))
    return lab


class SearchState:
    """Everything the containment search of one step needs, from one environment's type plane [y, x]."""

    def __init__(self, types):
        t = np.asarray(types).astype(np.int64)
        self.H, self.W = t.shape
        self.passable = t < 3                              # fire_mobility != inf: not dirt / water / codes 5, 6, 13
        self.lab = regions(self.passable)
        ys, xs = np.nonzero((t & 3) == 1)                  # World.burning_cells: fire (1), burning dirt (5), blocked fire (13)
        self.burning = list(zip(xs.tolist(), ys.tolist()))
        self.border = border_points(self.W, self.H)

    def reach_labels(self, cell):
        """Regions the search started at `cell` walks through: the cell's own region if it is passable, else (its
        weight is never charged) the regions of its passable 4-neighbours."""
        x, y = cell
        if self.passable[y, x]:
            return {int(self.lab[y, x])}
        out = set()
        for nx, ny in ((x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)):
            if 0 <= nx < self.W and 0 <= ny < self.H and self.passable[ny, nx]:
                out.add(int(self.lab[ny, nx]))
        return out

    def reaches(self, cell, point):
        px, py = point
        return point != tuple(cell) and self.passable[py, px] and int(self.lab[py, px]) in self.reach_labels(cell)

    def top_index(self, cell, length=None):
        """Highest index k < length with border point k reachable from `cell`; -1 if none."""
        n = len(self.border) if length is None else length
        for k in range(n - 1, -1, -1):
            if self.reaches(cell, self.border[k]):
                return k
        return -1

    def is_split(self):
        """The burning cells do not all search the same set of regions."""
        return len({frozenset(self.reach_labels(c)) for c in self.burning}) > 1

    def sigma_star(self):
        """The build's pop order: cells of the region with the highest reachable border index first."""
        return sorted(self.burning, key=lambda c: -self.top_index(c))


def containment_search(st, L, order):
    """One evaluation of environment.py:345-377.  st: SearchState; L: len(border_points) before; order: the burning
    cells in pop order (only as many as the search consumes are looked at).
    Returns (contained, L_after, cells_popped)."""
    assert L > 0 and st.burning
    full = len(st.border)
    todo = list(order)
    b = todo.pop(0)                                         # :350
    popped = [b]
    left = len(st.burning) - 1                              # cells still in the copy
    k = st.top_index(b, L)                                  # :355-373: scan the prefix from the right
    while k < 0:
        if left == 0:
            return True, 0, popped                          # :362-367 contained, deque left empty
        b = todo.pop(0)                                     # :369-371 refill, next cell
        popped.append(b)
        left -= 1
        k = st.top_index(b, full)
    return False, k + 1, popped                             # :377 the working end point goes back on the right


def replay_episode(types_per_step, fab_per_step, L0, orders=None):
    """Containment results of one episode.  types_per_step[t]: type plane [y, x] AFTER step t; fab_per_step[t]:
    World.fire_at_border after step t.  orders: None -> SIGMA_STAR; else orders[t] = recorded pop order of step t
    (may be empty when the search did not run).  Returns lists (contained[t], L[t], split[t], ran[t])."""
    L = int(L0)
    out_c, out_L, out_s, out_r = [], [], [], []
    for t, types in enumerate(types_per_step):
        st = SearchState(types)
        ran = (not fab_per_step[t]) and L > 0 and len(st.burning) > 0        # :345-346
        contained = False
        if ran:
            order = st.sigma_star() if orders is None else [tuple(int(v) for v in c) for c in orders[t]]
            if orders is not None:
                # the recorded cells are a prefix of the reference's order; cells it never popped cannot matter
                rest = [c for c in st.burning if c not in order]
                order = order + rest
            contained, L, popped = containment_search(st, L, order)
            if orders is not None:
                assert popped == order[:len(popped)] and len(popped) == len(orders[t]), (t, popped, orders[t])
        else:
            assert orders is None or len(orders[t]) == 0, (t, orders[t])
        out_c.append(contained)
        out_L.append(L)
        out_s.append(st.is_split() if ran else False)
        out_r.append(ran)
    return out_c, out_L, out_s, out_r
