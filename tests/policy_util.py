"""Test-side restatement of the reference's clockwise random-walk baseline policy
(DQN.choose_randomwalk_action, DQN.py:353-389) and of its baseline episode loop
(DQN.collect_memories with perform_baseline=True, DQN.py:286-346), vectorised over a batch
of environments.  The reference publishes the mean episode reward of this policy:
1130.2 (10x10) and 1151.4 (14x14) over 10 x 10 000 episodes (Plots/results_first.txt:2,20) -
the one end-to-end pin of reset spawn law + move / dig / death + containment + burnt-out
reward that the reference's authors wrote down themselves.
"""
import numpy as np

REFERENCE_BASELINE_MEAN = {10: 1130.2, 14: 1151.4}     # Plots/results_first.txt:2,20
REFERENCE_BASELINE_EPISODES = 100000                    # 10 runs x 10 000 episodes (Logs/Baselines/Size10, Size14)

# action codes N,S,E,W = 0,1,2,3 (key_map DQN.py:360); (dx, dy) of environment.py:163-171
_DX = np.array([0, 0, 1, -1])
_DY = np.array([-1, 1, 0, 0])


def randomwalk_actions(x, y, alive, fire, rng, avoid_fire=True):
    """One action per environment.
    x, y   agent position (B,);  alive (B,) bool: W.agents non-empty (dead -> action 0, DQN.py:356-357)
    fire   (B, H, W) bool: type == fire, indexed [b, y, x]
    The pair of candidate actions depends on the quadrant around the grid centre (DQN.py:369-376); a candidate
    that leads into a burning cell (Agent.fire_in_direction, environment.py:158-160) is redrawn, at most 11 times
    (`count > 10`, DQN.py:384-387)."""
    B, H, W = fire.shape
    x = x.astype(np.int64)
    y = y.astype(np.int64)
    mx, my = W // 2, H // 2
    pair = np.zeros((B, 2), dtype=np.int64)
    c1 = (x >= mx) & (y > my)
    c2 = (x > mx) & (y <= my)
    c3 = (x <= mx) & (y < my)
    c4 = (x < mx) & (y >= my)
    pair[c1] = (1, 3)      # ["S", "W"]
    pair[c2] = (1, 2)      # ["S", "E"]
    pair[c3] = (0, 2)      # ["N", "E"]
    pair[c4] = (0, 3)      # ["N", "W"]
    assert (c1 | c2 | c3 | c4 | ~alive).all(), "agent on the centre cell: the reference policy has no rule there"
    act = np.zeros(B, dtype=np.int8)
    todo = alive.copy()
    ar = np.arange(B)
    for count in range(12):
        idx = ar[todo]
        if len(idx) == 0:
            break
        a = pair[idx, rng.integers(0, 2, size=len(idx))]
        act[idx] = a
        if not avoid_fire:
            break
        nx, ny = x[idx] + _DX[a], y[idx] + _DY[a]
        inb = (nx >= 0) & (nx < W) & (ny >= 0) & (ny < H)
        burning = np.zeros(len(idx), dtype=bool)
        burning[inb] = fire[idx[inb], ny[inb], nx[inb]]
        if count > 10:
            break
        todo[idx[~burning]] = False
    return act


def run_baseline_episodes(reset, step, state, batch, rng):
    """One baseline episode per environment of a batch (DQN.py:286-346 with perform_baseline=True: the episode
    runs until `done`, containment does not end it).
      reset()            -> None            fresh episode in every environment
      state()            -> (x, y, alive, fire[b, y, x])
      step(actions int8) -> (reward float64 (B,), done bool (B,)); finished environments stay frozen (reward 0)
    Returns (total_reward (B,), agent_died (B,), steps (B,))."""
    reset()
    total = np.zeros(batch)
    steps = np.zeros(batch, dtype=np.int64)
    running = np.ones(batch, dtype=bool)
    alive = np.ones(batch, dtype=bool)
    for _ in range(100000):
        x, y, alive, fire = state()
        a = randomwalk_actions(x, y, alive & running, fire, rng)
        r, d = step(a)
        total[running] += r[running]
        steps[running] += 1
        running &= ~d
        if not running.any():
            break
    else:
        raise AssertionError("episodes did not finish")
    _, _, alive, _ = state()
    return total, ~alive, steps


def baseline_tolerance(std, n):
    """4 standard errors of the difference between our mean over n episodes and the reference's mean over its
    100 000 episodes (both sample the same law if the environments agree; std is estimated from our sample)."""
    return 4.0 * std * np.sqrt(1.0 / n + 1.0 / REFERENCE_BASELINE_EPISODES)
