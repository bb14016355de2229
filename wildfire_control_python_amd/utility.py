"""Tables and helpers the hot path needs (mirror of Simulation/utility.py).

The reference derives its grayscale constants from the PyPI package ``colour``
(utility.py:1, :87-91); only five fixed web colours are ever used, so their
grayscale values are computed here from the sRGB triples directly."""
import time

import numpy as np


def circle_points(midx, midy, r):
    """Midpoint-circle perimeter points in the reference's generation order
    (utility.py:8-52): r=1 -> 8 points, r=2 -> 12, r=3 -> 16."""
    pts = []
    x, y = r, 0
    pts.append((x + midx, y + midy))
    if r > 0:
        pts += [(-x + midx, -y + midy), (y + midx, -x + midy), (-y + midx, x + midy)]
    p = 1 - r
    while x > y:
        y += 1
        if p <= 0:
            p += 2 * y + 1
        else:
            x -= 1
            p += 2 * y - 2 * x + 1
        if x < y:
            break
        pts += [(x + midx, y + midy), (-x + midx, y + midy), (x + midx, -y + midy), (-x + midx, -y + midy)]
        if x != y:
            pts += [(y + midx, x + midy), (-y + midx, x + midy), (y + midx, -x + midy), (-y + midx, -x + midy)]
    return pts


def get_fire_location(width, height):
    """utility.py:61-64"""
    return (int(width / 2), int(height / 2))


def get_agent_location(width, height, rng=np.random):
    """utility.py:66-78: radius uniform in {1,2,3}, then a uniform point of that circle.
    Consumes the numpy legacy stream exactly like the reference (two choice() calls)."""
    assert width >= 10 and height >= 10
    radius = rng.choice([1, 2, 3])
    midx, midy = get_fire_location(width, height)
    locations = circle_points(midx, midy, radius)
    idx = rng.choice(np.array(range(len(locations))))
    loc = locations[idx]
    return (int(loc[0]), int(loc[1]))


def get_name(size, episodes, memories, name):
    """utility.py:81-85"""
    return f"""{name}-{size}s-{episodes}k-{memories}m-{time.strftime("%m-%d-%H%M")}"""


def _gray(rgb):
    r, g, b = rgb
    return 0.2989 * r + 0.5870 * g + 0.1140 * b       # utility.py:88-91


_GREEN, _RED, _BLACK = (0.0, 128 / 255.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.0, 0.0)
_BROWN, _BLUE = (165 / 255.0, 42 / 255.0, 42 / 255.0), (0.0, 0.0, 1.0)

grass = {                                              # utility.py:94-102
    "gray": _gray(_GREEN),
    "gray_burning": _gray(_RED),
    "gray_burnt": _gray(_BLACK),
    "heat": 0.3,
    "fuel": 20,
    "threshold": 3,
    "radius": 1,
}
dirt = {"gray": _gray(_BROWN)}                         # utility.py:105-107
water = {"gray": _gray(_BLUE)}                         # utility.py:110-112

layer = {                                              # utility.py:115-125
    "type": 0, "gray": 1, "temp": 2, "heat": 3, "fuel": 4, "threshold": 5,
    "agent_pos": 6, "fire_mobility": 7, "agent_mobility": 8,
}

types = {                                              # utility.py:128-140
    0: "grass", 1: "fire", 2: "burnt", 3: "dirt", 4: "water",
    "grass": 0, "fire": 1, "burnt": 2, "dirt": 3, "water": 4,
}

color2ascii = {                                        # utility.py:143-149
    grass["gray"]: "+", grass["gray_burning"]: "@", grass["gray_burnt"]: "#",
    dirt["gray"]: "0", water["gray"]: "x",
}
type2ascii = {0: "+", 1: "@", 2: "#", 3: "0", 4: "x"}
type2gray = {0: grass["gray"], 1: grass["gray_burning"], 2: grass["gray_burnt"],
             3: dirt["gray"], 4: water["gray"]}


def heat_quantum(wind_speed, heat=grass["heat"]):
    """One heat addend of apply_heat_from_to for wind vector (0,0) and manhattan
    distance 1 (environment.py:286-287): wind_speed * heat * (0 + 1)**-1."""
    return wind_speed * heat * (0 + 1) ** (-1)


def ignite_count(wind_speed, heat=grass["heat"], threshold=grass["threshold"]):
    """Smallest n such that the float64 running sum of n quanta exceeds the threshold
    (the reference adds one quantum at a time and tests `temp > threshold`,
    environment.py:290-294).  19 for the default wind 0.54."""
    q = heat_quantum(wind_speed, heat)
    if not q > 0:
        raise ValueError("wind speed and heat must be positive for fire to spread")
    t, n = 0.0, 0
    while not t > threshold:
        t += q
        n += 1
        if n > 120:
            raise ValueError("ignition needs more than 120 heat quanta: not representable in int8")
    return n


def temp_from_count(n, wind_speed, threshold=grass["threshold"]):
    """float64 temperature the reference would hold after n quanta (HEAT_FORCED -> threshold+1)."""
    if n == 25:
        return float(threshold + 1)
    q, t = heat_quantum(wind_speed), 0.0
    for _ in range(int(n)):
        t += q
    return t
