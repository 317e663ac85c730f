"""Round 6, VERDICT r5 item 7 (measure first, no kernel): if the sources a wave of bp_beam_fast_kernel carries
together shared an LDS read whenever their moveouts to a (station, phase) agree, how many reads would go?

Host-side only (NumPy; no GPU).  Geometry of bench.py: cfg3 (50 x 50 x 20 lattice over 100 x 100 x 30 km, 20
stations, 50 Hz) and one GPU's share of configs[4] (125 x 125 x 8 slab of the 125 x 125 x 64 lattice, 40 stations,
100 Hz), ALL stations weighted (the 17-32 / 33-64-station classes: 9 carried sources per wave, groups of 144).
Groups = bricks of neighbouring lattice points (what the planner's median bisection of the moveout vectors produces
on a lattice); within a group the 16 waves' sets of 9 sources are chosen three ways:
  lattice   -- 9 consecutive lattice points,
  sorted    -- the group sorted by its moveout to the FIRST station, 9 consecutive of that order (the review's proposal),
  best-of-station -- per (station, phase) the group re-sorted by THAT moveout: an upper bound no single carry order reaches.
Reported: 1 - (distinct moveout values among the 9 carried sources) / 9, averaged over waves and (station, phase)
terms = the fraction of gathers that could be shared."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seismic_bpmf_amd import synthetic as syn  # noqa: E402


def saved(tau_sets):
    """tau_sets: (waves, 9, terms) -> mean over waves and terms of 1 - distinct / 9"""
    s = np.sort(tau_sets, axis=1)
    distinct = 1 + (np.diff(s, axis=1) != 0).sum(axis=1)
    return float(1.0 - distinct.mean() / tau_sets.shape[1])


def run(name, grid, S, sr, slab=None, brick=(6, 6, 4)):
    geo = syn.make_bp_geometry(grid, S, 2, sr, n_closest=S, depth_slab=slab)
    tau = geo["moveouts"].reshape(grid[0], grid[1], grid[2], S * 2)
    bx, by, bz = brick
    rng = np.random.default_rng(0)
    res = {"lattice": [], "sorted": [], "best": []}
    n_groups = 0
    for _ in range(200):                              # 200 random bricks of 144 sources
        x0 = int(rng.integers(0, grid[0] - bx + 1)); y0 = int(rng.integers(0, grid[1] - by + 1)); z0 = int(rng.integers(0, grid[2] - bz + 1))
        g = tau[x0:x0 + bx, y0:y0 + by, z0:z0 + bz].reshape(-1, S * 2)        # (144, terms), z fastest
        n = g.shape[0] // 9 * 9
        res["lattice"].append(saved(g[:n].reshape(-1, 9, S * 2)))
        o = np.argsort(g[:, 0], kind="stable")
        res["sorted"].append(saved(g[o][:n].reshape(-1, 9, S * 2)))
        per_term = []
        for k in range(S * 2):
            col = np.sort(g[:, k])[:n].reshape(-1, 9)
            per_term.append(1.0 - (1 + (np.diff(col, axis=1) != 0).sum(axis=1)).mean() / 9.0)
        res["best"].append(float(np.mean(per_term)))
        n_groups += 1
    spread = float(np.mean([np.ptp(tau[x:x + bx, y:y + by, z:z + bz].reshape(-1, S * 2), axis=0).mean()
                            for x, y, z in [(0, 0, 0), (grid[0] // 2, grid[1] // 2, 0), (grid[0] - bx, grid[1] - by, grid[2] - bz)]]))
    print(f"{name}: {n_groups} groups of {bx * by * bz} sources, mean moveout spread per (station, phase) inside a group {spread:.0f} samples")
    for k, label in (("lattice", "9 consecutive lattice points"), ("sorted", "group sorted by its first-station moveout"),
                     ("best", "upper bound: re-sorted per (station, phase)")):
        print(f"   {label:48s}: {100 * np.mean(res[k]):5.1f} % of the gathers coincide with another carried source's")


if __name__ == "__main__":
    run("cfg3, all 20 stations (50 Hz, 2 x 2 x 1.5 km lattice)", (50, 50, 20), 20, 50.0)
    run("configs[4] share, all 40 stations (100 Hz, 0.8 x 0.8 x 0.47 km lattice)", (125, 125, 8), 40, 100.0, slab=(0, 64), brick=(6, 6, 4))
