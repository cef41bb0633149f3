#!/usr/bin/env python3
"""Model of the render kernel's stage scheduler: one wave of 64 lanes over synthetic paths whose walk statistics match the
benchmark scene (box tests / ray, primitive tests / ray, rays / sample), to price scheduling variants in VALU issue units
before building them.  R rays per lane (R = 1 is the kernel as built); with R = 2 a lane keeps an active and a parked ray and
swaps them (cost SWAP per pass in which any lane swaps) when the stage being run wants the parked one.

    python scratch/sched_sim.py
"""
import random
import sys

BOX, PRIM, DONE, IDLE = 0, 1, 2, 3
C_BOX, C_PRIM, C_SERV, C_SCHED = 22.0, 50.0, 300.0, 6.0


def new_walk(rng):
    """state sequence of one ray's walk: runs of box steps separated by primitive tests, then DONE"""
    n_prim = 0
    # primitive visits: mean 2.1
    while rng.random() < 2.1 / 3.1:
        n_prim += 1
    seq = []
    for _ in range(n_prim + 1):
        k = 1
        while rng.random() < 1.0 - 1.0 / 7.7:
            k += 1
        seq.extend([BOX] * k)
        seq.append(PRIM)
    seq.pop()  # ends after a box run (walks fall off the array through a miss link)
    seq.append(DONE)
    return seq


class Ray:
    __slots__ = ("seq", "i", "state")

    def __init__(self):
        self.state = IDLE
        self.seq = None
        self.i = 0

    def begin(self, rng):
        self.seq = new_walk(rng)
        self.i = 0
        self.state = self.seq[0]

    def advance(self):
        self.i += 1
        self.state = self.seq[self.i]


def simulate(R, swap_cost, n_samples, seed=1, k=6, w_prim=8, w_serv=3, p_continue=0.655, verbose=False):
    rng = random.Random(seed)
    lanes = [[Ray() for _ in range(R)] for _ in range(64)]  # lanes[l][0] is the active ray
    pool = n_samples
    cost = 0.0
    rays = 0
    stat = {"box_steps": 0, "box_lanes": 0, "prim": 0, "prim_lanes": 0, "serv": 0, "serv_lanes": 0, "swaps": 0, "passes": 0}

    def count(s):
        # lanes that can run stage s this pass (active or parked ray in state s)
        return sum(1 for ln in lanes if any(r.state == s for r in ln))

    def select(s):
        """make the active ray of each lane one that is in state s, if the lane has any; returns True if any lane swapped"""
        swapped = False
        for ln in lanes:
            if ln[0].state != s:
                for j in range(1, R):
                    if ln[j].state == s:
                        ln[0], ln[j] = ln[j], ln[0]
                        swapped = True
                        break
        return swapped

    while True:
        n_box, n_prim, n_done = count(BOX), count(PRIM), count(DONE)
        n_idle_rays = sum(1 for ln in lanes for r in ln if r.state == IDLE)
        n_idle = count(IDLE)
        n_serv = max(n_done, min(n_idle, pool)) if R > 1 else n_done + min(n_idle, pool)
        if n_box + n_prim + n_serv == 0:
            break
        cost += C_SCHED
        stat["passes"] += 1
        sb, sp, ss = n_box * 8, n_prim * w_prim, n_serv * w_serv
        if n_box and sb >= sp and sb >= ss:
            if R > 1 and select(BOX):
                cost += swap_cost
                stat["swaps"] += 1
            n_fixed = n_box + n_prim + n_serv
            while True:
                act = [ln[0] for ln in lanes if ln[0].state == BOX]
                nb = len(act)
                if nb == 0 or nb * (k + 1) < n_fixed:
                    break
                cost += C_BOX
                stat["box_steps"] += 1
                stat["box_lanes"] += nb
                for r in act:
                    r.advance()
        elif n_prim and (sp >= ss or n_serv == 0):
            if R > 1 and select(PRIM):
                cost += swap_cost
                stat["swaps"] += 1
            act = [ln[0] for ln in lanes if ln[0].state == PRIM]
            cost += C_PRIM
            stat["prim"] += 1
            stat["prim_lanes"] += len(act)
            for r in act:
                r.advance()
        else:
            # service: shade the DONE active rays; lanes whose active ray is IDLE generate
            if R > 1:
                sw = select(DONE)
                # lanes without a DONE ray bring an IDLE one forward if the pool has samples
                if pool:
                    for ln in lanes:
                        if ln[0].state not in (DONE, IDLE):
                            for j in range(1, R):
                                if ln[j].state == IDLE:
                                    ln[0], ln[j] = ln[j], ln[0]
                                    sw = True
                                    break
                if sw:
                    cost += swap_cost
                    stat["swaps"] += 1
            cost += C_SERV
            stat["serv"] += 1
            n = 0
            for ln in lanes:
                r = ln[0]
                if r.state == DONE:
                    n += 1
                    if rng.random() < p_continue:
                        r.begin(rng)
                        rays += 1
                    else:
                        r.state = IDLE
                if r.state == IDLE and pool:
                    pool -= 1
                    r.begin(rng)
                    rays += 1
                    n += 0
            stat["serv_lanes"] += n
    if verbose:
        print(f"  box lanes/step {stat['box_lanes'] / max(1, stat['box_steps']):.1f}  prim lanes/pass {stat['prim_lanes'] / max(1, stat['prim']):.1f}  "
              f"shade lanes/pass {stat['serv_lanes'] / max(1, stat['serv']):.1f}  box steps/ray {stat['box_lanes'] / rays:.1f}  passes {stat['passes']}  swaps {stat['swaps']}")
    return cost / rays


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    base = simulate(1, 0.0, n, verbose=True)
    print(f"R=1: {base:.1f} units / ray")
    for R in (2, 3):
        for sw in (0.0, 24.0, 40.0, 60.0):
            c = simulate(R, sw, n, verbose=(sw == 24.0))
            print(f"R={R} swap={sw:4.0f}: {c:.1f} units / ray  ({base / c:.3f}x)")
