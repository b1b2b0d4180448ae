#!/usr/bin/env python
"""Greedy dispatch of the causal attention forward's workgroups onto the 512 slots of the chip (256 CUs x 2 workgroups): one query block
per workgroup in the kernel's launch order (head groups of 8, longest block first inside a group) against two blocks per workgroup
(block nt-1-p then block p: constant work).  Prints makespan / balanced time.  No GPU.

    python scripts/sim_attn_order.py
"""
import heapq


def makespan(jobs, slots=512):
    h = [0.0] * slots
    heapq.heapify(h)
    for j in jobs:
        heapq.heappush(h, heapq.heappop(h) + j)
    return max(h), sum(jobs) / slots


def single(nbh, nt, ovh=0.5):
    jobs = []
    for L in range(nt * nbh):
        tile = nt - 1 - (L >> 3) % nt
        jobs.append(2 * (tile + 1) + ovh)                # 64-key tiles of a 128-query block + prologue / epilogue
    return makespan(jobs)


def paired(nbh, nt, ovh=0.5):
    np_ = (nt + 1) // 2
    jobs = []
    for L in range(np_ * nbh):
        p = (L >> 3) % np_
        a, b = nt - 1 - p, p
        jobs.append(2 * (a + 1) + ovh + (2 * (b + 1) + ovh if b != a else 0))
    return makespan(jobs)


if __name__ == "__main__":
    for (B, S) in [(2, 2048), (8, 2048), (4, 512), (8, 371), (1, 2048), (32, 2048)]:
        nbh = B * 32
        bq = 128 if -(-S // 128) * nbh >= 1024 else 64
        nt = -(-S // bq)
        scale = bq // 64
        (m1, i1), (m2, i2) = single(nbh, nt), paired(nbh, nt)
        print(f"B={B} S={S} block {bq}: one block per workgroup {m1 / i1:.2f} x balanced ({nt * nbh} workgroups), pairs {m2 / i2:.2f} x ({(nt + 1) // 2 * nbh})")
