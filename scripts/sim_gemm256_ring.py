#!/usr/bin/env python
"""Symbolic check of gemm256.hip's LDS ring protocol (no GPU needed).

Replays the kernel's issue / wait / barrier schedule for one wave (every wave runs the same schedule) and checks
  * every unit a phase reads was retired by a counted vmcnt wait that precedes the barrier opening that phase,
  * the slot it is read from still holds that unit (no later request targeted the slot before the read phase ended),
  * a slot is never re-requested before the barrier that ends the last phase reading its previous content.
Run: python scripts/sim_gemm256_ring.py
"""
NUNITS = 10


def wrap2(x):
    return x + 2 - NUNITS if x + 2 >= NUNITS else x + 2


def simulate(nk):
    fifo = []            # outstanding requests of one wave, oldest first: (kstep, unit, slot)
    content = {}         # slot -> (kstep, unit) landed AND retired by a wait
    inflight = {}        # slot -> (kstep, unit) requested, not yet retired
    readers = {}         # slot -> phase index until which the slot is read (exclusive end = barrier index)

    def req(k, u, slot, now):
        assert slot not in inflight, f"slot {slot} requested twice (k={k} u={u})"
        assert readers.get(slot, -1) <= now, f"slot {slot} re-requested at phase {now} while still read until {readers[slot]}"
        inflight[slot] = (k, u)
        content.pop(slot, None)
        n = 2                      # two DMA instructions per unit and wave
        for _ in range(n):
            fifo.append((k, u, slot))

    def issue_TW(k, u0, now):
        u2 = wrap2(u0)
        req(k, 0, u0, now); req(k, 1, u0 + 1, now); req(k, 2, u2, now); req(k, 3, u2 + 1, now)

    def issue_B(k, u0, now):
        u4 = wrap2(wrap2(u0))
        req(k, 4, u4, now); req(k, 5, u4 + 1, now)

    def vmcnt(n):
        while len(fifo) > n:
            k, u, slot = fifo.pop(0)
            if not any(f[2] == slot for f in fifo):
                content[slot] = inflight.pop(slot)

    def read(k, units, slots, phase_idx):
        for u, s in zip(units, slots):
            assert content.get(s) == (k, u), f"phase {phase_idx}: slot {s} holds {content.get(s)} / in flight {inflight.get(s)}, wanted {(k, u)}"
            readers[s] = phase_idx + 1

    issue_TW(0, 0, 0); issue_B(0, 0, 0); issue_TW(1, 6, 0)
    vmcnt(12)
    u0, ph = 0, 0
    for k in range(nk):
        u2 = wrap2(u0); u4 = wrap2(u2); n0 = wrap2(u4)
        more = k + 1 < nk
        # phase T(k)
        if k > 0 and more:
            issue_TW(k + 1, n0, ph)
        read(k, (0, 1, 2, 3), (u0, u0 + 1, u2, u2 + 1), ph)
        vmcnt(8 if more else 0)
        ph += 1
        # phase B(k)
        if more:
            issue_B(k + 1, n0, ph)
        read(k, (4, 5, 2, 3), (u4, u4 + 1, u2, u2 + 1), ph)
        vmcnt(4)
        ph += 1
        u0 = n0
    vmcnt(0)
    assert not fifo and not inflight


if __name__ == "__main__":
    for nk in (2, 3, 4, 5, 6, 7, 19, 75, 76):
        simulate(nk)
    print("ring protocol ok")
