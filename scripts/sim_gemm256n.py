#!/usr/bin/env python
"""Symbolic check of the K loop of csrc/gemm256n.hip (the two-pass fp16 256x256x64 tile with phases over N): no GPU needed.

Two wave groups run the same instruction stream, the trailing one (waves 4..7 = tile rows 128..255) one half-phase ("slot") behind
the leading one (waves 0..3 = tile rows 0..127); a workgroup barrier separates consecutive slots.  Every wave requests its own 8
rows of each LDS unit by LDS-DMA; a request is only known to have landed once the requesting wave has executed an
`s_waitcnt vmcnt(N)` that retires it (requests retire in order).

LDS: W ring of 3 units (WL(k), WR(k), WL(k+1), ... in slot i % 3), A ring of 7 quarter units (Qj(k) = rows 64j..64j+63, hi + lo
planes, in slot (4k + j) % 7).  Q0, Q1 are READ by the leading waves only, Q2, Q3 by the trailing waves only.

Checked for every read of a unit by the group that performs it:
  * every wave group (both request their rows of every unit) retired its request in a slot BEFORE the slot of the read (so a
    barrier lies between the wait and the read);
  * no request for a different content of the same buffer is issued in a slot <= the read's slot once the content was requested --
    a buffer is only re-requested in a slot AFTER the last read of its old content (requests and reads of one slot are not ordered
    by a barrier).
The schedule restates the table in front of kstep() in csrc/gemm256n.hip; tests/test_gemm256n_protocol_cpu.py runs it for even and
odd K-step counts and checks that plausible-but-wrong variants are rejected (Q3 requested together with Q0-Q2; the end-of-R wait
missing; a 6-slot A ring; WR(k+1) requested at the phase start instead of the half-phase barrier).
Run: python scripts/sim_gemm256n.py
"""


class ProtocolError(AssertionError):
    pass


def program(nk, variant="ok"):
    """One wave's instruction stream as a list of slots; each slot a list of ops.
    Ops: ("req", buf, content, n_instr), ("wait", n), ("read", buf, content, by) with by in {"lead", "trail", "both"}."""
    na = 6 if variant == "ring6" else 7
    slots, cur = [], []

    def barrier():
        nonlocal cur
        slots.append(cur)
        cur = []

    def q(j, k, aq):
        return (f"A{(aq + j) % na}", ("Q%d" % j, k))

    # prologue (every wave, before the skew barrier): WL(0) -> W0, Q0..Q3(0) -> A0..A3, WR(0) -> W1; vmcnt(2); barrier
    cur.append(("req", "W0", ("WL", 0), 2))
    for j in range(4):
        cur.append(("req",) + q(j, 0, 0) + (2,))
    cur += [("req", "W1", ("WR", 0), 2), ("wait", 2)]
    barrier()
    aq, wl = 0, 0
    for k in range(nk):
        kn = min(k + 1, nk - 1)
        wr, wn = (wl + 1) % 3, (wl + 2) % 3
        an = (aq + 4) % na
        readers = {0: "lead", 1: "lead", 2: "trail", 3: "trail"}
        # La(k): fragments of sub-steps 0..2 of A(k), WL(k); requests Q0..Q2(kn) behind sub-step 0; wait vmcnt(6) before the barrier
        for j in range(4):
            cur.append(("read",) + q(j, k, aq) + (readers[j],))
        cur.append(("read", f"W{wl}", ("WL", k), "both"))
        nl = 4 if variant == "q3_early" else (2 if variant == "sched2" else 3)      # A quarters requested in phase L
        for j in range(nl):
            cur.append(("req",) + q(j, kn, an) + (2,))
        if variant == "wr_early" and k > 0:
            pass
        cur.append(("wait", 8 if variant == "q3_early" else (4 if variant == "sched2" else 6)))
        barrier()
        # Lb(k): WL(kn) right after the barrier; fragments of sub-step 3 of A(k), WL(k)
        cur.append(("req", f"W{wn}", ("WL", kn), 2))
        for j in range(4):
            cur.append(("read",) + q(j, k, aq) + (readers[j],))
        cur.append(("read", f"W{wl}", ("WL", k), "both"))
        barrier()
        # Ra(k): WR(k); request Q3(kn) behind sub-step 0; wait vmcnt(2) before the barrier
        cur.append(("read", f"W{wr}", ("WR", k), "both"))
        if variant == "sched2":
            cur.append(("req",) + q(2, kn, an) + (2,))
        if variant != "q3_early":
            cur.append(("req",) + q(3, kn, an) + (2,))
        if variant == "wr_early":
            cur.append(("req", f"W{wl}", ("WR", kn), 2))
        cur.append(("wait", 0 if variant == "q3_early" else (4 if variant == "sched2" else 2)))
        barrier()
        # Rb(k): WR(kn) right after the barrier; WR(k); wait vmcnt(2) before the barrier that ends the phase
        if variant != "wr_early":
            cur.append(("req", f"W{wl}", ("WR", kn), 2))
        cur.append(("read", f"W{wr}", ("WR", k), "both"))
        if variant != "no_end_wait":
            cur.append(("wait", 2))
        barrier()
        aq, wl = an, (wl + 2) % 3
    return slots


def check(nk, variant="ok"):
    prog = program(nk, variant)
    groups = {"lead": 0, "trail": 1}                   # slot offset of the group's K loop (the prologue slot is common)
    events = []                                        # (abs_slot, order, group, op)
    for g, off in groups.items():
        for i, ops in enumerate(prog):
            t = 0 if i == 0 else i + off
            for j, op in enumerate(ops):
                events.append((t, j, g, op))
    retired, issued = {}, {}
    for g in groups:
        fifo = []
        for t, j, gg, op in sorted(e for e in events if e[2] == g):
            if op[0] == "req":
                _, buf, content, n = op
                issued.setdefault((g, buf, content), t)
                fifo += [(buf, content)] * n
            elif op[0] == "wait":
                while len(fifo) > op[1]:
                    buf, content = fifo.pop(0)
                    if (buf, content) not in fifo:
                        retired.setdefault((g, buf, content), t)
    reads = [(t, g, op[1], op[2]) for t, j, g, op in events if op[0] == "read" and op[3] in (g, "both")]
    reqs = sorted((t, g, buf, content) for (g, buf, content), t in issued.items())
    for t, g, buf, content in reads:
        for o in groups:
            r = retired.get((o, buf, content))
            if r is None or r >= t:
                raise ProtocolError(f"nk={nk} {variant}: {g} reads {content} from {buf} in slot {t}, but the {o} waves' share is only "
                                    f"known to have landed in slot {r} (requested in slot {issued.get((o, buf, content))})")
        first_req = min(tt for tt, gg, b, c in reqs if b == buf and c == content)
        for tt, gg, b, c in reqs:
            if b == buf and c != content and first_req < tt <= t:
                raise ProtocolError(f"nk={nk} {variant}: {gg} waves request {c} into {buf} in slot {tt} while {g} still reads {content} in slot {t}")
    return len(reads)


if __name__ == "__main__":
    for nk in (2, 3, 4, 7, 8, 19, 75):
        print(f"nk={nk}: {check(nk)} reads checked, protocol ok; G256N_SCHED=2 (Q2 requested in phase R): {check(nk, 'sched2')} reads ok")
    for bad in ("q3_early", "no_end_wait", "ring6", "wr_early"):
        try:
            check(9, bad)
            print(f"variant {bad}: NOT rejected")
        except ProtocolError as e:
            print(f"variant {bad}: rejected -- {str(e)[:170]}")
