#!/usr/bin/env python
"""Symbolic check of the skewed K loop of gemm256_lo8n.hip (LO8N_SKEW=1): no GPU needed.

Two wave groups run the same instruction stream, the trailing one (waves 4..7) one half-phase ("slot") behind the leading one
(waves 0..3); a workgroup barrier separates consecutive slots.  Every wave requests its own share of each LDS unit by LDS-DMA, except
the single-buffered fp8 weight units, whose two shares per SIMD pair are both requested by the leading wave.  A request is only
known to have landed once the requesting wave has executed an `s_waitcnt vmcnt(N)` that retires it (requests retire in order).

Checked for every read of a unit by either group:
  * every wave that requests a share of that unit retired its request in a slot BEFORE the slot of the read (so a barrier lies
    between the wait and the read);
  * no request for a different content of the same buffer is issued in a slot <= the read's slot once the content was requested:
    the buffer still holds what is read -- equivalently, a buffer is only re-requested in a slot AFTER the last read of its old
    content (requests and reads of one slot are not ordered by a barrier).
The schedule below restates the table in front of kstep() in csrc/gemm256_lo8n.hip; tests/test_lo8n_protocol_cpu.py runs it for
even and odd K-step counts and checks that known-bad variants (the pre-skew request order, both waves requesting the fp8 units with
the leading waves' wait counts, a missing half-phase wait) are rejected.
Run: python scripts/sim_lo8n_skew.py
"""
import itertools


class ProtocolError(AssertionError):
    pass


def program(nk, variant="ok"):
    """One wave's instruction stream as a list of slots; each slot a list of ops.  Ops: ("req", buf, content, n_instr, who),
    ("wait", n), ("read", buf, content).  `who`: "all" (every wave requests its share) / "lead" (leading waves request both shares).
    Buffers: A stage 0/1 (Ahi + A8 together: same request / read times), W ring slots 0..2, fp8 units "8L" / "8R"."""
    slots = []
    cur = []

    def barrier():
        nonlocal cur
        slots.append(cur)
        cur = []

    w8_who = "all" if variant == "w8_all" else "lead"
    w8_n = 1 if variant == "w8_all" else 2
    # prologue (every wave, before the skew barrier): W8L(0), WL(0) -> slot 0, A(0) -> stage 0, WR(0) -> slot 1; vmcnt(2); barrier
    cur += [("req", "8L", ("W8L", 0), w8_n, w8_who), ("req", "W0", ("WL", 0), 2, "all"), ("req", "A0", ("A", 0), 6, "all"),
            ("req", "W1", ("WR", 0), 2, "all"), ("wait", 2)]
    barrier()
    wl = 0
    for k in range(nk):
        kn = min(k + 1, nk - 1)
        st = k & 1
        wr, wn = (wl + 1) % 3, (wl + 2) % 3
        # La(k)
        cur += [("read", f"A{st}", ("A", k)), ("read", f"W{wl}", ("WL", k)), ("req", f"A{st ^ 1}", ("A", kn), 6, "all")]
        if variant == "old_order":                       # the pre-skew order: W ring and fp8 unit requested at the phase start
            cur += [("req", "8R", ("W8R", k), w8_n, w8_who), ("req", f"W{wn}", ("WL", kn), 2, "all")]
        if variant != "no_mid_wait":
            cur += [("wait", 6 if variant != "old_order" else 8)]
        barrier()
        # Lb(k)
        if variant != "old_order":
            cur += [("req", f"W{wn}", ("WL", kn), 2, "all"), ("req", "8R", ("W8R", k), w8_n, w8_who)]
        cur += [("read", f"A{st}", ("A", k)), ("read", f"W{wl}", ("WL", k)), ("read", "8L", ("W8L", k))]
        barrier()
        # Ra(k)
        cur += [("read", f"W{wr}", ("WR", k))]
        if variant == "old_order":
            cur += [("req", "8L", ("W8L", kn), w8_n, w8_who), ("req", f"W{wl}", ("WR", kn), 2, "all")]
        cur += [("wait", 0 if variant != "old_order" else 2)]
        barrier()
        # Rb(k)
        if variant != "old_order":
            cur += [("req", f"W{wl}", ("WR", kn), 2, "all"), ("req", "8L", ("W8L", kn), w8_n, w8_who)]
        cur += [("read", f"W{wr}", ("WR", k)), ("read", "8R", ("W8R", k))]
        barrier()
        wl = (wl + 2) % 3
    return slots


def check(nk, variant="ok"):
    prog = program(nk, variant)
    groups = {"lead": 0, "trail": 1}                   # slot offset of the group's K loop (the prologue slot is common)
    # absolute slot of program slot i for group g: prologue at 0; loop slot j (1-based in prog) at j + offset
    events = []                                        # (abs_slot, order, group, op)
    for g, off in groups.items():
        for i, ops in enumerate(prog):
            t = 0 if i == 0 else i + off
            for j, op in enumerate(ops):
                events.append((t, j, g, op))
    # per group: request retire slots
    retired = {}                                       # (group, buf, content) -> slot in which a wait retired the group's share
    issued = {}                                        # (group, buf, content) -> slot of the request
    for g in groups:
        fifo = []
        for t, j, gg, op in sorted(e for e in events if e[2] == g):
            if op[0] == "req":
                _, buf, content, n, who = op
                if who == "lead" and g != "lead":
                    continue
                issued.setdefault((g, buf, content), t)
                fifo += [(buf, content)] * n
            elif op[0] == "wait":
                while len(fifo) > op[1]:
                    buf, content = fifo.pop(0)
                    if (buf, content) not in fifo:
                        retired.setdefault((g, buf, content), t)
    reads = [(t, g, op[1], op[2]) for t, j, g, op in events if op[0] == "read"]
    reqs = sorted((t, g, buf, content) for (g, buf, content), t in issued.items())
    for t, g, buf, content in reads:
        owners = ["lead"] if buf.startswith("8") and variant != "w8_all" else list(groups)
        for o in owners:
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
    for nk in (2, 3, 4, 19, 75):
        print(f"nk={nk}: {check(nk)} reads checked, protocol ok")
    for bad in ("old_order", "w8_all", "no_mid_wait"):
        try:
            check(8, bad)
            print(f"variant {bad}: NOT rejected")
        except ProtocolError as e:
            print(f"variant {bad}: rejected -- {str(e)[:160]}")
