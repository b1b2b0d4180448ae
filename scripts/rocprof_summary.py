#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max."""
import re
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3,
           max(d.end-d.start)/1e3, max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {db}", f"# total kernel time {tot:.2f} ms",
             f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s}"]
    for r in rows:
        name = re.sub(r"\.kd$", "", r[0])[:90]
        lines.append(f"{name:90s} {r[1]:6d} {r[2]:10.3f} {r[3]:10.1f} {r[4]:9.1f} {r[5]:9.1f} {100*r[2]/tot:6.2f} {r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:7d}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
