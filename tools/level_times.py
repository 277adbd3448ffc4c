#!/usr/bin/env python3
"""Per-level duration of the batch FAST/blur launches from a rocprofv3 rocpd database (kernel trace).
usage: level_times.py <results.db> <W> <H> <frames per launch> [kernel substring]
The last 8 launches of the kernel (one per pyramid level, in level order) are listed: us per launch, us per Mpx."""
import sqlite3
import sys


def level_rows(db, W, H, B, needle="fast_blur"):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
    ev = [r for r in c.execute(f"select d.start,d.end,s.kernel_name,d.grid_size_x,d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start")
          if needle in r[2] and r[3] * 1 >= 64 * B // 8]
    ev = ev[-8:]
    rows = []
    for l, e in enumerate(ev):
        lw, lh = round(W / 1.2 ** l), round(H / 1.2 ** l)
        px = lw * lh * B
        name = e[2].split("ILi")[0].split("aria")[-1].lstrip("0123456789")
        rows.append((l, lw, lh, (e[1] - e[0]) / 1e3, (e[1] - e[0]) / 1e3 / (px / 1e6), name, e[3] // max(e[4], 1)))
    return rows


if __name__ == "__main__":
    W, H, B = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rows = level_rows(sys.argv[1], W, H, B, sys.argv[5] if len(sys.argv) > 5 else "fast_blur")
    tot = 0.0
    for l, lw, lh, us, per, name, wgs in rows:
        tot += us
        print("L%d %4dx%-4d %9.1f us  %.2f us/Mpx  %s  %d workgroups" % (l, lw, lh, us, per, name, wgs))
    print("sum %.1f us = %.3f us/frame" % (tot, tot / B))
