#!/usr/bin/env python3
"""Kernel statistics of a rocprofv3 kernel trace, split by LAUNCH SHAPE.

rocprofv3's own --stats merges every dispatch of a kernel into one row, e.g. the 40 batch launches of the FAST/blur kernel
(8192 frames each) with the several hundred single-frame launches of bench.py's host-path leg, so its AverageNs says nothing
about either. This groups by (kernel, grid size, workgroup size) instead.

usage: kernel_stats_by_shape.py <*_kernel_trace.csv | *_results.db> [min total us] > stats_by_shape.csv
columns: kernel, grid (work-items, x times y), workgroup, calls, total_us, mean_us, median_us, min_us, max_us
"""
import csv
import re
import sqlite3
import statistics
import subprocess
import sys
from collections import defaultdict


_CACHE = {}


def short(name):
    """Demangled kernel name without its argument list, e.g. aria::k_fast_blur_band<1, 1>."""
    if name in _CACHE:
        return _CACHE[name]
    out = name
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool, name.replace(".kd", "")], capture_output=True, text=True, timeout=10).stdout.strip() or name
            break
        except (OSError, subprocess.SubprocessError):
            continue
    out = re.sub(r"^void ", "", out)
    depth, cut = 0, len(out)
    for i, ch in enumerate(out):                       # cut at the '(' of the argument list (outside template brackets)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not out[:i].endswith("(anonymous namespace"):
            if out[i:].startswith("(anonymous namespace)"):
                continue
            cut = i
            break
    out = out[:cut].replace("(anonymous namespace)::", "")
    _CACHE[name] = out
    return out


def rows_from(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
        for r in c.execute(f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id"):
            yield r[0], "%dx%d" % (int(r[1]), int(r[2])), int(r[3]), (r[5] - r[4]) / 1e3
    else:
        for r in csv.DictReader(open(path)):
            yield (r["Kernel_Name"], "%dx%d" % (int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1)),
                   int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def main():
    groups = defaultdict(list)
    for name, grid, wg, us in rows_from(sys.argv[1]):
        groups[(short(name), grid, wg)].append(us)
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid_work_items_xy", "workgroup_x", "calls", "total_us", "mean_us", "median_us", "min_us", "max_us"])
    for (k, g, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < floor:
            continue
        w.writerow([k, g, wg, len(v), "%.1f" % sum(v), "%.2f" % (sum(v) / len(v)), "%.2f" % statistics.median(v), "%.2f" % min(v), "%.2f" % max(v)])


if __name__ == "__main__":
    main()
