#!/usr/bin/env python3
# per-level duration of the batch FAST/blur launches from a rocprofv3 rocpd database
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); W = int(sys.argv[2]); H = int(sys.argv[3]); B = int(sys.argv[4])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
ev = [(r[0], r[1], r[2], r[3]) for r in c.execute(f"select d.start,d.end,s.kernel_name,d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start") if 'fast_blur_band' in r[2]]
ev = ev[-8:]
w, h = W, H
for l, e in enumerate(ev):
    lw, lh = round(W / 1.2 ** l), round(H / 1.2 ** l)
    px = lw * lh * B
    print("L%d %4dx%-4d %9.1f us  %.2f us/Mpx  %s" % (l, lw, lh, (e[1] - e[0]) / 1e3, (e[1] - e[0]) / 1e3 / (px / 1e6), 'NB1' if 'ELi1EEE' in e[2] else 'NB>1'))
