#!/usr/bin/env python3
# print one steady-state iteration of tools/host_path_rate.py from a rocprofv3 rocpd database
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
mc = [t for t in tabs if t.startswith('rocpd_memory_copy')][0]
ev = [(r[0], r[1], r[2]) for r in c.execute(f"select d.start,d.end,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id")]
ev += [(r[0], r[1], 'COPY %d' % r[2]) for r in c.execute(f"select start,end,size from {mc}")]
ev.sort()
idx = [i for i, e in enumerate(ev) if 'knn2' in e[2]]
i0 = idx[len(idx) // 2]
base = ev[i0 - 12][0]
for e in ev[i0 - 12:i0 + 4]:
    print("%9.1f %7.1f  %s" % ((e[0] - base) / 1e3, (e[1] - e[0]) / 1e3, e[2][:44]))
