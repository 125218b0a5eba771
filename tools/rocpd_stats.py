#!/usr/bin/env python
"""Per-kernel statistics (calls, total/avg/min/max ns, share) from a rocprofv3 `*_results.db`
(rocpd sqlite output of `rocprofv3 --kernel-trace`).  Equivalent to the `--stats` kernel summary;
used when only the database was kept.   usage: rocpd_stats.py results.db [out.csv]"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.group_segment_size, "
                       "s.arch_vgpr_count, s.sgpr_count from %s d join %s s on d.kernel_id = s.id" % (disp, sym)).fetchall()
    agg = collections.defaultdict(list)
    meta = {}
    for n, a, b, gx, gy, wx, lds, vg, sg in rows:
        agg[n].append(b - a)
        meta[n] = (vg, sg, lds, wx)
    tot = sum(sum(v) for v in agg.values()) or 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPRs,SGPRs,LDS_Bytes,WorkgroupSize\n")
    for n, v in sorted(agg.items(), key=lambda x: -sum(x[1])):
        vg, sg, lds, wx = meta[n]
        out.write('"%s",%d,%d,%.0f,%.2f,%d,%d,%s,%s,%s,%s\n' % (n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v), vg, sg, lds, wx))


if __name__ == "__main__":
    main()
