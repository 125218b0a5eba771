"""FETCH_SIZE / WRITE_SIZE (KB, mean per dispatch) of tools/fetch_calib's kernels against their known byte / sector
counts → the calibration table quoted in DESIGN.md and used for bench.py's SECTOR constant."""
import csv
import glob
import json
import os
import re
import sys

known = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
pmc = {}
for l in open(sys.argv[2]):
    m = re.match(r"PMC (.{40}) (\S+)\s+mean (\S+) over", l)
    if m:
        pmc[(re.sub(r"[(<].*", "", m.group(1)).strip(), m.group(2))] = float(m.group(3))
dur = {}
for f in glob.glob(os.path.join(sys.argv[3], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[re.sub(r"[(<].*", "", row["Name"]).strip()] = float(row["AverageNs"])
rows = [("stream16", known["stream16_bytes"], "bytes streamed"), ("gather1", known["gather1_accesses"], "1-byte gathers"),
        ("probe8k_spread", known["probe8k_spread_sectors"], "distinct sectors (64 lanes -> 64 lines)"),
        ("probe8k_dense", known["probe8k_dense_sectors"], "distinct sectors (64 lanes -> 8 sectors)"),
        ("gather4", known["gather4_accesses"], "4-byte gathers"), ("gather_pair", known.get("gather_pair_lines", 1), "128-B lines, both halves read"),
        ("write16", known["write16_bytes"], "bytes stored"),
        ("scratch96", known["scratch_lanes"] * 96 * known["scratch_rounds"], "scratch bytes stored (and reloaded)")]
print("%-16s %14s  %-42s %14s %12s %14s %12s %10s" % ("kernel", "known", "unit", "FETCH_SIZE B", "per unit", "WRITE_SIZE B", "per unit", "avg ms"))
for name, n, unit in rows:
    f = pmc.get((name, "FETCH_SIZE"), 0.0) * 1024.0
    w = pmc.get((name, "WRITE_SIZE"), 0.0) * 1024.0
    print("%-16s %14d  %-42s %14.4g %12.3f %14.4g %12.3f %10.3f" % (name, n, unit, f, f / n, w, w / n, dur.get(name, 0.0) / 1e6))
