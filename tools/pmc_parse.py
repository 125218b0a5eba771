"""Per-kernel mean of every counter in rocprofv3 --pmc output directories (counter_collection.csv).

Keyed on the FULL kernel name (round 4; rounds 2-3 cut it at 40 characters, which merged the product instantiation
`xgm_andw_kernel<uchar, false, 0, false>` with the TALLYING one `<…, true>` that bench.py's byte-count passes launch — a kernel the
product never runs, with its own spills: VERDICT r3 #13).  Output lines: `PMC <name>\t<counter> mean <v> over <n> dispatches`; the
tallying instantiations (last template argument `true`) are listed last under `PMC-TALLY` and ignored by tools/traffic.py."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def is_tally(name):
    m = re.search(r"<(.*)>", name)
    if not m:
        return False
    args = [a.strip() for a in m.group(1).split(",")]
    base = name.replace("void ", "").replace("(anonymous namespace)::", "")
    # which template argument is TALLY: xgm_orw_kernel<TabT, TALLY, FLAT> (round 5: FLAT came after it); the others carry it last
    if base.startswith("xgm_orw_kernel"):
        flag = args[1] if len(args) > 1 else "false"
    elif base.startswith(("xgm_andw_kernel", "xgm_orw2_kernel", "xgm_dense_kernel")):
        flag = args[-1]
    else:
        return False
    return flag in ("true", "1", "(bool)1")


def short(name):
    return re.sub(r"\(.*$", "", name.replace("void ", "").replace("(anonymous namespace)::", "")).strip()


for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            name = short(row.get("Kernel_Name", ""))
            key = (is_tally(name), name, row.get("Counter_Name"))
            acc[key][0] += float(row.get("Counter_Value", 0) or 0)
            acc[key][1] += 1
        for (tally, name, c), (v, n) in sorted(acc.items()):
            print("%s %s\t%-24s mean %.6g over %d dispatches" % ("PMC-TALLY" if tally else "PMC", name, c, v / max(1, n), n))
