"""Register / scratch / LDS use of every kernel of a HIP source, read from the device assembly's metadata:
    python tools/regs.py xapiand_amd/csrc/xgm_kernels.hip [-DXXX ...] [filter]
Compiles with the product's flags (--cuda-device-only -S) into /tmp and prints one line per kernel."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xapiand_amd import build as B


def main():
    src = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    filt = [a for a in sys.argv[2:] if not a.startswith("-")]
    out = "/tmp/regs_%s.s" % os.path.basename(src)
    flags = [f for f in B.FLAGS if f not in ("-x", "hip")]
    subprocess.check_call([B._hipcc()] + flags + defs + ["-x", "hip", "--cuda-device-only", "-S", src, "-o", out])
    txt = open(out).read()
    if "amdhsa.kernels" not in txt:
        print("no metadata"); return
    meta = txt[txt.index("amdhsa.kernels"):]
    rows = []
    for blk in re.split(r"\n  - \.", meta)[1:]:
        g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        try:
            dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        if filt and not any(f in dem for f in filt):
            continue
        rows.append((dem, g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print("%-70s %5s %5s %6s %5s %6s %8s %6s" % ("kernel", "vgpr", "agpr", "vspill", "sgpr", "sspill", "scratch", "lds"))
    for r in rows:
        print("%-70s %5s %5s %6s %5s %6s %8s %6s" % r)
    # instruction count per kernel
    if "--count" in sys.argv:
        pass


if __name__ == "__main__":
    main()
