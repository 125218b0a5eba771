"""ISA contract of the last-unit merge (xapiand_amd/csrc/xgm_unit_finish.h), checked on the device assembly:

    python tools/isa_contract.py [xapiand_amd/csrc/xgm_kernels.hip]

For every kernel that bumps a query's arrival counter (`flat_atomic_add … sc0` / `global_atomic_add … sc0` returning the old
value at agent scope) the unit's list and header must have been written THROUGH and acknowledged before the bump, and the last
unit must read the other units' lists past its own caches:
  1. between the arrival atomic and the nearest earlier vector-memory STORE there is an `s_waitcnt` with `vmcnt(0)`
     (every path into the atomic's block: the check walks the straight-line text backwards, which is how the compiler lays
     the arrive sequence out — a branch target in between is accepted only if the wait sits after it);
  2. the kernel holds `global_store_dwordx2 … sc1` stores (list + header, 8 bytes each) before the atomic;
  3. after the atomic the kernel holds `global_load_dwordx2 … sc1` loads (the merge of the other units' lists).
A compiler upgrade that drops the wait, the `sc1` on the stores or on the loads fails this check (tests/test_isa_contract.py runs
it in the CPU suite; the MI355X stress test tests/test_gpu_stress.py is the behavioural side)."""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from xapiand_amd import build as B  # noqa: E402


def device_asm(src, defs=()):
    """Device-only assembly of `src` with the product's flags; cached under /tmp by content digest of the inputs."""
    csrc = os.path.dirname(os.path.abspath(src))
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".inc"))] + [os.path.abspath(src)]
    deps += [os.path.join(ROOT, "include", "xgm.h"), os.path.join(ROOT, "tools", "xgm_corpus.h")]
    flags = [f for f in B.FLAGS if f not in ("-x", "hip")] + list(defs)
    dg = B._digest(deps, " ".join(flags))[:16]
    out = "/tmp/xgm_isa_%s_%s.s" % (os.path.basename(src), dg)
    if not os.path.exists(out):
        tmp = out + ".%d" % os.getpid()
        subprocess.check_call([B._hipcc()] + flags + ["-x", "hip", "--cuda-device-only", "-S", src, "-o", tmp], stderr=subprocess.DEVNULL)
        os.replace(tmp, out)
    return out


def functions(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r'^(_Z[\w$.]+):', line)
        if m and cur is None:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            if re.match(r'^\.Lfunc_end\d+:', line):
                cur = None
                continue
            text = re.sub(r';.*$', '', line).strip()
            if text:
                out[cur].append(text)
    return out


ARRIVE = re.compile(r'^(flat|global)_atomic_add\s+v\d+,.*\bsc0\b')          # returns the old value: the arrival counter
STORE = re.compile(r'^(global|flat|buffer)_store_')
THROUGH_STORE = re.compile(r'^global_store_dwordx2\s.*\bsc1\b')
THROUGH_LOAD = re.compile(r'^global_load_dwordx2\s.*\bsc1\b')


def check(funcs):
    """Returns (checked kernels, list of violations)."""
    bad, seen = [], 0
    for name, ins in funcs.items():
        sites = [i for i, t in enumerate(ins) if ARRIVE.match(t)]
        if not sites:
            continue
        seen += 1
        for p in sites:
            waited = False
            for j in range(p - 1, -1, -1):
                t = ins[j]
                if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                    waited = True
                    break
                if STORE.match(t):
                    break
            if not waited:
                bad.append("%s: arrival atomic at instruction %d is not preceded by s_waitcnt vmcnt(0) after the last store" % (name, p))
            if not any(THROUGH_STORE.match(t) for t in ins[:p]):
                bad.append("%s: no write-through (sc1) dwordx2 store before the arrival atomic at %d" % (name, p))
            if not any(THROUGH_LOAD.match(t) for t in ins[p:]):
                bad.append("%s: no sc1 dwordx2 load after the arrival atomic at %d" % (name, p))
    return seen, bad


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "xapiand_amd", "csrc", "xgm_kernels.hip")
    seen, bad = check(functions(device_asm(src)))
    print("kernels with an arrival counter: %d; violations: %d" % (seen, len(bad)))
    for b in bad:
        print("  " + b)
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main())
