"""Per-kernel diff of two device assembly files: `python tools/isa_diff.py before.s after.s`.

Make the files with the product's flags, device side only:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip --cuda-device-only -S xapiand_amd/csrc/xgm_kernels.hip -o after.s
Functions are compared after normalising basic-block labels and dropping comments, so a kernel is reported identical when its
instructions are; new and missing functions are listed.  Used to show that an edit (a new kernel beside the old ones, a
preprocessor hook, moving a kernel's text into an include) left the code of the kernels already verified on the GPU untouched."""
import re
import sys


def funcs(path):
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
            text = re.sub(r'\.LBB\d+_', '.LBB_', line)
            text = re.sub(r'\.Ltmp\d+', '.Ltmp', text)
            text = re.sub(r';.*$', '', text).rstrip()
            if text:
                out[cur].append(text)
    return out


def main():
    a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
    same = diff = 0
    for k in a:
        if k not in b:
            print("MISSING after:", k[:120])
        elif a[k] == b[k]:
            same += 1
        else:
            diff += 1
            n = sum(1 for x, y in zip(a[k], b[k]) if x != y) + abs(len(a[k]) - len(b[k]))
            print("DIFF", k[:120], len(a[k]), len(b[k]), n)
    print("identical:", same, "different:", diff, "new:", [k[:70] for k in b if k not in a])
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
