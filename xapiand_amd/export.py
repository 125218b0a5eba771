"""Command-line exporter: glass shard (the reference's on-disk format) → device segment, through the native
glass reader of libxgm.so (no Xapian needed, no GPU needed).

    python -m xapiand_amd.export <glass shard dir> <out.seg> [--stripe-bits N]

Prints one JSON line: revision, doccount, lastdocid, total_length, segment_bytes.  The shard must not be
modified while it is read (export a checked-in revision, or hold the shard lock)."""
import argparse
import ctypes as C
import json
import os
import sys

from . import _lib


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m xapiand_amd.export", description=__doc__.split("\n\n")[0])
    ap.add_argument("glass_dir")
    ap.add_argument("segment")
    ap.add_argument("--stripe-bits", type=int, default=0)
    a = ap.parse_args(argv)
    L = _lib.lib()
    rev, dc, ld, tl = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    _lib.check(L.xgm_glass_info(a.glass_dir.encode(), C.byref(rev), C.byref(dc), C.byref(ld), C.byref(tl)))
    _lib.check(L.xgm_segment_build_from_glass(a.glass_dir.encode(), a.stripe_bits, a.segment.encode()))
    print(json.dumps(dict(revision=rev.value, doccount=dc.value, lastdocid=ld.value, total_length=tl.value,
                          segment_bytes=os.path.getsize(a.segment))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
