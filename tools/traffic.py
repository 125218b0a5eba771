"""HBM traffic of a match kernel per launch from the rocprofv3 PMC passes, calibrated (profiles/r02_fetch_calib.txt):

  * FETCH_SIZE counts every read request as 64 B.  A random gather (container probe, doclen, positions) IS one 64-byte
    request (tools/fetch_calib.hip: gather1 / gather4 / probe8k report 64.0 B per access, and reading both halves of a
    128-byte line costs two requests) — counted exactly.  A wide coalesced load (16 B per lane) is a 128-byte request —
    counted at half.  The kernel's own request tallies (bench.py roofline.model_counts) say how many bytes it STREAMED, so
        read bytes = FETCH_SIZE + streamed_bytes / 2.
  * WRITE_SIZE is exact for coalesced stores and scratch spills (write16 / scratch96: 1.000 per byte).

usage: tools/traffic.py <pmc.txt> <bench.json> <kernel substring>  → one traffic.json entry on stdout"""
import json
import re
import sys

pmc, bench, kern = sys.argv[1:4]
per_name = {}
for l in open(pmc):
    m = re.match(r"PMC ([^\t]+)\t(\S+)\s+mean (\S+) over (\d+)", l)          # (PMC-TALLY lines: the tallying instantiation, not the product's)
    if m and kern in m.group(1):
        per_name.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
assert per_name, "the kernel substring %r matches no kernel in %s" % (kern, pmc)
# (a template with a tallying instantiation — <true>, the few profiled batches of bench.py's model pass — beside the product's: the product's is
#  the one with the dispatches)
name = max(per_name, key=lambda n: per_name[n].get("FETCH_SIZE", (0, 0))[1])
names = {name}
vals = {k: v[0] for k, v in per_name[name].items()}
d = json.load(open(bench))
r = d["roofline"]
c = r["model_counts"] or dict(bitmap_words=0, payload_words=0, block_headers=0, aux_words=0)      # (a kernel without a tallying instantiation: counters alone)
streamed = 4 * c["bitmap_words"] + 4 * c["payload_words"] + 12 * c["block_headers"] + 4 * c["aux_words"]
fetch = vals["FETCH_SIZE"] * 1024.0
write = vals["WRITE_SIZE"] * 1024.0
total = fetch + streamed / 2.0 + write
cfg = d["config"]
import os
sha_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xapiand_amd", "csrc", "libxgm.so.sha")
lib_sha = open(sha_path).read().strip() if os.path.exists(sha_path) else None     # the build these counters were taken from (bench.py refuses another)
print(json.dumps({
    "lib_sha": lib_sha,
    "kernel": r["kernel"], "kernel_instantiation": sorted(names)[0], "op": cfg["op"], "docs_per_gpu": cfg["docs_per_gpu"], "top_k": cfg["top_k"],
    "terms": cfg["terms_per_query"] if cfg["op"] != "PHRASE" else 0, "batch": cfg["batch"], "replay_bits": cfg.get("replay_bits", 0), "streamed_bytes_tallied": r["model_counts"] is not None,
    "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb_raw": vals["WRITE_SIZE"], "streamed_bytes_model": streamed,
    "hbm_bytes_per_launch": total,
    "counters": {k: v for k, v in vals.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")},
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean per launch; reads = FETCH_SIZE (64 B per request: exact for the "
            "random gathers) + half of the bytes the kernel streamed with 16-B/lane loads (128-byte requests, tallied at 64: "
            "tools/fetch_calib.hip, profiles/r02_fetch_calib.txt); writes = WRITE_SIZE (calibrated 1:1, mostly the scratch spill)"}))
