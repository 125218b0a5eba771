"""The figures DESIGN.md 6 and profiles/README.md quote, read from a round's evidence files:
    python tools/round_table.py gpurun_out/r06    (prefix: <prefix>_bench.json, <prefix>_pmc_*.txt, <prefix>_prof_*/…kernel_stats.csv)"""
import csv, glob, json, sys

pre = sys.argv[1]
d = json.loads(open(pre + "_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("C2", round(d["value"]), "q/s  ms/step", round(d["ms_per_step"], 3), "ms/batch", round(d["ms_per_batch"], 4), "kernel_ms", round(r["kernel_ms"], 4),
      "frac", round(r["frac"], 3), r["basis"], "traffic GB", r["traffic"] and round(r["traffic"] / 1e9, 3), "alg GB", round(r["algorithmic"]["bytes_per_launch"] / 1e9, 2),
      "alg frac", round(r["algorithmic"]["frac"], 2), "p50/p99", round(d["p50_latency_us"]), round(d["p99_latency_us"]), "host ms/batch", d["host_ms_per_batch"])
eb = d.get("exact_bounds_mode", {})
print("C2 exact bounds:", {k: eb.get(k) for k in ("value", "ms_per_batch", "kernel", "kernel_ms", "known_matching_docs_checked_queries", "error")})
for n, o in d.get("other_configs", {}).items():
    if "error" in o:
        print(n, "ERROR", o["error"]); continue
    ro = o["roofline"]
    print(n, round(o["value"]), "q/s ms/batch", round(o["ms_per_batch"], 4), "kernel", ro["kernel"], "kernel_ms", round(ro["kernel_ms"], 4), "frac", round(ro["frac"], 3), ro["basis"],
          "traffic GB", ro["traffic"] and round(ro["traffic"] / 1e9, 3), "alg frac", round(ro["algorithmic"]["frac"], 2), "p50/p99", o["p50_latency_us"] and round(o["p50_latency_us"]), o["p99_latency_us"] and round(o["p99_latency_us"]),
          "parity", o["parity_checked_queries"], o["timed_batch_rows_checked_against_oracle"])
    cb = o.get("cpu_baseline") or {}
    print("   cpu:", cb.get("kind"), cb.get("value"), "all cores:", (cb.get("all_cores") or {}).get("value"), (cb.get("all_cores") or {}).get("threads"), "port:", (cb.get("port_same_queries") or {}).get("value"))
    oq = o.get("one_query_per_call_mode") or {}
    print("   one query per call:", {k: oq.get(k) for k in ("value", "p50_us", "full_pages", "error")}, "concurrent", (oq.get("concurrent") or {}).get("value"))
    for m in ("intended_semantics_mode", "reference_identical_with_exact_count_mode"):
        if m in o:
            print("   ", m, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in o[m].items()})
    if "answers_that_differ_between_the_two_semantics" in o:
        print("    differ:", o["answers_that_differ_between_the_two_semantics"])
c = d.get("cpu_baseline", {})
print("cpu_baseline:", {k: c.get(k) for k in ("kind", "value", "cores", "threads", "p50_ms")}, "all cores", {k: (c.get("all_cores") or {}).get(k) for k in ("value", "threads", "cores")},
      "port", (c.get("port") or {}).get("value"), (c.get("port_all_cores") or {}).get("value"))
print("   sample:", str(c.get("sample"))[:400])
hp = d.get("hook_parity", {})
print("hook_parity:", {k: v for k, v in hp.items() if not isinstance(v, (dict, list))})
for k, v in hp.items():
    if isinstance(v, dict) and "legs" not in k:
        print("   ", k, {x: v.get(x) for x in ("seconds", "segment_bytes", "docs") if x in v})
for leg in hp.get("legs", []) if isinstance(hp.get("legs"), list) else []:
    print("   leg", {x: leg.get(x) for x in ("leg", "mode", "queries", "mismatches", "answered_on_device", "replayed", "cpu_matcher_seconds", "hook_seconds", "threads", "threaded_queries", "threaded_seconds", "threaded_mismatches", "combined_searches")})
print("server_mode:", json.dumps(d.get("server_mode"))[:600])
for f in sorted(glob.glob(pre + "_prof_*/**/*kernel_stats.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "xgm_" in row["Name"] and int(row["Calls"]) > 8:
            print("stats", f.split("_prof_")[1].split("/")[0], row["Name"][row["Name"].index("xgm_"):][:60], "calls", row["Calls"], "avg us", round(float(row["AverageNs"]) / 1e3, 1))
for f in sorted(glob.glob(pre + "_pmc_*.txt")):
    for l in open(f):
        if l.startswith("PMC ") and any(k in l for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")):
            print("pmc", f.split("_pmc_")[1][:-4], " ".join(l.split()[1:]))
for f in sorted(glob.glob(pre + "_bench_*.json")) + sorted(glob.glob(pre + "_n2*.json")):
    try:
        x = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(x["value"]), x["roofline"]["kernel"], round(x["roofline"]["kernel_ms"], 4), "parity", (x.get("cpu_baseline") or {}).get("parity_checked_queries") or x.get("parity_checked_queries"))
    except Exception as e:
        print(f, "unreadable", e)
