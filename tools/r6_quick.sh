# round 6: the default line without the reference legs (minutes instead of a quarter of an hour): C2 + the C3 / C5 sub-legs with their modes
cd $GRAFT_REPO_ROOT
tag=${1:-r6_quick}
timeout 1500 python bench.py --no-cpu-baseline --no-hook-parity --ref-docs 0 --steps 10 --threads 0 > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err; tail -c 600 gpurun_out/${tag}.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}.json").read().strip().splitlines()[-1])
print("C2", round(d["value"]), "kernel_ms", d["roofline"]["kernel_ms"])
for n, oc in d.get("other_configs", {}).items():
    if "error" in oc: print(n, oc["error"]); continue
    print(n, round(oc["value"]), oc["roofline"]["kernel"], "kernel_ms", oc["roofline"]["kernel_ms"], "parity", oc["parity_checked_queries"], oc["timed_batch_rows_checked_against_oracle"])
    for m in ("intended_semantics_mode", "reference_identical_with_exact_count_mode"):
        if m in oc: print("   ", m, round(oc[m]["value"]), oc[m]["kernel"], oc[m]["kernel_ms"], "parity", oc[m]["parity_checked_queries"])
    if "answers_that_differ_between_the_two_semantics" in oc: print("    differ", oc["answers_that_differ_between_the_two_semantics"])
    c = oc.get("one_query_per_call_mode", {})
    print("    one per call:", {k: (round(v) if isinstance(v, float) else v) for k, v in c.items() if k in ("value", "p50_us", "p99_us", "error", "full_pages", "parity_checked_queries")}, "concurrent", c.get("concurrent", {}).get("value"))
PY
