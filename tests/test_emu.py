"""CPU-only: the library's device code run on the host.  tests/emu builds libxgm_emu.so from the library's OWN sources — the
HIP kernels included — against a stand-in for <hip/hip_runtime.h> that executes a kernel one workgroup at a time with a fiber
per work-item, real barriers and lane-exact wave operations (tests/emu/shim/hip/hip_runtime.h).  The device tests then run
unchanged against that library (XGM_LIB_PATH): their oracle comparisons check the kernels' LOGIC where there is no GPU.  What
emulation cannot show: hardware behaviour (memory ordering between waves, occupancy, timing) — the -m gpu run on an MI355X
stays the parity gate.  Test sizes are small: a workgroup barrier costs 256 fiber switches.

Round 3: the WAVE-autonomous kernels (xgm_andw_kernel incl. xgm_dense_unit and the positional filter, xgm_orw_kernel) run under it
too — the emulated wave operations are marked convergent / noduplicate (the host compiler had duplicated a __ballot into both arms of
a per-lane branch: two call sites, two rendezvous), the kernels read lanes (v_readlane) outside per-lane conditions, and the dense
body fences its ring between consumer and producer — and so does the B1 matcher hook: the reference's own Xapian with the hook
compiled in (oracle/_ref/xapian_hook_b1) loads the emulated library in place of libxgm.so."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not present")


@pytest.fixture(scope="module")
def emu_lib(built):
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU])
    return os.path.join(EMU, "libxgm_emu.so")


EMU_SELECT = "golden or edge_cases"
# every query through the workgroup kernels: the wave-autonomous ones use v_readlane under per-lane conditions
WORKGROUP = dict(XGM_NO_ANDW="1", XGM_NO_ORW="1", XGM_NO_PHRASEW="1", XGM_NO_AND_KERNEL="1")
PARITY, SORTED = os.path.join("tests", "test_gpu_parity.py"), os.path.join("tests", "test_gpu_sorted.py")
# The emulated runs of the device tests: (pytest arguments, extra environment).  Each is a process of its own (the library reads its
# switches once) that keeps ONE core busy for minutes — they are all started when this module's first test begins and run side by
# side, beside the tests that drive the matcher hook; a test then waits for its own (the CPU tier's wall time was the SUM of these runs: 24 of its 28 minutes).
DEVICE_RUNS = {
    "value_sorts": ([SORTED], dict(WORKGROUP, XGM_RUN_UNVERIFIED="1")),
    "match_kernels": ([PARITY, "-k", EMU_SELECT], WORKGROUP),
    # conjunctions through xgm_and_kernel (its guarded payload loads carry XGM_EMU hooks)
    "and_workgroup_kernel": ([PARITY, "-k", "edge_cases or (golden_single_shard and and_paging)"], {k: v for k, v in WORKGROUP.items() if k != "XGM_NO_AND_KERNEL"}),
    "wave_kernels": ([PARITY, "-k", EMU_SELECT], {}),
    "search_all": ([os.path.join("tests", "test_gpu_all.py")], dict(WORKGROUP, XGM_REPLAY_SEG_MIN="128")),      # (the replay also in parallel segments)
    "flat": ([os.path.join("tests", "test_gpu_flat.py")], {}),
    "positional": ([os.path.join("tests", "test_gpu_positional.py"), "-k", "slow_path or colocated"], {}),
    # round 6: the reference's collation INSIDE a batch — positional queries listed and replayed (xgm_andw_list_kernel, xgm_frozen.hip), conjunctions counted
    # (xgm_andw_all_kernel, xgm_count.hip)
    "batch_replay": ([os.path.join("tests", "test_gpu_frozen_batch.py")], {}),
    # ... with listing units of a few documents: every stripe of a frequent-term phrase in quarters, the look-back between units at work everywhere
    "batch_replay_parts": ([os.path.join("tests", "test_gpu_frozen_batch.py"), "-k", "per_query_replay_and_the_reference"], dict(XGM_LIST_UNIT_DOCS="8")),
}


@pytest.fixture(scope="module", autouse=True)
def device_runs(emu_lib, tmp_path_factory):
    d = tmp_path_factory.mktemp("emu_runs")
    procs = {}

    def start_all():
        for name, (args, extra) in DEVICE_RUNS.items():
            # guard pages behind every device buffer, a canary behind the LDS a launch asked for, a backtrace if a kernel faults
            env = dict(os.environ, XGM_LIB_PATH=emu_lib, XGM_EMU_QUICK="1", XGM_EMU_GUARD="1", XGM_EMU_FAULT_TRACE="1")
            env.update(extra)
            out = open(str(d / (name + ".out")), "w")
            procs[name] = (subprocess.Popen([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                                            stdout=out, stderr=subprocess.STDOUT), out)

    start_all()                                        # (autouse: when the module's first test begins; the waiting tests run last, conftest.py)

    def result(name, timeout=3000):
        p, out = procs[name]
        rc = p.wait(timeout=timeout)
        out.close()
        text = open(str(d / (name + ".out"))).read()
        assert rc == 0, text[-4000:]
        return text

    yield result
    for p, out in procs.values():
        if p.poll() is None:
            p.kill()                                   # (the exact processes this fixture started)
        out.close()


def test_value_sorts_under_emulation(device_runs):
    """xgm_match_sorted_kernel + xgm_search_sorted (written without a GPU at hand) against the pinned oracle: the three sorts,
    both directions, AND / OR / AND_NOT / AND_MAYBE, deep pages, two stripe widths; the ValueCountMatchSpy of the same pass; set_collapse_key by relevance and under the sorts; and, through the Enquire mirror,
    the MSets the compiled reference itself recorded (golden fixtures: docid, weight bits, percentage, sort key at every rank)."""
    out = device_runs("value_sorts")
    assert "9 passed" in out, out


def test_match_kernels_under_emulation(device_runs):
    """The emulator's own credentials: device tests that are green on the MI355X are green on it too — the golden fixtures from
    the reference (AND-3, OR-5 top-100, paging, the two-sided operators, nested trees, PHRASE, four shards with the device shard
    merge) and the edge cases, through xgm_match_kernel and the merge kernels."""
    out = device_runs("match_kernels")
    assert "passed" in out and "failed" not in out, out


def test_and_workgroup_kernel_under_emulation(device_runs):
    """Conjunctions through xgm_and_kernel (candidate-driven decode, two waves' blocks in flight): paging golden and edge cases."""
    out = device_runs("and_workgroup_kernel")
    assert "2 passed" in out, out


def test_wave_kernels_under_emulation(device_runs):
    """The default kernels — xgm_andw_kernel (queue path, xgm_dense_unit, positional filter K6), xgm_orw_kernel, the merge — on the
    golden fixtures of the reference (AND-3 top-10, OR-5 top-100, paging, the two-sided operators, nested trees, PHRASE, four
    shards) and the edge cases, guard pages behind every buffer."""
    out = device_runs("wave_kernels")
    assert "8 passed" in out, out


def test_matcher_hook_under_emulation(emu_lib, tmp_path):
    """Seam B1 without a GPU: the compiled reference with integration/xgm_matcher_hook.cc linked in, libxgm.so resolved to the
    emulated library (LD_LIBRARY_PATH precedes the binary's RUNPATH).  Hook on == hook off — docids, weight bits, percentages,
    bounds — for conjunctions, disjunctions, two-sided operators, nested trees, prefix / pattern wildcards and edit-distance
    expansions on a small glass index; most of them answered by the (emulated) device."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import test_gpu_hook_b1 as T
    if not (H.have_xapian_ref() and os.path.exists(T.HOOK_B1)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    alias = tmp_path / "lib"
    alias.mkdir()
    os.symlink(emu_lib, str(alias / "libxgm.so"))
    one = str(tmp_path / "one")
    H.xapian_ref("build", one, hex(H.CORPUS_SEED), 6000, T.VOCAB, 50, 150)
    qs = T.supported_queries()[::3] + T.tree_queries()[::4] + T.wildcard_queries()[:-1:6] + T.expansion_queries()[::3]
    qf = str(tmp_path / "q.txt")
    H.write_queries(qf, qs)
    env = dict(os.environ, LD_LIBRARY_PATH=str(alias) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([T.HOOK_B1, qf, one], capture_output=True, text=True, timeout=1500, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-3000:] + r.stderr[-2000:]
    import json
    out = json.loads(line[-1])
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0, out
    assert out["answered_on_device"] >= len(qs) * 2 // 3, out
    # ... and through Xapiand's per-shard protocol (prepare_mset / add_prepared_mset / get_mset / merge_mset) over three shards
    shards = []
    for sh in range(3):
        p = str(tmp_path / ("shard%d" % sh))
        H.xapian_ref("build", p, hex(H.CORPUS_SEED), 6000, T.VOCAB, 50, 150, 3, sh)
        shards.append(p)
    r = subprocess.run([T.HOOK_B1, qf] + shards, capture_output=True, text=True, timeout=1500, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-3000:] + r.stderr[-2000:]
    out3 = json.loads(line[-1])
    assert out3["mismatches"] == 0 and out3["shards"] == 3 and out3["answered_on_device"] >= len(qs) * 2, out3


def _run_hook_emulated(T, alias, *args, mismatches_expected=False):
    import json
    env = dict(os.environ, LD_LIBRARY_PATH=str(alias) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([T.HOOK_B1] + [str(a) for a in args], capture_output=True, text=True, timeout=1500, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert (r.returncode == 0 or (mismatches_expected and r.returncode == 1)) and line, r.stdout[-3000:] + r.stderr[-2000:]     # (1: hook on != hook off somewhere)
    return json.loads(line[-1])


def test_matcher_hook_modes_under_emulation(emu_lib, tmp_path):
    """The hook's other modes without a GPU, on a small glass index with value slots: value and KEY sorts with ValueCountMatchSpy
    spies (device columns built from the value stream / the KeyMaker); REPLAY + COLLAPSE_REFERENCE (collapse with any collapse_max and
    pages inside the match, cut-offs, spies by relevance: the reference's own collation over the device's match list);
    POSITIONAL_REFERENCE (PHRASE / NEAR pages inside the match: the reference's stale-weight ranking reproduced).  Hook on == hook
    off in every case — sort keys, collapse keys and counts, spy counts, percentages, every match-count figure."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import test_gpu_hook_b1 as T
    if not (H.have_xapian_ref() and os.path.exists(T.HOOK_B1)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    alias = tmp_path / "lib"
    alias.mkdir()
    os.symlink(emu_lib, str(alias / "libxgm.so"))
    n_docs = 6000
    one = str(tmp_path / "one")
    H.xapian_ref("build_values", one, hex(H.CORPUS_SEED), n_docs, T.VOCAB, 50, 150)
    # value / key sorts and spies
    qs, n_sorted = T.sorted_queries()
    qs = qs[:n_sorted:2] + qs[n_sorted::4]
    # ... some of the spies as a MatchSpy class of the application's own, bound through xgm_hook::register_spy_adapter (the way
    # Xapiand's AggregationMatchSpy would be): the driver's DriverCountSpy
    n_custom = 0
    for q in qs:
        if q.get("spy") is not None and n_custom % 2 == 0:
            q["spy_custom"] = True
        n_custom += q.get("spy") is not None
    assert sum(1 for q in qs if q.get("spy_custom")) >= 3
    qf = str(tmp_path / "qs.txt")
    H.write_queries(qf, qs)
    out = _run_hook_emulated(T, alias, qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0 and out["answered_on_device"] == len(qs), out
    assert out["answered_sorted"] >= n_sorted // 2 and out["answered_spied"] >= 6 and out["columns_built"] >= 3, out
    # the same kinds of search from 8 threads under shared sort specs and spy slots: they go out in shared launches (the hook's lanes)
    qs = T.combined_queries()
    qf = str(tmp_path / "qc.txt")
    H.write_queries(qf, qs)
    out = _run_hook_emulated(T, alias, "--threads", "8", "--thread-repeat", "2", qf, one)
    assert out["mismatches"] == 0 and out["threaded_mismatches"] == 0 and out["threaded_answered_on_device"] == 2 * len(qs), out
    assert out["combined_searches"] > 0 and out["combined_launches"] < out["combined_searches"], out
    # replay: collapse / cut-offs / spies by relevance with pages inside the match
    c = H.Corpus(n_docs, T.VOCAB)
    qs = []
    for q in (H.gen_term_queries("AND", 60, 2, 1, 150, maxitems=10, seed=111) + H.gen_term_queries("OR", 40, 2, 100, 2000, maxitems=10, seed=112) +
              H.gen_sided_queries("AND_MAYBE", 20, 1, 1, 50, 1000, maxitems=10, seed=113)):
        m = H.oracle_search(c, q["op"], q["terms"], 0, 1, n_required=q.get("n_required", 0))[1].matches
        if 30 <= m <= 1000 and len(qs) < 24:
            i = len(qs)
            kind = i % 4
            if kind == 0:
                qs.append(dict(q, first=0, maxitems=10, collapse=(i % 3, 1 + (i // 4) % 3)))
            elif kind == 1:
                qs.append(dict(q, first=2, maxitems=8, collapse=((i + 1) % 3, 1), sort=(("VR", "RV")[i % 2], i % 3, False)))
            elif kind == 2:
                qs.append(dict(q, first=0, maxitems=10, cutoff=((40, 0.0), (0, 1.5), (70, 0.5))[(i // 4) % 3]))
            else:
                qs.append(dict(q, first=0, maxitems=10, spy=i % 3))
    assert len(qs) >= 16, len(qs)
    qf = str(tmp_path / "qr.txt")
    H.write_queries(qf, qs)
    out = _run_hook_emulated(T, alias, "--collapse-reference", "--replay", qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0, out
    assert out["replayed"] == len(qs) and out["http_total_equal"] == len(qs), out
    # PHRASE / NEAR pages inside the match: the reference's frozen weights
    qs = []
    for q in (H.gen_phrase_queries(120, n_docs, T.VOCAB, seed=101, lengths=(2, 3)) + H.gen_phrase_queries(40, n_docs, T.VOCAB, seed=103, window_extra=4, op="NEAR")):
        m = H.oracle_search(c, q["op"], q["terms"], 0, 1, window=q.get("window", 0))[1].matches
        if 12 <= m <= 1000 and len(qs) < 16:
            qs.append(dict(q, first=(0, 0, 3)[len(qs) % 3], maxitems=(10, 5, 7)[len(qs) % 3]))
    c.close()
    assert len(qs) >= 8, len(qs)
    qf = str(tmp_path / "qp.txt")
    H.write_queries(qf, qs)
    out = _run_hook_emulated(T, alias, "--positional-reference", "--exact-bounds", qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0 and out["http_total_equal"] == len(qs), out
    assert out["answered_on_device"] == len(qs), out


def test_search_all_under_emulation(device_runs):
    """xgm_search_all (every match in docid order; round 4) against the oracle's full ranking: all operator classes, trees, matches
    beyond one device page — and xgm_search_replay (round 5): ProtoMSet's collation replayed by xgm_replay_kernel over that list (the page,
    known_matching_docs for every operator class and check_at_least; the frozen weight of PHRASE / NEAR of 2-6 terms, two stripe widths)."""
    out = device_runs("search_all")
    assert "6 passed" in out, out


def test_byte_compatible_modes_beyond_one_device_page_under_emulation(emu_lib, tmp_path):
    """The hook's exact bounds / replay / positional-reference modes on matches LARGER than XGM_MAX_K (round 4, xgm_search_all):
    hook on == hook off incl. the HTTP total, without a GPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import test_gpu_hook_b1 as T
    if not (H.have_xapian_ref() and os.path.exists(T.HOOK_B1)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    alias = tmp_path / "lib"
    alias.mkdir()
    os.symlink(emu_lib, str(alias / "libxgm.so"))
    n_docs = 6000
    one = str(tmp_path / "one")
    H.xapian_ref("build", one, hex(H.CORPUS_SEED), n_docs, T.VOCAB, 50, 150)
    c = H.Corpus(n_docs, T.VOCAB)
    plain, positional = T.big_match_queries(c, n_docs, T.VOCAB, 1100, 3)
    c.close()
    assert len(plain) >= 10 and len(positional) >= 4, (len(plain), len(positional))
    qf = str(tmp_path / "qbig.txt")
    H.write_queries(qf, plain)
    out = _run_hook_emulated(T, alias, "--exact-bounds", qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0 and out["answered_on_device"] == len(plain), out
    assert out["http_total_equal"] == len(plain) and out["replayed"] >= 3, out
    qf = str(tmp_path / "qbigpos.txt")
    H.write_queries(qf, positional)
    out = _run_hook_emulated(T, alias, "--positional-reference", "--exact-bounds", qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0 and out["http_total_equal"] == len(positional), out
    assert out["answered_on_device"] == len(positional), out


def test_flat_led_conjunctions_under_emulation(device_runs):
    """xgm_flat_unit and the flat posting arrays (round 4) against the oracle, every path through the tallies, without a GPU."""
    out = device_runs("flat")
    assert "3 passed" in out, out


def test_batch_replay_modes_under_emulation(device_runs):
    """XGM_REPLAY_BATCH_FROZEN / _COUNT without a GPU: the listing units, the look-back between them, the frozen-weight walk per query, the lists of every
    match with their chunks, the scan of the units' top-k lists and the per-unit count — against the one-query replay and the oracle's reference mode."""
    text = device_runs("batch_replay")
    assert " passed" in text and "failed" not in text, text[-2000:]
    text = device_runs("batch_replay_parts")
    assert " passed" in text and "failed" not in text, text[-2000:]


def test_positional_slow_paths_under_emulation(device_runs):
    """The positional bodies' three ways to test a survivor — positions staged 64 documents at a time (<= 16 per term), 16 at a time (<= 64),
    one at a time from a copy in LDS (more, or 4-byte lists) — on hand-made documents, against the oracle, without a GPU."""
    # (colocated: NEAR where several terms share a position — NearPostList's duplicate-position step restated on the device, round 4)
    out = device_runs("positional")
    assert "2 passed" in out, out


def test_near_colocated_switch_through_the_hook_under_emulation(emu_lib, tmp_path):
    """xgm_hook::set_near_colocated_terms(true) without a GPU.  On an index WITHOUT shared positions the shards then run NearPostList's procedure
    in full and the hook's answers stay the reference's (whole matches: no stale weights involved); on documents with several terms per
    position (helpers.coloc_postings) POSITIONAL_INTENDED answers NEAR on the device — and differs from the CPU matcher on part of the
    queries, the reference's history dependence — while the byte-compatible mode keeps those queries on the CPU matcher."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import test_gpu_hook_b1 as T
    if not (H.have_xapian_ref() and os.path.exists(T.HOOK_B1)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    alias = tmp_path / "lib"
    alias.mkdir()
    os.symlink(emu_lib, str(alias / "libxgm.so"))
    one = str(tmp_path / "one")
    H.xapian_ref("build", one, hex(H.CORPUS_SEED), 4000, T.VOCAB, 50, 150)
    # (whole matches inside a page the wave kernels take, k <= 192: a query with more matches than its page would show the stale-weight
    #  quirk of 7.1, not this switch)
    c = H.Corpus(4000, T.VOCAB)
    plain = [dict(q, maxitems=150) for q in H.gen_phrase_queries(40, 4000, T.VOCAB, seed=4, window_extra=2, lengths=(2, 3), op="NEAR")]
    plain = [q for q in plain if 0 < H.oracle_search(c, q["op"], q["terms"], 0, 150, q["window"])[1].matches < 150][:12]
    assert len(plain) >= 8
    qf = str(tmp_path / "qn.txt")
    H.write_queries(qf, plain)
    out = _run_hook_emulated(T, alias, "--near-colocated", qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0 and out["answered_on_device"] == len(plain), out
    post, doclen = H.coloc_postings()
    pf, dbc = str(tmp_path / "p.txt"), str(tmp_path / "dbc")
    H.write_postings_file(pf, post, doclen)
    H.xapian_ref("build_postings", dbc, pf)
    qs = [dict(q, maxitems=150) for q in H.coloc_near_queries() if q["maxitems"] > 10]
    qc = str(tmp_path / "qc.txt")
    H.write_queries(qc, qs)
    intended = _run_hook_emulated(T, alias, "--near-colocated", qc, dbc, mismatches_expected=True)
    assert intended["answered_on_device"] == len(qs) and 0 < intended["mismatches"] <= len(qs), intended
    compat = _run_hook_emulated(T, alias, "--near-colocated", "--positional-reference", "--exact-bounds", qc, dbc)
    assert compat["answered_on_device"] == 0 and compat["mismatches"] == 0 and compat["bounds_violations"] == 0, compat


def test_xapiand_own_keymaker_under_emulation(emu_lib, tmp_path):
    """Xapiand's own Multi_MultiValueKeyMaker (compiled from the reference's sources into the hook driver) in front of the hook, without a GPU:
    hook on == hook off incl. the class's key strings."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import test_gpu_hook_b1 as T
    if not (H.have_xapian_ref() and os.path.exists(T.HOOK_B1)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    alias = tmp_path / "lib"
    alias.mkdir()
    os.symlink(emu_lib, str(alias / "libxgm.so"))
    one = str(tmp_path / "one")
    H.xapian_ref("build_values", one, hex(H.CORPUS_SEED), 6000, T.VOCAB, 50, 150)
    qs = T.xapiand_keymaker_queries()[::2]
    qf = str(tmp_path / "qx.txt")
    H.write_queries(qf, qs)
    out = _run_hook_emulated(T, alias, qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0 and out["answered_on_device"] == len(qs), out
    assert out["answered_sorted"] == len(qs) and out["http_total_equal"] == len(qs), out


def test_xapiand_own_aggregation_spy_under_emulation(emu_lib, tmp_path):
    """Xapiand's own AggregationMatchSpy (src/aggregations/, compiled from the reference's sources into the hook driver) with a `_values`
    aggregation over single- and multi-valued slots, in front of the hook, without a GPU: the device's per-value counts fed to the
    reference's class through the adapter — hook on == hook off on the `_aggregations` object and its wire form."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import test_gpu_hook_b1 as T
    if not (H.have_xapian_ref() and os.path.exists(T.HOOK_B1)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    alias = tmp_path / "lib"
    alias.mkdir()
    os.symlink(emu_lib, str(alias / "libxgm.so"))
    one = str(tmp_path / "one")
    H.xapian_ref("build_values", one, hex(H.CORPUS_SEED), 3000, T.VOCAB, 50, 150)
    qs = T.xapiand_aggregation_queries(3000)[::3]
    qf = str(tmp_path / "qa.txt")
    H.write_queries(qf, qs)
    out = _run_hook_emulated(T, alias, qf, one)
    on_device = sum(q["agg_kind"] <= 4 for q in qs)                 # (kinds 5, 6: two fields / `_median` — declined by the adapter, answered by the CPU matcher)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0, out
    assert out["answered_on_device"] == on_device and out["answered_spied"] == on_device, out


def test_commit_glue_and_http_bodies_under_emulation(emu_lib, tmp_path):
    """The Xapiand side of the seam (integration/xgm_xapiand_glue.cc: what integration/xapiand_shard_hook.patch calls from Shard::commit /
    do_close) without a GPU: full export on the first commit, incremental refresh on the next, searches on a revision the device does not
    hold yet declined; and the search response's body as http_client.cc:2544-2599 forms it, byte-identical hook off vs hook on."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    import test_gpu_hook_b1 as T
    if not (H.have_xapian_ref() and os.path.exists(T.HOOK_B1)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    alias = tmp_path / "lib"
    alias.mkdir()
    os.symlink(emu_lib, str(alias / "libxgm.so"))
    one = str(tmp_path / "one")
    H.xapian_ref("build", one, hex(H.CORPUS_SEED), 3000, T.VOCAB, 50, 150)
    qs = H.gen_term_queries("AND", 8, 2, 1, 100, maxitems=10, seed=223) + H.gen_term_queries("OR", 4, 3, 1, 400, maxitems=10, seed=224)
    qf = str(tmp_path / "qg.txt")
    H.write_queries(qf, qs)
    out = _run_hook_emulated(T, alias, "--commit-glue", "--stale", "--exact-bounds", qf, one)
    assert out["mismatches"] == 0 and out["bounds_violations"] == 0 and out["declined_revision"] >= 8, out
    assert out["glue_full_exports"] == 1 and out["glue_refreshes"] == 1 and out["glue_failures"] == 0, out
    assert out["http_bodies_equal"] == len(qs) and out["answered_on_device"] >= len(qs), out
    # the hook under Xapiand's load shape: 8 threads search (each its own Database handles) WHILE the writer commits and the glue registers the
    # new revision — the index of the readers' revision is replaced under them: shared ownership keeps it alive through their calls,
    # afterwards it is released; every answer equals the CPU matcher's
    two = str(tmp_path / "two")
    shutil.copytree(one, two)
    out = _run_hook_emulated(T, alias, "--commit-glue", "--threads", "8", "--thread-repeat", "2", "--commit-during", "--exact-bounds", qf, two)
    assert out["mismatches"] == 0 and out["threaded_mismatches"] == 0 and out["threaded_queries"] == 2 * len(qs), out
    assert out["commit_during"] and out["glue_refreshes"] == 1 and out["glue_failures"] == 0 and out["glue_released"] >= 1, out
    assert out["threaded_answered_on_device"] + out["declined_revision"] >= 2 * len(qs), out


def test_integration_patches_apply_to_the_reference():
    """integration/matcher_hook.patch (src/xapian/matcher/matcher.cc) and integration/xapiand_shard_hook.patch (src/database/shard.cc)
    apply to the reference's files as they are."""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("the reference is not present")
    for patch, target in (("matcher_hook.patch", "xapian/matcher/matcher.cc"), ("xapiand_shard_hook.patch", "database/shard.cc")):
        r = subprocess.run(["patch", "--dry-run", "-s", "-o", "/dev/null", os.path.join(ref, target), os.path.join(ROOT, "integration", patch)],
                           capture_output=True, text=True)
        assert r.returncode == 0, (patch, r.stdout, r.stderr)
