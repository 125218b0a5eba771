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


def run_device_tests(emu_lib, args, extra_env=None, timeout=900, and_kernel=False):
    # every query through the workgroup kernels: the wave-autonomous ones use v_readlane under per-lane conditions
    # guard pages behind every device buffer, a canary behind the LDS a launch asked for, a backtrace if a kernel faults
    env = dict(os.environ, XGM_LIB_PATH=emu_lib, XGM_EMU_QUICK="1", XGM_NO_ANDW="1", XGM_NO_ORW="1", XGM_NO_PHRASEW="1", XGM_NO_AND_KERNEL="1",
               XGM_EMU_GUARD="1", XGM_EMU_FAULT_TRACE="1")
    if and_kernel:
        del env["XGM_NO_AND_KERNEL"]           # conjunctions through xgm_and_kernel (its guarded payload loads carry XGM_EMU hooks)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "%s\n%s" % (r.stdout[-3000:], r.stderr[-2000:])
    return r.stdout


def test_value_sorts_under_emulation(emu_lib):
    """xgm_match_sorted_kernel + xgm_search_sorted (written without a GPU at hand) against the pinned oracle: the three sorts,
    both directions, AND / OR / AND_NOT / AND_MAYBE, deep pages, two stripe widths; the ValueCountMatchSpy of the same pass; set_collapse_key by relevance and under the sorts; and, through the Enquire mirror,
    the MSets the compiled reference itself recorded (golden fixtures: docid, weight bits, percentage, sort key at every rank)."""
    out = run_device_tests(emu_lib, [os.path.join("tests", "test_gpu_sorted.py")], {"XGM_RUN_UNVERIFIED": "1"})
    assert "8 passed" in out, out


def test_match_kernels_under_emulation(emu_lib):
    """The emulator's own credentials: device tests that are green on the MI355X are green on it too — the golden fixtures from
    the reference (AND-3, OR-5 top-100, paging, the two-sided operators, nested trees, PHRASE, four shards with the device shard
    merge) and the edge cases, through xgm_match_kernel and the merge kernels."""
    out = run_device_tests(emu_lib, [os.path.join("tests", "test_gpu_parity.py"), "-k", EMU_SELECT])
    assert "passed" in out and "failed" not in out, out


def test_and_workgroup_kernel_under_emulation(emu_lib):
    """Conjunctions through xgm_and_kernel (candidate-driven decode, two waves' blocks in flight): paging golden and edge cases."""
    out = run_device_tests(emu_lib, [os.path.join("tests", "test_gpu_parity.py"), "-k", "edge_cases or (golden_single_shard and and_paging)"], and_kernel=True)
    assert "2 passed" in out, out


EMU_SELECT = "golden or edge_cases"


def test_wave_kernels_under_emulation(emu_lib):
    """The default kernels — xgm_andw_kernel (queue path, xgm_dense_unit, positional filter K6), xgm_orw_kernel, the merge — on the
    golden fixtures of the reference (AND-3 top-10, OR-5 top-100, paging, the two-sided operators, nested trees, PHRASE, four
    shards) and the edge cases, guard pages behind every buffer."""
    env = dict(os.environ, XGM_LIB_PATH=emu_lib, XGM_EMU_QUICK="1", XGM_EMU_GUARD="1", XGM_EMU_FAULT_TRACE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join("tests", "test_gpu_parity.py"),
                        "-k", EMU_SELECT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "8 passed" in r.stdout, "%s\n%s" % (r.stdout[-3000:], r.stderr[-2000:])


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
