"""Every kernel variant must give the same answers: the fast paths (probe containers, wave-autonomous
AND / OR kernels, MaxScore pruning) are optimisations of one semantics, so the parity suite is re-run
with each of them switched off through the library's A/B environment switches (read once per
process, hence the subprocesses)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = "random_queries_vs_oracle or other_stripe_widths or edge_cases or batch_equals_single"


PHRASE_SELECT = "phrase or other_stripe_widths or edge_cases"


@pytest.mark.gpu
@pytest.mark.parametrize("switch,select", [("XGM_NO_DENSE", SELECT), ("XGM_NO_ANDW", SELECT + " or phrase"), ("XGM_NO_ORW", SELECT),
                                           ("XGM_NO_PRUNE", SELECT), ("XGM_NO_PHASE_A", SELECT), ("XGM_NO_BOUND_SUM", SELECT),
                                           ("XGM_NO_PHRASEW", PHRASE_SELECT),
                                           # round 3: the conjunction body for all-container queries off (the queue path serves them), its
                                           # narrow document lengths off, and the same body as a kernel of its own (plain and positional)
                                           ("XGM_NO_DENSE_BODY", SELECT), ("XGM_NO_NARROW_DOCLEN", SELECT), ("XGM_DENSE_KERNEL", SELECT + " or phrase"),
                                           # the conjunction kernel finishing its queries itself (the last unit merges) off: the merge launches
                                           ("XGM_NO_FUSED_MERGE", SELECT + " or phrase or sided"),
                                           # round 4: conjunctions led by a long-tail term through the queue path again (no flat arrays); positional
                                           # all-container queries through xgm_andw_kernel's own positional path again
                                           ("XGM_NO_FLAT", SELECT + " or phrase"), ("XGM_NO_DENSE_PHRASE_BODY", PHRASE_SELECT), ("XGM_NO_FLAT_PHRASE", PHRASE_SELECT),
                                           # the disjunction's guess of the k-th weight far too high: every unit must go round again
                                           # below it (second pass) and still skip what the first pass weighed; and a little too high
                                           ("XGM_OR_SEED_SCALE=8", SELECT), ("XGM_OR_SEED_SCALE=1.3", SELECT),
                                           # round 5: the disjunction's terms without containers through the block decode again (no flat arrays); the
                                           # opt-in word-major kernel xgm_orw2_kernel, also with a guess that forces its repair pass
                                           ("XGM_NO_OR_FLAT", SELECT), ("XGM_ORW2=1", SELECT), ("XGM_ORW2=1,XGM_OR_SEED_SCALE=8", SELECT),
                                           # the wave kernels' events recorded around the launch again instead of riding in its dispatch packet
                                           ("XGM_NO_EXT_LAUNCH", SELECT + " or phrase"),
                                           # round 6: the disjunction kernel finishing its queries itself (opt-in: measured slower than the merge launch)
                                           ("XGM_OR_FUSED_MERGE", SELECT),
                                           # round 6: the bound sum's plane count (default: 4 where every query of the batch has 4-8 terms, else 6) forced either way —
                                           # a coarser or finer rounding of the bounds changes which documents are weighed, never the answer
                                           ("XGM_ORW_PLANES=4", SELECT), ("XGM_ORW_PLANES=6", SELECT), ("XGM_ORW_PLANES=4,XGM_OR_SEED_SCALE=8", SELECT)])
def test_parity_with_fast_path_disabled(built, switch, select):
    env = dict(os.environ)
    for one in switch.split(","):
        name, _, val = one.partition("=")
        env[name] = val or "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                        "-k", select, "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s=1:\n%s\n%s" % (switch, r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_replay_in_parallel_segments(built):
    """xgm_search_replay's parallel formulation (segments replayed from the prefix's top k, xgm_replay.hip) on lists far smaller than the
    4 x 4 096 entries it starts at by default: XGM_REPLAY_SEG_MIN=128 sends the counting test's lists through it (the test asserts so)."""
    env = dict(os.environ, XGM_REPLAY_SEG_MIN="128")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_all.py"), "-x", "-q", "-m", "gpu", "-k", "replay_counts",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["XGM_LIST_UNIT_DOCS=8", "XGM_LIST_UNIT_DOCS=8,XGM_LIST_NO_STRIPE_PARTS=1", "XGM_LIST_LPT_ORDER=1", "XGM_NO_LIST_KERNEL=1",
                                    "XGM_COUNT_ARENA_ENTRIES=4096"])
def test_batch_replay_variants(built, switch):
    """The reference's collation inside a batch (tests/test_gpu_frozen_batch.py) under its own switches: listing units of a few documents — every stripe of a
    frequent-term phrase cut into quarters (bits 24.. of a unit's s_end), the look-back between them at work everywhere —, whole stripes only, the work
    list heaviest-first instead of in stripe order, no listing / counting kernels at all (every row answered by the one-query replay when the batch is
    collected), and an arena far too small (the counting units overflow: the queries are counted when the batch is collected)."""
    env = dict(os.environ)
    for one in switch.split(","):
        name, _, val = one.partition("=")
        env[name] = val or "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_frozen_batch.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s:\n%s\n%s" % (switch, r.stdout[-3000:], r.stderr[-2000:])
