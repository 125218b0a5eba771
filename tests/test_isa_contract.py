"""The last-unit merge's visibility contract, checked on the device ISA (no GPU needed): see tools/isa_contract.py and
xapiand_amd/csrc/xgm_unit_finish.h.  VERDICT r3 #7 / ADVICE r3 (high): a workgroup-scope release fence alone emits no vmcnt wait
on gfx950, so the wait is explicit and this test keeps it there."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_arrival_counter_is_bumped_after_the_write_through_stores_are_acknowledged():
    import isa_contract as C
    src = os.path.join(C.ROOT, "xapiand_amd", "csrc", "xgm_kernels.hip")
    seen, bad = C.check(C.functions(C.device_asm(src)))
    assert seen >= 8, "expected the fused instantiations of xgm_andw_kernel, saw %d" % seen
    assert not bad, "\n".join(bad)


def test_the_checker_catches_a_missing_wait():
    import isa_contract as C
    ok = ["global_store_dwordx2 v[10:11], v[12:13], off sc1", "s_waitcnt vmcnt(0) lgkmcnt(0)", "flat_atomic_add v6, v[6:7], v3 sc0",
          "global_load_dwordx2 v[1:2], v[3:4], off sc1"]
    assert C.check({"k": ok}) == (1, [])
    no_wait = [ok[0], "s_waitcnt lgkmcnt(0)", ok[2], ok[3]]
    assert len(C.check({"k": no_wait})[1]) == 1
    plain_store = ["global_store_dwordx2 v[10:11], v[12:13], off", ok[1], ok[2], ok[3]]
    assert len(C.check({"k": plain_store})[1]) == 1
    plain_load = [ok[0], ok[1], ok[2], "global_load_dwordx2 v[1:2], v[3:4], off"]
    assert len(C.check({"k": plain_load})[1]) == 1
