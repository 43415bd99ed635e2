"""CPU side of the GEMM dispatcher's coverage argument: enumerate, through the library's own plan query
(dinov2_hip_op_gemm_plan -- nothing is launched, no device needed), every kernel plan that the DINOv2 family can reach -- ViT-S / B / L / g,
batch 1 ... 64, 224 and 518 pixels, with and without register tokens, f16 and bf16, all five GEMMs of a layer plus the patch embedding -- and
assert that each (kernel, epilogue, dtype) is launched by a case of tests/gemm_plan_cases.py::COVERAGE_CASES, which
tests/test_gpu_ops.py::test_gemm_plan_coverage_case_bits runs on the GPU and holds to bit-equality with the small-tile kernel."""
import re

import pytest

from gemm_plan_cases import BF16, COVERAGE_CASES, EPI_GELU, EPI_PLAIN, EPI_QKV, EPI_RESID, F16, LN_COVERAGE_CASES, cases_leaves, reachable, reachable_ln

LEAF = re.compile(r"^(gemm2<(128|192|256)>|gemm2_mixed<256\+192>|gemm4<256>|gemm4_mixed<256\+192>|gemm4_short<(64|96|128)>|small<\d+x\d+,w\dx\d,st\d,ks\d>)$")


def test_every_reachable_plan_is_covered(api):
    need = reachable(api)
    have = cases_leaves(api, COVERAGE_CASES)
    missing = sorted(k for k in need if k not in have)
    assert not missing, f"plans no bit-equality case launches (regenerate tests/gemm_plan_cases.py): {[(k, need[k]) for k in missing][:10]}"
    assert all(LEAF.match(leaf) for leaf, _, _ in need), sorted({leaf for leaf, _, _ in need if not LEAF.match(leaf)})
    # the headline shapes land where DESIGN.md section 3 says they do
    assert api.gemm_plan(F16, EPI_GELU, 43968, 4096, 1024) == "gemm4_mixed<256+192>"
    assert api.gemm_plan(F16, EPI_RESID, 43968, 1024, 4096) == "gemm4_mixed<256+192>"
    assert api.gemm_plan(F16, EPI_RESID, 43968, 1024, 1024) == "gemm2_mixed<256+192>"
    assert api.gemm_plan(F16, EPI_QKV, 43968, 3072, 1024).startswith("gemm4<256>;small<")
    assert api.gemm_plan(F16, EPI_QKV, 1374, 3072, 1024) == "gemm4_short<96>"


def test_cases_are_current_and_not_padded(api):
    """every case still reaches a plan that is needed (a dispatcher change must come with a regenerated list), and none is refused"""
    need = reachable(api)
    for case in COVERAGE_CASES:
        got = cases_leaves(api, [case])
        assert got and got <= set(need), case


def test_every_reachable_ln_fold_plan_is_covered(api):
    """The same for the LN-fold launches (epilogues 6 .. 9; forward() with ln_fold): every (kernel, epilogue, dtype) a model of the family can
    reach with the option on is run by tests/test_gpu_ln_fold.py::test_ln_plan_coverage_case_bits, and the list holds nothing stale."""
    need = set(reachable_ln(api))
    have = cases_leaves(api, LN_COVERAGE_CASES)
    assert not (need - have), sorted(need - have)[:10]
    assert not (have - need), sorted(have - need)[:10]
    assert all(leaf.startswith(("gemm4", "small<")) for leaf, _, _ in need)  # (the LN epilogues do not exist in gemm2.hip)


def test_plan_query_refuses_what_launch_gemm_refuses(api):
    with pytest.raises(ValueError):
        api.gemm_plan(F16, EPI_PLAIN, 100, 256, 100)  # K % 64 != 0
    with pytest.raises(ValueError):
        api.gemm_plan(BF16, EPI_PLAIN, 0, 256, 128)


def test_tuning_switch_changes_the_plan_and_resets(api):
    base = api.gemm_plan(F16, EPI_GELU, 43968, 4096, 1024)
    try:
        api.set_tuning("gemm_gen", 2)
        assert api.gemm_plan(F16, EPI_GELU, 43968, 4096, 1024) == "gemm2_mixed<256+192>"
        api.set_tuning("gemm_tile", 128)
        assert api.gemm_plan(F16, EPI_GELU, 43968, 4096, 1024).startswith("small<")
    finally:
        api.reset_tuning("gemm_gen")
        api.reset_tuning("gemm_tile")
    assert api.gemm_plan(F16, EPI_GELU, 43968, 4096, 1024) == base
    with pytest.raises(ValueError):
        api.set_tuning("no_such_switch", 1)
