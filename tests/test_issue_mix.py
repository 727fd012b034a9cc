"""tools/issue_mix.py sorts a kernel's vector instructions into the issue classes tools/ubench.hip measured and
bench.py prices `valu_issue` against the resulting ceiling: the classification of an assembly line, and the table's plumbing."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import issue_mix  # noqa: E402


@pytest.mark.parametrize("line, cls", [
    ("v_mul_f32_e32 v4, v5, v4", "fast"),
    ("v_add_f32_e32 v1, v8, v1", "fast"),
    ("v_fma_f32 v7, |v4|, v5, -v7", "fast"),
    ("v_mul_f32_e32 v4, 2.0, v4", "fast"),                      # inline constant
    ("v_mov_b32_e32 v3, 0", "fast"),
    ("v_mul_f32_e32 v4, s5, v4", "slow"),                       # a scalar-register operand costs the instruction its speed
    ("v_fma_f32 v7, v4, s5, s6", "slow"),
    ("v_mul_f32_e32 v1, 0x4f7ffffe, v1", "slow"),               # a 32-bit literal travels the scalar operand's way
    ("v_cndmask_b32_e32 v8, v9, v8, vcc", "slow"),
    ("v_mov_b32_dpp v5, v5 row_shr:1 row_mask:0xf bank_mask:0xf", "slow"),
    ("v_pk_mul_f32 v[4:5], v[6:7], v[4:5]", "slow"),            # (four cycles for two results)
    ("v_cmp_lt_f32_e32 vcc, v5, v6", "slow"),
    ("v_min3_f32 v5, v5, v6, v7", "slow"),
    ("v_add_u32_e32 v1, v2, v3", "int"),
    ("v_xor_b32_e32 v1, v2, v3", "int"),
    ("v_add_u32_e32 v1, s4, v3", "slow"),
    ("v_add_u32_e32 v1, 12, v0", "int"),
    ("v_rcp_f32_e32 v1, v1", "trans"),
    ("v_sqrt_f32_e32 v1, v1", "trans"),
])
def test_issue_class_of_an_instruction(line, cls):
    assert issue_mix.classify(line) == cls


def test_kernels_are_found_and_counted():
    asm = "\n".join([
        "\t.text", "_Zk1:", "\tv_mul_f32_e32 v0, v1, v0", "\tv_pk_add_f32 v[0:1], v[2:3], v[0:1]", "\ts_add_i32 s0, s0, 1", "\tds_read_b32 v1, v2",
        "\tglobal_load_dword v1, v[2:3], off", "\tv_rcp_f32_e32 v1, v1", "\ts_endpgm", ".Lfunc_end0:", "\t.amdhsa_kernel _Zk1", "\t.end_amdhsa_kernel"])
    k = issue_mix.kernels_of(asm)
    assert k == {"_Zk1": {"fast": 1, "int": 0, "slow": 1, "trans": 1, "packed": 1, "salu": 2, "lds": 1, "vmem": 1}}


def test_committed_table_matches_the_kernel_sources_and_feeds_bench():
    import bench
    path = os.path.join(ROOT, "profiles", "%s_issue_mix.json" % bench.ROUND)
    doc = json.load(open(path))
    assert doc["csrc_sha16"] == bench.csrc_digest(), "profiles/*_issue_mix.json is stale: run tools/issue_mix.py"
    trace = next(v for k, v in doc["kernels"].items() if k.startswith("mirt::k_rt_trace2<false, false"))      # (the instantiation the loop's frames run)
    assert trace["valu"] == trace["fast"] + trace["int"] + trace["slow"] + trace["trans"]
    assert 0.23 < trace["ceiling"] < 0.43
    assert bench.issue_ceiling("k_rt_trace2") == pytest.approx(trace["ceiling"])
    assert bench.issue_ceiling("no such kernel") == bench.ISSUE_CEILING
