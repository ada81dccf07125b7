"""The shipped code object, checked where the compiler cannot: the 64-bit DPP FMAs of the resident round kernel are inline asm, so the wait
states between a VALU write of their broadcast operand and the DPP read are the source's business (rk_dpp_settle) - and this test's, which scans the
disassembly of fast-racing_amd/libfrx.so for a violation.  CPU test: llvm-objdump only."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_dpp_hazards as chk  # noqa: E402


def test_the_scanner_sees_a_hazard_and_honours_wait_states():
    bad = ["\tv_cndmask_b32_e32 v11, 0, v3, vcc", "\tv_fmac_f64_dpp v[0:1], v[10:11], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf"]
    assert len(chk.scan(bad)["hazards"]) == 1
    one = [bad[0], "\ts_nop 0", bad[1]]
    assert len(chk.scan(one)["hazards"]) == 1                      # one wait state is not enough
    ok = [bad[0], "\ts_nop 1", bad[1]]
    assert chk.scan(ok)["hazards"] == []
    far = [bad[0], "\tv_mov_b32_e32 v20, v21", "\tv_mov_b32_e32 v22, v23", bad[1]]
    assert chk.scan(far)["hazards"] == []
    other = ["\tv_mul_f64 v[4:5], v[6:7], v[8:9]", bad[1]]          # writes the plain operand, not the DPP one
    assert chk.scan(other)["hazards"] == []


@pytest.mark.skipif(not os.path.exists(chk.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not installed")
def test_no_dpp_read_follows_a_valu_write_of_its_operand_too_closely():
    lib = os.path.join(ROOT, "fast-racing_amd", "libfrx.so")
    assert os.path.exists(lib), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    res = chk.scan(chk.disassemble(lib))
    assert res["inline_asm_dpp_fma"] > 1000, res                    # the dense passes and pass A of the round kernel are there
    assert res["hazards"] == [], res["hazards"][:5]
    assert res["unknown_predecessor_inline_asm"] == 0, res          # every inline-asm DPP FMA has two known wait states in front of it
