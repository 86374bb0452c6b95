"""tools/isa_resched.py -- the pass every kernel's assembly goes through on its way into libmcq_hip.so (csrc/Makefile).
CPU tests of its three parts: the opcode rewrite, the re-ordering inside a run of vector instructions, and the hazard
distances of gfx950 that the re-ordering must keep (checked on the build's own text when it is there)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_resched as X  # noqa: E402

BUILD = os.path.join(ROOT, "neuron_poker_amd", "csrc", "build")


def test_register_sets():
    assert X.regs_of("s[4:5], 0, v1, vcc_lo, v[10:12] exec") == {"s4", "s5", "v1", "vcc", "v10", "v11", "v12", "exec"}
    r, w = X.rw("v_mad_u64_u32", " v[4:5], s[10:11], v1, v2, v[6:7]")
    assert w == {"v4", "v5", "s10", "s11"} and {"v1", "v2", "v6", "v7"} <= r
    r, w = X.rw("v_cmp_eq_u32_e32", " vcc, 0, v6")
    assert w == {"vcc"} and "v6" in r
    r, w = X.rw("v_cndmask_b32_e32", " v1, v2, v3, vcc")
    assert "vcc" in r and w == {"v1"}
    r, w = X.rw("v_addc_co_u32_e32", " v1, vcc, v2, v3, vcc")
    assert w == {"v1", "vcc"} and "vcc" in r
    # left where they are: partial writes, lane instructions, transcendental ops
    for op, args in (("v_or_b32_sdwa", " v1, v2, v3 dst_sel:DWORD"), ("v_add_u32_dpp", " v1, v2, v3 row_shr:1"),
                     ("v_readfirstlane_b32", " s4, v1"), ("v_rcp_iflag_f32_e32", " v1, v2"), ("v_cmpx_eq_u32_e32", " 0, v1")):
        assert X.rw(op, args) is None, op


def test_bitop3_truth_tables():
    a, b, c = 0xF0F0A5C3, 0xCC33FF00, 0xAA5A0FF0
    def bitop3(x, y, z, tt):
        out = 0
        for i in range(32):
            k = ((x >> i) & 1) << 2 | ((y >> i) & 1) << 1 | ((z >> i) & 1)
            out |= ((tt >> k) & 1) << i
        return out
    assert bitop3(a, b, c, int(X.BITOP3["v_or3_b32"], 16)) == a | b | c
    assert bitop3(a, b, c, int(X.BITOP3["v_and_or_b32"], 16)) == (a & b) | c
    assert bitop3(a, b, c, int(X.BITOP3["v_bfi_b32"], 16)) == (a & b) | (~a & c & 0xFFFFFFFF)
    assert X.to_bitop3("v_or3_b32", " v1, v2, s3, v4") == ("v_bitop3_b32", " v1, v2, s3, v4 bitop3:0xfe")


def test_reorder_keeps_dependencies_and_pads_hazards():
    body = [("v_cmp_ne_u32_e64", " s[4:5], 0, v1"), ("v_add_u32_e32", " v2, v3, v4"), ("v_sub_u32_e32", " v5, v3, v4"),
            ("v_cndmask_b32_e64", " v6, v7, v8, s[4:5]"), ("v_add_u32_e32", " v9, v6, v2")]
    out, st = X.transform(body, bitop3=False, reorder=True, sep=True)
    ops = [o for o, _ in out]
    # the two independent fast instructions lead; the compare's mask is read two wait states behind its write
    assert ops[:2] == ["v_add_u32_e32", "v_sub_u32_e32"] and ops[2] == "v_cmp_ne_u32_e64"
    i_cmp, i_sel = ops.index("v_cmp_ne_u32_e64"), ops.index("v_cndmask_b32_e64")
    between = out[i_cmp + 1:i_sel]
    assert sum(int(a) + 1 if o == "s_nop" else 1 for o, a in between) >= 2
    assert ops.index("v_cndmask_b32_e64") < len(ops) - 1 and ops[-1] == "v_add_u32_e32"
    # a write-after-read inside the run pins the writer behind the reader
    body = [("v_bcnt_u32_b32", " v1, v2, v1"), ("v_add_u32_e32", " v2, v5, v6"), ("v_xor_b32_e32", " v7, v8, v9")]
    out, _ = X.transform(body, bitop3=False, reorder=True, sep=False)
    ops = [o for o, _ in out]
    assert ops.index("v_bcnt_u32_b32") < ops.index("v_add_u32_e32") and ops[0] == "v_xor_b32_e32"
    # nothing crosses a non-vector instruction
    body = [("v_bcnt_u32_b32", " v1, v2, v1"), ("s_waitcnt", " lgkmcnt(0)"), ("v_add_u32_e32", " v3, v5, v6")]
    out, _ = X.transform(body, bitop3=False, reorder=True, sep=True)
    assert [o for o, _ in out] == ["v_bcnt_u32_b32", "s_waitcnt", "v_add_u32_e32"]


def test_hazard_rules():
    st = {}
    out = X.fix_hazards([("v_cmp_eq_u32_e32", " vcc, 0, v6"), ("v_cndmask_b32_e32", " v1, v2, v3, vcc")], st)
    assert out[1] == ("s_nop", "1") and st["hazard_nops"] == 1
    out = X.fix_hazards([("v_mov_b32_e32", " v1, v2"), ("v_add_u32_dpp", " v3, v1, v1 row_shr:1")], {})
    assert out[1] == ("s_nop", "1")
    out = X.fix_hazards([("v_mov_b32_e32", " v1, v2"), ("v_readfirstlane_b32", " s4, v1")], {})
    assert out[1] == ("s_nop", "0")
    out = X.fix_hazards([("v_readfirstlane_b32", " s4, v1"), ("global_load_dword", " v2, v3, s[4:5]")], {})
    assert out[1] == ("s_nop", "4")
    # far enough apart already: nothing added
    body = [("v_cmp_eq_u32_e32", " vcc, 0, v6"), ("v_mov_b32_e32", " v9, v8"), ("s_nop", " 0"), ("v_cndmask_b32_e32", " v1, v2, v3, vcc")]
    assert X.fix_hazards(body, {}) == body


@pytest.mark.skipif(not os.path.exists(os.path.join(BUILD, "mcq_kernels.s")), reason="the library has not been built here")
def test_build_texts_need_no_padding():
    """the rules ask for nothing the compiler has not already provided (so they are not stricter than the hardware's
    documented ones), and the text that went into the library satisfies them too"""
    for name in ("mcq_kernels.s", "mcq_kernels.post.s"):
        with open(os.path.join(BUILD, name)) as f:
            lines = f.readlines()
        _, st = X.process_file(lines, [], bitop3=False, reorder=True, check_only=True)
        assert st.get("hazard_nops", 0) == 0, (name, st)
