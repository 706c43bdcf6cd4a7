"""The static DPP-hazard checker (tools/check_dpp_hazards.py) that guards the
inline-asm broadcast-FMA blocks: it must flag a VALU write -> DPP read with
fewer than 2 wait states on any path, and accept padded code."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("chk", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

DPP = "v_fmac_f64_dpp v[10:11], v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf"


def _probs(text):
    return chk.check(chk.parse_kernel(text.strip().split("\n")), "k")


def test_flags_back_to_back_write_then_dpp_read():
    assert len(_probs(f"\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\t{DPP}")) == 1
    assert len(_probs(f"\tv_accvgpr_read_b32 v3, a7\n\tv_mov_b32_e32 v40, v41\n\t{DPP}")) == 1


def test_accepts_two_wait_states():
    assert _probs(f"\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_nop 1\n\t{DPP}") == []
    assert _probs(f"\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\tv_mov_b32_e32 v40, v41\n\ts_nop 0\n\t{DPP}") == []
    # one s_nop 0 is a single wait state: still a hazard
    assert len(_probs(f"\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_nop 0\n\t{DPP}")) == 1


def test_ignores_non_valu_writers_and_other_registers():
    assert _probs(f"\tds_read_b64 v[2:3], v9\n\t{DPP}") == []
    assert _probs(f"\tglobal_load_dwordx2 v[2:3], v[20:21], off\n\t{DPP}") == []
    assert _probs(f"\tv_mul_f64 v[4:5], v[6:7], v[8:9]\n\t{DPP}") == []      # src1, not the DPP source


def test_follows_branches_into_labels():
    text = f"""
\tv_mul_f64 v[2:3], v[6:7], v[8:9]
\ts_cbranch_scc1 .LBB0_5
\ts_nop 4
\ts_nop 4
.LBB0_5:
\t{DPP}
"""
    assert len(_probs(text)) == 1          # via the taken branch: only one wait state


def test_flags_vcmpx_exec_write():
    assert len(_probs(f"\tv_cmpx_gt_u32_e32 v1, v2\n\ts_nop 1\n\t{DPP}")) == 1


def test_flags_trans_result_used_by_next_valu():
    """gfx940+: a TRANS result (v_rcp_f64 ...) needs one wait state before a VALU reads it."""
    rcp = "\tv_rcp_f64_e32 v[2:3], v[6:7]"
    assert len(_probs(f"{rcp}\n\tv_fma_f64 v[8:9], v[2:3], v[6:7], v[10:11]")) == 1
    assert len(_probs(f"{rcp}\n\tv_fmac_f64_e32 v[2:3], v[6:7], v[10:11]")) == 1     # accumulator read
    assert _probs(f"{rcp}\n\ts_nop 0\n\tv_fma_f64 v[8:9], v[2:3], v[6:7], v[10:11]") == []
    assert _probs(f"{rcp}\n\tv_mul_f64 v[20:21], v[6:7], v[6:7]\n\tv_fma_f64 v[8:9], v[2:3], v[6:7], v[10:11]") == []
    assert _probs(f"{rcp}\n\tv_fma_f64 v[8:9], v[12:13], v[6:7], v[10:11]") == []     # unrelated registers


def test_shipped_kernels_are_hazard_free():
    """Every fused chain kernel of the library as built (the build keeps the device assembly of each
    translation unit: __graft_entry__.build_hip) passes the hazard checker."""
    import glob
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        import pytest
        pytest.skip("no hipcc")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pattern = os.path.join(ROOT, "build", "obj", "*", "*-hip-amdgcn-amd-amdhsa-gfx950.s")
    entry.build_hip()
    if len(glob.glob(pattern)) < 4 + entry.QW16_SLICES or not entry.listings_current():
        entry.build_hip(force=True)  # library from elsewhere (no build/obj): rebuild with listings
    listings = sorted(glob.glob(pattern))
    units = {os.path.basename(os.path.dirname(p)) for p in listings}
    assert {"sip_lqr_amd", "tree_qw16"} | {"qw16_extra_%d" % k for k in range(entry.QW16_SLICES)} <= units
    checked = 0
    for path in listings:
        if "qw16" not in open(path).read():
            continue
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), path,
                              "chain_factor_solve_qw16", "chain_solve_mrhs_qw16", "tree_factor_solve_qw16"], capture_output=True, text=True)
        assert out.returncode == 0, path + "\n" + out.stdout[-2000:]
        checked += 1
    assert checked >= 2 + entry.QW16_SLICES
    # The staged kernels conclude from counted `s_waitcnt vmcnt(N)` that their LDS-DMA has landed
    # (chain_qw16.hpp); a register spill would add scratch loads / stores to that count.
    import re
    staged = 0
    for path in listings:
        text = open(path).read()
        for m in re.finditer(r"\.amdhsa_kernel (\S*chain_factor_solve_qw16\S*)(.*?)\.end_amdhsa_kernel", text, re.S):
            body = text[text.index(m.group(1) + ":"):]
            body = body[:body.index("s_endpgm")]
            if "global_load_lds" not in body:
                continue
            staged += 1
            assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", m.group(2)), m.group(1) + " uses scratch"
    assert staged >= 10


def _wait_probs(text):
    return chk.check_counted_waits(chk.parse_kernel(text.strip().split("\n")), "k")


def test_counted_vmcnt_waits_must_not_reach_an_older_dma_group():
    """`s_waitcnt vmcnt(k)` behind LDS-DMA staging (chain_qw16.hpp): the k youngest vector-memory
    operations may be spill stores and / or the NEWEST DMA group, never part of a group before it."""
    dma = "\tglobal_load_lds_dwordx4 v[2:3], off"
    store = "\tglobal_store_dwordx2 v[4:5], v[6:7], off"
    group_a, group_b = "\n".join([dma] * 2), "\n".join([dma] * 2)
    ok = f"{group_a}\n{store}\n{store}\n{group_b}\n\ts_waitcnt vmcnt(2)\n\tds_read_b64 v[8:9], v1"
    assert _wait_probs(ok) == []                                   # exactly the newest group stays in flight
    assert _wait_probs(ok.replace("vmcnt(2)", "vmcnt(4)")) == []   # ... plus the stores before it
    assert len(_wait_probs(ok.replace("vmcnt(2)", "vmcnt(5)"))) == 1  # one operation of group A too: stale image
    stores = f"{group_a}\n{store}\n{store}\n{store}\n\ts_waitcnt vmcnt(3)"
    assert _wait_probs(stores) == []
    # a branch that skips the stores changes nothing for a wait that only covers the newest group
    branchy = f"{group_a}\n\ts_cbranch_execz .LBB0_1\n{store}\n.LBB0_1:\n{group_b}\n\ts_waitcnt vmcnt(2)"
    assert _wait_probs(branchy) == []
    # known limit of the check: where two groups are ADJACENT on a path (the store between them skipped), a
    # wait that reaches from one into the other is one unbroken DMA run and is not told apart
    assert len(_wait_probs(branchy.replace("vmcnt(2)", "vmcnt(3)"))) == 0
    crossing = f"{group_a}\n{store}\n{group_b}\n{store}\n\ts_waitcnt vmcnt(5)"
    assert len(_wait_probs(crossing)) == 1
    # scratch traffic in an LDS-DMA kernel shifts every count
    assert len(_wait_probs(f"{dma}\n\tscratch_store_dword off, v1, s0\n\ts_waitcnt vmcnt(0)")) == 1
    # kernels without LDS-DMA are not looked at
    assert _wait_probs(f"{store}\n\ts_waitcnt vmcnt(1)") == []
