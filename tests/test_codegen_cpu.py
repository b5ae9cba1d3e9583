"""What hipcc compiled the hand-scheduled kernels into (CPU suite: hipcc cross-compiles gfx950 without a GPU).

The generated attention streams (csrc/attn_*_asm.inc) name their temporaries as clobbers and leave accumulators, fragments and
addresses to hipcc; the ring-buffered GEMMs are written for two waves per SIMD.  A compiler update that spills inside a matrix loop or
needs more than 256 registers would only show up as a slow benchmark: this test holds the device code of attention.hip and gemm.hip to
the bounds DESIGN.md documents, and every source of the library to the no-packed-fp32-half-swap rule (tools/kernel_resources.py prints the whole table).  The LDS budgets are static_asserts in the sources."""
import pytest

from tools import kernel_resources as KR

# kernel-name fragment -> (max VGPRs, min waves/SIMD, max scratch bytes per lane, scratch instructions allowed inside ANY loop that holds MFMAs).
# In every kernel the hot loop (the innermost loop with the most MFMAs) must be free of scratch traffic.  attn_fwd_asm_kernel used to be the
# one kernel with spill traffic inside (cold) loops (88 bytes per lane: what the forward's WRITE_SIZE of 2.3x its output was); since round 4
# its steps outside the steady loop are single any-slot blocks instead of a switch over four, and it has no scratch at all.
BOUNDS = {
    "attn_bwd_dq_asm_kernel": (256, 2, 0, 0),
    "attn_bwd_dkdv_asm_kernel": (256, 2, 20, 0),
    "attn_bwd_dkdvw_asm_kernel": (448, 1, 0, 0),          # one wave per SIMD by design: 245 VGPRs + 192 AGPRs (accumulators, K / V fragments)
    "attn_fwd_asm_kernel": (256, 2, 0, 0),
    "attn_fwd_ps_kernel": (128, 4, 0, 0),
    "attn_bwd_dq_ps_kernel": (256, 2, 0, 0),
    "attn_bwd_dkdv_ps_kernel": (256, 2, 0, 0),
    "gemm_nt_ring2_kernel": (256, 2, 0, 0),
    "gemm_nt_ring192_kernel": (256, 2, 0, 0),
    "gemm_nt_ring_kernel": (256, 2, 0, 0),
    "gemm_tn_big_kernel": (256, 2, 0, 0),
    # the token-on-the-lane kernels of mlp_fused.hip (round 4): the backward chain keeps 192 accumulator registers in the AGPR half at one wave
    # per SIMD (built without -amdgpu-mfma-vgpr-form), the two forward kernels run two waves per SIMD
    "mlp_bwd_fused_kernel": (448, 1, 0, 0),
    "mlp_bwd_fused_asm_kernel": (464, 1, 0, 0),           # + the generated step's pinned operand tuples: 256 VGPRs + 208 AGPRs
    "mlp_up_fused_kernel": (256, 2, 0, 0),
    "qkv_rope_fused_kernel": (256, 2, 0, 0),
}


@pytest.fixture(scope="module")
def table(tmp_path_factory):
    return KR.survey(tmp_path_factory.mktemp("isa"))


@pytest.mark.parametrize("frag", sorted(BOUNDS))
def test_compiled_kernel_stays_inside_its_register_and_spill_budget(table, frag):
    max_vgpr, min_occ, max_spill, max_inloop = BOUNDS[frag]
    hits = {n: r for n, r in table.items() if frag in n}
    assert hits, f"no kernel named *{frag}* in the compiled sources"
    for name, r in hits.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= max_vgpr, (name, r)
        assert r["Occupancy"] >= min_occ, (name, r)
        # SGPR spills go to VGPR lanes (v_writelane), not to memory; only the one-body comparison kernel gemm_nt_ring2_kernel<.., -1>
        # (FK_NT_RING2_GENERIC=1, not used by default) has any
        assert r["ScratchSize"] <= max_spill and r["SGPRs Spill"] <= (24 if "ring2_kernel" in name and "Lin1E" in name else 0), (name, r)
        loops = KR.mfma_loops(r["_asm"], name)
        inloop = max((l[3] for l in loops), default=0)
        assert inloop <= max_inloop, f"{name}: {inloop} scratch loads/stores inside a loop that contains MFMAs"
        hot = KR.hot_loop(loops)
        assert hot is None or hot[3] == 0, f"{name}: scratch traffic in the hot loop {hot}"
        if "asm_kernel" in frag:
            assert hot is not None and hot[2] >= 24, (name, hot)          # the generated tile step(s) are in that loop
        if "mlp_bwd_fused_asm" in frag:
            # the generated step pins dx to a[0:191]: no compiler register traffic with the accumulator half anywhere in the chunk loop
            lines = open(r["_asm"]).read().splitlines()
            start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
            body, inside = [], False
            for l in lines[start + hot[0]:start + hot[1]]:
                if "#ASMSTART" in l:
                    inside = True
                elif "#ASMEND" in l:
                    inside = False
                elif not inside:
                    body.append(l)
            assert not [l for l in body if "v_accvgpr" in l], "compiler moves into / out of the accumulator half inside the chunk loop"
        if "dkdvw" in frag:
            # the wide stream carries values from step to step in pinned registers and keeps accumulators / fragments in the accumulator
            # half: no register traffic (v_mov / v_accvgpr) may appear between the steps of its loop
            lines = open(r["_asm"]).read().splitlines()
            start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
            body = lines[start + hot[0]:start + hot[1]]
            moves = [l for l in body if "v_accvgpr" in l or "v_mov_b" in l]
            assert not moves, moves[:4]
            assert r.get("AGPRs", 0) >= 192


def test_fused_mlp_backward_keeps_its_accumulators_in_the_agpr_half(table):
    """fk_mlp_bwd_fused fits one wave per SIMD only because the 12 dx accumulator tiles live in AGPRs (its source is compiled without the
    vgpr-form flag, build.flags_for): a build-flag or compiler change that moves them would spill or halve nothing silently — it would
    simply not fit 256 VGPRs."""
    hits = {n: r for n, r in table.items() if "mlp_bwd_fused_kernel" in n}
    assert hits
    for name, r in hits.items():
        assert r.get("AGPRs", 0) >= 192 and r["VGPRs"] <= 256, (name, r)


def test_lds_budgets_are_static_asserts():
    """160 KiB of LDS per CU: every large dynamic allocation is checked at compile time in the source itself"""
    from frankenstein_amd import build as B
    att, gem = (B.CSRC / "attention.hip").read_text(), (B.CSRC / "gemm.hip").read_text()
    for sym in ("FWD_ASM_LDS", "DQ_PS_LDS", "DKDV_PS_LDS"):
        assert f"static_assert({sym} <= 160 * 1024" in att, sym
    for sym in ("R2_LDS", "R192_LDS"):
        assert f"static_assert({sym} <= 160 * 1024" in gem, sym
    mlp = (B.CSRC / "mlp_fused.hip").read_text()
    for sym in ("MF_LDS", "MU_LDS", "QK_LDS"):
        assert f"static_assert({sym} <= 160 * 1024" in mlp, sym


def test_no_compiler_packed_fp32_with_half_swaps(table):
    """v_pk_*_f32 with op_sel / op_sel_hi half-swaps is what hipcc's SLP vectorizer makes of scalar fp32 pairs; those chains returned
    wrong low-half results in lanes 48-63 beside a co-resident GEMM.  The build switches the vectorizer off; hand-written
    f32x2 arithmetic (no half-swaps) stays."""
    import re
    from pathlib import Path
    from frankenstein_amd import build as B
    assert "-fno-slp-vectorize" in B.FLAGS
    asms = {str(r["_asm"]) for r in table.values()}
    assert {Path(a).name for a in asms} == {s + ".s" for s in B.SOURCES}, "the guard has to see every source of the library (fk_rope lives in elementwise.hip)"
    for asm in asms:
        bad = [l for l in open(asm) if re.search(r"v_pk_(mul|fma|add)_f32.*op_sel", l)]
        assert not bad, (asm, bad[:3])
