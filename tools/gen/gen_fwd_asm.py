#!/usr/bin/env python3
"""Generates frankenstein_amd/csrc/attn_fwd_asm.inc: the hand-placed instruction streams of the lean (reference 0, pre-scaled Q) forward
attention loop (bf16, D = 64, 32 queries per wave, 64-key tiles = two 32-key halves), one inline-asm block per tile step and ring slot
(4 slots); same method and scheduler as gen_dkdv_asm.py (fkstream.py).

The forward step has about as many VALU cycles (exp2, row sums, bf16 packing) as MFMA cycles, so the step is software-pipelined ACROSS
tiles to give every MFMA group its share of them: the exp2 of half 1 of a tile happens at the end of its own step, its row sums / packing
and its P.V MFMAs in the NEXT step (reading V from the previous ring slot, which therefore stays untouched one step longer):

  steady step (tile i in slot s, tile i-1 in slot s-1):
    MFMA  1.. 4   S'(half 0, tile i)          | gaps: row sums + packing of half 1 of tile i-1 (scores carried in the sc1 operand)
    MFMA  5.. 8   O += V^T P (half 1, i-1)    | gaps: exp2 of half 0
    MFMA  9..12   S'(half 1, tile i)          | gaps: row sums + packing of half 0
    MFMA 13..16   O += V^T P (half 0, tile i) | gaps: exp2 of half 1 (left in sc1 for the next step)
  first step: the same without the second group; drain step: only the first gaps' work and the second group.

Row sums go to the l operand, their running maximum to rmax (the caller's overflow check: a step never branches).
Register map (temporaries, listed as clobbers):
  SC0 v[100:115]   KR0 v[116:131] KR1 v[132:147]  K row fragments (4 k-steps x 4) of the two halves
  VTP v[148:163]   V^T fragments (s, dt) x 4 of half 1 of the previous tile     VTC v[164:179]  of half 0 of this tile
  PK0 v[180:187]   PK1 v[188:195]  packed P       RS0 v[196:203]  RS1 v[204:211]  row-sum trees
  SC1 v[212:227]   the carried scores of half 1: an operand pinned to these registers ("+{v[212:227]}"), so every step variant finds
                   them in the same place and hipcc keeps them there between steps
Operands: o0 o1 (f32x16), sc1 (f32x16, pinned), l, rmax (float), qf0..3 (bf16x8), aq0..3 / va0 va1 (LDS byte addresses, slot 0),
vo0..3 (per-lane byte offsets of the tile requests), kb vb (64-bit tile bases), ldsw.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fkstream import vr, schedule, clobbers  # noqa: E402

IMG, NS = 64 * 128, 4
SC0, KR, VTP, VTC, PK, RS, SC1 = 100, [116, 132], 148, 164, [180, 188], [196, 204], 212
LDS_PER_GAP = int(os.environ.get("FK_GEN_LDS_PER_GAP", "2"))
VALU_UNITS = int(os.environ.get("FK_GEN_VALU_UNITS", "9"))


def requests(ps, base="ldsw"):
    return [(f"s_add_u32 m0, %[{base}], {ps * IMG}", "global_load_lds_dwordx4 %[vo0], %[kb]"),
            (f"s_add_u32 m0, %[{base}], {ps * IMG + 1024}", "global_load_lds_dwordx4 %[vo1], %[kb]"),
            (f"s_add_u32 m0, %[{base}], {(NS + ps) * IMG}", "global_load_lds_dwordx4 %[vo2], %[vb]"),
            (f"s_add_u32 m0, %[{base}], {(NS + ps) * IMG + 1024}", "global_load_lds_dwordx4 %[vo3], %[vb]")]


def gen(slot, kind, generic=False):
    """kind: 'steady', 'first' (no previous tile) or 'drain' (only the previous tile's half 1; `slot` = the slot of that tile + 1).
    generic: ONE block for any ring slot — the caller passes the LDS addresses already moved to the slots (aq / va0 / va1: this tile's
    slot; vp0 / vp1: the previous tile's slot; ldsn: the slot of the tile requested by this step), the immediates are those of slot 0.
    Used for the few steps outside the unrolled steady loop (first, aligning, tail, drain), where a switch over four per-slot blocks made
    hipcc move the carried scores and an accumulator through scratch."""
    sp = (slot + NS - 1) % NS
    koff, voff, vpoff = slot * IMG, (NS + slot) * IMG, (NS + sp) * IMG
    vpn = "va"
    if generic:
        koff, voff, vpoff, vpn = 0, NS * IMG, NS * IMG, "vp"
    prev, cur = kind != "first", kind != "drain"
    mf, lds, va = [None], {}, {}
    idx = {}                                                   # group -> index of its first MFMA

    def scn(u, r):
        return vr((SC0 if u == 0 else SC1) + r)

    # ---- MFMA groups in issue order
    if cur:
        idx["s0"] = len(mf)
        for s in range(4):
            mf.append((f"v_mfma_f32_32x32x16_bf16 {vr(SC0, 16)}, {vr(KR[0] + 4 * s, 4)}, %[qf{s}], " + ("0" if s == 0 else vr(SC0, 16)), [("rk", 0, s)], []))
    if prev:
        idx["pv1"] = len(mf)
        for s in range(2):
            for dt in range(2):
                mf.append((f"v_mfma_f32_32x32x16_bf16 %[o{dt}], {vr(VTP + 4 * (2 * s + dt), 4)}, {vr(PK[1] + 4 * s, 4)}, %[o{dt}]",
                           [("tp", s, dt, t) for t in range(2)], [("c", 1, 4 * s + j) for j in range(4)]))
    if cur:
        idx["s1"] = len(mf)
        for s in range(4):
            mf.append((f"v_mfma_f32_32x32x16_bf16 {vr(SC1, 16)}, {vr(KR[1] + 4 * s, 4)}, %[qf{s}], " + ("0" if s == 0 else vr(SC1, 16)), [("rk", 1, s)], []))
        idx["pv0"] = len(mf)
        for s in range(2):
            for dt in range(2):
                mf.append((f"v_mfma_f32_32x32x16_bf16 %[o{dt}], {vr(VTC + 4 * (2 * s + dt), 4)}, {vr(PK[0] + 4 * s, 4)}, %[o{dt}]",
                           [("tc", s, dt, t) for t in range(2)], [("c", 0, 4 * s + j) for j in range(4)]))
    n = len(mf) - 1
    end = n + 1
    # ---- LDS reads
    if cur:
        for u in range(2):
            for s in range(4):
                lds[("rk", u, s)] = (f"ds_read_b128 {vr(KR[u] + 4 * s, 4)}, %[aq{s}] offset:{koff + 4096 * u}", 0, idx["s0" if u == 0 else "s1"] + s)
        for s in range(2):
            for dt in range(2):
                for t in range(2):
                    # this step's V^T fragments are read by the LAST MFMAs of a step: the reads of the next step stay two MFMAs behind them
                    lds[("tc", s, dt, t)] = (f"ds_read_b64_tr_b16 {vr(VTC + 4 * (2 * s + dt) + 2 * t, 2)}, %[va{dt ^ t}] offset:{voff + (16 * s + 8 * t) * 128}",
                                             2, idx["pv0"] + 2 * s + dt)
    if prev:
        for s in range(2):
            for dt in range(2):
                for t in range(2):
                    lds[("tp", s, dt, t)] = (f"ds_read_b64_tr_b16 {vr(VTP + 4 * (2 * s + dt) + 2 * t, 2)}, %[{vpn}{dt ^ t}] offset:{vpoff + 4096 + (16 * s + 8 * t) * 128}",
                                             0, idx["pv1"] + 2 * s + dt)
    # ---- VALU: per half  part a = exp2 (in place), part b = row-sum tree, l, rmax, packing
    def part_a(u, rel, dl):
        for r in range(16):
            va[("e", u, r)] = (f"v_exp_f32_e32 {scn(u, r)}, {scn(u, r)}", 2, rel, dl(r), [])

    def part_b(u, rel, dl_c, dl_sum, have_e):
        dep = (lambda *rs: [("e", u, r) for r in rs]) if have_e else (lambda *rs: [])
        T = RS[u]
        for i in range(8):
            va[("a", u, 0, i)] = (f"v_add_f32_e32 {vr(T + i)}, {scn(u, 2 * i)}, {scn(u, 2 * i + 1)}", 1, rel, dl_sum, dep(2 * i, 2 * i + 1))
        for i in range(4):
            va[("a", u, 1, i)] = (f"v_add_f32_e32 {vr(T + 2 * i)}, {vr(T + 2 * i)}, {vr(T + 2 * i + 1)}", 1, rel, dl_sum, [("a", u, 0, 2 * i), ("a", u, 0, 2 * i + 1)])
        for i in range(2):
            va[("a", u, 2, i)] = (f"v_add_f32_e32 {vr(T + 4 * i)}, {vr(T + 4 * i)}, {vr(T + 4 * i + 2)}", 1, rel, dl_sum, [("a", u, 1, 2 * i), ("a", u, 1, 2 * i + 1)])
        va[("a", u, 3)] = (f"v_add_f32_e32 {vr(T)}, {vr(T)}, {vr(T + 4)}", 1, rel, dl_sum, [("a", u, 2, 0), ("a", u, 2, 1)])
        va[("l", u)] = (f"v_add_f32_e32 %[l], %[l], {vr(T)}", 1, rel, dl_sum, [("a", u, 3)])
        va[("x", u)] = (f"v_max_f32_e32 %[rmax], %[rmax], {vr(T)}", 1, rel, dl_sum, [("a", u, 3)])
        for sj in range(8):
            s, j = divmod(sj, 4)
            va[("c", u, sj)] = (f"v_cvt_pk_bf16_f32 {vr(PK[u] + 4 * s + j)}, {scn(u, 8 * s + 2 * j)}, {scn(u, 8 * s + 2 * j + 1)}", 1, rel, dl_c(s),
                                dep(8 * s + 2 * j, 8 * s + 2 * j + 1))

    if prev:
        part_b(1, 0, lambda s: idx["pv1"] + 2 * s, idx["pv1"] + 4, False)
    if cur:
        rel0 = idx["s0"] + 3 + 2                               # S'(half 0) is final with the group's 4th MFMA: two MFMAs on
        part_a(0, rel0, lambda r: idx["pv0"] + 2 * (r // 8))
        part_b(0, rel0, lambda s: idx["pv0"] + 2 * s, end, True)
        part_a(1, min(idx["s1"] + 3 + 2, n), lambda r: end)
    dma_at = {}
    tail = []
    if cur:
        gaps = [2, 5, 8, 11] if prev else [2, 4, 7, 10]
        dma_at = dict(zip(gaps, requests(0, "ldsn") if generic else requests((slot + 2) % NS)))
        if os.environ.get("FK_GEN_ABLATE_DMA"):                  # timing experiments only (wrong results)
            dma_at = {}
        # end of the step: tile i + 1 (requested one step ago) has landed, this step's requests stay in flight; the barrier publishes it and
        # says every wave is done with tile i - 1's slot (the one the NEXT step's requests overwrite)
        tail = ["s_waitcnt vmcnt(4)"] + ([] if os.environ.get("FK_GEN_ABLATE_BARRIER") else ["s_barrier"])
    return schedule(mf, lds, va, dma_at, LDS_PER_GAP, VALU_UNITS, tail)


def main():
    out = sys.argv[1]
    clob = clobbers(100, 212)
    req_ops = '[vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [kb] "s"(kb), [vb] "s"(vb), [ldsw] "s"(ldsw)'
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/gen_fwd_asm.py - do not edit.  Lean forward tile steps per ring slot (see the generator's header).\n")
        for kind in ("steady",):                               # per-slot blocks: the unrolled steady loop only (first / drain / stray steps: the generic blocks below)
            for slot in range(NS):
                ins = gen(slot, kind)
                body = ins
                f.write(f"FK_DEV void fwd_{kind}_asm_slot{slot}(f32x16& o0, f32x16& o1, f32x16& sc1, float& l, float& rmax, const bf16x8 (&qf)[4],\n"
                        f"    const unsigned (&aq)[4], unsigned va0, unsigned va1, const unsigned (&vo)[4], uint64_t kb, uint64_t vb, unsigned ldsw) {{\n")
                f.write("  asm volatile(\n")
                for i in body:
                    f.write(f'      "{i}\\n\\t"\n')
                # the first step only WRITES the carried scores (its S'(half 1) chain starts from src_c = 0): an output-only operand, so
                # nothing has to be initialised or kept alive in v[212:227] in front of it
                sc1c = '"=&{v[212:227]}"(sc1)' if kind == "first" else '"+{v[212:227]}"(sc1)'
                f.write(f'      : [o0] "+v"(o0), [o1] "+v"(o1), {sc1c}, [l] "+v"(l), [rmax] "+v"(rmax)\n')
                f.write('      : [qf0] "v"(qf[0]), [qf1] "v"(qf[1]), [qf2] "v"(qf[2]), [qf3] "v"(qf[3]),\n')
                f.write('        [aq0] "v"(aq[0]), [aq1] "v"(aq[1]), [aq2] "v"(aq[2]), [aq3] "v"(aq[3]), [va0] "v"(va0), [va1] "v"(va1),\n')
                f.write(f"        {req_ops}\n")
                f.write(f"      : {clob}, \"scc\", \"memory\");\n}}\n")
                f.write(f"// {kind}: {len(ins)} instructions\n")
        for kind in ("steady", "first", "drain"):
            ins = gen(0, kind, generic=True)
            f.write(f"FK_DEV void fwd_{kind}_asm_gen(f32x16& o0, f32x16& o1, f32x16& sc1, float& l, float& rmax, const bf16x8 (&qf)[4],\n"
                    f"    const unsigned (&aq)[4], unsigned va0, unsigned va1, unsigned vp0, unsigned vp1, const unsigned (&vo)[4], uint64_t kb, uint64_t vb, unsigned ldsn) {{\n")
            f.write("  asm volatile(\n")
            for i in ins:
                f.write(f'      "{i}\\n\\t"\n')
            sc1c = '"=&{v[212:227]}"(sc1)' if kind == "first" else '"+{v[212:227]}"(sc1)'
            f.write(f'      : [o0] "+v"(o0), [o1] "+v"(o1), {sc1c}, [l] "+v"(l), [rmax] "+v"(rmax)\n')
            f.write('      : [qf0] "v"(qf[0]), [qf1] "v"(qf[1]), [qf2] "v"(qf[2]), [qf3] "v"(qf[3]),\n')
            f.write('        [aq0] "v"(aq[0]), [aq1] "v"(aq[1]), [aq2] "v"(aq[2]), [aq3] "v"(aq[3]), [va0] "v"(va0), [va1] "v"(va1), [vp0] "v"(vp0), [vp1] "v"(vp1),\n')
            f.write('        [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [kb] "s"(kb), [vb] "s"(vb), [ldsn] "s"(ldsn)\n')
            f.write(f"      : {clob}, \"scc\", \"memory\");\n}}\n")
            f.write(f"// {kind} (any slot): {len(ins)} instructions\n")
        for slot in range(NS):
            f.write(f"FK_DEV void fwd_request_asm_slot{slot}(const unsigned (&vo)[4], uint64_t kb, uint64_t vb, unsigned ldsw) {{\n  asm volatile(\n")
            for m0, ld in requests(slot):
                f.write(f'      "{m0}\\n\\t"\n      "s_nop 0\\n\\t"\n      "{ld}\\n\\t"\n')
            f.write(f"      :\n      : {req_ops}\n")
            f.write('      : "scc", "memory");\n}\n')
    print(f"{out}: written")


if __name__ == "__main__":
    main()
