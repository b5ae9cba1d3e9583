#!/usr/bin/env python3
"""Generates frankenstein_amd/csrc/attn_dq_asm.inc: the hand-placed instruction stream of one fully visible tile step of the dQ
attention-backward kernel (bf16, D = 64, 32 queries per wave, 64-key tile = two 32-key halves u = 0, 1), one inline-asm block per ring
slot; same method and scheduler as gen_dkdv_asm.py (fkstream.py).

  MFMA  1.. 8   S'/dP' of half 0 (K / V row fragments from LDS, Q' / dO fragments in registers, row constants as the C operand)
                | gaps: row reads of half 1, transposed K reads
  MFMA  9..16   S'/dP' of half 1        | gaps: exp2 / dS / bf16 packing of half 0
  MFMA 17..20   dQ += K^T dS of half 0  | gaps: exp2 / dS / packing of half 1
  MFMA 21..24   dQ of half 1

Register map (temporaries, listed as clobbers):
  SC0 v[100:115]  DP0 v[116:131]  SC1 v[132:147]  DP1 v[148:163]     scores / dP (accumulators, then dS in place of the scores)
  ROW v[164:195]  four k-steps x (K row fragment 4, V row fragment 4), half 1 reuses the registers of half 0
  TR  v[196:227]  eight transposed K fragments x 4: (u, s, dt)
  PK  v[228:243]  packed dS: (u, s) x 4
Operands: dq0 dq1 (f32x16, read-write), qf0..3 gf0..3 (bf16x8), cl cd (f32x16 row constants), aq0..3 (row-read byte addresses, slot 0),
va0 va1 (transposed-read byte addresses), vo0..3 (per-lane byte offsets of the tile requests), kb vb (64-bit tile bases), ldsw.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fkstream import vr, schedule, clobbers  # noqa: E402

IMG, NS = 64 * 128, 3
SC = [100, 132]
DP = [116, 148]
ROW, TR, PK = 164, 196, 228
LDS_PER_GAP = int(os.environ.get("FK_GEN_LDS_PER_GAP", "2"))
VALU_UNITS = int(os.environ.get("FK_GEN_VALU_UNITS", "8"))
DMA_GAPS = [int(x) for x in os.environ.get("FK_GEN_DMA_GAPS", "2,6,10,14").split(",")]


def tr_reg(u, s, dt):
    return TR + 4 * ((u * 2 + s) * 2 + dt)


def requests(ps):
    """the four LDS-DMA requests of one wave for a tile going to ring slot ps: (set M0, load) pairs"""
    return [(f"s_add_u32 m0, %[ldsw], {ps * IMG}", "global_load_lds_dwordx4 %[vo0], %[kb]"),
            (f"s_add_u32 m0, %[ldsw], {ps * IMG + 1024}", "global_load_lds_dwordx4 %[vo1], %[kb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG}", "global_load_lds_dwordx4 %[vo2], %[vb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG + 1024}", "global_load_lds_dwordx4 %[vo3], %[vb]")]


def gen(slot):
    koff, voff = slot * IMG, (NS + slot) * IMG
    mf = [None]
    for u in range(2):
        for s in range(4):
            mf.append((f"v_mfma_f32_32x32x16_bf16 {vr(SC[u], 16)}, {vr(ROW + 8 * s, 4)}, %[qf{s}], " + ("%[cl]" if s == 0 else vr(SC[u], 16)),
                       [("rk", u, s)], []))
            mf.append((f"v_mfma_f32_32x32x16_bf16 {vr(DP[u], 16)}, {vr(ROW + 8 * s + 4, 4)}, %[gf{s}], " + ("%[cd]" if s == 0 else vr(DP[u], 16)),
                       [("rv", u, s)], []))
    for u in range(2):
        for s in range(2):
            for dt in range(2):
                mf.append((f"v_mfma_f32_32x32x16_bf16 %[dq{dt}], {vr(tr_reg(u, s, dt), 4)}, {vr(PK + 8 * u + 4 * s, 4)}, %[dq{dt}]",
                           [("tr", u, s, dt, t) for t in range(2)], [("c", u, 4 * s + j) for j in range(4)]))
    assert len(mf) == 25
    lds = {}
    for u in range(2):
        for s in range(4):
            rel = 0 if u == 0 else 2 * s + 3                 # half 1 reuses the registers: their two MFMAs (2s+1, 2s+2) and one more are out
            lds[("rk", u, s)] = (f"ds_read_b128 {vr(ROW + 8 * s, 4)}, %[aq{s}] offset:{koff + 4096 * u}", rel, 8 * u + 2 * s + 1)
            lds[("rv", u, s)] = (f"ds_read_b128 {vr(ROW + 8 * s + 4, 4)}, %[aq{s}] offset:{voff + 4096 * u}", rel, 8 * u + 2 * s + 2)
        for s in range(2):
            for dt in range(2):
                for t in range(2):
                    lds[("tr", u, s, dt, t)] = (
                        f"ds_read_b64_tr_b16 {vr(tr_reg(u, s, dt) + 2 * t, 2)}, %[va{dt ^ t}] offset:{koff + 4096 * u + (16 * s + 8 * t) * 128}",
                        0, 17 + 4 * u + 2 * s + dt)
    va = {}
    for u in range(2):
        rel_e, rel_m = 9 + 8 * u, 10 + 8 * u                  # S' of half u is final with MFMA 7 + 8u, dP' with 8 + 8u: two MFMAs on
        for r in range(16):
            dl = 17 + 4 * u + 2 * (r // 8)
            va[("e", u, r)] = (f"v_exp_f32_e32 {vr(SC[u] + r)}, {vr(SC[u] + r)}", 2, rel_e, dl, [])
            va[("m", u, r)] = (f"v_mul_f32_e32 {vr(SC[u] + r)}, {vr(SC[u] + r)}, {vr(DP[u] + r)}", 1, rel_m, dl, [("e", u, r)])
        for sj in range(8):
            s, j = divmod(sj, 4)
            va[("c", u, sj)] = (f"v_cvt_pk_bf16_f32 {vr(PK + 8 * u + 4 * s + j)}, {vr(SC[u] + 8 * s + 2 * j)}, {vr(SC[u] + 8 * s + 2 * j + 1)}", 1, rel_m,
                                17 + 4 * u + 2 * s, [("m", u, 8 * s + 2 * j), ("m", u, 8 * s + 2 * j + 1)])
    dma_at = dict(zip(DMA_GAPS, requests((slot + 2) % NS)))
    if os.environ.get("FK_GEN_ABLATE_DMA"):                      # timing experiments only (wrong results)
        dma_at = {}
    # end of the step: the four requests of tile t + 1 (issued one step ago) have landed, this step's four stay in flight; the barrier
    # publishes tile t + 1 and frees this step's slot for the requests of the step after next
    tail = ["s_waitcnt vmcnt(4)"] + ([] if os.environ.get("FK_GEN_ABLATE_BARRIER") else ["s_barrier"])
    return schedule(mf, lds, va, dma_at, LDS_PER_GAP, VALU_UNITS, tail)


def main():
    out = sys.argv[1]
    clob = clobbers(100, 244)
    req_ops = ('[vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [kb] "s"(kb), [vb] "s"(vb), [ldsw] "s"(ldsw)')
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/gen_dq_asm.py - do not edit.  One fully visible dQ tile step per ring slot (see the generator's header).\n")
        for slot in range(NS):
            ins = gen(slot)
            f.write(f"FK_DEV void dq_tile_asm_slot{slot}(f32x16& dq0, f32x16& dq1, const bf16x8 (&qf)[4], const bf16x8 (&gf)[4], const f32x16& cl, const f32x16& cd,\n"
                    f"                                 const unsigned (&aq)[4], unsigned va0, unsigned va1, const unsigned (&vo)[4], uint64_t kb, uint64_t vb, unsigned ldsw) {{\n")
            f.write("  asm volatile(\n")
            for i in ins:
                f.write(f'      "{i}\\n\\t"\n')
            f.write('      : [dq0] "+v"(dq0), [dq1] "+v"(dq1)\n')
            f.write('      : [qf0] "v"(qf[0]), [qf1] "v"(qf[1]), [qf2] "v"(qf[2]), [qf3] "v"(qf[3]), [gf0] "v"(gf[0]), [gf1] "v"(gf[1]), [gf2] "v"(gf[2]), [gf3] "v"(gf[3]),\n')
            f.write('        [cl] "v"(cl), [cd] "v"(cd), [aq0] "v"(aq[0]), [aq1] "v"(aq[1]), [aq2] "v"(aq[2]), [aq3] "v"(aq[3]), [va0] "v"(va0), [va1] "v"(va1),\n')
            f.write(f"        {req_ops}\n")
            f.write(f"      : {clob}, \"scc\", \"memory\");\n}}\n")
        for slot in range(2):                                  # prologue requests (tiles 0 and 1), as asm for the reason given in gen_dkdv_asm.py
            f.write(f"FK_DEV void dq_request_asm_slot{slot}(const unsigned (&vo)[4], uint64_t kb, uint64_t vb, unsigned ldsw) {{\n  asm volatile(\n")
            for m0, ld in requests(slot):
                f.write(f'      "{m0}\\n\\t"\n      "s_nop 0\\n\\t"\n      "{ld}\\n\\t"\n')
            f.write(f"      :\n      : {req_ops}\n")
            f.write('      : "scc", "memory");\n}\n')
        f.write(f"// instructions per tile step: {len(ins)}\n")
    print(f"{out}: {len(ins)} instructions per tile step")


if __name__ == "__main__":
    main()
