#!/usr/bin/env python3
"""Generates frankenstein_amd/csrc/attn_dq16_asm.inc: the dQ tile step of gen_dq_asm.py rebuilt on v_mfma_f32_16x16x32_bf16.

Same flop per cycle as 32x32x16, but on random operands the chip holds a higher clock with the small shape (tools/probes/mfma_shape.hip:
1.32 -> 1.47 PFLOP/s with the LDS reads and VALU of a tile step beside the MFMAs, two waves per SIMD).  Layout (lane l: c = l % 16,
g = l / 16; checked by tools/probes/mfma16_layout.hip): A[i = c][k = 8 g + e], B[k = 8 g + e][j = c], D[i = 4 g + r][j = c].

Per wave 32 queries = two 16-query blocks qb; a 64-key tile = two halves u of two 16-key blocks kb; D = 64 = two k-steps ks.
  S'^T(u, kb, qb) = sum_ks  K rows (A: key 16 kb + c, d-chunk 4 ks + tau(g))  x  Q' fragments (B, registers: query 16 qb + c, same chunk)
                    starting from the row constant cl[qb]; tau = [0, 3, 1, 2] makes the 16-row x 4-chunk ds_read_b128 conflict-free in
                    the swizzled image (any assignment of d-chunks to lane groups is legal as long as A and B agree)
  dP'^T likewise from V rows, dO fragments, cd[qb]
  dS^T as the B operand of dQ: lane group g holds key slots e < 4: keys 4 g + e of block kb = 0, e >= 4: keys 4 g + e - 4 of block kb = 1
                    = its own accumulator registers, packed; the A operand K^T(u, db) comes from two ds_read_b64_tr_b16 with the same keys
  dQ^T(db, qb) (d = 16 db + 4 g + r, query 16 qb + c) += K^T(u, db) x dS^T(u, qb)

  MFMA  1..16  S' / dP' of half 0      17..32  of half 1      33..40  dQ of half 0      41..48  dQ of half 1        (16 cycles each)

Register map (temporaries, clobbers): SC0 v[100:115] DP0 v[116:131] SC1 v[132:147] DP1 v[148:163] (block (kb, qb) at + 4 (2 kb + qb)),
ROW v[164:195] (K fragments (kb, ks) at + 4 (2 kb + ks), V fragments 16 further; half 1 reuses them), KT v[196:227] ((u, db) at + 16 u + 4 db),
PK v[228:243] ((u, qb) at + 8 u + 4 qb).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fkstream import vr, schedule, clobbers  # noqa: E402

IMG, NS = 64 * 128, 3
SC = [100, 132]
DP = [116, 148]
ROW, KT, PK = 164, 196, 228
LDS_PER_GAP = int(os.environ.get("FK_GEN_LDS_PER_GAP", "1"))
VALU_UNITS = int(os.environ.get("FK_GEN_VALU_UNITS", "3"))
DMA_GAPS = [int(x) for x in os.environ.get("FK_GEN_DMA_GAPS", "4,12,20,28").split(",")]
MF = "v_mfma_f32_16x16x32_bf16"
PKMUL = os.environ.get("FK_GEN_PKMUL", "0") != "0"


def requests(ps):
    return [(f"s_add_u32 m0, %[ldsw], {ps * IMG}", "global_load_lds_dwordx4 %[vo0], %[kb]"),
            (f"s_add_u32 m0, %[ldsw], {ps * IMG + 1024}", "global_load_lds_dwordx4 %[vo1], %[kb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG}", "global_load_lds_dwordx4 %[vo2], %[vb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG + 1024}", "global_load_lds_dwordx4 %[vo3], %[vb]")]


def gen(slot):
    koff, voff = slot * IMG, (NS + slot) * IMG
    mf = [None]
    fin_s, fin_p, use_k, use_v = {}, {}, {}, {}
    for u in range(2):
        for ks in range(2):
            for kb in range(2):
                for qb in range(2):
                    blk = 4 * (2 * kb + qb)
                    s_dst, p_dst = vr(SC[u] + blk, 4), vr(DP[u] + blk, 4)
                    mf.append((f"{MF} {s_dst}, {vr(ROW + 4 * (2 * kb + ks), 4)}, %[qf{qb}{ks}], " + (f"%[cl{qb}]" if ks == 0 else s_dst), [("rk", u, kb, ks)], []))
                    use_k[(u, kb, ks)] = len(mf) - 1
                    if ks == 1:
                        fin_s[(u, kb, qb)] = len(mf) - 1
                    mf.append((f"{MF} {p_dst}, {vr(ROW + 16 + 4 * (2 * kb + ks), 4)}, %[gf{qb}{ks}], " + (f"%[cd{qb}]" if ks == 0 else p_dst), [("rv", u, kb, ks)], []))
                    use_v[(u, kb, ks)] = len(mf) - 1
                    if ks == 1:
                        fin_p[(u, kb, qb)] = len(mf) - 1
    first_dq = {}
    for u in range(2):
        for db in range(4):
            for qb in range(2):
                mf.append((f"{MF} %[dq{db}{qb}], {vr(KT + 16 * u + 4 * db, 4)}, {vr(PK + 8 * u + 4 * qb, 4)}, %[dq{db}{qb}]",
                           [("tr", u, db, t) for t in range(2)], [("c", u, qb, j) for j in range(4)]))
                first_dq.setdefault((u, qb), len(mf) - 1)
                first_dq.setdefault(("kt", u, db), len(mf) - 1)
    assert len(mf) == 49
    lds = {}
    first_use = {}
    for key, idx in use_k.items():
        pass
    for u in range(2):
        for kb in range(2):
            for ks in range(2):
                # first / last MFMA reading this fragment: qb = 0 comes first, qb = 1 two MFMAs later
                last_k0 = use_k[(0, kb, ks)]
                last_v0 = use_v[(0, kb, ks)]
                first_k = use_k[(u, kb, ks)] - 2
                first_v = use_v[(u, kb, ks)] - 2
                rel_k = 0 if u == 0 else last_k0 + 2            # half 1 overwrites the registers: their readers and two more MFMAs are out
                rel_v = 0 if u == 0 else last_v0 + 2
                lds[("rk", u, kb, ks)] = (f"ds_read_b128 {vr(ROW + 4 * (2 * kb + ks), 4)}, %[aq{ks}] offset:{koff + 4096 * u + 2048 * kb}", rel_k, first_k)
                lds[("rv", u, kb, ks)] = (f"ds_read_b128 {vr(ROW + 16 + 4 * (2 * kb + ks), 4)}, %[aq{ks}] offset:{voff + 4096 * u + 2048 * kb}", rel_v, first_v)
        for db in range(4):
            for t in range(2):
                lds[("tr", u, db, t)] = (f"ds_read_b64_tr_b16 {vr(KT + 16 * u + 4 * db + 2 * t, 2)}, %[va{db}] offset:{koff + 4096 * u + 2048 * t}",
                                         0, first_dq[("kt", u, db)])
    va = {}
    for u in range(2):
        for kb in range(2):
            for qb in range(2):
                blk = 4 * (2 * kb + qb)
                rel_e, rel_m = fin_s[(u, kb, qb)] + 3, fin_p[(u, kb, qb)] + 3      # result of a 4-pass MFMA: three further MFMAs (48 cycles) on
                dl = first_dq[(u, qb)]
                for r in range(4):
                    x, y = SC[u] + blk + r, DP[u] + blk + r
                    va[("e", u, kb, qb, r)] = (f"v_exp_f32_e32 {vr(x)}, {vr(x)}", 2, rel_e, dl, [])
                    if not PKMUL:
                        va[("m", u, kb, qb, r)] = (f"v_mul_f32_e32 {vr(x)}, {vr(x)}, {vr(y)}", 1, rel_m, dl, [("e", u, kb, qb, r)])
                    elif r % 2 == 0:                          # two rows per instruction: the step is bound by instruction issue
                        va[("m", u, kb, qb, r)] = (f"v_pk_mul_f32 {vr(x, 2)}, {vr(x, 2)}, {vr(y, 2)}", 1, rel_m, dl,
                                                   [("e", u, kb, qb, r), ("e", u, kb, qb, r + 1)])
                        va[("m", u, kb, qb, r + 1)] = None
                for h in range(2):                             # key slots 4 kb + 2 h, + 1 of lane group g
                    va[("c", u, qb, 2 * kb + h)] = (f"v_cvt_pk_bf16_f32 {vr(PK + 8 * u + 4 * qb + 2 * kb + h)}, {vr(SC[u] + blk + 2 * h)}, {vr(SC[u] + blk + 2 * h + 1)}",
                                                    1, rel_m, dl, [("m", u, kb, qb, 2 * h)] + ([] if PKMUL else [("m", u, kb, qb, 2 * h + 1)]))
    va = {k: v for k, v in va.items() if v is not None}
    dma_at = dict(zip(DMA_GAPS, requests((slot + 2) % NS)))
    if os.environ.get("FK_GEN_ABLATE_DMA"):
        dma_at = {}
    tail = ["s_waitcnt vmcnt(4)"] + ([] if os.environ.get("FK_GEN_ABLATE_BARRIER") else ["s_barrier"])
    return schedule(mf, lds, va, dma_at, LDS_PER_GAP, VALU_UNITS, tail)


def main():
    out = sys.argv[1]
    clob = clobbers(100, 244)
    req_ops = ('[vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [kb] "s"(kb), [vb] "s"(vb), [ldsw] "s"(ldsw)')
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/gen_dq16_asm.py - do not edit.  One fully visible dQ tile step per ring slot, 16x16x32 MFMAs.\n")
        for slot in range(NS):
            ins = gen(slot)
            f.write(f"FK_DEV void dq16_tile_asm_slot{slot}(f32x4 (&dq)[4][2], const bf16x8 (&qf)[2][2], const bf16x8 (&gf)[2][2], const f32x4 (&cl)[2], const f32x4 (&cd)[2],\n"
                    f"                                   const unsigned (&aq)[2], const unsigned (&va)[4], const unsigned (&vo)[4], uint64_t kb, uint64_t vb, unsigned ldsw) {{\n")
            f.write("  asm volatile(\n")
            for i in ins:
                f.write(f'      "{i}\\n\\t"\n')
            f.write("      : " + ", ".join(f'[dq{db}{qb}] "+v"(dq[{db}][{qb}])' for db in range(4) for qb in range(2)) + "\n")
            f.write("      : " + ", ".join(f'[qf{qb}{ks}] "v"(qf[{qb}][{ks}]), [gf{qb}{ks}] "v"(gf[{qb}][{ks}])' for qb in range(2) for ks in range(2)) + ",\n")
            f.write('        [cl0] "v"(cl[0]), [cl1] "v"(cl[1]), [cd0] "v"(cd[0]), [cd1] "v"(cd[1]), [aq0] "v"(aq[0]), [aq1] "v"(aq[1]),\n')
            f.write('        [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [va3] "v"(va[3]),\n')
            f.write(f"        {req_ops}\n")
            f.write(f"      : {clob}, \"scc\", \"memory\");\n}}\n")
        f.write(f"// instructions per tile step: {len(ins)}\n")
    print(f"{out}: {len(ins)} instructions per tile step")


if __name__ == "__main__":
    main()
