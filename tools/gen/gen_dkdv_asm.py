#!/usr/bin/env python3
"""Generates frankenstein_amd/csrc/attn_dkdv_asm.inc: the hand-placed instruction stream of one fully visible tile step of the dK/dV
attention-backward kernel (bf16, D = 64, 32 keys per wave, 64-query tile = two 32-row halves u = 0, 1) as ONE inline-asm block per ring slot.

Why generated: hipcc schedules this loop as matrix phase -> VALU phase -> matrix phase (or, given one basic block, waits on every LDS read in
front of its MFMA); on a SIMD the phases of the co-resident waves then ADD.  Here the 32 MFMAs of a tile step are a fixed
sequence and every gap behind an MFMA gets its share of the LDS reads and of the VALU work explicitly (earliest-deadline-first over release
/ deadline windows derived from the data flow and the register reuse):

  MFMA  1.. 8   S'/dP' of half 0        | gaps: transposed reads of half 0, row reads of half 1, statistics of half 1
  MFMA  9..16   S'/dP' of half 1        | gaps: exp2 / dS / bf16 packing of half 0
  MFMA 17..24   dV/dK of half 0         | gaps: exp2 / dS / packing of half 1, transposed reads of half 1 (slot by slot as they free up)
  MFMA 25..32   dV/dK of half 1         | gaps: the rest of the packing of half 1

Register map (hard-coded temporaries, listed as clobbers so hipcc keeps its own values out of them):
  SC0 v[100:115]  DP0 v[116:131]  SC1 v[132:147]  DP1 v[148:163]     scores / dP of the two halves (accumulators, then P / dS in place)
  ROW v[164:195]  four k-steps x (Q fragment 4, dO fragment 4); the first 16 are reused for the packed P / dS of half 1
  TR  v[196:227]  eight transposed fragments x 4 (dO^T, Q^T for (s, dt))
  PK  v[228:243]  packed P (s = 0, 1) and dS (s = 0, 1) of half 0
Operands: dk0 dk1 dv0 dv1 (f32x16, read-write), kf0..3 vf0..3 (bf16x8), aq0..3 (row-read byte addresses in the Q image of slot 0),
va0 va1 (transposed-read byte addresses), ast (statistics byte address, slot 0).
LDS reads return in order, so every wait is a counted s_waitcnt lgkmcnt(n) computed by the model below.
Hazards kept by construction: a VALU result is consumed (by VALU or as an MFMA operand) at least 2 instructions later; an MFMA result is
read by VALU only after two further MFMAs; an LDS read overwrites a fragment register only after the MFMA that read it AND one more MFMA
have been issued.
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
VALU_UNITS = int(os.environ.get("FK_GEN_VALU_UNITS", "7"))       # issue budget of a gap in 4-cycle units (exp2 = 2)
DMA_GAPS = [int(x) for x in os.environ.get("FK_GEN_DMA_GAPS", "2,6,10,14,18").split(",")]   # the gaps that carry the 5 LDS-DMA requests


def tr_reg(s, dt, w):
    return TR + 4 * ((s * 2 + dt) * 2 + w)


def requests(ps):
    """the five LDS-DMA requests of one wave for a tile going to ring slot ps: (set M0, load) pairs"""
    return [(f"s_add_u32 m0, %[ldsw], {ps * IMG}", "global_load_lds_dwordx4 %[vo0], %[qb]"),
            (f"s_add_u32 m0, %[ldsw], {ps * IMG + 1024}", "global_load_lds_dwordx4 %[vo1], %[qb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG}", "global_load_lds_dwordx4 %[vo2], %[gb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG + 1024}", "global_load_lds_dwordx4 %[vo3], %[gb]"),
            (f"s_add_u32 m0, %[ldss], {ps * 2 * 64 * 4}", "global_load_lds_dword %[vo4], %[sb]")]


def gen(slot):
    qoff, soff = slot * IMG, slot * 2 * 64 * 4
    goff = qoff + NS * IMG

    # ------------------------------------------------------------------ the 32 MFMAs: (text, LDS reads needed, VALU results needed)
    mf = [None]
    for u in range(2):
        for s in range(4):
            st_sc = [("st", u, 0, g) for g in range(4)] if s == 0 else []
            st_dp = [("st", u, 1, g) for g in range(4)] if s == 0 else []
            mf.append((f"v_mfma_f32_32x32x16_bf16 {vr(SC[u], 16)}, {vr(ROW + 8 * s, 4)}, %[kf{s}], {vr(SC[u], 16)}", [("rq", u, s)] + st_sc, []))
            mf.append((f"v_mfma_f32_32x32x16_bf16 {vr(DP[u], 16)}, {vr(ROW + 8 * s + 4, 4)}, %[vf{s}], {vr(DP[u], 16)}", [("rg", u, s)] + st_dp, []))
    for u in range(2):
        pk = PK if u == 0 else ROW
        for s in range(2):
            for dt in range(2):
                for w, acc in ((0, "dv"), (1, "dk")):
                    b = pk + 4 * s + (0 if w == 0 else 8)
                    vneed = [("cp" if w == 0 else "cd", u, 4 * s + j) for j in range(4)]
                    mf.append((f"v_mfma_f32_32x32x16_bf16 %[{acc}{dt}], {vr(tr_reg(s, dt, w), 4)}, {vr(b, 4)}, %[{acc}{dt}]",
                               [("tr", u, s, dt, w, t) for t in range(2)], vneed))
    assert len(mf) == 33

    # ------------------------------------------------------------------ LDS read tasks: id -> (text, release gap, deadline MFMA)
    lds = {}
    for u in range(2):
        for g in range(4):
            lds[("st", u, 0, g)] = (f"ds_read_b128 {vr(SC[u] + 4 * g, 4)}, %[ast] offset:{soff + (32 * u + 8 * g) * 4}", 0, 1 + 8 * u)
            lds[("st", u, 1, g)] = (f"ds_read_b128 {vr(DP[u] + 4 * g, 4)}, %[ast] offset:{soff + 256 + (32 * u + 8 * g) * 4}", 0, 2 + 8 * u)
        for s in range(4):
            rel = 0 if u == 0 else 2 * s + 3                 # half 1 reuses the slot: its two MFMAs (2s+1, 2s+2) and one more are out
            lds[("rq", u, s)] = (f"ds_read_b128 {vr(ROW + 8 * s, 4)}, %[aq{s}] offset:{qoff + 4096 * u}", rel, 8 * u + 2 * s + 1)
            lds[("rg", u, s)] = (f"ds_read_b128 {vr(ROW + 8 * s + 4, 4)}, %[aq{s}] offset:{goff + 4096 * u}", rel, 8 * u + 2 * s + 2)
        for s in range(2):
            for dt in range(2):
                for w in range(2):
                    f = (s * 2 + dt) * 2 + w
                    for t in range(2):
                        base = goff if w == 0 else qoff
                        rel = 0 if u == 0 else 17 + f + 1         # the slot's half-0 fragment is read by MFMA 17 + f
                        lds[("tr", u, s, dt, w, t)] = (
                            f"ds_read_b64_tr_b16 {vr(tr_reg(s, dt, w) + 2 * t, 2)}, %[va{dt ^ t}] offset:{base + 4096 * u + (16 * s + 8 * t) * 128}",
                            rel, 17 + 8 * u + f)
    # ------------------------------------------------------------------ VALU tasks: id -> (text, cost, release gap, deadline MFMA, producers)
    va = {}
    for u in range(2):
        pk = PK if u == 0 else ROW
        rel_e, rel_m = 9 + 8 * u, 10 + 8 * u                  # S' of half u is final with MFMA 7 + 8u, dP' with 8 + 8u: two MFMAs on
        for r in range(16):
            s, j = r // 8, (r % 8) // 2
            dl_p, dl_d = 17 + 8 * u + 4 * s, 18 + 8 * u + 4 * s
            va[("e", u, r)] = (f"v_exp_f32_e32 {vr(SC[u] + r)}, {vr(SC[u] + r)}", 2, rel_e, dl_p, [])
            va[("m", u, r)] = (f"v_mul_f32_e32 {vr(DP[u] + r)}, {vr(DP[u] + r)}, {vr(SC[u] + r)}", 1, rel_m, dl_d, [("e", u, r)])
        for sj in range(8):
            s, j = divmod(sj, 4)
            va[("cp", u, sj)] = (f"v_cvt_pk_bf16_f32 {vr(pk + 4 * s + j)}, {vr(SC[u] + 8 * s + 2 * j)}, {vr(SC[u] + 8 * s + 2 * j + 1)}", 1, rel_e,
                                 17 + 8 * u + 4 * s, [("e", u, 8 * s + 2 * j), ("e", u, 8 * s + 2 * j + 1)])
            va[("cd", u, sj)] = (f"v_cvt_pk_bf16_f32 {vr(pk + 8 + 4 * s + j)}, {vr(DP[u] + 8 * s + 2 * j)}, {vr(DP[u] + 8 * s + 2 * j + 1)}", 1, rel_m,
                                 18 + 8 * u + 4 * s, [("m", u, 8 * s + 2 * j), ("m", u, 8 * s + 2 * j + 1)])
    # the packed P / dS of half 1 overwrite row-fragment registers of k-steps 0 and 1, last read by MFMAs 11 / 12: free from gap 13 on (fine)

    # ------------------------------------------------------------------ the prefetch of tile t + 2 into ring slot (slot + 2) % 3
    # (free since the barrier that ended the previous step): 2 KiB of the Q image, 2 KiB of the dO image and one row of statistics per
    # wave.  Issued from inside the stream, one request every few MFMAs: the same five requests issued back to back after the barrier
    # cost every wave ~1500 cycles of issue stall per tile (all eight waves of the CU queue up at the one address unit), 28 % of the kernel.
    dma_at = dict(zip(DMA_GAPS, requests((slot + 2) % NS)))
    if os.environ.get("FK_GEN_ABLATE_DMA"):                      # timing experiments only (wrong results)
        dma_at = {}

    # end of the step: the five requests of tile t + 1 (issued one step ago) have landed, this step's five may stay in flight; every wave
    # is done reading this slot's images -> the barrier frees slot (slot + 2) % 3 ... of the NEXT step and publishes tile t + 1
    tail = ["s_waitcnt vmcnt(5)"] + ([] if os.environ.get("FK_GEN_ABLATE_BARRIER") else ["s_barrier"])
    return schedule(mf, lds, va, dma_at, LDS_PER_GAP, VALU_UNITS, tail)


def main():
    out = sys.argv[1]
    clob = clobbers(100, 244)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/gen_dkdv_asm.py - do not edit.  One fully visible dK/dV tile step per ring slot (see the generator's header).\n")
        for slot in range(NS):
            ins = gen(slot)
            f.write(f"FK_DEV void dkdv_tile_asm_slot{slot}(f32x16& dk0, f32x16& dk1, f32x16& dv0, f32x16& dv1, const bf16x8 (&kf)[4], const bf16x8 (&vf)[4],\n"
                    f"                                   const unsigned (&aq)[4], unsigned va0, unsigned va1, unsigned ast,\n"
                    f"                                   const unsigned (&vo)[5], uint64_t qb, uint64_t gb, uint64_t sb, unsigned ldsw, unsigned ldss) {{\n")
            f.write("  asm volatile(\n")
            for i in ins:
                f.write(f'      "{i}\\n\\t"\n')
            f.write('      : [dk0] "+v"(dk0), [dk1] "+v"(dk1), [dv0] "+v"(dv0), [dv1] "+v"(dv1)\n')
            f.write('      : [kf0] "v"(kf[0]), [kf1] "v"(kf[1]), [kf2] "v"(kf[2]), [kf3] "v"(kf[3]), [vf0] "v"(vf[0]), [vf1] "v"(vf[1]), [vf2] "v"(vf[2]), [vf3] "v"(vf[3]),\n')
            f.write('        [aq0] "v"(aq[0]), [aq1] "v"(aq[1]), [aq2] "v"(aq[2]), [aq3] "v"(aq[3]), [va0] "v"(va0), [va1] "v"(va1), [ast] "v"(ast),\n')
            f.write('        [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [vo4] "v"(vo[4]), [qb] "s"(qb), [gb] "s"(gb), [sb] "s"(sb), [ldsw] "s"(ldsw), [ldss] "s"(ldss)\n')
            f.write(f"      : {clob}, \"scc\", \"memory\");\n}}\n")
        # the same requests on their own (prologue: tiles 0 and 1), also as asm so that hipcc's wait-count pass never sees an LDS-DMA in
        # this kernel (it would put a full vmcnt(0) in front of every block that may read LDS)
        for slot in range(2):
            f.write(f"FK_DEV void dkdv_request_asm_slot{slot}(const unsigned (&vo)[5], uint64_t qb, uint64_t gb, uint64_t sb, unsigned ldsw, unsigned ldss) {{\n  asm volatile(\n")
            for m0, ld in requests(slot):
                f.write(f'      "{m0}\\n\\t"\n      "s_nop 0\\n\\t"\n      "{ld}\\n\\t"\n')
            f.write('      :\n      : [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [vo4] "v"(vo[4]), [qb] "s"(qb), [gb] "s"(gb), [sb] "s"(sb), [ldsw] "s"(ldsw), [ldss] "s"(ldss)\n')
            f.write('      : "scc", "memory");\n}\n')
        f.write(f"// instructions per tile step: {len(ins)}\n")
    print(f"{out}: {len(ins)} instructions per tile step")


if __name__ == "__main__":
    main()
