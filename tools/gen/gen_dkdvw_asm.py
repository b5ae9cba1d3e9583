#!/usr/bin/env python3
"""Generates frankenstein_amd/csrc/attn_dkdvw_asm.inc: the tile step of the WIDE dK/dV attention-backward kernel (bf16, D = 64) —
64 keys per wave (two 32-key blocks kb = 0, 1), ONE wave per SIMD (512 registers: the dK / dV accumulators and the K / V fragments live in
the accumulator half, "a" operands), 64-query tile = two 32-row halves u = 0, 1.

Why: in the 32-keys-per-wave stream (gen_dkdv_asm.py) every LDS fragment feeds one MFMA — 2 ds_read per MFMA, the most expensive
instruction class to leave in (-26...-33 % without them, LDS command FIFO full 28 M times per launch).  None of the reads of
this kernel depends on the key: Q / dO row fragments, their transposed fragments and the row statistics serve every key of the wave.  With
64 keys per wave each fragment feeds TWO MFMAs: 64 MFMAs per step for 48 fragment reads + 32 statistics reads (1.25 per MFMA), half the
LDS-DMA pieces per MFMA, and one Q / dO tile fetch serves 256 keys of a workgroup instead of 128.

  MFMA  1..16   S' / dP' of half 0: per k-step s  [S' kb0, S' kb1, dP' kb0, dP' kb1]   | gaps: statistics + row reads of half 1, transposed reads of half 0
  MFMA 17..32   S' / dP' of half 1                                                     | gaps: exp2 / dS / packing of half 0
  MFMA 33..48   dV / dK of half 0: per (s, dt)  [dV kb0, dV kb1, dK kb0, dK kb1]       | gaps: exp2 / dS / packing of half 1, transposed reads of half 1
  MFMA 49..64   dV / dK of half 1                                                      | gaps: the rest

Software pipelined ACROSS steps (a wave alone on its SIMD has no partner to cover the start of a step): half 0's row statistics
(the initial accumulators SC / DP(0, kb)) and its eight row fragments (ROW) are read from the NEXT tile at the end of this step and are
carried into the next block in pinned registers (operands "+{v[..]}"), so a step begins with its MFMAs.  That moves the ring protocol's
wait + barrier into the step: `s_waitcnt vmcnt(0); s_barrier` right behind MFMA 33 — tile t + 1 (requested one step ago, nothing younger
is outstanding) has landed and every wave has left step t - 1, so the prefetch may read tile t + 1 and the five requests of tile t + 2
(into the slot of tile t - 1) follow in the gaps behind it.  The prologue runs dkdvw_prefetch_asm (the same reads for tile 0).

P and dS are packed IN PLACE: bf16 pairs of SC[8s + 2j], SC[8s + 2j + 1] go to SC[8s + j] (j ascending, each after its readers), so the
B operand of the dV / dK MFMAs of k-step s is v[SC + 8s : SC + 8s + 3] and no separate registers are needed.

Register map (hard-coded temporaries, clobbers):  SC(u, kb) v[40 + 64u + 32kb ..+15]   DP(u, kb) = SC(u, kb) + 16
  ROW v[168:199] four k-steps x (Q fragment 4, dO fragment 4)      TR v[200:231] eight transposed fragments x 4
Operands: dk{kb}{dt}, dv{kb}{dt} (f32x16, "+a"), kf{kb}{s}, vf{kb}{s} (bf16x8, "a"), aq0..3 / va0 va1 / ast (LDS byte addresses, slot 0),
vo0..4 + qb gb sb + ldsw ldss (the LDS-DMA requests of tile t + 2, as in gen_dkdv_asm.py).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fkstream import vr, schedule, clobbers  # noqa: E402

IMG, NS = 64 * 128, 3
ROW, TR = 168, 200
LDS_PER_GAP = int(os.environ.get("FK_GEN_LDS_PER_GAP", "2"))
VALU_UNITS = int(os.environ.get("FK_GEN_VALU_UNITS", "6"))
DMA_GAPS = [int(x) for x in os.environ.get("FK_GEN_DMAW_GAPS", "35,41,47,53,59").split(",")]
BARRIER_AT = 33                     # the wait + barrier sits right behind this MFMA (the first of phase B)


def SC(u, kb):
    return 40 + 64 * u + 32 * kb


def DP(u, kb):
    return SC(u, kb) + 16


def tr_reg(s, dt, w):
    return TR + 4 * ((s * 2 + dt) * 2 + w)


def requests(ps):
    return [(f"s_add_u32 m0, %[ldsw], {ps * IMG}", "global_load_lds_dwordx4 %[vo0], %[qb]"),
            (f"s_add_u32 m0, %[ldsw], {ps * IMG + 1024}", "global_load_lds_dwordx4 %[vo1], %[qb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG}", "global_load_lds_dwordx4 %[vo2], %[gb]"),
            (f"s_add_u32 m0, %[ldsw], {(NS + ps) * IMG + 1024}", "global_load_lds_dwordx4 %[vo3], %[gb]"),
            (f"s_add_u32 m0, %[ldss], {ps * 2 * 64 * 4}", "global_load_lds_dword %[vo4], %[sb]")]


def a_idx(u, s, k):          # phase A: k = 0 S' kb0, 1 S' kb1, 2 dP' kb0, 3 dP' kb1
    return 16 * u + 4 * s + k + 1


def b_idx(u, s, dt, w, kb):  # phase B
    return 32 + 16 * u + 8 * s + 4 * dt + 2 * w + kb + 1


def gen(slot):
    qoff, soff = slot * IMG, slot * 2 * 64 * 4
    goff = qoff + NS * IMG
    mf = [None] * 65
    for u in range(2):
        for s in range(4):
            for kb in range(2):
                # half 0's statistics and row fragments arrive with the block (prefetched by the previous step / the prologue)
                st_sc = [("st", u, 0, kb, g) for g in range(4)] if (s == 0 and u == 1) else []
                st_dp = [("st", u, 1, kb, g) for g in range(4)] if (s == 0 and u == 1) else []
                mf[a_idx(u, s, kb)] = (f"v_mfma_f32_32x32x16_bf16 {vr(SC(u, kb), 16)}, {vr(ROW + 8 * s, 4)}, %[kf{kb}{s}], {vr(SC(u, kb), 16)}",
                                       ([("rq", u, s)] if u == 1 else []) + st_sc, [])
                mf[a_idx(u, s, 2 + kb)] = (f"v_mfma_f32_32x32x16_bf16 {vr(DP(u, kb), 16)}, {vr(ROW + 8 * s + 4, 4)}, %[vf{kb}{s}], {vr(DP(u, kb), 16)}",
                                           ([("rg", u, s)] if u == 1 else []) + st_dp, [])
    for u in range(2):
        for s in range(2):
            for dt in range(2):
                for w, acc in ((0, "dv"), (1, "dk")):
                    for kb in range(2):
                        b = (SC(u, kb) if w == 0 else DP(u, kb)) + 8 * s
                        vneed = [("cp" if w == 0 else "cd", u, kb, s, j) for j in range(4)]
                        mf[b_idx(u, s, dt, w, kb)] = (f"v_mfma_f32_32x32x16_bf16 %[{acc}{kb}{dt}], {vr(tr_reg(s, dt, w), 4)}, {vr(b, 4)}, %[{acc}{kb}{dt}]",
                                                      [("tr", u, s, dt, w, t) for t in range(2)], vneed)
    assert all(m is not None for m in mf[1:])

    nslot = (slot + 1) % NS                                    # the tile the prefetch reads: t + 1
    nq, nso = nslot * IMG, nslot * 2 * 64 * 4
    ng = nq + NS * IMG
    LAST = 65                                                  # "deadline" of a read without a consumer in this step
    lds = {}
    for kb in range(2):
        for g in range(4):
            # half 1: its accumulators held the packed operands of the previous step's last MFMAs: two MFMAs of this step first
            lds[("st", 1, 0, kb, g)] = (f"ds_read_b128 {vr(SC(1, kb) + 4 * g, 4)}, %[ast] offset:{soff + (32 + 8 * g) * 4}", 2, a_idx(1, 0, kb))
            lds[("st", 1, 1, kb, g)] = (f"ds_read_b128 {vr(DP(1, kb) + 4 * g, 4)}, %[ast] offset:{soff + 256 + (32 + 8 * g) * 4}", 2, a_idx(1, 0, 2 + kb))
            # half 0 of the NEXT tile: SC(0, kb) / DP(0, kb) were last read (packed P / dS) by MFMAs b_idx(0, 1, 1, w, kb)
            lds[("pst", 0, kb, g)] = (f"ds_read_b128 {vr(SC(0, kb) + 4 * g, 4)}, %[ast] offset:{nso + (8 * g) * 4}", b_idx(0, 1, 1, 0, 1) + 2, LAST)
            lds[("pst", 1, kb, g)] = (f"ds_read_b128 {vr(DP(0, kb) + 4 * g, 4)}, %[ast] offset:{nso + 256 + (8 * g) * 4}", b_idx(0, 1, 1, 1, 1) + 2, LAST)
    for s in range(4):
        rel_q = a_idx(0, s, 1) + 1                             # half 0's S' MFMAs of this k-step and one more are out
        rel_g = a_idx(0, s, 3) + 1
        lds[("rq", 1, s)] = (f"ds_read_b128 {vr(ROW + 8 * s, 4)}, %[aq{s}] offset:{qoff + 4096}", rel_q, a_idx(1, s, 0))
        lds[("rg", 1, s)] = (f"ds_read_b128 {vr(ROW + 8 * s + 4, 4)}, %[aq{s}] offset:{goff + 4096}", rel_g, a_idx(1, s, 2))
        # next tile's half-0 rows: ROW was last read by phase A of half 1 (MFMAs <= 32); behind the barrier (tile t + 1 has landed)
        lds[("prq", s)] = (f"ds_read_b128 {vr(ROW + 8 * s, 4)}, %[aq{s}] offset:{nq}", BARRIER_AT + 1, LAST)
        lds[("prg", s)] = (f"ds_read_b128 {vr(ROW + 8 * s + 4, 4)}, %[aq{s}] offset:{ng}", BARRIER_AT + 1, LAST)
    for u in range(2):
        for s in range(2):
            for dt in range(2):
                for w in range(2):
                    for t in range(2):
                        base = goff if w == 0 else qoff
                        if u == 0:                                   # the previous step's half-1 MFMAs 49 + f, 50 + f read this fragment
                            rel = max(0, b_idx(1, s, dt, w, 1) + 1 - 64)
                        else:
                            rel = b_idx(0, s, dt, w, 1) + 1
                        lds[("tr", u, s, dt, w, t)] = (
                            f"ds_read_b64_tr_b16 {vr(tr_reg(s, dt, w) + 2 * t, 2)}, %[va{dt ^ t}] offset:{base + 4096 * u + (16 * s + 8 * t) * 128}",
                            rel, b_idx(u, s, dt, w, 0))
    va = {}
    for u in range(2):
        for kb in range(2):
            sc, dp = SC(u, kb), DP(u, kb)
            rel_e = a_idx(u, 3, kb) + 2                            # S' final with its k-step-3 MFMA: two MFMAs on
            rel_m = a_idx(u, 3, 2 + kb) + 2
            for r in range(16):
                s = r // 8
                va[("e", u, kb, r)] = (f"v_exp_f32_e32 {vr(sc + r)}, {vr(sc + r)}", 2, rel_e, b_idx(u, s, 0, 0, kb), [])
                va[("m", u, kb, r)] = (f"v_mul_f32_e32 {vr(dp + r)}, {vr(dp + r)}, {vr(sc + r)}", 1, rel_m, b_idx(u, s, 0, 1, kb), [("e", u, kb, r)])
            for s in range(2):
                for j in range(4):
                    # in place: the destination sc + 8s + j is read by exp / mul of that element and by the pack that consumes it
                    dep_p = [("e", u, kb, 8 * s + 2 * j), ("e", u, kb, 8 * s + 2 * j + 1), ("m", u, kb, 8 * s + j)]
                    dep_d = [("m", u, kb, 8 * s + 2 * j), ("m", u, kb, 8 * s + 2 * j + 1)]
                    if j >= 1:
                        dep_p.append(("cp", u, kb, s, j // 2))
                        dep_d.append(("cd", u, kb, s, j // 2))
                    va[("cp", u, kb, s, j)] = (f"v_cvt_pk_bf16_f32 {vr(sc + 8 * s + j)}, {vr(sc + 8 * s + 2 * j)}, {vr(sc + 8 * s + 2 * j + 1)}", 1, rel_e,
                                               b_idx(u, s, 0, 0, kb), dep_p)
                    va[("cd", u, kb, s, j)] = (f"v_cvt_pk_bf16_f32 {vr(dp + 8 * s + j)}, {vr(dp + 8 * s + 2 * j)}, {vr(dp + 8 * s + 2 * j + 1)}", 1, rel_m,
                                               b_idx(u, s, 0, 1, kb), dep_d)
    dma_at = dict(zip(DMA_GAPS, requests((slot + 2) % NS)))
    if os.environ.get("FK_GEN_ABLATE_DMA"):
        dma_at = {}
    mid = {BARRIER_AT: ["s_waitcnt vmcnt(0)"] + ([] if os.environ.get("FK_GEN_ABLATE_BARRIER") else ["s_barrier"])}
    return schedule(mf, lds, va, dma_at, LDS_PER_GAP, VALU_UNITS, [], mid=mid)


def gen_prefetch():
    """half 0's statistics and row fragments of tile 0 (slot 0): what a tile step expects to find in the pinned registers"""
    ins = []
    for kb in range(2):
        for g in range(4):
            ins.append(f"ds_read_b128 {vr(SC(0, kb) + 4 * g, 4)}, %[ast] offset:{(8 * g) * 4}")
            ins.append(f"ds_read_b128 {vr(DP(0, kb) + 4 * g, 4)}, %[ast] offset:{256 + (8 * g) * 4}")
    for s in range(4):
        ins.append(f"ds_read_b128 {vr(ROW + 8 * s, 4)}, %[aq{s}] offset:0")
        ins.append(f"ds_read_b128 {vr(ROW + 8 * s + 4, 4)}, %[aq{s}] offset:{NS * IMG}")
    ins.append("s_waitcnt lgkmcnt(0)")
    return ins


ACC = [f"{a}{kb}{dt}" for a in ("dk", "dv") for kb in range(2) for dt in range(2)]
FRG = [f"{a}{kb}{s}" for a in ("kf", "vf") for kb in range(2) for s in range(4)]


# carried from step to step in pinned registers: SC(0, 0), DP(0, 0), SC(0, 1), DP(0, 1) and the two halves of ROW
CARRY = [("c0", SC(0, 0)), ("c1", DP(0, 0)), ("c2", SC(0, 1)), ("c3", DP(0, 1)), ("c4", ROW), ("c5", ROW + 16)]
CARRIED = {r for _, b in CARRY for r in range(b, b + 16)}


def main():
    out = sys.argv[1]
    clob = ", ".join(f'"v{r}"' for r in range(40, 232) if r not in CARRIED)
    pins = ", ".join(f'"+{{v[{b}:{b + 15}]}}"({n})' for n, b in CARRY)
    carry_args = ", ".join(f"f32x16& {n}" for n, _ in CARRY)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/gen_dkdvw_asm.py - do not edit.  One fully visible tile step of the wide (64 keys per wave, one wave per SIMD)\n"
                "// dK/dV kernel per ring slot, software pipelined across steps (see the generator's header).\n")
        for slot in range(NS):
            ins = gen(slot)
            f.write(f"FK_DEV void dkdvw_tile_asm_slot{slot}(f32x16 (&dk)[2][2], f32x16 (&dv)[2][2], const bf16x8 (&kf)[2][4], const bf16x8 (&vf)[2][4],\n"
                    f"                                    {carry_args},\n"
                    f"                                    const unsigned (&aq)[4], unsigned va0, unsigned va1, unsigned ast,\n"
                    f"                                    const unsigned (&vo)[5], uint64_t qb, uint64_t gb, uint64_t sb, unsigned ldsw, unsigned ldss) {{\n")
            f.write("  asm volatile(\n")
            for i in ins:
                f.write(f'      "{i}\\n\\t"\n')
            f.write("      : " + ", ".join(f'[{n}] "+a"({n[:2]}[{n[2]}][{n[3]}])' for n in ACC) + ",\n        " + pins + "\n")
            f.write("      : " + ", ".join(f'[{n}] "a"({n[:2]}[{n[2]}][{n[3]}])' for n in FRG) + ",\n")
            f.write('        [aq0] "v"(aq[0]), [aq1] "v"(aq[1]), [aq2] "v"(aq[2]), [aq3] "v"(aq[3]), [va0] "v"(va0), [va1] "v"(va1), [ast] "v"(ast),\n')
            f.write('        [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [vo4] "v"(vo[4]), [qb] "s"(qb), [gb] "s"(gb), [sb] "s"(sb), [ldsw] "s"(ldsw), [ldss] "s"(ldss)\n')
            f.write(f"      : {clob}, \"scc\", \"memory\");\n}}\n")
        f.write(f"FK_DEV void dkdvw_prefetch_asm({carry_args}, const unsigned (&aq)[4], unsigned ast) {{\n  asm volatile(\n")
        for i in gen_prefetch():
            f.write(f'      "{i}\\n\\t"\n')
        f.write("      : " + ", ".join(f'"=&{{v[{b}:{b + 15}]}}"({n})' for n, b in CARRY) + "\n")
        f.write('      : [aq0] "v"(aq[0]), [aq1] "v"(aq[1]), [aq2] "v"(aq[2]), [aq3] "v"(aq[3]), [ast] "v"(ast)\n      : "memory");\n}\n')
        f.write(f"// instructions per tile step: {len(ins)}\n")
    print(f"{out}: {len(ins)} instructions per tile step")


if __name__ == "__main__":
    main()
