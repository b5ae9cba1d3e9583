#!/usr/bin/env python3
"""Generates frankenstein_amd/csrc/mlp_bwd_asm.inc: the hand-placed instruction stream of one chunk step of the fused MLP backward
(mlp_bwd_fused_asm_kernel, csrc/mlp_fused.hip) as ONE inline-asm block:

    dx^T += W13T_c dh13^T(c)                     48 MFMAs (12 feature tiles x 4 k16-steps), A fragments from the W13T image in LDS
  BESIDE the SwiGLU derivative of chunk c + 1    16 hidden units per lane (13 VALU + 2 packing instructions per unit), its dh13 tile round
                                                 trip (4 ds_write, 4 row reads, 4 row stores), 6 + 4 LDS-DMA requests, the barrier

Why generated: a wave of this kernel is alone on its SIMD (96 + 192 stationary registers), so everything it issues shares ONE in-order
stream.  The stamps of the hipcc forms (tools/stamp_mlp.py, profiles/r04_mf_stamps*.txt) fit one model: a gap behind an MFMA costs
max(32, 8 + the issue time of what is placed in it) cycles.  hipcc places the SwiGLU arithmetic as a phase of its own (first form: 1856
cycles per chunk with the matrix pipe idle) or, asked to interleave, as 8-9 VALU incl. two transcendentals per gap in 32 of the 48 gaps
(64 cycles per gap; the other 16 gaps stay at 32): no gain either way.  Here every gap gets an equal share by budget (BUDGET 4-cycle
units: LDS read 1, VALU 1, transcendental 2, ds_write 3, store 2, LDS-DMA request 3), earliest work first.

Same arithmetic as the hipcc kernels it replaces, instruction for instruction (mf_sigmoid<true> / the (T) casts of mlp_bwd_fused_kernel):
    a1 = bf16->f32 (shift / mask), t = exp2(a1 * -log2 e), sg = rcp(1 + t), ds = g * sg, d1 = (ds * a3) * fma(1 - sg, a1, 1), d3 = ds * a1,
    packed with v_cvt_pk_bf16_f32 — the bits of dh13 and dx do not change (tests/test_kernels_gpu.py).

Register map (physical; the kernel pins its operands to them, the rest are clobbers):
  a[0:191]   dx^T accumulators (12 tiles)        a[192:207] dg^T of chunk c + 1 (input)
  v[132:163] FA   two groups of four A fragments; v[132:147] is also an operand: in = feature tile 0's fragments (read by the caller under
                  the first product's last MFMAs), out = the NEXT first product's first fragments (W2T image), requested behind the barrier
  v[164:179] BFC  dh13^T(c): the B fragments of the 48 MFMAs (input)
  v[180:195] BFN  dh13^T(c + 1) (output)
  v[196:211] HV   the lane's h13 pieces of chunk c + 1 (input); reused for the row read-back RB
  v[212:223] TMP  three units' temporaries
  v[224:239] ADR  LDS byte addresses: AF[4] W13T fragments (tile 0), AH[4] the lane's tile pieces, AR[4] tile rows, A2[4] next W2T fragments
  v[240:255] OFS  global byte offsets: OFF2[6] W2T request pieces, HOFF[4] h13 tile pieces, VST[4] dh13 row stores (two spare)
  scalars    %[g2] W2T(c + 3) base, %[gh] h13 chunk c + 2 base, %[gst] dh13 chunk c + 1 base, %[ldsw2] / %[ldsh] LDS destinations of the requests
Hazards kept by construction and re-checked by tools/gen/verify_stream.py: counted lgkmcnt waits (LDS reads AND writes return in order), a VALU
result consumed at least two instructions later, dg (an MFMA result of the caller) read only after two MFMAs of the stream, a fragment
register rewritten only after its MFMA and one more, M0 written one instruction ahead of its request; the block opens with s_nop 4
(operands fresh from the caller's VALU / readfirstlane).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fkstream import vr, clobbers  # noqa: E402

FA, BFC, BFN, HV, TMP, ADR, OFS = 132, 164, 180, 196, 212, 224, 240
AF, AH, AR, A2 = ADR, ADR + 4, ADR + 8, ADR + 12
OFF2, HOFF, VST = OFS, OFS + 6, OFS + 10
DG = 192
NT, NK = 12, 4
BUDGET = int(os.environ.get("FK_GEN_MLPB_BUDGET", "7"))
LDS_PER_GAP = int(os.environ.get("FK_GEN_MLPB_LDS", "2"))
W2_GAPS = [int(x) for x in os.environ.get("FK_GEN_MLPB_W2_GAPS", "3,9,15,21,27,33").split(",")]
WAIT_PER_TILE = int(os.environ.get("FK_GEN_MLPB_WAIT_PER_TILE", "1"))
BARRIER_AT = 45                       # in front of MFMA 45 (feature tile 11): every W13T fragment of this wave is in registers


def ar(a, n=1):
    return f"a{a}" if n == 1 else f"a[{a}:{a + n - 1}]"


class Stream:
    def __init__(self):
        self.out = []
        self.queue = []        # outstanding LDS operations (reads and writes) in issue order: ids
        self.done = set()
        self.pos = {}          # task id -> index in out

    def emit(self, text, tid=None):
        if tid is not None:
            self.pos[tid] = len(self.out)
        self.out.append(text)

    def lds(self, tid, text):
        self.queue.append(tid)
        self.emit(text, tid)

    def need(self, ids):
        ids = [i for i in ids if i not in self.done]
        if not ids:
            return
        last = max(self.queue.index(i) for i in ids)
        self.out.append(f"s_waitcnt lgkmcnt({min(len(self.queue) - 1 - last, 15)})")
        self.done.update(self.queue[:last + 1])
        self.queue = self.queue[last + 1:]


def unit_tasks(k):
    """the 13 VALU instructions of hidden unit k = 4 s + e (slice s = k16-step of the second product, e = 0..3) -> [(id, text, cost, deps)]"""
    s, e = divmod(k, 4)
    t0, t1, t2, t3 = (TMP + 4 * (k % 3) + i for i in range(4))
    w1, w3 = HV + 4 * s + e // 2, HV + 4 * s + 2 + e // 2          # words holding a1 = h1 and a3 = h3 of this unit
    cvt = (lambda d, w: f"v_lshlrev_b32_e32 {vr(d)}, 16, {vr(w)}") if e % 2 == 0 else (lambda d, w: f"v_and_b32_e32 {vr(d)}, 0xffff0000, {vr(w)}")
    u = lambda i: ("u", k, i)
    free = [("pk", (k - 3) // 2, j) for j in range(2)] if k >= 3 else []      # the temporaries' previous owner has been packed
    return [
        (u(0), cvt(t0, w1), 1, free),
        (u(1), f"v_mul_f32_e32 {vr(t2)}, 0xbfb8aa3b, {vr(t0)}", 1, [u(0)] + free),
        (u(2), f"v_exp_f32_e32 {vr(t2)}, {vr(t2)}", 2, [u(1)]),
        (u(3), cvt(t1, w3), 1, free),
        (u(4), f"v_accvgpr_read_b32 {vr(t3)}, {ar(DG + k)}", 1, free),
        (u(5), f"v_add_f32_e32 {vr(t2)}, 1.0, {vr(t2)}", 1, [u(2)]),
        (u(6), f"v_rcp_f32_e32 {vr(t2)}, {vr(t2)}", 2, [u(5)]),
        (u(7), f"v_mul_f32_e32 {vr(t3)}, {vr(t3)}, {vr(t2)}", 1, [u(4), u(6)]),
        (u(8), f"v_sub_f32_e32 {vr(t2)}, 1.0, {vr(t2)}", 1, [u(7)]),
        (u(9), f"v_mul_f32_e32 {vr(t1)}, {vr(t3)}, {vr(t1)}", 1, [u(3), u(7)]),
        (u(10), f"v_fma_f32 {vr(t2)}, {vr(t2)}, {vr(t0)}, 1.0", 1, [u(8), u(0)]),
        (u(11), f"v_mul_f32_e32 {vr(t1)}, {vr(t1)}, {vr(t2)}", 1, [u(9), u(10)]),
        (u(12), f"v_mul_f32_e32 {vr(t0)}, {vr(t3)}, {vr(t0)}", 1, [u(7), u(10)]),
    ]


def gen():
    S = Stream()
    # ---- VALU work in program order: units 0..15, each pair (2p, 2p+1) followed by its two packing instructions; a finished slice s is
    #      written back into the tile (ds_write), all four slices -> row reads -> row stores -> the next tile's requests
    valu = []
    for k in range(16):
        valu += unit_tasks(k)
        if k % 2 == 1:
            p, s, h = k // 2, k // 4, (k // 2) % 2
            ta, tb = TMP + 4 * ((k - 1) % 3), TMP + 4 * (k % 3)
            valu.append((("pk", p, 0), f"v_cvt_pk_bf16_f32 {vr(BFN + 4 * s + h)}, {vr(ta + 1)}, {vr(tb + 1)}", 1, [("u", k - 1, 11), ("u", k, 11)]))
            valu.append((("pk", p, 1), f"v_cvt_pk_bf16_f32 {vr(BFN + 4 * s + 2 + h)}, {vr(ta)}, {vr(tb)}", 1, [("u", k - 1, 12), ("u", k, 12)]))
    chain = []          # (id, kind, text, cost, deps): the tile round trip, in order
    for s in range(4):
        chain.append((("w", s), "lds", f"ds_write_b128 {vr(AH + s)}, {vr(BFN + 4 * s, 4)}", 3, [("pk", 2 * s, 0), ("pk", 2 * s, 1), ("pk", 2 * s + 1, 0), ("pk", 2 * s + 1, 1)]))
    for j in range(4):
        chain.append((("r", j), "lds", f"ds_read_b128 {vr(HV + 4 * j, 4)}, {vr(AR + j)}", 1, [("w", 3)]))
    for j in range(4):
        chain.append((("st", j), "st", f"global_store_dwordx4 {vr(VST + j)}, {vr(HV + 4 * j, 4)}, %[gst] nt", 2, [("r", j)]))
    for j in range(4):
        chain.append((("dh", j), "dma", (f"s_add_u32 m0, %[ldsh], {j * 1024}", f"global_load_lds_dwordx4 {vr(HOFF + j)}, %[gh]"), 3, [("st", 3)]))
    w2 = {g: (f"s_add_u32 m0, %[ldsw2], {j * 1024}", f"global_load_lds_dwordx4 {vr(OFF2 + j)}, %[g2]") for j, g in enumerate(W2_GAPS)}

    # ---- A fragments: tile t, k16-step s in FA[t & 1][s]; tile 0's are the caller's
    def frag(t, s):
        return FA + 16 * (t & 1) + 4 * s
    frag_reads = {}     # (t, s) -> (text, release gap)
    for t in range(1, NT):
        for s in range(NK):
            rel = 0 if t == 1 else 4 * (t - 2) + s + 2                  # the register's previous MFMA (tile t - 2, step s) and one more are out
            frag_reads[(t, s)] = (f"ds_read_b128 {vr(frag(t, s), 4)}, {vr(AF + s)} offset:{t * 4096}", rel)
    next_reads = [(("n", s), f"ds_read_b128 {vr(FA + 4 * s, 4)}, {vr(A2 + s)}") for s in range(NK)]     # behind the barrier; FA[0] is free after MFMA 44 (tile 10)

    def ready(deps, consumer_is_valu=True):
        n = len(S.out)
        return all(d in S.pos and n - S.pos[d] >= 2 for d in deps)

    S.emit("s_nop 4")
    ci = 0
    for g in range(1, NT * NK + 1):
        t, s = divmod(g - 1, NK)
        if g == BARRIER_AT:
            for key in sorted(frag_reads):                              # (normally all issued by now)
                S.lds(("f",) + key, frag_reads.pop(key)[0])
            S.need([q for q in S.queue if q[0] == "f"])
            S.emit("s_barrier")
        if t >= 1:
            # ONE wait per feature tile, in front of its first MFMA, for the tile's last fragment (LDS operations return in order, and the
            # fragments are requested seven gaps ahead): a wait per MFMA costs the lone wave an issue slot each
            for ss in range(NK if WAIT_PER_TILE and s == 0 else 1):
                key = (t, ss if WAIT_PER_TILE and s == 0 else s)
                if key in frag_reads:                                   # not scheduled in time: now
                    S.lds(("f",) + key, frag_reads.pop(key)[0])
            S.need([("f", t, NK - 1)] if WAIT_PER_TILE and s == 0 else [("f", t, s)])
        S.emit(f"v_mfma_f32_32x32x16_bf16 {ar(16 * t, 16)}, {vr(frag(t, s), 4)}, {vr(BFC + 4 * s, 4)}, {ar(16 * t, 16)}", ("m", g))
        units = BUDGET
        # 1. the fragment reads that come due first
        n = 0
        while n < LDS_PER_GAP and units > 0:
            cands = sorted(k for k, v in frag_reads.items() if v[1] <= g)
            if not cands:
                break
            k = cands[0]
            S.lds(("f",) + k, frag_reads.pop(k)[0])
            n += 1
            units -= 1
        if g >= BARRIER_AT and next_reads and n < LDS_PER_GAP:
            while next_reads and n < LDS_PER_GAP:
                tid, text = next_reads.pop(0)
                S.lds(tid, text)
                n += 1
                units -= 1
        # 2. a W2T request: M0 first, the request at the end of the gap
        m0 = w2.get(g)
        if m0:
            S.emit(m0[0])
            units -= 3
        n0 = len(S.out)
        # 3. the tile round trip as soon as its inputs exist (it ends the step: nothing may wait for it at the end)
        while ci < len(chain) and units > 0:
            tid, kind, text, cost, deps = chain[ci]
            if not ready(deps) or (m0 and kind == "dma"):
                break
            if kind == "lds":
                S.lds(tid, text)
            elif kind == "st":
                S.need(deps)
                S.emit(text, tid)
            else:
                S.emit(text[0])
                S.emit("s_nop 0")
                S.emit(text[1], tid)
            units -= cost
            ci += 1
        # 4. the SwiGLU arithmetic: the earliest instruction (program order) whose inputs are two instructions old — up to three units are
        #    in flight (three sets of temporaries), so a unit's dependent chain is interleaved with its neighbours'.  dg is the caller's
        #    MFMA result: nothing before two MFMAs of this stream are out
        while units > 0 and g >= 2:
            pick = next((i for i, v in enumerate(valu) if v[2] <= units and ready(v[3])), None)
            if pick is None:
                break
            tid, text, cost, deps = valu.pop(pick)
            S.emit(text, tid)
            units -= cost
        if m0:
            if len(S.out) == n0:
                S.emit("s_nop 0")
            S.emit(m0[1])
    # whatever the gaps did not take
    while valu:
        pick = next((i for i, v in enumerate(valu) if ready(v[3])), None)
        if pick is None:
            S.emit("s_nop 0")
            continue
        tid, text, cost, deps = valu.pop(pick)
        S.emit(text, tid)
    while ci < len(chain):
        tid, kind, text, cost, deps = chain[ci]
        if not ready(deps):
            S.emit("s_nop 0")
            continue
        if kind == "lds":
            S.lds(tid, text)
        elif kind == "st":
            S.need(deps)
            S.emit(text, tid)
        else:
            S.emit(text[0])
            S.emit("s_nop 0")
            S.emit(text[1], tid)
        ci += 1
    for tid, text in next_reads:
        S.lds(tid, text)
    assert not frag_reads
    if S.queue:
        S.need(list(S.queue))                                          # the next first product's fragments are the block's outputs
    S.emit("s_nop 1")                                                  # the last row store / request has read its registers
    return S.out


def main():
    out = sys.argv[1]
    ins = gen()
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/gen_mlpb_asm.py - do not edit.  One chunk step of mlp_bwd_fused_asm_kernel (see the generator's header).\n")
        f.write("FK_DEV void mlpb_step_asm(f32x16 (&dx)[12], const f32x16& dg, u32x16& fa0, const u32x16& bfc, u32x16& bfn, u32x16& hv, const u32x16& adr,\n"
                "                          const u32x16& ofs, const void* g2, const void* gh, void* gst, unsigned ldsw2, unsigned ldsh) {\n")
        f.write("  asm volatile(\n")
        for i in ins:
            f.write(f'      "{i}\\n\\t"\n')
        ops = ", ".join(f'"+{{a[{16 * t}:{16 * t + 15}]}}"(dx[{t}])' for t in range(NT))
        f.write(f"      : {ops},\n")
        f.write(f'        "+{{v[{FA}:{FA + 15}]}}"(fa0), "={{v[{BFN}:{BFN + 15}]}}"(bfn), "+{{v[{HV}:{HV + 15}]}}"(hv)\n')
        f.write(f'      : "{{a[{DG}:{DG + 15}]}}"(dg), "{{v[{BFC}:{BFC + 15}]}}"(bfc), "{{v[{ADR}:{ADR + 15}]}}"(adr), "{{v[{OFS}:{OFS + 15}]}}"(ofs),\n')
        f.write('        [g2] "s"(g2), [gh] "s"(gh), [gst] "s"(gst), [ldsw2] "s"(ldsw2), [ldsh] "s"(ldsh)\n')
        f.write(f"      : {clobbers(FA + 16, FA + 32)}, {clobbers(TMP, TMP + 12)}, \"scc\", \"memory\");\n}}\n")
        n_mfma = sum(i.startswith("v_mfma") for i in ins)
        f.write(f"// instructions per chunk step: {len(ins)} ({n_mfma} MFMAs)\n")
    print(f"{out}: {len(ins)} instructions per chunk step")


if __name__ == "__main__":
    main()
