"""Shared by the stream generators (gen_dkdv_asm.py, gen_dq_asm.py): a model of one wave's instruction stream around a fixed MFMA sequence.

A tile step is `mf[1..n]`, the MFMAs in issue order, each with the LDS reads and VALU results it consumes.  Everything else is a task with
a release gap (the first gap behind an MFMA in which it may be issued: data-flow and register-reuse constraints) and a deadline (the MFMA
that consumes it); every gap takes the most urgent released tasks up to a budget (earliest deadline first).  LDS reads return in order, so
the wait in front of an MFMA is a counted `s_waitcnt lgkmcnt(n)` computed from the reads issued since.
Hazards kept by construction: a VALU result is consumed (by VALU or as an MFMA operand) at least 2 instructions later; release gaps keep
an MFMA result away from VALU until two further MFMAs are out, and an LDS read away from a fragment register until the MFMA that read it
and one more have been issued.
"""


import os


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


class Model:
    def __init__(self):
        self.out, self.reads, self.done = [], [], set()

    def emit(self, s):
        self.out.append(s)

    def lds(self, rid, text):
        self.reads.append(rid)
        self.out.append(text)

    def need(self, rids):
        rids = [r for r in rids if r not in self.done]
        if not rids:
            return
        last = max(self.reads.index(r) for r in rids)
        if not os.environ.get("FK_GEN_ABLATE_LGKM"):            # timing experiments only (wrong results): no waits for LDS reads
            self.out.append(f"s_waitcnt lgkmcnt({min(len(self.reads) - 1 - last, 15)})")
        self.done.update(self.reads[:last + 1])
        self.reads = self.reads[last + 1:]


def schedule(mf, lds, va, dma_at, lds_per_gap, valu_units, tail, mid=None):
    """mf: [None, (text, [lds ids], [valu ids]), ...]; lds: id -> (text, release gap, deadline MFMA);
    va: id -> (text, cost in 4-cycle units, release gap, deadline MFMA, [producer ids]); dma_at: gap -> (set-M0 text, request text);
    tail: instructions that end the step; mid: gap -> instructions placed right behind that MFMA (a wait + barrier inside the step).
    An LDS read with a deadline past the last MFMA has no consumer in this step (a prefetch for the next one): it is issued in the gaps
    from its release on and waited for (lgkmcnt(0)) in front of the tail.  Returns the instruction list."""
    M = Model()
    lds_todo = dict(lds)
    va_todo = dict(va)
    va_pos = {}

    def issue_lds(gap, cap, only_due=None):
        n = 0
        while n < cap:
            cands = [(v[2], k) for k, v in lds_todo.items() if v[1] <= gap and (only_due is None or v[2] <= only_due)]
            if not cands:
                break
            _, k = min(cands)
            M.lds(k, lds_todo.pop(k)[0])
            n += 1

    def issue_valu(gap, units, only_due=None):
        while units > 0:
            n = len(M.out)
            cands = [(v[3], k) for k, v in va_todo.items()
                     if v[2] <= gap and v[1] <= units and (only_due is None or v[3] <= only_due)
                     and all(d in va_pos and n - va_pos[d] >= 2 for d in v[4])]
            if not cands:
                break
            _, k = min(cands)
            v = va_todo.pop(k)
            va_pos[k] = len(M.out)
            M.emit(v[0])
            units -= v[1]

    # gap 0: what the first MFMA needs, nothing else (a long burst fills the LDS command queue and stalls the issue of everything behind it)
    issue_lds(0, 99, only_due=1)
    for g in range(1, len(mf)):
        text, lneed, vneed = mf[g]
        # late producers: VALU results this MFMA consumes must exist (and be 2 instructions old); flush them if the gaps did not fit them
        missing = [k for k in vneed if k in va_todo]
        while missing:
            before = len(va_todo)
            issue_valu(99, 99, only_due=g)
            missing = [k for k in vneed if k in va_todo]
            if len(va_todo) == before:
                M.emit("s_nop 0")
        while any(len(M.out) - va_pos[k] < 2 for k in vneed):
            M.emit("s_nop 0")
        for k in lneed:
            if k in lds_todo:                                   # not released / scheduled in time: issue now
                M.lds(k, lds_todo.pop(k)[0])
        M.need(lneed)
        M.emit(text)
        if mid and g in mid:
            for t in mid[g]:
                M.emit(t)
        if g == 1:
            issue_lds(1, 99, only_due=2)                        # what the second MFMA needs
        if g in dma_at:
            M.emit(dma_at[g][0])
        n0 = len(M.out)
        issue_lds(g, lds_per_gap)
        issue_valu(g, valu_units)
        if g in dma_at:
            if len(M.out) == n0:
                M.emit("s_nop 0")                               # one wait state between the write of M0 and the LDS-DMA that uses it
            M.emit(dma_at[g][1])
    while va_todo:                                              # work without a consumer inside the step (results carried to the next one)
        before = len(va_todo)
        issue_valu(99, 99)
        if len(va_todo) == before:
            M.emit("s_nop 0")
    for k in list(lds_todo):                                    # prefetches the gaps did not fit
        M.lds(k, lds_todo.pop(k)[0])
    if M.reads:                                                 # reads without a consumer in this step: landed before the step ends
        M.need(list(M.reads))
    assert not lds_todo and not va_todo and not M.reads, (lds_todo.keys(), va_todo.keys(), M.reads)
    for t in tail:
        M.emit(t)
    # timing experiments only (wrong results): leave out one class of instructions
    drop = [pre for flag, pre in (("FK_GEN_ABLATE_VALU", ("v_exp", "v_add", "v_mul", "v_max", "v_cvt", "v_pk")), ("FK_GEN_ABLATE_LDS", ("ds_read",)),
                                   ("FK_GEN_ABLATE_MFMA", ("v_mfma",)), ("FK_GEN_ABLATE_EXP", ("v_exp",))) if os.environ.get(flag)]
    for pre in drop:
        M.out = [i for i in M.out if not i.startswith(pre)]
    return M.out


# M0 (written by the streams for the LDS-DMA destination) is a reserved register to hipcc: it cannot be named as a clobber, and hipcc
# itself sets M0 immediately in front of each of its own uses (none in the kernels that include these streams).
def clobbers(lo, hi):
    return ", ".join(f'"v{r}"' for r in range(lo, hi))
