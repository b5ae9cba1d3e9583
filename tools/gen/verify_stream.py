#!/usr/bin/env python3
"""Checks the generated instruction streams (csrc/attn_*_asm.inc) against the rules their generators promise, by replaying every asm block:

  1. LDS reads return in order: a register written by a ds_read is not touched (read or written) while the read may still be outstanding
     according to the counted s_waitcnt lgkmcnt(n) instructions of the block; nothing is outstanding at the end of the block.
  2. A VALU result is not consumed (by VALU or as an MFMA operand) by the instruction right behind it.
  3. An MFMA result is read or overwritten by a VALU instruction only after enough further MFMAs have been issued for it to have retired
     (two for the 8-pass 32x32x16, three for the 4-pass 16x16x32).
  4. A ds_read overwrites a register only after the last MFMA that read it as an operand AND one more MFMA have been issued.
  5. M0 is written at least one instruction ahead of the LDS-DMA request that uses it.
  6. (mlp_bwd_asm.inc) ds_write counts in lgkmcnt like a read; neither it nor a global_store reads a register of an outstanding ds_read, nor
     the VALU result of the instruction right in front of it.
Only the hard-coded temporaries (vNNN) are tracked; named operands (%[..]) are hipcc's registers and never alias them.
usage: verify_stream.py file.inc [...]   -> exit code 1 and a list of violations on failure
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def split_ops(ins):
    op, _, rest = ins.partition(" ")
    rest = re.sub(r"offset:\d+", "", rest)
    return op, [x.strip() for x in rest.split(",") if x.strip()]


def check_block(name, ins):
    errs = []
    lds_q = []                     # outstanding ds_reads, oldest first: (index, set of dst regs)
    last_valu = None               # (index, dst regs) of the previous instruction if it was VALU
    mfma_count = 0
    mfma_dst = {}                  # reg -> (mfma ordinal that wrote it, passes)
    mfma_src = {}                  # reg -> ordinal of the last MFMA that read it as A / B / C
    m0_at = None
    for i, s in enumerate(ins):
        op, ops = split_ops(s)
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                while len(lds_q) > n:
                    lds_q.pop(0)
            last_valu = None
            continue
        if op in ("s_barrier", "s_nop"):
            last_valu = None
            continue
        if op.startswith("s_add_u32") and ops and ops[0] == "m0":
            m0_at = i
            last_valu = None
            continue
        if op.startswith("global_load_lds"):
            if m0_at is None or i - m0_at < 2:
                errs.append(f"{name}[{i}] {s}: M0 written {0 if m0_at is None else i - m0_at} instructions earlier")
            last_valu = None
            continue
        if op.startswith("ds_write"):                 # an LDS operation like a read (lgkmcnt counts it), without a destination register
            src = [r for o in ops for r in regs(o)]
            for (j, d) in lds_q:
                if d & set(src):
                    errs.append(f"{name}[{i}] {s}: reads v{sorted(d & set(src))[0]} of the ds_read at [{j}] that may still be outstanding")
            if last_valu and set(last_valu[1]) & set(src):
                errs.append(f"{name}[{i}] {s}: stores the VALU result of the instruction right in front of it")
            lds_q.append((i, set()))
            last_valu = None
            continue
        if op.startswith("global_store"):
            src = [r for o in ops for r in regs(o)]
            for (j, d) in lds_q:
                if d & set(src):
                    errs.append(f"{name}[{i}] {s}: stores v{sorted(d & set(src))[0]} of the ds_read at [{j}] that may still be outstanding")
            last_valu = None
            continue
        dst = regs(ops[0]) if ops else []
        src = [r for o in ops[1:] for r in regs(o)]
        touched = set(dst) | set(src)
        for (j, d) in lds_q:
            if d & touched:
                errs.append(f"{name}[{i}] {s}: touches v{sorted(d & touched)[0]} of the ds_read at [{j}] that may still be outstanding")
        if last_valu and set(last_valu[1]) & set(src):
            errs.append(f"{name}[{i}] {s}: consumes the VALU result of the instruction right in front of it")
        if op.startswith("ds_read"):
            for r in dst:
                if r in mfma_src and mfma_count - mfma_src[r] < 1:
                    errs.append(f"{name}[{i}] {s}: overwrites v{r}, an operand of the MFMA issued last")
            lds_q.append((i, set(dst)))
            last_valu = None
        elif op.startswith("v_mfma"):
            mfma_count += 1
            passes = 4 if "16x16x32" in op else 8
            for r in src:
                mfma_src[r] = mfma_count
            for r in dst:
                mfma_dst[r] = (mfma_count, passes)
            last_valu = None
        elif op.startswith("v_"):
            for r in touched:
                if r in mfma_dst:
                    k, passes = mfma_dst[r]
                    need = 2 if passes == 8 else 3
                    if mfma_count - k < need:
                        errs.append(f"{name}[{i}] {s}: v{r} is the result of MFMA #{k}, only {mfma_count - k} further MFMA(s) issued")
            for r in dst:
                mfma_dst.pop(r, None)
            last_valu = (i, dst)
        else:
            errs.append(f"{name}[{i}] {s}: unknown instruction class")
    if lds_q:
        errs.append(f"{name}: {len(lds_q)} ds_read(s) not waited for at the end of the block")
    return errs


def blocks(text):
    for m in re.finditer(r"FK_DEV void (\w+)\(.*?asm volatile\(\n(.*?)\n\s*:", text, flags=re.S):
        ins = re.findall(r'"(.*?)\\n\\t"', m.group(2))
        yield m.group(1), ins


def main():
    bad = 0
    for path in sys.argv[1:]:
        text = open(path).read()
        n = 0
        for name, ins in blocks(text):
            n += 1
            for e in check_block(name, ins):
                print(f"{path}: {e}")
                bad += 1
        print(f"{path}: {n} blocks checked")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
