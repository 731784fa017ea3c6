#!/usr/bin/env python3
"""ISA check for the gfx950 store-data hazard found in round 1.

A vector store of more than 64 bits reads its data VGPRs over several cycles after it has been
issued.  hipcc (ROCm 7.2) keeps one wait state between such a store and a VALU write of one of its
data registers only when the store has no SGPR soffset; with an SGPR soffset it schedules the
overwrite directly behind the store.  On MI355X that corrupts the upper data dwords of the last
lanes of each 16-lane group now and then (K2's `buffer_store_dwordx4 v[2:5], ..., s30 offen`
followed by `v_sub_f32 v4, ...`: the intermittent wide-plan failure of DESIGN.md section 3).

The requirement used here is the one the compiler itself applies to gfx940-class stores without an
SGPR soffset: two wait states between the store and the overwrite (an instruction counts one wait
state, `s_nop N` counts N + 1).  The library's 16-byte buffer store helper adds `s_nop 1` itself.

This script disassembles nothing itself: give it the .s files hipcc writes with
`--cuda-device-only -S`.  It reports every VMEM / FLAT instruction with more than 64 bits of store data (dwordx3/x4, typed and
formatted xyz/xyzw stores, 128-bit compare-and-swap) whose data registers are written again with
fewer than NEED wait states in between, in straight-line code.  Exit code 1 if any is found."""
import re
import sys

WINDOW = 12
NEED = 2
# every VMEM / FLAT instruction whose store data is wider than 64 bits: plain, typed and
# formatted stores, and the 128-bit compare-and-swap atomics (data = two 64-bit values)
WIDE = r"(?:dwordx[34]|format_xyzw?|format_d16_hi_xyzw?|b96|b128)"
STORE = re.compile(r"^\s*((?:buffer|tbuffer|global|flat|scratch)_store_" + WIDE +
                   r"|(?:buffer|global|flat)_atomic_cmpswap_x2)\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = VREG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def written(line):
    """VGPRs an instruction writes (first operand of VALU / loads / ds reads / accvgpr moves)."""
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        return set()
    op, _, rest = t.partition(" ")
    if op.startswith(("s_", "buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store")):
        return set()
    first = rest.split(",")[0]
    return regs(first)


def check(path):
    lines = open(path).read().split("\n")
    kernel = None
    found = []
    body = []
    for ln in lines:
        if re.match(r"^[A-Za-z_][\w$.]*:", ln) and not ln.startswith(".L"):
            kernel = ln.split(":")[0]
        t = ln.split(";")[0].rstrip()
        if t.strip() and not t.strip().startswith("."):
            body.append((kernel, t))
    for i, (k, t) in enumerate(body):
        m = STORE.match(t)
        if not m:
            continue
        ops = m.group(2).split(",")
        # operand order: buffer/tbuffer: vdata first; global/flat/scratch: address first, then vdata
        # (a returning atomic has its destination in front of both)
        first_is_data = m.group(1).startswith(("buffer", "tbuffer"))
        idx = 0 if first_is_data else 1
        if "atomic" in m.group(1) and re.search(r"\b(sc0|glc)\b", t):
            idx += 1
        data = regs(ops[idx]) if len(ops) > idx else set()
        waits = 0
        for j in range(i + 1, min(i + 1 + WINDOW, len(body))):
            k2, t2 = body[j]
            s2 = t2.strip()
            if s2.endswith(":") or s2.startswith(("s_endpgm", "s_branch", "s_cbranch", "s_barrier", "s_setpc")):
                break
            hit = written(t2) & data
            if hit:
                if waits < NEED:
                    found.append((k, t.strip(), waits, s2, sorted(hit)))
                break
            m2 = re.match(r"s_nop\s+(\d+)", s2)
            waits += (int(m2.group(1)) + 1) if m2 else 1
            if waits >= NEED:
                break
    return found


def main():
    bad = 0
    for path in sys.argv[1:]:
        for k, store, dist, instr, hit in check(path):
            bad += 1
            print(f"{path}: {k}\n    {store}\n    after {dist} wait state(s): {instr}   (overwrites v{hit})")
    print(f"{bad} store(s) whose data registers are overwritten after fewer than {NEED} wait states")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
