#!/usr/bin/env python3
"""Per-kernel statistics of a `hipcc -S` listing: register counts, instruction histogram, wait / nop census.

Used for the side-by-side comparison of two builds of k_fast_blur_stream (profiles/r4_hazard_isa_diff.md) and by
tests/test_isa_lint.py. Usage: isa_kernel_stats.py <listing.s> <kernel-name-substring> [--dump]"""
import collections
import re
import sys


def kernel_body(text, name_sub):
    """Instruction lines of the first kernel whose mangled name contains name_sub, and its trailing metadata comment block."""
    lines = text.splitlines()
    start = None
    for i, ln in enumerate(lines):
        head = ln.split(";")[0].strip()
        if head.endswith(":") and name_sub in head and not ln.startswith("\t") and not ln.startswith(".L"):
            start = i
            break
    if start is None:
        raise SystemExit("kernel %r not found" % name_sub)
    body, meta = [], {}
    for ln in lines[start + 1:]:
        s = ln.split(";")[0].strip() if not ln.strip().startswith(";") else ln.strip()
        if s.startswith(".section") or s.startswith(".rodata") or s.startswith(".amdhsa_kernel"):
            break
        if s.startswith("s_endpgm"):
            body.append(s)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        body.append(s)
    tail = "\n".join(lines[start:start + 20000])
    for key in ("NumVgprs", "NumAgprs", "TotalNumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        m = re.search(r"; %s: (\d+)" % key, tail)
        if m:
            meta[key] = int(m.group(1))
    return body, meta


def scratch_accesses(text, name_sub):
    """(in_loop, outside) scratch_load / scratch_store lines of the kernel: LLVM's listing marks every basic block that belongs
    to a loop ("in Loop: Header=...", "Loop Header", "Parent Loop") on its label or its "; %bb.N:" comment."""
    lines = text.splitlines()
    start = None
    for i, ln in enumerate(lines):
        head = ln.split(";")[0].strip()
        if head.endswith(":") and name_sub in head and not ln.startswith("\t") and not ln.startswith(".L"):
            start = i
            break
    if start is None:
        raise SystemExit("kernel %r not found" % name_sub)
    in_loop, outside, looping = [], [], False
    for ln in lines[start + 1:]:
        t = ln.strip()
        if t.startswith(".section") or t.startswith(".amdhsa_kernel"):
            break
        if t.startswith(".LBB") or t.startswith("; %bb."):
            looping = "Loop" in t
        elif t.startswith("scratch_"):
            (in_loop if looping else outside).append(t)
    return in_loop, outside


def stats(body):
    hist = collections.Counter(ln.split()[0] for ln in body)
    waits = collections.Counter(ln for ln in body if ln.startswith("s_waitcnt"))
    nops = collections.Counter(ln for ln in body if ln.startswith("s_nop"))
    return hist, waits, nops


PK_F32 = re.compile(r"^v_pk_(add|mul|fma)_f32\b")


def swizzled_pk_f32(body):
    """Packed-fp32 instructions whose op_sel / op_sel_hi select a CROSSED half (low result from a high half or vice versa).

    The natural (unswizzled) form is op_sel:[0,0(,0)] op_sel_hi:[1,1(,1)], which the assembler prints without modifiers.
    op_sel_hi:[x,0] with op_sel:[x,0] for an operand is the BROADCAST form (both results from the low half), used by
    k_describe for scalar * vector products; it is reported separately."""
    crossed, broadcast = [], []
    for ln in body:
        if not PK_F32.match(ln):
            continue
        sel = re.search(r"op_sel:\[([01,]+)\]", ln)
        selh = re.search(r"op_sel_hi:\[([01,]+)\]", ln)
        lo = [int(x) for x in sel.group(1).split(",")] if sel else None
        hi = [int(x) for x in selh.group(1).split(",")] if selh else None
        if lo is None and hi is None:
            continue
        n = len(lo or hi)
        lo = lo or [0] * n
        hi = hi or [1] * n
        kinds = []
        for a, b in zip(lo, hi):
            kinds.append("n" if (a, b) == (0, 1) else "x" if (a, b) == (1, 0) else "b")   # natural / crossed / broadcast
        (crossed if "x" in kinds else broadcast).append(ln)
    return crossed, broadcast


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    body, meta = kernel_body(text, sys.argv[2])
    hist, waits, nops = stats(body)
    print("kernel", sys.argv[2], "instructions", len(body), meta)
    for op, n in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("%6d %s" % (n, op))
    print("-- s_waitcnt forms"); [print("%6d %s" % (n, w)) for w, n in sorted(waits.items(), key=lambda kv: -kv[1])]
    print("-- s_nop forms"); [print("%6d %s" % (n, w)) for w, n in sorted(nops.items(), key=lambda kv: -kv[1])]
    c, b = swizzled_pk_f32(body)
    print("-- packed fp32 with a crossed half:", len(c), " broadcast form:", len(b))
    if "--dump" in sys.argv:
        print("\n".join(body))
