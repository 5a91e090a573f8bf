#!/usr/bin/env python3
"""check_store_hazard.py <object or .s / .dis file>... — build gate for the wide-store write-data hazard.

gfx950: a buffer/global/flat store of MORE than 8 bytes reads its data registers late; a VALU instruction that overwrites one of
them needs 2 wait states behind the store.  The compiler's hazard recogniser inserts them (`s_nop`, or independent
instructions) -- but for a `buffer_store_dwordx4` whose row offset sits in an SGPR it does not (it models the hazard as absent
there), and round 3 saw the second dword of such a store arrive corrupted now and then (the kernel now keeps
the scalar offset 0).  Nothing in the language stops an edit or a compiler from re-creating that form, so the kept disassembly
is scanned: every store of 3 or 4 dwords, of any encoding, must be followed by at least WAIT_STATES wait states before a
VALU instruction writes one of its data registers.  A wait state = one instruction issued (`s_nop N` counts N + 1).

Round 5: the scan no longer stops at a conditional branch -- the fall-through path goes on (the branch counts as one
wait state) AND the branch target is scanned with the same count (a loop's back edge: the stores of the kernels sit at the end
of the per-block loop, the first VALU writes of the next iteration at its head); an unconditional branch follows its target.  An
object that yields no device disassembly, or a device object without a single wide store where the caller expects some
(--expect-stores), is an error instead of "0 violations".

Exit code 1 and one line per violation; used by csrc/Makefile on each overlap-save unit and by tests/test_host.py.
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_tools  # noqa: E402

WAIT_STATES = 2

_STORE = re.compile(r"^\s*(buffer_store_(?:dwordx[34]|format_xyzw?)|global_store_dwordx[34]|flat_store_dwordx[34]|"
                    r"scratch_store_dwordx[34])\s+(.*)$")
_VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def _regs(tok):
    m = _VREG.fullmatch(tok.strip())
    if not m:
        return None
    if m.group(3) is not None:
        return int(m.group(3)), int(m.group(3))
    return int(m.group(1)), int(m.group(2))


_instr = isa_tools.instr
_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<[^>+]+\+0x([0-9A-Fa-f]+)>|<([^>+]+)>\s*$")


def _store_data(mn, ops):
    """the data register range of a wide store (first VGPR-range operand of more than two registers)."""
    toks = [t.strip() for t in ops.split(",")]
    cands = toks[:1] if mn.startswith("buffer_") else toks[:3]
    for t in cands:
        r = _regs(t.split()[0]) if t else None
        if r and r[1] - r[0] >= 2:
            return r
    return None


def _valu_writes(mn, ops):
    """VGPR ranges a VALU instruction writes (its first operand; the permlane swaps write both)."""
    if not mn.startswith("v_") or mn.startswith(("v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane")):
        return []
    toks = [t.strip() for t in ops.split(",")]
    out = []
    n = 2 if "swap" in mn else 1
    for t in toks[:n]:
        r = _regs(t.split()[0]) if t else None
        if r:
            out.append(r)
    return out


def _branch_target(line, labels, addr_index, sym_addr):
    """instruction index a branch goes to: an assembly label operand, or the `<symbol+0xoff>` of a disassembly line"""
    d = _instr(line)
    tok = d[1].split(",")[-1].strip().split()[0] if d and d[1].strip() else ""
    if tok in labels:
        return labels[tok]
    m = re.search(r"<([^>+]+)\+0x([0-9A-Fa-f]+)>", line)
    if m and m.group(1) in sym_addr:
        return addr_index.get(sym_addr[m.group(1)] + int(m.group(2), 16))
    m = re.search(r"<([^>+]+)>\s*$", line)
    if m and m.group(1) in sym_addr:
        return addr_index.get(sym_addr[m.group(1)])
    return None


def scan_text(text, where=""):
    """list of violation strings"""
    lines = text.splitlines()
    ins = []          # (line number, mnemonic, operands, raw line)
    labels = {}       # assembly label -> index of the next instruction
    addr_index = {}   # disassembly address -> instruction index
    sym_addr = {}     # disassembly symbol -> address
    for i, ln in enumerate(lines):
        lab = isa_tools.label(ln)
        if lab is not None:
            labels[lab] = len(ins)
            m = re.match(r"^([0-9a-fA-F]+)\s+<", ln.strip())
            if m:
                sym_addr[lab] = int(m.group(1), 16)
            continue
        d = _instr(ln)
        if d:
            m = _ADDR.search(ln)
            if m:
                addr_index[int(m.group(1), 16)] = len(ins)
            ins.append((i + 1, d[0], d[1], ln))
    bad = set()
    for k, (lineno, mn, ops, _) in enumerate(ins):
        if not _STORE.match(mn + " " + ops):
            continue
        data = _store_data(mn, ops)
        if not data:
            continue
        # walk every path behind the store until WAIT_STATES wait states have passed
        work, seen = [(k + 1, 0)], set()
        while work:
            j, ws = work.pop()
            while j < len(ins) and ws < WAIT_STATES and (j, ws) not in seen:
                seen.add((j, ws))
                l2, m2, o2, raw = ins[j]
                if m2 in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                    break
                if m2 == "s_branch" or m2.startswith("s_cbranch"):
                    t = _branch_target(raw, labels, addr_index, sym_addr)
                    if t is None:
                        bad.add("%s:%d: the %d-dword store at line %d is followed within %d wait state(s) by `%s %s` whose target the "
                                "scanner cannot resolve" % (where, l2, data[1] - data[0] + 1, lineno, ws, m2, o2.strip()))
                    else:
                        work.append((t, ws + 1))
                    if m2 == "s_branch":
                        break
                    ws += 1
                    j += 1
                    continue
                for (a, b) in _valu_writes(m2, o2):
                    if a <= data[1] and b >= data[0]:
                        bad.add("%s:%d: `%s %s` overwrites data registers v[%d:%d] of the %d-dword store at line %d after %d wait "
                                "state(s) (need %d)" % (where, l2, m2, o2.strip(), data[0], data[1], data[1] - data[0] + 1,
                                                        lineno, ws, WAIT_STATES))
                if m2 == "s_nop":
                    try:
                        ws += int(o2.strip(), 0) + 1
                    except ValueError:
                        ws += 1
                else:
                    ws += 1
                j += 1
    return sorted(bad)


def count_wide_stores(text):
    return sum(1 for ln in text.splitlines() if (lambda d: d and _STORE.match(d[0] + " " + d[1]))(_instr(ln)))


def disassemble(path):
    """device disassembly of a hipcc object (offload bundle), or the text of a .s / .dis file; a bundle without device code raises"""
    return isa_tools.disassemble(path)


def main(argv):
    expect = False
    if argv and argv[0] == "--expect-stores":
        expect, argv = True, argv[1:]
    rc = 0
    for p in argv:
        try:
            text = disassemble(p)
        except isa_tools.NoDeviceCode as e:
            print("%s: %s" % (os.path.basename(p), e))
            rc = 1
            continue
        bad = scan_text(text, os.path.basename(p))
        n = count_wide_stores(text)
        print("%s: %d wide stores, %d hazard violation(s)" % (os.path.basename(p), n, len(bad)))
        if not any(_instr(ln) for ln in text.splitlines()):
            print("  %s: the disassembly holds no instruction" % os.path.basename(p))
            rc = 1
        if expect and n == 0:
            print("  %s: no wide store found where the build expects some (--expect-stores): is the disassembly what it should be?" % os.path.basename(p))
            rc = 1
        for b in bad:
            print("  " + b)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
