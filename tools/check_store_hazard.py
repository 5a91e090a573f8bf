#!/usr/bin/env python3
"""check_store_hazard.py <object or .s / .dis file>... — build gate for the wide-store write-data hazard.

gfx950: a buffer/global/flat store of MORE than 8 bytes reads its data registers late; a VALU instruction that overwrites one of
them needs 2 wait states behind the store.  The compiler's hazard recogniser inserts them (`s_nop`, or independent
instructions) -- but for a `buffer_store_dwordx4` whose row offset sits in an SGPR it does not (it models the hazard as absent
there), and round 3 saw the second dword of such a store arrive corrupted now and then (the kernel now keeps
the scalar offset 0).  Nothing in the language stops an edit or a compiler from re-creating that form, so the kept disassembly
is scanned: every store of 3 or 4 dwords, of any encoding, must be followed by at least WAIT_STATES wait states before a
VALU instruction writes one of its data registers.  A wait state = one instruction issued (`s_nop N` counts N + 1).

Exit code 1 and one line per violation; used by csrc/Makefile on each overlap-save unit and by tests/test_host.py.
"""
import os
import re
import subprocess
import sys
import tempfile

WAIT_STATES = 2
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

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


def _instr(line):
    """mnemonic and operand string of a disassembly / assembly line, or None (labels, directives, comments)."""
    line = line.split("//")[0].split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    m = re.match(r"^(?:[0-9a-f]+:\s+)?([a-z][a-z0-9_]*)\s*(.*)$", line)
    if not m:
        return None
    return m.group(1), m.group(2)


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


def scan_text(text, where=""):
    """list of violation strings"""
    lines = text.splitlines()
    ins = []
    for i, ln in enumerate(lines):
        d = _instr(ln)
        if d:
            ins.append((i + 1, d[0], d[1]))
    bad = []
    for k, (lineno, mn, ops) in enumerate(ins):
        if not _STORE.match(mn + " " + ops):
            continue
        data = _store_data(mn, ops)
        if not data:
            continue
        ws = 0
        j = k + 1
        while j < len(ins) and ws < WAIT_STATES:
            _, m2, o2 = ins[j]
            if m2 in ("s_endpgm", "s_branch", "s_setpc_b64") or m2.startswith("s_cbranch"):
                break
            for (a, b) in _valu_writes(m2, o2):
                if a <= data[1] and b >= data[0]:
                    bad.append("%s:%d: `%s %s` overwrites data registers v[%d:%d] of the %d-dword store at line %d after %d wait "
                               "state(s) (need %d)" % (where, ins[j][0], m2, o2.strip(), data[0], data[1], data[1] - data[0] + 1,
                                                       lineno, ws, WAIT_STATES))
            if m2 == "s_nop":
                try:
                    ws += int(o2.strip(), 0) + 1
                except ValueError:
                    ws += 1
            else:
                ws += 1
            j += 1
    return bad


def count_wide_stores(text):
    return sum(1 for ln in text.splitlines() if (lambda d: d and _STORE.match(d[0] + " " + d[1]))(_instr(ln)))


def disassemble(path):
    """device disassembly of a hipcc object (offload bundle), or the text of a .s / .dis file"""
    if path.endswith((".s", ".dis", ".txt")):
        return open(path).read()
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(path))
        os.symlink(os.path.abspath(path), local)
        subprocess.check_call([OBJDUMP, "--offloading", local], cwd=td, stdout=subprocess.DEVNULL)
        cos = [f for f in os.listdir(td) if "amdgcn" in f]
        if not cos:
            return ""   # a host-only object
        return "".join(subprocess.check_output([OBJDUMP, "-d", os.path.join(td, f)], text=True) for f in cos)


def main(argv):
    rc = 0
    for p in argv:
        text = disassemble(p)
        bad = scan_text(text, os.path.basename(p))
        print("%s: %d wide stores, %d hazard violation(s)" % (os.path.basename(p), count_wide_stores(text), len(bad)))
        for b in bad:
            print("  " + b)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
