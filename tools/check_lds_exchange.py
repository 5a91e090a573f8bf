#!/usr/bin/env python3
"""check_lds_exchange.py [--source if_fir_fft.hip] <object or -l disassembly>... — build gate for the order of the LDS lane
exchanges of the overlap-save kernels.

The kernels' lanes exchange data through a wave-private LDS buffer: 16 stores per lane, then 16 loads, then the next
exchange's 16 stores into the same buffer.  The hardware runs a wave's LDS instructions in order, so the exchange is correct
exactly when the INSTRUCTION STREAM keeps the phases apart -- a load ahead of its phase's last store reads a slot another lane has
not written yet, a store ahead of the previous phase's last load overwrites a slot another lane has not read yet.  Both would be
silent data corruption.  The source pins the order (every exchange load goes through an opaque base address, so no alias-freedom
proof exists: csrc/if_fir_fft.hip, lds_opaque); this gate checks the result in the generated code:

  * every exchange access is made by one of two helpers, `xst16<>` / `xld16<>` (their lines are tagged `LDSX:STORE` / `LDSX:LOAD`);
  * the units are compiled with -gline-tables-only (no effect on the generated code), so every DS instruction of the
    disassembly (`llvm-objdump -d -l`) carries a source line: the helper's own line, or -- when the compiler merged two inlined
    calls into one ds_read2_b64 -- the line of the call site.  DS instructions attributed to a line that holds one of the helpers
    or calls one are exchange stores / loads, everything else (table reads, the block queue's words) is ignored; a DS
    instruction of the exchanges' width WITHOUT a source line cannot be classified and fails the build;
  * per kernel, in program order, the exchange stream must be (16 stores, 16 loads)*: a load while fewer than 16 stores of the
    phase have been issued, a 17th store without loads in between, a store while fewer than 16 loads of the phase have been issued,
    or an incomplete last phase fails the build.  ds_read2/ds_write2 (two elements per instruction) count twice.
  * a kernel of the overlap-save family (`fir_fft_kernel`, `fir_odd_kernel`) without any classified access fails too (line
    tables missing, helpers renamed: the gate would otherwise pass everything).

Exit code 1 and one line per violation; used by csrc/Makefile on every overlap-save unit and by tests/test_host.py (which also
compiles a deliberately mis-ordered probe and sees it flagged)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_tools  # noqa: E402

PHASE = 16
_LINE = re.compile(r"^;\s*(\S+):(\d+)\s*$")
_WEIGHT = {"ds_write_b64": 1, "ds_write2_b64": 2, "ds_write2st64_b64": 2, "ds_write_b128": 2,
           "ds_read_b64": 1, "ds_read2_b64": 2, "ds_read2st64_b64": 2, "ds_read_b128": 2}
FAMILY = ("fir_fft_kernel", "fir_odd_kernel")


_CALL_W, _CALL_R = re.compile(r"\bxst16<"), re.compile(r"\bxld16<")


SOURCE_PARTS = ("if_fir_fft.hip", "if_fir_fft_dev.h", "if_fir_fft_odd.inc", "if_fir_fft_kernel.inc", "if_fir_fft_tails_bank.inc",
                "if_fir_fft_tails_single.inc", "if_fir_fft_launch.inc")


def source_files(source):
    """the kernel source and the parts it includes (round 5: if_fir_fft.hip is a driver that includes its families)"""
    d = os.path.dirname(os.path.abspath(source))
    return [os.path.join(d, f) for f in SOURCE_PARTS if os.path.exists(os.path.join(d, f))]


def tagged_lines(source):
    """{(file name, line number): 'W' | 'R'}: the helpers' own lines (tagged LDSX:STORE / LDSX:LOAD) and every line that calls one of
    them (a DS instruction the compiler merged from two accesses of different inlined copies carries a CALL SITE's line), over the
    kernel source and every part it includes"""
    out, helpers = {}, set()
    for path in source_files(source):
        base = os.path.basename(path)
        for k, ln in enumerate(open(path), 1):
            code = ln.split("//")[0]
            w = bool(_CALL_W.search(code)) or "LDSX:STORE" in code
            r = bool(_CALL_R.search(code)) or "LDSX:LOAD" in code
            if w and r:
                raise SystemExit("check_lds_exchange: %s:%d holds a store and a load of the exchanges: the gate classifies by source line" % (path, k))
            if w:
                out[(base, k)] = "W"
            if r:
                out[(base, k)] = "R"
            if "LDSX:STORE" in code and "reinterpret_cast" in code:
                helpers.add("W")
            if "LDSX:LOAD" in code and "reinterpret_cast" in code:
                helpers.add("R")
    if helpers != {"R", "W"}:
        raise SystemExit("check_lds_exchange: %s must hold the LDSX:STORE and LDSX:LOAD helper lines (found %s)" % (source, sorted(helpers)))
    return out


def scan_text(text, tags, source_base=None, where=""):
    """(violations, {kernel: exchanges}) of a line-annotated disassembly"""
    bad, units = [], {}
    kern, cur, known = None, None, False
    phase, cnt, nunits, first_line = "R", PHASE, 0, 0   # "a complete load phase is behind us": a store phase may start

    def close():
        if kern is None:
            return
        if not (phase == "R" and cnt == PHASE):
            bad.append("%s: %s: the exchange stream ends inside a phase (%s, %d of %d)" % (where, kern[:100], phase, cnt, PHASE))
        units[kern] = nunits

    for lineno, ln in enumerate(text.splitlines(), 1):
        lab = isa_tools.label(ln)
        if lab is not None and not lab.startswith(".L") and "+0x" not in lab:
            close()
            kern, cur, known = lab, None, False
            phase, cnt, nunits = "R", PHASE, 0
            continue
        m = _LINE.match(ln.strip())
        if m:
            known = int(m.group(2)) > 0
            cur = tags.get((os.path.basename(m.group(1)), int(m.group(2))))
            continue
        d = isa_tools.instr(ln)
        if not d or kern is None or not d[0].startswith("ds_"):
            continue
        if not known and d[0] in _WEIGHT and any(f in kern for f in FAMILY):
            bad.append("%s:%d: %s: `%s %s` has no source line: cannot be classified" % (where, lineno, kern[:60], d[0], d[1].strip()))
            continue
        if cur is None:
            continue
        mn = d[0]
        w = _WEIGHT.get(mn)
        if w is None or (cur == "W") != mn.startswith("ds_write"):
            bad.append("%s:%d: %s: `%s %s` on the exchange helpers' source line is not a %s the gate knows" %
                       (where, lineno, kern[:60], mn, d[1].strip(), "store" if cur == "W" else "load"))
            continue
        if cur == phase:
            cnt += w
            if cnt > PHASE:
                bad.append("%s:%d: %s: %s number %d of an exchange phase (`%s %s`): the other phase's instructions are not in "
                           "between" % (where, lineno, kern[:60], "store" if cur == "W" else "load", cnt, mn, d[1].strip()))
                cnt = w
        else:
            if cnt != PHASE:
                bad.append("%s:%d: %s: `%s %s` is issued after only %d of the %d %s of its exchange phase" %
                           (where, lineno, kern[:60], mn, d[1].strip(), cnt, PHASE, "stores" if phase == "W" else "loads"))
            phase, cnt = cur, w
            if cur == "W":
                nunits += 1
    close()
    for k, n in units.items():
        if any(f in k for f in FAMILY) and n == 0 and not k.endswith(".kd"):
            bad.append("%s: %s: no exchange access was classified (line tables missing? helpers renamed?)" % (where, k[:100]))
    return bad, units


def main(argv):
    source = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "qo-100-tools_amd", "csrc", "if_fir_fft.hip")
    if argv and argv[0] == "--source":
        source, argv = argv[1], argv[2:]
    tags = tagged_lines(source)
    rc = 0
    for p in argv:
        text = isa_tools.disassemble(p, lines=True)
        bad, units = scan_text(text, tags, os.path.basename(source), os.path.basename(p))
        fam = {k: n for k, n in units.items() if any(f in k for f in FAMILY) and not k.endswith(".kd")}
        print("%s: %d kernels, %d lane exchanges, %d order violation(s)" % (os.path.basename(p), len(fam), sum(fam.values()), len(bad)))
        if not fam:
            print("  %s: no overlap-save kernel found" % os.path.basename(p))
            rc = 1
        for b in bad[:40]:
            print("  " + b)
        if bad:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
