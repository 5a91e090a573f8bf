"""isa_tools.py — shared helpers of the build gates (check_store_hazard.py, check_lds_exchange.py): device disassembly of a
hipcc object (an offload bundle) with llvm-objdump, and a tolerant instruction-line parser."""
import os
import re
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


class NoDeviceCode(Exception):
    """the object unbundled to no amdgcn code object (a host-only object, or an unbundler that names its outputs differently)"""


def disassemble(path, lines=False, allow_host_only=False):
    """device disassembly of a hipcc object, or the text of a .s / .dis / .txt file.  lines=True: with source line comments
    (`; file:line`, needs -gline-tables-only).  Raises NoDeviceCode when the bundle yields no device code object -- a gate that
    silently scans nothing would pass everything -- unless allow_host_only."""
    if path.endswith((".s", ".dis", ".txt")):
        return open(path).read()
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(path))
        os.symlink(os.path.abspath(path), local)
        subprocess.check_call([OBJDUMP, "--offloading", local], cwd=td, stdout=subprocess.DEVNULL)
        cos = sorted(f for f in os.listdir(td) if "amdgcn" in f)
        if not cos:
            if allow_host_only:
                return ""
            raise NoDeviceCode("%s: llvm-objdump --offloading extracted no amdgcn code object (files: %s)" %
                               (path, sorted(os.listdir(td))))
        flags = ["-d", "-l"] if lines else ["-d"]
        return "".join(subprocess.check_output([OBJDUMP] + flags + [os.path.join(td, f)], text=True) for f in cos)


_INSTR = re.compile(r"^(?:[0-9a-f]+:\s+)?([a-z][a-z0-9_]*)\s*(.*)$")


def instr(line):
    """(mnemonic, operand string) of a disassembly / assembly line, or None (labels, directives, comments)."""
    line = line.split("//")[0].split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    m = _INSTR.match(line)
    if not m:
        return None
    return m.group(1), m.group(2)


_LABEL = re.compile(r"^(?:[0-9a-f]+\s+)?<([^>]+)>:\s*$|^([A-Za-z_.$][\w.$]*):")


def label(line):
    """name of a label line (`0000 <name>:` in a disassembly, `name:` in assembly), or None"""
    m = _LABEL.match(line.strip())
    if not m:
        return None
    return m.group(1) or m.group(2)
