#!/usr/bin/env python3
"""gen_walk.py — emits the hand-scheduled gfx950 assembly of the direct-form FIR "walk" (the FMA stream of one tile
for one wave) as inline-asm C++ functions, one per (T, D, R, SEG) instantiation, into
qo-100-tools_amd/csrc/generated/if_fir_walk_gen.h.

Why generated assembly: the walk is ~2000 v_pk_fma_f32 in straight-line code whose schedule decides the kernel's
speed (LDS reads must be issued a fixed distance ahead of their use, tap blocks must stream through a small SGPR
ring, waits must be counted).  hipcc's scheduler hoists every LDS read of the unrolled C++ version to the top and
spills; here every instruction is placed explicitly.

The walk, per lane (one lane = R consecutive decimated outputs, SPEC §2/§3):
  for each input sample c = -(T-1) .. D(R-1), oldest first (one ds_read_b64/_b128 per 1/2 samples, issued Q units
  ahead into a register ring):
      for r in 0..R-1:  k = D*r - c;  if 0 <= k < T:  acc[r] = fma(x[c], h[k], acc[r])        (v_pk_fma_f32: I and Q)
  taps h[k] sit in a ring of 4 x 16 SGPRs filled by s_load_dwordx16; the tap is broadcast to both halves of the
  packed FMA with op_sel.  Accumulation segments of SEG taps: the top segment accumulates straight into tot[r], every
  other segment into acc[r] and is added to tot[r] when it completes (descending segment order = SPEC §3).

lgkmcnt discipline (LDS reads return in order, scalar loads out of order, both count in lgkmcnt):
  * a wait for LDS data uses lgkmcnt(n) with n = LDS reads issued after the wanted one; outstanding scalar loads
    only make that wait more conservative (if the wanted read were still pending so would the n younger reads: n+1);
  * a tap block is only trusted after an lgkmcnt(0); the LDS reads that would be youngest at that drain are issued
    one step early so the drain finds them (almost) landed.
"""
import argparse
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "qo-100-tools_amd", "csrc", "generated", "if_fir_walk_gen.h")

SGPR_RING_BASE = 36      # s[36:99]: 4 blocks x 16 taps
SGPR_RING_BLOCKS = 4


def gen_walk(T, D, R, SEG, Q, U, W=1, nodrain=False):
    """Returns (function name, asm lines, clobber list, n_tot)."""
    assert T % 2 == 1 and U in (1, 2)
    DR = D * R
    HALO = ((T - 1 + DR - 1) // DR) * DR
    CH = DR * 8 + 16
    cmin, cmax = -(T - 1), D * (R - 1)
    unit_starts = list(range(cmin, cmax + 1, U))
    nunits = len(unit_starts)
    top_seg = (T - 1) // SEG

    def lds_off(c):
        u = c + HALO
        return (u // DR) * CH + (u % DR) * 8

    # ---- register map ---------------------------------------------------------------------------------------
    ring_slots = Q + W + 1                       # longest lead of a read over its use is Q + W units
    need = 2 * R + 2 * U * ring_slots
    acc_base = (256 - need) & ~3                 # temporaries sit at the top of the arch VGPR file
    ring_base = acc_base + 2 * R                 # slot i = v[ring_base + 2*U*i ...]
    vtop = ring_base + 2 * U * ring_slots
    assert vtop <= 256, "out of VGPRs"

    def acc(r):
        b = acc_base + 2 * r
        return "v[%d:%d]" % (b, b + 1)

    def tot(r):
        return "%%%d" % r                        # operands 0..R-1 are the tot outputs

    def slot_regs(unit_idx):
        b = ring_base + 2 * U * (unit_idx % ring_slots)
        return b

    def tap_pair(k):
        i = SGPR_RING_BASE + ((k % (16 * SGPR_RING_BLOCKS)) // 2) * 2
        return "s[%d:%d]" % (i, i + 1), (k % 2)

    # ---- which taps does unit u use -------------------------------------------------------------------------
    def taps_of_unit(ui):
        ks = []
        for c in range(unit_starts[ui], min(unit_starts[ui] + U, cmax + 1)):
            for r in range(R):
                k = D * r - c
                if 0 <= k < T:
                    ks.append(k)
        return ks

    nblocks = (T + 15) // 16
    first_use, last_use = {}, {}
    for ui in range(nunits):
        for k in taps_of_unit(ui):
            b = k // 16
            first_use.setdefault(b, ui)
            last_use[b] = ui
    # blocks are first used in descending order
    order = sorted(first_use, key=lambda b: first_use[b])
    prologue_blocks = order[:SGPR_RING_BLOCKS]
    smem_at = {}   # unit index -> list of blocks to load at the start of that unit
    drain_at = set()
    for b in order[SGPR_RING_BLOCKS:]:
        prev = b + SGPR_RING_BLOCKS          # block that occupied this ring slot
        issue = last_use[prev] + 1
        assert issue < first_use[b], "tap ring too small"
        smem_at.setdefault(issue, []).append(b)
        drain_at.add(first_use[b])

    # ---- LDS read issue schedule ---------------------------------------------------------------------------
    # A tap block is only trusted after an lgkmcnt(0), which also waits for every LDS read in flight.  No LDS read is
    # issued in the W steps ahead of a drain (round 3; W = 1 is round 1's "one step early"): whatever would fall there is
    # issued just before that window, so the drain finds the youngest read at least W steps old.
    issue_step = {}
    for ui in range(nunits):
        s = ui - Q
        moved = True
        while moved:
            moved = False
            for d in drain_at:
                if d - W <= s <= d - 1:
                    s = d - W - 1
                    moved = True
        issue_step[ui] = s
    reads_at = {}
    for ui, s in issue_step.items():
        reads_at.setdefault(max(s, -1), []).append(ui)   # -1 = prologue
    for k_ in reads_at:
        reads_at[k_].sort()

    lines = []
    emit = lines.append
    issued = []                                  # LDS reads in issue order

    def emit_read(ui):
        c = unit_starts[ui]
        b = slot_regs(ui)
        if U == 2:
            emit("ds_read_b128 v[%d:%d], %%%d offset:%d" % (b, b + 3, R, lds_off(c)))
        else:
            emit("ds_read_b64 v[%d:%d], %%%d offset:%d" % (b, b + 1, R, lds_off(c)))
        issued.append(ui)

    def emit_smem(b):
        slot = SGPR_RING_BASE + 16 * (b % SGPR_RING_BLOCKS)
        emit("s_load_dwordx16 s[%d:%d], %%%d, 0x%x" % (slot, slot + 15, R + 1, 64 * b))

    # ---- prologue ---------------------------------------------------------------------------------------------
    emit("s_waitcnt lgkmcnt(0)")                 # the tile was just written to LDS by this wave
    for b in prologue_blocks:
        emit_smem(b)
    for ui in reads_at.get(-1, []):
        emit_read(ui)
    emit("s_waitcnt lgkmcnt(0)")
    n_fma = 0
    # ---- steps ------------------------------------------------------------------------------------------------
    started = set()                              # (r) accumulators that hold a live partial sum
    for ui in range(nunits):
        if ui in drain_at and not nodrain:   # nodrain: TIMING STUDY ONLY (a tap block is used without waiting for it)
            emit("s_waitcnt lgkmcnt(0)")
        for b in smem_at.get(ui, []):
            emit_smem(b)
        for uj in reads_at.get(ui, []):
            emit_read(uj)
        younger = len(issued) - 1 - issued.index(ui)
        assert younger <= 15
        emit("s_waitcnt lgkmcnt(%d)" % younger)
        base = slot_regs(ui)
        for si, c in enumerate(range(unit_starts[ui], min(unit_starts[ui] + U, cmax + 1))):
            x = "v[%d:%d]" % (base + 2 * si, base + 2 * si + 1)
            for r in range(R):
                k = D * r - c
                if not (0 <= k < T):
                    continue
                seg = k // SEG
                in_top = (seg == top_seg)
                dst = tot(r) if in_top else acc(r)
                first = (k == T - 1) if in_top else (k % SEG == SEG - 1)
                sp, hi = tap_pair(k)
                if first:
                    mods = "op_sel:[0,1,0] op_sel_hi:[1,1,0]" if hi else "op_sel_hi:[1,0,0]"
                    emit("v_pk_fma_f32 %s, %s, %s, 0 %s" % (dst, x, sp, mods))
                else:
                    mods = "op_sel:[0,1,0]" if hi else "op_sel_hi:[1,0,1]"
                    emit("v_pk_fma_f32 %s, %s, %s, %s %s" % (dst, x, sp, dst, mods))
                n_fma += 1
                if (k % SEG == 0) and not in_top:
                    emit("v_pk_add_f32 %s, %s, %s" % (tot(r), tot(r), acc(r)))
    assert n_fma == R * T, (n_fma, R * T)
    name = "walk_asm_T%d_D%d_R%d_S%d" % (T, D, R, SEG)
    clobbers = ["v%d" % i for i in range(acc_base, vtop)]
    clobbers += ["s%d" % i for i in range(SGPR_RING_BASE, SGPR_RING_BASE + 16 * SGPR_RING_BLOCKS)]
    return name, lines, clobbers, dict(T=T, D=D, R=R, SEG=SEG, Q=Q, U=U, W=W, fma=n_fma, reads=len(issued),
                                       drains=len(drain_at), vgpr_top=vtop)


def emit_function(f, name, lines, clobbers, meta, suffix=""):
    R = meta["R"]
    f.write("// %s%s: T=%d D=%d R=%d SEG=%d  prefetch Q=%d units of %d sample(s), no LDS read in the %d step(s) ahead of a "
            "tap drain; %d v_pk_fma_f32, %d LDS reads, %d mid-walk tap drains\n" %
            (name, suffix, meta["T"], meta["D"], R, meta["SEG"], meta["Q"], meta["U"], meta["W"], meta["fma"], meta["reads"],
             meta["drains"]))
    f.write("__device__ __forceinline__ void %s%s(unsigned lds_lane_addr, const float *taps, f2 (&tot)[%d])\n{\n"
            % (name, suffix, R))
    f.write("    asm volatile(\n")
    for ln in lines:
        f.write('        "%s\\n\\t"\n' % ln)
    outs = ", ".join('"=&v"(tot[%d])' % r for r in range(R))
    f.write("        : %s\n" % outs)
    f.write('        : "v"(lds_lane_addr), "s"(taps)\n')
    f.write("        : %s);\n}\n\n" % ", ".join('"%s"' % c for c in ["memory"] + clobbers))


CONFIGS = [
    # (T, D, R, SEG, [(suffix, Q, U), ...])
    # only what the library launches: the decimating walks (the D = 1 wave-kernel walks of round 1 were reachable through
    # tuning variants only -- the D = 1 configurations run the compiler-scheduled workgroup kernel -- and were removed)
    # (round 4: the drain-shadow walks _b128_w2 / _w3 / _q4w2 and the no-drain timing study of round 3 measured nothing --
    # profiles/r03_direct_form_walks.txt -- and are no longer emitted; `--experiments` brings them back into a scratch header)
    (255, 4, 8, 32, [("", 4, 1), ("_b128", 2, 2)]),
    (127, 4, 8, 32, [("", 4, 1)]),
]


EXPERIMENTS = [
    # closed experiments of round 3 (never part of the library): drain shadow W = 2, 3, deeper prefetch, and the walk without
    # its tap drains (WRONG results by construction, a timing bound)
    (255, 4, 8, 32, [("_b128_w2", 3, 2, 2), ("_b128_w3", 3, 2, 3), ("_b128_q4w2", 4, 2, 2), ("_b128_nodrain", 2, 2, 1, True)]),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--experiments", action="store_true", help="also emit the closed round-3 experiment walks (use with --out)")
    args = ap.parse_args()
    if args.experiments:
        if os.path.abspath(args.out) == os.path.abspath(OUT):
            ap.error("--experiments needs --out <scratch header>: the library's header holds only what it launches")
        CONFIGS.extend(EXPERIMENTS)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("// GENERATED by tools/gen_walk.py — do not edit; regenerate with `python tools/gen_walk.py`.\n")
        f.write("// Hand-scheduled gfx950 assembly of the direct-form FIR walk (see the generator's docstring).\n")
        f.write("#pragma once\n\nnamespace if_fir\n{\n\n")
        for (T, D, R, SEG, variants) in CONFIGS:
            for variant in variants:
                suffix, Q, U = variant[:3]
                W = variant[3] if len(variant) > 3 else 1
                nodrain = variant[4] if len(variant) > 4 else False
                name, lines, clobbers, meta = gen_walk(T, D, R, SEG, Q, U, W, nodrain)
                emit_function(f, name, lines, clobbers, meta, suffix)
        f.write("} // namespace if_fir\n")
    print("wrote", os.path.normpath(args.out))


if __name__ == "__main__":
    main()
