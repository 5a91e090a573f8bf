#!/usr/bin/env python3
"""short_launch_rounds.py [taps=127] — what the END of a short launch costs (development tool; round 5).  BASELINE
configs[1] (127 taps, D = 1, 2^26 samples) is 17 477 blocks of 3840 samples on 2048 waves = 8.53 blocks per wave: waves that
finish their 8th block wait for those with a 9th.  Any finer tail (smaller last blocks, shared blocks) can at best remove that
wait.  This times launches of exactly 8.0, 8.25, 8.5, 8.53 (= 2^26 samples), 8.75 and 9.0 blocks per wave, alternately in one
process, and prints the time per block of each: the distance between the 2^26-sample launch and the whole-round launches is the
upper bound of what a finer tail could return."""
import os
import statistics
import sys

os.environ.setdefault("IF_FIR_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402


def main():
    taps_n = int(sys.argv[1]) if len(sys.argv) > 1 else 127
    fir = g.load_pkg().if_fir
    torch.cuda.set_device(0)
    taps = fir.bpf_design(taps_n)
    L, waves = 3840, 2048
    cases = [("8.00", 8 * waves), ("8.25", 8 * waves + waves // 4), ("8.50", 8 * waves + waves // 2), ("2^26", -(-(1 << 26) // L)),
             ("8.75", 8 * waves + 3 * waves // 4), ("9.00", 9 * waves)]
    nmax = max(nb for _, nb in cases) * L
    x = torch.empty(2 * nmax, dtype=torch.float32, device="cuda")
    y = torch.empty(2 * nmax, dtype=torch.float32, device="cuda")
    f = fir.IfFir(taps, 1, 0, dev=True)
    f.set_backend(fir.BACKEND_HIP_FFT)
    f.synth_device(x.data_ptr(), 0, nmax, 0)
    f.synchronize()
    sizes = [(name, nb, (1 << 26) if name == "2^26" else nb * L) for name, nb in cases]
    for _ in range(3):
        for _, _, n in sizes:
            f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 40)
    times = {name: [] for name, _, _ in sizes}
    for r in range(10):
        for name, nb, n in (sizes if r % 2 == 0 else sizes[::-1]):
            times[name].append(f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 40))
    base = None
    for name, nb, n in sizes:
        med = statistics.median(times[name])
        per = med * 1e3 / nb * waves          # microseconds of launch time per block and wave
        if name == "8.00":
            base = per
        print("%s taps D=1: %5s blocks per wave (%6d blocks, %9d samples): median %.4f ms (min %.4f max %.4f)  %.3f us per block-round  x%.4f  "
              "frac %.4f" % (taps_n, name, nb, n, med, min(times[name]), max(times[name]), per, per / base, 16.0 * n / (med * 1e-3) / 8e12), flush=True)


if __name__ == "__main__":
    main()
