#!/usr/bin/env python3
"""sweep.py — times every tuning variant of one workload through the C-ABI (development tool).
usage: python tools/sweep.py [workload] [variants...]"""
import os, sys
os.environ.setdefault("IF_FIR_DEBUG", "1")   # development tool: diagnostic tuning variants allowed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
import bench

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "fir255_dec4_2p28"
    variants = [int(v) for v in sys.argv[2:]] or list(range(0, 7))   # variant 100 = overlap-save FFT backend
    taps_n, decim, log2n, _ = bench.WORKLOADS[wl]
    n = 1 << log2n
    fir = g.load_pkg().if_fir
    taps = fir.bpf_design(taps_n)
    torch.cuda.set_device(0)
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    ref = None
    with fir.IfFir(taps, decim, 0, dev=True) as f:
        m = f.out_count(n)
        y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 0)
        f.synchronize()
        for v in variants:
            if v == 200:
                f.set_backend(fir.BACKEND_HIP_TAPSPLIT)
            elif v == 300:
                f.set_backend(fir.BACKEND_HIP_GENERIC)
            elif v >= 100:
                f.set_backend(fir.BACKEND_HIP_FFT)
                f.set_tuning(v if v >= 1000 else 0)   # 1001/1002/1003: FFT diagnostics (skip loads / stores / both)
            else:
                f.set_backend(fir.BACKEND_HIP_DIRECT)
                f.set_tuning(v)
            f.reset()
            y.zero_()
            torch.cuda.synchronize()
            f.process_device(x.data_ptr(), y.data_ptr(), n)
            f.synchronize()
            ck = (y.double().sum().item(), y.double().abs().sum().item())
            if ref is None:
                ref = y.clone()
            same = bool(torch.equal(ref, y))
            # sustained rate: the chip throttles after the first few launches (power management), so time 4 x 10
            # back-to-back launches and keep the last batch
            ms = [f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 10) for _ in range(4)][-1]
            gbs = bench.algorithmic_bytes_per_sample(decim) * n / (ms * 1e-3) / 1e9
            tf = bench.algorithmic_flops_per_sample(taps_n, decim) * n / (ms * 1e-3) / 1e12
            print("%s variant %d: %.4f ms  %.1f GS/s  %.0f GB/s (%.1f%% HBM)  %.1f TF (%.1f%% VALU)  same_as_v%d=%s ck=%.6e" %
                  (wl, v, ms, n / ms / 1e6, gbs, gbs / 80.0, tf, tf / 1.573, variants[0], same, ck[1]), flush=True)

if __name__ == "__main__":
    main()
