import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g
fir = g.load_pkg().if_fir
oracle = g.load_oracle()
taps = fir.bpf_design(255)
for n, cuts in ((400_007, [0, 215_041, 400_007]), (430_080, [0, 215_040, 430_080]), (100_001, [0, 100_001]), (100_002, [0, 100_002])):
    x = oracle.synth_iq(n, 3)
    ref = oracle.fir_f64(taps, x, 2)
    with fir.IfFir(taps, 2, n) as f:
        y1 = f.process(x)
        f.reset()
        y2 = np.concatenate([f.process(x[2*a:2*b]) for a, b in zip(cuts[:-1], cuts[1:])])
    l2, mx = oracle.err_metrics(y1, ref)
    d = np.nonzero(y1 != y2)[0]
    e = np.abs(y1 - ref.astype(np.float32))
    print("n", n, "one call vs oracle", l2, mx, "| pieces differ at", d[:6], len(d), "| worst abs err index", int(np.argmax(e)), float(e.max()), "of", y1.size, flush=True)
