#!/bin/bash
# r03_exp9.sh <tag> — GPU tests with the NCO on the decimate-by-8/16 route and the bank's common offset; tuned + decimated timing
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
python3 - > $O/route_nco.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
n = 1 << 28
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
xi = None
for t, d, i16 in ((255, 8, False), (255, 16, False), (1023, 16, False), (255, 16, True)):
    with fir.IfFir(fir.bpf_design(t, 0.0, 0.02), d, 0, dev=True) as f:
        y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
        f.synth_device(x.data_ptr(), 0, n, 0)
        f.synchronize()
        src = x
        if i16:
            f.set_input_format(fir.INPUT_I16)
            xi = torch.clamp(torch.round(x * 16384.0), -32768, 32767).to(torch.int16)
            torch.cuda.synchronize()
            src = xi
        f.set_nco(0.1371)
        row = []
        for v in (0, 3000, 0, 3000):
            f.set_tuning(v)
            f.reset()
            ms = [f.time_device(src.data_ptr(), y.data_ptr(), n, 0, 10) for _ in range(4)][-1]
            row.append("%s %.4f ms" % ("bank route" if v == 0 else "selecting store", ms))
        print("NCO + %d taps /%d%s, 2^28: %s" % (t, d, " int16 input" if i16 else "", " | ".join(row)), flush=True)
PY
cat $O/route_nco.txt
