#!/bin/bash
# (historical: the one-channel filter-bank route for decimation 8 / 16 / 32 / 64 compared here was removed later in round 3 --
# every multiple of 4 now runs behind the decimate-by-4 tail; kept as the record of how the profiles/ file was produced)
# r03_exp11.sh <tag> — GPU tests with decimation 32 / 64 through the decimate-by-16 tail; route on / off (3000) timing
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
python3 - > $O/route.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
n = 1 << 28
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
for t, d, nco in ((255, 32, 0.0), (1023, 64, 0.0), (255, 32, 0.1371)):
    with fir.IfFir(fir.bpf_design(t, 0.0, 0.01), d, 0, dev=True) as f:
        y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
        f.synth_device(x.data_ptr(), 0, n, 0)
        f.synchronize()
        if nco:
            f.set_nco(nco)
        row = []
        for v in (0, 3000, 0, 3000):
            f.set_tuning(v)
            f.reset()
            ms = [f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 10) for _ in range(4)][-1]
            row.append("%s %.4f ms" % ("bank route" if v == 0 else "selecting store", ms))
        print("%s%d taps /%d, 2^28: %s" % ("NCO + " if nco else "", t, d, " | ".join(row)), flush=True)
PY
cat $O/route.txt
