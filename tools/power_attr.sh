#!/bin/bash
# power_attr.sh <tag> — where the package power of the headline launch goes: rocm-smi sampled while (a) the memory skeleton of the
# kernel runs alone (tools/ubench_mem2 soak), (b) the one-shot 4:1 mix streams, (c) diagnostic launches of the kernel run (loads
# and/or stores skipped; timing-study builds without the LDS twiddle reads / without exchange 1), (d) the kernel itself.
# energy per launch = (W - idle W) x ms.  Development tool (DESIGN §3.4 finding 8).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
cp qo-100-tools_amd/libif_fir_dev.so /tmp/libif_fir_base.so
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo; }
sample() { # <count> <file>
  for i in $(seq 1 $1); do smi >> $2; sleep 0.5; done
}
report() { # <label> <file> <result line>
  python3 - "$1" "$2" "$3" <<'PY' | tee -a $O/power_attr.txt
import re, sys
label, path, res = sys.argv[1:4]
w, clk = [], []
for line in open(path):
    m = re.search(r"\((\d+)Mhz\).*?([\d.]+)\s*$", line)
    if m:
        clk.append(int(m.group(1))); w.append(float(m.group(2)))
w, clk = w[2:], clk[2:]
print("%-46s %6.0f W  sclk %4d-%4d MHz  | %s" % (label, sum(w) / max(1, len(w)), min(clk or [0]), max(clk or [0]), res))
PY
}
run_variant() { # <label> <lib> <variant>
  cp "$2" qo-100-tools_amd/libif_fir_dev.so
  python3 - $3 > $O/run.txt 2>&1 <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
n = 1 << 28
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(fir.bpf_design(255), 4, 0, dev=True) as f:
    f.synth_device(x.data_ptr(), 0, n, 0)
    y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
    f.set_tuning(int(sys.argv[1]))
    t0 = time.time()
    while time.time() - t0 < 9:
        ms = f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 200)
    print("%.4f ms per launch" % ms)
PY
  P=$!
  sleep 5
  rm -f $O/s.txt; sample 7 $O/s.txt
  wait $P
  report "$1" $O/s.txt "$(tail -1 $O/run.txt)"
  cp /tmp/libif_fir_base.so qo-100-tools_amd/libif_fir_dev.so
}
: > $O/power_attr.txt
rm -f $O/s.txt; sample 5 $O/s.txt; report "idle" $O/s.txt "-"
./tools/ubench_mem2 soak 9 > $O/run.txt 2>&1 & P=$!; sleep 4; rm -f $O/s.txt; sample 8 $O/s.txt; wait $P
report "memory skeleton alone (persistent, nt)" $O/s.txt "$(tail -1 $O/run.txt)"
./tools/ubench_mem2 soak 9 mix > $O/run.txt 2>&1 & P=$!; sleep 4; rm -f $O/s.txt; sample 8 $O/s.txt; wait $P
report "one-shot 4:1 mix alone" $O/s.txt "$(tail -1 $O/run.txt)"
B=/tmp/libif_fir_base.so
run_variant "kernel" $B 100
run_variant "kernel, stores skipped" $B 1002
run_variant "kernel, loads skipped" $B 1001
run_variant "kernel, loads+stores skipped (compute)" $B 1003
run_variant "compute, twiddles from registers (study)" qo-100-tools_amd/libif_fir_ab_notw.so 1003
run_variant "compute, no exchange 1 (study)" qo-100-tools_amd/libif_fir_ab_nox1.so 1003
run_variant "kernel, twiddles from registers (study)" qo-100-tools_amd/libif_fir_ab_notw.so 100
run_variant "kernel, no exchange 1 (study)" qo-100-tools_amd/libif_fir_ab_nox1.so 100
run_variant "kernel (again)" $B 100
