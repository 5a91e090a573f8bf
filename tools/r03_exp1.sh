#!/bin/bash
# r03_exp1.sh <tag> — round-3 experiment batch 1 on ONE box (development tool): the GPU test suite, then the overlap-save
# kernel's head/tail/prefetch variants (runtime diag bits of the current library, interleaved with the default launch)
# and alternative builds (libif_fir_ab_*.so), on the three single-GPU BASELINE workloads.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
D=1000000
# runtime variants: 8 = prefetch of the next group by the slot-0 taker, 128 = per-taker prefetch of its slot's block,
# 4 + (n << 12) = second wave of every SIMD starts n sleeps (~1 us each) later
V="100 $((D+8)) 100 $((D+128)) 100 $((D+4+(3<<12))) 100 $((D+4+(6<<12))) 100 $((D+8)) 100 $((D+128)) 100 $((D+4+(3<<12))) 100"
for w in fir255_dec4_2p28 fir127_2p26 fir1023_2p28; do
  timeout -k 10 300 python3 tools/sweep.py $w $V > $O/diag_$w.txt 2>&1
  cut -c1-150 $O/diag_$w.txt | grep variant
done
# wave run times / launch span / clock for the three workloads (tail share)
for w in fir255_dec4_2p28 fir127_2p26 fir1023_2p28; do
  timeout -k 10 300 python3 tools/fft_clock.py $w 0 $((D+8)) 0 $((D+4+(3<<12))) > $O/clock_$w.txt 2>&1
  cat $O/clock_$w.txt | grep variant
done
# alternative builds, interleaved with the base library
for lib in lf0 eg0 contig; do
  echo "== A = libif_fir_ab_$lib.so, B = libif_fir.so"
  VARIANTS="100 100 100" bash tools/ab_interleaved.sh qo-100-tools_amd/libif_fir_ab_$lib.so 2 fir255_dec4_2p28 fir127_2p26 2>&1 | tee -a $O/ab_$lib.txt
done
