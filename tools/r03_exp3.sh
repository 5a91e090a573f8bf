#!/bin/bash
# r03_exp3.sh <tag> — round-3 batch 3 on ONE box: GPU tests with the single-word block queue and the 16-slot filter bank,
# headline / configs timing with the new queue, filter-bank benchmark at decimation 4 and 16.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
for w in fir255_dec4_2p28 fir127_2p26 fir1023_2p28; do
  timeout -k 10 300 python3 tools/sweep.py $w 100 100 100 > $O/sweep_$w.txt 2>&1; cut -c1-150 $O/sweep_$w.txt | grep variant
  timeout -k 10 300 python3 tools/fft_clock.py $w 0 0 > $O/clock_$w.txt 2>&1; grep variant $O/clock_$w.txt
done
for spec in "8 28 255 4" "8 28 255 16" "16 28 255 16" "16 28 1023 16"; do
  timeout -k 10 300 python3 tools/fbank_bench.py $spec 2>&1 | tail -1 | tee -a $O/fbank.txt
done
