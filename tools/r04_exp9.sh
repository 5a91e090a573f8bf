#!/bin/bash
# r04_exp9.sh <tag> — round 4, batch 9 on ONE box (development tool): the odd-decimation kernel without its own load transposition (the
# columns are rotated in registers and delivered into the first forward transposition) against the form with it
# (libif_fir_ab_oddold.so): its GPU tests, then in-process A/B; the filter bank's general form timed in full.
# (needs qo-100-tools_amd/libif_fir_ab_oddold.so: `tools/build_ab.sh oddold` on a tree at commit cd7127b, the last one with the kernel's own
# transposition phase; the A/B libraries are not kept in the repository)
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "odd_decimation or any_decimation or chunked_equals" > $O/gpu_tests_odd.txt 2>&1; rc=$?; echo "odd tests rc=$rc"; tail -6 $O/gpu_tests_odd.txt | cut -c1-300
[ $rc -eq 0 ] || exit 1
A=qo-100-tools_amd
for w in fir255_dec3_2p28 fir511_dec3_2p28 fir255_dec9_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_ab_oddold.so $A/libif_fir_dev.so --rounds 8 --reps 30 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_odd_merge.txt
done
timeout -k 10 200 python3 tools/ab_inproc.py fir255_dec3_2p28 --i16 --libs $A/libif_fir_ab_oddold.so $A/libif_fir_dev.so --rounds 6 --reps 30 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_odd_merge.txt
for spec in "8 28 255 8" "8 28 255 8 freq" "16 28 255 8 freq"; do timeout -k 10 200 python3 tools/fbank_bench.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1500 | tee -a $O/fbank.txt | cut -c1-200; done
