#!/bin/bash
# r05_exp12.sh <tag> — round 5, batch 12 on ONE box (development tool): the library with single LDS reads (decimate-by-2 tails in paired units of
# their own, the X exchange's own region stride) against the round's committed kernels (libif_fir_ab_base.so): every family, the LDS counters, the GPU tests.
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1 TMPDIR=/tmp
LIBS="$A/libif_fir_ab_base.so $A/libif_fir_dev.so"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt || exit 1
for w in fir255_dec4_2p28 fir255_2p28 fir127_2p26 fir1023_2p28 fir255_dec3_2p28 fir1023_dec8_2p28 fir255_dec2_2p28 fir2047_dec8_2p26 fir255_dec9_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --nco 0.01 --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 600 python3 tools/fbank_ab.py --libs $LIBS --cases 4:8:freq 8:8:freq 16:8:freq 64:8:freq 4:8:slots 8:8:slots 16:16:slots 8:16:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
cd /tmp
for w in fir255_dec4_2p28 fir255_2p28; do
  for lib in ab_base dev; do
    rm -rf $O/p
    timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p -- python3 $R/tools/ab_inproc.py $w --libs $R/$A/libif_fir_$lib.so --rounds 2 --reps 20 > $O/run.log 2>&1 || { echo "pass failed: $w $lib"; tail -3 $O/run.log; continue; }
    echo "== $w $lib: $(grep median $O/run.log | cut -c1-110)"
    python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_fft_kernel<4, \(true\|false\), false, false, 0,"
  done
done > $O/pmc_lds.txt 2>&1
grep -c mean $O/pmc_lds.txt
