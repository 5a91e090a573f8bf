#!/bin/bash
# r05_exp18.sh <tag> — round 5, batch 18 (development tool): the odd-decimation kernel's forward Y^-1 on its own region stride (2240 bytes: its single reads
# interleave on the 64 banks; libif_fir_dev.so) against the inverse's 2208 (libif_fir_ab_y2208.so); LDS counters of both; the odd-decimation GPU tests.
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1 TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "odd or any_decimation or random_configurations" 2>&1 | tail -2 | tee $O/pytest.txt
LIBS="$A/libif_fir_ab_y2208.so $A/libif_fir_dev.so"
for w in fir255_dec3_2p28 fir255_dec9_2p28 fir511_dec3_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec3_2p28 --i16 --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
cd /tmp
for lib in ab_y2208 dev; do
  rm -rf $O/p
  timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p -- python3 $R/tools/ab_inproc.py fir255_dec3_2p28 --libs $R/$A/libif_fir_$lib.so --rounds 2 --reps 20 > $O/run.log 2>&1 || { echo "pass failed: $lib"; tail -3 $O/run.log; continue; }
  echo "== $lib: $(grep median $O/run.log | cut -c1-110)"
  python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_odd_kernel"
done > $O/pmc_lds.txt 2>&1
