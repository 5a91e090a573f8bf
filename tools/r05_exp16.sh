#!/bin/bash
# r05_exp16.sh <tag> — round 5, batch 16 (development tool): LDS counters of the kernels whose access patterns were laid out under the 32-bank rule of the paired
# reads, now that every read is a single ds_read_b64 (64 banks, lanes 0-31 / 32-63 per cycle): the odd-decimation kernel, the bank's slot forms (tails 16, 9, 8,
# 4) and general forms (17, 8 with NCO, 5), the decimate-by-2 tail.
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp IF_FIR_DEBUG=1
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
CTRS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for w in fir255_dec3_2p28 fir255_dec2_2p28 fir1023_dec8_2p28; do
  rm -rf $O/p
  timeout -k 5 200 rocprofv3 --pmc $CTRS --output-format csv -d $O/p -- python3 $R/tools/ab_inproc.py $w --rounds 2 --reps 20 > $O/run.log 2>&1 || { echo "pass failed: $w"; tail -3 $O/run.log; continue; }
  echo "== $w: $(grep median $O/run.log | cut -c1-110)"
  python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_fft_kernel\|fir_odd_kernel" | grep -v "direct_copy" | head -12
done > $O/pmc_lds_other.txt 2>&1
for spec in "8 28 255 16 freq" "8 28 255 8 freq" "8 28 255 4 freq" "8 28 255 4" "8 28 255 8" "16 28 255 16"; do
  rm -rf $O/p
  timeout -k 5 200 rocprofv3 --pmc $CTRS --output-format csv -d $O/p -- python3 $R/tools/fbank_bench.py $spec > $O/run.log 2>&1 || { echo "pass failed: $spec"; tail -3 $O/run.log; continue; }
  echo "== bank $spec"
  python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_fft_kernel<4, true, false, \(true\|false\), \(4\|5\|8\|9\|16\|17\)," | head -22
done >> $O/pmc_lds_other.txt 2>&1
grep -c mean $O/pmc_lds_other.txt
