#!/bin/bash
# pmc_i16.sh <tag> — counters of the int16-input instantiation of the headline kernel (fir_fft_kernel<4, true, true, false, 0>) beside the
# float32 one: SQ instruction / wait counters, LDS, VMEM, and HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes), each set in its
# own rocprofv3 --pmc pass, no trace domains (development tool; round 5)
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp IF_FIR_DEBUG=1
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for fmt in "--i16" ""; do
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
              "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf $O/p
    timeout -k 5 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/p -- python3 $R/tools/ab_inproc.py fir255_dec4_2p28 $fmt --rounds 2 --reps 20 > $O/run.log 2>&1 || { echo "pass failed: $ctrs"; tail -3 $O/run.log; continue; }
    echo "== input ${fmt:-float32}: $(grep median $O/run.log | cut -c1-110)"
    python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_fft_kernel<4, true, \(true\|false\), false, 0,"
  done
done
