#!/bin/bash
# r03_exp2.sh <tag> — round-3 batch 2 on ONE box (development tool): direct-form walk variants (tap-drain shadow), the
# rocprofv3 passes behind profiles/r03a_* (kernel trace + FETCH_SIZE + WRITE_SIZE) for the three single-GPU BASELINE
# configs, and the combined TA/TCP counter set that round 2 saw abort the profiler (command and message kept).
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
echo "== direct form, walk variants (0 = round 2's schedule, 7/8/9 = no LDS read in the 2/3/2 steps ahead of a tap drain, Q = 3/3/4)"
timeout -k 10 300 python3 tools/sweep.py fir255_dec4_2p28 0 7 0 8 0 9 0 7 0 8 0 9 0 > $O/direct_walks.txt 2>&1
cut -c1-140 $O/direct_walks.txt | grep variant
for spec in "r03a fir255_dec4_2p28" "r03a_fir127 fir127_2p26" "r03a_fir1023 fir1023_2p28"; do
  set -- $spec
  echo "== rocprofv3 passes $1 ($2)"
  timeout -k 10 400 bash tools/profile_round.sh $1 $2 || echo "profile_round $1 failed"
done
echo "== combined TA/TCP derived counters in ONE pass (round 2: 'combined sets made the profiler abort on this pool')"
cd /tmp
export TMPDIR=/tmp
CMD="rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/$O/pmc_combined -- python3 $R/tools/sweep.py fir255_dec4_2p24 100"
echo "\$ $CMD" > $R/$O/pmc_combined_attempt.txt
timeout -k 10 120 $CMD >> $R/$O/pmc_combined_attempt.txt 2>&1
echo "exit code: $?" >> $R/$O/pmc_combined_attempt.txt
tail -15 $R/$O/pmc_combined_attempt.txt
