#!/bin/bash
# collect_round.sh <tag> — copies the summaries of a tools/profile_final.sh <tag> batch from gpurun_out/ into profiles/ (tracked):
# kernel-trace summaries and stats + traffic.json entries of the three BASELINE configs, the bench lines, the result table, the policy
# sweep, the same-box memory floors, small calls, the filter-bank lines.
cd "$(dirname "$0")/.."
T=$1
G=gpurun_out
python3 tools/collect_profiles.py $T fir255_dec4_2p28 $G/$T/kt $G/$T/pf $G/$T/pw
python3 tools/collect_profiles.py ${T}_fir127 fir127_2p26 $G/${T}_fir127/kt $G/${T}_fir127/pf $G/${T}_fir127/pw "fir_fft_kernel<4, false, false, false, 0, false, false>"
python3 tools/collect_profiles.py ${T}_fir1023 fir1023_2p28 $G/${T}_fir1023/kt $G/${T}_fir1023/pf $G/${T}_fir1023/pw "fir_fft_kernel<16, false, false, false, 0, false, false>"
cp $G/$T/bench.json profiles/${T}_bench.json
cp $G/${T}_fir127/bench.json profiles/${T}_bench_fir127.json
cp $G/${T}_fir1023/bench.json profiles/${T}_bench_fir1023.json
cp $G/${T}_final/bench_driver_form.json profiles/${T}_bench_driver_form.json
cp $G/${T}_final/bench_8channels_1gpu_dec4.json profiles/${T}_bench_8channels_1gpu_dec4.json
cp $G/${T}_final/bench_8channels_1gpu_d1.json profiles/${T}_bench_8channels_1gpu_d1.json
cp $G/${T}_all/bench_all.txt profiles/${T}_bench_all.txt
cp $G/${T}_final/policy_sweep.txt profiles/${T}_policy_sweep.txt
cp $G/${T}_final/small_calls.txt profiles/${T}_small_calls.txt
cp $G/${T}_final/fbank.txt profiles/${T}_filter_bank.txt
grep -A40 "memory floor and sustained" $G/${T}_final.log | grep -B40 "small calls" | grep -v "^==" > profiles/${T}_same_box.txt
ls profiles | grep "^${T}_" | wc -l
