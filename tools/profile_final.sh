#!/bin/bash
# profile_final.sh <tag> — the round's closing measurements on ONE box (run through gpurun): GPU tests, the rocprofv3 passes
# behind profiles/<tag>_* for the three single-GPU BASELINE configs, the result table (bench_all), configs[3] on one GPU
# (8 channels), the AUTO policy sweep, the memory floor + sustained run (same_box), small calls.
cd "$(dirname "$0")/.."
R=$PWD
TAG=$1
O=gpurun_out/${TAG}_final
mkdir -p $O
step() { echo "== $(date +%T) $*"; }
step "GPU tests"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt | tail -3
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
for spec in "$TAG fir255_dec4_2p28" "${TAG}_fir127 fir127_2p26" "${TAG}_fir1023 fir1023_2p28"; do
  set -- $spec
  step "rocprofv3 passes $1 ($2)"
  timeout -k 10 400 bash tools/profile_round.sh $1 $2 || echo "profile_round $1 failed"
done
step "driver form of the bench line"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; tail -1 $O/bench_driver_form.json | cut -c1-260
step "result table"
timeout -k 10 900 bash tools/bench_all.sh ${TAG}_all
step "configs[3] on one GPU: 8 channels x 255 taps, /4 and D=1"
timeout -k 10 300 python3 bench.py --channels 8 --gpus 1 --no-cpu-baseline --no-extra-configs > $O/bench_8channels_1gpu_dec4.json 2> /dev/null; tail -1 $O/bench_8channels_1gpu_dec4.json | cut -c1-200
timeout -k 10 300 python3 bench.py --channels 8 --gpus 1 --workload fir255_2p28 --no-cpu-baseline --no-extra-configs > $O/bench_8channels_1gpu_d1.json 2> /dev/null; tail -1 $O/bench_8channels_1gpu_d1.json | cut -c1-200
step "AUTO policy sweep (2^27 samples)"
IF_FIR_DEBUG=1 timeout -k 10 600 python3 tools/policy_sweep.py 27 > $O/policy_sweep.txt 2>&1; tail -5 $O/policy_sweep.txt | cut -c1-200
step "memory floor and sustained run on this box"
timeout -k 10 600 bash tools/same_box.sh ${TAG}_same | tail -12
step "small calls"
IF_FIR_DEBUG=1 timeout -k 10 200 python3 tools/small_calls.py > $O/small_calls.txt 2>&1; tail -6 $O/small_calls.txt
step "filter bank"
for spec in "8 28 255 4" "8 28 255 8" "16 28 255 8" "8 28 255 8 tuning=1004096" "8 28 255 4 freq" "8 28 255 8 freq" "16 28 255 8 freq" "8 28 255 16 freq" "16 28 255 16 freq" "8 28 255 64 freq" "16 28 255 16"; do
  IF_FIR_DEBUG=1 timeout -k 10 300 python3 tools/fbank_bench.py $spec 2>&1 | tail -1 | cut -c1-1500 | tee -a $O/fbank.txt | cut -c1-300
done
step done
