#!/bin/bash
# r03_exp2.sh <tag> — round-3 batch 2 on ONE box (development tool): the
# rocprofv3 passes behind profiles/r03a_* (kernel trace + FETCH_SIZE + WRITE_SIZE) for the three single-GPU BASELINE
# configs.
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
# (round 4: the direct-form walk variants 7/8/9 this script swept -- drain shadow, profiles/r03_direct_form_walks.txt -- were
# closed experiments and are no longer generated: `python3 tools/gen_walk.py --experiments --out <scratch>` re-creates them)
for spec in "r03a fir255_dec4_2p28" "r03a_fir127 fir127_2p26" "r03a_fir1023 fir1023_2p28"; do
  set -- $spec
  echo "== rocprofv3 passes $1 ($2)"
  timeout -k 10 400 bash tools/profile_round.sh $1 $2 || echo "profile_round $1 failed"
done
# (round 4, ADVICE r3: the step that re-ran the combined TA/TCP counter set was removed -- it makes rocprofv3 abort in its
# counter-set validation, error 38, before any launch, and then hang until the timeout; the record is
# profiles/r03_pmc_combined_abort.txt, and tools/profile_pmc.sh collects those counters one per pass.)
