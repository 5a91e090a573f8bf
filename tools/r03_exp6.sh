#!/bin/bash
# r03_exp6.sh <tag> — direct-form kernel: the generated walk with (0) and without (10, timing study) its 12 tap-block drains
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 300 python3 tools/sweep.py fir255_dec4_2p28 0 10 0 10 0 10 0 10 0 > $O/direct_nodrain.txt 2>&1
cut -c1-160 $O/direct_nodrain.txt | grep variant
