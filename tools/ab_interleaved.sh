#!/bin/bash
# ab_interleaved.sh <lib_a.so> <rounds> <workloads...> — alternates an A/B build with the current libif_fir.so, several
# rounds, so that box drift (clock state, temperature) averages out (development tool)
cd "$(dirname "$0")/.."
A=$1; R=$2; shift 2
cp qo-100-tools_amd/libif_fir_dev.so /tmp/libif_fir_base.so
for r in $(seq 1 $R); do
  for w in "$@"; do
    cp "$A" qo-100-tools_amd/libif_fir_dev.so
    echo "A $(python tools/sweep.py $w ${VARIANTS:-100 100 100} 2>/dev/null | grep variant | tail -1 | cut -c1-60)"
    cp /tmp/libif_fir_base.so qo-100-tools_amd/libif_fir_dev.so
    echo "B $(python tools/sweep.py $w ${VARIANTS:-100 100 100} 2>/dev/null | grep variant | tail -1 | cut -c1-60)"
  done
done
cp /tmp/libif_fir_base.so qo-100-tools_amd/libif_fir_dev.so
