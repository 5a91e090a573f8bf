#!/bin/bash
# times the FFT backend with alternative builds of libif_fir.so (development tool)
cd "$(dirname "$0")/.."
cp qo-100-tools_amd/libif_fir_dev.so /tmp/libif_fir_orig.so
for lib in qo-100-tools_amd/libif_fir_ab_*.so; do
  cp "$lib" qo-100-tools_amd/libif_fir_dev.so
  echo "== $lib"
  for w in "$@"; do python tools/sweep.py $w ${VARIANTS:-100} 2>/dev/null | grep variant; done
done
cp /tmp/libif_fir_orig.so qo-100-tools_amd/libif_fir_dev.so
echo "== baseline"
for w in "$@"; do python tools/sweep.py $w ${VARIANTS:-100} 2>/dev/null | grep variant; done
