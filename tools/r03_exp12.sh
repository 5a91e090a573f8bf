#!/bin/bash
# r03_exp12.sh <tag> — cache-policy bits of the overlap-save kernel's row loads / output stores in the power-limited regime
# (aux: 2 = nt, 1 = sc0, 16 = sc1): A = alternative build, B = the library (loads nt with cached edge rows, stores nt)
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
for lib in aux_l3s2 aux_l18s2 aux_l2s18 aux_l18s18 aux_l0s2 aux_l2s3; do
  echo "== A = libif_fir_ab_$lib.so (load aux / store aux in the name), B = base"
  VARIANTS="100 100 100" bash tools/ab_interleaved.sh qo-100-tools_amd/libif_fir_ab_$lib.so 2 fir255_dec4_2p28 2>&1 | tee -a $O/ab_$lib.txt
done
