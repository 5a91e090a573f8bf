#!/bin/bash
cd "$(dirname "$0")/.."
echo "== base (b128 stores)"; python3 tools/diag_dec2.py 2>&1 | grep "^n "
cp qo-100-tools_amd/libif_fir.so /tmp/base.so
# the diagnostic uses the product library: swap it for the A/B build's kernels (dev build exports a superset)
cp qo-100-tools_amd/libif_fir_ab_dec2b64.so qo-100-tools_amd/libif_fir.so
echo "== b64 stores"; python3 tools/diag_dec2.py 2>&1 | grep "^n "
cp /tmp/base.so qo-100-tools_amd/libif_fir.so
