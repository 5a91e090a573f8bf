#!/bin/bash
# build_ab.sh <name> <extra hipcc flags...> — alternative build of libif_fir.so with other flags for if_fir_fft.hip
# (a DEVELOPMENT build: it replaces libif_fir_dev.so, which the tools load; A/B timing through tools/sweep_libs.sh; development tool).  Output: qo-100-tools_amd/libif_fir_ab_<name>.so
set -e
cd "$(dirname "$0")/../qo-100-tools_amd/csrc"
NAME=$1; shift
R=$(cd ../.. && pwd)
/opt/rocm/bin/hipcc "$@" -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function \
  -I$R/include -I$R/qo-100-tools_amd/csrc -c if_fir_fft.hip -o /tmp/if_fir_fft_ab_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libif_fir_ab_$NAME.so if_fir_kernels.o /tmp/if_fir_fft_ab_$NAME.o \
  if_fir_shim_dev.o if_fir_mc_dev.o if_bpf_design.o wb_detect.o rc_reg.o -lm -ldl
echo built libif_fir_ab_$NAME.so
