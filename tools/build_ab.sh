#!/bin/bash
# build_ab.sh <name> <extra hipcc flags...> — alternative build of libif_fir.so with other flags for if_fir_fft.hip
# (a DEVELOPMENT build: it replaces libif_fir_dev.so, which the tools load; A/B timing through tools/sweep_libs.sh; development tool).  Output: qo-100-tools_amd/libif_fir_ab_<name>.so
set -e
cd "$(dirname "$0")/../qo-100-tools_amd/csrc"
NAME=$1; shift
R=$(cd ../.. && pwd)
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function -I$R/include -I$R/qo-100-tools_amd/csrc"
# the overlap-save kernel's six units (host side + one per overlap length), in parallel
/opt/rocm/bin/hipcc "$@" $FLAGS -c if_fir_fft.hip -o /tmp/if_fir_fft_ab_$NAME.o &
for r in 4 8 16 32 48; do
  /opt/rocm/bin/hipcc "$@" $FLAGS -DIF_FIR_FFT_ROWS=$r -c if_fir_fft.hip -o /tmp/if_fir_fft_ab_${NAME}_r$r.o &
done
/opt/rocm/bin/hipcc "$@" $FLAGS -DIF_FIR_FFT_ODD=1 -c if_fir_fft.hip -o /tmp/if_fir_fft_ab_${NAME}_odd.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libif_fir_ab_$NAME.so if_fir_kernels.o /tmp/if_fir_fft_ab_$NAME.o \
  /tmp/if_fir_fft_ab_${NAME}_r4.o /tmp/if_fir_fft_ab_${NAME}_r8.o /tmp/if_fir_fft_ab_${NAME}_r16.o /tmp/if_fir_fft_ab_${NAME}_r32.o /tmp/if_fir_fft_ab_${NAME}_r48.o /tmp/if_fir_fft_ab_${NAME}_odd.o \
  if_fir_shim_dev.o if_fir_mc_dev.o if_bpf_design.o wb_detect.o rc_reg.o -lm -ldl
echo built libif_fir_ab_$NAME.so
