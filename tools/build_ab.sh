#!/bin/bash
# build_ab.sh <name> <extra hipcc flags...> — alternative build of libif_fir.so with other flags for if_fir_fft.hip
# (a DEVELOPMENT build: it replaces libif_fir_dev.so, which the tools load; A/B timing through tools/sweep_libs.sh; development tool).  Output: qo-100-tools_amd/libif_fir_ab_<name>.so
# The units are those of csrc/Makefile: host side, one per overlap length, one per overlap length for the decimate-by-2 tails
# (-DIF_FIR_FFT_DEC2_UNIT), the odd-decimation kernel; the device units with the IR load-store vectorizer off (NOPAIR; NOPAIR= in the environment
# builds them with the vectorizer on).  D2FLAGS="..." adds flags to the decimate-by-2 units only; REUSE_MAIN=1 links the other units' objects of the regular build
# (csrc/*.o) and compiles the decimate-by-2 units only.
set -e
cd "$(dirname "$0")/../qo-100-tools_amd/csrc"
NAME=$1; shift
R=$(cd ../.. && pwd)
NOPAIR=${NOPAIR--mllvm -amdgpu-load-store-vectorizer=0}
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function -I$R/include -I$R/qo-100-tools_amd/csrc"
OBJS=""
if [ -n "$REUSE_MAIN" ]; then
  HOSTO=if_fir_fft.o; ODDO=if_fir_fft_odd.o
  for r in 4 8 16 32 48; do OBJS="$OBJS if_fir_fft_r$r.o /tmp/if_fir_fft_ab_${NAME}_d2_r$r.o"; done
else
  HOSTO=/tmp/if_fir_fft_ab_$NAME.o; ODDO=/tmp/if_fir_fft_ab_${NAME}_odd.o
  /opt/rocm/bin/hipcc "$@" $FLAGS -c if_fir_fft.hip -o $HOSTO &
  for r in 4 8 16 32 48; do
    /opt/rocm/bin/hipcc "$@" $FLAGS $NOPAIR -DIF_FIR_FFT_ROWS=$r -c if_fir_fft.hip -o /tmp/if_fir_fft_ab_${NAME}_r$r.o &
    OBJS="$OBJS /tmp/if_fir_fft_ab_${NAME}_r$r.o /tmp/if_fir_fft_ab_${NAME}_d2_r$r.o"
  done
  /opt/rocm/bin/hipcc "$@" $FLAGS $NOPAIR -DIF_FIR_FFT_ODD=1 -c if_fir_fft.hip -o $ODDO &
  wait
fi
for r in 4 8 16 32 48; do
  /opt/rocm/bin/hipcc "$@" $FLAGS $NOPAIR $D2FLAGS -DIF_FIR_FFT_ROWS=$r -DIF_FIR_FFT_DEC2_UNIT=1 -c if_fir_fft.hip -o /tmp/if_fir_fft_ab_${NAME}_d2_r$r.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libif_fir_ab_$NAME.so if_fir_kernels.o $HOSTO $OBJS $ODDO \
  if_fir_shim_dev.o if_fir_mc_dev.o if_bpf_design.o wb_detect.o rc_reg.o -lm -ldl
echo built libif_fir_ab_$NAME.so
