// if_fir_kernels.h — internal interface between the C-ABI shim and the HIP kernels (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace if_fir
{

enum
{
    BACKEND_AUTO = 0,
    BACKEND_DIRECT = 1,
    BACKEND_TAPSPLIT = 2,
    BACKEND_GENERIC = 3,
    BACKEND_FFT = 4
};

struct LaunchArgs
{
    const void *in;    // device, interleaved float32 I/Q, N samples
    void *out;         // device, M samples
    const float *taps; // device, T floats
    const void *hist;  // device, T-1 samples (most recent last)
    int T, D;
    int in_i16; // 1: input samples are interleaved int16 I,Q (value = int16 * 2^-15); FFT and generic backends only
    int ctaps;  // 1: taps are complex (interleaved re,im), FFT and generic backends only
    int64_t N;  // input samples of this call
    int32_t n0; // offset of the first output sample inside this call's input (0 ≤ n0 < D)
    int64_t M;  // outputs of this call
    int backend;
    int device;
    hipStream_t stream;
    const void *fft_tables; // device, FFT_TABLE_FLOATS floats (overlap-save backend) or nullptr
    void *queue; // device, 16 bytes: atomic run queue of the persistent kernel (zeroed by the launcher)
    int diag;  // development diagnostics for the FFT kernel (0 in production)
    int grid_limit; // FFT backend: at most this many workgroups (0 = one per CU); same results, used by the queue tests
    void *dbg; // optional diagnostic stamp buffer (8192 x 4 x u64) or nullptr
};

bool direct_supported(int T, int D);
hipError_t launch_fir(const LaunchArgs &a, int variant);
// overlap-save FFT backend (if_fir_fft.hip)
constexpr int FFT_TABLE_FLOATS = 2 * (4096 + 4096 + 256 + 1024 + 1024);
bool fft_supported(int T, int D);
hipError_t launch_fft(const LaunchArgs &a);
void fft_build_tables(const float *taps, int T, int ctaps, int D, float *tables);

hipError_t launch_history(const void *in, const void *hist_in, void *hist_out, int T, int64_t N, int in_i16,
                          hipStream_t stream);
hipError_t launch_synth(void *iq, uint64_t first, uint64_t count, uint32_t channel, const float *tone10,
                        hipStream_t stream);

} // namespace if_fir
