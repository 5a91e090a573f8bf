// if_fir_kernels.h — internal interface between the C-ABI shim and the HIP kernels (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

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

// uniform filter bank on the overlap-save kernel: channel c sits at slot[c]/16 cycles/sample
constexpr int CHAN_MAX = 16;
struct ChanArgs
{
    uint32_t count;
    uint32_t slot[CHAN_MAX];
    float tw[CHAN_MAX][30];  // W4096^(a bin), a = 1..15 (re, im); slots: bin = 256 slot, i.e. W16^(a slot).  Decimation 4 uses a = 1..3, 8: 1..7, 16 (per channel): all
    float rot0[CHAN_MAX][2]; // exp(-j 2 pi slot (abs0 + n0) / 16): mix-down phase at this call's first output
    float2 *out[CHAN_MAX];   // device, M samples each
    uint32_t mask16;         // decimation 16: bit s set = slot s is wanted (out[s] non-null)
    uint32_t rot_e;          // decimation 16: (abs0 + n0) mod 16: slot s is rotated by W16^(s rot_e) at this call's first output
    uint32_t sub;            // decimate-by-4 tail at decimation 4 * sub (single channel): keep every sub-th output (set by the launcher)
    // decimation 8 and 16 (round 4): every channel has its own centre on the fs/4096 grid and its own mix-down frequency
    uint32_t bin[CHAN_MAX];   // centre of the channel's filter: the prototype moved up by bin / 4096 cycles/sample (0..4095)
    uint32_t pword[CHAN_MAX]; // mix-down frequency as a 32-bit phase word (cycles/sample x 2^32), the context's NCO included:
                              // output o of a call is rotated by exp(-j 2 pi pword (abs0n0 + D o) / 2^32)
    uint32_t abs0n0;          // absolute index (mod 2^32) of the input sample this call's first output belongs to
    uint32_t general;         // 1: channels at their own centres (if_fir_channelizer_process_device_freq): bin / pword / out are indexed
                              // by CHANNEL at every decimation (decimation 16 otherwise indexes by slot)
};

struct LaunchArgs
{
    const void *in;    // device, interleaved float32 I/Q, N samples
    void *out;         // device, M samples
    const float *taps; // device, T floats
    const void *hist;  // device, the T-1 most recent samples before this call (most recent last)
    const void *hist_full; // device, start of the whole history buffer: hist_len >= T-1 samples, most recent last (the
    int hist_len;          // overlap-save kernel keeps a whole block overlap so that call boundaries do not show)
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
    const void *fft_tables_b; // second partition's tables (3074..4096 taps); filter bank at decimation 8: the all-slots form's two images; or nullptr
    int in_shift;             // overlap-save kernel: the input is read delayed by this many samples (second partition)
    void *queue; // device, 32 bytes: ticket counters of the persistent kernels (words 0-3, zeroed by the launcher) + fault count
    int diag;  // development diagnostics for the FFT kernel (0 in production)
    int grid_limit; // FFT backend: at most this many workgroups (0 = one per CU); same results, used by the queue tests
    void *dbg; // optional diagnostic stamp buffer (8192 x 4 x u64) or nullptr
    uint32_t nco_word; // SPEC §3.2 phase word P (0 = no NCO); taps are then the complex g[k] = h[k] e^{+j theta k}
    uint32_t nco_abs0; // absolute index (mod 2^32) of this call's input sample 0
    void *hist_out;       // device or nullptr: next history buffer (ping-pong) for the overlap-save kernel to write
                          // itself; nullptr = the caller runs launch_history (all other backends, or no outputs)
    uint32_t *queue_base; // host: which of the two counters in `queue` the next overlap-save launch draws from (each
    bool *queue_valid;    //       launch zeroes the other one); *queue_valid = false after anybody else touched them
    int no_fold;          // development: decimation 2 through the selecting store instead of the frequency-domain fold
    const ChanArgs *chan; // filter-bank launch (overlap-save backend, D = 4 or 16): `out` is unused, outputs go to chan->out[]
                          // (D = 16: all 16 slots are computed; out[] and rot0[] are indexed by SLOT, nullptr = not wanted)
};

// output m of a call is rotated by exp(+j*2*pi*phi/2^32), phi = nco_phi0 + m * nco_delta (mod 2^32):
// the input sample under output m has absolute index a = abs0 + n0 + m*D and the NCO phase there is -(P*a)
inline uint32_t nco_phi0(const LaunchArgs &a) { return 0u - a.nco_word * (a.nco_abs0 + (uint32_t)a.n0); }
inline uint32_t nco_delta(const LaunchArgs &a) { return 0u - a.nco_word * (uint32_t)a.D; }

#if defined(__HIPCC__)
// exp(+j*2*pi*ph/2^32) from a 32-bit phase: both sincospif arguments are exact in float32 (16 bits each)
__device__ __forceinline__ float2 nco_phasor(uint32_t ph)
{
    float sh, ch, sl, cl;
    sincospif((float)(ph >> 16) * (1.0f / 32768.0f), &sh, &ch);
    sincospif((float)(ph & 0xffffu) * (1.0f / 2147483648.0f), &sl, &cl);
    return make_float2(fmaf(ch, cl, -sh * sl), fmaf(sh, cl, ch * sl));
}
#endif

// One-time per-(kernel, device) launcher setup (dynamic-LDS attribute, CU count), safe when distinct contexts launch
// from distinct threads (include/if_fir.h promises that) and indexed by the real device id.
constexpr int MAX_DEVICES = 64;
struct DeviceSetup
{
    std::mutex mu;
    bool done[MAX_DEVICES] = {};
    int cus[MAX_DEVICES] = {};
};
// `kern` gets its dynamic-LDS limit raised to lds_bytes once per device; *cus (optional) = CU count of the device
inline hipError_t device_setup(DeviceSetup &d, int device, const void *kern, int lds_bytes, int *cus)
{
    if (device < 0 || device >= MAX_DEVICES)
        return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(d.mu);
    if (!d.done[device])
    {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess)
            return e;
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess)
            return e;
        d.cus[device] = prop.multiProcessorCount;
        d.done[device] = true;
    }
    if (cus)
        *cus = d.cus[device];
    return hipSuccess;
}

bool direct_supported(int T, int D);
hipError_t launch_fir(const LaunchArgs &a, int variant);
// overlap-save FFT backend (if_fir_fft.hip)
constexpr int FFT_TABLE_FLOATS = 2 * (4096 + 4096 + 256 + 1024 + 1024 + 64 + 256); // ... + 64 NCO row phasors + 256 W2048 twiddles
bool fft_supported(int T, int D);
bool fft_two_partitions(int T); // 3074..4096 taps: two launches (2048 + the rest), see launch_fft
struct FftSchedule
{
    int64_t RA, nA, RB, nB, tickets, wgs; // blocks per group, groups, static groups per workgroup, 0, ticket bound, workgroups
};
void fft_schedule(int64_t nblocks, int64_t wgs_max, FftSchedule &s); // host-only: run-queue layout of a launch
hipError_t launch_fft(const LaunchArgs &a);
// bank = 8 / 16: the merged table of the filter bank at decimation 8 / 16 in place of H; full_rate (D != 4, no bank): the image of
// the full-rate pipeline (D = 1, the selecting store) with its twiddles in (cos, tan) form -- the decimate-by-2 tails keep the plain one
int fft_bank_tail(int D, bool general);
void fft_bank8_plan(const uint32_t *slots, uint32_t count, bool all_slots_available, uint32_t pmask[2], uint32_t *rest);
void fft_build_tables(const float *taps, int T, int ctaps, int D, uint32_t nco_delta, double in_scale, float *tables,
                      int bank = 0, int full_rate = 0, int bank_parity = 0);

// Odd decimations F x sub, F = 3 or 5 (round 4): blocks of F x 1024 input samples, F forward 1024-point transforms of the phase
// streams and one inverse (fir_odd_kernel); *pOvlr = dropped 64-output rows of a block (2, 4 or 8).  False: no such tail (the
// full-rate pipeline + selecting store serves the pair).
bool fft_odd_tail(int T, int D, int *pF, int *pSub, int *pOvlr);
constexpr int fft_odd_table_floats(int F) { return 2 * (F * 1024 + 256 + 768 + 1024 + 64 + 64 + 256); } // G_p | TB | TC | TWD | TWE | NCO | phasor tables (round 5)
void fft_build_tables_odd(const float *taps, int T, int ctaps, int F, uint32_t nco_delta, double in_scale, float *tables);
hipError_t launch_fft_odd(const LaunchArgs &a);
int fft_overlap_rows(int T, int D);
int fft_block_advance(int T, int D); // new input samples per block: the unit at which a stream can be cut without changing a bit
// decimating tail of (T, D): D = F * sub, F = 2 or 4 the tail's own decimation (false, F = 1: full-rate kernel + selecting store)
bool fft_tail(int T, int D, int *pF, int *pSub);
// history buffers hold the last `hist_len` samples of the stream (>= T-1; hist_in/hist_out: whole buffers)
hipError_t launch_history(const void *in, const void *hist_in, void *hist_out, int hist_len, int64_t N, int in_i16,
                          hipStream_t stream);
hipError_t launch_power(const void *iq, uint64_t samples, double *acc, hipStream_t stream); // acc: device double
hipError_t launch_synth(void *iq, uint64_t first, uint64_t count, uint32_t channel, const float *tone10,
                        hipStream_t stream);

} // namespace if_fir
