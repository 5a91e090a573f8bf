// wb_detect.hip — batched WB-transponder signal detector for gfx950 (include/wb_detect.h).
//
// What it replaces: WBSpectrumMonitor.detect_signals(Buffer), one 918-bin frame per call, in the reference's Node.js
// daemon (/root/reference/software/opi-rf-manager/lib/wb_spectrum_monitor.js:36-180; round_sr :9-35).  The routine is a
// sequential state machine over the bins of one frame, so the parallelism is across FRAMES: one lane per frame, 64
// frames per workgroup.  The workgroup first copies its 64 frames from HBM into LDS with coalesced loads (a lane-per-
// frame walk over global memory would touch 64 different cache lines per instruction); each lane then scans its own
// LDS row.  Rows are padded to an odd number of dwords, so the 64 lanes of a read hit 64 different banks.
// Arithmetic: float64, same operations in the same order as the reference (every JS number is a double); this file is
// compiled with -ffp-contract=off because a fused multiply-add would round differently.  HBM-bound byte work: 2 bytes
// read per bin, ~0.2 bytes written per bin; no MFMA, nothing to tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "if_fir_kernels.h"
#include "wb_detect.h"

#define WB_API extern "C" __attribute__((visibility("default")))

namespace
{
constexpr int WB_LANES = 64;            // frames per workgroup: one lane of wave 0 each
constexpr int WB_WAVES = 4;             // waves per workgroup that copy the frames into LDS
constexpr int WB_LDS_BYTES = 160 * 1024;
constexpr int WB_EVENTS = 16;           // pending signals per lane kept in LDS before they are measured
constexpr int WB_EVENT_BYTES = WB_EVENTS * WB_LANES * 8;

thread_local char g_wb_err[256] = "";

void wb_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_wb_err, sizeof(g_wb_err), fmt, ap);
    va_end(ap);
}

// round_sr(), wb_spectrum_monitor.js:9-35
__device__ __forceinline__ double wb_round_sr(double bw)
{
    if (bw < 0.022)
        return 0;
    else if (bw < 0.060)
        return 35;
    else if (bw < 0.086)
        return 66;
    else if (bw < 0.185)
        return 125;
    else if (bw < 0.277)
        return 250;
    else if (bw < 0.388)
        return 333;
    else if (bw < 0.700)
        return 500;
    else if (bw < 1.2)
        return 1000;
    else if (bw < 1.6)
        return 1500;
    else if (bw < 2.2)
        return 2000;
    else
        return floor(bw * 5 + 0.5) / 5; // Math.round
}

// Per-frame state carried from the scan to the signal evaluation (one lane = one frame)
struct WbState
{
    double fft_freq_step;
    bool have_beacon;
    uint32_t nsig;
    wb_signal_t beacon;
};

// A signal has ended (wb_spectrum_monitor.js:85-156): measure it.  The sums of bins are sums of integers below 2^53,
// exact in float64 in any order, so they are accumulated as integers (the reference adds doubles one by one and gets
// the same number).
template <typename BinPtr>
__device__ __forceinline__ void wb_signal_end(BinPtr row, int64_t fft_length, int64_t full_start_bin, int64_t full_end_bin,
                                              WbState &st, wb_signal_t *signals, uint32_t max_signals)
{
    const double fft_full_scale_power = 16.7;
    const double fft_zero_scale_power = -3.35;
    const double fft_power_slope = (fft_full_scale_power - fft_zero_scale_power) / 65535;
    const double fft_signal_start_freq = 10490.5;
    const double fft_freq_step = st.fft_freq_step;

    const int64_t full_bin_count = full_end_bin - full_start_bin;
    const double full_start_freq = (double)full_start_bin * fft_freq_step + fft_signal_start_freq;
    const double full_end_freq = (double)full_end_bin * fft_freq_step + fft_signal_start_freq;
    const double full_center_freq = full_start_freq + (full_end_freq - full_start_freq) / 2;
    const double full_bandwidth = full_end_freq - full_start_freq;
    uint64_t full_sum = 0;
    int64_t cnt = 0;
    for (int64_t j = (int64_t)floor((double)full_start_bin + 0.3 * (double)full_bin_count);
         (double)j < (double)full_end_bin - 0.3 * (double)full_bin_count; j++)
    {
        full_sum += (uint64_t)row[j];
        cnt++;
    }
    const double full_power = (double)full_sum / (double)cnt;

    int64_t used_start_bin = full_start_bin, used_end_bin = full_end_bin;
    const double used_power_threshold = 0.75 * full_power;
    // the bounds can not be reached (the averaged middle holds a bin >= 0.75 x its mean); they keep a lane from ever
    // leaving its row
    for (int64_t j = full_start_bin; j < fft_length && (double)row[j] < used_power_threshold; j++)
        used_start_bin = j;
    for (int64_t j = full_end_bin; j >= 0 && (double)row[j] < used_power_threshold; j--)
        used_end_bin = j;

    const int64_t used_bin_count = used_end_bin - used_start_bin;
    const double used_start_freq = (double)used_start_bin * fft_freq_step + fft_signal_start_freq;
    const double used_end_freq = (double)used_end_bin * fft_freq_step + fft_signal_start_freq;
    const double used_center_freq = used_start_freq + (used_end_freq - used_start_freq) / 2;
    const double used_bandwidth = used_end_freq - used_start_freq;
    uint64_t used_sum = 0;
    for (int64_t j = used_start_bin; j < used_end_bin; j++)
        used_sum += (uint64_t)row[j];
    const double used_power = (double)used_sum / (double)used_bin_count;

    wb_signal_t s;
    s.full_start_freq = full_start_freq;
    s.full_end_freq = full_end_freq;
    s.full_center_freq = full_center_freq;
    s.full_bandwidth = full_bandwidth;
    s.full_power = full_power * fft_power_slope + fft_zero_scale_power;
    s.used_start_freq = used_start_freq;
    s.used_end_freq = used_end_freq;
    s.used_center_freq = used_center_freq;
    s.used_bandwidth = used_bandwidth;
    s.used_power = used_power * fft_power_slope + fft_zero_scale_power;
    s.symbolrate = wb_round_sr(used_bandwidth);
    s.snr = 0;
    s.sbr = 0;
    s.out_of_band = (full_end_bin == fft_length - 1);
    s.over_powered = 0;
    if (used_center_freq < 10492.0 && used_bandwidth >= 1)
    {
        st.beacon = s;
        st.have_beacon = true;
    }
    else if (s.symbolrate > 0)
    {
        if (st.nsig < max_signals)
            signals[st.nsig] = s;
        st.nsig++;
    }
}

// One lane = one frame.  `row` points at the frame's bins (LDS when staged, global otherwise).
// The bin loop only FINDS the signals (start, end): measuring one is a data-dependent walk back over its bins, and
// doing it inside the loop would serialise the 64 lanes (each lane's signals end at different bins).  The (start, end)
// pairs go to a small per-lane list `events` (capacity `event_cap`, may be 0) and are measured afterwards, event k of
// every lane at the same time; a lane whose list is full measures its pending signals on the spot (keeps the order).
template <typename BinPtr>
__device__ void wb_scan_frame(BinPtr row, int64_t fft_length, wb_frame_t *frame, wb_signal_t *signals,
                              uint32_t max_signals, uint2 *events, uint32_t event_cap, uint32_t event_stride)
{
    const double fft_full_scale_power = 16.7;
    const double fft_zero_scale_power = -3.35;
    const double fft_power_slope = (fft_full_scale_power - fft_zero_scale_power) / 65535;
    const double fft_signal_start_freq = 10490.5;
    const double fft_end_freq = 10499.5;
    const int64_t fft_avg_count = 3;
    const uint32_t threshold_sum = 3u * 16500u; // (a + b + c) / 3 >= 16500  <=>  a + b + c >= 49500 (integers)

    WbState st;
    st.fft_freq_step = (fft_end_freq - fft_signal_start_freq) / (double)fft_length;
    st.have_beacon = false;
    st.nsig = 0;
    st.beacon = wb_signal_t{};
    double noise_power = 0;
    int64_t signal_start = 0, noise_cnt = 0;
    uint32_t nev = 0;

    uint32_t b1 = fft_length > 1 ? (uint32_t)row[1] : 0u, b2 = (uint32_t)row[0];
    for (int64_t i = fft_avg_count - 1; i < fft_length; i++)
    {
        const uint32_t b0 = (uint32_t)row[i];
        const uint32_t sum = b0 + b1 + b2;
        b2 = b1;
        b1 = b0;
        if (signal_start == 0)
        {
            if (sum >= threshold_sum)
            {
                signal_start = i;
                continue;
            }
            // the reference's `sample` = sum / 3 rounded once.  For sum >= 6 that is exactly q + fl(r / 3) with
            // sum = 3 q + r (compared over every possible sum 0 .. 3 * 65535: only sum = 5 differs), which spares the
            // float64 division in the loop
            const uint32_t q = sum / 3u, r = sum - 3u * q;
            double sample = (double)q + (r == 0 ? 0.0 : (r == 1 ? 1.0 / 3.0 : 2.0 / 3.0));
            if (sum < 6u)
                sample = (double)sum / (double)fft_avg_count;
            noise_power += sample;
            noise_cnt++;
            continue;
        }
        if (sum < threshold_sum || i == fft_length - 1)
        {
            if (nev == event_cap)
            {
                for (uint32_t k = 0; k < nev; k++)
                    wb_signal_end(row, fft_length, (int64_t)events[k * event_stride].x, (int64_t)events[k * event_stride].y,
                                  st, signals, max_signals);
                nev = 0;
                if (event_cap == 0)
                    wb_signal_end(row, fft_length, signal_start, i, st, signals, max_signals);
            }
            if (event_cap)
                events[nev++ * event_stride] = make_uint2((uint32_t)signal_start, (uint32_t)i);
            signal_start = 0;
        }
    }
    // the pending signals: every lane works on its k-th one at the same time
    for (uint32_t k = 0; __any(k < nev); k++)
        if (k < nev)
            wb_signal_end(row, fft_length, (int64_t)events[k * event_stride].x, (int64_t)events[k * event_stride].y, st,
                          signals, max_signals);

    if (noise_cnt)
        noise_power /= (double)noise_cnt;
    noise_power = noise_power * fft_power_slope + fft_zero_scale_power;
    if (st.have_beacon)
        st.beacon.snr = st.beacon.full_power - noise_power;
    const uint32_t stored = st.nsig < max_signals ? st.nsig : max_signals;
    for (uint32_t k = 0; k < stored; k++)
    {
        wb_signal_t s = signals[k];
        s.snr = s.full_power - noise_power;
        if (st.have_beacon)
        {
            s.sbr = s.full_power - st.beacon.full_power;
            s.over_powered = (s.symbolrate > 500 && s.sbr > -0.7) ? 1u : 0u;
        }
        signals[k] = s;
    }
    frame->noise_power = noise_power;
    frame->beacon_valid = st.have_beacon ? 1u : 0u;
    frame->signal_count = st.nsig;
    frame->beacon = st.beacon;
}

// frames_per_wg frames per workgroup, staged through LDS (row stride `stride` uint16, an odd number of dwords)
__global__ __launch_bounds__(WB_LANES * WB_WAVES) void wb_detect_kernel(const uint16_t *__restrict__ bins, uint32_t frames,
                                                             uint32_t nbins, uint32_t frames_per_wg, uint32_t stride,
                                                             wb_frame_t *__restrict__ out_frames,
                                                             wb_signal_t *__restrict__ out_signals, uint32_t max_signals)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t rows[];
    const uint32_t first = blockIdx.x * frames_per_wg;
    const uint32_t here = (frames - first) < frames_per_wg ? (frames - first) : frames_per_wg;
    // coalesced copy HBM -> LDS by all four waves: wave w takes rows w, w+4, ...; per row every lane has up to 8 dword
    // loads in flight before the first LDS store (a load-store-load-store loop waits one HBM latency per dword)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t f = wave; f < here; f += WB_WAVES)
    {
        const uint16_t *src = bins + ((uint64_t)(first + f)) * nbins;
        uint16_t *dst = rows + (size_t)f * stride;
        if ((((uintptr_t)src) & 3) == 0)
        {
            const uint32_t pairs = nbins / 2;
            const uint32_t *src2 = reinterpret_cast<const uint32_t *>(src);
            uint32_t *dst2 = reinterpret_cast<uint32_t *>(dst); // stride is even: every LDS row starts on a dword
            for (uint32_t c = 0; c < pairs; c += 8 * WB_LANES)
            {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    const uint32_t d = c + lane + (uint32_t)u * WB_LANES;
                    v[u] = d < pairs ? src2[d] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    const uint32_t d = c + lane + (uint32_t)u * WB_LANES;
                    if (d < pairs)
                        dst2[d] = v[u];
                }
            }
            if ((nbins & 1) && lane == 0)
                dst[nbins - 1] = src[nbins - 1];
        }
        else
            for (uint32_t c = 0; c < nbins; c += 8 * WB_LANES)
            {
                uint16_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    const uint32_t d = c + lane + (uint32_t)u * WB_LANES;
                    v[u] = d < nbins ? src[d] : (uint16_t)0;
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    const uint32_t d = c + lane + (uint32_t)u * WB_LANES;
                    if (d < nbins)
                        dst[d] = v[u];
                }
            }
    }
    __syncthreads();
    // per-lane lists of pending signals behind the rows: entry k of lane l at [k * 64 + l] (conflict-free)
    uint2 *events = reinterpret_cast<uint2 *>(rows + (size_t)frames_per_wg * stride) + threadIdx.x;
    if (threadIdx.x < here)
    {
        const uint32_t f = first + threadIdx.x;
        wb_scan_frame(rows + (size_t)threadIdx.x * stride, (int64_t)nbins, out_frames + f,
                      out_signals + (size_t)f * max_signals, max_signals, events, WB_EVENTS, WB_LANES);
    }
}

// frames too long for LDS: the lane walks its frame in global memory (not a performance path)
__global__ __launch_bounds__(WB_LANES) void wb_detect_global_kernel(const uint16_t *__restrict__ bins, uint32_t frames,
                                                                    uint32_t nbins, wb_frame_t *__restrict__ out_frames,
                                                                    wb_signal_t *__restrict__ out_signals,
                                                                    uint32_t max_signals)
{
    const uint32_t f = blockIdx.x * WB_LANES + threadIdx.x;
    if (f < frames)
        wb_scan_frame(bins + (size_t)f * nbins, (int64_t)nbins, out_frames + f, out_signals + (size_t)f * max_signals,
                      max_signals, nullptr, 0, 0);
}
} // namespace

WB_API const char *wb_detect_last_error(void)
{
    return g_wb_err;
}

WB_API uint8_t wb_detect_frames_device(const uint16_t *pusDevBins, uint32_t ulFrames, uint32_t ulBins,
                                       wb_frame_t *pDevFrames, wb_signal_t *pDevSignals, uint32_t ulMaxSignals,
                                       int32_t lDevice, void *pStream)
{
    if (!ulBins)
    {
        wb_err("wb_detect_frames_device: empty frame (the reference throws \"Invalid FFT data\")");
        return 0;
    }
    if (!ulFrames)
        return 1;
    if (!pusDevBins || !pDevFrames || (!pDevSignals && ulMaxSignals))
    {
        wb_err("wb_detect_frames_device: NULL buffer");
        return 0;
    }
    hipError_t e = hipSetDevice(lDevice);
    if (e != hipSuccess)
    {
        wb_err("wb_detect_frames_device: device %d: %s (there is no CPU fallback)", lDevice, hipGetErrorString(e));
        return 0;
    }
    hipStream_t stream = (hipStream_t)pStream;
    // row stride: bins rounded up to a whole, ODD number of dwords (64 lanes x same column -> 64 different banks)
    uint32_t stride_dw = (ulBins + 1) / 2;
    stride_dw |= 1u;
    const uint32_t stride = 2 * stride_dw;
    uint32_t per_wg = (uint32_t)((WB_LDS_BYTES - WB_EVENT_BYTES) / (stride * sizeof(uint16_t)));
    if (per_wg > (uint32_t)WB_LANES)
        per_wg = WB_LANES;
    per_wg &= ~1u; // rows are odd dword counts: an even number of them keeps the event lists behind them 8-byte aligned
    if (per_wg >= 8)
    {
        const size_t lds = (size_t)per_wg * stride * sizeof(uint16_t) + WB_EVENT_BYTES; // rows (8-byte multiple) + lists
        static if_fir::DeviceSetup setup; // once per device, safe from several threads
        e = if_fir::device_setup(setup, (int)lDevice, reinterpret_cast<const void *>(wb_detect_kernel), WB_LDS_BYTES, nullptr);
        if (e != hipSuccess)
        {
            wb_err("wb_detect_frames_device: %s", hipGetErrorString(e));
            return 0;
        }
        const uint32_t wgs = (ulFrames + per_wg - 1) / per_wg;
        hipLaunchKernelGGL(wb_detect_kernel, dim3(wgs), dim3(WB_LANES * WB_WAVES), lds, stream, pusDevBins, ulFrames, ulBins, per_wg,
                           stride, pDevFrames, pDevSignals, ulMaxSignals);
    }
    else
    {
        const uint32_t wgs = (ulFrames + WB_LANES - 1) / WB_LANES;
        hipLaunchKernelGGL(wb_detect_global_kernel, dim3(wgs), dim3(WB_LANES), 0, stream, pusDevBins, ulFrames, ulBins,
                           pDevFrames, pDevSignals, ulMaxSignals);
    }
    e = hipGetLastError();
    if (e != hipSuccess)
    {
        wb_err("wb_detect_frames_device: launch failed: %s", hipGetErrorString(e));
        return 0;
    }
    return 1;
}

WB_API uint8_t wb_detect_frames(const uint16_t *pusBins, uint32_t ulFrames, uint32_t ulBins, wb_frame_t *pFrames,
                                wb_signal_t *pSignals, uint32_t ulMaxSignals, int32_t lDevice)
{
    if (!ulBins)
    {
        wb_err("wb_detect_frames: empty frame (the reference throws \"Invalid FFT data\")");
        return 0;
    }
    if (!ulFrames)
        return 1;
    if (!pusBins || !pFrames || (!pSignals && ulMaxSignals))
    {
        wb_err("wb_detect_frames: NULL buffer");
        return 0;
    }
    hipError_t e = hipSetDevice(lDevice);
    void *d_bins = nullptr, *d_frames = nullptr, *d_sig = nullptr;
    const size_t nb = (size_t)ulFrames * ulBins * sizeof(uint16_t), nf = (size_t)ulFrames * sizeof(wb_frame_t),
                 ns = (size_t)ulFrames * ulMaxSignals * sizeof(wb_signal_t);
    if (e == hipSuccess)
        e = hipMalloc(&d_bins, nb);
    if (e == hipSuccess)
        e = hipMalloc(&d_frames, nf);
    if (e == hipSuccess)
        e = hipMalloc(&d_sig, ns ? ns : 16);
    if (e == hipSuccess)
        e = hipMemcpy(d_bins, pusBins, nb, hipMemcpyHostToDevice);
    uint8_t ok = 0;
    if (e == hipSuccess)
    {
        ok = wb_detect_frames_device((const uint16_t *)d_bins, ulFrames, ulBins, (wb_frame_t *)d_frames,
                                     (wb_signal_t *)d_sig, ulMaxSignals, lDevice, nullptr);
        if (ok)
        {
            e = hipDeviceSynchronize();
            if (e == hipSuccess)
                e = hipMemcpy(pFrames, d_frames, nf, hipMemcpyDeviceToHost);
            if (e == hipSuccess && ns)
                e = hipMemcpy(pSignals, d_sig, ns, hipMemcpyDeviceToHost);
        }
    }
    if (d_bins)
        (void)hipFree(d_bins);
    if (d_frames)
        (void)hipFree(d_frames);
    if (d_sig)
        (void)hipFree(d_sig);
    if (e != hipSuccess)
    {
        wb_err("wb_detect_frames: %s (there is no CPU fallback)", hipGetErrorString(e));
        return 0;
    }
    return ok;
}
