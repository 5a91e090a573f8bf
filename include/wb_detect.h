/*
 * wb_detect.h — C ABI of the batched WB-transponder signal detector.
 *
 * Replaces, for many spectrum frames at once, the static routine WBSpectrumMonitor.detect_signals(Buffer) of the
 * reference's host daemon (/root/reference/software/opi-rf-manager/lib/wb_spectrum_monitor.js:36-180): a frame is
 * ulBins uint16 little-endian magnitudes (918 on the air); the result is the noise power, the beacon and the list of
 * signals with the fields the reference returns (same names, same float64 arithmetic in the same order, so results
 * are bit-identical to the reference's: tests/golden/wb_detect_golden.json holds vectors captured from it).
 * Conventions as include/if_fir.h: uint8_t status, 1 = success, 0 = failure + wb_detect_last_error().
 */
#ifndef WB_DETECT_H
#define WB_DETECT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one detected signal: the reference's `signal` object (wb_spectrum_monitor.js:127-143), booleans as 0/1 */
typedef struct
{
    double full_start_freq, full_end_freq, full_center_freq, full_bandwidth, full_power;
    double used_start_freq, used_end_freq, used_center_freq, used_bandwidth, used_power;
    double symbolrate, snr, sbr;
    uint32_t out_of_band, over_powered;
} wb_signal_t;

/* one frame's result: the reference's `data` object (wb_spectrum_monitor.js:51-55) */
typedef struct
{
    double noise_power;
    uint32_t beacon_valid;  /* data.beacon !== undefined */
    uint32_t signal_count;  /* data.signals.length; may exceed the capacity given by the caller: the rest is dropped */
    wb_signal_t beacon;
} wb_frame_t;

/* Device buffers: pusDevBins = ulFrames x ulBins uint16; pDevFrames = ulFrames results; pDevSignals = ulFrames x
 * ulMaxSignals (signal k of frame f at [f * ulMaxSignals + k]).  ulBins >= 1 (the reference rejects an empty frame).
 * Asynchronous on pStream (a hipStream_t, NULL = the default stream) of device lDevice. */
uint8_t wb_detect_frames_device(const uint16_t *pusDevBins, uint32_t ulFrames, uint32_t ulBins, wb_frame_t *pDevFrames,
                                wb_signal_t *pDevSignals, uint32_t ulMaxSignals, int32_t lDevice, void *pStream);
/* Host buffers: copies in, runs the kernel, copies out, synchronous. */
uint8_t wb_detect_frames(const uint16_t *pusBins, uint32_t ulFrames, uint32_t ulBins, wb_frame_t *pFrames,
                         wb_signal_t *pSignals, uint32_t ulMaxSignals, int32_t lDevice);
const char *wb_detect_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* WB_DETECT_H */
