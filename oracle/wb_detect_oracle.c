/*
 * wb_detect_oracle.c — CPU restatement of the reference's WB-transponder signal detector.  TEST INFRASTRUCTURE ONLY
 * (only tests/ may call it; the product is qo-100-tools_amd/csrc/wb_detect.hip).
 *
 * PARITY PINNED: follows WBSpectrumMonitor.detect_signals() and round_sr() of
 * /root/reference/software/opi-rf-manager/lib/wb_spectrum_monitor.js:9-35 and :36-180 statement by statement (every JS
 * number is an IEEE double; the same operations in the same order give the same bits), and is checked against
 * tests/golden/wb_detect_golden.json, vectors captured by running that routine under node (make_wb_golden.js).
 * Build without floating-point contraction (oracle/Makefile: -ffp-contract=off): an fma would change the rounding.
 */
#include <math.h>
#include <stdint.h>

#include "wb_detect.h"

#define ORACLE_API __attribute__((visibility("default")))

/* round_sr(), wb_spectrum_monitor.js:9-35 */
static double wb_round_sr(double bw)
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
        return floor(bw * 5 + 0.5) / 5; /* Math.round: half towards +infinity */
}

/* detect_signals(), wb_spectrum_monitor.js:36-180.  pusBins: fft_length words (readUInt16LE).  Returns 0 for the frames
 * the reference rejects (fft.length === 0), 2 where it would throw a RangeError, 1 otherwise. */
ORACLE_API int oracle_wb_detect(const uint16_t *pusBins, uint32_t ulBins, wb_frame_t *pFrame, wb_signal_t *pSignals,
                                uint32_t ulMaxSignals)
{
    if (!ulBins)
        return 0;

    const int64_t fft_length = (int64_t)ulBins;
    const double fft_signal_threshold = 16500;
    const double fft_full_scale_power = 16.7;
    const double fft_zero_scale_power = -3.35;
    const double fft_power_slope = (fft_full_scale_power - fft_zero_scale_power) / 65535;
    const double fft_signal_start_freq = 10490.5;
    const double fft_end_freq = 10499.5;
    const double fft_freq_step = (fft_end_freq - fft_signal_start_freq) / (double)fft_length;
    const int64_t fft_avg_count = 3;

    double noise_power = 0;
    int have_beacon = 0;
    uint32_t ulSignals = 0;
    int64_t signal_start = 0;
    int64_t noise_cnt = 0;

    for (int64_t i = fft_avg_count - 1; i < fft_length; i++)
    {
        double sample = 0;

        for (int64_t j = 0; j < fft_avg_count; j++)
            sample += (double)pusBins[i - j];

        sample /= (double)fft_avg_count;

        if (signal_start == 0)
        {
            if (sample >= fft_signal_threshold)
            {
                signal_start = i;
                continue;
            }
            noise_power += sample;
            noise_cnt++;
            continue;
        }

        if (sample < fft_signal_threshold || i == fft_length - 1)
        {
            const int64_t full_start_bin = signal_start;
            const int64_t full_end_bin = i;
            const int64_t full_bin_count = full_end_bin - full_start_bin;
            const double full_start_freq = (double)full_start_bin * fft_freq_step + fft_signal_start_freq;
            const double full_end_freq = (double)full_end_bin * fft_freq_step + fft_signal_start_freq;
            const double full_center_freq = full_start_freq + (full_end_freq - full_start_freq) / 2;
            const double full_bandwidth = full_end_freq - full_start_freq;
            double full_power = 0;
            int64_t cnt = 0;

            for (int64_t j = (int64_t)floor((double)full_start_bin + 0.3 * (double)full_bin_count);
                 (double)j < (double)full_end_bin - 0.3 * (double)full_bin_count; j++)
            {
                full_power += (double)pusBins[j];
                cnt++;
            }
            full_power /= (double)cnt;

            int64_t used_start_bin = full_start_bin;
            int64_t used_end_bin = full_end_bin;
            const double used_power_threshold = 0.75 * full_power;

            /* readUInt16LE throws a RangeError outside the buffer (cannot happen: the averaged middle holds a bin at
             * or above 0.75 x its own mean); returned as 2 instead of reading out of bounds */
            for (int64_t j = full_start_bin; ; j++)
            {
                if (j >= fft_length)
                    return 2;
                if (!((double)pusBins[j] < used_power_threshold))
                    break;
                used_start_bin = j;
            }
            for (int64_t j = full_end_bin; ; j--)
            {
                if (j < 0)
                    return 2;
                if (!((double)pusBins[j] < used_power_threshold))
                    break;
                used_end_bin = j;
            }

            const int64_t used_bin_count = used_end_bin - used_start_bin;
            const double used_start_freq = (double)used_start_bin * fft_freq_step + fft_signal_start_freq;
            const double used_end_freq = (double)used_end_bin * fft_freq_step + fft_signal_start_freq;
            const double used_center_freq = used_start_freq + (used_end_freq - used_start_freq) / 2;
            const double used_bandwidth = used_end_freq - used_start_freq;
            double used_power = 0;

            for (int64_t j = used_start_bin; j < used_end_bin; j++)
                used_power += (double)pusBins[j];
            used_power /= (double)used_bin_count;

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
                pFrame->beacon = s;
                have_beacon = 1;
            }
            else if (s.symbolrate > 0)
            {
                if (ulSignals < ulMaxSignals)
                    pSignals[ulSignals] = s;
                ulSignals++;
            }
            signal_start = 0;
        }
    }

    if (noise_cnt)
        noise_power /= (double)noise_cnt;
    noise_power = noise_power * fft_power_slope + fft_zero_scale_power;

    if (have_beacon)
        pFrame->beacon.snr = pFrame->beacon.full_power - noise_power;

    for (uint32_t k = 0; k < ulSignals && k < ulMaxSignals; k++)
    {
        wb_signal_t *s = &pSignals[k];

        s->snr = s->full_power - noise_power;
        if (have_beacon)
        {
            s->sbr = s->full_power - pFrame->beacon.full_power;
            s->over_powered = s->symbolrate > 500 && s->sbr > -0.7;
        }
    }
    pFrame->noise_power = noise_power;
    pFrame->beacon_valid = (uint32_t)have_beacon;
    pFrame->signal_count = ulSignals;
    return 1;
}
