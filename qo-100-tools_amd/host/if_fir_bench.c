/*
 * if_fir_bench.c — plain-C host program on the libif_fir.so C ABI (include/if_fir.h): designs the IF band-pass taps,
 * generates the synthetic IQ stream on the device, filters it, and prints one JSON line per configuration.
 * No HIP headers: device memory goes through if_fir_dev_alloc/free.  BUILD-DEFINED (the reference has no host
 * program on a sample path: /root/reference/software/opi-rf-manager/index.js:3148-3535 is I2C/MQTT house-keeping).
 *
 * usage: if_fir_bench [taps decimation log2_samples [reps]]      (default: the three BASELINE GPU configurations)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "if_fir.h"

static uint8_t bench_run(uint32_t ulTaps, uint32_t ulDecimation, uint32_t ulLog2Samples, uint32_t ulReps)
{
    const uint64_t ullSamples = 1ull << ulLog2Samples;
    float *pfTaps = (float *)malloc(sizeof(float) * ulTaps);
    if_fir_ctx_t *pFir = NULL;
    void *pDevIn = NULL, *pDevOut = NULL;
    uint64_t ullOut = 0;
    float fMs = 0.0f;
    char szInfo[160];
    uint8_t ubOK = 0;

    if(!pfTaps)
        return 0;

    if(!if_bpf_design(pfTaps, ulTaps, 0.15, 0.25, IF_BPF_WINDOW_BLACKMAN))
    {
        fprintf(stderr, "if_bpf_design failed for %u taps\n", ulTaps);
        goto done;
    }

    if(!if_fir_init(&pFir, pfTaps, ulTaps, ulDecimation, 0, 0))
    {
        fprintf(stderr, "if_fir_init: %s\n", if_fir_last_error(NULL));
        goto done;
    }

    ullOut = if_fir_out_count(pFir, ullSamples);

    if(!if_fir_dev_alloc(pFir, &pDevIn, 8 * ullSamples) || !if_fir_dev_alloc(pFir, &pDevOut, 8 * (ullOut + 2)))
    {
        fprintf(stderr, "device allocation: %s\n", if_fir_last_error(pFir));
        goto done;
    }

    if(!if_fir_synth_device(pFir, pDevIn, 0, ullSamples, 0) || !if_fir_synchronize(pFir))
    {
        fprintf(stderr, "if_fir_synth_device: %s\n", if_fir_last_error(pFir));
        goto done;
    }

    /* product ABI only: back-to-back asynchronous calls between two synchronisations, host clock around them */
    {
        struct timespec t0, t1;
        uint32_t i;

        for(i = 0; i < 10; i++)
            if(!if_fir_process_device(pFir, pDevIn, pDevOut, ullSamples, NULL))
                break;

        if(i < 10 || !if_fir_synchronize(pFir))
        {
            fprintf(stderr, "if_fir_process_device: %s\n", if_fir_last_error(pFir));
            goto done;
        }

        clock_gettime(CLOCK_MONOTONIC, &t0);

        for(i = 0; i < ulReps; i++)
            if(!if_fir_process_device(pFir, pDevIn, pDevOut, ullSamples, NULL))
                break;

        if(i < ulReps || !if_fir_synchronize(pFir))
        {
            fprintf(stderr, "if_fir_process_device: %s\n", if_fir_last_error(pFir));
            goto done;
        }

        clock_gettime(CLOCK_MONOTONIC, &t1);
        fMs = (float)(((double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6) / (double)ulReps);
    }

    if_fir_device_info(pFir, szInfo, sizeof(szInfo));

    {
        const double dBytes = (8.0 + 8.0 / ulDecimation) * (double)ullSamples;
        const double dFlops = 4.0 * ulTaps / ulDecimation * (double)ullSamples;

        printf("{\"program\": \"if_fir_bench.c\", \"taps\": %u, \"decimation\": %u, \"samples\": %llu, \"backend\": %u, "
               "\"ms_per_call\": %.4f, \"msamples_per_s\": %.1f, \"hbm_gbs\": %.1f, \"hbm_frac_of_8TBs\": %.4f, "
               "\"valu_tflops\": %.2f, \"device\": \"%s\"}\n",
               ulTaps, ulDecimation, (unsigned long long)ullSamples, if_fir_get_backend(pFir), fMs,
               (double)ullSamples / fMs / 1e3, dBytes / fMs / 1e6, dBytes / fMs / 1e6 / 8000.0, dFlops / fMs / 1e9, szInfo);
    }

    ubOK = 1;

done:
    if(pFir)
    {
        if(pDevIn)
            if_fir_dev_free(pFir, pDevIn);
        if(pDevOut)
            if_fir_dev_free(pFir, pDevOut);
        if_fir_destroy(pFir);
    }
    free(pfTaps);

    return ubOK;
}

int main(int argc, char **argv)
{
    if(argc >= 4)
        return bench_run((uint32_t)atoi(argv[1]), (uint32_t)atoi(argv[2]), (uint32_t)atoi(argv[3]),
                         argc >= 5 ? (uint32_t)atoi(argv[4]) : 30) ? 0 : 1;

    uint8_t ubOK = 1;

    ubOK &= bench_run(127, 1, 26, 30); /* BASELINE configs[1] */
    ubOK &= bench_run(255, 4, 28, 30); /* BASELINE configs[2] */
    ubOK &= bench_run(255, 1, 28, 10); /* 255 taps without decimation (info row of SURVEY §8d) */

    return ubOK ? 0 : 1;
}
