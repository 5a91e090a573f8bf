/*
 * if_fir_selftest.c — the C ABI used the way a C host program would use it (gcc, no HIP headers, no Python):
 * design taps, filter a stream through the host-buffer entry points in ragged pieces, the NCO, the int16 format, the
 * filter bank through device buffers, error reporting.  Results are compared with a float64 direct convolution written
 * out here (test code: y[n] = sum h[k] x[n-k], SPEC §2/§3.2).  Exit code 0 = all checks passed.
 * Built and run by tests/test_c_abi.py on the GPU box:  gcc -std=c99 -Iinclude tests/c/if_fir_selftest.c -L... -lif_fir -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "if_fir.h"

#define CHECK(cond, ...)                          \
    do                                            \
    {                                             \
        if(!(cond))                               \
        {                                         \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);         \
            fprintf(stderr, "\n");                \
            return 1;                             \
        }                                         \
    } while(0)

static uint32_t g_ulSeed = 12345u;
static float frand(void) /* uniform in [-1, 1) */
{
    g_ulSeed = g_ulSeed * 1664525u + 1013904223u;
    return (float)((double)(g_ulSeed >> 8) / 8388608.0 - 1.0);
}

/* float64 reference: optional NCO (phase word), real taps, decimation; x interleaved I,Q */
static void reference(const float *pfTaps, uint32_t ulTaps, uint32_t ulDecimation, uint32_t ulPhaseWord, const float *pfX,
                      uint64_t ullSamples, double *pdY)
{
    for(uint64_t m = 0; m * ulDecimation < ullSamples; m++)
    {
        const uint64_t n = m * ulDecimation;
        double dRe = 0.0, dIm = 0.0;

        for(uint32_t k = 0; k < ulTaps && k <= n; k++)
        {
            const uint64_t a = n - k;
            const uint32_t ph = (uint32_t)((uint64_t)ulPhaseWord * a);
            const double dAng = -6.283185307179586476925286766559 * ((double)ph / 4294967296.0);
            const double dC = cos(dAng), dS = sin(dAng);
            const double dXr = (double)pfX[2 * a] * dC - (double)pfX[2 * a + 1] * dS;
            const double dXi = (double)pfX[2 * a] * dS + (double)pfX[2 * a + 1] * dC;

            dRe += (double)pfTaps[k] * dXr;
            dIm += (double)pfTaps[k] * dXi;
        }
        pdY[2 * m] = dRe;
        pdY[2 * m + 1] = dIm;
    }
}

static double max_rel_err(const float *pfGot, const double *pdRef, uint64_t ullValues)
{
    double dMax = 0.0, dScale = 0.0;

    for(uint64_t i = 0; i < ullValues; i++)
    {
        const double dE = fabs((double)pfGot[i] - pdRef[i]);

        if(dE > dMax)
            dMax = dE;
        if(fabs(pdRef[i]) > dScale)
            dScale = fabs(pdRef[i]);
    }
    return dScale > 0.0 ? dMax / dScale : dMax;
}

int main(void)
{
    enum { TAPS = 255, DECIM = 4, SAMPLES = 60001 };
    float *pfTaps = (float *)malloc(sizeof(float) * TAPS);
    float *pfX = (float *)malloc(sizeof(float) * 2 * SAMPLES);
    int16_t *psX = (int16_t *)malloc(sizeof(int16_t) * 2 * SAMPLES);
    float *pfXq = (float *)malloc(sizeof(float) * 2 * SAMPLES);
    const uint64_t ullOutMax = (SAMPLES + DECIM - 1) / DECIM;
    float *pfY = (float *)malloc(sizeof(float) * 2 * (ullOutMax + 2));
    double *pdRef = (double *)malloc(sizeof(double) * 2 * (ullOutMax + 2));
    if_fir_ctx_t *pFir = NULL;
    uint64_t ullOut = 0, ullTotal = 0;
    double dFreq = 0.0;

    CHECK(pfTaps && pfX && psX && pfXq && pfY && pdRef, "out of memory");
    for(uint32_t i = 0; i < 2 * SAMPLES; i++)
    {
        pfX[i] = frand();
        psX[i] = (int16_t)(frand() * 20000.0f);
        pfXq[i] = (float)psX[i] * (1.0f / 32768.0f);
    }

    /* argument errors come back as 0 + message, nothing aborts */
    CHECK(!if_bpf_design(pfTaps, 254, 0.15, 0.25, IF_BPF_WINDOW_BLACKMAN), "even tap count accepted");
    CHECK(!if_fir_init(&pFir, pfTaps, 0, 1, 16, 0) && pFir == NULL && strlen(if_fir_last_error(NULL)) > 0, "0 taps accepted");

    CHECK(if_bpf_design(pfTaps, TAPS, 0.0, 0.05, IF_BPF_WINDOW_BLACKMAN), "if_bpf_design");
    CHECK(if_fir_init(&pFir, pfTaps, TAPS, DECIM, SAMPLES, 0), "if_fir_init: %s", if_fir_last_error(NULL));
    CHECK(if_fir_get_backend(pFir) == IF_FIR_BACKEND_HIP_FFT, "AUTO backend = %u", if_fir_get_backend(pFir));

    /* 1. plain filter, stream cut into ragged pieces (history and decimation phase carried by the context) */
    {
        const uint64_t aullCuts[] = {0, 1, 1022, 20003, 40002, SAMPLES};

        reference(pfTaps, TAPS, DECIM, 0, pfX, SAMPLES, pdRef);
        ullTotal = 0;
        for(uint32_t i = 0; i + 1 < sizeof(aullCuts) / sizeof(aullCuts[0]); i++)
        {
            CHECK(if_fir_process(pFir, pfX + 2 * aullCuts[i], pfY + 2 * ullTotal, aullCuts[i + 1] - aullCuts[i], &ullOut),
                  "if_fir_process: %s", if_fir_last_error(pFir));
            ullTotal += ullOut;
        }
        CHECK(ullTotal == ullOutMax, "output count %llu != %llu", (unsigned long long)ullTotal, (unsigned long long)ullOutMax);
        CHECK(max_rel_err(pfY, pdRef, 2 * ullTotal) <= 1e-6, "plain filter error %g", max_rel_err(pfY, pdRef, 2 * ullTotal));
    }
    /* 2. a call beyond ullMaxSamples fails and the context stays usable */
    CHECK(!if_fir_process(pFir, pfX, pfY, SAMPLES + 1, &ullOut) && strlen(if_fir_last_error(pFir)) > 0, "oversized call accepted");

    /* 3. NCO */
    CHECK(if_fir_reset(pFir) && if_fir_set_nco(pFir, 0.1875), "if_fir_set_nco: %s", if_fir_last_error(pFir));
    CHECK(if_fir_get_nco(pFir, &dFreq) && dFreq == 0.1875, "if_fir_get_nco %g", dFreq);
    reference(pfTaps, TAPS, DECIM, 0x30000000u, pfX, SAMPLES, pdRef);
    CHECK(if_fir_process(pFir, pfX, pfY, SAMPLES, &ullOut) && ullOut == ullOutMax, "NCO process: %s", if_fir_last_error(pFir));
    CHECK(max_rel_err(pfY, pdRef, 2 * ullOut) <= 1e-6, "NCO error %g", max_rel_err(pfY, pdRef, 2 * ullOut));
    CHECK(!if_fir_set_backend(pFir, IF_FIR_BACKEND_HIP_DIRECT), "direct form accepted an NCO");
    CHECK(if_fir_set_nco(pFir, 0.0), "NCO off");

    /* 4. int16 input (the same host entry point takes the int16 buffer) */
    CHECK(if_fir_set_input_format(pFir, IF_FIR_INPUT_I16), "int16 format: %s", if_fir_last_error(pFir));
    reference(pfTaps, TAPS, DECIM, 0, pfXq, SAMPLES, pdRef);
    CHECK(if_fir_process(pFir, (const float *)(const void *)psX, pfY, SAMPLES, &ullOut) && ullOut == ullOutMax, "int16 process");
    CHECK(max_rel_err(pfY, pdRef, 2 * ullOut) <= 1e-6, "int16 error %g", max_rel_err(pfY, pdRef, 2 * ullOut));
    CHECK(if_fir_set_input_format(pFir, IF_FIR_INPUT_F32), "float32 format");

    /* 5. filter bank through device buffers: channel c = NCO(slot/16) + prototype + decimate-by-4 */
    {
        const uint32_t aulSlots[3] = {0, 3, 14};
        void *pDevIn = NULL, *apDevOut[3] = {NULL, NULL, NULL};

        CHECK(if_fir_reset(pFir), "reset");
        CHECK(if_fir_dev_alloc(pFir, &pDevIn, 8 * SAMPLES) && if_fir_dev_upload(pFir, pDevIn, pfX, 8 * SAMPLES), "device input");
        for(uint32_t c = 0; c < 3; c++)
            CHECK(if_fir_dev_alloc(pFir, &apDevOut[c], 8 * (ullOutMax + 2)), "device output");
        CHECK(if_fir_channelizer_process_device(pFir, 3, aulSlots, pDevIn, apDevOut, SAMPLES, &ullOut) && ullOut == ullOutMax,
              "filter bank: %s", if_fir_last_error(pFir));
        CHECK(if_fir_synchronize(pFir), "synchronize");
        for(uint32_t c = 0; c < 3; c++)
        {
            reference(pfTaps, TAPS, DECIM, aulSlots[c] << 28, pfX, SAMPLES, pdRef);
            CHECK(if_fir_dev_download(pFir, pfY, apDevOut[c], 8 * ullOut), "download");
            CHECK(max_rel_err(pfY, pdRef, 2 * ullOut) <= 1e-6, "filter bank slot %u error %g", aulSlots[c],
                  max_rel_err(pfY, pdRef, 2 * ullOut));
            CHECK(if_fir_dev_free(pFir, apDevOut[c]), "free");
        }
        CHECK(if_fir_dev_free(pFir, pDevIn), "free");
    }
    if_fir_destroy(pFir);
    pFir = NULL;

    /* 6. the bank at the channel rate: decimation 16, all 16 slots from one forward transform, three of them wanted */
    {
        enum { DECIM16 = 16 };
        const uint32_t aulSlots[3] = {1, 8, 15};
        const uint64_t ullOut16 = (SAMPLES + DECIM16 - 1) / DECIM16;
        void *pDevIn = NULL, *apDevOut[3] = {NULL, NULL, NULL};

        CHECK(if_bpf_design(pfTaps, TAPS, 0.0, 0.02, IF_BPF_WINDOW_BLACKMAN), "prototype design");
        CHECK(if_fir_init(&pFir, pfTaps, TAPS, DECIM16, SAMPLES, 0), "init (decimation 16): %s", if_fir_last_error(NULL));
        CHECK(if_fir_dev_alloc(pFir, &pDevIn, 8 * SAMPLES) && if_fir_dev_upload(pFir, pDevIn, pfX, 8 * SAMPLES), "device input");
        for(uint32_t c = 0; c < 3; c++)
            CHECK(if_fir_dev_alloc(pFir, &apDevOut[c], 8 * (ullOut16 + 2)), "device output");
        CHECK(if_fir_channelizer_process_device(pFir, 3, aulSlots, pDevIn, apDevOut, SAMPLES, &ullOut) && ullOut == ullOut16,
              "filter bank at decimation 16: %s", if_fir_last_error(pFir));
        CHECK(if_fir_synchronize(pFir), "synchronize");
        for(uint32_t c = 0; c < 3; c++)
        {
            reference(pfTaps, TAPS, DECIM16, aulSlots[c] << 28, pfX, SAMPLES, pdRef);
            CHECK(if_fir_dev_download(pFir, pfY, apDevOut[c], 8 * ullOut), "download");
            CHECK(max_rel_err(pfY, pdRef, 2 * ullOut) <= 1e-6, "filter bank (16) slot %u error %g", aulSlots[c],
                  max_rel_err(pfY, pdRef, 2 * ullOut));
            CHECK(if_fir_dev_free(pFir, apDevOut[c]), "free");
        }
        /* the same context: three channels at their own centres (multiples of fs/4096: exactly what three contexts with
         * if_fir_set_nco compute), from one pass */
        {
            const double adCentre[3] = {300.0 / 4096.0, -1000.0 / 4096.0, 0.25};
            CHECK(if_fir_reset(pFir), "reset");
            for(uint32_t c = 0; c < 3; c++)
                CHECK(if_fir_dev_alloc(pFir, &apDevOut[c], 8 * (ullOut16 + 2)), "device output");
            CHECK(if_fir_channelizer_process_device_freq(pFir, 3, adCentre, pDevIn, apDevOut, SAMPLES, &ullOut) && ullOut == ullOut16,
                  "channels at their own centres: %s", if_fir_last_error(pFir));
            CHECK(if_fir_synchronize(pFir), "synchronize");
            for(uint32_t c = 0; c < 3; c++)
            {
                reference(pfTaps, TAPS, DECIM16, (uint32_t)(int64_t)llround(adCentre[c] * 4294967296.0), pfX, SAMPLES, pdRef);
                CHECK(if_fir_dev_download(pFir, pfY, apDevOut[c], 8 * ullOut), "download");
                CHECK(max_rel_err(pfY, pdRef, 2 * ullOut) <= 1e-6, "channel at centre %g error %g", adCentre[c],
                      max_rel_err(pfY, pdRef, 2 * ullOut));
                CHECK(if_fir_dev_free(pFir, apDevOut[c]), "free");
            }
        }
        /* and one tuned, decimated channel the ordinary way: NCO + decimate-by-16 through if_fir_process */
        CHECK(if_fir_reset(pFir) && if_fir_set_nco(pFir, 0.1371) && if_fir_get_nco(pFir, &dFreq), "NCO: %s", if_fir_last_error(pFir));
        reference(pfTaps, TAPS, DECIM16, (uint32_t)(int64_t)llround(dFreq * 4294967296.0), pfX, SAMPLES, pdRef);
        CHECK(if_fir_process(pFir, pfX, pfY, SAMPLES, &ullOut) && ullOut == ullOut16, "NCO + decimate-by-16");
        CHECK(max_rel_err(pfY, pdRef, 2 * ullOut) <= 1e-6, "NCO + decimate-by-16 error %g", max_rel_err(pfY, pdRef, 2 * ullOut));
        CHECK(if_fir_dev_free(pFir, pDevIn), "free");
    }
    if_fir_destroy(pFir);
    pFir = NULL;

    /* 7. decimations that run behind a tail keeping every sub-th output (12 = 4 x 3, 6 = 2 x 3), in two ragged pieces */
    {
        static const uint32_t aulDecim[2] = {12, 6};
        for(uint32_t k = 0; k < 2; k++)
        {
            const uint32_t ulD = aulDecim[k];
            const uint64_t ullCut = 33333, ullExpect = (SAMPLES + ulD - 1) / ulD;
            uint64_t ullOutA = 0, ullOutB = 0;

            CHECK(if_fir_init(&pFir, pfTaps, TAPS, ulD, SAMPLES, 0), "init (decimation %u): %s", ulD, if_fir_last_error(NULL));
            CHECK(if_fir_process(pFir, pfX, pfY, ullCut, &ullOutA), "decimation %u, piece 1: %s", ulD, if_fir_last_error(pFir));
            CHECK(if_fir_process(pFir, pfX + 2 * ullCut, pfY + 2 * ullOutA, SAMPLES - ullCut, &ullOutB) && ullOutA + ullOutB == ullExpect,
                  "decimation %u, piece 2: %s", ulD, if_fir_last_error(pFir));
            reference(pfTaps, TAPS, ulD, 0, pfX, SAMPLES, pdRef);
            CHECK(max_rel_err(pfY, pdRef, 2 * ullExpect) <= 1e-6, "decimation %u error %g", ulD, max_rel_err(pfY, pdRef, 2 * ullExpect));
            if_fir_destroy(pFir);
            pFir = NULL;
        }
    }
    printf("if_fir_selftest: all checks passed\n");
    return 0;
}
