/*
 * if_fir_pipe.c — stream filter for SDR pipelines on the libif_fir.so C ABI:  stdin (interleaved I,Q) -> stdout
 * (interleaved float32 I,Q), e.g.   rx_tool ... | if_fir_pipe -t 255 -d 4 -b 0.15:0.25 -i s16 | demod ...
 * Plain C (gcc, no HIP headers).  BUILD-DEFINED: the reference has no sample-path program to replace
 * (/root/reference/software/opi-rf-manager/index.js:3148-3535 is I2C/MQTT house-keeping).
 *
 *   -t taps (odd, default 255)    -d decimation (default 4)     -b low:high pass band in cycles/sample (default 0.15:0.25)
 *   -n nco frequency in cycles/sample, mixed down ahead of the filter (default 0 = off)
 *   -i f32|s16 input sample format (default f32)                  -c samples per call (default 2^20)
 *   -g device (default 0)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "if_fir.h"

static size_t read_fully(void *pBuf, size_t ulSize, size_t ulCount, FILE *pIn)
{
    size_t ulGot = 0;

    while(ulGot < ulCount)
    {
        const size_t ulNow = fread((char *)pBuf + ulGot * ulSize, ulSize, ulCount - ulGot, pIn);

        if(!ulNow)
            break;
        ulGot += ulNow;
    }
    return ulGot;
}

int main(int argc, char **argv)
{
    uint32_t ulTaps = 255, ulDecimation = 4, ulChunk = 1u << 20, ubInt16 = 0;
    int32_t lDevice = 0;
    double dLow = 0.15, dHigh = 0.25, dNco = 0.0;

    for(int i = 1; i + 1 < argc; i += 2)
    {
        if(!strcmp(argv[i], "-t"))
            ulTaps = (uint32_t)atoi(argv[i + 1]);
        else if(!strcmp(argv[i], "-d"))
            ulDecimation = (uint32_t)atoi(argv[i + 1]);
        else if(!strcmp(argv[i], "-b") && sscanf(argv[i + 1], "%lf:%lf", &dLow, &dHigh) == 2)
            ;
        else if(!strcmp(argv[i], "-n"))
            dNco = atof(argv[i + 1]);
        else if(!strcmp(argv[i], "-i"))
            ubInt16 = !strcmp(argv[i + 1], "s16");
        else if(!strcmp(argv[i], "-c"))
            ulChunk = (uint32_t)atoi(argv[i + 1]);
        else if(!strcmp(argv[i], "-g"))
            lDevice = atoi(argv[i + 1]);
        else
        {
            fprintf(stderr, "usage: if_fir_pipe [-t taps] [-d decimation] [-b low:high] [-n nco] [-i f32|s16] [-c samples] [-g device]\n");
            return 2;
        }
    }
    if(ulChunk < 16)
        ulChunk = 16;
    ulChunk &= ~3u; /* int16 pieces stay 16-byte multiples */

    float *pfTaps = (float *)malloc(sizeof(float) * (ulTaps ? ulTaps : 1));
    if_fir_ctx_t *pFir = NULL;
    void *pIn = NULL, *pOut = NULL;
    const size_t ulInBytes = ubInt16 ? 4 : 8;

    if(!pfTaps || !if_bpf_design(pfTaps, ulTaps, dLow, dHigh, IF_BPF_WINDOW_BLACKMAN))
    {
        fprintf(stderr, "if_fir_pipe: cannot design %u taps for the band %g:%g\n", ulTaps, dLow, dHigh);
        return 1;
    }
    if(!if_fir_init(&pFir, pfTaps, ulTaps, ulDecimation, ulChunk, lDevice))
    {
        fprintf(stderr, "if_fir_pipe: %s\n", if_fir_last_error(NULL));
        return 1;
    }
    if((ubInt16 && !if_fir_set_input_format(pFir, IF_FIR_INPUT_I16)) || (dNco != 0.0 && !if_fir_set_nco(pFir, dNco)) ||
       !if_fir_host_alloc(pFir, &pIn, ulInBytes * ulChunk) || !if_fir_host_alloc(pFir, &pOut, 8 * ((size_t)ulChunk / ulDecimation + 2)))
    {
        fprintf(stderr, "if_fir_pipe: %s\n", if_fir_last_error(pFir));
        return 1;
    }

    uint64_t ullIn = 0, ullOut = 0;

    for(;;)
    {
        const size_t ulGot = read_fully(pIn, ulInBytes, ulChunk, stdin);
        uint64_t ullNow = 0;

        if(!ulGot)
            break;
        if(!if_fir_process(pFir, (const float *)pIn, (float *)pOut, ulGot, &ullNow))
        {
            fprintf(stderr, "if_fir_pipe: %s\n", if_fir_last_error(pFir));
            return 1;
        }
        if(ullNow && fwrite(pOut, 8, ullNow, stdout) != ullNow)
        {
            fprintf(stderr, "if_fir_pipe: short write\n");
            return 1;
        }
        ullIn += ulGot;
        ullOut += ullNow;
        if(ulGot < ulChunk)
            break;
    }
    fflush(stdout);
    fprintf(stderr, "if_fir_pipe: %llu samples in, %llu out\n", (unsigned long long)ullIn, (unsigned long long)ullOut);
    if_fir_host_free(pFir, pIn);
    if_fir_host_free(pFir, pOut);
    if_fir_destroy(pFir);
    free(pfTaps);
    return 0;
}
