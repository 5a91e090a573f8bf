/*
 * if_fir_channelize.c — several narrow channels out of ONE wideband IQ stream, on the libif_fir.so C ABI:
 * stdin (interleaved I,Q) -> one file of interleaved float32 I,Q per channel, e.g.
 *     rx_tool -r 2400000 ... | if_fir_channelize -t 255 -d 64 -w 0.006 -f 0.10742,0.10810,-0.2012 -o nb_%u.cf32 -i s16
 * Every channel = the low-pass prototype (half width -w, cycles/sample) centred at its own frequency (-f, cycles/sample: the
 * filter sits on the multiple of 1/4096 nearest to it), mixed down to 0 and decimated by -d (any multiple of 4 up to 64), all from
 * one pass over the input per call (if_fir_channelizer_process_device_freq).  Plain C (gcc, no HIP headers).  BUILD-DEFINED: the
 * reference has no sample-path program to replace (/root/reference/software/opi-rf-manager/index.js:3148-3535 is I2C/MQTT
 * house-keeping).
 *
 *   -t taps (odd, default 255)   -d decimation (default 8)   -w half width of the prototype's pass band (default 0.02)
 *   -f centre frequencies, comma separated, 1..16 of them    -o output name pattern with one %u (default "channel_%u.cf32")
 *   -i f32|s16 input sample format (default f32)             -c samples per call (default 2^20)     -g device (default 0)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "if_fir.h"

#define MAX_CHANNELS 16

/* a command-line number: the whole argument must parse and lie in [lMin, lMax] (atoi would turn "-1" into 4 billion taps and
 * "x" into 0, which the chunk arithmetic below divides by -- ADVICE r4) */
static uint8_t parse_u32(const char *pszArg, long lMin, long lMax, uint32_t *pulOut)
{
    char *pszEnd = NULL;
    const long lVal = strtol(pszArg, &pszEnd, 10);

    if(pszEnd == pszArg || *pszEnd || lVal < lMin || lVal > lMax)
        return 0;
    *pulOut = (uint32_t)lVal;
    return 1;
}

/* the output pattern goes to snprintf as a format: it must hold exactly one conversion, a plain %u (anything else -- %s, %n, two
 * %u, none -- is undefined behaviour or one file for all channels) */
static uint8_t pattern_ok(const char *pszPattern)
{
    unsigned uConversions = 0;

    for(const char *p = pszPattern; *p; p++)
    {
        if(*p != '%')
            continue;
        if(p[1] == '%')
        {
            p++;
            continue;
        }
        if(p[1] != 'u')
            return 0;
        uConversions++;
        p++;
    }
    return uConversions == 1 && strlen(pszPattern) < 400;
}

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
    uint32_t ulTaps = 255, ulDecimation = 8, ulChunk = 1u << 20, ubInt16 = 0, ulChannels = 0;
    int32_t lDevice = 0;
    double dWidth = 0.02, adCentre[MAX_CHANNELS];
    const char *pszPattern = "channel_%u.cf32";

    uint8_t ubArgsOk = 1;

    for(int i = 1; i + 1 < argc; i += 2)
    {
        if(!strcmp(argv[i], "-t"))
            ubArgsOk &= parse_u32(argv[i + 1], 1, 4096, &ulTaps);
        else if(!strcmp(argv[i], "-d"))
            ubArgsOk &= parse_u32(argv[i + 1], 4, 64, &ulDecimation);
        else if(!strcmp(argv[i], "-w"))
            dWidth = atof(argv[i + 1]);
        else if(!strcmp(argv[i], "-f"))
        {
            char *pszList = argv[i + 1];

            for(char *pszTok = strtok(pszList, ","); pszTok && ulChannels < MAX_CHANNELS; pszTok = strtok(NULL, ","))
                adCentre[ulChannels++] = atof(pszTok);
        }
        else if(!strcmp(argv[i], "-o"))
            pszPattern = argv[i + 1];
        else if(!strcmp(argv[i], "-i"))
            ubInt16 = !strcmp(argv[i + 1], "s16");
        else if(!strcmp(argv[i], "-c"))
            ubArgsOk &= parse_u32(argv[i + 1], 64, 1L << 28, &ulChunk);
        else if(!strcmp(argv[i], "-g"))
            lDevice = atoi(argv[i + 1]);
        else
            ulChannels = 0, i = argc;
    }
    /* validated before any arithmetic: decimation a multiple of 4 in 4..64 (the channelizer's tails), taps 1..4096, chunk 64..2^28,
     * the pattern one %u */
    if(!ubArgsOk || (ulDecimation & 3u) || !pattern_ok(pszPattern) || !(dWidth > 0.0 && dWidth < 0.5))
    {
        fprintf(stderr, "if_fir_channelize: bad argument (taps 1..4096, decimation 4, 8, ..., 64, chunk 64..2^28 samples, half width in "
                        "(0, 0.5), pattern with exactly one %%u)\n");
        ulChannels = 0;
    }
    if(!ulChannels)
    {
        fprintf(stderr, "usage: if_fir_channelize -f centre[,centre...] [-t taps] [-d decimation] [-w half width] [-o pattern] "
                        "[-i f32|s16] [-c samples] [-g device]\n");
        return 2;
    }
    if(ulChunk < 64)
        ulChunk = 64;
    ulChunk &= ~3u; /* int16 pieces stay 16-byte multiples */

    float *pfTaps = (float *)malloc(sizeof(float) * (ulTaps ? ulTaps : 1));
    if_fir_ctx_t *pFir = NULL;
    const size_t ulInBytes = ubInt16 ? 4 : 8, ulOutMax = (size_t)ulChunk / ulDecimation + 2;
    void *pHostIn = malloc(ulInBytes * ulChunk), *pHostOut = malloc(8 * ulOutMax), *pDevIn = NULL, *apDevOut[MAX_CHANNELS];
    FILE *apFile[MAX_CHANNELS];

    if(!pfTaps || !pHostIn || !pHostOut || !if_bpf_design(pfTaps, ulTaps, 0.0, dWidth, IF_BPF_WINDOW_BLACKMAN))
    {
        fprintf(stderr, "if_fir_channelize: cannot design %u taps of half width %g\n", ulTaps, dWidth);
        return 1;
    }
    if(!if_fir_init(&pFir, pfTaps, ulTaps, ulDecimation, ulChunk, lDevice))
    {
        fprintf(stderr, "if_fir_channelize: %s\n", if_fir_last_error(NULL));
        return 1;
    }
    if((ubInt16 && !if_fir_set_input_format(pFir, IF_FIR_INPUT_I16)) || !if_fir_dev_alloc(pFir, &pDevIn, ulInBytes * ulChunk))
    {
        fprintf(stderr, "if_fir_channelize: %s\n", if_fir_last_error(pFir));
        return 1;
    }
    for(uint32_t c = 0; c < ulChannels; c++)
    {
        char szName[512];

        snprintf(szName, sizeof(szName), pszPattern, c);
        apFile[c] = fopen(szName, "wb");
        if(!apFile[c] || !if_fir_dev_alloc(pFir, &apDevOut[c], 8 * ulOutMax))
        {
            fprintf(stderr, "if_fir_channelize: channel %u: cannot open %s or allocate its buffer\n", c, szName);
            return 1;
        }
    }

    uint64_t ullIn = 0, ullOut = 0;

    for(;;)
    {
        const size_t ulGot = read_fully(pHostIn, ulInBytes, ulChunk, stdin);
        uint64_t ullNow = 0;

        if(!ulGot)
            break;
        /* one pass over this piece for all channels; the context carries history, decimation phase and sample index on */
        if(!if_fir_dev_upload(pFir, pDevIn, pHostIn, ulInBytes * ulGot) ||
           !if_fir_channelizer_process_device_freq(pFir, ulChannels, adCentre, pDevIn, apDevOut, ulGot, &ullNow) ||
           !if_fir_synchronize(pFir))
        {
            fprintf(stderr, "if_fir_channelize: %s\n", if_fir_last_error(pFir));
            return 1;
        }
        for(uint32_t c = 0; c < ulChannels && ullNow; c++)
            if(!if_fir_dev_download(pFir, pHostOut, apDevOut[c], 8 * ullNow) || fwrite(pHostOut, 8, ullNow, apFile[c]) != ullNow)
            {
                fprintf(stderr, "if_fir_channelize: channel %u: %s\n", c, if_fir_last_error(pFir));
                return 1;
            }
        ullIn += ulGot;
        ullOut += ullNow;
        if(ulGot < ulChunk)
            break;
    }
    fprintf(stderr, "if_fir_channelize: %llu samples in, %llu out per channel, %u channels\n", (unsigned long long)ullIn,
            (unsigned long long)ullOut, ulChannels);
    for(uint32_t c = 0; c < ulChannels; c++)
    {
        fclose(apFile[c]);
        if_fir_dev_free(pFir, apDevOut[c]);
    }
    if_fir_dev_free(pFir, pDevIn);
    if_fir_destroy(pFir);
    free(pfTaps);
    free(pHostIn);
    free(pHostOut);
    return 0;
}
