/*
 * if_bpf_design.c — IF band-pass tap generator (host side, plain C99, runs once per filter).
 *
 * BUILD-DEFINED: the north_star places a tap generator in the reference's
 * `util/if-bandpass-filter`, but that directory is an analog 3rd-order Butterworth LC band-pass, 300–500 MHz
 * (/root/reference/util/if-bandpass-filter/schematic.svg:174-222).  Its pass band, mapped onto a notional
 * 2 GS/s IF stream (0.15–0.25 cycles/sample), is the default band of this digital designer; the formula is
 * docs/SPEC.md §4 (windowed sinc, type-I linear phase, unity gain at the band centre, float64 → float32 once).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "if_fir.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double bpf_sinc(double dT)
{
    if (dT == 0.0)
        return 1.0;
    return sin(M_PI * dT) / (M_PI * dT);
}

static double bpf_window(uint32_t ulWindow, uint32_t ulIndex, uint32_t ulTaps)
{
    const double dA = 2.0 * M_PI * (double)ulIndex / (double)(ulTaps - 1);

    switch (ulWindow)
    {
    case IF_BPF_WINDOW_RECT:
        return 1.0;
    case IF_BPF_WINDOW_HAMMING:
        return 0.54 - 0.46 * cos(dA);
    case IF_BPF_WINDOW_HANN:
        return 0.5 - 0.5 * cos(dA);
    default:
        return 0.42 - 0.5 * cos(dA) + 0.08 * cos(2.0 * dA);
    }
}

uint8_t if_bpf_design(float *pfTaps, uint32_t ulTaps, double dLow, double dHigh, uint32_t ulWindow)
{
    if (!pfTaps || ulTaps < 3 || !(ulTaps & 1) || ulTaps > IF_FIR_MAX_TAPS)
        return 0;
    if (!(dLow >= 0.0) || !(dHigh > dLow) || !(dHigh <= 0.5) || ulWindow > IF_BPF_WINDOW_BLACKMAN)
        return 0;

    double *pdIdeal = (double *)malloc(sizeof(double) * ulTaps);

    if (!pdIdeal)
        return 0;

    const double dMid = (double)(ulTaps - 1) / 2.0;
    const double dCentre = 0.5 * (dLow + dHigh);
    double dGain = 0.0;

    for (uint32_t i = 0; i < ulTaps; i++)
    {
        const double dT = (double)i - dMid;

        pdIdeal[i] = (2.0 * dHigh * bpf_sinc(2.0 * dHigh * dT) - 2.0 * dLow * bpf_sinc(2.0 * dLow * dT)) *
                     bpf_window(ulWindow, i, ulTaps);
        dGain += pdIdeal[i] * cos(2.0 * M_PI * dCentre * dT);
    }

    if (!(fabs(dGain) > 0.0))
    {
        free(pdIdeal);
        return 0;
    }

    for (uint32_t i = 0; i < ulTaps; i++)
        pfTaps[i] = (float)(pdIdeal[i] / dGain);

    free(pdIdeal);

    return 1;
}

/*
 * Complex taps for channel selection: a windowed-sinc low-pass prototype of two-sided bandwidth dBandwidth
 * (cycles/sample) shifted to dCentre, g[k] = hlp[k] * exp(j*2*pi*dCentre*(k - M)), unity gain at dCentre.
 * pfTapsIQ receives ulTaps interleaved (re, im) float32 pairs.  BUILD-DEFINED like if_bpf_design().
 */
uint8_t if_bpf_design_complex(float *pfTapsIQ, uint32_t ulTaps, double dCentre, double dBandwidth, uint32_t ulWindow)
{
    if (!pfTapsIQ || ulTaps < 3 || !(ulTaps & 1) || ulTaps > IF_FIR_MAX_TAPS)
        return 0;
    if (!(dBandwidth > 0.0) || !(dBandwidth <= 1.0) || !(fabs(dCentre) <= 0.5) || ulWindow > IF_BPF_WINDOW_BLACKMAN)
        return 0;

    double *pdProto = (double *)malloc(sizeof(double) * ulTaps);

    if (!pdProto)
        return 0;

    const double dMid = (double)(ulTaps - 1) / 2.0;
    const double dCut = 0.5 * dBandwidth;
    double dGain = 0.0;

    for (uint32_t i = 0; i < ulTaps; i++)
    {
        const double dT = (double)i - dMid;

        pdProto[i] = 2.0 * dCut * bpf_sinc(2.0 * dCut * dT) * bpf_window(ulWindow, i, ulTaps);
        dGain += pdProto[i];
    }

    if (!(fabs(dGain) > 0.0))
    {
        free(pdProto);
        return 0;
    }

    for (uint32_t i = 0; i < ulTaps; i++)
    {
        const double dPhase = 2.0 * M_PI * dCentre * ((double)i - dMid);

        pfTapsIQ[2 * i + 0] = (float)(pdProto[i] / dGain * cos(dPhase));
        pfTapsIQ[2 * i + 1] = (float)(pdProto[i] / dGain * sin(dPhase));
    }

    free(pdProto);

    return 1;
}
