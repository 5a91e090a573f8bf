/*
 * if_fir_oracle.c — CPU ORACLE for the IF-chain FIR path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product library (libif_fir.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference vankxr/qo-100-tools contains no implementation of this path.
 * `util/if-bandpass-filter/` is an analog LC filter drawing
 * (/root/reference/util/if-bandpass-filter/schematic.svg:174-222), `software/opi-rf-manager/index.js`
 * (:3148-3535, :4527-4534) never touches IQ samples, and the repository holds no tests or golden
 * vectors.  There is therefore no reference file:line this restatement can
 * follow; it follows docs/SPEC.md (BUILD-DEFINED) and is cross-checked in the
 * dev container against numpy/scipy (third-party tools, not the reference) by tests/test_oracle.py.
 *
 * Build: see oracle/Makefile  (gcc -O3 -march=native -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------- */
/* SPEC §4 — windowed-sinc band-pass designer (float64 → float32).  SURVEY.md §8a-1.            */
/* ------------------------------------------------------------------------------------------- */
static double oracle_sinc(double t)
{
    if (t == 0.0)
        return 1.0;
    const double a = M_PI * t;
    return sin(a) / a;
}

/* window ids: 0 rect, 1 hamming, 2 hann, 3 blackman */
ORACLE_API int oracle_bpf_design(float *pfTaps, uint32_t ulTaps, double dLow, double dHigh, uint32_t ulWindow)
{
    if (!pfTaps || ulTaps < 3 || !(ulTaps & 1) || !(dLow >= 0.0) || !(dHigh > dLow) || !(dHigh <= 0.5) || ulWindow > 3)
        return 0;
    double *g = (double *)malloc(sizeof(double) * ulTaps);
    if (!g)
        return 0;
    const double M = (double)(ulTaps - 1) / 2.0;
    const double fc = 0.5 * (dLow + dHigh);
    double gain = 0.0;
    for (uint32_t n = 0; n < ulTaps; n++)
    {
        const double t = (double)n - M;
        const double a = 2.0 * M_PI * (double)n / (double)(ulTaps - 1);
        double w;
        switch (ulWindow)
        {
        case 0: w = 1.0; break;
        case 1: w = 0.54 - 0.46 * cos(a); break;
        case 2: w = 0.5 - 0.5 * cos(a); break;
        default: w = 0.42 - 0.5 * cos(a) + 0.08 * cos(2.0 * a); break;
        }
        g[n] = (2.0 * dHigh * oracle_sinc(2.0 * dHigh * t) - 2.0 * dLow * oracle_sinc(2.0 * dLow * t)) * w;
        gain += g[n] * cos(2.0 * M_PI * fc * t);
    }
    for (uint32_t n = 0; n < ulTaps; n++)
        pfTaps[n] = (float)(g[n] / gain);
    free(g);
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* SPEC §5 — synthetic IQ generator (SplitMix64 noise + 5-periodic two-tone table).             */
/* ------------------------------------------------------------------------------------------- */
static inline uint64_t oracle_splitmix_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

ORACLE_API void oracle_synth_tone_table(float *pfTone10)
{
    for (int i = 0; i < 5; i++)
    {
        const double a1 = 2.0 * M_PI * 0.2 * (double)i, a2 = 2.0 * M_PI * 0.4 * (double)i;
        pfTone10[2 * i + 0] = (float)(0.5 * cos(a1) + 0.5 * cos(a2));
        pfTone10[2 * i + 1] = (float)(0.5 * sin(a1) + 0.5 * sin(a2));
    }
}

ORACLE_API void oracle_synth_iq(float *pfIQ, uint64_t ullFirst, uint64_t ullSamples, uint32_t ulChannel)
{
    float tone[10];
    oracle_synth_tone_table(tone);
    const uint64_t seed = 0x5130303100000000ULL + (uint64_t)ulChannel;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)ullSamples; i++)
    {
        const uint64_t n = ullFirst + (uint64_t)i;
        const uint64_t z = oracle_splitmix_mix(seed + (n + 1) * 0x9E3779B97F4A7C15ULL);
        const float ui = ((float)(uint32_t)(z >> 40) * 0x1p-24f - 0.5f) * 0.5f;
        const float uq = ((float)(uint32_t)((z >> 16) & 0xFFFFFFu) * 0x1p-24f - 0.5f) * 0.5f;
        const uint32_t p = (uint32_t)(n % 5u);
        pfIQ[2 * i + 0] = tone[2 * p + 0] + ui;
        pfIQ[2 * i + 1] = tone[2 * p + 1] + uq;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* SPEC §2 — helpers shared by every FIR variant                                               */
/* ------------------------------------------------------------------------------------------- */
/* sample j of the logical stream (j may be negative: history, most recent sample last) */
static inline void oracle_fetch(const float *pfHist, uint32_t T, const float *pfIn, int64_t j, float *pI, float *pQ)
{
    if (j >= 0)
    {
        *pI = pfIn[2 * j];
        *pQ = pfIn[2 * j + 1];
    }
    else if (pfHist && -j <= (int64_t)(T - 1))
    {
        const int64_t h = (int64_t)(T - 1) + j;
        *pI = pfHist[2 * h];
        *pQ = pfHist[2 * h + 1];
    }
    else
    {
        *pI = 0.0f;
        *pQ = 0.0f;
    }
}

ORACLE_API uint64_t oracle_out_count(uint64_t ullConsumed, uint64_t ullSamples, uint32_t D)
{
    const uint64_t n0 = (D - ullConsumed % D) % D;
    return (ullSamples > n0) ? (ullSamples - n0 + D - 1) / D : 0;
}

/* new history = last T-1 samples of (hist ‖ in) */
ORACLE_API void oracle_update_history(float *pfHistOut, const float *pfHist, uint32_t T, const float *pfIn, uint64_t N)
{
    for (int64_t i = 0; i < (int64_t)T - 1; i++)
    {
        float re, im;
        oracle_fetch(pfHist, T, pfIn, (int64_t)N - (int64_t)(T - 1) + i, &re, &im);
        pfHistOut[2 * i] = re;
        pfHistOut[2 * i + 1] = im;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* SPEC §2/§3 — THE oracle: direct form, float64 accumulation, float64 output.                 */
/* ------------------------------------------------------------------------------------------- */
ORACLE_API uint64_t oracle_fir_c64_f64(const float *pfTaps, uint32_t T, uint32_t D, const float *pfHist,
                                      uint64_t ullConsumed, const float *pfIn, uint64_t N, double *pdOut)
{
    const uint64_t M = oracle_out_count(ullConsumed, N, D);
    const int64_t n0 = (int64_t)((D - ullConsumed % D) % D);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)M; m++)
    {
        const int64_t n = n0 + m * (int64_t)D;
        double ai = 0.0, aq = 0.0;
        for (uint32_t k = 0; k < T; k++)
        {
            float xi, xq;
            oracle_fetch(pfHist, T, pfIn, n - (int64_t)k, &xi, &xq);
            ai += (double)pfTaps[k] * (double)xi;
            aq += (double)pfTaps[k] * (double)xq;
        }
        pdOut[2 * m] = ai;
        pdOut[2 * m + 1] = aq;
    }
    return M;
}

/* real-sample variant (BASELINE.json configs[0]: 127 taps over 2^20 real float samples, CPU plumbing) */
ORACLE_API uint64_t oracle_fir_r32_f64(const float *pfTaps, uint32_t T, const float *pfIn, uint64_t N, double *pdOut)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++)
    {
        double a = 0.0;
        const uint32_t kmax = (n + 1 < (int64_t)T) ? (uint32_t)(n + 1) : T;
        for (uint32_t k = 0; k < kmax; k++)
            a += (double)pfTaps[k] * (double)pfIn[n - k];
        pdOut[n] = a;
    }
    return N;
}

/* ------------------------------------------------------------------------------------------- */
/* SPEC §3 — bit-exact model of the HIP kernels' float32 summation order.                       */
/*   ulSegMode 0: one chain over all taps.                                                      */
/*   ulSegMode 1: contiguous segments of ulSegLen taps, segment s = taps [s·L, (s+1)·L).         */
/*   ulSegMode 2: residue classes, segment s = taps with k mod ulSegLen == s.                    */
/*   ulSegMode 3: tap-split kernel: lane q of a quad owns the taps k = 4j + q (j descending);    */
/*                inside a lane, segments of ulSegLen consecutive j (aligned to multiples of    */
/*                ulSegLen) are added as they complete; the four lane sums are combined        */
/*                pairwise as (q0 + q1) + (q2 + q3)  (two DPP butterfly steps).                  */
/*   Inside a segment taps are visited in DESCENDING k with acc = fmaf(x, h, acc), acc0 = +0;   */
/*   segments are added in the order they complete, i.e. DESCENDING s:                          */
/*   tot = seg[S-1]; tot += seg[S-2]; … ; tot += seg[0]   (plain float adds).                    */
/* ------------------------------------------------------------------------------------------- */
static void oracle_chain_residue(const float *pfTaps, uint32_t T, const float *pfHist, const float *pfIn, int64_t n,
                                 uint32_t q, uint32_t L, float *pI, float *pQ)
{
    /* taps k = 4j + q < T, j descending, segments of L consecutive j aligned to multiples of L */
    float ti = 0.0f, tq = 0.0f, ai = 0.0f, aq = 0.0f;
    int have_tot = 0;
    const int64_t J = ((int64_t)T + 3) / 4;
    for (int64_t j = J - 1; j >= 0; j--)
    {
        const int64_t k = 4 * j + q;
        if (k < (int64_t)T)
        {
            float xi, xq;
            oracle_fetch(pfHist, T, pfIn, n - k, &xi, &xq);
            ai = fmaf(xi, pfTaps[k], ai);
            aq = fmaf(xq, pfTaps[k], aq);
        }
        if (j % (int64_t)L == 0)
        {
            if (have_tot)
            {
                ti += ai;
                tq += aq;
            }
            else
            {
                ti = ai;
                tq = aq;
                have_tot = 1;
            }
            ai = 0.0f;
            aq = 0.0f;
        }
    }
    *pI = ti;
    *pQ = tq;
}

ORACLE_API uint64_t oracle_fir_c64_f32fma(const float *pfTaps, uint32_t T, uint32_t D, const float *pfHist,
                                         uint64_t ullConsumed, const float *pfIn, uint64_t N, float *pfOut,
                                         uint32_t ulSegMode, uint32_t ulSegLen)
{
    const uint64_t M = oracle_out_count(ullConsumed, N, D);
    const int64_t n0 = (int64_t)((D - ullConsumed % D) % D);
    uint32_t nseg = 1;
    if (ulSegMode == 1)
        nseg = (T + ulSegLen - 1) / ulSegLen;
    else if (ulSegMode == 2)
        nseg = ulSegLen;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)M; m++)
    {
        const int64_t n = n0 + m * (int64_t)D;
        float ti = 0.0f, tq = 0.0f;
        if (ulSegMode == 3)
        {
            float qi[4], qq[4];
            for (uint32_t q = 0; q < 4; q++)
                oracle_chain_residue(pfTaps, T, pfHist, pfIn, n, q, ulSegLen, &qi[q], &qq[q]);
            pfOut[2 * m] = (qi[0] + qi[1]) + (qi[2] + qi[3]);
            pfOut[2 * m + 1] = (qq[0] + qq[1]) + (qq[2] + qq[3]);
            continue;
        }
        for (int64_t ss = (int64_t)nseg - 1; ss >= 0; ss--)
        {
            const uint32_t s = (uint32_t)ss;
            float ai = 0.0f, aq = 0.0f;
            for (int64_t k = (int64_t)T - 1; k >= 0; k--)
            {
                if (ulSegMode == 1 && (uint32_t)k / ulSegLen != s)
                    continue;
                if (ulSegMode == 2 && (uint32_t)k % ulSegLen != s)
                    continue;
                float xi, xq;
                oracle_fetch(pfHist, T, pfIn, n - k, &xi, &xq);
                ai = fmaf(xi, pfTaps[k], ai);
                aq = fmaf(xq, pfTaps[k], aq);
            }
            if (s == nseg - 1)
            {
                ti = ai;
                tq = aq;
            }
            else
            {
                ti += ai;
                tq += aq;
            }
        }
        pfOut[2 * m] = ti;
        pfOut[2 * m + 1] = tq;
    }
    return M;
}

/* ------------------------------------------------------------------------------------------- */
/* Timed CPU baseline ("port"): float32 accumulate, OpenMP over output blocks, vectorised       */
/* across outputs (no reassociation needed).  History must be contiguous in front of the        */
/* input for speed, so the caller passes a buffer that already holds T-1 history samples        */
/* before pfIn (pfIn - 2*(T-1) is readable).  Decimating path de-interleaves the D polyphase     */
/* branches of each block first so the inner loops stay unit-stride.                            */
/* ------------------------------------------------------------------------------------------- */
#define ORACLE_BLK 512 /* outputs per block */

ORACLE_API uint64_t oracle_fir_c64_f32_omp(const float *pfTaps, uint32_t T, uint32_t D, const float *pfInWithHist,
                                          uint64_t N, float *pfOut, int32_t lThreads)
{
    /* pfInWithHist points at history sample -(T-1); stream sample 0 is at pfInWithHist + 2*(T-1); phase 0 */
    const float *x0 = pfInWithHist + 2 * (size_t)(T - 1);
    const uint64_t M = (N + D - 1) / D;
    const int64_t nblk = (int64_t)((M + ORACLE_BLK - 1) / ORACLE_BLK);
#ifdef _OPENMP
    if (lThreads > 0)
        omp_set_num_threads(lThreads);
#endif
#pragma omp parallel
    {
        /* per-thread scratch: D phase streams, each (ORACLE_BLK + ceil(T/D)) complex samples */
        const uint32_t J = (T + D - 1) / D; /* taps per phase */
        const size_t plen = (size_t)ORACLE_BLK + J;
        float *ph = (D > 1) ? (float *)aligned_alloc(64, ((2 * plen * D * sizeof(float)) + 63) / 64 * 64) : NULL;
        float acc[2 * ORACLE_BLK] __attribute__((aligned(64)));
#pragma omp for schedule(static)
        for (int64_t b = 0; b < nblk; b++)
        {
            const int64_t m0 = b * ORACLE_BLK;
            const int64_t mb = ((int64_t)M - m0 < ORACLE_BLK) ? (int64_t)M - m0 : ORACLE_BLK;
            for (int64_t i = 0; i < 2 * mb; i++)
                acc[i] = 0.0f;
            if (D == 1)
            {
                for (int64_t k = (int64_t)T - 1; k >= 0; k--)
                {
                    const float h = pfTaps[k];
                    const float *xs = x0 + 2 * (m0 - k);
                    for (int64_t i = 0; i < 2 * mb; i++)
                        acc[i] += h * xs[i];
                }
            }
            else
            {
                /* phase p stream: xp[i] = x[D*(m0 + i - (J-1)) - p], i = 0 .. mb+J-2 ; y[m0+i'] = Σ_p Σ_j h[Dj+p]·xp[i'+J-1-j] */
                for (uint32_t p = 0; p < D; p++)
                {
                    float *xp = ph + 2 * plen * p;
                    for (int64_t i = 0; i < mb + (int64_t)J - 1; i++)
                    {
                        const int64_t s = (int64_t)D * (m0 + i - (int64_t)(J - 1)) - (int64_t)p;
                        if (s < -(int64_t)(T - 1))
                        {
                            xp[2 * i] = 0.0f;
                            xp[2 * i + 1] = 0.0f;
                        }
                        else
                        {
                            xp[2 * i] = x0[2 * s];
                            xp[2 * i + 1] = x0[2 * s + 1];
                        }
                    }
                }
                for (uint32_t p = 0; p < D; p++)
                {
                    const float *xp = ph + 2 * plen * p;
                    for (int64_t j = (int64_t)J - 1; j >= 0; j--)
                    {
                        const uint32_t k = (uint32_t)(D * j + p);
                        if (k >= T)
                            continue;
                        const float h = pfTaps[k];
                        const float *xs = xp + 2 * ((int64_t)J - 1 - j);
                        for (int64_t i = 0; i < 2 * mb; i++)
                            acc[i] += h * xs[i];
                    }
                }
            }
            memcpy(pfOut + 2 * m0, acc, sizeof(float) * 2 * (size_t)mb);
        }
        free(ph);
    }
    return M;
}

/* ------------------------------------------------------------------------------------------- */
/* Complex taps (one-sided band-pass / channel selection): y[n] = Σ g[k]·x[n-k], g complex.     */
/* pfTapsIQ holds T interleaved (re, im) float32 pairs.                                         */
/* ------------------------------------------------------------------------------------------- */
/* low-pass prototype of bandwidth dBandwidth (cycles/sample, two-sided) shifted to dCentre: g[k] = hlp[k]·e^{j2π·fc·(k-M)} */
ORACLE_API int oracle_bpf_design_complex(float *pfTapsIQ, uint32_t ulTaps, double dCentre, double dBandwidth, uint32_t ulWindow)
{
    if (!pfTapsIQ || ulTaps < 3 || !(ulTaps & 1) || !(dBandwidth > 0.0) || !(dBandwidth <= 1.0) || !(fabs(dCentre) <= 0.5) || ulWindow > 3)
        return 0;
    const double M = (double)(ulTaps - 1) / 2.0, fc = 0.5 * dBandwidth;
    double *g = (double *)malloc(sizeof(double) * ulTaps), gain = 0.0;
    if (!g)
        return 0;
    for (uint32_t n = 0; n < ulTaps; n++)
    {
        const double t = (double)n - M, a = 2.0 * M_PI * (double)n / (double)(ulTaps - 1);
        double w;
        switch (ulWindow)
        {
        case 0: w = 1.0; break;
        case 1: w = 0.54 - 0.46 * cos(a); break;
        case 2: w = 0.5 - 0.5 * cos(a); break;
        default: w = 0.42 - 0.5 * cos(a) + 0.08 * cos(2.0 * a); break;
        }
        g[n] = 2.0 * fc * oracle_sinc(2.0 * fc * t) * w;
        gain += g[n];
    }
    for (uint32_t n = 0; n < ulTaps; n++)
    {
        const double t = (double)n - M, ph = 2.0 * M_PI * dCentre * t;
        pfTapsIQ[2 * n] = (float)(g[n] / gain * cos(ph));
        pfTapsIQ[2 * n + 1] = (float)(g[n] / gain * sin(ph));
    }
    free(g);
    return 1;
}

ORACLE_API uint64_t oracle_fir_c64_ctaps_f64(const float *pfTapsIQ, uint32_t T, uint32_t D, const float *pfHist,
                                            uint64_t ullConsumed, const float *pfIn, uint64_t N, double *pdOut)
{
    const uint64_t M = oracle_out_count(ullConsumed, N, D);
    const int64_t n0 = (int64_t)((D - ullConsumed % D) % D);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)M; m++)
    {
        const int64_t n = n0 + m * (int64_t)D;
        double ar = 0.0, ai = 0.0;
        for (uint32_t k = 0; k < T; k++)
        {
            float xr, xi;
            oracle_fetch(pfHist, T, pfIn, n - (int64_t)k, &xr, &xi);
            const double hr = pfTapsIQ[2 * k], hi = pfTapsIQ[2 * k + 1];
            ar += (double)xr * hr - (double)xi * hi;
            ai += (double)xr * hi + (double)xi * hr;
        }
        pdOut[2 * m] = ar;
        pdOut[2 * m + 1] = ai;
    }
    return M;
}

/* NCO + FIR by the definition (docs/SPEC.md §3.2): every input sample with absolute index a is multiplied by         */
/* exp(-j*2*pi*((P*a) mod 2^32)/2^32) in float64, then filtered (real or complex float32 taps) and decimated.        */
/* Samples before the start of the stream (a < 0) are zero, so the history is only consulted for a >= 0.            */
ORACLE_API uint64_t oracle_fir_c64_nco_f64(const float *pfTaps, uint32_t T, uint32_t bComplexTaps, uint32_t D,
                                          const float *pfHist, uint64_t ullConsumed, const float *pfIn, uint64_t N,
                                          uint32_t ulPhaseWord, double *pdOut)
{
    const uint64_t M = oracle_out_count(ullConsumed, N, D);
    const int64_t n0 = (int64_t)((D - ullConsumed % D) % D);
    const int64_t lead = (int64_t)T - 1;
    double *pdMixed = (double *)malloc(sizeof(double) * 2 * (size_t)(N + (uint64_t)lead + 1));

    if (!pdMixed)
        return 0;
#pragma omp parallel for schedule(static)
    for (int64_t j = -lead; j < (int64_t)N; j++)
    {
        float xr, xi;
        const int64_t a = (int64_t)ullConsumed + j;
        oracle_fetch(pfHist, T, pfIn, j, &xr, &xi);
        if (a < 0)
            xr = xi = 0.0f;
        const uint32_t ph = (uint32_t)((uint64_t)ulPhaseWord * (uint64_t)a);
        const double ang = -6.283185307179586476925286766559 * ((double)ph / 4294967296.0);
        const double c = cos(ang), s = sin(ang);
        pdMixed[2 * (j + lead)] = (double)xr * c - (double)xi * s;
        pdMixed[2 * (j + lead) + 1] = (double)xr * s + (double)xi * c;
    }
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)M; m++)
    {
        const int64_t n = n0 + m * (int64_t)D;
        double ar = 0.0, ai = 0.0;
        for (uint32_t k = 0; k < T; k++)
        {
            const double xr = pdMixed[2 * (n - (int64_t)k + lead)], xi = pdMixed[2 * (n - (int64_t)k + lead) + 1];
            const double hr = bComplexTaps ? pfTaps[2 * k] : pfTaps[k], hi = bComplexTaps ? pfTaps[2 * k + 1] : 0.0;
            ar += xr * hr - xi * hi;
            ai += xr * hi + xi * hr;
        }
        pdOut[2 * m] = ar;
        pdOut[2 * m + 1] = ai;
    }
    free(pdMixed);
    return M;
}

/* float32 order model of the generic kernel with complex taps: descending k, segments of L taps [sL,(s+1)L) added  */
/* as they complete; per tap  re = fma(-xi, hi, fma(xr, hr, re)),  im = fma(xi, hr, fma(xr, hi, im)).                 */
ORACLE_API uint64_t oracle_fir_c64_ctaps_f32fma(const float *pfTapsIQ, uint32_t T, uint32_t D, const float *pfHist,
                                               uint64_t ullConsumed, const float *pfIn, uint64_t N, float *pfOut, uint32_t L)
{
    const uint64_t M = oracle_out_count(ullConsumed, N, D);
    const int64_t n0 = (int64_t)((D - ullConsumed % D) % D);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)M; m++)
    {
        const int64_t n = n0 + m * (int64_t)D;
        float tr = 0.0f, ti = 0.0f, ar = 0.0f, ai = 0.0f;
        const int64_t top = ((int64_t)T - 1) / L;
        for (int64_t k = (int64_t)T - 1; k >= 0; k--)
        {
            float xr, xi;
            oracle_fetch(pfHist, T, pfIn, n - k, &xr, &xi);
            const float hr = pfTapsIQ[2 * k], hi = pfTapsIQ[2 * k + 1];
            if (k == (int64_t)T - 1 || k % L == L - 1)
            {
                ar = 0.0f;
                ai = 0.0f;
            }
            ar = fmaf(-xi, hi, fmaf(xr, hr, ar));
            ai = fmaf(xi, hr, fmaf(xr, hi, ai));
            if (k % L == 0)
            {
                if (k / L == top)
                {
                    tr = ar;
                    ti = ai;
                }
                else
                {
                    tr += ar;
                    ti += ai;
                }
            }
        }
        pfOut[2 * m] = tr;
        pfOut[2 * m + 1] = ti;
    }
    return M;
}

ORACLE_API int32_t oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* error metrics of a float32 result against the float64 oracle: out[0] = ||y-ŷ||2/||ŷ||2, out[1] = max|y-ŷ|/max|ŷ| */
ORACLE_API void oracle_err_metrics(const float *pfY, const double *pdRef, uint64_t ullFloats, double *pdOut2)
{
    double se = 0.0, sr = 0.0, me = 0.0, mr = 0.0;
#pragma omp parallel for reduction(+ : se, sr) reduction(max : me, mr) schedule(static)
    for (int64_t i = 0; i < (int64_t)ullFloats; i++)
    {
        const double e = (double)pfY[i] - pdRef[i];
        se += e * e;
        sr += pdRef[i] * pdRef[i];
        if (fabs(e) > me)
            me = fabs(e);
        if (fabs(pdRef[i]) > mr)
            mr = fabs(pdRef[i]);
    }
    pdOut2[0] = (sr > 0.0) ? sqrt(se / sr) : sqrt(se);
    pdOut2[1] = (mr > 0.0) ? me / mr : me;
}
