/* call_overhead.c — host-side cost of one if_fir_process_device call (development check, plain C on the C ABI):
 * issue time per call and end-to-end time per call for small inputs.   usage: call_overhead [log2 samples] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "if_fir.h"

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
    const uint32_t ulLog2 = argc > 1 ? (uint32_t)atoi(argv[1]) : 16;
    const uint64_t ullN = 1ull << ulLog2;
    float afTaps[255];
    if_fir_ctx_t *pFir = NULL;
    void *pIn = NULL, *pOut = NULL;
    uint64_t ullOut = 0;

    if(!if_bpf_design(afTaps, 255, 0.15, 0.25, IF_BPF_WINDOW_BLACKMAN) || !if_fir_init(&pFir, afTaps, 255, 4, 0, 0))
    {
        fprintf(stderr, "init: %s\n", if_fir_last_error(NULL));
        return 1;
    }
    if(!if_fir_dev_alloc(pFir, &pIn, 8 * ullN) || !if_fir_dev_alloc(pFir, &pOut, 2 * ullN + 64) || !if_fir_synth_device(pFir, pIn, 0, ullN, 0))
        return 1;
    for(int i = 0; i < 200; i++)
        if_fir_process_device(pFir, pIn, pOut, ullN, &ullOut);
    if_fir_synchronize(pFir);
    const int reps = 5000;
    const double t0 = now();
    for(int i = 0; i < reps; i++)
        if_fir_process_device(pFir, pIn, pOut, ullN, &ullOut);
    const double t1 = now();
    if_fir_synchronize(pFir);
    const double t2 = now();
    printf("n = 2^%u: host issue %.2f us per call, end to end %.2f us per call\n", ulLog2, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6);
    if_fir_dev_free(pFir, pIn);
    if_fir_dev_free(pFir, pOut);
    if_fir_destroy(pFir);
    return 0;
}
