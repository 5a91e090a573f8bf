// host_tables_asan.cpp — the overlap-save source's HOST side (table builders, filter-bank routing) under AddressSanitizer +
// UndefinedBehaviorSanitizer on the CPU (tests/test_host.py compiles if_fir_fft.hip with `hipcc --offload-host-only -fsanitize=...`
// and links this file; GPU sanitizers are not available on the pool).  Every table is built into a heap buffer of exactly its
// documented size, so a write past the end lands in a red zone.  Test infrastructure only.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "if_fir_kernels.h"
int main()
{
    using namespace if_fir;
    std::vector<float> taps(2 * 4096);
    for (size_t i = 0; i < taps.size(); i++)
        taps[i] = (float)((int)(i * 2654435761u % 2001) - 1000) / 1000.0f;
    // exact-size buffers: an out-of-bounds write of a table builder lands in ASan's red zone
    const int Ts[] = {1, 2, 63, 127, 255, 256, 257, 1023, 1025, 2047, 3073, 4096};
    for (int T : Ts)
        for (int ct = 0; ct < 2; ct++)
        {
            for (int D : {1, 2, 4})
                for (int full = 0; full < 2; full++)
                {
                    float *t = (float *)malloc(sizeof(float) * FFT_TABLE_FLOATS);
                    fft_build_tables(taps.data(), T > 2049 ? 2048 : T, ct, D, 12345u, 1.0, t, 0, full, 0);
                    free(t);
                }
            if (T <= 3073)
                for (int bank : {8, 16})
                    for (int par = 0; par < 2; par++)
                    {
                        float *t = (float *)malloc(sizeof(float) * FFT_TABLE_FLOATS);
                        fft_build_tables(taps.data(), T, ct, bank, 777u, 1.0, t, bank, 0, par);
                        free(t);
                    }
            int F = 0, sub = 0, ovlr = 0;
            if (fft_odd_tail(T, 3, &F, &sub, &ovlr))
            {
                float *t = (float *)malloc(sizeof(float) * fft_odd_table_floats(F));
                fft_build_tables_odd(taps.data(), T, ct, F, 99u, 1.0, t);
                free(t);
            }
        }
    for (uint32_t k = 1; k <= 16; k++)
        for (uint32_t seed = 0; seed < 200; seed++)
        {
            uint32_t slots[16], pm[2], rest;
            for (uint32_t c = 0; c < k; c++)
                slots[c] = (seed * 2654435761u + c * 40503u) >> 28;
            fft_bank8_plan(slots, k, true, pm, &rest);
        }
    for (int d = -3; d < 80; d++)
        (void)fft_bank_tail(d, d & 1);
    printf("host table builders: clean\n");
    return 0;
}
// stand-ins for the per-overlap-length units' launchers (never called by the harness)
#include <hip/hip_runtime.h>
namespace if_fir
{
template <int ROWS> hipError_t launch_fft_rows(const LaunchArgs &);
template <> hipError_t launch_fft_rows<4>(const LaunchArgs &) { return hipErrorUnknown; }
template <> hipError_t launch_fft_rows<8>(const LaunchArgs &) { return hipErrorUnknown; }
template <> hipError_t launch_fft_rows<16>(const LaunchArgs &) { return hipErrorUnknown; }
template <> hipError_t launch_fft_rows<32>(const LaunchArgs &) { return hipErrorUnknown; }
template <> hipError_t launch_fft_rows<48>(const LaunchArgs &) { return hipErrorUnknown; }
hipError_t launch_fft_two_partitions(const LaunchArgs &) { return hipErrorUnknown; }
hipError_t launch_fft_odd(const LaunchArgs &) { return hipErrorUnknown; }
}
