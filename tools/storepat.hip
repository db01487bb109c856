// Store-pattern yardstick for the fused final stage (round 6): 256 workgroups x 5 waves write a [B][N][2H][2W] fp16 tensor the way
// k_dynconv_up2m's consumer waves do -- wave = 32 query planes, per step 2 output rows x RUN bytes of each plane -- with RUN = 128 (the
// kernel's: one 64-pixel half), 256, 512, 1024 (a whole output row).  Prints TB/s of pure writes per pattern (no reads, no compute).
//   hipcc --offload-arch=gfx950 -O3 tools/storepat.hip -o tools/storepat && tools/storepat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int RUN, bool NT>
__global__ __launch_bounds__(320) void k_store(uint16_t* __restrict__ out, int B, int N, int H2, int W2, int rows_per_wg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)H2 * W2;
    const int total_rows = B * (H2 / 2);                       // low-resolution image rows over all frames
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, total_rows);
    constexpr int LPR = RUN / 16;                               // lanes per run
    constexpr int RPI = 64 / LPR;                               // runs per store instruction
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = {(unsigned)lane, (unsigned)wave, blockIdx.x, 0x3c003c00u};
    for (int r = r0; r < r1; ++r) {
        const int b = r / (H2 / 2), y = r - b * (H2 / 2);
        for (int x0 = 0; x0 < W2 * 2; x0 += RUN) {              // byte offset inside an output row
            // 32 queries x 2 output rows = 64 runs of RUN bytes -> 64 / RPI instructions
#pragma unroll
            for (int i = 0; i < 64 / RPI; ++i) {
                const int run = i * RPI + lane / LPR;           // 0 .. 63
                const int q = wave * 32 + (run >> 1), orow = 2 * y + (run & 1);
                if (q < N) {
                    char* p = (char*)(out + ((int64_t)b * N + q) * plane + (int64_t)orow * W2) + x0 + (lane % LPR) * 16;
                    if (NT) __builtin_nontemporal_store(v, (v4u*)p);
                    else *(v4u*)p = v;
                }
            }
        }
    }
}

template <int RUN, bool NT> static void run(uint16_t* d, int B, int N, int H2, int W2, const char* name) {
    const int total_rows = B * (H2 / 2), wgs = 256, rpw = (total_rows + wgs - 1) / wgs;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_store<RUN, NT>), dim3(wgs), dim3(320), 0, 0, d, B, N, H2, W2, rpw);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k_store<RUN, NT>), dim3(wgs), dim3(320), 0, 0, d, B, N, H2, W2, rpw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)B * N * H2 * W2 * 2;
    printf("%-28s run %4d B  %s  %7.1f us per launch  %.2f TB/s\n", name, RUN, NT ? "nt    " : "cached", ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 24, N = 153, H2 = 256, W2 = 512;
    uint16_t* d;
    const size_t bytes = (size_t)B * N * H2 * W2 * 2;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("[B=%d][N=%d][%d][%d] fp16 = %.2f GB, 256 workgroups x 5 waves, wave = 32 planes x 2 rows per step\n", B, N, H2, W2, bytes / 1e9);
    run<128, true>(d, B, N, H2, W2, "k_dynconv_up2m's pattern");
    run<256, true>(d, B, N, H2, W2, "two halves per plane");
    run<512, true>(d, B, N, H2, W2, "half an output row");
    run<1024, true>(d, B, N, H2, W2, "a whole output row");
    run<128, false>(d, B, N, H2, W2, "k_dynconv_up2m's pattern");
    run<256, false>(d, B, N, H2, W2, "two halves per plane");
    run<1024, false>(d, B, N, H2, W2, "a whole output row");
    hipFree(d);
    return 0;
}
