// microbenchmark: LDS-DMA ring read bandwidth as a function of the row stride of a [256 rows][128 B] tile
#include <hip/hip_runtime.h>
#include <stdint.h>
#define PH_LDS __attribute__((address_space(3)))
template <int NBUF, int NW, int AUX>
__global__ __launch_bounds__(NW * 64) void k_dmabw(const char* __restrict__ base, int64_t row_stride, int64_t tile_stride,
                                                   int tiles_per_wg, int64_t wg_stride, uint32_t* out, int nmfma, int nlds) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int DPW = 32 / NW;
    const char* wbase = base + (int64_t)blockIdx.x * wg_stride;
    const uint32_t lane_off = (uint32_t)((lane >> 3) * row_stride + (lane & 7) * 16);
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int jj = wave + NW * k;
            const char* ub = wbase + (int64_t)t * tile_stride + (int64_t)(jj * 8) * row_stride;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + lane_off),
                                             (PH_LDS void*)(lds + buf * 32768 + jj * 1024), 16, 0, AUX);
        }
    };
    int ti = 0;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d) if (ti < tiles_per_wg) { issue(ti, d); ++ti; }
    int cur = 0;
    uint32_t acc = 0;
    typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
    typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
    f32x16_t c = {0};
    bf16x8_t av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (__bf16)(float)(tid + e); bv[e] = (__bf16)(float)(lane * e); }
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int younger = (tiles_per_wg - 1 - t) < (NBUF - 2) ? (tiles_per_wg - 1 - t) : (NBUF - 2);
        if (younger >= 2 && NBUF >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ti < tiles_per_wg) { int nb = cur + NBUF - 1; if (nb >= NBUF) nb -= NBUF; issue(ti, nb); ++ti; }
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(PH_LDS char*)(lds + cur * 32768 + tid * 4)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc ^= v;
        for (int k = 0; k < nlds; ++k) { uint32_t w2[2]; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(*(unsigned long long*)w2) : "v"((uint32_t)(uintptr_t)(PH_LDS char*)(lds + cur * 32768 + ((tid * 8 + k * 2048) & 32767)))); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc ^= w2[0]; }
        for (int k = 0; k < nmfma; ++k) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    if (acc == 0x9E3779B9u || c[0] == 12345.f) out[0] = acc;
}
extern "C" int dmabw(const void* base, int64_t row_stride, int64_t tile_stride, int tiles_per_wg, int64_t wg_stride, int wgs,
                     int nbuf, void* out, void* stream, int nmfma, int nlds) {
    hipStream_t s = (hipStream_t)stream;
    if (nbuf == 4) {
        hipFuncSetAttribute((const void*)k_dmabw<4, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        hipLaunchKernelGGL((k_dmabw<4, 8, 0>), dim3(wgs), dim3(512), 4 * 32768, s, (const char*)base, row_stride, tile_stride, tiles_per_wg, wg_stride, (uint32_t*)out, nmfma, nlds);
    } else if (nbuf == 2) {
        hipFuncSetAttribute((const void*)k_dmabw<2, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
        hipLaunchKernelGGL((k_dmabw<2, 8, 0>), dim3(wgs), dim3(512), 2 * 32768, s, (const char*)base, row_stride, tile_stride, tiles_per_wg, wg_stride, (uint32_t*)out, nmfma, nlds);
    } else {   // nbuf == 418: ring of 4 with the nt | sc1 cache policy the product kernels use
        hipFuncSetAttribute((const void*)k_dmabw<4, 8, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        hipLaunchKernelGGL((k_dmabw<4, 8, 18>), dim3(wgs), dim3(512), 4 * 32768, s, (const char*)base, row_stride, tile_stride, tiles_per_wg, wg_stride, (uint32_t*)out, nmfma, nlds);
    }
    return (int)hipGetLastError();
}
