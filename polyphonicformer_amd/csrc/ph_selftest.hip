// Hardware self tests of the fragment layouts every other kernel in this library assumes.
#include "ph_common.h"

__global__ void k_selftest_mfma16(const uint16_t* a, const uint16_t* bt, float* d) {
    const int l = threadIdx.x;
    uint4 av = *(const uint4*)(a + (l & 15) * 32 + (l >> 4) * 8);
    uint4 bv = *(const uint4*)(bt + (l & 15) * 32 + (l >> 4) * 8);
    f32x4_t acc = {0, 0, 0, 0};
    acc = mfma16(av, bv, acc);
    for (int r = 0; r < 4; ++r) d[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

__global__ void k_selftest_mfma32(const uint16_t* a, const uint16_t* bt, float* d) {
    const int l = threadIdx.x;
    uint4 av = *(const uint4*)(a + (l & 31) * 16 + (l >> 5) * 8);
    uint4 bv = *(const uint4*)(bt + (l & 31) * 16 + (l >> 5) * 8);
    f32x16_t acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma32(av, bv, acc);
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

// src: [16 rows][16 cols]; group g (16 lanes) reads rows 4g..4g+3; out[lane][j] must equal
// src[4g + j][lane & 15].
__global__ void k_selftest_trread(const uint16_t* src, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[16 * 24];   // padded row stride 24 (48 B)
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[(i >> 4) * 24 + (i & 15)] = src[i];
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    uint2 v = lds_read_tr16(&lds[(4 * g + (i >> 2)) * 24 + (i & 3) * 4]);
    out[l * 4 + 0] = v.x & 0xFFFF;
    out[l * 4 + 1] = v.x >> 16;
    out[l * 4 + 2] = v.y & 0xFFFF;
    out[l * 4 + 3] = v.y >> 16;
}

// pure streaming read: the achievable HBM read bandwidth on this box (the yardstick for the read-dominated
// pooling / conv kernels).  16 B per lane, UNROLL independent loads in flight per lane.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_selftest_readbw(const uint4* __restrict__ p, int64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9E3779B9u) out[0] = acc;      // practically never: keeps the loads alive
}

extern "C" int ph_selftest_readbw(const void* p, int64_t bytes, int blocks, void* out, void* stream) {
    hipLaunchKernelGGL(k_selftest_readbw<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)p, bytes / 16,
                       (uint32_t*)out);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_selftest_mfma16(const uint16_t* a, const uint16_t* bt, float* d, void* stream) {
    hipLaunchKernelGGL(k_selftest_mfma16, dim3(1), dim3(64), 0, (hipStream_t)stream, a, bt, d);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
extern "C" int ph_selftest_mfma32(const uint16_t* a, const uint16_t* bt, float* d, void* stream) {
    hipLaunchKernelGGL(k_selftest_mfma32, dim3(1), dim3(64), 0, (hipStream_t)stream, a, bt, d);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
extern "C" int ph_selftest_trread(const uint16_t* src, uint16_t* out, void* stream) {
    hipLaunchKernelGGL(k_selftest_trread, dim3(1), dim3(64), 0, (hipStream_t)stream, src, out);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// Occupies `blocks` workgroup slots (each with `lds_bytes` of LDS) for `microseconds`: the "another kernel holds CUs" condition
// under which ph_khead_onepass's persistent grid cannot become resident in time (tests/test_gpu_khead1.py forces its time-out
// and the in-call fallback with this).
__global__ __launch_bounds__(64) void k_selftest_hog(unsigned long long ticks, unsigned* out) {
    extern __shared__ unsigned hog_lds[];
    hog_lds[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0) atomicAdd(out + 1, 1u);            // out[1]: blocks that hold their slot by now (the host waits for them)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(16); ++n; }
    if (n == 0xFFFFFFFFu) out[0] = hog_lds[(threadIdx.x + 1) & 63];      // practically never: keeps the LDS allocation alive
}
// scratch4: 4 device words; word 1 counts the blocks that have started (zero it before the call)
extern "C" int ph_selftest_hog(int blocks, int lds_bytes, int microseconds, void* scratch4, void* stream) {
    PH_CHECK_ARG(blocks > 0 && lds_bytes >= 256 && lds_bytes <= 160 * 1024 && microseconds > 0 && scratch4, "bad argument");
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)k_selftest_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL(k_selftest_hog, dim3(blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)microseconds * 100ull, (unsigned*)scratch4);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
