// A7 -- masked pooling  u[n][c] = sum_hw M[n][hw] * feat[c][hw]   (kernel_update_head.py:241-242)
//
// NT GEMM with M = Npad query rows, N = 256 channels per map, K = HW pixels, split-K over
// `nsplit` pixel ranges.  One workgroup = (pixel range, 128-channel group, frame); its 4 waves own
// 32 channels each (one 32x32x16 MFMA column tile) and ALL query rows:
//   B operand (features): straight from HBM into VGPRs -- lane (channel j, half g) loads 64
//     contiguous pixels (8 x 16 B); MFMA step t consumes its t-th 16-byte piece, so k-slot
//     (g, e) of step t <-> pixel 128*chunk + 64*g + 8*t + e.  Every feature byte is read once.
//   A operand (mask bits -> {0,1} bf16): expanded ONCE per workgroup per 128-pixel chunk into an
//     LDS tile [Npad][128 px] (row stride 272 B: conflict-free ds_read_b128) and shared by the 4
//     waves; {0,1} is exact in bf16, so in split precision only the feature operand has two planes.
// Roofline: HBM: 2*256*HWp*2 B of features per frame vs 2*Npad*512*HWp flop.
#include "ph_common.h"

constexpr int POOL_CHUNK = 128;                 // pixels per LDS mask tile
constexpr int POOL_LDA = POOL_CHUNK + 8;        // bf16 elements per LDS row (272 B)

__device__ __forceinline__ uint4 expand8(uint32_t byte) {
    // 8 mask bits -> 8 bf16 {0, 1.0}
    uint32_t r[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t lo = (byte >> (2 * p)) & 1u, hi = (byte >> (2 * p + 1)) & 1u;
        r[p] = lo * 0x3F80u + hi * 0x3F800000u;
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

template <int PA /*feature planes: 1 or 2*/, int NRT /*Npad/32*/>
__global__ __launch_bounds__(256) void k_pool(const uint16_t* __restrict__ xplanes, const uint16_t* __restrict__ dplanes,
                                              const uint32_t* __restrict__ bits, float* __restrict__ partial,
                                              int B, int64_t HWp, int nsplit) {
    constexpr int Npad = NRT * 32;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [Npad][POOL_LDA]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, cg = blockIdx.y, b = blockIdx.z;
    const int map = cg >> 1;
    const int ch = (cg & 1) * 128 + wave * 32 + (lane & 31);
    const int g = lane >> 5;
    const uint16_t* feat = (map == 0 ? xplanes : dplanes);
    const int64_t plane_stride = (int64_t)B * PH_C * HWp;
    const uint16_t* frow = feat + ((int64_t)b * PH_C + ch) * HWp;
    const int64_t words_per_row = HWp / 32;
    const uint32_t* brow = bits + (int64_t)b * Npad * words_per_row;

    const int nchunks = (int)(HWp / POOL_CHUNK);
    const int c0 = (int)((int64_t)split * nchunks / nsplit), c1 = (int)((int64_t)(split + 1) * nchunks / nsplit);

    f32x16_t acc[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

    for (int c = c0; c < c1; ++c) {
        // (1) feature fragments for this chunk: 8 x 16 B per lane per plane, issued first
        uint4 xf[PA][8];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const uint16_t* s = frow + p * plane_stride + (int64_t)c * POOL_CHUNK + g * 64;
#pragma unroll
            for (int t = 0; t < 8; ++t) xf[p][t] = *(const uint4*)(s + t * 8);
        }
        // (2) expand this chunk's mask words into the LDS A tile (all 256 threads)
        for (int wi = tid; wi < Npad * 4; wi += 256) {
            const int row = wi >> 2, wq = wi & 3;
            const uint32_t w = brow[row * words_per_row + c * 4 + wq];
            uint16_t* d = lds + row * POOL_LDA + wq * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(uint4*)(d + q * 8) = expand8((w >> (8 * q)) & 0xFFu);
        }
        __syncthreads();
        // (3) MFMA: every row tile against the 8 k-steps of this chunk
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const uint16_t* arow = lds + (rt * 32 + (lane & 31)) * POOL_LDA + g * 64;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint4 a = *(const uint4*)(arow + t * 8);
#pragma unroll
                for (int p = 0; p < PA; ++p) acc[rt] = mfma32(a, xf[p][t], acc[rt]);
            }
        }
        __syncthreads();
    }

    // epilogue: partial[b][split][row][map*256 + ch]
    float* out = partial + (((int64_t)b * nsplit + split) * Npad) * 512 + map * 256 + ch;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            out[(int64_t)row * 512] = acc[rt][r];
        }
}

template <int PA, int NRT>
static void launch_pool(const uint16_t* x, const uint16_t* d, const uint32_t* bits, float* partial, int B, int64_t HWp,
                        int nsplit, hipStream_t s) {
    const size_t lds = (size_t)NRT * 32 * POOL_LDA * sizeof(uint16_t);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)k_pool<PA, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    hipLaunchKernelGGL((k_pool<PA, NRT>), dim3(nsplit, d ? 4 : 2, B), dim3(256), lds, s, x, d, bits, partial, B, HWp,
                       nsplit);
}

extern "C" int ph_pool(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, float* partial, int B,
                       int N, int64_t HW, int nsplit, int prec, void* stream) {
    PH_CHECK_ARG(xplanes && bits && partial && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    PH_CHECK_ARG(N <= 256, "at most 256 queries");
    const int64_t HWp = ph_hw_padded(HW);
    PH_CHECK_ARG(nsplit >= 1 && nsplit <= HWp / POOL_CHUNK, "nsplit out of range");
    const int nrt = ph_n_padded(N) / 32;
    hipStream_t s = (hipStream_t)stream;
#define PH_POOL_CASE(R)                                                                         \
    case R:                                                                                     \
        if (prec == PH_PREC_BF16) launch_pool<1, R>(xplanes, dplanes, bits, partial, B, HWp, nsplit, s); \
        else launch_pool<2, R>(xplanes, dplanes, bits, partial, B, HWp, nsplit, s);             \
        break;
    switch (nrt) {
        PH_POOL_CASE(1) PH_POOL_CASE(2) PH_POOL_CASE(3) PH_POOL_CASE(4)
        PH_POOL_CASE(5) PH_POOL_CASE(6) PH_POOL_CASE(7) PH_POOL_CASE(8)
        default: ph_set_error("ph_pool: unsupported N"); return PH_EUNSUPPORTED;
    }
#undef PH_POOL_CASE
    PH_CHECK_LAUNCH();
    return PH_OK;
}
