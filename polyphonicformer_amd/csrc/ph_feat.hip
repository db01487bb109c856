// Boundary format kernels: fp32 NCHW -> bf16 planes, mask logits -> mask bits, x2 bilinear upsample.
// All three are pure HBM streaming kernels: 16 B per lane, grid-stride.
#include "ph_common.h"

// ---------------------------------------------------------------------------------------------
// ingest: src fp32 [rows = B*256][HW] -> planes [P][rows][HWp]; 8 pixels per thread.
template <int P>
__global__ __launch_bounds__(256) void k_ingest(const float* __restrict__ src, uint16_t* __restrict__ planes,
                                                int64_t rows, int64_t HW, int64_t HWp) {
    const int64_t per_row = HWp / 8;
    const int64_t total = rows * per_row;
    const int64_t plane_stride = rows * HWp;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / per_row;
        const int64_t px = (idx - row * per_row) * 8;
        float v[8];
        const float* s = src + row * HW + px;
        if (px + 8 <= HW && ((HW & 3) == 0)) {
            const float4 a = *(const float4*)(s), b = *(const float4*)(s + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (px + e < HW) ? s[e] : 0.f;
        }
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (P == 1) hi[e] = f2bf(v[e]);
            else f2bf_split(v[e], hi[e], lo[e]);
        }
        uint16_t* d = planes + row * HWp + px;
        *(uint4*)d = make_uint4(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7]));
        if (P == 2)
            *(uint4*)(d + plane_stride) =
                make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(lo[4], lo[5]), pack2(lo[6], lo[7]));
    }
}

extern "C" int ph_ingest_features(const float* src, uint16_t* planes, int B, int64_t HW, int prec, void* stream) {
    PH_CHECK_ARG(src && planes && B > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    const int64_t HWp = ph_hw_padded(HW), rows = (int64_t)B * PH_C;
    const int64_t total = rows * (HWp / 8);
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (prec == PH_PREC_BF16)
        hipLaunchKernelGGL(k_ingest<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, planes, rows, HW, HWp);
    else
        hipLaunchKernelGGL(k_ingest<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, planes, rows, HW, HWp);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---------------------------------------------------------------------------------------------
// binarize: logits [B][N][HW] -> bits [B][Npad][HWp/32].  One wave produces two words per step
// with __ballot (lane = pixel).  Rows >= N and pixels >= HW come out 0.
__global__ __launch_bounds__(256) void k_binarize(const float* __restrict__ logits, uint32_t* __restrict__ bits,
                                                  int B, int N, int Npad, int64_t HW, int64_t HWp) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t w64_per_row = HWp / 64;
    const int64_t total = (int64_t)B * Npad * w64_per_row;
    for (int64_t t = wave; t < total; t += nwaves) {
        const int64_t row = t / w64_per_row;          // b*Npad + n
        const int64_t px = (t - row * w64_per_row) * 64 + lane;
        const int b = (int)(row / Npad), n = (int)(row - (int64_t)b * Npad);
        float v = -1.f;
        if (n < N && px < HW) v = logits[((int64_t)b * N + n) * HW + px];
        const unsigned long long m = __ballot(v > 0.f);
        if (lane == 0) *(uint2*)(bits + row * (HWp / 32) + (px >> 5)) = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    }
}

extern "C" int ph_binarize(const float* logits, uint32_t* bits, int B, int N, int64_t HW, void* stream) {
    PH_CHECK_ARG(logits && bits && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    const int Npad = ph_n_padded(N);
    const int64_t HWp = ph_hw_padded(HW);
    const int64_t waves = (int64_t)B * Npad * (HWp / 64);
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_binarize, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, logits, bits, B, N, Npad, HW, HWp);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---------------------------------------------------------------------------------------------
// x2 bilinear, align_corners=False.  ATen's formula (aten/native/UpSample.h area_pixel_compute_
// source_index): src = max((dst + 0.5) * 0.5 - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, n - 1);
// l1 = src - i0; l0 = 1 - l1; out = h0*(w0*a + w1*b) + h1*(w0*c + w1*d).
template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_as_f32<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_from_f32<uint16_t>(uint16_t* p, float v) { *p = (uint16_t)f2bf(v); }

template <typename T>
__global__ __launch_bounds__(256) void k_upsample2x(const T* __restrict__ src, T* __restrict__ dst, int64_t planes,
                                                    int H, int W) {
    const int W2 = 2 * W, H2 = 2 * H;
    const int64_t total = planes * H2 * (int64_t)W;   // one thread = 2 horizontally adjacent outputs
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(idx % W);
        const int64_t t = idx / W;
        const int yo = (int)(t % H2);
        const int64_t p = t / H2;
        const float sy = fmaxf((yo + 0.5f) * 0.5f - 0.5f, 0.f);
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
        const float hy1 = sy - y0, hy0 = 1.f - hy1;
        const T* r0 = src + (p * H + y0) * W;
        const T* r1 = src + (p * H + y1) * W;
        const int xm = xo > 0 ? xo - 1 : 0, xp = xo < W - 1 ? xo + 1 : xo;
        const float a_m = ld_as_f32(r0 + xm), a_c = ld_as_f32(r0 + xo), a_p = ld_as_f32(r0 + xp);
        const float b_m = ld_as_f32(r1 + xm), b_c = ld_as_f32(r1 + xo), b_p = ld_as_f32(r1 + xp);
        // output column 2*xo   : src x = xo - 0.25 -> (x0 = xo-1, l1 = 0.75) except at xo = 0 (clamped: x0 = 0, l1 = 0)
        // output column 2*xo+1 : src x = xo + 0.25 -> (x0 = xo,   l1 = 0.25), x1 clamped at the border
        float e, o;
        if (xo > 0) e = hy0 * (0.25f * a_m + 0.75f * a_c) + hy1 * (0.25f * b_m + 0.75f * b_c);
        else e = hy0 * (1.f * a_c + 0.f * a_p) + hy1 * (1.f * b_c + 0.f * b_p);
        o = hy0 * (0.75f * a_c + 0.25f * a_p) + hy1 * (0.75f * b_c + 0.25f * b_p);
        T* d = dst + (p * H2 + yo) * W2 + 2 * xo;
        st_from_f32(d, e);
        st_from_f32(d + 1, o);
    }
}

extern "C" int ph_upsample2x(const void* src, void* dst, int dtype, int64_t planes, int H, int W, void* stream) {
    PH_CHECK_ARG(src && dst && planes > 0 && H > 0 && W > 0, "bad pointer or size");
    PH_CHECK_ARG(dtype == PH_OUT_F32 || dtype == PH_OUT_BF16, "dtype must be PH_OUT_F32 or PH_OUT_BF16");
    const int64_t total = planes * 2 * H * (int64_t)W;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (dtype == PH_OUT_F32)
        hipLaunchKernelGGL(k_upsample2x<float>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, (float*)dst, planes, H, W);
    else
        hipLaunchKernelGGL(k_upsample2x<uint16_t>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)src, (uint16_t*)dst, planes, H, W);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
