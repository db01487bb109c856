// A13 -- dynamic 1x1 convolution  logits[n][hw] = sum_c kern[n][c] * feat[c][hw] + kbias[n]
// (kernel_update_head.py:317-329, with feat_transform folded into kern/kbias by the query kernel).
//
// GEMM M = Npad, N = HW, K = 256.  The feature map is [c][hw] (hw contiguous) but a bf16 MFMA wants
// 8 consecutive k (= c) per lane, so the 64-pixel feature tile goes through LDS as it lies in HBM
// ([256 c][64 px], row stride 192 B) and the B fragments are fetched with the gfx950 transposing
// read ds_read_b64_tr_b16: the tile is read from HBM once, coalesced (128 B per channel row).
// One workgroup = Npad/32 waves; wave w owns query rows 32w..32w+31 and keeps its whole A operand
// (32 x 256 kernel slice = 64 VGPRs per plane) in registers across `tiles_per_wg` pixel tiles.
// Non-final stages emit only the mask bits the next pooling pass consumes (__ballot -> 1 bit/px),
// the final stage writes the logits.  Roofline: HBM.
#include "ph_common.h"

constexpr int CONV_T = 64;            // pixels per tile
constexpr int CONV_LDT = CONV_T + 32; // LDS row stride in elements (192 B): 4 consecutive rows hit disjoint banks

__device__ __forceinline__ void st_out(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_out(uint16_t* p, float v) { *p = (uint16_t)f2bf(v); }

template <int PA, int NRT, bool BITS, typename OutT>
__global__ __launch_bounds__(NRT * 64) void k_dynconv(const uint16_t* __restrict__ planes,
                                                      const uint16_t* __restrict__ kern, int64_t kern_plane_stride,
                                                      int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                      int64_t kbias_batch_stride, uint32_t* __restrict__ bits_out,
                                                      OutT* __restrict__ logits_out, int64_t out_batch_stride, int B,
                                                      int N, int64_t HW, int64_t HWp, int tiles_per_wg) {
    constexpr int Npad = NRT * 32, NT = NRT * 64;
    constexpr int PIECES = (256 * (CONV_T / 8) + NT - 1) / NT;   // 16-byte pieces per thread per plane
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [PA][256][CONV_LDT]
    constexpr int LDS_PLANE = 256 * CONV_LDT;

    const int tid = threadIdx.x, lane = tid & 63, rt = tid >> 6;
    const int b = blockIdx.y;
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    const int64_t fplane = (int64_t)B * PH_C * HWp;
    const uint16_t* fbase = planes + (int64_t)b * PH_C * HWp;

    // A operand: this wave's 32 kernel rows, all 16 k-steps
    uint4 af[PA][16];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const uint16_t* kr = kern + p * kern_plane_stride + (int64_t)b * kern_batch_stride + (rt * 32 + (lane & 31)) * PH_C + g * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) af[p][ks] = *(const uint4*)(kr + ks * 16);
    }
    float kb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) kb[r] = kbias[(int64_t)b * kbias_batch_stride + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];

    const int ntiles = (int)(HWp / CONV_T);
    const int t0 = blockIdx.x * tiles_per_wg;
    const int t1 = (t0 + tiles_per_wg < ntiles) ? t0 + tiles_per_wg : ntiles;

    uint4 st[PA][PIECES];
    auto issue_loads = [&](int t) {
#pragma unroll
        for (int p = 0; p < PA; ++p)
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int idx = tid + q * NT;
                if (idx < 256 * (CONV_T / 8)) {
                    const int row = idx >> 3, piece = idx & 7;
                    st[p][q] = *(const uint4*)(fbase + p * fplane + (int64_t)row * HWp + (int64_t)t * CONV_T + piece * 8);
                }
            }
    };
    if (t0 < t1) issue_loads(t0);

    for (int t = t0; t < t1; ++t) {
        __syncthreads();   // previous tile's readers are done
#pragma unroll
        for (int p = 0; p < PA; ++p)
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int idx = tid + q * NT;
                if (idx < 256 * (CONV_T / 8)) {
                    const int row = idx >> 3, piece = idx & 7;
                    *(uint4*)(lds + p * LDS_PLANE + row * CONV_LDT + piece * 8) = st[p][q];
                }
            }
        __syncthreads();
        if (t + 1 < t1) issue_loads(t + 1);   // in flight during the MFMA phase

#pragma unroll
        for (int ct = 0; ct < CONV_T / 32; ++ct) {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                uint4 bf[PA];
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const uint16_t* a0 = lds + p * LDS_PLANE + (ks * 16 + g * 8 + (i16 >> 2)) * CONV_LDT + ct * 32 +
                                         gi * 16 + (i16 & 3) * 4;
                    const uint2 lo = lds_read_tr16(a0);
                    const uint2 hi = lds_read_tr16(a0 + 4 * CONV_LDT);
                    bf[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                acc = mfma32(af[0][ks], bf[0], acc);
                if (PA == 2) {
                    acc = mfma32(af[0][ks], bf[PA - 1], acc);
                    acc = mfma32(af[PA - 1][ks], bf[0], acc);
                }
            }
            const int64_t px = (int64_t)t * CONV_T + ct * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                const float v = acc[r] + kb[r];
                if (BITS) {
                    const unsigned long long m = __ballot(v > 0.f && px < HW && row < N);
                    // lanes 0..31 -> row with g = 0, lanes 32..63 -> the row 4 below
                    if (lane == 0) {
                        const int row0 = rt * 32 + (r & 3) + 8 * (r >> 2);
                        uint32_t* w = bits_out + ((int64_t)b * Npad + row0) * (HWp / 32) + (px >> 5);
                        w[0] = (uint32_t)m;
                        w[4 * (HWp / 32)] = (uint32_t)(m >> 32);
                    }
                } else {
                    if (row < N && px < HW) st_out(logits_out + (int64_t)b * out_batch_stride + (int64_t)row * HW + px, v);
                }
            }
        }
    }
}

template <int PA, int NRT>
static int launch_conv(const uint16_t* planes, const uint16_t* kern, int64_t kps, int64_t kbs, const float* kbias,
                       int64_t bbs, uint32_t* bits_out, void* logits_out, int out_dtype, int64_t obs, int B, int N,
                       int64_t HW, hipStream_t s) {
    const int64_t HWp = ph_hw_padded(HW);
    const int ntiles = (int)(HWp / CONV_T);
    int tpw = (int)(((int64_t)ntiles * B + 1023) / 1024);
    if (tpw < 1) tpw = 1;
    if (tpw > 16) tpw = 16;
    const dim3 grid((ntiles + tpw - 1) / tpw, B), block(NRT * 64);
    const size_t lds = (size_t)PA * 256 * CONV_LDT * sizeof(uint16_t);
#define PH_CONV_LAUNCH(BITS, T)                                                                                      \
    do {                                                                                                             \
        static bool once = false;                                                                                    \
        if (!once) {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)k_dynconv<PA, NRT, BITS, T>,                                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
            once = true;                                                                                             \
        }                                                                                                            \
        hipLaunchKernelGGL((k_dynconv<PA, NRT, BITS, T>), grid, block, lds, s, planes, kern, kps, kbs, kbias, bbs,   \
                           bits_out, (T*)logits_out, obs, B, N, HW, HWp, tpw);                                       \
    } while (0)
    if (bits_out) PH_CONV_LAUNCH(true, float);
    else if (out_dtype == PH_OUT_F32) PH_CONV_LAUNCH(false, float);
    else PH_CONV_LAUNCH(false, uint16_t);
#undef PH_CONV_LAUNCH
    return 0;
}

extern "C" int ph_dynconv(const uint16_t* planes, const uint16_t* kern, int64_t kern_plane_stride,
                          int64_t kern_batch_stride, const float* kbias, int64_t kbias_batch_stride, uint32_t* bits_out,
                          void* logits_out, int out_dtype, int64_t out_batch_stride, int B, int N, int64_t HW, int prec,
                          void* stream) {
    PH_CHECK_ARG(planes && kern && kbias && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG((bits_out != nullptr) != (logits_out != nullptr), "exactly one of bits_out / logits_out");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    PH_CHECK_ARG(out_dtype == PH_OUT_F32 || out_dtype == PH_OUT_BF16, "bad out_dtype");
    PH_CHECK_ARG(N <= 256, "at most 256 queries");
    const int nrt = ph_n_padded(N) / 32;
    hipStream_t s = (hipStream_t)stream;
#define PH_CONV_CASE(R)                                                                                         \
    case R:                                                                                                     \
        if (prec == PH_PREC_BF16)                                                                               \
            launch_conv<1, R>(planes, kern, kern_plane_stride, kern_batch_stride, kbias, kbias_batch_stride, bits_out,   \
                              logits_out, out_dtype, out_batch_stride, B, N, HW, s); \
        else                                                                                                    \
            launch_conv<2, R>(planes, kern, kern_plane_stride, kern_batch_stride, kbias, kbias_batch_stride, bits_out,   \
                              logits_out, out_dtype, out_batch_stride, B, N, HW, s); \
        break;
    switch (nrt) {
        PH_CONV_CASE(1) PH_CONV_CASE(2) PH_CONV_CASE(3) PH_CONV_CASE(4)
        PH_CONV_CASE(5) PH_CONV_CASE(6) PH_CONV_CASE(7) PH_CONV_CASE(8)
        default: ph_set_error("ph_dynconv: unsupported N"); return PH_EUNSUPPORTED;
    }
#undef PH_CONV_CASE
    PH_CHECK_LAUNCH();
    return PH_OK;
}
