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
#include <stdlib.h>

#include "ph_common.h"

constexpr int CONV_T = 64;            // pixels per tile: one 128-byte line per channel row

__device__ __forceinline__ void st_out(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_out(uint16_t* p, float v) { *p = (uint16_t)f2bf(v); }

// LDS image of a tile: [256 rows][8 x 16-byte pieces], rows contiguous (128 B) because the tile is
// written by LDS-DMA (global_load_lds: wave-uniform base + lane*16, no padding possible).  To keep the
// transposing reads conflict free the piece index is XOR-swizzled with bit 1 of the row on the SOURCE
// side (same 128-byte line, so coalescing is unchanged) and the same XOR is applied by the readers:
// the 4 rows x 2 half-tiles a 32-lane read touches then cover 8 distinct 32-byte bank windows.
__device__ __forceinline__ int conv_swz(int row) { return ((row >> 1) & 1) << 2; }

// row stride (elements) of the per-wave epilogue patch: 64 px + 16 bytes of padding
template <typename OutT> __device__ __host__ constexpr int EP_LD() { return CONV_T + 16 / (int)sizeof(OutT); }

template <int PA, int NRT, bool BITS, typename OutT>
__global__ __launch_bounds__(NRT * 64) void k_dynconv(const uint16_t* __restrict__ planes,
                                                      const uint16_t* __restrict__ kern, int64_t kern_plane_stride,
                                                      int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                      int64_t kbias_batch_stride, uint32_t* __restrict__ bits_out,
                                                      OutT* __restrict__ logits_out, int64_t out_batch_stride, int B,
                                                      int N, int64_t HW, int64_t HWp, int tiles_per_wg) {
    constexpr int Npad = NRT * 32;
    constexpr int TILE = 256 * CONV_T;                               // elements per plane per buffer
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [2 buffers][PA][256][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int rt = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const float* kbp = kbias + (int64_t)b * kbias_batch_stride + rt * 32 + 4 * g;

    const int ntiles = (int)(HWp / CONV_T);
    const int t0 = blockIdx.x * tiles_per_wg;
    const int t1 = (t0 + tiles_per_wg < ntiles) ? t0 + tiles_per_wg : ntiles;

    // LDS-DMA of tile t into buffer `buf`: 32 wave-instructions of 1 KiB (8 rows x 128 B) per plane
    auto issue_tile = [&](int t, int buf) {
        for (int j = rt; j < 32 * PA; j += NRT) {
            const int p = j >> 5, jj = j & 31;
            const int row = jj * 8 + (lane >> 3);
            const int q = (lane & 7) ^ conv_swz(row);
            const uint16_t* src = fbase + p * fplane + (int64_t)row * HWp + (int64_t)t * CONV_T + q * 8;
            uint16_t* dst = lds + (buf * PA + p) * TILE + jj * 512;          // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (PH_LDS void*)dst, 16, 0, 0);
        }
    };
    if (t0 < t1) issue_tile(t0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        if (t + 1 < t1) issue_tile(t + 1, cur ^ 1);      // DMA overlaps the MFMA phase below
        const uint16_t* tile = lds + cur * PA * TILE;
        f32x16_t acc[CONV_T / 32];
#pragma unroll
        for (int ct = 0; ct < CONV_T / 32; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                uint4 bf[PA];
                const int row = ks * 16 + g * 8 + (i16 >> 2);
                const int off = row * CONV_T + (((ct * 4 + gi * 2) ^ conv_swz(row)) * 8) + (i16 & 3) * 4;
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const uint2 lo = lds_read_tr16(tile + p * TILE + off);
                    const uint2 hi = lds_read_tr16(tile + p * TILE + off + 4 * CONV_T);
                    bf[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                acc[ct] = mfma32(af[0][ks], bf[0], acc[ct]);
                if (PA == 2) {
                    acc[ct] = mfma32(af[0][ks], bf[PA - 1], acc[ct]);
                    acc[ct] = mfma32(af[PA - 1][ks], bf[0], acc[ct]);
                }
            }
        }
        const int64_t px0 = (int64_t)t * CONV_T;
        if (BITS) {
#pragma unroll
            for (int ct = 0; ct < CONV_T / 32; ++ct) {
                const int64_t px = px0 + ct * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    const int row = rt * 32 + rr + 4 * g;
                    const float v = acc[ct][r] + kbp[rr];
                    const unsigned long long m = __ballot(v > 0.f && px < HW && row < N);
                    // lanes 0..31 -> row with g = 0, lanes 32..63 -> the row 4 below
                    if (lane == 0) {
                        uint32_t* w = bits_out + ((int64_t)b * Npad + rt * 32 + rr) * (HWp / 32) + (px >> 5);
                        w[0] = (uint32_t)m;
                        w[4 * (HWp / 32)] = (uint32_t)(m >> 32);
                    }
                }
            }
        } else {
            constexpr int PER16 = 16 / (int)sizeof(OutT);                    // elements per 16-byte store
            // (split precision keeps both LDS buffers at 64 KiB each: no room for the patch, scalar stores)
            const bool fast = (PA == 1) && (HW % PER16 == 0) && (px0 + CONV_T <= HW);
            if (fast) {
                // transpose the [32 rows][64 px] result through a per-wave LDS patch so that every lane stores
                // 16 contiguous bytes of one row (whole 128 / 256-byte row segments per instruction group)
                OutT* ep = (OutT*)(lds + 2 * PA * TILE) + rt * (32 * EP_LD<OutT>());
#pragma unroll
                for (int ct = 0; ct < CONV_T / 32; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = (r & 3) + 8 * (r >> 2);
                        st_out(ep + (rr + 4 * g) * EP_LD<OutT>() + ct * 32 + (lane & 31), acc[ct][r] + kbp[rr]);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                constexpr int LPR = CONV_T / PER16;                          // lanes per row: 8 (bf16) / 16 (fp32)
                constexpr int RPI = 64 / LPR;                                // rows per iteration
#pragma unroll
                for (int it = 0; it < 32 / RPI; ++it) {
                    const int rl = it * RPI + lane / LPR, piece = lane % LPR;
                    const int row = rt * 32 + rl;
                    const uint4 v = *(const uint4*)(ep + rl * EP_LD<OutT>() + piece * PER16);
                    if (row < N)
                        *(uint4*)(logits_out + (int64_t)b * out_batch_stride + (int64_t)row * HW + px0 + piece * PER16) = v;
                }
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int ct = 0; ct < CONV_T / 32; ++ct) {
                    const int64_t px = px0 + ct * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = (r & 3) + 8 * (r >> 2);
                        const int row = rt * 32 + rr + 4 * g;
                        if (row < N && px < HW)
                            st_out(logits_out + (int64_t)b * out_batch_stride + (int64_t)row * HW + px, acc[ct][r] + kbp[rr]);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile t+1 have landed
        __syncthreads();                                    // ... and everybody's; tile t's readers are done
    }
}

template <int PA, int NRT>
static int launch_conv(const uint16_t* planes, const uint16_t* kern, int64_t kps, int64_t kbs, const float* kbias,
                       int64_t bbs, uint32_t* bits_out, void* logits_out, int out_dtype, int64_t obs, int B, int N,
                       int64_t HW, hipStream_t s) {
    const int64_t HWp = ph_hw_padded(HW);
    const int ntiles = (int)(HWp / CONV_T);
    // at most one resident generation of workgroups (2 per CU = 512), evenly split over the frames: no tail
    int gx = 512 / B;
    if (gx < 1) gx = 1;
    int tpw = (ntiles + gx - 1) / gx;
    if (tpw < 1) tpw = 1;
    if (const char* e = getenv("PH_CONV_TPW")) tpw = atoi(e);   // tuning knob
    const dim3 grid((ntiles + tpw - 1) / tpw, B), block(NRT * 64);
    size_t lds = (size_t)2 * PA * 256 * CONV_T * sizeof(uint16_t);
    if (!bits_out && PA == 1) lds += (size_t)NRT * 32 * (out_dtype == PH_OUT_F32 ? EP_LD<float>() * 4 : EP_LD<uint16_t>() * 2);
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
