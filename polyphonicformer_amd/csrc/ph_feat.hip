// Boundary format kernels: fp32 NCHW -> bf16 planes, mask logits -> mask bits, x2 bilinear upsample.
// All three are pure HBM streaming kernels: 16 B per lane, grid-stride.
#include "ph_common.h"

// ---------------------------------------------------------------------------------------------
// ingest: src fp32 [rows = B*256][HW] -> planes [P][rows][HWp]; 8 pixels per thread.
template <int P, int E>
__global__ __launch_bounds__(256) void k_ingest(const float* __restrict__ src, uint16_t* __restrict__ planes,
                                                int64_t rows, int64_t HW, int64_t HWp) {
    const int64_t plane_stride = rows * HWp;
    const int64_t row = blockIdx.y;
    const int per_row = (int)(HWp / 8);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_row; i += gridDim.x * blockDim.x) {
        const int64_t px = (int64_t)i * 8;
        float v[8];
        const float* s = src + row * HW + px;
        if (px + 8 <= HW && ((HW & 3) == 0)) {
            const uint4 a = ld_nt16(s), b = ld_nt16(s + 4);
            v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
            v[4] = __uint_as_float(b.x); v[5] = __uint_as_float(b.y); v[6] = __uint_as_float(b.z); v[7] = __uint_as_float(b.w);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (px + e < HW) ? s[e] : 0.f;
        }
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (P == 1) hi[e] = f2e<E>(v[e]);
            else f2bf_split(v[e], hi[e], lo[e]);
        }
        uint16_t* d = planes + row * HWp + px;
        st_nt16(d, make_uint4(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7])));
        if (P == 2)
            st_nt16(d + plane_stride, make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(lo[4], lo[5]), pack2(lo[6], lo[7])));
    }
}

extern "C" int ph_ingest_features(const float* src, uint16_t* planes, int B, int64_t HW, int prec, void* stream) {
    PH_CHECK_ARG(src && planes && B > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16, "prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16");
    const int64_t HWp = ph_hw_padded(HW), rows = (int64_t)B * PH_C;
    PH_CHECK_ARG(rows <= 65535, "B * 256 must be <= 65535");
    int gx = (int)((HWp / 8 + 255) / 256);
    if (gx > 8) gx = 8;                          // 8 x 256 threads x 32 B in flight per row
    const dim3 grid(gx, (unsigned)rows);
    if (prec == PH_PREC_BF16)
        hipLaunchKernelGGL((k_ingest<1, PH_E_BF16>), grid, dim3(256), 0, (hipStream_t)stream, src, planes, rows, HW, HWp);
    else if (prec == PH_PREC_F16)
        hipLaunchKernelGGL((k_ingest<1, PH_E_F16>), grid, dim3(256), 0, (hipStream_t)stream, src, planes, rows, HW, HWp);
    else
        hipLaunchKernelGGL((k_ingest<2, PH_E_BF16>), grid, dim3(256), 0, (hipStream_t)stream, src, planes, rows, HW, HWp);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

template <int E> __device__ __forceinline__ float ld_as_f32(const float* p) { return *p; }
template <int E> __device__ __forceinline__ float ld_as_f32(const uint16_t* p) { return e2f<E>(*p); }

// ---------------------------------------------------------------------------------------------
// binarize: logits [B][N][HW] -> bits [B][Npad][HWp/32].  One wave produces two words per step
// with __ballot (lane = pixel).  Rows >= N and pixels >= HW come out 0.
template <typename T, int E = PH_E_F16>
__global__ __launch_bounds__(256) void k_binarize(const T* __restrict__ logits, int64_t lbs, uint32_t* __restrict__ bits,
                                                  int B, int N, int Npad, int64_t HW, int64_t HWp, const unsigned* run_if) {
    if (run_if && *run_if == 0) return;
    // one row (b, n) per blockIdx.y; each lane tests V consecutive pixels of ONE 16-byte load (V = 4 fp32 / 8 sixteen-bit values); a
    // wave covers 64 V pixels = 2 V words; the 32 / V lanes of a word OR their V-bit pieces together with xor-shuffles
    constexpr int V = 16 / (int)sizeof(T), LPW = 32 / V;              // pixels per lane, lanes per word
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.y; row < B * Npad; row += gridDim.y) {
    const int b = row / Npad, n = row - b * Npad;
    const bool live = n < N;
    const T* src = logits + (int64_t)b * lbs + (int64_t)n * HW;
    uint32_t* dst = bits + (int64_t)row * (HWp / 32);
    const bool vec_ok = (HW % V) == 0 && (((uintptr_t)src) & 15) == 0;
    for (int c = blockIdx.x * 4 + wave; (int64_t)c * (64 * V) < HWp; c += gridDim.x * 4) {
        const int64_t px = (int64_t)c * (64 * V) + lane * V;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = -1.f;
        if (live && px < HW) {
            if (vec_ok && px + V <= HW) {
                const uint4 q = ld_nt16(src + px);
                if constexpr (sizeof(T) == 4) {
                    v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w);
                } else {
                    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] = e2f<E>(w[e] & 0xFFFFu); v[2 * e + 1] = e2f<E>(w[e] >> 16); }
                }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) if (px + e < HW) v[e] = ld_as_f32<E>(src + px + e);
            }
        }
        uint32_t piece = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) piece |= (v[e] > PH_BIN_THR ? 1u : 0u) << e;
        uint32_t word = piece << (V * (lane & (LPW - 1)));
        word |= __shfl_xor(word, 1);
        word |= __shfl_xor(word, 2);
        if (LPW == 8) word |= __shfl_xor(word, 4);
        const int wi = lane / LPW;                                    // word of this wave's 2 V
        if ((lane & (LPW - 1)) == 0 && (int64_t)c * (64 * V) + wi * 32 < HWp) dst[c * (2 * V) + wi] = word;
    }
    }
}

// logits fp32 (PH_OUT_F32), fp16 (PH_OUT_F16) or bf16 (PH_OUT_BF16: round 5 -- 16-bit mask logits are what a 16-bit KernelHead grade hands
// over and what cfg2's bf16 / SURVEY 8d's N * HW * e_f "initial mask logits read" mean); run_if: optional device predicate (the launch returns at once when *run_if == 0)
extern "C" int ph_binarize_if(const void* logits, int dtype, int64_t logits_batch_stride, uint32_t* bits, int B, int N, int64_t HW,
                              const uint32_t* run_if, void* stream) {
    PH_CHECK_ARG(logits && bits && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG(dtype == PH_OUT_F32 || dtype == PH_OUT_F16 || dtype == PH_OUT_BF16, "dtype must be PH_OUT_F32, PH_OUT_F16 or PH_OUT_BF16");
    PH_CHECK_ARG(logits_batch_stride == 0 || logits_batch_stride >= (int64_t)N * HW, "batch stride smaller than a frame");
    if (!logits_batch_stride) logits_batch_stride = (int64_t)N * HW;
    const int Npad = ph_n_padded(N);
    const int64_t HWp = ph_hw_padded(HW);
    PH_CHECK_ARG((int64_t)B * Npad <= 65535, "B * Npad must be <= 65535");
    int gx = (int)((HWp + 1023) / 1024);          // 4 waves x 256 px per block step
    if (gx > 8) gx = 8;
    // predicated use (the fallback of a one-pass KernelHead launch): a grid that walks the rows, so that the launch that returns
    // at once is 8 x 256 workgroups and not 8 x B x Npad (20 480 at cfg2, 16 frames: 9 us of empty workgroups per call)
    int gy = B * Npad;
    if (run_if && gy > 256) gy = 256;
    if (dtype == PH_OUT_F32)
        hipLaunchKernelGGL(k_binarize<float>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                           logits_batch_stride, bits, B, N, Npad, HW, HWp, (const unsigned*)run_if);
    else if (dtype == PH_OUT_F16)
        hipLaunchKernelGGL((k_binarize<uint16_t, PH_E_F16>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)logits,
                           logits_batch_stride, bits, B, N, Npad, HW, HWp, (const unsigned*)run_if);
    else
        hipLaunchKernelGGL((k_binarize<uint16_t, PH_E_BF16>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)logits,
                           logits_batch_stride, bits, B, N, Npad, HW, HWp, (const unsigned*)run_if);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_binarize(const float* logits, int64_t logits_batch_stride, uint32_t* bits, int B, int N, int64_t HW,
                           void* stream) {
    return ph_binarize_if(logits, PH_OUT_F32, logits_batch_stride, bits, B, N, HW, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------
// x2 bilinear, align_corners=False.  ATen's formula (aten/native/UpSample.h area_pixel_compute_
// source_index): src = max((dst + 0.5) * 0.5 - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, n - 1);
// l1 = src - i0; l0 = 1 - l1; out = h0*(w0*a + w1*b) + h1*(w0*c + w1*d).
template <int E> __device__ __forceinline__ void st_from_f32(float* p, float v) { *p = v; }
template <int E> __device__ __forceinline__ void st_from_f32(uint16_t* p, float v) { *p = (uint16_t)f2e<E>(v); }

// One thread = 4 source columns x 2 source rows -> an 8-column x 4-row output patch (16 B bf16 / 32 B fp32
// stores per row).  It needs the 4 x 6 source neighbourhood: the 4 rows are loaded once (vector loads), the
// two edge columns come from the neighbouring lanes by cross-lane shuffles (a real load only at wave / row
// boundaries), so HBM sees the source once and the destination once (5 * planes*H*W elements).
template <typename T, bool VEC, int E>
__global__ __launch_bounds__(256) void k_upsample2x(const T* __restrict__ src, T* __restrict__ dst, int64_t planes,
                                                    int H, int W) {
    const int W2 = 2 * W;
    const unsigned segs = (W + 3) / 4, hp = (H + 1) / 2;
    const unsigned total = (unsigned)planes * hp * segs;             // checked < 2^31 by the launcher
    const unsigned nthreads = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    const bool uniform_row = (segs % 64) == 0;                       // fp32 store exchange below
    for (unsigned base = blockIdx.x * blockDim.x; base < total; base += nthreads) {
        const unsigned idx = base + threadIdx.x;
        const bool live = idx < total;
        const unsigned t = (live ? idx : total - 1) / segs;          // = p*hp + pair
        const int j = (int)((live ? idx : total - 1) - t * segs);
        const int64_t p = t / hp;
        const int y = 2 * (int)(t - (unsigned)p * hp);               // first source row of the pair
        const int x0 = 4 * j;
        int rows[4];
        rows[0] = y > 0 ? y - 1 : 0;
        rows[1] = y;
        rows[2] = y + 1 < H ? y + 1 : H - 1;
        rows[3] = y + 2 < H ? y + 2 : H - 1;
        float h[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const T* row = src + (p * H + rows[r]) * W;
            float v[6];
            if (VEC) {
                if (sizeof(T) == 2) {
                    const uint2 q = *(const uint2*)(row + x0);   // cached: every source row is read by two row pairs (nt: 175 -> 223 us)
                    v[1] = e2f<E>(q.x & 0xFFFF); v[2] = e2f<E>(q.x >> 16); v[3] = e2f<E>(q.y & 0xFFFF); v[4] = e2f<E>(q.y >> 16);
                } else {
                    const uint4 q = *(const uint4*)(row + x0);   // cached, as above
                    v[1] = __uint_as_float(q.x); v[2] = __uint_as_float(q.y); v[3] = __uint_as_float(q.z); v[4] = __uint_as_float(q.w);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[1 + c] = ld_as_f32<E>(row + (x0 + c < W ? x0 + c : W - 1));
            }
            // edge columns x0-1 and x0+4 from the neighbouring lanes (same source row when j-1 / j+1 exist
            // in this wave), otherwise clamp or load
            const float from_left = __shfl_up(v[4], 1), from_right = __shfl_down(v[1], 1);
            if (j == 0) v[0] = v[1];
            else if (lane > 0) v[0] = from_left;
            else v[0] = ld_as_f32<E>(row + x0 - 1);
            if (x0 + 4 >= W) v[5] = VEC ? v[4] : ld_as_f32<E>(row + W - 1);
            else if (lane < 63) v[5] = from_right;
            else v[5] = ld_as_f32<E>(row + x0 + 4);
            // output column 2x   : src = x - 0.25 -> (x-1, x) weights (0.25, 0.75); x = 0: clamped, weights (1, 0)
            // output column 2x+1 : src = x + 0.25 -> (x, x+1 clamped) weights (0.75, 0.25)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool first = (x0 + c == 0);
                h[r][2 * c] = first ? (1.f * v[1] + 0.f * v[2]) : (0.25f * v[c] + 0.75f * v[c + 1]);
                h[r][2 * c + 1] = 0.75f * v[c + 1] + 0.25f * v[c + 2];
            }
        }
        if (!live) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // output row 2*y + r: even -> source rows (s-1, s) weights (.25,.75) [row 0: (1,0)], odd -> (s, s+1) (.75,.25)
            const int yo = 2 * y + r;
            if (yo >= 2 * H) break;
            const int ia = (r + 1) >> 1, ib = ia + 1;            // r=0:(0,1) r=1:(1,2) r=2:(1,2) r=3:(2,3)
            const bool even = (r & 1) == 0;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (even) o[e] = (yo > 0) ? (0.25f * h[ia][e] + 0.75f * h[ib][e]) : (1.f * h[ib][e] + 0.f * h[ib + 1 < 4 ? ib + 1 : 3][e]);
                else o[e] = 0.75f * h[ia][e] + 0.25f * h[ib][e];
            }
            T* d = dst + (p * 2 * H + yo) * W2 + 2 * x0;
            if (VEC) {
                if (sizeof(T) == 2) {
                    st_nt16(d, make_uint4(f2e_pk<E>(o[0], o[1]), f2e_pk<E>(o[2], o[3]), f2e_pk<E>(o[4], o[5]), f2e_pk<E>(o[6], o[7])));
                } else {
                    // each lane owns 32 contiguous bytes: as two 16-byte stores per lane every instruction would write
                    // half of each 32-byte piece (partial lines with non-temporal stores: 3.3 TB/s).  Exchange so that
                    // one instruction covers 1 KiB contiguously: lane l stores what lane (l >> 1) [+ 32] computed.
                    if (uniform_row) {       // the wave = 64 consecutive segments of one output row, all live
                        float a4[4], b4[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float lo_v = __shfl(o[c], lane >> 1), hi_v = __shfl(o[4 + c], lane >> 1);
                            const float lo_w = __shfl(o[c], 32 + (lane >> 1)), hi_w = __shfl(o[4 + c], 32 + (lane >> 1));
                            a4[c] = (lane & 1) ? hi_v : lo_v;
                            b4[c] = (lane & 1) ? hi_w : lo_w;
                        }
                        float* wrow = (float*)d - 8 * lane;                   // first output of lane 0 of this wave
                        st_nt16(wrow + 4 * lane, make_float4(a4[0], a4[1], a4[2], a4[3]));
                        st_nt16(wrow + 256 + 4 * lane, make_float4(b4[0], b4[1], b4[2], b4[3]));
                    } else {
                        st_nt16(d, make_float4(o[0], o[1], o[2], o[3]));
                        st_nt16((float*)d + 4, make_float4(o[4], o[5], o[6], o[7]));
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (2 * x0 + e < W2) st_from_f32<E>(d + e, o[e]);
            }
        }
    }
}

extern "C" int ph_upsample2x(const void* src, void* dst, int dtype, int64_t planes, int H, int W, void* stream) {
    PH_CHECK_ARG(src && dst && planes > 0 && H > 0 && W > 0, "bad pointer or size");
    PH_CHECK_ARG(dtype == PH_OUT_F32 || dtype == PH_OUT_BF16 || dtype == PH_OUT_F16, "dtype must be PH_OUT_F32, PH_OUT_BF16 or PH_OUT_F16");
    const int64_t total = planes * ((H + 1) / 2) * (int64_t)((W + 3) / 4);
    PH_CHECK_ARG(total < (1ll << 31), "planes * H * W / 8 must be < 2^31");
    int64_t blocks = (total + 255) / 256;

    const bool vec = (W % 4) == 0;      // rows then start 8-byte (bf16) / 16-byte (fp32) aligned
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PH_OUT_F32) {
        if (vec) hipLaunchKernelGGL((k_upsample2x<float, true, 0>), dim3((int)blocks), dim3(256), 0, s, (const float*)src, (float*)dst, planes, H, W);
        else hipLaunchKernelGGL((k_upsample2x<float, false, 0>), dim3((int)blocks), dim3(256), 0, s, (const float*)src, (float*)dst, planes, H, W);
    } else if (dtype == PH_OUT_BF16) {
        if (vec) hipLaunchKernelGGL((k_upsample2x<uint16_t, true, PH_E_BF16>), dim3((int)blocks), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, planes, H, W);
        else hipLaunchKernelGGL((k_upsample2x<uint16_t, false, PH_E_BF16>), dim3((int)blocks), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, planes, H, W);
    } else {
        if (vec) hipLaunchKernelGGL((k_upsample2x<uint16_t, true, PH_E_F16>), dim3((int)blocks), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, planes, H, W);
        else hipLaunchKernelGGL((k_upsample2x<uint16_t, false, PH_E_F16>), dim3((int)blocks), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, planes, H, W);
    }
    PH_CHECK_LAUNCH();
    return PH_OK;
}
