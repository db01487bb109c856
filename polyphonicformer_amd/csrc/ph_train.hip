// libpolyhead, training side: the map-sized products of the backward pass and of the
// fp32 training forward, on fp32 NCHW maps as autograd hands them over.
//
//   ph_rows_x_map  : Y[b][m][p] = sum_k A[b][m][k] * X[b][k][p]      rows x map   -> map
//                    forward of every 1x1 convolution of the path (static or dynamic kernels: kernel_head.py:250-295,
//                    kernel_update_head.py:317-329) and, with A transposed by the caller, the gradient w.r.t. the feature
//                    map of the dynamic convolution (A = kernels^T, X = dL/dlogits) and of the hard-mask pooling
//                    (A = dL/dpooled^T, X = mask logits binarised on the fly).
//   ph_map_x_map_t  : O[b][m][k] = sum_p G[b][m][p] * X[b][k][p]      map x map^T  -> rows
//                    the hard-mask pooling forward (G = mask logits binarised on the fly, kernel_update_head.py:236-242),
//                    and the gradient w.r.t. the kernels of a 1x1 convolution (G = dL/dlogits, X = features).
//   ph_upsample2x_bwd : the transpose of the x2 bilinear upsample (kernel_update.py:131-143).
//
// Arithmetic: every fp32 operand is split in registers into hi + lo bf16 and a product is three MFMAs
// (hi*hi + hi*lo + lo*hi, fp32 accumulation): ~2^-17 relative per operand, the grade of the inference path's
// PH_PREC_SPLIT.  Both products are HBM-bound (K, M <= 320 against maps of 10^4..10^5 pixels): one pass over the
// map operand(s), no LDS -- the MFMA fragment maps let every lane read its operands straight from global memory
// (32 contiguous bytes along the contraction for the row operands; for the map operand of ph_rows_x_map, whose
// contiguous axis is the OUTPUT pixel, eight 4-byte loads that a 16-lane group coalesces into 64-byte segments).
#include "ph_common.h"

namespace {

struct Frag {
    uint4 hi, lo;
};

__device__ __forceinline__ Frag split8(const float* v) {
    uint32_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f2bf_split(v[e], h[e], l[e]);
    Frag f;
    f.hi = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
    f.lo = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
    return f;
}

__device__ __forceinline__ f32x4_t mfma3(const Frag& a, const Frag& b, f32x4_t c) {
    c = mfma16(a.lo, b.hi, c);
    c = mfma16(a.hi, b.lo, c);
    return mfma16(a.hi, b.hi, c);
}

__device__ __forceinline__ float binz(float z) { return z > PH_BIN_THR ? 1.f : 0.f; }

// ---- rows x map --------------------------------------------------------------------------------------------------------------
// grid (ceil(HW / (64 * CT)), B, m-passes); 4 waves, each CT column tiles of 16 pixels x RT row tiles of 16 rows: (CT, RT) =
// (4, 10), one pass per 160 rows; (2, 20) = all of up to 320 rows in one pass for the binarised map operand (160 accumulator
// registers either way).
// A: [B or 1][Mpad][lda] fp32, Mpad % 16 == 0, lda % 8 == 0, zero padded (the caller pads: it is a few hundred KB).
// epilogue extras (round 5): bias [B][M] added to every pixel of row m (the scalar bias of a folded dynamic kernel), and
// add [B][M][HW] (may be Y itself: a second gradient contribution lands where the first one lies, no ATen add over the map)
template <bool BIN, int CT, int RT>
__global__ __launch_bounds__(256, 2) void k_rows_x_map(const float* __restrict__ A, int64_t a_batch_stride, int lda, int Mpad, int M, int K,
                                                       const float* __restrict__ X, float* __restrict__ Y, int64_t HW,
                                                       const float* __restrict__ bias, const float* add) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * (64 * CT) + wave * (16 * CT);
    if (p0 >= HW) return;
    const int mt0 = blockIdx.z * RT;
    const int nrt = min(RT, Mpad / 16 - mt0);
    const float* Ab = A + b * a_batch_stride + (int64_t)mt0 * 16 * lda;
    const float* Xb = X + (int64_t)b * K * HW;
    f32x4_t acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        Frag xb[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int64_t p = p0 + t * 16 + c;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + g * 8 + e;
                const bool ok = k < K && p < HW;
                const float z = ok ? Xb[(int64_t)k * HW + p] : 0.f;
                v[e] = BIN ? (ok ? binz(z) : 0.f) : z;
            }
            xb[t] = split8(v);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            if (r < nrt) {
                const int k = k0 + g * 8;
                float v[8];
                if (k < lda) {
                    const float4* src = (const float4*)(Ab + (int64_t)(r * 16 + c) * lda + k);
                    const float4 q0 = src[0], q1 = src[1];
                    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
                const Frag a = split8(v);
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[r][t] = mfma3(a, xb[t], acc[r][t]);
            }
        }
    }
    float* Yb = Y + (int64_t)b * M * HW;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (r < nrt) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int64_t p = p0 + t * 16 + c;
                {
                    // every load of the epilogue's extras before the first store of the group: interleaved, each load would wait
                    // behind the previous store (possible aliasing) -- a dependent memory round trip per element
                    float ev[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = (mt0 + r) * 16 + g * 4 + i;
                        ev[i] = acc[r][t][i];
                        if (m < M && p < HW) {
                            if (bias) ev[i] += bias[(int64_t)b * M + m];
                            if (add) ev[i] += add[((int64_t)b * M + m) * HW + p];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = (mt0 + r) * 16 + g * 4 + i;
                        if (m < M && p < HW) __builtin_nontemporal_store(ev[i], &Yb[(int64_t)m * HW + p]);
                    }
                }
            }
        }
    }
}

// ---- 3x3 convolution as nine shifted rows-x-map products (round 5: the training forward / backward of the neck's towers) ------
// Y[b][m][q] = sum_tap sum_k A[tap][m][k] X[b][k][src_tap(q)], zero where src falls outside the input.  The map operand of
// k_rows_x_map is read with 4-byte loads whose address each lane computes itself, so a tap is nothing but another address:
//   mode 0 (forward, stride s, pad 1): q = (oy, ox) of the Ho x Wo output, src = (oy s + dy - 1, ox s + dx - 1) of the Hi x Wi input;
//          also the input gradient of a stride-1 conv (A = flipped, transposed taps, X = dL/dY);
//   mode 1 (input gradient of the stride-2 conv): q = (iy, ix) of the Ho x Wo = INPUT-resolution map, src = ((iy + 1 - dy) / 2,
//          (ix + 1 - dx) / 2) of the Hi x Wi gradient map where both numerators are even and in range.
// A: [9][Mpad][lda] fp32, zero padded like k_rows_x_map's.  Grid (pixel tiles, B, m passes); accumulators stay in registers over
// the nine taps (one pass over X per m pass, X re-reads of the taps hit L2).
template <int CT, int RT>
__global__ __launch_bounds__(256, 2) void k_conv3x3_rows(const float* __restrict__ A, int lda, int Mpad, int M, int K, const float* __restrict__ X,
                                                         float* __restrict__ Y, int Hi, int Wi, int Ho, int Wo, int stride, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int64_t HWo = (int64_t)Ho * Wo, HWi = (int64_t)Hi * Wi;
    const int64_t p0 = (int64_t)blockIdx.x * (64 * CT) + wave * (16 * CT);
    if (p0 >= HWo) return;
    const int mt0 = blockIdx.z * RT;
    const int nrt = min(RT, Mpad / 16 - mt0);
    const float* Xb = X + (int64_t)b * K * HWi;
    f32x4_t acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int oy[CT], ox[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int64_t p = p0 + t * 16 + c;
        oy[t] = p < HWo ? (int)(p / Wo) : -100000;
        ox[t] = (int)(p % Wo);
    }
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        int64_t src[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            int sy, sx;
            bool ok;
            if (mode == 0) {
                sy = oy[t] * stride + dy - 1; sx = ox[t] * stride + dx - 1;
                ok = oy[t] >= 0 && sy >= 0 && sy < Hi && sx >= 0 && sx < Wi;
            } else {
                const int ny = oy[t] + 1 - dy, nx = ox[t] + 1 - dx;
                sy = ny >> 1; sx = nx >> 1;
                ok = oy[t] >= 0 && ny >= 0 && nx >= 0 && !(ny & 1) && !(nx & 1) && sy < Hi && sx < Wi;
            }
            src[t] = ok ? (int64_t)sy * Wi + sx : -1;
        }
        const float* At = A + (int64_t)tap * Mpad * lda + (int64_t)mt0 * 16 * lda;
        for (int k0 = 0; k0 < K; k0 += 32) {
            Frag xb[CT];
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + g * 8 + e;
                    v[e] = (k < K && src[t] >= 0) ? Xb[(int64_t)k * HWi + src[t]] : 0.f;
                }
                xb[t] = split8(v);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (r < nrt) {
                    const int k = k0 + g * 8;
                    float v[8];
                    if (k < lda) {
                        const float4* sp = (const float4*)(At + (int64_t)(r * 16 + c) * lda + k);
                        const float4 q0 = sp[0], q1 = sp[1];
                        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 0.f;
                    }
                    const Frag a = split8(v);
#pragma unroll
                    for (int t = 0; t < CT; ++t) acc[r][t] = mfma3(a, xb[t], acc[r][t]);
                }
            }
        }
    }
    float* Yb = Y + (int64_t)b * M * HWo;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (r < nrt) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int64_t p = p0 + t * 16 + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = (mt0 + r) * 16 + g * 4 + i;
                    if (m < M && p < HWo) __builtin_nontemporal_store(acc[r][t][i], &Yb[(int64_t)m * HWo + p]);
                }
            }
        }
    }
}

// taps[tap][a][b] <- W[m][k][dy][dx]:  transpose == 0: (a, b) = (m, k), tap = dy * 3 + dx  (forward / weight-gradient order);
//                                      transpose != 0: (a, b) = (k, m), tap = 8 - (dy * 3 + dx)  (input gradient of a stride-1 conv;
//                                      with flip == 0 the tap order is kept: the stride-2 input gradient indexes taps itself)
__global__ __launch_bounds__(256) void k_conv3x3_taps(const float* __restrict__ W, float* __restrict__ taps, int M, int K, int transpose, int flip,
                                                      int to_weight) {
    const int64_t n = (int64_t)M * K * 9;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i % 9);
        const int64_t mk = i / 9;
        const int k = (int)(mk % K), m = (int)(mk / K);
        const int tap = flip ? 8 - t : t;
        const int64_t j = transpose ? ((int64_t)tap * K + k) * M + m : ((int64_t)tap * M + m) * K + k;
        if (to_weight) ((float*)W)[i] = taps[j];        // the inverse: tap-major gradient -> [m][k][3][3]
        else taps[j] = W[i];
    }
}

// ---- rows x map, map operand staged through LDS -------------------------------------------------------------------------------
// The direct form above reads the map operand with 4-byte loads (its contiguous axis is the output pixel, the MFMA wants 8
// consecutive k per lane).  Here a [32 k][256 px] step is loaded with 16-byte accesses, split to hi / lo bf16 and stored to LDS
// as it lies ([k][px], 8-byte stores); B fragments come back through the gfx950 transposing read ds_read_b64_tr_b16 (as in
// k_dynconv), two per plane and fragment.  One pass per 160 rows; the next step's loads are in flight during the MFMAs.
// Measured (8 images, tools/train_kernels_time.py): better than the direct form where the map operand is binarised on the fly
// (267 against 324 us), not otherwise (194 / 250 against 169 / 196 us) -- used for the binarised case only.  Also tried: the row
// operand pre-split into bf16 planes by the caller (no conversions per wave and step): no gain, three more launches per call.
constexpr int RX2_PITCH = 272;             // uint16 elements per LDS row: 256 pixels + 32 bytes (rows 0..3 of a read: disjoint bank windows)

template <bool BIN, bool VEC>
__global__ __launch_bounds__(512) void k_rows_x_map_lds(const float* __restrict__ A, int64_t a_batch_stride, int lda, int Mpad, int M, int K,
                                                           const float* __restrict__ X, float* __restrict__ Y, int64_t HW,
                                                           const float* __restrict__ bias, const float* add) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2][32][RX2_PITCH];            // [hi | lo][k][px]
    constexpr int RT = 5, CT = 4;                                  // per wave: 8 waves = 2 row halves x 4 pixel quarters
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wave = wv & 3, wr = wv >> 2;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int64_t pw = (int64_t)blockIdx.x * 256;                  // first pixel of the workgroup
    const int mt0 = blockIdx.z * (2 * RT) + wr * RT;
    const int nrt = min(RT, Mpad / 16 - mt0);                      // may be <= 0 for the second row half
    const float* Ab = A + b * a_batch_stride + (int64_t)mt0 * 16 * lda;
    const float* Xb = X + (int64_t)b * K * HW;
    const int kr = tid >> 6, p4 = (tid & 63) * 4;                  // staging: rows kr + 8 j (j < 4), 4 pixels at p4
    float4 st[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + kr + 8 * j;
            const int64_t p = pw + p4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K) {
                const float* src = Xb + (int64_t)k * HW + p;
                if (VEC) { if (p < HW) v = *(const float4*)src; }
                else {
                    if (p + 0 < HW) v.x = src[0];
                    if (p + 1 < HW) v.y = src[1];
                    if (p + 2 < HW) v.z = src[2];
                    if (p + 3 < HW) v.w = src[3];
                }
                if (BIN) {
                    v.x = p + 0 < HW ? binz(v.x) : 0.f; v.y = p + 1 < HW ? binz(v.y) : 0.f;
                    v.z = p + 2 < HW ? binz(v.z) : 0.f; v.w = p + 3 < HW ? binz(v.w) : 0.f;
                }
            }
            st[j] = v;
        }
    };
    f32x4_t acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t h[4], l[4];
            f2bf_split(st[j].x, h[0], l[0]); f2bf_split(st[j].y, h[1], l[1]); f2bf_split(st[j].z, h[2], l[2]); f2bf_split(st[j].w, h[3], l[3]);
            *(uint2*)&lds[0][kr + 8 * j][p4] = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
            *(uint2*)&lds[1][kr + 8 * j][p4] = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
        }
        __syncthreads();
        if (k0 + 32 < K) fetch(k0 + 32);
        Frag xb[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            // lane (g, i = c): rows g * 8 + (i >> 2) [+ 4], pixels of this wave's column tile t at (i & 3) * 4
            const int px = wave * 64 + t * 16 + (c & 3) * 4;
            const int row = g * 8 + (c >> 2);
            const uint2 h0 = lds_read_tr16(&lds[0][row][px]), h1 = lds_read_tr16(&lds[0][row + 4][px]);
            const uint2 l0 = lds_read_tr16(&lds[1][row][px]), l1 = lds_read_tr16(&lds[1][row + 4][px]);
            xb[t].hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
            xb[t].lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            if (r < nrt) {
                const int k = k0 + g * 8;
                float v[8];
                if (k < lda) {
                    const float4* src = (const float4*)(Ab + (int64_t)(r * 16 + c) * lda + k);
                    const float4 q0 = src[0], q1 = src[1];
                    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
                const Frag a = split8(v);
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[r][t] = mfma3(a, xb[t], acc[r][t]);
            }
        }
    }
    float* Yb = Y + (int64_t)b * M * HW;
    const int64_t p0 = pw + wave * 64;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (r < nrt) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int64_t p = p0 + t * 16 + c;
                {
                    // every load of the epilogue's extras before the first store of the group: interleaved, each load would wait
                    // behind the previous store (possible aliasing) -- a dependent memory round trip per element
                    float ev[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = (mt0 + r) * 16 + g * 4 + i;
                        ev[i] = acc[r][t][i];
                        if (m < M && p < HW) {
                            if (bias) ev[i] += bias[(int64_t)b * M + m];
                            if (add) ev[i] += add[((int64_t)b * M + m) * HW + p];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = (mt0 + r) * 16 + g * 4 + i;
                        if (m < M && p < HW) __builtin_nontemporal_store(ev[i], &Yb[(int64_t)m * HW + p]);
                    }
                }
            }
        }
    }
}

// ---- map x map^T ---------------------------------------------------------------------------------------------------------------
// grid (nsplit, B, m-passes); 8 waves = 2 row halves (MXM_RT / 2 row tiles of m each) x 4 column groups (64 of the K <= 256 columns).
// Both operands are contraction-contiguous, so a 32-pixel step of (up to 160 G rows + 256 X rows) is staged ONCE per workgroup:
// global fp32 -> registers (the next step's loads are in flight during the MFMAs) -> hi / lo bf16 -> LDS rows of 32 pixels
// (80-byte pitch: the 16 rows x 16 bytes of a fragment read hit 16 distinct bank groups) -> fragments by ds_read_b128.
// (First version: every wave converted its own fragments straight from global memory, G four times over: 1.1 - 1.5 TB/s.)
// partial [B][nsplit][M][K]; a fixed-order second pass sums the splits.
constexpr int MXM_RT = 10;                 // 160 rows of G per pass
constexpr int MXM_ROWS = 448;               // 160 rows of G + 256 rows of X, rounded up to the 64 rows one staging pass covers
constexpr int MXM_PITCH = 40;              // uint16 elements per LDS row: 32 pixels + 16 bytes of padding
constexpr int MXM_LOADS = MXM_ROWS / 64;   // float4 loads per thread and step (row = tid / 8 + 64 j, 4 pixels at (tid % 8) * 4)

// rs_partial (nullable, round 5): [B][nsplit][M] row sums of G over the split's pixels, from the staging threads' own fp32
// loads (binarised: the pixel COUNT of each hard mask; otherwise the gradient of a dynamic kernel's scalar bias)
// shift (round 5, the weight gradient of a 3x3 tap): X is a [K][Hi][Wi] map read at src_tap(q) for every pixel q of G's Ho x Wo map
struct TapShift {
    int on, Hi, Wi, Wo, stride, dy, dx;
};
template <bool VEC, bool BIN>
__global__ __launch_bounds__(512) void k_map_x_mapT(const float* __restrict__ G, const float* __restrict__ X, float* __restrict__ partial,
                                                       int M, int K, int64_t HW, int64_t chunk, float* __restrict__ rs_partial, const TapShift sh) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2][MXM_ROWS][MXM_PITCH];     // [hi | lo][row][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int s = blockIdx.x, nsplit = gridDim.x, b = blockIdx.y;
    const int m0 = blockIdx.z * (MXM_RT * 16);
    const int64_t pa = s * chunk, pe = min(HW, pa + chunk);
    const float* Gb = G + ((int64_t)b * M + m0) * HW;
    const int64_t HWx = sh.on ? (int64_t)sh.Hi * sh.Wi : HW;
    const float* Xb = X + (int64_t)b * K * HWx;
    const int r0 = tid >> 3, p4 = (tid & 7) * 4;
    // row j of this thread: LDS row r0 + 32 j; rows < 160 are G rows m0 + .., the others X rows
    float4 st[MXM_LOADS];
    float rsum[3] = {0.f, 0.f, 0.f};            // G rows r0, r0 + 64, r0 + 128 of this pass
    auto fetch = [&](int64_t p) {
#pragma unroll
        for (int j = 0; j < MXM_LOADS; ++j) {
            const int row = r0 + 64 * j;
            const bool isg = row < MXM_RT * 16;
            const int rr = isg ? row : row - MXM_RT * 16;
            const bool ok = isg ? (m0 + rr < M) : (rr < K);
            const float* src = (isg ? Gb : Xb) + (int64_t)rr * HW + p + p4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && sh.on && !isg) {          // a tap of a 3x3 conv: every pixel of the step computes its own source address
                const float* xr = Xb + (int64_t)rr * HWx;
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t q = p + p4 + u;
                    e[u] = 0.f;
                    if (q < pe) {
                        const int sy = (int)(q / sh.Wo) * sh.stride + sh.dy - 1, sx = (int)(q % sh.Wo) * sh.stride + sh.dx - 1;
                        if (sy >= 0 && sy < sh.Hi && sx >= 0 && sx < sh.Wi) e[u] = xr[(int64_t)sy * sh.Wi + sx];
                    }
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            } else if (ok) {
                if (VEC) { if (p + p4 < pe) v = *(const float4*)src; }      // pe, HW multiples of 4 here
                else {
                    if (p + p4 + 0 < pe) v.x = src[0];
                    if (p + p4 + 1 < pe) v.y = src[1];
                    if (p + p4 + 2 < pe) v.z = src[2];
                    if (p + p4 + 3 < pe) v.w = src[3];
                }
                if (BIN && isg) {
                    v.x = p + p4 + 0 < pe ? binz(v.x) : 0.f; v.y = p + p4 + 1 < pe ? binz(v.y) : 0.f;
                    v.z = p + p4 + 2 < pe ? binz(v.z) : 0.f; v.w = p + p4 + 3 < pe ? binz(v.w) : 0.f;
                }
            }
            st[j] = v;
            if (j < 3 && isg) rsum[j] += (v.x + v.y) + (v.z + v.w);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < MXM_LOADS; ++j) {
            const int row = r0 + 64 * j;
            uint32_t h[4], l[4];
            f2bf_split(st[j].x, h[0], l[0]); f2bf_split(st[j].y, h[1], l[1]); f2bf_split(st[j].z, h[2], l[2]); f2bf_split(st[j].w, h[3], l[3]);
            *(uint2*)&lds[0][row][p4] = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
            *(uint2*)&lds[1][row][p4] = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
        }
    };
    constexpr int WRT = MXM_RT / 2;             // row tiles per wave
    const int wr = wave >> 2, wc = wave & 3;
    f32x4_t acc[WRT][4];
#pragma unroll
    for (int r = 0; r < WRT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nrt = min(WRT, (M - m0 + 15) / 16 - wr * WRT);      // valid row tiles of this wave (may be <= 0)
    if (pa < pe) fetch(pa);
    for (int64_t p = pa; p < pe; p += 32) {
        __syncthreads();                        // every wave is done with the previous step's fragments
        stash();
        __syncthreads();
        if (p + 32 < pe) fetch(p + 32);         // in flight during the MFMAs
        Frag xb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = MXM_RT * 16 + wc * 64 + t * 16 + c;
            xb[t].hi = *(const uint4*)&lds[0][row][g * 8];
            xb[t].lo = *(const uint4*)&lds[1][row][g * 8];
        }
#pragma unroll
        for (int r = 0; r < WRT; ++r) {
            if (r < nrt) {                      // wave-uniform
                Frag a;
                a.hi = *(const uint4*)&lds[0][(wr * WRT + r) * 16 + c][g * 8];
                a.lo = *(const uint4*)&lds[1][(wr * WRT + r) * 16 + c][g * 8];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[r][t] = mfma3(a, xb[t], acc[r][t]);
            }
        }
    }
    float* out = partial + ((int64_t)b * nsplit + s) * M * K;
#pragma unroll
    for (int r = 0; r < WRT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = wc * 64 + t * 16 + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + (wr * WRT + r) * 16 + g * 4 + i;
                if (r < nrt && m < M && k < K) out[(int64_t)m * K + k] = acc[r][t][i];
            }
        }
    if (rs_partial) {           // the 8 threads of a row are 8 adjacent lanes: xor-1/2/4 butterfly, lane 0 of the group writes
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = rsum[j];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            const int row = r0 + 64 * j, m = m0 + row;
            if ((tid & 7) == 0 && row < MXM_RT * 16 && m < M) rs_partial[((int64_t)b * nsplit + s) * M + m] = v;
        }
    }
}

// the number of pixels of each hard mask, [rows]: one wave per row (used where no pooling pass delivers it)
__global__ __launch_bounds__(256) void k_hard_count(const float* __restrict__ z, float* __restrict__ out, int64_t rows, int64_t HW) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* p = z + r * HW;
    int n = 0;
    for (int64_t i = threadIdx.x & 63; i < HW; i += 64) n += p[i] > PH_BIN_THR ? 1 : 0;
#pragma unroll
    for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) out[r] = (float)n;
}

__global__ __launch_bounds__(256) void k_sum_splits(const float* __restrict__ partial, float* __restrict__ out, int nsplit, int64_t MK, int B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= MK * B) return;
    const int64_t b = i / MK, j = i - b * MK;
    const float* p = partial + b * nsplit * MK + j;
    // fixed order (run-to-run identical): four interleaved partial sums so that 8 loads are in flight per thread -- the plain
    // one-accumulator loop read the 40 MB of a cfg2 pooling pass at 1 TB/s (39 us per call, 1.1 ms per training step)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        const float v0 = p[(int64_t)(s + 0) * MK], v1 = p[(int64_t)(s + 1) * MK], v2 = p[(int64_t)(s + 2) * MK], v3 = p[(int64_t)(s + 3) * MK];
        const float v4 = p[(int64_t)(s + 4) * MK], v5 = p[(int64_t)(s + 5) * MK], v6 = p[(int64_t)(s + 6) * MK], v7 = p[(int64_t)(s + 7) * MK];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        a0 += v4; a1 += v5; a2 += v6; a3 += v7;
    }
    for (; s < nsplit; ++s) a0 += p[(int64_t)s * MK];
    out[i] = (a0 + a1) + (a2 + a3);
}

// ---- transpose of the x2 bilinear upsample -----------------------------------------------------------------------------------------
// forward (align_corners=False): out[2j] = .25 in[j-1] + .75 in[j] (out[0] = in[0]), out[2j+1] = .75 in[j] + .25 in[j+1] (clamped).
// Hence d in[j] gathers output rows 2j-1 .. 2j+2 with weights (.25, .75, .75, .25); at the borders the tap that falls outside
// folds onto its neighbour (j = 0: (0, 1, .75, .25); j = H-1: (.25, .75, 1, 0)).  Separable in y and x.
__device__ __forceinline__ void taps(int j, int n, float w[4]) {
    w[0] = j > 0 ? 0.25f : 0.f;
    w[1] = j > 0 ? 0.75f : 1.f;
    w[2] = j < n - 1 ? 0.75f : 1.f;
    w[3] = j < n - 1 ? 0.25f : 0.f;
}

__global__ __launch_bounds__(256) void k_upsample2x_bwd(const float* __restrict__ g, float* __restrict__ out, int64_t planes, int H, int W) {
    const int64_t total = planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int64_t pl = i / ((int64_t)W * H);
        float wy[4], wx[4];
        taps(y, H, wy);
        taps(x, W, wx);
        const float* gp = g + pl * 4 * H * W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int Y = 2 * y - 1 + a;
            if (wy[a] == 0.f) continue;
            float row = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int X = 2 * x - 1 + c;
                if (wx[c] != 0.f) row += wx[c] * gp[(int64_t)Y * 2 * W + X];
            }
            acc += wy[a] * row;
        }
        out[i] = acc;
    }
}

}  // namespace

extern "C" int ph_rows_x_map_ex(const float* A, int64_t a_batch_stride, int lda, int Mpad, int M, int K, const float* X, float* Y, int B,
                                int64_t HW, int binarize_x, const float* bias /* [B][M] or null */, const float* add /* [B][M][HW], may alias Y, or null */,
                                void* stream) {
    PH_CHECK_ARG(A && X && Y && B > 0 && M > 0 && K > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG(Mpad % 16 == 0 && Mpad >= M && lda % 8 == 0 && lda >= K && (a_batch_stride % 4) == 0, "A must be zero padded: rows to 16, row stride to 8");
    PH_CHECK_ARG(((uintptr_t)A & 15) == 0, "A must be 16-byte aligned");
    const int tiles = Mpad / 16;
    hipStream_t s = (hipStream_t)stream;
#define PH_RXM(BIN, CT, RT)                                                                                               \
    hipLaunchKernelGGL((k_rows_x_map<BIN, CT, RT>), dim3((unsigned)((HW + 64 * CT - 1) / (64 * CT)), B, (tiles + RT - 1) / RT), \
                       dim3(256), 0, s, A, a_batch_stride, lda, Mpad, M, K, X, Y, HW, bias, add)
    if (binarize_x) {           // the map operand staged through LDS (one comparison per element, 16-byte loads)
        const dim3 grid((unsigned)((HW + 255) / 256), B, (tiles + 9) / 10);
        if ((HW % 4) == 0 && ((uintptr_t)X & 15) == 0)
            hipLaunchKernelGGL((k_rows_x_map_lds<true, true>), grid, dim3(512), 0, s, A, a_batch_stride, lda, Mpad, M, K, X, Y, HW, bias, add);
        else
            hipLaunchKernelGGL((k_rows_x_map_lds<true, false>), grid, dim3(512), 0, s, A, a_batch_stride, lda, Mpad, M, K, X, Y, HW, bias, add);
    } else PH_RXM(false, 4, 10);
#undef PH_RXM
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_rows_x_map(const float* A, int64_t a_batch_stride, int lda, int Mpad, int M, int K, const float* X, float* Y, int B,
                             int64_t HW, int binarize_x, void* stream) {
    return ph_rows_x_map_ex(A, a_batch_stride, lda, Mpad, M, K, X, Y, B, HW, binarize_x, nullptr, nullptr, stream);
}

extern "C" int ph_hard_count(const float* logits, float* out, int64_t rows, int64_t HW, void* stream) {
    PH_CHECK_ARG(logits && out && rows > 0 && HW > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_hard_count, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, out, rows, HW);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_map_x_map_t_nsplit(int B, int M, int64_t HW) {
    const int passes = ((M + 15) / 16 + MXM_RT - 1) / MXM_RT;
    int64_t want = 1024 / ((int64_t)B * passes);
    const int64_t most = (HW + 255) / 256;          // at least 256 pixels per split
    if (want > most) want = most;
    if (want < 1) want = 1;
    return (int)want;
}

extern "C" int ph_map_x_map_t_ex(const float* G, const float* X, float* partial /* [B][nsplit][M][K] */, float* out /* [B][M][K] */, int B,
                                int M, int K, int64_t HW, int nsplit, int binarize_g, float* rs_partial /* [B][nsplit][M] or null */,
                                float* rowsum /* [B][M] or null */, int sum_batch /* out [M][K], rowsum [M]: summed over the images too */,
                                void* stream) {
    PH_CHECK_ARG((rs_partial == nullptr) == (rowsum == nullptr), "rs_partial and rowsum go together");
    PH_CHECK_ARG(G && X && partial && out && B > 0 && M > 0 && K > 0 && K <= 256 && HW > 0 && nsplit >= 1, "bad pointer or size (K <= 256)");
    int64_t chunk = (HW + nsplit - 1) / nsplit;
    chunk = (chunk + 31) / 32 * 32;
    const bool vec = (HW % 4) == 0 && (((uintptr_t)G | (uintptr_t)X) & 15) == 0;
    const dim3 grid(nsplit, B, ((M + 15) / 16 + MXM_RT - 1) / MXM_RT);
    hipStream_t s = (hipStream_t)stream;
    const TapShift sh{};
#define PH_MXM(V, Bn) hipLaunchKernelGGL((k_map_x_mapT<V, Bn>), grid, dim3(512), 0, s, G, X, partial, M, K, HW, chunk, rs_partial, sh)
    if (vec) { if (binarize_g) PH_MXM(true, true); else PH_MXM(true, false); }
    else { if (binarize_g) PH_MXM(false, true); else PH_MXM(false, false); }
#undef PH_MXM
    PH_CHECK_LAUNCH();
    const int64_t MK = (int64_t)M * K;
    // sum_batch (the gradient of a STATIC 1x1 kernel): the B * nsplit records are one run of splits
    const int sB = sum_batch ? 1 : B, sN = sum_batch ? nsplit * B : nsplit;
    hipLaunchKernelGGL(k_sum_splits, dim3((unsigned)((MK * sB + 255) / 256)), dim3(256), 0, s, partial, out, sN, MK, sB);
    PH_CHECK_LAUNCH();
    if (rowsum) {
        hipLaunchKernelGGL(k_sum_splits, dim3((unsigned)(((int64_t)M * sB + 255) / 256)), dim3(256), 0, s, rs_partial, rowsum, sN, (int64_t)M, sB);
        PH_CHECK_LAUNCH();
    }
    return PH_OK;
}

extern "C" int ph_map_x_map_t(const float* G, const float* X, float* partial, float* out, int B, int M, int K, int64_t HW, int nsplit,
                             int binarize_g, void* stream) {
    return ph_map_x_map_t_ex(G, X, partial, out, B, M, K, HW, nsplit, binarize_g, nullptr, nullptr, 0, stream);
}

extern "C" int ph_upsample2x_bwd(const float* grad_out /* [planes][2H][2W] */, float* grad_in /* [planes][H][W] */, int64_t planes, int H,
                                 int W, void* stream) {
    PH_CHECK_ARG(grad_out && grad_in && planes > 0 && H > 0 && W > 0, "bad pointer or size");
    const int64_t total = planes * H * W;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_upsample2x_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_out, grad_in, planes, H, W);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---- 3x3 convolution of the neck's towers in training (semantic_fpn.py:75-150: ConvModule 3x3, padding 1, stride 1 or 2, no bias) --
extern "C" int ph_conv3x3_taps(const float* W /* [M][K][3][3] */, float* taps /* [9][..][..] */, int M, int K, int transpose, int flip,
                               int to_weight, void* stream) {
    PH_CHECK_ARG(W && taps && M > 0 && K > 0, "bad pointer or size");
    const int64_t n = (int64_t)M * K * 9;
    hipLaunchKernelGGL(k_conv3x3_taps, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, taps, M,
                       K, transpose, flip, to_weight);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_conv3x3_train(const float* taps /* [9][M][K], M % 16 == 0, K % 8 == 0 */, int M, int K, const float* X, float* Y, int B, int Hi,
                                int Wi, int Ho, int Wo, int stride, int mode, void* stream) {
    PH_CHECK_ARG(taps && X && Y && B > 0 && M > 0 && K > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bad pointer or size");
    PH_CHECK_ARG(M % 16 == 0 && K % 8 == 0 && ((uintptr_t)taps & 15) == 0, "M % 16, K % 8 (the shipped towers: 256 x 256)");
    PH_CHECK_ARG((mode == 0 && (stride == 1 || stride == 2)) || (mode == 1 && stride == 2), "mode 0: stride 1 / 2 forward; mode 1: stride-2 input gradient");
    const int64_t HWo = (int64_t)Ho * Wo;
    constexpr int CT = 4, RT = 8;
    hipLaunchKernelGGL((k_conv3x3_rows<CT, RT>), dim3((unsigned)((HWo + 64 * CT - 1) / (64 * CT)), B, (M / 16 + RT - 1) / RT), dim3(256), 0,
                       (hipStream_t)stream, taps, K, M, M, K, X, Y, Hi, Wi, Ho, Wo, stride, mode);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// dW taps [9][M][K] = sum_b sum_q dY[b][m][q] X[b][k][src_tap(q)]  (nine shifted map x map^T products, summed over the batch)
extern "C" int ph_conv3x3_wgrad(const float* dY /* [B][M][Ho][Wo] */, const float* X /* [B][K][Hi][Wi] */, float* partial, float* taps_out, int B,
                                int M, int K, int Hi, int Wi, int Ho, int Wo, int stride, int nsplit, void* stream) {
    PH_CHECK_ARG(dY && X && partial && taps_out && B > 0 && M > 0 && K > 0 && K <= 256 && nsplit >= 1, "bad pointer or size (K <= 256)");
    const int64_t HW = (int64_t)Ho * Wo;
    int64_t chunk = (HW + nsplit - 1) / nsplit;
    chunk = (chunk + 31) / 32 * 32;
    const bool vec = (HW % 4) == 0 && ((uintptr_t)dY & 15) == 0;
    const dim3 grid(nsplit, B, ((M + 15) / 16 + MXM_RT - 1) / MXM_RT);
    hipStream_t s = (hipStream_t)stream;
    const int64_t MK = (int64_t)M * K;
    for (int tap = 0; tap < 9; ++tap) {
        const TapShift sh{1, Hi, Wi, Wo, stride, tap / 3, tap % 3};
        if (vec) hipLaunchKernelGGL((k_map_x_mapT<true, false>), grid, dim3(512), 0, s, dY, X, partial, M, K, HW, chunk, (float*)nullptr, sh);
        else hipLaunchKernelGGL((k_map_x_mapT<false, false>), grid, dim3(512), 0, s, dY, X, partial, M, K, HW, chunk, (float*)nullptr, sh);
        hipLaunchKernelGGL(k_sum_splits, dim3((unsigned)((MK + 255) / 256)), dim3(256), 0, s, partial, taps_out + tap * MK, nsplit * B, MK, 1);
    }
    PH_CHECK_LAUNCH();
    return PH_OK;
}
