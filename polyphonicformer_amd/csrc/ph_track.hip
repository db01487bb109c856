// SURVEY.md 8f row N1 -- device side of the video association step (polyphonic_former_video.py:359-396):
//   ph_segment_boxes : panoptic id map -> per thing segment the RoI box (centre +- 2 * mean |deviation|,
//                      polyphonic/video/utils.py:39-83) and the tight extent box (funcs/utils.py:4-22)
//   ph_roi_align_fpn : SingleRoIExtractor (FPN level by sqrt(area)) + mmcv RoIAlign(7, sampling_ratio 2, avg, aligned)
//   ph_gemm_rows     : Y = act(X W^T + b) on bf16 MFMA with packed weight fragments (conv3x3-as-GEMM, fc, fc_embed)
//   ph_im2col7 / ph_gn_relu_cl : 3x3 patches of the 7x7 RoI maps / per-sample GroupNorm(32) + ReLU, channels-last
// The tensors are tiny (<= 100 RoIs): these kernels are latency bound and written for simplicity, not roofline.
#include "ph_common.h"

// ---------------------------------------------------------------------------------------------------------------
// segment statistics.  st [nseg][3] = count, sum_row, sum_col ; emin [nseg][2] = min_r, min_c ; emax [nseg][2] = max_r, max_c
// Per-block privatisation: a block accumulates its pixels in LDS (integer atomics) and adds its non-empty rows to the global
// record once.  With atomics straight to global memory every pixel of a segment hit the same few addresses: 80-100 ms per
// 1024x2048 map.  All sums are integers (exact, order free); the mean absolute deviations are accumulated in fp64.
constexpr int SEG_LDS_MAX = 1024;          // segments per id map this path handles (ids beyond: global-atomic fallback)

// VEC consecutive pixels of one image row per thread (VEC = 8 when W % 8 == 0: two 16-byte loads); a run that lies inside one
// segment -- nearly all of them -- is added as ONE record (count VEC, VEC * row, the arithmetic series of its columns, its end
// points), so the atomics per pixel drop by VEC; runs that straddle a boundary fall back to per-pixel updates.
struct SegAcc {
    unsigned int* lst; int* lmin; int* lmax; int nl;
    unsigned long long* st; int* emin; int* emax;
    __device__ __forceinline__ void add(int s, int n, int sr, int sc, int r, int c_lo, int c_hi) const {
        if (s < nl) {
            atomicAdd(&lst[s * 3 + 0], (unsigned)n); atomicAdd(&lst[s * 3 + 1], (unsigned)sr); atomicAdd(&lst[s * 3 + 2], (unsigned)sc);
            atomicMin(&lmin[s * 2 + 0], r); atomicMin(&lmin[s * 2 + 1], c_lo);
            atomicMax(&lmax[s * 2 + 0], r); atomicMax(&lmax[s * 2 + 1], c_hi);
        } else {
            atomicAdd(&st[s * 3 + 0], (unsigned long long)n); atomicAdd(&st[s * 3 + 1], (unsigned long long)sr);
            atomicAdd(&st[s * 3 + 2], (unsigned long long)sc);
            atomicMin(&emin[s * 2 + 0], r); atomicMin(&emin[s * 2 + 1], c_lo);
            atomicMax(&emax[s * 2 + 0], r); atomicMax(&emax[s * 2 + 1], c_hi);
        }
    }
};

template <int VEC>
__device__ __forceinline__ void seg_load(const int* __restrict__ pan, int64_t p, int (&id)[VEC]) {
    if constexpr (VEC == 8) {
        const int4 a = *(const int4*)(pan + p), b = *(const int4*)(pan + p + 4);
        id[0] = a.x; id[1] = a.y; id[2] = a.z; id[3] = a.w; id[4] = b.x; id[5] = b.y; id[6] = b.z; id[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) id[j] = pan[p + j];
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_seg_stats(const int* __restrict__ pan, int H, int W, int nseg,
                                                   unsigned long long* __restrict__ st, int* __restrict__ emin, int* __restrict__ emax) {
    extern __shared__ unsigned int lst[];          // [nl][3] count, sum_row, sum_col | [nl][2] min | [nl][2] max
    const int nl = nseg < SEG_LDS_MAX ? nseg : SEG_LDS_MAX;
    int* lmin = (int*)(lst + 3 * nl);
    int* lmax = lmin + 2 * nl;
    for (int k = threadIdx.x; k < 3 * nl; k += blockDim.x) lst[k] = 0u;
    for (int k = threadIdx.x; k < 2 * nl; k += blockDim.x) { lmin[k] = 0x7f7f7f7f; lmax[k] = 0; }
    __syncthreads();
    const SegAcc acc{lst, lmin, lmax, nl, st, emin, emax};
    const int64_t nrun = (int64_t)H * W / VEC;     // VEC divides W
    // a block's share is < 2^20 pixels with coordinates < 2^16 -> its coordinate sums fit 32 bits as long as
    // pixels-per-block * max-coordinate < 2^32 (checked by the launcher)
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nrun; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = q * VEC;
        const int r = (int)(p / W), c0 = (int)(p - (int64_t)r * W);
        int id[VEC];
        seg_load<VEC>(pan, p, id);
        bool uni = true;
#pragma unroll
        for (int j = 1; j < VEC; ++j) uni = uni && id[j] == id[0];
        if (uni) {
            if (id[0] >= 1 && id[0] <= nseg) acc.add(id[0] - 1, VEC, VEC * r, VEC * c0 + VEC * (VEC - 1) / 2, r, c0, c0 + VEC - 1);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (id[j] >= 1 && id[j] <= nseg) acc.add(id[j] - 1, 1, r, c0 + j, r, c0 + j, c0 + j);
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nl; s += blockDim.x) {
        if (!lst[s * 3]) continue;
        atomicAdd(&st[s * 3 + 0], (unsigned long long)lst[s * 3 + 0]);
        atomicAdd(&st[s * 3 + 1], (unsigned long long)lst[s * 3 + 1]);
        atomicAdd(&st[s * 3 + 2], (unsigned long long)lst[s * 3 + 2]);
        atomicMin(&emin[s * 2 + 0], lmin[s * 2 + 0]); atomicMin(&emin[s * 2 + 1], lmin[s * 2 + 1]);
        atomicMax(&emax[s * 2 + 0], lmax[s * 2 + 0]); atomicMax(&emax[s * 2 + 1], lmax[s * 2 + 1]);
    }
}
// dev [nseg][2] (double): sum |row - mean_row|, sum |col - mean_col| with the fp32 means the reference uses
template <int VEC>
__global__ __launch_bounds__(256) void k_seg_absdev(const int* __restrict__ pan, int H, int W, int nseg,
                                                    const unsigned long long* __restrict__ st, double* __restrict__ dev) {
    extern __shared__ double ldev[];               // [nl][2]
    const int nl = nseg < SEG_LDS_MAX ? nseg : SEG_LDS_MAX;
    for (int k = threadIdx.x; k < 2 * nl; k += blockDim.x) ldev[k] = 0.0;
    __syncthreads();
    const int64_t nrun = (int64_t)H * W / VEC;
    auto means = [&](int s, float& mr, float& mc) {
        const double n = (double)st[s * 3];
        mr = (float)((double)st[s * 3 + 1] / n);
        mc = (float)((double)st[s * 3 + 2] / n);
    };
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nrun; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = q * VEC;
        const int r = (int)(p / W), c0 = (int)(p - (int64_t)r * W);
        int id[VEC];
        seg_load<VEC>(pan, p, id);
        bool uni = true;
#pragma unroll
        for (int j = 1; j < VEC; ++j) uni = uni && id[j] == id[0];
        if (uni) {
            if (id[0] < 1 || id[0] > nseg) continue;
            const int s = id[0] - 1;
            float mr, mc;
            means(s, mr, mc);
            const double dr = (double)VEC * (double)fabsf((float)r - mr);      // VEC equal terms (exact: a power of two)
            double dc = 0.0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) dc += (double)fabsf((float)(c0 + j) - mc);
            double* d = s < nl ? ldev : dev;
            atomicAdd(&d[s * 2 + 0], dr);
            atomicAdd(&d[s * 2 + 1], dc);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (id[j] < 1 || id[j] > nseg) continue;
                const int s = id[j] - 1;
                float mr, mc;
                means(s, mr, mc);
                double* d = s < nl ? ldev : dev;
                atomicAdd(&d[s * 2 + 0], (double)fabsf((float)r - mr));
                atomicAdd(&d[s * 2 + 1], (double)fabsf((float)(c0 + j) - mc));
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * nl; k += blockDim.x)
        if (ldev[k] != 0.0) atomicAdd(&dev[k], ldev[k]);
}
__global__ void k_seg_init(unsigned long long* __restrict__ st, double* __restrict__ dev, int* __restrict__ emin, int* __restrict__ emax, int nseg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    st[s * 3] = st[s * 3 + 1] = st[s * 3 + 2] = 0ull;
    dev[s * 2] = dev[s * 2 + 1] = 0.0;
    emin[s * 2] = emin[s * 2 + 1] = 0x7f7f7f7f;      // > any coordinate
    emax[s * 2] = emax[s * 2 + 1] = 0;
}
// rois [nseg][5] = (0, x1, y1, x2, y2) clamped at 0 ; ext_boxes [nseg][4] xyxy
__global__ void k_seg_boxes(const unsigned long long* __restrict__ st, const int* __restrict__ emin, const int* __restrict__ emax,
                            const double* __restrict__ dev,
                            int nseg, float* __restrict__ rois, float* __restrict__ ext_boxes) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const double n = (double)st[s * 3];
    if (n == 0) {          // empty mask: [0,0,0,0] (video/utils.py:75) and (-1,-1,10,10) (funcs/utils.py:19)
        for (int k = 0; k < 5; ++k) rois[s * 5 + k] = 0.f;
        ext_boxes[s * 4 + 0] = -1.f; ext_boxes[s * 4 + 1] = -1.f; ext_boxes[s * 4 + 2] = 10.f; ext_boxes[s * 4 + 3] = 10.f;
        return;
    }
    const float mr = (float)((double)st[s * 3 + 1] / n), mc = (float)((double)st[s * 3 + 2] / n);
    const float dr = fmaxf((float)(dev[s * 2] / n), 1.f), dc = fmaxf((float)(dev[s * 2 + 1] / n), 1.f);
    rois[s * 5 + 0] = 0.f;
    rois[s * 5 + 1] = fmaxf(mc - dc * 2.f, 0.f);
    rois[s * 5 + 2] = fmaxf(mr - dr * 2.f, 0.f);
    rois[s * 5 + 3] = fmaxf(mc + dc * 2.f, 0.f);
    rois[s * 5 + 4] = fmaxf(mr + dr * 2.f, 0.f);
    ext_boxes[s * 4 + 0] = (float)emin[s * 2 + 1];      // x1 = min col
    ext_boxes[s * 4 + 1] = (float)emin[s * 2 + 0];      // y1 = min row
    ext_boxes[s * 4 + 2] = (float)emax[s * 2 + 1];
    ext_boxes[s * 4 + 3] = (float)emax[s * 2 + 0];
}

extern "C" size_t ph_segment_boxes_workspace_bytes(int nseg) {
    return (size_t)nseg * (3 * sizeof(unsigned long long) + 4 * sizeof(int) + 2 * sizeof(double));
}

extern "C" int ph_segment_boxes(const int32_t* pan, int H, int W, int nseg, float* rois, float* ext_boxes, void* workspace,
                                size_t workspace_bytes, void* stream) {
    PH_CHECK_ARG(pan && rois && ext_boxes && workspace && H > 0 && W > 0 && nseg > 0, "bad pointer or size");
    PH_CHECK_ARG(workspace_bytes >= ph_segment_boxes_workspace_bytes(nseg), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* st = (unsigned long long*)workspace;
    double* dev = (double*)(st + 3 * nseg);
    int* emin = (int*)(dev + 2 * nseg);
    int* emax = emin + 2 * nseg;
    // one kernel instead of three memset nodes (a captured hipMemsetAsync node misbehaves on replay, see ph_khead1.hip)
    hipLaunchKernelGGL(k_seg_init, dim3((nseg + 63) / 64), dim3(64), 0, s, st, dev, emin, emax, nseg);
    const int64_t npx = (int64_t)H * W;
    const int vec = (W % 8 == 0 && ((uintptr_t)pan & 15) == 0) ? 8 : 1;
    const int64_t nrun = npx / vec;
    int grid = (int)((nrun + 255) / 256 < 2048 ? (nrun + 255) / 256 : 2048);
    // a block's coordinate sums are 32-bit in LDS: pixels per block x largest coordinate must stay below 2^32
    PH_CHECK_ARG(((nrun + grid - 1) / grid + 256) * vec * (int64_t)(H > W ? H : W) < (1ll << 32), "id map too large");
    const int nl = nseg < SEG_LDS_MAX ? nseg : SEG_LDS_MAX;
    if (vec == 8) {
        hipLaunchKernelGGL(k_seg_stats<8>, dim3(grid), dim3(256), (size_t)nl * 7 * sizeof(int), s, pan, H, W, nseg, st, emin, emax);
        hipLaunchKernelGGL(k_seg_absdev<8>, dim3(grid), dim3(256), (size_t)nl * 2 * sizeof(double), s, pan, H, W, nseg, st, dev);
    } else {
        hipLaunchKernelGGL(k_seg_stats<1>, dim3(grid), dim3(256), (size_t)nl * 7 * sizeof(int), s, pan, H, W, nseg, st, emin, emax);
        hipLaunchKernelGGL(k_seg_absdev<1>, dim3(grid), dim3(256), (size_t)nl * 2 * sizeof(double), s, pan, H, W, nseg, st, dev);
    }
    hipLaunchKernelGGL(k_seg_boxes, dim3((nseg + 63) / 64), dim3(64), 0, s, st, emin, emax, dev, nseg, rois, ext_boxes);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// RoIAlign over an FPN.  out_cl: bf16 planes [P][n][49][256] (channels last, what the track head consumes);
// out_f32 (optional): [n][256][7][7] like the reference's roi_feats.
struct FpnArgs { const float* feat[4]; int H[4], W[4]; float scale[4]; int nlev; };

__device__ __forceinline__ float roi_bilinear(const float* __restrict__ f, int H, int W, float y, float x) {
    if (y < -1.f || y > (float)H || x < -1.f || x > (float)W) return 0.f;
    y = fmaxf(y, 0.f); x = fmaxf(x, 0.f);
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
    return hy * hx * f[yl * W + xl] + hy * lx * f[yl * W + xh] + ly * hx * f[yh * W + xl] + ly * lx * f[yh * W + xh];
}

template <int PA>
__global__ __launch_bounds__(256) void k_roi_align_fpn(FpnArgs a, const float* __restrict__ rois, int n, float finest,
                                                        uint16_t* __restrict__ out_cl, float* __restrict__ out_f32) {
    // one block per (RoI, output bin), one thread per channel: 49 n blocks of 16 scattered loads per thread (the planes are NCHW, a
    // wave's 64 channels are 64 cache lines per tap) instead of n blocks walking 784 of them -- same arithmetic per bin
    const int roi = blockIdx.x, c = threadIdx.x, ph = blockIdx.y / 7, pw = blockIdx.y - 7 * (blockIdx.y / 7);
    const float* r = rois + roi * 5;
    const float sc = sqrtf((r[3] - r[1]) * (r[4] - r[2]));
    int lv = (int)floorf(log2f(sc / finest + 1e-6f));
    lv = lv < 0 ? 0 : (lv > a.nlev - 1 ? a.nlev - 1 : lv);
    const float s = a.scale[lv];
    const int H = a.H[lv], W = a.W[lv];
    const float* f = a.feat[lv] + (int64_t)c * H * W;
    const float x1 = r[1] * s - 0.5f, y1 = r[2] * s - 0.5f, x2 = r[3] * s - 0.5f, y2 = r[4] * s - 0.5f;
    const float bw = (x2 - x1) / 7.f, bh = (y2 - y1) / 7.f;
    const int64_t plane = (int64_t)n * 49 * 256;
    float acc = 0.f;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
        const float yy = y1 + ph * bh + (iy + 0.5f) * bh / 2.f;
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) acc += roi_bilinear(f, H, W, yy, x1 + pw * bw + (ix + 0.5f) * bw / 2.f);
    }
    const float v = acc / 4.f;
    uint32_t hi, lo;
    f2bf_split(v, hi, lo);
    const int64_t o = ((int64_t)roi * 49 + ph * 7 + pw) * 256 + c;
    out_cl[o] = (uint16_t)hi;
    if (PA == 2) out_cl[o + plane] = (uint16_t)lo;
    if (out_f32) out_f32[((int64_t)roi * 256 + c) * 49 + ph * 7 + pw] = v;
}

extern "C" int ph_roi_align_fpn(const float* const* feats, const int32_t* hw /*[nlev][2]*/, const float* scales, int nlev,
                                const float* rois, int n, float finest_scale, uint16_t* out_cl, float* out_f32, int prec,
                                void* stream) {
    PH_CHECK_ARG(feats && hw && scales && rois && out_cl && nlev >= 1 && nlev <= 4 && n > 0, "bad pointer or size");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    FpnArgs a;
    a.nlev = nlev;
    for (int l = 0; l < nlev; ++l) { a.feat[l] = feats[l]; a.H[l] = hw[2 * l]; a.W[l] = hw[2 * l + 1]; a.scale[l] = scales[l]; }
    if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_roi_align_fpn<1>, dim3(n, 49), dim3(256), 0, (hipStream_t)stream, a, rois, n, finest_scale, out_cl, out_f32);
    else hipLaunchKernelGGL(k_roi_align_fpn<2>, dim3(n, 49), dim3(256), 0, (hipStream_t)stream, a, rois, n, finest_scale, out_cl, out_f32);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// generic row GEMM: Y[M][N] = act(X[M][K] W^T + b).  X: bf16 planes [PA][M][K] row major; W: packed B fragments
// (pack.pack_b_fragments, [N/16][K/32] blocks); Y: fp32 [M][N] and/or bf16 planes [PA][M][N].
template <int PA>
__global__ __launch_bounds__(256) void k_gemm_rows(const uint16_t* __restrict__ X, int64_t x_plane, const uint16_t* __restrict__ Wp,
                                                   int64_t w_plane, const float* __restrict__ bias, int relu,
                                                   float* __restrict__ Yf, uint16_t* __restrict__ Yp, int64_t y_plane,
                                                   int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 32, ct = blockIdx.y * 4 + wave;
    if (ct * 16 >= N) return;
    const int KS = K / 32;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const uint16_t* wb = Wp + ((int64_t)ct * KS) * 512 + lane * 8;
    for (int ks = 0; ks < KS; ++ks) {
        uint4 a[PA][2], b[PA];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            b[p] = *(const uint4*)(wb + p * w_plane + (int64_t)ks * 512);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = row0 + rt * 16 + i;
                a[p][rt] = row < M ? *(const uint4*)(X + p * x_plane + (int64_t)row * K + ks * 32 + g * 8) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            acc[rt] = mfma16(a[0][rt], b[0], acc[rt]);
            if (PA == 2) { acc[rt] = mfma16(a[0][rt], b[PA - 1], acc[rt]); acc[rt] = mfma16(a[PA - 1][rt], b[0], acc[rt]); }
        }
    }
    const int col = ct * 16 + i;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + rt * 16 + g * 4 + r;
            if (row >= M) continue;
            float v = acc[rt][r] + bv;
            if (relu) v = fmaxf(v, 0.f);
            if (Yf) Yf[(int64_t)row * N + col] = v;
            if (Yp) {
                uint32_t hi, lo;
                f2bf_split(v, hi, lo);
                Yp[(int64_t)row * N + col] = (uint16_t)hi;
                if (PA == 2) Yp[(int64_t)row * N + col + y_plane] = (uint16_t)lo;
            }
        }
}

extern "C" int ph_gemm_rows(const uint16_t* X, const uint16_t* Wp, int64_t w_plane_elems, const float* bias, int relu,
                            float* Yf, uint16_t* Yp, int M, int N, int K, int prec, void* stream) {
    PH_CHECK_ARG(X && Wp && (Yf || Yp) && M > 0 && N > 0 && K > 0, "bad pointer or size");
    PH_CHECK_ARG(N % 16 == 0 && K % 32 == 0, "N % 16 == 0 and K % 32 == 0 required");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    const dim3 grid((M + 31) / 32, (N / 16 + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_gemm_rows<1>, grid, dim3(256), 0, s, X, (int64_t)M * K, Wp, w_plane_elems, bias, relu, Yf, Yp, (int64_t)M * N, M, N, K);
    else hipLaunchKernelGGL(k_gemm_rows<2>, grid, dim3(256), 0, s, X, (int64_t)M * K, Wp, w_plane_elems, bias, relu, Yf, Yp, (int64_t)M * N, M, N, K);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// The same GEMM for the association step's shapes (a dozen RoIs: M = 49 n = 539 rows for the convs, M = n = 11 for the fc whose
// weight matrix is 51 MB): K is split over blockIdx.z so that several hundred workgroups stream the operands instead of 16-68, four
// k-steps of loads are issued before their MFMAs, and the 3x3 patches are gathered by the A-operand loads themselves (IM2COL: X is the
// channels-last [P][n][49][256] maps, k = tap * 256 + channel) instead of through a materialised [M][2304] matrix.  Every split
// accumulates its k-steps in order into part[z][M][N]; k_gemm_finish adds the splits in order: deterministic, and per element the same
// products as k_gemm_rows in a different association.  The split depends on (M, N, K) only.
template <int PA, bool IM2COL>
__global__ __launch_bounds__(256) void k_gemm_rows_sk(const uint16_t* __restrict__ X, int64_t x_plane, const uint16_t* __restrict__ Wp,
                                                      int64_t w_plane, float* __restrict__ part, int M, int N, int K, int steps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 32, ct = blockIdx.y * 4 + wave;
    if (ct * 16 >= N) return;
    const int KS = K / 32, ks0 = blockIdx.z * steps, ks1 = ks0 + steps < KS ? ks0 + steps : KS;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const uint16_t* wb = Wp + ((int64_t)ct * KS) * 512 + lane * 8;
    const uint16_t* xr[2];
    bool rv[2];
    int py[2], px[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int row = row0 + rt * 16 + i;
        rv[rt] = row < M;
        const int rc = rv[rt] ? row : 0;
        if (IM2COL) {
            const int smp = rc / 49, pos = rc - smp * 49;
            py[rt] = pos / 7; px[rt] = pos - 7 * py[rt];
            xr[rt] = X + (int64_t)smp * 49 * 256 + g * 8;
        } else {
            py[rt] = px[rt] = 0;
            xr[rt] = X + (int64_t)rc * K + g * 8;
        }
    }
    constexpr int U = 4;
    for (int kb = ks0; kb < ks1; kb += U) {
        uint4 a[U][PA][2], b[U][PA];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = kb + u;
            if (ks >= ks1) continue;                    // uniform
#pragma unroll
            for (int p = 0; p < PA; ++p) b[u][p] = *(const uint4*)(wb + p * w_plane + (int64_t)ks * 512);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                bool ok = rv[rt];
                int64_t off;
                if (IM2COL) {
                    const int tap = ks >> 3, y = py[rt] + tap / 3 - 1, x = px[rt] + tap % 3 - 1;
                    ok = ok && y >= 0 && y < 7 && x >= 0 && x < 7;
                    off = (int64_t)(y * 7 + x) * 256 + (ks & 7) * 32;
                } else off = (int64_t)ks * 32;
#pragma unroll
                for (int p = 0; p < PA; ++p) a[u][p][rt] = ok ? *(const uint4*)(xr[rt] + p * x_plane + off) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kb + u >= ks1) continue;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt] = mfma16(a[u][0][rt], b[u][0], acc[rt]);
                if (PA == 2) { acc[rt] = mfma16(a[u][0][rt], b[u][PA - 1], acc[rt]); acc[rt] = mfma16(a[u][PA - 1][rt], b[u][0], acc[rt]); }
            }
        }
    }
    float* out = part + (int64_t)blockIdx.z * M * N;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + rt * 16 + g * 4 + r;
            if (row < M) out[(int64_t)row * N + ct * 16 + i] = acc[rt][r];
        }
}

template <int PA>
__global__ __launch_bounds__(256) void k_gemm_finish(const float* __restrict__ part, int S, const float* __restrict__ bias, int relu,
                                                     float* __restrict__ Yf, uint16_t* __restrict__ Yp, int64_t y_plane, int M, int N) {
    const int64_t total = (int64_t)M * N;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        float v = part[idx];
        for (int z = 1; z < S; ++z) v += part[(int64_t)z * total + idx];
        if (bias) v += bias[idx % N];
        if (relu) v = fmaxf(v, 0.f);
        if (Yf) Yf[idx] = v;
        if (Yp) {
            uint32_t hi, lo;
            f2bf_split(v, hi, lo);
            Yp[idx] = (uint16_t)hi;
            if (PA == 2) Yp[idx + y_plane] = (uint16_t)lo;
        }
    }
}

static void gemm_split(int M, int N, int K, int& S, int& steps) {
    const int64_t tiles = (int64_t)((M + 31) / 32) * ((N / 16 + 3) / 4);
    const int KS = K / 32;
    steps = 8;
    S = (KS + steps - 1) / steps;
    while (S > 1 && tiles * S > 2048) { steps *= 2; S = (KS + steps - 1) / steps; }
}

extern "C" size_t ph_gemm_rows_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 32) return 0;
    int S, steps;
    gemm_split(M, N, K, S, steps);
    return (size_t)S * M * N * sizeof(float);
}

extern "C" int ph_gemm_rows_splitk(const uint16_t* X, int im2col7, const uint16_t* Wp, int64_t w_plane_elems, const float* bias, int relu,
                                   float* Yf, uint16_t* Yp, int M, int N, int K, int prec, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    PH_CHECK_ARG(X && Wp && (Yf || Yp) && workspace && M > 0 && N > 0 && K > 0, "bad pointer or size");
    PH_CHECK_ARG(N % 16 == 0 && K % 32 == 0, "N % 16 == 0 and K % 32 == 0 required");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    PH_CHECK_ARG(!im2col7 || (K == 2304 && M % 49 == 0), "im2col7: X is [P][M / 49][49][256] and K = 9 * 256");
    if (workspace_bytes < ph_gemm_rows_workspace_bytes(M, N, K)) { ph_set_error("ph_gemm_rows_splitk: workspace too small"); return PH_EWORKSPACE; }
    int S, steps;
    gemm_split(M, N, K, S, steps);
    const dim3 grid((M + 31) / 32, (N / 16 + 3) / 4, S);
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    const int64_t x_plane = im2col7 ? (int64_t)M * 256 : (int64_t)M * K;
    if (prec == PH_PREC_BF16) {
        if (im2col7) hipLaunchKernelGGL((k_gemm_rows_sk<1, true>), grid, dim3(256), 0, s, X, x_plane, Wp, w_plane_elems, part, M, N, K, steps);
        else hipLaunchKernelGGL((k_gemm_rows_sk<1, false>), grid, dim3(256), 0, s, X, x_plane, Wp, w_plane_elems, part, M, N, K, steps);
    } else {
        if (im2col7) hipLaunchKernelGGL((k_gemm_rows_sk<2, true>), grid, dim3(256), 0, s, X, x_plane, Wp, w_plane_elems, part, M, N, K, steps);
        else hipLaunchKernelGGL((k_gemm_rows_sk<2, false>), grid, dim3(256), 0, s, X, x_plane, Wp, w_plane_elems, part, M, N, K, steps);
    }
    const int64_t total = (int64_t)M * N;
    const int fgrid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_gemm_finish<1>, dim3(fgrid), dim3(256), 0, s, part, S, bias, relu, Yf, Yp, total, M, N);
    else hipLaunchKernelGGL(k_gemm_finish<2>, dim3(fgrid), dim3(256), 0, s, part, S, bias, relu, Yf, Yp, total, M, N);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 (pad 1) patches of 7x7 channels-last maps: in [P][n][49][256] -> out [P][n*49][9*256], K order (tap, channel)
__global__ __launch_bounds__(256) void k_im2col7(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int n, int P) {
    const int64_t chunks = (int64_t)P * n * 49 * 9 * 32;               // 16-byte chunks (8 channels)
    const int64_t in_plane = (int64_t)n * 49 * 256, out_plane = (int64_t)n * 49 * 2304;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx & 31);
        int64_t t = idx >> 5;
        const int tap = (int)(t % 9); t /= 9;
        const int pos = (int)(t % 49); t /= 49;
        const int s = (int)(t % n);
        const int p = (int)(t / n);
        const int y = pos / 7 + tap / 3 - 1, x = pos % 7 + tap % 3 - 1;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (y >= 0 && y < 7 && x >= 0 && x < 7) v = *(const uint4*)(in + p * in_plane + ((int64_t)s * 49 + y * 7 + x) * 256 + c8 * 8);
        *(uint4*)(out + p * out_plane + ((int64_t)s * 49 + pos) * 2304 + tap * 256 + c8 * 8) = v;
    }
}

// per-sample GroupNorm (groups of 256/groups channels over the 49 positions) + ReLU on fp32 [n*49][256] -> bf16 planes
template <int PA>
__global__ __launch_bounds__(256) void k_gn_relu_cl(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    int groups, float eps, uint16_t* __restrict__ out, int n) {
    __shared__ float red[2][256];
    const int s = blockIdx.x, c = threadIdx.x, cpg = 256 / groups;
    float v[49];
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 49; ++p) { v[p] = y[((int64_t)s * 49 + p) * 256 + c]; sum += v[p]; }
    red[0][c] = sum;
    __syncthreads();
    float gs = 0.f;
    for (int j = 0; j < cpg; ++j) gs += red[0][(c / cpg) * cpg + j];
    const float mean = gs / (float)(cpg * 49);
    float sq = 0.f;
#pragma unroll
    for (int p = 0; p < 49; ++p) { const float d = v[p] - mean; sq += d * d; }
    red[1][c] = sq;
    __syncthreads();
    float gq = 0.f;
    for (int j = 0; j < cpg; ++j) gq += red[1][(c / cpg) * cpg + j];
    const float rstd = 1.f / sqrtf(gq / (float)(cpg * 49) + eps);
    const float ga = gamma[c], be = beta[c];
    const int64_t plane = (int64_t)n * 49 * 256;
#pragma unroll
    for (int p = 0; p < 49; ++p) {
        const float o = fmaxf((v[p] - mean) * rstd * ga + be, 0.f);
        uint32_t hi, lo;
        f2bf_split(o, hi, lo);
        const int64_t idx = ((int64_t)s * 49 + p) * 256 + c;
        out[idx] = (uint16_t)hi;
        if (PA == 2) out[idx + plane] = (uint16_t)lo;
    }
}

extern "C" int ph_im2col7(const uint16_t* in, uint16_t* out, int n, int prec, void* stream) {
    PH_CHECK_ARG(in && out && n > 0, "bad pointer or size");
    const int P = prec == PH_PREC_SPLIT ? 2 : 1;
    const int64_t chunks = (int64_t)P * n * 49 * 9 * 32;
    int grid = (int)((chunks + 255) / 256 < 4096 ? (chunks + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_im2col7, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, n, P);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_gn_relu_cl(const float* y, const float* gamma, const float* beta, int groups, float eps, uint16_t* out, int n,
                             int prec, void* stream) {
    PH_CHECK_ARG(y && gamma && beta && out && n > 0 && groups > 0 && 256 % groups == 0, "bad pointer or size");
    if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_gn_relu_cl<1>, dim3(n), dim3(256), 0, (hipStream_t)stream, y, gamma, beta, groups, eps, out, n);
    else hipLaunchKernelGGL(k_gn_relu_cl<2>, dim3(n), dim3(256), 0, (hipStream_t)stream, y, gamma, beta, groups, eps, out, n);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Tracker affinity (polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py:165-182): match scores between the n
// detections of a frame and the m columns of the tracker's memory (tracklets, then backdrops), both as fp32 [.][256]
// embeddings ON THE DEVICE next to the track head that produced them:
//   bisoftmax: (softmax over the columns + softmax over the detections) / 2 of the dot products; softmax: the first term;
//   cosine: dot products of the normalised vectors; `with_cats`: zero where the labels differ.
// n <= 128, m <= 4096: three tiny launches (dot products: one thread per entry; column maxima / sums; row pass), all fp32.
// The greedy assignment that consumes the matrix stays on the host (sequential, data dependent) -- one D2H of n * m floats.
__global__ __launch_bounds__(256) void k_aff_dot(const float* __restrict__ emb, const float* __restrict__ memo, int n, int m,
                                                 int cosine, float* __restrict__ dot) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= n || j >= m) return;
    const float4* a = (const float4*)(emb + (int64_t)i * 256);
    const float4* b = (const float4*)(memo + (int64_t)j * 256);
    float s = 0.f, na = 0.f, nb = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        const float4 x = a[k], y = b[k];
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        if (cosine) {
            na += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            nb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
        }
    }
    if (cosine) s = s / (fmaxf(sqrtf(na), 1e-12f) * fmaxf(sqrtf(nb), 1e-12f));      // F.normalize(eps = 1e-12)
    dot[(int64_t)i * m + j] = s;
}
// per column j: max_i dot[i][j] and sum_i exp(dot[i][j] - max)
__global__ __launch_bounds__(256) void k_aff_colstat(const float* __restrict__ dot, int n, int m, float* __restrict__ cmax,
                                                     float* __restrict__ csum) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = fmaxf(mx, dot[(int64_t)i * m + j]);
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += expf(dot[(int64_t)i * m + j] - mx);
    cmax[j] = mx;
    csum[j] = s;
}
// one workgroup per detection row: row softmax, combined with the column softmax, category mask
__global__ __launch_bounds__(256) void k_aff_rows(const float* __restrict__ dot, const int* __restrict__ lab, const int* __restrict__ memo_lab,
                                                  const float* __restrict__ cmax, const float* __restrict__ csum, int n, int m,
                                                  int metric, int with_cats, float* __restrict__ score) {
    __shared__ float red[256];
    const int i = blockIdx.x, t = threadIdx.x;
    const float* d = dot + (int64_t)i * m;
    float mx = -INFINITY;
    for (int j = t; j < m; j += 256) mx = fmaxf(mx, d[j]);
    red[t] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] = fmaxf(red[t], red[t + s]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sm = 0.f;
    for (int j = t; j < m; j += 256) sm += expf(d[j] - mx);
    red[t] = sm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    sm = red[0];
    const int li = lab[i];
    for (int j = t; j < m; j += 256) {
        float v;
        if (metric == 2) v = d[j];
        else {
            v = expf(d[j] - mx) / sm;
            if (metric == 0) v = 0.5f * (v + expf(d[j] - cmax[j]) / csum[j]);
        }
        if (with_cats && li != memo_lab[j]) v = 0.f;
        score[(int64_t)i * m + j] = v;
    }
}

extern "C" size_t ph_track_affinity_workspace_bytes(int n, int m) { return ((size_t)n * m + 2 * (size_t)m) * sizeof(float); }

extern "C" int ph_track_affinity(const float* emb, const int32_t* labels, const float* memo_emb, const int32_t* memo_labels, int n,
                                 int m, int metric, int with_cats, float* score, void* workspace, size_t workspace_bytes, void* stream) {
    PH_CHECK_ARG(emb && labels && memo_emb && memo_labels && score && workspace, "null pointer");
    PH_CHECK_ARG(n > 0 && n <= 128 && m > 0 && m <= 4096, "n must be in 1..128, m in 1..4096");
    PH_CHECK_ARG(metric >= 0 && metric <= 2, "metric: 0 bisoftmax, 1 softmax, 2 cosine");
    if (workspace_bytes < ph_track_affinity_workspace_bytes(n, m)) { ph_set_error("ph_track_affinity: workspace too small"); return PH_EWORKSPACE; }
    float* dot = (float*)workspace;
    float* cmax = dot + (size_t)n * m;
    float* csum = cmax + m;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_aff_dot, dim3((m + 63) / 64, (n + 3) / 4), dim3(256), 0, s, emb, memo_emb, n, m, metric == 2 ? 1 : 0, dot);
    if (metric == 0) hipLaunchKernelGGL(k_aff_colstat, dim3((m + 255) / 256), dim3(256), 0, s, dot, n, m, cmax, csum);
    hipLaunchKernelGGL(k_aff_rows, dim3(n), dim3(256), 0, s, dot, labels, memo_labels, cmax, csum, n, m, metric, with_cats, score);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
