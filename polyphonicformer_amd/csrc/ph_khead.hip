// A1-A5 -- KernelHead after `localization_fpn` (polyphonic/kernel_head.py:245-347):
//   loc / sem / dfe = ReLU(GroupNorm(conv1x1(f0 / f1 / f2)))   (ConvModule: conv(no bias) -> GN -> ReLU)
//   x = sem + loc
// GroupNorm needs per-(frame, group) statistics of the conv OUTPUT over all pixels, so the conv is
// evaluated twice: a statistics pass that keeps only per-channel sum / sum of
// squares, and an apply pass that normalises and writes bf16 planes (and, on request, the fp32 NCHW
// tensors the reference API hands out).  Recomputing the 256x256xHW GEMM is cheaper than a round
// trip of its output through HBM: both passes are bound by reading the fp32 input map.
//
// GEMM core (shared by both passes): M = 256 output channels, N = 64-pixel tile, K = 256.
// One workgroup = 8 waves; wave w owns output channels 32w..32w+31 and holds its A operand (conv
// weight rows, bf16 planes) in registers.  The fp32 input tile is converted to bf16 plane(s) while it
// is staged into LDS as [256 c][64 px] and read back with ds_read_b64_tr_b16 (same recipe as ph_conv).
#include "ph_common.h"

constexpr int KH_T = 64;
constexpr int KH_LDT = KH_T + 32;
constexpr int KH_THREADS = 512;

struct KHArgs {
    const float* f[2];            // input maps fp32 [B][256][HW]
    const uint16_t* w[2];         // conv weights, bf16 planes [PA][256][256] (out, in)
    int64_t w_plane;
    const float* gamma[2];
    const float* beta[2];
    const float* stats[2];        // [B][groups][2] (mean, rstd)              (apply)
    float* partial[2];            // [B][nwg][256][2] (sum, sumsq)            (stats)
    uint16_t* planes[2];          // outputs bf16 planes [PA][B][256][HWp]   (apply)
    uint16_t* sum_planes;         // planes of map0 + map1 (x_feats) or null
    float* f32[2];                // optional fp32 NCHW outputs
    float* f32_sum;               // optional fp32 x_feats
    int B, groups, tiles_per_wg;
    int64_t HW, HWp;
};

// tile [256 c][64 px] fp32: 16 threads per channel row, 4 px each, 8 rows-of-threads per pass.
// Split in two so that the HBM loads of the NEXT tile are in flight during the MFMA phase of the current one.
__device__ __forceinline__ void kh_load_tile(float (&v)[8][4], const float* __restrict__ src, int64_t HW, int64_t px0, int tid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + q * KH_THREADS;
        const int row = idx >> 4, p4 = (idx & 15) * 4;
        const int64_t px = px0 + p4;
        const float* s = src + (int64_t)row * HW + px;
        if (px + 4 <= HW && ((HW & 3) == 0)) {
            const uint4 t = ld_nt16(s);      // the fp32 map is read once per pass
            v[q][0] = __uint_as_float(t.x); v[q][1] = __uint_as_float(t.y); v[q][2] = __uint_as_float(t.z); v[q][3] = __uint_as_float(t.w);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = (px + e < HW) ? s[e] : 0.f;
        }
    }
}

template <int PA>
__device__ __forceinline__ void kh_store_tile(const float (&v)[8][4], uint16_t* lds, int tid) {
    // fp32 -> bf16 plane(s) in LDS
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + q * KH_THREADS;
        const int row = idx >> 4, p4 = (idx & 15) * 4;
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) f2bf_split(v[q][e], hi[e], lo[e]);
        *(uint2*)(lds + row * KH_LDT + p4) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
        if (PA == 2) *(uint2*)(lds + 256 * KH_LDT + row * KH_LDT + p4) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
    }
}

template <int PA>
__device__ __forceinline__ void kh_load_a(uint4 (&af)[PA][16], const uint16_t* __restrict__ w, int64_t w_plane, int wave,
                                          int lane) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const uint16_t* r = w + p * w_plane + (wave * 32 + (lane & 31)) * 256 + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) af[p][ks] = *(const uint4*)(r + ks * 16);
    }
}

template <int PA>
__device__ __forceinline__ f32x16_t kh_gemm(const uint4 (&af)[PA][16], const uint16_t* lds, int ct, int lane) {
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        uint4 bf[PA];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const uint16_t* a0 = lds + p * 256 * KH_LDT + (ks * 16 + g * 8 + (i16 >> 2)) * KH_LDT + ct * 32 + gi * 16 + (i16 & 3) * 4;
            const uint2 lo = lds_read_tr16(a0);
            const uint2 hi = lds_read_tr16(a0 + 4 * KH_LDT);
            bf[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        acc = mfma32(af[0][ks], bf[0], acc);
        if (PA == 2) {
            acc = mfma32(af[0][ks], bf[PA - 1], acc);
            acc = mfma32(af[PA - 1][ks], bf[0], acc);
        }
    }
    return acc;
}

// ---- pass 1: per-channel sum / sum of squares of the conv output -------------------------------
template <int PA>
__global__ __launch_bounds__(KH_THREADS) void k_khead_stats(const KHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int b = blockIdx.y;
    const float* src = a.f[0] + (int64_t)b * 256 * a.HW;
    uint4 af[PA][16];
    kh_load_a<PA>(af, a.w[0], a.w_plane, wave, lane);
    const int ntiles = (int)(a.HWp / KH_T);
    const int t0 = blockIdx.x * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    float stg[8][4];
    if (t0 < t1) kh_load_tile(stg, src, a.HW, (int64_t)t0 * KH_T, tid);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();
        kh_store_tile<PA>(stg, lds, tid);
        __syncthreads();
        if (t + 1 < t1) kh_load_tile(stg, src, a.HW, (int64_t)(t + 1) * KH_T, tid);   // in flight during the MFMAs
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x16_t acc = kh_gemm<PA>(af, lds, ct, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1[r] += acc[r]; s2[r] += acc[r] * acc[r]; }
        }
    }
    // reduce over the 32 pixel lanes of each half-wave, lanes 0 / 32 hold the channel totals
    float* out = a.partial[0] + ((int64_t)b * gridDim.x + blockIdx.x) * 256 * 2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x = s1[r], y = s2[r];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) { x += __shfl_xor(x, m); y += __shfl_xor(y, m); }
        if ((lane & 31) == 0) {
            const int ch = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            out[ch * 2] = x;
            out[ch * 2 + 1] = y;
        }
    }
}

// partial [B][nwg][256][2] -> stats [B][groups][2] = (mean, rstd); one block of 1024 threads per frame: 4 threads
// per channel take interleaved workgroups with 4 independent accumulator pairs each (the loads of a chain of nwg
// dependent additions were 80 us at nwg = 256), fp64 combine in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void k_gn_finalize(const float* __restrict__ partial, float* __restrict__ stats, int nwg,
                                                      int groups, int64_t HW, float eps) {
    __shared__ double sh[4][256][2];
    const int b = blockIdx.x, c = threadIdx.x & 255, q = threadIdx.x >> 8;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, t[4] = {0.0, 0.0, 0.0, 0.0};
    const float2* p = (const float2*)partial + ((int64_t)b * nwg) * 256 + c;
    int w = q;
    for (; w + 12 < nwg; w += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 v = p[(int64_t)(w + 4 * j) * 256];
            s[j] += (double)v.x;
            t[j] += (double)v.y;
        }
    }
    for (; w < nwg; w += 4) {
        const float2 v = p[(int64_t)w * 256];
        s[0] += (double)v.x;
        t[0] += (double)v.y;
    }
    sh[q][c][0] = (s[0] + s[1]) + (s[2] + s[3]);
    sh[q][c][1] = (t[0] + t[1]) + (t[2] + t[3]);
    __syncthreads();
    const int cpg = 256 / groups;
    if (threadIdx.x < groups) {
        const int gidx = threadIdx.x;
        double ss = 0.0, qq = 0.0;
        for (int j = 0; j < cpg; ++j)
            for (int k = 0; k < 4; ++k) { ss += sh[k][gidx * cpg + j][0]; qq += sh[k][gidx * cpg + j][1]; }
        const double n = (double)cpg * (double)HW;
        const double mean = ss / n;
        double var = qq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((int64_t)b * groups + gidx) * 2] = (float)mean;
        stats[((int64_t)b * groups + gidx) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

extern "C" int ph_gn_finalize(const float* partial, float* stats, int nwg, int groups, int64_t HW, float eps, int B,
                              void* stream) {
    PH_CHECK_ARG(partial && stats && nwg > 0 && groups > 0 && 256 % groups == 0 && HW > 0 && B > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_gn_finalize, dim3(B), dim3(1024), 0, (hipStream_t)stream, partial, stats, nwg, groups, HW, eps);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---- pass 2: normalise + ReLU, write planes (and fp32), optional sum of two maps ----------------
constexpr int KH_PLD = KH_T + 8;    // row stride (elements) of a per-wave [32 ch][64 px] store patch

// bf16 patch [32 ch][64 px] of this wave -> planes: 8 lanes x 16 B per channel row = whole 128-byte lines
__device__ __forceinline__ void kh_flush_patch(const uint16_t* patch, uint16_t* dst_plane, int64_t row0_off, int64_t HWp,
                                               int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rl = it * 8 + (lane >> 3), piece = lane & 7;
        const uint4 v = *(const uint4*)(patch + rl * KH_PLD + piece * 8);
        st_nt16(dst_plane + row0_off + (int64_t)rl * HWp + piece * 8, v);
    }
}

template <int PA, int NMAP>
__global__ __launch_bounds__(KH_THREADS) void k_khead_apply(const KHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* patches = lds + PA * 256 * KH_LDT;                     // [8 waves][32][KH_PLD] (one plane at a time)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    uint16_t* patch = patches + wave * (32 * KH_PLD);
    const int b = blockIdx.y;
    const int cpg = 256 / a.groups;
    const int64_t oplane = (int64_t)a.B * 256 * a.HWp;
    // per-channel affine of the normalisation (rstd*gamma, beta - mean*rstd*gamma) in LDS: 64 registers saved
    float2* ss = (float2*)(patches + 8 * 32 * KH_PLD);               // [NMAP][256]
    for (int i = tid; i < NMAP * 256; i += KH_THREADS) {
        const int m = i >> 8, ch = i & 255;
        const float* st = a.stats[m] + ((int64_t)b * a.groups + ch / cpg) * 2;
        const float mean = st[0], rstd = st[1];
        ss[i] = make_float2(rstd * a.gamma[m][ch], a.beta[m][ch] - mean * rstd * a.gamma[m][ch]);
    }
    __syncthreads();
    constexpr bool HOIST = (PA == 1 && NMAP == 1);   // a single map: its weights stay in registers across tiles
    uint4 af[HOIST ? NMAP : 1][PA][16];
    if (HOIST) {
#pragma unroll
        for (int m = 0; m < NMAP; ++m) kh_load_a<PA>(af[m], a.w[m], a.w_plane, wave, lane);
    }
    const int ntiles = (int)(a.HWp / KH_T);
    const int t0 = blockIdx.x * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    float stg[8][4];
    if (t0 < t1) kh_load_tile(stg, a.f[0] + (int64_t)b * 256 * a.HW, a.HW, (int64_t)t0 * KH_T, tid);
    for (int t = t0; t < t1; ++t) {
        const int64_t px0 = (int64_t)t * KH_T;
        const int64_t row0 = ((int64_t)b * 256 + wave * 32) * a.HWp + px0;       // this wave's first channel row, tile start
        float keep[2][16];
#pragma unroll
        for (int m = 0; m < NMAP; ++m) {
            __syncthreads();
            kh_store_tile<PA>(stg, lds, tid);
            __syncthreads();
            // next (tile, map) in flight during this one's MFMAs and stores
            if (m + 1 < NMAP) kh_load_tile(stg, a.f[m + 1] + (int64_t)b * 256 * a.HW, a.HW, px0, tid);
            else if (t + 1 < t1) kh_load_tile(stg, a.f[0] + (int64_t)b * 256 * a.HW, a.HW, px0 + KH_T, tid);
            if (!HOIST) kh_load_a<PA>(af[0], a.w[m], a.w_plane, wave, lane);
            float vals[2][16];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f32x16_t acc = kh_gemm<PA>(af[HOIST ? m : 0], lds, ct, lane);
                const int64_t px = px0 + ct * 32 + (lane & 31);
                const bool inside = px < a.HW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float2 af2 = ss[m * 256 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];
                    float v = fmaxf(acc[r] * af2.x + af2.y, 0.f);
                    if (!inside) v = 0.f;                                    // planes are zero padded
                    vals[ct][r] = v;
                    if (a.f32[m] && inside) {
                        const int ch = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        a.f32[m][((int64_t)b * 256 + ch) * a.HW + px] = v;
                    }
                }
            }
            // map m through the store patch, one bf16 plane at a time
#pragma unroll
            for (int p = 0; p < PA; ++p) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
                        uint32_t hi, lo;
                        f2bf_split(vals[ct][r], hi, lo);
                        patch[rl * KH_PLD + ct * 32 + (lane & 31)] = (uint16_t)(p == 0 ? hi : lo);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                kh_flush_patch(patch, a.planes[m] + p * oplane, row0, a.HWp, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            if (NMAP == 2) {
                if (m == 0) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r) keep[ct][r] = vals[ct][r];
                } else {
                    // x_feats = semantic_feats + loc_feats   (kernel_head.py:303)
#pragma unroll
                    for (int p = 0; p < PA; ++p) {
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            const int64_t px = px0 + ct * 32 + (lane & 31);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
                                const float sum = keep[ct][r] + vals[ct][r];
                                uint32_t hi, lo;
                                f2bf_split(sum, hi, lo);
                                patch[rl * KH_PLD + ct * 32 + (lane & 31)] = (uint16_t)(p == 0 ? hi : lo);
                                if (p == 0 && a.f32_sum && px < a.HW)
                                    a.f32_sum[((int64_t)b * 256 + wave * 32 + rl) * a.HW + px] = sum;
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        kh_flush_patch(patch, a.sum_planes + p * oplane, row0, a.HWp, lane);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
    }
}

// ---- proposal kernels: k0 = init_kernels.weight + pooled object features (kernel_head.py:299-300,324-326),
//      stuff kernels = conv_seg.weight[num_thing:num_classes] (:332-335) --------------------------------
__global__ __launch_bounds__(256) void k_khead_proposals(const float* __restrict__ partial, int nsplit, int Npad_th,
                                                         const float* __restrict__ w_init, const float* __restrict__ w_stuff,
                                                         float* __restrict__ out, int n_th, int n_stuff) {
    const int b = blockIdx.y, n = blockIdx.x, c = threadIdx.x, N = n_th + n_stuff;
    float v;
    if (n < n_th) {
        v = w_init[n * 256 + c];
        for (int s = 0; s < nsplit; ++s) v += partial[(((int64_t)b * nsplit + s) * Npad_th + n) * 512 + c];
    } else {
        v = w_stuff[(n - n_th) * 256 + c];
    }
    out[((int64_t)b * N + n) * 256 + c] = v;
}

// ================================================================================================
static int kh_tiles_per_wg(int64_t HWp, int B) {
    const int ntiles = (int)(HWp / KH_T);
    int tpw = (int)(((int64_t)ntiles * B + 511) / 512);
    if (tpw < 1) tpw = 1;
    if (tpw > 32) tpw = 32;
    return tpw;
}

extern "C" size_t ph_khead_workspace_bytes(int B, int64_t HW, int groups) {
    const int64_t HWp = ph_hw_padded(HW);
    const int tpw = kh_tiles_per_wg(HWp, B);
    const int nwg = (int)((HWp / KH_T + tpw - 1) / tpw);
    return (size_t)3 * B * nwg * 256 * 2 * sizeof(float) + (size_t)3 * B * groups * 2 * sizeof(float);
}

extern "C" int ph_khead_conv_gn(const float* f0, const float* f1, const float* f2, const uint16_t* wplanes,
                                const float* gn_affine, int groups, float eps, uint16_t* loc_planes,
                                uint16_t* sem_planes, uint16_t* x_planes, uint16_t* dfe_planes, float* x_f32,
                                float* dfe_f32, void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec,
                                void* stream) {
    PH_CHECK_ARG(f0 && f1 && f2 && wplanes && gn_affine && loc_planes && sem_planes && x_planes && dfe_planes && workspace,
                 "null pointer");
    PH_CHECK_ARG(B > 0 && HW > 0 && groups > 0 && 256 % groups == 0, "bad size");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    if (workspace_bytes < ph_khead_workspace_bytes(B, HW, groups)) {
        ph_set_error("ph_khead_conv_gn: workspace too small");
        return PH_EWORKSPACE;
    }
    const int PA = prec == PH_PREC_SPLIT ? 2 : 1;
    const int64_t HWp = ph_hw_padded(HW);
    const int tpw = kh_tiles_per_wg(HWp, B);
    const int nwg = (int)((HWp / KH_T + tpw - 1) / tpw);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* stats = partial + (size_t)3 * B * nwg * 256 * 2;
    const float* fm[3] = {f0, f1, f2};
    const size_t lds = (size_t)PA * 256 * KH_LDT * sizeof(uint16_t);
    const size_t lds_apply = lds + (size_t)8 * 32 * KH_PLD * sizeof(uint16_t) + 2 * 256 * sizeof(float2);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)k_khead_stats<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_khead_stats<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_khead_apply<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_khead_apply<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_khead_apply<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_khead_apply<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    KHArgs a;
    a.B = B; a.groups = groups; a.tiles_per_wg = tpw; a.HW = HW; a.HWp = HWp;
    a.w_plane = (int64_t)3 * 256 * 256;
    const dim3 grid(nwg, B), block(KH_THREADS);
    for (int m = 0; m < 3; ++m) {   // pass 1 (+ finalize) per map
        a.f[0] = fm[m];
        a.w[0] = wplanes + (size_t)m * 256 * 256;
        a.partial[0] = partial + (size_t)m * B * nwg * 256 * 2;
        if (PA == 1) hipLaunchKernelGGL(k_khead_stats<1>, grid, block, lds, s, a);
        else hipLaunchKernelGGL(k_khead_stats<2>, grid, block, lds, s, a);
        hipLaunchKernelGGL(k_gn_finalize, dim3(B), dim3(1024), 0, s, a.partial[0], stats + (size_t)m * B * groups * 2, nwg,
                           groups, HW, eps);
    }
    // pass 2: (loc, sem) -> loc, sem, x ; then depth
    for (int m = 0; m < 2; ++m) {
        a.f[m] = fm[m];
        a.w[m] = wplanes + (size_t)m * 256 * 256;
        a.gamma[m] = gn_affine + (size_t)m * 512;
        a.beta[m] = gn_affine + (size_t)m * 512 + 256;
        a.stats[m] = stats + (size_t)m * B * groups * 2;
        a.f32[m] = nullptr;
    }
    a.planes[0] = loc_planes; a.planes[1] = sem_planes; a.sum_planes = x_planes; a.f32_sum = x_f32;
    if (PA == 1) hipLaunchKernelGGL((k_khead_apply<1, 2>), grid, block, lds_apply, s, a);
    else hipLaunchKernelGGL((k_khead_apply<2, 2>), grid, block, lds_apply, s, a);
    a.f[0] = f2; a.w[0] = wplanes + (size_t)2 * 256 * 256;
    a.gamma[0] = gn_affine + 2 * 512; a.beta[0] = gn_affine + 2 * 512 + 256;
    a.stats[0] = stats + (size_t)2 * B * groups * 2;
    a.planes[0] = dfe_planes; a.f32[0] = dfe_f32; a.sum_planes = nullptr; a.f32_sum = nullptr;
    if (PA == 1) hipLaunchKernelGGL((k_khead_apply<1, 1>), grid, block, lds_apply, s, a);
    else hipLaunchKernelGGL((k_khead_apply<2, 1>), grid, block, lds_apply, s, a);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_khead_proposals(const float* partial, int nsplit, const float* w_init, const float* w_stuff,
                                  float* proposal_feats, int B, int n_thing_queries, int n_stuff, void* stream) {
    PH_CHECK_ARG(partial && w_init && proposal_feats && (w_stuff || n_stuff == 0), "null pointer");
    PH_CHECK_ARG(B > 0 && n_thing_queries > 0 && n_stuff >= 0 && nsplit >= 1, "bad size");
    hipLaunchKernelGGL(k_khead_proposals, dim3(n_thing_queries + n_stuff, B), dim3(256), 0, (hipStream_t)stream, partial,
                       nsplit, ph_n_padded(n_thing_queries), w_init, w_stuff, proposal_feats, n_thing_queries, n_stuff);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
