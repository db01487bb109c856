// A1-A5 -- KernelHead after `localization_fpn` (polyphonic/kernel_head.py:245-347):
//   loc / sem / dfe = ReLU(GroupNorm(conv1x1(f0 / f1 / f2)))   (ConvModule: conv(no bias) -> GN -> ReLU)
//   x = sem + loc ; mask logits = init_kernels(loc) ; seg_preds = conv_seg(sem) ; depth_pred = conv_direct_depth(dfe)
// GroupNorm needs per-(frame, group) statistics of the conv OUTPUT over all pixels, so the conv is
// evaluated twice: a statistics pass that keeps only per-channel sum / sum of
// squares, and an apply pass that normalises, writes bf16 planes (and, on request, the fp32 NCHW
// tensors the reference API hands out) and -- fused entry point -- multiplies the normalised tile, still in
// LDS, with the static 1x1 kernels that consume it, so loc / sem never exist in HBM.
// Both passes run at the speed of their HBM streams: with the GEMM compiled out they take the same time
// (stats 5.0 TB/s of fp32 reads; apply ~4 TB/s of mixed fp32 reads / 128-byte bf16 row writes), and two tiles
// in flight or a double-buffered LDS tile change nothing (measured) -- bytes are what is left to remove.
//
// GEMM core (shared by both passes): M = 256 output channels, N = 64-pixel tile, K = 256.
// One workgroup = 8 waves; wave w owns output channels 32w..32w+31 and holds its A operand (conv
// weight rows, bf16 planes) in registers.  The fp32 input tile is converted to bf16 plane(s) while it
// is staged into LDS as [256 c][64 px] and read back with ds_read_b64_tr_b16 (same recipe as ph_conv).
#include "ph_common.h"
#include <stdlib.h>

constexpr int KH_T = 64;
constexpr int KH_LDT = KH_T + 32;       // row stride 48 dwords: the transposing reads are bank-conflict free
constexpr int KH_THREADS = 512;

struct KHArgs {
    const float* f[3];            // input maps fp32 [B][256][HW]       (stats: all three, map = blockIdx.z; apply: f[0])
    const uint16_t* fp[3];        // INFMT 2: the same maps as bf16 planes [PA][B][256][HWp], zero in [HW, HWp)
    const uint16_t* w[3];         // conv weights, bf16 planes [PA][256][256] (out, in)
    int64_t w_plane;
    float* partial[3];            // [B][nwg][256][2] (sum, sumsq)            (stats)
    const float* gamma;           // apply: one map per launch
    const float* beta;
    const float* stats;           // [B][groups][2] (mean, rstd)
    uint16_t* planes;             // output bf16 planes [PA][B][256][HWp] of this map, or null
    const uint16_t* add_planes;   // ADD: planes of the map added to this one (loc_feats, written by the previous launch)
    uint16_t* sum_planes;         // ADD: planes of the sum (x_feats); may alias add_planes
    float* f32;                   // optional fp32 NCHW output of this map
    float* f32_sum;               // optional fp32 x_feats
    // fused static 1x1 conv on the normalised tile (null w2: none)
    const uint16_t* w2;           // MFMA 32x32x16 A fragments [PA][m2_tiles][16][64 lanes][8]
    int64_t w2_plane;
    const float* bias2;           // [m2_tiles * 32] or null
    void* out2;                   // fp32 (or fp16: out2_f16) [B][out2_rows][HW], rows [0, n2) written
    void* out2b;                  // optional second destination of rows [dual_lo, dual_lo + dual_n): row0b + (row - dual_lo)
    int out2_f16;
    const unsigned* run_if;       // optional predicate: the launch returns at once when *run_if == 0 (ph_khead_fused_if)
    int m2_tiles, n2, out2_rows, dual_lo, dual_n, out2b_rows, out2b_row0;
    uint16_t* blocks_out;         // optional: this map as per-(frame, wave, tile) register-layout blocks (see kh_block)
    const uint16_t* blocks_in;    // ADD == 2: the map to add, in that form
    int w2_lds;                   // 1: the fragments are copied to LDS once per workgroup (bf16 precision, <= 6 row tiles)
    // plain3 (ph_neck_out_convs): the three maps of ONE input in one apply launch, 1-D XCD-aware grid -- the three workgroups
    // that read the same tiles sit on the same XCD in consecutive slots, so the input crosses HBM once and L2 three times
    int plain3, nwg;
    uint16_t* planes3[3];
    float* f32o3[3];
    const float* gn3;             // [3][2][256]
    const float* stats3;          // [3][B][groups][2]
    int B, groups, tiles_per_wg;
    int64_t HW, HWp;
};

// tile [256 c][64 px] fp32: 16 threads per channel row, 4 px each, 32 channel rows per pass (8 passes).
// Every load is unconditional (column clamped into the map, zeroed when it is staged into LDS) and addressed as
// a uniform base (map + 32 q rows) plus ONE 32-bit lane offset: no branches, no per-load address registers.
// `more` = false (nothing left to prefetch): the same loads, all on the first bytes of the map (one cached line).
// AL4: HW % 4 == 0 (16-byte loads); otherwise four 4-byte loads per thread and pass.
template <bool AL4>
__device__ __forceinline__ void kh_load_tile(float (&v)[8][4], const float* __restrict__ src, int64_t HW, int64_t px0, int tid,
                                             bool more) {
    const int row_in = tid >> 4, p4 = (tid & 15) * 4;
    const int64_t qstride = more ? 32 * HW : 0;
    if (AL4) {
        int64_t col = px0 + p4;
        if (col > HW - 4) col = HW - 4;
        const uint32_t voff = more ? (uint32_t)(((int64_t)row_in * HW + col) * 4) : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const char* ub = (const char*)(src + (int64_t)q * qstride);
            const uint4 t = ld_nt16(ub + voff);      // the fp32 map is read once per pass
            v[q][0] = __uint_as_float(t.x); v[q][1] = __uint_as_float(t.y); v[q][2] = __uint_as_float(t.z); v[q][3] = __uint_as_float(t.w);
        }
    } else {
        uint32_t voff[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int64_t col = px0 + p4 + e;
            if (col > HW - 1) col = HW - 1;
            voff[e] = more ? (uint32_t)(((int64_t)row_in * HW + col) * 4) : 0u;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const char* ub = (const char*)(src + (int64_t)q * qstride);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = __builtin_nontemporal_load((const float*)(ub + voff[e]));
        }
    }
}

// element format of the tile / weights: bf16 (hi [+ lo] planes) or ONE IEEE fp16 plane (PH_PREC_F16: 2^-12 per operand,
// one MFMA per product -- the grade of the decode's `fp16` mode)
template <int E> __device__ __forceinline__ void kh_split(float x, uint32_t& hi, uint32_t& lo) {
    if constexpr (E == PH_E_F16) { hi = f2h(x); lo = 0; }
    else f2bf_split(x, hi, lo);
}

template <int PA, int E>
__device__ __forceinline__ void kh_store_tile(const float (&v)[8][4], uint16_t* lds, int tid, int64_t HW, int64_t px0) {
    // fp32 -> bf16 plane(s) (or ONE fp16 plane, E = PH_E_F16) in LDS; columns past the map are zero
    const int row_in = tid >> 4, p4 = (tid & 15) * 4;
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ok[e] = px0 + p4 + e < HW;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = q * 32 + row_in;
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) kh_split<E>(ok[e] ? v[q][e] : 0.f, hi[e], lo[e]);
        *(uint2*)(lds + row * KH_LDT + p4) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
        if (PA == 2) *(uint2*)(lds + 256 * KH_LDT + row * KH_LDT + p4) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
    }
}

// INFMT 2 -- the maps arrive as bf16 channel planes (the neck's hand-off, ph_gn_apply PH_GN_TO_CPLANES): a tile is 256 rows
// x 128 bytes per plane, 4 x 16 bytes per thread, copied to LDS as it is (no conversion, half the bytes)
template <int PA>
__device__ __forceinline__ void kh_load_tile_planes(uint4 (&q)[PA][4], const uint16_t* __restrict__ planes, int64_t oplane,
                                                    int64_t HWp, int64_t px0, int tid, bool more) {
    const uint32_t voff = more ? (uint32_t)((((int64_t)(tid >> 3)) * HWp + px0 + (tid & 7) * 8) * 2) : 0u;
    const int64_t qstride = more ? 64 * HWp : 0;
#pragma unroll
    for (int p = 0; p < PA; ++p)
#pragma unroll
        for (int it = 0; it < 4; ++it)
            q[p][it] = ld_nt16((const char*)(planes + p * oplane + it * qstride) + voff);
}
template <int PA>
__device__ __forceinline__ void kh_store_tile_planes(const uint4 (&q)[PA][4], uint16_t* lds, int tid) {
#pragma unroll
    for (int p = 0; p < PA; ++p)
#pragma unroll
        for (int it = 0; it < 4; ++it)
            *(uint4*)(lds + p * 256 * KH_LDT + (it * 64 + (tid >> 3)) * KH_LDT + (tid & 7) * 8) = q[p][it];
}
// INFMT 3 -- ONE input in channels-last planes [PA][B][HW][256] (the neck's level sum, ph_gn_sum_planes): a tile of 64 pixels is
// 32 KiB of consecutive bytes per plane (4 x 16 bytes per thread, fully coalesced), kept in LDS as [64 px][256 + 8] so that the
// B fragment of a k-step -- 8 consecutive channels of one pixel per lane -- is ONE ds_read_b128 (16 lanes of a group on 16
// distinct 16-byte slots: pixel pitch 132 dwords), no transposing read.  Pixels past the map are zero.
constexpr int KH_NLDP = 256 + 8;
template <int PA>
__device__ __forceinline__ void kh_load_tile_nhwc(uint4 (&q)[PA][4], const uint16_t* __restrict__ planes, int64_t iplane,
                                                  int64_t HW, int64_t px0, int tid, bool more) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * KH_THREADS;
        const int64_t px = px0 + (idx >> 5);
        const bool ok = more && px < HW;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            q[p][it] = make_uint4(0, 0, 0, 0);
            if (ok) q[p][it] = ld_nt16(planes + p * iplane + px * 256 + (idx & 31) * 8);
        }
    }
}
template <int PA>
__device__ __forceinline__ void kh_store_tile_nhwc(const uint4 (&q)[PA][4], uint16_t* lds, int tid) {
#pragma unroll
    for (int p = 0; p < PA; ++p)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * KH_THREADS;
            *(uint4*)(lds + p * 256 * KH_LDT + (idx >> 5) * KH_NLDP + (idx & 31) * 8) = q[p][it];
        }
}
// staging registers of one tile in either input format
template <int PA, int INFMT, int E> struct KhStage {
    float v[8][4];
    __device__ __forceinline__ void load(const KHArgs& a, int m, int b, int64_t px0, int tid, bool more) {
        kh_load_tile<INFMT == 1>(v, a.f[m] + (int64_t)b * 256 * a.HW, a.HW, px0, tid, more);
    }
    __device__ __forceinline__ void store(uint16_t* lds, int tid, int64_t HW, int64_t px0) const { kh_store_tile<PA, E>(v, lds, tid, HW, px0); }
};
template <int PA, int E> struct KhStage<PA, 2, E> {
    uint4 q[PA][4];
    __device__ __forceinline__ void load(const KHArgs& a, int m, int b, int64_t px0, int tid, bool more) {
        kh_load_tile_planes<PA>(q, a.fp[m] + (int64_t)b * 256 * a.HWp, (int64_t)a.B * 256 * a.HWp, a.HWp, px0, tid, more);
    }
    __device__ __forceinline__ void store(uint16_t* lds, int tid, int64_t, int64_t) const { kh_store_tile_planes<PA>(q, lds, tid); }
};

template <int PA, int E> struct KhStage<PA, 3, E> {
    uint4 q[PA][4];
    __device__ __forceinline__ void load(const KHArgs& a, int m, int b, int64_t px0, int tid, bool more) {
        kh_load_tile_nhwc<PA>(q, a.fp[m] + (int64_t)b * a.HW * 256, (int64_t)a.B * a.HW * 256, a.HW, px0, tid, more);
    }
    __device__ __forceinline__ void store(uint16_t* lds, int tid, int64_t, int64_t) const { kh_store_tile_nhwc<PA>(q, lds, tid); }
};

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

// acc[32 rows of A][32 px of column tile ct] over K = 256 channels of the LDS tile
// NH: the tile is the channels-last image [64 px][KH_NLDP] (INFMT 3) instead of [256 c][KH_LDT]
template <int PA, int E, bool NH = false>
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
            if constexpr (NH) {
                bf[p] = *(const uint4*)(lds + p * 256 * KH_LDT + (ct * 32 + (lane & 31)) * KH_NLDP + ks * 16 + g * 8);
                continue;
            }
            const uint16_t* a0 = lds + p * 256 * KH_LDT + (ks * 16 + g * 8 + (i16 >> 2)) * KH_LDT + ct * 32 + gi * 16 + (i16 & 3) * 4;
            const uint2 lo = lds_read_tr16(a0);
            const uint2 hi = lds_read_tr16(a0 + 4 * KH_LDT);
            bf[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        acc = mfma32e<E>(af[0][ks], bf[0], acc);
        if (PA == 2) {
            acc = mfma32e<E>(af[0][ks], bf[PA - 1], acc);
            acc = mfma32e<E>(af[PA - 1][ks], bf[0], acc);
        }
        if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // bounds how many B fragments are read ahead (registers)
    }
    return acc;
}

// ---- pass 1: per-channel sum / sum of squares of the conv output; the three maps in one launch (blockIdx.z) ------
template <int PA, int INFMT, int E = PH_E_BF16>
__global__ __launch_bounds__(KH_THREADS) void k_khead_stats(const KHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5;
    int b = blockIdx.y, m = blockIdx.z, bx = blockIdx.x, nbx = gridDim.x;
    if (a.plain3) {
        // the three maps of ONE input (ph_neck_out_convs; round 5, like the apply pass since round 4): 1-D XCD-aware grid -- the three
        // workgroups that read the same tiles sit on the same XCD in consecutive slots, so the input crosses HBM once, not three times
        // (PMC: 762 MB of reads per 16-frame launch for a 268 MB plane)
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        const int grp = (slot / 3) * 8 + xcd;
        if (grp >= a.nwg * a.B) return;
        m = slot - (slot / 3) * 3;
        b = grp / a.nwg;
        bx = grp - b * a.nwg;
        nbx = a.nwg;
    }
    if (a.run_if && *a.run_if == 0) return;
    uint4 af[PA][16];
    kh_load_a<PA>(af, a.w[m], a.w_plane, wave, lane);
    const int ntiles = (int)(a.HWp / KH_T);
    const int t0 = bx * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    KhStage<PA, INFMT, E> stg;
    stg.load(a, m, b, (int64_t)t0 * KH_T, tid, t0 < t1);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();
        stg.store(lds, tid, a.HW, (int64_t)t * KH_T);
        __syncthreads();
        stg.load(a, m, b, (int64_t)(t + 1) * KH_T, tid, t + 1 < t1);   // in flight during the MFMAs
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x16_t acc = kh_gemm<PA, E, INFMT == 3>(af, lds, ct, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1[r] += acc[r]; s2[r] += acc[r] * acc[r]; }
        }
    }
    // reduce over the 32 pixel lanes of each half-wave, lanes 0 / 32 hold the channel totals
    float* out = a.partial[m] + ((int64_t)b * nbx + bx) * 256 * 2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x = s1[r], y = s2[r];
#pragma unroll
        for (int q = 1; q < 32; q <<= 1) { x += __shfl_xor(x, q); y += __shfl_xor(y, q); }
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
                                                      int groups, int64_t HW, float eps, const unsigned* run_if) {
    __shared__ double sh[4][256][2];
    if (run_if && *run_if == 0) return;
    const int b = blockIdx.x, c = threadIdx.x & 255, q = threadIdx.x >> 8;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, t[4] = {0.0, 0.0, 0.0, 0.0};
    const float2* p = (const float2*)partial + ((int64_t)b * nwg) * 256 + c;
    int w = q;
    // round 6: sixteen loads in flight per thread instead of four (the same additions in the same order: the sums are bit-identical) -- the
    // kernel is a chain of global-load latencies: 26 -> 9 us per launch at 256 partials per frame, nine launches per neck run
    for (; w + 60 < nwg; w += 64) {
        float2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[(int64_t)(w + 4 * j) * 256];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s[j & 3] += (double)v[j].x;
            t[j & 3] += (double)v[j].y;
        }
    }
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
    hipLaunchKernelGGL(k_gn_finalize, dim3(B), dim3(1024), 0, (hipStream_t)stream, partial, stats, nwg, groups, HW, eps,
                       (const unsigned*)nullptr);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---- pass 2: normalise + ReLU; planes (+ fp32) out; ADD: x_feats = this map + the previous launch's map;
//      w2: static 1x1 conv of the normalised tile (second GEMM, tile still in LDS) -----------------------------
__device__ __forceinline__ void kh_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// this wave's 32 channel rows of the LDS tile <-> planes: 8 lanes x 16 B per channel row = whole 128-byte lines
__device__ __forceinline__ void kh_flush_rows(const uint16_t* rows, uint16_t* dst_plane, int64_t row0_off, int64_t HWp, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rl = it * 8 + (lane >> 3), piece = lane & 7;
        const uint4 v = *(const uint4*)(rows + rl * KH_LDT + piece * 8);
        st_nt16(dst_plane + row0_off + (int64_t)rl * HWp + piece * 8, v);
    }
}
// vals (fp32, D layout of the two 32x32 tiles) -> bf16 plane(s) in this wave's rows of the LDS tile
template <int PA, int E>
__device__ __forceinline__ void kh_vals_to_rows(const float (&vals)[2][16], uint16_t* rows, int lane) {
    const int g = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * g;
            uint32_t hi, lo;
            kh_split<E>(vals[ct][r], hi, lo);
            rows[rl * KH_LDT + ct * 32 + (lane & 31)] = (uint16_t)hi;
            if (PA == 2) rows[256 * KH_LDT + rl * KH_LDT + ct * 32 + (lane & 31)] = (uint16_t)lo;
        }
}

template <int PA, int ADD, int INFMT, int E = PH_E_BF16>
__global__ __launch_bounds__(KH_THREADS) void k_khead_apply(const KHArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5;
    uint16_t* rows = lds + wave * 32 * KH_LDT;                       // this wave's channel rows (plane p: + p * 256 * KH_LDT)
    int b = blockIdx.y, bx = blockIdx.x, m = 0;
    const float *gamma = a.gamma, *beta = a.beta, *stats = a.stats;
    uint16_t* oplanes = a.planes;
    float* of32 = a.f32;
    if (ADD == 0 && a.plain3) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        const int grp = (slot / 3) * 8 + xcd;                        // (frame, tile range) groups dealt round-robin to the XCDs
        if (grp >= a.nwg * a.B) return;
        m = slot - (slot / 3) * 3;
        b = grp / a.nwg;
        bx = grp - b * a.nwg;
        gamma = a.gn3 + m * 512; beta = gamma + 256;
        stats = a.stats3 + (int64_t)m * a.B * a.groups * 2;
        oplanes = a.planes3[m]; of32 = a.f32o3[m];
    }
    if (a.run_if && *a.run_if == 0) return;
    const int cpg = 256 / a.groups;
    const int64_t oplane = (int64_t)a.B * 256 * a.HWp;
    // per-channel affine of the normalisation (rstd*gamma, beta - mean*rstd*gamma) in LDS
    float2* ss = (float2*)(lds + PA * 256 * KH_LDT);                 // [256]
    for (int ch = tid; ch < 256; ch += KH_THREADS) {
        const float* st = stats + ((int64_t)b * a.groups + ch / cpg) * 2;
        const float mean = st[0], rstd = st[1];
        ss[ch] = make_float2(rstd * gamma[ch], beta[ch] - mean * rstd * gamma[ch]);
    }
    // bf16 precision: the A fragments of the second GEMM live in LDS behind the tile (<= 96 KiB); split precision
    // (two planes of everything) and more than 192 rows read them from L2
    float* b2l = (float*)(ss + 256);                                 // [<= 256] bias of the second GEMM
    if (a.w2)
        for (int i = tid; i < a.m2_tiles * 32; i += KH_THREADS) b2l[i] = a.bias2 ? a.bias2[i] : 0.f;
    uint16_t* w2l = lds + PA * 256 * KH_LDT + 256 * 4 + 256 * 2;     // after ss ([256] float2) and b2l ([256] float)
    if (PA == 1 && a.w2 && a.w2_lds) {
        const int n16 = a.m2_tiles * 16 * 64;                        // 16-byte pieces
        for (int i = tid; i < n16; i += KH_THREADS) *(uint4*)(w2l + i * 8) = *(const uint4*)(a.w2 + (int64_t)i * 8);
    }
    constexpr bool HOIST = (PA == 1);   // bf16 precision: the 64 weight registers stay resident across the tiles
    uint4 af[PA][16];
    if (HOIST) kh_load_a<PA>(af, a.w[m], a.w_plane, wave, lane);
    const int ntiles = (int)(a.HWp / KH_T);
    const int t0 = bx * a.tiles_per_wg;
    const int t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    KhStage<PA, INFMT, E> stg;
    stg.load(a, m, b, (int64_t)t0 * KH_T, tid, t0 < t1);
    for (int t = t0; t < t1; ++t) {
        const int64_t px0 = (int64_t)t * KH_T;
        const int64_t row0 = ((int64_t)b * 256 + wave * 32) * a.HWp + px0;       // this wave's first channel row, tile start
        __syncthreads();                      // every wave is done with the previous tile in LDS (second GEMM / flush)
        stg.store(lds, tid, a.HW, px0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        stg.load(a, m, b, px0 + KH_T, tid, t + 1 < t1);                            // next tile in flight during this one
        __builtin_amdgcn_sched_barrier(0);
        uint4 addv[PA][4];
        uint32_t addb[PA][2][8];
        if (ADD == 1) {                                                            // the other map's bf16 rows of this wave
#pragma unroll
            for (int p = 0; p < PA; ++p)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    addv[p][it] = *(const uint4*)(a.add_planes + p * oplane + row0 + (int64_t)(it * 8 + (lane >> 3)) * a.HWp + (lane & 7) * 8);
        }
        const int64_t blk = (((int64_t)b * 8 + wave) * ntiles + t) * 2048;         // this wave's block of this tile (kh_block)
        if (ADD == 2) {
#pragma unroll
            for (int p = 0; p < PA; ++p)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp)
                        addb[p][ct][rp] = *(const uint32_t*)(a.blocks_in + p * oplane + blk + ((ct * 8 + rp) * 64 + lane) * 2);
        }
        if (!HOIST) kh_load_a<PA>(af, a.w[m], a.w_plane, wave, lane);
        float vals[2][16];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x16_t acc = kh_gemm<PA, E, INFMT == 3>(af, lds, ct, lane);
            const int64_t px = px0 + ct * 32 + (lane & 31);
            const bool inside = px < a.HW;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 af2 = ss[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g];
                float v = fmaxf(acc[r] * af2.x + af2.y, 0.f);
                if (!inside) v = 0.f;                                    // planes are zero padded
                vals[ct][r] = v;
                if (of32 && inside) {
                    float* ub = of32 + ((int64_t)b * 256 + wave * 32 + (r & 3) + 8 * (r >> 2)) * a.HW + px0 + ct * 32;   // uniform
                    ub[(uint32_t)(4 * g * a.HW + (lane & 31))] = v;
                }
            }
        }
        if (a.blocks_out) {
            // kh_block: [ct 2][r pair 8][64 lanes] x 32 bit = accumulator registers (2 rp, 2 rp + 1) as a bf16 pair --
            // what the next launch adds to its own accumulators with 16 coalesced loads and no LDS round trip
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    uint32_t h0, l0, h1, l1;
                    kh_split<E>(vals[ct][2 * rp], h0, l0);
                    kh_split<E>(vals[ct][2 * rp + 1], h1, l1);
                    *(uint32_t*)(a.blocks_out + blk + ((ct * 8 + rp) * 64 + lane) * 2) = pack2(h0, h1);
                    if (PA == 2) *(uint32_t*)(a.blocks_out + oplane + blk + ((ct * 8 + rp) * 64 + lane) * 2) = pack2(l0, l1);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                      // every wave is done reading the input tile: its rows now take the outputs
        if (ADD) {
            // x_feats = semantic_feats + loc_feats (kernel_head.py:303): loc comes back as the bf16 plane(s) the previous
            // launch wrote (exact to 2^-17 in fp32 precision; one extra bf16 rounding of loc in bf16 precision)
            float sumv[2][16];
            if (ADD == 1) {         // from planes: through this wave's LDS rows into the accumulator layout
#pragma unroll
                for (int p = 0; p < PA; ++p)
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        *(uint4*)(rows + p * 256 * KH_LDT + (it * 8 + (lane >> 3)) * KH_LDT + (lane & 7) * 8) = addv[p][it];
                kh_wave_sync();
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ((r & 3) + 8 * (r >> 2) + 4 * g) * KH_LDT + ct * 32 + (lane & 31);
                        float s = vals[ct][r] + e2f<E>(rows[o]);
                        if (PA == 2) s += bf2f(rows[256 * KH_LDT + o]);
                        sumv[ct][r] = s;
                    }
                kh_wave_sync();
            } else {                // from register-layout blocks: no transposition at all
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp) {
                        float s0 = vals[ct][2 * rp], s1 = vals[ct][2 * rp + 1];
#pragma unroll
                        for (int p = 0; p < PA; ++p) {
                            s0 += e2f<E>(addb[p][ct][rp] & 0xFFFFu);
                            s1 += e2f<E>(addb[p][ct][rp] >> 16);
                        }
                        sumv[ct][2 * rp] = s0;
                        sumv[ct][2 * rp + 1] = s1;
                    }
            }
            if (a.f32_sum) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int64_t px = px0 + ct * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (px < a.HW) {
                            float* ub = a.f32_sum + ((int64_t)b * 256 + wave * 32 + (r & 3) + 8 * (r >> 2)) * a.HW + px0 + ct * 32;
                            ub[(uint32_t)(4 * g * a.HW + (lane & 31))] = sumv[ct][r];
                        }
                }
            }
            kh_vals_to_rows<PA, E>(sumv, rows, lane);
            kh_wave_sync();
#pragma unroll
            for (int p = 0; p < PA; ++p) kh_flush_rows(rows + p * 256 * KH_LDT, a.sum_planes + p * oplane, row0, a.HWp, lane);
            kh_wave_sync();
        }
        kh_vals_to_rows<PA, E>(vals, rows, lane);
        if (oplanes) {
            kh_wave_sync();
#pragma unroll
            for (int p = 0; p < PA; ++p) kh_flush_rows(rows + p * 256 * KH_LDT, oplanes + p * oplane, row0, a.HWp, lane);
        }
        if (a.w2) {
            // second GEMM: rows of the static 1x1 conv (init_kernels / conv_seg / conv_direct_depth, kernel_head.py:256,295,285)
            // x the normalised tile; 32-row x 32-px output tiles dealt round-robin to the waves, weights as ready-made A
            // fragments (LDS in bf16 precision, L2 otherwise)
            __syncthreads();
            const int ntile2 = a.m2_tiles * 2;
            for (int i = wave; i < ntile2; i += 8) {
                const int rt = i >> 1, ct = i & 1;
                uint4 a2[PA][16];
#pragma unroll
                for (int p = 0; p < PA; ++p)
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks)
                        a2[p][ks] = (PA == 1 && a.w2_lds) ? *(const uint4*)(w2l + ((rt * 16 + ks) * 64 + lane) * 8)
                                            : *(const uint4*)(a.w2 + p * a.w2_plane + ((int64_t)(rt * 16 + ks) * 64 + lane) * 8);
                const f32x16_t acc = kh_gemm<PA, E>(a2, lds, ct, lane);
                const int64_t px = px0 + ct * 32 + (lane & 31);
                const bool inside = px < a.HW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowu = rt * 32 + (r & 3) + 8 * (r >> 2);        // uniform part of the row
                    const int row = rowu + 4 * g;
                    const float v = acc[r] + b2l[row];
                    const bool wr = inside && row < a.n2;
                    const bool dual = a.out2b && row >= a.dual_lo && row < a.dual_lo + a.dual_n;
                    if (wr) {
                        const int64_t o1 = ((int64_t)b * a.out2_rows + rowu) * a.HW + px0 + ct * 32;                                  // uniform
                        const int64_t o2 = ((int64_t)b * a.out2b_rows + a.out2b_row0 + rowu - a.dual_lo) * a.HW + px0 + ct * 32;       // uniform
                        const uint32_t lo = (uint32_t)(4 * g * a.HW + (lane & 31));
                        if (a.out2_f16) {
                            ((uint16_t*)a.out2 + o1)[lo] = (uint16_t)f2h(v);
                            if (dual) ((uint16_t*)a.out2b + o2)[lo] = (uint16_t)f2h(v);
                        } else {
                            __builtin_nontemporal_store(v, (float*)a.out2 + o1 + lo);
                            if (dual) __builtin_nontemporal_store(v, (float*)a.out2b + o2 + lo);
                        }
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

// plain mode (the neck's output convs): the tile runs are those of a ONE-frame launch at any batch, so that a frame's partial
// sums -- and with them every bit of its outputs -- do not depend on how many frames share the launch (the video runner batches
// a whole clip per launch and promises the bits of the per-frame loop; round 6: for every B, it was B <= 3 until round 5)
// (a run length that keeps a ONE-frame launch at >= 128 workgroups per map -- 384 over the three maps -- and a 16-frame one at a
// quarter of round 5's weight re-reads: 4 tiles at cfg2's 512, 1 below 256)
static int kh_tiles_per_wg_plain(int64_t HWp, int B) {
    (void)B;
    const int ntiles = (int)(HWp / KH_T);
    int tpw = ntiles / 128;
    return tpw < 1 ? 1 : (tpw > 32 ? 32 : tpw);
}

static size_t kh_ws_bytes(int B, int64_t HW, int groups, bool plain) {
    const int64_t HWp = ph_hw_padded(HW);
    const int tpw = plain ? kh_tiles_per_wg_plain(HWp, B) : kh_tiles_per_wg(HWp, B);
    const int nwg = (int)((HWp / KH_T + tpw - 1) / tpw);
    return (size_t)3 * B * nwg * 256 * 2 * sizeof(float) + (size_t)3 * B * groups * 2 * sizeof(float);
}
extern "C" size_t ph_khead_workspace_bytes(int B, int64_t HW, int groups) { return kh_ws_bytes(B, HW, groups, false); }
extern "C" size_t ph_neck_out_convs_workspace_bytes(int B, int64_t HW, int groups) { return kh_ws_bytes(B, HW, groups, true); }

struct KhFused {                  // the static 1x1 convs of the fused entry point (all null for ph_khead_conv_gn)
    const uint16_t* w2[3];
    const float* bias2[3];
    int n2[3];
    void* out2[3];
    int out2_rows[3];
    int stuff_lo, n_stuff, n_init;
    uint16_t* loc_blocks;         // scratch for loc between the first two launches (a planes-sized buffer)
    int out2_f16;                 // logits as fp16 instead of fp32
    const unsigned* run_if;       // optional device predicate of every launch
};

// f32o (nullable): plain mode -- three independent maps (no x = sem + loc), fp32 NCHW output f32o[m] of every map that has one
static int kh_run(const void* const fm[3], int in_planes, const uint16_t* wplanes, const float* gn_affine, int groups, float eps,
                  uint16_t* const outp[3], const uint16_t* add1, uint16_t* sum1, float* x_f32, float* dfe_f32,
                  const KhFused* fu, void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec, void* stream,
                  const char* fn, float* const* f32o = nullptr) {
    if (!(B > 0 && HW > 0 && groups > 0 && 256 % groups == 0)) { ph_set_error("%s: bad size", fn); return PH_EINVAL; }
    if (!(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16)) {
        ph_set_error("%s: prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16", fn);
        return PH_EINVAL;
    }
    const bool f16 = prec == PH_PREC_F16;      // ONE fp16 plane of everything (weights packed as fp16 by the caller; plane inputs must be fp16)
    if (B > 65535) { ph_set_error("%s: B must be <= 65535", fn); return PH_EINVAL; }
    if (workspace_bytes < kh_ws_bytes(B, HW, groups, f32o != nullptr)) {
        ph_set_error("%s: workspace too small", fn);
        return PH_EWORKSPACE;
    }
    const int PA = prec == PH_PREC_SPLIT ? 2 : 1;
    const int64_t HWp = ph_hw_padded(HW);
    const int tpw = f32o ? kh_tiles_per_wg_plain(HWp, B) : kh_tiles_per_wg(HWp, B);
    const int nwg = (int)((HWp / KH_T + tpw - 1) / tpw);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* stats = partial + (size_t)3 * B * nwg * 256 * 2;
    const size_t lds = (size_t)PA * 256 * KH_LDT * sizeof(uint16_t);
    const size_t lds_apply = lds + 256 * sizeof(float2) + 256 * sizeof(float);
    static const bool once = [&] {
        const void* ks[] = {
#define KH_ALL(F) (const void*)k_khead_stats<1, F>, (const void*)k_khead_stats<2, F>, (const void*)k_khead_apply<1, 0, F>, \
    (const void*)k_khead_apply<1, 1, F>, (const void*)k_khead_apply<1, 2, F>, (const void*)k_khead_apply<2, 0, F>,         \
    (const void*)k_khead_apply<2, 1, F>, (const void*)k_khead_apply<2, 2, F>
            KH_ALL(0), KH_ALL(1), KH_ALL(2),
#define KH_F16(F) (const void*)k_khead_stats<1, F, PH_E_F16>, (const void*)k_khead_apply<1, 0, F, PH_E_F16>, \
    (const void*)k_khead_apply<1, 1, F, PH_E_F16>, (const void*)k_khead_apply<1, 2, F, PH_E_F16>
            KH_F16(0), KH_F16(1), KH_F16(2),
            (const void*)k_khead_stats<1, 3>, (const void*)k_khead_stats<2, 3>, (const void*)k_khead_stats<1, 3, PH_E_F16>,
            (const void*)k_khead_apply<1, 0, 3>, (const void*)k_khead_apply<2, 0, 3>, (const void*)k_khead_apply<1, 0, 3, PH_E_F16>
#undef KH_F16
#undef KH_ALL
        };
        for (const void* k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    KHArgs a = {};
    a.run_if = fu ? fu->run_if : nullptr;
    a.out2_f16 = fu ? fu->out2_f16 : 0;
    a.B = B; a.groups = groups; a.tiles_per_wg = tpw; a.HW = HW; a.HWp = HWp;
    a.w_plane = (int64_t)3 * 256 * 256;
    const dim3 grid(nwg, B), block(KH_THREADS);
    for (int m = 0; m < 3; ++m) {
        a.f[m] = (const float*)fm[m];
        a.fp[m] = (const uint16_t*)fm[m];
        a.w[m] = wplanes + (size_t)m * 256 * 256;
        a.partial[m] = partial + (size_t)m * B * nwg * 256 * 2;
    }
    const int fmt = in_planes == 2 ? 3 : in_planes ? 2 : ((HW % 4) == 0 ? 1 : 0);      // kernel input format: fp32 scalar / fp32 x4 / channel planes / channels-last planes
    if (fmt == 3 && !f32o) { ph_set_error("%s: channels-last input only in plain mode", fn); return PH_EINVAL; }
#define KH_LAUNCH(K, G, L, ...)                                                                      \
    do {                                                                                             \
        if (f16 && fmt == 0) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 0, PH_E_F16>), G, block, L, s, a);      \
        else if (f16 && fmt == 1) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 1, PH_E_F16>), G, block, L, s, a); \
        else if (f16) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 2, PH_E_F16>), G, block, L, s, a);             \
        else if (PA == 1 && fmt == 0) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 0>), G, block, L, s, a);    \
        else if (PA == 1 && fmt == 1) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 1>), G, block, L, s, a); \
        else if (PA == 1) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 2>), G, block, L, s, a);           \
        else if (fmt == 0) hipLaunchKernelGGL((K<2, ##__VA_ARGS__, 0>), G, block, L, s, a);          \
        else if (fmt == 1) hipLaunchKernelGGL((K<2, ##__VA_ARGS__, 1>), G, block, L, s, a);          \
        else hipLaunchKernelGGL((K<2, ##__VA_ARGS__, 2>), G, block, L, s, a);                        \
    } while (0)
    // channels-last input (plain mode only: the neck's output convs): its own two instantiations per grade
#define KH_LAUNCH3(K, G, L, ...)                                                                          \
    do {                                                                                                  \
        if (f16) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 3, PH_E_F16>), G, block, L, s, a);               \
        else if (PA == 1) hipLaunchKernelGGL((K<1, ##__VA_ARGS__, 3>), G, block, L, s, a);                \
        else hipLaunchKernelGGL((K<2, ##__VA_ARGS__, 3>), G, block, L, s, a);                             \
    } while (0)
    // pass 1: the three maps in one launch, then one finalize over the 3 * B (map, frame) pairs
    static const bool stats3_on = [] { const char* e = getenv("PH_NECK_STATS3"); return !(e && e[0] == '0'); }();   // 0: the 3-D grid (A/B)
    const bool same_input = fm[0] == fm[1] && fm[1] == fm[2] && stats3_on;
    if (same_input) {
        a.plain3 = 1; a.nwg = nwg;
        const dim3 g3((unsigned)(((nwg * B + 7) / 8) * 8 * 3));
        if (fmt == 3) KH_LAUNCH3(k_khead_stats, g3, lds);
        else KH_LAUNCH(k_khead_stats, g3, lds);
        a.plain3 = 0;
    } else if (fmt == 3) KH_LAUNCH3(k_khead_stats, dim3(nwg, B, 3), lds);
    else KH_LAUNCH(k_khead_stats, dim3(nwg, B, 3), lds);
    hipLaunchKernelGGL(k_gn_finalize, dim3(3 * B), dim3(1024), 0, s, partial, stats, nwg, groups, HW, eps, a.run_if);
    static const bool plain3_on = [] { const char* e = getenv("PH_NECK_APPLY3"); return !(e && e[0] == '0'); }();   // 0: one launch per map (A/B)
    if (f32o && plain3_on) {
        // plain mode: the three maps in ONE apply launch (see KHArgs::plain3)
        a.plain3 = 1; a.nwg = nwg; a.gn3 = gn_affine; a.stats3 = stats;
        for (int m = 0; m < 3; ++m) {
            a.planes3[m] = outp[m]; a.f32o3[m] = f32o[m];
            a.f[m] = (const float*)fm[m]; a.fp[m] = (const uint16_t*)fm[m]; a.w[m] = wplanes + (size_t)m * 256 * 256;
        }
        a.w2 = nullptr; a.out2b = nullptr; a.blocks_out = nullptr; a.blocks_in = nullptr; a.w2_lds = 0;
        const dim3 g3((unsigned)(((nwg * B + 7) / 8) * 8 * 3));
        if (fmt == 3) KH_LAUNCH3(k_khead_apply, g3, lds_apply, 0);
        else KH_LAUNCH(k_khead_apply, g3, lds_apply, 0);
        PH_CHECK_LAUNCH();
        return PH_OK;
    }
    // pass 2: loc ; sem (+ x = sem + loc, loc read back from the planes the first launch wrote) ; depth
    for (int m = 0; m < 3; ++m) {
        a.f[0] = (const float*)fm[m];
        a.fp[0] = (const uint16_t*)fm[m];
        a.w[0] = wplanes + (size_t)m * 256 * 256;
        a.gamma = gn_affine + (size_t)m * 512;
        a.beta = gn_affine + (size_t)m * 512 + 256;
        a.stats = stats + (size_t)m * B * groups * 2;
        a.planes = outp[m];
        a.add_planes = m == 1 ? add1 : nullptr;
        a.sum_planes = m == 1 ? sum1 : nullptr;
        a.f32 = m == 2 ? dfe_f32 : nullptr;
        a.f32_sum = m == 1 ? x_f32 : nullptr;
        a.w2 = nullptr; a.out2b = nullptr; a.blocks_out = nullptr; a.blocks_in = nullptr;
        int add = m == 1 ? 1 : 0;
        if (f32o) { a.add_planes = nullptr; a.sum_planes = nullptr; a.f32_sum = nullptr; a.f32 = f32o[m]; add = 0; }
        if (fu) {
            a.w2 = fu->w2[m];
            a.m2_tiles = (fu->n2[m] + 31) / 32;
            a.w2_plane = (int64_t)a.m2_tiles * 16 * 512;
            a.bias2 = fu->bias2[m];
            a.n2 = fu->n2[m];
            a.out2 = fu->out2[m];
            a.out2_rows = fu->out2_rows[m];
            if (m == 0) {                                    // loc leaves as register-layout blocks, not as planes
                a.blocks_out = fu->loc_blocks;
            }
            if (m == 1) {
                a.blocks_in = fu->loc_blocks;
                add = 2;
                if (fu->n_stuff > 0) {                       // stuff logits also go to the mask tensor (kernel_head.py:329-331)
                    a.out2b = fu->out2[0];
                    a.out2b_rows = fu->out2_rows[0];
                    a.out2b_row0 = fu->n_init;
                    a.dual_lo = fu->stuff_lo;
                    a.dual_n = fu->n_stuff;
                }
            }
        }
        a.w2_lds = (PA == 1 && a.w2 && lds_apply + (size_t)a.m2_tiles * 16 * 1024 <= 160 * 1024) ? 1 : 0;
        const size_t lds_m = lds_apply + (a.w2_lds ? (size_t)a.m2_tiles * 16 * 1024 : 0);
        if (fmt == 3) KH_LAUNCH3(k_khead_apply, grid, lds_m, 0);
        else if (add == 2) KH_LAUNCH(k_khead_apply, grid, lds_m, 2);
        else if (add == 1) KH_LAUNCH(k_khead_apply, grid, lds_m, 1);
        else KH_LAUNCH(k_khead_apply, grid, lds_m, 0);
    }
#undef KH_LAUNCH
#undef KH_LAUNCH3
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_khead_conv_gn(const float* f0, const float* f1, const float* f2, const uint16_t* wplanes,
                                const float* gn_affine, int groups, float eps, uint16_t* loc_planes,
                                uint16_t* sem_planes, uint16_t* x_planes, uint16_t* dfe_planes, float* x_f32,
                                float* dfe_f32, void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec,
                                void* stream) {
    PH_CHECK_ARG(f0 && f1 && f2 && wplanes && gn_affine && loc_planes && sem_planes && x_planes && dfe_planes && workspace,
                 "null pointer");
    const void* fm[3] = {f0, f1, f2};
    uint16_t* outp[3] = {loc_planes, sem_planes, dfe_planes};
    return kh_run(fm, 0, wplanes, gn_affine, groups, eps, outp, loc_planes, x_planes, x_f32, dfe_f32, nullptr, workspace,
                  workspace_bytes, B, HW, prec, stream, __func__);
}

// The three output convs of the neck (SemanticFPNWrapper conv_pred + 2 aux convs, semantic_fpn.py:156-178,223-231: each a
// 1x1 conv + GroupNorm + ReLU of the SAME level sum) with the two passes above: statistics from a recompute pass, then
// normalise + ReLU + store.  Against k_conv_nhwc + k_gn_finalize + k_gn_apply per map (fp32 NHWC conv output written, read
// back, converted) each map moves 16.8 MB of input per pass and its output instead of 16.8 + 33.5 + 33.5 + output MB per frame.
extern "C" int ph_neck_out_convs(const uint16_t* in_planes, int in_channels_last, const uint16_t* wplanes, const float* gn_affine,
                                 int groups, float eps, uint16_t* out_planes0, uint16_t* out_planes1, uint16_t* out_planes2,
                                 float* out_f32_0, float* out_f32_1, float* out_f32_2, void* workspace, size_t workspace_bytes,
                                 int B, int64_t HW, int prec, void* stream) {
    PH_CHECK_ARG(in_planes && wplanes && gn_affine && workspace, "null pointer");
    PH_CHECK_ARG((out_planes0 || out_f32_0) && (out_planes1 || out_f32_1) && (out_planes2 || out_f32_2), "every map needs an output");
    const void* fm[3] = {in_planes, in_planes, in_planes};
    uint16_t* outp[3] = {out_planes0, out_planes1, out_planes2};
    float* f32o[3] = {out_f32_0, out_f32_1, out_f32_2};
    return kh_run(fm, in_channels_last ? 2 : 1, wplanes, gn_affine, groups, eps, outp, nullptr, nullptr, nullptr, nullptr, nullptr,
                  workspace, workspace_bytes, B, HW, prec, stream, __func__, f32o);
}

// ph_khead_fused with (a) a device predicate -- every launch returns at once when *run_if == 0 (null: always run) -- and (b)
// the logit dtype (PH_OUT_F32 / PH_OUT_F16).  ph_khead_onepass's callers issue it behind every one-pass launch with run_if =
// that launch's status word: when the persistent launch gave up (its workgroups could not all become resident in time) the
// two-pass kernels produce the call's results instead, without a host round trip and inside a HIP graph.
extern "C" int ph_khead_fused_if(const void* f0, const void* f1, const void* f2, const uint16_t* wplanes,
                                 const float* gn_affine, int groups, float eps, const uint16_t* w2_init, int n_init,
                                 const uint16_t* w2_seg, const float* bias_seg, int n_seg, const uint16_t* w2_dd,
                                 const float* bias_dd, int stuff_lo, int n_stuff, uint16_t* x_planes, uint16_t* dfe_planes,
                                 float* x_f32, float* dfe_f32, void* mask_preds, void* seg_preds, void* depth_pred,
                                 int logits_dtype, const uint32_t* run_if, void* workspace, size_t workspace_bytes, int B,
                                 int64_t HW, int prec, int input_format, void* stream) {
    PH_CHECK_ARG(f0 && f1 && f2 && wplanes && gn_affine && x_planes && dfe_planes && workspace, "null pointer");
    PH_CHECK_ARG(input_format == PH_IN_F32_NCHW || input_format == PH_IN_PLANES, "bad input_format");
    PH_CHECK_ARG(w2_init && w2_seg && w2_dd && mask_preds && seg_preds && depth_pred, "null pointer");
    PH_CHECK_ARG(logits_dtype == PH_OUT_F32 || logits_dtype == PH_OUT_F16, "logits: PH_OUT_F32 or PH_OUT_F16");
    PH_CHECK_ARG(n_init > 0 && n_init <= 256 && n_seg > 0 && n_seg <= 256 && n_stuff >= 0 && stuff_lo >= 0 &&
                     stuff_lo + n_stuff <= n_seg, "bad row counts (at most 256 rows per static conv)");
    const void* fm[3] = {f0, f1, f2};
    // loc travels from the first to the second launch as register-layout blocks parked in the depth planes (which the
    // third launch then overwrites); sem is never stored
    uint16_t* outp[3] = {nullptr, nullptr, dfe_planes};
    KhFused fu;
    fu.w2[0] = w2_init; fu.w2[1] = w2_seg; fu.w2[2] = w2_dd;
    fu.bias2[0] = nullptr; fu.bias2[1] = bias_seg; fu.bias2[2] = bias_dd;
    fu.n2[0] = n_init; fu.n2[1] = n_seg; fu.n2[2] = 1;
    fu.out2[0] = mask_preds; fu.out2[1] = seg_preds; fu.out2[2] = depth_pred;
    fu.out2_rows[0] = n_init + n_stuff; fu.out2_rows[1] = n_seg; fu.out2_rows[2] = 1;
    fu.stuff_lo = stuff_lo; fu.n_stuff = n_stuff; fu.n_init = n_init;
    fu.loc_blocks = dfe_planes;
    fu.out2_f16 = logits_dtype == PH_OUT_F16 ? 1 : 0;
    fu.run_if = (const unsigned*)run_if;
    return kh_run(fm, input_format == PH_IN_PLANES ? 1 : 0, wplanes, gn_affine, groups, eps, outp, nullptr, x_planes, x_f32, dfe_f32, &fu, workspace,
                  workspace_bytes, B, HW, prec, stream, __func__);
}

extern "C" int ph_khead_fused(const void* f0, const void* f1, const void* f2, const uint16_t* wplanes,
                              const float* gn_affine, int groups, float eps, const uint16_t* w2_init, int n_init,
                              const uint16_t* w2_seg, const float* bias_seg, int n_seg, const uint16_t* w2_dd,
                              const float* bias_dd, int stuff_lo, int n_stuff, uint16_t* x_planes, uint16_t* dfe_planes,
                              float* x_f32, float* dfe_f32, float* mask_preds, float* seg_preds, float* depth_pred,
                              void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec, int input_format,
                              void* stream) {
    return ph_khead_fused_if(f0, f1, f2, wplanes, gn_affine, groups, eps, w2_init, n_init, w2_seg, bias_seg, n_seg, w2_dd, bias_dd,
                             stuff_lo, n_stuff, x_planes, dfe_planes, x_f32, dfe_f32, mask_preds, seg_preds, depth_pred, PH_OUT_F32,
                             nullptr, workspace, workspace_bytes, B, HW, prec, input_format, stream);
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
