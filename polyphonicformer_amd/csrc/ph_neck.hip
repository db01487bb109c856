// SURVEY.md 8(f) row N3 -- the step before the hot path: SemanticFPNWrapper.forward
// (polyphonic/funcs/semantic_fpn.py:198-235): per FPN level 3x3 conv + GroupNorm(32) + ReLU towers with x2
// bilinear upsampling, sum of the levels, three 1x1 conv + GN + ReLU outputs.
//
// Layout: inside the neck every map is CHANNELS-LAST.  A 3x3 tap is then a pixel offset of whole 512-byte
// channel vectors (no alignment problem for dx = +-1, which the [c][hw] plane format + transposing LDS reads
// cannot express), the MFMA K dimension (tap, channel) is contiguous per pixel, and GroupNorm / ReLU /
// upsample / sum are elementwise over the channel vector.
//   k_nhwc_ingest : fp32 NCHW (+ optional positional encoding, level 3) -> bf16 NHWC plane(s)
//   k_conv_nhwc   : implicit-GEMM KSxKS conv (stride 1 / 2, pad KS/2), bf16 NHWC in -> fp32 NHWC out
//                   + per-workgroup per-channel (sum, sum of squares) for the GroupNorm that follows
//   k_gn_apply    : (x - mean) * rstd * gamma + beta, ReLU, then one of: bf16 NHWC plane(s) | x2 bilinear
//                   upsample to bf16 NHWC plane(s) | accumulate into the fp32 NHWC level sum | fp32 NCHW
// GroupNorm statistics are finalised by k_gn_finalize (ph_khead.hip) from the conv kernel's partials.
#include "ph_common.h"

// ------------------------------------------------------------------------------------------------------------
// fp32 NCHW [B][256][HW] (+ add[256][HW], nullable) -> bf16 NHWC planes [PA][B][HW][256]; 64 channels x 64 pixels per
// block (256-byte runs in, whole 128-byte lines out).  A 256-channel x 32-pixel variant that writes whole 512-byte
// pixel vectors was 50 % slower at the stride-4 level (256 rows 512 KB apart per block).
template <int PA, int E = PH_E_BF16, bool C16 = false>
__global__ __launch_bounds__(256) void k_nhwc_ingest(const float* __restrict__ src, const float* __restrict__ add,
                                                     uint16_t* __restrict__ dst, int B, int64_t HW) {
    __shared__ float t[64][65];                       // [channel][pixel] tile
    const int b = blockIdx.z, c0 = blockIdx.y * 64;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4) {
        const int64_t p = p0 + tx;
        float v = 0.f;
        if (p < HW) {
            v = __builtin_nontemporal_load(src + ((int64_t)b * 256 + c0 + c) * HW + p);
            if (add) v += add[(int64_t)(c0 + c) * HW + p];
        }
        t[c][tx] = v;
    }
    __syncthreads();
    const int64_t plane = (int64_t)B * HW * 256;
    for (int p = ty; p < 64; p += 4) {
        if (p0 + p >= HW) continue;
        uint32_t hi, lo;
        f2e_split<E>(t[tx][p], hi, lo);
        // C16: [frame][16-channel chunk][pixel][16] (PH_PLANES_C16, the stride-2 conv's input)
        const int64_t o = C16 ? (((int64_t)b * 16 + ((c0 + tx) >> 4)) * HW + p0 + p) * 16 + ((c0 + tx) & 15)
                              : ((int64_t)b * HW + p0 + p) * 256 + c0 + tx;
        dst[o] = (uint16_t)hi;
        if (PA == 2) dst[o + plane] = (uint16_t)lo;
    }
}

// the same tile with 16-byte accesses on both sides (HW % 4 == 0): 4 x float4 per thread in, 2 x 16 bytes (8 channels of
// one pixel) per thread out -- a quarter / an eighth of the memory instructions of the scalar kernel above
template <int PA, int E = PH_E_BF16, bool C16 = false>
__global__ __launch_bounds__(256) void k_nhwc_ingest_v4(const float* __restrict__ src, const float* __restrict__ add,
                                                        uint16_t* __restrict__ dst, int B, int64_t HW) {
    __shared__ float t[64][65];                       // [channel][pixel] tile
    const int b = blockIdx.z, c0 = blockIdx.y * 64;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int r16 = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
    const int64_t p = p0 + c4;
    uint4 q[4], qa[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                     // all loads of the tile in flight together
        const int c = k * 16 + r16;
        q[k] = make_uint4(0, 0, 0, 0);
        qa[k] = make_uint4(0, 0, 0, 0);
        if (p < HW) {                                 // HW % 4 == 0: the four pixels are inside together
            q[k] = ld_nt16(src + ((int64_t)b * 256 + c0 + c) * HW + p);
            if (add) qa[k] = *(const uint4*)(add + (int64_t)(c0 + c) * HW + p);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = k * 16 + r16;
        t[c][c4 + 0] = __uint_as_float(q[k].x) + __uint_as_float(qa[k].x);
        t[c][c4 + 1] = __uint_as_float(q[k].y) + __uint_as_float(qa[k].y);
        t[c][c4 + 2] = __uint_as_float(q[k].z) + __uint_as_float(qa[k].z);
        t[c][c4 + 3] = __uint_as_float(q[k].w) + __uint_as_float(qa[k].w);
    }
    __syncthreads();
    const int64_t plane = (int64_t)B * HW * 256;
    const int piece = threadIdx.x & 7, px0 = threadIdx.x >> 3;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int px = k * 32 + px0;
        if (p0 + px >= HW) continue;
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f2e_split<E>(t[piece * 8 + e][px], hi[e], lo[e]);
        uint16_t* d = C16 ? dst + (((int64_t)b * 16 + ((c0 + piece * 8) >> 4)) * HW + p0 + px) * 16 + ((piece & 1) * 8)
                          : dst + ((int64_t)b * HW + p0 + px) * 256 + c0 + piece * 8;
        *(uint4*)d = make_uint4(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7]));
        if (PA == 2) *(uint4*)(d + plane) = make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(lo[4], lo[5]), pack2(lo[6], lo[7]));
    }
}

extern "C" int ph_nhwc_ingest(const float* src, const float* add, uint16_t* dst, int B, int64_t HW, int prec, void* stream) {
    PH_CHECK_ARG(src && dst && B > 0 && HW > 0, "bad pointer or size");
    const bool c16 = (prec & PH_PLANES_C16) != 0;
    prec &= ~PH_PLANES_C16;
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16, "prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16");
    PH_CHECK_ARG(!c16 || prec != PH_PREC_SPLIT, "PH_PLANES_C16 output: one-plane formats only");
    const dim3 grid((unsigned)((HW + 63) / 64), 4, B);
    if (c16) {
        hipStream_t st = (hipStream_t)stream;
        if (prec == PH_PREC_F16) {
            if ((HW & 3) == 0) hipLaunchKernelGGL((k_nhwc_ingest_v4<1, PH_E_F16, true>), grid, dim3(256), 0, st, src, add, dst, B, HW);
            else hipLaunchKernelGGL((k_nhwc_ingest<1, PH_E_F16, true>), grid, dim3(256), 0, st, src, add, dst, B, HW);
        } else {
            if ((HW & 3) == 0) hipLaunchKernelGGL((k_nhwc_ingest_v4<1, PH_E_BF16, true>), grid, dim3(256), 0, st, src, add, dst, B, HW);
            else hipLaunchKernelGGL((k_nhwc_ingest<1, PH_E_BF16, true>), grid, dim3(256), 0, st, src, add, dst, B, HW);
        }
        PH_CHECK_LAUNCH();
        return PH_OK;
    }
    if (prec == PH_PREC_F16) {
        if ((HW & 3) == 0) hipLaunchKernelGGL((k_nhwc_ingest_v4<1, PH_E_F16>), grid, dim3(256), 0, (hipStream_t)stream, src, add, dst, B, HW);
        else hipLaunchKernelGGL((k_nhwc_ingest<1, PH_E_F16>), grid, dim3(256), 0, (hipStream_t)stream, src, add, dst, B, HW);
    } else if ((HW & 3) == 0) {
        if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_nhwc_ingest_v4<1>, grid, dim3(256), 0, (hipStream_t)stream, src, add, dst, B, HW);
        else hipLaunchKernelGGL(k_nhwc_ingest_v4<2>, grid, dim3(256), 0, (hipStream_t)stream, src, add, dst, B, HW);
    } else if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_nhwc_ingest<1>, grid, dim3(256), 0, (hipStream_t)stream, src, add, dst, B, HW);
    else hipLaunchKernelGGL(k_nhwc_ingest<2>, grid, dim3(256), 0, (hipStream_t)stream, src, add, dst, B, HW);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution, M = output pixels, N = 256 output channels, K = KS*KS*256 ordered (tap, channel).
// Workgroup (8 waves) = TH output rows x 64 output pixels x all 256 channels (TH = 4 or 2, ConvGeo); wave w owns
// channels 32 w .. 32 w + 31 of all TH * 64 pixels: 2 TH 32-pixel M tiles of 32x32x16 MFMA.  Per k-step a wave reads 2 TH A
// fragments from LDS and ONE B fragment from L2.  (-DCV_WAVES_2X4: wave = 64 channels x half of the rows, half the A reads
// and twice the B stream -- measured equal, 611 against 604 us at 128 x 256 x 16 frames: timing ablations put the A reads at
// 46 us and the weight stream at 24 of the 604, the patch prefetch at 40 and the output stores at 60.)
// Round 5, from those ablations: (i) the next chunk's patch is requested ONE 16-byte piece per few k-steps instead of as
// a burst behind the first weight fragments -- vector loads of a wave retire in order, so every weight fragment requested
// after the burst waited for all of it (stride 2: 207 of 806 us); the loads are unconditional (clamped addresses, zeros
// selected at the LDS write) so that hipcc's vmcnt counting stays exact; (ii) the epilogue transposes each wave's
// accumulators through LDS and stores 16 bytes per lane: a quarter of the store instructions (the tail of a workgroup is
// store-ISSUE bound and nothing overlaps it at one workgroup per CU).
// The input patch ((2-1)*S + KS rows x (64-1)*S + KS pixels) is staged through LDS one channel chunk at a
// time ([pixel][CH + 8] bf16: the 16-byte pad makes the 16 lanes of a ds_read_b128 group hit 16 distinct slots);
// an A fragment is one ds_read_b128 at a pixel offset given by the tap.  Weights are pre-packed B fragments
// ([col tile][k-step] blocks of 1 KiB, pack.pack_b32) streamed from L2, one k-step ahead.
// Roofline: MFMA (2*9*256*256 flop per output pixel); L2 weight stream = 1.18 MB per 128 output pixels.
constexpr int CV_TW = 64;

template <int KS, int S, int PA, int TH_ = (PA == 1 ? 4 : 2)> struct ConvGeo {
    // output rows per workgroup: 4 for the double-buffered single-plane kernels (one workgroup per CU, 128 accumulator
    // VGPRs per wave, every weight fragment feeds 8 MFMAs; stride 2 takes 16-channel chunks so that two 9-row patches
    // fit in LDS), 2 otherwise -- and, round 4, 2 for the single-plane kernels when 4-row tiles would leave CUs idle (one
    // frame per launch, the video loop: a 128 x 256 map is 128 four-row tiles, and a tile's 144 k-steps are a serial
    // chain of ~35 us whatever the map size; two-row tiles are twice the workgroups at half the chain)
    static constexpr int TH = TH_;
    static constexpr int MT = TH * 2;                                              // 32-pixel M tiles per wave
    static constexpr int IR = (TH - 1) * S + KS, IC = (CV_TW - 1) * S + KS;       // input patch rows / cols
    static constexpr int CH = (S == 1) ? 64 : (PA == 1 ? 16 : 32);                 // channels per LDS stage
    static constexpr int LDP = CH + 8;                                             // pixel stride (elements)
    // stride 2: the patch columns are stored de-interleaved (even columns, then odd columns), so that the 32 lanes of an
    // A-fragment read (consecutive OUTPUT pixels = every second input column) walk consecutive LDS pixels like the
    // stride-1 kernels do
    static constexpr int HALF = (IC + 1) / 2;
    static constexpr int ICS = (S == 2) ? 2 * HALF : IC;                           // stored columns per patch row
    static constexpr int PLANE = IR * ICS * LDP;                                   // elements per precision plane
};

template <int PA, int KS, int S, int E = PH_E_BF16, int TH_ = (PA == 1 ? 4 : 2)>
__global__ __launch_bounds__(512) void k_conv_nhwc(const uint16_t* __restrict__ X, int64_t x_plane,
                                                   const uint16_t* __restrict__ Wp, int64_t w_plane, float* __restrict__ Y,
                                                   float* __restrict__ partial, int B, int H, int W, int Ho, int Wo, int c16) {
    using G = ConvGeo<KS, S, PA, TH_>;
    constexpr int CH = G::CH, LDP = G::LDP, IR = G::IR, IC = G::IC, PAD = KS / 2;
    constexpr int KSTEPS_TOTAL = KS * KS * 256 / 16;
#ifndef CV_DEPTH
#define CV_DEPTH 4
#endif
#ifndef CV_ABL
#define CV_ABL 0          // timing-only ablations (wrong results): 1 no patch prefetch, 2 no weight stream, 4 no stores, 8 no A reads, 16 patch pieces from one 16 KB window, 32 no patch LDS writes, 64 every chunk's k-steps twice
#endif
    constexpr int DEPTH = CV_DEPTH;                                         // B fragments requested this many k-steps ahead
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];        // [PA][IR][IC][LDP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef CV_WAVES_2X4
    constexpr int NT = 1, MT = G::MT;                                       // wave = 32 channels x all rows
    const int wc = wv, row_base = 0;
#else
    constexpr int NT = 2, MT = G::TH;                                       // wave = 64 channels x half of the rows
    const int wc = wv & 3, row_base = (wv >> 2) * (G::TH / 2);
#endif
    const int b = blockIdx.z, oy0 = blockIdx.y * G::TH, ox0 = blockIdx.x * CV_TW;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const int m = lane & 31, kg = lane >> 5;

    f32x16_t acc[MT][NT];     // [output row (of this wave) * 2 + 32-pixel half][column tile]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const uint16_t* xb = X + (int64_t)b * H * W * 256;
    const uint16_t* wbase = Wp + ((int64_t)wc * NT * KSTEPS_TOTAL) * 512 + lane * 8;   // this lane's B-fragment stream(s)
    constexpr int64_t W_NT = (int64_t)KSTEPS_TOTAL * 512;                             // next column tile

    // Stride-1 single-plane instantiations double-buffer the patch: chunk c + 1 travels HBM -> registers while the
    // MFMAs of chunk c run and is written to the other LDS buffer after them (one barrier per chunk).  Vector-memory
    // operations of a wave retire in order, so the prefetch is issued AFTER the first DEPTH weight fragments: the
    // k-steps that use those do not wait for it.
    constexpr bool DB = (PA == 1);
    constexpr int PIECES = CH / 8;
    constexpr int NPRE = (IR * IC * PIECES + 511) / 512;
    // k-th 16-byte piece of this thread: ALWAYS loaded (clamped to the image and to the patch; the select happens at the
    // LDS write), so that no branch sits between the weight-fragment loads.  Where the piece comes from and where it goes
    // does not depend on the chunk: the index arithmetic (two divisions per piece) is done ONCE -- recomputed per chunk it
    // was 15 % of the stride-1 kernel and 30 % of the stride-2 one (16 chunks), VALU work in front of the MFMAs.
    int goff[DB ? NPRE : 1], loff[DB ? NPRE : 1];         // plane element offset; LDS element offset | zero flag, -1 = no piece
    if (DB) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int idx = tid + k * 512, idc = min(idx, IR * IC * PIECES - 1);
            const int piece = idc % PIECES, pix = idc / PIECES;
            const int r = pix / IC, x = pix - r * IC;
            const int iy = iy0 + r, ix = ix0 + x;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int slot = r * G::ICS + (S == 2 ? (x & 1) * G::HALF + (x >> 1) : x);
            // PH_PLANES_C16 input (16-channel stages only): [chunk][pixel][16] -- a stage's rows are contiguous in memory
            goff[k] = (min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)) * (c16 ? 16 : 256) + piece * 8;
            if (CV_ABL & 16) goff[k] &= 0x1ff8;           // timing only: every piece from one 16 KB window (cache hits)
            loff[k] = idx < IR * IC * PIECES ? ((slot * LDP + piece * 8) | (in ? 0 : 0x40000000)) : -1;
        }
    }
    auto patch_load = [&](int c0, int k, uint4 (&v)[PA]) {
#pragma unroll
        for (int p = 0; p < PA; ++p) v[p] = *(const uint4*)(xb + p * x_plane + goff[k] + (c16 ? (int64_t)(c0 >> 4) * H * W * 16 : (int64_t)c0));
    };
    auto patch_put = [&](uint16_t* buf, int k, const uint4 (&v)[PA]) {
        if (loff[k] >= 0) {
            const bool zero = (loff[k] & 0x40000000) != 0;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                if ((CV_ABL & 32) && v[p].x != 0x12345678u) continue;     // timing only: loads awaited, no LDS write
                *(uint4*)(buf + p * G::PLANE + (loff[k] & 0xffffff)) = zero ? make_uint4(0, 0, 0, 0) : v[p];
            }
        }
    };
    auto patch_load_cond = [&](int c0, int k, uint4 (&v)[PA]) {      // the kernels without a prefetch: only what is inside
        const int idx = tid + k * 512;
        const int piece = idx % PIECES, pix = idx / PIECES;
        const int r = pix / IC, x = pix - r * IC;
        const int iy = iy0 + r, ix = ix0 + x;
        const bool in = idx < IR * IC * PIECES && iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            v[p] = make_uint4(0, 0, 0, 0);
            if (in) v[p] = *(const uint4*)(xb + p * x_plane + ((int64_t)iy * W + ix) * 256 + c0 + piece * 8);
        }
    };
    auto patch_store = [&](uint16_t* buf, int k, const uint4 (&v)[PA]) {
        const int idx = tid + k * 512;
        if (idx < IR * IC * PIECES) {
            const int piece = idx % PIECES, pix = idx / PIECES;
            const int r = pix / IC, x = pix - r * IC;
            const int slot = r * G::ICS + (S == 2 ? (x & 1) * G::HALF + (x >> 1) : x);
#pragma unroll
            for (int p = 0; p < PA; ++p) *(uint4*)(buf + p * G::PLANE + slot * LDP + piece * 8) = v[p];
        }
    };
    if (DB) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            uint4 v[PA];
            patch_load(0, k, v);
            patch_put(lds, k, v);
        }
        __syncthreads();
    }

    for (int c0 = 0; c0 < 256; c0 += CH) {
        const uint16_t* cur = lds + (DB ? ((c0 / CH) & 1) * (PA * G::PLANE) : 0);
        if (!DB) {
            __syncthreads();                              // previous chunk's readers are done
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                uint4 v[PA];
                patch_load_cond(c0, k, v);
                patch_store(lds, k, v);
            }
            __syncthreads();
        }
        // ---- MFMAs of this chunk: taps x (CH / 16) k-steps; the wave's B fragment of k-step j + DEPTH is requested
        //      while k-step j runs (L2 latency is several k-steps long)
        constexpr int NK = KS * KS * (CH / 16);
        auto kstep_of = [&](int j) {                      // j-th k-step of the chunk -> global k-step index
            const int tap = j / (CH / 16), kk = j - tap * (CH / 16);
            return tap * 16 + (c0 >> 4) + kk;             // (tap * 256 + c0 + kk * 16) / 16
        };
        uint4 bq[DEPTH][NT][PA];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < NK) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int p = 0; p < PA; ++p) bq[d][nt][p] = *(const uint4*)(wbase + nt * W_NT + p * w_plane + (int64_t)kstep_of(d) * 512);
            }
        uint4 pre[DB ? NPRE : 1][PA];
        const bool more = DB && c0 + CH < 256;
        const int cn = min(c0 + CH, 256 - CH);            // the last chunk re-requests itself (L2 hits) rather than branch
        constexpr int NKS = (NK - DEPTH > 1) ? NK - DEPTH : 1;   // the pieces are spread over the first NKS k-steps
#if CV_ABL & 8
        uint4 a[PA][MT];
#endif
#if CV_ABL & 64            // timing only: the chunk's k-steps run twice (is the loop or what surrounds it the overhead?)
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep)
#endif
#pragma unroll
        for (int j = 0; j < NK; ++j) {
#if !(CV_ABL & 8)
            uint4 a[PA][MT];
#endif
            uint4 bcur[NT][PA];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int p = 0; p < PA; ++p) bcur[nt][p] = bq[j % DEPTH][nt][p];
            if (j + DEPTH < NK && !(CV_ABL & 2)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int p = 0; p < PA; ++p)
                        bq[j % DEPTH][nt][p] = *(const uint4*)(wbase + nt * W_NT + p * w_plane + (int64_t)kstep_of(j + DEPTH) * 512);
            }
            if (DB && !(CV_ABL & 1)) {
#pragma unroll
                for (int k = 0; k < NPRE; ++k)
                    if (k >= (j * NPRE + NKS - 1) / NKS && (k < ((j + 1) * NPRE + NKS - 1) / NKS || j == NK - 1)) patch_load(cn, k, pre[k]);
            }
            const int tap = j / (CH / 16), kk = j - tap * (CH / 16);
            const int dy = tap / KS, dx = tap - dy * KS;
            if (!(CV_ABL & 8) || j == 0)
#pragma unroll
            for (int p = 0; p < PA; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int row = (row_base + (mt >> 1)) * S + dy;
                    const int px = (S == 2) ? (dx & 1) * G::HALF + (mt & 1) * 32 + m + (dx >> 1) : (mt & 1) * 32 + m + dx;
                    a[p][mt] = *(const uint4*)(cur + p * G::PLANE + (row * G::ICS + px) * LDP + kk * 16 + kg * 8);
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = mfma32e<E>(a[0][mt], bcur[nt][0], acc[mt][nt]);
                    if (PA == 2) {
                        acc[mt][nt] = mfma32e<E>(a[0][mt], bcur[nt][PA - 1], acc[mt][nt]);
                        acc[mt][nt] = mfma32e<E>(a[PA - 1][mt], bcur[nt][0], acc[mt][nt]);
                    }
                }
            if (MT * NT > 4) __builtin_amdgcn_sched_barrier(0);   // 128 accumulator VGPRs: keep the A fragments of later k-steps out
        }
        if (DB) {
            if (more) {
                uint16_t* nxt = lds + (((c0 / CH) + 1) & 1) * (PA * G::PLANE);
#pragma unroll
                for (int k = 0; k < NPRE; ++k) patch_put(nxt, k, pre[k]);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: GroupNorm partial sums from the accumulators; fp32 NHWC store through a per-wave LDS transposition
    //      (C layout: a lane holds 16 pixels of ONE channel; memory wants 4 consecutive channels of one pixel per lane).  One
    //      output row of the wave at a time: 64 pixels x 32 channels = 8 KB per wave, pixel rows swizzled by bit 2 (= kg on the
    //      write side) so that both the 4-byte writes and the 16-byte reads touch every bank once; then 8 stores of 16 bytes
    //      per lane (8 pixels x 128 contiguous bytes per instruction) instead of 32 of 4 bytes.
    // GroupNorm partial sums: ONE entry per PAIR of output rows (rows 2p, 2p + 1 of the image x this tile's 64 pixels), in the 2-row
    // kernel's order -- the lane's two rows x two halves x 16 registers, then the two lane halves -- whatever TH is (round 6): a 4-row
    // tile emits two entries, so the sums `ph_gn_finalize` adds, and with them every bit of the normalised output, do not depend on
    // the tile form the launch size picked (batch-invariant kernels; the 2 x 4 wave arrangement of -DCV_WAVES_2X4
    // keeps the older per-tile entries).
    float s1[NT], s2[NT];
    float* tbuf = (float*)lds + wv * 2048;
    const int64_t npair_x = gridDim.x, npairs = (int64_t)gridDim.x * ((Ho + 1) / 2);
    // (the single-plane kernels: the last chunk's closing barrier has passed, the LDS is free)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = 0.f;
        s2[nt] = 0.f;
#pragma unroll
        for (int row = 0; row < MT / 2; ++row) {
            const int oy = oy0 + row_base + row;
#ifndef CV_WAVES_2X4
            if ((row & 1) == 0) { s1[nt] = 0.f; s2[nt] = 0.f; }
#endif
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;  // C layout: row = pixel, col = channel
                    const float v = acc[row * 2 + half][nt][r];
                    if (DB) tbuf[(px ^ kg) * 32 + m] = v;
                    if (oy < Ho && ox0 + px < Wo) {
                        if (!DB) Y[(((int64_t)b * Ho + oy) * Wo + ox0 + px) * 256 + (wc * NT + nt) * 32 + m] = v;
                        s1[nt] += v;
                        s2[nt] += v * v;
                    }
                }
            if (DB && oy < Ho) {                          // (the two-plane kernels keep the direct stores above)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int P = (lane >> 3) + 8 * i;
                    const float4 v = *(const float4*)(tbuf + (P ^ ((P >> 2) & 1)) * 32 + (lane & 7) * 4);
                    if ((CV_ABL & 4) && v.x != 12345.f) continue;
                    if (ox0 + P < Wo) *(float4*)(Y + (((int64_t)b * Ho + oy) * Wo + ox0 + P) * 256 + (wc * NT + nt) * 32 + (lane & 7) * 4) = v;
                }
            }
#ifndef CV_WAVES_2X4
            if ((row & 1) == 1 || row == MT / 2 - 1) {
                // the pair (oy - 1, oy) is complete: its entry (pairs past the image's last row have none)
                const float t1 = s1[nt] + __shfl_xor(s1[nt], 32), t2 = s2[nt] + __shfl_xor(s2[nt], 32);
                const int pair_y = (oy0 + row_base + (row & ~1)) >> 1;
                if (kg == 0 && oy0 + row_base + (row & ~1) < Ho) {
                    float* o = partial + (((int64_t)b * npairs + (int64_t)pair_y * npair_x + blockIdx.x) * 256 + (wc * NT + nt) * 32 + m) * 2;
                    o[0] = t1;
                    o[1] = t2;
                }
            }
#endif
        }
#ifdef CV_WAVES_2X4
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
#endif
    }
#ifdef CV_WAVES_2X4
    const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, nwg = (int64_t)gridDim.x * gridDim.y;
    // the two row halves of a channel meet in LDS (fixed order: upper rows + lower rows)
    float* red = (float*)lds;                             // [256 channels][2]
    __syncthreads();                                      // the transposition buffers are done
    if (row_base != 0 && kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            red[((wc * NT + nt) * 32 + m) * 2 + 0] = s1[nt];
            red[((wc * NT + nt) * 32 + m) * 2 + 1] = s2[nt];
        }
    }
    __syncthreads();
    if (row_base == 0 && kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = (wc * NT + nt) * 32 + m;
            float* o = partial + (((int64_t)b * nwg + wg) * 256 + c) * 2;
            o[0] = s1[nt] + red[c * 2 + 0];
            o[1] = s2[nt] + red[c * 2 + 1];
        }
    }
#endif
}

// one-plane formats: 4-row tiles, unless those are fewer than one per CU over the whole launch (then 2-row tiles).  The choice may
// follow the launch size: the kernels' GroupNorm partial sums are per PAIR of output rows in either form (round 6), so a frame's bits do
// not depend on it (tests/test_gpu_neck.py::test_conv_tile_forms_give_identical_bits, tests/test_gpu_video.py::test_heads_are_batch_invariant)
static int conv_th(int Ho, int Wo, int prec, int B) {
    if (prec == PH_PREC_SPLIT) return 2;
    static const int force = [] { const char* e = getenv("PH_CONV_TH"); return e ? atoi(e) : 0; }();     // 2 / 4: A/B timing, tests
    if (const char* e = getenv("PH_CONV_TH_NOW")) { const int f = atoi(e); if (f == 2 || f == 4) return f; }   // tests: read per call
    if (force == 2 || force == 4) return force;
    const int64_t t4 = (int64_t)((Wo + CV_TW - 1) / CV_TW) * ((Ho + 3) / 4);
    return B * t4 < 256 ? 2 : 4;
}

// entries per frame of ph_conv_nhwc's `partial` output = pairs of output rows x 64-pixel column tiles (whatever the tile form;
// `ksize`, `stride`, `prec`, `B` are kept for the callers of rounds 3-5, when the count was the workgroups of the chosen form)
extern "C" int ph_conv_nhwc_workgroups_b(int ksize, int stride, int Ho, int Wo, int prec, int B) {
    (void)ksize; (void)stride; (void)prec; (void)B;
    return ((Wo + CV_TW - 1) / CV_TW) * ((Ho + 1) / 2);
}
// (round 5: the B-less `ph_conv_nhwc_workgroups` is REMOVED -- it answered for launches of >= 256 four-row tiles only and
// handed ph_gn_finalize a wrong count and partial stride on small launches; a stale caller now fails at link / dlsym time)

// upper bound over all instantiations (2-row tiles)
extern "C" size_t ph_conv_nhwc_partial_floats(int B, int Ho, int Wo) {
    return (size_t)B * ((Wo + CV_TW - 1) / CV_TW) * ((Ho + 1) / 2) * 256 * 2;
}

extern "C" int ph_conv_nhwc(const uint16_t* X, const uint16_t* Wp, int64_t w_plane_elems, float* Y, float* partial, int ksize,
                            int stride, int B, int H, int W, int prec, void* stream) {
    PH_CHECK_ARG(X && Wp && Y && partial && B > 0 && H > 0 && W > 0, "bad pointer or size");
    PH_CHECK_ARG((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1), "supported: 3x3 stride 1/2, 1x1 stride 1");
    const int c16 = (prec & PH_PLANES_C16) ? 1 : 0;
    prec &= ~PH_PLANES_C16;
    PH_CHECK_ARG(!c16 || (ksize == 3 && stride == 2 && prec != PH_PREC_SPLIT), "PH_PLANES_C16 input: the one-plane 3x3 stride-2 kernel only");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16, "prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16");
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const int th = conv_th(Ho, Wo, prec, B);
    const dim3 grid((Wo + CV_TW - 1) / CV_TW, (Ho + th - 1) / th, B);
    const int64_t x_plane = (int64_t)B * H * W * 256;
    hipStream_t s = (hipStream_t)stream;
#define PH_CV_T(PA, KS, S, EE, TT)                                                                                       \
    do {                                                                                                                 \
        size_t lds = (size_t)(PA == 1 ? 2 : 1) * PA * ConvGeo<KS, S, PA, TT>::PLANE * sizeof(uint16_t);                  \
        if (lds < 65536) lds = 65536;                          /* the epilogue's 8 x 8 KB transposition buffers */        \
        static const bool once = [&] {                                                                                   \
            (void)hipFuncSetAttribute((const void*)k_conv_nhwc<PA, KS, S, EE, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                         \
            return true;                                                                                                 \
        }();                                                                                                             \
        (void)once;                                                                                                      \
        hipLaunchKernelGGL((k_conv_nhwc<PA, KS, S, EE, TT>), grid, dim3(512), lds, s, X, x_plane, Wp, w_plane_elems, Y, partial, \
                           B, H, W, Ho, Wo, c16);                                                                        \
    } while (0)
#define PH_CV(PA, KS, S, EE)                                                                                             \
    do {                                                                                                                 \
        if (PA == 1 && th == 4) PH_CV_T(PA, KS, S, EE, 4); else PH_CV_T(PA, KS, S, EE, 2);                               \
    } while (0)
    if (prec == PH_PREC_BF16) {
        if (ksize == 1) PH_CV(1, 1, 1, PH_E_BF16); else if (stride == 1) PH_CV(1, 3, 1, PH_E_BF16); else PH_CV(1, 3, 2, PH_E_BF16);
    } else if (prec == PH_PREC_F16) {
        if (ksize == 1) PH_CV(1, 1, 1, PH_E_F16); else if (stride == 1) PH_CV(1, 3, 1, PH_E_F16); else PH_CV(1, 3, 2, PH_E_F16);
    } else {
        if (ksize == 1) PH_CV(2, 1, 1, PH_E_BF16); else if (stride == 1) PH_CV(2, 3, 1, PH_E_BF16); else PH_CV(2, 3, 2, PH_E_BF16);
    }
#undef PH_CV
#undef PH_CV_T
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ------------------------------------------------------------------------------------------------------------
// GroupNorm affine + ReLU on fp32 NHWC, elementwise over 4-channel vectors; `mode`:
//   PH_GN_TO_PLANES   bf16 NHWC plane(s), same resolution
//   PH_GN_UP2_PLANES  bf16 NHWC plane(s) at 2H x 2W, bilinear align_corners=False (F.interpolate / nn.Upsample)
//   PH_GN_ACCUM       fp32 NHWC accumulator (+= when accumulate != 0, = otherwise)
//   PH_GN_TO_NCHW     fp32 NCHW (the format KernelHead takes)
// stats == nullptr: no normalisation, no ReLU (plain conversion of the level sum to planes).
__device__ __forceinline__ float4 gn_relu4(float4 v, float4 sc, float4 sh, bool act) {
    float4 o = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    if (act) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    return o;
}

template <int PA, int E = PH_E_BF16>
__device__ __forceinline__ void st_planes4(uint16_t* dst, int64_t plane, float4 o) {
    uint32_t h[4], l[4];
    f2e_split<E>(o.x, h[0], l[0]); f2e_split<E>(o.y, h[1], l[1]); f2e_split<E>(o.z, h[2], l[2]); f2e_split<E>(o.w, h[3], l[3]);
    *(uint2*)dst = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    if (PA == 2) *(uint2*)(dst + plane) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}

template <int PA, int E = PH_E_BF16>
__global__ __launch_bounds__(256) void k_gn_apply(const float* __restrict__ y, const float* __restrict__ stats,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int groups,
                                                  int mode, int accumulate, uint16_t* __restrict__ planes,
                                                  float* __restrict__ outf, int B, int H, int W) {
    const int b = blockIdx.z;
    const int c4 = (threadIdx.x & 63) * 4;                       // 4 channels per thread, 64 threads per pixel
    const int cpg = 256 / groups;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool act = stats != nullptr;
    if (stats) {
        float s_[4], h_[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* st = stats + ((int64_t)b * groups + (c4 + e) / cpg) * 2;
            const float g = st[1] * gamma[c4 + e];
            s_[e] = g;
            h_[e] = beta[c4 + e] - st[0] * g;
        }
        sc = make_float4(s_[0], s_[1], s_[2], s_[3]);
        sh = make_float4(h_[0], h_[1], h_[2], h_[3]);
    }
    const float* yb = y + (int64_t)b * H * W * 256;
    if (mode == PH_GN_UP2_PLANES) {
        const int Ho = 2 * H, Wo = 2 * W;
        const int64_t plane = (int64_t)B * Ho * Wo * 256;
        // A wave takes the 2x2 output block anchored at the ODD output coordinates (2y+1, 2x+1): its four pixels blend the
        // same four source pixels (y, x) .. (y+1, x+1) with weights {.75, .25} x {.75, .25}, so a source vector is loaded and
        // normalised once per block instead of once per output pixel (4 loads per 4 outputs, was 16).  Blocks y = -1 /
        // x = -1 / y = H-1 / x = W-1 hold the first / last output row / column: their taps are the clamped ones of the
        // per-pixel formula below, which every output goes through, so the values are those of the per-pixel kernel.
        const int nbx = W + 1;
        const int64_t nblk = (int64_t)(H + 1) * nbx;
        for (int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); q < nblk; q += (int64_t)gridDim.x * 4) {
            const int by = (int)(q / nbx) - 1, bx = (int)(q - (int64_t)(by + 1) * nbx) - 1;
            const int r0 = by < 0 ? 0 : by, r1 = by + 1 > H - 1 ? H - 1 : by + 1;
            const int k0 = bx < 0 ? 0 : bx, k1 = bx + 1 > W - 1 ? W - 1 : bx + 1;
            const float4 s00 = gn_relu4(*(const float4*)(yb + ((int64_t)r0 * W + k0) * 256 + c4), sc, sh, act);
            const float4 s01 = gn_relu4(*(const float4*)(yb + ((int64_t)r0 * W + k1) * 256 + c4), sc, sh, act);
            const float4 s10 = gn_relu4(*(const float4*)(yb + ((int64_t)r1 * W + k0) * 256 + c4), sc, sh, act);
            const float4 s11 = gn_relu4(*(const float4*)(yb + ((int64_t)r1 * W + k1) * 256 + c4), sc, sh, act);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oy = 2 * by + 1 + dy, ox = 2 * bx + 1 + dx;
                    if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                    // source coordinate (o + 0.5) / 2 - 0.5, clamped at 0 (PyTorch area_pixel_compute_source_index)
                    const float fy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f);
                    const int y0 = (int)fy, x0 = (int)fx;
                    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
                    const float ly = fy - y0, lx = fx - x0;
                    // (y0, y1) is (r0, r1) except in the clamped border blocks, where a tap with weight 0 may lie outside
                    // the block: pick by index, the weight-0 operand only has to be finite
                    const float4 v00 = y0 == r0 ? (x0 == k0 ? s00 : s01) : (x0 == k0 ? s10 : s11);
                    const float4 v01 = y0 == r0 ? (x1 == k1 ? s01 : s00) : (x1 == k1 ? s11 : s10);
                    const float4 v10 = y1 == r1 ? (x0 == k0 ? s10 : s11) : (x0 == k0 ? s00 : s01);
                    const float4 v11 = y1 == r1 ? (x1 == k1 ? s11 : s10) : (x1 == k1 ? s01 : s00);
                    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
                    float4 o;
                    o.x = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
                    o.y = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
                    o.z = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
                    o.w = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
                    st_planes4<PA, E>(planes + ((int64_t)b * Ho * Wo + (int64_t)oy * Wo + ox) * 256 + c4, plane, o);
                }
        }
        return;
    }
    const int64_t HW = (int64_t)H * W, plane = (int64_t)B * HW * 256;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < HW; p += (int64_t)gridDim.x * 4) {
        const float4 o = gn_relu4(*(const float4*)(yb + p * 256 + c4), sc, sh, act);
        if (mode == PH_GN_TO_PLANES) {
            st_planes4<PA, E>(planes + ((int64_t)b * HW + p) * 256 + c4, plane, o);
        } else if (mode == PH_GN_ACCUM) {
            float4* d = (float4*)(outf + ((int64_t)b * HW + p) * 256 + c4);
            if (accumulate) {
                const float4 t = *d;
                *d = make_float4(t.x + o.x, t.y + o.y, t.z + o.z, t.w + o.w);
            } else {
                *d = o;
            }
        }
    }
}

// fp32 NHWC -> GroupNorm affine + ReLU -> fp32 NCHW through an LDS transpose: a block takes 32 pixels x 256
// channels, reads whole 1 KiB pixel vectors and writes 128-byte runs of 32 pixels per channel row
__global__ __launch_bounds__(256) void k_gn_to_nchw(const float* __restrict__ y, const float* __restrict__ stats,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, int groups,
                                                    float* __restrict__ out, int64_t HW) {
    __shared__ float t[256][33];
    const int b = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c4 = (threadIdx.x & 63) * 4, pq = threadIdx.x >> 6;
    const int cpg = 256 / groups;
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sc[e] = 1.f; sh[e] = 0.f;
        if (stats) {
            const float* st = stats + ((int64_t)b * groups + (c4 + e) / cpg) * 2;
            sc[e] = st[1] * gamma[c4 + e];
            sh[e] = beta[c4 + e] - st[0] * sc[e];
        }
    }
    const bool act = stats != nullptr;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int pl = k * 4 + pq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + pl < HW) {
            const uint4 q = ld_nt16(y + ((int64_t)b * HW + p0 + pl) * 256 + c4);
            v = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
        }
        float o[4] = {v.x * sc[0] + sh[0], v.y * sc[1] + sh[1], v.z * sc[2] + sh[2], v.w * sc[3] + sh[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) t[c4 + e][pl] = act ? fmaxf(o[e], 0.f) : o[e];
    }
    __syncthreads();
    const int pl = threadIdx.x & 31, cr = threadIdx.x >> 5;       // 8 channel rows per pass
    if (p0 + pl < HW) {
#pragma unroll 4
        for (int c = cr; c < 256; c += 8) out[((int64_t)b * 256 + c) * HW + p0 + pl] = t[c][pl];
    }
}

// the same transpose, but to the decode path's feature format: bf16 channel planes [P][B][256][HWp] (hi, or hi/lo in
// split precision), zero in [HW, HWp) -- what KernelHead's kernels take directly (ph_khead_fused, PH_IN_PLANES), so the
// neck's three outputs cross HBM once at 2 bytes instead of 4 and are never converted again.  A block takes 64 pixels x
// 256 channels: whole 1 KiB pixel vectors in, 128-byte runs of 64 pixels per channel row out.
template <int PA, int E = PH_E_BF16>
__global__ __launch_bounds__(256) void k_gn_to_cplanes(const float* __restrict__ y, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int groups, uint16_t* __restrict__ planes, int64_t HW, int64_t HWp,
                                                       int B, int tiles_per_wg) {
    extern __shared__ float tt[];                                 // [256][65]
    const int b = blockIdx.y;
    const int c4 = (threadIdx.x & 63) * 4, pq = threadIdx.x >> 6;
    const int cpg = 256 / groups;
    float sc[4], sh[4];                                           // set up once, used for `tiles_per_wg` tiles
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float* st = stats + ((int64_t)b * groups + (c4 + e) / cpg) * 2;
        sc[e] = st[1] * gamma[c4 + e];
        sh[e] = beta[c4 + e] - st[0] * sc[e];
    }
    const int piece = threadIdx.x & 7, r0 = threadIdx.x >> 3;       // 32 channel rows per pass, 8 x 16 B per row
    const int64_t oplane = (int64_t)B * 256 * HWp;
    const int64_t ntiles = HWp / 64;
    for (int ti = 0; ti < tiles_per_wg; ++ti) {
        const int64_t tile = (int64_t)blockIdx.x * tiles_per_wg + ti;
        if (tile >= ntiles) break;
        const int64_t p0 = tile * 64;
        if (ti) __syncthreads();                                  // the previous tile has been read out of LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int pl = k * 4 + pq;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            if (p0 + pl < HW) {
                const uint4 q = ld_nt16(y + ((int64_t)b * HW + p0 + pl) * 256 + c4);
                o[0] = fmaxf(__uint_as_float(q.x) * sc[0] + sh[0], 0.f);
                o[1] = fmaxf(__uint_as_float(q.y) * sc[1] + sh[1], 0.f);
                o[2] = fmaxf(__uint_as_float(q.z) * sc[2] + sh[2], 0.f);
                o[3] = fmaxf(__uint_as_float(q.w) * sc[3] + sh[3], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tt[(c4 + e) * 65 + pl] = o[e];
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 32 + r0;
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f2e_split<E>(tt[row * 65 + piece * 8 + e], hi[e], lo[e]);
            uint16_t* d = planes + ((int64_t)b * 256 + row) * HWp + p0 + piece * 8;
            *(uint4*)d = make_uint4(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7]));
            if (PA == 2) *(uint4*)(d + oplane) = make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(lo[4], lo[5]), pack2(lo[6], lo[7]));
        }
    }
}

// sum over `nlev` levels of ReLU(GroupNorm(y_l)) -> bf16 NHWC plane(s) of the sum (semantic_fpn.py:221): the fp32
// sum never exists in memory (one pass over the level outputs instead of a read-modify-write per level)
struct GnSumArgs {
    const float* y[4];
    const float* stats[4];
    const float* gamma[4];
    const float* beta[4];
};

template <int PA, int E = PH_E_BF16>
__global__ __launch_bounds__(256) void k_gn_sum_planes(const GnSumArgs a, int nlev, int groups, uint16_t* __restrict__ planes,
                                                       int B, int64_t HW) {
    const int b = blockIdx.z;
    const int c4 = (threadIdx.x & 63) * 4;
    const int cpg = 256 / groups;
    float4 sc[4], sh[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        sc[l] = make_float4(0.f, 0.f, 0.f, 0.f); sh[l] = sc[l];
        if (l >= nlev) continue;
        float s_[4], h_[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* st = a.stats[l] + ((int64_t)b * groups + (c4 + e) / cpg) * 2;
            s_[e] = st[1] * a.gamma[l][c4 + e];
            h_[e] = a.beta[l][c4 + e] - st[0] * s_[e];
        }
        sc[l] = make_float4(s_[0], s_[1], s_[2], s_[3]);
        sh[l] = make_float4(h_[0], h_[1], h_[2], h_[3]);
    }
    const int64_t plane = (int64_t)B * HW * 256;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < HW; p += (int64_t)gridDim.x * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        uint4 q[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)        // all level loads in flight together
            if (l < nlev) q[l] = ld_nt16(a.y[l] + ((int64_t)b * HW + p) * 256 + c4);
#pragma unroll
        for (int l = 0; l < 4; ++l) {      // level order 0, 1, 2, 3 like Python's sum()
            if (l < nlev) {
                const float4 o = gn_relu4(make_float4(__uint_as_float(q[l].x), __uint_as_float(q[l].y), __uint_as_float(q[l].z), __uint_as_float(q[l].w)), sc[l], sh[l], true);
                acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            }
        }
        st_planes4<PA, E>(planes + ((int64_t)b * HW + p) * 256 + c4, plane, acc);
    }
}

// the same level sum, written as the decode path's CHANNEL planes [P][B][256][HWp] (zero in [HW, HWp)) through an LDS transpose
// (k_gn_to_cplanes' tile: 64 pixels x 256 channels, whole 1 KiB pixel vectors in, 128-byte runs of 64 pixels per channel row out):
// the input format of the 1x1 conv + GroupNorm + ReLU kernels of ph_khead.hip, which the three output convs of the neck reuse
// (ph_neck_out_convs: their GroupNorm statistics come from a recompute pass instead of an fp32 round trip of the conv output)
template <int PA, int E = PH_E_BF16>
__global__ __launch_bounds__(256) void k_gn_sum_cplanes(const GnSumArgs a, int nlev, int groups, uint16_t* __restrict__ planes,
                                                        int B, int64_t HW, int64_t HWp, int tiles_per_wg) {
    extern __shared__ float tt[];                                 // [256][65]
    const int b = blockIdx.y;
    const int c4 = (threadIdx.x & 63) * 4, pq = threadIdx.x >> 6;
    const int cpg = 256 / groups;
    float4 sc[4], sh[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        sc[l] = make_float4(0.f, 0.f, 0.f, 0.f); sh[l] = sc[l];
        if (l >= nlev) continue;
        float s_[4], h_[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* st = a.stats[l] + ((int64_t)b * groups + (c4 + e) / cpg) * 2;
            s_[e] = st[1] * a.gamma[l][c4 + e];
            h_[e] = a.beta[l][c4 + e] - st[0] * s_[e];
        }
        sc[l] = make_float4(s_[0], s_[1], s_[2], s_[3]);
        sh[l] = make_float4(h_[0], h_[1], h_[2], h_[3]);
    }
    const int piece = threadIdx.x & 7, r0 = threadIdx.x >> 3;       // 32 channel rows per pass, 8 x 16 B per row
    const int64_t oplane = (int64_t)B * 256 * HWp;
    const int64_t ntiles = HWp / 64;
    for (int ti = 0; ti < tiles_per_wg; ++ti) {
        const int64_t tile = (int64_t)blockIdx.x * tiles_per_wg + ti;
        if (tile >= ntiles) break;
        const int64_t p0 = tile * 64;
        if (ti) __syncthreads();                                  // the previous tile has been read out of LDS
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int pl = k * 4 + pq;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p0 + pl < HW) {
                uint4 q[4];
#pragma unroll
                for (int l = 0; l < 4; ++l)        // all level loads in flight together
                    if (l < nlev) q[l] = ld_nt16(a.y[l] + ((int64_t)b * HW + p0 + pl) * 256 + c4);
#pragma unroll
                for (int l = 0; l < 4; ++l) {      // level order 0, 1, 2, 3 like Python's sum()
                    if (l < nlev) {
                        const float4 o = gn_relu4(make_float4(__uint_as_float(q[l].x), __uint_as_float(q[l].y), __uint_as_float(q[l].z), __uint_as_float(q[l].w)), sc[l], sh[l], true);
                        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                    }
                }
            }
            tt[(c4 + 0) * 65 + pl] = acc.x; tt[(c4 + 1) * 65 + pl] = acc.y; tt[(c4 + 2) * 65 + pl] = acc.z; tt[(c4 + 3) * 65 + pl] = acc.w;
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 32 + r0;
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f2e_split<E>(tt[row * 65 + piece * 8 + e], hi[e], lo[e]);
            uint16_t* d = planes + ((int64_t)b * 256 + row) * HWp + p0 + piece * 8;
            *(uint4*)d = make_uint4(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7]));
            if (PA == 2) *(uint4*)(d + oplane) = make_uint4(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(lo[4], lo[5]), pack2(lo[6], lo[7]));
        }
    }
}

extern "C" int ph_gn_sum_cplanes(const float* const* ys, const float* const* stats, const float* const* gammas,
                                 const float* const* betas, int nlev, int groups, uint16_t* planes, int B, int64_t HW, int prec,
                                 void* stream) {
    PH_CHECK_ARG(ys && stats && gammas && betas && planes && nlev >= 1 && nlev <= 4 && B > 0 && B <= 65535 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG(groups > 0 && 256 % groups == 0, "bad group count");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16, "prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16");
    GnSumArgs a;
    for (int l = 0; l < 4; ++l) {
        const int k = l < nlev ? l : 0;
        a.y[l] = ys[k]; a.stats[l] = stats[k]; a.gamma[l] = gammas[k]; a.beta[l] = betas[k];
    }
    const int64_t HWp = ph_hw_padded(HW), ntiles = HWp / 64;
    int tpw = (int)((ntiles * B + 2047) / 2048);            // ~2048 workgroups: the 4-level affine set-up is amortised over the tiles
    if (tpw < 1) tpw = 1;
    static const int tpw_env = [] { const char* e = getenv("PH_GNSUM_TPW"); return e ? atoi(e) : 0; }();     // tuning knobs, read once
    if (tpw_env > 0) tpw = tpw_env;
    const dim3 grid((unsigned)((ntiles + tpw - 1) / tpw), B);
    const size_t lds = 256 * 65 * sizeof(float);
    static const bool once = [&] {
        (void)hipFuncSetAttribute((const void*)k_gn_sum_cplanes<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute((const void*)k_gn_sum_cplanes<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute((const void*)k_gn_sum_cplanes<1, PH_E_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        return true;
    }();
    (void)once;
    hipStream_t s = (hipStream_t)stream;
    if (prec == PH_PREC_F16) hipLaunchKernelGGL((k_gn_sum_cplanes<1, PH_E_F16>), grid, dim3(256), lds, s, a, nlev, groups, planes, B, HW, HWp, tpw);
    else if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_gn_sum_cplanes<1>, grid, dim3(256), lds, s, a, nlev, groups, planes, B, HW, HWp, tpw);
    else hipLaunchKernelGGL(k_gn_sum_cplanes<2>, grid, dim3(256), lds, s, a, nlev, groups, planes, B, HW, HWp, tpw);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_gn_sum_planes(const float* const* ys, const float* const* stats, const float* const* gammas,
                                const float* const* betas, int nlev, int groups, uint16_t* planes, int B, int64_t HW, int prec,
                                void* stream) {
    PH_CHECK_ARG(ys && stats && gammas && betas && planes && nlev >= 1 && nlev <= 4 && B > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG(groups > 0 && 256 % groups == 0, "bad group count");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16, "prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16");
    GnSumArgs a;
    for (int l = 0; l < 4; ++l) {
        const int k = l < nlev ? l : 0;
        a.y[l] = ys[k]; a.stats[l] = stats[k]; a.gamma[l] = gammas[k]; a.beta[l] = betas[k];
    }
    int gx = 1024;                     // several pixels per workgroup: its per-channel affine set-up (4 levels) is amortised
    static const int gx_env = [] { const char* e = getenv("PH_GNSUM_WGS"); return e ? atoi(e) : 0; }();
    if (gx_env) gx = gx_env;
    if ((HW + 3) / 4 < gx) gx = (int)((HW + 3) / 4);
    if (prec == PH_PREC_F16) hipLaunchKernelGGL((k_gn_sum_planes<1, PH_E_F16>), dim3(gx, 1, B), dim3(256), 0, (hipStream_t)stream, a, nlev, groups, planes, B, HW);
    else if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_gn_sum_planes<1>, dim3(gx, 1, B), dim3(256), 0, (hipStream_t)stream, a, nlev, groups, planes, B, HW);
    else hipLaunchKernelGGL(k_gn_sum_planes<2>, dim3(gx, 1, B), dim3(256), 0, (hipStream_t)stream, a, nlev, groups, planes, B, HW);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_gn_apply(const float* y, const float* stats, const float* gamma, const float* beta, int groups, int mode,
                           int accumulate, uint16_t* planes, float* outf, int B, int H, int W, int prec, void* stream) {
    PH_CHECK_ARG(y && B > 0 && H > 0 && W > 0, "bad pointer or size");
    PH_CHECK_ARG(!stats || (gamma && beta && groups > 0 && 256 % groups == 0), "stats need gamma, beta and a valid group count");
    PH_CHECK_ARG(mode >= PH_GN_TO_PLANES && mode <= PH_GN_TO_CPLANES, "bad mode");
    PH_CHECK_ARG(((mode == PH_GN_TO_PLANES || mode == PH_GN_UP2_PLANES || mode == PH_GN_TO_CPLANES) && planes) ||
                     ((mode == PH_GN_ACCUM || mode == PH_GN_TO_NCHW) && outf),
                 "output pointer missing for this mode");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16, "prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16");
    if (mode == PH_GN_TO_CPLANES) {
        PH_CHECK_ARG(stats && B <= 65535, "PH_GN_TO_CPLANES needs statistics (and B <= 65535)");
        const int64_t HW = (int64_t)H * W, HWp = ph_hw_padded(HW);
        const int64_t ntiles = HWp / 64;
        int tpw = (int)((ntiles * B + 1023) / 1024);          // ~1024 workgroups: the affine set-up is amortised over the tiles
        if (tpw < 1) tpw = 1;
        static const int ctpw_env = [] { const char* e = getenv("PH_CPLANES_TPW"); return e ? atoi(e) : 0; }();
        if (ctpw_env) tpw = ctpw_env;
        const dim3 grid((unsigned)((ntiles + tpw - 1) / tpw), B);
        const size_t lds = 256 * 65 * sizeof(float);
        static const bool once = [&] {
            (void)hipFuncSetAttribute((const void*)k_gn_to_cplanes<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute((const void*)k_gn_to_cplanes<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute((const void*)k_gn_to_cplanes<1, PH_E_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            return true;
        }();
        (void)once;
        if (prec == PH_PREC_F16) hipLaunchKernelGGL((k_gn_to_cplanes<1, PH_E_F16>), grid, dim3(256), lds, (hipStream_t)stream, y, stats, gamma, beta, groups, planes, HW, HWp, B, tpw);
        else if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_gn_to_cplanes<1>, grid, dim3(256), lds, (hipStream_t)stream, y, stats, gamma, beta, groups, planes, HW, HWp, B, tpw);
        else hipLaunchKernelGGL(k_gn_to_cplanes<2>, grid, dim3(256), lds, (hipStream_t)stream, y, stats, gamma, beta, groups, planes, HW, HWp, B, tpw);
        PH_CHECK_LAUNCH();
        return PH_OK;
    }
    if (mode == PH_GN_TO_NCHW) {
        const int64_t HW = (int64_t)H * W;
        hipLaunchKernelGGL(k_gn_to_nchw, dim3((unsigned)((HW + 31) / 32), B), dim3(256), 0, (hipStream_t)stream, y, stats, gamma, beta,
                           groups ? groups : 1, outf, HW);
        PH_CHECK_LAUNCH();
        return PH_OK;
    }
    const int64_t npix = mode == PH_GN_UP2_PLANES ? (int64_t)(H + 1) * (W + 1) : (int64_t)H * W;   // UP2: 2x2 output blocks
    int gx = (int)((npix + 3) / 4 < 2048 ? (npix + 3) / 4 : 2048);
    const dim3 grid(gx, 1, B);
    if (!groups) groups = 1;
    if (prec == PH_PREC_F16) hipLaunchKernelGGL((k_gn_apply<1, PH_E_F16>), grid, dim3(256), 0, (hipStream_t)stream, y, stats, gamma, beta, groups, mode, accumulate, planes, outf, B, H, W);
    else if (prec == PH_PREC_BF16) hipLaunchKernelGGL(k_gn_apply<1>, grid, dim3(256), 0, (hipStream_t)stream, y, stats, gamma, beta, groups, mode, accumulate, planes, outf, B, H, W);
    else hipLaunchKernelGGL(k_gn_apply<2>, grid, dim3(256), 0, (hipStream_t)stream, y, stats, gamma, beta, groups, mode, accumulate, planes, outf, B, H, W);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
