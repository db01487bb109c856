// A13 of a NON-final stage fused with the x half of the NEXT stage's A7 (round 6): `ph_dynconv_poolx`
//
//     bits[n][hw]   = ( sum_c kern[n][c] * x[c][hw] + kbias[n] ) > thr            (kernel_update_head.py:317-329, :236-238)
//     ux[n][c]     += sum_hw bits[n][hw] * x[c][hw]                               (kernel_update_head.py:241, next stage)
//
// from ONE read of the x plane.  The unfused sequence reads it twice -- `k_dynconv` (bits) and then `k_pool` (x and depth_feats, with
// those bits): 50 MB per frame and stage boundary at cfg2 against 33.5 MB here (`k_pool` then pools depth_feats alone).  A 64-pixel tile
// is in LDS for the convolution anyway; once every wave has balloted its 32 x 32 block of mask bits the same tile is the B operand of the
// pooling product, a wave = one 32-query block x one 128-channel half (four 32 x 32 accumulators), the A operand = the tile's mask words
// expanded through the 256-entry byte -> 8 x {0, 1} table k_pool uses.
//
// Geometry: workgroup = (pixel range `split` of `nsplit`, frame): the pooled partial sums are per (frame, range), the layout `k_pool`
// writes ([B][nsplit][Npad][512], columns 0 .. 255 here) and the query kernel sums in fixed order.  2 NRT waves (32-query block x
// 32-pixel half for the convolution, x 128-channel half for the pooling), the 4-deep LDS-DMA ring of `k_dynconv`.  Three barriers per
// tile: tile landed | (bf16 -> fp16 conversion, `mixed16`) | mask words of the tile complete.
//
// LDS image of a tile: [256 rows][8 x 16-byte pieces], piece index XOR f(row), f = ((row >> 1) & 1) << 2 | ((row >> 2) & 3): the
// convolution's transposing reads (4 rows x 2 halves per 32 lanes) stay conflict free as with `conv_swz` (rows r, r + 1 share f, rows
// r + 2, r + 3 flip bit 2), and the pooling's 16-byte reads of 16 CONSECUTIVE rows at one piece index hit 16 distinct slots (f runs
// through all 8 values over 16 rows, twice with different row parity) -- with `conv_swz` alone they would be 4-way conflicts.
//
// Arithmetic: the bits are `k_dynconv`'s (same MFMA sequence per tile).  The pooling multiplies exact {0, 1} by the tile as it lies in
// LDS: bf16 / fp16 planes as they are; in the `mixed16` grade the tile has been converted to fp16 for the convolution, which is exact
// for every bf16 value inside fp16's normal range (values below 6.1e-5 in magnitude lose bits there: features after GroupNorm + ReLU
// are O(1), the pooled sums are not affected at the 1e-3 contract's scale; tests compare with `k_pool` at 1e-6).
#include <stdlib.h>

#include <type_traits>

#include "ph_conv_inl.h"

constexpr int CP_NBUF = 4;
// waves that issue the tile's 32 LDS-DMA instructions: the largest even divisor of 32 that is <= the wave count
constexpr int conv_dma_waves_cp(int nw) { return nw >= 8 ? 8 : (nw >= 4 ? 4 : 2); }

__device__ __forceinline__ int cp_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ uint32_t cp_lds_read32(uint32_t byte_addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
__device__ __forceinline__ void cp_lds_write32(uint32_t byte_addr, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(byte_addr), "v"(v) : "memory");
}
template <int OFF> __device__ __forceinline__ u32x4_t cp_lds_read128(uint32_t byte_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}

template <int E> __device__ __forceinline__ uint4 cp_expand8(uint32_t byte) {
    constexpr uint32_t ONE = E == PH_E_BF16 ? 0x3F80u : 0x3C00u;      // 1.0 in the element format the pooling product runs in
    uint32_t r[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t lo = (byte >> (2 * p)) & 1u, hi = (byte >> (2 * p + 1)) & 1u;
        r[p] = lo * ONE + hi * (ONE << 16);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

template <int NRT> struct CpCfg {
    static constexpr int NW = 2 * NRT;
    static constexpr int TILEB = 256 * CONV_T * 2;
    static constexpr int NDW = conv_dma_waves_cp(NW);
    static constexpr int LUTB = 256 * 16, BWB = NRT * 32 * 2 * 4, KBB = NW * 32 * 4;
    static constexpr int LDSB = CP_NBUF * TILEB + LUTB + BWB + KBB;
};

template <int E, int NRT, bool COOP>
__global__ __launch_bounds__((2 * NRT * 64)) void k_dynconv_poolx(const uint16_t* __restrict__ planes, const uint16_t* __restrict__ kern,
                                                                  int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                                  int64_t kbias_batch_stride, uint32_t* __restrict__ bits_out,
                                                                  float* __restrict__ partial, int B, int N, int64_t HW, int64_t HWp, int nsplit) {
    using C = CpCfg<NRT>;
    constexpr int Npad = NRT * 32, NBUF = CP_NBUF, NDW = C::NDW, DPW = 32 / NDW;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [NBUF][256][64] | lut[256] x 16 B | mask words [Npad][2] | biases [NW][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave % NRT, half = wave / NRT;                     // 32-query block; 32-pixel half (convolution) = 128-channel half (pooling)
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    const int split = blockIdx.x, b = blockIdx.y;

    const int ntiles = (int)(HWp / CONV_T);
    const int t0 = (int)((int64_t)split * ntiles / nsplit), t1 = (int)((int64_t)(split + 1) * ntiles / nsplit);

    uint4* lut = (uint4*)((unsigned char*)lds + NBUF * C::TILEB);
    uint32_t* bw = (uint32_t*)((unsigned char*)lds + NBUF * C::TILEB + C::LUTB);
    float* kb_lds = (float*)((unsigned char*)lds + NBUF * C::TILEB + C::LUTB + C::BWB) + wave * 32;
    for (int i = tid; i < 256; i += C::NW * 64) lut[i] = cp_expand8<E>((uint32_t)i);

    // ---- LDS-DMA of tile t into ring buffer `buf`: 32 instructions of 1 KiB (8 rows x 128 B), dealt to the first NDW waves.  Instruction
    //      j covers rows 8 j .. 8 j + 7; f(row) needs bit 3 of the row = the parity of j, which for this wave's instructions (j = wave + NDW k,
    //      NDW even) is the parity of the wave
    static_assert(NDW % 2 == 0, "the swizzle's per-instruction part is taken from the wave's parity");
    const int r8 = lane >> 3;
    const int frow = (wave & 1) * 8 + r8;                             // a row with this instruction class's bits 1 .. 3
    const uint32_t dma_lane_off = 2u * (uint32_t)(r8 * HWp + (((lane & 7) ^ cp_swz(frow)) * 8));
    int64_t piece_off[DPW];
    uint32_t piece_lds[DPW];
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
        const int j = wave + NDW * k;
        piece_off[k] = 2 * ((int64_t)(j * 8) * HWp);
        piece_lds[k] = 2u * (uint32_t)(j * 512);
    }
    int ti = t0;
    const char* iptr = (const char*)planes + 2 * ((int64_t)b * PH_C * HWp + (int64_t)t0 * CONV_T);
    auto issue_next = [&](int buf) {
        if (wave < NDW) {
#pragma unroll
            for (int k = 0; k < DPW; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(iptr + piece_off[k] + dma_lane_off),
                                                 (PH_LDS void*)((PH_LDS char*)lds + buf * C::TILEB + piece_lds[k]), 16, 0, PH_CPOL_STREAM);
        }
        ++ti;
        iptr += 2 * CONV_T;
    };

    // ---- the A operand of the convolution (this wave's 32 kernel rows) and the biases: ordinary loads, once (one frame per workgroup)
    uint4 af[16];
    {
        const uint16_t* kr = kern + (int64_t)b * kern_batch_stride + (rt * 32 + (lane & 31)) * PH_C + g * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) af[ks] = *(const uint4*)(kr + ks * 16);
        if (lane < 32) kb_lds[lane] = kbias[(int64_t)b * kbias_batch_stride + rt * 32 + lane];
        __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    }
    __syncthreads();                                    // lut, biases
    if (t0 >= t1) return;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d)
        if (ti < t1) issue_next(d);

    // ---- convolution B fragments (transposing reads): lane's row (k-step 0) = g * 8 + (i16 >> 2); its f has bit 2 = (i16 >> 3) & 1 and low
    //      bits 2 g for that row and every k-step's (+ 16 rows), 2 g + 1 for the rows 4 below: two base addresses
    const int row0 = g * 8 + (i16 >> 2);
    const int pc = half * 4 + gi * 2 + ((i16 & 3) >> 1);                 // piece of the lane's 8 bytes before the swizzle
    const uint32_t lds0 = lds_addr(lds);
    const uint32_t fo_a = 2u * (uint32_t)(row0 * CONV_T + ((pc ^ cp_swz(row0)) * 8) + (i16 & 1) * 4);
    const uint32_t fo_b = 2u * (uint32_t)((row0 + 4) * CONV_T + ((pc ^ cp_swz(row0 + 4)) * 8) + (i16 & 1) * 4);
    // ---- pooling: B fragment of channel block blk, k-step s = 16 bytes of row 128 half + 32 blk + (lane & 31) at piece (4 g + s) ^ f(row);
    //      f's low bits depend on blk only through bit 3 of the row (blk * 32 keeps bits 1 .. 3): one lane constant
    const int prow = 128 * half + (lane & 31);
    const uint32_t pb_base = lds0 + 2u * (uint32_t)(prow * CONV_T);
    const int pf = cp_swz(prow);
    const uint32_t lut_addr = lds_addr(lut);
    const uint32_t bw_addr = lds_addr(bw);

    f32x16_t pacc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[k][r] = 0.f;

    int pend = 0;
    uint32_t pend_word = 0;
    char* pend_ptr = nullptr;
    const uint32_t kb_base = lds_addr(kb_lds);                                                // (wave uniform)
    const uint32_t words_per_row4 = 4u * (uint32_t)(HWp / 32);
    char* optr = (char*)(bits_out + ((int64_t)b * Npad + rt * 32) * (HWp / 32) + t0 * 2 + half);
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        // lane-derived addresses are RECOMPUTED per tile from a laundered copy of the lane id: held across the loop they cost the
        // registers that decide between 168 and a spilled A operand (reloaded from scratch every tile)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t kb_addr = kb_base + 16u * (uint32_t)(ln >> 5);
        const uint32_t bwl = bw_addr + 8u * (uint32_t)(rt * 32 + (ln & 31));
        const uint32_t bw_w = bwl + 4u * (uint32_t)half, bw_r = bwl + 4u * (uint32_t)(ln >> 5);   // this wave's word | the word of the lane's pixel half
        const uint32_t bits_lane_off = (uint32_t)(ln & 31) * words_per_row4;
        float bias[16];
        conv_bias_get(kb_addr, bias);
        {
            const int younger = (t1 - 1 - t) < (NBUF - 2) ? (t1 - 1 - t) : (NBUF - 2);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                   // (A) tile t has landed; every wave is past the pooling of t - 1
        __builtin_amdgcn_sched_barrier(0);
        if (pend && lane < 32) *(uint32_t*)(pend_ptr + bits_lane_off) = pend_word;      // the previous tile's mask words, one iteration late
        pend = 0;
        if (ti < t1) {
            int nb = cur + NBUF - 1;
            if (nb >= NBUF) nb -= NBUF;
            issue_next(nb);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef CP_ABL_NO_COOP
        if constexpr (COOP) {
            constexpr int PIECES = 256 * CONV_T * 2 / 16, LANES = C::NW * 64, ROUNDS = (PIECES + LANES - 1) / LANES;
            const uint32_t tb = lds0 + cur * C::TILEB + 16u * (uint32_t)tid;
            // one round of 16-byte pieces at a time (4 registers in flight: the register file is the limit here, see the pooling below;
            // the other waves of the SIMD cover the round trips)
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                if ((r + 1) * LANES <= PIECES || wave * 64 < PIECES - r * LANES) {
                    const u32x4_t cv = lds_read128_asm(tb + r * LANES * 16);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const uint4 h16 = bf2h_x8(__builtin_bit_cast(uint4, cv));
                    lds_write128_asm(tb + r * LANES * 16, __builtin_bit_cast(u32x4_t, h16));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // (B) the tile is fp16
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        // ---- convolution of this wave's 32 queries x 32 pixels: 16 k-steps, two per batch, one batch of reads ahead
        {
            const uint32_t fa = lds0 + cur * C::TILEB + fo_a, fb = lds0 + cur * C::TILEB + fo_b;
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias[r];
            u32x2_t q[2][2];                                            // [k-step parity][rows / rows + 4]: one k-step of reads ahead
            auto rd = [&](auto ks_tag, u32x2_t (&d)[2]) {
                constexpr int KS = decltype(ks_tag)::value;
                d[0] = lds_tr16_asm<KS * 2048>(fa);
                d[1] = lds_tr16_asm<KS * 2048>(fb);
            };
            auto step = [&](auto ks_tag) {
                constexpr int KS = decltype(ks_tag)::value;
                if constexpr (KS + 1 < 16) {
                    rd(std::integral_constant<int, KS + 1>{}, q[(KS + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma32e<E>(af[KS], make_uint4(q[KS & 1][0].x, q[KS & 1][0].y, q[KS & 1][1].x, q[KS & 1][1].y), acc);
                __builtin_amdgcn_sched_barrier(0);
            };
#ifndef CP_ABL_NO_CONV
            rd(std::integral_constant<int, 0>{}, q[0]);
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
            step(std::integral_constant<int, 15>{});
#endif
            // mask word of this wave's 32 rows for its 32 pixels (k_dynconv's ballot; pixels past HW and rows past N cleared)
            const int px_base = t * CONV_T + half * 32;
            const int64_t left = HW - px_base;
            const uint32_t pxmask = left >= 32 ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
            unsigned long long m[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = __ballot(acc[r] > PH_BIN_THR);
            __builtin_amdgcn_sched_barrier(0);
            int word = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)m[r] & pxmask), "n"(rr));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)(m[r] >> 32) & pxmask), "n"(rr + 4));
            }
            if (rt * 32 + lane >= N) word = 0;
            if (lane < 32) cp_lds_write32(bw_w, (uint32_t)word);
            pend_word = (uint32_t)word; pend = 1; pend_ptr = optr;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // (C) the tile's mask words are complete
        __builtin_amdgcn_sched_barrier(0);
        // ---- pooling: 4 k-steps of 8 pixels (this lane's group: pixels 32 g + 8 s ..), 4 channel blocks
        {
            const uint32_t w = cp_lds_read32(bw_r);
            const uint32_t tb = pb_base + cur * C::TILEB;
            // (one k-step of B fragments at a time: the register file holds the convolution's A operand (64) and the pooled sums (64) of
            // ten waves, three of them on one SIMD -- 168 registers per wave; the other waves of the SIMD cover the reads' latency)
            auto pstep = [&](auto s_tag) {
                constexpr int S = decltype(s_tag)::value;
                const u32x4_t a = lds_read128_asm(lut_addr + (((w >> (8 * S)) & 0xFFu) << 4));
                const uint32_t ad = tb + 16u * (uint32_t)((4 * g + S) ^ pf);
                u32x4_t d0 = cp_lds_read128<0 * 32 * 128>(ad), d1 = cp_lds_read128<1 * 32 * 128>(ad);
                u32x4_t d2 = cp_lds_read128<2 * 32 * 128>(ad), d3 = cp_lds_read128<3 * 32 * 128>(ad);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                pacc[0] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d0), pacc[0]);
                pacc[1] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d1), pacc[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                pacc[2] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d2), pacc[2]);
                pacc[3] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d3), pacc[3]);
                __builtin_amdgcn_sched_barrier(0);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the word
            __builtin_amdgcn_sched_barrier(0);
#ifndef CP_ABL_NO_POOL
            pstep(std::integral_constant<int, 0>{}); pstep(std::integral_constant<int, 1>{});
            pstep(std::integral_constant<int, 2>{}); pstep(std::integral_constant<int, 3>{});
#endif
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        optr += 2 * 4;
    }
    if (pend && lane < 32) *(uint32_t*)(pend_ptr + (uint32_t)(lane & 31) * words_per_row4) = pend_word;
    // ---- partial[b][split][row][128 half + 32 blk + (lane & 31)]  (the x map's columns 0 .. 255)
    float* out = partial + (((int64_t)b * nsplit + split) * Npad) * 512 + 128 * half + (lane & 31);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            out[(int64_t)row * 512 + 32 * k] = pacc[k][r];
        }
}

template <int E, int NRT, bool COOP>
static void launch_cp(const uint16_t* planes, const uint16_t* kern, int64_t kbs, const float* kbias, int64_t bbs, uint32_t* bits_out,
                      float* partial, int B, int N, int64_t HW, int nsplit, hipStream_t s) {
    constexpr int lds = CpCfg<NRT>::LDSB;
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)k_dynconv_poolx<E, NRT, COOP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((k_dynconv_poolx<E, NRT, COOP>), dim3(nsplit, B), dim3(2 * NRT * 64), lds, s, planes, kern, kbs, kbias, bbs, bits_out,
                       partial, B, N, HW, ph_hw_padded(HW), nsplit);
}

// 1 when ph_dynconv_poolx exists for this query count / arithmetic (one-plane grades, 33 .. 192 queries)
extern "C" int ph_dynconv_poolx_supported(int N, int prec) {
    if (N <= 32 || N > 192) return 0;
    return prec == PH_PREC_BF16 || prec == PH_PREC_F16 || prec == PH_PREC_BF16_KF16;
}

extern "C" int ph_dynconv_poolx(const uint16_t* planes, const uint16_t* kern, int64_t kern_batch_stride, const float* kbias,
                                int64_t kbias_batch_stride, uint32_t* bits_out, float* partial, int nsplit, int B, int N, int64_t HW,
                                int prec, void* stream) {
    PH_CHECK_ARG(planes && kern && kbias && bits_out && partial && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    if (!ph_dynconv_poolx_supported(N, prec)) {
        ph_set_error("ph_dynconv_poolx: unsupported query count or arithmetic (use ph_dynconv + ph_pool)");
        return PH_EUNSUPPORTED;
    }
    PH_CHECK_ARG(nsplit >= 1 && nsplit <= ph_hw_padded(HW) / CONV_T && B <= 65535, "nsplit out of range");
    const int nrt = ph_n_padded(N) / 32;
    hipStream_t s = (hipStream_t)stream;
#define CP_ARGS planes, kern, kern_batch_stride, kbias, kbias_batch_stride, bits_out, partial, B, N, HW, nsplit, s
#define CP_CASE(R)                                                                              \
    case R:                                                                                     \
        if (prec == PH_PREC_BF16) launch_cp<PH_E_BF16, R, false>(CP_ARGS);                      \
        else if (prec == PH_PREC_F16) launch_cp<PH_E_F16, R, false>(CP_ARGS);                   \
        else launch_cp<PH_E_F16_FROM_BF16, R, true>(CP_ARGS);                                   \
        break;
    switch (nrt) {
        CP_CASE(2) CP_CASE(3) CP_CASE(4) CP_CASE(5) CP_CASE(6)
        default: ph_set_error("ph_dynconv_poolx: unsupported N"); return PH_EUNSUPPORTED;
    }
#undef CP_CASE
#undef CP_ARGS
    PH_CHECK_LAUNCH();
    return PH_OK;
}
