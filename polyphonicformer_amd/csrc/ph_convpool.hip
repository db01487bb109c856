// A13 of a NON-final stage fused with the x half of the NEXT stage's A7 (round 6): `ph_dynconv_poolx`
//
//     bits[n][hw]   = ( sum_c kern[n][c] * x[c][hw] + kbias[n] ) > thr            (kernel_update_head.py:317-329, :236-238)
//     ux[n][c]     += sum_hw bits[n][hw] * x[c][hw]                               (kernel_update_head.py:241, next stage)
//
// from ONE read of the x plane.  The unfused sequence reads it twice -- `k_dynconv` (bits) and then `k_pool` (x and depth_feats, with
// those bits): 50 MB per frame and stage boundary at cfg2 against 33.5 MB here (`k_pool` then pools depth_feats alone).
//
// Geometry: workgroup = (pixel range `split` of `nsplit`, frame): the pooled partial sums are per (frame, range), the layout `k_pool`
// writes ([B][nsplit][Npad][512], columns 0 .. 255 here) and the query kernel sums in fixed order.  2 NRT waves in TWO ROLES, one 64-pixel
// tile apart, over a 4-deep LDS-DMA ring:
//   waves 0 .. NRT-1 ("convolution"): 32 queries x the 64 pixels of tile t -- `k_dynconv`'s MFMA sequence per 32-pixel half (two
//       independent accumulators, 32 MFMAs), ballots -> the two mask words of each query: to HBM, and to LDS for ...
//   waves NRT .. 2 NRT-1 ("pooling"): 32 queries x 256 channels over tile t - 1, which is still in the ring: A = its mask words expanded
//       through the 256-entry byte -> 8 x {0, 1} table `k_pool` uses, B = the tile (8 accumulators, 32 MFMAs); the first four of these
//       waves also issue the ring's DMA (tile t + 2 over tile t - 2).
// ONE barrier per tile (tile t landed + tile t - 1's words complete), a second one in the `mixed16` grade, whose tile is converted
// bf16 -> fp16 in place by all waves first.  The first form of this kernel (every wave: convolution, barrier, pooling of the same tile;
// three barriers) ran 207 us per 24 frames at cfg2 against 130 + 200 - 98 for the kernels it replaces -- no gain; with the roles the
// matrix pipe has the other role's MFMAs while one waits on LDS: 145 us (`mixed16`), 133 us (fp16), +4.9 % on the step
// (profiles/r06/poolx_ab.txt).  Counters: MFMA busy 0.53 (0.64 on the two SIMDs that host three waves), LDS bank conflicts 3 % of cycles.
//
// LDS image of a tile: [256 rows][8 x 16-byte pieces], piece index XOR f(row), f = ((row >> 1) & 1) << 2 | ((row >> 2) & 3): the
// convolution's transposing reads (4 rows x 2 halves per 32 lanes) stay conflict free as with `conv_swz` (rows r, r + 1 share f, rows
// r + 2, r + 3 flip bit 2), and the pooling's 16-byte reads of 16 CONSECUTIVE rows at one piece index hit 16 distinct slots (f runs
// through all 8 values over 16 rows, twice with different row parity) -- with `conv_swz` alone they would be 4-way conflicts.
//
// Registers: ten waves put three on one SIMD -- 168 per wave.  A convolution wave holds its A operand (64) + two accumulators (32) + two
// k-steps of B fragments (16); a pooling wave 8 accumulators (128) + one batch of B fragments (16): the roles fit where one wave doing both
// (64 + 64 + 16 + fragments) had to recompute its addresses every tile.
//
// Arithmetic: the bits are `k_dynconv`'s (same MFMA sequence per tile and pixel half).  The pooling multiplies exact {0, 1} by the tile
// as it lies in LDS: bf16 / fp16 planes as they are; in the `mixed16` grade the tile has been converted to fp16 for the convolution, which
// is exact for every bf16 value inside fp16's normal range (values below 6.1e-5 in magnitude lose bits there: features after GroupNorm +
// ReLU are O(1), the pooled sums are not affected at the 1e-3 contract's scale; tests compare with `k_pool` at 1e-6).
//
// -DCP_ABL_NO_CONV / NO_POOL / NO_BALLOT: timing experiments (wrong results) -- what each phase costs inside the ring.
#include <stdlib.h>

#include <type_traits>

#include "ph_conv_inl.h"

constexpr int CP_NBUF = 4;

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
    static constexpr int NW = 2 * NRT;                                  // NRT convolution waves + NRT pooling waves
    static constexpr int TILEB = 256 * CONV_T * 2;
    static constexpr int NDW = NRT >= 4 ? 4 : 2;                        // pooling waves that issue the tile's 32 LDS-DMA instructions
    static constexpr int LUTB = 256 * 16, BWB = 2 * NRT * 32 * 2 * 4, KBB = NRT * 32 * 4;
    static constexpr int LDSB = CP_NBUF * TILEB + LUTB + BWB + KBB;
};

template <int E, int NRT, bool COOP>
__global__ __launch_bounds__((2 * NRT * 64)) void k_dynconv_poolx(const uint16_t* __restrict__ planes, const uint16_t* __restrict__ kern,
                                                                  int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                                  int64_t kbias_batch_stride, uint32_t* __restrict__ bits_out,
                                                                  float* __restrict__ partial, int B, int N, int64_t HW, int64_t HWp, int nsplit) {
    using C = CpCfg<NRT>;
    constexpr int Npad = NRT * 32, NBUF = CP_NBUF, NDW = C::NDW, DPW = 32 / NDW;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [NBUF][256][64] | lut[256] x 16 B | mask words [2][Npad][2] | biases [Npad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_conv = wave < NRT;                                  // waves 0 .. NRT - 1: convolution of tile t; NRT .. 2 NRT - 1: pooling of tile t - 1
    const int rt = is_conv ? wave : wave - NRT;                       // the wave's 32-query block
    const int g = lane >> 5;
    const int split = blockIdx.x, b = blockIdx.y;

    const int ntiles = (int)(HWp / CONV_T);
    // pixel ranges = k_pool's (whole pairs of 64-pixel chunks, ph_pool.hip): with the same nsplit the partial sums of a range are then
    // k_pool's BIT FOR BIT in the bf16 / fp16 grades (same operands, same k-step order per tile) -- which kernel pooled a frame's x map is
    // invisible in its outputs, so the choice may follow the launch size even in batch-invariant plans
    int t0 = (int)((int64_t)split * ntiles / nsplit) & ~1, t1 = (int)((int64_t)(split + 1) * ntiles / nsplit) & ~1;
    if (split + 1 == nsplit) t1 = ntiles;

    uint4* lut = (uint4*)((unsigned char*)lds + NBUF * C::TILEB);
    uint32_t* bw = (uint32_t*)((unsigned char*)lds + NBUF * C::TILEB + C::LUTB);
    float* kb_lds = (float*)((unsigned char*)lds + NBUF * C::TILEB + C::LUTB + C::BWB);
    for (int i = tid; i < 256; i += C::NW * 64) lut[i] = cp_expand8<E>((uint32_t)i);
    if (is_conv && lane < 32) kb_lds[rt * 32 + lane] = kbias[(int64_t)b * kbias_batch_stride + rt * 32 + lane];
    __syncthreads();                                                  // lut, biases
    if (t0 >= t1) {                                                   // an empty range (more ranges than chunk pairs): its sums are zeros, as k_pool's
        float* out = partial + (((int64_t)b * nsplit + split) * Npad) * 512;
        for (int i = tid; i < Npad * 256; i += C::NW * 64) out[(int64_t)(i >> 8) * 512 + (i & 255)] = 0.f;
        return;
    }
    const uint32_t lds0 = lds_addr(lds);

    // bf16 -> fp16 of a whole tile in place (`mixed16`), every wave of the workgroup: 2048 16-byte pieces, one round at a time
#ifndef CP_CONV_PIECES
#define CP_CONV_PIECES 1024          // of the tile's 2048 16-byte pieces, how many the CONVOLUTION waves convert (the pooling waves: the rest).
                                     // Measured on one box, 32 frames: 1024 (equal) 246-252 us, 640: 249-251, 320: 254, 0: 343 -- equal shares stay
#endif
    auto convert = [&](int buf, auto role_tag) {
        constexpr bool CONVW = decltype(role_tag)::value;
        constexpr int PIECES = 256 * CONV_T * 2 / 16, P0 = CONVW ? 0 : CP_CONV_PIECES, NP = CONVW ? CP_CONV_PIECES : PIECES - CP_CONV_PIECES;
        constexpr int LANES = NRT * 64, ROUNDS = (NP + LANES - 1) / LANES;
        const int rl = (wave - (CONVW ? 0 : NRT)) * 64 + lane;                  // lane index inside the role
        const uint32_t tb = lds0 + buf * C::TILEB + 16u * (uint32_t)(P0 + rl);
        u32x4_t cv[ROUNDS > 0 ? ROUNDS : 1];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if ((r + 1) * LANES <= NP || rl < NP - r * LANES) cv[r] = lds_read128_asm(tb + r * LANES * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if ((r + 1) * LANES <= NP || rl < NP - r * LANES) {
                const uint4 h16 = bf2h_x8(__builtin_bit_cast(uint4, cv[r]));
                lds_write128_asm(tb + r * LANES * 16, __builtin_bit_cast(u32x4_t, h16));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    if (is_conv) {
        // =========================================== convolution waves: 32 queries x the tile's 64 pixels -> two mask words per query
        uint4 af[16];                                                 // the A operand (this wave's 32 kernel rows): ordinary loads, once
        {
            const uint16_t* kr = kern + (int64_t)b * kern_batch_stride + (rt * 32 + (lane & 31)) * PH_C + g * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) af[ks] = *(const uint4*)(kr + ks * 16);
        }
        // B fragments (transposing reads): lane's row (k-step 0) = g * 8 + (i16 >> 2); its f has bit 2 = (i16 >> 3) & 1 and low bits 2 g for
        // that row and every k-step's (+ 16 rows), 2 g + 1 for the rows 4 below: two base addresses per pixel half; the second half's pieces
        // are the first's + 4 = the first's address with bit 6 flipped
        const int i16 = lane & 15, gi = (lane >> 4) & 1;
        const int row0 = g * 8 + (i16 >> 2);
        const int pc = gi * 2 + ((i16 & 3) >> 1);
        const uint32_t fo_a = 2u * (uint32_t)(row0 * CONV_T + ((pc ^ cp_swz(row0)) * 8) + (i16 & 1) * 4);
        const uint32_t fo_b = 2u * (uint32_t)((row0 + 4) * CONV_T + ((pc ^ cp_swz(row0 + 4)) * 8) + (i16 & 1) * 4);
        const uint32_t kb_addr = lds_addr(kb_lds + rt * 32) + 16u * (uint32_t)g;
        const uint32_t bw_w = lds_addr(bw) + 4u * (uint32_t)((rt * 32 + (lane & 31)) * 2 + g);
        const uint32_t words_per_row4 = 4u * (uint32_t)(HWp / 32);
        char* optr = (char*)(bits_out + ((int64_t)b * Npad + rt * 32 + (lane & 31)) * (HWp / 32) + t0 * 2 + g);
        const bool live_row = rt * 32 + (lane & 31) < N;
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): the A operand
        int cur = 0, par = 0;
        for (int t = t0;; ++t) {
            __builtin_amdgcn_s_barrier();                             // (A) tile t has landed; the pooling waves are past tile t - 2
            __builtin_amdgcn_sched_barrier(0);
            if (t == t1) break;
            if constexpr (COOP) {
                convert(cur, std::true_type{});
                __builtin_amdgcn_s_barrier();                         // (B) the tile is fp16
                __builtin_amdgcn_sched_barrier(0);
            }
            float bias[16];
            conv_bias_get(kb_addr, bias);
            const uint32_t tb = lds0 + cur * C::TILEB;
            const uint32_t fa0 = tb + fo_a, fb0 = tb + fo_b, fa1 = tb + (fo_a ^ 64u), fb1 = tb + (fo_b ^ 64u);
            f32x16_t acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = bias[r]; acc1[r] = bias[r]; }
#ifndef CP_ABL_NO_CONV
            u32x2_t q[2][4];                                            // [k-step parity][half 0 rows, rows + 4, half 1 rows, rows + 4]
            auto rd = [&](auto ks_tag, u32x2_t (&d)[4]) {
                constexpr int KS = decltype(ks_tag)::value;
                d[0] = lds_tr16_asm<KS * 2048>(fa0);
                d[1] = lds_tr16_asm<KS * 2048>(fb0);
                d[2] = lds_tr16_asm<KS * 2048>(fa1);
                d[3] = lds_tr16_asm<KS * 2048>(fb1);
            };
            auto step = [&](auto ks_tag) {
                constexpr int KS = decltype(ks_tag)::value;
                if constexpr (KS + 1 < 16) {
                    rd(std::integral_constant<int, KS + 1>{}, q[(KS + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x2_t* qq = q[KS & 1];
                acc0 = mfma32e<E>(af[KS], make_uint4(qq[0].x, qq[0].y, qq[1].x, qq[1].y), acc0);
                acc1 = mfma32e<E>(af[KS], make_uint4(qq[2].x, qq[2].y, qq[3].x, qq[3].y), acc1);
                __builtin_amdgcn_sched_barrier(0);
            };
            rd(std::integral_constant<int, 0>{}, q[0]);
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
            step(std::integral_constant<int, 15>{});
#endif
            // mask words (k_dynconv's ballot; pixels past HW and rows past N cleared): lane l < 32 = row l's word of pixels 0 .. 31, lane
            // 32 + l = its word of pixels 32 .. 63
            int word = 0;
            auto ballots = [&](const f32x16_t& acc, int& wd, auto h_tag) {
                constexpr int H = decltype(h_tag)::value;
                const int64_t left = HW - ((int64_t)t * CONV_T + H * 32);
                const uint32_t pxmask = left >= 32 ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
                unsigned long long m[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) m[r] = __ballot(acc[r] > PH_BIN_THR);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 32 * H;
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(wd) : "s"((uint32_t)m[r] & pxmask), "n"(rr));
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(wd) : "s"((uint32_t)(m[r] >> 32) & pxmask), "n"(rr + 4));
                }
            };
#ifndef CP_ABL_NO_BALLOT
            ballots(acc0, word, std::integral_constant<int, 0>{});
            ballots(acc1, word, std::integral_constant<int, 1>{});
#else
            word = (int)(__float_as_uint(acc0[0] + acc1[3]));
#endif
            if (!live_row) word = 0;
            cp_lds_write32(bw_w + (uint32_t)par * (Npad * 2 * 4), (uint32_t)word);
            *(uint32_t*)optr = (uint32_t)word;
            optr += 2 * 4;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the words are in LDS before barrier (A) of the next tile
            cur = cur + 1 == NBUF ? 0 : cur + 1;
            par ^= 1;
        }
        return;
    }

    // =============================================== pooling waves: the tile's DMA, and 32 queries x 256 channels over the previous tile
    const int pw = wave - NRT;
    // LDS-DMA of a tile: 32 instructions of 1 KiB (8 rows x 128 B), dealt to the first NDW pooling waves.  Instruction j covers rows
    // 8 j .. 8 j + 7; f(row) needs bit 3 of the row = the parity of j, which for this wave's instructions (j = pw + NDW k, NDW even) is
    // the parity of pw
    static_assert(NDW % 2 == 0, "the swizzle's per-instruction part is taken from the wave's parity");
    const int r8 = lane >> 3;
    const int frow = (pw & 1) * 8 + r8;                               // a row with this instruction class's bits 1 .. 3
    const uint32_t dma_lane_off = 2u * (uint32_t)(r8 * HWp + (((lane & 7) ^ cp_swz(frow)) * 8));
    int ti = t0;
    const char* iptr = (const char*)planes + 2 * ((int64_t)b * PH_C * HWp + (int64_t)t0 * CONV_T) + dma_lane_off;
    const int64_t row8 = 2 * (int64_t)8 * HWp;
    auto issue_next = [&](int buf) {
        if (pw < NDW) {
#pragma unroll
            for (int k = 0; k < DPW; ++k) {
                const int j = pw + NDW * k;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(iptr + j * row8),
                                                 (PH_LDS void*)((PH_LDS char*)lds + buf * C::TILEB + j * 1024), 16, 0, PH_CPOL_STREAM);
            }
        }
        ++ti;
        iptr += 2 * CONV_T;
    };
    issue_next(0);
    if (ti < t1) issue_next(1);

    // B fragment of channel block blk, k-step s = 16 bytes of row 32 blk + (lane & 31) at piece (4 g + s) ^ f(row); f does not depend on blk
    const int prow = lane & 31;
    const uint32_t pb_base = lds0 + 2u * (uint32_t)(prow * CONV_T);
    const int pf = cp_swz(prow);
    const uint32_t lut_addr = lds_addr(lut);
    const uint32_t bw_r = lds_addr(bw) + 4u * (uint32_t)((rt * 32 + (lane & 31)) * 2 + g);

    f32x16_t pacc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[k][r] = 0.f;

    int cur = 0, par = 0;
    for (int t = t0; t <= t1; ++t) {
        if (pw < NDW && t < t1) {
            if (t + 1 < t1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                   // (A) tile t has landed; tile t - 1's mask words are complete
        __builtin_amdgcn_sched_barrier(0);
        if (ti < t1) {
            int nb = cur + 2;
            if (nb >= NBUF) nb -= NBUF;
            issue_next(nb);                                             // tile t + 2 over tile t - 2
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (COOP) {
            if (t < t1) {
                convert(cur, std::false_type{});
                __builtin_amdgcn_s_barrier();                           // (B) the tile is fp16
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t > t0) {
            const int prev = cur == 0 ? NBUF - 1 : cur - 1;
            const uint32_t w = cp_lds_read32(bw_r + (uint32_t)(par ^ 1) * (Npad * 2 * 4));
            const uint32_t tb = pb_base + prev * C::TILEB;
            // 4 k-steps of 8 pixels per lane group (group g: pixels 32 g + 8 s ..), 8 channel blocks in two batches of four
            auto pstep = [&](auto s_tag) {
                constexpr int S = decltype(s_tag)::value;
                const u32x4_t a = lds_read128_asm(lut_addr + (((w >> (8 * S)) & 0xFFu) << 4));
                const uint32_t ad = tb + 16u * (uint32_t)((4 * g + S) ^ pf);
                u32x4_t d0 = cp_lds_read128<0 * 4096>(ad), d1 = cp_lds_read128<1 * 4096>(ad);
                u32x4_t d2 = cp_lds_read128<2 * 4096>(ad), d3 = cp_lds_read128<3 * 4096>(ad);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                pacc[0] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d0), pacc[0]);
                pacc[1] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d1), pacc[1]);
                __builtin_amdgcn_sched_barrier(0);
                d0 = cp_lds_read128<4 * 4096>(ad); d1 = cp_lds_read128<5 * 4096>(ad);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                pacc[2] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d2), pacc[2]);
                pacc[3] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d3), pacc[3]);
                __builtin_amdgcn_sched_barrier(0);
                d2 = cp_lds_read128<6 * 4096>(ad); d3 = cp_lds_read128<7 * 4096>(ad);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                pacc[4] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d0), pacc[4]);
                pacc[5] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d1), pacc[5]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                pacc[6] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d2), pacc[6]);
                pacc[7] = mfma32e<E>(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, d3), pacc[7]);
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
        par ^= 1;
    }
    // ---- partial[b][split][row][32 blk + (lane & 31)]  (the x map's columns 0 .. 255)
    float* out = partial + (((int64_t)b * nsplit + split) * Npad) * 512 + (lane & 31);
#pragma unroll
    for (int k = 0; k < 8; ++k)
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
