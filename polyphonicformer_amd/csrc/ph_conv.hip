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

#include "ph_conv_inl.h"

constexpr int conv_dma_waves(int nw, int pieces) {
    int d = nw < pieces ? nw : pieces;
    while (pieces % d) --d;
    return d;
}

// Compile-time geometry of one instantiation.  A workgroup has 2 waves per 32-row block of queries (one per
// 32-pixel half of the tile) and owns the whole LDS of its CU: a ring of NBUF tile buffers, NBUF-1 tiles of DMA
// in flight while one is consumed (96 KiB at cfg2), plus -- logits output, single-plane precision -- a per-wave
// transposition patch.
template <int PF, int PK, int NRT, bool BITS, typename OutT, bool F2 = false> struct ConvCfg {
    // 32-pixel halves per wave: one (two waves per row block) while that keeps <= 3 waves per SIMD (168 VGPRs for
    // the 64-VGPR A operand + fragments); two for wide N.  Two kernel planes over one feature plane (A operand = 128
    // VGPRs, the `mixed` mode): a wave that walks both halves is a serial chain of 2 x 16 k-steps with nothing else on its
    // SIMD (113 us per 24 frames whether it has 2, 3 or 4 row blocks); one half per wave fits 2 waves per SIMD up to 4
    // row blocks (86 / 95 / 103 us) and, for the bits output, 3 waves per SIMD at 5-6 row blocks (131 -> 120 us at cfg2;
    // the logits variants would spill 40+ VGPRs there and stay on two halves per wave)
    static constexpr int HPW = (PK == 1 && NRT <= 6) || (!F2 && PF == 1 && PK == 2 && (NRT <= 4 || (BITS && NRT <= 6))) ? 1 : 2;
    static constexpr int NW = 2 * NRT / HPW;
    static constexpr int TILE = 256 * CONV_T;                                  // elements per plane per buffer
    static constexpr int TILEB = PF * TILE * 2;                                // bytes per ring stage
    static constexpr int PATCH_LD = 32 + 16 / (int)sizeof(OutT);               // elements: 32 px + 16 B padding
    static constexpr bool PATCH = !BITS && PF == 1;
    static constexpr int PATCHB = PATCH ? NW * 32 * PATCH_LD * (int)sizeof(OutT) : 0;
    static constexpr int LDS_MAX = 160 * 1024;
    static constexpr int NDW = conv_dma_waves(NW, 32 * PF);                    // waves that issue DMA
    static constexpr int DPW = 32 * PF / NDW;                                  // ... instructions each per tile
    static constexpr int KBB = NW * 32 * 4;                                    // per-wave copy of its 32 biases
#ifndef CONV_NBUF_MAX        // A/B timing: -DCONV_NBUF_MAX=2 (+ PH_CONV_WGS=512): two half-CU workgroups per CU instead of one that owns the CU
#define CONV_NBUF_MAX 4
#endif
    static constexpr int NBUF_FIT = 4 * TILEB + PATCHB + KBB <= LDS_MAX ? 4 : (3 * TILEB + PATCHB + KBB <= LDS_MAX ? 3 : 2);
    static constexpr int NBUF = NBUF_FIT < CONV_NBUF_MAX ? NBUF_FIT : CONV_NBUF_MAX;
    static constexpr int LDSB = NBUF * TILEB + PATCHB + KBB;
};

// COOP (E = PH_E_F16_FROM_BF16 only): the landed bf16 tile is converted to fp16 ONCE, in place in LDS, every wave taking a
// share of its 16-byte pieces, instead of by each of the NRT row-block waves on its own fragments (NRT x the VALU work);
// costs one more barrier and one LDS round trip of the tile per tile.
template <int PF, int PK, int E, int NRT, bool BITS, typename OutT, bool F2 = false, bool COOP = false>
__global__ __launch_bounds__((ConvCfg<PF, PK, NRT, BITS, OutT, F2>::NW * 64)) void k_dynconv(const uint16_t* __restrict__ planes,
                                                          const uint16_t* __restrict__ kern, int64_t kern_plane_stride,
                                                          int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                          int64_t kbias_batch_stride, uint32_t* __restrict__ bits_out,
                                                          OutT* __restrict__ logits_out, int64_t out_batch_stride, int B,
                                                          int N, int64_t HW, int64_t HWp) {
    using C = ConvCfg<PF, PK, NRT, BITS, OutT, F2>;
    constexpr int Npad = NRT * 32, TILE = C::TILE, NBUF = C::NBUF;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [NBUF][PF][256][64] | patch[NW][32][PATCH_LD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave % NRT, half0 = (wave / NRT) * C::HPW;       // 32-row block, first 32-pixel half of the tile
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    const int64_t fplane = (int64_t)B * PH_C * HWp;

    // this workgroup's share of the B * ntiles tiles: one contiguous range (may span frames).  Measured
    // alternative: teams of workgroups interleaving the tiles of one frame (adjacent 128-byte pieces of a channel
    // row read at the same time, whole DRAM pages) -- 7 % slower, the extra A-operand reloads cost more than the
    // page locality gives.
    const int ntiles = (int)(HWp / CONV_T);
    const int64_t total64 = (int64_t)B * ntiles;
    const int tg0 = (int)(total64 * blockIdx.x / gridDim.x), tg1 = (int)(total64 * (blockIdx.x + 1) / gridDim.x);
    if (tg0 >= tg1) return;

    // LDS-DMA of tile (b, t) into ring buffer `buf`: 32 wave-instructions of 1 KiB (8 rows x 128 B) per plane.
    // Address = wave-uniform base (SGPRs) + ONE per-lane byte offset: row-in-piece * HWp + swizzled 16-byte piece
    // (the swizzle needs bit 1 of the row, which is bit 1 of lane >> 3 whatever the piece).
    const uint32_t dma_lane_off = 2u * (uint32_t)((lane >> 3) * HWp + (((lane & 7) ^ conv_swz(lane >> 3)) * 8));
    // Instruction slots are the scarce resource of this loop (a wave issues at most one instruction every 4
    // cycles, three waves share a SIMD): tile addresses are running wave-uniform pointers (+128 B per tile, one
    // jump per frame) plus per-piece constants computed once.
    int64_t piece_off[C::DPW];                                   // bytes from the tile's first row to this wave's pieces
    uint32_t piece_lds[C::DPW];                                  // ... and inside a ring buffer
#pragma unroll
    for (int k = 0; k < C::DPW; ++k) {
        const int j = wave + C::NDW * k;
        const int p = j >> 5, jj = j & 31;
        piece_off[k] = 2 * (p * fplane + (int64_t)(jj * 8) * HWp);
        piece_lds[k] = 2u * (uint32_t)(p * TILE + jj * 512);
    }
    const int64_t frame_jump = 2 * ((int64_t)PH_C * HWp - (int64_t)ntiles * CONV_T);   // bytes, after the last tile's +128
    int it = tg0 % ntiles, ti = tg0;                                                    // next tile to request
    const char* iptr = (const char*)planes + 2 * ((int64_t)(tg0 / ntiles) * PH_C * HWp + (int64_t)it * CONV_T);
    auto issue_next = [&](int buf) {
        if (wave < C::NDW) {
#pragma unroll
            for (int k = 0; k < C::DPW; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(iptr + piece_off[k] + dma_lane_off),
                                                 (PH_LDS void*)((PH_LDS char*)lds + buf * C::TILEB + piece_lds[k]), 16, 0, PH_CPOL_STREAM);
        }
        ++ti;
        iptr += 2 * CONV_T;
        if (++it == ntiles) { it = 0; iptr += frame_jump; }
    };
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d)
        if (ti < tg1) issue_next(d);

    // B-fragment read address of this lane inside a tile (bytes): row (ks = 0) = g*8 + (i16 >> 2); the swizzle
    // depends on bit 1 of the row = bit 3 of i16 only, so every k-step is an immediate offset of ks * 2 KiB
    const int row0 = g * 8 + (i16 >> 2);
    uint32_t frag_off[C::HPW];
#pragma unroll
    for (int h = 0; h < C::HPW; ++h)
        frag_off[h] = 2u * (uint32_t)(row0 * CONV_T + ((((half0 + h) * 4 + gi * 2) ^ conv_swz(row0)) * 8) + (i16 & 3) * 4);
    const uint32_t lds0 = lds_addr(lds);
    const uint32_t patch_addr = lds0 + NBUF * C::TILEB + wave * (32 * C::PATCH_LD * (int)sizeof(OutT));
    // this wave's 32 biases live in LDS (a select between two wave-uniform array elements is turned into an
    // indexed scratch access by the optimiser, and scratch traffic shares vmcnt with the DMA ring)
    float* kb_lds = (float*)((unsigned char*)lds + NBUF * C::TILEB + C::PATCHB) + wave * 32;
    const uint32_t kb_addr = lds_addr(kb_lds) + 16 * g;

    uint4 af[PK][16];      // A operand: this wave's 32 kernel rows, all 16 k-steps

    // results of the previous tile, written out one iteration late (after the next barrier) so that the
    // vector-memory queue in front of each counted wait is [stores(t-1), DMA(t+1) .. DMA(t+NBUF-1)]
    int pend = 0, pend_h = 0;                             // 0 none, 1 bits word(s), 2 patch of half half0 + pend_h
    char* pend_ptr = nullptr;                             // wave-uniform output address of the pending tile
    uint32_t pend_word[C::HPW] = {};
    constexpr int PER16 = 16 / (int)sizeof(OutT);                // elements per 16-byte store
    constexpr int LPR = 32 / PER16;                              // lanes per row: 4 (bf16) / 8 (fp32)
    constexpr int RPI = 64 / LPR;                                // rows per store instruction
    // per-lane parts of the output addresses (bytes), one VGPR each; everything else is wave-uniform
    const uint32_t bits_lane_off = 4u * (uint32_t)((lane & 31) * (HWp / 32));
    const uint32_t patch_lane_off = (uint32_t)sizeof(OutT) * (uint32_t)((lane / LPR) * HW + (lane % LPR) * PER16);
    const uint32_t slow_lane_off = (uint32_t)sizeof(OutT) * (uint32_t)((4 * g) * HW + (lane & 31));
    const int64_t rpi_bytes = (int64_t)RPI * HW * (int)sizeof(OutT);
    auto flush = [&]() {
        if (BITS) {
            if (pend == 1 && lane < 32) {
                uint32_t* w = (uint32_t*)(pend_ptr + bits_lane_off);
#pragma unroll
                for (int h = 0; h < C::HPW; ++h) w[h] = pend_word[h];
            }
        } else if (C::PATCH) {
            if (pend == 2) {
                u32x4_t v[32 / RPI];
#pragma unroll
                for (int k = 0; k < 32 / RPI; ++k)
                    v[k] = lds_read128_asm(patch_addr + ((k * RPI + lane / LPR) * C::PATCH_LD + (lane % LPR) * PER16) * (int)sizeof(OutT));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                char* ub = pend_ptr + pend_h * (32 * (int)sizeof(OutT));
#pragma unroll
                for (int k = 0; k < 32 / RPI; ++k) {
                    const int row = rt * 32 + k * RPI + lane / LPR;
                    // default (cached) stores: the x2 upsample reads these logits next (non-temporal: 117 -> 150 us)
                    if (row < N) *(u32x4_t*)(ub + k * rpi_bytes + patch_lane_off) = v[k];
                }
            }
        }
        pend = 0;
    };

    // output address of the tile being computed (wave-uniform, running): mask words / logits of rows rt*32.., half0
    const int64_t out_step = BITS ? 2 * 4 : CONV_T * (int)sizeof(OutT);
    const int64_t out_jump = BITS ? 4 * ((int64_t)Npad * (HWp / 32) - (int64_t)ntiles * 2)
                                  : (int64_t)sizeof(OutT) * (out_batch_stride - (int64_t)ntiles * CONV_T);
    char* optr;
    {
        const int b0 = tg0 / ntiles, t0 = tg0 - b0 * ntiles;
        if (BITS) optr = (char*)(bits_out + ((int64_t)b0 * Npad + rt * 32) * (HWp / 32) + t0 * 2 + half0);
        else optr = (char*)(logits_out + (int64_t)b0 * out_batch_stride + (int64_t)(rt * 32) * HW + (int64_t)t0 * CONV_T + half0 * 32);
    }
    int b = tg0 / ntiles, t = tg0 - b * ntiles, cur = 0;
    for (int tg = tg0; tg < tg1; ++b, t = 0) {      // one pass per frame this workgroup's range touches
    int seg_end = tg + (ntiles - t);
    if (seg_end > tg1) seg_end = tg1;
    // The A operand of the frame: ordinary VGPR loads OUTSIDE the tile loop (inside it they would be a
    // loop-carried phi and the compiler would wait vmcnt(0) on them every tile).  The compiler-visible
    // s_waitcnt retires them in its scoreboard; it also drains the ring, once per frame.
#pragma unroll
    for (int p = 0; p < PK; ++p) {
        const uint16_t* kr = kern + p * kern_plane_stride + (int64_t)b * kern_batch_stride + (rt * 32 + (lane & 31)) * PH_C + g * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) af[p][ks] = *(const uint4*)(kr + ks * 16);
    }
    if (lane < 32) kb_lds[lane] = kbias[(int64_t)b * kbias_batch_stride + rt * 32 + lane];
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    for (; tg < seg_end; ++tg, ++t) {
        float bias[16];
        conv_bias_get(kb_addr, bias);   // its LDS round trip overlaps the wait for the tile
        {
            // tile tg must have landed; up to NBUF-2 younger tiles stay in flight across the barrier
            const int younger = (tg1 - 1 - tg) < (NBUF - 2) ? (tg1 - 1 - tg) : (NBUF - 2);
            // (vmcnt holds 6 bits: a smaller count than needed only waits for more than needed)
            if (NBUF >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::DPW < 63 ? 2 * C::DPW : 63) : "memory");
            else if (NBUF >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::DPW < 63 ? C::DPW : 63) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        flush();
        // every wave is past compute(tg-1): its buffer is free for tile tg + NBUF-1
        if (ti < tg1) {
            int nb = cur + NBUF - 1;
            if (nb >= NBUF) nb -= NBUF;
            issue_next(nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (COOP) {
            static_assert(PF == 1 && E == PH_E_F16_FROM_BF16, "cooperative conversion: one bf16 feature plane");
            constexpr int PIECES = 256 * CONV_T * 2 / 16, LANES = C::NW * 64, ROUNDS = (PIECES + LANES - 1) / LANES;
            const uint32_t tb = lds0 + cur * C::TILEB + 16u * (uint32_t)tid;
            u32x4_t cv[ROUNDS];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r)
                if ((r + 1) * LANES <= PIECES || wave * 64 < PIECES - r * LANES) cv[r] = lds_read128_asm(tb + r * LANES * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r)
                if ((r + 1) * LANES <= PIECES || wave * 64 < PIECES - r * LANES) {
                    const uint4 h16 = bf2h_x8(__builtin_bit_cast(uint4, cv[r]));
                    lds_write128_asm(tb + r * LANES * 16, __builtin_bit_cast(u32x4_t, h16));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef CONV_ABL_NO_COOP_BARRIER          // timing only (wrong results): what the conversion's barrier costs per tile
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
        }

#pragma unroll
        for (int h = 0; h < C::HPW; ++h) {
        // ---- MFMA phase: 16 k-steps, B fragments by transposing reads, KB k-steps per batch, one batch ahead
        const uint32_t fa = lds0 + cur * C::TILEB + frag_off[h];
        const int half = half0 + h;
        f32x16_t acc;
        constexpr int KB = 2;                        // 4 * PF reads per batch (KB = 4 measured 7 % slower)
        u32x2_t bq[2][PF][KB][2];
        conv_read_batch<PF, KB, 0>(fa, bq[0]);
        conv_batches<PF, PK, E, KB, 0, COOP>(fa, af, bq, acc, bias);
        // ---- epilogue of tile (b, t), 32-pixel half `half`
        const int px_base = t * CONV_T + half * 32;
        const int64_t px = (int64_t)px_base + (lane & 31);
        if (BITS) {
            // sign bits of the 32 rows: ballot -> SGPR pair (lanes 0..31 voted for row rr, lanes 32..63 for the row
            // 4 below) -> lane rr / rr + 4 of `word` by v_writelane; pixels past HW and rows past N are cleared
            const int64_t left = HW - px_base;
            const uint32_t pxmask = left >= 32 ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
            int word = 0;
            if (pxmask == 0xFFFFFFFFu) {
                // all 16 compares first: a v_writelane that reads the SGPR a v_cmp wrote in the previous slot gets
                // stale data (observed; the documented hazard names only the lane-select operand)
                unsigned long long m[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) m[r] = __ballot(acc[r] > PH_BIN_THR);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)m[r]), "n"(rr));
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)(m[r] >> 32)), "n"(rr + 4));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    const unsigned long long m = __ballot(acc[r] > PH_BIN_THR);
                    const uint32_t lo = (uint32_t)m & pxmask, hi = (uint32_t)(m >> 32) & pxmask;
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(lo), "n"(rr));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(hi), "n"(rr + 4));
                }
            }
            if (rt * 32 + lane >= N) word = 0;
            pend_word[h] = (uint32_t)word; pend = 1; pend_ptr = optr;
        } else {
            const bool fast = C::PATCH && (HW % PER16 == 0) && ((int64_t)(t + 1) * CONV_T <= HW);
            if (fast) {
                // transpose the [32 rows][32 px] result through the per-wave LDS patch so that every lane stores
                // 16 contiguous bytes of one row
                if (C::HPW > 1) flush();   // the one patch still holds the previous half: write it out first
                const uint32_t wa = patch_addr + ((4 * g) * C::PATCH_LD + (lane & 31)) * (int)sizeof(OutT);
                // MFMA result -> LDS write hazard: the compiler pads it with wait states for instructions it knows,
                // not for inline asm (fp32 output feeds the accumulator registers to ds_write directly)
                asm volatile("s_nop 15\n s_nop 7" ::: "memory");
                conv_patch_put<OutT, C::PATCH_LD>(wa, acc);
                pend = 2; pend_ptr = optr; pend_h = h;
            } else {
                const OutT* ub = (const OutT*)optr + h * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    const int row = rt * 32 + rr + 4 * g;
                    if (row < N && px < HW) st_out((OutT*)((char*)(ub + (int64_t)rr * HW) + slow_lane_off), acc[r]);
                }
            }
        }
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        optr += out_step;
    }
    if (t == ntiles) optr += out_jump;   // next frame
    }
    flush();
}

static int conv_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

template <int PF, int PK, int E, int NRT>
static int launch_conv(const uint16_t* planes, const uint16_t* kern, int64_t kps, int64_t kbs, const float* kbias,
                       int64_t bbs, uint32_t* bits_out, void* logits_out, int out_dtype, int64_t obs, int B, int N,
                       int64_t HW, hipStream_t s) {
    const int64_t HWp = ph_hw_padded(HW);
    const int64_t total = (int64_t)B * (HWp / CONV_T);
    // one persistent workgroup per CU (it owns the CU's LDS), tiles split evenly: no tail generation
    int wgs = conv_num_cus();
    static const int wgs_env = [] { const char* e = getenv("PH_CONV_WGS"); return e ? atoi(e) : 0; }();   // tuning knob, read once
    if (wgs_env) wgs = wgs_env;
    if (wgs > total) wgs = (int)total;
    if (wgs < 1) wgs = 1;
    const dim3 grid(wgs);
    // tuning knob: PH_CONV_TWO_HALVES=1 forces the two-halves-per-wave form of the two-kernel-plane kernels (A/B measurements)
    static const bool two_halves = [] { const char* e = getenv("PH_CONV_TWO_HALVES"); return e && atoi(e) != 0; }();
    // the mixed16 conv converts its bf16 tile to fp16 once per tile in LDS (see k_dynconv); PH_CONV_COOP=0 restores the
    // per-wave conversion in registers for A/B measurements (same box, 24 frames: bits 155 -> 130-136 us, logits 157-162 ->
    // 148-151 us, the 96-frame step 13.16 k -> 13.53 k frames/s; results identical, the conversion is exact either way)
    static const bool coop = [] { const char* e = getenv("PH_CONV_COOP"); return !(e && atoi(e) == 0); }();
    (void)coop;
#define PH_CONV_LAUNCH__(BITS, T, F2, CO)                                                                            \
    do {                                                                                                             \
        constexpr int lds = ConvCfg<PF, PK, NRT, BITS, T, F2>::LDSB;                                                 \
        const dim3 block(ConvCfg<PF, PK, NRT, BITS, T, F2>::NW * 64);                                                \
        static const bool once = [&] {                                                                               \
            (void)hipFuncSetAttribute((const void*)k_dynconv<PF, PK, E, NRT, BITS, T, F2, CO>,                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);                              \
            return true;                                                                                             \
        }();                                                                                                         \
        (void)once;                                                                                                  \
        hipLaunchKernelGGL((k_dynconv<PF, PK, E, NRT, BITS, T, F2, CO>), grid, block, lds, s, planes, kern, kps, kbs, kbias, bbs, \
                           bits_out, (T*)logits_out, obs, B, N, HW, HWp);                                            \
    } while (0)
#define PH_CONV_LAUNCH_(BITS, T, F2)                                                                                 \
    do {                                                                                                             \
        if constexpr (E == PH_E_F16_FROM_BF16 && PF == 1) {                                                          \
            if (coop) PH_CONV_LAUNCH__(BITS, T, F2, true);                                                           \
            else PH_CONV_LAUNCH__(BITS, T, F2, false);                                                               \
        } else PH_CONV_LAUNCH__(BITS, T, F2, false);                                                                 \
    } while (0)
#define PH_CONV_LAUNCH(BITS, T)                                                                                      \
    do {                                                                                                             \
        if constexpr (PF == 1 && PK == 2) {                                                                          \
            if (two_halves) PH_CONV_LAUNCH_(BITS, T, true);                                                          \
            else PH_CONV_LAUNCH_(BITS, T, false);                                                                    \
        } else PH_CONV_LAUNCH_(BITS, T, false);                                                                      \
    } while (0)
    if (bits_out) PH_CONV_LAUNCH(true, float);
    else if (out_dtype == PH_OUT_F32) PH_CONV_LAUNCH(false, float);
    else if (out_dtype == PH_OUT_F16) PH_CONV_LAUNCH(false, ph_h16);
    else PH_CONV_LAUNCH(false, uint16_t);
#undef PH_CONV_LAUNCH
#undef PH_CONV_LAUNCH_
#undef PH_CONV_LAUNCH__
    return 0;
}

extern "C" int ph_dynconv(const uint16_t* planes, const uint16_t* kern, int64_t kern_plane_stride,
                          int64_t kern_batch_stride, const float* kbias, int64_t kbias_batch_stride, uint32_t* bits_out,
                          void* logits_out, int out_dtype, int64_t out_batch_stride, int B, int N, int64_t HW, int prec,
                          void* stream) {
    PH_CHECK_ARG(planes && kern && kbias && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    PH_CHECK_ARG((bits_out != nullptr) != (logits_out != nullptr), "exactly one of bits_out / logits_out");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_BF16_KSPLIT || prec == PH_PREC_F16 ||
                 prec == PH_PREC_BF16_KF16, "bad prec");
    PH_CHECK_ARG(out_dtype == PH_OUT_F32 || out_dtype == PH_OUT_BF16 || out_dtype == PH_OUT_F16, "bad out_dtype");
    PH_CHECK_ARG(N <= 256, "at most 256 queries");
    const int nrt = ph_n_padded(N) / 32;
    hipStream_t s = (hipStream_t)stream;
#define PH_CONV_ARGS planes, kern, kern_plane_stride, kern_batch_stride, kbias, kbias_batch_stride, bits_out, logits_out, \
                     out_dtype, out_batch_stride, B, N, HW, s
#define PH_CONV_CASE(R)                                                                                         \
    case R:                                                                                                     \
        if (prec == PH_PREC_BF16) launch_conv<1, 1, PH_E_BF16, R>(PH_CONV_ARGS);                                \
        else if (prec == PH_PREC_F16) launch_conv<1, 1, PH_E_F16, R>(PH_CONV_ARGS);                             \
        else if (prec == PH_PREC_BF16_KSPLIT) launch_conv<1, 2, PH_E_BF16, R>(PH_CONV_ARGS);                    \
        else if (prec == PH_PREC_BF16_KF16) launch_conv<1, 1, PH_E_F16_FROM_BF16, R>(PH_CONV_ARGS);             \
        else launch_conv<2, 2, PH_E_BF16, R>(PH_CONV_ARGS);                                                     \
        break;
    switch (nrt) {
        PH_CONV_CASE(1) PH_CONV_CASE(2) PH_CONV_CASE(3) PH_CONV_CASE(4)
        PH_CONV_CASE(5) PH_CONV_CASE(6) PH_CONV_CASE(7) PH_CONV_CASE(8)
        default: ph_set_error("ph_dynconv: unsupported N"); return PH_EUNSUPPORTED;
    }
#undef PH_CONV_CASE
#undef PH_CONV_ARGS
    PH_CHECK_LAUNCH();
    return PH_OK;
}
