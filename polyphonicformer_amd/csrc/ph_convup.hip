// A13 of the FINAL stage + A14 in one kernel (round 4): logits = kern . feat + kbias (kernel_update_head.py:317-329) and their
// x2 bilinear upsample (kernel_update.py:131-143, F.interpolate(scale_factor=2, mode='bilinear', align_corners=False)) from ONE
// read of the feature plane.  The two-kernel form (ph_dynconv -> ph_upsample2x) writes the low-resolution logits and reads
// them back twice over (every source row serves two row pairs): per frame at cfg2 10 MB written + 10 MB read for the mask
// branch, and the same for the depth branch whose low-resolution logits no caller ever sees (simple_test_mask_preds returns
// mask_preds and scaled_mask_preds, simple_test hands on scaled_depth_preds: kernel_update.py:338-345,401).  Here the low-
// resolution tile never leaves the CU: what leaves is the upsampled rows (and, mask branch only, the low-resolution logits
// the API returns).
//
// Geometry.  W must be NTR * 64 (NTR tiles of 64 pixels per image row; instantiated for NTR = 4, i.e. W = 256 = 2048 / 8)
// so that tiles never straddle image rows; other widths keep the two-kernel form.  One persistent workgroup per CU owns a
// contiguous range of image ROWS (of the B * H rows of the batch) and walks them tile by tile in row-major order:
//   * NRT consumer waves, wave rt = query rows 32 rt .. 32 rt + 31 with its whole A operand in registers (64 VGPRs), BOTH
//     32-pixel halves of a tile (2 x 16 MFMA 32x32x16), and one PRODUCER wave that issues the LDS-DMA ring (ph_conv.hip's
//     recipe: whole 128-byte lines, XOR swizzle on the source side, counted vmcnt).  The consumers' vector-memory queues then
//     hold nothing but their own output stores -- ~40 per tile and wave -- which never have to drain inside the loop; in
//     ph_conv.hip's form (every wave issues DMA and counts it with vmcnt) each tile would wait for the stores in front of it.
//   * vertical direction: the previous image row of a wave's own accumulator positions stays in REGISTERS as packed 16-bit
//     pairs (NTR tiles x 2 halves x 8 VGPRs = 64 at W = 256): out row 2r - 1 = .75 P + .25 C, out row 2r = .25 P + .75 C are
//     lane-local arithmetic in the MFMA D layout.  A workgroup whose range starts inside a frame first runs the row above it
//     silently (halo: + 1 / 12 of the tiles at cfg2's 24 frames).
//   * horizontal direction: the vertically blended [32 q][32 px] tile goes through a per-wave fp32 LDS patch; a lane then owns
//     8 consecutive output pixels of one query row (16-byte stores).  The window a half emits is shifted LEFT by 8 output
//     pixels (= 16 bytes, so stores stay aligned): it needs 5 source columns of the previous half -- kept in the patch -- and
//     nothing of the next one; the last half of an image row flushes the remaining 8 pixels.
// Same results as ph_upsample2x on the 16-bit logits to fp32 rounding (vertical-then-horizontal instead of ATen's
// horizontal-then-vertical), i.e. equal after the final 16-bit rounding except on rounding-boundary cases.
#include <stdlib.h>

#include <type_traits>

#include "ph_conv_inl.h"

template <int OFF> __device__ __forceinline__ float lds_read32_asm(uint32_t byte_addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ u32x4_t lds_read128o_asm(uint32_t byte_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ void lds_write128o_asm(uint32_t byte_addr, u32x4_t v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(byte_addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait_all() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int NRT> struct UpCfg {
    static constexpr int NW = NRT + 1;                           // consumer waves + the producer wave
    static constexpr int TILEB = 256 * CONV_T * 2;               // bytes per ring stage (one 16-bit plane)
    static constexpr int PLD = 40;                               // patch row: 8 history + 32 current source columns (floats)
    static constexpr int PATCH1 = 32 * PLD * 4;
    static constexpr int PATCHB = NRT * 2 * PATCH1;              // two patches per consumer wave (odd / even output row)
    static constexpr int KBB = NRT * 32 * 4;
    static constexpr int NBUF = 3 * TILEB + PATCHB + KBB <= 160 * 1024 ? 3 : 2;
    static constexpr int LDSB = NBUF * TILEB + PATCHB + KBB;
};

template <typename OutT> struct UpElem;
template <> struct UpElem<ph_h16> { static constexpr int E = PH_E_F16; };
template <> struct UpElem<uint16_t> { static constexpr int E = PH_E_BF16; };

// E: MFMA element format (PH_E_BF16 / PH_E_F16 / PH_E_F16_FROM_BF16 = bf16 plane converted to fp16 once per tile in LDS);
// OutT: ph_h16 (fp16) or uint16_t (bf16) outputs; LOWRES: also write the low-resolution logits [B][N][H][W]
template <int E, int NRT, int NTR, bool LOWRES, typename OutT>
__global__ __launch_bounds__(((NRT + 1) * 64)) void k_dynconv_up2(const uint16_t* __restrict__ planes, const uint16_t* __restrict__ kern,
                                                                 int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                                 int64_t kbias_batch_stride, OutT* __restrict__ logits_out,
                                                                 OutT* __restrict__ up_out, int B, int N, int H) {
    using C = UpCfg<NRT>;
    constexpr int NBUF = C::NBUF, W = NTR * 64, PLD = C::PLD, EO = UpElem<OutT>::E;
    constexpr bool COOP = E == PH_E_F16_FROM_BF16;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [NBUF][256][64] | patches [NRT][2][32][PLD] f32 | biases [NRT][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave == NRT;
    const int rt = producer ? 0 : wave;
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1, col = lane & 31;
    const int64_t HW = (int64_t)H * W;
    const int ntiles = H * NTR;

    // this workgroup's image rows [R0, R1) of the B * H rows; a range that starts inside a frame first runs the row above silently
    const int64_t rows_total = (int64_t)B * H;
    const int R0 = (int)(rows_total * blockIdx.x / gridDim.x), R1 = (int)(rows_total * (blockIdx.x + 1) / gridDim.x);
    if (R0 >= R1) return;
    const int halo = (R0 % H) != 0 ? 1 : 0;
    const int tg0 = (R0 - halo) * NTR, tg1 = R1 * NTR;

    // ---- producer state: the LDS-DMA ring (32 instructions of 1 KiB = 8 channel rows x 128 B per tile)
    const uint32_t dma_lane_off = 2u * (uint32_t)((lane >> 3) * HW + (((lane & 7) ^ conv_swz(lane >> 3)) * 8));
    const int64_t row8_bytes = 2 * 8 * HW;
    const int64_t frame_jump = 2 * ((int64_t)PH_C * HW - (int64_t)ntiles * CONV_T);
    int it = tg0 % ntiles, ti = tg0;
    const char* iptr = (const char*)planes + 2 * ((int64_t)(tg0 / ntiles) * PH_C * HW + (int64_t)it * CONV_T);
    auto issue_next = [&](int buf) {
        const char* src = iptr + dma_lane_off;
#pragma unroll
        for (int j = 0; j < 32; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * row8_bytes),
                                             (PH_LDS void*)((PH_LDS char*)lds + buf * C::TILEB + j * 1024), 16, 0, PH_CPOL_STREAM);
        ++ti;
        iptr += 2 * CONV_T;
        if (++it == ntiles) { it = 0; iptr += frame_jump; }
    };
    if (producer) {
#pragma unroll
        for (int d = 0; d < NBUF - 1; ++d)
            if (ti < tg1) issue_next(d);
    }

    // ---- consumer state
    const int row0 = g * 8 + (i16 >> 2);
    uint32_t frag_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) frag_off[h] = 2u * (uint32_t)(row0 * CONV_T + (((h * 4 + gi * 2) ^ conv_swz(row0)) * 8) + (i16 & 3) * 4);
    const uint32_t lds0 = lds_addr(lds);
    const uint32_t patchA = lds0 + NBUF * C::TILEB + rt * 2 * C::PATCH1, patchB = patchA + C::PATCH1;
    float* kb_lds = (float*)((unsigned char*)lds + NBUF * C::TILEB + C::PATCHB) + rt * 32;
    const uint32_t kb_addr = lds_addr(kb_lds) + 16 * g;
    // patch addresses of this lane (bytes).  D layout: lane = source column `col`, rows rr + 4 g
    const uint32_t pw_off = (uint32_t)(((4 * g) * PLD + 8 + col) * 4);                 // + rr * PLD * 4 immediates
    const int hq = lane >> 1, hpart = lane & 1;                                        // history copy: row, 4-column part
    const uint32_t hist_off = (uint32_t)((hq * PLD + 4 * hpart) * 4);                   // reads + 32 columns
    const int sq = lane >> 3, ss = lane & 7;                                           // horizontal pass: row (+ 8 kk), 8-pixel segment
    const uint32_t pr_off = (uint32_t)((sq * PLD + 4 * ss) * 4);                        // + kk * 8 * PLD * 4, element offsets 3 / 4 / 8
    const int lq = lane >> 2, lseg = lane & 3;                                         // low-resolution pass: row (+ 16 j), 8-pixel segment
    const uint32_t lr_off = (uint32_t)((lq * PLD + 8 + 8 * lseg) * 4);
    // per-lane parts of the output addresses (bytes)
    const int64_t up_plane = (int64_t)2 * H * 2 * W;                                   // elements per (frame, query) of the upsampled tensor
    const uint32_t up_lane_off = 2u * (uint32_t)(sq * up_plane + 8 * ss);
    const int64_t up_kk_bytes = 2 * 8 * up_plane;
    const uint32_t upf_lane_off = 2u * (uint32_t)((lane & 31) * up_plane);               // row-end flush: one row per lane (lanes 0..31)
    const uint32_t lr_lane_off = 2u * (uint32_t)(lq * HW + 8 * lseg);
    const int64_t lr_j_bytes = 2 * 16 * HW;

    uint4 af[1][16];
    uint32_t prev[NTR][2][8];                        // the previous image row at this wave's accumulator positions, packed 16-bit pairs
#pragma unroll
    for (int a = 0; a < NTR; ++a)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[a][h][j] = 0;

    // one horizontal pass: the [32 q][40] fp32 patch (history in columns 0..7) -> 16-bit output row `R` of the upsampled tensor.
    // HH = half index in the image row (compile time).
    auto hpass = [&](uint32_t patch, char* ubase /* wave-uniform: (b, rt * 32, row R, column 64 HH - 8) */, auto hh_tag) {
        constexpr int HH = decltype(hh_tag)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint32_t a = patch + pr_off + kk * 8 * PLD * 4;
            const float f3 = lds_read32_asm<12>(a);
            const u32x4_t m = lds_read128o_asm<16>(a);
            const float f8 = lds_read32_asm<32>(a);
            lds_wait_all();
            const float f4 = __uint_as_float(m.x), f5 = __uint_as_float(m.y), f6 = __uint_as_float(m.z), f7 = __uint_as_float(m.w);
            float o0 = 0.25f * f3 + 0.75f * f4;
            if (HH == 0 && ss == 1) o0 = f4;                         // output column 0: source index clamped at 0, weights (1, 0)
            const float o1 = 0.75f * f4 + 0.25f * f5, o2 = 0.25f * f4 + 0.75f * f5, o3 = 0.75f * f5 + 0.25f * f6;
            const float o4 = 0.25f * f5 + 0.75f * f6, o5 = 0.75f * f6 + 0.25f * f7, o6 = 0.25f * f6 + 0.75f * f7;
            const float o7 = 0.75f * f7 + 0.25f * f8;
            const uint4 pk = make_uint4(f2e_pk<EO>(o0, o1), f2e_pk<EO>(o2, o3), f2e_pk<EO>(o4, o5), f2e_pk<EO>(o6, o7));
            const bool ok = rt * 32 + sq + 8 * kk < N && !(HH == 0 && ss == 0);
            if (ok) st_nt16(ubase + kk * up_kk_bytes + up_lane_off, pk);
        }
        if (HH == 2 * NTR - 1) {
            // the last 8 output columns of the image row: source columns W-5 .. W-1 = patch columns 35 .. 39, the column
            // past the end clamped (ATen: i1 = min(i0 + 1, W - 1))
            const uint32_t a = patch + (uint32_t)(((lane & 31) * PLD) * 4);
            const float f3 = lds_read32_asm<35 * 4>(a);
            const u32x4_t m = lds_read128o_asm<36 * 4>(a);
            lds_wait_all();
            const float f4 = __uint_as_float(m.x), f5 = __uint_as_float(m.y), f6 = __uint_as_float(m.z), f7 = __uint_as_float(m.w);
            const float f8 = f7;
            const float o0 = 0.25f * f3 + 0.75f * f4, o1 = 0.75f * f4 + 0.25f * f5, o2 = 0.25f * f4 + 0.75f * f5, o3 = 0.75f * f5 + 0.25f * f6;
            const float o4 = 0.25f * f5 + 0.75f * f6, o5 = 0.75f * f6 + 0.25f * f7, o6 = 0.25f * f6 + 0.75f * f7, o7 = 0.75f * f7 + 0.25f * f8;
            const uint4 pk = make_uint4(f2e_pk<EO>(o0, o1), f2e_pk<EO>(o2, o3), f2e_pk<EO>(o4, o5), f2e_pk<EO>(o6, o7));
            if (lane < 32 && rt * 32 + lane < N) st_nt16(ubase + 2 * 64 + upf_lane_off, pk);     // window base + 64 = column 2 W - 8
        }
    };
    // history: the previous half's last 8 source columns (patch columns 32..39) -> columns 0..7; must run before ANYTHING
    // overwrites columns 32..39 of that patch for the new half
    auto hist_move = [&](uint32_t patch) {
        const u32x4_t t = lds_read128o_asm<32 * 4>(patch + hist_off);
        lds_wait_all();
        lds_write128o_asm<0>(patch + hist_off, t);
    };
    // the new [32 q][32 px] values into columns 8..39 of a patch
    auto patch_fill = [&](uint32_t patch, const f32x16_t& v, auto hh_tag) {
        constexpr int HH = decltype(hh_tag)::value;
        conv_patch_put<float, PLD>(patch + pw_off, v);
        if (HH == 0) {
            // left image border: source column -1 := column 0 (patch column 7 := 8)
            lds_wait_all();
            const float t = lds_read32_asm<8 * 4>(patch + (uint32_t)((lane & 31) * PLD * 4));
            lds_wait_all();
            if (lane < 32) lds_write_asm<7 * 4>(patch + (uint32_t)((lane & 31) * PLD * 4), t, (float*)nullptr);
        }
        lds_wait_all();
    };

    int cur = 0, cur_b = -1;
    for (int gr = R0 - halo; gr < R1; ++gr) {
        const int b = gr / H, r = gr - b * H;
        const bool silent = gr < R0, first = r == 0, last = r == H - 1;
        if (!producer && b != cur_b) {
            // the A operand and biases of the frame (ordinary loads; the wait also drains this wave's stores, once per frame)
            const uint16_t* kr = kern + (int64_t)b * kern_batch_stride + (rt * 32 + (lane & 31)) * PH_C + g * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) af[0][ks] = *(const uint4*)(kr + ks * 16);
            if (lane < 32) kb_lds[lane] = kbias[(int64_t)b * kbias_batch_stride + rt * 32 + lane];
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
        }
        cur_b = b;
        // wave-uniform output bases of this image row
        char* up_row = (char*)(up_out + ((int64_t)b * N + rt * 32) * up_plane + (int64_t)(2 * r) * (2 * W));    // output row 2r
        char* lr_row = LOWRES ? (char*)(logits_out + ((int64_t)b * N + rt * 32) * HW + (int64_t)r * W) : nullptr;

        auto tile = [&](auto tc_tag) {
            constexpr int TC = decltype(tc_tag)::value;
            const int tg = gr * NTR + TC;
            if (producer) {
                // tile tg must have landed; NBUF - 2 younger tiles may stay in flight
                const int younger = (tg1 - 1 - tg) < (NBUF - 2) ? (tg1 - 1 - tg) : (NBUF - 2);
                if (NBUF >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (producer && ti < tg1) {
                int nb = cur + NBUF - 1;
                if (nb >= NBUF) nb -= NBUF;
                issue_next(nb);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (COOP) {
                // bf16 -> fp16, once per tile, in place, every wave (the producer included) a share of the 16-byte pieces
                constexpr int PIECES = 256 * CONV_T * 2 / 16, LANES = C::NW * 64, ROUNDS = (PIECES + LANES - 1) / LANES;
                const uint32_t tb = lds0 + cur * C::TILEB + 16u * (uint32_t)tid;
                u32x4_t cv[ROUNDS];
#pragma unroll
                for (int q = 0; q < ROUNDS; ++q)
                    if ((q + 1) * LANES <= PIECES || tid < PIECES - q * LANES) cv[q] = lds_read128_asm(tb + q * LANES * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < ROUNDS; ++q)
                    if ((q + 1) * LANES <= PIECES || tid < PIECES - q * LANES) {
                        const uint4 h16 = bf2h_x8(__builtin_bit_cast(uint4, cv[q]));
                        lds_write128_asm(tb + q * LANES * 16, __builtin_bit_cast(u32x4_t, h16));
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!producer) {
                float bias[16];
                conv_bias_get(kb_addr, bias);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    constexpr int HH0 = 2 * TC;
                    const uint32_t fa = lds0 + cur * C::TILEB + frag_off[h];
                    f32x16_t acc;
                    constexpr int KB = 2;
                    u32x2_t bq[2][1][KB][2];
                    conv_read_batch<1, KB, 0>(fa, bq[0]);
                    conv_batches<1, 1, E, KB, 0, true>(fa, af, bq, acc, bias);
                    // the low-resolution logits as the 16-bit values the API returns; everything below blends THOSE
                    uint32_t cu[8];
                    f32x16_t Cf, Pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        cu[j] = f2e_pk<EO>(acc[2 * j], acc[2 * j + 1]);
                        Cf[2 * j] = e2f<EO>(cu[j] & 0xFFFFu);
                        Cf[2 * j + 1] = e2f<EO>(cu[j] >> 16);
                        Pf[2 * j] = e2f<EO>(prev[TC][h][j] & 0xFFFFu);
                        Pf[2 * j + 1] = e2f<EO>(prev[TC][h][j] >> 16);
                        prev[TC][h][j] = cu[j];
                    }
                    if (!silent) {
                        auto body = [&](auto hh_tag) {
                            constexpr int HH = decltype(hh_tag)::value;
                            char* ub = up_row + 2 * (64 * HH - 8);                      // window base of output row 2r
                            if (HH > 0) {           // both patches' histories first: the passes below overwrite columns 32..39
                                if (!first) hist_move(patchA);
                                hist_move(patchB);
                            }
                            if (LOWRES) {
                                conv_patch_put<float, PLD>(patchA + pw_off, Cf);        // columns 8..39; the history (0..7) is untouched
                                lds_wait_all();
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const u32x4_t x0 = lds_read128o_asm<0>(patchA + lr_off + j * 16 * PLD * 4);
                                    const u32x4_t x1 = lds_read128o_asm<16>(patchA + lr_off + j * 16 * PLD * 4);
                                    lds_wait_all();
                                    const uint4 pk = make_uint4(f2e_pk<EO>(__uint_as_float(x0.x), __uint_as_float(x0.y)),
                                                                f2e_pk<EO>(__uint_as_float(x0.z), __uint_as_float(x0.w)),
                                                                f2e_pk<EO>(__uint_as_float(x1.x), __uint_as_float(x1.y)),
                                                                f2e_pk<EO>(__uint_as_float(x1.z), __uint_as_float(x1.w)));
                                    // default (cached) stores: a caller reads these next
                                    if (rt * 32 + lq + 16 * j < N) *(uint4*)(lr_row + 2 * 32 * HH + j * lr_j_bytes + lr_lane_off) = pk;
                                }
                            }
                            if (!first) {
                                // output row 2r - 1: source rows (r-1, r), weights (.75, .25)
                                f32x16_t v;
#pragma unroll
                                for (int q = 0; q < 16; ++q) v[q] = 0.75f * Pf[q] + 0.25f * Cf[q];
                                patch_fill(patchA, v, hh_tag);
                                hpass(patchA, ub - 2 * (2 * W), hh_tag);
                            }
                            {
                                // output row 2r: source rows (r-1, r), weights (.25, .75); r = 0: clamped, weights (0, 1)
                                f32x16_t v;
#pragma unroll
                                for (int q = 0; q < 16; ++q) v[q] = first ? Cf[q] : 0.25f * Pf[q] + 0.75f * Cf[q];
                                patch_fill(patchB, v, hh_tag);
                                hpass(patchB, ub, hh_tag);
                            }
                        };
                        if (h == 0) body(std::integral_constant<int, HH0>{});
                        else body(std::integral_constant<int, HH0 + 1>{});
                    }
                }
            }
            cur = cur + 1 == NBUF ? 0 : cur + 1;
        };
        // the NTR tiles of this image row
        tile(std::integral_constant<int, 0>{});
        if constexpr (NTR > 1) tile(std::integral_constant<int, 1>{});
        if constexpr (NTR > 2) tile(std::integral_constant<int, 2>{});
        if constexpr (NTR > 3) tile(std::integral_constant<int, 3>{});
        static_assert(NTR >= 1 && NTR <= 4, "image rows of 64, 128, 192 or 256 pixels");

        if (!producer && last && !silent) {
            // output row 2H - 1: source index clamped at H - 1 on both sides -> .75 C + .25 C of the row just finished (in `prev`)
            char* ub_last = up_row + 2 * (2 * W);
            auto fin = [&](auto tc_tag, auto h_tag) {
                constexpr int TC = decltype(tc_tag)::value, HX = decltype(h_tag)::value;
                f32x16_t v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float c0 = e2f<EO>(prev[TC][HX][j] & 0xFFFFu), c1 = e2f<EO>(prev[TC][HX][j] >> 16);
                    v[2 * j] = 0.75f * c0 + 0.25f * c0;
                    v[2 * j + 1] = 0.75f * c1 + 0.25f * c1;
                }
                if (2 * TC + HX > 0) hist_move(patchA);
                patch_fill(patchA, v, std::integral_constant<int, 2 * TC + HX>{});
                hpass(patchA, ub_last + 2 * (64 * (2 * TC + HX) - 8), std::integral_constant<int, 2 * TC + HX>{});
            };
#define UP_FIN(T)                                                                        \
    do {                                                                                 \
        fin(std::integral_constant<int, T>{}, std::integral_constant<int, 0>{});         \
        fin(std::integral_constant<int, T>{}, std::integral_constant<int, 1>{});         \
    } while (0)
            UP_FIN(0);
            if constexpr (NTR > 1) UP_FIN(1);
            if constexpr (NTR > 2) UP_FIN(2);
            if constexpr (NTR > 3) UP_FIN(3);
#undef UP_FIN
        }
    }
}

// ====================================================================================================================
static int up_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// 1 when the fused final conv + x2 upsample exists for this geometry / arithmetic (otherwise: ph_dynconv + ph_upsample2x)
extern "C" int ph_dynconv_up2_supported(int N, int H, int W, int prec, int out_dtype) {
    // instantiated for NTR = 4 (W = 256 = 2048 / 8) and 3..7 row blocks: 65 <= N <= 224 (8 row blocks would be 9 waves, three on
    // one SIMD, i.e. 168 registers per wave against the ~230 this kernel keeps live)
    if (N <= 64 || N > 224 || H <= 0 || W != 256) return 0;
    if (((int64_t)H * W) % 128) return 0;
    if (prec == PH_PREC_BF16) return out_dtype == PH_OUT_BF16;
    if (prec == PH_PREC_F16 || prec == PH_PREC_BF16_KF16) return out_dtype == PH_OUT_F16;
    return 0;
}

template <int E, int NRT, typename OutT>
static void launch_up(const uint16_t* planes, const uint16_t* kern, int64_t kbs, const float* kbias, int64_t bbs, void* logits_out,
                      void* up_out, int B, int N, int H, hipStream_t s) {
    constexpr int NTR = 4;
    const int64_t rows = (int64_t)B * H;
    int wgs = up_num_cus();
    if (const char* e = getenv("PH_UP2_WGS")) wgs = atoi(e) > 0 ? atoi(e) : wgs;      // tuning / test knob: rows per workgroup
    if (wgs > rows) wgs = (int)rows;
    const dim3 grid(wgs), block((NRT + 1) * 64);
    constexpr int lds = UpCfg<NRT>::LDSB;
#define UP_GO(LR)                                                                                                             \
    do {                                                                                                                      \
        static const bool once = [] {                                                                                         \
            (void)hipFuncSetAttribute((const void*)k_dynconv_up2<E, NRT, NTR, LR, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            return true;                                                                                                      \
        }();                                                                                                                  \
        (void)once;                                                                                                           \
        hipLaunchKernelGGL((k_dynconv_up2<E, NRT, NTR, LR, OutT>), grid, block, lds, s, planes, kern, kbs, kbias, bbs, (OutT*)logits_out, \
                           (OutT*)up_out, B, N, H);                                                                           \
    } while (0)
    if (logits_out) UP_GO(true);
    else UP_GO(false);
#undef UP_GO
}

extern "C" int ph_dynconv_up2(const uint16_t* planes, const uint16_t* kern, int64_t kern_batch_stride, const float* kbias,
                              int64_t kbias_batch_stride, void* logits_out, void* up_out, int out_dtype, int B, int N, int H, int W,
                              int prec, void* stream) {
    PH_CHECK_ARG(planes && kern && kbias && up_out && B > 0, "bad pointer or size");
    if (!ph_dynconv_up2_supported(N, H, W, prec, out_dtype)) {
        ph_set_error("ph_dynconv_up2: unsupported geometry or arithmetic (use ph_dynconv + ph_upsample2x)");
        return PH_EUNSUPPORTED;
    }
    const int nrt = ph_n_padded(N) / 32;
    hipStream_t s = (hipStream_t)stream;
#define UP_ARGS planes, kern, kern_batch_stride, kbias, kbias_batch_stride, logits_out, up_out, B, N, H, s
#define UP_CASE(R)                                                                                   \
    case R:                                                                                          \
        if (prec == PH_PREC_BF16) launch_up<PH_E_BF16, R, uint16_t>(UP_ARGS);                        \
        else if (prec == PH_PREC_F16) launch_up<PH_E_F16, R, ph_h16>(UP_ARGS);                       \
        else launch_up<PH_E_F16_FROM_BF16, R, ph_h16>(UP_ARGS);                                      \
        break;
    switch (nrt) {
        UP_CASE(3) UP_CASE(4) UP_CASE(5) UP_CASE(6) UP_CASE(7)
        default: ph_set_error("ph_dynconv_up2: unsupported N"); return PH_EUNSUPPORTED;
    }
#undef UP_CASE
#undef UP_ARGS
    PH_CHECK_LAUNCH();
    return PH_OK;
}
