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

template <int OFF> __device__ __forceinline__ uint32_t lds_read32u_asm(uint32_t byte_addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ u32x2_t lds_read64_asm(uint32_t byte_addr) {
    u32x2_t v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ u32x4_t lds_read128o_asm(uint32_t byte_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
// the low / the high 16 bits of a register to LDS
template <int OFF> __device__ __forceinline__ void lds_write16lo_asm(uint32_t byte_addr, uint32_t v) {
    asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(byte_addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write16hi_asm(uint32_t byte_addr, uint32_t v) {
    asm volatile("ds_write_b16_d16_hi %0, %1 offset:%2" ::"v"(byte_addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait_all() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// The upsampled rows and the low-resolution rows leave as whole, aligned 128-byte lines (8 lanes x 16 bytes per query row), so
// they can bypass the caches like the stand-alone upsample kernel's output (-DUP2_CACHED_STORES: default policy, for A/B runs).
// The first version of this kernel emitted windows shifted by 16 bytes (no data of the NEXT half needed, but every line
// completed by two stores): 609 us per 24 frames with cached stores, 1 ms with non-temporal ones, 268 us with the stores
// compiled out -- the memory system, not the arithmetic, and the reason for the one-half-late emission below.
__device__ __forceinline__ void up_store(void* p, uint4 v) {
#ifdef UP2_CACHED_STORES
    *(uint4*)p = v;
#else
    st_nt16(p, v);
#endif
}

// timing experiments only (-DUP2_TIMING + PH_UP2_DBG=bits at run time): 1 = no stores, 2 = no window passes, 4 = every row silent
#ifdef UP2_TIMING
#define UP2_DBG(bit) (dbg & (bit))
#else
#define UP2_DBG(bit) false
#endif

template <int NRT> struct UpCfg {
    static constexpr int NW = NRT + 1;                           // consumer waves + the producer wave
    static constexpr int TILEB = 256 * CONV_T * 2;               // bytes per ring stage (one 16-bit plane)
    // per consumer wave: two 16-bit patches (P = previous image row, C = current one) of [32 q][3 slots x 32 source columns]:
    // half hh of an image row lives in slot hh % 3, so that when half hh has been written the window of half hh - 1 -- which
    // needs the last column of hh - 2 and the first of hh -- can be emitted
    static constexpr int ROWB = 3 * 64 + 16;                     // bytes per patch row (16 bytes of padding)
    static constexpr int PATCH1 = 32 * ROWB;
    static constexpr int PATCHB = NRT * 2 * PATCH1;
    static constexpr int KBB = NRT * 32 * 4;
    static constexpr int NBUF = 3 * TILEB + PATCHB + KBB <= 160 * 1024 ? 3 : 2;
    static constexpr int LDSB = NBUF * TILEB + PATCHB + KBB;
};

template <typename OutT> struct UpElem;
template <> struct UpElem<ph_h16> { static constexpr int E = PH_E_F16; };
template <> struct UpElem<uint16_t> { static constexpr int E = PH_E_BF16; };

// patch writes of a lane's 16 D-layout values (packed pairs: register j = rows rr(2j) | rr(2j+1) << 16), immediate row offsets
template <int ROWB, int BASE, int J = 0>
__device__ __forceinline__ void up_patch_put(uint32_t wa, const uint32_t (&v)[8]) {
    if constexpr (J < 8) {
        constexpr int r0 = 2 * J, r1 = 2 * J + 1;
        constexpr int rr0 = (r0 & 3) + 8 * (r0 >> 2), rr1 = (r1 & 3) + 8 * (r1 >> 2);
        lds_write16lo_asm<BASE + rr0 * ROWB>(wa, v[J]);
        lds_write16hi_asm<BASE + rr1 * ROWB>(wa, v[J]);
        up_patch_put<ROWB, BASE, J + 1>(wa, v);
    }
}

// E: MFMA element format (PH_E_BF16 / PH_E_F16 / PH_E_F16_FROM_BF16 = bf16 plane converted to fp16 once per tile in LDS);
// OutT: ph_h16 (fp16) or uint16_t (bf16) outputs; LOWRES: also write the low-resolution logits [B][N][H][W]
template <int E, int NRT, int NTR, bool LOWRES, typename OutT>
__global__ __launch_bounds__(((NRT + 1) * 64)) void k_dynconv_up2(const uint16_t* __restrict__ planes, const uint16_t* __restrict__ kern,
                                                                 int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                                 int64_t kbias_batch_stride, OutT* __restrict__ logits_out,
                                                                 OutT* __restrict__ up_out, int B, int N, int H, int dbg) {
    using C = UpCfg<NRT>;
    constexpr int NBUF = C::NBUF, W = NTR * 64, ROWB = C::ROWB, EO = UpElem<OutT>::E, NH = 2 * NTR;
    constexpr bool COOP = E == PH_E_F16_FROM_BF16;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [NBUF][256][64] | patches [NRT][P, C][32][ROWB] | biases [NRT][32]
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
    const uint32_t patchP = lds0 + NBUF * C::TILEB + rt * 2 * C::PATCH1;            // the C patch follows at + PATCH1
    float* kb_lds = (float*)((unsigned char*)lds + NBUF * C::TILEB + C::PATCHB) + rt * 32;
    const uint32_t kb_addr = lds_addr(kb_lds) + 16 * g;
    // D layout (lane = source column `col`, rows rr + 4 g) -> patch write address; + rr * ROWB + slot * 64 as immediates
    const uint32_t pw_addr = patchP + (uint32_t)((4 * g) * ROWB + 2 * col);
    // window pass: lane = query row sq (+ 8 kk), 4 source columns 4 ss .. 4 ss + 3 of a half (8 output pixels)
    const int sq = lane >> 3, ss = lane & 7;
    const uint32_t pr_addr = patchP + (uint32_t)(sq * ROWB + 8 * ss);               // + slot * 64 + kk * 8 * ROWB (+ PATCH1 for C)
    // per-lane parts of the output addresses (bytes)
    const int64_t up_plane = (int64_t)2 * H * 2 * W;                                 // elements per (frame, query) of the upsampled tensor
    const uint32_t up_lane_off = 2u * (uint32_t)(sq * up_plane + 8 * ss);
    const int64_t up_kk_bytes = 2 * 8 * up_plane;
    const uint32_t lr_lane_off = 2u * (uint32_t)(sq * HW + 8 * ss);
    const int64_t lr_kk_bytes = 2 * 8 * HW;

    uint4 af[1][16];
    uint32_t prev[NTR][2][8];                        // the previous image row at this wave's accumulator positions, packed 16-bit pairs
#pragma unroll
    for (int a = 0; a < NTR; ++a)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[a][h][j] = 0;

    // blend of one image-row pair over 6 source columns (index 0 = left neighbour, 1..4 = the lane's columns, 5 = right
    // neighbour) -> 8 output pixels of one output row: ATen's x2 weights, even output column 2j = .25 in[j-1] + .75 in[j], odd
    // 2j + 1 = .75 in[j] + .25 in[j+1]
    auto hblend = [&](const float (&f)[6], bool exact0) -> uint4 {
        float o0 = 0.25f * f[0] + 0.75f * f[1];
        if (exact0) o0 = f[1];                                   // output column 0: source index clamped at 0, weights (1, 0)
        const float o1 = 0.75f * f[1] + 0.25f * f[2], o2 = 0.25f * f[1] + 0.75f * f[2], o3 = 0.75f * f[2] + 0.25f * f[3];
        const float o4 = 0.25f * f[2] + 0.75f * f[3], o5 = 0.75f * f[3] + 0.25f * f[4], o6 = 0.25f * f[3] + 0.75f * f[4];
        const float o7 = 0.75f * f[4] + 0.25f * f[5];
        return make_uint4(f2e_pk<EO>(o0, o1), f2e_pk<EO>(o2, o3), f2e_pk<EO>(o4, o5), f2e_pk<EO>(o6, o7));
    };

    // one query row's share of a window: 4 source columns (+ 2 neighbours) of the row pair (P, C) -> 8 output pixels of output rows
    // 2r - 1 (unless r = 0), 2r and, on the last image row, 2H - 1; `dst` = the 16 bytes of output row 2r
    auto blend_store = [&](uint32_t pl, u32x2_t pb, uint32_t pr, uint32_t cl, u32x2_t cb, uint32_t cr, bool left_hi, bool right_hi, bool exact0,
                           bool ok, bool rfirst, bool rlast, char* dst) {
        float pf[6], cf[6];
        pf[0] = e2f<EO>(left_hi ? pl >> 16 : pl & 0xFFFFu);
        pf[1] = e2f<EO>(pb.x & 0xFFFFu); pf[2] = e2f<EO>(pb.x >> 16);
        pf[3] = e2f<EO>(pb.y & 0xFFFFu); pf[4] = e2f<EO>(pb.y >> 16);
        pf[5] = e2f<EO>(right_hi ? pr >> 16 : pr & 0xFFFFu);
        cf[0] = e2f<EO>(left_hi ? cl >> 16 : cl & 0xFFFFu);
        cf[1] = e2f<EO>(cb.x & 0xFFFFu); cf[2] = e2f<EO>(cb.x >> 16);
        cf[3] = e2f<EO>(cb.y & 0xFFFFu); cf[4] = e2f<EO>(cb.y >> 16);
        cf[5] = e2f<EO>(right_hi ? cr >> 16 : cr & 0xFFFFu);
        float v[6];
        if (!rfirst) {
            // output row 2r - 1: source rows (r - 1, r), weights (.75, .25)
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = 0.75f * pf[i] + 0.25f * cf[i];
            const uint4 pk = hblend(v, exact0);
            if (ok) up_store(dst - 2 * (2 * W), pk);
        }
        // output row 2r: source rows (r - 1, r), weights (.25, .75); r = 0: clamped, weights (0, 1)
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = rfirst ? cf[i] : 0.25f * pf[i] + 0.75f * cf[i];
        const uint4 pk = hblend(v, exact0);
        if (ok) up_store(dst, pk);
        if (rlast) {
            // output row 2H - 1: source index clamped at H - 1 on both sides -> .75 C + .25 C
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = 0.75f * cf[i] + 0.25f * cf[i];
            const uint4 pl2 = hblend(v, exact0);
            if (ok) up_store(dst + 2 * (2 * W), pl2);
        }
    };
    int cur = 0, cur_b = -1;
    for (int gr = R0 - halo; gr < R1; ++gr) {
        const int b = gr / H, r = gr - b * H;
        const bool silent = gr < R0 || UP2_DBG(4), first = r == 0, last = r == H - 1;
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

        // the window of half EH of this image row: output columns 64 EH .. 64 EH + 63 of output rows 2r - 1 (unless r = 0), 2r and,
        // on the last image row, 2H - 1; needs halves EH - 1 (last column) and EH + 1 (first column) in their slots
        auto window = [&](auto eh_tag) {
            constexpr int EH = decltype(eh_tag)::value;
            constexpr int SB = (EH % 3) * 64;                                            // byte offset of the half's slot in a patch row
            // neighbours: lane ss = 0 takes the last column of half EH - 1 (image border: its own first column, the value is unused),
            // lane ss = 7 the first column of half EH + 1 (image border: its own last column = the clamped source index)
            constexpr int LEFT_IN = SB - 4, LEFT_EDGE = EH > 0 ? ((EH - 1) % 3) * 64 + 60 : SB;          // the dword holding the neighbour
            constexpr int RIGHT_IN = SB + 8, RIGHT_EDGE = EH < NH - 1 ? ((EH + 1) % 3) * 64 - 56 : SB + 4;  // (relative to the lane's 8 ss)
            // three address registers, everything else immediate offsets (laundered: the optimiser otherwise keeps one address per
            // (half, query-row group) alive across the row loop -- 60 registers' worth, spilled to scratch)
            uint32_t pa = pr_addr;
            int ssl = ss;
            asm volatile("" : "+v"(pa), "+v"(ssl));
            const uint32_t la = pa + (uint32_t)(ssl == 0 ? LEFT_EDGE : LEFT_IN);
            const uint32_t ra = pa + (uint32_t)(ssl == 7 ? RIGHT_EDGE : RIGHT_IN);
            const bool left_hi = !(EH == 0 && ss == 0);                                  // which half of the dword is the neighbour
            const bool right_hi = EH == NH - 1 && ss == 7;
            char* ub = up_row + 2 * 64 * EH;
            auto pair = [&](auto kp_tag) {
                constexpr int KP = decltype(kp_tag)::value;
                constexpr int K0 = (2 * KP) * 8 * ROWB, K1 = (2 * KP + 1) * 8 * ROWB;
                const uint32_t pl0 = lds_read32u_asm<K0>(la), pl1 = lds_read32u_asm<K1>(la);
                const u32x2_t pb0 = lds_read64_asm<SB + K0>(pa), pb1 = lds_read64_asm<SB + K1>(pa);
                const uint32_t pr0 = lds_read32u_asm<K0>(ra), pr1 = lds_read32u_asm<K1>(ra);
                const uint32_t cl0 = lds_read32u_asm<C::PATCH1 + K0>(la), cl1 = lds_read32u_asm<C::PATCH1 + K1>(la);
                const u32x2_t cb0 = lds_read64_asm<SB + C::PATCH1 + K0>(pa), cb1 = lds_read64_asm<SB + C::PATCH1 + K1>(pa);
                const uint32_t cr0 = lds_read32u_asm<C::PATCH1 + K0>(ra), cr1 = lds_read32u_asm<C::PATCH1 + K1>(ra);
                lds_wait_all();
                blend_store(pl0, pb0, pr0, cl0, cb0, cr0, left_hi, right_hi, EH == 0 && ss == 0,
                            rt * 32 + sq + 8 * (2 * KP) < N && !UP2_DBG(1), first, last, ub + (2 * KP) * up_kk_bytes + up_lane_off);
                blend_store(pl1, pb1, pr1, cl1, cb1, cr1, left_hi, right_hi, EH == 0 && ss == 0,
                            rt * 32 + sq + 8 * (2 * KP + 1) < N && !UP2_DBG(1), first, last, ub + (2 * KP + 1) * up_kk_bytes + up_lane_off);
            };
            pair(std::integral_constant<int, 0>{});
            pair(std::integral_constant<int, 1>{});
        };

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
                    const uint32_t fa = lds0 + cur * C::TILEB + frag_off[h];
                    f32x16_t acc;
                    constexpr int KB = 2;
                    u32x2_t bq[2][1][KB][2];
                    conv_read_batch<1, KB, 0>(fa, bq[0]);
                    conv_batches<1, 1, E, KB, 0, true>(fa, af, bq, acc, bias);
                    // the low-resolution logits as the 16-bit values the API returns: everything below blends THOSE (exactly what
                    // ph_upsample2x reads in the two-kernel form); fp32 arithmetic on them, one rounding at the end
                    uint32_t cu[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) cu[j] = f2e_pk<EO>(acc[2 * j], acc[2 * j + 1]);
                    if (!silent) {
                        auto body = [&](auto hh_tag) {
                            constexpr int HH = decltype(hh_tag)::value;
                            constexpr int SB = (HH % 3) * 64;
                            up_patch_put<ROWB, SB>(pw_addr, prev[TC][h]);                       // P: the previous image row
                            up_patch_put<ROWB, SB + C::PATCH1>(pw_addr, cu);                    // C: this one
                            lds_wait_all();
#ifdef UP2_ABL_SKIP_RT           // timing only (wrong results): the window passes of ONE row block left out -- which SIMD bounds the kernel?
                            if (rt != UP2_ABL_SKIP_RT)
#endif
                            if (!UP2_DBG(2)) {
                                if constexpr (HH > 0) window(std::integral_constant<int, HH - 1>{});
                                if constexpr (HH == NH - 1) window(std::integral_constant<int, HH>{});
                            }
                            if constexpr (LOWRES && (HH & 1) == 1) {
                                // the tile's low-resolution logits: 64 pixels = one 128-byte line per query row, halves HH - 1 | HH
                                constexpr int S0 = ((HH - 1) % 3) * 64, S1 = SB;
                                uint32_t a = pr_addr + (uint32_t)(ss < 4 ? S0 + 8 * ss : S1 + 8 * ss - 64);     // 16 ss bytes in all
                                asm volatile("" : "+v"(a));
                                u32x4_t x[4];
                                x[0] = lds_read128o_asm<C::PATCH1 + 0 * 8 * ROWB>(a);
                                x[1] = lds_read128o_asm<C::PATCH1 + 1 * 8 * ROWB>(a);
                                x[2] = lds_read128o_asm<C::PATCH1 + 2 * 8 * ROWB>(a);
                                x[3] = lds_read128o_asm<C::PATCH1 + 3 * 8 * ROWB>(a);
                                lds_wait_all();
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk)     // default (cached) stores: a caller reads these next
                                    if (rt * 32 + sq + 8 * kk < N)
                                        *(uint4*)(lr_row + 2 * 64 * TC + kk * lr_kk_bytes + lr_lane_off) = __builtin_bit_cast(uint4, x[kk]);
                            }
                        };
                        if (h == 0) body(std::integral_constant<int, 2 * TC>{});
                        else body(std::integral_constant<int, 2 * TC + 1>{});
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) prev[TC][h][j] = cu[j];
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
    }
}


// ====================================================================================================================
// Round 6: the x2 upsample as a SECOND MFMA product instead of a VALU window pass (`k_dynconv_up2m`).
//
// Where round 4's kernel (above) spends its time: per 32-pixel half a consumer wave issues 16 MFMAs and then ~440 VALU instructions,
// 64 `ds_write_b16` and 24 LDS reads to blend 4 source columns per lane (the window pass) -- MFMA busy 0.10, and the SIMD that hosts
// two of the five consumer waves serialises 2 x that instruction stream (same-box ablations).  Bilinear x2 is LINEAR in
// the low-resolution logits, so both directions fold into one small GEMM per half:
//
//     out[opx][q] = sum_px  U[opx][px] * ( wP * P[px][q] + wC * C[px][q] )          P / C = previous / current image row, 16-bit logits
//
// with U the 64 x 34 horizontal interpolation matrix (two non-zeros per row: .75 / .25) and (wP, wC) = (.75, .25) for output row 2r - 1,
// (.25, .75) for row 2r.  The first product is computed TRANSPOSED (operands swapped: D'[px][q], a lane = one query, its 16 registers =
// 16 of the half's 32 pixels), so that its packed 16-bit results ARE B fragments of the second product -- k-slot e of lane (q, g) is
// pixel (e & 3) + 8 (e >> 2) + 4 g of a 16-pixel k-step, and the constant A fragments (U with the vertical weight folded in; every
// product of two weights is exact in fp16 / bf16) are built with the same permutation.  No data moves between the two products.  A
// 32-column output tile depends on its 16 source pixels + one neighbour on each side: the neighbours of both tiles of a half (pixels
// -1, 16 / 15, 32) travel in a third, 4-slot k-step ("edge step"), gathered with two cross-half-lane moves.  Per half: 16 + 12 MFMAs and
// ~100 VALU instructions; what remains of the LDS traffic is the transposition of the finished tiles for whole-line stores
// (8-byte writes, 16-byte reads, per wave, no barrier).
//
// Same values as the window pass up to fp32 summation order: fp32 accumulation of exact products of the 16-bit logits with exact
// weights, one final rounding (tests/test_gpu_kernels.py::test_dynconv_up2_fused_final_stage: within one 16-bit ulp of F.interpolate).
// Geometry, ring, producer wave, vertical direction (previous row in registers), silent halo row: as above.  Horizontal direction:
// half hh is emitted when half hh + 1 has been computed (its first pixel is hh's right neighbour); the last half of an image row
// clamps.  The low-resolution logits of the mask branch go through the same transposition.
template <int NRT> struct UpmCfg {
    static constexpr int NW = NRT + 1;
    static constexpr int TILEB = 256 * CONV_T * 2;
    static constexpr int ROWT = 128 + 16;                        // bytes per row of a transposition buffer: 64 columns + 16 B padding
    static constexpr int TB = 32 * ROWT;                         // one buffer: [32 q][64 columns] 16-bit
    static constexpr int PATCHB = NRT * 2 * TB;                  // per consumer wave: upsampled rows | low-resolution rows
    static constexpr int KBB = NRT * 32 * 4;
    static constexpr int CONSTB = 6 * 1024;                      // the second product's six constant A fragments (64 lanes x 16 B each)
    static constexpr int NBUF = 3 * TILEB + PATCHB + KBB + CONSTB <= 160 * 1024 ? 3 : 2;
    static constexpr int LDSB = NBUF * TILEB + PATCHB + KBB + CONSTB;
};

template <int OFF> __device__ __forceinline__ void lds_write64_asm(uint32_t byte_addr, uint32_t lo, uint32_t hi) {
    const u32x2_t v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(byte_addr), "v"(v), "n"(OFF) : "memory");
}

// constant A fragment of the second product for this lane (output column `opx` = lane & 31 of a 32-column tile, k group g): slot e <->
// source pixel (e & 3) + 8 (e >> 2) + 4 g of the tile's 16; weight w * U[opx][px]
template <int EO> __device__ __forceinline__ uint4 upm_main_frag(int lane, float w) {
    const int opx = lane & 31, g = lane >> 5, j = opx >> 1;
    uint32_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int px = (e & 3) + 8 * (e >> 2) + 4 * g;
        float u = 0.f;
        if (px == j) u = 0.75f;
        else if ((opx & 1) ? px == j + 1 : px == j - 1) u = 0.25f;
        h[e] = f2e<EO>(w * u);
    }
    return make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
}
// ... and of the edge step: slots 0 / 1 = left / right neighbour of the P row, 2 / 3 of the C row (k group 0 only); the left neighbour
// feeds output column 0, the right one column 31, each with horizontal weight .25
template <int EO> __device__ __forceinline__ uint4 upm_edge_frag(int lane, float wP, float wC) {
    const int opx = lane & 31, g = lane >> 5;
    const float l = (g == 0 && opx == 0) ? 0.25f : 0.f, r = (g == 0 && opx == 31) ? 0.25f : 0.f;
    return make_uint4(pack2(f2e<EO>(wP * l), f2e<EO>(wP * r)), pack2(f2e<EO>(wC * l), f2e<EO>(wC * r)), 0u, 0u);
}

template <int E, int NRT, int NTR, bool LOWRES, typename OutT>
__global__ __launch_bounds__(((NRT + 1) * 64)) void k_dynconv_up2m(const uint16_t* __restrict__ planes, const uint16_t* __restrict__ kern,
                                                                  int64_t kern_batch_stride, const float* __restrict__ kbias,
                                                                  int64_t kbias_batch_stride, OutT* __restrict__ logits_out,
                                                                  OutT* __restrict__ up_out, int B, int N, int H, int dbg) {
    using C = UpmCfg<NRT>;
    constexpr int NBUF = C::NBUF, W = NTR * 64, ROWT = C::ROWT, EO = UpElem<OutT>::E, NH = 2 * NTR;
    constexpr bool COOP = E == PH_E_F16_FROM_BF16;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];   // [NBUF][256][64] | per wave [up T | low-res T2] | biases [NRT][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave == NRT;
    const int rt = producer ? 0 : wave;
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1, ql = lane & 31;
    const int64_t HW = (int64_t)H * W;
    const int ntiles = H * NTR;

    const int64_t rows_total = (int64_t)B * H;
    const int R0 = (int)(rows_total * blockIdx.x / gridDim.x), R1 = (int)(rows_total * (blockIdx.x + 1) / gridDim.x);
    if (R0 >= R1) return;
    const int halo = (R0 % H) != 0 ? 1 : 0;
    const int tg0 = (R0 - halo) * NTR, tg1 = R1 * NTR;

    // ---- producer state (as k_dynconv_up2)
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
    const uint32_t tbuf = lds0 + NBUF * C::TILEB + rt * 2 * C::TB;                  // upsampled rows; + TB: the low-resolution rows
    float* kb_lds = (float*)((unsigned char*)lds + NBUF * C::TILEB + C::PATCHB) + rt * 32;
    const uint32_t kb_addr = lds_addr(kb_lds) + 4 * ql;
    // writes: lane (q, g), quad i of a 32-column tile = columns 8 i + 4 g .. + 3 (8 bytes); reads: lane (sq, ss) = 16 bytes of row sq + 8 kk
    const uint32_t tw_addr = tbuf + (uint32_t)(ql * ROWT + 8 * g);
    const int sq = lane >> 3, ss = lane & 7;
    const uint32_t tr_addr = tbuf + (uint32_t)(sq * ROWT + 16 * ss);
    const int64_t up_plane = (int64_t)2 * H * 2 * W;
    const uint32_t up_lane_off = 2u * (uint32_t)(sq * up_plane + 8 * ss);
    const int64_t up_kk_bytes = 2 * 8 * up_plane;
    const uint32_t lr_lane_off = 2u * (uint32_t)(sq * HW + 8 * ss);
    const int64_t lr_kk_bytes = 2 * 8 * HW;

    // constant A fragments of the second product, in LDS (one 16-byte slot per lane and fragment; every wave reads its own lane's):
    // 0 / 1: .75 U / .25 U (P / C of output row 2r - 1, C / P of row 2r), 2 / 3: the edge steps of those two rows, 4 / 5: U and the
    // C-only edge step (output row 0 of a frame and its last row: the source row index is clamped)
    const uint32_t cfrag = lds0 + NBUF * C::TILEB + C::PATCHB + C::KBB + 16u * (uint32_t)lane;
    if (wave == 0) {
        uint4* cw = (uint4*)((unsigned char*)lds + NBUF * C::TILEB + C::PATCHB + C::KBB) + lane;
        cw[0 * 64] = upm_main_frag<EO>(lane, 0.75f);
        cw[1 * 64] = upm_main_frag<EO>(lane, 0.25f);
        cw[2 * 64] = upm_edge_frag<EO>(lane, 0.75f, 0.25f);
        cw[3 * 64] = upm_edge_frag<EO>(lane, 0.25f, 0.75f);
        cw[4 * 64] = upm_main_frag<EO>(lane, 1.0f);
        cw[5 * 64] = upm_edge_frag<EO>(lane, 0.f, 1.0f);
    }
    __syncthreads();                                 // (before the ring's first counted wait: drains the producer's prologue DMA once)

    uint4 af[1][16];
    // the previous image row, packed 16-bit pairs in the transposed accumulator layout: [tile][half][k-step of the second product]
    uint4 prev[NTR][2][2];
#pragma unroll
    for (int a = 0; a < NTR; ++a)
#pragma unroll
        for (int h = 0; h < 2; ++h) { prev[a][h][0] = make_uint4(0, 0, 0, 0); prev[a][h][1] = make_uint4(0, 0, 0, 0); }
    // the half that waits for its right neighbour: its C row (its P row is still in `prev`, which takes the C row only after the
    // emission); Pe / Ce: register 7 of the half before that one (its pixel 31 = the g = 1 lanes' high half)
    uint4 Cp0 = make_uint4(0, 0, 0, 0), Cp1 = Cp0;
    uint32_t Pe = 0, Ce = 0;

    auto mf = [&](uint4 a, uint4 b, f32x16_t c) -> f32x16_t {
        if constexpr (EO == PH_E_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
        else return mfma32(a, b, c);
    };
    auto xg = [&](uint32_t v) -> uint32_t { return (uint32_t)__shfl_xor((int)v, 32); };
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // one output row of a half: its two 32-column tiles one after the other (16 accumulator registers), transposed through the wave's
    // buffer, stored as whole 128-byte lines.  MP / MC / ME: indices of the constant fragments (P rows, C rows, edge step); MP < 0: C only
    auto out_row = [&](auto mp_tag, auto mc_tag, auto me_tag, uint4 P0, uint4 P1, uint4 C0, uint4 C1, uint4 e0, uint4 e1, char* dst_row) {
        constexpr int MP = decltype(mp_tag)::value, MC = decltype(mc_tag)::value, ME = decltype(me_tag)::value;
        uint32_t cf = cfrag;
        asm volatile("" : "+v"(cf));
        u32x4_t kc = lds_read128o_asm<MC * 1024>(cf), ke = lds_read128o_asm<ME * 1024>(cf), kp = kc;
        if constexpr (MP >= 0) kp = lds_read128o_asm<MP * 1024>(cf);
        lds_wait_all();
        const uint4 KC = __builtin_bit_cast(uint4, kc), KE = __builtin_bit_cast(uint4, ke), KP = __builtin_bit_cast(uint4, kp);
#ifndef UPM_SERIAL_TILES
        // the two tiles' chains interleaved (each MFMA waits for its own predecessor only: one other product in between)
        f32x16_t a0 = zero16, a1 = zero16;
        if constexpr (MP >= 0) { a0 = mf(KP, P0, a0); a1 = mf(KP, P1, a1); }
        a0 = mf(KC, C0, a0); a1 = mf(KC, C1, a1);
        a0 = mf(KE, e0, a0); a1 = mf(KE, e1, a1);
        {
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = f2e_pk<EO>(a0[2 * j], a0[2 * j + 1]);
            lds_write64_asm<0>(tw_addr, w[0], w[1]);  lds_write64_asm<16>(tw_addr, w[2], w[3]);
            lds_write64_asm<32>(tw_addr, w[4], w[5]); lds_write64_asm<48>(tw_addr, w[6], w[7]);
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = f2e_pk<EO>(a1[2 * j], a1[2 * j + 1]);
            lds_write64_asm<64>(tw_addr, w[0], w[1]); lds_write64_asm<80>(tw_addr, w[2], w[3]);
            lds_write64_asm<96>(tw_addr, w[4], w[5]); lds_write64_asm<112>(tw_addr, w[6], w[7]);
        }
#else
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16_t a = zero16;
            if constexpr (MP >= 0) a = mf(KP, t == 0 ? P0 : P1, a);
            a = mf(KC, t == 0 ? C0 : C1, a);
            a = mf(KE, t == 0 ? e0 : e1, a);
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = f2e_pk<EO>(a[2 * j], a[2 * j + 1]);
            if (t == 0) {
                lds_write64_asm<0>(tw_addr, w[0], w[1]);  lds_write64_asm<16>(tw_addr, w[2], w[3]);
                lds_write64_asm<32>(tw_addr, w[4], w[5]); lds_write64_asm<48>(tw_addr, w[6], w[7]);
            } else {
                lds_write64_asm<64>(tw_addr, w[0], w[1]); lds_write64_asm<80>(tw_addr, w[2], w[3]);
                lds_write64_asm<96>(tw_addr, w[4], w[5]); lds_write64_asm<112>(tw_addr, w[6], w[7]);
            }
        }
#endif
        lds_wait_all();
        uint32_t tr = tr_addr;
        asm volatile("" : "+v"(tr));
        u32x4_t x[4];
        x[0] = lds_read128o_asm<0 * 8 * ROWT>(tr); x[1] = lds_read128o_asm<1 * 8 * ROWT>(tr);
        x[2] = lds_read128o_asm<2 * 8 * ROWT>(tr); x[3] = lds_read128o_asm<3 * 8 * ROWT>(tr);
        lds_wait_all();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            if (rt * 32 + sq + 8 * kk < N && !UP2_DBG(1)) up_store(dst_row + kk * up_kk_bytes + up_lane_off, __builtin_bit_cast(uint4, x[kk]));
    };
    int cur = 0, cur_b = -1;
    for (int gr = R0 - halo; gr < R1; ++gr) {
        const int b = gr / H, r = gr - b * H;
        const bool silent = gr < R0 || UP2_DBG(4), first = r == 0, last = r == H - 1;
        if (!producer && b != cur_b) {
            const uint16_t* kr = kern + (int64_t)b * kern_batch_stride + (rt * 32 + (lane & 31)) * PH_C + g * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) af[0][ks] = *(const uint4*)(kr + ks * 16);
            if (lane < 32) kb_lds[lane] = kbias[(int64_t)b * kbias_batch_stride + rt * 32 + lane];
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
        }
        cur_b = b;
        char* up_row = (char*)(up_out + ((int64_t)b * N + rt * 32) * up_plane + (int64_t)(2 * r) * (2 * W));    // output row 2r
        char* lr_row = LOWRES ? (char*)(logits_out + ((int64_t)b * N + rt * 32) * HW + (int64_t)r * W) : nullptr;

        // emission of half EH of this image row: P / C = its packed rows, (lP, lC) = registers whose HIGH half, in the g = 1 lanes, is
        // the pixel left of it (ignored for EH = 0: clamped), (rP, rC) = registers whose LOW half, in the g = 0 lanes, is the pixel right
        // of it (ignored for EH = NH - 1: clamped)
        auto emit = [&](auto eh_tag, uint4 P0, uint4 P1, uint4 C0, uint4 C1, uint32_t lP, uint32_t lC, uint32_t rP, uint32_t rC) {
            constexpr int EH = decltype(eh_tag)::value;
            // neighbours of tile 0: pixel -1 (left) and pixel 16 (register 4, low half, g = 0 lanes); of tile 1: pixel 15 (register 3,
            // high half, g = 1 lanes) and pixel 32 (right)
            uint32_t l0P, l0C, r1P, r1C;
            if constexpr (EH == 0) { l0P = P0.x & 0xFFFFu; l0C = C0.x & 0xFFFFu; }
            else { l0P = xg(lP) >> 16; l0C = xg(lC) >> 16; }
            if constexpr (EH == NH - 1) { r1P = xg(P1.w) >> 16; r1C = xg(C1.w) >> 16; }
            else { r1P = rP & 0xFFFFu; r1C = rC & 0xFFFFu; }
            const uint32_t p15 = xg(P0.w) >> 16, c15 = xg(C0.w) >> 16;
            const uint32_t keep = g ? 0u : 0xFFFFFFFFu;                                 // the edge step lives in k group 0
            const uint4 e0 = make_uint4((l0P | (P1.x << 16)) & keep, (l0C | (C1.x << 16)) & keep, 0u, 0u);
            const uint4 e1 = make_uint4((p15 | (r1P << 16)) & keep, (c15 | (r1C << 16)) & keep, 0u, 0u);
            char* ub = up_row + 2 * 64 * EH;
            constexpr auto c0 = std::integral_constant<int, 0>{}; constexpr auto c1 = std::integral_constant<int, 1>{};
            constexpr auto c2 = std::integral_constant<int, 2>{}; constexpr auto c3 = std::integral_constant<int, 3>{};
            constexpr auto c4 = std::integral_constant<int, 4>{}; constexpr auto c5 = std::integral_constant<int, 5>{};
            constexpr auto cn = std::integral_constant<int, -1>{};
            if (!first) {
                out_row(c0, c1, c2, P0, P1, C0, C1, e0, e1, ub - 2 * (2 * W));                  // output row 2r - 1 = .75 P + .25 C
                out_row(c1, c0, c3, P0, P1, C0, C1, e0, e1, ub);                                // output row 2r     = .25 P + .75 C
            } else {
                out_row(cn, c4, c5, P0, P1, C0, C1, e0, e1, ub);                                // r = 0: output row 0 = the C row (clamped source index)
            }
            if (last) out_row(cn, c4, c5, P0, P1, C0, C1, e0, e1, ub + 2 * (2 * W));            // output row 2H - 1: clamped on both sides -> the C row
        };

        auto tile = [&](auto tc_tag) {
            constexpr int TC = decltype(tc_tag)::value;
            const int tg = gr * NTR + TC;
            if (producer) {
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
                // this lane's bias (a lane = one query in the transposed product)
                float bl;
                asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(bl) : "v"(kb_addr) : "memory");
                float bias[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) bias[q] = bl;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t fa = lds0 + cur * C::TILEB + frag_off[h];
                    f32x16_t acc;
                    constexpr int KB = 2;
                    u32x2_t bq[2][1][KB][2];
                    conv_read_batch<1, KB, 0>(fa, bq[0]);
                    conv_batches<1, 1, E, KB, 0, true, true>(fa, af, bq, acc, bias);          // SWAP: D'[pixel][query]
                    const uint4 cu0 = make_uint4(f2e_pk<EO>(acc[0], acc[1]), f2e_pk<EO>(acc[2], acc[3]), f2e_pk<EO>(acc[4], acc[5]), f2e_pk<EO>(acc[6], acc[7]));
                    const uint4 cu1 = make_uint4(f2e_pk<EO>(acc[8], acc[9]), f2e_pk<EO>(acc[10], acc[11]), f2e_pk<EO>(acc[12], acc[13]), f2e_pk<EO>(acc[14], acc[15]));
                    if (!silent) {
                        auto body = [&](auto hh_tag) {
                            constexpr int HH = decltype(hh_tag)::value;
                            if constexpr (LOWRES) {
                                // the tile's low-resolution logits through the second buffer: quads of this half, then (second half) whole lines
                                asm volatile("s_nop 7" ::: "memory");
                                constexpr int HB = (HH & 1) * 64;
                                lds_write64_asm<C::TB + HB + 0>(tw_addr, cu0.x, cu0.y);  lds_write64_asm<C::TB + HB + 16>(tw_addr, cu0.z, cu0.w);
                                lds_write64_asm<C::TB + HB + 32>(tw_addr, cu1.x, cu1.y); lds_write64_asm<C::TB + HB + 48>(tw_addr, cu1.z, cu1.w);
                                if constexpr ((HH & 1) == 1) {
                                    lds_wait_all();
                                    u32x4_t x[4];
                                    x[0] = lds_read128o_asm<C::TB + 0 * 8 * ROWT>(tr_addr); x[1] = lds_read128o_asm<C::TB + 1 * 8 * ROWT>(tr_addr);
                                    x[2] = lds_read128o_asm<C::TB + 2 * 8 * ROWT>(tr_addr); x[3] = lds_read128o_asm<C::TB + 3 * 8 * ROWT>(tr_addr);
                                    lds_wait_all();
#pragma unroll
                                    for (int kk = 0; kk < 4; ++kk)     // default (cached) stores: a caller reads these next
                                        if (rt * 32 + sq + 8 * kk < N)
                                            *(uint4*)(lr_row + 2 * 64 * TC + kk * lr_kk_bytes + lr_lane_off) = __builtin_bit_cast(uint4, x[kk]);
                                }
                            }
                            // the half before this one (HH - 1): `prev` of ITS position still holds its P row
                            constexpr int TCp = (HH > 0) ? (HH - 1) / 2 : 0, hp = (HH > 0) ? (HH - 1) % 2 : 0;
                            if constexpr (HH > 0) {
                                if (!UP2_DBG(2))
                                    emit(std::integral_constant<int, HH - 1>{}, prev[TCp][hp][0], prev[TCp][hp][1], Cp0, Cp1, Pe, Ce, prev[TC][h][0].x, cu0.x);
                                Pe = prev[TCp][hp][1].w; Ce = Cp1.w;             // its pixel 31: the left neighbour of half HH
                                prev[TCp][hp][0] = Cp0; prev[TCp][hp][1] = Cp1;  // ... and now it becomes the next image row's P row
                            }
                            if constexpr (HH == NH - 1) {
                                if (!UP2_DBG(2)) emit(std::integral_constant<int, HH>{}, prev[TC][h][0], prev[TC][h][1], cu0, cu1, Pe, Ce, 0u, 0u);
                                prev[TC][h][0] = cu0; prev[TC][h][1] = cu1;
                            } else {
                                Cp0 = cu0; Cp1 = cu1;                            // waits for its right neighbour
                            }
                        };
                        if (h == 0) body(std::integral_constant<int, 2 * TC>{});
                        else body(std::integral_constant<int, 2 * TC + 1>{});
                    } else {
                        prev[TC][h][0] = cu0; prev[TC][h][1] = cu1;                 // silent halo row: nothing is emitted, only remembered
                    }
                }
            }
            cur = cur + 1 == NBUF ? 0 : cur + 1;
        };
        tile(std::integral_constant<int, 0>{});
        if constexpr (NTR > 1) tile(std::integral_constant<int, 1>{});
        if constexpr (NTR > 2) tile(std::integral_constant<int, 2>{});
        if constexpr (NTR > 3) tile(std::integral_constant<int, 3>{});
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
                      void* up_out, int B, int N, int H, int want_wgs, hipStream_t s) {
    constexpr int NTR = 4;
    const int64_t rows = (int64_t)B * H;
    int wgs = want_wgs > 0 ? want_wgs : up_num_cus();
    // test knob, read per launch on purpose (tests/test_gpu_kernels.py switches it inside one process; an eager launch pays one
    // environment look-up, a graph replay none)
    if (const char* e = getenv("PH_UP2_WGS")) wgs = atoi(e) > 0 ? atoi(e) : wgs;
    if (wgs > rows) wgs = (int)rows;
    const dim3 grid(wgs), block((NRT + 1) * 64);
    static const int dbg = [] { const char* e = getenv("PH_UP2_DBG"); return e ? atoi(e) : 0; }();     // timing experiments only, read once
    // round 6: the upsample as a second MFMA product (k_dynconv_up2m); PH_UP2_MFMA=0: round 4's window-pass kernel (A/B timing, tests)
    bool mfma_form = true;
    if (const char* e = getenv("PH_UP2_MFMA")) mfma_form = atoi(e) != 0;
    if (mfma_form) {
        constexpr int ldsm = UpmCfg<NRT>::LDSB;
#define UPM_GO(LR)                                                                                                            \
    do {                                                                                                                      \
        static const bool once = [] {                                                                                         \
            (void)hipFuncSetAttribute((const void*)k_dynconv_up2m<E, NRT, NTR, LR, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsm); \
            return true;                                                                                                      \
        }();                                                                                                                  \
        (void)once;                                                                                                           \
        hipLaunchKernelGGL((k_dynconv_up2m<E, NRT, NTR, LR, OutT>), grid, block, ldsm, s, planes, kern, kbs, kbias, bbs, (OutT*)logits_out, \
                           (OutT*)up_out, B, N, H, dbg);                                                                           \
    } while (0)
        if (logits_out) UPM_GO(true);
        else UPM_GO(false);
#undef UPM_GO
        return;
    }
    constexpr int lds = UpCfg<NRT>::LDSB;
#define UP_GO(LR)                                                                                                             \
    do {                                                                                                                      \
        static const bool once = [] {                                                                                         \
            (void)hipFuncSetAttribute((const void*)k_dynconv_up2<E, NRT, NTR, LR, OutT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            return true;                                                                                                      \
        }();                                                                                                                  \
        (void)once;                                                                                                           \
        hipLaunchKernelGGL((k_dynconv_up2<E, NRT, NTR, LR, OutT>), grid, block, lds, s, planes, kern, kbs, kbias, bbs, (OutT*)logits_out, \
                           (OutT*)up_out, B, N, H, dbg);                                                                           \
    } while (0)
    if (logits_out) UP_GO(true);
    else UP_GO(false);
#undef UP_GO
}

// `workgroups`: 0 = one per CU (each a contiguous range of image rows).  A launch that has the GPU to itself is fastest that way; a
// launch that SHARES it (the multi-stream step: other parts' query kernels hold CUs when it starts) ends sooner with 1.5 per CU --
// workgroups that could not start at once leave a shorter tail: +0.9 % on the 128-frame step, 0.54 against 0.48 ms alone
// (profiles/r06/knob_sweep.txt).  Same values whatever the count (rows are independent; tests/test_gpu_kernels.py sweeps it).
extern "C" int ph_dynconv_up2_wgs(const uint16_t* planes, const uint16_t* kern, int64_t kern_batch_stride, const float* kbias,
                                  int64_t kbias_batch_stride, void* logits_out, void* up_out, int out_dtype, int B, int N, int H, int W,
                                  int prec, int workgroups, void* stream) {
    PH_CHECK_ARG(planes && kern && kbias && up_out && B > 0 && workgroups >= 0, "bad pointer or size");
    if (!ph_dynconv_up2_supported(N, H, W, prec, out_dtype)) {
        ph_set_error("ph_dynconv_up2: unsupported geometry or arithmetic (use ph_dynconv + ph_upsample2x)");
        return PH_EUNSUPPORTED;
    }
    const int nrt = ph_n_padded(N) / 32;
    hipStream_t s = (hipStream_t)stream;
#define UP_ARGS planes, kern, kern_batch_stride, kbias, kbias_batch_stride, logits_out, up_out, B, N, H, workgroups, s
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

extern "C" int ph_dynconv_up2(const uint16_t* planes, const uint16_t* kern, int64_t kern_batch_stride, const float* kbias,
                              int64_t kbias_batch_stride, void* logits_out, void* up_out, int out_dtype, int B, int N, int H, int W,
                              int prec, void* stream) {
    return ph_dynconv_up2_wgs(planes, kern, kern_batch_stride, kbias, kbias_batch_stride, logits_out, up_out, out_dtype, B, N, H, W, prec, 0,
                              stream);
}
