// Device helpers shared by the dynamic-convolution kernels (ph_conv.hip: k_dynconv; ph_convup.hip: k_dynconv_up2): the LDS
// accesses of the tile loop as inline asm, the swizzle of the LDS-DMA tile image, the MFMA phase of one 32-pixel half.
#pragma once
#include "ph_common.h"

constexpr int CONV_T = 64;            // pixels per tile: one 128-byte line per channel row

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// ---- LDS accesses of the tile loop, hidden from the compiler -------------------------------------------
// hipcc waits vmcnt(0) before any DS access it can see while an LDS-DMA that may alias it is in flight
// (SIInsertWaitcnts), which would drain the ring every tile.  Ordering is done by hand: counted vmcnt + raw
// s_barrier before the first read of a tile, counted lgkmcnt before the first use of a read.
template <int OFF> __device__ __forceinline__ u32x2_t lds_tr16_asm(uint32_t byte_addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ u32x4_t lds_read128_asm(uint32_t byte_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
template <int OFF> __device__ __forceinline__ void lds_write_asm(uint32_t byte_addr, float v, float*) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(byte_addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write_asm(uint32_t byte_addr, float v, uint16_t*) {
    const uint32_t h = f2bf(v);
    asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(byte_addr), "v"(h), "n"(OFF) : "memory");
}
struct ph_h16 { uint16_t v; };    // fp16 output element (PH_OUT_F16); uint16_t = bf16 output (PH_OUT_BF16)
template <int OFF> __device__ __forceinline__ void lds_write_asm(uint32_t byte_addr, float v, ph_h16*) {
    const uint32_t h = f2h(v);
    asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(byte_addr), "v"(h), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_write128_asm(uint32_t byte_addr, u32x4_t v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(byte_addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(PH_LDS const void*)p; }

__device__ __forceinline__ void st_out(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_out(uint16_t* p, float v) { *p = (uint16_t)f2bf(v); }
__device__ __forceinline__ void st_out(ph_h16* p, float v) { p->v = (uint16_t)f2h(v); }

// LDS image of a tile: [256 rows][8 x 16-byte pieces], rows contiguous (128 B) because the tile is
// written by LDS-DMA (global_load_lds: wave-uniform base + lane*16, no padding possible).  To keep the
// transposing reads conflict free the piece index is XOR-swizzled with bit 1 of the row on the SOURCE
// side (same 128-byte line, so coalescing is unchanged) and the same XOR is applied by the readers:
// the 4 rows x 2 half-tiles a 32-lane read touches then cover 8 distinct 32-byte bank windows.
__device__ __forceinline__ int conv_swz(int row) { return ((row >> 1) & 1) << 2; }


// ---- MFMA phase of one 32-pixel half: 16 k-steps, B fragments by transposing reads, KB k-steps per batch, the
// reads of batch i+1 in flight while the MFMAs of batch i run.  Compile-time recursion (immediate offsets).
template <int PF, int KB, int BI, int K = 0, int P = 0>
__device__ __forceinline__ void conv_read_batch(uint32_t fa, u32x2_t (&dst)[PF][KB][2]) {
    if constexpr (K < KB) {
        constexpr int OFF = P * (256 * CONV_T * 2) + (BI * KB + K) * 2048;     // plane, k-step (16 rows x 128 B)
        dst[P][K][0] = lds_tr16_asm<OFF>(fa);
        dst[P][K][1] = lds_tr16_asm<OFF + 4 * CONV_T * 2>(fa);                 // 4 rows below
        if constexpr (P + 1 < PF) conv_read_batch<PF, KB, BI, K, P + 1>(fa, dst);
        else conv_read_batch<PF, KB, BI, K + 1, 0>(fa, dst);
    }
}

// PF feature planes x PK kernel planes: (1,1) a.b; (1,2) (a_hi + a_lo).b; (2,2) a_hi.b_hi + a_hi.b_lo + a_lo.b_hi
// SWAP (one plane each): the operands exchanged -- D'[pixel][query], the transposed product (ph_convup.hip: k_dynconv_up2m)
template <int PF, int PK, int E, int KB, int BI, bool COOP = false, bool SWAP = false>
__device__ __forceinline__ void conv_batches(uint32_t fa, const uint4 (&af)[PK][16], u32x2_t (&bq)[2][PF][KB][2], f32x16_t& acc,
                                             const float (&bias)[16]) {
    constexpr int NBATCH = 16 / KB;
    if constexpr (BI < NBATCH) {
        if constexpr (BI + 1 < NBATCH) {
            conv_read_batch<PF, KB, BI + 1>(fa, bq[(BI + 1) & 1]);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * PF * KB) : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BI == 0) {   // the accumulator starts from the bias
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias[r];
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            uint4 bf[PF];
#pragma unroll
            for (int p = 0; p < PF; ++p)
            {
                bf[p] = make_uint4(bq[BI & 1][p][k][0].x, bq[BI & 1][p][k][0].y, bq[BI & 1][p][k][1].x, bq[BI & 1][p][k][1].y);
                if constexpr (E == PH_E_F16_FROM_BF16 && !COOP) bf[p] = bf2h_x8(bf[p]);      // 12 VALU ops under the previous MFMA
            }
            if constexpr (SWAP) {
                static_assert(!SWAP || (PF == 1 && PK == 1), "transposed product: one plane each");
                acc = mfma32e<E>(bf[0], af[0][BI * KB + k], acc);
            } else
                acc = mfma32e<E>(af[0][BI * KB + k], bf[0], acc);
            if (PF == 2) acc = mfma32e<E>(af[0][BI * KB + k], bf[PF - 1], acc);
            if (PK == 2) acc = mfma32e<E>(af[PK - 1][BI * KB + k], bf[0], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        conv_batches<PF, PK, E, KB, BI + 1, COOP, SWAP>(fa, af, bq, acc, bias);
    }
}

// The 16 biases of a lane's C-layout rows (rr + 4g, rr = (r & 3) + 8 (r >> 2)) from the wave's LDS copy.  Reads AND
// their wait are ONE asm statement: the values are later copied into the accumulator tuple by compiler-generated
// moves, which must never run ahead of a hand-placed s_waitcnt.
__device__ __forceinline__ void conv_bias_get(uint32_t addr, float (&b)[16]) {
    asm volatile(
        "ds_read_b32 %0, %16\n ds_read_b32 %1, %16 offset:4\n ds_read_b32 %2, %16 offset:8\n ds_read_b32 %3, %16 offset:12\n"
        "ds_read_b32 %4, %16 offset:32\n ds_read_b32 %5, %16 offset:36\n ds_read_b32 %6, %16 offset:40\n ds_read_b32 %7, %16 offset:44\n"
        "ds_read_b32 %8, %16 offset:64\n ds_read_b32 %9, %16 offset:68\n ds_read_b32 %10, %16 offset:72\n ds_read_b32 %11, %16 offset:76\n"
        "ds_read_b32 %12, %16 offset:96\n ds_read_b32 %13, %16 offset:100\n ds_read_b32 %14, %16 offset:104\n ds_read_b32 %15, %16 offset:108\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7]), "=&v"(b[8]),
          "=&v"(b[9]), "=&v"(b[10]), "=&v"(b[11]), "=&v"(b[12]), "=&v"(b[13]), "=&v"(b[14]), "=&v"(b[15])
        : "v"(addr)
        : "memory");
}

// patch writes of the 16 C-layout registers of a lane (rows rr + 4g), immediate row offsets
template <typename OutT, int LD, int R = 0>
__device__ __forceinline__ void conv_patch_put(uint32_t wa, const f32x16_t& acc) {
    if constexpr (R < 16) {
        constexpr int rr = (R & 3) + 8 * (R >> 2);
        lds_write_asm<rr * LD * (int)sizeof(OutT)>(wa, acc[R], (OutT*)nullptr);
        conv_patch_put<OutT, LD, R + 1>(wa, acc);
    }
}

