// A7 -- masked pooling  u[n][c] = sum_hw M[n][hw] * feat[c][hw]   (kernel_update_head.py:241-242)
//
// NT GEMM with M = Npad query rows, N = 256 channels per map, K = HW pixels, split-K over
// `nsplit` pixel ranges.  One workgroup = (pixel range, 128-channel group, frame); its 4 waves own
// 32 channels each (one 32x32x16 MFMA column tile) and ALL query rows:
//   B operand (features): the [128 ch][64 px] tile goes HBM -> LDS by LDS-DMA in whole 128-byte lines
//     (global_load_lds_dwordx4 with the nt|sc1 stream policy, a POOL_NBUF = 4 deep ring: three chunks in flight
//     while one is consumed, counted vmcnt + raw s_barrier, XOR-swizzled on the source side) and is read back
//     as fragments with ds_read_b128: lane (channel j, half g), step t <-> pixel 64*chunk + 32*g + 8*t + e (the
//     k-slot <-> pixel map is a free permutation as long as A agrees).  Every feature byte is read from HBM
//     exactly once.
//   A operand (mask bits -> {0,1} bf16): the mask words also arrive by LDS-DMA (round 3: the 4 words of a query
//     row for a PAIR of chunks in one 16-byte lane request, default cache policy); an A fragment is ONE ds_read_b128 from a
//     256-entry byte -> 8 x bf16 lookup table in LDS, so no expanded mask tile exists at all.  LDS per
//     workgroup at cfg2 (bf16): 4 x 16 KiB feature ring + 4 KiB table + 4 x 1.25 KiB mask words = 73 KiB
//     -> 2 workgroups per CU, 96 KiB of feature loads in flight per CU.  {0,1} is exact in bf16, so split
//     precision only doubles the feature operand.
// Measured history at cfg2, B = 24 (805 MB per launch): fragment-shaped 16-byte loads straight from HBM into
// VGPRs 3.4 TB/s; 32 B per lane at a 64 KB lane stride with a 3-deep register ring and no barriers 2.1 TB/s
// (request amplification at the L1/TA); whole-line DMA, double buffered 3.7 TB/s (70 % of wave cycles waiting on
// vmcnt/barrier with 64 KiB in flight per CU); this 4-deep ring 4.3 TB/s, with the nt|sc1 policy 4.5-4.7 TB/s
// against a 5.2-5.8 TB/s pure-read yardstick (profiles/r01).
// Roofline: HBM: 2*256*HWp*2 B of features per frame vs 2*Npad*512*HWp flop.
#include "ph_common.h"

constexpr int POOL_CHUNK = 64;                  // pixels per step: one 128-byte line per channel row
constexpr int POOL_FT = 128 * POOL_CHUNK;       // elements of one feature tile [128 ch][64 px]
constexpr int POOL_NBUF = 4;                    // LDS ring depth: NBUF-1 chunks of DMA in flight while one is consumed

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// LDS reads hidden from the compiler: hipcc waits vmcnt(0) before ANY DS read while an LDS-DMA that may alias it
// is in flight (SIInsertWaitcnts), which would drain the ring every chunk.  These reads are ordered by hand:
// counted vmcnt + raw s_barrier before the first read of a chunk, counted lgkmcnt before the first use.
__device__ __forceinline__ u32x4_t lds_read128_asm(uint32_t byte_addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
__device__ __forceinline__ uint32_t lds_read32_asm(uint32_t byte_addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(PH_LDS const void*)p; }

template <int E> __device__ __forceinline__ uint4 expand8(uint32_t byte) {
    // 8 mask bits -> 8 x {0, 1.0} in the planes' element format (bf16 0x3F80 / fp16 0x3C00)
    constexpr uint32_t ONE = E == PH_E_F16 ? 0x3C00u : 0x3F80u;
    uint32_t r[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t lo = (byte >> (2 * p)) & 1u, hi = (byte >> (2 * p + 1)) & 1u;
        r[p] = lo * ONE + hi * (ONE << 16);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

// LDS image of a feature tile: [128 channel rows][8 x 16-byte pieces], rows contiguous (written by LDS-DMA).
// Piece index XOR ((row >> 1) & 7) -- applied on the DMA *source* address and by the readers -- makes the
// ds_read_b128 of 16 consecutive channel rows at one piece index hit 16 distinct 16-byte slots.
__device__ __forceinline__ int pool_swz(int row) { return (row >> 1) & 7; }

template <int PA /*feature planes: 1 or 2*/, int NRT /*Npad/32*/, int E /*element format of the planes*/>
__global__ __launch_bounds__(256, 2) void k_pool(const uint16_t* __restrict__ xplanes, const uint16_t* __restrict__ dplanes,
                                              const uint32_t* __restrict__ bits, float* __restrict__ partial,
                                              int B, int64_t HWp, int nsplit, int bits_rows, int32_t* __restrict__ pcount) {
    constexpr int Npad = NRT * 32;
    constexpr int NBI = (Npad * 2 + 63) / 64;                         // DMA instructions for the mask words of a chunk
#ifdef POOL_NO_BITS_DMA   // timing experiment only (results are wrong): what the mask-word requests cost (8 % at cfg2, round 3)
    constexpr int NBW = 0;
#else
    constexpr int NBW = (NBI + 3) / 4;                                // ... issued per wave (at most)
#endif
    constexpr int PER_CHUNK = 4 * PA + NBW;                           // vector-memory instructions per wave per chunk ...
#ifdef POOL_DUP_BITS
    constexpr int NBREM = 0;
#else
    constexpr int NBREM = NBW ? NBI % 4 : 0;                          // ... one fewer in waves >= NBREM when NBI is no multiple of 4
#endif
    // POOL_PAIR_BITS (default): the mask words of TWO chunks travel together -- 4 words = 16 bytes per row and lane, 64 rows per
    // instruction, NBI2 instructions per pair of chunks instead of 2 NBI (cfg2: 3 instead of 10), issued with the first chunk of
    // the pair in front of that chunk's feature pieces; the words of a row land next to each other ([row][4 words], two pair
    // buffers).  The 4-byte-per-lane requests of the per-chunk form touch 32 cache lines per instruction and cost 8 % of the
    // kernel (POOL_NO_BITS_DMA).  -DPOOL_CHUNK_BITS restores the per-chunk form for A/B timing.
#if defined(POOL_CHUNK_BITS) || defined(POOL_NO_BITS_DMA) || defined(POOL_DUP_BITS)
    constexpr bool PAIRB = false;
#else
    constexpr bool PAIRB = true;
#endif
    constexpr int NBI2 = (Npad + 63) / 64;                            // pair form: DMA instructions per pair of chunks (<= 4: one per wave)
    constexpr int NR64 = NBI2 * 64;                                   // rows of a pair buffer
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    // [NBUF][PA][128][64] feature tiles | lut[256] (16 B each) | mask words: [NBUF][NBI*64] (per chunk) or [2][NR64][4] (per pair)
    uint4* lut = (uint4*)(lds + POOL_NBUF * PA * POOL_FT);
    uint32_t* lbits = (uint32_t*)(lut + 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, cg = blockIdx.y, b = blockIdx.z;
    const int map = cg >> 1;
    const int ch0 = (cg & 1) * 128;                                  // first channel of this workgroup
    const int g = lane >> 5;
    const uint16_t* feat = (map == 0 ? xplanes : dplanes);
    const int64_t plane_stride = (int64_t)B * PH_C * HWp;
    const uint16_t* fbase = feat + ((int64_t)b * PH_C + ch0) * HWp;
    const int64_t words_per_row = HWp / 32;
    const uint32_t* brow = bits + (int64_t)b * bits_rows * words_per_row;      // bits_rows >= Npad rows per frame

    const int nchunks = (int)(HWp / POOL_CHUNK);                      // even: HWp is a multiple of 128
    int c0 = (int)((int64_t)split * nchunks / nsplit), c1 = (int)((int64_t)(split + 1) * nchunks / nsplit);
    if (PAIRB) { c0 &= ~1; c1 = split + 1 == nsplit ? nchunks : (c1 & ~1); }      // pixel ranges of whole chunk pairs
    const bool pair_wave = PAIRB && wave < NBI2;                      // this wave carries one mask-word instruction per pair
    // pair form: mask words of the chunk pair starting at (even) chunk c -> pair buffer gb; 64 rows x 16 bytes per instruction
    auto issue_pair_bits = [&](int c, int gb) {
        if (!pair_wave) return;
        int row = wave * 64 + lane;
        if (row > Npad - 1) row = Npad - 1;                           // tail lanes re-read the last row (their LDS rows are never read)
        const uint32_t* src = brow + (int64_t)row * words_per_row + c * 2;
        uint32_t* dst = lbits + gb * (NR64 * 4) + wave * 256;         // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (PH_LDS void*)dst, 16, 0, 0);
    };

    lut[tid] = expand8<E>((uint32_t)tid);                               // byte -> A fragment (8 x {0,1} bf16)

    f32x16_t acc[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
    // pixel counts of the hard masks over this pixel range (the bias term of the folded feat_transform needs them,
    // kernel_update_head.py:225,241): the mask words pass through this kernel anyway -- wave 0 of the first channel group
    // counts them, so the query kernel does not read the bit rows again
    const bool counting = pcount != nullptr && cg == 0 && wave == 0;
    int pc[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) pc[rt] = 0;

    // LDS-DMA of chunk c: features = 16 wave-instructions of 1 KiB (8 channel rows x 128 B) per plane, mask
    // words = NBI instructions of 64 x 4 B (row = l>>1, word = l&1).  Nothing in the loop is a VGPR load, so the
    // only vector-memory wait is the vmcnt(0) in front of the barrier.
    // piece k of this wave's PER_CHUNK DMA instructions for chunk c: k < 4 PA features, then mask words
    auto issue_piece = [&](int c, int buf, int k) {
        if (k < 4 * PA) {
            const int j = wave + 4 * k;
            const int p = j >> 4, jj = j & 15;
            const int row = jj * 8 + (lane >> 3);
            const int q = (lane & 7) ^ pool_swz(row);
            const uint16_t* src = fbase + p * plane_stride + (int64_t)row * HWp + (int64_t)c * POOL_CHUNK + q * 8;
            uint16_t* dst = lds + (buf * PA + p) * POOL_FT + jj * 512;          // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (PH_LDS void*)dst, 16, 0, PH_CPOL_STREAM);
        } else if (!PAIRB) {
            // NBI mask-word instructions per chunk, dealt round-robin: a wave past the end issues one fewer (round 3; it used
            // to repeat the last one so that ONE vmcnt immediate served every wave -- 3 of 8 requests per chunk were duplicates at
            // cfg2, and the mask-word requests cost 8 % of this kernel, POOL_NO_BITS_DMA); the counted waits below are per wave class
#ifdef POOL_DUP_BITS   // round 2's form, for A/B timing
            int j = wave + 4 * (k - 4 * PA);
            if (j > NBI - 1) j = NBI - 1;
#else
            const int j = wave + 4 * (k - 4 * PA);
            if (j > NBI - 1) return;
#endif
            int row = j * 32 + (lane >> 1);
            if (row > Npad - 1) row = Npad - 1;                                    // tail lanes re-read the last row
            const uint32_t* src = brow + (int64_t)row * words_per_row + c * 2 + (lane & 1);
            uint32_t* dst = lbits + buf * (NBI * 64) + j * 64;                    // wave-uniform
            // default policy: a 128-byte line of mask words serves 16 chunks (nt here: 181 -> 270 us)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (PH_LDS void*)dst, 4, 0, 0);
        }
    };
    auto issue_chunk = [&](int c, int buf) {
#pragma unroll
        for (int k = 0; k < PER_CHUNK; ++k) issue_piece(c, buf, k);
    };

    if (PAIRB && c0 < c1) issue_pair_bits(c0, 0);
#pragma unroll
    for (int d = 0; d < POOL_NBUF - 1; ++d)
        if (c0 + d < c1) issue_chunk(c0 + d, d);
    __syncthreads();                                       // LUT visible (this barrier also drains the prologue DMA)

    const uint32_t lut_addr = lds_addr(lut);
    const int frow = wave * 32 + (lane & 31);
    int cur = 0;
    for (int c = c0; c < c1; ++c) {
        // chunk c must have landed; the up to NBUF-2 younger chunks stay in flight ACROSS the barrier:
        // counted vmcnt + RAW s_barrier (__syncthreads would drain the DMA queue with vmcnt(0))
        const int younger = (c1 - 1 - c) < (POOL_NBUF - 2) ? (c1 - 1 - c) : (POOL_NBUF - 2);
        const int pos = (c - c0) & 1;                 // pair form: first / second chunk of its pair
        if (PAIRB) {
            // issue slot of chunk x: E(x) = [mask words of the pair after x's, if x is the first chunk of its pair][features of x + 3].
            // In front of the FIRST chunk of a pair its own words sit in E(c-2), ahead of that slot's feature pieces: the youngest
            // 4 PA x `younger` instructions (features of c+1, c+2) may stay in flight.  In front of the SECOND chunk the words
            // arrived with E(c-3); E(c-1) holds the NEXT pair's words, which may stay in flight as well
            const bool extra = pos == 1 && younger >= 1 && pair_wave;
            if (younger >= 2) {
                if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * PA + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * PA) : "memory");
            } else if (younger == 1) {
                if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PA + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PA) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (NBREM != 0 && wave >= NBREM) {       // this wave issues PER_CHUNK - 1 instructions per chunk (wave-uniform branch)
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PER_CHUNK - 1)) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_CHUNK) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // every wave is past compute(c-1): its buffer, (cur + NBUF-1) % NBUF, is free for chunk c + NBUF-1
        // ... its DMA instructions are issued BETWEEN the row tiles of this chunk's MFMAs: a DMA issue can block on the
        // memory pipe for hundreds of cycles, during which the matrix pipe now has this wave's MFMAs to run
        int nb = cur + POOL_NBUF - 1;
        if (nb >= POOL_NBUF) nb -= POOL_NBUF;
        const bool more = c + POOL_NBUF - 1 < c1;
        const uint32_t ft = lds_addr(lds + cur * PA * POOL_FT);
        const uint32_t wb = PAIRB ? lds_addr(lbits + (((c - c0) >> 1) & 1) * (NR64 * 4)) : lds_addr(lbits + cur * (NBI * 64));
        // B fragments (this lane's channel row, 4 k-steps of 8 pixels in its 32-pixel half) and the mask words
        u32x4_t xf[PA][4];
#pragma unroll
        for (int p = 0; p < PA; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                xf[p][t] = lds_read128_asm(ft + 2 * (p * POOL_FT + frow * POOL_CHUNK + (((g * 4 + t) ^ pool_swz(frow)) * 8)));
        uint32_t w[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
            w[rt] = PAIRB ? lds_read32_asm(wb + 4 * ((rt * 32 + (lane & 31)) * 4 + pos * 2 + g))
                          : lds_read32_asm(wb + 4 * ((rt * 32 + (lane & 31)) * 2 + g));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (counting) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) pc[rt] += __popc(w[rt]);
        }
        // A fragments from the lookup table, one row tile ahead of the MFMAs (LDS returns in order: lgkmcnt(4))
        u32x4_t a[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[0][t] = lds_read128_asm(lut_addr + (((w[0] >> (8 * t)) & 0xFFu) << 4));
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            if (rt + 1 < NRT) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[(rt + 1) & 1][t] = lds_read128_asm(lut_addr + (((w[rt + 1] >> (8 * t)) & 0xFFu) << 4));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < PA; ++p)
                    acc[rt] = mfma32e<E>(__builtin_bit_cast(uint4, a[rt & 1][t]), __builtin_bit_cast(uint4, xf[p][t]), acc[rt]);
            __builtin_amdgcn_sched_barrier(0);
            // pair form: with the first chunk of a pair, the NEXT pair's mask words go out first (into the other pair buffer:
            // its last reader was chunk c - 1, behind this chunk's barrier), then this slot's feature pieces
            if (PAIRB && rt == 0 && pos == 0 && c + 2 < c1) issue_pair_bits(c + 2, (((c - c0) >> 1) + 1) & 1);
            if (more) {
#pragma unroll
                for (int k = rt; k < PER_CHUNK; k += NRT) issue_piece(c + POOL_NBUF - 1, nb, k);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = cur + 1 == POOL_NBUF ? 0 : cur + 1;
    }

    if (counting) {
        // lane (row l & 31, half g) holds the count of its 32-pixel halves: add the two halves, lanes 0..31 write
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int tot = pc[rt] + __shfl_xor(pc[rt], 32);
            if (lane < 32) pcount[((int64_t)b * nsplit + split) * Npad + rt * 32 + lane] = tot;
        }
    }
    // epilogue: partial[b][split][row][map*256 + ch]
    float* out = partial + (((int64_t)b * nsplit + split) * Npad) * 512 + map * 256 + ch0 + wave * 32 + (lane & 31);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            out[(int64_t)row * 512] = acc[rt][r];
        }
}

template <int PA, int NRT, int E>
static void launch_pool(const uint16_t* x, const uint16_t* d, const uint32_t* bits, float* partial, int B, int64_t HWp,
                        int nsplit, int bits_rows, int32_t* pcount, hipStream_t s) {
    // mask words: per-chunk form 4 x NRT x 256 B, pair form 2 x ceil(Npad / 64) x 64 rows x 16 B -- the larger of the two
    const size_t lbits_chunk = (size_t)POOL_NBUF * ((NRT * 64 + 63) / 64) * 64 * 4, lbits_pair = (size_t)2 * ((NRT * 32 + 63) / 64) * 64 * 16;
    const size_t lds = (size_t)POOL_NBUF * PA * POOL_FT * sizeof(uint16_t) + 256 * 16 + (lbits_chunk > lbits_pair ? lbits_chunk : lbits_pair);
    static const bool once = [&] {
        (void)hipFuncSetAttribute((const void*)k_pool<PA, NRT, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((k_pool<PA, NRT, E>), dim3(nsplit, d ? 4 : 2, B), dim3(256), lds, s, x, d, bits, partial, B, HWp,
                       nsplit, bits_rows, pcount);
}

static int pool_run(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, int bits_rows, float* partial, int32_t* pcount,
                    int B, int N, int64_t HW, int nsplit, int prec, void* stream, const char* fn) {
    if (!(xplanes && bits && partial && B > 0 && N > 0 && HW > 0)) { ph_set_error("%s: bad pointer or size", fn); return PH_EINVAL; }
    if (!(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_F16)) {
        ph_set_error("%s: prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_F16", fn);
        return PH_EINVAL;
    }
    if (N > 256) { ph_set_error("%s: at most 256 queries", fn); return PH_EINVAL; }
    const int64_t HWp = ph_hw_padded(HW);
    if (!(nsplit >= 1 && nsplit <= HWp / POOL_CHUNK)) { ph_set_error("%s: nsplit out of range", fn); return PH_EINVAL; }
    const int nrt = ph_n_padded(N) / 32;
    if (bits_rows == 0) bits_rows = nrt * 32;
    if (bits_rows < nrt * 32) { ph_set_error("%s: bits_rows must be >= N rounded up to 32", fn); return PH_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
#define PH_POOL_CASE(R)                                                                         \
    case R:                                                                                     \
        if (prec == PH_PREC_BF16) launch_pool<1, R, PH_E_BF16>(xplanes, dplanes, bits, partial, B, HWp, nsplit, bits_rows, pcount, s); \
        else if (prec == PH_PREC_F16) launch_pool<1, R, PH_E_F16>(xplanes, dplanes, bits, partial, B, HWp, nsplit, bits_rows, pcount, s); \
        else launch_pool<2, R, PH_E_BF16>(xplanes, dplanes, bits, partial, B, HWp, nsplit, bits_rows, pcount, s);             \
        break;
    switch (nrt) {
        PH_POOL_CASE(1) PH_POOL_CASE(2) PH_POOL_CASE(3) PH_POOL_CASE(4)
        PH_POOL_CASE(5) PH_POOL_CASE(6) PH_POOL_CASE(7) PH_POOL_CASE(8)
        default: ph_set_error("%s: unsupported N", fn); return PH_EUNSUPPORTED;
    }
#undef PH_POOL_CASE
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_pool(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, float* partial, int B,
                       int N, int64_t HW, int nsplit, int prec, void* stream) {
    return pool_run(xplanes, dplanes, bits, 0, partial, nullptr, B, N, HW, nsplit, prec, stream, __func__);
}

// the same over the FIRST ph_n_padded(N) rows of a bits tensor with `bits_rows` rows per frame (KernelHead's object
// pooling over the thing rows of the full mask-bit tensor, kernel_head.py:314-320: no copy of the thing rows)
extern "C" int ph_pool_rows(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, int bits_rows, float* partial,
                            int B, int N, int64_t HW, int nsplit, int prec, void* stream) {
    return pool_run(xplanes, dplanes, bits, bits_rows, partial, nullptr, B, N, HW, nsplit, prec, stream, __func__);
}

// ph_pool that also writes pcount[B][nsplit][Npad] (int32): the number of set mask bits of every row in every pixel range --
// what ph_query_stage_counts takes instead of re-reading the bit rows
extern "C" int ph_pool_counts(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, float* partial, int32_t* pcount,
                              int B, int N, int64_t HW, int nsplit, int prec, void* stream) {
    if (!pcount) { ph_set_error("ph_pool_counts: null pcount"); return PH_EINVAL; }
    return pool_run(xplanes, dplanes, bits, 0, partial, pcount, B, N, HW, nsplit, prec, stream, __func__);
}
