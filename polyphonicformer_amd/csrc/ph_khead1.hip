// A1-A5 in ONE pass over the three post-neck maps (polyphonic/kernel_head.py:245-347) -- round 3.
//
// ph_khead.hip evaluates every 256x256 conv twice because GroupNorm needs statistics of the conv OUTPUT over all pixels
// of a (frame, map): a statistics pass and an apply pass, i.e. two reads of 100 MB of fp32 maps per frame at cfg2.  Here
// the conv output of a 128-pixel slice, y = W f [256 ch][128 px] fp32, stays in the ACCUMULATOR REGISTERS of one
// workgroup per CU (8 waves x 64 registers) while the per-group sums of the whole map are exchanged between the
// workgroups INSIDE the launch; the slice is then normalised in registers and handed to the same second GEMM
// (init_kernels / conv_seg / conv_direct_depth on the normalised tile in LDS) as in ph_khead_fused.  Every input byte
// is read once, loc never leaves the CU (x = sem + loc: loc waits in LDS in accumulator layout), and the mask bits come
// from the second GEMM's accumulators (no ph_binarize pass).
//
// Launch geometry: P = HWp / 128 slices per frame, F = floor(#CU / P) frames side by side, grid = F * P workgroups, all
// resident (one per CU: the kernel needs the CU's whole register file); workgroup (slot, slice) walks the frames
// slot, slot + F, ... and per frame the maps loc -> sem -> dfe ("phases").  Needs HWp <= 128 * #CU (cfg2: exactly),
// 32 groups, one 16-bit plane (bf16 or fp16 grade); everything else takes the two-pass path.
//
// Statistics hand-off of one (frame, map) item (cdna_hip_programming.md Guideline 16, form R2: the data is the flag; no
// ticket, no fence, two memory hops): every slice's wave 0 stores its 64 partial sums as eight-byte {tag, value} granules,
// column-major; slice c < 64 of the frame OWNS column c: it polls that column's P granules, adds them in a fixed order
// (lane-local in slice order, then an xor butterfly, fp64: deterministic) and publishes the total as two granules;
// every workgroup's wave 0 polls the 64 totals and derives (mean, rstd) of the 32 groups.  Wave 0 is the only wave with
// no input prefetch in flight -- vector memory returns in order, so a poll behind 16 HBM loads would wait for them.
// Spins are bounded in TIME (s_memrealtime, K1Args::timeout_ticks): a launch whose workgroups cannot all become resident --
// another kernel holds CUs for longer than the bound, or two such launches starve each other -- sets the per-call `status`
// word and every workgroup that sees it (in its poll loops, or when it starts) LEAVES the kernel.  The entry point then runs
// the two-pass kernels of ph_khead.hip predicated on that word (ph_khead_fused_if / ph_binarize_if: launched behind every
// one-pass launch, they return at once when the word is 0), so the SAME call still produces the reference's tensors, also
// inside a HIP graph; a sticky word (never cleared by the call) counts the time-outs for diagnostics.  All polled words are
// zeroed by a memset node in front of every launch.
#include "ph_common.h"

// Two geometries of the same kernel.  A wave always owns FOUR 32 x 32 accumulator tiles of the conv output (64 registers):
//   K1Geo<1, 4> ("wide"):  8 waves, wave w = channels 32w..32w+31 of a 128-pixel slice; 159 KB of LDS -> ONE workgroup per CU;
//   K1Geo<2, 2> ("pair"):  4 waves, wave w = channels 64w..64w+63 of a  64-pixel slice;  78 KB of LDS -> TWO workgroups per CU
//                          (same registers per wave, same waves per CU).  Built on the SQ counters of the wide form (60 % of
//                          the wave cycles waiting, MFMA busy 15 %, profiles/r03) in the hope that one workgroup computes
//                          while the other waits; measured slower (see the launcher), compiled only with -DK1_WITH_PAIR.
template <int RT_, int CT_> struct K1Geo {
    static constexpr int RT = RT_, CT = CT_;                  // row tiles (32 channels) x column tiles (32 px) per wave
    static constexpr int WAVES = 8 / RT, THREADS = 64 * WAVES, PX = 32 * CT;
    // LDS row stride of the slice.  CT = 4: 160 elements = 80 dwords: rows r .. r+3 of a transposing read start at banks
    // 0 / 16 / 32 / 48 -> conflict free.  CT = 2: 72 elements = 36 dwords (a conflict-free 96 would not leave room for two
    // workgroups per CU): the second 16-pixel half of a 32-lane pass collides 2-way with one row of the first.
    static constexpr int LD = CT == 4 ? 160 : 72;
    static constexpr int WGS_PER_CU = CT == 4 ? 1 : 2;
    static constexpr int SSB = CT == 4 ? 4096 : 2048;         // bytes of the ss / bitsl union: [256] float2, [256 rows][CT] words
    static constexpr size_t LDS_BYTES = (size_t)256 * LD * 2 + (size_t)WAVES * 8192 + SSB + 3 * 512 * 4 + 288 * 4 + 64 * 4 + 64 * 4 + 16;
};
typedef K1Geo<1, 4> K1Wide;
typedef K1Geo<2, 2> K1Pair;
constexpr unsigned long long K1_TICKS_PER_US = 100;            // s_memrealtime counts at 100 MHz

#define K1_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#ifdef K1_SKIP_OUT        // timing experiments only
#define K1_OUT_ON 0
#else
#define K1_OUT_ON 1
#endif
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

struct K1Args {
    const void* f[3];             // maps: fp32 [B][256][HW] (INFMT 1) or 16-bit planes [B][256][HWp] (INFMT 2)
    const uint16_t* wfrag[3];     // conv weights as MFMA 32x32x16 A fragments [8 row tiles][16 k-steps][64 lanes][8]
    const float* gn;              // [3][2][256] gamma, beta
    const uint16_t* w2[3];        // static 1x1 kernels as A fragments [m2_tiles][16][64][8]
    const float* bias2[3];        // [m2_tiles * 32] or null
    int m2_tiles[3], n2[3];
    void* out2[3];                // logits [B][out2_rows][HW], rows [0, n2) written; out_dtype PH_OUT_*
    int out2_rows[3];
    int out_dtype;
    int stuff_lo, n_stuff, n_init;     // seg rows [stuff_lo, stuff_lo + n_stuff) also go to rows n_init + . of out2[0]
    uint16_t* x_planes;           // [B][256][HWp]
    uint16_t* dfe_planes;
    float* x_f32;                 // optional fp32 NCHW copies (the reference API's x_feats / depth_feats)
    float* dfe_f32;
    uint32_t* bits;               // optional mask bits [B][bits_rows][HWp/32] of out2[0]'s rows (rows >= n_init + n_stuff: 0)
    int bits_rows;
    // hand-off state
    unsigned long long* gran1;    // [3B][64 columns][P] {tag, fp32 sum of one slice}          } zeroed before every launch
    unsigned long long* gran2;    // [3B][128] {tag, half of the fp64 total of a column}       }
    unsigned* status;             // [1] != 0: a bounded spin timed out, the launch gave up    }
    unsigned* sticky;             // [2] never cleared by a call: {any time-out so far, number of workgroup time-outs}
    unsigned long long timeout_ticks;   // bound of one hand-off wait, in s_memrealtime ticks
    int B, P, F;
    int64_t HW, HWp;
    float eps;
    unsigned long long* timeline;  // debugging: [workgroup][wave 0 / wave 1][phase][24] s_memtime stamps, or null
};

__device__ __forceinline__ void k1_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- staging registers of one [256 ch][128 px] input slice ----------------------------------------------------------
template <int INFMT, int E, typename G> struct K1Stage;
// fp32 NCHW, HW % 4 == 0: PX / 4 threads x 16 B per channel row, 16 rows per pass, 16 passes
template <int E, typename G> struct K1Stage<1, E, G> {
    static constexpr int TPR = G::PX / 4;                     // threads per channel row; THREADS / TPR = 16 rows per pass
    uint4 v[16];
    // part < 0: the whole slice; otherwise request `part` alone (the requests of the next slice are paced through the first
    // GEMM: a wave blocks in the issue of a vector-memory instruction while the CU's memory pipe is full, and 8 waves issuing
    // 128 KB at once serialise the phase's compute with its memory time)
    static constexpr int NREQ = 16;
    template <int part = -1> __device__ __forceinline__ void load(const K1Args& a, int m, int b, int64_t px0, int tid) {
        const float* src = (const float*)a.f[m] + (int64_t)b * 256 * a.HW;
        int64_t col = px0 + (tid % TPR) * 4;
        if (col > a.HW - 4) col = a.HW - 4;                                   // clamped; zeroed in store()
        const uint32_t voff = (uint32_t)((((int64_t)(tid / TPR)) * a.HW + col) * 4);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (part < 0 || q == part) v[q] = ld_nt16((const char*)(src + (int64_t)q * 16 * a.HW) + voff);
    }
    // fp32 -> 16-bit in registers (64 -> 32): done as soon as the slice has landed and T is still busy, so that the second
    // GEMM and its epilogue have the registers
    uint2 pk[16];
    __device__ __forceinline__ void pack(int tid, int64_t HW, int64_t px0) {
        const bool ok = px0 + (tid % TPR) * 4 < HW;                         // a 4-pixel group is inside or outside as a whole
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float x0 = ok ? __uint_as_float(v[q].x) : 0.f, x1 = ok ? __uint_as_float(v[q].y) : 0.f;
            const float x2 = ok ? __uint_as_float(v[q].z) : 0.f, x3 = ok ? __uint_as_float(v[q].w) : 0.f;
            pk[q] = make_uint2(f2e_pk<E>(x0, x1), f2e_pk<E>(x2, x3));
        }
    }
    __device__ __forceinline__ void store(uint16_t* T, int tid) const {
#pragma unroll
        for (int q = 0; q < 16; ++q) *(uint2*)(T + (q * 16 + tid / TPR) * G::LD + (tid % TPR) * 4) = pk[q];
    }
};
// 16-bit planes (zero padded to HWp by their producer): PX / 8 threads x 16 B per channel row, 32 rows per pass, 8 passes
template <int E, typename G> struct K1Stage<2, E, G> {
    static constexpr int TPR = G::PX / 8;
    uint4 v[8];
    static constexpr int NREQ = 8;
    template <int part = -1> __device__ __forceinline__ void load(const K1Args& a, int m, int b, int64_t px0, int tid) {
        const uint16_t* src = (const uint16_t*)a.f[m] + (int64_t)b * 256 * a.HWp;
        const uint32_t voff = (uint32_t)((((int64_t)(tid / TPR)) * a.HWp + px0 + (tid % TPR) * 8) * 2);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (part < 0 || q == part) v[q] = ld_nt16((const char*)(src + (int64_t)q * 32 * a.HWp) + voff);
    }
    __device__ __forceinline__ void pack(int, int64_t, int64_t) {}
    __device__ __forceinline__ void store(uint16_t* T, int tid) const {
#pragma unroll
        for (int q = 0; q < 8; ++q) *(uint4*)(T + (q * 32 + tid / TPR) * G::LD + (tid % TPR) * 8) = v[q];
    }
};

// opaque copy of a lane index: everything derived from it is recomputed where it is used instead of being hoisted out of
// the phase loop and kept (or spilled -- a scratch reload would wait behind the slice prefetch) across it
__device__ __forceinline__ int k1_fresh(int x) { asm volatile("" : "+v"(x)); return x; }

// A fragments of one 32-row tile: 16 coalesced 1 KiB wave loads
__device__ __forceinline__ void k1_load_a(uint4 (&af)[16], const uint16_t* __restrict__ frag, int rt, int lane) {
    const uint16_t* p = frag + ((int64_t)rt * 16 * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[ks] = *(const uint4*)(p + ks * 512);
}

// acc0 / acc1 [32 rows of A][32 px of column tiles ct, ct + 1] over K = 256 channel rows of the LDS tile.  Two independent
// accumulator chains: consecutive MFMAs on ONE accumulator wait for each other's result (64 cycles), two chains issue at
// the matrix pipe's rate.  `every2(j)` runs after k-steps 2j, 2j + 1 (the paced requests of the next slice).
template <int E, int K1_LD, typename F>
__device__ __forceinline__ void k1_gemm2(const uint4 (&af)[16], const uint16_t* T, int ct, int lane, f32x16_t& acc0,
                                         f32x16_t& acc1, F&& every2) {
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const uint16_t* base = T + (g * 8 + (i16 >> 2)) * K1_LD + ct * 32 + gi * 16 + (i16 & 3) * 4;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const uint2 lo0 = lds_read_tr16(base + ks * 16 * K1_LD);
        const uint2 hi0 = lds_read_tr16(base + (ks * 16 + 4) * K1_LD);
        const uint2 lo1 = lds_read_tr16(base + ks * 16 * K1_LD + 32);
        const uint2 hi1 = lds_read_tr16(base + (ks * 16 + 4) * K1_LD + 32);
        acc0 = mfma32e<E>(af[ks], make_uint4(lo0.x, lo0.y, hi0.x, hi0.y), acc0);
        acc1 = mfma32e<E>(af[ks], make_uint4(lo1.x, lo1.y, hi1.x, hi1.y), acc1);
        if ((ks & 1) == 1) {
            __builtin_amdgcn_sched_barrier(0);
            every2(ks >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// the second GEMM's forms: D[px][kernel row] (operands swapped), accumulators start at the lane's bias
template <int E, int K1_LD>
__device__ __forceinline__ f32x16_t k1_gemm_t(const uint4 (&w)[16], const uint16_t* T, int ct, int lane, float bz) {
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bz;
    const uint16_t* base = T + (g * 8 + (i16 >> 2)) * K1_LD + ct * 32 + gi * 16 + (i16 & 3) * 4;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const uint2 lo = lds_read_tr16(base + ks * 16 * K1_LD);
        const uint2 hi = lds_read_tr16(base + (ks * 16 + 4) * K1_LD);
        acc = mfma32e<E>(make_uint4(lo.x, lo.y, hi.x, hi.y), w[ks], acc);
        if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}
template <int E, int K1_LD>
__device__ __forceinline__ void k1_gemm2_t(const uint4 (&w)[16], const uint16_t* T, int ct, int lane, float bz, f32x16_t& acc0,
                                           f32x16_t& acc1) {
    const int g = lane >> 5, i16 = lane & 15, gi = (lane >> 4) & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = bz; acc1[r] = bz; }
    const uint16_t* base = T + (g * 8 + (i16 >> 2)) * K1_LD + ct * 32 + gi * 16 + (i16 & 3) * 4;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const uint2 lo0 = lds_read_tr16(base + ks * 16 * K1_LD);
        const uint2 hi0 = lds_read_tr16(base + (ks * 16 + 4) * K1_LD);
        const uint2 lo1 = lds_read_tr16(base + ks * 16 * K1_LD + 32);
        const uint2 hi1 = lds_read_tr16(base + (ks * 16 + 4) * K1_LD + 32);
        acc0 = mfma32e<E>(make_uint4(lo0.x, lo0.y, hi0.x, hi0.y), w[ks], acc0);
        acc1 = mfma32e<E>(make_uint4(lo1.x, lo1.y, hi1.x, hi1.y), w[ks], acc1);
        if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ float k1_wave_sum(float x) {
    x = wave_group16_sum(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

template <typename T> __device__ __forceinline__ void k1_store_logit(T* p, float v);
template <> __device__ __forceinline__ void k1_store_logit<float>(float* p, float v) { __builtin_nontemporal_store(v, p); }
template <> __device__ __forceinline__ void k1_store_logit<uint16_t>(uint16_t* p, float v) { *p = (uint16_t)f2h(v); }

#ifdef K1_TIMELINE        // build with PH_EXTRA_HIPCC_FLAGS=-DK1_TIMELINE (scratch/a1_timeline.py); costs registers
#define K1_STAMP(i)                                                                                         \
    do {                                                                                                    \
        if (a.timeline && wave < 2 && lane == 0)                                                            \
            a.timeline[(((int64_t)blockIdx.x * 2 + wave) * nph + ph) * 24 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define K1_STAMP(i) do { } while (0)
#endif

// The phase loop of one wave.  W0 = wave 0: it alone talks to the other workgroups, and it requests its share of the next
// slice only after its polls (vector memory returns in order: a poll behind 16 HBM loads would wait for them).  Two
// instantiations instead of `if (wave == 0)` inside one loop keep every register array's live range straight-line (a
// conditionally reloaded array stays alive through the whole iteration in the other arm).
// Rule of the loop body: between the request of the next slice and the top of the next phase a wave must not WAIT for any
// vector-memory load (global or scratch) -- constants live in LDS, the second GEMM's weights are requested before the
// slice, the next phase's conv weights after it.
template <bool W0, int INFMT, int E, typename OutT, bool F32O, typename G>
__device__ __forceinline__ void k1_run(const K1Args& a, uint16_t* lds, int tid, int lane, int wave) {
    constexpr int RT = G::RT, CT = G::CT, LD = G::LD;
    uint16_t* T = lds;                                              // [256][LD]: input slice, later the normalised slice
    uint16_t* keepw = lds + 256 * LD + wave * 4096;                 // 8 KB per wave: loc in accumulator layout, later x rows
    float2* ss = (float2*)(lds + 256 * LD + G::WAVES * 4096);       // [256] per-channel scale / shift of the normalisation ...
    uint32_t* bitsl = (uint32_t*)ss;                                // ... and, later in the phase, [256 rows][CT words] mask bits
    float* gnl = (float*)((char*)ss + G::SSB);                      // [3][2][256] gamma, beta of the three GroupNorms
    float* b2l = gnl + 3 * 512;                                     // [256 + 32] bias of conv_seg, conv_direct_depth
    float* red = b2l + 288;                                         // [64] this slice's sums
    float* statl = red + 64;                                        // [64] (mean, rstd) x 32 groups
    volatile int* abortl = (volatile int*)(statl + 64);             // [1] != 0: this launch gave up (set by wave 0, see `expired`)
    const int g = lane >> 5;
    const int ch0 = wave * RT * 32;                                 // this wave's first channel
    uint16_t* rows = T + ch0 * LD;                                  // this wave's channel rows
    const int slot = blockIdx.x / a.P, pair = blockIdx.x - slot * a.P;
    const int64_t px0 = (int64_t)pair * G::PX;
    const int64_t wpr = a.HWp / 32;                                 // mask words per row
    const int nfr = slot < a.B ? (a.B - slot + a.F - 1) / a.F : 0;  // frames of this workgroup: slot, slot + F, ...
    int nph = 3 * nfr;
    if (nph == 0) return;
    gu64* g1 = (gu64*)a.gran1;
    gu64* g2 = (gu64*)a.gran2;
    gu32* gstatus = (gu32*)a.status;
    (void)g;

    K1Stage<INFMT, E, G> stg;
    uint4 af[16];
    stg.load(a, 0, slot, px0, tid);
    k1_load_a(af, a.wfrag[0], wave * RT, lane);
    stg.pack(tid, a.HW, px0);
    for (int ph = 0; ph < nph; ++ph) {
        const int m = ph % 3, b = slot + (ph / 3) * a.F;
        const int item = b * 3 + m;
        // the phase after this one; after the last phase the same slice is requested once more and never used (an
        // unconditional reload keeps the staging registers dead between their store and the request)
        const int ph1 = ph + 1 < nph ? ph + 1 : ph;
        const int nm = ph1 % 3, nb = slot + (ph1 / 3) * a.F;
        K1_STAMP(0);
        // second GEMM: wpr2 waves share a row tile of the static kernels (each takes CT / wpr2 column tiles of the slice);
        // WAVES / wpr2 row tiles per pass, as many passes as the map's kernels need (one in the wide geometry)
        const int m2 = a.m2_tiles[m];
        int wpr2 = CT;
        while (wpr2 > 1 && G::WAVES / wpr2 < m2) wpr2 >>= 1;
        const int rpass = G::WAVES / wpr2;
        const int rt2 = wave / wpr2, nct2 = CT / wpr2, ct20 = (wave % wpr2) * nct2;
        const bool act2 = rt2 < m2;
        stg.store(T, k1_fresh(tid));
        __syncthreads();
        K1_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        // first GEMM, tile q = t * CT + ct (row tile t of this wave, column tile ct).  The next slice's 16 (8) requests are
        // issued between the MFMAs of the LAST row tile (behind the reload of the conv weights of that tile: vector memory
        // returns in order), evenly paced -- except in wave 0, whose polls must not return behind them: it requests its share
        // after the polls
        f32x16_t y[4];
        {
            const int l = k1_fresh(lane), t = k1_fresh(tid);
            constexpr int NR = K1Stage<INFMT, E, G>::NREQ;
            constexpr int QPS = 4 / CT;                     // request slots (of 16) per `every2` call of the last row tile
#define K1_REQ(Q)                                                                       \
    if constexpr ((Q) < 16 && (Q) % (16 / NR) == 0) { if (!W0) stg.template load<(Q) / (16 / NR)>(a, nm, nb, px0, t); }
#define K1_REQJ(BASE, J) { K1_REQ((BASE) + (J) * QPS); if constexpr (QPS == 2) { K1_REQ((BASE) + (J) * QPS + 1); } }
#define K1_REQS(BASE)                                                                   \
    [&](int j) {                                                                        \
        if (j == 0) K1_REQJ(BASE, 0) else if (j == 1) K1_REQJ(BASE, 1) else if (j == 2) K1_REQJ(BASE, 2)      \
        else if (j == 3) K1_REQJ(BASE, 3) else if (j == 4) K1_REQJ(BASE, 4) else if (j == 5) K1_REQJ(BASE, 5) \
        else if (j == 6) K1_REQJ(BASE, 6) else K1_REQJ(BASE, 7)                                               \
    }
            if constexpr (RT == 1) {
                k1_gemm2<E, LD>(af, T, 0, l, y[0], y[1], K1_REQS(0));
                k1_gemm2<E, LD>(af, T, 2, l, y[2], y[3], K1_REQS(8));
            } else {
                k1_gemm2<E, LD>(af, T, 0, l, y[0], y[1], [](int) {});
                __builtin_amdgcn_sched_barrier(0);
                k1_load_a(af, a.wfrag[m], wave * RT + 1, l);
                __builtin_amdgcn_sched_barrier(0);
                k1_gemm2<E, LD>(af, T, 0, l, y[2], y[3], K1_REQS(0));
            }
#undef K1_REQS
#undef K1_REQJ
#undef K1_REQ
        }
        __builtin_amdgcn_sched_barrier(0);
        K1_STAMP(2);
        // ---- sums of this slice: the wave owns groups 4 (RT wave + t) + j (8 channels each: accumulator rows 8j .. 8j+7 of
        // row tile t).  (one group at a time, two running sums: the tree the compiler builds from a flat loop needs 30 more
        // registers at the point where y, the second GEMM's weights and the next slice are all live)
#pragma unroll
        for (int tt = 0; tt < RT; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = 0.f, w = 0.f;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float v = y[tt * CT + ct][j * 4 + q]; u += v; w = fmaf(v, v, w); }
                u = k1_wave_sum(u);
                w = k1_wave_sum(w);
                if (lane == 0) { red[((wave * RT + tt) * 4 + j) * 2] = u; red[((wave * RT + tt) * 4 + j) * 2 + 1] = w; }
                __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();                                             // red complete; every wave is past its GEMM
        K1_STAMP(3);
        if (W0) {
            const double inv_n = 1.0 / (8.0 * (double)a.HW);
            // the bound of this phase's waits.  `expired` is wave-uniform: every 16th poll it looks at the per-call status word
            // (another workgroup gave up: follow at once) and at the clock; the first workgroup to pass the bound raises the word
            const unsigned long long t_wait0 = __builtin_amdgcn_s_memrealtime();
            bool dead = false;
            auto expired = [&](unsigned& spins) -> bool {
                if ((++spins & 15u) != 0) return false;
                if (__hip_atomic_load(gstatus, K1_RLX) != 0) return true;
                if (__builtin_amdgcn_s_memrealtime() - t_wait0 <= a.timeout_ticks) return false;
                if (lane == 0) {
                    __hip_atomic_store(gstatus, 1u, K1_RLX);
                    __hip_atomic_fetch_or((gu32*)a.sticky, 1u, K1_RLX);
                    __hip_atomic_fetch_add((gu32*)a.sticky + 1, 1u, K1_RLX);
                }
                return true;
            };
            // publish this slice's 64 sums: one granule {tag, value} per lane, column-major ([column][slice]) so that a
            // column's owner reads one contiguous run
            const unsigned long long tag = (unsigned long long)(unsigned)(item + 1) << 32;
            __hip_atomic_store(g1 + ((int64_t)item * 64 + lane) * a.P + pair, tag | __float_as_uint(red[lane]), K1_RLX);
            K1_STAMP(4);
            // owner duty: slice p adds column p (and p + P, ... when the frame has fewer than 64 slices) over all slices in a
            // fixed order -- lane-local in slice order, then an xor butterfly -- and publishes the fp64 total as two granules
            for (int c = pair; c < 64 && !dead; c += a.P) {
                const gu64* col = g1 + ((int64_t)item * 64 + c) * a.P;
                double acc = 0.0;
                unsigned spins = 0;
                for (int j0 = 0; j0 < a.P && !dead; j0 += 256) {
                    unsigned long long x[4];
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int j = j0 + lane + 64 * i;
                            x[i] = j < a.P ? __hip_atomic_load(col + j, K1_RLX) : tag;
                            ok = ok && (unsigned)(x[i] >> 32) == (unsigned)(item + 1);
                        }
#ifdef K1_NO_POLL
                        break;
#endif
                        if (__all(ok)) break;
                        if (expired(spins)) { dead = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (j0 + lane + 64 * i < a.P) acc += (double)__uint_as_float((unsigned)x[i]);
                }
#pragma unroll
                for (int q = 1; q < 64; q <<= 1) acc += __shfl_xor(acc, q);
                const unsigned long long bits64 = (unsigned long long)__double_as_longlong(acc);
                if (lane < 2)
                    __hip_atomic_store(g2 + (int64_t)item * 128 + 2 * c + lane,
                                       tag | (unsigned)(lane ? (bits64 >> 32) : (bits64 & 0xFFFFFFFFull)), K1_RLX);
            }
            K1_STAMP(5);
            // every workgroup: poll the 64 totals (two granules per lane) until all carry this item's tag
            {
                const gu64* gp = g2 + (int64_t)item * 128 + 2 * lane;
                unsigned long long lo = 0, hi = 0;
                unsigned spins = 0;
                while (!dead) {
                    lo = __hip_atomic_load(gp, K1_RLX);
                    hi = __hip_atomic_load(gp + 1, K1_RLX);
                    if (__all((unsigned)(lo >> 32) == (unsigned)(item + 1) && (unsigned)(hi >> 32) == (unsigned)(item + 1))) break;
                    if (expired(spins)) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                // column 2g = sum, 2g + 1 = sum of squares of group g: the even lane computes (mean, rstd)
                const double tot = __longlong_as_double((long long)(((hi & 0xFFFFFFFFull) << 32) | (lo & 0xFFFFFFFFull)));
                const double other = __shfl_xor(tot, 1);
                if ((lane & 1) == 0) {
                    const double mean = tot * inv_n;
                    double var = other * inv_n - mean * mean;
                    if (var < 0.0) var = 0.0;
                    statl[lane] = (float)mean;
                    statl[lane + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
                }
            }
            K1_STAMP(6);
            if (dead && lane == 0) *abortl = 1;
            stg.load(a, nm, nb, px0, k1_fresh(tid));
        }
        __syncthreads();
        // the launch gave up (uniform: written before the barrier): this becomes the workgroup's LAST phase.  Not an early exit --
        // `return` / `break` here adds an exit edge in the middle of the phase body and costs 36-40 spilled registers (a scratch
        // reload waits behind the slice prefetch: a1 0.95 -> 1.22 ms per 16 frames, measured); the rest of the body has no wait on
        // other workgroups, its stores are garbage that the caller's predicated fallback overwrites
        if (*abortl) nph = 0;
        K1_STAMP(7);
        __builtin_amdgcn_sched_barrier(0);
        // ---- normalise in registers ------------------------------------------------------------------------------------
        for (int t = k1_fresh(tid); t < 256; t += G::THREADS) {
            const float mean = statl[(t >> 3) * 2], rstd = statl[(t >> 3) * 2 + 1];
            const float ga = gnl[m * 512 + t], be = gnl[m * 512 + 256 + t];
            ss[t] = make_float2(rstd * ga, be - mean * rstd * ga);
        }
        __syncthreads();
        const int ln = k1_fresh(lane), gn = ln >> 5;                           // lane / half of this section
#pragma unroll
        for (int tt = 0; tt < RT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 s = ss[ch0 + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * gn];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bool inside = px0 + ct * 32 + (ln & 31) < a.HW;
                    const float v = fmaxf(y[tt * CT + ct][r] * s.x + s.y, 0.f);
                    y[tt * CT + ct][r] = inside ? v : 0.f;                     // planes are zero padded
                }
            }
        K1_STAMP(8);
        // one tile at a time: the normalised values go to this wave's rows of T (the B operand of the second GEMM);
        //   loc: a 16-bit copy in accumulator layout waits for the sem phase in this wave's 8 KB of `keep` (2 KB per
        //        tile: 2 x 16 bytes per lane, lane-linear);
        //   sem: x_feats = semantic_feats + loc_feats (kernel_head.py:303) overwrites that copy as [32 rows][32 px] rows;
        //   dfe: nothing else
        float* f32o = !F32O ? nullptr : (m == 1 ? a.x_f32 : (m == 2 ? a.dfe_f32 : nullptr));
        // the weights of the second GEMM (one row tile per wave and pass, 64 registers) are requested a quarter per tile,
        // as the accumulators of that tile die: by now the slice requested during the first GEMM has landed, so waiting for
        // them later (in order) costs nothing
        uint4 a2[16];
        const uint16_t* a2p = a.w2[m] + ((int64_t)(act2 ? rt2 : 0) * 16 * 64 + k1_fresh(lane)) * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tt = q / CT, ct = q % CT;
            const int chq = ch0 + tt * 32;                                    // first channel of this tile
            uint4* kq = (uint4*)(keepw + q * 1024);                           // this tile's 2 KB
            const bool f32w = F32O && f32o && px0 + ct * 32 + (ln & 31) < a.HW;
            if (m == 1) {
                // x_feats = semantic_feats + loc_feats (kernel_head.py:303), 8 accumulator rows at a time; x overwrites loc in
                // the lane's own 16-byte slots (same accumulator layout: no other lane's data is touched)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const uint4 k4 = kq[ii * 64 + ln];
                    const uint32_t w[4] = {k4.x, k4.y, k4.z, k4.w};
                    uint32_t xo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = ii * 8 + e * 2;
                        const float x0 = y[q][r] + e2f<E>(w[e] & 0xFFFFu), x1 = y[q][r + 1] + e2f<E>(w[e] >> 16);
                        if (f32w) {
                            float* ub = f32o + ((int64_t)b * 256 + chq + (r & 3) + 8 * (r >> 2)) * a.HW + px0 + ct * 32;
                            ub[(uint32_t)(4 * gn * a.HW + (ln & 31))] = x0;
                            ub[(uint32_t)((4 * gn + 1) * a.HW + (ln & 31))] = x1;
                        }
                        xo[e] = f2e_pk<E>(x0, x1);
                    }
                    kq[ii * 64 + ln] = make_uint4(xo[0], xo[1], xo[2], xo[3]);
                }
            } else {
                if (m == 0) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        uint32_t w[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = f2e_pk<E>(y[q][ii * 8 + e * 2], y[q][ii * 8 + e * 2 + 1]);
                        kq[ii * 64 + ln] = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
                if (f32w) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* ub = f32o + ((int64_t)b * 256 + chq + (r & 3) + 8 * (r >> 2)) * a.HW + px0 + ct * 32;
                        ub[(uint32_t)(4 * gn * a.HW + (ln & 31))] = y[q][r];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rows[(tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * gn) * LD + ct * 32 + (ln & 31)] = (uint16_t)f2e<E>(y[q][r]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = q * 4; ks < q * 4 + 4; ++ks) a2[ks] = *(const uint4*)(a2p + ks * 512);
            __builtin_amdgcn_sched_barrier(0);
        }
        K1_STAMP(9);
        if (m >= 1) {
            // planes out, whole row segments (PX pixels = PX / 8 lanes x 16 bytes) of this wave's 32 RT channel rows, 8 wave
            // stores.  dfe: straight from this wave's rows of T.  x: its accumulator-layout copy in `keep` is first rewritten
            // in place as [32 rows][32 px] rows per tile (the wave's rows of T hold sem for the second GEMM)
            k1_wave_sync();
            constexpr int TPRO = G::PX / 8, RPI = 64 / TPRO;              // lanes per row segment, rows per wave store
            char* dstp = (char*)((m == 1 ? a.x_planes : a.dfe_planes) + ((int64_t)b * 256 + ch0) * a.HWp + px0);   // uniform
            const int lp = k1_fresh(lane), piece = lp % TPRO, rowl = lp / TPRO;
            const uint32_t voff = (uint32_t)(((int64_t)rowl * a.HWp + piece * 8) * 2);
            if (m == 1) {
                // keep (accumulator layout, pairs of rows r, r+1 packed) -> [32 rows][32 px] rows per tile, in place:
                // every lane first reads its 2 x 16 bytes of the tile, the wave syncs, then writes 16 two-byte elements
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint4* kq = (uint4*)(keepw + q * 1024);
                    const uint4 q0 = kq[lp], q1 = kq[64 + lp];
                    k1_wave_sync();
                    const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    uint16_t* xr = keepw + q * 1024;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = e * 2;
                        xr[((r & 3) + 8 * (r >> 2) + 4 * (lp >> 5)) * 32 + (lp & 31)] = (uint16_t)(w[e] & 0xFFFFu);
                        xr[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * (lp >> 5)) * 32 + (lp & 31)] = (uint16_t)(w[e] >> 16);
                    }
                }
                k1_wave_sync();
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                // row it * RPI + rowl of the wave's 32 RT rows: row tile (it * RPI) / 32 (compile time: RPI divides 32)
                const int tt = (it * RPI) / 32, rin = (it * RPI) % 32 + rowl;
                const uint16_t* src = m == 1 ? keepw + (tt * CT + (piece >> 2)) * 1024 + rin * 32 + (piece & 3) * 8
                                             : rows + (it * RPI + rowl) * LD + piece * 8;
                if (K1_OUT_ON) st_nt16(dstp + (int64_t)it * RPI * a.HWp * 2 + voff, *(const uint4*)src);
            }
        }
        K1_STAMP(10);
        __syncthreads();
        K1_STAMP(11);
        __builtin_amdgcn_sched_barrier(0);
        stg.pack(k1_fresh(tid), a.HW, px0);          // the next slice has landed by now
        __builtin_amdgcn_sched_barrier(0);
        // ---- second GEMM, TRANSPOSED: D[px][kernel row] = slice^T x kernels^T.  The fragment registers of the two operands
        // are the same as for D[row][px] (A lane l: row l & 31, k = 8 (l >> 5) ..; B lane l: column l & 31, same k), only
        // their roles swap -- and every lane then owns ONE kernel row and 16 pixels in groups of 4 consecutive ones
        // (register r <-> pixel (r & 3) + 8 (r >> 2) + 4 (l >> 5)): one bias per lane, 16-byte row stores straight from the
        // accumulators (no LDS transposition: the 4-byte-per-lane stores of the D[row][px] form are store-issue bound at
        // 2.5k cycles per tile, an LDS patch costs four wave syncs), the mask bits by sixteen compares per lane ---------------
        {
            OutT* out2 = (OutT*)a.out2[m];
            OutT* out2b = (m == 1 && a.n_stuff > 0) ? (OutT*)a.out2[0] : nullptr;
            const float* bias = m == 1 ? b2l : b2l + 256;
            auto epilogue = [&](const f32x16_t& v, int rt, int ct) {
                const int l2 = k1_fresh(lane), g2h = l2 >> 5;   // per call: nothing lane-derived is hoisted across tiles
                const int row = rt * 32 + (l2 & 31);
                const int64_t pxb = px0 + ct * 32;
                OutT* tb = out2 + ((int64_t)b * a.out2_rows[m] + rt * 32) * a.HW + pxb;   // uniform
                const bool rok = row < a.n2[m];
                const bool dual = out2b && row >= a.stuff_lo && row < a.stuff_lo + a.n_stuff;
                OutT* tb2 = out2b ? out2b + ((int64_t)b * a.out2_rows[0] + a.n_init - a.stuff_lo + rt * 32) * a.HW + pxb : nullptr;
                if (K1_OUT_ON) {
                    if (a.HW % 4 == 0) {
                        const uint32_t go = (uint32_t)(((int64_t)(l2 & 31) * a.HW + 4 * g2h) * sizeof(OutT));   // lane's byte offset in the tile
#pragma unroll
                        for (int jq = 0; jq < 4; ++jq) {
                            const bool pin = pxb + 8 * jq + 4 * g2h < a.HW;        // 4 pixels are inside or outside as a whole
                            if (sizeof(OutT) == 4) {
                                const uint4 q4 = make_uint4(__float_as_uint(v[4 * jq]), __float_as_uint(v[4 * jq + 1]),
                                                            __float_as_uint(v[4 * jq + 2]), __float_as_uint(v[4 * jq + 3]));
                                if (rok && pin) *(uint4*)((char*)tb + go + jq * 8 * sizeof(OutT)) = q4;
                                if (dual && pin) *(uint4*)((char*)tb2 + go + jq * 8 * sizeof(OutT)) = q4;
                            } else {
                                const uint2 q2 = make_uint2(f2h_pk(v[4 * jq], v[4 * jq + 1]), f2h_pk(v[4 * jq + 2], v[4 * jq + 3]));
                                if (rok && pin) *(uint2*)((char*)tb + go + jq * 8 * sizeof(OutT)) = q2;
                                if (dual && pin) *(uint2*)((char*)tb2 + go + jq * 8 * sizeof(OutT)) = q2;
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int px = (r & 3) + 8 * (r >> 2) + 4 * g2h;
                            if (pxb + px < a.HW) {
                                if (rok) k1_store_logit<OutT>(tb + (int64_t)(l2 & 31) * a.HW + px, v[r]);
                                if (dual) k1_store_logit<OutT>(tb2 + (int64_t)(l2 & 31) * a.HW + px, v[r]);
                            }
                        }
                    }
                }
#ifndef K1_NO_BITS
                if (a.bits && m < 2) {
                    // hard mask of these logits (kernel_head.py:314-317): bit (r & 3) + 8 (r >> 2) of this lane's half word,
                    // the upper half-wave's bits sit 4 places higher; lanes l and l + 32 together hold the row's word
                    uint32_t mk = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int px = (r & 3) + 8 * (r >> 2);
                        const bool on = v[r] > PH_BIN_THR && pxb + px + 4 * g2h < a.HW;
                        mk |= on ? (1u << px) : 0u;
                    }
                    mk <<= 4 * g2h;
                    mk |= (uint32_t)__shfl_xor((int)mk, 32);
                    if (l2 < 32) bitsl[(rt * 32 + l2) * CT + ct] = mk;
                }
#endif
            };
            auto row_tile = [&](int rt) {
                // the bias of a lane's kernel row initialises its accumulators
                const float bz = m == 0 ? 0.f : bias[rt * 32 + (k1_fresh(lane) & 31)];
                if (nct2 == 1) {
                    const f32x16_t acc = k1_gemm_t<E, LD>(a2, T, ct20, k1_fresh(lane), bz);
                    epilogue(acc, rt, ct20);
                } else {
                    for (int cc = 0; cc < nct2; cc += 2) {
                        f32x16_t acc0, acc1;
                        k1_gemm2_t<E, LD>(a2, T, ct20 + cc, k1_fresh(lane), bz, acc0, acc1);
                        epilogue(acc0, rt, ct20 + cc);
                        __builtin_amdgcn_sched_barrier(0);
                        epilogue(acc1, rt, ct20 + cc + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
#ifdef K1_TIMELINE
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) asm volatile("" ::"v"(__builtin_bit_cast(ph_u32x4, a2[ks])));
            K1_STAMP(15);
#endif
            if (act2) row_tile(rt2);
            K1_STAMP(16);
            if constexpr (RT > 1) {
                // further passes (more row tiles of static kernels than WAVES / wpr2): their weights are requested whole
                for (int rt = rt2 + rpass; rt < m2; rt += rpass) {
                    const uint16_t* ap = a.w2[m] + ((int64_t)rt * 16 * 64 + k1_fresh(lane)) * 8;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) a2[ks] = *(const uint4*)(ap + ks * 512);
                    __builtin_amdgcn_sched_barrier(0);
                    row_tile(rt);
                }
            }
            K1_STAMP(17);
            (void)rpass;
        }
        // the conv weights of the next phase: requested behind the slice, needed only when it has landed
        k1_load_a(af, a.wfrag[nm], wave * RT, k1_fresh(lane));
        K1_STAMP(12);
        __syncthreads();                                              // T is free for the next slice; bitsl complete
        K1_STAMP(13);
        if (a.bits && m < 2) {
            const int N = a.n_init + a.n_stuff, t = k1_fresh(tid);
            auto put_row = [&](int dst_row, const uint32_t* w /* null: zeros */) {
                uint32_t* d = a.bits + ((int64_t)b * a.bits_rows + dst_row) * wpr + pair * CT;
                if constexpr (CT == 4) *(uint4*)d = w ? *(const uint4*)w : make_uint4(0, 0, 0, 0);
                else *(uint2*)d = w ? *(const uint2*)w : make_uint2(0, 0);
            };
            if (m == 0) {
                for (int r = t; r < a.n_init; r += G::THREADS) put_row(r, bitsl + r * CT);
            } else {
                for (int r = a.n_init + t; r < a.bits_rows; r += G::THREADS)
                    put_row(r, r < N ? bitsl + (a.stuff_lo + r - a.n_init) * CT : nullptr);
            }
            // bitsl is rewritten only after the next phase's barriers
        }
        K1_STAMP(14);
    }
}

template <int INFMT, int E, typename OutT, bool F32O, typename G>
__global__ __launch_bounds__(G::THREADS, 2) void k_khead_onepass(const K1Args a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // constants of all three phases go to LDS once (see k1_run's rule)
    float* gnl = (float*)((char*)(lds + 256 * G::LD + G::WAVES * 4096) + G::SSB);
    float* b2l = gnl + 3 * 512;
    for (int i = tid; i < 3 * 512; i += G::THREADS) gnl[i] = a.gn[i];
    for (int i = tid; i < 288; i += G::THREADS)
        b2l[i] = i < 256 ? ((a.bias2[1] && i < a.m2_tiles[1] * 32) ? a.bias2[1][i] : 0.f) : (a.bias2[2] ? a.bias2[2][i - 256] : 0.f);
    // a workgroup that starts after the launch has given up (status raised by a workgroup whose wait timed out) leaves at once
    volatile int* abortl = (volatile int*)(b2l + 288 + 64 + 64);
    if (tid == 0) *abortl = __hip_atomic_load((gu32*)a.status, K1_RLX) != 0 ? 1 : 0;
    __syncthreads();
    if (*abortl) return;
    if (wave == 0) k1_run<true, INFMT, E, OutT, F32O, G>(a, lds, tid, lane, wave);
    else k1_run<false, INFMT, E, OutT, F32O, G>(a, lds, tid, lane, wave);
}

__global__ __launch_bounds__(256) void k_k1_clear(uint4* __restrict__ p, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4(0, 0, 0, 0);
}

// ====================================================================================================================
static unsigned long long* g_k1_timeline = nullptr;
// debugging aid (tools): device buffer of [grid][2][3 * rounds][16] uint64 that the next launches fill with s_memtime stamps
extern "C" void ph_khead_onepass_set_timeline(void* buf) { g_k1_timeline = (unsigned long long*)buf; }

static int k1_cus() {
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return cu;
    }();
    return n;
}

extern "C" int ph_khead_onepass_supported(int B, int64_t HW, int groups, int prec, int input_format) {
    if (B <= 0 || B > 4096 || HW <= 0) return 0;
    const int cu = k1_cus();
    const int64_t HWp = ph_hw_padded(HW);
    if (cu <= 0 || HWp / K1Wide::PX > cu) return 0;
    if (groups != 32) return 0;
    if (!(prec == PH_PREC_BF16 || prec == PH_PREC_F16)) return 0;
    if (input_format == PH_IN_F32_NCHW && (HW % 4) != 0) return 0;
    return 1;
}

// hand-off state: [status (256 B)] [gran2: 3B x 128 x 8 B] [gran1: 3B x 64 x P x 8 B], cleared by every call, followed by 256
// bytes the calls never clear (the sticky time-out words; the caller zeroes the workspace once after allocating it)
static size_t k1_zeroed_bytes(int B, int64_t HW) {
    const int64_t P = ph_hw_padded(HW) / K1Pair::PX;                 // the geometry with more slices
    return 256 + (size_t)3 * B * 128 * 8 + (size_t)3 * B * 64 * P * 8;
}
extern "C" size_t ph_khead_onepass_workspace_bytes(int B, int64_t HW) { return k1_zeroed_bytes(B, HW) + 256; }

// bound of one statistics hand-off wait (default 20 ms: an undisturbed hand-off takes ~5 us).  A process-wide knob for tests and
// for deployments that run long kernels beside the head; a time-out costs the two-pass fallback of that call, never the result.
static int g_k1_timeout_us = 20000;
extern "C" void ph_khead_onepass_set_timeout_us(int us) { g_k1_timeout_us = us > 0 ? us : 20000; }

extern "C" int ph_khead_onepass(const void* f0, const void* f1, const void* f2, const uint16_t* conv_frags,
                                const float* gn_affine, int groups, float eps, const uint16_t* w2_init, int n_init,
                                const uint16_t* w2_seg, const float* bias_seg, int n_seg, const uint16_t* w2_dd,
                                const float* bias_dd, int stuff_lo, int n_stuff, uint16_t* x_planes, uint16_t* dfe_planes,
                                float* x_f32, float* dfe_f32, void* mask_preds, void* seg_preds, void* depth_pred,
                                int out_dtype, uint32_t* bits, int bits_rows, void* workspace, size_t workspace_bytes, int B,
                                int64_t HW, int prec, int input_format, void* stream) {
    PH_CHECK_ARG(f0 && f1 && f2 && conv_frags && gn_affine && x_planes && dfe_planes && workspace, "null pointer");
    PH_CHECK_ARG(w2_init && w2_seg && w2_dd && mask_preds && seg_preds && depth_pred, "null pointer");
    PH_CHECK_ARG(input_format == PH_IN_F32_NCHW || input_format == PH_IN_PLANES, "bad input_format");
    PH_CHECK_ARG(out_dtype == PH_OUT_F32 || out_dtype == PH_OUT_F16, "logits: PH_OUT_F32 or PH_OUT_F16");
    PH_CHECK_ARG(n_init > 0 && n_init <= 256 && n_seg > 0 && n_seg <= 256 && n_stuff >= 0 && stuff_lo >= 0 &&
                     stuff_lo + n_stuff <= n_seg, "bad row counts (at most 256 rows per static conv)");
    PH_CHECK_ARG(!bits || (bits_rows >= n_init + n_stuff && bits_rows <= 256), "bits_rows out of range");
    if (!ph_khead_onepass_supported(B, HW, groups, prec, input_format)) {
        ph_set_error("ph_khead_onepass: unsupported geometry or precision (use ph_khead_fused)");
        return PH_EUNSUPPORTED;
    }
    if (workspace_bytes < ph_khead_onepass_workspace_bytes(B, HW)) {
        ph_set_error("ph_khead_onepass: workspace too small");
        return PH_EWORKSPACE;
    }
    const int64_t HWp = ph_hw_padded(HW);
    K1Args a = {};
    a.f[0] = f0; a.f[1] = f1; a.f[2] = f2;
    for (int m = 0; m < 3; ++m) a.wfrag[m] = conv_frags + (size_t)m * 256 * 256;
    a.gn = gn_affine;
    a.w2[0] = w2_init; a.w2[1] = w2_seg; a.w2[2] = w2_dd;
    a.bias2[0] = nullptr; a.bias2[1] = bias_seg; a.bias2[2] = bias_dd;
    a.n2[0] = n_init; a.n2[1] = n_seg; a.n2[2] = 1;
    for (int m = 0; m < 3; ++m) a.m2_tiles[m] = (a.n2[m] + 31) / 32;
    a.out2[0] = mask_preds; a.out2[1] = seg_preds; a.out2[2] = depth_pred;
    a.out2_rows[0] = n_init + n_stuff; a.out2_rows[1] = n_seg; a.out2_rows[2] = 1;
    a.out_dtype = out_dtype;
    a.stuff_lo = stuff_lo; a.n_stuff = n_stuff; a.n_init = n_init;
    a.x_planes = x_planes; a.dfe_planes = dfe_planes; a.x_f32 = x_f32; a.dfe_f32 = dfe_f32;
    a.bits = bits; a.bits_rows = bits_rows;
    char* ws = (char*)workspace;
    a.status = (unsigned*)ws;
    a.gran2 = (unsigned long long*)(ws + 256);
    a.gran1 = a.gran2 + (size_t)3 * B * 128;
    a.sticky = (unsigned*)(ws + k1_zeroed_bytes(B, HW));
    a.timeout_ticks = (unsigned long long)g_k1_timeout_us * K1_TICKS_PER_US;
    a.B = B; a.HW = HW; a.HWp = HWp; a.eps = eps;
    a.timeline = g_k1_timeline;
    hipStream_t s = (hipStream_t)stream;
    const bool planes = input_format == PH_IN_PLANES, h = prec == PH_PREC_F16, o16 = out_dtype == PH_OUT_F16;
    const bool f32o = x_f32 || dfe_f32;          // the variant that also writes fp32 x_feats / depth_feats (the reference API's tensors)
    // The hand-off state is cleared by a kernel of this library, not by hipMemsetAsync: captured into a HIP graph (torch.cuda.graph,
    // ROCm 7.2) the memset node left pointer-like garbage in the first 16 bytes of the region from the second replay on (round 4,
    // scratch/k1_diag.py: status word 0 after eager calls and after replay 0, {0x80600000, 0x7366, ...} after every later replay) --
    // harmless while nothing read the status word before the polls, fatal once late workgroups leave on a raised status.
    {
        const size_t n16 = k1_zeroed_bytes(B, HW) / 16;
        const int blocks = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_k1_clear, dim3(blocks), dim3(256), 0, s, (uint4*)workspace, n16);
    }
    // geometry: one 128-pixel workgroup per CU.  The pair geometry (two 64-pixel workgroups per CU) is compiled only with
    // -DK1_WITH_PAIR (PH_EXTRA_HIPCC_FLAGS) and chosen with PH_KHEAD1_PAIR=1: it passes the same tests and is SLOWER -- cfg2,
    // 16 frames: 1.02 against 0.96 ms; cfg5, 32 frames: 0.61 against 0.53 ms on one box.  A frame's pair-slices are spread
    // over the whole chip, so the two workgroups of a CU sit in the same phase of the same frame and wait for the same
    // statistics together, and the exchange itself grows from 10-11k to 16-20k cycles per phase with twice the workgroups.
    const int cus = k1_cus();
#ifdef K1_WITH_PAIR
    static const bool want_pair = getenv("PH_KHEAD1_PAIR") != nullptr;
#define K1_GO(I_, E_, O_, F_)                                                                                      \
    do {                                                                                                           \
        static const int pair_ok_ = [] {                                                                           \
            (void)hipFuncSetAttribute((const void*)k_khead_onepass<I_, E_, O_, F_, K1Wide>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            (void)hipFuncSetAttribute((const void*)k_khead_onepass<I_, E_, O_, F_, K1Pair>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
            int nb = 0;                                                                                            \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_khead_onepass<I_, E_, O_, F_, K1Pair>, K1Pair::THREADS, \
                                                             K1Pair::LDS_BYTES) != hipSuccess) nb = 0;              \
            return nb >= 2 ? 1 : 0;                                                                                \
        }();                                                                                                       \
        if (pair_ok_ && want_pair) {                                                                               \
            a.P = (int)(HWp / K1Pair::PX);                                                                         \
            a.F = 2 * cus / a.P;                                                                                   \
            if (a.F > B) a.F = B;                                                                                  \
            hipLaunchKernelGGL((k_khead_onepass<I_, E_, O_, F_, K1Pair>), dim3(a.F * a.P), dim3(K1Pair::THREADS), K1Pair::LDS_BYTES, s, a); \
        } else {                                                                                                   \
            a.P = (int)(HWp / K1Wide::PX);                                                                         \
            a.F = cus / a.P;                                                                                       \
            if (a.F > B) a.F = B;                                                                                  \
            hipLaunchKernelGGL((k_khead_onepass<I_, E_, O_, F_, K1Wide>), dim3(a.F * a.P), dim3(K1Wide::THREADS), K1Wide::LDS_BYTES, s, a); \
        }                                                                                                          \
    } while (0)
#else
    a.P = (int)(HWp / K1Wide::PX);
    a.F = cus / a.P;
    if (a.F > B) a.F = B;
#define K1_GO(I_, E_, O_, F_)                                                                                      \
    do {                                                                                                           \
        static const bool once_ = [] {                                                                             \
            (void)hipFuncSetAttribute((const void*)k_khead_onepass<I_, E_, O_, F_, K1Wide>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            return true;                                                                                           \
        }();                                                                                                       \
        (void)once_;                                                                                               \
        hipLaunchKernelGGL((k_khead_onepass<I_, E_, O_, F_, K1Wide>), dim3(a.F * a.P), dim3(K1Wide::THREADS), K1Wide::LDS_BYTES, s, a); \
    } while (0)
#endif
#define K1_GO_F(I, E, O)              \
    do {                              \
        if (f32o) K1_GO(I, E, O, true); \
        else K1_GO(I, E, O, false);   \
    } while (0)
    if (!planes && !h && !o16) K1_GO_F(1, PH_E_BF16, float);
    else if (planes && !h && !o16) K1_GO_F(2, PH_E_BF16, float);
    else if (!planes && h && !o16) K1_GO_F(1, PH_E_F16, float);
    else if (planes && h && !o16) K1_GO_F(2, PH_E_F16, float);
    else if (!planes && !h) K1_GO_F(1, PH_E_BF16, uint16_t);
    else if (planes && !h) K1_GO_F(2, PH_E_BF16, uint16_t);
    else if (!planes) K1_GO_F(1, PH_E_F16, uint16_t);
    else K1_GO_F(2, PH_E_F16, uint16_t);
#undef K1_GO_F
#undef K1_GO
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// 1 if the last ph_khead_onepass call on this workspace gave up (a hand-off wait timed out; its caller's predicated fallback
// then produced the results); synchronises the stream
extern "C" int ph_khead_onepass_status(const void* workspace, int B, void* stream) {
    unsigned v = 0;
    (void)B;
    if (hipMemcpyAsync(&v, workspace, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return (int)v;
}
// number of workgroup time-outs since the workspace was zeroed (sticky: survives the calls' own clearing and graph replays);
// synchronises the stream.  0 = every call so far ran in one pass.
extern "C" int ph_khead_onepass_timeouts(const void* workspace, int B, int64_t HW, void* stream) {
    unsigned v[2] = {0, 0};
    if (hipMemcpyAsync(v, (const char*)workspace + k1_zeroed_bytes(B, HW), 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return (int)v[1];
}
