// libpolyhead, training side (device half): the QUERY SIDE of one KernelUpdateHead stage in training mode --
// KernelUpdator x2, self-attention + LN x2, FFN + LN x2, the cls / mask / depth towers (kernel_update_head.py:245-288,
// funcs/kernel_updator.py:55-93, mmcv MultiheadAttention / FFN) -- as ONE forward call that keeps what the backward needs and ONE
// hand-written backward call that returns the gradient of every parameter and input.  The reference runs this under autograd as
// ~120 ATen / BLAS launches per stage and direction; round 4's training step did the same through torch (hipBLASLt + ATen:
// 70 % of its GPU time).  Here a stage's query side is 19 launches forward and 22 backward, issued from C.
//
// Rows: R = B * N query rows of C = 256 features (row r = b * N + n), two branches (mask / depth) with separate weights that run
// side by side in every launch (a launch = a table of independent jobs).
//
// Arithmetic: fp32 throughout.  The dense products run on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32: exact fp32
// products, fp32 accumulation -- the reference's arithmetic up to the order of the additions); a few thousand rows against 4 MB
// of weights are latency bound, not MFMA bound, so the 1/16 rate of the fp32 pipe against bf16 does not matter here
// (22 GFLOP per training step in all).  Operands are read where they lie (nn.Parameter storage, [out][in] row-major): one tile
// kernel covers the four layouts X W^T, dY W, dY^T X and their accumulate / bias / rank-1 / ReLU-mask epilogues, so no weight is
// ever transposed, packed or copied for training (weights change every step).  LayerNorm / gate / softmax statistics are
// recomputed in the backward from the saved pre-normalisation rows (one wave per row), not stored.
// Summation orders are fixed (no atomics): run-to-run identical results.
#include "ph_common.h"

namespace {

constexpr int QC = 256;            // feature width (in_channels = out_channels = 256 throughout this code base)
constexpr int QHEADS = 8, QD = 32; // attention heads x head width
constexpr float QEPS = 1e-5f;      // LayerNorm eps (torch default, as the reference builds them)

// ===================================================================================================================
// the tile GEMM:  C[m][n] = sum_k A(m,k) B(k,n)  (+ second pair A2 B2) (+ bias[n]) (+ rs_row[m] rs_col[n]) (+ add[m][n])
// 64 x 64 tile per workgroup of 4 waves (2 x 2, each 32 x 32 = 2 x 2 MFMA tiles), 32 k per step through LDS (the loop is bound by
// the latency of the next step's global loads, not by the fp32 matrix pipe: 16 k per step measured 30 us per launch on average).
// kcA / kcB: the operand's contiguous axis is k ("k-contiguous": X[row][k], W[out][in]) or the row / column index.
// ===================================================================================================================
constexpr int GT = 64, GK = 32, GP = 36;       // tile edge, k per step (two 16-k halves), LDS pitch in floats (144-byte rows: 16-byte aligned)
constexpr int GEMM_MAX_JOBS = 10;
enum { GF_RELU = 1, GF_ACCUM = 2, GF_COLSUM_ACCUM = 4 };

struct GemmJob {
    const float *A, *B, *A2, *B2;
    float* C;
    const float *bias, *rs_row, *rs_col, *add, *mask;
    float* colsum;               // [M]: sum_k A(m, k) over the first pair (the bias gradient of a dY^T X product)
    int M, N, K, K2;
    int lda, ldb, lda2, ldb2, ldc, ldadd;
    int kcA, kcB, kcA2, kcB2;
    int rowsA, KA, KB;           // extents the A / B loads may touch (default M, K, K): an operand padded to a multiple of 4 on its
                                 // vectorised axis (d cls, L = 19 / 133 classes) is read past M / K, the other one is not
    int ksplit, flags, tiles, nt;
};
struct GemmBatch {
    GemmJob j[GEMM_MAX_JOBS];
};

// one [64 rows][16 k] HALF of an operand tile: 4 values per thread.  kc: thread (row t >> 2, k (t & 3) * 4 + e); else (k t >> 4, rows (t & 15) * 4 + e)
typedef unsigned qt_u4 __attribute__((ext_vector_type(4)));

// One unconditional 16-byte BUFFER load per call: the descriptor's range check returns zeros for a lane whose offset is pushed
// past the operand's end, so there is no branch and no exec masking around the load (round 5 history: per-element bounds branches
// with scalar fallbacks compiled to 600 load instructions and 1300 branches, ~4000 cycles per 32-k step; a plain load under
// `in ? .. : ..` still became an exec-masked branch, and ANY conditional load -- like __syncthreads()'s fence -- makes the
// compiler's vmcnt bookkeeping fall back to vmcnt(0), which serialises the prefetch ring).  The host guarantees that a float4 is
// wholly inside or wholly outside the operand: base 16-byte aligned, ld % 4 == 0, the vectorised axis (k when the operand is
// k-contiguous, else its row index) a multiple of 4 long (Launcher::job checks).
__device__ __forceinline__ float4 load_op(__amdgpu_buffer_rsrc_t rs, int ld, int kc, int row0, int nrows, int k0, int kend, int t) {
    const int row = kc ? row0 + (t >> 2) : row0 + (t & 15) * 4;
    const int k = kc ? k0 + (t & 3) * 4 : k0 + (t >> 4);
    const bool in = row < nrows && k < kend;
    const unsigned off = in ? 4u * (unsigned)(kc ? row * ld + k : k * ld + row) : 0x7FFFFFF0u;
    const qt_u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    return make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t op_rsrc(const float* P, int ld, int kc, int nrows, int kend) {
    // bytes the operand spans (0 for an absent second pair: every load returns zeros)
    const long n = !P || nrows <= 0 || kend <= 0 ? 0 : (kc ? (long)(nrows - 1) * ld + kend : (long)(kend - 1) * ld + nrows) * 4;
    return __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, (int)n, 0x00020000);
}

// branch-free for both layouts: element e of the thread's run goes to (row + e (1 - kc), k + e kc)
__device__ __forceinline__ void stash_op(float* S, int kc, int t, float4 v, int half) {
    const int r = kc ? (t >> 2) : (t & 15) * 4, k = half * 16 + (kc ? (t & 3) * 4 : (t >> 4));
    const int step = kc ? 1 : GP;
    float* d = S + r * GP + k;
    d[0] = v.x;
    d[step] = v.y;
    d[2 * step] = v.z;
    d[3 * step] = v.w;
}

__global__ __launch_bounds__(256) void k_gemm32(const GemmBatch gb) {
    const GemmJob& J = gb.j[blockIdx.y];
    if ((int)blockIdx.x >= J.tiles) return;
    __shared__ __attribute__((aligned(16))) float As[GT * GP], Bs[GT * GP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    int tile = blockIdx.x;
    const int ks = tile % J.ksplit;
    tile /= J.ksplit;
    const int tn = tile % J.nt, tm = tile / J.nt;
    const int m0 = tm * GT, n0 = tn * GT;
    const int steps1 = (J.K + GK - 1) / GK, steps2 = (J.K2 + GK - 1) / GK;
    int sbeg = 0, send = steps1 + steps2;
    if (J.ksplit > 1) {
        const int per = (steps1 + J.ksplit - 1) / J.ksplit;
        sbeg = ks * per;
        send = min(steps1, sbeg + per);
    }
    const bool want_cs = J.colsum != nullptr && tn == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // The loop is bound by the LATENCY of the operand loads (every launch measured ~24 us whatever its size: 8 steps x ~2.5 us with
    // one step of prefetch), so the loads run PD steps ahead of the MFMAs through a register ring.
    constexpr int PD = 4;
    float4 ra[PD][2], rb[PD][2];
    const float csw = want_cs ? 1.f : 0.f;
    auto fetch = [&](int s, int d) {
        const bool dead = s >= send;                       // a padding step: every lane out of range
        const bool one = dead || s < steps1;               // uniform: first or second operand pair -- selected, not branched on
        const int la = one ? J.lda : J.lda2, lb = one ? J.ldb : J.ldb2, kca = one ? J.kcA : J.kcA2, kcb = one ? J.kcB : J.kcB2;
        const int kea = one ? J.KA : J.K2, keb = one ? J.KB : J.K2, kb = dead ? (1 << 28) : (one ? s : s - steps1) * GK;
        const int ra_rows = one ? J.rowsA : J.M;
        const float w = one && !dead ? csw : 0.f;
        // descriptors from SELECTED scalars (a branch between two loads would make them conditional)
        const __amdgpu_buffer_rsrc_t rsA = op_rsrc(one ? J.A : J.A2, la, kca, ra_rows, kea), rsB = op_rsrc(one ? J.B : J.B2, lb, kcb, J.N, keb);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            ra[d][hf] = load_op(rsA, la, kca, m0, ra_rows, kb + hf * 16, kea, t);
            rb[d][hf] = load_op(rsB, lb, kcb, n0, J.N, kb + hf * 16, keb, t);
            // column sums of the first pair's A (selects, no branch): k-contiguous -> one row per thread, else four rows
            const float4 q = ra[d][hf];
            cs[0] += w * (kca ? (q.x + q.y) + (q.z + q.w) : q.x);
            cs[1] += w * (kca ? 0.f : q.y);
            cs[2] += w * (kca ? 0.f : q.z);
            cs[3] += w * (kca ? 0.f : q.w);
        }
    };
    // No conditional loads: steps past the end read nothing useful (every lane out of range -> zeros) and multiply zeros, so that
    // the compiler's vmcnt bookkeeping stays exact -- with `if (s + PD < send) fetch(...)` it fell back to vmcnt(0) in front of every
    // LDS stash and each step waited for the loads issued one step earlier (1.2 us per step).
    const int send_p = sbeg + (send - sbeg + PD - 1) / PD * PD;
    auto fetch_c = [&](int s, int d) { fetch(s, d); };
#pragma unroll
    for (int d = 0; d < PD; ++d) fetch_c(sbeg + d, d);
    for (int s0 = sbeg; s0 < send_p; s0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int s = s0 + d;
            {
                const bool one = s < steps1;
                const int kca = one ? J.kcA : J.kcA2, kcb = one ? J.kcB : J.kcB2;
                // RAW barriers: __syncthreads() carries a workgroup fence, i.e. s_waitcnt vmcnt(0) -- it would drain the prefetch ring at
                // every step.  Only LDS traffic has to be ordered here: lgkmcnt(0) + s_barrier (the register dependencies of the stash
                // on its own slot's loads are the compiler's counted vmcnt waits).
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();     // every wave is done with the previous step's fragments
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    stash_op(As, kca, t, ra[d][hf], hf);
                    stash_op(Bs, kcb, t, rb[d][hf], hf);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                fetch_c(s + PD, d);
                // k-slot g of MFMA e carries k = 16 hf + 4 g + e of the step (the same permutation for both operands)
#define QT_MF(av, bv, c) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 a0 = *(const float4*)&As[(wm * 32 + li) * GP + hf * 16 + lg * 4];
                    const float4 a1 = *(const float4*)&As[(wm * 32 + 16 + li) * GP + hf * 16 + lg * 4];
                    const float4 b0 = *(const float4*)&Bs[(wn * 32 + li) * GP + hf * 16 + lg * 4];
                    const float4 b1 = *(const float4*)&Bs[(wn * 32 + 16 + li) * GP + hf * 16 + lg * 4];
                    QT_MF(a0.x, b0.x, acc[0][0]); QT_MF(a0.x, b1.x, acc[0][1]); QT_MF(a1.x, b0.x, acc[1][0]); QT_MF(a1.x, b1.x, acc[1][1]);
                    QT_MF(a0.y, b0.y, acc[0][0]); QT_MF(a0.y, b1.y, acc[0][1]); QT_MF(a1.y, b0.y, acc[1][0]); QT_MF(a1.y, b1.y, acc[1][1]);
                    QT_MF(a0.z, b0.z, acc[0][0]); QT_MF(a0.z, b1.z, acc[0][1]); QT_MF(a1.z, b0.z, acc[1][0]); QT_MF(a1.z, b1.z, acc[1][1]);
                    QT_MF(a0.w, b0.w, acc[0][0]); QT_MF(a0.w, b1.w, acc[0][1]); QT_MF(a1.w, b0.w, acc[1][0]); QT_MF(a1.w, b1.w, acc[1][1]);
                }
#undef QT_MF
            }
        }
    }
    // ---- epilogue: D lane (li, lg), reg r -> row lg * 4 + r, col li of its 16 x 16 tile.  Two passes: every load of the extras
    // first, then the stores -- interleaved, the compiler must keep each element's loads behind the previous element's store (the
    // pointers may alias), i.e. 16 dependent memory round trips: every launch measured ~24 us whatever its size.
    const bool first = ks == 0;
    float* Cb = J.C + (J.ksplit > 1 ? (int64_t)ks * J.M * J.ldc : 0);
    float ov[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 32 + b * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + a * 16 + lg * 4 + r;
                float v = acc[a][b][r];
                if (m < J.M && n < J.N) {
                    if (first) {
                        if (J.bias) v += J.bias[n];
                        if (J.rs_row) v += J.rs_row[m] * J.rs_col[n];
                        if (J.add) v += J.add[(int64_t)m * J.ldadd + n];
                    }
                    if (J.flags & GF_RELU) v = fmaxf(v, 0.f);
                    if (J.mask) v = J.mask[(int64_t)m * J.ldc + n] > 0.f ? v : 0.f;
                    if (J.flags & GF_ACCUM) v += Cb[(int64_t)m * J.ldc + n];
                }
                ov[a][b][r] = v;
            }
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 32 + b * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + a * 16 + lg * 4 + r;
                if (m < J.M && n < J.N) Cb[(int64_t)m * J.ldc + n] = ov[a][b][r];
            }
        }
    if (want_cs) {          // sum_k A(m, k): fixed-order reduction of the staging threads' partial sums through LDS
        __syncthreads();
        float* red = As;    // 1024 floats of the 64 x 36
        if (J.kcA) red[t] = cs[0];
        else { red[t * 4 + 0] = cs[0]; red[t * 4 + 1] = cs[1]; red[t * 4 + 2] = cs[2]; red[t * 4 + 3] = cs[3]; }
        __syncthreads();
        if (t < GT && m0 + t < J.M) {
            float s = 0.f;
            if (J.kcA) s = (red[t * 4] + red[t * 4 + 1]) + (red[t * 4 + 2] + red[t * 4 + 3]);
            else
                for (int kk = 0; kk < 16; ++kk) s += red[(kk * 16 + (t >> 2)) * 4 + (t & 3)];
            float* d = J.colsum + m0 + t;
            *d = (J.flags & GF_COLSUM_ACCUM) ? *d + s : s;
        }
    }
}

// ===================================================================================================================
// row-wise kernels: one wave per row, lane l holds features 4 l .. 4 l + 3
// ===================================================================================================================
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4relu(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float4 f4sig(float4 a) { return make_float4(sigm(a.x), sigm(a.y), sigm(a.z), sigm(a.w)); }

struct LNS {
    float mean, rstd;
};
__device__ __forceinline__ LNS ln_stat(float4 x) {
    LNS s;
    s.mean = wsum((x.x + x.y) + (x.z + x.w)) * (1.f / QC);
    const float a = x.x - s.mean, b = x.y - s.mean, c = x.z - s.mean, d = x.w - s.mean;
    const float var = wsum((a * a + b * b) + (c * c + d * d)) * (1.f / QC);
    s.rstd = 1.f / sqrtf(var + QEPS);
    return s;
}
__device__ __forceinline__ float4 ln_hat(float4 x, LNS s) {
    return make_float4((x.x - s.mean) * s.rstd, (x.y - s.mean) * s.rstd, (x.z - s.mean) * s.rstd, (x.w - s.mean) * s.rstd);
}
__device__ __forceinline__ float4 ln_fwd(float4 x, const float* g, const float* b, int c4) {
    const float4 h = ln_hat(x, ln_stat(x)), gg = ld4(g + c4), bb = ld4(b + c4);
    return make_float4(h.x * gg.x + bb.x, h.y * gg.y + bb.y, h.z * gg.z + bb.z, h.w * gg.w + bb.w);
}
// d/dx of y = gamma * xhat + beta given dy: rstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)), dyh = dy * gamma
__device__ __forceinline__ float4 ln_bwd(float4 xh, float rstd, float4 dy, const float* g, int c4) {
    const float4 dh = f4mul(dy, ld4(g + c4));
    const float m1 = wsum((dh.x + dh.y) + (dh.z + dh.w)) * (1.f / QC);
    const float m2 = wsum((dh.x * xh.x + dh.y * xh.y) + (dh.z * xh.z + dh.w * xh.w)) * (1.f / QC);
    return make_float4(rstd * (dh.x - m1 - xh.x * m2), rstd * (dh.y - m1 - xh.y * m2), rstd * (dh.z - m1 - xh.z * m2),
                       rstd * (dh.w - m1 - xh.w * m2));
}

// ---- g = i_in * p_in (kernel_updator.py:69) and its backward ------------------------------------------------------
struct GateArgs {
    const float *p[2], *i[2];     // [R][2C]
    float* g[2];                  // [R][C]
    const float* gg[2];           // backward: dL/dg
    float *gp[2], *gi[2];         // backward: first halves of dL/dp, dL/di ([R][2C])
};
__global__ __launch_bounds__(256) void k_qt_gate(const GateArgs a, int R) {
    const int br = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = (threadIdx.x & 63) * 4;
    if (r >= R) return;
    st4(a.g[br] + (int64_t)r * QC + c4, f4mul(ld4(a.i[br] + (int64_t)r * 2 * QC + c4), ld4(a.p[br] + (int64_t)r * 2 * QC + c4)));
}
__global__ __launch_bounds__(256) void k_qt_gate_bwd(const GateArgs a, int R) {
    const int br = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = (threadIdx.x & 63) * 4;
    if (r >= R) return;
    const float4 gg = ld4(a.gg[br] + (int64_t)r * QC + c4);
    st4(a.gp[br] + (int64_t)r * 2 * QC + c4, f4mul(gg, ld4(a.i[br] + (int64_t)r * 2 * QC + c4)));
    st4(a.gi[br] + (int64_t)r * 2 * QC + c4, f4mul(gg, ld4(a.p[br] + (int64_t)r * 2 * QC + c4)));
}

// ---- the gated update f = sigmoid(LN(b)) * LN(p_out) + sigmoid(LN(a)) * LN(i_out) (kernel_updator.py:73-87) ------
struct UpdArgs {
    const float *a[2], *b[2], *p[2], *i[2];             // a = input_gate(g), b = update_gate(g) [R][C]; p, i [R][2C]
    const float *g_ig[2], *b_ig[2], *g_ug[2], *b_ug[2], *g_po[2], *b_po[2], *g_io[2], *b_io[2];   // LN affine
    float* f[2];
    // backward
    const float* gf[2];
    float *ga[2], *gb[2], *gp[2], *gi[2];               // gp / gi: second halves of [R][2C]
    float* cs[2];                                       // 8 column-sum sources [8][R][C]: (dy * xhat, dy) of the four LayerNorms
};
__global__ __launch_bounds__(256) void k_qt_update(const UpdArgs a, int R) {
    const int br = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = (threadIdx.x & 63) * 4;
    if (r >= R) return;
    const int64_t o = (int64_t)r * QC + c4, o2 = (int64_t)r * 2 * QC + QC + c4;
    const float4 ig = f4sig(ln_fwd(ld4(a.a[br] + o), a.g_ig[br], a.b_ig[br], c4));
    const float4 ug = f4sig(ln_fwd(ld4(a.b[br] + o), a.g_ug[br], a.b_ug[br], c4));
    const float4 po = ln_fwd(ld4(a.p[br] + o2), a.g_po[br], a.b_po[br], c4);
    const float4 io = ln_fwd(ld4(a.i[br] + o2), a.g_io[br], a.b_io[br], c4);
    st4(a.f[br] + o, f4add(f4mul(ug, po), f4mul(ig, io)));
}
__global__ __launch_bounds__(256) void k_qt_update_bwd(const UpdArgs a, int R) {
    const int br = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = (threadIdx.x & 63) * 4;
    if (r >= R) return;
    const int64_t o = (int64_t)r * QC + c4, o2 = (int64_t)r * 2 * QC + QC + c4, RC = (int64_t)R * QC;
    const float4 xa = ld4(a.a[br] + o), xb = ld4(a.b[br] + o), xp = ld4(a.p[br] + o2), xi = ld4(a.i[br] + o2);
    const LNS sa = ln_stat(xa), sb = ln_stat(xb), sp = ln_stat(xp), si = ln_stat(xi);
    const float4 ha = ln_hat(xa, sa), hb = ln_hat(xb, sb), hp = ln_hat(xp, sp), hi = ln_hat(xi, si);
    auto aff = [&](float4 h, const float* g, const float* b) {
        const float4 gg = ld4(g + c4), bb = ld4(b + c4);
        return make_float4(h.x * gg.x + bb.x, h.y * gg.y + bb.y, h.z * gg.z + bb.z, h.w * gg.w + bb.w);
    };
    const float4 ig = f4sig(aff(ha, a.g_ig[br], a.b_ig[br])), ug = f4sig(aff(hb, a.g_ug[br], a.b_ug[br]));
    const float4 po = aff(hp, a.g_po[br], a.b_po[br]), io = aff(hi, a.g_io[br], a.b_io[br]);
    const float4 gf = ld4(a.gf[br] + o);
    auto dsig = [](float4 g, float4 s) {
        return make_float4(g.x * s.x * (1.f - s.x), g.y * s.y * (1.f - s.y), g.z * s.z * (1.f - s.z), g.w * s.w * (1.f - s.w));
    };
    const float4 dya = dsig(f4mul(gf, io), ig), dyb = dsig(f4mul(gf, po), ug), dyp = f4mul(gf, ug), dyi = f4mul(gf, ig);
    float* cs = a.cs[br];
    st4(cs + 0 * RC + o, f4mul(dya, ha)); st4(cs + 1 * RC + o, dya);
    st4(cs + 2 * RC + o, f4mul(dyb, hb)); st4(cs + 3 * RC + o, dyb);
    st4(cs + 4 * RC + o, f4mul(dyp, hp)); st4(cs + 5 * RC + o, dyp);
    st4(cs + 6 * RC + o, f4mul(dyi, hi)); st4(cs + 7 * RC + o, dyi);
    st4(a.ga[br] + o, ln_bwd(ha, sa.rstd, dya, a.g_ig[br], c4));
    st4(a.gb[br] + o, ln_bwd(hb, sb.rstd, dyb, a.g_ug[br], c4));
    st4(a.gp[br] + o2, ln_bwd(hp, sp.rstd, dyp, a.g_po[br], c4));
    st4(a.gi[br] + o2, ln_bwd(hi, si.rstd, dyi, a.g_io[br], c4));
}

// ---- generic: t = sum of `nparts` slices of x (+ xbias) (+ res);  y = [relu](LN(t)) ---------------------------------
constexpr int ROW_JOBS = 3;
struct LnArgs {
    const float* x[ROW_JOBS];       // [nparts][R][C]
    const float* xbias[ROW_JOBS];   // [C] or null (the bias of a split-K product)
    const float* res[ROW_JOBS];     // [R][C] or null
    const float *gamma[ROW_JOBS], *beta[ROW_JOBS];
    float* t[ROW_JOBS];             // pre-normalisation rows kept for the backward (or null: x is kept by the caller)
    float *y[ROW_JOBS], *y2[ROW_JOBS];   // outputs (y2: a second copy handed to the API, or null)
    int nparts[ROW_JOBS], relu[ROW_JOBS];
    // backward: dy = sum of `nparts` slices of dy0 (+ dy1), masked by ysaved > 0 when relu
    const float *dy0[ROW_JOBS], *dy1[ROW_JOBS], *ysaved[ROW_JOBS];
    float *dx[ROW_JOBS], *cs[ROW_JOBS];        // cs: [2][R][C] (dy * xhat, dy)
};
__global__ __launch_bounds__(256) void k_qt_ln(const LnArgs a, int R) {
    const int j = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = (threadIdx.x & 63) * 4;
    if (r >= R) return;
    const int64_t o = (int64_t)r * QC + c4, RC = (int64_t)R * QC;
    float4 t = ld4(a.x[j] + o);
    for (int s = 1; s < a.nparts[j]; ++s) t = f4add(t, ld4(a.x[j] + s * RC + o));
    if (a.xbias[j]) t = f4add(t, ld4(a.xbias[j] + c4));
    if (a.res[j]) t = f4add(t, ld4(a.res[j] + o));
    if (a.t[j]) st4(a.t[j] + o, t);
    float4 y = ln_fwd(t, a.gamma[j], a.beta[j], c4);
    if (a.relu[j]) y = f4relu(y);
    st4(a.y[j] + o, y);
    if (a.y2[j]) st4(a.y2[j] + o, y);
}
__global__ __launch_bounds__(256) void k_qt_ln_bwd(const LnArgs a, int R) {
    const int j = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = (threadIdx.x & 63) * 4;
    if (r >= R) return;
    const int64_t o = (int64_t)r * QC + c4, RC = (int64_t)R * QC;
    float4 dy = ld4(a.dy0[j] + o);
    for (int s = 1; s < a.nparts[j]; ++s) dy = f4add(dy, ld4(a.dy0[j] + s * RC + o));
    if (a.dy1[j]) dy = f4add(dy, ld4(a.dy1[j] + o));
    if (a.relu[j]) {
        const float4 y = ld4(a.ysaved[j] + o);
        dy = make_float4(y.x > 0.f ? dy.x : 0.f, y.y > 0.f ? dy.y : 0.f, y.z > 0.f ? dy.z : 0.f, y.w > 0.f ? dy.w : 0.f);
    }
    const float4 x = ld4(a.x[j] + o);
    const LNS s = ln_stat(x);
    const float4 h = ln_hat(x, s);
    st4(a.cs[j] + o, f4mul(dy, h));
    st4(a.cs[j] + RC + o, dy);
    st4(a.dx[j] + o, ln_bwd(h, s.rstd, dy, a.gamma[j], c4));
}

// out [R][Lp] = in [R][L] with zero columns behind
__global__ __launch_bounds__(256) void k_qt_pad_cols(const float* __restrict__ in, float* __restrict__ out, int R, int L, int Lp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * Lp) return;
    const int r = i / Lp, c = i - r * Lp;
    out[i] = c < L ? in[(int64_t)r * L + c] : 0.f;
}

// ---- kbias[r] = kraw[r] . bt (the scalar bias of the folded dynamic kernel) --------------------------------------------
struct DotArgs {
    const float *x[2], *v[2];
    float* out[2];
};
__global__ __launch_bounds__(256) void k_qt_rowdot(const DotArgs a, int R) {
    const int br = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = (threadIdx.x & 63) * 4;
    if (r >= R) return;
    const float4 x = ld4(a.x[br] + (int64_t)r * QC + c4), v = ld4(a.v[br] + c4);
    const float s = wsum((x.x * v.x + x.y * v.y) + (x.z * v.z + x.w * v.w));
    if ((threadIdx.x & 63) == 0) a.out[br][r] = s;
}

// ---- column sums (parameter gradients that are sums over the rows): dst[c] (+)= sum_r w[r] src[r][c] -------------------
constexpr int CS_MAX_JOBS = 48;
struct CsJob {
    const float *src, *roww;       // roww: [R] or null (weights 1)
    const float *src2, *roww2;     // optional second source (same ld): dst = sum_r w src + sum_r w2 src2
    float* dst;
    int ld, ncols;
};
struct CsBatch {
    CsJob j[CS_MAX_JOBS];
};
__global__ __launch_bounds__(1024) void k_qt_colsum(const CsBatch cb, int R) {
    const CsJob& J = cb.j[blockIdx.y];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;      // 64 columns x 16 row lanes
    __shared__ float red[16][64];
    float s = 0.f;
    if (c < J.ncols) {
        for (int r = rl; r < R; r += 16) s += (J.roww ? J.roww[r] : 1.f) * J.src[(int64_t)r * J.ld + c];
        if (J.src2)
            for (int r = rl; r < R; r += 16) s += (J.roww2 ? J.roww2[r] : 1.f) * J.src2[(int64_t)r * J.ld + c];
    }
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < J.ncols) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x & 63];
        J.dst[c] = v;
    }
}

// ===================================================================================================================
// self-attention over the N query rows of one image, 8 heads of 32 (nn.MultiheadAttention with q = k = v, dropout 0):
// one workgroup (8 waves) per (image, head, branch); q, k, v slices of the [R][768] projection in LDS (pitch 33).
// forward keeps the probabilities P [B][8][N][N]; backward writes dS next to them and returns d(qkv).
// ===================================================================================================================
constexpr int AP = QD + 1;
struct AttArgs {
    const float* qkv[2];     // [R][3C]
    float* prob[2];          // [B][8][N][N]
    float* att[2];           // [R][C]
    const float* gatt[2];    // backward
    float* ds[2];            // [B][8][N][N] scratch
    float* gqkv[2];          // [R][3C]
};
// grid (B * 8, 2 branches, ACH row chunks): a workgroup takes the query rows n = chunk, chunk + ACH, ... -- 32 workgroups of one
// (image, head, branch) each measured 116 us forward / 160 us backward at cfg2 (2 images), latency bound on 32 of 256 CUs
constexpr int ACH = 4;
__global__ __launch_bounds__(512) void k_qt_attn_fwd(const AttArgs a, int N) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / QHEADS, h = blockIdx.x % QHEADS, br = blockIdx.y;
    float *q = sm, *k = q + N * AP, *v = k + N * AP, *prow = v + N * AP;       // prow: [8 waves][N]
    const float* src = a.qkv[br] + (int64_t)b * N * 3 * QC + h * QD;
    const float scale = 0.17677669529663687f;                                   // 1 / sqrt(32): torch scales q first
    for (int e = threadIdx.x; e < N * QD; e += 512) {
        const int n = e >> 5, c = e & 31;
        q[n * AP + c] = src[(int64_t)n * 3 * QC + c] * scale;
        k[n * AP + c] = src[(int64_t)n * 3 * QC + QC + c];
        v[n * AP + c] = src[(int64_t)n * 3 * QC + 2 * QC + c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* P = a.prob[br] + ((int64_t)(b * QHEADS + h) * N) * N;
    float* pr = prow + wave * N;
    for (int n = blockIdx.z + ACH * wave; n < N; n += 8 * ACH) {
        float mx = -INFINITY;
        for (int j = lane; j < N; j += 64) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < QD; ++c) s += q[n * AP + c] * k[j * AP + c];
            pr[j] = s;
            mx = fmaxf(mx, s);
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float den = 0.f;
        for (int j = lane; j < N; j += 64) {
            const float e = expf(pr[j] - mx);
            pr[j] = e;
            den += e;
        }
        den = wsum(den);
        const float inv = 1.f / den;
        for (int j = lane; j < N; j += 64) {
            const float p = pr[j] * inv;
            pr[j] = p;
            P[(int64_t)n * N + j] = p;
        }
        // att[n][c] = sum_j P[n][j] v[j][c]: lane = c + 32 * half, halves over even / odd j
        const int c = lane & 31, hf = lane >> 5;
        float o = 0.f;
        for (int j = hf; j < N; j += 2) o += pr[j] * v[j * AP + c];
        o += __shfl_xor(o, 32);
        if (hf == 0) a.att[br][((int64_t)b * N + n) * QC + h * QD + c] = o;
    }
}

__global__ __launch_bounds__(512) void k_qt_attn_bwd(const AttArgs a, int N) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / QHEADS, h = blockIdx.x % QHEADS, br = blockIdx.y;
    float *q = sm, *k = q + N * AP, *v = k + N * AP, *go = v + N * AP, *prow = go + N * AP;     // prow: [8][N]
    const float* src = a.qkv[br] + (int64_t)b * N * 3 * QC + h * QD;
    const float* gsrc = a.gatt[br] + (int64_t)b * N * QC + h * QD;
    const float scale = 0.17677669529663687f;
    for (int e = threadIdx.x; e < N * QD; e += 512) {
        const int n = e >> 5, c = e & 31;
        q[n * AP + c] = src[(int64_t)n * 3 * QC + c] * scale;
        k[n * AP + c] = src[(int64_t)n * 3 * QC + QC + c];
        v[n * AP + c] = src[(int64_t)n * 3 * QC + 2 * QC + c];
        go[n * AP + c] = gsrc[(int64_t)n * QC + c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* P = a.prob[br] + ((int64_t)(b * QHEADS + h) * N) * N;
    float* dS = a.ds[br] + ((int64_t)(b * QHEADS + h) * N) * N;
    float* gq = a.gqkv[br] + (int64_t)b * N * 3 * QC + h * QD;
    float* pr = prow + wave * N;
    // phase 1, one wave per query row: dP = go v^T, dS = P (dP - sum_j dP P), d q = scale * dS k
    for (int n = blockIdx.z + ACH * wave; n < N; n += 8 * ACH) {
        float dl = 0.f;
        for (int j = lane; j < N; j += 64) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < QD; ++c) s += go[n * AP + c] * v[j * AP + c];
            const float p = P[(int64_t)n * N + j];
            pr[j] = s;
            dl += s * p;
        }
        dl = wsum(dl);
        for (int j = lane; j < N; j += 64) {
            const float d = P[(int64_t)n * N + j] * (pr[j] - dl);
            pr[j] = d;
            dS[(int64_t)n * N + j] = d;
        }
        const int c = lane & 31, hf = lane >> 5;
        float o = 0.f;
        for (int j = hf; j < N; j += 2) o += pr[j] * k[j * AP + c];
        o += __shfl_xor(o, 32);
        if (hf == 0) gq[(int64_t)n * 3 * QC + c] = o * scale;
    }
}

// phase 2 (its own launch: it needs the dS rows of ALL query rows): d k[j][c] = sum_n dS[n][j] q_scaled[n][c];
// d v[j][c] = sum_n P[n][j] go[n][c].  grid (B * 8, 2, ACH): key rows j = chunk, chunk + ACH, ...; thread = (j, quarter of c)
__global__ __launch_bounds__(256) void k_qt_attn_bwd2(const AttArgs a, int N) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / QHEADS, h = blockIdx.x % QHEADS, br = blockIdx.y;
    float *q = sm, *go = q + N * AP;
    const float* src = a.qkv[br] + (int64_t)b * N * 3 * QC + h * QD;
    const float* gsrc = a.gatt[br] + (int64_t)b * N * QC + h * QD;
    const float scale = 0.17677669529663687f;
    for (int e = threadIdx.x; e < N * QD; e += 256) {
        const int n = e >> 5, c = e & 31;
        q[n * AP + c] = src[(int64_t)n * 3 * QC + c] * scale;
        go[n * AP + c] = gsrc[(int64_t)n * QC + c];
    }
    __syncthreads();
    const float* P = a.prob[br] + ((int64_t)(b * QHEADS + h) * N) * N;
    const float* dS = a.ds[br] + ((int64_t)(b * QHEADS + h) * N) * N;
    float* gq = a.gqkv[br] + (int64_t)b * N * 3 * QC + h * QD;
    const int nj = (N - blockIdx.z + ACH - 1) / ACH;                 // key rows of this chunk
    for (int e = threadIdx.x; e < nj * 4; e += 256) {
        const int j = blockIdx.z + ACH * (e % nj), cq = (e / nj) * 8;
        float ak[8], av[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) ak[c] = av[c] = 0.f;
        for (int n = 0; n < N; ++n) {
            const float d = dS[(int64_t)n * N + j], p = P[(int64_t)n * N + j];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                ak[c] += d * q[n * AP + cq + c];
                av[c] += p * go[n * AP + cq + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            gq[(int64_t)j * 3 * QC + QC + cq + c] = ak[c];
            gq[(int64_t)j * 3 * QC + 2 * QC + cq + c] = av[c];
        }
    }
}

// ===================================================================================================================
// host side: the launch sequences
// ===================================================================================================================
// parameter table indices (per branch; include/polyhead.h documents the same list)
enum {
    P_FT_W, P_FT_B, P_DYN_W, P_DYN_B, P_INP_W, P_INP_B, P_IG_W, P_IG_B, P_UG_W, P_UG_B,
    P_LN_IG_G, P_LN_IG_B, P_LN_UG_G, P_LN_UG_B, P_LN_PO_G, P_LN_PO_B, P_LN_IO_G, P_LN_IO_B,
    P_FC_W, P_FC_B, P_LN_FC_G, P_LN_FC_B, P_QKV_W, P_QKV_B, P_OUT_W, P_OUT_B, P_LN_ATT_G, P_LN_ATT_B,
    P_FFN1_W, P_FFN1_B, P_FFN2_W, P_FFN2_B, P_LN_FFN_G, P_LN_FFN_B,
    P_T0_W, P_LN_T0_G, P_LN_T0_B, P_K_W, P_K_B,
    P_T1_W, P_LN_T1_G, P_LN_T1_B, P_CLS_W, P_CLS_B, P_COUNT
};
static_assert(P_COUNT == PH_QTRAIN_NPARAM, "include/polyhead.h: PH_QTRAIN_NPARAM");

constexpr int KSPLIT = 8;

// offsets (floats) into the `saved` buffer of one stage, per branch
struct Saved {
    int64_t U, P, I, G, A, BB, F, H, O1, QKV, ATT, T2, O2, Z, T3, O3, T0P, T0, KRAW, T1P, T1, PROB, PART, per_branch;
    Saved(int B, int N, int Fd) {
        const int64_t R = (int64_t)B * N, RC = R * QC;
        int64_t o = 0;
        auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) / 4 * 4; return at; };
        U = take(RC); P = take(2 * RC); I = take(2 * RC); G = take(RC); A = take(RC); BB = take(RC); F = take(RC); H = take(RC);
        O1 = take(RC); QKV = take(3 * RC); ATT = take(RC); T2 = take(RC); O2 = take(RC); Z = take(R * Fd); T3 = take(RC);
        O3 = take(RC); T0P = take(RC); T0 = take(RC); KRAW = take(RC); T1P = take(RC); T1 = take(RC);
        PROB = take((int64_t)B * QHEADS * N * N); PART = take(KSPLIT * RC);
        per_branch = o;
    }
};
// offsets into the backward's scratch, per branch
struct Scratch {
    int64_t GKRAW, GT0, GT1, GT0P, GT1P, GO3, GT3, GZ, PART, GT2, GATT, GQKV, DS, GO1, GH, GF, GA, GB, GP, GI, GG, GU, GCLSP, CS_UPD,
        CS_LN, per_branch;
    Scratch(int B, int N, int L, int Fd) {
        const int64_t R = (int64_t)B * N, RC = R * QC;
        int64_t o = 0;
        auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) / 4 * 4; return at; };
        GKRAW = take(RC); GT0 = take(RC); GT1 = take(RC); GT0P = take(RC); GT1P = take(RC); GO3 = take(RC); GT3 = take(RC);
        GZ = take(R * Fd); PART = take(KSPLIT * RC); GT2 = take(RC); GATT = take(RC); GQKV = take(3 * RC);
        DS = take((int64_t)B * QHEADS * N * N); GO1 = take(RC); GH = take(RC); GF = take(RC); GA = take(RC); GB = take(RC);
        GP = take(2 * RC); GI = take(2 * RC); GG = take(RC); GU = take(RC); GCLSP = take(R * ((L + 3) / 4 * 4));
        CS_UPD = take(8 * RC);             // the four LayerNorms of the updator
        CS_LN = take(5 * 2 * RC);          // fc_norm, attention_norm, ffn_norm, tower 0, tower 1
        per_branch = o;
    }
};

struct Launcher {
    hipStream_t s;
    int R;
    GemmBatch gb;
    int nj = 0, maxtiles = 0;
    bool failed = false;

    // a float4 of the operand is wholly inside or wholly outside: aligned base, ld % 4 == 0, the vectorised axis (k when the operand is
    // k-contiguous, else its row index) a multiple of 4 long
    static bool vec_ok(const float* p, int ld, int run) { return ((uintptr_t)p & 15) == 0 && (ld % 4) == 0 && (run % 4) == 0; }

    // C[M][N] = A(m,k) B(k,n): kcA / kcB as in GemmJob.  Returns the job for optional extras.
    GemmJob& job(const float* A, int lda, int kcA, const float* B, int ldb, int kcB, float* C, int ldc, int M, int N, int K) {
        if (nj == GEMM_MAX_JOBS) flush();
        GemmJob& j = gb.j[nj++];
        j = GemmJob{};
        j.A = A; j.lda = lda; j.kcA = kcA; j.B = B; j.ldb = ldb; j.kcB = kcB; j.C = C; j.ldc = ldc; j.M = M; j.N = N; j.K = K;
        j.rowsA = M; j.KA = K; j.KB = K;
        if (!vec_ok(A, lda, kcA ? K : M) || !vec_ok(B, ldb, kcB ? K : N)) failed = true;       // callers pad (see `padded`)
        j.ksplit = 1;
        j.kcA2 = j.kcB2 = 1;
        return j;
    }
    static void second(GemmJob& j, const float* A2, int lda2, int kcA2, const float* B2, int ldb2, int kcB2, int K2) {
        j.A2 = A2; j.lda2 = lda2; j.kcA2 = kcA2; j.B2 = B2; j.ldb2 = ldb2; j.kcB2 = kcB2; j.K2 = K2;
    }
    // operand A is a buffer padded on its vectorised axis (zeros behind the data): let its loads run to the padded extent
    void padded_A(GemmJob& j, int rows_padded, int k_padded) {
        j.rowsA = rows_padded; j.KA = k_padded;
        if (k_padded > j.K) j.K = k_padded;
        failed = false;
        for (int i = 0; i < nj; ++i) {
            const GemmJob& q = gb.j[i];
            if (!vec_ok(q.A, q.lda, q.kcA ? q.KA : q.rowsA) || !vec_ok(q.B, q.ldb, q.kcB ? q.KB : q.N)) failed = true;
        }
    }
    void flush() {
        if (!nj) return;
        maxtiles = 0;
        for (int i = 0; i < nj; ++i) {
            GemmJob& j = gb.j[i];
            j.nt = (j.N + GT - 1) / GT;
            j.tiles = ((j.M + GT - 1) / GT) * j.nt * j.ksplit;
            if (j.tiles > maxtiles) maxtiles = j.tiles;
        }
        hipLaunchKernelGGL(k_gemm32, dim3(maxtiles, nj), dim3(256), 0, s, gb);
        if (hipGetLastError() != hipSuccess) failed = true;
        nj = 0;
    }
    dim3 rows(int jobs) const { return dim3((R + 3) / 4, jobs); }
};

}  // namespace

extern "C" size_t ph_qtrain_saved_floats(int B, int N, int L, int F) { return (size_t)(2 * Saved(B, N, F).per_branch); }
extern "C" size_t ph_qtrain_scratch_floats(int B, int N, int L, int F) { return (size_t)(2 * Scratch(B, N, L, F).per_branch); }

// the attention backward keeps q, k, v, d out of one (image, head) + 8 probability rows in LDS: (4 * 33 + 8) * 4 * N bytes <= 160 KB
#define QT_ARGS_OK(B, N, L, F) (B > 0 && N > 0 && N <= 280 && L > 0 && F > 0 && F % 4 == 0)

extern "C" int ph_qtrain_forward(const float* const* params, const float* pooled, const float* cnt, const float* k, const float* q,
                                 float* cls, float* kern, float* kbias, float* obj, float* saved, int B, int N, int L, int F,
                                 void* stream) {
    PH_CHECK_ARG(params && pooled && cnt && k && q && cls && kern && kbias && obj && saved, "null pointer");
    PH_CHECK_ARG(QT_ARGS_OK(B, N, L, F), "sizes: N <= 280 queries, ffn width a multiple of 4");
    for (int br = 0; br < 2; ++br)
        for (int i = 0; i < P_COUNT; ++i)
            PH_CHECK_ARG(params[br * P_COUNT + i] || (br == 1 && i >= P_T1_W), "null parameter pointer");
    const int R = B * N, C = QC;
    const int64_t RC = (int64_t)R * C;
    const Saved sv(B, N, F);
    Launcher Lh;
    Lh.s = (hipStream_t)stream;
    Lh.R = R;
    auto Pm = [&](int br, int i) { return params[br * P_COUNT + i]; };
    auto S = [&](int br, int64_t off) { return saved + br * sv.per_branch + off; };
    const float* kin[2] = {k, q};
    // F0: u = pooled W_t^T + cnt (x) b_t (feat_transform folded: pooling is linear in it, kernel_update_head.py:225,241);
    //     i = input_layer(k)  (depth branch: k_depth = q + k.detach(), :250 -- as a second operand pair on the same weights)
    for (int br = 0; br < 2; ++br) {
        GemmJob& j = Lh.job(pooled + br * RC, C, 1, Pm(br, P_FT_W), C, 1, S(br, sv.U), C, R, C, C);
        j.rs_row = cnt; j.rs_col = Pm(br, P_FT_B);
        GemmJob& ji = Lh.job(kin[br], C, 1, Pm(br, P_INP_W), C, 1, S(br, sv.I), 2 * C, R, 2 * C, C);
        ji.bias = Pm(br, P_INP_B);
        if (br == 1) Launcher::second(ji, k, C, 1, Pm(br, P_INP_W), C, 1, C);
    }
    Lh.flush();
    // F1: p = dynamic_layer(u)
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.U), C, 1, Pm(br, P_DYN_W), C, 1, S(br, sv.P), 2 * C, R, 2 * C, C).bias = Pm(br, P_DYN_B);
    Lh.flush();
    {   // R1: g = i_in * p_in
        GateArgs a{};
        for (int br = 0; br < 2; ++br) { a.p[br] = S(br, sv.P); a.i[br] = S(br, sv.I); a.g[br] = S(br, sv.G); }
        hipLaunchKernelGGL(k_qt_gate, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // F2: input_gate(g), update_gate(g)
    for (int br = 0; br < 2; ++br) {
        Lh.job(S(br, sv.G), C, 1, Pm(br, P_IG_W), C, 1, S(br, sv.A), C, R, C, C).bias = Pm(br, P_IG_B);
        Lh.job(S(br, sv.G), C, 1, Pm(br, P_UG_W), C, 1, S(br, sv.BB), C, R, C, C).bias = Pm(br, P_UG_B);
    }
    Lh.flush();
    {   // R2: f
        UpdArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.a[br] = S(br, sv.A); a.b[br] = S(br, sv.BB); a.p[br] = S(br, sv.P); a.i[br] = S(br, sv.I); a.f[br] = S(br, sv.F);
            a.g_ig[br] = Pm(br, P_LN_IG_G); a.b_ig[br] = Pm(br, P_LN_IG_B); a.g_ug[br] = Pm(br, P_LN_UG_G); a.b_ug[br] = Pm(br, P_LN_UG_B);
            a.g_po[br] = Pm(br, P_LN_PO_G); a.b_po[br] = Pm(br, P_LN_PO_B); a.g_io[br] = Pm(br, P_LN_IO_G); a.b_io[br] = Pm(br, P_LN_IO_B);
        }
        hipLaunchKernelGGL(k_qt_update, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // F3: h = fc_layer(f);  R3: o1 = relu(fc_norm(h))
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.F), C, 1, Pm(br, P_FC_W), C, 1, S(br, sv.H), C, R, C, C).bias = Pm(br, P_FC_B);
    Lh.flush();
    {
        LnArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.x[br] = S(br, sv.H); a.nparts[br] = 1; a.gamma[br] = Pm(br, P_LN_FC_G); a.beta[br] = Pm(br, P_LN_FC_B);
            a.relu[br] = 1; a.y[br] = S(br, sv.O1);
        }
        hipLaunchKernelGGL(k_qt_ln, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // F4: qkv;  A: attention;  F5: out_proj;  R5: o2 = attention_norm(o1 + y)
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.O1), C, 1, Pm(br, P_QKV_W), C, 1, S(br, sv.QKV), 3 * C, R, 3 * C, C).bias = Pm(br, P_QKV_B);
    Lh.flush();
    {
        AttArgs a{};
        for (int br = 0; br < 2; ++br) { a.qkv[br] = S(br, sv.QKV); a.prob[br] = S(br, sv.PROB); a.att[br] = S(br, sv.ATT); }
        const size_t lds = (size_t)(3 * N * AP + 8 * N) * 4;
        static const hipError_t attr = hipFuncSetAttribute((const void*)k_qt_attn_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)attr;
        hipLaunchKernelGGL(k_qt_attn_fwd, dim3(B * QHEADS, 2, ACH), dim3(512), lds, Lh.s, a, N);
    }
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.ATT), C, 1, Pm(br, P_OUT_W), C, 1, S(br, sv.PART), C, R, C, C).bias = Pm(br, P_OUT_B);
    Lh.flush();
    {
        LnArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.x[br] = S(br, sv.PART); a.nparts[br] = 1; a.res[br] = S(br, sv.O1); a.t[br] = S(br, sv.T2);
            a.gamma[br] = Pm(br, P_LN_ATT_G); a.beta[br] = Pm(br, P_LN_ATT_B); a.y[br] = S(br, sv.O2);
        }
        hipLaunchKernelGGL(k_qt_ln, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // F6: z = relu(ffn.layers.0.0(o2));  F7: layers.1(z) split over k;  R7: o3 = ffn_norm(o2 + sum of the splits + bias)
    for (int br = 0; br < 2; ++br) {
        GemmJob& j = Lh.job(S(br, sv.O2), C, 1, Pm(br, P_FFN1_W), C, 1, S(br, sv.Z), F, R, F, C);
        j.bias = Pm(br, P_FFN1_B); j.flags = GF_RELU;
    }
    Lh.flush();
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.Z), F, 1, Pm(br, P_FFN2_W), F, 1, S(br, sv.PART), C, R, C, F).ksplit = KSPLIT;
    Lh.flush();
    {
        LnArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.x[br] = S(br, sv.PART); a.nparts[br] = KSPLIT; a.xbias[br] = Pm(br, P_FFN2_B); a.res[br] = S(br, sv.O2); a.t[br] = S(br, sv.T3);
            a.gamma[br] = Pm(br, P_LN_FFN_G); a.beta[br] = Pm(br, P_LN_FFN_B); a.y[br] = S(br, sv.O3); a.y2[br] = obj + br * RC;
        }
        hipLaunchKernelGGL(k_qt_ln, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // F8 / R8: the towers -- mask_fcs (LN + ReLU), depth_regs (LN, NO activation: kernel_update_head.py:182-187), cls_fcs (LN + ReLU)
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.O3), C, 1, Pm(br, P_T0_W), C, 1, S(br, sv.T0P), C, R, C, C);
    Lh.job(S(0, sv.O3), C, 1, Pm(0, P_T1_W), C, 1, S(0, sv.T1P), C, R, C, C);
    Lh.flush();
    {
        LnArgs a{};
        for (int j = 0; j < 3; ++j) {
            const int br = j == 1 ? 1 : 0;
            a.x[j] = S(br, j == 2 ? sv.T1P : sv.T0P); a.nparts[j] = 1; a.relu[j] = j != 1;
            a.gamma[j] = Pm(br, j == 2 ? P_LN_T1_G : P_LN_T0_G); a.beta[j] = Pm(br, j == 2 ? P_LN_T1_B : P_LN_T0_B);
            a.y[j] = S(br, j == 2 ? sv.T1 : sv.T0);
        }
        hipLaunchKernelGGL(k_qt_ln, Lh.rows(3), dim3(256), 0, Lh.s, a, R);
    }
    // F9: fc_mask / fc_depth / fc_cls
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.T0), C, 1, Pm(br, P_K_W), C, 1, S(br, sv.KRAW), C, R, C, C).bias = Pm(br, P_K_B);
    Lh.job(S(0, sv.T1), C, 1, Pm(0, P_CLS_W), C, 1, cls, L, R, L, C).bias = Pm(0, P_CLS_B);
    Lh.flush();
    // F10: the dynamic kernel folded with feat_transform: conv(W_t x + b_t, kraw) = (kraw W_t) x + kraw . b_t (:317-329)
    for (int br = 0; br < 2; ++br) Lh.job(S(br, sv.KRAW), C, 1, Pm(br, P_FT_W), C, 0, kern + br * RC, C, R, C, C);
    Lh.flush();
    {
        DotArgs a{};
        for (int br = 0; br < 2; ++br) { a.x[br] = S(br, sv.KRAW); a.v[br] = Pm(br, P_FT_B); a.out[br] = kbias + br * R; }
        hipLaunchKernelGGL(k_qt_rowdot, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    PH_CHECK_LAUNCH();
    if (Lh.failed) { ph_set_error("ph_qtrain_forward: a launch failed"); return PH_ELAUNCH; }
    return PH_OK;
}

extern "C" int ph_qtrain_backward(const float* const* params, const float* pooled, const float* cnt, const float* k, const float* q,
                                  const float* saved, const float* g_cls, const float* g_kern, const float* g_kbias,
                                  const float* g_obj, float* const* grads, float* g_pooled, float* g_k, float* g_q,
                                  float* scratch, int B, int N, int L, int F, void* stream) {
    PH_CHECK_ARG(params && pooled && cnt && k && q && saved && g_cls && g_kern && g_kbias && g_obj && grads && g_pooled && g_k && g_q && scratch,
                 "null pointer");
    PH_CHECK_ARG(QT_ARGS_OK(B, N, L, F), "sizes: N <= 280 queries, ffn width a multiple of 4");
    for (int br = 0; br < 2; ++br)
        for (int i = 0; i < P_COUNT; ++i)
            PH_CHECK_ARG((params[br * P_COUNT + i] && grads[br * P_COUNT + i]) || (br == 1 && i >= P_T1_W), "null parameter / gradient pointer");
    const int R = B * N, C = QC;
    const int64_t RC = (int64_t)R * C;
    const Saved sv(B, N, F);
    const Scratch sc(B, N, L, F);
    Launcher Lh;
    Lh.s = (hipStream_t)stream;
    Lh.R = R;
    auto Pm = [&](int br, int i) { return params[br * P_COUNT + i]; };
    auto Gd = [&](int br, int i) { return grads[br * P_COUNT + i]; };
    auto S = [&](int br, int64_t off) { return saved + br * sv.per_branch + off; };
    auto X = [&](int br, int64_t off) { return scratch + br * sc.per_branch + off; };
    const float* kin[2] = {k, q};
    CsBatch cb;
    int ncs = 0;
    auto colsum = [&](const float* src, const float* roww, float* dst, const float* src2 = nullptr, const float* roww2 = nullptr) {
        if (ncs >= CS_MAX_JOBS) { ++ncs; return; }
        CsJob& j = cb.j[ncs++];
        j.src = src; j.ld = QC; j.roww = roww; j.src2 = src2; j.roww2 = roww2; j.dst = dst; j.ncols = QC;
    };
    // dW[out][in] = dY^T X  (+ the bias gradient = column sums of dY, from the same launch)
    auto dweight = [&](const float* dY, int ldy, const float* Xr, int ldx, float* dW, int Mout, int Nin, float* dbias) -> GemmJob& {
        GemmJob& j = Lh.job(dY, ldy, 0, Xr, ldx, 0, dW, Nin, Mout, Nin, R);
        j.colsum = dbias;
        return j;
    };
    // L1: d kraw = d kern W_t^T + d kbias (x) b_t   (feat_transform's own gradients: L11, both of its uses together)
    for (int br = 0; br < 2; ++br) {
        GemmJob& j = Lh.job(g_kern + br * RC, C, 1, Pm(br, P_FT_W), C, 1, X(br, sc.GKRAW), C, R, C, C);
        j.rs_row = g_kbias + br * R; j.rs_col = Pm(br, P_FT_B);
    }
    Lh.flush();
    // L2: through fc_mask / fc_depth / fc_cls
    for (int br = 0; br < 2; ++br) {
        Lh.job(X(br, sc.GKRAW), C, 1, Pm(br, P_K_W), C, 0, X(br, sc.GT0), C, R, C, C);
        dweight(X(br, sc.GKRAW), C, S(br, sv.T0), C, Gd(br, P_K_W), C, C, Gd(br, P_K_B));
    }
    {   // d cls [R][L] (L = 19 / 133: no multiple of 4) goes through a zero-padded copy [R][Lp] so that every operand load of the
        // tile GEMM is one aligned 16-byte load
        const int Lp = (L + 3) / 4 * 4;
        float* gcp = X(0, sc.GCLSP);
        hipLaunchKernelGGL(k_qt_pad_cols, dim3((R * Lp + 255) / 256), dim3(256), 0, Lh.s, g_cls, gcp, R, L, Lp);
        GemmJob& j1 = Lh.job(gcp, Lp, 1, Pm(0, P_CLS_W), C, 0, X(0, sc.GT1), C, R, C, L);
        j1.KB = L;                                   // fc_cls.weight has L rows; d cls is padded to Lp columns of zeros
        Lh.padded_A(j1, R, Lp);
        GemmJob& j2 = dweight(gcp, Lp, S(0, sv.T1), C, Gd(0, P_CLS_W), L, C, Gd(0, P_CLS_B));
        Lh.padded_A(j2, Lp, R);                      // A = d cls^T: rows (= classes) padded to Lp; stores and column sums stop at L
    }
    Lh.flush();
    {   // R8b: tower LayerNorms (+ ReLU)
        LnArgs a{};
        for (int j = 0; j < 3; ++j) {
            const int br = j == 1 ? 1 : 0;
            a.x[j] = S(br, j == 2 ? sv.T1P : sv.T0P); a.nparts[j] = 1; a.relu[j] = j != 1; a.ysaved[j] = S(br, j == 2 ? sv.T1 : sv.T0);
            a.gamma[j] = Pm(br, j == 2 ? P_LN_T1_G : P_LN_T0_G);
            a.dy0[j] = X(br, j == 2 ? sc.GT1 : sc.GT0); a.dx[j] = X(br, j == 2 ? sc.GT1P : sc.GT0P);
            a.cs[j] = X(br, sc.CS_LN + (j == 2 ? 4 : 3) * 2 * RC);
            colsum(a.cs[j], nullptr, Gd(br, j == 2 ? P_LN_T1_G : P_LN_T0_G));
            colsum(a.cs[j] + RC, nullptr, Gd(br, j == 2 ? P_LN_T1_B : P_LN_T0_B));
        }
        hipLaunchKernelGGL(k_qt_ln_bwd, Lh.rows(3), dim3(256), 0, Lh.s, a, R);
    }
    // L3: d o3 = d obj + d t0pre W_t0 (+ d t1pre W_t1);  tower weights
    for (int br = 0; br < 2; ++br) {
        GemmJob& j = Lh.job(X(br, sc.GT0P), C, 1, Pm(br, P_T0_W), C, 0, X(br, sc.GO3), C, R, C, C);
        j.add = g_obj + br * RC; j.ldadd = C;
        if (br == 0) Launcher::second(j, X(0, sc.GT1P), C, 1, Pm(0, P_T1_W), C, 0, C);
        dweight(X(br, sc.GT0P), C, S(br, sv.O3), C, Gd(br, P_T0_W), C, C, nullptr);
    }
    dweight(X(0, sc.GT1P), C, S(0, sv.O3), C, Gd(0, P_T1_W), C, C, nullptr);
    Lh.flush();
    {   // R7b: ffn_norm
        LnArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.x[br] = S(br, sv.T3); a.nparts[br] = 1; a.gamma[br] = Pm(br, P_LN_FFN_G); a.dy0[br] = X(br, sc.GO3); a.dx[br] = X(br, sc.GT3);
            a.cs[br] = X(br, sc.CS_LN + 2 * 2 * RC);
            colsum(a.cs[br], nullptr, Gd(br, P_LN_FFN_G));
            colsum(a.cs[br] + RC, nullptr, Gd(br, P_LN_FFN_B));
        }
        hipLaunchKernelGGL(k_qt_ln_bwd, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // L4: d z = (d t3 W_2) masked by z > 0;  d W_2 = d t3^T z
    for (int br = 0; br < 2; ++br) {
        GemmJob& j = Lh.job(X(br, sc.GT3), C, 1, Pm(br, P_FFN2_W), F, 0, X(br, sc.GZ), F, R, F, C);
        j.mask = S(br, sv.Z);
        dweight(X(br, sc.GT3), C, S(br, sv.Z), F, Gd(br, P_FFN2_W), C, F, Gd(br, P_FFN2_B));
    }
    Lh.flush();
    // L5: d o2 (FFN part, split over k) = d z W_1;  d W_1 = d z^T o2
    for (int br = 0; br < 2; ++br) {
        Lh.job(X(br, sc.GZ), F, 1, Pm(br, P_FFN1_W), C, 0, X(br, sc.PART), C, R, C, F).ksplit = KSPLIT;
        dweight(X(br, sc.GZ), F, S(br, sv.O2), C, Gd(br, P_FFN1_W), F, C, Gd(br, P_FFN1_B));
    }
    Lh.flush();
    {   // R5b: attention_norm; d o2 = d t3 + the splits
        LnArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.x[br] = S(br, sv.T2); a.nparts[br] = KSPLIT; a.gamma[br] = Pm(br, P_LN_ATT_G); a.dy0[br] = X(br, sc.PART); a.dy1[br] = X(br, sc.GT3);
            a.dx[br] = X(br, sc.GT2); a.cs[br] = X(br, sc.CS_LN + 1 * 2 * RC);
            colsum(a.cs[br], nullptr, Gd(br, P_LN_ATT_G));
            colsum(a.cs[br] + RC, nullptr, Gd(br, P_LN_ATT_B));
        }
        hipLaunchKernelGGL(k_qt_ln_bwd, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // L6: out_proj
    for (int br = 0; br < 2; ++br) {
        Lh.job(X(br, sc.GT2), C, 1, Pm(br, P_OUT_W), C, 0, X(br, sc.GATT), C, R, C, C);
        dweight(X(br, sc.GT2), C, S(br, sv.ATT), C, Gd(br, P_OUT_W), C, C, Gd(br, P_OUT_B));
    }
    Lh.flush();
    {
        AttArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.qkv[br] = S(br, sv.QKV); a.prob[br] = (float*)S(br, sv.PROB); a.gatt[br] = X(br, sc.GATT); a.ds[br] = X(br, sc.DS);
            a.gqkv[br] = X(br, sc.GQKV);
        }
        const size_t lds = (size_t)(4 * N * AP + 8 * N) * 4;
        static const hipError_t attr = hipFuncSetAttribute((const void*)k_qt_attn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)attr;
        hipLaunchKernelGGL(k_qt_attn_bwd, dim3(B * QHEADS, 2, ACH), dim3(512), lds, Lh.s, a, N);
        static const hipError_t attr2 = hipFuncSetAttribute((const void*)k_qt_attn_bwd2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)attr2;
        hipLaunchKernelGGL(k_qt_attn_bwd2, dim3(B * QHEADS, 2, ACH), dim3(256), (size_t)(2 * N * AP) * 4, Lh.s, a, N);
    }
    // L7: in_proj; d o1 = d t2 + d qkv W_in
    for (int br = 0; br < 2; ++br) {
        GemmJob& j = Lh.job(X(br, sc.GQKV), 3 * C, 1, Pm(br, P_QKV_W), C, 0, X(br, sc.GO1), C, R, C, 3 * C);
        j.add = X(br, sc.GT2); j.ldadd = C;
        dweight(X(br, sc.GQKV), 3 * C, S(br, sv.O1), C, Gd(br, P_QKV_W), 3 * C, C, Gd(br, P_QKV_B));
    }
    Lh.flush();
    {   // R3b: fc_norm + ReLU
        LnArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.x[br] = S(br, sv.H); a.nparts[br] = 1; a.relu[br] = 1; a.ysaved[br] = S(br, sv.O1); a.gamma[br] = Pm(br, P_LN_FC_G);
            a.dy0[br] = X(br, sc.GO1); a.dx[br] = X(br, sc.GH); a.cs[br] = X(br, sc.CS_LN + 0 * 2 * RC);
            colsum(a.cs[br], nullptr, Gd(br, P_LN_FC_G));
            colsum(a.cs[br] + RC, nullptr, Gd(br, P_LN_FC_B));
        }
        hipLaunchKernelGGL(k_qt_ln_bwd, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // L8: fc_layer
    for (int br = 0; br < 2; ++br) {
        Lh.job(X(br, sc.GH), C, 1, Pm(br, P_FC_W), C, 0, X(br, sc.GF), C, R, C, C);
        dweight(X(br, sc.GH), C, S(br, sv.F), C, Gd(br, P_FC_W), C, C, Gd(br, P_FC_B));
    }
    Lh.flush();
    {   // R2b: the gated update
        UpdArgs a{};
        static const int lnp[4][2] = {{P_LN_IG_G, P_LN_IG_B}, {P_LN_UG_G, P_LN_UG_B}, {P_LN_PO_G, P_LN_PO_B}, {P_LN_IO_G, P_LN_IO_B}};
        for (int br = 0; br < 2; ++br) {
            a.a[br] = S(br, sv.A); a.b[br] = S(br, sv.BB); a.p[br] = S(br, sv.P); a.i[br] = S(br, sv.I);
            a.g_ig[br] = Pm(br, P_LN_IG_G); a.b_ig[br] = Pm(br, P_LN_IG_B); a.g_ug[br] = Pm(br, P_LN_UG_G); a.b_ug[br] = Pm(br, P_LN_UG_B);
            a.g_po[br] = Pm(br, P_LN_PO_G); a.b_po[br] = Pm(br, P_LN_PO_B); a.g_io[br] = Pm(br, P_LN_IO_G); a.b_io[br] = Pm(br, P_LN_IO_B);
            a.gf[br] = X(br, sc.GF); a.ga[br] = X(br, sc.GA); a.gb[br] = X(br, sc.GB); a.gp[br] = X(br, sc.GP); a.gi[br] = X(br, sc.GI);
            a.cs[br] = X(br, sc.CS_UPD);
            for (int n = 0; n < 4; ++n) {
                colsum(a.cs[br] + (2 * n) * RC, nullptr, Gd(br, lnp[n][0]));
                colsum(a.cs[br] + (2 * n + 1) * RC, nullptr, Gd(br, lnp[n][1]));
            }
        }
        hipLaunchKernelGGL(k_qt_update_bwd, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // L9: the gates; d g = d a W_ig + d b W_ug
    for (int br = 0; br < 2; ++br) {
        GemmJob& j = Lh.job(X(br, sc.GA), C, 1, Pm(br, P_IG_W), C, 0, X(br, sc.GG), C, R, C, C);
        Launcher::second(j, X(br, sc.GB), C, 1, Pm(br, P_UG_W), C, 0, C);
        dweight(X(br, sc.GA), C, S(br, sv.G), C, Gd(br, P_IG_W), C, C, Gd(br, P_IG_B));
        dweight(X(br, sc.GB), C, S(br, sv.G), C, Gd(br, P_UG_W), C, C, Gd(br, P_UG_B));
    }
    Lh.flush();
    {   // R1b: first halves of d p, d i
        GateArgs a{};
        for (int br = 0; br < 2; ++br) {
            a.p[br] = S(br, sv.P); a.i[br] = S(br, sv.I); a.gg[br] = X(br, sc.GG); a.gp[br] = X(br, sc.GP); a.gi[br] = X(br, sc.GI);
        }
        hipLaunchKernelGGL(k_qt_gate_bwd, Lh.rows(2), dim3(256), 0, Lh.s, a, R);
    }
    // L10: dynamic_layer / input_layer
    float* gkin[2] = {g_k, g_q};
    for (int br = 0; br < 2; ++br) {
        Lh.job(X(br, sc.GP), 2 * C, 1, Pm(br, P_DYN_W), C, 0, X(br, sc.GU), C, R, C, 2 * C);
        Lh.job(X(br, sc.GI), 2 * C, 1, Pm(br, P_INP_W), C, 0, gkin[br], C, R, C, 2 * C);
        dweight(X(br, sc.GP), 2 * C, S(br, sv.U), C, Gd(br, P_DYN_W), 2 * C, C, Gd(br, P_DYN_B));
        GemmJob& j = dweight(X(br, sc.GI), 2 * C, kin[br], C, Gd(br, P_INP_W), 2 * C, C, Gd(br, P_INP_B));
        if (br == 1) Launcher::second(j, X(br, sc.GI), 2 * C, 0, k, C, 0, R);       // k_depth = q + k.detach(): no gradient into k
    }
    Lh.flush();
    // L11: d pooled = d u W_t;  feat_transform's gradients over BOTH of its uses (pooled features in, dynamic kernel out):
    //      d W_t = d u^T pooled + kraw^T d kern,   d b_t = sum_r cnt[r] d u[r] + sum_r d kbias[r] kraw[r]
    for (int br = 0; br < 2; ++br) {
        Lh.job(X(br, sc.GU), C, 1, Pm(br, P_FT_W), C, 0, g_pooled + br * RC, C, R, C, C);
        GemmJob& j = Lh.job(X(br, sc.GU), C, 0, pooled + br * RC, C, 0, Gd(br, P_FT_W), C, C, C, R);
        Launcher::second(j, S(br, sv.KRAW), C, 0, g_kern + br * RC, C, 0, R);
        colsum(X(br, sc.GU), cnt, Gd(br, P_FT_B), S(br, sv.KRAW), g_kbias + br * R);
    }
    Lh.flush();
    if (ncs > CS_MAX_JOBS) { ph_set_error("ph_qtrain_backward: column-sum table overflow"); return PH_EINVAL; }
    hipLaunchKernelGGL(k_qt_colsum, dim3((QC + 63) / 64, ncs), dim3(1024), 0, Lh.s, cb, R);
    PH_CHECK_LAUNCH();
    if (Lh.failed) { ph_set_error("ph_qtrain_backward: a launch failed"); return PH_ELAUNCH; }
    return PH_OK;
}

// one product of the tile GEMM on its own (tests / tools/gemm32_time.py): C [M][N] = A(m, k) B(k, n) (+ bias[n]); kcA / kcB != 0: the
// operand's contiguous axis is k (X [row][k], W [out][in]); ksplit > 1: C is [ksplit][M][ldc] partial results
extern "C" int ph_gemm32(const float* A, int lda, int kcA, const float* B, int ldb, int kcB, float* C, int ldc, int M, int N, int K,
                         int ksplit, const float* bias, void* stream) {
    PH_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && ksplit >= 1, "bad pointer or size");
    Launcher Lh;
    Lh.s = (hipStream_t)stream;
    Lh.R = M;
    GemmJob& j = Lh.job(A, lda, kcA, B, ldb, kcB, C, ldc, M, N, K);
    j.bias = bias;
    j.ksplit = ksplit;
    Lh.flush();
    PH_CHECK_LAUNCH();
    return Lh.failed ? PH_ELAUNCH : PH_OK;
}
