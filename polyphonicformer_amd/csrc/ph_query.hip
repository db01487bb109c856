// A8-A12 -- the query side of one KernelUpdateHead stage (kernel_update_head.py:245-288,
// funcs/kernel_updator.py:55-93) as two kernels:
//
//   k_query_pre  : deterministic reduce of the pooling partials, KernelUpdator for both branches,
//                  attention in-projection (q scaled by 1/sqrt(32), like torch's MHA).
//   k_query_post : per-branch self-attention over the N queries of one frame, out-proj + residual +
//                  LN, FFN (chunked over the hidden dim, never materialised) + residual + LN, then
//                  cls / mask-kernel / depth-kernel heads with feat_transform folded in.
//
// Decomposition: one workgroup (4 waves) = (ROWS query rows, branch, frame).  The [ROWS x 256] state
// lives in REGISTERS in MFMA C-fragment form (a "Tile": wave w owns columns 64w..64w+63 as four
// 16-column tiles); LayerNorm statistics are reduced with 16-lane butterflies + one tiny LDS
// exchange across the 4 waves.  Activations feeding the next GEMM are written to LDS as bf16
// (one plane, or hi/lo planes in split precision) and read back as A fragments (ds_read_b128,
// row stride 528 B = conflict free).  Weights stream from L2 as pre-packed B fragments: one
// contiguous 1 KiB block per (16-column tile, 32-deep k-step), see DESIGN.md 3.4.
// These kernels are latency/L2-stream bound, not HBM bound.
#include "ph_common.h"

constexpr int LDA = 264;   // LDS row stride (elements) of a [rows][256] bf16 activation buffer
constexpr float LN_EPS = 1e-5f;

constexpr int NW = 8;             // waves per workgroup
constexpr int CT = 2;             // 16-column tiles per wave: NW * CT * 16 = 256 columns
constexpr int WCOLS = CT * 16;
constexpr int NTHREADS = NW * 64;

template <int NRT> struct Tile { f32x4_t v[NRT][CT]; };

struct QArgs {
    const float* partial; const uint32_t* bits; const float* k_in; const float* q_in;
    const uint16_t* wb; const float* wf;
    float* obj; float* dobj; float* cls; uint16_t* kern; float* kbias;
    uint16_t* Qp; uint16_t* Kp; uint16_t* Vt; float* o1;
    ph_stage_layout lay;
    int nsplit, B, N, Npad, cls_sigmoid;
    int64_t HWp;
};

// ------------------------------------------------------------------------------------------------
template <int NRT> __device__ __forceinline__ void tile_zero(f32x4_t (&a)[NRT][CT]) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) a[rt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// acc[rt][ct] += A(LDS, [NRT*16 rows][K]) x W(col tiles ct0.., k-steps wks0..wks0+NKS-1).
// All weight fragments of the call are requested up front (NKS*NCT 16-byte loads per lane in flight):
// with one or two waves per SIMD nothing else hides the L2 latency of this stream.
template <int PA, int NRT, int NCT, int NKS>
__device__ __forceinline__ void gemm_tile(f32x4_t (&acc)[NRT][NCT], const uint16_t* A, int a_plane,
                                          const uint16_t* __restrict__ W, int64_t w_plane, int ct0, int ks_total,
                                          int wks0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    uint4 b[PA][NKS][NCT];
#pragma unroll
    for (int p = 0; p < PA; ++p)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                b[p][ks][ct] = *(const uint4*)(W + p * w_plane + ((int64_t)(ct0 + ct) * ks_total + wks0 + ks) * 512 + lane * 8);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        uint4 a[PA][NRT];
#pragma unroll
        for (int p = 0; p < PA; ++p)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
                a[p][rt] = *(const uint4*)(A + p * a_plane + (rt * 16 + i) * LDA + ks * 32 + g * 8);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                acc[rt][ct] = mfma16(a[0][rt], b[0][ks][ct], acc[rt][ct]);
                if (PA == 2) {
                    acc[rt][ct] = mfma16(a[0][rt], b[PA - 1][ks][ct], acc[rt][ct]);
                    acc[rt][ct] = mfma16(a[PA - 1][rt], b[0][ks][ct], acc[rt][ct]);
                }
            }
    }
}

// t[row][col] += bias[col]   (col = WCOLS*wave + 16*ct + (lane&15))
template <int NRT>
__device__ __forceinline__ void tile_add_bias(Tile<NRT>& t, const float* __restrict__ bias, int wave, int lane) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float bv = bias[wave * WCOLS + ct * 16 + (lane & 15)];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) t.v[rt][ct][r] += bv;
    }
}

// bf16 plane(s) of a tile -> LDS activation buffer [rows][LDA]
template <int PA, int NRT>
__device__ __forceinline__ void tile_to_lds(const Tile<NRT>& t, uint16_t* dst, int plane, int wave, int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (rt * 16 + g * 4 + r) * LDA + wave * WCOLS + ct * 16 + i;
                if (PA == 1) dst[off] = (uint16_t)f2bf(t.v[rt][ct][r]);
                else {
                    uint32_t hi, lo;
                    f2bf_split(t.v[rt][ct][r], hi, lo);
                    dst[off] = (uint16_t)hi;
                    dst[off + plane] = (uint16_t)lo;
                }
            }
}

// LayerNorm over the 256 columns of NT tiles at once (two-pass: mean, then centred variance).
// red: float [2][NT][NW waves][NRT*16]
template <int NRT, int NT>
__device__ __forceinline__ void ln_tiles(Tile<NRT> (&t)[NT], const float* const (&gam)[NT], const float* const (&bet)[NT],
                                         float* red, int wave, int lane) {
    constexpr int ROWS = NRT * 16;
    const int i = lane & 15, g = lane >> 4;
    float mean[NT][NRT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) s += t[n].v[rt][ct][r];
                s = wave_group16_sum(s);
                if (i == 0) red[((0 * NT + n) * NW + wave) * ROWS + rt * 16 + g * 4 + r] = s;
            }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* p = red + (0 * NT + n) * NW * ROWS + rt * 16 + g * 4 + r;
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += p[w * ROWS];
                mean[n][rt][r] = tot * (1.f / 256.f);
                float s = 0.f;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float d = t[n].v[rt][ct][r] - mean[n][rt][r];
                    t[n].v[rt][ct][r] = d;
                    s += d * d;
                }
                s = wave_group16_sum(s);
                if (i == 0) red[((1 * NT + n) * NW + wave) * ROWS + rt * 16 + g * 4 + r] = s;
            }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float gm[CT], bt[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            gm[ct] = gam[n][wave * WCOLS + ct * 16 + i];
            bt[ct] = bet[n][wave * WCOLS + ct * 16 + i];
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* p = red + (1 * NT + n) * NW * ROWS + rt * 16 + g * 4 + r;
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += p[w * ROWS];
                const float var = tot * (1.f / 256.f);
                const float rstd = 1.f / sqrtf(var + LN_EPS);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) t[n].v[rt][ct][r] = t[n].v[rt][ct][r] * rstd * gm[ct] + bt[ct];
            }
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ================================================================================================
//  k_query_pre
// ================================================================================================
template <int PA, int NRT>
__global__ __launch_bounds__(NTHREADS) void k_query_pre(const QArgs a) {
    constexpr int ROWS = NRT * 16;
    constexpr int PLANE = ROWS * LDA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* actA = (uint16_t*)smem;             // pooled feature u_raw   [PA][ROWS][LDA]
    uint16_t* actB = actA + PA * PLANE;           // kernel k (or q + k)
    uint16_t* actG = actB + PA * PLANE;           // g, then f, then o1
    float* red = (float*)(actG + PA * PLANE);     // [2][2][4][ROWS]
    float* cnt = red + 2 * 2 * NW * ROWS;          // [ROWS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS, br = blockIdx.y, b = blockIdx.z;
    const int N = a.N, Npad = a.Npad;
    const uint16_t* wb = a.wb;
    const int64_t wpl = a.lay.wb_plane_elems;
    const float* wf = a.wf;
    const int64_t* WO = a.lay.w[br];
    const int64_t* VO = a.lay.v[br];

    // ---- step 0: reduce pooling partials (fixed order), pixel counts, kernel rows -> LDS ----------
    {
        const int r = tid >> 4, cb = (tid & 15) * 16;   // 16 threads per row, 16 columns each
        for (int rr = r; rr < ROWS; rr += 32) {
            const int row = row0 + rr;
            float u[16], kv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { u[e] = 0.f; kv[e] = 0.f; }
            if (row < Npad) {
                for (int s = 0; s < a.nsplit; ++s) {
                    const float* p = a.partial + (((int64_t)b * a.nsplit + s) * Npad + row) * 512 + br * 256 + cb;
#pragma unroll
                    for (int e = 0; e < 16; e += 4) {
                        const float4 v = *(const float4*)(p + e);
                        u[e] += v.x; u[e + 1] += v.y; u[e + 2] += v.z; u[e + 3] += v.w;
                    }
                }
            }
            if (row < N) {
                const float* kp = a.k_in + ((int64_t)b * N + row) * 256 + cb;
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float4 v = *(const float4*)(kp + e);
                    kv[e] = v.x; kv[e + 1] = v.y; kv[e + 2] = v.z; kv[e + 3] = v.w;
                }
                if (br == 1) {   // depth_proposal + proposal_feat   (kernel_update_head.py:250)
                    const float* qp = a.q_in + ((int64_t)b * N + row) * 256 + cb;
#pragma unroll
                    for (int e = 0; e < 16; e += 4) {
                        const float4 v = *(const float4*)(qp + e);
                        kv[e] += v.x; kv[e + 1] += v.y; kv[e + 2] += v.z; kv[e + 3] += v.w;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                uint32_t h0, l0, h1, l1;
                f2bf_split(u[e], h0, l0); f2bf_split(u[e + 1], h1, l1);
                *(uint32_t*)(actA + rr * LDA + cb + e) = pack2(h0, h1);
                if (PA == 2) *(uint32_t*)(actA + PLANE + rr * LDA + cb + e) = pack2(l0, l1);
                f2bf_split(kv[e], h0, l0); f2bf_split(kv[e + 1], h1, l1);
                *(uint32_t*)(actB + rr * LDA + cb + e) = pack2(h0, h1);
                if (PA == 2) *(uint32_t*)(actB + PLANE + rr * LDA + cb + e) = pack2(l0, l1);
            }
            // pixel count of this row's mask (multiplies the folded feat_transform bias)
            int c = 0;
            if (row < Npad) {
                const uint32_t* bw = a.bits + ((int64_t)b * Npad + row) * (a.HWp / 32);
                for (int w = (tid & 15); w < a.HWp / 32; w += 16) c += __popc(bw[w]);
            }
            c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4); c += __shfl_xor(c, 8);
            if ((tid & 15) == 0) cnt[rr] = (float)c;
        }
    }
    __syncthreads();

    // ---- step 1: P = dynamic_layer(u), I = input_layer(k)   (kernel_updator.py:58-67) ------------
    Tile<NRT> Pin, Iin, PI[2];   // PI[0] = P_out, PI[1] = I_out
    tile_zero(Pin.v); tile_zero(Iin.v); tile_zero(PI[0].v); tile_zero(PI[1].v);
    gemm_tile<PA, NRT, CT, 8>(Pin.v, actA, PLANE, wb + WO[PH_W_DYN], wpl, wave * CT, 8, 0, lane);
    gemm_tile<PA, NRT, CT, 8>(PI[0].v, actA, PLANE, wb + WO[PH_W_DYN], wpl, 16 + wave * CT, 8, 0, lane);
    gemm_tile<PA, NRT, CT, 8>(Iin.v, actB, PLANE, wb + WO[PH_W_INP], wpl, wave * CT, 8, 0, lane);
    gemm_tile<PA, NRT, CT, 8>(PI[1].v, actB, PLANE, wb + WO[PH_W_INP], wpl, 16 + wave * CT, 8, 0, lane);
    {
        const float* vc = wf + VO[PH_V_DYN_CNT];
        const float* bd = wf + VO[PH_V_DYN_B];
        const float* bi = wf + VO[PH_V_INP_B];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = wave * WCOLS + ct * 16 + i;
            const float vc0 = vc[col], vc1 = vc[256 + col], bd0 = bd[col], bd1 = bd[256 + col];
            const float bi0 = bi[col], bi1 = bi[256 + col];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float cn = cnt[rt * 16 + g * 4 + r];
                    const float pin = Pin.v[rt][ct][r] + cn * vc0 + bd0;
                    PI[0].v[rt][ct][r] += cn * vc1 + bd1;
                    const float iin = Iin.v[rt][ct][r] + bi0;
                    PI[1].v[rt][ct][r] += bi1;
                    Pin.v[rt][ct][r] = iin * pin;   // gate_feats = input_in * param_in   (:69)
                }
        }
    }
    tile_to_lds<PA, NRT>(Pin, actG, PLANE, wave, lane);
    {
        const float* const gm[2] = {wf + VO[PH_V_LN_PO_G], wf + VO[PH_V_LN_IO_G]};
        const float* const bt[2] = {wf + VO[PH_V_LN_PO_B], wf + VO[PH_V_LN_IO_B]};
        ln_tiles<NRT, 2>(PI, gm, bt, red, wave, lane);   // norm_out(param_out), input_norm_out(input_out)  (:78-79)
    }
    __syncthreads();

    // ---- step 2: gates (kernel_updator.py:73-77), features (:86-87) --------------------------------
    Tile<NRT> G[2];   // G[0] = input_gate, G[1] = update_gate
    tile_zero(G[0].v); tile_zero(G[1].v);
    gemm_tile<PA, NRT, CT, 8>(G[0].v, actG, PLANE, wb + WO[PH_W_IG], wpl, wave * CT, 8, 0, lane);
    gemm_tile<PA, NRT, CT, 8>(G[1].v, actG, PLANE, wb + WO[PH_W_UG], wpl, wave * CT, 8, 0, lane);
    tile_add_bias(G[0], wf + VO[PH_V_IG_B], wave, lane);
    tile_add_bias(G[1], wf + VO[PH_V_UG_B], wave, lane);
    {
        const float* const gm[2] = {wf + VO[PH_V_LN_IG_G], wf + VO[PH_V_LN_UG_G]};
        const float* const bt[2] = {wf + VO[PH_V_LN_IG_B], wf + VO[PH_V_LN_UG_B]};
        ln_tiles<NRT, 2>(G, gm, bt, red, wave, lane);
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                G[0].v[rt][ct][r] = sigmoidf_(G[1].v[rt][ct][r]) * PI[0].v[rt][ct][r] +
                                    sigmoidf_(G[0].v[rt][ct][r]) * PI[1].v[rt][ct][r];
    // ln_tiles ended with a barrier after every wave's last read of actG in the gate GEMMs
    tile_to_lds<PA, NRT>(G[0], actG, PLANE, wave, lane);
    __syncthreads();

    // ---- step 3: fc_layer + fc_norm + ReLU (kernel_updator.py:89-91) -------------------------------
    Tile<NRT> O[1];
    tile_zero(O[0].v);
    gemm_tile<PA, NRT, CT, 8>(O[0].v, actG, PLANE, wb + WO[PH_W_FC], wpl, wave * CT, 8, 0, lane);
    tile_add_bias(O[0], wf + VO[PH_V_FC_B], wave, lane);
    {
        const float* const gm[1] = {wf + VO[PH_V_LN_FC_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_FC_B]};
        ln_tiles<NRT, 1>(O, gm, bt, red, wave, lane);
    }
    float* o1 = a.o1 + (((int64_t)b * 2 + br) * Npad + row0) * 256;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fmaxf(O[0].v[rt][ct][r], 0.f);
                O[0].v[rt][ct][r] = v;
                o1[(rt * 16 + g * 4 + r) * 256 + wave * WCOLS + ct * 16 + i] = v;   // residual for the post kernel
            }
    tile_to_lds<PA, NRT>(O[0], actG, PLANE, wave, lane);
    __syncthreads();

    // ---- step 4: attention in-projection (nn.MultiheadAttention in_proj, kernel_update_head.py:259) -
    const int64_t qk_base = (((int64_t)b * 2 + br) * Npad + row0) * 256;
    const int64_t qk_plane = (int64_t)a.B * 2 * Npad * 256;
    const int64_t vt_base = ((int64_t)b * 2 + br) * 256 * Npad + row0;
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        Tile<NRT> T;
        tile_zero(T.v);
        gemm_tile<PA, NRT, CT, 8>(T.v, actG, PLANE, wb + WO[PH_W_QKV], wpl, part * 16 + wave * CT, 8, 0, lane);
        const float* bias = wf + VO[PH_V_QKV_B] + part * 256;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = wave * WCOLS + ct * 16 + i;
            const float bv = bias[col];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = T.v[rt][ct][r] + bv;
                    if (part == 0) v *= 0.17677669529663687f;   // q * head_dim^-0.5
                    f2bf_split(v, hi[r], lo[r]);
                }
                if (part < 2) {
                    uint16_t* dst = (part == 0 ? a.Qp : a.Kp) + qk_base + (rt * 16 + g * 4) * 256 + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[r * 256] = (uint16_t)hi[r];
                        if (PA == 2) dst[qk_plane + r * 256] = (uint16_t)lo[r];
                    }
                } else {   // V transposed: [feature][query row], 4 consecutive rows per lane
                    uint16_t* dst = a.Vt + vt_base + (int64_t)col * Npad + rt * 16 + g * 4;
                    *(uint2*)dst = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
                    if (PA == 2) *(uint2*)(dst + qk_plane) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
                }
            }
        }
    }
}

// ================================================================================================
//  k_query_post
// ================================================================================================
template <int PA, int NRT>
__global__ __launch_bounds__(NTHREADS) void k_query_post(const QArgs a) {
    constexpr int ROWS = NRT * 16;
    constexpr int PLANE = ROWS * LDA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* actA = (uint16_t*)smem;                       // [PA][ROWS][LDA]
    float* red = (float*)(actA + PA * PLANE);               // [2][2][4][ROWS]
    uint16_t* region = (uint16_t*)(red + 2 * 2 * NW * ROWS); // attention P buffers, then FFN h buffers, then head buffers
    const int Npad = a.Npad, N = a.N;
    const int LDP = Npad + 8;                               // row stride of a per-wave P buffer

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS, br = blockIdx.y, b = blockIdx.z;
    const uint16_t* wb = a.wb;
    const int64_t wpl = a.lay.wb_plane_elems;
    const float* wf = a.wf;
    const int64_t* WO = a.lay.w[br];
    const int64_t* VO = a.lay.v[br];

    const int64_t qk_plane = (int64_t)a.B * 2 * Npad * 256;
    const uint16_t* Qb = a.Qp + (((int64_t)b * 2 + br) * Npad + row0) * 256;
    const uint16_t* Kb = a.Kp + ((int64_t)b * 2 + br) * Npad * 256;
    const uint16_t* Vb = a.Vt + ((int64_t)b * 2 + br) * 256 * Npad;

    // ---- attention: wave w owns head w for this block's ROWS query rows ---------------------------
    Tile<NRT> At[1];
    uint16_t* Pb = region + wave * (PA * ROWS * LDP);
    const int pplane = ROWS * LDP;
    {
        const int h = wave;
        uint4 qf[PA][NRT];
#pragma unroll
        for (int p = 0; p < PA; ++p)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
                qf[p][rt] = *(const uint4*)(Qb + p * qk_plane + (rt * 16 + i) * 256 + h * 32 + g * 8);
        float mx[NRT][4], sm[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { mx[rt][r] = -INFINITY; sm[rt][r] = 0.f; }
        // pass 1: row maxima
        for (int kt = 0; kt < Npad / 16; ++kt) {
            uint4 kf[PA];
#pragma unroll
            for (int p = 0; p < PA; ++p) kf[p] = *(const uint4*)(Kb + p * qk_plane + (kt * 16 + i) * 256 + h * 32 + g * 8);
            const bool valid = kt * 16 + i < N;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                f32x4_t s = {0.f, 0.f, 0.f, 0.f};
                s = mfma16(qf[0][rt], kf[0], s);
                if (PA == 2) { s = mfma16(qf[0][rt], kf[PA - 1], s); s = mfma16(qf[PA - 1][rt], kf[0], s); }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx[rt][r] = fmaxf(mx[rt][r], valid ? s[r] : -INFINITY);
            }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[rt][r] = wave_group16_max(mx[rt][r]);
        // pass 2: p = exp(s - max) -> LDS (bf16 planes), row sums
        for (int kt = 0; kt < Npad / 16; ++kt) {
            uint4 kf[PA];
#pragma unroll
            for (int p = 0; p < PA; ++p) kf[p] = *(const uint4*)(Kb + p * qk_plane + (kt * 16 + i) * 256 + h * 32 + g * 8);
            const bool valid = kt * 16 + i < N;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                f32x4_t s = {0.f, 0.f, 0.f, 0.f};
                s = mfma16(qf[0][rt], kf[0], s);
                if (PA == 2) { s = mfma16(qf[0][rt], kf[PA - 1], s); s = mfma16(qf[PA - 1][rt], kf[0], s); }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = valid ? expf(s[r] - mx[rt][r]) : 0.f;
                    sm[rt][r] += pv;
                    uint32_t hi, lo;
                    f2bf_split(pv, hi, lo);
                    const int off = (rt * 16 + g * 4 + r) * LDP + kt * 16 + i;
                    Pb[off] = (uint16_t)hi;
                    if (PA == 2) Pb[pplane + off] = (uint16_t)lo;
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[rt][r] = 1.f / wave_group16_sum(sm[rt][r]);
        __syncthreads();   // P visible (per-wave buffers; the barrier orders this wave's LDS writes before its reads)
        // PV: out[rows][32 d] = P[rows][keys] x V[keys][d]
        f32x4_t o[NRT][2];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) { o[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; o[rt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        for (int ks = 0; ks < Npad / 32; ++ks) {
            uint4 pf[PA][NRT];
#pragma unroll
            for (int p = 0; p < PA; ++p)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
                    pf[p][rt] = *(const uint4*)(Pb + p * pplane + (rt * 16 + i) * LDP + ks * 32 + g * 8);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                uint4 vf[PA];
#pragma unroll
                for (int p = 0; p < PA; ++p)
                    vf[p] = *(const uint4*)(Vb + p * qk_plane + (int64_t)(h * 32 + ct * 16 + i) * Npad + ks * 32 + g * 8);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    o[rt][ct] = mfma16(pf[0][rt], vf[0], o[rt][ct]);
                    if (PA == 2) {
                        o[rt][ct] = mfma16(pf[0][rt], vf[PA - 1], o[rt][ct]);
                        o[rt][ct] = mfma16(pf[PA - 1][rt], vf[0], o[rt][ct]);
                    }
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) At[0].v[rt][ct][r] = o[rt][ct][r] * sm[rt][r];
    }
    tile_to_lds<PA, NRT>(At[0], actA, PLANE, wave, lane);
    __syncthreads();

    // ---- out_proj + identity + attention_norm (kernel_update_head.py:259-260) ----------------------
    Tile<NRT> O2[1];
    tile_zero(O2[0].v);
    gemm_tile<PA, NRT, CT, 8>(O2[0].v, actA, PLANE, wb + WO[PH_W_OUT], wpl, wave * CT, 8, 0, lane);
    tile_add_bias(O2[0], wf + VO[PH_V_OUT_B], wave, lane);
    {
        const float* o1 = a.o1 + (((int64_t)b * 2 + br) * Npad + row0) * 256;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) O2[0].v[rt][ct][r] += o1[(rt * 16 + g * 4 + r) * 256 + wave * WCOLS + ct * 16 + i];
        const float* const gm[1] = {wf + VO[PH_V_LN_ATT_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_ATT_B]};
        ln_tiles<NRT, 1>(O2, gm, bt, red, wave, lane);   // includes barriers: actA readers are done
    }
    tile_to_lds<PA, NRT>(O2[0], actA, PLANE, wave, lane);
    __syncthreads();

    // ---- FFN (mmcv FFN: x + W2 relu(W1 x + b1) + b2) + ffn_norm (kernel_update_head.py:270-272) ----
    uint16_t* hbuf[2] = {region, region + PA * PLANE};
    Tile<NRT> O3[1];
    tile_zero(O3[0].v);
    const int nchunk = a.lay.ffn_dim / 256;
    for (int c = 0; c < nchunk; ++c) {
        Tile<NRT> Hc;
        tile_zero(Hc.v);
        gemm_tile<PA, NRT, CT, 8>(Hc.v, actA, PLANE, wb + WO[PH_W_FFN1], wpl, c * 16 + wave * CT, 8, 0, lane);
        const float* b1 = wf + VO[PH_V_FFN1_B] + c * 256;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float bv = b1[wave * WCOLS + ct * 16 + i];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Hc.v[rt][ct][r] = fmaxf(Hc.v[rt][ct][r] + bv, 0.f);
        }
        tile_to_lds<PA, NRT>(Hc, hbuf[c & 1], PLANE, wave, lane);
        __syncthreads();
        gemm_tile<PA, NRT, CT, 8>(O3[0].v, hbuf[c & 1], PLANE, wb + WO[PH_W_FFN2], wpl, wave * CT, a.lay.ffn_dim / 32, c * 8,
                                 lane);
    }
    tile_add_bias(O3[0], wf + VO[PH_V_FFN2_B], wave, lane);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) O3[0].v[rt][ct] += O2[0].v[rt][ct];
    {
        const float* const gm[1] = {wf + VO[PH_V_LN_FFN_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_FFN_B]};
        ln_tiles<NRT, 1>(O3, gm, bt, red, wave, lane);
    }
    {   // stage output: obj_feat / depth_feat_new  (kernel_update_head.py:349-353)
        float* out = (br == 0 ? a.obj : a.dobj) + ((int64_t)b * N + row0) * 256;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = rt * 16 + g * 4 + r;
                    if (row0 + rr < N) out[rr * 256 + wave * WCOLS + ct * 16 + i] = O3[0].v[rt][ct][r];
                }
    }
    tile_to_lds<PA, NRT>(O3[0], actA, PLANE, wave, lane);   // ln_tiles' barriers: FFN readers of actA are done
    __syncthreads();

    // ---- heads (kernel_update_head.py:274-288) ----------------------------------------------------
    uint16_t* bufM = region;                   // mask_fcs / depth_regs activation
    uint16_t* bufC = region + PA * PLANE;      // cls_fcs activation (mask branch)
    Tile<NRT> Hd[2];
    tile_zero(Hd[0].v);
    gemm_tile<PA, NRT, CT, 8>(Hd[0].v, actA, PLANE, wb + WO[PH_W_H0A], wpl, wave * CT, 8, 0, lane);
    if (br == 0) {
        tile_zero(Hd[1].v);
        gemm_tile<PA, NRT, CT, 8>(Hd[1].v, actA, PLANE, wb + WO[PH_W_H0B], wpl, wave * CT, 8, 0, lane);
        const float* const gm[2] = {wf + VO[PH_V_LN_H0A_G], wf + VO[PH_V_LN_H0B_G]};
        const float* const bt[2] = {wf + VO[PH_V_LN_H0A_B], wf + VO[PH_V_LN_H0B_B]};
        ln_tiles<NRT, 2>(Hd, gm, bt, red, wave, lane);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Hd[n].v[rt][ct][r] = fmaxf(Hd[n].v[rt][ct][r], 0.f);   // ReLU (:167,180)
        tile_to_lds<PA, NRT>(Hd[0], bufC, PLANE, wave, lane);
        tile_to_lds<PA, NRT>(Hd[1], bufM, PLANE, wave, lane);
    } else {
        Tile<NRT> D[1] = {Hd[0]};
        const float* const gm[1] = {wf + VO[PH_V_LN_H0A_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_H0A_B]};
        ln_tiles<NRT, 1>(D, gm, bt, red, wave, lane);        // depth_regs: Linear + LN, NO activation (:182-187)
        tile_to_lds<PA, NRT>(D[0], bufM, PLANE, wave, lane);
    }
    __syncthreads();

    if (br == 0) {   // fc_cls  (:285)
        const int L = a.lay.num_classes, nct = (L + 15) / 16;
        const float* bc = wf + VO[PH_V_CLS_B];
        for (int ct = wave; ct < nct; ct += NW) {
            f32x4_t acc[NRT][1];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            gemm_tile<PA, NRT, 1, 8>(acc, bufC, PLANE, wb + WO[PH_W_CLS], wpl, ct, 8, 0, lane);
            const int col = ct * 16 + i;
            if (col < L) {
                const float bv = bc[col];
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = row0 + rt * 16 + g * 4 + r;
                        if (row < N) {
                            const float z = acc[rt][0][r] + bv;
                            a.cls[((int64_t)b * N + row) * L + col] = a.cls_sigmoid ? sigmoidf_(z) : z;
                        }
                    }
            }
        }
    }
    {   // fc_mask / fc_depth folded with feat_transform / feat_depth_transform -> conv kernel + bias
        Tile<NRT> Kt;
        tile_zero(Kt.v);
        gemm_tile<PA, NRT, CT, 8>(Kt.v, bufM, PLANE, wb + WO[PH_W_KERN], wpl, wave * CT, 8, 0, lane);
        const float* bk = wf + VO[PH_V_KERN_B];
        const int64_t kplane = (int64_t)2 * a.B * Npad * 256;
        uint16_t* kd = a.kern + (((int64_t)br * a.B + b) * Npad + row0) * 256;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = wave * WCOLS + ct * 16 + i;
            const float bv = bk[col];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint32_t hi, lo;
                    f2bf_split(Kt.v[rt][ct][r] + bv, hi, lo);
                    const int off = (rt * 16 + g * 4 + r) * 256 + col;
                    kd[off] = (uint16_t)hi;
                    if (PA == 2) kd[kplane + off] = (uint16_t)lo;
                }
        }
        if (wave == 0) {   // column 256 of the folded matrix: kernel . transform bias
            f32x4_t acc[NRT][1];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            gemm_tile<PA, NRT, 1, 8>(acc, bufM, PLANE, wb + WO[PH_W_KERN], wpl, 16, 8, 0, lane);
            if (i == 0) {
                const float bv = bk[256];
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        a.kbias[((int64_t)br * a.B + b) * Npad + row0 + rt * 16 + g * 4 + r] = acc[rt][0][r] + bv;
            }
        }
    }
}

// ================================================================================================
static size_t q_ws_bytes(int B, int Npad, int PA) {
    return (size_t)PA * 3 * B * 2 * Npad * 256 * sizeof(uint16_t) + (size_t)B * 2 * Npad * 256 * sizeof(float);
}

extern "C" size_t ph_query_workspace_updator_offset(int B, int N, int prec) {
    return (size_t)(prec == PH_PREC_SPLIT ? 2 : 1) * 3 * B * 2 * ph_n_padded(N) * 256 * sizeof(uint16_t);
}

extern "C" size_t ph_query_workspace_bytes(int B, int N, int prec) {
    return q_ws_bytes(B, ph_n_padded(N), prec == PH_PREC_SPLIT ? 2 : 1);
}

template <int PA, int NRT>
static void launch_query(const QArgs& a, int phases, hipStream_t s) {
    constexpr int ROWS = NRT * 16, PLANE = ROWS * LDA;
    const size_t red = 2 * 2 * NW * ROWS * sizeof(float);
    const size_t lds_pre = (size_t)3 * PA * PLANE * 2 + red + ROWS * sizeof(float);
    size_t region = (size_t)NW * PA * ROWS * (a.Npad + 8) * 2;       // attention P buffers
    if (region < (size_t)2 * PA * PLANE * 2) region = (size_t)2 * PA * PLANE * 2;
    const size_t lds_post = (size_t)PA * PLANE * 2 + red + region;
    static bool once = false;
    if (!once) {   // allow the full 160 KiB of a CU; the per-launch size below is what is actually used
        (void)hipFuncSetAttribute((const void*)k_query_pre<PA, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_query_post<PA, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    const dim3 grid(a.Npad / ROWS, 2, a.B);
    if (phases & 1) hipLaunchKernelGGL((k_query_pre<PA, NRT>), grid, dim3(NTHREADS), lds_pre, s, a);
    if (phases & 2) hipLaunchKernelGGL((k_query_post<PA, NRT>), grid, dim3(NTHREADS), lds_post, s, a);
}

extern "C" int ph_query_stage(const float* partial, int nsplit, const uint32_t* bits, const float* k_in, const float* q_in,
                              const uint16_t* wb, const float* wf, const ph_stage_layout* layout, float* obj, float* dobj,
                              float* cls, int cls_sigmoid, uint16_t* kern, float* kbias, void* workspace,
                              size_t workspace_bytes, int B, int N, int64_t HW, int prec, int phases,
                              void* stream) {
    PH_CHECK_ARG(phases >= 1 && phases <= 3, "phases must be PH_QUERY_PRE | PH_QUERY_POST");
    PH_CHECK_ARG(partial && bits && k_in && q_in && wb && wf && layout && obj && dobj && cls && kern && kbias && workspace,
                 "null pointer");
    PH_CHECK_ARG(B > 0 && N > 0 && N <= 256 && HW > 0 && nsplit >= 1, "bad size");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT, "prec must be PH_PREC_BF16 or PH_PREC_SPLIT");
    PH_CHECK_ARG(layout->ffn_dim > 0 && layout->ffn_dim % 256 == 0, "ffn_dim must be a multiple of 256");
    PH_CHECK_ARG(layout->num_classes > 0 && layout->num_classes <= 1024, "bad num_classes");
    const int PA = prec == PH_PREC_SPLIT ? 2 : 1, Npad = ph_n_padded(N);
    if (workspace_bytes < q_ws_bytes(B, Npad, PA)) {
        ph_set_error("ph_query_stage: workspace too small (%zu < %zu)", workspace_bytes, q_ws_bytes(B, Npad, PA));
        return PH_EWORKSPACE;
    }
    QArgs a;
    a.partial = partial; a.bits = bits; a.k_in = k_in; a.q_in = q_in; a.wb = wb; a.wf = wf;
    a.obj = obj; a.dobj = dobj; a.cls = cls; a.kern = kern; a.kbias = kbias;
    const size_t pl = (size_t)PA * B * 2 * Npad * 256;
    a.Qp = (uint16_t*)workspace; a.Kp = a.Qp + pl; a.Vt = a.Kp + pl; a.o1 = (float*)(a.Vt + pl);
    a.lay = *layout; a.cls_sigmoid = cls_sigmoid; a.nsplit = nsplit; a.B = B; a.N = N; a.Npad = Npad; a.HWp = ph_hw_padded(HW);
    hipStream_t s = (hipStream_t)stream;
    if (PA == 1) launch_query<1, 2>(a, phases, s);
    else launch_query<2, 1>(a, phases, s);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
