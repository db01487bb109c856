// A8-A12 -- the query side of one KernelUpdateHead stage (kernel_update_head.py:245-288,
// funcs/kernel_updator.py:55-93) as two kernels:
//
//   k_query_pre  : deterministic reduce of the pooling partials, KernelUpdator for both branches,
//                  attention in-projection (q scaled by 1/sqrt(32), like torch's MHA).
//   k_query_post : per-branch self-attention over the N queries of one frame, out-proj + residual +
//                  LN, FFN (chunked over the hidden dim, never materialised) + residual + LN, then
//                  cls / mask-kernel / depth-kernel heads with feat_transform folded in.
//
// Decomposition: one workgroup (8 waves) = (ROWS query rows, branch, frame).  The [ROWS x 256] state
// lives in REGISTERS in MFMA C-fragment form (a "Tile": wave w owns columns 32w..32w+31 as two
// 16-column tiles); LayerNorm statistics are reduced with 16-lane DPP butterflies + one tiny LDS
// exchange across the 8 waves.  Activations feeding the next GEMM are written to LDS as bf16
// (one plane, or hi/lo planes in split precision) and read back as A fragments (ds_read_b128,
// row stride 528 B = conflict free).  Weights stream from L2 as pre-packed B fragments: one
// contiguous 1 KiB block per (16-column tile, 32-deep k-step), see DESIGN.md 3.4.
// These kernels are bound by the L2 -> CU weight stream (every workgroup reads the stage's 4 MB of
// weights for its 32 rows: ~2 us per 128 KiB call measured, ~15 TB/s aggregate), not by HBM.
#include <stdlib.h>

#include "ph_common.h"

// 16-bit stores of activations / q, k, v / dynamic kernels.  fp16 SATURATES at +-65504: the hybrid grade keeps the
// FFN hidden activations (a ReLU output, not LayerNorm-bounded) and q / k / v in ONE fp16 plane; a trained checkpoint that
// produces a value beyond fp16's range must clip there, not turn into inf and then NaN in the softmax / LayerNorm behind it
// (the bf16 hi / lo planes of the other grades cannot overflow).  One v_med3_f32 per element.
template <int E> __device__ __forceinline__ uint32_t q_f2e(float x) {
    if constexpr (E == PH_E_F16) return f2h(__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f));
    else return f2bf(x);
}

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
    int nsplit, B, N, Npad, cls_sigmoid, kern_f16;
    int64_t HWp;
    const int32_t* pcount;     // optional [B][nsplit][Npad]: set bits per row and pixel range from ph_pool_counts (else counted from `bits`)
};

// ------------------------------------------------------------------------------------------------
template <int NRT> __device__ __forceinline__ void tile_zero(f32x4_t (&a)[NRT][CT]) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) a[rt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// One GEMM call of a wave = acc[rt][ct] += A(LDS, [NRT*16 rows][NKS*32]) x W(col tiles ct0.., k-steps wks0..).
// A call's weight fragments (NKS*NCT 16-byte loads per lane, 16 KiB per wave) are requested as ONE batch.
// k_query_post (single-plane precision) requests them one call AHEAD: `w_issue` for call k+1 is placed
// before `w_use` of call k, so the L2 round trip and the 128 KiB-per-workgroup transfer through the CU's L1
// overlap the MFMAs, LayerNorms and barriers of the previous call.  With the two bf16 planes of the fp32
// mode two fragment sets do not fit the register file; there `w_issue` is a no-op and `w_use` loads on
// demand.  k_query_pre keeps up to four [ROWS x 256] fp32 tiles live next to the fragments: a second set
// in flight spills (measured: 52 us on-demand vs 60-65 us pipelined), so it uses `w_run` throughout and
// leaves the hoisting inside its barrier-free regions to the scheduler.
struct WRef { const uint16_t* W; int ct0, ks_total, wks0; };
template <int PA, int NKS, int NCT> struct WFrag { uint4 b[PA][NKS][NCT]; };

template <int PA, int NKS, int NCT>
__device__ __forceinline__ void w_load(WFrag<PA, NKS, NCT>& f, const WRef& r, int64_t w_plane, int lane) {
#pragma unroll
    for (int p = 0; p < PA; ++p)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                f.b[p][ks][ct] = *(const uint4*)(r.W + p * w_plane + ((int64_t)(r.ct0 + ct) * r.ks_total + r.wks0 + ks) * 512 + lane * 8);
}

template <int PA, int NKS, int NCT>
__device__ __forceinline__ void w_issue(WFrag<PA, NKS, NCT>& f, const WRef& r, int64_t w_plane, int lane) {
    if constexpr (PA == 1) {
        w_load(f, r, w_plane, lane);
        //SB
    }
}

// the MFMA loop over fragments that have been requested already
template <int PA, int NRT, int NCT, int NKS>
__device__ __forceinline__ void w_use_loaded(f32x4_t (&acc)[NRT][NCT], const uint16_t* A, int a_plane,
                                             const WFrag<PA, NKS, NCT>& f, int lane) {
    const int i = lane & 15, g = lane >> 4;
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
                acc[rt][ct] = mfma16(a[0][rt], f.b[0][ks][ct], acc[rt][ct]);
                if (PA == 2) {
                    acc[rt][ct] = mfma16(a[0][rt], f.b[PA - 1][ks][ct], acc[rt][ct]);
                    acc[rt][ct] = mfma16(a[PA - 1][rt], f.b[0][ks][ct], acc[rt][ct]);
                }
            }
    }
}

template <int PA, int NRT, int NCT, int NKS>
__device__ __forceinline__ void w_use(f32x4_t (&acc)[NRT][NCT], const uint16_t* A, int a_plane, WFrag<PA, NKS, NCT>& f,
                                      const WRef& r, int64_t w_plane, int lane) {
    if constexpr (PA != 1) w_load(f, r, w_plane, lane);
    w_use_loaded<PA, NRT, NCT, NKS>(acc, A, a_plane, f, lane);
}

// load-on-demand flavour (the scheduler may still hoist the loads inside a barrier-free region)
template <int PA, int NRT, int NCT, int NKS>
__device__ __forceinline__ void w_run(f32x4_t (&acc)[NRT][NCT], const uint16_t* A, int a_plane, WFrag<PA, NKS, NCT>& f,
                                      const WRef& r, int64_t w_plane, int lane) {
    w_load(f, r, w_plane, lane);
    w_use_loaded<PA, NRT, NCT, NKS>(acc, A, a_plane, f, lane);
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
template <int PA, int NRT, int E = PH_E_BF16>
__device__ __forceinline__ void tile_to_lds(const Tile<NRT>& t, uint16_t* dst, int plane, int wave, int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (rt * 16 + g * 4 + r) * LDA + wave * WCOLS + ct * 16 + i;
                if (PA == 1) dst[off] = (uint16_t)q_f2e<E>(t.v[rt][ct][r]);
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
        for (int rt = 0; rt < NRT; ++rt) {
            // bound the scheduler's hoisting of the 8-reads-per-row LDS gathers: all NT*NRT groups at once
            // cost 128 VGPRs next to a prefetched weight set
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");      // also pins the IR-level order of the LDS gathers (sched_barrier is IntrNoMem)
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
            // pin: this row tile's arithmetic is finished before the next tile's gathers are issued (volatile asm
            // statements keep their order; pure VALU work is otherwise free to sink below every later gather)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(t[n].v[rt][ct]));
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
        for (int rt = 0; rt < NRT; ++rt) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");      // also pins the IR-level order of the LDS gathers (sched_barrier is IntrNoMem)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* p = red + (1 * NT + n) * NW * ROWS + rt * 16 + g * 4 + r;
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += p[w * ROWS];
                const float var = tot * (1.f / 256.f);
                const float rstd = fast_rsqrt(var + LN_EPS);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) t[n].v[rt][ct][r] = t[n].v[rt][ct][r] * rstd * gm[ct] + bt[ct];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(t[n].v[rt][ct]));
        }
    }
}


// ================================================================================================
//  k_query_pre
// ================================================================================================
// in-projection epilogue: + bias, q scaled by head_dim^-0.5, bf16 plane(s) to the workspace
// QF16: ONE fp16 plane instead (the hybrid grade: the post kernel runs its products in fp16)
template <int PA, int NRT, int PART, bool QF16 = false>
__device__ __forceinline__ void store_qkv(const Tile<NRT>& T, const QArgs& a, const float* bias, int64_t qk_base,
                                          int64_t qk_plane, int64_t vt_base, int wave, int lane) {
    const int i = lane & 15, g = lane >> 4;
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
                if (PART == 0) v *= 0.17677669529663687f;   // q * head_dim^-0.5
                if (QF16) { hi[r] = q_f2e<PH_E_F16>(v); lo[r] = 0; }
                else if (PA == 2) f2bf_split(v, hi[r], lo[r]);
                else hi[r] = f2bf(v);
            }
            if (PART < 2) {
                uint16_t* dst = (PART == 0 ? a.Qp : a.Kp) + qk_base + (rt * 16 + g * 4) * 256 + col;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dst[r * 256] = (uint16_t)hi[r];
                    if (PA == 2 && !QF16) dst[qk_plane + r * 256] = (uint16_t)lo[r];
                }
            } else {   // V transposed: [feature][query row], 4 consecutive rows per lane
                uint16_t* dst = a.Vt + vt_base + (int64_t)col * a.Npad + rt * 16 + g * 4;
                *(uint2*)dst = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
                if (PA == 2 && !QF16) *(uint2*)(dst + qk_plane) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
            }
        }
    }
}

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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: weight / bias addresses live in SGPRs
    const int i = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS, br = blockIdx.y, b = blockIdx.z;
    const int N = a.N, Npad = a.Npad;
    const uint16_t* wb = a.wb;
    const int64_t wpl = a.lay.wb_plane_elems;
    const float* wf = a.wf;
    const int64_t* WO = a.lay.w[br];
    const int64_t* VO = a.lay.v[br];

    // the ten GEMM calls of this kernel
    const WRef c_dyn0{wb + WO[PH_W_DYN], wave * CT, 8, 0}, c_dyn1{wb + WO[PH_W_DYN], 16 + wave * CT, 8, 0};
    const WRef c_inp0{wb + WO[PH_W_INP], wave * CT, 8, 0}, c_inp1{wb + WO[PH_W_INP], 16 + wave * CT, 8, 0};
    const WRef c_ig{wb + WO[PH_W_IG], wave * CT, 8, 0}, c_ug{wb + WO[PH_W_UG], wave * CT, 8, 0};
    const WRef c_fc{wb + WO[PH_W_FC], wave * CT, 8, 0};
    const WRef c_q{wb + WO[PH_W_QKV], wave * CT, 8, 0}, c_k{wb + WO[PH_W_QKV], 16 + wave * CT, 8, 0},
        c_v{wb + WO[PH_W_QKV], 32 + wave * CT, 8, 0};
    WFrag<PA, 8, CT> fa, fb;

    // ---- step 0: reduce pooling partials (fixed order), pixel counts, kernel rows -> LDS ----------
    {
        const int r = tid >> 4, cb = (tid & 15) * 16;   // 16 threads per row, 16 columns each
        for (int rr = r; rr < ROWS; rr += NTHREADS / 16) {
            const int row = row0 + rr;
            float u[16], kv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { u[e] = 0.f; kv[e] = 0.f; }
            // pixel count of this row's mask (multiplies the folded feat_transform bias): 16 lanes x uint4,
            // eight independent loads per round trip
            int c = 0;
            if (row < Npad) {
                const uint4* bw = (const uint4*)(a.bits + ((int64_t)b * Npad + row) * (a.HWp / 32));
                const int nq = (int)(a.HWp / 128);        // HWp is a multiple of 128 pixels = 4 words
                for (int w0 = tid & 15; w0 < nq; w0 += 128) {
                    uint4 q[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) q[j] = w0 + 16 * j < nq ? bw[w0 + 16 * j] : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) c += __popc(q[j].x) + __popc(q[j].y) + __popc(q[j].z) + __popc(q[j].w);
                }
                // split-K partial sums, ascending split order (deterministic), four splits per round trip
                for (int s0 = 0; s0 < a.nsplit; s0 += 4) {
                    float4 v[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* p = a.partial + (((int64_t)b * a.nsplit + s0 + j) * Npad + row) * 512 + br * 256 + cb;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[j][e] = s0 + j < a.nsplit ? *(const float4*)(p + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            u[4 * e] += v[j][e].x; u[4 * e + 1] += v[j][e].y; u[4 * e + 2] += v[j][e].z; u[4 * e + 3] += v[j][e].w;
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
            c = (int)wave_group16_sum((float)c);          // exact: counts < 2^24
            if ((tid & 15) == 0) cnt[rr] = (float)c;
        }
    }
    __syncthreads();

    // ---- step 1: P = dynamic_layer(u), I = input_layer(k)   (kernel_updator.py:58-67) ------------
    Tile<NRT> PI[2];   // PI[0] = P_out, PI[1] = I_out
    const float* vc = wf + VO[PH_V_DYN_CNT];
    const float* bd = wf + VO[PH_V_DYN_B];
    const float* bi = wf + VO[PH_V_INP_B];
    {
        Tile<NRT> Pin, Iin;
        tile_zero(Pin.v); tile_zero(Iin.v);
        w_run<PA, NRT, CT, 8>(Pin.v, actA, PLANE, fa, c_dyn0, wpl, lane);
        w_run<PA, NRT, CT, 8>(Iin.v, actB, PLANE, fb, c_inp0, wpl, lane);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = wave * WCOLS + ct * 16 + i;
            const float vc0 = vc[col], bd0 = bd[col], bi0 = bi[col];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float cn = cnt[rt * 16 + g * 4 + r];
                    const float pin = Pin.v[rt][ct][r] + cn * vc0 + bd0;
                    const float iin = Iin.v[rt][ct][r] + bi0;
                    Pin.v[rt][ct][r] = iin * pin;   // gate_feats = input_in * param_in   (:69)
                }
        }
        tile_to_lds<PA, NRT>(Pin, actG, PLANE, wave, lane);
    }
    tile_zero(PI[0].v); tile_zero(PI[1].v);
    w_run<PA, NRT, CT, 8>(PI[0].v, actA, PLANE, fa, c_dyn1, wpl, lane);
    w_run<PA, NRT, CT, 8>(PI[1].v, actB, PLANE, fb, c_inp1, wpl, lane);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = wave * WCOLS + ct * 16 + i;
        const float vc1 = vc[256 + col], bd1 = bd[256 + col], bi1 = bi[256 + col];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                PI[0].v[rt][ct][r] += cnt[rt * 16 + g * 4 + r] * vc1 + bd1;
                PI[1].v[rt][ct][r] += bi1;
            }
    }
    {
        const float* const gm[2] = {wf + VO[PH_V_LN_PO_G], wf + VO[PH_V_LN_IO_G]};
        const float* const bt[2] = {wf + VO[PH_V_LN_PO_B], wf + VO[PH_V_LN_IO_B]};
        ln_tiles<NRT, 2>(PI, gm, bt, red, wave, lane);   // norm_out(param_out), input_norm_out(input_out)  (:78-79)
    }
    __syncthreads();

    // ---- step 2: gates (kernel_updator.py:73-77), features (:86-87) --------------------------------
    Tile<NRT> G[2];   // G[0] = input_gate, G[1] = update_gate
    tile_zero(G[0].v); tile_zero(G[1].v);
    w_run<PA, NRT, CT, 8>(G[0].v, actG, PLANE, fa, c_ig, wpl, lane);
    w_run<PA, NRT, CT, 8>(G[1].v, actG, PLANE, fb, c_ug, wpl, lane);
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
                G[0].v[rt][ct][r] = fast_sigmoid(G[1].v[rt][ct][r]) * PI[0].v[rt][ct][r] +
                                    fast_sigmoid(G[0].v[rt][ct][r]) * PI[1].v[rt][ct][r];
    // ln_tiles ended with a barrier after every wave's last read of actG in the gate GEMMs
    tile_to_lds<PA, NRT>(G[0], actG, PLANE, wave, lane);
    __syncthreads();

    // ---- step 3: fc_layer + fc_norm + ReLU (kernel_updator.py:89-91) -------------------------------
    Tile<NRT> O[1];
    tile_zero(O[0].v);
    w_run<PA, NRT, CT, 8>(O[0].v, actG, PLANE, fa, c_fc, wpl, lane);
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
    const float* qkv_bias = wf + VO[PH_V_QKV_B];
    {
        Tile<NRT> T;
        tile_zero(T.v);
        w_run<PA, NRT, CT, 8>(T.v, actG, PLANE, fb, c_q, wpl, lane);
        store_qkv<PA, NRT, 0>(T, a, qkv_bias, qk_base, qk_plane, vt_base, wave, lane);
        tile_zero(T.v);
        w_run<PA, NRT, CT, 8>(T.v, actG, PLANE, fa, c_k, wpl, lane);
        store_qkv<PA, NRT, 1>(T, a, qkv_bias + 256, qk_base, qk_plane, vt_base, wave, lane);
        tile_zero(T.v);
        w_run<PA, NRT, CT, 8>(T.v, actG, PLANE, fb, c_v, wpl, lane);
        store_qkv<PA, NRT, 2>(T, a, qkv_bias + 512, qk_base, qk_plane, vt_base, wave, lane);
    }
}

// ================================================================================================
//  k_query_post
// ================================================================================================
constexpr int MAXKT = 16;   // N <= 256 queries -> at most 16 key tiles / 8 key k-steps per head

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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: weight / bias addresses live in SGPRs
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

    WFrag<PA, 8, CT> fa, fb;
    const WRef c_out{wb + WO[PH_W_OUT], wave * CT, 8, 0};
    const int ffn_ks = a.lay.ffn_dim / 32;
    // ---- attention: wave w owns head w for this block's ROWS query rows ---------------------------
    Tile<NRT> At[1];
    uint16_t* Pb = region + wave * (PA * ROWS * LDP);
    const int pplane = ROWS * LDP;
    {
        const int h = wave;
        const int nkt = Npad / 16;
        uint4 qf[PA][NRT];
#pragma unroll
        for (int p = 0; p < PA; ++p)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
                qf[p][rt] = *(const uint4*)(Qb + p * qk_plane + (rt * 16 + i) * 256 + h * 32 + g * 8);
        // single-plane precision: the head's whole K and V^T slices (<= 16 + 16 fragments) are requested
        // up front and reused by both softmax passes; split precision loads them where they are used.
        constexpr int NPRE = PA == 1 ? MAXKT : 1;
        uint4 kpre[NPRE], vpre[NPRE];
        if constexpr (PA == 1) {
#pragma unroll
            for (int kt = 0; kt < MAXKT; ++kt)
                if (kt < nkt) kpre[kt] = *(const uint4*)(Kb + (kt * 16 + i) * 256 + h * 32 + g * 8);
#pragma unroll
            for (int ks = 0; ks < MAXKT / 2; ++ks)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    if (2 * ks < nkt) vpre[ks * 2 + ct] = *(const uint4*)(Vb + (int64_t)(h * 32 + ct * 16 + i) * Npad + ks * 32 + g * 8);
            __builtin_amdgcn_sched_barrier(0);
        }
        float mx[NRT][4], sm[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { mx[rt][r] = -INFINITY; sm[rt][r] = 0.f; }
        // pass 1: row maxima
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            if (kt < nkt) {
                uint4 kf[PA];
                if constexpr (PA == 1) kf[0] = kpre[kt];
                else {
#pragma unroll
                    for (int p = 0; p < PA; ++p) kf[p] = *(const uint4*)(Kb + p * qk_plane + (kt * 16 + i) * 256 + h * 32 + g * 8);
                }
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
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[rt][r] = wave_group16_max(mx[rt][r]);
        // pass 2: p = exp(s - max) -> LDS (bf16 planes), row sums
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            if (kt < nkt) {
                uint4 kf[PA];
                if constexpr (PA == 1) kf[0] = kpre[kt];
                else {
#pragma unroll
                    for (int p = 0; p < PA; ++p) kf[p] = *(const uint4*)(Kb + p * qk_plane + (kt * 16 + i) * 256 + h * 32 + g * 8);
                }
                const bool valid = kt * 16 + i < N;
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
                    s = mfma16(qf[0][rt], kf[0], s);
                    if (PA == 2) { s = mfma16(qf[0][rt], kf[PA - 1], s); s = mfma16(qf[PA - 1][rt], kf[0], s); }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = valid ? fast_exp(s[r] - mx[rt][r]) : 0.f;
                        sm[rt][r] += pv;
                        const int off = (rt * 16 + g * 4 + r) * LDP + kt * 16 + i;
                        if (PA == 2) {
                            uint32_t hi, lo;
                            f2bf_split(pv, hi, lo);
                            Pb[off] = (uint16_t)hi;
                            Pb[pplane + off] = (uint16_t)lo;
                        } else Pb[off] = (uint16_t)f2bf(pv);
                    }
                }
            }
        }
        w_issue(fa, c_out, wpl, lane);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[rt][r] = fast_rcp(wave_group16_sum(sm[rt][r]));
        // The P buffer is private to this wave and LDS operations of one wave complete in order: no barrier.
        // PV: out[rows][32 d] = P[rows][keys] x V[keys][d]
        f32x4_t o[NRT][2];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) { o[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; o[rt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < MAXKT / 2; ++ks) {
            if (2 * ks < nkt) {
                uint4 pf[PA][NRT];
#pragma unroll
                for (int p = 0; p < PA; ++p)
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt)
                        pf[p][rt] = *(const uint4*)(Pb + p * pplane + (rt * 16 + i) * LDP + ks * 32 + g * 8);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    uint4 vf[PA];
                    if constexpr (PA == 1) vf[0] = vpre[ks * 2 + ct];
                    else {
#pragma unroll
                        for (int p = 0; p < PA; ++p)
                            vf[p] = *(const uint4*)(Vb + p * qk_plane + (int64_t)(h * 32 + ct * 16 + i) * Npad + ks * 32 + g * 8);
                    }
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
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) At[0].v[rt][ct][r] = o[rt][ct][r] * sm[rt][r];
    }
    tile_to_lds<PA, NRT>(At[0], actA, PLANE, wave, lane);
    // residual of the attention block (written by the pre kernel): requested before the barrier
    float res[NRT][CT][4];
    {
        const float* o1 = a.o1 + (((int64_t)b * 2 + br) * Npad + row0) * 256;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) res[rt][ct][r] = o1[(rt * 16 + g * 4 + r) * 256 + wave * WCOLS + ct * 16 + i];
    }
    __syncthreads();   // also: every wave is done with its P buffer before `region` is reused below

    // ---- out_proj + identity + attention_norm (kernel_update_head.py:259-260) ----------------------
    Tile<NRT> O2[1];
    tile_zero(O2[0].v);
    w_issue(fb, WRef{wb + WO[PH_W_FFN1], wave * CT, 8, 0}, wpl, lane);
    w_use<PA, NRT, CT, 8>(O2[0].v, actA, PLANE, fa, c_out, wpl, lane);
    tile_add_bias(O2[0], wf + VO[PH_V_OUT_B], wave, lane);
    {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) O2[0].v[rt][ct][r] += res[rt][ct][r];
        const float* const gm[1] = {wf + VO[PH_V_LN_ATT_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_ATT_B]};
        ln_tiles<NRT, 1>(O2, gm, bt, red, wave, lane);   // includes barriers: actA readers are done
    }
    tile_to_lds<PA, NRT>(O2[0], actA, PLANE, wave, lane);
    __syncthreads();

    // ---- FFN (mmcv FFN: x + W2 relu(W1 x + b1) + b2) + ffn_norm (kernel_update_head.py:270-272) ----
    // software pipeline over the 2 * nchunk calls: fb = W1 chunk c (in flight on entry), fa = W2 chunk c
    uint16_t* hbuf[2] = {region, region + PA * PLANE};
    Tile<NRT> O3[1];
    tile_zero(O3[0].v);
    const int nchunk = a.lay.ffn_dim / 256;
    const WRef c_h0a{wb + WO[PH_W_H0A], wave * CT, 8, 0};
    for (int c = 0; c < nchunk; ++c) {
        const WRef w1{wb + WO[PH_W_FFN1], c * 16 + wave * CT, 8, 0};
        const WRef w2{wb + WO[PH_W_FFN2], wave * CT, ffn_ks, c * 8};
        Tile<NRT> Hc;
        tile_zero(Hc.v);
        w_issue(fa, w2, wpl, lane);
        w_use<PA, NRT, CT, 8>(Hc.v, actA, PLANE, fb, w1, wpl, lane);
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
        // next W1 chunk, or -- after the last chunk -- the first head GEMM
        const WRef nxt = c + 1 < nchunk ? WRef{wb + WO[PH_W_FFN1], (c + 1) * 16 + wave * CT, 8, 0} : c_h0a;
        w_issue(fb, nxt, wpl, lane);
        w_use<PA, NRT, CT, 8>(O3[0].v, hbuf[c & 1], PLANE, fa, w2, wpl, lane);
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
    const WRef c_kern{wb + WO[PH_W_KERN], wave * CT, 8, 0};
    Tile<NRT> Hd[2];
    tile_zero(Hd[0].v);
    if (br == 0) {
        const WRef c_h0b{wb + WO[PH_W_H0B], wave * CT, 8, 0};
        w_issue(fa, c_h0b, wpl, lane);
        w_use<PA, NRT, CT, 8>(Hd[0].v, actA, PLANE, fb, c_h0a, wpl, lane);
        tile_zero(Hd[1].v);
        w_issue(fb, c_kern, wpl, lane);
        w_use<PA, NRT, CT, 8>(Hd[1].v, actA, PLANE, fa, c_h0b, wpl, lane);
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
        w_use<PA, NRT, CT, 8>(Hd[0].v, actA, PLANE, fb, c_h0a, wpl, lane);
        w_issue(fb, c_kern, wpl, lane);
        Tile<NRT> D[1] = {Hd[0]};
        const float* const gm[1] = {wf + VO[PH_V_LN_H0A_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_H0A_B]};
        ln_tiles<NRT, 1>(D, gm, bt, red, wave, lane);        // depth_regs: Linear + LN, NO activation (:182-187)
        tile_to_lds<PA, NRT>(D[0], bufM, PLANE, wave, lane);
    }
    __syncthreads();

    WFrag<PA, 8, 1> f1;
    if (br == 0) {   // fc_cls  (:285)
        const int L = a.lay.num_classes, nct = (L + 15) / 16;
        const float* bc = wf + VO[PH_V_CLS_B];
        for (int ct = wave; ct < nct; ct += NW) {
            f32x4_t acc[NRT][1];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const WRef c_cls{wb + WO[PH_W_CLS], ct, 8, 0};
            w_load(f1, c_cls, wpl, lane);
            w_use_loaded<PA, NRT, 1, 8>(acc, bufC, PLANE, f1, lane);
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
                            a.cls[((int64_t)b * N + row) * L + col] = a.cls_sigmoid ? fast_sigmoid(z) : z;
                        }
                    }
            }
        }
    }
    {   // fc_mask / fc_depth folded with feat_transform / feat_depth_transform -> conv kernel + bias
        if (wave == 0) w_load(f1, WRef{wb + WO[PH_W_KERN], 16, 8, 0}, wpl, lane);   // column 256: kernel . transform bias
        Tile<NRT> Kt;
        tile_zero(Kt.v);
        w_use<PA, NRT, CT, 8>(Kt.v, bufM, PLANE, fb, c_kern, wpl, lane);
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
                    const int off = (rt * 16 + g * 4 + r) * 256 + col;
                    if (a.kern_f16) {            // PH_KERN_F16: one fp16 plane (the fp16 dynconv's A operand)
                        kd[off] = (uint16_t)q_f2e<PH_E_F16>(Kt.v[rt][ct][r] + bv);
                    } else {
                        uint32_t hi, lo;
                        f2bf_split(Kt.v[rt][ct][r] + bv, hi, lo);
                        kd[off] = (uint16_t)hi;
                        if (PA == 2) kd[kplane + off] = (uint16_t)lo;
                    }
                }
        }
        if (wave == 0) {
            f32x4_t acc[NRT][1];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            w_use_loaded<PA, NRT, 1, 8>(acc, bufM, PLANE, f1, lane);
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
//  Second generation: the same arithmetic with up to 80 query rows per workgroup.
//
//  The first-generation kernels above are bound by the L2 -> CU weight stream: every workgroup reads the stage's
//  4 MB (8 MB in split precision) of weights for its 32 (16) rows, and a launch takes as long as ONE workgroup's
//  stream whatever the number of workgroups.  The lever is rows per workgroup; what stood in the way
//  was register and LDS capacity.  Here
//   * the weight fragments of a GEMM call are streamed in chunks of KC k-steps, double buffered in registers
//     (64 VGPRs in split precision instead of 128 per call), the scheduler pinned chunk by chunk;
//   * k_query_pre2 keeps ONE activation buffer in LDS (u, then k, then the gate input, the gated feature, the
//     updator output) and at most three [ROWS x 32] fp32 tiles per wave: norm_out(param_out) and
//     input_norm_out(input_out) are parked in a global scratch (each lane reads back exactly what it wrote);
//   * k_query_post2 runs the attention with the key tiles in the OUTER loop (K / V^T fragments are read once per
//     pass, the exponentiated scores go through a 16 x 32 per-wave LDS tile straight into the PV MFMAs: no
//     [rows x keys] buffer), the FFN over 128 hidden units at a time, and the heads one after the other through the
//     same activation buffer.
//  LDS at 80 rows: 42 KB (84 KB split) activation buffer + 21 (43) KB hidden chunk(s).
// ================================================================================================
constexpr int LDPT = 40;   // row stride of the per-wave [16][32] attention probability tile

// Weight stream of a wave: the B fragments of a GEMM call arrive in chunks of KCH k-steps, two chunk buffers in
// registers.  Contract: chunk 0 of a call is ALREADY in flight in f[0] when the call starts ("primed" -- by the previous
// call during its last chunk, or explicitly before a LayerNorm / barrier / the attention), every call has an even number
// of chunks, so its last chunk sits in f[1] and f[0] is free for the next call's chunk 0 by then.  The L2 round trip of
// a call's first fragments therefore overlaps whatever separates two calls.
struct WRef2 { const uint16_t* W; int ct0, ks_total, wks0; };
template <int PA> constexpr int kch() { return PA == 1 ? 4 : 2; }      // 32 registers per chunk buffer (NCT = 2) either way
// (a chunk buffer is 4 x NCT fragments whatever PA is -- slot p * kch<PA>() + k -- so that ONE stream object can change its
// plane count between two calls: the hybrid PRE kernel goes from hi / lo bf16 to one fp16 plane half way)
template <int NCT> struct WStreamF { uint4 f[2][4][NCT]; };
template <int PA, int NCT> using WStream = WStreamF<NCT>;

template <int PA, int NCT>
__device__ __forceinline__ void w_chunk_load(uint4 (&d)[4][NCT], const WRef2& r, int kc, int64_t w_plane, int lane) {
#pragma unroll
    for (int p = 0; p < PA; ++p)
#pragma unroll
        for (int k = 0; k < kch<PA>(); ++k)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                d[p * kch<PA>() + k][ct] = *(const uint4*)(r.W + p * w_plane + ((int64_t)(r.ct0 + ct) * r.ks_total + r.wks0 + kc * kch<PA>() + k) * 512 + lane * 8);
}
template <int PA, int NCT>
__device__ __forceinline__ void w_prime(WStream<PA, NCT>& ws, const WRef2& r, int64_t w_plane, int lane) {
    w_chunk_load<PA, NCT>(ws.f[0], r, 0, w_plane, lane);
}
struct NoNext { __device__ __forceinline__ void operator()() const {} };

// acc[rt][ct] += A(LDS [NRT*16 rows][lda], k-steps 0..NKS-1) x W(r);  `next()` is run when f[0] has become free
template <int PA, int NRT, int NCT, int NKS, int E = PH_E_BF16, typename Next>
__device__ __forceinline__ void gemm_stream(f32x4_t (&acc)[NRT][NCT], const uint16_t* A, int lda, int a_plane, WStream<PA, NCT>& ws,
                                            const WRef2& r, int64_t w_plane, int lane, Next next) {
    constexpr int KCH = kch<PA>(), NCH = NKS / KCH;
    static_assert(NKS % KCH == 0 && NCH % 2 == 0, "a call is an even number of chunks");
    const int i = lane & 15, g = lane >> 4;
    __builtin_amdgcn_sched_barrier(0);          // phases do not interleave: the register pressure of a call is local
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
        if (kc + 1 < NCH) w_chunk_load<PA, NCT>(ws.f[(kc + 1) & 1], r, kc + 1, w_plane, lane);
        else next();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            uint4 a[PA][NRT];
#pragma unroll
            for (int p = 0; p < PA; ++p)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
                    a[p][rt] = *(const uint4*)(A + p * a_plane + (rt * 16 + i) * lda + (kc * KCH + k) * 32 + g * 8);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    acc[rt][ct] = mfma16e<E>(a[0][rt], ws.f[kc & 1][k][ct], acc[rt][ct]);
                    if (PA == 2) {
                        acc[rt][ct] = mfma16(a[0][rt], ws.f[kc & 1][(PA - 1) * KCH + k][ct], acc[rt][ct]);
                        acc[rt][ct] = mfma16(a[PA - 1][rt], ws.f[kc & 1][k][ct], acc[rt][ct]);
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// bf16 plane(s) of an [NRT*16 x 16*NCT]-per-wave tile -> LDS buffer with row stride `ld`, column offset `col0`
template <int PA, int NRT, int NCT, int E = PH_E_BF16>
__device__ __forceinline__ void frag_to_lds(const f32x4_t (&t)[NRT][NCT], uint16_t* dst, int ld, int plane, int col0, int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (rt * 16 + g * 4 + r) * ld + col0 + ct * 16 + i;
                if (PA == 1) dst[off] = (uint16_t)q_f2e<E>(t[rt][ct][r]);
                else {
                    uint32_t hi, lo;
                    f2bf_split(t[rt][ct][r], hi, lo);
                    dst[off] = (uint16_t)hi;
                    dst[off + plane] = (uint16_t)lo;
                }
            }
}

struct QArgs2 {
    QArgs q;
    unsigned long long* tl;   // debug (PH_QUERY_TIMELINE=1): s_memrealtime stamps of workgroup (0,0,0), else null
    float* pi;      // [B][2][Npad][2][256] fp32: norm_out(param_out), input_norm_out(input_out) parked by the pre kernel
};

#define PH_TL(k)                                                                                             \
    do {                                                                                                     \
        if (aa.tl && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)              \
            aa.tl[(k)] = __builtin_amdgcn_s_memrealtime();                                                   \
    } while (0)

template <int PA, int NRT, bool QF16 = false>
__global__ __launch_bounds__(NTHREADS) void k_query_pre2(const QArgs2 aa) {
    const QArgs& a = aa.q;
    constexpr int ROWS = NRT * 16;
    constexpr int PLANE = ROWS * LDA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* act = (uint16_t*)smem;              // [PA][ROWS][LDA]: u -> k -> gate input -> gated feature -> updator output
    float* red = (float*)(act + PA * PLANE);      // [2][2][NW][ROWS]
    float* cnt = red + 2 * 2 * NW * ROWS;         // [ROWS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS, br = blockIdx.y, b = blockIdx.z;
    const int N = a.N, Npad = a.Npad;
    const uint16_t* wb = a.wb;
    const int64_t wpl = a.lay.wb_plane_elems;
    const float* wf = a.wf;
    const int64_t* WO = a.lay.w[br];
    const int64_t* VO = a.lay.v[br];
    float* pi = aa.pi + ((((int64_t)b * 2 + br) * Npad + row0) * 2) * 256;

    // the ten GEMM calls of this kernel, in order; each one primes its successor's first weight chunk
    WStream<PA, CT> ws;
    const WRef2 c_dyn_o{wb + WO[PH_W_DYN], 16 + wave * CT, 8, 0}, c_dyn_i{wb + WO[PH_W_DYN], wave * CT, 8, 0};
    const WRef2 c_inp_i{wb + WO[PH_W_INP], wave * CT, 8, 0}, c_inp_o{wb + WO[PH_W_INP], 16 + wave * CT, 8, 0};
    const WRef2 c_ug{wb + WO[PH_W_UG], wave * CT, 8, 0}, c_ig{wb + WO[PH_W_IG], wave * CT, 8, 0}, c_fc{wb + WO[PH_W_FC], wave * CT, 8, 0};
    const WRef2 c_q{wb + WO[PH_W_QKV], wave * CT, 8, 0}, c_k{wb + WO[PH_W_QKV], 16 + wave * CT, 8, 0}, c_v{wb + WO[PH_W_QKV], 32 + wave * CT, 8, 0};
    auto prime = [&](const WRef2& r) { return [&, r]() { w_prime<PA, CT>(ws, r, wpl, lane); }; };
    // the tail of the kernel -- the attention in-projection, whose input is a LayerNorm output and whose results leave as fp16
    // anyway -- runs on ONE fp16 plane in the hybrid grade (QF16; pack.py packs QKV accordingly), like the POST kernel.
    // (fc_layer on one fp16 plane as well: worst per-stage error 6.3e-4 -> 8.1e-4 at cfg2 for 4 us -- its output is the
    // residual every later layer adds to; it stays hi / lo.)
    constexpr int PT = QF16 ? 1 : PA;
    constexpr int ET = QF16 ? PH_E_F16 : PH_E_BF16;
    auto primeT = [&](const WRef2& r) { return [&, r]() { w_prime<PT, CT>(ws, r, wpl, lane); }; };
    w_prime<PA, CT>(ws, c_dyn_o, wpl, lane);        // in flight under the reduction of the pooling partials
    PH_TL(0);

    // two fp32 rows of 16 columns per thread -> bf16 plane(s) in `act`
    auto put16 = [&](int rr, int cb, const float (&v)[16]) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            uint32_t h0, l0, h1, l1;
            f2bf_split(v[e], h0, l0); f2bf_split(v[e + 1], h1, l1);
            *(uint32_t*)(act + rr * LDA + cb + e) = pack2(h0, h1);
            if (PA == 2) *(uint32_t*)(act + PLANE + rr * LDA + cb + e) = pack2(l0, l1);
        }
    };

    // ---- step 0: pooled feature u = fixed-order sum of the split-K partials, pixel counts ------------------------
    // Round 4: 32 threads per row with 8 columns each (all 512 threads work on the 16 rows of a one-frame launch, where half of
    // them idled) and the partials of 8 pixel ranges in flight per trip instead of 4 (16: spills in the wide forms): the sum is a chain of dependent L2 round
    // trips -- 19 of the kernel's 56 us at one frame per launch (nsplit = 32; PH_QUERY_TIMELINE) -- and the order of the
    // additions per element (pixel range 0, 1, 2, ...) is unchanged, so every result keeps its bits.
    {
        const int r = tid >> 5, cb = (tid & 31) * 8;
        for (int rr = r; rr < ROWS; rr += NTHREADS / 32) {
            const int row = row0 + rr;
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = 0.f;
            int c = 0;
            if (row < Npad) {
                if (a.pcount) {
                    // the pooling kernel counted the set bits of its pixel range: nsplit integers per row instead of HWp / 32 words
                    for (int s0 = tid & 31; s0 < a.nsplit; s0 += 32) c += a.pcount[((int64_t)b * a.nsplit + s0) * Npad + row];
                } else {
                    const uint4* bw = (const uint4*)(a.bits + ((int64_t)b * Npad + row) * (a.HWp / 32));
                    const int nq = (int)(a.HWp / 128);
                    for (int w0 = tid & 31; w0 < nq; w0 += 256) {
                        uint4 q[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) q[j] = w0 + 32 * j < nq ? bw[w0 + 32 * j] : make_uint4(0, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 8; ++j) c += __popc(q[j].x) + __popc(q[j].y) + __popc(q[j].z) + __popc(q[j].w);
                    }
                }
                constexpr int TRIP = NRT == 1 ? 16 : 8;          // 16-row workgroups have the registers for 16 ranges per trip
                for (int s0 = 0; s0 < a.nsplit; s0 += TRIP) {
                    float4 v[TRIP][2];
#pragma unroll
                    for (int j = 0; j < TRIP; ++j) {
                        const float* p = a.partial + (((int64_t)b * a.nsplit + s0 + j) * Npad + row) * 512 + br * 256 + cb;
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            v[j][e] = s0 + j < a.nsplit ? *(const float4*)(p + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < TRIP; ++j)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            u[4 * e] += v[j][e].x; u[4 * e + 1] += v[j][e].y; u[4 * e + 2] += v[j][e].z; u[4 * e + 3] += v[j][e].w;
                        }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                uint32_t h0, l0, h1, l1;
                f2bf_split(u[e], h0, l0); f2bf_split(u[e + 1], h1, l1);
                *(uint32_t*)(act + rr * LDA + cb + e) = pack2(h0, h1);
                if (PA == 2) *(uint32_t*)(act + PLANE + rr * LDA + cb + e) = pack2(l0, l1);
            }
            float cf = wave_group16_sum((float)c);       // exact: counts < 2^24
            cf += __shfl_xor(cf, 16);
            if ((tid & 31) == 0) cnt[rr] = cf;
        }
    }
    __syncthreads();

    PH_TL(1);
    // ---- step 1: [param_in | param_out] = dynamic_layer(u)   (kernel_updator.py:58-62) ---------------------------
    const float* vc = wf + VO[PH_V_DYN_CNT];
    const float* bd = wf + VO[PH_V_DYN_B];
    {
        Tile<NRT> Po[1];
        tile_zero(Po[0].v);
        gemm_stream<PA, NRT, CT, 8>(Po[0].v, act, LDA, PLANE, ws, c_dyn_o, wpl, lane, prime(c_dyn_i));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = wave * WCOLS + ct * 16 + i;
            const float vc1 = vc[256 + col], bd1 = bd[256 + col];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Po[0].v[rt][ct][r] += cnt[rt * 16 + g * 4 + r] * vc1 + bd1;
        }
        const float* const gm[1] = {wf + VO[PH_V_LN_PO_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_PO_B]};
        ln_tiles<NRT, 1>(Po, gm, bt, red, wave, lane);          // norm_out(param_out)  (:78)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pi[((rt * 16 + g * 4 + r) * 2 + 0) * 256 + wave * WCOLS + ct * 16 + i] = Po[0].v[rt][ct][r];
    }
    Tile<NRT> Pin;                                               // param_in, then the gate input
    tile_zero(Pin.v);
    gemm_stream<PA, NRT, CT, 8>(Pin.v, act, LDA, PLANE, ws, c_dyn_i, wpl, lane, prime(c_inp_i));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = wave * WCOLS + ct * 16 + i;
        const float vc0 = vc[col], bd0 = bd[col];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pin.v[rt][ct][r] += cnt[rt * 16 + g * 4 + r] * vc0 + bd0;
    }
    __syncthreads();                                             // every wave is done reading u
    PH_TL(2);
    // ---- step 2: kernel rows (k, or depth_proposal + k for the depth branch, kernel_update_head.py:250) -> act ------
    {
        const int r = tid >> 4, cb = (tid & 15) * 16;
        for (int rr = r; rr < ROWS; rr += NTHREADS / 16) {
            const int row = row0 + rr;
            float kv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) kv[e] = 0.f;
            if (row < N) {
                const float* kp = a.k_in + ((int64_t)b * N + row) * 256 + cb;
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float4 v = *(const float4*)(kp + e);
                    kv[e] = v.x; kv[e + 1] = v.y; kv[e + 2] = v.z; kv[e + 3] = v.w;
                }
                if (br == 1) {
                    const float* qp = a.q_in + ((int64_t)b * N + row) * 256 + cb;
#pragma unroll
                    for (int e = 0; e < 16; e += 4) {
                        const float4 v = *(const float4*)(qp + e);
                        kv[e] += v.x; kv[e + 1] += v.y; kv[e + 2] += v.z; kv[e + 3] += v.w;
                    }
                }
            }
            put16(rr, cb, kv);
        }
    }
    __syncthreads();
    PH_TL(3);
    // [input_in | input_out] = input_layer(k)   (:64-67); gate input = input_in * param_in (:69)
    const float* bi = wf + VO[PH_V_INP_B];
    {
        Tile<NRT> Iin;
        tile_zero(Iin.v);
        gemm_stream<PA, NRT, CT, 8>(Iin.v, act, LDA, PLANE, ws, c_inp_i, wpl, lane, prime(c_inp_o));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float bi0 = bi[wave * WCOLS + ct * 16 + i];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Pin.v[rt][ct][r] = (Iin.v[rt][ct][r] + bi0) * Pin.v[rt][ct][r];
        }
    }
    {
        Tile<NRT> Io[1];
        tile_zero(Io[0].v);
        gemm_stream<PA, NRT, CT, 8>(Io[0].v, act, LDA, PLANE, ws, c_inp_o, wpl, lane, prime(c_ug));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float bi1 = bi[256 + wave * WCOLS + ct * 16 + i];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Io[0].v[rt][ct][r] += bi1;
        }
        const float* const gm[1] = {wf + VO[PH_V_LN_IO_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_IO_B]};
        ln_tiles<NRT, 1>(Io, gm, bt, red, wave, lane);          // input_norm_out(input_out)  (:79); barriers: k readers are done
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pi[((rt * 16 + g * 4 + r) * 2 + 1) * 256 + wave * WCOLS + ct * 16 + i] = Io[0].v[rt][ct][r];
    }
    tile_to_lds<PA, NRT>(Pin, act, PLANE, wave, lane);
    __syncthreads();

    PH_TL(4);
    // ---- step 3: gates (:73-77), gated feature (:86-87) ------------------------------------------------------------------
    {
        // one gate at a time (two live tiles): f = sigmoid(LN(update_gate)) * norm_out + sigmoid(LN(input_gate)) * input_norm_out
        Tile<NRT> Fg;
        auto gate = [&](const WRef2& cur, auto&& next, int b_idx, int g_idx, int be_idx, int which, bool first) {
            Tile<NRT> G[1];
            tile_zero(G[0].v);
            gemm_stream<PA, NRT, CT, 8>(G[0].v, act, LDA, PLANE, ws, cur, wpl, lane, next);
            tile_add_bias(G[0], wf + VO[b_idx], wave, lane);
            const float* const gm[1] = {wf + VO[g_idx]};
            const float* const bt[1] = {wf + VO[be_idx]};
            ln_tiles<NRT, 1>(G, gm, bt, red, wave, lane);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                float pv[CT][4];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[ct][r] = pi[((rt * 16 + g * 4 + r) * 2 + which) * 256 + wave * WCOLS + ct * 16 + i];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = fast_sigmoid(G[0].v[rt][ct][r]) * pv[ct][r];
                        Fg.v[rt][ct][r] = first ? t : Fg.v[rt][ct][r] + t;
                    }
            }
        };
        gate(c_ug, prime(c_ig), PH_V_UG_B, PH_V_LN_UG_G, PH_V_LN_UG_B, 0, true);       // update_gate * norm_out(param_out)
        gate(c_ig, prime(c_fc), PH_V_IG_B, PH_V_LN_IG_G, PH_V_LN_IG_B, 1, false);      // + input_gate * input_norm_out(input_out)
        // the second LayerNorm's barriers: every wave is done reading the gate input
        tile_to_lds<PA, NRT>(Fg, act, PLANE, wave, lane);
    }
    __syncthreads();

    PH_TL(5);
    // ---- step 4: fc_layer + fc_norm + ReLU (:89-91) -------------------------------------------------------------------------
    {
        Tile<NRT> O[1];
        tile_zero(O[0].v);
        gemm_stream<PA, NRT, CT, 8>(O[0].v, act, LDA, PLANE, ws, c_fc, wpl, lane, primeT(c_q));
        tile_add_bias(O[0], wf + VO[PH_V_FC_B], wave, lane);
        const float* const gm[1] = {wf + VO[PH_V_LN_FC_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_FC_B]};
        ln_tiles<NRT, 1>(O, gm, bt, red, wave, lane);           // barriers: readers of the gated feature are done
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
        tile_to_lds<PT, NRT, ET>(O[0], act, PLANE, wave, lane);
    }
    __syncthreads();

    PH_TL(6);
    // ---- step 5: attention in-projection (kernel_update_head.py:259) ---------------------------------------------------------
    const int64_t qk_base = (((int64_t)b * 2 + br) * Npad + row0) * 256;
    const int64_t qk_plane = (int64_t)a.B * 2 * Npad * 256;
    const int64_t vt_base = ((int64_t)b * 2 + br) * 256 * Npad + row0;
    const float* qkv_bias = wf + VO[PH_V_QKV_B];
    {
        Tile<NRT> T;
        tile_zero(T.v);
        gemm_stream<PT, NRT, CT, 8, ET>(T.v, act, LDA, PLANE, ws, c_q, wpl, lane, primeT(c_k));
        store_qkv<PA, NRT, 0, QF16>(T, a, qkv_bias, qk_base, qk_plane, vt_base, wave, lane);
        tile_zero(T.v);
        gemm_stream<PT, NRT, CT, 8, ET>(T.v, act, LDA, PLANE, ws, c_k, wpl, lane, primeT(c_v));
        store_qkv<PA, NRT, 1, QF16>(T, a, qkv_bias + 256, qk_base, qk_plane, vt_base, wave, lane);
        tile_zero(T.v);
        gemm_stream<PT, NRT, CT, 8, ET>(T.v, act, LDA, PLANE, ws, c_v, wpl, lane, NoNext());
        store_qkv<PA, NRT, 2, QF16>(T, a, qkv_bias + 512, qk_base, qk_plane, vt_base, wave, lane);
    }
    PH_TL(7);
}

// FFN hidden units per chunk (256 in single-plane precision: every call is then two 4-k-step weight chunks) and the
// number of hidden-chunk buffers that fit next to the activation buffer
template <int PA> constexpr int post2_hc() { return PA == 1 ? 256 : 128; }
template <int PA, int NRT> constexpr int post2_nhb() {
    return ((size_t)PA * NRT * 16 * LDA * 2 + 2 * (size_t)PA * NRT * 16 * (post2_hc<PA>() + 8) * 2 + 2 * 2 * NW * NRT * 16 * 4 + NW * PA * 16 * LDPT * 2 <= 150 * 1024) ? 2 : 1;
}

// E: element format of the single-plane form (PA = 1): bf16 (the fast mode) or fp16 (the hybrid grade: q / k / v, weights,
// LDS activations and attention probabilities as ONE fp16 plane -- every product of this kernel has a LayerNorm- or
// softmax-bounded operand, 2^-12 per operand instead of hi + lo bf16 at three MFMAs)
template <int PA, int NRT, int E = PH_E_BF16>
__global__ __launch_bounds__(NTHREADS) void k_query_post2(const QArgs2 aa) {
    const QArgs& a = aa.q;
    constexpr int ROWS = NRT * 16;
    constexpr int PLANE = ROWS * LDA;
    constexpr int HC = post2_hc<PA>(), LDH = HC + 8, CTH = HC / 128;   // hidden chunk, its row stride, 16-col tiles per wave
    constexpr int HPLANE = ROWS * LDH;
    constexpr int NHB = post2_nhb<PA, NRT>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* act = (uint16_t*)smem;                          // [PA][ROWS][LDA]
    uint16_t* hb = act + PA * PLANE;                          // [NHB][PA][ROWS][LDH]
    float* red = (float*)(hb + NHB * PA * HPLANE);            // [2][2][NW][ROWS]
    uint16_t* ptile = (uint16_t*)(red + 2 * 2 * NW * ROWS);   // [NW][PA][16][LDPT]
    const int Npad = a.Npad, N = a.N;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    WStream<PA, CT> ws;               // 32-column calls
    WStream<PA, 1> ws1;               // 16-column calls (fc_cls, the bias column; the FFN's first layer in split precision)
    const WRef2 c_out{wb + WO[PH_W_OUT], wave * CT, 8, 0};
    const WRef2 c_h0a{wb + WO[PH_W_H0A], wave * CT, 8, 0}, c_h0b{wb + WO[PH_W_H0B], wave * CT, 8, 0};
    const WRef2 c_kern{wb + WO[PH_W_KERN], wave * CT, 8, 0}, c_kb{wb + WO[PH_W_KERN], 16, 8, 0};
    auto prime = [&](const WRef2& r) { return [&, r]() { w_prime<PA, CT>(ws, r, wpl, lane); }; };
    auto ffn1_ref = [&](int c) { return WRef2{wb + WO[PH_W_FFN1], c * (HC / 16) + wave * CTH, 8, 0}; };
    auto ffn1_prime = [&](int c) {
        if constexpr (CTH == CT) w_prime<PA, CT>(ws, ffn1_ref(c), wpl, lane);
        else w_prime<PA, 1>(ws1, ffn1_ref(c), wpl, lane);
    };
    w_prime<PA, CT>(ws, c_out, wpl, lane);           // in flight under the attention
    PH_TL(8);

    // ---- attention: wave w = head w for this block's rows; key tiles in the outer loop -------------------------------
    Tile<NRT> At;
    {
        const int h = wave;
        const int nkt = Npad / 16;
        uint16_t* Pt = ptile + wave * (PA * 16 * LDPT);
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
        auto load_k = [&](uint4 (&kf)[PA], int kt) {
#pragma unroll
            for (int p = 0; p < PA; ++p) kf[p] = *(const uint4*)(Kb + p * qk_plane + (kt * 16 + i) * 256 + h * 32 + g * 8);
        };
        auto score = [&](const uint4 (&kf)[PA], int rt) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
            s = mfma16e<E>(qf[0][rt], kf[0], s);
            if (PA == 2) { s = mfma16(qf[0][rt], kf[PA - 1], s); s = mfma16(qf[PA - 1][rt], kf[0], s); }
            return s;
        };
        // pass 1: row maxima
        for (int kt = 0; kt < nkt; ++kt) {
            uint4 kf[PA];
            load_k(kf, kt);
            const bool valid = kt * 16 + i < N;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const f32x4_t s = score(kf, rt);
#pragma unroll
                for (int r = 0; r < 4; ++r) mx[rt][r] = fmaxf(mx[rt][r], valid ? s[r] : -INFINITY);
            }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[rt][r] = wave_group16_max(mx[rt][r]);
        // pass 2: p = exp(s - max) for 32 keys at a time -> [16][32] LDS tile (bf16 planes) -> PV
        tile_zero(At.v);
        for (int ks = 0; ks < nkt / 2; ++ks) {
            uint4 kf[2][PA], vf[2][PA];
            load_k(kf[0], 2 * ks);
            load_k(kf[1], 2 * ks + 1);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int p = 0; p < PA; ++p)
                    vf[ct][p] = *(const uint4*)(Vb + p * qk_plane + (int64_t)(h * 32 + ct * 16 + i) * Npad + ks * 32 + g * 8);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4_t s = score(kf[half], rt);
                    const bool valid = (2 * ks + half) * 16 + i < N;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = valid ? fast_exp(s[r] - mx[rt][r]) : 0.f;
                        sm[rt][r] += pv;
                        const int off = (g * 4 + r) * LDPT + half * 16 + i;
                        if (PA == 2) {
                            uint32_t hi, lo;
                            f2bf_split(pv, hi, lo);
                            Pt[off] = (uint16_t)hi;
                            Pt[16 * LDPT + off] = (uint16_t)lo;
                        } else Pt[off] = (uint16_t)q_f2e<E>(pv);
                    }
                }
                // the tile is private to this wave and LDS operations of one wave complete in order: no barrier
                uint4 pf[PA];
#pragma unroll
                for (int p = 0; p < PA; ++p) pf[p] = *(const uint4*)(Pt + p * 16 * LDPT + i * LDPT + g * 8);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    At.v[rt][ct] = mfma16e<E>(pf[0], vf[ct][0], At.v[rt][ct]);
                    if (PA == 2) {
                        At.v[rt][ct] = mfma16(pf[0], vf[ct][PA - 1], At.v[rt][ct]);
                        At.v[rt][ct] = mfma16(pf[PA - 1], vf[ct][0], At.v[rt][ct]);
                    }
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float inv = fast_rcp(wave_group16_sum(sm[rt][r]));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) At.v[rt][ct][r] *= inv;
            }
    }
    tile_to_lds<PA, NRT, E>(At, act, PLANE, wave, lane);
    __syncthreads();

    PH_TL(9);
    // ---- out_proj + identity + attention_norm (kernel_update_head.py:259-260) ----------------------------------------------
    Tile<NRT> O2[1];
    tile_zero(O2[0].v);
    gemm_stream<PA, NRT, CT, 8, E>(O2[0].v, act, LDA, PLANE, ws, c_out, wpl, lane, [&]() { ffn1_prime(0); });
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
        ln_tiles<NRT, 1>(O2, gm, bt, red, wave, lane);          // barriers: readers of the attention output are done
    }
    tile_to_lds<PA, NRT, E>(O2[0], act, PLANE, wave, lane);
    // the FFN's residual waits in the scratch (each lane reads back what it wrote): 40 registers less across the FFN loop
    float* park = aa.pi + ((((int64_t)b * 2 + br) * Npad + row0) * 2) * 256;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) park[(rt * 16 + g * 4 + r) * 512 + wave * WCOLS + ct * 16 + i] = O2[0].v[rt][ct][r];
    __syncthreads();

    PH_TL(10);
    // ---- FFN (x + W2 relu(W1 x + b1) + b2) + ffn_norm (:270-272), 128 hidden units at a time ------------------------------------
    Tile<NRT> O3[1];
    tile_zero(O3[0].v);
    {
        const int nchunk = a.lay.ffn_dim / HC;
        const int ffn_ks = a.lay.ffn_dim / 32;
        for (int c = 0; c < nchunk; ++c) {
            uint16_t* hc = hb + (NHB == 2 ? (c & 1) : 0) * PA * HPLANE;
            const WRef2 w2{wb + WO[PH_W_FFN2], wave * CT, ffn_ks, c * (HC / 32)};
            f32x4_t Hc[NRT][CTH];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CTH; ++ct) Hc[rt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (CTH == CT) gemm_stream<PA, NRT, CT, 8, E>(Hc, act, LDA, PLANE, ws, ffn1_ref(c), wpl, lane, prime(w2));
            else gemm_stream<PA, NRT, 1, 8, E>(Hc, act, LDA, PLANE, ws1, ffn1_ref(c), wpl, lane, prime(w2));
#pragma unroll
            for (int ct = 0; ct < CTH; ++ct) {
                const float bv = (wf + VO[PH_V_FFN1_B])[c * HC + (wave * CTH + ct) * 16 + i];
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Hc[rt][ct][r] = fmaxf(Hc[rt][ct][r] + bv, 0.f);
            }
            if (NHB == 1 && c > 0) __syncthreads();             // every wave is done reading the previous chunk
            frag_to_lds<PA, NRT, CTH, E>(Hc, hc, LDH, HPLANE, wave * CTH * 16, lane);
            __syncthreads();
            // while the last weight chunk of this call is consumed: the next hidden chunk's first layer, or the first head
            gemm_stream<PA, NRT, CT, HC / 32, E>(O3[0].v, hc, LDH, HPLANE, ws, w2, wpl, lane, [&]() {
                if (c + 1 < nchunk) ffn1_prime(c + 1);
                else w_prime<PA, CT>(ws, br == 0 ? c_h0b : c_h0a, wpl, lane);
            });
        }
    }
    PH_TL(11);
    tile_add_bias(O3[0], wf + VO[PH_V_FFN2_B], wave, lane);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) O3[0].v[rt][ct][r] += park[(rt * 16 + g * 4 + r) * 512 + wave * WCOLS + ct * 16 + i];
    {
        const float* const gm[1] = {wf + VO[PH_V_LN_FFN_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_FFN_B]};
        ln_tiles<NRT, 1>(O3, gm, bt, red, wave, lane);          // barriers: FFN readers of `act` are done
    }
    {   // stage output: obj_feat / depth_feat_new  (:349-353)
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
    tile_to_lds<PA, NRT, E>(O3[0], act, PLANE, wave, lane);
    __syncthreads();

    PH_TL(12);
    // ---- heads (:274-288), one after the other through `act` ---------------------------------------------------------------------
    Tile<NRT> Hm[1];                                              // mask_fcs / depth_regs activation, kept in registers
    if (br == 0) {
        // mask_fcs first (its activation waits in registers), then cls_fcs; each Linear -> LN -> ReLU (:161-180)
        tile_zero(Hm[0].v);
        gemm_stream<PA, NRT, CT, 8, E>(Hm[0].v, act, LDA, PLANE, ws, c_h0b, wpl, lane, prime(c_h0a));
        {
            const float* const gm[1] = {wf + VO[PH_V_LN_H0B_G]};
            const float* const bt[1] = {wf + VO[PH_V_LN_H0B_B]};
            ln_tiles<NRT, 1>(Hm, gm, bt, red, wave, lane);
        }
        Tile<NRT> Hd[1];
        tile_zero(Hd[0].v);
        const int L = a.lay.num_classes, nct = (L + 15) / 16;
        auto cls_ref = [&](int ct) { return WRef2{wb + WO[PH_W_CLS], ct, 8, 0}; };
        gemm_stream<PA, NRT, CT, 8, E>(Hd[0].v, act, LDA, PLANE, ws, c_h0a, wpl, lane, [&]() {
            w_prime<PA, CT>(ws, c_kern, wpl, lane);                                   // waits through fc_cls
            if (wave < nct) w_prime<PA, 1>(ws1, cls_ref(wave), wpl, lane);
        });
        {
            const float* const gm[1] = {wf + VO[PH_V_LN_H0A_G]};
            const float* const bt[1] = {wf + VO[PH_V_LN_H0A_B]};
            ln_tiles<NRT, 1>(Hd, gm, bt, red, wave, lane);      // barriers: readers of the stage output are done
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Hd[0].v[rt][ct][r] = fmaxf(Hd[0].v[rt][ct][r], 0.f);
                    Hm[0].v[rt][ct][r] = fmaxf(Hm[0].v[rt][ct][r], 0.f);
                }
        tile_to_lds<PA, NRT, E>(Hd[0], act, PLANE, wave, lane);
        __syncthreads();
        // fc_cls (:285)
        const float* bc = wf + VO[PH_V_CLS_B];
        for (int ct = wave; ct < nct; ct += NW) {
            f32x4_t acc[NRT][1];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            gemm_stream<PA, NRT, 1, 8, E>(acc, act, LDA, PLANE, ws1, cls_ref(ct), wpl, lane, [&]() {
                if (ct + NW < nct) w_prime<PA, 1>(ws1, cls_ref(ct + NW), wpl, lane);
            });
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
                            a.cls[((int64_t)b * N + row) * L + col] = a.cls_sigmoid ? fast_sigmoid(z) : z;
                        }
                    }
            }
        }
        __syncthreads();                                          // fc_cls readers are done
    } else {
        tile_zero(Hm[0].v);
        gemm_stream<PA, NRT, CT, 8, E>(Hm[0].v, act, LDA, PLANE, ws, c_h0a, wpl, lane, prime(c_kern));
        const float* const gm[1] = {wf + VO[PH_V_LN_H0A_G]};
        const float* const bt[1] = {wf + VO[PH_V_LN_H0A_B]};
        ln_tiles<NRT, 1>(Hm, gm, bt, red, wave, lane);          // depth_regs: Linear + LN, NO activation (:182-187)
    }
    tile_to_lds<PA, NRT, E>(Hm[0], act, PLANE, wave, lane);
    __syncthreads();
    {   // fc_mask / fc_depth folded with feat_transform / feat_depth_transform -> conv kernel + bias
        Tile<NRT> Kt;
        tile_zero(Kt.v);
        gemm_stream<PA, NRT, CT, 8, E>(Kt.v, act, LDA, PLANE, ws, c_kern, wpl, lane, [&]() {
            if (wave == 0) w_prime<PA, 1>(ws1, c_kb, wpl, lane);                      // column 256: kernel . transform bias
        });
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
                    const int off = (rt * 16 + g * 4 + r) * 256 + col;
                    if (a.kern_f16) {
                        kd[off] = (uint16_t)q_f2e<PH_E_F16>(Kt.v[rt][ct][r] + bv);
                    } else {
                        uint32_t hi, lo;
                        f2bf_split(Kt.v[rt][ct][r] + bv, hi, lo);
                        kd[off] = (uint16_t)hi;
                        if (PA == 2) kd[kplane + off] = (uint16_t)lo;
                    }
                }
        }
        if (wave == 0) {                                          // column 256: kernel . transform bias
            f32x4_t acc[NRT][1];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            gemm_stream<PA, NRT, 1, 8, E>(acc, act, LDA, PLANE, ws1, c_kb, wpl, lane, NoNext());
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
    PH_TL(13);
}

// ================================================================================================
static size_t q_ws_bytes(int B, int Npad, int PA) {
    // q / k / v^T planes | updator output (residual) | parked norm_out(param_out), input_norm_out(input_out)
    return (size_t)PA * 3 * B * 2 * Npad * 256 * sizeof(uint16_t) + (size_t)3 * B * 2 * Npad * 256 * sizeof(float);
}

extern "C" size_t ph_query_workspace_updator_offset(int B, int N, int prec) {
    return (size_t)((prec == PH_PREC_SPLIT || prec == PH_PREC_QHYBRID) ? 2 : 1) * 3 * B * 2 * ph_n_padded(N) * 256 * sizeof(uint16_t);
}

extern "C" size_t ph_query_workspace_bytes(int B, int N, int prec) {
    return q_ws_bytes(B, ph_n_padded(N), (prec == PH_PREC_SPLIT || prec == PH_PREC_QHYBRID) ? 2 : 1);
}

template <int PA, int NRT>
static void launch_query(const QArgs& a, int phases, hipStream_t s) {
    constexpr int ROWS = NRT * 16, PLANE = ROWS * LDA;
    const size_t red = 2 * 2 * NW * ROWS * sizeof(float);
    const size_t lds_pre = (size_t)3 * PA * PLANE * 2 + red + ROWS * sizeof(float);
    size_t region = (size_t)NW * PA * ROWS * (a.Npad + 8) * 2;       // attention P buffers
    if (region < (size_t)2 * PA * PLANE * 2) region = (size_t)2 * PA * PLANE * 2;
    const size_t lds_post = (size_t)PA * PLANE * 2 + red + region;
    static const bool once = [&] {   // allow the full 160 KiB of a CU; the per-launch size below is what is actually used
        (void)hipFuncSetAttribute((const void*)k_query_pre<PA, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_query_post<PA, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    const dim3 grid(a.Npad / ROWS, 2, a.B);
    if (phases & 1) hipLaunchKernelGGL((k_query_pre<PA, NRT>), grid, dim3(NTHREADS), lds_pre, s, a);
    if (phases & 2) hipLaunchKernelGGL((k_query_post<PA, NRT>), grid, dim3(NTHREADS), lds_post, s, a);
}

// HY: the hybrid grade -- pre kernel in split precision with fp16 q / k / v out, post kernel with ONE fp16 plane
template <int PA, int NRT, bool HY = false>
static void launch_query2(const QArgs2& a, int phases, hipStream_t s) {
    constexpr int ROWS = NRT * 16;
    if constexpr (HY) {
        static_assert(PA == 2, "hybrid = split pre kernel");
        const size_t red = 2 * 2 * NW * ROWS * sizeof(float);
        const size_t lds_pre = (size_t)2 * ROWS * LDA * 2 + red + ROWS * sizeof(float);
        const size_t lds_post = (size_t)ROWS * LDA * 2 + (size_t)post2_nhb<1, NRT>() * ROWS * (post2_hc<1>() + 8) * 2 + red + (size_t)NW * 16 * LDPT * 2;
        static const bool once = [] {
            (void)hipFuncSetAttribute((const void*)k_query_pre2<2, NRT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_query_post2<1, NRT, PH_E_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return true;
        }();
        (void)once;
        const dim3 grid(a.q.Npad / ROWS, 2, a.q.B);
        if (phases & 1) hipLaunchKernelGGL((k_query_pre2<2, NRT, true>), grid, dim3(NTHREADS), lds_pre, s, a);
        if (phases & 2) hipLaunchKernelGGL((k_query_post2<1, NRT, PH_E_F16>), grid, dim3(NTHREADS), lds_post, s, a);
        return;
    }
    const size_t red = 2 * 2 * NW * ROWS * sizeof(float);
    const size_t lds_pre = (size_t)PA * ROWS * LDA * 2 + red + ROWS * sizeof(float);
    const size_t lds_post = (size_t)PA * ROWS * LDA * 2 + (size_t)post2_nhb<PA, NRT>() * PA * ROWS * (post2_hc<PA>() + 8) * 2 + red +
                            (size_t)NW * PA * 16 * LDPT * 2;
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)k_query_pre2<PA, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_query_post2<PA, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    const dim3 grid(a.q.Npad / ROWS, 2, a.q.B);
    if (phases & 1) hipLaunchKernelGGL((k_query_pre2<PA, NRT>), grid, dim3(NTHREADS), lds_pre, s, a);
    if (phases & 2) hipLaunchKernelGGL((k_query_post2<PA, NRT>), grid, dim3(NTHREADS), lds_post, s, a);
}

// (error messages of both entry points name this function)
static int query_run(const float* partial, int nsplit, const uint32_t* bits, const int32_t* pcount, const float* k_in, const float* q_in,
                     const uint16_t* wb, const float* wf, const ph_stage_layout* layout, float* obj, float* dobj,
                     float* cls, int cls_sigmoid, uint16_t* kern, float* kbias, void* workspace,
                     size_t workspace_bytes, int B, int N, int64_t HW, int prec, int kern_format, int phases,
                     void* stream) {
    PH_CHECK_ARG((phases & ~(PH_QUERY_BOTH | PH_QUERY_WIDE)) == 0 && (phases & PH_QUERY_BOTH) != 0,
                 "phases must be PH_QUERY_PRE | PH_QUERY_POST [| PH_QUERY_WIDE]");
    PH_CHECK_ARG(partial && bits && k_in && q_in && wb && wf && layout && obj && dobj && cls && kern && kbias && workspace,
                 "null pointer");
    PH_CHECK_ARG(B > 0 && N > 0 && N <= 256 && HW > 0 && nsplit >= 1, "bad size");
    PH_CHECK_ARG(prec == PH_PREC_BF16 || prec == PH_PREC_SPLIT || prec == PH_PREC_QHYBRID, "prec must be PH_PREC_BF16, PH_PREC_SPLIT or PH_PREC_QHYBRID");
    PH_CHECK_ARG(kern_format == PH_KERN_BF16_PLANES || kern_format == PH_KERN_F16, "bad kern_format");
    PH_CHECK_ARG(prec != PH_PREC_QHYBRID || kern_format == PH_KERN_F16, "PH_PREC_QHYBRID emits the dynamic kernels as one fp16 plane (PH_KERN_F16)");
    PH_CHECK_ARG(layout->ffn_dim > 0 && layout->ffn_dim % 256 == 0, "ffn_dim must be a multiple of 256");
    PH_CHECK_ARG(layout->num_classes > 0 && layout->num_classes <= 1024, "bad num_classes");
    const bool hybrid = prec == PH_PREC_QHYBRID;
    const int PA = (prec == PH_PREC_SPLIT || hybrid) ? 2 : 1, Npad = ph_n_padded(N);
    const bool wide = (phases & PH_QUERY_WIDE) != 0;
    phases &= PH_QUERY_BOTH;
    if (workspace_bytes < q_ws_bytes(B, Npad, PA)) {
        ph_set_error("ph_query_stage: workspace too small (%zu < %zu)", workspace_bytes, q_ws_bytes(B, Npad, PA));
        return PH_EWORKSPACE;
    }
    QArgs a;
    a.partial = partial; a.bits = bits; a.pcount = pcount; a.k_in = k_in; a.q_in = q_in; a.wb = wb; a.wf = wf;
    a.obj = obj; a.dobj = dobj; a.cls = cls; a.kern = kern; a.kbias = kbias;
    const size_t pl = (size_t)PA * B * 2 * Npad * 256;
    a.Qp = (uint16_t*)workspace; a.Kp = a.Qp + pl; a.Vt = a.Kp + pl; a.o1 = (float*)(a.Vt + pl);
    a.lay = *layout; a.cls_sigmoid = cls_sigmoid; a.kern_f16 = kern_format == PH_KERN_F16; a.nsplit = nsplit; a.B = B; a.N = N; a.Npad = Npad; a.HWp = ph_hw_padded(HW);
    hipStream_t s = (hipStream_t)stream;
    static const bool v1 = [] { const char* e = getenv("PH_QUERY_V1"); return e && atoi(e) != 0; }();
    if (v1 && !hybrid) {                  // first-generation kernels (32 / 16 rows per workgroup), kept for A/B measurements
        if (PA == 1) launch_query<1, 2>(a, phases, s);
        else launch_query<2, 1>(a, phases, s);
    } else {
        QArgs2 a2;
        a2.q = a;
        static unsigned long long* tl = [] {            // debug only: PH_QUERY_TIMELINE=1 prints phase times of one workgroup
            unsigned long long* p = nullptr;
            const char* e = getenv("PH_QUERY_TIMELINE");
            if (e && atoi(e) != 0 && hipMalloc((void**)&p, 16 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemset(p, 0, 128);
            return p;
        }();
        a2.tl = tl;
        a2.pi = a.o1 + (size_t)B * 2 * Npad * 256;
        // rows per workgroup (16 * nrt, nrt | Npad / 16).  Two regimes, measured at cfg2 (tools/query_time.py, split
        // precision, pre + post): 24 frames alone on the GPU 306 us at 80 rows (96 workgroups) against 166 us at 32 rows (240
        // workgroups) -- a launch that has the chip to itself wants the chip FILLED; inside the multi-stream step the 80-row
        // form costs 34 CU-ms per launch against 48 and leaves 160 CUs to the other parts' HBM kernels (8.16 against 8.58 ms per
        // 96-frame step) -- PH_QUERY_WIDE asks for that one.  Default: the largest divisor that still gives >= 200 workgroups,
        // else the smallest one above 16 rows.
        const int t = Npad / 16;
        int nrt = 1;
        if (wide) {
            nrt = t % 5 == 0 ? 5 : (t % 4 == 0 ? 4 : (t % 3 == 0 ? 3 : (t % 2 == 0 ? 2 : 1)));
        } else {
            int smallest = 1;
            for (int c = 5; c >= 2; --c) {
                if (t % c) continue;
                smallest = c;
                if (nrt == 1 && (int64_t)B * 2 * (t / c) >= 200) nrt = c;
            }
            if (nrt == 1) nrt = smallest;
        }
        static const int cap = [] { const char* e = getenv("PH_QUERY_NRT"); return e ? atoi(e) : 0; }();   // tuning knob
        if (cap > 0) { nrt = cap; while (t % nrt) --nrt; }
#define PH_Q2(P, R) launch_query2<P, R>(a2, phases, s)
#define PH_QH(R) launch_query2<2, R, true>(a2, phases, s)
        if (hybrid) { switch (nrt) { case 5: PH_QH(5); break; case 4: PH_QH(4); break; case 3: PH_QH(3); break; case 2: PH_QH(2); break; default: PH_QH(1); } }
        else if (PA == 1) { switch (nrt) { case 5: PH_Q2(1, 5); break; case 4: PH_Q2(1, 4); break; case 3: PH_Q2(1, 3); break; case 2: PH_Q2(1, 2); break; default: PH_Q2(1, 1); } }
        else { switch (nrt) { case 5: PH_Q2(2, 5); break; case 4: PH_Q2(2, 4); break; case 3: PH_Q2(2, 3); break; case 2: PH_Q2(2, 2); break; default: PH_Q2(2, 1); } }
#undef PH_Q2
#undef PH_QH
        if (tl) {
            unsigned long long h[16];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, tl, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "query timeline (100 MHz ticks, x10 ns):");
            for (int k = 1; k < 14; ++k) fprintf(stderr, " %s%lld", k == 8 ? "| " : "", (long long)(h[k] - h[k - 1]));
            fprintf(stderr, "\n");
        }
    }
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_query_stage(const float* partial, int nsplit, const uint32_t* bits, const float* k_in, const float* q_in,
                              const uint16_t* wb, const float* wf, const ph_stage_layout* layout, float* obj, float* dobj,
                              float* cls, int cls_sigmoid, uint16_t* kern, float* kbias, void* workspace,
                              size_t workspace_bytes, int B, int N, int64_t HW, int prec, int kern_format, int phases,
                              void* stream) {
    return query_run(partial, nsplit, bits, nullptr, k_in, q_in, wb, wf, layout, obj, dobj, cls, cls_sigmoid, kern, kbias, workspace,
                     workspace_bytes, B, N, HW, prec, kern_format, phases, stream);
}

// the same with the hard masks' pixel counts handed over by ph_pool_counts (pcount [B][nsplit][Npad] int32) instead of counted
// from the bit rows (second-generation kernels; PH_QUERY_V1 ignores it)
extern "C" int ph_query_stage_counts(const float* partial, int nsplit, const uint32_t* bits, const int32_t* pcount, const float* k_in,
                                     const float* q_in, const uint16_t* wb, const float* wf, const ph_stage_layout* layout,
                                     float* obj, float* dobj, float* cls, int cls_sigmoid, uint16_t* kern, float* kbias,
                                     void* workspace, size_t workspace_bytes, int B, int N, int64_t HW, int prec, int kern_format,
                                     int phases, void* stream) {
    if (!pcount) { ph_set_error("ph_query_stage_counts: null pcount"); return PH_EINVAL; }
    return query_run(partial, nsplit, bits, pcount, k_in, q_in, wb, wf, layout, obj, dobj, cls, cls_sigmoid, kern, kbias, workspace,
                     workspace_bytes, B, N, HW, prec, kern_format, phases, stream);
}
