// N4 (first part) -- the matching costs of the mask Hungarian assigner (polyphonic/funcs/assigner.py:113-129 DiceCost,
// :164-194 MaskCost, both with pred_act = sigmoid and the optional gt_valid pixel mask, as
// MaskHungarianAssignerWithDepth.assign :478-498 calls them).  All pixel sums of one image in ONE pass over the mask
// logits and the (soft, bilinear-downsampled) ground-truth masks:
//     A[n][g] = sum_h p[n,h] t[g,h] v[h]          p = sigmoid(logit)
//     S[n]    = sum_h p[n,h] v[h]                  (column G of A: an extra "ground truth" row equal to v)
//     Q[n]    = sum_h p[n,h]^2 v[h]                C[g] = sum_h t[g,h]^2 v[h]      T[g] = sum_h t[g,h] v[h]
//     V       = sum_h v[h]
// from which the host forms   dice = -2A / (Q + eps + C + eps)      mask = -(A + (V - S - T + A)) / V.
// A is a [N x HW] x [HW x (G+1)] contraction over pixels -- both operands are K(=pixel)-contiguous rows, i.e. exactly
// the MFMA 32x32x16 A / B fragment order, so the tiles go fp32 -> (sigmoid) -> bf16 hi/lo planes -> LDS -> ds_read_b128
// with no transposition.  hi*hi + hi*lo + lo*hi keeps 16 mantissa bits of either operand (costs within ~1e-5 of the
// fp32 einsum; the assignment is decided by them).  HBM: every input byte once; the problem is small (tens of MB) and
// split over the pixel axis so that it fills the chip; partial sums are written per split and added in a fixed order.
#include "ph_common.h"

constexpr int MT_K = 64;              // pixels per chunk
constexpr int MT_LD = MT_K + 8;       // LDS row stride (elements)
constexpr int MT_THREADS = 256;

struct MatchArgs {
    const float* logits;   // [B][N][HW]
    const float* gt;       // [B][G][HW]
    const float* valid;    // [B][HW] or null (all valid)
    float* part;           // [B][nsplit][stride]: A [Npad][Gpad], then S-free layout below
    int N, G, Npad, Gpad, nsplit, chunks_per_split;
    int64_t HW, stride;
};
// layout of one partial record (floats): A[Npad][Gpad] | Q[Npad] | C[Gpad] | T[Gpad] | V[1]

template <int RTW /* row tiles per wave */, int NCT /* 32-column tiles */>
__global__ __launch_bounds__(MT_THREADS) void k_match_sums(const MatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, split = blockIdx.x;
    const int r16 = tid >> 4, c4 = (tid & 15) * 4;
    uint16_t* Ah = lds;                                   // [2 planes][Npad][MT_LD]
    uint16_t* Bh = lds + 2 * a.Npad * MT_LD;              // [2 planes][Gpad][MT_LD]
    const bool al4 = (a.HW & 3) == 0;
    const int nchunks = (int)((a.HW + MT_K - 1) / MT_K);
    const int c0 = split * a.chunks_per_split;
    const int c1 = c0 + a.chunks_per_split < nchunks ? c0 + a.chunks_per_split : nchunks;
    const float* lg = a.logits + (int64_t)b * a.N * a.HW;
    const float* gt = a.gt + (int64_t)b * a.G * a.HW;
    const float* vl = a.valid ? a.valid + (int64_t)b * a.HW : nullptr;

    f32x16_t acc[RTW][NCT];
#pragma unroll
    for (int i = 0; i < RTW; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int NP = RTW * 4 * 2;       // 16-row passes over the logits tile (Npad = RTW * 128 at most)
    constexpr int GP = NCT * 2;           // 16-row passes over the gt tile
    float q[NP], cc[GP], tt[GP], vsum = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = 0.f;
#pragma unroll
    for (int i = 0; i < GP; ++i) { cc[i] = 0.f; tt[i] = 0.f; }

    // the chunk after the one being multiplied is already in flight in registers (raw fp32, converted when it is staged)
    float zr[NP][4], tr[GP][4], vr[4];
    auto fetch = [&](int c, bool more) {
        // `more` false: the same loads on element 0 (one cached line) instead of a branch around them
        const int64_t px = more ? (int64_t)c * MT_K + c4 : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t pe = px + e < a.HW ? px + e : a.HW - 1;
            vr[e] = vl ? vl[more ? pe : 0] : 1.f;
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            int row = ps * 16 + r16;
            if (row >= a.N) row = a.N - 1;                          // clamped: zeroed at staging
            const float* src = lg + (more ? (int64_t)row * a.HW : 0);
            if (al4) {
                const int64_t pc = px + 4 <= a.HW ? px : a.HW - 4;
                const float4 t4 = *(const float4*)(src + pc);
                zr[ps][0] = t4.x; zr[ps][1] = t4.y; zr[ps][2] = t4.z; zr[ps][3] = t4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) zr[ps][e] = src[px + e < a.HW ? px + e : a.HW - 1];
            }
        }
#pragma unroll
        for (int ps = 0; ps < GP; ++ps) {
            int row = ps * 16 + r16;
            if (row >= a.G) row = a.G > 0 ? a.G - 1 : 0;
            const float* src = a.G > 0 ? gt + (more ? (int64_t)row * a.HW : 0) : lg;
            if (al4) {
                const int64_t pc = px + 4 <= a.HW ? px : a.HW - 4;
                const float4 t4 = *(const float4*)(src + pc);
                tr[ps][0] = t4.x; tr[ps][1] = t4.y; tr[ps][2] = t4.z; tr[ps][3] = t4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) tr[ps][e] = src[px + e < a.HW ? px + e : a.HW - 1];
            }
        }
    };
    fetch(c0, c0 < c1);
    for (int c = c0; c < c1; ++c) {
        const int64_t px = (int64_t)c * MT_K + c4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (px + e < a.HW) ? vr[e] : 0.f;
        vsum += (r16 == 0) ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
        __syncthreads();                                   // the previous chunk's MFMAs are done with the tiles
        // logits -> p = sigmoid -> hi/lo planes; rows >= N and pixels >= HW are zero
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int row = ps * 16 + r16;
            if (row < a.Npad) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p = 0.f;
                    if (row < a.N && px + e < a.HW) p = fast_sigmoid(zr[ps][e]);
                    q[ps] += p * p * v[e];
                    f2bf_split(p, hi[e], lo[e]);
                }
                *(uint2*)(Ah + row * MT_LD + c4) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
                *(uint2*)(Ah + a.Npad * MT_LD + row * MT_LD + c4) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
            }
        }
        // ground truth rows times v; row G is v itself (its column of A is S)
#pragma unroll
        for (int ps = 0; ps < GP; ++ps) {
            const int row = ps * 16 + r16;
            if (row < a.Gpad) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = (row < a.G && px + e < a.HW) ? tr[ps][e] : 0.f;
                    if (row < a.G) { cc[ps] += t * t * v[e]; tt[ps] += t * v[e]; }
                    const float tv = (row == a.G) ? v[e] : t * v[e];
                    f2bf_split(tv, hi[e], lo[e]);
                }
                *(uint2*)(Bh + row * MT_LD + c4) = make_uint2(pack2(hi[0], hi[1]), pack2(hi[2], hi[3]));
                *(uint2*)(Bh + a.Gpad * MT_LD + row * MT_LD + c4) = make_uint2(pack2(lo[0], lo[1]), pack2(lo[2], lo[3]));
            }
        }
        __syncthreads();
        fetch(c + 1, c + 1 < c1);                           // in flight during the MFMAs
#pragma unroll
        for (int i = 0; i < RTW; ++i) {
            const int rt = wave + i * 4;                    // uniform
            if (rt * 32 < a.Npad) {
#pragma unroll
                for (int ks = 0; ks < MT_K / 16; ++ks) {
                    const int ko = ks * 16 + (lane >> 5) * 8;
                    const uint4 ah = *(const uint4*)(Ah + (rt * 32 + (lane & 31)) * MT_LD + ko);
                    const uint4 al = *(const uint4*)(Ah + a.Npad * MT_LD + (rt * 32 + (lane & 31)) * MT_LD + ko);
#pragma unroll
                    for (int j = 0; j < NCT; ++j) {
                        if (j * 32 < a.Gpad) {
                            const uint4 bh = *(const uint4*)(Bh + (j * 32 + (lane & 31)) * MT_LD + ko);
                            const uint4 bl = *(const uint4*)(Bh + a.Gpad * MT_LD + (j * 32 + (lane & 31)) * MT_LD + ko);
                            acc[i][j] = mfma32(ah, bh, acc[i][j]);
                            acc[i][j] = mfma32(ah, bl, acc[i][j]);
                            acc[i][j] = mfma32(al, bh, acc[i][j]);
                        }
                    }
                }
            }
        }
    }
    // ---- partial record of this split ----
    float* out = a.part + ((int64_t)b * a.nsplit + split) * a.stride;
#pragma unroll
    for (int i = 0; i < RTW; ++i) {
        const int rt = wave + i * 4;
        if (rt * 32 < a.Npad) {
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                if (j * 32 < a.Gpad) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = j * 32 + (lane & 31);
                        out[row * a.Gpad + col] = acc[i][j][r];
                    }
                }
            }
        }
    }
    float* oq = out + a.Npad * a.Gpad;
    float* oc = oq + a.Npad;
    float* ot = oc + a.Gpad;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const float s = wave_group16_sum(q[ps]);
        const int row = ps * 16 + r16;
        if ((tid & 15) == 0 && row < a.Npad) oq[row] = s;
    }
#pragma unroll
    for (int ps = 0; ps < GP; ++ps) {
        const float s1 = wave_group16_sum(cc[ps]), s2 = wave_group16_sum(tt[ps]);
        const int row = ps * 16 + r16;
        if ((tid & 15) == 0 && row < a.Gpad) { oc[row] = s1; ot[row] = s2; }
    }
    const float vs = wave_group16_sum(vsum);               // threads 0..15 (r16 == 0) carry the chunk's v
    if (tid == 0) ot[a.Gpad] = vs;
}

extern "C" int64_t ph_match_record_floats(int N, int G) {
    const int Npad = ph_n_padded(N), Gpad = ph_n_padded(G + 1);
    return (int64_t)Npad * Gpad + Npad + 2 * Gpad + 1;
}

extern "C" int ph_match_nsplit(int64_t HW, int B) {
    const int nchunks = (int)((HW + MT_K - 1) / MT_K);
    int ns = (512 + B - 1) / B;                             // ~2 workgroups per CU over the batch
    if (ns > nchunks) ns = nchunks;
    if (ns < 1) ns = 1;
    const int cps = (nchunks + ns - 1) / ns;
    return (nchunks + cps - 1) / cps;
}

extern "C" int ph_match_sums(const float* logits, const float* gt, const float* valid, float* partial, int B, int N, int G,
                             int64_t HW, void* stream) {
    PH_CHECK_ARG(logits && partial && (gt || G == 0), "null pointer");
    PH_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && G >= 0 && HW > 0, "bad size");
    const int Npad = ph_n_padded(N), Gpad = ph_n_padded(G + 1);
    PH_CHECK_ARG(Npad <= 256 && Gpad <= 128, "at most 256 predictions and 127 ground-truth masks per image");
    MatchArgs a;
    a.logits = logits; a.gt = gt; a.valid = valid; a.part = partial;
    a.N = N; a.G = G; a.Npad = Npad; a.Gpad = Gpad; a.HW = HW;
    a.nsplit = ph_match_nsplit(HW, B);
    const int nchunks = (int)((HW + MT_K - 1) / MT_K);
    a.chunks_per_split = (nchunks + a.nsplit - 1) / a.nsplit;
    a.stride = ph_match_record_floats(N, G);
    const size_t lds = (size_t)2 * (Npad + Gpad) * MT_LD * sizeof(uint16_t);
    const dim3 grid(a.nsplit, B), block(MT_THREADS);
    const int rtw = Npad > 128 ? 2 : 1, nct = Gpad / 32;
    hipStream_t s = (hipStream_t)stream;
#define MT_CASE(R, C)                                                                                             \
    if (rtw == R && nct <= C) {                                                                                   \
        static const bool once = [&] {                                                                                              \
            (void)hipFuncSetAttribute((const void*)k_match_sums<R, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            return true;                                                                                          \
        }();                                                                                          \
        (void)once;                                                                                                         \
        hipLaunchKernelGGL((k_match_sums<R, C>), grid, block, lds, s, a);                                          \
        PH_CHECK_LAUNCH();                                                                                        \
        return PH_OK;                                                                                             \
    }
    MT_CASE(1, 1) MT_CASE(1, 2) MT_CASE(1, 4) MT_CASE(2, 1) MT_CASE(2, 2) MT_CASE(2, 4)
#undef MT_CASE
    ph_set_error("ph_match_sums: unsupported tile combination");
    return PH_EINVAL;
}

// ---- DepthCost (polyphonic/funcs/assigner.py:17-80): per (prediction n, ground truth g) the DepthMatchLoss sums over the
// pixels where gt_depth * gt_mask[g] > 0 -- with d = depth_act(z_n) + eps and t = gt_depth * gt_mask[g] + eps:
//   out[n][g] = { sum (log d - log t)^2, sum (log d - log t), sum ((d - t) / t)^2, sum |(d - t) / t| },  nvalid[g] = #pixels.
// The absolute value does not factor into a GEMM, so this is a direct pass: one workgroup per (n, g), pixels strided over
// the threads, pixels outside the (small) instance mask skipped before any transcendental; fixed-order reduction.
// (Outside the valid pixels the reference's eps-shifted zeros cancel exactly: log(eps) - log(eps), eps - eps.)
namespace {
constexpr int DC_T = 256;
__global__ __launch_bounds__(DC_T) void k_depth_cost(const float* __restrict__ z, const float* __restrict__ gt_depth,
                                                     const float* __restrict__ gt_masks, int G, int64_t HW, int mode, float eps,
                                                     float* __restrict__ out, float* __restrict__ nvalid) {
    __shared__ double lds[4][5];
    const int g = blockIdx.x, n = blockIdx.y;
    const float* zn = z + (int64_t)n * HW;
    const float* mg = gt_masks + (int64_t)g * HW;
    double v[5] = {0, 0, 0, 0, 0};
    for (int64_t p = threadIdx.x; p < HW; p += DC_T) {
        const float tm = gt_depth[p] * mg[p];
        if (!(tm > 0.f)) continue;
        const float s = 1.f / (1.f + expf(-zn[p]));
        float d;
        if (mode == 0) d = s * (80.f - 0.01f) + 0.01f;
        else d = 1.f / (1.f / 80.f + (1.f / 0.01f - 1.f / 80.f) * s);
        d += eps;
        const float t = tm + eps;
        const float lm = logf(d) - logf(t), r = (d - t) / t;
        v[0] += (double)(lm * lm);
        v[1] += (double)lm;
        v[2] += (double)(r * r);
        v[3] += (double)fabsf(r);
        v[4] += 1.0;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) lds[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k] = lds[0][k] + lds[1][k] + lds[2][k] + lds[3][k];
        float* o = out + ((int64_t)n * G + g) * 4;
        o[0] = (float)t[0]; o[1] = (float)t[1]; o[2] = (float)t[2]; o[3] = (float)t[3];
        if (n == 0) nvalid[g] = (float)t[4];
    }
}
}  // namespace

extern "C" int ph_depth_cost_sums(const float* depth_logits /* [N][HW] */, const float* gt_depth /* [HW] */,
                                  const float* gt_masks /* [G][HW] */, int N, int G, int64_t HW, int depth_mode, float eps,
                                  float* out /* [N][G][4] */, float* nvalid /* [G] */, void* stream) {
    PH_CHECK_ARG(depth_logits && gt_depth && gt_masks && out && nvalid, "null pointer");
    PH_CHECK_ARG(N > 0 && N <= 65535 && G > 0 && HW > 0 && (depth_mode == 0 || depth_mode == 1), "bad size or mode");
    hipLaunchKernelGGL(k_depth_cost, dim3(G, N), dim3(DC_T), 0, (hipStream_t)stream, depth_logits, gt_depth, gt_masks, G, HW, depth_mode,
                       eps, out, nvalid);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
