// SURVEY.md 8f row N4 -- the pixel passes of one stage's training losses (polyphonic/kernel_update_head.py:355-441):
//   ph_mask_loss_sums / _grad : loss_mask (BCE with logits, mmdet cross_entropy_loss.py:74-113) and loss_dice (dice_loss.py:9-46)
//                               over the valid pixels of the POSITIVE prediction rows
//   ph_rank_loss_sum  / _grad : loss_rank, a softmax cross entropy over the N mask channels per pixel (:9-47, ignore_index)
//   ph_depth_loss_sums / _grad: DepthLoss (polyphonic/losses/depth_loss.py:9-65): scale-invariant, squared-relative and
//                               absolute-relative error over the pixels with 0 < target < 80 and a non-zero weight
//   ph_focal_loss_sum / _grad : FocalLoss (focal_loss.py:12-60) on the [rows][classes] scores
// The reference evaluates these as ~40 ATen launches per stage with boolean-mask gathers (pred[weights] copies of every
// positive mask); here each loss is ONE pass that produces fixed-order partial sums (fp64 accumulators, one record per
// workgroup, combined in index order by the caller: bit-reproducible) and ONE pass that writes d loss / d logits, the
// first step of the backward pass.  Pure HBM streaming kernels, 16 bytes per lane where the layout allows.
#include "ph_common.h"

constexpr int LOSS_T = 256;

// block-wide sum of K doubles per thread -> thread 0 holds the totals (fixed order: lanes by xor-butterfly, waves ascending)
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* lds /* [4][K] */) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = lds[k] + lds[K + k] + lds[2 * K + k] + lds[3 * K + k];
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }
// binary_cross_entropy_with_logits: max(z, 0) - z t + log1p(exp(-|z|))
__device__ __forceinline__ float bce_logits(float z, float t) { return fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z))); }

// ---- mask: BCE + dice over the weighted pixels of the positive rows -----------------------------------------------------
// out [P][nsplit][5] doubles: sum bce, count, a = sum sig*t, b = sum sig^2, c = sum t^2
__global__ __launch_bounds__(LOSS_T) void k_mask_loss_sums(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                           const float* __restrict__ wgt, const int* __restrict__ rows, int64_t HW,
                                                           int nsplit, double* __restrict__ out) {
    __shared__ double lds[4 * 5];
    const int p = blockIdx.y, sp = blockIdx.x;
    const int64_t base = (int64_t)rows[p] * HW;
    const int64_t i0 = HW * sp / nsplit, i1 = HW * (sp + 1) / nsplit;
    double v[5] = {0, 0, 0, 0, 0};
    for (int64_t i = i0 + threadIdx.x; i < i1; i += LOSS_T) {
        if (wgt[base + i] == 0.f) continue;                  // mask_weights[pos_inds].bool()   (:413)
        const float z = pred[base + i], t = tgt[base + i], s = sigmoidf_(z);
        v[0] += (double)bce_logits(z, t);
        v[1] += 1.0;
        v[2] += (double)(s * t);
        v[3] += (double)(s * s);
        v[4] += (double)(t * t);
    }
    block_sum<5>(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) out[((int64_t)p * nsplit + sp) * 5 + k] = v[k];
}

// grad[row][i] += w * ( c_bce (sig - t) + dice_row terms ); coef [P][3] floats: bce scale, dice A = -2 lw / (P (b + c)),
// dice B = 4 lw a / (P (b + c)^2):  d dice / dz = (A t + B sig) sig (1 - sig)
__global__ __launch_bounds__(LOSS_T) void k_mask_loss_grad(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                           const float* __restrict__ wgt, const int* __restrict__ rows, int64_t HW,
                                                           const float* __restrict__ coef, float* __restrict__ grad) {
    const int p = blockIdx.y;
    const int64_t base = (int64_t)rows[p] * HW;
    const float cb = coef[p * 3], cA = coef[p * 3 + 1], cB = coef[p * 3 + 2];
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < HW; i += (int64_t)gridDim.x * LOSS_T) {
        if (wgt[base + i] == 0.f) continue;
        const float z = pred[base + i], t = tgt[base + i], s = sigmoidf_(z);
        grad[base + i] += cb * (s - t) + (cA * t + cB * s) * s * (1.f - s);
    }
}

// ---- rank: softmax cross entropy over the N channels of every pixel -------------------------------------------------------
// One thread owns V consecutive pixels (V = 4: 16-byte loads when HW % 4 == 0) and walks the N rows once with an online
// softmax (running max m and sum s of exp(z - m)), four rows of loads in flight; the first version walked the rows three
// times with one dependent 4-byte load per step (0.6 TB/s).
template <int V> struct PxVec;
template <> struct PxVec<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) { const float4 q = *(const float4*)p; v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    __device__ __forceinline__ void store(float* p) const { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct PxVec<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = v[0]; }
};

// m[j], s[j]: running maximum and sum of exp(z - m) over the N rows of pixel i + j
template <int V>
__device__ __forceinline__ void online_softmax(const float* __restrict__ pb, int N, int64_t HW, int64_t i, float* m, float* s) {
#pragma unroll
    for (int j = 0; j < V; ++j) { m[j] = -INFINITY; s[j] = 0.f; }
    int n = 0;
    for (; n + 4 <= N; n += 4) {
        PxVec<V> r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k].load(pb + (int64_t)(n + k) * HW + i);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float cm = fmaxf(fmaxf(r[0].v[j], r[1].v[j]), fmaxf(r[2].v[j], r[3].v[j]));
            const float nm = fmaxf(m[j], cm);
            s[j] = s[j] * __expf(m[j] - nm) + __expf(r[0].v[j] - nm) + __expf(r[1].v[j] - nm) + __expf(r[2].v[j] - nm) + __expf(r[3].v[j] - nm);
            m[j] = nm;
        }
    }
    for (; n < N; ++n) {
        PxVec<V> r;
        r.load(pb + (int64_t)n * HW + i);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float nm = fmaxf(m[j], r.v[j]);
            s[j] = s[j] * __expf(m[j] - nm) + __expf(r.v[j] - nm);
            m[j] = nm;
        }
    }
}

// out [B * nblk] doubles: sum over the non-ignored pixels of (logsumexp - z_target)
template <int V>
__global__ __launch_bounds__(LOSS_T) void k_rank_loss_sum(const float* __restrict__ pred, const int* __restrict__ target, int N,
                                                          int64_t HW, int ignore, double* __restrict__ out) {
    __shared__ double lds[4];
    const int b = blockIdx.y;
    const float* pb = pred + (int64_t)b * N * HW;
    const int* tb = target + (int64_t)b * HW;
    double v[1] = {0};
    for (int64_t i = (blockIdx.x * (int64_t)LOSS_T + threadIdx.x) * V; i < HW; i += (int64_t)gridDim.x * LOSS_T * V) {
        int t[V];
        bool any = false;
#pragma unroll
        for (int j = 0; j < V; ++j) { t[j] = tb[i + j]; any |= t[j] != ignore; }
        if (!any) continue;
        float m[V], s[V];
        online_softmax<V>(pb, N, HW, i, m, s);
#pragma unroll
        for (int j = 0; j < V; ++j)
            if (t[j] != ignore) v[0] += (double)(m[j] + logf(s[j]) - pb[(int64_t)t[j] * HW + i + j]);
    }
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) out[(int64_t)b * gridDim.x + blockIdx.x] = v[0];
}

// grad[b][n][i] = scale * (softmax_n - [n == target]) on the non-ignored pixels, 0 elsewhere (OVERWRITES grad)
template <int V>
__global__ __launch_bounds__(LOSS_T) void k_rank_loss_grad(const float* __restrict__ pred, const int* __restrict__ target, int N,
                                                           int64_t HW, int ignore, float scale, float* __restrict__ grad) {
    const int b = blockIdx.y;
    const float* pb = pred + (int64_t)b * N * HW;
    float* gb = grad + (int64_t)b * N * HW;
    for (int64_t i = (blockIdx.x * (int64_t)LOSS_T + threadIdx.x) * V; i < HW; i += (int64_t)gridDim.x * LOSS_T * V) {
        int t[V];
        bool any = false;
#pragma unroll
        for (int j = 0; j < V; ++j) { t[j] = target ? target[(int64_t)b * HW + i + j] : ignore; any |= t[j] != ignore; }
        float m[V], inv[V];
        if (any) {
            float s[V];
            online_softmax<V>(pb, N, HW, i, m, s);
#pragma unroll
            for (int j = 0; j < V; ++j) inv[j] = t[j] != ignore ? scale / s[j] : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) { m[j] = 0.f; inv[j] = 0.f; }
        }
        for (int n = 0; n < N; ++n) {
            PxVec<V> r{}, g;
            if (any) r.load(pb + (int64_t)n * HW + i);              // second walk: the rows are still in L2
#pragma unroll
            for (int j = 0; j < V; ++j)
                g.v[j] = inv[j] != 0.f ? inv[j] * __expf(r.v[j] - m[j]) - (n == t[j] ? scale : 0.f) : 0.f;
            g.store(gb + (int64_t)n * HW + i);
        }
    }
}

// ---- depth ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float depth_act_f(float z, int mode) {
    const float s = sigmoidf_(z);
    if (mode == 0) return s * (80.f - 0.01f) + 0.01f;                    // funcs/depth_utils.py 'sigmoid'
    const float mn = 1.f / 80.f, mxd = 1.f / 0.01f;
    return 1.f / (mn + (mxd - mn) * s);                                  // 'monodepth'
}
// d depth_act / dz
__device__ __forceinline__ float depth_act_df(float z, int mode) {
    const float s = sigmoidf_(z), ds = s * (1.f - s);
    if (mode == 0) return (80.f - 0.01f) * ds;
    const float mn = 1.f / 80.f, mxd = 1.f / 0.01f, q = mn + (mxd - mn) * s;
    return -(mxd - mn) * ds / (q * q);
}
// out [nblk][5] doubles: n, sum lm^2, sum lm, sum (m / t)^2, sum |m / t|   (lm = (log p - log t) w, m = (p - t) w)
__global__ __launch_bounds__(LOSS_T) void k_depth_loss_sums(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                            const float* __restrict__ wgt, int64_t total, int mode,
                                                            double* __restrict__ out) {
    __shared__ double lds[4 * 5];
    double v[5] = {0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < total; i += (int64_t)gridDim.x * LOSS_T) {
        const float t = tgt[i], w = wgt[i];
        if (!(t > 0.f && t < 80.f && w != 0.f)) continue;
        const float p = depth_act_f(pred[i], mode);
        const float lm = (logf(p) - logf(t)) * w, r = (p - t) * w / t;
        v[0] += 1.0;
        v[1] += (double)(lm * lm);
        v[2] += (double)lm;
        v[3] += (double)(r * r);
        v[4] += (double)fabsf(r);
    }
    block_sum<5>(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) out[(int64_t)blockIdx.x * 5 + k] = v[k];
}
// grad = [ c0 lm (w / p) + c1 (w / p) + c2 r (w / t) + c3 sign(r) (w / t) ] * d depth_act / dz   (OVERWRITES grad)
__global__ __launch_bounds__(LOSS_T) void k_depth_loss_grad(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                            const float* __restrict__ wgt, int64_t total, int mode, float c0, float c1,
                                                            float c2, float c3, float* __restrict__ grad) {
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < total; i += (int64_t)gridDim.x * LOSS_T) {
        const float t = tgt[i], w = wgt[i];
        float g = 0.f;
        if (t > 0.f && t < 80.f && w != 0.f) {
            const float z = pred[i], p = depth_act_f(z, mode);
            const float lm = (logf(p) - logf(t)) * w, r = (p - t) * w / t;
            const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
            g = ((c0 * lm + c1) * (w / p) + (c2 * r + c3 * sg) * (w / t)) * depth_act_df(z, mode);
        }
        grad[i] = g;
    }
}

// ---- focal ------------------------------------------------------------------------------------------------------------------
// pred [R][L], labels [R] (class index; >= L = background), weight [R][L]; out [nblk] doubles / grad [R][L]
template <bool GRAD>
__global__ __launch_bounds__(LOSS_T) void k_focal(const float* __restrict__ pred, const int64_t* __restrict__ labels,
                                                  const float* __restrict__ wgt, int64_t R, int L, float gamma, float alpha, float scale,
                                                  double* __restrict__ out, float* __restrict__ grad) {
    __shared__ double lds[4];
    double v[1] = {0};
    const int64_t total = R * L;
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < total; i += (int64_t)gridDim.x * LOSS_T) {
        const int64_t r = i / L;
        const int c = (int)(i - r * L);
        const float z = pred[i], t = labels[r] == c ? 1.f : 0.f, p = sigmoidf_(z), w = wgt[i];
        const float pt = (1.f - p) * t + p * (1.f - t);
        const float at = alpha * t + (1.f - alpha) * (1.f - t);
        const float bce = bce_logits(z, t);
        if (!GRAD) {
            v[0] += (double)(bce * at * powf(pt, gamma) * w);
        } else {
            // d/dz [ at pt^g bce ] = at ( g pt^(g-1) dpt bce + pt^g (p - t) ),  dpt/dz = (1 - 2 t) p (1 - p)
            const float dpt = (1.f - 2.f * t) * p * (1.f - p);
            grad[i] = scale * w * at * (gamma * powf(pt, gamma - 1.f) * dpt * bce + powf(pt, gamma) * (p - t));
        }
    }
    if (!GRAD) {
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) out[blockIdx.x] = v[0];
    }
}

// ---- focal loss over a class-major map: pred [B][L][HW], target [B][HW] (class index; == L: pixel not selected) -------------
// KernelHead's loss_rpn_seg (kernel_head.py:538-551): the pixels with seg_targets != L, all L classes of each
template <bool GRAD>
__global__ __launch_bounds__(LOSS_T) void k_seg_focal(const float* __restrict__ pred, const int* __restrict__ target, int L, int64_t HW,
                                                      float gamma, float alpha, float scale, double* __restrict__ out,
                                                      float* __restrict__ grad) {
    __shared__ double lds[4];
    const int b = blockIdx.y;
    const float* pb = pred + (int64_t)b * L * HW;
    double v[1] = {0};
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < HW; i += (int64_t)gridDim.x * LOSS_T) {
        const int tg = target[(int64_t)b * HW + i];
        const bool sel = tg != L;
        for (int c = 0; c < L; ++c) {
            const int64_t idx = (int64_t)c * HW + i;
            if (!sel) {
                if (GRAD) grad[(int64_t)b * L * HW + idx] = 0.f;
                continue;
            }
            const float z = pb[idx], t = tg == c ? 1.f : 0.f, p = sigmoidf_(z);
            const float pt = (1.f - p) * t + p * (1.f - t);
            const float at = alpha * t + (1.f - alpha) * (1.f - t);
            const float bce = bce_logits(z, t);
            // gamma == 2 (every shipped config): two multiplies instead of two powf
            const float ptg = gamma == 2.f ? pt * pt : powf(pt, gamma), ptg1 = gamma == 2.f ? pt : powf(pt, gamma - 1.f);
            if (!GRAD) v[0] += (double)(bce * at * ptg);
            else {
                const float dpt = (1.f - 2.f * t) * p * (1.f - p);
                grad[(int64_t)b * L * HW + idx] = scale * at * (gamma * ptg1 * dpt * bce + ptg * (p - t));
            }
        }
    }
    if (!GRAD) {
        block_sum<1>(v, lds);
        if (threadIdx.x == 0) out[(int64_t)b * gridDim.x + blockIdx.x] = v[0];
    }
}

// =================================================================================================================================
static int loss_grid(int64_t n, int cap) {
    int64_t g = (n + LOSS_T - 1) / LOSS_T;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int ph_mask_loss_sums(const float* pred, const float* target, const float* weight, const int32_t* pos_rows, int P,
                                 int64_t HW, int nsplit, double* out, void* stream) {
    PH_CHECK_ARG(pred && target && weight && pos_rows && out && P > 0 && HW > 0 && nsplit >= 1 && nsplit <= 1024, "bad pointer or size");
    hipLaunchKernelGGL(k_mask_loss_sums, dim3(nsplit, P), dim3(LOSS_T), 0, (hipStream_t)stream, pred, target, weight, pos_rows, HW,
                       nsplit, out);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_mask_loss_grad(const float* pred, const float* target, const float* weight, const int32_t* pos_rows, int P,
                                 int64_t HW, const float* coef, float* grad, void* stream) {
    PH_CHECK_ARG(pred && target && weight && pos_rows && coef && grad && P > 0 && HW > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_mask_loss_grad, dim3(loss_grid(HW, 64), P), dim3(LOSS_T), 0, (hipStream_t)stream, pred, target, weight,
                       pos_rows, HW, coef, grad);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_rank_loss_blocks(int64_t HW) { return loss_grid(HW, 256); }

extern "C" int ph_rank_loss_sum(const float* pred, const int32_t* rank_target, int B, int N, int64_t HW, int ignore_index,
                                double* out /* [B * ph_rank_loss_blocks(HW)] */, void* stream) {
    PH_CHECK_ARG(pred && rank_target && out && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    if (HW % 4 == 0 && ((uintptr_t)pred & 15) == 0)
        hipLaunchKernelGGL(k_rank_loss_sum<4>, dim3(loss_grid(HW, 256), B), dim3(LOSS_T), 0, (hipStream_t)stream, pred, rank_target, N, HW,
                           ignore_index, out);
    else
        hipLaunchKernelGGL(k_rank_loss_sum<1>, dim3(loss_grid(HW, 256), B), dim3(LOSS_T), 0, (hipStream_t)stream, pred, rank_target, N, HW,
                           ignore_index, out);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_rank_loss_grad(const float* pred, const int32_t* rank_target /* NULL: all zero */, int B, int N, int64_t HW,
                                 int ignore_index, float scale, float* grad, void* stream) {
    PH_CHECK_ARG(pred && grad && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    if (HW % 4 == 0 && (((uintptr_t)pred | (uintptr_t)grad) & 15) == 0)
        hipLaunchKernelGGL(k_rank_loss_grad<4>, dim3(loss_grid(HW / 4, 1024), B), dim3(LOSS_T), 0, (hipStream_t)stream, pred, rank_target, N,
                           HW, ignore_index, scale, grad);
    else
        hipLaunchKernelGGL(k_rank_loss_grad<1>, dim3(loss_grid(HW, 1024), B), dim3(LOSS_T), 0, (hipStream_t)stream, pred, rank_target, N, HW,
                           ignore_index, scale, grad);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_depth_loss_blocks(int64_t total) { return loss_grid(total, 1024); }

extern "C" int ph_depth_loss_sums(const float* pred, const float* target, const float* weight, int64_t total, int depth_mode,
                                  double* out /* [ph_depth_loss_blocks(total)][5] */, void* stream) {
    PH_CHECK_ARG(pred && target && weight && out && total > 0 && (depth_mode == 0 || depth_mode == 1), "bad pointer, size or mode");
    hipLaunchKernelGGL(k_depth_loss_sums, dim3(loss_grid(total, 1024)), dim3(LOSS_T), 0, (hipStream_t)stream, pred, target, weight,
                       total, depth_mode, out);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_depth_loss_grad(const float* pred, const float* target, const float* weight, int64_t total, int depth_mode,
                                  float c0, float c1, float c2, float c3, float* grad, void* stream) {
    PH_CHECK_ARG(pred && target && weight && grad && total > 0 && (depth_mode == 0 || depth_mode == 1), "bad pointer, size or mode");
    hipLaunchKernelGGL(k_depth_loss_grad, dim3(loss_grid(total, 2048)), dim3(LOSS_T), 0, (hipStream_t)stream, pred, target, weight,
                       total, depth_mode, c0, c1, c2, c3, grad);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_focal_loss_blocks(int64_t total) { return loss_grid(total, 256); }

extern "C" int ph_focal_loss_sum(const float* pred, const int64_t* labels, const float* weight, int64_t R, int L, float gamma,
                                 float alpha, double* out /* [ph_focal_loss_blocks(R * L)] */, void* stream) {
    PH_CHECK_ARG(pred && labels && weight && out && R > 0 && L > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_focal<false>, dim3(loss_grid(R * L, 256)), dim3(LOSS_T), 0, (hipStream_t)stream, pred, labels, weight, R, L,
                       gamma, alpha, 0.f, out, (float*)nullptr);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_focal_loss_grad(const float* pred, const int64_t* labels, const float* weight, int64_t R, int L, float gamma,
                                  float alpha, float scale, float* grad, void* stream) {
    PH_CHECK_ARG(pred && labels && weight && grad && R > 0 && L > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_focal<true>, dim3(loss_grid(R * L, 256)), dim3(LOSS_T), 0, (hipStream_t)stream, pred, labels, weight, R, L,
                       gamma, alpha, scale, (double*)nullptr, grad);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_seg_focal_sum(const float* pred, const int32_t* target, int B, int L, int64_t HW, float gamma, float alpha,
                                double* out /* [B * ph_rank_loss_blocks(HW)] */, void* stream) {
    PH_CHECK_ARG(pred && target && out && B > 0 && L > 0 && HW > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_seg_focal<false>, dim3(loss_grid(HW, 256), B), dim3(LOSS_T), 0, (hipStream_t)stream, pred, target, L, HW, gamma,
                       alpha, 0.f, out, (float*)nullptr);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_seg_focal_grad(const float* pred, const int32_t* target, int B, int L, int64_t HW, float gamma, float alpha,
                                 float scale, float* grad, void* stream) {
    PH_CHECK_ARG(pred && target && grad && B > 0 && L > 0 && HW > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_seg_focal<true>, dim3(loss_grid(HW, 256), B), dim3(LOSS_T), 0, (hipStream_t)stream, pred, target, L, HW, gamma,
                       alpha, scale, (double*)nullptr, grad);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// ---- target assembly -----------------------------------------------------------------------------------------------------------
// rank target (kernel_update_head.py:420-432, kernel_head.py:516-528): pixel -> index, within its image, of the LAST positive
// row whose mask target covers it (the reference paints the rows one after the other), `ignore` where none does.
__global__ __launch_bounds__(LOSS_T) void k_rank_target(const float* __restrict__ mask_targets, const uint8_t* __restrict__ pos, int N,
                                                        int64_t HW, int ignore, int* __restrict__ out) {
    const int b = blockIdx.y;
    for (int64_t p = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; p < HW; p += (int64_t)gridDim.x * LOSS_T) {
        int t = ignore;
        for (int j = 0; j < N; ++j)
            if (pos[b * N + j] && mask_targets[((int64_t)b * N + j) * HW + p] != 0.f) t = j;      // .bool()
        out[(int64_t)b * HW + p] = t;
    }
}

// dense semantic target of ONE image (kernel_head.py:590-605): background L, then the stuff masks in order, then the assigned
// thing masks in order, each painting its class over what is there
__global__ __launch_bounds__(LOSS_T) void k_seg_target(const float* __restrict__ sem_seg, const int64_t* __restrict__ sem_cls, int S,
                                                       const float* __restrict__ pos_masks, const int64_t* __restrict__ pos_labels, int P,
                                                       int L, int64_t HW, int64_t* __restrict__ out) {
    for (int64_t p = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; p < HW; p += (int64_t)gridDim.x * LOSS_T) {
        int64_t t = L;
        for (int s = 0; s < S; ++s)
            if (sem_seg[(int64_t)s * HW + p] != 0.f) t = sem_cls[s];
        for (int i = 0; i < P; ++i)
            if (pos_masks[(int64_t)i * HW + p] != 0.f) t = pos_labels[i];
        out[p] = t;
    }
}

extern "C" int ph_rank_target(const float* mask_targets /* [B*N][HW] */, const uint8_t* pos /* [B*N] */, int B, int N, int64_t HW,
                              int ignore_index, int32_t* out /* [B][HW] */, void* stream) {
    PH_CHECK_ARG(mask_targets && pos && out && B > 0 && N > 0 && HW > 0, "bad pointer or size");
    hipLaunchKernelGGL(k_rank_target, dim3(loss_grid(HW, 1024), B), dim3(LOSS_T), 0, (hipStream_t)stream, mask_targets, pos, N, HW,
                       ignore_index, out);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_seg_target(const float* sem_seg /* [S][HW], may be NULL when S == 0 */, const int64_t* sem_cls, int S,
                             const float* pos_masks /* [P][HW], may be NULL when P == 0 */, const int64_t* pos_labels, int P, int L,
                             int64_t HW, int64_t* out /* [HW] */, void* stream) {
    PH_CHECK_ARG(out && S >= 0 && P >= 0 && HW > 0 && (S == 0 || (sem_seg && sem_cls)) && (P == 0 || (pos_masks && pos_labels)),
                 "bad pointer or size");
    hipLaunchKernelGGL(k_seg_target, dim3(loss_grid(HW, 1024)), dim3(LOSS_T), 0, (hipStream_t)stream, sem_seg, sem_cls, S, pos_masks,
                       pos_labels, P, L, HW, out);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

// =================================================================================================================================
// Round 5 (VERDICT r04 #1d): ONE call per head and stage for targets + losses + d(losses)/d(predictions), from DESCRIPTORS.
// The reference materialises per stage and image labels / label_weights / mask_targets / mask_weights / depth_targets /
// depth_weights as [rows][H][W] tensors (kernel_update_head.py:443-591, kernel_head.py:571-698: index_put, cat, fill -- 30 % of
// round 4's training-step GPU time went into ATen copies of ground-truth masks) and evaluates the losses over them with
// boolean-mask gathers.  Every one of those rows is a ground-truth mask, the image's valid map, its depth map, all ones or all
// zeros -- so a row is described by POINTERS: tptr / wptr (mask target / weight row, 0 = zeros), depth items (target row, weight
// row or 1 = ones, scale) grouped per prediction row, paint lists for the dense semantic target.  The host builds the tables from
// the Hungarian result it already holds (no device gather, no D2H), the kernels read the ground truth where it lies.
// Launch sequence: [seg target, seg focal sum] mask sums, rank target, rank sum, depth sums, focal sum -> finalize (loss values,
// gradient coefficients: on the device, no host round trip) -> rank / mask / depth / focal / seg gradients.
// =================================================================================================================================
namespace {

__device__ __forceinline__ const float* as_row(int64_t p) { return (const float*)(uintptr_t)p; }

__global__ __launch_bounds__(LOSS_T) void k_mask_sums_p(const float* __restrict__ pred, const int* __restrict__ rows,
                                                        const int64_t* __restrict__ tptr, const int64_t* __restrict__ wptr, int64_t HW,
                                                        int nsplit, double* __restrict__ out) {
    __shared__ double lds[4 * 5];
    const int p = blockIdx.y, sp = blockIdx.x, row = rows[p];
    const float* z_ = pred + (int64_t)row * HW;
    const float* t_ = as_row(tptr[row]);
    const float* w_ = as_row(wptr[row]);
    const int64_t i0 = HW * sp / nsplit, i1 = HW * (sp + 1) / nsplit;
    double v[5] = {0, 0, 0, 0, 0};
    if (w_)
        for (int64_t i = i0 + threadIdx.x; i < i1; i += LOSS_T) {
            if (w_[i] == 0.f) continue;
            const float z = z_[i], t = t_ ? t_[i] : 0.f, s = sigmoidf_(z);
            v[0] += (double)bce_logits(z, t);
            v[1] += 1.0;
            v[2] += (double)(s * t);
            v[3] += (double)(s * s);
            v[4] += (double)(t * t);
        }
    block_sum<5>(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) out[((int64_t)p * nsplit + sp) * 5 + k] = v[k];
}

__global__ __launch_bounds__(LOSS_T) void k_mask_grad_p(const float* __restrict__ pred, const int* __restrict__ rows,
                                                        const int64_t* __restrict__ tptr, const int64_t* __restrict__ wptr, int64_t HW,
                                                        const float* __restrict__ coef, float* __restrict__ grad) {
    const int p = blockIdx.y, row = rows[p];
    const float* z_ = pred + (int64_t)row * HW;
    const float* t_ = as_row(tptr[row]);
    const float* w_ = as_row(wptr[row]);
    if (!w_) return;
    float* g_ = grad + (int64_t)row * HW;
    const float cb = coef[p * 3], cA = coef[p * 3 + 1], cB = coef[p * 3 + 2];
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < HW; i += (int64_t)gridDim.x * LOSS_T) {
        if (w_[i] == 0.f) continue;
        const float z = z_[i], t = t_ ? t_[i] : 0.f, s = sigmoidf_(z);
        g_[i] += cb * (s - t) + (cA * t + cB * s) * s * (1.f - s);
    }
}

// pixel -> index (within its image) of the LAST positive row whose target covers it (kernel_update_head.py:420-432)
__global__ __launch_bounds__(LOSS_T) void k_rank_target_p(const int64_t* __restrict__ tptr, const uint8_t* __restrict__ pos, int N, int64_t HW,
                                                          int ignore, int* __restrict__ out) {
    __shared__ int64_t lp[512];
    __shared__ int lj[512];
    __shared__ int cnt;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        int c = 0;
        for (int j = 0; j < N && c < 512; ++j)
            if (pos[b * N + j] && tptr[b * N + j]) { lp[c] = tptr[b * N + j]; lj[c] = j; ++c; }
        cnt = c;
    }
    __syncthreads();
    const int c = cnt;
    for (int64_t p = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; p < HW; p += (int64_t)gridDim.x * LOSS_T) {
        int t = ignore;
        for (int k = 0; k < c; ++k)
            if (as_row(lp[k])[p] != 0.f) t = lj[k];
        out[(int64_t)b * HW + p] = t;
    }
}

// depth items of prediction row r: [dstart[r], dstart[r + 1]); item = (target row, weight row (1: ones), scale)
__global__ __launch_bounds__(LOSS_T) void k_depth_sums_csr(const float* __restrict__ pred, const int* __restrict__ dstart,
                                                           const int64_t* __restrict__ it_t, const int64_t* __restrict__ it_w,
                                                           const float* __restrict__ it_s, int64_t HW, int nsplit, int mode,
                                                           double* __restrict__ out) {
    __shared__ double lds[4 * 5];
    const int r = blockIdx.y, sp = blockIdx.x;
    const float* z_ = pred + (int64_t)r * HW;
    const int64_t i0 = HW * sp / nsplit, i1 = HW * (sp + 1) / nsplit;
    double v[5] = {0, 0, 0, 0, 0};
    for (int it = dstart[r]; it < dstart[r + 1]; ++it) {
        const float* t_ = as_row(it_t[it]);
        const float* w_ = it_w[it] == 1 ? nullptr : as_row(it_w[it]);
        const float sc = it_s[it];
        if (sc == 0.f) continue;
        for (int64_t i = i0 + threadIdx.x; i < i1; i += LOSS_T) {
            const float t = t_[i], w = w_ ? w_[i] * sc : sc;
            if (!(t > 0.f && t < 80.f && w != 0.f)) continue;
            const float p = depth_act_f(z_[i], mode);
            const float lm = (logf(p) - logf(t)) * w, q = (p - t) * w / t;
            v[0] += 1.0;
            v[1] += (double)(lm * lm);
            v[2] += (double)lm;
            v[3] += (double)(q * q);
            v[4] += (double)fabsf(q);
        }
    }
    block_sum<5>(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) out[((int64_t)r * nsplit + sp) * 5 + k] = v[k];
}

__global__ __launch_bounds__(LOSS_T) void k_depth_grad_csr(const float* __restrict__ pred, const int* __restrict__ dstart,
                                                           const int64_t* __restrict__ it_t, const int64_t* __restrict__ it_w,
                                                           const float* __restrict__ it_s, int64_t HW, int mode,
                                                           const float* __restrict__ coef, float* __restrict__ grad) {
    const int r = blockIdx.y;
    const float* z_ = pred + (int64_t)r * HW;
    float* g_ = grad + (int64_t)r * HW;
    const int a = dstart[r], e = dstart[r + 1];
    const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3];
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < HW; i += (int64_t)gridDim.x * LOSS_T) {
        float g = 0.f;
        if (a < e) {
            const float z = z_[i];
            float p = 0.f, df = 0.f;
            bool have = false;
            for (int it = a; it < e; ++it) {
                const float sc = it_s[it];
                const float t = as_row(it_t[it])[i];
                const float w = it_w[it] == 1 ? sc : as_row(it_w[it])[i] * sc;
                if (!(t > 0.f && t < 80.f && w != 0.f)) continue;
                if (!have) { p = depth_act_f(z, mode); df = depth_act_df(z, mode); have = true; }
                const float lm = (logf(p) - logf(t)) * w, q = (p - t) * w / t;
                const float sg = q > 0.f ? 1.f : (q < 0.f ? -1.f : 0.f);
                g += ((c0 * lm + c1) * (w / p) + (c2 * q + c3 * sg) * (w / t)) * df;
            }
        }
        g_[i] = g;
    }
}

// dense semantic target of image b (kernel_head.py:590-605): background L, then the paint list in order; counts the selected pixels
__global__ __launch_bounds__(LOSS_T) void k_seg_target_p(const int* __restrict__ sstart, const int64_t* __restrict__ it_m,
                                                         const int* __restrict__ it_l, int L, int64_t HW, int* __restrict__ out,
                                                         int* __restrict__ nsel) {
    const int b = blockIdx.y;
    const int a = sstart[b], e = sstart[b + 1];
    int n = 0;
    for (int64_t p = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; p < HW; p += (int64_t)gridDim.x * LOSS_T) {
        int t = L;
        for (int k = a; k < e; ++k)
            if (as_row(it_m[k])[p] != 0.f) t = it_l[k];
        out[(int64_t)b * HW + p] = t;
        n += (t >= 0 && t < L) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(nsel, n);          // integer: order independent
}

__global__ __launch_bounds__(LOSS_T) void k_fill0(float* __restrict__ p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < n; i += (int64_t)gridDim.x * LOSS_T) p[i] = 0.f;
}
__global__ void k_zero_int(int* p) { *p = 0; }

struct FinArgs {
    const double *mask_part, *rank_part, *depth_part, *focal_part, *seg_part;
    int P, mask_ns, rank_n, depth_n, focal_n, seg_n;
    const int* nsel;
    ph_loss_cfg c;
    const float* cls;
    const int64_t* labels;
    const int* pos_rows;
    float* losses;        // [8]: depth, cls, mask, dice, rank, seg, pos_acc, n_dense
    float* mask_coef;     // [P][3]
    float* depth_coef;    // [4]
    float* seg_scale;     // [1]
};

// one workgroup: fixed-order sums of the partial records, the loss values, the coefficients of the gradient passes
__global__ __launch_bounds__(LOSS_T) void k_loss_finalize(const FinArgs a) {
    __shared__ double sh[LOSS_T];
    __shared__ double tot[8];
    const int t = threadIdx.x;
    auto total = [&](const double* p, int n, int stride, int off) {      // sum_{i < n} p[i * stride + off], fixed order
        double s = 0;
        for (int i = t; i < n; i += LOSS_T) s += p[(int64_t)i * stride + off];
        sh[t] = s;
        __syncthreads();
        for (int o = LOSS_T / 2; o > 0; o >>= 1) {
            if (t < o) sh[t] += sh[t + o];
            __syncthreads();
        }
        const double r = sh[0];
        __syncthreads();
        return r;
    };
    const ph_loss_cfg& c = a.c;
    // ---- masks: per positive row the 5 sums over its splits (thread p), then BCE mean / dice mean
    double bce = 0, cnt = 0, dice = 0;
    if (a.P > 0) {
        double loc_b = 0, loc_c = 0, loc_d = 0;
        for (int p = t; p < a.P; p += LOSS_T) {
            double s[5] = {0, 0, 0, 0, 0};
            for (int k = 0; k < a.mask_ns; ++k)
                for (int j = 0; j < 5; ++j) s[j] += a.mask_part[((int64_t)p * a.mask_ns + k) * 5 + j];
            const double bc = s[3] + s[4] + 2.0 * (double)c.dice_eps;
            loc_b += s[0]; loc_c += s[1]; loc_d += 1.0 - 2.0 * s[2] / bc;
            a.mask_coef[p * 3 + 1] = (float)(-2.0 * c.lw_dice / (a.P * bc));
            a.mask_coef[p * 3 + 2] = (float)(4.0 * c.lw_dice * s[2] / (a.P * bc * bc));
        }
        sh[t] = loc_b; __syncthreads();
        for (int o = LOSS_T / 2; o > 0; o >>= 1) { if (t < o) sh[t] += sh[t + o]; __syncthreads(); }
        bce = sh[0]; __syncthreads();
        sh[t] = loc_c; __syncthreads();
        for (int o = LOSS_T / 2; o > 0; o >>= 1) { if (t < o) sh[t] += sh[t + o]; __syncthreads(); }
        cnt = sh[0]; __syncthreads();
        sh[t] = loc_d; __syncthreads();
        for (int o = LOSS_T / 2; o > 0; o >>= 1) { if (t < o) sh[t] += sh[t + o]; __syncthreads(); }
        dice = sh[0]; __syncthreads();
        for (int p = t; p < a.P; p += LOSS_T) a.mask_coef[p * 3] = (float)(c.lw_mask / cnt);
    }
    const double rank = a.rank_n ? total(a.rank_part, a.rank_n, 1, 0) : 0.0;
    double d[5] = {0, 0, 0, 0, 0};
    for (int j = 0; j < 5; ++j) d[j] = a.depth_n ? total(a.depth_part, a.depth_n, 5, j) : 0.0;
    const double focal = a.focal_n ? total(a.focal_part, a.focal_n, 1, 0) : 0.0;
    const double seg = a.seg_n ? total(a.seg_part, a.seg_n, 1, 0) : 0.0;
    // pos_acc: top-1 accuracy of the positive rows (mmdet accuracy), one thread per row
    double acc = 0;
    if (a.cls && a.P > 0) {
        double hit = 0;
        for (int p = t; p < a.P; p += LOSS_T) {
            const float* row = a.cls + (int64_t)a.pos_rows[p] * c.L;
            int best = 0;
            for (int l = 1; l < c.L; ++l)
                if (row[l] > row[best]) best = l;
            hit += best == (int)a.labels[a.pos_rows[p]] ? 1.0 : 0.0;
        }
        sh[t] = hit; __syncthreads();
        for (int o = LOSS_T / 2; o > 0; o >>= 1) { if (t < o) sh[t] += sh[t + o]; __syncthreads(); }
        acc = sh[0] * 100.0 / a.P;
        __syncthreads();
    }
    if (t == 0) {
        // DepthLoss from the five sums (depth_loss.py:19-32; the reference's sum(log_minus) / n^2 kept)
        const double n = d[0];
        double ld = 0;
        float dc[4] = {0.f, 0.f, 0.f, 0.f};
        if (n > 0) {
            const double si = d[1] / n - d[2] / (n * n), sq = sqrt(d[3] / n), ab = d[4] / n, K = c.lw_depth / 3.0;
            ld = K * (c.dw_si * si + c.dw_sq * sq + c.dw_abs * ab);
            dc[0] = (float)(K * c.dw_si * 2.0 / n);
            dc[1] = (float)(-K * c.dw_si / (n * n));
            dc[2] = sq > 0 ? (float)(K * c.dw_sq / (n * sq)) : 0.f;
            dc[3] = (float)(K * c.dw_abs / n);
        }
        for (int j = 0; j < 4; ++j) a.depth_coef[j] = dc[j];
        const double nd = a.nsel ? (double)max(*a.nsel, 1) : 1.0;
        a.seg_scale[0] = (float)(c.lw_seg / nd);
        a.losses[0] = (float)ld;
        a.losses[1] = (float)(c.lw_cls * focal / (double)c.cls_avg);
        a.losses[2] = a.P > 0 ? (float)(c.lw_mask * bce / cnt) : 0.f;
        a.losses[3] = a.P > 0 ? (float)(c.lw_dice * dice / a.P) : 0.f;
        a.losses[4] = a.P > 0 ? (float)(c.lw_rank * rank / ((double)c.B * (double)c.HW)) : 0.f;
        a.losses[5] = (float)(c.lw_seg * seg / nd);
        a.losses[6] = (float)acc;
        a.losses[7] = (float)nd;
    }
}

// the class-major focal gradient with its scale read from the device (lw / number of selected pixels, from k_loss_finalize)
__global__ __launch_bounds__(LOSS_T) void k_seg_focal_grad_dev(const float* __restrict__ pred, const int* __restrict__ target, int L, int64_t HW,
                                                               float gamma, float alpha, const float* __restrict__ scale_dev,
                                                               float* __restrict__ grad) {
    const int b = blockIdx.y;
    const float scale = *scale_dev;
    const float* pb = pred + (int64_t)b * L * HW;
    for (int64_t i = blockIdx.x * (int64_t)LOSS_T + threadIdx.x; i < HW; i += (int64_t)gridDim.x * LOSS_T) {
        const int tg = target[(int64_t)b * HW + i];
        const bool sel = tg != L;
        for (int c = 0; c < L; ++c) {
            const int64_t idx = (int64_t)c * HW + i;
            float g = 0.f;
            if (sel) {
                const float z = pb[idx], t = tg == c ? 1.f : 0.f, p = sigmoidf_(z);
                const float pt = (1.f - p) * t + p * (1.f - t);
                const float at = alpha * t + (1.f - alpha) * (1.f - t);
                const float bce = bce_logits(z, t);
                const float ptg = gamma == 2.f ? pt * pt : powf(pt, gamma), ptg1 = gamma == 2.f ? pt : powf(pt, gamma - 1.f);
                const float dpt = (1.f - 2.f * t) * p * (1.f - p);
                g = scale * at * (gamma * ptg1 * dpt * bce + ptg * (p - t));
            }
            grad[(int64_t)b * L * HW + idx] = g;
        }
    }
}

struct LossLayout {        // doubles of scratch: [mask P * ns * 5][rank B * nb][depth rows * dns * 5][focal fb][seg B * nb] + floats / ints
    int mask_ns, rank_nb, depth_ns, focal_nb;
    int64_t o_mask, o_rank, o_depth, o_focal, o_seg, o_coef, total_doubles;
    LossLayout(const ph_loss_cfg& c) {
        const int64_t R = (int64_t)c.B * c.N;
        mask_ns = c.P > 0 ? (int)(2048 / c.P < 1 ? 1 : (2048 / c.P > 64 ? 64 : 2048 / c.P)) : 1;
        rank_nb = loss_grid(c.HW, 256);
        depth_ns = c.depth_rows >= 64 ? 8 : 64;
        focal_nb = loss_grid(R * c.L, 256);
        int64_t o = 0;
        o_mask = o; o += (int64_t)(c.P > 0 ? c.P : 1) * mask_ns * 5;
        o_rank = o; o += (int64_t)c.B * rank_nb;
        o_depth = o; o += (int64_t)c.depth_rows * depth_ns * 5;
        o_focal = o; o += focal_nb;
        o_seg = o; o += (int64_t)c.B * rank_nb;
        o_coef = o; o += ((int64_t)(c.P > 0 ? c.P : 1) * 3 + 4 + 1 + 1 + 1) / 2 + 4;      // floats: mask coef, depth coef, seg scale; 1 int
        total_doubles = o;
    }
};

}  // namespace

extern "C" size_t ph_train_losses_scratch_bytes(const ph_loss_cfg* c) {
    return c ? (size_t)LossLayout(*c).total_doubles * 8 + (size_t)c->B * c->HW * 4 * 2 : 0;      // + rank target, seg target (int32 [B][HW] each)
}

extern "C" int ph_train_losses(const ph_loss_cfg* cfg, const float* mask_pred, const float* cls_score, const float* depth_pred,
                               const float* seg_pred, const int32_t* pos_rows, const uint8_t* pos_u8, const int64_t* tptr,
                               const int64_t* wptr, const int32_t* dstart, const int64_t* dit_t, const int64_t* dit_w,
                               const float* dit_s, const int64_t* labels, const float* label_w, const int32_t* sstart,
                               const int64_t* sit_m, const int32_t* sit_l, float* losses, float* g_mask, float* g_cls, float* g_depth,
                               float* g_seg, void* scratch, size_t scratch_bytes, void* stream) {
    PH_CHECK_ARG(cfg && mask_pred && depth_pred && pos_u8 && tptr && wptr && dstart && losses && scratch, "null pointer");
    const ph_loss_cfg& c = *cfg;
    PH_CHECK_ARG(c.B > 0 && c.N > 0 && c.HW > 0 && c.P >= 0 && c.depth_rows > 0 && (c.P == 0 || pos_rows), "bad size");
    PH_CHECK_ARG(scratch_bytes >= ph_train_losses_scratch_bytes(cfg), "scratch too small");
    PH_CHECK_ARG((cls_score == nullptr) == (labels == nullptr) && (cls_score == nullptr) == (label_w == nullptr), "cls_score, labels, label_w go together");
    PH_CHECK_ARG((seg_pred == nullptr) == (sstart == nullptr), "seg_pred and its paint lists go together");
    const bool want_grads = g_mask != nullptr;
    PH_CHECK_ARG(!want_grads || (g_depth && (!cls_score || g_cls) && (!seg_pred || g_seg)), "gradient outputs");
    hipStream_t s = (hipStream_t)stream;
    const LossLayout lay(c);
    double* D = (double*)scratch;
    float* coef = (float*)(D + lay.o_coef);
    float *mask_coef = coef, *depth_coef = coef + (c.P > 0 ? c.P : 1) * 3, *seg_scale = depth_coef + 4;
    int* nsel = (int*)(seg_scale + 1);
    int* rank_target = (int*)(D + lay.total_doubles);
    int* seg_target = rank_target + (int64_t)c.B * c.HW;
    const int64_t R = (int64_t)c.B * c.N, HW = c.HW;
    const bool have_rank = c.has_rank && c.P > 0;
    if (seg_pred) {
        hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(1), 0, s, nsel);
        hipLaunchKernelGGL(k_seg_target_p, dim3(loss_grid(HW, 256), c.B), dim3(LOSS_T), 0, s, sstart, sit_m, sit_l, c.seg_L, HW, seg_target, nsel);
        hipLaunchKernelGGL(k_seg_focal<false>, dim3(lay.rank_nb, c.B), dim3(LOSS_T), 0, s, seg_pred, seg_target, c.seg_L, HW, c.seg_gamma,
                           c.seg_alpha, 0.f, D + lay.o_seg, (float*)nullptr);
    }
    if (c.P > 0) {
        hipLaunchKernelGGL(k_mask_sums_p, dim3(lay.mask_ns, c.P), dim3(LOSS_T), 0, s, mask_pred, pos_rows, tptr, wptr, HW, lay.mask_ns, D + lay.o_mask);
        if (have_rank) {
            hipLaunchKernelGGL(k_rank_target_p, dim3(loss_grid(HW, 1024), c.B), dim3(LOSS_T), 0, s, tptr, pos_u8, c.N, HW, c.ignore, rank_target);
            if (HW % 4 == 0 && ((uintptr_t)mask_pred & 15) == 0)
                hipLaunchKernelGGL(k_rank_loss_sum<4>, dim3(lay.rank_nb, c.B), dim3(LOSS_T), 0, s, mask_pred, rank_target, c.N, HW, c.ignore, D + lay.o_rank);
            else
                hipLaunchKernelGGL(k_rank_loss_sum<1>, dim3(lay.rank_nb, c.B), dim3(LOSS_T), 0, s, mask_pred, rank_target, c.N, HW, c.ignore, D + lay.o_rank);
        }
    }
    hipLaunchKernelGGL(k_depth_sums_csr, dim3(lay.depth_ns, c.depth_rows), dim3(LOSS_T), 0, s, depth_pred, dstart, dit_t, dit_w, dit_s, HW,
                       lay.depth_ns, c.depth_mode, D + lay.o_depth);
    if (cls_score)
        hipLaunchKernelGGL(k_focal<false>, dim3(lay.focal_nb), dim3(LOSS_T), 0, s, cls_score, labels, label_w, R, c.L, c.cls_gamma, c.cls_alpha,
                           0.f, D + lay.o_focal, (float*)nullptr);
    FinArgs fa{};
    fa.mask_part = D + lay.o_mask; fa.rank_part = D + lay.o_rank; fa.depth_part = D + lay.o_depth; fa.focal_part = D + lay.o_focal;
    fa.seg_part = D + lay.o_seg;
    fa.P = c.P; fa.mask_ns = lay.mask_ns; fa.rank_n = have_rank ? c.B * lay.rank_nb : 0; fa.depth_n = c.depth_rows * lay.depth_ns;
    fa.focal_n = cls_score ? lay.focal_nb : 0; fa.seg_n = seg_pred ? c.B * lay.rank_nb : 0;
    fa.nsel = seg_pred ? nsel : nullptr;
    fa.c = c; fa.cls = cls_score; fa.labels = labels; fa.pos_rows = pos_rows; fa.losses = losses; fa.mask_coef = mask_coef;
    fa.depth_coef = depth_coef; fa.seg_scale = seg_scale;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(LOSS_T), 0, s, fa);
    PH_CHECK_LAUNCH();
    if (want_grads) {
        if (have_rank) {
            const float scale = c.lw_rank / ((float)c.B * (float)HW);
            if (HW % 4 == 0 && (((uintptr_t)mask_pred | (uintptr_t)g_mask) & 15) == 0)
                hipLaunchKernelGGL(k_rank_loss_grad<4>, dim3(loss_grid(HW / 4, 1024), c.B), dim3(LOSS_T), 0, s, mask_pred, rank_target, c.N, HW, c.ignore, scale, g_mask);
            else
                hipLaunchKernelGGL(k_rank_loss_grad<1>, dim3(loss_grid(HW, 1024), c.B), dim3(LOSS_T), 0, s, mask_pred, rank_target, c.N, HW, c.ignore, scale, g_mask);
        } else
            hipLaunchKernelGGL(k_fill0, dim3(loss_grid(R * HW, 4096)), dim3(LOSS_T), 0, s, g_mask, R * HW);
        if (c.P > 0)
            hipLaunchKernelGGL(k_mask_grad_p, dim3(loss_grid(HW, 64), c.P), dim3(LOSS_T), 0, s, mask_pred, pos_rows, tptr, wptr, HW, mask_coef, g_mask);
        hipLaunchKernelGGL(k_depth_grad_csr, dim3(loss_grid(HW, c.depth_rows >= 64 ? 16 : 256), c.depth_rows), dim3(LOSS_T), 0, s, depth_pred, dstart,
                           dit_t, dit_w, dit_s, HW, c.depth_mode, depth_coef, g_depth);
        if (cls_score)
            hipLaunchKernelGGL(k_focal<true>, dim3(lay.focal_nb), dim3(LOSS_T), 0, s, cls_score, labels, label_w, R, c.L, c.cls_gamma, c.cls_alpha,
                               c.lw_cls / c.cls_avg, (double*)nullptr, g_cls);
        if (seg_pred)
            hipLaunchKernelGGL(k_seg_focal_grad_dev, dim3(loss_grid(HW, 256), c.B), dim3(LOSS_T), 0, s, seg_pred, seg_target, c.seg_L, HW,
                               c.seg_gamma, c.seg_alpha, seg_scale, g_seg);
        PH_CHECK_LAUNCH();
    }
    return PH_OK;
}
