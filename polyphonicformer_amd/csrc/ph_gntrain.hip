// libpolyhead, training side: GroupNorm + ReLU of a ConvModule (mmcv: conv -> GN -> ReLU) on fp32 NCHW
// maps, forward AND backward -- the three 1x1 towers of KernelHead (kernel_head.py:250-278) and the 3x3 towers / output convs of
// SemanticFPNWrapper (funcs/semantic_fpn.py:75-178) in training mode.  Replaces ATen's group_norm / threshold kernels and their
// autograd (RowwiseMoments + three elementwise passes forward, five backward).
//
// forward : pass 1 sums x and x^2 of every (image, group) in fp64 over fixed pixel slices (partial records, fixed order);
//           pass 2 turns them into mean / rstd (fp64: E[x^2] - mean^2 is safe there), writes out = relu(gamma xhat + beta),
//           optionally out_sum = out + add (KernelHead's x_feats = sem + loc, kernel_head.py:303) and the statistics.
// backward: dy = dyA (+ dyB: a second gradient contribution, no ATen add over the map) masked by out > 0 (recomputed);
//           pass 1: per (image, channel) sums of dy and dy * xhat (fp64 partials) -> d gamma, d beta and the group sums;
//           pass 2: dx = rstd * (gamma dy - mean_g(gamma dy) - xhat mean_g(gamma dy xhat)).
// Both are HBM streaming kernels (16-byte accesses where HW % 4 == 0), one workgroup per (slice, row).
#include "ph_common.h"

namespace {

constexpr int GT_T = 256;

__device__ __forceinline__ void block_sum2(double& a, double& b, double* lds /* [8] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        lds[wave * 2] = a;
        lds[wave * 2 + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (lds[0] + lds[2]) + (lds[4] + lds[6]);
        b = (lds[1] + lds[3]) + (lds[5] + lds[7]);
    }
}

// ---- forward pass 1: partial[(b * G + g)][split] = (sum x, sum x^2) over the split's share of the group's cpg * HW contiguous floats
template <bool VEC>
__global__ __launch_bounds__(GT_T) void k_gnt_stats(const float* __restrict__ y, int64_t n /* cpg * HW */, int nsplit, double* __restrict__ partial) {
    __shared__ double lds[8];
    const int64_t row = blockIdx.y;
    const int sp = blockIdx.x;
    const float* p = y + row * n;
    int64_t i0 = n * sp / nsplit, i1 = n * (sp + 1) / nsplit;
    double s = 0, q = 0;
    if (VEC) {
        i0 &= ~(int64_t)3;
        i1 = sp + 1 == nsplit ? n : (i1 & ~(int64_t)3);
        for (int64_t i = i0 + threadIdx.x * 4; i < i1; i += GT_T * 4) {
            const float4 v = *(const float4*)(p + i);
            s += (double)((v.x + v.y) + (v.z + v.w));
            q += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += GT_T) {
            const float v = p[i];
            s += (double)v;
            q += (double)(v * v);
        }
    }
    block_sum2(s, q, lds);
    if (threadIdx.x == 0) {
        partial[(row * nsplit + sp) * 2] = s;
        partial[(row * nsplit + sp) * 2 + 1] = q;
    }
}

__device__ __forceinline__ void group_stat(const double* __restrict__ partial, int64_t bg, int nsplit, int64_t n, float eps, float& mean, float& rstd) {
    double s = 0, q = 0;
    for (int k = 0; k < nsplit; ++k) {
        s += partial[(bg * nsplit + k) * 2];
        q += partial[(bg * nsplit + k) * 2 + 1];
    }
    const double m = s / (double)n;
    double var = q / (double)n - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- forward pass 2: grid (slices, B * C)
template <bool VEC>
__global__ __launch_bounds__(GT_T) void k_gnt_apply(const float* __restrict__ y, const double* __restrict__ partial, int nsplit,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, int C, int cpg,
                                                    int64_t HW, float eps, const float* __restrict__ add, float* __restrict__ out,
                                                    float* __restrict__ out_sum, float* __restrict__ stats) {
    const int64_t bc = blockIdx.y;
    const int c = (int)(bc % C);
    const int64_t bg = bc / cpg;             // = b * G + c / cpg because C = G * cpg
    float mean, rstd;
    group_stat(partial, bg, nsplit, (int64_t)cpg * HW, eps, mean, rstd);
    if (stats && blockIdx.x == 0 && threadIdx.x == 0 && c % cpg == 0) {
        stats[bg * 2] = mean;
        stats[bg * 2 + 1] = rstd;
    }
    const float sc = gamma[c] * rstd, sh = beta[c] - mean * sc;
    const float* p = y + bc * HW;
    const float* a = add ? add + bc * HW : nullptr;
    float* o = out ? out + bc * HW : nullptr;            // null: only the sum is wanted (a level of the neck's tower sum)
    float* o2 = out_sum ? out_sum + bc * HW : nullptr;
    if (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * GT_T + threadIdx.x) * 4; i < HW; i += (int64_t)gridDim.x * GT_T * 4) {
            const float4 v = *(const float4*)(p + i);
            const float4 r = make_float4(fmaxf(v.x * sc + sh, 0.f), fmaxf(v.y * sc + sh, 0.f), fmaxf(v.z * sc + sh, 0.f), fmaxf(v.w * sc + sh, 0.f));
            if (o) *(float4*)(o + i) = r;
            if (o2) {
                const float4 w = *(const float4*)(a + i);
                *(float4*)(o2 + i) = make_float4(r.x + w.x, r.y + w.y, r.z + w.z, r.w + w.w);
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * GT_T + threadIdx.x; i < HW; i += (int64_t)gridDim.x * GT_T) {
            const float r = fmaxf(p[i] * sc + sh, 0.f);
            if (o) o[i] = r;
            if (o2) o2[i] = r + a[i];
        }
    }
}

// ---- backward pass 1: partial[(b * C + c)][split] = (sum dy, sum dy xhat) over the split's pixels, dy masked by out > 0
template <bool VEC>
__global__ __launch_bounds__(GT_T) void k_gnt_bwd_sums(const float* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int C, int cpg, int64_t HW,
                                                       const float* __restrict__ dyA, const float* __restrict__ dyB, int nsplit,
                                                       double* __restrict__ partial) {
    __shared__ double lds[8];
    const int64_t bc = blockIdx.y;
    const int c = (int)(bc % C), sp = blockIdx.x;
    const int64_t bg = bc / cpg;
    const float mean = stats[bg * 2], rstd = stats[bg * 2 + 1];
    const float sc = gamma[c] * rstd, sh = beta[c] - mean * sc;
    const float* p = y + bc * HW;
    const float* ga = dyA + bc * HW;
    const float* gb = dyB ? dyB + bc * HW : nullptr;
    int64_t i0 = HW * sp / nsplit, i1 = HW * (sp + 1) / nsplit;
    double s = 0, q = 0;
    if (VEC) {
        i0 &= ~(int64_t)3;
        i1 = sp + 1 == nsplit ? HW : (i1 & ~(int64_t)3);
        for (int64_t i = i0 + threadIdx.x * 4; i < i1; i += GT_T * 4) {
            const float4 v = *(const float4*)(p + i);
            float4 d = *(const float4*)(ga + i);
            if (gb) {
                const float4 e = *(const float4*)(gb + i);
                d = make_float4(d.x + e.x, d.y + e.y, d.z + e.z, d.w + e.w);
            }
            const float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (vv[k] * sc + sh > 0.f) {
                    s += (double)dd[k];
                    q += (double)(dd[k] * ((vv[k] - mean) * rstd));
                }
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += GT_T) {
            const float v = p[i];
            if (v * sc + sh > 0.f) {
                const float d = ga[i] + (gb ? gb[i] : 0.f);
                s += (double)d;
                q += (double)(d * ((v - mean) * rstd));
            }
        }
    }
    block_sum2(s, q, lds);
    if (threadIdx.x == 0) {
        partial[(bc * nsplit + sp) * 2] = s;
        partial[(bc * nsplit + sp) * 2 + 1] = q;
    }
}

// d gamma[c] = sum_b sum dy xhat, d beta[c] = sum_b sum dy  (C threads)
__global__ __launch_bounds__(GT_T) void k_gnt_bwd_params(const double* __restrict__ partial, int B, int C, int nsplit, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta) {
    const int c = blockIdx.x * GT_T + threadIdx.x;
    if (c >= C) return;
    double s = 0, q = 0;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < nsplit; ++k) {
            s += partial[(((int64_t)b * C + c) * nsplit + k) * 2];
            q += partial[(((int64_t)b * C + c) * nsplit + k) * 2 + 1];
        }
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

// ---- backward pass 2: grid (slices, B * C)
template <bool VEC>
__global__ __launch_bounds__(GT_T) void k_gnt_bwd_apply(const float* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int C, int cpg, int64_t HW,
                                                        const float* __restrict__ dyA, const float* __restrict__ dyB, int nsplit,
                                                        const double* __restrict__ partial, float* __restrict__ dx) {
    const int64_t bc = blockIdx.y;
    const int c = (int)(bc % C);
    const int64_t bg = bc / cpg, b = bc / C;
    const float mean = stats[bg * 2], rstd = stats[bg * 2 + 1];
    // group sums of gamma dy and gamma dy xhat over the group's channels (every block of the group: the same order)
    double s1 = 0, s2 = 0;
    const int c0 = c / cpg * cpg;
    for (int cc = c0; cc < c0 + cpg; ++cc) {
        double a = 0, q = 0;
        for (int k = 0; k < nsplit; ++k) {
            a += partial[((b * C + cc) * nsplit + k) * 2];
            q += partial[((b * C + cc) * nsplit + k) * 2 + 1];
        }
        s1 += (double)gamma[cc] * a;
        s2 += (double)gamma[cc] * q;
    }
    const double n = (double)cpg * (double)HW;
    const float m1 = (float)(s1 / n), m2 = (float)(s2 / n);
    const float g = gamma[c], sc = g * rstd, sh = beta[c] - mean * sc;
    const float* p = y + bc * HW;
    const float* ga = dyA + bc * HW;
    const float* gb = dyB ? dyB + bc * HW : nullptr;
    float* o = dx + bc * HW;
    auto one = [&](float v, float d) {
        const float xh = (v - mean) * rstd;
        const float dm = v * sc + sh > 0.f ? d * g : 0.f;
        return rstd * (dm - m1 - xh * m2);
    };
    if (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * GT_T + threadIdx.x) * 4; i < HW; i += (int64_t)gridDim.x * GT_T * 4) {
            const float4 v = *(const float4*)(p + i);
            float4 d = *(const float4*)(ga + i);
            if (gb) {
                const float4 e = *(const float4*)(gb + i);
                d = make_float4(d.x + e.x, d.y + e.y, d.z + e.z, d.w + e.w);
            }
            *(float4*)(o + i) = make_float4(one(v.x, d.x), one(v.y, d.y), one(v.z, d.z), one(v.w, d.w));
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * GT_T + threadIdx.x; i < HW; i += (int64_t)gridDim.x * GT_T)
            o[i] = one(p[i], ga[i] + (gb ? gb[i] : 0.f));
    }
}

int slices(int64_t n) {
    int64_t g = (n + GT_T * 4 * 4 - 1) / (GT_T * 4 * 4);          // >= 4 float4 per thread
    return (int)(g < 1 ? 1 : (g > 64 ? 64 : g));
}

}  // namespace

extern "C" int ph_gn_train_nsplit(int64_t HW, int cpg) { return slices(HW * cpg) > 16 ? 16 : slices(HW * cpg); }

extern "C" int ph_gn_train_fwd(const float* y, const float* gamma, const float* beta, int groups, float eps, const float* add, float* out,
                               float* out_sum, float* stats, double* partial, int B, int C, int64_t HW, void* stream) {
    PH_CHECK_ARG(y && gamma && beta && (out || out_sum) && stats && partial && B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "bad pointer or size");
    PH_CHECK_ARG((add == nullptr) == (out_sum == nullptr), "add and out_sum go together");
    const int cpg = C / groups, ns = ph_gn_train_nsplit(HW, cpg);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = HW % 4 == 0 && (((uintptr_t)y | (uintptr_t)out | (uintptr_t)add | (uintptr_t)out_sum) & 15) == 0;
    if (vec) hipLaunchKernelGGL(k_gnt_stats<true>, dim3(ns, B * groups), dim3(GT_T), 0, s, y, (int64_t)cpg * HW, ns, partial);
    else hipLaunchKernelGGL(k_gnt_stats<false>, dim3(ns, B * groups), dim3(GT_T), 0, s, y, (int64_t)cpg * HW, ns, partial);
    PH_CHECK_LAUNCH();
    const dim3 grid(slices(HW), B * C);
    if (vec) hipLaunchKernelGGL(k_gnt_apply<true>, grid, dim3(GT_T), 0, s, y, partial, ns, gamma, beta, C, cpg, HW, eps, add, out, out_sum, stats);
    else hipLaunchKernelGGL(k_gnt_apply<false>, grid, dim3(GT_T), 0, s, y, partial, ns, gamma, beta, C, cpg, HW, eps, add, out, out_sum, stats);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_gn_train_bwd_nsplit(int64_t HW) { return slices(HW) > 8 ? 8 : slices(HW); }

extern "C" int ph_gn_train_bwd(const float* y, const float* stats, const float* gamma, const float* beta, int groups, const float* dyA,
                               const float* dyB, float* dx, float* dgamma, float* dbeta, double* partial, int B, int C, int64_t HW,
                               void* stream) {
    PH_CHECK_ARG(y && stats && gamma && beta && dyA && dx && dgamma && dbeta && partial && B > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0,
                 "bad pointer or size");
    const int cpg = C / groups, ns = ph_gn_train_bwd_nsplit(HW);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = HW % 4 == 0 && (((uintptr_t)y | (uintptr_t)dyA | (uintptr_t)dyB | (uintptr_t)dx) & 15) == 0;
    if (vec) hipLaunchKernelGGL(k_gnt_bwd_sums<true>, dim3(ns, B * C), dim3(GT_T), 0, s, y, stats, gamma, beta, C, cpg, HW, dyA, dyB, ns, partial);
    else hipLaunchKernelGGL(k_gnt_bwd_sums<false>, dim3(ns, B * C), dim3(GT_T), 0, s, y, stats, gamma, beta, C, cpg, HW, dyA, dyB, ns, partial);
    PH_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_gnt_bwd_params, dim3((C + GT_T - 1) / GT_T), dim3(GT_T), 0, s, partial, B, C, ns, dgamma, dbeta);
    PH_CHECK_LAUNCH();
    const dim3 grid(slices(HW), B * C);
    if (vec) hipLaunchKernelGGL(k_gnt_bwd_apply<true>, grid, dim3(GT_T), 0, s, y, stats, gamma, beta, C, cpg, HW, dyA, dyB, ns, partial, dx);
    else hipLaunchKernelGGL(k_gnt_bwd_apply<false>, grid, dim3(GT_T), 0, s, y, stats, gamma, beta, C, cpg, HW, dyA, dyB, ns, partial, dx);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
