// A16-A18 -- panoptic merge (polyphonic/kernel_update.py:421-535, kernel_update_head.py:593-626).
//
// The reference materialises ~N full-resolution fp32 probability and depth maps (1.3 GB at
// 1024x2048, N=153) and then runs a host loop with three device syncs per segment.  Here:
//   k_pan_activate : sigmoid / depth_act of the K selected stride-4 logit maps (gathered by query
//                    index), once, at stride 4 (K*h2*w2 values).
//   k_pan_argmax   : per OUTPUT pixel, bilinear(->batch_input_shape) . crop . bilinear(->ori_shape)
//                    of every selected map on the fly, running first-index argmax of score*prob
//                    (torch.argmax tie rule), integer histograms area[k] = #{ids == k} and
//                    orig[k] = #{prob_k >= 0.5} (LDS histogram + integer atomics: exact, order free).
//   host           : the accept loop over <= K segments on the two histograms (one D2H copy).
//   k_pan_paste    : pan = new_id[ids], depth_final = accepted ? depth_k : depth_init.
// Integer work (ids, areas, thresholds, pasted ids) is bit-exact given the same probability values;
// `from_probs` mode takes materialised full-resolution maps so that tests can prove exactly that.
// Bilinear index/weight arithmetic follows ATen (area_pixel_compute_source_index, align_corners =
// False): src = scale*(dst+0.5)-0.5 clamped at 0, scale = in/out in fp32.
#include "ph_common.h"

struct PanGeom {
    int sh, sw;     // source (stride-4) map size
    int Hb, Wb;     // batch_input_shape
    int h, w;       // img_shape (crop of the batch-resolution map)
    int Ho, Wo;     // ori_shape
};

struct Tap { int i0, i1; float l0, l1; };

__device__ __forceinline__ Tap make_tap(int dst, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = fmaxf(s, 0.f);
    Tap t;
    t.i0 = (int)s;
    if (t.i0 > in_size - 1) t.i0 = in_size - 1;
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = fminf(fmaxf(s - (float)t.i0, 0.f), 1.f);
    t.l0 = 1.f - t.l1;
    return t;
}

// ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11) with the roundings PINNED (the form the compiler
// had chosen for this expression, and what the golden id maps were checked with): each of the three sums is one fma whose addend is a
// rounded product.  Both forms of the argmax kernel and the paste kernel go through these two helpers, so that a pixel's probability
// is the same bit pattern wherever it is evaluated.
__device__ __forceinline__ float lerp_pinned(float l0, float a, float l1, float b) {
#pragma clang fp contract(off)
    const float t = l1 * b;
    return __builtin_fmaf(l0, a, t);
}
__device__ __forceinline__ float bilerp_pinned(const Tap& ty, const Tap& tx, float v00, float v01, float v10, float v11) {
    return lerp_pinned(ty.l0, lerp_pinned(tx.l0, v00, tx.l1, v01), ty.l1, lerp_pinned(tx.l0, v10, tx.l1, v11));
}

// value of the batch-resolution map at integer (y, x): bilinear of the source map
__device__ __forceinline__ float sample_mid(const float* __restrict__ src, const PanGeom& G, int y, int x) {
    const Tap ty = make_tap(y, G.sh, G.Hb), tx = make_tap(x, G.sw, G.Wb);
    const float* r0 = src + (int64_t)ty.i0 * G.sw;
    const float* r1 = src + (int64_t)ty.i1 * G.sw;
    return bilerp_pinned(ty, tx, r0[tx.i0], r0[tx.i1], r1[tx.i0], r1[tx.i1]);
}

struct OutTaps { Tap ty, tx; bool identity; };

__device__ __forceinline__ OutTaps make_out_taps(const PanGeom& G, int y, int x) {
    OutTaps o;
    o.identity = (G.h == G.Ho && G.w == G.Wo);
    if (!o.identity) { o.ty = make_tap(y, G.h, G.Ho); o.tx = make_tap(x, G.w, G.Wo); }
    else { o.ty.i0 = y; o.tx.i0 = x; }
    return o;
}

__device__ __forceinline__ float sample_out(const float* __restrict__ src, const PanGeom& G, const OutTaps& o) {
    if (o.identity) return sample_mid(src, G, o.ty.i0, o.tx.i0);
    const float a = sample_mid(src, G, o.ty.i0, o.tx.i0), b = sample_mid(src, G, o.ty.i0, o.tx.i1);
    const float c = sample_mid(src, G, o.ty.i1, o.tx.i0), d = sample_mid(src, G, o.ty.i1, o.tx.i1);
    return o.ty.l0 * (o.tx.l0 * a + o.tx.l1 * b) + o.ty.l1 * (o.tx.l0 * c + o.tx.l1 * d);
}

struct pan_h16 { uint16_t v; };     // fp16 logits (PH_OUT_F16); uint16_t = bf16 logits
__device__ __forceinline__ float ld_logit(const float* p) { return *p; }
__device__ __forceinline__ float ld_logit(const uint16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float ld_logit(const pan_h16* p) { return h2f(p->v); }

__device__ __forceinline__ float sigmoid_exact(float z) { return (float)(1.0 / (1.0 + exp(-(double)z))); }

// depth_act (funcs/depth_utils.py): mode 0 'sigmoid' -> s*(80-0.01)+0.01 ; mode 1 'monodepth'
__device__ __forceinline__ float depth_act_dev(float z, int mode) {
    const float s = sigmoid_exact(z);
    if (mode == 0) return __fadd_rn(__fmul_rn(s, 79.99f), 0.01f);   // disp * (max - min) + min, two roundings like torch
    const float min_disp = 1.0f / 80.0f, max_disp = 1.0f / 0.01f;
    return 1.0f / (min_disp + (max_disp - min_disp) * s);
}

template <typename T>
__global__ __launch_bounds__(256) void k_pan_activate(const T* __restrict__ mask_up, const T* __restrict__ depth_up,
                                                      const float* __restrict__ depth_init, const int* __restrict__ qidx,
                                                      int K, int64_t hw, int depth_mode, float* __restrict__ act_mask,
                                                      float* __restrict__ act_depth, float* __restrict__ act_depth0) {
    const int64_t total = (int64_t)(K + 1) * hw;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx / hw);
        const int64_t p = idx - (int64_t)k * hw;
        if (k < K) {
            const int64_t q = qidx[k];
            act_mask[idx] = sigmoid_exact(ld_logit(mask_up + q * hw + p));
            act_depth[idx] = depth_act_dev(ld_logit(depth_up + q * hw + p), depth_mode);
        } else {
            act_depth0[p] = depth_act_dev(depth_init[p], depth_mode);
        }
    }
}

template <bool FROM_PROBS>
__global__ __launch_bounds__(256) void k_pan_argmax(const float* __restrict__ act_mask, const float* __restrict__ scores,
                                                    int K, PanGeom G, int* __restrict__ ids, int* __restrict__ counts) {
    extern __shared__ int hist[];   // [2][K]
    for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int64_t npx = (int64_t)G.Ho * G.Wo;
    const int64_t src_hw = FROM_PROBS ? npx : (int64_t)G.sh * G.sw;
    for (int64_t px = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; px < npx; px += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(px / G.Wo), x = (int)(px - (int64_t)y * G.Wo);
        OutTaps o;
        if (!FROM_PROBS) o = make_out_taps(G, y, x);
        float best = -INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float p = FROM_PROBS ? act_mask[(int64_t)k * src_hw + px] : sample_out(act_mask + (int64_t)k * src_hw, G, o);
            if (p >= 0.5f) atomicAdd(&hist[K + k], 1);          // original_area  (kernel_update.py:508)
            const float v = scores[k] * p;                       // cur_prob_masks (:492)
            // argmax(0): first maximal index (:494); like torch.argmax a NaN counts as the maximum (the first NaN wins)
            if (v > best || (v != v && best == best)) { best = v; bi = k; }
        }
        ids[px] = bi;
        atomicAdd(&hist[bi], 1);                                 // mask_area      (:506-507)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * K; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// The shipped geometry -- stride-4 logits, no padding, ori_shape = img_shape: the second resize is the identity and the first an exact x4
// -- as blocks of 4 x 2 output pixels per thread.  The generic kernel walks K maps with four scattered loads per pixel and map (the
// texture path saturates: 0.36 ms per 1024x2048 frame and 111 maps); here a block's eight pixels share 2 source rows x 3 source columns,
// i.e. 6 loads per map for 8 pixels, the horizontal blends are shared by the two output rows, and the >= 0.5 histogram is taken with
// wave ballots (lane l of a wave keeps the count of map 64 j + l, one LDS atomic per lane and 64 maps) instead of one LDS atomic per
// pixel and map.  Taps come from the same make_tap, values from the same pinned blend: ids and both histograms are bit-identical to the
// generic kernel's (tests/test_gpu_panoptic.py runs both).
// x = 4 j + d: d in {0, 1} blends source columns (cb, cb + 1), d in {2, 3} the pair that follows it (the same pair in the first block of
// a row, where the source index is clamped at 0); y = 2 by + e: both e share one pair of source rows.
__global__ __launch_bounds__(256) void k_pan_argmax_x4(const float* __restrict__ act_mask, const float* __restrict__ scores, int K,
                                                       PanGeom G, int* __restrict__ ids, int* __restrict__ counts) {
    extern __shared__ int hist[];   // [2][K]
    for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int nbx = (G.Wo + 3) >> 2, nby = (G.Ho + 1) >> 1, lane = threadIdx.x & 63;
    const int64_t nblk = (int64_t)nbx * nby, src_hw = (int64_t)G.sh * G.sw;
    for (int64_t t0 = blockIdx.x * (int64_t)blockDim.x; t0 < nblk; t0 += (int64_t)gridDim.x * blockDim.x) {      // wave-uniform trip count
        const int64_t t = t0 + threadIdx.x;
        const bool live_t = t < nblk;
        const int64_t tc = live_t ? t : nblk - 1;
        const int by = (int)(tc / nbx), bx = (int)(tc - (int64_t)by * nbx), x0 = 4 * bx, y0 = 2 * by;
        Tap tx[4], ty[2];
        bool lx[4], ly[2];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            lx[d] = live_t && x0 + d < G.Wo;
            tx[d] = make_tap(min(x0 + d, G.Wo - 1), G.sw, G.Wb);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ly[e] = live_t && y0 + e < G.Ho;
            ty[e] = make_tap(min(y0 + e, G.Ho - 1), G.sh, G.Hb);
            if (!ly[e]) ty[e].l0 = ty[e].l1 = 0.f;            // a pixel outside the map evaluates to 0: never counted, never stored
        }
        const int cb = tx[0].i0, c1 = min(cb + 1, G.sw - 1), c2 = min(cb + 2, G.sw - 1);
        const bool a_first = tx[2].i0 == cb, b_second = tx[2].i1 == c1;       // which of the three columns the second pair is
        const int o00 = ty[0].i0 * G.sw + cb, o01 = ty[0].i0 * G.sw + c1, o02 = ty[0].i0 * G.sw + c2;
        const int o10 = ty[0].i1 * G.sw + cb, o11 = ty[0].i1 * G.sw + c1, o12 = ty[0].i1 * G.sw + c2;
        float best[2][4];
        int bi[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int d = 0; d < 4; ++d) { best[e][d] = -INFINITY; bi[e][d] = 0; }
        int cnt = 0;
        const float* m = act_mask;
        for (int k = 0; k < K; ++k, m += src_hw) {
            const float v00 = m[o00], v01 = m[o01], v02 = m[o02], v10 = m[o10], v11 = m[o11], v12 = m[o12];
            const float sk = scores[k];
            float h[2][4];
            h[0][0] = lerp_pinned(tx[0].l0, v00, tx[0].l1, v01);
            h[0][1] = lerp_pinned(tx[1].l0, v00, tx[1].l1, v01);
            h[1][0] = lerp_pinned(tx[0].l0, v10, tx[0].l1, v11);
            h[1][1] = lerp_pinned(tx[1].l0, v10, tx[1].l1, v11);
            const float a0 = a_first ? v00 : v01, b0 = b_second ? v01 : v02, a1 = a_first ? v10 : v11, b1 = b_second ? v11 : v12;
            h[0][2] = lerp_pinned(tx[2].l0, a0, tx[2].l1, b0);
            h[0][3] = lerp_pinned(tx[3].l0, a0, tx[3].l1, b0);
            h[1][2] = lerp_pinned(tx[2].l0, a1, tx[2].l1, b1);
            h[1][3] = lerp_pinned(tx[3].l0, a1, tx[3].l1, b1);
            int c = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float p = lerp_pinned(ty[e].l0, h[0][d], ty[e].l1, h[1][d]);
                    c += __popcll(__ballot(lx[d] && p >= 0.5f));               // original_area  (kernel_update.py:508)
                    const float v = sk * p;                                     // cur_prob_masks (:492)
                    if (v > best[e][d] || (v != v && best[e][d] == best[e][d])) { best[e][d] = v; bi[e][d] = k; }
                }
            cnt += lane == (k & 63) ? c : 0;
            if ((k & 63) == 63 || k == K - 1) {
                if (cnt) atomicAdd(&hist[K + (k & ~63) + lane], cnt);
                cnt = 0;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!ly[e]) continue;
            int* row = ids + (int64_t)(y0 + e) * G.Wo + x0;
            if (lx[3] && (G.Wo & 3) == 0) *(int4*)row = make_int4(bi[e][0], bi[e][1], bi[e][2], bi[e][3]);
            else {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (lx[d]) row[d] = bi[e][d];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (lx[d]) atomicAdd(&hist[bi[e][d]], 1);                       // mask_area      (:506-507)
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * K; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

__global__ void k_pan_clear(int* __restrict__ counts, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) counts[i] = 0;
}

template <bool FROM_PROBS>
__global__ __launch_bounds__(256) void k_pan_paste(const int* __restrict__ ids, const int* __restrict__ newid,
                                                   const float* __restrict__ act_depth, const float* __restrict__ act_depth0,
                                                   PanGeom G, int* __restrict__ pan, float* __restrict__ depth_basic,
                                                   float* __restrict__ depth_final) {
    const int64_t npx = (int64_t)G.Ho * G.Wo;
    const int64_t src_hw = FROM_PROBS ? npx : (int64_t)G.sh * G.sw;
    for (int64_t px = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; px < npx; px += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(px / G.Wo), x = (int)(px - (int64_t)y * G.Wo);
        OutTaps o;
        if (!FROM_PROBS) o = make_out_taps(G, y, x);
        const int k = ids[px];
        const int nid = newid[k];
        const float d0 = FROM_PROBS ? act_depth0[px] : sample_out(act_depth0, G, o);
        float df = d0;
        if (nid > 0) df = FROM_PROBS ? act_depth[(int64_t)k * src_hw + px] : sample_out(act_depth + (int64_t)k * src_hw, G, o);
        pan[px] = nid;                                           // panoptic_seg[mask] = id   (:515)
        depth_basic[px] = d0;                                    // depth_basic = depth_init  (:445-446)
        depth_final[px] = df;                                    // depth_all[mask] = total_depth[k][mask]  (:517)
    }
}

// ---- segment selection on the device (kernel_update.py:428-434, :448-459): the top max_per_img (query, thing class) pairs by score,
// descending, then the stuff queries' own-class scores, descending -- what panoptic.select_segments computes on the host from a D2H
// copy of the class scores.  With this kernel the merge up to the histograms has no host step, so a frame loop can queue it right
// behind the decode (video.VideoStreamRunner captures it into the heads' HIP graph).  Rank by counting: candidate i's position is
// the number of candidates that sort before it, ties by ascending flat index (the reference's topk / sort leave the order among
// EQUAL scores unspecified), NaN first like torch.  One workgroup per 256 thing candidates (all keys in LDS) + one for the stuff.
__device__ __forceinline__ uint32_t select_key(float v) {
    if (v != v) return 0xFFFFFFFFu;
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_pan_select(const float* __restrict__ cls, int64_t cls_stride, int L, int num_proposals,
                                                    int num_thing, int nstuff, int max_per_img, int32_t* __restrict__ q,
                                                    int32_t* __restrict__ labels, float* __restrict__ scores, int64_t out_stride) {
    extern __shared__ uint32_t keys[];
    const int b = blockIdx.y, T = num_proposals * num_thing, nb_thing = (T + 255) / 256;
    cls += (int64_t)b * cls_stride;
    q += (int64_t)b * out_stride; labels += (int64_t)b * out_stride; scores += (int64_t)b * out_stride;
    if ((int)blockIdx.x < nb_thing) {
        for (int j = threadIdx.x; j < T; j += blockDim.x) keys[j] = select_key(cls[(int64_t)(j / num_thing) * L + j % num_thing]);
        __syncthreads();
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= T) return;
        const uint32_t ki = keys[i];
        int rank = 0;
        for (int j = 0; j < T; ++j) rank += (keys[j] > ki || (keys[j] == ki && j < i)) ? 1 : 0;
        if (rank < max_per_img) {
            q[rank] = i / num_thing;
            labels[rank] = i % num_thing;
            scores[rank] = cls[(int64_t)(i / num_thing) * L + i % num_thing];
        }
    } else {
        for (int j = threadIdx.x; j < nstuff; j += blockDim.x) keys[j] = select_key(cls[(int64_t)(num_proposals + j) * L + num_thing + j]);
        __syncthreads();
        for (int i = threadIdx.x; i < nstuff; i += blockDim.x) {
            const uint32_t ki = keys[i];
            int rank = 0;
            for (int j = 0; j < nstuff; ++j) rank += (keys[j] > ki || (keys[j] == ki && j < i)) ? 1 : 0;
            q[max_per_img + rank] = num_proposals + i;
            labels[max_per_img + rank] = num_thing + i;
            scores[max_per_img + rank] = cls[(int64_t)(num_proposals + i) * L + num_thing + i];
        }
    }
}

extern "C" int ph_panoptic_select(const float* cls_scores, int64_t cls_batch_stride, int B, int N, int L, int num_proposals,
                                  int num_thing_classes, int max_per_img, int32_t* q_idx, int32_t* labels, float* scores,
                                  int64_t out_batch_stride, void* stream) {
    PH_CHECK_ARG(cls_scores && q_idx && labels && scores && B > 0, "null pointer or empty batch");
    PH_CHECK_ARG(num_proposals > 0 && num_proposals <= N && num_thing_classes > 0 && num_thing_classes <= L, "bad head geometry");
    const int T = num_proposals * num_thing_classes;
    const int nstuff = (N - num_proposals) < (L - num_thing_classes) ? (N - num_proposals) : (L - num_thing_classes);   // the diagonal
    PH_CHECK_ARG(max_per_img > 0 && max_per_img <= T && T <= 16384 && nstuff >= 0 && nstuff <= 16384, "max_per_img / candidates out of range");
    PH_CHECK_ARG(out_batch_stride >= max_per_img + nstuff, "output stride too small");
    const int nb_thing = (T + 255) / 256;
    const size_t lds = (size_t)(T > nstuff ? T : nstuff) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_pan_select, dim3(nb_thing + (nstuff > 0 ? 1 : 0), B), dim3(256), lds, (hipStream_t)stream, cls_scores,
                       cls_batch_stride, L, num_proposals, num_thing_classes, nstuff, max_per_img, q_idx, labels, scores, out_batch_stride);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

static int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int ph_panoptic_activate(const void* mask_up, const void* depth_up, int dtype, const float* depth_init_up,
                                    const int32_t* q_idx, int K, int h2, int w2, int depth_mode, float* act_mask,
                                    float* act_depth, float* act_depth0, void* stream) {
    PH_CHECK_ARG(mask_up && depth_up && depth_init_up && q_idx && act_mask && act_depth && act_depth0, "null pointer");
    PH_CHECK_ARG(K > 0 && h2 > 0 && w2 > 0 && (depth_mode == 0 || depth_mode == 1), "bad size / mode");
    PH_CHECK_ARG(dtype == PH_OUT_F32 || dtype == PH_OUT_BF16 || dtype == PH_OUT_F16, "bad dtype");
    const int64_t hw = (int64_t)h2 * w2;
    const int grid = grid_for((int64_t)(K + 1) * hw);
    if (dtype == PH_OUT_F16)
        hipLaunchKernelGGL(k_pan_activate<pan_h16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const pan_h16*)mask_up,
                           (const pan_h16*)depth_up, depth_init_up, q_idx, K, hw, depth_mode, act_mask, act_depth, act_depth0);
    else if (dtype == PH_OUT_F32)
        hipLaunchKernelGGL(k_pan_activate<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)mask_up,
                           (const float*)depth_up, depth_init_up, q_idx, K, hw, depth_mode, act_mask, act_depth, act_depth0);
    else
        hipLaunchKernelGGL(k_pan_activate<uint16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)mask_up,
                           (const uint16_t*)depth_up, depth_init_up, q_idx, K, hw, depth_mode, act_mask, act_depth, act_depth0);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

static int fill_geom(PanGeom& G, const int32_t* geom, int from_probs) {
    G.sh = geom[0]; G.sw = geom[1]; G.Hb = geom[2]; G.Wb = geom[3]; G.h = geom[4]; G.w = geom[5]; G.Ho = geom[6]; G.Wo = geom[7];
    if (G.Ho <= 0 || G.Wo <= 0) return -1;
    if (!from_probs && (G.sh <= 0 || G.sw <= 0 || G.h <= 0 || G.w <= 0 || G.h > G.Hb || G.w > G.Wb)) return -1;
    return 0;
}

extern "C" int ph_panoptic_argmax(const float* act_mask, const float* scores, int K, const int32_t* geom, int from_probs,
                                  int32_t* ids, int32_t* counts, void* stream) {
    PH_CHECK_ARG(act_mask && scores && geom && ids && counts && K > 0 && K <= 4096, "bad pointer or K");
    PanGeom G;
    PH_CHECK_ARG(fill_geom(G, geom, from_probs) == 0, "bad geometry");
    hipStream_t s = (hipStream_t)stream;
    // a kernel, not hipMemsetAsync: a captured memset node misbehaves on replay (see ph_khead1.hip)
    hipLaunchKernelGGL(k_pan_clear, dim3((2 * K + 255) / 256), dim3(256), 0, s, counts, 2 * K);
    const int grid = grid_for((int64_t)G.Ho * G.Wo);
    const size_t lds = (size_t)2 * K * sizeof(int);
    // tests: the generic kernel on the x4 geometry (read per launch on purpose: tests/test_gpu_panoptic.py switches it in-process)
    const bool generic_only = getenv("PH_PAN_GENERIC") != nullptr;
    const bool x4 = !from_probs && !generic_only && G.h == G.Ho && G.w == G.Wo && G.Hb == 4 * G.sh && G.Wb == 4 * G.sw;
    if (x4) {
        const int64_t nblk = (int64_t)((G.Wo + 3) >> 2) * ((G.Ho + 1) >> 1);
        hipLaunchKernelGGL(k_pan_argmax_x4, dim3(grid_for(nblk)), dim3(256), lds, s, act_mask, scores, K, G, ids, counts);
    } else if (from_probs) hipLaunchKernelGGL(k_pan_argmax<true>, dim3(grid), dim3(256), lds, s, act_mask, scores, K, G, ids, counts);
    else hipLaunchKernelGGL(k_pan_argmax<false>, dim3(grid), dim3(256), lds, s, act_mask, scores, K, G, ids, counts);
    PH_CHECK_LAUNCH();
    return PH_OK;
}

extern "C" int ph_panoptic_paste(const int32_t* ids, const int32_t* newid, const float* act_depth, const float* act_depth0,
                                 const int32_t* geom, int from_probs, int32_t* pan, float* depth_basic, float* depth_final,
                                 void* stream) {
    PH_CHECK_ARG(ids && newid && act_depth && act_depth0 && geom && pan && depth_basic && depth_final, "null pointer");
    PanGeom G;
    PH_CHECK_ARG(fill_geom(G, geom, from_probs) == 0, "bad geometry");
    const int grid = grid_for((int64_t)G.Ho * G.Wo);
    hipStream_t s = (hipStream_t)stream;
    if (from_probs) hipLaunchKernelGGL(k_pan_paste<true>, dim3(grid), dim3(256), 0, s, ids, newid, act_depth, act_depth0, G, pan, depth_basic, depth_final);
    else hipLaunchKernelGGL(k_pan_paste<false>, dim3(grid), dim3(256), 0, s, ids, newid, act_depth, act_depth0, G, pan, depth_basic, depth_final);
    PH_CHECK_LAUNCH();
    return PH_OK;
}
