// libpolyhead: the quasi-dense embedding tracker of the video path as a NATIVE object (round 5).
// Semantics: polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py:47-207 -- `match` (de-duplication by IoU, affinity of
// the detections' embeddings against the memory, greedy one-to-one assignment in score order, new tracks) and `update_memo`
// (exponential-moving-average embeddings of matched tracklets, backdrops, expiry) -- with the integer ids of the reference
// (tests/golden/tracker.npz).
//
// Why native: round 4's tracker kept this bookkeeping in numpy (~60 small array operations, six pinned index uploads, ~10 torch
// gathers / concatenations per frame: 0.41 ms per frame on the host).  In the sharded video mode every rank replays the tracker over
// ALL frames of a step, so that cost caps a node at 1 / 0.41 ms = 2.4 k frames/s whatever the GPUs do.
// Here the boxes / labels / ids / ages are plain C++ arrays, the embeddings never leave the device: a fixed POOL of 256-float rows
// (tracklets and backdrops own slots), and a frame is
//     one small upload (row / slot tables) -> gather + `ph_track_affinity` (4 launches) -> one download of the [n x m] scores
//     -> the greedy walk in C++ -> one small upload -> one update launch (EMA / copy into pool slots).
// fp32 operations are written one per statement (no contraction) in the order the reference's tensor expressions evaluate them.
#include <algorithm>
#include <chrono>
#include <vector>

#include "ph_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int TE = 256;      // embedding width

struct Backdrop {
    std::vector<float> box;          // [k][5]
    std::vector<int64_t> lab;
    std::vector<int> slot;
};

}  // namespace

struct ph_tracker {
    ph_tracker_cfg c;
    float* pool;                     // device [capacity][256]
    int capacity;
    std::vector<int> free_slots;
    // tracklets in creation order
    std::vector<int64_t> ids, lab, seen;
    std::vector<float> box;          // [rows][5]
    std::vector<int> slot;
    std::vector<Backdrop> backdrops; // newest first
    int64_t num_tracklets;
    // staging
    int32_t* h_tab;                  // pinned host
    float* h_score;                  // pinned host
    int32_t* d_tab;                  // device
    float *d_det, *d_memo, *d_score; // device dense scratch
    void* d_ws;
    size_t ws_bytes;
    int max_n, max_m;
    double tacc[6];                  // seconds spent in the phases of `match` (ph_tracker_debug_times)
    hipEvent_t uploaded;             // the last upload from the pinned table has left the host buffer
    bool pending;
    bool poisoned;                   // a device error struck after the bookkeeping of a frame had been committed: ids would be undefined from here on
};

namespace {

__global__ __launch_bounds__(256) void k_trk_gather(const float* __restrict__ det_src, const int* __restrict__ det_rows, int n,
                                                    const float* __restrict__ pool, const int* __restrict__ slots, int m,
                                                    float* __restrict__ det, float* __restrict__ memo) {
    const int r = blockIdx.x;
    const float* src = r < n ? det_src + (int64_t)det_rows[r] * TE : pool + (int64_t)slots[r - n] * TE;
    float* dst = r < n ? det + (int64_t)r * TE : memo + (int64_t)(r - n) * TE;
    dst[threadIdx.x] = src[threadIdx.x];
}

// action table per kept detection: (slot, mode) -- mode 1: pool[slot] = (1 - mom) pool[slot] + mom det (two rounded products, one
// rounded sum, like the reference's tensor expression), mode 2: pool[slot] = det
__global__ __launch_bounds__(256) void k_trk_update(const float* __restrict__ det, const int* __restrict__ act, int n, float one_minus, float mom,
                                                    float* __restrict__ pool) {
    const int i = blockIdx.x;
    const int slot = act[2 * i], mode = act[2 * i + 1];
    if (mode == 0) return;
    const float v = det[(int64_t)i * TE + threadIdx.x];
    float* p = pool + (int64_t)slot * TE + threadIdx.x;
    *p = mode == 2 ? v : __fadd_rn(__fmul_rn(one_minus, *p), __fmul_rn(mom, v));
}

inline float iou1(const float* a, const float* b) {
    const float x1 = std::max(a[0], b[0]), y1 = std::max(a[1], b[1]), x2 = std::min(a[2], b[2]), y2 = std::min(a[3], b[3]);
    const float w = std::max(x2 - x1, 0.f), h = std::max(y2 - y1, 0.f);
    const float inter = w * h;
    const float aw = a[2] - a[0], ah = a[3] - a[1], bw = b[2] - b[0], bh = b[3] - b[1];
    const float a1 = aw * ah, a2 = bw * bh;
    const float s = a1 + a2;
    const float u = std::max(s - inter, 1e-6f);
    return inter / u;
}

}  // namespace

extern "C" size_t ph_tracker_device_bytes(int capacity, int max_dets) {
    const size_t m = capacity, n = max_dets;
    return (m * TE + n * TE + m * TE + n * m) * sizeof(float) + ph_track_affinity_workspace_bytes((int)n, (int)m) + (4 * n + 2 * m + 64) * sizeof(int32_t);
}

extern "C" ph_tracker* ph_tracker_create(const ph_tracker_cfg* cfg, void* device_mem, size_t device_bytes, int capacity, int max_dets) {
    if (!cfg || !device_mem || capacity < 16 || capacity > 4096 || max_dets < 1 || max_dets > 128 ||
        device_bytes < ph_tracker_device_bytes(capacity, max_dets) || cfg->metric < 0 || cfg->metric > 2) {
        ph_set_error("ph_tracker_create: bad configuration (capacity 16..4096 rows, up to 128 detections per frame) or device memory too small");
        return nullptr;
    }
    ph_tracker* t = new ph_tracker();
    t->h_tab = nullptr;
    t->h_score = nullptr;
    t->c = *cfg;
    t->capacity = capacity;
    t->max_n = max_dets;
    t->max_m = capacity;
    float* f = (float*)device_mem;
    t->pool = f; f += (size_t)capacity * TE;
    t->d_det = f; f += (size_t)max_dets * TE;
    t->d_memo = f; f += (size_t)capacity * TE;
    t->d_score = f; f += (size_t)max_dets * capacity;
    t->d_ws = f;
    t->ws_bytes = ph_track_affinity_workspace_bytes(max_dets, capacity);
    t->d_tab = (int32_t*)((char*)t->d_ws + t->ws_bytes);
    const size_t tab = (size_t)4 * max_dets + 2 * capacity + 64;
    if (hipHostMalloc((void**)&t->h_tab, tab * sizeof(int32_t)) != hipSuccess ||
        hipHostMalloc((void**)&t->h_score, (size_t)max_dets * capacity * sizeof(float)) != hipSuccess) {
        ph_set_error("ph_tracker_create: pinned host allocation failed");
        if (t->h_tab) (void)hipHostFree(t->h_tab);
        if (t->h_score) (void)hipHostFree(t->h_score);
        delete t;
        return nullptr;
    }
    if (hipEventCreateWithFlags(&t->uploaded, hipEventDisableTiming) != hipSuccess) {
        ph_set_error("ph_tracker_create: event creation failed");
        (void)hipHostFree(t->h_tab);
        (void)hipHostFree(t->h_score);
        delete t;
        return nullptr;
    }
    t->pending = false;
    t->poisoned = false;
    for (double& v : t->tacc) v = 0;
    t->num_tracklets = 0;
    for (int s = capacity - 1; s >= 0; --s) t->free_slots.push_back(s);
    return t;
}

extern "C" void ph_tracker_destroy(ph_tracker* t) {
    if (!t) return;
    (void)hipEventDestroy(t->uploaded);
    (void)hipHostFree(t->h_tab);
    (void)hipHostFree(t->h_score);
    delete t;
}

extern "C" void ph_tracker_reset(ph_tracker* t) {
    if (!t) return;
    t->ids.clear(); t->lab.clear(); t->seen.clear(); t->box.clear(); t->slot.clear(); t->backdrops.clear();
    t->free_slots.clear();
    for (int s = t->capacity - 1; s >= 0; --s) t->free_slots.push_back(s);
    t->num_tracklets = 0;
    t->poisoned = false;
}

// accumulated host seconds per phase: [0] wait for the previous frame's upload, [1] sort + de-duplication + tables, [2] launches up to
// the score download, [3] the synchronising wait, [4] greedy walk + bookkeeping, [5] update launch
extern "C" void ph_tracker_debug_times(const ph_tracker* t, double* out6) {
    if (t && out6) for (int i = 0; i < 6; ++i) out6[i] = t->tacc[i];
}

extern "C" int64_t ph_tracker_num_tracklets(const ph_tracker* t) { return t ? t->num_tracklets : -1; }
extern "C" int ph_tracker_rows(const ph_tracker* t) { return t ? (int)t->ids.size() : -1; }

// boxes [n][5] (x1, y1, x2, y2, score), labels [n] on the HOST; embeds [n][256] on the DEVICE.  Writes the kept detections in
// descending-score order: kept_out [k] (indices into the input), ids_out [k] (>= 0 track id, -1 unmatched, -2 suppressed).
// Returns k, or a negative error code.
extern "C" int ph_tracker_match(ph_tracker* t, const float* boxes, const int64_t* labels, const float* embeds_dev, int n, int64_t frame_id,
                                int32_t* kept_out, int64_t* ids_out, void* stream) {
    PH_CHECK_ARG(t && (n == 0 || (boxes && labels && embeds_dev)) && kept_out && ids_out && n >= 0, "bad pointer or size");
    PH_CHECK_ARG(n <= t->max_n, "more detections than the tracker was created for");
    if (t->poisoned) {
        ph_set_error("ph_tracker_match: an earlier frame failed on the device after its bookkeeping was committed; reset or recreate the tracker");
        return PH_ELAUNCH;
    }
    hipStream_t s = (hipStream_t)stream;
    const ph_tracker_cfg& c = t->c;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto t0 = now();
    if (t->pending) { (void)hipEventSynchronize(t->uploaded); t->pending = false; }      // the previous frame's last upload (long done)
    auto t1 = now();
    t->tacc[0] += secs(t0, t1);
    // ---- descending score (stable: ascending index among equal scores, torch's CPU order), de-duplication (:147-155)
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return boxes[a * 5 + 4] > boxes[b * 5 + 4]; });
    std::vector<int> kept;
    std::vector<char> keep(n, 1);
    for (int i = 0; i < n; ++i) {
        const float* bi = boxes + order[i] * 5;
        const float thr = bi[4] < c.obj_score_thr ? c.nms_backdrop_iou_thr : c.nms_class_iou_thr;
        for (int j = 0; j < i; ++j)
            if (iou1(bi, boxes + order[j] * 5) > thr) { keep[i] = 0; break; }       // ANY higher-scored detection, kept or not
        if (keep[i]) kept.push_back(order[i]);
    }
    const int k = (int)kept.size();
    std::vector<int64_t> ids(k, -1);
    // ---- memory columns: tracklets in creation order, then the backdrops newest first (id -1)
    std::vector<int64_t> memo_ids, memo_lab;
    std::vector<int> memo_slot;
    for (size_t r = 0; r < t->ids.size(); ++r) { memo_ids.push_back(t->ids[r]); memo_lab.push_back(t->lab[r]); memo_slot.push_back(t->slot[r]); }
    const bool table_empty = t->ids.empty();                       // `empty` looks at the tracklets only (:39-41)
    for (const Backdrop& b : t->backdrops)
        for (size_t r = 0; r < b.lab.size(); ++r) { memo_ids.push_back(-1); memo_lab.push_back(b.lab[r]); memo_slot.push_back(b.slot[r]); }
    const int m = (int)memo_ids.size();
    // upload the frame's tables: [kept rows k][det labels k][memo slots m][memo labels m]
    int32_t* tab = t->h_tab;
    for (int i = 0; i < k; ++i) { tab[i] = kept[i]; tab[k + i] = (int32_t)labels[kept[i]]; }
    for (int j = 0; j < m; ++j) { tab[2 * k + j] = memo_slot[j]; tab[2 * k + m + j] = (int32_t)memo_lab[j]; }
    const bool need_aff = k > 0 && !table_empty && m > 0;
    auto t2 = now();
    t->tacc[1] += secs(t1, t2);
    auto t3 = t2, t4 = t2;
    if (k > 0) {
        if (hipMemcpyAsync(t->d_tab, tab, (size_t)(2 * k + 2 * m) * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) {
            ph_set_error("ph_tracker_match: table upload failed");
            return PH_ELAUNCH;
        }
        hipLaunchKernelGGL(k_trk_gather, dim3(k + (need_aff ? m : 0)), dim3(TE), 0, s, embeds_dev, t->d_tab, k, t->pool, t->d_tab + 2 * k,
                           need_aff ? m : 0, t->d_det, t->d_memo);
    }
    if (need_aff) {
        const int rc = ph_track_affinity(t->d_det, t->d_tab + k, t->d_memo, t->d_tab + 2 * k + m, k, m, c.metric, c.with_cats, t->d_score, t->d_ws,
                                         t->ws_bytes, stream);
        if (rc != PH_OK) return rc;
        if (hipMemcpyAsync(t->h_score, t->d_score, (size_t)k * m * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) {
            ph_set_error("ph_tracker_match: score download failed");
            return PH_ELAUNCH;
        }
        t3 = now();
        if (hipStreamSynchronize(s) != hipSuccess) {
            ph_set_error("ph_tracker_match: score download failed");
            return PH_ELAUNCH;
        }
        t4 = now();
        t->tacc[2] += secs(t2, t3);
        t->tacc[3] += secs(t3, t4);
        // ---- greedy, in detection (score) order: best still-free column (:183-197)
        std::vector<char> taken(m, 0);
        for (int i = 0; i < k; ++i) {
            const float* row = t->h_score + (size_t)i * m;
            int best = 0;
            float conf = taken[0] ? 0.f : row[0];
            for (int j = 1; j < m; ++j) {
                const float v = taken[j] ? 0.f : row[j];
                if (v > conf) { conf = v; best = j; }              // first maximal column
            }
            if (!(conf > c.match_score_thr) || memo_ids[best] < 0) continue;
            if (boxes[kept[i] * 5 + 4] > c.obj_score_thr) {
                ids[i] = memo_ids[best];
                taken[best] = 1;
            } else if (conf > c.nms_conf_thr)
                ids[i] = -2;
        }
    }
    // ---- new tracks (:198-205)
    int64_t born = 0;                                           // committed to num_tracklets once the frame's slots are known to exist
    for (int i = 0; i < k; ++i)
        if (ids[i] == -1 && boxes[kept[i] * 5 + 4] > c.init_score_thr) ids[i] = t->num_tracklets + born++;
    // ---- every pool slot this frame takes is reserved BEFORE the first mutation: a frame that does not fit fails with the tracker
    // exactly as the previous frame left it
    {
        size_t need = 0;
        for (int i = 0; i < k; ++i) {
            const float* bi = boxes + kept[i] * 5;
            if (ids[i] > -1) {
                bool found = false;
                for (size_t r = 0; r < t->ids.size() && !found; ++r) found = t->ids[r] == ids[i];
                need += found ? 0 : 1;
            } else if (ids[i] == -1 && c.memo_backdrop_frames > 0) {
                bool covered = false;
                for (int j = 0; j < i && !covered; ++j) covered = iou1(bi, boxes + kept[j] * 5) > c.nms_backdrop_iou_thr;
                need += covered ? 0 : 1;
            }
        }
        if (need > t->free_slots.size()) {
            ph_set_error("ph_tracker_match: embedding pool exhausted (%zu slots needed, %zu free; raise the capacity)", need, t->free_slots.size());
            return PH_EWORKSPACE;
        }
    }
    t->num_tracklets += born;
    // ---- update_memo (:47-102): tracked detections refresh / append rows; unmatched, uncovered ones become this frame's backdrops
    const int act_off = 2 * t->max_n + 2 * t->capacity;      // its own region of the pinned / device tables: the first upload may still be in flight
    int32_t* act = t->h_tab + act_off;    // [k][2] (slot, mode)
    auto take_slot = [&]() -> int {
        if (t->free_slots.empty()) return -1;
        const int sl = t->free_slots.back();
        t->free_slots.pop_back();
        return sl;
    };
    Backdrop bd;
    for (int i = 0; i < k; ++i) {
        act[2 * i] = 0; act[2 * i + 1] = 0;
        const float* bi = boxes + kept[i] * 5;
        if (ids[i] > -1) {
            size_t r = 0;
            for (; r < t->ids.size(); ++r)
                if (t->ids[r] == ids[i]) break;
            if (r < t->ids.size()) {
                act[2 * i] = t->slot[r]; act[2 * i + 1] = 1;
                std::copy(bi, bi + 5, t->box.begin() + r * 5);
                t->lab[r] = labels[kept[i]];
                t->seen[r] = frame_id;
            } else {
                const int sl = take_slot();                 // reserved above
                act[2 * i] = sl; act[2 * i + 1] = 2;
                t->ids.push_back(ids[i]); t->lab.push_back(labels[kept[i]]); t->seen.push_back(frame_id); t->slot.push_back(sl);
                t->box.insert(t->box.end(), bi, bi + 5);
            }
        } else if (ids[i] == -1) {
            bool covered = false;
            for (int j = 0; j < i && !covered; ++j) covered = iou1(bi, boxes + kept[j] * 5) > c.nms_backdrop_iou_thr;
            if (!covered && c.memo_backdrop_frames > 0) {
                const int sl = take_slot();                 // reserved above
                act[2 * i] = sl; act[2 * i + 1] = 2;
                bd.box.insert(bd.box.end(), bi, bi + 5); bd.lab.push_back(labels[kept[i]]); bd.slot.push_back(sl);
            }
        }
    }
    auto t5 = now();
    t->tacc[4] += secs(t4, t5);
    if (k > 0) {
        if (hipMemcpyAsync(t->d_tab + act_off, act, (size_t)2 * k * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) {
            ph_set_error("ph_tracker_match: action upload failed");
            t->poisoned = true;                              // the tables above already describe this frame
            return PH_ELAUNCH;
        }
        hipLaunchKernelGGL(k_trk_update, dim3(k), dim3(TE), 0, s, t->d_det, t->d_tab + act_off, k, c.one_minus_momentum, c.memo_momentum, t->pool);
        (void)hipEventRecord(t->uploaded, s);
        t->pending = true;
    }
    // backdrops: newest first, `memo_backdrop_frames` frames of them
    t->backdrops.insert(t->backdrops.begin(), bd);
    while ((int)t->backdrops.size() > c.memo_backdrop_frames) {
        for (int sl : t->backdrops.back().slot) t->free_slots.push_back(sl);
        t->backdrops.pop_back();
    }
    // expiry: tracklets unseen for memo_tracklet_frames frames
    size_t w = 0;
    for (size_t r = 0; r < t->ids.size(); ++r) {
        if (frame_id - t->seen[r] < c.memo_tracklet_frames) {
            if (w != r) {
                t->ids[w] = t->ids[r]; t->lab[w] = t->lab[r]; t->seen[w] = t->seen[r]; t->slot[w] = t->slot[r];
                std::copy(t->box.begin() + r * 5, t->box.begin() + r * 5 + 5, t->box.begin() + w * 5);
            }
            ++w;
        } else
            t->free_slots.push_back(t->slot[r]);
    }
    t->ids.resize(w); t->lab.resize(w); t->seen.resize(w); t->slot.resize(w); t->box.resize(w * 5);
    for (int i = 0; i < k; ++i) { kept_out[i] = kept[i]; ids_out[i] = ids[i]; }
    t->tacc[5] += secs(t5, now());
    {
        const hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) {
            ph_set_error("ph_tracker_match: launch failed: %s", hipGetErrorString(e_));
            t->poisoned = true;
            return PH_ELAUNCH;
        }
    }
    return k;
}

// A whole step's frames in ONE call (the sharded video step replays every rank's frames in frame order, `video.replay_tracking`):
// boxes [sum n][5] / labels [sum n] on the host, frame f's rows at offset sum(counts[0..f-1]); embeds_dev[f] = that frame's
// [counts[f]][256] device rows.  Frames without detections are skipped and do not advance the frame counter (the reference calls
// `match` only when there are thing segments, polyphonic_former_video.py:391-402); the counter starts at first_frame_id.
// kept_out / ids_out: per frame at the same offsets, kept_counts[f] entries valid.  Returns the number of frames matched (>= 0).
extern "C" int ph_tracker_match_frames(ph_tracker* t, const float* boxes, const int64_t* labels, const float* const* embeds_dev, const int32_t* counts,
                                       int nframes, int64_t first_frame_id, int32_t* kept_out, int64_t* ids_out, int32_t* kept_counts, void* stream) {
    PH_CHECK_ARG(t && counts && kept_out && ids_out && kept_counts && nframes >= 0, "bad pointer or size");
    int64_t fid = first_frame_id;
    int matched = 0;
    size_t off = 0;
    for (int f = 0; f < nframes; ++f) {
        const int n = counts[f];
        PH_CHECK_ARG(n >= 0, "negative detection count");
        kept_counts[f] = 0;
        if (n > 0) {
            PH_CHECK_ARG(boxes && labels && embeds_dev && embeds_dev[f], "bad pointer or size");
            const int k = ph_tracker_match(t, boxes + off * 5, labels + off, embeds_dev[f], n, fid, kept_out + off, ids_out + off, stream);
            if (k < 0) return k;
            kept_counts[f] = k;
            ++fid;
            ++matched;
        }
        off += (size_t)n;
    }
    return matched;
}
