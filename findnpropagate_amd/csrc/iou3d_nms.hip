// Rotated / axis-aligned BEV IoU, 3D IoU and NMS for gfx950.
//
// Results are defined by the reference's arithmetic (pcdet/ops/iou3d_nms/src/
// iou3d_nms_kernel.cu): rotated-rectangle overlap = edge/edge crossings + corners-inside
// (MARGIN 1e-2, :52-62) -> angular sort about the mean point (:201-210) -> fan shoelace
// (:220-224); iou_bev :227-234; iou_normal :327-338; NMS tile predicate :306-323,:367-384 and
// greedy sweep iou3d_nms.cpp:139-155.  The same IEEE operation sequence is evaluated here
// (-ffp-contract=off), but organised for the machine:
//   * per-box trigonometry and corner rotation are computed once per box, not per pair and
//     per corner test;
//   * polygon vertices carry their atan2 key, computed once, and are ordered by a stable
//     insertion sort (same permutation as the reference's stable bubble sort, which calls
//     atan2 twice per comparison);
//   * NMS: 64-lane waves build the 64x64 suppression tiles with one ballot-free u64 per lane;
//     the greedy sweep that the reference runs on the host after a blocking cudaMemcpy runs in
//     one wave on the device (diagonal tile resolved in registers, later column words updated
//     64 at a time), so the whole op is asynchronous on the caller's stream.
#include "common.h"
#include <cmath>
#include <vector>

namespace {

struct P2 {
    float x, y;
};

struct RBox {           // one rotated rectangle, prepared once
    float raw[7];
    P2 c[5];            // rotated corners, c[4] == c[0]
    float ncos, nsin;   // cos(-heading), sin(-heading) for the corner-inside test
    float hx, hy;       // dx/2 + 1e-2f, dy/2 + 1e-2f (f32, :61)
};

__host__ __device__ __forceinline__ float cross3(const P2 &p1, const P2 &p2, const P2 &p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__host__ __device__ __forceinline__ void prep_box(const float *b, RBox &r) {
#pragma unroll
    for (int i = 0; i < 7; ++i) r.raw[i] = b[i];
    const float hx = b[3] / 2, hy = b[4] / 2;
    const float cx = b[0], cy = b[1];
    const float cs = cosf(b[6]), sn = sinf(b[6]);
    const float ox[4] = {cx - hx, cx + hx, cx + hx, cx - hx};
    const float oy[4] = {cy - hy, cy - hy, cy + hy, cy + hy};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r.c[k].x = (ox[k] - cx) * cs + (oy[k] - cy) * (-sn) + cx;
        r.c[k].y = (ox[k] - cx) * sn + (oy[k] - cy) * cs + cy;
    }
    r.c[4] = r.c[0];
    r.ncos = cosf(-b[6]);
    r.nsin = sinf(-b[6]);
    r.hx = b[3] / 2 + 1e-2f;
    r.hy = b[4] / 2 + 1e-2f;
}

__host__ __device__ __forceinline__ bool corner_inside(const RBox &box, const P2 &p) {
    const float rx = (p.x - box.raw[0]) * box.ncos + (p.y - box.raw[1]) * (-box.nsin);
    const float ry = (p.x - box.raw[0]) * box.nsin + (p.y - box.raw[1]) * box.ncos;
    return fabsf(rx) < box.hx && fabsf(ry) < box.hy;
}

__host__ __device__ __forceinline__ bool seg_cross(const P2 &p1, const P2 &p0, const P2 &q1, const P2 &q0, P2 &ans) {
    const bool bb = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                    fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
    if (!bb) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__host__ __device__ float overlap_area(const RBox &A, const RBox &B) {
    P2 poly[24];
    float key[24];
    int cnt = 0;
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 x;
            if (seg_cross(A.c[i + 1], A.c[i], B.c[j + 1], B.c[j], x)) {
                sx = sx + x.x;
                sy = sy + x.y;
                poly[cnt++] = x;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (corner_inside(A, B.c[k])) {
            sx = sx + B.c[k].x;
            sy = sy + B.c[k].y;
            poly[cnt++] = B.c[k];
        }
        if (corner_inside(B, A.c[k])) {
            sx = sx + A.c[k].x;
            sy = sy + A.c[k].y;
            poly[cnt++] = A.c[k];
        }
    }
    if (cnt < 3) return 0.f;  // fan over < 3 vertices has zero area (reference: empty loop / zero cross)
    const float mx = sx / cnt, my = sy / cnt;
    for (int i = 0; i < cnt; ++i) key[i] = atan2f(poly[i].y - my, poly[i].x - mx);
    for (int i = 1; i < cnt; ++i) {  // stable insertion sort, ascending
        const P2 p = poly[i];
        const float k = key[i];
        int j = i - 1;
        while (j >= 0 && key[j] > k) {
            poly[j + 1] = poly[j];
            key[j + 1] = key[j];
            --j;
        }
        poly[j + 1] = p;
        key[j + 1] = k;
    }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = poly[k].x - poly[0].x, uy = poly[k].y - poly[0].y;
        const float vx = poly[k + 1].x - poly[0].x, vy = poly[k + 1].y - poly[0].y;
        area += ux * vy - uy * vx;
    }
    return fabsf(area) * 0.5f;  // == (float)(fabs(area) / 2.0)
}

__host__ __device__ __forceinline__ float iou_from_overlap(const float *a, const float *b, float ov) {
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    return ov / fmaxf(sa + sb - ov, 1e-8f);
}

// boxes_iou3d_gpu, iou3d_nms_utils.py:59-80: BEV overlap x height overlap / union, z = box centre
__host__ __device__ __forceinline__ float iou3d_from_overlap(const float *a, const float *b, float ov) {
    const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2;
    const float b_max = b[2] + b[5] / 2, b_min = b[2] - b[5] / 2;
    const float max_of_min = a_min > b_min ? a_min : b_min;
    const float min_of_max = a_max < b_max ? a_max : b_max;
    float oh = min_of_max - max_of_min;
    if (oh < 0.f) oh = 0.f;
    const float o3 = ov * oh;
    const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
    float den = va + vb - o3;
    if (den < 1e-6f) den = 1e-6f;
    return o3 / den;
}

__device__ __forceinline__ float iou_axis_aligned(const float *a, const float *b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float Sa = a[3] * a[4], Sb = b[3] * b[4];
    return inter / fmaxf(Sa + Sb - inter, 1e-8f);
}

enum { MODE_OVERLAP = 0, MODE_IOU_BEV = 1, MODE_IOU3D = 2 };

// Pairwise (N x M): 16x16 pairs per workgroup; the 16 A boxes and 16 B boxes of the tile are
// prepared once into LDS by the first 32 lanes.
template <int MODE>
__global__ __launch_bounds__(256) void pairwise_kernel(const float *__restrict__ a, int na,
                                                       const float *__restrict__ b, int nb,
                                                       float *__restrict__ out) {
    __shared__ RBox sa[16], sb[16];
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
    if (threadIdx.x < 16) {
        if (a0 + threadIdx.x < na) prep_box(a + (size_t)(a0 + threadIdx.x) * 7, sa[threadIdx.x]);
    } else if (threadIdx.x < 32) {
        const int t = threadIdx.x - 16;
        if (b0 + t < nb) prep_box(b + (size_t)(b0 + t) * 7, sb[t]);
    }
    __syncthreads();
    const int ai = a0 + ty, bi = b0 + tx;
    if (ai >= na || bi >= nb) return;
    const RBox &A = sa[ty];
    const RBox &B = sb[tx];
    const float ov = overlap_area(A, B);
    float r;
    if (MODE == MODE_OVERLAP) {
        r = ov;
    } else if (MODE == MODE_IOU_BEV) {
        r = iou_from_overlap(A.raw, B.raw, ov);
    } else {
        r = iou3d_from_overlap(A.raw, B.raw, ov);
    }
    out[(size_t)ai * nb + bi] = r;
}

__global__ __launch_bounds__(64) void aligned_overlap_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             int n, float *__restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    RBox A, B;
    prep_box(a + (size_t)i * 7, A);
    prep_box(b + (size_t)i * 7, B);
    out[i] = overlap_area(A, B);
}

// Suppression mask, same layout as the reference: mask[i * col_blocks + cb] bit j set iff
// box (cb*64 + j) comes after i and IoU > thresh.  grid (col_blocks, col_blocks), one wave each.
// BATCHED form (counts != nullptr; round 5: the per-class NMS of multi_classes_nms, model_nms_utils.py:30-66, as ONE launch pair):
// blockIdx.z = list, list z holds counts[z] <= cap boxes at boxes + z * cap * 7 and its mask at mask + z * cap * ceil(cap / 64);
// tiles beyond a list's own count leave at once.
template <bool ROTATED>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, int n, float thresh,
                                                      unsigned long long *__restrict__ mask, const int *__restrict__ counts = nullptr, int cap = 0) {
    __shared__ RBox cols[ROTATED ? 64 : 1];
    __shared__ float cols_raw[64 * 7];
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (counts) {
        n = min(counts[blockIdx.z], cap);
        boxes += (size_t)blockIdx.z * cap * 7;
        mask += (size_t)blockIdx.z * cap * ((cap + 63) / 64);
        if (rb * 64 >= n || cb * 64 >= n) return;   // (whole workgroup)
    }
    const int col_blocks = (n + 63) / 64;
    const int row_size = min(n - rb * 64, 64), col_size = min(n - cb * 64, 64);
    const int lane = threadIdx.x;
    if (lane < col_size) {
        const float *src = boxes + (size_t)(cb * 64 + lane) * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) cols_raw[lane * 7 + k] = src[k];
        if (ROTATED) prep_box(src, cols[lane]);
    }
    __syncthreads();
    if (lane >= row_size) return;
    const int row = rb * 64 + lane;
    unsigned long long bits = 0;
    if (cb >= rb) {  // tiles left of the diagonal are never read by the sweep; they stay 0
        const int start = (rb == cb) ? lane + 1 : 0;
        if (ROTATED) {
            RBox me;
            prep_box(boxes + (size_t)row * 7, me);
            for (int j = start; j < col_size; ++j) {
                const float ov = overlap_area(me, cols[j]);
                if (iou_from_overlap(me.raw, cols[j].raw, ov) > thresh) bits |= 1ull << j;
            }
        } else {
            float me[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) me[k] = boxes[(size_t)row * 7 + k];
            for (int j = start; j < col_size; ++j)
                if (iou_axis_aligned(me, cols_raw + j * 7) > thresh) bits |= 1ull << j;
        }
    }
    mask[(size_t)row * col_blocks + cb] = bits;
}

// Greedy sweep in ONE wave (iou3d_nms.cpp:139-155 on the device).
constexpr int kMaxColBlocks = 1024;  // up to 65536 boxes
__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long *__restrict__ mask, int n,
                                                       int64_t *__restrict__ keep, int *__restrict__ num_keep,
                                                       const int *__restrict__ counts = nullptr, int cap = 0) {
    __shared__ unsigned long long remv[kMaxColBlocks];
    const int lane = threadIdx.x;
    if (counts) {   // (batched: blockIdx.x = list; see nms_mask_kernel)
        n = min(counts[blockIdx.x], cap);
        mask += (size_t)blockIdx.x * cap * ((cap + 63) / 64);
        keep += (size_t)blockIdx.x * cap;
        num_keep += blockIdx.x;
        if (n <= 0) {
            if (lane == 0) *num_keep = 0;
            return;
        }
    }
    const int col_blocks = (n + 63) / 64;
    for (int j = lane; j < col_blocks; j += 64) remv[j] = 0;
    __syncthreads();
    int nk = 0;
    for (int bi = 0; bi < col_blocks; ++bi) {
        unsigned long long cur = remv[bi];
        const int row = bi * 64 + lane;
        const unsigned long long diag = row < n ? mask[(size_t)row * col_blocks + bi] : 0ull;
        const int rows_here = min(64, n - bi * 64);
        unsigned long long kept = 0;
        for (int l = 0; l < rows_here; ++l) {  // wave-uniform loop: cur/kept are uniform values
            if (!((cur >> l) & 1ull)) {
                kept |= 1ull << l;
                const unsigned lo = __shfl((unsigned)(diag & 0xffffffffull), l);
                const unsigned hi = __shfl((unsigned)(diag >> 32), l);
                cur |= ((unsigned long long)hi << 32) | lo;
            }
        }
        if ((kept >> lane) & 1ull) keep[nk + __popcll(kept & ((1ull << lane) - 1ull))] = row;
        nk += __popcll(kept);
        for (int j = bi + 1 + lane; j < col_blocks; j += 64) {
            unsigned long long acc = 0;
            unsigned long long k = kept;
            while (k) {
                const int l = __ffsll((long long)k) - 1;
                k &= k - 1;
                acc |= mask[(size_t)(bi * 64 + l) * col_blocks + j];
            }
            remv[j] |= acc;
        }
        __syncthreads();
    }
    if (lane == 0) *num_keep = nk;
}

// boxes_aligned_iou3d_gpu (iou3d_nms_utils.py:83-117) in one launch
__global__ __launch_bounds__(64) void aligned_iou3d_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           int n, float *__restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    RBox A, B;
    prep_box(a + (size_t)i * 7, A);
    prep_box(b + (size_t)i * 7, B);
    out[i] = iou3d_from_overlap(A.raw, B.raw, overlap_area(A, B));
}

// Recall bookkeeping of one frame (Detector3DTemplate.generate_recall_record, detector3d_template.py:342-397) in ONE
// workgroup, accumulated into a device-resident counter vector: no host synchronisation per frame (the reference does
// ~6 `.item()` per IoU threshold plus one per ground-truth box).  Layout of `counters` (int64):
//   [gt, num_3known, num_6known, num_4unknown, num_7unknown] then per threshold
//   [roi, rcnn, rcnn_3known, rcnn_6known, rcnn_4unknown, rcnn_7unknown].
struct RecallCfg {
    int T;
    float thr[8];
    unsigned known3_bits, known6_bits;   // bit l set: class label l (1-based) is a "known" class
};

__global__ __launch_bounds__(256) void recall_kernel(const float *__restrict__ preds, int pred_stride, int max_preds,
                                                     const float *__restrict__ pred_count,
                                                     const float *__restrict__ gt, int G, int gt_stride,
                                                     const float *__restrict__ rois, int n_rois, int rois_stride,
                                                     RecallCfg cfg, unsigned long long *__restrict__ counters) {
    __shared__ RBox sa[16], sb[16];
    __shared__ int best[16], best_roi[16];
    __shared__ int last_nonzero;
    __shared__ int acc[5 + 6 * 8];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    if (tid == 0) last_nonzero = -1;
    if (tid < 5 + 6 * 8) acc[tid] = 0;
    __syncthreads();
    // trailing all-zero rows are padding (:342-346): row j counts iff some row at or after it has a non-zero sum
    for (int j = tid; j < G; j += 256) {
        float sum = 0.f;
        for (int c = 0; c < gt_stride; ++c) sum += gt[(size_t)j * gt_stride + c];
        if (sum != 0.f) atomicMax(&last_nonzero, j);
    }
    __syncthreads();
    const int n_gt = last_nonzero + 1;
    int n_pred = max_preds;
    if (pred_count) n_pred = min(max_preds, max(0, (int)*pred_count));
    for (int g0 = 0; g0 < n_gt; g0 += 16) {
        if (tid < 16) {
            best[tid] = 0;       // IoUs are >= 0: their float bits order like ints
            best_roi[tid] = 0;
            if (g0 + tid < n_gt) prep_box(gt + (size_t)(g0 + tid) * gt_stride, sb[tid]);
        }
        for (int pass = 0; pass < 2; ++pass) {
            const float *src = pass ? rois : preds;
            const int n_src = pass ? (rois ? n_rois : 0) : n_pred, stride = pass ? rois_stride : pred_stride;
            for (int p0 = 0; p0 < n_src; p0 += 16) {
                __syncthreads();
                if (tid < 16 && p0 + tid < n_src) prep_box(src + (size_t)(p0 + tid) * stride, sa[tid]);
                __syncthreads();
                if (p0 + ty < n_src && g0 + tx < n_gt) {
                    const float iou = iou3d_from_overlap(sa[ty].raw, sb[tx].raw, overlap_area(sa[ty], sb[tx]));
                    atomicMax(pass ? &best_roi[tx] : &best[tx], __float_as_int(iou));
                }
            }
        }
        __syncthreads();
        if (tid < 16 && g0 + tid < n_gt) {
            const int label = (int)(long long)gt[(size_t)(g0 + tid) * gt_stride + (gt_stride - 1)];
            const bool k3 = label >= 0 && label < 32 && ((cfg.known3_bits >> label) & 1u);
            const bool k6 = label >= 0 && label < 32 && ((cfg.known6_bits >> label) & 1u);
            atomicAdd(&acc[0], 1);
            atomicAdd(&acc[k3 ? 1 : 4], 1);      // num_3known | num_7unknown
            atomicAdd(&acc[k6 ? 2 : 3], 1);      // num_6known | num_4unknown
            const float b = __int_as_float(best[tid]), br = __int_as_float(best_roi[tid]);
            for (int t = 0; t < cfg.T; ++t) {
                int *a = acc + 5 + 6 * t;
                if (rois && br > cfg.thr[t]) atomicAdd(&a[0], 1);
                if (n_pred > 0 && b > cfg.thr[t]) {
                    atomicAdd(&a[1], 1);
                    atomicAdd(&a[k3 ? 2 : 5], 1);    // rcnn_3known | rcnn_7unknown
                    atomicAdd(&a[k6 ? 3 : 4], 1);    // rcnn_6known | rcnn_4unknown
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid < 5 + 6 * cfg.T && acc[tid]) atomicAdd(&counters[tid], (unsigned long long)acc[tid]);
}

template <int MODE>
int launch_pairwise(const float *a, int na, const float *b, int nb, float *out, fnp_stream_t stream) {
    if (na < 0 || nb < 0) return FNP_ERR_ARG;
    if (na == 0 || nb == 0) return FNP_OK;
    if (!a || !b || !out) return FNP_ERR_ARG;
    dim3 grid(fnp_divup(nb, 16), fnp_divup(na, 16));
    hipLaunchKernelGGL(pairwise_kernel<MODE>, grid, dim3(256), 0, (hipStream_t)stream, a, na, b, nb, out);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

template <bool ROTATED>
int launch_nms(const float *boxes, int n, float thresh, void *ws, int64_t *keep, int *num_keep, fnp_stream_t stream) {
    if (n < 0 || !num_keep) return FNP_ERR_ARG;
    if (n == 0) {
        FNP_HIP_TRY(hipMemsetAsync(num_keep, 0, sizeof(int), (hipStream_t)stream));
        return FNP_OK;
    }
    if (!boxes || !ws || !keep) return FNP_ERR_ARG;
    const int cb = (n + 63) / 64;
    if (cb > kMaxColBlocks) return FNP_ERR_ARG;
    hipLaunchKernelGGL(nms_mask_kernel<ROTATED>, dim3(cb, cb), dim3(64), 0, (hipStream_t)stream, boxes, n, thresh,
                       (unsigned long long *)ws);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long *)ws, n,
                       keep, num_keep);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

}  // namespace

extern "C" int fnp_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out, fnp_stream_t s) {
    return launch_pairwise<MODE_OVERLAP>(a, na, b, nb, out, s);
}
extern "C" int fnp_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out, fnp_stream_t s) {
    return launch_pairwise<MODE_IOU_BEV>(a, na, b, nb, out, s);
}
extern "C" int fnp_boxes_iou3d(const float *a, int na, const float *b, int nb, float *out, fnp_stream_t s) {
    return launch_pairwise<MODE_IOU3D>(a, na, b, nb, out, s);
}
extern "C" int fnp_boxes_aligned_overlap_bev(const float *a, const float *b, int n, float *out, fnp_stream_t s) {
    if (n < 0) return FNP_ERR_ARG;
    if (n == 0) return FNP_OK;
    if (!a || !b || !out) return FNP_ERR_ARG;
    hipLaunchKernelGGL(aligned_overlap_kernel, dim3(fnp_divup(n, 64)), dim3(64), 0, (hipStream_t)s, a, b, n, out);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
extern "C" int fnp_boxes_aligned_iou3d(const float *a, const float *b, int n, float *out, fnp_stream_t s) {
    if (n < 0) return FNP_ERR_ARG;
    if (n == 0) return FNP_OK;
    if (!a || !b || !out) return FNP_ERR_ARG;
    hipLaunchKernelGGL(aligned_iou3d_kernel, dim3(fnp_divup(n, 64)), dim3(64), 0, (hipStream_t)s, a, b, n, out);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
extern "C" int fnp_recall_counters(const float *preds, int pred_stride, int max_preds, const float *pred_count,
                                   const float *gt, int num_gt, int gt_stride, const float *rois, int num_rois,
                                   int rois_stride, const float *thresh, int num_thresh, unsigned known3_bits,
                                   unsigned known6_bits, int64_t *counters, fnp_stream_t s) {
    if (max_preds < 0 || num_gt < 0 || num_rois < 0 || num_thresh < 1 || num_thresh > 8 || !thresh || !counters) return FNP_ERR_ARG;
    if (num_gt == 0) return FNP_OK;
    if (!gt || gt_stride < 8 || (max_preds > 0 && (!preds || pred_stride < 7)) || (rois && rois_stride < 7)) return FNP_ERR_ARG;
    RecallCfg cfg{};
    cfg.T = num_thresh;
    for (int t = 0; t < num_thresh; ++t) cfg.thr[t] = thresh[t];
    cfg.known3_bits = known3_bits;
    cfg.known6_bits = known6_bits;
    hipLaunchKernelGGL(recall_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, preds, pred_stride, max_preds, pred_count, gt,
                       num_gt, gt_stride, rois, num_rois, rois_stride, cfg, (unsigned long long *)counters);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
// Several score-sorted box lists at once (the classes of multi_classes_nms): list z = counts[z] boxes (a DEVICE array: the counts
// come out of a score threshold) in a (lists, cap, 7) tensor; keep (lists, cap) int64, num_keep (lists) int32.  Two launches, no
// host synchronisation; per list the result of fnp_nms_rotated / fnp_nms_normal on its first counts[z] boxes.
extern "C" int64_t fnp_nms_batched_workspace_bytes(int lists, int cap) {
    const int64_t cb = (cap + 63) / 64;
    return (int64_t)(lists > 0 ? lists : 1) * (cap > 0 ? cap : 1) * (cb > 0 ? cb : 1) * 8;
}
extern "C" int fnp_nms_batched(const float *boxes, const int *counts, int lists, int cap, float thresh, int rotated, void *ws, int64_t *keep,
                               int *num_keep, fnp_stream_t s) {
    if (lists < 0 || cap < 0 || !num_keep) return FNP_ERR_ARG;
    if (lists == 0) return FNP_OK;
    if (!counts || (cap > 0 && (!boxes || !ws || !keep))) return FNP_ERR_ARG;
    const int cb = (cap + 63) / 64;
    if (cb > kMaxColBlocks || lists > 65535) return FNP_ERR_ARG;
    if (cap > 0) {
        if (rotated)
            hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(cb, cb, lists), dim3(64), 0, (hipStream_t)s, boxes, 0, thresh, (unsigned long long *)ws, counts, cap);
        else
            hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(cb, cb, lists), dim3(64), 0, (hipStream_t)s, boxes, 0, thresh, (unsigned long long *)ws, counts, cap);
        FNP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(lists), dim3(64), 0, (hipStream_t)s, (const unsigned long long *)ws, 0, keep, num_keep, counts, cap);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int64_t fnp_nms_workspace_bytes(int n) {
    const int64_t cb = (n + 63) / 64;
    return (int64_t)(n > 0 ? n : 1) * (cb > 0 ? cb : 1) * 8;
}
extern "C" int fnp_nms_rotated(const float *boxes, int n, float thresh, void *ws, int64_t *keep, int *num_keep,
                               fnp_stream_t s) {
    return launch_nms<true>(boxes, n, thresh, ws, keep, num_keep, s);
}
extern "C" int fnp_nms_normal(const float *boxes, int n, float thresh, void *ws, int64_t *keep, int *num_keep,
                              fnp_stream_t s) {
    return launch_nms<false>(boxes, n, thresh, ws, keep, num_keep, s);
}

// ---- host entry points (iou3d_cpu.cpp:232-272: boxes_iou_bev_cpu / boxes_aligned_iou_bev_cpu) ----
// Plain host loops over the same prepared-box / overlap functions as the kernels above (compiled for
// the host; libm's cosf/sinf/atan2f instead of the device library's).  Used by the pseudo-label
// mixing in dataloader workers (pseudo_loader.py:29-55): no stream, no device memory.
extern "C" int fnp_host_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out) {
    if (na < 0 || nb < 0 || ((na > 0 && nb > 0) && (!a || !b || !out))) return FNP_ERR_ARG;
    std::vector<RBox> pb((size_t)nb);
    for (int j = 0; j < nb; ++j) prep_box(b + (size_t)j * 7, pb[j]);
    for (int i = 0; i < na; ++i) {
        RBox A;
        prep_box(a + (size_t)i * 7, A);
        for (int j = 0; j < nb; ++j)
            out[(size_t)i * nb + j] = iou_from_overlap(a + (size_t)i * 7, b + (size_t)j * 7, overlap_area(A, pb[j]));
    }
    return FNP_OK;
}

extern "C" int fnp_host_boxes_aligned_iou_bev(const float *a, const float *b, int n, float *out) {
    if (n < 0 || (n > 0 && (!a || !b || !out))) return FNP_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        RBox A, B;
        prep_box(a + (size_t)i * 7, A);
        prep_box(b + (size_t)i * 7, B);
        out[i] = iou_from_overlap(a + (size_t)i * 7, b + (size_t)i * 7, overlap_area(A, B));
    }
    return FNP_OK;
}
