/*
 * fnp_oracle.c — CPU restatement of the reference algorithms on the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under findnpropagate_amd/ may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * tree).  Arithmetic is kept expression-for-expression (f32 vs f64 promotion, strictness of
 * comparisons) because the GPU parity bar is bit-exactness of the integer outputs.
 *
 * Pinning status:
 *   - point-in-box, rotated BEV overlap / IoU, axis-aligned IoU, NMS sweeps: pinned against the
 *     reference's own sources built by oracle/Makefile into oracle/_ref (see tests/).
 *   - voxeliser and sparse convolution restate third-party spconv (not vendored, not version
 *     pinned by the reference: docker/Dockerfile:55, docs/INSTALL.md:9, setup.py:48) from its
 *     documented semantics: PARITY UNPINNED against spconv itself;
 *     self-pinned by dense-convolution equivalence with torch.nn.functional.conv3d and
 *     hand-computed known-answer cases in tests/test_oracle_spconv.py.
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * roiaware_pool3d: check_pt_in_box3d
 *   GPU variant: roiaware_pool3d_kernel.cu:16-36 (MARGIN 1e-5)
 *   CPU variant: roiaware_pool3d.cpp:121-140   (MARGIN 1e-2)
 * ------------------------------------------------------------------------------------------ */
static int orc_pt_in_box(const float *pt, const float *box, float margin) {
    float x = pt[0], y = pt[1], z = pt[2];
    float cx = box[0], cy = box[1], cz = box[2];
    float dx = box[3], dy = box[4], dz = box[5], rz = box[6];
    if ((double)fabsf(z - cz) > (double)dz / 2.0) return 0; /* kernel.cu:32 */
    float sx = x - cx, sy = y - cy;
    float cosa = cosf(-rz), sina = sinf(-rz); /* kernel.cu:17 (float overloads) */
    float lx = sx * cosa + sy * (-sina);      /* kernel.cu:18 */
    float ly = sx * sina + sy * cosa;         /* kernel.cu:19 */
    int in_x = (double)fabsf(lx) < (double)dx / 2.0 + (double)margin; /* kernel.cu:34 */
    int in_y = (double)fabsf(ly) < (double)dy / 2.0 + (double)margin;
    return in_x & in_y;
}

/* points_in_boxes_kernel, roiaware_pool3d_kernel.cu:313-336: first box index or -1. */
ORC_API void orc_points_in_boxes(const float *boxes, const float *pts, int *out, int B, int T, int M) {
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < M; ++m) {
            int found = -1;
            for (int t = 0; t < T; ++t)
                if (orc_pt_in_box(pts + ((size_t)b * M + m) * 3, boxes + ((size_t)b * T + t) * 7, 1e-5f)) {
                    found = t;
                    break;
                }
            out[(size_t)b * M + m] = found;
        }
}

/* Box Seeker hot loop 4 (frustum_proposals_v1.py:930-932): per-candidate counts. */
ORC_API void orc_points_in_boxes_count(const float *boxes, const float *pts, int *counts, int T, int M) {
    for (int t = 0; t < T; ++t) {
        int c = 0;
        for (int m = 0; m < M; ++m) c += orc_pt_in_box(pts + (size_t)m * 3, boxes + (size_t)t * 7, 1e-5f);
        counts[t] = c;
    }
}

/* points_in_boxes_cpu, roiaware_pool3d.cpp:143-168: dense (T,M) flags, MARGIN 1e-2. */
ORC_API void orc_points_in_boxes_dense(const float *boxes, const float *pts, int *out, int T, int M) {
    for (int t = 0; t < T; ++t)
        for (int m = 0; m < M; ++m)
            out[(size_t)t * M + m] = orc_pt_in_box(pts + (size_t)m * 3, boxes + (size_t)t * 7, 1e-2f);
}

/* ------------------------------------------------------------------------------------------
 * iou3d_nms: rotated rectangle overlap.  iou3d_nms_kernel.cu:36-225 (== iou3d_cpu.cpp:59-226).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float x, y;
} orc_pt2;

static float orc_cross2(orc_pt2 a, orc_pt2 b) { return a.x * b.y - a.y * b.x; } /* :36-38 */
static float orc_cross3(orc_pt2 p1, orc_pt2 p2, orc_pt2 p0) {                  /* :40-42 */
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static float orc_minf(float a, float b) { return a > b ? b : a; }
static float orc_maxf(float a, float b) { return a > b ? a : b; }

static int orc_rect_cross(orc_pt2 p1, orc_pt2 p2, orc_pt2 q1, orc_pt2 q2) { /* :44-50 */
    return orc_minf(p1.x, p2.x) <= orc_maxf(q1.x, q2.x) && orc_minf(q1.x, q2.x) <= orc_maxf(p1.x, p2.x) &&
           orc_minf(p1.y, p2.y) <= orc_maxf(q1.y, q2.y) && orc_minf(q1.y, q2.y) <= orc_maxf(p1.y, p2.y);
}

static int orc_in_box2d(const float *box, orc_pt2 p) { /* :52-62, MARGIN 1e-2, all f32 */
    const float margin = 1e-2f;
    float ac = cosf(-box[6]), as = sinf(-box[6]);
    float rx = (p.x - box[0]) * ac + (p.y - box[1]) * (-as);
    float ry = (p.x - box[0]) * as + (p.y - box[1]) * ac;
    return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

static int orc_seg_intersection(orc_pt2 p1, orc_pt2 p0, orc_pt2 q1, orc_pt2 q0, orc_pt2 *ans) { /* :64-94 */
    if (!orc_rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = orc_cross3(q0, p1, p0);
    float s2 = orc_cross3(p1, q1, p0);
    float s3 = orc_cross3(p0, q1, q0);
    float s4 = orc_cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = orc_cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static orc_pt2 orc_rot_about(orc_pt2 c, float ac, float as, orc_pt2 p) { /* :96-100 */
    orc_pt2 r;
    r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    return r;
}

ORC_API float orc_box_overlap(const float *A, const float *B) { /* :104-225 */
    float a_ang = A[6], b_ang = B[6];
    float ahx = A[3] / 2, bhx = B[3] / 2, ahy = A[4] / 2, bhy = B[4] / 2;
    orc_pt2 ca = {A[0], A[1]}, cb = {B[0], B[1]};
    orc_pt2 pa[5] = {{A[0] - ahx, A[1] - ahy}, {A[0] + ahx, A[1] - ahy}, {A[0] + ahx, A[1] + ahy}, {A[0] - ahx, A[1] + ahy}};
    orc_pt2 pb[5] = {{B[0] - bhx, B[1] - bhy}, {B[0] + bhx, B[1] - bhy}, {B[0] + bhx, B[1] + bhy}, {B[0] - bhx, B[1] + bhy}};
    float acs = cosf(a_ang), asn = sinf(a_ang), bcs = cosf(b_ang), bsn = sinf(b_ang);
    for (int k = 0; k < 4; ++k) {
        pa[k] = orc_rot_about(ca, acs, asn, pa[k]);
        pb[k] = orc_rot_about(cb, bcs, bsn, pb[k]);
    }
    pa[4] = pa[0];
    pb[4] = pb[0];

    orc_pt2 poly[24]; /* the reference sizes this 16 (:157); 16 edge crossings + 8 corners = 24 is the true bound */
    orc_pt2 centre = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (orc_seg_intersection(pa[i + 1], pa[i], pb[j + 1], pb[j], &poly[cnt])) {
                centre.x = centre.x + poly[cnt].x;
                centre.y = centre.y + poly[cnt].y;
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) { /* :178-195 */
        if (orc_in_box2d(A, pb[k])) {
            centre.x = centre.x + pb[k].x;
            centre.y = centre.y + pb[k].y;
            poly[cnt++] = pb[k];
        }
        if (orc_in_box2d(B, pa[k])) {
            centre.x = centre.x + pa[k].x;
            centre.y = centre.y + pa[k].y;
            poly[cnt++] = pa[k];
        }
    }
    centre.x /= cnt; /* :197-198 (cnt == 0 gives NaN that is never read) */
    centre.y /= cnt;
    /* :201-210 bubble sort ascending by atan2 about the centre (stable: swaps on strict >) */
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(poly[i].y - centre.y, poly[i].x - centre.x) > atan2f(poly[i + 1].y - centre.y, poly[i + 1].x - centre.x)) {
                orc_pt2 t = poly[i];
                poly[i] = poly[i + 1];
                poly[i + 1] = t;
            }
    float area = 0; /* :220-224 fan shoelace */
    for (int k = 0; k < cnt - 1; ++k) {
        orc_pt2 u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
        orc_pt2 v = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
        area += orc_cross2(u, v);
    }
    return (float)((double)fabsf(area) / 2.0);
}

ORC_API float orc_iou_bev(const float *A, const float *B) { /* iou3d_nms_kernel.cu:227-234 */
    float sa = A[3] * A[4], sb = B[3] * B[4];
    float so = orc_box_overlap(A, B);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

ORC_API float orc_iou_normal(const float *a, const float *b) { /* iou3d_nms_kernel.cu:327-338 */
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float Sa = a[3] * a[4], Sb = b[3] * b[4];
    return inter / fmaxf(Sa + Sb - inter, 1e-8f);
}

ORC_API void orc_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out) { /* :236-250 */
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_box_overlap(a + i * 7, b + j * 7);
}
ORC_API void orc_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out) { /* :266-278 */
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_iou_bev(a + i * 7, b + j * 7);
}
ORC_API void orc_boxes_aligned_overlap_bev(const float *a, const float *b, int n, float *out) { /* :252-264 */
    for (int i = 0; i < n; ++i) out[i] = orc_box_overlap(a + i * 7, b + i * 7);
}

/* boxes_iou3d_gpu, iou3d_nms_utils.py:48-81 (torch f32 elementwise ops). */
ORC_API void orc_boxes_iou3d(const float *a, int na, const float *b, int nb, float *out) {
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            const float *A = a + i * 7, *B = b + j * 7;
            float a_max = A[2] + A[5] / 2, a_min = A[2] - A[5] / 2;
            float b_max = B[2] + B[5] / 2, b_min = B[2] - B[5] / 2;
            float bev = orc_box_overlap(A, B);
            float max_of_min = a_min > b_min ? a_min : b_min;
            float min_of_max = a_max < b_max ? a_max : b_max;
            float oh = min_of_max - max_of_min;
            if (oh < 0.f) oh = 0.f;
            float o3 = bev * oh;
            float va = A[3] * A[4] * A[5], vb = B[3] * B[4] * B[5];
            float den = va + vb - o3;
            if (den < 1e-6f) den = 1e-6f;
            out[(size_t)i * nb + j] = o3 / den;
        }
}

/* nms_gpu / nms_normal_gpu: mask predicate iou3d_nms_kernel.cu:306-323,367-384 (bit j of row i set
 * iff j > i and IoU(i,j) > thresh) + greedy sweep iou3d_nms.cpp:139-155,189-205.
 * boxes pre-sorted by score.  Returns number kept; keep holds indices into the sorted list. */
ORC_API int orc_nms(const float *boxes, int n, float thresh, int rotated, int64_t *keep) {
    unsigned char *removed = (unsigned char *)calloc(n > 0 ? n : 1, 1);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (removed[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; ++j) {
            float v = rotated ? orc_iou_bev(boxes + i * 7, boxes + j * 7) : orc_iou_normal(boxes + i * 7, boxes + j * 7);
            if (v > thresh) removed[j] = 1;
        }
    }
    free(removed);
    return nk;
}

/* ------------------------------------------------------------------------------------------
 * Coordinate hash used by the voxeliser / rulebook restatements (replaces spconv's dense
 * coor_to_voxelidx table: same lookups, less memory).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t *keys;
    int *vals;
    size_t mask;
} orc_map;

static void orc_map_init(orc_map *m, size_t n) {
    size_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    m->keys = (int64_t *)malloc(cap * sizeof(int64_t));
    m->vals = (int *)malloc(cap * sizeof(int));
    for (size_t i = 0; i < cap; ++i) m->keys[i] = -1;
    m->mask = cap - 1;
}
static void orc_map_free(orc_map *m) {
    free(m->keys);
    free(m->vals);
}
static size_t orc_mix(int64_t k) {
    uint64_t x = (uint64_t)k;
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (size_t)x;
}
static int orc_map_get(const orc_map *m, int64_t k) {
    size_t s = orc_mix(k) & m->mask;
    while (m->keys[s] != -1) {
        if (m->keys[s] == k) return m->vals[s];
        s = (s + 1) & m->mask;
    }
    return -1;
}
/* returns existing value, or inserts v and returns -1 */
static int orc_map_put_if_absent(orc_map *m, int64_t k, int v) {
    size_t s = orc_mix(k) & m->mask;
    while (m->keys[s] != -1) {
        if (m->keys[s] == k) return m->vals[s];
        s = (s + 1) & m->mask;
    }
    m->keys[s] = k;
    m->vals[s] = v;
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * Voxeliser: spconv Point2VoxelCPU3d.point_to_voxel as called at
 * pcdet/datasets/processor/data_processor.py:38-61 (semantics: SURVEY.md Appendix A.1).
 *   points (n,C) f32; range_min/voxel_size xyz f32; grid xyz int.
 *   voxels (max_voxels,max_points,C) zero-filled by the caller; coords (max_voxels,3) [z,y,x];
 *   num_points (max_voxels,).  Returns the number of voxels.
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_voxelize(const float *points, int n, int C, const float *range_min, const float *voxel_size,
                         const int *grid, int max_points, int max_voxels, float *voxels, int *coords,
                         int *num_points) {
    orc_map map;
    orc_map_init(&map, (size_t)(n < max_voxels ? n : max_voxels));
    int voxel_num = 0;
    for (int i = 0; i < n; ++i) {
        int c[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            float q = (points[(size_t)i * C + j] - range_min[j]) / voxel_size[j]; /* f32 */
            int cj = (int)floorf(q);
            if (cj < 0 || cj >= grid[j]) {
                failed = 1;
                break;
            }
            c[j] = cj;
        }
        if (failed) continue;
        int64_t key = ((int64_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
        int vid = orc_map_get(&map, key);
        if (vid == -1) {
            if (voxel_num >= max_voxels) continue; /* spconv >= 1.2 / 2.x: skip the point */
            vid = voxel_num++;
            orc_map_put_if_absent(&map, key, vid);
            coords[vid * 3 + 0] = c[2];
            coords[vid * 3 + 1] = c[1];
            coords[vid * 3 + 2] = c[0];
            num_points[vid] = 0;
        }
        int np = num_points[vid];
        if (np < max_points) {
            memcpy(voxels + ((size_t)vid * max_points + np) * C, points + (size_t)i * C, sizeof(float) * C);
            num_points[vid] = np + 1;
        }
    }
    orc_map_free(&map);
    return voxel_num;
}

/* MeanVFE.forward, pcdet/models/backbones_3d/vfe/mean_vfe.py:25-29: `voxels.sum(dim=1) / clamp_min(num_points, 1)`.
 *
 * The ORDER in which the <= max_points addends of a column meet is torch's (round 6: tests/golden/meanvfe_golden.npz, written by the
 * reference's own class on the CPU, showed that a plain slot-order sum differs from it in the last bit of the LAST column of
 * 5-feature points — the time lag of multi-sweep nuScenes points; single-sweep scenes have t = 0 and never showed it).  torch's CPU
 * `sum` of a float tensor is cascade_sum (aten/src/ATen/native/cpu/SumKernel.cpp; torch 2.10 installed here, the kernel has had this
 * form since 1.7).  For a contiguous (M, P, C) tensor reduced over dim 1 the 2-D loop it sees is size0 = P (the reduced dimension,
 * stride C) x size1 = C (stride 1); with C < 8 = Vec<float>::size() of the narrowest vector build, neither vectorised form applies
 * and it runs scalar_outer_sum:
 *     columns 0 .. 4 * (C / 4) - 1, four at a time: multi_row_sum  — every column on its own, slots in order, cascaded through four
 *                                                     accumulator levels in blocks of 2^max(4, ceil_log2(P) / 4) slots;
 *     the C % 4 columns left over, one at a time:   row_sum        — FOUR interleaved partial sums (slot j goes to partial j & 3
 *                                                     for j < 4 * (P / 4), each partial a multi_row_sum over its P / 4 slots), then
 *                                                     partial 0 takes the P % 4 tail slots, then partials 1, 2, 3.
 * With P = 10, C = 5 (transfusion_lidar.yaml): x y z intensity in slot order, t = ((((t0 + t4) + t8) + t9) + (t1 + t5)) + (t2 + t6))
 * + (t3 + t7).  C >= 8 columns take torch's vectorised forms, whose grouping depends on the host's vector ISA (AVX2 / AVX-512): not
 * restated — such inputs are summed in slot order here (they are not on the path: nuScenes 5, KITTI 4 features). */
static int orc_ceil_log2(int64_t x) {
    int l = 0;
    while (((int64_t)1 << l) < x) ++l;
    return l;
}
/* multi_row_sum for one column: elements e = 0 .. size - 1 at base[e * stride] */
static float orc_cascade_sum(const float *base, int64_t stride, int64_t size) {
    const int levels = 4;
    int lp = orc_ceil_log2(size) / levels;
    if (lp < 4) lp = 4;
    const int64_t step = (int64_t)1 << lp, mask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = 0;
    while (i + step <= size) {
        for (int64_t j = 0; j < step; ++j, ++i) acc[0] += base[i * stride];
        for (int j = 1; j < levels; ++j) {
            acc[j] += acc[j - 1];
            acc[j - 1] = 0.f;
            if ((i & (mask << (j * lp))) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] += base[i * stride];
    for (int j = 1; j < levels; ++j) acc[0] += acc[j];
    return acc[0];
}
ORC_API void orc_mean_vfe(const float *voxels, const int *num_points, int M, int max_points, int C, float *out) {
    const int P = max_points, c4 = C < 8 ? (C / 4) * 4 : C;
    for (int v = 0; v < M; ++v) {
        const float norm = (float)(num_points[v] < 1 ? 1 : num_points[v]);
        const float *row = voxels + (size_t)v * P * C;
        for (int c = 0; c < C; ++c) {
            float s;
            if (C >= 8) {                       /* (outside the restated domain: slot order) */
                s = 0.f;
                for (int p = 0; p < P; ++p) s += row[(size_t)p * C + c];
            } else if (c < c4) {
                s = orc_cascade_sum(row + c, C, P);
            } else {                            /* row_sum: ilp_factor 4 */
                const int ilp = P / 4;
                float part[4];
                for (int k = 0; k < 4; ++k) part[k] = orc_cascade_sum(row + (size_t)k * C + c, (int64_t)4 * C, ilp);
                for (int p = ilp * 4; p < P; ++p) part[0] += row[(size_t)p * C + c];
                for (int k = 1; k < 4; ++k) part[0] += part[k];
                s = part[0];
            }
            out[(size_t)v * C + c] = s / norm;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Sparse convolution (spconv semantics; call sites
 * pcdet/models/backbones_3d/spconv_backbone.py:12-17,39-46,193-234).
 * Rulebook = gather/scatter pair lists per kernel offset, built first-come like spconv's CPU
 * path; convolution = spconv's native gather -> per-offset GEMM -> scatter-add in f32.
 * ------------------------------------------------------------------------------------------ */
static int64_t orc_key4(int b, int z, int y, int x, const int *shape) {
    return (((int64_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

/* SubM: out sites == in sites.  pairs_in/pairs_out (K, n) with counts pair_n (K,).
 * Kernel offset order: k = (kz*kH + ky)*kW + kx, neighbour = idx + (kz - kD/2, ...). */
ORC_API void orc_rulebook_subm(const int *coords, int n, const int *shape, const int *ksize, int *pairs_in,
                               int *pairs_out, int *pair_n) {
    orc_map map;
    orc_map_init(&map, (size_t)n);
    for (int i = 0; i < n; ++i)
        orc_map_put_if_absent(&map, orc_key4(coords[i * 4], coords[i * 4 + 1], coords[i * 4 + 2], coords[i * 4 + 3], shape), i);
    int K = ksize[0] * ksize[1] * ksize[2];
    for (int k = 0; k < K; ++k) pair_n[k] = 0;
    for (int o = 0; o < n; ++o) {
        int b = coords[o * 4], z = coords[o * 4 + 1], y = coords[o * 4 + 2], x = coords[o * 4 + 3];
        for (int kz = 0; kz < ksize[0]; ++kz)
            for (int ky = 0; ky < ksize[1]; ++ky)
                for (int kx = 0; kx < ksize[2]; ++kx) {
                    int iz = z + kz - ksize[0] / 2, iy = y + ky - ksize[1] / 2, ix = x + kx - ksize[2] / 2;
                    if (iz < 0 || iy < 0 || ix < 0 || iz >= shape[0] || iy >= shape[1] || ix >= shape[2]) continue;
                    int i = orc_map_get(&map, orc_key4(b, iz, iy, ix, shape));
                    if (i < 0) continue;
                    int k = (kz * ksize[1] + ky) * ksize[2] + kx;
                    pairs_in[(size_t)k * n + pair_n[k]] = i;
                    pairs_out[(size_t)k * n + pair_n[k]] = o;
                    pair_n[k]++;
                }
    }
    orc_map_free(&map);
}

/* Strided SparseConv3d: out_shape = floor((D + 2p - k)/s) + 1; output site o exists iff some
 * input i = o*s - p + kappa.  Output rows are numbered first-come (input order, then kernel
 * offset order); the row order is implementation-defined in spconv, so tests compare by
 * coordinate.  out_coords (cap,4), pairs (K, n_in).  Returns n_out. */
ORC_API int orc_rulebook_strided(const int *coords, int n, const int *shape, const int *ksize, const int *stride,
                                 const int *padding, const int *out_shape, int *out_coords, int cap,
                                 int *pairs_in, int *pairs_out, int *pair_n) {
    orc_map map;
    orc_map_init(&map, (size_t)cap);
    int K = ksize[0] * ksize[1] * ksize[2];
    for (int k = 0; k < K; ++k) pair_n[k] = 0;
    int n_out = 0;
    for (int i = 0; i < n; ++i) {
        int b = coords[i * 4];
        int pos[3] = {coords[i * 4 + 1], coords[i * 4 + 2], coords[i * 4 + 3]};
        for (int kz = 0; kz < ksize[0]; ++kz)
            for (int ky = 0; ky < ksize[1]; ++ky)
                for (int kx = 0; kx < ksize[2]; ++kx) {
                    int kk[3] = {kz, ky, kx};
                    int o[3];
                    int ok = 1;
                    for (int d = 0; d < 3; ++d) {
                        int t = pos[d] + padding[d] - kk[d];
                        if (t < 0 || t % stride[d] != 0) {
                            ok = 0;
                            break;
                        }
                        o[d] = t / stride[d];
                        if (o[d] >= out_shape[d]) {
                            ok = 0;
                            break;
                        }
                    }
                    if (!ok) continue;
                    int64_t key = orc_key4(b, o[0], o[1], o[2], out_shape);
                    int row = orc_map_get(&map, key);
                    if (row < 0) {
                        if (n_out >= cap) continue;
                        row = n_out++;
                        orc_map_put_if_absent(&map, key, row);
                        out_coords[row * 4] = b;
                        out_coords[row * 4 + 1] = o[0];
                        out_coords[row * 4 + 2] = o[1];
                        out_coords[row * 4 + 3] = o[2];
                    }
                    int k = (kz * ksize[1] + ky) * ksize[2] + kx;
                    pairs_in[(size_t)k * n + pair_n[k]] = i;
                    pairs_out[(size_t)k * n + pair_n[k]] = row;
                    pair_n[k]++;
                }
    }
    orc_map_free(&map);
    return n_out;
}

static int orc_threads = 1; /* OpenMP threads of orc_spconv_apply (a global, not the per-thread OpenMP ICV: the
                             * python side may call from a thread pool, one scene per thread) */

/* Gather -> GEMM -> scatter-add, f32, fused multiply-add in (k, pair, cin) order.
 * weight layout = spconv 2.x (Cout, K, Cin) (detector3d_template.py:401-433 describes both
 * layouts; the python side converts 1.x checkpoints).  out must be zero on entry. */
ORC_API void orc_spconv_apply(const float *feat_in, const float *weight, const int *pairs_in, const int *pairs_out,
                              const int *pair_n, int pair_stride, int K, int Cin, int Cout, float *out) {
    float *wk = (float *)malloc(sizeof(float) * (size_t)Cin * Cout);
    for (int k = 0; k < K; ++k) {
        if (pair_n[k] == 0) continue;
        for (int ci = 0; ci < Cin; ++ci)
            for (int co = 0; co < Cout; ++co) wk[(size_t)ci * Cout + co] = weight[((size_t)co * K + k) * Cin + ci];
        /* all-core leg of the CPU baseline: within one kernel offset every output row
         * occurs in at most one pair, so the pairs of an offset are independent; the per-element fmaf chain
         * (k, then cin ascending) and therefore the result do not depend on the thread count. */
#pragma omp parallel for schedule(static) num_threads(orc_threads) if (orc_threads > 1 && pair_n[k] > 2048)
        for (int p = 0; p < pair_n[k]; ++p) {
            const float *x = feat_in + (size_t)pairs_in[(size_t)k * pair_stride + p] * Cin;
            float *y = out + (size_t)pairs_out[(size_t)k * pair_stride + p] * Cout;
            for (int ci = 0; ci < Cin; ++ci) {
                const float xv = x[ci];
                const float *w = wk + (size_t)ci * Cout;
                for (int co = 0; co < Cout; ++co) y[co] = fmaf(xv, w[co], y[co]);
            }
        }
    }
    free(wk);
}

/* threads used by the OpenMP loops above (default 1: set by oracle.py when the library is loaded) */
ORC_API void orc_set_threads(int n) { orc_threads = n > 0 ? n : 1; }
ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* BatchNorm1d eval (eps 1e-3, spconv_backbone.py:189) folded the way torch's CPU inference
 * path does it: w = gamma * rsqrt(var + eps), b = beta - mean * w, y = x*w + b; then optional
 * residual add and ReLU (spconv_backbone.py:51-67). */
ORC_API void orc_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                         int C, float *scale, float *shift) {
    for (int c = 0; c < C; ++c) {
        float inv = 1.0f / sqrtf(var[c] + eps);
        scale[c] = gamma[c] * inv;
        shift[c] = beta[c] - mean[c] * scale[c];
    }
}
ORC_API void orc_scale_shift_act(float *x, int n, int C, const float *scale, const float *shift,
                                 const float *residual, int relu) {
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < C; ++c) {
            float v = x[(size_t)i * C + c];
            if (scale) v = v * scale[c] + shift[c];
            if (residual) v = v + residual[(size_t)i * C + c];
            if (relu && v < 0.f) v = 0.f;
            x[(size_t)i * C + c] = v;
        }
}

/* SparseConvTensor.dense(), used by HeightCompression (height_compression.py:20-24):
 * out (B,C,D,H,W) zero on entry. */
ORC_API void orc_sparse_to_dense(const float *feats, const int *coords, int n, int C, const int *shape, float *out) {
    size_t vol = (size_t)shape[0] * shape[1] * shape[2];
    for (int i = 0; i < n; ++i) {
        size_t sp = ((size_t)coords[i * 4 + 1] * shape[1] + coords[i * 4 + 2]) * shape[2] + coords[i * 4 + 3];
        for (int c = 0; c < C; ++c) out[((size_t)coords[i * 4] * C + c) * vol + sp] = feats[(size_t)i * C + c];
    }
}

/* round-to-nearest-even f32 -> bf16 -> f32, to emulate the bf16 storage of the MFMA path. */
ORC_API void orc_round_bf16(float *x, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t u;
        memcpy(&u, &x[i], 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) {
            u |= 0x00400000u; /* quiet NaN */
            u &= 0xffff0000u;
        } else {
            u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        }
        memcpy(&x[i], &u, 4);
    }
}
