// Point-in-rotated-box tests for the Greedy Box Seeker (gfx950).
//
// Semantics follow the reference device function check_pt_in_box3d
// (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:23-36):
//   z:  out if  fabsf(z - cz) > dz / 2.0            (strict, compared in double)
//   xy: rotate (x-cx, y-cy) by -heading in f32, in if |lx| < dx/2.0 + MARGIN and
//       |ly| < dy/2.0 + MARGIN, sums evaluated in double with MARGIN a f32 constant.
// MARGIN = 1e-5f for the GPU op (kernel.cu:27), 1e-2f for the dense "cpu" op
// (roiaware_pool3d.cpp:131).
//
// MI355X design: the per-box trigonometry is hoisted out of the (point, box) loop — each
// workgroup stages a tile of boxes in LDS as {cx, cy, cz, half-extents(+margin) in double, cos,
// sin} once, then every lane streams its point against the tile.  The reference evaluates
// cos/sin per (point, box) pair.  Built with -ffp-contract=off so the f32 rotate is the same
// mul/mul/add sequence the CPU oracle evaluates.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kBoxTile = 128;

struct BoxLds {
    float cx, cy, cz, hz;   // hz = dz * 0.5f (exact), z test: |z-cz| > hz
    float cosa, sina;       // cos(-heading), sin(-heading)
    double hx, hy;          // dx/2.0 + MARGIN, dy/2.0 + MARGIN in double
};

__device__ __forceinline__ void load_box(const float *b, float margin, BoxLds &o) {
    o.cx = b[0];
    o.cy = b[1];
    o.cz = b[2];
    o.hz = b[5] * 0.5f;
    o.hx = (double)b[3] / 2.0 + (double)margin;
    o.hy = (double)b[4] / 2.0 + (double)margin;
    const float a = -b[6];
    o.cosa = cosf(a);
    o.sina = sinf(a);
}

__device__ __forceinline__ bool pt_in_box(float x, float y, float z, const BoxLds &b) {
    if (fabsf(z - b.cz) > b.hz) return false;
    const float sx = x - b.cx, sy = y - b.cy;
    const float lx = sx * b.cosa + sy * (-b.sina);
    const float ly = sx * b.sina + sy * b.cosa;
    return ((double)fabsf(lx) < b.hx) && ((double)fabsf(ly) < b.hy);
}

// grid (ceil(M/256), B): first containing box per point.
__global__ __launch_bounds__(kThreads) void points_in_boxes_first_kernel(
    const float *__restrict__ boxes, const float *__restrict__ pts, int *__restrict__ out, int T, int M) {
    __shared__ BoxLds tile[kBoxTile];
    const int b = blockIdx.y;
    const int m = blockIdx.x * kThreads + threadIdx.x;
    const float *bx = boxes + (size_t)b * T * 7;
    float x = 0.f, y = 0.f, z = 0.f;
    if (m < M) {
        const float *p = pts + ((size_t)b * M + m) * 3;
        x = p[0];
        y = p[1];
        z = p[2];
    }
    int found = -1;
    for (int t0 = 0; t0 < T; t0 += kBoxTile) {
        const int nt = min(kBoxTile, T - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += kThreads) load_box(bx + (size_t)(t0 + i) * 7, 1e-5f, tile[i]);
        __syncthreads();
        if (found < 0) {
            for (int i = 0; i < nt; ++i) {
                if (pt_in_box(x, y, z, tile[i])) {
                    found = t0 + i;
                    break;
                }
            }
        }
    }
    if (m < M) out[(size_t)b * M + m] = found;
}

// counts[t] = #points inside box t.  grid ceil(M/256); T boxes looped in LDS tiles; one
// wave ballot + popcount per (wave, box), LDS integer atomics, one global atomic per (block, box).
__global__ __launch_bounds__(kThreads) void points_in_boxes_count_kernel(
    const float *__restrict__ boxes, const float *__restrict__ pts, int *__restrict__ counts, int T, int M) {
    __shared__ BoxLds tile[kBoxTile];
    __shared__ int lcount[kBoxTile];
    const int m = blockIdx.x * kThreads + threadIdx.x;
    const bool valid = m < M;
    float x = 0.f, y = 0.f, z = 0.f;
    if (valid) {
        x = pts[(size_t)m * 3 + 0];
        y = pts[(size_t)m * 3 + 1];
        z = pts[(size_t)m * 3 + 2];
    }
    for (int t0 = 0; t0 < T; t0 += kBoxTile) {
        const int nt = min(kBoxTile, T - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += kThreads) {
            load_box(boxes + (size_t)(t0 + i) * 7, 1e-5f, tile[i]);
            lcount[i] = 0;
        }
        __syncthreads();
        for (int i = 0; i < nt; ++i) {
            const bool in = valid && pt_in_box(x, y, z, tile[i]);
            const unsigned long long mask = __ballot(in);
            if (fnp_lane() == 0 && mask) atomicAdd(&lcount[i], __popcll(mask));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += kThreads)
            if (lcount[i]) atomicAdd(&counts[t0 + i], lcount[i]);
    }
}

// (T,M) 0/1 matrix, CPU-variant margin.  grid (ceil(M/256), ceil(T/kBoxTile)).
__global__ __launch_bounds__(kThreads) void points_in_boxes_dense_kernel(
    const float *__restrict__ boxes, const float *__restrict__ pts, int *__restrict__ out, int T, int M) {
    __shared__ BoxLds tile[kBoxTile];
    const int t0 = blockIdx.y * kBoxTile;
    const int nt = min(kBoxTile, T - t0);
    for (int i = threadIdx.x; i < nt; i += kThreads) load_box(boxes + (size_t)(t0 + i) * 7, 1e-2f, tile[i]);
    __syncthreads();
    const int m = blockIdx.x * kThreads + threadIdx.x;
    if (m >= M) return;
    const float x = pts[(size_t)m * 3 + 0], y = pts[(size_t)m * 3 + 1], z = pts[(size_t)m * 3 + 2];
    for (int i = 0; i < nt; ++i) out[(size_t)(t0 + i) * M + m] = pt_in_box(x, y, z, tile[i]) ? 1 : 0;
}

}  // namespace

extern "C" int fnp_points_in_boxes(const float *boxes, const float *pts, int *box_idx_of_points,
                                   int B, int T, int M, fnp_stream_t stream) {
    if (B < 0 || T < 0 || M < 0) return FNP_ERR_ARG;
    if (B == 0 || M == 0) return FNP_OK;
    if (!pts || !box_idx_of_points || (T > 0 && !boxes)) return FNP_ERR_ARG;
    dim3 grid(fnp_divup(M, kThreads), B);
    hipLaunchKernelGGL(points_in_boxes_first_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream,
                       boxes, pts, box_idx_of_points, T, M);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_points_in_boxes_count(const float *boxes, const float *pts, int *counts,
                                         int T, int M, fnp_stream_t stream) {
    if (T < 0 || M < 0) return FNP_ERR_ARG;
    if (T == 0) return FNP_OK;
    if (!boxes || !counts || (M > 0 && !pts)) return FNP_ERR_ARG;
    FNP_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)T, (hipStream_t)stream));
    if (M == 0) return FNP_OK;
    hipLaunchKernelGGL(points_in_boxes_count_kernel, dim3(fnp_divup(M, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, boxes, pts, counts, T, M);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_points_in_boxes_dense(const float *boxes, const float *pts, int *pts_indices,
                                         int T, int M, fnp_stream_t stream) {
    if (T < 0 || M < 0) return FNP_ERR_ARG;
    if (T == 0 || M == 0) return FNP_OK;
    if (!boxes || !pts || !pts_indices) return FNP_ERR_ARG;
    dim3 grid(fnp_divup(M, kThreads), fnp_divup(T, kBoxTile));
    hipLaunchKernelGGL(points_in_boxes_dense_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream,
                       boxes, pts, pts_indices, T, M);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// ---- host entry point: PseudoSampler.points_in_boxes (pcdet/datasets/augmentor/pseudo_loader.py:270-316) ----
// Dense (T, N) membership of N points in T boxes for the copy-paste / pseudo-label mixing in dataloader
// workers, with the box-frame points (T, N, C) the reference returns alongside (centre subtracted, rotated
// by -heading: x' = x cos(-h) - y sin(-h), y' = x sin(-h) + y cos(-h); common_utils.py:35-57), faces
// INCLUSIVE (>= lower, <= upper: unlike the CUDA operator's strict test).  Plain host loops, no device.
#include <cmath>
extern "C" int fnp_host_points_in_boxes_frame(const float *points, int n, int C, const float *boxes, int t,
                                              unsigned char *in_box, float *points_out) {
    if (n < 0 || t < 0 || C < 3 || ((n > 0 && t > 0) && (!points || !boxes || !in_box))) return FNP_ERR_ARG;
    for (int b = 0; b < t; ++b) {
        const float *bx = boxes + (size_t)b * 7;
        const float ca = cosf(-bx[6]), sa = sinf(-bx[6]);
        // corner template +-1/2 of the extents: min/max per axis (pseudo_loader.py:275-290)
        const float hx = bx[3] * 0.5f, hy = bx[4] * 0.5f, hz = bx[5] * 0.5f;
        const float x1 = fminf(hx, -hx), x2 = fmaxf(hx, -hx), y1 = fminf(hy, -hy), y2 = fmaxf(hy, -hy);
        const float z1 = fminf(hz, -hz), z2 = fmaxf(hz, -hz);
        for (int i = 0; i < n; ++i) {
            const float *p = points + (size_t)i * C;
            const float px = p[0] - bx[0], py = p[1] - bx[1], pz = p[2] - bx[2];
            const float rx = px * ca + py * (-sa), ry = px * sa + py * ca;
            in_box[(size_t)b * n + i] = (rx >= x1 && rx <= x2 && ry >= y1 && ry <= y2 && pz >= z1 && pz <= z2) ? 1 : 0;
            if (points_out) {
                float *o = points_out + ((size_t)b * n + i) * C;
                o[0] = rx; o[1] = ry; o[2] = pz;
                for (int c = 3; c < C; ++c) o[c] = p[c];
            }
        }
    }
    return FNP_OK;
}
