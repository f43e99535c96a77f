// CLIP-crop scoring, geometry half (pcdet/models/dense_heads/clip_box_classification.py:68-379, the
// re-labelling stage of the Greedy Box Seeker pipeline).
//
// The reference walks cameras x boxes in Python: 8 corners -> project_to_camera (:187-228) -> integer
// pixel coordinates -> on-image test -> clipped bounding box (clip_coords :103-113) -> square crop of
// side max(w, h) >= 64 px anchored at (x1, y1) -> F.grid_sample of a 224 x 224 uniform grid (:297-334),
// one small kernel chain and several host round trips per (box, camera).  Here:
//   plan   : one thread per (box, camera) does the projection and the crop decision for all pairs;
//   sample : one workgroup per 8 output rows of a crop, bilinear taps with grid_sample's arithmetic
//            (align_corners = False, zeros padding), coalesced 224-wide output rows.
// The image encoder between them is the caller's (third-party CLIP), the softmax / per-camera mean /
// arg-max after it are a few tiny tensor ops on the host side.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

constexpr int kThreads = 256;

struct PlanMats {
    float rinv[9];      // inverse of the lidar augmentation rotation (row-major), computed by the caller
    float taug[3];      // lidar augmentation translation
};

// rect[(b*6 + c)*4 + ..] = x1, y1, side, valid;  cam_mask[b*6 + c] = box b is seen by camera c
__global__ __launch_bounds__(kThreads) void clipcrop_plan_kernel(const float *__restrict__ boxes, int n, PlanMats pm,
                                                                 const float *__restrict__ lidar2image,
                                                                 const float *__restrict__ img_aug, int H, int W, int min_crop,
                                                                 float *__restrict__ rect, unsigned char *__restrict__ cam_mask) {
    const int t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= n * 6) return;
    const int b = t / 6, c = t % 6;
    const float *bx = boxes + (size_t)b * 7;
    const float cs = cosf(bx[6]), sn = sinf(bx[6]);
    const float *L = lidar2image + c * 16, *A = img_aug + c * 16;
    bool any = false;
    long long xmin = 0, xmax = 0, ymin = 0, ymax = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // boxes_to_corners_3d (box_utils.py:28-53): template * dims, rotate about z, + centre
        const float tx = ((k & 3) == 0 || (k & 3) == 1) ? 0.5f : -0.5f;
        const float ty = ((k & 3) == 0 || (k & 3) == 3) ? 0.5f : -0.5f;
        const float tz = k < 4 ? -0.5f : 0.5f;
        const float lx = bx[3] * tx, ly = bx[4] * ty, lz = bx[5] * tz;
        float px = lx * cs + ly * (-sn) + bx[0];
        float py = lx * sn + ly * cs + bx[1];
        float pz = lz + bx[2];
        // project_to_camera (:187-228)
        px -= pm.taug[0]; py -= pm.taug[1]; pz -= pm.taug[2];
        const float ux = pm.rinv[0] * px + pm.rinv[1] * py + pm.rinv[2] * pz;
        const float uy = pm.rinv[3] * px + pm.rinv[4] * py + pm.rinv[5] * pz;
        const float uz = pm.rinv[6] * px + pm.rinv[7] * py + pm.rinv[8] * pz;
        float qx = L[0] * ux + L[1] * uy + L[2] * uz + L[3];
        float qy = L[4] * ux + L[5] * uy + L[6] * uz + L[7];
        float qz = L[8] * ux + L[9] * uy + L[10] * uz + L[11];
        qz = fminf(fmaxf(qz, 1e-5f), 1e5f);
        qx /= qz; qy /= qz;
        const float rx = A[0] * qx + A[1] * qy + A[2] * qz + A[3];
        const float ry = A[4] * qx + A[5] * qy + A[6] * qz + A[7];
        const float rz = A[8] * qx + A[9] * qy + A[10] * qz + A[11];
        const long long ix = (long long)rx, iy = (long long)ry;   // .long(): truncation toward zero
        any = any || (ix < W && ix >= 0 && iy < H && iy >= 0 && rz >= 0.01f);
        if (k == 0) { xmin = xmax = ix; ymin = ymax = iy; }
        else { xmin = min(xmin, ix); xmax = max(xmax, ix); ymin = min(ymin, iy); ymax = max(ymax, iy); }
    }
    float *r = rect + (size_t)t * 4;
    r[0] = r[1] = r[2] = r[3] = 0.f;
    cam_mask[t] = any ? 1 : 0;
    if (!any) return;
    const long long x1 = min(max(xmin, 0ll), (long long)W), x2 = min(max(xmax, 0ll), (long long)W);
    const long long y1 = min(max(ymin, 0ll), (long long)H), y2 = min(max(ymax, 0ll), (long long)H);
    const long long side = max(x2 - x1, y2 - y1);
    if (side < min_crop) return;
    r[0] = (float)x1; r[1] = (float)y1; r[2] = (float)side; r[3] = 1.f;
}

__device__ __forceinline__ float ld(const float *p) { return *p; }
__device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
__device__ __forceinline__ void st(float *p, float v) { *p = v; }
__device__ __forceinline__ void st(__half *p, float v) { *p = __float2half(v); }

// grid (ceil(S/8), M); pairs[m] = (box, cam); unit[j] = the [0,1] grid coordinate of output column/row j
template <typename T>
__global__ __launch_bounds__(kThreads) void clipcrop_sample_kernel(const T *__restrict__ images, int C, int H, int W,
                                                                   const float *__restrict__ rect, const int *__restrict__ pairs,
                                                                   const float *__restrict__ unit, int S, T *__restrict__ out) {
    const int m = blockIdx.y;
    const int b = pairs[2 * m], c = pairs[2 * m + 1];
    const float *r = rect + ((size_t)b * 6 + c) * 4;
    const float x1 = r[0], y1 = r[1], side = r[2];
    const T *img = images + (size_t)c * C * H * W;
    const int rows0 = blockIdx.x * 8;
    for (int e = threadIdx.x; e < 8 * S; e += kThreads) {
        const int i = rows0 + e / S, j = e % S;
        if (i >= S) continue;
        // current_grid * square_size + x1, normalised to [-1,1], un-normalised by grid_sample (align_corners=False)
        const float gx = unit[j] * side + x1, gy = unit[i] * side + y1;
        const float nx = (gx / (float)W) * 2.0f - 1.0f, ny = (gy / (float)H) * 2.0f - 1.0f;
        const float ix = ((nx + 1.f) * (float)W - 1.f) / 2.f, iy = ((ny + 1.f) * (float)H - 1.f) / 2.f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
        const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
        for (int ch = 0; ch < C; ++ch) {
            const T *p = img + (size_t)ch * H * W;
            float v = 0.f;
            if (vy0 && vx0) v += ld(p + (size_t)y0 * W + x0) * (wx0 * wy0);
            if (vy0 && vx1) v += ld(p + (size_t)y0 * W + x0 + 1) * (wx1 * wy0);
            if (vy1 && vx0) v += ld(p + (size_t)(y0 + 1) * W + x0) * (wx0 * wy1);
            if (vy1 && vx1) v += ld(p + (size_t)(y0 + 1) * W + x0 + 1) * (wx1 * wy1);
            st(out + (((size_t)m * C + ch) * S + i) * S + j, v);
        }
    }
}

}  // namespace

extern "C" int fnp_clipcrop_plan(const float *boxes, int n, const float *lidar_aug_rot_inv, const float *lidar_aug_trans,
                                 const float *lidar2image, const float *img_aug, int image_h, int image_w, int min_crop,
                                 float *rect, unsigned char *cam_mask, fnp_stream_t stream) {
    if (n < 0 || image_h <= 0 || image_w <= 0 || !lidar_aug_rot_inv || !lidar_aug_trans) return FNP_ERR_ARG;
    if (n == 0) return FNP_OK;
    if (!boxes || !lidar2image || !img_aug || !rect || !cam_mask) return FNP_ERR_ARG;
    PlanMats pm;
    for (int i = 0; i < 9; ++i) pm.rinv[i] = lidar_aug_rot_inv[i];   // HOST pointers: 12 floats by value
    for (int i = 0; i < 3; ++i) pm.taug[i] = lidar_aug_trans[i];
    hipLaunchKernelGGL(clipcrop_plan_kernel, dim3(fnp_divup(n * 6, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, boxes, n,
                       pm, lidar2image, img_aug, image_h, image_w, min_crop, rect, cam_mask);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_clipcrop_sample(const void *images, int dtype, int channels, int image_h, int image_w, const float *rect,
                                   const int *pairs, int num_pairs, const float *unit_grid, int out_size, void *crops,
                                   fnp_stream_t stream) {
    if (num_pairs < 0 || channels <= 0 || image_h <= 0 || image_w <= 0 || out_size <= 0) return FNP_ERR_ARG;
    if (num_pairs == 0) return FNP_OK;
    if (!images || !rect || !pairs || !unit_grid || !crops || num_pairs > 65535) return FNP_ERR_ARG;
    const dim3 grid(fnp_divup(out_size, 8), num_pairs);
    if (dtype == FNP_F32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(clipcrop_sample_kernel<float>), grid, dim3(kThreads), 0, (hipStream_t)stream,
                           (const float *)images, channels, image_h, image_w, rect, pairs, unit_grid, out_size, (float *)crops);
    else if (dtype == FNP_F16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(clipcrop_sample_kernel<__half>), grid, dim3(kThreads), 0, (hipStream_t)stream,
                           (const __half *)images, channels, image_h, image_w, rect, pairs, unit_grid, out_size, (__half *)crops);
    else
        return FNP_ERR_ARG;
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
