// f32 rows -> (hi, lo) bf16 pairs with hi + lo == x to 2^-17 of |x| — the activation format of the fused engine's "bf16x3"
// precision (features and weights split in two bf16 terms, three v_mfma_f32_16x16x32_bf16 products per term pair, f32
// accumulate: the f32 result to ~1e-5 relative on the matrix pipe that is 16x the f32 one).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// one thread = 4 consecutive elements (16 bytes in, 8 + 8 bytes out); rows past *n_rows are not touched.
// ADD: x + float(t) (a bf16 tensor: the two cross terms of a bf16x3 convolution, which need bf16 precision only), then ReLU if
// asked, written back to y (f32) before the split.
template <bool ADD>
// (x and y are NOT __restrict__: fnp_split_bf16_add is called in place, y == x — each thread reads its float4 and then writes it)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *x, const __bf16 *__restrict__ t, int relu,
                                                         const int *__restrict__ n_rows, int cap_rows, int C, float *y,
                                                         __bf16 *__restrict__ hi, __bf16 *__restrict__ lo) {
    const long long total4 = (long long)min(*n_rows, cap_rows) * C / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        if constexpr (ADD) {
            const bf16x4 tv = reinterpret_cast<const bf16x4 *>(t)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = v[j] + (float)tv[j];
                if (relu) v[j] = v[j] < 0.f ? 0.f : v[j];
            }
            reinterpret_cast<f32x4 *>(y)[i] = v;
        }
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)v[j];
            l[j] = (__bf16)(v[j] - (float)h[j]);   // (exact difference: the f32 holds hi's 8 bits and 16 more)
        }
        reinterpret_cast<bf16x4 *>(hi)[i] = h;
        reinterpret_cast<bf16x4 *>(lo)[i] = l;
    }
}

}  // namespace

extern "C" int fnp_split_bf16(const float *x, const int *n_rows, int cap_rows, int C, void *hi, void *lo, fnp_stream_t stream) {
    if (!x || !n_rows || !hi || !lo || cap_rows <= 0 || C <= 0 || (C & 3)) return FNP_ERR_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7)) return FNP_ERR_ARG;
    const long long total4 = (long long)cap_rows * C / 4;
    hipLaunchKernelGGL(split_bf16_kernel<false>, dim3(fnp_grid_for(total4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, (const __bf16 *)nullptr, 0,
                       n_rows, cap_rows, C, (float *)nullptr, (__bf16 *)hi, (__bf16 *)lo);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_split_bf16_add(const float *x, const void *t, int relu, const int *n_rows, int cap_rows, int C, float *y, void *hi, void *lo,
                                  fnp_stream_t stream) {
    if (!x || !t || !y || !n_rows || !hi || !lo || cap_rows <= 0 || C <= 0 || (C & 3)) return FNP_ERR_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)t & 7) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7)) return FNP_ERR_ARG;
    const long long total4 = (long long)cap_rows * C / 4;
    hipLaunchKernelGGL(split_bf16_kernel<true>, dim3(fnp_grid_for(total4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, (const __bf16 *)t, relu,
                       n_rows, cap_rows, C, y, (__bf16 *)hi, (__bf16 *)lo);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
