// Sparse convolution forward for gfx950: output-stationary implicit GEMM over the rulebook
// nbr[k][o] (rulebook.hip) with the BatchNorm1d(eval) / residual / ReLU epilogue of
// SparseBasicBlock and post_act_block fused in
// (pcdet/models/backbones_3d/spconv_backbone.py:8-27,51-67).  Replaces spconv's
// SubMConv3d / SparseConv3d forward (gather -> GEMM -> scatter-add).
//
//   out[o, :] = act( (sum_k W_k^T x[nbr[k][o], :]) * scale + shift + residual[o, :] )
//
// Two code paths:
//   * bf16 features/weights, fp32 accumulate on MFMA (v_mfma_f32_16x16x32_bf16).  The product is
//     formed transposed, D = W_k^T (A operand, 16 out-channels x 32 in-channels) times
//     X^T (B operand, 32 in-channels x 16 sites): both fragments are 16 contiguous bytes per
//     lane straight from HBM/L2 (weights are pre-packed [K][Cout][Cin]; a gathered feature row
//     is contiguous in Cin), so no LDS transpose is needed, and each lane ends up with 4
//     consecutive output channels of one site -> 8-byte bf16 stores.  One wave owns MB*16 sites
//     and all Cout; absent neighbours are exec-masked zero fragments; a kernel offset none of
//     the wave's sites uses is skipped with one ballot.
//   * f32 validation path on the VALU: a k-ascending, cin-ascending fmaf chain per output
//     element, the same chain the CPU oracle evaluates, so it is bit-comparable.
// No atomics anywhere: every output row is written once, results are run-to-run identical.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// ------------------------------------------------------------------------------------------
// VALU path (any Cin/Cout, any dtype mix): thread per (row, cout).
// ------------------------------------------------------------------------------------------
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void spconv_valu_kernel(const TIn *__restrict__ x, const TIn *__restrict__ w,
                                                          const int *__restrict__ nbr, int nbr_stride, int K,
                                                          const int *__restrict__ n_out, int cap,
                                                          TOut *__restrict__ y, const float *__restrict__ scale,
                                                          const float *__restrict__ shift,
                                                          const TOut *__restrict__ residual, int relu, int Cin, int Cout) {
    const int n = min(*n_out, cap);
    const long long total = (long long)n * Cout;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / Cout), co = (int)(t % Cout);
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const int idx = nbr[(size_t)k * nbr_stride + row];
            if (idx < 0) continue;
            const TIn *xr = x + (size_t)idx * Cin;
            const TIn *wr = w + ((size_t)k * Cout + co) * Cin;
            for (int ci = 0; ci < Cin; ++ci) acc = fmaf(to_f32(xr[ci]), to_f32(wr[ci]), acc);
        }
        float v = acc;
        if (scale) v = v * scale[co] + shift[co];
        if (residual) v = v + to_f32(residual[(size_t)row * Cout + co]);
        if (relu && v < 0.f) v = 0.f;
        y[(size_t)row * Cout + co] = from_f32<TOut>(v);
    }
}

// ------------------------------------------------------------------------------------------
// MFMA path.
// ------------------------------------------------------------------------------------------
// KVOL: kernel volume known at compile time (27 for the 3x3x3 layers) or 0 = runtime K.  Besides
// letting the compiler unroll, it gives the 3x3x3 layers and conv_out (K = 3) distinct kernel
// names, so profiler statistics per kernel name are per layer class.
template <int CIN, int COUT, int MB, int KVOL, typename TOut>
__global__ __launch_bounds__(256) void spconv_mfma_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                          const int *__restrict__ nbr, int nbr_stride, int Krt,
                                                          const int *__restrict__ n_out, int cap,
                                                          TOut *__restrict__ y, const float *__restrict__ scale,
                                                          const float *__restrict__ shift,
                                                          const TOut *__restrict__ residual, int relu) {
    constexpr int KS = (CIN + 31) / 32;   // 32-wide K steps of the MFMA
    constexpr int NB = COUT / 16;         // 16-channel output blocks
    constexpr int ROWS_PER_WAVE = MB * 16;
    constexpr int ROWS_PER_WG = 4 * ROWS_PER_WAVE;
    static_assert(CIN % 16 == 0 && COUT % 16 == 0, "channel counts must be multiples of 16");

    const int K = KVOL > 0 ? KVOL : Krt;
    const int n = min(*n_out, cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int tiles = (n + ROWS_PER_WG - 1) / ROWS_PER_WG;
    const bool kvalid0 = (q * 8) < CIN;  // for CIN == 16 only lanes 0..31 carry data in a K step

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int row0 = tile * ROWS_PER_WG + wave * ROWS_PER_WAVE;
        if (row0 >= n) continue;  // wave-uniform
        f32x4 acc[NB][MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int k = 0; k < K; ++k) {
            int idx[MB];
            bool any = false;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int r = row0 + mb * 16 + l15;
                idx[mb] = r < n ? nbr[(size_t)k * nbr_stride + r] : -1;
                any = any || idx[mb] >= 0;
            }
            if (!__any(any)) continue;
            const __bf16 *wk = w + (size_t)k * COUT * CIN;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bool kvalid = KS > 1 ? true : kvalid0;
                const int coff = ks * 32 + q * 8;
                bf16x8 xb[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (idx[mb] >= 0 && kvalid) v = *reinterpret_cast<const bf16x8 *>(x + (size_t)idx[mb] * CIN + coff);
                    xb[mb] = v;
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    bf16x8 wa = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (kvalid) wa = *reinterpret_cast<const bf16x8 *>(wk + (size_t)(nb * 16 + l15) * CIN + coff);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb[mb], acc[nb][mb], 0, 0, 0);
                }
            }
        }

        // epilogue: lane holds out[site = row0 + mb*16 + l15][c0 .. c0+3], c0 = nb*16 + q*4
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int c0 = nb * 16 + q * 4;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if (scale) {
                const float4 s4 = *reinterpret_cast<const float4 *>(scale + c0);
                const float4 h4 = *reinterpret_cast<const float4 *>(shift + c0);
                sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
                sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int r = row0 + mb * 16 + l15;
                if (r >= n) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = scale ? acc[nb][mb][j] * sc[j] + sh[j] : acc[nb][mb][j];
                if (residual) {
                    const TOut *rp = residual + (size_t)r * COUT + c0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] + to_f32(rp[j]);
                }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                }
                TOut *yp = y + (size_t)r * COUT + c0;
                if constexpr (sizeof(TOut) == 2) {
                    bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4 *>(yp) = o;
                } else {
                    *reinterpret_cast<float4 *>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

template <int CIN, int COUT, typename TOut>
int launch_mfma(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap, void *y,
                const float *scale, const float *shift, const void *residual, int relu, hipStream_t s) {
    constexpr int MB = 4;
    const int tiles = fnp_divup(cap, 4 * MB * 16);
    const int grid = tiles < 256 * 6 ? tiles : 256 * 6;
    if (K == 27)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_mfma_kernel<CIN, COUT, MB, 27, TOut>), dim3(grid), dim3(256), 0, s,
                           (const __bf16 *)x, (const __bf16 *)w, nbr, nbr_stride, K, n_out, cap, (TOut *)y, scale,
                           shift, (const TOut *)residual, relu);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_mfma_kernel<CIN, COUT, MB, 0, TOut>), dim3(grid), dim3(256), 0, s,
                           (const __bf16 *)x, (const __bf16 *)w, nbr, nbr_stride, K, n_out, cap, (TOut *)y, scale,
                           shift, (const TOut *)residual, relu);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

template <typename TIn, typename TOut>
int launch_valu(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap, void *y,
                const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                hipStream_t s) {
    const int grid = fnp_grid_for((long long)cap * Cout, 256, 256 * 16);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(spconv_valu_kernel<TIn, TOut>), dim3(grid), dim3(256), 0, s, (const TIn *)x,
                       (const TIn *)w, nbr, nbr_stride, K, n_out, cap, (TOut *)y, scale, shift, (const TOut *)residual,
                       relu, Cin, Cout);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

template <typename TOut>
int dispatch_bf16(const void *x, const void *w, const int *nbr, int nbr_stride, int K, const int *n_out, int cap,
                  void *y, const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                  hipStream_t s) {
#define FNP_CASE(CI, CO)                                                                                       \
    if (Cin == CI && Cout == CO)                                                                               \
        return launch_mfma<CI, CO, TOut>(x, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual, relu, s);
    FNP_CASE(16, 16)
    FNP_CASE(16, 32)
    FNP_CASE(32, 32)
    FNP_CASE(32, 64)
    FNP_CASE(64, 64)
    FNP_CASE(64, 128)
    FNP_CASE(128, 128)
#undef FNP_CASE
    return launch_valu<__bf16, TOut>(x, w, nbr, nbr_stride, K, n_out, cap, y, scale, shift, residual, relu, Cin, Cout, s);
}

// SparseConvTensor.dense(): thread per (row, channel).
template <typename T>
__global__ __launch_bounds__(256) void dense_kernel(const T *__restrict__ feats, const int *__restrict__ coords,
                                                    const int *__restrict__ n_rows, int cap, int C, int D, int H, int W,
                                                    T *__restrict__ out) {
    const int n = min(*n_rows, cap);
    const long long total = (long long)n * C;
    const long long vol = (long long)D * H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(t / C), c = (int)(t % C);
        const int4 cd = reinterpret_cast<const int4 *>(coords)[row];
        const long long sp = ((long long)cd.y * H + cd.z) * W + cd.w;
        out[((long long)cd.x * C + c) * vol + sp] = feats[t];
    }
}

}  // namespace

extern "C" int fnp_spconv_forward(const void *feat_in, int in_dtype, const void *weight, const int *nbr, int nbr_stride,
                                  int K, const int *n_out, int cap_out, void *feat_out, int out_dtype,
                                  const float *scale, const float *shift, const void *residual, int relu, int Cin,
                                  int Cout, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat_in || !weight || !nbr || !n_out || !feat_out || K <= 0 || Cin <= 0 || Cout <= 0 || cap_out <= 0 ||
        nbr_stride < cap_out)
        return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return FNP_ERR_ARG;
    if (in_dtype == FNP_F32) {
        if (out_dtype == FNP_F32)
            return launch_valu<float, float>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                             residual, relu, Cin, Cout, s);
        if (out_dtype == FNP_BF16)
            return launch_valu<float, __bf16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale,
                                              shift, residual, relu, Cin, Cout, s);
        return FNP_ERR_ARG;
    }
    if (in_dtype == FNP_BF16) {
        if (out_dtype == FNP_BF16)
            return dispatch_bf16<__bf16>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                         residual, relu, Cin, Cout, s);
        if (out_dtype == FNP_F32)
            return dispatch_bf16<float>(feat_in, weight, nbr, nbr_stride, K, n_out, cap_out, feat_out, scale, shift,
                                        residual, relu, Cin, Cout, s);
        return FNP_ERR_ARG;
    }
    return FNP_ERR_ARG;
}

extern "C" int fnp_sparse_to_dense(const void *feats, int dtype, const int *coords, const int *n_rows, int cap, int C,
                                   int B, int D, int H, int W, void *out, fnp_stream_t stream) {
    if (!feats || !coords || !n_rows || !out || cap <= 0 || C <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0)
        return FNP_ERR_ARG;
    const int grid = fnp_grid_for((long long)cap * C, 256, 256 * 16);
    if (dtype == FNP_F32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float *)feats, coords, n_rows, cap, C, D, H, W, (float *)out);
    else if (dtype == FNP_BF16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_kernel<__bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16 *)feats, coords, n_rows, cap, C, D, H, W, (__bf16 *)out);
    else
        return FNP_ERR_ARG;
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
