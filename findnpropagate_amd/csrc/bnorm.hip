// BatchNorm1d in TRAINING mode on the rows of a sparse tensor, fused with the ReLU and the residual add that follow it
// in post_act_block / SparseBasicBlock (pcdet/models/backbones_3d/spconv_backbone.py:8-27,51-67), forward and backward:
// the dense part of the self-training step's backbone (tools/train_st.py; BASELINE.json configs[4]).
//
//   forward :  mean_c, var_c over the n valid rows (biased), y = act((x - mean) * invstd * gamma + beta [+ residual]),
//              running_mean / running_var updated like torch (momentum, unbiased variance)
//   backward:  g = dy * [y > 0],  dbeta = sum g,  dgamma = sum g * xhat,
//              dx = gamma * invstd * (g - dbeta / n - xhat * dgamma / n),  dresidual = g
//
// The reference runs this as separate torch kernels per op on (N, C) row tensors with dtype casts in between (bf16 / fp16
// activations): ~11 of the 20.5 ms of a 16-scene training step.  Here each direction is two passes over the rows:
// a statistics pass (per-thread f64 sums of 8 channels -> LDS tree -> per-workgroup partials -> one finishing workgroup
// that adds the partials in workgroup order) and an elementwise pass.  No atomics: results are bit-reproducible.
// Row count lives in device memory (`n_rows`), so nothing synchronises with the host.
#include "common.h"

namespace {

constexpr int kThreads = 256;
#ifndef FNP_BN_PARTS
#define FNP_BN_PARTS 256
#endif
constexpr int kMaxParts = FNP_BN_PARTS;   // statistics workgroups (one per CU; 1024 measured: the passes 7 % faster, the finishing launches
                                          // 2.4x slower — 7.2 -> 17.5 us for 42 of them per step)
#ifndef FNP_BN_UNROLL
#define FNP_BN_UNROLL 4
#endif
constexpr int kFinC = 16;        // channels per finishing workgroup

__device__ __forceinline__ float ldf(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const __bf16 *p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ float ldf(const _Float16 *p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ void stf(float *p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(__bf16 *p, size_t i, float v) { p[i] = (__bf16)v; }
__device__ __forceinline__ void stf(_Float16 *p, size_t i, float v) { p[i] = (_Float16)v; }

// 8 consecutive channels of one row as floats
template <typename T> __device__ __forceinline__ void load8(const T *p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(p);
        const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)e[j];
    } else {
        const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
template <typename T> __device__ __forceinline__ void store8(T *p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        uint4 raw;
        T *e = reinterpret_cast<T *>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (T)v[j];
        *reinterpret_cast<uint4 *>(p) = raw;
    } else {
        reinterpret_cast<float4 *>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4 *>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// Statistics pass.  MODE 0 (forward): a = x, b = x * x.  MODE 1 (backward): a = g, b = g * xhat.
// part[(wg * C + c) * 2 + {0, 1}] = the workgroup's sums for channel c (f64).
// Round 5: kStatThreads = 512 threads per workgroup instead of 256 (one workgroup per CU either way — the number of partials the
// finishing launch adds stays 256 — but 8 waves per CU with U x 1-3 16-byte loads in flight each instead of 4), and the sums of a
// wave meet through lane shuffles (an xor tree over the lanes that hold the same channels: a fixed order) before ONE line per wave
// goes to LDS.  Same-process A/B (tools/ab_bnorm.py, forward / backward of one layer = statistics + finish + apply, bf16 rows of the
// four stages at 16 scenes): 22.8 / 44.4 / 51.0 / 44.0 -> 21.5 / 39.5 / 43.2 / 38.4 us forward, 29.5 / 62.4 / 71.1 / 60.8 -> 27.1 /
// 55.8 / 63.0 / 54.0 us backward, every output bit-identical (1,024 threads: the 16-channel stage slower than before, the rest
// as 512).
#ifndef FNP_BN_STAT_THREADS
#define FNP_BN_STAT_THREADS 512
#endif
constexpr int kStatThreads = FNP_BN_STAT_THREADS;
// whole waves; 32 lanes per row at most (C <= 256) must fit a wave-aligned workgroup; red[NW][32][16] f64 is 4 KB per wave of
// static LDS: 1,024 threads sit exactly on the 64 KB limit
static_assert(kStatThreads % 64 == 0 && kStatThreads >= 64 && kStatThreads <= 1024, "FNP_BN_STAT_THREADS: a multiple of 64 in [64, 1024]");
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}
template <typename T, int MODE>
__global__ __launch_bounds__(kStatThreads) void bn_stats_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                                const T *__restrict__ y, const int *__restrict__ n_rows, int cap,
                                                                int C, const float *__restrict__ mean,
                                                                const float *__restrict__ invstd, int relu,
                                                                double *__restrict__ part) {
    constexpr int NW = kStatThreads / 64;
    __shared__ double red[NW][32][16];          // [wave][channel group (<= 32: C <= 256)][8 sums a, 8 sums b]
    const int n = min(*n_rows, cap);
    const int tpr = C / 8;                     // threads per row (a power of two, <= 32)
    const int rows_per_iter = kStatThreads / tpr;
    const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    double sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sa[j] = sb[j] = 0.0;
    float mu[8], is[8];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mu[j] = mean[cg * 8 + j];
            is[j] = invstd[cg * 8 + j];
        }
    }
    // contiguous row range per workgroup (so that the partial order is a function of n and the grid only)
    const long long per = ((long long)n + gridDim.x - 1) / gridDim.x;
    const long long r0 = per * blockIdx.x, r1 = min((long long)n, r0 + per);
    // U rows per trip: their loads (U, 2 U or 3 U of 16 bytes) are all requested before the first is used; the sums stay in row order
    constexpr int U = MODE == 1 ? (FNP_BN_UNROLL + 1) / 2 : FNP_BN_UNROLL;   // (backward: three tensors per row, 128 registers at 16 waves per CU)
    for (long long r = r0 + rl; r < r1; r += (long long)U * rows_per_iter) {
        float xv[U][8], gv[U][8], yv[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long ru = r + (long long)u * rows_per_iter;
            if (ru < r1) {
                load8(x + (size_t)ru * C + cg * 8, xv[u]);
                if (MODE == 1) {
                    load8(dy + (size_t)ru * C + cg * 8, gv[u]);
                    if (relu) load8(y + (size_t)ru * C + cg * 8, yv[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long ru = r + (long long)u * rows_per_iter;
            if (ru >= r1) break;
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sa[j] += (double)xv[u][j];
                    sb[j] += (double)xv[u][j] * (double)xv[u][j];
                }
            } else {
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) gv[u][j] = yv[u][j] > 0.f ? gv[u][j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (xv[u][j] - mu[j]) * is[j];
                    sa[j] += (double)gv[u][j];
                    sb[j] += (double)gv[u][j] * (double)xh;
                }
            }
        }
    }
    // lanes l, l + tpr, l + 2 tpr, ... of a wave hold the same channels: xor tree over them (tpr <= 32 divides 64)
    for (int m = 32; m >= tpr; m >>= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sa[j] += shfl_xor_f64(sa[j], m);
            sb[j] += shfl_xor_f64(sb[j], m);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < tpr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[wave][lane][j] = sa[j];
            red[wave][lane][8 + j] = sb[j];
        }
    }
    __syncthreads();
    // slot (channel group g, sum j of 16) adds the waves' lines in wave order; a strided loop over the tpr * 16 <= 512 slots, so that
    // any workgroup size of the FNP_BN_STAT_THREADS switch writes every partial (with `threadIdx.x < tpr * 16` a
    // 256-thread build left the partials of channel groups 16-31 of a 256-channel layer unwritten)
    for (int slot = threadIdx.x; slot < tpr * 16; slot += kStatThreads) {
        const int g = slot >> 4, j = slot & 15;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][g][j];
        part[((size_t)blockIdx.x * C + g * 8 + (j & 7)) * 2 + (j >> 3)] = v;
    }
}

// A workgroup per 16 channels adds the partials: thread (slice, c) sums the partials p = slice, slice + S, ... of channel c
// (S = 16 slices), the slices meet in an LDS tree — a fixed order, whatever the timing.  Forward: mean / invstd / running
// statistics.  Backward: dbeta = sum g, dgamma = sum g * xhat.
__global__ __launch_bounds__(kThreads) void bn_finish_kernel(const double *__restrict__ part, int parts, int C,
                                                             const int *__restrict__ n_rows, int cap, int mode, float eps,
                                                             float momentum, float *__restrict__ out_a,
                                                             float *__restrict__ out_b, float *__restrict__ running_mean,
                                                             float *__restrict__ running_var, long long *__restrict__ batches_tracked) {
    if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;   // (nn.BatchNorm1d.num_batches_tracked: a torch launch less per layer)
    __shared__ double ra[kThreads], rb[kThreads];
    const int n = min(*n_rows, cap);
    // a workgroup finishes kFinC channels (grid = C / kFinC workgroups; C is a power of two >= 8)
    const int CW = C < kFinC ? C : kFinC, S = kThreads / CW;
    const int cl = threadIdx.x % CW, slice = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    double a = 0.0, b = 0.0;
    for (int p = slice; p < parts; p += S) {
        a += part[((size_t)p * C + c) * 2];
        b += part[((size_t)p * C + c) * 2 + 1];
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int s = S / 2; s > 0; s >>= 1) {
        if (slice < s) {
            ra[threadIdx.x] += ra[threadIdx.x + s * CW];
            rb[threadIdx.x] += rb[threadIdx.x + s * CW];
        }
        __syncthreads();
    }
    if (slice == 0) {
        a = ra[cl];
        b = rb[cl];
        if (mode == 0) {
            const double m = n > 0 ? a / n : 0.0;
            double var = n > 0 ? b / n - m * m : 0.0;
            if (var < 0.0) var = 0.0;
            out_a[c] = (float)m;
            out_b[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean && n > 0) {
                const double unb = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
                running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
                running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
            }
        } else {
            out_a[c] = (float)a;   // dbeta
            out_b[c] = (float)b;   // dgamma
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const T *__restrict__ x, const T *__restrict__ residual,
                                                            const int *__restrict__ n_rows, int cap, int C,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            int relu, T *__restrict__ y) {
    const int n = min(*n_rows, cap);
    const long long chunks = (long long)n * (C / 8);
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < chunks; i += (long long)gridDim.x * kThreads) {
        const int c0 = (int)(i % (C / 8)) * 8;
        float xv[8], rv[8], o[8];
        load8(x + i * 8, xv);
        if (residual) load8(residual + i * 8, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = (xv[j] - mean[c0 + j]) * invstd[c0 + j] * gamma[c0 + j] + beta[c0 + j];
            if (residual) v = v + rv[j];
            if (relu && v < 0.f) v = 0.f;
            o[j] = v;
        }
        store8(y + i * 8, o);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bn_backward_apply_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                                     const T *__restrict__ y,
                                                                     const int *__restrict__ n_rows, int cap, int C,
                                                                     const float *__restrict__ mean,
                                                                     const float *__restrict__ invstd,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ dbeta,
                                                                     const float *__restrict__ dgamma, int relu,
                                                                     T *__restrict__ dx, T *__restrict__ dres) {
    const int n = min(*n_rows, cap);
    const float inv_n = n > 0 ? 1.0f / (float)n : 0.f;
    const long long chunks = (long long)n * (C / 8);
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < chunks; i += (long long)gridDim.x * kThreads) {
        const int c0 = (int)(i % (C / 8)) * 8;
        float gv[8], xv[8], yv[8], o[8];
        load8(dy + i * 8, gv);
        load8(x + i * 8, xv);
        if (relu) {
            load8(y + i * 8, yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - mean[c0 + j]) * invstd[c0 + j];
            o[j] = gamma[c0 + j] * invstd[c0 + j] * (gv[j] - dbeta[c0 + j] * inv_n - xh * dgamma[c0 + j] * inv_n);
        }
        store8(dx + i * 8, o);
        if (dres) store8(dres + i * 8, gv);
    }
}

int stats_grid(int cap, int C) {
    const int rows_per_iter = kStatThreads / (C / 8);
    int g = fnp_divup(cap, rows_per_iter * 8);
    if (g > kMaxParts) g = kMaxParts;
    if (g < 1) g = 1;
    return g;
}

template <typename T>
int run_forward(const void *x, const int *n_rows, int cap, int C, const float *gamma, const float *beta, float *rm, float *rv,
                float momentum, float eps, const void *residual, int relu, void *y, float *save_mean, float *save_invstd,
                void *ws, hipStream_t s, long long *nbt) {
    const int g = stats_grid(cap, C);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_stats_kernel<T, 0>), dim3(g), dim3(kStatThreads), 0, s,
                       (const T *)x, (const T *)nullptr, (const T *)nullptr, n_rows, cap, C, (const float *)nullptr,
                       (const float *)nullptr, 0, (double *)ws);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finish_kernel, dim3(C <= kFinC ? 1 : C / kFinC), dim3(kThreads), 0, s, (const double *)ws, g, C, n_rows, cap, 0, eps, momentum,
                       save_mean, save_invstd, rm, rv, nbt);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_apply_kernel<T>), dim3(fnp_grid_for((long long)cap * (C / 8), kThreads, 2048)), dim3(kThreads), 0,
                       s, (const T *)x, (const T *)residual, n_rows, cap, C, save_mean, save_invstd, gamma, beta, relu, (T *)y);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

template <typename T>
int run_backward(const void *dy, const void *x, const void *y, const int *n_rows, int cap, int C, const float *gamma,
                 const float *save_mean, const float *save_invstd, int relu, void *dx, void *dres, float *dgamma, float *dbeta,
                 void *ws, hipStream_t s) {
    const int g = stats_grid(cap, C);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_stats_kernel<T, 1>), dim3(g), dim3(kStatThreads), 0, s,
                       (const T *)x, (const T *)dy, (const T *)y, n_rows, cap, C, save_mean, save_invstd, relu, (double *)ws);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finish_kernel, dim3(C <= kFinC ? 1 : C / kFinC), dim3(kThreads), 0, s, (const double *)ws, g, C, n_rows, cap, 1, 0.f, 0.f, dbeta,
                       dgamma, (float *)nullptr, (float *)nullptr, (long long *)nullptr);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_backward_apply_kernel<T>), dim3(fnp_grid_for((long long)cap * (C / 8), kThreads, 2048)),
                       dim3(kThreads), 0, s, (const T *)dy, (const T *)x, (const T *)y, n_rows, cap, C, save_mean, save_invstd, gamma,
                       dbeta, dgamma, relu, (T *)dx, (T *)dres);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

bool shape_ok(int cap, int C) { return cap > 0 && C >= 8 && C <= 256 && (C & (C - 1)) == 0; }   // 8 | C, C/8 | 256

}  // namespace

extern "C" int64_t fnp_bn_workspace_bytes(int C) { return C > 0 ? (int64_t)kMaxParts * C * 2 * sizeof(double) : 0; }

extern "C" int fnp_bn_train_forward(const void *x, int dtype, const int *n_rows, int cap, int C, const float *gamma,
                                    const float *beta, float *running_mean, float *running_var, float momentum, float eps,
                                    const void *residual, int relu, void *y, float *save_mean, float *save_invstd,
                                    long long *num_batches_tracked, void *workspace, int64_t workspace_bytes, fnp_stream_t stream) {
    if (!x || !n_rows || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || !shape_ok(cap, C)) return FNP_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return FNP_ERR_ARG;
    if (workspace_bytes < fnp_bn_workspace_bytes(C)) return FNP_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FNP_F32)
        return run_forward<float>(x, n_rows, cap, C, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, y,
                                  save_mean, save_invstd, workspace, s, num_batches_tracked);
    if (dtype == FNP_BF16)
        return run_forward<__bf16>(x, n_rows, cap, C, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, y,
                                   save_mean, save_invstd, workspace, s, num_batches_tracked);
    if (dtype == FNP_F16)
        return run_forward<_Float16>(x, n_rows, cap, C, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, y,
                                     save_mean, save_invstd, workspace, s, num_batches_tracked);
    return FNP_ERR_ARG;
}

extern "C" int fnp_bn_train_backward(const void *grad_out, const void *x, const void *y, int dtype, const int *n_rows, int cap,
                                     int C, const float *gamma, const float *save_mean, const float *save_invstd, int relu,
                                     void *grad_x, void *grad_residual, float *grad_gamma, float *grad_beta, void *workspace,
                                     int64_t workspace_bytes, fnp_stream_t stream) {
    if (!grad_out || !x || !n_rows || !gamma || !save_mean || !save_invstd || !grad_x || !grad_gamma || !grad_beta || !workspace ||
        !shape_ok(cap, C) || (relu && !y))
        return FNP_ERR_ARG;
    if (workspace_bytes < fnp_bn_workspace_bytes(C)) return FNP_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FNP_F32)
        return run_backward<float>(grad_out, x, y, n_rows, cap, C, gamma, save_mean, save_invstd, relu, grad_x, grad_residual,
                                   grad_gamma, grad_beta, workspace, s);
    if (dtype == FNP_BF16)
        return run_backward<__bf16>(grad_out, x, y, n_rows, cap, C, gamma, save_mean, save_invstd, relu, grad_x, grad_residual,
                                    grad_gamma, grad_beta, workspace, s);
    if (dtype == FNP_F16)
        return run_backward<_Float16>(grad_out, x, y, n_rows, cap, C, gamma, save_mean, save_invstd, relu, grad_x, grad_residual,
                                      grad_gamma, grad_beta, workspace, s);
    return FNP_ERR_ARG;
}
