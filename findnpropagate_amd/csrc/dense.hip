// SparseConvTensor.dense() as consumed by HeightCompression (pcdet/models/backbones_2d/map_to_bev/
// height_compression.py:20-24: dense (B,C,D,H,W) then viewed as (B, C*D, H, W)): the largest single
// HBM write after the backbone (265 MB bf16 for 16 scenes of (128, 2, 180, 180)).
//
// A row-driven scatter writes C separate 2-byte elements per site, each into a different channel
// plane: 13.9 M sector-sized partial writes on top of the memset of the whole tensor.  Here the
// tensor is written ONCE, zeros included, in plane order:
//   1. index map: cell (b,z,y,x) -> row or -1 (4 bytes per cell, memset + one 4-byte scatter per row);
//   2. one workgroup per 64 consecutive cells of a (b,z) plane: the feature rows of the occupied
//      cells (~10 %) are copied whole (coalesced) into a padded LDS tile, then every wave writes,
//      channel plane by channel plane, 64 consecutive elements (one 128-byte bf16 segment per store);
//      a tile without sites only writes zeros.
// Without a workspace the scatter path is used and `out` must be zero on entry.
#include "common.h"
#include <cstdint>

namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ __launch_bounds__(kThreads) void dense_scatter_kernel(const T *__restrict__ feats, const int *__restrict__ coords,
                                                                 const int *__restrict__ n_rows, int cap, int C, int D, int H,
                                                                 int W, T *__restrict__ out) {
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

__global__ __launch_bounds__(kThreads) void dense_index_kernel(const int *__restrict__ coords, const int *__restrict__ n_rows,
                                                               int cap, int B, int D, int H, int W, int *__restrict__ index) {
    const int n = min(*n_rows, cap);
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < n; r += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[r];
        if (c.x < 0 || c.x >= B || c.y < 0 || c.y >= D || c.z < 0 || c.z >= H || c.w < 0 || c.w >= W) continue;
        index[(((long long)c.x * D + c.y) * H + c.z) * W + c.w] = r;
    }
}

// T: element type (2 or 4 bytes); row bytes must be a multiple of 4.  One tile = 64*VEC consecutive cells
// of a (b,z) plane; lane l owns cells VEC*l .. VEC*l+VEC-1 and stores them with one VEC*sizeof(T)-byte
// store per channel plane (a wave store covers 64*VEC*sizeof(T) contiguous bytes).
// LDS: kSlots rows of C*sizeof(T) + 4 bytes (the + 4 spreads the rows over the banks): the occupied
// cells of a tile (~10 %) get consecutive slots; a tile with more than kSlots of them is written in VEC
// passes of one cell per lane (element stores) — rare, and still each element exactly once.
constexpr int kSlots = 64;

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void dense_write_kernel(const T *__restrict__ feats, const int *__restrict__ index, int C,
                                                               int B, int D, long long plane, int tiles_per_plane,
                                                               T *__restrict__ out, const float *__restrict__ fill) {
    // fill (optional, per channel): the value of a cell without a row instead of 0 — the BEV map behind a convolution whose
    // epilogue (BatchNorm shift, ReLU) gives cells with no input in reach a constant per channel
    extern __shared__ __attribute__((aligned(16))) unsigned char fnp_dense_smem[];
    typedef int IVec __attribute__((ext_vector_type(VEC)));
    typedef T TVec __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = blockIdx.x;
    const long long bz = tile / tiles_per_plane;                 // = b*D + z
    const long long l0 = (tile % tiles_per_plane) * (64 * VEC);  // first cell of the tile inside the plane
    const int b = (int)(bz / D), z = (int)(bz % D);
    const bool in = l0 + (long long)VEC * lane < plane;          // (plane % VEC == 0: all of the lane's cells or none)
    int r[VEC];
    if (in) {
        if constexpr (VEC == 1) {
            r[0] = index[bz * plane + l0 + lane];
        } else {
            const IVec v = *reinterpret_cast<const IVec *>(index + bz * plane + l0 + (long long)VEC * lane);
#pragma unroll
            for (int j = 0; j < VEC; ++j) r[j] = v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) r[j] = -1;
    }
    unsigned long long occ[VEC];                                  // same in all four waves
    int total = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        occ[j] = __ballot(r[j] >= 0);
        total += __popcll(occ[j]);
    }
    const int row_bytes = C * (int)sizeof(T), stride = row_bytes + 4;
    T *o = out + ((long long)b * C * D + z) * plane + l0 + (long long)VEC * lane;   // channel c adds c * D * plane
    const long long cstep = (long long)D * plane;
    const unsigned long long below = (1ull << lane) - 1ull;

    // rows of the cells in mask m (taken from r[j]) -> slots first, first + 1, ...; wave w copies every 4th
    auto stage = [&](unsigned long long m, const int rj, int first) {
        int k = first;
        while (m) {
            const int x = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int rr = __shfl(rj, x);
            const int slot = k++;
            if ((slot & 3) != wave) continue;
            const unsigned *src = reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned char *>(feats) + (size_t)rr * row_bytes);
            unsigned *dst = reinterpret_cast<unsigned *>(fnp_dense_smem + slot * stride);
            for (int q = lane; q * 4 < row_bytes; q += 64) dst[q] = src[q];
        }
        return k;
    };

    if (total == 0) {
        if (!in) return;
        for (int c = wave; c < C; c += 4) {
            const T bg = fill ? (T)fill[c] : (T)0;
            TVec v;
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = bg;
            *reinterpret_cast<TVec *>(o + c * cstep) = v;
        }
    } else if (total <= kSlots) {
        int first = 0, myslot[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            myslot[j] = first + __popcll(occ[j] & below);
            first = stage(occ[j], r[j], first);
        }
        __syncthreads();
        if (!in) return;
        for (int c = wave; c < C; c += 4) {
            const T bg = fill ? (T)fill[c] : (T)0;
            TVec v;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                v[j] = ((occ[j] >> lane) & 1ull) ? reinterpret_cast<const T *>(fnp_dense_smem + myslot[j] * stride)[c] : bg;
            *reinterpret_cast<TVec *>(o + c * cstep) = v;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {   // (workgroup-uniform path: the barriers match)
            stage(occ[j], r[j], 0);
            __syncthreads();
            if (in) {
                const bool has = (occ[j] >> lane) & 1ull;
                const T *mine = reinterpret_cast<const T *>(fnp_dense_smem + __popcll(occ[j] & below) * stride);
                for (int c = wave; c < C; c += 4) o[c * cstep + j] = has ? mine[c] : (fill ? (T)fill[c] : (T)0);
            }
            __syncthreads();
        }
    }
}

// Cells per lane.  Measured on MI355X (16 scenes, 265 MB bf16): 2 -> 101 us (2.6 TB/s), 4 -> 120 us,
// 8 -> 296 us (tiles of 256+ cells overflow the 64 LDS slots near the ego vehicle and take the element-
// store path), 1 -> 134 us; memset + row scatter 264 us.
#ifndef FNP_DENSE_VEC
#define FNP_DENSE_VEC 2
#endif

template <typename T, int VEC>
int launch_dense_write(const void *feats, const int *index, int C, int B, int D, long long plane, void *out, const float *fill, hipStream_t s) {
    const int row_bytes = C * (int)sizeof(T);
    const int tiles_per_plane = (int)((plane + 64 * VEC - 1) / (64 * VEC));
    const long long tiles = (long long)B * D * tiles_per_plane;
    if (tiles > 0x7fffffffll) return FNP_ERR_ARG;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_write_kernel<T, VEC>), dim3((unsigned)tiles), dim3(kThreads), kSlots * (row_bytes + 4), s,
                       (const T *)feats, index, C, B, D, plane, tiles_per_plane, (T *)out, fill);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

template <typename T>
int run_dense(const void *feats, const int *coords, const int *n_rows, int cap, int C, int B, int D, int H, int W, void *out,
              void *ws, int64_t ws_bytes, hipStream_t s, const float *fill = nullptr) {
    const long long plane = (long long)H * W, cells = (long long)B * D * plane;
    const int row_bytes = C * (int)sizeof(T);
    const bool tiled = ws != nullptr && row_bytes % 4 == 0 && 64 * (row_bytes + 4) <= 64 * 1024;
    if (!tiled) {
        if (fill) return FNP_ERR_ARG;   // (a per-channel background needs the single-pass writer: workspace + a row size it takes)
        if (ws) {   // (workspace given: out may be dirty)
            const long long bytes = cells * C * (long long)sizeof(T);
            if (bytes % 4 == 0 && ((uintptr_t)out & 3) == 0) {
                const int frc = fnp_fill_words(out, bytes / 4, 0u, s);
                if (frc) return frc;
            } else {
                FNP_HIP_TRY(hipMemsetAsync(out, 0, (size_t)bytes, s));   // (odd sizes only: not a shape of the backbone)
            }
        }
        const int grid = fnp_grid_for((long long)cap * C, kThreads, 256 * 16);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_scatter_kernel<T>), dim3(grid), dim3(kThreads), 0, s, (const T *)feats, coords,
                           n_rows, cap, C, D, H, W, (T *)out);
        FNP_LAUNCH_CHECK();
        return FNP_OK;
    }
    if (ws_bytes < cells * 4) return FNP_ERR_WORKSPACE;
    int *index = (int *)ws;
    {
        const int frc = fnp_fill_words(index, cells, 0xffffffffu, s);
        if (frc) return frc;
    }
    hipLaunchKernelGGL(dense_index_kernel, dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0, s, coords, n_rows, cap, B, D, H,
                       W, index);
    FNP_LAUNCH_CHECK();
    // widest store (<= 16 bytes per lane) that the plane size and the buffer alignment allow
    const bool al = ((uintptr_t)out % 16 == 0) && ((uintptr_t)ws % 16 == 0);
    constexpr int kMaxVec = 16 / (int)sizeof(T) < FNP_DENSE_VEC ? 16 / (int)sizeof(T) : FNP_DENSE_VEC;
    if constexpr (kMaxVec >= 8) {
        if (al && plane % 8 == 0) return launch_dense_write<T, 8>(feats, index, C, B, D, plane, out, fill, s);
    }
    if constexpr (kMaxVec >= 4) {
        if (al && plane % 4 == 0) return launch_dense_write<T, 4>(feats, index, C, B, D, plane, out, fill, s);
    }
    if constexpr (kMaxVec >= 2) {
        if (al && plane % 2 == 0) return launch_dense_write<T, 2>(feats, index, C, B, D, plane, out, fill, s);
    }
    return launch_dense_write<T, 1>(feats, index, C, B, D, plane, out, fill, s);
}

}  // namespace

extern "C" int64_t fnp_sparse_to_dense_workspace_bytes(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)B * D * H * W * 4;
}

extern "C" int fnp_sparse_to_dense(const void *feats, int dtype, const int *coords, const int *n_rows, int cap, int C,
                                   int B, int D, int H, int W, void *out, void *workspace, int64_t workspace_bytes,
                                   fnp_stream_t stream) {
    if (!feats || !coords || !n_rows || !out || cap <= 0 || C <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0)
        return FNP_ERR_ARG;
    if (dtype == FNP_F32)
        return run_dense<float>(feats, coords, n_rows, cap, C, B, D, H, W, out, workspace, workspace_bytes, (hipStream_t)stream);
    if (dtype == FNP_BF16)
        return run_dense<__bf16>(feats, coords, n_rows, cap, C, B, D, H, W, out, workspace, workspace_bytes, (hipStream_t)stream);
    return FNP_ERR_ARG;
}

// fnp_sparse_to_dense with a per-channel value for the cells without a row (fill: C floats in device memory).  Needs the
// workspace (the single-pass writer).  The first BaseBEVBackbone block evaluated on sparse rows ends with it.
extern "C" int fnp_sparse_to_dense_fill(const void *feats, int dtype, const int *coords, const int *n_rows, int cap, int C,
                                        int B, int D, int H, int W, void *out, const float *fill, void *workspace,
                                        int64_t workspace_bytes, fnp_stream_t stream) {
    if (!feats || !coords || !n_rows || !out || !fill || !workspace || cap <= 0 || C <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0)
        return FNP_ERR_ARG;
    if (dtype == FNP_F32)
        return run_dense<float>(feats, coords, n_rows, cap, C, B, D, H, W, out, workspace, workspace_bytes, (hipStream_t)stream, fill);
    if (dtype == FNP_BF16)
        return run_dense<__bf16>(feats, coords, n_rows, cap, C, B, D, H, W, out, workspace, workspace_bytes, (hipStream_t)stream, fill);
    return FNP_ERR_ARG;
}
