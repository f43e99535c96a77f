// Development probe: what the GATHER PATTERN of the four 128 -> 128 layers can give, without the kernel around it.
// No MFMA, no weight staging, no epilogue: a workgroup of 8 waves walks 384-position tiles like spconv_mfma_kernel<128,128,3,27,...,sorted>
// (48 positions per wave as three 16-position blocks; lane (l15, q) fetches the 16 bytes [64 ks + 16 q, +16) of its position's neighbour row
// for ks = 0..3, i.e. a whole 256-byte bf16 row per four lanes), offset by offset over the offsets that are live for the tile, on the
// REAL stage-4 rulebook, with the rows taken through an optional position -> row permutation.  The loaded words are xor-ed into a sink.
// Not product code: built by tools/build_probe.sh into tools/probe/libgather_probe.so, driven by tools/gather_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int kTile = 384, kWaves = 8, kThreads = 512, kK = 27;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcd_map(unsigned G, unsigned b) {
    if (G < 16u) return b;
    const unsigned per = G >> 3, rem = G & 7u, x = b & 7u, sl = b >> 3;
    return (x < rem ? x * (per + 1u) : rem * (per + 1u) + (x - rem) * per) + sl;
}

// tilemask[t] bit k: some position of tile t has a neighbour at offset k;  pairs += valid (position, offset) entries
__global__ __launch_bounds__(kThreads) void tile_mask_kernel(const int *__restrict__ nbr, int stride, const int *__restrict__ perm, int n,
                                                             unsigned *__restrict__ tilemask, unsigned long long *__restrict__ pairs) {
    __shared__ unsigned m_s;
    __shared__ unsigned cnt_s;
    const int t = blockIdx.x;
    if (threadIdx.x == 0) m_s = 0u, cnt_s = 0u;
    __syncthreads();
    unsigned m = 0u, c = 0u;
    const int p = t * kTile + threadIdx.x;
    if (threadIdx.x < kTile && p < n) {
        const int r = perm ? perm[p] : p;
        for (int k = 0; k < kK; ++k)
            if (nbr[(size_t)k * stride + r] >= 0) m |= 1u << k, ++c;
    }
    atomicOr(&m_s, m);
    atomicAdd(&cnt_s, c);
    __syncthreads();
    if (threadIdx.x == 0) {
        tilemask[t] = m_s;
        atomicAdd(pairs, (unsigned long long)cnt_s);
    }
}

// ROW-WISE variant: the same rows, but a wave instruction fetches FOUR WHOLE ROWS (16 lanes x 16 B = 256 contiguous bytes per row: two
// full 128-byte lines per row, 8 line touches per instruction) instead of 16 quarter-rows (16 half-line touches): what an LDS-DMA
// (global_load_lds_dwordx4) or any "row per 16 lanes" fetch would ask of the address path.  Entries are read by the lane that needs them.
// ROW-WISE + LDS TRANSPOSE: as gather_rows_kernel, and every offset's 48 rows of a wave go through a wave-private LDS strip
// (12 x ds_write_b128 row-major with an XOR swizzle of the 16-byte chunks, 12 x ds_read_b128 in the MFMA B-fragment layout: lane
// (l15, q) reads chunk 4 ks + q of row l15) — what it costs to turn whole-row fetches into the fragments the matrix pipe wants.
template <bool SKIP>
__global__ __launch_bounds__(kThreads, 1) void gather_rows_lds_kernel(const unsigned char *__restrict__ x, long long x_bytes, const int *__restrict__ nbr, int stride,
                                                                      const int *__restrict__ perm, const unsigned *__restrict__ tilemask, int n, int identity,
                                                                      unsigned *__restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char strip[kWaves][48 * 256];
    const int ntiles = (n + kTile - 1) / kTile;
    const unsigned G = gridDim.x, range = xcd_map(G, blockIdx.x);
    const int t_begin = (int)(((long long)ntiles * range) / G), t_end = (int)(((long long)ntiles * (range + 1)) / G);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, r4 = lane >> 4, l15 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    unsigned char *mine = strip[wave];
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int t = t_begin; t < t_end; ++t) {
        const unsigned live = SKIP ? tilemask[t] : 0x7ffffffu;
        int row[3];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
            const int p = t * kTile + wave * 48 + mb * 16 + c16;
            row[mb] = p < n ? (perm ? perm[p] : p) : -1;
        }
        for (int k = 0; k < kK; ++k) {
            if (!((live >> k) & 1u)) continue;
            int ent[3];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                ent[mb] = -1;
                if (row[mb] >= 0) ent[mb] = identity ? row[mb] : nbr[(size_t)k * stride + row[mb]];
            }
            u32x4 v[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int e = __shfl(ent[i / 4], (i % 4) * 4 + r4);
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, e >= 0 ? (unsigned)e * 256u + (unsigned)c16 * 16u : 0x80000000u, 0, 0);
            }
            // row rr = 4 i + r4 of the strip, chunk c16 at position c16 ^ (rr & 15)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int rr = 4 * i + r4;
                *reinterpret_cast<u32x4 *>(mine + rr * 256 + ((c16 ^ (rr & 15)) << 4)) = v[i];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own writes have landed (wave-private strip: no barrier)
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int rr = mb * 16 + l15;
                    acc ^= *reinterpret_cast<const u32x4 *>(mine + rr * 256 + (((ks * 4 + q) ^ (rr & 15)) << 4));
                }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[blockIdx.x] = acc[0];
}

template <bool SKIP>
__global__ __launch_bounds__(kThreads, 1) void gather_rows_kernel(const unsigned char *__restrict__ x, long long x_bytes, const int *__restrict__ nbr, int stride,
                                                                  const int *__restrict__ perm, const unsigned *__restrict__ tilemask, int n, int identity,
                                                                  unsigned *__restrict__ sink) {
    const int ntiles = (n + kTile - 1) / kTile;
    const unsigned G = gridDim.x, range = xcd_map(G, blockIdx.x);
    const int t_begin = (int)(((long long)ntiles * range) / G), t_end = (int)(((long long)ntiles * (range + 1)) / G);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, r4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int t = t_begin; t < t_end; ++t) {
        const unsigned live = SKIP ? tilemask[t] : 0x7ffffffu;
        // entries are loaded as in the quarter-row form (lane l15 of block mb: three loads per lane and offset) and handed to the
        // lanes that fetch the row through a lane shuffle: instruction i covers positions 4 i .. 4 i + 3, this lane's is 4 i + r4
        int row[3];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
            const int p = t * kTile + wave * 48 + mb * 16 + c16;
            row[mb] = p < n ? (perm ? perm[p] : p) : -1;
        }
        for (int k = 0; k < kK; ++k) {
            if (!((live >> k) & 1u)) continue;
            int ent[3];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                ent[mb] = -1;
                if (row[mb] >= 0) ent[mb] = identity ? row[mb] : nbr[(size_t)k * stride + row[mb]];
            }
            u32x4 v[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int e = __shfl(ent[i / 4], (i % 4) * 4 + r4);
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, e >= 0 ? (unsigned)e * 256u + (unsigned)c16 * 16u : 0x80000000u, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) acc ^= v[i];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[blockIdx.x] = acc[0];
}

template <bool SKIP>
__global__ __launch_bounds__(kThreads, 1) void gather_kernel(const unsigned char *__restrict__ x, long long x_bytes, const int *__restrict__ nbr, int stride,
                                                             const int *__restrict__ perm, const unsigned *__restrict__ tilemask, int n, int identity,
                                                             unsigned *__restrict__ sink) {
    const int ntiles = (n + kTile - 1) / kTile;
    const unsigned G = gridDim.x, range = xcd_map(G, blockIdx.x);
    const int t_begin = (int)(((long long)ntiles * range) / G), t_end = (int)(((long long)ntiles * (range + 1)) / G);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int t = t_begin; t < t_end; ++t) {
        const unsigned live = SKIP ? tilemask[t] : 0x7ffffffu;
        int row[3];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
            const int p = t * kTile + wave * 48 + mb * 16 + l15;
            row[mb] = p < n ? (perm ? perm[p] : p) : -1;
        }
        for (int k = 0; k < kK; ++k) {
            if (!((live >> k) & 1u)) continue;   // (uniform per workgroup)
            unsigned off[3];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                int e = -1;
                if (row[mb] >= 0) e = identity ? row[mb] : nbr[(size_t)k * stride + row[mb]];
                off[mb] = e >= 0 ? (unsigned)e * 256u + (unsigned)q * 16u : 0x80000000u;   // absent: out of range, hardware zeros, no traffic
            }
            u32x4 v[3][4];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) v[mb][ks] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off[mb] + ks * 64u, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc ^= v[mb][ks];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[blockIdx.x] = acc[0];   // (never true in practice: keeps the loads alive)
}
}  // namespace

extern "C" int probe_tile_mask(const int *nbr, int stride, const int *perm, int n, unsigned *tilemask, unsigned long long *pairs, void *stream) {
    const int ntiles = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(tile_mask_kernel, dim3(ntiles), dim3(kThreads), 0, (hipStream_t)stream, nbr, stride, perm, n, tilemask, pairs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// x: (rows, 128) bf16; x_bytes < 2^31.  grid: workgroups (256 = one per CU, as the product kernel).  skip: honour tilemask.
extern "C" int probe_gather(const void *x, long long x_bytes, const int *nbr, int stride, const int *perm, const unsigned *tilemask, int n, int identity,
                            int skip, int grid, unsigned *sink, void *stream) {
    if (x_bytes >= 0x7fffffffll) return -2;
    if (skip == 4) {   // row-wise fetch through a wave-private LDS transpose
        hipLaunchKernelGGL(gather_rows_lds_kernel<true>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, (const unsigned char *)x, x_bytes, nbr, stride, perm, tilemask, n, identity, sink);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (skip == 3) {   // row-wise fetch, dead offsets skipped
        hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, (const unsigned char *)x, x_bytes, nbr, stride, perm, tilemask, n, identity, sink);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (skip)
        hipLaunchKernelGGL(gather_kernel<true>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, (const unsigned char *)x, x_bytes, nbr, stride, perm, tilemask, n, identity, sink);
    else
        hipLaunchKernelGGL(gather_kernel<false>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, (const unsigned char *)x, x_bytes, nbr, stride, perm, tilemask, n, identity, sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
