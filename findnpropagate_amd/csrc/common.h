// Shared helpers for the gfx950 kernels behind the C ABI in include/fnp.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fnp.h"

// Launch-error check: the C ABI never exit()s (the reference's pybind launchers do,
// roiaware_pool3d_kernel.cu:350-354); it returns a negative fnp error code instead.
#define FNP_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return FNP_ERR_LAUNCH;        \
    } while (0)

#define FNP_HIP_TRY(expr)                                    \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return FNP_ERR_HIP;           \
    } while (0)

static inline int fnp_divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// Persistent-style grid size for the row-parallel kernels whose row count lives in device
// memory: enough workgroups to fill 256 CUs several times, capped by the capacity.
static inline int fnp_grid_for(long long capacity_items, int items_per_block, int max_blocks = 256 * 8) {
    long long need = (capacity_items + items_per_block - 1) / items_per_block;
    if (need < 1) need = 1;
    if (need > max_blocks) need = max_blocks;
    return (int)need;
}

__device__ __forceinline__ int fnp_lane() { return threadIdx.x & 63; }

// XCD-CONTIGUOUS WORKGROUP ORDER (round 5) for the row- and point-parallel index kernels.  The hardware deals the workgroups
// of a launch to the 8 XCDs round-robin (workgroup b runs on XCD b & 7) and every XCD has an L2 of its own (4 MB).  With rows
// taken as blockIdx.x * 256 + thread, eight NEIGHBOURING 256-row chunks — which read the same occupancy words, the same scene's
// points, the same lines of the per-scene arrays — land in eight different L2s: every line is fetched (and every partly written
// line written back) up to eight times.  fnp_xcd_block() renumbers the workgroups so that each XCD owns ONE contiguous run of
// logical blocks (1/8 of the launch, or of every grid-stride round): a scene's arrays then live in one L2.  A bijection of
// [0, gridDim.x): any kernel that derives its rows from blockIdx.x alone can take it; results cannot change.  Measured per kernel
// at 128 scenes (rocprofv3 averages of two builds on one box, round 5): vox_emit -6 %, vox_flag -13 %, the 64-channel tile-rulebook
// kernel -8 %, the small-grid prefix pass -13 %, the sparse clear -4 %; every kernel whose time is ATOMICS got slower (vox_mark
// +26 %, strided_mark2 +8 / +13 %, vox_insert +4 %) and keeps the interleaved order.  The whole step did not move (+-0.3 %).
#include "xcdmap.h"   // fnp_xcd_map(G, b): plain C++, compiled by the host test as it stands
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned fnp_xcd_block() { return fnp_xcd_map(gridDim.x, blockIdx.x); }
#endif


// Fill `count` 32-bit words at a 4-byte aligned address with `value`.  The library never issues hipMemsetAsync on a path a
// caller may capture into a hipGraph: a memset node captured by torch.cuda.graph (ROCm 7.2, torch 2.10) was seen to replay
// with the fill value of the last memset issued OUTSIDE the graph (a tensor.zero_()) once an eager forward had run between
// two replays — tools/dbg_graph.py shows it on the voxeliser's slot lists; a kernel node keeps its arguments.
static __global__ __launch_bounds__(256) void fnp_fill_words_kernel(unsigned *__restrict__ p, long long count, unsigned value) {
    const long long head = (((16 - ((uintptr_t)p & 15)) & 15) >> 2) < (unsigned long long)count ? (((16 - ((uintptr_t)p & 15)) & 15) >> 2) : count;
    uint4 *body = reinterpret_cast<uint4 *>(p + head);
    const long long n4 = (count - head) >> 2, tail0 = head + (n4 << 2);
    const uint4 v = make_uint4(value, value, value, value);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) body[i] = v;
    if (blockIdx.x == 0) {
        if ((long long)threadIdx.x < head) p[threadIdx.x] = value;
        if ((long long)threadIdx.x < count - tail0) p[tail0 + threadIdx.x] = value;
    }
}
static inline int fnp_fill_words(void *p, long long count, unsigned value, hipStream_t s) {
    if (count <= 0) return FNP_OK;
    if (!p || ((uintptr_t)p & 3)) return FNP_ERR_ARG;
    hipLaunchKernelGGL(fnp_fill_words_kernel, dim3(fnp_grid_for(count / 4 + 1, 256, 2048)), dim3(256), 0, s, (unsigned *)p, count, value);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
