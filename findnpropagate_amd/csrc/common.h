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
