// GPU voxelisation with the sequential semantics of spconv's Point2VoxelCPU3d.point_to_voxel
// (call site pcdet/datasets/processor/data_processor.py:38-61; semantics SURVEY.md App. A.1)
// fused with MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:25-29).
//
// The reference is a single-threaded loop whose outputs depend on point order:
//   voxel row order   = order of each voxel's first point,
//   points per voxel  = the first max_points points of the voxel, in point order,
//   max_voxels        = voxels whose first-come rank >= max_voxels are dropped (their points too).
// Here the same result is produced in parallel, bit-exactly and deterministically:
//   1. every point sets its cell's bit in the rank grid (atomicOr)             [mark]
//   2. popcount scan -> rank of every occupied cell                            [scan.hip]
//   3. every point bubble-inserts its index into its cell's sorted list of the max_points
//      smallest point indices (chain of atomicMin; order independent)          [insert]
//   4. flag "I am my cell's smallest index", scan in point order -> first-come rank
//   5. per-scene max_voxels cut, then one pass per voxel writes coords, counts, the zero padded
//      (M, max_points, C) block and the mean feature row                       [emit]
// No sort, no hash probing, no floating-point atomics.
#include "rankgrid.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSentinel = 0x7f7f7f7f;  // memset(0x7f) pattern: larger than any point index
constexpr int kMaxBatch = 1024;

struct VoxWs {          // carve-up of the caller's workspace
    long long *code;    // (N)   (block << 6) | bit, or -1
    int *rank;          // (N)   sorted rank of the point's cell, or -1
    int *flag;          // (N)   first-point flag, then exclusive scan (first-come rank)
    int *top;           // (cap, max_points) point indices per cell: the cell's points in arrival order, or — a cell with more than
                        //                   max_points of them — its max_points smallest, ascending
    int *cnt;           // (cap) points per cell
    int *scene;         // (3*B + 4): fc_start[B+1], out_base[B+1], misc
    int *n_sorted;      // (1)
    int *n_first;       // (1)
    int *fscan;         // (n) exclusive scan of the first-point flags (its own array: the one-launch scan of small inputs is not in-place)
    void *scan_ws;
};

__host__ long long align_up(long long v) { return (v + 255) & ~255ll; }

__host__ long long carve(VoxWs &w, char *base, long long n, int B, int cap, int maxp, long long nsum) {
    long long off = 0;
    auto take = [&](long long bytes) {
        char *p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    w.code = (long long *)take(8 * n);
    w.rank = (int *)take(4 * n);
    w.flag = (int *)take(4 * n);
    w.fscan = (int *)take(4 * n);
    w.top = (int *)take(4ll * cap * maxp);
    w.cnt = (int *)take(4ll * cap + 4);   // (+ 1: "some cell is crowded", cleared with the counters)
    w.scene = (int *)take(4ll * (3 * B + 8));
    w.n_sorted = (int *)take(4);
    w.n_first = (int *)take(4);
    const long long a = fnp_scan::rank_grid_workspace_bytes(nsum), b2 = fnp_scan::workspace_bytes(n);
    w.scan_ws = take(a > b2 ? a : b2);
    return off;
}

__device__ __forceinline__ int batch_of(const int *__restrict__ off, int B, int i) {
    int lo = 0, hi = B;  // off[lo] <= i < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kThreads) void vox_mark_kernel(const float *__restrict__ pts, int n, int C,
                                                            const int *__restrict__ boff, int B, fnp_voxel_cfg cfg,
                                                            RG g, long long *__restrict__ code, int *__restrict__ cnt) {
    // (the per-cell counters of vox_insert_kernel start at zero: n + 1 words, one per thread here — a launch less than a fill)
    {
        const int z = blockIdx.x * kThreads + threadIdx.x;
        if (z <= n) cnt[z] = 0;
        if (z == 0 && (long long)gridDim.x * kThreads <= n) cnt[n] = 0;   // (n a multiple of the workgroup size)
    }
    __shared__ MarkTab tab;   // the marks of the workgroup's 256 points meet here first (rankgrid.h): one atomic per distinct block
    if (FNP_MARK_TAB) mark_tab_init(&tab, threadIdx.x, kThreads);
    const int i = blockIdx.x * kThreads + threadIdx.x;   // (whole waves stay: the shuffles below need them.  Plain workgroup order: XCD-contiguous
                                                         //  runs — common.h — made this kernel's atomics 26 % slower, round 5)
    const int lane = fnp_lane();
    long long blk = -1;
    unsigned long long m = 0ull;
    if (i < n) {
        const float *p = pts + (size_t)i * C;
        int c[3];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // f32 subtract, f32 divide (IEEE, correctly rounded), floor -> int: same as the reference loop
            const float q = (p[j] - cfg.range_min[j]) / cfg.voxel_size[j];
            const float f = floorf(q);
            ok = ok && (f >= 0.f) && (f < (float)cfg.grid[j]);
            c[j] = (int)f;
        }
        long long cd = -1;
        if (ok) {
            // scene of the point: searched once per wave for its first point (uniform: scalar loads, not queued
            // behind the point loads); a wave that crosses a scene border searches per lane
            const int i0 = __builtin_amdgcn_readfirstlane(i);
            int b = batch_of(boff, B, i0);
            if (b + 1 < B && i >= boff[b + 1]) b = batch_of(boff, B, i);
            blk = rg_block_of(g.d, b, c[2], c[1], c[0]);
            const int bit = rg_bit_of(c[2], c[1], c[0]);
            m = 1ull << bit;
            cd = (blk << 6) | bit;
        }
        code[i] = cd;
    }
    // A lidar sweep visits a 4x4x4 block with runs of consecutive points: OR the bits of equal blocks along
    // the wave (a lane may take in any earlier lane's bits of the same block) and let the last lane of each
    // run issue the one atomic — same-address atomics serialise in L2.  Any point order gives the same grid.
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long nb = __shfl_up(blk, d);
        const unsigned long long nm = __shfl_up(m, d);
        if (lane >= d && nb == blk) m |= nm;
    }
    const long long nxt = __shfl_down(blk, 1);
    if (blk >= 0 && (lane == 63 || nxt != blk)) mark_put(FNP_MARK_TAB ? &tab : nullptr, g, blk, m);
    if (FNP_MARK_TAB) mark_tab_flush(&tab, g, threadIdx.x, kThreads);
}

__global__ __launch_bounds__(kThreads) void vox_insert_kernel(int n, int maxp, int cap,
                                                              const unsigned long long *__restrict__ bits,
                                                              const unsigned *__restrict__ base,
                                                              const long long *__restrict__ code,
                                                              int *__restrict__ rank, int *__restrict__ top, int *__restrict__ cnt,
                                                              int *__restrict__ crowded, int *__restrict__ late) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    late[i] = 0;
    const long long cd = code[i];
    int r = -1;
    if (cd >= 0) {
        const long long blk = cd >> 6;
        const int bit = (int)(cd & 63);
        r = (int)base[blk] + __popcll(bits[blk] & ((1ull << bit) - 1ull));
        if (r >= cap) r = -1;  // cannot happen when cap >= n; defensive
    }
    rank[i] = r;
    if (r < 0) return;
    // Count and append (round 3).  The order-independent bubble insert (atomicMin down the cell's sorted list) cost 2.4 M
    // read-modify-writes of a 77 MB array per step, almost all of them L2 misses, for cells that hold 1.6 points on average:
    // a point now takes a slot of its cell with ONE atomic on a compact counter array (4 bytes per cell: it stays in L2) and
    // stores its index there; which slot depends on timing, the SET of a cell's points does not, and the consumers order
    // it.  Only a cell with more than max_points points needs the max_points SMALLEST: the points that came too late for a
    // slot (late[i] = 1) are bubbled into the full list by the small kernel below.
    const int s = atomicAdd(&cnt[r], 1);
    if (s < maxp) {
        top[(size_t)r * maxp + s] = i;
    } else {
        late[i] = 1;
        if (s == maxp) *crowded = 1;   // (plain store of the same value by whoever sees it: the kernel below leaves at once without it)
    }
}

// A cell with more points than a voxel keeps: its max_points slots hold the points that came first (any order); every LATE point
// walks down the list with atomicMin, leaving the smaller of (slot, carried) behind and carrying the larger on, and drops what
// it carries at the end.  A step preserves the multiset {slot, carried}, slots only ever decrease and a carried value never
// decreases along its walk, so a dropped value is larger than everything the list holds at the end: the list ends as the
// max_points smallest indices of the cell whatever the interleaving and whatever order the first-comers stand in (the consumers
// order a list themselves).  The reset-to-sentinel kernel of the first form is gone.
__global__ __launch_bounds__(kThreads) void vox_crowded_insert_kernel(int n, int maxp, const int *__restrict__ rank, const int *__restrict__ late,
                                                                     int *__restrict__ top, const int *__restrict__ crowded) {
    if (*crowded == 0) return;   // (uniform)
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n || late[i] == 0) return;
    int carry = i;
    int *slots = top + (size_t)rank[i] * maxp;
    for (int j = 0; j < maxp; ++j) {
        const int old = atomicMin(&slots[j], carry);
        carry = old > carry ? old : carry;  // keep pushing the larger one down
    }
}

// flag[i] = point i is the FIRST point of its cell (the smallest index among the cell's kept points)
__global__ __launch_bounds__(kThreads) void vox_flag_kernel(int n, int maxp, const int *__restrict__ rank,
                                                            const int *__restrict__ top, const int *__restrict__ cnt, int *__restrict__ flag) {
    const int i = fnp_xcd_block() * kThreads + threadIdx.x;
    if (i >= n) return;
    const int r = rank[i];
    int f = 0;
    if (r >= 0) {
        const int np = min(cnt[r], maxp);
        int m = kSentinel;
        for (int j = 0; j < np; ++j) m = min(m, top[(size_t)r * maxp + j]);
        f = m == i ? 1 : 0;
    }
    flag[i] = f;
}

// one wave: per-scene first-come starts and output bases after the max_voxels cut (B <= kMaxBatch;
// lane-strided over the scenes, running prefix carried across the 64-scene rounds).
__global__ __launch_bounds__(64) void vox_scene_kernel(const int *__restrict__ boff, int B, int n, const int *__restrict__ fc,
                                                       const int *__restrict__ n_first, int max_voxels, int cap,
                                                       int *__restrict__ scene, int *__restrict__ n_voxels) {
    int *fc_start = scene;           // (B+1)
    int *out_base = scene + (B + 1); // (B+1)
    const int lane = threadIdx.x;
    const int total = *n_first;
    int acc = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        int c = 0;
        if (b < B) {
            const int o0 = boff[b], o1 = boff[b + 1];
            const int f0 = (o0 < n) ? fc[o0] : total, f1 = (o1 < n) ? fc[o1] : total;
            fc_start[b] = f0;
            if (b == B - 1) fc_start[B] = f1;
            c = min(f1 - f0, max_voxels);
        }
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (b < B) out_base[b] = acc + inc - c;
        acc += __shfl(inc, 63);
    }
    if (lane == 0) {
        out_base[B] = acc;
        scene[2 * (B + 1)] = 0;   // dropped-cell counter of the emit pass
        *n_voxels = acc;  // true count; consumers clamp to their capacity
    }
}

// The points a cell keeps: min(cnt, max_points) slots in arrival order (a crowded cell: already ascending); they are consumed in
// ASCENDING index order — next = the smallest index above the previous one — so that sums and the (M, max_points, C) block come
// out as from a sorted list.
struct CellPts {
    const int *slots;
    int np, s0, s1;
    __device__ __forceinline__ int next_above(int prev) const {
        if (np <= 2) {
            const int a = s0, b = np == 2 ? s1 : kSentinel;
            const int lo = a < b ? a : b, hi = a < b ? b : a;
            return lo > prev ? lo : hi;
        }
        int m = kSentinel;
        for (int j = 0; j < np; ++j) {
            const int v = slots[j];
            m = (v > prev && v < m) ? v : m;
        }
        return m;
    }
};

// one voxel row: coordinates, point count, mean feature row (MeanVFE) and, when asked for, the zero padded (max_points, C) block
__device__ __forceinline__ void vox_write_row(const float *__restrict__ pts, int C, int maxp, const CellPts &cp, int p0, int r, int id,
                                              int bb, int z, int y, int x, int *__restrict__ perm, int *__restrict__ coords,
                                              int *__restrict__ num_points, float *__restrict__ mean, float *__restrict__ voxels) {
    const int np = cp.np;
    auto next_above = [&](int prev) -> int { return cp.next_above(prev); };
    perm[r] = id;
    reinterpret_cast<int4 *>(coords)[id] = make_int4(bb, z, y, x);
    num_points[id] = np;
    const float norm = (float)(np < 1 ? 1 : np);
    // Per channel the kept points meet in the order torch's CPU `voxels.sum(dim=1)` adds the slots of the (M, max_points, C) block
    // (mean_vfe.py:26; round 6: held to the reference's own class bit for bit, tests/golden/meanvfe_golden.npz).  That order
    // is cascade_sum's (aten/src/ATen/native/cpu/SumKernel.cpp, restated in oracle/fnp_oracle.c orc_mean_vfe): for C < 8 the
    // columns 0 .. 4 * (C / 4) - 1 in slot order, the C % 4 columns left over — t of a 5-feature nuScenes point — as FOUR
    // interleaved partial sums (slot j to partial j & 3 for j < 4 * (P / 4)), partial 0 then takes the P % 4 tail slots and
    // the partials 1, 2, 3; blocks of 16 slots cascade through accumulator levels (P >= 16 only).  Padding slots are zeros.
    if ((C == 5 || C == 4) && maxp < 16) {
        // nuScenes / KITTI point rows (x y z intensity [t]) as ONE 16-byte access + one dword instead of five dword accesses:
        // this kernel is bound by the NUMBER of scattered requests it puts through L2 (~22 per voxel before, round 5), not by
        // bytes.  Rows are 4-byte aligned only: f4u carries that alignment (global memory takes dword-aligned wide accesses).
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        f4u s4 = {0.f, 0.f, 0.f, 0.f};
        float part[4] = {0.f, 0.f, 0.f, 0.f}, tl[3] = {0.f, 0.f, 0.f};
        const int P4 = maxp & ~3;
        int pi = p0;
        for (int j = 0; j < np; ++j) {
            const float *pr = pts + (size_t)pi * C;
            const f4u v4 = *reinterpret_cast<const f4u *>(pr);
            const float v1 = C == 5 ? pr[4] : 0.f;
            if (j + 1 < np) pi = next_above(pi);   // (the slots are in L1: the search runs under the point's loads)
            s4 += v4;
            // (register selects, no indexed array: a partial is never -0.0, so the + 0.f of the other three is exact)
#pragma unroll
            for (int k = 0; k < 4; ++k) part[k] += (j < P4 && (j & 3) == k) ? v1 : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) tl[k] = (j == P4 + k) ? v1 : tl[k];
        }
        float *mo = mean + (size_t)id * C;
        f4u m4 = {s4[0] / norm, s4[1] / norm, s4[2] / norm, s4[3] / norm};
        *reinterpret_cast<f4u *>(mo) = m4;
        if (C == 5) {
            float s1f = part[0];
#pragma unroll
            for (int k = 0; k < 3; ++k) s1f += (P4 + k < maxp) ? tl[k] : 0.f;
            s1f += part[1];
            s1f += part[2];
            s1f += part[3];
            mo[4] = s1f / norm;
        }
    } else {
        // any other shape: column by column with the cascade written out (a pass over the voxel's points per column)
        for (int c = 0; c < C; ++c) {
            const bool ilp = C < 8 && c >= (C / 4) * 4;           // a left-over column: four interleaved partials
            const int nsum = ilp ? 4 : 1, size = ilp ? maxp / 4 : maxp;
            int lp = 0;
            while ((1 << lp) < size) ++lp;
            lp = C >= 8 ? 30 : max(4, lp / 4);                       // (C >= 8: slot order, as the oracle — torch's form depends on the host ISA)
            const int mask = (1 << lp) - 1;
            float acc[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int l = 0; l < 4; ++l) acc[k][l] = 0.f;
            float tail = 0.f;
            bool tail_started = false;
            int pi = p0;
            for (int j = 0; j < maxp; ++j) {                         // every slot, the zero padding included (it moves the cascade's levels)
                const float v = j < np ? pts[(size_t)pi * C + c] : 0.f;
                if (j + 1 < np) pi = next_above(pi);
                if (j < size * nsum) {
                    const int e = j / nsum + 1;                      // elements this partial holds after the add
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k != (ilp ? (j & 3) : 0)) continue;
                        acc[k][0] += v;
                        if ((e & mask) == 0) {
#pragma unroll
                            for (int l = 1; l < 4; ++l) {
                                acc[k][l] += acc[k][l - 1];
                                acc[k][l - 1] = 0.f;
                                if ((e & (mask << (l * lp))) != 0) break;
                            }
                        }
                    }
                } else {                                             // the P % 4 tail slots of a left-over column go to partial 0
                    if (!tail_started) {
#pragma unroll
                        for (int l = 1; l < 4; ++l) acc[0][0] += acc[0][l];
                        tail = acc[0][0];
                        tail_started = true;
                    }
                    tail += v;
                }
            }
            float sacc;
            if (ilp) {
                if (!tail_started) {
#pragma unroll
                    for (int l = 1; l < 4; ++l) acc[0][0] += acc[0][l];
                    tail = acc[0][0];
                }
                sacc = tail;
#pragma unroll
                for (int k = 1; k < 4; ++k) {
#pragma unroll
                    for (int l = 1; l < 4; ++l) acc[k][0] += acc[k][l];
                    sacc += acc[k][0];
                }
            } else {
#pragma unroll
                for (int l = 1; l < 4; ++l) acc[0][0] += acc[0][l];
                sacc = acc[0][0];
            }
            mean[(size_t)id * C + c] = sacc / norm;
        }
    }
    if (voxels) {
        float *v = voxels + (size_t)id * maxp * C;
        int pi = p0;
        for (int j = 0; j < maxp; ++j) {
            for (int c = 0; c < C; ++c) v[j * C + c] = j < np ? pts[(size_t)pi * C + c] : 0.f;
            if (j + 1 < np) pi = next_above(pi);
        }
    }
}

// CELL form (rounds 1-5; FNP_VOX_EMIT=cell): a thread per occupied cell in RANK order.  Everything behind the cell's slot list is a
// scattered access: code and first-come rank of its first point, its points' rows, and the voxel row it writes (first-come order).
__global__ __launch_bounds__(kThreads) void vox_emit_kernel(const float *__restrict__ pts, int C, int maxp,
                                                            const int *__restrict__ boff, int B, int max_voxels,
                                                            RankGridDims g, const long long *__restrict__ code,
                                                            const int *__restrict__ top, const int *__restrict__ cnt,
                                                            const int *__restrict__ fc, const int *__restrict__ scene,
                                                            const int *__restrict__ n_sorted, int cap,
                                                            int *__restrict__ perm, int *__restrict__ coords,
                                                            int *__restrict__ num_points, float *__restrict__ mean,
                                                            float *__restrict__ voxels, int *__restrict__ n_cells,
                                                            int *__restrict__ n_dropped) {
    const int ns = min(*n_sorted, cap);
    const int *fc_start = scene;
    const int *out_base = scene + (B + 1);
    if (n_cells && blockIdx.x == 0 && threadIdx.x == 0) *n_cells = ns;
    // ranks beyond the occupied cells map to no row (consumers walk perm[0 .. cap))
    for (int r = ns + blockIdx.x * kThreads + threadIdx.x; r < cap; r += gridDim.x * kThreads) perm[r] = -1;
    // Every XCD takes ONE contiguous eighth of the ranks (ranks run scene by scene): the cell -> voxel-row scatter below is random
    // inside a scene — rank order is spatial, row order first-come — but touches that scene's ~2 MB of points, codes, first-come
    // ranks, coordinates and means only, which one 4 MB L2 holds.  With interleaved workgroups every one of those lines went
    // through all eight L2s (418 MB of HBM traffic per 64-scene launch against ~165 MB of tensors, round 3's PMC).
    const long long lb = fnp_xcd_block(), nchunk = (ns + kThreads - 1) / kThreads;      // balanced runs of whole 256-rank chunks
    const int r_begin = (int)(nchunk * lb / gridDim.x) * kThreads, r_end = min(ns, (int)(nchunk * (lb + 1) / gridDim.x) * kThreads);
    for (int r = r_begin + threadIdx.x; r < r_end; r += kThreads) {
        CellPts cp;
        cp.slots = top + (size_t)r * maxp;
        const int cn = cnt[r];
        cp.s0 = cp.slots[0];
        cp.s1 = maxp > 1 ? cp.slots[1] : kSentinel;   // (independent loads: most cells hold one or two points)
        cp.np = min(cn, maxp);
        const int p0 = cp.next_above(-1);
        // the scene of the cell is in its block number: no search in the batch offsets
        const long long cd = code[p0];
        int bb, z, y, x;
        rg_decode(g, cd >> 6, (int)(cd & 63), bb, z, y, x);
        const int srank = fc[p0] - fc_start[bb];
        const int id = out_base[bb] + srank;
        if (srank >= max_voxels || id >= cap) {
            // dropped by the per-scene cut: no voxel row; its cell goes behind the voxels in the coordinate
            // list (any order) so that the sparse clear of a persistent grid reaches it
            perm[r] = -1;
            if (n_cells) {
                const int t = out_base[B] + atomicAdd(n_dropped, 1);
                if (t < cap) reinterpret_cast<int4 *>(coords)[t] = make_int4(bb, z, y, x);
            }
            continue;
        }
        vox_write_row(pts, C, maxp, cp, p0, r, id, bb, z, y, x, perm, coords, num_points, mean, voxels);
    }
}

// POINT form (round 6, default): a thread per POINT in point order; the thread of a cell's FIRST point (fc[i + 1] != fc[i]) writes
// the voxel.  Point rows, codes, ranks and first-come ranks are then read coalesced, and the voxel rows — first-come order IS point
// order — are written nearly coalesced: consecutive first points own consecutive rows.  What stays scattered is what is keyed by
// the cell: its point count, its slot list when it holds more than one point (and those points' rows), and perm[rank].  The cell
// form paid ~9 scattered accesses per voxel (it is bound by their number, not by bytes: round 5), this one 2-4.
__global__ __launch_bounds__(kThreads) void vox_emit_points_kernel(const float *__restrict__ pts, int n, int C, int maxp,
                                                                   const int *__restrict__ boff, int B, int max_voxels,
                                                                   RankGridDims g, const long long *__restrict__ code,
                                                                   const int *__restrict__ rank, const int *__restrict__ top,
                                                                   const int *__restrict__ cnt, const int *__restrict__ fc,
                                                                   const int *__restrict__ n_first, const int *__restrict__ scene,
                                                                   const int *__restrict__ n_sorted, int cap,
                                                                   int *__restrict__ perm, int *__restrict__ coords,
                                                                   int *__restrict__ num_points, float *__restrict__ mean,
                                                                   float *__restrict__ voxels, int *__restrict__ n_cells,
                                                                   int *__restrict__ n_dropped) {
    const int ns = min(*n_sorted, cap);
    const int *fc_start = scene;
    const int *out_base = scene + (B + 1);
    if (n_cells && blockIdx.x == 0 && threadIdx.x == 0) *n_cells = ns;
    for (int r = ns + blockIdx.x * kThreads + threadIdx.x; r < cap; r += gridDim.x * kThreads) perm[r] = -1;
    const int i = (int)fnp_xcd_block() * kThreads + threadIdx.x;   // (one eighth of the points — whole scenes' worth — per XCD)
    if (i >= n) return;
    const int r = rank[i];
    if (r < 0) return;
    const int f0 = fc[i], f1 = i + 1 < n ? fc[i + 1] : *n_first;
    if (f1 == f0) return;                                          // not its cell's first point
    const long long cd = code[i];
    int bb, z, y, x;
    rg_decode(g, cd >> 6, (int)(cd & 63), bb, z, y, x);
    const int srank = f0 - fc_start[bb];
    const int id = out_base[bb] + srank;
    if (srank >= max_voxels || id >= cap) {
        perm[r] = -1;
        if (n_cells) {
            const int t = out_base[B] + atomicAdd(n_dropped, 1);
            if (t < cap) reinterpret_cast<int4 *>(coords)[t] = make_int4(bb, z, y, x);
        }
        return;
    }
    CellPts cp;
    cp.slots = top + (size_t)r * maxp;
    cp.np = min(cnt[r], maxp);
    cp.s0 = i;                     // (a one-point cell reads no slot at all)
    cp.s1 = kSentinel;
    if (cp.np == 2) {              // the other point of a two-point cell: whichever slot is not i
        const int a = cp.slots[0], b = cp.slots[1];
        cp.s1 = a == i ? b : a;
    }
    vox_write_row(pts, C, maxp, cp, i, r, id, bb, z, y, x, perm, coords, num_points, mean, voxels);
}

// ---- rank grid from an explicit coordinate list ---------------------------------------------
__device__ __forceinline__ bool coord_ok(const RankGridDims &g, const int4 &c) {
    return c.x >= 0 && c.x < g.B && c.y >= 0 && c.y < g.D && c.z >= 0 && c.z < g.H && c.w >= 0 && c.w < g.W;
}

// the counters of the counted marks (rankgrid.h) go back to zero with the grid: a dense loop over a few thousand words
__device__ __forceinline__ void rg_zero_counters(const RG &g, long long first, long long stride) {
    if (!g.ctr) return;
    const long long words = g.nunits + ((g.nunits + 63) >> 6) + ((g.nunits + 1023) >> 10);   // (the words this grid's unit size uses)
    for (long long i = first; i < words; i += stride) g.ctr[i] = 0u;
}

template <bool CLEAR>
__global__ __launch_bounds__(kThreads) void rg_mark_coords_kernel(const int *__restrict__ coords,
                                                                  const int *__restrict__ n_rows, int cap, RG g) {
    const int n = min(*n_rows, cap);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[i];
        if (!coord_ok(g.d, c)) continue;
        const long long blk = rg_block_of(g.d, c.x, c.y, c.z, c.w);
        if (CLEAR) {
            g.bits[blk] = 0ull;
            g.summ[blk >> 6] = 0ull;   // every occupied block of that summary word has a row here
        } else {
            rg_mark(g, blk, rg_bit_of(c.y, c.z, c.w));
        }
    }
    if (CLEAR) rg_zero_counters(g, (long long)blockIdx.x * kThreads + threadIdx.x, (long long)gridDim.x * kThreads);
}

__global__ __launch_bounds__(kThreads) void rg_perm_kernel(const int *__restrict__ coords, const int *__restrict__ n_rows,
                                                           int cap, RG g) {
    const int n = min(*n_rows, cap);
    RG lookup = g;
    lookup.perm = nullptr;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[i];
        if (!coord_ok(g.d, c)) continue;
        const int r = rg_lookup(lookup, c.x, c.y, c.z, c.w);
        if (r >= 0 && r < cap) g.perm[r] = i;
    }
}

}  // namespace

extern "C" int64_t fnp_rankgrid_num_blocks(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return fnp_num_blocks(fnp_make_dims(B, D, H, W));
}
extern "C" int64_t fnp_rankgrid_num_summary(int B, int D, int H, int W) {
    return (fnp_rankgrid_num_blocks(B, D, H, W) + 63) >> 6;
}

static bool grid_covers(const fnp_voxel_cfg *cfg, const fnp_rankgrid *g) {
    return g->D >= cfg->grid[2] && g->H >= cfg->grid[1] && g->W >= cfg->grid[0];
}

extern "C" int64_t fnp_voxelize_workspace_bytes(int64_t n_points, const fnp_voxel_cfg *cfg, const fnp_rankgrid *grid) {
    if (!cfg || n_points < 0 || !grid || grid->B <= 0 || !grid_covers(cfg, grid)) return FNP_ERR_ARG;
    VoxWs w;
    const RankGridDims g = fnp_make_dims(grid->B, grid->D, grid->H, grid->W);
    const long long n = n_points > 0 ? n_points : 1;
    return carve(w, nullptr, n, grid->B, (int)n, cfg->max_points, (fnp_num_blocks(g) + 63) >> 6);
}

// A frame into the static inputs of a captured forward, in one launch: the points, the padding value behind them as far as the
// previous frame reached, and the scene offsets (three stream operations before: a one-scene forward is a chain of ~50 launches).
__global__ __launch_bounds__(kThreads) void stage_points_kernel(const float *__restrict__ src, long long n_words, long long n_prev_words, float pad,
                                                                float *__restrict__ dst, const int *__restrict__ off_src, int n_off,
                                                                int *__restrict__ off_dst) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i < n_words) dst[i] = src[i];
    else if (i < n_prev_words) dst[i] = pad;
    if (i < n_off) off_dst[i] = off_src[i];
}
extern "C" int fnp_stage_points(const float *points, int64_t n_points, int64_t n_prev_points, int num_features, float pad, float *dst_points,
                                const int *batch_offsets, int n_offsets, int *dst_offsets, fnp_stream_t stream) {
    if (n_points < 0 || n_prev_points < 0 || num_features <= 0 || !dst_points || n_offsets < 0 || (n_points > 0 && !points) ||
        (n_offsets > 0 && (!batch_offsets || !dst_offsets)))
        return FNP_ERR_ARG;
    const long long nw = (long long)n_points * num_features, pw = (long long)n_prev_points * num_features;
    const long long span = nw > pw ? nw : pw;
    const long long items = span > n_offsets ? span : n_offsets;
    if (items == 0) return FNP_OK;
    if (items > (long long)kThreads * 0x7fffffff) return FNP_ERR_ARG;
    hipLaunchKernelGGL(stage_points_kernel, dim3((unsigned)((items + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, points, nw, pw, pad,
                       dst_points, batch_offsets, n_offsets, dst_offsets);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_voxelize(const float *points, int n, const int *batch_offsets, const fnp_voxel_cfg *cfg,
                            const fnp_rankgrid *grid, void *workspace, int64_t workspace_bytes, int *coords,
                            int *num_points, float *mean_feats, float *voxels, int *n_voxels, int *n_cells, int cap,
                            fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!cfg || n < 0 || cap <= 0 || !n_voxels || !fnp_rg_valid(grid, true) || grid->B > kMaxBatch || !grid_covers(cfg, grid))
        return FNP_ERR_ARG;
    if (cfg->num_features < 3 || cfg->max_points <= 0 || cfg->max_points > 64 || cfg->max_voxels <= 0) return FNP_ERR_ARG;
    if (n == 0) {
        int frc = fnp_fill_words(n_voxels, 1, 0u, s);
        if (!frc && n_cells) frc = fnp_fill_words(n_cells, 1, 0u, s);
        return frc;
    }
    if (!points || !batch_offsets || !workspace || !coords || !num_points || !mean_feats) return FNP_ERR_ARG;
    if (cap < n) return FNP_ERR_ARG;  // every point could open a voxel
    const RG g = fnp_rg_view(grid);
    const int B = grid->B;
    VoxWs w;
    const long long need = carve(w, (char *)workspace, n, B, n, cfg->max_points, g.nsum);
    if (need > workspace_bytes) return FNP_ERR_WORKSPACE;
    const int maxp = cfg->max_points, C = cfg->num_features;
    const int pgrid = fnp_divup(n, kThreads);

    // (the per-cell counters are zeroed by the marking kernel; the slot lists need no initial value: a cell's first
    //  min(cnt, max_points) slots are written before they are read)
    hipLaunchKernelGGL(vox_mark_kernel, dim3(pgrid), dim3(kThreads), 0, s, points, n, C, batch_offsets, B, *cfg, g, w.code, w.cnt);
    FNP_LAUNCH_CHECK();
    int rc = fnp_scan::rank_grid(g, w.n_sorted, w.scan_ws, s, nullptr, 0, g.ctr != nullptr);   // (vox_mark_kernel counted its marks)
    if (rc) return rc;
    hipLaunchKernelGGL(vox_insert_kernel, dim3(pgrid), dim3(kThreads), 0, s, n, maxp, n,
                       (const unsigned long long *)g.bits, (const unsigned *)g.base, w.code, w.rank, w.top, w.cnt, w.cnt + n, w.flag);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(vox_crowded_insert_kernel, dim3(pgrid), dim3(kThreads), 0, s, n, maxp, (const int *)w.rank, (const int *)w.flag, w.top, (const int *)(w.cnt + n));
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(vox_flag_kernel, dim3(pgrid), dim3(kThreads), 0, s, n, maxp, w.rank, w.top, (const int *)w.cnt, w.flag);
    FNP_LAUNCH_CHECK();
    rc = fnp_scan::int32(w.flag, n, w.fscan, w.n_first, w.scan_ws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_scene_kernel, dim3(1), dim3(64), 0, s, batch_offsets, B, n, w.fscan, w.n_first,
                       cfg->max_voxels, cap, w.scene, n_voxels);
    FNP_LAUNCH_CHECK();
    static const bool cell_form = [] { const char *e = getenv("FNP_VOX_EMIT"); return e && e[0] == 'c'; }();   // (development A/B)
    if (cell_form)
        hipLaunchKernelGGL(vox_emit_kernel, dim3(fnp_grid_for(n, kThreads)), dim3(kThreads), 0, s, points, C, maxp,
                           batch_offsets, B, cfg->max_voxels, g.d, w.code, w.top, (const int *)w.cnt, w.fscan, w.scene, w.n_sorted, n,
                           g.perm, coords, num_points, mean_feats, voxels, n_cells, w.scene + 2 * (B + 1));
    else
        hipLaunchKernelGGL(vox_emit_points_kernel, dim3(pgrid), dim3(kThreads), 0, s, points, n, C, maxp, batch_offsets, B, cfg->max_voxels,
                           g.d, w.code, (const int *)w.rank, w.top, (const int *)w.cnt, w.fscan, (const int *)w.n_first, w.scene, w.n_sorted, n,
                           g.perm, coords, num_points, mean_feats, voxels, n_cells, w.scene + 2 * (B + 1));
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_rankgrid_build(const int *coords, const int *n_rows, int cap, const fnp_rankgrid *grid,
                                  void *workspace, int64_t workspace_bytes, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!coords || !n_rows || cap <= 0 || !fnp_rg_valid(grid) || !workspace) return FNP_ERR_ARG;
    const RG g = fnp_rg_view(grid);
    if (fnp_scan::rank_grid_workspace_bytes(g.nsum) + 256 > workspace_bytes) return FNP_ERR_WORKSPACE;
    int *total = (int *)workspace;
    void *scan_ws = (char *)workspace + 256;
    const int blocks = fnp_grid_for(cap, kThreads);
    hipLaunchKernelGGL(rg_mark_coords_kernel<false>, dim3(blocks), dim3(kThreads), 0, s, coords, n_rows, cap, g);
    FNP_LAUNCH_CHECK();
    int rc = fnp_scan::rank_grid(g, total, scan_ws, s);   // (rg_mark_coords_kernel does not count: three-launch prefix; the counters stay zero)
    if (rc) return rc;
    if (g.perm) {
        {
            const int frc = fnp_fill_words(g.perm, cap, 0xffffffffu, s);
            if (frc) return frc;
        }
        hipLaunchKernelGGL(rg_perm_kernel, dim3(blocks), dim3(kThreads), 0, s, coords, n_rows, cap, g);
        FNP_LAUNCH_CHECK();
    }
    return FNP_OK;
}

extern "C" int fnp_rankgrid_clear(const int *coords, const int *n_rows, int cap, const fnp_rankgrid *grid,
                                  fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || !fnp_rg_valid(grid)) return FNP_ERR_ARG;
    const RG g = fnp_rg_view(grid);
    hipLaunchKernelGGL(rg_mark_coords_kernel<true>, dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, coords, n_rows, cap, g);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// All persistent grids of a forward in ONE launch (blockIdx.y = grid): the fused backbone clears five of them per step.
struct ClearJob {
    const int *coords, *n_rows;
    int cap;
    RG g;
};
struct ClearJobs { ClearJob j[8]; };

__global__ __launch_bounds__(kThreads) void rg_clear_multi_kernel(ClearJobs jobs) {
    const ClearJob &J = jobs.j[blockIdx.y];
    const int n = min(*J.n_rows, J.cap);
    for (int i = fnp_xcd_block() * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(J.coords)[i];
        if (!coord_ok(J.g.d, c)) continue;
        const long long blk = rg_block_of(J.g.d, c.x, c.y, c.z, c.w);
        J.g.bits[blk] = 0ull;
        J.g.summ[blk >> 6] = 0ull;
    }
    rg_zero_counters(J.g, blockIdx.x * kThreads + threadIdx.x, gridDim.x * kThreads);
}

extern "C" int fnp_rankgrid_clear_multi(int count, const int *const *coords, const int *const *n_rows, const int *caps,
                                        const fnp_rankgrid *grids, fnp_stream_t stream) {
    if (count <= 0 || count > 8 || !coords || !n_rows || !caps || !grids) return FNP_ERR_ARG;
    ClearJobs jobs{};
    int cap_max = 1;
    for (int i = 0; i < count; ++i) {
        if (!coords[i] || !n_rows[i] || caps[i] <= 0 || !fnp_rg_valid(&grids[i])) return FNP_ERR_ARG;
        jobs.j[i].coords = coords[i];
        jobs.j[i].n_rows = n_rows[i];
        jobs.j[i].cap = caps[i];
        jobs.j[i].g = fnp_rg_view(&grids[i]);
        if (caps[i] > cap_max) cap_max = caps[i];
    }
    // (gridDim.x a multiple of 8: fnp_xcd_block()'s one-run-per-XCD property on a 2-D grid needs it — common.h)
    const int gx = (fnp_grid_for(cap_max, kThreads, 512) + 7) & ~7;
    hipLaunchKernelGGL(rg_clear_multi_kernel, dim3(gx, count), dim3(kThreads), 0, (hipStream_t)stream, jobs);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// SUMMARY-DRIVEN CLEAR (round 6).  The row form above reads every coordinate row of every stage (5.8 M rows, 93 MB at 128 scenes)
// and zeroes the occupancy word and the summary word of each — 7 rows per occupied block, 11.6 M scattered stores, 79 us.  The
// summary level already names the occupied blocks: a wave per 64 summary words reads them (27 MB for the five grids of a 128-scene
// batch, coalesced), and for every non-zero word the lanes whose bit is set zero their block's occupancy word; the summary word
// goes last.  ~0.8 M stores instead of 11.6 M — and it needs no coordinate list: cells of voxels dropped by max_voxels and sites
// beyond a stage's row capacity (which the row form could not see: the engine wiped the whole grid after an overflow) go too.
__device__ __forceinline__ unsigned long long clr_readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
__global__ __launch_bounds__(kThreads) void rg_clear_summary_kernel(ClearJobs jobs) {
    const RG &g = jobs.j[blockIdx.y].g;
    const int lane = fnp_lane();
    const long long nunits = (g.nsum + 63) >> 6, wstride = (long long)gridDim.x * (kThreads / 64);
    for (long long U = (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); U < nunits; U += wstride) {
        const long long S0 = U * 64;
        const unsigned long long sw = (S0 + lane < g.nsum) ? g.summ[S0 + lane] : 0ull;
        unsigned long long todo = __ballot(sw != 0ull);
        if (todo == 0ull) continue;   // (uniform)
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const unsigned long long swj = clr_readlane64(sw, j);
            if ((swj >> lane) & 1ull) g.bits[(S0 + j) * 64 + lane] = 0ull;
        }
        if (sw != 0ull) g.summ[S0 + lane] = 0ull;
    }
    rg_zero_counters(g, (long long)blockIdx.x * kThreads + threadIdx.x, (long long)gridDim.x * kThreads);
}

extern "C" int fnp_rankgrid_clear_summary(int count, const fnp_rankgrid *grids, fnp_stream_t stream) {
    if (count <= 0 || count > 8 || !grids) return FNP_ERR_ARG;
    ClearJobs jobs{};
    long long units_max = 1;
    for (int i = 0; i < count; ++i) {
        if (!fnp_rg_valid(&grids[i])) return FNP_ERR_ARG;
        jobs.j[i].g = fnp_rg_view(&grids[i]);
        const long long u = (jobs.j[i].g.nsum + 63) >> 6;
        if (u > units_max) units_max = u;
    }
    const int gx = (fnp_grid_for(units_max, kThreads / 64, 2048) + 7) & ~7;
    hipLaunchKernelGGL(rg_clear_summary_kernel, dim3(gx, count), dim3(kThreads), 0, (hipStream_t)stream, jobs);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// ---- host entry point ---------------------------------------------------------------------------------------------
// The dataloader-side voxeliser (spconv.utils.Point2VoxelCPU3d.point_to_voxel as DataProcessor calls it inside
// DataLoader workers, pcdet/datasets/processor/data_processor.py:38-61,255-302): a plain host loop with the
// reference's own sequential first-come semantics, no device, no stream — safe in forked worker processes, where a
// GPU context cannot be (re)created.  A hash map over the linear cell id stands in for spconv's dense
// coordinate -> voxel table (340 MB for a 41 x 1440 x 1440 grid).  Same coordinate arithmetic as vox_mark_kernel:
// f32 subtract, f32 divide, floor.  Returns the number of voxels, or a negative error code.
#include <vector>

extern "C" int fnp_host_voxelize(const float *points, int n, const fnp_voxel_cfg *cfg, float *voxels, int *coords,
                                 int *num_points, int max_rows) {
    if (!cfg || n < 0 || max_rows < 0 || cfg->num_features < 3 || cfg->max_points <= 0 || (n > 0 && !points)) return FNP_ERR_ARG;
    if (n > 0 && max_rows > 0 && (!voxels || !coords || !num_points)) return FNP_ERR_ARG;
    const int C = cfg->num_features, P = cfg->max_points;
    const int cap_rows = cfg->max_voxels < max_rows ? cfg->max_voxels : max_rows;
    size_t tbl = 64;
    while (tbl < (size_t)(n > 0 ? n : 1) * 2) tbl <<= 1;
    std::vector<long long> keys(tbl, -1ll);
    std::vector<int> vals(tbl, 0);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = points + (size_t)i * C;
        int c[3];
        bool ok = true;
        for (int j = 0; j < 3; ++j) {
            const float q = (p[j] - cfg->range_min[j]) / cfg->voxel_size[j];
            const float f = floorf(q);
            ok = ok && (f >= 0.f) && (f < (float)cfg->grid[j]);
            c[j] = (int)f;
        }
        if (!ok) continue;
        const long long key = ((long long)c[2] * cfg->grid[1] + c[1]) * cfg->grid[0] + c[0];
        size_t h = (size_t)((unsigned long long)key * 0x9E3779B97F4A7C15ull) & (tbl - 1);
        while (keys[h] != -1ll && keys[h] != key) h = (h + 1) & (tbl - 1);
        int row;
        if (keys[h] == key) {
            row = vals[h];
        } else {
            if (m >= cap_rows) continue;             // `continue` of spconv >= 1.2 / 2.x
            row = m++;
            keys[h] = key;
            vals[h] = row;
            coords[(size_t)row * 3] = c[2];
            coords[(size_t)row * 3 + 1] = c[1];
            coords[(size_t)row * 3 + 2] = c[0];
            num_points[row] = 0;
            for (int k = 0; k < P * C; ++k) voxels[(size_t)row * P * C + k] = 0.f;
        }
        if (num_points[row] < P) {
            float *dst = voxels + ((size_t)row * P + num_points[row]) * C;
            for (int k = 0; k < C; ++k) dst[k] = p[k];
            num_points[row]++;
        }
    }
    return m;
}
