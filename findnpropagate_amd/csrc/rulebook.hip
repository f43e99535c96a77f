// Rulebooks (kernel maps) for SubMConv3d / SparseConv3d on the rank grid.
//
// Replaces spconv's indice-pair generation (call sites
// pcdet/models/backbones_3d/spconv_backbone.py:12-17,39-46,193-234; semantics SURVEY.md
// Appendix A.3-A.4).  Layout is output-stationary: nbr[k*cap + o] = input row that feeds output
// row o through kernel offset k = (kz*kH + ky)*kW + kx, or -1.  The convolution kernels walk it
// without atomics, so results are bit-reproducible run to run (the reference's gather-GEMM-
// scatter sums in an order set by atomics / pair order).
//
//   SubM     : out rows == in rows; input cell = o + (kappa - k/2).
//   strided  : input cell = o*s - p + kappa; the output site set is marked in the output rank
//              grid by the inputs (atomicOr), ranked by a popcount scan, and the coordinate list
//              is emitted in rank order — a deterministic, spatially blocked row order (spconv
//              leaves the output order implementation-defined).
#include "rankgrid.h"
#include "tilerb.h"

namespace {

constexpr int kThreads = 256;

struct Geom {
    int k[3], s[3], p[3];
};

__device__ __forceinline__ void unpack_k(int kidx, const Geom &ge, int &kz, int &ky, int &kx) {
    kx = kidx % ge.k[2];
    const int t = kidx / ge.k[2];
    ky = t % ge.k[1];
    kz = t / ge.k[1];
}

// grid (blocks over rows, K)
__global__ __launch_bounds__(kThreads) void subm_nbr_kernel(const int *__restrict__ coords, const int *__restrict__ n_rows,
                                                            int cap, RG g, Geom ge, int *__restrict__ nbr) {
    const int n = min(*n_rows, cap);
    const int kidx = blockIdx.y;
    int kz, ky, kx;
    unpack_k(kidx, ge, kz, ky, kx);
    const int dz = kz - ge.k[0] / 2, dy = ky - ge.k[1] / 2, dx = kx - ge.k[2] / 2;
    for (int o = blockIdx.x * kThreads + threadIdx.x; o < n; o += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[o];
        const int z = c.y + dz, y = c.z + dy, x = c.w + dx;
        int r = -1;
        if (z >= 0 && z < g.d.D && y >= 0 && y < g.d.H && x >= 0 && x < g.d.W) r = rg_lookup(g, c.x, z, y, x);
        nbr[(size_t)kidx * cap + o] = r;
    }
}

// one thread per input row: mark every output cell it feeds (at most ceil(k/s)^3, 8 for k3 s2)
__global__ __launch_bounds__(kThreads) void strided_mark_kernel(const int *__restrict__ in_coords,
                                                                const int *__restrict__ n_in, int cap_in, RG go,
                                                                Geom ge) {
    const int n = min(*n_in, cap_in);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(in_coords)[i];
        for (int kz = 0; kz < ge.k[0]; ++kz) {
            const int tz = c.y + ge.p[0] - kz;
            if (tz < 0 || tz % ge.s[0]) continue;
            const int oz = tz / ge.s[0];
            if (oz >= go.d.D) continue;
            for (int ky = 0; ky < ge.k[1]; ++ky) {
                const int ty = c.z + ge.p[1] - ky;
                if (ty < 0 || ty % ge.s[1]) continue;
                const int oy = ty / ge.s[1];
                if (oy >= go.d.H) continue;
                for (int kx = 0; kx < ge.k[2]; ++kx) {
                    const int tx = c.w + ge.p[2] - kx;
                    if (tx < 0 || tx % ge.s[2]) continue;
                    const int ox = tx / ge.s[2];
                    if (ox >= go.d.W) continue;
                    rg_mark(go, rg_block_of(go.d, c.x, oz, oy, ox), rg_bit_of(oz, oy, ox));
                }
            }
        }
    }
}

// Fast path of the marking step for ceil(k/s) <= 2 on every axis (all convolutions of the backbone):
// the <= 8 output cells of an input cell are found in closed form (no loop over the kernel volume),
// and their bits are merged per output block first, so an input costs one atomic per touched block
// (1.3 on average, and none once the bits are there) instead of one per output cell.
struct AxisOut {
    int b0;          // block of the first output
    unsigned s0, s1; // 4-bit sets of the outputs inside block b0 / block b0 + 1
    bool any;
};
template <int S>
__device__ __forceinline__ AxisOut axis_outputs(int i, int k, int s_rt, int p, int dout) {
    const int s = S > 0 ? S : s_rt;
    const int num = i + p - (k - 1);
    const int lo = num <= 0 ? 0 : (num + s - 1) / s;
    const int hi = min((i + p) / s, dout - 1);
    AxisOut a;
    a.any = lo <= hi;
    a.b0 = lo >> 2;
    a.s0 = 1u << (lo & 3);
    a.s1 = 0u;
    if (hi > lo) {   // (hi == lo + 1)
        if ((hi >> 2) == a.b0) a.s0 |= 1u << (hi & 3);
        else a.s1 = 1u << (hi & 3);
    }
    return a;
}
__device__ __forceinline__ unsigned spread4(unsigned s) { return (s & 1u) | ((s & 2u) << 3) | ((s & 4u) << 6) | ((s & 8u) << 9); }
__device__ __forceinline__ unsigned long long spread16(unsigned s) {
    return (unsigned long long)(s & 1u) | ((unsigned long long)(s & 2u) << 15) | ((unsigned long long)(s & 4u) << 30) |
           ((unsigned long long)(s & 8u) << 45);
}
// The output cells one input row (cell c of scene c.x; !valid: none) feeds, marked in the output grid `go`.  Every lane of the
// wave must call it (wave-level merging: inputs in rank-grid order reach one output block from ~30 consecutive rows; the bits
// of equal blocks are OR-ed along the wave — a lane may take in any earlier lane's bits of the same block — and the last lane
// of each run issues the one atomic: same-address atomics serialise in L2).
// (MarkTab, mark_put, mark_tab_init / mark_tab_flush: rankgrid.h — shared with the voxeliser's marking kernel)
template <int SZ, int SY, int SX>
__device__ __forceinline__ void mark2_row(bool valid, const int4 c, const RG &go, const Geom &ge, int lane, MarkTab *tab = nullptr) {
    long long blk0 = -1;          // block of the first outputs (corner 0,0,0) and its bits
    unsigned long long m0 = 0ull;
    if (valid) {
        const AxisOut az = axis_outputs<SZ>(c.y, ge.k[0], ge.s[0], ge.p[0], go.d.D);
        const AxisOut ay = axis_outputs<SY>(c.z, ge.k[1], ge.s[1], ge.p[1], go.d.H);
        const AxisOut ax = axis_outputs<SX>(c.w, ge.k[2], ge.s[2], ge.p[2], go.d.W);
        if (az.any && ay.any && ax.any) {
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const unsigned sz = cz ? az.s1 : az.s0;
                if (!sz) continue;
#pragma unroll
                for (int cy = 0; cy < 2; ++cy) {
                    const unsigned sy = cy ? ay.s1 : ay.s0;
                    if (!sy) continue;
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const unsigned sx = cx ? ax.s1 : ax.s0;
                        if (!sx) continue;
                        const unsigned long long m = (unsigned long long)(sx * spread4(sy)) * spread16(sz);
                        const long long blk = rg_block_of(go.d, c.x, (az.b0 + cz) << 2, (ay.b0 + cy) << 2, (ax.b0 + cx) << 2);
                        if (cz + cy + cx == 0) {
                            blk0 = blk;
                            m0 = m;
                        } else {
                            mark_put(tab, go, blk, m);   // outputs across a block border
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long nb = __shfl_up(blk0, d);
        const unsigned long long nm = __shfl_up(m0, d);
        if (lane >= d && nb == blk0) m0 |= nm;
    }
    const long long nxt = __shfl_down(blk0, 1);
    if (blk0 >= 0 && (lane == 63 || nxt != blk0)) mark_put(tab, go, blk0, m0);
}

// A marking job a rulebook kernel carries along (round 3): while it resolves the neighbours of the rows of stage l it also
// marks, for the same rows (their coordinates are in its registers), the output sites of the strided convolution that
// consumes stage l — the separate marking launch (a chain of dependent loads and atomics per wave) and its second read of
// the coordinates go.  on == 0: nothing to mark.
struct MarkJob {
    RG go;
    Geom ge;
    int on;   // 0 none, 1 stride (2,2,2), 2 stride (2,1,1), 3 any stride with at most two outputs per cell and axis
};
__device__ __forceinline__ void mark_job_row(const MarkJob &mk, bool valid, const int4 c, int lane) {
    if (mk.on == 1) mark2_row<2, 2, 2>(valid, c, mk.go, mk.ge, lane);
    else if (mk.on == 2) mark2_row<2, 1, 1>(valid, c, mk.go, mk.ge, lane);
    else if (mk.on == 3) mark2_row<0, 0, 0>(valid, c, mk.go, mk.ge, lane);
}

template <int SZ, int SY, int SX>
// order: optional rank -> row map of the input grid (stage 1, whose rows are in the voxeliser's first-come
// order): the inputs are then visited in rank order, which is what makes the wave-level merging bite.
__global__ __launch_bounds__(kThreads) void strided_mark2_kernel(const int *__restrict__ in_coords,
                                                                 const int *__restrict__ n_in, int cap_in, RG go, Geom ge,
                                                                 const int *__restrict__ order) {
    __shared__ MarkTab tab;
    const int n = min(*n_in, cap_in);
    const int lane = fnp_lane();
    const int span = order ? cap_in : n;   // ranks of dropped / absent cells map to -1
    const int nround = (span + (int)(gridDim.x * kThreads) - 1) / (int)(gridDim.x * kThreads);
    if (FNP_MARK_TAB) mark_tab_init(&tab, threadIdx.x, kThreads);
    // the row id and coordinates of the NEXT round are requested before this round's dependent chain
    // (occupancy-word read -> atomic -> summary atomic) is walked: two of its five memory round trips overlap
    auto fetch = [&](int it, int &row, int4 &c) {
        const int i = (it * gridDim.x + blockIdx.x) * kThreads + threadIdx.x;   // (NOT XCD-contiguous: the marking kernels' atomics ran 8-13 % slower that way, round 5)
        row = (it < nround && i < span) ? i : -1;
        if (order && row >= 0) row = order[row];
        if (!(row >= 0 && row < n)) row = -1;
        c = make_int4(0, 0, 0, 0);
        if (row >= 0) c = reinterpret_cast<const int4 *>(in_coords)[row];
    };
    int row_n;
    int4 c_n;
    fetch(0, row_n, c_n);
    for (int it = 0; it < nround; ++it) {   // (whole waves stay in the loop: the shuffles need them)
        const int row = row_n;
        const int4 c = c_n;
        fetch(it + 1, row_n, c_n);
        mark2_row<SZ, SY, SX>(row >= 0, c, go, ge, lane, FNP_MARK_TAB ? &tab : nullptr);
        if (FNP_MARK_TAB) mark_tab_flush(&tab, go, threadIdx.x, kThreads);   // (the passes of a workgroup lie far apart: nothing to merge across them)
    }
}

// grid (blocks over output rows, K)
__global__ __launch_bounds__(kThreads) void strided_nbr_kernel(const int *__restrict__ out_coords,
                                                               const int *__restrict__ n_out, int cap_out, RG gi,
                                                               Geom ge, int *__restrict__ nbr) {
    const int n = min(*n_out, cap_out);
    const int kidx = blockIdx.y;
    int kz, ky, kx;
    unpack_k(kidx, ge, kz, ky, kx);
    for (int o = blockIdx.x * kThreads + threadIdx.x; o < n; o += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(out_coords)[o];
        const int z = c.y * ge.s[0] - ge.p[0] + kz, y = c.z * ge.s[1] - ge.p[1] + ky, x = c.w * ge.s[2] - ge.p[2] + kx;
        int r = -1;
        if (z >= 0 && z < gi.d.D && y >= 0 && y < gi.d.H && x >= 0 && x < gi.d.W) r = rg_lookup(gi, c.x, z, y, x);
        nbr[(size_t)kidx * cap_out + o] = r;
    }
}

// MASKS: also write rowmask[o] = which of the K offsets row o has a neighbour at (the class sort of the 128-channel layers)
template <int KZ, int KY, int KX, bool MASKS = false>
__global__ __launch_bounds__(kThreads) void subm_nbr_row_kernel(const int *__restrict__ coords, const int *__restrict__ n_rows,
                                                                int cap, RG g, int *__restrict__ nbr, unsigned *__restrict__ rowmask = nullptr,
                                                                MarkJob mk = MarkJob{}) {
    constexpr int K = KZ * KY * KX;
    __shared__ __attribute__((aligned(16))) int strips[kThreads / 64][K * 64];
    int *strip_wave = strips[threadIdx.x >> 6];
    const int n = min(*n_rows, cap);
    for (int base = blockIdx.x * kThreads; base < n; base += gridDim.x * kThreads) {   // (whole waves stay in the loop)
        const int o = base + threadIdx.x;
        int4 cm = make_int4(0, 0, 0, 0);
        if (o < n) cm = reinterpret_cast<const int4 *>(coords)[o];
        if (mk.on) mark_job_row(mk, o < n, cm, fnp_lane());
        if (o < n) {
            const int4 c = cm;
            if constexpr (MASKS) {
                unsigned msk;
                nbr_row<KZ, KY, KX>(g, c.x, c.y - KZ / 2, c.z - KY / 2, c.w - KX / 2, strip_wave + fnp_lane(), 64, &msk);
                rowmask[o] = msk;
            } else
            nbr_row<KZ, KY, KX>(g, c.x, c.y - KZ / 2, c.z - KY / 2, c.w - KX / 2, strip_wave + fnp_lane(), 64);
        }
        nbr_flush<K>(strip_wave, base + (threadIdx.x & ~63), n, cap, nbr);
    }
}

// subm_nbr_row_kernel<3, 3, 3> that also writes the TILE RULEBOOK of the table (tilerb.h; same records as
// fnp_tile_rulebook_build makes from the table afterwards, without reading the table back): the 256 rows a workgroup
// resolves per pass are one 256-row tile or two 128-row tiles, and the entries are restated from the LDS strips the int32
// rows are flushed from.
// LEAN: the int32 table is written only for the rows of tiles that hold an ESCAPE entry — the only rows the tiled
// convolutions ever look up in it (rank-ordered inputs: ~1e-5 of the tiles; the 108 bytes per row of the full table were
// two thirds of this kernel's stores).  For a caller whose every consumer of the table is fnp_spconv_forward_tiled.
// NT: threads per workgroup = rows per pass (a multiple of the tile: the wide 512-row tiles take 512)
template <typename G, bool LEAN = false, int NT = kThreads>
__global__ __launch_bounds__(NT) void subm_nbr_row_tile_kernel(const int *__restrict__ coords, const int *__restrict__ n_rows, int cap,
                                                                     RG g, int *__restrict__ nbr, unsigned char *__restrict__ tile_rb,
                                                                     MarkJob mk = MarkJob{}, int *__restrict__ esc_groups = nullptr) {
    // esc_groups (nullable): += the number of 32-row groups whose entries hold an ESCAPE — the statistic a caller gates the
    // tiled convolutions on (rank-ordered single-sweep lidar: ~1e-5 of the groups; 10-sweep density: 2 % / 11 % at 32 / 64 channels)
    constexpr int K = tilerb::kK, TPP = NT / G::TILE;   // tiles per pass
    static_assert(NT % G::TILE == 0 && TPP * G::OVF <= NT, "at most one table slot per thread");
    __shared__ __attribute__((aligned(16))) int strips[NT / 64][K * 64];
    __shared__ int table[TPP][G::OVF];
    __shared__ int esc[NT / 32];
    __shared__ unsigned short lut[G::WIN + 2];
    int *strip_wave = strips[threadIdx.x >> 6];
    const int n = min(*n_rows, cap), tid = threadIdx.x, lane = fnp_lane();
    tilerb::fill_lut<G>(lut, tid, NT);
    // (XCD-contiguous workgroup order — common.h — for the 64-channel geometry only: measured -8 % there and +14 % on the 32-channel
    //  one, twice, on rocprofv3 averages of the 128-scene step: round 5)
    const int blk0 = G::ROWB == 128 ? (int)fnp_xcd_block() : (int)blockIdx.x;
    for (int base = blk0 * NT; base < n; base += gridDim.x * NT) {   // (whole workgroups stay in the loop)
        const int o = base + tid;
        if (tid < TPP * G::OVF) (&table[0][0])[tid] = -1;
        if (tid < NT / 32) esc[tid] = 0;
        int4 cm = make_int4(0, 0, 0, 0);
        if (o < n) cm = reinterpret_cast<const int4 *>(coords)[o];
        if (mk.on) mark_job_row(mk, o < n, cm, lane);
        if (o < n) {
            const int4 c = cm;
            nbr_row<3, 3, 3, false>(g, c.x, c.y - 1, c.z - 1, c.w - 1, strip_wave + lane, 64);
        }
        if constexpr (!LEAN) nbr_flush<K>(strip_wave, base + (tid & ~63), n, cap, nbr);
        __syncthreads();   // empty tables
        const int tl = tid / G::TILE, r = tid % G::TILE, tile = base / G::TILE + tl;
        int id_keep[LEAN ? K : 1];   // (LEAN: the row's int32 entries, kept for the rare tile that needs them in the table)
        const int wlo = max(0, tile * G::TILE - G::HALO);
        unsigned char *rec = tile_rb + (size_t)tile * G::REC;
        if (tile * G::TILE < n) {
            int id[K];
            unsigned code[K];
#pragma unroll
            for (int k = 0; k < K; ++k) id[k] = o < n ? strip_wave[k * 64 + lane] : -1;
            if constexpr (LEAN) {
#pragma unroll
                for (int k = 0; k < K; ++k) id_keep[k] = id[k];
            }
#ifdef FNP_RBT_ABLATE
            bool any_esc = false;
            if (FNP_RBT_ABLATE & 1) {
#pragma unroll
                for (int k = 0; k < K; ++k) code[k] = (unsigned)id[k];
            } else any_esc = tilerb::entries_of_row<G>(id, wlo, table[tl], lut, code);
#else
            const bool any_esc = tilerb::entries_of_row<G>(id, wlo, table[tl], lut, code);
#endif
            // the entries take the place of the wave's int32 strip (flushed above, read into id[]): [offset][row of the wave]
            unsigned short *cw = reinterpret_cast<unsigned short *>(strip_wave);
#pragma unroll
            for (int k = 0; k < K; ++k) cw[k * 64 + lane] = (unsigned short)code[k];
            if (any_esc) esc[tid >> 5] = 1;
        }
        __syncthreads();   // every far row has its slot, every entry is in the strips
        // the entries leave 16 bytes (8 rows of one offset) per lane
        constexpr int CPK = G::TILE / 8, CPT = K * CPK;   // chunks per offset, per tile
        for (int c = tid; c < TPP * CPT; c += NT) {
            const int tlc = c / CPT, cc = c % CPT, k = cc / CPK, r8 = tlc * G::TILE + (cc % CPK) * 8;   // r8: first row, inside the pass
            const int tc = base / G::TILE + tlc;
#ifdef FNP_RBT_ABLATE
            if (FNP_RBT_ABLATE & 2) continue;
#endif
            if (tc * G::TILE < n)
                reinterpret_cast<uint4 *>(tile_rb + (size_t)tc * G::REC)[cc] =
                    *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned short *>(strips[r8 >> 6]) + k * 64 + (r8 & 63));
        }
        if (tile * G::TILE < n) {
            if (r < G::OVF) reinterpret_cast<int *>(rec + G::REC_FAR)[r] = table[tl][r];
            if (r < 16) rec[G::REC_ESC + r] = r < G::TILE / 32 ? (unsigned char)esc[tl * (G::TILE / 32) + r] : 0;
            if (esc_groups && r < G::TILE / 32 && esc[tl * (G::TILE / 32) + r]) atomicAdd(esc_groups, 1);
            if constexpr (LEAN) {
                int any = 0;
#pragma unroll
                for (int q = 0; q < G::TILE / 32; ++q) any |= esc[tl * (G::TILE / 32) + q];
                if (any && o < n) {   // (uniform per tile) this tile's rows go to the table after all
#pragma unroll
                    for (int k = 0; k < K; ++k) nbr[(size_t)k * cap + o] = id_keep[k];
                }
            }
        }
        __syncthreads();   // (the next pass clears the tables and rewrites the strips)
    }
}

template <int KZ, int KY, int KX>
__global__ __launch_bounds__(kThreads) void strided_nbr_row_kernel(const int *__restrict__ out_coords,
                                                                   const int *__restrict__ n_out, int cap_out, RG gi, Geom ge,
                                                                   int *__restrict__ nbr) {
    constexpr int K = KZ * KY * KX;
    __shared__ __attribute__((aligned(16))) int strips[kThreads / 64][K * 64];
    int *strip_wave = strips[threadIdx.x >> 6];
    const int n = min(*n_out, cap_out);
    for (int base = blockIdx.x * kThreads; base < n; base += gridDim.x * kThreads) {
        const int o = base + threadIdx.x;
        if (o < n) {
            const int4 c = reinterpret_cast<const int4 *>(out_coords)[o];
            nbr_row<KZ, KY, KX>(gi, c.x, c.y * ge.s[0] - ge.p[0], c.z * ge.s[1] - ge.p[1], c.w * ge.s[2] - ge.p[2], strip_wave + fnp_lane(), 64);
        }
        nbr_flush<K>(strip_wave, base + (threadIdx.x & ~63), n, cap_out, nbr);
    }
}

bool geom_ok(const fnp_conv_geom *g) {
    if (!g) return false;
    for (int d = 0; d < 3; ++d)
        if (g->ksize[d] <= 0 || g->stride[d] <= 0 || g->padding[d] < 0 || g->in_shape[d] <= 0 || g->out_shape[d] <= 0)
            return false;
    return (long long)g->ksize[0] * g->ksize[1] * g->ksize[2] <= 343;
}

Geom to_geom(const fnp_conv_geom *g) {
    Geom r;
    for (int d = 0; d < 3; ++d) {
        r.k[d] = g->ksize[d];
        r.s[d] = g->stride[d];
        r.p[d] = g->padding[d];
    }
    return r;
}

}  // namespace

static bool shape_is(const fnp_rankgrid *g, const int *shape) {
    return g->D == shape[0] && g->H == shape[1] && g->W == shape[2];
}

// mark_grid / mark_geom of a rulebook entry point -> the job its kernel carries (both NULL: none).  The rows whose rulebook is
// being built are the INPUT sites of the strided convolution `mark_geom` (in_shape = their grid), `mark_grid` its output grid
// (all zero, as fnp_rulebook_strided expects it).  Returns false for a geometry the closed-form marking does not cover.
static bool make_mark_job(const fnp_rankgrid *rows_grid, const fnp_rankgrid *mark_grid, const fnp_conv_geom *mark_geom, MarkJob &mk) {
    mk = MarkJob{};
    if (!mark_grid && !mark_geom) return true;
    if (!mark_grid || !geom_ok(mark_geom) || !fnp_rg_valid(mark_grid)) return false;
    if (mark_grid->B != rows_grid->B || !shape_is(rows_grid, mark_geom->in_shape) || !shape_is(mark_grid, mark_geom->out_shape)) return false;
    for (int d = 0; d < 3; ++d) {
        const int expect = (mark_geom->in_shape[d] + 2 * mark_geom->padding[d] - mark_geom->ksize[d]) / mark_geom->stride[d] + 1;
        if (expect != mark_geom->out_shape[d] || (mark_geom->ksize[d] + mark_geom->stride[d] - 1) / mark_geom->stride[d] > 2) return false;
    }
    mk.go = fnp_rg_view(mark_grid);
    mk.go.perm = nullptr;
    mk.go.ctr = nullptr;   // (marks carried by a rulebook kernel are not counted: fnp_rulebook_strided_premarked ranks with the three-launch prefix)
    mk.ge = to_geom(mark_geom);
    const int *st = mark_geom->stride;
    mk.on = (st[0] == 2 && st[1] == 2 && st[2] == 2) ? 1 : (st[0] == 2 && st[1] == 1 && st[2] == 1) ? 2 : 3;
    return true;
}

extern "C" int fnp_rulebook_subm(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom,
                                 const fnp_rankgrid *grid, int *nbr, fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || !geom_ok(geom) || !fnp_rg_valid(grid) || !nbr) return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (!(geom->ksize[d] & 1) || geom->in_shape[d] != geom->out_shape[d]) return FNP_ERR_ARG;
    if (!shape_is(grid, geom->in_shape)) return FNP_ERR_ARG;
    const int K = geom->ksize[0] * geom->ksize[1] * geom->ksize[2];
    if (geom->ksize[0] == 3 && geom->ksize[1] == 3 && geom->ksize[2] == 3) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_nbr_row_kernel<3, 3, 3>), dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0,
                           (hipStream_t)stream, coords, n_rows, cap, fnp_rg_view(grid), nbr);
    } else {
        dim3 blocks(fnp_grid_for(cap, kThreads, 1024), K);
        hipLaunchKernelGGL(subm_nbr_kernel, blocks, dim3(kThreads), 0, (hipStream_t)stream, coords, n_rows, cap,
                           fnp_rg_view(grid), to_geom(geom), nbr);
    }
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_rulebook_subm_masked(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom, const fnp_rankgrid *grid,
                                        int *nbr, unsigned *rowmask, const fnp_rankgrid *mark_grid, const fnp_conv_geom *mark_geom,
                                        fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || !nbr || !rowmask || !geom_ok(geom) || !fnp_rg_valid(grid)) return FNP_ERR_ARG;
    if (!shape_is(grid, geom->in_shape)) return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (geom->ksize[d] != 3 || geom->in_shape[d] != geom->out_shape[d]) return FNP_ERR_ARG;
    MarkJob mk;
    if (!make_mark_job(grid, mark_grid, mark_geom, mk)) return FNP_ERR_ARG;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_nbr_row_kernel<3, 3, 3, true>), dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, coords,
                       n_rows, cap, fnp_rg_view(grid), nbr, rowmask, mk);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_rulebook_subm_tiled_lean(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom, const fnp_rankgrid *grid,
                                            int *nbr, int channels, void *tile_rb, const fnp_rankgrid *mark_grid,
                                            const fnp_conv_geom *mark_geom, int *escape_groups, fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || !nbr || !tile_rb || !geom_ok(geom) || !fnp_rg_valid(grid)) return FNP_ERR_ARG;
    if (!shape_is(grid, geom->in_shape) || ((uintptr_t)tile_rb & 15) || (channels != 32 && channels != 64)) return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (geom->ksize[d] != 3) return FNP_ERR_ARG;
    MarkJob mk;
    if (!make_mark_job(grid, mark_grid, mark_geom, mk)) return FNP_ERR_ARG;
    if (channels == 32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_nbr_row_tile_kernel<tilerb::G32, true>), dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0,
                           (hipStream_t)stream, coords, n_rows, cap, fnp_rg_view(grid), nbr, (unsigned char *)tile_rb, mk, escape_groups);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_nbr_row_tile_kernel<tilerb::G64, true>), dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0,
                           (hipStream_t)stream, coords, n_rows, cap, fnp_rg_view(grid), nbr, (unsigned char *)tile_rb, mk, escape_groups);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_rulebook_subm_tiled(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom, const fnp_rankgrid *grid,
                                       int *nbr, int channels, void *tile_rb, fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || !nbr || !tile_rb || !geom_ok(geom) || !fnp_rg_valid(grid)) return FNP_ERR_ARG;
    if (!shape_is(grid, geom->in_shape) || ((uintptr_t)tile_rb & 15) || (channels != 32 && channels != 64)) return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (geom->ksize[d] != 3) return FNP_ERR_ARG;
    if (channels == 32)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_nbr_row_tile_kernel<tilerb::G32>), dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0,
                           (hipStream_t)stream, coords, n_rows, cap, fnp_rg_view(grid), nbr, (unsigned char *)tile_rb);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_nbr_row_tile_kernel<tilerb::G64>), dim3(fnp_grid_for(cap, kThreads)), dim3(kThreads), 0,
                           (hipStream_t)stream, coords, n_rows, cap, fnp_rg_view(grid), nbr, (unsigned char *)tile_rb);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

static int rulebook_strided_impl(const int *in_coords, const int *n_in, int cap_in, const fnp_conv_geom *geom,
                                 const fnp_rankgrid *in_grid, const fnp_rankgrid *out_grid, int *out_coords,
                                 int *n_out, int cap_out, int *nbr, void *workspace, int64_t workspace_bytes,
                                 fnp_stream_t stream, bool premarked);

extern "C" int fnp_rulebook_strided(const int *in_coords, const int *n_in, int cap_in, const fnp_conv_geom *geom,
                                    const fnp_rankgrid *in_grid, const fnp_rankgrid *out_grid, int *out_coords,
                                    int *n_out, int cap_out, int *nbr, void *workspace, int64_t workspace_bytes,
                                    fnp_stream_t stream) {
    return rulebook_strided_impl(in_coords, n_in, cap_in, geom, in_grid, out_grid, out_coords, n_out, cap_out, nbr, workspace, workspace_bytes, stream, false);
}

extern "C" int fnp_rulebook_strided_premarked(const int *in_coords, const int *n_in, int cap_in, const fnp_conv_geom *geom,
                                              const fnp_rankgrid *in_grid, const fnp_rankgrid *out_grid, int *out_coords,
                                              int *n_out, int cap_out, int *nbr, void *workspace, int64_t workspace_bytes,
                                              fnp_stream_t stream) {
    return rulebook_strided_impl(in_coords, n_in, cap_in, geom, in_grid, out_grid, out_coords, n_out, cap_out, nbr, workspace, workspace_bytes, stream, true);
}

static int rulebook_strided_impl(const int *in_coords, const int *n_in, int cap_in, const fnp_conv_geom *geom,
                                 const fnp_rankgrid *in_grid, const fnp_rankgrid *out_grid, int *out_coords,
                                 int *n_out, int cap_out, int *nbr, void *workspace, int64_t workspace_bytes,
                                 fnp_stream_t stream, bool premarked) {
    hipStream_t s = (hipStream_t)stream;
    if (!in_coords || !n_in || cap_in <= 0 || cap_out <= 0 || !geom_ok(geom) || !fnp_rg_valid(in_grid) ||
        !fnp_rg_valid(out_grid) || !out_coords || !n_out || !workspace)
        return FNP_ERR_ARG;
    if (in_grid->B != out_grid->B || !shape_is(in_grid, geom->in_shape) || !shape_is(out_grid, geom->out_shape))
        return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d) {
        const int expect = (geom->in_shape[d] + 2 * geom->padding[d] - geom->ksize[d]) / geom->stride[d] + 1;
        if (expect != geom->out_shape[d]) return FNP_ERR_ARG;
    }
    const int K = geom->ksize[0] * geom->ksize[1] * geom->ksize[2];
    const RG gi = fnp_rg_view(in_grid);
    RG go = fnp_rg_view(out_grid);
    go.perm = nullptr;
    if (fnp_scan::rank_grid_workspace_bytes(go.nsum) > workspace_bytes) return FNP_ERR_WORKSPACE;
    const Geom ge = to_geom(geom);

    bool two = true;   // at most two outputs per input cell and axis
    for (int d = 0; d < 3; ++d) two = two && (ge.k[d] + ge.s[d] - 1) / ge.s[d] <= 2;
    // counted marks (rankgrid.h): the closed-form marking kernels add the cells they set to the output grid's counters and the
    // prefix is one launch; the generic marking loop and marks carried by a rulebook kernel leave the counting to the prefix
    const bool counted = go.ctr != nullptr && two && !premarked;
    if (!counted) go.ctr = nullptr;
    const dim3 mgrid(fnp_grid_for(cap_in, kThreads));
    if (premarked) {
        // (the output sites were marked by the kernel that built the rulebook of the input rows: fnp_rulebook_subm_masked /
        //  fnp_rulebook_subm_tiled_lean with mark_grid = out_grid, mark_geom = geom)
    } else if (two && ge.s[0] == 2 && ge.s[1] == 2 && ge.s[2] == 2)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(strided_mark2_kernel<2, 2, 2>), mgrid, dim3(kThreads), 0, s, in_coords, n_in, cap_in, go, ge, (const int *)gi.perm);
    else if (two && ge.s[0] == 2 && ge.s[1] == 1 && ge.s[2] == 1)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(strided_mark2_kernel<2, 1, 1>), mgrid, dim3(kThreads), 0, s, in_coords, n_in, cap_in, go, ge, (const int *)gi.perm);
    else if (two)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(strided_mark2_kernel<0, 0, 0>), mgrid, dim3(kThreads), 0, s, in_coords, n_in, cap_in, go, ge, (const int *)gi.perm);
    else
        hipLaunchKernelGGL(strided_mark_kernel, mgrid, dim3(kThreads), 0, s, in_coords, n_in, cap_in, go, ge);
    FNP_LAUNCH_CHECK();
    int rc = fnp_scan::rank_grid(go, n_out, workspace, s, out_coords, cap_out, counted);   // ranks + coordinates in rank order
    if (rc) return rc;
    if (!nbr) return FNP_OK;   // grid + coordinates only (the convolution resolves its neighbours itself)
    const dim3 rgrid(fnp_grid_for(cap_out, kThreads));
    if (ge.k[0] == 3 && ge.k[1] == 3 && ge.k[2] == 3)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(strided_nbr_row_kernel<3, 3, 3>), rgrid, dim3(kThreads), 0, s, out_coords, n_out,
                           cap_out, gi, ge, nbr);
    else if (ge.k[0] == 3 && ge.k[1] == 1 && ge.k[2] == 1)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(strided_nbr_row_kernel<3, 1, 1>), rgrid, dim3(kThreads), 0, s, out_coords, n_out,
                           cap_out, gi, ge, nbr);
    else
        hipLaunchKernelGGL(strided_nbr_kernel, dim3(fnp_grid_for(cap_out, kThreads, 1024), K), dim3(kThreads), 0, s,
                           out_coords, n_out, cap_out, gi, ge, nbr);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
