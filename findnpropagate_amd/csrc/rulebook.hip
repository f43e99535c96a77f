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
#include "rankgrid.cuh"

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
                                                            int cap, RankGridDims g, Geom ge,
                                                            const unsigned long long *__restrict__ bits,
                                                            const unsigned *__restrict__ base,
                                                            const int *__restrict__ perm, int *__restrict__ nbr) {
    const int n = min(*n_rows, cap);
    const int kidx = blockIdx.y;
    int kz, ky, kx;
    unpack_k(kidx, ge, kz, ky, kx);
    const int dz = kz - ge.k[0] / 2, dy = ky - ge.k[1] / 2, dx = kx - ge.k[2] / 2;
    for (int o = blockIdx.x * kThreads + threadIdx.x; o < n; o += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[o];
        const int z = c.y + dz, y = c.z + dy, x = c.w + dx;
        int r = -1;
        if (z >= 0 && z < g.D && y >= 0 && y < g.H && x >= 0 && x < g.W) r = rg_lookup(g, bits, base, perm, c.x, z, y, x);
        nbr[(size_t)kidx * cap + o] = r;
    }
}

// grid (blocks over input rows, K): mark output cells
__global__ __launch_bounds__(kThreads) void strided_mark_kernel(const int *__restrict__ in_coords,
                                                                const int *__restrict__ n_in, int cap_in,
                                                                RankGridDims go, Geom ge,
                                                                unsigned long long *__restrict__ out_bits) {
    const int n = min(*n_in, cap_in);
    int kz, ky, kx;
    unpack_k(blockIdx.y, ge, kz, ky, kx);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(in_coords)[i];
        const int tz = c.y + ge.p[0] - kz, ty = c.z + ge.p[1] - ky, tx = c.w + ge.p[2] - kx;
        if (tz < 0 || ty < 0 || tx < 0) continue;
        if (tz % ge.s[0] || ty % ge.s[1] || tx % ge.s[2]) continue;
        const int oz = tz / ge.s[0], oy = ty / ge.s[1], ox = tx / ge.s[2];
        if (oz >= go.D || oy >= go.H || ox >= go.W) continue;
        atomicOr(&out_bits[rg_block_of(go, c.x, oz, oy, ox)], 1ull << rg_bit_of(oz, oy, ox));
    }
}

// one thread per occupancy word: emit the coordinates of its set bits at their ranks
__global__ __launch_bounds__(kThreads) void emit_coords_kernel(RankGridDims go, long long nblk,
                                                               const unsigned long long *__restrict__ bits,
                                                               const unsigned *__restrict__ base, int cap_out,
                                                               int *__restrict__ out_coords) {
    for (long long w = (long long)blockIdx.x * kThreads + threadIdx.x; w < nblk; w += (long long)gridDim.x * kThreads) {
        unsigned long long m = bits[w];
        if (!m) continue;
        int r = (int)base[w];
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (r < cap_out) {
                int b, z, y, x;
                rg_decode(go, w, bit, b, z, y, x);
                reinterpret_cast<int4 *>(out_coords)[r] = make_int4(b, z, y, x);
            }
            ++r;
        }
    }
}

// grid (blocks over output rows, K)
__global__ __launch_bounds__(kThreads) void strided_nbr_kernel(const int *__restrict__ out_coords,
                                                               const int *__restrict__ n_out, int cap_out,
                                                               RankGridDims gi, Geom ge,
                                                               const unsigned long long *__restrict__ in_bits,
                                                               const unsigned *__restrict__ in_base,
                                                               const int *__restrict__ in_perm, int *__restrict__ nbr) {
    const int n = min(*n_out, cap_out);
    const int kidx = blockIdx.y;
    int kz, ky, kx;
    unpack_k(kidx, ge, kz, ky, kx);
    for (int o = blockIdx.x * kThreads + threadIdx.x; o < n; o += gridDim.x * kThreads) {
        const int4 c = reinterpret_cast<const int4 *>(out_coords)[o];
        const int z = c.y * ge.s[0] - ge.p[0] + kz, y = c.z * ge.s[1] - ge.p[1] + ky, x = c.w * ge.s[2] - ge.p[2] + kx;
        int r = -1;
        if (z >= 0 && z < gi.D && y >= 0 && y < gi.H && x >= 0 && x < gi.W)
            r = rg_lookup(gi, in_bits, in_base, in_perm, c.x, z, y, x);
        nbr[(size_t)kidx * cap_out + o] = r;
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

extern "C" int fnp_rulebook_subm(const int *coords, const int *n_rows, int cap, int B, const fnp_conv_geom *geom,
                                 const uint64_t *grid_bits, const uint32_t *grid_base, const int *grid_perm, int *nbr,
                                 fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || B <= 0 || !geom_ok(geom) || !grid_bits || !grid_base || !nbr) return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (!(geom->ksize[d] & 1) || geom->in_shape[d] != geom->out_shape[d]) return FNP_ERR_ARG;
    const int K = geom->ksize[0] * geom->ksize[1] * geom->ksize[2];
    const RankGridDims g = fnp_make_dims(B, geom->in_shape[0], geom->in_shape[1], geom->in_shape[2]);
    dim3 grid(fnp_grid_for(cap, kThreads, 1024), K);
    hipLaunchKernelGGL(subm_nbr_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, coords, n_rows, cap, g,
                       to_geom(geom), (const unsigned long long *)grid_bits, grid_base, grid_perm, nbr);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_rulebook_strided(const int *in_coords, const int *n_in, int cap_in, int B, const fnp_conv_geom *geom,
                                    const uint64_t *in_bits, const uint32_t *in_base, const int *in_perm,
                                    uint64_t *out_bits, uint32_t *out_base, int *out_coords, int *n_out, int cap_out,
                                    int *nbr, void *workspace, int64_t workspace_bytes, fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!in_coords || !n_in || cap_in <= 0 || cap_out <= 0 || B <= 0 || !geom_ok(geom) || !in_bits || !in_base ||
        !out_bits || !out_base || !out_coords || !n_out || !nbr || !workspace)
        return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d) {
        const int expect = (geom->in_shape[d] + 2 * geom->padding[d] - geom->ksize[d]) / geom->stride[d] + 1;
        if (expect != geom->out_shape[d]) return FNP_ERR_ARG;
    }
    const int K = geom->ksize[0] * geom->ksize[1] * geom->ksize[2];
    const RankGridDims gi = fnp_make_dims(B, geom->in_shape[0], geom->in_shape[1], geom->in_shape[2]);
    const RankGridDims go = fnp_make_dims(B, geom->out_shape[0], geom->out_shape[1], geom->out_shape[2]);
    const long long nblk_out = fnp_num_blocks(go);
    if (fnp_scan::workspace_bytes(nblk_out) > workspace_bytes) return FNP_ERR_WORKSPACE;
    const Geom ge = to_geom(geom);

    hipLaunchKernelGGL(strided_mark_kernel, dim3(fnp_grid_for(cap_in, kThreads, 1024), K), dim3(kThreads), 0, s,
                       in_coords, n_in, cap_in, go, ge, (unsigned long long *)out_bits);
    FNP_LAUNCH_CHECK();
    int rc = fnp_scan::popcount_u64((const unsigned long long *)out_bits, nblk_out, out_base, n_out, workspace, s);
    if (rc) return rc;
    hipLaunchKernelGGL(emit_coords_kernel, dim3(fnp_grid_for(nblk_out, kThreads)), dim3(kThreads), 0, s, go, nblk_out,
                       (const unsigned long long *)out_bits, out_base, cap_out, out_coords);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(strided_nbr_kernel, dim3(fnp_grid_for(cap_out, kThreads, 1024), K), dim3(kThreads), 0, s,
                       out_coords, n_out, cap_out, gi, ge, (const unsigned long long *)in_bits, in_base, in_perm, nbr);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
