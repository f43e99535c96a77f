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

// emit output coordinates in rank order: one wave per summary word (a zero word retires after one
// load); lane j owns block 64*S + j, decodes the block origin once and writes its cells at
// base[block], base[block] + 1, ...
__global__ __launch_bounds__(kThreads) void emit_coords_kernel(RG go, int cap_out, int *__restrict__ out_coords) {
    const int lane = fnp_lane();
    const long long S = ((long long)blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (S >= go.nsum) return;
    const unsigned long long sw = go.summ[S];
    if (!((sw >> lane) & 1ull)) return;
    const long long w = S * 64 + lane;
    unsigned long long m = go.bits[w];
    int r = (int)go.base[w];
    int b, z0, y0, x0;
    rg_decode(go.d, w, 0, b, z0, y0, x0);
    while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (r < cap_out)
            reinterpret_cast<int4 *>(out_coords)[r] = make_int4(b, z0 | (bit >> 4), y0 | ((bit >> 2) & 3), x0 | (bit & 3));
        ++r;
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

extern "C" int fnp_rulebook_subm(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom,
                                 const fnp_rankgrid *grid, int *nbr, fnp_stream_t stream) {
    if (!coords || !n_rows || cap <= 0 || !geom_ok(geom) || !fnp_rg_valid(grid) || !nbr) return FNP_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (!(geom->ksize[d] & 1) || geom->in_shape[d] != geom->out_shape[d]) return FNP_ERR_ARG;
    if (!shape_is(grid, geom->in_shape)) return FNP_ERR_ARG;
    const int K = geom->ksize[0] * geom->ksize[1] * geom->ksize[2];
    dim3 blocks(fnp_grid_for(cap, kThreads, 1024), K);
    hipLaunchKernelGGL(subm_nbr_kernel, blocks, dim3(kThreads), 0, (hipStream_t)stream, coords, n_rows, cap,
                       fnp_rg_view(grid), to_geom(geom), nbr);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_rulebook_strided(const int *in_coords, const int *n_in, int cap_in, const fnp_conv_geom *geom,
                                    const fnp_rankgrid *in_grid, const fnp_rankgrid *out_grid, int *out_coords,
                                    int *n_out, int cap_out, int *nbr, void *workspace, int64_t workspace_bytes,
                                    fnp_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!in_coords || !n_in || cap_in <= 0 || cap_out <= 0 || !geom_ok(geom) || !fnp_rg_valid(in_grid) ||
        !fnp_rg_valid(out_grid) || !out_coords || !n_out || !nbr || !workspace)
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

    hipLaunchKernelGGL(strided_mark_kernel, dim3(fnp_grid_for(cap_in, kThreads)), dim3(kThreads), 0, s, in_coords, n_in,
                       cap_in, go, ge);
    FNP_LAUNCH_CHECK();
    int rc = fnp_scan::rank_grid(go, n_out, workspace, s);
    if (rc) return rc;
    hipLaunchKernelGGL(emit_coords_kernel, dim3(fnp_divup(go.nsum * 64, kThreads)), dim3(kThreads), 0, s, go, cap_out,
                       out_coords);
    FNP_LAUNCH_CHECK();
    hipLaunchKernelGGL(strided_nbr_kernel, dim3(fnp_grid_for(cap_out, kThreads, 1024), K), dim3(kThreads), 0, s,
                       out_coords, n_out, cap_out, gi, ge, nbr);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
