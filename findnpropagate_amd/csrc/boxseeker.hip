// Greedy Box Seeker, fused: one workgroup per frustum (one 2D detection of one camera).
//
// Replaces hot loops 2-4 of FrustumProposerOG.get_proposals
// (pcdet/models/dense_heads/frustum_proposals_v1.py:593-1048), which in the reference are a
// Python triple loop with ~25 tiny torch launches per frustum, three sorts for the depth
// quantiles, a CPU round trip for the 2D IoU and one points_in_boxes_gpu launch + host sync per
// candidate (<= 60 per frustum).  Here a frustum is a single workgroup that
//   A  projects the scene's points into its camera and keeps those inside the 2D box
//      (:590,:606-613; wave ballot + prefix compaction, order is irrelevant downstream),
//   B  finds the depth quantiles lq/uq/cq by radix select on the float bits (:616-629; no sort),
//   C  builds the camera frustum and back-projects it to lidar (:128-140,:1509-1545),
//   D  back-projects the selected points and takes their per-axis extent (:812-826),
//   E  generates the num_mags x num_rotations x num_sizes candidates with the softmin front shift
//      and the max_dist filter (:828-879),
//   F  projects their corners and scores the 2D IoU against the detection (:1392-1411),
//   G  counts the points inside every surviving candidate (:930-932, same test as
//      points_in_boxes.hip: one pass over the points for all candidates),
//   H  scores (:994-999) and keeps the best candidate (topk = 1, :1041-1045).
// Arithmetic follows the reference's f32 expression order (-ffp-contract=off); transcendental
// functions come from ocml, so values agree with the CPU reference to float rounding, not bitwise.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCand = 256;

struct Mat3 { float m[9]; };

__device__ __forceinline__ void mat3_apply(const float *M, float x, float y, float z, float &ox, float &oy, float &oz) {
    ox = M[0] * x + M[1] * y + M[2] * z;
    oy = M[3] * x + M[4] * y + M[5] * z;
    oz = M[6] * x + M[7] * y + M[8] * z;
}

constexpr int kCam = 45;   // floats per (scene, camera): L33 (9) | Lt (3) | combine (9) | c2l_t (3) | post_rot (9) | post_trans (3) | post_rot^-1 (9)

// project_to_camera (:1431-1475) for one point; img_aug: the image-plane augmentation of :1456-1458 is applied
__device__ __forceinline__ void project(const float *aug_inv, const float *aug_t, const float *cam, bool img_aug, float x, float y,
                                        float z, float &u, float &v, float &d) {
    float px, py, pz, qx, qy, qz;
    mat3_apply(aug_inv, x - aug_t[0], y - aug_t[1], z - aug_t[2], px, py, pz);
    mat3_apply(cam, px, py, pz, qx, qy, qz);
    qx += cam[9];
    qy += cam[10];
    qz += cam[11];
    d = fminf(fmaxf(qz, 1e-5f), 1e5f);
    u = qx / d;
    v = qy / d;
    if (img_aug) {
        float a, b, c;
        mat3_apply(cam + 24, u, v, d, a, b, c);
        u = a + cam[33];
        v = b + cam[34];
        d = c + cam[35];
    }
}

// get_geometry_at_image_coords (:1509-1545); img_aug: the post-transformation is undone first (:1525-1527)
__device__ __forceinline__ void backproject(const float *aug_R, const float *aug_t, const float *cam, bool img_aug, float u, float v,
                                            float d, float &x, float &y, float &z) {
    if (img_aug) {
        float a, b, c;
        mat3_apply(cam + 36, u - cam[33], v - cam[34], d - cam[35], a, b, c);
        u = a;
        v = b;
        d = c;
    }
    float cx, cy, cz;
    mat3_apply(cam + 12, u * d, v * d, d, cx, cy, cz);
    cx += cam[21];
    cy += cam[22];
    cz += cam[23];
    mat3_apply(aug_R, cx, cy, cz, x, y, z);
    x += aug_t[0];
    y += aug_t[1];
    z += aug_t[2];
}

struct Shared {
    unsigned hist[256];
    unsigned sel_prefix, sel_mask, sel_k;
    unsigned scan_w[kThreads / 64];
    unsigned count, scratch_u;
    float cam[kCam], aug_R[9], aug_inv[9], aug_t[3];
    float fr[8][3];
    float ext_min[3], ext_max[3];
    float bev_pts[16][3];
    float wc[3];
    float cbox[kMaxCand][7];
    float ccos[kMaxCand], csin[kMaxCand];
    double chx[kMaxCand], chy[kMaxCand];
    float ciou[kMaxCand], cdist[kMaxCand];
    float cwfc[kMaxCand][3];
    int cvalid[kMaxCand];
    int ccount[kMaxCand];
    float cm1[kMaxCand];      // occlusion terms: range of the candidate's nearest corner
    int cbeyond[kMaxCand];    //                  points of the frustum beyond that range
    float cscore[kMaxCand];   // second-stage score (topk > 1: the 3D NMS walks them)
    float red_f[kThreads / 64][6];
    float q[3];
};

__device__ __forceinline__ unsigned fbits(float f) { return __float_as_uint(f); }  // depths are > 0: order preserving

// k-th smallest (0-based) depth of the list: 4 x 8-bit histogram passes on the float bits.
// LAT: the workgroup-parallel bin search (small launches: a frustum's latency is the launch's); without it thread 0 walks the
// bins — ten times slower per pass, but free when four workgroups share a CU and the launch is bound by throughput (64 scenes
// per launch: 56 k scenes/s against 52.6 k with the parallel search; one scene: 1,590 against 1,780 the other way round)
template <bool LAT>
__device__ float select_kth(Shared &S, const float *__restrict__ depth3, int m, unsigned k) {
    if (threadIdx.x == 0) {
        S.sel_prefix = 0;
        S.sel_mask = 0;
        S.sel_k = k;
    }
    for (int shift = 24; shift >= 0; shift -= 8) {
        S.hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned prefix = S.sel_prefix, mask = S.sel_mask;
        for (int i = threadIdx.x; i < m; i += kThreads) {
            const unsigned key = fbits(depth3[(size_t)i * 3 + 2]);
            if ((key & mask) == prefix) atomicAdd(&S.hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if constexpr (!LAT) {
            if (threadIdx.x == 0) {
                unsigned kk = S.sel_k, b = 0;
                for (; b < 256; ++b) {
                    const unsigned h = S.hist[b];
                    if (kk < h) break;
                    kk -= h;
                }
                S.sel_k = kk;
                S.sel_prefix = prefix | (b << shift);
                S.sel_mask = mask | (255u << shift);
            }
        } else {
            // the bin the k-th key falls into: inclusive prefix of the 256 counts across the workgroup (thread = bin), the one
            // thread whose bin straddles k publishes it (round 3: thread 0 walking the bins one LDS read at a time was ~10 us
            // per pass, twelve passes per frustum — a third of the kernel's latency at small batch sizes)
            static_assert(kThreads == 256, "one thread per histogram bin");
            const unsigned h = S.hist[threadIdx.x], kk = S.sel_k;
            unsigned inc = h;
            const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned t = __shfl_up(inc, d);
                if (lane_ >= d) inc += t;
            }
            if (lane_ == 63) S.scan_w[wave_] = inc;
            __syncthreads();
            unsigned off = 0;
            for (int w = 0; w < wave_; ++w) off += S.scan_w[w];
            const unsigned incl = off + inc, excl = incl - h;
            if (kk >= excl && kk < incl) {   // (exactly one thread: k < the number of keys that carry the prefix)
                S.sel_k = kk - excl;
                S.sel_prefix = prefix | ((unsigned)threadIdx.x << shift);
                S.sel_mask = mask | (255u << shift);
            }
        }
        __syncthreads();
    }
    return __uint_as_float(S.sel_prefix);
}

// torch.quantile(depth, q), linear interpolation via torch.lerp
template <bool LAT>
__device__ float quantile(Shared &S, const float *__restrict__ list, int m, float q) {
    const float rank = q * (float)(m - 1);
    const int lo = (int)floorf(rank), hi = (int)ceilf(rank);
    const float a = select_kth<LAT>(S, list, m, (unsigned)lo);
    float b = a;
    if (hi != lo) {
        // (lo+1)-th smallest: a again if a is duplicated past position lo, else the smallest key > a
        if (threadIdx.x == 0) {
            S.count = 0;
            S.scratch_u = 0xffffffffu;
        }
        __syncthreads();
        const unsigned ka = fbits(a);
        unsigned le = 0, mn = 0xffffffffu;
        for (int i = threadIdx.x; i < m; i += kThreads) {
            const unsigned key = fbits(list[(size_t)i * 3 + 2]);
            le += key <= ka;
            if (key > ka) mn = min(mn, key);
        }
        atomicAdd(&S.count, le);
        atomicMin(&S.scratch_u, mn);
        __syncthreads();
        b = ((unsigned)hi < S.count) ? a : __uint_as_float(S.scratch_u);
        __syncthreads();
    }
    const float w = rank - (float)lo;
    return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w);
}

// OPT: the instantiation that carries the options no shipped configuration sets (MULTICAM_IOU, the occlusion terms, search_depth,
// rand_center, topk > 1).  The shipped path compiles them out: with them in one body the candidate loop holds 175 registers
// instead of 126 and the kernel runs two waves per SIMD instead of four (57 k -> 42 k scenes/s at 64 scenes per launch).
template <bool OPT, bool LAT>
__global__ __launch_bounds__(kThreads) void boxseeker_kernel(
    const float *__restrict__ points, const int *__restrict__ scene_off, const fnp_seeker_params prm,
    const float *__restrict__ scene_mats,   // (S, 21): aug_R 9 | aug_inv 9 | aug_t 3
    const float *__restrict__ cam_mats,     // (S, 6, kCam)
    const float *__restrict__ frusts,       // (F, 8): scene, cam, x1, y1, x2, y2, label, score
    const float *__restrict__ base_boxes,   // (10, R, 7)
    const float *__restrict__ base_corners, // (10, R, 8, 3)
    const float *__restrict__ mags,         // (num_mags)
    float *__restrict__ ws_uvd, float *__restrict__ ws_xyz, int ws_stride,
    int *__restrict__ out_valid, float *__restrict__ out_box, float *__restrict__ out_score, int *__restrict__ out_best,
    int *__restrict__ dbg_npts, float *__restrict__ dbg_frust, float *__restrict__ dbg_cand, float *__restrict__ dbg_iou,
    int *__restrict__ dbg_count, int *__restrict__ dbg_valid) {
    __shared__ Shared S;
    const int f = blockIdx.x, tid = threadIdx.x, lane = fnp_lane(), wave = tid >> 6;
    const float *fr = frusts + (size_t)f * 8;
    const int scene = (int)fr[0], cam = (int)fr[1], label = (int)fr[6];
    const float x1 = fr[2], y1 = fr[3], x2 = fr[4], y2 = fr[5];
    const int R = prm.num_rotations * prm.num_sizes, NC = prm.num_mags * R;
    if (tid < kCam) S.cam[tid] = cam_mats[((size_t)scene * 6 + cam) * kCam + tid];
    const bool ia = prm.has_img_aug != 0;
    if (tid < 9) {
        S.aug_R[tid] = scene_mats[(size_t)scene * 21 + tid];
        S.aug_inv[tid] = scene_mats[(size_t)scene * 21 + 9 + tid];
    }
    if (tid < 3) S.aug_t[tid] = scene_mats[(size_t)scene * 21 + 18 + tid];
    if (tid == 0) S.count = 0;
    __syncthreads();

    // ---- A: select the points of this 2D box ------------------------------------------------
    const int p0 = scene_off[scene], p1 = scene_off[scene + 1];
    float *list = ws_uvd + (size_t)f * ws_stride * 3;
    float *lxyz = ws_xyz + (size_t)f * ws_stride * 3;
    // (four passes of points per trip: their loads are in flight together — one load per trip made the scan of a 30 k-point
    //  scene 118 dependent memory round trips; the order of the list is free: quantiles, extents and counts do not see it)
    constexpr int kPB = LAT ? 4 : 1;
    for (int base = p0; base < p1; base += kPB * kThreads) {
        float px[kPB], py[kPB], pz[kPB];
#pragma unroll
        for (int j = 0; j < kPB; ++j) {
            const int i = base + j * kThreads + tid;
            px[j] = py[j] = pz[j] = 0.f;
            if (i < p1) {
                const float *p = points + (size_t)i * prm.point_stride + prm.xyz_offset;
                px[j] = p[0];
                py[j] = p[1];
                pz[j] = p[2];
            }
        }
#pragma unroll
        for (int j = 0; j < kPB; ++j) {
            const int i = base + j * kThreads + tid;
            bool in = false;
            float u = 0, v = 0, d = 0;
            if (i < p1) {
                project(S.aug_inv, S.aug_t, S.cam, ia, px[j], py[j], pz[j], u, v, d);
                const bool on_img = (v < (float)prm.image_h) && (v >= 0.f) && (u < (float)prm.image_w) && (u >= 0.f);
                in = on_img && (v < y2) && (v >= y1) && (u < x2) && (u >= x1);
            }
            const unsigned long long mask = __ballot(in);
            unsigned wbase = 0;
            if (lane == 0 && mask) wbase = atomicAdd(&S.count, (unsigned)__popcll(mask));
            wbase = __shfl(wbase, 0);
            if (in) {
                const unsigned pos = wbase + __popcll(mask & ((1ull << lane) - 1ull));
                list[(size_t)pos * 3 + 0] = u;
                list[(size_t)pos * 3 + 1] = v;
                list[(size_t)pos * 3 + 2] = d;
            }
        }
    }
    __syncthreads();
    const int m = (int)S.count;
    if (dbg_npts && tid == 0) dbg_npts[f] = m;
    if (OPT && prm.count_only) return;   // (the pre-pass of MULTICAM_IOU: which frustums hold points at all)
    if (m == 0) {  // "no pts in box": the frustum is dropped (:634-637); every output gets its empty value (callers need not clear)
        const int tk = OPT ? prm.topk : 1;
        if (tid == 0) out_valid[f] = 0;
        for (int i = tid; i < tk; i += kThreads) {
            out_best[(size_t)f * tk + i] = -1;
            out_score[(size_t)f * tk + i] = 0.f;
        }
        for (int i = tid; i < tk * 7; i += kThreads) out_box[(size_t)f * tk * 7 + i] = 0.f;
        return;
    }
    __threadfence_block();

    // ---- B: depth quantiles -----------------------------------------------------------------
    const float qlo = quantile<LAT>(S, list, m, prm.lq);
    const float qhi = quantile<LAT>(S, list, m, prm.uq);
    const float qc = quantile<LAT>(S, list, m, prm.cq);

    // ---- C: frustum corners in lidar, weighted centre ---------------------------------------
    if (tid < 8) {
        const float far_q = (OPT && prm.search_depth > 0.f) ? qlo + prm.search_depth : qhi;   // :617-622 (search_depth: a fixed depth behind the near quantile)
        float fmax_ = fminf(far_q, prm.max_dist), fmin_ = fmaxf(qlo, 2.0f);        // :647-648
        const float lo3[3] = {x1, y1, fmin_}, hi3[3] = {x2, y2, fmax_};
        const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {1, -1, -1, 1, 1, -1, -1, 1},
                    sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
        float c3[3];
        const float sg[3] = {sx[tid], sy[tid], sz[tid]};
#pragma unroll
        for (int a = 0; a < 3; ++a) c3[a] = (hi3[a] - lo3[a]) * (sg[a] / 2.0f) + (hi3[a] + lo3[a]) / 2.0f;   // :128-140
        backproject(S.aug_R, S.aug_t, S.cam, ia, c3[0], c3[1], c3[2], S.fr[tid][0], S.fr[tid][1], S.fr[tid][2]);
    }
    if (tid == 8)
        backproject(S.aug_R, S.aug_t, S.cam, ia, (x1 + x2) / 2.0f, (y1 + y2) / 2.0f, qc, S.wc[0], S.wc[1], S.wc[2]);   // :630-632

    // ---- D: back-project the selected points, per-axis extent --------------------------------
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < m; i += kThreads) {
        float x, y, z;
        backproject(S.aug_R, S.aug_t, S.cam, ia, list[(size_t)i * 3], list[(size_t)i * 3 + 1], list[(size_t)i * 3 + 2], x, y, z);
        lxyz[(size_t)i * 3] = x;
        lxyz[(size_t)i * 3 + 1] = y;
        lxyz[(size_t)i * 3 + 2] = z;
        mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
        if (lane == 0) {
            S.red_f[wave][a] = mn[a];
            S.red_f[wave][3 + a] = mx[a];
        }
    }
    __syncthreads();
    if (tid < 3) {
        float lo = S.red_f[0][tid], hi = S.red_f[0][3 + tid];
        for (int w = 1; w < kThreads / 64; ++w) {
            lo = fminf(lo, S.red_f[w][tid]);
            hi = fmaxf(hi, S.red_f[w][3 + tid]);
        }
        if (prm.clamp_bottom > 0) {                                                 // :817-826
            float flo = S.fr[0][tid], fhi = S.fr[0][tid];
            for (int c = 1; c < 8; ++c) {
                flo = fminf(flo, S.fr[c][tid]);
                fhi = fmaxf(fhi, S.fr[c][tid]);
            }
            const float f1 = fmaxf(lo, flo), f2 = fminf(hi, fhi);
            for (int c = 0; c < 8; ++c) S.fr[c][tid] = fminf(fmaxf(S.fr[c][tid], f1), f2);   // torch.clamp(min, max)
        }
    }
    __syncthreads();
    if (dbg_frust && tid < 24) dbg_frust[(size_t)f * 24 + tid] = S.fr[tid / 3][tid % 3];

    // ---- E: search positions along the frustum axis (:828-847) -------------------------------
    if (tid < prm.num_mags * 3) {
        const int i = tid / 3, a = tid % 3;
        float close[3], vec[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float b0 = (S.fr[0][e] + S.fr[1][e]) / 2.0f, b1 = (S.fr[2][e] + S.fr[3][e]) / 2.0f;
            const float b2 = (S.fr[4][e] + S.fr[5][e]) / 2.0f, b3 = (S.fr[6][e] + S.fr[7][e]) / 2.0f;
            close[e] = (b0 + b1) / 2.0f;
            vec[e] = (b2 + b3) / 2.0f - close[e];
        }
        if (OPT && prm.search_depth > 0.f) {   // :841-842: the axis at unit length times the depth (a collapsed frustum gives NaN there too)
            const float nrm = sqrtf(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
#pragma unroll
            for (int e = 0; e < 3; ++e) vec[e] = (vec[e] / nrm) * prm.search_depth;
        }
        if (OPT && prm.rand_noise) S.bev_pts[i][a] = S.wc[a] + prm.rand_noise[((size_t)f * prm.num_mags + i) * 3 + a];   // rand_center (:847)
        else S.bev_pts[i][a] = close[a] + vec[a] * mags[i];
    }
    __syncthreads();

    // ---- E/F: candidates, softmin front shift, distance filter, 2D IoU ------------------------
    for (int c = tid; c < NC; c += kThreads) {
        const int mi = c / R, ri = c % R;
        const float *bb = base_boxes + ((size_t)(label - 1) * R + ri) * 7;
        const float *bc = base_corners + ((size_t)(label - 1) * R + ri) * 24;
        float box[7], cor[8][3], nrm[8];
#pragma unroll
        for (int j = 0; j < 7; ++j) box[j] = bb[j];
#pragma unroll
        for (int a = 0; a < 3; ++a) box[a] = box[a] + S.bev_pts[mi][a];
        float nmax = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int a = 0; a < 3; ++a) cor[k][a] = bc[k * 3 + a] + S.bev_pts[mi][a];
            nrm[k] = -sqrtf(cor[k][0] * cor[k][0] + cor[k][1] * cor[k][1] + cor[k][2] * cor[k][2]);
            nmax = fmaxf(nmax, nrm[k]);
        }
        float e[8], es = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            e[k] = expf(nrm[k] - nmax);
            es += e[k];
        }
        float wf[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float r = e[k] / es;
#pragma unroll
            for (int a = 0; a < 3; ++a) wf[a] += r * cor[k][a];
        }
        float f2c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            f2c[a] = box[a] - wf[a];
            box[a] = box[a] + f2c[a];
        }
        const float dist = sqrtf(wf[0] * wf[0] + wf[1] * wf[1] + wf[2] * wf[2]);
        int valid = dist < prm.max_dist;                                            // :871-872
        // calc_iou (:1392-1411): hull box of the clamped corner projections vs the detection
        float bx1 = INFINITY, by1 = INFINITY, bx2 = -INFINITY, by2 = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float u, v, d;
            project(S.aug_inv, S.aug_t, S.cam, ia, cor[k][0] + f2c[0], cor[k][1] + f2c[1], cor[k][2] + f2c[2], u, v, d);
            u = fminf(fmaxf(u, 0.f), (float)prm.image_w);
            v = fminf(fmaxf(v, 0.f), (float)prm.image_h);
            bx1 = fminf(bx1, u); by1 = fminf(by1, v); bx2 = fmaxf(bx2, u); by2 = fmaxf(by2, v);
        }
        const float a1 = (bx2 - bx1) * (by2 - by1), a2 = (x2 - x1) * (y2 - y1);
        const float iw = fmaxf(fminf(bx2, x2) - fmaxf(bx1, x1), 0.f), ih = fmaxf(fminf(by2, y2) - fmaxf(by1, y1), 0.f);
        const float inter = iw * ih;
        float iou = inter / (a1 + a2 - inter);
        if (OPT && prm.multicam) {
            // multicam_ious (:1413-1429): the candidate against EVERY frustum of the scene with this label that holds points
            // (this one included), each in its own camera; mean over the cameras that see it at all
            float tot = 0.f;
            int nz = 0;
            for (int j = 0; j < prm.num_frustums; ++j) {
                const float *fj = frusts + (size_t)j * 8;
                if ((int)fj[0] != scene || (int)fj[6] != label || prm.npts_all[j] <= 0) continue;
                const float *camj = cam_mats + ((size_t)scene * 6 + (int)fj[1]) * kCam;
                float jx1 = INFINITY, jy1 = INFINITY, jx2 = -INFINITY, jy2 = -INFINITY;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float u, v, d;
                    project(S.aug_inv, S.aug_t, camj, ia, cor[k][0] + f2c[0], cor[k][1] + f2c[1], cor[k][2] + f2c[2], u, v, d);
                    u = fminf(fmaxf(u, 0.f), (float)prm.image_w);
                    v = fminf(fmaxf(v, 0.f), (float)prm.image_h);
                    jx1 = fminf(jx1, u); jy1 = fminf(jy1, v); jx2 = fmaxf(jx2, u); jy2 = fmaxf(jy2, v);
                }
                const float ja1 = (jx2 - jx1) * (jy2 - jy1), ja2 = (fj[4] - fj[2]) * (fj[5] - fj[3]);
                const float jw = fmaxf(fminf(jx2, fj[4]) - fmaxf(jx1, fj[2]), 0.f), jh = fmaxf(fminf(jy2, fj[5]) - fmaxf(jy1, fj[3]), 0.f);
                const float ji = jw * jh;
                const float v = ji / (ja1 + ja2 - ji);
                tot = tot + v;
                nz += v > 0.f;
            }
            iou = tot / ((float)nz + 1e-6f);
        }
        if (OPT && (prm.occl_w > 0.f || prm.occl_mult)) {
            // range of the candidate's nearest corner, corners as boxes_to_corners_3d makes them from the final box (:423-426)
            const float ca = cosf(box[6]), sa = sinf(box[6]);
            float m1 = INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float tx = ((k & 3) == 0 || (k & 3) == 1) ? 0.5f : -0.5f, ty = ((k & 3) == 0 || (k & 3) == 3) ? 0.5f : -0.5f, tz = k < 4 ? -0.5f : 0.5f;
                const float lx = box[3] * tx, ly = box[4] * ty, lz = box[5] * tz;
                const float cx = (lx * ca + ly * (-sa)) + box[0], cy = (lx * sa + ly * ca) + box[1], cz = lz + box[2];
                m1 = fminf(m1, sqrtf(cx * cx + cy * cy + cz * cz));
            }
            S.cm1[c] = m1;
        }
        if (OPT) S.cbeyond[c] = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) S.cbox[c][j] = box[j];
#pragma unroll
        for (int a = 0; a < 3; ++a) S.cwfc[c][a] = wf[a];
        S.ciou[c] = iou;
        S.cdist[c] = sqrtf((wf[0] - S.wc[0]) * (wf[0] - S.wc[0]) + (wf[1] - S.wc[1]) * (wf[1] - S.wc[1]) +
                           (wf[2] - S.wc[2]) * (wf[2] - S.wc[2]));
        S.cvalid[c] = valid ? (iou > prm.min_cam_iou ? 2 : 1) : 0;   // 1 = passed the distance filter only
        S.ccount[c] = 0;
        const float ang = -box[6];
        S.ccos[c] = cosf(ang);
        S.csin[c] = sinf(ang);
        S.chx[c] = (double)box[3] / 2.0 + (double)1e-5f;
        S.chy[c] = (double)box[4] / 2.0 + (double)1e-5f;
    }
    __syncthreads();

    // ---- G: points inside every surviving candidate (roiaware_pool3d_kernel.cu:23-36) ---------
    for (int base = 0; base < m; base += kThreads) {
        const int i = base + tid;
        float x = 0, y = 0, z = 0;
        const bool have = i < m;
        if (have) {
            x = lxyz[(size_t)i * 3];
            y = lxyz[(size_t)i * 3 + 1];
            z = lxyz[(size_t)i * 3 + 2];
        }
        const bool occl = OPT && (prm.occl_w > 0.f || prm.occl_mult);
        const float mag = occl ? sqrtf(x * x + y * y + z * z) : 0.f;
        for (int c = 0; c < NC; ++c) {
            if (S.cvalid[c] != 2) continue;   // wave-uniform
            if (occl) {
                const unsigned long long far = __ballot(have && mag > S.cm1[c]);
                if (lane == 0 && far) atomicAdd(&S.cbeyond[c], __popcll(far));
            }
            bool in = false;
            if (have && !(fabsf(z - S.cbox[c][2]) > S.cbox[c][5] * 0.5f)) {
                const float sx = x - S.cbox[c][0], sy = y - S.cbox[c][1];
                const float lx = sx * S.ccos[c] + sy * (-S.csin[c]);
                const float ly = sx * S.csin[c] + sy * S.ccos[c];
                in = ((double)fabsf(lx) < S.chx[c]) && ((double)fabsf(ly) < S.chy[c]);
            }
            const unsigned long long mask = __ballot(in);
            if (lane == 0 && mask) atomicAdd(&S.ccount[c], __popcll(mask));
        }
    }
    __syncthreads();

    // ---- H: second-stage score; the 3D NMS in score order and its first topk boxes (:994-1045) --------
    if (tid == 0) {
        int nmax = 0, any = 0;
        float dmin = INFINITY, dmax = -INFINITY, emax = -INFINITY, omax = -INFINITY;
        const bool occl = OPT && (prm.occl_w > 0.f || prm.occl_mult);
        // calc_occl_scores as the reference RUNS it (:463): `mags` is (N, 1) and the in-box mask (N,), so the conjunction
        // broadcasts to (N, N) and its sum is (points beyond the nearest corner) x (points outside the box)
        auto occl_of = [&](int c) -> float { return (float)((long long)S.cbeyond[c] * (long long)(m - S.ccount[c])); };
        for (int c = 0; c < NC; ++c) {
            if (S.cvalid[c] >= 1) {   // dists are ranked over the distance-filtered set (:886-893)
                dmin = fminf(dmin, S.cdist[c]);
                dmax = fmaxf(dmax, S.cdist[c]);
            }
            if (S.cvalid[c] == 2) {
                any = 1;
                nmax = max(nmax, S.ccount[c]);
                if (prm.ego_w > 0.f)   // distance of the candidate's centre to the ego vehicle (:1017-1019)
                    emax = fmaxf(emax, sqrtf(S.cbox[c][0] * S.cbox[c][0] + S.cbox[c][1] * S.cbox[c][1] + S.cbox[c][2] * S.cbox[c][2]));
                if (occl) omax = fmaxf(omax, occl_of(c));
            }
        }
        for (int c = 0; c < NC; ++c) {
            if (S.cvalid[c] != 2) continue;
            const float soft = (float)S.ccount[c] / ((float)nmax + 1e-8f);
            const float dr = 1.0f - (S.cdist[c] - dmin) / (dmax - dmin + 1e-8f);
            float s;
            if (!prm.mult) {          // :997
                s = soft * prm.dns_w + S.ciou[c] * prm.iou_w;
                s = s + dr * prm.dst_w;
            } else {                  // MULT, :999
                s = soft * prm.dns_w * S.ciou[c] * prm.iou_w * dr * prm.dst_w;
            }
            if (OPT && prm.occl_w > 0.f)     // :1007-1014
                s = s + prm.occl_w * (1.0f - occl_of(c) / (omax + 1e-6f));
            if (prm.ego_w > 0.f) {    // :1017-1021
                const float ego = sqrtf(S.cbox[c][0] * S.cbox[c][0] + S.cbox[c][1] * S.cbox[c][1] + S.cbox[c][2] * S.cbox[c][2]);
                s = s + prm.ego_w * (ego / emax);
            }
            if (OPT && prm.occl_mult)        // OCCL_MULT, :1022-1027: replaces the score
                s = soft * S.ciou[c] * occl_of(c);
            S.cscore[c] = s;
        }
        // nms_normal_gpu (:1030) in score order = pick the best unsuppressed candidate, drop those whose axis-aligned BEV
        // footprint overlaps it beyond the threshold (iou3d_nms_kernel.cu:327-338), repeat; the first maximum is the order
        // of a stable descending sort.  The shipped topk 1 needs the first pick only.
        int kept = 0;
        const int topk = OPT ? prm.topk : 1;
        for (int k = 0; k < topk; ++k) {
            int best = -1;
            float best_s = -INFINITY;
            for (int c = 0; c < NC; ++c) {
                if (S.cvalid[c] != 2) continue;
                const float sc = S.cscore[c];   // (a NaN score — search_depth on a collapsed frustum — sorts first, as in torch.sort)
                if (best < 0 || (sc != sc && best_s == best_s) || sc > best_s) {
                    best_s = sc;
                    best = c;
                }
            }
            if (best < 0) break;
            out_best[(size_t)f * topk + k] = best;
            out_score[(size_t)f * topk + k] = best_s;
            for (int j = 0; j < 7; ++j) out_box[((size_t)f * topk + k) * 7 + j] = S.cbox[best][j];
            ++kept;
            S.cvalid[best] = 3;   // taken
            if (k + 1 < topk) {
                const float *a = S.cbox[best];
                for (int c = 0; c < NC; ++c) {
                    if (S.cvalid[c] != 2) continue;
                    const float *b = S.cbox[c];
                    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
                    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
                    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
                    const float interS = width * height;
                    if (interS / fmaxf(a[3] * a[4] + b[3] * b[4] - interS, 1e-8f) > prm.nms_normal) S.cvalid[c] = 4;   // suppressed
                }
            }
        }
        for (int c = 0; c < NC; ++c)
            if (S.cvalid[c] > 2) S.cvalid[c] = 2;   // (the dump below reports "scored")
        out_valid[f] = kept;
        for (int k = kept; k < topk; ++k) {
            out_best[(size_t)f * topk + k] = -1;
            out_score[(size_t)f * topk + k] = 0.f;
            for (int j = 0; j < 7; ++j) out_box[((size_t)f * topk + k) * 7 + j] = 0.f;
        }
        (void)any;
    }
    __syncthreads();
    if (dbg_cand)
        for (int i = tid; i < NC * 7; i += kThreads) dbg_cand[(size_t)f * NC * 7 + i] = S.cbox[i / 7][i % 7];
    if (dbg_iou)
        for (int c = tid; c < NC; c += kThreads) dbg_iou[(size_t)f * NC + c] = S.ciou[c];
    if (dbg_count)
        for (int c = tid; c < NC; c += kThreads) dbg_count[(size_t)f * NC + c] = S.ccount[c];
    if (dbg_valid)
        for (int c = tid; c < NC; c += kThreads) dbg_valid[(size_t)f * NC + c] = S.cvalid[c];
}

// The matrices the kernel reads, made on the device from the batch's own (device) tensors: lidar_aug (S,4,4), lidar2image /
// camera2lidar / camera_intrinsics / img_aug (S,6,4,4).  The host version of this (five device-to-host copies with their
// synchronisations, three LAPACK calls and a dozen small tensor operations per launch) was 0.2 ms of the 0.6 ms a one-scene
// launch costs on the host.  3x3 inverses by Gauss-Jordan elimination with partial pivoting in f32 (the reference calls
// torch.inverse, an LU with partial pivoting too; the two agree to rounding).  One thread per (scene, camera) + one per scene.
__device__ __forceinline__ void inv3(const float *A, int stride, float *out) {
    float a[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[r][c] = A[r * stride + c];
            a[r][3 + c] = r == c ? 1.f : 0.f;
        }
#pragma unroll
    for (int col = 0; col < 3; ++col) {
        int piv = col;
        float best = fabsf(a[col][col]);
#pragma unroll
        for (int r = col + 1; r < 3; ++r)
            if (fabsf(a[r][col]) > best) {
                best = fabsf(a[r][col]);
                piv = r;
            }
#pragma unroll
        for (int r = col + 1; r < 3; ++r)
            if (r == piv) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float t = a[col][c];
                    a[col][c] = a[r][c];
                    a[r][c] = t;
                }
            }
        const float d = a[col][col];
#pragma unroll
        for (int c = 0; c < 6; ++c) a[col][c] = a[col][c] / d;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (r == col) continue;
            const float f = a[r][col];
#pragma unroll
            for (int c = 0; c < 6; ++c) a[r][c] = a[r][c] - f * a[col][c];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = a[r][3 + c];
}

__global__ __launch_bounds__(64) void seeker_prepare_kernel(const float *__restrict__ aug, const float *__restrict__ l2i,
                                                            const float *__restrict__ c2l, const float *__restrict__ K,
                                                            const float *__restrict__ ia, int S, float *__restrict__ scene,
                                                            float *__restrict__ cam) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < S) {   // scene: R | inv(R) | t
        const float *A = aug + (size_t)i * 16;
        float *o = scene + (size_t)i * 21;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[r * 3 + c] = A[r * 4 + c];
        inv3(A, 4, o + 9);
#pragma unroll
        for (int r = 0; r < 3; ++r) o[18 + r] = A[r * 4 + 3];
    }
    if (i < S * 6) {
        const float *L = l2i + (size_t)i * 16, *C = c2l + (size_t)i * 16, *Kc = K + (size_t)i * 16;
        float *o = cam + (size_t)i * kCam;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[r * 3 + c] = L[r * 4 + c];
            o[9 + r] = L[r * 4 + 3];
            o[21 + r] = C[r * 4 + 3];
        }
        float kinv[9];
        inv3(Kc, 4, kinv);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)   // camera2lidar rotation x inverse intrinsics (:1530)
                o[12 + r * 3 + c] = (C[r * 4 + 0] * kinv[0 * 3 + c] + C[r * 4 + 1] * kinv[1 * 3 + c]) + C[r * 4 + 2] * kinv[2 * 3 + c];
        if (ia) {
            const float *P = ia + (size_t)i * 16;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) o[24 + r * 3 + c] = P[r * 4 + c];
                o[33 + r] = P[r * 4 + 3];
            }
            inv3(P, 4, o + 36);
        } else {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                o[24 + j] = (j % 4 == 0) ? 1.f : 0.f;
                o[36 + j] = (j % 4 == 0) ? 1.f : 0.f;
            }
            o[33] = o[34] = o[35] = 0.f;
        }
    }
}

// Exchange records of the sharded extraction (findnpropagate_amd/extract.py): scene s of the batch gets one
// (rows_per_scene, 9) f32 record, row 0 = [count, tag, 0 ...], rows 1.. = [box (7), detection score, label] of its
// frustums that yielded a box, in frustum order; every other row is zeroed.  One workgroup per scene; the position of
// a box is the number of valid frustums of its scene before it (ballot + prefix), so the packing needs no host-side
// knowledge of which frustums survived.
struct PackTags { float tag[64]; };

__global__ __launch_bounds__(kThreads) void seeker_pack_kernel(const float *__restrict__ frusts, const int *__restrict__ out_valid,
                                                               const float *__restrict__ out_box, int F, PackTags tags,
                                                               int rows_per_scene, float *__restrict__ rec) {
    __shared__ int wave_cnt[kThreads / 64];
    __shared__ int base_s;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *my = rec + (size_t)s * rows_per_scene * 9;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int f0 = 0; f0 < F; f0 += kThreads) {
        const int f = f0 + tid;
        const bool v = f < F && (int)frusts[(size_t)f * 8] == s && out_valid[f] != 0;
        const unsigned long long m = __ballot(v);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = base_s;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
        if (v && pos + 1 < rows_per_scene) {
            float *row = my + (size_t)(pos + 1) * 9;
#pragma unroll
            for (int j = 0; j < 7; ++j) row[j] = out_box[(size_t)f * 7 + j];
            row[7] = frusts[(size_t)f * 8 + 7];
            row[8] = frusts[(size_t)f * 8 + 6];
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < kThreads / 64; ++w) tot += wave_cnt[w];
            base_s += tot;
        }
        __syncthreads();
    }
    const int count = base_s;
    for (int i = tid; i < 9; i += kThreads) my[i] = i == 0 ? (float)count : i == 1 ? tags.tag[s] : 0.f;
    const int first_free = min(count, rows_per_scene - 1) + 1;
    for (int i = first_free * 9 + tid; i < rows_per_scene * 9; i += kThreads) my[i] = 0.f;
}

}  // namespace

extern "C" int fnp_seeker_pack_records(const float *frustums, const int *out_valid, const float *out_box, int num_frustums,
                                       const float *tags, int num_scenes, int rows_per_scene, float *records,
                                       fnp_stream_t stream) {
    if (num_frustums < 0 || num_scenes <= 0 || num_scenes > 64 || rows_per_scene < 1 || !tags || !records) return FNP_ERR_ARG;
    if (num_frustums > 0 && (!frustums || !out_valid || !out_box)) return FNP_ERR_ARG;
    PackTags t{};
    for (int s = 0; s < num_scenes; ++s) t.tag[s] = tags[s];
    hipLaunchKernelGGL(seeker_pack_kernel, dim3(num_scenes), dim3(kThreads), 0, (hipStream_t)stream, frustums, out_valid,
                       out_box, num_frustums, t, rows_per_scene, records);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// Host side of the frustum enumeration (:561-594): per scene, per camera in image_order, the
// reference runs torchvision.batched_nms on a few dozen CPU boxes and drops low scores.  Same
// arithmetic here (coordinate trick: boxes + label * (max_coordinate + 1) in f32, IoU =
// inter / (a1 + a2 - inter), suppress on IoU > thr, stable score-descending order), as a plain host
// loop: tens of microseconds per scene instead of ~1 ms of Python/torch small-op overhead.
extern "C" int fnp_host_enumerate_frustums(const float *boxes, const int64_t *labels, const float *scores,
                                           const int64_t *batch_idx, const int64_t *cam_idx, int num_dets,
                                           int num_scenes, const int *image_order, int num_cams, float nms_thr,
                                           float score_thr, float *rows, int max_rows) {
    if (num_dets < 0 || num_scenes <= 0 || num_cams <= 0 || !image_order || (num_dets > 0 && (!boxes || !labels || !scores || !batch_idx || !cam_idx)) || !rows)
        return FNP_ERR_ARG;
    int n_rows = 0;
    int *idx = (int *)malloc(sizeof(int) * (size_t)(num_dets > 0 ? num_dets : 1));
    float *ob = (float *)malloc(sizeof(float) * 4 * (size_t)(num_dets > 0 ? num_dets : 1));
    unsigned char *removed = (unsigned char *)malloc((size_t)(num_dets > 0 ? num_dets : 1));
    for (int b = 0; b < num_scenes; ++b) {
        for (int ci = 0; ci < num_cams; ++ci) {
            const int c = image_order[ci];
            int n = 0;
            float mx = -INFINITY;
            for (int i = 0; i < num_dets; ++i)
                if (batch_idx[i] == b && cam_idx[i] == c) {
                    idx[n++] = i;
                    for (int k = 0; k < 4; ++k) mx = fmaxf(mx, boxes[(size_t)i * 4 + k]);
                }
            if (n == 0) continue;
            // stable insertion sort by score, descending
            for (int i = 1; i < n; ++i) {
                const int v = idx[i];
                int j = i - 1;
                while (j >= 0 && scores[idx[j]] < scores[v]) {
                    idx[j + 1] = idx[j];
                    --j;
                }
                idx[j + 1] = v;
            }
            const float shift = mx + 1.0f;
            for (int i = 0; i < n; ++i) {
                const float off = (float)labels[idx[i]] * shift;
                for (int k = 0; k < 4; ++k) ob[i * 4 + k] = boxes[(size_t)idx[i] * 4 + k] + off;
                removed[i] = 0;
            }
            for (int i = 0; i < n; ++i) {
                if (removed[i]) continue;
                const float ai = (ob[i * 4 + 2] - ob[i * 4]) * (ob[i * 4 + 3] - ob[i * 4 + 1]);
                for (int j = i + 1; j < n; ++j) {
                    if (removed[j]) continue;
                    const float aj = (ob[j * 4 + 2] - ob[j * 4]) * (ob[j * 4 + 3] - ob[j * 4 + 1]);
                    const float w = fmaxf(fminf(ob[i * 4 + 2], ob[j * 4 + 2]) - fmaxf(ob[i * 4], ob[j * 4]), 0.f);
                    const float h = fmaxf(fminf(ob[i * 4 + 3], ob[j * 4 + 3]) - fmaxf(ob[i * 4 + 1], ob[j * 4 + 1]), 0.f);
                    const float inter = w * h;
                    if (inter / (ai + aj - inter) > nms_thr) removed[j] = 1;
                }
                const int d = idx[i];
                if (scores[d] < score_thr) continue;   // :594, after the NMS like the reference
                if (n_rows >= max_rows) {
                    free(idx); free(ob); free(removed);
                    return FNP_ERR_WORKSPACE;
                }
                float *r = rows + (size_t)n_rows * 8;
                r[0] = (float)b; r[1] = (float)c;
                r[2] = boxes[(size_t)d * 4]; r[3] = boxes[(size_t)d * 4 + 1]; r[4] = boxes[(size_t)d * 4 + 2]; r[5] = boxes[(size_t)d * 4 + 3];
                r[6] = (float)labels[d]; r[7] = scores[d];
                ++n_rows;
            }
        }
    }
    free(idx); free(ob); free(removed);
    return n_rows;
}

extern "C" int fnp_seeker_prepare_matrices(const float *lidar_aug, const float *lidar2image, const float *camera2lidar,
                                           const float *intrinsics, const float *img_aug, int num_scenes, float *scene_mats,
                                           float *cam_mats, fnp_stream_t stream) {
    if (!lidar_aug || !lidar2image || !camera2lidar || !intrinsics || !scene_mats || !cam_mats || num_scenes <= 0) return FNP_ERR_ARG;
    hipLaunchKernelGGL(seeker_prepare_kernel, dim3(fnp_divup(num_scenes * 6, 64)), dim3(64), 0, (hipStream_t)stream, lidar_aug, lidar2image,
                       camera2lidar, intrinsics, img_aug, num_scenes, scene_mats, cam_mats);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int64_t fnp_boxseeker_workspace_bytes(int num_frustums, int max_points_per_scene) {
    if (num_frustums < 0 || max_points_per_scene < 0) return FNP_ERR_ARG;
    return (int64_t)num_frustums * (max_points_per_scene > 0 ? max_points_per_scene : 1) * 3 * 4 * 2 + 256;
}

extern "C" int fnp_boxseeker(const float *points, const int *scene_offsets, int num_scenes, int max_points_per_scene,
                             const fnp_seeker_params *params, const float *scene_mats, const float *cam_mats,
                             const float *frustums, int num_frustums, const float *base_boxes,
                             const float *base_corners, const float *mags, void *workspace, int64_t workspace_bytes,
                             int *out_valid, float *out_box, float *out_score, int *out_best, int *dbg_npts,
                             float *dbg_frust, float *dbg_cand, float *dbg_iou, int *dbg_count, int *dbg_valid,
                             fnp_stream_t stream) {
    if (!params || num_scenes <= 0 || num_frustums < 0) return FNP_ERR_ARG;
    if (num_frustums == 0) return FNP_OK;
    const int NC = params->num_mags * params->num_rotations * params->num_sizes;
    if (NC <= 0 || NC > kMaxCand || params->num_mags > 16 || params->point_stride < 3 || params->topk < 1 || params->topk > NC) return FNP_ERR_ARG;
    if (params->multicam && !params->count_only && (!params->npts_all || params->num_frustums != num_frustums)) return FNP_ERR_ARG;
    if (params->count_only && !dbg_npts) return FNP_ERR_ARG;
    if (!points || !scene_offsets || !scene_mats || !cam_mats || !frustums || !base_boxes || !base_corners || !mags ||
        !workspace || !out_valid || !out_box || !out_score || !out_best)
        return FNP_ERR_ARG;
    if (fnp_boxseeker_workspace_bytes(num_frustums, max_points_per_scene) > workspace_bytes) return FNP_ERR_WORKSPACE;
    const int stride = max_points_per_scene > 0 ? max_points_per_scene : 1;
    float *ws_uvd = (float *)workspace;
    float *ws_xyz = ws_uvd + (size_t)num_frustums * stride * 3;
    const bool opt = params->topk > 1 || params->search_depth > 0.f || params->occl_w > 0.f || params->occl_mult || params->multicam ||
                     params->count_only || params->rand_noise;
    // small launches (fewer workgroups than two per CU): the instantiation built for a frustum's latency
    const bool lat = num_frustums <= 512;
#define FNP_BS_LAUNCH(O, L)                                                                                                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(boxseeker_kernel<O, L>), dim3(num_frustums), dim3(kThreads), 0, (hipStream_t)stream, points,            \
                       scene_offsets, *params, scene_mats, cam_mats, frustums, base_boxes, base_corners, mags, ws_uvd,         \
                       ws_xyz, stride, out_valid, out_box, out_score, out_best, dbg_npts, dbg_frust, dbg_cand, dbg_iou,        \
                       dbg_count, dbg_valid)
    if (opt && lat) FNP_BS_LAUNCH(true, true);
    else if (opt) FNP_BS_LAUNCH(true, false);
    else if (lat) FNP_BS_LAUNCH(false, true);
    else FNP_BS_LAUNCH(false, false);
#undef FNP_BS_LAUNCH
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
