// Sparse convolution of the ranked 64 -> 64 and 128 -> 128 SubM layers (3x3x3, stages 3 and 4 of VoxelResBackBone8x,
// spconv_backbone.py:212-224) on WIDE tiles (round 4).
//
// What bounds the tile kernels of spconv_tile.hip is not memory but the operand path LDS -> matrix pipe: a wave that owns
// 32 rows x 64 output channels reads (32 + 64) x 64 x 2 B = 12 KB of fragments per kernel offset for 256 cycles of MFMA;
// eight such waves per CU ask the LDS for 0.75 of its 256 B/clk while the matrix pipe is supposed to be full (measured: MFMA
// 0.35 busy, a barrier and an 8 KB slab hand-over per 256 cycles of matrix work).  Here ONE wave per SIMD owns the whole
// 512-register file: 128 rows x 64 output channels (64 -> 64) or 64 rows x 128 output channels (128 -> 128) — 128
// accumulator registers, 24 KB of fragments per 1024 cycles of MFMA (0.375 of the LDS rate at a full pipe), a quarter of the
// barriers and weight-slab bytes per row — and the NEXT tile's image travels memory -> registers during the whole sweep (100
// registers: what two waves per SIMD cannot afford; forms with 8 waves x 64 rows and the image fetched at the tile boundary
// spent a third of their time there, every CU pulling ~250 KB at ~28 GB/s with nothing to compute: measured).
// One 256-thread workgroup per CU, tiles of 512 (256) rows.
//
//   step s = (offset k, 64-wide input-channel half kh), k ascending, kh ascending: the products of every output element are
//   summed in the order of spconv_mfma_kernel (offsets ascending, input channels ascending, one v_mfma_f32_16x16x32 per 32 of
//   them) and the epilogue is its arithmetic: bit-identical output (tests/test_gpu_spconv.py::test_wide_tile_kernel_*).
//   ring     the weights of step s (C x 64 bf16) live in LDS ring slot s & 1.  A step has two halves of 32 MFMAs; the
//            fragments of the second half are read during the first, those of the NEXT step's first half (other slot) during
//            the second, together with the store of step s + 2's weights into slot s & 1 — legal behind the ONE barrier of the
//            step, between its halves, at which every wave has finished reading that slot.
//   image    rows [tile - HALO, tile + TILE + HALO) of the input, the tile's distinct far neighbour rows (tile rulebook's
//            far list), a row of zeros; entries are 16-bit addresses in 16-byte units with the row's rotation in the low bits
//            (tilerb.h: conflict-free for ANY run of 16 consecutive image rows).  The entries themselves stay in memory: a
//            lane needs 16 (8) bytes of them per offset and reads those three offsets ahead.
// An entry the tile record could not hold (ESCAPE: more distinct far rows than overflow rows — arbitrary row orders) is
// fetched through the int32 table, as in spconv_tile.hip: correct for any row order.
#include "tilerb.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct W16 {
    typedef T v8 __attribute__((ext_vector_type(8)));
    typedef T v4 __attribute__((ext_vector_type(4)));
};
// MFMA with the accumulator PINNED: C and D are the same AGPR quad (inline asm, "+a"; the s_nop in front covers what the compiler
// would pad by itself around a builtin: a register it wrote or reloaded just before — an accumulator's zero, a spilled operand —
// read by the MFMA too early came out as NaN in the fourth register of some accumulators).  Through the builtin hipcc puts the 128
// accumulator registers of this kernel in AGPRs all the same but moves them about (v_accvgpr_read / mov / write, ~40 per step
// around MFMAs whose C and D differ): the sweep then runs at half the matrix rate even with its LDS reads removed (measured).
#ifndef FNP_WTILE_ASM_MFMA
#define FNP_WTILE_ASM_MFMA 1
#endif
__device__ __forceinline__ f32x4 wmfma(W16<__bf16>::v8 a, W16<__bf16>::v8 b, f32x4 c) {
#if FNP_WTILE_ASM_MFMA
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 wmfma(W16<_Float16>::v8 a, W16<_Float16>::v8 b, f32x4 c) {
#if FNP_WTILE_ASM_MFMA
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

using tilerb::kEscape;
using tilerb::kK;

template <int C> struct WGeom;
template <> struct WGeom<64> { using G = tilerb::G64W; };
template <> struct WGeom<128> { using G = tilerb::G128W; };

template <int C> struct WCfg {
    using G = typename WGeom<C>::G;
    static constexpr int CH = G::CH;                 // 16-byte chunks per feature row
    static constexpr int KH = C / 64;                // 64-wide halves of the input channels = steps per offset
    static constexpr int NW = 4, NT = NW * 64;       // one wave per SIMD
    static constexpr int NB = C / 16;                // 16-channel output blocks of a wave: all of them
    static constexpr int MB = G::SPLIT;              // 16-row blocks of a wave (8 / 4): NB x MB = 32 accumulator tiles
    static constexpr int WR = MB * 16;               // rows of a wave
    static constexpr int SLOTC = C * 8;              // 16-byte chunks of a ring slot: C output channels x 64 input channels
    static constexpr int NSL = SLOTC / NT;           // ... per thread
    static constexpr int XB = (G::WIN + G::OVF + 1) * G::ROWB;   // the image
    static constexpr int NWL = G::WIN * CH / NT, NOL = G::OVF * CH / NT;
    static constexpr int LDS = 2 * SLOTC * 16 + XB + 64 + G::OVF * 4 + 2 * C * 4;
    static_assert(NB + MB + 1 + MB * 2 * KH <= 30 && G::TILE == NW * WR && NB * MB == 32 && G::WIN * CH % NT == 0 && G::OVF * CH % NT == 0 && SLOTC % NT == 0 && G::OVF <= NT, "shape");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

#ifndef FNP_WTILE_SCHED
#define FNP_WTILE_SCHED 1
#endif
#ifndef FNP_WTILE_SLOTS
#define FNP_WTILE_SLOTS 1
#endif
#ifndef FNP_WTILE_SPREAD
#define FNP_WTILE_SPREAD 1
#endif
// Development-only timing probes (results are wrong; the shipped library has 0): 1 = no MFMA, 2 = no fragment reads,
// 4 = no weight-slab streaming, 8 = no barrier per step, 16 = no sweep at all
#ifndef FNP_WTILE_ABLATE
#define FNP_WTILE_ABLATE 0
#endif

// Development-only phase clocks (FNP_WTILE_STAMP builds): every wave sums the s_memtime ticks of each phase of a tile;
// fnp_debug_wtile_stamps() returns and clears the sums.  [0] sweep (+ the next image's requests), [1] wait for the other waves,
// [2] image -> LDS, [3] epilogue + barrier
#ifdef FNP_WTILE_STAMP
__device__ unsigned long long g_wtile_stamps[8];
#define FNP_WS_NOW(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FNP_WS(ph) do { unsigned long long n_; FNP_WS_NOW(n_); ws_acc[ph] += n_ - ws_prev; ws_prev = n_; } while (0)
#else
#define FNP_WS(ph)
#endif

// Workgroup barrier that orders LDS accesses only (no wait for the vector-memory queue: the next image, the stores, the slabs)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename TAct, int C>
__global__ __launch_bounds__(256, 1) void spconv_wtile_kernel(const TAct *__restrict__ x, int x_bytes, const TAct *__restrict__ w,
                                                               const unsigned char *__restrict__ tile_rb, int rb_bytes,
                                                               const int *__restrict__ nbr, int nbr_stride,
                                                               const int *__restrict__ n_out, int cap, TAct *__restrict__ y,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               const TAct *__restrict__ residual, int relu) {
    using Cfg = WCfg<C>;
    using G = typename Cfg::G;
    using frag8 = typename W16<TAct>::v8;
    using act4 = typename W16<TAct>::v4;
    constexpr int CH = Cfg::CH, KH = Cfg::KH, NT = Cfg::NT, SLOTC = Cfg::SLOTC, NSL = Cfg::NSL, NB = Cfg::NB, MB = Cfg::MB, WR = Cfg::WR;
    constexpr int XB = Cfg::XB, NWL = Cfg::NWL, NOL = Cfg::NOL;
    constexpr int S = kK * KH;                       // steps of a tile's sweep
    constexpr unsigned MASK4 = (unsigned)(CH - 1) << 4;
    constexpr int EW = MB / 2;                       // 32-bit words of entries per lane and offset
    typedef unsigned int ewords __attribute__((ext_vector_type(EW)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *const wl = reinterpret_cast<uint4 *>(smem);                 // [2][SLOTC]
    unsigned char *const img = smem + 2 * SLOTC * 16;
    int *const esc_flags = reinterpret_cast<int *>(img + XB);         // [NW]
    int *const id_lds = esc_flags + 16;                               // [OVF] far-row ids of the tile being staged
    float *const ss_lds = reinterpret_cast<float *>(id_lds + G::OVF);   // [2][C] BatchNorm scale, shift

    const int n = min(*n_out, cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)nbr, 0, kK * nbr_stride * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tile_rb, 0, rb_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, kK * C * C * 2, 0x00020000);

    // contiguous runs of tiles per workgroup, runs of one XCD next to each other (blocks b and b + 8 share an XCD)
    const int ntiles = (n + G::TILE - 1) / G::TILE;
    const int Gd = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = Gd >> 3, rem = Gd & 7;
    const int range = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    const int t_begin = (int)(((long long)ntiles * range) / Gd), t_end = (int)(((long long)ntiles * (range + 1)) / Gd);
    if (t_begin >= t_end) return;   // (whole workgroup, before any barrier)

    // ring slot image: row r (one output channel, 8 chunks = 64 input channels) stores logical chunk c at c ^ ((r >> 1) & 7)
    const int st_pos = (tid / 8) * 8 + ((tid % 8) ^ (((tid / 8) >> 1) & 7));   // (+ j * NT: 32 rows further, same swizzle)
    // ... and where its chunk (tid + j NT) comes from: slab k, row p / 8, bytes [128 kh + 16 (p % 8), + 16)
    unsigned wsrc[NSL];
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const unsigned p = (unsigned)tid + j * NT;
        wsrc[j] = (p / 8) * (unsigned)(C * 2) + (p % 8) * 16u;
    }
    auto slab_req = [&](int s, u32x4 (&dst)[NSL]) {   // step s's weights on their way (steps past the sweep: no traffic)
        const unsigned so = s < S ? (unsigned)(s / KH) * (unsigned)(C * C * 2) + (unsigned)(s % KH) * 128u : 0x80000000u;
#pragma unroll
        for (int j = 0; j < NSL; ++j) dst[j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wsrc[j], so, 0);
    };
    auto slab_put = [&](int ring_slot, const u32x4 (&src)[NSL]) {
#pragma unroll
        for (int j = 0; j < NSL; ++j) *reinterpret_cast<u32x4 *>(&wl[ring_slot * SLOTC + st_pos + j * NT]) = src[j];
    };
    int aoff[2];   // A fragments: output channel 16 nb + l15, chunk 4 ks + q
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[ks] = l15 * 8 + ((ks * 4 + q) ^ ((l15 >> 1) & 7));
    if (tid < CH) reinterpret_cast<uint4 *>(img + G::ZERO * G::ROWB)[tid] = make_uint4(0u, 0u, 0u, 0u);
    for (int c = tid; c < 2 * C; c += NT) ss_lds[c] = scale ? (c < C ? scale[c] : shift[c - C]) : (c < C ? 1.f : 0.f);

    const int rloc = wave * WR + MB * l15;           // this lane's row of block 0 inside the tile (block mb: + mb)
    const int poff = (q & 1) * 32 + (q >> 1) * 16;   // epilogue: this lane's 16 bytes of a 64-byte channel-block pair
    // ---------------------------------------------------------------------------------------- staging of a tile's image
    u32x4 pwin[NWL], povf[NOL], wreg[2][NSL];
    ewords enext[3];      // the staged tile's entries of offsets 0, 1, 2 (this lane's rows)
    int far_id = -1;      // thread s < OVF: id of overflow row s of the tile after the staged one
    unsigned pesc = 0;
    auto rec_off = [&](int t) -> unsigned { return t < t_end ? (unsigned)t * (unsigned)G::REC : 0x80000000u; };
    auto req_far_ids = [&](int t) {
        far_id = __builtin_amdgcn_raw_buffer_load_b32(trsrc, tid < G::OVF ? rec_off(t) + (unsigned)(G::REC_FAR + tid * 4) : 0x80000000u, 0, 0);
    };
    auto entries_load = [&](unsigned ebase, int k) -> ewords {
        const unsigned o = ebase + (unsigned)(k < kK ? k : kK - 1) * (unsigned)(G::TILE * 2);
        if constexpr (EW == 4) return __builtin_amdgcn_raw_buffer_load_b128(trsrc, o, 0, 0);
        else return __builtin_amdgcn_raw_buffer_load_b64(trsrc, o, 0, 0);
    };
    // every load of tile t's image (t >= t_end: nothing is fetched); its far ids are in id_lds (publish_far_ids + a barrier)
    auto stage = [&](int t) {
        const unsigned ro = rec_off(t);
        const unsigned wbase = t < t_end ? (unsigned)max(0, t * G::TILE - G::HALO) * G::ROWB + (unsigned)tid * 16u : 0x80000000u;
#pragma unroll
        for (int j = 0; j < NWL; ++j) pwin[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, wbase + j * (NT * 16), 0, 0);
#pragma unroll
        for (int j = 0; j < NOL; ++j) {
            const unsigned p = (unsigned)tid + j * NT;
            const int key = t < t_end ? id_lds[p / CH] : -1;
            povf[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, key >= 0 ? (unsigned)key * G::ROWB + (p % CH) * 16u : 0x80000000u, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) enext[k] = entries_load(ro + (unsigned)rloc * 2u, k);
        // escape flags: one byte per 32 rows; thread w < NW reads the WR / 32 bytes of wave w (4 or 2) and keeps whether any is set
        if constexpr (WR / 32 == 4) pesc = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(trsrc, tid < Cfg::NW ? ro + (unsigned)(G::REC_ESC + tid * 4) : 0x80000000u, 0, 0);
        else pesc = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(trsrc, tid < Cfg::NW ? ro + (unsigned)(G::REC_ESC + tid * 2) : 0x80000000u, 0, 0);
        req_far_ids(t + 1);
    };
    // the same loads ONE AT A TIME (piece J of NPIECES), for the plain sweep, which issues one per half-step: vector memory returns in
    // order, so behind a burst of all of them the sweep's own loads (weight slabs, entries) wait for the whole image
    constexpr int NPIECES = NWL + NOL + 3 + 1 + 2 * NSL + 1;
    auto stage_piece = [&](auto j_tag, int t) {
        constexpr int J = decltype(j_tag)::value;
        const unsigned ro = rec_off(t);
        if constexpr (J < NWL) {
            const unsigned wbase = t < t_end ? (unsigned)max(0, t * G::TILE - G::HALO) * G::ROWB + (unsigned)tid * 16u : 0x80000000u;
            pwin[J] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, wbase + J * (NT * 16), 0, 0);
        } else if constexpr (J < NWL + NOL) {
            const unsigned p = (unsigned)tid + (J - NWL) * NT;
            const int key = t < t_end ? id_lds[p / CH] : -1;
            povf[J - NWL] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, key >= 0 ? (unsigned)key * G::ROWB + (p % CH) * 16u : 0x80000000u, 0, 0);
        } else if constexpr (J < NWL + NOL + 3) {
            enext[J - NWL - NOL] = entries_load(ro + (unsigned)rloc * 2u, J - NWL - NOL);
        } else if constexpr (J == NWL + NOL + 3) {
            if constexpr (WR / 32 == 4) pesc = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(trsrc, tid < Cfg::NW ? ro + (unsigned)(G::REC_ESC + tid * 4) : 0x80000000u, 0, 0);
            else pesc = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(trsrc, tid < Cfg::NW ? ro + (unsigned)(G::REC_ESC + tid * 2) : 0x80000000u, 0, 0);
        } else if constexpr (J < NWL + NOL + 4 + 2 * NSL) {
            constexpr int w = J - (NWL + NOL + 4), h = w / NSL, j = w % NSL;     // slab h (steps 0 and 1 of the next sweep), chunk j
            wreg[h][j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wsrc[j], (unsigned)(h / KH) * (unsigned)(C * C * 2) + (unsigned)(h % KH) * 128u, 0);
        } else if constexpr (J == NPIECES - 1) {
            req_far_ids(t + 1);
        }
    };
    auto publish_far_ids = [&]() {   // far_id -> LDS; visible behind the next barrier
        if (tid < G::OVF) id_lds[tid] = far_id;
    };
    auto unit_of = [](unsigned slot, unsigned c) -> unsigned {   // 16-byte unit of logical chunk c of the image row at `slot`
        const unsigned code = G::code(slot);
        return (code & ~(unsigned)(CH - 1)) | ((c + code) & (unsigned)(CH - 1));
    };
    auto put_image = [&]() {
#pragma unroll
        for (int j = 0; j < NOL; ++j) {
            const unsigned p = (unsigned)tid + j * NT;
            *reinterpret_cast<u32x4 *>(img + unit_of((unsigned)G::WIN + p / CH, p % CH) * 16u) = povf[j];
        }
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const unsigned p = (unsigned)tid + j * NT;
            *reinterpret_cast<u32x4 *>(img + unit_of(tilerb::win_slot<G>(p / CH), p % CH) * 16u) = pwin[j];
        }
        if (tid < Cfg::NW) esc_flags[tid] = (int)pesc;
    };

    f32x4 acc[NB][MB];
    ewords ecur[3];   // the CURRENT tile's entries of offsets 0, 1, 2 (copied out of enext before the next tile is staged)

    // ---------------------------------------------------------------------------------------- the sweep of one tile
    auto sweep = [&](auto esc_tag, const int tile_base, const int t_next) {
        constexpr bool ESC = decltype(esc_tag)::value;
        // entries of offset k: MB 16-bit image addresses (this lane's rows of the blocks 0..MB-1), kept as one word per block, the
        // address pre-shifted to bytes (a fragment address is an and, an add3 and an and-or); ESC: the byte offset of the row in
        // memory where the entry is an escape.  They come from the tile's record in memory, requested three offsets ahead
        // (eraw[k & 1]) and not touched before the step in front of their first use.
        unsigned e4[2][MB], eoff[2][MB];
        const unsigned q4 = (unsigned)q << 4;
        const unsigned ebase = rec_off(tile_base / G::TILE) + (unsigned)rloc * 2u;
        ewords eraw[2];
        auto entries_cvt = [&](const ewords e, int k, int set) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                unsigned em = (mb & 1) ? e[mb >> 1] >> 16 : e[mb >> 1] & 0xffffu;
                if constexpr (ESC) {
                    eoff[set][mb] = 0x80000000u;
                    if (__ballot(em == kEscape) != 0ull) {
                        if (em == kEscape) {
                            const int id = __builtin_amdgcn_raw_buffer_load_b32(nrsrc, (unsigned)(tile_base + rloc + mb) * 4u,
                                                                                (unsigned)(k < kK ? k : kK - 1) * (unsigned)nbr_stride * 4u, 0);
                            eoff[set][mb] = (unsigned)id * G::ROWB + (unsigned)q * 16u;
                            em = G::code(G::ZERO);
                        }
                    }
                }
                e4[set][mb] = em << 4;
            }
        };
        frag8 fa[2][NB];
        u32x4 fb[2][MB];
        unsigned baddr[2][MB][2 * KH];   // (plain sweep) LDS byte address of the B fragment of (offset set, block, half-step)
        // fragments of half-step (step s = KH k + kh, ks) into register set `set`; A from ring slot s & 1
        auto load_frags = [&](int ring_slot, int eset, int kh, int ks, int set) {
            if (FNP_WTILE_ABLATE & 2) return;
            const uint4 *wk = wl + ring_slot * SLOTC;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint4 tw = wk[aoff[ks] + nb * 16 * 8];
                fa[set][nb] = *reinterpret_cast<const frag8 *>(&tw);
            }
            const unsigned cc = (unsigned)(kh * 8 + ks * 4) << 4;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned a = (e4[eset][mb] & ~MASK4) | ((e4[eset][mb] + q4 + cc) & MASK4);
                fb[set][mb] = *reinterpret_cast<const u32x4 *>(img + a);
                if constexpr (ESC) {
                    if (__ballot(eoff[eset][mb] != 0x80000000u) != 0ull)
                        fb[set][mb] |= __builtin_amdgcn_raw_buffer_load_b128(xrsrc, eoff[eset][mb] + (unsigned)(kh * 8 + ks * 4) * 16u, 0, 0);
                }
            }
        };
        auto mfmas = [&](int set) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const frag8 xv = *reinterpret_cast<const frag8 *>(&fb[set][mb]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (FNP_WTILE_ABLATE & 1) asm volatile("" ::"v"(fa[set][nb]), "v"(xv));
                    else acc[nb][mb] = wmfma(fa[set][nb], xv, acc[nb][mb]);
                }
            }
        };
        auto interleave = [&](auto writes_tag) {
            constexpr int writes = decltype(writes_tag)::value;
            // the 12 LDS reads of the NEXT half-step go out behind the first six of this half-step's 32 MFMAs (two per MFMA), so
            // that they have landed when the half ends (a read issued late is waited for at the barrier with the matrix pipe idle)
            if constexpr (!ESC && FNP_WTILE_SCHED) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU (addresses)
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
                }
                if constexpr (writes > 0) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, writes, 0);  // DS write (weight slab)
                    __builtin_amdgcn_sched_group_barrier(0x020, writes, 0);  // VMEM read (next slab request)
                    __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 26, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        u32x4 wst[NSL];      // the weights of step s + 2 on their way: requested in the second half of step s - 1, stored in that of step s
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (inline-asm MFMAs: the compiler pads no hazards around them — the accumulators' v_accvgpr_write zeros must have landed
        //  before the first MFMA reads them as C; without this the last-written registers came out as NaN on some tiles)
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        entries_cvt(ecur[0], 0, 0);
        entries_cvt(ecur[1], 1, 1);
        eraw[0] = ecur[2];
        if (!(FNP_WTILE_ABLATE & 4)) slab_req(2, wst);
        load_frags(0, 0, 0, 0, 0);
        // one step; PAR = s & 1 and the offset's parity are compile-time, the offset itself is not (13 + 1 trips of two offsets)
        // SLOTS (the plain sweep; FNP_WTILE_SLOTS).  One wave per SIMD has nobody to issue into its gaps, and with the MFMAs as
        // inline asm (pinned accumulators) the compiler's scheduler neither recognises them as matrix instructions nor spreads the
        // other work between them: left alone it issues a half-step's ~60 LDS / VALU instructions first and its 32 MFMAs behind
        // (measured: MFMAs at half their rate; a bare loop of the same asm MFMAs runs at 16.9 cycles each, tools/repro/mfma_rate.hip).
        // So a half-step is written out as 32 slots — one MFMA, then at most a few instructions of side work, fenced by
        // sched_barrier(0) so that the order stands: the 12 fragment reads of the next half-step in the first slots (they land
        // long before the half ends), address arithmetic, the entry conversion and the slab hand-over behind them.
        auto mfma_at = [&](auto set_tag, auto i_tag) {
            constexpr int set = decltype(set_tag)::value, i = decltype(i_tag)::value, mb = i / NB, nb = i % NB;
            const frag8 xv = *reinterpret_cast<const frag8 *>(&fb[set][mb]);
            if (FNP_WTILE_ABLATE & 1) asm volatile("" ::"v"(fa[set][nb]), "v"(xv));
            else acc[nb][mb] = wmfma(fa[set][nb], xv, acc[nb][mb]);
        };
        // side work of a half-step as numbered pieces: [0, NB) A reads, [NB, NB + MB) B reads (address + read), then `extra` pieces
        auto read_piece = [&](auto j_tag, int ring_slot, int eset, int kh, int ks, auto set_tag) {
            constexpr int j = decltype(j_tag)::value, set = decltype(set_tag)::value;
            if (FNP_WTILE_ABLATE & 2) return;
            if constexpr (j < NB) {
                const uint4 tw = (wl + ring_slot * SLOTC)[aoff[ks] + j * 16 * 8];
                fa[set][j] = *reinterpret_cast<const frag8 *>(&tw);
            } else {
                constexpr int mb = j - NB;
                fb[set][mb] = *reinterpret_cast<const u32x4 *>(img + baddr[eset][mb][kh * 2 + ks]);   // (address made offsets ago: one instruction)
            }
        };
        // the B-fragment addresses of an offset (MB blocks x 2 KH half-steps), one per piece, made a step before their first use in
        // slots that have nothing else to do: a read slot is then ONE instruction (a dependent and / add / and-or chain in front of
        // each read cost the single wave ~20 cycles per read)
        auto addr_piece = [&](auto m_tag, int eset) {
            constexpr int M = decltype(m_tag)::value, mb = M / (2 * KH), idx = M % (2 * KH);
            const unsigned cc = (unsigned)((idx / 2) * 8 + (idx % 2) * 4) << 4;
            baddr[eset][mb][idx] = (e4[eset][mb] & ~MASK4) | ((e4[eset][mb] + q4 + cc) & MASK4);
        };
        auto cvt_piece = [&](auto mb_tag, const ewords e, int set) {   // one block of entries_cvt (plain sweep: no escapes)
            constexpr int mb = decltype(mb_tag)::value;
            const unsigned em = (mb & 1) ? e[mb >> 1] >> 16 : e[mb >> 1] & 0xffffu;
            e4[set][mb] = em << 4;
        };
        auto half_slots = [&](auto set_tag, auto second_tag, auto kpar_tag, auto kh_tag, auto hs_tag, const int k) {
            constexpr int HS = decltype(hs_tag)::value;              // index of this half-step in the sweep, or -1 (no image piece)
            constexpr int set = decltype(set_tag)::value;            // register set the MFMAs read; the reads fill set ^ 1
            constexpr bool second = decltype(second_tag)::value;
            constexpr int KP = decltype(kpar_tag)::value, KHv = decltype(kh_tag)::value;
            constexpr int PAR = (KH == 1) ? KP : KHv;
            constexpr bool last_kh = KHv == KH - 1;
            const int s = k * KH + KHv;
            // where the reads of this half go: first half -> (s, ks = 1) of the same ring slot; second -> (s + 1, ks = 0) of the other
            constexpr int r_slot = second ? (PAR ^ 1) : PAR;
            constexpr int r_eset = second ? (last_kh ? (KP ^ 1) : KP) : KP;
            constexpr int r_kh = second ? (last_kh ? 0 : KHv + 1) : KHv;
            constexpr int r_ks = second ? 0 : 1;
            constexpr int NR = NB + MB;                              // 12 fragment reads
            auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
            fence();
#define FNP_SLOT(I)                                                                                                        \
            mfma_at(std::integral_constant<int, set>{}, std::integral_constant<int, (I)>{});                               \
            fence();                                                                                                       \
            if constexpr ((I) < NR) {                                                                                      \
                read_piece(std::integral_constant<int, (I)>{}, r_slot, r_eset, r_kh, r_ks, std::integral_constant<int, set ^ 1>{}); \
                fence();                                                                                                   \
            } else if constexpr (second && last_kh && (I) >= NR && (I) < NR + MB) {                                        \
                cvt_piece(std::integral_constant<int, (I) - NR>{}, eraw[KP], KP);   /* offset k + 2 into the set offset k leaves */ \
                fence();                                                                                                   \
            } else if constexpr (second && (I) == NR + MB) {                                                               \
                if (!(FNP_WTILE_ABLATE & 4)) { slab_put(PAR, wst); slab_req(s + 3, wst); }                                 \
                fence();                                                                                                   \
            } else if constexpr (!second && last_kh && (I) == NR) {                                                        \
                eraw[KP ^ 1] = entries_load(ebase, k + 3);                                                                 \
                fence();                                                                                                   \
            } else if constexpr (!second && KHv == 0 && (I) > NR && (I) <= NR + MB * 2 * KH) {                             \
                addr_piece(std::integral_constant<int, (I) - NR - 1>{}, KP ^ 1);   /* offset k + 1's addresses, from its entries */ \
                fence();                                                                                                   \
            } else if constexpr (FNP_WTILE_SPREAD && (I) == 30 && HS >= 0 && HS < NPIECES) {                                                   \
                stage_piece(std::integral_constant<int, (HS >= 0 && HS < NPIECES) ? HS : 0>{}, t_next);                     \
                fence();                                                                                                   \
            }
            FNP_SLOT(0) FNP_SLOT(1) FNP_SLOT(2) FNP_SLOT(3) FNP_SLOT(4) FNP_SLOT(5) FNP_SLOT(6) FNP_SLOT(7)
            FNP_SLOT(8) FNP_SLOT(9) FNP_SLOT(10) FNP_SLOT(11) FNP_SLOT(12) FNP_SLOT(13) FNP_SLOT(14) FNP_SLOT(15)
            FNP_SLOT(16) FNP_SLOT(17) FNP_SLOT(18) FNP_SLOT(19) FNP_SLOT(20) FNP_SLOT(21) FNP_SLOT(22) FNP_SLOT(23)
            FNP_SLOT(24) FNP_SLOT(25) FNP_SLOT(26) FNP_SLOT(27) FNP_SLOT(28) FNP_SLOT(29) FNP_SLOT(30) FNP_SLOT(31)
#undef FNP_SLOT
        };
        auto step = [&](auto kpar_tag, auto kh_tag, const int k, auto hs_tag) {
            constexpr int KP = decltype(kpar_tag)::value, KHv = decltype(kh_tag)::value, HS0 = decltype(hs_tag)::value;
            constexpr int PAR = (KH == 1) ? KP : KHv;        // s & 1
            const int s = k * KH + KHv;
            if constexpr (!ESC && FNP_WTILE_SLOTS) {
                // NOTE the second half's entry conversion writes the set offset k's B reads used: those reads (of (s, ks = 1)) are
                // all issued in the first half, in front of the barrier
                half_slots(std::integral_constant<int, 0>{}, std::false_type{}, kpar_tag, kh_tag, std::integral_constant<int, HS0>{}, k);
                if (!(FNP_WTILE_ABLATE & 8)) lds_barrier();
                half_slots(std::integral_constant<int, 1>{}, std::true_type{}, kpar_tag, kh_tag, std::integral_constant<int, (HS0 < 0 ? -1 : HS0 + 1)>{}, k);
                return;
            }
            // first half: MFMAs of (s, 0); fragments of (s, 1) from the same slot; the entries of offset k + 3 requested
            load_frags(PAR, KP, KHv, 1, 1);
            if (KHv == KH - 1) eraw[KP ^ 1] = entries_load(ebase, k + 3);   // (eraw[KP ^ 1] held offset k + 1: taken out a step ago)
            mfmas(0);
            interleave(std::integral_constant<int, 0>{});
            if (!(FNP_WTILE_ABLATE & 8)) lds_barrier();   // every wave has read slot PAR for the last time in this step
            // second half: MFMAs of (s, 1); fragments of (s + 1, 0) from the other slot; step s + 2's weights into slot PAR
            if (KHv == KH - 1) entries_cvt(eraw[KP], k + 2, KP);   // (this offset's entries are used up; k + 2's were requested a step ago)
            if (KHv == KH - 1) load_frags(PAR ^ 1, KP ^ 1, 0, 0, 0);
            else load_frags(PAR ^ 1, KP, KHv + 1, 0, 0);
            if (!(FNP_WTILE_ABLATE & 4)) {
                slab_put(PAR, wst);
                slab_req(s + 3, wst);
            }
            mfmas(1);
            interleave(std::integral_constant<int, NSL>{});
        };
        if (!(FNP_WTILE_ABLATE & 16)) {
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using NOHS = std::integral_constant<int, -1>;
            if constexpr (!ESC && FNP_WTILE_SLOTS) {
                // plain sweep: the addresses of offset 0 now (offset 1's are made in step 0), and the first KU offsets written out with
                // one piece of the next image per half-step; KU even, 2 KH KU >= NPIECES
                constexpr int KU = ((NPIECES + 2 * KH - 1) / (2 * KH) + 1) & ~1;
                static_assert(KU <= kK - 1 && (KU & 1) == 0, "image pieces fit the sweep");
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int idx = 0; idx < 2 * KH; ++idx)
                        baddr[0][mb][idx] = (e4[0][mb] & ~MASK4) | ((e4[0][mb] + q4 + ((unsigned)((idx / 2) * 8 + (idx % 2) * 4) << 4)) & MASK4);
                // (the fragments of (0, 0) were read by load_frags above, through e4: same addresses)
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // (see the accumulators' zeros above)
#define FNP_UOFF(K)                                                                                                          \
                if constexpr ((K) < KU) {                                                                                    \
                    step(std::integral_constant<int, (K) & 1>{}, I0{}, (K), std::integral_constant<int, 2 * KH * (K)>{});    \
                    if constexpr (KH == 2) step(std::integral_constant<int, (K) & 1>{}, I1{}, (K), std::integral_constant<int, 2 * KH * (K) + 2>{}); \
                }
                FNP_UOFF(0) FNP_UOFF(1) FNP_UOFF(2) FNP_UOFF(3) FNP_UOFF(4) FNP_UOFF(5) FNP_UOFF(6) FNP_UOFF(7) FNP_UOFF(8) FNP_UOFF(9)
                FNP_UOFF(10) FNP_UOFF(11) FNP_UOFF(12) FNP_UOFF(13) FNP_UOFF(14) FNP_UOFF(15) FNP_UOFF(16) FNP_UOFF(17) FNP_UOFF(18) FNP_UOFF(19)
                FNP_UOFF(20) FNP_UOFF(21) FNP_UOFF(22) FNP_UOFF(23) FNP_UOFF(24) FNP_UOFF(25)
#undef FNP_UOFF
#pragma unroll 1
                for (int k = KU; k + 1 < kK; k += 2) {
                    step(I0{}, I0{}, k, NOHS{});
                    if constexpr (KH == 2) step(I0{}, I1{}, k, NOHS{});
                    step(I1{}, I0{}, k + 1, NOHS{});
                    if constexpr (KH == 2) step(I1{}, I1{}, k + 1, NOHS{});
                }
                step(I0{}, I0{}, kK - 1, NOHS{});
                if constexpr (KH == 2) step(I0{}, I1{}, kK - 1, NOHS{});
            } else {
#pragma unroll 1
                for (int k = 0; k + 1 < kK; k += 2) {
                    step(I0{}, I0{}, k, NOHS{});
                    if constexpr (KH == 2) step(I0{}, I1{}, k, NOHS{});
                    step(I1{}, I0{}, k + 1, NOHS{});
                    if constexpr (KH == 2) step(I1{}, I1{}, k + 1, NOHS{});
                }
                step(I0{}, I0{}, kK - 1, NOHS{});
                if constexpr (KH == 2) step(I0{}, I1{}, kK - 1, NOHS{});
            }
        }
    };

    // ---------------------------------------------------------------------------------------- epilogue of one tile
    // the arithmetic of spconv_mfma_kernel (scale / shift, residual, ReLU, one rounding), 16 bytes per lane: the 8-byte pieces of
    // two channel blocks are exchanged between the lane rows q, q ^ 1 of a site (v_permlane16_swap)
    uint4 rv[MB][NB / 2];   // residual rows: all requested right behind the sweep (one latency, not one per block)
    auto req_residual = [&](const int tile_base) {
        const int row_end = min(n, tile_base + G::TILE);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int kp = 0; kp < NB / 2; ++kp) {
                const int r = tile_base + rloc + mb;
                rv[mb][kp] = make_uint4(0u, 0u, 0u, 0u);
                if (residual && r < row_end)
                    rv[mb][kp] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(residual) + (size_t)r * (C * 2) + kp * 64 + poff);
            }
    };
    auto epilogue = [&](const int tile_base) {
        const int row_end = min(n, tile_base + G::TILE);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int r = tile_base + rloc + mb;
            const bool live = r < row_end;
#pragma unroll
            for (int kp = 0; kp < NB / 2; ++kp) {
                uint2 ra = make_uint2(0u, 0u), rbb = make_uint2(0u, 0u);
                if (residual) {
                    auto t0 = __builtin_amdgcn_permlane16_swap(rv[mb][kp].x, rv[mb][kp].z, false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(rv[mb][kp].y, rv[mb][kp].w, false, false);
                    ra = make_uint2(t0[0], t1[0]);    // block 2 kp,     channels q*4 .. q*4+3
                    rbb = make_uint2(t0[1], t1[1]);   // block 2 kp + 1
                }
                uint2 o[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nb = 2 * kp + h, c0 = nb * 16 + q * 4;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[nb][mb][j];
                    if (scale) {
                        const float4 s4 = *reinterpret_cast<const float4 *>(ss_lds + c0);
                        const float4 h4 = *reinterpret_cast<const float4 *>(ss_lds + C + c0);
                        v[0] = v[0] * s4.x + h4.x; v[1] = v[1] * s4.y + h4.y; v[2] = v[2] * s4.z + h4.z; v[3] = v[3] * s4.w + h4.w;
                    }
                    if (residual) {
                        const uint2 rr = h ? rbb : ra;
                        const act4 tr = *reinterpret_cast<const act4 *>(&rr);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] + (float)tr[j];
                    }
                    if (relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
                    }
                    const act4 ob = {(TAct)v[0], (TAct)v[1], (TAct)v[2], (TAct)v[3]};
                    o[h] = *reinterpret_cast<const uint2 *>(&ob);
                }
                auto t0 = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);
                if (live)
                    *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(y) + (size_t)r * (C * 2) + kp * 64 + poff) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            }
        }
    };

    // ---------------------------------------------------------------------------------------- tiles
#ifdef FNP_WTILE_STAMP
    unsigned long long ws_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ws_prev;
    FNP_WS_NOW(ws_prev);
#endif
    // prologue: the first image, and the requests of the second
    req_far_ids(t_begin);
    publish_far_ids();
    __syncthreads();   // far ids, zero row, scale / shift
    stage(t_begin);    // (also requests the far ids of tile t_begin + 1)
    slab_req(0, wreg[0]);
    slab_req(1, wreg[1]);
    put_image();
    slab_put(0, wreg[0]);
    slab_put(1, wreg[1]);
#pragma unroll
    for (int k = 0; k < 3; ++k) ecur[k] = enext[k];
    publish_far_ids();
    lds_barrier();
    for (int t = t_begin; t < t_end; ++t) {
        const int tile_base = t * G::TILE;
        // the NEXT tile's image travels memory -> registers during this sweep (its far ids were published behind the last barrier): the
        // plain sweep requests it piece by piece, one load per half-step; a sweep with escapes (rare) in one burst in front
#ifdef FNP_WTILE_NOESC   // (development: register budget of the plain sweep alone)
        sweep(std::false_type{}, tile_base, t + 1);
#else
        if (esc_flags[wave] || !FNP_WTILE_SLOTS) {
            stage(t + 1);
            slab_req(0, wreg[0]);   // (the next sweep's first two slabs)
            slab_req(1, wreg[1]);
            sweep(std::true_type{}, tile_base, t + 1);
        } else {
            if (!FNP_WTILE_SPREAD) {
                stage(t + 1);
                slab_req(0, wreg[0]);
                slab_req(1, wreg[1]);
            }
            sweep(std::false_type{}, tile_base, t + 1);
        }
#endif
        FNP_WS(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the inline-asm MFMAs' results are read below: no hazard padding by the compiler)
        req_residual(tile_base);
        publish_far_ids();                 // (of tile t + 2, requested by stage(t + 1))
        lds_barrier();                     // every wave has left the image and the ring
        FNP_WS(1);
        // image -> LDS BEFORE the epilogue issues its stores: the wait for the image's loads (issued before the sweep loop) cannot be
        // counted across that loop, so it is vmcnt(0) — behind the stores it waited until all of them had reached memory (measured:
        // 16 k cycles per tile); here it waits for the residual rows just requested, which the epilogue needs next anyway
        if (t + 1 < t_end) {               // (uniform)
            put_image();
            slab_put(0, wreg[0]);
            slab_put(1, wreg[1]);
#pragma unroll
            for (int k = 0; k < 3; ++k) ecur[k] = enext[k];
        }
        FNP_WS(2);
        epilogue(tile_base);               // (its stores drain during the next sweep)
        lds_barrier();                     // the image and the first two slabs are in LDS
        FNP_WS(3);
    }
#ifdef FNP_WTILE_STAMP
    if (lane == 0)
        for (int ph = 0; ph < 8; ++ph) atomicAdd(&g_wtile_stamps[ph], ws_acc[ph]);
#endif
}

template <typename TAct, int C>
int launch_wtile(const void *x, long long x_bytes, const void *w, const void *tile_rb, long long rb_bytes, const int *nbr, int nbr_stride,
                 const int *n_out, int cap, void *y, const float *scale, const float *shift, const void *residual, int relu, hipStream_t s) {
    using Cfg = WCfg<C>;
    auto kern = spconv_wtile_kernel<TAct, C>;
    static bool raised = false;   // (idempotent; a race only repeats the call)
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS) != hipSuccess) return FNP_ERR_HIP;
        raised = true;
    }
    const int tiles = fnp_divup(cap, Cfg::G::TILE);
    int grid = tiles < 256 ? tiles : 256;
#ifdef FNP_WTILE_GRID_ENV   // (development probe: workgroups per launch from the environment)
    if (const char *e = getenv("FNP_WTILE_GRID")) grid = atoi(e) < tiles ? atoi(e) : tiles;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS, s, (const TAct *)x, (int)x_bytes, (const TAct *)w, (const unsigned char *)tile_rb, (int)rb_bytes,
                       nbr, nbr_stride, n_out, cap, (TAct *)y, scale, shift, (const TAct *)residual, relu);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

}  // namespace

extern "C" long long fnp_wtile_rulebook_bytes(int cap_out, int channels);

#ifdef FNP_WTILE_STAMP
extern "C" int fnp_debug_wtile_stamps(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wtile_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return FNP_ERR_HIP;
    unsigned long long zero[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wtile_stamps), zero, sizeof(zero)) != hipSuccess) return FNP_ERR_HIP;
    return FNP_OK;
}
#endif

extern "C" int fnp_spconv_forward_wtiled(const void *feat_in, int dtype, int n_in_rows, const void *weight, const void *tile_rb, const int *nbr,
                                         int nbr_stride, const int *n_out, int cap_out, void *feat_out, const float *scale, const float *shift,
                                         const void *residual, int relu, int Cin, int Cout, fnp_stream_t stream) {
    if (!feat_in || !weight || !tile_rb || !nbr || !n_out || !feat_out || cap_out <= 0 || nbr_stride < cap_out || n_in_rows <= 0) return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr) || Cin != Cout || (Cin != 64 && Cin != 128)) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2, rbb = fnp_wtile_rulebook_bytes(cap_out, Cin);
    // 32-bit buffer offsets into the features, the int32 table (escape fetches) and the tile rulebook
    if (xb >= 0x7fffffffll || (long long)kK * nbr_stride * 4 >= 0x7fffffffll || rbb >= 0x7fffffffll) return FNP_ERR_ARG;
    if (((uintptr_t)tile_rb & 15) || ((uintptr_t)feat_in & 15) || ((uintptr_t)feat_out & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)residual & 15)) return FNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 64) {
        if (dtype == FNP_BF16) return launch_wtile<__bf16, 64>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
        if (dtype == FNP_F16) return launch_wtile<_Float16, 64>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
    } else {
        if (dtype == FNP_BF16) return launch_wtile<__bf16, 128>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
        if (dtype == FNP_F16) return launch_wtile<_Float16, 128>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
    }
    return FNP_ERR_ARG;
}
