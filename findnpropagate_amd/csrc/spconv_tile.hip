// Sparse convolution for the ranked 32 -> 32 SubM layers (the four 3x3x3 layers of stage 2) on a TILE RULEBOOK: the
// rulebook of 256-row tiles restated once, relative to the tile, as 16-bit LDS addresses — half the bytes of the (27, cap)
// int32 table it is built from — so that the convolution holds a tile's whole neighbourhood in LDS and its offset sweep
// issues no gather at all.
//
// spconv_mfma_kernel gathers every (site, offset) fragment from L2 and is bound by the per-CU rate of that gather path
// (DESIGN.md section 5); its window variant still issues a gather instruction per fragment next to the LDS read.  Rows of a
// ranked tensor sit next to their neighbours: ~95 % of a tile's neighbour rows lie within +-32 rows of the tile, and the
// rest are ~40 DISTINCT rows per tile, each referenced several times.  fnp_tile_rulebook_build (one pass per rulebook;
// the stage-2 rulebook serves four convolutions per forward) therefore writes, per tile:
//   codes   27 x 256 16-bit entries: the LDS byte address, inside the tile image below, of the neighbour row — a window row
//           (rows [tile - 32, tile + 288) of the input), an overflow row (a far neighbour, deduplicated through a small
//           open-addressing table keyed by row id: the slot number is the overflow row), the row of zeros (no neighbour),
//           or ESCAPE (the tile has more distinct far rows than overflow rows — not seen on rank-ordered inputs, common on
//           arbitrary ones: the convolution then fetches that fragment through the int32 table; correct for any order);
//   far     the 256 row ids of the overflow rows (-1: unused);
//   escape  per 32-row group, whether any of its entries is ESCAPE.
// The convolution (spconv_tile32_kernel) runs one 1024-thread workgroup per CU over a contiguous run of tiles.  Eight
// PRODUCER waves only move data: window, codes, overflow rows and flags of the tile after next go from memory to
// registers, and on to the LDS image the consumers have just left.  Eight CONSUMER waves sweep the 27 offsets of the
// current tile from LDS only — 32-bit entry pair, 16-byte fragment, MFMA — and run the epilogue.  Two images alternate;
// they are handed over through counters in LDS, not workgroup barriers (see below).
// Products, their order (offsets ascending, one 32-wide MFMA step per offset) and the epilogue arithmetic are those of
// spconv_mfma_kernel: the output is bit-identical (tests/test_gpu_spconv.py::test_tile_kernel_equals_gather_kernel).
#include "tilerb.h"
#include <type_traits>

#ifndef FNP_NT_STORE
#define FNP_NT_STORE 0   // (development: output rows stored with the non-temporal hint)
#endif
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct V16 {
    typedef T v8 __attribute__((ext_vector_type(8)));
    typedef T v4 __attribute__((ext_vector_type(4)));
};
__device__ __forceinline__ f32x4 tmfma(V16<__bf16>::v8 a, V16<__bf16>::v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 tmfma(V16<_Float16>::v8 a, V16<_Float16>::v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

using tilerb::G32;
using tilerb::G64;
using tilerb::kEscape;
constexpr int kK = tilerb::kK, kC = 32, kCH = 4, kRowB = G32::ROWB;   // offsets; 32-channel kernel: channels, 16-byte chunks and bytes per row
constexpr int kTile = G32::TILE, kHalo = G32::HALO, kWin = G32::WIN, kOvf = G32::OVF, kZeroRow = G32::ZERO;
constexpr int kSlab = kC * kCH;                         // chunks per weight slab
constexpr int kWBytes = kK * kSlab * 16;                // 55,296: all 27 slabs resident
constexpr int kXBytes = (kWin + kOvf + 1) * kRowB;      // window + overflow + zero row
constexpr int kRbBytes = kK * kTile * 2;                // the entries of a tile
constexpr int kImgBytes = kXBytes + kRbBytes;           // one tile image
constexpr int kLds = kWBytes + 2 * kImgBytes + 64 + 64; // + escape flags + hand-over counters
constexpr int kRecFar = G32::REC_FAR, kRecEsc = G32::REC_ESC, kRecBytes = G32::REC;
static_assert(kLds <= 160 * 1024, "LDS budget");
static_assert(kTile == 256, "shape");
__host__ __device__ constexpr unsigned row_code(unsigned rs) { return G32::code(rs); }
__host__ __device__ constexpr unsigned win_slot(unsigned d) { return tilerb::win_slot<G32>(d); }

// ------------------------------------------------------------------------------------------ tile rulebook
// One 256-thread workgroup per tile; a thread restates 4 consecutive rows of the offsets kq, kq + NG, ... (16-byte loads
// of the int32 table, 8-byte stores of the entries).  Far rows go through an LDS open-addressing table (compare-and-
// swap, linear probing): which slot a row gets depends on the order the threads arrive in, the BYTES the convolution
// reads through it do not.
template <typename G>
__global__ __launch_bounds__(256) void tile_rulebook_kernel(const int *__restrict__ nbr, int nbr_stride, const int *__restrict__ n_out, int cap,
                                                            unsigned char *__restrict__ out) {
    constexpr int RL = G::TILE / 4, NG = 256 / RL, NJ = (kK + NG - 1) / NG;   // lanes per offset, offsets per pass, passes
    __shared__ int table[G::OVF];
    __shared__ int esc[16];
    const int n = min(*n_out, cap);
    const int t = blockIdx.x, tile_base = t * G::TILE, tid = threadIdx.x;
    if (tile_base >= n) return;
    if (tid < G::OVF) table[tid] = -1;
    if (tid < 16) esc[tid] = 0;
    __syncthreads();
    const int wlo = max(0, tile_base - G::HALO), kq = tid / RL, r0 = (tid % RL) * 4, row0 = tile_base + r0;
    int4 id[NJ];
    const bool wide = !(nbr_stride & 3) && !((uintptr_t)nbr & 15);   // (uniform) table rows are 16-byte aligned
#pragma unroll
    for (int j = 0; j < NJ; ++j) {   // (rows past n are masked below)
        const int k = kq + NG * j;
        id[j] = make_int4(-1, -1, -1, -1);
        if (k < kK && row0 < n) {
            const int *p = nbr + (size_t)k * nbr_stride + row0;
            if (wide) id[j] = *reinterpret_cast<const int4 *>(p);
            else id[j] = make_int4(p[0], row0 + 1 < n ? p[1] : -1, row0 + 2 < n ? p[2] : -1, row0 + 3 < n ? p[3] : -1);
        }
    }
    unsigned char *rec = out + (size_t)t * G::REC;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = kq + NG * j;
        if (k >= kK) break;
        const int ids[4] = {id[j].x, id[j].y, id[j].z, id[j].w};
        unsigned code[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            code[u] = tilerb::entry_of<G>(row0 + u < n ? ids[u] : -1, wlo, table);
            if (code[u] == kEscape) esc[(r0 + u) >> 5] = 1;
        }
        *reinterpret_cast<uint2 *>(rec + ((size_t)k * G::TILE + r0) * 2) = make_uint2(code[0] | (code[1] << 16), code[2] | (code[3] << 16));
    }
    __syncthreads();
    if (tid < G::OVF) reinterpret_cast<int *>(rec + G::REC_FAR)[tid] = table[tid];
    if (tid < 16) rec[G::REC_ESC + tid] = (unsigned char)esc[tid];
}

// Development-only timing probes (results are wrong; the shipped library has 0): 1 = escape entries are taken as absent,
// 2 = consumers skip the offset sweep, 4 = producers write the first two tiles only, 16 / 32 = consumers skip weight / fragment reads,
// 64 = no MFMA, 512 = no output stores; 64-channel kernel: 128 = the weight slabs are not streamed, 256 = no barrier per offset
#ifndef FNP_TILE_ABLATE
#define FNP_TILE_ABLATE 0
#endif
#ifndef FNP_TILE_SCHED
#define FNP_TILE_SCHED 1
#endif
#ifndef FNP_TILE64_SCHED
#define FNP_TILE64_SCHED 1
#endif
#ifndef FNP_TILE64_RESK
#define FNP_TILE64_RESK 6   // residual rows are requested this many offsets before the sweep ends
#endif
#ifndef FNP_CONV_PRIO
#define FNP_CONV_PRIO 0
#endif
#ifndef FNP_TILE32_PRIO
#define FNP_TILE32_PRIO 0
#endif
// wave priority during the 64-channel kernel's offset sweep (round 5).  The two workgroups of a CU share its SIMDs wave by wave;
// when one of them is between two sweeps — image and slab stores, residual loads, the epilogue's arithmetic and row stores — its
// instructions compete at equal priority with the other one's matrix and LDS-read stream, which is what bounds the launch.  With
// the sweep at priority 2 (the rest at 0): 0.607 -> 0.577 ms per launch at 128 scenes (-4.9 %; priority 1 / 3: 0.579 / 0.577;
// tools/ab_tiled.py, interleaved, bit-identical).  The same switch on the 32-channel kernel's consumer waves (against its producer
// waves) and on the gather kernels' sweeps measured within +-0.5 %: not taken there.
#ifndef FNP_TILE64_PRIO
#define FNP_TILE64_PRIO 2
#endif
#ifndef FNP_TILE64_WDEPTH
#define FNP_TILE64_WDEPTH 2   // weight slabs in flight in registers (64-channel kernel)
#endif
#ifndef FNP_TILE64_STORE_AT
#define FNP_TILE64_STORE_AT 6
#endif
#ifndef FNP_TILE64_SPREAD
#define FNP_TILE64_SPREAD 2   // (1: one piece per offset from the sweep's second offset on — round 4; 2: two per offset behind the last slab requests — round 5)
#endif
#ifndef FNP_TILE32_NPW
#define FNP_TILE32_NPW 8   // producer waves of the 32-channel kernel
#endif
#ifndef FNP_TILE_PSLEEP
#define FNP_TILE_PSLEEP 8
#endif
#ifndef FNP_TILE_DW
#define FNP_TILE_DW 1
#endif
#ifndef FNP_TILE_DX
#define FNP_TILE_DX 2
#endif

// Development-only phase clocks (FNP_TILE_STAMP builds): every wave sums the s_memtime ticks it spends in each phase
// of its role; fnp_debug_tile_stamps() returns and clears the sums.  [role 0 = consumer, 1 = producer][phase]
#ifdef FNP_TILE_STAMP
__device__ unsigned long long g_tile_stamps[4][8];   // [role][phase] sums, [2 + role][phase] maxima over waves
#define FNP_STAMP_NOW(v)                                                       \
    do {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                     \
    } while (0)
#define FNP_STAMP_DECL                                        \
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev; \
    FNP_STAMP_NOW(st_prev)
#define FNP_STAMP(ph)                                                          \
    do {                                                                       \
        unsigned long long st_now;                                             \
        FNP_STAMP_NOW(st_now);                                                 \
        st_acc[ph] += st_now - st_prev;                                        \
        st_prev = st_now;                                                      \
    } while (0)
#define FNP_STAMP_FLUSH(role)                                                  \
    do {                                                                       \
        if (lane == 0)                                                         \
            for (int ph = 0; ph < 8; ++ph) {                                   \
                atomicAdd(&g_tile_stamps[role][ph], st_acc[ph]);               \
                atomicMax(&g_tile_stamps[2 + role][ph], st_acc[ph]);           \
            }                                                                  \
    } while (0)
#else
#define FNP_STAMP_DECL
#define FNP_STAMP(ph)
#define FNP_STAMP_FLUSH(role)
#endif

// ------------------------------------------------------------------------------------------ convolution
__device__ unsigned g_tile_aborts;   // hand-over waits that timed out, since the library was loaded (never, unless the protocol is broken)
__device__ int g_tile_hold;          // test hook (fnp_debug_tile_hold): producers never publish an image, so every consumer wait times out

// Geometry: NCW = 16 / MB consumer waves x (16 MB) rows, MB = FNP_TILE32_MB = 2 as shipped: consumer wave w owns tile rows
// [32 w, 32 w + 32); its MFMA column l15 of block mb is row 32 w + MB l15 + mb, so the MB entries of a lane sit in one 32-bit (MB = 4:
// 64-bit) word of the natural [offset][row] table.  Neighbours of such a column set are rows of ONE residue mod MB, so the window
// image keeps the residues in separate parts (tilerb.h SPLIT): what a fragment read touches is then 16 consecutive 64-byte rows, as
// in a dense tile.
// MB = 4 (round 5; four consumer waves x 64 rows, 768 threads, 166 registers): a weight fragment read from LDS serves four row
// blocks instead of two — 6.5 LDS reads per 8 MFMAs instead of 4.5 per 4, -28 % of the array cycles per flop, the remedy if the LDS
// array (SQ_LDS_IDX_ACTIVE = 0.87 of the busy CU cycles, r04 PMC) were what bounds the kernel.  Bit-identical and SLOWER: 0.304
// against 0.289 ms per launch at 128 scenes, 0.159 against 0.152 at 64 (tools/ab_tiled.py, interleaved in one process; four
// producer waves instead of eight: 0.304 / 0.164) — ONE consumer wave per SIMD has nobody to issue into its waits, which costs more
// than the array cycles saved.  Kept as a build switch with the numbers.
// X3 (fnp_spconv_forward_tiled_split, the bf16x3 engine's main product): the epilogue keeps f32 — f32 residual in, a 16-bit addend
// (the two cross terms) joined before the ReLU, f32 rows out AND their (hi, lo) 16-bit split, the next layer's operands.  The
// kernel's own y / residual are unused then.
struct X3Args {
    const float *residual;
    const void *addend;
    float *y;
    void *hi, *lo;
};
template <typename TAct>
__device__ __forceinline__ void x3_store(const X3Args &a, size_t elem, float (&v)[4], int relu) {
    typedef TAct act4 __attribute__((ext_vector_type(4)));
    if (a.residual) {
        const float4 r4 = *reinterpret_cast<const float4 *>(a.residual + elem);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
    }
    if (a.addend) {
        const act4 t4 = *reinterpret_cast<const act4 *>(reinterpret_cast<const TAct *>(a.addend) + elem);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (float)t4[j];
    }
    if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.f ? 0.f : v[j];
    }
    act4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (TAct)v[j];
        l[j] = (TAct)(v[j] - (float)h[j]);
    }
    if (a.y) *reinterpret_cast<float4 *>(a.y + elem) = make_float4(v[0], v[1], v[2], v[3]);   // (null: only the split is wanted)
    *reinterpret_cast<act4 *>(reinterpret_cast<TAct *>(a.hi) + elem) = h;
    *reinterpret_cast<act4 *>(reinterpret_cast<TAct *>(a.lo) + elem) = l;
}

template <typename TAct, bool X3 = false>
__global__ __launch_bounds__((16 / FNP_TILE32_MB + FNP_TILE32_NPW) * 64, (16 / FNP_TILE32_MB + FNP_TILE32_NPW) / 4) void spconv_tile32_kernel(const TAct *__restrict__ x, int x_bytes, const TAct *__restrict__ w,
                                                                 const unsigned char *__restrict__ tile_rb, int rb_bytes,
                                                                 const int *__restrict__ nbr, int nbr_stride,
                                                                 const int *__restrict__ n_out, int cap, TAct *__restrict__ y,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 const TAct *__restrict__ residual, int relu, X3Args x3) {
    using frag8 = typename V16<TAct>::v8;
    using act4 = typename V16<TAct>::v4;
    constexpr int MB = FNP_TILE32_MB, NCW = 16 / MB, NPW = FNP_TILE32_NPW, NT = (NCW + NPW) * 64, PT = NPW * 64, NB = kC / 16;
    static_assert(MB == 2 || MB == 4, "consumer tile");
    constexpr int NWL = (kWin * kCH + PT - 1) / PT;          // window chunks per producer thread
    constexpr int NCL = (kRbBytes / 16 + PT - 1) / PT;       // entry-table chunks per producer thread
    constexpr int NGL = kOvf / NPW / 16;                     // overflow-row loads per producer thread (4 lanes per row)
    constexpr int OPW = kOvf / NPW;                          // overflow rows a producer wave fetches
    static_assert(NB == 2 && kTile == NCW * MB * 16 && OPW == NGL * 16 && OPW <= 64, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *wl = reinterpret_cast<uint4 *>(smem);
    unsigned char *const img0 = smem + kWBytes;
    int *const esc_flags = reinterpret_cast<int *>(smem + kWBytes + 2 * kImgBytes);   // [image][consumer wave]
    // Hand-over counters (no workgroup barrier after the prologue: a barrier per tile made every wave wait for the slowest
    // one of either role, and the fill and drain of each wave's pipeline fell on the same moment for all of them).  Each
    // counts WAVES that completed an event and only grows; i = tile number inside the workgroup's run, image i & 1:
    //   READY[i & 1]  producer waves that finished writing the image     consumers of tile i wait for NPW (i / 2 + 1)
    //   FREED[i & 1]  consumer waves that finished reading the image     producers of tile i wait for NCW (i / 2)
    // A wait that outlasts kSpinLimit polls (tens of milliseconds; a hand-over takes microseconds) raises ABORT, which ends
    // every wave of the workgroup — wrong output instead of a hung GPU — and counts in g_tile_aborts, which the host layer
    // reads with its per-forward counts (fnp_spconv_tiled_aborts_copy) and turns into an error.
    int *const cnt = esc_flags + 16;
    enum { READY = 0, FREED = 2, ABORT = 8 };
    constexpr int kSpinLimit = 1 << 20;
    auto wait_for = [&](int which, int need) -> bool {
        int spins = 0;
        while (__atomic_load_n(&cnt[which], __ATOMIC_RELAXED) < need) {
            if (++spins > kSpinLimit || __atomic_load_n(&cnt[ABORT], __ATOMIC_RELAXED)) {
                if (spins > kSpinLimit && (threadIdx.x & 63) == 0) atomicAdd(&g_tile_aborts, 1u);
                __atomic_store_n(&cnt[ABORT], 1, __ATOMIC_RELAXED);
                return false;
            }
            // (a poll is an LDS read in the consumers' queue: the producers, a tile ahead, poll slowly)
            if (which >= FREED) __builtin_amdgcn_s_sleep(FNP_TILE_PSLEEP);
            else __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return true;
    };
    auto signal = [&](int which) {   // after this wave's LDS accesses of the event
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&cnt[which], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

#if FNP_CONV_PRIO
    __builtin_amdgcn_s_setprio(FNP_CONV_PRIO);   // (probe: convolution waves ahead of the index kernels that share the CUs in the replayed step)
#endif
    const int n = min(*n_out, cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)nbr, 0, kK * nbr_stride * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tile_rb, 0, rb_bytes, 0x00020000);

    // contiguous runs of tiles per workgroup, runs of one XCD next to each other (blocks b and b + 8 share an XCD)
    const int ntiles = (n + kTile - 1) / kTile;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = G >> 3, rem = G & 7;
    const int range = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    const int t_begin = (int)(((long long)ntiles * range) / G), t_end = (int)(((long long)ntiles * (range + 1)) / G);
    if (t_begin >= t_end) return;   // (whole workgroup, before any barrier)
    const int nt = t_end - t_begin;

    // Swizzles: a ds_read_b128 is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
    // same + 32 — i.e. MFMA rows 0-3 and 12-15 of one 8-channel chunk with rows 4-11 of the next.  Weight image: row r
    // (one output channel, 4 chunks) stores logical chunk c at c ^ ((r >> 1) & 3); feature rows: row_code().  Both are
    // conflict-free for 16 consecutive rows starting at a multiple of 16.
    auto wpos = [](int row, int chunk) { return row * kCH + (chunk ^ ((row >> 1) & 3)); };
    for (int p = tid; p < kK * kSlab; p += NT) {
        const int kk = p / kSlab, r = p % kSlab;
        wl[kk * kSlab + wpos(r / kCH, r % kCH)] = reinterpret_cast<const uint4 *>(w)[p];
    }
    if (tid < 2 * kCH) reinterpret_cast<uint4 *>(img0 + (tid / kCH) * kImgBytes + kZeroRow * kRowB)[tid % kCH] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 16) cnt[tid] = 0;

    if (wave >= NCW) {
        // ------------------------------------------------------------------ producer waves: memory -> registers -> image
        const int ptid = tid - NCW * 64, pw = wave - NCW;
        u32x4 pwin[NWL], pcod[NCL], g[NGL];
        int far_id = -1;       // lanes 0-31: row id of overflow row 32 pw + lane of the tile after the one in the registers
        unsigned pesc = 0;     // producer thread c < 8: escape flag of consumer wave c
        // LDS offsets (inside an image) of this thread's window chunks and overflow-row pieces
        unsigned win_dst[NWL], ovf_dst[NGL];
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const unsigned p = (unsigned)ptid + j * PT;
            win_dst[j] = row_code(win_slot(p / kCH)) ^ ((p % kCH) << 4);
        }
#pragma unroll
        for (int i = 0; i < NGL; ++i) ovf_dst[i] = row_code((unsigned)(kWin + pw * OPW + i * 16 + (lane >> 2))) ^ ((unsigned)(lane & 3) << 4);
        auto rec_off = [&](int t) -> unsigned { return t < t_end ? (unsigned)t * (unsigned)kRecBytes : 0x80000000u; };
        auto req_far_ids = [&](int t) {
            far_id = __builtin_amdgcn_raw_buffer_load_b32(trsrc, lane < OPW ? rec_off(t) + (unsigned)(kRecFar + (pw * OPW + lane) * 4) : 0x80000000u, 0, 0);
        };
        auto req_tile = [&](int t) {   // everything of tile t but its far-row ids, which must be in far_id already
            const unsigned ro = rec_off(t);
            const unsigned wbase = t < t_end ? (unsigned)max(0, t * kTile - kHalo) * kRowB + (unsigned)ptid * 16u : 0x80000000u;
#pragma unroll
            for (int j = 0; j < NWL; ++j)
                pwin[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (ptid + j * PT < kWin * kCH) ? wbase + j * (PT * 16) : 0x80000000u, 0, 0);
#pragma unroll
            for (int j = 0; j < NCL; ++j)
                pcod[j] = __builtin_amdgcn_raw_buffer_load_b128(trsrc, (ptid + j * PT < kRbBytes / 16) ? ro + (unsigned)(ptid + j * PT) * 16u : 0x80000000u, 0, 0);
#pragma unroll
            for (int i = 0; i < NGL; ++i) {
                const int key = t < t_end ? __shfl(far_id, i * 16 + (lane >> 2)) : -1;   // (a load that was not issued left 0, which is a row)
                g[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, key >= 0 ? (unsigned)key * kRowB + (unsigned)(lane & 3) * 16u : 0x80000000u, 0, 0);
            }
            pesc = __builtin_amdgcn_raw_buffer_load_b8(trsrc, ptid < 8 ? ro + (unsigned)(kRecEsc + ptid) : 0x80000000u, 0, 0);
        };
        auto put_tile = [&](unsigned char *img, int image) {
#pragma unroll
            for (int j = 0; j < NCL; ++j)
                if (ptid + j * PT < kRbBytes / 16) *reinterpret_cast<u32x4 *>(img + kXBytes + (ptid + j * PT) * 16) = pcod[j];
#pragma unroll
            for (int i = 0; i < NGL; ++i) *reinterpret_cast<u32x4 *>(img + ovf_dst[i]) = g[i];
#pragma unroll
            for (int j = 0; j < NWL; ++j)
                if (ptid + j * PT < kWin * kCH) *reinterpret_cast<u32x4 *>(img + win_dst[j]) = pwin[j];
            if (ptid < 8) esc_flags[image * 8 + ptid] = (FNP_TILE_ABLATE & 1) ? 0 : (int)pesc;   // (per 32-row group)
        };
        // Iteration i: write tile i + 1 (requested last iteration) into its image as soon as the consumers have left it,
        // request tile i + 2 (its far-row ids were requested last iteration), request the far-row ids of tile i + 3.
        req_far_ids(t_begin);
        req_tile(t_begin);
        req_far_ids(t_begin + 1);
        const bool hold = __atomic_load_n(&g_tile_hold, __ATOMIC_RELAXED) != 0;
        __syncthreads();   // weights, zero rows, counters
        FNP_STAMP_DECL;
        for (int i = -1; i + 1 < nt; ++i) {
            const int t = t_begin + i, p = (i + 1) & 1;
            if (!wait_for(FREED + p, NCW * ((i + 1) / 2))) break;
            FNP_STAMP(5);
            if (!(FNP_TILE_ABLATE & 4) || i < 1) put_tile(img0 + p * kImgBytes, p);
            if (!hold) signal(READY + p);
            FNP_STAMP(0);
            if (!(FNP_TILE_ABLATE & 4)) {
                req_tile(t + 2);
                req_far_ids(t + 3);
            }
            FNP_STAMP(1);
        }
        FNP_STAMP_FLUSH(1);
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int aoff = wpos(l15, q);
    const int rloc = wave * (16 * MB) + MB * l15;    // this lane's row of block 0 inside the tile (block mb: mb rows further)
    const unsigned qx = (unsigned)q << 4;
    const int poff = (q & 1) * 32 + (q >> 1) * 16;   // epilogue: this lane's 16 bytes of a row
    float sc[2][4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, sh[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (scale) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[h][j] = scale[h * 16 + q * 4 + j];
                sh[h][j] = shift[h * 16 + q * 4 + j];
            }
    }
    __syncthreads();   // weights, zero rows, counters
#if FNP_TILE32_PRIO
    __builtin_amdgcn_s_setprio(FNP_TILE32_PRIO);   // (the consumer waves — matrix work and LDS reads — ahead of the producers' data movement)
#endif
    if ((FNP_TILE_SCHED == 2 || FNP_TILE_SCHED == 3) && wave >= 4) __builtin_amdgcn_s_sleep(4);
    FNP_STAMP_DECL;
    for (int t = t_begin; t < t_end; ++t) {
        const int tile_base = t * kTile, row_end = min(n, tile_base + kTile);
        const int image = (t - t_begin) & 1;
        if (!wait_for(READY + image, NPW * ((t - t_begin) / 2 + 1))) break;
        FNP_STAMP(2);
        const unsigned char *img = img0 + image * kImgBytes;
        // the MB entries of a lane's rows are MB consecutive 16-bit words of the [offset][row] table
        using ent_t = typename std::conditional<MB == 4, unsigned long long, unsigned>::type;
        const ent_t *rbE = reinterpret_cast<const ent_t *>(img + kXBytes) + wave * 16 + l15;
        // residual rows requested now, used after the sweep
        uint4 rv[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int r = tile_base + rloc + mb;
            rv[mb] = make_uint4(0u, 0u, 0u, 0u);
            if (residual && r < row_end) rv[mb] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(residual) + (size_t)r * kRowB + poff);
        }
        f32x4 acc[NB][MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto entry = [&](int k) -> ent_t { return rbE[(k < kK ? k : kK - 1) * (kTile / MB)]; };
        auto weights = [&](int k, frag8 (&wa)[NB]) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint4 tw = wl[(k < kK ? k : kK - 1) * kSlab + aoff + nb * 16 * kCH];
                wa[nb] = *reinterpret_cast<const frag8 *>(&tw);
            }
        };
        // The sweep: weights one offset ahead, fragments two, entries four.  ESC: this wave's entries may hold escapes.
        auto sweep = [&](auto esc_tag) {
            constexpr bool ESC = decltype(esc_tag)::value;
            auto fragments = [&](ent_t e, int k, u32x4 (&xv)[MB]) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const unsigned em = (unsigned)(e >> (16 * mb)) & 0xffffu;
                    if constexpr (ESC) {
                        if (__ballot(em == kEscape) != 0ull) {
                            // more far rows than overflow slots: this fragment comes from memory
                            unsigned off = 0x80000000u;
                            if (em == kEscape)
                                off = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(nrsrc, (unsigned)(tile_base + rloc + mb) * 4u, (unsigned)k * (unsigned)nbr_stride * 4u, 0) * kRowB +
                                      (unsigned)q * 16u;
                            const u32x4 gv = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0);
                            const u32x4 lv = *reinterpret_cast<const u32x4 *>(img + ((em == kEscape ? row_code(kZeroRow) : em) ^ qx));
                            xv[mb] = gv | lv;
                            continue;
                        }
                    }
                    xv[mb] = *reinterpret_cast<const u32x4 *>(img + (em ^ qx));
                }
            };
            // LDS reads run ahead of the matrix work: weights DW offsets, fragments DX, entries DX + 2
            constexpr int DW = FNP_TILE_DW, DX = FNP_TILE_DX;
            ent_t en[2];
            frag8 wa[DW + 1][NB];
            u32x4 xf[DX + 1][MB];
#pragma unroll
            for (int u = 0; u < DX; ++u) fragments(entry(u), u, xf[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) en[(DX + u) & 1] = entry(DX + u);
#pragma unroll
            for (int u = 0; u < DW; ++u) weights(u, wa[u]);
#pragma unroll
            for (int k = 0; k < ((FNP_TILE_ABLATE & 2) ? 0 : kK); ++k) {
                if (!(FNP_TILE_ABLATE & 16)) weights(k + DW, wa[(k + DW) % (DW + 1)]);
                const ent_t e_new = (FNP_TILE_ABLATE & 32) ? en[0] : entry(k + DX + 2);
                if (k + DX < kK && !(FNP_TILE_ABLATE & 32)) fragments(en[(k + DX) & 1], k + DX, xf[(k + DX) % (DX + 1)]);
                if constexpr (!ESC && FNP_TILE_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const frag8 xv = *reinterpret_cast<const frag8 *>(&xf[(FNP_TILE_ABLATE & 32) ? 0 : k % (DX + 1)][mb]);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (FNP_TILE_ABLATE & 64) asm volatile("" ::"v"(wa[(FNP_TILE_ABLATE & 16) ? 0 : k % (DW + 1)][nb]), "v"(xv));
                        else acc[nb][mb] = tmfma(wa[(FNP_TILE_ABLATE & 16) ? 0 : k % (DW + 1)][nb], xv, acc[nb][mb]);
                    }
                }
                if constexpr (!ESC && FNP_TILE_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
                if constexpr (!ESC && FNP_TILE_SCHED != 0 && MB == 4) {
                    // one iteration = 8 MFMAs with the 7 LDS reads (2 weight fragments, 4 row fragments, the entry word) and their
                    // address arithmetic spread between them, one read per MFMA: the ONE consumer wave of a SIMD has nobody to
                    // fill its gaps, and reads bunched together queue up in the LDS array (measured on the 64-channel kernel, round 5)
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);   // (nothing moves between iterations: the read-ahead distances stand)
                }
                if constexpr (!ESC && FNP_TILE_SCHED != 0 && MB == 2) {
                    // one iteration = 4 MFMAs with the 5 LDS reads and their address arithmetic spread between them: the two
                    // consumer waves of a SIMD then keep both pipes busy instead of bursting into each in turn
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);   // (nothing moves between iterations: the read-ahead distances stand)
                }
                en[(k + DX) & 1] = e_new;
            }
        };
        bool any_esc = false;
#pragma unroll
        for (int g = 0; g < MB / 2; ++g) any_esc |= esc_flags[image * 8 + wave * (MB / 2) + g] != 0;   // (flags per 32-row group)
        if (any_esc) sweep(std::true_type{});
        else sweep(std::false_type{});
        FNP_STAMP(0);

        // epilogue: BatchNorm(eval) scale / shift, residual, ReLU, one rounding — the arithmetic of spconv_mfma_kernel.
        // The 8-byte pieces of the two channel blocks are exchanged between the lane rows q, q ^ 1 of a site
        // (v_permlane16_swap), after which a lane holds 16 contiguous bytes of the 64-byte row.
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int r = tile_base + rloc + mb;
            const bool live = r < row_end;
            if constexpr (X3) {
                if (live) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = scale ? acc[h][mb][j] * sc[h][j] + sh[h][j] : acc[h][mb][j];
                        x3_store<TAct>(x3, (size_t)r * kC + h * 16 + q * 4, v, relu);
                    }
                }
                continue;
            }
            uint2 ra = make_uint2(0u, 0u), rbb = make_uint2(0u, 0u);
            if (residual) {
                auto t0 = __builtin_amdgcn_permlane16_swap(rv[mb].x, rv[mb].z, false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(rv[mb].y, rv[mb].w, false, false);
                ra = make_uint2(t0[0], t1[0]);    // block 0, channels q*4 .. q*4+3
                rbb = make_uint2(t0[1], t1[1]);   // block 1
            }
            uint2 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[h][mb][j];
                if (scale) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * sc[h][j] + sh[h][j];
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
            if (live && !(FNP_TILE_ABLATE & 512))
                *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(y) + (size_t)r * kRowB + poff) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            if (FNP_TILE_ABLATE & 512) asm volatile("" ::"v"(t0[0]), "v"(t1[0]), "v"(t0[1]), "v"(t1[1]));
        }
        signal(FREED + image);
        FNP_STAMP(1);
    }
    FNP_STAMP_FLUSH(0);
}


// ------------------------------------------------------------------------------------------ 64 -> 64 channels
// The same idea for the ranked 64 -> 64 layers (the four SubM convolutions of stage 3).  Their 27 weight slabs (8 KB each)
// do not fit LDS next to a tile image, so the slabs stream through a two-slot ring, one offset per slot and workgroup
// barrier, as in spconv_mfma_kernel — what goes is everything else that kernel issues per (site, offset): the int32 entry
// loads, the row-address arithmetic and the gather instructions, which (not the matrix work) bound it.  128-row tiles,
// one 256-thread workgroup (4 waves x 32 rows), two workgroups per CU (one fills its image while the other sweeps); the
// image of the next tile travels memory -> registers during the sweep and registers -> LDS between two sweeps.
template <typename TAct, bool X3 = false>
__global__ __launch_bounds__(256, 2) void spconv_tile64_kernel(const TAct *__restrict__ x, int x_bytes, const TAct *__restrict__ w,
                                                                const unsigned char *__restrict__ tile_rb, int rb_bytes,
                                                                const int *__restrict__ nbr, int nbr_stride,
                                                                const int *__restrict__ n_out, int cap, TAct *__restrict__ y,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                const TAct *__restrict__ residual, int relu, X3Args x3) {
    using G = G64;
    using frag8 = typename V16<TAct>::v8;
    using act4 = typename V16<TAct>::v4;
    constexpr int C = 64, CH = 8, KS = 2, NB = 4, MB = 2, NW = 4, NT = 256;
    constexpr int SLABC = C * CH;                                  // 512 chunks (8 KB) per weight slab
    constexpr int XB = (G::WIN + G::OVF + 1) * G::ROWB;            // feature rows of the image
    constexpr int EB = kK * G::TILE * 2;                           // entries
    constexpr int NWL = G::WIN * CH / NT, NEL = (EB / 16 + NT - 1) / NT, NOL = G::OVF * CH / NT, NSL = SLABC / NT;
    static_assert(G::WIN * CH % NT == 0 && G::OVF * CH % NT == 0 && SLABC % NT == 0 && G::TILE == NW * MB * 16 && G::OVF <= NT, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *wl = reinterpret_cast<uint4 *>(smem);                   // [2][SLABC]
    unsigned char *const img = smem + 2 * SLABC * 16;
    int *const esc_flags = reinterpret_cast<int *>(img + XB + EB);

#if FNP_CONV_PRIO
    __builtin_amdgcn_s_setprio(FNP_CONV_PRIO);   // (probe: convolution waves ahead of the index kernels that share the CUs in the replayed step)
#endif
    const int n = min(*n_out, cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)nbr, 0, kK * nbr_stride * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tile_rb, 0, rb_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, kK * C * C * 2, 0x00020000);   // (slab number in an SGPR offset)

    const int ntiles = (n + G::TILE - 1) / G::TILE;
    const int Gd = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = Gd >> 3, rem = Gd & 7;
    const int range = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    const int t_begin = (int)(((long long)ntiles * range) / Gd), t_end = (int)(((long long)ntiles * (range + 1)) / Gd);
    if (t_begin >= t_end) return;   // (whole workgroup, before any barrier)

    // weight slab image: row r (one output channel, 8 chunks) stores logical chunk c at c ^ ((r >> 1) & 7)
    const int st_pos = (tid / CH) * CH + ((tid % CH) ^ (((tid / CH) >> 1) & 7));   // (+ j * NT: 32 rows further, same swizzle)
    int aoff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) aoff[ks] = l15 * CH + ((ks * 4 + q) ^ ((l15 >> 1) & 7));
    if (tid < CH) reinterpret_cast<uint4 *>(img + G::ZERO * G::ROWB)[tid] = make_uint4(0u, 0u, 0u, 0u);

    // the next tile's image on its way: window, entries, overflow rows (4 lanes... 8 chunks per row), flags; far-row ids one tile further
    u32x4 pwin[NWL], pent[NEL], povf[NOL];
    int far_id = -1;      // thread s < OVF: row id of overflow row s
    unsigned pesc = 0;
    unsigned win_dst[NWL], ovf_dst[NOL];
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const unsigned p = (unsigned)tid + j * NT;
        win_dst[j] = G::code(tilerb::win_slot<G>(p / CH)) ^ ((p % CH) << 4);
    }
#pragma unroll
    for (int j = 0; j < NOL; ++j) {
        const unsigned p = (unsigned)tid + j * NT;
        ovf_dst[j] = G::code((unsigned)G::WIN + p / CH) ^ ((p % CH) << 4);
    }
    auto rec_off = [&](int t) -> unsigned { return t < t_end ? (unsigned)t * (unsigned)G::REC : 0x80000000u; };
    auto req_far_ids = [&](int t) {
        far_id = __builtin_amdgcn_raw_buffer_load_b32(trsrc, tid < G::OVF ? rec_off(t) + (unsigned)(G::REC_FAR + tid * 4) : 0x80000000u, 0, 0);
    };
    int *const id_lds = reinterpret_cast<int *>(img + XB + EB + 16);   // [OVF] far-row ids of the tile being requested
    float *const ss_lds = reinterpret_cast<float *>(img + XB + EB + 16 + G::OVF * 4);   // [2][C] BatchNorm scale, shift (read per tile from LDS, not L2)
    if (tid < 2 * C) ss_lds[tid] = scale ? (tid < C ? scale[tid] : shift[tid - C]) : (tid < C ? 1.f : 0.f);
    // The next tile's image is requested in PIECES (FNP_TILE64_SPREAD, round 4): vector-memory loads return in order, so behind one
    // burst of all 14 image loads the weight slab requested at the sweep's first offset — waited for at its second — came back
    // only after the whole image (~42 KB per workgroup) had arrived; one piece per offset, issued behind that offset's slab
    // request, keeps every such wait one load deep.
    constexpr int NPIECE = NWL + NEL + NOL;
    auto req_piece = [&](int t, int j) {
        const unsigned ro = rec_off(t);
        if (j < NWL) {
            const unsigned wbase = t < t_end ? (unsigned)max(0, t * G::TILE - G::HALO) * G::ROWB + (unsigned)tid * 16u : 0x80000000u;
            pwin[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, wbase + j * (NT * 16), 0, 0);
        } else if (j < NWL + NEL) {
            const int e = j - NWL;
            pent[e] = __builtin_amdgcn_raw_buffer_load_b128(trsrc, (tid + e * NT < EB / 16) ? ro + (unsigned)(tid + e * NT) * 16u : 0x80000000u, 0, 0);
            if (e == 0) pesc = __builtin_amdgcn_raw_buffer_load_b8(trsrc, tid < NW ? ro + (unsigned)(G::REC_ESC + tid) : 0x80000000u, 0, 0);
        } else if (j < NPIECE) {
            const unsigned p = (unsigned)tid + (j - NWL - NEL) * NT;
            const int key = id_lds[p / CH];
            povf[j - NWL - NEL] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, key >= 0 ? (unsigned)key * G::ROWB + (p % CH) * 16u : 0x80000000u, 0, 0);
        }
    };
    auto req_ids = [&](int t) {    // (far_id holds tile t's ids; they pass through LDS to the lanes that fetch the rows)
        if (tid < G::OVF) id_lds[tid] = t < t_end ? far_id : -1;   // (a load that was not issued left 0, which is a row)
        __syncthreads();
    };
    auto req_tile = [&](int t) {
        req_ids(t);
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) req_piece(t, j);
    };
    auto put_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NEL; ++j)
            if (tid + j * NT < EB / 16) *reinterpret_cast<u32x4 *>(img + XB + (tid + j * NT) * 16) = pent[j];
#pragma unroll
        for (int j = 0; j < NOL; ++j) *reinterpret_cast<u32x4 *>(img + ovf_dst[j]) = povf[j];
#pragma unroll
        for (int j = 0; j < NWL; ++j) *reinterpret_cast<u32x4 *>(img + win_dst[j]) = pwin[j];
        if (tid < NW) esc_flags[tid] = (FNP_TILE_ABLATE & 1) ? 0 : (int)pesc;
    };

    const int rloc = wave * 32 + 2 * l15;            // this lane's row of block 0 inside the tile (block 1: the next row)
    const int poff = (q & 1) * 32 + (q >> 1) * 16;   // epilogue: this lane's 16 bytes of a 64-byte channel-block pair
    req_far_ids(t_begin);
    req_tile(t_begin);
    req_far_ids(t_begin + 1);
    FNP_STAMP_DECL;
    for (int t = t_begin; t < t_end; ++t) {
        const int tile_base = t * G::TILE, row_end = min(n, tile_base + G::TILE);
        // Weight slabs: slab k lives in ring slot k & 1.  The A fragments of offset k + 1 are read (slot (k + 1) & 1) during
        // offset k, so a wave leaves the barrier with everything its next 16 MFMAs need in registers; slab k + 2 is stored
        // (slot k & 1, whose fragments were read an offset ago) during offset k, requested from L2 an offset before that.
        // Slabs 0 - 2 are requested before the image is written: they land meanwhile.
        constexpr int WD = FNP_TILE64_WDEPTH;   // slabs on their way to LDS in registers (slab j in wslab[j % WD])
        u32x4 wreg[2][NSL], wslab[WD][NSL];
        if (FNP_TILE_ABLATE & 2048) {
#pragma unroll
            for (int j = 0; j < NSL; ++j) wslab[1 % WD][j] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < NSL; ++j) wreg[h][j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)(tid + j * NT) * 16u, (unsigned)h * (C * C * 2), 0);
#pragma unroll
        for (int sl = 2; sl <= WD; ++sl)
#pragma unroll
            for (int j = 0; j < NSL; ++j) wslab[sl % WD][j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)(tid + j * NT) * 16u, (unsigned)sl * (C * C * 2), 0);
        FNP_STAMP(0);   // (slab requests)
        __syncthreads();   // every wave has left the previous tile's image and slabs
        FNP_STAMP(1);   // (barrier: the slowest wave's epilogue)
        put_tile();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < NSL; ++j) *reinterpret_cast<u32x4 *>(&wl[h * SLABC + st_pos + j * NT]) = wreg[h][j];
        // (the far-row ids of the NEXT tile pass through LDS here, under the same barrier as the image: their readers — the overflow
        //  pieces requested late in this sweep — come behind it, their previous readers finished before the barrier above; round 5:
        //  one workgroup barrier per tile less)
        if (FNP_TILE64_SPREAD && tid < G::OVF) id_lds[tid] = t + 1 < t_end ? far_id : -1;
        __syncthreads();
        FNP_STAMP(2);   // (image + slabs 0, 1 -> LDS, barrier)
        if (!FNP_TILE64_SPREAD) req_tile(t + 1);   // (one more barrier inside: the ids' pass through LDS)
        req_far_ids(t + 2);
        FNP_STAMP(3);   // (next tile requested)
        f32x4 acc[NB][MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned *rb32 = reinterpret_cast<const unsigned *>(img + XB) + wave * 16 + l15;   // both blocks' entries
        // (FNP_TILE_ABLATE & 1024, timing probe, wrong results: every entry names the row of zeros — all 16 lanes of a read group then
        //  share one address, a broadcast — i.e. the sweep with its B-fragment reads free of bank conflicts)
        auto entry = [&](int k) -> unsigned {
            const unsigned e = rb32[(k < kK ? k : kK - 1) * (G::TILE / 2)];
            return (FNP_TILE_ABLATE & 1024) ? ((e & 0u) | (G::code(G::ZERO) * 0x10001u)) : e;
        };
        uint4 rv[MB][NB / 2];   // residual rows: requested a few offsets before the sweep ends
        auto req_residual = [&]() {
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
        if (FNP_TILE_ABLATE & 2) req_residual();
        auto sweep = [&](auto esc_tag) {
            constexpr bool ESC = decltype(esc_tag)::value;
            constexpr bool kStoreEarly = !ESC && FNP_TILE64_SCHED == 2;
            // fragments of one offset: [ks][mb]; lane (l15, q) takes chunk 4 ks + q of its row
            auto fragments = [&](unsigned e, int k, u32x4 (&xv)[KS][MB]) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const unsigned em = mb ? e >> 16 : e & 0xffffu;
                    if constexpr (ESC) {
                        if (__ballot(em == kEscape) != 0ull) {
                            // more far rows than overflow slots: these fragments come from memory
                            unsigned off = 0x80000000u;
                            if (em == kEscape)
                                off = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(nrsrc, (unsigned)(tile_base + rloc + mb) * 4u, (unsigned)k * (unsigned)nbr_stride * 4u, 0) * G::ROWB +
                                      (unsigned)q * 16u;
                            const unsigned la = em == kEscape ? G::code(G::ZERO) : em;
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) {
                                const u32x4 gv = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off + ks * 64u, 0, 0);
                                xv[ks][mb] = gv | *reinterpret_cast<const u32x4 *>(img + (la ^ ((unsigned)(ks * 4 + q) << 4)));
                            }
                            continue;
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) xv[ks][mb] = *reinterpret_cast<const u32x4 *>(img + (em ^ ((unsigned)(ks * 4 + q) << 4)));
                }
            };
            auto weights = [&](int k, frag8 (&wa)[KS][NB]) {
                const uint4 *wk = wl + (k & 1) * SLABC;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const uint4 tw = wk[aoff[ks] + nb * 16 * CH];
                        wa[ks][nb] = *reinterpret_cast<const frag8 *>(&tw);
                    }
            };
            unsigned en[2];
            u32x4 xf[2][KS][MB];
            frag8 wa[2][KS][NB];
            fragments(entry(0), 0, xf[0]);
            weights(0, wa[0]);
            en[1] = entry(1);
            en[0] = entry(2);
            __syncthreads();   // slab 0's fragments are read: offset 0 may store slab 2 over it
#pragma unroll
            for (int k = 0; k < ((FNP_TILE_ABLATE & 2) ? 0 : kK); ++k) {
                if (k + WD + 1 < kK && !(FNP_TILE_ABLATE & (128 | 2048))) {   // (2048: probe — the slabs are stored but not loaded)
#pragma unroll
                    for (int j = 0; j < NSL; ++j)
                        wslab[(k + 1) % WD][j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)(tid + j * NT) * 16u, (unsigned)(k + WD + 1) * (C * C * 2), 0);
                }
                if (FNP_TILE64_SPREAD == 1 && k >= 1 && k <= NPIECE) req_piece(t + 1, k - 1);
                if (FNP_TILE64_SPREAD == 2) {
                    // LATE PIECES (round 5).  The image pieces are the only loads of the sweep that go to HBM (~2 us); issued one per
                    // offset from the second offset on they sat, in the wave's in-order memory queue, IN FRONT of every weight-slab
                    // load of offsets 2-15 — L2 hits that then returned with the HBM latency of the piece ahead of them (probes at
                    // 128 scenes: no slab loads -12 % of the launch, no slab stores -5 %).  Now two per offset in the LAST seven
                    // offsets, behind the sweep's last slab requests; the epilogue and the other workgroup of the CU cover their
                    // latency, and the 56 prefetch registers are free for most of the sweep (256 registers with 2 spilt -> 243, none):
                    // 0.610 -> 0.593 ms per launch at 128 scenes, 0.313 -> 0.301 at 64, 50.2 -> 48.3 us at 8 (tools/ab_tiled.py,
                    // interleaved; bit-identical).  A third slab in flight on top (FNP_TILE64_WDEPTH 3) adds nothing (0.592 / 0.305).
                    constexpr int K0 = kK - (NPIECE + 1) / 2;
                    if (k >= K0) {
                        req_piece(t + 1, 2 * (k - K0));
                        if (2 * (k - K0) + 1 < NPIECE) req_piece(t + 1, 2 * (k - K0) + 1);
                    }
                }
                if (k == kK - FNP_TILE64_RESK) req_residual();
                const unsigned e_new = entry(k + 3);
                if (k + 1 < kK) {
                    fragments(en[(k + 1) & 1], k + 1, xf[(k + 1) & 1]);
                    weights(k + 1, wa[(k + 1) & 1]);
                }
                if (kStoreEarly && k + 2 < kK && !(FNP_TILE_ABLATE & (128 | 4096))) {
#pragma unroll
                    for (int j = 0; j < NSL; ++j) *reinterpret_cast<u32x4 *>(&wl[(k & 1) * SLABC + st_pos + j * NT]) = wslab[(k + 2) % WD][j];
                }
                // the 12 LDS reads of offset k + 1 interleaved with the 16 MFMAs of offset k and nothing moved across: left to
                // itself the scheduler sinks every read to just before its first use and the wave waits for LDS four times an offset
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const frag8 xv = *reinterpret_cast<const frag8 *>(&xf[k & 1][ks][mb]);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[nb][mb] = tmfma(wa[k & 1][ks][nb], xv, acc[nb][mb]);
                    }
                if constexpr (!ESC && FNP_TILE64_SCHED) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU (fragment addresses)
                        // (SCHED 2, measured and NOT taken: the slab stores in the middle of the block instead of behind its last MFMA, so that
                        //  the barrier's LDS drain finds them done — legal at any point of offset k, slot k & 1 was last read during offset
                        //  k - 1.  0.555 -> 0.587 ms per launch at 128 scenes behind MFMA 6; with a third slab in flight 0.566 / 0.559 /
                        //  0.579 behind MFMA 6 / 9 / 2: the stores are cheapest where nothing else wants the LDS, at the end.  Round 5)
                        if (FNP_TILE64_SCHED == 2 && i == FNP_TILE64_STORE_AT) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                en[(k + 1) & 1] = e_new;
                if ((FNP_TILE_ABLATE & 4096) && k + 2 < kK) asm volatile("" ::"v"(wslab[(k + 2) % WD][0]), "v"(wslab[(k + 2) % WD][NSL - 1]));
                if (k + 1 < kK) {
                    if (!kStoreEarly && k + 2 < kK && !(FNP_TILE_ABLATE & (128 | 4096))) {   // (4096: probe — the slabs are loaded but not stored)
#pragma unroll
                        for (int j = 0; j < NSL; ++j) *reinterpret_cast<u32x4 *>(&wl[(k & 1) * SLABC + st_pos + j * NT]) = wslab[(k + 2) % WD][j];
                    }
                    if (!(FNP_TILE_ABLATE & 256)) __syncthreads();
                }
            }
        };
#if FNP_TILE64_PRIO
        __builtin_amdgcn_s_setprio(FNP_TILE64_PRIO);   // (the sweeping workgroup's waves ahead of the other workgroup's staging / epilogue waves)
#endif
        if (esc_flags[wave]) sweep(std::true_type{});
        else sweep(std::false_type{});
#if FNP_TILE64_PRIO
        __builtin_amdgcn_s_setprio(FNP_CONV_PRIO);
#endif
        FNP_STAMP(4);   // (sweep)

        // epilogue: the arithmetic of spconv_mfma_kernel (scale / shift, residual, ReLU, one rounding), 16 bytes per lane
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int r = tile_base + rloc + mb;
            const bool live = r < row_end;
            if constexpr (X3) {
                if (live) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int c0 = nb * 16 + q * 4;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[nb][mb][j];
                        if (scale) {
                            const float4 s4 = *reinterpret_cast<const float4 *>(ss_lds + c0);
                            const float4 h4 = *reinterpret_cast<const float4 *>(ss_lds + C + c0);
                            v[0] = v[0] * s4.x + h4.x; v[1] = v[1] * s4.y + h4.y; v[2] = v[2] * s4.z + h4.z; v[3] = v[3] * s4.w + h4.w;
                        }
                        x3_store<TAct>(x3, (size_t)r * C + c0, v, relu);
                    }
                }
                continue;
            }
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
                if (live) {
#if FNP_NT_STORE
                    typedef unsigned int nt4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((nt4){t0[0], t1[0], t0[1], t1[1]}, reinterpret_cast<nt4 *>(reinterpret_cast<unsigned char *>(y) + (size_t)r * (C * 2) + kp * 64 + poff));
#else
                    *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(y) + (size_t)r * (C * 2) + kp * 64 + poff) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
#endif
                }
            }
        }
        FNP_STAMP(5);   // (epilogue)
    }
    FNP_STAMP_FLUSH(0);
}

constexpr int kLds64 = 2 * 64 * 8 * 16 + (G64::WIN + G64::OVF + 1) * G64::ROWB + kK * G64::TILE * 2 + 16 + G64::OVF * 4 + 2 * 64 * 4;
static_assert(2 * kLds64 <= 160 * 1024, "two workgroups per CU");

template <typename TAct, bool X3 = false>
int launch_tile64(const void *x, long long x_bytes, const void *w, const void *tile_rb, long long rb_bytes, const int *nbr, int nbr_stride,
                  const int *n_out, int cap, void *y, const float *scale, const float *shift, const void *residual, int relu, hipStream_t s,
                  X3Args x3 = X3Args{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    auto kern = spconv_tile64_kernel<TAct, X3>;
    static bool raised = false;   // (idempotent; a race only repeats the call)
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds64) != hipSuccess) return FNP_ERR_HIP;
        raised = true;
    }
    const int tiles = fnp_divup(cap, G64::TILE);
    const int grid = tiles < 512 ? tiles : 512;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLds64, s, (const TAct *)x, (int)x_bytes, (const TAct *)w, (const unsigned char *)tile_rb, (int)rb_bytes,
                       nbr, nbr_stride, n_out, cap, (TAct *)y, scale, shift, (const TAct *)residual, relu, x3);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

template <typename TAct, bool X3 = false>
int launch_tile32(const void *x, long long x_bytes, const void *w, const void *tile_rb, long long rb_bytes, const int *nbr, int nbr_stride,
                  const int *n_out, int cap, void *y, const float *scale, const float *shift, const void *residual, int relu, hipStream_t s,
                  X3Args x3 = X3Args{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    auto kern = spconv_tile32_kernel<TAct, X3>;
    static bool raised = false;   // (idempotent; a race only repeats the call)
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return FNP_ERR_HIP;
        raised = true;
    }
    const int tiles = fnp_divup(cap, kTile);
    const int grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((16 / FNP_TILE32_MB + FNP_TILE32_NPW) * 64), kLds, s, (const TAct *)x, (int)x_bytes, (const TAct *)w, (const unsigned char *)tile_rb, (int)rb_bytes,
                       nbr, nbr_stride, n_out, cap, (TAct *)y, scale, shift, (const TAct *)residual, relu, x3);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

}  // namespace

#ifdef FNP_TILE_STAMP
extern "C" int fnp_debug_tile_stamps(unsigned long long *out32) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_tile_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return FNP_ERR_HIP;
    unsigned long long zero[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), zero, sizeof(zero)) != hipSuccess) return FNP_ERR_HIP;
    return FNP_OK;
}
#endif

extern "C" int fnp_spconv_tiled_aborts(void) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_tile_aborts), sizeof(v)) != hipSuccess) return FNP_ERR_HIP;
    return (int)v;
}

__global__ void tile_aborts_copy_kernel(int *dst) { *dst = (int)g_tile_aborts; }   // (a kernel, not a memcpy from the symbol: capturable in a hipGraph)

extern "C" int fnp_spconv_tiled_aborts_copy(int *dst, fnp_stream_t stream) {
    if (!dst) return FNP_ERR_ARG;
    hipLaunchKernelGGL(tile_aborts_copy_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

// The per-stage counts a forward hands to the host in its one synchronisation, collected by ONE launch: dst[i] = *srcs[i], and the
// time-out counter above where srcs[i] is NULL (a concatenation of one-word tensors plus a copy of the counter cost a one-scene
// forward two launches and a gap).
struct GatherJobs {
    int *src[16];
    int n;
    unsigned reset_mask;
    int *host;        // (fnp_gather_counts_host) pinned host memory: the counts and, behind them, the launch's sequence number
    unsigned *seq;    // device word: launches so far
};
// (one wave.)  The host form stores the counts into pinned host memory itself and then — behind a system-scope fence that every lane's
// count store has passed — the number of this launch in word FNP_COUNTS_SEQ_SLOT: a host thread that polls that word knows the counts
// are there without an event, a copy, or the end of the hipGraph the launch is a node of.
__global__ __launch_bounds__(64) void gather_counts_kernel(GatherJobs j, int *__restrict__ dst) {
    const int t = threadIdx.x;
    int v = 0;
    if (t < j.n) {
        v = j.src[t] ? *j.src[t] : (int)g_tile_aborts;
        dst[t] = v;
        if (j.src[t] && ((j.reset_mask >> t) & 1u)) *j.src[t] = 0;   // (a counter its owner wants at zero for the next forward)
    }
    if (j.host) {
        if (t < j.n) __hip_atomic_store(&j.host[t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (t == 0) {
            const unsigned sq = *j.seq + 1u;
            *j.seq = sq;
            __hip_atomic_store(reinterpret_cast<unsigned *>(j.host) + FNP_COUNTS_SEQ_SLOT, sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
extern "C" int fnp_gather_counts(int *const *srcs, int n, unsigned reset_mask, int *dst, fnp_stream_t stream) {
    if (!srcs || !dst || n <= 0 || n > 16) return FNP_ERR_ARG;
    GatherJobs j;
    for (int i = 0; i < 16; ++i) j.src[i] = i < n ? srcs[i] : nullptr;
    j.n = n;
    j.reset_mask = reset_mask;
    j.host = nullptr;
    j.seq = nullptr;
    hipLaunchKernelGGL(gather_counts_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, j, dst);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}
extern "C" int fnp_gather_counts_host(int *const *srcs, int n, unsigned reset_mask, int *dst, int *host_dst, unsigned *seq, fnp_stream_t stream) {
    if (!srcs || !dst || !host_dst || !seq || n <= 0 || n > 16) return FNP_ERR_ARG;
    GatherJobs j;
    for (int i = 0; i < 16; ++i) j.src[i] = i < n ? srcs[i] : nullptr;
    j.n = n;
    j.reset_mask = reset_mask;
    j.host = host_dst;
    j.seq = seq;
    hipLaunchKernelGGL(gather_counts_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, j, dst);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_debug_tile_hold(int on) {
    const int v = on ? 1 : 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_hold), &v, sizeof(v)) != hipSuccess) return FNP_ERR_HIP;
    return FNP_OK;
}

extern "C" long long fnp_tile_rulebook_bytes(int cap_out, int channels) {
    if (cap_out <= 0) return 0;
    if (channels == 32) return (long long)fnp_divup(cap_out, G32::TILE) * G32::REC;
    if (channels == 64) return (long long)fnp_divup(cap_out, G64::TILE) * G64::REC;
    return 0;
}

extern "C" int fnp_tile_rulebook_build(const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out, int channels, void *tile_rb,
                                       fnp_stream_t stream) {
    if (!nbr || !n_out || !tile_rb || K != kK || cap_out <= 0 || nbr_stride < cap_out || (channels != 32 && channels != 64)) return FNP_ERR_ARG;
    if ((uintptr_t)tile_rb & 15) return FNP_ERR_ARG;
    if (channels == 32)
        hipLaunchKernelGGL(tile_rulebook_kernel<G32>, dim3(fnp_divup(cap_out, G32::TILE)), dim3(256), 0, (hipStream_t)stream, nbr, nbr_stride, n_out, cap_out,
                           (unsigned char *)tile_rb);
    else
        hipLaunchKernelGGL(tile_rulebook_kernel<G64>, dim3(fnp_divup(cap_out, G64::TILE)), dim3(256), 0, (hipStream_t)stream, nbr, nbr_stride, n_out, cap_out,
                           (unsigned char *)tile_rb);
    FNP_LAUNCH_CHECK();
    return FNP_OK;
}

extern "C" int fnp_spconv_forward_tiled(const void *feat_in, int dtype, int n_in_rows, const void *weight, const void *tile_rb, const int *nbr,
                                        int nbr_stride, const int *n_out, int cap_out, void *feat_out, const float *scale, const float *shift,
                                        const void *residual, int relu, int Cin, int Cout, fnp_stream_t stream) {
    if (!feat_in || !weight || !tile_rb || !nbr || !n_out || !feat_out || cap_out <= 0 || nbr_stride < cap_out || n_in_rows <= 0) return FNP_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr) || Cin != Cout || (Cin != 32 && Cin != 64)) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2, rbb = fnp_tile_rulebook_bytes(cap_out, Cin);
    // 32-bit buffer offsets into the features, the int32 table (escape fetches) and the tile rulebook
    if (xb >= 0x7fffffffll || (long long)kK * nbr_stride * 4 >= 0x7fffffffll || rbb >= 0x7fffffffll) return FNP_ERR_ARG;
    if (((uintptr_t)tile_rb & 15) || ((uintptr_t)feat_in & 15) || ((uintptr_t)feat_out & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)residual & 15)) return FNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 32) {
        if (dtype == FNP_BF16) return launch_tile32<__bf16>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
        if (dtype == FNP_F16) return launch_tile32<_Float16>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
    } else {
        if (dtype == FNP_BF16) return launch_tile64<__bf16>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
        if (dtype == FNP_F16) return launch_tile64<_Float16>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, feat_out, scale, shift, residual, relu, s);
    }
    return FNP_ERR_ARG;
}

extern "C" int fnp_spconv_forward_tiled_split(const void *feat_in, int dtype, int n_in_rows, const void *weight, const void *tile_rb, const int *nbr,
                                              int nbr_stride, const int *n_out, int cap_out, float *feat_out, const float *scale, const float *shift,
                                              const float *residual, const void *addend, int relu, int Cin, int Cout, void *out_hi, void *out_lo,
                                              fnp_stream_t stream) {
    if (!feat_in || !weight || !tile_rb || !nbr || !n_out || !out_hi || !out_lo || cap_out <= 0 || nbr_stride < cap_out || n_in_rows <= 0)
        return FNP_ERR_ARG;   // (feat_out may be null: only the split is written)
    if ((scale == nullptr) != (shift == nullptr) || Cin != Cout || (Cin != 32 && Cin != 64)) return FNP_ERR_ARG;
    const long long xb = (long long)n_in_rows * Cin * 2, rbb = fnp_tile_rulebook_bytes(cap_out, Cin);
    if (xb >= 0x7fffffffll || (long long)kK * nbr_stride * 4 >= 0x7fffffffll || rbb >= 0x7fffffffll) return FNP_ERR_ARG;
    if (((uintptr_t)tile_rb & 15) || ((uintptr_t)feat_in & 15) || ((uintptr_t)feat_out & 15) || ((uintptr_t)weight & 15) || ((uintptr_t)residual & 15) ||
        (((uintptr_t)addend | (uintptr_t)out_hi | (uintptr_t)out_lo) & 7))
        return FNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const X3Args x3{residual, addend, feat_out, out_hi, out_lo};
    if (Cin == 32) {
        if (dtype == FNP_BF16) return launch_tile32<__bf16, true>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, nullptr, scale, shift, nullptr, relu, s, x3);
        if (dtype == FNP_F16) return launch_tile32<_Float16, true>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, nullptr, scale, shift, nullptr, relu, s, x3);
    } else {
        if (dtype == FNP_BF16) return launch_tile64<__bf16, true>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, nullptr, scale, shift, nullptr, relu, s, x3);
        if (dtype == FNP_F16) return launch_tile64<_Float16, true>(feat_in, xb, weight, tile_rb, rbb, nbr, nbr_stride, n_out, cap_out, nullptr, scale, shift, nullptr, relu, s, x3);
    }
    return FNP_ERR_ARG;
}
