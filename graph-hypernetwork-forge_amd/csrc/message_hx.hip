// message_hx.hip — K2+K3 with the fp32 contraction on the fp16 matrix pipe as three products of two-piece operands
// (hidden 128).  Round 1's default for d = 128; message_bx.hip (block sums in registers, larger blocks) has taken over,
// and this kernel stays selectable (GHF_KERNEL=hx) as the reference point it is measured against.
//
// Plan geometry, block sums in LDS, segment-sum scatter and fused tail are message_pp.hip's (read its header first);
// the producer / consumer wave roles are described below.  What is specific to this file is how a chunk's small GEMM
// [rows, 2d] x [2d, d] is evaluated, and why:
//
//   * v_mfma_f32_16x16x4_f32 (message_pp.hip) runs at 1/16 of the 16-bit matrix rate and bounds that kernel (7.4 ms
//     per C3 layer at 73 % MFMA-busy).
//   * With 16-bit pieces the matrix pipe stops being the limit and the bytes a CU pulls in per chunk become it
//     (tools/micro/l2stream.hip: ~33 TB/s chip-wide while the streamed footprint fits the 4 MiB L2s, ~16 TB/s at
//     12 MiB, ~7 TB/s from HBM).  Per chunk that is one relation's [2d, d] weights plus ~34 gathered rows, so the
//     weights' bytes decide: three bf16 pieces (exact, no scaling; a round-1 kernel, since removed) are 6 bytes per
//     weight and measured 5.9 ms; TWO fp16 pieces are 4 bytes — the size of the fp32 weights themselves — and 22 significand
//     bits:
//          x * 2^s = hi + lo + eps,   hi = fp16(x 2^s),  lo = fp16(x 2^s - hi),  |eps| <= 2^-22 |x 2^s|
//     where the power of two 2^s (one per activation row, one per relation's weight matrix) lifts the largest
//     element to [2^13, 2^14), so that lo stays a normal fp16 for every element within 2^-16 of the largest (smaller
//     ones keep an absolute error below 2^-38 of the largest).  A product is accumulated in fp32 from
//          hi*hi + hi*lo + lo*hi        (dropped: lo*lo <= 2^-22 |ab|)
//     with three v_mfma_f32_16x16x32_f16 per 16x16x32 block (48 cycles, against 256 for fp32 MFMAs), and the
//     exact scales 2^-s are taken out again when a K-phase ends.  Worst case 3 * 2^-22 per product, i.e. the same
//     order as the rounding of the 256-term fp32 fma chain this replaces; tests/test_hip_parity.py compares both
//     contractions with the oracle at the same tolerance and with an fp64 evaluation.
//
// Wave roles (one workgroup of 8 waves per CU; waves w and w+4 share a SIMD):
//   waves 4-7, PRODUCERS: only stage A tiles.  h_split holds every row of h already cut into its two pieces (plus
//     the row's 2^-s), so a producer moves bytes: 16-byte loads into registers two stages ahead (stage_load),
//     16-byte LDS writes one stage ahead (stage_commit), and it publishes a chunk's row words and row scales.
//   waves 0-3, CONSUMERS: B fragments straight from L2 into a 3-deep register ring, MFMAs, unscale, and after a
//     chunk's second phase its bias and the scatter into the block sums.  A wave owns 32 output columns.
// A stage = one K-phase of one chunk; one barrier per stage; two A tiles (stage parity) and two sets of row words
// (chunk parity) in LDS.  Sums are added in chunk order by the same wave: bitwise reproducible.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

template <class T>
__device__ __forceinline__ const T* hx_at(const void* base, uint32_t byte_off) {   // uniform base + 32-bit byte offset
    return (const T*)((const char*)base + byte_off);
}

// Diagnostic build only (-DGHF_STAMPS): per-wave s_memtime totals
#ifdef GHF_STAMPS
__device__ unsigned long long ghf_hx_stamp_buf[8192 * 8 * 8];
#define HX_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0
#define HX_STAMP(i)                                                                            \
    do {                                                                                       \
        unsigned long long _t;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((i) >= 0) st_acc[(i) < 0 ? 0 : (i)] += _t - st_last;                               \
        st_last = _t;                                                                          \
    } while (0)
#else
#define HX_STAMP_DECL
#define HX_STAMP(i)
#endif

// Compile-time ablations (GHF_VARIANT=exp<mask>, timing only, wrong results): 1 no B refills, 2 no gathers, 4 no main
// MFMAs, 8 no scatter (also lets the compiler drop the MFMAs: use 192), 16 gather one hot row, 32 one relation's weights,
// 64 no segment-sum MFMAs, 128 no LDS read-add-write, 256 no tail
#ifndef GHF_EXP
#define GHF_EXP 0
#endif
#ifndef GHF_B_AUX
#define GHF_B_AUX 0          // cache policy bits of the weight loads (experiment)
#endif
#ifndef GHF_OPT
#define GHF_OPT 0            // A/B switches (GHF_VARIANT=opt<mask>, same results; tools/ab.sh): none at the moment.
                             // Tried this way and measured slower on one box: next tail batch's reads issued early (3.88 vs
                             // 3.85 ms: see RB), adjacent columns per lane in the tail so that rows move as contiguous 512-byte runs (3.84 vs
                             // 3.82), A fragments read 3 positions ahead instead of 2 (3.86 vs 3.87), block-sum strips XOR-swizzled by row against the scatter's 4-way bank conflicts (3.99 vs 3.92),
                             // s_setprio 3 for the consumers (3.85 vs 3.85) or the producers (3.96), ds_add_f32 per value
                             // instead of segment sum + read-add-write (12.3 ms)
#endif

template <int D> struct HxCfg;
template <> struct HxCfg<128> { static constexpr int BN = 216, MTC = 3; };   // 159 KB LDS: 1 workgroup/CU

struct HxChunk { int r; int e0; int rows; int cross; };     // rows == 0: none

template <int D>
__global__ __launch_bounds__(512, 2) void message_hx_kernel(
    const float* __restrict__ h, const void* __restrict__ h_split, int64_t N, const uint32_t* __restrict__ sorted_key,
    const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ chunk_tab,
    const int32_t* __restrict__ item_tab, int64_t item0, float* __restrict__ partial,
    const int32_t* __restrict__ indeg, int R,
    const void* __restrict__ Wsplit, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int64_t row0, int64_t row_end, float* __restrict__ h_out, void* __restrict__ h_split_out, int no_tail, int dbg_arg,
    int32_t* __restrict__ range_flag) {
#ifdef GHF_ABLATE
    const int dbg = dbg_arg;      // 1: gather one hot row, 2: one relation's weights, 4: no main MFMAs, 8: no scatter, 16: no B loads in the stream
#else
    constexpr int dbg = 0;
#endif
    using C = HxCfg<D>;
    constexpr int BN = C::BN, MTC = C::MTC;
    constexpr int NWV = 8, TW = 4;            // waves per workgroup, per role
    constexpr int KS = D / 32;                // k-steps of 32 per phase (one 16-bit MFMA deep)
    constexpr int NKS = 2 * KS;               // k-steps of the whole contraction [h_u | h_v]
    constexpr int NT = D / 16;                // 16-column fragments of the output
    constexpr int NTW = NT / TW;              // fragments per wave (2)
    constexpr int NPL = 2;                    // pieces (hi, lo)
    constexpr int ROWB = D * 2;               // bytes per row of one fp16 plane of the A tile
    constexpr int CR = 16 * MTC;              // rows per chunk
    constexpr int PLANE = CR * ROWB;          // bytes per plane
    constexpr int GPR = D / 8;                // 16-byte granules (8 fp16) per row of a plane (16)
    constexpr int RPW = 64 / GPR;             // rows of one plane per wave-instruction (4)
    constexpr int IPW = NPL * MTC;            // 16-byte loads per producer lane per stage: piece i = (tile i / 2, plane i % 2)
    constexpr int HROW = NPL * D * 2;         // bytes per node of h_split: [2 planes][D] fp16 (the row scales follow all rows)
    constexpr int MSTR = CR + 16 + 2 * CR;    // words per chunk in s_meta: row words, run masks, row scales of phase 0 and 1
    static_assert(NTW * TW == NT && NTW == 2 && RPW * TW == 16 && GPR == 16 && MTC == 3, "bad tile config");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = (float*)smem;                    // [BN + 4][D]: block sums + 4 dummy rows
    char* Abase = (char*)(acc_lds + (BN + 4) * D);    // [2 stages][2 planes][CR][D] fp16, 16-byte granules XOR-swizzled by (row & 15)
    int* s_meta = (int*)(Abase + 2 * NPL * PLANE);    // [2 chunks][MSTR]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = w >= TW;
    const int tw = w & 3;
    const int q = lane >> 4, c16 = lane & 15;
    // work item: { block, first chunk, one past last chunk, scratch slot or -1 } (plan.hip); a heavy block (the hub
    // of a power-law graph) is several items, whose raw sums go to scratch slots and are combined by a second kernel
    const i32x4 item = *(const i32x4*)(item_tab + 4 * (size_t)(item0 + blockIdx.x));
    const int64_t blk = __builtin_amdgcn_readfirstlane(item[0]);
    const int slot = __builtin_amdgcn_readfirstlane(item[3]);
    const int64_t node0 = blk * BN;
    const int nrows = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const uint32_t seg0 = (uint32_t)(blk * R);
    auto a_tile = [&](int s) -> char* { return Abase + (s & 1) * NPL * PLANE; };  // A tile of stage s (stage = 2*chunk + phase)
    auto chunk_meta = [&](int k) -> int* { return s_meta + (k & 1) * MSTR; };     // row words / scales of the block's k-th chunk

    for (int i = tid; i < (BN + 4) * D / 4; i += NWV * 64) ((f32x4*)acc_lds)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 2 * MSTR; i += NWV * 64) s_meta[i] = (i % MSTR) < CR ? ((BN + ((i >> 2) & 3)) * (D * 4)) | (i & 15) : 0;

    HX_STAMP_DECL;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));    // opaque 0: keeps the descriptor loads on the vector path
    const int c_begin = __builtin_amdgcn_readfirstlane(item[1]);
    const int c_end = __builtin_amdgcn_readfirstlane(item[2]);
    const int nchunks = c_end - c_begin;

    // k = position in this item's chunk sequence (relation order: workgroups that run side by side then stream the
    // same relations' weights through their shared L2 at about the same time; starting every block at a different
    // relation measured 13 % slower)
    auto load_desc = [&](int k) -> i32x2 {
        const int c = c_begin + (k < nchunks ? k : 0) + vzero;          // past the end: a valid (ignored) entry
        return *hx_at<i32x2>(chunk_tab, (uint32_t)c * 8u);
    };
    // Past the end of the sequence this returns the item's first chunk again: the programs below issue the SAME loads in
    // every iteration (the extra ones, once per item, are never used).  Loads under a uniform branch would make hipcc's
    // s_waitcnt insertion assume the path without them, i.e. wait for the newest requests where the oldest are meant —
    // which silently turns every prefetch into a blocking load.
    auto decode = [&](i32x2 d) -> HxChunk {
        const int w0 = __builtin_amdgcn_readfirstlane(d[0]), w1 = __builtin_amdgcn_readfirstlane(d[1]);
        return HxChunk{w1 >> 8, w0, w1 & 127, (w1 >> 7) & 1};
    };

    // ---- producers -----------------------------------------------------------------------------------------------
    // A chunk's plan words, lane = row: one vector load per array per chunk; the gather pieces pick their rows'
    // words out of these registers with lane shuffles.  Then, lane = row again, the rows' scales 2^-s.
    struct Words { int src; int key; };
    struct Scales { float u; float v; };
    const uint32_t hsc_off = (uint32_t)((uint64_t)N * HROW);            // the row scales follow the N split rows
    auto load_words = [&](const HxChunk& c) -> Words {
        const int rc = lane < c.rows ? lane : c.rows - 1;                   // rows >= 1 here; pad lanes repeat the last row
        const uint32_t eo = (uint32_t)(c.e0 + rc) * 4u;
        return Words{*hx_at<int>(sorted_src, eo), *hx_at<int>(sorted_key, eo)};
    };
    auto load_scales = [&](const HxChunk& c, const Words& wd) -> Scales {
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        const uint32_t nu = (uint32_t)(wd.src & SRC_MASK), nv = (uint32_t)node0 + ((uint32_t)wd.key - kbase);
        return Scales{*hx_at<float>(h_split, hsc_off + nu * 4u), *hx_at<float>(h_split, hsc_off + nv * 4u)};
    };
    // Gather one stage's A tile: piece i = plane i % 2 of rows 4*tw .. 4*tw+3 of row tile i / 2, 16 lanes per row;
    // the pieces of dead tiles are skipped.  LDS image of a plane: rows of 256 bytes, the sixteen 16-byte granules of
    // a row XOR-swizzled by (row & 15), so that an MFMA fragment read (16 rows x one granule) covers all banks.
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)h_split, 0, (int)hsc_off, 0x00020000);
    auto stage_load = [&](i32x4 (&stg)[IPW], const HxChunk& c, int ph, const Words& wd) {
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        const int mts = (c.rows + 15) >> 4;                                // live row tiles
        const int word = ph == 0 ? wd.src : wd.key;
        const uint32_t nbase = ph == 0 ? 0u : (uint32_t)node0 - kbase;
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            const int v = __shfl(word, m * 16 + tw * RPW + lane / GPR, 64);
            uint32_t node = (ph == 0 ? (uint32_t)(v & SRC_MASK) : (uint32_t)v) + nbase;
            if ((dbg & 1) || (GHF_EXP & 16)) node = (uint32_t)node0;
            // no branch around the loads of a dead tile (see decode): they get an offset past the end of the buffer,
            // which the buffer's range check answers with zeros without touching memory
            const int off = m < mts ? (int)(node * (uint32_t)HROW) + ((lane % GPR) << 4) : (int)0xFFFFF000u;
            if (GHF_EXP & 2) continue;
#pragma unroll
            // source rows are read about once per CU from a 512 MB table: non-temporal, so that they do not push the
            // relations' weights (re-read by every workgroup) out of the 4 MiB L2; destination rows are re-read ~10x
            for (int pl = 0; pl < NPL; ++pl)
                stg[m * NPL + pl] = ph == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rsH, off + pl * ROWB, 0, 2)
                                            : __builtin_amdgcn_raw_buffer_load_b128(rsH, off + pl * ROWB, 0, 0);
        }
    };
    auto stage_commit = [&](const i32x4 (&stg)[IPW], char* Abuf) {
        const int r16 = tw * RPW + lane / GPR;                             // row within its tile
        char* dst = Abuf + r16 * ROWB + (((lane % GPR) ^ r16) << 4);
#pragma unroll
        for (int m = 0; m < MTC; ++m)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *(i32x4*)(dst + m * 16 * ROWB + pl * PLANE) = stg[m * NPL + pl];
    };
    // a chunk's words for the consumers: (byte offset of the row's target in the block sums) | run head; per row
    // tile the mask of rows that continue a run of equal destinations; the rows' scales for either phase
    auto publish_rows = [&](const HxChunk& c, const Words& wd, const Scales& sc, int* meta) {
        if (tw != 0) return;
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        const int head = (int)((uint32_t)wd.src >> SRC_BITS), row16 = lane & 15;
        const bool live = lane < c.rows;
        const int tgt = (live && head == row16) ? (int)((uint32_t)wd.key - kbase) : BN + ((lane >> 2) & 3);
        const unsigned long long runs = __ballot(live && head != row16);
        if (lane < CR) {
            meta[lane] = (tgt * (D * 4)) | (live ? head : row16);          // D*4 = 512: the low 4 bits stay free
            meta[CR + 16 + lane] = __float_as_int(sc.u);
            meta[CR + 16 + CR + lane] = __float_as_int(sc.v);
        }
        if (lane < MTC) meta[CR + lane] = (int)((runs >> (16 * lane)) & 0xFFFFull);
    };

    // ---- consumers -----------------------------------------------------------------------------------------------
    // B fragments (GHF_WLAYOUT_SPLIT2H, written by K1): Wh[r][o/16][kk/32][piece][lane = ((kk%32)/8)*16 + o%16][kk%8]
    // fp16, kk in [0, 2d), followed by one float 2^-s per relation.  Buffer loads: resource descriptor + scalar offset
    // (relation, fragment, k-step) + lane*16 + immediate (piece), so one VGPR addresses them all.
    const uint32_t wsc_off = (uint32_t)((uint64_t)R * 2 * D * D * (NPL * 2));
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wsplit, 0, (int)wsc_off, 0x00020000);
    auto b_soff = [&](int r, int ph, int t) -> int {
        if ((dbg & 2) || (GHF_EXP & 32)) r = 0;
        return __builtin_amdgcn_readfirstlane((((r * NT + tw * NTW + t) * NKS + ph * KS) * NPL) * 1024);
    };
    const int lane16 = lane * 16;
    // The B pieces of one whole stage live in registers, slot = k-step.  As soon as the MFMAs of k-step j have been
    // issued, slot j is refilled with k-step j of the NEXT stage: every request has a full stage period (MFMAs,
    // unscale, scatter, barrier) to arrive.  (Two stages deep, 128 KB in flight per CU, measured the same.)
    i32x4 b[KS][NTW][NPL];
    auto load_b_step = [&](int r, int ph, int j) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                b[j][t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16 + pl * 1024, b_soff(r, ph, t) + j * (NPL * 1024), GHF_B_AUX);
    };

    f32x4 acc[MTC][NTW];
    // One stage = one K-phase of the chunk = KS k-steps of 32.  Per (k-step, row tile): the 2 A-piece fragments from
    // LDS (read one step ahead), and for each of the wave's column fragments the three piece products, smallest
    // first, into per-stage accumulators; when the phase ends they are scaled by 2^-s(row) 2^-s(relation) — exact —
    // and added to the chunk's rows.  One code path: dead row tiles (m >= mt) skip their MFMAs.
    // ---- scatter --------------------------------------------------------------------------------------------------
    // After a chunk's second phase its rows (acc) join the block sums.  All MTC tiles, no branch around memory
    // operations (see decode): rows of dead tiles have dummy targets.
    // Segment sum: a tile with a run of equal destinations is multiplied by S[i][k] = (head(k) == i) on the fp32 MFMA;
    // then a plain LDS read-add-write through inline asm (see message_pp.hip).  The block sums keep a wave's 32
    // columns INTERLEAVED (LDS position 32*tw + 2*c16 + t holds column 32*tw + 16*t + c16; the tail undoes it), so a
    // lane's two values are adjacent and move with one 64-bit access.
    // (Interleaving these pieces with the next chunk's MFMAs was tried and measured the same: the stage period is set
    // by how long the weights take to arrive, not by this wave's instruction count.)
    const unsigned strip = (unsigned)(size_t)(lptr_t)(acc_lds + tw * 16 * NTW + c16 * NTW);
    auto scatter_chunk = [&](const int* meta, bool in_order) {
        i32x4 mq[MTC];
#pragma unroll
        for (int m = 0; m < MTC; ++m) mq[m] = *(const i32x4*)(meta + m * 16 + 4 * q);
        const i32x4 runs = *(const i32x4*)(meta + CR);
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            if (__builtin_amdgcn_readfirstlane(runs[m]) && !(GHF_EXP & 64)) {
                f32x4 y[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) y[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float sel = ((mq[m][s] & 15) == c16) ? 1.0f : 0.0f;
#pragma unroll
                    for (int t = 0; t < NTW; ++t) y[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, acc[m][t][s], y[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[m][t] = y[t];
            }
        }
        unsigned addr[MTC][4];
        f32x2 v[MTC][4];
        auto rd = [&](int m) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                addr[m][s] = strip + ((unsigned)mq[m][s] & ~15u);           // the run's target row, or a dummy
                asm volatile("ds_read_b64 %0, %1" : "=v"(v[m][s]) : "v"(addr[m][s]) : "memory");
            }
        };
        auto wr = [&](int m) {
            asm volatile("" : "+v"(v[m][0]), "+v"(v[m][1]), "+v"(v[m][2]), "+v"(v[m][3]));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x2 r = v[m][s] + (f32x2){acc[m][0][s], acc[m][1][s]};
                asm volatile("ds_write_b64 %0, %1" :: "v"(addr[m][s]), "v"(r) : "memory");
            }
        };
        if (GHF_EXP & 128) {                            // keep the rows alive (no dead-code elimination of the MFMAs), skip the LDS part
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) asm volatile("" :: "v"(acc[m][t]));
            return;
        }
        if (!in_order) {                                // no run continues into the next tile: the tiles touch disjoint rows
#pragma unroll
            for (int m = 0; m < MTC; ++m) rd(m);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < MTC; ++m) wr(m);
        } else {
#pragma unroll
            for (int m = 0; m < MTC; ++m) {
                rd(m);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wr(m);
            }
        }
    };

    const int arow = c16 * ROWB;
    auto compute_stage = [&](int mt, int ph, const char* Abuf, const int* meta, float wscale, int r_next, int ph_next,
                             const float (&bias_v)[NTW]) {
        f32x4 part[MTC][NTW];
#pragma unroll
        for (int m = 0; m < MTC; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) part[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 sc[MTC];                                     // the rows' scales (rows 4q .. 4q+3 of tile m), needed when the phase ends
#pragma unroll
        for (int m = 0; m < MTC; ++m) sc[m] = *(const f32x4*)(meta + CR + 16 + ph * CR + m * 16 + 4 * q);
        i32x4 a[3][NPL];                                   // A fragments of (k-step, tile) positions p, p+1, p+2
        auto lda = [&](int j, int m, i32x4 (&dst)[NPL]) {
            const char* src = Abuf + arow + (((4 * j + q) ^ c16) << 4) + m * 16 * ROWB;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) dst[pl] = *(const i32x4*)(src + pl * PLANE);
        };
        lda(0, 0, a[0]);
        lda(0, 1, a[1]);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int m = 0; m < MTC; ++m) {
                const int p = j * MTC + m, cur = p % 3;
                if (p + 2 < KS * MTC) lda((p + 2) / MTC, (p + 2) % MTC, a[(p + 2) % 3]);
                if (GHF_EXP & 4) {                                          // no MFMAs, but their operands stay alive (no DCE of the loads)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) asm volatile("" :: "v"(a[cur][pl]));
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) asm volatile("" :: "v"(b[j][t][pl]));
                } else if (m < mt && !(dbg & 4)) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        auto fma = [&](int pa, int pb) {
                            part[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[cur][pa]),
                                                                                __builtin_bit_cast(f16x8, b[j][t][pb]),
                                                                                part[m][t], 0, 0, 0);
                        };
                        fma(1, 0); fma(0, 1);                               // lo*hi, hi*lo
                        fma(0, 0);                                          // hi*hi
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 16) && !(GHF_EXP & 1)) load_b_step(r_next, ph_next, j);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) HX_STAMP(5); else if (j == KS - 1) HX_STAMP(7);
        }
        // take the scales out: rows 4q .. 4q+3 of tile m; bias[r], once per edge row, is the addend of phase 0
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            if (m >= mt) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float f = sc[m][s] * wscale;
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[m][t][s] = fmaf(part[m][t][s], f, ph == 0 ? bias_v[t] : acc[m][t][s]);
            }
        }
    };

    __syncthreads();                                   // sums zeroed, row words initialised
    HX_STAMP(-1);

    // Two programs, 2*nchunks + 1 barriers each.  Barrier interval s belongs to stage s:
    //   consumers:  MFMAs of stage s from A tile s & 1; after a chunk's phase 1, its bias and its scatter
    //   producers:  request the gather of stage s + 2 into registers, then write the rows of stage s + 1 (requested
    //               one interval ago) into the other A tile: a gather has a whole interval to arrive.  A chunk's
    //               words are published together with its phase-0 tile, one interval before the consumers need them.
    if (producer) {
        i32x4 stgA[IPW], stgB[IPW];                    // phase-0 / phase-1 stages in flight
        Words wdI{0, 0}, wdN{0, 0};
        Scales scN{1.f, 1.f};
        HxChunk chI{0, 0, 1, 0}, chN{0, 0, 1, 0};
        i32x2 dNN{0, 0};
        if (nchunks > 0) {
            chI = decode(load_desc(0));
            chN = decode(load_desc(1));
            dNN = load_desc(2);
            wdI = load_words(chI);
            wdN = load_words(chN);
            const Scales scI = load_scales(chI, wdI);
            stage_load(stgA, chI, 0, wdI);
            stage_load(stgB, chI, 1, wdI);
            stage_commit(stgA, a_tile(0));
            publish_rows(chI, wdI, scI, chunk_meta(0));
        }
        for (int k = 0; k < nchunks; ++k) {            // chI = chunk k, chN = chunk k + 1 (past the end: see decode)
            __syncthreads();                           // interval 2k
            HX_STAMP(0);
            scN = load_scales(chN, wdN);
            stage_load(stgA, chN, 0, wdN);
            HX_STAMP(2);
            stage_commit(stgB, a_tile(1));
            HX_STAMP(3);
            __syncthreads();                           // interval 2k + 1
            HX_STAMP(0);
            stage_load(stgB, chN, 1, wdN);
            HX_STAMP(2);
            stage_commit(stgA, a_tile(0));
            publish_rows(chN, wdN, scN, chunk_meta(k + 1));
            chI = chN;
            wdI = wdN;
            chN = decode(dNN);
            wdN = load_words(chN);
            dNN = load_desc(k + 3);
            HX_STAMP(3);
        }
        __syncthreads();
    } else {
        HxChunk ch{0, 0, 1, 0};
        i32x2 dn{0, 0};
        // a chunk's small per-relation words (weight scale, bias) are requested one chunk ahead: a load issued at the
        // start of the stage that needs it at its end sat in the CU's memory queue longer than the stage's MFMAs take
        float bias_v[NTW] = {}, bias_n[NTW] = {}, wscale = 1.f, wscale_n = 1.f;
        auto load_rel_words = [&](int r, float& ws, float (&bv)[NTW]) {
            ws = *hx_at<float>(Wsplit, wsc_off + (uint32_t)((((dbg & 2) || (GHF_EXP & 32)) ? 0 : r) + vzero) * 4u);
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[t] = *hx_at<float>(bias, (uint32_t)(r * D + (tw * NTW + t) * 16 + c16) * 4u);
        };
        if (nchunks > 0) {
            ch = decode(load_desc(0));
            dn = load_desc(1);
            load_rel_words(ch.r, wscale, bias_v);
#pragma unroll
            for (int j = 0; j < KS; ++j) load_b_step(ch.r, 0, j);
        }
        for (int k = 0; k < nchunks; ++k) {
            const int mt = (ch.rows + 15) >> 4;
            const int* meta = chunk_meta(k);
            __syncthreads();                           // interval 2k: phase 0
            HX_STAMP(0);
            const HxChunk nx = decode(dn);
            compute_stage(mt, 0, a_tile(0), meta, wscale, ch.r, 1, bias_v);
            HX_STAMP(1);
            __syncthreads();                           // interval 2k + 1: phase 1, then the chunk's rows join the sums
            HX_STAMP(0);
            dn = load_desc(k + 2);                     // BEFORE this stage's B refills: its use then waits for nothing younger
            load_rel_words(nx.r, wscale_n, bias_n);
            compute_stage(mt, 1, a_tile(1), meta, wscale, nx.r, 0, bias_v);
            HX_STAMP(1);
            if (!(dbg & 8) && !(GHF_EXP & 8)) scatter_chunk(meta, ch.cross != 0);
            ch = nx;
            wscale = wscale_n;
#pragma unroll
            for (int t = 0; t < NTW; ++t) bias_v[t] = bias_n[t];
            HX_STAMP(4);
        }
        __syncthreads();
    }

    // ---- fused tail: one wave per destination row, RB rows in flight -----------------------------------------
    if (GHF_EXP & 256) return;                              // (timing only: no tail — it costs 0.43 of 4.1 ms)
    constexpr int CPL = D / 64;
    // LDS position lane*CPL + c of a row of the sums holds output column col[c] (see the scatter): the lane's two
    // positions are columns o and o + 16.
    int col[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) col[c] = 32 * (lane >> 4) + 16 * c + (lane & 15);
    float gm[CPL], bt[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        gm[c] = no_tail ? 1.f : gamma[col[c]];
        bt[c] = no_tail ? 0.f : beta[col[c]];
    }
    if (slot >= 0) {                                   // one item of a split block: raw sums (column order) to my slot
        float* __restrict__ ps = partial + (size_t)slot * BN * D;
        for (int v = w; v < BN; v += NWV)
#pragma unroll
            for (int c = 0; c < CPL; ++c) ps[(size_t)v * D + col[c]] = acc_lds[v * D + lane * CPL + c];
        return;
    }
    // Rows v0, v0 + NWV, ... of one batch: their global reads first, then the arithmetic.
    // RB = 14: a wave's 27 rows in two batches.  A batch is one memory round trip plus the wait for the previous
    // batch's stores (vmcnt retires in order), so fewer, larger batches: RB 4 -> 9 -> 14 measured 3.85 -> 3.75 ->
    // 3.72 ms per launch on one box, 27 the same as 14.
    constexpr int RB = 14;
    auto load_batch = [&](int v0, float (&x)[RB][CPL], float (&inv)[RB]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int v = v0 + rb * NWV;
            const int64_t node = node0 + (v < nrows ? v : nrows - 1);
            const int deg = indeg[node];
            inv[rb] = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg > 1 ? deg : 1);
#pragma unroll
            for (int c = 0; c < CPL; ++c) x[rb][c] = no_tail ? 0.f : h[(size_t)node * D + col[c]];
        }
    };
    auto finish_batch = [&](int v0, float (&x)[RB][CPL], const float (&inv)[RB]) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int v = v0 + rb * NWV;
            const int vc = v < nrows ? v : nrows - 1;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const float a = acc_lds[vc * D + lane * CPL + c] * inv[rb];
                x[rb][c] = no_tail ? a : fmaxf(a + x[rb][c], 0.f);
                s += x[rb][c];
            }
            if (!no_tail) {
                const float mean = wave_sum(s) * (1.0f / D);
                float var = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) { const float t = x[rb][c] - mean; var += t * t; }
                const float rstd = 1.0f / sqrtf(wave_sum(var) * (1.0f / D) + eps);
#pragma unroll
                for (int c = 0; c < CPL; ++c) x[rb][c] = (x[rb][c] - mean) * rstd * gm[c] + bt[c];
            }
            // the same row cut into fp16 pieces, for the next layer's gathers (uniform branch: all lanes reduce)
            float up = 1.f;
            if (h_split_out) {
                float mx = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) mx = fmaxf(mx, fabsf(x[rb][c]));
                const int sh = split2h_shift(wave_absmax(mx));
                up = pow2f(sh);
                if (lane == 0 && v < nrows) *(float*)((char*)h_split_out + (size_t)hsc_off + (size_t)(node0 + v) * 4) = pow2f(-sh);
            }
            if (v < nrows) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) h_out[(size_t)(node0 + v) * D + col[c]] = x[rb][c];
                if (h_split_out) {
                    _Float16* __restrict__ sp = (_Float16*)h_split_out + (size_t)(node0 + v) * (NPL * D);
                    int tiny = 0, nz = 0;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        _Float16 hi, lo;
                        split2h(x[rb][c] * up, hi, lo);
                        sp[col[c]] = hi;
                        sp[D + col[c]] = lo;
                        tiny += range_tiny(x[rb][c] * up);
                        nz += x[rb][c] != 0.f;
                    }
                    if (__ballot(tiny != 0)) {                          // (rare) some lane of this row holds a tiny entry
                        tiny = (int)wave_sum((float)tiny);
                        nz = (int)wave_sum((float)nz);
                        if (lane == 0) range_raise(range_flag, GHF_RANGE_ROWS, tiny, nz);
                    }
                }
            }
        }
    };
    // (Issuing the next batch's reads before this batch's arithmetic was measured on one box against this loop: 3.88
    // vs 3.85 ms per launch — the tail is not waiting for memory.)
    for (int v0 = w; v0 < nrows; v0 += NWV * RB) {
        float x[RB][CPL], inv[RB];
        load_batch(v0, x, inv);
        finish_batch(v0, x, inv);
    }
#ifdef GHF_STAMPS
    HX_STAMP(6);                                        // drain + tail
    if (lane == 0 && blockIdx.x < 8192)
        for (int i = 0; i < 8; ++i) ghf_hx_stamp_buf[((size_t)blockIdx.x * 8 + w) * 8 + i] = st_acc[i];
#endif
}

template <int D>
static int launch_hx_for(const MsgArgs& a, hipStream_t stream) {
    using C = HxCfg<D>;
    constexpr int CR = 16 * C::MTC;
    constexpr size_t lds = (size_t)(C::BN + 4) * D * 4 + (size_t)2 * 2 * CR * (D * 2) + 2 * (CR + 16 + 2 * CR) * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    GHF_REQUIRE(a.block_nodes == C::BN, "message(hx): plan block_nodes=%d, kernel for d=%d needs %d", a.block_nodes, D, C::BN);
    GHF_REQUIRE(a.wlayout == GHF_WLAYOUT_SPLIT2H, "message(hx): weights must be in SPLIT2H layout");
    GHF_REQUIRE(a.chunk_tab && a.item_tab && a.blk_item_off, "message(hx): the plan's chunk / item tables are missing");
    GHF_REQUIRE(a.h_split, "message(hx): h_split is missing (ghf_split_rows)");
    const int64_t row_end = a.row0 + a.rows;
    GHF_REQUIRE(row_end == a.N || row_end % C::BN == 0, "message(hx): row range must end on a block boundary or at N");
    if (a.rows <= 0) return GHF_OK;
    GHF_REQUIRE((uint64_t)a.N * (D * 4 + 4) <= 0xFFFFF000ull && (uint64_t)a.E * 4 < (1ull << 32) &&
                    (uint64_t)a.R * (2 * D * D * 4 + 4) < (1ull << 32),
                "message(hx): 32-bit byte offsets need N*(4d+4), E*4 and R*(8d*d+4) below 4 GiB");
    static const int dbg = getenv("GHF_DEBUG_FLAGS") ? atoi(getenv("GHF_DEBUG_FLAGS")) : 0;   // honoured by -DGHF_ABLATE builds only
    GHF_SET_MAX_LDS(message_hx_kernel<D>, lds);
    GHF_REQUIRE(a.n_items >= cdiv(a.rows, C::BN), "message(hx): n_items=%lld is fewer than the blocks of the row range", (long long)a.n_items);
    GHF_REQUIRE(a.n_items == cdiv(a.rows, C::BN) || a.partial, "message(hx): split blocks need the `partial` scratch");
    message_hx_kernel<D><<<(unsigned)a.n_items, 512, lds, stream>>>(a.h, a.h_split, a.N, a.sorted_key, a.sorted_src, a.chunk_tab, a.item_tab,
                                                                   a.item0, a.partial, a.indeg, a.R, a.W_msg, a.bias, a.ln_gamma,
                                                                   a.ln_beta, a.ln_eps, a.row0, row_end, a.h_out, a.h_split_out,
                                                                   a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM), dbg, range_flag_ptr());
    GHF_LAUNCH_CHECK();
    if (a.n_items > cdiv(a.rows, C::BN)) return launch_combine_split(a, stream);     // some block of the range is split
    return GHF_OK;
}

// ghf_split_rows, SPLIT2H: one wave per row — the row's largest magnitude picks the power of two
__global__ __launch_bounds__(256) void split2h_rows_kernel(const float* __restrict__ h, int64_t N, int64_t row0, int64_t rows,
                                                           int d, char* __restrict__ out, int32_t* __restrict__ range_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const int64_t row = row0 + i;
    const float* __restrict__ src = h + row * d;
    float mx = 0.f;
    for (int k = lane; k < d; k += 64) mx = fmaxf(mx, fabsf(src[k]));
    const int sh = split2h_shift(wave_absmax(mx));
    const float up = pow2f(sh);
    _Float16* __restrict__ dst = (_Float16*)(out + row * (4 * (int64_t)d));
    int tiny = 0, nz = 0;
    for (int k = lane; k < d; k += 64) {
        _Float16 hi, lo;
        const float xs = src[k] * up;
        split2h(xs, hi, lo);
        dst[k] = hi;
        dst[d + k] = lo;
        tiny += __popcll(__ballot(range_tiny(xs)));
        nz += __popcll(__ballot(xs != 0.f));
    }
    if (lane == 0) {
        *(float*)(out + N * (4 * (int64_t)d) + row * 4) = pow2f(-sh);
        range_raise(range_flag, GHF_RANGE_ROWS, tiny, nz);
    }
}

int launch_split2h_rows(const float* h, int64_t N, int d, int64_t row0, int64_t rows, void* h_split, hipStream_t stream) {
    if (rows <= 0) return GHF_OK;
    GHF_REQUIRE(cdiv(rows, 4) < (1ll << 31), "split_rows: too many rows per launch");
    split2h_rows_kernel<<<(unsigned)cdiv(rows, 4), 256, 0, stream>>>(h, N, row0, rows, d, (char*)h_split, range_flag_ptr());
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

bool message_hx_config(int d, int* block_nodes, int* chunk_rows, int* split_chunks) {
    if (d != 128) return false;
    *block_nodes = HxCfg<128>::BN;
    *chunk_rows = 16 * HxCfg<128>::MTC;
    *split_chunks = 128;
    return true;
}

int launch_message_hx(const MsgArgs& a, hipStream_t stream) {
    if (a.d == 128) return launch_hx_for<128>(a, stream);
    return set_err(GHF_EUNSUPPORTED, "message(hx): no fp16 two-piece kernel for d=%d", a.d);
}

}  // namespace ghf

#ifdef GHF_STAMPS
extern "C" int ghf_debug_read_stamps_hx(unsigned long long* host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ghf::ghf_hx_stamp_buf), count * sizeof(unsigned long long));
}
#endif
