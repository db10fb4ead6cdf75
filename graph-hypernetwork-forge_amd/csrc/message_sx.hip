// message_sx.hip — K2+K3 with the fp32 contraction run on the bf16 matrix pipe as six split products (hidden 128).
//
// Same plan geometry, ping-pong schedule, segment-sum scatter and fused tail as message_pp.hip (read its header and
// message_mfma.hip's first).  What differs is how a chunk's small GEMM  [rows, 2d] x [2d, d]  is evaluated:
//
//   v_mfma_f32_16x16x4_f32 (message_pp.hip) runs at 1/16 of the bf16 matrix rate, and it is what bounds that kernel
//   (73 % MFMA-busy at 56 % of the fp32 matrix peak).  Here every fp32 operand is cut, EXACTLY, into three bf16
//   pieces by truncation,
//        x = x1 + x2 + x3,   x1 = top 8 significand bits, x2 = the next 8, x3 = the last 8  (24 = 8 + 8 + 8),
//   and a product a*b is accumulated in fp32 from the six piece products of weight >= 2^-16:
//        a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)
//   each of them exact in fp32 (8-bit x 8-bit significands).  Dropped: a2b3 + a3b2 + a3b3 <= 2^-23 |ab|, i.e. below
//   half an ulp of the product — the same order as the rounding of the fp32 fma chain this replaces.  Six
//   v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) do the work of eight v_mfma_f32_16x16x4_f32 (32 cycles each,
//   K = 4 each): 96 cycles instead of 256 per 16x16x32 block.
//
//   Weights arrive pre-split from K1 (GHF_WLAYOUT_SPLIT3: bf16 B fragments, 3 pieces); gathered h rows are split by
//   the PREP team while it stages them, and the A tile in LDS holds three bf16 planes ([3][48 rows][128] bf16,
//   16-byte granules XOR-swizzled by row & 15).  At 6 bytes per A element the two teams' tiles take 72 KB, so a
//   workgroup owns BN = 162 destination nodes (83 KB of fp32 sums) instead of 216.
//
// Everything else — who scatters when, the order of the sums, the tail — is message_pp.hip's, so results are still
// bitwise reproducible run to run.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SX_WAIT_VMCNT0 = 0x0F70;      // s_waitcnt vmcnt(0) only (builtin form: modelled by hipcc)

// uniform base + 32-bit byte offset: lets the backend use the SGPR-base addressing form (one VGPR per address);
// with 64-bit per-lane pointers the loop-invariant parts hoisted out of the chunk loop spilled.  All arrays
// indexed this way are < 4 GiB here (checked by the launcher).
template <class T>
__device__ __forceinline__ const T* at(const void* base, uint32_t byte_off) {
    return (const T*)((const char*)base + byte_off);
}

// Diagnostic build only (-DGHF_STAMPS): per-wave s_memtime totals, as in message_mfma.hip
#ifdef GHF_STAMPS
__device__ unsigned long long ghf_sx_stamp_buf[8192 * 8 * 8];
#define PP_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0
#define PP_STAMP(i)                                                                            \
    do {                                                                                       \
        unsigned long long _t;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((i) >= 0) st_acc[(i) < 0 ? 0 : (i)] += _t - st_last;                               \
        st_last = _t;                                                                          \
    } while (0)
#else
#define PP_STAMP_DECL
#define PP_STAMP(i)
#endif

template <int D> struct SxCfg;
template <> struct SxCfg<128> { static constexpr int BN = 162, MTC = 3, WAVES_PER_SIMD = 2; };   // 162 KB LDS: 1 workgroup/CU

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = p1 + p2 + p3 exactly (truncation: the 24 significand bits cut 8 + 8 + 8); the pieces of 4 consecutive
// elements packed as 4 bf16 = 8 bytes per plane
__device__ __forceinline__ void split3(const f32x4& x, i32x2 (&p)[3]) {
    int u1[4], u2[4], u3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u1[e] = __float_as_int(x[e]);
        const float r1 = x[e] - __int_as_float(u1[e] & 0xFFFF0000);
        u2[e] = __float_as_int(r1);
        const float r2 = r1 - __int_as_float(u2[e] & 0xFFFF0000);
        u3[e] = __float_as_int(r2);
    }
    // v_perm_b32: the upper halves of two dwords -> one dword (element e in the low half)
    p[0] = (i32x2){(int)__builtin_amdgcn_perm(u1[1], u1[0], 0x07060302), (int)__builtin_amdgcn_perm(u1[3], u1[2], 0x07060302)};
    p[1] = (i32x2){(int)__builtin_amdgcn_perm(u2[1], u2[0], 0x07060302), (int)__builtin_amdgcn_perm(u2[3], u2[2], 0x07060302)};
    p[2] = (i32x2){(int)__builtin_amdgcn_perm(u3[1], u3[0], 0x07060302), (int)__builtin_amdgcn_perm(u3[3], u3[2], 0x07060302)};
}

struct SxChunk { int r; int e0; int rows; int cross; };     // rows == 0: none

template <int D>
__global__ __launch_bounds__(512, SxCfg<D>::WAVES_PER_SIMD) void message_sx_kernel(
    const float* __restrict__ h, const void* __restrict__ h_split, int64_t N, const uint32_t* __restrict__ sorted_key,
    const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ chunk_tab,
    const int32_t* __restrict__ item_tab, int64_t item0, float* __restrict__ partial,
    const int32_t* __restrict__ indeg, int R,
    const void* __restrict__ Wsplit, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int64_t row0, int64_t row_end, float* __restrict__ h_out, void* __restrict__ h_split_out, int no_tail, uint32_t w_bytes,
    int dbg_arg) {
#ifdef GHF_ABLATE
    const int dbg = dbg_arg;      // 1: gather one hot row, 2: one relation's weights, 4: no main MFMAs, 8: no scatter, 16: no B loads in the stream, 32: two B pieces / three products only
#else
    constexpr int dbg = 0;
#endif
    using C = SxCfg<D>;
    constexpr int BN = C::BN, MTC = C::MTC;
    constexpr int NWV = 8, TW = 4;            // waves per workgroup, per role
    constexpr int KS = D / 32;                // k-steps of 32 per phase (one bf16 MFMA deep)
    constexpr int NKS = 2 * KS;               // k-steps of the whole contraction [h_u | h_v]
    constexpr int NT = D / 16;                // 16-column fragments of the output
    constexpr int NTW = NT / TW;              // fragments per wave (2)
    constexpr int ROWB = D * 2 + 16;          // bytes per row of one bf16 plane of the A tile: padded by one granule, so
                                              // that 16 rows x one granule cover all banks with plain immediate offsets
    constexpr int PLANE = 16 * MTC * ROWB;    // bytes per plane
    constexpr int CR = 16 * MTC;              // rows per chunk
    constexpr int GPR = D / 8;                // 16-byte granules (8 bf16) per row of a plane (16)
    constexpr int RPW = 64 / GPR;             // rows of one plane per wave-instruction (4)
    constexpr int IPW = 3 * MTC;              // 16-byte loads per producer lane per stage: piece i = (tile i / 3, plane i % 3)
    constexpr int HROW = 3 * D * 2;           // bytes per node of h_split: [3 planes][D] bf16
    static_assert(NTW * TW == NT && NTW == 2 && RPW * TW == 16 && MTC == 3, "bad tile config");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = (float*)smem;                    // [BN + 4][D]: block sums + 4 dummy rows
    char* Abase = (char*)(acc_lds + (BN + 4) * D);    // [2 stages][3 planes][CR][D + 8] bf16
    constexpr int MSTR = CR + 16;                     // ints of row words per chunk: CR rows, then one run mask per row tile
    int* s_meta = (int*)(Abase + 2 * 3 * PLANE);      // [2 chunks][MSTR]: row words (target row << 4) | run head; per tile: rows continuing a run

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = w >= TW;                    // waves 4..7 stage the A tiles, waves 0..3 run the MFMAs and the sums;
    const int tw = w & 3;                             // waves w and w+4 share a SIMD: one of each kind per SIMD
    const int q = lane >> 4, c16 = lane & 15;
    // work item: { block, first chunk, one past last chunk, scratch slot or -1 } (plan.hip); a heavy block (the hub
    // of a power-law graph) is several items, whose raw sums go to scratch slots and are combined by a second kernel
    const i32x4 item = *(const i32x4*)(item_tab + 4 * (size_t)(item0 + blockIdx.x));
    const int64_t blk = __builtin_amdgcn_readfirstlane(item[0]);
    const int slot = __builtin_amdgcn_readfirstlane(item[3]);
    const int64_t node0 = blk * BN;
    const int nrows = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const uint32_t seg0 = (uint32_t)(blk * R);
    auto a_tile = [&](int s) -> char* { return Abase + (s & 1) * 3 * PLANE; };    // A tile of stage s (stage = 2*chunk + phase)
    auto row_words = [&](int k) -> int* { return s_meta + (k & 1) * MSTR; };      // row words of the block's k-th chunk

    for (int i = tid; i < (BN + 4) * D / 4; i += NWV * 64) ((f32x4*)acc_lds)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 2 * MSTR; i += NWV * 64) s_meta[i] = (i % MSTR) < CR ? ((BN + ((i >> 2) & 3)) * (D * 4)) | (i & 15) : 0;

    PP_STAMP_DECL;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));    // opaque 0: keeps the descriptor loads on the vector path
    const int c_begin = __builtin_amdgcn_readfirstlane(item[1]);
    const int c_end = __builtin_amdgcn_readfirstlane(item[2]);
    const int nchunks = c_end - c_begin;

    auto load_desc = [&](int c) -> i32x2 {
        const int cc = (c < c_end ? c : c_begin) + vzero;               // clamp: a valid (ignored) entry
        return *at<i32x2>(chunk_tab, (uint32_t)cc * 8u);
    };
    auto decode = [&](i32x2 d, int c) -> SxChunk {
        const int w0 = __builtin_amdgcn_readfirstlane(d[0]), w1 = __builtin_amdgcn_readfirstlane(d[1]);
        return c < c_end ? SxChunk{w1 >> 8, w0, w1 & 127, (w1 >> 7) & 1} : SxChunk{0, 0, 0, 0};
    };

    // A chunk's plan words, lane = row: ONE vector load per array per chunk (a vector-memory instruction issued
    // beside the SIMD partner's MFMA stream costs ~250 cycles here, so six per-piece index loads per PREP were
    // most of it); the DMA pieces pick their rows' words out of these registers with lane shuffles.
    struct Words { int src; int key; };
    auto load_words = [&](const SxChunk& c) -> Words {
        const int rc = lane < c.rows ? lane : c.rows - 1;                   // rows >= 1 here; pad lanes repeat the last row
        const uint32_t eo = (uint32_t)(c.e0 + rc) * 4u;
        return Words{*at<int>(sorted_src, eo), *at<int>(sorted_key, eo)};
    };

    // ---- producers -----------------------------------------------------------------------------------------------
    // Gather one stage's A tile.  h_split holds every row of h already cut into its three bf16 pieces
    // ([N][3 planes][D] bf16, ghf_split3_rows or the previous layer's tail), so a producer only moves bytes:
    // 16-byte loads into registers (stage_load) and, one barrier interval later, 16-byte LDS writes (stage_commit).
    // (Cutting the rows here cost ~130 vector-ALU instructions per lane and stage; the SIMD's issue slots, shared
    // with the consumer wave's MFMA stream, were what bounded that kernel.)
    // LDS image of a plane: rows of 256 + 16 bytes, so that an MFMA fragment read (16 rows x one 16-byte granule)
    // covers all banks with plain immediate offsets.  Piece i of wave tw: plane i % 3 of rows 4*tw .. 4*tw+3 of row
    // tile i / 3, 16 lanes per row; the pieces of dead tiles are skipped.
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)h_split, 0, (int)(uint32_t)((uint64_t)N * HROW), 0x00020000);
    auto stage_load = [&](i32x4 (&stg)[IPW], const SxChunk& c, int ph, const Words& wd) {
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        const int mts = (c.rows + 15) >> 4;                                // live row tiles
        const int word = ph == 0 ? wd.src : wd.key;
        const uint32_t nbase = ph == 0 ? 0u : (uint32_t)node0 - kbase;
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            if (m >= mts) continue;
            const int v = __shfl(word, m * 16 + tw * RPW + lane / GPR, 64);
            uint32_t node = (ph == 0 ? (uint32_t)(v & SRC_MASK) : (uint32_t)v) + nbase;
            if (dbg & 1) node = (uint32_t)node0;
            const int off = (int)(node * (uint32_t)HROW) + ((lane % GPR) << 4);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) stg[m * 3 + pl] = __builtin_amdgcn_raw_buffer_load_b128(rsH, off + pl * (D * 2), 0, 0);
        }
    };
    auto stage_commit = [&](const i32x4 (&stg)[IPW], char* Abuf, int rows) {
        const int mts = (rows + 15) >> 4;
        char* dst = Abuf + (tw * RPW + lane / GPR) * ROWB + ((lane % GPR) << 4);
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            if (m >= mts) continue;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *(i32x4*)(dst + m * 16 * ROWB + pl * PLANE) = stg[m * 3 + pl];
        }
    };
    // a chunk's row words for the consumers' scatter: (byte offset of the row's target in the block sums) | run head,
    // and per row tile the mask of rows that continue a run of equal destinations
    auto publish_rows = [&](const SxChunk& c, const Words& wd, int* meta) {
        if (tw != 0) return;
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        const int head = (int)((uint32_t)wd.src >> SRC_BITS), row16 = lane & 15;
        const bool live = lane < c.rows;
        const int tgt = (live && head == row16) ? (int)((uint32_t)wd.key - kbase) : BN + ((lane >> 2) & 3);
        const unsigned long long runs = __ballot(live && head != row16);
        if (lane < CR) meta[lane] = (tgt * (D * 4)) | (live ? head : row16);  // D*4 = 512: the low 4 bits stay free
        if (lane < MTC) meta[CR + lane] = (int)((runs >> (16 * lane)) & 0xFFFFull);
    };

    // ---- consumers -----------------------------------------------------------------------------------------------
    // B fragments (GHF_WLAYOUT_SPLIT3, written by K1): Wsplit[r][o/16][kk/32][piece][lane = ((kk%32)/8)*16 + o%16][kk%8]
    // bf16, kk in [0, 2d).  Buffer loads: resource descriptor + scalar offset (relation, fragment, k-step) + lane*16 +
    // immediate (piece), so one VGPR addresses them all.
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wsplit, 0, (int)w_bytes, 0x00020000);
    auto b_soff = [&](int r, int ph, int t) -> int {
        if (dbg & 2) r = 0;
        return __builtin_amdgcn_readfirstlane((((r * NT + tw * NTW + t) * NKS + ph * KS) * 3) * 1024);
    };
    const int lane16 = lane * 16;
    // ring of BRING = BPRE + 1 k-steps of B pieces: k-step j of a stage sits in slot j % BRING.  A slot is refilled
    // (for k-step j + BPRE, or at the end of a stage for the next stage's head) only after the MFMAs of the k-step it
    // held have been issued.
    constexpr int BPRE = 2, BRING = BPRE + 1;
    i32x4 b[BRING][NTW][3];
    auto load_b_head = [&](int r, int ph) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int j = 0; j < BPRE; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (pl == 2 && (dbg & 32)) continue;
                    b[j % BRING][t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16 + pl * 1024, b_soff(r, ph, t) + j * 3072, 0);
                }
    };

    f32x4 acc[MTC][NTW];
    // One stage = one K-phase of the chunk = KS k-steps of 32.  Per (k-step, row tile): 3 A-piece fragments from LDS
    // (read one step ahead), and for each of the wave's column fragments the six piece products, smallest first.
    // B pieces of k-step j+BPRE are requested while k-step j computes.  One code path: dead row tiles (m >= mt) skip
    // their MFMAs (three tile-count variants of this loop cost 80 spilled registers).
    auto compute_stage = [&](int mt, int r, int ph, const char* Abuf) {
        int bs[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bs[t] = b_soff(r, ph, t);
        i32x4 a[2][3];
        auto lda = [&](int j, int m, i32x4 (&dst)[3]) {
            const char* src = Abuf + (m * 16 + c16) * ROWB + ((4 * j + q) << 4);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[pl] = *(const i32x4*)(src + pl * PLANE);
        };
        lda(0, 0, a[0]);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (j + BPRE < KS && !(dbg & 16)) {
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        if (pl == 2 && (dbg & 32)) continue;
                        b[(j + BPRE) % BRING][t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16 + pl * 1024, bs[t] + (j + BPRE) * 3072, 0);
                    }
            }
#pragma unroll
            for (int m = 0; m < MTC; ++m) {
                const int cur = (j * MTC + m) & 1;
                if (j * MTC + m + 1 < KS * MTC) lda((j * MTC + m + 1) / MTC, (j * MTC + m + 1) % MTC, a[cur ^ 1]);
                if (m < mt && !(dbg & 4)) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        auto fma = [&](int pa, int pb) {
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[cur][pa]),
                                                                                __builtin_bit_cast(bf16x8, b[j % BRING][t][pb]),
                                                                                acc[m][t], 0, 0, 0);
                        };
                        if (!(dbg & 32)) { fma(2, 0); fma(0, 2); fma(1, 1); }  // weight 2^-16
                        fma(1, 0); fma(0, 1);                               // weight 2^-8
                        fma(0, 0);                                          // weight 1
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Segment-sum the finished rows by destination into this wave's column strips.  live = row tiles of the chunk;
    // a tile without a run of equal destinations is its own segment sum and skips the MFMAs.  Then a plain LDS
    // read-add-write through inline asm (see message_mfma.hip), tile by tile: a run may continue into the next tile.
    // The block sums keep a wave's 32 columns INTERLEAVED (LDS position 32*tw + 2*c16 + t holds column
    // 32*tw + 16*t + c16; the tail undoes it), so a lane's two values are adjacent and move with one 64-bit access.
    auto scatter_chunk = [&](int live, const int* meta) {
        static_assert(NTW == 2, "two column fragments per wave");
        const unsigned strip = (unsigned)(size_t)(lptr_t)(acc_lds + tw * 16 * NTW + c16 * NTW);
        i32x4 mq[MTC];
#pragma unroll
        for (int m = 0; m < MTC; ++m) mq[m] = *(const i32x4*)(meta + m * 16 + 4 * q);
        const i32x4 runs = *(const i32x4*)(meta + CR);
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            if (m >= live) continue;
            f32x4 y[NTW];
            if (__builtin_amdgcn_readfirstlane(runs[m])) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) y[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float sel = ((mq[m][s] & 15) == c16) ? 1.0f : 0.0f;   // S[i = c16][k = 4q + s] = (head(k) == i)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        y[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, acc[m][t][s], y[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NTW; ++t) y[t] = acc[m][t];
            }
            unsigned addr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) addr[s] = strip + ((unsigned)mq[m][s] & ~15u);           // the run's target row, or a dummy
            f32x2 v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("ds_read_b64 %0, %1" : "=v"(v[s]) : "v"(addr[s]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x2 r = v[s] + (f32x2){y[0][s], y[1][s]};
                asm volatile("ds_write_b64 %0, %1" :: "v"(addr[s]), "v"(r) : "memory");
            }
        }
    };

    __syncthreads();                                   // sums zeroed, row words initialised
    PP_STAMP(-1);

    // Two programs, 2*nchunks + 1 barriers each.  Barrier interval s belongs to stage s:
    //   consumers:  MFMAs of stage s from A tile s & 1; after a chunk's phase 1, its bias and its scatter
    //   producers:  request the gather of stage s + 2 into registers, then cut the rows of stage s + 1 (requested one
    //               interval ago) into the other A tile: a gather has a whole interval to arrive
    if (producer) {
        i32x4 stgA[IPW], stgB[IPW];                    // phase-0 / phase-1 stages in flight
        SxChunk chI = decode(load_desc(c_begin), c_begin), chN = decode(load_desc(c_begin + 1), c_begin + 1);
        i32x2 dNN = load_desc(c_begin + 2);
        Words wdI{0, 0}, wdN{0, 0};
        if (chI.rows) {
            wdI = load_words(chI);
            if (chN.rows) wdN = load_words(chN);
            stage_load(stgA, chI, 0, wdI);
            stage_load(stgB, chI, 1, wdI);
            stage_commit(stgA, a_tile(0), chI.rows);
        }
        for (int k = 0; k < nchunks; ++k) {            // chI = chunk k, chN = chunk k + 1
            __syncthreads();                           // interval 2k
            PP_STAMP(0);
            if (chN.rows) stage_load(stgA, chN, 0, wdN);
            PP_STAMP(2);
            stage_commit(stgB, a_tile(1), chI.rows);
            publish_rows(chI, wdI, row_words(k));
            PP_STAMP(3);
            __syncthreads();                           // interval 2k + 1
            PP_STAMP(0);
            if (chN.rows) {
                stage_load(stgB, chN, 1, wdN);
                PP_STAMP(2);
                stage_commit(stgA, a_tile(0), chN.rows);
            }
            chI = chN;
            wdI = wdN;
            chN = decode(dNN, c_begin + k + 2);
            if (chN.rows) wdN = load_words(chN);
            dNN = load_desc(c_begin + k + 3);
            PP_STAMP(3);
        }
        __syncthreads();
    } else {
        SxChunk ch = decode(load_desc(c_begin), c_begin);
        i32x2 dn = load_desc(c_begin + 1);
        float bias_v[NTW] = {};
        if (ch.rows) load_b_head(ch.r, 0);
        for (int k = 0; k < nchunks; ++k) {
            const int mt = (ch.rows + 15) >> 4;
            __syncthreads();                           // interval 2k: phase 0
            PP_STAMP(0);
#pragma unroll
            for (int t = 0; t < NTW; ++t) bias_v[t] = *at<float>(bias, (uint32_t)(ch.r * D + (tw * NTW + t) * 16 + c16) * 4u);
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            compute_stage(mt, ch.r, 0, a_tile(0));
            load_b_head(ch.r, 1);
            PP_STAMP(1);
            __syncthreads();                           // interval 2k + 1: phase 1, then the chunk's rows join the sums
            PP_STAMP(0);
            compute_stage(mt, ch.r, 1, a_tile(1));
            const SxChunk nx = decode(dn, c_begin + k + 1);
            if (nx.rows) load_b_head(nx.r, 0);         // lands during the scatter
            dn = load_desc(c_begin + k + 2);
            PP_STAMP(1);
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[m][t] = acc[m][t] + bias_v[t];   // bias[r] once per edge row
            if (!(dbg & 8)) scatter_chunk(mt, row_words(k));
            ch = nx;
            PP_STAMP(4);
        }
        __syncthreads();
    }

    // ---- fused tail: one wave per destination row, RB rows in flight -----------------------------------------
    constexpr int CPL = D / 64;
    // LDS position lane*CPL + c of a row of the sums holds output column col[c] (see the scatter): for D = 128 the
    // lane's two positions are columns o and o + 16, for D = 64 position = column.
    int col[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) col[c] = NTW == 2 ? 32 * (lane >> 4) + 16 * c + (lane & 15) : lane * CPL + c;
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
    constexpr int RB = 4;
    for (int v0 = w; v0 < nrows; v0 += NWV * RB) {
        float x[RB][CPL], inv[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int v = v0 + rb * NWV;
            const int vc = v < nrows ? v : v0;
            const int64_t node = node0 + vc;
            const int deg = indeg[node];
            inv[rb] = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg > 1 ? deg : 1);
#pragma unroll
            for (int c = 0; c < CPL; ++c) x[rb][c] = no_tail ? 0.f : h[(size_t)node * D + col[c]];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int v = v0 + rb * NWV;
            const int vc = v < nrows ? v : v0;
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
            if (v < nrows) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) h_out[(size_t)(node0 + v) * D + col[c]] = x[rb][c];
                if (h_split_out) {                     // the same rows cut into bf16 pieces, for the next layer's gathers
                    uint16_t* __restrict__ sp = (uint16_t*)h_split_out + (size_t)(node0 + v) * (3 * D);
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        uint16_t pc[3];
                        split3_pieces(x[rb][c], pc);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) sp[pl * D + col[c]] = pc[pl];
                    }
                }
            }
        }
    }
#ifdef GHF_STAMPS
    PP_STAMP(6);                                        // drain + tail
    if (lane == 0 && blockIdx.x < 8192)
        for (int i = 0; i < 8; ++i) ghf_sx_stamp_buf[((size_t)blockIdx.x * 8 + w) * 8 + i] = st_acc[i];
#endif
}

template <int D>
static int launch_sx_for(const MsgArgs& a, hipStream_t stream) {
    using C = SxCfg<D>;
    constexpr int CR = 16 * C::MTC;
    constexpr size_t lds = (size_t)(C::BN + 4) * D * 4 + (size_t)2 * 3 * CR * (D * 2 + 16) + 2 * (CR + 16) * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    GHF_REQUIRE(a.block_nodes == C::BN, "message(sx): plan block_nodes=%d, kernel for d=%d needs %d", a.block_nodes, D, C::BN);
    GHF_REQUIRE(a.wlayout == GHF_WLAYOUT_SPLIT3, "message(sx): weights must be in SPLIT3 layout");
    GHF_REQUIRE(a.chunk_tab && a.item_tab && a.blk_item_off, "message(sx): the plan's chunk / item tables are missing");
    const int64_t row_end = a.row0 + a.rows;
    GHF_REQUIRE(row_end == a.N || row_end % C::BN == 0, "message(sx): row range must end on a block boundary or at N");
    if (a.rows <= 0) return GHF_OK;
    GHF_REQUIRE((uint64_t)a.N * D * 4 < (1ull << 32) && (uint64_t)a.E * 4 < (1ull << 32) && (uint64_t)a.R * 2 * D * D * 6 < (1ull << 32),
                "message(sx): 32-bit byte offsets need N*d*4, E*4 and R*2*d*d*6 below 4 GiB");
    static const int dbg = getenv("GHF_DEBUG_FLAGS") ? atoi(getenv("GHF_DEBUG_FLAGS")) : 0;   // honoured by -DGHF_ABLATE builds only
    GHF_SET_MAX_LDS(message_sx_kernel<D>, lds);
    GHF_REQUIRE(a.n_items >= cdiv(a.rows, C::BN), "message(sx): n_items=%lld is fewer than the blocks of the row range", (long long)a.n_items);
    GHF_REQUIRE(a.n_items == cdiv(a.rows, C::BN) || a.partial, "message(sx): split blocks need the `partial` scratch");
    GHF_REQUIRE(a.h_split, "message(sx): h_split is missing");
    GHF_REQUIRE((uint64_t)a.N * D * 6 < (1ull << 32), "message(sx): N*d*6 must stay below 4 GiB");
    message_sx_kernel<D><<<(unsigned)a.n_items, 512, lds, stream>>>(a.h, a.h_split, a.N, a.sorted_key, a.sorted_src, a.chunk_tab, a.item_tab,
                                                                   a.item0, a.partial, a.indeg, a.R, a.W_msg, a.bias, a.ln_gamma,
                                                                   a.ln_beta, a.ln_eps, a.row0, row_end, a.h_out, a.h_split_out,
                                                                   a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM),
                                                                   (uint32_t)((uint64_t)a.R * 2 * D * D * 6), dbg);
    GHF_LAUNCH_CHECK();
    if (a.n_items > cdiv(a.rows, C::BN)) return launch_combine_split(a, stream);     // some block of the range is split
    return GHF_OK;
}

// ghf_split3_rows: one thread per 4 consecutive elements of a row
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ h, int64_t row0, int64_t rows, int d,
                                                          char* __restrict__ out) {
    const int q4 = d >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * q4) return;
    const int64_t row = row0 + i / q4;
    const int k4 = (int)(i % q4);
    i32x2 pc[3];
    split3(*(const f32x4*)(h + row * d + 4 * k4), pc);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *(i32x2*)(out + row * (6 * (int64_t)d) + pl * 2 * d + k4 * 8) = pc[pl];
}

int launch_split3_rows(const float* h, int64_t N, int d, int64_t row0, int64_t rows, void* h_split, hipStream_t stream) {
    if (rows <= 0) return GHF_OK;
    const int64_t n = rows * (d >> 2);
    GHF_REQUIRE(cdiv(n, 256) < (1ll << 31), "split3_rows: too many rows per launch");
    split3_rows_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>(h, row0, rows, d, (char*)h_split);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

bool message_sx_config(int d, int* block_nodes, int* chunk_rows, int* split_chunks) {
    if (d != 128) return false;
    *block_nodes = SxCfg<128>::BN;
    *chunk_rows = 16 * SxCfg<128>::MTC;
    *split_chunks = 128;
    return true;
}

int launch_message_sx(const MsgArgs& a, hipStream_t stream) {
    if (a.d == 128) return launch_sx_for<128>(a, stream);
    return set_err(GHF_EUNSUPPORTED, "message(sx): no split-bf16 kernel for d=%d", a.d);
}

}  // namespace ghf

#ifdef GHF_STAMPS
extern "C" int ghf_debug_read_stamps_sx(unsigned long long* host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ghf::ghf_sx_stamp_buf), count * sizeof(unsigned long long));
}
#endif
