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
    const float* __restrict__ h, int64_t N, const uint32_t* __restrict__ sorted_key,
    const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ chunk_tab,
    const int32_t* __restrict__ item_tab, int64_t item0, float* __restrict__ partial,
    const int32_t* __restrict__ indeg, int R,
    const void* __restrict__ Wsplit, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int64_t row0, int64_t row_end, float* __restrict__ h_out, int no_tail, uint32_t w_bytes, int dbg_arg) {
#ifdef GHF_ABLATE
    const int dbg = dbg_arg;      // 1: gather one hot row, 2: one relation's weights, 4: no main MFMAs, 8: no scatter, 16: no B loads in the stream
#else
    constexpr int dbg = 0;
#endif
    using C = SxCfg<D>;
    constexpr int BN = C::BN, MTC = C::MTC;
    constexpr int NWV = 8, TW = 4;            // waves per workgroup, per team
    constexpr int KS = D / 32;                // k-steps of 32 per phase (one bf16 MFMA deep)
    constexpr int NKS = 2 * KS;               // k-steps of the whole contraction [h_u | h_v]
    constexpr int NT = D / 16;                // 16-column fragments of the output
    constexpr int NTW = NT / TW;              // fragments per wave (2)
    constexpr int ROWB = D * 2 + 16;          // bytes per row of one bf16 plane of the A tile: padded by one granule, so
                                              // that 16 rows x one granule cover all banks with plain immediate offsets
    constexpr int PLANE = 16 * MTC * ROWB;    // bytes per plane
    constexpr int CPR = D / 4;                // 16-byte chunks per A row
    constexpr int RPI = 256 / D;              // A rows per 1 KiB LDS-DMA wave-instruction
    constexpr int CR = 16 * MTC;              // rows per chunk
    constexpr int IPW = CR / RPI / TW;        // LDS-DMA instructions per wave per stage (6)
    static_assert(NTW * TW == NT && (NTW == 1 || NTW == 2) && CR % (RPI * TW) == 0 && MTC == 3, "bad tile config");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = (float*)smem;                    // [BN + 4][D]: block sums + 4 dummy rows
    char* Abase = (char*)(acc_lds + (BN + 4) * D);    // [2 teams][3 planes][CR][D + 8] bf16
    constexpr int MSTR = CR + 16;                     // ints of row words per team: CR rows, then one flag per row tile
    int* s_meta = (int*)(Abase + 2 * 3 * PLANE);      // [2 teams][MSTR]: row words (target row << 4) | run head; tile flags: has a run > 1

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = w >> 2, tw = w & 3;              // waves w and w+4 share a SIMD: one of each team per SIMD
    const int q = lane >> 4, c16 = lane & 15;
    // work item: { block, first chunk, one past last chunk, scratch slot or -1 } (plan.hip); a heavy block (the hub
    // of a power-law graph) is several items, whose raw sums go to scratch slots and are combined by a second kernel
    const i32x4 item = *(const i32x4*)(item_tab + 4 * (size_t)(item0 + blockIdx.x));
    const int64_t blk = __builtin_amdgcn_readfirstlane(item[0]);
    const int slot = __builtin_amdgcn_readfirstlane(item[3]);
    const int64_t node0 = blk * BN;
    const int nrows = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const uint32_t seg0 = (uint32_t)(blk * R);
    char* const Abuf = Abase + team * 3 * PLANE;      // this team's A tile
    int* const meta = s_meta + team * MSTR;           // this team's row words

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

    // Gather the (chunk, phase) A tile, register-staged as in message_pp.hip: the 16-byte fp32 loads are issued early
    // in PREP (stage_load); at its end (stage_commit) every loaded f32x4 is cut into its three bf16 pieces and
    // written, 8 bytes per plane, to this team's tile.  LDS image of a plane: rows linear (256 bytes), the sixteen
    // 16-byte granules of a row XOR-swizzled by (row & 15) so that the MFMA fragment reads (16 rows x one granule)
    // spread over all banks.  Phase 1 also publishes the chunk's row words: (byte offset of the row's target in the
    // block sums) | run head.  Branch-free: all CR rows are gathered (pad rows repeat the last live row: an L2 hit).
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)h, 0, (int)(uint32_t)((uint64_t)N * D * 4), 0x00020000);
    f32x4 stg[IPW];
    auto stage_load = [&](const SxChunk& c, int ph, const Words& wd) {
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        if (ph == 1 && tw == 0) {
            const int head = (int)((uint32_t)wd.src >> SRC_BITS), row16 = lane & 15;
            const bool live = lane < c.rows;
            const int tgt = (live && head == row16) ? (int)((uint32_t)wd.key - kbase) : BN + ((lane >> 2) & 3);
            const unsigned long long runs = __ballot(live && head != row16);      // rows that continue a run
            if (lane < CR) meta[lane] = (tgt * (D * 4)) | (live ? head : row16);  // D*4 = 512: the low 4 bits stay free
            if (lane < MTC) meta[CR + lane] = (int)((runs >> (16 * lane)) & 0xFFFFull);
        }
        const int mts = (c.rows + 15) >> 4;                                // live row tiles: pieces of dead tiles are skipped
        const int word = ph == 0 ? wd.src : wd.key;
        int v[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i) v[i] = __shfl(word, (i * TW + tw) * RPI + lane / CPR, 64);
        PP_STAMP(5);                                    // prep: row words + shuffles
        const uint32_t nbase = ph == 0 ? 0u : (uint32_t)node0 - kbase;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            if (i / 2 >= mts) continue;                                    // piece i holds rows of tile i/2 only (see stage_commit)
            uint32_t node = (ph == 0 ? (uint32_t)(v[i] & SRC_MASK) : (uint32_t)v[i]) + nbase;
            if (dbg & 1) node = (uint32_t)node0;
            stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsH, (int)(node * (uint32_t)(D * 4)) + ((lane % CPR) << 4), 0, 0));
        }
        PP_STAMP(7);                                    // prep: gather issue
    };
    auto stage_commit = [&](int mts) {
        const int k4 = lane % CPR;                                         // which f32x4 of the row: elements 4*k4 .. +3
        static_assert(TW * RPI * 2 == 16 && IPW == 2 * MTC, "piece i of every wave must lie in row tile i / 2");
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            if (i / 2 >= mts) continue;
            const int rho = (i * TW + tw) * RPI + lane / CPR;              // tile row this lane fills: rows interleaved over the waves
            char* dst = Abuf + rho * ROWB + (k4 << 3);
            i32x2 pc[3];
            split3(stg[i], pc);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *(i32x2*)(dst + pl * PLANE) = pc[pl];
        }
    };

    // B fragments (GHF_WLAYOUT_SPLIT3, written by K1): Wsplit[r][o/16][kk/32][piece][lane = ((kk%32)/8)*16 + o%16][kk%8]
    // bf16, kk in [0, 2d).  Byte offset of piece 0 of k-step 0 of (relation r, phase ph) for this wave's fragment t:
    // Buffer loads: resource descriptor + scalar offset (relation, fragment, k-step) + lane*16 + immediate (piece), so
    // one VGPR addresses them all (with plain pointers hipcc kept a 64-bit VGPR address per fragment and k-step).
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wsplit, 0, (int)w_bytes, 0x00020000);
    auto b_soff = [&](int r, int ph, int t) -> int {
        if (dbg & 2) r = 0;
        return __builtin_amdgcn_readfirstlane((((r * NT + tw * NTW + t) * NKS + ph * KS) * 3) * 1024);
    };
    const int lane16 = lane * 16;
    constexpr int BRING = 3;                           // = BPRE + 1 (declared below)
    constexpr int BPRE = 2;                            // k-steps of B requested ahead (end of my previous interval); the rest just in time
    auto load_b_head = [&](int r, int ph, i32x4 (&b)[BRING][NTW][3]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int j = 0; j < BPRE; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[j % BRING][t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16 + pl * 1024, b_soff(r, ph, t) + j * 3072, 0);
    };

    f32x4 acc[MTC][NTW];
    // ring of BRING = BPRE + 1 k-steps of B pieces: k-step j of a phase sits in slot j % BRING.  A slot is refilled
    // (for k-step j + BPRE, or at the end of the interval for the next phase's head) only after the MFMAs of the
    // k-step it held have been issued.
    i32x4 b[BRING][NTW][3];

    // MFMA interval: one K-phase of the chunk = KS k-steps of 32; M = live row tiles (compile-time per variant).
    // Per (k-step, row tile): 3 A-piece fragments from LDS (read one step ahead), and for each of the wave's column
    // fragments the six piece products, smallest first.  B pieces of k-step j+BPRE are requested while k-step j
    // computes (message_pp.hip: a load issued from inside the MFMA stream is cheap, one issued by the SIMD partner
    // beside it is not).
    auto compute_tiles = [&](int mt, int r, int ph) {
        constexpr int M = MTC;                          // one code path; dead row tiles (m >= mt) skip their MFMAs
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
                    for (int pl = 0; pl < 3; ++pl)
                        b[(j + BPRE) % BRING][t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16 + pl * 1024, bs[t] + (j + BPRE) * 3072, 0);
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int cur = (j * M + m) & 1;
                if (j * M + m + 1 < KS * M) lda((j * M + m + 1) / M, (j * M + m + 1) % M, a[cur ^ 1]);
                if (m < mt && !(dbg & 4)) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    auto fma = [&](int pa, int pb) {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[cur][pa]),
                                                                            __builtin_bit_cast(bf16x8, b[j % BRING][t][pb]),
                                                                            acc[m][t], 0, 0, 0);
                    };
                    fma(2, 0); fma(0, 2); fma(1, 1);                        // weight 2^-16
                    fma(1, 0); fma(0, 1);                                   // weight 2^-8
                    fma(0, 0);                                              // weight 1
                }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // PREP: segment-sum finished rows by destination into this wave's column strips, tiles [M0, M1).
    // x = the rows (acc, or the copy kept for the deferred tile), mq = their row words.
    // live = row tiles of the chunk; runs = per-tile masks of rows that continue a run of equal destinations: a tile
    // without any is its own segment sum, and skips the MFMAs.
    auto scatter_tiles = [&](auto M0tag, auto M1tag, f32x4 (&x)[MTC][NTW], const i32x4 (&mq)[MTC], int live, const i32x4& runs) {
        constexpr int M0 = decltype(M0tag)::value, M1 = decltype(M1tag)::value;
        const unsigned strip = (unsigned)(size_t)(lptr_t)(acc_lds + tw * 16 * NTW + c16 * NTW);   // NTW == 2: interleaved
        // plain LDS read-add-write through inline asm (see message_mfma.hip), tile by tile: a run of equal
        // destinations may continue into the next tile.  With two fragments per wave (D = 128) the block sums keep a
        // wave's 32 columns INTERLEAVED (LDS position 32*tw + 2*c16 + t holds column 32*tw + 16*t + c16; the tail
        // undoes it), so a lane's two values are adjacent and move with one 64-bit LDS access.
#pragma unroll
        for (int m = M0; m < M1; ++m) {
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
                        y[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, x[m][t][s], y[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NTW; ++t) y[t] = x[m][t];
            }
            unsigned addr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) addr[s] = strip + ((unsigned)mq[m][s] & ~15u);           // the run's target row, or a dummy
            static_assert(NTW == 2, "two column fragments per wave");
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
    auto load_row_words = [&](i32x4 (&mq)[MTC], i32x4& runs) {
#pragma unroll
        for (int m = 0; m < MTC; ++m) mq[m] = *(const i32x4*)(meta + m * 16 + 4 * q);
        runs = *(const i32x4*)(meta + CR);
    };

    // ---- team state --------------------------------------------------------------------------------------------
    int kc = c_begin + team;                           // index of my current chunk `ch`
    SxChunk ch = decode(load_desc(kc), kc);
    SxChunk ch_next{0, 0, 0, 0};
    i32x2 d_next = load_desc(kc + 2);
    Words wd{0, 0}, wd_next{0, 0};                     // plan words of `ch` / of my next chunk
    float bias_v[NTW] = {};
    int pending = 0;                                   // live row tiles of my finished, not yet scattered chunk

    // The scatter of a finished chunk is split over my next two PREPs so that neither exceeds the partner's MFMA
    // interval (a PREP instruction gets about one issue slot per partner MFMA): tiles 0..1 in the phase-0 PREP,
    // tile 2 — rows and row words copied to registers there — in the phase-1 PREP.
    f32x4 x2[MTC][NTW];                                // only [MTC-1] is used: the deferred tile's rows
    i32x4 mq2[MTC];                                    // only [MTC-1] is used: its row words
    int deferred = 0;
    int runs2 = 0;                                     // the deferred tile's run mask

    // PREP before a phase-0 MFMA interval: move to my next chunk and request what its phase 0 needs FIRST, then
    // scatter (part of) the chunk that just finished while those loads are in flight.
    auto prep_ph0 = [&]() {
        __builtin_amdgcn_s_setprio(3);                  // PREP is short, latency-critical work beside the partner's MFMA stream
        const int pend = pending;
        if (pend) {
            pending = 0;
            ch = ch_next;                              // decoded, and its words loaded, in my previous PREP
            wd = wd_next;
            kc += 2;
        }
        const bool staged = ch.rows != 0;
        if (staged) {
            stage_load(ch, 0, wd);
#pragma unroll
            for (int t = 0; t < NTW; ++t) bias_v[t] = *at<float>(bias, (uint32_t)(ch.r * D + (tw * NTW + t) * 16 + c16) * 4u);
            d_next = load_desc(kc + 2);
        }
        asm volatile("" ::: "memory");
        PP_STAMP(2);                                    // prep: issue
        if (pend) {                                    // consumes registers and LDS only: nothing just requested
            i32x4 mq[MTC], runs;
            load_row_words(mq, runs);
            if (!(dbg & 8)) scatter_tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, MTC - 1>{}, acc, mq, pend, runs);
            mq2[MTC - 1] = mq[MTC - 1];
            runs2 = runs[MTC - 1];
#pragma unroll
            for (int t = 0; t < NTW; ++t) x2[MTC - 1][t] = acc[MTC - 1][t];
            deferred = pend == MTC;                    // a dead last tile carries zeros into dummy rows: skip it
        }
        PP_STAMP(3);                                    // prep: scatter
        if (staged) stage_commit((ch.rows + 15) >> 4);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(SX_WAIT_VMCNT0);     // everything landed before the barrier that hands it over
        PP_STAMP(4);                                    // prep: wait for memory + LDS commit
    };
    // PREP before a phase-1 MFMA interval
    auto prep_ph1 = [&]() {
        __builtin_amdgcn_s_setprio(3);
        const bool staged = ch.rows != 0;
        if (staged) {
            stage_load(ch, 1, wd);
            ch_next = decode(d_next, kc + 2);          // loaded one PREP ago
            if (ch_next.rows) wd_next = load_words(ch_next);
        }
        PP_STAMP(2);
        if (deferred) {
            if (!(dbg & 8)) scatter_tiles(std::integral_constant<int, MTC - 1>{}, std::integral_constant<int, MTC>{}, x2, mq2, MTC, (i32x4){0, 0, runs2, 0});
            deferred = 0;
        }
        PP_STAMP(3);
        if (staged) stage_commit((ch.rows + 15) >> 4);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(SX_WAIT_VMCNT0);
        PP_STAMP(4);
    };
    auto mfma_phase = [&](int ph) {
        PP_STAMP(0);                                    // barrier wait
        if (!ch.rows) return;
        const int mt = (ch.rows + 15) >> 4;
        if (ph == 0) {
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        compute_tiles(mt, ch.r, ph);
        if (ph == 1) {                                 // finish the rows: bias[r] once per edge row; dead tiles -> zeros
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[m][t] = m < mt ? acc[m][t] + bias_v[t] : (f32x4){0.f, 0.f, 0.f, 0.f};
            pending = mt;
        }
        // the first B fragments of my NEXT step, requested here at the end of my interval (my SIMD partner is in
        // PREP, no MFMA stream to compete with); they land during my own PREP, which ends with vmcnt(0)
        if (ph == 0) load_b_head(ch.r, 1, b);
        else if (ch_next.rows) load_b_head(ch_next.r, 0, b);
        PP_STAMP(1);                                    // mfma interval
    };

    if (ch.rows) {
        wd = load_words(ch);
        load_b_head(ch.r, 0, b);
    }
    __syncthreads();                                   // sums zeroed, row words initialised
    PP_STAMP(-1);

    // Two static programs, one per team, offset by one barrier interval; both execute 4*iters + 1 barriers.
    //   interval:   4i        4i+1      4i+2      4i+3
    //   team 0:     MFMA ph0  PREP ph1  MFMA ph1  PREP ph0 (scatter + next chunk)
    //   team 1:     PREP ph0  MFMA ph0  PREP ph1  MFMA ph1
    const int iters = (nchunks + 1) >> 1;              // team 0 never has fewer chunks than team 1
    if (team == 0) {
        prep_ph0();
        for (int it = 0; it < iters; ++it) {
            __syncthreads();  mfma_phase(0);
            __syncthreads();  PP_STAMP(0); prep_ph1();
            __syncthreads();  mfma_phase(1);
            __syncthreads();  PP_STAMP(0); prep_ph0();
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            __syncthreads();  PP_STAMP(0); prep_ph0();
            __syncthreads();  mfma_phase(0);
            __syncthreads();  PP_STAMP(0); prep_ph1();
            __syncthreads();  mfma_phase(1);
        }
    }
    // drain, one team per interval (their read-add-writes must not overlap): team 0's deferred tile, then team 1's
    // last chunk
    __syncthreads();
    if (team == 0 && deferred)
        scatter_tiles(std::integral_constant<int, MTC - 1>{}, std::integral_constant<int, MTC>{}, x2, mq2, MTC, (i32x4){0, 0, runs2, 0});
    __syncthreads();
    if (team == 1 && pending) {
        i32x4 mq[MTC], runs;
        load_row_words(mq, runs);
        scatter_tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, MTC>{}, acc, mq, pending, runs);
    }
    __syncthreads();

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
            inv[rb] = 1.0f / (float)(deg > 1 ? deg : 1);
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
    GHF_HIP_CHECK(hipFuncSetAttribute((const void*)message_sx_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GHF_REQUIRE(a.n_items >= cdiv(a.rows, C::BN), "message(sx): n_items=%lld is fewer than the blocks of the row range", (long long)a.n_items);
    GHF_REQUIRE(a.n_items == cdiv(a.rows, C::BN) || a.partial, "message(sx): split blocks need the `partial` scratch");
    message_sx_kernel<D><<<(unsigned)a.n_items, 512, lds, stream>>>(a.h, a.N, a.sorted_key, a.sorted_src, a.chunk_tab, a.item_tab,
                                                                   a.item0, a.partial, a.indeg, a.R, a.W_msg, a.bias, a.ln_gamma,
                                                                   a.ln_beta, a.ln_eps, a.row0, row_end, a.h_out,
                                                                   (a.flags & GHF_FLAG_NO_TAIL) ? 1 : 0,
                                                                   (uint32_t)((uint64_t)a.R * 2 * D * D * 6), dbg);
    GHF_LAUNCH_CHECK();
    if (a.n_items > cdiv(a.rows, C::BN)) return launch_combine_split(a, stream);     // some block of the range is split
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
