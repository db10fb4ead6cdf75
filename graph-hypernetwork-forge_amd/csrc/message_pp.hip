// message_pp.hip — K2+K3 in exact fp32 on the matrix cores, "ping-pong" schedule (hidden sizes 128 and 64).
//
// Replaces models/hypergnn.py:281-296 of the reference:
//   out_v = (1/max(indeg_v,1)) * sum_{e=(u->v)} ( h_u W_msg[r_e] + bias[r_e] + h_v W_self[r_e] ),  h'_v = LayerNorm(ReLU(out_v + h_v))
// with v_mfma_f32_16x16x4_f32 (an fp32 fma chain from 0): the d = 64 kernel, and at d = 128 the exact kernel the range
// guard (ghf.h: ghf_set_range_flag) and GHF_KERNEL=pp route to; the default d = 128 kernel is message_bx.hip.
//
// Geometry.  A workgroup owns BN consecutive destination nodes and keeps their fp32 sums [BN][D] in LDS for the whole
// kernel: no global atomics, the tail is fused, every h' row is written once.  plan.hip has sorted the block's in-edges by
// (relation, destination) and cut them into chunks of <= 48 rows of ONE relation; a chunk is a small GEMM
// [rows, 2D] x [2D, D] — A row = [h_src | h_dst], gathered into an LDS tile; B = [W_msg[r]; W_self[r]] in MFMA fragment
// order (GHF_WLAYOUT_FRAG16), streamed from L2 into registers — run as two K-phases of D.  A wave owns 32 output columns
// and is the only writer of its strip of the block sums: fixed summation order, bitwise reproducible.
// Adding a chunk's rows to the sums: rows are sorted by destination, so per 16-row tile one more MFMA product with the 0/1
// matrix S[i][k] = (head(k) == i) (head = first row of k's run of equal destinations, from the plan) moves each run's sum
// into its first row — exact — and the live rows, now of distinct destinations, are added by a plain LDS read-add-write
// (ds_add_f32 measured ~110 cycles per wave instruction: more than the contraction).
// LDS (D=128): sums (216+4)*512 B + 2 A tiles 48*512 B + row words = 162,176 B (1 workgroup/CU); (D=64): 81,280 B (2/CU).
//
// Schedule.  With all 8 waves on one program the two waves of a SIMD reach their non-MFMA segments (prefetch issue,
// waits, barrier, scatter) together and the matrix pipe idles there (stamped 66 % busy).  Here the waves form two TEAMS
// of 4 (one wave per SIMD each; waves w and w+4 share a SIMD); teams take alternate chunks and alternate ROLES every
// barrier interval:
//      interval g     team g&1      : MFMA   — one K-phase of its chunk, nothing but ds_read + v_mfma
//                     the other team: PREP   — scatter of its finished chunk, LDS-DMA gather of its next A tile,
//                                              B-fragment / index / descriptor loads, then s_waitcnt vmcnt(0)
// so each SIMD always has one wave feeding the matrix pipe while its partner absorbs every memory latency.
// Each team has its own A tile, B fragments are loaded in PREP into the registers the team's previous MFMA interval
// just released, and no global load is issued inside an MFMA interval.  Scatters of the two teams never overlap in time
// and a team scatters its chunks in order, so the sums stay bitwise reproducible.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int PP_WAIT_VMCNT0 = 0x0F70;      // s_waitcnt vmcnt(0) only (builtin form: modelled by hipcc)

// uniform base + 32-bit byte offset: lets the backend use the SGPR-base addressing form (one VGPR per address);
// with 64-bit per-lane pointers the loop-invariant parts hoisted out of the chunk loop spilled.  All arrays
// indexed this way are < 4 GiB here (checked by the launcher).
template <class T>
__device__ __forceinline__ const T* at(const void* base, uint32_t byte_off) {
    return (const T*)((const char*)base + byte_off);
}

// Diagnostic build only (-DGHF_STAMPS): per-wave s_memtime totals
#ifdef GHF_STAMPS
__device__ unsigned long long ghf_pp_stamp_buf[8192 * 8 * 8];
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

template <int D> struct PpCfg;
template <> struct PpCfg<128> { static constexpr int BN = 216, MTC = 3, WAVES_PER_SIMD = 2; };   // 162 KB LDS: 1 workgroup/CU
template <> struct PpCfg<64>  { static constexpr int BN = 216, MTC = 3, WAVES_PER_SIMD = 4; };   //  81 KB LDS: 2 workgroups/CU

struct PpChunk { int r; int e0; int rows; int cross; };     // rows == 0: none

template <int D>
__global__ __launch_bounds__(512, PpCfg<D>::WAVES_PER_SIMD) void message_pp_kernel(
    const float* __restrict__ h, int64_t N, const uint32_t* __restrict__ sorted_key,
    const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ chunk_tab,
    const int32_t* __restrict__ item_tab, int64_t item0, float* __restrict__ partial,
    const int32_t* __restrict__ indeg, int R,
    const float* __restrict__ Wfrag, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int64_t row0, int64_t row_end, float* __restrict__ h_out, int no_tail) {
    using C = PpCfg<D>;
    constexpr int BN = C::BN, MTC = C::MTC;
    constexpr int NWV = 8, TW = 4;            // waves per workgroup, per team
    constexpr int NJ = D / 16;                // k-groups of 16 per phase
    constexpr int NT = D / 16;                // 16-column fragments of the output
    constexpr int NTW = NT / TW;              // fragments per wave (2)
    constexpr int NJ2 = 2 * NJ;
    constexpr int CPR = D / 4;                // 16-byte chunks per A row
    constexpr int RPI = 256 / D;              // A rows per 1 KiB LDS-DMA wave-instruction
    constexpr int CR = 16 * MTC;              // rows per chunk
    constexpr int IPW = CR / RPI / TW;        // LDS-DMA instructions per wave per stage (6)
    static_assert(NTW * TW == NT && (NTW == 1 || NTW == 2) && CR % (RPI * TW) == 0 && MTC == 3, "bad tile config");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = (float*)smem;                    // [BN + 4][D]: block sums + 4 dummy rows
    float* Abase = acc_lds + (BN + 4) * D;            // [2 teams][CR][D], 16-byte chunks XOR-swizzled by (row & 15)
    int* s_meta = (int*)(Abase + 2 * CR * D);         // [2 teams][CR] row words: (target row << 4) | run head

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
    float* const Abuf = Abase + team * CR * D;        // this team's A tile
    int* const meta = s_meta + team * CR;             // this team's row words

    for (int i = tid; i < (BN + 4) * D / 4; i += NWV * 64) ((f32x4*)acc_lds)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 2 * CR; i += NWV * 64) s_meta[i] = ((BN + ((i >> 2) & 3)) * (D * 4)) | (i & 15);

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
    auto decode = [&](i32x2 d, int c) -> PpChunk {
        const int w0 = __builtin_amdgcn_readfirstlane(d[0]), w1 = __builtin_amdgcn_readfirstlane(d[1]);
        return c < c_end ? PpChunk{w1 >> 8, w0, w1 & 127, (w1 >> 7) & 1} : PpChunk{0, 0, 0, 0};
    };

    // A chunk's plan words, lane = row: ONE vector load per array per chunk (a vector-memory instruction issued
    // beside the SIMD partner's MFMA stream costs ~250 cycles here, so six per-piece index loads per PREP were
    // most of it); the DMA pieces pick their rows' words out of these registers with lane shuffles.
    struct Words { int src; int key; };
    auto load_words = [&](const PpChunk& c) -> Words {
        const int rc = lane < c.rows ? lane : c.rows - 1;                   // rows >= 1 here; pad lanes repeat the last row
        const uint32_t eo = (uint32_t)(c.e0 + rc) * 4u;
        return Words{*at<int>(sorted_src, eo), *at<int>(sorted_key, eo)};
    };

    // Gather the (chunk, phase) A tile into this team's buffer, register-staged: the 16-byte loads are issued early
    // in PREP (stage_load) and written to LDS at its end (stage_commit), after the scatter has covered their latency.
    // (LDS-DMA, measured in a schedule where no MFMA runs during the issue, stamped ~450 cycles per instruction to
    // issue from a wave whose SIMD partner streams MFMAs: 2,700 of a 6,700-cycle PREP.)  The LDS image is the same:
    // 16-byte chunks XOR-swizzled by (row & 15) via the SOURCE address, rows linear.  Phase 1 also publishes the
    // chunk's row words: (byte offset of the row's target in the block sums) | run head.  Every instruction here
    // competes with the partner's MFMA stream for issue, so the code is branch-free: all CR rows are gathered (pad
    // rows repeat the last live row: an L2 hit), shuffles first, one wait, then the loads.
    f32x4 stg[IPW];
    auto stage_load = [&](const PpChunk& c, int ph, const Words& wd) {
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        if (ph == 1 && tw == 0 && lane < CR) {
            const int head = (int)((uint32_t)wd.src >> SRC_BITS), row16 = lane & 15;
            const bool live = lane < c.rows;
            const int tgt = (live && head == row16) ? (int)((uint32_t)wd.key - kbase) : BN + ((lane >> 2) & 3);
            meta[lane] = (tgt * (D * 4)) | (live ? head : row16);          // D*4 = 512: the low 4 bits stay free
        }
        const int word = ph == 0 ? wd.src : wd.key;
        int v[IPW];
#pragma unroll
        for (int i = 0; i < IPW; ++i) v[i] = __shfl(word, (tw * IPW + i) * RPI + lane / CPR, 64);
        PP_STAMP(5);                                    // prep: row words + shuffles
        const uint32_t nbase = ph == 0 ? 0u : (uint32_t)node0 - kbase;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int rho = (tw * IPW + i) * RPI + lane / CPR;             // LDS row this lane fills
            const uint32_t node = (ph == 0 ? (uint32_t)(v[i] & SRC_MASK) : (uint32_t)v[i]) + nbase;
            stg[i] = *at<f32x4>(h, node * (uint32_t)(D * 4) + (uint32_t)(((lane % CPR) ^ (rho & 15)) << 4));
        }
        PP_STAMP(7);                                    // prep: gather issue
    };
    auto stage_commit = [&]() {
#pragma unroll
        for (int i = 0; i < IPW; ++i) *(f32x4*)(Abuf + (tw * IPW + i) * 256 + lane * 4) = stg[i];
    };

    // B fragments of (relation r, phase ph): byte offset of k-group 0 of this wave's first fragment
    auto b_off = [&](int r, int ph, int t) -> uint32_t {
        return (uint32_t)((r * NT + tw * NTW + t) * NJ2 + ph * NJ) * 1024u + (uint32_t)lane * 16u;
    };
    constexpr int BPRE = 2;                            // k-groups of B requested ahead (end of my previous interval); the rest just in time
    auto load_b_head = [&](int r, int ph, f32x4 (&b)[NJ][NTW]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int j = 0; j < BPRE; ++j) b[j][t] = *at<f32x4>(Wfrag, b_off(r, ph, t) + (uint32_t)j * 1024u);
    };

    f32x4 acc[MTC][NTW];
    f32x4 b[NJ][NTW];

    // MFMA interval: one K-phase of the chunk; M = live row tiles (compile-time per variant).
    //  - A fragments of k-group j+1 are read while the MFMAs of k-group j run (explicit two-stage pipeline).
    //  - B fragments: k-groups 0..BPRE-1 arrive from PREP; k-group j+BPRE is requested while k-group j computes.
    //    A vector-memory instruction issued by the SIMD partner beside this MFMA stream cost it 100-185 cycles
    //    each (stamped: 7,150 cycles of PREP issue per interval with all 16 B loads there); issued from inside the
    //    stream it costs ~60, and only BPRE+1 k-groups of B are live at a time (24 VGPRs instead of 64).
    auto compute_tiles = [&](auto Mtag, int r, int ph) {
        constexpr int M = decltype(Mtag)::value;
        uint32_t boff[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) boff[t] = b_off(r, ph, t);
        f32x4 a[2][M];
#pragma unroll
        for (int m = 0; m < M; ++m) a[0][m] = *(const f32x4*)(Abuf + (m * 16 + c16) * D + ((q ^ c16) << 2));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j + BPRE < NJ) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) b[j + BPRE][t] = *at<f32x4>(Wfrag, boff[t] + (uint32_t)(j + BPRE) * 1024u);
            }
            if (j + 1 < NJ) {
#pragma unroll
                for (int m = 0; m < M; ++m)
                    a[(j + 1) & 1][m] = *(const f32x4*)(Abuf + (m * 16 + c16) * D + (((4 * (j + 1) + q) ^ c16) << 2));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 1][m][s], b[j][t][s], acc[m][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // PREP: segment-sum finished rows by destination into this wave's column strips, tiles [M0, M1).
    // x = the rows (acc, or the copy kept for the deferred tile), mq = their row words.
    auto scatter_tiles = [&](auto M0tag, auto M1tag, f32x4 (&x)[MTC][NTW], const i32x4 (&mq)[MTC]) {
        constexpr int M0 = decltype(M0tag)::value, M1 = decltype(M1tag)::value;
        f32x4 y[MTC][NTW];
        const unsigned strip = (unsigned)(size_t)(lptr_t)(acc_lds + tw * 16 * NTW + c16 * NTW);   // NTW == 2: interleaved
#pragma unroll
        for (int m = M0; m < M1; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) y[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)                                          // independent chains, interleaved
#pragma unroll
            for (int m = M0; m < M1; ++m) {
                const float sel = ((mq[m][s] & 15) == c16) ? 1.0f : 0.0f;   // S[i = c16][k = 4q + s] = (head(k) == i)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    y[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, x[m][t][s], y[m][t], 0, 0, 0);
            }
        // plain LDS read-add-write through inline asm (see the header), tile by tile: a run of equal
        // destinations may continue into the next tile.  With two fragments per wave (D = 128) the block sums keep a
        // wave's 32 columns INTERLEAVED (LDS position 32*tw + 2*c16 + t holds column 32*tw + 16*t + c16; the tail
        // undoes it), so a lane's two values are adjacent and move with one 64-bit LDS access.
#pragma unroll
        for (int m = M0; m < M1; ++m) {
            unsigned addr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) addr[s] = strip + ((unsigned)mq[m][s] & ~15u);           // the run's target row, or a dummy
            if constexpr (NTW == 2) {
                f32x2 v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("ds_read_b64 %0, %1" : "=v"(v[s]) : "v"(addr[s]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f32x2 r = v[s] + (f32x2){y[m][0][s], y[m][1][s]};
                    asm volatile("ds_write_b64 %0, %1" :: "v"(addr[s]), "v"(r) : "memory");
                }
            } else {
                float v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("ds_read_b32 %0, %1" : "=v"(v[s]) : "v"(addr[s]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("ds_write_b32 %0, %1" :: "v"(addr[s]), "v"(v[s] + y[m][0][s]) : "memory");
            }
        }
    };
    auto load_row_words = [&](i32x4 (&mq)[MTC]) {
#pragma unroll
        for (int m = 0; m < MTC; ++m) mq[m] = *(const i32x4*)(meta + m * 16 + 4 * q);
    };

    // ---- team state --------------------------------------------------------------------------------------------
    int kc = c_begin + team;                           // index of my current chunk `ch`
    PpChunk ch = decode(load_desc(kc), kc);
    PpChunk ch_next{0, 0, 0, 0};
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
            i32x4 mq[MTC];
            load_row_words(mq);
            scatter_tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, MTC - 1>{}, acc, mq);
            mq2[MTC - 1] = mq[MTC - 1];
#pragma unroll
            for (int t = 0; t < NTW; ++t) x2[MTC - 1][t] = acc[MTC - 1][t];
            deferred = pend == MTC;                    // a dead last tile carries zeros into dummy rows: skip it
        }
        PP_STAMP(3);                                    // prep: scatter
        if (staged) stage_commit();
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(PP_WAIT_VMCNT0);     // everything landed before the barrier that hands it over
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
            scatter_tiles(std::integral_constant<int, MTC - 1>{}, std::integral_constant<int, MTC>{}, x2, mq2);
            deferred = 0;
        }
        PP_STAMP(3);
        if (staged) stage_commit();
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(PP_WAIT_VMCNT0);
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
        if (mt == 3) compute_tiles(std::integral_constant<int, 3>{}, ch.r, ph);
        else if (mt == 2) compute_tiles(std::integral_constant<int, 2>{}, ch.r, ph);
        else compute_tiles(std::integral_constant<int, 1>{}, ch.r, ph);
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
        scatter_tiles(std::integral_constant<int, MTC - 1>{}, std::integral_constant<int, MTC>{}, x2, mq2);
    __syncthreads();
    if (team == 1 && pending) {
        i32x4 mq[MTC];
        load_row_words(mq);
        scatter_tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, MTC>{}, acc, mq);
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
            }
        }
    }
#ifdef GHF_STAMPS
    PP_STAMP(6);                                        // drain + tail
    if (lane == 0 && blockIdx.x < 8192)
        for (int i = 0; i < 8; ++i) ghf_pp_stamp_buf[((size_t)blockIdx.x * 8 + w) * 8 + i] = st_acc[i];
#endif
}

template <int D>
static int launch_pp_for(const MsgArgs& a, hipStream_t stream) {
    using C = PpCfg<D>;
    constexpr int CR = 16 * C::MTC;
    constexpr size_t lds = (size_t)((C::BN + 4) * D + 2 * CR * D) * 4 + 2 * CR * 4;
    GHF_REQUIRE(a.block_nodes == C::BN, "message(pp): plan block_nodes=%d, kernel for d=%d needs %d", a.block_nodes, D, C::BN);
    GHF_REQUIRE(a.wlayout == GHF_WLAYOUT_FRAG16, "message(pp): weights must be in FRAG16 layout");
    GHF_REQUIRE(a.chunk_tab && a.item_tab && a.blk_item_off, "message(pp): the plan's chunk / item tables are missing");
    const int64_t row_end = a.row0 + a.rows;
    GHF_REQUIRE(row_end == a.N || row_end % C::BN == 0, "message(pp): row range must end on a block boundary or at N");
    if (a.rows <= 0) return GHF_OK;
    GHF_REQUIRE((uint64_t)a.N * D * 4 < (1ull << 32) && (uint64_t)a.E * 4 < (1ull << 32) && (uint64_t)a.R * 2 * D * D * 4 < (1ull << 32),
                "message(pp): 32-bit byte offsets need N*d*4, E*4 and R*2*d*d*4 below 4 GiB");
    GHF_SET_MAX_LDS(message_pp_kernel<D>, lds);
    GHF_REQUIRE(a.n_items >= cdiv(a.rows, C::BN), "message(pp): n_items=%lld is fewer than the blocks of the row range", (long long)a.n_items);
    GHF_REQUIRE(a.n_items == cdiv(a.rows, C::BN) || a.partial, "message(pp): split blocks need the `partial` scratch");
    message_pp_kernel<D><<<(unsigned)a.n_items, 512, lds, stream>>>(a.h, a.N, a.sorted_key, a.sorted_src, a.chunk_tab, a.item_tab,
                                                                   a.item0, a.partial, a.indeg, a.R, a.W_msg, a.bias, a.ln_gamma,
                                                                   a.ln_beta, a.ln_eps, a.row0, row_end, a.h_out,
                                                                   a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM));
    GHF_LAUNCH_CHECK();
    if (a.n_items > cdiv(a.rows, C::BN)) return launch_combine_split(a, stream);     // some block of the range is split
    return GHF_OK;
}

// split_chunks: a destination block with more chunks than this is cut into several work items (plan.hip).  A block of a
// uniform graph at the BASELINE configs has ~65 chunks, so only real hubs are split.
bool message_pp_config(int d, int* block_nodes, int* chunk_rows, int* split_chunks) {
    *split_chunks = 128;
    switch (d) {
        case 128: *block_nodes = PpCfg<128>::BN; *chunk_rows = 16 * PpCfg<128>::MTC; return true;
        case 64:  *block_nodes = PpCfg<64>::BN;  *chunk_rows = 16 * PpCfg<64>::MTC;  return true;
        default:  return false;
    }
}

int launch_message_pp(const MsgArgs& a, hipStream_t stream) {
    if (a.d == 128) return launch_pp_for<128>(a, stream);
    if (a.d == 64) return launch_pp_for<64>(a, stream);
    return set_err(GHF_EUNSUPPORTED, "message(pp): no ping-pong kernel for d=%d", a.d);
}

}  // namespace ghf

#ifdef GHF_STAMPS
extern "C" int ghf_debug_read_stamps_pp(unsigned long long* host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ghf::ghf_pp_stamp_buf), count * sizeof(unsigned long long));
}
#endif
