// message_mfma.hip — K2+K3 on the matrix cores (gfx950), hidden sizes 64 and 128.
//
// Replaces models/hypergnn.py:281-296 of the reference:
//   out_v = (1/max(indeg_v,1)) * sum_{e=(u->v)} ( h_u W_msg[r_e] + bias[r_e] + h_v W_self[r_e] )
//   h'_v  = LayerNorm(ReLU(out_v + h_v))
//
// Geometry.  One workgroup (NW waves) owns BN consecutive destination nodes and keeps
// their fp32 sums [BN][D] in LDS for the whole kernel: no global atomics, the tail is
// fused, and every h' row is written exactly once.  The plan (plan.hip) has sorted the
// block's in-edges by (relation, destination) and cut them into "chunks": <= CR rows of
// ONE relation r, listed in a per-block chunk table.  A chunk is a small dense GEMM
// [rows, 2D] x [2D, D]  with
//   A row  = [h_src | h_dst]   (gathered into LDS by LDS-DMA, one 1 KiB piece per wave-instr)
//   B      = [W_msg[r]; W_self[r]]  (pre-arranged by K1 in MFMA fragment order, GHF_WLAYOUT_FRAG16)
// run as two K-phases of D (phase 0: h_src x W_msg, phase 1: h_dst x W_self) so that the
// A tile of one phase is gathered while the other phase computes (1 barrier per phase).
// Waves split the OUTPUT COLUMNS (one 16-column strip each): a wave streams its own column
// slab of W[r] from L2 straight into registers (each fragment is reused by every row tile
// of the chunk), all waves share the A tile through LDS, and a wave is the only writer of
// its strip of the block sums — so the summation order is fixed and results are bitwise
// reproducible.
// Math: v_mfma_f32_16x16x4_f32, an exact fp32 fma chain from 0; bias[r] is added to each row's result.
//
// Accumulating a chunk's rows into the block sums.  LDS float atomics (ds_add_f32) measured
// ~110 cycles per wave-instruction here and cost more than the whole contraction, so they are
// not used.  Rows of a chunk are sorted by destination, so equal destinations are adjacent:
// per 16-row tile one extra MFMA product Y = S.X with the 0/1 matrix S[i][k] = (head(k) == i),
// head(k) = the first row of k's run of equal destinations (from the plan), moves each run's sum
// into its first row (exact: an fma chain with 1.0 / 0.0); the tile's live rows then have DISTINCT
// destinations and are added with a plain, conflict-free LDS read-add-write (the other rows go
// to per-lane-group dummy rows).
//
// LDS (D=128): sums (216+4)*512 B + 2 A tiles 48*512 B + 2*48 row words = 162,176 B (1 workgroup/CU);
//     (D=64) : (216+4)*256 + 2*48*256 + 384 = 81,280 B (2 workgroups of 4 waves per CU).
#include "common.h"

#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int D> struct MfmaCfg;
// NW waves per workgroup = D/16 column strips; BN destination nodes per block; MTC row tiles per chunk
template <> struct MfmaCfg<128> { static constexpr int BN = 216, MTC = 3, NW = 8; };
template <> struct MfmaCfg<64>  { static constexpr int BN = 216, MTC = 3, NW = 4; };

struct Chunk { int r; int e0; int rows; int cross; };     // rows == 0: no chunk

// s_waitcnt vmcnt(0) with expcnt/lgkmcnt left at their maxima (gfx9 simm16: vm[3:0], exp[6:4], lgkm[11:8], vm[15:14]).
// The BUILTIN form is used on purpose: hipcc models it, so after it the compiler knows no load is pending; an
// inline-asm wait is invisible to it, and it then re-waits (vmcnt(0), mid-phase) before reusing "pending" registers.
constexpr int WAIT_VMCNT0 = 0x0F70;

// Diagnostic build only (-DGHF_STAMPS, libghf_hip_stamps.so): per-wave s_memtime totals of the loop
// segments, written to a buffer of their own that nothing else reads (tools/stamps.py prints shares).
#ifdef GHF_STAMPS
constexpr int STAMP_SLOTS = 8;
__device__ unsigned long long ghf_stamp_buf[8192 * 8 * STAMP_SLOTS];
#define GHF_STAMP_DECL unsigned long long st_acc[STAMP_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0
#define GHF_STAMP(i)                                                                           \
    do {                                                                                       \
        unsigned long long _t;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((i) >= 0) st_acc[(i) < 0 ? 0 : (i)] += _t - st_last;                               \
        st_last = _t;                                                                          \
    } while (0)
#else
#define GHF_STAMP_DECL
#define GHF_STAMP(i)
#endif

template <int D>
__global__ __launch_bounds__(MfmaCfg<D>::NW * 64) void message_mfma_kernel(
    const float* __restrict__ h, int64_t N, const uint32_t* __restrict__ sorted_key,
    const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ chunk_tab,
    const int32_t* __restrict__ blk_chunk_off, const int32_t* __restrict__ indeg, int R,
    const float* __restrict__ Wfrag, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int64_t row0, int64_t row_end, float* __restrict__ h_out, int no_tail, int dbg) {
    using C = MfmaCfg<D>;
    constexpr int BN = C::BN, MTC = C::MTC, NW = C::NW;
    constexpr int NJ = D / 16;            // k-groups of 16 per phase
    constexpr int NT = D / 16;            // 16-column tiles of the output (= NW: one per wave)
    constexpr int NJ2 = 2 * NJ;
    constexpr int CPR = D / 4;            // 16-byte chunks per A row
    constexpr int RPI = 256 / D;          // A rows per 1 KiB LDS-DMA wave-instruction
    constexpr int CR = 16 * MTC;          // rows per chunk
    constexpr int IPW = CR / RPI / NW;    // LDS-DMA instructions per wave per stage
    static_assert(NT == NW && CR % (RPI * NW) == 0 && MTC == 3, "bad tile config");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = (float*)smem;                    // [BN + 4][D]: block sums + 4 dummy rows
    float* A0 = acc_lds + (BN + 4) * D;               // [CR][D], 16-byte chunks XOR-swizzled by (row & 15)
    float* A1 = A0 + CR * D;
    int* s_meta = (int*)(A1 + CR * D);                // [2][CR] per row: (target row of the sums << 4) | run head; by chunk parity

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = output column strip [16w, 16w+16)
    const int q = lane >> 4, c16 = lane & 15;
    const int64_t blk = row0 / BN + blockIdx.x;
    const int64_t node0 = blk * BN;
    const int nrows = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const uint32_t seg0 = (uint32_t)(blk * R);

    for (int i = tid; i < (BN + 4) * D / 4; i += NW * 64) ((f32x4*)acc_lds)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Chunk descriptors stream from the plan's table with VECTOR loads issued two chunks ahead and decoded one
    // barrier later.  (Scalar loads inside the loop were poison: SMEM returns out of order, so while one is
    // outstanding every lgkmcnt wait of the LDS fragment reads becomes a full drain that also waits for it.)
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));    // opaque 0: keeps the table loads on the vector path
    const int c_begin = __builtin_amdgcn_readfirstlane(blk_chunk_off[blk]);
    const int c_end = __builtin_amdgcn_readfirstlane(blk_chunk_off[blk + 1]);
    auto load_desc = [&](int c) -> i32x2 {
        const int cc = (c < c_end ? c : c_begin) + vzero;               // clamp: a valid (ignored) entry
        return *(const i32x2*)(chunk_tab + 2 * (size_t)cc);
    };
    auto decode = [&](i32x2 d, int c) -> Chunk {
        const int w0 = __builtin_amdgcn_readfirstlane(d[0]), w1 = __builtin_amdgcn_readfirstlane(d[1]);
        return c < c_end ? Chunk{w1 >> 8, w0, w1 & 127, (w1 >> 7) & 1} : Chunk{0, 0, 0, 0};
    };

    // Raw plan words of the rows this lane's LDS-DMA pieces gather for (chunk, phase): sorted_src (source id +
    // run head) for phase 0, the sort key for phase 1.  They are decoded only in issue_stage, one barrier
    // later, so that no arithmetic on a just-loaded value forces a wait for memory here.
    auto load_idx = [&](const Chunk& c, int ph, int (&idx)[IPW]) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            int rho = (w * IPW + i) * RPI + lane / CPR;
            rho = rho < c.rows ? rho : c.rows - 1;
            const int e = c.e0 + rho;
            idx[i] = ph == 0 ? sorted_src[e] : (int)sorted_key[e];
        }
    };

    // phase 0: gather h[src] rows into Abuf.  phase 1: gather h[dst] rows and publish the chunk's row words
    // (srcw = the phase-0 words of the same rows: they carry the run heads).
    auto issue_stage = [&](const Chunk& c, int ph, float* Abuf, const int (&idx)[IPW], const int (&srcw)[IPW], int* meta) {
        const int live_rows = (c.rows + 15) & ~15;
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
        // the row words first, while no LDS-DMA of this stage is in flight (an LDS store behind a pending DMA
        // makes hipcc drain vmcnt(0) in front of it).  All CR rows: the scatter always walks MTC tiles.
        if (ph == 1) {
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int rho = (w * IPW + i) * RPI + lane / CPR;
                if ((lane % CPR) == 0) {
                    const int head = (int)((uint32_t)srcw[i] >> SRC_BITS), row16 = rho & 15;
                    const int dummy = BN + ((rho >> 2) & 3);
                    const bool live = rho < c.rows;
                    const int tgt = (live && head == row16) ? (int)((uint32_t)idx[i] - kbase) : dummy;
                    meta[rho] = (tgt << 4) | (live ? head : row16);            // pad rows: their own (dead) run
                }
            }
        }
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int piece = w * IPW + i;                 // wave-uniform
            if (piece * RPI < live_rows) {
                const int rho = piece * RPI + lane / CPR;  // LDS row this lane writes
                const int p = lane % CPR;                  // LDS 16-byte slot within the row
                int64_t node = ph == 0 ? (int64_t)(idx[i] & SRC_MASK) : node0 + (int)((uint32_t)idx[i] - kbase);
                if (dbg & 1) node = node0;                 // diagnostic: no random gather
                const float* src = h + (size_t)node * D + ((p ^ (rho & 15)) << 2);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Abuf + piece * 256), 16, 0, 0);
            }
        }
    };

    auto b_base = [&](int r, int ph) -> const float* {
        const int rr = (dbg & 2) ? 0 : r;                   // diagnostic: one relation's weights only (L2-hot)
        return Wfrag + ((size_t)(rr * NT + w) * NJ2 + ph * NJ) * 256 + lane * 4;
    };
    auto load_b = [&](const float* base, f32x4 (&b)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *(const f32x4*)(base + j * 256);
    };

    f32x4 acc[MTC];

    // One K-phase of the chunk GEMM.  M = live row tiles: a compile-time count keeps each variant
    // straight-line, so the compiler runs the LDS fragment reads ahead of the MFMAs with counted lgkmcnt
    // waits.  The NEXT phase's B fragments (bn, from nbase) are requested one per k-group between the
    // MFMAs: issued as a burst ahead of the phase, the 8 waves' 64 KiB of fragment loads held the CU's
    // 64 B/clk vector-memory path for ~2,000 cycles before the first MFMA could issue (stamped: 23 % of
    // the loop).  bn is consumed only after the next barrier.
    auto compute_tiles = [&](auto Mtag, const float* Abuf, const f32x4 (&b)[NJ], const float* nbase, f32x4 (&bn)[NJ]) {
        constexpr int M = decltype(Mtag)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 a[M];
#pragma unroll
            for (int m = 0; m < M; ++m)                                   // row = 16m + c16, so row & 15 == c16
                a[m] = *(const f32x4*)(Abuf + (m * 16 + c16) * D + (((4 * j + q) ^ c16) << 2));
            if (2 * j < NJ) {                    // two per k-group over the first half of the phase: the last request
                bn[2 * j] = *(const f32x4*)(nbase + (2 * j) * 256);          // then has half a phase to land before the
                bn[2 * j + 1] = *(const f32x4*)(nbase + (2 * j + 1) * 256);  // phase-end wait
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < M; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[j][s], acc[m], 0, 0, 0);
        }
    };

    // Segment-sum a finished chunk's rows by destination and add them into this wave's strip of the block sums.
    // Always walks MTC tiles: dead tiles carry zeros and pad row words, so they only touch the dummy rows.
    f32x4 pend[MTC];                                   // the finished chunk's rows (this wave's 16 columns)
    auto scatter_pending = [&](const int* meta, int cross) {
        // The scatter's 12 MFMAs are a short dependent tail; beside the SIMD partner's back-to-back compute MFMAs
        // they otherwise lose the matrix-pipe arbitration (priority, then age).
        __builtin_amdgcn_s_setprio(3);
        f32x4 y[MTC];
        unsigned addr[MTC][4];
        const unsigned strip = (unsigned)(size_t)(lptr_t)(acc_lds + w * 16 + c16);
#pragma unroll
        for (int m = 0; m < MTC; ++m) {
            const i32x4 mq = *(const i32x4*)(meta + m * 16 + 4 * q);       // row words of rows 4q .. 4q+3
            y[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float sel = ((mq[s] & 15) == c16) ? 1.0f : 0.0f;      // S[i = c16][k = 4q + s] = (head(k) == i)
                y[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, pend[m][s], y[m], 0, 0, 0);
                addr[m][s] = strip + (unsigned)(mq[s] >> 4) * (D * 4);      // output row 4q + s: its run's target, or a dummy
            }
        }
        // The RMW goes through inline asm: for a compiler-visible LDS store hipcc first drains vmcnt(0)
        // (the next tile's LDS-DMA is in flight and it cannot prove the two regions apart), which would
        // put a wait for memory in the middle of the phase.  The DMA targets the A tiles, these stores the
        // block sums, so no ordering between them is needed; LDS executes one wave's DS ops in order.
        // cross (from the plan): a run of equal destinations spans a tile boundary, so two tiles hit one address:
        // then the tiles go one after the other.  Otherwise all live targets of the chunk are distinct and the
        // loads of every tile are in flight together, issued before the MFMAs retire.
        if (!cross) {
            float v[MTC][4];
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("ds_read_b32 %0, %1" : "=v"(v[m][s]) : "v"(addr[m][s]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[1][0]), "+v"(v[1][1]),
                           "+v"(v[1][2]), "+v"(v[1][3]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[2][3])
                         :: "memory");
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("ds_write_b32 %0, %1" :: "v"(addr[m][s]), "v"(v[m][s] + y[m][s]) : "memory");
        } else {
#pragma unroll
            for (int m = 0; m < MTC; ++m) {
                float v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("ds_read_b32 %0, %1" : "=v"(v[s]) : "v"(addr[m][s]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("ds_write_b32 %0, %1" :: "v"(addr[m][s]), "v"(v[s] + y[m][s]) : "memory");
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    int idx0[IPW], idx1[IPW];
    f32x4 b0[NJ], b1[NJ];

    int kc = c_begin;                                   // index of the current chunk
    Chunk cur = decode(load_desc(kc), kc);
    Chunk nxt = decode(load_desc(kc + 1), kc + 1);
    if (cur.rows) {
        load_idx(cur, 0, idx0);
        load_idx(cur, 1, idx1);
        issue_stage(cur, 0, A0, idx0, idx0, nullptr);
        load_b(b_base(cur.r, 0), b0);
    }

    // One chunk = two K-phases; its rows are scattered into the block sums during the NEXT chunk's phase 0.
    // Stagger: the two waves that share a SIMD (w and w + NW/2) run the same program in lockstep between the
    // barriers, so with the scatter at the end of the chunk both leave the SIMD's matrix pipe idle together.
    // Instead the finished rows stay in registers (pend) and the first half of the waves scatters them BEFORE its
    // phase-0 MFMAs of the next chunk, the second half AFTER them: a wave's scatter runs beside its partner's MFMAs.
    //
    // The body is instantiated per live tile count M and dispatched ONCE per chunk: with a dispatch per phase the
    // variants join between the phases and hipcc's wait-count pass, seeing B-fragment loads possibly pending from
    // a sibling variant, puts a vmcnt(0) at the head of the next variant (a full wait for memory inside the phase).
    // Every global load is consumed only after the NEXT barrier, whose vmcnt(0) retires it.
    GHF_STAMP_DECL;
    const bool early_half = w < NW / 2;
    int par = 0;                                       // parity of the current chunk: which row-word buffer it fills
    int pend_cross = 0;
#pragma unroll
    for (int m = 0; m < MTC; ++m) pend[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 2 * CR; i += NW * 64) s_meta[i] = ((BN + ((i >> 2) & 3)) << 4) | (i & 15);   // nothing pending yet
    auto chunk_body = [&](auto Mtag) {
        constexpr int M = decltype(Mtag)::value;
        // The chunk's opening wait + barrier sit at the head of EACH variant: the if/else chain below is
        // linearised by the compiler, so a variant's head is (falsely) reachable from its sibling's tail, where
        // loads are pending; with the wait here that path is harmless instead of costing a mid-phase vmcnt(0).
        // (The per-variant asm comment keeps SimplifyCFG from hoisting the identical waits back out.)
        asm volatile("; chunk body, %0 row tiles" ::"n"(M) : "memory");
        __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
        GHF_STAMP(0);                           // wait for memory
        __syncthreads();                        // A0 = stage(cur,0) landed; all waves are past the previous chunk
        GHF_STAMP(1);                           // barrier
        issue_stage(cur, 1, A1, idx1, idx0, s_meta + par * CR);
        if (nxt.rows) load_idx(nxt, 0, idx0);
        const float bias_v = bias[(size_t)cur.r * D + w * 16 + c16];       // used after the next barrier
        const i32x2 d_nn = load_desc(kc + 2);                               // decoded after the next barrier
        asm volatile("" ::: "memory");          // keep the prefetch issue above ahead of the compute phase
        GHF_STAMP(2);                           // prefetch issue
        if (early_half && !(dbg & 8)) scatter_pending(s_meta + (par ^ 1) * CR, pend_cross);
        GHF_STAMP(4);                           // scatter
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 4)) compute_tiles(Mtag, A0, b0, b_base(cur.r, 1), b1);
        GHF_STAMP(3);                           // compute
        if (!early_half && !(dbg & 8)) scatter_pending(s_meta + (par ^ 1) * CR, pend_cross);
        GHF_STAMP(4);

        __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
        GHF_STAMP(0);
        __syncthreads();                        // A1 = stage(cur,1) + row words landed; all waves are past phase 0
        GHF_STAMP(1);
        const Chunk nn = decode(d_nn, kc + 2);
        if (nxt.rows) {
            issue_stage(nxt, 0, A0, idx0, idx0, nullptr);
            load_idx(nxt, 1, idx1);
        }
        asm volatile("" ::: "memory");          // hipcc otherwise sinks this prefetch block below the compute phase
        GHF_STAMP(2);
        // no next chunk: the fragment prefetch reloads this chunk's (harmless)
        if (!(dbg & 4)) compute_tiles(Mtag, A1, b1, b_base(nxt.rows ? nxt.r : cur.r, 0), b0);
        GHF_STAMP(3);
#pragma unroll
        for (int m = 0; m < MTC; ++m)                                       // bias[r] once per edge row
            pend[m] = m < M ? acc[m] + bias_v : (f32x4){0.f, 0.f, 0.f, 0.f};
        pend_cross = cur.cross;
        cur = nxt;
        nxt = nn;
        ++kc;
        par ^= 1;
        GHF_STAMP(5);                           // bookkeeping
    };

    GHF_STAMP(-1);
    while (cur.rows) {
        const int mt_cur = (cur.rows + 15) >> 4;                            // workgroup-uniform
        if (mt_cur == 3) chunk_body(std::integral_constant<int, 3>{});
        else if (mt_cur == 2) chunk_body(std::integral_constant<int, 2>{});
        else chunk_body(std::integral_constant<int, 1>{});
    }
    __syncthreads();                                   // row words of the last chunk (and the initial fill) visible
    if (!(dbg & 8)) scatter_pending(s_meta + (par ^ 1) * CR, pend_cross);  // the last chunk's rows
    GHF_STAMP(4);
    __syncthreads();

    // ---- fused tail: one wave per destination row, RB rows in flight ---------------------
    constexpr int CPL = D / 64;              // columns per lane
    float g[CPL], bt[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        g[c] = no_tail ? 1.f : gamma[lane * CPL + c];
        bt[c] = no_tail ? 0.f : beta[lane * CPL + c];
    }
    constexpr int RB = 8;                    // rows in flight per wave: the tail is a latency chain per row otherwise
    for (int v0 = w; v0 < nrows; v0 += NW * RB) {
        float x[RB][CPL], inv[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int v = v0 + rb * NW;
            const int vc = v < nrows ? v : v0;                          // clamp: surplus rows recompute row v0, not stored
            const int64_t node = node0 + vc;
            const int deg = indeg[node];
            inv[rb] = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg > 1 ? deg : 1);
#pragma unroll
            for (int c = 0; c < CPL; ++c) x[rb][c] = no_tail ? 0.f : h[(size_t)node * D + lane * CPL + c];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int v = v0 + rb * NW;
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
                for (int c = 0; c < CPL; ++c) x[rb][c] = (x[rb][c] - mean) * rstd * g[c] + bt[c];
            }
            if (v < nrows) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) h_out[(size_t)(node0 + v) * D + lane * CPL + c] = x[rb][c];
            }
        }
    }
#ifdef GHF_STAMPS
    GHF_STAMP(6);                               // tail
    if (lane == 0 && blockIdx.x < 8192)
        for (int i = 0; i < STAMP_SLOTS; ++i) ghf_stamp_buf[((size_t)blockIdx.x * 8 + w) * STAMP_SLOTS + i] = st_acc[i];
#endif
}

template <int D>
static int launch_for(const MsgArgs& a, hipStream_t stream) {
    using C = MfmaCfg<D>;
    constexpr int CR = 16 * C::MTC;
    constexpr size_t lds = (size_t)((C::BN + 4) * D + 2 * CR * D) * 4 + 2 * CR * 4;
    GHF_REQUIRE(a.block_nodes == C::BN, "message(mfma): plan block_nodes=%d, kernel for d=%d needs %d", a.block_nodes, D, C::BN);
    GHF_REQUIRE(a.wlayout == GHF_WLAYOUT_FRAG16, "message(mfma): weights must be in FRAG16 layout");
    GHF_REQUIRE(a.chunk_tab && a.blk_chunk_off, "message(mfma): the plan's chunk table is missing");
    const int64_t row_end = a.row0 + a.rows;
    GHF_REQUIRE(row_end == a.N || row_end % C::BN == 0, "message(mfma): row range must end on a block boundary or at N");
    if (a.rows <= 0) return GHF_OK;
    GHF_SET_MAX_LDS(message_mfma_kernel<D>, lds);
    const unsigned grid = (unsigned)cdiv(a.rows, C::BN);
    static const int dbg = getenv("GHF_DEBUG_FLAGS") ? atoi(getenv("GHF_DEBUG_FLAGS")) : 0;   // diagnostic ablations, see DESIGN.md
    message_mfma_kernel<D><<<grid, C::NW * 64, lds, stream>>>(a.h, a.N, a.sorted_key, a.sorted_src, a.chunk_tab, a.blk_chunk_off,
                                                              a.indeg, a.R, a.W_msg, a.bias, a.ln_gamma, a.ln_beta, a.ln_eps,
                                                              a.row0, row_end, a.h_out, a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM), dbg);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

// split_chunks: a destination block with more chunks than this is cut into several work items (plan.hip).  A block of a
// uniform graph at the BASELINE configs has ~65 chunks, so only real hubs are split.
constexpr int SPLIT_CHUNKS = 128;

bool message_mfma_config(int d, int* block_nodes, int* chunk_rows, int* split_chunks) {
    *split_chunks = SPLIT_CHUNKS;
    switch (d) {
        case 128: *block_nodes = MfmaCfg<128>::BN; *chunk_rows = 16 * MfmaCfg<128>::MTC; return true;
        case 64:  *block_nodes = MfmaCfg<64>::BN;  *chunk_rows = 16 * MfmaCfg<64>::MTC;  return true;
        default:  return false;
    }
}

int launch_message_mfma(const MsgArgs& a, hipStream_t stream) {
    // the ping-pong schedule (message_pp.hip) is the default; GHF_KERNEL=lockstep selects this file's kernel for A/B runs
    static const bool lockstep = getenv("GHF_KERNEL") && !strcmp(getenv("GHF_KERNEL"), "lockstep");
    switch (a.d) {
        case 128: return lockstep ? launch_for<128>(a, stream) : launch_message_pp(a, stream);
        case 64:  return lockstep ? launch_for<64>(a, stream) : launch_message_pp(a, stream);
        default:  return set_err(GHF_EUNSUPPORTED, "message(mfma): no tuned kernel for d=%d", a.d);
    }
}

}  // namespace ghf

#ifdef GHF_STAMPS
extern "C" int ghf_debug_read_stamps(unsigned long long* host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ghf::ghf_stamp_buf), count * sizeof(unsigned long long));
}
#endif
