// message_rs.hip — the message layer for wide hidden sizes (d a multiple of 128, d >= 256: BASELINE config 5), relation-
// stationary in two passes.  reference: models/hypergnn.py:201-230 (messages, mean, mean-W_self self-loop), :288-296 (tail).
//
// Why not the destination-block kernels (message_hx.hip & co.): they re-stream a relation's [2d, d] weights for every
// (destination block, relation) chunk — 512 KB per ~10 rows at d = 256 with 256 relations — and their block sums plus
// A tiles no longer fit 160 KB of LDS.  Here the edges are grouped by RELATION instead (inside one by destination) and
// cut into tiles of 128 edges; a workgroup multiplies a tile's gathered rows [h_src | h_dst] (128 x 2d) with 128
// columns of the relation's [2d, d] weights as an ordinary LDS-tiled GEMM (fp32 MFMA 16x16x4, exact fma chain), so the
// weights are read once per 128 rows, from L2.  The per-edge results Y_e = h_u W_msg[r] + b[r] + h_v W_self[r] go to HBM
// at the edge's position in DESTINATION order; pass 2 sums each destination's contiguous rows in that fixed order
// (reproducible), divides by the in-degree and applies the tail.  Cost: one extra round trip of E x d floats through
// HBM — which is why this is the kernel for wide rows / many relations and not for C3.
//
// Pass 1 tiling: workgroup = 4 waves, C tile 128 edges x 128 columns, K = 2d in steps of 16.  LDS holds A [128][16] and
// B^T [128][16] (the weights arrive transposed, [R][d][d] with the contraction index contiguous, so both tiles are
// "16 contiguous k per row"), double buffered.  A wave owns 32 rows x 128 columns = 2 x 8 accumulator tiles; per k-step
// its operands are 10 ds_read_b128 (lane (i, q) reads k = 4q..4q+3 of row i: register s feeds MFMA s of the step, the
// same k-permutation on both operands).
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RS_TM = 128, RS_TN = 128, RS_KB = 16;

__global__ __launch_bounds__(256) void edge_transform_kernel(
    const float* __restrict__ h, int d, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
    const int64_t* __restrict__ ypos, const int64_t* __restrict__ slice_tab, const float* __restrict__ WmT,
    const float* __restrict__ WsT, const float* __restrict__ bias, float* __restrict__ Y) {
    __shared__ __attribute__((aligned(16))) float At[2][RS_TM][RS_KB];
    __shared__ __attribute__((aligned(16))) float Bt[2][RS_TN][RS_KB];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int64_t r = slice_tab[3 * (size_t)blockIdx.x], e0 = slice_tab[3 * (size_t)blockIdx.x + 1],
                  e1 = slice_tab[3 * (size_t)blockIdx.x + 2];
    const int n0 = (int)blockIdx.y * RS_TN;

    // staging map: thread t moves float4 (row t/4 + 64 j, k 4 (t%4) .. +3) of both tiles, j = 0, 1
    const int sk = 4 * (t & 3);
    int64_t su[2], sv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int64_t e = e0 + (t >> 2) + 64 * j;
        if (e >= e1) e = e1 - 1;                           // rows past the tile's end repeat its last edge (never stored)
        su[j] = src[e];
        sv[j] = dst[e];
    }
    const float* __restrict__ wm = WmT + (size_t)r * d * d;
    const float* __restrict__ ws = WsT + (size_t)r * d * d;
    f32x4 sa[2], sb[2];
    auto fetch = [&](int k0) {                             // k0: first contraction index of the step, in [0, 2d)
        const bool self = k0 >= d;
        const int kk = (self ? k0 - d : k0) + sk;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t node = self ? sv[j] : su[j];
            sa[j] = *(const f32x4*)(h + (size_t)node * d + kk);
            const int n = n0 + (t >> 2) + 64 * j;
            sb[j] = *(const f32x4*)((self ? ws : wm) + (size_t)n * d + kk);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *(f32x4*)&At[buf][(t >> 2) + 64 * j][sk] = sa[j];
            *(f32x4*)&Bt[buf][(t >> 2) + 64 * j][sk] = sb[j];
        }
    };

    f32x4 acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = 2 * d / RS_KB;
    fetch(0);
    commit(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        fetch((s + 1 < nsteps ? s + 1 : s) * RS_KB);       // (the last step is fetched twice: no branch around the loads)
        f32x4 a[2], b[8];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a[rt] = *(const f32x4*)&At[buf][32 * w + 16 * rt + c16][4 * q];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) b[ct] = *(const f32x4*)&Bt[buf][16 * ct + c16][4 * q];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][ks], b[ct][ks], acc[rt][ct], 0, 0, 0);
        commit(buf ^ 1);
        __syncthreads();
    }
    // D: lane holds rows 4q + s, column c16 of a tile
    float bv[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) bv[ct] = bias[(size_t)r * d + n0 + 16 * ct + c16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t e = e0 + 32 * w + 16 * rt + 4 * q + s;
            if (e < e1) {
                float* __restrict__ y = Y + (size_t)ypos[e] * d + n0 + c16;
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) y[16 * ct] = acc[rt][ct][s] + bv[ct];
            }
        }
}

// ---- pass 1 with the contraction of message_hx.hip: two fp16 pieces per operand, three 16x16x32 products -----------
// (its header: x 2^s = hi + lo, hi*hi + hi*lo + lo*hi in fp32, the exact scales taken out again).  Operands:
//   h_split        ghf_split_rows(h, SPLIT2H): N rows [hi d | lo d] fp16, then N float 2^-s(row)
//   w2h            ghf_weights_pack_rs: per relation [half: msg, self][piece: hi, lo][n][k] fp16 (TRANSPOSED: k contiguous),
//                  all relations, then R float 2^-s(relation)
// K runs over the source half and then the destination half of a row pair; the two halves have different row scales:
// between them the accumulators are multiplied by 2^-s(src) / 2^-s(dst) — a power of two, exact — so that one set
// serves both (64 registers less: two waves per SIMD instead of one) and 2^-s(dst) 2^-s(relation) is taken out at the end.  Tiles per step: 32 k of 128 rows and 128 columns, both pieces (4 x 8 KB), double
// buffered.  fp32 MFMA: 64 MFMAs x 32 cycles per 16 k; here 48 x 16 cycles per 32 k — 5.3 x less matrix time.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// KH = k per step: 32 (64-byte pieces of a row per visit; two workgroups per CU) or, for d % 256 == 0, 64 (whole 128-byte
// lines, a half-row in four visits instead of eight; 129 KB of LDS, one workgroup per CU, the 16-byte granules of a row
// XOR-swizzled by row & 7 so that a fragment read of 16 rows covers all banks).
// NCT = 128-column groups per workgroup (round 3): at d % 256 == 0 a workgroup of EIGHT waves takes a row tile's 128 rows
// times 256 columns — waves 0-3 the first 128 columns, waves 4-7 the next, all eight reading the one A tile in LDS — so the
// tile's rows are gathered once per 256 columns instead of once per 128: at d = 256 every gathered row is fetched once, not
// twice (BASELINE config 5: 91 GB of 205 GB per layer were pass 1's fetches, profiles/r02_c5_kernel_pmc.json).
#ifndef GHF_RSEXP
#define GHF_RSEXP 0          // timing-only ablations of pass 1 (wrong results): 1 no MFMAs, 2 no global fetches, 4 no commits to LDS, 8 no fragment reads, 16 every gathered row one of 64 (L2-resident), 32 no barrier per step, 64 no result rows written
#endif
template <int KH, int NCT>
__global__ __launch_bounds__(256 * NCT, KH == 32 ? 2 : 1) void edge_transform_h_kernel(
    const char* __restrict__ h_split, int64_t N, int d, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
    const int64_t* __restrict__ ypos, const int64_t* __restrict__ slice_tab, const char* __restrict__ w2h, int R,
    const float* __restrict__ bias, const char* __restrict__ x_split, int64_t NX, const float* __restrict__ row_cnt,
    float* __restrict__ Y) {
    constexpr int RS_KH = KH, GR = KH / 8, GPT = KH / 16;  // granules (8 fp16) per tile row; per thread, piece and step of B
    constexpr int GPA = GPT / NCT, TPR = GR / GPA;         // A tile: granules per thread, threads per row (NCT x 256 threads, 128 rows)
    static_assert(GPA >= 1 && (NCT == 1 || NCT == 2), "at most two column groups per workgroup");
    extern __shared__ __attribute__((aligned(16))) char rs_lds[];
    typedef _Float16 (*tile_t)[2][RS_TM][KH];              // [buffer][piece][row][k]
    typedef _Float16 (*btile_t)[2][RS_TN * NCT][KH];
    tile_t At = (tile_t)rs_lds;
    btile_t Bt = (btile_t)(rs_lds + (size_t)2 * 2 * RS_TM * KH * 2);
    // per tile row: 2^-s of its source row, n 2^-s of its destination row, n — n = the number of edges the row stands for
    // (ghf.h: rows of pre-summed runs; 1 without them, and every product below is then what it was)
    float (*rsc)[RS_TM] = (float (*)[RS_TM])(rs_lds + (size_t)2 * 2 * (1 + NCT) * RS_TM * KH * 2);
    // 16-byte granule g of tile row `row` sits at slot swz(row, g) of the row.  KH = 64: 8 granules, key row & 7.  KH = 32: a row
    // is 64 bytes, four rows share a 256-byte bank line, and a ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31}, ... together (MI355X_MICROARCH.md, LDS): unswizzled, rows r and r + 12 (and r + 4, r + 8 of the neighbouring
    // k-quarter) of such a group shared banks — every fragment read took two passes (round 3: LDS time per step was at the
    // matrix time).  Key (-(row >> 2)) & 3 = 0, 3, 2, 1 for the four row quads puts the sixteen lanes of every group in
    // sixteen different slots.
    auto swz = [](int row, int g) { return KH == 64 ? (g ^ (row & 7)) : (g ^ ((0 - (row >> 2)) & 3)); };
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int w = wv & 3, cg = wv >> 2;                     // row group (32 rows) and column group (128 columns) of this wave
    const int c16 = lane & 15, q = lane >> 4;
    const unsigned ncol = (unsigned)d / (RS_TN * NCT), tile = blockIdx.x / ncol;
    const int64_t r = slice_tab[3 * (size_t)tile], e0 = slice_tab[3 * (size_t)tile + 1], e1 = slice_tab[3 * (size_t)tile + 2];
    const int n0 = (int)(blockIdx.x % ncol) * RS_TN * NCT;
    const size_t hrow = (size_t)4 * d;                     // bytes per split row
    const float* __restrict__ hscale = (const float*)(h_split + (size_t)N * hrow);
    // staging map: thread t moves the 16-byte granules (row t/TPR, k 8 (GPA (t%TPR) + i) .. +7), i < GPA, of both pieces of
    // the A tile, and (row t/2 of 128 NCT, k 8 (GPT (t%2) + i) .. +7), i < GPT, of the B tile
    const int srow = t / TPR, sg = GPA * (t % TPR);
    const int brow_i = t >> 1, sgb = GPT * (t & 1);
    int64_t e = e0 + srow;
    if (e >= e1) e = e1 - 1;                               // rows past the tile's end repeat its last edge (never stored)
    const int64_t su = src[e], sv = dst[e];               // su < 0: row ~su of x_split (the sum of a run's source rows)
    const char* __restrict__ urow = su >= 0 ? h_split + (size_t)su * hrow : x_split + (size_t)(~su) * hrow;
    if ((t % TPR) == 0) {
        const float n = row_cnt ? row_cnt[e] : 1.0f;
        rsc[0][srow] = su >= 0 ? hscale[su] : ((const float*)(x_split + (size_t)NX * hrow))[~su];
        rsc[1][srow] = hscale[sv] * n;
        rsc[2][srow] = n;
    }
    const char* __restrict__ wr = w2h + (size_t)r * 8 * d * d;          // [half][piece][n][k] fp16 = 8 d^2 bytes per relation
    const float wscale = ((const float*)(w2h + (size_t)R * 8 * d * d))[r];
    // Four register sets: the rows of a step are requested three steps before they are written to LDS — this kernel's 293
    // registers leave one wave per SIMD, so a gather's latency (an HBM miss, ~2 us) has to be covered by this wave's own
    // MFMAs (768 cycles per step).
    struct Stage { i32x4 a[2][GPA], b[2][GPT]; };          // [piece][i]
    Stage st[4];
    auto fetch = [&](int k0, Stage& S) {                   // k0: first contraction index of the step, in [0, 2d)
        if ((GHF_RSEXP & 2) && k0 > 2 * RS_KH) return;
        const int half = k0 >= d;
        const int kk = half ? k0 - d : k0;
        const char* arow = half ? h_split + (size_t)sv * hrow : urow;
        const char* brow = wr + ((size_t)half * 2 * d + (n0 + brow_i)) * (size_t)d * 2;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < GPA; ++i) S.a[pl][i] = *(const i32x4*)(arow + (size_t)pl * d * 2 + (size_t)(kk + 8 * (sg + i)) * 2);
#pragma unroll
            for (int i = 0; i < GPT; ++i) S.b[pl][i] = *(const i32x4*)(brow + (size_t)pl * d * d * 2 + (size_t)(kk + 8 * (sgb + i)) * 2);
        }
    };
    auto commit = [&](int buf, const Stage& S) {
        if ((GHF_RSEXP & 4) && buf >= 0) { asm volatile("" :: "v"(S.a[0][0]), "v"(S.b[0][0]), "v"(S.a[1][0]), "v"(S.b[1][0])); return; }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < GPA; ++i) *(i32x4*)&At[buf][pl][srow][8 * swz(srow, sg + i)] = S.a[pl][i];
#pragma unroll
            for (int i = 0; i < GPT; ++i) *(i32x4*)&Bt[buf][pl][brow_i][8 * swz(brow_i, sgb + i)] = S.b[pl][i];
        }
    };

    f32x4 acc[2][8];                                       // [row tile][column tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int half_steps = d / RS_KH, total = 2 * half_steps;              // (half_steps % 4 == 0: d % 128 == 0, d % 256 == 0 for KH = 64)
    fetch(0, st[0]);
    fetch(1 * RS_KH, st[1]);
    fetch(2 * RS_KH, st[2]);
    commit(0, st[0]);
    __syncthreads();
    auto run_step = [&](auto hf_tag, auto i_tag, int step) {
        constexpr int hf = decltype(hf_tag)::value, i4 = decltype(i_tag)::value;
        {
            const int buf = step & 1;
            const int pre = step + 3 < total ? step + 3 : total - 1;       // (past the end: the last step again, no branch around loads)
            fetch(pre * RS_KH, st[(i4 + 3) & 3]);
#pragma unroll
            for (int kk = 0; kk < KH / 32; ++kk) {
                i32x4 a[2][2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const int row = 32 * w + 16 * rt + c16;
                        a[rt][pl] = *(const i32x4*)&At[buf][pl][row][8 * swz(row, 4 * kk + q)];
                    }
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {           // the column tiles in two groups of four: 32 fragment registers live, not 64
                    i32x4 b[4][2];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            const int row = RS_TN * cg + 16 * (4 * ch + c4) + c16;
                            b[c4][pl] = *(const i32x4*)&Bt[buf][pl][row][8 * swz(row, 4 * kk + q)];
                        }
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const int ct = 4 * ch + c4;
                            // the product TRANSPOSED (the weights' fragment as the A operand): a lane then holds four
                            // consecutive columns of ONE row — 16-byte stores of the results (one column of four rows:
                            // 4-byte stores, four times the store instructions)
                            auto fma = [&](int pa, int pb) {
                                if (GHF_RSEXP & 1) { asm volatile("" :: "v"(b[c4][pb]), "v"(a[rt][pa])); return; }
                                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[c4][pb]),
                                                                                     __builtin_bit_cast(f16x8, a[rt][pa]), acc[rt][ct], 0, 0, 0);
                            };
                            fma(1, 0); fma(0, 1);          // lo*hi, hi*lo
                            fma(0, 0);                     // hi*hi
                        }
                }
            }
            commit(buf ^ 1, st[(i4 + 1) & 3]);
            __syncthreads();
        }
    };
    auto run_half = [&](auto hf_tag) {
        constexpr int hf = decltype(hf_tag)::value;
        for (int s = 0; s < half_steps; s += 4) {
            const int step = hf * half_steps + s;
            run_step(hf_tag, std::integral_constant<int, 0>{}, step);
            run_step(hf_tag, std::integral_constant<int, 1>{}, step + 1);
            run_step(hf_tag, std::integral_constant<int, 2>{}, step + 2);
            run_step(hf_tag, std::integral_constant<int, 3>{}, step + 3);
        }
    };
    run_half(std::integral_constant<int, 0>{});
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {                       // from the source rows' scale to the destination rows' (exact)
        const int row = 32 * w + 16 * rt + c16;
        const float ratio = rsc[0][row] / rsc[1][row];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[rt][ct] *= ratio;
    }
    run_half(std::integral_constant<int, 1>{});
    // D (transposed product): lane holds row c16 of a row tile, columns 4q + s of a column tile
    f32x4 bv[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[ct][s] = bias[(size_t)r * d + n0 + RS_TN * cg + 16 * ct + 4 * q + s];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int row = 32 * w + 16 * rt + c16;
        const int64_t ee = e0 + row;
        if (ee < e1) {
            const float fv = rsc[1][row] * wscale, n = rsc[2][row];
            float* __restrict__ y = Y + (size_t)ypos[ee] * d + n0 + RS_TN * cg + 4 * q;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                f32x4 o;
#pragma unroll
                for (int s = 0; s < 4; ++s) o[s] = fmaf(acc[rt][ct][s], fv, bv[ct][s] * n);
                if (!(GHF_RSEXP & 64) || o[0] == 123.456f) *(f32x4*)(y + 16 * ct) = o;   // (64, timing: no result rows written)
            }
        }
    }
}

// ---- pass 1, round 4: the same contraction with the tiles gathered by LDS-DMA into a three-deep ring -------------------------
// edge_transform_h_kernel<32, 2> stages its tiles through registers (96 of them) behind ONE barrier per k-step that all eight
// waves reach in lockstep: per step every wave waits for its fragment reads, then every wave issues its MFMAs, then every wave
// writes the next tile — the matrix pipe, the LDS and the vector-memory path take turns (ablations, tools/c5_shard_check.py with
// GHF_VARIANT=rsexp<mask>: MFMAs, fetches and commits each cost ~0.7 of ~3.6 ms and add up).  Here (d % 256 == 0, 128 rows x
// 256 columns per workgroup, k-steps of 32, eight waves as before):
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4: per-lane 64-bit source address, 1 KiB per wave instruction, the granule
//     swizzle applied on the source side), TWO steps ahead, into a ring of three buffers — no staging registers, no ds_write;
//   * a wave's fragment reads run one half-step ahead of its MFMAs (the second half's weights are requested in front of the first
//     half's products, the next step's first fragments in front of the second half's), so the matrix pipe always has reads in
//     flight behind it and the two waves of a SIMD interleave without a common phase;
//   * one barrier per step, in its middle, raw s_barrier (a __syncthreads() would drain the DMAs in flight).  Ordering, by the
//     count (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"): step s + 1's tile is
//     complete when every wave has passed vmcnt(6) — its six DMAs for step s + 2, issued at the top of step s, are the only ones
//     left in flight — and the barrier; it is read behind that barrier only.  Buffer (s + 2) % 3 = (s - 1) % 3 is regathered into
//     at the top of step s: its last reads (step s - 1's second-half weights) were retired by every wave's lgkmcnt(0) in front of
//     step s - 1's barrier.
// The rest — operands, scales, the exact rescale between the row halves, the transposed product and the 16-byte stores — is
// edge_transform_h_kernel's.
// Measured (one GPU's share of C5, 27,118 tiles; tools/c5_shard_check.py under rocprofv3): 4.37 -> 3.95 ms per launch, the whole
// C5 layer 41.3 -> 38.8 ms.  Ablations of THIS kernel (GHF_VARIANT=rsexp<mask>): without MFMAs 3.14, without the gathers 2.60,
// with every gathered row one of 64 L2-resident rows 3.21 — its matrix work alone is 1.09 ms at the dense fp16 peak.  What is left
// (Second set, another box, 3.80 ms: no fragment reads 3.34, no barrier 3.68, no result rows 3.77, neither MFMAs nor fragment
// reads 3.11 — a kernel that only moves its tiles into LDS still takes 82 % of the time.  The ring is a FIFO (loads return in
// order): two steps = 96 KB in flight per CU against a loaded round trip of ~3 us for the gathered rows' 64-byte pieces is
// 5.5 TB/s of tile traffic, what the kernel runs at.  More in flight needs more LDS than there is: the weights' tile is two
// thirds of a step's bytes and must be there for all four row groups; a deeper ring for the rows alone does not help, because
// a weight tile requested later with an earlier deadline waits in line behind the rows requested before it.)
// What is left
// is per work item: the descriptor chain in front of the first tile (slice entry -> row ids -> rows: ~5 us of ~37 us, nothing else
// runs on the CU meanwhile at one 149 KB workgroup per CU) and row latency at two steps of lookahead.  Tried and NOT kept: a
// persistent form (one workgroup per CU walking the items, rows three / weights two steps ahead in 4 + 3 buffers = all 160 KiB,
// the next item's descriptors prefetched in stages): as soon as a compiler-visible load is in flight across the loop's merges
// hipcc puts an s_waitcnt vmcnt(0) in front of the ring's next request — every step drains the ring — and hiding those loads in
// asm makes the register copies at the merges read them before they land; the descriptors would have to come through LDS.
typedef __attribute__((address_space(3))) void* rs_lptr_t;
typedef __attribute__((address_space(1))) const void* rs_gptr_t;
__global__ __launch_bounds__(512, 2) void edge_transform_h3_kernel(
    const char* __restrict__ h_split, int64_t N, int d, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
    const int64_t* __restrict__ ypos, const int64_t* __restrict__ slice_tab, const char* __restrict__ w2h, int R,
    const float* __restrict__ bias, const char* __restrict__ x_split, int64_t NX, const float* __restrict__ row_cnt,
    float* __restrict__ Y) {
    constexpr int KH = 32, NB = 3;
    constexpr unsigned A_BYTES = 2 * RS_TM * KH * 2, B_BYTES = 2 * 2 * RS_TN * KH * 2, BUF = A_BYTES + B_BYTES;   // 16 KB + 32 KB
    extern __shared__ __attribute__((aligned(16))) char rs_lds[];
    float (*rsc)[RS_TM] = (float (*)[RS_TM])(rs_lds + (size_t)NB * BUF);
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int w = wv & 3, cg = wv >> 2;
    const int c16 = lane & 15, q = lane >> 4;
    const unsigned ncol = (unsigned)d / (2 * RS_TN), tile = blockIdx.x / ncol;
    const int64_t r = slice_tab[3 * (size_t)tile], e0 = slice_tab[3 * (size_t)tile + 1], e1 = slice_tab[3 * (size_t)tile + 2];
    const int n0 = (int)(blockIdx.x % ncol) * 2 * RS_TN;
    const size_t hrow = (size_t)4 * d;
    const float* __restrict__ hscale = (const float*)(h_split + (size_t)N * hrow);
    auto key = [](int row) { return (0 - (row >> 2)) & 3; };      // granule g of tile row `row` sits at slot g ^ key(row) (see above)
    // ---- what this wave gathers per step: A pieces wv (hi plane) and wv + 8 (lo plane) = rows 16 wv .. + 15; B pieces: rows
    // 16 wv .. and 16 (wv + 8) .. of both planes.  Lane l: row l / 4 of the piece, slot l % 4.
    const int prow = lane >> 2, slot = lane & 3;
    const int arow_i = 16 * wv + prow;
    int64_t e = e0 + arow_i;
    if (e >= e1) e = e1 - 1;                               // rows past the tile's end repeat its last edge (never stored)
    const int64_t su = src[e], sv = dst[e];               // su < 0: row ~su of x_split (the sum of a run's source rows)
    const char* __restrict__ urow = (GHF_RSEXP & 16) ? h_split + (size_t)(arow_i & 63) * hrow          // (timing: rows that stay in L2)
                                    : su >= 0 ? h_split + (size_t)su * hrow : x_split + (size_t)(~su) * hrow;
    const char* __restrict__ vrow = h_split + (size_t)((GHF_RSEXP & 16) ? (arow_i & 63) : sv) * hrow;
    const int ga = (slot ^ key(arow_i)) * 16;              // byte offset of my granule inside a 64-byte k-step of the row
    if (slot == 0) {
        const float n = row_cnt ? row_cnt[e] : 1.0f;
        rsc[0][arow_i] = su >= 0 ? hscale[su] : ((const float*)(x_split + (size_t)NX * hrow))[~su];
        rsc[1][arow_i] = hscale[sv] * n;
        rsc[2][arow_i] = n;
    }
    const char* __restrict__ wr = w2h + (size_t)r * 8 * d * d;          // [half][piece][n][k] fp16 = 8 d^2 bytes per relation
    const float wscale = ((const float*)(w2h + (size_t)R * 8 * d * d))[r];
    const int brow0 = 16 * wv + prow, brow1 = 16 * (wv + 8) + prow;     // B tile rows (= output columns n0 + brow)
    const int gb0 = (slot ^ key(brow0)) * 16, gb1 = (slot ^ key(brow1)) * 16;
    const char* __restrict__ bq0 = wr + (size_t)(n0 + brow0) * d * 2 + gb0;
    const char* __restrict__ bq1 = wr + (size_t)(n0 + brow1) * d * 2 + gb1;
    const size_t plane_b = (size_t)d * d * 2, half_b = (size_t)2 * d * d * 2;
    const int half_steps = d / KH, total = 2 * half_steps;
    auto dma = [&](int step) {                             // six 1 KiB pieces of step `step`'s tile into buffer step % 3
        if (GHF_RSEXP & 2) return;
        const int st = step < total ? step : total - 1;    // (past the end: the last step again — the count stays six)
        const int half = st >= half_steps;
        const size_t kb = (size_t)(half ? st - half_steps : st) * KH * 2;
        const char* ap = (half ? vrow : urow) + kb + ga;
        const unsigned base = (unsigned)(st % NB) * BUF;
        const unsigned la = base + (unsigned)wv * 1024u;                       // A [plane][128 rows][64 B]: piece wv of plane 0
        const unsigned lb = base + A_BYTES + (unsigned)wv * 1024u;             // B [plane][256 rows][64 B]
        const size_t bo = (size_t)half * half_b + kb;
        __builtin_amdgcn_global_load_lds((rs_gptr_t)ap, (rs_lptr_t)(rs_lds + la), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((rs_gptr_t)(ap + (size_t)d * 2), (rs_lptr_t)(rs_lds + la + RS_TM * KH * 2), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((rs_gptr_t)(bq0 + bo), (rs_lptr_t)(rs_lds + lb), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((rs_gptr_t)(bq1 + bo), (rs_lptr_t)(rs_lds + lb + 8 * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((rs_gptr_t)(bq0 + bo + plane_b), (rs_lptr_t)(rs_lds + lb + 2 * RS_TN * KH * 2), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((rs_gptr_t)(bq1 + bo + plane_b), (rs_lptr_t)(rs_lds + lb + 2 * RS_TN * KH * 2 + 8 * 1024), 16, 0, 0);
    };
    // ---- fragment reads (through asm: hipcc would order a plain LDS read behind every DMA in flight — vmcnt(0)) ----
    const unsigned lds0 = (unsigned)(size_t)(rs_lptr_t)rs_lds;
    unsigned aoff[2], boff[8];                              // byte offsets inside a buffer of my fragments' hi pieces
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int row = 32 * w + 16 * rt + c16;
        aoff[rt] = (unsigned)row * 64u + (unsigned)((q ^ key(row)) * 16);
    }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
        const int row = RS_TN * cg + 16 * ct + c16;
        boff[ct] = A_BYTES + (unsigned)row * 64u + (unsigned)((q ^ key(row)) * 16);
    }
    i32x4 a[2][2], b0[4][2], b1[4][2];
    auto read_a = [&](unsigned buf) {
        if (GHF_RSEXP & 8) return;                         // (timing: no fragment reads)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const unsigned ad = lds0 + buf + aoff[rt];
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(a[rt][0]), "=&v"(a[rt][1]) : "v"(ad) : "memory");
        }
    };
    auto read_b = [&](unsigned buf, int ch, i32x4 (&b)[4][2]) {
        if (GHF_RSEXP & 8) return;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const unsigned ad = lds0 + buf + boff[4 * ch + c4];
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16384" : "=&v"(b[c4][0]), "=&v"(b[c4][1]) : "v"(ad) : "memory");
        }
    };
    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mfma_half = [&](int ch, i32x4 (&b)[4][2]) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int ct = 4 * ch + c4;
                auto fma = [&](int pa, int pb) {
                    if (GHF_RSEXP & 1) { asm volatile("" :: "v"(b[c4][pb]), "v"(a[rt][pa])); return; }
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[c4][pb]),
                                                                         __builtin_bit_cast(f16x8, a[rt][pa]), acc[rt][ct], 0, 0, 0);
                };
                fma(1, 0); fma(0, 1);                      // lo*hi, hi*lo
                fma(0, 0);                                 // hi*hi
            }
    };
    // ---- prologue: steps 0 and 1 requested, step 0 landed, its first fragments requested ----
    dma(0);
    dma(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the scales written above)
    __builtin_amdgcn_s_barrier();
    read_a(0);
    read_b(0, 0, b0);
    for (int s = 0; s < total; ++s) {
        const unsigned cur = (unsigned)(s % NB) * BUF, nxt = (unsigned)((s + 1) % NB) * BUF;
        dma(s + 2);
        read_b(cur, 1, b1);
        // the first half's fragments (a, b0: requested during the previous step) have landed once only b1's eight reads are out
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]),
                     "+v"(b0[0][0]), "+v"(b0[0][1]), "+v"(b0[1][0]), "+v"(b0[1][1]), "+v"(b0[2][0]), "+v"(b0[2][1]), "+v"(b0[3][0]), "+v"(b0[3][1]) :: "memory");
        if (s == half_steps) {                             // from the source rows' scale to the destination rows' (exact: a power of two)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = 32 * w + 16 * rt + c16;
                const float ratio = rsc[0][row] / rsc[1][row];
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) acc[rt][ct] *= ratio;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                 // (register-only instructions float across asm waits: pin the first half's
        mfma_half(0, b0);                                  //  products between the fragments' wait and the barrier)
        __builtin_amdgcn_sched_barrier(0);
        // step s + 1's tile: my pieces have landed (only the six just requested are in flight), my reads of this step's tile are
        // done; behind the barrier that holds for every wave
        asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" : "+v"(b1[0][0]), "+v"(b1[0][1]), "+v"(b1[1][0]), "+v"(b1[1][1]),
                     "+v"(b1[2][0]), "+v"(b1[2][1]), "+v"(b1[3][0]), "+v"(b1[3][1]) :: "memory");
        if (!(GHF_RSEXP & 32)) __builtin_amdgcn_s_barrier();   // (32, timing: no barrier per step)
        i32x4 a_keep[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a_keep[rt][pl] = a[rt][pl];
        if (s + 1 < total) {
            read_a(nxt);                                   // (into a: the second half below multiplies a_keep)
            read_b(nxt, 0, b0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int ct = 4 + c4;
                auto fma = [&](int pa, int pb) {
                    if (GHF_RSEXP & 1) { asm volatile("" :: "v"(b1[c4][pb]), "v"(a_keep[rt][pa])); return; }
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b1[c4][pb]),
                                                                         __builtin_bit_cast(f16x8, a_keep[rt][pa]), acc[rt][ct], 0, 0, 0);
                };
                fma(1, 0); fma(0, 1);
                fma(0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the two clamped requests past the end: nothing may land after the exit)
    // D (transposed product): lane holds row c16 of a row tile, columns 4q + s of a column tile
    f32x4 bv[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bv[ct][s4] = bias[(size_t)r * d + n0 + RS_TN * cg + 16 * ct + 4 * q + s4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int row = 32 * w + 16 * rt + c16;
        const int64_t ee = e0 + row;
        if (ee < e1) {
            const float fv = rsc[1][row] * wscale, n = rsc[2][row];
            float* __restrict__ y = Y + (size_t)ypos[ee] * d + n0 + RS_TN * cg + 4 * q;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                f32x4 o;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) o[s4] = fmaf(acc[rt][ct][s4], fv, bv[ct][s4] * n);
                *(f32x4*)(y + 16 * ct) = o;
            }
        }
    }
}

// ghf_weights_pack_rs: natural W_msg, W_self [R][d][d] -> w2h (see above).  One workgroup per relation finds the
// largest magnitude of both matrices; a second kernel scales, cuts and transposes.
__global__ __launch_bounds__(256) void rs_wmax_kernel(const float* __restrict__ Wm, const float* __restrict__ Ws, int d,
                                                      float* __restrict__ inv_scale, int* __restrict__ shift,
                                                      int32_t* __restrict__ range_flag) {
    __shared__ float red[4];
    __shared__ WeakRows weak_red[4];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t n = (size_t)d * d;
    // a wave per input row: the largest magnitude, and the rows' L1 norms for the weak-row guard (common.h: WeakRows)
    float mx = 0.f;
    WeakRows wr;
    wr.init();
    for (int row = wv; row < 2 * d; row += 4) {
        const float* __restrict__ p = (row < d ? Wm : Ws) + r * n + (size_t)(row < d ? row : row - d) * d;
        float s = 0.f;
        for (int o = lane; o < d; o += 64) { const float a = fabsf(p[o]); s += a; mx = fmaxf(mx, a); }
        wr.add(row >= d, wave_sum(s));
    }
    mx = wave_absmax(mx);
    if (lane == 0) { red[wv] = mx; weak_red[wv] = wr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int sh = split2h_shift(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        shift[r] = sh;
        inv_scale[r] = pow2f(-sh);
        for (int i = 1; i < 4; ++i) wr.merge(weak_red[i]);
        range_raise_weak(range_flag, wr, d);
    }
}
__global__ __launch_bounds__(256) void rs_wpack_kernel(const float* __restrict__ Wm, const float* __restrict__ Ws, int d,
                                                       const int* __restrict__ shift, _Float16* __restrict__ out,
                                                       int32_t* __restrict__ range_flag) {
    __shared__ float tile[32][33];
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    const int r = blockIdx.z >> 1, half = blockIdx.z & 1;
    const float* __restrict__ W = (half ? Ws : Wm) + (size_t)r * d * d;             // natural [k][n]
    const float up = pow2f(shift[r]);
    const int k0 = blockIdx.y * 32, nn0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i >> 5][i & 31] = W[(size_t)(k0 + (i >> 5)) * d + nn0 + (i & 31)];
    __syncthreads();
    _Float16* __restrict__ base = out + ((size_t)r * 4 + (size_t)half * 2) * d * d;    // [piece][n][k]
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int n = nn0 + (i >> 5), k = k0 + (i & 31);
        _Float16 hi, lo;
        const float xs = tile[i & 31][i >> 5] * up;
        split2h(xs, hi, lo);
        base[(size_t)n * d + k] = hi;
        base[(size_t)d * d + (size_t)n * d + k] = lo;
        if (range_tiny(xs)) atomicAdd(&cnt[0], 1);       // range guard (common.h), per 32 x 32 tile of the relation's matrices
        if (xs != 0.f) atomicAdd(&cnt[1], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) range_raise(range_flag, GHF_RANGE_WEIGHTS, cnt[0], cnt[1]);
}

constexpr int RS_MAX_D = 1024, RS_PER_LANE = RS_MAX_D / 64;

// Hubs (a power-law graph's destinations with more rows than one wave should walk): their rows are cut into chunks
// (hub_chunks: first row, end row, slot), one workgroup sums a chunk — each wave a quarter, the quarters added in wave
// order — into P[slot]; pass 2 then sums a hub's slots instead of its rows.  Fixed order throughout.
__global__ __launch_bounds__(256) void segment_partial_kernel(const float* __restrict__ Y, const int64_t* __restrict__ hub_chunks,
                                                              int d, float* __restrict__ P) {
    __shared__ float red[4][RS_MAX_D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t p0 = hub_chunks[3 * (size_t)blockIdx.x], p1 = hub_chunks[3 * (size_t)blockIdx.x + 1],
                  slot = hub_chunks[3 * (size_t)blockIdx.x + 2];
    const int64_t per = (p1 - p0 + 3) / 4;
    const int64_t a = p0 + w * per, bnd = a + per < p1 ? a + per : p1;
    for (int o = lane; o < d; o += 64) {
        const float* __restrict__ p = Y + (size_t)a * d + o;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int64_t j = 0;
        const int64_t n = bnd - a;
        for (; j + 4 <= n; j += 4) {
            t0 += p[(size_t)j * d];
            t1 += p[(size_t)(j + 1) * d];
            t2 += p[(size_t)(j + 2) * d];
            t3 += p[(size_t)(j + 3) * d];
        }
        for (; j < n; ++j) t0 += p[(size_t)j * d];
        red[w][o] = (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < d; o += 256) P[(size_t)slot * d + o] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
}

// Pass 2: out_v = (1/max(indeg,1)) * sum of v's rows of Y (contiguous: off[v] .. off[v+1]; a hub's slots of P instead),
// then the tail.  One wave per destination; CPL = d / 64 columns per lane, all of a row's loads issued together and two
// rows in flight (most destinations of a sharded power-law graph have a handful of rows: the wave's time is the
// latency of its few loads, so they must not queue behind each other).  Fixed summation order.
template <int CPL>
__global__ __launch_bounds__(256) void segment_tail_kernel(
    const float* __restrict__ Y, const int64_t* __restrict__ off, const int32_t* __restrict__ deg_of, const int32_t* __restrict__ hub_of,
    const int64_t* __restrict__ hub_tab, const float* __restrict__ P, const float* __restrict__ h, const float* __restrict__ g,
    const float* __restrict__ b, float eps, int64_t row0, int64_t row_end, float* __restrict__ h_out,
    char* __restrict__ h_split_out, int64_t n_split, int no_tail, int32_t* __restrict__ range_flag) {
    constexpr int d = 64 * CPL;
    const int lane = threadIdx.x & 63;
    const int64_t v = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= row_end) return;
    int64_t p0 = off[v], p1 = off[v + 1];
    const int64_t indeg = deg_of ? deg_of[v] : p1 - p0;      // (rows of pre-summed runs stand for several edges each)
    float hv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) hv[c] = no_tail ? 0.f : h[(size_t)v * d + lane + 64 * c];
    const int hub = hub_of ? hub_of[v] : -1;
    if (hub >= 0) {                                        // (uniform per wave)
        p0 = hub_tab[2 * (size_t)hub];
        p1 = p0 + hub_tab[2 * (size_t)hub + 1];
        Y = P;
    }
    const int64_t deg = p1 - p0;                           // rows to add; the mean divides by the in-degree
    const float inv = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(indeg > 1 ? indeg : 1);
    float t0[CPL], t1[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) t0[c] = t1[c] = 0.f;
    const float* __restrict__ p = Y + (size_t)p0 * d + lane;
    int64_t j = 0;
    for (; j + 2 <= deg; j += 2) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            t0[c] += p[(size_t)j * d + 64 * c];
            t1[c] += p[(size_t)(j + 1) * d + 64 * c];
        }
    }
    if (j < deg) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) t0[c] += p[(size_t)j * d + 64 * c];
    }
    float x[CPL];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const float tt = (t0[c] + t1[c]) * inv;
        x[c] = no_tail ? tt : fmaxf(tt + hv[c], 0.f);
        s += x[c];
    }
    if (!no_tail) {
        const float mean = wave_sum(s) * (1.0f / d);
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const float tt = x[c] - mean; var += tt * tt; }
        const float rstd = 1.0f / sqrtf(wave_sum(var) * (1.0f / d) + eps);
#pragma unroll
        for (int c = 0; c < CPL; ++c) x[c] = (x[c] - mean) * rstd * g[lane + 64 * c] + b[lane + 64 * c];
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) h_out[(size_t)v * d + lane + 64 * c] = x[c];
    if (h_split_out) {                                     // the same row cut into its two fp16 pieces, for the next layer's pass 1
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) mx = fmaxf(mx, fabsf(x[c]));
        const int sh = split2h_shift(wave_absmax(mx));
        const float up = pow2f(sh);
        _Float16* __restrict__ sp = (_Float16*)(h_split_out + (size_t)v * 4 * d);
        int tiny = 0, nz = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            _Float16 hi, lo;
            split2h(x[c] * up, hi, lo);
            sp[lane + 64 * c] = hi;
            sp[d + lane + 64 * c] = lo;
            tiny += range_tiny(x[c] * up);
            nz += x[c] != 0.f;
        }
        if (lane == 0) *(float*)(h_split_out + (size_t)n_split * 4 * d + (size_t)v * 4) = pow2f(-sh);
        if (__ballot(tiny != 0)) {                         // range guard (common.h)
            tiny = (int)wave_sum((float)tiny);
            nz = (int)wave_sum((float)nz);
            if (lane == 0) range_raise(range_flag, GHF_RANGE_ROWS, tiny, nz);
        }
    }
}

// Pass 0 (graphs with hubs): the layer is linear in the source rows, so the edges of one (destination, relation) run need one
// row of pass 1 between them: x = sum of the run's source rows (here, fp32, in edge order, four rows in flight), and
//   sum_{e in run} (h_u W_msg[r] + b[r] + h_v W_self[r]) = x W_msg[r] + n (b[r] + h_v W_self[r]).
// One wave per run writes x cut into its two fp16 pieces (the form pass 1 gathers), row x of x_split [nruns rows, nruns scales].
template <int CPL>
__global__ __launch_bounds__(256) void run_rows_kernel(const float* __restrict__ h, const int64_t* __restrict__ run_src,
                                                       const int64_t* __restrict__ run_start, int64_t nruns,
                                                       char* __restrict__ x_split, int32_t* __restrict__ range_flag) {
    constexpr int d = 64 * CPL;
    const int lane = threadIdx.x & 63;
    const int64_t x = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (x >= nruns) return;
    const int64_t p0 = run_start[x], p1 = run_start[x + 1];
    float t[4][CPL];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < CPL; ++c) t[a][c] = 0.f;
    int64_t j = p0;
    for (; j + 4 <= p1; j += 4) {
        const float* __restrict__ r0 = h + (size_t)run_src[j] * d + lane;
        const float* __restrict__ r1 = h + (size_t)run_src[j + 1] * d + lane;
        const float* __restrict__ r2 = h + (size_t)run_src[j + 2] * d + lane;
        const float* __restrict__ r3 = h + (size_t)run_src[j + 3] * d + lane;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            t[0][c] += r0[64 * c];
            t[1][c] += r1[64 * c];
            t[2][c] += r2[64 * c];
            t[3][c] += r3[64 * c];
        }
    }
    for (; j < p1; ++j) {
        const float* __restrict__ r0 = h + (size_t)run_src[j] * d + lane;
#pragma unroll
        for (int c = 0; c < CPL; ++c) t[0][c] += r0[64 * c];
    }
    float v[CPL];
    float mx = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        v[c] = (t[0][c] + t[1][c]) + (t[2][c] + t[3][c]);
        mx = fmaxf(mx, fabsf(v[c]));
    }
    const int sh = split2h_shift(wave_absmax(mx));
    const float up = pow2f(sh);
    _Float16* __restrict__ sp = (_Float16*)(x_split + (size_t)x * 4 * d);
    int tiny = 0, nz = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        _Float16 hi, lo;
        split2h(v[c] * up, hi, lo);
        sp[lane + 64 * c] = hi;
        sp[d + lane + 64 * c] = lo;
        tiny += range_tiny(v[c] * up);
        nz += v[c] != 0.f;
    }
    if (lane == 0) *(float*)(x_split + (size_t)nruns * 4 * d + (size_t)x * 4) = pow2f(-sh);
    if (__ballot(tiny != 0)) {                             // range guard (common.h)
        tiny = (int)wave_sum((float)tiny);
        nz = (int)wave_sum((float)nz);
        if (lane == 0) range_raise(range_flag, GHF_RANGE_ROWS, tiny, nz);
    }
}

int launch_run_rows(const float* h, int64_t N, int d, const int64_t* run_src, const int64_t* run_start, int64_t nruns,
                    void* x_split, hipStream_t stream) {
    GHF_REQUIRE(message_rs_supported(d), "run_rows: d = %d is not a relation-stationary width", d);
    if (nruns <= 0) return GHF_OK;
    GHF_REQUIRE(cdiv(nruns, 4) < (1ll << 31), "run_rows: too many runs per launch");
    const unsigned grid = (unsigned)cdiv(nruns, 4);
    switch (d / 64) {
#define GHF_RS_CASE(CPL)                                                                                                        \
    case CPL:                                                                                                                   \
        run_rows_kernel<CPL><<<grid, 256, 0, stream>>>(h, run_src, run_start, nruns, (char*)x_split, range_flag_ptr());          \
        break;
        GHF_RS_CASE(2) GHF_RS_CASE(4) GHF_RS_CASE(6) GHF_RS_CASE(8) GHF_RS_CASE(10) GHF_RS_CASE(12) GHF_RS_CASE(14) GHF_RS_CASE(16)
#undef GHF_RS_CASE
    }
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

// (d = 128 runs too — GHF_KERNEL=rs selects it there for A/B — but the destination-block kernel is the default at 128)
int message_rs_supported(int d) { return d >= 128 && d <= RS_MAX_D && (d % RS_TN) == 0; }

int launch_edge_transform(const float* h, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                          const int64_t* slice_tab, int64_t nslices, const float* WmT, const float* WsT, const float* bias,
                          float* Y, hipStream_t stream) {
    GHF_REQUIRE(message_rs_supported(d), "edge_transform: d = %d has no relation-stationary kernel (d %% 128 == 0, 128 <= d <= %d)", d, RS_MAX_D);
    GHF_REQUIRE(N > 0 && nslices > 0 && nslices < (1ll << 31), "edge_transform: bad sizes");
    edge_transform_kernel<<<dim3((unsigned)nslices, (unsigned)(d / RS_TN)), 256, 0, stream>>>(h, d, src, dst, ypos, slice_tab,
                                                                                            WmT, WsT, bias, Y);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

size_t weights_rs_bytes(int R, int d) { return (size_t)R * 8 * d * d + (size_t)R * 4; }

int launch_weights_pack_rs(const float* Wm, const float* Ws, int R, int d, void* out, int* shift_ws, hipStream_t stream) {
    GHF_REQUIRE(message_rs_supported(d) && R > 0, "weights_pack_rs: d = %d is not a relation-stationary width", d);
    float* inv = (float*)((char*)out + (size_t)R * 8 * d * d);
    rs_wmax_kernel<<<R, 256, 0, stream>>>(Wm, Ws, d, inv, shift_ws, range_flag_ptr());
    GHF_LAUNCH_CHECK();
    rs_wpack_kernel<<<dim3(d / 32, d / 32, 2 * R), 256, 0, stream>>>(Wm, Ws, d, shift_ws, (_Float16*)out, range_flag_ptr());
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_edge_transform_h(const void* h_split, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                            const int64_t* slice_tab, int64_t nslices, const void* w2h, int R, const float* bias,
                            const void* x_split, int64_t NX, const float* row_cnt, float* Y, hipStream_t stream) {
    GHF_REQUIRE(message_rs_supported(d), "edge_transform: d = %d has no relation-stationary kernel (d %% 128 == 0, 128 <= d <= %d)", d, RS_MAX_D);
    GHF_REQUIRE(N > 0 && nslices > 0 && nslices < (1ll << 31), "edge_transform: bad sizes");
    // (one-dimensional grid, the column tile the fast index: the workgroups that share a tile's rows run side by side
    // and share them in L2)
    GHF_REQUIRE(nslices * (d / RS_TN) < (1ll << 31), "edge_transform: too many tiles per launch");
    const unsigned grid = (unsigned)(nslices * (d / RS_TN));
    // K-steps of 64 (whole 128-byte lines per visit, one workgroup per CU) measured 17.9 ms against 13.2 ms for K-steps of
    // 32 (two workgroups per CU) on one GPU's share of C5: occupancy beats line efficiency here.  GHF_RS_K=64 for A/B.
    static const bool k64 = getenv("GHF_RS_K") && atoi(getenv("GHF_RS_K")) == 64;
    // d % 256 == 0: eight waves per workgroup, 256 columns per row tile (every gathered row fetched once per 256 columns);
    // GHF_RS_NCT=1 keeps the four-wave workgroups for A/B
    static const bool nct1 = getenv("GHF_RS_NCT") && atoi(getenv("GHF_RS_NCT")) == 1;
    if ((d % 256) == 0 && k64) {
        constexpr size_t lds = (size_t)2 * 2 * 2 * RS_TM * 64 * 2 + 3 * RS_TM * 4;
        GHF_SET_MAX_LDS((edge_transform_h_kernel<64, 1>), lds);
        edge_transform_h_kernel<64, 1><<<grid, 256, lds, stream>>>((const char*)h_split, N, d, src, dst, ypos, slice_tab,
                                                                 (const char*)w2h, R, bias, (const char*)x_split, NX, row_cnt, Y);
    } else if ((d % 256) == 0 && !nct1 && !(getenv("GHF_RS_P1") && atoi(getenv("GHF_RS_P1")) == 2)) {
        // round 4: tiles by LDS-DMA into a three-deep ring (GHF_RS_P1=2 keeps round 3's register-staged kernel for A/B)
        constexpr size_t lds = (size_t)3 * (2 * RS_TM * 32 * 2 + 2 * 2 * RS_TN * 32 * 2) + 3 * RS_TM * 4;
        GHF_SET_MAX_LDS(edge_transform_h3_kernel, lds);
        edge_transform_h3_kernel<<<grid / 2, 512, lds, stream>>>((const char*)h_split, N, d, src, dst, ypos, slice_tab,
                                                                 (const char*)w2h, R, bias, (const char*)x_split, NX, row_cnt, Y);
    } else if ((d % 256) == 0 && !nct1) {
        constexpr size_t lds = (size_t)2 * 2 * 3 * RS_TM * 32 * 2 + 3 * RS_TM * 4;
        GHF_SET_MAX_LDS((edge_transform_h_kernel<32, 2>), lds);
        edge_transform_h_kernel<32, 2><<<grid / 2, 512, lds, stream>>>((const char*)h_split, N, d, src, dst, ypos, slice_tab,
                                                                     (const char*)w2h, R, bias, (const char*)x_split, NX, row_cnt, Y);
    } else {
        constexpr size_t lds = (size_t)2 * 2 * 2 * RS_TM * 32 * 2 + 3 * RS_TM * 4;
        GHF_SET_MAX_LDS((edge_transform_h_kernel<32, 1>), lds);
        edge_transform_h_kernel<32, 1><<<grid, 256, lds, stream>>>((const char*)h_split, N, d, src, dst, ypos, slice_tab,
                                                                 (const char*)w2h, R, bias, (const char*)x_split, NX, row_cnt, Y);
    }
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_segment_partial(const float* Y, const int64_t* hub_chunks, int64_t nchunks, int d, float* P, hipStream_t stream) {
    GHF_REQUIRE(d >= 1 && d <= RS_MAX_D, "segment_partial: d=%d outside [1,%d]", d, RS_MAX_D);
    if (nchunks <= 0) return GHF_OK;
    GHF_REQUIRE(nchunks < (1ll << 31), "segment_partial: too many chunks");
    segment_partial_kernel<<<(unsigned)nchunks, 256, 0, stream>>>(Y, hub_chunks, d, P);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_segment_tail(const float* Y, const int64_t* off, const int32_t* deg_of, const int32_t* hub_of, const int64_t* hub_tab, const float* P,
                        const float* h, const float* g, const float* b, float eps, int64_t row0, int64_t rows, int d,
                        float* h_out, void* h_split_out, int64_t n_split, int flags, hipStream_t stream) {
    GHF_REQUIRE(message_rs_supported(d), "segment_tail: d = %d is not a relation-stationary width", d);
    if (rows <= 0) return GHF_OK;
    GHF_REQUIRE(cdiv(rows, 4) < (1ll << 31), "segment_tail: too many rows per launch");
    const unsigned grid = (unsigned)cdiv(rows, 4);
    const int nt = flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM);
    switch (d / 64) {
#define GHF_RS_CASE(CPL)                                                                                                        \
    case CPL:                                                                                                                   \
        segment_tail_kernel<CPL><<<grid, 256, 0, stream>>>(Y, off, deg_of, hub_of, hub_tab, P, h, g, b, eps, row0, row0 + rows, h_out,    \
                                                           (char*)h_split_out, n_split, nt, range_flag_ptr());                 \
        break;
        GHF_RS_CASE(2) GHF_RS_CASE(4) GHF_RS_CASE(6) GHF_RS_CASE(8) GHF_RS_CASE(10) GHF_RS_CASE(12) GHF_RS_CASE(14) GHF_RS_CASE(16)
#undef GHF_RS_CASE
    }
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
