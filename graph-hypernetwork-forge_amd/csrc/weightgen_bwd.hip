// Backward of the WeightGenerator's three MLP heads (reference models/weight_generator.py:96-143 under plain autograd:
// demo.py:79-101, tests/test_hypergnn.py:183-226) as THREE launches per generator instead of the ~68 small ones the
// per-operation chain took (ghf_dot, ghf_scale_exp, ghf_relu_mask, ghf_group_outer, ghf_colsum, ghf_transpose_batched per
// head and layer).  At BASELINE config 3 those hide beside the message layers' gradient passes; at configs 1 and 2 — graphs
// whose whole step is a few hundred launch latencies — they were most of the step (config 1: 137 of 210 launches).
//
//   out_k = exp(ls_k) * (a_k W_last_k^T + b_last_k),  a_k = the head's last hidden activation (or text_emb without hidden layers)
//   given g_k = dL/d out_k:
//     dls_k      = sum g_k . out_k
//     dy         = exp(ls_k) g_k                                    [R, D_k]   (D_k = d_in d_out for the matrix heads, d_out for the bias head)
//     dW_last    = dy^T a_k        [D_k, Hl]      db_last = column sums of dy
//     dyh        = dy W_last       [R, Hl]        (the contraction over D_k cut into ranges: partial sums, added in order)
//   then per hidden layer l = nh-1 .. 0 (post-ReLU, post-dropout activations a_l saved by the forward):
//     dy_l = dyh . [a_l > 0] * keep,  db_l = column sums,  dW_l = dy_l^T a_{l-1},  dyh = dy_l W_l
//   and d text_emb = the three heads' last dyh, added.
//
// wgb_out_kernel (grid: ranges of 256 output columns x 3 heads) does the last layer; wgb_hidden_kernel (one workgroup per head)
// adds the partial dyh in range order and walks the hidden layers with the row tile in LDS; ghf_add3 sums the heads' d text_emb.
// Plain fp32 FMAs, every sum in a fixed order: bitwise reproducible.  Widths: max(T, Hh) <= 256 (a thread per column and group
// of rows / output units: 1,024 threads); wider
// generators keep the per-operation chain (autograd.py).
#include "common.h"

#include <string.h>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WGB_W = 256;          // max(T, Hh) of the fused backward
constexpr int WGB_NT = 1024;        // threads per workgroup: NC columns (a power of two covering the width) x 1024 / NC groups
constexpr int WGB_OC = 64;          // output columns per tile of wgb_out_kernel
constexpr int WGB_OR = 256;         // ... per workgroup (its partial dyh covers this range of the contraction)
constexpr int WGB_RT = 64;          // rows per tile of wgb_out_kernel
constexpr int WGB_HT = 32;          // rows per tile of wgb_hidden_kernel
constexpr int WGB_HS = 36;          // LDS row stride of its transposed tile (floats: 16-byte aligned, not a multiple of 32)

struct WgbOutHead {
    const float* g;        // [R, Dk] dL/d out
    const float* out;      // [R, Dk]
    const float* a;        // [R, Hl] the last layer's input
    const float* W;        // [Dk, Hl]
    const float* ls;       // [1] log-scale (device)
    float* dW;             // [Dk, Hl]
    float* db;             // [Dk]
    int Dk;
};
struct WgbOutArgs { WgbOutHead h[3]; };

// part_dyh[((head * nsplit + split) * R + r) * Hl + j], part_dls[head * nsplit + split].
// Thread (j = t % NC, grp = t / NC): column j of the last layer's input; of a 64 x 64 tile of dy the group takes 64 / G output
// columns for dW and 64 / G rows for dyh (G = 1024 / NC groups): every output has one owner, nothing is reduced across threads.
template <int NC>
__global__ __launch_bounds__(WGB_NT) void wgb_out_kernel(WgbOutArgs A, int R, int Hl, int nsplit, float* __restrict__ part_dyh,
                                                         float* __restrict__ part_dls) {
    constexpr int G = WGB_NT / NC, OPG = WGB_OC / G, RPG = WGB_RT / G;
    static_assert(OPG >= 1 && RPG >= 1, "at most 64 groups");
    const WgbOutHead& H = A.h[blockIdx.y];
    const int split = blockIdx.x, o_lo = split * WGB_OR;
    const int t = threadIdx.x, j = t % NC, grp = t / NC;
    __shared__ __attribute__((aligned(16))) float dy_ro[WGB_RT][WGB_OC];       // [r][o]: dW reads a group's o side by side
    __shared__ __attribute__((aligned(16))) float dy_or[WGB_OC][WGB_RT + 4];   // [o][r]: dyh reads a group's rows side by side (+4: the transposed writes spread over banks)
    __shared__ float red[WGB_NT / 64];
    float dls = 0.f;
    if (o_lo < H.Dk) {
        const float s = expf(H.ls[0]);
        const int Dk = H.Dk;
        const bool col = j < Hl;
        for (int rt = 0; rt < R; rt += WGB_RT) {
            const int nr = R - rt < WGB_RT ? R - rt : WGB_RT;
            float acc2[RPG];
#pragma unroll
            for (int r = 0; r < RPG; ++r) acc2[r] = 0.f;
            for (int o0 = o_lo; o0 < o_lo + WGB_OR && o0 < Dk; o0 += WGB_OC) {
                const int no = Dk - o0 < WGB_OC ? Dk - o0 : WGB_OC;
                __syncthreads();
                // the tile of dy = s g (zeros past the edges), and this tile's share of sum g . out
#pragma unroll
                for (int i = 0; i < WGB_RT * WGB_OC / WGB_NT; ++i) {
                    const int e = t + WGB_NT * i, r = e / WGB_OC, o = e % WGB_OC;
                    float v = 0.f;
                    if (r < nr && o < no) {
                        const size_t at = (size_t)(rt + r) * Dk + o0 + o;
                        const float gv = H.g[at];
                        dls += gv * H.out[at];
                        v = gv * s;
                    }
                    dy_ro[r][o] = v;
                    dy_or[o][r] = v;
                }
                __syncthreads();
                if (t < no) {                                                    // db: rows in order
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < WGB_RT; ++r) sum += dy_or[t][r];
                    H.db[o0 + t] = rt == 0 ? sum : H.db[o0 + t] + sum;
                }
                if (col) {
                    // dW[o][j] = sum_r dy[r][o] a[r][j] for the group's o
                    float acc[OPG];
#pragma unroll
                    for (int k = 0; k < OPG; ++k) acc[k] = 0.f;
                    // (unrolled: sixteen independent loads in flight instead of one round trip per row)
#pragma unroll 16
                    for (int r = 0; r < nr; ++r) {
                        const float av = H.a[(size_t)(rt + r) * Hl + j];
#pragma unroll
                        for (int k = 0; k < OPG; ++k) acc[k] += dy_ro[r][grp * OPG + k] * av;
                    }
#pragma unroll
                    for (int k = 0; k < OPG; ++k)
                        if (grp * OPG + k < no) {
                            float* p = H.dW + (size_t)(o0 + grp * OPG + k) * Hl + j;
                            *p = rt == 0 ? acc[k] : *p + acc[k];
                        }
                    // dyh[r][j] += sum_{o in tile} dy[r][o] W[o][j] for the group's rows
#pragma unroll 16
                    for (int o = 0; o < no; ++o) {
                        const float w = H.W[(size_t)(o0 + o) * Hl + j];
#pragma unroll
                        for (int k = 0; k < RPG; ++k) acc2[k] += dy_or[o][grp * RPG + k] * w;
                    }
                }
            }
            if (col) {
                float* P = part_dyh + ((size_t)(blockIdx.y * nsplit + split) * R + rt) * Hl + j;
#pragma unroll
                for (int k = 0; k < RPG; ++k)
                    if (grp * RPG + k < nr) P[(size_t)(grp * RPG + k) * Hl] = acc2[k];
            }
        }
    }
    // sum g . out of this range: lanes, then waves, in a fixed order
    dls = wave_sum(dls);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = dls;
    __syncthreads();
    if (t == 0) {
        float sum = 0.f;
        for (int w = 0; w < WGB_NT / 64; ++w) sum += red[w];
        part_dls[blockIdx.y * nsplit + split] = o_lo < H.Dk ? sum : 0.f;
    }
}

struct WgbHidHead {
    const float* act[7];   // [R, Hh] hidden layer l's output (post-ReLU, post-dropout)
    const float* W[7];     // hidden layer l's weight [Hh, in_l], in_0 = T
    float* dW[7];
    float* db[7];
    float* dls;            // [1]
    float* dx;             // [R, T] this head's share of d text_emb
    int nsplit;            // ranges of wgb_out_kernel that exist for this head
};
struct WgbHidArgs { WgbHidHead h[3]; };

// One workgroup per head; the row tile (32 rows, transposed: [column][row]) in LDS, ping-pong.  Thread (i = t % NC, grp = t / NC):
// column i; of a layer's Hh output units the group takes Hh / G for the weight gradient (all 32 rows), of the tile's rows 32 / G
// for the next dy (all units) — again one owner per output.
template <int NC>
__global__ __launch_bounds__(WGB_NT) void wgb_hidden_kernel(WgbHidArgs A, const float* __restrict__ x, int R, int T, int Hh, int nh,
                                                            int nsplit, const float* __restrict__ part_dyh,
                                                            const float* __restrict__ part_dls, const float* __restrict__ log_keep) {
    constexpr int G = WGB_NT / NC, RPG = WGB_HT / G;
    static_assert(RPG >= 1, "at most 32 groups");
    const WgbHidHead& H = A.h[blockIdx.x];
    const int t = threadIdx.x, i = t % NC, grp = t / NC;
    const int Hl = nh ? Hh : T;
    extern __shared__ __attribute__((aligned(16))) float wgb_lds[];
    float (*tile)[NC][WGB_HS] = (float (*)[NC][WGB_HS])wgb_lds;               // [2][column][row of the tile]
    float (*dbs)[NC] = (float (*)[NC])(wgb_lds + 2 * NC * WGB_HS);             // [7][column]: the bias gradients, summed over row tiles
    if (t == 0) {
        float s = 0.f;
        for (int k = 0; k < H.nsplit; ++k) s += part_dls[blockIdx.x * nsplit + k];
        H.dls[0] = s;
    }
    const float keep = log_keep ? expf(log_keep[0]) : 1.f;
    if (grp == 0)
        for (int l = 0; l < 7; ++l) dbs[l][i] = 0.f;                           // (a column's sums stay with thread (i, 0))
    for (int rt = 0; rt < R; rt += WGB_HT) {
        const int nr = R - rt < WGB_HT ? R - rt : WGB_HT;
        int cur = 0;
        __syncthreads();
        // dyh of the last layer: the ranges' partial sums in range order
#pragma unroll
        for (int k = 0; k < RPG; ++k) {
            const int r = grp * RPG + k;
            float s = 0.f;
            if (i < Hl && r < nr)
#pragma unroll 16
                for (int q = 0; q < H.nsplit; ++q) s += part_dyh[((size_t)(blockIdx.x * nsplit + q) * R + rt + r) * Hl + i];
            tile[cur][i][r] = s;
        }
        __syncthreads();
        for (int l = nh - 1; l >= 0; --l) {
            const int in_l = l == 0 ? T : Hh;
            // through the ReLU (and the dropout mask's scale), in place
#pragma unroll
            for (int k = 0; k < RPG; ++k) {
                const int r = grp * RPG + k;
                if (i < Hh && r < nr) {
                    const float v = tile[cur][i][r];
                    tile[cur][i][r] = H.act[l][(size_t)(rt + r) * Hh + i] > 0.f ? v * keep : 0.f;
                }
            }
            __syncthreads();
            if (grp == 0 && i < Hh) {                                           // bias gradient: rows in order
                float bsum = dbs[l][i];
                for (int r = 0; r < nr; ++r) bsum += tile[cur][i][r];
                dbs[l][i] = bsum;
            }
            const bool on = i < in_l;
            const float* __restrict__ prev = l == 0 ? x + (size_t)rt * T : H.act[l - 1] + (size_t)rt * Hh;
            if (on) {
                // dW_l[j][i] = sum_r dy[r][j] a_prev[r][i] for the group's j
                float ap[WGB_HT];
#pragma unroll
                for (int r = 0; r < WGB_HT; ++r) ap[r] = r < nr ? prev[(size_t)r * in_l + i] : 0.f;
                const int jn = (Hh + G - 1) / G, j0 = grp * jn, j1 = j0 + jn < Hh ? j0 + jn : Hh;
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int r4 = 0; r4 < WGB_HT / 4; ++r4) {
                        const f32x4 d4 = *(const f32x4*)&tile[cur][j][4 * r4];
                        s += d4[0] * ap[4 * r4];
                        s += d4[1] * ap[4 * r4 + 1];
                        s += d4[2] * ap[4 * r4 + 2];
                        s += d4[3] * ap[4 * r4 + 3];
                    }
                    float* p = H.dW[l] + (size_t)j * in_l + i;
                    *p = rt == 0 ? s : *p + s;
                }
            }
            // next dy[r][i] = sum_j dy[r][j] W_l[j][i] for the group's rows
            float acc[RPG];
#pragma unroll
            for (int k = 0; k < RPG; ++k) acc[k] = 0.f;
            if (on)
#pragma unroll 16
                for (int j = 0; j < Hh; ++j) {
                    const float w = H.W[l][(size_t)j * in_l + i];
#pragma unroll
                    for (int k = 0; k < RPG; ++k) acc[k] += tile[cur][j][grp * RPG + k] * w;
                }
#pragma unroll
            for (int k = 0; k < RPG; ++k) tile[cur ^ 1][i][grp * RPG + k] = acc[k];   // (columns past in_l: zeros)
            cur ^= 1;
            __syncthreads();
        }
        if (i < T)
#pragma unroll
            for (int k = 0; k < RPG; ++k)
                if (grp * RPG + k < nr) H.dx[(size_t)(rt + grp * RPG + k) * T + i] = tile[cur][i][grp * RPG + k];
    }
    __syncthreads();
    if (grp == 0 && i < Hh)
        for (int l = 0; l < nh; ++l) H.db[l][i] = dbs[l][i];
}

int weightgen_bwd_supported(int T, int Hh, int num_hidden) {
    return T > 0 && T <= WGB_W && (num_hidden == 0 || (Hh > 0 && Hh <= WGB_W)) && num_hidden >= 0 && num_hidden <= 7;
}

static int wgb_nsplit(int d_in, int d_out) {
    const int64_t dk = (int64_t)d_in * d_out;
    return (int)((dk + WGB_OR - 1) / WGB_OR);
}

size_t weightgen_bwd_workspace_floats(int R, int T, int Hh, int num_hidden, int d_in, int d_out) {
    const size_t Hl = num_hidden ? Hh : T, ns = wgb_nsplit(d_in, d_out);
    return 3 * ns * (size_t)R * Hl + 3 * ns + 3 * (size_t)R * T;
}

int launch_weightgen_bwd(const float* text_emb, const float* const* head_params, const float* acts, const float* const* outs,
                         const float* const* grads, const float* const* log_scales, int R, int T, int Hh, int num_hidden, int d_in,
                         int d_out, const float* log_keep, float* const* dparams, float* const* dls, float* dx, float* workspace,
                         hipStream_t stream) {
    GHF_REQUIRE(weightgen_bwd_supported(T, Hh, num_hidden), "weightgen_bwd: text_dim %d / hidden_dim %d / %d hidden layers have no fused "
                "backward (widths up to %d, at most 7 layers)", T, Hh, num_hidden, WGB_W);
    GHF_REQUIRE(R > 0 && d_in > 0 && d_out > 0, "weightgen_bwd: R, d_in, d_out must be positive");
    const int nl = num_hidden + 1, Hl = num_hidden ? Hh : T, ns = wgb_nsplit(d_in, d_out);
    float* part_dyh = workspace;
    float* part_dls = part_dyh + (size_t)3 * ns * R * Hl;
    float* dxk = part_dls + (size_t)3 * ns;
    WgbOutArgs OA;
    WgbHidArgs HA;
    memset(&HA, 0, sizeof(HA));
    for (int k = 0; k < 3; ++k) {
        GHF_REQUIRE(outs[k] && grads[k] && log_scales[k] && dls[k], "weightgen_bwd: null pointer (head %d)", k);
        WgbOutHead& O = OA.h[k];
        O.g = grads[k];
        O.out = outs[k];
        O.a = num_hidden ? acts + ((size_t)(k * num_hidden + num_hidden - 1) * R) * Hh : text_emb;
        O.W = head_params[(k * nl + nl - 1) * 2];
        O.ls = log_scales[k];
        O.dW = dparams[(k * nl + nl - 1) * 2];
        O.db = dparams[(k * nl + nl - 1) * 2 + 1];
        O.Dk = k == 2 ? d_out : d_in * d_out;
        GHF_REQUIRE(O.W && O.dW && O.db, "weightgen_bwd: null parameter pointer (head %d, output layer)", k);
        WgbHidHead& Hd = HA.h[k];
        for (int l = 0; l < num_hidden; ++l) {
            Hd.act[l] = acts + ((size_t)(k * num_hidden + l) * R) * Hh;
            Hd.W[l] = head_params[(k * nl + l) * 2];
            Hd.dW[l] = dparams[(k * nl + l) * 2];
            Hd.db[l] = dparams[(k * nl + l) * 2 + 1];
            GHF_REQUIRE(Hd.W[l] && Hd.dW[l] && Hd.db[l], "weightgen_bwd: null parameter pointer (head %d, layer %d)", k, l);
        }
        Hd.dls = dls[k];
        Hd.dx = dxk + (size_t)k * R * T;
        Hd.nsplit = (O.Dk + WGB_OR - 1) / WGB_OR;
    }
    GHF_REQUIRE(num_hidden == 0 || acts, "weightgen_bwd: the hidden activations are needed");
    // columns: a power of two covering the width (the last layer's input for the first kernel, every layer's for the second)
    auto cover = [](int w) { int nc = 32; while (nc < w) nc <<= 1; return nc; };
    const dim3 og((unsigned)ns, 3);
    switch (cover(Hl)) {
        case 32: wgb_out_kernel<32><<<og, WGB_NT, 0, stream>>>(OA, R, Hl, ns, part_dyh, part_dls); break;
        case 64: wgb_out_kernel<64><<<og, WGB_NT, 0, stream>>>(OA, R, Hl, ns, part_dyh, part_dls); break;
        case 128: wgb_out_kernel<128><<<og, WGB_NT, 0, stream>>>(OA, R, Hl, ns, part_dyh, part_dls); break;
        default: wgb_out_kernel<256><<<og, WGB_NT, 0, stream>>>(OA, R, Hl, ns, part_dyh, part_dls); break;
    }
    GHF_LAUNCH_CHECK();
    const int nc = cover(num_hidden ? (T > Hh ? T : Hh) : T);
    const size_t hid_lds = (size_t)(2 * nc * WGB_HS + 7 * nc) * sizeof(float);
#define GHF_WGB_HIDDEN(NC)                                                                                                      \
    do {                                                                                                                        \
        GHF_SET_MAX_LDS(wgb_hidden_kernel<NC>, hid_lds);                                                                        \
        wgb_hidden_kernel<NC><<<3, WGB_NT, hid_lds, stream>>>(HA, text_emb, R, T, Hh, num_hidden, ns, part_dyh, part_dls, log_keep); \
    } while (0)
    switch (nc) {
        case 32: GHF_WGB_HIDDEN(32); break;
        case 64: GHF_WGB_HIDDEN(64); break;
        case 128: GHF_WGB_HIDDEN(128); break;
        default: GHF_WGB_HIDDEN(256); break;
    }
#undef GHF_WGB_HIDDEN
    GHF_LAUNCH_CHECK();
    if (dx) return launch_add3(dxk, dxk + (size_t)R * T, dxk + (size_t)2 * R * T, (int64_t)R * T, dx, stream);
    return GHF_OK;
}

}  // namespace ghf
