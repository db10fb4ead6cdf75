// backward.hip — gradients of the hot path: the pieces that are not the message kernel itself.
//
// The reference trains through these ops with autograd (demo.py:79-101, tests/test_hypergnn.py:183-226).  With
//   out_v = (1/c_v) sum_{e=(u->v)} (h_u Wm[r_e] + b[r_e] + h_v Ws[r_e]),   x_v = relu(out_v + h_v),   h'_v = LN(x_v)
// and g' = dL/dh':
//   tail_bwd      : dpre = LN'(x) g' * [pre > 0]  (pre = out + h);  G_v = dpre_v / c_v;  T = g' * xhat (for dgamma)
//   colsum        : dbeta = colsum(g'), dgamma = colsum(T), bias gradients of the linear layers        (deterministic)
//   group_outer   : dWm[r] = sum_{e in r} h_u^T G_v,  dWs[r] = sum h_v^T G_v,  db[r] = sum G_v  — and the input projection's
//                   and the weight generator's weight gradients, which are the same contraction over rows
//   the gradient with respect to h is two more passes of the forward message kernel with transposed weights
//   (GHF_FLAG_RAW_SUM): sum_{e->v} G_v Ws[r]^T on the forward plan, sum_{e: src=u} G_v Wm[r]^T on the reversed one.
// First version: exact fp32 (v_mfma_f32_16x16x4_f32 / vector ALU), deterministic summation orders, any d <= 1024.
#include "common.h"
#include <type_traits>

#include <stdlib.h>
#include <string.h>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BW_MAX_D = 1024;
constexpr int BW_PER_LANE = BW_MAX_D / 64;

// ---- tail_bwd -----------------------------------------------------------------------------------------------------------
// A fixed grid of workgroups walks the rows (wave w of workgroup b: row groups (4 b + w) + k * 4 * gridDim.x), so that the two
// column sums dgamma = sum_v g'_v * xhat_v and dbeta = sum_v g'_v are accumulated in registers on the way — every wave over
// its rows in order, the four waves of a workgroup in order through LDS, the workgroups' partial rows [gridDim.x][2][d] by
// ghf_colsum's tree — instead of writing g' * xhat out and reading it and g' again.  G can leave in the two-fp16-piece
// form as well (the gradient passes gather it that way): the same values ghf_split_rows would cut from the fp32 G.
constexpr int TB_MAX_BLOCKS = 2048;
typedef _Float16 tb_f16x4 __attribute__((ext_vector_type(4)));

// sum / max over the groups of LPR consecutive lanes (LPR a power of two); every lane of a group gets the same bits
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LPR >= 2) v += dpp_take<0xB1, 0xF>(v);
    if constexpr (LPR >= 4) v += dpp_take<0x4E, 0xF>(v);
    if constexpr (LPR >= 8) v += dpp_take<0x141, 0xF>(v);
    if constexpr (LPR >= 16) v += dpp_take<0x140, 0xF>(v);
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16);
    if constexpr (LPR >= 64) v += __shfl_xor(v, 32);
    return v;
}
template <int LPR>
__device__ __forceinline__ float group_absmax(float v) {
    if constexpr (LPR >= 2) v = fmaxf(v, dpp_take<0xB1, 0xF>(v));
    if constexpr (LPR >= 4) v = fmaxf(v, dpp_take<0x4E, 0xF>(v));
    if constexpr (LPR >= 8) v = fmaxf(v, dpp_take<0x141, 0xF>(v));
    if constexpr (LPR >= 16) v = fmaxf(v, dpp_take<0x140, 0xF>(v));
    if constexpr (LPR >= 32) v = fmaxf(v, __shfl_xor(v, 16));
    if constexpr (LPR >= 64) v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// d = 4 * LPR * NV: a row is NV float4 per lane of a group of LPR lanes, a wave visits 64 / LPR consecutive rows at a time
template <int LPR, int NV>
__global__ __launch_bounds__(256) void tail_bwd_v4_kernel(const float* __restrict__ g_out, const float* __restrict__ agg,
                                                          const float* __restrict__ h, const float* __restrict__ gamma, float eps,
                                                          const int32_t* __restrict__ indeg, int64_t N,
                                                          float* __restrict__ dpre, float* __restrict__ G, char* __restrict__ Gs,
                                                          float* __restrict__ part, const float* __restrict__ drop,
                                                          int32_t* __restrict__ range_flag) {
    constexpr int D = 4 * LPR * NV, RPW = 64 / LPR;
    __shared__ float red[4][2][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, cl = lane % LPR;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gam[NV], dg[NV], db[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gam[j] = *(const f32x4*)(gamma + 4 * (j * LPR + cl));
        dg[j] = db[j] = zero4;
    }
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * RPW; base < N; base += stride) {
        const bool ok = base + sub < N;
        const int64_t v = ok ? base + sub : N - 1;                 // (a short last visit: the spare groups redo the last row, unstored)
        const size_t row = (size_t)v * D;
        f32x4 pre[NV], x[NV], dm[NV], go[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const size_t o = row + 4 * (j * LPR + cl);
            pre[j] = *(const f32x4*)(agg + o) + *(const f32x4*)(h + o);
            go[j] = *(const f32x4*)(g_out + o);
            dm[j] = drop ? *(const f32x4*)(drop + o) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x[j][k] = fmaxf(pre[j][k], 0.f) * dm[j][k];
                s += x[j][k];
            }
        }
        const float mean = group_sum<LPR>(s) / (float)D;
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = x[j][k] - mean; var += t * t; }
        const float rstd = 1.0f / sqrtf(group_sum<LPR>(var) / (float)D + eps);
        float s1 = 0.f, s2 = 0.f;                                  // mean(gamma g'), mean(gamma g' xhat)
        f32x4 gg[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (x[j][k] - mean) * rstd;
                gg[j][k] = go[j][k] * gam[j][k];
                s1 += gg[j][k];
                s2 += gg[j][k] * xh;
                if (ok) { dg[j][k] += go[j][k] * xh; db[j][k] += go[j][k]; }
                x[j][k] = xh;
            }
        s1 = group_sum<LPR>(s1) / (float)D;
        s2 = group_sum<LPR>(s2) / (float)D;
        const int deg = indeg[v];
        const float inv = 1.0f / (float)(deg > 1 ? deg : 1);
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            f32x4 dp, gv;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dx = rstd * (gg[j][k] - s1 - x[j][k] * s2);
                dp[k] = pre[j][k] > 0.f ? dx * dm[j][k] : 0.f;
                gv[k] = dp[k] * inv;
                mx = fmaxf(mx, fabsf(gv[k]));
            }
            if (ok) {
                *(f32x4*)(dpre + row + 4 * (j * LPR + cl)) = dp;
                *(f32x4*)(G + row + 4 * (j * LPR + cl)) = gv;
            }
            x[j] = gv;
        }
        if (Gs) {                                                  // as split2h_rows_kernel (message_hx.hip) cuts a row
            const int sh = split2h_shift(group_absmax<LPR>(mx));
            const float up = pow2f(sh);
            float tiny = 0.f, nz = 0.f;
            _Float16* __restrict__ dst = (_Float16*)(Gs + (size_t)v * (4 * (size_t)D));
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                tb_f16x4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xs = x[j][k] * up;
                    _Float16 a, b;
                    split2h(xs, a, b);
                    hi[k] = a;
                    lo[k] = b;
                    tiny += range_tiny(xs) ? 1.f : 0.f;
                    nz += xs != 0.f ? 1.f : 0.f;
                }
                if (ok) {
                    *(tb_f16x4*)(dst + 4 * (j * LPR + cl)) = hi;
                    *(tb_f16x4*)(dst + D + 4 * (j * LPR + cl)) = lo;
                }
            }
            tiny = group_sum<LPR>(tiny);
            nz = group_sum<LPR>(nz);
            if (ok && cl == 0) {
                *(float*)(Gs + (size_t)N * (4 * (size_t)D) + (size_t)v * 4) = pow2f(-sh);
                range_raise(range_flag, GHF_RANGE_ROWS, (int)tiny, (int)nz);
            }
        }
    }
    // the wave's row groups, then the workgroup's waves, in order
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = dg[j][k], b = db[j][k];
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                a += __shfl_xor(a, o);
                b += __shfl_xor(b, o);
            }
            if (sub == 0) {
                red[wave][0][4 * (j * LPR + cl) + k] = a;
                red[wave][1][4 * (j * LPR + cl) + k] = b;
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const float* r = &red[0][0][0] + i;
        part[(size_t)blockIdx.x * (2 * D) + i] = ((r[0] + r[2 * D]) + r[4 * D]) + r[6 * D];
    }
}

// any d <= BW_MAX_D: one wave per row, columns lane + 64 c
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ agg,
                                                       const float* __restrict__ h, const float* __restrict__ gamma, float eps,
                                                       const int32_t* __restrict__ indeg, int64_t N, int d,
                                                       float* __restrict__ dpre, float* __restrict__ G, float* __restrict__ part,
                                                       const float* __restrict__ drop) {
    extern __shared__ float tb_red[];                            // [4][2][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[BW_PER_LANE], db[BW_PER_LANE];
#pragma unroll
    for (int c = 0; c < BW_PER_LANE; ++c) dg[c] = db[c] = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < N; v += (int64_t)gridDim.x * 4) {
        float pre[BW_PER_LANE], x[BW_PER_LANE], gg[BW_PER_LANE], dm[BW_PER_LANE];   // dm: the dropout mask (scaled; 1 without dropout)
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < BW_PER_LANE; ++c) {
            const int o = lane + 64 * c;
            pre[c] = x[c] = gg[c] = 0.f;
            dm[c] = 1.f;
            if (o < d) {
                pre[c] = agg[(size_t)v * d + o] + h[(size_t)v * d + o];
                if (drop) dm[c] = drop[(size_t)v * d + o];
                x[c] = fmaxf(pre[c], 0.f) * dm[c];
                s += x[c];
            }
        }
        const float mean = wave_sum(s) / (float)d;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < BW_PER_LANE; ++c)
            if (lane + 64 * c < d) { const float t = x[c] - mean; var += t * t; }
        const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)d + eps);
        float s1 = 0.f, s2 = 0.f;                            // mean(gamma g'), mean(gamma g' xhat)
#pragma unroll
        for (int c = 0; c < BW_PER_LANE; ++c) {
            const int o = lane + 64 * c;
            if (o < d) {
                const float go = g_out[(size_t)v * d + o], xh = (x[c] - mean) * rstd;
                gg[c] = go * gamma[o];
                s1 += gg[c];
                s2 += gg[c] * xh;
                dg[c] += go * xh;
                db[c] += go;
                x[c] = xh;
            }
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
        const int deg = indeg[v];
        const float inv = 1.0f / (float)(deg > 1 ? deg : 1);
#pragma unroll
        for (int c = 0; c < BW_PER_LANE; ++c) {
            const int o = lane + 64 * c;
            if (o < d) {
                const float dx = rstd * (gg[c] - s1 - x[c] * s2);
                const float dp = pre[c] > 0.f ? dx * dm[c] : 0.f;
                dpre[(size_t)v * d + o] = dp;
                G[(size_t)v * d + o] = dp * inv;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < BW_PER_LANE; ++c) {
        const int o = lane + 64 * c;
        if (o < d) {
            tb_red[(wave * 2 + 0) * d + o] = dg[c];
            tb_red[(wave * 2 + 1) * d + o] = db[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d; i += 256)
        part[(size_t)blockIdx.x * (2 * d) + i] = ((tb_red[i] + tb_red[2 * d + i]) + tb_red[4 * d + i]) + tb_red[6 * d + i];
}

// out[o] = sum_v X[v][o] * (mask ? (mask[v][o] > 0) : 1): a tree of fixed shape.  One pass cuts the rows into runs of
// CS_ROWS; a workgroup covers a run and up to 256 column vectors (VW floats each): thread (rsub, c) walks rows
// rsub, rsub + nsub, ... of its columns, and the nsub row lanes are added in order through LDS.  Passes repeat on the
// partial sums until one workgroup per column tile is left.
constexpr int CS_ROWS = 512;                              // rows per workgroup and pass
template <int VW>
__global__ __launch_bounds__(256) void colsum_pass_kernel(const float* __restrict__ X, const float* __restrict__ mask,
                                                          int64_t N, int d, float* __restrict__ out, int accumulate) {
    typedef float vec __attribute__((ext_vector_type(VW)));
    __shared__ float red[256 * VW];
    const int dv = d / VW;                                // column vectors per row
    const int cols = dv - (int)blockIdx.y * 256 < 256 ? dv - (int)blockIdx.y * 256 : 256;   // ... of this tile
    const int nsub = 256 / cols;
    const int c = threadIdx.x % cols, rsub = threadIdx.x / cols;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
    const int64_t r1 = r0 + CS_ROWS < N ? r0 + CS_ROWS : N;
    const size_t col = ((size_t)blockIdx.y * 256 + c) * VW;
    vec s;
#pragma unroll
    for (int k = 0; k < VW; ++k) s[k] = 0.f;
    if (rsub < nsub)
        for (int64_t r = r0 + rsub; r < r1; r += nsub) {
            const vec xv = *(const vec*)(X + (size_t)r * d + col);
            if (mask) {
                const vec mv = *(const vec*)(mask + (size_t)r * d + col);
#pragma unroll
                for (int k = 0; k < VW; ++k) s[k] += mv[k] > 0.f ? xv[k] : 0.f;
            } else {
                s += xv;
            }
        }
#pragma unroll
    for (int k = 0; k < VW; ++k) red[threadIdx.x * VW + k] = s[k];
    __syncthreads();
    if (rsub == 0) {
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            float tot = 0.f;
            for (int j = 0; j < nsub; ++j) tot += red[(j * cols + c) * VW + k];
            float* dst = out + (size_t)blockIdx.x * d + col + k;
            *dst = accumulate ? *dst + tot : tot;
        }
    }
}

// dmask[v][o] = X[v][o] * (ref[v][o] > 0)   (ReLU backward as a matrix, for the contractions below)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ X, const float* __restrict__ ref, int64_t n,
                                                        float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = ref[i] > 0.f ? X[i] : 0.f;
}

// C[g][i][o] (+)= sum_{e in group g} A[ia[e]][i] * B[ib[e]][o],  group g = the range gstart[g] .. gend[g] of e.
// One workgroup per (group, 16-row tile of i): its 4 waves take contiguous quarters of the group's edges, every wave
// runs v_mfma_f32_16x16x4_f32 over 4 edges per step (A fragment: lane (i = l&15, k = l>>4) = A[ia[e+k]][i0+i];
// B fragment: B[ib[e+k]][16t + (l&15)]), and the four partial tiles are added in wave order through LDS: the
// summation order is fixed.  da == 0 means A = 1 (column sums of the gathered rows).
template <int NTB>   // 16-column tiles of B handled per pass (db <= 16*NTB per pass)
__global__ __launch_bounds__(256) void group_outer_kernel(const float* __restrict__ A, const int64_t* __restrict__ ia, int da,
                                                          const float* __restrict__ B, const int64_t* __restrict__ ib, int db,
                                                          const int64_t* __restrict__ gstart, const int64_t* __restrict__ gend,
                                                          float* __restrict__ Cout, int o_base, int accumulate) {
    __shared__ float red[4][16][16 * NTB + 1];
    const int g = blockIdx.x, it = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int64_t e0 = gstart[g], e1 = gend[g];
    const int64_t per = ((e1 - e0 + 3) / 4 + 3) & ~(int64_t)3;            // edges per wave, a multiple of 4
    const int64_t w0 = e0 + w * per, w1 = (w0 + per < e1) ? w0 + per : e1;
    const int i = it * 16 + c16;
    const int rows_a = da > 0 ? da : 1;
    f32x4 acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int64_t e = w0;
    if (da > 0 && it * 16 + 16 <= da && o_base + 16 * NTB <= db) {      // whole tiles (a wave-uniform test): four steps' loads in flight, the same chain of MFMAs
        for (; e + 16 <= w1; e += 16) {
            float a[4], b[4][NTB];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t ek = e + 4 * u + q;
                a[u] = A[(size_t)(ia ? ia[ek] : ek) * da + i];
                const float* __restrict__ br = B + (size_t)(ib ? ib[ek] : ek) * db + o_base + c16;
#pragma unroll
                for (int t = 0; t < NTB; ++t) b[u][t] = br[16 * t];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < NTB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][t], acc[t], 0, 0, 0);
        }
    }
    for (; e < w1; e += 4) {
        const int64_t ek = e + q;                         // this lane's edge for both fragments
        const bool ok = ek < w1;
        float a = 0.f;
        if (ok && i < rows_a) a = da > 0 ? A[(size_t)(ia ? ia[ek] : ek) * da + i] : 1.0f;
        const int64_t rb = ok ? (ib ? ib[ek] : ek) : 0;
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            const int o = o_base + 16 * t + c16;
            const float b = (ok && o < db) ? B[(size_t)rb * db + o] : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
    }
    // D: lane holds rows 4q + s, column 16t + c16
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) red[w][4 * q + s][16 * t + c16] = acc[t][s];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * 16 * NTB; idx += 256) {
        const int r = idx / (16 * NTB), c = idx % (16 * NTB);
        const int ii = it * 16 + r, o = o_base + c;
        if (ii < rows_a && o < db) {
            const float s = ((red[0][r][c] + red[1][r][c]) + red[2][r][c]) + red[3][r][c];
            float* dst = Cout + ((size_t)g * rows_a + ii) * db + o;
            *dst = accumulate ? *dst + s : s;
        }
    }
}

// ---- per-relation weight gradients of a whole layer in one pass -------------------------------------------------------
//   dW[r] = sum_{e in r} [h_src(e) | h_dst(e)]^T G_dst(e)    ([2D, D]: dWm on top of dWs),   db[r] = sum_{e in r} G_dst(e)
// over edges grouped by relation and cut into slices (slice_tab: relation, first edge, end edge; a slice never crosses a
// relation).  One workgroup per slice: tiles of ET edges are gathered into LDS as rows [h_src | h_dst] and G_dst (double
// buffered, the next tile's rows in flight during this tile's MFMAs), and the workgroup keeps the whole [2D, D] partial
// product in accumulators: wave (rg, cg) owns 64 x 64 of it as 4 x 4 tiles of v_mfma_f32_16x16x4_f32 whose operands are
// single ds_read_b128 per k-step (lane (i, k) reads columns 4i..4i+3 of row k: register a feeds the tile of rows
// {4i + a}).  Exact fp32; partial products are summed per relation in slice order by edge_outer_reduce_kernel.
// Range guard of the one-scale-per-tensor pieces (ghf.h: ghf_edge_outer_scaled).  A row whose largest magnitude lies 2^-15
// or more below the tensor's sits, after the tensor's scale, below 0.5 throughout — the two fp16 pieces then keep fewer
// than 22 bits of it (common.h: range_tiny).  That alone is harmless and common: a training step's G has 27 - 47 % such rows
// at BASELINE config 3 (the nodes the loss does not touch: their gradient arrives through two layers of 0.01-scale weights), and
// what they add to a relation's sum is 2^-15 of what its other rows add.  It matters when the rows that SET the scale are
// few — an outlier row 2^20 above everything else leaves every relation that does not touch it a sum of far-down rows only.
// cnt[2 y] += far-down rows of tensor y, cnt[2 y + 1] += its nonzero rows; the contraction kernels read the four counters:
// eo_wide() — at least 7/8 of either tensor's nonzero rows that far down — sends the call to the exact fp32 chain.  (Row scales
// are 2^-s(row), s = split2h_shift(the row's largest magnitude); a zero row's shift is the clamp, 100.)
__global__ __launch_bounds__(256) void rowscale_guard_kernel(const float* __restrict__ s0, const float* __restrict__ s1, int64_t n,
                                                             const unsigned* __restrict__ amax, int* __restrict__ cnt) {
    const float* __restrict__ sc = blockIdx.y ? s1 : s0;
    const int st = split2h_shift(__uint_as_float(amax[blockIdx.y]));
    int tiny = 0, nz = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int sr = 127 - (int)((__float_as_uint(sc[i]) >> 23) & 255);       // the row's shift
        if (sr < 100) {
            ++nz;
            tiny += sr - st >= 15;
        }
    }
    tiny = (int)wave_sum((float)tiny);
    nz = (int)wave_sum((float)nz);
    if ((threadIdx.x & 63) == 0 && nz) {
        if (tiny) atomicAdd(cnt + 2 * blockIdx.y, tiny);
        atomicAdd(cnt + 2 * blockIdx.y + 1, nz);
    }
}
__device__ __forceinline__ bool eo_wide(const int* __restrict__ cnt) {
    return cnt && ((cnt[0] > 0 && (int64_t)cnt[0] * 8 >= (int64_t)cnt[1] * 7) || (cnt[2] > 0 && (int64_t)cnt[2] * 8 >= (int64_t)cnt[3] * 7));
}

constexpr int EO_ET = 32;                                 // edges per tile

template <int D>
// Wider rows (d % 128 == 0) run the D = 128 instance once per [256, 128] tile of the [2d, d] gradient: ia / ib name the
// rows (sources or destinations) and xa_col / xb_col the two 128-column pieces of them that make the tile's 256 rows, g_col
// the tile's 128 columns of G (always indexed by destination), ld = d the row stride of h and G.
__device__ __forceinline__ void edge_outer_slice(
    const float* __restrict__ h, const float* __restrict__ G, const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
    const int64_t* __restrict__ dst, int xa_col, int xb_col, int g_col, int ld,
    const int64_t* __restrict__ slice_tab, float* __restrict__ partial, float* __restrict__ partial_b, const size_t slice) {
    constexpr int RG = 2 * D / 64, CG = D / 64, NT = RG * CG * 64;
    constexpr int F4 = D / 4;                             // float4 per row of h / G
    constexpr int LPR = EO_ET * F4 / NT;                  // rows x float4 each thread moves per region and tile
    constexpr int QN = NT / D, RQ = EO_ET / QN;           // db: QN groups of threads, RQ tile rows each
    static_assert(LPR >= 1 && EO_ET * F4 % NT == 0 && NT % D == 0 && EO_ET % QN == 0, "tile does not divide");
    extern __shared__ float eo_lds[];
    float* Xt = eo_lds;                                   // [2][ET][2D]
    float* Gt = eo_lds + 2 * EO_ET * 2 * D;               // [2][ET][D]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int rg = w / CG, cg = w % CG;
    const int c16 = lane & 15, q = lane >> 4;
    const int64_t e0 = slice_tab[3 * slice + 1], e1 = slice_tab[3 * slice + 2];
    const int ntiles = (int)((e1 - e0 + EO_ET - 1) / EO_ET);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (every load unconditional — rows past the slice's end read its last edge's and are zeroed on the way to LDS: branches
    // around the loads put each group of them behind an s_waitcnt vmcnt(0), see edge_outer_h_kernel)
    int64_t is[LPR], iv[LPR], id[LPR];                    // indices of the tile to gather next
    bool okn[LPR], okc[LPR];                              // ... which of its rows exist; the same for the rows in st
    f32x4 st[3][LPR];                                     // its rows on their way to LDS
    auto load_idx = [&](int tile) {
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const int64_t e = e0 + (int64_t)tile * EO_ET + (t + NT * j) / F4;
            okn[j] = e < e1;
            const int64_t ec = okn[j] ? e : e1 - 1;
            is[j] = ia[ec];
            iv[j] = ib[ec];
            id[j] = dst[ec];
        }
    };
    auto gather = [&]() {
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const int c4 = (t + NT * j) % F4;
            okc[j] = okn[j];
            st[0][j] = *(const f32x4*)(h + (size_t)is[j] * ld + xa_col + 4 * c4);
            st[1][j] = *(const f32x4*)(h + (size_t)iv[j] * ld + xb_col + 4 * c4);
            st[2][j] = *(const f32x4*)(G + (size_t)id[j] * ld + g_col + 4 * c4);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const int row = (t + NT * j) / F4, c4 = (t + NT * j) % F4;
            float* xr = Xt + ((size_t)buf * EO_ET + row) * 2 * D;
            *(f32x4*)(xr + 4 * c4) = okc[j] ? st[0][j] : zero4;
            *(f32x4*)(xr + D + 4 * c4) = okc[j] ? st[1][j] : zero4;
            *(f32x4*)(Gt + ((size_t)buf * EO_ET + row) * D + 4 * c4) = okc[j] ? st[2][j] : zero4;
        }
    };

    load_idx(0);
    gather();
    load_idx(1);                                          // (past the slice: every lane reads nothing)
    commit(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        gather();                                         // tile + 1 (indices were loaded an iteration ago)
        load_idx(tile + 2);
        const float* Xb = Xt + (size_t)buf * EO_ET * 2 * D + rg * 64 + 4 * c16;
        const float* Gb = Gt + (size_t)buf * EO_ET * D + cg * 64 + 4 * c16;
#pragma unroll
        for (int ks = 0; ks < EO_ET / 4; ++ks) {
            const f32x4 a = *(const f32x4*)(Xb + (size_t)(4 * ks + q) * 2 * D);
            const f32x4 b = *(const f32x4*)(Gb + (size_t)(4 * ks + q) * D);
#pragma unroll
            for (int ai = 0; ai < 4; ++ai)
#pragma unroll
                for (int bi = 0; bi < 4; ++bi)
                    acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ai], b[bi], acc[ai][bi], 0, 0, 0);
        }
        {
            const float* gcol = Gt + ((size_t)buf * EO_ET + (t / D) * RQ) * D + (t % D);
#pragma unroll
            for (int r = 0; r < RQ; ++r) bsum += gcol[(size_t)r * D];
        }
        commit(buf ^ 1);
        __syncthreads();
    }
    // partial product: tile (ai, bi) register s is row rg*64 + 4*(4q + s) + ai, column cg*64 + 4*c16 + bi
    float* P = partial + slice * 2 * D * D;
#pragma unroll
    for (int ai = 0; ai < 4; ++ai)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = rg * 64 + 4 * (4 * q + s) + ai;
            *(f32x4*)(P + (size_t)row * D + cg * 64 + 4 * c16) = (f32x4){acc[ai][0][s], acc[ai][1][s], acc[ai][2][s], acc[ai][3][s]};
        }
    float* red = eo_lds;                                  // [QN][D] (every tile read is behind the loop's last barrier)
    red[t] = bsum;
    __syncthreads();
    if (t < D) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < QN; ++k) s += red[k * D + t];
        partial_b[slice * D + t] = s;
    }
}

template <int D>
__global__ __launch_bounds__((2 * D / 64) * (D / 64) * 64) void edge_outer_kernel(
    const float* __restrict__ h, const float* __restrict__ G, const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
    const int64_t* __restrict__ dst, int xa_col, int xb_col, int g_col, int ld,
    const int64_t* __restrict__ slice_tab, float* __restrict__ partial, float* __restrict__ partial_b,
    const int32_t* __restrict__ order) {
    edge_outer_slice<D>(h, G, ia, ib, dst, xa_col, xb_col, g_col, ld, slice_tab, partial, partial_b,
                        order ? (size_t)order[blockIdx.x] : (size_t)blockIdx.x);
}

// The range guard's fallback (ghf.h: ghf_edge_outer_scaled): the same slices on the exact chain when eo_wide(guard) says so —
// a small grid whose workgroups walk the slices, so that the launch costs next to nothing in the usual case (every workgroup
// returns at once) instead of dispatching one 98 KB workgroup per slice to do so.
template <int D>
__global__ __launch_bounds__((2 * D / 64) * (D / 64) * 64) void edge_outer_guarded_kernel(
    const float* __restrict__ h, const float* __restrict__ G, const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
    const int64_t* __restrict__ dst, int xa_col, int xb_col, int g_col, int ld,
    const int64_t* __restrict__ slice_tab, int64_t nslices, float* __restrict__ partial, float* __restrict__ partial_b,
    const int* __restrict__ guard) {
    if (!eo_wide(guard)) return;
    for (int64_t slice = blockIdx.x; slice < nslices; slice += gridDim.x) {
        edge_outer_slice<D>(h, G, ia, ib, dst, xa_col, xb_col, g_col, ld, slice_tab, partial, partial_b, (size_t)slice);
        __syncthreads();
    }
}

// dW[r] tile = sum of its slices' partial products in slice order (x: float4 of the [2D, D] tile, y: relation); db likewise.
// The tile sits at (row0, col0) of the relation's [2d, d] matrix (row0 = col0 = 0, D = d for the single-tile sizes).
__global__ __launch_bounds__(256) void edge_outer_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_b,
                                                                const int64_t* __restrict__ slice_off, int D, int d, int row0,
                                                                int col0, float* __restrict__ dW, float* __restrict__ db) {
    const int r = blockIdx.y;
    const int64_t s0 = slice_off[r], s1 = slice_off[r + 1];
    const int n4 = 2 * D * D / 4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int64_t k = s0; k < s1; ++k) s += *(const f32x4*)(partial + (size_t)k * 2 * D * D + 4 * (size_t)i);
        const int row = (4 * i) / D, col = (4 * i) % D;
        *(f32x4*)(dW + ((size_t)r * 2 * d + row0 + row) * d + col0 + col) = s;
    } else if (i - n4 < D && db) {
        float s = 0.f;
        for (int64_t k = s0; k < s1; ++k) s += partial_b[(size_t)k * D + (i - n4)];
        db[(size_t)r * d + col0 + (i - n4)] = s;
    }
}

template <int D>
static int edge_outer_launch(const float* h, const float* G, const int64_t* ia, const int64_t* ib, const int64_t* dst, int xa_col,
                             int xb_col, int g_col, int ld, const int64_t* slice_tab, int64_t nslices, float* partial,
                             float* partial_b, const int32_t* order, hipStream_t stream, const int* guard = nullptr) {
    if (guard) {
        constexpr int NTG = (2 * D / 64) * (D / 64) * 64;
        const size_t ldsg = (size_t)2 * EO_ET * 3 * D * sizeof(float);
        GHF_SET_MAX_LDS(edge_outer_guarded_kernel<D>, ldsg);
        edge_outer_guarded_kernel<D><<<(unsigned)(nslices < 256 ? nslices : 256), NTG, ldsg, stream>>>(h, G, ia, ib, dst, xa_col, xb_col, g_col, ld,
                                                                                                       slice_tab, nslices, partial, partial_b, guard);
        GHF_LAUNCH_CHECK();
        return GHF_OK;
    }
    constexpr int NT = (2 * D / 64) * (D / 64) * 64;
    const size_t lds = (size_t)2 * EO_ET * 3 * D * sizeof(float);
    GHF_SET_MAX_LDS(edge_outer_kernel<D>, lds);
    edge_outer_kernel<D><<<(unsigned)nslices, NT, lds, stream>>>(h, G, ia, ib, dst, xa_col, xb_col, g_col, ld, slice_tab, partial,
                                                                 partial_b, order);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

// ---- the same gradients on the 16-bit matrix pipe (d % 128 == 0) ------------------------------------------------------
// Both operands of dW = X^T G have the contraction index — the edge — as their ROW index in memory, and a
// v_mfma_f32_16x16x32_f16 operand wants eight consecutive k per lane: a transpose.  gfx950's ds_read_b64_tr_b16 does it on
// the way out of LDS (a 4-row x 16-column block of 16-bit elements per 16 lanes, delivered column-major), so the tiles
// are stored row-major [edge][feature] as the gather delivers them — cut into two fp16 pieces with ONE power-of-two scale
// per tensor (the largest magnitude of h resp. G lifted into [2^13, 2^14): a per-row scale cannot leave a sum over rows) —
// and read back as fragments: hi*hi + hi*lo + lo*hi in fp32, 22 significand bits relative to the tensor's largest
// entries (norm-wise the accuracy of the fp32 chain; entries far below the tensor's largest lose bits, which a sum over
// thousands of edges does not see).  5.3x less matrix time than v_mfma_f32_16x16x4_f32.
// LDS image of a [32 edges][128 features] fp16 tile, 256-byte rows: chunk (16 bytes) ch of row r at
//   256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3)))        (conflict-free for the transposed reads; cdna guide, T10)
typedef _Float16 eo_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 eo_f16x8 __attribute__((ext_vector_type(8)));
typedef short eo_s16x4 __attribute__((ext_vector_type(4)));

// out[y] = bits of max |x_y[i]| for the two tensors y = blockIdx.y (a maximum: any order gives the same bits)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int64_t n,
                                                     unsigned* __restrict__ out) {
    const float* __restrict__ x = blockIdx.y ? x1 : x0;
    const int64_t n4 = ((uintptr_t)x & 15) == 0 ? n / 4 : 0, stride = (int64_t)gridDim.x * 256;
    const f32x4* __restrict__ x4 = (const f32x4*)x;
    f32x4 m4 = {0.f, 0.f, 0.f, 0.f};
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {            // four 16-byte loads in flight per lane
        const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], e = x4[i + 3 * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) m4[k] = fmaxf(fmaxf(m4[k], fmaxf(fabsf(a[k]), fabsf(b[k]))), fmaxf(fabsf(c[k]), fabsf(e[k])));
    }
    for (; i < n4; i += stride) {
        const f32x4 a = x4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) m4[k] = fmaxf(m4[k], fabsf(a[k]));
    }
    float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) m = fmaxf(m, fabsf(x[j]));
    m = wave_absmax(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out + blockIdx.y, __float_as_uint(m));   // (non-negative floats order as their bits)
}

// The same two words from the row scales of the tensors' split forms (ghf_split_rows: one float 2^-s(row) per row, s =
// split2h_shift(the row's largest magnitude)): 2^13 max_row 2^-s(row) = 2^e of the tensor's largest magnitude — the one thing
// edge_outer_h_kernel takes from its maximum (split2h_shift reads the exponent only).  n floats per tensor instead of n d.
__global__ __launch_bounds__(256) void rowscale_absmax_kernel(const float* __restrict__ s0, const float* __restrict__ s1, int64_t n,
                                                              unsigned* __restrict__ out) {
    const float* __restrict__ sc = blockIdx.y ? s1 : s0;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, sc[i]);
    m = wave_absmax(m) * 8192.0f;
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out + blockIdx.y, __float_as_uint(m));
}

#ifndef GHF_EO_SRC_LAST
#define GHF_EO_SRC_LAST 1
#endif
#ifndef GHF_EO_STAGES_A
#define GHF_EO_STAGES_A 2  // register sets of gathered SOURCE rows in flight in edge_outer_h_kernel (2 or 3)
#endif
#ifndef GHF_EO_STAGES_B
#define GHF_EO_STAGES_B 2  // ... of destination rows (of h and G)
#endif
constexpr int EO_STAGES_A = GHF_EO_STAGES_A, EO_STAGES_B = GHF_EO_STAGES_B;
#ifndef GHF_EOEXP
#define GHF_EOEXP 0      // timing experiments (wrong results): 1 no products, 2 no row gathers, 4 no cutting / LDS writes, 8 one barrier less
#endif
__device__ __forceinline__ unsigned eo_off(int row, int ch) { return 256u * row + 16u * (unsigned)(ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <bool OFF32>   // h and G below 4 GB each: 32-bit byte offsets (see Idx)
__global__ __launch_bounds__(512) void edge_outer_h_kernel(
    const float* __restrict__ h, const float* __restrict__ G, const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
    const int64_t* __restrict__ dst, int xa_col, int xb_col, int g_col, int ld,
    const int64_t* __restrict__ slice_tab, const unsigned* __restrict__ amax /* bits of max|h|, max|G| */,
    float* __restrict__ partial, float* __restrict__ partial_b, const int32_t* __restrict__ order,
    const int* __restrict__ guard /* rowscale_guard_kernel's counters, or NULL */) {
    constexpr int D = 128, RG = 4, CG = 2, NT = 512;
    if (eo_wide(guard)) return;                           // (the exact kernel, launched behind this one, takes the call)
    constexpr int F4 = D / 4;                             // float4 per row of h / G
    constexpr int LPR = EO_ET * F4 / NT;                  // (row, float4) items each thread moves per region and tile: 2
    constexpr int IMG = EO_ET * 256;                      // bytes of one [32][128] fp16 image
    // per buffer: X_src hi, lo; X_dst hi, lo; G hi, lo
    extern __shared__ __attribute__((aligned(16))) char eoh_lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int rg = w / CG, cg = w % CG;                   // rows rg*64 .. +63 of [X_src | X_dst], columns cg*64 .. +63 of G
    const int c16 = lane & 15, Q = lane >> 4;
    const size_t slice = order ? (size_t)order[blockIdx.x] : (size_t)blockIdx.x;   // (launch order: ghf.h, ghf_edge_outer)
    const int64_t e0 = slice_tab[3 * slice + 1], e1 = slice_tab[3 * slice + 2];
    const int ntiles = (int)((e1 - e0 + EO_ET - 1) / EO_ET);
    const int sx = split2h_shift(__uint_as_float(amax[0])), sg = split2h_shift(__uint_as_float(amax[1]));
    const float upx = pow2f(sx), upg = pow2f(sg);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bsum[LPR];
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LPR; ++j) bsum[j] = zero4;

    // The indices of a tile are loaded a whole step before its rows (two sets), and every load is unconditional (rows past the
    // slice's end read its last edge's and are zeroed when they are cut): straight-line code, so the compiler's counted waits
    // name exactly the loads they need.  (With `ok ? load : 0` branches each group of loads sat in its own basic block behind an
    // s_waitcnt vmcnt(0): three dependent HBM round trips per tile.)
    // What travels from the index loads to the row loads is each row's BYTE OFFSET (this thread's column quad included) when
    // h and G are under 4 GB (OFF32; else the row id): the loads then take the tensor's base from scalar registers and one
    // 32-bit register each — no 64-bit addresses kept in vector registers across the step (NS sets of rows in flight leave none).
    struct IdxA { unsigned s[LPR]; };                     // (node ids fit 31 bits: the plan's limit)
    struct IdxB { unsigned v[LPR], d[LPR]; };
    IdxA ixa[2];
    IdxB ixb[2];
    // Rows on their way to LDS, in registers: NSA tiles of source rows and NSB tiles of destination rows (of h and of G) in
    // flight.  With one set, a tile's rows had one tile's MFMAs (~0.7 us) to arrive and the workgroup waited for them every
    // iteration (4.7 TB/s over the chip).  With two they have a step and a bit, ~2 us against a loaded round trip of ~2.6 us:
    // the gathers alone (no products, no cutting) run at 9.5 TB/s of HBM + Infinity Cache — 1.6 ms at C3 — the products and the
    // cutting alone take as long, and the two together 2.6 ms (round 4's ablations, GHF_EOEXP).  A third set would cover the
    // round trip, but does not fit: three whole sets (72 registers beside the 64 accumulators, the 40 of a step's fragments
    // and ~20 LDS addresses) spill ~30 registers inside the loop and the launch takes 3.9 ms instead of 2.9; a third set for the
    // source rows only (NSA = 3, NSB = 2: the rows that come from HBM, the destination rows of a band of slices launched
    // together sit in the Infinity Cache) still spills 19, because the trip is then six steps long and hipcc's allocation over
    // it needs ~45 registers more than over two.  Two and two it stays; what is left is a pipeline that keeps the rows in LDS
    // (LDS-DMA ring, as message_rs.hip's pass 1) instead of registers.
    constexpr int NSA = OFF32 ? EO_STAGES_A : 2, NSB = OFF32 ? EO_STAGES_B : 2;
    f32x4 sa[NSA][LPR], sb[NSB][2][LPR];
    // (positions inside the slice in 32 bits — a slice holds a few thousand edges; the index arrays from the slice's first edge)
    const int nedge = (int)(e1 - e0);
    const unsigned* __restrict__ ja = (const unsigned*)(ia + e0);
    const unsigned* __restrict__ jb = (const unsigned*)(ib + e0);
    const unsigned* __restrict__ jd = (const unsigned*)(dst + e0);
    auto row_ok = [&](int tile, int j) { return tile * EO_ET + (t + NT * j) / F4 < nedge; };
    const unsigned ldb = (unsigned)ld * 4u;
    auto edge_of = [&](int tile, int j) { const int e = tile * EO_ET + (t + NT * j) / F4; return e < nedge ? e : nedge - 1; };   // (a slice is never empty)
    // (little-endian int64 ids below 2^31: the low word is the id)
    auto load_idx_a = [&](int tile, IdxA& I) {
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const unsigned is = ja[2 * edge_of(tile, j)];
            I.s[j] = OFF32 ? is * ldb + 4u * xa_col + 16u * ((t + NT * j) % F4) : is;
        }
    };
    auto load_idx_b = [&](int tile, IdxB& I) {
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const int ec = edge_of(tile, j);
            const unsigned c = 16u * ((t + NT * j) % F4);
            const unsigned iv = jb[2 * ec], id = jd[2 * ec];
            I.v[j] = OFF32 ? iv * ldb + 4u * xb_col + c : iv;
            I.d[j] = OFF32 ? id * ldb + 4u * g_col + c : id;
        }
    };
    auto gather_a = [&](f32x4 (&S)[LPR], const IdxA& I) {
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            if (GHF_EOEXP & 2) { S[j] = (f32x4){1.f * I.s[j], 0.f, 0.f, 0.f}; continue; }
            if (OFF32) S[j] = *(const f32x4*)((const char*)h + I.s[j]);
            else S[j] = *(const f32x4*)(h + (size_t)I.s[j] * ld + xa_col + 4 * ((t + NT * j) % F4));
        }
    };
    auto gather_b = [&](f32x4 (&S)[2][LPR], const IdxB& I) {
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            if (GHF_EOEXP & 2) { S[0][j] = S[1][j] = (f32x4){1.f * I.v[j], 1.f * I.d[j], 0.f, 0.f}; continue; }
            if (OFF32) {
                S[0][j] = *(const f32x4*)((const char*)h + I.v[j]);
                S[1][j] = *(const f32x4*)((const char*)G + I.d[j]);
            } else {
                const int c4 = (t + NT * j) % F4;
                S[0][j] = *(const f32x4*)(h + (size_t)I.v[j] * ld + xb_col + 4 * c4);
                S[1][j] = *(const f32x4*)(G + (size_t)I.d[j] * ld + g_col + 4 * c4);
            }
        }
    };
    // One (row, column quad) of one of the three tensors cut into its two pieces (unit = 3 j + reg, the order the rows were
    // requested in): row-major images.  Rows past the slice's end (copies of its last edge's rows) are cut as zeros by a zero
    // scale (a copy of a row the sums hold anyway: 0 x inf could only put a NaN where the result is no number already).
    auto commit_unit = [&](int buf, int k, int unit, int tile) __attribute__((always_inline)) {   // tile's rows: sa[k % NSA], sb[k % NSB], k = tile mod TRIP
#if GHF_EO_SRC_LAST
        // units in the order the rows were requested in: the destination rows of h and G first, the source rows — the ones that
        // come from HBM rather than the Infinity Cache — last: half a step more for them to arrive
        const int j = unit < 2 * LPR ? unit / 2 : unit - 2 * LPR, reg = unit < 2 * LPR ? 1 + unit % 2 : 0;
#else
        const int j = unit / 3, reg = unit % 3;
#endif
        char* base = eoh_lds + (size_t)buf * 6 * IMG;
        const int row = (t + NT * j) / F4, c4 = (t + NT * j) % F4;
        const unsigned o = eo_off(row, c4 >> 1) + 8u * (c4 & 1);
        const bool ok = row_ok(tile, j);
        const float up = ok ? (reg == 2 ? upg : upx) : 0.f;
        const f32x4 x = reg == 0 ? sa[k % NSA][j] : sb[k % NSB][reg - 1][j];
        eo_f16x4 hi4, lo4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 hi, lo;
            split2h(x[e] * up, hi, lo);
            hi4[e] = hi;
            lo4[e] = lo;
        }
        *(eo_f16x4*)(base + (2 * reg) * IMG + o) = hi4;
        *(eo_f16x4*)(base + (2 * reg + 1) * IMG + o) = lo4;
        if (reg == 2 && ok) bsum[j] += x;                 // db rides along: this thread's column quad, exact fp32
    };
    // fragment of 16 columns (features fb .. fb+15) x 32 rows (edges) of an image: what lane (Q, c16) needs is column c16, rows
    // 8Q .. 8Q+7 — two transposed reads (rows 8Q.. and 8Q+4..); lane 4q+p of a 16-lane group supplies row r0+q, columns 4p..
    const int gq = c16 >> 2, gp = c16 & 3;
    auto frag = [&](const char* img, int fb) -> eo_f16x8 {
#ifdef GHF_EO_SLOW_FRAG
        {   // (debug) the same fragment, element by element
            eo_f16x8 r;
            const int col = fb + c16;
            for (int e = 0; e < 8; ++e) r[e] = *(const _Float16*)(img + eo_off(8 * Q + e, col >> 3) + 2 * (col & 7));
            return r;
        }
#endif
        const int ch = (fb >> 3) + (gp >> 1);
        typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
        const auto p0 = (__attribute__((address_space(3))) h4_t*)(img + eo_off(8 * Q + gq, ch) + 8u * (gp & 1));
        const auto p1 = (__attribute__((address_space(3))) h4_t*)(img + eo_off(8 * Q + 4 + gq, ch) + 8u * (gp & 1));
        const eo_f16x4 a = __builtin_bit_cast(eo_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(p0));
        const eo_f16x4 b = __builtin_bit_cast(eo_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(p1));
        // (whole-vector casts and one shuffle: hipcc turned an element-by-element copy of the two halves into permutes of
        // their first dwords only)
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // tile k's rows travel in sa[k % NSA] and sb[k % NSB]; before the loop: tiles 0 .. NSA-1 / NSB-1 requested, the indices of
    // the next tile loaded, tile 0 cut
#pragma unroll
    for (int k = 0; k < (NSA > NSB ? NSA : NSB); ++k) {
        if (k < NSB) { load_idx_b(k, ixb[k & 1]); gather_b(sb[k], ixb[k & 1]); }
        if (k < NSA) { load_idx_a(k, ixa[k & 1]); gather_a(sa[k], ixa[k & 1]); }
    }
    load_idx_a(NSA, ixa[NSA & 1]);
    load_idx_b(NSB, ixb[NSB & 1]);
#pragma unroll
    for (int u = 0; u < 3 * LPR; ++u) commit_unit(0, 0, u, 0);
    __syncthreads();
    // step K of a trip (tile = first + K, K = tile mod TRIP): this tile's products on image buffer K % 2; the source rows of
    // tile + NSA and the destination rows of tile + NSB requested into the sets tile's rows left a step ago (their indices came a
    // step ago; the next ones are loaded ahead of these row loads in the counter's order); tile + 1's rows cut into the other buffer
    auto step = [&](int tile, auto K_) __attribute__((always_inline)) {
        constexpr int K = decltype(K_)::value;
        load_idx_a(tile + NSA + 1, ixa[(K + NSA + 1) & 1]);
        load_idx_b(tile + NSB + 1, ixb[(K + NSB + 1) & 1]);
#if GHF_EO_SRC_LAST
        gather_b(sb[K % NSB], ixb[(K + NSB) & 1]);
        gather_a(sa[K % NSA], ixa[(K + NSA) & 1]);
#else
        gather_a(sa[K % NSA], ixa[(K + NSA) & 1]);
        gather_b(sb[K % NSB], ixb[(K + NSB) & 1]);
#endif
        __builtin_amdgcn_sched_barrier(0);                // (the loads stay at the top of the step: left alone hipcc sinks them to its end)
        const char* base = eoh_lds + (size_t)(K & 1) * 6 * IMG;
        const char* ximg = base + (rg >> 1) * 2 * IMG;    // rows 0..127 of [X_src | X_dst] are the source image, 128..255 the destination one
        const char* gimg = base + 4 * IMG;
        // This tile's products, four fragments of X at a time, and between them the next tile's rows cut two units at a time:
        // the matrix pipe and the vector ALU side by side within every group, and nothing moved from one group to another
        // (left to itself hipcc hoists every fragment read and address to the top of the step and runs out of registers: NS
        // sets of rows in flight leave ~40 for all of this).
        eo_f16x8 bh[4], bl[4];
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
            bh[bi] = frag(gimg, cg * 64 + 16 * bi);
            bl[bi] = frag(gimg + IMG, cg * 64 + 16 * bi);
        }
        static_assert(LPR == 2, "six units over the first three of the four groups");
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {
            const int fb = (rg & 1) * 64 + 16 * ai;
            const eo_f16x8 ah = frag(ximg, fb), al = frag(ximg + IMG, fb);
            if (!(GHF_EOEXP & 1)) {
#pragma unroll
                for (int bi = 0; bi < 4; ++bi) {
                    acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[bi], acc[ai][bi], 0, 0, 0);
                    acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[bi], acc[ai][bi], 0, 0, 0);
                    acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[bi], acc[ai][bi], 0, 0, 0);
                }
            }
            if (ai < 3 && !(GHF_EOEXP & 4)) {
                commit_unit((K & 1) ^ 1, K + 1, 2 * ai, tile + 1);
                commit_unit((K & 1) ^ 1, K + 1, 2 * ai + 1, tile + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (GHF_EOEXP & 4) { for (int j = 0; j < LPR; ++j) bsum[j] += sa[(K + 1) % NSA][j] + sb[(K + 1) % NSB][0][j] + sb[(K + 1) % NSB][1][j]; }
        // (The two halves of a step touch different buffers, so their order is free.  Taken in opposite orders by the two
        // waves that share a SIMD — one's products beside the other's cutting — the launch got SLOWER: 2.92 -> 3.35 ms at C3,
        // round 4.  Every wave in the same half at the same time it stays.)
        __syncthreads();
    };
    // A trip is lcm(NSA, NSB, 2) steps (the sets turn with their periods, the image buffers and the index sets with period 2), all of them
    // unconditional — a count that is no multiple of it ends in steps on tiles of zeros (rows past the slice's end are cut as
    // zeros; at most 5 of a slice's ~128 tiles, and slices are cut at multiples of 6 tiles) — so a trip is ONE basic
    // block.  With a step behind a branch hipcc rotated the loop and the next step's loads met the loads they depend on in one
    // block: a round trip in the open per trip.
    constexpr int TRIP = (NSA == 3 || NSB == 3) ? 6 : 2;
    static_assert(TRIP % NSA == 0 && TRIP % NSB == 0, "two or three sets");
    for (int tile = 0; tile < ntiles; tile += TRIP) {
        step(tile + 0, std::integral_constant<int, 0>{});
        step(tile + 1, std::integral_constant<int, 1>{});
        if constexpr (TRIP == 6) {
            step(tile + 2, std::integral_constant<int, 2>{});
            step(tile + 3, std::integral_constant<int, 3>{});
            step(tile + 4, std::integral_constant<int, 4>{});
            step(tile + 5, std::integral_constant<int, 5>{});
        }
    }
    // partial product: tile (ai, bi) register s is row rg*64 + 16 ai + 4Q + s, column cg*64 + 16 bi + c16
    const float down = pow2f(-sx) * pow2f(-sg);
    float* P = partial + slice * 2 * D * D;
#pragma unroll
    for (int ai = 0; ai < 4; ++ai)
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                P[(size_t)(rg * 64 + 16 * ai + 4 * Q + s) * D + cg * 64 + 16 * bi + c16] = acc[ai][bi][s] * down;
    // db: threads (row, c4) with equal c4 hold partial column sums; rows in thread order: fixed summation order
    float* red = (float*)eoh_lds;                         // [NT / F4 = 16 row groups][D] (every tile read is behind the loop's last barrier)
    f32x4 bs = zero4;
#pragma unroll
    for (int j = 0; j < LPR; ++j) bs += bsum[j];
    *(f32x4*)(red + (size_t)(t / F4) * D + 4 * (t % F4)) = bs;
    __syncthreads();
    if (t < D) {
        float s_ = 0.f;
#pragma unroll
        for (int k = 0; k < NT / F4; ++k) s_ += red[k * D + t];
        partial_b[slice * D + t] = s_;
    }
}

int edge_outer_supported(int d) { return d == 64 || (d >= 128 && d <= BW_MAX_D && (d % 128) == 0); }

int launch_edge_outer(const float* h, const float* G, const int64_t* src, const int64_t* dst, const int64_t* slice_tab,
                      const int64_t* slice_off, int64_t nslices, int R, int d, int64_t N, float* workspace, float* dW, float* db,
                      hipStream_t stream, const float* h_rowscale, const float* G_rowscale, const int32_t* order) {
    GHF_REQUIRE(edge_outer_supported(d), "edge_outer: d = %d has no tile (64 and multiples of 128 do; use ghf_group_outer)", d);
    GHF_REQUIRE(nslices > 0 && R > 0, "edge_outer: nothing to do");
    const int D = d == 64 ? 64 : 128;                      // tile: [2D, D] of the relation's [2d, d] gradient
    float* partial = workspace;
    float* partial_b = workspace + (size_t)nslices * 2 * D * D;
    const unsigned gx = (unsigned)cdiv((int64_t)2 * D * D / 4 + D, 256);
    // d % 128 == 0: two fp16 pieces on the 16-bit matrix pipe (edge_outer_h_kernel); GHF_EDGE_OUTER=exact keeps the fp32 MFMAs
    static const bool exact = getenv("GHF_EDGE_OUTER") && !strcmp(getenv("GHF_EDGE_OUTER"), "exact");
    unsigned* amax = (unsigned*)(partial_b + (size_t)nslices * D);     // two words behind the partial sums: max |h|, max |G|
    const bool pieces = D == 128 && !exact && N > 0;      // (N <= 0: the caller asks for the exact fp32 chain, ghf.h)
    int* guard = nullptr;                                  // four counters behind the two maxima (ghf.h: the workspace's last 64 floats)
    if (pieces) {
        GHF_HIP_CHECK(hipMemsetAsync(amax, 0, 6 * sizeof(unsigned), stream));
        if (h_rowscale && G_rowscale) {                    // the caller holds both tensors' split forms: their row scales say it
            // (few workgroups: every wave ends in atomics on the same two — six — words, which serialise in L2: 1,000
            // workgroups of them cost ~0.1 ms per call at N = 10^6, 128 next to nothing; the scales are 4 MB per tensor)
            const unsigned ag = (unsigned)(cdiv(N, 256 * 4) < 128 ? cdiv(N, 256 * 4) : 128);
            rowscale_absmax_kernel<<<dim3(ag, 2), 256, 0, stream>>>(h_rowscale, G_rowscale, N, amax);
            static const bool guarded = !(getenv("GHF_EO_GUARD") && !strcmp(getenv("GHF_EO_GUARD"), "0"));
            if (guarded) {
                guard = (int*)(amax + 2);
                rowscale_guard_kernel<<<dim3(ag, 2), 256, 0, stream>>>(h_rowscale, G_rowscale, N, amax, guard);
            }
        } else {
            const unsigned ag = (unsigned)(cdiv(N * d, 256 * 16) < 2048 ? cdiv(N * d, 256 * 16) : 2048);
            absmax_kernel<<<dim3(ag, 2), 256, 0, stream>>>(h, G, N * d, amax);
        }
        GHF_LAUNCH_CHECK();
    }
    for (int rb = 0; rb < d / D; ++rb)                     // tile rows 2D*rb ..: two D-column pieces of [h_src | h_dst]
        for (int cb = 0; cb < d / D; ++cb) {
            const int fa = 2 * rb * D, fb = fa + D;        // first stacked feature of the two pieces
            const int64_t* ia = fa < d ? src : dst;
            const int64_t* ib = fb < d ? src : dst;
            int rc = GHF_OK;
            if (pieces) {
                constexpr size_t lds = (size_t)2 * 6 * EO_ET * 256;
                if ((uint64_t)N * (uint64_t)d * 4u < (1ull << 32)) {
                    GHF_SET_MAX_LDS(edge_outer_h_kernel<true>, lds);
                    edge_outer_h_kernel<true><<<(unsigned)nslices, 512, lds, stream>>>(h, G, ia, ib, dst, fa % d, fb % d, cb * D, d, slice_tab,
                                                                                       amax, partial, partial_b, order, guard);
                } else {
                    GHF_SET_MAX_LDS(edge_outer_h_kernel<false>, lds);
                    edge_outer_h_kernel<false><<<(unsigned)nslices, 512, lds, stream>>>(h, G, ia, ib, dst, fa % d, fb % d, cb * D, d, slice_tab,
                                                                                        amax, partial, partial_b, order, guard);
                }
                GHF_LAUNCH_CHECK();
                // the same slices on the exact fp32 chain, behind it: every workgroup of one of the two returns at once, which
                // one the counters say — no host round trip
                if (guard) rc = edge_outer_launch<128>(h, G, ia, ib, dst, fa % d, fb % d, cb * D, d, slice_tab, nslices, partial, partial_b,
                                                       order, stream, guard);
            } else
                rc = D == 128 ? edge_outer_launch<128>(h, G, ia, ib, dst, fa % d, fb % d, cb * D, d, slice_tab, nslices, partial,
                                                       partial_b, order, stream)
                              : edge_outer_launch<64>(h, G, ia, ib, dst, fa % d, fb % d, cb * D, d, slice_tab, nslices, partial,
                                                      partial_b, order, stream);
            if (rc != GHF_OK) return rc;
            edge_outer_reduce_kernel<<<dim3(gx, (unsigned)R), 256, 0, stream>>>(partial, partial_b, slice_off, D, d, 2 * rb * D, cb * D,
                                                                               dW, rb == 0 ? db : nullptr);
            GHF_LAUNCH_CHECK();
        }
    return GHF_OK;
}

// out[b][j][i] = in[b][i][j]
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in, int rows, int cols,
                                                                float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const float* src = in + (size_t)b * rows * cols;
    float* dst = out + (size_t)b * rows * cols;
    for (int k = threadIdx.x; k < 1024; k += 256) {
        const int r = r0 + k / 32, c = c0 + k % 32;
        if (r < rows && c < cols) tile[k / 32][k % 32] = src[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256) {
        const int c = c0 + k / 32, r = r0 + k % 32;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[k % 32][k / 32];
    }
}

static int64_t tail_bwd_blocks(int64_t N) { const int64_t b = cdiv(N, 4); return b < TB_MAX_BLOCKS ? (b > 0 ? b : 1) : TB_MAX_BLOCKS; }

size_t colsum_workspace_floats(int64_t N, int d);
size_t tail_bwd_workspace_floats(int64_t N, int d) {          // the workgroups' partial sums + their colsum's levels
    const int64_t nb = tail_bwd_blocks(N);
    return (size_t)nb * 2 * d + colsum_workspace_floats(nb, 2 * d);
}

int launch_tail_bwd(const float* g_out, const float* agg, const float* h, const float* gamma, float eps, const int32_t* indeg,
                    int64_t N, int d, float* dpre, float* G, void* G_split, float* dgb, float* workspace, const float* drop,
                    hipStream_t stream) {
    GHF_REQUIRE(d >= 1 && d <= BW_MAX_D, "tail_bwd: d=%d outside [1,%d]", d, BW_MAX_D);
    if (N <= 0) {
        GHF_HIP_CHECK(hipMemsetAsync(dgb, 0, sizeof(float) * 2 * d, stream));
        return GHF_OK;
    }
    const unsigned nb = (unsigned)tail_bwd_blocks(N);
    float* part = workspace;
    const bool a16 = ((((uintptr_t)g_out | (uintptr_t)agg | (uintptr_t)h | (uintptr_t)gamma | (uintptr_t)dpre | (uintptr_t)G |
                       (uintptr_t)drop | (uintptr_t)G_split) & 15) == 0);
    bool split_done = false;
#define GHF_TB_V4(LPR, NV)                                                                                                      \
    do {                                                                                                                        \
        tail_bwd_v4_kernel<LPR, NV><<<nb, 256, 0, stream>>>(g_out, agg, h, gamma, eps, indeg, N, dpre, G, (char*)G_split, part, \
                                                            drop, range_flag_ptr());                                            \
        split_done = true;                                                                                                      \
    } while (0)
    if (a16 && d == 32) GHF_TB_V4(8, 1);
    else if (a16 && d == 64) GHF_TB_V4(16, 1);
    else if (a16 && d == 128) GHF_TB_V4(32, 1);
    else if (a16 && d == 256) GHF_TB_V4(64, 1);
    else if (a16 && d == 512) GHF_TB_V4(64, 2);
    else if (a16 && d == 1024) GHF_TB_V4(64, 4);
    else tail_bwd_kernel<<<nb, 256, sizeof(float) * 8 * d, stream>>>(g_out, agg, h, gamma, eps, indeg, N, d, dpre, G, part, drop);
#undef GHF_TB_V4
    GHF_LAUNCH_CHECK();
    if (G_split && !split_done) {
        const int rc = launch_split2h_rows(G, N, d, 0, N, G_split, stream);
        if (rc != GHF_OK) return rc;
    }
    return launch_colsum(part, nullptr, nb, 2 * d, workspace + (size_t)nb * 2 * d, dgb, 0, stream);
}


size_t colsum_workspace_floats(int64_t N, int d) {         // two ping-pong levels of partial sums
    const int64_t n1 = cdiv(N, CS_ROWS);
    return (size_t)(n1 + cdiv(n1, CS_ROWS)) * (size_t)d;
}

int launch_colsum(const float* X, const float* mask, int64_t N, int d, float* workspace, float* out, int accumulate, hipStream_t stream) {
    GHF_REQUIRE(N > 0 && d > 0, "colsum: bad shape");
    const bool v4 = (d % 4) == 0 && ((((uintptr_t)X | (uintptr_t)mask | (uintptr_t)workspace) & 15) == 0);
    const unsigned gy = (unsigned)cdiv(v4 ? d / 4 : d, 256);
    const float* in = X;
    int64_t rows = N;
    float* level[2] = {workspace, workspace + (size_t)cdiv(N, CS_ROWS) * d};
    for (int pass = 0;; ++pass) {
        const int64_t nblk = cdiv(rows, CS_ROWS);
        const bool last = nblk == 1;
        float* dst = last ? out : level[pass & 1];
        const float* m = pass == 0 ? mask : nullptr;
        const int acc = last ? accumulate : 0;
        if (v4) colsum_pass_kernel<4><<<dim3((unsigned)nblk, gy), 256, 0, stream>>>(in, m, rows, d, dst, acc);
        else colsum_pass_kernel<1><<<dim3((unsigned)nblk, gy), 256, 0, stream>>>(in, m, rows, d, dst, acc);
        GHF_LAUNCH_CHECK();
        if (last) break;
        in = dst;
        rows = nblk;
    }
    return GHF_OK;
}

int launch_relu_mask(const float* X, const float* ref, int64_t n, float* out, hipStream_t stream) {
    if (n <= 0) return GHF_OK;
    relu_mask_kernel<<<(unsigned)cdiv(n, 256), 256, 0, stream>>>(X, ref, n, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_group_outer(const float* A, const int64_t* ia, int da, const float* B, const int64_t* ib, int db,
                       const int64_t* gstart, const int64_t* gend, int ngroups, float* C, int accumulate, hipStream_t stream) {
    GHF_REQUIRE(da >= 0 && db > 0 && ngroups > 0, "group_outer: bad shape");
    const int rows_a = da > 0 ? da : 1;
    const dim3 grid((unsigned)ngroups, (unsigned)cdiv(rows_a, 16));
    for (int o_base = 0; o_base < db; o_base += 128) {
        group_outer_kernel<8><<<grid, 256, 0, stream>>>(A, ia, da, B, ib, db, gstart, gend, C, o_base, accumulate);
        GHF_LAUNCH_CHECK();
    }
    return GHF_OK;
}

// out[0] = sum_i X[i] * Y[i], two deterministic stages (workspace: cdiv(n, 8192) floats)
constexpr int64_t DOT_BLOCK = 8192;                      // elements per workgroup of the first stage
__global__ __launch_bounds__(256) void dot_stage1_kernel(const float* __restrict__ X, const float* __restrict__ Y, int64_t n,
                                                         float* __restrict__ part) {
    __shared__ float red[4];
    const int64_t b0 = (int64_t)blockIdx.x * DOT_BLOCK;
    float s = 0.f;
    for (int64_t i = b0 + threadIdx.x; i < b0 + DOT_BLOCK && i < n; i += 256) s = fmaf(X[i], Y[i], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(64) void dot_stage2_kernel(const float* __restrict__ part, int64_t nblk, float* __restrict__ out) {
    float s = 0.f;                                           // one wave: lane l adds blocks l, l + 64, ... in order
    for (int64_t b = threadIdx.x; b < nblk; b += 64) s += part[b];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

int launch_dot(const float* X, const float* Y, int64_t n, float* workspace, float* out, hipStream_t stream) {
    GHF_REQUIRE(n > 0, "dot: empty input");
    const int64_t nblk = cdiv(n, DOT_BLOCK);
    dot_stage1_kernel<<<(unsigned)nblk, 256, 0, stream>>>(X, Y, n, workspace);
    GHF_LAUNCH_CHECK();
    dot_stage2_kernel<<<1, 64, 0, stream>>>(workspace, nblk, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

// elementwise helpers of the backward (HBM-bound, 16 bytes per lane where the length allows)
__global__ __launch_bounds__(256) void scale_exp_kernel(const float* __restrict__ X, int64_t n, const float* __restrict__ log_scale,
                                                        float* __restrict__ out) {
    const float s = expf(log_scale[0]);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = X[i] * s;
}
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ c, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = c ? (a[i] + b[i]) + c[i] : a[i] + b[i];
}
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ X, const float* __restrict__ g, int64_t n, int d,
                                                       float* __restrict__ out) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] = X[i] * g[i / d];
}

static unsigned ew_grid(int64_t n) { const int64_t b = cdiv(n, 256); return (unsigned)(b < 8192 ? (b > 0 ? b : 1) : 8192); }

int launch_scale_exp(const float* X, int64_t n, const float* log_scale, float* out, hipStream_t stream) {
    scale_exp_kernel<<<ew_grid(n), 256, 0, stream>>>(X, n, log_scale, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}
int launch_add3(const float* a, const float* b, const float* c, int64_t n, float* out, hipStream_t stream) {
    add3_kernel<<<ew_grid(n), 256, 0, stream>>>(a, b, c, n, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}
int launch_rowscale(const float* X, const float* g, int64_t n, int d, float* out, hipStream_t stream) {
    GHF_REQUIRE(d > 0, "rowscale: d must be positive");
    rowscale_kernel<<<ew_grid(n * d), 256, 0, stream>>>(X, g, n, d, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

// out[v][:] = sum_{e in off[v] .. off[v+1]} w[iw[e]] * X[ix[e]][:], e ascending (fixed order); empty segments give zeros.  One
// wave per segment, VW adjacent columns per lane and 64 * VW columns per pass, two entries' rows in flight.  (The gradient of
// fused edge scores: per node, the pairs it takes part in.)
template <int VW>
__global__ __launch_bounds__(256) void segment_axpy_kernel(const float* __restrict__ w, const int64_t* __restrict__ iw,
                                                           const float* __restrict__ X, const int64_t* __restrict__ ix,
                                                           const int64_t* __restrict__ off, int64_t nseg, int64_t nx, int d,
                                                           float* __restrict__ out) {
    typedef float vec __attribute__((ext_vector_type(VW)));
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= nseg) return;
    const int64_t e0 = off[v], e1 = off[v + 1];
    for (int c = VW * lane; c < d; c += 64 * VW) {
        vec acc;
#pragma unroll
        for (int k = 0; k < VW; ++k) acc[k] = 0.f;
        auto row = [&](int64_t e) -> const vec* {
            int64_t r = ix[e];
            r = r < 0 ? 0 : (r >= nx ? nx - 1 : r);       // (ids are the caller's: clamped, never a fault)
            return (const vec*)(X + (size_t)r * d + c);
        };
        int64_t e = e0;
        for (; e + 2 <= e1; e += 2) {
            const float w0 = w[iw[e]], w1 = w[iw[e + 1]];
            const vec x0 = *row(e), x1 = *row(e + 1);
#pragma unroll
            for (int k = 0; k < VW; ++k) acc[k] = fmaf(w1, x1[k], fmaf(w0, x0[k], acc[k]));
        }
        if (e < e1) {
            const float w0 = w[iw[e]];
            const vec x0 = *row(e);
#pragma unroll
            for (int k = 0; k < VW; ++k) acc[k] = fmaf(w0, x0[k], acc[k]);
        }
        *(vec*)(out + (size_t)v * d + c) = acc;
    }
}

int launch_segment_axpy(const float* w, const int64_t* iw, const float* X, const int64_t* ix, const int64_t* off, int64_t nseg,
                        int64_t nx, int d, float* out, hipStream_t stream) {
    GHF_REQUIRE(d > 0 && nx > 0, "segment_axpy: bad shape");
    if (nseg <= 0) return GHF_OK;
    GHF_REQUIRE(cdiv(nseg, 4) < (1ll << 31), "segment_axpy: too many segments per launch");
    const unsigned grid = (unsigned)cdiv(nseg, 4);
    const bool a16 = ((((uintptr_t)X | (uintptr_t)out) & 15) == 0);
    if (a16 && d % 4 == 0 && d >= 256) segment_axpy_kernel<4><<<grid, 256, 0, stream>>>(w, iw, X, ix, off, nseg, nx, d, out);
    else if (a16 && d % 2 == 0 && d >= 128) segment_axpy_kernel<2><<<grid, 256, 0, stream>>>(w, iw, X, ix, off, nseg, nx, d, out);
    else segment_axpy_kernel<1><<<grid, 256, 0, stream>>>(w, iw, X, ix, off, nseg, nx, d, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_transpose_batched(const float* in, int batch, int rows, int cols, float* out, hipStream_t stream) {
    GHF_REQUIRE(batch > 0 && rows > 0 && cols > 0 && batch < 65536, "transpose: bad shape");
    transpose_batched_kernel<<<dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32), (unsigned)batch), 256, 0, stream>>>(in, rows, cols, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
