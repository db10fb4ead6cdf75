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

// Pass 2: out_v = (1/max(indeg,1)) * sum of v's rows of Y (contiguous: off[v] .. off[v+1]), then the tail.  One wave per
// destination, four rows in flight per lane and column, fixed summation order.
constexpr int RS_MAX_D = 1024, RS_PER_LANE = RS_MAX_D / 64;
__global__ __launch_bounds__(256) void segment_tail_kernel(
    const float* __restrict__ Y, const int64_t* __restrict__ off, const float* __restrict__ h, const float* __restrict__ g,
    const float* __restrict__ b, float eps, int64_t row0, int64_t row_end, int d, float* __restrict__ h_out, int no_tail) {
    const int lane = threadIdx.x & 63;
    const int64_t v = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= row_end) return;
    const int64_t p0 = off[v], p1 = off[v + 1];
    const int64_t deg = p1 - p0;
    const float inv = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg > 1 ? deg : 1);
    float x[RS_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < RS_PER_LANE; ++c) {
        const int o = lane + 64 * c;
        x[c] = 0.f;
        if (o < d) {
            const float* __restrict__ p = Y + (size_t)p0 * d + o;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            int64_t j = 0;
            for (; j + 4 <= deg; j += 4) {
                t0 += p[(size_t)j * d];
                t1 += p[(size_t)(j + 1) * d];
                t2 += p[(size_t)(j + 2) * d];
                t3 += p[(size_t)(j + 3) * d];
            }
            for (; j < deg; ++j) t0 += p[(size_t)j * d];
            const float tt = ((t0 + t1) + (t2 + t3)) * inv;
            x[c] = no_tail ? tt : fmaxf(tt + h[(size_t)v * d + o], 0.f);
            s += x[c];
        }
    }
    if (!no_tail) {
        const float mean = wave_sum(s) / (float)d;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < RS_PER_LANE; ++c)
            if (lane + 64 * c < d) { const float tt = x[c] - mean; var += tt * tt; }
        const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)d + eps);
#pragma unroll
        for (int c = 0; c < RS_PER_LANE; ++c) {
            const int o = lane + 64 * c;
            if (o < d) x[c] = (x[c] - mean) * rstd * g[o] + b[o];
        }
    }
#pragma unroll
    for (int c = 0; c < RS_PER_LANE; ++c) {
        const int o = lane + 64 * c;
        if (o < d) h_out[(size_t)v * d + o] = x[c];
    }
}

int message_rs_supported(int d) { return d >= 256 && d <= RS_MAX_D && (d % RS_TN) == 0; }

int launch_edge_transform(const float* h, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                          const int64_t* slice_tab, int64_t nslices, const float* WmT, const float* WsT, const float* bias,
                          float* Y, hipStream_t stream) {
    GHF_REQUIRE(message_rs_supported(d), "edge_transform: d = %d has no relation-stationary kernel (d %% 128 == 0, 256 <= d <= %d)", d, RS_MAX_D);
    GHF_REQUIRE(N > 0 && nslices > 0 && nslices < (1ll << 31), "edge_transform: bad sizes");
    edge_transform_kernel<<<dim3((unsigned)nslices, (unsigned)(d / RS_TN)), 256, 0, stream>>>(h, d, src, dst, ypos, slice_tab,
                                                                                            WmT, WsT, bias, Y);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_segment_tail(const float* Y, const int64_t* off, const float* h, const float* g, const float* b, float eps,
                        int64_t row0, int64_t rows, int d, float* h_out, int flags, hipStream_t stream) {
    GHF_REQUIRE(d >= 1 && d <= RS_MAX_D, "segment_tail: d=%d outside [1,%d]", d, RS_MAX_D);
    if (rows <= 0) return GHF_OK;
    GHF_REQUIRE(cdiv(rows, 4) < (1ll << 31), "segment_tail: too many rows per launch");
    segment_tail_kernel<<<(unsigned)cdiv(rows, 4), 256, 0, stream>>>(Y, off, h, g, b, eps, row0, row0 + rows, d, h_out,
                                                                     flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM));
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
