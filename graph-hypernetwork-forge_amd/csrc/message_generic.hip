// message_generic.hip — K2+K3 for any hidden size d (vector ALU), and K3 alone.
//
// Replaces models/hypergnn.py:281-296 of the reference without ever forming the
// per-edge weight copies of :281-283.  Plan geometry: CSR by destination
// (block_nodes == 1, key = dst*R + rel, seg_off = row offsets), weights in the
// reference's natural layout.  One workgroup per destination node v:
//     acc[o] = sum_{e=(u->v)} ( bias[r_e][o] + sum_i h_u[i] W_msg[r_e][i][o] + h_v[i] W_self[r_e][i][o] )
//     x[o]   = acc[o] / max(indeg_v, 1) + h_v[o];  h'_v = LayerNorm(ReLU(x))
// Lanes own output columns o (coalesced reads of W[r][i][:]); h_u and h_v sit in LDS.
// Edges arrive sorted by (dst, rel), so the summation order is fixed: results are
// bitwise reproducible run to run.
// This is the correctness-first kernel; the MFMA kernels (message_bx.hip, message_pp.hip) take over
// for the hidden sizes it is built for.
#include "common.h"

#include <stdlib.h>

namespace ghf {

constexpr int GEN_MAX_D = 1024;
constexpr int GEN_TPB = 256;
constexpr int GEN_MAX_PER_THREAD = GEN_MAX_D / GEN_TPB;

template <int NW>
__device__ __forceinline__ void tail_store(const float (&x)[GEN_MAX_PER_THREAD], int d, float* red,
                                           const float* __restrict__ g, const float* __restrict__ b, float eps,
                                           float* __restrict__ out_row) {
    // LayerNorm over d values spread as x[c] <-> column threadIdx.x + c*blockDim.x (biased variance)
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) {
        const int o = threadIdx.x + c * GEN_TPB;
        if (o < d) s += x[c];
    }
    const float mean = block_sum<NW>(s, red) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) {
        const int o = threadIdx.x + c * GEN_TPB;
        if (o < d) { const float t = x[c] - mean; v += t * t; }
    }
    const float var = block_sum<NW>(v, red + NW) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) {
        const int o = threadIdx.x + c * GEN_TPB;
        if (o < d) out_row[o] = (x[c] - mean) * rstd * g[o] + b[o];
    }
}

__global__ __launch_bounds__(GEN_TPB) void message_generic_kernel(
    const float* __restrict__ h, int d, const uint32_t* __restrict__ key, const int32_t* __restrict__ srcs,
    const int32_t* __restrict__ row_off, const int32_t* __restrict__ indeg, int R,
    const float* __restrict__ Wm, const float* __restrict__ Ws, const float* __restrict__ bias,
    const float* __restrict__ g, const float* __restrict__ b, float eps, int64_t row0,
    float* __restrict__ h_out, int no_tail) {
    __shared__ float hv[GEN_MAX_D];
    __shared__ float hu[GEN_MAX_D];
    __shared__ float red[2 * (GEN_TPB / 64)];
    const int64_t v = row0 + blockIdx.x;
    const float* __restrict__ hrow = h + (size_t)v * d;
    for (int i = threadIdx.x; i < d; i += GEN_TPB) hv[i] = hrow[i];

    float acc[GEN_MAX_PER_THREAD];
#pragma unroll
    for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) acc[c] = 0.f;

    const int e0 = row_off[v], e1 = row_off[v + 1];
    const size_t dd = (size_t)d * d;
    for (int e = e0; e < e1; ++e) {
        const int r = (int)(key[e] - (uint32_t)v * (uint32_t)R);
        const float* __restrict__ urow = h + (size_t)srcs[e] * d;
        __syncthreads();                                   // previous hu fully consumed; hv visible on first pass
        for (int i = threadIdx.x; i < d; i += GEN_TPB) hu[i] = urow[i];
        __syncthreads();
        const float* __restrict__ wm = Wm + (size_t)r * dd;
        const float* __restrict__ ws = Ws + (size_t)r * dd;
#pragma unroll
        for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) {
            const int o = threadIdx.x + c * GEN_TPB;
            if (o < d) {
                float s = bias[(size_t)r * d + o];
                for (int i = 0; i < d; ++i) {
                    s = fmaf(hu[i], wm[(size_t)i * d + o], s);
                    s = fmaf(hv[i], ws[(size_t)i * d + o], s);
                }
                acc[c] += s;
            }
        }
    }
    __syncthreads();

    const int deg = indeg[v];
    const float inv = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg > 1 ? deg : 1);
    float* __restrict__ orow = h_out + (size_t)v * d;
    if (no_tail) {
#pragma unroll
        for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) {
            const int o = threadIdx.x + c * GEN_TPB;
            if (o < d) orow[o] = acc[c] * inv;
        }
        return;
    }
    float x[GEN_MAX_PER_THREAD];
#pragma unroll
    for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) {
        const int o = threadIdx.x + c * GEN_TPB;
        x[c] = (o < d) ? fmaxf(acc[c] * inv + hv[o], 0.f) : 0.f;
    }
    tail_store<GEN_TPB / 64>(x, d, red, g, b, eps, orow);
}

// K3 alone: h' = LayerNorm(ReLU(agg + h)) on rows [row0, row0+rows)
__global__ __launch_bounds__(GEN_TPB) void tail_kernel(const float* __restrict__ agg, const float* __restrict__ h,
                                                       const float* __restrict__ g, const float* __restrict__ b,
                                                       float eps, int64_t row0, int d, float* __restrict__ h_out,
                                                       const float* __restrict__ drop) {
    __shared__ float red[2 * (GEN_TPB / 64)];
    const int64_t v = row0 + blockIdx.x;
    float x[GEN_MAX_PER_THREAD];
#pragma unroll
    for (int c = 0; c < GEN_MAX_PER_THREAD; ++c) {
        const int o = threadIdx.x + c * GEN_TPB;
        x[c] = (o < d) ? fmaxf(agg[(size_t)v * d + o] + h[(size_t)v * d + o], 0.f) : 0.f;
        if (drop && o < d) x[c] *= drop[(size_t)v * d + o];              // dropout between ReLU and LayerNorm (reference :293-294)
    }
    tail_store<GEN_TPB / 64>(x, d, red, g, b, eps, h_out + (size_t)v * d);
}

// Second stage for split destination blocks (hubs): sum the block's partial slots in a fixed order (reproducible), then
// the K3 tail.  One workgroup per COMB_ROWS destination rows of a split block — a hub block's hundred-odd
// slots are then read by dozens of workgroups with four loads in flight per lane instead of by one workgroup's serial
// chain (power-law C3: 4.7 ms -> 0.1 ms).  The grid covers the split blocks only: (items beyond one per block) x (row
// groups of a block), each workgroup finding its block by a 64-way search (a grid over all blocks cost 77 us per launch at
// BASELINE config 3, where only the last round's 45 blocks are split).  One wave per destination row, lanes stride
// the columns.
constexpr int COMB_MAX_PER_LANE = GEN_MAX_D / 64;
constexpr int COMB_ROWS = 8;
__global__ __launch_bounds__(256) void combine_split_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ item_tab, const int32_t* __restrict__ blk_item_off,
    const float* __restrict__ h, const int32_t* __restrict__ indeg, const float* __restrict__ g,
    const float* __restrict__ b, float eps, int64_t N, int d, int BN, int64_t blk0, int nblk, int64_t row_end,
    float* __restrict__ h_out, void* __restrict__ h_split_out, int split_layout, int no_tail, int32_t* __restrict__ range_flag,
    float* __restrict__ agg_out) {
    // blockIdx.x numbers the range's items beyond one per block: a block of k items owns k - 1 of them, found by bisection
    // on f(b) = (items before block b) - b; its row groups go round these k - 1 workgroup columns
    // (64 probes per step, one per lane: two or three dependent loads instead of a bisection's twelve)
    const int base = blk_item_off[blk0];
    int lo = 0;
    for (int span = nblk; span > 1;) {                    // the block is in [lo, lo + span)
        const int step = (span + 63) >> 6, probe = (int)(threadIdx.x & 63) * step;
        const bool le = probe < span && blk_item_off[blk0 + lo + probe] - base - (lo + probe) <= (int)blockIdx.x;
        const int l1 = 63 - __builtin_clzll(__ballot(le));             // f is monotone and f(lo) <= blockIdx.x: the set lanes are a prefix
        lo += l1 * step;
        span = span - l1 * step < step ? span - l1 * step : step;
    }
    const int64_t blk = blk0 + lo;
    const int i0 = blk_item_off[blk], i1 = blk_item_off[blk + 1];
    if (i1 - i0 <= 1) return;
    const int slot0 = item_tab[4 * (size_t)i0 + 3], nslots = i1 - i0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t node0 = blk * BN;
    const int nrows_blk = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    // the block's groups of COMB_ROWS rows go round its (nslots - 1) x gridDim.y workgroups (a grid with one workgroup per
    // group and column — 48 x 7 per block of eight items, six of seven returning at once — took 55 us per launch at BASELINE
    // config 3: 15 k workgroups, each a chain of dependent loads; now 2.5 k)
    const int wg = (int)blockIdx.y * (nslots - 1) + ((int)blockIdx.x - (i0 - base - lo)), nwg = (int)gridDim.y * (nslots - 1);
    for (int vb = wg * COMB_ROWS; vb < nrows_blk; vb += nwg * COMB_ROWS) {
    const int nrows = nrows_blk < vb + COMB_ROWS ? nrows_blk : vb + COMB_ROWS;
    for (int v = vb + w; v < nrows; v += 4) {
        const int64_t node = node0 + v;
        const int deg = indeg[node];
        const float inv = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg > 1 ? deg : 1);
        float x[COMB_MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < COMB_MAX_PER_LANE; ++c) {
            const int o = lane + 64 * c;
            x[c] = 0.f;
            if (o < d) {
                const float* __restrict__ p = partial + ((size_t)slot0 * BN + v) * d + o;
                const size_t sstr = (size_t)BN * d;                 // slot stride
                float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
                int j = 0;
                for (; j + 4 <= nslots; j += 4) {
                    t0 += p[(size_t)j * sstr];
                    t1 += p[(size_t)(j + 1) * sstr];
                    t2 += p[(size_t)(j + 2) * sstr];
                    t3 += p[(size_t)(j + 3) * sstr];
                }
                for (; j < nslots; ++j) t0 += p[(size_t)j * sstr];
                float t = ((t0 + t1) + (t2 + t3)) * inv;
                if (agg_out) agg_out[(size_t)node * d + o] = t;
                x[c] = no_tail ? ((no_tail & GHF_FLAG_ADD_H) ? t + h[(size_t)node * d + o] : t) : fmaxf(t + h[(size_t)node * d + o], 0.f);
                s += x[c];
            }
        }
        if (!no_tail) {
            const float mean = wave_sum(s) / (float)d;
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < COMB_MAX_PER_LANE; ++c)
                if (lane + 64 * c < d) { const float t = x[c] - mean; var += t * t; }
            const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)d + eps);
#pragma unroll
            for (int c = 0; c < COMB_MAX_PER_LANE; ++c) {
                const int o = lane + 64 * c;
                if (o < d) x[c] = (x[c] - mean) * rstd * g[o] + b[o];
            }
        }
#pragma unroll
        for (int c = 0; c < COMB_MAX_PER_LANE; ++c) {
            const int o = lane + 64 * c;
            if (o < d) h_out[(size_t)node * d + o] = x[c];
        }
        if (h_split_out) {                                 // SPLIT2H: the wave holds the whole row
            float mx = 0.f;
#pragma unroll
            for (int c = 0; c < COMB_MAX_PER_LANE; ++c)
                if (lane + 64 * c < d) mx = fmaxf(mx, fabsf(x[c]));
            const int sh = split2h_shift(wave_absmax(mx));
            const float up = pow2f(sh);
            _Float16* sp = (_Float16*)h_split_out + (size_t)node * 2 * d;
            int tiny = 0, nz = 0;
#pragma unroll
            for (int c = 0; c < COMB_MAX_PER_LANE; ++c) {
                const int o = lane + 64 * c;
                if (o < d) {
                    _Float16 hi, lo;
                    split2h(x[c] * up, hi, lo);
                    sp[o] = hi;
                    sp[d + o] = lo;
                    tiny += range_tiny(x[c] * up);
                    nz += x[c] != 0.f;
                }
            }
            tiny = (int)wave_sum((float)tiny);
            nz = (int)wave_sum((float)nz);
            if (lane == 0) range_raise(range_flag, GHF_RANGE_ROWS, tiny, nz);
            if (lane == 0) *(float*)((char*)h_split_out + (size_t)N * d * 4 + (size_t)node * 4) = pow2f(-sh);
        }
    }
    }
}

int launch_combine_split(const MsgArgs& a, hipStream_t stream) {
    const int64_t blk0 = a.row0 / a.block_nodes, row_end = a.row0 + a.rows;
    const int64_t nblk = cdiv(a.rows, a.block_nodes), extra = a.n_items - nblk;     // items beyond one per block
    if (extra <= 0) return GHF_OK;
    GHF_REQUIRE(extra < (1ll << 31), "combine_split: too many work items");
    // gridDim.y x (a block's items - 1) workgroups share a block's row groups: eight rows of them cover a block of eight items
    // (seven columns) in one pass, a block cut in two (one column) in six
    static const int ymax = getenv("GHF_COMB_Y") ? atoi(getenv("GHF_COMB_Y")) : 8;
    const dim3 grid((unsigned)extra, (unsigned)(cdiv(a.block_nodes, COMB_ROWS) < ymax ? cdiv(a.block_nodes, COMB_ROWS) : ymax));
    combine_split_kernel<<<grid, 256, 0, stream>>>(a.partial, a.item_tab, a.blk_item_off, a.h, a.indeg, a.ln_gamma, a.ln_beta,
                                                   a.ln_eps, a.N, a.d, a.block_nodes, blk0, (int)nblk, row_end, a.h_out, a.h_split_out, a.wlayout,
                                                   a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM | GHF_FLAG_ADD_H), range_flag_ptr(), a.agg_out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_message_generic(const MsgArgs& a, hipStream_t stream) {
    GHF_REQUIRE(a.block_nodes == 1, "message(generic): plan must be CSR (block_nodes == 1), got %d", a.block_nodes);
    GHF_REQUIRE(a.wlayout == GHF_WLAYOUT_NATURAL && a.W_self, "message(generic): needs NATURAL weights");
    GHF_REQUIRE(a.d >= 1 && a.d <= GEN_MAX_D, "message(generic): d=%d outside [1,%d]", a.d, GEN_MAX_D);
    GHF_REQUIRE(a.rows < (1ll << 31), "message(generic): too many rows per launch");
    if (a.rows <= 0) return GHF_OK;
    message_generic_kernel<<<(unsigned)a.rows, GEN_TPB, 0, stream>>>(
        a.h, a.d, a.sorted_key, a.sorted_src, a.seg_off, a.indeg, a.R, a.W_msg, a.W_self, a.bias,
        a.ln_gamma, a.ln_beta, a.ln_eps, a.row0, a.h_out, a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM));
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_tail(const float* agg, const float* h, const float* g, const float* b, float eps,
                int64_t row0, int64_t rows, int d, float* h_out, const float* drop, hipStream_t stream) {
    GHF_REQUIRE(d >= 1 && d <= GEN_MAX_D, "tail: d=%d outside [1,%d]", d, GEN_MAX_D);
    GHF_REQUIRE(rows < (1ll << 31), "tail: too many rows per launch");
    if (rows <= 0) return GHF_OK;
    tail_kernel<<<(unsigned)rows, GEN_TPB, 0, stream>>>(agg, h, g, b, eps, row0, d, h_out, drop);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
