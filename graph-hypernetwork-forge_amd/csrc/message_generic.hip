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

// Second stage for split destination blocks (hubs, and the blocks of a launch's last, partly filled round): sum the block's
// partial slots in a fixed order (reproducible), then the K3 tail.  Grid: (the row range's work items) x COMB_Y; an item of
// an unsplit block returns at once, item j of a block cut into k items takes the block's groups of rows j, j + k, ... (with
// blockIdx.y: j + k y, ...) — no search for the block, its id is in the item.  Two kernels: d % 4 == 0 up to 256 on
// 16-byte accesses, d / 4 lanes per row, every slot's load of a row in flight at once (round 3: the previous kernel — one
// wave per row, two columns per lane, four loads in flight — took 55-85 us per launch at BASELINE config 3 for 45 blocks of
// eight slots, 0.25 ms per forward); any d on the old scheme.
constexpr int COMB_MAX_PER_LANE = GEN_MAX_D / 64;
constexpr int COMB_ROWS = 8, COMB_Y = 6;
typedef float comb_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 comb_f16x4 __attribute__((ext_vector_type(4)));

struct CombBlock { int64_t blk; int nslots, slot0, j; bool live; };
__device__ __forceinline__ CombBlock comb_block(const int32_t* __restrict__ item_tab, const int32_t* __restrict__ blk_item_off, int64_t item0) {
    const int64_t it = item0 + blockIdx.x;
    const int64_t blk = item_tab[4 * (size_t)it];
    const int slot = item_tab[4 * (size_t)it + 3];
    CombBlock c{blk, 0, 0, 0, slot >= 0};
    if (!c.live) return c;
    const int i0 = blk_item_off[blk], i1 = blk_item_off[blk + 1];
    c.nslots = i1 - i0;
    c.j = (int)(it - i0);
    c.slot0 = item_tab[4 * (size_t)i0 + 3];
    return c;
}

template <int LPR>   // lanes per row: d = 4 LPR
__global__ __launch_bounds__(256) void combine_split4_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ item_tab, const int32_t* __restrict__ blk_item_off, int64_t item0,
    const float* __restrict__ h, const int32_t* __restrict__ indeg, const float* __restrict__ g,
    const float* __restrict__ b, float eps, int64_t N, int BN, int64_t row_end,
    float* __restrict__ h_out, void* __restrict__ h_split_out, int no_tail, int32_t* __restrict__ range_flag,
    float* __restrict__ agg_out) {
    constexpr int D = 4 * LPR, RPW = 64 / LPR;              // rows per wave and pass
    const CombBlock cb = comb_block(item_tab, blk_item_off, item0);
    if (!cb.live) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sub = lane / LPR, c0 = 4 * (lane % LPR);
    const int64_t node0 = cb.blk * BN;
    const int nrows_blk = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const int nwg = (int)gridDim.y * cb.nslots;
    const size_t sstr = (size_t)BN * D;                     // slot stride
    auto across = [&](float v, bool take_max) -> float {    // over the LPR lanes of a row
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) {
            const float o = __shfl_xor(v, off);
            v = take_max ? fmaxf(v, o) : v + o;
        }
        return v;
    };
    const comb_f32x4 gm = no_tail ? (comb_f32x4){1.f, 1.f, 1.f, 1.f} : *(const comb_f32x4*)(g + c0);
    const comb_f32x4 bt = no_tail ? (comb_f32x4){0.f, 0.f, 0.f, 0.f} : *(const comb_f32x4*)(b + c0);
    for (int vb = ((int)blockIdx.y * cb.nslots + cb.j) * (4 * RPW); vb < nrows_blk; vb += nwg * (4 * RPW)) {
        const int v = vb + RPW * w + sub;
        const bool live = v < nrows_blk;
        const int vc = live ? v : nrows_blk - 1;
        const int64_t node = node0 + vc;
        const float* __restrict__ p = partial + ((size_t)cb.slot0 * BN + vc) * D + c0;
        // every load of the row first: the slots (eight at a time), the in-degree, the residual row
        const int deg = indeg[node];
        const bool need_h = !no_tail || (no_tail & GHF_FLAG_ADD_H);
        const comb_f32x4 hv = need_h ? *(const comb_f32x4*)(h + (size_t)node * D + c0) : (comb_f32x4){0.f, 0.f, 0.f, 0.f};
        comb_f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < cb.nslots; j0 += 8) {
            comb_f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u < cb.nslots ? j0 + u : cb.nslots - 1;         // (past the end: the last slot again, added as zero)
                x[u] = *(const comb_f32x4*)(p + (size_t)j * sstr);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < cb.nslots) t += x[u];                             // slots in order: reproducible
        }
        const float inv = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg > 1 ? deg : 1);
        t *= inv;
        if (agg_out && live) *(comb_f32x4*)(agg_out + (size_t)node * D + c0) = t;
        comb_f32x4 y;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = no_tail ? t[e] + hv[e] : fmaxf(t[e] + hv[e], 0.f);          // (hv = 0 without ADD_H)
            s += y[e];
        }
        if (!no_tail) {
            const float mean = across(s, false) * (1.0f / D);
            float var = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float q = y[e] - mean; var += q * q; }
            const float rstd = 1.0f / sqrtf(across(var, false) * (1.0f / D) + eps);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (y[e] - mean) * rstd * gm[e] + bt[e];
        }
        if (live) *(comb_f32x4*)(h_out + (size_t)node * D + c0) = y;
        if (h_split_out) {                                 // SPLIT2H: the row's pieces and scale, and its range-guard count
            float mx = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(y[e]));
            const int sh = split2h_shift(across(mx, true));
            const float up = pow2f(sh);
            comb_f16x4 hi4, lo4;
            float tiny = 0.f, nz = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 hi, lo;
                split2h(y[e] * up, hi, lo);
                hi4[e] = hi;
                lo4[e] = lo;
                tiny += (float)range_tiny(y[e] * up);
                nz += y[e] != 0.f ? 1.f : 0.f;
            }
            tiny = across(tiny, false);
            nz = across(nz, false);
            if (live) {
                _Float16* sp = (_Float16*)h_split_out + (size_t)node * 2 * D + c0;
                *(comb_f16x4*)sp = hi4;
                *(comb_f16x4*)(sp + D) = lo4;
                if (lane % LPR == 0) {
                    *(float*)((char*)h_split_out + (size_t)N * D * 4 + (size_t)node * 4) = pow2f(-sh);
                    range_raise(range_flag, GHF_RANGE_ROWS, (int)tiny, (int)nz);
                }
            }
        }
    }
}

// any d: one wave per destination row, lanes stride the columns
__global__ __launch_bounds__(256) void combine_split_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ item_tab, const int32_t* __restrict__ blk_item_off, int64_t item0,
    const float* __restrict__ h, const int32_t* __restrict__ indeg, const float* __restrict__ g,
    const float* __restrict__ b, float eps, int64_t N, int d, int BN, int64_t row_end,
    float* __restrict__ h_out, void* __restrict__ h_split_out, int split_layout, int no_tail, int32_t* __restrict__ range_flag,
    float* __restrict__ agg_out) {
    const CombBlock cb = comb_block(item_tab, blk_item_off, item0);
    if (!cb.live) return;
    const int slot0 = cb.slot0, nslots = cb.nslots;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t node0 = cb.blk * BN;
    const int nrows_blk = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const int wg = (int)blockIdx.y * nslots + cb.j, nwg = (int)gridDim.y * nslots;
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
    const int64_t row_end = a.row0 + a.rows;
    const int64_t nblk = cdiv(a.rows, a.block_nodes);
    if (a.n_items <= nblk) return GHF_OK;                   // no block of the range is split
    GHF_REQUIRE(a.n_items < (1ll << 31), "combine_split: too many work items");
    const dim3 grid((unsigned)a.n_items, (unsigned)COMB_Y);
    const int nt = a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM | GHF_FLAG_ADD_H);
    const bool vec = (a.d % 4) == 0 && (a.d == 64 || a.d == 128 || a.d == 256) &&
                     (!a.h_split_out || a.wlayout == GHF_WLAYOUT_SPLIT2H);
#define GHF_COMB4(LPR)                                                                                                      \
    combine_split4_kernel<LPR><<<grid, 256, 0, stream>>>(a.partial, a.item_tab, a.blk_item_off, a.item0, a.h, a.indeg, a.ln_gamma, \
                                                        a.ln_beta, a.ln_eps, a.N, a.block_nodes, row_end, a.h_out, a.h_split_out, \
                                                        nt, range_flag_ptr(), a.agg_out)
    if (vec && a.d == 64) GHF_COMB4(16);
    else if (vec && a.d == 128) GHF_COMB4(32);
    else if (vec && a.d == 256) GHF_COMB4(64);
    else
        combine_split_kernel<<<grid, 256, 0, stream>>>(a.partial, a.item_tab, a.blk_item_off, a.item0, a.h, a.indeg, a.ln_gamma, a.ln_beta,
                                                       a.ln_eps, a.N, a.d, a.block_nodes, row_end, a.h_out, a.h_split_out, a.wlayout,
                                                       nt, range_flag_ptr(), a.agg_out);
#undef GHF_COMB4
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
