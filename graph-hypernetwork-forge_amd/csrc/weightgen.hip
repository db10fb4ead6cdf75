// weightgen.hip — K1: relation-batched WeightGenerator forward.
//
// Replaces models/weight_generator.py:137-141 of the reference: for each head
// k in {W_msg, W_self, bias}:  flat_k = Lin_last(ReLU(Lin(...ReLU(Lin_0(x)))))
// (nn.Linear: y = x W^T + b, W stored [out,in]), out_k = flat_k * exp(log_scale_k).
//
//  wg_hidden_kernel : the num_hidden small Linear+ReLU layers, one workgroup per
//                     (relation, head), activations ping-ponged in LDS.
//  wg_out_mfma_kernel: the last layer [R,Hl] x [Hl, n_out] as an fp32 MFMA GEMM
//                     (v_mfma_f32_16x16x4_f32: exact fp32 fma chain), transposed so that the
//                     output-element index is the MFMA row and the relation the
//                     MFMA column; the epilogue scales by exp(log_scale) and stores
//                     either the reference's natural layout or the B-fragment
//                     layout the message kernel reads (GHF_WLAYOUT_FRAG16).
//  wg_out_simple_kernel: same contraction on the vector ALU for shapes the MFMA
//                     tile does not cover (Hl % 16 != 0) and for the tiny bias head.
#include "common.h"

namespace ghf {

constexpr int WG_MAX_WIDTH = 1024;   // max(T, Hh) supported by the LDS ping-pong buffers
constexpr int WG_UNROLL = 8;         // independent dot products per wave and step in the latency-bound small kernels
#ifndef GHF_WG_HU
#define GHF_WG_HU 8
#endif
constexpr int WG_HU = GHF_WG_HU;     // ... of wg_hidden_kernel (32 — a wave's whole share of a 128-unit layer at once — needs 264 registers: one workgroup per CU, 56 -> 130 us at config 3's 576 workgroups; 41 vs 43 us at config 2)

struct HeadPtrs {
    const float* w[3][8];     // [head][layer] weight
    const float* b[3][8];     // [head][layer] bias
};
constexpr int WG_MAX_L = 8;   // generators per batched launch (ghf_weightgen_fwd_batched): kernel arguments stay below 4 KB
struct HeadPtrsL { HeadPtrs p[WG_MAX_L]; };

// grid (R, 3); block 256.  hidden_ws[(head*R + r)*Hl + j], Hl = num_hidden ? Hh : T.
// acts (optional): every hidden layer's output, acts[((head*num_hidden + layer)*R + r)*Hh + j]  (the backward's input)
// blockIdx.z = generator (layer of the model) of a batched launch: its pointers P.p[z], its slice of hidden_ws
__global__ __launch_bounds__(256) void wg_hidden_kernel(const float* __restrict__ text_emb, HeadPtrsL PL,
                                                        int R, int T, int Hh, int num_hidden,
                                                        float* __restrict__ hidden_ws, float* __restrict__ acts,
                                                        const float* __restrict__ drop /* acts' layout, or NULL */) {
    __shared__ float buf[2][WG_MAX_WIDTH];
    const int r = blockIdx.x, head = blockIdx.y;
    const HeadPtrs& P = PL.p[blockIdx.z];
    if (hidden_ws) hidden_ws += (size_t)blockIdx.z * 3 * R * (num_hidden ? Hh : T);
    for (int k = threadIdx.x; k < T; k += blockDim.x) buf[0][k] = text_emb[(size_t)r * T + k];
    __syncthreads();
    int cur = 0, in_dim = T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int li = 0; li < num_hidden; ++li) {
        const float* __restrict__ W = P.w[head][li];
        const float* __restrict__ B = P.b[head][li];
        // one wave per output unit: lanes stride the contraction, then a wave reduction; WG_HU units at a time so
        // that their weight loads are in flight together (one at a time this kernel was 64 serial L2 latencies long)
        for (int j0 = wv * WG_HU; j0 < Hh; j0 += nw * WG_HU) {
            float s[WG_HU];
#pragma unroll
            for (int u = 0; u < WG_HU; ++u) s[u] = 0.f;
            for (int k = lane; k < in_dim; k += 64) {
                const float xk = buf[cur][k];
#pragma unroll
                for (int u = 0; u < WG_HU; ++u)
                    if (j0 + u < Hh) s[u] = fmaf(xk, W[(size_t)(j0 + u) * in_dim + k], s[u]);
            }
#pragma unroll
            for (int u = 0; u < WG_HU; ++u) {
                const float t = wave_sum(s[u]);
                if (lane == 0 && j0 + u < Hh) {             // Linear -> ReLU -> Dropout (reference weight_generator.py:96-107)
                    const float a = fmaxf(t + B[j0 + u], 0.f);
                    buf[cur ^ 1][j0 + u] = drop ? a * drop[(((size_t)head * num_hidden + li) * R + r) * Hh + j0 + u] : a;
                }
            }
        }
        __syncthreads();
        cur ^= 1;
        in_dim = Hh;
        if (acts) {
            float* a = acts + (((size_t)head * num_hidden + li) * R + r) * Hh;
            for (int k = threadIdx.x; k < Hh; k += blockDim.x) a[k] = buf[cur][k];
        }
    }
    if (!hidden_ws) return;
    float* out = hidden_ws + ((size_t)head * R + r) * in_dim;
    for (int k = threadIdx.x; k < in_dim; k += blockDim.x) out[k] = buf[cur][k];
}

// Destination index of element (r, kk, o) of the combined [W_msg; W_self] matrix of relation r
// in FRAG16 order: Wfrag[r][o/16][kk/16][lane = ((kk%16)/4)*16 + o%16][kk%4], kk in [0, 2d).
__device__ __forceinline__ size_t frag16_index(int r, int kk, int o, int d) {
    const int NT = d >> 4, NJ2 = d >> 3;
    return ((((size_t)r * NT + (o >> 4)) * NJ2 + (kk >> 4)) * 64 + (((kk & 15) >> 2) << 4) + (o & 15)) * 4 + (kk & 3);
}

// Vector-ALU last layer: one wave per output element n (lanes stride K), looping relations.
// grid (ceil(n_out / 4)), block 256 (4 waves).
__global__ __launch_bounds__(256) void wg_out_simple_kernel(const float* __restrict__ z /* [R,Hl] */,
                                                            const float* __restrict__ W3, const float* __restrict__ b3,
                                                            const float* __restrict__ log_scale,
                                                            int R, int Hl, int n_out, int head, int d_in, int d_out,
                                                            int layout, size_t rstride, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_out) return;
    const float scale = expf(log_scale[0]);
    const float bn = b3[n];
    for (int r0 = 0; r0 < R; r0 += WG_UNROLL) {
      float sv[WG_UNROLL];
#pragma unroll
      for (int u = 0; u < WG_UNROLL; ++u) sv[u] = 0.f;
      for (int k = lane; k < Hl; k += 64) {
          const float wk = W3[(size_t)n * Hl + k];
#pragma unroll
          for (int u = 0; u < WG_UNROLL; ++u)
              if (r0 + u < R) sv[u] = fmaf(z[(size_t)(r0 + u) * Hl + k], wk, sv[u]);
      }
#pragma unroll
      for (int u = 0; u < WG_UNROLL; ++u) {
        const int r = r0 + u;
        const float s = wave_sum(sv[u]);
        if (lane == 0 && r < R) {
            const float v = (s + bn) * scale;
            if (head == 2 || layout == GHF_WLAYOUT_NATURAL) {
                out[(size_t)r * rstride + n] = v;
            } else {
                const int i = n / d_out, o = n - i * d_out;
                out[frag16_index(r, head * d_in + i, o, d_out)] = v;
            }
        }
      }
    }
}

// MFMA last layer.  GEMM: D[m][c] = sum_k A[m][k] * B[k][c] with m = output element
// (row n of W3), c = relation, k = hidden unit.  One wave computes a 16(m) x 16(c) tile per
// relation tile; a workgroup of 4 waves covers 4 m-tiles.
//  A fragment (v_mfma_f32_16x16x4_f32): lane l supplies A[row l&15][k = l>>4] per step.  Each lane
//  loads 16 contiguous bytes W3[n(l&15)][16j + 4(l>>4) .. +3] and uses element s in step s,
//  i.e. step (j,s) contracts k = 16j + 4(l>>4) + s: a permutation of k that the B side
//  mirrors (B fragment: z[c = l&15][same k]).
//  m-tile -> n mapping: NATURAL: n = 16*mt + row (contiguous).  FRAG16: an m-tile is 16
//  consecutive input indices i for one output column o: n = (16*it + row)*d + o, so that a
//  lane's 4 accumulator registers (rows 4q..4q+3) are the 4 consecutive kk of one fragment
//  slot and are stored with one 16-byte store.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT>
__global__ __launch_bounds__(256) void wg_out_mfma_kernel(const float* __restrict__ z /* [R,Hl] */,
                                                          const float* __restrict__ W3, const float* __restrict__ b3,
                                                          const float* __restrict__ log_scale,
                                                          int R, int Hl, int n_out, int head, int d, size_t rstride,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int mt = blockIdx.x * 4 + wv;                 // m-tile index
    if (mt * 16 >= n_out) return;
    // row n handled by this lane as A-operand supplier (row = lane & 15)
    int n_a;
    int o = 0, it = 0;
    if (LAYOUT != GHF_WLAYOUT_NATURAL) {
        const int tiles_per_o = d >> 4;                  // i-tiles per output column
        o = mt / tiles_per_o;
        it = mt - o * tiles_per_o;
        n_a = (16 * it + c16) * d + o;
    } else {
        n_a = 16 * mt + c16;
    }
    const bool a_ok = n_a < n_out;
    const float* __restrict__ arow = W3 + (size_t)(a_ok ? n_a : 0) * Hl + 4 * q;
    const float scale = expf(log_scale[0]);
    const int NJ = Hl >> 4;

    for (int r0 = 0; r0 < R; r0 += 16) {
        const int rc = r0 + c16;                          // relation of this lane as B supplier / D column
        const bool b_ok = rc < R;
        const float* __restrict__ brow = z + (size_t)(b_ok ? rc : 0) * Hl + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < NJ; ++j) {
            f32x4 a = *(const f32x4*)(arow + 16 * j);
            f32x4 b = *(const f32x4*)(brow + 16 * j);
            if (!a_ok) a = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!b_ok) b = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
        }
        // D layout: lane holds rows 4q + reg (reg 0..3), column c16 (= relation rc)
        if (!b_ok) continue;
        if (LAYOUT == GHF_WLAYOUT_FRAG16) {
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int n = (16 * it + 4 * q + s) * d + o;
                v[s] = (acc[s] + b3[n]) * scale;
            }
            const int kk0 = head * d + 16 * it + 4 * q;   // kk of reg 0; kk & 3 == 0
            *(f32x4*)(out + frag16_index(rc, kk0, o, d)) = v;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int n = 16 * mt + 4 * q + s;
                if (n < n_out) out[(size_t)rc * rstride + n] = (acc[s] + b3[n]) * scale;
            }
        }
    }
}

// The three heads' last layers in ONE launch (natural order outputs; blockIdx.y = head): on the critical path of a forward
// the generator is a chain of small latency-bound kernels, and three of its five were these (round 3: 0.18 -> 0.11 ms).
struct OutHeads {
    const float* z[3];
    const float* W3[3];
    const float* b3[3];
    const float* log_scale[3];
    float* out[3];
    int n_out[3];
    size_t rstride[3];
};
struct OutHeadsL { OutHeads h[WG_MAX_L]; };
// NJT > 0: Hl = 16 NJT known at compile time — the wave's fragments of W3 are loaded ONCE (they do not depend on the row tile)
// and a row tile's fragments of z all at once: five round trips per wave instead of one per (row tile, k-step) — the loop
// below waited for its two loads in every one of its R/16 x Hl/16 iterations (config 3: 77 -> ~20 us).  Same products in the
// same order: the same bits.
template <int NJT>
__global__ __launch_bounds__(256) void wg_out_mfma3_kernel(OutHeadsL HL, int R, int Hl) {
    const int head = blockIdx.y % 3;
    const OutHeads& H = HL.h[blockIdx.y / 3];
    const float* __restrict__ z = H.z[head];
    const float* __restrict__ W3 = H.W3[head];
    const float* __restrict__ b3 = H.b3[head];
    float* __restrict__ out = H.out[head];
    const int n_out = H.n_out[head];
    const size_t rstride = H.rstride[head];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int mt = blockIdx.x * 4 + wv;
    if (mt * 16 >= n_out) return;
    const int n_a = 16 * mt + c16;
    const bool a_ok = n_a < n_out;
    const float* __restrict__ arow = W3 + (size_t)(a_ok ? n_a : 0) * Hl + 4 * q;
    const float scale = expf(H.log_scale[head][0]);
    const int NJ = Hl >> 4;
    if constexpr (NJT > 0) {
        f32x4 a[NJT];
#pragma unroll
        for (int j = 0; j < NJT; ++j) {
            a[j] = *(const f32x4*)(arow + 16 * j);
            if (!a_ok) a[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int r0 = 0; r0 < R; r0 += 16) {
            const int rc = r0 + c16;
            const bool b_ok = rc < R;
            const float* __restrict__ brow = z + (size_t)(b_ok ? rc : 0) * Hl + 4 * q;
            f32x4 b[NJT];
#pragma unroll
            for (int j = 0; j < NJT; ++j) {
                b[j] = *(const f32x4*)(brow + 16 * j);
                if (!b_ok) b[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJT; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][s], b[j][s], acc, 0, 0, 0);
            if (!b_ok) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int n = 16 * mt + 4 * q + s;
                if (n < n_out) out[(size_t)rc * rstride + n] = (acc[s] + b3[n]) * scale;
            }
        }
        return;
    }
    for (int r0 = 0; r0 < R; r0 += 16) {
        const int rc = r0 + c16;
        const bool b_ok = rc < R;
        const float* __restrict__ brow = z + (size_t)(b_ok ? rc : 0) * Hl + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < NJ; ++j) {
            f32x4 a = *(const f32x4*)(arow + 16 * j);
            f32x4 b = *(const f32x4*)(brow + 16 * j);
            if (!a_ok) a = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!b_ok) b = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
        }
        if (!b_ok) continue;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int n = 16 * mt + 4 * q + s;
            if (n < n_out) out[(size_t)rc * rstride + n] = (acc[s] + b3[n]) * scale;
        }
    }
}

// GHF_WLAYOUT_SPLIT2H, second step.  W holds [R][2d][d] fp32 ([W_msg[r]; W_self[r]] row-major); one workgroup per
// relation pulls its matrix into LDS, finds the largest magnitude, and rewrites the same bytes as fp16 B fragments
//   Wh[r][o/16][kk/32][piece][lane = ((kk%32)/8)*16 + o%16][kk%8],  piece 0 = fp16(w 2^s), piece 1 = fp16(w 2^s - piece 0)
// (4 bytes per weight either way: in place); scales[r] = 2^-s.
struct PackL { float* W[WG_MAX_L]; };     // per generator of a batched launch: its [R][2d][d] buffer, the scales behind it
__global__ __launch_bounds__(1024) void wg_pack2h_kernel(PackL WL, int R, int d, int32_t* __restrict__ range_flag) {
    float* __restrict__ W = WL.W[blockIdx.y];
    float* __restrict__ scales = W + (size_t)R * 2 * d * d;
    extern __shared__ float wbuf[];                      // [2d][d]
    __shared__ float red[16];
    __shared__ int cnt[2];
    __shared__ WeakRows weak_red[16];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    const int r = blockIdx.x, tid = threadIdx.x, n = 2 * d * d;
    float* __restrict__ mine = W + (size_t)r * n;
    float mx = 0.f;
    for (int i = tid; i < n / 4; i += 1024) {
        const float4 v = ((const float4*)mine)[i];
        ((float4*)wbuf)[i] = v;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    mx = wave_absmax(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();                                     // also: every load of `mine` happened before any store below
    mx = 0.f;
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[i]);
    {   // range guard, weak input rows (common.h: WeakRows): L1 norms of the 2d rows, a wave per row
        WeakRows wr;
        wr.init();
        const int lane = tid & 63, wv = tid >> 6;
        for (int row = wv; row < 2 * d; row += 16) {
            float s = 0.f;
            for (int o = lane; o < d; o += 64) s += fabsf(wbuf[row * d + o]);
            wr.add(row >= d, wave_sum(s));
        }
        if (lane == 0) weak_red[wv] = wr;
    }
    const int sh = split2h_shift(mx);
    const float up = pow2f(sh);
    if (tid == 0) scales[r] = pow2f(-sh);
    const int NKS = d >> 4;                              // k-steps of 32 over kk in [0, 2d)
    _Float16* __restrict__ dst = (_Float16*)mine;
    // one 16-byte granule (8 consecutive kk of one column) of one piece per thread and step
    for (int g = tid; g < n / 8; g += 1024) {
        const int slot = g & 63, fk = g >> 6;            // lane slot within the fragment; fragment = (ct, ks)
        const int ks = fk % NKS, ct = fk / NKS;
        const int o = ct * 16 + (slot & 15), kk0 = ks * 32 + (slot >> 4) * 8;
        _Float16 hi[8], lo[8];
        int tiny = 0, nz = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xs = wbuf[(kk0 + e) * d + o] * up;
            split2h(xs, hi[e], lo[e]);
            tiny += range_tiny(xs);
            nz += xs != 0.f;
        }
        _Float16* p = dst + ((size_t)fk * 2 * 64 + slot) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { p[e] = hi[e]; p[512 + e] = lo[e]; }
        if (tiny) atomicAdd(&cnt[0], tiny);              // range guard (common.h): this relation's matrix as a whole
        if (nz) atomicAdd(&cnt[1], nz);
    }
    __syncthreads();
    if (tid == 0) {
        range_raise(range_flag, GHF_RANGE_WEIGHTS, cnt[0], cnt[1]);
        WeakRows wr = weak_red[0];
        for (int i = 1; i < 16; ++i) wr.merge(weak_red[i]);
        range_raise_weak(range_flag, wr, d);
    }
}

int launch_weightgen_acts(const float* text_emb, const float* const* head_params, int R, int T, int Hh, int num_hidden,
                          float* acts, const float* hidden_drop, hipStream_t stream) {
    GHF_REQUIRE(R > 0 && T > 0 && num_hidden >= 0 && num_hidden <= 7, "weightgen_acts: bad shape");
    GHF_REQUIRE(T <= WG_MAX_WIDTH && Hh <= WG_MAX_WIDTH, "weightgen_acts: text_dim/hidden_dim > %d unsupported", WG_MAX_WIDTH);
    if (num_hidden == 0) return GHF_OK;
    HeadPtrsL PL;
    HeadPtrs& P = PL.p[0];
    const int nl = num_hidden + 1;
    for (int h = 0; h < 3; ++h)
        for (int l = 0; l < nl; ++l) {
            P.w[h][l] = head_params[(h * nl + l) * 2 + 0];
            P.b[h][l] = head_params[(h * nl + l) * 2 + 1];
        }
    wg_hidden_kernel<<<dim3(R, 3, 1), 256, 0, stream>>>(text_emb, PL, R, T, Hh, num_hidden, nullptr, acts, hidden_drop);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

// ghf_weights_pack: [W_top[r]; W_bottom[r]] (each [d,d] natural, optionally transposed, NULL = zeros) -> the [R][2d][d]
// fp32 matrix the in-place packer takes.  Used for the backward passes' transposed weights.
__global__ __launch_bounds__(256) void wg_combine_kernel(const float* __restrict__ top, const float* __restrict__ bottom,
                                                         int transpose, int d, float* __restrict__ out) {
    const int r = blockIdx.y;
    const int n = 2 * d * d;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int kk = idx / d, o = idx - kk * d;
        const float* src = kk < d ? top : bottom;
        const int k = kk < d ? kk : kk - d;
        float v = 0.f;
        if (src) v = transpose ? src[((size_t)r * d + o) * d + k] : src[((size_t)r * d + k) * d + o];
        out[(size_t)r * n + idx] = v;
    }
}

// the same stacked matrix straight into FRAG16 order (fp32, one float per weight: no second pass)
__global__ __launch_bounds__(256) void wg_combine_frag16_kernel(const float* __restrict__ top, const float* __restrict__ bottom,
                                                                int transpose, int d, float* __restrict__ out) {
    const int r = blockIdx.y;
    const int n = 2 * d * d;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int kk = idx / d, o = idx - kk * d;
        const float* src = kk < d ? top : bottom;
        const int k = kk < d ? kk : kk - d;
        float v = 0.f;
        if (src) v = transpose ? src[((size_t)r * d + o) * d + k] : src[((size_t)r * d + k) * d + o];
        out[frag16_index(r, kk, o, d)] = v;
    }
}

int launch_weights_pack(const float* top, const float* bottom, int transpose, int R, int d, int layout, float* out,
                        hipStream_t stream) {
    GHF_REQUIRE(R > 0 && d > 0 && out, "weights_pack: bad arguments");
    if (layout == GHF_WLAYOUT_FRAG16) {
        GHF_REQUIRE((d % 16) == 0, "weights_pack: FRAG16 needs d %% 16 == 0");
        wg_combine_frag16_kernel<<<dim3(32, (unsigned)R), 256, 0, stream>>>(top, bottom, transpose, d, out);
        GHF_LAUNCH_CHECK();
        return GHF_OK;
    }
    GHF_REQUIRE(layout == GHF_WLAYOUT_SPLIT2H, "weights_pack: layout %d is not packed from natural matrices here", layout);
    GHF_REQUIRE((d % 32) == 0 && (size_t)2 * d * d * 4 <= 128 * 1024, "weights_pack: SPLIT2H needs d %% 32 == 0, d <= 128");
    wg_combine_kernel<<<dim3(32, (unsigned)R), 256, 0, stream>>>(top, bottom, transpose, d, out);
    GHF_LAUNCH_CHECK();
    const size_t lds = (size_t)2 * d * d * 4;
    GHF_SET_MAX_LDS(wg_pack2h_kernel, lds);
    PackL WL;
    WL.W[0] = out;
    wg_pack2h_kernel<<<dim3(R, 1), 1024, lds, stream>>>(WL, R, d, range_flag_ptr());
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

// L generators of identical shape (the layers of one model) in one launch sequence: hidden layers (grid z = generator), the
// three heads' output layers (grid y = 3 generators), the SPLIT2H packing (grid y = generator).  head_params / log_scales:
// generator g's entries at [g * 3 * (num_hidden + 1) * 2 ...] / [g * 3 ...]; hidden_ws: L times the single-call size.
// Shapes or layouts the merged kernels do not cover run the per-head kernels generator by generator (same results).
int launch_weightgen_batched(int L, const float* text_emb, const float* const* head_params, const float* const* log_scales,
                             int R, int T, int Hh, int num_hidden, int d_in, int d_out, int layout,
                             float* hidden_ws, float* const* W_msg, float* const* W_self, float* const* bias,
                             const float* hidden_drop, hipStream_t stream, float* acts) {
    GHF_REQUIRE(L >= 1 && L <= WG_MAX_L, "weightgen: %d generators per call (1..%d)", L, WG_MAX_L);
    GHF_REQUIRE(R > 0 && T > 0 && d_in > 0 && d_out > 0, "weightgen: R, T, d_in, d_out must be positive");
    GHF_REQUIRE(num_hidden >= 0 && num_hidden <= 7, "weightgen: num_hidden=%d outside [0,7]", num_hidden);
    GHF_REQUIRE(num_hidden == 0 || Hh > 0, "weightgen: hidden_dim must be positive");
    GHF_REQUIRE(T <= WG_MAX_WIDTH && Hh <= WG_MAX_WIDTH, "weightgen: text_dim/hidden_dim > %d unsupported", WG_MAX_WIDTH);
    GHF_REQUIRE(layout == GHF_WLAYOUT_NATURAL || layout == GHF_WLAYOUT_FRAG16 || layout == GHF_WLAYOUT_SPLIT2H, "weightgen: bad layout %d", layout);
    GHF_REQUIRE((!hidden_drop && !acts) || L == 1, "weightgen: dropout masks / saved activations go with one generator per call");
    for (int g = 0; g < L; ++g) {
        GHF_REQUIRE(W_msg[g] && bias[g], "weightgen: null output pointer (generator %d)", g);
        if (layout == GHF_WLAYOUT_SPLIT2H)
            GHF_REQUIRE(d_in == d_out && (d_in % 32) == 0 && (size_t)2 * d_in * d_out * 4 <= 128 * 1024 && (!W_self || W_self[g] == nullptr),
                        "weightgen: SPLIT2H needs d_in == d_out, d %% 32 == 0, d <= 128 and W_self == NULL");
        if (layout == GHF_WLAYOUT_FRAG16)
            GHF_REQUIRE(d_in == d_out && (d_in % 16) == 0 && (!W_self || W_self[g] == nullptr),
                        "weightgen: FRAG16 needs d_in == d_out, d %% 16 == 0 and W_self == NULL");
        else if (layout == GHF_WLAYOUT_NATURAL)
            GHF_REQUIRE(W_self && W_self[g] != nullptr, "weightgen: NATURAL layout needs W_self");
    }
    HeadPtrsL PL;
    const int nl = num_hidden + 1;
    for (int g = 0; g < L; ++g)
        for (int h = 0; h < 3; ++h)
            for (int l = 0; l < nl; ++l) {
                PL.p[g].w[h][l] = head_params[((size_t)(g * 3 + h) * nl + l) * 2 + 0];
                PL.p[g].b[h][l] = head_params[((size_t)(g * 3 + h) * nl + l) * 2 + 1];
                GHF_REQUIRE(PL.p[g].w[h][l] && PL.p[g].b[h][l], "weightgen: null parameter pointer (generator %d head %d layer %d)", g, h, l);
            }
    wg_hidden_kernel<<<dim3(R, 3, L), 256, 0, stream>>>(text_emb, PL, R, T, Hh, num_hidden, hidden_ws, acts, hidden_drop);
    GHF_LAUNCH_CHECK();

    const int Hl = num_hidden ? Hh : T;
    const int n_mat = d_in * d_out;
    // natural-order outputs (NATURAL, and SPLIT2H before its packing step): all heads of all generators in one launch when the
    // MFMA tile applies to each of them
    bool merged = (layout == GHF_WLAYOUT_NATURAL || layout == GHF_WLAYOUT_SPLIT2H) && (Hl % 16) == 0;
    OutHeadsL HL;
    if (merged)
        for (int g = 0; g < L; ++g) {
            OutHeads& H = HL.h[g];
            for (int head = 0; head < 3; ++head) {
                H.z[head] = hidden_ws + ((size_t)g * 3 + head) * R * Hl;
                H.W3[head] = PL.p[g].w[head][num_hidden];
                H.b3[head] = PL.p[g].b[head][num_hidden];
                H.log_scale[head] = log_scales[g * 3 + head];
                GHF_REQUIRE(H.log_scale[head], "weightgen: null log-scale pointer (generator %d head %d)", g, head);
                H.n_out[head] = head == 2 ? d_out : n_mat;
                H.rstride[head] = (layout == GHF_WLAYOUT_SPLIT2H && head != 2) ? (size_t)2 * n_mat : (size_t)H.n_out[head];
                H.out[head] = head == 2 ? bias[g] : (layout == GHF_WLAYOUT_SPLIT2H ? W_msg[g] + (size_t)head * n_mat : (head == 0 ? W_msg[g] : W_self[g]));
                merged = merged && ((((uintptr_t)H.W3[head] | (uintptr_t)H.z[head]) & 15) == 0);
            }
        }
    if (merged) {
        const int mtiles = (n_mat + 15) / 16;
        const dim3 og((mtiles + 3) / 4, 3 * L);
        switch (Hl) {
            case 32: wg_out_mfma3_kernel<2><<<og, 256, 0, stream>>>(HL, R, Hl); break;
            case 64: wg_out_mfma3_kernel<4><<<og, 256, 0, stream>>>(HL, R, Hl); break;
            case 128: wg_out_mfma3_kernel<8><<<og, 256, 0, stream>>>(HL, R, Hl); break;
            case 256: wg_out_mfma3_kernel<16><<<og, 256, 0, stream>>>(HL, R, Hl); break;
            default: wg_out_mfma3_kernel<0><<<og, 256, 0, stream>>>(HL, R, Hl); break;
        }
        GHF_LAUNCH_CHECK();
    } else {
        for (int g = 0; g < L; ++g)
            for (int head = 0; head < 3; ++head) {
                const float* z = hidden_ws + ((size_t)g * 3 + head) * R * Hl;
                const float* W3 = PL.p[g].w[head][num_hidden];
                const float* b3 = PL.p[g].b[head][num_hidden];
                const float* ls = log_scales[g * 3 + head];
                GHF_REQUIRE(ls, "weightgen: null log-scale pointer (generator %d head %d)", g, head);
                const int n_out = head == 2 ? d_out : n_mat;
                // SPLIT2H: the two matrix heads first write [R][2d][d] fp32 into W_msg (natural order, W_self below W_msg),
                // which wg_pack2h_kernel then rewrites in place
                const bool nat = layout == GHF_WLAYOUT_NATURAL || layout == GHF_WLAYOUT_SPLIT2H;
                const size_t rstride = (layout == GHF_WLAYOUT_SPLIT2H && head != 2) ? (size_t)2 * n_mat : (size_t)n_out;
                float* out = head == 2 ? bias[g] : (layout == GHF_WLAYOUT_SPLIT2H ? W_msg[g] + (size_t)head * n_mat
                                                    : (layout != GHF_WLAYOUT_NATURAL ? W_msg[g] : (head == 0 ? W_msg[g] : W_self[g])));
                const int klayout = nat ? GHF_WLAYOUT_NATURAL : layout;
                // (the bias head runs the same MFMA chain in every layout: its values do not depend on the layout asked for)
                const bool mfma_ok = (Hl % 16) == 0 && ((((uintptr_t)W3 | (uintptr_t)z) & 15) == 0);
                if (mfma_ok && layout == GHF_WLAYOUT_FRAG16 && head != 2) {
                    const int mtiles = n_mat / 16;
                    wg_out_mfma_kernel<GHF_WLAYOUT_FRAG16><<<(mtiles + 3) / 4, 256, 0, stream>>>(z, W3, b3, ls, R, Hl, n_out, head, d_out, rstride, out);
                } else if (mfma_ok) {
                    const int mtiles = (n_out + 15) / 16;
                    wg_out_mfma_kernel<GHF_WLAYOUT_NATURAL><<<(mtiles + 3) / 4, 256, 0, stream>>>(z, W3, b3, ls, R, Hl, n_out, head, d_out, rstride, out);
                } else {
                    wg_out_simple_kernel<<<(n_out + 3) / 4, 256, 0, stream>>>(z, W3, b3, ls, R, Hl, n_out, head, d_in, d_out, klayout, rstride, out);
                }
                GHF_LAUNCH_CHECK();
            }
    }
    if (layout == GHF_WLAYOUT_SPLIT2H) {
        const size_t lds = (size_t)2 * n_mat * 4;
        PackL WL;
        for (int g = 0; g < L; ++g) WL.W[g] = W_msg[g];
        GHF_SET_MAX_LDS(wg_pack2h_kernel, lds);
        wg_pack2h_kernel<<<dim3(R, L), 1024, lds, stream>>>(WL, R, d_out, range_flag_ptr());
        GHF_LAUNCH_CHECK();
    }
    return GHF_OK;
}

int launch_weightgen(const float* text_emb, const float* const* head_params, const float* const* log_scales,
                     int R, int T, int Hh, int num_hidden, int d_in, int d_out, int layout,
                     float* hidden_ws, float* W_msg, float* W_self, float* bias, const float* hidden_drop, hipStream_t stream,
                     float* acts) {
    float* wm[1] = {W_msg};
    float* wsf[1] = {W_self};
    float* bs[1] = {bias};
    return launch_weightgen_batched(1, text_emb, head_params, log_scales, R, T, Hh, num_hidden, d_in, d_out, layout, hidden_ws,
                                    wm, wsf, bs, hidden_drop, stream, acts);
}

}  // namespace ghf
