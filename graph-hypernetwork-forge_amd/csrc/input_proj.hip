// input_proj.hip — h0 = relu(x @ W_in^T + b_in)   (reference models/hypergnn.py:261).
//
// x [N,F] and W_in [d,F] are both row-major with the contraction index contiguous, so
// both MFMA operands load the same way: lane l reads 16 bytes at [row l&15][16j + 4(l>>4)]
// and uses element s in step s of v_mfma_f32_16x16x4_f32 (a k-permutation shared by A and
// B; exact fp32 fma chain).  A wave owns IP_MT*16 rows and all d output columns, so each
// W_in fragment loaded from L1/L2 is reused IP_MT times and x is read from HBM once.
// The product is formed TRANSPOSED (W_in's fragment as the A operand, x's as B): a lane then holds four CONSECUTIVE output
// columns of one row, so h0 goes out in 16-byte stores and its fp16 pieces in 8-byte stores (with x as the A operand a
// lane held one column of four rows: 4- and 2-byte stores, 0.61 ms at C3 against 0.44 without the pieces).
// Shapes the tile does not cover (F % 16, d % 16, d > 256) take a vector-ALU kernel.
#include "common.h"

#include <stdlib.h>
#include <string.h>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ip_f16x4 __attribute__((ext_vector_type(4)));

constexpr int IP_MT = 2;       // m-tiles (16 rows each) per wave
#ifndef GHF_IPEXP
#define GHF_IPEXP 0           // timing experiments only (GHF_VARIANT=ipexp<mask>): 1 no fp32 stores, 2 no piece stores, 4 no range guard
#endif

// SPLIT: also write the rows in GHF_WLAYOUT_SPLIT2H form (ghf_split_rows) for the first message layer's gathers: a row
// lives in the four lanes {c16, c16 + 16, c16 + 32, c16 + 48}, so its largest magnitude is two lane exchanges away.
// WLDS: W_in staged once per workgroup in LDS (rows padded by four floats: the 16 lanes of a fragment read then cover
// every bank once) and the workgroup walks row tiles — with W_in's fragments coming from L2 one column tile ahead the
// matrix pipe was 48 % busy (0.44 ms at C3 for 33 GFLOP)
template <int NT, bool SPLIT, bool WLDS>  // n-tiles of 16 output columns: d = 16*NT
__global__ __launch_bounds__(256) void input_proj_mfma_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int64_t N, int F,
                                                              float* __restrict__ h0, char* __restrict__ h_split,
                                                              int32_t* __restrict__ range_flag) {
    constexpr int D = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) float ip_w[];      // WLDS: [D][F + 4]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int wp = WLDS ? F + 4 : F;                                  // pitch of a W row where the fragments are read
    if (WLDS) {
        for (int i = threadIdx.x; i < D * (F >> 2); i += 256) {
            const int row = i / (F >> 2), c4 = i - row * (F >> 2);
            *(f32x4*)(ip_w + (size_t)row * wp + 4 * c4) = *(const f32x4*)(W + (size_t)row * F + 4 * c4);
        }
        __syncthreads();
    }
    const int64_t ntiles = (N + 16 * IP_MT - 1) / (16 * IP_MT);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += WLDS ? (int64_t)gridDim.x * 4 : ntiles) {
    const int64_t row_base = tile * (16 * IP_MT);

    const float* arow[IP_MT];
#pragma unroll
    for (int m = 0; m < IP_MT; ++m) {
        int64_t r = row_base + 16 * m + c16;
        if (r >= N) r = N - 1;                           // clamp: rows past N are computed but not stored
        arow[m] = x + (size_t)r * F + 4 * q;
    }
    f32x4 acc[IP_MT][NT];                                // [row tile][column tile]: element s = column 16t + 4q + s of row c16
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 bv = *(const f32x4*)(bias + 16 * t + 4 * q);
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) acc[m][t] = bv;
    }
    // x fragments one 16-wide k-slice ahead (x streams from HBM once); W fragments from LDS (or, shapes too big for it, L2)
    const int NJ = F >> 4;
    const float* __restrict__ wrow = (WLDS ? ip_w : W) + (size_t)c16 * wp + 4 * q;
    f32x4 a[IP_MT], an[IP_MT];
#pragma unroll
    for (int m = 0; m < IP_MT; ++m) a[m] = *(const f32x4*)(arow[m]);
    // column tiles in groups of TG: the MFMAs of a group go round its TG * IP_MT accumulators, so that consecutive ones are
    // independent (two chains per wave — s inside one accumulator — left the pipe waiting for results: 52 % busy)
    constexpr int TG = NT % 4 == 0 ? 4 : (NT % 2 == 0 ? 2 : 1);
    for (int j = 0; j < NJ; ++j) {
        const int jn = j + 1 < NJ ? j + 1 : j;            // (the last slice is read twice: no branch around the loads)
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) an[m] = *(const f32x4*)(arow[m] + 16 * jn);
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TG) {
            f32x4 b[TG];
#pragma unroll
            for (int u = 0; u < TG; ++u) b[u] = *(const f32x4*)(wrow + (size_t)16 * (t0 + u) * wp + 16 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < TG; ++u)
#pragma unroll
                    for (int m = 0; m < IP_MT; ++m)      // D[i = output column][j = row] = sum_k W[i][k] x[j][k]
                        acc[m][t0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][s], a[m][s], acc[m][t0 + u], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) a[m] = an[m];
    }
    auto across_q = [&](float v, bool take_max) -> float {     // over the four lanes that hold one row
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            const float o = __shfl_xor(v, off);
            v = take_max ? fmaxf(v, o) : v + o;
        }
        return v;
    };
#pragma unroll
    for (int m = 0; m < IP_MT; ++m) {
        const int64_t r = row_base + 16 * m + c16;
        f32x4 v[NT];
        float mx = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v[t][s] = fmaxf(acc[m][t][s], 0.f);
                mx = fmaxf(mx, v[t][s]);
            }
        float up = 1.f;
        if (SPLIT) {                                      // (all lanes take part in the reduction)
            const int sh = split2h_shift(across_q(mx, true));
            up = pow2f(sh);
            if (r < N && q == 0) *(float*)(h_split + (size_t)N * (4 * D) + (size_t)r * 4) = pow2f(-sh);
        }
        if (r < N) {
            float* __restrict__ o = h0 + (size_t)r * D + 4 * q;
            if (!(GHF_IPEXP & 1))
#pragma unroll
                for (int t = 0; t < NT; ++t) *(f32x4*)(o + 16 * t) = v[t];
            if (SPLIT && !(GHF_IPEXP & 2)) {
                _Float16* __restrict__ sp = (_Float16*)(h_split + (size_t)r * (4 * D)) + 4 * q;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    ip_f16x4 hi4, lo4;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        _Float16 hi, lo;
                        split2h(v[t][s] * up, hi, lo);
                        hi4[s] = hi;
                        lo4[s] = lo;
                    }
                    *(ip_f16x4*)(sp + 16 * t) = hi4;
                    *(ip_f16x4*)(sp + D + 16 * t) = lo4;
                }
            }
        }
        if (SPLIT && !(GHF_IPEXP & 4)) {                  // range guard (common.h); all lanes take part in the reduction
            float tiny = 0.f, nz = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) { tiny += (float)range_tiny(v[t][s] * up); nz += v[t][s] != 0.f ? 1.f : 0.f; }
            if (__ballot(tiny != 0.f)) {
                tiny = across_q(tiny, false);
                nz = across_q(nz, false);
                if (r < N && q == 0) range_raise(range_flag, GHF_RANGE_ROWS, (int)tiny, (int)nz);
            }
        }
    }
  }
}

// ---- the projection on two fp16 pieces per operand (round 3) -----------------------------------------------------------
// The fp32-MFMA kernel above is bound by the fp32 matrix rate (33 GFLOP at C3: 0.21 ms at its peak, 0.5 ms measured) where
// the bytes it moves (x once, h0 and its pieces once: 1.5 GB) are worth 0.25 ms.  Callers that ask for the SPLIT2H rows are
// on the two-piece path anyway (their message kernels contract the same pieces, under the same range guard: ghf.h), so for
// them the projection runs the message kernels' contraction: x rows scaled by a power of two per row and cut into hi + lo
// fp16 in registers, W_in cut once per workgroup into LDS (one power of two for the matrix), hi*hi + hi*lo + lo*hi by
// v_mfma_f32_16x16x32_f16 in fp32 accumulators, the exact scales taken out before bias and ReLU: 3/16 of the matrix time.
// A row or W_in with too much dynamic range for two pieces raises the guard bits (ghf.h) and the forward is repeated on the
// exact kernels — the fp32 kernel above among them (callers that ask for no split rows get it).
typedef _Float16 ip_f16x8 __attribute__((ext_vector_type(8)));
typedef int ip_i32x4 __attribute__((ext_vector_type(4)));
template <int NT, int KS>   // d = 16 NT <= 128, F = 32 KS <= 128
__global__ __launch_bounds__(256, 2) void input_proj_h_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int64_t N,
                                                              float* __restrict__ h0, char* __restrict__ h_split,
                                                              int32_t* __restrict__ range_flag) {
    constexpr int D = 16 * NT, F = 32 * KS, WP = F + 8;               // LDS pitch of a W row in halfs: +16 bytes, the 16 rows of
    extern __shared__ __attribute__((aligned(16))) char ip_raw[];     // a fragment read then cover every bank once
    _Float16* wp = (_Float16*)ip_raw;                                 // [piece][D][WP]
    __shared__ float red[4];
    __shared__ int wcnt[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    // ---- W_in -> two fp16 planes in LDS, one power-of-two scale for the matrix ----
    constexpr int WPT = D * F / 4 / 256 > 0 ? D * F / 4 / 256 : 1;    // float4 per thread (D F / 1024: 16 at 128 x 128)
    f32x4 wr[WPT];
    float wmx = 0.f;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int e = threadIdx.x + 256 * i;
        wr[i] = e < D * F / 4 ? *(const f32x4*)(W + 4 * (size_t)e) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) wmx = fmaxf(wmx, fabsf(wr[i][c]));
    }
    wmx = wave_absmax(wmx);
    if (threadIdx.x < 2) wcnt[threadIdx.x] = 0;
    if (lane == 0) red[wv] = wmx;
    __syncthreads();
    const int wsh = split2h_shift(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    const float wup = pow2f(wsh), winv = pow2f(-wsh);
    int wtiny = 0, wnz = 0;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < D * F / 4) {
            const int row = (4 * e) / F, k = (4 * e) % F;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                _Float16 hi, lo;
                const float xs = wr[i][c] * wup;
                split2h(xs, hi, lo);
                wp[(size_t)row * WP + k + c] = hi;
                wp[(size_t)(D + row) * WP + k + c] = lo;
                wtiny += range_tiny(xs);
                wnz += xs != 0.f;
            }
        }
    }
    if (blockIdx.x == 0 && range_flag) {                 // the range guard's view of W_in (one workgroup speaks for all)
        if (wtiny) atomicAdd(&wcnt[0], wtiny);
        if (wnz) atomicAdd(&wcnt[1], wnz);
    }
    __syncthreads();
    if (blockIdx.x == 0 && range_flag) {
        if (threadIdx.x == 0) range_raise(range_flag, GHF_RANGE_WEIGHTS, wcnt[0], wcnt[1]);
        // weak input columns (common.h: WeakRows — here the contraction index is the feature k): L1 norm of W_in[:, k]
        WeakRows wk;
        wk.init();
        for (int k = wv; k < F; k += 4) {
            float sm = 0.f;
            for (int o = lane; o < D; o += 64) sm += fabsf((float)wp[(size_t)o * WP + k]);
            wk.add(0, wave_sum(sm));
        }
        __shared__ WeakRows wkr[4];
        if (lane == 0) wkr[wv] = wk;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < 4; ++i) wk.merge(wkr[i]);
            range_raise_weak(range_flag, wk, F < D ? D : F);
        }
    }
    const int64_t ntiles = (N + 16 * IP_MT - 1) / (16 * IP_MT);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row_base = tile * (16 * IP_MT);
        // ---- this wave's 32 rows of x: lane (c16, q) holds columns 32 j + 8 q .. + 7 of row c16 of each row tile ----
        f32x4 xr[IP_MT][KS][2];
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) {
            int64_t r = row_base + 16 * m + c16;
            if (r >= N) r = N - 1;                       // clamp: rows past N are computed but not stored
            const float* __restrict__ p = x + (size_t)r * F + 8 * q;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                xr[m][j][0] = *(const f32x4*)(p + 32 * j);
                xr[m][j][1] = *(const f32x4*)(p + 32 * j + 4);
            }
        }
        auto across_q = [&](float v, bool take_max) -> float {     // over the four lanes that hold one row
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                const float o = __shfl_xor(v, off);
                v = take_max ? fmaxf(v, o) : v + o;
            }
            return v;
        };
        ip_i32x4 xh[IP_MT][KS], xl[IP_MT][KS];
        float xinv[IP_MT];
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) {
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < KS; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) mx = fmaxf(mx, fabsf(xr[m][j][u][c]));
            const int sh = split2h_shift(across_q(mx, true));
            const float up = pow2f(sh);
            xinv[m] = pow2f(-sh) * winv;
            float tiny = 0.f, nz = 0.f;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                ip_f16x8 hi8, lo8;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        _Float16 hi, lo;
                        const float xs = xr[m][j][u][c] * up;
                        split2h(xs, hi, lo);
                        hi8[4 * u + c] = hi;
                        lo8[4 * u + c] = lo;
                        tiny += (float)range_tiny(xs);
                        nz += xs != 0.f ? 1.f : 0.f;
                    }
                xh[m][j] = __builtin_bit_cast(ip_i32x4, hi8);
                xl[m][j] = __builtin_bit_cast(ip_i32x4, lo8);
            }
            if (__ballot(tiny != 0.f)) {                 // range guard (common.h): the rows of x as the contraction sees them
                tiny = across_q(tiny, false);
                nz = across_q(nz, false);
                if (row_base + 16 * m + c16 < N && q == 0) range_raise(range_flag, GHF_RANGE_ROWS, (int)tiny, (int)nz);
            }
        }
        f32x4 acc[IP_MT][NT];
#pragma unroll
        for (int m = 0; m < IP_MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (the W fragments are the same for every tile: an opaque zero in their address keeps hipcc from hoisting all 2 KS NT
        // of them — 256 registers — out of the tile loop and spilling them)
        int zofs;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zofs));
        const _Float16* wbase = wp + (size_t)c16 * WP + 8 * q + zofs;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const _Float16* wrow = wbase + (size_t)(16 * t) * WP + 32 * j;
                const ip_i32x4 wh = *(const ip_i32x4*)wrow, wl = *(const ip_i32x4*)(wrow + (size_t)D * WP);
#pragma unroll
                for (int m = 0; m < IP_MT; ++m) {        // D[i = output column][j = row] = sum_k W[i][k] x[j][k]
                    auto fma = [&](const ip_i32x4& a, const ip_i32x4& b) {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ip_f16x8, a), __builtin_bit_cast(ip_f16x8, b),
                                                                         acc[m][t], 0, 0, 0);
                    };
                    fma(wl, xh[m][j]); fma(wh, xl[m][j]);    // lo*hi, hi*lo
                    fma(wh, xh[m][j]);                       // hi*hi
                }
            }
        // ---- epilogue: scales out, bias, ReLU, h0 and its pieces (as the fp32 kernel's) ----
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) {
            const int64_t r = row_base + 16 * m + c16;
            f32x4 v[NT];
            float mx = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 bv = *(const f32x4*)(bias + 16 * t + 4 * q);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) {
                    v[t][sI] = fmaxf(fmaf(acc[m][t][sI], xinv[m], bv[sI]), 0.f);
                    mx = fmaxf(mx, v[t][sI]);
                }
            }
            const int sh = split2h_shift(across_q(mx, true));
            const float up = pow2f(sh);
            if (r < N && q == 0) *(float*)(h_split + (size_t)N * (4 * D) + (size_t)r * 4) = pow2f(-sh);
            if (r < N) {
                float* __restrict__ o = h0 + (size_t)r * D + 4 * q;
#pragma unroll
                for (int t = 0; t < NT; ++t) *(f32x4*)(o + 16 * t) = v[t];
                _Float16* __restrict__ sp = (_Float16*)(h_split + (size_t)r * (4 * D)) + 4 * q;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    ip_f16x4 hi4, lo4;
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) {
                        _Float16 hi, lo;
                        split2h(v[t][sI] * up, hi, lo);
                        hi4[sI] = hi;
                        lo4[sI] = lo;
                    }
                    *(ip_f16x4*)(sp + 16 * t) = hi4;
                    *(ip_f16x4*)(sp + D + 16 * t) = lo4;
                }
            }
            float tiny = 0.f, nz = 0.f;                  // range guard on the rows of h0 (common.h)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) { tiny += (float)range_tiny(v[t][sI] * up); nz += v[t][sI] != 0.f ? 1.f : 0.f; }
            if (__ballot(tiny != 0.f)) {
                tiny = across_q(tiny, false);
                nz = across_q(nz, false);
                if (r < N && q == 0) range_raise(range_flag, GHF_RANGE_ROWS, (int)tiny, (int)nz);
            }
        }
    }
}

// Fallback: one wave per row, lanes stride output columns.
__global__ __launch_bounds__(256) void input_proj_simple_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ bias, int64_t N, int F, int d,
                                                                float* __restrict__ h0) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const float* __restrict__ xr = x + (size_t)r * F;
    for (int o = lane; o < d; o += 64) {
        float s = bias[o];
        const float* __restrict__ wr = W + (size_t)o * F;
        for (int k = 0; k < F; ++k) s = fmaf(xr[k], wr[k], s);
        h0[(size_t)r * d + o] = fmaxf(s, 0.f);
    }
}

int launch_input_proj(const float* x, const float* W_in, const float* b_in, int64_t N, int F, int d,
                      float* h0, void* h_split, int split_layout, hipStream_t stream) {
    GHF_REQUIRE(N > 0 && F > 0 && d > 0, "input_proj: N, F, d must be positive");
    GHF_REQUIRE(!h_split || split_layout == GHF_WLAYOUT_SPLIT2H, "input_proj: h_split needs the SPLIT2H layout, got %d", split_layout);
    const bool fuse = h_split && split_layout == GHF_WLAYOUT_SPLIT2H;
    const bool aligned = ((((uintptr_t)x | (uintptr_t)W_in | (uintptr_t)b_in | (uintptr_t)h0 | (uintptr_t)h_split) & 15) == 0);
    const bool mfma_ok = aligned && (F % 16) == 0 && (d % 16) == 0 && d <= 256;
    // callers on the two-piece path (they ask for the SPLIT2H rows): the projection on fp16 pieces too, where its tile fits;
    // GHF_INPUT_PROJ=exact keeps the fp32 MFMAs (A/B)
    static const bool ip_exact = getenv("GHF_INPUT_PROJ") && !strcmp(getenv("GHF_INPUT_PROJ"), "exact");
    const bool pieces_ok = fuse && aligned && !ip_exact && (F % 32) == 0 && F <= 128 && (d % 16) == 0 && d <= 128 && N >= 1024;
    if (pieces_ok) {
        const int64_t nblk = cdiv(N, (int64_t)4 * 16 * IP_MT);
        const unsigned grid = (unsigned)(nblk < 512 ? nblk : 512);       // two workgroups per CU, each cuts W_in once and walks its tiles
        const size_t lds = (size_t)2 * d * (F + 8) * 2;
        bool done = true;
        switch ((d / 16) * 8 + F / 32) {
#define GHF_IPH_CASE(NT, KS)                                                                                            \
    case (NT) * 8 + (KS):                                                                                               \
        GHF_SET_MAX_LDS((input_proj_h_kernel<NT, KS>), 72 * 1024);                                                      \
        input_proj_h_kernel<NT, KS><<<grid, 256, lds, stream>>>(x, W_in, b_in, N, h0, (char*)h_split, range_flag_ptr()); \
        break;
            GHF_IPH_CASE(8, 4) GHF_IPH_CASE(8, 2) GHF_IPH_CASE(8, 1) GHF_IPH_CASE(8, 3)
            GHF_IPH_CASE(4, 4) GHF_IPH_CASE(4, 2) GHF_IPH_CASE(4, 1) GHF_IPH_CASE(4, 3)
#undef GHF_IPH_CASE
            default: done = false;
        }
        if (done) {
            GHF_LAUNCH_CHECK();
            return GHF_OK;
        }
    }
    if (mfma_ok) {
        const int64_t rows_per_block = 4 * 16 * IP_MT;
        const int64_t nblk = cdiv(N, rows_per_block);
        const size_t wlds = (size_t)d * (F + 4) * 4;                      // W_in in LDS when it leaves room for two workgroups per CU
        const bool use_lds = wlds <= 72 * 1024 && nblk >= 64;
        const unsigned grid = (unsigned)(use_lds ? (nblk < 1024 ? nblk : 1024) : nblk);
        const size_t lds = use_lds ? wlds : 0;
        switch (d / 16) {
#define GHF_IP_LAUNCH(NT, SP, WL, ...)                                                                                  \
    do {                                                                                                                \
        GHF_SET_MAX_LDS((input_proj_mfma_kernel<NT, SP, WL>), 72 * 1024);                                               \
        input_proj_mfma_kernel<NT, SP, WL><<<grid, 256, lds, stream>>>(__VA_ARGS__);                                    \
    } while (0)
#define GHF_IP_CASE(NT)                                                                                                 \
    case NT:                                                                                                            \
        if (fuse && use_lds) GHF_IP_LAUNCH(NT, true, true, x, W_in, b_in, N, F, h0, (char*)h_split, range_flag_ptr());   \
        else if (fuse) GHF_IP_LAUNCH(NT, true, false, x, W_in, b_in, N, F, h0, (char*)h_split, range_flag_ptr());        \
        else if (use_lds) GHF_IP_LAUNCH(NT, false, true, x, W_in, b_in, N, F, h0, nullptr, nullptr);                     \
        else GHF_IP_LAUNCH(NT, false, false, x, W_in, b_in, N, F, h0, nullptr, nullptr);                                 \
        break;
            GHF_IP_CASE(1) GHF_IP_CASE(2) GHF_IP_CASE(3) GHF_IP_CASE(4) GHF_IP_CASE(5) GHF_IP_CASE(6)
            GHF_IP_CASE(7) GHF_IP_CASE(8) GHF_IP_CASE(9) GHF_IP_CASE(10) GHF_IP_CASE(11) GHF_IP_CASE(12)
            GHF_IP_CASE(13) GHF_IP_CASE(14) GHF_IP_CASE(15) GHF_IP_CASE(16)
#undef GHF_IP_CASE
#undef GHF_IP_LAUNCH
        }
    } else {
        input_proj_simple_kernel<<<(unsigned)cdiv(N, 4), 256, 0, stream>>>(x, W_in, b_in, N, F, d, h0);
    }
    GHF_LAUNCH_CHECK();
    if (h_split && !(fuse && mfma_ok)) {                  // shapes / layouts without the fused epilogue: a separate pass
        return launch_split2h_rows(h0, N, d, 0, N, h_split, stream);
    }
    return GHF_OK;
}

}  // namespace ghf
