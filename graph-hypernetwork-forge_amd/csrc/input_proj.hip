// input_proj.hip — h0 = relu(x @ W_in^T + b_in)   (reference models/hypergnn.py:261).
//
// x [N,F] and W_in [d,F] are both row-major with the contraction index contiguous, so
// both MFMA operands load the same way: lane l reads 16 bytes at [row l&15][16j + 4(l>>4)]
// and uses element s in step s of v_mfma_f32_16x16x4_f32 (a k-permutation shared by A and
// B; exact fp32 fma chain).  A wave owns IP_MT*16 rows and all d output columns, so each
// W_in fragment loaded from L1/L2 is reused IP_MT times and x is read from HBM once.
// Shapes the tile does not cover (F % 16, d % 16, d > 256) take a vector-ALU kernel.
#include "common.h"

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IP_MT = 2;       // m-tiles (16 rows each) per wave

// SPLIT: also write the rows in GHF_WLAYOUT_SPLIT2H form (ghf_split_rows) for the first message layer's gathers:
// a row lives in the 16 lanes of one DPP row, so its largest magnitude is four DPP steps away.
template <int NT, bool SPLIT>  // n-tiles of 16 output columns: d = 16*NT
__global__ __launch_bounds__(256) void input_proj_mfma_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int64_t N, int F,
                                                              float* __restrict__ h0, char* __restrict__ h_split,
                                                              int32_t* __restrict__ range_flag) {
    constexpr int D = 16 * NT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int64_t row_base = ((int64_t)blockIdx.x * 4 + wv) * (16 * IP_MT);
    if (row_base >= N) return;

    const float* arow[IP_MT];
#pragma unroll
    for (int m = 0; m < IP_MT; ++m) {
        int64_t r = row_base + 16 * m + c16;
        if (r >= N) r = N - 1;                           // clamp: rows past N are computed but not stored
        arow[m] = x + (size_t)r * F + 4 * q;
    }
    f32x4 acc[IP_MT][NT];
#pragma unroll
    for (int m = 0; m < IP_MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float bv = bias[16 * t + c16];         // D column = lane & 15
            acc[m][t] = (f32x4){bv, bv, bv, bv};
        }
    // x fragments one 16-wide k-slice ahead, W fragments one column tile ahead: the MFMAs of a tile (256 cycles) cover
    // both loads (W_in is L1/L2 resident; x streams from HBM once)
    const int NJ = F >> 4;
    const float* __restrict__ wrow = W + (size_t)c16 * F + 4 * q;
    f32x4 a[IP_MT], an[IP_MT];
#pragma unroll
    for (int m = 0; m < IP_MT; ++m) a[m] = *(const f32x4*)(arow[m]);
    f32x4 b = *(const f32x4*)(wrow);
    for (int j = 0; j < NJ; ++j) {
        const int jn = j + 1 < NJ ? j + 1 : j;            // (the last slice is read twice: no branch around the loads)
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) an[m] = *(const f32x4*)(arow[m] + 16 * jn);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 bn = t + 1 < NT ? *(const f32x4*)(wrow + (size_t)16 * (t + 1) * F + 16 * j)
                                        : *(const f32x4*)(wrow + 16 * jn);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < IP_MT; ++m)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[s], acc[m][t], 0, 0, 0);
            b = bn;
        }
#pragma unroll
        for (int m = 0; m < IP_MT; ++m) a[m] = an[m];
    }
    // D: lane holds rows 4q + reg, column 16t + c16
#pragma unroll
    for (int m = 0; m < IP_MT; ++m)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t r = row_base + 16 * m + 4 * q + s;
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = fmaxf(acc[m][t][s], 0.f);
            float up = 1.f;
            if (SPLIT) {                                  // (all lanes take part in the reduction)
                float mx = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) mx = fmaxf(mx, v[t]);
                mx = fmaxf(mx, dpp_take<0xB1, 0xF>(mx));
                mx = fmaxf(mx, dpp_take<0x4E, 0xF>(mx));
                mx = fmaxf(mx, dpp_take<0x141, 0xF>(mx));
                mx = fmaxf(mx, dpp_take<0x140, 0xF>(mx));  // every lane of the 16 holds the row's maximum
                const int sh = split2h_shift(mx);
                up = pow2f(sh);
                if (r < N && c16 == 0) *(float*)(h_split + (size_t)N * (4 * D) + (size_t)r * 4) = pow2f(-sh);
            }
            if (r < N) {
#pragma unroll
                for (int t = 0; t < NT; ++t) h0[(size_t)r * D + 16 * t + c16] = v[t];
                if (SPLIT) {
                    _Float16* __restrict__ sp = (_Float16*)(h_split + (size_t)r * (4 * D));
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        _Float16 hi, lo;
                        split2h(v[t] * up, hi, lo);
                        sp[16 * t + c16] = hi;
                        sp[D + 16 * t + c16] = lo;
                    }
                }
            }
            if (SPLIT) {                                  // range guard (common.h); all lanes take part in the reduction
                float tiny = 0.f, nz = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) { tiny += (float)range_tiny(v[t] * up); nz += v[t] != 0.f ? 1.f : 0.f; }
                if (__ballot(tiny != 0.f)) {
                    for (int i = 0; i < 2; ++i) {
                        float& z = i ? nz : tiny;
                        z += dpp_take<0xB1, 0xF>(z);
                        z += dpp_take<0x4E, 0xF>(z);
                        z += dpp_take<0x141, 0xF>(z);
                        z += dpp_take<0x140, 0xF>(z);
                    }
                    if (r < N && c16 == 0) range_raise(range_flag, GHF_RANGE_ROWS, (int)tiny, (int)nz);
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
    const bool aligned = ((((uintptr_t)x | (uintptr_t)W_in) & 15) == 0);
    const bool mfma_ok = aligned && (F % 16) == 0 && (d % 16) == 0 && d <= 256;
    if (mfma_ok) {
        const int64_t rows_per_block = 4 * 16 * IP_MT;
        const unsigned grid = (unsigned)cdiv(N, rows_per_block);
        switch (d / 16) {
#define GHF_IP_CASE(NT)                                                                                                 \
    case NT:                                                                                                            \
        if (fuse) input_proj_mfma_kernel<NT, true><<<grid, 256, 0, stream>>>(x, W_in, b_in, N, F, h0, (char*)h_split, range_flag_ptr());   \
        else input_proj_mfma_kernel<NT, false><<<grid, 256, 0, stream>>>(x, W_in, b_in, N, F, h0, nullptr, nullptr);           \
        break;
            GHF_IP_CASE(1) GHF_IP_CASE(2) GHF_IP_CASE(3) GHF_IP_CASE(4) GHF_IP_CASE(5) GHF_IP_CASE(6)
            GHF_IP_CASE(7) GHF_IP_CASE(8) GHF_IP_CASE(9) GHF_IP_CASE(10) GHF_IP_CASE(11) GHF_IP_CASE(12)
            GHF_IP_CASE(13) GHF_IP_CASE(14) GHF_IP_CASE(15) GHF_IP_CASE(16)
#undef GHF_IP_CASE
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
