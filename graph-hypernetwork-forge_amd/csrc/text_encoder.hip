// text_encoder.hip — the reference's character-bag TextEncoder for all U unique relation strings at once.
//
// Replaces models/hypergnn.py:39-81 of the reference (per-string Python loop: ids = min(ord(c), 127), '' -> [0];
// emb = mean_c E[id_c]; out = tanh(emb W^T + b)).  The host tokenises once per graph plan (padded [U, Lmax] ids +
// lengths, cached on the device); this kernel is one workgroup per string: lanes own the embedding columns for the
// mean (coalesced rows of E), then the output units for the projection.  Microseconds; it exists so that the whole
// forward is C-ABI calls with no host round trip in between.
#include "common.h"

namespace ghf {

constexpr int TE_MAX_C = 1024;

__global__ __launch_bounds__(256) void text_encode_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                          int Lmax, const float* __restrict__ E, int V, int Cdim,
                                                          const float* __restrict__ W, const float* __restrict__ b, int T,
                                                          float* __restrict__ out) {
    __shared__ float pooled[TE_MAX_C];
    const int u = blockIdx.x;
    const int len = lens[u] > 0 ? (lens[u] < Lmax ? lens[u] : Lmax) : 1;
    const int32_t* __restrict__ my = ids + (size_t)u * Lmax;
    for (int c = threadIdx.x; c < Cdim; c += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < len; ++l) {
            int id = my[l];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            s += E[(size_t)id * Cdim + c];
        }
        pooled[c] = s / (float)len;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float* __restrict__ w = W + (size_t)t * Cdim;
        float s = b[t];
        for (int c = 0; c < Cdim; ++c) s = fmaf(pooled[c], w[c], s);
        out[(size_t)u * T + t] = tanhf(s);
    }
}

// ---- backward: U <= a few hundred strings, so three single-purpose kernels with plain loops (fixed summation order) ----
// ws: pooled [U][C], dpre [U][T], dpooled [U][C]
__global__ __launch_bounds__(256) void text_bwd_rows_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                            int Lmax, const float* __restrict__ E, int V, int Cdim,
                                                            const float* __restrict__ W, int T, const float* __restrict__ te,
                                                            const float* __restrict__ dte, float* __restrict__ pooled,
                                                            float* __restrict__ dpre, float* __restrict__ dpooled) {
    const int u = blockIdx.x;
    const int len = lens[u] > 0 ? (lens[u] < Lmax ? lens[u] : Lmax) : 1;
    const int32_t* __restrict__ my = ids + (size_t)u * Lmax;
    for (int c = threadIdx.x; c < Cdim; c += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < len; ++l) {
            int id = my[l];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            s += E[(size_t)id * Cdim + c];
        }
        pooled[(size_t)u * Cdim + c] = s / (float)len;
    }
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float y = te[(size_t)u * T + t];
        dpre[(size_t)u * T + t] = dte[(size_t)u * T + t] * (1.0f - y * y);          // tanh'
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cdim; c += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s = fmaf(dpre[(size_t)u * T + t], W[(size_t)t * Cdim + c], s);
        dpooled[(size_t)u * Cdim + c] = s;
    }
}
__global__ __launch_bounds__(256) void text_bwd_proj_kernel(int U, int Cdim, int T, const float* __restrict__ pooled,
                                                            const float* __restrict__ dpre, float* __restrict__ dW,
                                                            float* __restrict__ db) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < Cdim; c += blockDim.x) {
        float s = 0.f;
        for (int u = 0; u < U; ++u) s = fmaf(dpre[(size_t)u * T + t], pooled[(size_t)u * Cdim + c], s);
        dW[(size_t)t * Cdim + c] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int u = 0; u < U; ++u) s += dpre[(size_t)u * T + t];
        db[t] = s;
    }
}
__global__ __launch_bounds__(256) void text_bwd_emb_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                           int U, int Lmax, int V, int Cdim, const float* __restrict__ dpooled,
                                                           float* __restrict__ dE) {
    const int id = blockIdx.x;
    for (int c = threadIdx.x; c < Cdim; c += blockDim.x) {
        float s = 0.f;
        for (int u = 0; u < U; ++u) {
            const int len = lens[u] > 0 ? (lens[u] < Lmax ? lens[u] : Lmax) : 1;
            int cnt = 0;
            for (int l = 0; l < len; ++l) {
                int x = ids[(size_t)u * Lmax + l];
                x = x < 0 ? 0 : (x >= V ? V - 1 : x);
                cnt += (x == id);
            }
            if (cnt) s += dpooled[(size_t)u * Cdim + c] * ((float)cnt / (float)len);
        }
        dE[(size_t)id * Cdim + c] = s;
    }
}

int launch_text_encode_bwd(const int32_t* ids, const int32_t* lens, int U, int Lmax, const float* E, int V, int C, const float* W,
                           int T, const float* te, const float* dte, float* workspace, float* dE, float* dW, float* db,
                           hipStream_t stream) {
    GHF_REQUIRE(U > 0 && Lmax > 0 && V > 0 && C > 0 && T > 0, "text_encode_bwd: U, Lmax, V, C, T must be positive");
    float* pooled = workspace;
    float* dpre = pooled + (size_t)U * C;
    float* dpooled = dpre + (size_t)U * T;
    text_bwd_rows_kernel<<<U, 256, 0, stream>>>(ids, lens, Lmax, E, V, C, W, T, te, dte, pooled, dpre, dpooled);
    GHF_LAUNCH_CHECK();
    text_bwd_proj_kernel<<<T, 256, 0, stream>>>(U, C, T, pooled, dpre, dW, db);
    GHF_LAUNCH_CHECK();
    text_bwd_emb_kernel<<<V, 256, 0, stream>>>(ids, lens, U, Lmax, V, C, dpooled, dE);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

int launch_text_encode(const int32_t* ids, const int32_t* lens, int U, int Lmax, const float* E, int V, int C,
                       const float* W, const float* b, int T, float* out, hipStream_t stream) {
    GHF_REQUIRE(U > 0 && Lmax > 0 && V > 0 && C > 0 && T > 0, "text_encode: U, Lmax, V, C, T must be positive");
    GHF_REQUIRE(C <= TE_MAX_C, "text_encode: char_emb_dim=%d above %d", C, TE_MAX_C);
    text_encode_kernel<<<U, 256, 0, stream>>>(ids, lens, Lmax, E, V, C, W, b, T, out);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
