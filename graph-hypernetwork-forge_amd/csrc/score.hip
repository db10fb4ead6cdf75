// score.hip — link-prediction scores: row-wise dot products, with the row gathers of the call sites fused in.
//
// Replaces models/hypergnn.py:304-318 of the reference (score_triple: (head * tail).sum(-1)) together with the
// advanced-indexing copies its callers make first (demo.py:90-94: score_triple(embs[src], embs[dst]) materialises two
// [E, d] matrices).  HBM-bound: two rows per pair.  One wave per pair at a time, lanes stride the row in 16-byte
// pieces, DPP wave reduction; an index outside [0, rows) yields NaN instead of a fault.
#include "common.h"

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void score_pairs_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
                                                          int64_t rows_a, int64_t rows_b, int64_t n, int d,
                                                          float* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const bool vec = (d & 3) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t ra = ia ? ia[i] : i, rb = ib ? ib[i] : i;
        if (ra < 0 || ra >= rows_a || rb < 0 || rb >= rows_b) {
            if (lane == 0) scores[i] = __int_as_float(0x7FC00000);
            continue;
        }
        const float* __restrict__ pa = a + (size_t)ra * d;
        const float* __restrict__ pb = b + (size_t)rb * d;
        float s = 0.f;
        if (vec) {
            for (int k = lane * 4; k < d; k += 256) {
                const f32x4 x = *(const f32x4*)(pa + k), y = *(const f32x4*)(pb + k);
                s = fmaf(x[0], y[0], fmaf(x[1], y[1], fmaf(x[2], y[2], fmaf(x[3], y[3], s))));
            }
        } else {
            for (int k = lane; k < d; k += 64) s = fmaf(pa[k], pb[k], s);
        }
        s = wave_sum(s);
        if (lane == 0) scores[i] = s;
    }
}

int launch_score_pairs(const float* a, const float* b, const int64_t* ia, const int64_t* ib, int64_t rows_a, int64_t rows_b,
                       int64_t n, int d, float* scores, hipStream_t stream) {
    GHF_REQUIRE(d > 0 && rows_a > 0 && rows_b > 0 && n >= 0, "score_pairs: bad shape");
    GHF_REQUIRE(ia || n <= rows_a, "score_pairs: n exceeds the rows of a");
    GHF_REQUIRE(ib || n <= rows_b, "score_pairs: n exceeds the rows of b");
    if (n == 0) return GHF_OK;
    const int64_t want = cdiv(n, 4);
    const unsigned grid = (unsigned)(want < 256 * 64 ? want : 256 * 64);       // grid-stride beyond 64 workgroups per CU
    score_pairs_kernel<<<grid, 256, 0, stream>>>(a, b, ia, ib, rows_a, rows_b, n, d, scores);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
