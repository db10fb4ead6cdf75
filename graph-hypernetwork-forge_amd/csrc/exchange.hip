// exchange.hip — the sparse row exchange's pack and unpack (multi-GPU forward; dist.py: exchange="sparse").
//
// A rank gathers the SOURCE rows of its in-edges, so per layer it needs from every peer only the rows its edges read; the
// lists are fixed at plan time.  Before the pairwise send the owner packs the listed rows of its slot into one contiguous
// message; the receiver scatters the message to the rows' places.  What travels is what the next layer's kernel gathers:
// for the two-piece kernels a row of the split form (4d bytes: hi and lo fp16 planes) AND its power-of-two scale (4 bytes),
// which ghf_split_rows keeps in two regions of one buffer — `rows` [N][row_bytes] and `extra` [N][extra_bytes].  One message
// row = row_bytes + extra_bytes bytes; one wave per row at a time, 16 bytes per lane.  HBM-bound copies: no arithmetic.
#include "common.h"

namespace ghf {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool UNPACK>
__global__ __launch_bounds__(256) void rows_pack_kernel(char* __restrict__ rows, int64_t row_bytes, char* __restrict__ extra,
                                                        int64_t extra_bytes, const int64_t* __restrict__ idx, int64_t n,
                                                        int64_t nrows, char* __restrict__ packed) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t stride = row_bytes + extra_bytes;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t r = idx[i];
        if (r < 0 || r >= nrows) continue;                  // (a list is built from the plan's own edges: cannot happen; no fault if it does)
        char* __restrict__ m = packed + i * stride;
        char* __restrict__ p = rows + r * row_bytes;
        for (int64_t k = (int64_t)lane * 16; k < row_bytes; k += 1024) {
            if (UNPACK) *(i32x4*)(p + k) = *(const i32x4*)(m + k);
            else *(i32x4*)(m + k) = *(const i32x4*)(p + k);
        }
        if (extra) {
            char* __restrict__ x = extra + r * extra_bytes;
            for (int64_t k = (int64_t)lane * 4; k < extra_bytes; k += 256) {
                if (UNPACK) *(int*)(x + k) = *(const int*)(m + row_bytes + k);
                else *(int*)(m + row_bytes + k) = *(const int*)(x + k);
            }
        }
    }
}

int launch_rows_pack(bool unpack, void* rows, int64_t row_bytes, void* extra, int64_t extra_bytes, const int64_t* idx, int64_t n,
                     int64_t nrows, void* packed, hipStream_t stream) {
    GHF_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0 && extra_bytes >= 0 && extra_bytes % 4 == 0,
                "rows_pack: row_bytes must be a multiple of 16 and extra_bytes of 4");
    GHF_REQUIRE((((uintptr_t)rows | (uintptr_t)packed) & 15) == 0 && (((uintptr_t)extra) & 3) == 0, "rows_pack: misaligned buffer");
    GHF_REQUIRE((extra != nullptr) == (extra_bytes > 0), "rows_pack: extra and extra_bytes go together");
    if (n == 0) return GHF_OK;
    const int64_t want = cdiv(n, 4);
    const unsigned grid = (unsigned)(want < 256 * 32 ? want : 256 * 32);
    if (unpack) rows_pack_kernel<true><<<grid, 256, 0, stream>>>((char*)rows, row_bytes, (char*)extra, extra_bytes, idx, n, nrows, (char*)packed);
    else rows_pack_kernel<false><<<grid, 256, 0, stream>>>((char*)rows, row_bytes, (char*)extra, extra_bytes, idx, n, nrows, (char*)packed);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
