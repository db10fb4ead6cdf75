// plan.hip — K0: graph plan (once per graph; cached by the host).
//
// Replaces the implicit edge order of the reference (models/hypergnn.py:191) and
// its fp32 in-degree count (:207-212) with: int32 in-degree, edges radix-sorted by
//   key = (dst / BN) * (R * BN) + rel * BN + (dst % BN)
// so that all edges of one (destination block, relation) group are contiguous —
// the order the message kernels consume — plus group (or CSR-row) offsets.
// dst and rel are recovered from the key, so a sorted edge costs 8 bytes
// (uint32 key + int32 src) instead of the reference's 24 (three int64).
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace ghf {

__global__ void plan_keys_kernel(const int64_t* __restrict__ ei, const int64_t* __restrict__ rel,
                                 int64_t N, int64_t E, int R, int BN,
                                 uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                 int32_t* __restrict__ indeg, int32_t* __restrict__ status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int bad = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
        const int64_t s = ei[e], t = ei[E + e], r = rel[e];
        const bool ok_n = (s >= 0) & (s < N) & (t >= 0) & (t < N);
        const bool ok_r = (r >= 0) & (r < R);
        uint32_t key = KEY_INVALID;
        int32_t val = 0;
        if (ok_n && ok_r) {
            const uint32_t blk = (uint32_t)(t / BN), loc = (uint32_t)(t % BN);
            key = blk * (uint32_t)(R * BN) + (uint32_t)r * (uint32_t)BN + loc;
            val = (int32_t)s;
            atomicAdd(&indeg[t], 1);
        } else {
            bad |= (ok_n ? 0 : 1) | (ok_r ? 0 : 2);
        }
        keys[e] = key;
        vals[e] = val;
    }
    if (bad) atomicOr(status, bad);
}

// seg_off[s] = first sorted position whose key >= s * seg_div (lower bound), s in [0, nseg].
__global__ void plan_offsets_kernel(const uint32_t* __restrict__ sorted_key, int64_t E,
                                    int64_t nseg, uint32_t seg_div, int32_t* __restrict__ seg_off) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nseg) return;
    const uint64_t target = (uint64_t)s * seg_div;     // keys < 2^32 - 1; invalid keys = 2^32 - 1 sort last
    int64_t lo = 0, hi = E;
    if (target >= (uint64_t)KEY_INVALID) {
        // everything valid is below: count of valid keys = lower_bound(KEY_INVALID)
        const uint32_t tk = KEY_INVALID;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_key[mid] < tk) lo = mid + 1; else hi = mid; }
    } else {
        const uint32_t tk = (uint32_t)target;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_key[mid] < tk) lo = mid + 1; else hi = mid; }
    }
    seg_off[s] = (int32_t)lo;
}

static int key_bits(uint64_t max_key_exclusive) {
    int b = 1;
    while (b < 32 && (1ull << b) < max_key_exclusive) ++b;
    return b;
}

static size_t cub_temp_bytes(int64_t E) {
    size_t tb = 0;
    hipcub::DeviceRadixSort::SortPairs<uint32_t, int32_t>(nullptr, tb, nullptr, nullptr, nullptr, nullptr,
                                                          (int)E, 0, 32, (hipStream_t)0);
    return tb;
}

size_t plan_workspace_bytes(int64_t N, int64_t E, int R, int block_nodes) {
    (void)N; (void)R; (void)block_nodes;
    if (E <= 0) return 256;
    return align_up((size_t)E * 4, 256) * 2 + align_up(cub_temp_bytes(E), 256) + 256;
}

int launch_plan_build(const int64_t* edge_index, const int64_t* rel_id, int64_t N, int64_t E, int R,
                      int BN, void* ws, size_t ws_bytes, uint32_t* sorted_key, int32_t* sorted_src,
                      int32_t* seg_off, int32_t* indeg, int32_t* status, hipStream_t stream) {
    GHF_REQUIRE(N > 0 && E > 0 && R > 0 && BN > 0, "plan: N, E, R, block_nodes must be positive");
    GHF_REQUIRE(E < (1ll << 31), "plan: E=%lld needs < 2^31 edges", (long long)E);
    const int64_t NB = cdiv(N, BN);
    const uint64_t key_space = (uint64_t)NB * (uint64_t)BN * (uint64_t)R;
    GHF_REQUIRE(key_space < 0xFFFFFFFFull, "plan: ceil(N/BN)*BN*R = %llu does not fit 32-bit keys",
                (unsigned long long)key_space);
    GHF_REQUIRE(ws_bytes >= plan_workspace_bytes(N, E, R, BN), "plan: workspace too small");
    GHF_REQUIRE(((uintptr_t)ws & 255) == 0, "plan: workspace must be 256-byte aligned");

    char* p = (char*)ws;
    uint32_t* keys_in = (uint32_t*)p;            p += align_up((size_t)E * 4, 256);
    int32_t* vals_in = (int32_t*)p;              p += align_up((size_t)E * 4, 256);
    void* cub_tmp = p;
    size_t cub_bytes = cub_temp_bytes(E);

    GHF_HIP_CHECK(hipMemsetAsync(indeg, 0, (size_t)N * 4, stream));
    GHF_HIP_CHECK(hipMemsetAsync(status, 0, 4, stream));
    const int tpb = 256;
    const int grid = (int)((E + tpb - 1) / tpb < 8192 ? (E + tpb - 1) / tpb : 8192);
    plan_keys_kernel<<<grid, tpb, 0, stream>>>(edge_index, rel_id, N, E, R, BN, keys_in, vals_in, indeg, status);
    GHF_LAUNCH_CHECK();

    GHF_HIP_CHECK((hipcub::DeviceRadixSort::SortPairs<uint32_t, int32_t>(
        cub_tmp, cub_bytes, keys_in, sorted_key, vals_in, sorted_src, (int)E, 0, 32, stream)));
    (void)key_bits;   // full 32 bits are sorted so that KEY_INVALID lands last

    const int64_t nseg = (BN == 1) ? N : NB * R;
    const uint32_t seg_div = (BN == 1) ? (uint32_t)R : (uint32_t)BN;
    const int64_t nthreads = nseg + 1;
    plan_offsets_kernel<<<(int)((nthreads + 255) / 256), 256, 0, stream>>>(sorted_key, E, nseg, seg_div, seg_off);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
