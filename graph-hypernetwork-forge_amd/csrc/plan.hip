// plan.hip — K0: graph plan (once per graph; cached by the host).
//
// Replaces the implicit edge order of the reference (models/hypergnn.py:191) and
// its fp32 in-degree count (:207-212) with: int32 in-degree, edges radix-sorted by
//   key = (dst / BN) * (R * BN) + rel * BN + (dst % BN)
// so that all edges of one (destination block, relation) group are contiguous —
// the order the message kernels consume — plus group (or CSR-row) offsets.
// dst and rel are recovered from the key, so a sorted edge costs 8 bytes
// (uint32 key + int32 src) instead of the reference's 24 (three int64).
// For block plans (BN > 1) two things the MFMA kernel would otherwise recompute in
// every wave of every chunk are done here once per graph:
//  - the chunk table: each group cut into chunks of <= chunk_rows edges, listed per block;
//  - each edge's "run head": the first row of its run of equal destinations inside its
//    16-row tile, packed into bits 28..31 of sorted_src (the kernel's segment-sum selector);
//  - work items: a destination block with more than `split_chunks` chunks (a hub of a power-law
//    graph) is cut into several items, each a contiguous range of its chunks, so that no
//    workgroup walks more than ~split_chunks chunks; a split block's items write partial sums
//    to scratch slots and are combined by a second kernel.
#include "common.h"

#include <hipcub/hipcub.hpp>
#include <stdlib.h>

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

// chunks per group: cnt[s] = ceil(group size / CR)
__global__ void plan_chunk_count_kernel(const int32_t* __restrict__ seg_off, int64_t nseg, int CR,
                                        int32_t* __restrict__ cnt) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nseg) cnt[s] = (seg_off[s + 1] - seg_off[s] + CR - 1) / CR;
}

// chunk_tab[c] = { first sorted edge, (rel << 8) | (cross << 7) | rows };  coff = exclusive scan of cnt.
// cross = 1 when a run of equal destinations continues across a 16-row tile boundary inside the chunk
// (then two tiles of the chunk add into the same row of the block sums and must do so in order).
__global__ void plan_chunk_fill_kernel(const uint32_t* __restrict__ sorted_key, const int32_t* __restrict__ seg_off,
                                       const int32_t* __restrict__ coff, int64_t nseg, int R, int CR,
                                       int32_t* __restrict__ chunk_tab) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const int e0 = seg_off[s], n = seg_off[s + 1] - e0, r = (int)(s % R);
    int c = coff[s];
    for (int j = 0; j < n; j += CR, ++c) {
        const int rows = (n - j) < CR ? (n - j) : CR;
        int cross = 0;
        for (int t = 16; t < rows; t += 16) cross |= sorted_key[e0 + j + t - 1] == sorted_key[e0 + j + t];
        chunk_tab[2 * c] = e0 + j;
        chunk_tab[2 * c + 1] = (r << 8) | (cross << 7) | rows;
    }
}

// blk_chunk_off[b] = coff[b * R] (b < NB), total chunk count at b == NB
__global__ void plan_block_chunks_kernel(const int32_t* __restrict__ coff, const int32_t* __restrict__ cnt,
                                         int64_t NB, int R, int32_t* __restrict__ blk_chunk_off) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < NB) blk_chunk_off[b] = coff[b * R];
    else if (b == NB) blk_chunk_off[b] = coff[NB * R - 1] + cnt[NB * R - 1];
}

// run head of every sorted edge inside its 16-row tile, into bits 28..31 of sorted_src
__global__ void plan_heads_kernel(const uint32_t* __restrict__ sorted_key, const int32_t* __restrict__ seg_off,
                                  int64_t nvalid, uint32_t BN, int32_t* __restrict__ sorted_src) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nvalid) return;
    const uint32_t key = sorted_key[e];
    if (key == KEY_INVALID) return;                          // dropped edge (sorted last)
    const int t = (int)((e - seg_off[key / BN]) & 15);       // row of e inside its tile
    int head = t;
    while (head > 0 && sorted_key[e - (t - head) - 1] == key) --head;
    sorted_src[e] = (int32_t)(((uint32_t)sorted_src[e] & (uint32_t)SRC_MASK) | ((uint32_t)head << SRC_BITS));
}

// items per block: 1, or ceil(chunks / T) for a heavy block; slots = items of split blocks only.
// The blocks of the last, partly filled round of workgroups (tail0 .. NB-1; one workgroup per CU per round) are cut into
// tail_items items each, so that the round's work spreads over the CUs that would otherwise idle through it.
__global__ void plan_item_count_kernel(const int32_t* __restrict__ blk_chunk_off, int64_t NB, int T, int64_t tail0,
                                       int tail_items, int32_t* __restrict__ ni, int32_t* __restrict__ nslot) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    const int n = blk_chunk_off[b + 1] - blk_chunk_off[b];
    int k = n > T ? (n + T - 1) / T : 1;
    if (b >= tail0 && tail_items > k) k = n / 8 < tail_items ? (n / 8 > k ? n / 8 : k) : tail_items;   // >= 8 chunks per item
    ni[b] = k;
    nslot[b] = k > 1 ? k : 0;
}

// item_tab[i] = { block, first chunk, one past last chunk, scratch slot or -1 }; blk_item_off = exclusive scan of ni
__global__ void plan_item_fill_kernel(const int32_t* __restrict__ blk_chunk_off, const int32_t* __restrict__ ni,
                                      const int32_t* __restrict__ ioff, const int32_t* __restrict__ soff, int64_t NB,
                                      int32_t* __restrict__ item_tab, int32_t* __restrict__ blk_item_off,
                                      int32_t* __restrict__ counts) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > NB) return;
    if (b == NB) {                                       // totals: items, scratch slots
        const int items = ioff[NB - 1] + ni[NB - 1];
        blk_item_off[NB] = items;
        counts[0] = items;
        counts[1] = soff[NB - 1] + (ni[NB - 1] > 1 ? ni[NB - 1] : 0);
        return;
    }
    const int c0 = blk_chunk_off[b], n = blk_chunk_off[b + 1] - c0, k = ni[b];
    const int per = (n + k - 1) / k;
    blk_item_off[b] = ioff[b];
    for (int j = 0; j < k; ++j) {
        int32_t* it = item_tab + 4 * (size_t)(ioff[b] + j);
        const int a = c0 + j * per, z = c0 + ((j + 1) * per < n ? (j + 1) * per : n);
        it[0] = (int32_t)b; it[1] = a < z ? a : z; it[2] = z; it[3] = k > 1 ? soff[b] + j : -1;
    }
}

static size_t sort_temp_bytes(int64_t E) {
    size_t tb = 0;
    hipcub::DeviceRadixSort::SortPairs<uint32_t, int32_t>(nullptr, tb, nullptr, nullptr, nullptr, nullptr,
                                                          (int)E, 0, 32, (hipStream_t)0);
    return tb;
}

static size_t scan_temp_bytes(int64_t n) {
    size_t tb = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, (hipStream_t)0);
    return tb;
}

static int64_t num_segments(int64_t N, int R, int BN) { return BN == 1 ? N : cdiv(N, BN) * R; }

// The last round of workgroups: with NB blocks on `cus` compute units, NB % cus blocks remain for a last round that keeps
// only as many CUs busy.  When that is less than half of them (and there is more than one round), each of these blocks
// becomes floor(cus / remainder) <= 8 work items.  GHF_TAIL_SPLIT=0 turns it off.
static int tail_split(int64_t NB, int64_t* tail0) {
    static const int cus = [] {
        const char* e = getenv("GHF_TAIL_SPLIT");
        if (e && atoi(e) == 0) return 0;
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return n;
    }();
    *tail0 = NB;
    if (cus <= 0 || NB <= cus) return 1;
    const int64_t rem = NB % cus;
    if (rem == 0 || rem * 2 > cus) return 1;
    *tail0 = NB - rem;
    const int64_t s = cus / rem;
    return (int)(s > 8 ? 8 : s);
}

int64_t plan_max_items(int64_t N, int64_t E, int R, int BN, int CR, int T) {
    if (BN == 1 || CR <= 0 || T <= 0) return 0;
    // every block one item, heavy ones chunks/T more, the last round's blocks up to 8 each (at most one per CU)
    return cdiv(N, BN) + plan_max_chunks(N, E, R, BN, CR) / T + 1 + 1024;
}

int64_t plan_max_chunks(int64_t N, int64_t E, int R, int BN, int CR) {
    if (BN == 1 || CR <= 0) return 0;
    const int64_t nseg = num_segments(N, R, BN);
    return (nseg < E ? nseg : E) + E / CR + 1;          // every non-empty group adds at most one partial chunk
}

size_t plan_workspace_bytes(int64_t N, int64_t E, int R, int BN, int CR) {
    if (E <= 0) return 256;
    size_t b = align_up((size_t)E * 4, 256) * 2 + align_up(sort_temp_bytes(E), 256) + 256;
    if (BN > 1 && CR > 0) {
        const int64_t nseg = num_segments(N, R, BN);
        b += align_up((size_t)nseg * 4, 256) * 2 + align_up(scan_temp_bytes(nseg), 256);
        b += align_up((size_t)(cdiv(N, BN) + 1) * 4, 256) * 4;          // item counts and their scans
    }
    return b;
}

int launch_plan_build(const int64_t* edge_index, const int64_t* rel_id, int64_t N, int64_t E, int R,
                      int BN, int CR, int T, void* ws, size_t ws_bytes, uint32_t* sorted_key, int32_t* sorted_src,
                      int32_t* seg_off, int32_t* indeg, int32_t* chunk_tab, int32_t* blk_chunk_off,
                      int32_t* item_tab, int32_t* blk_item_off, int32_t* status, hipStream_t stream) {
    GHF_REQUIRE(N > 0 && E > 0 && R > 0 && BN > 0, "plan: N, E, R, block_nodes must be positive");
    GHF_REQUIRE(E < (1ll << 31), "plan: E=%lld needs < 2^31 edges", (long long)E);
    const int64_t NB = cdiv(N, BN);
    const uint64_t key_space = (uint64_t)NB * (uint64_t)BN * (uint64_t)R;
    GHF_REQUIRE(key_space < 0xFFFFFFFFull, "plan: ceil(N/BN)*BN*R = %llu does not fit 32-bit keys",
                (unsigned long long)key_space);
    if (BN > 1) {
        GHF_REQUIRE(N <= (1ll << SRC_BITS), "plan: block plans pack the run head above bit %d of the source id", SRC_BITS);
        // (kernels that use the run heads / the `cross` bit need chunks that start on a 16-row tile of their group: CR % 16 == 0)
        GHF_REQUIRE(CR > 0 && CR < 128 && (CR % 4) == 0 && R < (1 << 23), "plan: chunk_rows must be a multiple of 4 below 128, R < 2^23");
        GHF_REQUIRE(T > 0 && item_tab && blk_item_off, "plan: block plans need split_chunks > 0, item_tab and blk_item_off");
    }
    GHF_REQUIRE(ws_bytes >= plan_workspace_bytes(N, E, R, BN, CR), "plan: workspace too small");
    GHF_REQUIRE(((uintptr_t)ws & 255) == 0, "plan: workspace must be 256-byte aligned");

    char* p = (char*)ws;
    uint32_t* keys_in = (uint32_t*)p;            p += align_up((size_t)E * 4, 256);
    int32_t* vals_in = (int32_t*)p;              p += align_up((size_t)E * 4, 256);
    void* sort_tmp = p;                          p += align_up(sort_temp_bytes(E), 256);
    size_t sort_bytes = sort_temp_bytes(E);

    GHF_HIP_CHECK(hipMemsetAsync(indeg, 0, (size_t)N * 4, stream));
    GHF_HIP_CHECK(hipMemsetAsync(status, 0, 12, stream));
    const int tpb = 256;
    const int grid = (int)((E + tpb - 1) / tpb < 8192 ? (E + tpb - 1) / tpb : 8192);
    plan_keys_kernel<<<grid, tpb, 0, stream>>>(edge_index, rel_id, N, E, R, BN, keys_in, vals_in, indeg, status);
    GHF_LAUNCH_CHECK();

    // all 32 key bits are sorted so that KEY_INVALID (dropped edges) lands last
    GHF_HIP_CHECK((hipcub::DeviceRadixSort::SortPairs<uint32_t, int32_t>(
        sort_tmp, sort_bytes, keys_in, sorted_key, vals_in, sorted_src, (int)E, 0, 32, stream)));

    const int64_t nseg = num_segments(N, R, BN);
    const uint32_t seg_div = (BN == 1) ? (uint32_t)R : (uint32_t)BN;
    plan_offsets_kernel<<<(int)((nseg + 1 + 255) / 256), 256, 0, stream>>>(sorted_key, E, nseg, seg_div, seg_off);
    GHF_LAUNCH_CHECK();

    if (BN > 1) {
        int32_t* cnt = (int32_t*)p;              p += align_up((size_t)nseg * 4, 256);
        int32_t* coff = (int32_t*)p;             p += align_up((size_t)nseg * 4, 256);
        void* scan_tmp = p;
        size_t scan_bytes = scan_temp_bytes(nseg);
        const int gs = (int)((nseg + 255) / 256);
        plan_chunk_count_kernel<<<gs, 256, 0, stream>>>(seg_off, nseg, CR, cnt);
        GHF_LAUNCH_CHECK();
        GHF_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, (const int32_t*)cnt, coff, (int)nseg, stream));
        plan_chunk_fill_kernel<<<gs, 256, 0, stream>>>(sorted_key, seg_off, coff, nseg, R, CR, chunk_tab);
        GHF_LAUNCH_CHECK();
        plan_block_chunks_kernel<<<(int)((NB + 1 + 255) / 256), 256, 0, stream>>>(coff, cnt, NB, R, blk_chunk_off);
        GHF_LAUNCH_CHECK();
        // heads for the valid edges only: seg_off[nseg] (device) bounds them, so the kernel re-reads it per edge
        plan_heads_kernel<<<(int)((E + 255) / 256), 256, 0, stream>>>(sorted_key, seg_off, E, (uint32_t)BN, sorted_src);
        GHF_LAUNCH_CHECK();
        // work items (status[1], status[2] = number of items, of scratch slots)
        p = (char*)scan_tmp + align_up(scan_bytes, 256);
        int32_t* ni = (int32_t*)p;               p += align_up((size_t)(NB + 1) * 4, 256);
        int32_t* nslot = (int32_t*)p;            p += align_up((size_t)(NB + 1) * 4, 256);
        int32_t* ioff = (int32_t*)p;             p += align_up((size_t)(NB + 1) * 4, 256);
        int32_t* soff = (int32_t*)p;
        const int gb = (int)((NB + 1 + 255) / 256);
        int64_t tail0 = NB;
        const int tail_items = tail_split(NB, &tail0);
        plan_item_count_kernel<<<gb, 256, 0, stream>>>(blk_chunk_off, NB, T, tail0, tail_items, ni, nslot);
        GHF_LAUNCH_CHECK();
        GHF_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, (const int32_t*)ni, ioff, (int)NB, stream));
        GHF_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, (const int32_t*)nslot, soff, (int)NB, stream));
        plan_item_fill_kernel<<<gb, 256, 0, stream>>>(blk_chunk_off, ni, ioff, soff, NB, item_tab, blk_item_off, status + 1);
        GHF_LAUNCH_CHECK();
    }
    return GHF_OK;
}

// ---- ghf_group_edges: edges grouped by relation, for the backward's per-relation contractions -------------------------
// perm[0..E) = edge ids sorted (stably) by relation, goff[r] = first position of relation r, goff[R] = E.
__global__ void group_keys_kernel(const int64_t* __restrict__ rel, int64_t E, int R, uint32_t* __restrict__ key, int32_t* __restrict__ val) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t r = rel[e];
    key[e] = (r >= 0 && r < R) ? (uint32_t)r : (uint32_t)(R - 1);       // (range was checked by the plan build)
    val[e] = (int32_t)e;
}
__global__ void group_finish_kernel(const uint32_t* __restrict__ skey, const int32_t* __restrict__ sval, int64_t E, int R,
                                    int64_t* __restrict__ perm, int64_t* __restrict__ goff) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) perm[i] = sval[i];
    if (i <= R) {
        int64_t lo = 0, hi = E;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (skey[mid] < (uint32_t)i) lo = mid + 1; else hi = mid; }
        goff[i] = lo;
    }
}

size_t group_workspace_bytes(int64_t E) {
    return align_up((size_t)E * 4, 256) * 4 + align_up(sort_temp_bytes(E), 256);
}

int launch_group_edges(const int64_t* rel, int64_t E, int R, void* ws, size_t ws_bytes, int64_t* perm, int64_t* goff,
                       hipStream_t stream) {
    GHF_REQUIRE(E > 0 && E < (1ll << 31) && R > 0, "group_edges: bad sizes");
    GHF_REQUIRE(ws_bytes >= group_workspace_bytes(E), "group_edges: workspace too small");
    char* p = (char*)ws;
    uint32_t* key = (uint32_t*)p;  p += align_up((size_t)E * 4, 256);
    int32_t* val = (int32_t*)p;    p += align_up((size_t)E * 4, 256);
    uint32_t* skey = (uint32_t*)p; p += align_up((size_t)E * 4, 256);
    int32_t* sval = (int32_t*)p;   p += align_up((size_t)E * 4, 256);
    size_t tb = sort_temp_bytes(E);
    const int gb = (int)((E + 255) / 256);
    group_keys_kernel<<<gb, 256, 0, stream>>>(rel, E, R, key, val);
    GHF_LAUNCH_CHECK();
    int bits = 1;
    while ((1ll << bits) < R) ++bits;
    GHF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(p, tb, key, skey, val, sval, (int)E, 0, bits, stream));   // stable
    const int64_t n = E > R + 1 ? E : R + 1;
    group_finish_kernel<<<(int)((n + 255) / 256), 256, 0, stream>>>(skey, sval, E, R, perm, goff);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

}  // namespace ghf
