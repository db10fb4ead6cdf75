// capi.hip — the extern "C" surface declared in include/ghf.h.
#include "common.h"
#include <vector>
#include <thread>
#include <algorithm>

#include <stdlib.h>
#include <string.h>

namespace ghf {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ghf

namespace ghf {
static std::atomic<int32_t*> g_range_flag{nullptr};
int32_t* range_flag_ptr() { return g_range_flag.load(std::memory_order_acquire); }
}  // namespace ghf

using namespace ghf;

extern "C" {

int ghf_set_range_flag(int32_t* device_word) {
    g_range_flag.store(device_word, std::memory_order_release);
    return GHF_OK;
}

int ghf_abi_version(void) { return GHF_ABI_VERSION; }

unsigned long long ghf_host_checksum64(const void* p, size_t nbytes, unsigned long long seed) {
    // four independent multiply-xor lanes over 32-byte strides (memory-bound from ~2 lanes on), folded with their positions
    const unsigned long long K0 = 0x9E3779B97F4A7C15ull, K1 = 0xC2B2AE3D27D4EB4Full, K2 = 0x165667B19E3779F9ull, K3 = 0x27D4EB2F165667C5ull;
    const unsigned long long* w = (const unsigned long long*)p;
    const size_t n = nbytes / 8;
    unsigned long long h0 = seed ^ K0, h1 = seed ^ K1, h2 = seed ^ K2, h3 = seed ^ K3;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        h0 = (h0 ^ w[i]) * K1;      h0 ^= h0 >> 29;
        h1 = (h1 ^ w[i + 1]) * K2;  h1 ^= h1 >> 31;
        h2 = (h2 ^ w[i + 2]) * K3;  h2 ^= h2 >> 27;
        h3 = (h3 ^ w[i + 3]) * K0;  h3 ^= h3 >> 33;
    }
    for (; i < n; ++i) { h0 = (h0 ^ w[i]) * K1; h0 ^= h0 >> 29; }
    unsigned long long h = (h0 * K0) ^ (h1 * K1 + 1) ^ (h2 * K2 + 2) ^ (h3 * K3 + 3) ^ (unsigned long long)n;
    h ^= h >> 32; h *= K2; h ^= h >> 29;
    return h;
}

const char* ghf_last_error(void) { return err_buf(); }

int ghf_message_config(int d, int* block_nodes, int* wlayout, int* chunk_rows, int* split_chunks) {
    if (!block_nodes || !wlayout || !chunk_rows || !split_chunks) return set_err(GHF_EINVAL, "message_config: null output pointer");
    int bn = 1, cr = 0, sc = 0;
    // GHF_KERNEL selects the d = 128 kernel for A/B runs: "bx" (default) = two fp16 pieces, three products, block sums in
    // registers (message_bx.hip; round 1's LDS-sum variant message_hx.hip was removed in round 3); "pp" = exact fp32,
    // v_mfma_f32_16x16x4_f32 (message_pp.hip: also what the range guard falls back to)
    // "rs" / "rs32" / "generic": a CSR plan also where a destination-block kernel exists (A/B of the relation-stationary
    // layer at d = 128)
    const char* kv = getenv("GHF_KERNEL");
    if (kv && (!strcmp(kv, "rs") || !strcmp(kv, "rs32") || !strcmp(kv, "generic"))) {
        *block_nodes = 1;
        *wlayout = GHF_WLAYOUT_NATURAL;
        *chunk_rows = 0;
        *split_chunks = 0;
    } else if ((!kv || !strcmp(kv, "bx")) && message_bx_config(d, &bn, &cr, &sc)) {
        *block_nodes = bn;
        *wlayout = GHF_WLAYOUT_SPLIT2H;
        *chunk_rows = cr;
        *split_chunks = sc;
    } else if (message_pp_config(d, &bn, &cr, &sc)) {
        *block_nodes = bn;
        *wlayout = GHF_WLAYOUT_FRAG16;
        *chunk_rows = cr;
        *split_chunks = sc;
    } else {
        *block_nodes = 1;
        *wlayout = GHF_WLAYOUT_NATURAL;
        *chunk_rows = 0;
        *split_chunks = 0;
    }
    return GHF_OK;
}

int64_t ghf_plan_max_items(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows, int split_chunks) {
    return plan_max_items(N, E, R, block_nodes, chunk_rows, split_chunks);
}

size_t ghf_plan_workspace_bytes(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows) {
    return plan_workspace_bytes(N, E, R, block_nodes, chunk_rows);
}

int64_t ghf_plan_max_chunks(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows) {
    return plan_max_chunks(N, E, R, block_nodes, chunk_rows);
}

int ghf_plan_build(const int64_t* edge_index, const int64_t* rel_id, int64_t N, int64_t E, int R, int block_nodes,
                   int chunk_rows, int split_chunks, void* workspace, size_t workspace_bytes, uint32_t* sorted_key,
                   int32_t* sorted_src, int32_t* seg_off, int32_t* indeg, int32_t* chunk_tab, int32_t* blk_chunk_off,
                   int32_t* item_tab, int32_t* blk_item_off, int32_t* status, void* stream) {
    GHF_REQUIRE(edge_index && rel_id && workspace && sorted_key && sorted_src && seg_off && indeg && status,
                "plan_build: null pointer argument");
    GHF_REQUIRE(block_nodes == 1 || (chunk_tab && blk_chunk_off && chunk_rows > 0),
                "plan_build: block plans need chunk_tab, blk_chunk_off and chunk_rows");
    return launch_plan_build(edge_index, rel_id, N, E, R, block_nodes, chunk_rows, split_chunks, workspace, workspace_bytes,
                             sorted_key, sorted_src, seg_off, indeg, chunk_tab, blk_chunk_off, item_tab, blk_item_off, status,
                             (hipStream_t)stream);
}

int ghf_weightgen_fwd(const float* text_emb, const float* const* head_params, const float* const* log_scales,
                      int R, int T, int Hh, int num_hidden, int d_in, int d_out, int layout, float* hidden_ws,
                      float* W_msg, float* W_self, float* bias, const float* hidden_drop, float* acts, void* stream) {
    GHF_REQUIRE(text_emb && head_params && log_scales && log_scales[0] && log_scales[1] && log_scales[2] && hidden_ws && W_msg && bias,
                "weightgen_fwd: null pointer argument");
    return launch_weightgen(text_emb, head_params, log_scales, R, T, Hh, num_hidden, d_in, d_out, layout,
                            hidden_ws, W_msg, W_self, bias, hidden_drop, (hipStream_t)stream, acts);
}

int ghf_weightgen_fwd_batched(int L, const float* text_emb, const float* const* head_params, const float* const* log_scales,
                              int R, int T, int Hh, int num_hidden, int d_in, int d_out, int layout, float* hidden_ws,
                              float* const* W_msg, float* const* W_self, float* const* bias, void* stream) {
    GHF_REQUIRE(text_emb && head_params && log_scales && hidden_ws && W_msg && bias, "weightgen_fwd_batched: null pointer argument");
    return launch_weightgen_batched(L, text_emb, head_params, log_scales, R, T, Hh, num_hidden, d_in, d_out, layout, hidden_ws,
                                    W_msg, W_self, bias, nullptr, (hipStream_t)stream);
}

int ghf_text_encode_fwd(const int32_t* ids, const int32_t* lens, int U, int Lmax, const float* char_emb, int V, int C,
                        const float* W, const float* b, int T, float* out, void* stream) {
    GHF_REQUIRE(ids && lens && char_emb && W && b && out, "text_encode_fwd: null pointer argument");
    return launch_text_encode(ids, lens, U, Lmax, char_emb, V, C, W, b, T, out, (hipStream_t)stream);
}

int ghf_input_proj_fwd(const float* x, const float* W_in, const float* b_in, int64_t N, int F, int d,
                       float* h0, void* h_split, int split_layout, void* stream) {
    GHF_REQUIRE(x && W_in && b_in && h0, "input_proj_fwd: null pointer argument");
    return launch_input_proj(x, W_in, b_in, N, F, d, h0, h_split, split_layout, (hipStream_t)stream);
}

size_t ghf_split_rows_bytes(int64_t N, int d, int wlayout) {
    if (wlayout == GHF_WLAYOUT_SPLIT2H) return (size_t)N * d * 4 + (size_t)N * 4;
    return 0;
}

size_t ghf_weights_bytes(int R, int d_in, int d_out, int wlayout) {
    const size_t n = (size_t)R * d_in * d_out;
    switch (wlayout) {
        case GHF_WLAYOUT_FRAG16:  return 2 * n * 4;
        case GHF_WLAYOUT_SPLIT2H: return 2 * n * 4 + (size_t)R * 4;
        default:                  return n * 4;
    }
}

int ghf_split_rows(const float* h, int64_t N, int d, int64_t row0, int64_t rows, int wlayout, void* h_split, void* stream) {
    GHF_REQUIRE(h && h_split, "split_rows: null pointer argument");
    GHF_REQUIRE(N > 0 && d > 0 && d % 4 == 0 && row0 >= 0 && rows >= 0 && row0 + rows <= N, "split_rows: bad shape or row range");
    if (wlayout == GHF_WLAYOUT_SPLIT2H) return launch_split2h_rows(h, N, d, row0, rows, h_split, (hipStream_t)stream);
    return set_err(GHF_EINVAL, "split_rows: layout %d gathers h itself", wlayout);
}

namespace {
// open-addressing table of machine words -> dense ids (0 is not a valid object address)
struct WordTable {
    std::vector<unsigned long long> key;
    std::vector<long long> val;
    size_t mask;
    explicit WordTable(size_t max_uniq) {
        size_t cap = 64;
        while (cap < max_uniq * 4) cap <<= 1;
        key.assign(cap, 0ull);
        val.assign(cap, -1);
        mask = cap - 1;
    }
    // id of x, or -1 (then *slot is where it belongs)
    long long find(unsigned long long x, size_t* slot) const {
        size_t h = (size_t)((x >> 4) * 0x9E3779B97F4A7C15ull >> 20) & mask;
        for (;;) {
            if (key[h] == x) return val[h];
            if (val[h] < 0) { *slot = h; return -1; }
            h = (h + 1) & mask;
        }
    }
    void put(size_t slot, unsigned long long x, long long id) { key[slot] = x; val[slot] = id; }
};
}  // namespace

long long ghf_host_word_ids(const void* words, long long n, long long* ids, void** uniq, long long max_uniq, int threads) {
    if (!words || !ids || !uniq || n < 0 || max_uniq <= 0) return -1;
    const unsigned long long* w = (const unsigned long long*)words;
    // One pass in order over a prefix finds (nearly always all of) the distinct words; the rest of the array is mapped by
    // `threads` threads against a copy of that table.  A word a thread meets that the prefix did not hold gets a provisional id
    // (>= PROV, per thread, in the thread's order); those are ranked afterwards — thread by thread, i.e. in array order — and the
    // few entries that carry one rewritten.  The result equals the sequential pass: ids by first appearance.
    const long long PROV = 1ll << 40;
    WordTable tab((size_t)max_uniq);
    long long k = 0;
    auto seq = [&](long long a, long long b) -> bool {
        unsigned long long last = 0;
        long long last_id = -1;
        for (long long i = a; i < b; ++i) {
            const unsigned long long x = w[i];
            if (x == last && last_id >= 0) { ids[i] = last_id; continue; }
            size_t slot;
            long long id = tab.find(x, &slot);
            if (id < 0) {
                if (k >= max_uniq) return false;
                id = k;
                tab.put(slot, x, id);
                uniq[k++] = (void*)x;
            }
            last = x;
            last_id = id;
            ids[i] = id;
        }
        return true;
    };
    const long long prefix = n < (1ll << 18) || threads <= 1 ? n : (1ll << 16);
    if (!seq(0, prefix)) return -1;
    if (prefix == n) return k;
    const int T = (int)std::min<long long>(threads, (n - prefix + (1ll << 18) - 1) >> 18);
    std::vector<std::vector<unsigned long long>> fresh(T);             // a thread's new words, in its order
    std::vector<int> bad(T, 0);
    const long long per = (n - prefix + T - 1) / T;
    auto work = [&](int t) {
        WordTable mine = tab;                                          // (a few KB .. MB: max_uniq * 4 slots)
        const long long a = prefix + t * per, b = std::min(n, a + per);
        unsigned long long last = 0;
        long long last_id = -1;
        for (long long i = a; i < b; ++i) {
            const unsigned long long x = w[i];
            if (x == last && last_id >= 0) { ids[i] = last_id; continue; }
            size_t slot;
            long long id = mine.find(x, &slot);
            if (id < 0) {
                if ((long long)fresh[t].size() + k >= max_uniq) { bad[t] = 1; return; }
                id = PROV + (long long)fresh[t].size();
                mine.put(slot, x, id);
                fresh[t].push_back(x);
            }
            last = x;
            last_id = id;
            ids[i] = id;
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
    }
    for (int t = 0; t < T; ++t)
        if (bad[t]) return -1;
    bool any = false;
    std::vector<std::vector<long long>> rank(T);
    for (int t = 0; t < T; ++t) {                                      // threads in array order: first appearance
        rank[t].resize(fresh[t].size());
        for (size_t j = 0; j < fresh[t].size(); ++j) {
            any = true;
            size_t slot;
            long long id = tab.find(fresh[t][j], &slot);
            if (id < 0) {
                if (k >= max_uniq) return -1;
                id = k;
                tab.put(slot, fresh[t][j], id);
                uniq[k++] = (void*)fresh[t][j];
            }
            rank[t][j] = id;
        }
    }
    if (any) {
        auto fix = [&](int t) {
            if (fresh[t].empty()) return;
            const long long a = prefix + t * per, b = std::min(n, a + per);
            for (long long i = a; i < b; ++i)
                if (ids[i] >= PROV) ids[i] = rank[t][(size_t)(ids[i] - PROV)];
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t) pool.emplace_back(fix, t);
        fix(0);
        for (auto& th : pool) th.join();
    }
    return k;
}

int ghf_message_side_output_supported(int d, int block_nodes, int wlayout) {
    return wlayout == GHF_WLAYOUT_SPLIT2H && message_bx_owns(d, block_nodes) ? 1 : 0;
}

int ghf_message_layer_fwd(const float* h, const void* h_split, int64_t N, int d, const uint32_t* sorted_key, const int32_t* sorted_src,
                          const int32_t* seg_off, const int32_t* indeg, const int32_t* chunk_tab,
                          const int32_t* blk_chunk_off, const int32_t* item_tab, const int32_t* blk_item_off,
                          int64_t item0, int64_t n_items, float* partial, int64_t E, int R, int block_nodes,
                          const float* W_msg, const float* W_self, const float* bias, int wlayout,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, int64_t row0, int64_t rows,
                          float* h_out, void* h_split_out, float* agg_out, int flags, void* stream) {
    GHF_REQUIRE(h && sorted_key && sorted_src && seg_off && indeg && W_msg && bias && h_out,
                "message_layer_fwd: null pointer argument");
    if (flags & GHF_FLAG_RAW_SUM) flags |= GHF_FLAG_NO_TAIL;
    GHF_REQUIRE((flags & GHF_FLAG_NO_TAIL) || (ln_gamma && ln_beta), "message_layer_fwd: LayerNorm parameters missing");
    GHF_REQUIRE(h != h_out, "message_layer_fwd: h_out must not alias h");
    GHF_REQUIRE(N > 0 && d > 0 && R > 0 && block_nodes > 0, "message_layer_fwd: N, d, R, block_nodes must be positive");
    GHF_REQUIRE(row0 >= 0 && rows >= 0 && row0 + rows <= N, "message_layer_fwd: row range [%lld,+%lld) outside [0,%lld)",
                (long long)row0, (long long)rows, (long long)N);
    GHF_REQUIRE(row0 % block_nodes == 0, "message_layer_fwd: row0 must be a multiple of block_nodes");
    GHF_REQUIRE(block_nodes == 1 || (chunk_tab && blk_chunk_off && item_tab && blk_item_off && item0 >= 0 && n_items >= 0),
                "message_layer_fwd: block plans need the chunk and item tables");
    const bool split = wlayout == GHF_WLAYOUT_SPLIT2H;
    GHF_REQUIRE(!split || h_split, "message_layer_fwd: SPLIT2H weights need h_split (ghf_split_rows)");
    GHF_REQUIRE(!h_split_out || (split && !(flags & GHF_FLAG_NO_TAIL) && h_split_out != h_split),
                "message_layer_fwd: h_split_out needs split weights and the fused tail, and must not alias h_split");
    MsgArgs a{h, h_split, N, d, sorted_key, sorted_src, seg_off, indeg, chunk_tab, blk_chunk_off, item_tab, blk_item_off, item0, n_items,
              partial, E, R, block_nodes, W_msg, W_self, bias, wlayout,
              ln_gamma, ln_beta, ln_eps, row0, rows, h_out, h_split_out, flags, agg_out};
    GHF_REQUIRE((flags & (GHF_FLAG_ZERO_SRC | GHF_FLAG_ZERO_DST)) != (GHF_FLAG_ZERO_SRC | GHF_FLAG_ZERO_DST),
                "message_layer_fwd: GHF_FLAG_ZERO_SRC and GHF_FLAG_ZERO_DST together leave nothing to compute");
    GHF_REQUIRE(!agg_out || (!(flags & GHF_FLAG_NO_TAIL) && agg_out != h_out), "message_layer_fwd: agg_out goes with the fused tail and must not alias h_out");
    GHF_REQUIRE(!(flags & GHF_FLAG_ADD_H) || (flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM)), "message_layer_fwd: GHF_FLAG_ADD_H goes with NO_TAIL / RAW_SUM");
    if ((flags & GHF_FLAG_ADD_H) && !ghf_message_side_output_supported(d, block_nodes, wlayout))
        return set_err(GHF_EUNSUPPORTED, "message_layer_fwd: GHF_FLAG_ADD_H is not implemented by the kernel for d=%d, block_nodes=%d, layout %d", d, block_nodes, wlayout);
    if ((flags & (GHF_FLAG_ZERO_SRC | GHF_FLAG_ZERO_DST)) && !ghf_message_side_output_supported(d, block_nodes, wlayout))
        return set_err(GHF_EUNSUPPORTED, "message_layer_fwd: GHF_FLAG_ZERO_SRC / ZERO_DST (a half of the weights that must not be read) is "
                       "not implemented by the kernel for d=%d, block_nodes=%d, layout %d: pack that half as zeros instead", d, block_nodes, wlayout);
    if (agg_out && !ghf_message_side_output_supported(d, block_nodes, wlayout))
        return set_err(GHF_EUNSUPPORTED, "message_layer_fwd: no side output from the kernel for d=%d, block_nodes=%d, layout %d", d, block_nodes, wlayout);
    if (block_nodes == 1) return launch_message_generic(a, (hipStream_t)stream);
    if (wlayout == GHF_WLAYOUT_SPLIT2H) {
        GHF_REQUIRE(message_bx_owns(d, block_nodes), "message_layer_fwd: no SPLIT2H kernel for d=%d with blocks of %d nodes", d, block_nodes);
        return launch_message_bx(a, (hipStream_t)stream);
    }
    GHF_REQUIRE(wlayout == GHF_WLAYOUT_FRAG16, "message_layer_fwd: unknown weight layout %d", wlayout);
    return launch_message_pp(a, (hipStream_t)stream);
}

size_t ghf_group_workspace_bytes(int64_t E) { return group_workspace_bytes(E); }

int ghf_group_edges(const int64_t* rel_id, int64_t E, int R, void* workspace, size_t workspace_bytes, int64_t* perm,
                    int64_t* goff, void* stream) {
    GHF_REQUIRE(rel_id && workspace && perm && goff, "group_edges: null pointer argument");
    return launch_group_edges(rel_id, E, R, workspace, workspace_bytes, perm, goff, (hipStream_t)stream);
}

size_t ghf_tail_bwd_workspace_floats(int64_t N, int d) { return (N >= 0 && d > 0) ? tail_bwd_workspace_floats(N, d) : 0; }

int ghf_tail_bwd(const float* grad_out, const float* agg, const float* h, const float* ln_gamma, float ln_eps,
                 const int32_t* indeg, int64_t N, int d, float* dpre, float* G, void* G_split, float* dgamma_dbeta,
                 float* workspace, const float* drop, void* stream) {
    GHF_REQUIRE(grad_out && agg && h && ln_gamma && indeg && dpre && G && dgamma_dbeta && workspace, "tail_bwd: null pointer argument");
    return launch_tail_bwd(grad_out, agg, h, ln_gamma, ln_eps, indeg, N, d, dpre, G, G_split, dgamma_dbeta, workspace, drop,
                           (hipStream_t)stream);
}

size_t ghf_colsum_workspace_floats(int64_t N, int d) { return colsum_workspace_floats(N, d); }

int ghf_colsum(const float* X, const float* mask, int64_t N, int d, float* workspace, float* out, int accumulate, void* stream) {
    GHF_REQUIRE(X && workspace && out, "colsum: null pointer argument");
    return launch_colsum(X, mask, N, d, workspace, out, accumulate, (hipStream_t)stream);
}

int ghf_relu_mask(const float* X, const float* ref, int64_t n, float* out, void* stream) {
    GHF_REQUIRE(X && ref && out, "relu_mask: null pointer argument");
    return launch_relu_mask(X, ref, n, out, (hipStream_t)stream);
}

int ghf_group_outer(const float* A, const int64_t* ia, int da, const float* B, const int64_t* ib, int db,
                    const int64_t* gstart, const int64_t* gend, int ngroups, float* C, int accumulate, void* stream) {
    GHF_REQUIRE((A || da == 0) && B && gstart && gend && C, "group_outer: null pointer argument");
    return launch_group_outer(A, ia, da, B, ib, db, gstart, gend, ngroups, C, accumulate, (hipStream_t)stream);
}

int ghf_message_rs_supported(int d) { return message_rs_supported(d); }

int ghf_edge_transform_fwd(const float* h, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                           const int64_t* slice_tab, int64_t nslices, const float* WmT, const float* WsT, const float* bias,
                           float* Y, void* stream) {
    GHF_REQUIRE(h && src && dst && ypos && slice_tab && WmT && WsT && bias && Y, "edge_transform_fwd: null pointer argument");
    return launch_edge_transform(h, N, d, src, dst, ypos, slice_tab, nslices, WmT, WsT, bias, Y, (hipStream_t)stream);
}

size_t ghf_weights_rs_bytes(int R, int d) { return weights_rs_bytes(R, d); }

int ghf_weights_pack_rs(const float* W_msg, const float* W_self, int R, int d, void* w2h, int* shift_ws, void* stream) {
    GHF_REQUIRE(W_msg && W_self && w2h && shift_ws, "weights_pack_rs: null pointer argument");
    return launch_weights_pack_rs(W_msg, W_self, R, d, w2h, shift_ws, (hipStream_t)stream);
}

int ghf_edge_transform_h_fwd(const void* h_split, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                             const int64_t* slice_tab, int64_t nslices, const void* w2h, int R, const float* bias,
                             const void* x_split, int64_t NX, const float* row_cnt, float* Y, void* stream) {
    GHF_REQUIRE(h_split && src && dst && ypos && slice_tab && w2h && bias && Y, "edge_transform_h_fwd: null pointer argument");
    GHF_REQUIRE((x_split != nullptr) == (NX > 0), "edge_transform_h_fwd: x_split and its row count go together");
    return launch_edge_transform_h(h_split, N, d, src, dst, ypos, slice_tab, nslices, w2h, R, bias, x_split, NX, row_cnt, Y,
                                   (hipStream_t)stream);
}

int ghf_run_rows_fwd(const float* h, int64_t N, int d, const int64_t* run_src, const int64_t* run_start, int64_t nruns,
                     void* x_split, void* stream) {
    GHF_REQUIRE((h && run_src && run_start && x_split) || nruns == 0, "run_rows_fwd: null pointer argument");
    return launch_run_rows(h, N, d, run_src, run_start, nruns, x_split, (hipStream_t)stream);
}

int ghf_segment_partial_fwd(const float* Y, const int64_t* hub_chunks, int64_t nchunks, int d, float* P, void* stream) {
    GHF_REQUIRE((Y && hub_chunks && P) || nchunks == 0, "segment_partial_fwd: null pointer argument");
    return launch_segment_partial(Y, hub_chunks, nchunks, d, P, (hipStream_t)stream);
}

int ghf_segment_tail_fwd(const float* Y, const int64_t* off, const int32_t* deg_of, const int32_t* hub_of, const int64_t* hub_tab, const float* P,
                         const float* h, const float* ln_gamma, const float* ln_beta, float ln_eps, int64_t row0,
                         int64_t rows, int d, float* h_out, void* h_split_out, int64_t n_split, int flags, void* stream) {
    GHF_REQUIRE(Y && off && h_out, "segment_tail_fwd: null pointer argument");
    GHF_REQUIRE(!hub_of || (hub_tab && P), "segment_tail_fwd: hub_of without hub_tab / P");
    GHF_REQUIRE((flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM)) || (h && ln_gamma && ln_beta), "segment_tail_fwd: tail inputs missing");
    GHF_REQUIRE(!h_split_out || n_split >= row0 + rows, "segment_tail_fwd: h_split_out has fewer rows than the range written");
    return launch_segment_tail(Y, off, deg_of, hub_of, hub_tab, P, h, ln_gamma, ln_beta, ln_eps, row0, rows, d, h_out, h_split_out,
                               n_split, flags, (hipStream_t)stream);
}

int ghf_edge_outer_supported(int d) { return edge_outer_supported(d); }

int ghf_edge_outer(const float* h, const float* G, const int64_t* src, const int64_t* dst, const int64_t* slice_tab,
                   const int64_t* slice_off, const int32_t* order, int64_t nslices, int R, int d, int64_t N, float* workspace,
                   float* dW, float* db, void* stream) {
    GHF_REQUIRE(h && G && src && dst && slice_tab && slice_off && workspace && dW && db, "edge_outer: null pointer argument");
    return launch_edge_outer(h, G, src, dst, slice_tab, slice_off, nslices, R, d, N, workspace, dW, db, (hipStream_t)stream, nullptr,
                             nullptr, order);
}

int ghf_edge_outer_scaled(const float* h, const float* G, const float* h_rowscale, const float* G_rowscale, const int64_t* src,
                          const int64_t* dst, const int64_t* slice_tab, const int64_t* slice_off, const int32_t* order,
                          int64_t nslices, int R, int d, int64_t N, float* workspace, float* dW, float* db, void* stream) {
    GHF_REQUIRE(h && G && h_rowscale && G_rowscale && src && dst && slice_tab && slice_off && workspace && dW && db,
                "edge_outer_scaled: null pointer argument");
    GHF_REQUIRE(N > 0, "edge_outer_scaled: N = %lld (the exact chain has no use for row scales: ghf_edge_outer with N <= 0)", (long long)N);
    return launch_edge_outer(h, G, src, dst, slice_tab, slice_off, nslices, R, d, N, workspace, dW, db, (hipStream_t)stream,
                             h_rowscale, G_rowscale, order);
}

int ghf_scale_exp(const float* X, int64_t n, const float* log_scale, float* out, void* stream) {
    GHF_REQUIRE((X && log_scale && out) || n == 0, "scale_exp: null pointer argument");
    return n > 0 ? launch_scale_exp(X, n, log_scale, out, (hipStream_t)stream) : GHF_OK;
}

int ghf_add3(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream) {
    GHF_REQUIRE((a && b && out) || n == 0, "add3: null pointer argument");
    return n > 0 ? launch_add3(a, b, c, n, out, (hipStream_t)stream) : GHF_OK;
}

int ghf_rowscale(const float* X, const float* g, int64_t n, int d, float* out, void* stream) {
    GHF_REQUIRE((X && g && out) || n == 0, "rowscale: null pointer argument");
    return n > 0 ? launch_rowscale(X, g, n, d, out, (hipStream_t)stream) : GHF_OK;
}

int ghf_segment_axpy(const float* w, const int64_t* iw, const float* X, const int64_t* ix, const int64_t* off, int64_t nseg,
                     int64_t nx, int d, float* out, void* stream) {
    GHF_REQUIRE((w && iw && X && ix && off && out) || nseg == 0, "segment_axpy: null pointer argument");
    return launch_segment_axpy(w, iw, X, ix, off, nseg, nx, d, out, (hipStream_t)stream);
}

int ghf_dot(const float* X, const float* Y, int64_t n, float* workspace, float* out, void* stream) {
    GHF_REQUIRE(X && Y && workspace && out, "dot: null pointer argument");
    return launch_dot(X, Y, n, workspace, out, (hipStream_t)stream);
}

int ghf_weightgen_acts(const float* text_emb, const float* const* head_params, int R, int T, int Hh, int num_hidden,
                       float* acts, const float* hidden_drop, void* stream) {
    GHF_REQUIRE(text_emb && head_params && (acts || num_hidden == 0), "weightgen_acts: null pointer argument");
    return launch_weightgen_acts(text_emb, head_params, R, T, Hh, num_hidden, acts, hidden_drop, (hipStream_t)stream);
}

int ghf_weightgen_bwd_supported(int T, int Hh, int num_hidden) { return weightgen_bwd_supported(T, Hh, num_hidden); }

size_t ghf_weightgen_bwd_workspace_floats(int R, int T, int Hh, int num_hidden, int d_in, int d_out) {
    return weightgen_bwd_workspace_floats(R, T, Hh, num_hidden, d_in, d_out);
}

int ghf_weightgen_bwd(const float* text_emb, const float* const* head_params, const float* acts, const float* const* outs,
                      const float* const* grads, const float* const* log_scales, int R, int T, int Hh, int num_hidden, int d_in,
                      int d_out, const float* log_keep, float* const* dparams, float* const* dls, float* d_text_emb,
                      float* workspace, void* stream) {
    GHF_REQUIRE(text_emb && head_params && outs && grads && log_scales && dparams && dls && workspace,
                "weightgen_bwd: null pointer argument");
    return launch_weightgen_bwd(text_emb, head_params, acts, outs, grads, log_scales, R, T, Hh, num_hidden, d_in, d_out, log_keep,
                                dparams, dls, d_text_emb, workspace, (hipStream_t)stream);
}

int ghf_text_encode_bwd(const int32_t* ids, const int32_t* lens, int U, int Lmax, const float* char_emb, int V, int C,
                        const float* W, int T, const float* te, const float* dte, float* workspace, float* d_char_emb,
                        float* dW, float* db, void* stream) {
    GHF_REQUIRE(ids && lens && char_emb && W && te && dte && workspace && d_char_emb && dW && db, "text_encode_bwd: null pointer argument");
    return launch_text_encode_bwd(ids, lens, U, Lmax, char_emb, V, C, W, T, te, dte, workspace, d_char_emb, dW, db,
                                  (hipStream_t)stream);
}

int ghf_transpose_batched(const float* in, int batch, int rows, int cols, float* out, void* stream) {
    GHF_REQUIRE(in && out && in != out, "transpose_batched: null or aliased pointers");
    return launch_transpose_batched(in, batch, rows, cols, out, (hipStream_t)stream);
}

int ghf_weights_pack(const float* top, const float* bottom, int transpose, int R, int d, int wlayout, float* out, void* stream) {
    return launch_weights_pack(top, bottom, transpose, R, d, wlayout, out, (hipStream_t)stream);
}

int ghf_rows_pack(const void* rows, int64_t row_bytes, const void* extra, int64_t extra_bytes, const int64_t* idx, int64_t n,
                  int64_t nrows, void* packed, void* stream) {
    GHF_REQUIRE(rows && idx && packed, "rows_pack: null pointer argument");
    return launch_rows_pack(false, (void*)rows, row_bytes, (void*)extra, extra_bytes, idx, n, nrows, packed, (hipStream_t)stream);
}

int ghf_rows_unpack(const void* packed, const int64_t* idx, int64_t n, int64_t nrows, void* rows, int64_t row_bytes, void* extra,
                    int64_t extra_bytes, void* stream) {
    GHF_REQUIRE(rows && idx && packed, "rows_unpack: null pointer argument");
    return launch_rows_pack(true, rows, row_bytes, extra, extra_bytes, idx, n, nrows, (void*)packed, (hipStream_t)stream);
}

int ghf_score_pairs_fwd(const float* a, const float* b, const int64_t* ia, const int64_t* ib, int64_t rows_a, int64_t rows_b,
                        int64_t n, int d, float* scores, void* stream) {
    GHF_REQUIRE(a && b && (scores || n == 0), "score_pairs_fwd: null pointer argument");
    return launch_score_pairs(a, b, ia, ib, rows_a, rows_b, n, d, scores, (hipStream_t)stream);
}

int ghf_tail_fwd(const float* agg, const float* h, const float* ln_gamma, const float* ln_beta, float ln_eps,
                 int64_t row0, int64_t rows, int d, float* h_out, const float* drop, void* stream) {
    GHF_REQUIRE(agg && h && ln_gamma && ln_beta && h_out, "tail_fwd: null pointer argument");
    GHF_REQUIRE(row0 >= 0 && rows >= 0, "tail_fwd: bad row range");
    return launch_tail(agg, h, ln_gamma, ln_beta, ln_eps, row0, rows, d, h_out, drop, (hipStream_t)stream);
}

}  // extern "C"
