// common.h — shared helpers for libghf_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ghf.h"

namespace ghf {

// thread-local last-error text behind ghf_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

#define GHF_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return ghf::set_err(GHF_EHIP, "%s failed: %s (%s:%d)", #expr,                \
                                hipGetErrorString(_e), __FILE__, __LINE__);              \
    } while (0)

#define GHF_LAUNCH_CHECK() GHF_HIP_CHECK(hipGetLastError())

// Raise a kernel's dynamic-LDS limit once per process and size (one process drives one GPU), not per launch: the
// launch path stays free of non-stream API calls, so a warm forward can be captured into a HIP graph.
#define GHF_SET_MAX_LDS(kernel, bytes)                                                                          \
    do {                                                                                                        \
        static std::atomic<int> _ghf_lds_set{-1};                                                               \
        const int _ghf_b = (int)(bytes);                                                                        \
        if (_ghf_lds_set.load(std::memory_order_acquire) < _ghf_b) {                                            \
            GHF_HIP_CHECK(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, _ghf_b)); \
            _ghf_lds_set.store(_ghf_b, std::memory_order_release);                                              \
        }                                                                                                       \
    } while (0)

#define GHF_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) return ghf::set_err(GHF_EINVAL, __VA_ARGS__);                       \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int WAVE = 64;
constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;

// ---- wave / block reductions (64-lane wavefronts) --------------------------------
// Sum over the 64 lanes, result in every lane.  Data-parallel-primitive moves only (no LDS crossbar):
// butterfly inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 fold the four rows into lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {
    // value of the lane selected by CTRL for rows enabled in ROW_MASK, 0 elsewhere
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_take<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xF>(v);      // row_half_mirror
    v += dpp_take<0x140, 0xF>(v);      // row_mirror: every lane of a row holds the row's sum
    v += dpp_take<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// max over the wave of non-negative values (0 is the identity the disabled DPP rows contribute)
__device__ __forceinline__ float wave_absmax(float v) {
    v = fmaxf(v, dpp_take<0xB1, 0xF>(v));
    v = fmaxf(v, dpp_take<0x4E, 0xF>(v));
    v = fmaxf(v, dpp_take<0x141, 0xF>(v));
    v = fmaxf(v, dpp_take<0x140, 0xF>(v));
    v = fmaxf(v, dpp_take<0x142, 0xA>(v));
    v = fmaxf(v, dpp_take<0x143, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// GHF_WLAYOUT_SPLIT2H pieces.  s = split2h_shift(largest magnitude of the row / matrix) lifts it into [2^13, 2^14);
// x 2^s = hi + lo + eps with hi = fp16(x 2^s), lo = fp16(x 2^s - hi), |eps| <= 2^-22 |x 2^s|.  |s| <= 100 keeps 2^s and
// 2^-s normal fp32 numbers (rows below 2^-87 or above 2^113 lose precision; nothing sane is there).
__device__ __forceinline__ int split2h_shift(float maxabs) {
    const int e = ((__float_as_int(maxabs) >> 23) & 255) - 127;
    const int s = 13 - e;
    return s < -100 ? -100 : (s > 100 ? 100 : s);
}
__device__ __forceinline__ float pow2f(int s) { return __int_as_float((s + 127) << 23); }   // |s| <= 126
__device__ __forceinline__ void split2h(float xs, _Float16& hi, _Float16& lo) {
    hi = (_Float16)xs;
    lo = (_Float16)(xs - (float)hi);
}

// ---- range guard of the two-fp16-piece form (include/ghf.h: ghf_set_range_flag) -----------------------------------
// x 2^s = hi + lo keeps 22 bits of every element within 2^-14 of the row's (matrix's) largest; smaller ones lose bits and
// below 2^-38 of it vanish.  A row / matrix in which at least 1/8 of the NONZERO entries lie that far down is "wide":
// the kernels that cut rows or weights OR a bit into the registered flag word, and the host routes the forward to the exact
// fp32 kernels (models/hypergnn.py).  Scaled values: the largest is in [2^13, 2^14), so "that far down" is |x 2^s| < 0.5.
int32_t* range_flag_ptr();                                 // capi.hip: the registered device word, or nullptr
// (GHF_RANGE_ROWS, GHF_RANGE_WEIGHTS: include/ghf.h)
__device__ __forceinline__ int range_tiny(float xs) { const float a = fabsf(xs); return a != 0.f && a < 0.5f; }
__device__ __forceinline__ void range_raise(int32_t* flag, int bit, int tiny, int nonzero) {
    if (flag && tiny > 0 && tiny * 8 >= nonzero) atomicOr(flag, bit);
}
// GHF_RANGE_WEAK_W (include/ghf.h): one relation's [2d, d] matrix has an input row k whose L1 norm s[k] is below
// d 2^-16 of the largest row's — the case in which far-down entries of a row of h (or of the matrix) can carry the whole
// result.  Without such a row both error terms of the two-piece product stay within 2^-22 sum |x_k w_k| (ghf.h).
// A wave's rows: (smallest, largest) row norm of each half [W_msg; W_self]; a half that is zero throughout (the backward's
// GHF_FLAG_ZERO_* passes) takes no part.
struct WeakRows {
    float mn[2], mx[2];
    __device__ __forceinline__ void init() { mn[0] = mn[1] = 3.0e38f; mx[0] = mx[1] = 0.f; }
    __device__ __forceinline__ void add(int half, float s) { mn[half] = fminf(mn[half], s); mx[half] = fmaxf(mx[half], s); }
    __device__ __forceinline__ void merge(const WeakRows& o) {
        for (int h = 0; h < 2; ++h) { mn[h] = fminf(mn[h], o.mn[h]); mx[h] = fmaxf(mx[h], o.mx[h]); }
    }
    __device__ __forceinline__ bool weak(int d) const {
        const float top = fmaxf(mx[0], mx[1]), thr = top * ((float)d * (1.0f / 65536.0f));
        return (mx[0] > 0.f && mn[0] < thr) || (mx[1] > 0.f && mn[1] < thr);
    }
};
__device__ __forceinline__ void range_raise_weak(int32_t* flag, const WeakRows& w, int d) {
    if (flag && w.weak(d)) atomicOr(flag, 4 /* GHF_RANGE_WEAK_W */);
}

// Sum over a block of NWAVES*64 threads; `red` is >= NWAVES floats of LDS.
// All threads get the result.  Contains two barriers.
template <int NWAVES>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if constexpr (NWAVES == 1) return v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) t += red[i];
    return t;
}

// Launchers (one per translation unit), called from capi.hip.
int launch_plan_build(const int64_t* edge_index, const int64_t* rel_id, int64_t N, int64_t E, int R,
                      int block_nodes, int chunk_rows, int split_chunks, void* ws, size_t ws_bytes,
                      uint32_t* sorted_key, int32_t* sorted_src, int32_t* seg_off, int32_t* indeg,
                      int32_t* chunk_tab, int32_t* blk_chunk_off, int32_t* item_tab, int32_t* blk_item_off,
                      int32_t* status, hipStream_t stream);
size_t plan_workspace_bytes(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows);
int64_t plan_max_chunks(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows);
int64_t plan_max_items(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows, int split_chunks);
constexpr int SRC_BITS = 28;                          // sorted_src: node id in bits 0..27, run head in 28..31
constexpr int32_t SRC_MASK = (1 << SRC_BITS) - 1;

int launch_weightgen(const float* text_emb, const float* const* head_params, const float* const* log_scales,
                     int R, int T, int Hh, int num_hidden, int d_in, int d_out, int layout,
                     float* hidden_ws, float* W_msg, float* W_self, float* bias, const float* hidden_drop, hipStream_t stream,
                     float* acts = nullptr);
int launch_weightgen_batched(int L, const float* text_emb, const float* const* head_params, const float* const* log_scales,
                             int R, int T, int Hh, int num_hidden, int d_in, int d_out, int layout,
                             float* hidden_ws, float* const* W_msg, float* const* W_self, float* const* bias,
                             const float* hidden_drop, hipStream_t stream, float* acts = nullptr);

int launch_rows_pack(bool unpack, void* rows, int64_t row_bytes, void* extra, int64_t extra_bytes, const int64_t* idx, int64_t n,
                     int64_t nrows, void* packed, hipStream_t stream);
int launch_score_pairs(const float* a, const float* b, const int64_t* ia, const int64_t* ib, int64_t rows_a, int64_t rows_b,
                       int64_t n, int d, float* scores, hipStream_t stream);
int launch_text_encode(const int32_t* ids, const int32_t* lens, int U, int Lmax, const float* E, int V, int C,
                       const float* W, const float* b, int T, float* out, hipStream_t stream);
int launch_input_proj(const float* x, const float* W_in, const float* b_in, int64_t N, int F, int d,
                      float* h0, void* h_split, int split_layout, hipStream_t stream);


struct MsgArgs {
    const float* h; const void* h_split; int64_t N; int d;
    const uint32_t* sorted_key; const int32_t* sorted_src; const int32_t* seg_off; const int32_t* indeg;
    const int32_t* chunk_tab; const int32_t* blk_chunk_off;
    const int32_t* item_tab; const int32_t* blk_item_off; int64_t item0; int64_t n_items; float* partial;
    int64_t E; int R; int block_nodes;
    const float* W_msg; const float* W_self; const float* bias; int wlayout;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    int64_t row0; int64_t rows; float* h_out; void* h_split_out; int flags;
    float* agg_out;              // optional side output: the aggregate before the tail (ghf.h)
};
int launch_split2h_rows(const float* h, int64_t N, int d, int64_t row0, int64_t rows, void* h_split, hipStream_t stream);
int launch_message_bx(const MsgArgs& a, hipStream_t stream);       // the same contraction, block sums in registers (message_bx.hip)
bool message_bx_config(int d, int* block_nodes, int* chunk_rows, int* split_chunks);
bool message_bx_owns(int d, int block_nodes);                      // whether a SPLIT2H plan of this geometry is message_bx's
int launch_message_generic(const MsgArgs& a, hipStream_t stream);
int launch_message_pp(const MsgArgs& a, hipStream_t stream);       // exact fp32 MFMA, ping-pong schedule (d = 128, 64), FRAG16 weights
bool message_pp_config(int d, int* block_nodes, int* chunk_rows, int* split_chunks);
int launch_combine_split(const MsgArgs& a, hipStream_t stream);     // sums the partial slots of split blocks + tail

size_t group_workspace_bytes(int64_t E);
int launch_group_edges(const int64_t* rel, int64_t E, int R, void* ws, size_t ws_bytes, int64_t* perm, int64_t* goff,
                       hipStream_t stream);
// backward.hip
int launch_tail_bwd(const float* g_out, const float* agg, const float* h, const float* gamma, float eps, const int32_t* indeg,
                    int64_t N, int d, float* dpre, float* G, void* G_split, float* dgb, float* workspace, const float* drop,
                    hipStream_t stream);
size_t tail_bwd_workspace_floats(int64_t N, int d);
size_t colsum_workspace_floats(int64_t N, int d);
int launch_colsum(const float* X, const float* mask, int64_t N, int d, float* workspace, float* out, int accumulate, hipStream_t stream);
int launch_relu_mask(const float* X, const float* ref, int64_t n, float* out, hipStream_t stream);
int launch_group_outer(const float* A, const int64_t* ia, int da, const float* B, const int64_t* ib, int db,
                       const int64_t* gstart, const int64_t* gend, int ngroups, float* C, int accumulate, hipStream_t stream);
int launch_scale_exp(const float* X, int64_t n, const float* log_scale, float* out, hipStream_t stream);
int launch_add3(const float* a, const float* b, const float* c, int64_t n, float* out, hipStream_t stream);
int launch_segment_axpy(const float* w, const int64_t* iw, const float* X, const int64_t* ix, const int64_t* off, int64_t nseg,
                        int64_t nx, int d, float* out, hipStream_t stream);
int launch_rowscale(const float* X, const float* g, int64_t n, int d, float* out, hipStream_t stream);
int edge_outer_supported(int d);
int launch_edge_outer(const float* h, const float* G, const int64_t* src, const int64_t* dst, const int64_t* slice_tab,
                      const int64_t* slice_off, int64_t nslices, int R, int d, int64_t N, float* workspace, float* dW, float* db,
                      hipStream_t stream, const float* h_rowscale = nullptr,
                      const float* G_rowscale = nullptr, const int32_t* order = nullptr);
int message_rs_supported(int d);
int launch_edge_transform(const float* h, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                          const int64_t* slice_tab, int64_t nslices, const float* WmT, const float* WsT, const float* bias,
                          float* Y, hipStream_t stream);
size_t weights_rs_bytes(int R, int d);
int launch_weights_pack_rs(const float* Wm, const float* Ws, int R, int d, void* out, int* shift_ws, hipStream_t stream);
int launch_edge_transform_h(const void* h_split, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                            const int64_t* slice_tab, int64_t nslices, const void* w2h, int R, const float* bias,
                            const void* x_split, int64_t NX, const float* row_cnt, float* Y, hipStream_t stream);
int launch_run_rows(const float* h, int64_t N, int d, const int64_t* run_src, const int64_t* run_start, int64_t nruns,
                    void* x_split, hipStream_t stream);
int launch_segment_partial(const float* Y, const int64_t* hub_chunks, int64_t nchunks, int d, float* P, hipStream_t stream);
int launch_segment_tail(const float* Y, const int64_t* off, const int32_t* deg_of, const int32_t* hub_of, const int64_t* hub_tab, const float* P,
                        const float* h, const float* g, const float* b, float eps, int64_t row0, int64_t rows, int d,
                        float* h_out, void* h_split_out, int64_t n_split, int flags, hipStream_t stream);
int launch_dot(const float* X, const float* Y, int64_t n, float* workspace, float* out, hipStream_t stream);
int weightgen_bwd_supported(int T, int Hh, int num_hidden);
size_t weightgen_bwd_workspace_floats(int R, int T, int Hh, int num_hidden, int d_in, int d_out);
int launch_weightgen_bwd(const float* text_emb, const float* const* head_params, const float* acts, const float* const* outs,
                         const float* const* grads, const float* const* log_scales, int R, int T, int Hh, int num_hidden, int d_in,
                         int d_out, const float* log_keep, float* const* dparams, float* const* dls, float* dx, float* workspace,
                         hipStream_t stream);
int launch_weightgen_acts(const float* text_emb, const float* const* head_params, int R, int T, int Hh, int num_hidden,
                          float* acts, const float* hidden_drop, hipStream_t stream);
int launch_text_encode_bwd(const int32_t* ids, const int32_t* lens, int U, int Lmax, const float* E, int V, int C, const float* W,
                           int T, const float* te, const float* dte, float* workspace, float* dE, float* dW, float* db,
                           hipStream_t stream);
int launch_transpose_batched(const float* in, int batch, int rows, int cols, float* out, hipStream_t stream);
int launch_weights_pack(const float* top, const float* bottom, int transpose, int R, int d, int layout, float* out,
                        hipStream_t stream);

int launch_tail(const float* agg, const float* h, const float* g, const float* b, float eps,
                int64_t row0, int64_t rows, int d, float* h_out, const float* drop, hipStream_t stream);

}  // namespace ghf
