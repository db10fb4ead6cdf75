/* ghf.h — C ABI of libghf_hip.so: the MI355X (gfx950) HyperGNN forward hot path.
 *
 * The reference (danieleschmidt/Graph-Hypernetwork-Forge v0.2.0) has no FFI or
 * plugin interface; its boundary for this path is two nn.Module call signatures
 * (SURVEY.md §8b).  Each entry point below replaces a span of stock ATen CPU ops
 * inside those two calls; the span is cited as reference file:line (paths
 * relative to graph_hypernetwork_forge/).  The Python host mirror that binds
 * these with ctypes is graph-hypernetwork-forge_amd/_native.py; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; every pointer is DEVICE memory owned by the
 *    caller (torch's caching allocator in the Python host).  Inputs are never
 *    written.  Nothing is allocated, freed or synchronised inside any call.
 *  - Every call enqueues on `stream` (a hipStream_t passed as void*) and returns
 *    immediately; calls are re-entrant and graph-capturable.
 *  - Return 0 on success, a negative GHF_E* code otherwise; ghf_last_error()
 *    gives a thread-local message.
 *  - All floating point is IEEE fp32; indices are int64 at the API (as in
 *    models/hypergnn.py:191) and int32/uint32 inside a plan.
 */
#ifndef GHF_H
#define GHF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GHF_ABI_VERSION 15

#define GHF_OK            0
#define GHF_EINVAL       -1   /* bad argument (shape, alignment, unsupported size) */
#define GHF_EHIP         -2   /* a HIP runtime call failed */
#define GHF_EUNSUPPORTED -3   /* no kernel for this configuration */

/* ghf_message_layer_fwd flags */
#define GHF_FLAG_NO_TAIL  1   /* write sum_e(...) / max(indeg,1) only: skip residual+ReLU+LayerNorm */
#define GHF_FLAG_RAW_SUM  2   /* (implies NO_TAIL) write sum_e(...) itself, no division: the backward passes */
#define GHF_FLAG_ZERO_SRC 4   /* the kernel MUST leave the source half of every relation's weights (W_msg) out: no source-row
                                 gathers, no products with that half — whatever the packed tensor holds there (the backward
                                 hands ONE packed [W_msg^T; W_self^T] to both of its gradient passes and names the half each
                                 pass must not read; a half that really is zero gives the same result without the flag).
                                 Honoured by the kernels with the side output (ghf_message_side_output_supported) only:
                                 elsewhere the call returns GHF_EUNSUPPORTED — pack that half as zeros and drop the flag */
#define GHF_FLAG_ZERO_DST 8   /* the same for the destination half (W_self); not both */
#define GHF_FLAG_ADD_H   16   /* with NO_TAIL / RAW_SUM: h_out = (that result) + h — `h` then only names the rows to add (the gathers use
                                 h_split): the backward accumulates its gradient terms this way.  Kernels with the side
                                 output (ghf_message_side_output_supported) only */

/* Weight layouts produced by ghf_weightgen_fwd and consumed by ghf_message_layer_fwd */
#define GHF_WLAYOUT_NATURAL 0 /* W_msg[R][d_in][d_out], W_self[R][d_in][d_out] row-major, as the reference returns them */
#define GHF_WLAYOUT_FRAG16  1 /* MFMA 16x16x4 B-fragment order: Wfrag[R][d/16][2d/16][64 lanes][4] */
/*      (2: three exact bf16 pieces — retired with its kernel in ABI 10; the number is not reused) */
#define GHF_WLAYOUT_SPLIT2H 3 /* fp16 MFMA 16x16x32 B-fragment order, every weight of relation r scaled by a power of two
                                 2^s(r) and cut into 2 fp16 pieces: Wh[R][d/16][2d/32][2 pieces][64 lanes][8] fp16 = 4 bytes
                                 per weight, followed by float 2^-s(r) [R] */

int         ghf_abi_version(void);
const char* ghf_last_error(void);

/* HOST function (no device work, no stream): a position-dependent 64-bit checksum of `nbytes` bytes (a multiple of 8) at `p`.
 * The plan cache confirms a hit on a long relation list with it: the list's array of object pointers against the checksums
 * taken when the plan was built — the per-call string -> id mapping of models/hypergnn.py:264-268 at memory speed. */
unsigned long long ghf_host_checksum64(const void* p, size_t nbytes, unsigned long long seed);
/* HOST function: dense ids of an array of `n` machine words (a relation list's object pointers) in first-appearance order —
 * ids[i] (int64, what ghf_plan_build takes) = the rank of words[i] among the distinct words by first appearance, uniq[k] = the
 * k-th distinct word.  Returns the number of distinct words, or -1 when there are more than max_uniq (the caller then falls
 * back to hashing the strings).  `threads` > 1: the array beyond a prefix is mapped by that many host threads (same result).
 * The reference's dict.fromkeys pass (models/hypergnn.py:264-268) for lists that reference a few string objects many times. */
long long ghf_host_word_ids(const void* words, long long n, long long* ids, void** uniq, long long max_uniq, int threads);

/* Which plan geometry and weight layout the message kernel for hidden size d wants.
 * block_nodes == 1 means "CSR by destination" (the generic kernel; chunk_rows == split_chunks == 0 then). */
int ghf_message_config(int d, int* block_nodes, int* wlayout, int* chunk_rows, int* split_chunks);

/* ---- K0: graph plan -----------------------------------------------------------
 * Replaces the implicit edge order of models/hypergnn.py:191 (src,dst = edge_index)
 * and the in-degree count of :207-212.  Sorts edges by (dst / block_nodes, rel,
 * dst % block_nodes), i.e. key = (dst/BN)*R*BN + rel*BN + dst%BN, and emits
 *   sorted_key [E] uint32,
 *   sorted_src [E] int32: the source node id; for BN > 1 bits 28..31 also hold the edge's
 *              "run head": with t = (position in its (block, relation) group) % 16, the
 *              smallest t' <= t such that positions t'..t of that 16-row tile all have the
 *              same destination (needs N <= 2^28),
 *   seg_off [nseg+1] int32 with nseg = ceil(N/BN)*R  (BN > 1)  or  N  (BN == 1: CSR rows),
 *   indeg [N] int32,
 *   chunk_tab [2*max_chunks] int32 and blk_chunk_off [ceil(N/BN)+1] int32 (BN > 1 only, else
 *              NULL): every (block, relation) group cut into chunks of <= chunk_rows edges;
 *              chunk c = { first sorted edge, (rel << 8) | (cross << 7) | rows }, cross = a run of
 *              equal destinations spans a 16-row tile boundary inside the chunk; block b owns
 *              chunks [blk_chunk_off[b], blk_chunk_off[b+1]),
 *   item_tab [4*max_items] int32 and blk_item_off [ceil(N/BN)+1] int32 (BN > 1 only): the work items.
 *              A block with more than split_chunks chunks (the hub of a power-law graph) is cut into
 *              ceil(chunks/split_chunks) items; item i = { block, first chunk, one past its last chunk,
 *              scratch slot (-1 for the only item of a block) }; block b owns items
 *              [blk_item_off[b], blk_item_off[b+1]),
 *   status [3] int32: [0] 0 ok, bit0 = a src/dst outside [0,N), bit1 = a rel outside [0,R) (read it back
 *              before trusting the plan; offending edges are dropped); [1] number of items; [2] number of
 *              scratch slots (each BN*d floats) ghf_message_layer_fwd needs in `partial`.
 * Requires ceil(N/BN)*BN*R < 2^32, E < 2^31, R < 2^23, chunk_rows % 4 == 0 (% 16 for the kernels that read the run heads), chunk_rows < 128, split_chunks > 0. */
size_t  ghf_plan_workspace_bytes(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows);
int64_t ghf_plan_max_chunks(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows);
int64_t ghf_plan_max_items(int64_t N, int64_t E, int R, int block_nodes, int chunk_rows, int split_chunks);
int ghf_plan_build(const int64_t* edge_index /* [2,E] row 0 = src, row 1 = dst */,
                   const int64_t* rel_id /* [E] */, int64_t N, int64_t E, int R, int block_nodes,
                   int chunk_rows, int split_chunks, void* workspace, size_t workspace_bytes,
                   uint32_t* sorted_key, int32_t* sorted_src, int32_t* seg_off, int32_t* indeg,
                   int32_t* chunk_tab, int32_t* blk_chunk_off, int32_t* item_tab, int32_t* blk_item_off,
                   int32_t* status, void* stream);

/* ---- K1: weight generation ------------------------------------------------------
 * Replaces models/weight_generator.py:137-141 (three nn.Sequential heads, reshape,
 * * exp(log_scale)) for B = R relation embeddings at once.
 *  head_params: 3 heads (W_msg, W_self, bias) x (num_hidden+1) Linear layers x {weight,bias},
 *               flattened as head_params[(head*(num_hidden+1) + layer)*2 + {0,1}];
 *               weights are [out,in] row-major as nn.Linear stores them.
 *  log_scales: host array of three device pointers, one float each (the reference's three 1-element parameters
 *               log_scales.{W_msg,W_self,bias}, weight_generator.py:85-88, read in place).
 *  hidden_ws:   scratch for hidden activations, >= 3*2*R*max(Hh,T) floats.
 *  layout NATURAL: W_msg,W_self [R,d_in,d_out], bias [R,d_out] (any d_in,d_out).
 *  layout FRAG16:  requires d_in == d_out == d, d % 16 == 0; W_msg is the combined
 *                  fragment buffer of 2*R*d*d floats and W_self must be NULL.
 *  hidden_drop:    training with dropout > 0 (the reference's Linear -> ReLU -> Dropout, weight_generator.py:96-107): the masks,
 *                  already scaled by 1/(1-p), as floats [3 heads][num_hidden][R][Hh] (ghf_weightgen_acts' layout), multiplied
 *                  into every hidden activation; NULL: none.  The caller draws them (the reference draws them with torch's
 *                  generator; so does the Python mirror).
 *  acts:           NULL, or where the same launch leaves every hidden layer's output (ghf_weightgen_acts' result and layout:
 *                  what the backward needs besides the outputs). */
int ghf_weightgen_fwd(const float* text_emb /* [R,T] */, const float* const* head_params,
                      const float* const* log_scales /* [3] host array of device pointers */, int R, int T, int Hh, int num_hidden,
                      int d_in, int d_out, int layout, float* hidden_ws,
                      float* W_msg, float* W_self, float* bias, const float* hidden_drop /* or NULL */, float* acts /* or NULL */,
                      void* stream);

/* The L generators of one model (identical shapes: the reference builds one WeightGenerator per layer, hypergnn.py:131-143) in
 * ONE launch sequence — hidden layers, output layers, packing: three kernels for all layers instead of three per layer; on
 * small graphs the generators' launches are a large share of a forward (BASELINE configs 1 and 2).  head_params: generator
 * g's pointers at [g * 3 * (num_hidden + 1) * 2 ...] in ghf_weightgen_fwd's order; log_scales [L * 3]; hidden_ws: L times
 * ghf_weightgen_fwd's size; W_msg / W_self / bias: host arrays of L device pointers (W_self NULL or its entries NULL where
 * the layout has none).  Inference only (no dropout masks).  L <= 8.  Same values as L calls of ghf_weightgen_fwd. */
int ghf_weightgen_fwd_batched(int L, const float* text_emb, const float* const* head_params, const float* const* log_scales,
                              int R, int T, int Hh, int num_hidden, int d_in, int d_out, int layout, float* hidden_ws,
                              float* const* W_msg, float* const* W_self, float* const* bias, void* stream);

/* ---- text encoder ------------------------------------------------------------------
 * Replaces models/hypergnn.py:39-81 (TextEncoder) for U strings at once:
 *   out[u] = tanh( mean_{c < lens[u]} char_emb[ids[u][c]] . W^T + b )
 * ids [U, Lmax] int32 = min(ord(ch), 127) per character, padded ('' is the single id 0: lens[u] >= 1); ids outside
 * [0, V) are clamped.  char_emb [V, C], W [T, C] and b [T] as nn.Embedding / nn.Linear store them. */
int ghf_text_encode_fwd(const int32_t* ids, const int32_t* lens, int U, int Lmax,
                        const float* char_emb, int V, int C, const float* W, const float* b, int T,
                        float* out /* [U,T] */, void* stream);

/* ---- input projection -------------------------------------------------------------
 * Replaces models/hypergnn.py:261: h0 = relu(x @ W_in^T + b_in).  x [N,F], W_in [d,F]. */
int ghf_input_proj_fwd(const float* x, const float* W_in, const float* b_in,
                       int64_t N, int F, int d, float* h0,
                       void* h_split /* optional: the rows of h0 as ghf_split_rows(split_layout) would write them */,
                       int split_layout, void* stream);

/* ---- K2+K3: one message-passing layer ---------------------------------------------
 * Replaces models/hypergnn.py:281-296: per-edge weight gather, h_u @ W_msg[r] + bias[r],
 * mean at targets, mean-W_self self-loop, residual, ReLU, LayerNorm — as
 *   out_v = (1/max(indeg_v,1)) * sum_{e=(u->v)} (h_u W_msg[r_e] + bias[r_e] + h_v W_self[r_e])
 *   h'_v  = LayerNorm(ReLU(out_v + h_v))              
 * for destination rows [row0, row0+rows) (row0 % block_nodes == 0); other rows of
 * h_out are not touched.  The plan arrays must come from ghf_plan_build with the
 * same N, E, R, block_nodes; W/bias from ghf_weightgen_fwd with `wlayout`.
 * item0 / n_items: the work items of the blocks of the row range, i.e. blk_item_off[row0/BN] and
 * blk_item_off[ceil((row0+rows)/BN)] - item0 (host copies of two plan words); partial: scratch of
 * status[2]*BN*d floats for the split blocks (may be NULL when status[2] == 0).
 * h_split: the rows of h cut into pieces by ghf_split_rows (or by a previous call's h_split_out); required when
 * wlayout is SPLIT2H, ignored otherwise.  h_split_out (optional, that layout only, not with
 * GHF_FLAG_NO_TAIL): receives the split form of the h_out rows written by this call, for the next layer.
 * agg_out (optional, not with GHF_FLAG_NO_TAIL; ghf_message_side_output_supported): also receives, for the rows written,
 * the aggregate before the tail — what GHF_FLAG_NO_TAIL would have written to h_out — which a training forward keeps for
 * ghf_tail_bwd (the reference's autograd keeps the same tensor, hypergnn.py:281-296). */
int ghf_message_side_output_supported(int d, int block_nodes, int wlayout);
int ghf_message_layer_fwd(const float* h /* [N,d] */, const void* h_split /* ghf_split_rows output or NULL */, int64_t N, int d,
                          const uint32_t* sorted_key, const int32_t* sorted_src,
                          const int32_t* seg_off, const int32_t* indeg,
                          const int32_t* chunk_tab, const int32_t* blk_chunk_off,
                          const int32_t* item_tab, const int32_t* blk_item_off,
                          int64_t item0, int64_t n_items, float* partial,
                          int64_t E, int R, int block_nodes,
                          const float* W_msg, const float* W_self, const float* bias, int wlayout,
                          const float* ln_gamma, const float* ln_beta, float ln_eps,
                          int64_t row0, int64_t rows, float* h_out /* [N,d] */,
                          void* h_split_out /* or NULL */, float* agg_out /* [N,d] or NULL */, int flags, void* stream);

/* Rows [row0, row0+rows) of h [N,d] in the form the message kernel of `wlayout` gathers (the split is done once per
 * row here instead of once per edge there); h_split holds ghf_split_rows_bytes(N, d, wlayout) bytes:
 *   SPLIT2H: h_split[v][2 pieces][d] fp16 — x 2^s(v) = hi + lo (22 significand bits), 2^s(v) lifting the row's largest
 *            magnitude into [2^13, 2^14) — followed by float 2^-s(v) [N].
 * Other layouts gather h itself (ghf_split_rows_bytes == 0).  d % 4 == 0. */
size_t ghf_split_rows_bytes(int64_t N, int d, int wlayout);
int ghf_split_rows(const float* h, int64_t N, int d, int64_t row0, int64_t rows, int wlayout, void* h_split, void* stream);

/* Bytes of the W_msg buffer ghf_weightgen_fwd fills in `wlayout` (NATURAL: one of the two [R,d_in,d_out] matrices). */
size_t ghf_weights_bytes(int R, int d_in, int d_out, int wlayout);

/* ---- range guard of the two-fp16-piece forms ------------------------------------------------------------------------
 * GHF_WLAYOUT_SPLIT2H rows and weights keep 22 significand bits of every element within 2^-14 of the largest magnitude of
 * their row (activations) or of their relation's [2d, d] matrix (weights); an element further down loses one bit per
 * factor of two and vanishes below 2^-38 of the largest.  Bound on a product sum_k x_k w_k: 3 * 2^-22 * sum |x_k w_k|
 * (the order of the fp32 fma chain it replaces) + 2^-38 * max|x| * sum_k |w_k| over the far-down x_k (and the same with x
 * and w exchanged).  The second term is invisible unless the large entries meet (near-)zero partners, e.g. one feature
 * 2^30 above the rest whose weight column is 0 — the reference's fp32 bmm (hypergnn.py:202,228) has no such case.
 * Three conditions are watched by every kernel that cuts rows or weights; each ORs a bit into the int32 device word
 * registered here (NULL: no guard):
 *   GHF_RANGE_ROWS / GHF_RANGE_WEIGHTS — per row / per relation matrix (per 32 x 32 tile in ghf_weights_pack_rs) the nonzero
 *     entries more than 2^14 below the largest are at least 1/8 of the nonzero entries;
 *   GHF_RANGE_WEAK_W — some relation's [W_msg; W_self] has an input row whose L1 norm is below d * 2^-16 of the largest
 *     row's (a half that is zero throughout, as in the backward's GHF_FLAG_ZERO_* passes, takes no part).
 * GUARANTEE while GHF_RANGE_WEAK_W is clear (s[k] = L1 norm of input row k, all s[k] >= d 2^-16 max s): summed over a
 * result row's d outputs, the second term is at most 2^-38 * max|x| * d * max s on either side, and sum_o sum_k |x_k w_ko| =
 * sum_k |x_k| s[k] >= max|x| * d 2^-16 * max s — i.e. the whole error stays within 4 * 2^-22 * sum |x_k w_ko|, the kind
 * of bound the fp32 fma chain has, however few or many entries of a row or a matrix lie far down.  A matrix with a weak
 * row is where a far-down entry can carry a result on its own (rows whose large entries meet that row); generated dense
 * weights do not have one.
 * The word belongs to the caller, who clears it before a forward and reads it after (the host mirror then repeats the
 * forward on the exact fp32 kernels when any bit is set).  One word per process (one process drives one GPU). */
#define GHF_RANGE_ROWS 1
#define GHF_RANGE_WEIGHTS 2
#define GHF_RANGE_WEAK_W 4
int ghf_set_range_flag(int32_t* device_word);

/* ---- K3 alone -------------------------------------------------------------------------
 * Replaces models/hypergnn.py:288-296 on rows [row0,row0+rows): agg already holds
 * out_v (GHF_FLAG_NO_TAIL output, e.g. after a cross-GPU reduction).  drop (training with dropout > 0, :293-294): the
 * mask, scaled by 1/(1-p), multiplied in between ReLU and LayerNorm; ghf_tail_bwd takes the same mask. */
int ghf_tail_fwd(const float* agg /* [N,d] */, const float* h /* [N,d] */,
                 const float* ln_gamma, const float* ln_beta, float ln_eps,
                 int64_t row0, int64_t rows, int d, float* h_out, const float* drop /* [N,d] or NULL */, void* stream);

/* ---- wide hidden sizes: the relation-stationary message layer (csrc/message_rs.hip) ---------------------------------
 * For d % 128 == 0, 128 <= d <= 1024 (ghf_message_rs_supported; BASELINE config 5, and d = 128 with many relations).  Same statement as
 * ghf_message_layer_fwd, in two passes over a CSR plan (block_nodes == 1):
 *   ghf_edge_transform_fwd: Y[ypos[k]] = h[src[k]] W_msg[r] + bias[r] + h[dst[k]] W_self[r] for the edges k in RELATION order
 *       (src / dst / ypos [E] int64: the edge's ends and its position in destination order); slice_tab [nslices][3] int64 =
 *       (relation, first edge, end edge) cuts every relation's range into tiles of at most 128 edges; WmT / WsT [R][d][d]
 *       are the weights TRANSPOSED ([r][out][in], ghf_transpose_batched of the natural layout); Y [E][d] fp32 scratch.
 *   ghf_edge_transform_h_fwd: the same per-edge results with the two-fp16-piece contraction of the d = 128 kernel (three
 *       16x16x32 products, 22 significand bits, 5x less matrix time): h_split = ghf_split_rows(h, GHF_WLAYOUT_SPLIT2H), w2h =
 *       ghf_weights_pack_rs(W_msg, W_self natural [R][d][d]) (ghf_weights_rs_bytes(R, d) bytes; shift_ws: R ints of scratch).
 *       Rows that stand for several edges (graphs with hubs): the layer is linear in the source rows, so the n edges of one
 *       (destination, relation) run need one row of this pass:  sum_e (h_u W_msg[r] + b[r] + h_v W_self[r]) =
 *       x W_msg[r] + n (b[r] + h_v W_self[r]),  x = the sum of the run's source rows.  ghf_run_rows_fwd writes those sums:
 *       x_split row i = sum of h[run_src[run_start[i] .. run_start[i+1])] (fp32, in that order) in ghf_split_rows form
 *       (ghf_split_rows_bytes(nruns, d, SPLIT2H) bytes).  A row k of this pass then has src[k] < 0 = row ~src[k] of x_split
 *       (NX rows; NULL / 0: none) and row_cnt[k] = n (float; NULL: every row is one edge).
 *   ghf_segment_partial_fwd (hubs only): P[slot] = sum(Y[first .. end)) for every (first, end, slot) of hub_chunks
 *       [nchunks][3] int64 — a destination with more rows than one wave should walk is summed in chunks first.
 *   ghf_segment_tail_fwd: rows [row0, row0+rows): out_v = sum(Y[off[v] .. off[v+1])) / max(indeg, 1), then the tail of
 *       ghf_tail_fwd (flags: GHF_FLAG_NO_TAIL / GHF_FLAG_RAW_SUM as for the message layer); off [N+1] int64.  Hubs:
 *       hub_of [N] int32 (hub index or -1; NULL = no hubs), hub_tab [H][2] int64 (first slot, slots) select rows of P
 *       to add instead of rows of Y; the mean divides by deg_of[v] (int32 [N]: the in-degree, when rows stand for several
 *       edges) or, deg_of == NULL, by off[v+1] - off[v].  h_split_out (optional): the rows
 *       written, also in ghf_split_rows(.., GHF_WLAYOUT_SPLIT2H) form for the next layer's ghf_edge_transform_h_fwd — a buffer
 *       of n_split rows (ghf_split_rows_bytes(n_split, d, GHF_WLAYOUT_SPLIT2H)). */
size_t ghf_weights_rs_bytes(int R, int d);
int ghf_weights_pack_rs(const float* W_msg, const float* W_self, int R, int d, void* w2h, int* shift_ws, void* stream);
int ghf_edge_transform_h_fwd(const void* h_split, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                             const int64_t* slice_tab, int64_t nslices, const void* w2h, int R, const float* bias,
                             const void* x_split, int64_t NX, const float* row_cnt, float* Y, void* stream);
int ghf_run_rows_fwd(const float* h, int64_t N, int d, const int64_t* run_src, const int64_t* run_start, int64_t nruns,
                     void* x_split, void* stream);
int ghf_segment_partial_fwd(const float* Y, const int64_t* hub_chunks, int64_t nchunks, int d, float* P, void* stream);
int ghf_message_rs_supported(int d);
int ghf_edge_transform_fwd(const float* h, int64_t N, int d, const int64_t* src, const int64_t* dst, const int64_t* ypos,
                           const int64_t* slice_tab, int64_t nslices, const float* WmT, const float* WsT, const float* bias,
                           float* Y, void* stream);
int ghf_segment_tail_fwd(const float* Y, const int64_t* off, const int32_t* deg_of, const int32_t* hub_of, const int64_t* hub_tab, const float* P,
                         const float* h, const float* ln_gamma, const float* ln_beta, float ln_eps, int64_t row0,
                         int64_t rows, int d, float* h_out, void* h_split_out, int64_t n_split, int flags, void* stream);

/* ---- backward of the path (the reference trains through it with autograd: demo.py:79-101) ----------
 * With out_v = (1/c_v) sum_e(h_u Wm[r] + b[r] + h_v Ws[r]), x = relu(out + h), h' = LayerNorm(x), g' = dL/dh':
 *   ghf_tail_bwd   : dpre = dL/d(out+h) (also the residual's share of dL/dh), G_v = dpre_v / c_v, and the LayerNorm's
 *                    dgamma_dbeta [2][d] = (sum_v g'_v * xhat_v, sum_v g'_v), summed in a fixed order on the way;  agg is the
 *                    forward's GHF_FLAG_NO_TAIL output (or agg_out).  G_split (optional): also G in the form ghf_split_rows
 *                    (GHF_WLAYOUT_SPLIT2H) would cut from it, for the gradient passes that gather G.  workspace:
 *                    ghf_tail_bwd_workspace_floats(N, d) floats
 *   ghf_group_outer: C[g][i][o] (+)= sum_{e in gstart[g]..gend[g]} A[ia[e]][i] * B[ib[e]][o]; ia / ib NULL = e itself,
 *                    da == 0: A = 1 (C is [ngroups][1][db]).  dWm[r] = sum h_u^T G_v, dWs[r], db[r], and the Linear
 *                    layers' weight gradients.  Summation order fixed: reproducible.
 *   ghf_colsum     : out[o] (+)= sum_v X[v][o] * (mask ? mask[v][o] > 0 : 1); workspace: ghf_colsum_workspace_floats
 *   ghf_relu_mask  : out = X * (ref > 0);   ghf_transpose_batched: out[b][j][i] = in[b][i][j]
 *   ghf_weights_pack: [top[r] ; bottom[r]] ([d,d] natural each, optionally transposed, NULL = zeros) -> weight layout
 *                    (the gradient with respect to h is ghf_message_layer_fwd with GHF_FLAG_RAW_SUM on transposed
 *                    weights: sum_{e->v} G_v Ws[r]^T on the plan, sum_{e: src=u} G_v Wm[r]^T on the reversed plan). */
/* Edge ids grouped (stably) by relation: perm [E], goff [R+1] (goff[r] = first position of relation r). */
size_t ghf_group_workspace_bytes(int64_t E);
int ghf_group_edges(const int64_t* rel_id, int64_t E, int R, void* workspace, size_t workspace_bytes,
                    int64_t* perm, int64_t* goff, void* stream);
size_t ghf_tail_bwd_workspace_floats(int64_t N, int d);
int ghf_tail_bwd(const float* grad_out, const float* agg, const float* h, const float* ln_gamma, float ln_eps,
                 const int32_t* indeg, int64_t N, int d, float* dpre, float* G, void* G_split /* or NULL */,
                 float* dgamma_dbeta /* [2][d] */, float* workspace, const float* drop /* [N,d] or NULL */, void* stream);
size_t ghf_colsum_workspace_floats(int64_t N, int d);
int ghf_colsum(const float* X, const float* mask, int64_t N, int d, float* workspace, float* out, int accumulate, void* stream);
int ghf_relu_mask(const float* X, const float* ref, int64_t n, float* out, void* stream);
int ghf_group_outer(const float* A, const int64_t* ia, int da, const float* B, const int64_t* ib, int db,
                    const int64_t* gstart, const int64_t* gend, int ngroups, float* C, int accumulate, void* stream);
/* The three per-relation gradients of a layer in one pass over the edges (d = 64 or a multiple of 128, tile by tile;
 * ghf_edge_outer_supported):
 *   dW[r] = sum_{e in r} [h[src[e]] | h[dst[e]]]^T G[dst[e]]   ([R][2d][d]: dW_msg[r] stacked on dW_self[r]),
 *   db[r] = sum_{e in r} G[dst[e]]                             ([R][d])
 * src / dst [E]: the edges grouped by relation (ghf_group_edges); slice_tab [nslices][3] = (relation, first edge, end edge)
 * cuts every relation's range into slices, relations ascending, never across a relation; slice_off [R+1] = first slice of
 * each relation.  workspace: nslices * (2*D*D + D) + 64 floats, D = min(d, 128).  Fixed summation order.  d = 64: exact fp32
 * (v_mfma_f32_16x16x4_f32); d % 128 == 0: two fp16 pieces per operand with ONE power-of-two scale per tensor (the largest
 * magnitude of h resp. G — found by the call — lifted into [2^13, 2^14)), three v_mfma_f32_16x16x32_f16 per product, fp32
 * accumulation: 22 significand bits relative to each tensor's largest entries.  N <= 0 (or GHF_EDGE_OUTER=exact in the
 * environment) keeps the exact fp32 chain at every d: what the host mirror passes when a training step fell back to the exact
 * kernels (range guard).
 * order (or NULL = table order): a permutation of 0 .. nslices-1, the slice each workgroup of the launch takes, in launch
 * order.  It changes no bit of the result (a slice's partial sums land at the slice's index, the reduction runs in table order)
 * — only which slices are resident together: with destinations ascending inside a relation, the slices ordered by their
 * position WITHIN their relation (then by relation) make the workgroups in flight read the same band of destination rows of
 * h and G, once from HBM and then out of the Infinity Cache, instead of every relation sweeping all rows on its own. */
int ghf_edge_outer_supported(int d);
int ghf_edge_outer(const float* h, const float* G, const int64_t* src, const int64_t* dst, const int64_t* slice_tab,
                   const int64_t* slice_off, const int32_t* order /* [nslices] or NULL */, int64_t nslices, int R, int d,
                   int64_t N /* rows of h and G */, float* workspace, float* dW, float* db, void* stream);
/* The same for a caller that holds the split forms of h and G (ghf_split_rows / the h_split_out of a layer / ghf_tail_bwd's
 * G_split): h_rowscale, G_rowscale [N] = their row scales (the N floats behind the N split rows).  The one scale per tensor is
 * then read off those (2^13 max_row 2^-s(row) has the exponent of the tensor's largest magnitude) instead of by a pass over
 * both tensors: same pieces, same products, same bits as ghf_edge_outer with N > 0.  N > 0 required.
 * Range guard (round 4): the row scales also say how many NONZERO rows of a tensor lie 2^-15 or more below its largest
 * magnitude — rows the one scale leaves below 0.5, where two fp16 pieces keep fewer than 22 bits.  Some such rows are normal
 * (a training step's G: the nodes the loss does not touch) and add 2^-15 of what the others add; when they are at least 7/8
 * of either tensor's nonzero rows — a few outlier rows set the scale and the bulk of the tensor would be cut short — the call
 * runs on the exact fp32 chain instead (the bits of N <= 0), decided on the device: both kernels are enqueued and the
 * workgroups of one return at once.  GHF_EO_GUARD=0 in the environment removes it.  The four counters (far-down rows, nonzero
 * rows; of h, of G) stay in the workspace behind the partial sums and the two maxima:
 * ints at float offset nslices * (2*D*D + D) + 2. */
int ghf_edge_outer_scaled(const float* h, const float* G, const float* h_rowscale, const float* G_rowscale, const int64_t* src,
                          const int64_t* dst, const int64_t* slice_tab, const int64_t* slice_off, const int32_t* order,
                          int64_t nslices, int R, int d, int64_t N, float* workspace, float* dW, float* db, void* stream);
/* Elementwise pieces of the backward: out = X * exp(log_scale[0]) (log_scale on the device: the generator's learnable
 * scale, reference weight_generator.py:137-141); out = a + b (+ c when non-NULL); out[i][:] = g[i] * X[i][:] (the two
 * gradients of score_triple, reference hypergnn.py:304-318).  out may alias an input. */
int ghf_scale_exp(const float* X, int64_t n, const float* log_scale, float* out, void* stream);
int ghf_add3(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream);
int ghf_rowscale(const float* X, const float* g, int64_t n, int d, float* out, void* stream);
/* out[v][:] = sum_{e = off[v] .. off[v+1]-1} w[iw[e]] * X[ix[e]][:]  (out [nseg,d], X [nx,d], off [nseg+1]; e ascending: fixed
 * order; ix clamped into [0, nx)).  The gradient of the fused edge scores (ghf_score_pairs_fwd with index arrays): per node, the
 * pairs it takes part in (grouped by ghf_group_edges with the node ids as keys). */
int ghf_segment_axpy(const float* w, const int64_t* iw, const float* X, const int64_t* ix, const int64_t* off, int64_t nseg,
                     int64_t nx, int d, float* out, void* stream);
/* out[0] = sum_i X[i] Y[i] (fixed order); workspace: (n + 8191) / 8192 floats */
int ghf_dot(const float* X, const float* Y, int64_t n, float* workspace, float* out, void* stream);
/* Hidden activations of the weight generator's three heads (what its backward needs besides the outputs):
 * acts[head][layer][r][Hh], layer = 0 .. num_hidden-1 (post-ReLU).  Same head_params as ghf_weightgen_fwd. */
int ghf_weightgen_acts(const float* text_emb, const float* const* head_params, int R, int T, int Hh, int num_hidden,
                       float* acts, const float* hidden_drop /* or NULL */, void* stream);
/* Backward of ghf_weightgen_fwd with GHF_WLAYOUT_NATURAL outputs (reference models/weight_generator.py:96-143 under plain
 * autograd), all three heads and all their layers in three launches.  outs[k] / grads[k] (k = message matrices, self matrices,
 * biases): the forward's outputs [R, d_in*d_out] (k < 2) / [R, d_out] and dL/d of them; log_scales[k]: the heads' log-scales
 * (device, 1 float each); acts: ghf_weightgen_acts' result; log_keep: NULL, or log(1/(1-p)) (device) when acts are post-dropout.
 * Writes dparams (one buffer per entry of head_params: dL/dW [out, in] and dL/db [out] of every layer), dls[k] (1 float each,
 * device: dL/d log-scale) and d_text_emb [R, T] (or NULL).  Exact fp32, fixed summation order.  text_dim and hidden_dim up to
 * 256 (ghf_weightgen_bwd_supported; wider generators: the per-operation chain — ghf_dot, ghf_scale_exp, ghf_relu_mask,
 * ghf_group_outer, ghf_colsum).  workspace: ghf_weightgen_bwd_workspace_floats floats. */
int ghf_weightgen_bwd_supported(int T, int Hh, int num_hidden);
size_t ghf_weightgen_bwd_workspace_floats(int R, int T, int Hh, int num_hidden, int d_in, int d_out);
int ghf_weightgen_bwd(const float* text_emb, const float* const* head_params, const float* acts, const float* const* outs,
                      const float* const* grads, const float* const* log_scales, int R, int T, int Hh, int num_hidden, int d_in,
                      int d_out, const float* log_keep, float* const* dparams, float* const* dls, float* d_text_emb,
                      float* workspace, void* stream);
/* Backward of ghf_text_encode_fwd: given te = its output and dte = dL/dte, writes dL/dchar_emb [V,C], dL/dW [T,C],
 * dL/db [T] (overwritten, fixed summation order).  workspace: 2*U*C + U*T floats. */
int ghf_text_encode_bwd(const int32_t* ids, const int32_t* lens, int U, int Lmax, const float* char_emb, int V, int C,
                        const float* W, int T, const float* te, const float* dte, float* workspace,
                        float* d_char_emb, float* dW, float* db, void* stream);
int ghf_transpose_batched(const float* in, int batch, int rows, int cols, float* out, void* stream);
int ghf_weights_pack(const float* top, const float* bottom, int transpose, int R, int d, int wlayout, float* out, void* stream);

/* ---- link-prediction scores ------------------------------------------------------------
 * Replaces models/hypergnn.py:304-318 (score_triple) and the row gathers of its call sites (demo.py:90-94,
 * score_triple(embs[src], embs[dst])):  scores[i] = sum_k a[ia[i]][k] * b[ib[i]][k],  i < n.
 * a [rows_a, d], b [rows_b, d] fp32 row-major (may be the same matrix); ia / ib int64 [n] or NULL (= i).
 * An index outside its matrix gives NaN for that pair (the reference raises from ATen). */
int ghf_score_pairs_fwd(const float* a, const float* b, const int64_t* ia, const int64_t* ib,
                        int64_t rows_a, int64_t rows_b, int64_t n, int d, float* scores, void* stream);

/* ---- the sparse row exchange of the multi-GPU forward (no counterpart in the single-process reference) ----
 * packed[i] = rows[idx[i]] (row_bytes, a multiple of 16) followed by extra[idx[i]] (extra_bytes, a multiple of 4; extra may be
 * NULL with extra_bytes = 0), i < n: the listed rows of a [nrows, row_bytes] table (and of a second table indexed alike — the
 * split form's scales) as one contiguous message; ghf_rows_unpack writes a received message to the rows' places.  idx: int64,
 * distinct for unpack (entries outside [0, nrows) are skipped).  Bit-exact copies. */
int ghf_rows_pack(const void* rows, int64_t row_bytes, const void* extra, int64_t extra_bytes, const int64_t* idx, int64_t n,
                  int64_t nrows, void* packed, void* stream);
int ghf_rows_unpack(const void* packed, const int64_t* idx, int64_t n, int64_t nrows, void* rows, int64_t row_bytes, void* extra,
                    int64_t extra_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GHF_H */
