// message_bx.hip — K2+K3 for hidden 128 and 64 with the block sums in REGISTERS (reference: models/hypergnn.py:201-230, 288-296).
//
// message_hx.hip keeps a destination block's fp32 sums in LDS, which caps the block at 216 nodes; a (block, relation)
// chunk then has ~34 rows at C3 and the relation's 128 KB of weights is pulled through the CU's 64 B/clk vector-memory
// path once per 34 rows (54 GB through L2 per launch: profiles/r01_message_kernel_pmc.json).  The register file of a CU is
// 512 KB against 160 KB of LDS, and most of it was idle.  Here (d = 128; BxCfg<64> below for hidden 64):
//
//   * the block is BN = 384 nodes (chunks of ~60 rows: 1.8x the rows per weight fetch, 0.57x the chunks);
//   * waves 4-7 (HELPERS) own the block sums: helper wave hw holds the NPW = BN / 4 nodes NPW*hw .. NPW*hw + NPW - 1, lane l the
//     PL = d / 64 sum positions PL*l .. PL*l + PL - 1 of EVERY one of them, as plain VGPRs pinned to v64 .. v(63 + NPW*PL)
//     (192 registers per lane at BN = 384);
//   * waves 0-3 (CONSUMERS) run the contraction (SPLIT2H weights and row pieces, three v_mfma_f32_16x16x32_f16 per product,
//     B fragments in a register ring refilled straight from L2) and write a chunk's finished rows Y [rows][d] fp32 to an LDS
//     staging tile instead of scattering them;
//   * the helpers FOLD the staged rows into their registers ROW BY ROW (round 3): a chunk's rows are sorted by destination,
//     so the rows of a wave's nodes are one contiguous range; for each of them the wave reads the row — 4 PL bytes per lane,
//     one conflict-free 256 PL-byte access — and adds it to the registers of the row's node: the node is a run-time but
//     wave-uniform value, and gfx950's VGPR indexing mode (s_set_gpr_idx_on: the destination and the second source of a
//     VALU instruction relative to M0) turns that into two instructions per row and lane.  Rows in order, chunks in
//     order, no atomics: bitwise reproducible.  (Round 2 folded node by node — a table node -> row, every owner lane
//     reading its node's row or zeros: 48 ds_read_b128 + 96 packed adds per lane and chunk whatever the chunk held,
//     3,000 + 1,540 of the helpers' 8,700 cycles per chunk: tools/stamps_bx.py.)
//   * the helpers gather the A tiles with LDS-DMA (buffer_load_dwordx4 ... lds, per-lane source address, 1 KiB per wave
//     instruction, the XOR swizzle applied on the source side): no staging registers, no ds_write;
//   * the fused tail runs from LDS after the helpers have dumped their registers there (two halves of the block).
//
// LDS: P0[2] (source-row tiles), P1[2] (destination-row tiles), CR * 4d bytes each; four chunk descriptors ("meta": the
// rows' scales and node ids); a KiB for DMA pieces past a tile, a row of zeros, sixteen flag words.
// ONE workgroup barrier per chunk.  During chunk k (between barriers k and k + 1):
//   consumers: stage Y(k-1) into P1[(k-1)&1] (behind the barrier every consumer is through with that tile; flag "staged");
//              phase 0 (h_src x W_msg) from P0[k&1]; wait for the flag "destination rows of chunk k landed"; phase 1
//              (h_dst x W_self) from P1[k&1]; the accumulators keep Y(k) until the next barrier
//   helpers:   DMA P0[(k+1)&1] <- source rows of chunk k+1 (HBM: a whole chunk to land); counted vmcnt wait: the destination
//              rows of chunk k, requested at the end of chunk k-1, are in (flag "landed"); wait for "staged", fold Y(k-1) out
//              of P1[(k-1)&1]; once all four have folded (flag words), DMA that tile <- destination rows of chunk k+1 (L2; may
//              land after the barrier); the descriptor pipeline (chunk_tab entry k+5, edge words k+4, row
//              scales k+3, publish k+2); counted vmcnt wait: the source rows of chunk k+1 are in
// The indexing mode's switch (round 4: the cause of round 3's "unexplained hazard").  s_set_gpr_idx_on writes MODE.gpr_idx_en; a
// VALU instruction issued in the very next slot is not guaranteed to see it — the same class as the ISA's "s_setreg of MODE ->
// vector instruction" rule, which the assembler's and hipcc's hazard handling do not apply to this instruction.  With one helper
// wave per SIMD (hidden 128, one workgroup per CU) it never showed; at four waves per SIMD (hidden 64, two workgroups per CU) the
// first indexed add of a four-row block went wrong a few times per launch whenever the schedule put the helpers last at the
// chunk barrier (round 3's b64DEFER1: 40 of 40 launches, bxLATE0: 300 of 300) and the damage was not confined to the sums: whole
// 8-row DMA pieces were then gathered with a zeroed id register — from row 0 of h (tools/diag_rows.py).  Located by a
// single-pad bisect inside the asm block (profiles/r04_hazard_bisect.txt): wait states behind s_set_gpr_idx_on alone make both
// builds clean, wait states anywhere else do not; a fold without the mode (a select chain over the 48 sums) is clean too; M0
// traffic and mode switches without a VALU instruction inside are harmless; one wait state is enough (0 of 60), BX_IDX_WAIT
// puts four behind s_set_gpr_idx_on and four behind s_set_gpr_idx_off (M0's low byte is the index: a VALU instruction that still
// saw the mode after the restore would be displaced by the low byte of an LDS address).  The instruction pair alone, at 16 waves
// per CU without MFMA / LDS-DMA neighbours, does not reproduce it (tools/micro/gpr_idx_on_hazard.hip): the rule is kept on the
// strength of the in-kernel bisect.  RULE: never let the first VALU instruction of an indexing-mode block follow
// s_set_gpr_idx_on directly, nor a VALU instruction that must not be indexed follow s_set_gpr_idx_off directly.
// Template parameter SKIP: the instances for the backward's two gradient passes, whose weights have one zero half
// (GHF_FLAG_ZERO_SRC / GHF_FLAG_ZERO_DST): that half's gathers and products are compiled out.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

template <class T>
__device__ __forceinline__ const T* bx_at(const void* base, uint32_t byte_off) {   // uniform base + 32-bit byte offset
    return (const T*)((const char*)base + byte_off);
}

// LDS accesses of the helper waves go through inline asm: hipcc's s_waitcnt insertion orders every LDS read it can see
// behind all outstanding LDS-DMA of the wave (vmcnt(0)), which would turn the two-stage prefetch into a blocking load.
// An asm load's destination counts as written when the statement ends, so the wait must be in the same statement (or name
// the destinations, as the fold's counted waits do): otherwise the compiler may read the register before the data lands.
__device__ __forceinline__ int lds_ld_b32(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_ld_b32_x2(unsigned a0, unsigned a1, int& v0, int& v1) {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void lds_ld_b32_x6(const unsigned (&a)[6], int (&v)[6]) {
    asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %1, %7\n\tds_read_b32 %2, %8\n\tds_read_b32 %3, %9\n\tds_read_b32 %4, %10\n\t"
                 "ds_read_b32 %5, %11\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]) : "memory");
}
__device__ __forceinline__ void lds_ld_b32_x10(const unsigned (&a)[10], int (&v)[10]) {
    asm volatile("ds_read_b32 %0, %10\n\tds_read_b32 %1, %11\n\tds_read_b32 %2, %12\n\tds_read_b32 %3, %13\n\tds_read_b32 %4, %14\n\t"
                 "ds_read_b32 %5, %15\n\tds_read_b32 %6, %16\n\tds_read_b32 %7, %17\n\tds_read_b32 %8, %18\n\tds_read_b32 %9, %19\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]) : "memory");
}
__device__ __forceinline__ void lds_ld_b64_x2(unsigned a0, unsigned a1, f32x2& v0, f32x2& v1) {
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void lds_st_b32(unsigned addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st_b64(unsigned addr, f32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st_b128(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
#define BX_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// The lane id again, through an asm the optimiser cannot see through: what a stage derives from it (row numbers, swizzles,
// LDS addresses — dozens of values) is then recomputed per stage instead of being hoisted out of the chunk loop into
// registers that the block sums need.
__device__ __forceinline__ int opaque_lane(int lane) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return lane + z;
}

#ifndef GHF_BX_TAILNT
#define GHF_BX_TAILNT 1      // 1: the tail's loads of h and its stores are non-temporal (each line is touched once per launch: 2.94 -> 2.92 ms per C3 launch)
#endif
template <class T> __device__ __forceinline__ T bx_tail_ld(const T* p) {
#if defined(GHF_BX_TAILNT) && GHF_BX_TAILNT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T> __device__ __forceinline__ void bx_tail_st(T* p, T v) {
#if defined(GHF_BX_TAILNT) && GHF_BX_TAILNT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#ifndef GHF_BX_NPW
#define GHF_BX_NPW 96        // nodes per helper wave at d = 128 (BN = 4 NPW; two registers per node and lane)
#endif
// Compile-time ablations (GHF_VARIANT=bxexp<mask>, timing only, wrong results; tools/pmc_attr.sh): 1 no B refills, 2 no A-tile
// DMA, 4 no MFMAs, 8 no fold, 16 no staging writes, 32 no tail, 64 no descriptor pipeline (words / scales / publish / table),
// 128 one relation's weights for every chunk, 256 / 512 the destination / source rows' gathers without memory access
#ifndef GHF_BXEXP
#define GHF_BXEXP 0
#endif
#ifndef GHF_BX_LATE
#define GHF_BX_LATE 0         // 0 (round 4): every gathered tile is waited for (vmcnt(0)) BEFORE the chunk barrier and read behind it — the
                              // guide's order for LDS-DMA.  1 (rounds 2-3): the destination-row tile of chunk k+1 may land after the barrier,
                              // behind a flag the consumers wait for before phase 1, and the first phase's first fragments are prefetched:
                              // measured the same within the box noise in round 4 (2.90-2.93 vs 2.92-2.93 ms at C3, 0.165 vs 0.164 at C2), so
                              // the shortcut is off
#endif
#ifndef GHF_BX_DEFER
#define GHF_BX_DEFER 1        // 1: a chunk's rows are staged AFTER the next barrier (see the consumers' loop): no hand-shake among the
                              // consumers, 3.25 -> 3.13 ms at C3 together with GHF_BX_LATE and the helpers' batched descriptor reads
#endif
#ifndef GHF_BX_CR
#define GHF_BX_CR 76         // rows per chunk
#endif
#ifndef GHF_BX64_NPW
#define GHF_BX64_NPW 48      // hidden 64: nodes per helper wave (one register per node and lane): blocks of 192 nodes
#endif
#ifndef GHF_BX64_CR
#define GHF_BX64_CR 64       // hidden 64: rows per chunk (see BxCfg<64>)
#endif
#ifndef GHF_BX64_DEFER
#define GHF_BX64_DEFER 0     // hidden 64: GHF_BX_DEFER's choice for this size (measured slower there: 0.168 vs 0.164 ms per C2 launch)
#endif
#ifndef GHF_BX_SRCNT
#define GHF_BX_SRCNT 0       // 1: the source rows' gathers non-temporal.  Round 4, same box: 2.92 (1) vs 2.85 ms (0) per C3 launch — a source row
                             // is gathered ~10 times per layer by different blocks, and with the default policy the Infinity Cache serves part
                             // of those (the DESTINATION rows non-temporal: 3.12 ms — they live on the caches between their ~10 uses)
#endif
#ifndef GHF_B_AUX
#define GHF_B_AUX 0          // cache-policy bits of the consumers' weight-fragment loads (experiment: GHF_VARIANT=baux<n>)
#endif
#ifndef GHF_BX_IDXWAIT
#define GHF_BX_IDXWAIT 4     // wait states behind s_set_gpr_idx_on / _off in the fold (header: "The indexing mode's switch"); 0 = round 3
#endif
#if GHF_BX_IDXWAIT == 0
#define BX_IDX_WAIT ""
#elif GHF_BX_IDXWAIT == 1
#define BX_IDX_WAIT "s_nop 0\n\t"
#elif GHF_BX_IDXWAIT == 2
#define BX_IDX_WAIT "s_nop 1\n\t"
#elif GHF_BX_IDXWAIT == 4
#define BX_IDX_WAIT "s_nop 3\n\t"
#elif GHF_BX_IDXWAIT == 8
#define BX_IDX_WAIT "s_nop 7\n\t"
#else
#define BX_IDX_WAIT "s_nop 7\n\ts_nop 7\n\t"
#endif
// measured and fixed (round 2-3 A/B records in DESIGN_HISTORY.md): A fragments two positions ahead; weight refills behind their
// k-step's MFMAs (scheduling barriers); staged rows drained before their flag; the first phase's first fragments requested
// behind the barrier; non-temporal source gathers; ZERO_SRC as a first phase; the tail's batches of three four-row groups
constexpr int GHF_BX_AD = 2, GHF_BX_TGB = 3;

// Diagnostic build only (-DGHF_STAMPS): per-wave s_memtime totals per segment.
// consumers: 0 barrier wait, 1 phase-0 stage, 2 phase-1 stage, 3 staging writes
// helpers:   0 barrier wait, 1 DMA issue, 2 fold + table clear, 3 wait for the P1 pieces, 4 descriptor work, 6 epilogue + tail
#ifdef GHF_STAMPS
__device__ unsigned long long ghf_bx_stamp_buf[8192 * 8 * 8];
__device__ unsigned long long ghf_bx_life_buf[8192 * 3];     // per workgroup: first stamp, last stamp, (XCC id << 32) | HW_ID
#define BX_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0, st_first = 0
#define BX_STAMP(i)                                                                            \
    do {                                                                                       \
        unsigned long long _t;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((i) >= 0) st_acc[(i) < 0 ? 0 : (i)] += _t - st_last;                               \
        else st_first = _t;                                                                    \
        st_last = _t;                                                                          \
    } while (0)
#define BX_STAMP_FLUSH()                                                                       \
    do {                                                                                       \
        if (lane == 0 && blockIdx.x < 8192)                                                    \
            for (int i = 0; i < 8; ++i) ghf_bx_stamp_buf[((size_t)blockIdx.x * 8 + w) * 8 + i] = st_acc[i]; \
        if (lane == 0 && w == 0 && blockIdx.x < 8192) {  /* the workgroup's life and place: tools/stamps_gap.py */ \
            unsigned _hw, _xcc;                                                                \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(_hw), "=s"(_xcc)); \
            ghf_bx_life_buf[(size_t)blockIdx.x * 3] = st_first;                                \
            ghf_bx_life_buf[(size_t)blockIdx.x * 3 + 1] = st_last;                             \
            ghf_bx_life_buf[(size_t)blockIdx.x * 3 + 2] = ((unsigned long long)_xcc << 32) | _hw; \
        }                                                                                      \
    } while (0)
#else
#define BX_STAMP_DECL
#define BX_STAMP(i)
#define BX_STAMP_FLUSH()
#endif


template <int D> struct BxCfg;
template <> struct BxCfg<128> {
    static constexpr int NPW = GHF_BX_NPW, CR = GHF_BX_CR;
    static constexpr int BN = 4 * NPW;             // four helper waves
    static constexpr int MTC = (CR + 15) / 16;     // row tiles per chunk
    static constexpr bool YT = false;               // (a fifth tile for the staged rows was measured and dropped: round 2)
    static constexpr size_t LDS = (size_t)(YT ? 5 : 4) * 2 * CR * 256 + 4 * (4 * 16 * MTC + 4) * 4 + 1024 + 512 + 64;
};
// hidden 64: a chunk is [rows, 128] x [128, 64] — a quarter of the matrix work per row, so the fixed cost per chunk (~7,000
// cycles: barrier, hand-shakes, the descriptor pipeline, two DMA round trips) decides, and two workgroups per CU hide it
// behind each other: 71 KB of LDS and <= 128 registers per workgroup.  Blocks of 192 nodes: at C2's 32 relations a block and
// relation hold ~60 rows — one chunk of <= 64 rows, not a full one and a remainder — and C2's 100 k nodes make 521 workgroups
// for the 512 slots.  Round 3: 0.220 -> 0.155 ms per C2 launch against 256-node blocks, 112-row chunks, one workgroup per CU.
// GHF_BX64_DEFER = 0 here: the deferred staging is slower at this size (0.168 vs 0.164 ms per C2 launch, same box).  (Round 3
// shipped without it because that build was not bitwise reproducible: the indexing mode's switch — see the header.)
template <> struct BxCfg<64> {
    static constexpr int NPW = GHF_BX64_NPW, CR = GHF_BX64_CR;
    static constexpr int BN = 4 * NPW;
    static constexpr int MTC = (CR + 15) / 16;
    static constexpr bool YT = false;
    static constexpr size_t LDS = (size_t)4 * 2 * CR * 128 + 4 * (4 * 16 * MTC + 4) * 4 + 1024 + 512 + 64;
};

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// The helpers' block sums: tuples of 32 registers (16 at hidden 64) pinned to v64 .. — every asm that touches them names them
// with these constraints, so the register allocator keeps them there and the indexed adds can name v64 + index.
#define BX_PIN_128(s) "+{v[64:95]}"(s[0]), "+{v[96:127]}"(s[1]), "+{v[128:159]}"(s[2]), "+{v[160:191]}"(s[3]), "+{v[192:223]}"(s[4]), "+{v[224:255]}"(s[5])
#if GHF_BX64_NPW == 64
#define BX_PIN_64(s) "+{v[64:79]}"(s[0]), "+{v[80:95]}"(s[1]), "+{v[96:111]}"(s[2]), "+{v[112:127]}"(s[3])
#elif GHF_BX64_NPW == 48
#define BX_PIN_64(s) "+{v[64:79]}"(s[0]), "+{v[80:95]}"(s[1]), "+{v[96:111]}"(s[2])
#elif GHF_BX64_NPW == 32
#define BX_PIN_64(s) "+{v[64:79]}"(s[0]), "+{v[80:95]}"(s[1])
#else
#error "GHF_BX64_NPW: 32, 48 or 64"
#endif

struct BxChunk { int r; int e0; int rows; };

// SKIP: bit 0 = the source half of the weights is zero (GHF_FLAG_ZERO_SRC), bit 1 = the destination half: that half's gathers
// and products are compiled out (a run-time switch cost the consumers' loop a spilled weight fragment)
template <int D, int SKIP>
__global__ __launch_bounds__(512, D == 64 ? 4 : 2) void message_bx_kernel(   // (hidden 64: <= 128 registers, two workgroups per CU)
    const float* __restrict__ h, const void* __restrict__ h_split, int64_t N, const uint32_t* __restrict__ sorted_key,
    const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ chunk_tab,
    const int32_t* __restrict__ item_tab, int64_t item0, float* __restrict__ partial,
    const int32_t* __restrict__ indeg, int R,
    const void* __restrict__ Wsplit, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int64_t row0, int64_t row_end, float* __restrict__ h_out, void* __restrict__ h_split_out, int no_tail,
    int32_t* __restrict__ range_flag, float* __restrict__ agg_out) {
    using C = BxCfg<D>;
    // ZERO_SRC (SKIP = 1, the backward's pass over the forward plan) runs as the one-phase kernel of ZERO_DST does — its rows
    // gathered a whole chunk ahead into the P0 tiles, no hand-shake for a late tile — with the destination ids, the
    // destination half of the weights and the destination rows' scales in that phase: as a second phase without a first one
    // its gathers were issued behind the fold and waited for at the next chunk's start (2.41 -> 2.34 ms per C3 launch, same box,
    // GHF_VARIANT=bxSWAP10 for the old form; the 4.8 ms of round 2's training profile was this launch beside the weight
    // gradients' kernel on a second stream)
    constexpr bool SWAP1 = SKIP == 1;
    constexpr int skip = SWAP1 ? 2 : SKIP;
    constexpr int P0_IDS = SWAP1 ? 3 : 2;          // which ids the P0 tiles' rows follow (2: source, 3: destination)
    constexpr int P0_HALF = SWAP1 ? 1 : 0;         // the half of the weights (and the row scales) of the P0 phase
    constexpr bool P0_NT = GHF_BX_SRCNT && !SWAP1;   // (a destination row is gathered once per in-edge: default cache policy)
    constexpr int BN = C::BN, MTC = C::MTC, CR = C::CR, NPW = C::NPW;
    constexpr bool DEFER = D == 64 ? (GHF_BX64_DEFER != 0) : (GHF_BX_DEFER != 0);
    constexpr int NWV = 8, TW = 4;            // waves per workgroup, per role
    constexpr int KS = D / 32;                // k-steps of 32 per phase
    constexpr int NKS = 2 * KS;
    constexpr int NT = D / 16;                // 16-column fragments of the output
    constexpr int NTW = NT / TW;              // fragments per consumer wave (2)
    constexpr int NPL = 2;                    // pieces (hi, lo)
    constexpr int ROWB = D * 2;               // bytes per row of one fp16 plane of an A tile
    constexpr int PLANE = CR * ROWB;          // bytes per plane (a multiple of 1 KiB: CR % 4 == 0)
    constexpr int TILE = NPL * PLANE;         // one A tile; also the size of the staging tile Y [CR][D] fp32
    constexpr int HROW = NPL * D * 2;         // bytes per node of h_split
    constexpr int CRP = 16 * MTC;             // rows an A tile is read as (the last tile's dead rows are never used)
    constexpr int MSTR = 4 * CRP + 4;         // words per chunk descriptor: sc_u, sc_v, src id, dst id [CRP each], rows
    constexpr int PL = D / 64;                // sum positions (registers) per node and helper lane
    constexpr int TUP = D == 128 ? 32 : 16;   // registers per tuple of sums
    constexpr int NSV = NPW * PL / TUP;       // tuples that hold a helper lane's sums
    constexpr int RPP = 1024 / ROWB;          // rows per 1 KiB DMA piece of a plane (4 at d = 128, 8 at d = 64)
    constexpr int LPR = 64 / RPP;             // lanes (16-byte granules) per row of a piece
    constexpr int RBN = CR / RPP;             // pieces per plane
    constexpr int CPL = D / 16;               // tail: columns per lane (16 lanes per row)
    constexpr int RBW = (RBN + TW - 1) / TW;  // pieces per helper wave and plane
    constexpr int RPH = (CR + TW - 1) / TW;   // rows of a descriptor per helper wave
    static_assert((NTW == 2 || NTW == 1) && CR % RPP == 0 && RPH <= 64 && NPW * TW == BN && (NPW * PL) % TUP == 0 && NPW * PL <= 192 &&
                      NPW % 2 == 0 && (BN / 2) % 4 == 0 && CR <= 128 && NSV <= 6 && (D == 128 || D == 64), "bad config");
    // the XOR key of a row's 16-byte granules in an A tile: 16 granules per row at d = 128 (key = row mod 16); 8 at d = 64, where
    // two rows share a 256-byte bank line, so the key is (row / 2) mod 8 — either way the 16 rows of a fragment read hit every
    // bank once
    auto akey = [](int row) -> int { return D == 128 ? (row & 15) : ((row >> 1) & 7); };
    // P0[2] source-row tiles, P1[2] destination-row tiles; a chunk's staged rows Y overwrite its own P1 tile — or, where
    // five tiles fit (YT: rows per chunk <= 60), have a tile of their own: then neither DMA waits for the fold
    constexpr bool YT = C::YT;
    constexpr unsigned P0_OFF = 0, P1_OFF = 2 * TILE, Y_OFF = 4 * TILE, META_OFF = (YT ? 5 : 4) * TILE,
                       DUMMY_OFF = META_OFF + 4 * MSTR * 4, ZERO_OFF = DUMMY_OFF + 1024,       // ZERO: a staged row of zeros (512 bytes)
                       FLAG_OFF = ZERO_OFF + 512;         // FLAG: 4 helper words (chunks folded), 4 consumer words (chunks whose tiles are read),
                                                          // 4 helper words (chunks whose destination-row tile has landed), 4 consumer
                                                          // words (the last chunk's hand-shake when staging is deferred)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;       // LDS byte address of smem (0 unless static LDS exists)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = w >= TW;
    const int tw = w & 3;                      // consumer: column group; helper: hw
    const int q = lane >> 4, c16 = lane & 15;
    const i32x4 item = *(const i32x4*)(item_tab + 4 * (size_t)(item0 + blockIdx.x));
    const int64_t blk = __builtin_amdgcn_readfirstlane(item[0]);
    const int slot = __builtin_amdgcn_readfirstlane(item[3]);
    const int64_t node0 = blk * BN;
    const int nrows = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const uint32_t seg0 = (uint32_t)(blk * R);
    const int c_begin = __builtin_amdgcn_readfirstlane(item[1]);
    const int c_end = __builtin_amdgcn_readfirstlane(item[2]);
    const int nchunks = c_end - c_begin;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));    // opaque 0: keeps the descriptor loads on the vector path
    BX_STAMP_DECL;
    BX_STAMP(-1);

    auto load_desc = [&](int k) -> i32x2 {                              // past the end: the item's first chunk (ignored)
        const int c = c_begin + (k < nchunks ? k : 0) + vzero;
        return *bx_at<i32x2>(chunk_tab, (uint32_t)c * 8u);
    };
    auto decode = [&](i32x2 d) -> BxChunk {
        const int w0 = __builtin_amdgcn_readfirstlane(d[0]), w1 = __builtin_amdgcn_readfirstlane(d[1]);
        return BxChunk{w1 >> 8, w0, w1 & 127};
    };
    auto meta_off = [&](int j) -> unsigned { return META_OFF + (unsigned)(j & 3) * (MSTR * 4); };
    const uint32_t hsc_off = (uint32_t)((uint64_t)N * HROW);            // the row scales follow the N split rows

    // ---- fused tail from LDS (both roles; the helpers dump their registers first, NPW / 2 nodes per wave at a time) ----
    // dump row v = (NPW/2) * hw + i  <->  node NPW * hw + (NPW/2) * half + i
    constexpr int HN = BN / 2, HPW = NPW / 2;
    // 16 lanes per row (a wave works on four rows at a time), CPL = d / 16 adjacent columns per lane: every load and store is
    // 16 bytes per lane, and a row reduction is four DPP steps inside its 16-lane row — for four rows at once.  (One wave
    // per row, two columns per lane, took 1,800 cycles per row: ~110 dependent instructions, 2- and 4-byte stores.)
    auto tail_half = [&](int half, auto gb_c) __attribute__((always_inline)) {          // gb_c: four-row groups in flight per wave
        if (GHF_BXEXP & 32) return;
        const float* acc_lds = (const float*)smem;       // dump rows, natural column order
        // a lane's CPL columns: 4 (lane mod 16) + 64 i + (0..3), i < NV — every 16-byte load / store of a row's 16 lanes is one
        // contiguous 256 bytes (whole 128-byte lines per instruction); adjacent columns per lane left half of each line to the
        // lane's next instruction
        constexpr int CS = 64;
        const int sub = lane >> 4, c0 = 4 * (lane & 15);
        constexpr int NV = CPL / 4;                      // 16-byte pieces per lane and row
        auto node_of = [&](int v) -> int { return (v / HPW) * NPW + half * HPW + (v % HPW); };   // block-local node of dump row v
        auto row_sum = [&](float v) -> float {           // over the 16 lanes of a row, result in each of them
            v += dpp_take<0xB1, 0xF>(v);
            v += dpp_take<0x4E, 0xF>(v);
            v += dpp_take<0x141, 0xF>(v);
            v += dpp_take<0x140, 0xF>(v);
            return v;
        };
        auto row_max = [&](float v) -> float {           // non-negative values
            v = fmaxf(v, dpp_take<0xB1, 0xF>(v));
            v = fmaxf(v, dpp_take<0x4E, 0xF>(v));
            v = fmaxf(v, dpp_take<0x141, 0xF>(v));
            v = fmaxf(v, dpp_take<0x140, 0xF>(v));
            return v;
        };
        constexpr int NG = HN / 4;                        // four-row groups of this half
        if (slot >= 0) {                                  // one item of a split block: raw sums to my slot
            float* __restrict__ ps = partial + (size_t)slot * BN * D;
            for (int g = w; g < NG; g += NWV) {
                const int v = 4 * g + sub;
                float* __restrict__ o = ps + (size_t)node_of(v) * D + c0;
#pragma unroll
                for (int i = 0; i < NV; ++i) *(f32x4*)(o + CS * i) = *(const f32x4*)(acc_lds + v * D + c0 + CS * i);
            }
            return;
        }
        float gm[CPL], bt[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            gm[c] = no_tail ? 1.f : gamma[c0 + CS * (c >> 2) + (c & 3)];
            bt[c] = no_tail ? 0.f : beta[c0 + CS * (c >> 2) + (c & 3)];
        }
        constexpr int GB = decltype(gb_c)::value;
        for (int g0 = w; g0 < NG; g0 += NWV * GB) {
            f32x4 x[GB][NV];
            float inv[GB];
            int deg[GB];
            // every load of the batch first, their uses behind a scheduling barrier: hipcc otherwise turns each in-degree into
            // its reciprocal at once — a wait for that load and, the counter being in order, for every row load before it:
            // three latencies per batch instead of one
#pragma unroll
            for (int gb = 0; gb < GB; ++gb) {
                const int g = g0 + gb * NWV;
                const int nl = node_of(4 * (g < NG ? g : NG - 1) + sub);
                const int64_t node = node0 + (nl < nrows ? nl : nrows - 1);
                deg[gb] = indeg[node];
                const float* __restrict__ hp = h + (size_t)node * D + c0;
#pragma unroll
                for (int i = 0; i < NV; ++i)              // (GHF_FLAG_ADD_H: the residual operand also without the tail)
                    x[gb][i] = (no_tail && !(no_tail & GHF_FLAG_ADD_H)) ? (f32x4){0.f, 0.f, 0.f, 0.f} : bx_tail_ld((const f32x4*)(hp + CS * i));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gb = 0; gb < GB; ++gb) inv[gb] = (no_tail & GHF_FLAG_RAW_SUM) ? 1.0f : 1.0f / (float)(deg[gb] > 1 ? deg[gb] : 1);
#pragma unroll
            for (int gb = 0; gb < GB; ++gb) {
                const int g = g0 + gb * NWV;
                const int v = 4 * (g < NG ? g : NG - 1) + sub;
                const int nl = node_of(v);
                const bool live = g < NG && nl < nrows;
                f32x4 a[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) a[i] = *(const f32x4*)(acc_lds + v * D + c0 + CS * i);
                if (agg_out && live) {                   // side output: the mean before the tail (what the backward keeps)
                    float* __restrict__ o = agg_out + (size_t)(node0 + nl) * D + c0;
#pragma unroll
                    for (int i = 0; i < NV; ++i) bx_tail_st((f32x4*)(o + CS * i), a[i] * inv[gb]);
                }
                float y[CPL];
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const float av = a[c >> 2][c & 3] * inv[gb];
                    y[c] = no_tail ? av + x[gb][c >> 2][c & 3] : fmaxf(av + x[gb][c >> 2][c & 3], 0.f);      // (x = 0 without ADD_H)
                    s += y[c];
                }
                if (!no_tail) {
                    const float mean = row_sum(s) * (1.0f / D);
                    float var = 0.f;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) { const float t = y[c] - mean; var += t * t; }
                    const float rstd = 1.0f / sqrtf(row_sum(var) * (1.0f / D) + eps);
#pragma unroll
                    for (int c = 0; c < CPL; ++c) y[c] = (y[c] - mean) * rstd * gm[c] + bt[c];
                }
                float up = 1.f;
                if (h_split_out) {                       // the same row cut into fp16 pieces, for the next layer's gathers
                    float mx = 0.f;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) mx = fmaxf(mx, fabsf(y[c]));
                    const int sh = split2h_shift(row_max(mx));
                    up = pow2f(sh);
                    if ((lane & 15) == 0 && live) *(float*)((char*)h_split_out + (size_t)hsc_off + (size_t)(node0 + nl) * 4) = pow2f(-sh);
                }
                if (live) {
                    float* __restrict__ o = h_out + (size_t)(node0 + nl) * D + c0;
#pragma unroll
                    for (int i = 0; i < NV; ++i) bx_tail_st((f32x4*)(o + CS * i), (f32x4){y[4 * i], y[4 * i + 1], y[4 * i + 2], y[4 * i + 3]});
                    if (h_split_out) {
                        _Float16* __restrict__ sp = (_Float16*)h_split_out + (size_t)(node0 + nl) * (NPL * D) + c0;
                        _Float16 hi[CPL], lo[CPL];
#pragma unroll
                        for (int c = 0; c < CPL; ++c) split2h(y[c] * up, hi[c], lo[c]);
                        if constexpr (CPL == 8 && CS == 4) {
                            *(f16x8*)sp = (f16x8){hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
                            *(f16x8*)(sp + D) = (f16x8){lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
                        } else if constexpr (CPL == 8) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                bx_tail_st((f16x4*)(sp + CS * i), (f16x4){hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]});
                                bx_tail_st((f16x4*)(sp + D + CS * i), (f16x4){lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]});
                            }
                        } else {
                            bx_tail_st((f16x4*)sp, (f16x4){hi[0], hi[1], hi[2], hi[3]});
                            bx_tail_st((f16x4*)(sp + D), (f16x4){lo[0], lo[1], lo[2], lo[3]});
                        }
                    }
                }
                if (h_split_out) {                       // range guard (common.h): rows with many entries far below their largest
                    int tiny = 0, nz = 0;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) { tiny += range_tiny(y[c] * up); nz += y[c] != 0.f; }
                    if (__ballot(tiny != 0)) {           // (rare)
                        tiny = (int)row_sum((float)tiny);
                        nz = (int)row_sum((float)nz);
                        if ((lane & 15) == 0 && live) range_raise(range_flag, GHF_RANGE_ROWS, tiny, nz);
                    }
                }
            }
        }
    };

    if (helper) {
        typename std::conditional<D == 128, f32x32, f32x16>::type sm[NSV];   // the block sums (see header): register PL*n + e = position
#pragma unroll                                                          // PL*lane + e of this wave's node n
        for (int k = 0; k < NSV; ++k)
#pragma unroll
            for (int i = 0; i < TUP; ++i) sm[k][i] = 0.f;
        if constexpr (D == 128) asm volatile("" : BX_PIN_128(sm)); else asm volatile("" : BX_PIN_64(sm));
        // ================================================ HELPERS ================================================
        const int hw = tw;
        const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)h_split, 0, (int)hsc_off, 0x00020000);
        // ---- a chunk's descriptor: lane l < RPH handles row RPH*hw + l ---------------------------------------------
        struct Words { int src; int key; };
        struct Scales { float u; float v; };
        auto load_words = [&](const BxChunk& c, int ln) -> Words {
            const int mrow = hw * RPH + (ln < RPH ? ln : RPH - 1);
            const int rc = mrow < c.rows ? mrow : c.rows - 1;               // rows >= 1; pad rows repeat the last row
            const uint32_t eo = (uint32_t)(c.e0 + rc) * 4u;
            return Words{*bx_at<int>(sorted_src, eo), *bx_at<int>(sorted_key, eo)};
        };
        auto dst_of = [&](const BxChunk& c, const Words& wd) -> uint32_t {
            const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
            return (uint32_t)node0 + ((uint32_t)wd.key - kbase);
        };
        auto load_scales = [&](const BxChunk& c, const Words& wd) -> Scales {
            const uint32_t nu = (uint32_t)(wd.src & SRC_MASK), nv = dst_of(c, wd);
            return Scales{*bx_at<float>(h_split, hsc_off + nu * 4u), *bx_at<float>(h_split, hsc_off + nv * 4u)};
        };
        auto publish = [&](int j, const BxChunk& c, const Words& wd, const Scales& sc, int ln) {
            const unsigned m = lds0 + meta_off(j);
            const int mrow = hw * RPH + (ln < RPH ? ln : RPH - 1);
            if (ln < RPH && mrow < CRP) {
                lds_st_b32(m + 4 * mrow, __float_as_int(sc.u));
                lds_st_b32(m + 4 * (CRP + mrow), __float_as_int(sc.v));
                lds_st_b32(m + 4 * (2 * CRP + mrow), wd.src & SRC_MASK);
                lds_st_b32(m + 4 * (3 * CRP + mrow), (int)dst_of(c, wd));
            }
            if (hw == 0 && ln == 0) lds_st_b32(m + 4 * (4 * CRP), c.rows);
        };
        // ---- LDS-DMA gather of one A tile: piece = RPP consecutive rows of one plane (1 KiB), LPR lanes per row; the
        // granules of a row are XOR-swizzled by akey(row), as the consumers' fragment reads expect ----
        // (the ids of a tile's rows are read apart from the issue: one LDS round trip — ~250 cycles here — that the loop below
        // places behind other waits instead of in front of every tile's DMA instructions)
        static_assert(RBW <= 6, "ids of a tile: one batched read");
        auto id_addr = [&](int j, int which /*2: src ids, 3: dst ids*/, int i, int lane) -> unsigned {
            return lds0 + meta_off(j) + 4 * (which * CRP) + 4 * (RPP * (hw + TW * (i < RBW ? i : 0)) + lane / LPR);
        };
        auto dma_ids = [&](int j, int which, int lane, int (&id)[6]) {
            unsigned ia[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) ia[i] = id_addr(j, which, i, lane);
            lds_ld_b32_x6(ia, id);
        };
        auto dma_issue = [&](unsigned tile_off, int rows, bool nt, int lane, const int (&id)[6]) {
            if (GHF_BXEXP & 2) return;
#pragma unroll
            for (int i_ = 0; i_ < RBW; ++i_) {
                const int i = i_;
                const int rb = hw + TW * i, row = RPP * rb + lane / LPR;
                // (a piece without live rows is issued too when GHF_BX_LATE: every lane past the buffer — zeros, no memory
                // access — so that the number of DMA instructions per tile is a constant the counted waits can name)
                if (!GHF_BX_LATE && RPP * rb >= rows) continue;
                const int g = (lane & (LPR - 1)) ^ akey(row);
                // dead rows: an offset past the end of the buffer (zeros, no memory access)
                // (GHF_BXEXP 256 / 512: the destination / source tile's loads all past the buffer — same instructions, no memory access)
                const bool no_mem = ((GHF_BXEXP & 256) && tile_off >= P1_OFF) || ((GHF_BXEXP & 512) && tile_off < P1_OFF);
                const int voff = (row < rows && !no_mem) ? (int)((uint32_t)id[i] * (uint32_t)HROW) + (g << 4) : (int)0xFFFFF000u;
                // a piece past the tile (RBN not a multiple of 4 waves) lands in a scratch KiB
                const unsigned dst = rb < RBN ? tile_off + (unsigned)rb * 1024u : DUMMY_OFF;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const unsigned d = __builtin_amdgcn_readfirstlane(dst + (rb < RBN ? pl * PLANE : 0));
                    if (nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsH, (lptr_t)(smem + d), 16, voff + pl * ROWB, 0, 0, 2);
                    else    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsH, (lptr_t)(smem + d), 16, voff + pl * ROWB, 0, 0, 0);
                }
            }
        };
        auto dma_tile = [&](unsigned tile_off, int j, int which, const BxChunk& c, bool nt, int lane) {
            int id[6];
            dma_ids(j, which, lane, id);
            dma_issue(tile_off, c.rows, nt, lane, id);
        };
        // ---- fold chunk j's staged rows into the registers, row by row -------------------------------------------
        // Y [row][position], the two 32-position halves of a 64-position group swapped when (row >> 2) & 1 (the consumers'
        // ds_write_b64 of four row groups then touch every bank exactly twice).  The chunk's rows are sorted by destination:
        // the rows of this wave's nodes are rows [ra, ra + cnt).  Four rows per step: their reads (one 4 PL-byte access per
        // lane and row), then per row the node's registers += the row, through the VGPR indexing mode: M0[7:0] = PL * node,
        // destination and second source of the adds relative to it (mode 0xa), base register v64.  M0 also holds the LDS
        // base of the wave's LDS-DMA instructions: saved and restored around the mode.  Steps past cnt read the row of
        // zeros into node 0 (no branch inside a step).
        struct FoldPlan { int n0, n1, ra, cnt; };
        // which of chunk j's rows are mine (one LDS round trip; taken BEFORE the wait for the staged rows: the descriptor has
        // long been published)
        auto fold_plan = [&](int j, int rows, int lane) __attribute__((always_inline)) -> FoldPlan {
            const unsigned dd = lds0 + meta_off(j) + 4 * (3 * CRP);
            int d0, d1;
            lds_ld_b32_x2(dd + 4 * lane, dd + 4 * (lane + 64 < CRP ? lane + 64 : CRP - 1), d0, d1);
            const int base = (int)node0 + hw * NPW;
            const int n0 = d0 - base, n1 = d1 - base;                    // wave-local node of rows lane, 64 + lane
            const unsigned long long q0 = __ballot(lane < rows && (unsigned)n0 < (unsigned)NPW);
            const unsigned long long q1 = __ballot(lane + 64 < rows && (unsigned)n1 < (unsigned)NPW);
            const int cnt = __builtin_popcountll(q0) + __builtin_popcountll(q1);
            const int ra = q0 ? (int)__builtin_ctzll(q0) : 64 + (q1 ? (int)__builtin_ctzll(q1) : 0);
            return FoldPlan{n0, n1, ra, cnt};
        };
        auto fold_rows = [&](int j, const FoldPlan& fp, int lane) __attribute__((always_inline)) {
            const unsigned Y = lds0 + (YT ? Y_OFF : P1_OFF + (unsigned)(j & 1) * TILE);
            const int ra = fp.ra, cnt = fp.cnt;
            const unsigned zrow = lds0 + ZERO_OFF + (unsigned)(PL * 4 * lane), lb = (unsigned)(PL * 4 * lane);
            typedef typename std::conditional<D == 128, f32x2, float>::type yv_t;
            // Eight rows per step, read as two batches of four: the second lands while the first is added (an LDS round trip is
            // ~250 cycles here, an indexed add ~12; more rows in flight and the register allocator starts to move the pinned
            // sums around).  Rows past the end read the row of zeros into node 0.
            constexpr int FB = 4;
            auto issue = [&](yv_t (&y)[FB], int r0) __attribute__((always_inline)) {
                unsigned a[FB];
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    const int rr = ra + r0 + i;
                    a[i] = r0 + i < cnt ? Y + (unsigned)rr * (D * 4) + (lb ^ (unsigned)(((rr >> 2) & 1) << 7)) : zrow;
                }
                if constexpr (D == 128)
                    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7"
                                 : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
                else
                    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7"
                                 : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
            };
            // the wait names the batch's registers: nothing may read them above it.  behind: the next batch is in flight
            auto add_rows = [&](yv_t (&y)[FB], int r0, bool behind) __attribute__((always_inline)) {
                if (behind) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3])::"memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3])::"memory");
                int ix[FB];
#pragma unroll
                for (int i = 0; i < FB; ++i) {                           // the row's node: lane (row mod 64) of n0 or n1 (both read: no branch)
                    const int rr = ra + r0 + i;
                    const int a = __builtin_amdgcn_readlane(fp.n0, rr & 63), b = __builtin_amdgcn_readlane(fp.n1, rr & 63);
                    ix[i] = r0 + i < cnt ? PL * (rr < 64 ? a : b) : 0;
                }
                int keep;
#define BX_ROW128(i) "s_set_gpr_idx_idx %[i" #i "]\n\tv_pk_add_f32 v[64:65], %[y" #i "], v[64:65]\n\t"
#define BX_ROW64(i) "s_set_gpr_idx_idx %[i" #i "]\n\tv_add_f32 v64, %[y" #i "], v64\n\t"
#define BX_ROW_OPS [i0] "s"(ix[0]), [i1] "s"(ix[1]), [i2] "s"(ix[2]), [i3] "s"(ix[3]), [y0] "v"(y[0]), [y1] "v"(y[1]), [y2] "v"(y[2]), [y3] "v"(y[3])
                // BX_IDX_WAIT: wait states behind BOTH mode switches.  Measured (round 4, tools/hazard_r4*.sh): without them behind
                // s_set_gpr_idx_on the first add of a block went wrong a few times per launch at four waves per SIMD (hidden 64,
                // two workgroups per CU) — see the kernel header, "The indexing mode's switch".
                if constexpr (D == 128)
                    asm volatile("s_mov_b32 %[kp], m0\n\ts_set_gpr_idx_on %[i0], 0xa\n\t" BX_IDX_WAIT "v_pk_add_f32 v[64:65], %[y0], v[64:65]\n\t"
                                 BX_ROW128(1) BX_ROW128(2) BX_ROW128(3)
                                 "s_set_gpr_idx_off\n\t" BX_IDX_WAIT "s_mov_b32 m0, %[kp]"
                                 : BX_PIN_128(sm), [kp] "=&s"(keep) : BX_ROW_OPS);
                else
                    asm volatile("s_mov_b32 %[kp], m0\n\ts_set_gpr_idx_on %[i0], 0xa\n\t" BX_IDX_WAIT "v_add_f32 v64, %[y0], v64\n\t"
                                 BX_ROW64(1) BX_ROW64(2) BX_ROW64(3)
                                 "s_set_gpr_idx_off\n\t" BX_IDX_WAIT "s_mov_b32 m0, %[kp]"
                                 : BX_PIN_64(sm), [kp] "=&s"(keep) : BX_ROW_OPS);
#undef BX_ROW128
#undef BX_ROW64
#undef BX_ROW_OPS
            };
            // (straight-line per iteration: a batch in flight across a branch or the loop's back edge gets copied by the
            // compiler — a phi — before its wait, i.e. read before it has landed)
            for (int g = 0; g < cnt; g += 2 * FB) {
                yv_t ya[FB], yb[FB];
                issue(ya, g);
                issue(yb, g + FB);
                add_rows(ya, g, true);
                add_rows(yb, g + FB, false);
            }
        };
        // wait until all four words at `flags` have reached v (the waves of one role run the same program: short waits)
        auto wait_flags = [&](unsigned flags, int v) {
            for (;;) {
                i32x4 f;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(flags) : "memory");
                const int lo = min(min(f[0], f[1]), min(f[2], f[3]));
                if (__builtin_amdgcn_readfirstlane(lo) >= v) break;
                __builtin_amdgcn_s_sleep(1);
            }
        };

        // ---- one workgroup barrier per chunk.  During chunk k (between barriers k and k + 1):
        //   consumers: phase 0 from P0[k&1], phase 1 from P1[k&1]; once all four have read that tile, the chunk's rows Y(k)
        //              overwrite it
        //   helpers:   DMA P0[(k+1)&1] <- source rows of chunk k+1 (HBM: a whole chunk to land); fold Y(k-1) out of
        //              P1[(k-1)&1]; once all four have folded, DMA that tile <- destination rows of chunk k+1 (L2); table of
        //              chunk k; descriptor of chunk k+2
        // Descriptor pipeline of chunk j: chunk_tab entry requested during chunk j-5, words (source id, key) j-4, the rows'
        // scales j-3, published j-2.  Values move up one place at the START of a chunk, when they have long arrived (a
        // register copy of a value still in flight waits for it, and for every DMA issued before it).
        BxChunk ch[5];                                     // ch[i] = chunk k + i
        Words wdP{0, 0}, wdN{0, 0}, wdL{0, 0};             // words of chunks k+2 (to publish), k+3, k+4 (just requested)
        Scales scP{1.f, 1.f}, scN{1.f, 1.f};               // scales of chunks k+2, k+3 (just requested)
        i32x2 d5{0, 0};                                    // chunk_tab entry of chunk k+5 (just requested)
        if (hw == 0)                                                   // the row of zeros and the sixteen flags behind it
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (lane + 64 * i < 128 + 16) lds_st_b32(lds0 + ZERO_OFF + 4 * (lane + 64 * i), 0);
        int prev_rows = 1;
        if (nchunks > 0) {
            i32x2 dd[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) dd[j] = load_desc(j);
#pragma unroll
            for (int j = 0; j < 5; ++j) ch[j] = decode(dd[j]);
            Words wd[4];
            Scales sc[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) wd[j] = load_words(ch[j], lane);
#pragma unroll
            for (int j = 0; j < 3; ++j) sc[j] = load_scales(ch[j], wd[j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) publish(j, ch[j], wd[j], sc[j], lane);
            wdP = wd[2];
            scP = sc[2];
            wdN = wd[3];
        }
        BX_LGKM0();
        __builtin_amdgcn_s_barrier();                      // barrier A: descriptors 0 and 1 visible to all helper waves
        // (raw barriers in this role: __syncthreads() drains every LDS-DMA in flight — vmcnt(0) — before it)
        if (nchunks > 0) {
            if (!(skip & 1)) dma_tile(P0_OFF, 0, P0_IDS, ch[0], P0_NT, lane);
            if (!(skip & 2)) dma_tile(P1_OFF, 0, 3, ch[0], false, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (GHF_BX_LATE && lane == 0) lds_st_b32(lds0 + FLAG_OFF + 32 + 4 * hw, 1);   // chunk 0's destination rows are in
        }
        // DMA instructions per tile and helper wave (constant: dma_tile) — what the counted waits below leave in flight
        constexpr int N_SRC = (skip & 1) ? 0 : RBW * NPL, N_DST = ((skip & 2) || YT) ? 0 : RBW * NPL;
        int sid[6] = {0, 0, 0, 0, 0, 0};                     // source ids of the NEXT chunk's rows (read at the end of a chunk)
        if (nchunks > 0 && !(skip & 1)) dma_ids(1, P0_IDS, lane, sid);
        for (int k = 0; k < nchunks; ++k) {
            __builtin_amdgcn_s_barrier();                  // ---- chunk k
            BX_STAMP(0);
            const int l0 = opaque_lane(lane);
            if (k > 0) {                                   // everything requested during the last chunk has arrived
                wdP = wdN;
                scP = scN;
                wdN = wdL;
#pragma unroll
                for (int j = 0; j < 4; ++j) ch[j] = ch[j + 1];
                ch[4] = decode(d5);
                asm volatile("" : "+v"(wdP.src), "+v"(wdP.key), "+v"(scP.u), "+v"(scP.v), "+v"(wdN.src), "+v"(wdN.key));
            }
            if (!(GHF_BXEXP & 64)) publish(k + 2, ch[2], wdP, scP, l0);
            BX_STAMP(4);
            if (!(skip & 1)) dma_issue(P0_OFF + ((k + 1) & 1) * TILE, ch[1].rows, P0_NT, l0, sid);
            if (YT && !(skip & 2)) dma_tile(P1_OFF + ((k + 1) & 1) * TILE, k + 1, 3, ch[1], false, l0);
            if (GHF_BX_LATE && N_DST > 0) {
                // chunk k's destination rows (requested at the end of chunk k-1) have landed once only the source-row DMAs
                // just issued are in flight: tell the consumers, who need that tile for phase 1 only
                if (k > 0) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_SRC) : "memory");
                    if (lane == 0) lds_st_b32(lds0 + FLAG_OFF + 32 + 4 * hw, k + 1);
                }
            }
            BX_STAMP(1);
            if (k > 0 && !(GHF_BXEXP & 8)) {
                const FoldPlan fp = fold_plan(k - 1, prev_rows, l0);
                if (DEFER) wait_flags(lds0 + FLAG_OFF + 16, k);   // all four consumer waves have staged Y(k-1)
                BX_STAMP(5);                               // (stamps: the wait for the staged rows)
                fold_rows(k - 1, fp, l0);
            }
            BX_LGKM0();
            if (lane == 0) lds_st_b32(lds0 + FLAG_OFF + 4 * hw, k + 1);   // this wave is through with Y(k-1)
            // the destination ids of chunk k+1's rows (its tile's DMA): one round trip taken while the other helper waves catch up
            int did[6];
            if (!(skip & 2)) dma_ids(k + 1, 3, opaque_lane(lane), did);
            BX_STAMP(2);
            // without a tile of their own the staged rows sit where the next destination rows go: every helper wave must have
            // folded them first
            wait_flags(lds0 + FLAG_OFF, k + 1);
            BX_STAMP(3);
            const int l1 = opaque_lane(lane);
            // the next requests of the descriptor pipeline: behind the loops above (a load in flight across a loop makes hipcc
            // wait for everything — the source-row DMA included — at the loop), ahead of the rest of the chunk (requested
            // at its end they were waited for right behind the next barrier)
            if (!(GHF_BXEXP & 64)) {
                scN = load_scales(ch[3], wdN);
                wdL = load_words(ch[4], l1);
            }
            d5 = load_desc(k + 5);
            if (!YT && !(skip & 2)) dma_issue(P1_OFF + ((k + 1) & 1) * TILE, ch[1].rows, false, l1, did);
            BX_STAMP(1);
            prev_rows = ch[0].rows;
            if (!(skip & 1)) dma_ids(k + 2, P0_IDS, l1, sid);     // (published at this chunk's start by every helper wave; all are past their flag)
            // the source rows of chunk k+1 (requested at this chunk's start) must be in before the barrier; GHF_BX_LATE: the
            // destination-row DMAs issued last stay in flight across it
            // (counted in DMA instructions only — one per builtin; the descriptor loads before them are waited for as well:
            // how many instructions the compiler makes of those is not this code's to assume)
            if (GHF_BX_LATE && N_DST > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_DST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BX_LGKM0();
            BX_STAMP(4);
        }
        __builtin_amdgcn_s_barrier();                      // ---- epilogue: the last chunk's rows
        if (nchunks > 0 && !(GHF_BXEXP & 8)) {
            const int le = opaque_lane(lane);
            fold_rows(nchunks - 1, fold_plan(nchunks - 1, prev_rows, le), le);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be landing when the tiles are reused below
        BX_LGKM0();
        BX_STAMP(5);
        if constexpr (D == 128) asm volatile("" : BX_PIN_128(sm)); else asm volatile("" : BX_PIN_64(sm));
        // dump row HPW*hw + i = node NPW*hw + HPW*half + i, natural column order: position PL*lane + e is column
        // 32 (lane / 16) + 16 e + lane % 16 at d = 128 (a consumer wave's two fragments, interleaved), column lane at d = 64
        // (two explicit halves, not a loop: in a loop body both halves' registers stay live through the first half's tail)
        auto dump_half = [&](auto half_c) __attribute__((always_inline)) {
            constexpr int H = decltype(half_c)::value;
            float* __restrict__ o = (float*)smem + (size_t)(hw * HPW) * D + (D == 128 ? 32 * (lane >> 4) + (lane & 15) : lane);
#pragma unroll
            for (int i = 0; i < HPW; ++i)
#pragma unroll
                for (int e = 0; e < PL; ++e) {
                    const int reg = PL * (H * HPW + i) + e;
                    o[i * D + 16 * e] = sm[reg / TUP][reg % TUP];
                }
        };
        __syncthreads();
        dump_half(std::integral_constant<int, 0>{});
        BX_LGKM0();
        __syncthreads();
        tail_half(0, std::integral_constant<int, 1>{});              // (the other half of the sums is still in registers)
        BX_STAMP(6);
        __syncthreads();
        dump_half(std::integral_constant<int, 1>{});
        BX_LGKM0();
        __syncthreads();
        tail_half(1, std::integral_constant<int, GHF_BX_TGB>{});
        BX_STAMP(7);
        BX_STAMP_FLUSH();
    } else {
        // =============================================== CONSUMERS ===============================================
        // B fragments (GHF_WLAYOUT_SPLIT2H): Wh[r][o/16][kk/32][piece][lane][8] fp16, then one float 2^-s per relation
        const uint32_t wsc_off = (uint32_t)((uint64_t)R * 2 * D * D * (NPL * 2));
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wsplit, 0, (int)wsc_off, 0x00020000);
        auto b_soff = [&](int r, int ph, int t) -> int {
            if (GHF_BXEXP & 128) r = 0;                    // (timing experiment: one relation's weights, always hot in L2)
            return __builtin_amdgcn_readfirstlane((((r * NT + tw * NTW + t) * NKS + ph * KS) * NPL) * 1024);
        };
        const int lane16 = lane * 16;
        i32x4 b[KS][NTW][NPL];
        auto load_b_step = [&](int r, int ph, int j) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    b[j][t][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16 + pl * 1024, b_soff(r, ph, t) + j * (NPL * 1024), GHF_B_AUX);
        };
        f32x4 acc[MTC][NTW];
        const int arow = c16 * ROWB;
        // One instance per number of live row tiles: no branch, no LDS read and no zeroing for dead tiles inside the loop
        // (a uniform branch per (k-step, tile) position cost the MFMA stream a fetch bubble each)
        // (always_inline: with two call sites and little work per instance — d = 64 — hipcc otherwise makes the stage a real
        // function, called through s_swappc with its captures in scratch: 10x the time)
        constexpr int AD = GHF_BX_AD;
        static_assert(AD <= MTC - 2, "the prefetched positions are row tiles 0 .. AD-1 of the first k-step in every instance");
        i32x4 apre[AD][NPL];                               // the fragments of a stage's first AD positions, requested ahead of the stage
        auto lda_from = [&](const char* Abuf, int j, int m, i32x4 (&dst)[NPL]) __attribute__((always_inline)) {
            const char* src = Abuf + arow + (((4 * j + q) ^ akey(c16)) << 4) + m * 16 * ROWB;     // akey(16 m + c16) = akey(c16)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) dst[pl] = *(const i32x4*)(src + pl * PLANE);
        };
        auto prefetch_a = [&](const char* Abuf) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < AD; ++p) lda_from(Abuf, 0, p, apre[p]);
        };
        auto compute_stage = [&](auto mt_c, auto pre_c, int ph, bool first, const char* Abuf, const int* meta, float wscale, int r_next, int ph_next,
                                 const float (&bias_v)[NTW], auto&& between, auto&& after_k) __attribute__((always_inline)) {
            constexpr int MT = decltype(mt_c)::value;
            constexpr bool PRE = decltype(pre_c)::value;
            f32x4 part[MTC][NTW];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) part[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 sc[MTC];                                  // the rows' scales: read during the last k-step (short live range)
            // A fragments of (k-step, tile) positions p .. p + AD: with the helpers' DMA writes and fold reads on the LDS a read
            // takes ~250 cycles, a position's six MFMAs 96 (tools/stamps_bx.py: two positions ahead left the stage
            // latency-bound)
            i32x4 a[AD + 1][NPL];
            auto lda = [&](int j, int m, i32x4 (&dst)[NPL]) { lda_from(Abuf, j, m, dst); };
#pragma unroll
            for (int p = 0; p < AD && p < KS * MT; ++p) {
                if constexpr (PRE) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) a[p][pl] = apre[p][pl];
                } else {
                    lda(p / MT, p % MT, a[p]);
                }
            }
            BX_STAMP(4);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j == KS - 1) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) sc[m] = *(const f32x4*)(meta + ph * CRP + m * 16 + 4 * q);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int p = j * MT + m, cur = p % (AD + 1);
                    if (p + AD < KS * MT) lda((p + AD) / MT, (p + AD) % MT, a[(p + AD) % (AD + 1)]);
                    if (GHF_BXEXP & 4) {                                    // operands stay alive (no DCE of the loads)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) asm volatile("" ::"v"(a[cur][pl]));
#pragma unroll
                        for (int t = 0; t < NTW; ++t)
#pragma unroll
                            for (int pl = 0; pl < NPL; ++pl) asm volatile("" ::"v"(b[j][t][pl]));
                    } else {
#pragma unroll
                        for (int t = 0; t < NTW; ++t) {
                            auto fma = [&](int pa, int pb) {
                                part[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[cur][pa]),
                                                                                    __builtin_bit_cast(f16x8, b[j][t][pb]),
                                                                                    part[m][t], 0, 0, 0);
                            };
                            fma(1, 0); fma(0, 1);                               // lo*hi, hi*lo
                            fma(0, 0);                                          // hi*hi
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                BX_STAMP(5);
                between(j);
                if (!(GHF_BXEXP & 1)) load_b_step(r_next, ph_next, j);
                __builtin_amdgcn_sched_barrier(0);
                BX_STAMP(7);
            }
            after_k();                                      // (hook: the next stage's first fragments, requested before the unscale)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float f = sc[m][s] * wscale;
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[m][t][s] = fmaf(part[m][t][s], f, first ? bias_v[t] : acc[m][t][s]);
                }
            }
            // (the tiles this instance does not compute: defined here, so that their old values need not survive the stage)
#pragma unroll
            for (int m = MT; m < MTC; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        auto stage_for = [&](int mt, auto pre_c, int ph, bool first, const char* Abuf, const int* meta, float wscale, int r_next, int ph_next,
                             const float (&bias_v)[NTW], auto&& between, auto&& after_k) __attribute__((always_inline)) {
            // three instances: all tiles, one fewer, two fewer (shorter chunks — 7 % at C3 — run the last one: their dead
            // tiles cost MFMAs on stale rows that are never written)
            static_assert(MTC >= 3, "three compute_stage instances");
            if (mt >= MTC) compute_stage(std::integral_constant<int, MTC>{}, pre_c, ph, first, Abuf, meta, wscale, r_next, ph_next, bias_v, between, after_k);
            else if (mt == MTC - 1) compute_stage(std::integral_constant<int, MTC - 1>{}, pre_c, ph, first, Abuf, meta, wscale, r_next, ph_next, bias_v, between, after_k);
            else compute_stage(std::integral_constant<int, MTC - 2>{}, pre_c, ph, first, Abuf, meta, wscale, r_next, ph_next, bias_v, between, after_k);
        };
        // a chunk's finished rows -> Y: lane (q, c16) holds rows 16m + 4q + s, positions 32tw + 2c16 + t (t = 0, 1) — with one
        // fragment per wave (d = 64), position 16tw + c16
        const unsigned yoff = (unsigned)(4 * q) * (D * 4) + (unsigned)((((NTW == 2 ? 32 * tw + 2 * c16 : 16 * tw + c16)) ^ ((q & 1) << 5)) * 4);
        auto write_rows = [&](int mt, unsigned ytile) __attribute__((always_inline)) {
            const unsigned ybase = lds0 + ytile + yoff;
#pragma unroll
            for (int m = 0; m < MTC; ++m) {
                if (m >= mt) continue;
                // the last tile's rows past CR do not exist in Y (the table follows it)
                if (16 * m + 16 > CR && 16 * m + 4 * q >= CR) continue;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (NTW == 2)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ybase), "v"((f32x2){acc[m][0][s], acc[m][NTW - 1][s]}),
                                     "n"((16 * m + s) * (D * 4)) : "memory");
                    else
                        asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(ybase), "v"(acc[m][0][s]), "n"((16 * m + s) * (D * 4)) : "memory");
                }
            }
        };

        // the same rows, slice j of KS (row tiles m = j, j + KS, ...; every tile, live or not: no branch — the helpers read the
        // chunk's rows only)
        auto write_rows_slice = [&](int j, unsigned ytile) __attribute__((always_inline)) {
            const unsigned ybase = lds0 + ytile + yoff;
#pragma unroll
            for (int m = 0; m < MTC; ++m) {
                if (m % KS != j) continue;
                if (16 * m + 16 > CR && 16 * m + 4 * q >= CR) continue;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (NTW == 2)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ybase), "v"((f32x2){acc[m][0][s], acc[m][NTW - 1][s]}),
                                     "n"((16 * m + s) * (D * 4)) : "memory");
                    else
                        asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(ybase), "v"(acc[m][0][s]), "n"((16 * m + s) * (D * 4)) : "memory");
                }
            }
        };
        auto nothing = [](int) {};
        const int ph_first = (skip & 1) ? 1 : P0_HALF;      // the weights' half of a chunk's first live phase
        BxChunk ch{0, 0, 1};
        i32x2 dn{0, 0};
        float bias_v[NTW] = {}, bias_n[NTW] = {}, wscale = 1.f, wscale_n = 1.f;
        auto load_rel_words = [&](int r, float& ws, float (&bv)[NTW]) {
            ws = *bx_at<float>(Wsplit, wsc_off + (uint32_t)(r + vzero) * 4u);
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[t] = *bx_at<float>(bias, (uint32_t)(r * D + (tw * NTW + t) * 16 + c16) * 4u);
        };
        int mt_prev = 0;
        static_assert(!(DEFER && YT), "GHF_BX_DEFER stages into the aliased tile");
        if (nchunks > 0) {
            ch = decode(load_desc(0));
            dn = load_desc(1);
            load_rel_words(ch.r, wscale, bias_v);
#pragma unroll
            for (int j = 0; j < KS; ++j) load_b_step(ch.r, ph_first, j);
        }
        __syncthreads();                                   // barrier A
        for (int k = 0; k < nchunks; ++k) {
            const int mt = (ch.rows + 15) >> 4;
            const int* meta = (const int*)(smem + meta_off(k));
            __syncthreads();                               // ---- chunk k
            BX_STAMP(0);
            #ifndef GHF_BX_PRE0
#define GHF_BX_PRE0 0          // 1: the first phase's first fragments requested right behind the barrier (round 4, LATE off: 2.86 vs 2.83 ms — not kept)
#endif
            constexpr bool PRE_OK = skip == 0 && !YT && (GHF_BX_LATE || GHF_BX_PRE0);     // (the source-row tile has landed before the barrier either way)
            constexpr bool PRE0 = PRE_OK, PRE1 = false;        // first / second phase (the second phase's ahead of the unscale: 44 spilled registers)
            using pre0_t = std::integral_constant<bool, PRE0>;
            using pre1_t = std::integral_constant<bool, PRE1>;
            if (PRE0) prefetch_a(smem + P0_OFF + (k & 1) * TILE);      // (chunk k's source rows landed before the barrier)
            // GHF_BX_DEFER: the previous chunk's rows go to their staging tile (that chunk's destination-row tile) only now:
            // behind the barrier every consumer wave is through with that tile, so there is no hand-shake among the consumers,
            // and the accumulators are not needed before this chunk's first phase ends.  The helpers wait for the flag.
            constexpr bool ILV = false;                         // (staging writes between the k-steps: measured slower, round 3)
            if (DEFER && k > 0 && !ILV) {
                if (!(GHF_BXEXP & 16)) write_rows(mt_prev, P1_OFF + ((k - 1) & 1) * TILE);
                BX_LGKM0();
                if (lane == 0) lds_st_b32(lds0 + FLAG_OFF + 16 + 4 * tw, k);
                BX_STAMP(3);
            }
            const BxChunk nx = decode(dn);
            // (a half whose weights the caller declared zero is not computed: the next live stage's weights are prefetched)
            // behind the first phase's k-steps: this chunk's destination rows have landed (helpers' flags) -> the second phase's
            // first fragments are requested before the first phase's unscale
            auto wait_landed = [&]() __attribute__((always_inline)) {
                for (;;) {
                    i32x4 f;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(lds0 + FLAG_OFF + 32) : "memory");
                    const int lo = min(min(f[0], f[1]), min(f[2], f[3]));
                    if (__builtin_amdgcn_readfirstlane(lo) >= k + 1) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            };
            auto next_pre = [&]() __attribute__((always_inline)) {
                if (PRE1) {
                    wait_landed();
                    prefetch_a(smem + P1_OFF + (k & 1) * TILE);
                }
            };
            auto no_after = []() {};
            if (!(skip & 1)) {
                if (ILV && k > 0) {
                    const unsigned ytile = P1_OFF + ((k - 1) & 1) * TILE;
                    stage_for(mt, pre0_t{}, P0_HALF, true, smem + P0_OFF + (k & 1) * TILE, meta, wscale, (skip & 2) ? nx.r : ch.r, (skip & 2) ? P0_HALF : 1, bias_v,
                              [&](int j) __attribute__((always_inline)) { write_rows_slice(j, ytile); }, next_pre);
                } else {
                    stage_for(mt, pre0_t{}, P0_HALF, true, smem + P0_OFF + (k & 1) * TILE, meta, wscale, (skip & 2) ? nx.r : ch.r, (skip & 2) ? P0_HALF : 1, bias_v, nothing, next_pre);
                }
            }
            if (ILV && k > 0) {                            // (the stage's unscale has not touched acc's OLD values before this point:
                BX_LGKM0();                                //  compute_stage writes acc after its k-steps — the writes above read it before)
                if (lane == 0) lds_st_b32(lds0 + FLAG_OFF + 16 + 4 * tw, k);
            }
            BX_STAMP(1);
            dn = load_desc(k + 2);
            load_rel_words(nx.r, wscale_n, bias_n);
            if (GHF_BX_LATE && !(skip & 2) && !YT && !PRE1) wait_landed();   // the destination-row tile of this chunk has landed (helpers' flags)
            if (skip & 1) {                                // no source phase ran: the destination phase adds to the bias
#pragma unroll
                for (int m = 0; m < MTC; ++m)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){bias_v[t], bias_v[t], bias_v[t], bias_v[t]};
            }
            if (!(skip & 2)) stage_for(mt, pre1_t{}, 1, false, smem + P1_OFF + (k & 1) * TILE, meta, wscale, nx.r, ph_first, bias_v, nothing, no_after);
            BX_STAMP(2);
            // YT: the staging tile is free once every helper wave has folded the previous chunk's rows (flag = k + 1, set during
            // this chunk); else the chunk's rows overwrite its destination-row tile once every consumer wave has read it
            // the last chunk (and every chunk without GHF_BX_DEFER) stages its rows here, once all four consumer waves have read
            // the tile they overwrite (flag words; YT: once every helper wave has folded the previous chunk's rows)
            const bool stage_now = !DEFER || k + 1 == nchunks;
            if (stage_now) {
                const unsigned fw = DEFER ? FLAG_OFF + 48 : FLAG_OFF + (YT ? 0 : 16);      // (DEFER: words of their own)
                const int fv = DEFER ? 1 : k + 1;
                if (!YT && lane == 0) lds_st_b32(lds0 + (DEFER ? FLAG_OFF + 48 : FLAG_OFF + 16) + 4 * tw, fv);
                for (;;) {
                    i32x4 f;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(lds0 + fw) : "memory");
                    const int lo = min(min(f[0], f[1]), min(f[2], f[3]));
                    if (__builtin_amdgcn_readfirstlane(lo) >= fv) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!(GHF_BXEXP & 16)) write_rows(mt, YT ? Y_OFF : P1_OFF + (k & 1) * TILE);
            }
            if (GHF_BXEXP & 16)
#pragma unroll
                for (int m = 0; m < MTC; ++m)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) asm volatile("" ::"v"(acc[m][t]));
            mt_prev = mt;
            ch = nx;
            wscale = wscale_n;
#pragma unroll
            for (int t = 0; t < NTW; ++t) bias_v[t] = bias_n[t];
            BX_LGKM0();
            BX_STAMP(3);
        }
        __syncthreads();                                   // epilogue stage
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            __syncthreads();
            tail_half(half, std::integral_constant<int, GHF_BX_TGB>{});
            if (half == 0) BX_STAMP(6);
        }
        BX_STAMP(6);
        BX_STAMP_FLUSH();
    }
}

template <int D>
static int launch_bx_for(const MsgArgs& a, hipStream_t stream) {
    using C = BxCfg<D>;
    constexpr size_t lds = C::LDS;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((size_t)(C::BN / 2) * D * 4 <= (size_t)3 * 2 * C::CR * (D * 2), "the tail's dump of half a block must fit the three A tiles");
    GHF_REQUIRE(a.block_nodes == C::BN, "message(bx): plan block_nodes=%d, kernel for d=%d needs %d", a.block_nodes, D, C::BN);
    GHF_REQUIRE(a.wlayout == GHF_WLAYOUT_SPLIT2H, "message(bx): weights must be in SPLIT2H layout");
    GHF_REQUIRE(a.chunk_tab && a.item_tab && a.blk_item_off, "message(bx): the plan's chunk / item tables are missing");
    GHF_REQUIRE(a.h_split, "message(bx): h_split is missing (ghf_split_rows)");
    const int64_t row_end = a.row0 + a.rows;
    GHF_REQUIRE(row_end == a.N || row_end % C::BN == 0, "message(bx): row range must end on a block boundary or at N");
    if (a.rows <= 0) return GHF_OK;
    // (dead rows point at byte 0xFFFFF000 of the row table's buffer: past its end — zeros, no memory access — while N*(4d+4)
    // stays below that; plan.block_kernel_max_nodes routes larger graphs to a CSR plan)
    GHF_REQUIRE((uint64_t)a.N * (D * 4 + 4) <= 0xFFFFF000ull && (uint64_t)a.E * 4 < (1ull << 32) &&
                    (uint64_t)a.R * (2 * D * D * 4 + 4) < (1ull << 32),
                "message(bx): 32-bit byte offsets need N*(4d+4) <= 4 GiB - 4 KiB, E*4 and R*(8d*d+4) below 4 GiB");
    GHF_REQUIRE(a.n_items >= cdiv(a.rows, C::BN), "message(bx): n_items=%lld is fewer than the blocks of the row range", (long long)a.n_items);
    GHF_REQUIRE(a.n_items == cdiv(a.rows, C::BN) || a.partial, "message(bx): split blocks need the `partial` scratch");
    const int skip = ((a.flags & GHF_FLAG_ZERO_SRC) ? 1 : 0) | ((a.flags & GHF_FLAG_ZERO_DST) ? 2 : 0);
    const size_t lds_dyn = lds;
    auto go = [&](auto skip_c) {
        constexpr int S = decltype(skip_c)::value;
        GHF_SET_MAX_LDS((message_bx_kernel<D, S>), lds_dyn);
        message_bx_kernel<D, S><<<(unsigned)a.n_items, 512, lds_dyn, stream>>>(a.h, a.h_split, a.N, a.sorted_key, a.sorted_src, a.chunk_tab, a.item_tab,
                                                                          a.item0, a.partial, a.indeg, a.R, a.W_msg, a.bias, a.ln_gamma,
                                                                          a.ln_beta, a.ln_eps, a.row0, row_end, a.h_out, a.h_split_out,
                                                                          a.flags & (GHF_FLAG_NO_TAIL | GHF_FLAG_RAW_SUM | GHF_FLAG_ADD_H), range_flag_ptr(), a.agg_out);
        return GHF_OK;
    };
    if (skip == 1) go(std::integral_constant<int, 1>{});
    else if (skip == 2) go(std::integral_constant<int, 2>{});
    else go(std::integral_constant<int, 0>{});
    GHF_LAUNCH_CHECK();
    if (a.n_items > cdiv(a.rows, C::BN)) return launch_combine_split(a, stream);     // some block of the range is split
    return GHF_OK;
}

// ghf_split_rows, SPLIT2H: one wave per row — the row's largest magnitude picks the power of two
__global__ __launch_bounds__(256) void split2h_rows_kernel(const float* __restrict__ h, int64_t N, int64_t row0, int64_t rows,
                                                           int d, char* __restrict__ out, int32_t* __restrict__ range_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const int64_t row = row0 + i;
    const float* __restrict__ src = h + row * d;
    float mx = 0.f;
    for (int k = lane; k < d; k += 64) mx = fmaxf(mx, fabsf(src[k]));
    const int sh = split2h_shift(wave_absmax(mx));
    const float up = pow2f(sh);
    _Float16* __restrict__ dst = (_Float16*)(out + row * (4 * (int64_t)d));
    int tiny = 0, nz = 0;
    for (int k = lane; k < d; k += 64) {
        _Float16 hi, lo;
        const float xs = src[k] * up;
        split2h(xs, hi, lo);
        dst[k] = hi;
        dst[d + k] = lo;
        tiny += __popcll(__ballot(range_tiny(xs)));
        nz += __popcll(__ballot(xs != 0.f));
    }
    if (lane == 0) {
        *(float*)(out + N * (4 * (int64_t)d) + row * 4) = pow2f(-sh);
        range_raise(range_flag, GHF_RANGE_ROWS, tiny, nz);
    }
}

int launch_split2h_rows(const float* h, int64_t N, int d, int64_t row0, int64_t rows, void* h_split, hipStream_t stream) {
    if (rows <= 0) return GHF_OK;
    GHF_REQUIRE(cdiv(rows, 4) < (1ll << 31), "split_rows: too many rows per launch");
    split2h_rows_kernel<<<(unsigned)cdiv(rows, 4), 256, 0, stream>>>(h, N, row0, rows, d, (char*)h_split, range_flag_ptr());
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

bool message_bx_config(int d, int* block_nodes, int* chunk_rows, int* split_chunks) {
    if (d != 128 && d != 64) return false;
    *block_nodes = d == 128 ? BxCfg<128>::BN : BxCfg<64>::BN;
    *chunk_rows = d == 128 ? BxCfg<128>::CR : BxCfg<64>::CR;
    *split_chunks = 128;
    return true;
}

bool message_bx_owns(int d, int block_nodes) {
    return (d == 128 && block_nodes == BxCfg<128>::BN) || (d == 64 && block_nodes == BxCfg<64>::BN);
}

int launch_message_bx(const MsgArgs& a, hipStream_t stream) {
    if (a.d == 128) return launch_bx_for<128>(a, stream);
    if (a.d == 64) return launch_bx_for<64>(a, stream);
    return set_err(GHF_EUNSUPPORTED, "message(bx): no kernel for d=%d", a.d);
}

}  // namespace ghf

#ifdef GHF_STAMPS
extern "C" int ghf_debug_read_stamps_bx(unsigned long long* host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ghf::ghf_bx_stamp_buf), count * sizeof(unsigned long long));
}
extern "C" int ghf_debug_read_life_bx(unsigned long long* host, size_t count) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ghf::ghf_bx_life_buf), count * sizeof(unsigned long long));
}
#endif
