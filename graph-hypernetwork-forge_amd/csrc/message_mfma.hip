// message_mfma.hip — K2+K3 on the matrix cores (gfx950), hidden sizes 64 and 128.
//
// Replaces models/hypergnn.py:281-296 of the reference:
//   out_v = (1/max(indeg_v,1)) * sum_{e=(u->v)} ( h_u W_msg[r_e] + bias[r_e] + h_v W_self[r_e] )
//   h'_v  = LayerNorm(ReLU(out_v + h_v))
//
// Geometry.  One workgroup (8 waves) owns BN consecutive destination nodes and keeps
// their fp32 sums [BN][D] in LDS for the whole kernel: no global atomics, the tail is
// fused, and every h' row is written exactly once.  The plan (plan.hip) has sorted the
// block's in-edges by relation, so the block walks "chunks": <= CR rows of ONE relation r.
// A chunk is a small dense GEMM  [rows, 2D] x [2D, D]  with
//   A row  = [h_src | h_dst]   (gathered into LDS by LDS-DMA, one 1 KiB piece per wave-instr)
//   B      = [W_msg[r]; W_self[r]]  (pre-arranged by K1 in MFMA fragment order, GHF_WLAYOUT_FRAG16)
// run as two K-phases of D (phase 0: h_src x W_msg, phase 1: h_dst x W_self) so that the
// A tile of one phase is gathered while the other phase computes (1 barrier per phase).
// Waves split the OUTPUT COLUMNS (wave = 16*NTW columns, x MG row groups): a wave streams its
// own column slab of W[r] from L2 straight into registers (each fragment is reused by every
// row tile of the chunk) and all waves share the A tile through LDS.
// Math: v_mfma_f32_16x16x4_f32, an exact fp32 fma chain; bias enters as the initial accumulator.
// After phase 1 each wave adds its 16-column strip of the chunk's rows into the LDS sums with
// ds_add_f32 (rows of one chunk may share a destination).
//
// LDS (D=128): sums 216*512 B + 2 A tiles 48*512 B + 48 dst-local ids = 159,936 B (1 workgroup/CU).
#include "common.h"

namespace ghf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int D> struct MfmaCfg;
template <> struct MfmaCfg<128> { static constexpr int BN = 216, MTC = 3, NCG = 8, NTW = 1, MG = 1; };
template <> struct MfmaCfg<64>  { static constexpr int BN = 432, MTC = 6, NCG = 4, NTW = 1, MG = 2; };

struct Chunk { int r; int e0; int rows; };     // rows == 0: no chunk

template <int D>
__global__ __launch_bounds__(512) void message_mfma_kernel(
    const float* __restrict__ h, int64_t N, const uint32_t* __restrict__ sorted_key,
    const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ indeg, int R, const float* __restrict__ Wfrag, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    int64_t row0, int64_t row_end, float* __restrict__ h_out, int no_tail) {
    using C = MfmaCfg<D>;
    constexpr int BN = C::BN, MTC = C::MTC, NCG = C::NCG, NTW = C::NTW, MG = C::MG;
    constexpr int NJ = D / 16;            // k-groups of 16 per phase
    constexpr int NT = D / 16;            // 16-column tiles of the output
    constexpr int NJ2 = 2 * NJ;
    constexpr int CPR = D / 4;            // 16-byte chunks per A row
    constexpr int RPI = 256 / D;          // A rows per 1 KiB LDS-DMA wave-instruction
    constexpr int CR = 16 * MTC;          // rows per chunk
    constexpr int IPW = CR / RPI / 8;     // LDS-DMA instructions per wave per stage
    constexpr int MTW = MTC / MG;         // row tiles per wave
    static_assert(NCG * MG == 8 && NCG * NTW == NT && CR % (RPI * 8) == 0 && MTC % MG == 0, "bad tile config");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = (float*)smem;                    // [BN][D]
    float* A0 = acc_lds + BN * D;                     // [CR][D], 16-byte chunks XOR-swizzled by (row & 15)
    float* A1 = A0 + CR * D;
    int* s_dstl = (int*)(A1 + CR * D);                // [CR] destination index local to the block

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, c16 = lane & 15;
    const int cg = w % NCG, mg = w / NCG;
    const int64_t blk = row0 / BN + blockIdx.x;
    const int64_t node0 = blk * BN;
    const int nrows = (int)((row_end - node0) < BN ? (row_end - node0) : BN);
    const int32_t* __restrict__ goff = seg_off + blk * R;
    const uint32_t seg0 = (uint32_t)(blk * R);

    for (int i = tid; i < BN * D / 4; i += 512) ((f32x4*)acc_lds)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int pos = __builtin_amdgcn_readfirstlane(goff[0]);
    const int blk_end = __builtin_amdgcn_readfirstlane(goff[R]);

    auto next_chunk = [&](int p) -> Chunk {
        Chunk c{0, p, 0};
        if (p < blk_end) {
            const uint32_t key = __builtin_amdgcn_readfirstlane(sorted_key[p]);
            c.r = (int)(key / (uint32_t)BN - seg0);
            const int end = __builtin_amdgcn_readfirstlane(goff[c.r + 1]);
            c.rows = (end - p) < CR ? (end - p) : CR;
        }
        return c;
    };

    // node ids of the rows this lane's LDS-DMA pieces gather for (chunk, phase)
    auto load_idx = [&](const Chunk& c, int ph, int (&idx)[IPW]) {
        const uint32_t kbase = (seg0 + (uint32_t)c.r) * (uint32_t)BN;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            int rho = (w * IPW + i) * RPI + lane / CPR;
            rho = rho < c.rows ? rho : c.rows - 1;
            const int e = c.e0 + rho;
            idx[i] = ph == 0 ? sorted_src[e] : (int)(node0 + (sorted_key[e] - kbase));
        }
    };

    auto issue_stage = [&](const Chunk& c, int ph, float* Abuf, const int (&idx)[IPW]) {
        const int live_rows = (c.rows + 15) & ~15;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int piece = w * IPW + i;                 // wave-uniform
            if (piece * RPI < live_rows) {
                const int rho = piece * RPI + lane / CPR;  // LDS row this lane writes
                const int p = lane % CPR;                  // LDS 16-byte slot within the row
                const float* src = h + (size_t)idx[i] * D + ((p ^ (rho & 15)) << 2);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Abuf + piece * 256), 16, 0, 0);
                if (ph == 1 && p == 0 && rho < c.rows) s_dstl[rho] = idx[i] - (int)node0;
            }
        }
    };

    auto load_b = [&](int r, int ph, f32x4 (&b)[NJ][NTW]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float* base = Wfrag + ((size_t)(r * NT + cg * NTW + t) * NJ2 + ph * NJ) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j][t] = *(const f32x4*)(base + j * 256);
        }
    };

    f32x4 acc[MTW][NTW];

    auto compute_phase = [&](const float* Abuf, const f32x4 (&b)[NJ][NTW], int mtw_cur) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 a[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
                if (m < mtw_cur) {
                    const int row = (mg + m * MG) * 16 + c16;            // row & 15 == c16
                    a[m] = *(const f32x4*)(Abuf + row * D + (((4 * j + q) ^ c16) << 2));
                }
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        if (m < mtw_cur)
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[j][t][s], acc[m][t], 0, 0, 0);
        }
    };

    int idx0[IPW], idx1[IPW];
    f32x4 b0[NJ][NTW], b1[NJ][NTW];

    Chunk cur = next_chunk(pos);  pos += cur.rows;
    Chunk nxt = next_chunk(pos);  pos += nxt.rows;
    if (cur.rows) {
        load_idx(cur, 0, idx0);
        load_idx(cur, 1, idx1);
        issue_stage(cur, 0, A0, idx0);
        load_b(cur.r, 0, b0);
    }

    while (cur.rows) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // A0 = stage(cur,0) landed; all waves are past phase 1 of the previous chunk
        issue_stage(cur, 1, A1, idx1);
        load_b(cur.r, 1, b1);
        if (nxt.rows) load_idx(nxt, 0, idx0);

        const int mt_cur = (cur.rows + 15) >> 4;
        const int mtw_cur = (mt_cur - mg + MG - 1) / MG;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float bv = bias[(size_t)cur.r * D + (cg * NTW + t) * 16 + c16];
#pragma unroll
            for (int m = 0; m < MTW; ++m) acc[m][t] = (f32x4){bv, bv, bv, bv};
        }
        compute_phase(A0, b0, mtw_cur);

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // A1 = stage(cur,1) + s_dstl landed; all waves are past phase 0
        const Chunk nn = next_chunk(pos);  pos += nn.rows;
        if (nxt.rows) {
            issue_stage(nxt, 0, A0, idx0);
            load_b(nxt.r, 0, b0);
            load_idx(nxt, 1, idx1);
        }
        compute_phase(A1, b1, mtw_cur);

        // add this wave's 16*NTW-column strip of the chunk's rows into the block sums
#pragma unroll
        for (int m = 0; m < MTW; ++m)
            if (m < mtw_cur) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int rho = (mg + m * MG) * 16 + 4 * q + s;
                    if (rho < cur.rows) {
                        float* dst = acc_lds + s_dstl[rho] * D + cg * NTW * 16 + c16;
#pragma unroll
                        for (int t = 0; t < NTW; ++t) atomicAdd(dst + 16 * t, acc[m][t][s]);
                    }
                }
            }
        cur = nxt;
        nxt = nn;
    }
    __syncthreads();

    // ---- fused tail: one wave per destination row ------------------------------------
    constexpr int CPL = D / 64;              // columns per lane
    float g[CPL], bt[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        g[c] = no_tail ? 1.f : gamma[lane * CPL + c];
        bt[c] = no_tail ? 0.f : beta[lane * CPL + c];
    }
    for (int v = w; v < nrows; v += 8) {
        const int64_t node = node0 + v;
        const int deg = indeg[node];
        const float inv = 1.0f / (float)(deg > 1 ? deg : 1);
        float x[CPL];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const float a = acc_lds[v * D + lane * CPL + c] * inv;
            x[c] = no_tail ? a : fmaxf(a + h[(size_t)node * D + lane * CPL + c], 0.f);
            s += x[c];
        }
        if (!no_tail) {
            const float mean = wave_sum(s) * (1.0f / D);
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) { const float t = x[c] - mean; var += t * t; }
            const float rstd = 1.0f / sqrtf(wave_sum(var) * (1.0f / D) + eps);
#pragma unroll
            for (int c = 0; c < CPL; ++c) x[c] = (x[c] - mean) * rstd * g[c] + bt[c];
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) h_out[(size_t)node * D + lane * CPL + c] = x[c];
    }
}

template <int D>
static int launch_for(const MsgArgs& a, hipStream_t stream) {
    using C = MfmaCfg<D>;
    constexpr int CR = 16 * C::MTC;
    constexpr size_t lds = (size_t)(C::BN * D + 2 * CR * D) * 4 + CR * 4;
    GHF_REQUIRE(a.block_nodes == C::BN, "message(mfma): plan block_nodes=%d, kernel for d=%d needs %d", a.block_nodes, D, C::BN);
    GHF_REQUIRE(a.wlayout == GHF_WLAYOUT_FRAG16, "message(mfma): weights must be in FRAG16 layout");
    const int64_t row_end = a.row0 + a.rows;
    GHF_REQUIRE(row_end == a.N || row_end % C::BN == 0, "message(mfma): row range must end on a block boundary or at N");
    if (a.rows <= 0) return GHF_OK;
    static bool attr_set = false;
    if (!attr_set) {
        GHF_HIP_CHECK(hipFuncSetAttribute((const void*)message_mfma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const unsigned grid = (unsigned)cdiv(a.rows, C::BN);
    message_mfma_kernel<D><<<grid, 512, lds, stream>>>(a.h, a.N, a.sorted_key, a.sorted_src, a.seg_off, a.indeg, a.R,
                                                       a.W_msg, a.bias, a.ln_gamma, a.ln_beta, a.ln_eps, a.row0, row_end,
                                                       a.h_out, (a.flags & GHF_FLAG_NO_TAIL) ? 1 : 0);
    GHF_LAUNCH_CHECK();
    return GHF_OK;
}

bool message_mfma_config(int d, int* block_nodes) {
    switch (d) {
        case 128: *block_nodes = MfmaCfg<128>::BN; return true;
        case 64:  *block_nodes = MfmaCfg<64>::BN;  return true;
        default:  return false;
    }
}

int launch_message_mfma(const MsgArgs& a, hipStream_t stream) {
    switch (a.d) {
        case 128: return launch_for<128>(a, stream);
        case 64:  return launch_for<64>(a, stream);
        default:  return set_err(GHF_EUNSUPPORTED, "message(mfma): no tuned kernel for d=%d", a.d);
    }
}

}  // namespace ghf
