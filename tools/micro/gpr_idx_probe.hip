// Probe (gfx950): VGPR indexing mode — s_set_gpr_idx_on with a uniform index, then VALU adds whose destination and second
// source are relative to that index — as the fold of message_bx.hip uses it: a wave adds a staged row into the registers of
// the row's destination node, the node being a run-time (wave-uniform) value.
//   hipcc --offload-arch=gfx950 -O3 -o gpr_idx_probe gpr_idx_probe.hip && ./gpr_idx_probe
// Checks: (1) v_add_f32 with DST_REL | SRC1_REL; (2) v_pk_add_f32 on a register pair; (3) M0 saved and restored around the
// mode (LDS-DMA keeps its LDS base there); (4) cycles per indexed add pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PIN "+{v[64:95]}"(a0), "+{v[96:127]}"(a1)
template <int MODE>
__global__ __launch_bounds__(64) void k(const int* __restrict__ idx, const float* __restrict__ val, int n, int reps, float* __restrict__ out,
                                        long long* __restrict__ cyc) {
    const int lane = threadIdx.x;
    f32x32 a0, a1;
    for (int i = 0; i < 32; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    asm volatile("" : PIN);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < n; ++i) {
            const int j = __builtin_amdgcn_readfirstlane(idx[i]);     // register pair 0..31
            const f32x2 v = *(const f32x2*)(val + (size_t)i * 128 + 2 * lane);
            int keep;
            if (MODE == 0)
                asm volatile("s_mov_b32 %2, m0\n\ts_set_gpr_idx_on %5, 0xa\n\tv_add_f32 v64, %3, v64\n\tv_add_f32 v65, %4, v65\n\t"
                             "s_set_gpr_idx_off\n\ts_mov_b32 m0, %2"
                             : PIN, "=&s"(keep) : "v"(v[0]), "v"(v[1]), "s"(2 * j));
            else
                asm volatile("s_mov_b32 %2, m0\n\ts_set_gpr_idx_on %4, 0xa\n\tv_pk_add_f32 v[64:65], %3, v[64:65]\n\t"
                             "s_set_gpr_idx_off\n\ts_mov_b32 m0, %2"
                             : PIN, "=&s"(keep) : "v"(v), "s"(2 * j));
        }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("" : PIN);
    for (int i = 0; i < 32; ++i) { out[i * 64 + lane] = a0[i]; out[(32 + i) * 64 + lane] = a1[i]; }
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    const int n = 500, reps = 4;
    std::vector<int> idx(n); std::vector<float> val(n * 128), ref(64 * 64, 0.f);
    for (int i = 0; i < n; ++i) {
        idx[i] = (i * 37 + 5) % 32;
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 2; ++e) {
            val[i * 128 + 2 * l + e] = (float)((i * 7 + l + 3 * e) % 13) - 6.f;
            ref[(2 * idx[i] + e) * 64 + l] += reps * val[i * 128 + 2 * l + e];
        }
    }
    int* di; float *dv, *dout; long long* dc;
    (void)hipMalloc(&di, n * 4); (void)hipMalloc(&dv, n * 128 * 4); (void)hipMalloc(&dout, 64 * 64 * 4); (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(di, idx.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dv, val.data(), n * 128 * 4, hipMemcpyHostToDevice);
    int rc = 0;
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) k<0><<<1, 64>>>(di, dv, n, reps, dout, dc); else k<1><<<1, 64>>>(di, dv, n, reps, dout, dc);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        std::vector<float> out(64 * 64); long long c;
        (void)hipMemcpy(out.data(), dout, 64 * 64 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < 64 * 64; ++i) if (out[i] != ref[i]) { if (bad++ < 5) printf("reg %d lane %d: got %g want %g\n", i / 64, i % 64, out[i], ref[i]); }
        printf("gpr idx probe mode %d (%s): %ld wrong of %d; %.1f cycles per indexed pair add (incl. its load)\n", mode,
               mode ? "v_pk_add_f32" : "2 x v_add_f32", bad, 64 * 64, (double)c / (n * reps));
        rc |= bad != 0;
    }
    return rc;
}
