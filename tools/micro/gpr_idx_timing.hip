// Timing probe (gfx950): cycles per row of the indexed-add sequences message_bx.hip's fold can use — data in registers,
// one wave, 4 rows per asm block:  A: on/add/add/off per row;  B: one on, s_set_gpr_idx_idx between rows, one off;
// C: as A with v_pk_add_f32;  D: as B with v_pk_add_f32.   Also checks B / D give the sums A gives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PIN "+{v[64:95]}"(a0), "+{v[96:127]}"(a1)
template <int MODE>
__global__ __launch_bounds__(64) void k(const int* __restrict__ idx, int n, float* __restrict__ out, long long* __restrict__ cyc) {
    const int lane = threadIdx.x;
    f32x32 a0, a1;
    for (int i = 0; i < 32; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    asm volatile("" : PIN);
    f32x2 y[4];
    for (int i = 0; i < 4; ++i) y[i] = (f32x2){(float)(lane + i), (float)(2 * lane - i)};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < n; g += 4) {
        const int i0 = __builtin_amdgcn_readfirstlane(idx[g]) * 2, i1 = __builtin_amdgcn_readfirstlane(idx[g + 1]) * 2,
                  i2 = __builtin_amdgcn_readfirstlane(idx[g + 2]) * 2, i3 = __builtin_amdgcn_readfirstlane(idx[g + 3]) * 2;
        int keep;
        if (MODE == 0)
            asm volatile("s_mov_b32 %[kp], m0\n\t"
                         "s_set_gpr_idx_on %[i0], 0xa\n\tv_add_f32 v64, %[a0], v64\n\tv_add_f32 v65, %[b0], v65\n\ts_set_gpr_idx_off\n\t"
                         "s_set_gpr_idx_on %[i1], 0xa\n\tv_add_f32 v64, %[a1], v64\n\tv_add_f32 v65, %[b1], v65\n\ts_set_gpr_idx_off\n\t"
                         "s_set_gpr_idx_on %[i2], 0xa\n\tv_add_f32 v64, %[a2], v64\n\tv_add_f32 v65, %[b2], v65\n\ts_set_gpr_idx_off\n\t"
                         "s_set_gpr_idx_on %[i3], 0xa\n\tv_add_f32 v64, %[a3], v64\n\tv_add_f32 v65, %[b3], v65\n\ts_set_gpr_idx_off\n\t"
                         "s_mov_b32 m0, %[kp]"
                         : PIN, [kp] "=&s"(keep) : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3),
                           [a0] "v"(y[0][0]), [b0] "v"(y[0][1]), [a1] "v"(y[1][0]), [b1] "v"(y[1][1]), [a2] "v"(y[2][0]), [b2] "v"(y[2][1]), [a3] "v"(y[3][0]), [b3] "v"(y[3][1]));
        else if (MODE == 1)
            asm volatile("s_mov_b32 %[kp], m0\n\t"
                         "s_set_gpr_idx_on %[i0], 0xa\n\tv_add_f32 v64, %[a0], v64\n\tv_add_f32 v65, %[b0], v65\n\t"
                         "s_set_gpr_idx_idx %[i1]\n\tv_add_f32 v64, %[a1], v64\n\tv_add_f32 v65, %[b1], v65\n\t"
                         "s_set_gpr_idx_idx %[i2]\n\tv_add_f32 v64, %[a2], v64\n\tv_add_f32 v65, %[b2], v65\n\t"
                         "s_set_gpr_idx_idx %[i3]\n\tv_add_f32 v64, %[a3], v64\n\tv_add_f32 v65, %[b3], v65\n\ts_set_gpr_idx_off\n\t"
                         "s_mov_b32 m0, %[kp]"
                         : PIN, [kp] "=&s"(keep) : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3),
                           [a0] "v"(y[0][0]), [b0] "v"(y[0][1]), [a1] "v"(y[1][0]), [b1] "v"(y[1][1]), [a2] "v"(y[2][0]), [b2] "v"(y[2][1]), [a3] "v"(y[3][0]), [b3] "v"(y[3][1]));
        else if (MODE == 2)
            asm volatile("s_mov_b32 %[kp], m0\n\t"
                         "s_set_gpr_idx_on %[i0], 0xa\n\tv_pk_add_f32 v[64:65], %[a0], v[64:65]\n\ts_set_gpr_idx_off\n\t"
                         "s_set_gpr_idx_on %[i1], 0xa\n\tv_pk_add_f32 v[64:65], %[a1], v[64:65]\n\ts_set_gpr_idx_off\n\t"
                         "s_set_gpr_idx_on %[i2], 0xa\n\tv_pk_add_f32 v[64:65], %[a2], v[64:65]\n\ts_set_gpr_idx_off\n\t"
                         "s_set_gpr_idx_on %[i3], 0xa\n\tv_pk_add_f32 v[64:65], %[a3], v[64:65]\n\ts_set_gpr_idx_off\n\t"
                         "s_mov_b32 m0, %[kp]"
                         : PIN, [kp] "=&s"(keep) : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3),
                           [a0] "v"(y[0]), [a1] "v"(y[1]), [a2] "v"(y[2]), [a3] "v"(y[3]));
        else
            asm volatile("s_mov_b32 %[kp], m0\n\t"
                         "s_set_gpr_idx_on %[i0], 0xa\n\tv_pk_add_f32 v[64:65], %[a0], v[64:65]\n\t"
                         "s_set_gpr_idx_idx %[i1]\n\tv_pk_add_f32 v[64:65], %[a1], v[64:65]\n\t"
                         "s_set_gpr_idx_idx %[i2]\n\tv_pk_add_f32 v[64:65], %[a2], v[64:65]\n\t"
                         "s_set_gpr_idx_idx %[i3]\n\tv_pk_add_f32 v[64:65], %[a3], v[64:65]\n\ts_set_gpr_idx_off\n\t"
                         "s_mov_b32 m0, %[kp]"
                         : PIN, [kp] "=&s"(keep) : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3),
                           [a0] "v"(y[0]), [a1] "v"(y[1]), [a2] "v"(y[2]), [a3] "v"(y[3]));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("" : PIN);
    for (int i = 0; i < 32; ++i) { out[i * 64 + lane] = a0[i]; out[(32 + i) * 64 + lane] = a1[i]; }
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    const int n = 4096;
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = (i * 37 + 5) % 32;
    int* di; float* dout; long long* dc;
    (void)hipMalloc(&di, n * 4); (void)hipMalloc(&dout, 64 * 64 * 4); (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(di, idx.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> ref;
    int rc = 0;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k<0><<<1, 64>>>(di, n, dout, dc); else if (mode == 1) k<1><<<1, 64>>>(di, n, dout, dc);
            else if (mode == 2) k<2><<<1, 64>>>(di, n, dout, dc); else k<3><<<1, 64>>>(di, n, dout, dc);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        }
        std::vector<float> out(64 * 64); long long c;
        (void)hipMemcpy(out.data(), dout, 64 * 64 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        if (mode == 0) ref = out;
        long bad = 0; for (int i = 0; i < 64 * 64; ++i) bad += out[i] != ref[i];
        printf("mode %d: %.1f cycles per row (incl. the index loads), %ld sums differ from mode 0\n", mode, (double)c / n, bad);
        rc |= bad != 0;
    }
    return rc;
}
