// What ds_read_b64_tr_b16 hands to each lane: a [64 rows][64 columns] image of 16-bit elements whose value is its own
// linear index; lane 4q+p of 16-lane group g supplies the address of row 8g+q, columns 4p..4p+3.  Prints (row, col) of the
// four elements every lane receives.   hipcc --offload-arch=gfx950 -O2 -o tr_probe tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ unsigned short t[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) t[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    auto ptr = (__attribute__((address_space(3))) s16x4*)(t + (8 * g + q) * 64 + 4 * p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = (unsigned short)v[i];
}
int main() {
    unsigned short* d;
    if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) return 1;
    k<<<1, 64>>>(d);
    unsigned short h[256];
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int i = 0; i < 4; ++i) printf(" (r%2d,c%2d)", h[l * 4 + i] / 64, h[l * 4 + i] % 64);
        printf("\n");
    }
    return 0;
}
