// ds_read_b64_tr_b16 on the XOR-swizzled [32 rows][128 columns] image of backward.hip (edge_outer_h_kernel): element (r, c) holds
// r * 128 + c; every lane reads the fragment of columns fb .. fb+15 as the kernel does and prints what it got.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ unsigned eo_off(int row, int ch) { return 256u * row + 16u * (unsigned)(ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__global__ void k(unsigned short* out, int fb) {
    __shared__ __attribute__((aligned(16))) char img[32 * 256];
    for (int i = threadIdx.x; i < 32 * 128; i += 64) {
        const int r = i / 128, c = i % 128;
        *(unsigned short*)(img + eo_off(r, c >> 3) + 2 * (c & 7)) = (unsigned short)i;
    }
    __syncthreads();
    const int lane = threadIdx.x, Q = lane >> 4, c16 = lane & 15, gq = c16 >> 2, gp = c16 & 3;
    const int ch = (fb >> 3) + (gp >> 1);
    auto p0 = (__attribute__((address_space(3))) s16x4*)(img + eo_off(8 * Q + gq, ch) + 8u * (gp & 1));
    auto p1 = (__attribute__((address_space(3))) s16x4*)(img + eo_off(8 * Q + 4 + gq, ch) + 8u * (gp & 1));
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
    for (int i = 0; i < 4; ++i) { out[lane * 8 + i] = (unsigned short)a[i]; out[lane * 8 + 4 + i] = (unsigned short)b[i]; }
}
int main() {
    unsigned short* d;
    if (hipMalloc(&d, 64 * 8 * 2) != hipSuccess) return 1;
    for (int fb : {0, 16, 48}) {
        k<<<1, 64>>>(d, fb);
        unsigned short h[512];
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                const int want = (8 * (l >> 4) + e) * 128 + fb + (l & 15);
                if (h[l * 8 + e] != want) { if (bad < 6) printf("fb %d lane %d e %d: got (r%d,c%d) want (r%d,c%d)\n", fb, l, e, h[l*8+e] / 128, h[l*8+e] % 128, want / 128, want % 128); ++bad; }
            }
        printf("fb %d: %d of 512 elements wrong\n", fb, bad);
    }
    return 0;
}
