// LDS-DMA semantics check (gfx950): buffer_load_dwordx4 ... lds with per-lane source offsets — where do the 64 x 16 bytes of one
// wave instruction land, at LDS addresses below and above 64 KiB, and what do out-of-range lanes write?
//   hipcc --offload-arch=gfx950 -O3 -o dma_gather dma_gather.hip && ./dma_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int ROWS = 76, ROWB = 256, PLANE = ROWS * ROWB, NT = 4;     // four tiles of two planes, as message_bx.hip
__global__ __launch_bounds__(256) void k(const char* __restrict__ table, int table_bytes, const int* __restrict__ ids, int rows,
                                         int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < NT * 2 * PLANE / 4; i += 256) ((int*)smem)[i] = 0x7fc00000 + i;   // poison
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, table_bytes, 0x00020000);
    for (int t = 0; t < NT; ++t)
        for (int i = 0; i < 5; ++i) {
            const int rb = hw + 4 * i, row = 4 * rb + (lane >> 4);
            if (rb >= ROWS / 4) continue;
            const int g = (lane & 15) ^ (row & 15);
            const int voff = row < rows ? ids[row] * 512 + g * 16 : 0x7FFFFF00;
            for (int pl = 0; pl < 2; ++pl) {
                const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(t * 2 * PLANE + pl * PLANE + rb * 1024));
                if (t & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + d), 16, voff + pl * ROWB, 0, 0, 2);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + d), 16, voff + pl * ROWB, 0, 0, 0);
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < NT * 2 * PLANE / 4; i += 256) out[i] = ((const int*)smem)[i];
}
int main() {
    const int N = 1000, rows = 61;
    std::vector<int> table(N * 128), ids(80);
    for (int n = 0; n < N; ++n) for (int j = 0; j < 128; ++j) table[n * 128 + j] = n * 1000 + j;    // word j of node n
    for (int r = 0; r < 80; ++r) ids[r] = (r * 37 + 11) % N;
    int *dt, *di, *dout;
    const size_t ob = (size_t)NT * 2 * PLANE;
    hipMalloc(&dt, table.size() * 4); hipMalloc(&di, 80 * 4); hipMalloc(&dout, ob);
    hipMemcpy(dt, table.data(), table.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(di, ids.data(), 80 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ob);
    k<<<1, 256, ob>>>((const char*)dt, N * 512, di, rows, dout);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<int> out(ob / 4);
    hipMemcpy(out.data(), dout, ob, hipMemcpyDeviceToHost);
    long bad = 0, dead_zero = 0, dead_poison = 0, dead_other = 0;
    for (int t = 0; t < NT; ++t) for (int pl = 0; pl < 2; ++pl) for (int row = 0; row < ROWS; ++row) for (int s = 0; s < 16; ++s) for (int e = 0; e < 4; ++e) {
        const size_t w = ((size_t)(t * 2 + pl) * PLANE + row * ROWB + s * 16) / 4 + e;
        const int got = out[w];
        if (row < rows) {
            const int g = s ^ (row & 15), want = ids[row] * 1000 + pl * 64 + g * 4 + e;
            if (got != want && bad++ < 8) printf("tile %d plane %d row %d slot %d word %d: got %d want %d\n", t, pl, row, s, e, got, want);
        } else if (got == 0) ++dead_zero; else if (got == (int)(0x7fc00000 + w)) ++dead_poison; else ++dead_other;
    }
    printf("live words wrong: %ld; dead-row words: zero %ld, untouched %ld, other %ld\n", bad, dead_zero, dead_poison, dead_other);
    return bad != 0;
}
