// Does the VGPR indexing mode (s_set_gpr_idx_on, VDST_REL) displace the destination of an LDS read that was issued BEFORE
// the mode was switched on and returns while it is on?  (Round 4: the hidden-64 hazard of message_bx.hip.)
//   hipcc --offload-arch=gfx950 -O2 -o gpr_idx_lds_probe gpr_idx_lds_probe.hip && ./gpr_idx_lds_probe
// Per lane: v10 = 0x222, v40 = 0x111 (= v10 + 30), a queue of ds_read_b128 to make the LDS slow, then ds_read_b32 v10,
// then the mode goes on with index 30 and mode bits `mode`; the wave waits for the read INSIDE the mode window, switches the
// mode off and reports v10 and v40.  Undisturbed: v10 = the LDS word, v40 = 0x111.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int NQ, int VMEM>
__global__ void probe(int* out, const int* gsrc) {
    __shared__ int lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1000 + i;
    __syncthreads();
    const unsigned addr = (unsigned)(threadIdx.x & 63) * 4u + 4096u * (threadIdx.x >> 6);
    const unsigned qaddr = (unsigned)(threadIdx.x & 63) * 16u;
    int a, b, keep;
    const int idx = 30;
    if (VMEM) {
        asm volatile(
            "v_mov_b32 v10, 0x222\n\tv_mov_b32 v40, 0x111\n\t"
            "global_load_dword v10, %[ga], off\n\t"
            "s_mov_b32 %[kp], m0\n\ts_set_gpr_idx_on %[ix], %[md]\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "s_nop 15\n\t"
            "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[kp]\n\t"
            "v_mov_b32 %[a], v10\n\tv_mov_b32 %[b], v40\n\t"
            : [a] "=&v"(a), [b] "=&v"(b), [kp] "=&s"(keep)
            : [ga] "v"(gsrc + threadIdx.x), [ix] "s"(idx), [md] "n"(MODE)
            : "v10", "v40", "memory");
    } else {
        asm volatile(
            "v_mov_b32 v10, 0x222\n\tv_mov_b32 v40, 0x111\n\t"
            ".rept %c[nq]\n\tds_read_b128 v[20:23], %[qa]\n\tds_read_b128 v[24:27], %[qa] offset:1024\n\t.endr\n\t"
            "ds_read_b32 v10, %[ad]\n\t"
            "s_mov_b32 %[kp], m0\n\ts_set_gpr_idx_on %[ix], %[md]\n\t"
            "s_nop 7\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 7\n\t"
            "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[kp]\n\t"
            "v_mov_b32 %[a], v10\n\tv_mov_b32 %[b], v40\n\t"
            : [a] "=&v"(a), [b] "=&v"(b), [kp] "=&s"(keep)
            : [ad] "v"(addr), [qa] "v"(qaddr), [ix] "s"(idx), [md] "n"(MODE), [nq] "n"(NQ)
            : "v10", "v40", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
    }
    out[2 * (blockIdx.x * blockDim.x + threadIdx.x)] = a;
    out[2 * (blockIdx.x * blockDim.x + threadIdx.x) + 1] = b;
}

template <int MODE, int NQ, int VMEM>
static void run(const char* what, int blocks, int threads) {
    int *out, *src;
    const int n = blocks * threads;
    hipMalloc(&out, n * 2 * sizeof(int));
    hipMalloc(&src, n * sizeof(int));
    std::vector<int> hs(n);
    for (int i = 0; i < n; ++i) hs[i] = 5000 + i;
    hipMemcpy(src, hs.data(), n * sizeof(int), hipMemcpyHostToDevice);
    hipMemset(out, 0, n * 2 * sizeof(int));
    probe<MODE, NQ, VMEM><<<blocks, threads>>>(out, src);
    hipDeviceSynchronize();
    std::vector<int> h(n * 2);
    hipMemcpy(h.data(), out, n * 2 * sizeof(int), hipMemcpyDeviceToHost);
    long in_place = 0, displaced = 0, other = 0;
    for (int i = 0; i < n; ++i) {
        const int a = h[2 * i], b = h[2 * i + 1];
        const int t = i % threads;
        const int want = VMEM ? 5000 + t + (i / threads) * 0 : 1000 + (t & 63) + 1024 * (t >> 6);
        const bool ok_val = VMEM ? (a >= 5000) : (a == want);
        if (ok_val && b == 0x111) ++in_place;
        else if (a == 0x222 && b != 0x111) ++displaced;
        else ++other;
    }
    printf("%-58s lanes: %ld in place, %ld displaced to v40, %ld other\n", what, in_place, displaced, other);
    hipFree(out);
    hipFree(src);
}

int main() {
    run<0x8, 0, 0>("LDS read, VDST_REL, idle LDS, 1 wave", 1, 64);
    run<0x8, 8, 0>("LDS read, VDST_REL, 16 b128 reads queued ahead, 1 wave", 1, 64);
    run<0x8, 8, 0>("LDS read, VDST_REL, queued, 16 waves per CU everywhere", 1024, 1024);
    run<0xa, 8, 0>("LDS read, VDST_REL|VSRC1_REL (the fold's mode), busy", 1024, 1024);
    run<0x2, 8, 0>("LDS read, VSRC1_REL only, busy", 1024, 1024);
    run<0x8, 0, 1>("global load, VDST_REL, 1 wave", 1, 64);
    run<0x8, 0, 1>("global load, VDST_REL, busy", 1024, 1024);
    return 0;
}
