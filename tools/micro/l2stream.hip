// Diagnostic microbenchmark: how fast can one CU stream L2-resident bytes into registers?
// Each workgroup (WAVES waves) reads a `span`-byte window of a 12 MiB buffer over and over with 16-byte-per-lane buffer
// loads, DEPTH wave-loads in flight per wave, for several footprints (L2 4 MiB per XCD, Infinity Cache 256 MiB, HBM).  Prints bytes per clock per CU (2.4 GHz nominal; see the printed ms too).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2stream.hip -o gpurun_out/l2stream && gpurun_out/l2stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const void* buf, uint32_t bytes, int iters, int span_kb, int* sink) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // every workgroup walks windows of span_kb KiB; window start depends on the block so that blocks share L2 lines loosely
    i32x4 acc = {0, 0, 0, 0};
    i32x4 r[DEPTH];
    const int per_wave = span_kb * 1024 / nw;          // bytes of the window this wave reads
    const int steps = per_wave / 1024;                 // 1 KiB wave-loads per pass
    for (int it = 0; it < iters; ++it) {
        const int win = (int)(((blockIdx.x * 7 + it) * (uint32_t)span_kb * 1024u) % (bytes - span_kb * 1024u)) & ~1023;
        const int base = __builtin_amdgcn_readfirstlane(win + w * per_wave);
        for (int s0 = 0; s0 < steps; s0 += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, base + (s0 + k) * 1024, 0);
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) acc += r[k];
        }
    }
    if (acc[0] == 0x12345678) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int DEPTH>
static void run(const void* buf, uint32_t bytes, int waves, int span_kb, int* sink) {
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    stream_kernel<DEPTH><<<256, waves * 64, 0, 0>>>(buf, bytes, 10, span_kb, sink);
    hipEventRecord(a);
    stream_kernel<DEPTH><<<256, waves * 64, 0, 0>>>(buf, bytes, iters, span_kb, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double total = 256.0 * iters * span_kb * 1024.0;
    printf("waves=%d depth=%2d span=%3d KiB: %7.3f ms  %6.2f TB/s chip  %5.1f GB/s per CU  %5.1f B/clk/CU @2.4GHz\n", waves, DEPTH, span_kb,
           ms, total / ms / 1e9, total / 256 / ms / 1e6, total / 256 / (ms * 1e-3) / 2.4e9);
}

int main() {
    int* sink; hipMalloc(&sink, 4);
    for (uint32_t mb : {2u, 4u, 12u, 32u, 128u, 1024u}) {
        const uint32_t bytes = mb << 20;
        void* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
        printf("footprint %u MiB\n", mb);
        for (int waves : {4, 8}) {
            run<4>(buf, bytes, waves, 96, sink);
            run<12>(buf, bytes, waves, 96, sink);
        }
        hipFree(buf);
    }
    return 0;
}
