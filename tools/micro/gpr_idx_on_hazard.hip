// s_set_gpr_idx_on followed at once by a VALU instruction in the indexing mode: is the first instruction always indexed?
// (Round 4: the root cause of message_bx.hip's hidden-64 hazard at four waves per SIMD.)
//   hipcc --offload-arch=gfx950 -O2 -o gpr_idx_on_hazard gpr_idx_on_hazard.hip && ./gpr_idx_on_hazard
// Every wave keeps 48 counters in v64..v111 and a canary block in v20..v51.  Per iteration: M0 <- an LDS-address-like value
// (what an LDS-DMA leaves there), then  s_set_gpr_idx_on idx, 0xa ; [PAD wait states] ; v_add_f32 v64, 1.0, v64 ;
// s_set_gpr_idx_off ; with idx walking over 0..47.  After ITER iterations counter n must hold the number of iterations with
// idx == n and the canaries must be untouched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4800

template <int PAD, int BUSY>
__global__ __launch_bounds__(1024) void probe(int* bad_counts, int* bad_canary, float* dump) {
    const int lane = threadIdx.x & 63;
    float one = 1.0f;
    int errs = 0, cerr = 0;
    // counters and canaries in fixed registers
    asm volatile(".irp r,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111\n\t"
                 "v_mov_b32 v\\r, 0\n\t.endr\n\t"
                 ".irp r,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51\n\t"
                 "v_mov_b32 v\\r, 0x7777\n\t.endr\n\t" ::: "memory",
                 "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51",
                 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95",
                 "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111");
    float filler = (float)lane;
    for (int it = 0; it < ITER; ++it) {
        const int idx = __builtin_amdgcn_readfirstlane((it * 7 + (int)blockIdx.x) % 48);
        const int junk = __builtin_amdgcn_readfirstlane(0x2400 * (1 + (it & 7)));     // 0x2400 .. 0x12000: bits 12..15 vary, bits 0..7 zero
        int keep;
        if (BUSY) {                                                       // other VALU work of this wave around the switch
            filler = filler * 1.0001f + 0.5f;
            filler = filler * 0.9999f - 0.25f;
        }
        asm volatile("s_mov_b32 %[kp], m0\n\ts_mov_b32 m0, %[jk]\n\ts_nop 3\n\t"
                     "s_set_gpr_idx_on %[ix], 0xa\n\t"
                     ".rept %c[pad]\n\ts_nop 0\n\t.endr\n\t"
                     "v_add_f32 v64, %[one], v64\n\t"
                     "s_set_gpr_idx_off\n\ts_mov_b32 m0, %[kp]"
                     : [kp] "=&s"(keep) : [ix] "s"(idx), [jk] "s"(junk), [one] "v"(one), [pad] "n"(PAD)
                     : "memory", "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95",
                       "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111",
                       "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51");
    }
    // expected count of idx n for this block
    int expect[48];
    for (int n = 0; n < 48; ++n) expect[n] = 0;
    for (int it = 0; it < ITER; ++it) expect[(it * 7 + (int)blockIdx.x) % 48]++;
#define CHK(n) { float c; asm volatile("v_mov_b32 %0, v" #n : "=v"(c)); if (c != (float)expect[n - 64]) { ++errs; if (lane == 0 && (threadIdx.x >> 6) == 0 && blockIdx.x == 0) dump[n - 64] = c; } }
    CHK(64) CHK(65) CHK(66) CHK(67) CHK(68) CHK(69) CHK(70) CHK(71) CHK(72) CHK(73) CHK(74) CHK(75) CHK(76) CHK(77) CHK(78) CHK(79)
    CHK(80) CHK(81) CHK(82) CHK(83) CHK(84) CHK(85) CHK(86) CHK(87) CHK(88) CHK(89) CHK(90) CHK(91) CHK(92) CHK(93) CHK(94) CHK(95)
    CHK(96) CHK(97) CHK(98) CHK(99) CHK(100) CHK(101) CHK(102) CHK(103) CHK(104) CHK(105) CHK(106) CHK(107) CHK(108) CHK(109) CHK(110) CHK(111)
#define CAN(n) { int c; asm volatile("v_mov_b32 %0, v" #n : "=v"(c)); if (c != 0x7777) ++cerr; }
    CAN(20) CAN(21) CAN(22) CAN(23) CAN(24) CAN(25) CAN(26) CAN(27) CAN(28) CAN(29) CAN(30) CAN(31) CAN(32) CAN(33) CAN(34) CAN(35)
    CAN(36) CAN(37) CAN(38) CAN(39) CAN(40) CAN(41) CAN(42) CAN(43) CAN(44) CAN(45) CAN(46) CAN(47) CAN(48) CAN(49) CAN(50) CAN(51)
    if (filler == 123456.f) errs += 1000;
    if (errs) atomicAdd(bad_counts, 1);
    if (cerr) atomicAdd(bad_canary, 1);
}

template <int PAD, int BUSY>
static void run(const char* what, int blocks, int threads) {
    int *bc, *bn;
    float* dump;
    hipMalloc(&bc, 4); hipMalloc(&bn, 4); hipMalloc(&dump, 48 * 4);
    hipMemset(bc, 0, 4); hipMemset(bn, 0, 4); hipMemset(dump, 0, 48 * 4);
    probe<PAD, BUSY><<<blocks, threads>>>(bc, bn, dump);
    hipDeviceSynchronize();
    int a = 0, b = 0;
    hipMemcpy(&a, bc, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&b, bn, 4, hipMemcpyDeviceToHost);
    printf("%-64s lanes with a wrong counter: %d, with a touched canary: %d (of %d)\n", what, a, b, blocks * threads);
    hipFree(bc); hipFree(bn); hipFree(dump);
}

int main() {
    run<0, 0>("no wait state, 1 wave per CU", 256, 64);
    run<0, 0>("no wait state, 4 waves per SIMD (16 per CU)", 1024, 1024);
    run<0, 1>("no wait state, 4 waves per SIMD, VALU work around the switch", 1024, 1024);
    run<1, 1>("1 wait state, 4 waves per SIMD, VALU work around", 1024, 1024);
    run<2, 1>("2 wait states", 1024, 1024);
    run<4, 1>("4 wait states", 1024, 1024);
    run<8, 1>("8 wait states", 1024, 1024);
    return 0;
}
