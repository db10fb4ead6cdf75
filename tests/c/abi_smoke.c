/* A C99 translation unit that uses include/ghf.h as a plain C header and drives libghf_hip.so through dlopen — what a
 * cgo / JNI / FFI binding of the boundary does.  Host-side entry points only (no kernel launch): runs without a GPU.
 * usage: abi_smoke <path to libghf_hip.so>; prints "ok <abi version>" */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "ghf.h"

#define LOAD(name)                                                             \
    do {                                                                       \
        *(void**)(&p_##name) = dlsym(lib, #name);                              \
        if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    /* pointers typed from the header's own declarations: a signature drift is a compile error here */
    __typeof__(&ghf_abi_version) p_ghf_abi_version;
    __typeof__(&ghf_last_error) p_ghf_last_error;
    __typeof__(&ghf_message_config) p_ghf_message_config;
    __typeof__(&ghf_plan_workspace_bytes) p_ghf_plan_workspace_bytes;
    __typeof__(&ghf_weights_bytes) p_ghf_weights_bytes;
    __typeof__(&ghf_split_rows_bytes) p_ghf_split_rows_bytes;
    __typeof__(&ghf_tail_fwd) p_ghf_tail_fwd;
    __typeof__(&ghf_edge_outer_supported) p_ghf_edge_outer_supported;
    LOAD(ghf_abi_version); LOAD(ghf_last_error); LOAD(ghf_message_config); LOAD(ghf_plan_workspace_bytes);
    LOAD(ghf_weights_bytes); LOAD(ghf_split_rows_bytes); LOAD(ghf_tail_fwd); LOAD(ghf_edge_outer_supported);

    if (p_ghf_abi_version() != GHF_ABI_VERSION) { fprintf(stderr, "header %d != library %d\n", GHF_ABI_VERSION, p_ghf_abi_version()); return 3; }
    int bn = 0, wl = -1, cr = 0, sc = 0;
    if (p_ghf_message_config(128, &bn, &wl, &cr, &sc) != 0 || bn <= 1 || cr <= 0) return 4;
    if (p_ghf_message_config(20, &bn, &wl, &cr, &sc) != 0 || bn != 1 || wl != GHF_WLAYOUT_NATURAL) return 5;
    if (p_ghf_plan_workspace_bytes(1000, 5000, 7, 216, 48) == 0) return 6;
    if (p_ghf_weights_bytes(4, 8, 24, GHF_WLAYOUT_NATURAL) != (size_t)4 * 8 * 24 * 4) return 7;
    if (p_ghf_split_rows_bytes(10, 128, GHF_WLAYOUT_SPLIT2H) != (size_t)10 * 128 * 4 + 40) return 8;
    if (!p_ghf_edge_outer_supported(128) || p_ghf_edge_outer_supported(20)) return 9;
    /* argument errors come back as a code plus a thread-local message, never as a fault */
    if (p_ghf_tail_fwd(NULL, NULL, NULL, NULL, 1e-5f, 0, 1, 8, NULL, NULL, NULL) != -1) return 10;
    if (!strstr(p_ghf_last_error(), "null")) return 11;
    printf("ok %d\n", p_ghf_abi_version());
    dlclose(lib);
    return 0;
}
