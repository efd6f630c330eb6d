// Development probe: which lane does a DPP row rotation read?  (row_ror:n on gfx950; used by row16_reduce16, mfma_tile.h)
// Build: hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip -o tools/dpp_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x124, 0xf, 0xf, false);        // row_ror:4
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x12c, 0xf, 0xf, false);   // row_ror:12
    out[128 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x128, 0xf, 0xf, false);  // row_ror:8
    out[192 + l] = __builtin_amdgcn_update_dpp(-1, __builtin_amdgcn_update_dpp(-1, l, 0x141, 0xf, 0xf, false), 0x1b, 0xf, 0xf, false);
    out[256 + l] = __builtin_amdgcn_update_dpp(-1, l, 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    out[320 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x4e, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
}
int main() {
    int* d; hipMalloc(&d, 384 * 4);
    probe<<<1, 64>>>(d);
    int h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* n[6] = {"row_ror:4", "row_ror:12", "row_ror:8", "half_mirror+quad_rev", "quad[1,0,3,2]", "quad[2,3,0,1]"};
    for (int k = 0; k < 6; ++k) { printf("%-22s", n[k]); for (int l = 0; l < 20; ++l) printf(" %2d", h[64 * k + l]); printf("\n"); }
    return 0;
}
