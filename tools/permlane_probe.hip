// Development probe: semantics of v_permlane16_swap / v_permlane32_swap (gfx950) as exposed by the builtins.
// Build: hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o tools/permlane_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const unsigned a = l, b = 100 + l;
    u32x2 r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    u32x2 r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l * 4 + 0] = r16[0];
    out[l * 4 + 1] = r16[1];
    out[l * 4 + 2] = r32[0];
    out[l * 4 + 3] = r32[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64 * 4 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int c = 0; c < 4; ++c) {
        printf("%s:", c == 0 ? "p16[0]" : c == 1 ? "p16[1]" : c == 2 ? "p32[0]" : "p32[1]");
        for (int l = 0; l < 64; ++l) printf(" %u", h[l * 4 + c]);
        printf("\n");
    }
    return 0;
}
