// Development probe: does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs (needed by the two-piece fp16 split:
// the lo piece of a value below 0.25 is an fp16 subnormal), and does v_cvt_pk_f16_f32 produce them?
// Build: hipcc --offload-arch=gfx950 -O3 tools/f16_denorm_probe.hip -o tools/f16_denorm_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void probe(float* out, float tiny, float big) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    f16x2 pk;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(tiny), "v"(tiny * 3.0f));
    a[0] = pk[0];           // subnormal fp16 (if the conversion keeps it)
    a[1] = pk[1];
    b[0] = (_Float16)big;
    b[1] = (_Float16)big;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (lane == 0) {
        out[0] = (float)pk[0];
        out[1] = (float)pk[1];
        out[2] = c[0];
    }
}

int main() {
    float* out;
    (void)hipMalloc(&out, 64);
    const float tiny = 9.5367431640625e-07f;  // 2^-20: fp16 subnormal (smallest normal 2^-14)
    const float big = 1024.f;
    probe<<<1, 64>>>(out, tiny, big);
    float h[3];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    // each lane supplies k-slots; lane 0..15 x 4 groups all hold the same a/b -> c[0] = sum over the 4 lane groups of
    // (tiny*big + 3*tiny*big) = 4 * 4 * tiny * big
    printf("cvt_pk_f16_f32(2^-20) = %g (expect 9.53674e-07), (3*2^-20) = %g\n", h[0], h[1]);
    printf("mfma f16 with subnormal A: %g (expect %g if kept, 0 if flushed)\n", h[2], 16.f * tiny * big);
    return 0;
}
