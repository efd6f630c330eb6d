// Development probe: do bf16 MFMAs (v_mfma_f32_16x16x32_bf16) and fp32 VALU work overlap on a SIMD?
//   mode 0: 4 waves, MFMA only           mode 1: 4 waves, VALU only
//   mode 2: 8 waves, 4 MFMA + 4 VALU     mode 3: 4 waves, each interleaving both streams (same wave)
//   mode 4: 8 waves, all interleaving (two mixed waves per SIMD)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSCALAR_VALU] tools/mfma_probe_bf16.hip -o /tmp/probe_bf16
// (plain -O3 turns the fmaf chain into v_pk_fma_f32; -DSCALAR_VALU pins it to v_fma_f32)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV>
__device__ __forceinline__ void body(f32x4 (&acc)[8], float (&v)[16], bf16x8 a, bf16x8 b, int rounds) {
    const float fa = 1.0001f, fc = 0.0001f;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (NM) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
            }
            if (NV) {
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
#ifdef SCALAR_VALU   // keep hipcc from SLP-packing the chain into v_pk_fma_f32
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(fa), "v"(fc));
#else
                        v[i] = fmaf(v[i], fa, fc);
#endif
                    }
            }
        }
    }
}

__global__ __launch_bounds__(512) void probe(float* __restrict__ out, int rounds, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
    // per k: 8 MFMAs = 128 matrix cycles; NV=2 -> 32 v_fma = 128 VALU cycles (4 cycles per wave64 fp32 op)
    if (mode == 0 || (mode == 2 && wave < 4)) body<1, 0>(acc, v, a, b, rounds);
    else if (mode == 1 || mode == 2) body<0, 2>(acc, v, a, b, rounds);
    else body<1, 2>(acc, v, a, b, rounds);
    float res = 0.f;
    for (int t = 0; t < 8; ++t) res += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < 16; ++i) res += v[i];
    out[blockIdx.x * blockDim.x + tid] = res;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rounds = 2000;
    const char* names[5] = {"4 MFMA waves", "4 VALU waves", "4 MFMA + 4 VALU waves", "4 mixed waves", "8 mixed waves"};
    for (int mode = 0; mode < 5; ++mode) {
        const int nw = (mode == 2 || mode == 4) ? 8 : 4;
        probe<<<256, nw * 64>>>(out, rounds, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<256, nw * 64>>>(out, rounds, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-24s: %.3f ms  (%d rounds x 8 x {8 MFMA | 32 v_fma} per wave)\n", names[mode], ms, rounds);
    }
    return 0;
}
