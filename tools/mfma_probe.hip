// Development probe: sustained v_mfma_f32_16x16x4_f32 rate of the ksteps<> structure (LDS-fed A
// operand) with 1 or 2 wavefronts per SIMD and no global traffic.  Build: hipcc --offload-arch=gfx950
// -O3 -std=c++17 -I include -I tsp-gnn_amd/csrc tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "mfma_tile.h"
using namespace tspgnn;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, float* __restrict__ out, int rounds) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4;
    copy_to_lds(lds, w, 64 * 256, tid, blockDim.x);
    __syncthreads();
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float b[16];
    for (int i = 0; i < 16; ++i) b[i] = 0.001f * (lane + i);
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 0) {
            ksteps<16, 16>(acc, lds + frag_off<16>(0, g, rl), b);
        } else {  // register-fed A operand
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[t] = MFMA16(b[(s + t) & 15], b[s], acc[t]);
        }
    }
    f32x4 s = acc[0];
    for (int t = 1; t < 16; ++t) s += acc[t];
    out[blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    float *w, *out;
    hipMalloc(&w, 64 * 256 * 4);
    hipMalloc(&out, 1024 * 1024 * 4);
    hipMemset(w, 0, 64 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rounds = 400;
    for (int mode = 0; mode < 2; ++mode)
        for (int nw : {4, 8, 12}) {
            auto kern = mode == 0 ? probe<0> : probe<1>;
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
            kern<<<256, nw * 64, 65536 + 64, 0>>>(w, out, rounds);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            kern<<<256, nw * 64, 65536 + 64, 0>>>(w, out, rounds);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = 256.0 * nw * rounds * 256 * 2048.0;
            printf("mode %d (A from %s) waves/CU %2d: %.3f ms  %.1f TFLOP/s\n", mode, mode ? "regs" : "LDS", nw, ms,
                   flops / ms / 1e9);
        }
    return 0;
}
