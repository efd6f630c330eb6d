// Development probe: does an fp32 VALU stream slow down the fp32 MFMA stream of the OTHER wavefront on
// the same SIMD?  4 waves run ksteps<> MFMAs (one per SIMD), 4 waves run dependent-free v_fma chains.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I tsp-gnn_amd/csrc tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "mfma_tile.h"
using namespace tspgnn;

// mode: 0 = MFMA waves only (4), 1 = VALU waves only (4), 2 = both (8 waves: 0-3 MFMA, 4-7 VALU)
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, float* __restrict__ out, int rounds, int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    copy_to_lds(lds, w, 64 * 256, tid, blockDim.x);
    __syncthreads();
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    float res = 0.f;
    if (do_mfma) {
        f32x4 acc[16];
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float b[16];
        for (int i = 0; i < 16; ++i) b[i] = 0.001f * (lane + i);
        for (int r = 0; r < rounds; ++r) ksteps<16, 16>(acc, lds + frag_off<16>(0, g, rl), b);
        f32x4 s = acc[0];
        for (int t = 1; t < 16; ++t) s += acc[t];
        res = s[0] + s[1] + s[2] + s[3];
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
        const float a = 1.0001f, c = 0.0001f;
        // same cycle budget as the MFMA loop: 256 MFMA x 32 cyc = 8192 cyc per round = 4096 v_fma (2 cyc each)
        for (int r = 0; r < rounds; ++r)
            for (int k = 0; k < 256; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], a, c);
        for (int i = 0; i < 16; ++i) res += v[i];
    }
    out[blockIdx.x * blockDim.x + tid] = res;
}

int main() {
    float *w, *out;
    hipMalloc(&w, 64 * 256 * 4);
    hipMalloc(&out, 1024 * 1024 * 4);
    hipMemset(w, 0, 64 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rounds = 200;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    const char* names[3] = {"4 MFMA waves", "4 VALU waves", "4 MFMA + 4 VALU waves"};
    for (int mode = 0; mode < 3; ++mode) {
        const int nw = mode == 2 ? 8 : 4;
        probe<<<256, nw * 64, 65536 + 64, 0>>>(w, out, rounds, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<256, nw * 64, 65536 + 64, 0>>>(w, out, rounds, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-24s: %.3f ms\n", names[mode], ms);
    }
    return 0;
}
