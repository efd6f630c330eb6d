// Development probe (VERDICT r02 #5a): what does a grid-wide barrier cost on MI355X, against the 1.5-1.9 us of the
// kernel boundary it would replace in a persistent T-step forward?  256 workgroups (one per CU) x 256 / 768 threads run
// ROUNDS barriers; block 0 stamps s_memrealtime (100 MHz) every round, so the host gets median and tail, and the whole
// loop is also timed with HIP events.  Three forms:
//   flat    one monotonic counter: lane 0 release fence -> atomicAdd -> relaxed sc1 poll (+ s_sleep) -> acquire fence
//   xcd     hierarchical: per-XCC counter, the last arriver of an XCC bumps the top counter, the last of those publishes
//           the generation; everybody polls the generation word
//   flat+w  the flat form after every workgroup has written 16 KB (dirty lines for the release fence to write back: the
//           cell launch's tile stores), i.e. the barrier as it would sit between a producer and a consumer phase
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned poll(unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// returns false on timeout (a block not resident: never on a 256-block grid, but every spin is bounded)
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned want) {
    for (unsigned spins = 0; poll(p) < want; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 24)) return false;
    }
    return true;
}

__device__ __forceinline__ void release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void barrier_flat(unsigned* counter, unsigned round, unsigned grid) {
    __syncthreads();
    if (threadIdx.x == 0) {
        release_agent();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_ge(counter, (round + 1) * grid);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// words: [0..7] per-XCC arrival counters (64 B apart), [8] top counter, [9] generation
__device__ __forceinline__ void barrier_xcd(unsigned* w, unsigned round, unsigned grid) {
    __syncthreads();
    if (threadIdx.x == 0) {
        release_agent();
        const unsigned xcc = blockIdx.x & 7u;                     // (dispatch order: speed only -- any split of the
        const unsigned per = (grid + 7u - xcc) / 8u;              //  grid into 8 groups by blockIdx is correct)
        const unsigned a = __hip_atomic_fetch_add(w + xcc * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == (round + 1) * per - 1) {
            const unsigned groups = grid < 8u ? grid : 8u;
            const unsigned t = __hip_atomic_fetch_add(w + 8 * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (round + 1) * groups - 1)
                __hip_atomic_store(w + 9 * 16, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wait_ge(w + 9 * 16, round + 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int MODE>
__global__ void probe(unsigned* words, unsigned long long* stamps, float* slab, int rounds) {
    const unsigned grid = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 2) {   // 16 KB of fresh dirty lines per workgroup and round
            float* dst = slab + (size_t)blockIdx.x * 4096;
            for (int i = threadIdx.x; i < 4096; i += blockDim.x) dst[i] = (float)(r + i);
        }
        if (MODE == 1) barrier_xcd(words, (unsigned)r, grid);
        else barrier_flat(words, (unsigned)r, grid);
        if (blockIdx.x == 0 && threadIdx.x == 0) stamps[r] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 10000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    unsigned* words;
    unsigned long long* stamps;
    float* slab;
    CHECK(hipMalloc(&words, 4096));
    CHECK(hipMalloc(&stamps, sizeof(unsigned long long) * rounds));
    CHECK(hipMalloc(&slab, (size_t)grid * 4096 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char* names[3] = {"flat", "xcd", "flat+16KB/WG dirty"};
    for (int threads : {256, 768}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms up
                CHECK(hipMemset(words, 0, 4096));
                CHECK(hipEventRecord(e0));
                if (mode == 0) probe<0><<<grid, threads>>>(words, stamps, slab, rounds);
                if (mode == 1) probe<1><<<grid, threads>>>(words, stamps, slab, rounds);
                if (mode == 2) probe<2><<<grid, threads>>>(words, stamps, slab, rounds);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
            }
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> st(rounds);
            CHECK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * rounds, hipMemcpyDeviceToHost));
            std::vector<double> d;
            for (int r = 1; r < rounds; ++r) d.push_back((double)(st[r] - st[r - 1]) * 0.01);   // 100 MHz -> us
            std::sort(d.begin(), d.end());
            printf("%-20s %4d threads x %d WGs: %.3f us/round (events); per round median %.2f  p90 %.2f  p99 %.2f  max %.2f us\n",
                   names[mode], threads, grid, ms * 1e3 / rounds, d[d.size() / 2], d[d.size() * 9 / 10], d[d.size() * 99 / 100],
                   d.back());
        }
    }
    // reference: the kernel boundary the barrier would replace (trivial 256-WG kernels back to back, replayed from a HIP
    // graph so that the host is out of the picture)
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 500; ++i) probe<0><<<grid, 256, 0, st>>>(words, stamps, slab, 0);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, st));
        CHECK(hipGraphLaunch(ge, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("empty 256-WG kernels back to back (graph replay of 500): %.3f us per launch\n", ms * 1e3 / 500);
    return 0;
}
