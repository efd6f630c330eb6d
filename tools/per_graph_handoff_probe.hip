// Development probe (VERDICT r03 "next" #6): the data dependence of the V<-E row-sum is PER GRAPH -- a vertex of
// graph g needs only g's n(n-1)/2 message rows (instance_loader.py:56-66, block-diagonal EV) -- so what does a per-graph
// hand-off cost on MI355X, against the kernel boundary (1.56 us) + separate row-sum launch (7.3 us in situ at C2) that
// express the dependence today?
//
// Geometry = C2: G = 128 graphs of ROWS = 784 message rows x 256 B (780 padded to 49 tiles of 16), P = 2 producer workgroups
// per graph (the edge task hands a graph's tiles to ~2 workgroups), 768 threads each as the cell launch.  A producer
//   1. streams `work` KB from a big array (its share of the edge task's h / c / Zx traffic; skews the arrivals),
//   2. writes its half of the graph's rows with 16-byte write-through stores (`sc0 sc1`: no fence needed, DESIGN 7 / the
//      guide's R1 form), waits for them (vmcnt(0)),
//   3. stamps s_memrealtime and bumps the graph's arrival counter (relaxed, agent scope).
// Two consumers are measured in the same launch shape:
//   mode 0 "poller":       one 256-thread workgroup per graph polls ITS counter (lane 0, relaxed sc1 load + s_sleep), stamps
//                          the moment it sees P arrivals, then row-sums the graph (n = 40 vertices x 39 rows, `sc1` loads,
//                          the fixed CSR order) and stamps again;
//   mode 1 "last arriver": no consumer workgroups -- the producer whose increment returns P - 1 does the graph's row-sum
//                          itself (all 12 wavefronts), i.e. the design the verdict sketches.
// Reported per mode, idle (work = 0) and loaded: hand-off = seen - last arrival (mode 0), row-sum = done - seen / done -
// last arrival, and the launch's wall time by HIP events; medians / p99 over graphs x repetitions; every sum is checked.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/per_graph_handoff_probe.hip -o tools/per_graph_handoff_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int G = 128, NV = 40, ROWS = 784, REAL_ROWS = NV * (NV - 1) / 2, P = 2, D4 = 16;   // 16 float4 = 256 B per row

__device__ __forceinline__ void st_wt(f32x4* p, f32x4 v) {
    // (s_nop: the store reads its 16 bytes of data up to two wait states after issue -- a hazard the compiler cannot see
    // inside an asm)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ld_sc1(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 ld_sc1_nowait(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

// row-sum of graph g by one workgroup: wavefront w takes vertices w, w + nw, ...; 4 rows per step as csr_rowsum
__device__ void graph_rowsum(const f32x4* rows_g, f32x4* sums_g, int tid, int nthreads) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6, sub = lane >> 4, c = lane & 15;
    for (int w = wave; w < NV; w += nw) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // edges of vertex w in ascending id: (u, w) for u < w, then (w, v) for v > w
        f32x4 x[10];
#pragma unroll
        for (int k0 = 0; k0 < 40; k0 += 4) {
            const int k = k0 + sub;
            int e = 0;
            if (k < NV - 1) {
                if (k < w) e = k * (NV - 1) - k * (k - 1) / 2 + (w - k - 1);
                else e = w * (NV - 1) - w * (w - 1) / 2 + (k - w);
            }
            x[k0 / 4] = ld_sc1_nowait(rows_g + (size_t)e * D4 + c);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(x[i]));   // (the sums below stay behind the wait)
#pragma unroll
        for (int k0 = 0; k0 < 40; k0 += 4)
            if (k0 + sub < NV - 1) acc += x[k0 / 4];
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            acc[0] += __shfl_xor(acc[0], off);
            acc[1] += __shfl_xor(acc[1], off);
            acc[2] += __shfl_xor(acc[2], off);
            acc[3] += __shfl_xor(acc[3], off);
        }
        if (sub == 0) sums_g[(size_t)w * D4 + c] = acc;
    }
}

template <int MODE>
__global__ __launch_bounds__(768) void probe(f32x4* rows, f32x4* sums, unsigned* counters, unsigned long long* t_arrive,
                                             unsigned long long* t_seen, unsigned long long* t_done, const f32x4* big,
                                             size_t big_n, int work_kb, unsigned epoch, float* sink) {
    const int tid = threadIdx.x;
    __shared__ unsigned last;
    if ((int)blockIdx.x < G * P) {
        // producers of graph g: blocks g and g + G (different XCDs half the time, as tile ranges fall)
        const int g = blockIdx.x % G, part = blockIdx.x / G;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const size_t n16 = (size_t)work_kb * 64;   // float4 per workgroup
        const size_t base = ((size_t)blockIdx.x * 7919u * 4096u) % (big_n - n16 - 1);
        // skewed load: block b streams work_kb * (1 + (b % 5) / 4) so that arrivals spread over the launch
        const size_t mine = n16 + n16 * (blockIdx.x % 5) / 4;
        for (size_t i = tid; i < mine; i += blockDim.x) s += big[(base + i) % (big_n - 1)];
        f32x4* rg = rows + (size_t)g * ROWS * D4;
        const int r0 = part * (ROWS / P), r1 = r0 + ROWS / P;
        for (int i = r0 * D4 + tid; i < r1 * D4; i += blockDim.x) {
            const int r = i / D4;
            const float val = (float)(epoch + 1) * 0.5f + (float)((g * 31 + r) % 17) * 0.125f;
            st_wt(rg + i, f32x4{val, val + 1.f, val + 2.f, s[0] * 0.f + val + 3.f});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            t_arrive[g * P + part] = now();
            const unsigned old = __hip_atomic_fetch_add(counters + g * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (old == (epoch + 1) * P - 1) ? 1u : 0u;
        }
        if (MODE == 1) {
            __syncthreads();
            if (last) {
                if (tid == 0) t_seen[g] = now();
                graph_rowsum(rg, sums + (size_t)g * NV * D4, tid, blockDim.x);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) t_done[g] = now();
            }
        }
        if (tid == 0 && s[1] == 12345.f) sink[0] = s[2];
    } else if (MODE == 0) {
        const int g = blockIdx.x - G * P;
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(counters + g * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (epoch + 1) * P) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 26)) break;
            }
            t_seen[g] = now();
        }
        __syncthreads();
        graph_rowsum(rows + (size_t)g * ROWS * D4, sums + (size_t)g * NV * D4, tid, blockDim.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) t_done[g] = now();
    }
}

// the reference: the same row-sums as a launch of their own behind the producers (plain loads: the boundary publishes)
__global__ __launch_bounds__(768) void rowsum_all(const f32x4* rows, f32x4* sums) {
    graph_rowsum(rows + (size_t)blockIdx.x * ROWS * D4, sums + (size_t)blockIdx.x * NV * D4, threadIdx.x, blockDim.x);
}

static double pct(std::vector<double>& v, double p) {
    std::sort(v.begin(), v.end());
    return v[(size_t)(p * (v.size() - 1))];
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    f32x4 *rows, *sums, *big;
    unsigned* counters;
    unsigned long long *t_arrive, *t_seen, *t_done;
    float* sink;
    const size_t big_n = (size_t)512 << 20 >> 4;   // 512 MB: beyond the Infinity Cache
    CHECK(hipMalloc(&rows, (size_t)G * ROWS * D4 * 16));
    CHECK(hipMalloc(&sums, (size_t)G * NV * D4 * 16));
    CHECK(hipMalloc(&big, big_n * 16));
    CHECK(hipMemset(big, 0, big_n * 16));
    CHECK(hipMalloc(&counters, G * 64));
    CHECK(hipMalloc(&t_arrive, G * P * 8));
    CHECK(hipMalloc(&t_seen, G * 8));
    CHECK(hipMalloc(&t_done, G * 8));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> ta(G * P), ts(G), td(G);
    std::vector<float> hs((size_t)G * NV * D4 * 4);
    printf("per-graph hand-off probe: %d graphs x %d rows x 256 B, %d producer workgroups (768 threads) per graph\n", G, ROWS, P);
    for (int mode = 0; mode < 3; ++mode) {
        for (int work_kb : {0, 256, 1024}) {
            std::vector<double> hand, sum_after_seen, total, wall, spread;
            long bad = 0;
            CHECK(hipMemset(counters, 0, G * 64));
            for (int rep = 0; rep < reps + 3; ++rep) {
                const unsigned epoch = (unsigned)rep;
                const int grid = mode == 0 ? G * P + G : G * P;
                CHECK(hipEventRecord(e0));
                if (mode == 0) {
                    probe<0><<<grid, 768>>>(rows, sums, counters, t_arrive, t_seen, t_done, big, big_n, work_kb, epoch, sink);
                } else if (mode == 1) {
                    probe<1><<<grid, 768>>>(rows, sums, counters, t_arrive, t_seen, t_done, big, big_n, work_kb, epoch, sink);
                } else {   // two launches: producers (nobody consumes in-launch), then the row-sums
                    probe<2><<<grid, 768>>>(rows, sums, counters, t_arrive, t_seen, t_done, big, big_n, work_kb, epoch, sink);
                    rowsum_all<<<G, 768>>>(rows, sums);
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(ta.data(), t_arrive, G * P * 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(ts.data(), t_seen, G * 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(td.data(), t_done, G * 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(hs.data(), sums, hs.size() * 4, hipMemcpyDeviceToHost));
                // check every sum: vertex w of graph g = sum over its 39 edges of the row value
                for (int g = 0; g < G; ++g)
                    for (int w = 0; w < NV; ++w) {
                        double want = 0;
                        for (int k = 0; k < NV - 1; ++k) {
                            const int e = k < w ? k * (NV - 1) - k * (k - 1) / 2 + (w - k - 1) : w * (NV - 1) - w * (w - 1) / 2 + (k - w);
                            want += (double)(epoch + 1) * 0.5 + (double)((g * 31 + e) % 17) * 0.125;
                        }
                        const float got = hs[((size_t)g * NV + w) * D4 * 4];
                        if (fabs(got - want) > 1e-3 * fabs(want)) ++bad;
                    }
                if (rep < 3) continue;   // warm-up
                wall.push_back(ms * 1e3);
                unsigned long long amin = ~0ull, amax = 0;
                for (int g = 0; g < G; ++g) {
                    if (mode == 2) ts[g] = td[g] = std::max(ta[g * P], ta[g * P + 1]);
                    const unsigned long long la = std::max(ta[g * P], ta[g * P + 1]);
                    amin = std::min(amin, std::min(ta[g * P], ta[g * P + 1]));
                    amax = std::max(amax, la);
                    hand.push_back(((double)ts[g] - (double)la) * 0.01);
                    sum_after_seen.push_back(((double)td[g] - (double)ts[g]) * 0.01);
                    total.push_back(((double)td[g] - (double)la) * 0.01);
                }
                spread.push_back((double)(amax - amin) * 0.01);
            }
            printf("%-13s work %4d KB/WG: launch %.1f us (median), arrivals spread %.1f us | hand-off (seen - last arrival) "
                   "median %.2f p99 %.2f | row-sum after seen median %.2f p99 %.2f | last arrival -> sums done median %.2f "
                   "p99 %.2f us | wrong sums %ld\n",
                   mode == 0 ? "poller" : mode == 1 ? "last-arriver" : "two launches", work_kb, pct(wall, 0.5), pct(spread, 0.5), pct(hand, 0.5),
                   pct(hand, 0.99), pct(sum_after_seen, 0.5), pct(sum_after_seen, 0.99), pct(total, 0.5), pct(total, 0.99), bad);
        }
    }
    return 0;
}
