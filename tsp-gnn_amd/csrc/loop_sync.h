// Device-side hand-off primitives of the resident T-step loops (mp_loop_h2.hip, mp_resident_h2.hip): monotone device
// counters polled with agent-scope loads, write-through stores / L1-bypassing loads for rows that cross compute units,
// bounded waits.  The forms are those of the CDNA4 guide (producer: sc0 sc1 stores, vmcnt(0), agent-scope add; consumer:
// poll one word, then sc0 sc1 loads -- or ONE agent-scope acquire before loads that re-use lines).
#pragma once
#include "common.h"

namespace tspgnn {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxWT = 17;                 // sc0 sc1: write-through store / L1-bypassing load
constexpr unsigned kSpinLimit = 1u << 19;  // polls (each ~1 us: a load round trip + s_sleep)

__device__ __forceinline__ int vzero() {   // a zero the optimiser cannot see: keeps uniform addresses on the vector path
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
__device__ __forceinline__ unsigned ld_word(const unsigned* p) {
    return __hip_atomic_load(p + vzero(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wave-uniform wait until *cnt >= target.  `dead`: a wait of this wavefront (or, through *status, of any other) has
// expired -- nothing waits any more, the launch runs out with garbage and the host raises.
__device__ __forceinline__ void wait_ge(const unsigned* cnt, unsigned target, bool& dead, unsigned* status) {
    if (target == 0u || dead) return;
    unsigned spins = 0;
    for (;;) {
        const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_word(cnt));
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 127u) == 0u) {
            const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_word(status));
            if (s != 0u || spins > kSpinLimit) {
                dead = true;
                if ((threadIdx.x & 63) == 0) atomicOr(status, 1u);
                break;
            }
        }
    }
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void arrive(unsigned* cnt, unsigned n) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, kAuxWT));
}
__device__ __forceinline__ void st4wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, kAuxWT);
}

// Optional phase trace (a TRACE variant of a loop kernel): per (workgroup, wavefront) 16 sums of s_memrealtime ticks
// (100 MHz).
template <bool ON, int N = 16>
struct LoopTrace {
    unsigned long long* dst;
    unsigned long long prev;
    unsigned long long acc[ON ? N : 1];
    __device__ __forceinline__ void begin(unsigned long long* p) {
        if constexpr (ON) {
            dst = p;
            for (int i = 0; i < N; ++i) acc[i] = 0;
            prev = __builtin_amdgcn_s_memrealtime();
        }
    }
    __device__ __forceinline__ void mark(int i) {
        if constexpr (ON) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            acc[i] += now - prev;
            prev = now;
        }
    }
    // absolute time of an event of ONE chosen step (slots 8..15): timelines across wavefronts (tools/resident_trace.py)
    __device__ __forceinline__ void stamp(int i, bool chosen) {
        if constexpr (ON) {
            if (chosen) acc[8 + i] = __builtin_amdgcn_s_memrealtime();
        }
    }
    __device__ __forceinline__ void flush() {
        if constexpr (ON) {
            if ((threadIdx.x & 63) == 0)
                for (int i = 0; i < N; ++i) dst[i] = acc[i];
        }
    }
};

}  // namespace tspgnn
