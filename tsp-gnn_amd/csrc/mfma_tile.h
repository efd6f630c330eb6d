// Shared device-side building blocks of the MFMA kernels (forward and backward): fragment
// loads/stores, LDS staging of packed weights, software-pipelined k-steps, row statistics.
// See dense.hip for the "transposed chaining" layout these helpers implement.
#pragma once
#include "common.h"

namespace tspgnn {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Straight float4 copy global -> LDS with 8 loads in flight per thread (the weights are already in
// fragment order, so staging is a pure, fully coalesced stream).
__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int nfloats, int tid,
                                            int nthreads) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    const int n4 = nfloats >> 2;
    for (int base = tid; base < n4; base += nthreads * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * nthreads;
            if (idx < n4) v[u] = s4[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * nthreads;
            if (idx < n4) d4[idx] = v[u];
        }
    }
}

// acc[t] (t in [0,NT)) += W_frag(step s, tile t) * bval for all output tiles of one k-step.
// wrow points at the LDS fragment row of (s, g) for this lane (already offset by jl).
template <int NT>
__device__ __forceinline__ void kstep(f32x4 (&acc)[NT], const float* wrow, float bval) {
    if constexpr (NT == 2) {
        const float2 aw = *reinterpret_cast<const float2*>(wrow);
        acc[0] = MFMA16(aw.x, bval, acc[0]);
        acc[1] = MFMA16(aw.y, bval, acc[1]);
    } else {
#pragma unroll
        for (int u = 0; u < NT / 4; ++u) {
            const f32x4 aw = ld4(wrow + u * 64);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[u * 4 + tt] = MFMA16(aw[tt], bval, acc[u * 4 + tt]);
        }
    }
}

// LDS float offset of the fragment row (s,g) for lane jl, for a matrix with NT output tiles.
template <int NT>
__device__ __forceinline__ int frag_off(int s, int g, int jl) {
    if constexpr (NT == 2)
        return ((s * 4 + g) * 16 + jl) * 2;
    else
        return (s * 4 + g) * (NT / 4) * 64 + jl * 4;
}

// NSTEPS consecutive k-steps (fragment rows s0 .. s0+NSTEPS-1, contiguous in LDS) with the weight
// fragments double-buffered in registers: the ds_reads of step s+1 are issued before the MFMAs
// of step s.  b[s] is the B-operand value of step s.
template <int NT, int NSTEPS>
__device__ __forceinline__ void ksteps(f32x4 (&acc)[NT], const float* w0, const float (&b)[NSTEPS]) {
    static_assert(NT % 4 == 0, "b128 fragment path");
    constexpr int U = NT / 4;
    constexpr int STRIDE = NT * 64;  // floats between the fragment rows of consecutive k-steps
    f32x4 w[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[0][u] = ld4(w0 + u * 64);
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        if (s + 1 < NSTEPS) {
#pragma unroll
            for (int u = 0; u < U; ++u) w[(s + 1) & 1][u] = ld4(w0 + (s + 1) * STRIDE + u * 64);
        }
        // Pin the order: the ds_reads of step s+1 stay ABOVE the MFMAs of step s (their s_waitcnt
        // lands at their first use, one step later), so one wavefront alone keeps the matrix pipe
        // busy instead of alternating "read, wait, 4 MFMA".
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[u * 4 + tt] = MFMA16(w[s & 1][u][tt], b[s], acc[u * 4 + tt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Sum of the lane's D/4 values of one gate, reduced over the 4 lane groups that share a row.
template <int TPG>
__device__ __forceinline__ void ln_gate(f32x4 (&v)[TPG], const float* gamma, const float* beta, int g, int D) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TPG; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    s = sum_over_lane_groups16(s);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dlt = v[t][r] - mean;
            q = fmaf(dlt, dlt, q);
        }
    }
    q = sum_over_lane_groups16(q);
    const float var = q / (float)D;
    // tf.contrib.layers.layer_norm: variance_epsilon = 1e-12; x*inv + (beta - mean*inv)
    const float rstd = __builtin_amdgcn_rsqf(var + 1e-12f);  // v_rsq_f32, ~1 ulp
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        const f32x4 ga = ld4(gamma + t * 16 + g * 4);
        const f32x4 be = ld4(beta + t * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float inv = rstd * ga[r];
            v[t][r] = fmaf(v[t][r], inv, be[r] - mean * inv);
        }
    }
}

// k-loop of one 16-row tile over the 16-column blocks q in [q_beg, q_end) of the B operand, which is
// the concatenation [x | h] of two row-major arrays (xrow/hrow already point at this lane's row and
// lane-group column g*4; QX = number of blocks that come from x).  lds_k holds the fragment rows of
// the k-steps starting at block q_base.  The B fragments are fetched four blocks (>= 64 MFMAs per
// output tile group) ahead of their use so the global-load latency hides behind the MFMA stream.
template <int NT>
__device__ __forceinline__ void gemm_kloop(f32x4 (&acc)[NT], const float* lds_k, int q_base, int q_beg, int q_end,
                                           const float* xrow, const float* hrow, int QX, int g, int rl) {
    constexpr int GQ = 4;
    if (q_beg >= q_end) return;
    auto frag = [&](int q) -> f32x4 {
        const int qq = q < q_end ? q : q_end - 1;  // clamp: tail loads stay in bounds
        return ld4(qq < QX ? xrow + qq * 16 : hrow + (qq - QX) * 16);
    };
    f32x4 cur[GQ], nxt[GQ];
#pragma unroll
    for (int i = 0; i < GQ; ++i) cur[i] = frag(q_beg + i);
    for (int q0 = q_beg; q0 < q_end; q0 += GQ) {
        if (q0 + GQ < q_end) {
#pragma unroll
            for (int i = 0; i < GQ; ++i) nxt[i] = frag(q0 + GQ + i);
        }
        if (q0 + GQ <= q_end) {
            float b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = cur[i >> 2][i & 3];
            ksteps<NT, 16>(acc, lds_k + frag_off<NT>((q0 - q_base) * 4, g, rl), b);
        } else {
#pragma unroll
            for (int i = 0; i < GQ; ++i) {
                if (q0 + i < q_end) {
                    float b[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) b[p] = cur[i][p];
                    ksteps<NT, 4>(acc, lds_k + frag_off<NT>((q0 + i - q_base) * 4, g, rl), b);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GQ; ++i) cur[i] = nxt[i];
    }
}

template <int D>
__device__ __forceinline__ void lstm_kloop(f32x4 (&acc)[D / 4], const float* lds_k, int q_base, int q_beg, int q_end,
                                           const float* xrow, const float* hrow, int QX, int g, int rl) {
    gemm_kloop<D / 4>(acc, lds_k, q_base, q_beg, q_end, xrow, hrow, QX, g, rl);
}

// Sum of v over the 16 lanes of a DPP row (the 16 rows of a tile that share lane group g), by
// rotate-and-add; every lane ends with the total, in a fixed order (deterministic).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

}  // namespace tspgnn
