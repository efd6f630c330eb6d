// Shared device-side building blocks of the MFMA kernels (forward and backward): fragment
// loads/stores, LDS staging of packed weights, software-pipelined k-steps, row statistics.
// See dense.hip for the "transposed chaining" layout these helpers implement.
#pragma once
#include "common.h"

namespace tspgnn {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// global -> LDS, 16 bytes per lane, with gfx950's direct loads (global_load_lds_dwordx4: no VGPR round trip, so every
// request of the stage is in flight at once; the weights are already in fragment order, the copy is a pure stream).
// The LDS address of a lane is the wavefront's base + lane*16, so the base handed to the builtin is lane 0's.  Returns
// after the data has landed in LDS for THIS wavefront (s_waitcnt); callers still need their barrier.
__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int nfloats, int tid,
                                            int nthreads) {
    const int lane = tid & 63, n16 = nfloats >> 2;
    const char* s = reinterpret_cast<const char*>(src);
    char* d = reinterpret_cast<char*>(dst);
    for (int idx = tid; idx - lane < n16; idx += nthreads) {
        if (idx < n16)
            __builtin_amdgcn_global_load_lds(s + (size_t)idx * 16,
                                             (__attribute__((address_space(3))) void*)(d + (size_t)(idx - lane) * 16), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
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

// LayerNorm of one gate of the lane's row: the lane holds D/4 of the D values (TPG tiles x 4), the other three
// lane groups of the row hold the rest.  Written on float pairs so that it compiles to packed fp32 VALU
// instructions (v_pk_add/mul/fma_f32: two values per lane per issue) -- the LayerNorm / gate arithmetic shares
// the SIMD's issue port with the MFMAs (no co-execution), so every VALU instruction saved
// is kernel time.  tf.contrib.layers.layer_norm semantics: biased variance, variance_epsilon = 1e-12.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// The same sum (same order, bit-identical) by gfx950's lane-swap VALU instructions instead of two ds_bpermute round
// trips: v_permlane16_swap(v, v) leaves {even rows of v, odd rows of v} duplicated in the two results, so their sum is
// v[l] + v[l^16] in every lane; v_permlane32_swap does the same for the two 32-lane halves.  Needs the full EXEC mask
// (tools/permlane_probe.hip pins the semantics): forward kernels only.
__device__ __forceinline__ float sum_over_lane_groups16_swap(float v) {
    // (inline asm: hipcc 7.2's __builtin_amdgcn_permlane16_swap returns its two results in one register when both are
    // used in the same expression -- the probe's sum came out as 2 * a[0]; the s_nop covers the VALU-write -> swap-read
    // hazard the compiler cannot see inside an asm)
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float s = a + b, t = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s), "+v"(t));
    return s + t;
}

// CENTERED: the rows of v have zero mean already -- the caller's GEMM used operands whose columns were centred per gate
// (LayerNorm's mean subtraction is the linear map I - 11^T/D on a gate's columns and commutes with the product; see
// tspgnn_lstm_task.z_centered) -- so the mean pass (sum, lane reduction, subtraction: 23 of ~60 instructions) is skipped.
// TRACK: *vmin (unsigned, starts at 0xffffffff) keeps the smallest POSITIVE variance this lane has normalised by, as
// (bit pattern - 1) -- positive floats order like their bit patterns, and a variance of exactly 0 (a constant row: exact
// operands) wraps to the maximum and is ignored.  The f16x2 kernels turn it into bit 1 of the task's range_flag: a z whose
// spread is below the absolute error of its operands' fp16 pieces (h2_tile.h, kH2VarFloor).
template <int TPG, bool SWAP = false, bool CENTERED = false, bool TRACK = false>
__device__ __forceinline__ void ln_gate(f32x4 (&v)[TPG], const float* gamma, const float* beta, int g, int D,
                                        float eps = 1e-12f, unsigned* vmin = nullptr) {
    auto lane_sum = [](float x) { return SWAP ? sum_over_lane_groups16_swap(x) : sum_over_lane_groups16(x); };
    f32x2 q2 = {0.f, 0.f};
    if constexpr (CENTERED) {
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            q2 = fma2(v[t].lo, v[t].lo, q2);
            q2 = fma2(v[t].hi, v[t].hi, q2);
        }
    } else {
        f32x2 s2 = v[0].lo + v[0].hi;
#pragma unroll
        for (int t = 1; t < TPG; ++t) s2 += v[t].lo + v[t].hi;
        const float mean = lane_sum(s2[0] + s2[1]) * (1.0f / (float)D);
        const f32x2 m2 = {mean, mean};
#pragma unroll
        for (int t = 0; t < TPG; ++t) {  // centred values replace v: y = (x - mean) * (rstd * gamma) + beta
            const f32x2 a = v[t].lo - m2, b = v[t].hi - m2;
            q2 = fma2(a, a, q2);
            q2 = fma2(b, b, q2);
            v[t].lo = a;
            v[t].hi = b;
        }
    }
    const float var = lane_sum(q2[0] + q2[1]) * (1.0f / (float)D);
    if constexpr (TRACK) {
        const unsigned vb = __float_as_uint(var) - 1u;
        *vmin = vb < *vmin ? vb : *vmin;
    }
    const float rstd = __builtin_amdgcn_rsqf(var + eps);  // v_rsq_f32, ~1 ulp
    const f32x2 r2 = {rstd, rstd};
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        const f32x4 ga = ld4(gamma + t * 16 + g * 4);
        const f32x4 be = ld4(beta + t * 16 + g * 4);
        v[t].lo = fma2(v[t].lo, ga.lo * r2, be.lo);
        v[t].hi = fma2(v[t].hi, ga.hi * r2, be.hi);
    }
}

// 1/(1+e^-(x+shift)) on a pair: packed multiply-add feeding the two transcendental pairs (v_exp_f32, v_rcp_f32).
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x, float shift = 0.f) {
    constexpr float L = -1.4426950408889634f;
    const f32x2 l2 = {L, L}, sh = {L * shift, L * shift}, one = {1.f, 1.f};
    const f32x2 t = fma2(x, l2, sh);
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    const f32x2 d = e + one;
    return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 relu2(f32x2 x) { return f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)}; }
// the same on an argument that already is t = -log2(e) * (x + shift): 1 / (1 + 2^t)
__device__ __forceinline__ f32x2 sigmoid2_pre(f32x2 t) {
    const f32x2 one = {1.f, 1.f};
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    const f32x2 d = e + one;
    return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// The cell arithmetic on the lane's part of a 16-row tile.  acc = z in the kernel's column order i, j, f, o (TPG
// tiles each); cf = old c.  LN each gate; c' = LN_s(c*sig(f+1) + sig(i)*relu(j)); h' = relu(c')*sig(o); stores.
// PRE: lds_ln holds the gamma / beta of the gates i, f, o already multiplied by -log2(e) (and the forget bias folded into
// beta_f), so their LayerNorm output is the exponent of the sigmoid directly -- one packed multiply-add per pair less;
// eps_z: the epsilon of the four gate LayerNorms (a caller whose z is scaled by 2^s passes 2^2s * 1e-12, which makes
// the normalised gates those of the unscaled z exactly -- a power-of-two scale commutes with every rounding).
template <int D, bool PRE = false, bool SWAP = false, bool CENTERED = false, bool TRACK = false>
__device__ __forceinline__ void lstm_gates(f32x4 (&acc)[D / 4], f32x4 (&cf)[D / 16], const float* lds_ln, int g,
                                           f32x4 (&hn)[D / 16], f32x4 (&nc)[D / 16], float eps_z = 1e-12f,
                                           unsigned* vmin = nullptr) {
    constexpr int TPG = D / 16;
    f32x4 gi[TPG], gj[TPG], gf[TPG], go[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        gi[t] = acc[t];
        gj[t] = acc[TPG + t];
        gf[t] = acc[2 * TPG + t];
        go[t] = acc[3 * TPG + t];
    }
    ln_gate<TPG, SWAP, CENTERED, TRACK>(gi, lds_ln + 0 * D, lds_ln + 1 * D, g, D, eps_z, vmin);
    ln_gate<TPG, SWAP, CENTERED, TRACK>(gj, lds_ln + 2 * D, lds_ln + 3 * D, g, D, eps_z, vmin);
    ln_gate<TPG, SWAP, CENTERED, TRACK>(gf, lds_ln + 4 * D, lds_ln + 5 * D, g, D, eps_z, vmin);
    ln_gate<TPG, SWAP, CENTERED, TRACK>(go, lds_ln + 6 * D, lds_ln + 7 * D, g, D, eps_z, vmin);
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        if constexpr (PRE) {
            nc[t].lo = fma2(sigmoid2_pre(gi[t].lo), relu2(gj[t].lo), cf[t].lo * sigmoid2_pre(gf[t].lo));
            nc[t].hi = fma2(sigmoid2_pre(gi[t].hi), relu2(gj[t].hi), cf[t].hi * sigmoid2_pre(gf[t].hi));
        } else {
            nc[t].lo = fma2(sigmoid2(gi[t].lo), relu2(gj[t].lo), cf[t].lo * sigmoid2(gf[t].lo, 1.0f));
            nc[t].hi = fma2(sigmoid2(gi[t].hi), relu2(gj[t].hi), cf[t].hi * sigmoid2(gf[t].hi, 1.0f));
        }
    }
    ln_gate<TPG, SWAP>(nc, lds_ln + 8 * D, lds_ln + 9 * D, g, D);
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        if constexpr (PRE) {
            hn[t].lo = relu2(nc[t].lo) * sigmoid2_pre(go[t].lo);
            hn[t].hi = relu2(nc[t].hi) * sigmoid2_pre(go[t].hi);
        } else {
            hn[t].lo = relu2(nc[t].lo) * sigmoid2(go[t].lo);
            hn[t].hi = relu2(nc[t].hi) * sigmoid2(go[t].hi);
        }
    }
}

// lstm_gates<D, true, SWAP> in three stages -- f, then (i, j), then o -- for kernels that form z one gate (pair) at a
// time to keep a quarter / half of the accumulator registers live: the same operations on the same values in the same
// order per element, so the result is bit-identical to the one-stage form.
//   stage f:      cs = c * sig(LN_f(z_f))                     (cs arrives holding the old c)
//   stage (i, j): cs = LN_s(sig(LN_i(z_i)) * relu(LN_j(z_j)) + cs)   = the new (normalised) c
//   stage o:      h' = relu(c') * sig(LN_o(z_o))
template <int D, bool SWAP, bool CENTERED = false, bool TRACK = false>
__device__ __forceinline__ void lstm_stage_f(f32x4 (&zf)[D / 16], f32x4 (&cs)[D / 16], const float* lds_ln, int g,
                                             float eps_z = 1e-12f, unsigned* vmin = nullptr) {
    constexpr int TPG = D / 16;
    ln_gate<TPG, SWAP, CENTERED, TRACK>(zf, lds_ln + 4 * D, lds_ln + 5 * D, g, D, eps_z, vmin);
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        cs[t].lo = cs[t].lo * sigmoid2_pre(zf[t].lo);
        cs[t].hi = cs[t].hi * sigmoid2_pre(zf[t].hi);
    }
}
template <int D, bool SWAP, bool CENTERED = false, bool TRACK = false>
__device__ __forceinline__ void lstm_stage_ij(f32x4 (&zij)[D / 8], f32x4 (&cs)[D / 16], const float* lds_ln, int g,
                                              float eps_z = 1e-12f, unsigned* vmin = nullptr) {
    constexpr int TPG = D / 16;
    f32x4 gi[TPG], gj[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        gi[t] = zij[t];
        gj[t] = zij[TPG + t];
    }
    ln_gate<TPG, SWAP, CENTERED, TRACK>(gi, lds_ln + 0 * D, lds_ln + 1 * D, g, D, eps_z, vmin);
    ln_gate<TPG, SWAP, CENTERED, TRACK>(gj, lds_ln + 2 * D, lds_ln + 3 * D, g, D, eps_z, vmin);
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        cs[t].lo = fma2(sigmoid2_pre(gi[t].lo), relu2(gj[t].lo), cs[t].lo);
        cs[t].hi = fma2(sigmoid2_pre(gi[t].hi), relu2(gj[t].hi), cs[t].hi);
    }
    ln_gate<TPG, SWAP>(cs, lds_ln + 8 * D, lds_ln + 9 * D, g, D);
}
template <int D, bool SWAP, bool CENTERED = false, bool TRACK = false>
__device__ __forceinline__ void lstm_stage_o(f32x4 (&zo)[D / 16], const f32x4 (&nc)[D / 16], const float* lds_ln, int g,
                                             f32x4 (&hn)[D / 16], float eps_z = 1e-12f, unsigned* vmin = nullptr) {
    constexpr int TPG = D / 16;
    ln_gate<TPG, SWAP, CENTERED, TRACK>(zo, lds_ln + 6 * D, lds_ln + 7 * D, g, D, eps_z, vmin);
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        hn[t].lo = relu2(nc[t].lo) * sigmoid2_pre(zo[t].lo);
        hn[t].hi = relu2(nc[t].hi) * sigmoid2_pre(zo[t].hi);
    }
}

template <int D>
__device__ __forceinline__ void lstm_epilogue(f32x4 (&acc)[D / 4], f32x4 (&cf)[D / 16], const float* lds_ln, int g,
                                              bool valid, float* hd, float* cd) {
    constexpr int TPG = D / 16;
    f32x4 hn[TPG], nc[TPG];
    lstm_gates<D>(acc, cf, lds_ln, g, hn, nc);
    if (valid) {
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            st4(hd + t * 16, hn[t]);
            st4(cd + t * 16, nc[t]);
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

// Sums of SIXTEEN per-lane values over the 16 lanes of a DPP row at once: a reduce-scatter by XOR partners (masks 15, 7, 3, 1
// = row_mirror, row_half_mirror, quad_perm [3,2,1,0], quad_perm [1,0,3,2]; each step a lane keeps the half of its values its
// own bit selects and adds the partner's copy of that half).  Lane rl of the row ends with the total of v[rl]: 15 DPP
// additions + 30 selects for sixteen sums where row16_sum spends 64 dependent rotate-and-adds.  Fixed order (deterministic).
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_reduce_scatter(const float (&v)[16], int rl) {
    const bool b3 = (rl & 8) != 0, b2 = (rl & 4) != 0, b1 = (rl & 2) != 0, b0 = (rl & 1) != 0;
    float w[8], u[4], q[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = (b3 ? v[m + 8] : v[m]) + dpp_row<0x140>(b3 ? v[m] : v[m + 8]);
#pragma unroll
    for (int m = 0; m < 4; ++m) u[m] = (b2 ? w[m + 4] : w[m]) + dpp_row<0x141>(b2 ? w[m] : w[m + 4]);
#pragma unroll
    for (int m = 0; m < 2; ++m) q[m] = (b1 ? u[m + 2] : u[m]) + dpp_row<0x1b>(b1 ? u[m] : u[m + 2]);
    return (b0 ? q[1] : q[0]) + dpp_row<0xb1>(b0 ? q[0] : q[1]);
}

}  // namespace tspgnn
