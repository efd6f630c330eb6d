// Device-side building blocks of the f16x2 kernels (dense_h2.hip: forward; dense_bwd_h2.hip: backward): the two-piece
// fp16 split of an fp32 operand, the k-block MFMA step over packed weights, LDS staging.  See dense_h2.hip for the
// arithmetic and its range conventions.
#pragma once
#include "common.h"
#include "mfma_tile.h"

namespace tspgnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

constexpr int kH2ScaleLog2 = TSPGNN_H2_WEIGHT_SCALE_LOG2;
constexpr float kH2Scale = (float)(1 << kH2ScaleLog2);
constexpr float kH2InvScale = 1.0f / kH2Scale;
constexpr float kH2GateEps = 1e-12f * kH2Scale * kH2Scale;  // LayerNorm epsilon of a z scaled by 2^s
constexpr float kNegLog2e = -1.4426950408889634f;
// The low end of the range.  The lo piece of an operand below 2^-3 is an fp16 subnormal: the split of such a value carries
// an ABSOLUTE error up to 2^-25 where the reference's fp32 carries a relative one.  Through a GEMM with weights of
// magnitude w that is ~2^-25 * 2^s * w * sqrt(K) in the scaled z (7e-7 for xavier weights at K = 64) -- nothing next to a
// z of ordinary size, but LayerNorm divides by the row's spread: a gate row whose standard deviation is below 2^-5 (in
// units of the unscaled z) would come out with a relative error above ~1e-6 per step.  The cell kernels keep the smallest
// positive variance they normalise by (ln_gate<..., TRACK>) and raise bit 1 of the task's range_flag when it is below
// this floor; the caller repeats the batch on bf16x3 (whose pieces keep fp32's exponent range), exactly as for bit 0.
constexpr float kH2VarFloor = kH2Scale * kH2Scale / 1024.0f;   // (2^-5)^2 in units of the scaled z's variance

// x = hi + lo to 2^-24 relative: two v_cvt_pk_f16_f32 and two v_fma_mix_f32 (x - float(hi), the fp16 operand widened
// inside the instruction) per pair of values.
__device__ __forceinline__ void split2(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 v = {x[i], x[i + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x[i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x[i + 1]));
        const f32x2 r = {r0, r1};
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[i] = h[0];
        hi[i + 1] = h[1];
        lo[i] = l[0];
        lo[i + 1] = l[1];
    }
}

// The split for operands whose entries span many binades INSIDE one row (gradients): x = hi + 2^-11 * m with
// hi = rn16(x), m = rn16(2^11 (x - hi)).  The second piece is scaled up into fp16's NORMAL range -- the plain lo piece of
// split2 goes subnormal below 2^-14 and then carries an absolute error of 2^-25 of the row's scale however small the
// entry, which a column sum over 10^5 rows turns into a relative error several times fp32's (round 5: bias / LayerNorm
// shift gradients at full C2 size 6e-4..8e-4 against 1.5e-4 for the fp32-MFMA backward).  The products with m go to an
// accumulator of their own (kblock_h2_side) and are folded in with one fma per output: for |x| <= 1 the pair represents x
// to 2^-35 absolutely.
__device__ __forceinline__ void split2s(const float (&x)[8], f16x8& hi, f16x8& m) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 v = {x[i], x[i + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x[i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x[i + 1]));
        const f32x2 r = {r0 * 2048.0f, r1 * 2048.0f};
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[i] = h[0];
        hi[i + 1] = h[1];
        m[i] = l[0];
        m[i + 1] = l[1];
    }
}

// The same split, keeping a witness of fp16 overflow: an |x| >= 65520 rounds to hi = +-inf, and then the residual
// x - hi is -+inf (NaN for a non-finite x, which the maximum ignores: the fp32 reference is non-finite there too).
// `wit` accumulates max |residual| -- one v_max3_f32 per PAIR of values; h2_range_report() turns an infinite witness
// into bit 0 of the task's range_flag, once per wavefront.
__device__ __forceinline__ void split2w(const float (&x)[8], f16x8& hi, f16x8& lo, float& wit) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 v = {x[i], x[i + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x[i]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x[i + 1]));
#if !defined(H2_NO_WITNESS)   // (A/B builds only: the price of the guard)
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(wit) : "v"(r0), "v"(r1));
#endif
        const f32x2 r = {r0, r1};
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[i] = h[0];
        hi[i + 1] = h[1];
        lo[i] = l[0];
        lo[i + 1] = l[1];
    }
}
__device__ __forceinline__ void h2_range_report(unsigned* flag, float wit, unsigned vmin = 0xffffffffu) {
    if (flag == nullptr) return;
    unsigned bits = __any(!(wit <= 3.0e38f)) ? 1u : 0u;
    // vmin = (smallest positive variance's bit pattern) - 1, see ln_gate<..., TRACK>
    if (__any(vmin < __float_as_uint(kH2VarFloor) - 1u)) bits |= 2u;
    if (bits != 0u && (threadIdx.x & 63) == 0) atomicOr(flag, bits);
}

// The projected messages Zx = 2^s (y Kx) between an f16x2 projection (producer, vertex rows) and an f16x2 cell in
// gather-init mode (consumer, edge rows) are stored BLOCKED by 16 source rows: the float4 (columns 16t + 4g .. +3) of
// row v lives at  (((v / 16) * (D/4) + t) * 4 + g) * 64 + (v % 16) * 4  floats.  The producer's 16-row tile then stores
// 1 KiB contiguous per instruction, and the consumer -- lane (rl, g) gathers row v_rl, and consecutive edges of a graph
// have consecutive far endpoints -- finds the four lanes of a load quad in ONE 64-byte segment instead of in four
// different 1 KB rows (the L1 looks up a line per distinct segment of a quad: 64 -> ~20 cycles per gather instruction).
// Rows are padded to a multiple of 16.  h2_zx_row(v, g): offset of (v, t = 0, g); add 256 floats per tile t.
template <int D>
__device__ __forceinline__ unsigned h2_zx_row(unsigned v, int g) {
    return (v >> 4) * (unsigned)(D / 4 * 256) + (unsigned)g * 64u + (v & 15u) * 4u;
}

// The same blocking for a [rows, D] state array (h, c of the T-step loop's ping-pong buffers): offset of (r, t = 0, g)
// and the stride between tiles t, row-major when `blocked` is false.
template <int D>
__device__ __forceinline__ unsigned h2_state_row(unsigned r, int g, bool blocked) {
    return blocked ? (r >> 4) * (unsigned)(D / 16 * 256) + (unsigned)g * 64u + (r & 15u) * 4u : r * (unsigned)D + (unsigned)g * 4u;
}

// The lane id (0..63) through an asm the optimiser cannot hoist out of a loop or fold with another copy: quantities derived
// from it inside a loop body are recomputed per iteration rather than carried -- and spilled -- across iterations.
__device__ __forceinline__ int opaque_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ f16x8 ldw(const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); }

// A 16-byte store of a row nothing in this launch reads again, optionally write-through (`sc0 sc1`): the line then does
// not sit dirty in the XCD's L2 until the end-of-kernel release writes it back.
// The trailing s_nop is NOT optional: a VMEM store of more than 8 bytes reads its data registers up to two wait states
// after issue on gfx940+; the compiler's hazard recogniser covers that for its own stores (GCNHazardRecognizer, "store
// data overwritten by the next VALU") and cannot see inside an asm -- without it the first build of this helper stored
// garbage whenever the register allocator reused a data register at once (anchor C1: loss off by 2.4e-3).
template <bool WT>
__device__ __forceinline__ void st4o(float* p, f32x4 v) {
    if constexpr (WT) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    } else {
        *reinterpret_cast<f32x4*>(p) = v;
    }
}

// acc[t] += W-block(kb, all NT tiles) x B for one 32-feature k-block; wh / wl = the two pieces of the packed matrix
// (LDS), (bh, bl) = split2 of the lane's eight B values.  The fragments of tile t+1 are fetched while the three
// (dependent) MFMAs of tile t run.
#ifndef H2_PF
#define H2_PF 1   // fragment pairs in flight ahead of the MFMAs (2 and 3 measured slower: registers)
#endif
#ifndef H2_LN_SWAP
#define H2_LN_SWAP 1
#endif
// (NT_TOTAL = column tiles of the packed matrix, [T0, T0 + NT) = the tiles this call multiplies: a kernel that forms z one
// gate (pair) at a time passes a slice; per output tile the three MFMAs and their order are those of the whole-matrix
// call, so the results are bit-identical.)
template <int NT_TOTAL, int T0, int NT, int PFW = H2_PF>
__device__ __forceinline__ void kblock_h2_sub(f32x4 (&acc)[NT], const _Float16* wh, const _Float16* wl, int kb, int g, int jl,
                                              const f16x8& bh, const f16x8& bl) {
    const int off = ((kb * 4 + g) * NT_TOTAL * 16 + jl) * 8 + T0 * 128;
    // The fragments of the next PF tiles are in flight while the three MFMAs of this one run (the compiler interleaves
    // the MFMA chains of neighbouring tiles on top of that).  PFW: a call site with registers to spare asks for more.
    constexpr int PF = PFW < NT ? PFW : NT;
    f16x8 ah[PF + 1], al[PF + 1];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        ah[p] = ldw(wh + off + p * 128);
        al[p] = ldw(wl + off + p * 128);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + PF < NT) {
            ah[(t + PF) % (PF + 1)] = ldw(wh + off + (t + PF) * 128);
            al[(t + PF) % (PF + 1)] = ldw(wl + off + (t + PF) * 128);
        }
        const f16x8 a_h = ah[t % (PF + 1)], a_l = al[t % (PF + 1)];
        f32x4 c = acc[t];
        c = MFMA_F16(a_l, bh, c);  // smallest terms first
        c = MFMA_F16(a_h, bl, c);
        c = MFMA_F16(a_h, bh, c);
        acc[t] = c;
    }
}
template <int NT>
__device__ __forceinline__ void kblock_h2(f32x4 (&acc)[NT], const _Float16* wh, const _Float16* wl, int kb, int g, int jl,
                                          const f16x8& bh, const f16x8& bl) {
    kblock_h2_sub<NT, 0, NT>(acc, wh, wl, kb, g, jl, bh, bl);
}
// The same for NP row tiles at once (a wavefront that holds the B operands of several tiles): one fragment fetch feeds NP
// independent MFMA chains -- 1/NP of the LDS reads per tile and NP times the instruction-level parallelism; per output
// tile the three MFMAs and their order are unchanged.
template <int NT_TOTAL, int T0, int NT, int NP>
__device__ __forceinline__ void kblock_h2_multi(f32x4 (&acc)[NP][NT], const _Float16* wh, const _Float16* wl, int kb, int g,
                                                int jl, const f16x8 (&bh)[NP], const f16x8 (&bl)[NP]) {
    const int off = ((kb * 4 + g) * NT_TOTAL * 16 + jl) * 8 + T0 * 128;
    constexpr int PF = H2_PF < NT ? H2_PF : NT;
    f16x8 ah[PF + 1], al[PF + 1];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        ah[p] = ldw(wh + off + p * 128);
        al[p] = ldw(wl + off + p * 128);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + PF < NT) {
            ah[(t + PF) % (PF + 1)] = ldw(wh + off + (t + PF) * 128);
            al[(t + PF) % (PF + 1)] = ldw(wl + off + (t + PF) * 128);
        }
        const f16x8 a_h = ah[t % (PF + 1)], a_l = al[t % (PF + 1)];
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            f32x4 c = acc[n][t];
            c = MFMA_F16(a_l, bh[n], c);  // smallest terms first
            c = MFMA_F16(a_h, bl[n], c);
            c = MFMA_F16(a_h, bh[n], c);
            acc[n][t] = c;
        }
    }
}

// kblock_h2 for a B operand split by split2s: acc += (W_lo + W_hi) x B_hi, side += W_hi x B_m; the caller folds
// acc + 2^-11 * side.  Three MFMAs per product, as kblock_h2.
// (NT_TOTAL / T0 as in kblock_h2_sub: the column tiles [T0, T0 + NT) of a packed matrix NT_TOTAL tiles wide.  wh / wl may
// point into LDS or -- a matrix too large to stay resident, multiplied on few rows -- into global memory: the fragment
// reads are then 1 KB global loads served by the L2.)
template <int NT_TOTAL, int T0, int NT, int PFW = H2_PF>
__device__ __forceinline__ void kblock_h2_side_sub(f32x4 (&acc)[NT], f32x4 (&side)[NT], const _Float16* wh, const _Float16* wl,
                                                   int kb, int g, int jl, const f16x8& bh, const f16x8& bm) {
    const int off = ((kb * 4 + g) * NT_TOTAL * 16 + jl) * 8 + T0 * 128;
    constexpr int PF = PFW < NT ? PFW : NT;
    f16x8 ah[PF + 1], al[PF + 1];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        ah[p] = ldw(wh + off + p * 128);
        al[p] = ldw(wl + off + p * 128);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + PF < NT) {
            ah[(t + PF) % (PF + 1)] = ldw(wh + off + (t + PF) * 128);
            al[(t + PF) % (PF + 1)] = ldw(wl + off + (t + PF) * 128);
        }
        const f16x8 a_h = ah[t % (PF + 1)], a_l = al[t % (PF + 1)];
        side[t] = MFMA_F16(a_h, bm, side[t]);
        f32x4 c = acc[t];
        c = MFMA_F16(a_l, bh, c);
        c = MFMA_F16(a_h, bh, c);
        acc[t] = c;
    }
}
template <int NT, int PFW = H2_PF>
__device__ __forceinline__ void kblock_h2_side(f32x4 (&acc)[NT], f32x4 (&side)[NT], const _Float16* wh, const _Float16* wl,
                                               int kb, int g, int jl, const f16x8& bh, const f16x8& bm) {
    const int off = ((kb * 4 + g) * NT * 16 + jl) * 8;
    constexpr int PF = PFW < NT ? PFW : NT;
    f16x8 ah[PF + 1], al[PF + 1];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        ah[p] = ldw(wh + off + p * 128);
        al[p] = ldw(wl + off + p * 128);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + PF < NT) {
            ah[(t + PF) % (PF + 1)] = ldw(wh + off + (t + PF) * 128);
            al[(t + PF) % (PF + 1)] = ldw(wl + off + (t + PF) * 128);
        }
        const f16x8 a_h = ah[t % (PF + 1)], a_l = al[t % (PF + 1)];
        side[t] = MFMA_F16(a_h, bm, side[t]);
        f32x4 c = acc[t];
        c = MFMA_F16(a_l, bh, c);
        c = MFMA_F16(a_h, bh, c);
        acc[t] = c;
    }
}

// One Dense(D) layer on the lane's part of a 16-row tile, activations chained in registers (D layout).  `bias` holds
// 2^s * b; the output comes back at its true scale.
template <int D>
__device__ __forceinline__ void dense_layer_h2(f32x4 (&a)[D / 16], const _Float16* wh, const _Float16* wl, const float* bias,
                                               bool relu, int g, int rl, float& wit) {
    constexpr int NT = D / 16, KB = D / 32;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = ld4(bias + t * 16 + g * 4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a[2 * kb + (j >> 2)][j & 3];
        f16x8 bh, bl;
        split2w(x, bh, bl, wit);
        kblock_h2<NT>(acc, wh, wl, kb, g, rl, bh, bl);
    }
    const f32x2 inv = {kH2InvScale, kH2InvScale};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
        }
        a[t].lo = acc[t].lo * inv;
        a[t].hi = acc[t].hi * inv;
    }
}

// max over the four 16-lane groups of a wavefront (lanes l, l^16, l^32, l^48), full EXEC mask required
__device__ __forceinline__ float max_over_lane_groups16_swap(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float s = fmaxf(a, b), t = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s), "+v"(t));
    return fmaxf(s, t);
}

// The power of two a gradient row is normalised by before its fp16 split (m = the row's largest magnitude = f * 2^e,
// f in [0.5, 1)).  Clamped to +-100: 2^-e and 2^e stay normal fp32 numbers for a row of subnormal junk (m ~ 1e-40 made
// 2^-e = inf and the row NaN instead of ~0) as for an absurdly large one.
__device__ __forceinline__ int h2_row_exponent(float m) {
    const int e = m > 0.f ? __builtin_amdgcn_frexp_expf(m) : 0;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// kblock_h2_side for NP row tiles at once (one fragment fetch feeds NP independent chains; per output tile the MFMAs and
// their order are those of kblock_h2_side).
template <int NT, int NP>
__device__ __forceinline__ void kblock_h2_side_multi(f32x4 (&acc)[NP][NT], f32x4 (&side)[NP][NT], const _Float16* wh,
                                                     const _Float16* wl, int kb, int g, int jl, const f16x8 (&bh)[NP],
                                                     const f16x8 (&bm)[NP]) {
    const int off = ((kb * 4 + g) * NT * 16 + jl) * 8;
    constexpr int PF = H2_PF < NT ? H2_PF : NT;
    f16x8 ah[PF + 1], al[PF + 1];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        ah[p] = ldw(wh + off + p * 128);
        al[p] = ldw(wl + off + p * 128);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + PF < NT) {
            ah[(t + PF) % (PF + 1)] = ldw(wh + off + (t + PF) * 128);
            al[(t + PF) % (PF + 1)] = ldw(wl + off + (t + PF) * 128);
        }
        const f16x8 a_h = ah[t % (PF + 1)], a_l = al[t % (PF + 1)];
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            side[n][t] = MFMA_F16(a_h, bm[n], side[n][t]);
            f32x4 c = acc[n][t];
            c = MFMA_F16(a_l, bh[n], c);
            c = MFMA_F16(a_h, bh[n], c);
            acc[n][t] = c;
        }
    }
}

// bytes -> LDS, 16 bytes per lane, straight from global memory (global_load_lds_dwordx4); the LDS address of a lane is
// the wavefront's base + lane*16.  Callers follow up with h2_stage_wait() + a barrier.
__device__ __forceinline__ void h2_copy_to_lds(void* dst, const void* __restrict__ src, int nbytes, int tid, int nthreads) {
    const int lane = tid & 63, n16 = nbytes >> 4;
    const char* s = reinterpret_cast<const char*>(src);
    char* d = reinterpret_cast<char*>(dst);
    for (int idx = tid; idx - lane < n16; idx += nthreads) {
        if (idx < n16)
            __builtin_amdgcn_global_load_lds(s + (size_t)idx * 16,
                                             (__attribute__((address_space(3))) void*)(d + (size_t)(idx - lane) * 16), 16, 0, 0);
    }
}
__device__ __forceinline__ void h2_stage_wait() { __builtin_amdgcn_s_waitcnt(0); }

}  // namespace tspgnn
