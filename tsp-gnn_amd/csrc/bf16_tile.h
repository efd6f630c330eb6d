// Device-side helpers shared by the bf16-storage kernels (bf16.hip: forward; dense_bwd_bf16.hip: backward): bf16 vector
// loads / stores, the blocked projected-message and state layouts, widening / narrowing.
#pragma once
#include "common.h"
#include "mfma_tile.h"

namespace tspgnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x8 ldw8(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x4 ldw4(const __bf16* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ void stw4(__bf16* p, bf16x4 v) { *reinterpret_cast<bf16x4*>(p) = v; }

// The bf16 projected-message format (Zx between the vertex MLP's projection and the edge cell's gather): [rows padded to 16,
// 4D] bf16, blocked by 16 rows like the f16x2 path's (h2_tile.h) -- the four bf16 (cols 16t+4g..+3) of row v at
// (((v/16)*NT4 + t)*4 + g)*64 + (v%16)*4 -- so a gather instruction of 16 consecutive far endpoints reads 512 contiguous
// bytes instead of 16 rows.  zx_blocked(v, g): offset of (v, t = 0, g); add 256 elements per tile t.
template <int D>
__device__ __forceinline__ size_t zx_blocked(unsigned v, int g) {
    return (size_t)(v >> 4) * (D / 4 * 256) + (unsigned)g * 64u + (v & 15u) * 4u;
}
// The same blocking for the fp32 cell state c of the loop's ping-pong buffers (read and written by the cell only).
template <int D>
__device__ __forceinline__ size_t c_blocked(unsigned r, int g, bool blocked) {
    return blocked ? (size_t)(r >> 4) * (D / 16 * 256) + (unsigned)g * 64u + (r & 15u) * 4u : (size_t)r * D + (unsigned)g * 4u;
}

__device__ __forceinline__ f32x4 widen(bf16x4 v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
__device__ __forceinline__ bf16x4 narrow(f32x4 v) { return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; }
__device__ __forceinline__ bf16x8 join(bf16x4 lo, bf16x4 hi) {
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the lane's B operand of k-block kb from a bf16 row: features 16*(2kb)+4g.. and 16*(2kb+1)+4g.. (row + g*4 given)
__device__ __forceinline__ bf16x8 row_operand(const __bf16* row_g, int kb) {
    return join(ldw4(row_g + kb * 32), ldw4(row_g + kb * 32 + 16));
}
// the same from a row stored BLOCKED by 16 rows (c_blocked's element offsets: 256 elements between the 16-column tiles)
__device__ __forceinline__ bf16x8 row_operand(const __bf16* row_g, int kb, bool blocked) {
    return blocked ? join(ldw4(row_g + kb * 512), ldw4(row_g + kb * 512 + 256)) : row_operand(row_g, kb);
}

// acc[t] += sum_kb W(kb, T0 + t) x bv[kb] for the TN output tiles T0.. of a packed bf16 matrix with NTOT output tiles
// (LDS; base = the lane's fragment of (kb = 0, tile 0): + kb * 4 * NTOT * 128 + t * 128 elements), the weight fragments
// PF deep in flight ahead of the MFMAs: an LDS read takes ~100 cycles, an MFMA 16 -- read-then-multiply in program order
// (what the compiler emits for the plain loop) leaves the matrix pipe waiting on every fragment.
template <int NTOT, int T0, int TN, int KB, int PF>
__device__ __forceinline__ void gemm_frags_pf(f32x4 (&acc)[TN], const __bf16* base, const bf16x8 (&bv)[KB]) {
    constexpr int TOTAL = KB * TN;
    bf16x8 a[PF + 1];
#pragma unroll
    for (int i = 0; i < PF && i < TOTAL; ++i) a[i] = ldw8(base + (i / TN) * (4 * NTOT * 128) + (T0 + i % TN) * 128);
#pragma unroll
    for (int i = 0; i < TOTAL; ++i) {
        if (i + PF < TOTAL) a[(i + PF) % (PF + 1)] = ldw8(base + ((i + PF) / TN) * (4 * NTOT * 128) + (T0 + (i + PF) % TN) * 128);
        acc[i % TN] = MFMA_BF16(a[i % (PF + 1)], bv[i / TN], acc[i % TN]);
    }
}

}  // namespace tspgnn
