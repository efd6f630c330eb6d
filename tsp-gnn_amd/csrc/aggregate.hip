// The vertex<->edge aggregation (the reference's two dense tf.matmul on EV, graphnn.py:156-160)
// as pattern-only gather / CSR row-sum kernels for gfx950.  HBM-bound: every [M,d] row is
// touched exactly once with 16-byte lanes (a 64-float row = one 256 B coalesced segment,
// 16 lanes), the small [N,d] operand stays L2-resident.
#include "common.h"

namespace tspgnn {

// ---------------------------------------------------------------- E <- V : gather, 2 nnz / row
// One float4 per thread; a row of d floats is covered by d/4 adjacent lanes, so a wave64
// writes 64*16 B = 1 KiB of contiguous Y per store instruction.
__device__ __forceinline__ void gather2_sum_body(const int2* __restrict__ uv, const float4* __restrict__ X,
                                                 float4* __restrict__ Y, int M, int d4, unsigned blk, unsigned nblk) {
    const long long total = (long long)M * d4;
    const long long stride = (long long)nblk * blockDim.x;
    for (long long i = (long long)blk * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int e = (int)(i / d4);
        const int c = (int)(i - (long long)e * d4);
        const int2 ends = uv[e];
        const float4 a = X[(long long)ends.x * d4 + c];
        const float4 b = X[(long long)ends.y * d4 + c];
        f32x4 r;
        r[0] = a.x + b.x;
        r[1] = a.y + b.y;
        r[2] = a.z + b.z;
        r[3] = a.w + b.w;
        // streamed once, consumed by a later kernel: non-temporal store keeps X (re-read by every
        // workgroup) resident in L2
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(Y) + i);
    }
}

__global__ __launch_bounds__(256) void gather2_sum_kernel(const int2* __restrict__ uv, const float4* __restrict__ X,
                                                          float4* __restrict__ Y, int M, int d4) {
    gather2_sum_body(uv, X, Y, M, d4, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------- V <- E : CSR row-sum
// One wavefront per vertex.  LPR = d/4 lanes cover one edge row (float4 each), so the wave
// reads RPW = 64/LPR edge rows per step; lane-group s accumulates the edges k = s (mod RPW)
// in ascending order, the groups are then combined with wavefront shuffles in a fixed
// order (deterministic result).  Edge ids of the row are fetched with one coalesced load
// per 64 edges and broadcast by shuffle instead of one dependent scalar load per edge.
// CB > 1: rows wider than 16 float4 are split into CB column blocks of LPR float4, one wavefront each (adjacent
// wavefronts of a workgroup): four times the wavefronts and RPW rows in flight per load instead of one 1 KiB row at a
// time -- a wide row-sum is otherwise bound by the latency of its ~n dependent-in-order batches of loads.
template <int LPR, bool VALUED, int CB = 1>
__device__ __forceinline__ void csr_rowsum_body(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                const float* __restrict__ val, const float4* __restrict__ X,
                                                float4* __restrict__ Y, int N, unsigned blk, unsigned nblk) {
    constexpr int RPW = kWave / LPR;
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  Give each
    // XCD one contiguous eighth of the vertices, so the ~n vertices of one graph -- which together read
    // every edge row of that graph TWICE (once per endpoint) -- share one L2 and the second read hits.
    const unsigned nb = nblk, q = nb >> 3, r = nb & 7, xcd = blk & 7, slot = blk >> 3;
    const unsigned vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;  // bijective for any nb
    const int wv = (int)(((long long)vb * blockDim.x + threadIdx.x) >> 6);
    const int v = wv / CB;
    if (v >= N) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int c = (wv % CB) * LPR + lane % LPR;   // float4 column within the LPR*CB-wide row
    const int beg = rowptr[v], end = rowptr[v + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = beg; base < end; base += kWave) {
        const int cnt = min(kWave, end - base);
        const int my_e = (lane < cnt) ? eid[base + lane] : 0;
        float my_w = 1.0f;
        if (VALUED) my_w = (lane < cnt) ? val[base + lane] : 0.f;
#pragma unroll 4
        for (int k0 = 0; k0 < cnt; k0 += RPW) {
            const int k = k0 + sub;
            const int src = min(k, cnt - 1);
            const int e = __shfl(my_e, src);
            const float w = VALUED ? __shfl(my_w, src) : 1.0f;
            if (k < cnt) {
                const float4 x = X[(long long)e * (LPR * CB) + c];
                if (VALUED) {
                    acc.x = fmaf(w, x.x, acc.x);
                    acc.y = fmaf(w, x.y, acc.y);
                    acc.z = fmaf(w, x.z, acc.z);
                    acc.w = fmaf(w, x.w, acc.w);
                } else {
                    acc.x += x.x;
                    acc.y += x.y;
                    acc.z += x.z;
                    acc.w += x.w;
                }
            }
        }
    }
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    if (sub == 0) Y[(long long)v * (LPR * CB) + c] = acc;
}

template <int LPR, bool VALUED, int CB = 1>
__global__ __launch_bounds__(256) void csr_rowsum_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                         const float* __restrict__ val, const float4* __restrict__ X,
                                                         float4* __restrict__ Y, int N) {
    csr_rowsum_body<LPR, VALUED, CB>(rowptr, eid, val, X, Y, N, blockIdx.x, gridDim.x);
}

// Both directions of one message-passing step's aggregation in ONE launch: E <- V gather and V <- E
// row-sum read disjoint inputs and write disjoint outputs (both updates read the OLD states,
// graphnn.py:143), so their workgroups can share the chip and the HBM pipe instead of paying two
// launch latencies for ~5 us of streaming each.  Workgroups [0, nb_rowsum) run the row-sum (a multiple
// of 8 keeps its XCD-contiguous vertex order), the rest the gather.
template <int LPR, int CB = 1>
__global__ __launch_bounds__(256) void spmm_pair_kernel(const int2* __restrict__ uv, const float4* __restrict__ Xv,
                                                        float4* __restrict__ Ye, int M, const int* __restrict__ rowptr,
                                                        const int* __restrict__ eid, const float4* __restrict__ Xe,
                                                        float4* __restrict__ Yv, int N, unsigned nb_rowsum) {
    if (blockIdx.x < nb_rowsum)
        csr_rowsum_body<LPR, false, CB>(rowptr, eid, nullptr, Xe, Yv, N, blockIdx.x, nb_rowsum);
    else
        gather2_sum_body(uv, Xv, Ye, M, LPR * CB, blockIdx.x - nb_rowsum, gridDim.x - nb_rowsum);
}

// Any d that is a multiple of 4 but not 32/64/128/256: one thread per (row, float4 column).
template <bool VALUED>
__global__ __launch_bounds__(256) void csr_rowsum_generic_kernel(const int* __restrict__ rowptr,
                                                                 const int* __restrict__ eid,
                                                                 const float* __restrict__ val,
                                                                 const float4* __restrict__ X,
                                                                 float4* __restrict__ Y, int N, int d4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * d4) return;
    const int v = (int)(i / d4), c = (int)(i % d4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
        const float4 x = X[(long long)eid[k] * d4 + c];
        const float w = VALUED ? val[k] : 1.0f;
        acc.x = VALUED ? fmaf(w, x.x, acc.x) : acc.x + x.x;
        acc.y = VALUED ? fmaf(w, x.y, acc.y) : acc.y + x.y;
        acc.z = VALUED ? fmaf(w, x.z, acc.z) : acc.z + x.z;
        acc.w = VALUED ? fmaf(w, x.w, acc.w) : acc.w + x.w;
    }
    Y[i] = acc;
}

template <bool VALUED>
static int launch_csr(const int32_t* rowptr, const int32_t* idx, const float* val, const float* X,
                      float* Y, int R, int d, hipStream_t st) {
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4* Y4 = reinterpret_cast<float4*>(Y);
    const int d4 = d / 4;
    const unsigned grid = (unsigned)(((long long)R * kWave + 255) / 256);
    auto grid_cb = [&](int cb) { return (unsigned)(((long long)R * cb * kWave + 255) / 256); };
    switch (d4) {
        case 8: csr_rowsum_kernel<8, VALUED><<<grid, 256, 0, st>>>(rowptr, idx, val, X4, Y4, R); break;
        case 16: csr_rowsum_kernel<16, VALUED><<<grid, 256, 0, st>>>(rowptr, idx, val, X4, Y4, R); break;
        case 32: csr_rowsum_kernel<16, VALUED, 2><<<grid_cb(2), 256, 0, st>>>(rowptr, idx, val, X4, Y4, R); break;
        case 64: csr_rowsum_kernel<16, VALUED, 4><<<grid_cb(4), 256, 0, st>>>(rowptr, idx, val, X4, Y4, R); break;
        default: {
            const unsigned g2 = (unsigned)(((long long)R * d4 + 255) / 256);
            csr_rowsum_generic_kernel<VALUED><<<g2, 256, 0, st>>>(rowptr, idx, val, X4, Y4, R, d4);
        }
    }
    return launched(VALUED ? "tspgnn_csr_spmm_f32" : "tspgnn_csr_rowsum_f32");
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_gather2_sum_f32(const int32_t* uv, const float* X, float* Y, int M, int N, int d,
                                      void* stream) {
    TSPGNN_REQUIRE(M >= 0 && N >= 0, "gather2_sum: negative size (M=%d N=%d)", M, N);
    TSPGNN_REQUIRE(d > 0 && d % 4 == 0, "gather2_sum: d=%d must be a positive multiple of 4", d);
    if (M == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(uv && X && Y, "gather2_sum: null pointer");
    const int d4 = d / 4;
    const long long total = (long long)M * d4;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
    gather2_sum_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const int2*>(uv), reinterpret_cast<const float4*>(X), reinterpret_cast<float4*>(Y), M,
        d4);
    return launched("tspgnn_gather2_sum_f32");
}

extern "C" int tspgnn_spmm_pair_f32(const int32_t* uv, const float* Xv, float* Ye, const int32_t* rowptr,
                                    const int32_t* eid, const float* Xe, float* Yv, int M, int N, int d, void* stream) {
    TSPGNN_REQUIRE(M >= 0 && N >= 0, "spmm_pair: negative size (M=%d N=%d)", M, N);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128 || d == 256, "spmm_pair: d=%d must be 32, 64, 128 or 256", d);
    if (M == 0 || N == 0) {  // degenerate: fall back to the two single-direction entry points
        int rc = tspgnn_gather2_sum_f32(uv, Xv, Ye, M, N, d, stream);
        return rc ? rc : tspgnn_csr_rowsum_f32(rowptr, eid, Xe, Yv, N, M, d, stream);
    }
    TSPGNN_REQUIRE(uv && Xv && Ye && rowptr && eid && Xe && Yv, "spmm_pair: null pointer");
    const int d4 = d / 4;
    const int cb = d4 > 16 ? d4 / 16 : 1;   // column blocks of the row-sum (see csr_rowsum_body)
    unsigned nb_rowsum = (unsigned)(((long long)N * cb * kWave + 255) / 256);
    long long nb_gather = ((long long)M * d4 + 255) / 256;
    if (nb_gather > 256 * 16) nb_gather = 256 * 16;
    const unsigned grid = nb_rowsum + (unsigned)nb_gather;
    hipStream_t st = as_stream(stream);
    const int2* uv2 = reinterpret_cast<const int2*>(uv);
    const float4* Xv4 = reinterpret_cast<const float4*>(Xv);
    const float4* Xe4 = reinterpret_cast<const float4*>(Xe);
    float4* Ye4 = reinterpret_cast<float4*>(Ye);
    float4* Yv4 = reinterpret_cast<float4*>(Yv);
    switch (d4) {
        case 8: spmm_pair_kernel<8><<<grid, 256, 0, st>>>(uv2, Xv4, Ye4, M, rowptr, eid, Xe4, Yv4, N, nb_rowsum); break;
        case 16: spmm_pair_kernel<16><<<grid, 256, 0, st>>>(uv2, Xv4, Ye4, M, rowptr, eid, Xe4, Yv4, N, nb_rowsum); break;
        case 32: spmm_pair_kernel<16, 2><<<grid, 256, 0, st>>>(uv2, Xv4, Ye4, M, rowptr, eid, Xe4, Yv4, N, nb_rowsum); break;
        default: spmm_pair_kernel<16, 4><<<grid, 256, 0, st>>>(uv2, Xv4, Ye4, M, rowptr, eid, Xe4, Yv4, N, nb_rowsum); break;
    }
    return launched("tspgnn_spmm_pair_f32");
}

extern "C" int tspgnn_csr_rowsum_f32(const int32_t* rowptr, const int32_t* eid, const float* X, float* Y, int N,
                                     int M, int d, void* stream) {
    TSPGNN_REQUIRE(M >= 0 && N >= 0, "csr_rowsum: negative size (N=%d M=%d)", N, M);
    TSPGNN_REQUIRE(d > 0 && d % 4 == 0, "csr_rowsum: d=%d must be a positive multiple of 4", d);
    if (N == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(rowptr && Y && (M == 0 || (eid && X)), "csr_rowsum: null pointer");
    return launch_csr<false>(rowptr, eid, nullptr, X, Y, N, d, as_stream(stream));
}

extern "C" int tspgnn_csr_spmm_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                                   float* Y, int R, int C, int d, void* stream) {
    TSPGNN_REQUIRE(R >= 0 && C >= 0, "csr_spmm: negative size (R=%d C=%d)", R, C);
    TSPGNN_REQUIRE(d > 0 && d % 4 == 0, "csr_spmm: d=%d must be a positive multiple of 4", d);
    if (R == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(rowptr && Y && (C == 0 || (col && val && X)), "csr_spmm: null pointer");
    return launch_csr<true>(rowptr, col, val, X, Y, R, d, as_stream(stream));
}
