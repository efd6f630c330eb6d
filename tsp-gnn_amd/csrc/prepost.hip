// Once-per-batch pieces around the message-passing loop (model.py:33-51 and model.py:107-157):
// edge-embedding initialisation, vertex-embedding tiling, the vote head's final Dense(1),
// per-problem segment mean, and sigmoid / cross-entropy / confusion counts.
#include "common.h"
#include "mfma_tile.h"

namespace tspgnn {

// ------------------------------------------------------------- E0 = E_init_MLP([W, C])
// 256 edges per workgroup.  Phase 1: one thread per edge runs the two narrow layers (2 -> D/8 -> D/4, weights read with
// wave-uniform indices = scalar loads) and parks its D/4 activations in LDS.  Phase 2 (round 4): the two wide layers
// (D/4 -> D/2 -> D: 97 % of the flops and ALL of the output bytes) on the fp32 matrix instruction
// (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains): a wavefront takes 16 edges at a time, forms a3 = relu(a2 W3 + b3) in the
// accumulator layout, turns it into the next product's operand layout through its own 4 KB of LDS, and stores the
// 16 x D tile of a4 = a3 W4 + b4 as 64-byte runs (four rows per store instruction).  (Before: D/4 lanes per row with W4
// in registers / LDS and scalar FMAs: 17 us at C2 for a 25.6 MB array, 410 us at the C5 shard.)
template <int D>
__global__ __launch_bounds__(256) void einit_fwd_kernel(const float2* __restrict__ WC, const float* __restrict__ wb,
                                                        float* __restrict__ E0, int M) {
    constexpr int H1 = D / 8, H2 = D / 4, H3 = D / 2;
    const float* W1 = wb;
    const float* b1 = W1 + 2 * H1;
    const float* W2 = b1 + H1;
    const float* b2 = W2 + H1 * H2;
    const float* W3 = b2 + H2;
    const float* b3 = W3 + H2 * H3;
    const float* W4 = b3 + H3;
    const float* b4 = W4 + H3 * D;
    __shared__ float s_a2[256][H2 + 1];
    __shared__ float s_w3[H2][H3 + 1], s_w4[H3][D + 1];
    __shared__ float s_a3[4][16][H3 + 1];   // one tile per wavefront
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, lk = lane >> 4;
    const int e0 = blockIdx.x * 256;
    for (int i = tid; i < H2 * H3; i += 256) s_w3[i / H3][i % H3] = W3[i];
    for (int i = tid; i < H3 * D; i += 256) s_w4[i / D][i % D] = W4[i];
    {
        const int e = min(e0 + tid, M - 1);
        const float2 wc = WC[e];
        const float w = wc.x, c = wc.y;
        float a1[H1];
#pragma unroll
        for (int j = 0; j < H1; ++j) a1[j] = fmaxf(fmaf(c, W1[H1 + j], fmaf(w, W1[j], 0.f)) + b1[j], 0.f);
#pragma unroll
        for (int j = 0; j < H2; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < H1; ++k) s = fmaf(a1[k], W2[k * H2 + j], s);
            s_a2[tid][j] = fmaxf(s + b2[j], 0.f);
        }
    }
    __syncthreads();
    float bias3[H3 / 16], bias4[D / 16];
#pragma unroll
    for (int nt = 0; nt < H3 / 16; ++nt) bias3[nt] = b3[16 * nt + lm];
#pragma unroll
    for (int nt = 0; nt < D / 16; ++nt) bias4[nt] = b4[16 * nt + lm];
    for (int et = 0; et < 4; ++et) {
        const int r0 = 64 * wave + 16 * et;            // first edge (within the workgroup) of this 16-edge tile
        if (e0 + r0 >= M) break;                       // wave-uniform
        f32x4 c3[H3 / 16];
#pragma unroll
        for (int nt = 0; nt < H3 / 16; ++nt) c3[nt] = f32x4{bias3[nt], bias3[nt], bias3[nt], bias3[nt]};
#pragma unroll
        for (int ks = 0; ks < H2 / 4; ++ks) {
            const float a = s_a2[r0 + lm][4 * ks + lk];
#pragma unroll
            for (int nt = 0; nt < H3 / 16; ++nt) c3[nt] = MFMA16(a, s_w3[4 * ks + lk][16 * nt + lm], c3[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < H3 / 16; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s_a3[wave][4 * lk + i][16 * nt + lm] = fmaxf(c3[nt][i], 0.f);
        }
        // (the tile is private to the wavefront and LDS operations of one wavefront complete in order: no barrier)
        f32x4 c4[D / 16];
#pragma unroll
        for (int nt = 0; nt < D / 16; ++nt) c4[nt] = f32x4{bias4[nt], bias4[nt], bias4[nt], bias4[nt]};
#pragma unroll 4
        for (int ks = 0; ks < H3 / 4; ++ks) {
            const float a = s_a3[wave][lm][4 * ks + lk];
#pragma unroll
            for (int nt = 0; nt < D / 16; ++nt) c4[nt] = MFMA16(a, s_w4[4 * ks + lk][16 * nt + lm], c4[nt]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + r0 + 4 * lk + i;
            if (e < M) {
#pragma unroll
                for (int nt = 0; nt < D / 16; ++nt) E0[(size_t)e * D + 16 * nt + lm] = c4[nt][i];
            }
        }
    }
}

// ------------------------------------------------------------- Y[r,:] = scale * v
__global__ __launch_bounds__(256) void tile_rows_kernel(const float* __restrict__ v, float scale,
                                                        float4* __restrict__ Y, long long total4, int d4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const int c = (int)(i % d4);
        const float4 s = *reinterpret_cast<const float4*>(v + 4 * c);
        Y[i] = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
    }
}

// ------------------------------------------------------------- y[r] = X[r,:].w + b
// LPR = d/4 lanes per row, butterfly reduce inside the lane group (fixed order).
template <int LPR>
__global__ __launch_bounds__(256) void rowdot_kernel(const float4* __restrict__ X, const float4* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int rows) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = gid / LPR;
    const int c = (int)(gid % LPR);
    float s = 0.f;
    if (r < rows) {
        const float4 xv = X[r * LPR + c];
        const float4 wv = w[c];
        s = fmaf(xv.w, wv.w, fmaf(xv.z, wv.z, fmaf(xv.y, wv.y, xv.x * wv.x)));
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off);
    if (r < rows && c == 0) y[r] = s + b[0];
}

__global__ __launch_bounds__(256) void rowdot_generic_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ y,
                                                             int rows, int d) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int k = 0; k < d; ++k) s = fmaf(X[(size_t)r * d + k], w[k], s);
    y[r] = s + b[0];
}

// ------------------------------------------------------------- per-problem mean of edge votes
// One workgroup of four wavefronts per problem: thread-strided partial sums in four independent chains (a problem of
// n = 200 has 19 900 votes: one dependent chain per lane was 80 us of load latency), then a fixed-order reduction --
// butterfly within the wavefront, the four wavefronts in order through LDS.  Deterministic.
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ vote, const int* __restrict__ seg,
                                                           float* __restrict__ logits, int B) {
    __shared__ float part[4];
    const int p = blockIdx.x;
    if (p >= B) return;
    const int beg = seg[p], end = seg[p + 1];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = beg + (int)threadIdx.x;
    for (; k + 768 < end; k += 1024) {
        s0 += vote[k];
        s1 += vote[k + 256];
        s2 += vote[k + 512];
        s3 += vote[k + 768];
    }
    for (; k < end; k += 256) s0 += vote[k];
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        logits[p] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)(end - beg);  // 0/0 = NaN like tf.reduce_mean([])
}

// ------------------------------------------------------------- sigmoid, BCE, confusion counts
// Single workgroup (B is the number of problems, a few thousand at most).
__global__ __launch_bounds__(256) void bce_metrics_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ labels, float* __restrict__ pred,
                                                          float* __restrict__ stats, int B) {
    __shared__ float red[6][256];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float x = logits[i], z = labels[i];
        const float p = sigmoidf_(x);
        pred[i] = p;
        // tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))
        acc[0] += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
        const float eq = (z == rintf(p)) ? 1.f : 0.f;  // tf.round: half to even
        const float ne = 1.f - eq;
        acc[1] += eq;             // acc
        acc[2] += z * eq;         // 'TP'  (model.py:150)
        acc[3] += z * ne;         // 'FP'  (model.py:151, formula kept verbatim)
        acc[4] += (1.f - z) * eq; // 'TN'  (model.py:152)
        acc[5] += (1.f - z) * ne; // 'FN'  (model.py:153)
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) {
        const float v = red[threadIdx.x][0];
        stats[threadIdx.x] = (threadIdx.x < 2) ? v / (float)B : v;
    }
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_einit_fwd_f32(const float* WC, const float* wb, float* E0, int M, int d, void* stream) {
    TSPGNN_REQUIRE(M >= 0, "einit_fwd: M=%d", M);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "einit_fwd: d=%d must be 32, 64 or 128", d);
    if (M == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(WC && wb && E0, "einit_fwd: null pointer");
    const float2* WC2 = reinterpret_cast<const float2*>(WC);
    const unsigned grid = (unsigned)((M + 255) / 256);
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 32: einit_fwd_kernel<32><<<grid, 256, 0, st>>>(WC2, wb, E0, M); break;
        case 64: einit_fwd_kernel<64><<<grid, 256, 0, st>>>(WC2, wb, E0, M); break;
        default: einit_fwd_kernel<128><<<grid, 256, 0, st>>>(WC2, wb, E0, M); break;
    }
    return launched("tspgnn_einit_fwd_f32");
}

extern "C" int tspgnn_tile_rows_f32(const float* v, float scale, float* Y, int rows, int d, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "tile_rows: rows=%d", rows);
    TSPGNN_REQUIRE(d > 0 && d % 4 == 0, "tile_rows: d=%d must be a positive multiple of 4", d);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(v && Y, "tile_rows: null pointer");
    const long long total4 = (long long)rows * (d / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    tile_rows_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(v, scale, reinterpret_cast<float4*>(Y), total4,
                                                                      d / 4);
    return launched("tspgnn_tile_rows_f32");
}

extern "C" int tspgnn_rowdot_f32(const float* X, const float* w, const float* b, float* y, int rows, int d,
                                 void* stream) {
    TSPGNN_REQUIRE(rows >= 0 && d > 0, "rowdot: rows=%d d=%d", rows, d);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(X && w && b && y, "rowdot: null pointer");
    hipStream_t st = as_stream(stream);
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    auto grid = [&](int lpr) { return (unsigned)(((long long)rows * lpr + 255) / 256); };
    switch (d) {
        case 32: rowdot_kernel<8><<<grid(8), 256, 0, st>>>(X4, w4, b, y, rows); break;
        case 64: rowdot_kernel<16><<<grid(16), 256, 0, st>>>(X4, w4, b, y, rows); break;
        case 128: rowdot_kernel<32><<<grid(32), 256, 0, st>>>(X4, w4, b, y, rows); break;
        case 256: rowdot_kernel<64><<<grid(64), 256, 0, st>>>(X4, w4, b, y, rows); break;
        default: rowdot_generic_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, st>>>(X, w, b, y, rows, d);
    }
    return launched("tspgnn_rowdot_f32");
}

extern "C" int tspgnn_segment_mean_f32(const float* vote, const int32_t* seg, float* logits, int B, void* stream) {
    TSPGNN_REQUIRE(B >= 0, "segment_mean: B=%d", B);
    if (B == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(vote && seg && logits, "segment_mean: null pointer");
    segment_mean_kernel<<<(unsigned)B, 256, 0, as_stream(stream)>>>(vote, seg, logits, B);
    return launched("tspgnn_segment_mean_f32");
}

extern "C" int tspgnn_bce_metrics_f32(const float* logits, const float* labels, float* pred, float* stats, int B,
                                      void* stream) {
    TSPGNN_REQUIRE(B >= 0, "bce_metrics: B=%d", B);
    TSPGNN_REQUIRE(logits && labels && pred && stats, "bce_metrics: null pointer");
    bce_metrics_kernel<<<1, 256, 0, as_stream(stream)>>>(logits, labels, pred, stats, B);
    return launched("tspgnn_bce_metrics_f32");
}
