// Dense per-row updates of the message-passing step on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_16x16x4_f32 == an fmaf chain, so results stay inside the 1e-5 parity budget):
//   * mlp_fwd     : the chained tf.layers.Dense of a message MLP (mlp.py:57-63, graphnn.py:153)
//   * lnlstm_fwd  : LayerNormBasicLSTMCell step (graphnn.py:168-170)
//
// Layout idea shared by both kernels ("transposed chaining").  A wavefront owns a tile of 16
// rows and computes OUT^T = W^T * IN^T with the 16x16x4 MFMA: the MFMA's A operand is a weight
// fragment (read from LDS), its B operand is the activation fragment, and the D result has
//      lane (rl = lane&15, g = lane>>4), register r of output tile t  <->  OUT[row rl][t*16+g*4+r].
// A lane therefore holds, for ITS row, the features {t*16 + g*4 + r}: exactly what the next
// layer's B operand wants if k-step s = 4*q+p of that layer multiplies feature
// f(s,g) = q*16 + g*4 + p.  The weights are stored in LDS with their rows permuted accordingly,
// so activations never leave registers between layers and never cross lanes.  The same
// fragment shape (float4 at column q*16+g*4 of a row) is used for global loads and stores.
#include "common.h"

namespace tspgnn {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Copies a [krows, ncols] row-major weight matrix into LDS in MFMA A-fragment order:
//   dst[(((s*4+g)*U + u)*16 + jl)*4 + tt] = W[krow(s,g)][(u*4+tt)*16 + jl],
//   krow(s,g) = (s>>2)*16 + g*4 + (s&3),  U = ncols/64.
// One ds_read_b128 at dst + ((s*4+g)*U+u)*64 + jl*4 then feeds four MFMAs (tiles 4u..4u+3)
// and is bank-conflict free (16 lanes x 16 B = one 256 B bank row per lane group).
__device__ __forceinline__ void stage_weights_b128(float* dst, const float* __restrict__ W, int krows, int ncols,
                                                   int tid, int nthreads) {
    const int U = ncols >> 6;
    const int n4 = (krows * ncols) >> 2;
    for (int i = tid; i < n4; i += nthreads) {
        const int jl = i & 15;
        const int u = (i >> 4) % U;
        const int sg = (i >> 4) / U;
        const int g = sg & 3, s = sg >> 2;
        const int krow = ((s >> 2) << 4) + (g << 2) + (s & 3);
        const float* src = W + (size_t)krow * ncols + (u << 6) + jl;
        f32x4 v;
        v[0] = src[0];
        v[1] = src[16];
        v[2] = src[32];
        v[3] = src[48];
        st4(dst + (size_t)i * 4, v);
    }
}

// Same for ncols == 32 (two output tiles): dst[((s*4+g)*16 + jl)*2 + tt].
__device__ __forceinline__ void stage_weights_b64(float* dst, const float* __restrict__ W, int krows, int tid,
                                                  int nthreads) {
    const int n2 = (krows * 32) >> 1;
    for (int i = tid; i < n2; i += nthreads) {
        const int jl = i & 15;
        const int sg = i >> 4;
        const int g = sg & 3, s = sg >> 2;
        const int krow = ((s >> 2) << 4) + (g << 2) + (s & 3);
        const float* src = W + (size_t)krow * 32 + jl;
        dst[2 * i] = src[0];
        dst[2 * i + 1] = src[16];
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

// ---------------------------------------------------------------------------------- MLP
// Persistent workgroups: all layer weights (permuted) + biases live in LDS for the lifetime of
// the block; each wavefront pulls 16-row tiles from the block's contiguous tile range through
// an LDS ticket counter (keeps the four SIMDs of a CU evenly loaded at the tail).
template <int D, int MAXL>
__global__ __launch_bounds__(512) void mlp_fwd_kernel(const float* __restrict__ X, const float* __restrict__ wb,
                                                      float* __restrict__ Y, float* __restrict__ acts, int rows,
                                                      int n_layers, unsigned relu_mask, int tiles_total) {
    constexpr int NT = D / 16;
    __shared__ __attribute__((aligned(16))) float lds[MAXL * (D * D + D) + 4];
    float* lds_w = lds;
    float* lds_b = lds + MAXL * D * D;
    int* ticket = reinterpret_cast<int*>(lds + MAXL * (D * D + D));

    const int tid = threadIdx.x;
    for (int l = 0; l < n_layers; ++l) {
        const float* Wl = wb + (size_t)l * (D * D + D);
        if constexpr (NT == 2)
            stage_weights_b64(lds_w + l * D * D, Wl, D, tid, blockDim.x);
        else
            stage_weights_b128(lds_w + l * D * D, Wl, D, D, tid, blockDim.x);
        for (int i = tid; i < D; i += blockDim.x) lds_b[l * D + i] = Wl[D * D + i];
    }
    const int t_beg = (int)((long long)tiles_total * blockIdx.x / gridDim.x);
    const int t_end = (int)((long long)tiles_total * (blockIdx.x + 1) / gridDim.x);
    if (tid == 0) *ticket = t_beg;
    __syncthreads();

    const int lane = tid & 63, rl = lane & 15, g = lane >> 4;
    for (;;) {
        int tile = 0;
        if (lane == 0) tile = atomicAdd(ticket, 1);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= t_end) break;
        const int row = tile * 16 + rl;
        const bool valid = row < rows;
        const size_t rbase = (size_t)(valid ? row : rows - 1) * D + g * 4;
        f32x4 a[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) a[q] = ld4(X + rbase + q * 16);
        for (int l = 0; l < n_layers; ++l) {
            const float* wl = lds_w + l * D * D;
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = ld4(lds_b + l * D + t * 16 + g * 4);
#pragma unroll
            for (int s = 0; s < D / 4; ++s) kstep<NT>(acc, wl + frag_off<NT>(s, g, rl), a[s >> 2][s & 3]);
            const bool relu = (relu_mask >> l) & 1u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
                }
                a[t] = acc[t];
            }
            if (acts != nullptr && l < n_layers - 1 && valid) {
                float* dst = acts + (size_t)l * rows * D + rbase;
#pragma unroll
                for (int t = 0; t < NT; ++t) st4(dst + t * 16, a[t]);
            }
        }
        if (valid) {
#pragma unroll
            for (int t = 0; t < NT; ++t) st4(Y + rbase + t * 16, a[t]);
        }
    }
}

// ---------------------------------------------------------------------------------- LN-LSTM
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
    const float rstd = 1.0f / sqrtf(var + 1e-12f);
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

// K ([dx+D, 4D], permuted) and the five LayerNorm (gamma,beta) pairs stay resident in LDS;
// requires (dx+D)*4D*4 + 10*D*4 + 16 bytes <= 160 KiB (D=64, dx=64: 130.5 KiB).
template <int D>
__global__ __launch_bounds__(512) void lnlstm_fwd_kernel(const float* __restrict__ x, int dx,
                                                         const float* __restrict__ h, const float* __restrict__ c,
                                                         const float* __restrict__ K, const float* __restrict__ ln,
                                                         float* __restrict__ h_out, float* __restrict__ c_out,
                                                         int rows, int tiles_total) {
    constexpr int NT4 = D / 4;   // output tiles of z (4D columns)
    constexpr int TPG = D / 16;  // tiles per gate
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int krows = dx + D;
    float* lds_k = lds;
    float* lds_ln = lds + (size_t)krows * 4 * D;
    int* ticket = reinterpret_cast<int*>(lds_ln + 10 * D);

    const int tid = threadIdx.x;
    stage_weights_b128(lds_k, K, krows, 4 * D, tid, blockDim.x);
    for (int i = tid; i < 10 * D; i += blockDim.x) lds_ln[i] = ln[i];
    const int t_beg = (int)((long long)tiles_total * blockIdx.x / gridDim.x);
    const int t_end = (int)((long long)tiles_total * (blockIdx.x + 1) / gridDim.x);
    if (tid == 0) *ticket = t_beg;
    __syncthreads();

    const int lane = tid & 63, rl = lane & 15, g = lane >> 4;
    const int QX = dx >> 4, QT = QX + TPG;
    for (;;) {
        int tile = 0;
        if (lane == 0) tile = atomicAdd(ticket, 1);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= t_end) break;
        const int row = tile * 16 + rl;
        const bool valid = row < rows;
        const size_t rc = (size_t)(valid ? row : rows - 1);
        const float* xrow = x + rc * dx + g * 4;
        const float* hrow = h + rc * D + g * 4;
        f32x4 cf[TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) cf[t] = ld4(c + rc * D + g * 4 + t * 16);

        f32x4 acc[NT4];
#pragma unroll
        for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 cur = ld4(QX > 0 ? xrow : hrow);
        for (int q = 0; q < QT; ++q) {
            const int qn = q + 1;
            f32x4 nxt = cur;
            if (qn < QT) nxt = ld4(qn < QX ? xrow + qn * 16 : hrow + (qn - QX) * 16);
#pragma unroll
            for (int p = 0; p < 4; ++p) kstep<NT4>(acc, lds_k + frag_off<NT4>(q * 4 + p, g, rl), cur[p]);
            cur = nxt;
        }
        // split(z) = i, j, f, o (that order); LN each; c' = LN(c*sig(f+1) + sig(i)*relu(j)); h' = relu(c')*sig(o)
        f32x4 gi[TPG], gj[TPG], gf[TPG], go[TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            gi[t] = acc[t];
            gj[t] = acc[TPG + t];
            gf[t] = acc[2 * TPG + t];
            go[t] = acc[3 * TPG + t];
        }
        ln_gate<TPG>(gi, lds_ln + 0 * D, lds_ln + 1 * D, g, D);
        ln_gate<TPG>(gj, lds_ln + 2 * D, lds_ln + 3 * D, g, D);
        ln_gate<TPG>(gf, lds_ln + 4 * D, lds_ln + 5 * D, g, D);
        ln_gate<TPG>(go, lds_ln + 6 * D, lds_ln + 7 * D, g, D);
        f32x4 nc[TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                nc[t][r] = cf[t][r] * sigmoidf_(gf[t][r] + 1.0f) + sigmoidf_(gi[t][r]) * fmaxf(gj[t][r], 0.f);
        }
        ln_gate<TPG>(nc, lds_ln + 8 * D, lds_ln + 9 * D, g, D);
        if (valid) {
            float* hd = h_out + rc * D + g * 4;
            float* cd = c_out + rc * D + g * 4;
#pragma unroll
            for (int t = 0; t < TPG; ++t) {
                f32x4 hn;
#pragma unroll
                for (int r = 0; r < 4; ++r) hn[r] = fmaxf(nc[t][r], 0.f) * sigmoidf_(go[t][r]);
                st4(hd + t * 16, hn);
                st4(cd + t * 16, nc[t]);
            }
        }
    }
}

static int n_cus(void) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}

template <int D, int MAXL>
static int launch_mlp(const float* X, const float* wb, float* Y, float* acts, int rows, int n_layers,
                      unsigned relu_mask, hipStream_t st) {
    const int tiles = (rows + 15) / 16;
    // LDS per block decides residency: D=64 -> 65 KiB -> 2 blocks (16 waves) per CU.
    const int lds_bytes = MAXL * (D * D + D) * 4 + 16;
    const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
    int grid = n_cus() * per_cu;
    const int max_grid = (tiles + 7) / 8;  // at least one tile per wave
    if (grid > max_grid) grid = max_grid;
    mlp_fwd_kernel<D, MAXL><<<grid, 512, 0, st>>>(X, wb, Y, acts, rows, n_layers, relu_mask, tiles);
    return launched("tspgnn_mlp_fwd_f32");
}

template <int D>
static int launch_lnlstm(const float* x, int dx, const float* h, const float* c, const float* K, const float* ln,
                         float* h_out, float* c_out, int rows, hipStream_t st) {
    const size_t lds_bytes = ((size_t)(dx + D) * 4 * D + 10 * D + 4) * sizeof(float);
    if (lds_bytes > 160 * 1024)
        return fail(TSPGNN_EUNSUPPORTED, "lnlstm_fwd: K[%d,%d] does not fit LDS (%zu B)", dx + D, 4 * D, lds_bytes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_fwd_kernel<D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "lnlstm_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const int tiles = (rows + 15) / 16;
    const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
    int grid = n_cus() * per_cu;
    const int max_grid = (tiles + 7) / 8;
    if (grid > max_grid) grid = max_grid;
    lnlstm_fwd_kernel<D><<<grid, 512, lds_bytes, st>>>(x, dx, h, c, K, ln, h_out, c_out, rows, tiles);
    return launched("tspgnn_lnlstm_fwd_f32");
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_mlp_fwd_f32(const float* X, const float* wb, float* Y, float* acts, int rows, int d,
                                  int n_layers, unsigned relu_mask, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "mlp_fwd: rows=%d", rows);
    TSPGNN_REQUIRE(n_layers >= 1 && n_layers <= 4, "mlp_fwd: n_layers=%d must be in 1..4", n_layers);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "mlp_fwd: d=%d must be 32, 64 or 128", d);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(X && wb && Y, "mlp_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 32: return launch_mlp<32, 4>(X, wb, Y, acts, rows, n_layers, relu_mask, st);
        case 64: return launch_mlp<64, 4>(X, wb, Y, acts, rows, n_layers, relu_mask, st);
        default:
            if (n_layers > 2)
                return fail(TSPGNN_EUNSUPPORTED, "mlp_fwd: d=128 holds at most 2 layers in LDS (got %d)", n_layers);
            return launch_mlp<128, 2>(X, wb, Y, acts, rows, n_layers, relu_mask, st);
    }
}

extern "C" int tspgnn_lnlstm_fwd_f32(const float* x, int dx, const float* h, const float* c, const float* K,
                                     const float* ln, float* h_out, float* c_out, int rows, int d, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "lnlstm_fwd: rows=%d", rows);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "lnlstm_fwd: d=%d must be 32, 64 or 128", d);
    TSPGNN_REQUIRE(dx >= 0 && dx % 16 == 0, "lnlstm_fwd: dx=%d must be a non-negative multiple of 16", dx);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(h && c && K && ln && h_out && c_out && (dx == 0 || x), "lnlstm_fwd: null pointer");
    TSPGNN_REQUIRE(h_out != h && c_out != c, "lnlstm_fwd: outputs may not alias inputs");
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 32: return launch_lnlstm<32>(x, dx, h, c, K, ln, h_out, c_out, rows, st);
        case 64: return launch_lnlstm<64>(x, dx, h, c, K, ln, h_out, c_out, rows, st);
        default: return launch_lnlstm<128>(x, dx, h, c, K, ln, h_out, c_out, rows, st);
    }
}
