// Once-per-batch backward pieces around the loop and the optimiser (model.py:107-167):
// loss -> edge votes, the vote head's Dense(1), column sums, E_init_MLP weight gradients,
// L2 term + global-norm clip + Adam on the flat parameter buffer.
#include "common.h"

namespace tspgnn {

// d loss / d vote[e] for e in problem p:  (sigmoid(logit_p) - label_p) / (B * n_edges_p)
// (mean over problems of sigmoid cross entropy, model.py:157; per-problem mean of votes, model.py:134-145).
__global__ __launch_bounds__(256) void vote_grad_kernel(const float* __restrict__ logits,
                                                        const float* __restrict__ labels, const int* __restrict__ seg,
                                                        float* __restrict__ dvote, int B) {
    const int p = blockIdx.x;
    if (p >= B) return;
    const int beg = seg[p], end = seg[p + 1];
    const float gval = (sigmoidf_(logits[p]) - labels[p]) / ((float)B * (float)(end - beg));
    for (int k = beg + (int)threadIdx.x; k < end; k += blockDim.x) dvote[k] = gval;
}

// dX[r,:] = dy[r] * w      (backward of y = X w + b through X)
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dy, const float4* __restrict__ w,
                                                         float4* __restrict__ dX, long long total4, int d4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const long long r = i / d4;
        const int c = (int)(i - r * d4);
        const float s = dy[r];
        const float4 wv = w[c];
        dX[i] = make_float4(s * wv.x, s * wv.y, s * wv.z, s * wv.w);
    }
}

// Stage 1 of  out[f] = sum_r wt[r] * X[r,f]  (wt == NULL: plain column sum) and  sum_r wt[r]:
// each workgroup owns a contiguous row chunk; thread (rsub, c) walks rows rsub, rsub+RS, ... of the
// chunk for float4 column c; fixed-order LDS tree over rsub.  Partials: P[chunk][d] (+ Pw[chunk]).
__global__ __launch_bounds__(256) void wcolsum_kernel(const float4* __restrict__ X, const float* __restrict__ wt,
                                                      float4* __restrict__ P, float* __restrict__ Pw, long long rows,
                                                      int d4, long long chunk_rows) {
    __shared__ float4 red[256];
    __shared__ float redw[256];
    const int RS = 256 / d4;  // rows in flight per workgroup
    const int c = threadIdx.x % d4, rsub = threadIdx.x / d4;
    const long long r_beg = (long long)blockIdx.x * chunk_rows, r_end = min(rows, r_beg + chunk_rows);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accw = 0.f;
    if (rsub < RS) {
        for (long long r = r_beg + rsub; r < r_end; r += RS) {
            const float wv = wt ? wt[r] : 1.0f;
            const float4 x = X[r * d4 + c];
            acc.x = fmaf(wv, x.x, acc.x);
            acc.y = fmaf(wv, x.y, acc.y);
            acc.z = fmaf(wv, x.z, acc.z);
            acc.w = fmaf(wv, x.w, acc.w);
            if (c == 0) accw += wv;
        }
    }
    red[threadIdx.x] = acc;
    redw[threadIdx.x] = accw;
    __syncthreads();
    if (rsub == 0) {
        for (int s = 1; s < RS; ++s) {
            const float4 o = red[s * d4 + c];
            acc.x += o.x, acc.y += o.y, acc.z += o.z, acc.w += o.w;
            if (c == 0) accw += redw[s * d4];
        }
        P[(size_t)blockIdx.x * d4 + c] = acc;
        if (c == 0 && Pw) Pw[blockIdx.x] = accw;
    }
}

// ------------------------------------------------------------- E_init_MLP weight gradients
// Persistent workgroups over chunks of 64 edges; every thread owns a strided set of the NP parameters (2 824 at d=64,
// 11 088 at d=128) and keeps their sums in registers across ALL of the workgroup's chunks -- one partial row per
// workgroup (512 rows) instead of one per chunk (9 950 rows = 441 MB at the C5 shard's 636 800 edges).  Per chunk:
//   1a (all 256 threads, four per edge): recompute the forward chain of the edge, then a quarter of
//      d3 = relu'(a3) . (dE0 W4^T) -- the 64x128x64 product that dominated the one-thread-per-edge version -- with W4
//      staged in LDS once per workgroup (the same element for every edge: broadcast reads);
//   1b (one thread per edge): d2, d1 through the two small layers;
//   2  (all threads): outer-product sums of the chunk into the thread's parameters.
// Fixed chunk -> workgroup assignment and fixed summation order: deterministic.
template <int D>
__global__ __launch_bounds__(256) void einit_bwd_kernel(const float2* __restrict__ WC, const float* __restrict__ wb,
                                                        const float* __restrict__ dE0, float* __restrict__ P, int M,
                                                        int n_chunks) {
    constexpr int H1 = D / 8, H2 = D / 4, H3 = D / 2;
    constexpr int NP = 2 * H1 + H1 + H1 * H2 + H2 + H2 * H3 + H3 + H3 * D + D;
    constexpr int NPT = (NP + 255) / 256;
    constexpr int E = 64, KQ = H3 / 4;
    const float* W1 = wb;
    const float* b1 = W1 + 2 * H1;
    const float* W2 = b1 + H1;
    const float* b2 = W2 + H1 * H2;
    const float* W3 = b2 + H2;
    const float* b3 = W3 + H2 * H3;
    const float* W4 = b3 + H3;
    __shared__ float s_in[E][2 + 1];
    __shared__ float s_a1[E][H1 + 1], s_a2[E][H2 + 1], s_a3[E][H3 + 1];
    __shared__ float s_d1[E][H1 + 1], s_d2[E][H2 + 1], s_d3[E][H3 + 1], s_d4[E][D + 1];
    __shared__ __attribute__((aligned(16))) float s_w4[H3 * D];
    const int t = threadIdx.x;
    for (int i = t; i < H3 * D / 4; i += blockDim.x) reinterpret_cast<float4*>(s_w4)[i] = reinterpret_cast<const float4*>(W4)[i];
    float acc[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) acc[i] = 0.f;
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int e0 = chunk * E;
        __syncthreads();   // (the previous chunk's phase 2 has read the tiles; first pass: s_w4 is complete)
        for (int i = t; i < E * D; i += blockDim.x) {   // dE0 tile -> LDS (coalesced)
            const int le = i / D, j = i % D;
            s_d4[le][j] = (e0 + le < M) ? dE0[(size_t)(e0 + le) * D + j] : 0.f;
        }
        __syncthreads();
        {   // 1a
            const int le = t >> 2, q = t & 3;
            const int e = e0 + le;
            const bool ok = e < M;
            const float2 wc = ok ? WC[e] : make_float2(0.f, 0.f);
            float a1[H1], a2[H2];
#pragma unroll
            for (int j = 0; j < H1; ++j) a1[j] = fmaxf(fmaf(wc.y, W1[H1 + j], fmaf(wc.x, W1[j], 0.f)) + b1[j], 0.f);
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < H1; ++k) s = fmaf(a1[k], W2[k * H2 + j], s);
                a2[j] = fmaxf(s + b2[j], 0.f);
            }
            // this thread's quarter of the third layer: a3[k] and d3[k] = relu'(a3[k]) * (W4[k,:] . dE0[e,:])
            for (int kk = 0; kk < KQ; ++kk) {
                const int k = q * KQ + kk;
                float a3k = b3[k];
#pragma unroll
                for (int j = 0; j < H2; ++j) a3k = fmaf(a2[j], W3[j * H3 + k], a3k);
                a3k = fmaxf(a3k, 0.f);
                float s = 0.f;
                _Pragma("unroll 8") for (int j = 0; j < D; ++j) s = fmaf(s_w4[k * D + j], s_d4[le][j], s);
                s_a3[le][k] = a3k;
                s_d3[le][k] = (ok && a3k > 0.f) ? s : 0.f;
            }
            if (q == 0) {
                s_in[le][0] = wc.x, s_in[le][1] = wc.y;
#pragma unroll
                for (int k = 0; k < H1; ++k) s_a1[le][k] = a1[k];
#pragma unroll
                for (int k = 0; k < H2; ++k) s_a2[le][k] = a2[k];
            }
        }
        __syncthreads();
        if (t < E) {   // 1b
            const bool ok = e0 + t < M;
            float d2[H2];
#pragma unroll
            for (int k = 0; k < H2; ++k) {
                float s = 0.f;
                _Pragma("unroll 8") for (int j = 0; j < H3; ++j) s = fmaf(W3[k * H3 + j], s_d3[t][j], s);
                d2[k] = (ok && s_a2[t][k] > 0.f) ? s : 0.f;
                s_d2[t][k] = d2[k];
            }
#pragma unroll
            for (int k = 0; k < H1; ++k) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < H2; ++j) s = fmaf(W2[k * H2 + j], d2[j], s);
                s_d1[t][k] = (ok && s_a1[t][k] > 0.f) ? s : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPT; ++i) {   // 2
            const int p = t + i * 256;
            if (p >= NP) break;
            int q = p;
            float s = 0.f;
            if (q < 2 * H1) {  // W1[k][j]
                const int k = q / H1, j = q % H1;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s = fmaf(s_in[e][k], s_d1[e][j], s);
            } else if ((q -= 2 * H1) < H1) {
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s += s_d1[e][q];
            } else if ((q -= H1) < H1 * H2) {
                const int k = q / H2, j = q % H2;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s = fmaf(s_a1[e][k], s_d2[e][j], s);
            } else if ((q -= H1 * H2) < H2) {
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s += s_d2[e][q];
            } else if ((q -= H2) < H2 * H3) {
                const int k = q / H3, j = q % H3;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s = fmaf(s_a2[e][k], s_d3[e][j], s);
            } else if ((q -= H2 * H3) < H3) {
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s += s_d3[e][q];
            } else if ((q -= H3) < H3 * D) {
                const int k = q / D, j = q % D;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s = fmaf(s_a3[e][k], s_d4[e][j], s);
            } else {
                q -= H3 * D;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) s += s_d4[e][q];
            }
            acc[i] += s;
        }
    }
    float* Pb = P + (size_t)blockIdx.x * NP;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int p = t + i * 256;
        if (p < NP) Pb[p] = acc[i];
    }
}

// ------------------------------------------------------------- optimiser (model.py:160-167)
// g <- g + l2 * theta ; partial sums of g^2 (fixed order per workgroup).
__global__ __launch_bounds__(256) void l2_sumsq_kernel(float* __restrict__ g, const float* __restrict__ theta,
                                                       float l2, float* __restrict__ partial, int n,
                                                       int* __restrict__ step_counter,
                                                       const unsigned* __restrict__ skip_flag) {
    __shared__ float red[256];
    // this optimiser step's index t (not advanced for a step that is skipped, see adam_clip_kernel)
    if (step_counter && blockIdx.x == 0 && threadIdx.x == 0 && !(skip_flag && skip_flag[0])) step_counter[0] += 1;
    float s = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v = fmaf(l2, theta[i], g[i]);
        g[i] = v;
        s = fmaf(v, v, s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// gnorm = sqrt(sum partial); scale = clip / max(gnorm, clip)  (tf.clip_by_global_norm); then Adam with
// the bias-corrected step size lr_t (tf.train.AdamOptimizer._apply_dense).  state: [0]=gnorm (output).
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ partial, int n_partial, float clip,
                                                        float lr_t, float b1, float b2, float eps,
                                                        float* __restrict__ state, int n,
                                                        const int* __restrict__ step_counter,
                                                        const unsigned* __restrict__ skip_flag) {
    const bool skip = skip_flag && skip_flag[0];   // the step's forward left the f16x2 range: leave the variables alone
    if (step_counter) {  // lr_t holds the base rate: apply Adam's bias correction for step t on the device
        const float t = (float)step_counter[0];
        lr_t = lr_t * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
    }
    float ss = 0.f;
    for (int k = 0; k < n_partial; ++k) ss += partial[k];  // same fixed order in every thread
    const float gnorm = sqrtf(ss);
    const float scale = clip > 0.f ? clip / fmaxf(gnorm, clip) : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) state[0] = gnorm;
    if (skip) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float gi = g[i] * scale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

static void colsum_plan(long long rows, int* n_chunks, long long* chunk_rows) {
    long long nc = (rows + 1023) / 1024;
    const long long cap = (long long)n_cus() * 4;
    if (nc > cap) nc = cap;
    if (nc < 1) nc = 1;
    long long cr = (rows + nc - 1) / nc;
    nc = (rows + cr - 1) / cr;
    *n_chunks = (int)(nc < 1 ? 1 : nc);
    *chunk_rows = cr < 1 ? 1 : cr;
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_vote_grad_f32(const float* logits, const float* labels, const int32_t* seg, float* dvote, int B,
                                    void* stream) {
    TSPGNN_REQUIRE(B >= 0, "vote_grad: B=%d", B);
    if (B == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(logits && labels && seg && dvote, "vote_grad: null pointer");
    vote_grad_kernel<<<(unsigned)B, 256, 0, as_stream(stream)>>>(logits, labels, seg, dvote, B);
    return launched("tspgnn_vote_grad_f32");
}

extern "C" int tspgnn_rowdot_bwd_f32(const float* dy, const float* w, float* dX, int rows, int d, void* stream) {
    TSPGNN_REQUIRE(rows >= 0 && d > 0 && d % 4 == 0, "rowdot_bwd: rows=%d d=%d", rows, d);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(dy && w && dX, "rowdot_bwd: null pointer");
    const long long total4 = (long long)rows * (d / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    rowdot_bwd_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(
        dy, reinterpret_cast<const float4*>(w), reinterpret_cast<float4*>(dX), total4, d / 4);
    return launched("tspgnn_rowdot_bwd_f32");
}

extern "C" long long tspgnn_wcolsum_workspace_floats(long long rows, int d) {
    int nc;
    long long cr;
    colsum_plan(rows < 1 ? 1 : rows, &nc, &cr);
    return (long long)nc * (d + 1);
}

extern "C" int tspgnn_wcolsum_f32(const float* X, const float* wt, long long rows, int d, float scale, float* out,
                                  float* out_wsum, float* workspace, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "wcolsum: rows=%lld", rows);
    TSPGNN_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024 && 256 % (d / 4) == 0, "wcolsum: d=%d must divide 1024 and be a multiple of 4", d);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(X && out && workspace, "wcolsum: null pointer");
    int nc;
    long long cr;
    colsum_plan(rows, &nc, &cr);
    hipStream_t st = as_stream(stream);
    float* Pw = workspace + (size_t)nc * d;
    wcolsum_kernel<<<(unsigned)nc, 256, 0, st>>>(reinterpret_cast<const float4*>(X), wt,
                                                 reinterpret_cast<float4*>(workspace), out_wsum ? Pw : nullptr, rows,
                                                 d / 4, cr);
    int rc = launched("tspgnn_wcolsum_f32");
    if (rc) return rc;
    reduce_partials2(workspace, nc, d, out, d, Pw, 1, out_wsum, out_wsum ? 1 : 0, scale, 1, st);
    return launched("tspgnn_wcolsum_f32(reduce)");
}

static int einit_np(int d) {
    const int h1 = d / 8, h2 = d / 4, h3 = d / 2;
    return 2 * h1 + h1 + h1 * h2 + h2 + h2 * h3 + h3 + h3 * d + d;
}

extern "C" long long tspgnn_einit_bwd_workspace_floats(int M, int d) {
    return (long long)((M + 63) / 64) * einit_np(d);
}

extern "C" int tspgnn_einit_bwd_f32(const float* WC, const float* wb, const float* dE0, float* dwb, float* workspace,
                                    int M, int d, void* stream) {
    TSPGNN_REQUIRE(M >= 0, "einit_bwd: M=%d", M);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "einit_bwd: d=%d must be 32, 64 or 128", d);
    if (M == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(WC && wb && dE0 && dwb && workspace, "einit_bwd: null pointer");
    const int n_chunks = (M + 63) / 64;
    unsigned grid = (unsigned)n_cus() * (d >= 128 ? 1u : 2u);   // persistent workgroups (LDS: 1 / 2 per CU)
    if (grid > (unsigned)n_chunks) grid = (unsigned)n_chunks;
    hipStream_t st = as_stream(stream);
    const float2* WC2 = reinterpret_cast<const float2*>(WC);
    switch (d) {
        case 32: einit_bwd_kernel<32><<<grid, 256, 0, st>>>(WC2, wb, dE0, workspace, M, n_chunks); break;
        case 64: einit_bwd_kernel<64><<<grid, 256, 0, st>>>(WC2, wb, dE0, workspace, M, n_chunks); break;
        default: einit_bwd_kernel<128><<<grid, 256, 0, st>>>(WC2, wb, dE0, workspace, M, n_chunks); break;
    }
    int rc = launched("tspgnn_einit_bwd_f32");
    if (rc) return rc;
    const int np = einit_np(d);
    reduce_partials(workspace, (int)grid, np, dwb, np, 1.0f, 1, st);
    return launched("tspgnn_einit_bwd_f32(reduce)");
}

#define TSPGNN_OPT_PARTIALS 256

extern "C" long long tspgnn_adam_workspace_floats(void) { return TSPGNN_OPT_PARTIALS; }

extern "C" int tspgnn_adam_clip_step_f32(float* theta, float* g, float* m, float* v, int n, float l2_scale,
                                         float clip_norm, float lr_t, float beta1, float beta2, float eps,
                                         float* gnorm_out, float* workspace, int* step_counter,
                                         const unsigned* skip_flag, void* stream) {
    TSPGNN_REQUIRE(n >= 0, "adam_clip_step: n=%d", n);
    if (n == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(theta && g && m && v && gnorm_out && workspace, "adam_clip_step: null pointer");
    hipStream_t st = as_stream(stream);
    int blocks = (n + 255) / 256;
    if (blocks > TSPGNN_OPT_PARTIALS) blocks = TSPGNN_OPT_PARTIALS;
    l2_sumsq_kernel<<<blocks, 256, 0, st>>>(g, theta, l2_scale, workspace, n, step_counter, skip_flag);
    int rc = launched("tspgnn_adam_clip_step_f32(l2+norm)");
    if (rc) return rc;
    adam_clip_kernel<<<blocks, 256, 0, st>>>(theta, g, m, v, workspace, blocks, clip_norm, lr_t, beta1, beta2, eps,
                                             gnorm_out, n, step_counter, skip_flag);
    return launched("tspgnn_adam_clip_step_f32");
}

// ---------------------------------------------------------------- data-parallel bucket (Session.allreduce_grads)
// bucket = [ grad (n floats) | tail: B_r, B_r*loss_r, B_r*acc_r, TP_r, FP_r, TN_r, FN_r, range flag_r ].  One launch each side
// of the one all-reduce of a training step: pack weights the rank's gradient and statistics by its batch
// size, unpack divides by the reduced batch size -- on the device, so the step's two HIP graphs run back to back around
// the collective -- and hands the (summed) range flag back to the guard word every rank's optimiser launch looks at.
__global__ __launch_bounds__(256) void bucket_pack_kernel(float* __restrict__ bucket, int n, int with_grad, float nb,
                                                          const float* __restrict__ stats,
                                                          const unsigned* __restrict__ flag) {
    if (with_grad)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bucket[i] *= nb;
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        const int k = threadIdx.x;
        float v = 0.f;
        if (k == 0) v = nb;
        else if (k < 3) v = stats ? nb * stats[k - 1] : 0.f;
        else if (k < 7) v = stats ? stats[k - 1] : 0.f;
        else v = flag ? (float)flag[0] : 0.f;
        bucket[n + k] = v;
    }
}

__global__ __launch_bounds__(256) void bucket_unpack_kernel(float* __restrict__ bucket, int n, int with_grad,
                                                            float* __restrict__ stats, unsigned* __restrict__ flag) {
    const float inv = 1.0f / bucket[n];
    if (with_grad)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bucket[i] *= inv;
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        const int k = threadIdx.x;
        if (k >= 1 && k < 3 && stats) stats[k - 1] = bucket[n + k] * inv;
        if (k >= 3 && k < 7 && stats) stats[k - 1] = bucket[n + k];
        if (k == 7 && flag) flag[0] = bucket[n + 7] != 0.f ? 1u : 0u;
    }
}

extern "C" int tspgnn_bucket_pack_f32(float* bucket, int n, int with_grad, float local_batch, const float* stats,
                                      const unsigned* range_flag, void* stream) {
    TSPGNN_REQUIRE(n >= 0 && bucket, "bucket_pack: n=%d, bucket=%p", n, (void*)bucket);
    int blocks = with_grad ? (n + 255) / 256 : 1;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    bucket_pack_kernel<<<blocks, 256, 0, as_stream(stream)>>>(bucket, n, with_grad, local_batch, stats, range_flag);
    return launched("tspgnn_bucket_pack_f32");
}

extern "C" int tspgnn_bucket_unpack_f32(float* bucket, int n, int with_grad, float* stats, unsigned* range_flag,
                                        void* stream) {
    TSPGNN_REQUIRE(n >= 0 && bucket, "bucket_unpack: n=%d, bucket=%p", n, (void*)bucket);
    int blocks = with_grad ? (n + 255) / 256 : 1;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    bucket_unpack_kernel<<<blocks, 256, 0, as_stream(stream)>>>(bucket, n, with_grad, stats, range_flag);
    return launched("tspgnn_bucket_unpack_f32");
}
