// Once-per-batch backward pieces around the loop and the optimiser (model.py:107-167):
// loss -> edge votes, the vote head's Dense(1), column sums, E_init_MLP weight gradients,
// L2 term + global-norm clip + Adam on the flat parameter buffer.
#include "common.h"
#include "mfma_tile.h"

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
// Persistent workgroups over chunks of 64 edges; one partial row of the NP parameter gradients per workgroup, summed
// over ALL of its chunks in registers (fixed chunk -> workgroup assignment, fixed summation order: deterministic).
// Round 4: the two products that are 95 % of the arithmetic -- the last layer's data gradient
// d3 = relu'(a3) . (dE0 W4^T) and weight gradient dW4 = a3^T dE0, and (from d = 64 on, where the layer before is wide
// enough for 16-wide tiles) d2 = relu'(a2) . (d3 W3^T) and dW3 = a2^T d3 -- run on the fp32 matrix instruction
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, the edges of a chunk are the contraction of the weight gradients);
// the scalar loops keep the narrow first layers and the bias sums.  d = 128, 636 800 edges: 8.4 ms -> see DESIGN.
// Per chunk:
//   A  (256 threads, four per edge): the forward chain of the edge -> a1, a2, a3 in LDS;
//   B  (wavefront w = edge tile w): d3 by MFMA over the D columns of dE0, masked by a3 > 0;
//   B' (d >= 64): d2 by MFMA over d3 (else one thread per edge), then d1 (four threads per edge);
//   C  dW4 (+ dW3) tiles accumulated by MFMA over the chunk's 64 edges, the other parameters by scalar sums.
template <int D>
__global__ __launch_bounds__(256) void einit_bwd_kernel(const float2* __restrict__ WC, const float* __restrict__ wb,
                                                        const float* __restrict__ dE0, float* __restrict__ P, int M,
                                                        int n_chunks) {
    constexpr int H1 = D / 8, H2 = D / 4, H3 = D / 2;
    constexpr int NP = 2 * H1 + H1 + H1 * H2 + H2 + H2 * H3 + H3 + H3 * D + D;
    constexpr int O_W3 = 2 * H1 + H1 + H1 * H2 + H2, O_B3 = O_W3 + H2 * H3, O_W4 = O_B3 + H3, O_B4 = O_W4 + H3 * D;
    constexpr int NPT = (NP + 255) / 256;
    constexpr int E = 64;
    constexpr bool M3 = (H2 % 16) == 0;                       // level 3 on the matrix instruction as well
    constexpr int T4 = (H3 / 16) * (D / 16);                  // 16x16 tiles of dW4
    constexpr int TPW4 = T4 >= 4 ? T4 / 4 : 1;                // ... per wavefront
    constexpr int T3 = M3 ? (H2 / 16) * (H3 / 16) : 0;
    constexpr int TPW3 = T3 >= 4 ? T3 / 4 : 1;
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
    __shared__ float s_w4[H3][D + 1];                         // (padded: the MFMA operand reads walk the rows)
    __shared__ float s_w3[H2][H3 + 1];
    __shared__ float s_w12[2 * H1 + H1 + H1 * H2 + H2 + H3];  // W1, b1, W2, b2 as in wb, then b3
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 15, lk = lane >> 4;
    for (int i = t; i < H3 * D; i += blockDim.x) s_w4[i / D][i % D] = W4[i];
    for (int i = t; i < H2 * H3; i += blockDim.x) s_w3[i / H3][i % H3] = W3[i];
    for (int i = t; i < 2 * H1 + H1 + H1 * H2 + H2; i += blockDim.x) s_w12[i] = wb[i];
    for (int i = t; i < H3; i += blockDim.x) s_w12[2 * H1 + H1 + H1 * H2 + H2 + i] = b3[i];
    const float* l_W1 = s_w12;
    const float* l_b1 = l_W1 + 2 * H1;
    const float* l_W2 = l_b1 + H1;
    const float* l_b2 = l_W2 + H1 * H2;
    const float* l_b3 = l_b2 + H2;
    float acc[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) acc[i] = 0.f;
    f32x4 accW4[TPW4], accW3[TPW3];
#pragma unroll
    for (int i = 0; i < TPW4; ++i) accW4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TPW3; ++i) accW3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the dE0 tile of the NEXT chunk is fetched into registers while the current one is worked on (one workgroup of four
    // wavefronts per CU at d = 128: nothing else would hide the HBM round trip)
    constexpr int NPRE = E * D / 256;
    float pre[NPRE];
    auto fetch = [&](int chunk) {
        const int e0n = chunk * E;
#pragma unroll
        for (int r = 0; r < NPRE; ++r) {
            const int i = t + r * 256, le = i / D, j = i % D;
            pre[r] = (chunk < n_chunks && e0n + le < M) ? dE0[(size_t)(e0n + le) * D + j] : 0.f;   // rows past M are zero
        }
    };
    fetch(blockIdx.x);
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int e0 = chunk * E;
        __syncthreads();   // (the previous chunk's phase C has read the tiles; first pass: the staged weights are complete)
#pragma unroll
        for (int r = 0; r < NPRE; ++r) {
            const int i = t + r * 256;
            s_d4[i / D][i % D] = pre[r];
        }
        fetch(chunk + (int)gridDim.x);
        {   // A: a1 and a quarter of a2 per thread (four threads per edge; the small weights come from LDS)
            const int le = t >> 2, q = t & 3;
            const int e = e0 + le;
            const float2 wc = e < M ? WC[e] : make_float2(0.f, 0.f);
            float a1[H1];
#pragma unroll
            for (int j = 0; j < H1; ++j) a1[j] = fmaxf(fmaf(wc.y, l_W1[H1 + j], fmaf(wc.x, l_W1[j], 0.f)) + l_b1[j], 0.f);
#pragma unroll
            for (int jj = 0; jj < H2 / 4; ++jj) {
                const int j = q * (H2 / 4) + jj;
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < H1; ++k) sum = fmaf(a1[k], l_W2[k * H2 + j], sum);
                s_a2[le][j] = fmaxf(sum + l_b2[j], 0.f);
            }
            if (q == 0) {
                s_in[le][0] = wc.x, s_in[le][1] = wc.y;
#pragma unroll
                for (int k = 0; k < H1; ++k) s_a1[le][k] = a1[k];
            }
        }
        __syncthreads();
        {   // A': a3 = relu(a2 W3 + b3) by MFMA, wavefront w = edge tile w
            f32x4 c[H3 / 16];
#pragma unroll
            for (int nt = 0; nt < H3 / 16; ++nt) {
                const float b = l_b3[16 * nt + lm];
                c[nt] = f32x4{b, b, b, b};
            }
#pragma unroll
            for (int ks = 0; ks < H2 / 4; ++ks) {
                const float a = s_a2[16 * wave + lm][4 * ks + lk];
#pragma unroll
                for (int nt = 0; nt < H3 / 16; ++nt) c[nt] = MFMA16(a, s_w3[4 * ks + lk][16 * nt + lm], c[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < H3 / 16; ++nt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s_a3[16 * wave + 4 * lk + i][16 * nt + lm] = fmaxf(c[nt][i], 0.f);
            }
        }
        __syncthreads();
        {   // B: d3[e][k3] = (a3 > 0) * sum_j dE0[e][j] W4[k3][j]   (rows past M: dE0 is zero, so is d3)
            f32x4 c[H3 / 16];
#pragma unroll
            for (int nt = 0; nt < H3 / 16; ++nt) c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int js = 0; js < D / 4; ++js) {
                const float a = s_d4[16 * wave + lm][4 * js + lk];
#pragma unroll
                for (int nt = 0; nt < H3 / 16; ++nt) c[nt] = MFMA16(a, s_w4[16 * nt + lm][4 * js + lk], c[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < H3 / 16; ++nt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int le = 16 * wave + 4 * lk + i, k3 = 16 * nt + lm;
                    s_d3[le][k3] = s_a3[le][k3] > 0.f ? c[nt][i] : 0.f;
                }
            }
        }
        __syncthreads();
        if constexpr (M3) {   // B': d2[e][k2] = (a2 > 0) * sum_j d3[e][j] W3[k2][j]
            f32x4 c[H2 / 16];
#pragma unroll
            for (int nt = 0; nt < H2 / 16; ++nt) c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int js = 0; js < H3 / 4; ++js) {
                const float a = s_d3[16 * wave + lm][4 * js + lk];
#pragma unroll
                for (int nt = 0; nt < H2 / 16; ++nt) c[nt] = MFMA16(a, s_w3[16 * nt + lm][4 * js + lk], c[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < H2 / 16; ++nt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int le = 16 * wave + 4 * lk + i, k2 = 16 * nt + lm;
                    s_d2[le][k2] = s_a2[le][k2] > 0.f ? c[nt][i] : 0.f;
                }
            }
        } else if (t < E) {
#pragma unroll
            for (int k = 0; k < H2; ++k) {
                float sum = 0.f;
                _Pragma("unroll 8") for (int j = 0; j < H3; ++j) sum = fmaf(s_w3[k][j], s_d3[t][j], sum);
                s_d2[t][k] = s_a2[t][k] > 0.f ? sum : 0.f;
            }
        }
        __syncthreads();
        {   // d1, four threads per edge
            const int le = t >> 2, q = t & 3;
            for (int kk = 0; kk < (H1 + 3) / 4; ++kk) {
                const int k = q * ((H1 + 3) / 4) + kk;
                if (k < H1) {
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < H2; ++j) sum = fmaf(l_W2[k * H2 + j], s_d2[le][j], sum);
                    s_d1[le][k] = s_a1[le][k] > 0.f ? sum : 0.f;
                }
            }
        }
        __syncthreads();
        // C: weight-gradient tiles over the chunk's 64 edges: dW[m][n] += sum_e left[e][m] right[e][n]
        {
            const int t0 = wave * TPW4;
            if (t0 < T4) {
                const int mt = t0 / (D / 16);           // (a wavefront's tiles share their row of tiles)
#pragma unroll 4
                for (int es = 0; es < E / 4; ++es) {
                    const float a = s_a3[4 * es + lk][16 * mt + lm];
#pragma unroll
                    for (int i = 0; i < TPW4; ++i)
                        accW4[i] = MFMA16(a, s_d4[4 * es + lk][16 * ((t0 + i) % (D / 16)) + lm], accW4[i]);
                }
            }
        }
        if constexpr (M3) {
            const int t0 = wave * TPW3;
            if (t0 < T3) {
                const int mt = t0 / (H3 / 16);
#pragma unroll 4
                for (int es = 0; es < E / 4; ++es) {
                    const float a = s_a2[4 * es + lk][16 * mt + lm];
#pragma unroll
                    for (int i = 0; i < TPW3; ++i)
                        accW3[i] = MFMA16(a, s_d3[4 * es + lk][16 * ((t0 + i) % (H3 / 16)) + lm], accW3[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {   // the other parameters: scalar sums over the chunk
            const int p = t + i * 256;
            if (p >= NP) break;
            if ((p >= O_W4 && p < O_B4) || (M3 && p >= O_W3 && p < O_B3)) continue;   // (on the matrix instruction above)
            int q = p;
            float sum = 0.f;
            if (q < 2 * H1) {  // W1[k][j]
                const int k = q / H1, j = q % H1;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) sum = fmaf(s_in[e][k], s_d1[e][j], sum);
            } else if ((q -= 2 * H1) < H1) {
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) sum += s_d1[e][q];
            } else if ((q -= H1) < H1 * H2) {
                const int k = q / H2, j = q % H2;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) sum = fmaf(s_a1[e][k], s_d2[e][j], sum);
            } else if ((q -= H1 * H2) < H2) {
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) sum += s_d2[e][q];
            } else if ((q -= H2) < H2 * H3) {
                const int k = q / H3, j = q % H3;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) sum = fmaf(s_a2[e][k], s_d3[e][j], sum);
            } else if ((q -= H2 * H3) < H3) {
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) sum += s_d3[e][q];
            } else {
                q -= H3 + H3 * D;
                _Pragma("unroll 4") for (int e = 0; e < E; ++e) sum += s_d4[e][q];
            }
            acc[i] += sum;
        }
    }
    float* Pb = P + (size_t)blockIdx.x * NP;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int p = t + i * 256;
        if (p < NP && !((p >= O_W4 && p < O_B4) || (M3 && p >= O_W3 && p < O_B3))) Pb[p] = acc[i];
    }
    {   // the MFMA tiles: lane holds rows 4 lk + i, column lm of its 16x16 tile
        const int t0 = wave * TPW4;
        if (t0 < T4) {
#pragma unroll
            for (int i = 0; i < TPW4; ++i) {
                const int mt = (t0 + i) / (D / 16), nt = (t0 + i) % (D / 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) Pb[O_W4 + (16 * mt + 4 * lk + r) * D + 16 * nt + lm] = accW4[i][r];
            }
        }
    }
    if constexpr (M3) {
        const int t0 = wave * TPW3;
        if (t0 < T3) {
#pragma unroll
            for (int i = 0; i < TPW3; ++i) {
                const int mt = (t0 + i) / (H3 / 16), nt = (t0 + i) % (H3 / 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) Pb[O_W3 + (16 * mt + 4 * lk + r) * H3 + 16 * nt + lm] = accW3[i][r];
            }
        }
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
// bucket = [ grad (n floats) | tail: B_r, B_r*loss_r, B_r*acc_r, TP_r, FP_r, TN_r, FN_r, guard bit 0, bit 1, bit 2 ] (the
// bits of the rank's guard word as 0 / 1 floats, one slot each: their SUMS over the ranks keep each bit's meaning --
// bit 0 an operand beyond fp16's range, bit 1 a gate row below the split's floor, bit 2 "this rank's variables were
// assigned since the replicas were last made identical": Session.train_step).  One launch each side
// of the one all-reduce of a training step: pack weights the rank's gradient and statistics by its batch
// size, unpack divides by the reduced batch size -- on the device, so the step's two HIP graphs run back to back around
// the collective -- and hands the (summed) range flag back to the guard word every rank's optimiser launch looks at.
__global__ __launch_bounds__(256) void bucket_pack_kernel(float* __restrict__ bucket, int n, int with_grad, float nb,
                                                          const float* __restrict__ stats,
                                                          const unsigned* __restrict__ flag) {
    if (with_grad)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bucket[i] *= nb;
    if (blockIdx.x == 0 && threadIdx.x < 10) {
        const int k = threadIdx.x;
        float v = 0.f;
        if (k == 0) v = nb;
        else if (k < 3) v = stats ? nb * stats[k - 1] : 0.f;
        else if (k < 7) v = stats ? stats[k - 1] : 0.f;
        else v = flag ? (float)((flag[0] >> (k - 7)) & 1u) : 0.f;
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
        if (k == 7 && flag)
            flag[0] = (bucket[n + 7] != 0.f ? 1u : 0u) | (bucket[n + 8] != 0.f ? 2u : 0u) | (bucket[n + 9] != 0.f ? 4u : 0u);
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
