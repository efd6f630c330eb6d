// Elementwise part of the LN-LSTM backward pass on a 16-row tile held in the MFMA D layout, shared by the fp32-MFMA
// kernel (dense_bwd.hip) and the f16x2 kernel (dense_bwd_h2.hip).
#pragma once
#include "common.h"
#include "mfma_tile.h"

namespace tspgnn {

// LayerNorm forward statistics of one gate held in the D-layout (TPG tiles x 4 regs per lane, the rest
// of the row in the lanes l^16, l^32, l^48): v <- xhat = (v - mean) * rstd, returns rstd.
// SWAP: lane-group sums by v_permlane swaps (full EXEC mask required) instead of ds_bpermute.
template <int TPG, bool SWAP = false>
__device__ __forceinline__ float ln_normalize(f32x4 (&v)[TPG], int D, float eps = 1e-12f) {
    auto lane_sum = [](float x) { return SWAP ? sum_over_lane_groups16_swap(x) : sum_over_lane_groups16(x); };
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TPG; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    s = lane_sum(s);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[t][r] -= mean;
            q = fmaf(v[t][r], v[t][r], q);
        }
    }
    q = lane_sum(q);
    const float rstd = __builtin_amdgcn_rsqf(q / (float)D + eps);
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] *= rstd;
    }
    return rstd;
}

// LayerNorm backward of one gate: dn = gradient w.r.t. the LN output, xhat = normalised input.
// Overwrites dn with the gradient w.r.t. the LN input and adds this tile's contribution to the
// wavefront's (dgamma, dbeta) slab: the sums over the tile's 16 rows are reduce-scatters over the DPP row (sixteen
// per-lane values at a time, row16_reduce_scatter), after which lane rl holds the total of ONE feature and every lane
// issues one ds_add.
template <int TPG, bool SWAP = false>
__device__ __forceinline__ void ln_backward(f32x4 (&dn)[TPG], const f32x4 (&xhat)[TPG], float rstd,
                                            const float* gamma, float* slab_dgamma, float* slab_dbeta, int g, int rl,
                                            bool valid, int D) {
    constexpr int NV = 4 * TPG;               // values per lane and quantity (8, 16 or 32)
    static_assert(NV == 8 || NV % 16 == 0, "ln_backward: D must be 32 or a multiple of 64");
    if constexpr (NV == 8) {                  // one pass: [d xhat | d] of the lane's eight values
        float v[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d_ = valid ? dn[i >> 2][i & 3] : 0.f;
            v[i] = d_ * xhat[i >> 2][i & 3];
            v[8 + i] = d_;
        }
        const float tot = row16_reduce_scatter(v, rl);
        const int i = rl & 7;
        atomicAdd((rl & 8 ? slab_dbeta : slab_dgamma) + (i >> 2) * 16 + g * 4 + (i & 3), tot);   // wavefront-private slab
    } else {
        const int f = (rl >> 2) * 16 + g * 4 + (rl & 3);
#pragma unroll
        for (int c = 0; c < NV / 16; ++c) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = valid ? dn[4 * c + (i >> 2)][i & 3] * xhat[4 * c + (i >> 2)][i & 3] : 0.f;
            atomicAdd(slab_dgamma + c * 64 + f, row16_reduce_scatter(v, rl));                     // ds_add_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = valid ? dn[4 * c + (i >> 2)][i & 3] : 0.f;
            atomicAdd(slab_dbeta + c * 64 + f, row16_reduce_scatter(v, rl));
        }
    }
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        const f32x4 ga = ld4(gamma + t * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dxh = dn[t][r] * ga[r];
            dn[t][r] = dxh;
            m1 += dxh;
            m2 = fmaf(dxh, xhat[t][r], m2);
        }
    }
    m1 = (SWAP ? sum_over_lane_groups16_swap(m1) : sum_over_lane_groups16(m1)) / (float)D;
    m2 = (SWAP ? sum_over_lane_groups16_swap(m2) : sum_over_lane_groups16(m2)) / (float)D;
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dn[t][r] = rstd * (dn[t][r] - m1 - xhat[t][r] * m2);
    }
}

// Elementwise backward of one tile once z (the four pre-LayerNorm gates) is in acc.
//   forward (graphnn.py:168-170 / LayerNormBasicLSTMCell.call):
//     n_g = LN_g(z_g); c~ = c*sig(n_f+1) + sig(n_i)*relu(n_j); c' = LN_s(c~); h' = relu(c')*sig(n_o)
//   given c, dh', dc' of the tile (already in registers: a caller issues those loads before its GEMM so that their
//   latency hides behind it): leaves dz in acc, dc in dc_out.
// Gate activations are recomputed from the normalised gates where they are needed instead of being
// kept live (registers: 4D/16 xhat + a few D/16-wide temporaries).
template <int D, bool SWAP = false>
__device__ __forceinline__ void lstm_tile_backward(f32x4 (&acc)[D / 4], const f32x4 (&cf)[D / 16], const f32x4 (&dh_in)[D / 16],
                                                   const f32x4 (&dc_in)[D / 16], f32x4 (&dc_out)[D / 16],
                                                   const float* lds_ln, float* slab, int g, int rl, bool valid,
                                                   float eps_z = 1e-12f) {
    // eps_z: epsilon of the four gate LayerNorms; a caller whose z carries a factor 2^s passes 2^2s * 1e-12 -- the
    // normalised gates are then those of the unscaled z, and acc leaves as the gradient w.r.t. the SCALED z (dz / 2^s).
    constexpr int TPG = D / 16;
    f32x4 xi[TPG], xj[TPG], xf[TPG], xo[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        xi[t] = acc[t];
        xj[t] = acc[TPG + t];
        xf[t] = acc[2 * TPG + t];
        xo[t] = acc[3 * TPG + t];
    }
    const float rstd_i = ln_normalize<TPG, SWAP>(xi, D, eps_z);
    const float rstd_j = ln_normalize<TPG, SWAP>(xj, D, eps_z);
    const float rstd_f = ln_normalize<TPG, SWAP>(xf, D, eps_z);
    const float rstd_o = ln_normalize<TPG, SWAP>(xo, D, eps_z);
    auto gate = [&](const f32x4 (&xh)[TPG], int gi, int t, int r) -> float {  // LN output of gate gi
        return fmaf(xh[t][r], lds_ln[(2 * gi) * D + t * 16 + g * 4 + r], lds_ln[(2 * gi + 1) * D + t * 16 + g * 4 + r]);
    };
    f32x4 xs[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float si = sigmoidf_(gate(xi, 0, t, r));
            const float sf = sigmoidf_(gate(xf, 2, t, r) + 1.0f);
            xs[t][r] = cf[t][r] * sf + si * fmaxf(gate(xj, 1, t, r), 0.f);
        }
    }
    const float rstd_s = ln_normalize<TPG, SWAP>(xs, D);
    // through h' = relu(c')*sig(n_o) and c' = LN_s(c~)
    f32x4 dcn[TPG], don[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        const f32x4 dh = dh_in[t], dci = dc_in[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float cn = gate(xs, 4, t, r);
            const float so = sigmoidf_(gate(xo, 3, t, r));
            dcn[t][r] = dci[r] + (cn > 0.f ? dh[r] * so : 0.f);
            don[t][r] = dh[r] * fmaxf(cn, 0.f) * so * (1.0f - so);
        }
    }
    ln_backward<TPG, SWAP>(dcn, xs, rstd_s, lds_ln + 8 * D, slab + 8 * D, slab + 9 * D, g, rl, valid, D);  // dcn <- dc~
    ln_backward<TPG, SWAP>(don, xo, rstd_o, lds_ln + 6 * D, slab + 6 * D, slab + 7 * D, g, rl, valid, D);  // don <- dz_o
#pragma unroll
    for (int t = 0; t < TPG; ++t) acc[3 * TPG + t] = don[t];
    // gates f, i, j: reuse xs/don as temporaries
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sf = sigmoidf_(gate(xf, 2, t, r) + 1.0f);
            dc_out[t][r] = dcn[t][r] * sf;
            don[t][r] = dcn[t][r] * cf[t][r] * sf * (1.0f - sf);  // d n_f
        }
    }
    ln_backward<TPG, SWAP>(don, xf, rstd_f, lds_ln + 4 * D, slab + 4 * D, slab + 5 * D, g, rl, valid, D);
#pragma unroll
    for (int t = 0; t < TPG; ++t) acc[2 * TPG + t] = don[t];
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float si = sigmoidf_(gate(xi, 0, t, r));
            const float nj = gate(xj, 1, t, r);
            don[t][r] = dcn[t][r] * fmaxf(nj, 0.f) * si * (1.0f - si);  // d n_i
            xs[t][r] = nj > 0.f ? dcn[t][r] * si : 0.f;                  // d n_j
        }
    }
    ln_backward<TPG, SWAP>(don, xi, rstd_i, lds_ln + 0 * D, slab + 0 * D, slab + 1 * D, g, rl, valid, D);
    ln_backward<TPG, SWAP>(xs, xj, rstd_j, lds_ln + 2 * D, slab + 2 * D, slab + 3 * D, g, rl, valid, D);
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        acc[t] = don[t];
        acc[TPG + t] = xs[t];
    }
}

// The rows of c, dh', dc' of a tile (nullable gradients -> 0), for lstm_tile_backward.
template <int D>
__device__ __forceinline__ void lstm_tile_load(const float* c_row, const float* dh_row, const float* dcn_row, f32x4 (&cf)[D / 16],
                                               f32x4 (&dh)[D / 16], f32x4 (&dci)[D / 16]) {
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
        cf[t] = ld4(c_row + t * 16);
        dh[t] = dh_row ? ld4(dh_row + t * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        dci[t] = dcn_row ? ld4(dcn_row + t * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// LDS: [K chunk or all of K] [K^T (optional)] [ln 10*D] [NW slabs of 10*D].
constexpr int kMaxTasks = 4;

struct LstmBwdTaskTable {
    tspgnn_lstm_bwd_task task[kMaxTasks];
    int blk_end[kMaxTasks];  // exclusive prefix: task k owns workgroups [blk_end[k-1], blk_end[k])
    int qc[kMaxTasks];       // 16-row blocks of K per LDS chunk (>= all of K: resident)
    int n;
};

}  // namespace tspgnn
