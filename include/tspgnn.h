/*
 * tspgnn.h -- C ABI of libtspgnn.so: the MI355X (gfx950) implementation of the TSP-GNN
 * message-passing hot path.
 *
 * The reference (machine-reasoning-ufrgs/TSP-GNN) has no FFI of its own: the boundary of this
 * path is the TensorFlow op set its Python graph lowers to.  Each entry point below replaces
 * one such op group; the reference line it replaces is cited on the declaration
 * (paths are relative to the reference repository root).  INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the reference side.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is retained or freed;
 *   - every function takes the HIP stream to enqueue on (hipStream_t passed as void*;
 *     NULL = the default stream), launches asynchronously, never synchronises, never
 *     allocates, keeps no mutable global state (re-entrant across streams and devices);
 *   - return value: 0 = OK, <0 = TSPGNN_E* below, >0 = the hipError_t of a failed launch;
 *     nothing throws across the ABI; tspgnn_last_error() returns a thread-local message;
 *   - matrices are row-major contiguous fp32, rows of d floats; d must be 32, 64 or 128
 *     for the MFMA kernels (mlp, lnlstm), any multiple of 4 for the aggregation kernels;
 *   - index arrays are int32.
 */
#ifndef TSPGNN_H
#define TSPGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSPGNN_OK 0
#define TSPGNN_EINVAL (-1)      /* bad argument (null pointer, negative size, unsupported d) */
#define TSPGNN_EUNSUPPORTED (-2) /* valid request this build has no kernel for */

#define TSPGNN_ABI_VERSION 1

/* ABI version of the loaded library (== TSPGNN_ABI_VERSION of the header it was built from). */
int tspgnn_version(void);

/* Thread-local description of the last non-zero status returned on this thread ("" if none). */
const char* tspgnn_last_error(void);

/* ------------------------------------------------------------------ aggregation (the SpMM) */

/*
 * Y[e,:] = X[uv[e,0],:] + X[uv[e,1],:]                                 e in [0,M)
 * Replaces tf.matmul(EV, y) at graphnn.py:156-160 (E update, adjoint_a=False) for an EV whose
 * rows hold exactly two ones (instance_loader.py:63-66).  X:[N,d]  Y:[M,d]  uv:[M,2].
 */
int tspgnn_gather2_sum_f32(const int32_t* uv, const float* X, float* Y,
                           int M, int N, int d, void* stream);

/*
 * Y[v,:] = sum_{k in [rowptr[v],rowptr[v+1])} X[eid[k],:]               v in [0,N)
 * Replaces tf.matmul(EV, y, adjoint_a=True) at graphnn.py:156-160 (V update): EV^T in
 * pattern-only CSR (all stored values are 1).  Summation order is ascending k
 * (deterministic).  X:[M,d]  Y:[N,d]  rowptr:[N+1]  eid:[nnz].
 */
int tspgnn_csr_rowsum_f32(const int32_t* rowptr, const int32_t* eid, const float* X, float* Y,
                          int N, int M, int d, void* stream);

/*
 * Y[r,:] = sum_k val[k] * X[col[k],:], k in [rowptr[r],rowptr[r+1])     r in [0,R)
 * General valued CSR product for GraphNN matrices that are not 0/1 patterns
 * (graphnn.py:156-160 with an arbitrary `mat`).  X:[C,d]  Y:[R,d].
 */
int tspgnn_csr_spmm_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                        const float* X, float* Y, int R, int C, int d, void* stream);

/* ------------------------------------------------------------------ dense updates (MFMA) */

/*
 * P = W reordered into the MFMA A-fragment order the dense kernels stage into LDS (layout
 * documented in csrc/dense.hip).  W:[krows,ncols] row-major; krows % 16 == 0; ncols == 32 or
 * ncols % 64 == 0.  Run once per weight update (the reference re-reads its tf.Variables every
 * sess.run; here the packed copy is refreshed after initialisation / restore / Adam).
 */
int tspgnn_pack_weights_f32(const float* W, float* P, int krows, int ncols, void* stream);

/*
 * n_layers (1..4) chained tf.layers.Dense(d) layers: x <- act_l(x W_l + b_l); layer l applies
 * relu iff bit l of relu_mask is set.  Replaces Mlp.__call__ (mlp.py:57-63) as instantiated
 * for the message MLPs (graphnn.py:114-125,153) and the hidden part of E_vote
 * (model.py:107-115).  X,Y:[rows,d].  wb: per layer pack_weights(W[d,d] (in x out)) followed
 * by b[d], layers back to back ((d*d+d) floats each).  If acts != NULL the post-activation
 * output of every layer but the last is stored there as [n_layers-1][rows][d] (backward).
 */
int tspgnn_mlp_fwd_f32(const float* X, const float* wb, float* Y, float* acts,
                       int rows, int d, int n_layers, unsigned relu_mask, void* stream);

/*
 * One tf.contrib.rnn.LayerNormBasicLSTMCell(d, activation=relu) step, the cell call at
 * graphnn.py:168-170:  z=[x,h]K; i,j,f,o = LN_k(split(z)); c'=LN_s(c*sig(f+1)+sig(i)*relu(j));
 * h'=relu(c')*sig(o).  x:[rows,dx]  h,c,h_out,c_out:[rows,d]  K: pack_weights(kernel[dx+d,4d])
 * ln: [5][2][d] = (gamma,beta) for input, transform, forget, output, state.
 * h_out/c_out may not alias h/c.  dx must be a multiple of 16.
 */
int tspgnn_lnlstm_fwd_f32(const float* x, int dx, const float* h, const float* c,
                          const float* K, const float* ln, float* h_out, float* c_out,
                          int rows, int d, void* stream);

/* ------------------------------------------------------------------ pre / post loop */

/*
 * E0 = E_init_MLP(concat([W, C], 1)): Dense 2 -> d/8 -> d/4 -> d/2 -> d, relu x3 + linear
 * (model.py:33-43).  WC:[M,2] = (edge weight, target cost) per edge, E0:[M,d].
 * wb: the four layers back to back, each W[in,out] then b[out].
 */
int tspgnn_einit_fwd_f32(const float* WC, const float* wb, float* E0, int M, int d, void* stream);

/* Y[r,:] = v[:] for r in [0,rows): tf.tile(V_init/sqrt(d), [N,1]) (model.py:48-51), with
 * scale applied: Y = scale * v. */
int tspgnn_tile_rows_f32(const float* v, float scale, float* Y, int rows, int d, void* stream);

/* y[r] = dot(X[r,:], w) + b[0]: the final Dense(1) of E_vote (model.py:107-115,128).
 * b is a device pointer to one float. */
int tspgnn_rowdot_f32(const float* X, const float* w, const float* b, float* y,
                      int rows, int d, void* stream);

/*
 * logits[p] = mean(vote[seg[p]:seg[p+1]]) (model.py:134-145);  seg:[B+1] exclusive prefix
 * sums of n_edges.  An empty segment yields NaN like tf.reduce_mean of an empty slice.
 */
int tspgnn_segment_mean_f32(const float* vote, const int32_t* seg, float* logits,
                            int B, void* stream);

/*
 * predictions = sigmoid(logits); loss = mean(sigmoid_cross_entropy_with_logits);
 * TP/FP/TN/FN/acc with tf.round (half-to-even), formulas verbatim from model.py:147-157.
 * pred:[B]   stats:[6] = loss, acc, TP, FP, TN, FN.
 */
int tspgnn_bce_metrics_f32(const float* logits, const float* labels, float* pred, float* stats,
                           int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSPGNN_H */
