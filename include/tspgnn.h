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

#define TSPGNN_ABI_VERSION 5   /* 2: range_flag in the task structures, pack_weights_h2 / adam_clip_step arguments;
                                  3: tspgnn_mp_loop_h2 (the whole T-step loop as one launch);
                                  4: tspgnn_mp_resident_h2 (the loop as one launch, states through memory);
                                  5: tspgnn_lstm_bwd_task.KTg / dxg, tspgnn_mlp_bwd_task.pre_X / pre_wt / pre_k (the vertex
                                     side's data-gradient GEMMs inside the step's two backward launches) */

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
 * Both aggregations of one message-passing step in one launch (they are independent: every update
 * reads the OLD states, graphnn.py:143):  Ye = EV Xv  (gather, as tspgnn_gather2_sum_f32)  and
 * Yv = EV^T Xe  (row-sum, as tspgnn_csr_rowsum_f32).  Xv:[N,d] Ye:[M,d] Xe:[M,d] Yv:[N,d]; d in
 * {32,64,128,256}.  Shares the chip between the two streams instead of two launch latencies.
 */
int tspgnn_spmm_pair_f32(const int32_t* uv, const float* Xv, float* Ye, const int32_t* rowptr,
                         const int32_t* eid, const float* Xe, float* Yv, int M, int N, int d, void* stream);

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
 * ncols % 64 == 0.  transposed != 0: W is stored [ncols,krows] and P packs W^T (the backward
 * kernels multiply by the transposed weights).  Run once per weight update (the reference re-reads its tf.Variables every
 * sess.run; here the packed copy is refreshed after initialisation / restore / Adam).
 */
int tspgnn_pack_weights_f32(const float* W, float* P, int krows, int ncols, int transposed, void* stream);

/*
 * n_layers (1..4) chained tf.layers.Dense(d) layers: x <- act_l(x W_l + b_l); layer l applies
 * relu iff bit l of relu_mask is set.  Replaces Mlp.__call__ (mlp.py:57-63) as instantiated
 * for the message MLPs (graphnn.py:114-125,153) and the hidden part of E_vote
 * (model.py:107-115).  X,Y:[rows,d].  wb: per layer pack_weights(W[d,d] (in x out)) followed
 * by b[d], layers back to back ((d*d+d) floats each).  If acts != NULL the post-activation
 * output of every layer but the last is stored at acts + l*acts_stride + row*d (backward);
 * acts_stride (floats) = 0 means rows*d.
 */
int tspgnn_mlp_fwd_f32(const float* X, const float* wb, float* Y, float* acts, long long acts_stride,
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

/*
 * The same cell step with the adjacency product folded through the GEMM (fast path of the E update,
 * graphnn.py:156-170): z = Zx[uv[e,0]] + Zx[uv[e,1]] + h Kh, where Zx = V_msg_E(V.h) Kx is formed once
 * per VERTEX by tspgnn_linear_f32 ((EV y) Kx = EV (y Kx): the x-half of the GEMM moves from the M edge
 * rows to the N vertex rows).  Zx:[n_src,4d]  Kh: pack_weights(kernel[dx:, :]) = [d,4d].
 */
int tspgnn_lnlstm_gather_fwd_f32(const int32_t* uv, const float* Zx, const float* h, const float* c,
                                 const float* Kh, const float* ln, float* h_out, float* c_out,
                                 int rows, int n_src, int d, void* stream);

/*
 * Several independent tasks in ONE launch (<= 4): the workgroups are divided among the tasks in
 * proportion to their work and each stages its own task's weights.  One message-passing step runs the
 * edge-side and vertex-side MLPs as one launch and the two cells as another, so the small vertex-side
 * problems do not pay their own launch, weight staging and tail (graphnn.py:144-171 iterates the
 * variables; the updates of one step are mutually independent because they all read the OLD states).
 * The task arrays live in HOST memory; all pointers inside are device pointers.
 */
typedef struct tspgnn_mlp_task {
    const float* X; const float* wb; float* Y; float* acts; long long acts_stride;
    int rows; int n_layers; unsigned relu_mask;
    const float* proj_w; float* proj_out;  /* optional: proj_out[rows,4d] = Y * P, proj_w = pack_weights(P[d,4d]) */
    unsigned* range_flag;  /* _h2 entry points only (optional, others ignore it): device word, |= 1 when an activation
                              left the fp16 range of the f16x2 split at the top, |= 2 (cell entry points) when a gate row's
                              spread fell below the split's absolute error at the bottom (see tspgnn_pack_weights_h2) */
} tspgnn_mlp_task;   /* fields as the arguments of tspgnn_mlp_fwd_f32 */

typedef struct tspgnn_lstm_task {
    const float* x; int dx; const float* h; const float* c; const float* K; const float* ln;
    float* h_out; float* c_out; int rows;
    const int32_t* uv; const float* Zx;   /* gather-init mode when uv != NULL: dx == 0, K = Kh */
    const float* zbias; const float* zscale; /* optional: z starts at zscale[row] * zbias[4d] (a bias folded
                                                through a row-sum aggregation: degree * (b Kx)) */
    unsigned* range_flag;                    /* as tspgnn_mlp_task.range_flag */
    int z_centered;                          /* a promise, not a request: every row of z = [x|h] K (+ Zx[u] + Zx[v], + zscale
                                                * zbias) has zero mean over each gate's d columns, because the caller centred
                                                the columns of K (and of the Kx behind Zx, and zbias) per gate -- LayerNorm
                                                subtracts that mean anyway, and the subtraction commutes with the product.
                                                The f16x2 cell kernels then skip the mean pass of the four gate LayerNorms;
                                                every other entry point ignores the field. */
} tspgnn_lstm_task;  /* fields as the arguments of tspgnn_lnlstm_fwd_f32 / tspgnn_lnlstm_gather_fwd_f32 */

int tspgnn_mlp_fwd_multi_f32(const tspgnn_mlp_task* tasks, int n_tasks, int d, void* stream);
int tspgnn_lnlstm_fwd_multi_f32(const tspgnn_lstm_task* tasks, int n_tasks, int d, void* stream);

/*
 * The same two launches on the bf16 matrix cores with fp32-class accuracy ("bf16x3": every fp32 operand is
 * split exactly into three bf16 pieces, six piece products accumulate in fp32; dropped terms <= 2^-24
 * relative).  d in {32, 64}.  Weight operands are bf16x3 packings made by tspgnn_pack_weights_x3:
 *   mlp task:  wb = n_layers blocks of { packed[3*d*d] bf16, bias[d] float };  proj_w = packed [d,4d].
 *   lstm task: K  = packed [dx+d, 4d] (dx a multiple of 32; a kernel larger than LDS is streamed in k-block
 *              chunks with the workgroup in lock step);  every other field as in the _f32 functions.
 * tspgnn_pack_weights_x3: W:[krows,ncols] row-major fp32 (krows % 32 == 0, ncols % 16 == 0) ->
 *   P: 3*krows*ncols bf16 (2 bytes each), piece-major, each piece in MFMA fragment order.
 */
int tspgnn_pack_weights_x3(const float* W, void* P, int krows, int ncols, void* stream);
int tspgnn_mlp_fwd_multi_x3(const tspgnn_mlp_task* tasks, int n_tasks, int d, void* stream);
int tspgnn_lnlstm_fwd_multi_x3(const tspgnn_lstm_task* tasks, int n_tasks, int d, void* stream);

/*
 * Cell update fused with the message MLP that consumes the new h in the NEXT time step (graphnn.py:150-170 read
 * across the step boundary: msg(h') is the same value whether it is computed at the end of step t or at the start
 * of step t+1).  Per task: the cell exactly as in tspgnn_lnlstm_fwd_multi_x3, then, on the same rows,
 *   Y = Dense chain (mlp_layers <= 4 square layers, weights/bias blocks as in tspgnn_mlp_fwd_multi_x3) of h';
 *   mlp_out:[rows,d] = Y (optional);  proj_out:[rows,4d] = Y P (optional, P = pack_weights_x3 of [d,4d]).
 * mlp_layers == 0: plain cell.
 */
typedef struct tspgnn_cell_mlp_task {
    tspgnn_lstm_task cell;
    const void* mlp_wb; int mlp_layers; unsigned relu_mask; float* mlp_out;
    const void* proj_w; float* proj_out;
    /* _h2 entry points only (the others require 0): the states h, c are read / h_out, c_out are written BLOCKED by 16
     * rows -- float4 (columns 16t + 4g .. +3) of row r at (((r/16) * d/16 + t) * 4 + g) * 64 + (r%16) * 4, buffers of
     * ceil(rows/16)*16 rows -- instead of row-major.  A state that only this launch sequence reads (the T-step loop's
     * ping-pong buffers) is then loaded and stored 1 KiB contiguous per instruction instead of as 16-byte pieces of 16
     * different rows. */
    int state_in_blocked; int state_out_blocked;
    /* _h2 entry points only (the others require NULL): training forward -- the MLP's hidden activations (outputs of its
     * layers 0 .. mlp_layers-2, after the ReLU) are also stored, layer l of row r at mlp_acts[l*mlp_acts_stride + r*d],
     * exactly what tspgnn_mlp_task.acts receives from the plain MLP launch (the backward's tape). */
    float* mlp_acts; long long mlp_acts_stride;
} tspgnn_cell_mlp_task;
int tspgnn_lnlstm_mlp_fwd_multi_x3(const tspgnn_cell_mlp_task* tasks, int n_tasks, int d, void* stream);

/*
 * The same three launches on the fp16 matrix cores with fp32-class accuracy ("f16x2": every fp32 operand is split
 * into two fp16 pieces hi + lo = x to 2^-24 relative, three piece products accumulate in fp32; see
 * csrc/dense_h2.hip) -- half the matrix instructions and less than half the split arithmetic of bf16x3; the
 * default arithmetic of the inference forward.  d in {32, 64}.  Operands as for the _x3 functions, except:
 *   tspgnn_pack_weights_h2: W:[krows,ncols] -> P: 2*krows*ncols fp16, piece-major, the pieces of 2^s * W with
 *     s = TSPGNN_H2_WEIGHT_SCALE_LOG2 (keeps the lo piece of a typical weight out of the fp16 subnormals).
 *     RANGE: the hi piece is an fp16, so 2^s |W| must stay below 65504 (|W| < 1023.5) -- fp32, the reference's type
 *     (graphnn.py:18), has no such limit.  absmax_bits (optional device word, caller-zeroed): atomically raised to
 *     the IEEE bit pattern of max |2^s W| over the matrix (0x7f800000 or above: a non-finite entry).  A caller
 *     that gets >= the bits of 65504.0f (0x477fe000) must run that network on the _x3 / _f32 entry points, which
 *     have fp32's range.  Activations: an operand row of a GEMM (h, a message MLP's hidden activation, an aggregate)
 *     with an entry >= 65504 in magnitude overflows the same way; the kernels detect it where they split the operand
 *     and set bit 0 of the task's range_flag (then the launch's outputs are not to be used: re-run on _x3 / _f32);
 *     the LOW end: the lo piece of an operand below 2^-3 is an fp16 subnormal, so the split of a small value carries an
 *     ABSOLUTE error up to 2^-25 where fp32 carries a relative one -- immaterial next to a z of ordinary size, but
 *     LayerNorm divides by the row's spread.  The cell kernels therefore keep the smallest positive variance they
 *     normalise a gate row by and set bit 1 of range_flag when it is below (2^-5)^2 in units of the unscaled z (a row
 *     of exactly zero variance -- exact operands -- does not count): same consequence as bit 0;
 *   mlp task:  wb = n_layers blocks of { packed[2*d*d] fp16, 2^s * bias[d] float }; Y comes back unscaled;
 *              proj_out = the PROJECTED-MESSAGE FORMAT of this family: 2^s * (Y P), blocked by 16 source rows -- the
 *              float4 (columns 16t + 4g .. 4g+3) of row v at float offset (((v/16) * d/4 + t) * 4 + g) * 64 + (v%16) * 4;
 *              buffer of ceil(rows/16)*16 rows x 4d floats.  It only ever feeds the z of an f16x2 cell (Zx of
 *              gather-init mode): producer tiles store 1 KiB contiguous, and the edges' gathers find consecutive
 *              vertices in one 64-byte segment;
 *   lstm task: K packed [dx+d, 4d]; Zx in the projected-message format above; zbias unscaled; c may be NULL = the
 *     zero cell state (LSTM_initial_states' default at the first step of a run): nothing is read for it (the same in
 *     tspgnn_lstm_task_bf16).
 *     The cell normalises the scaled z with epsilon 2^2s * 1e-12, which reproduces the gates of the unscaled z
 *     bit for bit (power-of-two scaling commutes with rounding).
 */
#define TSPGNN_H2_WEIGHT_SCALE_LOG2 6
float tspgnn_h2_weight_scale(void);   /* 2^TSPGNN_H2_WEIGHT_SCALE_LOG2 */
int tspgnn_pack_weights_h2(const float* W, void* P, int krows, int ncols, unsigned* absmax_bits, void* stream);
/* Every square layer of an MLP in one launch, from the variables' own layout wb = n_layers blocks {W[d,d], b[d]} (mlp.py:57-63's
 * Dense layers as a flat parameter vector holds them): transposed == 0 -> out = n_layers blocks {tspgnn_pack_weights_h2(W_l)
 * [4 d d bytes], 2^s b_l [4 d bytes]} = tspgnn_mlp_task.wb of the _h2 entry points; transposed != 0 -> out = n_layers blocks
 * tspgnn_pack_weights_h2(W_l^T) [4 d d bytes] = tspgnn_mlp_bwd_task.wt.  absmax_bits as tspgnn_pack_weights_h2.  A training
 * step repacks after every optimiser step: one launch per MLP and direction instead of three per layer. */
int tspgnn_pack_mlp_h2(const float* wb, void* out, int d, int n_layers, int transposed, unsigned* absmax_bits, void* stream);
int tspgnn_mlp_fwd_multi_h2(const tspgnn_mlp_task* tasks, int n_tasks, int d, void* stream);
/* One task followed by a Dense(1) head on the rows in hand (the vote MLP of model.py:107-115,128: three relu layers and
 * a linear d -> 1): y[r] = <out row r, head_w[d]> + head_b[0] in fp32, where `out` is what tspgnn_mlp_fwd_multi_h2
 * would have written to task->Y.  task->Y may be NULL (the rows are then never written); task->proj_w must be NULL. */
int tspgnn_mlp_head_fwd_h2(const tspgnn_mlp_task* task, const float* head_w, const float* head_b, float* y, int d,
                           void* stream);
int tspgnn_lnlstm_fwd_multi_h2(const tspgnn_lstm_task* tasks, int n_tasks, int d, void* stream);
int tspgnn_lnlstm_mlp_fwd_multi_h2(const tspgnn_cell_mlp_task* tasks, int n_tasks, int d, void* stream);

/*
 * The WHOLE T-step loop of graphnn.py:175-179 (tf.while_loop over while_body, graphnn.py:142-173) as ONE launch, for the
 * wiring of model.py:53-104: two variables, the "edge" one updated from a two-ones-per-row matrix (rows of EV,
 * instance_loader.py:63-66) and the "vertex" one from its transpose.  Arithmetic, operand formats and summation orders
 * are those of tspgnn_lnlstm_mlp_fwd_multi_h2 + tspgnn_csr_rowsum_f32 launched T times (bit-identical results); what
 * changes is where the data lives and how steps are ordered:
 *   - one workgroup per compute unit, resident for all T steps; EV is block-diagonal by instance
 *     (instance_loader.py:56-66), so a step's dependences never leave a GROUP of consecutive instances: workgroups
 *     synchronise per group through device counters (arrivals of message tiles, of aggregated vertex rows, of projected
 *     vertex tiles) instead of per step through kernel boundaries, and run ahead where their inputs are ready;
 *   - the edge states h, c stay IN REGISTERS between the steps (a wavefront owns <= 4 tiles of 16 edge rows for the whole
 *     loop); per step an edge row only reads the projected messages of its two endpoints and writes its message row;
 *   - the V<-E row-sum of a group is shared by the wavefronts that produced its messages; the vertex cells run on a few
 *     workgroups of their own, one LDS residency for the cell kernel and one for the message MLP + projection per step.
 * All buffers are caller-owned device memory; `plan` (int32, built by the host from the batch's instance sizes, see
 * tspgnn/loop_plan.py) tells every wavefront its role, its tiles, its groups and the counts it waits for:
 * TSPGNN_LOOP_DESC_INTS ints per (workgroup, wavefront), grid * TSPGNN_LOOP_WAVES descriptors in all -- the layout is
 * documented where it is written (loop_plan._build) and read (mp_loop_h2.hip).  `counters` = 3 * 32 * n_groups unsigned
 * words + 32 (three counters per group, each split by step parity), ZEROED by the caller before every launch
 * (stream-ordered).  status (optional device word): |= 1 when a wait timed out (the
 * launch's outputs are then garbage; cannot happen unless fewer than `grid` workgroups are resident).
 * msg[0] / zx[0] hold the messages / projected messages of step 0 on entry (tspgnn_mlp_fwd_multi_h2).  T >= 1.
 */
#define TSPGNN_LOOP_WAVES 8
#define TSPGNN_LOOP_DESC_INTS 24
typedef struct tspgnn_mp_loop_args {
    /* edge variable: rows M, gather-init cell (Kh, projected messages of the two endpoints), message MLP */
    const float* e_h0; const float* e_c0;   /* [M,d] row-major initial states; e_c0 NULL = zeros */
    float* e_h; float* e_c;                 /* [M,d] final states */
    const int32_t* uv;                      /* [M,2] */
    const void* e_K; const float* e_ln;     /* tspgnn_pack_weights_h2(Kh[d,4d]); LayerNorm block [10d] */
    const void* e_mlp_wb; int e_mlp_layers; unsigned e_relu_mask;
    float* msg[2];                          /* [M,d] message rows, by step parity */
    /* vertex variable: rows N, cell over [row-sum of messages | h], message MLP, projection through the edge cell's Kx */
    const float* v_h0; const float* v_c0; float* v_h; float* v_c;
    const int32_t* rowptr; const int32_t* eid;   /* CSR of EV^T: [N+1], [2M] */
    const void* v_K; const float* v_ln;     /* tspgnn_pack_weights_h2(K[2d,4d]) */
    const float* v_zbias; const float* v_zscale;   /* optional, as tspgnn_lstm_task */
    const void* v_mlp_wb; int v_mlp_layers; unsigned v_relu_mask;
    const void* v_proj_w;                   /* tspgnn_pack_weights_h2(Kx[d,4d]) */
    float* zx[2];                           /* projected messages (blocked format), by step parity */
    float* vagg[2];                         /* [N,d] scratch: the row-sums, by step parity */
    const int32_t* plan; unsigned* counters; int n_groups; int grid;
    int M; int N; int T; int z_centered;
    unsigned* range_flag; unsigned* status;
    unsigned long long* trace;              /* optional (development): 16 words per (workgroup, wavefront), sums of
                                               s_memrealtime ticks per phase of the loop (tools/loop_trace.py) */
} tspgnn_mp_loop_args;
int tspgnn_mp_loop_h2(const tspgnn_mp_loop_args* args, int d, void* stream);

/*
 * The same loop (graphnn.py:175-179 over while_body, graphnn.py:142-173; wiring of model.py:53-104) as ONE launch of
 * resident workgroups with the edge states kept in MEMORY between the steps -- the form for batches whose edge tiles
 * outnumber what tspgnn_mp_loop_h2 holds in registers (BASELINE's n=40, batch=128 and beyond).  Arithmetic, operand
 * formats and summation orders are again those of tspgnn_lnlstm_mlp_fwd_multi_h2 + tspgnn_csr_rowsum_f32 launched T
 * times (bit-identical results).  Per compute unit one workgroup of TSPGNN_RESIDENT_WAVES wavefronts:
 *   - EDGE workgroups keep Kh and the message MLP in LDS for the whole loop and hand out WORK ITEMS -- 16-row edge tiles
 *     and shares of the V<-E row-sum -- through one LDS ticket that runs through all T steps: item k is item k mod W of
 *     step k div W.  A tile's states live in private slot arrays (`e_hs`, `e_cs`: [slots * 16, d], blocked by tile,
 *     updated in place, write-through stores and L1-bypassing loads: they never leave the Infinity Cache at C2 sizes);
 *     a tile of step t waits for its own step t-1 (an LDS word per tile) and for its group's projected messages;
 *   - the items of a step are ordered by CLASS (the groups of an XCD are cut into classes, tspgnn/resident_plan.py): while
 *     the vertex chain of one class runs -- row-sum, vertex cells, message MLP, projection: ~25 us that nothing else of
 *     that class can overlap -- the workgroup's wavefronts work on the tiles of the other class;
 *   - VERTEX workgroups come in two kinds, neither with a barrier or a second LDS residency inside the loop: CELL workgroups
 *     (K[2d,4d] resident) and MESSAGE workgroups (message MLP + projection resident); h' crosses between them through `vh`.
 * Synchronisation is per group of consecutive instances through the three parity-split counters of tspgnn_mp_loop_h2
 * (message tiles arrived, vertex rows aggregated, vertex tiles projected) and a fourth (vertex tiles updated): the
 * ordering argument is the same.  `counters` = 4 * 32 * n_groups + 32 words here (the last 32: the placement check).
 * The kernel's one dependence on WHERE workgroups run -- an L1-only invalidate before re-gathering projected messages, valid
 * when a group's producers and consumers share an XCD's L2 -- is verified inside the launch and replaced by an agent-scope
 * acquire when it does not hold.
 * `plan`: int32 -- grid headers of TSPGNN_RESIDENT_HDR_INTS, then items of TSPGNN_RESIDENT_ITEM_INTS (layout documented in
 * tspgnn/resident_plan.py and csrc/mp_resident_h2.hip).  `counters` as for tspgnn_mp_loop_h2, zeroed by the caller before
 * every launch.  `lds_words`: LDS words the plan's largest edge workgroup needs behind the weights (one per tile, and
 * TSPGNN_RESIDENT_SHARE_ROWS * (1 + TSPGNN_RESIDENT_SHARE_CAP / 2) per row-sum share whose edge lists it keeps there)).
 */
#define TSPGNN_RESIDENT_WAVES 12
#define TSPGNN_RESIDENT_HDR_INTS 8
#define TSPGNN_RESIDENT_ITEM_INTS 8
#define TSPGNN_RESIDENT_SHARE_ROWS 8   /* vertex rows per row-sum share */
#define TSPGNN_RESIDENT_SHARE_CAP 48   /* edge ids (16-bit offsets from the group's first edge) per vertex row of a share kept in LDS */
typedef struct tspgnn_mp_resident_args {
    const float* e_h0; const float* e_c0;   /* [M,d] row-major initial states; e_c0 NULL = zeros */
    float* e_h; float* e_c;                 /* [M,d] final states */
    float* e_hs; float* e_cs;               /* [n_slots*16, d] scratch: the states between the steps, by tile slot */
    const int32_t* uv;
    const void* e_K; const float* e_ln;
    const void* e_mlp_wb; int e_mlp_layers; unsigned e_relu_mask;
    float* msg[2];
    const float* v_h0; const float* v_c0; float* v_h; float* v_c;
    const int32_t* rowptr; const int32_t* eid;
    const void* v_K; const float* v_ln;
    const float* v_zbias; const float* v_zscale;
    const void* v_mlp_wb; int v_mlp_layers; unsigned v_relu_mask;
    const void* v_proj_w;
    float* zx[2];
    float* vagg[2];
    float* vh[2];                           /* [N,d] scratch: the vertex states h between the steps, by step parity */
    const int32_t* plan; unsigned* counters; int n_groups; int grid; int n_slots; int lds_words;
    int n_active;                           /* workgroups of the plan with work (edge + cell + message) */
    int flags;                              /* bit 0: take the placement-independent acquire (tests; see PLACEMENT in the kernel) */
    int M; int N; int T; int z_centered;
    unsigned* range_flag; unsigned* status;
    unsigned long long* trace;              /* optional (development), as tspgnn_mp_loop_args */
} tspgnn_mp_resident_args;
int tspgnn_mp_resident_h2(const tspgnn_mp_resident_args* args, int d, void* stream);

/* ------------------------------------------------------------------ bf16 storage, fp32 accumulate
 *
 * BASELINE config 5 ("bf16 embeddings with fp32 accumulate"): the same forward operators with
 * the embeddings h, the messages, the aggregates and the projected messages Zx stored as bf16 (void* = bf16
 * rows, row-major, 16-byte aligned), GEMMs as bf16 MFMA products accumulated in fp32, and the cell state c, the
 * LayerNorm statistics / parameters, the biases and the gate arithmetic in fp32.  Weight operands are the fp32
 * variables rounded to bf16: piece 0 (the first krows*ncols bf16) of tspgnn_pack_weights_x3.
 *   tspgnn_gather2_sum_bf16 / tspgnn_csr_rowsum_bf16: tf.matmul(EV, y [, adjoint_a]) (graphnn.py:156-160), sums in
 *     fp32, one rounding at the store; d % 8 == 0 (gather), d in {32..512} (row-sum).
 *   mlp task:  wb = n_layers blocks of { bf16 packed[d*d], float bias[d] }, hidden activations rounded to bf16
 *     between layers (what a stored embedding would be); proj_w = bf16 packed [d,4d], proj_out = the bf16 projected
 *     messages Zx in the blocked format of the f16x2 family (unscaled): [rows padded to 16, 4d] bf16, the four
 *     values (cols 16t+4g..+3) of row v at element (((v/16)*(4d/16) + t)*4 + g)*64 + (v%16)*4.
 *   lstm task: x, h, h_out bf16 row-major, Zx in the blocked format above; c, c_out, ln fp32; K = bf16 packed
 *     kernel[dx+d,4d] (Kh[d,4d] in gather-init mode, uv != NULL); a kernel larger than LDS is streamed in k-block
 *     chunks.  d in {32, 64, 128}.  state_in_blocked / state_out_blocked: h and c / h_out and c_out are stored
 *     [rows padded to 16, d] blocked by 16 rows (element (((r/16)*(d/16) + t)*4 + g)*64 + (r%16)*4 for cols
 *     16t+4g..+3) instead of row-major -- for the ping-pong state buffers inside a T-step loop, which only this
 *     kernel and the message MLP (mlp task: x_blocked) read.
 */
int tspgnn_gather2_sum_bf16(const int32_t* ev_uv, const void* X, void* Y, int M, int N, int d, void* stream);
int tspgnn_csr_rowsum_bf16(const int32_t* rowptr, const int32_t* eid, const void* X, void* Y, int N, int M, int d,
                           void* stream);
typedef struct tspgnn_mlp_task_bf16 {
    const void* X; const void* wb; void* Y; int rows; int n_layers; unsigned relu_mask;
    const void* proj_w; void* proj_out;
    void* acts; long long acts_stride;   /* optional (training): the stored (bf16) hidden activations, layer l of row r
                                            at acts[l*acts_stride + r*d] (elements); stride 0 = rows*d */
    int x_blocked;                       /* X is a loop state h stored blocked by 16 rows (see the lstm task) */
    int y_interleaved;                   /* the LAST layer's packed weights and bias have their output columns permuted --
                                            packed column 16t + 4g + j (t < d/16, g, j < 4) holds true column
                                            32(t/2) + 8g + 4(t%2) + j -- so that a lane's eight results of a tile pair are
                                            eight CONSECUTIVE columns of Y: one 16-byte store, 64-byte runs per row and
                                            instruction, instead of two 8-byte stores in 32-byte runs.  Y itself is the
                                            ordinary row-major [rows, d].  Not with a projection or saved activations. */
} tspgnn_mlp_task_bf16;
typedef struct tspgnn_lstm_task_bf16 {
    const void* x; int dx; const void* h; const float* c; const void* K; const float* ln;
    void* h_out; float* c_out; int rows;
    const int32_t* uv; const void* Zx;
    int state_in_blocked; int state_out_blocked;
} tspgnn_lstm_task_bf16;
int tspgnn_mlp_fwd_multi_bf16(const tspgnn_mlp_task_bf16* tasks, int n_tasks, int d, void* stream);
int tspgnn_lnlstm_fwd_multi_bf16(const tspgnn_lstm_task_bf16* tasks, int n_tasks, int d, void* stream);

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

/* ------------------------------------------------------------------ backward (tf.gradients, model.py:166) */

/*
 * Y = X W for a weight matrix resident in LDS: X:[rows,kin], Wp = pack_weights(W[kin,n1+n2]);
 * columns [0,n1) are written to Y1:[rows,n1], the rest to Y2:[rows,n2] (added to Y2 when
 * accumulate2 != 0).  n1+n2 in {64,128,256}; kin % 16 == 0.  Used for the data gradient of the LSTM
 * GEMM: [dx | dh] = dz K^T with Wp = pack_weights(K, transposed=1).
 */
int tspgnn_linear_f32(const float* X, int kin, const float* Wp, float* Y1, int n1, float* Y2, int n2,
                      int accumulate2, int rows, void* stream);

/* Workspace (floats) tspgnn_lnlstm_bwd_f32 needs for width d. */
long long tspgnn_lnlstm_bwd_workspace_floats(int d);

/*
 * Backward of one LayerNormBasicLSTMCell step (the gradient of graphnn.py:168-170).  Inputs are the
 * forward inputs (x,h,c,K packed,ln) plus dh_out/dc_out = gradients w.r.t. the step's outputs
 * (h', c'); either may be NULL (= zero).  Recomputes z, then writes dz:[rows,4d] (gradient w.r.t.
 * z = [x,h]K, consumed by tspgnn_linear_f32 and tspgnn_wgrad_f32), dc_in:[rows,d], and ADDS the
 * LayerNorm parameter gradients to ln_grad ([5][2][d], same layout as ln).
 */
int tspgnn_lnlstm_bwd_f32(const float* x, int dx, const float* h, const float* c, const float* K,
                          const float* ln, const float* dh_out, const float* dc_out, float* dz,
                          float* dc_in, float* ln_grad, float* workspace, int rows, int d, void* stream);

/* Backward of tspgnn_lnlstm_gather_fwd_f32: same outputs as tspgnn_lnlstm_bwd_f32 (dz, dc_in, LN grads);
 * the caller turns dz into dh = dz Kh^T (tspgnn_linear_f32), dZx = EV^T dz (tspgnn_csr_rowsum_f32, d=4d)
 * and the vertex-side gradients. */
int tspgnn_lnlstm_gather_bwd_f32(const int32_t* uv, const float* Zx, const float* h, const float* c,
                                 const float* Kh, const float* ln, const float* dh_out, const float* dc_out,
                                 float* dz, float* dc_in, float* ln_grad, float* workspace, int rows, int d,
                                 void* stream);

/*
 * Data gradient of tspgnn_mlp_fwd_f32: g = dY; for l = n_layers-1..0: mask by the saved activation of
 * layer l if it had relu (acts as written by mlp_fwd; Yout = forward output, only read when the
 * last layer has relu); dpre + l*dpre_stride <- g (gradient w.r.t. the layer's pre-activation; NULL
 * to skip); g <- g W_l^T.  Finally dX (+)= g (NULL to skip).  wt: per layer
 * pack_weights(W_l, transposed=1), d*d floats each, back to back.
 */
int tspgnn_mlp_bwd_f32(const float* dY, const float* wt, const float* acts, long long acts_stride,
                       const float* Yout, float* dpre, long long dpre_stride, float* dX,
                       int accumulate_dx, int rows, int d, int n_layers, unsigned relu_mask, void* stream);

/* Several independent backward tasks in one launch (see tspgnn_mlp_fwd_multi_f32): the edge-side and
 * vertex-side cells / message MLPs of one step.  Task arrays live in HOST memory. */
typedef struct tspgnn_lstm_bwd_task {
    const float* x; int dx; const float* h; const float* c; const float* K; const float* ln;
    const float* dh_out; const float* dc_out; float* dz; float* dc_in; float* ln_grad; float* workspace;
    int rows;
    const int32_t* uv; const float* Zx;   /* gather-init mode when uv != NULL: dx == 0, K = Kh */
    const float* KT; float* dxh;          /* optional (d == 64, gather-init mode): dxh[rows,d] = dz Kh^T in the same launch,
                                             KT = tspgnn_pack_weights_f32(Kh, 4d, d, transposed=1); dz is not re-read */
    int defer_reduce;                     /* != 0: ADD the workgroups' LayerNorm-gradient partials to `workspace` (zeroed by
                                             the caller before the first launch) and leave ln_grad alone; one
                                             tspgnn_lnlstm_bwd_finish_f32 after the last time step folds them -- one
                                             reduction per cell instead of one per time step */
    const float* zbias; const float* zscale; /* _h2 entry point only (the others require NULL): the forward's z started at
                                             zscale[row] * zbias[4d] (tspgnn_lstm_task: a bias folded through a row-sum
                                             aggregation); the recomputation of z starts there too */
    const void* KTg; float* dxg;          /* _h2 entry point only (the others require NULL), d == dx == 64, no KT, no uv: the
                                             data gradient [dxg[rows,dx] | dxh[rows,d]] = dz K^T in the same launch for a cell
                                             whose K^T cannot stay in LDS beside K: KTg = tspgnn_pack_weights_h2(K^T[4d, dx+d])
                                             in device memory, its fragments streamed through the L2 (few rows: the vertex
                                             cell).  Replaces one tspgnn_linear_f32 launch per time step */
} tspgnn_lstm_bwd_task;   /* fields as the arguments of tspgnn_lnlstm_bwd_f32 / tspgnn_lnlstm_gather_bwd_f32 */

typedef struct tspgnn_mlp_bwd_task {
    const float* dY; const float* wt; const float* acts; long long acts_stride; const float* Yout;
    float* dpre; long long dpre_stride; float* dX; int accumulate_dx; int rows; int n_layers; unsigned relu_mask;
    const int32_t* uv;                    /* optional, [rows,2]: gather-init mode -- dY is an [n_src,d] array and the chain
                                             starts from dY[uv[r][0]] + dY[uv[r][1]], i.e. the adjoint of the V<-E row-sum
                                             (EV x dY, graphnn.py:156-160 with adjoint_a) without materialising it */
    int acts_bf16;                        /* != 0: acts (and Yout) are bf16 arrays -- the tape of the bf16-storage mode; they
                                             only decide the relu masks.  The tasks of one launch share the flag */
    const float* pre_X; const void* pre_wt; int pre_k;
                                          /* _h2 entry point only (the others require NULL), d == 64, no uv: the chain starts
                                             from dY = pre_X[rows, pre_k] P^T instead of reading dY (then unused, may be
                                             NULL): pre_wt = tspgnn_pack_weights_h2(P^T[pre_k, d]) in device memory, fragments
                                             streamed through the L2; pre_k a multiple of 32, at most 256.  The vertex side of
                                             a folded cell input: dY = dZx Kx^T without its own tspgnn_linear_f32 launch */
} tspgnn_mlp_bwd_task;    /* fields as the arguments of tspgnn_mlp_bwd_f32 (uv = NULL there) */

int tspgnn_lnlstm_bwd_multi_f32(const tspgnn_lstm_bwd_task* tasks, int n_tasks, int d, void* stream);

/*
 * Training in the bf16-storage mode (model.py:160-167 with graphnn.py:18's float_dtype): backward kernels that read the
 * bf16 TAPE the forward wrote -- no widened copies -- and multiply on the bf16 matrix cores.  Gradients are fp32.
 *   tspgnn_lnlstm_bwd_multi_bf16: tspgnn_lstm_bwd_task with x, h (row-major) and Zx (projected-message format above) as
 *     bf16 arrays and K = the bf16 packing of the kernel (piece 0 of tspgnn_pack_weights_x3; Kh in gather-init mode);
 *     c, dh_out, dc_out, dz, dc_in, ln, ln_grad, workspace fp32 as in tspgnn_lnlstm_bwd_multi_f32.  z is recomputed with
 *     one bf16 MFMA per product (both operands are bf16-exact: the forward's own z).  KT / dxh / zbias must be NULL.
 *     d in {32, 64, 128}; at d = 128 Kh[128,512] is resident in LDS (128 KB).
 *   tspgnn_linear_bf16w_f32: tspgnn_linear_f32 with Wp = the bf16 packing of a bf16-exact W[kin, n1+n2] (e.g. K^T for
 *     [dx | dh] = dz K^T); X fp32, split into two bf16 pieces (16 significand bits), two MFMAs per product.
 *   tspgnn_wgrad_bf16x_f32: tspgnn_wgrad_f32 with X a bf16 array (h, messages, hidden activations of the tape): two
 *     bf16 MFMA terms per product (X exact, dY in two pieces = 16 significand bits, like the fp32 operand of
 *     tspgnn_linear_bf16w_f32); kin and nout multiples of 64 (else TSPGNN_EUNSUPPORTED).
 */
int tspgnn_lnlstm_bwd_multi_bf16(const tspgnn_lstm_bwd_task* tasks, int n_tasks, int d, void* stream);
int tspgnn_linear_bf16w_f32(const float* X, int kin, const void* Wp, float* Y1, int n1, float* Y2, int n2,
                            int accumulate_y2, int rows, void* stream);
int tspgnn_wgrad_bf16x_f32(const void* X, const float* dY, long long rows, int kin, int nout, float* dW, float* db,
                           float* workspace, void* stream);
/*
 * tspgnn_lnlstm_bwd_multi_f32 with both GEMMs (the recomputation of z and dh = dz Kh^T) on the fp16 matrix cores
 * (f16x2, csrc/dense_bwd_h2.hip).  Same task structure; d in {32, 64}, dx a multiple of 32, K (and K^T) resident in LDS:
 *   K  = tspgnn_pack_weights_h2 of kernel[dx+d, 4d] (Kh[d,4d] in gather-init mode);
 *   KT = tspgnn_pack_weights_h2 of Kh^T laid out [4d, d] (row-major transpose of Kh), dx == 0;
 *   Zx = the projected messages as the f16x2 forward wrote them (its projected-message format: scaled, blocked).
 * dz, dc_in, dxh and the LayerNorm gradients come back unscaled, exactly as from the _f32 function.
 */
int tspgnn_lnlstm_bwd_multi_h2(const tspgnn_lstm_bwd_task* tasks, int n_tasks, int d, void* stream);
/* ln_grad[10d] += fixed-order sum of the deferred partials in `workspace` (tspgnn_lnlstm_bwd_workspace_floats(d)). */
int tspgnn_lnlstm_bwd_finish_f32(const float* workspace, float* ln_grad, int d, void* stream);
int tspgnn_mlp_bwd_multi_f32(const tspgnn_mlp_bwd_task* tasks, int n_tasks, int d, void* stream);
/* tspgnn_mlp_bwd_multi_f32 with the data gradient on the fp16 matrix cores (f16x2: every row of dpre_l normalised by a
 * power of two, second piece scaled into fp16's normal range and accumulated apart -- csrc/mlp_bwd_rc.hip): the same task
 * structure with wt = n_layers blocks tspgnn_pack_weights_h2(W_l^T) (4 d d bytes each); d = 64, fp32 tapes, <= 4 tasks. */
int tspgnn_mlp_bwd_multi_h2(const tspgnn_mlp_bwd_task* tasks, int n_tasks, int d, void* stream);

/*
 * Backward of a message MLP's square layers that RECOMPUTES the hidden activations instead of reading a forward tape
 * (csrc/mlp_bwd_rc.hip) -- the data gradient of graphnn.py:150-155's `msg(h)` under model.py:160-167 from nothing but the
 * chain's input, its output and the incoming gradient; both GEMM chains on the fp16 matrix cores (f16x2):
 *   a_0 = X,  a_{l+1} = act_l(a_l W_l + b_l)  (recomputed as the f16x2 forward forms them; a_L = Yout is read),
 *   dpre_l = G_{l+1} * [a_{l+1} > 0]  (G_L = dY, or dY[uv[r][0]] + dY[uv[r][1]]),   G_l = dpre_l W_l^T,   dX (+)= G_0.
 * The weight gradients { a_l^T dpre_l, colsum(dpre_l) } are formed in the same launch (`partial`).  d = 64, 1 <= n_layers <= 3.
 * Opt-in (TSPGNN_RECOMPUTE=1): parity-green, slower than the taped backward at C2.
 */
typedef struct tspgnn_mlp_bwd_rc_task {
    const float* X;          /* [rows, d]: the chain's input rows */
    const void* wb;          /* the f16x2 forward's blocks {tspgnn_pack_weights_h2(W_l), 2^s b_l} (tspgnn_mlp_task.wb) */
    const void* wt;          /* n_layers blocks tspgnn_pack_weights_h2(W_l^T), 2 d d fp16 each */
    const float* Yout;       /* [rows, d]: the chain's output as the forward wrote it (needed if the last layer has relu) */
    const float* dY;         /* gradient w.r.t. the output: [rows, d], or [n_src, d] with uv */
    const int32_t* uv;       /* optional, [rows, 2]: gather-init mode as in tspgnn_mlp_bwd_task */
    float* dX; int accumulate_dx;
    int rows; int n_layers; unsigned relu_mask;
    float* acts; long long acts_stride;      /* must be NULL / 0 (ABI 4: the form that handed the recomputed activations and */
    float* dpre; long long dpre_stride;      /* pre-activation gradients to tspgnn_wgrad_f32 was removed; layout kept)        */
    float* partial;          /* required: the weight gradients are formed in the launch --
                                partial[workgroup] += { a_l^T dpre_l , colsum(dpre_l) }_l, per workgroup [W_0, b_0, W_1, ...];
                                tspgnn_mlp_bwd_rc_partial_floats(d, n_layers) floats, zeroed by the caller before the first
                                launch of a backward pass, ACCUMULATED by every launch, folded in a fixed order by
                                tspgnn_mlp_bwd_rc_finish_f32 into the flat [W,b,W,b,...] gradient slice.  Deterministic:
                                rows, LDS slots and the order of the sums are assigned statically */
} tspgnn_mlp_bwd_rc_task;
int tspgnn_mlp_bwd_rc_h2(const tspgnn_mlp_bwd_rc_task* task, int d, void* stream);
long long tspgnn_mlp_bwd_rc_partial_floats(int d, int n_layers);
int tspgnn_mlp_bwd_rc_finish_f32(const float* partial, float* grad_wb, int d, int n_layers, void* stream);

/* Workspace (floats) tspgnn_wgrad_f32 needs. */
long long tspgnn_wgrad_workspace_floats(long long rows, int kin, int nout);

/*
 * dW[kin,nout] += X^T dY and (db != NULL) db[nout] += colsum(dY); X:[rows,kin], dY:[rows,nout],
 * kin and nout multiples of 16.  rows may span all time steps of a batch (the per-step
 * activations and pre-activation gradients are stored contiguously), so each tf.Variable gets ONE
 * reduction.  Deterministic: fixed row chunks, partials in the workspace, one pass over chunks.
 */
int tspgnn_wgrad_f32(const float* X, const float* dY, long long rows, int kin, int nout, float* dW,
                     float* db, float* workspace, void* stream);

/* dvote[e] = (sigmoid(logits[p]) - labels[p]) / (B * n_edges[p]) for e in problem p: the gradient of
 * model.py:134-157 (mean cross entropy of per-problem mean votes) w.r.t. every edge vote. */
int tspgnn_vote_grad_f32(const float* logits, const float* labels, const int32_t* seg, float* dvote,
                         int B, void* stream);

/* dX[r,:] = dy[r] * w: backward of tspgnn_rowdot_f32 through X. */
int tspgnn_rowdot_bwd_f32(const float* dy, const float* w, float* dX, int rows, int d, void* stream);

long long tspgnn_wcolsum_workspace_floats(long long rows, int d);

/*
 * out[f] += scale * sum_r wt[r] * X[r,f]  (wt == NULL: plain column sums) and, if out_wsum != NULL,
 * out_wsum[0] += scale * sum_r wt[r].  Used for the gradients of the vote head's Dense(1) kernel /
 * bias and of V_init (model.py:47-51).  Deterministic two-stage reduction.
 */
int tspgnn_wcolsum_f32(const float* X, const float* wt, long long rows, int d, float scale, float* out,
                       float* out_wsum, float* workspace, void* stream);

long long tspgnn_einit_bwd_workspace_floats(int M, int d);

/* dwb += gradient of the E_init_MLP parameters (layout of tspgnn_einit_fwd_f32's wb) given
 * dE0:[M,d]; recomputes the forward chain per edge. */
int tspgnn_einit_bwd_f32(const float* WC, const float* wb, const float* dE0, float* dwb, float* workspace,
                         int M, int d, void* stream);

long long tspgnn_adam_workspace_floats(void);

/*
 * One optimiser step on the flat parameter buffer (model.py:160-167): g += l2_scale*theta (gradient
 * of l2_scale * sum l2_loss(var)); global_norm = ||g||; g *= clip/max(global_norm, clip)
 * (clip_norm <= 0: no clipping); Adam with the bias-corrected step lr_t.  gnorm_out[0] = global_norm.
 * step_counter == NULL: lr_t is the already bias-corrected rate lr*sqrt(1-b2^t)/(1-b1^t).
 * step_counter != NULL (device int): the counter is incremented to t and lr_t is the BASE rate, corrected on
 * the device -- no host value changes between steps, so the whole training step can be replayed as a HIP graph.
 * In data-parallel training g is the all-reduced (averaged) gradient.
 * skip_flag (optional device word): when non-zero at execution time the step is NOT applied -- theta, m, v and the
 * step counter stay as they are, gnorm_out is still written.  It is the range_flag of the step's f16x2 launches: a
 * gradient computed from an overflowed forward never reaches the variables, the caller repeats the step on the
 * _x3 / _f32 entry points.
 */
int tspgnn_adam_clip_step_f32(float* theta, float* g, float* m, float* v, int n, float l2_scale,
                              float clip_norm, float lr_t, float beta1, float beta2, float eps,
                              float* gnorm_out, float* workspace, int* step_counter, const unsigned* skip_flag,
                              void* stream);

/* ------------------------------------------------------------------ host-side batch packing (no GPU work) */

/*
 * Appends the edges of one instance to a batch under construction: for every non-zero Ma[i,j] in np.nonzero
 * (row-major) order writes uv = (v_off+i, v_off+j) and W = Mw[i,j] (instance_loader.py:56-67, without the dense
 * EV).  Ma:[n,n] of kind 0=int8/bool 1=int32 2=int64 3=float32 4=float64; Mw:[n,n] float64.  HOST pointers.
 * Returns the number of edges written, or -1 on a bad argument.
 */
long long tspgnn_host_pack_instance(const void* Ma, int ma_kind, const double* Mw, int n, int v_off,
                                    int32_t* uv, double* W);

/* Tour cost per vertex exactly as instance_loader.py:70 (closing pair (route[-1], route[1])). HOST pointers. */
double tspgnn_host_route_cost(const double* Mw, int n, const int64_t* route, int len);

/* CSR of EV^T (rowptr:[N+1], eid:[2M], ascending edge ids per vertex) from the endpoint list; HOST pointers.
 * 0 = OK, -2 = endpoint out of range. */
int tspgnn_host_csr_by_vertex(const int32_t* uv, long long M, int N, int32_t* rowptr, int32_t* eid);
/* Whole-batch packing (one call instead of one per instance): per-instance host pointers / sizes in arrays.
 * tspgnn_host_count_edges fills n_edges[B] (= count_nonzero(Ma)); tspgnn_host_pack_batch then fills uv[M,2], W[M]
 * and C[M] (target_cost if use_target, else (1 -/+ dev) * tour cost for even / odd instances) and returns M. */
int tspgnn_host_count_edges(const void* const* Ma, const int* ma_kind, const int* n, int B, int64_t* n_edges);
long long tspgnn_host_pack_batch(const void* const* Ma, const int* ma_kind, const double* const* Mw, const int* n,
                                 const int64_t* const* route, const int* route_len, int B, double dev, int use_target,
                                 double target_cost, int32_t* uv, double* W, double* C);
/*
 * Parser of the reference's ".graph" instance files (dataset.py:145-187 / instance_loader.py:95-127).  Call with
 * Ma == Mw == NULL to get n and the tour length, then with Ma[n*n] (int64 0/1), Mw[n*n] (double), route[len].
 * Returns 0, or <0: -1 bad arguments, -2 cannot open, -3 no/invalid DIMENSION, -4 missing section, -5 edge out of
 * range, -6 malformed weight matrix.
 */
/* One device batch's worth of arrays packed by one call into one caller-owned (pinned) buffer: off[7] = byte offsets of
 * uv int32[M][2], eid int32[2M], rowptr int32[N+1], wc float[M][2], labels float[B], seg int32[B+1], n_edges int32[B]
 * (instance_loader.py:29-80 + the conversions of the feed, train.py:25-33).  Returns M; -1 malformed, -2 a route leaves
 * its graph, -3 M / N differ from M_expected / N_expected.  Thread-safe (thread-local scratch). */
long long tspgnn_host_stage_batch(const void* const* Ma, const int* ma_kind, const double* const* Mw, const int* n,
                                  const int64_t* const* route, const int* route_len, int B, double dev, int use_target,
                                  double target_cost, long long M_expected, int N_expected, unsigned char* stage,
                                  const long long* off);
int tspgnn_host_read_graph(const char* path, int* n_out, int* route_len_out, int64_t* Ma, double* Mw, int64_t* route);

/*
 * Storage conversions of the bf16-storage mode (BASELINE config 5; graphnn.py:18 float_dtype): y[i] = bf16(x[i]), round to
 * nearest even / y[i] = float(x[i]), n elements, both pointers 16-byte aligned.  The mode needs them at its two ends -- the
 * caller's fp32 initial embeddings in, E.h out to the fp32 vote head (model.py:118-128).
 */
int tspgnn_convert_f32_to_bf16(const float* x, void* y, long long n, void* stream);
int tspgnn_convert_bf16_to_f32(const void* x, float* y, long long n, void* stream);

/*
 * The data-parallel bucket of a training step (the reference has no counterpart: model.py:157-167 run in
 * one process).  bucket = [ gradient of the rank's mean loss (n floats) | 8-float tail ].  pack: gradient *= local_batch
 * (with_grad != 0), tail = { local_batch, local_batch * stats[0] (loss), local_batch * stats[1] (acc), stats[2..5] (TP,
 * FP, TN, FN), (float) *range_flag }; stats / range_flag may be NULL (zeros).  After ONE all-reduce(sum) of the bucket --
 * of the tail alone when only statistics are reduced: with_grad == 0 and the gradient part is not touched -- unpack
 * divides the gradient and the two means by the reduced batch size tail[0], copies the four counts, and sets
 * *range_flag = (reduced flag != 0): every rank's optimiser launch then skips or applies the step together.
 */
int tspgnn_bucket_pack_f32(float* bucket, int n, int with_grad, float local_batch, const float* stats,
                           const unsigned* range_flag, void* stream);
int tspgnn_bucket_unpack_f32(float* bucket, int n, int with_grad, float* stats, unsigned* range_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSPGNN_H */
