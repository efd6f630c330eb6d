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
#include "mfma_tile.h"

namespace tspgnn {

// MFMA A-fragment order of a [krows, ncols] row-major weight matrix W (the "packed" layout the
// kernels keep in LDS; produced once per weight update by tspgnn_pack_weights_f32):
//   ncols % 64 == 0:  P[(((s*4+g)*U + u)*16 + jl)*4 + tt] = W[krow(s,g)][(u*4+tt)*16 + jl],  U = ncols/64
//   ncols == 32    :  P[((s*4+g)*16 + jl)*2 + tt]         = W[krow(s,g)][tt*16 + jl]
//   krow(s,g) = (s>>2)*16 + g*4 + (s&3)      (k-step s of 4 rows, lane group g)
// One ds_read_b128 at P + ((s*4+g)*U+u)*64 + jl*4 then feeds four MFMAs (tiles 4u..4u+3) and is
// bank-conflict free (16 lanes x 16 B = one 256 B bank row per lane group).
// transposed != 0: W is stored [ncols, krows] row-major and P packs its transpose.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ W, float* __restrict__ P,
                                                           int krows, int ncols, int transposed) {
    const int total = krows * ncols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int tt, jl, u, sg;
        if (ncols == 32) {
            tt = i & 1;
            jl = (i >> 1) & 15;
            u = 0;
            sg = i >> 5;
        } else {
            const int U = ncols >> 6;
            tt = i & 3;
            jl = (i >> 2) & 15;
            u = (i >> 6) % U;
            sg = (i >> 6) / U;
        }
        const int g = sg & 3, s = sg >> 2;
        const int krow = ((s >> 2) << 4) + (g << 2) + (s & 3);
        const int col = (u * 4 + tt) * 16 + jl;
        P[i] = transposed ? W[(size_t)col * krows + krow] : W[(size_t)krow * ncols + col];
    }
}

// ---------------------------------------------------------------------------------- MLP
// Persistent workgroups: all layer weights (permuted) + biases live in LDS for the lifetime of
// the block; each wavefront pulls 16-row tiles from the block's contiguous tile range through
// an LDS ticket counter (keeps the four SIMDs of a CU evenly loaded at the tail).
// One launch may carry several independent tasks (e.g. the edge-side and the vertex-side message MLP
// of one time step): the workgroups are split among the tasks in proportion to their tiles, and each
// workgroup stages the weights of its own task.  The small vertex-side problem then rides along with
// the large edge-side one instead of paying its own launch, staging and tail.
constexpr int kMaxTasks = 4;

struct MlpTaskTable {
    tspgnn_mlp_task task[kMaxTasks];
    int blk_end[kMaxTasks];  // exclusive prefix: task k owns workgroups [blk_end[k-1], blk_end[k])
    int n;
};

template <int D, int MAXL>
__global__ __launch_bounds__(1024) void mlp_fwd_kernel(const MlpTaskTable tt) {
    constexpr int NT = D / 16;
    __shared__ __attribute__((aligned(16))) float lds[MAXL * (D * D + D) + 4];
    float* lds_w = lds;
    float* lds_b = lds + MAXL * D * D;
    int* ticket = reinterpret_cast<int*>(lds + MAXL * (D * D + D));

    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const float* __restrict__ X = tt.task[k].X;
    const float* __restrict__ wb = tt.task[k].wb;
    float* __restrict__ Y = tt.task[k].Y;
    float* __restrict__ acts = tt.task[k].acts;
    const long long acts_stride = tt.task[k].acts_stride;
    const int rows = tt.task[k].rows, n_layers = tt.task[k].n_layers;
    const unsigned relu_mask = tt.task[k].relu_mask;
    const float* __restrict__ proj_w = tt.task[k].proj_w;
    float* __restrict__ proj_out = tt.task[k].proj_out;
    const int tiles_total = (rows + 15) / 16;

    const int tid = threadIdx.x;
    for (int l = 0; l < n_layers; ++l) {
        const float* Wl = wb + (size_t)l * (D * D + D);
        copy_to_lds(lds_w + l * D * D, Wl, D * D, tid, blockDim.x);
        for (int i = tid; i < D; i += blockDim.x) lds_b[l * D + i] = Wl[D * D + i];
    }
    const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
    const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
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
            if constexpr (NT == 2) {
#pragma unroll
                for (int s = 0; s < D / 4; ++s) kstep<NT>(acc, wl + frag_off<NT>(s, g, rl), a[s >> 2][s & 3]);
            } else {
#pragma unroll
                for (int q = 0; q < NT; q += 4) {  // 16 k-steps per call
                    float b[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) b[i] = a[q + (i >> 2)][i & 3];
                    ksteps<NT, 16>(acc, wl + frag_off<NT>(q * 4, g, rl), b);
                }
            }
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
                float* dst = acts + (size_t)l * acts_stride + rbase;
#pragma unroll
                for (int t = 0; t < NT; ++t) st4(dst + t * 16, a[t]);
            }
        }
        if (valid) {
#pragma unroll
            for (int t = 0; t < NT; ++t) st4(Y + rbase + t * 16, a[t]);
        }
    }
    // Optional second phase of a (small) task: proj_out = Y * P with P = proj_w packed [D, 4D] -- the
    // vertex-side pre-multiplication Zx = V_msg_E(V.h) K_x of the folded edge update -- after restaging
    // this workgroup's LDS with P.  Saves a launch per message-passing step.
    if constexpr (D <= 64) {
        if (proj_w != nullptr) {
            constexpr int NP = D / 4;  // output tiles of the projection (4D columns)
            __threadfence_block();
            __syncthreads();           // every wavefront is done with the MLP weights and has stored its Y rows
            copy_to_lds(lds_w, proj_w, D * 4 * D, tid, blockDim.x);
            if (tid == 0) *ticket = t_beg;
            __syncthreads();
            for (;;) {
                int tile = 0;
                if (lane == 0) tile = atomicAdd(ticket, 1);
                tile = __builtin_amdgcn_readfirstlane(tile);
                if (tile >= t_end) break;
                const int row = tile * 16 + rl;
                const bool valid = row < rows;
                const size_t rc = (size_t)(valid ? row : rows - 1);
                f32x4 acc[NP];
#pragma unroll
                for (int t = 0; t < NP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* yr = Y + rc * D + g * 4;
                gemm_kloop<NP>(acc, lds_w, 0, 0, NT, yr, yr, NT, g, rl);
                if (valid) {
#pragma unroll
                    for (int t = 0; t < NP; ++t) st4(proj_out + rc * 4 * D + t * 16 + g * 4, acc[t]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------- LN-LSTM
// lstm_epilogue<D> (mfma_tile.h): split(z) = i, j, f, o; LN each; c' = LN(c*sig(f+1) + sig(i)*relu(j)); h' = relu(c')*sig(o)

// Resident variant: K ([dx+D, 4D], packed) and the five LayerNorm (gamma,beta) pairs stay in LDS
// for the lifetime of the block; requires (dx+D)*4D*4 + 10*D*4 + 16 bytes <= 160 KiB
// (D=64, dx=64: 130.5 KiB).  Wavefronts pull tiles through an LDS ticket counter.
//
// Gather-init mode (uv != NULL, dx == 0): z = Zx[uv[e,0]] + Zx[uv[e,1]] + h K_h, where
// Zx = Y_V K_x was formed once per VERTEX ([N,4D], L2-resident).  Because the adjacency product is
// linear, (EV Y) K_x = EV (Y K_x): the x-half of the cell's GEMM moves from the M edge rows to the
// N = M/19.5 vertex rows, halving this kernel's MFMA work and LDS footprint and removing the
// [M,d] aggregate from HBM altogether.
struct LstmTaskTable {
    tspgnn_lstm_task task[kMaxTasks];
    int blk_end[kMaxTasks];
    int n;
};

template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void lnlstm_fwd_kernel(const LstmTaskTable tt) {
    constexpr int NT4 = D / 4;   // output tiles of z (4D columns)
    constexpr int TPG = D / 16;  // tiles per gate
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const float* __restrict__ x = tt.task[k].x;
    const int dx = tt.task[k].dx;
    const float* __restrict__ h = tt.task[k].h;
    const float* __restrict__ c = tt.task[k].c;
    const float* __restrict__ K = tt.task[k].K;
    const float* __restrict__ ln = tt.task[k].ln;
    float* __restrict__ h_out = tt.task[k].h_out;
    float* __restrict__ c_out = tt.task[k].c_out;
    const int rows = tt.task[k].rows;
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tt.task[k].uv);
    const float* __restrict__ Zx = tt.task[k].Zx;
    const float* __restrict__ zbias = tt.task[k].zbias;
    const float* __restrict__ zscale = tt.task[k].zscale;
    const int tiles_total = (rows + 15) / 16;

    const int krows = dx + D;
    float* lds_k = lds;
    float* lds_ln = lds + (size_t)krows * 4 * D;
    int* ticket = reinterpret_cast<int*>(lds_ln + 10 * D);

    const int tid = threadIdx.x;
    copy_to_lds(lds_k, K, krows * 4 * D, tid, blockDim.x);
    for (int i = tid; i < 10 * D; i += blockDim.x) lds_ln[i] = ln[i];
    const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
    const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
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
        f32x4 acc[NT4];
        if (uv != nullptr) {
            const int2 ends = uv[rc];
            const float* zu = Zx + (size_t)ends.x * 4 * D + g * 4;
            const float* zv = Zx + (size_t)ends.y * 4 * D + g * 4;
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zu + t * 16);
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] += ld4(zv + t * 16);
        } else if (zbias != nullptr) {
            // z starts at zscale[row] * zbias: the bias of a message MLP's last layer pushed through the
            // aggregation (row-sum of (a W + b) = (row-sum a) W + degree * b) and through K_x.
            const float sc = zscale[rc];
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zbias + t * 16 + g * 4) * sc;
        } else {
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 cf[TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) cf[t] = ld4(c + rc * D + g * 4 + t * 16);
        lstm_kloop<D>(acc, lds_k, 0, 0, QT, x + rc * dx + g * 4, h + rc * D + g * 4, QX, g, rl);
        lstm_epilogue<D>(acc, cf, lds_ln, g, valid, h_out + rc * D + g * 4, c_out + rc * D + g * 4);
    }
}

// Chunked variant for a K that does not fit LDS (D=128: K is 512 KiB): the block's 8 wavefronts
// take one tile each per round and walk K in chunks of `qc` 16-row blocks that are re-staged
// into LDS between barriers (packed K is k-step major, so a chunk is one contiguous slice).
template <int D>
__global__ __launch_bounds__(512) void lnlstm_fwd_chunked_kernel(const float* __restrict__ x, int dx,
                                                                 const float* __restrict__ h,
                                                                 const float* __restrict__ c,
                                                                 const float* __restrict__ K,
                                                                 const float* __restrict__ ln,
                                                                 float* __restrict__ h_out, float* __restrict__ c_out,
                                                                 int rows, int tiles_total, int qc) {
    constexpr int NT4 = D / 4;
    constexpr int TPG = D / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_k = lds;
    float* lds_ln = lds + (size_t)qc * 16 * 4 * D;
    const int tid = threadIdx.x;
    for (int i = tid; i < 10 * D; i += blockDim.x) lds_ln[i] = ln[i];
    const int lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int QX = dx >> 4, QT = QX + TPG;
    const int rounds = (tiles_total + 7) / 8;
    for (int r = blockIdx.x; r < rounds; r += gridDim.x) {
        const int tile = r * 8 + wave;
        const bool live = tile < tiles_total;  // wave-uniform
        const int row = tile * 16 + rl;
        const bool valid = live && row < rows;
        const size_t rc = (size_t)(valid ? row : rows - 1);
        f32x4 cf[TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) cf[t] = ld4(c + rc * D + g * 4 + t * 16);
        f32x4 acc[NT4];
#pragma unroll
        for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < QT; q0 += qc) {
            const int q1 = min(QT, q0 + qc);
            __syncthreads();  // previous chunk fully consumed
            copy_to_lds(lds_k, K + (size_t)q0 * 16 * 4 * D, (q1 - q0) * 16 * 4 * D, tid, blockDim.x);
            __syncthreads();
            if (live) lstm_kloop<D>(acc, lds_k, q0, q0, q1, x + rc * dx + g * 4, h + rc * D + g * 4, QX, g, rl);
        }
        lstm_epilogue<D>(acc, cf, lds_ln, g, valid, h_out + rc * D + g * 4, c_out + rc * D + g * 4);
    }
}

// Workgroups per task, proportional to cost[k] (at least one each); grid = sum.
static int split_blocks(const long long* cost, int n, int grid, int* blk_end) {
    long long total = 0;
    for (int k = 0; k < n; ++k) total += cost[k] > 0 ? cost[k] : 1;
    if (grid < n) grid = n;
    int used = 0;
    for (int k = 0; k < n; ++k) {
        const long long ck = cost[k] > 0 ? cost[k] : 1;
        int bk = (int)((ck * grid + total / 2) / total);
        if (bk < 1) bk = 1;
        used += bk;
        blk_end[k] = used;
    }
    return used;
}

template <int D, int MAXL>
static int launch_mlp(const tspgnn_mlp_task* tasks, int n, hipStream_t st) {
    MlpTaskTable tt;
    long long cost[kMaxTasks];
    long long tiles_all = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        if (tt.task[k].acts && tt.task[k].acts_stride == 0) tt.task[k].acts_stride = (long long)tasks[k].rows * D;
        cost[k] = ((long long)tasks[k].rows + 15) / 16 * (tasks[k].n_layers + (tasks[k].proj_w ? 5 : 0));
        tiles_all += ((long long)tasks[k].rows + 15) / 16;
    }
    tt.n = n;
    // LDS per block decides residency: D=64 -> 65 KiB -> 2 blocks (16 waves) per CU.
    const int lds_bytes = MAXL * (D * D + D) * 4 + 16;
    const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
    int grid = n_cus() * per_cu;
    int nw = tiles_all <= (long long)grid * 4 ? 4 : 8;  // few tiles: one wavefront per SIMD, more workgroups
    if (per_cu == 2 && tiles_all > (long long)grid * 16) {
        // many tiles: ONE workgroup of 16 wavefronts per CU instead of two of 8 -- the whole CU shares one
        // ticket counter, so its four SIMDs finish within one tile of each other
        grid = n_cus();
        nw = 16;
    }
    const long long max_grid = (tiles_all + nw - 1) / nw;     // at least one tile per wave
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks(cost, n, grid, tt.blk_end);
    mlp_fwd_kernel<D, MAXL><<<grid, nw * 64, 0, st>>>(tt);
    return launched("tspgnn_mlp_fwd_f32");
}

template <int D>
static int launch_lnlstm_chunked(const tspgnn_lstm_task& t, hipStream_t st) {
    const int tiles = (t.rows + 15) / 16;
    const size_t extra = (10 * D + 4) * sizeof(float);
    // chunk = as many 16-row blocks of K as fit 128 KiB
    const int qc = (int)((128 * 1024) / (16 * 4 * D * sizeof(float)));
    const size_t chunked = (size_t)qc * 16 * 4 * D * sizeof(float) + extra;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_fwd_chunked_kernel<D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)chunked);
    if (e != hipSuccess) return fail((int)e, "lnlstm_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int grid = n_cus();
    const int rounds = (tiles + 7) / 8;
    if (grid > rounds) grid = rounds;
    lnlstm_fwd_chunked_kernel<D><<<grid, 512, chunked, st>>>(t.x, t.dx, t.h, t.c, t.K, t.ln, t.h_out, t.c_out, t.rows,
                                                             tiles, qc);
    return launched("tspgnn_lnlstm_fwd_f32");
}

template <int D>
static int launch_lnlstm(const tspgnn_lstm_task* tasks, int n, hipStream_t st) {
    const size_t extra = (10 * D + 4) * sizeof(float);
    const size_t kLdsMax = 160 * 1024;
    size_t resident = 0;
    bool all_fit = true;
    for (int k = 0; k < n; ++k) {
        const size_t r = (size_t)(tasks[k].dx + D) * 4 * D * sizeof(float) + extra;
        if (r > kLdsMax) all_fit = false;
        if (r > resident) resident = r;
    }
    if (!all_fit) {  // a K that does not fit LDS: one chunked launch per task
        for (int k = 0; k < n; ++k) {
            if (tasks[k].uv || tasks[k].zbias)
                return fail(TSPGNN_EUNSUPPORTED, "lnlstm_fwd: gather-init / zbias need K[%d,%d] resident in LDS", D, 4 * D);
            const int rc = launch_lnlstm_chunked<D>(tasks[k], st);
            if (rc) return rc;
        }
        return TSPGNN_OK;
    }
    LstmTaskTable tt;
    long long cost[kMaxTasks];
    long long tiles_all = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        cost[k] = tiles * ((tasks[k].dx + D) / 16 + 3);  // k-blocks + ~3 blocks' worth of epilogue
        tiles_all += tiles;
    }
    tt.n = n;
    // Few tiles (a lone vertex-side task): one wavefront per SIMD and more, smaller workgroups, so every
    // tile gets a matrix pipe to itself; many tiles: one workgroup per CU, two wavefronts per SIMD.
    int grid = n_cus();
    const int nw = tiles_all <= (long long)grid * 4 ? 4 : 8;
    const long long max_grid = (tiles_all + nw - 1) / nw;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks(cost, n, grid, tt.blk_end);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_fwd_kernel<D, 8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)resident);
    if (e != hipSuccess) return fail((int)e, "lnlstm_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    lnlstm_fwd_kernel<D, 8><<<grid, nw * 64, resident, st>>>(tt);
    return launched("tspgnn_lnlstm_fwd_f32");
}

static int check_mlp_task(const tspgnn_mlp_task& t, int d) {
    TSPGNN_REQUIRE(t.rows >= 0, "mlp_fwd: rows=%d", t.rows);
    TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 4, "mlp_fwd: n_layers=%d must be in 1..4", t.n_layers);
    TSPGNN_REQUIRE(d != 128 || t.n_layers <= 2, "mlp_fwd: d=128 holds at most 2 layers in LDS (got %d)", t.n_layers);
    TSPGNN_REQUIRE(t.rows == 0 || (t.X && t.wb && t.Y), "mlp_fwd: null pointer");
    TSPGNN_REQUIRE(!t.proj_w || (t.proj_out && (d == 32 || d == 64)), "mlp_fwd: projection needs proj_out and d in {32,64}");
    return TSPGNN_OK;
}

static int check_lstm_task(const tspgnn_lstm_task& t, int d) {
    TSPGNN_REQUIRE(t.rows >= 0, "lnlstm_fwd: rows=%d", t.rows);
    TSPGNN_REQUIRE(t.dx >= 0 && t.dx % 16 == 0, "lnlstm_fwd: dx=%d must be a non-negative multiple of 16", t.dx);
    TSPGNN_REQUIRE(t.rows == 0 || (t.h && t.c && t.K && t.ln && t.h_out && t.c_out && (t.dx == 0 || t.x)),
                   "lnlstm_fwd: null pointer");
    TSPGNN_REQUIRE(t.h_out != t.h && t.c_out != t.c, "lnlstm_fwd: outputs may not alias inputs");
    TSPGNN_REQUIRE(!t.uv || (t.dx == 0 && t.Zx && (d == 32 || d == 64)),
                   "lnlstm_fwd: gather-init mode needs dx == 0, Zx and d in {32,64}");
    TSPGNN_REQUIRE(!t.zbias || (t.zscale && !t.uv), "lnlstm_fwd: zbias needs zscale and excludes gather-init mode");
    return TSPGNN_OK;
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_pack_weights_f32(const float* W, float* P, int krows, int ncols, int transposed, void* stream) {
    TSPGNN_REQUIRE(krows >= 0 && krows % 16 == 0, "pack_weights: krows=%d must be a multiple of 16", krows);
    TSPGNN_REQUIRE(ncols == 32 || (ncols > 0 && ncols % 64 == 0), "pack_weights: ncols=%d must be 32 or a multiple of 64",
                   ncols);
    if (krows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(W && P && W != P, "pack_weights: null or aliased pointer");
    const int total = krows * ncols;
    int grid = (total + 255) / 256;
    if (grid > 1024) grid = 1024;
    pack_weights_kernel<<<grid, 256, 0, as_stream(stream)>>>(W, P, krows, ncols, transposed);
    return launched("tspgnn_pack_weights_f32");
}

extern "C" int tspgnn_mlp_fwd_multi_f32(const tspgnn_mlp_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "mlp_fwd_multi: 1..%d tasks", kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "mlp_fwd: d=%d must be 32, 64 or 128", d);
    tspgnn_mlp_task live[kMaxTasks];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const int rc = check_mlp_task(tasks[k], d);
        if (rc) return rc;
        if (tasks[k].rows > 0) live[n++] = tasks[k];
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 32: return launch_mlp<32, 4>(live, n, st);
        case 64: return launch_mlp<64, 4>(live, n, st);
        default: return launch_mlp<128, 2>(live, n, st);
    }
}

extern "C" int tspgnn_mlp_fwd_f32(const float* X, const float* wb, float* Y, float* acts, long long acts_stride,
                                  int rows, int d, int n_layers, unsigned relu_mask, void* stream) {
    if (d == 128 && n_layers > 2 && n_layers <= 4)
        return fail(TSPGNN_EUNSUPPORTED, "mlp_fwd: d=128 holds at most 2 layers in LDS (got %d)", n_layers);
    const tspgnn_mlp_task t = {X, wb, Y, acts, acts_stride, rows, n_layers, relu_mask, nullptr, nullptr, nullptr};
    return tspgnn_mlp_fwd_multi_f32(&t, 1, d, stream);
}

extern "C" int tspgnn_lnlstm_fwd_multi_f32(const tspgnn_lstm_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "lnlstm_fwd_multi: 1..%d tasks", kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "lnlstm_fwd: d=%d must be 32, 64 or 128", d);
    tspgnn_lstm_task live[kMaxTasks];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const int rc = check_lstm_task(tasks[k], d);
        if (rc) return rc;
        if (tasks[k].rows > 0) live[n++] = tasks[k];
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 32: return launch_lnlstm<32>(live, n, st);
        case 64: return launch_lnlstm<64>(live, n, st);
        default: return launch_lnlstm<128>(live, n, st);
    }
}

extern "C" int tspgnn_lnlstm_fwd_f32(const float* x, int dx, const float* h, const float* c, const float* K,
                                     const float* ln, float* h_out, float* c_out, int rows, int d, void* stream) {
    const tspgnn_lstm_task t = {x, dx, h, c, K, ln, h_out, c_out, rows, nullptr, nullptr, nullptr, nullptr, nullptr};
    return tspgnn_lnlstm_fwd_multi_f32(&t, 1, d, stream);
}

extern "C" int tspgnn_lnlstm_gather_fwd_f32(const int32_t* uv, const float* Zx, const float* h, const float* c,
                                            const float* Kh, const float* ln, float* h_out, float* c_out, int rows,
                                            int n_src, int d, void* stream) {
    TSPGNN_REQUIRE(n_src >= 0, "lnlstm_gather_fwd: n_src=%d", n_src);
    TSPGNN_REQUIRE(rows == 0 || (uv && Zx), "lnlstm_gather_fwd: null pointer");
    const tspgnn_lstm_task t = {nullptr, 0, h, c, Kh, ln, h_out, c_out, rows, uv, Zx, nullptr, nullptr, nullptr};
    return tspgnn_lnlstm_fwd_multi_f32(&t, 1, d, stream);
}
