// fp32-accurate dense updates on the fp16 matrix cores ("f16x2").
//
// The dense work of a message-passing step (message MLPs, the cells' [x|h] K, the Kx projection) is too small per
// row to be anything but issue-bound, so what counts is the number of matrix and vector instructions per row.
// An fp16 carries 11 significand bits and rounding to nearest leaves an error <= 2^-12 relative, hence TWO fp16
// pieces  x = hi + lo,  hi = rn16(x), lo = rn16(x - hi)  represent an fp32 value to 2^-24 relative -- half an fp32
// ulp -- and the product needs three piece products,
//     a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (dropped: a_lo*b_lo <= 2^-24 relative),
// accumulated in fp32 inside v_mfma_f32_16x16x32_f16 (smallest terms first).  Against the three-piece bf16 split of
// dense_x3.hip: 3 matrix instructions per product instead of 6, two vector instructions per split value
// (v_cvt_pk_f16_f32 for a pair of hi, one v_fma_mix_f32 for x - hi, v_cvt_pk_f16_f32 for a pair of lo) instead of
// ~5.5, and 4 instead of 6 bytes of LDS per weight -- a cell's Kh[64,256] and the three message layers are resident
// in 112 KB with no piece left in L1.
//
// Range.  The lo piece of a value below 2^-2 is an fp16 subnormal (kept by the conversion and by the MFMA on gfx950 --
// tools/f16_denorm_probe.hip), i.e. carries an ABSOLUTE error up to 2^-25.  For activations (O(1) rows) that is below
// the fp32 rounding of the row's large entries; for weights (|w| ~ 0.1) it would cost a factor 2-4 in end-to-end
// accuracy (tools/h2_numerics.py), so the WEIGHT operand is packed pre-multiplied by 2^6 and the scale is taken out
// where that is free or nearly so:
//   * cell: every term of z carries the factor (weights, the projected messages Zx, the folded bias), and the four gate
//     LayerNorms run with epsilon 2^12 * 1e-12 -- a power-of-two scale commutes with every rounding, so the normalised
//     gates are bit-identical to those of the unscaled z;
//   * Dense layer: the bias block is stored pre-scaled and the output is multiplied by 2^-6 next to the relu.
// Weights must stay below 2^10 in magnitude (fp16 overflow of the scaled hi piece); activations below 65504.
//
// Layout: the "transposed chaining" of dense.hip / dense_x3.hip (OUT^T = W^T IN^T, a wavefront owns 16 rows, the D
// fragment of one layer is the next layer's B operand in the same lane).  A k-block covers 32 features: lane (rl, g)
// supplies the 8 features 16*(2kb + (j>>2)) + 4g + (j&3), j = 0..7; the packed weights use the same permutation.
#include "common.h"
#include "h2_tile.h"
#include "mfma_tile.h"

namespace tspgnn {

// The cell launch's output rows (the states h', c' and the next step's messages) go through st4o<WT> (h2_tile.h).  WT --
// write-through stores -- pays when the loop's arrays live in the Infinity Cache (C2: forward 1.467-1.485 -> 1.425-1.441 ms,
// five alternating runs on one box) and costs when they do not (C4: 10.31-10.43 -> 10.43-10.82 ms), so the launcher chooses by
// footprint (launch_cell_h2); profiles/r04_store_flavour_ab.txt.


// Packed weights of a [krows, ncols] matrix: P[piece][kb][g][t][jl][8] (fp16), piece 0 = hi, 1 = lo of 2^s * W,
//   value = piece(2^s * W[16*(2kb + (j>>2)) + 4g + (j&3)][t*16 + jl]),   KB = krows/32, NT = ncols/16.
// One ds_read_b128 per (piece, kb, t) and lane: 16 lanes x 16 B contiguous, lane groups a multiple of 256 B apart.
__global__ __launch_bounds__(256) void pack_weights_h2_kernel(const float* __restrict__ W, _Float16* __restrict__ P,
                                                              int krows, int ncols, unsigned* __restrict__ absmax_bits) {
    const int NT = ncols >> 4;
    const int total = krows * ncols;  // per piece
    unsigned mx = 0u;   // bit pattern of max |2^s W| (non-negative floats order like their bit patterns; NaN sorts above inf)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, jl = (i >> 3) & 15;
        int rest = i >> 7;
        const int t = rest % NT;
        rest /= NT;
        const int g = rest & 3, kb = rest >> 2;
        const int k = 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3);
        const float x = kH2Scale * W[(size_t)k * ncols + t * 16 + jl];
        const _Float16 h = (_Float16)x;
        P[i] = h;
        P[(size_t)total + i] = (_Float16)(x - (float)h);
        const unsigned b = __float_as_uint(x) & 0x7fffffffu;
        mx = b > mx ? b : mx;
    }
    if (absmax_bits != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)mx, off);
            mx = o > mx ? o : mx;
        }
        if ((threadIdx.x & 63) == 0 && mx != 0u) atomicMax(absmax_bits, mx);
    }
}

// All square layers of an MLP in ONE launch (blockIdx.y = layer), straight from the variables' flat {W [d,d], b [d]} blocks:
// forward form {pack(2^s W), 2^s b} per layer (what tspgnn_mlp_task.wb takes) or, transposed, pack(2^s W^T) back to back
// (tspgnn_mlp_bwd_task.wt).  A training step rebuilds every packing after the optimiser has moved the variables: per layer
// that was a pack launch, a bias scaling and a copy -- some fifty five-microsecond launches per step.
__global__ __launch_bounds__(256) void pack_mlp_h2_kernel(const float* __restrict__ wb, unsigned char* __restrict__ out, int d,
                                                          int transposed, unsigned* __restrict__ absmax_bits) {
    const int l = blockIdx.y, NT = d >> 4, total = d * d;
    const float* W = wb + (size_t)l * (total + d);
    const size_t per = transposed ? (size_t)4 * total : (size_t)4 * total + 4 * d;
    _Float16* P = reinterpret_cast<_Float16*>(out + (size_t)l * per);
    unsigned mx = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, jl = (i >> 3) & 15;
        int rest = i >> 7;
        const int t = rest % NT;
        rest /= NT;
        const int g = rest & 3, kb = rest >> 2;
        const int k = 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3), c = t * 16 + jl;
        const float x = kH2Scale * (transposed ? W[(size_t)c * d + k] : W[(size_t)k * d + c]);
        const _Float16 h = (_Float16)x;
        P[i] = h;
        P[(size_t)total + i] = (_Float16)(x - (float)h);
        const unsigned b = __float_as_uint(x) & 0x7fffffffu;
        mx = b > mx ? b : mx;
    }
    if (!transposed && blockIdx.x == 0) {
        float* bias = reinterpret_cast<float*>(out + (size_t)l * per + (size_t)4 * total);
        for (int f = threadIdx.x; f < d; f += blockDim.x) bias[f] = kH2Scale * W[total + f];
    }
    if (absmax_bits != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)mx, off);
            mx = o > mx ? o : mx;
        }
        if ((threadIdx.x & 63) == 0 && mx != 0u) atomicMax(absmax_bits, mx);
    }
}

constexpr int kMaxTasksH2 = 4;

// ---------------------------------------------------------------------------------- MLP (f16x2)
// Task fields as tspgnn_mlp_task; wb points at n_layers blocks of { fp16 packed[2*D*D] , float bias[D] (= 2^s b) };
// proj_w at an fp16 packed [2 * D * 4D] matrix; proj_out = 2^s * (Y P) in the blocked layout of h2_tile.h: it feeds the
// scaled z of an f16x2 cell.
struct MlpTaskTableH2 {
    tspgnn_mlp_task task[kMaxTasksH2];
    int blk_end[kMaxTasksH2];
    int n;
    // tspgnn_mlp_head_fwd_h2 (one task): y[row] = <last layer's output row, head_w> + head_b[0], taken while the row is
    // still in registers; the task's Y may then be NULL (nothing is written for it)
    const float* head_w;
    const float* head_b;
    float* head_y;
};

template <int D>
__global__ __launch_bounds__(1024) void mlp_fwd_h2_kernel(const MlpTaskTableH2 tt) {
    constexpr int NT = D / 16, KB = D / 32;
    constexpr int LAYER_BYTES = 2 * D * D * 2 + D * 4;
    constexpr int WBYTES = (4 * LAYER_BYTES > 2 * D * 4 * D * 2) ? 4 * LAYER_BYTES : 2 * D * 4 * D * 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[WBYTES + 16];
    int* ticket = reinterpret_cast<int*>(lds + WBYTES);

    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const float* __restrict__ X = tt.task[k].X;
    const unsigned char* __restrict__ wb = reinterpret_cast<const unsigned char*>(tt.task[k].wb);
    float* __restrict__ Y = tt.task[k].Y;
    float* __restrict__ acts = tt.task[k].acts;
    const long long acts_stride = tt.task[k].acts_stride;
    const int rows = tt.task[k].rows, n_layers = tt.task[k].n_layers;
    const unsigned relu_mask = tt.task[k].relu_mask;
    const _Float16* __restrict__ proj_w = reinterpret_cast<const _Float16*>(tt.task[k].proj_w);
    float* __restrict__ proj_out = tt.task[k].proj_out;
    const int tiles_total = (rows + 15) / 16;

    const int tid = threadIdx.x;
    h2_copy_to_lds(lds, wb, n_layers * LAYER_BYTES, tid, blockDim.x);
    const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
    const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
    if (tid == 0) *ticket = t_beg;
    h2_stage_wait();
    __syncthreads();

    const int lane = tid & 63, rl = lane & 15, g = lane >> 4;
    float wit = 0.f;   // fp16-overflow witness of this wavefront's operand splits (h2_tile.h)
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
            const _Float16* wl = reinterpret_cast<const _Float16*>(lds + (size_t)l * LAYER_BYTES);
            const float* bl = reinterpret_cast<const float*>(lds + (size_t)l * LAYER_BYTES + 2 * D * D * 2);
            dense_layer_h2<D>(a, wl, wl + D * D, bl, (relu_mask >> l) & 1u, g, rl, wit);
            if (acts != nullptr && l < n_layers - 1 && valid) {
                float* dst = acts + (size_t)l * acts_stride + rbase;
#pragma unroll
                for (int t = 0; t < NT; ++t) st4(dst + t * 16, a[t]);
            }
        }
        if (tt.head_y != nullptr) {   // Dense(1) on the rows in hand: this lane holds columns t*16 + g*4 .. +3
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 wv = ld4(tt.head_w + g * 4 + t * 16);
                s = fmaf(a[t][3], wv[3], fmaf(a[t][2], wv[2], fmaf(a[t][1], wv[1], fmaf(a[t][0], wv[0], s))));
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (valid && g == 0) tt.head_y[row] = s + tt.head_b[0];
        }
        if (valid && Y != nullptr) {
#pragma unroll
            for (int t = 0; t < NT; ++t) st4(Y + rbase + t * 16, a[t]);
        }
    }
    // second phase of a (small) task: proj_out = 2^s Y P, P packed [D, 4D]
    if (proj_w != nullptr) {
        constexpr int NP = D / 4;
        __threadfence_block();
        __syncthreads();
        h2_copy_to_lds(lds, proj_w, 2 * D * 4 * D * 2, tid, blockDim.x);
        if (tid == 0) *ticket = t_beg;
        h2_stage_wait();
        __syncthreads();
        const _Float16* wp = reinterpret_cast<const _Float16*>(lds);
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
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f32x4 lo4 = ld4(yr + (2 * kb) * 16), hi4 = ld4(yr + (2 * kb + 1) * 16);
                float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                f16x8 bh, bl;
                split2w(x, bh, bl, wit);
                kblock_h2<NP>(acc, wp, wp + D * 4 * D, kb, g, rl, bh, bl);
            }
            if (valid) {
#pragma unroll
                for (int t = 0; t < NP; ++t) st4(proj_out + h2_zx_row<D>((unsigned)rc, g) + t * 256, acc[t]);
            }
        }
    }
    h2_range_report(tt.task[k].range_flag, wit);
}

// ---------------------------------------------------------------------------------- LN-LSTM (+ MLP) (f16x2)
// z = 2^s ([x|h] K (+ gather-init / bias-init)), five LayerNorms and the gate arithmetic of dense.hip's cell --
// optionally followed, on the same 16 rows while h' is still in registers, by the message MLP that consumes h' in the
// NEXT time step (and its projection through the receiving cell's Kx), exactly as lnlstm_mlp_fwd_x3_kernel.
//   resident mode  -- K and the MLP layers fit LDS together (Kh of the edge cell in gather-init mode + three layers:
//     112 KB at D=64): staged once per workgroup; 16-row tiles are handed out by an LDS ticket.
//   lock-step mode -- otherwise (the vertex cell's [2D,4D] + MLP + projection): one tile per wavefront per round, the
//     workgroup in lock step through two LDS residencies -- K (whole if it fits alone: 128 KB at D=64, else in
//     k-block chunks), then the MLP layers together with the projection matrix (130 KB; one after the other when
//     they do not fit together).  The second staging is issued as soon as the last wavefront has left the K GEMM
//     and lands behind the LayerNorm / gate arithmetic.
struct CellTaskTableH2 {
    tspgnn_cell_mlp_task task[kMaxTasksH2];
    int blk_end[kMaxTasksH2];
    int kbc[kMaxTasksH2];       // k-blocks (32 rows of K) per LDS chunk
    int lockstep[kMaxTasksH2];  // 0: K and the MLP resident together, tiles by ticket; 1: lock-step rounds
    int together[kMaxTasksH2];  // lock-step: MLP layers and projection matrix staged together
    int lock_tiles[kMaxTasksH2];  // lock-step: tiles (= working wavefronts) per workgroup and round, <= wavefronts per workgroup
    int n;
};

// CENTERED: every task of the launch promises z_centered (include/tspgnn.h) -- the gate LayerNorms run without their
// mean pass (a compile-time variant: the same choice as a branch inside the tile loop cost 47 spilled registers).
// WT: write-through output stores (st4o).  12 wavefronts per workgroup (<= 168 registers; a 16-wavefront build spilled and
// was 4 % slower).
template <int D, bool CENTERED, bool WT>
__global__ __launch_bounds__(768) void lnlstm_mlp_fwd_h2_kernel(const CellTaskTableH2 tt) {
    constexpr int NT4 = D / 4, TPG = D / 16, KBH = D / 32;
    constexpr int LAYER_BYTES = 2 * D * D * 2 + D * 4;  // { hi, lo, bias } of one MLP layer
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const tspgnn_lstm_task& tk = tt.task[k].cell;
    const float* __restrict__ x = tk.x;
    const int dx = tk.dx;
    const float* __restrict__ h = tk.h;
    const float* __restrict__ c = tk.c;
    const _Float16* __restrict__ K = reinterpret_cast<const _Float16*>(tk.K);
    const float* __restrict__ ln = tk.ln;
    float* __restrict__ h_out = tk.h_out;
    float* __restrict__ c_out = tk.c_out;
    const int rows = tk.rows;
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tk.uv);
    const float* __restrict__ Zx = tk.Zx;
    const float* __restrict__ zbias = tk.zbias;
    const float* __restrict__ zscale = tk.zscale;
    const unsigned char* __restrict__ mlp_wb = reinterpret_cast<const unsigned char*>(tt.task[k].mlp_wb);
    const int n_layers = tt.task[k].mlp_layers;
    const unsigned relu_mask = tt.task[k].relu_mask;
    float* __restrict__ mlp_out = tt.task[k].mlp_out;
    const _Float16* __restrict__ proj_w = reinterpret_cast<const _Float16*>(tt.task[k].proj_w);
    float* __restrict__ proj_out = tt.task[k].proj_out;
    float* __restrict__ mlp_acts = tt.task[k].mlp_acts;   // training: hidden activations of the MLP, [n_layers-1][stride]
    const long long acts_stride = tt.task[k].mlp_acts_stride;
    const bool in_blk = tt.task[k].state_in_blocked != 0, out_blk = tt.task[k].state_out_blocked != 0;
    const int in_ts = in_blk ? 256 : 16, out_ts = out_blk ? 256 : 16;   // floats between the 16-column tiles of a state row
    const int tiles_total = (rows + 15) / 16;
    const int KBT = (dx + D) >> 5;       // k-blocks in total
    const int kbc = tt.kbc[k];
    const bool resident = !tt.lockstep[k];
    const int KBX = dx >> 5;             // k-blocks that come from x
    const int total = (dx + D) * 4 * D;  // elements per piece of the whole matrix
    const int chunk_total = kbc * 32 * 4 * D;

    // LDS: [ln 10*D floats][ticket, pad][weights region]
    float* lds_ln = reinterpret_cast<float*>(ldsb);
    int* ticket = reinterpret_cast<int*>(lds_ln + 10 * D);
    unsigned char* lds_wb = ldsb + (10 * D + 4) * sizeof(float);
    _Float16* lds_w = reinterpret_cast<_Float16*>(lds_wb);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int rl = lane & 15, g = lane >> 4;   // (re-derived per tile inside the loops: see opaque_lane)
    float wit = 0.f;                     // fp16-overflow witness of this wavefront's operand splits (h2_tile.h)
    unsigned vmin = 0xffffffffu;         // smallest positive gate variance normalised by (low end of the range, h2_tile.h)
    // LayerNorm parameters, rows [g_i, b_i, g_j, b_j, g_f, b_f, g_o, b_o, g_s, b_s]: the gates i, f, o feed sigmoids
    // only, so their gamma / beta are stored times -log2(e) with the forget bias folded into b_f (lstm_gates<D, true>)
    for (int i = tid; i < 10 * D; i += blockDim.x) {
        const int r = i / D;
        float v = ln[i];
        if (r == 5) v += 1.0f;
        if (r < 2 || (r >= 4 && r < 8)) v *= kNegLog2e;
        lds_ln[i] = v;
    }

    // stage k-blocks [kb0, kb1) of both pieces (each piece is k-block major in global memory)
    auto stage = [&](int kb0, int kb1) {
        const int n = (kb1 - kb0) * 32 * 4 * D;  // elements per piece
        for (int p = 0; p < 2; ++p)
            h2_copy_to_lds(lds_w + (size_t)p * chunk_total, K + (size_t)p * total + (size_t)kb0 * 32 * 4 * D, n * 2, tid,
                           blockDim.x);
    };
    // z starts at 2^s * (its non-GEMM part); Zx is stored scaled by its producer (the f16x2 projection)
    auto init_acc = [&](f32x4 (&acc)[NT4], unsigned rc, bool live) {
        if (uv != nullptr && live) {
            const int2 ends = uv[rc];
            const float* zu = Zx + h2_zx_row<D>((unsigned)ends.x, g);
            const float* zv = Zx + h2_zx_row<D>((unsigned)ends.y, g);
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zu + t * 256);
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] += ld4(zv + t * 256);
        } else if (zbias != nullptr) {
            // (an idle wavefront of a lock-step round keeps z == 0 in EVERY mode -- it skips the GEMM, and a partial z is
            // not a row of the batch: a variance of exactly 0 is ignored by the range guard, see ln_gate<..., TRACK>)
            const float sc = live ? zscale[rc] * kH2Scale : 0.f;
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zbias + t * 16 + g * 4) * sc;
        } else {
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // k-blocks [kb0, kb1) of the concatenated [x | h] operand; lds_w holds the chunk starting at kb_base
    auto kloop = [&](f32x4 (&acc)[NT4], unsigned rc, int kb_base, int kb0, int kb1) {
        const float* xrow = x + (rc * (unsigned)dx + g * 4);
        const float* hrow = h + h2_state_row<D>(rc, g, in_blk);
        for (int kb = kb0; kb < kb1; ++kb) {
            const float* src = kb < KBX ? xrow + kb * 32 : hrow + (kb - KBX) * 2 * in_ts;
            const f32x4 lo4 = ld4(src), hi4 = ld4(src + (kb < KBX ? 16 : in_ts));
            float xv[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            f16x8 bh, bl;
            split2w(xv, bh, bl, wit);
            kblock_h2<NT4>(acc, lds_w, lds_w + chunk_total, kb - kb_base, g, rl, bh, bl);
        }
    };
    // gates + state stores; returns h' in registers (the D layout is the next GEMM's B operand)
    auto cell = [&](f32x4 (&acc)[NT4], f32x4 (&cf)[TPG], unsigned rc, bool valid, f32x4 (&hn)[TPG]) {
        f32x4 nc[TPG];
        lstm_gates<D, true, H2_LN_SWAP != 0, CENTERED, true>(acc, cf, lds_ln, g, hn, nc, kH2GateEps, &vmin);
        if (valid) {
            float* hd = h_out + h2_state_row<D>(rc, g, out_blk);
            float* cd = c_out + h2_state_row<D>(rc, g, out_blk);
#pragma unroll
            for (int t = 0; t < TPG; ++t) {
                st4o<WT>(hd + t * out_ts, hn[t]);
                st4o<WT>(cd + t * out_ts, nc[t]);
            }
        }
    };

    if (resident) {
        // weights region: [K 2 pieces][MLP layers { hi, lo, bias } as in global memory]
        unsigned char* lds_mlp = lds_wb + (size_t)2 * chunk_total * 2;
        stage(0, KBT);
        if (n_layers > 0) h2_copy_to_lds(lds_mlp, mlp_wb, n_layers * LAYER_BYTES, tid, blockDim.x);
        const int pos = xcd_contiguous(my_blk, my_grid);   // the edges of an XCD gather from one slice of Zx
        const int t_beg = (int)((long long)tiles_total * pos / my_grid);
        const int t_end = (int)((long long)tiles_total * (pos + 1) / my_grid);
        if (tid == 0) *ticket = t_beg;
        h2_stage_wait();
        __syncthreads();
        {
        for (;;) {
            int tile = 0;
            if (lane == 0) tile = atomicAdd(ticket, 1);
            tile = __builtin_amdgcn_readfirstlane(tile);
            if (tile >= t_end) break;
            {   // the lane's coordinates in the tile, recomputed (two VALU instructions) instead of kept: as loop invariants
                // they and what is derived from them were SPILLED to scratch, a reload on every tile's critical path
                const int l = opaque_lane();
                rl = l & 15;
                g = l >> 4;
            }
            const int row = tile * 16 + rl;
            const bool valid = row < rows;
            const unsigned rc = (unsigned)(valid ? row : rows - 1);
            f32x4 hn[TPG];
            {
                f32x4 acc[NT4], cf[TPG];
                init_acc(acc, rc, true);
#pragma unroll
                for (int t = 0; t < TPG; ++t)
                    cf[t] = c != nullptr ? ld4(c + h2_state_row<D>(rc, g, in_blk) + t * in_ts) : f32x4{0.f, 0.f, 0.f, 0.f};
                kloop(acc, rc, 0, 0, KBT);
                cell(acc, cf, rc, valid, hn);
            }
            if (n_layers > 0) {
                for (int l = 0; l < n_layers; ++l) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_mlp + (size_t)l * LAYER_BYTES);
                    const float* bias = reinterpret_cast<const float*>(lds_mlp + (size_t)l * LAYER_BYTES + 2 * D * D * 2);
                    dense_layer_h2<D>(hn, wh, wh + D * D, bias, (relu_mask >> l) & 1u, g, rl, wit);
                    if (mlp_acts != nullptr && l < n_layers - 1 && valid) {
                        float* dst = mlp_acts + ((size_t)l * acts_stride + (size_t)rc * D + g * 4);
#pragma unroll
                        for (int t = 0; t < TPG; ++t) st4(dst + t * 16, hn[t]);
                    }
                }
                if (valid && mlp_out != nullptr) {
#pragma unroll
                    for (int t = 0; t < TPG; ++t) st4o<WT>(mlp_out + (rc * D + g * 4 + t * 16), hn[t]);
                }
            }
        }
        }
    } else {
        // lock-step rounds: one tile per wavefront; K (whole or chunk by chunk), then the MLP + the projection
        const bool together = tt.together[k] != 0;
        unsigned char* lds_proj = together ? lds_wb + (size_t)n_layers * LAYER_BYTES : lds_wb;
        // (fewer working wavefronts per workgroup than it has -- lock_tiles -- spread the task over more CUs: its GEMMs
        // are bound by the matrix pipes of the few CUs it runs on)
        const int lw = tt.lock_tiles[k];
        const int rounds = (tiles_total + lw - 1) / lw;
        for (int r = my_blk; r < rounds; r += my_grid) {
            {
                const int l = opaque_lane();
                rl = l & 15;
                g = l >> 4;
            }
            const int tile = r * lw + wave;
            const bool live = wave < lw && tile < tiles_total;
            const int row = tile * 16 + rl;
            const bool valid = live && row < rows;
            const unsigned rc = (unsigned)(valid ? row : rows - 1);
            f32x4 hn[TPG];
            {
                f32x4 acc[NT4], cf[TPG];
                init_acc(acc, rc, live);
                // The [x | h] operand rows of up to four k-blocks are fetched before the K staging is waited for: every
                // wavefront of the workgroup is in the same phase here, nobody hides a global round trip per k-block.
                constexpr int KBP = 4;
                const bool pre = KBT <= KBP;
                f32x4 opr[2 * KBP];
                if (pre) {
                    const float* xrow = x + (rc * (unsigned)dx + g * 4);
                    const float* hrow = h + h2_state_row<D>(rc, g, in_blk);
#pragma unroll
                    for (int kb = 0; kb < KBP; ++kb) {
                        if (kb < KBT) {
                            const float* src = kb < KBX ? xrow + kb * 32 : hrow + (kb - KBX) * 2 * in_ts;
                            opr[2 * kb] = ld4(src);
                            opr[2 * kb + 1] = ld4(src + (kb < KBX ? 16 : in_ts));
                        }
                    }
                }
                for (int kb0 = 0; kb0 < KBT; kb0 += kbc) {
                    const int kb1 = min(KBT, kb0 + kbc);
                    __syncthreads();
                    stage(kb0, kb1);
                    h2_stage_wait();
                    __syncthreads();
                    if (live && pre) {
#pragma unroll
                        for (int kb = 0; kb < KBP; ++kb) {
                            if (kb >= kb0 && kb < kb1) {
                                const f32x4 lo4 = opr[2 * kb], hi4 = opr[2 * kb + 1];
                                float xv[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                                f16x8 bh, bl;
                                split2w(xv, bh, bl, wit);
                                kblock_h2<NT4>(acc, lds_w, lds_w + chunk_total, kb - kb0, g, rl, bh, bl);
                            }
                        }
                    } else if (live) {
                        kloop(acc, rc, kb0, kb0, kb1);
                    }
                }
                if (n_layers > 0) {  // every wavefront is done with K: the next residency loads behind the gates
                    __syncthreads();
                    h2_copy_to_lds(lds_wb, mlp_wb, n_layers * LAYER_BYTES, tid, blockDim.x);
                    if (proj_w != nullptr && together) h2_copy_to_lds(lds_proj, proj_w, 2 * D * 4 * D * 2, tid, blockDim.x);
                }
                // (the row of c is fetched only now: held across the GEMM next to the preloaded operands it was spilled --
                // behind a wait for every load in flight; the four gate LayerNorms that precede its first use cover it)
#pragma unroll
                for (int t = 0; t < TPG; ++t)
                    cf[t] = c != nullptr ? ld4(c + h2_state_row<D>(rc, g, in_blk) + t * in_ts) : f32x4{0.f, 0.f, 0.f, 0.f};
                cell(acc, cf, rc, valid, hn);
            }
            if (n_layers > 0) {
                h2_stage_wait();
                __syncthreads();
                for (int l = 0; l < n_layers; ++l) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_wb + (size_t)l * LAYER_BYTES);
                    const float* bias = reinterpret_cast<const float*>(lds_wb + (size_t)l * LAYER_BYTES + 2 * D * D * 2);
                    dense_layer_h2<D>(hn, wh, wh + D * D, bias, (relu_mask >> l) & 1u, g, rl, wit);
                    if (mlp_acts != nullptr && l < n_layers - 1 && valid) {
                        float* dst = mlp_acts + ((size_t)l * acts_stride + (size_t)rc * D + g * 4);
#pragma unroll
                        for (int t = 0; t < TPG; ++t) st4(dst + t * 16, hn[t]);
                    }
                }
                if (valid && mlp_out != nullptr) {
#pragma unroll
                    for (int t = 0; t < TPG; ++t) st4o<WT>(mlp_out + (rc * D + g * 4 + t * 16), hn[t]);
                }
                if (proj_w != nullptr) {  // proj_out = 2^s mlp(h') P, P packed [D, 4D]
                    if (!together) {
                        __syncthreads();
                        h2_copy_to_lds(lds_proj, proj_w, 2 * D * 4 * D * 2, tid, blockDim.x);
                        h2_stage_wait();
                        __syncthreads();
                    }
                    const _Float16* wp = reinterpret_cast<const _Float16*>(lds_proj);
                    f32x4 acc[NT4];
#pragma unroll
                    for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < KBH; ++kb) {
                        float xv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = hn[2 * kb + (j >> 2)][j & 3];
                        f16x8 bh, bl;
                        split2w(xv, bh, bl, wit);
                        kblock_h2<NT4>(acc, wp, wp + D * 4 * D, kb, g, rl, bh, bl);
                    }
                    if (valid) {
#pragma unroll
                        for (int t = 0; t < NT4; ++t) st4(proj_out + h2_zx_row<D>(rc, g) + t * 256, acc[t]);
                    }
                }
            }
        }
    }
    h2_range_report(tk.range_flag, wit, vmin);
}

static int split_blocks_h2(const long long* cost, int n, int grid, int* blk_end) {
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

template <int D>
static int launch_mlp_h2(const tspgnn_mlp_task* tasks, int n, hipStream_t st, const float* head_w = nullptr,
                         const float* head_b = nullptr, float* head_y = nullptr) {
    MlpTaskTableH2 tt;
    tt.head_w = head_w;
    tt.head_b = head_b;
    tt.head_y = head_y;
    long long cost[kMaxTasksH2];
    long long tiles_all = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        if (tt.task[k].acts && tt.task[k].acts_stride == 0) tt.task[k].acts_stride = (long long)tasks[k].rows * D;
        cost[k] = ((long long)tasks[k].rows + 15) / 16 * (tasks[k].n_layers + (tasks[k].proj_w ? 5 : 0));
        tiles_all += ((long long)tasks[k].rows + 15) / 16;
    }
    tt.n = n;
    int grid = n_cus();
    int nw = 16;
    if (tiles_all <= (long long)grid * 16) nw = tiles_all <= (long long)grid * 4 ? 4 : 8;
    const long long max_grid = (tiles_all + nw - 1) / nw;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks_h2(cost, n, grid, tt.blk_end);
    mlp_fwd_h2_kernel<D><<<grid, nw * 64, 0, st>>>(tt);
    return launched("tspgnn_mlp_fwd_multi_h2");
}

// Working wavefronts per workgroup of a lock-step task (development switch TSPGNN_H2_LOCK_TILES; default: all of them --
// 8 / 6 / 4 measured 40.3 / 54.1 / 52.3 us per C2 launch against 37.9: the edge task misses the CUs more than the vertex
// chain gains from emptier matrix pipes).
static int h2_lock_tiles() {
    static const int v = [] {
        const char* e = getenv("TSPGNN_H2_LOCK_TILES");
        const int x = e ? atoi(e) : 0;
        return (x >= 1 && x <= 16) ? x : 16;
    }();
    return v;
}

template <int D>
static int launch_cell_h2(const tspgnn_cell_mlp_task* tasks, int n, hipStream_t st, const char* what) {
    const size_t head = (10 * D + 4) * sizeof(float);
    const size_t per_kb = (size_t)2 * 32 * 4 * D * 2;  // bytes of one k-block, two pieces
    const size_t budget = 160 * 1024 - head;
    const size_t layer_all = 2 * D * D * 2 + D * 4, proj_bytes = (size_t)2 * D * 4 * D * 2;
    CellTaskTableH2 tt;
    long long cost[kMaxTasksH2];
    long long tiles_all = 0;
    size_t lds_w = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        const tspgnn_lstm_task& c = tasks[k].cell;
        const int L = tasks[k].mlp_layers;
        const int KBT = (c.dx + D) / 32;
        const size_t k_bytes = (size_t)KBT * per_kb;
        size_t need;
        tt.together[k] = 0;
        tt.lock_tiles[k] = 0;
        if (k_bytes + L * layer_all <= budget && !tasks[k].proj_w) {
            tt.kbc[k] = KBT;
            tt.lockstep[k] = 0;
            need = k_bytes + L * layer_all;
        } else {
            const int kbc = k_bytes <= budget ? KBT : (int)(budget / per_kb);
            if (kbc < 1 || L * layer_all > budget || (tasks[k].proj_w && proj_bytes > budget))
                return fail(TSPGNN_EUNSUPPORTED, "%s: dx=%d, d=%d, %d MLP layers do not fit LDS", what, c.dx, D, L);
            tt.kbc[k] = kbc;
            tt.lockstep[k] = 1;
            need = (size_t)kbc * per_kb;
            size_t second = L * layer_all;
            if (tasks[k].proj_w) {
                if (L * layer_all + proj_bytes <= budget) {
                    tt.together[k] = 1;
                    second += proj_bytes;
                } else if (proj_bytes > second) {
                    second = proj_bytes;
                }
            }
            if (second > need) need = second;
        }
        if (need > lds_w) lds_w = need;
        const long long tiles = ((long long)c.rows + 15) / 16;
        cost[k] = tiles * (KBT * 4 + 2 * L + (tasks[k].proj_w ? 8 : 0) + 8);
        tiles_all += tiles;
    }
    tt.n = n;
    const size_t lds_bytes = lds_w + head;
    int grid = n_cus();
    const int nw_max = 12;
    const int nw = tiles_all <= (long long)grid * 4 ? 4 : (tiles_all <= (long long)grid * 8 ? 8 : nw_max);
    const long long max_grid = (tiles_all + nw - 1) / nw;
    if (grid > max_grid) grid = (int)max_grid;
    {
        // A lock-step task is a latency chain (two LDS stagings per round) that the resident tasks of the launch hide.
        // It gets the workgroups of a whole number of rounds -- ONE round while that stays within about twice its share
        // of the grid by cost (more workgroups would idle, fewer would double the chain: C2's 320 vertex tiles take 20
        // of 256), otherwise as many rounds as bring it back to its share (C4's 1 586 vertex tiles: 7 rounds on 15
        // workgroups instead of starving the edge task of 99); the resident tasks share the rest by cost.
        long long total_cost = 0;
        for (int k = 0; k < n; ++k) total_cost += cost[k] > 0 ? cost[k] : 1;
        int fixed[kMaxTasksH2], fixed_sum = 0, n_res = 0;
        long long res_cost[kMaxTasksH2];
        for (int k = 0; k < n; ++k) {
            fixed[k] = 0;
            if (tt.lockstep[k]) {
                const long long tiles = ((long long)tasks[k].cell.rows + 15) / 16;
                const int lw = nw < h2_lock_tiles() ? nw : h2_lock_tiles();
                tt.lock_tiles[k] = lw;
                const long long per_round = (tiles + lw - 1) / lw;
                long long share = (2 * cost[k] * grid + total_cost - 1) / total_cost;  // twice the proportional share
                if (share < 1) share = 1;
                const long long n_rounds = (per_round + share - 1) / share;
                fixed[k] = (int)((per_round + n_rounds - 1) / n_rounds);
            } else {
                ++n_res;
            }
            fixed_sum += fixed[k];
        }
        if (n_res == 0 || fixed_sum == 0 || fixed_sum > grid / 2) {
            grid = split_blocks_h2(cost, n, grid, tt.blk_end);
        } else {
            int res_end[kMaxTasksH2], j = 0;
            for (int k = 0; k < n; ++k)
                if (!fixed[k]) res_cost[j++] = cost[k];
            split_blocks_h2(res_cost, n_res, grid - fixed_sum, res_end);
            int used = 0;
            j = 0;
            for (int k = 0; k < n; ++k) {
                used += fixed[k] ? fixed[k] : res_end[j] - (j ? res_end[j - 1] : 0);
                if (!fixed[k]) ++j;
                tt.blk_end[k] = used;
            }
            grid = used;
        }
    }
    bool centered = true, loop_buffers = false;
    long long out_bytes = 0;   // h', c' of every task and its messages: what the launch stores (and the next ones re-read)
    for (int k = 0; k < n; ++k) {
        centered = centered && tasks[k].cell.z_centered != 0;
        out_bytes += (long long)tasks[k].cell.rows * D * 4 * (tasks[k].mlp_out ? 3 : 2);
        loop_buffers = loop_buffers || tasks[k].state_out_blocked != 0 || tasks[k].mlp_out != nullptr || tasks[k].proj_out != nullptr;
    }
    // write-through output stores while the loop's arrays (two copies of the states, two of the messages: ~2x out_bytes)
    // stay inside the 256 MB Infinity Cache; plain stores beyond (see st4o) -- and for a launch that writes nothing the next
    // step's launch reads from these buffers (no blocked state, no messages: the training forward's cell launch stores
    // into the tape, which the backward reads much later -- C2 training step 10.25-10.27 -> 10.12-10.16 ms)
    static const int wt_forced = [] {   // (development switch TSPGNN_H2_WT=0/1)
        const char* e = getenv("TSPGNN_H2_WT");
        return e ? atoi(e) : -1;
    }();
    const bool wt = wt_forced >= 0 ? wt_forced != 0 : (loop_buffers && 2 * out_bytes <= (long long)200 * 1024 * 1024);
    void (*fn)(const CellTaskTableH2) =
        centered ? (wt ? &lnlstm_mlp_fwd_h2_kernel<D, true, true> : &lnlstm_mlp_fwd_h2_kernel<D, true, false>)
                 : (wt ? &lnlstm_mlp_fwd_h2_kernel<D, false, true> : &lnlstm_mlp_fwd_h2_kernel<D, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%d B): %s", what, (int)lds_bytes, hipGetErrorString(e));
    fn<<<grid, nw * 64, lds_bytes, st>>>(tt);
    return launched(what);
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" float tspgnn_h2_weight_scale(void) { return kH2Scale; }


extern "C" int tspgnn_pack_weights_h2(const float* W, void* P, int krows, int ncols, unsigned* absmax_bits, void* stream) {
    TSPGNN_REQUIRE(krows >= 0 && krows % 32 == 0, "pack_weights_h2: krows=%d must be a multiple of 32", krows);
    TSPGNN_REQUIRE(ncols > 0 && ncols % 16 == 0, "pack_weights_h2: ncols=%d must be a multiple of 16", ncols);
    if (krows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(W && P, "pack_weights_h2: null pointer");
    int grid = (krows * ncols + 255) / 256;
    if (grid > 1024) grid = 1024;
    pack_weights_h2_kernel<<<grid, 256, 0, as_stream(stream)>>>(W, reinterpret_cast<_Float16*>(P), krows, ncols, absmax_bits);
    return launched("tspgnn_pack_weights_h2");
}

extern "C" int tspgnn_pack_mlp_h2(const float* wb, void* out, int d, int n_layers, int transposed, unsigned* absmax_bits,
                                  void* stream) {
    TSPGNN_REQUIRE(d > 0 && d % 32 == 0, "pack_mlp_h2: d=%d must be a positive multiple of 32", d);
    TSPGNN_REQUIRE(n_layers >= 0 && n_layers <= 64, "pack_mlp_h2: n_layers=%d", n_layers);
    if (n_layers == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(wb && out, "pack_mlp_h2: null pointer");
    int gx = (d * d + 255) / 256;
    if (gx > 64) gx = 64;
    pack_mlp_h2_kernel<<<dim3(gx, n_layers), 256, 0, as_stream(stream)>>>(wb, reinterpret_cast<unsigned char*>(out), d,
                                                                          transposed ? 1 : 0, absmax_bits);
    return launched("tspgnn_pack_mlp_h2");
}

extern "C" int tspgnn_mlp_fwd_multi_h2(const tspgnn_mlp_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasksH2, "mlp_fwd_multi_h2: 1..%d tasks", kMaxTasksH2);
    TSPGNN_REQUIRE(d == 32 || d == 64, "mlp_fwd_h2: d=%d must be 32 or 64", d);
    tspgnn_mlp_task live[kMaxTasksH2];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_mlp_task& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "mlp_fwd_h2: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 4, "mlp_fwd_h2: n_layers=%d must be in 1..4", t.n_layers);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE(t.X && t.wb && t.Y, "mlp_fwd_h2: null pointer");
        TSPGNN_REQUIRE(!t.proj_w || t.proj_out, "mlp_fwd_h2: projection needs proj_out");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    return d == 32 ? launch_mlp_h2<32>(live, n, as_stream(stream)) : launch_mlp_h2<64>(live, n, as_stream(stream));
}

extern "C" int tspgnn_mlp_head_fwd_h2(const tspgnn_mlp_task* task, const float* head_w, const float* head_b, float* y,
                                      int d, void* stream) {
    TSPGNN_REQUIRE(task, "mlp_head_fwd_h2: null task");
    TSPGNN_REQUIRE(d == 32 || d == 64, "mlp_head_fwd_h2: d=%d must be 32 or 64", d);
    const tspgnn_mlp_task& t = *task;
    TSPGNN_REQUIRE(t.rows >= 0, "mlp_head_fwd_h2: rows=%d", t.rows);
    TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 4, "mlp_head_fwd_h2: n_layers=%d must be in 1..4", t.n_layers);
    TSPGNN_REQUIRE(!t.proj_w, "mlp_head_fwd_h2: a head task has no projection");
    if (t.rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(t.X && t.wb && head_w && head_b && y, "mlp_head_fwd_h2: null pointer");
    return d == 32 ? launch_mlp_h2<32>(task, 1, as_stream(stream), head_w, head_b, y)
                   : launch_mlp_h2<64>(task, 1, as_stream(stream), head_w, head_b, y);
}

static int cell_mlp_h2(const tspgnn_cell_mlp_task* tasks, int n_tasks, int d, void* stream, const char* what) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasksH2, "%s: 1..%d tasks", what, kMaxTasksH2);
    TSPGNN_REQUIRE(d == 32 || d == 64, "%s: d=%d must be 32 or 64", what, d);
    tspgnn_cell_mlp_task live[kMaxTasksH2];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_lstm_task& t = tasks[k].cell;
        TSPGNN_REQUIRE(t.rows >= 0, "%s: rows=%d", what, t.rows);
        TSPGNN_REQUIRE((long long)t.rows * (4 * d > t.dx ? 4 * d : t.dx) < (1ll << 30), "%s: rows=%d too large for 32-bit offsets",
                       what, t.rows);
        TSPGNN_REQUIRE(t.dx >= 0 && t.dx % 32 == 0, "%s: dx=%d must be a non-negative multiple of 32", what, t.dx);
        TSPGNN_REQUIRE(tasks[k].mlp_layers >= 0 && tasks[k].mlp_layers <= 4, "%s: mlp_layers=%d must be in 0..4", what,
                       tasks[k].mlp_layers);
        if (t.rows == 0) continue;
        // (c == NULL: the zero cell state of a run's first step, nothing is read)
        TSPGNN_REQUIRE(t.h && t.K && t.ln && t.h_out && t.c_out && (t.dx == 0 || t.x), "%s: null pointer", what);
        // (h_out == h and c_out == c are fine: a tile reads its own rows of h and c, and only those, before it writes
        // them -- the in-place update keeps the states' footprint at one copy, inside the Infinity Cache)
        TSPGNN_REQUIRE(!t.uv || (t.dx == 0 && t.Zx), "%s: gather-init mode needs dx == 0 and Zx", what);
        TSPGNN_REQUIRE(!t.zbias || (t.zscale && !t.uv), "%s: zbias needs zscale and excludes gather-init mode", what);
        TSPGNN_REQUIRE(tasks[k].mlp_layers == 0 || tasks[k].mlp_wb, "%s: mlp_layers > 0 needs mlp_wb", what);
        TSPGNN_REQUIRE(!tasks[k].proj_w || (tasks[k].proj_out && tasks[k].mlp_layers > 0),
                       "%s: a projection needs proj_out and at least one MLP layer", what);
        TSPGNN_REQUIRE(tasks[k].mlp_acts_stride >= 0, "%s: mlp_acts_stride=%lld", what, tasks[k].mlp_acts_stride);
        live[n] = tasks[k];
        if (live[n].mlp_acts && live[n].mlp_acts_stride == 0) live[n].mlp_acts_stride = (long long)t.rows * d;
        ++n;
    }
    if (n == 0) return TSPGNN_OK;
    return d == 32 ? launch_cell_h2<32>(live, n, as_stream(stream), what) : launch_cell_h2<64>(live, n, as_stream(stream), what);
}

extern "C" int tspgnn_lnlstm_mlp_fwd_multi_h2(const tspgnn_cell_mlp_task* tasks, int n_tasks, int d, void* stream) {
    return cell_mlp_h2(tasks, n_tasks, d, stream, "tspgnn_lnlstm_mlp_fwd_multi_h2");
}

extern "C" int tspgnn_lnlstm_fwd_multi_h2(const tspgnn_lstm_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasksH2, "lnlstm_fwd_multi_h2: 1..%d tasks", kMaxTasksH2);
    tspgnn_cell_mlp_task wrapped[kMaxTasksH2];
    for (int k = 0; k < n_tasks; ++k) {
        wrapped[k] = tspgnn_cell_mlp_task{};
        wrapped[k].cell = tasks[k];
    }
    return cell_mlp_h2(wrapped, n_tasks, d, stream, "tspgnn_lnlstm_fwd_multi_h2");
}
