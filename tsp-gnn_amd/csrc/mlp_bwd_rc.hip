// Message-MLP backward on the fp16 matrix cores (f16x2).  Two kernels share one chain -- masks, then G_l = dpre_l W_l^T
// with every row normalised by a power of two and the second fp16 piece scaled into the normal range and accumulated apart
// (split2s / kblock_h2_side, as the cells' dh = dz Kh^T in dense_bwd_h2.hip):
//   mlp_bwd_h2_kernel   (tspgnn_mlp_bwd_multi_h2, the DEFAULT of the f16x2 and bf16 training steps): the TAPED backward --
//                       masks from the saved activations, dpre written for the weight-gradient reduction, several MLPs per
//                       launch, widths 64 and 128 -- replacing 3 x 256 (d = 64) v_mfma_f32_16x16x4_f32 per tile;
//   mlp_bwd_rcw_kernel  (tspgnn_mlp_bwd_rc_h2, opt-in): the hidden activations a_1 .. a_{L-1} are RECOMPUTED from the chain's
//                       input rows exactly as the f16x2 forward formed them (dense_layer_h2's arithmetic, same packed 2^s W,
//                       same bias block: the relu masks are the forward's own; a_L, the messages, is read), so the training
//                       forward tapes only the messages and can run the MLP inside the cell launch; the weight gradients
//                       a_l^T dpre_l are formed in the same launch (tspgnn_mlp_bwd_rc_task.partial).
// The taped form moves 205 MB per C2 step and is bound by them; the recomputing form is parity-green and deterministic but does
// not pay at C2 (81 us per launch; 11.6-11.7 against 11.1-11.2 ms per training step), hence opt-in (GraphNN.recompute_messages).
// A third form, which handed the recomputed a_l and dpre_l to wgrad_x3_kernel through chunk buffers, was removed in round 6.
#include "common.h"
#include "bf16_tile.h"
#include "h2_tile.h"
#include "mfma_tile.h"

namespace tspgnn {

// One Dense(64) layer of the f16x2 forward on NP row tiles (dense_layer_h2, dense_h2.hip: bias block in the accumulator,
// two k-blocks, relu, 2^-s) -- bit-identical per tile.
template <int NP>
__device__ __forceinline__ void dense_layer_h2_np(f32x4 (&out)[NP][4], const f32x4 (&in)[NP][4], const _Float16* wh,
                                                  const _Float16* wl, const float* bias, bool relu, int g, int rl) {
    f32x4 acc[NP][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4 b = ld4(bias + t * 16 + g * 4);
#pragma unroll
        for (int n = 0; n < NP; ++n) acc[n][t] = b;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f16x8 bh[NP], bl[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = in[n][2 * kb + (j >> 2)][j & 3];
            split2(x, bh[n], bl[n]);
        }
        kblock_h2_multi<4, 0, 4, NP>(acc, wh, wl, kb, g, rl, bh, bl);
    }
    const f32x2 inv = {kH2InvScale, kH2InvScale};
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][t][r] = fmaxf(acc[n][t][r], 0.f);
            }
            out[n][t].lo = acc[n][t].lo * inv;
            out[n][t].hi = acc[n][t].hi * inv;
        }
}

// ------------------------------------------------------------------------------------------------------------------
// The TAPED backward (dense_bwd.hip's mlp_bwd_kernel: masks from the saved activations, dpre written for the weight-gradient
// reduction) with its data gradient on the fp16 matrix cores instead of 256 (d = 64; 1024 at d = 128) v_mfma_f32_16x16x4_f32
// per layer and tile: tspgnn_mlp_bwd_multi_h2.  Same task structure (several MLPs per launch, gather-init mode, bf16
// tapes); wt = n_layers blocks tspgnn_pack_weights_h2(W_l^T).  d = 64 with up to four layers (64 KB of LDS), d = 128 with
// up to two (128 KB) -- the chunking of tspgnn_mlp_bwd_multi_f32.
constexpr int kMaxTasksBwdH2 = 4;
struct MlpBwdTaskTableH2 {
    tspgnn_mlp_bwd_task task[kMaxTasksBwdH2];
    int blk_end[kMaxTasksBwdH2];
    int n;
};

// ABF16: the saved activations (and Yout) are the bf16 arrays of a bf16-storage tape -- they only decide the relu masks.
template <int D, int MAXL, bool ABF16>
__global__ __launch_bounds__(1024) void mlp_bwd_h2_kernel(const MlpBwdTaskTableH2 tt) {
    constexpr int NT = D / 16, KB = D / 32;
    constexpr int WT_BYTES = 2 * D * D * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsw[];
    int* ticket = reinterpret_cast<int*>(ldsw + MAXL * WT_BYTES);
    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const float* __restrict__ dY = tt.task[k].dY;
    const float* __restrict__ acts = tt.task[k].acts;
    const long long acts_stride = tt.task[k].acts_stride;
    const float* __restrict__ Yout = tt.task[k].Yout;
    float* __restrict__ dpre = tt.task[k].dpre;
    const long long dpre_stride = tt.task[k].dpre_stride;
    float* __restrict__ dX = tt.task[k].dX;
    const int acc_dx = tt.task[k].accumulate_dx;
    const int rows = tt.task[k].rows, n_layers = tt.task[k].n_layers;
    const unsigned relu_mask = tt.task[k].relu_mask;
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tt.task[k].uv);
    const float* __restrict__ pre_X = tt.task[k].pre_X;
    const int pre_k = tt.task[k].pre_k;
    const int tiles_total = (rows + 15) / 16;
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4;
    h2_copy_to_lds(ldsw, tt.task[k].wt, n_layers * WT_BYTES, tid, blockDim.x);
    if (pre_X != nullptr) h2_copy_to_lds(ldsw + MAXL * WT_BYTES + 16, tt.task[k].pre_wt, pre_k * D * 4, tid, blockDim.x);
    const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
    const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
    if (tid == 0) *ticket = t_beg;
    h2_stage_wait();
    __syncthreads();
    for (;;) {
        int tile = 0;
        if (lane == 0) tile = atomicAdd(ticket, 1);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= t_end) break;
        const int row = tile * 16 + rl;
        const bool valid = row < rows;
        const unsigned rc = (unsigned)(valid ? row : rows - 1);
        const size_t rbase = (size_t)rc * D + g * 4;
        f32x4 gr[1][NT];
        if (pre_X != nullptr) {
            // dY = pre_X P^T first (tspgnn_mlp_bwd_task.pre_X): a gradient row of pre_k entries, normalised and split like
            // the layers' own (its largest entry found in a first pass over the row, the row re-read block by block: 128
            // registers do not hold it beside the accumulators); P^T resident in LDS behind the layers' weights
            if constexpr (D == 64) {
                const float* xr = pre_X + (size_t)rc * pre_k + g * 4;
                float m = 0.f;
                for (int i = 0; i * 16 < pre_k; ++i) {
                    const f32x4 v = ld4(xr + i * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) m = fmaxf(m, __builtin_fabsf(v[q]));
                }
                m = max_over_lane_groups16_swap(m);
                const int e = h2_row_exponent(m);
                const float up = __builtin_ldexpf(1.0f, -e), down = __builtin_ldexpf(kH2InvScale, e);
                f32x4 out[NT], side[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) out[t] = side[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const _Float16* ph = reinterpret_cast<const _Float16*>(ldsw + MAXL * WT_BYTES + 16);
                const _Float16* pl = ph + (size_t)pre_k * D;
                for (int kb = 0; kb * 32 < pre_k; ++kb) {
                    const f32x4 lo4 = ld4(xr + kb * 32), hi4 = ld4(xr + kb * 32 + 16);
                    float xv[8] = {lo4[0] * up, lo4[1] * up, lo4[2] * up, lo4[3] * up, hi4[0] * up, hi4[1] * up, hi4[2] * up, hi4[3] * up};
                    f16x8 bh, bm;
                    split2s(xv, bh, bm);
                    kblock_h2_side<NT>(out, side, ph, pl, kb, g, rl, bh, bm);
                }
                const float fold = 1.0f / 2048.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) gr[0][t][q] = fmaf(side[t][q], fold, out[t][q]) * down;
            }
        } else if (uv != nullptr) {
            const int2 ends = uv[rc];
            const float* pu = dY + (size_t)ends.x * D + g * 4;
            const float* pv = dY + (size_t)ends.y * D + g * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t) gr[0][t] = ld4(pu + t * 16) + ld4(pv + t * 16);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) gr[0][t] = ld4(dY + rbase + t * 16);
        }
        for (int l = n_layers - 1; l >= 0; --l) {
            if ((relu_mask >> l) & 1u) {
                if constexpr (ABF16) {
                    const __bf16* A = (l == n_layers - 1) ? reinterpret_cast<const __bf16*>(Yout)
                                                          : reinterpret_cast<const __bf16*>(acts) + (size_t)l * acts_stride;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f32x4 av = widen(ldw4(A + rbase + t * 16));
#pragma unroll
                        for (int q = 0; q < 4; ++q) gr[0][t][q] = av[q] > 0.f ? gr[0][t][q] : 0.f;
                    }
                } else {
                    const float* A = (l == n_layers - 1) ? Yout : acts + (size_t)l * acts_stride;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f32x4 av = ld4(A + rbase + t * 16);
#pragma unroll
                        for (int q = 0; q < 4; ++q) gr[0][t][q] = av[q] > 0.f ? gr[0][t][q] : 0.f;
                    }
                }
            }
            if (dpre != nullptr && valid) {
                float* dst = dpre + (size_t)l * dpre_stride + rbase;
#pragma unroll
                for (int t = 0; t < NT; ++t) st4(dst + t * 16, gr[0][t]);
            }
            // G = dpre_l W_l^T (packed 2^s W_l^T): the row normalised to [0.5, 1), scaled second piece apart
            const _Float16* wt = reinterpret_cast<const _Float16*>(ldsw + l * WT_BYTES);
            f32x4 out[1][NT], side[1][NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) out[0][t] = side[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) m = fmaxf(m, __builtin_fabsf(gr[0][t][q]));
            m = max_over_lane_groups16_swap(m);
            const int e = h2_row_exponent(m);
            const float up = __builtin_ldexpf(1.0f, -e), down = __builtin_ldexpf(kH2InvScale, e);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                f16x8 bh[1], bm[1];
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = gr[0][2 * kb + (j >> 2)][j & 3] * up;
                split2s(xv, bh[0], bm[0]);
                kblock_h2_side_multi<NT, 1>(out, side, wt, wt + D * D, kb, g, rl, bh, bm);
            }
            const float fold = 1.0f / 2048.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) gr[0][t][q] = fmaf(side[0][t][q], fold, out[0][t][q]) * down;
        }
        if (dX != nullptr && valid) {
            float* p = dX + rbase;
#pragma unroll
            for (int t = 0; t < NT; ++t) st4(p + t * 16, acc_dx ? ld4(p + t * 16) + gr[0][t] : gr[0][t]);
        }
    }
}

template <int D, int MAXL>
static int launch_mlp_bwd_h2(const tspgnn_mlp_bwd_task* tasks, int n, hipStream_t st) {
    MlpBwdTaskTableH2 tt;
    long long cost[kMaxTasksBwdH2], total = 0, tiles_all = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        if (tt.task[k].acts && tt.task[k].acts_stride == 0) tt.task[k].acts_stride = (long long)tasks[k].rows * D;
        if (tt.task[k].dpre && tt.task[k].dpre_stride == 0) tt.task[k].dpre_stride = (long long)tasks[k].rows * D;
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        cost[k] = tiles * (tasks[k].n_layers + (tasks[k].pre_X ? (tasks[k].pre_k + 63) / 64 : 0));
        total += cost[k];
        tiles_all += tiles;
    }
    tt.n = n;
    constexpr int threads = 1024;
    int grid = n_cus();
    const long long max_grid = (tiles_all + threads / 64 - 1) / (threads / 64);
    if (grid > max_grid) grid = (int)max_grid;
    if (grid < n) grid = n;
    // workgroups in proportion to tiles x layers, at least one each -- and never more than one wavefront per tile can use (a
    // small task with a long chain, e.g. the vertex side with its projection head: the surplus goes to the largest task)
    int bks[kMaxTasksBwdH2], used = 0, big = 0;
    for (int k = 0; k < n; ++k) {
        int bk = (int)((cost[k] * grid + total / 2) / (total > 0 ? total : 1));
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        const int useful = (int)((tiles + threads / 64 - 1) / (threads / 64));
        if (bk > useful) bk = useful;
        if (bk < 1) bk = 1;
        bks[k] = bk;
        used += bk;
        if (cost[k] > cost[big]) big = k;
    }
    if (used < grid) {
        const long long tiles = ((long long)tasks[big].rows + 15) / 16;
        const int useful = (int)((tiles + threads / 64 - 1) / (threads / 64));
        int add = grid - used;
        if (bks[big] + add > useful) add = useful > bks[big] ? useful - bks[big] : 0;
        bks[big] += add;
    }
    used = 0;
    for (int k = 0; k < n; ++k) {
        used += bks[k];
        tt.blk_end[k] = used;
    }
    int pre_bytes = 0;
    for (int k = 0; k < n; ++k)
        if (tasks[k].pre_X && tasks[k].pre_k * D * 4 > pre_bytes) pre_bytes = tasks[k].pre_k * D * 4;
    const int lds_bytes = MAXL * 2 * D * D * 2 + 16 + pre_bytes;
    const bool abf = tasks[0].acts_bf16 != 0;
    const void* fn = abf ? reinterpret_cast<const void*>(&mlp_bwd_h2_kernel<D, MAXL, true>)
                         : reinterpret_cast<const void*>(&mlp_bwd_h2_kernel<D, MAXL, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return fail((int)e, "mlp_bwd_h2: hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (abf) mlp_bwd_h2_kernel<D, MAXL, true><<<used, threads, lds_bytes, st>>>(tt);
    else mlp_bwd_h2_kernel<D, MAXL, false><<<used, threads, lds_bytes, st>>>(tt);
    return launched("tspgnn_mlp_bwd_multi_h2");
}

// ------------------------------------------------------------------------------------------------------------------
// The same chain with the WEIGHT GRADIENTS formed in the launch (tspgnn_mlp_bwd_rc_task.partial != NULL): nothing but the
// input rows, the messages, the incoming gradient and dX touches HBM.
//
// dW_l += a_l^T dpre_l has the ROWS as its contraction, while both operands live in the D layout (row = lane % 16) and the
// 3 x 64 x 64 accumulators do not fit one wavefront next to the chain.  So the eight wavefronts of a workgroup SHARE them:
// per round every wavefront runs the chain of one 16-row tile and, per layer, drops a_l and dpre_l as fp32 [16][64] tiles
// into its LDS slot; after a barrier every wavefront takes ALL slots of the round for the 16 x 32 block of dW_l it owns
// (two accumulator tiles per layer) on v_mfma_f32_16x16x4_f32 -- four rows per instruction, one ds_read_b32 per operand:
// lane (m, kg) of step s reads row 4s + kg, feature m of its block.  The 16-feature blocks of a row are rotated by
// row % 4 inside the slot, so the four lane groups of a read hit four different 64-byte segments (no bank conflict, no
// padding).  fp32 products -- no operand splitting at all; the data-gradient GEMM of the layer overlaps the products of
// the previous barrier interval.  db_l: per-lane partial sums over the wavefront's own rows, folded once at the end.
// Everything is static (tile of a wavefront, order of the slots, owner of an element): a dW element is summed in the same
// order in every run.
constexpr int kRcwWaves = 8;
constexpr int kRcwUnit = 2 * 16 * 64 * 4;   // {a_l, dpre_l} tiles of one wavefront

template <int L>
__global__ __launch_bounds__(kRcwWaves * 64) void mlp_bwd_rcw_kernel(const tspgnn_mlp_bwd_rc_task tk) {
    constexpr int D = 64;
    constexpr int LAYER_BYTES = 2 * D * D * 2 + D * 4;
    constexpr int WT_BYTES = 2 * D * D * 2;
    constexpr int WBYTES = (L - 1) * LAYER_BYTES + L * WT_BYTES;
    constexpr int PART = D * D + D;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* lds_f = ldsb;
    unsigned char* lds_t = ldsb + (L - 1) * LAYER_BYTES;
    unsigned char* slots = ldsb + WBYTES;
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;

    if (L > 1) h2_copy_to_lds(lds_f, tk.wb, (L - 1) * LAYER_BYTES, tid, blockDim.x);
    h2_copy_to_lds(lds_t, tk.wt, L * WT_BYTES, tid, blockDim.x);
    h2_stage_wait();
    __syncthreads();

    const float* __restrict__ X = tk.X;
    const float* __restrict__ Yout = tk.Yout;
    const float* __restrict__ dY = tk.dY;
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tk.uv);
    float* __restrict__ dX = tk.dX;
    const int rows = tk.rows;
    const unsigned relu_mask = tk.relu_mask;
    const int tiles_total = (rows + 15) / 16;
    const int t_beg = (int)((long long)tiles_total * blockIdx.x / gridDim.x);
    const int t_end = (int)((long long)tiles_total * (blockIdx.x + 1) / gridDim.x);
    const int n_t = t_end - t_beg;
    const int rounds = (n_t + kRcwWaves - 1) / kRcwWaves;

    // this wavefront's block of every dW_l: rows 16 ti .. + 15, columns 32 tjp .. + 31
    const int ti = wave >> 1, tjp = wave & 1;
    f32x4 acc[L][2], bsum[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        acc[l][0] = acc[l][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) bsum[l][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // writer: the float4 (features 16 t + 4 g ..) of row rl sits in block (t + rl % 4) % 4 of its row
    unsigned char* my_slot = slots + wave * kRcwUnit;
    const int wofs = rl * 256 + g * 16;
    // reader: lane (m = rl, kg = g), step s: row 4 s + kg (row % 4 = kg), block (t + kg) % 4
    const int rofs_a = g * 256 + (((ti + g) & 3) * 16 + rl) * 4;
    const int rofs_b0 = g * 256 + (((2 * tjp + g) & 3) * 16 + rl) * 4;
    const int rofs_b1 = g * 256 + (((2 * tjp + 1 + g) & 3) * 16 + rl) * 4;

    for (int r = 0; r < rounds; ++r) {
        const int base = t_beg + (int)((long long)n_t * r / rounds);
        const int cnt = t_beg + (int)((long long)n_t * (r + 1) / rounds) - base;   // tiles of this round (<= 8)
        const bool has = wave < cnt;
        const int row = (base + wave) * 16 + rl;
        const bool valid = has && row < rows;
        const unsigned rc = (unsigned)(valid ? row : rows - 1);
        const size_t rbase = (size_t)rc * D + g * 4;
        f32x4 a[L][1][4], gr[1][4];
        if (has) {
#pragma unroll
            for (int t = 0; t < 4; ++t) a[0][0][t] = ld4(X + rbase + t * 16);
            if (uv != nullptr) {
                const int2 ends = uv[rc];
                const float* pu = dY + (size_t)ends.x * D + g * 4;
                const float* pv = dY + (size_t)ends.y * D + g * 4;
#pragma unroll
                for (int t = 0; t < 4; ++t) gr[0][t] = ld4(pu + t * 16) + ld4(pv + t * 16);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) gr[0][t] = ld4(dY + rbase + t * 16);
            }
            if ((relu_mask >> (L - 1)) & 1u) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 y = ld4(Yout + rbase + t * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) gr[0][t][q] = y[q] > 0.f ? gr[0][t][q] : 0.f;
                }
            }
#pragma unroll
            for (int l = 0; l + 1 < L; ++l) {
                const _Float16* wf = reinterpret_cast<const _Float16*>(lds_f + l * LAYER_BYTES);
                const float* bf = reinterpret_cast<const float*>(lds_f + l * LAYER_BYTES + 2 * D * D * 2);
                dense_layer_h2_np<1>(a[l + 1], a[l], wf, wf + D * D, bf, (relu_mask >> l) & 1u, g, rl);
            }
        }
        if (!valid) {   // rows past the end contribute zeros
#pragma unroll
            for (int t = 0; t < 4; ++t) gr[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            if (has) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int o = wofs + (((t + rl) & 3) << 6);
                    *reinterpret_cast<f32x4*>(my_slot + o) = a[l][0][t];
                    *reinterpret_cast<f32x4*>(my_slot + 4096 + o) = gr[0][t];
                    bsum[l][t] += gr[0][t];
                }
            }
            __syncthreads();   // the round's tiles of layer l are in the slots
            for (int s = 0; s < cnt; ++s) {
                const unsigned char* u = slots + s * kRcwUnit;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float av = *reinterpret_cast<const float*>(u + rofs_a + ks * 1024);
                    const float b0 = *reinterpret_cast<const float*>(u + 4096 + rofs_b0 + ks * 1024);
                    const float b1 = *reinterpret_cast<const float*>(u + 4096 + rofs_b1 + ks * 1024);
                    acc[l][0] = MFMA16(av, b0, acc[l][0]);
                    acc[l][1] = MFMA16(av, b1, acc[l][1]);
                }
            }
            if (has) {
                // G_l = dpre_l W_l^T (packed 2^s W_l^T): the row normalised to [0.5, 1), scaled second piece apart
                const _Float16* wt = reinterpret_cast<const _Float16*>(lds_t + l * WT_BYTES);
                f32x4 out[1][4], side[1][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) out[0][t] = side[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                float m = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) m = fmaxf(m, __builtin_fabsf(gr[0][t][q]));
                m = max_over_lane_groups16_swap(m);
                const int e = h2_row_exponent(m);
                const float up = __builtin_ldexpf(1.0f, -e), down = __builtin_ldexpf(kH2InvScale, e);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f16x8 bh[1], bm[1];
                    float xv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = gr[0][2 * kb + (j >> 2)][j & 3] * up;
                    split2s(xv, bh[0], bm[0]);
                    kblock_h2_side_multi<4, 1>(out, side, wt, wt + D * D, kb, g, rl, bh, bm);
                }
                const float fold = 1.0f / 2048.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = fmaf(side[0][t][q], fold, out[0][t][q]) * down;
                        if (l > 0 && ((relu_mask >> (l - 1)) & 1u)) v = a[l][0][t][q] > 0.f ? v : 0.f;
                        gr[0][t][q] = valid ? v : 0.f;
                    }
            }
            __syncthreads();   // everyone has read the slots: they may be rewritten
        }
        if (dX != nullptr && valid) {
            float* p = dX + rbase;
#pragma unroll
            for (int t = 0; t < 4; ++t) st4(p + t * 16, tk.accumulate_dx ? ld4(p + t * 16) + gr[0][t] : gr[0][t]);
        }
    }
    // ---- the workgroup's partial.  dW: every element has one owner lane; launches of one backward pass are stream-ordered,
    // so the read-modify-write needs no atomics.  db: rows folded over the 16 lanes of a group (fixed tree), then over the
    // wavefronts in order.
    float* P = tk.partial + (size_t)blockIdx.x * (L * PART);
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* dst = P + l * PART + (16 * ti + 4 * g + q) * D + 16 * (2 * tjp + b) + rl;
                *dst += acc[l][b][q];
            }
    float* red = reinterpret_cast<float*>(slots);   // [wave][L][64]
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = bsum[l][t][q];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if (rl == 0) red[(wave * L + l) * D + 16 * t + 4 * g + q] = v;
            }
    __syncthreads();
    for (int i = tid; i < L * D; i += blockDim.x) {
        float v = 0.f;
        for (int w = 0; w < kRcwWaves; ++w) v += red[w * L * D + i];
        const int l = i / D, j = i % D;
        P[l * PART + D * D + j] += v;
    }
}

template <int L>
static int launch_mlp_bwd_rcw(const tspgnn_mlp_bwd_rc_task& tk, hipStream_t st) {
    constexpr int D = 64;
    const size_t lds_bytes = (size_t)(L - 1) * (2 * D * D * 2 + D * 4) + (size_t)L * 2 * D * D * 2 + (size_t)kRcwWaves * kRcwUnit;
    if (lds_bytes > 160 * 1024) return fail(TSPGNN_EUNSUPPORTED, "mlp_bwd_rc_h2: %d layers do not fit LDS", L);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_rcw_kernel<L>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "mlp_bwd_rc_h2: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const long long tiles = ((long long)tk.rows + 15) / 16;
    int grid = n_cus();
    const long long max_grid = (tiles + kRcwWaves - 1) / kRcwWaves;
    if (grid > max_grid) grid = (int)max_grid;
    mlp_bwd_rcw_kernel<L><<<grid, kRcwWaves * 64, lds_bytes, st>>>(tk);
    return launched("tspgnn_mlp_bwd_rc_h2");
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_mlp_bwd_rc_h2(const tspgnn_mlp_bwd_rc_task* task, int d, void* stream) {
    TSPGNN_REQUIRE(task != nullptr, "mlp_bwd_rc_h2: null task");
    TSPGNN_REQUIRE(d == 64, "mlp_bwd_rc_h2: d=%d must be 64", d);
    const tspgnn_mlp_bwd_rc_task& t = *task;
    TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 3, "mlp_bwd_rc_h2: n_layers=%d must be 1..3", t.n_layers);
    TSPGNN_REQUIRE(t.rows >= 0 && (long long)t.rows * d < (1ll << 31), "mlp_bwd_rc_h2: rows=%d", t.rows);
    if (t.rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(t.X && t.wt && t.dY && (t.n_layers == 1 || t.wb), "mlp_bwd_rc_h2: null pointer");
    TSPGNN_REQUIRE(t.Yout || !((t.relu_mask >> (t.n_layers - 1)) & 1u), "mlp_bwd_rc_h2: a relu on the last layer needs Yout");
    TSPGNN_REQUIRE(t.acts_stride >= 0 && t.dpre_stride >= 0, "mlp_bwd_rc_h2: negative stride");
    hipStream_t st = as_stream(stream);
    // (round 6: the form that handed the recomputed activations and pre-activation gradients to tspgnn_wgrad was removed --
    // as byte-bound as the tape it replaced, DESIGN_HISTORY round 5; what stays forms the weight gradients in the launch)
    TSPGNN_REQUIRE(t.partial != nullptr && !t.acts && !t.dpre,
                   "mlp_bwd_rc_h2: weight gradients are formed in the launch (partial != NULL; acts / dpre must be NULL)");
    switch (t.n_layers) {
        case 1: return launch_mlp_bwd_rcw<1>(t, st);
        case 2: return launch_mlp_bwd_rcw<2>(t, st);
        default: return launch_mlp_bwd_rcw<3>(t, st);
    }
}

extern "C" long long tspgnn_mlp_bwd_rc_partial_floats(int d, int n_layers) {
    return (long long)n_cus() * n_layers * ((long long)d * d + d);
}

extern "C" int tspgnn_mlp_bwd_rc_finish_f32(const float* partial, float* grad_wb, int d, int n_layers, void* stream) {
    TSPGNN_REQUIRE(partial && grad_wb, "mlp_bwd_rc_finish: null pointer");
    TSPGNN_REQUIRE(d == 64 && n_layers >= 1 && n_layers <= 3, "mlp_bwd_rc_finish: d=%d, n_layers=%d", d, n_layers);
    const int n = n_layers * (d * d + d);
    reduce_partials(partial, n_cus(), n, grad_wb, n, 1.0f, 1, as_stream(stream));
    return launched("tspgnn_mlp_bwd_rc_finish_f32");
}

extern "C" int tspgnn_mlp_bwd_multi_h2(const tspgnn_mlp_bwd_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasksBwdH2, "mlp_bwd_multi_h2: 1..%d tasks", kMaxTasksBwdH2);
    TSPGNN_REQUIRE(d == 64 || d == 128, "mlp_bwd_h2: d=%d must be 64 or 128", d);
    tspgnn_mlp_bwd_task live[kMaxTasksBwdH2];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_mlp_bwd_task& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0 && (long long)t.rows * d < (1ll << 31), "mlp_bwd_h2: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 4, "mlp_bwd_h2: n_layers=%d must be in 1..4", t.n_layers);
        if (d == 128 && t.n_layers > 2)
            return fail(TSPGNN_EUNSUPPORTED, "mlp_bwd_h2: d=128 holds at most 2 layers in LDS (got %d)", t.n_layers);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE((t.dY || t.pre_X) && t.wt, "mlp_bwd_h2: null pointer");
        TSPGNN_REQUIRE(!t.pre_X || (d == 64 && t.pre_wt && !t.uv && t.pre_k >= 32 && t.pre_k <= 256 && t.pre_k % 32 == 0),
                       "mlp_bwd_h2: pre_X needs d == 64, pre_wt, no uv and pre_k in 32..256 (a multiple of 32), got %d", t.pre_k);
        const unsigned inner = t.relu_mask & ((1u << (t.n_layers - 1)) - 1u);
        TSPGNN_REQUIRE(!inner || t.acts, "mlp_bwd_h2: relu layers need the saved activations");
        TSPGNN_REQUIRE(!((t.relu_mask >> (t.n_layers - 1)) & 1u) || t.Yout, "mlp_bwd_h2: relu on the last layer needs Yout");
        TSPGNN_REQUIRE(n == 0 || (t.acts_bf16 != 0) == (live[0].acts_bf16 != 0), "mlp_bwd_h2: the tasks of a launch share acts_bf16");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    return d == 64 ? launch_mlp_bwd_h2<64, 4>(live, n, as_stream(stream)) : launch_mlp_bwd_h2<128, 2>(live, n, as_stream(stream));
}
