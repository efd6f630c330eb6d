// Message-MLP backward that RECOMPUTES its hidden activations (tspgnn_mlp_bwd_rc_h2; the training step's byte diet,
// DESIGN 4).
//
// The taped form (dense_bwd.hip: mlp_bwd_kernel) made the training FORWARD write the L-1 hidden activations of every
// edge row and step (77 MB per C2 step for the three pushed layers of E_msg_V) -- which kept the message MLP out of the
// cell launch (the fused launch is write-bound with them) -- and ran its data gradient on the fp32 matrix instruction
// (3 x 256 v_mfma_f32_16x16x4_f32 per 16-row tile: the launch was MFMA-bound at 49 us).  Here the backward launch
//   * recomputes a_1 .. a_{L-1} from the chain's input rows exactly as the f16x2 forward formed them (dense_layer_h2's
//     arithmetic, same packed 2^s W, same bias block), so the relu masks are the forward's own; the chain's output a_L
//     (the messages: they exist anyway) is read for the last mask;
//   * runs the data gradient G_l = dpre_l W_l^T on the fp16 matrix cores like the cells' dh = dz Kh^T
//     (dense_bwd_h2.hip): row normalised by a power of two, second piece scaled into fp16's normal range and accumulated
//     apart (split2s / kblock_h2_side);
//   * hands a_1 .. a_{L-1} and dpre_0 .. dpre_{L-1} to the weight-gradient reduction (wgrad_x3_kernel) through chunk
//     buffers of the BACKWARD pass -- the forward tapes only the messages.
// With tspgnn_mlp_bwd_rc_task.partial the weight gradients are formed in the launch as well (mlp_bwd_rcw_kernel below).
// Both forms are parity-green; neither pays at C2 (DESIGN_HISTORY, round 5), so the training step keeps the taped form
// by default and these are opt-in (GraphNN.recompute_messages).
#include "common.h"
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

struct MlpBwdRcTable {
    tspgnn_mlp_bwd_rc_task task;
};

template <int L>
__global__ __launch_bounds__(1024) void mlp_bwd_rc_kernel(const tspgnn_mlp_bwd_rc_task tk) {
    constexpr int D = 64;
    constexpr int LAYER_BYTES = 2 * D * D * 2 + D * 4;   // forward block: two fp16 pieces + the bias (2^s b)
    constexpr int WT_BYTES = 2 * D * D * 2;
    constexpr int WBYTES = (L - 1) * LAYER_BYTES + L * WT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[WBYTES + 16];
    unsigned char* lds_f = lds;
    unsigned char* lds_t = lds + (L - 1) * LAYER_BYTES;
    int* ticket = reinterpret_cast<int*>(lds + WBYTES);
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4;

    const float* __restrict__ X = tk.X;
    const float* __restrict__ Yout = tk.Yout;
    const float* __restrict__ dY = tk.dY;
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tk.uv);
    float* __restrict__ dX = tk.dX;
    float* __restrict__ acts = tk.acts;
    float* __restrict__ dpre = tk.dpre;
    const int rows = tk.rows;
    const unsigned relu_mask = tk.relu_mask;
    const int tiles_total = (rows + 15) / 16;

    if (L > 1) h2_copy_to_lds(lds_f, tk.wb, (L - 1) * LAYER_BYTES, tid, blockDim.x);
    h2_copy_to_lds(lds_t, tk.wt, L * WT_BYTES, tid, blockDim.x);
    const int t_beg = (int)((long long)tiles_total * blockIdx.x / gridDim.x);
    const int t_end = (int)((long long)tiles_total * (blockIdx.x + 1) / gridDim.x);
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
        f32x4 a[L][1][4];     // a[l] = input of layer l (a[0] = the chain's input rows)
        f32x4 gr[1][4];       // the gradient travelling down the chain
#pragma unroll
        for (int t = 0; t < 4; ++t) a[0][0][t] = ld4(X + rbase + t * 16);
        if (uv != nullptr) {   // the adjoint of the V<-E row-sum, formed on the fly
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
                for (int r = 0; r < 4; ++r) gr[0][t][r] = y[r] > 0.f ? gr[0][t][r] : 0.f;
            }
        }
        // ---- forward again: a_1 .. a_{L-1}, handed to the weight-gradient reduction
#pragma unroll
        for (int l = 0; l + 1 < L; ++l) {
            const _Float16* wf = reinterpret_cast<const _Float16*>(lds_f + l * LAYER_BYTES);
            const float* bf = reinterpret_cast<const float*>(lds_f + l * LAYER_BYTES + 2 * D * D * 2);
            dense_layer_h2_np<1>(a[l + 1], a[l], wf, wf + D * D, bf, (relu_mask >> l) & 1u, g, rl);
            if (acts != nullptr && valid) {
                float* dst = acts + (size_t)l * tk.acts_stride + rbase;
#pragma unroll
                for (int t = 0; t < 4; ++t) st4(dst + t * 16, a[l + 1][0][t]);
            }
        }
        // ---- backward: gr = dpre_l on entry of layer l
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            if (dpre != nullptr && valid) {
                float* dst = dpre + (size_t)l * tk.dpre_stride + rbase;
#pragma unroll
                for (int t = 0; t < 4; ++t) st4(dst + t * 16, gr[0][t]);
            }
            // G_l = dpre_l W_l^T (packed 2^s W_l^T): the row normalised to [0.5, 1), scaled second piece apart
            const _Float16* wt = reinterpret_cast<const _Float16*>(lds_t + l * WT_BYTES);
            f32x4 out[1][4], side[1][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) out[0][t] = side[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, __builtin_fabsf(gr[0][t][r]));
            m = max_over_lane_groups16_swap(m);
            const int e = m > 0.f ? __builtin_amdgcn_frexp_expf(m) : 0;
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
                for (int r = 0; r < 4; ++r) {
                    float v = fmaf(side[0][t][r], fold, out[0][t][r]) * down;
                    if (l > 0 && ((relu_mask >> (l - 1)) & 1u)) v = a[l][0][t][r] > 0.f ? v : 0.f;
                    gr[0][t][r] = v;
                }
        }
        if (dX != nullptr && valid) {
            float* p = dX + rbase;
#pragma unroll
            for (int t = 0; t < 4; ++t) st4(p + t * 16, tk.accumulate_dx ? ld4(p + t * 16) + gr[0][t] : gr[0][t]);
        }
    }
}

template <int L>
static int launch_mlp_bwd_rc(const tspgnn_mlp_bwd_rc_task& tk, hipStream_t st) {
    const long long tiles = ((long long)tk.rows + 15) / 16;
    int grid = n_cus();
    const long long max_grid = (tiles + 15) / 16;
    if (grid > max_grid) grid = (int)max_grid;
    mlp_bwd_rc_kernel<L><<<grid, 1024, 0, st>>>(tk);
    return launched("tspgnn_mlp_bwd_rc_h2");
}

// ------------------------------------------------------------------------------------------------------------------
// The same chain with the WEIGHT GRADIENTS formed in the launch (tspgnn_mlp_bwd_rc_task.partial != NULL): nothing but the
// input rows, the messages, the incoming gradient and dX touches HBM.
//
// dW_l += a_l^T dpre_l has the ROWS as its contraction, while both operands live in the D layout (row = lane % 16), and a
// wavefront that kept all 3 x 64 x 64 accumulators would be alone on its SIMD (built and measured: 112 us, every latency of
// the chain exposed).  So the workgroup is SPECIALISED: twelve producer wavefronts run the chain of one 16-row tile each
// (recompute, masks, data gradient -- the code of mlp_bwd_rc_kernel) and, per layer, drop a_l and dpre_l as three bf16
// pieces into an LDS slot, [16 rows][16 features] blocks; four consumer wavefronts own one 16-row block of every dW_l
// (48 + 3 accumulator registers) and take the slots of two producers at a time -- 32 contraction indices, one
// v_mfma_f32_16x16x32_bf16 per term, six terms per product as in wgrad_x3_kernel -- through ds_read_b64_tr_b16, which hands
// a lane the four rows of its feature.  Everything is STATIC: producer p takes tiles t_beg + 12 r + p, the slot of
// producers p and p + 6 is p % 6, consumers walk (round, layer, half, pair) in one fixed order, so a dW element is summed
// in the same order in every run.  Two monotone counters per slot: `full` (units written) and `done` (consumer passes
// finished, four per unit); a producer writes unit n when done == 4 n, a consumer reads it when full > n.
typedef __bf16 bf16x8_r __attribute__((ext_vector_type(8)));
typedef short s16x4_r __attribute__((ext_vector_type(4)));
typedef short s16x8_r __attribute__((ext_vector_type(8)));
#define MFMA_BF16_R(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// Three bf16 pieces of four values, each piece packed as 4 x 16 bit (the 8 bytes a lane stores per block).
__device__ __forceinline__ void split3_pack(const f32x4& x, uint2& hi, uint2& mid, uint2& lo) {
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 bh = (__bf16)x[i];
        const float r1 = x[i] - (float)bh;
        const __bf16 bm = (__bf16)r1;
        const __bf16 bl = (__bf16)(r1 - (float)bm);
        h[i] = __builtin_bit_cast(unsigned short, bh);
        m[i] = __builtin_bit_cast(unsigned short, bm);
        l[i] = __builtin_bit_cast(unsigned short, bl);
    }
    hi = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    mid = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
    lo = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
}

// The MFMA operand of one 16-feature block over the 32 rows of two tiles: lane (f, kg) gets rows 4kg .. 4kg+3 of tile P
// (contraction indices 8kg .. 8kg+3) and of tile Q (8kg+4 .. 8kg+7) at feature f.  `blk_p` / `blk_q`: the two
// [16 rows][16 features] bf16 blocks (512 B, row-major); every lane passes the address of ITS 8-byte chunk (block +
// lane * 8 = row lane / 4, features 4 (lane % 4) ..) and the transposing read hands the 4 x 16 sub-block of a 16-lane group
// back column by column.
__device__ __forceinline__ bf16x8_r tr_operand(const unsigned char* blk_p, const unsigned char* blk_q, int lane) {
    typedef __attribute__((address_space(3))) s16x4_r* lds_p;
    const s16x4_r p = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(blk_p + lane * 8));
    const s16x4_r q = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(blk_q + lane * 8));
    const s16x8_r v = __builtin_shufflevector(p, q, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_r, v);
}

constexpr int kRcProd = 12, kRcCons = 4, kRcSlots = 6;
constexpr int kRcSlotBytes = 2 * 3 * 4 * 512;   // {dpre, a_l} x three pieces x four feature blocks x [16][16] bf16

__device__ __forceinline__ void rc_wait_at_least(const int* counter, int target) {
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}

template <int L>
__global__ __launch_bounds__(1024) void mlp_bwd_rcw_kernel(const tspgnn_mlp_bwd_rc_task tk) {
    constexpr int D = 64;
    constexpr int LAYER_BYTES = 2 * D * D * 2 + D * 4;
    constexpr int WT_BYTES = 2 * D * D * 2;
    constexpr int WBYTES = (L - 1) * LAYER_BYTES + L * WT_BYTES;
    constexpr int PART = D * D + D;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* lds_f = ldsb;
    unsigned char* lds_t = ldsb + (L - 1) * LAYER_BYTES;
    unsigned char* slots = ldsb + WBYTES;
    int* full = reinterpret_cast<int*>(slots + kRcSlots * kRcSlotBytes);
    int* done = full + 8;
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;

    if (L > 1) h2_copy_to_lds(lds_f, tk.wb, (L - 1) * LAYER_BYTES, tid, blockDim.x);
    h2_copy_to_lds(lds_t, tk.wt, L * WT_BYTES, tid, blockDim.x);
    // the slots start as zeros: a unit without a tile contributes 0 x (finite leftovers)
    for (int i = tid; i < (kRcSlots * kRcSlotBytes + 64) / 4; i += blockDim.x) reinterpret_cast<int*>(slots)[i] = 0;
    h2_stage_wait();
    __syncthreads();

    const int rows = tk.rows;
    const unsigned relu_mask = tk.relu_mask;
    const int tiles_total = (rows + 15) / 16;
    const int t_beg = (int)((long long)tiles_total * blockIdx.x / gridDim.x);
    const int t_end = (int)((long long)tiles_total * (blockIdx.x + 1) / gridDim.x);
    const int rounds = (t_end - t_beg + kRcProd - 1) / kRcProd;

    if (wave < kRcProd) {
        // ------------------------------------------------------------------ producer
        const float* __restrict__ X = tk.X;
        const float* __restrict__ Yout = tk.Yout;
        const float* __restrict__ dY = tk.dY;
        const int2* __restrict__ uv = reinterpret_cast<const int2*>(tk.uv);
        float* __restrict__ dX = tk.dX;
        const int half = wave / kRcSlots, slot = wave % kRcSlots;
        unsigned char* sg = slots + slot * kRcSlotBytes;        // dpre pieces: (piece * 4 + tj) * 512
        unsigned char* sa = sg + 12 * 512;                      // a_l pieces:  (piece * 4 + ti) * 512
        const int lofs = rl * 32 + g * 8;                       // this lane's 8 bytes of a [16][16] block
        for (int r = 0; r < rounds; ++r) {
            const int tile = t_beg + r * kRcProd + wave;
            const bool has = tile < t_end;
            const int row = tile * 16 + rl;
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
            if (!valid) {   // rows past the end (and wavefronts without a tile) contribute zeros
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[l][0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) gr[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                const int n = (r * L + (L - 1 - l)) * 2 + half;      // this unit's number in the slot's sequence
                rc_wait_at_least(done + slot, 4 * n);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint2 ph, pm, pl;
                    split3_pack(gr[0][t], ph, pm, pl);
                    *reinterpret_cast<uint2*>(sg + t * 512 + lofs) = ph;
                    *reinterpret_cast<uint2*>(sg + (4 + t) * 512 + lofs) = pm;
                    *reinterpret_cast<uint2*>(sg + (8 + t) * 512 + lofs) = pl;
                    split3_pack(a[l][0][t], ph, pm, pl);
                    *reinterpret_cast<uint2*>(sa + t * 512 + lofs) = ph;
                    *reinterpret_cast<uint2*>(sa + (4 + t) * 512 + lofs) = pm;
                    *reinterpret_cast<uint2*>(sa + (8 + t) * 512 + lofs) = pl;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the blocks are in LDS before the counter moves
                if (lane == 0) __hip_atomic_store(full + slot, n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!has) continue;
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
                const int e = m > 0.f ? __builtin_amdgcn_frexp_expf(m) : 0;
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
                        gr[0][t][q] = v;
                    }
            }
            if (dX != nullptr && valid) {
                float* p = dX + rbase;
#pragma unroll
                for (int t = 0; t < 4; ++t) st4(p + t * 16, tk.accumulate_dx ? ld4(p + t * 16) + gr[0][t] : gr[0][t]);
            }
        }
    } else if (wave < kRcProd + kRcCons) {
        // ------------------------------------------------------------------ consumer: rows 16 c .. 16 c + 15 of every dW_l,
        // and the bias gradient of features 16 c .. 16 c + 15
        const int c = wave - kRcProd;
        f32x4 acc[L][4], accb[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            accb[l] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[l][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bf16x8_r ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                for (int half = 0; half < 2; ++half) {
                    const int n = (r * L + (L - 1 - l)) * 2 + half;
                    for (int q = 0; q < kRcSlots / 2; ++q) {
                        const int s0 = 2 * q, s1 = 2 * q + 1;
                        rc_wait_at_least(full + s0, n + 1);
                        rc_wait_at_least(full + s1, n + 1);
                        const unsigned char* g0 = slots + s0 * kRcSlotBytes;
                        const unsigned char* g1 = slots + s1 * kRcSlotBytes;
                        const unsigned char* a0 = g0 + (12 + c) * 512;
                        const unsigned char* a1 = g1 + (12 + c) * 512;
                        const bf16x8_r ah = tr_operand(a0, a1, lane);
                        const bf16x8_r am = tr_operand(a0 + 4 * 512, a1 + 4 * 512, lane);
                        const bf16x8_r al = tr_operand(a0 + 8 * 512, a1 + 8 * 512, lane);
                        bf16x8_r bh[4], bm[4], bl[4];
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) {
                            bh[tj] = tr_operand(g0 + tj * 512, g1 + tj * 512, lane);
                            bm[tj] = tr_operand(g0 + (4 + tj) * 512, g1 + (4 + tj) * 512, lane);
                            bl[tj] = tr_operand(g0 + (8 + tj) * 512, g1 + (8 + tj) * 512, lane);
                        }
                        // (the operands are in registers: the slots may be refilled)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) {
                            __hip_atomic_fetch_add(done + s0, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(done + s1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        // the bias gradient's feature block (c is not a compile-time index: selected, not indexed)
                        const bf16x8_r ch = c == 0 ? bh[0] : c == 1 ? bh[1] : c == 2 ? bh[2] : bh[3];
                        const bf16x8_r cm = c == 0 ? bm[0] : c == 1 ? bm[1] : c == 2 ? bm[2] : bm[3];
                        const bf16x8_r cl = c == 0 ? bl[0] : c == 1 ? bl[1] : c == 2 ? bl[2] : bl[3];
                        // six terms per block, smallest first, term-major over the four blocks (independent chains)
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) acc[l][tj] = MFMA_BF16_R(al, bh[tj], acc[l][tj]);
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) acc[l][tj] = MFMA_BF16_R(am, bm[tj], acc[l][tj]);
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) acc[l][tj] = MFMA_BF16_R(ah, bl[tj], acc[l][tj]);
                        accb[l] = MFMA_BF16_R(ones, cl, accb[l]);
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) acc[l][tj] = MFMA_BF16_R(am, bh[tj], acc[l][tj]);
                        accb[l] = MFMA_BF16_R(ones, cm, accb[l]);
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) acc[l][tj] = MFMA_BF16_R(ah, bm[tj], acc[l][tj]);
                        accb[l] = MFMA_BF16_R(ones, ch, accb[l]);
#pragma unroll
                        for (int tj = 0; tj < 4; ++tj) acc[l][tj] = MFMA_BF16_R(ah, bh[tj], acc[l][tj]);
                    }
                }
            }
        }
        // the workgroup's partial: every element has exactly one owner lane; launches of one backward pass are
        // stream-ordered, so the read-modify-write needs no atomics
        float* P = tk.partial + (size_t)blockIdx.x * (L * PART);
#pragma unroll
        for (int l = 0; l < L; ++l) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* dst = P + l * PART + (16 * c + 4 * g + q) * D + 16 * tj + rl;
                    *dst += acc[l][tj][q];
                }
            if (g == 0) {   // (every row of the ones-product holds the column sums)
                float* dst = P + l * PART + D * D + 16 * c + rl;
                *dst += accb[l][0];
            }
        }
    }
}

template <int L>
static int launch_mlp_bwd_rcw(const tspgnn_mlp_bwd_rc_task& tk, hipStream_t st) {
    constexpr int D = 64;
    const size_t lds_bytes = (size_t)(L - 1) * (2 * D * D * 2 + D * 4) + (size_t)L * 2 * D * D * 2 + (size_t)kRcSlots * kRcSlotBytes + 64;
    if (lds_bytes > 160 * 1024) return fail(TSPGNN_EUNSUPPORTED, "mlp_bwd_rc_h2: %d layers do not fit LDS", L);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_rcw_kernel<L>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "mlp_bwd_rc_h2: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const long long tiles = ((long long)tk.rows + 15) / 16;
    int grid = n_cus();
    const long long max_grid = (tiles + kRcProd - 1) / kRcProd;
    if (grid > max_grid) grid = (int)max_grid;
    mlp_bwd_rcw_kernel<L><<<grid, 1024, lds_bytes, st>>>(tk);
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
    if (t.partial != nullptr) {   // weight gradients formed in the launch
        TSPGNN_REQUIRE(!t.acts && !t.dpre, "mlp_bwd_rc_h2: partial excludes acts / dpre");
        switch (t.n_layers) {
            case 1: return launch_mlp_bwd_rcw<1>(t, st);
            case 2: return launch_mlp_bwd_rcw<2>(t, st);
            default: return launch_mlp_bwd_rcw<3>(t, st);
        }
    }
    switch (t.n_layers) {
        case 1: return launch_mlp_bwd_rc<1>(t, st);
        case 2: return launch_mlp_bwd_rc<2>(t, st);
        default: return launch_mlp_bwd_rc<3>(t, st);
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
