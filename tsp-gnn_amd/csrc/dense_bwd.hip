// Backward of the dense per-row updates (tf.gradients through graphnn.py:142-173, model.py:166) on the
// gfx950 matrix cores, exact fp32, same "transposed chaining" layout as dense.hip:
//   * linear      : Y = X W with W resident in LDS (used for dz K^T -> (dx, dh) of the LSTM cell)
//   * lnlstm_bwd  : recomputes z = [x,h]K, then LayerNorm / gate backward -> dz, dc, LN-parameter grads
//   * mlp_bwd     : chained data gradient of a square Dense stack, emitting every layer's d(pre-activation)
//   * wgrad       : dW += X^T dY, db += colsum(dY): rows are the contraction, split over wavefronts with
//                   a deterministic two-stage reduction (per-chunk partials, then one pass over chunks)
#include "common.h"
#include "mfma_tile.h"
#include "bf16_tile.h"
#include "lstm_bwd_tile.h"

namespace tspgnn {

// ------------------------------------------------------------------------------------ linear
// Y[rows, NT*16] = X[rows, kin] * W (packed [kin, NT*16]).  Columns [0,n1) go to Y1, the rest to Y2
// (optionally accumulated).  qc = 16-row blocks of W per LDS chunk: qc >= kin/16 keeps W resident and
// lets wavefronts pull tiles through a ticket; otherwise the workgroup walks W chunk by chunk in
// lock step, one tile per wavefront per round.
template <int NT>
__global__ __launch_bounds__(512) void linear_kernel(const float* __restrict__ X, int kin,
                                                     const float* __restrict__ Wp, float* __restrict__ Y1, int n1,
                                                     float* __restrict__ Y2, int n2, int acc2, int rows,
                                                     int tiles_total, int qc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int QT = kin >> 4;
    const bool resident = qc >= QT;
    float* lds_w = lds;
    int* ticket = reinterpret_cast<int*>(lds + (size_t)(resident ? QT : qc) * 16 * NT * 16);
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int nw = blockDim.x >> 6;

    auto store = [&](f32x4 (&acc)[NT], size_t rc) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16;
            if (col < n1) {
                st4(Y1 + rc * n1 + col + g * 4, acc[t]);
            } else {
                float* p = Y2 + rc * n2 + (col - n1) + g * 4;
                st4(p, acc2 ? ld4(p) + acc[t] : acc[t]);
            }
        }
    };

    if (resident) {
        copy_to_lds(lds_w, Wp, kin * NT * 16, tid, blockDim.x);
        const int t_beg = (int)((long long)tiles_total * blockIdx.x / gridDim.x);
        const int t_end = (int)((long long)tiles_total * (blockIdx.x + 1) / gridDim.x);
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
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* xr = X + rc * kin + g * 4;
            gemm_kloop<NT>(acc, lds_w, 0, 0, QT, xr, xr, QT, g, rl);
            if (valid) store(acc, rc);
        }
    } else {
        const int rounds = (tiles_total + nw - 1) / nw;
        for (int r = blockIdx.x; r < rounds; r += gridDim.x) {
            const int tile = r * nw + wave;
            const bool live = tile < tiles_total;
            const int row = tile * 16 + rl;
            const bool valid = live && row < rows;
            const size_t rc = (size_t)(valid ? row : rows - 1);
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* xr = X + rc * kin + g * 4;
            for (int q0 = 0; q0 < QT; q0 += qc) {
                const int q1 = min(QT, q0 + qc);
                __syncthreads();
                copy_to_lds(lds_w, Wp + (size_t)q0 * 16 * NT * 16, (q1 - q0) * 16 * NT * 16, tid, blockDim.x);
                __syncthreads();
                if (live) gemm_kloop<NT>(acc, lds_w, q0, q0, q1, xr, xr, QT, g, rl);
            }
            if (valid) store(acc, rc);
        }
    }
}

// Few rows (the vertex side: a few hundred 16-row tiles on 256 CUs): one wavefront per tile would run the whole
// kin/4 * NT chain of MFMAs alone (512 at kin=256, NT=8: ~8 us) with most of the chip idle.  Here the NT output tiles
// of a row tile are split over four wavefronts (NT/4 each), two row tiles per workgroup; W resident in LDS.
template <int NT, int PART>
__device__ __forceinline__ void linear_split_part(const float* lds_w, const float* xr, int QT, int g, int rl,
                                                  f32x4 (&acc)[NT / 4]) {
    constexpr int TW = NT / 4, U = (PART * TW) / 4, C0 = (PART * TW) % 4;
    for (int q0 = 0; q0 < QT; q0 += 8) {
        f32x4 xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = ld4(xr + (q0 + i < QT ? q0 + i : QT - 1) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (q0 + i < QT) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 aw = ld4(lds_w + frag_off<NT>((q0 + i) * 4 + p, g, rl) + U * 64);
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[t] = MFMA16(aw[C0 + t], xv[i][p], acc[t]);
                }
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(512) void linear_split_kernel(const float* __restrict__ X, int kin,
                                                           const float* __restrict__ Wp, float* __restrict__ Y1, int n1,
                                                           float* __restrict__ Y2, int n2, int acc2, int rows,
                                                           int tiles_total) {
    static_assert(NT % 4 == 0, "four column parts");
    constexpr int TW = NT / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int QT = kin >> 4;
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    copy_to_lds(lds, Wp, kin * NT * 16, tid, blockDim.x);
    __syncthreads();
    const int tile = blockIdx.x * 2 + (wave >> 2), part = wave & 3;
    if (tile >= tiles_total) return;  // wave-uniform, after the only barrier
    const int row = tile * 16 + rl;
    const bool valid = row < rows;
    const size_t rc = (size_t)(valid ? row : rows - 1);
    const float* xr = X + rc * kin + g * 4;
    f32x4 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    switch (part) {
        case 0: linear_split_part<NT, 0>(lds, xr, QT, g, rl, acc); break;
        case 1: linear_split_part<NT, 1>(lds, xr, QT, g, rl, acc); break;
        case 2: linear_split_part<NT, 2>(lds, xr, QT, g, rl, acc); break;
        default: linear_split_part<NT, 3>(lds, xr, QT, g, rl, acc); break;
    }
    if (valid) {
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int col = (part * TW + t) * 16;
            if (col < n1) {
                st4(Y1 + rc * n1 + col + g * 4, acc[t]);
            } else {
                float* p = Y2 + rc * n2 + (col - n1) + g * 4;
                st4(p, acc2 ? ld4(p) + acc[t] : acc[t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ LN-LSTM backward
// (the elementwise tile backward lives in lstm_bwd_tile.h)
template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void lnlstm_bwd_kernel(const LstmBwdTaskTable tt) {
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
    const float* __restrict__ dh_out = tt.task[k].dh_out;
    const float* __restrict__ dc_out_in = tt.task[k].dc_out;
    float* __restrict__ dz = tt.task[k].dz;
    float* __restrict__ dc_in = tt.task[k].dc_in;
    float* __restrict__ ln_partial = tt.task[k].workspace;
    const int rows = tt.task[k].rows;
    const int tiles_total = (rows + 15) / 16;
    const int qc = tt.qc[k];
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tt.task[k].uv);
    const float* __restrict__ Zx = tt.task[k].Zx;
    const float* __restrict__ KT = tt.task[k].KT;
    float* __restrict__ dxh = tt.task[k].dxh;
    constexpr int NT4 = D / 4, TPG = D / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int QX = dx >> 4, QT = QX + TPG;
    const bool resident = qc >= QT;
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    float* lds_k = lds;
    float* lds_kt = lds + (size_t)(resident ? QT : qc) * 16 * 4 * D;   // K^T for the fused data gradient (optional)
    float* lds_ln = lds_kt + (KT != nullptr ? (size_t)4 * D * (dx + D) : 0);
    float* slabs = lds_ln + 10 * D;
    float* slab = slabs + wave * 10 * D;
    for (int i = tid; i < 10 * D; i += blockDim.x) lds_ln[i] = ln[i];
    for (int i = tid; i < nw * 10 * D; i += blockDim.x) slabs[i] = 0.f;

    auto finish = [&](f32x4 (&acc)[NT4], size_t rc, bool valid) {
        f32x4 dco[TPG];
        const size_t o = rc * D + g * 4;
        f32x4 cf[TPG], dhn[TPG], dcn[TPG];
        lstm_tile_load<D>(c + o, dh_out ? dh_out + o : nullptr, dc_out_in ? dc_out_in + o : nullptr, cf, dhn, dcn);
        lstm_tile_backward<D>(acc, cf, dhn, dcn, dco, lds_ln, slab, g, rl, valid);
        if (valid) {
#pragma unroll
            for (int t = 0; t < NT4; ++t) st4(dz + rc * 4 * D + t * 16 + g * 4, acc[t]);
#pragma unroll
            for (int t = 0; t < TPG; ++t) st4(dc_in + o + t * 16, dco[t]);
        }
    };

    if (resident) {
        copy_to_lds(lds_k, K, (dx + D) * 4 * D, tid, blockDim.x);
        if (KT != nullptr) copy_to_lds(lds_kt, KT, 4 * D * (dx + D), tid, blockDim.x);
        const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
        const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
        __syncthreads();
        // static round-robin over the workgroup's tiles (not a ticket): which tiles a wavefront sums into its
        // LayerNorm-gradient slab must not depend on timing, or the gradients differ in the last bit from run to run
        for (int tile = t_beg + (tid >> 6); tile < t_end; tile += (int)(blockDim.x >> 6)) {
            const int row = tile * 16 + rl;
            const bool valid = row < rows;
            const size_t rc = (size_t)(valid ? row : rows - 1);
            f32x4 acc[NT4];
            if (uv != nullptr) {  // gather-init mode, see lnlstm_fwd_kernel
                const int2 ends = uv[rc];
                const float* zu = Zx + (size_t)ends.x * 4 * D + g * 4;
                const float* zv = Zx + (size_t)ends.y * 4 * D + g * 4;
#pragma unroll
                for (int t = 0; t < NT4; ++t) acc[t] = ld4(zu + t * 16);
#pragma unroll
                for (int t = 0; t < NT4; ++t) acc[t] += ld4(zv + t * 16);
            } else {
#pragma unroll
                for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            gemm_kloop<NT4>(acc, lds_k, 0, 0, QT, x + rc * dx + g * 4, h + rc * D + g * 4, QX, g, rl);
            finish(acc, rc, valid);
            if constexpr (D == 64) {
                if (KT != nullptr) {
                    // dh = dz Kh^T while dz is still in registers (gather-init mode, dx == 0): the D layout of dz is
                    // the B operand of the transposed-chaining GEMM, as between two MLP layers; saves a launch and a
                    // [rows,4D] read.  sched_barrier: dz (acc) is dead after the last k-step, keep it that way.
                    f32x4 out[TPG];
#pragma unroll
                    for (int t = 0; t < TPG; ++t) out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < NT4; q += 4) {
                        float b[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) b[i] = acc[q + (i >> 2)][i & 3];
                        ksteps<TPG, 16>(out, lds_kt + frag_off<TPG>(q * 4, g, rl), b);
                    }
                    if (valid) {
#pragma unroll
                        for (int t = 0; t < TPG; ++t) st4(dxh + rc * D + t * 16 + g * 4, out[t]);
                    }
                }
            }
        }
    } else {
        __syncthreads();
        const int rounds = (tiles_total + nw - 1) / nw;
        for (int r = my_blk; r < rounds; r += my_grid) {
            const int tile = r * nw + wave;
            const bool live = tile < tiles_total;
            const int row = tile * 16 + rl;
            const bool valid = live && row < rows;
            const size_t rc = (size_t)(valid ? row : rows - 1);
            f32x4 acc[NT4];
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int q0 = 0; q0 < QT; q0 += qc) {
                const int q1 = min(QT, q0 + qc);
                __syncthreads();
                copy_to_lds(lds_k, K + (size_t)q0 * 16 * 4 * D, (q1 - q0) * 16 * 4 * D, tid, blockDim.x);
                __syncthreads();
                if (live) gemm_kloop<NT4>(acc, lds_k, q0, q0, q1, x + rc * dx + g * 4, h + rc * D + g * 4, QX, g, rl);
            }
            finish(acc, rc, valid);
        }
    }
    // workgroup partial of the LayerNorm parameter gradients: fixed-order sum over the wavefront slabs
    __syncthreads();
    for (int i = tid; i < 10 * D; i += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += slabs[w * 10 * D + i];
        float* dst = ln_partial + (size_t)my_blk * 10 * D + i;
        *dst = tt.task[k].defer_reduce ? *dst + s : s;   // (this workgroup owns the row; launches are stream-ordered)
    }
}

// out[i] (+)= sum_{c < n_chunks} partial[c*stride + i]   -- second stage of every split reduction here.
// Thread (i, slice) first sums the chunks c = slice, slice+S, ... (S = blockDim.y slices, loads of
// different slices in flight together, each coalesced over i), then the slices are folded through LDS
// in a fixed order: deterministic, and parallel enough for n = 4096 outputs x 2048 chunks.
struct ReduceSeg {
    const float* partial; long long stride; float* out; int n; int blocks;   // blocks: workgroups this segment owns
};

__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceSeg a, const ReduceSeg b, int n_chunks, float scale,
                                                              int accumulate) {
    // The chunk partials are folded in float64: a gradient that is a small difference of large per-graph
    // contributions (labels 0/1 pull in opposite directions) would otherwise lose its digits HERE, in the one
    // place where thousands of fp32 partials of either sign meet; n_chunks * n adds, free on this chip.
    __shared__ double red[256];
    const bool second = (int)blockIdx.x >= a.blocks;      // (a second output segment of the same chunking rides along:
    const ReduceSeg& sg = second ? b : a;                 //  a weight gradient's bias row -- one launch instead of two)
    const int blk = second ? blockIdx.x - a.blocks : blockIdx.x;
    const int S = blockDim.y;  // slices; blockDim.x * S == 256
    const int i = blk * blockDim.x + threadIdx.x;
    double s = 0.0;
    if (i < sg.n) {
        // four independent running sums (loads of four chunks in flight per thread), folded in a fixed order
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = threadIdx.y;
        for (; c + 3 * S < n_chunks; c += 4 * S) {
            const float p0 = sg.partial[(size_t)c * sg.stride + i], p1 = sg.partial[(size_t)(c + S) * sg.stride + i];
            const float p2 = sg.partial[(size_t)(c + 2 * S) * sg.stride + i], p3 = sg.partial[(size_t)(c + 3 * S) * sg.stride + i];
            s0 += (double)p0;
            s1 += (double)p1;
            s2 += (double)p2;
            s3 += (double)p3;
        }
        for (; c < n_chunks; c += S) s0 += (double)sg.partial[(size_t)c * sg.stride + i];
        s = (s0 + s1) + (s2 + s3);
    }
    red[threadIdx.y * blockDim.x + threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && i < sg.n) {
        double t = red[threadIdx.x];
        for (int k = 1; k < S; ++k) t += red[k * blockDim.x + threadIdx.x];
        t *= (double)scale;
        sg.out[i] = (float)(accumulate ? t + (double)sg.out[i] : t);
    }
}

void reduce_partials2(const float* partial, int n_chunks, long long stride, float* out, int n, const float* partial_b,
                      long long stride_b, float* out_b, int n_b, float scale, int accumulate, hipStream_t st) {
    // many chunks: 16 outputs x 16 slices per workgroup; few chunks: 256 outputs x 1 slice
    const int S = n_chunks >= 64 ? 16 : (n_chunks >= 8 ? 4 : 1);
    const dim3 block(256 / S, S);
    const ReduceSeg a = {partial, stride, out, n, (int)((n + block.x - 1) / block.x)};
    const ReduceSeg b = {partial_b, stride_b, out_b, n_b, out_b != nullptr ? (int)((n_b + block.x - 1) / block.x) : 0};
    reduce_partials_kernel<<<a.blocks + b.blocks, block, 0, st>>>(a, b, n_chunks, scale, accumulate);
}

void reduce_partials(const float* partial, int n_chunks, long long stride, float* out, int n, float scale,
                     int accumulate, hipStream_t st) {
    reduce_partials2(partial, n_chunks, stride, out, n, nullptr, 0, nullptr, 0, scale, accumulate, st);
}

// ------------------------------------------------------------------------------------ MLP backward (data)
// g = dY; for l = L-1 .. 0:  g *= [A_l > 0] if layer l had relu;  dPre_l = g;  g = g W_l^T.   dX (+)= g.
// A_l for l < L-1 comes from the saved activations, A_{L-1} (only if the last layer has relu) from Yout.
struct MlpBwdTaskTable {
    tspgnn_mlp_bwd_task task[kMaxTasks];
    int blk_end[kMaxTasks];
    int n;
};

// ABF16: the saved activations (and Yout) are the bf16 arrays of a bf16-storage tape -- they only decide the relu masks.
template <int D, int MAXL, bool ABF16>
__global__ __launch_bounds__(512) void mlp_bwd_kernel(const MlpBwdTaskTable tt) {
    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const float* __restrict__ dY = tt.task[k].dY;
    const float* __restrict__ wt = tt.task[k].wt;
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
    const int tiles_total = (rows + 15) / 16;
    constexpr int NT = D / 16;
    __shared__ __attribute__((aligned(16))) float lds[MAXL * D * D + 4];
    int* ticket = reinterpret_cast<int*>(lds + MAXL * D * D);
    const int tid = threadIdx.x;
    copy_to_lds(lds, wt, n_layers * D * D, tid, blockDim.x);
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
        if (uv != nullptr) {  // gather-init mode: the adjoint of the row-sum aggregation, formed on the fly
            const int2 ends = uv[valid ? row : rows - 1];
            const float* pu = dY + (size_t)ends.x * D + g * 4;
            const float* pv = dY + (size_t)ends.y * D + g * 4;
#pragma unroll
            for (int q = 0; q < NT; ++q) a[q] = ld4(pu + q * 16) + ld4(pv + q * 16);
        } else {
#pragma unroll
            for (int q = 0; q < NT; ++q) a[q] = ld4(dY + rbase + q * 16);
        }
        for (int l = n_layers - 1; l >= 0; --l) {
            if ((relu_mask >> l) & 1u) {
                if constexpr (ABF16) {
                    const __bf16* A = (l == n_layers - 1) ? reinterpret_cast<const __bf16*>(Yout)
                                                          : reinterpret_cast<const __bf16*>(acts) + (size_t)l * acts_stride;
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const f32x4 av = widen(ldw4(A + rbase + q * 16));
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[q][r] = av[r] > 0.f ? a[q][r] : 0.f;
                    }
                } else {
                    const float* A = (l == n_layers - 1) ? Yout : acts + (size_t)l * acts_stride;
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const f32x4 av = ld4(A + rbase + q * 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[q][r] = av[r] > 0.f ? a[q][r] : 0.f;
                    }
                }
            }
            if (dpre != nullptr && valid) {
                float* dst = dpre + (size_t)l * dpre_stride + rbase;
#pragma unroll
                for (int q = 0; q < NT; ++q) st4(dst + q * 16, a[q]);
            }
            const float* wl = lds + l * D * D;
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (NT == 2) {
#pragma unroll
                for (int s = 0; s < D / 4; ++s) kstep<NT>(acc, wl + frag_off<NT>(s, g, rl), a[s >> 2][s & 3]);
            } else {
#pragma unroll
                for (int q = 0; q < NT; q += 4) {
                    float b[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) b[i] = a[q + (i >> 2)][i & 3];
                    ksteps<NT, 16>(acc, wl + frag_off<NT>(q * 4, g, rl), b);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = acc[t];
        }
        if (dX != nullptr && valid) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float* p = dX + rbase + t * 16;
                st4(p, acc_dx ? ld4(p) + a[t] : a[t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ weight gradient
// P[c][xf][yf] = sum_{r in chunk c} X[r][xf] * dY[r][yf]  for the (16*AV) x (16*BV) output block of this
// wavefront;  MFMA 16x16x4 with the ROWS as the contraction: lane (fl = l&15, k = l>>4) loads AV
// consecutive X features and BV consecutive dY features of row r0+k -- one fully coalesced 16*AV*4-byte
// row segment per lane group -- and feeds them to AV*BV MFMAs.  Output tile (m,n), lane (j,g), reg r:
//   X feature ib*16*AV + (4g+r)*AV + m,   dY feature jb*16*BV + j*BV + n.
template <int V>
struct VecLoad;
template <>
struct VecLoad<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const f32x4 t = ld4(p);
        v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
    }
};
template <>
struct VecLoad<2> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[2]) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x, v[1] = t.y;
    }
};
template <>
struct VecLoad<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[1]) { v[0] = *p; }
};

template <int AV, int BV>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                    long long rows, int kin, int nout, float* __restrict__ P,
                                                    float* __restrict__ Pb, int n_chunks, long long chunk_rows) {
    const int lane = threadIdx.x & 63, fl = lane & 15, k = lane >> 4;
    const int nbi = kin / (16 * AV), nbj = nout / (16 * BV), nob = nbi * nbj;
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= (long long)n_chunks * nob) return;  // wave-uniform
    const int c = (int)(w / nob), ob = (int)(w % nob), ib = ob / nbj, jb = ob % nbj;
    const long long r_beg = c * chunk_rows, r_end = min(rows, r_beg + chunk_rows);
    const float* xp = X + (size_t)ib * 16 * AV + fl * AV;
    const float* yp = dY + (size_t)jb * 16 * BV + fl * BV;
    f32x4 acc[AV][BV];
#pragma unroll
    for (int m = 0; m < AV; ++m)
#pragma unroll
        for (int n = 0; n < BV; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[BV];
#pragma unroll
    for (int n = 0; n < BV; ++n) cs[n] = 0.f;
    constexpr int UN = 8;  // k-steps (of 4 rows) in flight
    for (long long r0 = r_beg; r0 < r_end; r0 += 4 * UN) {
        float a[UN][AV], b[UN][BV];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long r = r0 + 4 * u + k;
            const bool ok = r < r_end;
            const long long rr = ok ? r : r_beg;
            VecLoad<AV>::ld(xp + rr * kin, a[u]);
            VecLoad<BV>::ld(yp + rr * nout, b[u]);
            if (!ok) {
#pragma unroll
                for (int m = 0; m < AV; ++m) a[u][m] = 0.f;
#pragma unroll
                for (int n = 0; n < BV; ++n) b[u][n] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int n = 0; n < BV; ++n) cs[n] += b[u][n];
#pragma unroll
            for (int m = 0; m < AV; ++m)
#pragma unroll
                for (int n = 0; n < BV; ++n) acc[m][n] = MFMA16(a[u][m], b[u][n], acc[m][n]);
        }
    }
    float* Pc = P + (size_t)c * kin * nout;
#pragma unroll
    for (int m = 0; m < AV; ++m)
#pragma unroll
        for (int n = 0; n < BV; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xf = ib * 16 * AV + (4 * k + r) * AV + m;
                const int yf = jb * 16 * BV + fl * BV + n;
                Pc[(size_t)xf * nout + yf] = acc[m][n][r];
            }
    if (Pb != nullptr && ib == 0) {
#pragma unroll
        for (int n = 0; n < BV; ++n) {
            float s = cs[n];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (k == 0) Pb[(size_t)c * nout + jb * 16 * BV + fl * BV + n] = s;
        }
    }
}

// The same reduction on the bf16 matrix cores with fp32-class accuracy (bf16x3, see dense_x3.hip) for the hot shape
// class kin % 64 == 0, nout % 64 == 0: 32 rows per step -- lane (f, g) loads the float4 of rows r0 + 8g .. 8g+7, eight
// consecutive contraction indices of v_mfma_f32_16x16x32_bf16 -- 96 MFMAs of 16 cycles instead of 128 of 32 per 32 rows,
// which turns the cell's [T*M,64]^T [T*M,256] product from MFMA-bound into HBM-bound.  Output layout as wgrad_kernel<4,4>.
typedef __bf16 bf16x8_w __attribute__((ext_vector_type(8)));
#define MFMA_BF16_W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split3_w(const float (&x)[8], bf16x8_w& hi, bf16x8_w& mid, bf16x8_w& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        hi[i] = h;
        mid[i] = m;
        lo[i] = (__bf16)(r1 - (float)m);
    }
}

// XB: X is a bf16 array (a bf16-storage tape: h, messages, hidden activations) -- bf16-exact, so its split is the value
// itself, and dY is taken to 16 significand bits (two pieces): two MFMA terms per product instead of six, half the bytes
// of X, and a third less splitting arithmetic, which is what bounds this launch at the wide shapes.
// NBI: 64-feature blocks of X per wavefront (output block 64 NBI x 64).  NBI = 2 (kin = 128 on a bf16 tape) splits every dY
// value once for two output blocks instead of once per block -- the splitting arithmetic, not the matrix pipe, bounds the
// launch -- at 128 accumulator registers (two wavefronts per SIMD instead of five).
template <bool XB, int NBI>
__global__ __launch_bounds__(256) void wgrad_x3_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                       long long rows, int kin, int nout, float* __restrict__ P,
                                                       float* __restrict__ Pb, int n_chunks, long long chunk_rows,
                                                       int xcd_map) {
    const int lane = threadIdx.x & 63, fl = lane & 15, g = lane >> 4;
    const int nbi = kin / (64 * NBI), nbj = nout / 64, nob = nbi * nbj;
    // the wavefronts of one chunk read the same rows (X nbj times, dY nbi times): past four of them (one workgroup) they
    // are kept on ONE XCD, whose L2 then serves the repeats (xcd_map; d = 128 cell, 128 x 512: 4.03 -> 3.79 ms per 5.1 M rows)
    const int blk = xcd_map ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const long long w = ((long long)blk * blockDim.x + threadIdx.x) >> 6;
    if (w >= (long long)n_chunks * nob) return;  // wave-uniform
    const int c = (int)(w / nob), ob = (int)(w % nob), ib = ob / nbj, jb = ob % nbj;
    const long long r_beg = c * chunk_rows, r_end = min(rows, r_beg + chunk_rows);
    const float* xp = X + (size_t)ib * NBI * 64 + fl * 4;
    const __bf16* xpb = reinterpret_cast<const __bf16*>(X) + (size_t)ib * NBI * 64 + fl * 4;
    const float* yp = dY + (size_t)jb * 64 + fl * 4;
    f32x4 acc[NBI][4][4];
#pragma unroll
    for (int q = 0; q < NBI; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[q][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long r0 = r_beg; r0 < r_end; r0 += 32) {
        f32x4 a[NBI][8], b[8];
        bf16x4 ab[NBI][8];     // XB: the bf16 rows as they are (their own hi piece: no widening, no split)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long r = r0 + 8 * g + j;
            const bool ok = r < r_end;
            const long long rr = ok ? r : r_beg;
#pragma unroll
            for (int q = 0; q < NBI; ++q) {
                if constexpr (XB) ab[q][j] = ldw4(xpb + rr * kin + q * 64);
                else a[q][j] = ld4(xp + rr * kin + q * 64);
            }
            b[j] = ld4(yp + rr * nout);
            if (!ok) {
#pragma unroll
                for (int q = 0; q < NBI; ++q) {
                    if constexpr (XB) ab[q][j] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    else a[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                b[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        bf16x8_w ah[NBI][4], am[NBI][4], al[NBI][4];
#pragma unroll
        for (int q = 0; q < NBI; ++q)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if constexpr (XB) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ah[q][m][j] = ab[q][j][m];
                } else {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = a[q][j][m];
                    split3_w(v, ah[q][m], am[q][m], al[q][m]);
                }
            }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = b[j][n];
                cs[n] += v[j];
            }
            bf16x8_w bh, bm, bl;
            split3_w(v, bh, bm, bl);
#pragma unroll
            for (int q = 0; q < NBI; ++q)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 d = acc[q][m][n];
                    if constexpr (!XB) {
                        d = MFMA_BF16_W(al[q][m], bh, d);  // smallest terms first
                        d = MFMA_BF16_W(am[q][m], bm, d);
                        d = MFMA_BF16_W(ah[q][m], bl, d);
                        d = MFMA_BF16_W(am[q][m], bh, d);
                    }
                    // (XB: dY enters with two pieces = 16 significand bits, as the fp32 operand of tspgnn_linear_bf16w_f32
                    // does: next to activations stored to 8 bits, a third piece buys nothing and costs a third of the launch)
                    d = MFMA_BF16_W(ah[q][m], bm, d);
                    d = MFMA_BF16_W(ah[q][m], bh, d);
                    acc[q][m][n] = d;
                }
        }
    }
    float* Pc = P + (size_t)c * kin * nout;
#pragma unroll
    for (int q = 0; q < NBI; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int xf = (ib * NBI + q) * 64 + (4 * g + r) * 4 + m;
                    const int yf = jb * 64 + fl * 4 + n;
                    Pc[(size_t)xf * nout + yf] = acc[q][m][n][r];
                }
    if (Pb != nullptr && ib == 0) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float s = cs[n];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (g == 0) Pb[(size_t)c * nout + jb * 64 + fl * 4 + n] = s;
        }
    }
}

static int pick_vec(int width) { return width % 64 == 0 ? 4 : (width % 32 == 0 ? 2 : 1); }

// Split of the row range used by tspgnn_wgrad_f32 (and its workspace size).
static void wgrad_plan(long long rows, int kin, int nout, int* n_chunks, long long* chunk_rows) {
    const int av = pick_vec(kin), bv = pick_vec(nout);
    const int nob = (kin / (16 * av)) * (nout / (16 * bv));
    long long target = (long long)n_cus() * 8 / nob;  // ~8 wavefronts per CU in total (HBM-bound: loads in flight)
    if (target < 1) target = 1;
    long long by_rows = (rows + 255) / 256;             // at least 256 rows per chunk
    long long nc = target < by_rows ? target : by_rows;
    if (nc < 1) nc = 1;
    long long cr = (rows + nc - 1) / nc;
    cr = (cr + 15) / 16 * 16;
    if (cr < 16) cr = 16;
    nc = (rows + cr - 1) / cr;
    if (nc < 1) nc = 1;
    *n_chunks = (int)nc;
    *chunk_rows = cr;
}

template <int NT>
static int launch_linear(const float* X, int kin, const float* Wp, float* Y1, int n1, float* Y2, int n2, int acc2,
                         int rows, hipStream_t st) {
    const int tiles = (rows + 15) / 16;
    const int QT = kin / 16;
    const size_t per_q = (size_t)16 * NT * 16 * sizeof(float);
    int qc = QT;
    if ((size_t)QT * per_q + 16 > 150 * 1024) qc = (int)((128 * 1024) / per_q);
    const size_t lds_bytes = (size_t)qc * per_q + 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_kernel<NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "linear: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
    int grid = n_cus() * per_cu;
    if constexpr (NT % 4 == 0) {
        if (qc >= QT && tiles <= n_cus() * 4) {   // few tiles: split each tile's columns over four wavefronts
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_split_kernel<NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return fail((int)e, "linear: hipFuncSetAttribute: %s", hipGetErrorString(e));
            linear_split_kernel<NT><<<(tiles + 1) / 2, 512, lds_bytes, st>>>(X, kin, Wp, Y1, n1, Y2, n2, acc2, rows, tiles);
            return launched("tspgnn_linear_f32");
        }
    }
    const int nw = (qc >= QT && tiles <= grid * 4) ? 4 : 8;
    const int max_grid = (tiles + nw - 1) / nw;
    if (grid > max_grid) grid = max_grid;
    linear_kernel<NT><<<grid, nw * 64, lds_bytes, st>>>(X, kin, Wp, Y1, n1, Y2, n2, acc2, rows, tiles, qc);
    return launched("tspgnn_linear_f32");
}

// Workgroups per task, proportional to cost[k] (at least one each); returns the grid.
static int split_blocks_bwd(const long long* cost, int n, int grid, int* blk_end) {
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
static int launch_lnlstm_bwd(const tspgnn_lstm_bwd_task* tasks, int n, hipStream_t st) {
    // D=128 keeps 4D/16 + temporaries > 256 registers live: one wavefront per SIMD (512-register budget).
    constexpr int NWMAX = D >= 128 ? 4 : 8;
    const size_t per_q = (size_t)16 * 4 * D * sizeof(float);
    auto extra = [&](int nw_) { return (size_t)(10 * D + nw_ * 10 * D + 4) * sizeof(float); };
    LstmBwdTaskTable tt;
    long long cost[kMaxTasks];
    long long tiles_all = 0;
    size_t lds_k = 0;
    bool any_chunked = false;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        const int QT = (tasks[k].dx + D) / 16;
        int qc = QT;
        if ((size_t)QT * per_q + extra(NWMAX) > 160 * 1024) {
            qc = (int)((160 * 1024 - extra(NWMAX)) / per_q);
            if (qc < 1 || tasks[k].uv) return fail(TSPGNN_EUNSUPPORTED, "lnlstm_bwd: d=%d does not fit LDS", D);
            any_chunked = true;
        }
        tt.qc[k] = qc;
        size_t need = (size_t)qc * per_q;
        if (tasks[k].KT != nullptr) {
            need += (size_t)4 * D * (tasks[k].dx + D) * sizeof(float);
            if (D != 64 || tasks[k].dx != 0 || qc < QT || need + extra(NWMAX) > 160 * 1024)
                return fail(TSPGNN_EUNSUPPORTED, "lnlstm_bwd: the fused data gradient needs d=64, dx=0 and K, K^T resident in LDS");
        }
        if (need > lds_k) lds_k = need;
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        cost[k] = tiles * (QT + 8);  // k-blocks + ~8 blocks' worth of elementwise backward
        tiles_all += tiles;
    }
    tt.n = n;
    int nw = NWMAX;
    if (!any_chunked && tiles_all <= (long long)n_cus() * 4) nw = 4;
    const size_t lds_bytes = lds_k + extra(nw);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_bwd_kernel<D, NWMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "lnlstm_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int grid = n_cus();
    const long long max_grid = (tiles_all + nw - 1) / nw;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks_bwd(cost, n, grid, tt.blk_end);
    lnlstm_bwd_kernel<D, NWMAX><<<grid, nw * 64, lds_bytes, st>>>(tt);
    int rc = launched("tspgnn_lnlstm_bwd_f32");
    if (rc) return rc;
    for (int k = 0; k < n; ++k) {
        if (tasks[k].defer_reduce) continue;
        const int nblk = tt.blk_end[k] - (k ? tt.blk_end[k - 1] : 0);
        reduce_partials(tasks[k].workspace, nblk, 10 * D, tasks[k].ln_grad, 10 * D, 1.0f, 1, st);
        if ((rc = launched("tspgnn_lnlstm_bwd_f32(reduce)"))) return rc;
    }
    return TSPGNN_OK;
}

template <int D, int MAXL>
static int launch_mlp_bwd(const tspgnn_mlp_bwd_task* tasks, int n, hipStream_t st) {
    MlpBwdTaskTable tt;
    long long cost[kMaxTasks];
    long long tiles_all = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        if (tt.task[k].acts && tt.task[k].acts_stride == 0) tt.task[k].acts_stride = (long long)tasks[k].rows * D;
        if (tt.task[k].dpre && tt.task[k].dpre_stride == 0) tt.task[k].dpre_stride = (long long)tasks[k].rows * D;
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        cost[k] = tiles * tasks[k].n_layers;
        tiles_all += tiles;
    }
    tt.n = n;
    const int lds_bytes = MAXL * D * D * 4 + 16;
    const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
    int grid = n_cus() * per_cu;
    const int nw = tiles_all <= (long long)grid * 4 ? 4 : 8;
    const long long max_grid = (tiles_all + nw - 1) / nw;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks_bwd(cost, n, grid, tt.blk_end);
    // (one flag per launch: the tasks of a launch come from one tape)
    if (tasks[0].acts_bf16) mlp_bwd_kernel<D, MAXL, true><<<grid, nw * 64, 0, st>>>(tt);
    else mlp_bwd_kernel<D, MAXL, false><<<grid, nw * 64, 0, st>>>(tt);
    return launched("tspgnn_mlp_bwd_f32");
}

template <int AV, int BV>
static int launch_wgrad(const float* X, const float* dY, long long rows, int kin, int nout, float* P, float* Pb,
                        int n_chunks, long long chunk_rows, hipStream_t st) {
    const int nob = (kin / (16 * AV)) * (nout / (16 * BV));
    const long long waves = (long long)n_chunks * nob;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    wgrad_kernel<AV, BV><<<grid, 256, 0, st>>>(X, dY, rows, kin, nout, P, Pb, n_chunks, chunk_rows);
    return launched("tspgnn_wgrad_f32");
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_linear_f32(const float* X, int kin, const float* Wp, float* Y1, int n1, float* Y2, int n2,
                                 int accumulate2, int rows, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "linear: rows=%d", rows);
    TSPGNN_REQUIRE(kin > 0 && kin % 16 == 0, "linear: kin=%d must be a positive multiple of 16", kin);
    TSPGNN_REQUIRE(n1 >= 0 && n2 >= 0 && n1 % 16 == 0 && n2 % 16 == 0, "linear: n1=%d n2=%d must be multiples of 16", n1,
                   n2);
    const int nout = n1 + n2;
    TSPGNN_REQUIRE(nout == 64 || nout == 128 || nout == 256, "linear: n1+n2=%d must be 64, 128 or 256", nout);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(X && Wp && (n1 == 0 || Y1) && (n2 == 0 || Y2), "linear: null pointer");
    hipStream_t st = as_stream(stream);
    switch (nout) {
        case 64: return launch_linear<4>(X, kin, Wp, Y1, n1, Y2, n2, accumulate2, rows, st);
        case 128: return launch_linear<8>(X, kin, Wp, Y1, n1, Y2, n2, accumulate2, rows, st);
        default: return launch_linear<16>(X, kin, Wp, Y1, n1, Y2, n2, accumulate2, rows, st);
    }
}

extern "C" long long tspgnn_lnlstm_bwd_workspace_floats(int d) { return (long long)(n_cus() + 8) * 10 * d; }

extern "C" int tspgnn_lnlstm_bwd_multi_f32(const tspgnn_lstm_bwd_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "lnlstm_bwd_multi: 1..%d tasks", kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "lnlstm_bwd: d=%d must be 32, 64 or 128", d);
    tspgnn_lstm_bwd_task live[kMaxTasks];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_lstm_bwd_task& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "lnlstm_bwd: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.dx >= 0 && t.dx % 16 == 0, "lnlstm_bwd: dx=%d must be a non-negative multiple of 16", t.dx);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE(t.h && t.c && t.K && t.ln && t.dz && t.dc_in && t.ln_grad && t.workspace && (t.dx == 0 || t.x),
                       "lnlstm_bwd: null pointer");
        TSPGNN_REQUIRE(!t.uv || (t.dx == 0 && t.Zx && (d == 32 || d == 64)),
                       "lnlstm_bwd: gather-init mode needs dx == 0, Zx and d in {32,64}");
        TSPGNN_REQUIRE(!t.zbias && !t.KTg, "lnlstm_bwd: a bias-init z / a streamed data gradient are f16x2 features (tspgnn_lnlstm_bwd_multi_h2)");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 32: return launch_lnlstm_bwd<32>(live, n, st);
        case 64: return launch_lnlstm_bwd<64>(live, n, st);
        default: return launch_lnlstm_bwd<128>(live, n, st);
    }
}

extern "C" int tspgnn_lnlstm_bwd_finish_f32(const float* workspace, float* ln_grad, int d, void* stream) {
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "lnlstm_bwd_finish: d=%d must be 32, 64 or 128", d);
    TSPGNN_REQUIRE(workspace && ln_grad, "lnlstm_bwd_finish: null pointer");
    reduce_partials(workspace, n_cus() + 8, 10 * d, ln_grad, 10 * d, 1.0f, 1, as_stream(stream));
    return launched("tspgnn_lnlstm_bwd_finish_f32");
}

extern "C" int tspgnn_lnlstm_bwd_f32(const float* x, int dx, const float* h, const float* c, const float* K,
                                     const float* ln, const float* dh_out, const float* dc_out, float* dz,
                                     float* dc_in, float* ln_grad, float* workspace, int rows, int d, void* stream) {
    const tspgnn_lstm_bwd_task t = {x, dx, h, c, K, ln, dh_out, dc_out, dz, dc_in, ln_grad, workspace, rows, nullptr, nullptr};
    return tspgnn_lnlstm_bwd_multi_f32(&t, 1, d, stream);
}

extern "C" int tspgnn_lnlstm_gather_bwd_f32(const int32_t* uv, const float* Zx, const float* h, const float* c,
                                            const float* Kh, const float* ln, const float* dh_out, const float* dc_out,
                                            float* dz, float* dc_in, float* ln_grad, float* workspace, int rows, int d,
                                            void* stream) {
    TSPGNN_REQUIRE(rows == 0 || (uv && Zx), "lnlstm_gather_bwd: null pointer");
    const tspgnn_lstm_bwd_task t = {nullptr, 0, h, c, Kh, ln, dh_out, dc_out, dz, dc_in, ln_grad, workspace, rows, uv, Zx};
    return tspgnn_lnlstm_bwd_multi_f32(&t, 1, d, stream);
}

extern "C" int tspgnn_mlp_bwd_multi_f32(const tspgnn_mlp_bwd_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "mlp_bwd_multi: 1..%d tasks", kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "mlp_bwd: d=%d must be 32, 64 or 128", d);
    tspgnn_mlp_bwd_task live[kMaxTasks];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_mlp_bwd_task& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "mlp_bwd: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 4, "mlp_bwd: n_layers=%d must be in 1..4", t.n_layers);
        if (d == 128 && t.n_layers > 2)
            return fail(TSPGNN_EUNSUPPORTED, "mlp_bwd: d=128 holds at most 2 layers in LDS (got %d)", t.n_layers);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE(t.dY && t.wt, "mlp_bwd: null pointer");
        const unsigned inner = t.relu_mask & ((1u << (t.n_layers - 1)) - 1u);
        TSPGNN_REQUIRE(!inner || t.acts, "mlp_bwd: relu layers need the saved activations");
        TSPGNN_REQUIRE(!((t.relu_mask >> (t.n_layers - 1)) & 1u) || t.Yout, "mlp_bwd: relu on the last layer needs Yout");
        TSPGNN_REQUIRE(n == 0 || (t.acts_bf16 != 0) == (live[0].acts_bf16 != 0), "mlp_bwd: the tasks of a launch share acts_bf16");
        TSPGNN_REQUIRE(!t.pre_X, "mlp_bwd: pre_X is an f16x2 feature (tspgnn_mlp_bwd_multi_h2)");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    switch (d) {
        case 32: return launch_mlp_bwd<32, 4>(live, n, st);
        case 64: return launch_mlp_bwd<64, 4>(live, n, st);
        default: return launch_mlp_bwd<128, 2>(live, n, st);
    }
}

extern "C" int tspgnn_mlp_bwd_f32(const float* dY, const float* wt, const float* acts, long long acts_stride,
                                  const float* Yout, float* dpre, long long dpre_stride, float* dX, int accumulate_dx,
                                  int rows, int d, int n_layers, unsigned relu_mask, void* stream) {
    const tspgnn_mlp_bwd_task t = {dY, wt, acts, acts_stride, Yout, dpre, dpre_stride, dX, accumulate_dx, rows, n_layers,
                                   relu_mask, nullptr, 0};
    return tspgnn_mlp_bwd_multi_f32(&t, 1, d, stream);
}

extern "C" long long tspgnn_wgrad_workspace_floats(long long rows, int kin, int nout) {
    if (rows <= 0 || kin <= 0 || nout <= 0 || kin % 16 || nout % 16) return 0;
    int nc;
    long long cr;
    wgrad_plan(rows, kin, nout, &nc, &cr);
    return (long long)nc * ((long long)kin * nout + nout);
}

extern "C" int tspgnn_wgrad_bf16x_f32(const void* X, const float* dY, long long rows, int kin, int nout, float* dW,
                                      float* db, float* workspace, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "wgrad_bf16x: rows=%lld", rows);
    if (kin <= 0 || nout <= 0 || kin % 64 || nout % 64)
        return fail(TSPGNN_EUNSUPPORTED, "wgrad_bf16x: kin=%d and nout=%d must be positive multiples of 64", kin, nout);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(X && dY && dW && workspace, "wgrad_bf16x: null pointer");
    int nc;
    long long cr;
    wgrad_plan(rows, kin, nout, &nc, &cr);
    float* P = workspace;
    float* Pb = db ? workspace + (size_t)nc * kin * nout : nullptr;
    hipStream_t st = as_stream(stream);
    if (kin % 128 == 0) {   // two X blocks per wavefront: every dY value is split once per 128 features
        const int nob = (kin / 128) * (nout / 64);
        const unsigned grid = (unsigned)(((long long)nc * nob + 3) / 4);
        wgrad_x3_kernel<true, 2><<<grid, 256, 0, st>>>(reinterpret_cast<const float*>(X), dY, rows, kin, nout, P, Pb, nc, cr,
                                                      nob > 4);
    } else {
        const int nob = (kin / 64) * (nout / 64);
        const unsigned grid = (unsigned)(((long long)nc * nob + 3) / 4);
        wgrad_x3_kernel<true, 1><<<grid, 256, 0, st>>>(reinterpret_cast<const float*>(X), dY, rows, kin, nout, P, Pb, nc, cr,
                                                      nob > 4);
    }
    int rc = launched("tspgnn_wgrad_bf16x_f32");
    if (rc) return rc;
    const int n = kin * nout;
    reduce_partials2(P, nc, n, dW, n, Pb, nout, db, db ? nout : 0, 1.0f, 1, st);
    return launched("tspgnn_wgrad_bf16x_f32(reduce)");
}

extern "C" int tspgnn_wgrad_f32(const float* X, const float* dY, long long rows, int kin, int nout, float* dW,
                                float* db, float* workspace, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "wgrad: rows=%lld", rows);
    TSPGNN_REQUIRE(kin > 0 && kin % 16 == 0 && nout > 0 && nout % 16 == 0,
                   "wgrad: kin=%d and nout=%d must be positive multiples of 16", kin, nout);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(X && dY && dW && workspace, "wgrad: null pointer");
    int nc;
    long long cr;
    wgrad_plan(rows, kin, nout, &nc, &cr);
    float* P = workspace;
    float* Pb = db ? workspace + (size_t)nc * kin * nout : nullptr;
    hipStream_t st = as_stream(stream);
    const int av = pick_vec(kin), bv = pick_vec(nout);
    int rc;
#define TSPGNN_WG(A, B) rc = launch_wgrad<A, B>(X, dY, rows, kin, nout, P, Pb, nc, cr, st)
    if (av == 4 && bv == 4 && rows >= 4096) {   // the big reductions over T*rows: bf16 matrix cores, fp32-class accuracy
        const int nob = (kin / 64) * (nout / 64);
        const unsigned grid = (unsigned)(((long long)nc * nob + 3) / 4);
        wgrad_x3_kernel<false, 1><<<grid, 256, 0, st>>>(X, dY, rows, kin, nout, P, Pb, nc, cr, nob > 4);
        rc = launched("tspgnn_wgrad_f32");
    } else if (av == 4 && bv == 4) TSPGNN_WG(4, 4);
    else if (av == 4 && bv == 2) TSPGNN_WG(4, 2);
    else if (av == 4 && bv == 1) TSPGNN_WG(4, 1);
    else if (av == 2 && bv == 4) TSPGNN_WG(2, 4);
    else if (av == 2 && bv == 2) TSPGNN_WG(2, 2);
    else if (av == 2 && bv == 1) TSPGNN_WG(2, 1);
    else if (av == 1 && bv == 4) TSPGNN_WG(1, 4);
    else if (av == 1 && bv == 2) TSPGNN_WG(1, 2);
    else TSPGNN_WG(1, 1);
#undef TSPGNN_WG
    if (rc) return rc;
    const int n = kin * nout;
    reduce_partials2(P, nc, n, dW, n, Pb, nout, db, db ? nout : 0, 1.0f, 1, st);   // (the bias row in the same launch)
    return launched("tspgnn_wgrad_f32(reduce)");
}
