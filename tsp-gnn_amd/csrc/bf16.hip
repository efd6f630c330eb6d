// bf16-storage / fp32-accumulate variants of the forward path (BASELINE config 5: "bf16 embeddings with fp32
// accumulate"): embeddings h, messages, aggregates and the projected messages Zx live in
// HBM as bf16 (half the bytes of every [rows,d] stream), the GEMMs are single v_mfma_f32_16x16x32_bf16 products of
// bf16 operands accumulated in fp32, and everything the recurrence is sensitive to -- the cell state c, the
// LayerNorm statistics and parameters, biases, the gate arithmetic -- stays fp32.  Weights are the fp32 variables
// rounded to bf16 (round-to-nearest-even) = piece 0 of tspgnn_pack_weights_x3, same fragment order as dense_x3.hip.
// Same tile machinery as dense_x3.hip: a wavefront owns 16 rows, OUT^T = W^T IN^T, the D fragment of one layer is
// converted in registers into the next layer's B operand.  d = 128 fits here (Kh[128,512] bf16 = 128 KB of LDS).
#include "common.h"
#include "bf16_tile.h"
#include "mfma_tile.h"

namespace tspgnn {

constexpr int kMaxTasksB = 4;

static int split_blocks_b(const long long* cost, int n, int grid, int* blk_end) {
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

// ---------------------------------------------------------------------------------- aggregation (bf16 rows)
// One 16-byte lane = 8 bf16; LPR = d/8 lanes per row.  Sums in fp32, one rounding at the store.
__device__ __forceinline__ void add8(float (&acc)[8], uint4 w) {
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[2 * i] += __uint_as_float(u[i] << 16);
        acc[2 * i + 1] += __uint_as_float(u[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 round8(const float (&acc)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)acc[i];
    return *reinterpret_cast<uint4*>(&v);
}

__global__ __launch_bounds__(256) void gather2_sum_bf16_kernel(const int2* __restrict__ uv, const uint4* __restrict__ X,
                                                               uint4* __restrict__ Y, int M, int lpr) {
    const long long total = (long long)M * lpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int e = (int)(i / lpr);
        const int c = (int)(i - (long long)e * lpr);
        const int2 ends = uv[e];
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        add8(acc, X[(long long)ends.x * lpr + c]);
        add8(acc, X[(long long)ends.y * lpr + c]);
        Y[i] = round8(acc);
    }
}

// One wavefront per vertex, RPW = 64/LPR source rows per step (see csr_rowsum_body in aggregate.hip).
template <int LPR>
__global__ __launch_bounds__(256) void csr_rowsum_bf16_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                              const uint4* __restrict__ X, uint4* __restrict__ Y, int N) {
    constexpr int RPW = kWave / LPR;
    // XCD-aware vertex order (workgroup b runs on XCD b % 8): each XCD owns a contiguous eighth of the vertices, so
    // the two reads of every edge row (one per endpoint) meet in one L2 -- see csr_rowsum_body in aggregate.hip
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const unsigned vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;  // bijective for any nb
    const int v = (int)(((long long)vb * blockDim.x + threadIdx.x) >> 6);
    if (v >= N) return;  // wave-uniform
    const int lane = threadIdx.x & 63, sub = lane / LPR, c = lane % LPR;
    const int beg = rowptr[v], end = rowptr[v + 1];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int base = beg; base < end; base += kWave) {
        const int cnt = min(kWave, end - base);
        const int my_e = (lane < cnt) ? eid[base + lane] : 0;
#pragma unroll 4
        for (int k0 = 0; k0 < cnt; k0 += RPW) {
            const int k = k0 + sub;
            const int e = __shfl(my_e, min(k, cnt - 1));
            if (k < cnt) add8(acc, X[(long long)e * LPR + c]);
        }
    }
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off);
    }
    if (sub == 0) Y[(long long)v * LPR + c] = round8(acc);
}

// ---------------------------------------------------------------------------------- storage conversions
// fp32 <-> bf16 (round to nearest even), 8 elements per lane: what the bf16-storage mode needs at its two ends (the caller's
// fp32 embeddings in, the vote head's fp32 input out) without borrowing a tensor library's kernels.
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long long n) {
    const long long n8 = n >> 3, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = ld4(x + i * 8), b = ld4(x + i * 8 + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        *reinterpret_cast<uint4*>(y + i * 8) = round8(v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[(n8 << 3) + threadIdx.x] = (__bf16)x[(n8 << 3) + threadIdx.x];
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, long long n) {
    const long long n8 = n >> 3, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        add8(v, *reinterpret_cast<const uint4*>(x + i * 8));
        st4(y + i * 8, f32x4{v[0], v[1], v[2], v[3]});
        st4(y + i * 8 + 4, f32x4{v[4], v[5], v[6], v[7]});
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[(n8 << 3) + threadIdx.x] = (float)x[(n8 << 3) + threadIdx.x];
}

// ---------------------------------------------------------------------------------- MLP (bf16)
// wb: n_layers blocks of { bf16 packed[D*D] (piece 0 of pack_weights_x3), float bias[D] }; proj_w: bf16 packed [D,4D].
struct MlpTableB {
    tspgnn_mlp_task_bf16 task[kMaxTasksB];
    int blk_end[kMaxTasksB];
    int n;
};

// PROJ = false drops the projection phase (and its D/4 accumulator tiles) from the kernel: a launch without
// projections keeps to ~70 registers at d=128 and runs at twice the occupancy.
template <int D, int NW, bool PROJ>
__global__ __launch_bounds__(NW * 64) void mlp_fwd_bf16_kernel(const MlpTableB tt) {
    constexpr int NT = D / 16, KB = D / 32, NP = D / 4;
    constexpr int LAYER_BYTES = D * D * 2 + D * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(tt.task[k].X);
    const unsigned char* __restrict__ wb = reinterpret_cast<const unsigned char*>(tt.task[k].wb);
    __bf16* __restrict__ Y = reinterpret_cast<__bf16*>(tt.task[k].Y);
    const int rows = tt.task[k].rows, n_layers = tt.task[k].n_layers;
    const unsigned relu_mask = tt.task[k].relu_mask;
    const __bf16* __restrict__ proj_w = reinterpret_cast<const __bf16*>(tt.task[k].proj_w);
    __bf16* __restrict__ proj_out = reinterpret_cast<__bf16*>(tt.task[k].proj_out);
    const bool x_blk = tt.task[k].x_blocked != 0, y_inter = tt.task[k].y_interleaved != 0;
    __bf16* __restrict__ acts = reinterpret_cast<__bf16*>(tt.task[k].acts);   // training: the stored hidden activations
    const long long acts_stride = tt.task[k].acts_stride;
    const int tiles_total = (rows + 15) / 16;
    int* ticket = reinterpret_cast<int*>(lds);
    unsigned char* lds_w = lds + 16;

    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4;
    copy_to_lds(reinterpret_cast<float*>(lds_w), reinterpret_cast<const float*>(wb), n_layers * LAYER_BYTES / 4, tid, blockDim.x);
    const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
    const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
    if (tid == 0) *ticket = t_beg;
    __syncthreads();
    auto next_ticket = [&]() {
        int t = 0;
        if (lane == 0) t = atomicAdd(ticket, 1);
        return __builtin_amdgcn_readfirstlane(t);
    };
    auto load_row = [&](int tile, bf16x8 (&dst)[KB]) {   // (rows past the end: the last row, never stored)
        const __bf16* xr = X + c_blocked<D>((unsigned)min(tile * 16 + rl, rows - 1), g, x_blk);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) dst[kb] = row_operand(xr, kb, x_blk);
    };
    int tile = next_ticket();
    bf16x8 bn[KB];                                       // the input rows of the NEXT tile, loaded a tile ahead
    load_row(tile < t_end ? tile : t_beg, bn);
    while (tile < t_end) {
        const int ntile = next_ticket();
        const int row = tile * 16 + rl;
        const bool valid = row < rows;
        const size_t rbase = (size_t)(valid ? row : rows - 1) * D + g * 4;
        bf16x8 b[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b[kb] = bn[kb];
        load_row(ntile < t_end ? ntile : tile, bn);
        f32x4 acc[NT];
        for (int l = 0; l < n_layers; ++l) {
            const __bf16* wl = reinterpret_cast<const __bf16*>(lds_w + (size_t)l * LAYER_BYTES);
            const float* bl = reinterpret_cast<const float*>(lds_w + (size_t)l * LAYER_BYTES + D * D * 2);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = ld4(bl + t * 16 + g * 4);
            gemm_frags_pf<NT, 0, NT, KB, 4>(acc, wl + ((size_t)g * NT * 16 + rl) * 8, b);
            if ((relu_mask >> l) & 1u) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) b[kb] = join(narrow(acc[2 * kb]), narrow(acc[2 * kb + 1]));  // stored precision
            if (acts != nullptr && l < n_layers - 1 && valid) {
                __bf16* dst = acts + (size_t)l * acts_stride + rbase;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    stw4(dst + (2 * kb) * 16, bf16x4{b[kb][0], b[kb][1], b[kb][2], b[kb][3]});
                    stw4(dst + (2 * kb + 1) * 16, bf16x4{b[kb][4], b[kb][5], b[kb][6], b[kb][7]});
                }
            }
        }
        if (valid && y_inter) {   // (last layer packed with interleaved columns: the lane's 8 values are 8 consecutive columns)
            __bf16* dst = Y + (size_t)row * D + g * 8;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) *reinterpret_cast<bf16x8*>(dst + kb * 32) = b[kb];
        } else if (valid) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                stw4(Y + rbase + (2 * kb) * 16, bf16x4{b[kb][0], b[kb][1], b[kb][2], b[kb][3]});
                stw4(Y + rbase + (2 * kb + 1) * 16, bf16x4{b[kb][4], b[kb][5], b[kb][6], b[kb][7]});
            }
        }
        tile = ntile;
    }
    if (PROJ && proj_w != nullptr) {  // second phase: proj_out = Y P, P packed [D, 4D]
        __threadfence_block();
        __syncthreads();
        copy_to_lds(reinterpret_cast<float*>(lds_w), reinterpret_cast<const float*>(proj_w), D * 4 * D * 2 / 4, tid, blockDim.x);
        if (tid == 0) *ticket = t_beg;
        __syncthreads();
        const __bf16* wp = reinterpret_cast<const __bf16*>(lds_w);
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
            bf16x8 bv[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) bv[kb] = row_operand(Y + rc * D + g * 4, kb);
            gemm_frags_pf<NP, 0, NP, KB, 4>(acc, wp + ((size_t)g * NP * 16 + rl) * 8, bv);
            if (valid) {
#pragma unroll
                for (int t = 0; t < NP; ++t) stw4(proj_out + zx_blocked<D>((unsigned)rc, g) + t * 256, narrow(acc[t]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------- LN-LSTM (bf16)
// K: bf16 packed kernel[dx+D, 4D] (or Kh[D,4D] in gather-init mode: z starts at Zx[u] + Zx[v], Zx bf16 [n_src,4D]).
struct LstmTableB {
    tspgnn_lstm_task_bf16 task[kMaxTasksB];
    int blk_end[kMaxTasksB];
    int kbc[kMaxTasksB];  // k-blocks per LDS chunk; >= all of K: resident
    int n;
};

// Pieces of the staged gather-init tile (below): the raw projected-message rows of one gate range, the accumulator
// start z = Zx[u] + Zx[v], and the h Kh product of that range with the weight fragments PF deep in flight ahead of the
// MFMAs (an LDS read is ~100 cycles, an MFMA 16: read-then-multiply in program order leaves the matrix pipe idle).
template <int T0, int TN>
__device__ __forceinline__ void zx_load_part(bf16x4 (&ru)[TN], bf16x4 (&rv)[TN], const __bf16* zu, const __bf16* zv) {
#pragma unroll
    for (int t = 0; t < TN; ++t) ru[t] = ldw4(zu + (T0 + t) * 256);
#pragma unroll
    for (int t = 0; t < TN; ++t) rv[t] = ldw4(zv + (T0 + t) * 256);
}
template <int D, int T0, int TN, int PF>
__device__ __forceinline__ void gather_gemm_part(f32x4 (&acc)[TN], const bf16x4 (&ru)[TN], const bf16x4 (&rv)[TN],
                                                 const __bf16* lds_w, const bf16x8 (&bv)[D / 32], int g, int rl) {
    constexpr int NT4 = D / 4, KB = D / 32;
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = widen(ru[t]) + widen(rv[t]);
    gemm_frags_pf<NT4, T0, TN, KB, PF>(acc, lds_w + ((size_t)g * NT4 * 16 + rl) * 8, bv);
}

// STAGED: a resident gather-init (edge) task forms z gate by gate -- f, then (i, j), then o (mfma_tile.h's three
// stages, bit-identical to the one-stage cell) -- with at most 64 accumulator registers live, which leaves room for
// what the all-gates form at d = 128 (254 registers) cannot afford: every global load of a stage issued a stage ahead,
// the weight fragments prefetched, the next tile's endpoints fetched a tile ahead, and a third wavefront per SIMD.
template <int D, int NW, bool STAGED>
__global__ __launch_bounds__(NW * 64) void lnlstm_fwd_bf16_kernel(const LstmTableB tt) {
    constexpr int NT4 = D / 4, TPG = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const __bf16* __restrict__ x = reinterpret_cast<const __bf16*>(tt.task[k].x);
    const int dx = tt.task[k].dx;
    const __bf16* __restrict__ h = reinterpret_cast<const __bf16*>(tt.task[k].h);
    const float* __restrict__ c = tt.task[k].c;
    const __bf16* __restrict__ K = reinterpret_cast<const __bf16*>(tt.task[k].K);
    const float* __restrict__ ln = tt.task[k].ln;
    __bf16* __restrict__ h_out = reinterpret_cast<__bf16*>(tt.task[k].h_out);
    float* __restrict__ c_out = tt.task[k].c_out;
    const bool c_in_blk = tt.task[k].state_in_blocked != 0, c_out_blk = tt.task[k].state_out_blocked != 0;   // h and c alike
    const int rows = tt.task[k].rows;
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tt.task[k].uv);
    const __bf16* __restrict__ Zx = reinterpret_cast<const __bf16*>(tt.task[k].Zx);
    const int tiles_total = (rows + 15) / 16;
    const int KBT = (dx + D) >> 5, KBX = dx >> 5;
    const int kbc = tt.kbc[k];
    const bool resident = kbc >= KBT;

    float* lds_ln = reinterpret_cast<float*>(ldsb);
    int* ticket = reinterpret_cast<int*>(lds_ln + 10 * D);
    __bf16* lds_w = reinterpret_cast<__bf16*>(ldsb + (10 * D + 4) * sizeof(float));
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    // (gates i, f, o feed sigmoids only: gamma / beta stored times -log2(e), forget bias folded in -- lstm_gates<D, true>)
    for (int i = tid; i < 10 * D; i += blockDim.x) {
        const int r = i / D;
        float v = ln[i];
        if (r == 5) v += 1.0f;
        if (r < 2 || (r >= 4 && r < 8)) v *= -1.4426950408889634f;
        lds_ln[i] = v;
    }

    auto stage = [&](int kb0, int kb1) {  // K is k-block major: one contiguous range
        copy_to_lds(reinterpret_cast<float*>(lds_w), reinterpret_cast<const float*>(K + (size_t)kb0 * 32 * 4 * D),
                    (kb1 - kb0) * 32 * 4 * D * 2 / 4, tid, blockDim.x);
    };
    auto init_acc = [&](f32x4 (&acc)[NT4], size_t rc) {
        if (uv != nullptr) {
            const int2 ends = uv[rc];
            const __bf16* zu = Zx + zx_blocked<D>((unsigned)ends.x, g);
            const __bf16* zv = Zx + zx_blocked<D>((unsigned)ends.y, g);
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = widen(ldw4(zu + t * 256)) + widen(ldw4(zv + t * 256));
        } else {
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto kloop = [&](f32x4 (&acc)[NT4], size_t rc, int kb_base, int kb0, int kb1) {
        const __bf16* xrow = x + rc * dx + g * 4;
        const __bf16* hrow = h + c_blocked<D>((unsigned)rc, g, c_in_blk);
        for (int kb = kb0; kb < kb1; ++kb) {
            const bf16x8 bv = kb < KBX ? row_operand(xrow, kb) : row_operand(hrow, kb - KBX, c_in_blk);
            const __bf16* base = lds_w + ((size_t)((kb - kb_base) * 4 + g) * NT4 * 16 + rl) * 8;
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = MFMA_BF16(ldw8(base + t * 128), bv, acc[t]);
        }
    };
    auto cell = [&](f32x4 (&acc)[NT4], size_t rc, bool valid) {
        f32x4 cf[TPG], hn[TPG], nc[TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t)
            cf[t] = c != nullptr ? ld4(c + c_blocked<D>((unsigned)rc, g, c_in_blk) + t * (c_in_blk ? 256 : 16)) : f32x4{0.f, 0.f, 0.f, 0.f};
        lstm_gates<D, true, true>(acc, cf, lds_ln, g, hn, nc);
        if (valid) {
#pragma unroll
            for (int t = 0; t < TPG; ++t) {
                stw4(h_out + c_blocked<D>((unsigned)rc, g, c_out_blk) + t * (c_out_blk ? 256 : 16), narrow(hn[t]));
                st4(c_out + c_blocked<D>((unsigned)rc, g, c_out_blk) + t * (c_out_blk ? 256 : 16), nc[t]);
            }
        }
    };

    if (resident) {
        stage(0, KBT);
        const int pos = xcd_contiguous(my_blk, my_grid);   // the edges of an XCD gather from one slice of Zx
        const int t_beg = (int)((long long)tiles_total * pos / my_grid);
        const int t_end = (int)((long long)tiles_total * (pos + 1) / my_grid);
        if (tid == 0) *ticket = t_beg;
        __syncthreads();
        if (STAGED && uv != nullptr) {
            auto next_ticket = [&]() {
                int t = 0;
                if (lane == 0) t = atomicAdd(ticket, 1);
                return __builtin_amdgcn_readfirstlane(t);
            };
            auto row_of = [&](int tile) { return (unsigned)min(tile * 16 + rl, rows - 1); };
            int tile = next_ticket();
            int2 ends = uv[row_of(tile < t_end ? tile : t_beg)];
            while (tile < t_end) {
                const int ntile = next_ticket();
                const bool valid = tile * 16 + rl < rows;
                const unsigned rc = row_of(tile);
                const __bf16* zu = Zx + zx_blocked<D>((unsigned)ends.x, g);
                const __bf16* zv = Zx + zx_blocked<D>((unsigned)ends.y, g);
                bf16x4 fu[TPG], fv[TPG];
                zx_load_part<2 * TPG, TPG>(fu, fv, zu, zv);
                const __bf16* hrow = h + c_blocked<D>(rc, g, c_in_blk);
                bf16x8 bv[D / 32];
#pragma unroll
                for (int kb = 0; kb < D / 32; ++kb) bv[kb] = row_operand(hrow, kb, c_in_blk);
                f32x4 cs[TPG];
#pragma unroll
                for (int t = 0; t < TPG; ++t)
                    cs[t] = c != nullptr ? ld4(c + c_blocked<D>(rc, g, c_in_blk) + t * (c_in_blk ? 256 : 16)) : f32x4{0.f, 0.f, 0.f, 0.f};
                ends = uv[row_of(ntile < t_end ? ntile : tile)];   // the next tile's endpoints, a tile ahead
                bf16x4 iu[2 * TPG], iv[2 * TPG];
                zx_load_part<0, 2 * TPG>(iu, iv, zu, zv);           // stage (i, j)'s rows behind stage f's product
                constexpr int PF = 4;                               // weight fragments in flight
                bf16x4 ou[TPG], ov[TPG];
                {
                    f32x4 zf[TPG];
                    gather_gemm_part<D, 2 * TPG, TPG, PF>(zf, fu, fv, lds_w, bv, g, rl);
                    zx_load_part<3 * TPG, TPG>(ou, ov, zu, zv);     // stage o's rows behind stage (i, j)'s product
                    lstm_stage_f<D, true>(zf, cs, lds_ln, g);
                }
                {
                    f32x4 zij[2 * TPG];
                    gather_gemm_part<D, 0, 2 * TPG, PF>(zij, iu, iv, lds_w, bv, g, rl);
                    lstm_stage_ij<D, true>(zij, cs, lds_ln, g);
                }
                f32x4 hn[TPG];
                {
                    f32x4 zo[TPG];
                    gather_gemm_part<D, 3 * TPG, TPG, PF>(zo, ou, ov, lds_w, bv, g, rl);
                    lstm_stage_o<D, true>(zo, cs, lds_ln, g, hn);
                }
                if (valid) {   // (all stores after the tile's last load: a wait behind mixed loads and stores is vmcnt(0))
#pragma unroll
                    for (int t = 0; t < TPG; ++t) {
                        st4(c_out + c_blocked<D>(rc, g, c_out_blk) + t * (c_out_blk ? 256 : 16), cs[t]);
                        stw4(h_out + c_blocked<D>(rc, g, c_out_blk) + t * (c_out_blk ? 256 : 16), narrow(hn[t]));
                    }
                }
                tile = ntile;
            }
        } else {
            for (;;) {
                int tile = 0;
                if (lane == 0) tile = atomicAdd(ticket, 1);
                tile = __builtin_amdgcn_readfirstlane(tile);
                if (tile >= t_end) break;
                const int row = tile * 16 + rl;
                const bool valid = row < rows;
                const size_t rc = (size_t)(valid ? row : rows - 1);
                f32x4 acc[NT4];
                init_acc(acc, rc);
                kloop(acc, rc, 0, 0, KBT);
                cell(acc, rc, valid);
            }
        }
    } else {
        const int rounds = (tiles_total + NW - 1) / NW;
        for (int r = my_blk; r < rounds; r += my_grid) {
            const int tile = r * NW + wave;
            const bool live = tile < tiles_total;
            const int row = tile * 16 + rl;
            const bool valid = live && row < rows;
            const size_t rc = (size_t)(valid ? row : rows - 1);
            f32x4 acc[NT4];
            init_acc(acc, rc);
            for (int kb0 = 0; kb0 < KBT; kb0 += kbc) {
                const int kb1 = min(KBT, kb0 + kbc);
                __syncthreads();
                stage(kb0, kb1);
                __syncthreads();
                if (live) kloop(acc, rc, kb0, kb0, kb1);
            }
            cell(acc, rc, valid);
        }
    }
}

template <int D, int NW, bool PROJ>
static int launch_mlp_b(const tspgnn_mlp_task_bf16* tasks, int n, hipStream_t st) {
    MlpTableB tt;
    long long cost[kMaxTasksB];
    long long tiles_all = 0;
    size_t lds_w = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        const size_t need = tasks[k].proj_w ? (size_t)D * 4 * D * 2 : 0;
        const size_t lay = (size_t)tasks[k].n_layers * (D * D * 2 + D * 4);
        lds_w = lds_w > need ? lds_w : need;
        lds_w = lds_w > lay ? lds_w : lay;
        cost[k] = ((long long)tasks[k].rows + 15) / 16 * (tasks[k].n_layers + (tasks[k].proj_w ? 5 : 0));
        tiles_all += ((long long)tasks[k].rows + 15) / 16;
    }
    tt.n = n;
    const size_t lds_bytes = lds_w + 16;
    if (lds_bytes > 160 * 1024) return fail(TSPGNN_EUNSUPPORTED, "mlp_fwd_bf16: %zu bytes of weights do not fit LDS", lds_bytes);
    int grid = n_cus();
    const long long max_grid = (tiles_all + NW - 1) / NW;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks_b(cost, n, grid, tt.blk_end);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_bf16_kernel<D, NW, PROJ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "mlp_fwd_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    mlp_fwd_bf16_kernel<D, NW, PROJ><<<grid, NW * 64, lds_bytes, st>>>(tt);
    return launched("tspgnn_mlp_fwd_multi_bf16");
}

template <int D, int NW, bool STAGED>
static int launch_lstm_b(const tspgnn_lstm_task_bf16* tasks, int n, hipStream_t st) {
    const size_t head = (10 * D + 4) * sizeof(float);
    const size_t per_kb = (size_t)32 * 4 * D * 2;
    const size_t budget = 156 * 1024 - head;
    LstmTableB tt;
    long long cost[kMaxTasksB];
    long long tiles_all = 0;
    size_t lds_w = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        const int KBT = (tasks[k].dx + D) / 32;
        int kbc = KBT;
        if ((size_t)KBT * per_kb > budget) kbc = (int)(budget / per_kb);
        if (kbc < 1) return fail(TSPGNN_EUNSUPPORTED, "lnlstm_fwd_bf16: d=%d does not fit LDS", D);
        tt.kbc[k] = kbc;
        if ((size_t)kbc * per_kb > lds_w) lds_w = (size_t)kbc * per_kb;
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        cost[k] = tiles * (KBT + 4) * (kbc < KBT ? 2 : 1);
        tiles_all += tiles;
    }
    tt.n = n;
    const size_t lds_bytes = lds_w + head;
    int grid = n_cus();
    const long long max_grid = (tiles_all + NW - 1) / NW;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks_b(cost, n, grid, tt.blk_end);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_fwd_bf16_kernel<D, NW, STAGED>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "lnlstm_fwd_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    lnlstm_fwd_bf16_kernel<D, NW, STAGED><<<grid, NW * 64, lds_bytes, st>>>(tt);
    return launched("tspgnn_lnlstm_fwd_multi_bf16");
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_gather2_sum_bf16(const int32_t* ev_uv, const void* X, void* Y, int M, int N, int d, void* stream) {
    TSPGNN_REQUIRE(M >= 0 && N >= 0, "gather2_sum_bf16: M=%d N=%d", M, N);
    TSPGNN_REQUIRE(d > 0 && d % 8 == 0, "gather2_sum_bf16: d=%d must be a positive multiple of 8", d);
    if (M == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(ev_uv && X && Y, "gather2_sum_bf16: null pointer");
    const int lpr = d / 8;
    long long blocks = ((long long)M * lpr + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    gather2_sum_bf16_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const int2*>(ev_uv), reinterpret_cast<const uint4*>(X), reinterpret_cast<uint4*>(Y), M, lpr);
    return launched("tspgnn_gather2_sum_bf16");
}

extern "C" int tspgnn_csr_rowsum_bf16(const int32_t* rowptr, const int32_t* eid, const void* X, void* Y, int N, int M,
                                      int d, void* stream) {
    TSPGNN_REQUIRE(N >= 0 && M >= 0, "csr_rowsum_bf16: N=%d M=%d", N, M);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128 || d == 256 || d == 512, "csr_rowsum_bf16: d=%d must be 32..512 (power of 2)", d);
    if (N == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(rowptr && eid && X && Y, "csr_rowsum_bf16: null pointer");
    const unsigned grid = (unsigned)(((long long)N * 64 + 255) / 256);
    hipStream_t st = as_stream(stream);
    const uint4* Xp = reinterpret_cast<const uint4*>(X);
    uint4* Yp = reinterpret_cast<uint4*>(Y);
    switch (d / 8) {
        case 4: csr_rowsum_bf16_kernel<4><<<grid, 256, 0, st>>>(rowptr, eid, Xp, Yp, N); break;
        case 8: csr_rowsum_bf16_kernel<8><<<grid, 256, 0, st>>>(rowptr, eid, Xp, Yp, N); break;
        case 16: csr_rowsum_bf16_kernel<16><<<grid, 256, 0, st>>>(rowptr, eid, Xp, Yp, N); break;
        case 32: csr_rowsum_bf16_kernel<32><<<grid, 256, 0, st>>>(rowptr, eid, Xp, Yp, N); break;
        default: csr_rowsum_bf16_kernel<64><<<grid, 256, 0, st>>>(rowptr, eid, Xp, Yp, N); break;
    }
    return launched("tspgnn_csr_rowsum_bf16");
}

extern "C" int tspgnn_mlp_fwd_multi_bf16(const tspgnn_mlp_task_bf16* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasksB, "mlp_fwd_multi_bf16: 1..%d tasks", kMaxTasksB);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "mlp_fwd_bf16: d=%d must be 32, 64 or 128", d);
    tspgnn_mlp_task_bf16 live[kMaxTasksB];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_mlp_task_bf16& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "mlp_fwd_bf16: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 4, "mlp_fwd_bf16: n_layers=%d must be in 1..4", t.n_layers);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE(t.X && t.wb && t.Y, "mlp_fwd_bf16: null pointer");
        TSPGNN_REQUIRE(!t.proj_w || t.proj_out, "mlp_fwd_bf16: projection needs proj_out");
        TSPGNN_REQUIRE(t.acts_stride >= 0, "mlp_fwd_bf16: acts_stride=%lld", t.acts_stride);
        TSPGNN_REQUIRE(!t.y_interleaved || (!t.proj_w && !t.acts), "mlp_fwd_bf16: y_interleaved excludes a projection and saved activations");
        live[n] = t;
        if (live[n].acts && live[n].acts_stride == 0) live[n].acts_stride = (long long)t.rows * d;
        ++n;
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    bool proj = false;
    for (int k = 0; k < n; ++k) proj = proj || live[k].proj_w != nullptr;
    if (d == 32) return proj ? launch_mlp_b<32, 8, true>(live, n, st) : launch_mlp_b<32, 8, false>(live, n, st);
    if (d == 64) return proj ? launch_mlp_b<64, 8, true>(live, n, st) : launch_mlp_b<64, 8, false>(live, n, st);
    return proj ? launch_mlp_b<128, 8, true>(live, n, st) : launch_mlp_b<128, 16, false>(live, n, st);
}

// development switch TSPGNN_BF16_CELL=0: the all-gates kernel (8 wavefronts) instead of the staged one
static bool bf16_cell_staged() {
    static const bool v = [] {
        const char* e = getenv("TSPGNN_BF16_CELL");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

extern "C" int tspgnn_lnlstm_fwd_multi_bf16(const tspgnn_lstm_task_bf16* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasksB, "lnlstm_fwd_multi_bf16: 1..%d tasks", kMaxTasksB);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "lnlstm_fwd_bf16: d=%d must be 32, 64 or 128", d);
    tspgnn_lstm_task_bf16 live[kMaxTasksB];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_lstm_task_bf16& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "lnlstm_fwd_bf16: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.dx >= 0 && t.dx % 32 == 0, "lnlstm_fwd_bf16: dx=%d must be a non-negative multiple of 32", t.dx);
        if (t.rows == 0) continue;
        // (c == NULL: the zero cell state of a run's first step, nothing is read)
        TSPGNN_REQUIRE(t.h && t.K && t.ln && t.h_out && t.c_out && (t.dx == 0 || t.x), "lnlstm_fwd_bf16: null pointer");
        TSPGNN_REQUIRE(t.h_out != t.h && t.c_out != t.c, "lnlstm_fwd_bf16: outputs may not alias inputs");
        TSPGNN_REQUIRE(!t.uv || (t.dx == 0 && t.Zx), "lnlstm_fwd_bf16: gather-init mode needs dx == 0 and Zx");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    // staged edge task (see lnlstm_fwd_bf16_kernel): d = 128 at two wavefronts per SIMD (256 registers: 289 -> 248 us per
    // C5-shard launch), d = 64 at three (144 registers: 25.9 -> 22.5 us per C2-sized launch)
    const bool staged = bf16_cell_staged();
    if (d == 32) return launch_lstm_b<32, 8, false>(live, n, st);
    if (d == 64) return staged ? launch_lstm_b<64, 12, true>(live, n, st) : launch_lstm_b<64, 8, false>(live, n, st);
    return staged ? launch_lstm_b<128, 8, true>(live, n, st) : launch_lstm_b<128, 8, false>(live, n, st);
}

extern "C" int tspgnn_convert_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
    TSPGNN_REQUIRE(n >= 0, "convert_f32_to_bf16: n=%lld", n);
    if (n == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(x && y, "convert_f32_to_bf16: null pointer");
    TSPGNN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "convert_f32_to_bf16: pointers must be 16-byte aligned");
    long long blocks = ((n >> 3) + 255) / 256;
    if (blocks > (1ll << 30)) blocks = 1ll << 30;
    if (blocks < 1) blocks = 1;
    f32_to_bf16_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(x, reinterpret_cast<__bf16*>(y), n);
    return launched("tspgnn_convert_f32_to_bf16");
}

extern "C" int tspgnn_convert_bf16_to_f32(const void* x, float* y, long long n, void* stream) {
    TSPGNN_REQUIRE(n >= 0, "convert_bf16_to_f32: n=%lld", n);
    if (n == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(x && y, "convert_bf16_to_f32: null pointer");
    TSPGNN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "convert_bf16_to_f32: pointers must be 16-byte aligned");
    long long blocks = ((n >> 3) + 255) / 256;   // (one 16-byte load per thread: a pure stream, no grid-stride cap)
    if (blocks > (1ll << 30)) blocks = 1ll << 30;
    if (blocks < 1) blocks = 1;
    bf16_to_f32_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const __bf16*>(x), y, n);
    return launched("tspgnn_convert_bf16_to_f32");
}
