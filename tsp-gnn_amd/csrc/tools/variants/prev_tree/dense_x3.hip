// fp32-accurate dense updates on the bf16 matrix cores ("bf16x3").
//
// On gfx950 the fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate and -- measured, see
// DESIGN.md §4.1 -- occupies the same pipe as the fp32 VALU, so the GEMMs of a step and its LayerNorm /
// gate epilogues serialise.  v_mfma_f32_16x16x32_bf16 is 16x faster and runs on the matrix pipe proper.
// Every fp32 operand is split exactly into three bf16 pieces, x = x1 + x2 + x3 (8+8+8 mantissa bits), and the
// product is formed from the six piece products whose weight is >= 2^-16 relative,
//     a*b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1),
// accumulated in fp32 inside the MFMA; the dropped terms are <= 2^-24 relative, i.e. fp32 rounding class, so
// the 1e-5 parity budget is untouched.  6 bf16 MFMAs (K=32 each) replace 8 fp32 MFMAs (K=4 each): 2.5x fewer
// matrix cycles AND the VALU epilogue of one wavefront now overlaps the MFMAs of its SIMD partner.
//
// Layout: the same "transposed chaining" as dense.hip (OUT^T = W^T IN^T, a wavefront owns 16 rows, the D
// fragment of one layer feeds the next layer's B operand without leaving the lane).  A k-block of the bf16
// MFMA covers 32 features: lane (rl, g) supplies the 8 features  16*(2kb + (j>>2)) + 4g + (j&3), j = 0..7,
// i.e. its registers of tiles 2kb and 2kb+1; the packed weights use the same permutation.
#include "common.h"
#include "mfma_tile.h"

namespace tspgnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// x = hi + mid + lo exactly (each piece a bf16, round-to-nearest-even)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[i] = h;
        mid[i] = m;
        lo[i] = (__bf16)r2;
    }
}

// Packed weights of a [krows, ncols] matrix: P[piece][kb][g][t][jl][8] (bf16),
//   value = piece(W[16*(2kb + (j>>2)) + 4g + (j&3)][t*16 + jl]),   KB = krows/32, NT = ncols/16.
// One ds_read_b128 per (piece, kb, t) and lane: 16 lanes x 16 B contiguous, lane groups a multiple of 256 B apart.
__global__ __launch_bounds__(256) void pack_weights_x3_kernel(const float* __restrict__ W, __bf16* __restrict__ P,
                                                              int krows, int ncols) {
    const int NT = ncols >> 4;
    const int total = krows * ncols;  // per piece
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, jl = (i >> 3) & 15;
        int rest = i >> 7;
        const int t = rest % NT;
        rest /= NT;
        const int g = rest & 3, kb = rest >> 2;
        const int k = 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3);
        const float x = W[(size_t)k * ncols + t * 16 + jl];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        P[i] = h;
        P[(size_t)total + i] = m;
        P[(size_t)2 * total + i] = (__bf16)r2;
    }
}

__device__ __forceinline__ bf16x8 ldw(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// acc[t] += W-block(kb, all NT tiles) x B for one 32-feature k-block of the packed matrix whose three pieces start
// at wh / wm / wl (LDS; wl may also be a global pointer: the lo piece feeds one MFMA in six and can stay in L1/L2
// when LDS is full).  (bh, bm, bl) = split3 of the lane's eight B values.
template <int NT>
__device__ __forceinline__ void kblock_p3(f32x4 (&acc)[NT], const __bf16* wh, const __bf16* wm, const __bf16* wl, int kb,
                                          int g, int jl, const bf16x8& bh, const bf16x8& bm, const bf16x8& bl) {
    const int off = ((kb * 4 + g) * NT * 16 + jl) * 8;
    // The weight fragments of tile t+1 are fetched while the six (dependent) MFMAs of tile t run: hi and mid into a
    // second register set, lo -- used by the first MFMA only -- back into its own register right after that MFMA.
    bf16x8 ah = ldw(wh + off), am = ldw(wm + off), al = ldw(wl + off);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bf16x8 nah = ah, nam = am;
        if (t + 1 < NT) {
            nah = ldw(wh + off + (t + 1) * 128);
            nam = ldw(wm + off + (t + 1) * 128);
        }
        f32x4 c = acc[t];
        c = MFMA_BF16(al, bh, c);  // smallest terms first
        if (t + 1 < NT) al = ldw(wl + off + (t + 1) * 128);
        c = MFMA_BF16(am, bm, c);
        c = MFMA_BF16(ah, bl, c);
        c = MFMA_BF16(am, bh, c);
        c = MFMA_BF16(ah, bm, c);
        c = MFMA_BF16(ah, bh, c);
        acc[t] = c;
        ah = nah;
        am = nam;
    }
}

// lds_w: packed matrix, piece-major, `total` = krows*ncols elements per piece.
template <int NT>
__device__ __forceinline__ void kblock_x3(f32x4 (&acc)[NT], const __bf16* lds_w, int total, int kb, int g, int jl,
                                          const float (&x)[8]) {
    bf16x8 bh, bm, bl;
    split3(x, bh, bm, bl);
    kblock_p3<NT>(acc, lds_w, lds_w + total, lds_w + 2 * (size_t)total, kb, g, jl, bh, bm, bl);
}

// One Dense(D) layer on the lane's part of a 16-row tile, activations chained in registers (D layout).
template <int D>
__device__ __forceinline__ void dense_layer_x3(f32x4 (&a)[D / 16], const __bf16* wh, const __bf16* wm, const __bf16* wl,
                                               const float* bias, bool relu, int g, int rl) {
    constexpr int NT = D / 16, KB = D / 32;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = ld4(bias + t * 16 + g * 4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a[2 * kb + (j >> 2)][j & 3];
        bf16x8 bh, bm, bl;
        split3(x, bh, bm, bl);
        kblock_p3<NT>(acc, wh, wm, wl, kb, g, rl, bh, bm, bl);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
        }
        a[t] = acc[t];
    }
}

// bytes -> LDS, 16 bytes per lane, straight from global memory (global_load_lds_dwordx4: no VGPR round trip, so
// every request of the stage is in flight at once).  The LDS address of a lane is the wavefront's base + lane*16:
// the base passed to the builtin must be the one of lane 0.  Callers follow up with stage_wait() + a barrier.
__device__ __forceinline__ void copy_bytes_to_lds(void* dst, const void* __restrict__ src, int nbytes, int tid,
                                                  int nthreads) {
    const int lane = tid & 63, n16 = nbytes >> 4;
    const char* s = reinterpret_cast<const char*>(src);
    char* d = reinterpret_cast<char*>(dst);
    for (int idx = tid; idx - lane < n16; idx += nthreads) {
        if (idx < n16)
            __builtin_amdgcn_global_load_lds(s + (size_t)idx * 16,
                                             (__attribute__((address_space(3))) void*)(d + (size_t)(idx - lane) * 16), 16, 0, 0);
    }
}
__device__ __forceinline__ void stage_wait() { __builtin_amdgcn_s_waitcnt(0); }

constexpr int kMaxTasks = 4;

// ---------------------------------------------------------------------------------- MLP (x3)
// Task fields as tspgnn_mlp_task; wb points at n_layers blocks of { bf16 packed[3*D*D] , float bias[D] };
// proj_w at a bf16 packed [3 * D * 4D] matrix.
struct MlpTaskTableX3 {
    tspgnn_mlp_task task[kMaxTasks];
    int blk_end[kMaxTasks];
    int n;
};

template <int D>
__global__ __launch_bounds__(1024) void mlp_fwd_x3_kernel(const MlpTaskTableX3 tt) {
    constexpr int NT = D / 16, KB = D / 32;
    constexpr int LAYER_BYTES = 3 * D * D * 2 + D * 4;
    constexpr int WBYTES = (4 * LAYER_BYTES > 3 * D * 4 * D * 2) ? 4 * LAYER_BYTES : 3 * D * 4 * D * 2;
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
    const __bf16* __restrict__ proj_w = reinterpret_cast<const __bf16*>(tt.task[k].proj_w);
    float* __restrict__ proj_out = tt.task[k].proj_out;
    const int tiles_total = (rows + 15) / 16;

    const int tid = threadIdx.x;
    copy_bytes_to_lds(lds, wb, n_layers * LAYER_BYTES, tid, blockDim.x);
    const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
    const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
    if (tid == 0) *ticket = t_beg;
    stage_wait();
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
            const __bf16* wl = reinterpret_cast<const __bf16*>(lds + (size_t)l * LAYER_BYTES);
            const float* bl = reinterpret_cast<const float*>(lds + (size_t)l * LAYER_BYTES + 3 * D * D * 2);
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = ld4(bl + t * 16 + g * 4);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = a[2 * kb + (j >> 2)][j & 3];
                kblock_x3<NT>(acc, wl, D * D, kb, g, rl, x);
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
    // second phase of a (small) task: proj_out = Y P, P packed [D, 4D] (see dense.hip)
    if (proj_w != nullptr) {
        constexpr int NP = D / 4;
        __threadfence_block();
        __syncthreads();
        copy_bytes_to_lds(lds, proj_w, 3 * D * 4 * D * 2, tid, blockDim.x);
        if (tid == 0) *ticket = t_beg;
        stage_wait();
        __syncthreads();
        const __bf16* wp = reinterpret_cast<const __bf16*>(lds);
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
                kblock_x3<NP>(acc, wp, D * 4 * D, kb, g, rl, x);
            }
            if (valid) {
#pragma unroll
                for (int t = 0; t < NP; ++t) st4(proj_out + rc * 4 * D + t * 16 + g * 4, acc[t]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- LN-LSTM (+ MLP) (x3)
// z = [x|h] K (+ gather-init / bias-init), five LayerNorms and the gate arithmetic of dense.hip's cell, with
// the GEMM on the bf16 matrix cores -- optionally followed, on the same 16 rows while h' is still in registers,
// by the message MLP that consumes h' in the NEXT time step (and its projection through the receiving cell's
// Kx): the step's "cell" and the next step's "message" launches become one, h' is not re-read from HBM and the
// per-launch fixed cost (weight staging, ramp, tail) is paid once.
//   resident mode  -- K fits LDS in three pieces (Kh of the edge cell in gather-init mode, 96 KB at D=64): K, the
//     MLP's hi/mid pieces and as many lo pieces as still fit stay in LDS for the lifetime of the workgroup (the
//     other lo pieces are read through L1: one MFMA operand in six); 16-row tiles are handed out by an LDS ticket.
//   lock-step mode -- a larger K (the vertex cell's [2D,4D]) is streamed in k-block chunks, then the MLP weights,
//     then the projection matrix are staged into the same LDS region, one tile per wavefront per round.
// Measured (tools/mfma_probe_bf16.hip): bf16 MFMAs and VALU instructions do not co-execute on a gfx950 SIMD
// either, so kernel time ~ MFMA cycles + VALU cycles + what HBM does not hide.
struct CellTaskTableX3 {
    tspgnn_cell_mlp_task task[kMaxTasks];
    int blk_end[kMaxTasks];
    int kbc[kMaxTasks];       // k-blocks (32 rows of K) per LDS chunk; >= all of K: resident
    int n_lo_lds[kMaxTasks];  // resident mode: MLP layers whose lo piece is in LDS
    int n;
};

template <int D>
__global__ __launch_bounds__(768) void lnlstm_mlp_fwd_x3_kernel(const CellTaskTableX3 tt) {
    constexpr int NT4 = D / 4, TPG = D / 16, KBH = D / 32;
    constexpr int LAYER_BYTES = 3 * D * D * 2 + D * 4;  // { hi, mid, lo, bias } of one MLP layer in global memory
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
    const __bf16* __restrict__ K = reinterpret_cast<const __bf16*>(tk.K);
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
    const __bf16* __restrict__ proj_w = reinterpret_cast<const __bf16*>(tt.task[k].proj_w);
    float* __restrict__ proj_out = tt.task[k].proj_out;
    const int tiles_total = (rows + 15) / 16;
    const int KBT = (dx + D) >> 5;       // k-blocks in total
    const int kbc = tt.kbc[k];
    const bool resident = kbc >= KBT;
    const int KBX = dx >> 5;             // k-blocks that come from x
    const int total = (dx + D) * 4 * D;  // elements per piece of the whole matrix
    const int chunk_total = (resident ? KBT : kbc) * 32 * 4 * D;

    // LDS: [ln 10*D floats][ticket, pad][weights region]
    float* lds_ln = reinterpret_cast<float*>(ldsb);
    int* ticket = reinterpret_cast<int*>(lds_ln + 10 * D);
    unsigned char* lds_wb = ldsb + (10 * D + 4) * sizeof(float);
    __bf16* lds_w = reinterpret_cast<__bf16*>(lds_wb);
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    for (int i = tid; i < 10 * D; i += blockDim.x) lds_ln[i] = ln[i];

    // stage k-blocks [kb0, kb1) of all three pieces (each piece is k-block major in global memory)
    auto stage = [&](int kb0, int kb1) {
        const int n = (kb1 - kb0) * 32 * 4 * D;  // elements per piece
        for (int p = 0; p < 3; ++p)
            copy_bytes_to_lds(lds_w + (size_t)p * chunk_total, K + (size_t)p * total + (size_t)kb0 * 32 * 4 * D, n * 2,
                              tid, blockDim.x);
    };
    // 32-bit element offsets from uniform base pointers (scalar base + vector offset addressing)
    auto init_acc = [&](f32x4 (&acc)[NT4], unsigned rc) {
        if (uv != nullptr) {
            const int2 ends = uv[rc];
            const float* zu = Zx + ((unsigned)ends.x * (4 * D) + g * 4);
            const float* zv = Zx + ((unsigned)ends.y * (4 * D) + g * 4);
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zu + t * 16);
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] += ld4(zv + t * 16);
        } else if (zbias != nullptr) {
            const float sc = zscale[rc];
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
        const float* hrow = h + (rc * D + g * 4);
        for (int kb = kb0; kb < kb1; ++kb) {
            const float* src = kb < KBX ? xrow + kb * 32 : hrow + (kb - KBX) * 32;
            const f32x4 lo4 = ld4(src), hi4 = ld4(src + 16);
            float xv[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            kblock_x3<NT4>(acc, lds_w, chunk_total, kb - kb_base, g, rl, xv);
        }
    };
    // gates + state stores; returns h' in registers (the D layout is the next GEMM's B operand)
    auto cell = [&](f32x4 (&acc)[NT4], f32x4 (&cf)[TPG], unsigned rc, bool valid, f32x4 (&hn)[TPG]) {
        f32x4 nc[TPG];
        lstm_gates<D>(acc, cf, lds_ln, g, hn, nc);
        if (valid) {
            float* hd = h_out + (rc * D + g * 4);
            float* cd = c_out + (rc * D + g * 4);
#pragma unroll
            for (int t = 0; t < TPG; ++t) {
                st4(hd + t * 16, hn[t]);
                st4(cd + t * 16, nc[t]);
            }
        }
    };

    if (resident) {
        // weights region: [K 3 pieces][per layer hi, mid][lo of the first n_lo layers][biases]
        const int n_lo = tt.n_lo_lds[k];
        __bf16* lds_hm = lds_w + (size_t)3 * chunk_total;
        __bf16* lds_lo = lds_hm + (size_t)n_layers * 2 * D * D;
        float* lds_bias = reinterpret_cast<float*>(lds_lo + (size_t)n_lo * D * D);
        stage(0, KBT);
        for (int l = 0; l < n_layers; ++l) {
            const unsigned char* src = mlp_wb + (size_t)l * LAYER_BYTES;
            copy_bytes_to_lds(lds_hm + (size_t)l * 2 * D * D, src, 2 * D * D * 2, tid, blockDim.x);
            if (l < n_lo) copy_bytes_to_lds(lds_lo + (size_t)l * D * D, src + 2 * D * D * 2, D * D * 2, tid, blockDim.x);
            copy_bytes_to_lds(lds_bias + l * D, src + 3 * D * D * 2, D * 4, tid, blockDim.x);
        }
        const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
        const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
        if (tid == 0) *ticket = t_beg;
        stage_wait();
        __syncthreads();
        for (;;) {
            int tile = 0;
            if (lane == 0) tile = atomicAdd(ticket, 1);
            tile = __builtin_amdgcn_readfirstlane(tile);
            if (tile >= t_end) break;
            const int row = tile * 16 + rl;
            const bool valid = row < rows;
            const unsigned rc = (unsigned)(valid ? row : rows - 1);
            f32x4 hn[TPG];
            {
                f32x4 acc[NT4], cf[TPG];
                init_acc(acc, rc);
#pragma unroll
                for (int t = 0; t < TPG; ++t) cf[t] = ld4(c + (rc * D + g * 4 + t * 16));
                kloop(acc, rc, 0, 0, KBT);
                cell(acc, cf, rc, valid, hn);
            }
            if (n_layers > 0) {
                for (int l = 0; l < n_layers; ++l) {
                    const __bf16* wh = lds_hm + (size_t)l * 2 * D * D;
                    const bool relu = (relu_mask >> l) & 1u;
                    if (l < n_lo)
                        dense_layer_x3<D>(hn, wh, wh + D * D, lds_lo + (size_t)l * D * D, lds_bias + l * D, relu, g, rl);
                    else
                        dense_layer_x3<D>(hn, wh, wh + D * D,
                                          reinterpret_cast<const __bf16*>(mlp_wb + (size_t)l * LAYER_BYTES) + 2 * D * D,
                                          lds_bias + l * D, relu, g, rl);
                }
                if (valid && mlp_out != nullptr) {
#pragma unroll
                    for (int t = 0; t < TPG; ++t) st4(mlp_out + (rc * D + g * 4 + t * 16), hn[t]);
                }
            }
        }
    } else {
        // lock-step rounds: one tile per wavefront; K walked chunk by chunk, then the MLP, then the projection
        const int rounds = (tiles_total + nw - 1) / nw;
        for (int r = my_blk; r < rounds; r += my_grid) {
            const int tile = r * nw + wave;
            const bool live = tile < tiles_total;
            const int row = tile * 16 + rl;
            const bool valid = live && row < rows;
            const unsigned rc = (unsigned)(valid ? row : rows - 1);
            f32x4 hn[TPG];
            {
                f32x4 acc[NT4], cf[TPG];
                init_acc(acc, rc);
#pragma unroll
                for (int t = 0; t < TPG; ++t) cf[t] = ld4(c + (rc * D + g * 4 + t * 16));
                for (int kb0 = 0; kb0 < KBT; kb0 += kbc) {
                    const int kb1 = min(KBT, kb0 + kbc);
                    __syncthreads();
                    stage(kb0, kb1);
                    stage_wait();
                    __syncthreads();
                    if (live) kloop(acc, rc, kb0, kb0, kb1);
                }
                cell(acc, cf, rc, valid, hn);
            }
            if (n_layers > 0) {
                __syncthreads();
                copy_bytes_to_lds(lds_wb, mlp_wb, n_layers * LAYER_BYTES, tid, blockDim.x);
                stage_wait();
                __syncthreads();
                for (int l = 0; l < n_layers; ++l) {
                    const __bf16* wh = reinterpret_cast<const __bf16*>(lds_wb + (size_t)l * LAYER_BYTES);
                    const float* bias = reinterpret_cast<const float*>(lds_wb + (size_t)l * LAYER_BYTES + 3 * D * D * 2);
                    dense_layer_x3<D>(hn, wh, wh + D * D, wh + 2 * D * D, bias, (relu_mask >> l) & 1u, g, rl);
                }
                if (valid && mlp_out != nullptr) {
#pragma unroll
                    for (int t = 0; t < TPG; ++t) st4(mlp_out + (rc * D + g * 4 + t * 16), hn[t]);
                }
                if (proj_w != nullptr) {  // proj_out = mlp(h') P, P packed [D, 4D]
                    __syncthreads();
                    copy_bytes_to_lds(lds_wb, proj_w, 3 * D * 4 * D * 2, tid, blockDim.x);
                    stage_wait();
                    __syncthreads();
                    f32x4 acc[NT4];
#pragma unroll
                    for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < KBH; ++kb) {
                        float xv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = hn[2 * kb + (j >> 2)][j & 3];
                        kblock_x3<NT4>(acc, lds_w, D * 4 * D, kb, g, rl, xv);
                    }
                    if (valid) {
#pragma unroll
                        for (int t = 0; t < NT4; ++t) st4(proj_out + (rc * (4 * D) + t * 16 + g * 4), acc[t]);
                    }
                }
            }
        }
    }
}

static int split_blocks_x3(const long long* cost, int n, int grid, int* blk_end) {
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
static int launch_mlp_x3(const tspgnn_mlp_task* tasks, int n, hipStream_t st) {
    MlpTaskTableX3 tt;
    long long cost[kMaxTasks];
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
    grid = split_blocks_x3(cost, n, grid, tt.blk_end);
    mlp_fwd_x3_kernel<D><<<grid, nw * 64, 0, st>>>(tt);
    return launched("tspgnn_mlp_fwd_multi_x3");
}

template <int D>
static int launch_cell_x3(const tspgnn_cell_mlp_task* tasks, int n, hipStream_t st, const char* what) {
    const size_t head = (10 * D + 4) * sizeof(float);
    const size_t per_kb = (size_t)3 * 32 * 4 * D * 2;  // bytes of one k-block, three pieces
    const size_t budget = 160 * 1024 - head;
    const size_t layer_hm = (size_t)2 * D * D * 2, layer_lo = (size_t)D * D * 2, layer_all = 3 * D * D * 2 + D * 4;
    CellTaskTableX3 tt;
    long long cost[kMaxTasks];
    long long tiles_all = 0;
    size_t lds_w = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        const tspgnn_lstm_task& c = tasks[k].cell;
        const int L = tasks[k].mlp_layers;
        const int KBT = (c.dx + D) / 32;
        const size_t k_bytes = (size_t)KBT * per_kb;
        const size_t mlp_min = L * (layer_hm + D * 4);  // resident: hi, mid and biases must be in LDS
        size_t need;
        if (k_bytes + mlp_min <= budget && !tasks[k].proj_w) {
            tt.kbc[k] = KBT;
            int n_lo = (int)((budget - k_bytes - mlp_min) / layer_lo);
            if (n_lo > L) n_lo = L;
            tt.n_lo_lds[k] = n_lo;
            need = k_bytes + mlp_min + n_lo * layer_lo;
        } else {
            int kbc = k_bytes <= budget ? KBT : (int)(budget / per_kb);
            if (kbc >= KBT) kbc = KBT - 1;  // lock-step mode is selected by kbc < KBT
            if (kbc < 1 || L * layer_all > budget || (tasks[k].proj_w && (size_t)3 * D * 4 * D * 2 > budget))
                return fail(TSPGNN_EUNSUPPORTED, "%s: dx=%d, d=%d, %d MLP layers do not fit LDS", what, c.dx, D, L);
            tt.kbc[k] = kbc;
            tt.n_lo_lds[k] = L;
            need = (size_t)kbc * per_kb;
            if (L * layer_all > need) need = L * layer_all;
            if (tasks[k].proj_w && (size_t)3 * D * 4 * D * 2 > need) need = (size_t)3 * D * 4 * D * 2;
        }
        if (need > lds_w) lds_w = need;
        const long long tiles = ((long long)c.rows + 15) / 16;
        cost[k] = tiles * (KBT * 4 + 2 * L + (tasks[k].proj_w ? 8 : 0) + 6);
        tiles_all += tiles;
    }
    tt.n = n;
    const size_t lds_bytes = lds_w + head;
    int grid = n_cus();
    const int nw = tiles_all <= (long long)grid * 4 ? 4 : (tiles_all <= (long long)grid * 8 ? 8 : 12);
    const long long max_grid = (tiles_all + nw - 1) / nw;
    if (grid > max_grid) grid = (int)max_grid;
    {
        // A lock-step task is a latency chain (several LDS re-stagings per round) that the resident tasks of the
        // launch hide: it gets exactly the workgroups of ONE round (more would idle, fewer would double the chain),
        // capped at half the grid; the resident tasks share the rest in proportion to their cost.
        int fixed[kMaxTasks], fixed_sum = 0, n_res = 0;
        long long res_cost[kMaxTasks];
        for (int k = 0; k < n; ++k) {
            const bool lock = tt.kbc[k] < (tasks[k].cell.dx + D) / 32;
            const long long tiles = ((long long)tasks[k].cell.rows + 15) / 16;
            fixed[k] = lock ? (int)((tiles + nw - 1) / nw) : 0;
            fixed_sum += fixed[k];
            if (!lock) ++n_res;
        }
        if (n_res == 0 || fixed_sum == 0 || fixed_sum > grid / 2) {
            grid = split_blocks_x3(cost, n, grid, tt.blk_end);
        } else {
            int res_end[kMaxTasks], j = 0;
            for (int k = 0; k < n; ++k)
                if (!fixed[k]) res_cost[j++] = cost[k];
            split_blocks_x3(res_cost, n_res, grid - fixed_sum, res_end);
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
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_mlp_fwd_x3_kernel<D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%d B): %s", what, (int)lds_bytes, hipGetErrorString(e));
    lnlstm_mlp_fwd_x3_kernel<D><<<grid, nw * 64, lds_bytes, st>>>(tt);
    return launched(what);
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_pack_weights_x3(const float* W, void* P, int krows, int ncols, void* stream) {
    TSPGNN_REQUIRE(krows >= 0 && krows % 32 == 0, "pack_weights_x3: krows=%d must be a multiple of 32", krows);
    TSPGNN_REQUIRE(ncols > 0 && ncols % 16 == 0, "pack_weights_x3: ncols=%d must be a multiple of 16", ncols);
    if (krows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(W && P, "pack_weights_x3: null pointer");
    int grid = (krows * ncols + 255) / 256;
    if (grid > 1024) grid = 1024;
    pack_weights_x3_kernel<<<grid, 256, 0, as_stream(stream)>>>(W, reinterpret_cast<__bf16*>(P), krows, ncols);
    return launched("tspgnn_pack_weights_x3");
}

extern "C" int tspgnn_mlp_fwd_multi_x3(const tspgnn_mlp_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "mlp_fwd_multi_x3: 1..%d tasks", kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64, "mlp_fwd_x3: d=%d must be 32 or 64", d);
    tspgnn_mlp_task live[kMaxTasks];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_mlp_task& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "mlp_fwd_x3: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.n_layers >= 1 && t.n_layers <= 4, "mlp_fwd_x3: n_layers=%d must be in 1..4", t.n_layers);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE(t.X && t.wb && t.Y, "mlp_fwd_x3: null pointer");
        TSPGNN_REQUIRE(!t.proj_w || t.proj_out, "mlp_fwd_x3: projection needs proj_out");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    return d == 32 ? launch_mlp_x3<32>(live, n, as_stream(stream)) : launch_mlp_x3<64>(live, n, as_stream(stream));
}

static int cell_mlp_x3(const tspgnn_cell_mlp_task* tasks, int n_tasks, int d, void* stream, const char* what) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "%s: 1..%d tasks", what, kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64, "%s: d=%d must be 32 or 64", what, d);
    tspgnn_cell_mlp_task live[kMaxTasks];
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
        TSPGNN_REQUIRE(t.h && t.c && t.K && t.ln && t.h_out && t.c_out && (t.dx == 0 || t.x), "%s: null pointer", what);
        TSPGNN_REQUIRE(t.h_out != t.h && t.c_out != t.c, "%s: outputs may not alias inputs", what);
        TSPGNN_REQUIRE(!tasks[k].state_in_blocked && !tasks[k].state_out_blocked, "%s: blocked states are an f16x2 feature", what);
        TSPGNN_REQUIRE(!tasks[k].mlp_acts, "%s: saving the MLP's hidden activations is an f16x2 feature", what);
        TSPGNN_REQUIRE(!t.uv || (t.dx == 0 && t.Zx), "%s: gather-init mode needs dx == 0 and Zx", what);
        TSPGNN_REQUIRE(!t.zbias || (t.zscale && !t.uv), "%s: zbias needs zscale and excludes gather-init mode", what);
        TSPGNN_REQUIRE(tasks[k].mlp_layers == 0 || tasks[k].mlp_wb, "%s: mlp_layers > 0 needs mlp_wb", what);
        TSPGNN_REQUIRE(!tasks[k].proj_w || (tasks[k].proj_out && tasks[k].mlp_layers > 0),
                       "%s: a projection needs proj_out and at least one MLP layer", what);
        live[n++] = tasks[k];
    }
    if (n == 0) return TSPGNN_OK;
    return d == 32 ? launch_cell_x3<32>(live, n, as_stream(stream), what) : launch_cell_x3<64>(live, n, as_stream(stream), what);
}

extern "C" int tspgnn_lnlstm_mlp_fwd_multi_x3(const tspgnn_cell_mlp_task* tasks, int n_tasks, int d, void* stream) {
    return cell_mlp_x3(tasks, n_tasks, d, stream, "tspgnn_lnlstm_mlp_fwd_multi_x3");
}

extern "C" int tspgnn_lnlstm_fwd_multi_x3(const tspgnn_lstm_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "lnlstm_fwd_multi_x3: 1..%d tasks", kMaxTasks);
    tspgnn_cell_mlp_task wrapped[kMaxTasks];
    for (int k = 0; k < n_tasks; ++k) {
        wrapped[k] = tspgnn_cell_mlp_task{};
        wrapped[k].cell = tasks[k];
    }
    return cell_mlp_x3(wrapped, n_tasks, d, stream, "tspgnn_lnlstm_fwd_multi_x3");
}
