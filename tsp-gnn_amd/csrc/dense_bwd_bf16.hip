// Backward kernels of the bf16-storage mode (BASELINE config 5; model.py:160-167 on graphnn.py:18's float_dtype) that read
// the bf16 TAPE directly -- h, cell inputs / projected messages as the forward stored them -- and multiply on the bf16
// matrix cores:
//   lnlstm_bwd_bf16   recomputes z = [x | h] K (or Zx[u] + Zx[v] + h Kh) with ONE v_mfma_f32_16x16x32_bf16 per product --
//                     both operands are bf16-exact, so this IS the z the forward normalised --, then the fp32 tile backward
//                     of lstm_bwd_tile.h; K (bf16, fragment order = piece 0 of tspgnn_pack_weights_x3) is resident in LDS
//                     where the fp32 kernel streams a 4x larger matrix in chunks (d = 128: Kh[128,512] = 128 KB);
//   linear_bf16w      Y = X W for an fp32 X (a gradient: dz) and a bf16-exact W (K^T): X = hi + lo in two bf16 pieces
//                     (16 significand bits; the gradients' own parity bar is 2e-3), two MFMAs per product.
// Gradients stay fp32 end to end: dz, dc, dh are fp32 arrays as in the fp32 mode.
#include "bf16_tile.h"
#include "common.h"
#include "lstm_bwd_tile.h"
#include "mfma_tile.h"

namespace tspgnn {

// x = hi + lo, hi = bf16(x) (round to nearest even), lo = bf16(x - hi): 16 significand bits
__device__ __forceinline__ void split2_bf16(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}

template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void lnlstm_bwd_bf16_kernel(const LstmBwdTaskTable tt) {
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
    const float* __restrict__ dh_out = tt.task[k].dh_out;
    const float* __restrict__ dc_out_in = tt.task[k].dc_out;
    float* __restrict__ dz = tt.task[k].dz;
    float* __restrict__ dc_in = tt.task[k].dc_in;
    float* __restrict__ ln_partial = tt.task[k].workspace;
    const int rows = tt.task[k].rows;
    const int tiles_total = (rows + 15) / 16;
    const int kbc = tt.qc[k];                    // k-blocks (32 rows of K) per LDS chunk; >= all of K: resident
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tt.task[k].uv);
    const __bf16* __restrict__ Zx = reinterpret_cast<const __bf16*>(tt.task[k].Zx);
    constexpr int NT4 = D / 4, TPG = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int KBX = dx >> 5, KBT = (dx + D) >> 5;
    const bool resident = kbc >= KBT;
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    // LDS: [K chunk (bf16)][ln 10*D][NW slabs of 10*D]
    __bf16* lds_k = reinterpret_cast<__bf16*>(ldsb);
    float* lds_ln = reinterpret_cast<float*>(ldsb + (size_t)(resident ? KBT : kbc) * 32 * 4 * D * 2);
    float* slabs = lds_ln + 10 * D;
    float* slab = slabs + wave * 10 * D;
    for (int i = tid; i < 10 * D; i += blockDim.x) lds_ln[i] = ln[i];
    for (int i = tid; i < nw * 10 * D; i += blockDim.x) slabs[i] = 0.f;

    auto stage = [&](int kb0, int kb1) {   // K is k-block major: one contiguous range
        copy_to_lds(reinterpret_cast<float*>(lds_k), reinterpret_cast<const float*>(K + (size_t)kb0 * 32 * 4 * D),
                    (kb1 - kb0) * 32 * 4 * D * 2 / 4, tid, blockDim.x);
    };
    auto init_acc = [&](f32x4 (&acc)[NT4], size_t rc) {
        if (uv != nullptr) {   // gather-init mode: the bf16 projected messages as the forward's projection wrote them
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
        const __bf16* hrow = h + rc * D + g * 4;
        for (int kb = kb0; kb < kb1; ++kb) {
            const bf16x8 bv = kb < KBX ? row_operand(xrow, kb) : row_operand(hrow, kb - KBX);
            const __bf16* base = lds_k + ((size_t)((kb - kb_base) * 4 + g) * NT4 * 16 + rl) * 8;
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = MFMA_BF16(ldw8(base + t * 128), bv, acc[t]);
        }
    };
    auto finish = [&](f32x4 (&acc)[NT4], size_t rc, bool valid, const f32x4 (&cf)[TPG], const f32x4 (&dhn)[TPG],
                      const f32x4 (&dcn)[TPG]) {
        f32x4 dco[TPG];
        lstm_tile_backward<D, true>(acc, cf, dhn, dcn, dco, lds_ln, slab, g, rl, valid);
        if (valid) {
            const size_t o = rc * D + g * 4;
#pragma unroll
            for (int t = 0; t < NT4; ++t) st4(dz + rc * 4 * D + t * 16 + g * 4, acc[t]);
#pragma unroll
            for (int t = 0; t < TPG; ++t) st4(dc_in + o + t * 16, dco[t]);
        }
    };

    if (resident) {
        stage(0, KBT);
        const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
        const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
        __syncthreads();
        // static round-robin over the workgroup's tiles (not a ticket): which tiles a wavefront sums into its
        // LayerNorm-gradient slab must not depend on timing, or the gradients differ in the last bit from run to run
        for (int tile = t_beg + wave; tile < t_end; tile += nw) {
            const int row = tile * 16 + rl;
            const bool valid = row < rows;
            const size_t rc = (size_t)(valid ? row : rows - 1);
            f32x4 acc[NT4];
            init_acc(acc, rc);
            // c, dh', dc' of the tile: issued here so that their latency hides behind the GEMM
            const size_t o = rc * D + g * 4;
            f32x4 cf[TPG], dhn[TPG], dcn[TPG];
            lstm_tile_load<D>(c + o, dh_out ? dh_out + o : nullptr, dc_out_in ? dc_out_in + o : nullptr, cf, dhn, dcn);
            kloop(acc, rc, 0, 0, KBT);
            finish(acc, rc, valid, cf, dhn, dcn);
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
            init_acc(acc, rc);
            for (int kb0 = 0; kb0 < KBT; kb0 += kbc) {
                const int kb1 = min(KBT, kb0 + kbc);
                __syncthreads();
                stage(kb0, kb1);
                __syncthreads();
                if (live) kloop(acc, rc, kb0, kb0, kb1);
            }
            const size_t o = rc * D + g * 4;
            f32x4 cf[TPG], dhn[TPG], dcn[TPG];
            lstm_tile_load<D>(c + o, dh_out ? dh_out + o : nullptr, dc_out_in ? dc_out_in + o : nullptr, cf, dhn, dcn);
            finish(acc, rc, valid, cf, dhn, dcn);
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

static int split_blocks_bwd_b(const long long* cost, int n, int grid, int* blk_end) {
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
static int launch_lnlstm_bwd_bf16(const tspgnn_lstm_bwd_task* tasks, int n, hipStream_t st) {
    // D=128 keeps 4D/16 + temporaries > 256 registers live: one wavefront per SIMD (512-register budget)
    constexpr int NWMAX = D >= 128 ? 4 : 8;
    const size_t tail = (size_t)(10 * D + NWMAX * 10 * D + 4) * sizeof(float);
    const size_t per_kb = (size_t)32 * 4 * D * 2;
    const size_t budget = 160 * 1024 - tail;
    LstmBwdTaskTable tt;
    long long cost[kMaxTasks];
    long long tiles_all = 0;
    size_t lds_k = 0;
    int grid_cap = 1 << 30;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        const int KBT = (tasks[k].dx + D) / 32;
        int kbc = KBT;
        if ((size_t)KBT * per_kb > budget) kbc = (int)(budget / per_kb);
        if (kbc < 1) return fail(TSPGNN_EUNSUPPORTED, "lnlstm_bwd_bf16: d=%d does not fit LDS", D);
        tt.qc[k] = kbc;
        if ((size_t)kbc * per_kb > lds_k) lds_k = (size_t)kbc * per_kb;
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        cost[k] = tiles * (KBT + 12) * (kbc < KBT ? 2 : 1);
        tiles_all += tiles;
        (void)grid_cap;
    }
    tt.n = n;
    const size_t lds_bytes = lds_k + tail;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_bwd_bf16_kernel<D, NWMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "lnlstm_bwd_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int grid = n_cus();
    const long long max_grid = (tiles_all + NWMAX - 1) / NWMAX;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks_bwd_b(cost, n, grid, tt.blk_end);
    // (the workspace holds one row of LayerNorm-gradient partials per workgroup of a task: n_cus + 8 rows,
    // tspgnn_lnlstm_bwd_workspace_floats)
    lnlstm_bwd_bf16_kernel<D, NWMAX><<<grid, NWMAX * 64, lds_bytes, st>>>(tt);
    int rc = launched("tspgnn_lnlstm_bwd_multi_bf16");
    if (rc) return rc;
    for (int k = 0; k < n; ++k) {
        if (tasks[k].defer_reduce) continue;
        const int nblk = tt.blk_end[k] - (k ? tt.blk_end[k - 1] : 0);
        reduce_partials(tasks[k].workspace, nblk, 10 * D, tasks[k].ln_grad, 10 * D, 1.0f, 1, st);
        if ((rc = launched("tspgnn_lnlstm_bwd_multi_bf16(reduce)"))) return rc;
    }
    return TSPGNN_OK;
}

// ------------------------------------------------------------------------------------ Y = X W, W bf16 (fragment order)
// W: the bf16 packing (piece 0 of tspgnn_pack_weights_x3) of a [kin, nout_total] matrix; this launch forms the NT output
// tiles t0 .. t0+NT-1 (16 columns each) of it: LDS holds exactly those fragments (kin * NT * 16 bf16).  Column c of the
// result goes to Y1[:, c] for c < n1, else to Y2[:, c - n1] (optionally accumulated), like tspgnn_linear_f32.
template <int NT>
__global__ __launch_bounds__(512) void linear_bf16w_kernel(const float* __restrict__ X, int kin, const __bf16* __restrict__ Wp,
                                                           int nt_total, int t0, float* __restrict__ Y1, int n1,
                                                           float* __restrict__ Y2, int n2, int acc2, int rows, int tiles_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    __bf16* lds_w = reinterpret_cast<__bf16*>(ldsb);
    const int KB = kin >> 5;
    int* ticket = reinterpret_cast<int*>(ldsb + (size_t)kin * NT * 16 * 2);
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4;
    // fragment rows (kb, g): NT * 128 bf16 each, at ((kb*4+g) * nt_total + t0) * 128 in the full packing
    for (int r = (tid >> 6); r < KB * 4; r += (int)(blockDim.x >> 6)) {
        const __bf16* src = Wp + ((size_t)r * nt_total + t0) * 128;
        __bf16* dst = lds_w + (size_t)r * NT * 128;
        for (int i = lane; i < NT * 16; i += 64)   // 16-byte pieces
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(src) + (size_t)i * 16,
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(dst) + (size_t)(i - lane) * 16),
                                             16, 0, 0);
    }
    const int t_beg = (int)((long long)tiles_total * blockIdx.x / gridDim.x);
    const int t_end = (int)((long long)tiles_total * (blockIdx.x + 1) / gridDim.x);
    if (tid == 0) *ticket = t_beg;
    __builtin_amdgcn_s_waitcnt(0);
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
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 lo4 = ld4(xr + kb * 32), hi4 = ld4(xr + kb * 32 + 16);
            float xv[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            bf16x8 bh, bl;
            split2_bf16(xv, bh, bl);
            const __bf16* base = lds_w + ((size_t)(kb * 4 + g) * NT * 16 + rl) * 8;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 w = ldw8(base + t * 128);
                f32x4 d = acc[t];
                d = MFMA_BF16(w, bl, d);
                d = MFMA_BF16(w, bh, d);
                acc[t] = d;
            }
        }
        if (valid) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = (t0 + t) * 16;
                if (col < n1) {
                    st4(Y1 + rc * n1 + col + g * 4, acc[t]);
                } else {
                    float* p = Y2 + rc * n2 + (col - n1) + g * 4;
                    st4(p, acc2 ? ld4(p) + acc[t] : acc[t]);
                }
            }
        }
    }
}

template <int NT>
static int launch_linear_bf16w(const float* X, int kin, const __bf16* Wp, int nt_total, int t0, float* Y1, int n1, float* Y2,
                               int n2, int acc2, int rows, hipStream_t st) {
    const int tiles = (rows + 15) / 16;
    const size_t lds_bytes = (size_t)kin * NT * 16 * 2 + 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bf16w_kernel<NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "linear_bf16w: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int grid = n_cus() * (lds_bytes > 80 * 1024 ? 1 : 2);
    const int nw = 8;
    const int max_grid = (tiles + nw - 1) / nw;
    if (grid > max_grid) grid = max_grid;
    linear_bf16w_kernel<NT><<<grid, nw * 64, lds_bytes, st>>>(X, kin, Wp, nt_total, t0, Y1, n1, Y2, n2, acc2, rows, tiles);
    return launched("tspgnn_linear_bf16w_f32");
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_lnlstm_bwd_multi_bf16(const tspgnn_lstm_bwd_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "lnlstm_bwd_multi_bf16: 1..%d tasks", kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64 || d == 128, "lnlstm_bwd_bf16: d=%d must be 32, 64 or 128", d);
    tspgnn_lstm_bwd_task live[kMaxTasks];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_lstm_bwd_task& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "lnlstm_bwd_bf16: rows=%d", t.rows);
        TSPGNN_REQUIRE(t.dx >= 0 && t.dx % 32 == 0, "lnlstm_bwd_bf16: dx=%d must be a non-negative multiple of 32", t.dx);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE(t.h && t.c && t.K && t.ln && t.dz && t.dc_in && t.ln_grad && t.workspace && (t.dx == 0 || t.x),
                       "lnlstm_bwd_bf16: null pointer");
        TSPGNN_REQUIRE(!t.uv || (t.dx == 0 && t.Zx), "lnlstm_bwd_bf16: gather-init mode needs dx == 0 and Zx");
        TSPGNN_REQUIRE(!t.KT && !t.dxh && !t.zbias && !t.KTg, "lnlstm_bwd_bf16: no fused data gradient / bias-init in this mode");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    if (d == 32) return launch_lnlstm_bwd_bf16<32>(live, n, st);
    if (d == 64) return launch_lnlstm_bwd_bf16<64>(live, n, st);
    return launch_lnlstm_bwd_bf16<128>(live, n, st);
}

extern "C" int tspgnn_linear_bf16w_f32(const float* X, int kin, const void* Wp, float* Y1, int n1, float* Y2, int n2,
                                       int accumulate_y2, int rows, void* stream) {
    TSPGNN_REQUIRE(rows >= 0, "linear_bf16w: rows=%d", rows);
    TSPGNN_REQUIRE(kin > 0 && kin % 32 == 0 && kin <= 1024, "linear_bf16w: kin=%d must be a multiple of 32 up to 1024", kin);
    TSPGNN_REQUIRE(n1 >= 0 && n2 >= 0 && n1 % 16 == 0 && n2 % 16 == 0 && n1 + n2 > 0, "linear_bf16w: n1=%d n2=%d", n1, n2);
    if (rows == 0) return TSPGNN_OK;
    TSPGNN_REQUIRE(X && Wp && (n1 == 0 || Y1) && (n2 == 0 || Y2), "linear_bf16w: null pointer");
    hipStream_t st = as_stream(stream);
    const __bf16* W = reinterpret_cast<const __bf16*>(Wp);
    const int nt_total = (n1 + n2) / 16;
    // column blocks of at most 8 tiles (kin = 512: 128 KB of fragments per block), 4 / 2 / 1 for the remainder
    int t0 = 0;
    while (t0 < nt_total) {
        const int left = nt_total - t0;
        int rc;
        if (left >= 8 && (size_t)kin * 8 * 16 * 2 <= 150 * 1024) {
            rc = launch_linear_bf16w<8>(X, kin, W, nt_total, t0, Y1, n1, Y2, n2, accumulate_y2, rows, st);
            t0 += 8;
        } else if (left >= 4) {
            rc = launch_linear_bf16w<4>(X, kin, W, nt_total, t0, Y1, n1, Y2, n2, accumulate_y2, rows, st);
            t0 += 4;
        } else if (left >= 2) {
            rc = launch_linear_bf16w<2>(X, kin, W, nt_total, t0, Y1, n1, Y2, n2, accumulate_y2, rows, st);
            t0 += 2;
        } else {
            rc = launch_linear_bf16w<1>(X, kin, W, nt_total, t0, Y1, n1, Y2, n2, accumulate_y2, rows, st);
            t0 += 1;
        }
        if (rc) return rc;
    }
    return TSPGNN_OK;
}
