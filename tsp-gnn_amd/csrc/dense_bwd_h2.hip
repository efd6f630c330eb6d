// LN-LSTM backward with its two GEMMs on the fp16 matrix cores ("f16x2", see dense_h2.hip).
//
// The fp32-MFMA backward (dense_bwd.hip) spends 2 x 256 v_mfma_f32_16x16x4_f32 per 16-row tile -- 16 k cycles of a
// SIMD, the largest item of the training step -- on recomputing z = [x|h] K and on dh = dz Kh^T.  Here both run as
// three fp16 piece products per k-block: 96 + 96 v_mfma_f32_16x16x32_f16, 3 k cycles.
//   * z is recomputed exactly as the f16x2 forward forms it (same packed 2^s K, same split of h, the projected
//     messages Zx carrying the factor 2^s), so the LayerNorm statistics the backward differentiates are the forward's
//     own; the gate LayerNorms run with epsilon 2^2s * 1e-12 and the tile backward (lstm_bwd_tile.h) hands back the
//     gradient w.r.t. the SCALED pre-activation, dz' = dz / 2^s.  dz is stored as 2^s * dz' (exact).
//   * dh = dz Kh^T = dz' (2^s Kh)^T: the factor cancels against the packed weights.  Gradients span many binades
//     (1e-8 .. 1e-3 across rows and time steps), far outside fp16's normal range, so every row of dz' is brought to
//     [0.5, 1) by the power of two of its largest entry before the split, and the split's second piece is scaled up by
//     2^11 into fp16's normal range and accumulated apart (split2s / kblock_h2_side, h2_tile.h: the plain lo piece goes
//     subnormal and then quantises every small entry at 2^-25 of the ROW's maximum -- measured at full C2 size as 5-6x
//     the fp32-MFMA backward's error on the bias / LayerNorm-shift gradients, which sum 10^5 such rows); the 16 outputs
//     of the row are scaled back.
#include "common.h"
#include "h2_tile.h"
#include "lstm_bwd_tile.h"
#include "mfma_tile.h"

namespace tspgnn {

template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void lnlstm_bwd_h2_kernel(const LstmBwdTaskTable tt) {
    int k = 0;
    while (k + 1 < tt.n && (int)blockIdx.x >= tt.blk_end[k]) ++k;
    const int blk0 = k ? tt.blk_end[k - 1] : 0;
    const int my_blk = blockIdx.x - blk0, my_grid = tt.blk_end[k] - blk0;
    const float* __restrict__ x = tt.task[k].x;
    const int dx = tt.task[k].dx;
    const float* __restrict__ h = tt.task[k].h;
    const float* __restrict__ c = tt.task[k].c;
    const _Float16* __restrict__ K = reinterpret_cast<const _Float16*>(tt.task[k].K);
    const float* __restrict__ ln = tt.task[k].ln;
    const float* __restrict__ dh_out = tt.task[k].dh_out;
    const float* __restrict__ dc_out_in = tt.task[k].dc_out;
    float* __restrict__ dz = tt.task[k].dz;
    float* __restrict__ dc_in = tt.task[k].dc_in;
    float* __restrict__ ln_partial = tt.task[k].workspace;
    const int rows = tt.task[k].rows;
    const int tiles_total = (rows + 15) / 16;
    const int2* __restrict__ uv = reinterpret_cast<const int2*>(tt.task[k].uv);
    const float* __restrict__ Zx = tt.task[k].Zx;
    const _Float16* __restrict__ KT = reinterpret_cast<const _Float16*>(tt.task[k].KT);
    float* __restrict__ dxh = tt.task[k].dxh;
    const float* __restrict__ zbias = tt.task[k].zbias;
    const float* __restrict__ zscale = tt.task[k].zscale;
    const _Float16* __restrict__ KTg = reinterpret_cast<const _Float16*>(tt.task[k].KTg);
    float* __restrict__ dxg = tt.task[k].dxg;
    constexpr int NT4 = D / 4, TPG = D / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int KBX = dx >> 5;
    const int k_total = (dx + D) * 4 * D;       // elements per piece of K
    const int kt_total = 4 * D * D;             // elements per piece of Kh^T ([4D, D])
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    // LDS: [K hi, lo][Kh^T hi, lo (optional)][ln 10*D][NW slabs of 10*D]
    _Float16* lds_k = reinterpret_cast<_Float16*>(ldsb);
    _Float16* lds_kt = lds_k + (size_t)2 * k_total;
    float* lds_ln = reinterpret_cast<float*>(lds_kt + (KT != nullptr ? (size_t)2 * kt_total : 0));
    float* slabs = lds_ln + 10 * D;
    float* slab = slabs + wave * 10 * D;
    h2_copy_to_lds(lds_k, K, 2 * k_total * 2, tid, blockDim.x);
    if (KT != nullptr) h2_copy_to_lds(lds_kt, KT, 2 * kt_total * 2, tid, blockDim.x);
    for (int i = tid; i < 10 * D; i += blockDim.x) lds_ln[i] = ln[i];
    for (int i = tid; i < nw * 10 * D; i += blockDim.x) slabs[i] = 0.f;
    h2_stage_wait();
    __syncthreads();

    const int t_beg = (int)((long long)tiles_total * my_blk / my_grid);
    const int t_end = (int)((long long)tiles_total * (my_blk + 1) / my_grid);
    // static round-robin over the workgroup's tiles (not a ticket): which tiles a wavefront sums into its
    // LayerNorm-gradient slab must not depend on timing, or the gradients differ in the last bit from run to run.
    //
    // The tile loop is software-pipelined: with ~215 registers a SIMD holds two wavefronts, so a tile's memory round trips
    // (endpoints -> projected messages -> h rows) are not hidden by other wavefronts (traced: 3.4 + 4.5 of a tile's 26 us
    // were those waits).  A tile therefore arrives at the top of its iteration with its operands in flight or in registers:
    // the endpoints of the NEXT tile are fetched at the top of this one, its h rows and its
    // projected messages Zx[u] (into acc) and Zx[v] (into zvn) behind the k-blocks of this tile's dh GEMM, as the
    // registers of dz are consumed.
    constexpr int KBH = D / 32;
    f32x4 acc[NT4], zvn[NT4], hpre[2 * KBH];
    const bool gather = uv != nullptr;
    int tile = t_beg + wave;
    unsigned rc_n = 0;
    int2 ends_n = {0, 0};
#pragma unroll
    for (int t = 0; t < NT4; ++t) acc[t] = zvn[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2 * KBH; ++i) hpre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tile < t_end) {
        const int row = tile * 16 + rl;
        rc_n = (unsigned)(row < rows ? row : rows - 1);
        if (gather) {
            const int2 ends = uv[rc_n];
            const float* zu = Zx + h2_zx_row<D>((unsigned)ends.x, g);
            const float* zv = Zx + h2_zx_row<D>((unsigned)ends.y, g);
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zu + t * 256);
#pragma unroll
            for (int t = 0; t < NT4; ++t) zvn[t] = ld4(zv + t * 256);
        }
        const float* hrow = h + (rc_n * D + g * 4);
#pragma unroll
        for (int kb = 0; kb < KBH; ++kb) {
            hpre[2 * kb] = ld4(hrow + kb * 32);
            hpre[2 * kb + 1] = ld4(hrow + kb * 32 + 16);
        }
    }
    for (; tile < t_end; tile += nw) {
        const unsigned rc = rc_n;
        const bool valid = tile * 16 + rl < rows;
        const bool has_n = tile + nw < t_end;   // (wavefront-uniform)
        if (has_n) {
            const int row_n = (tile + nw) * 16 + rl;
            rc_n = (unsigned)(row_n < rows ? row_n : rows - 1);
            if (gather) ends_n = uv[rc_n];
        }
        if (gather) {  // gather-init mode: Zx as the f16x2 projection wrote it (2^s Zx, blocked by 16 rows)
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] += zvn[t];
        } else if (zbias != nullptr) {  // the forward's bias-init: z starts at 2^s * zscale[row] * zbias
            const float sc = zscale[rc] * kH2Scale;
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zbias + t * 16 + g * 4) * sc;
        } else {
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // c, dh', dc' of the tile: issued here so that their latency hides behind the GEMM
        const unsigned o = rc * D + g * 4;
        f32x4 cf[TPG], dhn[TPG], dcn[TPG];
        lstm_tile_load<D>(c + o, dh_out ? dh_out + o : nullptr, dc_out_in ? dc_out_in + o : nullptr, cf, dhn, dcn);
        {
            const float* xrow = x + (rc * (unsigned)dx + g * 4);
            for (int kb = 0; kb < KBX; ++kb) {
                const f32x4 lo4 = ld4(xrow + kb * 32), hi4 = ld4(xrow + kb * 32 + 16);
                float xv[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                f16x8 bh, bl;
                split2(xv, bh, bl);
                kblock_h2<NT4>(acc, lds_k, lds_k + k_total, kb, g, rl, bh, bl);
            }
#pragma unroll
            for (int kb = 0; kb < KBH; ++kb) {
                const f32x4 lo4 = hpre[2 * kb], hi4 = hpre[2 * kb + 1];
                float xv[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                f16x8 bh, bl;
                split2(xv, bh, bl);
                kblock_h2<NT4>(acc, lds_k, lds_k + k_total, KBX + kb, g, rl, bh, bl);
            }
        }
        f32x4 dco[TPG];
        lstm_tile_backward<D, true>(acc, cf, dhn, dcn, dco, lds_ln, slab, g, rl, valid, kH2GateEps);
        if (valid) {
            const f32x2 sc2 = {kH2Scale, kH2Scale};
#pragma unroll
            for (int t = 0; t < NT4; ++t) {
                f32x4 v;
                v.lo = acc[t].lo * sc2;
                v.hi = acc[t].hi * sc2;
                st4(dz + (size_t)rc * 4 * D + t * 16 + g * 4, v);
            }
#pragma unroll
            for (int t = 0; t < TPG; ++t) st4(dc_in + o + t * 16, dco[t]);
        }
        {  // the next tile's h rows fly behind this tile's dh GEMM (the last tile re-reads its own)
            const float* hrow = h + (rc_n * D + g * 4);
#pragma unroll
            for (int kb = 0; kb < KBH; ++kb) {
                hpre[2 * kb] = ld4(hrow + kb * 32);
                hpre[2 * kb + 1] = ld4(hrow + kb * 32 + 16);
            }
        }
        const bool pre = gather && has_n;
        const float* zu_n = Zx + h2_zx_row<D>((unsigned)ends_n.x, g);
        const float* zv_n = Zx + h2_zx_row<D>((unsigned)ends_n.y, g);
        if (KT != nullptr) {
            // dh = dz' (2^s Kh)^T with the row of dz' normalised to [0.5, 1) by a power of two
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < NT4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, __builtin_fabsf(acc[t][r]));
            }
            m = max_over_lane_groups16_swap(m);
            const int e = h2_row_exponent(m);   // m = f * 2^e, f in [0.5, 1)
            const float up = __builtin_ldexpf(1.0f, -e), down = __builtin_ldexpf(1.0f, e);
            // (the row's second piece is the SCALED one of split2s, accumulated apart: entries far below the row's
            // largest keep a relative, not an absolute, accuracy)
            f32x4 out[TPG], side[TPG];
#pragma unroll
            for (int t = 0; t < TPG; ++t) out[t] = side[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NT4 / 2; ++kb) {
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = acc[2 * kb + (j >> 2)][j & 3] * up;
                f16x8 bh, bm;
                split2s(xv, bh, bm);
                if (pre) {  // the two dz tiles just consumed make room for the next tile's projected messages
                    acc[2 * kb] = ld4(zu_n + (2 * kb) * 256);
                    acc[2 * kb + 1] = ld4(zu_n + (2 * kb + 1) * 256);
                    zvn[2 * kb] = ld4(zv_n + (2 * kb) * 256);
                    zvn[2 * kb + 1] = ld4(zv_n + (2 * kb + 1) * 256);
                } else {    // (defined on every path: nothing of the old contents stays live across the tile)
                    acc[2 * kb] = acc[2 * kb + 1] = zvn[2 * kb] = zvn[2 * kb + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                kblock_h2_side<TPG>(out, side, lds_kt, lds_kt + kt_total, kb, g, rl, bh, bm);
            }
            if (valid) {
                const float fold = 1.0f / 2048.0f;
#pragma unroll
                for (int t = 0; t < TPG; ++t) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(side[t][r], fold, out[t][r]) * down;
                    st4(dxh + (size_t)rc * D + t * 16 + g * 4, v);
                }
            }
        } else if (pre) {
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = ld4(zu_n + t * 256);
#pragma unroll
            for (int t = 0; t < NT4; ++t) zvn[t] = ld4(zv_n + t * 256);
        } else {
#pragma unroll
            for (int t = 0; t < NT4; ++t) acc[t] = zvn[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // Second phase of a task with KTg (tspgnn_lstm_bwd_task: d == dx == 64): [dxg | dxh] = dz K^T for a cell whose K^T
    // ([4D, 2D] in two pieces, 128 KB) has no room beside K -- so it takes K's place once the workgroup's tiles are through,
    // and the tiles' dz rows (just written, L2-hot) come back for the second GEMM.  Few rows (the vertex cell: 320 tiles at
    // C2); the launch this replaces (tspgnn_linear_f32 on dz) cost ~11 us of launch boundary, staging and tail per step.
    if (KTg != nullptr) {
        if constexpr (D == 64) {
            __syncthreads();    // every wavefront is done with K (and its dz rows are on their way: same-lane program order)
            h2_copy_to_lds(lds_k, KTg, 4 * D * 2 * D * 4, tid, blockDim.x);
            h2_stage_wait();
            __syncthreads();
            const _Float16* gh = lds_k;
            const _Float16* gl = lds_k + (size_t)4 * D * 2 * D;
            for (int t2 = t_beg + wave; t2 < t_end; t2 += nw) {
                const int row = t2 * 16 + rl;
                const bool valid = row < rows;
                const unsigned rc = (unsigned)(valid ? row : rows - 1);
                // dz was stored as 2^s dz' (this lane wrote exactly the elements it reads back)
                f32x4 dzr[NT4];
                float m = 0.f;
#pragma unroll
                for (int t = 0; t < NT4; ++t) {
                    dzr[t] = ld4(dz + (size_t)rc * 4 * D + t * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, __builtin_fabsf(dzr[t][r]));
                }
                m = max_over_lane_groups16_swap(m);
                const int e = h2_row_exponent(m);
                const float up = __builtin_ldexpf(1.0f, -e), down = __builtin_ldexpf(kH2InvScale, e);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    f32x4 out[TPG], side[TPG];
#pragma unroll
                    for (int t = 0; t < TPG; ++t) out[t] = side[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < NT4 / 2; ++kb) {
                        float xv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = dzr[2 * kb + (j >> 2)][j & 3] * up;
                        f16x8 bh, bm;
                        split2s(xv, bh, bm);
                        if (pass == 0) kblock_h2_side_sub<2 * TPG, 0, TPG>(out, side, gh, gl, kb, g, rl, bh, bm);
                        else kblock_h2_side_sub<2 * TPG, TPG, TPG>(out, side, gh, gl, kb, g, rl, bh, bm);
                    }
                    if (valid) {
                        const float fold = 1.0f / 2048.0f;
                        float* dst = pass == 0 ? dxg + (size_t)rc * D : dxh + (size_t)rc * D;
#pragma unroll
                        for (int t = 0; t < TPG; ++t) {
                            f32x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaf(side[t][r], fold, out[t][r]) * down;
                            st4(dst + t * 16 + g * 4, v);
                        }
                    }
                }
            }
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

static int split_blocks_bwd_h2(const long long* cost, int n, int grid, int* blk_end) {
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
static int launch_lnlstm_bwd_h2(const tspgnn_lstm_bwd_task* tasks, int n, hipStream_t st) {
#ifndef H2_BWD_NW
#define H2_BWD_NW 8
#endif
    constexpr int NWMAX = H2_BWD_NW;
    auto extra = [&](int nw_) { return (size_t)(10 * D + nw_ * 10 * D + 4) * sizeof(float); };
    LstmBwdTaskTable tt;
    long long cost[kMaxTasks];
    long long tiles_all = 0;
    size_t lds_k = 0;
    for (int k = 0; k < n; ++k) {
        tt.task[k] = tasks[k];
        tt.qc[k] = 0;
        size_t need = (size_t)(tasks[k].dx + D) * 4 * D * 4;               // two fp16 pieces
        if (tasks[k].KT != nullptr) need += (size_t)4 * D * D * 4;
        if (need + extra(NWMAX) > 160 * 1024)
            return fail(TSPGNN_EUNSUPPORTED, "lnlstm_bwd_h2: dx=%d, d=%d%s does not fit LDS", tasks[k].dx, D,
                        tasks[k].KT ? " with K^T" : "");
        if (need > lds_k) lds_k = need;
        const long long tiles = ((long long)tasks[k].rows + 15) / 16;
        cost[k] = tiles * ((tasks[k].dx + D) / 32 + (tasks[k].KT ? 2 : 0) + (tasks[k].KTg ? 5 : 0) + 10);
        tiles_all += tiles;
    }
    tt.n = n;
    int nw = NWMAX;
    if (tiles_all <= (long long)n_cus() * 4) nw = 4;
    const size_t lds_bytes = lds_k + extra(nw);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlstm_bwd_h2_kernel<D, NWMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail((int)e, "lnlstm_bwd_h2: hipFuncSetAttribute: %s", hipGetErrorString(e));
    int grid = n_cus();
    const long long max_grid = (tiles_all + nw - 1) / nw;
    if (grid > max_grid) grid = (int)max_grid;
    grid = split_blocks_bwd_h2(cost, n, grid, tt.blk_end);
    lnlstm_bwd_h2_kernel<D, NWMAX><<<grid, nw * 64, lds_bytes, st>>>(tt);
    int rc = launched("tspgnn_lnlstm_bwd_multi_h2");
    if (rc) return rc;
    for (int k = 0; k < n; ++k) {
        if (tasks[k].defer_reduce) continue;
        const int nblk = tt.blk_end[k] - (k ? tt.blk_end[k - 1] : 0);
        reduce_partials(tasks[k].workspace, nblk, 10 * D, tasks[k].ln_grad, 10 * D, 1.0f, 1, st);
        if ((rc = launched("tspgnn_lnlstm_bwd_multi_h2(reduce)"))) return rc;
    }
    return TSPGNN_OK;
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_lnlstm_bwd_multi_h2(const tspgnn_lstm_bwd_task* tasks, int n_tasks, int d, void* stream) {
    TSPGNN_REQUIRE(tasks && n_tasks >= 1 && n_tasks <= kMaxTasks, "lnlstm_bwd_multi_h2: 1..%d tasks", kMaxTasks);
    TSPGNN_REQUIRE(d == 32 || d == 64, "lnlstm_bwd_h2: d=%d must be 32 or 64", d);
    tspgnn_lstm_bwd_task live[kMaxTasks];
    int n = 0;
    for (int k = 0; k < n_tasks; ++k) {
        const tspgnn_lstm_bwd_task& t = tasks[k];
        TSPGNN_REQUIRE(t.rows >= 0, "lnlstm_bwd_h2: rows=%d", t.rows);
        TSPGNN_REQUIRE((long long)t.rows * (4 * d > t.dx ? 4 * d : t.dx) < (1ll << 30), "lnlstm_bwd_h2: rows=%d too large for 32-bit offsets",
                       t.rows);
        TSPGNN_REQUIRE(t.dx >= 0 && t.dx % 32 == 0, "lnlstm_bwd_h2: dx=%d must be a non-negative multiple of 32", t.dx);
        if (t.rows == 0) continue;
        TSPGNN_REQUIRE(t.h && t.c && t.K && t.ln && t.dz && t.dc_in && t.ln_grad && t.workspace && (t.dx == 0 || t.x),
                       "lnlstm_bwd_h2: null pointer");
        TSPGNN_REQUIRE(!t.uv || (t.dx == 0 && t.Zx), "lnlstm_bwd_h2: gather-init mode needs dx == 0 and Zx");
        TSPGNN_REQUIRE(!t.KT || (t.dxh && t.dx == 0), "lnlstm_bwd_h2: the fused data gradient needs dxh and dx == 0");
        TSPGNN_REQUIRE(!t.zbias || (t.zscale && !t.uv), "lnlstm_bwd_h2: zbias needs zscale and excludes gather-init mode");
        TSPGNN_REQUIRE(!t.KTg || (d == 64 && t.dx == 64 && t.dxg && t.dxh && !t.KT && !t.uv),
                       "lnlstm_bwd_h2: the streamed data gradient needs d == dx == 64, dxg, dxh and excludes KT / gather-init mode");
        live[n++] = t;
    }
    if (n == 0) return TSPGNN_OK;
    hipStream_t st = as_stream(stream);
    return d == 32 ? launch_lnlstm_bwd_h2<32>(live, n, st) : launch_lnlstm_bwd_h2<64>(live, n, st);
}
