// The whole T-step message-passing loop (graphnn.py:175-179 over while_body, graphnn.py:142-173) as ONE launch.
//
// Why.  Launched step by step (dense_h2.hip + aggregate.hip: row-sum, then cell + message MLP), a C2 step costs 44.6 us of
// which the matrix and vector pipes are busy for ~25: the edge states h, c make a round trip through HBM / the Infinity
// Cache every step (102 MB at C2, 16.8 of the edge task's 37.8 us by ablation), the V<-E row-sum is a launch of its own
// between two kernel boundaries, and the last wavefront round of a launch runs a quarter full.  None of that is demanded
// by the data flow: EV is block-diagonal by instance (instance_loader.py:56-66), so nothing a step computes for one
// instance depends on another instance, and an edge row's state is only ever read by the tile that owns the row.
//
// What.  One workgroup per compute unit (8 wavefronts = 2 per SIMD, 256 registers each), resident for all T steps.
//   * EDGE workgroups.  A wavefront owns <= 4 tiles of 16 edge rows for the whole loop and keeps their state in
//     registers: c as fp32 (16 registers per tile), h as the two fp16 pieces the next step's GEMM (and the message MLP's
//     first layer) consume (16 registers).  z is formed one gate (pair) at a time -- f, (i, j), o: 16 / 32 / 16 accumulator
//     registers instead of 64 -- which costs nothing here because the B operand is resident (lstm_stage_*: bit-identical
//     to the one-stage form).  Per step and tile: gather Zx[u] + Zx[v] (the projected vertex messages, L2), three staged
//     GEMMs against Kh (LDS), gates, the message MLP (LDS), one 16-byte write-through store per lane and column tile.
//   * The V<-E ROW-SUM of a group of instances is shared by the edge wavefronts whose first tile lies in the group: each
//     sums a few vertex rows (16 lanes per vertex, four vertices per pass, the summation order of csr_rowsum_kernel) as soon
//     as the group's message tiles of the previous step have all arrived.
//   * VERTEX workgroups run the vertex cells in lock step through two LDS residencies per step (the cell's kernel matrix,
//     then message MLP + projection), <= 2 tiles of 16 vertex rows per wavefront, states through memory (2.6 MB at C2).
//   * Synchronisation is per GROUP of consecutive instances, never grid-wide: three monotone device counters per group
//     and step parity (message tiles arrived, vertex rows aggregated, vertex tiles projected).  Producers store write-through (sc0 sc1),
//     drain (vmcnt(0)) and add to the counter (agent scope); consumers poll one word, then read with sc0 sc1 loads
//     (row-sum operands, aggregated rows: read once) or behind ONE agent-scope acquire per step (the Zx gathers, which
//     re-use lines across the lanes and tiles of a step) -- the hand-off forms of the CDNA4 guide, independent of where a
//     workgroup runs; the plan only PLACES a group's producers and consumers on one XCD (workgroup b -> XCD b mod 8,
//     observed) so that the traffic stays within one L2.
//   * Every wait is bounded: a wavefront that sees no progress for ~0.5 s raises args.status and stops waiting (so do all
//     others when they see the word), the launch then ends with garbage outputs instead of hanging the GPU.
//
// Dependences of step t (both updates read the OLD states, graphnn.py:143):
//     edge(g, t)   <-  Zx_t(g)   = projection of vertex messages, written by vertex(g, t-1)
//     rowsum(g, t) <-  msg_t(g)  = edge messages, written by edge(g, t-1)
//     vertex(g, t) <-  rowsum(g, t)
// so a group's vertex chain has one whole edge step of slack, and double buffering by step parity suffices (see the
// ordering argument in DESIGN.md).
#include "common.h"
#include "h2_tile.h"
#include "loop_sync.h"
#include "mfma_tile.h"

#include <type_traits>

namespace tspgnn {

constexpr int kLoopWaves = TSPGNN_LOOP_WAVES;
constexpr int kLoopDesc = TSPGNN_LOOP_DESC_INTS;
constexpr int kLoopEdgeTiles = 4;
constexpr int kLoopVertTiles = 2;
#ifndef LOOP_PAIR
#define LOOP_PAIR 4   // which stages step two resident tiles together (see pair_step)
#endif

// Dense(D) whose input arrives as the two fp16 pieces of the operand (dense_layer_h2 minus its split: same MFMAs, same
// order, same epilogue).
template <int D>
__device__ __forceinline__ void dense_layer_h2_pieces(const f16x8 (&bh)[D / 32], const f16x8 (&bl)[D / 32], f32x4 (&a)[D / 16],
                                                      const _Float16* wh, const _Float16* wl, const float* bias, bool relu,
                                                      int g, int rl) {
    constexpr int NT = D / 16, KB = D / 32;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = ld4(bias + t * 16 + g * 4);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) kblock_h2<NT>(acc, wh, wl, kb, g, rl, bh[kb], bl[kb]);
    const f32x2 inv = {kH2InvScale, kH2InvScale};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
        }
        a[t].lo = acc[t].lo * inv;
        a[t].hi = acc[t].hi * inv;
    }
}
// (args.trace != NULL selects the TRACE variant of the kernel, LoopTrace in loop_sync.h: see tools/loop_trace.py for the
// phases)
template <int D, bool CENTERED, bool TRACE>
__global__ __launch_bounds__(kLoopWaves * 64) void mp_loop_h2_kernel(const tspgnn_mp_loop_args a) {
    constexpr int TPG = D / 16, NT4 = D / 4, KBH = D / 32;
    constexpr int LAYER_BYTES = 2 * D * D * 2 + D * 4;   // { hi, lo, bias } of one MLP layer
    constexpr bool SWAP = H2_LN_SWAP != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    float* lds_ln = reinterpret_cast<float*>(ldsb);
    unsigned char* lds_wb = ldsb + (10 * D + 4) * sizeof(float);
    _Float16* lds_w = reinterpret_cast<_Float16*>(lds_wb);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = kLoopWaves * 64;
    const int T = a.T;
    const int* desc = a.plan + ((size_t)blockIdx.x * kLoopWaves + wave) * kLoopDesc;
    const int role = a.plan[(size_t)blockIdx.x * kLoopWaves * kLoopDesc];   // uniform over the workgroup
    auto ds = [&](int i) { return __builtin_amdgcn_readfirstlane(desc[i]); };
    unsigned* const counters = a.counters;
    // Counters are split by the PARITY of the step whose data they announce.  A producer can run one step ahead of a
    // sibling that feeds the same counter (a vertex workgroup only needs ITS rows' aggregates to start step t+1, an edge
    // wavefront without a row-sum share only its groups' projected tiles) but never two: with one monotone counter per
    // group the early arrival of step t+1 would complete the count of step t before the slow sibling has stored its rows
    // (found by the protocol model in tests/test_loop_plan.py, not by the GPU: the window is a few hundred ns).
    auto cnt_msg = [&](int grp, int par) { return counters + (size_t)(grp * 3 + 0) * 32 + par * 16; };
    auto cnt_vagg = [&](int grp, int par) { return counters + (size_t)(grp * 3 + 1) * 32 + par * 16; };
    auto cnt_zx = [&](int grp, int par) { return counters + (size_t)(grp * 3 + 2) * 32 + par * 16; };
    bool dead = false;
    float wit = 0.f;
    unsigned vmin = 0xffffffffu;
    if (role == 0) return;
    LoopTrace<TRACE> tr;   // (a compile-time variant: its sixteen scalar registers are not free in the production kernel)
    tr.begin(a.trace + ((size_t)blockIdx.x * kLoopWaves + wave) * 16);

    // LayerNorm parameters of this workgroup's cell, rows [g_i, b_i, g_j, b_j, g_f, b_f, g_o, b_o, g_s, b_s]; the gates
    // i, f, o feed sigmoids only: gamma / beta times -log2(e), forget bias folded into b_f (as lnlstm_mlp_fwd_h2_kernel)
    {
        const float* ln = role == 1 ? a.e_ln : a.v_ln;
        for (int i = tid; i < 10 * D; i += nthreads) {
            const int r = i / D;
            float v = ln[i];
            if (r == 5) v += 1.0f;
            if (r < 2 || (r >= 4 && r < 8)) v *= kNegLog2e;
            lds_ln[i] = v;
        }
    }
    const __amdgpu_buffer_rsrc_t r_msg0 = make_rsrc(a.msg[0], (long long)a.M * D * 4);
    const __amdgpu_buffer_rsrc_t r_msg1 = make_rsrc(a.msg[1], (long long)a.M * D * 4);
    const __amdgpu_buffer_rsrc_t r_vagg0 = make_rsrc(a.vagg[0], (long long)a.N * D * 4);
    const __amdgpu_buffer_rsrc_t r_vagg1 = make_rsrc(a.vagg[1], (long long)a.N * D * 4);

    if (role == 1) {
        // ------------------------------------------------------------------------------------------- edge workgroup
        constexpr int total = D * 4 * D;   // elements per piece of Kh
        unsigned char* lds_mlp = lds_wb + (size_t)2 * total * 2;
        const int L = a.e_mlp_layers;
        h2_copy_to_lds(lds_w, a.e_K, 2 * total * 2, tid, nthreads);
        if (L > 0) h2_copy_to_lds(lds_mlp, a.e_mlp_wb, L * LAYER_BYTES, tid, nthreads);
        h2_stage_wait();
        __syncthreads();
        const int nt = ds(1);
        if (nt == 0) return;   // (no block-wide barrier below this point in an edge workgroup)
        const int row0[kLoopEdgeTiles] = {ds(2), ds(3), ds(4), ds(5)};
        const int nvalid[kLoopEdgeTiles] = {ds(6), ds(7), ds(8), ds(9)};
        const int ga = ds(10), gb = ds(11), n_a = ds(12), n_b = ds(13), nvt_a = ds(14), nvt_b = ds(15), net_a = ds(16);
        const int sv0 = ds(17), sv1 = ds(18);
        const int2* uv = reinterpret_cast<const int2*>(a.uv);

        f32x4 cst[kLoopEdgeTiles][TPG];
        f16x8 hh[kLoopEdgeTiles][KBH], hl[kLoopEdgeTiles][KBH];
        auto load_state = [&](auto I) {
            constexpr int i = decltype(I)::value;
            const int l = opaque_lane();
            const int rl = l & 15, g = l >> 4;
            const unsigned rc = (unsigned)(row0[i] + (rl < nvalid[i] ? rl : 0));
            f32x4 hv[TPG];
#pragma unroll
            for (int q = 0; q < TPG; ++q) hv[q] = ld4(a.e_h0 + (size_t)rc * D + g * 4 + q * 16);
#pragma unroll
            for (int q = 0; q < TPG; ++q)
                cst[i][q] = a.e_c0 != nullptr ? ld4(a.e_c0 + (size_t)rc * D + g * 4 + q * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KBH; ++kb) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = hv[2 * kb + (j >> 2)][j & 3];
                split2w(x, hh[i][kb], hl[i][kb], wit);
            }
        };
        if (nt > 0) load_state(std::integral_constant<int, 0>{});
        if (nt > 1) load_state(std::integral_constant<int, 1>{});
        if (nt > 2) load_state(std::integral_constant<int, 2>{});
        if (nt > 3) load_state(std::integral_constant<int, 3>{});

        // The row-sum share's edge lists never change: up to 2 passes of 4 vertices (16 lanes each) x 48 edge ids wait in
        // LDS behind the weights (a share of more vertices, or a vertex of more than 48 edges, takes the general loop).
        constexpr int kSharePasses = 2, kShareChunks = 3, kShareInts = kSharePasses * (kShareChunks + 1) * 64;
        int* sh = reinterpret_cast<int*>(lds_mlp + (size_t)L * LAYER_BYTES) + wave * kShareInts;   // [pass][chunk | cnt][64]
        bool share_fast = sv1 - sv0 <= 4 * kSharePasses;
        int sh_mx[kSharePasses];
#pragma unroll
        for (int q = 0; q < kSharePasses; ++q) {
            const int v = sv0 + 4 * q + (lane >> 4);
            const int vv = v < sv1 ? v : (sv1 > sv0 ? sv1 - 1 : 0);
            const int beg = sv1 > sv0 ? a.rowptr[vv] : 0;
            const int cnt = sv1 > sv0 ? a.rowptr[vv + 1] - beg : 0;
            int mx = cnt;
            mx = max(mx, __shfl_xor(mx, 16));
            mx = max(mx, __shfl_xor(mx, 32));
            sh_mx[q] = __builtin_amdgcn_readfirstlane(mx);
            if (sh_mx[q] > 16 * kShareChunks) share_fast = false;
#pragma unroll
            for (int ch = 0; ch < kShareChunks; ++ch)
                sh[(q * (kShareChunks + 1) + ch) * 64 + lane] =
                    (ch * 16 + (lane & 15) < cnt && ch * 16 < 16 * kShareChunks) ? a.eid[beg + ch * 16 + (lane & 15)] : 0;
            sh[(q * (kShareChunks + 1) + kShareChunks) * 64 + lane] = cnt;
        }
        // The tiles' gather offsets into the projected messages (both endpoints of the lane's edge row) wait there too: the
        // tile then starts with an LDS read instead of a dependent global round trip (uv -> address -> gather).
        unsigned* zoff = reinterpret_cast<unsigned*>(lds_mlp + (size_t)L * LAYER_BYTES) + kLoopWaves * kShareInts +
                         wave * (kLoopEdgeTiles * 2 * 64);
#pragma unroll
        for (int i = 0; i < kLoopEdgeTiles; ++i) {
            if (i < nt) {
                const int rl = lane & 15, g = lane >> 4;
                const int2 ends = uv[row0[i] + (rl < nvalid[i] ? rl : 0)];
                zoff[(i * 2 + 0) * 64 + lane] = h2_zx_row<D>((unsigned)ends.x, g);
                zoff[(i * 2 + 1) * 64 + lane] = h2_zx_row<D>((unsigned)ends.y, g);
            }
        }
        // (each wavefront reads back only what it wrote itself: LDS operations of one wavefront complete in order)

        for (int t = 0; t < T; ++t) {
            const int p = t & 1;
            const bool last = t == T - 1;
            const __amdgpu_buffer_rsrc_t r_msg_in = p ? r_msg1 : r_msg0, r_msg_out = p ? r_msg0 : r_msg1;
            const __amdgpu_buffer_rsrc_t r_vagg = p ? r_vagg1 : r_vagg0;
            const float* zx = a.zx[p];

            // ---- this wavefront's share of the V<-E row-sum over the messages of step t (parity p)
            if (sv1 > sv0) {
                wait_ge(cnt_msg(ga, p), (unsigned)(((t + 1) >> 1) * net_a), dead, a.status);
                tr.mark(0);
                const int sub = lane >> 4, c = lane & 15;
                if (share_fast) {
                    // edge ids resident (sh_e): per pass and 16-edge chunk one batch of 16 write-through loads in flight
                    auto pass = [&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        const int v = sv0 + 4 * q + sub;
                        const int cnt = sh[(q * (kShareChunks + 1) + kShareChunks) * 64 + lane];
                        f32x4 s[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ch = 0; ch < kShareChunks; ++ch) {
                            if (ch * 16 < sh_mx[q]) {   // (uniform)
#pragma unroll
                                for (int h8 = 0; h8 < 16; h8 += 8) {
                                    f32x4 x[8];
#pragma unroll
                                    for (int kk = 0; kk < 8; ++kk) {
                                        const int e = sh[(q * (kShareChunks + 1) + ch) * 64 + (lane & 48) + h8 + kk];
                                        x[kk] = ld4wt(r_msg_in, ((unsigned)e * D + c * 4) * 4u);
                                    }
#pragma unroll
                                    for (int kk = 0; kk < 8; ++kk)
                                        if (ch * 16 + h8 + kk < cnt) s[kk & 3] += x[kk];
                                }
                            }
                        }
                        const f32x4 tot = (s[0] + s[1]) + (s[2] + s[3]);
                        if (v < sv1) st4wt(r_vagg, ((unsigned)v * D + c * 4) * 4u, tot);
                    };
                    pass(std::integral_constant<int, 0>{});
                    if (sv0 + 4 < sv1) pass(std::integral_constant<int, 1>{});
                } else {
                    for (int vb = sv0; vb < sv1; vb += 4) {
                        const int v = vb + sub;
                        const bool on = v < sv1;
                        const int vv = on ? v : sv1 - 1;
                        const int beg = a.rowptr[vv], cnt = a.rowptr[vv + 1] - beg;
                        int mx = cnt;
                        mx = max(mx, __shfl_xor(mx, 16));
                        mx = max(mx, __shfl_xor(mx, 32));
                        f32x4 s[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        for (int base = 0; base < mx; base += 16) {
                            const int my_e = (base + c < cnt) ? a.eid[beg + base + c] : 0;
#pragma unroll
                            for (int h8 = 0; h8 < 16; h8 += 8) {
                                f32x4 x[8];
#pragma unroll
                                for (int kk = 0; kk < 8; ++kk) {
                                    const int e = __shfl(my_e, (lane & 48) + h8 + kk);
                                    x[kk] = ld4wt(r_msg_in, ((unsigned)e * D + c * 4) * 4u);
                                }
#pragma unroll
                                for (int kk = 0; kk < 8; ++kk)
                                    if (base + h8 + kk < cnt) s[kk & 3] += x[kk];   // ((h8 + kk) & 3 == kk & 3)
                            }
                        }
                        const f32x4 tot = (s[0] + s[1]) + (s[2] + s[3]);   // the order of csr_rowsum_kernel's lane-group butterfly
                        if (on) st4wt(r_vagg, ((unsigned)v * D + c * 4) * 4u, tot);
                    }
                }
                drain_stores();
                arrive(cnt_vagg(ga, p), (unsigned)(sv1 - sv0));
                tr.mark(1);
            }

            // ---- the edge cells of step t on the resident tiles
            wait_ge(cnt_zx(ga, p), (unsigned)(((t + 1) >> 1) * nvt_a), dead, a.status);
            if (gb != ga) wait_ge(cnt_zx(gb, p), (unsigned)(((t + 1) >> 1) * nvt_b), dead, a.status);
            tr.mark(2);
            if (t > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            tr.mark(3);

            // each stage as a function of the tiles (compile-time indices) it runs on together
            const _Float16* Kl = lds_w + total;
            auto coords = [&](int tile, int& rl, int& g, bool& valid, unsigned& rc, const float*& zu, const float*& zv) {
                const int l = opaque_lane();
                rl = l & 15;
                g = l >> 4;
                valid = rl < nvalid[tile];
                rc = (unsigned)(row0[tile] + (valid ? rl : 0));
                zu = zx + zoff[(tile * 2 + 0) * 64 + l];
                zv = zx + zoff[(tile * 2 + 1) * 64 + l];
            };
            // GATE: 0 = f (column tiles 2 TPG..), 1 = o (3 TPG..)
            auto stage_fo = [&](auto GATE, auto... Is) {
                constexpr int gate = decltype(GATE)::value;
                constexpr int NP = sizeof...(Is);
                constexpr int ti[NP] = {decltype(Is)::value...};
                constexpr int T0 = gate == 0 ? 2 * TPG : 3 * TPG;
                f32x4 z[NP][TPG];
                int rl = 0, g = 0;
#pragma unroll
                for (int n = 0; n < NP; ++n) {
                    bool valid;
                    unsigned rc;
                    const float *zu, *zv;
                    coords(ti[n], rl, g, valid, rc, zu, zv);
#pragma unroll
                    for (int q = 0; q < TPG; ++q) z[n][q] = ld4(zu + (T0 + q) * 256);
#pragma unroll
                    for (int q = 0; q < TPG; ++q) z[n][q] += ld4(zv + (T0 + q) * 256);
                }
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) {
                    f16x8 bh[NP], bl[NP];
#pragma unroll
                    for (int n = 0; n < NP; ++n) {
                        bh[n] = hh[ti[n]][kb];
                        bl[n] = hl[ti[n]][kb];
                    }
                    kblock_h2_multi<NT4, T0, TPG, NP>(z, lds_w, Kl, kb, g, rl, bh, bl);
                }
#pragma unroll
                for (int n = 0; n < NP; ++n) {
                    if constexpr (gate == 0) {
                        lstm_stage_f<D, SWAP, CENTERED, true>(z[n], cst[ti[n]], lds_ln, g, kH2GateEps, &vmin);
                    } else {
                        // h' leaves the stage as the next GEMMs' operand pieces (resident); only the last step needs it as fp32
                        f32x4 hn[TPG];
                        lstm_stage_o<D, SWAP, CENTERED, true>(z[n], cst[ti[n]], lds_ln, g, hn, kH2GateEps, &vmin);
                        if (last) {
                            bool valid;
                            unsigned rc;
                            const float *zu, *zv;
                            coords(ti[n], rl, g, valid, rc, zu, zv);
                            if (valid) {
                                float* hd = a.e_h + (size_t)rc * D + g * 4;
                                float* cd = a.e_c + (size_t)rc * D + g * 4;
#pragma unroll
                                for (int q = 0; q < TPG; ++q) {
                                    st4(hd + q * 16, hn[q]);
                                    st4(cd + q * 16, cst[ti[n]][q]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int kb = 0; kb < KBH; ++kb) {
                                float x[8];
#pragma unroll
                                for (int j = 0; j < 8; ++j) x[j] = hn[2 * kb + (j >> 2)][j & 3];
                                split2w(x, hh[ti[n]][kb], hl[ti[n]][kb], wit);
                            }
                        }
                    }
                }
            };
            auto stage_ij = [&](auto I) {
                constexpr int i = decltype(I)::value;
                int rl, g;
                bool valid;
                unsigned rc;
                const float *zu, *zv;
                coords(i, rl, g, valid, rc, zu, zv);
                f32x4 z[2 * TPG];
#pragma unroll
                for (int q = 0; q < 2 * TPG; ++q) z[q] = ld4(zu + q * 256);
#pragma unroll
                for (int q = 0; q < 2 * TPG; ++q) z[q] += ld4(zv + q * 256);
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb)
                    kblock_h2_sub<NT4, 0, 2 * TPG>(z, lds_w, Kl, kb, g, rl, hh[i][kb], hl[i][kb]);
                lstm_stage_ij<D, SWAP, CENTERED, true>(z, cst[i], lds_ln, g, kH2GateEps, &vmin);
            };
            // the message MLP on h' (its pieces) + the message stores
            auto stage_out = [&](auto... Is) {
                constexpr int NP = sizeof...(Is);
                constexpr int ti[NP] = {decltype(Is)::value...};
                int rl = 0, g = 0;
                bool valid[NP];
                unsigned rc[NP];
#pragma unroll
                for (int n = 0; n < NP; ++n) {
                    const float *zu, *zv;
                    coords(ti[n], rl, g, valid[n], rc[n], zu, zv);
                }
                f32x4 hn[NP][TPG];
                const unsigned mask = a.e_relu_mask;
                for (int ly = 0; ly < L; ++ly) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_mlp + (size_t)ly * LAYER_BYTES);
                    const float* bias = reinterpret_cast<const float*>(lds_mlp + (size_t)ly * LAYER_BYTES + 2 * D * D * 2);
                    f32x4 acc[NP][TPG];
#pragma unroll
                    for (int n = 0; n < NP; ++n) {
#pragma unroll
                        for (int q = 0; q < TPG; ++q) acc[n][q] = ld4(bias + q * 16 + g * 4);
                    }
#pragma unroll
                    for (int kb = 0; kb < KBH; ++kb) {
                        f16x8 bh[NP], bl[NP];
#pragma unroll
                        for (int n = 0; n < NP; ++n) {
                            if (ly == 0) {   // h' arrives as the pieces just made (the next step's GEMM operand)
                                bh[n] = hh[ti[n]][kb];
                                bl[n] = hl[ti[n]][kb];
                            } else {
                                float x[8];
#pragma unroll
                                for (int j = 0; j < 8; ++j) x[j] = hn[n][2 * kb + (j >> 2)][j & 3];
                                split2w(x, bh[n], bl[n], wit);
                            }
                        }
                        kblock_h2_multi<TPG, 0, TPG, NP>(acc, wh, wh + D * D, kb, g, rl, bh, bl);
                    }
                    const bool relu = (mask >> ly) & 1u;
                    const f32x2 inv = {kH2InvScale, kH2InvScale};
#pragma unroll
                    for (int n = 0; n < NP; ++n) {
#pragma unroll
                        for (int q = 0; q < TPG; ++q) {
                            if (relu) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[n][q][r] = fmaxf(acc[n][q][r], 0.f);
                            }
                            hn[n][q].lo = acc[n][q].lo * inv;
                            hn[n][q].hi = acc[n][q].hi * inv;
                        }
                    }
                }
#pragma unroll
                for (int n = 0; n < NP; ++n) {
                    if (valid[n]) {
#pragma unroll
                        for (int q = 0; q < TPG; ++q) st4wt(r_msg_out, (rc[n] * D + g * 4 + q * 16) * 4u, hn[n][q]);
                    }
                }
            };
            // The resident tiles are stepped in PAIRS where the register budget allows (LOOP_PAIR: bit 0 the f stage, bit 1
            // the o stage, bit 2 the message MLP): the stages of two tiles then share every weight fragment
            // (kblock_h2_multi: half the LDS reads per tile, two independent MFMA chains per fragment, two LayerNorm /
            // transcendental chains to interleave).  A wavefront is otherwise bound by the LATENCY of its own dependent
            // chains (~10 us per tile against ~4 us of issue), which two wavefronts per SIMD only half hide.  The (i, j)
            // stage stays one tile at a time: 32 accumulator registers per tile.
            auto pair_step = [&](auto I0, auto I1) {
                constexpr std::integral_constant<int, 0> F{};
                constexpr std::integral_constant<int, 1> O{};
                if constexpr (LOOP_PAIR & 1) stage_fo(F, I0, I1);
                else { stage_fo(F, I0); stage_fo(F, I1); }
                tr.mark(8);
                stage_ij(I0);
                stage_ij(I1);
                tr.mark(9);
                if constexpr (LOOP_PAIR & 2) stage_fo(O, I0, I1);
                else { stage_fo(O, I0); stage_fo(O, I1); }
                tr.mark(10);
                if (!last && L > 0) {
                    if constexpr (LOOP_PAIR & 4) stage_out(I0, I1);
                    else { stage_out(I0); stage_out(I1); }
                }
                tr.mark(12);
            };
            auto single_step = [&](auto I0) {
                constexpr std::integral_constant<int, 0> F{};
                constexpr std::integral_constant<int, 1> O{};
                stage_fo(F, I0);
                tr.mark(8);
                stage_ij(I0);
                tr.mark(9);
                stage_fo(O, I0);
                tr.mark(10);
                if (!last && L > 0) stage_out(I0);
                tr.mark(12);
            };
            if (nt >= 2) pair_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            else single_step(std::integral_constant<int, 0>{});
            if (nt >= 4) pair_step(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
            else if (nt == 3) single_step(std::integral_constant<int, 2>{});
            tr.mark(4);
            if (!last) {
                drain_stores();
                arrive(cnt_msg(ga, 1 - p), (unsigned)n_a);
                if (n_b > 0) arrive(cnt_msg(gb, 1 - p), (unsigned)n_b);
            }
            tr.mark(5);
        }
        tr.flush();
        h2_range_report(a.range_flag, wit, vmin);
        return;
    }

    // ----------------------------------------------------------------------------------------------- vertex workgroup
    {
        constexpr int KBT = 2 * KBH;           // k-blocks of [x | h]
        constexpr int total = 2 * D * 4 * D;   // elements per piece of K[2D, 4D]
        const int L = a.v_mlp_layers;
        unsigned char* lds_proj = lds_wb + (size_t)L * LAYER_BYTES;
        const __amdgpu_buffer_rsrc_t r_zx0 = make_rsrc(a.zx[0], (long long)((a.N + 15) / 16) * 16 * 4 * D * 4);
        const __amdgpu_buffer_rsrc_t r_zx1 = make_rsrc(a.zx[1], (long long)((a.N + 15) / 16) * 16 * 4 * D * 4);
        const int nt = ds(1);
        const int row0[kLoopVertTiles] = {ds(2), ds(3)};
        const int nvalid[kLoopVertTiles] = {ds(6), ds(7)};
        const int grp[kLoopVertTiles] = {ds(10), ds(11)};
        const int nvg[kLoopVertTiles] = {ds(12), ds(13)};
        h2_copy_to_lds(lds_w, a.v_K, 2 * total * 2, tid, nthreads);
        for (int t = 0; t < T; ++t) {
            const int p = t & 1;
            const bool last = t == T - 1;
            const __amdgpu_buffer_rsrc_t r_vagg = p ? r_vagg1 : r_vagg0;
            const __amdgpu_buffer_rsrc_t r_zx_out = p ? r_zx0 : r_zx1;
            const float* h_in = t == 0 ? a.v_h0 : a.v_h;
            const float* c_in = t == 0 ? a.v_c0 : a.v_c;
            f32x4 hn[kLoopVertTiles][TPG];
            // operands first (the aggregated rows arrive through the counters), then the K residency is waited for
            f32x4 xo[kLoopVertTiles][TPG], ho[kLoopVertTiles][TPG];
            auto fetch = [&](auto J) {
                constexpr int j = decltype(J)::value;
                wait_ge(cnt_vagg(grp[j], p), (unsigned)(((t >> 1) + 1) * nvg[j]), dead, a.status);
                const int l = opaque_lane();
                const int rl = l & 15, g = l >> 4;
                const unsigned rc = (unsigned)(row0[j] + (rl < nvalid[j] ? rl : 0));
#pragma unroll
                for (int q = 0; q < TPG; ++q) xo[j][q] = ld4wt(r_vagg, (rc * D + g * 4 + q * 16) * 4u);
#pragma unroll
                for (int q = 0; q < TPG; ++q) ho[j][q] = ld4(h_in + (size_t)rc * D + g * 4 + q * 16);
            };
            if (nt > 0) fetch(std::integral_constant<int, 0>{});
            if (nt > 1) fetch(std::integral_constant<int, 1>{});
            tr.mark(0);
            h2_stage_wait();
            __syncthreads();
            tr.mark(1);
            auto cell = [&](auto J) {
                constexpr int j = decltype(J)::value;
                const int l = opaque_lane();
                const int rl = l & 15, g = l >> 4;
                const bool valid = rl < nvalid[j];
                const unsigned rc = (unsigned)(row0[j] + (valid ? rl : 0));
                f32x4 acc[NT4], cf[TPG];
                if (a.v_zbias != nullptr) {
                    const float sc = a.v_zscale[rc] * kH2Scale;
#pragma unroll
                    for (int q = 0; q < NT4; ++q) acc[q] = ld4(a.v_zbias + q * 16 + g * 4) * sc;
                } else {
#pragma unroll
                    for (int q = 0; q < NT4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int kb = 0; kb < KBT; ++kb) {
                    float x[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj)
                        x[jj] = kb < KBH ? xo[j][2 * kb + (jj >> 2)][jj & 3] : ho[j][2 * (kb - KBH) + (jj >> 2)][jj & 3];
                    f16x8 bh, bl;
                    split2w(x, bh, bl, wit);
                    kblock_h2<NT4>(acc, lds_w, lds_w + total, kb, g, rl, bh, bl);
                }
#pragma unroll
                for (int q = 0; q < TPG; ++q)
                    cf[q] = c_in != nullptr ? ld4(c_in + (size_t)rc * D + g * 4 + q * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 nc[TPG];
                lstm_gates<D, true, SWAP, CENTERED, true>(acc, cf, lds_ln, g, hn[j], nc, kH2GateEps, &vmin);
                if (valid) {
                    float* hd = a.v_h + (size_t)rc * D + g * 4;
                    float* cd = a.v_c + (size_t)rc * D + g * 4;
#pragma unroll
                    for (int q = 0; q < TPG; ++q) {
                        st4(hd + q * 16, hn[j][q]);
                        st4(cd + q * 16, nc[q]);
                    }
                }
            };
            if (nt > 0) cell(std::integral_constant<int, 0>{});
            if (nt > 1) cell(std::integral_constant<int, 1>{});
            tr.mark(2);
            if (last) break;   // (uniform over the workgroup)
            __syncthreads();   // every wavefront is done with K: the second residency
            h2_copy_to_lds(lds_wb, a.v_mlp_wb, L * LAYER_BYTES, tid, nthreads);
            h2_copy_to_lds(lds_proj, a.v_proj_w, 2 * D * 4 * D * 2, tid, nthreads);
            h2_stage_wait();
            __syncthreads();
            tr.mark(3);
            auto message = [&](auto J) {
                constexpr int j = decltype(J)::value;
                const int l = opaque_lane();
                const int rl = l & 15, g = l >> 4;
                const bool valid = rl < nvalid[j];
                const unsigned rc = (unsigned)(row0[j] + (valid ? rl : 0));
                const unsigned mask = a.v_relu_mask;
                for (int ly = 0; ly < L; ++ly) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_wb + (size_t)ly * LAYER_BYTES);
                    const float* bias = reinterpret_cast<const float*>(lds_wb + (size_t)ly * LAYER_BYTES + 2 * D * D * 2);
                    dense_layer_h2<D>(hn[j], wh, wh + D * D, bias, (mask >> ly) & 1u, g, rl, wit);
                }
                // Zx = 2^s (y Kx), a gate (TPG column tiles) at a time on the same operand pieces
                f16x8 yh[KBH], yl[KBH];
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) {
                    float x[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) x[jj] = hn[j][2 * kb + (jj >> 2)][jj & 3];
                    split2w(x, yh[kb], yl[kb], wit);
                }
                const _Float16* wp = reinterpret_cast<const _Float16*>(lds_proj);
                const unsigned zoff = h2_zx_row<D>(rc, g);
                auto gate = [&](auto S) {
                    constexpr int s = decltype(S)::value;
                    f32x4 acc[TPG];
#pragma unroll
                    for (int q = 0; q < TPG; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < KBH; ++kb)
                        kblock_h2_sub<NT4, s * TPG, TPG>(acc, wp, wp + D * 4 * D, kb, g, rl, yh[kb], yl[kb]);
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < TPG; ++q) st4wt(r_zx_out, (zoff + (unsigned)(s * TPG + q) * 256u) * 4u, acc[q]);
                    }
                };
                gate(std::integral_constant<int, 0>{});
                gate(std::integral_constant<int, 1>{});
                gate(std::integral_constant<int, 2>{});
                gate(std::integral_constant<int, 3>{});
            };
            if (nt > 0) message(std::integral_constant<int, 0>{});
            if (nt > 1) message(std::integral_constant<int, 1>{});
            tr.mark(4);
            drain_stores();
            if (nt > 0) arrive(cnt_zx(grp[0], 1 - p), 1u);
            if (nt > 1) arrive(cnt_zx(grp[1], 1 - p), 1u);
            tr.mark(5);
            __syncthreads();   // done with the MLP residency: K comes back behind the wait for the next aggregates
            h2_copy_to_lds(lds_w, a.v_K, 2 * total * 2, tid, nthreads);
            tr.mark(6);
        }
        tr.flush();
        h2_range_report(a.range_flag, wit, vmin);
    }
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_mp_loop_h2(const tspgnn_mp_loop_args* args, int d, void* stream) {
    TSPGNN_REQUIRE(args, "mp_loop_h2: null args");
    TSPGNN_REQUIRE(d == 64, "mp_loop_h2: d=%d must be 64", d);
    const tspgnn_mp_loop_args& a = *args;
    TSPGNN_REQUIRE(a.T >= 1, "mp_loop_h2: T=%d must be >= 1", a.T);
    TSPGNN_REQUIRE(a.M > 0 && a.N > 0 && a.n_groups > 0, "mp_loop_h2: M=%d, N=%d, n_groups=%d", a.M, a.N, a.n_groups);
    TSPGNN_REQUIRE((long long)a.M * d * 4 < (1ll << 31) && ((long long)a.N + 16) * 4 * d * 4 < (1ll << 31),
                   "mp_loop_h2: M=%d / N=%d too large for 32-bit byte offsets", a.M, a.N);
    TSPGNN_REQUIRE(a.grid >= 1 && a.grid <= n_cus(), "mp_loop_h2: grid=%d must be in 1..%d (one resident workgroup per CU)",
                   a.grid, n_cus());
    TSPGNN_REQUIRE(a.e_h0 && a.e_h && a.e_c && a.uv && a.e_K && a.e_ln && a.msg[0] && a.msg[1], "mp_loop_h2: null edge pointer");
    TSPGNN_REQUIRE(a.v_h0 && a.v_h && a.v_c && a.rowptr && a.eid && a.v_K && a.v_ln && a.zx[0] && a.zx[1] && a.vagg[0] &&
                       a.vagg[1],
                   "mp_loop_h2: null vertex pointer");
    TSPGNN_REQUIRE(a.plan && a.counters, "mp_loop_h2: null plan / counters");
    TSPGNN_REQUIRE(a.e_mlp_layers >= 0 && a.e_mlp_layers <= 3 && (a.e_mlp_layers == 0 || a.e_mlp_wb),
                   "mp_loop_h2: e_mlp_layers=%d must be in 0..3 (resident next to Kh)", a.e_mlp_layers);
    TSPGNN_REQUIRE(a.v_mlp_layers >= 1 && a.v_mlp_layers <= 4 && a.v_mlp_wb && a.v_proj_w,
                   "mp_loop_h2: v_mlp_layers=%d must be in 1..4, with a projection", a.v_mlp_layers);
    TSPGNN_REQUIRE(!a.v_zbias || a.v_zscale, "mp_loop_h2: v_zbias needs v_zscale");
    TSPGNN_REQUIRE(a.e_h0 != a.e_h && a.v_h0 != a.v_h, "mp_loop_h2: the final states must not alias the initial ones");
    constexpr int D = 64;
    const size_t head = (10 * D + 4) * sizeof(float);
    const size_t layer = 2 * D * D * 2 + D * 4;
    const size_t edge_bytes = (size_t)2 * D * 4 * D * 2 + a.e_mlp_layers * layer + (size_t)kLoopWaves * 2 * 4 * 64 * 4 +
                              (size_t)kLoopWaves * kLoopEdgeTiles * 2 * 64 * 4;   // + the row-sum shares' edge lists + gather offsets
    const size_t vert_k = (size_t)2 * 2 * D * 4 * D * 2, vert_m = a.v_mlp_layers * layer + (size_t)2 * D * 4 * D * 2;
    size_t lds_bytes = edge_bytes > vert_k ? edge_bytes : vert_k;
    if (vert_m > lds_bytes) lds_bytes = vert_m;
    lds_bytes += head;
    void (*fn)(const tspgnn_mp_loop_args) =
        a.trace ? (a.z_centered ? &mp_loop_h2_kernel<D, true, true> : &mp_loop_h2_kernel<D, false, true>)
                : (a.z_centered ? &mp_loop_h2_kernel<D, true, false> : &mp_loop_h2_kernel<D, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
        return fail(TSPGNN_EUNSUPPORTED, "mp_loop_h2: hipFuncSetAttribute(%d B): %s", (int)lds_bytes, hipGetErrorString(e));
    // every wait inside the launch assumes all `grid` workgroups are resident at once: ask the runtime (the caller falls
    // back to the stepwise launches on TSPGNN_EUNSUPPORTED)
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), kLoopWaves * 64, lds_bytes);
    if (e != hipSuccess || per_cu < 1)
        return fail(TSPGNN_EUNSUPPORTED, "mp_loop_h2: a workgroup of %d threads and %zu bytes of LDS is not resident on this device",
                    kLoopWaves * 64, lds_bytes);
    fn<<<a.grid, kLoopWaves * 64, lds_bytes, as_stream(stream)>>>(a);
    return launched("tspgnn_mp_loop_h2");
}
