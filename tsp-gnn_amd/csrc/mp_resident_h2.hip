// The whole T-step message-passing loop (graphnn.py:175-179 over while_body, graphnn.py:142-173) as ONE launch of resident
// workgroups with the edge states in MEMORY -- the middle between the stepwise launches (dense_h2.hip + aggregate.hip) and
// the register-resident loop (mp_loop_h2.hip).
//
// Why.  At BASELINE's n=40, batch=128 a step launched as {row-sum, cell + message MLP} costs ~41 us; the same cell launch
// looped 32 times INSIDE one launch -- weights staged once, no kernel boundary, the LDS ticket running on into the next
// pass -- costs 27 us per pass (tools/loop_bound_probe.py, profiles/r06_loop_bound.txt): ~14 us of every step are two launch
// boundaries, 112 KB of weights re-staged into every CU, a last wavefront round that runs a quarter full and a row-sum
// launch that overlaps nothing.  The register-resident loop removes all of that but pins a tile's state to ONE wavefront:
// at 4 tiles per wavefront its step is the sum of four ~10 us dependent chains, and it loses at this size.
//
// What.  One workgroup of 12 wavefronts per compute unit, resident for all T steps.
//   * EDGE workgroups keep Kh and the message MLP's layers in LDS for the whole loop.  Work is handed out by ONE LDS ticket
//     that runs through all steps: item k = item (k mod W) of step (k div W), W = the workgroup's items per step (plan).
//     An item is a 16-row edge TILE (gather Zx[u] + Zx[v], h Kh on the matrix cores, gates, message MLP, message rows out:
//     the body of lnlstm_mlp_fwd_h2_kernel, bit for bit) or a SHARE of the V<-E row-sum (<= 8 vertex rows, the summation
//     order of csr_rowsum_kernel; its edge lists wait in LDS, two rows' loads in flight at once).  Any wavefront takes any
//     item: three wavefronts per SIMD cover each other's latency chains within a step AND across the step boundary; the
//     next ticket is taken while the current item still computes.  A tile's states h, c live in slot arrays private to the
//     launch ([slots * 16, d], blocked by tile, in place), stored write-through and loaded past the L1 (a plain load was
//     measured to return a stale line at 256 instances).  A tile of step t waits for its own step t-1 (one LDS word per
//     tile, set by whoever ran it) and for its group's projected messages Zx_t (device counter; the value last seen is
//     cached in LDS per group, and the wavefront that saw it change invalidates the CU's L1 once -- the gathers of the other
//     wavefronts keep their L1 reuse).  A finished tile is PUBLISHED (its LDS word, its group's message counter) behind the
//     GEMM of the wavefront's next tile, where its stores have long drained -- not behind a wait of its own.
//   * The items of a step are ordered BY CLASS (plan): the groups of an XCD are cut into two classes, and a workgroup's
//     step is [row-sum shares A][tiles A][row-sum shares B][tiles B].  The vertex chain of a class -- row-sum, vertex cells,
//     message MLP, projection, and one tile latency before it can start -- then has 1.5 step times to complete instead
//     of one: the other class's tiles run meanwhile (measured: one class 49 us per C2 step, two 42-44, three 47).
//   * VERTEX workgroups come in two kinds, neither with a barrier or a second LDS residency inside the loop: CELL
//     workgroups keep K[2d,4d] (128 KB) and run the vertex cells, MESSAGE workgroups keep the message MLP + the projection
//     matrix (113 KB) and turn h' into the projected messages; h' crosses between them through the vertex states' parity
//     buffers `vh` and a fourth counter per group.  (The lock-step form with both residencies in one workgroup -- mp_loop's --
//     spent 15 of its 35 us per step in barriers and re-staging.)  Every vertex wavefront owns <= 2 tiles for the whole loop.
//   * Synchronisation: the three parity-split monotone counters per group of mp_loop_h2.hip (message tiles arrived, vertex
//     rows aggregated, vertex tiles projected) + one (vertex tiles updated); same thresholds, same ordering argument
//     (tests/test_resident_plan.py replays the protocol under random schedules); hand-offs through loop_sync.h.
//     Deadlock freedom: an item of step t waits only for items of steps < t, every workgroup takes its items in ticket
//     order, and all `grid` workgroups are resident (the entry point asks the runtime's occupancy query) -- by induction
//     on the step every wait ends.  Every wait is bounded all the same (status word).
//
// Plan (int32, tspgnn/resident_plan.py): grid headers of 8 ints
//     [0] role 0 idle / 1 edge / 2 vertex cells / 3 vertex messages   [1] first item   [2] items per step (edge: W; vertex: tiles)
//     [3] first state slot (edge)           [4] first group touched (edge: base of the LDS cache of Zx counters)
//     [5] tiles of the workgroup (edge: LDS words)   [6] class (vertex)
// then items of 8 ints
//     edge tile:    [0] first row  [1] valid rows  [2] group  [3] local tile index >= 0  [4] vertex tiles of the group
//     share:        [0] v0         [1] v1          [2] group  [3] -1                     [4] edge tiles of the group
//                   [5] offset (ints) of the share's edge-list block in the workgroup's LDS share area, or -1
//                   [6] first edge row of the group (the block holds 16-bit offsets from it)
//     vertex tile:  [0] first row  [1] valid rows  [2] group  [3] -2                     [4] cells: vertex rows of the group;
//                                                                                            messages: vertex tiles of the group
#include "common.h"
#include "h2_tile.h"
#include "loop_sync.h"
#include "mfma_tile.h"

#include <type_traits>

namespace tspgnn {

constexpr int kResWaves = TSPGNN_RESIDENT_WAVES;
constexpr int kResHdr = TSPGNN_RESIDENT_HDR_INTS;
constexpr int kResItem = TSPGNN_RESIDENT_ITEM_INTS;
constexpr int kResSlots = 32;       // groups per edge workgroup whose Zx counters are cached in LDS (others poll memory)
constexpr int kShareRows = TSPGNN_RESIDENT_SHARE_ROWS;   // vertex rows per row-sum share
constexpr int kShareCap = TSPGNN_RESIDENT_SHARE_CAP;     // edge ids per vertex row held in LDS
// An LDS flag word, read / written RELAXED behind a compiler barrier.  The ordering the protocol needs comes from the hardware's
// program order per wavefront (what follows is control-dependent on the value, and the writer drains its stores -- vmcnt(0) --
// before it writes the word); an ACQUIRE at workgroup scope would add an s_waitcnt vmcnt(0), i.e. make the wavefront sit out
// every load it has just issued (the endpoints' prefetch: ~1 us per tile, tools/resident_trace.py).
__device__ __forceinline__ unsigned lds_word(const unsigned* w) {
    asm volatile("" ::: "memory");
    const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_word_set(unsigned* w, unsigned v) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// Wave-uniform wait until the LDS word *w >= target (acquire at workgroup scope: what follows is not hoisted above it).
__device__ __forceinline__ void lds_wait_ge(const unsigned* w, unsigned target, bool& dead, unsigned* status) {
    if (dead) return;
    unsigned spins = 0;
    for (;;) {
        const unsigned v = lds_word(w);
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0u) {
            const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_word(status));
            if (s != 0u || spins > 8u * kSpinLimit) {
                dead = true;
                if ((threadIdx.x & 63) == 0) atomicOr(status, 1u);
                break;
            }
        }
    }
}


template <int D, bool CENTERED, bool TRACE>
__global__ __launch_bounds__(kResWaves * 64) void mp_resident_h2_kernel(const tspgnn_mp_resident_args a) {
    constexpr int TPG = D / 16, NT4 = D / 4, KBH = D / 32;
    constexpr int LAYER_BYTES = 2 * D * D * 2 + D * 4;   // { hi, lo, bias } of one MLP layer
    constexpr bool SWAP = H2_LN_SWAP != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    float* lds_ln = reinterpret_cast<float*>(ldsb);
    int* ctl = reinterpret_cast<int*>(lds_ln + 10 * D);          // [0] the ticket
    unsigned* zxc = reinterpret_cast<unsigned*>(ctl + 4);        // [2][kResSlots]: Zx counter values last seen, by parity
    unsigned char* lds_wb = ldsb + (10 * D + 4 + 2 * kResSlots) * sizeof(float);
    _Float16* lds_w = reinterpret_cast<_Float16*>(lds_wb);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nthreads = kResWaves * 64;
    const int T = a.T;
    const int* hdr = a.plan + (size_t)blockIdx.x * kResHdr;
    const int role = __builtin_amdgcn_readfirstlane(hdr[0]);
    const int item0 = __builtin_amdgcn_readfirstlane(hdr[1]);
    const int n_items = __builtin_amdgcn_readfirstlane(hdr[2]);
    const int* items = a.plan + (size_t)a.grid * kResHdr;
    unsigned* const counters = a.counters;
    auto cnt_msg = [&](int grp, int par) { return counters + (size_t)(grp * 4 + 0) * 32 + par * 16; };
    auto cnt_vagg = [&](int grp, int par) { return counters + (size_t)(grp * 4 + 1) * 32 + par * 16; };
    auto cnt_zx = [&](int grp, int par) { return counters + (size_t)(grp * 4 + 2) * 32 + par * 16; };
    auto cnt_vh = [&](int grp, int par) { return counters + (size_t)(grp * 4 + 3) * 32 + par * 16; };
    bool dead = false;
    float wit = 0.f;
    unsigned vmin = 0xffffffffu;
    if (role == 0 || n_items == 0) return;
    // PLACEMENT.  The plan puts everything a group needs on workgroups of one residue b mod 8, and workgroup b runs on XCD
    // b mod 8 (observed; nothing promises it).  The edge workgroups lean on that for ONE thing: before they gather this
    // parity's projected messages again they invalidate the CU's L1 only (buffer_inv sc0) -- the producer's write-through
    // stores are then visible because producer and consumer share an L2.  So the launch CHECKS it: every active workgroup
    // registers its XCC id under its residue; two ids under one residue (or flags & 1, the tests' switch) raise `mismatch`,
    // and the edge workgroups, which wait for all registrations before their first item, then take the agent-scope acquire
    // (L2 included) instead -- slower, correct wherever the workgroups run.
    unsigned* place = counters + (size_t)a.n_groups * 4 * 32;   // [0..7] XCC id + 1 of residue r, [8] registered, [9] mismatch
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        id = (id & 0xfu) + 1u;
        const unsigned old = atomicCAS(place + (blockIdx.x & 7), 0u, id);
        if ((old != 0u && old != id) || (a.flags & 1)) atomicOr(place + 9, 1u);
        __threadfence();
        atomicAdd(place + 8, 1u);
    }
    LoopTrace<TRACE, 32> tr;   // 0..7 phase sums, 8..15 stamps of one step (per-XCD timeline), 16..31 stamps inside one tile
    tr.begin(a.trace + ((size_t)blockIdx.x * kResWaves + wave) * 32);

    // LayerNorm parameters of this workgroup's cell (as lnlstm_mlp_fwd_h2_kernel: gates i, f, o times -log2(e), forget bias
    // folded into b_f)
    {
        const float* ln = role == 1 ? a.e_ln : a.v_ln;   // (a message workgroup has no cell: unused there)
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

    // ---- V<-E row-sum SHARES (held by the edge workgroups, among their items).  A workgroup's shares never change: their edge lists
    // wait in LDS -- per share a block of kShareRows counts (int) + kShareRows x kShareCap edge ids as 16-bit offsets from the
    // group's first edge row (item [6]); item [5] = the block's offset in ints from `shl`, < 0: no block (the general loop).
    int* shl = nullptr;   // set by the role, behind what it keeps in LDS
    auto fill_share_blocks = [&](int first, int count) {
        for (int j = wave; j < count; j += kResWaves) {
            const int* it = items + (size_t)(first + j) * kResItem;
            const int v0 = __builtin_amdgcn_readfirstlane(it[0]), v1 = __builtin_amdgcn_readfirstlane(it[1]);
            const int off = __builtin_amdgcn_readfirstlane(it[5]), e0 = __builtin_amdgcn_readfirstlane(it[6]);
            if (__builtin_amdgcn_readfirstlane(it[3]) != -1 || off < 0) continue;
            int* blk = shl + off;
            unsigned short* el = reinterpret_cast<unsigned short*>(blk + kShareRows);
            for (int q = 0; q < kShareRows; ++q) {
                const int v = v0 + q;
                const int beg = v < v1 ? a.rowptr[v] : 0;
                const int cnt = v < v1 ? a.rowptr[v + 1] - beg : 0;
                if (lane == 0) blk[q] = cnt;
                if (lane < cnt && lane < kShareCap) el[q * kShareCap + lane] = (unsigned short)(a.eid[beg + lane] - e0);
            }
        }
    };
    // the share [i0, i1) of group grp at step t (parity p): waits for the group's message tiles of step t - 1
    auto do_share = [&](int i0, int i1, int grp, int gcnt, int soff_i, int e0, int t) {
        const int p = t & 1;
        const __amdgpu_buffer_rsrc_t r_msg_in = p ? r_msg1 : r_msg0;
        const __amdgpu_buffer_rsrc_t r_vagg = p ? r_vagg1 : r_vagg0;
        wait_ge(cnt_msg(grp, p), (unsigned)(((t + 1) >> 1) * gcnt), dead, a.status);
        asm volatile("" ::: "memory");
        tr.mark(2);
        tr.stamp(5, t == (T >> 1));
        const int l = opaque_lane();
        const int sub = l >> 4, c = l & 15;
        bool fast = soff_i >= 0;
        const int* blk = shl + (soff_i >= 0 ? soff_i : 0);
        if (fast) {
#pragma unroll
            for (int q = 0; q < kShareRows; ++q) fast = fast && __builtin_amdgcn_readfirstlane(blk[q]) <= kShareCap;
        }
        if (fast) {
            // csr_rowsum_kernel's own form, two vertex rows at a time: lane group `sub` sums the edges k = sub (mod 4) in
            // ascending order -- every load of the pair in flight at once (a slot beyond the row's edges re-reads edge 0
            // and is dropped) -- then the fixed butterfly
            constexpr int NL = kShareCap / 4;
            const int rows = i1 - i0;
            const unsigned short* els = reinterpret_cast<const unsigned short*>(blk + kShareRows);
            for (int q0 = 0; q0 < rows; q0 += 2) {
                f32x4 x[2][NL];
                int cn[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = q0 + u < rows ? q0 + u : q0;
                    cn[u] = __builtin_amdgcn_readfirstlane(blk[q]);
                    const unsigned short* el = els + q * kShareCap;
#pragma unroll
                    for (int m = 0; m < NL; ++m) {
                        const int kk = 4 * m + sub;
                        const int e = e0 + (int)el[kk < cn[u] ? kk : 0];
                        x[u][m] = ld4wt(r_msg_in, ((unsigned)e * D + c * 4) * 4u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < NL; ++m) {
                        const f32x4 add = 4 * m + sub < cn[u] ? x[u][m] : f32x4{0.f, 0.f, 0.f, 0.f};
                        if (4 * m < cn[u]) s4 += add;   // (uniform: a row of fewer edges takes fewer additions, as csr_rowsum_kernel)
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s4[r] += __shfl_xor(s4[r], 16);
                        s4[r] += __shfl_xor(s4[r], 32);
                    }
                    if (sub == 0 && q0 + u < rows) st4wt(r_vagg, ((unsigned)(i0 + q0 + u) * D + c * 4) * 4u, s4);
                }
            }
        } else {
            for (int vb = i0; vb < i1; vb += 4) {
                const int v = vb + sub;
                const bool on = v < i1;
                const int vv = on ? v : i1 - 1;
                const int beg = a.rowptr[vv], cnt = a.rowptr[vv + 1] - beg;
                int mx = cnt;
                mx = max(mx, __shfl_xor(mx, 16));
                mx = max(mx, __shfl_xor(mx, 32));
                f32x4 s[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int base = 0; base < mx; base += 16) {
                    const int my_e = (base + c < cnt) ? a.eid[beg + base + c] : 0;
#pragma unroll
                    for (int h8 = 0; h8 < 16; h8 += 8) {
                        f32x4 x[8];
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            const int e = __shfl(my_e, (l & 48) + h8 + kk);
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
        arrive(cnt_vagg(grp, p), (unsigned)(i1 - i0));
        tr.mark(3);
        tr.stamp(6, t == (T >> 1));
    };

    if (role == 1) {
        // ------------------------------------------------------------------------------------------- edge workgroup
        constexpr int total = D * 4 * D;   // elements per piece of Kh
        unsigned char* lds_mlp = lds_wb + (size_t)2 * total * 2;
        const int L = a.e_mlp_layers;
        unsigned* done = reinterpret_cast<unsigned*>(lds_mlp + (size_t)L * LAYER_BYTES);   // [tiles]: steps completed
        const int slot_base = __builtin_amdgcn_readfirstlane(hdr[3]);
        const int g_first = __builtin_amdgcn_readfirstlane(hdr[4]);
        const int n_local = __builtin_amdgcn_readfirstlane(hdr[5]);
        h2_copy_to_lds(lds_w, a.e_K, 2 * total * 2, tid, nthreads);
        if (L > 0) h2_copy_to_lds(lds_mlp, a.e_mlp_wb, L * LAYER_BYTES, tid, nthreads);
        for (int i = tid; i < n_local; i += nthreads) done[i] = 0u;
        for (int i = tid; i < 2 * kResSlots; i += nthreads) zxc[i] = 0u;
        if (tid == 0) ctl[0] = 0;
        h2_stage_wait();
        __syncthreads();
        const _Float16* Kl = lds_w + total;
        const int2* uv = reinterpret_cast<const int2*>(a.uv);
        const __amdgpu_buffer_rsrc_t r_hs = make_rsrc(a.e_hs, (long long)a.n_slots * 16 * D * 4);
        const __amdgpu_buffer_rsrc_t r_cs = make_rsrc(a.e_cs, (long long)a.n_slots * 16 * D * 4);
        const int total_items = n_items * T;

        shl = reinterpret_cast<int*>(done + n_local);   // (an edge workgroup's share blocks: behind the per-tile words)
        fill_share_blocks(item0, n_items);
        if (wave == 0) {    // all registrations in (see PLACEMENT above), then the verdict for the whole workgroup
            wait_ge(place + 8, (unsigned)a.n_active, dead, a.status);
            const unsigned bad = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_word(place + 9));
            if (lane == 0) ctl[1] = (int)bad;
        }
        __syncthreads();
        const bool l2_shared = __builtin_amdgcn_readfirstlane(ctl[1]) == 0;
        tr.mark(0);

        // The tile this wavefront ran last is PUBLISHED (its LDS word, its group's message counter) only when its stores have
        // drained -- and that wait is not taken where the tile ends but behind the loads of the wavefront's NEXT tile, which
        // must be waited for anyway (vmcnt counts loads and stores alike).  It is taken at once whenever the next item could
        // depend on the publication: a later step's item, a share, a wait that does not succeed at first try.
        int pend_local = -1, pend_t = 0, pend_grp = 0;
        auto publish = [&]() {
            if (pend_local >= 0) {
                lds_word_set(done + pend_local, (unsigned)(pend_t + 1));   // (behind the caller's vmcnt(0))
                arrive(cnt_msg(pend_grp, 1 - (pend_t & 1)), 1u);
                pend_local = -1;
            }
        };
        auto flush = [&]() {
            if (pend_local >= 0) {
                drain_stores();
                publish();
            }
        };

        // The ticket of the NEXT item is taken while the current one still computes (take_next: before a tile's message MLP,
        // before a share's loads) -- the LDS round trip and the scalar loads of the item then cost nothing at the loop's top.
        // Tickets are still taken in order by every wavefront, and an item never depends on a later one.
        int k_next = 0;
        auto take_next = [&]() {
            if (lane == 0) k_next = atomicAdd(ctl, 1);
        };
        take_next();
        for (;;) {
            const int k = __builtin_amdgcn_readfirstlane(k_next);
            if (k >= total_items) break;
            const int t = k / n_items, j = k - t * n_items;
            const int* it = items + (size_t)(item0 + j) * kResItem;
            const int i0 = __builtin_amdgcn_readfirstlane(it[0]), i1 = __builtin_amdgcn_readfirstlane(it[1]);
            const int grp = __builtin_amdgcn_readfirstlane(it[2]), local = __builtin_amdgcn_readfirstlane(it[3]);
            const int gcnt = __builtin_amdgcn_readfirstlane(it[4]);
            const int p = t & 1;
            const bool last = t == T - 1;
            tr.mark(1);
            const bool chosen = t == (T >> 1);
            tr.stamp(local < 0 ? 4 : 0, chosen);
            const bool fine = chosen && local >= 0;
            tr.stamp(8, fine);    // item decoded

            if (local < 0) {
                // ---- a share of the V<-E row-sum over the messages of step t: vertex rows [i0, i1) of group grp
                const int soff_i = __builtin_amdgcn_readfirstlane(it[5]), e0 = __builtin_amdgcn_readfirstlane(it[6]);
                flush();
                take_next();
                do_share(i0, i1, grp, gcnt, soff_i, e0, t);
                continue;
            }

            // ---- the edge cells of step t on the 16-row tile [i0, i0 + i1) (+ the messages of step t + 1)
            const bool first = t == 0;
            if (pend_local >= 0 && pend_t < t) flush();
            int2 ends;   // the row's endpoints (static data): fetched before the waits
            {
                const int l = opaque_lane();
                const int rl0 = l & 15;
                ends = uv[i0 + (rl0 < i1 ? rl0 : 0)];
            }
            if (!first) {
                const unsigned v = lds_word(done + local);
                if (v < (unsigned)t) {
                    flush();
                    lds_wait_ge(done + local, (unsigned)t, dead, a.status);
                }
            }
            tr.mark(4);
            tr.stamp(1, chosen);
            tr.stamp(9, fine);    // own previous step ready
            {
                const unsigned tgt = (unsigned)(((t + 1) >> 1) * gcnt);
                if (tgt != 0u) {
                    const int slot = grp - g_first;
                    unsigned seen = 0u;
                    if (slot < kResSlots) seen = lds_word(zxc + p * kResSlots + slot);
                    if (seen < tgt) {
                        if ((unsigned)__builtin_amdgcn_readfirstlane((int)ld_word(cnt_zx(grp, p))) < tgt) {
                            flush();
                            wait_ge(cnt_zx(grp, p), tgt, dead, a.status);
                        }
                        // the CU's L1 may hold lines of this parity's Zx from two steps ago: invalidate once, let the
                        // invalidation pass the L1 (a dependent L1-bypassing load behind it), then tell the siblings
                        if (l2_shared) {
                            asm volatile("buffer_inv sc0" ::: "memory");
                        } else {
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        }
                        const unsigned again = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_word(cnt_zx(grp, p)));
                        if (slot < kResSlots && again >= tgt) lds_word_set(zxc + p * kResSlots + slot, tgt);
                    }
                }
            }
            tr.mark(5);
            tr.stamp(2, chosen);
            tr.stamp(10, fine);   // Zx ready
            const float* zx = a.zx[p];
            // the lane's coordinates, recomputed where they are needed instead of carried across the GEMMs (two VALU
            // instructions against a spilled register): (a lane beyond the tile's rows repeats row 0, state slot included --
            // it must never read a slot row nobody wrote)
            auto coords = [&](int& rl, int& g, bool& valid, unsigned& rc, unsigned& soff) {
                const int l = opaque_lane();
                rl = l & 15;
                g = l >> 4;
                valid = rl < i1;
                rc = (unsigned)(i0 + (valid ? rl : 0));
                soff = h2_state_row<D>((unsigned)(slot_base + local) * 16u + (unsigned)(valid ? rl : 0), g, true);   // floats
            };
            f32x4 hn[TPG];
            {
                int rl, g;
                bool valid;
                unsigned rc, soff;
                coords(rl, g, valid, rc, soff);
                f32x4 acc[NT4], cf[TPG];
                {
                    const float* zu = zx + h2_zx_row<D>((unsigned)ends.x, g);
                    const float* zv = zx + h2_zx_row<D>((unsigned)ends.y, g);
#pragma unroll
                    for (int q = 0; q < NT4; ++q) acc[q] = ld4(zu + q * 256);
#pragma unroll
                    for (int q = 0; q < NT4; ++q) acc[q] += ld4(zv + q * 256);
                }
                f32x4 hv[TPG];
                if (first) {
                    const float* hr = a.e_h0 + ((size_t)rc * D + g * 4);
#pragma unroll
                    for (int q = 0; q < TPG; ++q) hv[q] = ld4(hr + q * 16);
#pragma unroll
                    for (int q = 0; q < TPG; ++q)
                        cf[q] = a.e_c0 != nullptr ? ld4(a.e_c0 + ((size_t)rc * D + g * 4 + q * 16)) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    // (past the L1: a plain load here was measured to return a line another wavefront of the CU had since
                    // rewritten -- 256 instances of n = 40, max abs difference 4.8e-2 -- DESIGN_HISTORY round 6)
#pragma unroll
                    for (int q = 0; q < TPG; ++q) hv[q] = ld4wt(r_hs, (soff + q * 256u) * 4u);
#pragma unroll
                    for (int q = 0; q < TPG; ++q) cf[q] = ld4wt(r_cs, (soff + q * 256u) * 4u);
                }
                if constexpr (TRACE) {
                    if (fine) {
                        tr.stamp(11, true);   // loads issued
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        tr.stamp(12, true);   // loads landed
                    }
                }
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) {
                    float xv[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) xv[jj] = hv[2 * kb + (jj >> 2)][jj & 3];
                    f16x8 bh, bl;
                    split2w(xv, bh, bl, wit);
                    kblock_h2<NT4>(acc, lds_w, Kl, kb, g, rl, bh, bl);
                }
                if constexpr (TRACE) {
                    if (fine) {
                        asm volatile("s_nop 0" : : "v"(acc[0]), "v"(acc[NT4 - 1]));
                        tr.stamp(13, true);   // K GEMM done
                    }
                }
                if (pend_local >= 0) {   // every load of this tile has landed, the previous tile's stores are microseconds old:
                    drain_stores();      // the wait is (all but) free here
                    publish();
                }
                f32x4 nc[TPG];
                lstm_gates<D, true, SWAP, CENTERED, true>(acc, cf, lds_ln, g, hn, nc, kH2GateEps, &vmin);
                if constexpr (TRACE) {
                    if (fine) {
                        asm volatile("s_nop 0" : : "v"(hn[0]), "v"(nc[TPG - 1]));
                        tr.stamp(14, true);   // gates done
                    }
                }
                coords(rl, g, valid, rc, soff);
                if (last) {
                    if (valid) {
                        float* hd = a.e_h + ((size_t)rc * D + g * 4);
                        float* cd = a.e_c + ((size_t)rc * D + g * 4);
#pragma unroll
                        for (int q = 0; q < TPG; ++q) {
                            st4(hd + q * 16, hn[q]);
                            st4(cd + q * 16, nc[q]);
                        }
                    }
                } else if (valid) {
#pragma unroll
                    for (int q = 0; q < TPG; ++q) {
                        st4wt(r_hs, (soff + q * 256u) * 4u, hn[q]);
                        st4wt(r_cs, (soff + q * 256u) * 4u, nc[q]);
                    }
                }
            }
            tr.mark(6);
            tr.stamp(15, fine);   // state stores issued
            take_next();
            if (!last) {
                int rl, g;
                {
                    const int l = opaque_lane();
                    rl = l & 15;
                    g = l >> 4;
                }
                const unsigned mask = a.e_relu_mask;
                for (int ly = 0; ly < L; ++ly) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_mlp + (size_t)ly * LAYER_BYTES);
                    const float* bias = reinterpret_cast<const float*>(lds_mlp + (size_t)ly * LAYER_BYTES + 2 * D * D * 2);
                    dense_layer_h2<D>(hn, wh, wh + D * D, bias, (mask >> ly) & 1u, g, rl, wit);
                }
                if (L > 0) {
                    bool valid;
                    unsigned rc, soff;
                    coords(rl, g, valid, rc, soff);
                    const __amdgpu_buffer_rsrc_t r_msg_out = p ? r_msg0 : r_msg1;
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < TPG; ++q) st4wt(r_msg_out, (rc * D + g * 4 + q * 16) * 4u, hn[q]);
                    }
                }
                pend_local = local;
                pend_t = t;
                pend_grp = grp;
            }
            tr.stamp(16, fine);   // message MLP done, message stores issued
            tr.mark(7);
            tr.stamp(3, chosen);
        }
        flush();
        tr.flush();
        h2_range_report(a.range_flag, wit, vmin);
        return;
    }

    // ------------------------------------------------------------------------------------- vertex workgroups (two kinds)
    // No barrier after the staging, no second residency: a CELL workgroup (role 2) keeps K[2d,4d] in LDS and runs the vertex
    // cells, a MESSAGE workgroup (role 3) keeps the message MLP and the projection matrix and turns h' into the projected
    // messages the edge cells gather; h' crosses between them through the vertex states' parity buffers (vh) and a fourth
    // counter per group.  Every wavefront owns its tiles (wave, wave + 12) for the whole loop and waits for nobody but its
    // own tile's inputs.
    // tile s of this wavefront = tile (wave + s * waves) of the workgroup
    // A vertex wavefront's tiles: (wave + 12 s) of its workgroup, s = 0, 1, ...; ONE tile body per role, its parameters read per
    // round (templated copies per tile slot cost registers across the whole kernel).
    const int n_rounds = (n_items + kResWaves - 1) / kResWaves;
    const __amdgpu_buffer_rsrc_t r_vh0 = make_rsrc(a.vh[0], (long long)a.N * D * 4);
    const __amdgpu_buffer_rsrc_t r_vh1 = make_rsrc(a.vh[1], (long long)a.N * D * 4);
    if (role == 2) {
        constexpr int KBT = 2 * KBH;           // k-blocks of [x | h]
        constexpr int total = 2 * D * 4 * D;   // elements per piece of K[2D, 4D]
        h2_copy_to_lds(lds_w, a.v_K, 2 * total * 2, tid, nthreads);
        h2_stage_wait();
        __syncthreads();
        tr.mark(0);
        for (int t = 0; t < T; ++t) {
            const int p = t & 1;
            const bool last = t == T - 1;
            const bool chosen = t == (T >> 1);
            const __amdgpu_buffer_rsrc_t r_vagg = p ? r_vagg1 : r_vagg0;
            // step t reads h from vh[p] (the caller's states at step 0) and writes h' to vh[1 - p] (the caller's output at the
            // last step); c is updated in place
            const __amdgpu_buffer_rsrc_t r_h_in = p ? r_vh1 : r_vh0;
            const __amdgpu_buffer_rsrc_t r_h_out = p ? r_vh0 : r_vh1;
            const float* c_in = t == 0 ? a.v_c0 : a.v_c;
            auto cell = [&](int idx) {
                const int* it = items + (size_t)(item0 + idx) * kResItem;
                const int row0_j = __builtin_amdgcn_readfirstlane(it[0]), nvalid_j = __builtin_amdgcn_readfirstlane(it[1]);
                const int grp_j = __builtin_amdgcn_readfirstlane(it[2]), gcnt_j = __builtin_amdgcn_readfirstlane(it[4]);
                // the part of z that does not wait: bias-init and h K[d:2d]
                int rl, g;
                {
                    const int l = opaque_lane();
                    rl = l & 15;
                    g = l >> 4;
                }
                const bool valid = rl < nvalid_j;
                const unsigned rc = (unsigned)(row0_j + (valid ? rl : 0));
                f32x4 ho[TPG];
                if (t == 0) {
#pragma unroll
                    for (int q = 0; q < TPG; ++q) ho[q] = ld4(a.v_h0 + (size_t)rc * D + g * 4 + q * 16);
                } else {
#pragma unroll
                    for (int q = 0; q < TPG; ++q) ho[q] = ld4wt(r_h_in, (rc * D + g * 4 + q * 16) * 4u);
                }
                wait_ge(cnt_vagg(grp_j, p), (unsigned)(((t >> 1) + 1) * gcnt_j), dead, a.status);
                asm volatile("" ::: "memory");
                tr.stamp(0, chosen);
                f32x4 xo[TPG];
#pragma unroll
                for (int q = 0; q < TPG; ++q) xo[q] = ld4wt(r_vagg, (rc * D + g * 4 + q * 16) * 4u);
                tr.mark(1);
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
                        x[jj] = kb < KBH ? xo[2 * kb + (jj >> 2)][jj & 3] : ho[2 * (kb - KBH) + (jj >> 2)][jj & 3];
                    f16x8 bh, bl;
                    split2w(x, bh, bl, wit);
                    kblock_h2<NT4>(acc, lds_w, lds_w + total, kb, g, rl, bh, bl);
                }
#pragma unroll
                for (int q = 0; q < TPG; ++q)
                    cf[q] = c_in != nullptr ? ld4(c_in + (size_t)rc * D + g * 4 + q * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 hn[TPG], nc[TPG];
                lstm_gates<D, true, SWAP, CENTERED, true>(acc, cf, lds_ln, g, hn, nc, kH2GateEps, &vmin);
                if (valid) {
                    float* cd = a.v_c + (size_t)rc * D + g * 4;
#pragma unroll
                    for (int q = 0; q < TPG; ++q) st4(cd + q * 16, nc[q]);
                    if (last) {
                        float* hd = a.v_h + (size_t)rc * D + g * 4;
#pragma unroll
                        for (int q = 0; q < TPG; ++q) st4(hd + q * 16, hn[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < TPG; ++q) st4wt(r_h_out, (rc * D + g * 4 + q * 16) * 4u, hn[q]);
                    }
                }
                tr.mark(2);
                if (!last) {
                    drain_stores();
                    arrive(cnt_vh(grp_j, 1 - p), 1u);
                }
                tr.stamp(2, chosen);
                tr.mark(3);
            };
            for (int s_ = 0; s_ < n_rounds; ++s_) {
                const int j = wave + s_ * kResWaves;
                if (j < n_items) cell(j);
            }
        }
        tr.flush();
        h2_range_report(a.range_flag, wit, vmin);
        return;
    }
    {
        const int L = a.v_mlp_layers;
        unsigned char* lds_proj = lds_wb + (size_t)L * LAYER_BYTES;
        const __amdgpu_buffer_rsrc_t r_zx0 = make_rsrc(a.zx[0], (long long)((a.N + 15) / 16) * 16 * 4 * D * 4);
        const __amdgpu_buffer_rsrc_t r_zx1 = make_rsrc(a.zx[1], (long long)((a.N + 15) / 16) * 16 * 4 * D * 4);
        h2_copy_to_lds(lds_wb, a.v_mlp_wb, L * LAYER_BYTES, tid, nthreads);
        h2_copy_to_lds(lds_proj, a.v_proj_w, 2 * D * 4 * D * 2, tid, nthreads);
        h2_stage_wait();
        __syncthreads();
        tr.mark(0);
        for (int t = 0; t + 1 < T; ++t) {
            const int p = t & 1;
            const bool chosen = t == (T >> 1);
            const __amdgpu_buffer_rsrc_t r_h = p ? r_vh0 : r_vh1;      // h' of step t lives in vh[1 - p]
            const __amdgpu_buffer_rsrc_t r_zx_out = p ? r_zx0 : r_zx1;
            auto message = [&](int idx) {
                const int* it = items + (size_t)(item0 + idx) * kResItem;
                const int row0_j = __builtin_amdgcn_readfirstlane(it[0]), nvalid_j = __builtin_amdgcn_readfirstlane(it[1]);
                const int grp_j = __builtin_amdgcn_readfirstlane(it[2]), gcnt_j = __builtin_amdgcn_readfirstlane(it[4]);
                wait_ge(cnt_vh(grp_j, 1 - p), (unsigned)(((t >> 1) + 1) * gcnt_j), dead, a.status);
                asm volatile("" ::: "memory");
                tr.mark(1);
                tr.stamp(3, chosen);
                const int l = opaque_lane();
                const int rl = l & 15, g = l >> 4;
                const bool valid = rl < nvalid_j;
                const unsigned rc = (unsigned)(row0_j + (valid ? rl : 0));
                f32x4 hn[TPG];
#pragma unroll
                for (int q = 0; q < TPG; ++q) hn[q] = ld4wt(r_h, (rc * D + g * 4 + q * 16) * 4u);
                const unsigned mask = a.v_relu_mask;
                for (int ly = 0; ly < L; ++ly) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_wb + (size_t)ly * LAYER_BYTES);
                    const float* bias = reinterpret_cast<const float*>(lds_wb + (size_t)ly * LAYER_BYTES + 2 * D * D * 2);
                    dense_layer_h2<D>(hn, wh, wh + D * D, bias, (mask >> ly) & 1u, g, rl, wit);
                }
                // Zx = 2^s (y Kx), a gate (TPG column tiles) at a time on the same operand pieces
                f16x8 yh[KBH], yl[KBH];
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) {
                    float x[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) x[jj] = hn[2 * kb + (jj >> 2)][jj & 3];
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
                tr.mark(2);
                drain_stores();
                arrive(cnt_zx(grp_j, 1 - p), 1u);
                tr.mark(3);
                tr.stamp(5, chosen);
            };
            for (int s_ = 0; s_ < n_rounds; ++s_) {
                const int j = wave + s_ * kResWaves;
                if (j < n_items) message(j);
            }
        }
        tr.flush();
        h2_range_report(a.range_flag, wit, 0xffffffffu);
    }
}

}  // namespace tspgnn

using namespace tspgnn;

extern "C" int tspgnn_mp_resident_h2(const tspgnn_mp_resident_args* args, int d, void* stream) {
    TSPGNN_REQUIRE(args, "mp_resident_h2: null args");
    TSPGNN_REQUIRE(d == 64, "mp_resident_h2: d=%d must be 64", d);
    const tspgnn_mp_resident_args& a = *args;
    TSPGNN_REQUIRE(a.T >= 1, "mp_resident_h2: T=%d must be >= 1", a.T);
    TSPGNN_REQUIRE(a.M > 0 && a.N > 0 && a.n_groups > 0 && a.n_slots > 0, "mp_resident_h2: M=%d, N=%d, n_groups=%d, n_slots=%d",
                   a.M, a.N, a.n_groups, a.n_slots);
    TSPGNN_REQUIRE((long long)a.M * d * 4 < (1ll << 31) && ((long long)a.N + 16) * 4 * d * 4 < (1ll << 31) &&
                       (long long)a.n_slots * 16 * d * 4 < (1ll << 31),
                   "mp_resident_h2: M=%d / N=%d / n_slots=%d too large for 32-bit byte offsets", a.M, a.N, a.n_slots);
    TSPGNN_REQUIRE(a.grid >= 1 && a.grid <= n_cus(), "mp_resident_h2: grid=%d must be in 1..%d (one resident workgroup per CU)",
                   a.grid, n_cus());
    TSPGNN_REQUIRE(a.e_h0 && a.e_h && a.e_c && a.e_hs && a.e_cs && a.uv && a.e_K && a.e_ln && a.msg[0] && a.msg[1],
                   "mp_resident_h2: null edge pointer");
    TSPGNN_REQUIRE(a.v_h0 && a.v_h && a.v_c && a.rowptr && a.eid && a.v_K && a.v_ln && a.zx[0] && a.zx[1] && a.vagg[0] &&
                       a.vagg[1] && a.vh[0] && a.vh[1],
                   "mp_resident_h2: null vertex pointer");
    TSPGNN_REQUIRE(a.plan && a.counters && a.status, "mp_resident_h2: null plan / counters / status");
    TSPGNN_REQUIRE(a.e_mlp_layers >= 0 && a.e_mlp_layers <= 3 && (a.e_mlp_layers == 0 || a.e_mlp_wb),
                   "mp_resident_h2: e_mlp_layers=%d must be in 0..3 (resident next to Kh)", a.e_mlp_layers);
    TSPGNN_REQUIRE(a.v_mlp_layers >= 1 && a.v_mlp_layers <= 4 && a.v_mlp_wb && a.v_proj_w,
                   "mp_resident_h2: v_mlp_layers=%d must be in 1..4, with a projection", a.v_mlp_layers);
    TSPGNN_REQUIRE(!a.v_zbias || a.v_zscale, "mp_resident_h2: v_zbias needs v_zscale");
    TSPGNN_REQUIRE(a.e_h0 != a.e_h && a.v_h0 != a.v_h, "mp_resident_h2: the final states must not alias the initial ones");
    TSPGNN_REQUIRE(a.lds_words >= 0, "mp_resident_h2: lds_words=%d", a.lds_words);
    TSPGNN_REQUIRE(a.n_active >= 1 && a.n_active <= a.grid, "mp_resident_h2: n_active=%d must be in 1..grid", a.n_active);
    constexpr int D = 64;
    const size_t head = (10 * D + 4 + 2 * kResSlots) * sizeof(float);
    const size_t layer = 2 * D * D * 2 + D * 4;
    const size_t edge_bytes = (size_t)2 * D * 4 * D * 2 + a.e_mlp_layers * layer + (size_t)a.lds_words * 4;
    const size_t vert_k = (size_t)2 * 2 * D * 4 * D * 2, vert_m = a.v_mlp_layers * layer + (size_t)2 * D * 4 * D * 2;
    size_t lds_bytes = edge_bytes > vert_k ? edge_bytes : vert_k;
    if (vert_m > lds_bytes) lds_bytes = vert_m;
    lds_bytes += head;
    if (lds_bytes > (size_t)160 * 1024)
        return fail(TSPGNN_EUNSUPPORTED, "mp_resident_h2: %zu bytes of LDS (%d tiles in one workgroup)", lds_bytes, a.lds_words);
    void (*fn)(const tspgnn_mp_resident_args) =
        a.trace ? (a.z_centered ? &mp_resident_h2_kernel<D, true, true> : &mp_resident_h2_kernel<D, false, true>)
                : (a.z_centered ? &mp_resident_h2_kernel<D, true, false> : &mp_resident_h2_kernel<D, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail(TSPGNN_EUNSUPPORTED, "mp_resident_h2: hipFuncSetAttribute(%d B): %s", (int)lds_bytes, hipGetErrorString(e));
    // every wait inside the launch assumes all `grid` workgroups are resident at once: ask the runtime
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), kResWaves * 64, lds_bytes);
    if (e != hipSuccess || per_cu < 1)
        return fail(TSPGNN_EUNSUPPORTED, "mp_resident_h2: a workgroup of %d threads and %zu bytes of LDS is not resident on this device",
                    kResWaves * 64, lds_bytes);
    fn<<<a.grid, kResWaves * 64, lds_bytes, as_stream(stream)>>>(a);
    return launched("tspgnn_mp_resident_h2");
}
