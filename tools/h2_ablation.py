#!/usr/bin/env python
"""Development helper: writes a copy of csrc/dense_h2.hip with compile-time ablation switches (-DH2_ABL=mask) so that the
phases of the fused cell launch can be timed by removal (SRC=<copy> tools/build_variant.sh ...):
   1 no Zx gather (edge)   2 no K GEMM (edge)   4 no gates (edge)   8 no message MLP (edge)
  16 no lock-step (vertex) task   32 v-side gather rows made uniform (one cache line per quad)
  64 lock-step: no MLP / projection   128 lock-step: no K GEMM (staging kept)   256 lock-step: no gates
 512 lock-step: no K staging."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "tsp-gnn_amd", "csrc", "dense_h2.hip")).read()


def rep(old, new):
    global src
    assert old in src, old[:70]
    src = src.replace(old, new)


SKIP_GATES = '''{
#pragma unroll
                    for (int t = 0; t < TPG; ++t) hn[t] = acc[t] + acc[t + TPG] + acc[t + 2 * TPG] + acc[t + 3 * TPG] + cf[t];
                    if (valid) {
#pragma unroll
                        for (int t = 0; t < TPG; ++t) {
                            st4(h_out + h2_state_row<D>(rc, g, out_blk) + t * out_ts, hn[t]);
                            st4(c_out + h2_state_row<D>(rc, g, out_blk) + t * out_ts, hn[t]);
                        }
                    }
                }'''

rep('            const float* zv = Zx + h2_zx_row<D>((unsigned)ends.y, g);',
    '            const float* zv = Zx + h2_zx_row<D>((unsigned)((H2_ABL & 32) ? __builtin_amdgcn_readfirstlane(ends.y) : ends.y), g);')
# resident (edge) tile loop
rep('''                init_acc(acc, rc);
#pragma unroll
                for (int t = 0; t < TPG; ++t) cf[t] = ld4(c + h2_state_row<D>(rc, g, in_blk) + t * in_ts);
                kloop(acc, rc, 0, 0, KBT);
                cell(acc, cf, rc, valid, hn);
            }
            if (n_layers > 0) {
                for (int l = 0; l < n_layers; ++l) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_mlp''',
    '''                if constexpr (H2_ABL & 1) {
#pragma unroll
                    for (int t = 0; t < NT4; ++t) acc[t] = f32x4{0.1f * t, 0.2f, 0.3f * rl, 0.4f};
                } else
                    init_acc(acc, rc);
#pragma unroll
                for (int t = 0; t < TPG; ++t) cf[t] = ld4(c + h2_state_row<D>(rc, g, in_blk) + t * in_ts);
                if constexpr (!(H2_ABL & 2)) kloop(acc, rc, 0, 0, KBT);
                if constexpr (H2_ABL & 4) ''' + SKIP_GATES + ''' else
                    cell(acc, cf, rc, valid, hn);
            }
            if ((n_layers > 0) && !(H2_ABL & 8)) {
                for (int l = 0; l < n_layers; ++l) {
                    const _Float16* wh = reinterpret_cast<const _Float16*>(lds_mlp''')
# lock-step (vertex) rounds
rep('''        const int rounds = (tiles_total + lw - 1) / lw;
        for (int r = my_blk; r < rounds; r += my_grid) {''',
    '''        const int rounds = (H2_ABL & 16) ? 0 : (tiles_total + lw - 1) / lw;
        for (int r = my_blk; r < rounds; r += my_grid) {''')
rep('''                    __syncthreads();
                    stage(kb0, kb1);
                    h2_stage_wait();
                    __syncthreads();
                    if (live && pre) {''',
    '''                    __syncthreads();
                    if constexpr (!(H2_ABL & 512)) stage(kb0, kb1);
                    h2_stage_wait();
                    __syncthreads();
                    if (H2_ABL & 128) {
                    } else if (live && pre) {''')
rep('''                if (n_layers > 0) {  // every wavefront is done with K''', '''                if (n_layers > 0 && !(H2_ABL & 64)) {  // every wavefront is done with K''')
rep('''                cell(acc, cf, rc, valid, hn);
            }
            if (n_layers > 0) {
                h2_stage_wait();''',
    '''                if constexpr (H2_ABL & 256) ''' + SKIP_GATES + ''' else
                    cell(acc, cf, rc, valid, hn);
            }
            if (n_layers > 0 && !(H2_ABL & 64)) {
                h2_stage_wait();''')
rep('#include "mfma_tile.h"\n', '#include "mfma_tile.h"\n#ifndef H2_ABL\n#define H2_ABL 0\n#endif\n')
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "variants", "dense_h2_abl.hip")
os.makedirs(os.path.dirname(out), exist_ok=True)
open(out, "w").write(src)
print(out)
