// Development probe: issue cost (cycles per wave64 instruction per SIMD) of the VALU / MFMA instructions the
// fused cell kernel is made of, on gfx950.  One workgroup per CU, W waves per SIMD, each wave runs `rounds` x 64
// instances of ONE instruction over 8 independent register chains; cycles = wall time x clock / instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_rate_probe.hip -o /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM)                                                                        \
    __device__ __forceinline__ void NAME(float (&v)[8], float (&w)[8], float a, float b) {     \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                        \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { ASM; }                             \
        }                                                                                      \
    }

BODY(b_fma, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b)))
BODY(b_max, asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b)))
BODY(b_exp, asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])))
BODY(b_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i])))
BODY(b_rsq, asm volatile("v_rsq_f32 %0, %0" : "+v"(v[i])))
BODY(b_shl, asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[i])))
BODY(b_cvtbf, asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i])))
BODY(b_cvtf16, asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i])))
BODY(b_cvtrtz, asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i])))
BODY(b_cvtf32, asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i])))
BODY(b_mix, asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]" : "+v"(v[i]) : "v"(a), "v"(w[i])))
BODY(b_mixlo, asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,0]" : "+v"(v[i]) : "v"(w[i]), "v"(a)))
BODY(b_dpp, asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v[i])))
BODY(b_perm, asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v[i]) : "v"(w[i])))

__device__ __forceinline__ void b_pkfma(float (&v)[8], float (&w)[8], float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 x[4], aa = {a, a}, bb = {b, b};
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32x2{v[2 * i], v[2 * i + 1]};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(aa), "v"(bb));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = x[i][0]; v[2 * i + 1] = x[i][1]; }
}

template <int KIND>
__global__ __launch_bounds__(1024) void probe(float* __restrict__ out, int rounds) {
    const int tid = threadIdx.x, lane = tid & 63;
    float v[8], w[8];
    for (int i = 0; i < 8; ++i) { v[i] = 1.0f + 0.001f * (lane + i); w[i] = 0.5f + 0.002f * (lane - i); }
    const float a = 0.9999f, b = 0.0001f;
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 ha, hb;
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) {
        ha[i] = (_Float16)(0.01f * (lane + i)); hb[i] = (_Float16)(0.02f * (lane - i));
        ba[i] = (__bf16)(0.01f * (lane + i)); bb[i] = (__bf16)(0.02f * (lane - i));
    }
    for (int r = 0; r < rounds; ++r) {
        if constexpr (KIND == 0) b_fma(v, w, a, b);
        if constexpr (KIND == 1) b_pkfma(v, w, a, b);
        if constexpr (KIND == 2) b_exp(v, w, a, b);
        if constexpr (KIND == 3) b_rcp(v, w, a, b);
        if constexpr (KIND == 4) b_rsq(v, w, a, b);
        if constexpr (KIND == 5) b_max(v, w, a, b);
        if constexpr (KIND == 6) b_shl(v, w, a, b);
        if constexpr (KIND == 7) b_cvtbf(v, w, a, b);
        if constexpr (KIND == 8) b_cvtf16(v, w, a, b);
        if constexpr (KIND == 9) b_cvtrtz(v, w, a, b);
        if constexpr (KIND == 10) b_cvtf32(v, w, a, b);
        if constexpr (KIND == 11) b_mix(v, w, a, b);
        if constexpr (KIND == 12) b_mixlo(v, w, a, b);
        if constexpr (KIND == 13) b_dpp(v, w, a, b);
        if constexpr (KIND == 14) b_perm(v, w, a, b);
        if constexpr (KIND == 15) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
        }
        if constexpr (KIND == 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[t], 0, 0, 0);
        }
        if constexpr (KIND >= 20 && KIND <= 23) {  // dependent chains: 1, 2, 4 accumulators in rotation; 23: 3-chains per acc
            constexpr int NA = KIND == 20 ? 1 : (KIND == 21 ? 2 : 4);
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int t = KIND == 23 ? (k / 3) % 8 : k % NA;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
            }
        }
        if constexpr (KIND == 17) {  // 8 MFMA + 8 v_exp interleaved
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[t]));
                }
            }
        }
        if constexpr (KIND == 18) {  // 8 MFMA + 8 v_fma interleaved
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[t]) : "v"(a), "v"(b));
                }
            }
        }
        if constexpr (KIND == 19) {  // 1 MFMA : 4 v_fma
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(t + j) & 7]) : "v"(a), "v"(b));
                }
            }
        }
    }
    float res = 0.f;
    for (int t = 0; t < 8; ++t) res += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < 8; ++i) res += v[i];
    out[blockIdx.x * blockDim.x + tid] = res;
}

template <int KIND>
static void run(const char* name, float* out, int per_round) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rounds = 4000;
    printf("%-34s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD
        probe<KIND><<<256, wps * 256>>>(out, rounds);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<KIND><<<256, wps * 256>>>(out, rounds);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // cycles per instruction per SIMD at a nominal 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)rounds * per_round * wps);
        printf("  %dw/SIMD: %6.2f cyc (%.3f ms)", wps, cyc, ms);
    }
    printf("\n");
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 1024 * 4);
    run<0>("v_fma_f32", out, 64);
    run<1>("v_pk_fma_f32", out, 64);
    run<2>("v_exp_f32", out, 64);
    run<3>("v_rcp_f32", out, 64);
    run<4>("v_rsq_f32", out, 64);
    run<5>("v_max_f32", out, 64);
    run<6>("v_lshlrev_b32", out, 64);
    run<7>("v_cvt_pk_bf16_f32", out, 64);
    run<8>("v_cvt_pk_f16_f32", out, 64);
    run<9>("v_cvt_pkrtz_f16_f32", out, 64);
    run<10>("v_cvt_f32_f16", out, 64);
    run<11>("v_fma_mix_f32", out, 64);
    run<12>("v_fma_mixlo_f16", out, 64);
    run<13>("v_add_f32_dpp row_ror", out, 64);
    run<14>("ds_bpermute_b32 + wait", out, 64);
    run<15>("v_mfma_f32_16x16x32_f16", out, 64);
    run<16>("v_mfma_f32_16x16x32_bf16", out, 64);
    run<17>("mfma_f16 + v_exp (per pair)", out, 64);
    run<18>("mfma_f16 + v_fma (per pair)", out, 64);
    run<19>("mfma_f16 + 4 v_fma (per group)", out, 64);
    run<20>("mfma_f16 dependent chain (1 acc)", out, 64);
    run<21>("mfma_f16 2 accs alternating", out, 64);
    run<22>("mfma_f16 4 accs alternating", out, 64);
    run<23>("mfma_f16 3-chains, 8 accs", out, 64);
    return 0;
}
