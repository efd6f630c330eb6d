// Shared host/device helpers for libtspgnn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tspgnn.h"

namespace tspgnn {

// Records a thread-local message and returns `code` (see tspgnn_last_error()).
int fail(int code, const char* fmt, ...);
// hipGetLastError() after a launch; 0 or the positive hipError_t (message recorded).
int launched(const char* what);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[i] (+)= scale * sum_{c<n_chunks} partial[c*stride+i]: fixed-order second stage of the split reductions
// (dense_bwd.hip).
void reduce_partials(const float* partial, int n_chunks, long long stride, float* out, int n, float scale,
                     int accumulate, hipStream_t st);
// The same with a second output segment of the same chunking (b: partial_b / stride_b / out_b / n_b; out_b == NULL: none)
// in the one launch.
void reduce_partials2(const float* partial, int n_chunks, long long stride, float* out, int n, const float* partial_b,
                      long long stride_b, float* out_b, int n_b, float scale, int accumulate, hipStream_t st);

#define TSPGNN_REQUIRE(cond, ...) \
    do {                          \
        if (!(cond)) return ::tspgnn::fail(TSPGNN_EINVAL, __VA_ARGS__); \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// Compute units of the current device (256 on MI355X); sizes the persistent grids.
inline int n_cus(void) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}

// 1/(1+e^-x) on the hardware transcendental units: v_exp_f32 (via exp2(x*log2 e)) and v_rcp_f32,
// each ~1 ulp -> ~2e-7 relative error, two orders inside the 1e-5 parity budget, at a fraction of
// the ~60 VALU instructions of the IEEE expf + division sequence.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// Workgroup b of a launch runs on XCD b mod 8 (observed placement, used for speed only -- any value is correct).  Maps the
// task-relative workgroup index b in [0, n) to a position such that the workgroups of one XCD hold CONSECUTIVE positions:
// with contiguous tile ranges per position, an XCD then works on one contiguous eighth of the rows, so the vertex rows its
// edges gather (Zx, a few MB in total at the ragged / n=200 sizes) stay within its own 4 MB L2.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int per = n >> 3, rem = n & 7, x = b & 7;
    return x * per + (x < rem ? x : rem) + (b >> 3);
}

// Sum over the four 16-lane groups of a wavefront (lanes l, l^16, l^32, l^48); every lane
// ends with the total.  The order (l + l^16) + (l^32 + l^48) is fixed -> deterministic.
__device__ __forceinline__ float sum_over_lane_groups16(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

}  // namespace tspgnn
