/* The C ABI of libtspgnn.so used from plain C (no Python, no torch): what a non-Python host would bind.
 * Build (see tests/test_gpu_c_abi.py):
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_abi/abi_smoke.c \
 *       -L tsp-gnn_amd/tspgnn -ltspgnn -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o abi_smoke
 * E <- V gather and V <- E row-sum on a 3-vertex triangle, then an invalid call to read tspgnn_last_error(). */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>

#include "tspgnn.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s\n", (int)e_, #x); return 2; } } while (0)

int main(void) {
    enum { N = 3, M = 3, D = 32 };
    const int32_t uv[M][2] = {{0, 1}, {0, 2}, {1, 2}};
    const int32_t rowptr[N + 1] = {0, 2, 4, 6}, eid[2 * M] = {0, 1, 0, 2, 1, 2};
    float X[N][D], Y[M][D], Z[N][D];
    for (int v = 0; v < N; ++v)
        for (int j = 0; j < D; ++j) X[v][j] = (float)(v + 1) + 0.01f * (float)j;
    if (tspgnn_version() != TSPGNN_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
    void *d_uv, *d_rp, *d_eid, *d_X, *d_Y, *d_Z;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    CHECK(hipMalloc(&d_uv, sizeof uv)); CHECK(hipMalloc(&d_rp, sizeof rowptr)); CHECK(hipMalloc(&d_eid, sizeof eid));
    CHECK(hipMalloc(&d_X, sizeof X)); CHECK(hipMalloc(&d_Y, sizeof Y)); CHECK(hipMalloc(&d_Z, sizeof Z));
    CHECK(hipMemcpy(d_uv, uv, sizeof uv, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rp, rowptr, sizeof rowptr, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_eid, eid, sizeof eid, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_X, X, sizeof X, hipMemcpyHostToDevice));
    int rc = tspgnn_gather2_sum_f32((const int32_t*)d_uv, (const float*)d_X, (float*)d_Y, M, N, D, st);
    if (rc == 0) rc = tspgnn_csr_rowsum_f32((const int32_t*)d_rp, (const int32_t*)d_eid, (const float*)d_Y, (float*)d_Z, N, M, D, st);
    if (rc != 0) { printf("tspgnn call failed (%d): %s\n", rc, tspgnn_last_error()); return 1; }
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(Y, d_Y, sizeof Y, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(Z, d_Z, sizeof Z, hipMemcpyDeviceToHost));
    for (int e = 0; e < M; ++e)
        for (int j = 0; j < D; ++j)
            if (Y[e][j] != X[uv[e][0]][j] + X[uv[e][1]][j]) { printf("gather mismatch at %d,%d\n", e, j); return 1; }
    for (int v = 0; v < N; ++v)
        for (int j = 0; j < D; ++j) {
            const float want = Y[eid[2 * v]][j] + Y[eid[2 * v + 1]][j];
            if (Z[v][j] != want) { printf("row-sum mismatch at %d,%d\n", v, j); return 1; }
        }
    rc = tspgnn_gather2_sum_f32((const int32_t*)d_uv, (const float*)d_X, (float*)d_Y, M, N, 7, st);   /* d % 4 != 0 */
    if (rc != TSPGNN_EINVAL || strlen(tspgnn_last_error()) == 0) { printf("expected TSPGNN_EINVAL with a message, got %d\n", rc); return 1; }
    printf("C ABI OK: version %d, gather + row-sum exact, error path: %s\n", tspgnn_version(), tspgnn_last_error());
    return 0;
}
