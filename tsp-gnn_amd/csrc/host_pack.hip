// Host-side batch packing (no device code): the native counterpart of the reference's
// InstanceLoader.create_batch inner loops (instance_loader.py:56-73) and of the CSR-by-vertex build,
// so that packing a batch costs ~1 ms instead of the reference's O(M) Python loop + O(M*N) dense matrix.
#include <stdint.h>

#include "tspgnn.h"

namespace {

template <typename T>
long pack_edges(const T* Ma, const double* Mw, int n, int v_off, int32_t* uv, double* W) {
    long m = 0;
    for (int i = 0; i < n; ++i) {
        const T* row = Ma + (long)i * n;
        for (int j = 0; j < n; ++j) {
            if (row[j] != T(0)) {  // np.nonzero order: row-major over the whole matrix
                uv[2 * m] = v_off + i;
                uv[2 * m + 1] = v_off + j;
                W[m] = Mw[(long)i * n + j];
                ++m;
            }
        }
    }
    return m;
}

}  // namespace

extern "C" long long tspgnn_host_pack_instance(const void* Ma, int ma_kind, const double* Mw, int n, int v_off,
                                               int32_t* uv, double* W) {
    if (!Ma || !Mw || !uv || !W || n < 0) return -1;
    switch (ma_kind) {
        case 0: return pack_edges(static_cast<const int8_t*>(Ma), Mw, n, v_off, uv, W);
        case 1: return pack_edges(static_cast<const int32_t*>(Ma), Mw, n, v_off, uv, W);
        case 2: return pack_edges(static_cast<const int64_t*>(Ma), Mw, n, v_off, uv, W);
        case 3: return pack_edges(static_cast<const float*>(Ma), Mw, n, v_off, uv, W);
        case 4: return pack_edges(static_cast<const double*>(Ma), Mw, n, v_off, uv, W);
        default: return -1;
    }
}

// sum of Mw[min,max] over the pairs zip(route, route[1:] + route[1:]) divided by n -- including the
// reference's closing-edge quirk (instance_loader.py:70): the last pair is (route[-1], route[1]).
extern "C" double tspgnn_host_route_cost(const double* Mw, int n, const int64_t* route, int len) {
    double s = 0.0;
    for (int k = 0; k < len; ++k) {
        const int64_t x = route[k];
        const int64_t y = (k + 1 < len) ? route[k + 1] : (len > 1 ? route[1] : route[0]);
        const int64_t lo = x < y ? x : y, hi = x < y ? y : x;
        s += Mw[lo * n + hi];
    }
    return s / n;
}

// CSR of EV^T by counting sort: rowptr[N+1], eid[2M]; edge ids ascending inside a vertex.
extern "C" int tspgnn_host_csr_by_vertex(const int32_t* uv, long long M, int N, int32_t* rowptr, int32_t* eid) {
    if ((M > 0 && (!uv || !eid)) || !rowptr || N < 0 || M < 0) return -1;
    for (int v = 0; v <= N; ++v) rowptr[v] = 0;
    for (long long k = 0; k < 2 * M; ++k) {
        const int32_t v = uv[k];
        if (v < 0 || v >= N) return -2;
        ++rowptr[v + 1];
    }
    for (int v = 0; v < N; ++v) rowptr[v + 1] += rowptr[v];
    // fill using a moving cursor per vertex (rowptr copy kept in eid's tail is avoided: second pass with offsets)
    for (long long e = 0; e < M; ++e) {
        for (int s = 0; s < 2; ++s) {
            const int32_t v = uv[2 * e + s];
            eid[rowptr[v]++] = (int32_t)e;
        }
    }
    for (int v = N; v > 0; --v) rowptr[v] = rowptr[v - 1];  // undo the cursor advance
    rowptr[0] = 0;
    return 0;
}
