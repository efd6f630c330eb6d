// Host-side batch packing (no device code): the native counterpart of the reference's
// InstanceLoader.create_batch inner loops (instance_loader.py:56-73) and of the CSR-by-vertex build,
// so that packing a batch costs ~1 ms instead of the reference's O(M) Python loop + O(M*N) dense matrix.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

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

// Whole-batch variants: one call per batch instead of one per instance (the Python loop over 128 instances costs
// more than the packing itself).  Arrays of per-instance pointers / sizes, all in host memory.
template <typename T>
static long count_nonzero(const T* Ma, long n2) {
    long m = 0;
    for (long k = 0; k < n2; ++k) m += Ma[k] != T(0);
    return m;
}

extern "C" int tspgnn_host_count_edges(const void* const* Ma, const int* ma_kind, const int* n, int B, int64_t* n_edges) {
    if (B < 0 || (B > 0 && (!Ma || !ma_kind || !n || !n_edges))) return -1;
    for (int b = 0; b < B; ++b) {
        const long n2 = (long)n[b] * n[b];
        switch (ma_kind[b]) {
            case 0: n_edges[b] = count_nonzero(static_cast<const int8_t*>(Ma[b]), n2); break;
            case 1: n_edges[b] = count_nonzero(static_cast<const int32_t*>(Ma[b]), n2); break;
            case 2: n_edges[b] = count_nonzero(static_cast<const int64_t*>(Ma[b]), n2); break;
            case 3: n_edges[b] = count_nonzero(static_cast<const float*>(Ma[b]), n2); break;
            case 4: n_edges[b] = count_nonzero(static_cast<const double*>(Ma[b]), n2); break;
            default: return -1;
        }
    }
    return 0;
}

extern "C" double tspgnn_host_route_cost(const double* Mw, int n, const int64_t* route, int len);

// uv[M,2], W[M], C[M] of the block-diagonal batch (instance_loader.py:56-73).  C: target_cost if use_target != 0,
// else (1 - dev) * cost for even instances and (1 + dev) * cost for odd ones, cost = tspgnn_host_route_cost.
extern "C" long long tspgnn_host_pack_batch(const void* const* Ma, const int* ma_kind, const double* const* Mw,
                                            const int* n, const int64_t* const* route, const int* route_len, int B,
                                            double dev, int use_target, double target_cost, int32_t* uv, double* W,
                                            double* C) {
    if (B < 0 || (B > 0 && (!Ma || !ma_kind || !Mw || !n || !uv || !W || !C))) return -1;
    if (!use_target && B > 0 && (!route || !route_len)) return -1;
    long long m_acc = 0;
    int v_off = 0;
    for (int b = 0; b < B; ++b) {
        const long long m = tspgnn_host_pack_instance(Ma[b], ma_kind[b], Mw[b], n[b], v_off, uv + 2 * m_acc, W + m_acc);
        if (m < 0) return -1;
        double c = target_cost;
        if (!use_target) {
            if (route_len[b] > 0 && !route[b]) return -1;
            for (int k = 0; k < route_len[b]; ++k)  // the reference's Mw[x, y] raises IndexError here
                if (route[b][k] < 0 || route[b][k] >= n[b]) return -2;
            const double cost = tspgnn_host_route_cost(Mw[b], n[b], route[b], route_len[b]);
            c = (b % 2 == 0) ? (1.0 - dev) * cost : (1.0 + dev) * cost;
        }
        for (long long k = 0; k < m; ++k) C[m_acc + k] = c;
        m_acc += m;
        v_off += n[b];
    }
    return m_acc;
}

// sum of Mw[min,max] over the pairs zip(route, route[1:] + route[1:]) divided by n -- including the
// reference's closing-edge quirk (instance_loader.py:70): the last pair is (route[-1], route[1]).
// A vertex id outside [0, n) yields NaN instead of an out-of-bounds read (the reference raises IndexError).
extern "C" double tspgnn_host_route_cost(const double* Mw, int n, const int64_t* route, int len) {
    double s = 0.0;
    if (!Mw || n <= 0 || (len > 0 && !route)) return __builtin_nan("");
    for (int k = 0; k < len; ++k) {
        const int64_t x = route[k];
        const int64_t y = (k + 1 < len) ? route[k + 1] : (len > 1 ? route[1] : route[0]);
        if (x < 0 || x >= n || y < 0 || y >= n) return __builtin_nan("");
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

// ---------------------------------------------------------------------------------- one staged batch
// Everything a device batch is made of (Session.prepare), packed by ONE call into ONE caller-owned buffer -- meant to be
// pinned host memory, so that a batch reaches the GPU as one asynchronous copy (parallel.BatchStager): the worker thread
// spends its time here, outside the interpreter lock, instead of in eight numpy conversions and eight pageable uploads
// (round 4: the fresh-batch path ran 9-17 % behind the resident one, and its one packer thread was nearly as slow as the
// forward pass itself).  off[7] = byte offsets of
//   uv int32[M][2] | eid int32[2M] | rowptr int32[N+1] | wc float[M][2] (edge weight, target cost) | labels float[B]
//   (i mod 2, instance_loader.py:50) | seg int32[B+1] (prefix sums of the edge counts) | n_edges int32[B]
// Returns M, or -1 malformed input, -2 a route names a vertex outside its graph, -3 M / N differ from what the caller
// sized the buffer for.
extern "C" long long tspgnn_host_stage_batch(const void* const* Ma, const int* ma_kind, const double* const* Mw,
                                             const int* n, const int64_t* const* route, const int* route_len, int B,
                                             double dev, int use_target, double target_cost, long long M_expected,
                                             int N_expected, unsigned char* stage, const long long* off) {
    if (B < 0 || !stage || !off || (B > 0 && (!Ma || !ma_kind || !Mw || !n))) return -1;
    static thread_local std::vector<int64_t> counts;
    static thread_local std::vector<double> W, C;
    counts.assign((size_t)(B > 0 ? B : 1), 0);
    if (tspgnn_host_count_edges(Ma, ma_kind, n, B, counts.data()) != 0) return -1;
    long long M = 0, N = 0;
    for (int b = 0; b < B; ++b) {
        M += counts[b];
        N += n[b];
    }
    if (M != M_expected || N != N_expected) return -3;
    int32_t* uv = reinterpret_cast<int32_t*>(stage + off[0]);
    int32_t* eid = reinterpret_cast<int32_t*>(stage + off[1]);
    int32_t* rowptr = reinterpret_cast<int32_t*>(stage + off[2]);
    float* wc = reinterpret_cast<float*>(stage + off[3]);
    float* labels = reinterpret_cast<float*>(stage + off[4]);
    int32_t* seg = reinterpret_cast<int32_t*>(stage + off[5]);
    int32_t* ne = reinterpret_cast<int32_t*>(stage + off[6]);
    W.resize((size_t)(M > 0 ? M : 1));
    C.resize((size_t)(M > 0 ? M : 1));
    const long long got = tspgnn_host_pack_batch(Ma, ma_kind, Mw, n, route, route_len, B, dev, use_target, target_cost, uv,
                                                 W.data(), C.data());
    if (got < 0) return got;
    if (got != M) return -1;
    for (long long e = 0; e < M; ++e) {   // the float32 the device reads (Session.prepare: np.stack([W, C], 1) as float32)
        wc[2 * e] = (float)W[e];
        wc[2 * e + 1] = (float)C[e];
    }
    seg[0] = 0;
    for (int b = 0; b < B; ++b) {
        labels[b] = (float)(b % 2);
        ne[b] = (int32_t)counts[b];
        seg[b + 1] = seg[b] + (int32_t)counts[b];
    }
    const int rc = tspgnn_host_csr_by_vertex(uv, M, (int)N, rowptr, eid);
    return rc == 0 ? M : -1;
}

// ---------------------------------------------------------------------------------- .graph files
// The reference's TSPLIB-like text format (written by dataset.py:145-187, parsed by instance_loader.py:95-127 with
// Python string splitting): DIMENSION, EDGE_DATA_SECTION (pairs "i j" until a line with -1), EDGE_WEIGHT_SECTION
// (full n x n matrix), TOUR_SECTION (one line of vertex ids).  Two calls: the first (Ma == NULL) returns n and the
// tour length, the second fills Ma[n*n] (0/1, int64 like the reference's np.zeros(dtype=int)), Mw[n*n], route.
namespace {

bool slurp(const char* path, std::string* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, got);
    fclose(f);
    return true;
}

// pointer to the first character after the line that contains `key` (searching from `from`), or NULL
const char* after_line_with(const char* from, const char* key) {
    const char* p = strstr(from, key);
    if (!p) return nullptr;
    const char* nl = strchr(p, '\n');
    return nl ? nl + 1 : p + strlen(p);
}

}  // namespace

extern "C" int tspgnn_host_read_graph(const char* path, int* n_out, int* route_len_out, int64_t* Ma, double* Mw,
                                      int64_t* route) {
    if (!path || !n_out || !route_len_out) return -1;
    std::string text;
    if (!slurp(path, &text)) return -2;
    const char* t = text.c_str();
    const char* dim = strstr(t, "DIMENSION");
    if (!dim) return -3;
    const char* colon = dim + strlen("DIMENSION");
    while (*colon == ':' || *colon == ' ' || *colon == '\t') ++colon;
    const long n = strtol(colon, nullptr, 10);
    if (n <= 0 || n > (1 << 20)) return -3;
    const char* edges = after_line_with(dim, "EDGE_DATA_SECTION");
    const char* weights = edges ? after_line_with(edges, "EDGE_WEIGHT_SECTION") : nullptr;
    const char* tour = weights ? after_line_with(weights, "TOUR_SECTION") : nullptr;
    if (!edges || !weights || !tour) return -4;
    // tour: integers on the line after TOUR_SECTION
    int len = 0;
    {
        const char* p = tour;
        for (;;) {
            while (*p == ' ' || *p == '\t') ++p;
            if (*p == '\n' || *p == '\r' || *p == 0) break;
            char* end;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v < 0 || v >= n) return -7;  // a tour names vertices of this graph
            if (route) route[len] = v;
            ++len;
            p = end;
        }
    }
    *n_out = (int)n;
    *route_len_out = len;
    if (!Ma && !Mw) return 0;  // size query
    if (!Ma || !Mw) return -1;
    memset(Ma, 0, sizeof(int64_t) * n * n);
    {   // pairs until the line that contains -1 (the reference tests `"-1" in line`)
        const char* p = edges;
        while (p < weights) {
            const char* nl = strchr(p, '\n');
            const char* e = nl ? nl : p + strlen(p);
            if (memmem(p, e - p, "-1", 2) != nullptr) break;
            char* q;
            const long i = strtol(p, &q, 10);
            if (q == p) break;
            const long j = strtol(q, nullptr, 10);
            if (i < 0 || j < 0 || i >= n || j >= n) return -5;
            Ma[i * n + j] = 1;
            if (!nl) break;
            p = nl + 1;
        }
    }
    {
        const char* p = weights;
        for (long k = 0; k < n * n; ++k) {
            char* q;
            Mw[k] = strtod(p, &q);
            if (q == p) return -6;
            p = q;
        }
    }
    return 0;
}
