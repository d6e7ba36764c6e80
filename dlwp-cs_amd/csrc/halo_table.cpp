// Host-side cube-sphere halo tables.
//
// Replaces the 36-slice / 8-reverse / 8-transpose / 14-concat graph of the reference padding layer
// (CubeSpherePadding2D.call, DLWP/custom.py:1082-1308) by one integer gather table, built by composing the layer's
// two passes on index triples.  Face convention (DLWP/custom.py:1063): 0-3 equatorial going east, 4 south pole,
// 5 north pole.
#include <vector>
#include <string.h>
#include "common.h"

namespace dlwpcs {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error() { return g_err; }

namespace {

struct Cell { int f, i, j; };

// Pass 1 (rows): source of halo row `a` (0..p-1 counted downwards inside the strip), column b of face f.
// DLWP/custom.py:1201-1251 (channels_last) == :1089-1139 (channels_first).
Cell rows_source(int f, int a, int b, int N, int p, bool top) {
    switch (f) {
        case 0: return top ? Cell{4, N - p + a, b} : Cell{5, a, b};                          // :1203-1209
        case 1: return top ? Cell{4, N - 1 - b, N - p + a} : Cell{5, b, N - 1 - a};          // :1211-1217
        case 2: return top ? Cell{4, p - 1 - a, N - 1 - b} : Cell{5, N - 1 - a, N - 1 - b};  // :1219-1225
        case 3: return top ? Cell{4, b, p - 1 - a} : Cell{5, N - 1 - b, a};                  // :1227-1233
        case 4: return top ? Cell{2, p - 1 - a, N - 1 - b} : Cell{0, a, b};                  // :1235-1241
        default: return top ? Cell{0, N - p + a, b} : Cell{2, N - 1 - a, N - 1 - b};         // :1243-1249
    }
}

void build_table(int N, int p, std::vector<int32_t> &T) {
    const int M = N + 2 * p;
    std::vector<int32_t> out1((size_t)6 * M * N);
    auto o1 = [&](int f, int i, int j) -> int32_t & { return out1[((size_t)f * M + i) * N + j]; };
    auto t = [&](int f, int i, int j) -> int32_t & { return T[((size_t)f * M + i) * M + j]; };
    for (int f = 0; f < 6; ++f)
        for (int b = 0; b < N; ++b) {
            for (int a = 0; a < p; ++a) {
                Cell c = rows_source(f, a, b, N, p, true);
                o1(f, a, b) = (c.f * N + c.i) * N + c.j;
                c = rows_source(f, a, b, N, p, false);
                o1(f, N + p + a, b) = (c.f * N + c.i) * N + c.j;
            }
            for (int i = 0; i < N; ++i) o1(f, p + i, b) = (f * N + i) * N + b;
        }
    // Pass 2 (columns).  Equatorial faces: periodic neighbours' row-padded edge columns (:1256-1287).
    for (int f = 0; f < 4; ++f) {
        const int left = (f + 3) % 4, right = (f + 1) % 4;
        for (int i = 0; i < M; ++i) {
            for (int j = 0; j < N; ++j) t(f, i, p + j) = o1(f, i, j);
            for (int a = 0; a < p; ++a) {
                t(f, i, a) = o1(left, i, N - p + a);
                t(f, i, N + p + a) = o1(right, i, a);
            }
        }
    }
    // Polar faces: strips of the FULLY padded equatorial faces 3 and 1 (:1289-1303), so corners inherit pass 1.
    for (int f = 4; f < 6; ++f)
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) t(f, i, p + j) = o1(f, i, j);
    for (int r = 0; r < M; ++r)
        for (int a = 0; a < p; ++a) {
            t(4, r, a) = t(3, 2 * p - 1 - a, r);             // :1291
            t(4, r, N + p + a) = t(1, p + a, M - 1 - r);     // :1293
            t(5, r, a) = t(3, N + a, M - 1 - r);             // :1299
            t(5, r, N + p + a) = t(1, N + p - 1 - a, r);     // :1301
        }
}

}  // namespace
}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" int dlwpcs_version(void) { return DLWPCS_VERSION; }
extern "C" const char *dlwpcs_last_error(void) { return dlwpcs::last_error(); }

extern "C" int dlwpcs_halo_table(int N, int p, int32_t *out) {
    if (!out) return fail(DLWPCS_E_INVALID, "halo_table: null output");
    if (N < 1 || p < 0 || p > N || N > 4096) return fail(DLWPCS_E_INVALID, "halo_table: need 0 <= p <= N, got N=%d p=%d", N, p);
    const int M = N + 2 * p;
    std::vector<int32_t> T((size_t)6 * M * M);
    build_table(N, p, T);
    memcpy(out, T.data(), T.size() * sizeof(int32_t));
    return DLWPCS_OK;
}

extern "C" int dlwpcs_halo_inverse_table(int N, int p, int32_t *inv) {
    if (!inv) return fail(DLWPCS_E_INVALID, "halo_inverse_table: null output");
    if (N < 1 || p < 0 || p > N || N > 4096) return fail(DLWPCS_E_INVALID, "halo_inverse_table: bad N=%d p=%d", N, p);
    const int M = N + 2 * p;
    std::vector<int32_t> T((size_t)6 * M * M);
    build_table(N, p, T);
    const size_t ncell = (size_t)6 * N * N;
    std::vector<int> cnt(ncell, 0);
    for (size_t k = 0; k < ncell * 4; ++k) inv[k] = -1;
    for (int f = 0; f < 6; ++f)
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) {
                const bool interior = (i >= p && i < p + N && j >= p && j < p + N);
                if (interior) continue;     // identity copy is implicit
                const int32_t src = T[((size_t)f * M + i) * M + j];
                if (cnt[src] >= 4) return fail(DLWPCS_E_INVALID, "halo_inverse_table: fan-out > 5 at N=%d p=%d", N, p);
                inv[(size_t)src * 4 + cnt[src]++] = (int32_t)(((size_t)f * M + i) * M + j);
            }
    return DLWPCS_OK;
}
