// Host-side cube-sphere halo tables.
//
// Replaces the 36-slice / 8-reverse / 8-transpose / 14-concat graph of the reference padding layer
// (CubeSpherePadding2D.call, DLWP/custom.py:1082-1308) by one integer gather table, built by composing the layer's
// two passes on index triples.  Face convention (DLWP/custom.py:1063): 0-3 equatorial going east, 4 south pole,
// 5 north pole.
#include <vector>
#include <algorithm>
#include <map>
#include <mutex>
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

// ------------------------------------------------------------------------------------------------------------------
// Gather form of the data gradient (p = 1, 3x3): dlwpcs_dgrad_gather_plan.
//
// Forward: xpad[f][i][j] = x[T[f,i,j]], z[f][y][x] = sum_{ty,tx} xpad[f][y+ty][x+tx] . W_v(f)[ty][tx]  (DLWP/custom.py:921-1002
// behind :1198-1308; v = 0 faces 0-3, 1 face 4, 2 face 5).  Its adjoint gives every source cell c the terms
//     (q, v, ty, tx)  <=>  padded cell (face(q), y(q) + ty, x(q) + tx) gathers from c,  v = v(face(q)):   dz[q] . W_v[ty][tx]^T
// The data-gradient kernel forms, for c = (f0, y, x), the plain correlation of the HALO-PADDED dz (the forward's own gather, table T)
//     sum_{a,b} dzpad[f0][y+a][x+b] . W_v(f0)[2-a][2-b]^T
// Window positions (a, b) inside the face give true terms.  A position in the halo gives a true term where the neighbour uses the
// same kernel in the same orientation (equatorial-equatorial edges); elsewhere the term is WRONG -- bit a * 3 + b of the cell's
// wrong-tap mask: the kernel cancels it -- and the true terms of the cell that the correlation does not form (<= 6, on border
// cells only) all take their dz row from a halo cell of the cell's own 3 x 3 window (checked here): slot k < 3 holds the window
// position a * 3 + b of the first term with the k-th weight id of the cell's triple, slot 3 + k that of a second term with the same
// id (cube corners); -1 = none.  Per face two weight-id triples (cells of row 0; cells of row N - 1; cells in between use either,
// the plan checks that they agree), weight id = v * 9 + (2 - ty) * 3 + (2 - tx): variant and tap index of the operand pack.
// Per border cell 8 ints: six slots, the wrong-tap mask, 0.
// ------------------------------------------------------------------------------------------------------------------
namespace dlwpcs {
size_t dgrad_gather_plan_ints(int N) {
    const size_t M = N + 2;
    return (size_t)6 * N * N * 4 + DGG_HEADER + 6 * M * M + (size_t)6 * (4 * N - 4) * DGG_CELL + 6 * 2 * 3;
}
}

extern "C" size_t dlwpcs_dgrad_gather_plan_ints(int N) {
    if (N < 8 || N > 4096) return 0;
    return dlwpcs::dgrad_gather_plan_ints(N);
}

extern "C" int dlwpcs_dgrad_gather_plan(int N, int32_t *out) {
    if (!out) return fail(DLWPCS_E_INVALID, "dgrad_gather_plan: null output");
    if (N < 8 || N > 4096) return fail(DLWPCS_E_UNSUPPORTED, "dgrad_gather_plan: face size %d (needs 8 <= N <= 4096)", N);
    const int p = 1, M = N + 2, nb = 4 * N - 4;
    int rc = dlwpcs_halo_inverse_table(N, p, out);
    if (rc) return rc;
    const int32_t *inv = out;
    int32_t *hdr = out + (size_t)6 * N * N * 4;
    int32_t *tab = hdr + DGG_HEADER;
    int32_t *cells = tab + (size_t)6 * M * M;
    int32_t *ewid = cells + (size_t)6 * nb * DGG_CELL;
    hdr[0] = DGG_MAGIC; hdr[1] = N; hdr[2] = (int32_t)(tab - out); hdr[3] = (int32_t)(cells - out); hdr[4] = (int32_t)(ewid - out);
    hdr[5] = nb; hdr[6] = DGG_CELL; hdr[7] = 0;
    std::vector<int32_t> T((size_t)6 * M * M);
    build_table(N, p, T);
    memcpy(tab, T.data(), T.size() * sizeof(int32_t));
    auto t = [&](int f, int i, int j) { return T[((size_t)f * M + i) * M + j]; };
    auto var = [](int f) { return f < 4 ? 0 : (f == 4 ? 1 : 2); };
    // (q, v, ty, tx) is a true term of c  <=>  the padded cell q + (ty, tx) of q's face gathers from c and the face's variant is v
    auto is_term = [&](int c, int q, int v, int ty, int tx) {
        const int fq = q / (N * N), yq = (q / N) % N, xq = q % N;
        return var(fq) == v && t(fq, yq + ty, xq + tx) == c;
    };
    for (size_t k = 0; k < (size_t)6 * nb * DGG_CELL; ++k) cells[k] = (k % DGG_CELL) < 6 ? -1 : 0;
    for (int k = 0; k < 36; ++k) ewid[k] = -1;
    struct Term { int pos, wid; };
    for (int f0 = 0; f0 < 6; ++f0) {
        std::vector<std::vector<Term>> terms(nb);
        std::vector<int> wtop, wbot;
        auto add_wid = [](std::vector<int> &w, int id) { for (int u : w) if (u == id) return; w.push_back(id); };
        auto ordinal = [&](int y, int x) { return y == 0 ? x : (y == N - 1 ? N + x : 2 * N + 2 * (y - 1) + (x ? 1 : 0)); };
        auto in_face = [&](int i, int j) { return i >= 1 && i <= N && j >= 1 && j <= N; };
        for (int y = 0; y < N; ++y)
            for (int x = 0; x < N; ++x) {
                if (y != 0 && y != N - 1 && x != 0 && x != N - 1) continue;
                const int c = (f0 * N + y) * N + x, o = ordinal(y, x);
                // the halo positions of the window: a true term of c (free), or a wrong one (to be cancelled)
                int wrong = 0;
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b)
                        if (!in_face(y + a, x + b) && !is_term(c, t(f0, y + a, x + b), var(f0), 2 - a, 2 - b)) wrong |= 1 << (a * 3 + b);
                cells[((size_t)f0 * nb + o) * DGG_CELL + 6] = wrong;
                std::vector<Term> &tl = terms[o];
                for (int u = 0; u < 4; ++u) {
                    const int32_t pc = inv[(size_t)c * 4 + u];
                    if (pc < 0) continue;
                    const int f = pc / (M * M), i = (pc / M) % M, j = pc % M;
                    for (int ty = 0; ty < 3; ++ty)
                        for (int tx = 0; tx < 3; ++tx) {
                            const int yq = i - ty, xq = j - tx;
                            if (yq < 0 || yq >= N || xq < 0 || xq >= N) continue;
                            const int q = (f * N + yq) * N + xq, v = var(f);
                            // formed by the correlation?  it reads dzpad[f0][y + a][x + b] with (a, b) = (2 - ty, 2 - tx)
                            const int a0 = 2 - ty, b0 = 2 - tx;
                            if (v == var(f0) && t(f0, y + a0, x + b0) == q) continue;
                            // ... else it must be served out of a halo cell of the window
                            int pos = -1;
                            for (int a = 0; a < 3 && pos < 0; ++a)
                                for (int b = 0; b < 3 && pos < 0; ++b)
                                    if (!in_face(y + a, x + b) && t(f0, y + a, x + b) == q) pos = a * 3 + b;
                            if (pos < 0) return fail(DLWPCS_E_UNSUPPORTED, "dgrad_gather_plan: a term's source is outside the cell's window (N=%d face %d)", N, f0);
                            tl.push_back(Term{pos, v * 9 + a0 * 3 + b0});
                        }
                }
                for (const Term &tm : tl) {
                    if (y == 0) add_wid(wtop, tm.wid);
                    else if (y == N - 1) add_wid(wbot, tm.wid);
                }
            }
        if (wtop.size() > 3 || wbot.size() > 3)
            return fail(DLWPCS_E_UNSUPPORTED, "dgrad_gather_plan: more than three weight ids on an edge row (N=%d face %d)", N, f0);
        std::sort(wtop.begin(), wtop.end());
        std::sort(wbot.begin(), wbot.end());
        for (size_t k = 0; k < wtop.size(); ++k) ewid[(f0 * 2 + 0) * 3 + k] = wtop[k];
        for (size_t k = 0; k < wbot.size(); ++k) ewid[(f0 * 2 + 1) * 3 + k] = wbot[k];
        for (int y = 0; y < N; ++y)
            for (int x = 0; x < N; ++x) {
                if (y != 0 && y != N - 1 && x != 0 && x != N - 1) continue;
                const int o = ordinal(y, x);
                int32_t *slots = cells + ((size_t)f0 * nb + o) * DGG_CELL;
                for (const Term &tm : terms[o]) {
                    int k = -1;
                    if (y == N - 1) { for (size_t u = 0; u < wbot.size(); ++u) if (wbot[u] == tm.wid) k = (int)u; }
                    else { for (size_t u = 0; u < wtop.size(); ++u) if (wtop[u] == tm.wid) k = (int)u; }
                    // cells between the edge rows sit in M tiles that may also hold cells of row 0 or of row N - 1: both triples must agree
                    if (k >= 0 && y != 0 && y != N - 1 && !((size_t)k < wbot.size() && wbot[k] == tm.wid)) k = -1;
                    if (k < 0) return fail(DLWPCS_E_UNSUPPORTED, "dgrad_gather_plan: weight id outside the row's triple (N=%d face %d)", N, f0);
                    if (slots[k] < 0) slots[k] = tm.pos;
                    else if (slots[3 + k] < 0) slots[3 + k] = tm.pos;
                    else return fail(DLWPCS_E_UNSUPPORTED, "dgrad_gather_plan: more than two terms per weight id (N=%d face %d)", N, f0);
                }
            }
    }
    return DLWPCS_OK;
}

// the two weight-id triples per face of dlwpcs_dgrad_gather_plan(N) on the host (cached): the data-gradient launch passes them to
// the kernel by value
namespace dlwpcs {
int dgrad_gather_wids(int N, int32_t out[36]) {
    static std::mutex mu;
    static std::map<int, std::vector<int32_t>> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(N);
    if (it == cache.end()) {
        const size_t n = dlwpcs_dgrad_gather_plan_ints(N);
        if (!n) return fail(DLWPCS_E_UNSUPPORTED, "dgrad_gather_wids: face size %d", N);
        std::vector<int32_t> plan(n);
        const int rc = dlwpcs_dgrad_gather_plan(N, plan.data());
        if (rc) return rc;
        it = cache.emplace(N, std::vector<int32_t>(plan.end() - 36, plan.end())).first;
    }
    memcpy(out, it->second.data(), 36 * sizeof(int32_t));
    return DLWPCS_OK;
}
}
