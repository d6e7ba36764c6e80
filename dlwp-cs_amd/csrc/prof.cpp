// Opt-in per-launch timing of the MFMA kernels with HIP events recorded on the launch stream (used by bench.py for the
// `roofline` object; off by default, never active inside graph capture).  This is the only mutable global state in the
// library and it is inert unless dlwpcs_prof_enable(1) was called.
#include <mutex>
#include <string>
#include <vector>
#include <string.h>
#include "common.h"

namespace dlwpcs {

struct ProfRecord {
    std::string tag;
    double flops, bytes;
    hipEvent_t e0, e1;
};

static std::mutex g_mu;
static bool g_enabled = false;
static std::vector<ProfRecord> g_records;

bool prof_enabled() { return g_enabled; }

int prof_begin(const char *tag, double flops, double bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    ProfRecord r;
    r.tag = tag; r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    (void)hipEventRecord(r.e0, s);
    g_records.push_back(r);
    return (int)g_records.size() - 1;
}

void prof_end(int idx, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx >= 0 && idx < (int)g_records.size()) (void)hipEventRecord(g_records[idx].e1, s);
}

}  // namespace dlwpcs

using namespace dlwpcs;

extern "C" int dlwpcs_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_enabled = on != 0;
    return DLWPCS_OK;
}

extern "C" int dlwpcs_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &r : g_records) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_records.clear();
    return DLWPCS_OK;
}

extern "C" int dlwpcs_prof_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_records.size();
}

extern "C" int dlwpcs_prof_get(int i, char *tag, int tag_len, double *ms, double *flops, double *bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (i < 0 || i >= (int)g_records.size() || !tag || tag_len < 1 || !ms || !flops || !bytes)
        return fail(DLWPCS_E_INVALID, "prof_get: bad arguments");
    ProfRecord &r = g_records[i];
    if (hipEventSynchronize(r.e1) != hipSuccess) return fail(DLWPCS_E_LAUNCH, "prof_get: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return fail(DLWPCS_E_LAUNCH, "prof_get: elapsed time failed");
    strncpy(tag, r.tag.c_str(), tag_len - 1);
    tag[tag_len - 1] = 0;
    *ms = t; *flops = r.flops; *bytes = r.bytes;
    return DLWPCS_OK;
}
