// Opt-in per-launch timing of the MFMA kernels with HIP events recorded on the launch stream (used by bench.py for the
// `roofline` object; off by default).  On a stream that is being CAPTURED the two events become external event-record nodes of
// the graph (hipEventRecordExternal): every replay records them again, so dlwpcs_prof_get then returns the launch's duration
// INSIDE the replayed graph -- the form the timed region runs; such records carry the suffix "@graph" in their tag.  This is the
// only mutable global state in the library and it is inert unless dlwpcs_prof_enable(1) was called.
#include <mutex>
#include <string>
#include <vector>
#include <stdio.h>
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

// (function-local: the registrations run from other translation units' static initialisers)
static std::vector<std::string> &known_tags() { static std::vector<std::string> v; return v; }
int prof_register_tag(const char *tag) {
    auto &v = known_tags();
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == tag) return (int)i;
    v.push_back(tag);
    return (int)v.size() - 1;
}

static bool capturing(hipStream_t s) {
    if (s == nullptr) return false;                 // (the null stream cannot be captured; the query rejects it)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();                    // the failed query must not be taken for a failed launch (check_launch)
        return false;
    }
    return st == hipStreamCaptureStatusActive;
}

// An event record as a node of the graph being captured on s.  hipEventRecordExternal is what the API offers for it; the HIP
// runtime bundled with PyTorch 2.10 (ROCm 7.0) rejects the flag during capture, so the node is added by hand there: an event-record
// node behind the capture's current frontier, which becomes the new frontier.
static hipError_t record_in_capture(hipEvent_t e, hipStream_t s) {
    hipError_t r = hipEventRecordWithFlags(e, s, hipEventRecordExternal);
    if (r == hipSuccess) return r;
    (void)hipGetLastError();
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t g = nullptr;
    const hipGraphNode_t *deps = nullptr;
    size_t nd = 0;
    if ((r = hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &nd)) != hipSuccess) return r;
    hipGraphNode_t node = nullptr;
    if ((r = hipGraphAddEventRecordNode(&node, g, deps, nd, e)) != hipSuccess) return r;
    return hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
}

static void record(hipEvent_t e, hipStream_t s) {
    const bool cap = capturing(s);
    const hipError_t r = cap ? record_in_capture(e, s) : hipEventRecord(e, s);
    if (r != hipSuccess) {
        fprintf(stderr, "dlwpcs prof: %s on stream %p failed: %s\n", cap ? "external event record (capture)" : "event record", (void *)s,
                hipGetErrorString(r));
        (void)hipGetLastError();                    // a profiler failure is not a launch failure
    }
}

int prof_begin(const char *tag, double flops, double bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    ProfRecord r;
    r.tag = tag; r.flops = flops; r.bytes = bytes;
    if (capturing(s)) r.tag += "@graph";
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    record(r.e0, s);
    g_records.push_back(r);
    return (int)g_records.size() - 1;
}

void prof_end(int idx, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx >= 0 && idx < (int)g_records.size()) record(g_records[idx].e1, s);
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

extern "C" int dlwpcs_prof_known_tags(void) { return (int)known_tags().size(); }

extern "C" int dlwpcs_prof_known_tag(int i, char *tag, int tag_len) {
    const auto &v = known_tags();
    if (i < 0 || i >= (int)v.size() || !tag || tag_len < 1) return fail(DLWPCS_E_INVALID, "prof_known_tag: bad arguments");
    strncpy(tag, v[i].c_str(), tag_len - 1);
    tag[tag_len - 1] = 0;
    return DLWPCS_OK;
}
