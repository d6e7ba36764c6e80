// Probe: do external event-record nodes inside a captured hipGraph carry timestamps on this runtime?
// hipcc --offload-arch=gfx950 -o /tmp/pge tools/probe_graph_events.hip && /tmp/pge
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); printf("%-58s -> %s\n", #x, hipGetErrorName(e_)); } while (0)
__global__ void spin(float *p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
int main(int argc, char **argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;       // 0 = global capture, 1 = thread-local, 2 = relaxed
    float *d; CK(hipMalloc(&d, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    spin<<<1, 64, 0, s>>>(d, 1000); CK(hipStreamSynchronize(s));
    CK(hipStreamBeginCapture(s, (hipStreamCaptureMode)mode));
    CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));         // created DURING capture
    CK(hipEventRecordWithFlags(a0, s, hipEventRecordExternal));
    spin<<<1, 64, 0, s>>>(d, 200000);
    CK(hipGetLastError());
    CK(hipEventRecordWithFlags(a1, s, hipEventRecordExternal));
    CK(hipEventRecordWithFlags(b0, s, hipEventRecordExternal));
    spin<<<1, 64, 0, s>>>(d, 400000);
    CK(hipEventRecordWithFlags(b1, s, hipEventRecordExternal));
    hipGraph_t g; CK(hipStreamEndCapture(s, &g));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) {
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float ta = -1, tb = -1;
        CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1));
        printf("replay %d: kernel A %.1f us, kernel B %.1f us\n", r, ta * 1e3, tb * 1e3);
    }
    return 0;
}
