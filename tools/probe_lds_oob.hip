// Probe: what does a ds_read_b128 return for an address beyond the workgroup's LDS allocation on gfx950?  (GCN ISA: out-of-range
// LDS reads return 0.)  hipcc --offload-arch=gfx950 tools/probe_lds_oob.hip -o /tmp/probe_lds_oob && /tmp/probe_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out, unsigned off) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) smem[i] = 0xdead0000u + i;
    __syncthreads();
    unsigned addr = (threadIdx.x & 1) ? off + threadIdx.x * 16 : threadIdx.x * 16;    // odd lanes out of range
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    unsigned *d, h[64 * 4];
    hipMalloc(&d, sizeof(h));
    const unsigned offs[] = {1024u, 65536u, 163840u, 0x00f00000u, 0x7fff0000u, 0xfffff000u};
    for (unsigned off : offs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, off);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int nz = 0;
        for (int t = 1; t < 64; t += 2) for (int c = 0; c < 4; ++c) nz += h[t * 4 + c] != 0;
        printf("offset 0x%08x: odd lanes nonzero dwords %d (lane1 = %08x %08x), even lane0 = %08x\n", off, nz, h[4], h[5], h[0]);
    }
    return 0;
}
