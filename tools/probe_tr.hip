// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element does lane l / register element j receive when the 16 lanes
// of a group each supply the address of 4 contiguous 16-bit values?  hipcc --offload-arch=gfx950 tools/probe_tr.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int *addr, unsigned short *out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    const int ROW = 32;      // elements per row: block rows r = 0..3, lane i supplies row i/4, columns (i%4)*4..+3
    int h_addr[64]; unsigned short h_out[256];
    for (int l = 0; l < 64; ++l) { int i = l & 15; h_addr[l] = (i >> 2) * ROW + (i & 3) * 4 + (l >> 4) * 1024; }
    int *d_addr; unsigned short *d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            int v = h_out[l * 4 + j] - (l >> 4) * 1024;
            printf(" (r%d,c%2d)", v / ROW, v % ROW);
            if (v / ROW != j || v % ROW != (l & 15)) ok = 0;
        }
        printf("\n");
    }
    printf("expected semantics (lane n gets column n, element j = row j): %s\n", ok ? "YES" : "NO");
    return 0;
}
