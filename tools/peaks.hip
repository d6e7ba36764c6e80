// Measured peaks of the box the numbers in DESIGN.md / bench.py are priced against (SURVEY.md section 8 d: "verify with a
// micro-benchmark on the box and record the measured peaks").  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 tools/peaks.hip -o /tmp/peaks && /tmp/peaks
// 1. dense bf16 matrix peak: v_mfma_f32_32x32x16_bf16, 4 independent accumulator chains per wave, operands in registers
// 2. fp32 matrix peak:       v_mfma_f32_32x32x2_f32 (the exact-fp32 instruction of the parity mode)
// 3. HBM: streaming read, streaming write and copy of buffers far larger than the 256 MiB Infinity Cache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) mfma_bf16_kernel(float *out, int iters) {
    const unsigned t = threadIdx.x;
    uint4 ua = make_uint4(0x3f803f80u + t, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), ub = make_uint4(0x3f003f00u, 0x3f003f00u + t, 0x3f003f00u, 0x3f003f00u);
    const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
    out[blockIdx.x * blockDim.x + t] = s;
}

__global__ void __launch_bounds__(256) mfma_f32_kernel(float *out, int iters) {
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) read_kernel(const uint4 *__restrict__ p, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;       // keeps the loads alive
}
__global__ void __launch_bounds__(256) write_kernel(uint4 *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();                                            // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const int cus = prop.multiProcessorCount;
    float *out; CK(hipMalloc(&out, (size_t)cus * 64 * 256 * 4));
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = cus * waves_per_simd, iters = 20000;        // 256 threads = 4 waves = one per SIMD
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_bf16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
        double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("bf16 v_mfma_f32_32x32x16_bf16, %d wave(s)/SIMD: %8.1f TFLOP/s\n", waves_per_simd, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(mfma_f32_kernel, dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
        fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
        printf("fp32 v_mfma_f32_32x32x2_f32,   %d wave(s)/SIMD: %8.1f TFLOP/s\n", waves_per_simd, fl / ms / 1e9);
    }
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    uint4 *a, *b; unsigned *flag;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&flag, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    for (int blocks : {cus * 8, cus * 16, cus * 32}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, a, n, flag); }, 5);
        printf("HBM read  4 GiB, %5d workgroups: %7.0f GB/s\n", blocks, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, 0, b, n); }, 5);
        printf("HBM write 4 GiB, %5d workgroups: %7.0f GB/s\n", blocks, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        printf("HBM copy  4 GiB, %5d workgroups: %7.0f GB/s (read + write)\n", blocks, 2.0 * bytes / ms / 1e6);
    }
    // kernel launch floor: an empty-ish kernel back to back on one stream
    float ms = time_ms([&] { for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(write_kernel, dim3(1), dim3(64), 0, 0, b, (size_t)1); }, 5);
    printf("dependent one-wave launches on one stream: %.2f us each\n", ms * 1e3 / 200);
    return 0;
}
