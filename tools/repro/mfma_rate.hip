// Development probe: issue rate of v_mfma_f32_16x16x32_bf16 for ONE wave per SIMD with 32 independent accumulators — accumulators in
// VGPRs (builtin), pinned in AGPRs (inline asm "+a"), pinned in VGPRs (inline asm "+v"); cycles per MFMA from s_memtime and
// the clock the chip held from wall time.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8 *__restrict__ in, float *__restrict__ out, unsigned long long *__restrict__ cyc, int iters) {
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x * 12 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = in[threadIdx.x * 12 + 4 + i];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (MODE == 0) acc[m * 4 + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nb], b[m], acc[m * 4 + nb], 0, 0, 0);
                if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m * 4 + nb]) : "v"(a[nb]), "v"(b[m]));
                if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m * 4 + nb]) : "v"(a[nb]), "v"(b[m]));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, const bf16x8 *in, float *out, unsigned long long *cyc, int iters, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += v; c /= grid;
    const double n = (double)iters * 32;
    printf("%-22s grid %4d: %.1f cycles per MFMA (s_memtime), %.3f ms per launch, clock held %.2f GHz, %.0f TFLOP/s\n", name, grid, c / n, ms,
           c / (ms * 1e6), (double)grid * 4 * n * 16384 / (ms * 1e-3) / 1e12);
}

int main() {
    const int grid = 256, iters = 4000;
    bf16x8 *in; float *out; unsigned long long *cyc;
    hipMalloc(&in, 256 * 12 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    std::vector<unsigned short> h(256 * 12 * 8);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }   // random bf16 around +-1
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0>("builtin (VGPR acc)", in, out, cyc, iters, grid);
    run<1>("asm, acc in AGPRs", in, out, cyc, iters, grid);
    run<2>("asm, acc in VGPRs", in, out, cyc, iters, grid);
    run<0>("builtin, 1024 WGs", in, out, cyc, iters / 4, 1024);
    return 0;
}
