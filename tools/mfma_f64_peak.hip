// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (the local microarchitecture guide
// has no F64 row).  W waves per SIMD, NACC independent accumulators per wave, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak && gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double* out, int iters, double a0, double b0, long long* cyc) {
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd, int cus) {
    double* out;
    long long* cyc;
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks = cus * (256 * waves_per_simd / threads);
    (void)hipMalloc(&out, (size_t)blocks * threads * 8);
    (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0, 1.0, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0, cyc);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
    printf("waves/SIMD=%d acc=%d blocks=%d x %d thr: %.3f ms  %.2f TFLOP/s  ticks/MFMA/wave=%.1f  ns/MFMA/SIMD=%.2f\n",
           waves_per_simd, NACC, blocks, threads, ms, flops / (ms * 1e-3) / 1e12, (double)c / ((double)iters * NACC),
           ms * 1e6 / ((double)iters * NACC * waves_per_simd));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int cus = p.multiProcessorCount;
    run<1>(1, cus);
    run<4>(1, cus);
    run<4>(2, cus);
    run<4>(4, cus);
    run<2>(4, cus);
    run<4>(8, cus);
    run<1>(8, cus);
    return 0;
}
