// Probe of the v_mfma_f64_4x4x4_4b_f64 operand layout: A one-hot at lane la, B one-hot at lane lb -> where is D != 0 ?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {
    const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[(size_t)blockIdx.x * 64 + lane] = d;
}
int main() {
    double* d;
    hipMalloc(&d, 4096 * 64 * 8);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
    std::vector<double> h(4096 * 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    return 0;
}
