// LDS micro-benchmarks on one 512-thread workgroup (8 waves, the sparse kernel's shape): cycles per wave-instruction for
// reads / writes / atomics with random and with linear addresses, and dependent-chain latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 64
__global__ __launch_bounds__(512) void k(long long* out, const unsigned* idx_g, int mode) {
    extern __shared__ unsigned lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 32768; i += 512) lds[i] = (i * 2654435761u) & 32767u;
    unsigned idx[N];
    for (int i = 0; i < N; ++i) idx[i] = idx_g[(i * 512 + t) & 65535] & 32767u;   // random word index
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    if (mode == 0) {          // independent random b32 reads
#pragma unroll
        for (int i = 0; i < N; ++i) acc += lds[idx[i]];
    } else if (mode == 1) {   // linear b32 reads
#pragma unroll
        for (int i = 0; i < N; ++i) acc += lds[(i * 512 + t) & 32767];
    } else if (mode == 2) {   // dependent chain
        unsigned p = idx[0];
#pragma unroll
        for (int i = 0; i < N; ++i) p = lds[p];
        acc = p;
    } else if (mode == 3) {   // random atomics, no return
#pragma unroll
        for (int i = 0; i < N; ++i) atomicAdd(&lds[idx[i]], 1u);
    } else if (mode == 4) {   // random atomics with return
#pragma unroll
        for (int i = 0; i < N; ++i) acc += atomicAdd(&lds[idx[i]], 1u);
    } else if (mode == 5) {   // random b64 reads (8-byte aligned)
        const unsigned long long* l64 = reinterpret_cast<const unsigned long long*>(lds);
        unsigned long long a64 = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) a64 += l64[idx[i] >> 1];
        acc = (unsigned)a64 ^ (unsigned)(a64 >> 32);
    } else if (mode == 6) {   // random b32 writes
#pragma unroll
        for (int i = 0; i < N; ++i) lds[idx[i]] = i;
    } else if (mode == 7) {   // team gather: 4 lanes read 4 consecutive doubles of a random row (pitch 5 doubles)
        const double* ld = reinterpret_cast<const double*>(lds);
        double a = 0;
        const int j = t & 3;
#pragma unroll
        for (int i = 0; i < N; ++i) a += ld[(__shfl(idx[i], t & ~3) % 3000) * 5 + j];
        acc = (unsigned)a;
    } else if (mode == 8) {   // u16 reads random
        const unsigned short* l16 = reinterpret_cast<const unsigned short*>(lds);
#pragma unroll
        for (int i = 0; i < N; ++i) acc += l16[idx[i] * 2];
    } else if (mode >= 10 && mode <= 13) {   // one lane gathers a whole 4-double row: 2 x b128 (pitch 4 / 6) or 4 x b64 (pitch 5)
        const int pitch = mode == 10 ? 4 : (mode == 11 ? 6 : (mode == 12 ? 5 : 8));
        const double* ld = reinterpret_cast<const double*>(lds);
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double* row = ld + (idx[i] % (16000 / pitch)) * pitch;
            if (mode == 12) {
                a0 += row[0]; a1 += row[1]; a2 += row[2]; a3 += row[3];
            } else {
                const double2 lo = *reinterpret_cast<const double2*>(row), hi = *reinterpret_cast<const double2*>(row + 2);
                a0 += lo.x; a1 += lo.y; a2 += hi.x; a3 += hi.y;
            }
        }
        acc = (unsigned)(a0 + a1 + a2 + a3);
    } else if (mode == 14) {  // random u64 atomic add, no return
        unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
#pragma unroll
        for (int i = 0; i < N; ++i) atomicAdd(&l64[idx[i] >> 1], (unsigned long long)i + t);
    } else if (mode == 15) {  // same-address u64 atomic add, no return (all lanes of a wave one word)
        unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
#pragma unroll
        for (int i = 0; i < N; ++i) atomicAdd(&l64[i], (unsigned long long)i + t);
    } else if (mode == 16) {  // scatter product: per step 1 random b64 read + fma + 1 random u64 atomic (x N)
        unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
        const double* ld = reinterpret_cast<const double*>(lds);
        const double magic = 6755399441055744.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double x = __builtin_fma((double)(idx[i] & 255u), ld[idx[(i + 7) & (N - 1)] >> 1], magic);
            atomicAdd(&l64[(idx[i] >> 1) & 8191], (unsigned long long)__double_as_longlong(x) - (unsigned long long)__double_as_longlong(magic));
        }
    } else if (mode == 17) {  // u64 atomics, 8 distinct addresses per wave (hot rows)
        unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
#pragma unroll
        for (int i = 0; i < N; ++i) atomicAdd(&l64[(idx[i] >> 1) & 7], (unsigned long long)i + t);
    } else if (mode == 18 || mode == 19) {  // u64 atomics, 28 % of the lanes on 4 hot words, the rest random over 1024 (18) / 4096 (19) words
        unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
        const unsigned span = mode == 18 ? 1023u : 4095u;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const unsigned h = idx[i];
            const unsigned a = ((h >> 20) & 255u) < 72u ? (h & 3u) * 37u : (h >> 1) & span;
            atomicAdd(&l64[a], (unsigned long long)i + t);
        }
    } else if (mode == 20) {  // u64 atomics random over 1024 words
        unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
#pragma unroll
        for (int i = 0; i < N; ++i) atomicAdd(&l64[(idx[i] >> 1) & 1023u], (unsigned long long)i + t);
    } else if (mode == 9) {   // same-address atomics with return (all lanes one word per wave)
#pragma unroll
        for (int i = 0; i < N; ++i) acc += atomicAdd(&lds[i], 1u);
    }
    __syncthreads();
    long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) out[blockIdx.x * 2] = t1 - t0;
    if (acc == 0x12345678u) out[1] = acc;
}
int main() {
    long long* d; unsigned* idx;
    hipMalloc(&d, 4096 * 16); hipMalloc(&idx, 65536 * 4);
    std::vector<unsigned> h(65536);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    const char* names[] = {"random b32 read", "linear b32 read", "dependent chain", "random atomic add", "random atomic add rtn",
                           "random b64 read", "random b32 write", "team gather b64 pitch5", "random u16 read", "same-address atomic rtn", "row gather 2xb128 pitch4", "row gather 2xb128 pitch6", "row gather 4xb64 pitch5", "row gather 2xb128 pitch8", "random u64 atomic add", "same-address u64 atomic", "read b64 + fma + atomic u64", "u64 atomic on 8 addresses", "u64 atomic 28% hot4 + 1024", "u64 atomic 28% hot4 + 4096", "u64 atomic random 1024"};
    for (int mode = 0; mode < 21; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(512), 131072, 0, d, idx, mode);
        long long c[2];
        hipMemcpy(c, d, 16, hipMemcpyDeviceToHost);
        printf("%-28s %8lld cycles for %d instr/wave x 8 waves -> %.1f cycles per wave-instruction (CU), %.1f per instr in a wave\n",
               names[mode], c[0], N, (double)c[0] / (N * 8), (double)c[0] / N);
    }
    return 0;
}
