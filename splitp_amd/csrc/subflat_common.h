// Device helpers shared by the subflattening score kernels (subflat.hip: one wave per split; subflat_pair.hip: two splits
// per wave).
#pragma once
#include "common.h"

#define SUBT_MMAX 61           // side of the staged moment matrix the LDS forms take (20 taxa)
#define SPK_LDS_TOTAL 163840   // LDS of a CU

// idx(S) + [3n] for one split half
__device__ __forceinline__ int sub_index(const int8_t* taxa, int cnt, int n, int i) {
    return i < 3 * cnt ? 3 * taxa[i / 3] + (i % 3) : 3 * n;
}

// LDS traffic between the lanes of ONE wave: the hardware runs a wave's LDS instructions in order, what is needed is that
// the compiler does not move accesses across this point
__device__ __forceinline__ void wave_sync_lds2() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a double moved between lanes on the DPP path (no LDS crossbar); rows outside the mask ROWS receive 0
template <int CTRL, int ROWS = 0xF>
__device__ __forceinline__ double subt_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWS, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWS, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double subt_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// max(|a|, |b|) as one instruction (fmax(fabs(a), fabs(b)) adds a canonicalising v_max_f64 x, x per operand)
__device__ __forceinline__ double subt_max_abs(double a, double b) {
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
typedef double subt_d2 __attribute__((ext_vector_type(2)));

struct sp_alignment;
int launch_subscore_pair(sp_alignment* al, const int8_t* dtaxa, const int* da, const int* order, const PairClasses& pc, int rt,
                         double* scores_out, int* status_out);   // subflat_pair.hip

// Steps I, I + 1, ... r - 1 of the minor recurrence (k_subscore_tri), written out by recursion: constant LDS offsets, the
// pair (P_(i-1), P_i) changes registers instead of being moved, one scalar test a step (`#pragma unroll` leaves this loop
// rolled).  de[i] = {d_i, e2_(i-1)}, v = de[I]; the sign of every new minor is shifted into `mask`.
template <int I>
__device__ __forceinline__ void subt_minor_steps(int r, const subt_d2* de, double sigma, double pp, double pc, subt_d2 v,
                                                 unsigned& mask) {
    if constexpr (I < 31) {
        if (I < r) {
            // the next step's pair is requested before this step's arithmetic (nothing is scheduled across the barrier), or
            // the compiler moves the read down to its use and every step waits out an LDS round trip (de has 32 pairs)
            const subt_d2 vn = de[I + 1];
            __builtin_amdgcn_sched_barrier(0);
            double pn = fma(v.x - sigma, pc, -(v.y * pp));
            mask = __builtin_amdgcn_alignbit(mask, (unsigned)__double2hiint(pn), 31);   // (mask << 1) | sign
            if ((I & 7) == 0) {
                const int ex = __builtin_amdgcn_frexp_exp(subt_max_abs(pn, pc));
                pn = ldexp(pn, -ex);
                pc = ldexp(pc, -ex);
            }
            subt_minor_steps<I + 1>(r, de, sigma, pc, pn, vn, mask);
        }
    }
}
