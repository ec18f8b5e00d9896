// Subflattening scores, TWO splits per wave (round 4; replaces splitp/constructions.py:108-163 + phylogenetics.py:280-312 for
// every split of a batch, like subflat.hip's k_subscore_tri, whose numerical method this keeps: exact Gram matrix of the
// <= 31-row block on the fp64 matrix cores, Householder tridiagonalisation, the four largest eigenvalues by multisection on
// the Sturm count).
//
// What k_subscore_tri spent (profiles/r03_pmc_binding_config4_auto.json: 4 460 vector instructions a split, vector issue
// 60 %, LDS 53 %, waves parked 40 %): of the ~2 300 instructions of its tridiagonalisation only ~600 were the multiply-adds
// of the two row loops - the rest was paid per STEP, not per entry: two 32-lane sums, the reflector's square root and
// reciprocal, loop control, three LDS round trips; and its Sturm phase ran 16 shifts an eigenvalue where fewer shifts and
// more passes do the same bracketing with fewer evaluations.  Here:
//   * lanes 0-31 work on one split, lanes 32-63 on another of the same size class: every per-step instruction now serves
//     two splits;
//   * lane i of a half holds ROW i of its Gram matrix in registers (31 doubles, compile-time column indices: the
//     elimination runs from the last column down, so the active block is always the leading L x L and every loop bound is
//     a template parameter).  LDS carries only the two vectors of a step - x (the column) and w - as 16-byte broadcast
//     reads: a third of the LDS traffic, no row-partner shuffle, no bank conflicts to lay out around;
//   * p = beta A v is formed as beta (A x - alpha A e): the row loop runs on x as soon as the column is written, while the
//     norm, the square root and the reciprocal of the reflector are still in flight - the longest dependent chain of a step
//     is gone;
//   * sums over the 32 lanes of a half are butterflies on the DPP path plus one v_permlane16_swap (gfx950), so every lane
//     ends with the same bits and nothing goes through scalar registers (which a wave's two halves could not share);
//   * the Sturm phase brackets 4 eigenvalues x 8 shifts per half (9-section, 17 passes; 14 where the score is >= 0.05)
//     instead of 16 shifts and 13 (11) passes: 0.6 of the evaluations per split.
// Measured and dropped on the way (tools/experiments/README.md): x kept in registers, the Sturm table in registers, shifts
// placed around a secant guess, the Sturm phases of two pairs run together.
// A split's result does not depend on its partner: nothing crosses the halves but wave-uniform branches on the class's row
// count and the pass loop's exit, and a half whose brackets are final keeps them while the other goes on.  (Shards pair the
// splits differently and must return the same bits: tests/test_gpu_direct.py, tests/test_gpu_parity.py.)
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"
#include "subflat_common.h"

#ifndef SUBP_MAXWAVES
#define SUBP_MAXWAVES 16
#endif
#ifndef SUBP_PASSES
#define SUBP_PASSES 17   // 9^17 > 2^53
#define SUBP_EARLY 14    // passes after which a bracket is 9^-14 = 4.4e-14 of the (unit) Gershgorin interval wide
#endif

struct SubpAdd { static __device__ __forceinline__ double f(double a, double b) { return a + b; } };
struct SubpMin { static __device__ __forceinline__ double f(double a, double b) { return __builtin_fmin(a, b); } };
struct SubpMax { static __device__ __forceinline__ double f(double a, double b) { return __builtin_fmax(a, b); } };

// OP over the 32 lanes of each half of the wave, returned to every lane of the half.  Butterfly: lane <-> lane ^ 1, ^ 2
// (quad permutes), the mirror images inside 8 and 16 lanes (every lane of a group holds the group's value by then), rows
// 0 <-> 1 and 2 <-> 3 by v_permlane16_swap.  OP is commutative, so the two partners of every step compute the same bits.
template <class OP>
__device__ __forceinline__ double subp_all32(double x) {
    x = OP::f(x, subt_dpp<0xB1>(x));
    x = OP::f(x, subt_dpp<0x4E>(x));
    x = OP::f(x, subt_dpp<0x141>(x));
    x = OP::f(x, subt_dpp<0x140>(x));
    const int lo = __double2loint(x), hi = __double2hiint(x);
    // (vdst = src = x: the odd rows of the first copy change places with the even rows of the second, i.e. afterwards the
    // first copy holds row 0 | row 0 | row 2 | row 2 and the second row 1 | row 1 | row 3 | row 3)
    const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return OP::f(__hiloint2double(sh[0], sl[0]), __hiloint2double(sh[1], sl[1]));
}

// Per-wave LDS: G (Gram staging, rmax rows of pitch rmax | 1 doubles) - once both halves hold their rows in registers
// the same bytes carry x[64], w[64] (one entry a lane) and the Sturm tables de[2][32] = {d_i, e2_(i-1)} - then urow
// (32 x u16) and vcol (64 x u8) of the split whose Gram matrix is being formed.
__host__ __device__ __forceinline__ size_t subp_g_doubles(int rmax) {
    const size_t g = (size_t)rmax * (rmax | 1);
    return g > 256 ? g : 256;
}
__host__ __device__ __forceinline__ size_t subp_wave_bytes(int rmax) {
    return (subp_g_doubles(rmax) * 8 + 32 * 2 + 64 + 15) & ~(size_t)15;
}

// Householder steps L, L - 1, ... 2 on the leading (L + 1) x (L + 1) block: step L annihilates column L above the
// sub-diagonal entry (L - 1, L) with H = I - beta v v^T, v = x - alpha e_(L-1), and replaces the leading L x L block A by
// H A H = A - v w^T - w v^T, p = beta A v, w = p - (beta v^T p / 2) v.  a = this lane's row, row = its index in the half.
// svl / swl: this lane's entry of x and w in LDS; svh / swh: the vectors of this lane's half; deh: the half's Sturm table.
template <int L>
__device__ __forceinline__ void subp_steps(double (&a)[31], const int r, const int row, double* const svl,
                                           const double* const svh, double* const swl, const double* const swh,
                                           subt_d2* const deh) {
    if constexpr (L >= 2) {
        if (L < r) {
            const double x = row < L ? a[L] : 0.0;
            if (row == L) deh[L].x = a[L];   // d_L
            *svl = x;
            const double sig = subp_all32<SubpAdd>(x * x);
            wave_sync_lds2();
            // A x over the block's columns (four chains; for odd L the pair read past the end holds x_L = 0)
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            double x0 = 0.0;
#pragma unroll
            for (int j = 0; j < L; j += 2) {
                const subt_d2 xv = *reinterpret_cast<const subt_d2*>(svh + j);
                acc[(j >> 1) & 1] = fma(a[j], xv.x, acc[(j >> 1) & 1]);
                if (j + 1 < L) acc[2 + ((j >> 1) & 1)] = fma(a[j + 1], xv.y, acc[2 + ((j >> 1) & 1)]);
                if (j == L - 1) x0 = xv.x;
                if (j + 1 == L - 1) x0 = xv.y;
            }
            // reflector (square root and reciprocal from the hardware seeds + Newton steps, as in k_subscore_tri).  A column
            // that is already tridiagonal (nothing but its head) is left alone: beta = 0 makes p, w and the update vanish.
            const bool has = sig - x0 * x0 > 0;
            double ry = __builtin_amdgcn_rsq(sig);
            ry = ry * fma(-0.5 * sig * ry, ry, 1.5);
            double sq = sig * ry;
            sq = fma(0.5 * ry, fma(-sq, sq, sig), sq);
            const double alpha = has ? __builtin_copysign(sq, -x0) : 0.0;   // (sig = 0: the seeds return inf / NaN)
            const double den = sig - alpha * x0;
            double beta = __builtin_amdgcn_rcp(den);
            beta = fma(fma(-den, beta, 1.0), beta, beta);
            beta = fma(fma(-den, beta, 1.0), beta, beta);
            beta = has ? beta : 0.0;
            const double ax = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            double p = beta * fma(-alpha, a[L - 1], ax);      // beta (A x - alpha A e_(L-1))
            p = (has && row < L) ? p : 0.0;
            const double vi = row == L - 1 ? x - alpha : x;   // (x = 0 from row L on)
            const double kk = subp_all32<SubpAdd>(vi * p);
            const double wi = fma(-0.5 * beta * kk, vi, p);
            *swl = wi;
            if (row == 0) deh[L].y = has ? sig : x0 * x0;     // e2_(L-1) = alpha^2
            wave_sync_lds2();
#pragma unroll
            for (int j = 0; j < L; j += 2) {
                const subt_d2 xv = *reinterpret_cast<const subt_d2*>(svh + j);
                const subt_d2 wv = *reinterpret_cast<const subt_d2*>(swh + j);
                const double v0 = j == L - 1 ? xv.x - alpha : xv.x;
                a[j] = fma(-vi, wv.x, fma(-wi, v0, a[j]));
                if (j + 1 < L) {
                    const double v1 = j + 1 == L - 1 ? xv.y - alpha : xv.y;
                    a[j + 1] = fma(-vi, wv.y, fma(-wi, v1, a[j + 1]));
                }
            }
            wave_sync_lds2();
        }
        subp_steps<L - 1>(a, r, row, svl, svh, swl, swh, deh);
    }
}


#ifdef SUBP_STAMPS
// diagnostic build (tools/gpu_subpair_stamps.sh): s_memtime of wave 0 of workgroup 0 at the phase boundaries of its last pair
// (the classes come in ascending size: one of the longest)
__device__ long long g_subp_stamps[16];
extern "C" int sp_debug_subp_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_subp_stamps), sizeof(long long) * 16) == hipSuccess ? 0 : 2;
}
#define PSTAMP(i) do { if (blockIdx.x == 0 && w == 0 && lane == 0) g_subp_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(i)
#endif

template <bool EXACT, bool M32>
__global__ __launch_bounds__(SUBP_MAXWAVES * 64) void k_subscore_pair(const void* __restrict__ Mv, int n, int rmax,
                                                                      const int8_t* __restrict__ split_taxa,
                                                                      const int* __restrict__ split_a,
                                                                      const int* __restrict__ order,
                                                                      const PairClasses* __restrict__ pcd,
                                                                      double* __restrict__ scores, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    const int m = 3 * n + 1;
    typedef typename std::conditional<M32, int, double>::type MsT;
    MsT* Ms = reinterpret_cast<MsT*>(smem_p);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane >> 5, row = lane & 31;
    const int P = rmax | 1, nthreads = blockDim.x;
    unsigned char* wbase = smem_p + (((size_t)m * m * sizeof(MsT) + 15) & ~(size_t)15) + (size_t)w * subp_wave_bytes(rmax);
    double* const G = reinterpret_cast<double*>(wbase);
    double* const sv = G;                                              // x of both halves (after the row loads)
    double* const sw = G + 64;
    subt_d2* const deh = reinterpret_cast<subt_d2*>(G + 128) + 32 * half;
    unsigned short* const urow = reinterpret_cast<unsigned short*>(G + subp_g_doubles(rmax));
    unsigned char* const vcol = reinterpret_cast<unsigned char*>(urow + 32);
    for (int e = threadIdx.x; e < m * m; e += nthreads)
        Ms[e] = EXACT ? (MsT) reinterpret_cast<const long long*>(Mv)[e] : (MsT) reinterpret_cast<const double*>(Mv)[e];
    __syncthreads();
    const int wpb = nthreads >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    // (the class table sits in device memory and is read with wave-uniform indices - scalar loads; as a kernel argument
    // passed by value its runtime-indexed arrays were copied into ~110 scalar registers, most of them spilled)
    const int nclass = pcd->nclass;
    const int64_t npairs = pcd->poff[nclass];
    for (int64_t pr = (int64_t)blockIdx.x * wpb + w; pr < npairs; pr += nwaves) {
        int q = 0;
        while (q + 1 < nclass && pr >= pcd->poff[q + 1]) ++q;
        const int r = pcd->rows[q];
        const int64_t cstart = pcd->start[q];
        const int64_t posA = cstart + 2 * (pr - pcd->poff[q]);
        const int64_t posB = posA + 1 < cstart + pcd->count[q] ? posA + 1 : posA;   // (an odd class: the last split twice)
        const int64_t sidA = order ? (int64_t)order[posA] : posA, sidB = order ? (int64_t)order[posB] : posB;
        PSTAMP(0);
        double a[31];
#pragma unroll
        for (int j = 0; j < 31; ++j) a[j] = 0.0;
        double dg = 0.0;
        // ---- Gram matrices, one split after the other through the staging area; rows into registers ----------------------
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int64_t sid = h ? sidB : sidA;
            const int8_t* taxa = split_taxa + sid * n;
            const int sa = split_a[sid], sb = n - sa;
            const bool swap = sa > sb;   // rows = smaller side
            const int8_t* rt = swap ? taxa + sa : taxa;
            const int8_t* ct = swap ? taxa : taxa + sa;
            const int nr = swap ? sb : sa, nc = swap ? sa : sb;
            const int c = 3 * nc + 1;
            wave_sync_lds2();   // the previous reads of the staging area and the tables are done
            if (lane < r) urow[lane] = (unsigned short)(sub_index(rt, nr, n, lane) * m);
            if (lane < c) vcol[lane] = (unsigned char)sub_index(ct, nc, n, lane);
            wave_sync_lds2();
            PSTAMP(6 + 3 * h);
            // G = B B^T on the matrix cores, B[i][k] = M[urow_i + vcol_k] gathered from the staged moment matrix.
            // v_mfma_f64_16x16x4: lane (fr = lane & 15, fk = lane >> 4) supplies B[16 I + fr][4 s + fk] as A and
            // B[16 J + fr][4 s + fk] as B operand of tile (I, J); accumulator q holds G[16 I + fk + 4 q][16 J + fr].
            // (count tables: every term and sum is an integer below 2^53, so the result does not depend on the order)
            {
                typedef double d4 __attribute__((ext_vector_type(4)));
                // (the lane number made opaque once more: the 16 store addresses below are invariants of the pair loop - the
                // compiler computed them ahead of it, spilled them, and every store then waited for a scratch load)
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const int fr = lane_o & 15, fk = lane_o >> 4;
                const bool two = r > 16;
                // (every LDS read below is unconditional and its result selected afterwards: written as `cond ? table[i] : 0`
                // each gather became a branch around one read with its own s_waitcnt - eight LDS round trips in sequence
                // per trip, 7 of the phase's 10 k ticks a split)
                const int ur0 = (int)urow[fr], ur1 = (int)urow[16 + fr];   // (32 entries: in bounds; stale beyond r)
                const int u0 = fr < r ? ur0 : -1, u1 = (two && 16 + fr < r) ? ur1 : -1;
                d4 g00 = {0, 0, 0, 0}, g01 = {0, 0, 0, 0}, g11 = {0, 0, 0, 0};
                // 16 columns a trip: the four column indices, then the eight gathers, then the products - two LDS round trips
                // per trip (k_subscore_tri's loop made them per 4 columns, and this phase is nothing but their latency)
#pragma unroll 1
                for (int k0 = 0; k0 < c; k0 += 16) {
                    int v[4];
                    double x0[4], x1[4];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int k = k0 + 4 * qq + fk;   // (< 64: vcol has 64 entries)
                        const int vc = (int)vcol[k];
                        v[qq] = k < c ? vc : -1;
                    }
                    MsT m0[4], m1[4];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const bool ok0 = v[qq] >= 0 && u0 >= 0, ok1 = v[qq] >= 0 && u1 >= 0;
                        m0[qq] = Ms[ok0 ? u0 + v[qq] : 0];
                        m1[qq] = Ms[ok1 ? u1 + v[qq] : 0];
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        x0[qq] = (v[qq] >= 0 && u0 >= 0) ? (double)m0[qq] : 0.0;
                        x1[qq] = (v[qq] >= 0 && u1 >= 0) ? (double)m1[qq] : 0.0;
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        if (k0 + 4 * qq >= c) break;
                        g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[qq], x0[qq], g00, 0, 0, 0);
                        if (two) {
                            g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[qq], x1[qq], g01, 0, 0, 0);
                            g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[qq], x1[qq], g11, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int i = fk + 4 * qq;
                    if (i < r && fr < r) G[i * P + fr] = g00[qq];
                    if (two) {
                        if (16 + fr < r) {
                            G[i * P + 16 + fr] = g01[qq];
                            G[(16 + fr) * P + i] = g01[qq];
                            if (16 + i < r) G[(16 + i) * P + 16 + fr] = g11[qq];
                        }
                    }
                }
            }
            wave_sync_lds2();
            PSTAMP(7 + 3 * h);
            if (half == h && row < r) {
                const double* const grow = G + row * P;
#pragma unroll
                for (int j = 0; j < 31; ++j)
                    if (j < r) a[j] = grow[j];
                dg = grow[row];
            }
            PSTAMP(8 + 3 * h);
        }
        wave_sync_lds2();   // the staging area becomes x, w and the Sturm tables
        PSTAMP(1);
        const double tr = subp_all32<SubpAdd>(dg);
        const int64_t sid_mine = half ? sidB : sidA;
        if (r <= 4) {
            if (row == 0) {
                scores[sid_mine] = (tr > 0) ? 0.0 : __builtin_nan("");
                status[sid_mine] = 0;
            }
            continue;
        }
        // ---- Householder tridiagonalisation in registers -------------------------------------------------------------------
        // (the row index made opaque here: its ~60 comparisons with the steps' constants are loop invariants, which the
        // compiler otherwise computes ahead of the pair loop and keeps - 120 scalar registers, most of them spilled)
        int rowv = row;
        asm volatile("" : "+v"(rowv));
        subp_steps<30>(a, r, rowv, sv + lane, sv + 32 * half, sw + lane, sw + 32 * half, deh);
        PSTAMP(2);
        if (row == 1) deh[1].x = a[1];
        if (row == 0) {
            subt_d2 first;
            first.x = a[0];
            first.y = 0.0;
            deh[0] = first;
            deh[1].y = a[1] * a[1];
        }
        wave_sync_lds2();
        // ---- four largest eigenvalues by multisection on the Sturm count (k_subscore_tri's recurrence and scaling: the
        // matrix in units in which its Gershgorin interval has length in [0.5, 1), squared off-diagonals raised to 2^-120)
        double gl = 1e300, gu = -1e300, dmine = 0.0, e2mine = 0.0;
        if (row < r) {
            const subt_d2 me = deh[row];
            dmine = me.x;
            e2mine = row > 0 ? me.y : 0.0;
            const double e2r = row < r - 1 ? deh[row + 1].y : 0.0;
            const double el = sqrt(e2mine), er = sqrt(e2r);
            gl = dmine - el - er;
            gu = dmine + el + er;
        }
        gl = subp_all32<SubpMin>(gl);
        gu = subp_all32<SubpMax>(gu);
        const int E = __builtin_amdgcn_frexp_exp(fmax(gu - gl, fmax(fabs(gl), fabs(gu)) * 0x1p-40));
        wave_sync_lds2();   // every lane has read its neighbour's entry
        if (row < r) {
            subt_d2 sc;
            sc.x = ldexp(dmine, -E);
            sc.y = row > 0 ? fmax(ldexp(e2mine, -2 * E), 0x1p-120) : 0.0;
            deh[row] = sc;
        }
        wave_sync_lds2();
        double lo = ldexp(gl, -E) - 0x1p-44, hi = ldexp(gu, -E) + 0x1p-44;   // (per group of 8 lanes)
        const int grp = row >> 3, t = row & 7, gbase = lane & ~7;
        const int want = r - 1 - grp;                     // ascending index of this group's eigenvalue
        const unsigned rmask = (1u << r) - 1u;            // r <= 31
        const double d0 = deh[0].x;
        const double tr_s = ldexp(tr, -E);                // the trace in the scaled units
        const double frac = (double)(t + 1) * (1.0 / 9.0);
        bool frozen = false;
        int passes = SUBP_PASSES;
        PSTAMP(3);
        for (int pass = 0; pass < SUBP_PASSES; ++pass) {
            const double sigma = fma(hi - lo, frac, lo);
            const double pp = 1.0, pcur = d0 - sigma;
            unsigned mask = (unsigned)__double2hiint(pcur) >> 31;
            const subt_d2 cur = deh[1];
            subt_minor_steps<1>(r, deh, sigma, pp, pcur, cur, mask);
            // bit j of mask = sign of P_(r-j), bit r = 0 = sign of P_0: sign changes = eigenvalues below sigma
            const int cnt = __popc((mask ^ (mask >> 1)) & rmask);
            const unsigned long long above = __ballot(cnt > want);
            const unsigned int mine = (unsigned int)((above >> gbase) & 0xFFull);
            const int first = mine ? __builtin_ctz(mine) : 8;   // first shift of the group that is above the eigenvalue
            const double s_hi = __shfl(sigma, gbase + (first < 8 ? first : 7), 64);
            const double s_lo = __shfl(sigma, gbase + (first > 0 ? first - 1 : 0), 64);
            const double nlo = first > 0 ? s_lo : lo, nhi = first < 8 ? s_hi : hi;
            lo = frozen ? lo : nlo;
            hi = frozen ? hi : nhi;
            // After SUBP_EARLY passes the sum of the four midpoints is good to 4 x 2.2e-14 and 1 - top4 / trace to 9e-14 /
            // tr_s (tr_s ~ 0.1 at worst on count tables): where the score is >= 0.05 it moves by < 9e-12 in that worst case
            // and ~1e-12 typically (k_subscore_tri stops at 2.9e-14 by the same argument) - that half's brackets are
            // final.  The other half goes on; the wave leaves the loop when both are.
            if (pass == SUBP_EARLY - 1) {
                const double tq = subp_all32<SubpAdd>(t == 0 ? fmax(0.5 * (lo + hi), 0.0) : 0.0);
                if (1.0 - tq / tr_s >= 2.5e-3) {
                    frozen = true;
                    passes = pass + 1;
                }
                if (__ballot(frozen) == ~0ull) break;
            }
        }
        PSTAMP(4);
        const double top = subp_all32<SubpAdd>(t == 0 ? fmax(0.5 * (lo + hi), 0.0) : 0.0);
        if (row == 0) {
            const double op = 1.0 - top / tr_s;             // (top in the scaled units)
            const bool ok = tr > 0;
            scores[sid_mine] = ok ? sqrt(op > 0 ? op : 0.0) : __builtin_nan("");
            status[sid_mine] = ok ? passes << 8 : 0;
        }
        PSTAMP(5);
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
// Scores of the splits at positions [start, start + count) of every class (positions index `order`, a device array of
// split indices, or the split list itself when order is NULL).  rt = longest row side of the batch.
int launch_subscore_pair(sp_alignment* al, const int8_t* dtaxa, const int* da, const int* order, const PairClasses& pc, int rt,
                         double* scores_out, int* status_out) {
    sp_ctx* ctx = al->ctx;
    // The class table the kernel reads lives in the context and is uploaded when it changes (a loop over alignments of one
    // shape, the steps of a benchmark: never after the first call).  The copy is ordered behind the stream's earlier
    // kernels; the host copy must not change under it, hence the wait in that (rare) case.
    if (!ctx->pair_valid || memcmp(&ctx->pair_host, &pc, sizeof(PairClasses)) != 0) {
        ctx->pair_valid = false;
        SP_CHECK(ctx->pair_dev.ensure(sizeof(PairClasses)));
        memcpy(&ctx->pair_host, &pc, sizeof(PairClasses));
        SP_HIP(hipMemcpyAsync(ctx->pair_dev.p, &ctx->pair_host, sizeof(PairClasses), hipMemcpyHostToDevice, ctx->stream));
        SP_HIP(hipStreamSynchronize(ctx->stream));
        ctx->pair_valid = true;
    }
    const PairClasses* pcd = ctx->pair_dev.as<PairClasses>();
    const int n = al->n_taxa, mdim = 3 * n + 1;
    SP_REQUIRE(mdim <= SUBT_MMAX && rt <= 31, SP_ELIMIT, "paired subflattening kernel: %d taxa, %d rows", n, rt);
    const bool m32 = al->exact && al->N < ((int64_t)1 << 31);
    const size_t ms_bytes = ((size_t)mdim * mdim * (m32 ? 4 : 8) + 15) & ~(size_t)15;
    const int dev_cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const void* kfn = m32 ? reinterpret_cast<const void*>(k_subscore_pair<true, true>)
                          : (al->exact ? reinterpret_cast<const void*>(k_subscore_pair<true, false>)
                                       : reinterpret_cast<const void*>(k_subscore_pair<false, false>));
    static PerDeviceOnce attr_p;
    if (attr_p.need(ctx->device)) {
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_subscore_pair<true, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_TOTAL));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_subscore_pair<true, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_TOTAL));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_subscore_pair<false, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_TOTAL));
        attr_p.done(ctx->device);
    }
    // Workgroup shape as for k_subscore_tri: the shape that keeps the most waves on a CU (registers or LDS), whole multiples
    // of 4 waves, the smaller workgroup on ties; `subscore_waves` pins it.  Asked once per (kernel, longest side, matrix
    // bytes, pin) and remembered.
    struct ShapeKey { const void* fn; int rt; size_t ms; int pin; int waves, per_cu; };
    static thread_local std::vector<ShapeKey> shape_cache;
    int waves = 0, per_cu = 0;
    bool cached = false;
    for (const ShapeKey& k : shape_cache)
        if (k.fn == kfn && k.rt == rt && k.ms == ms_bytes && k.pin == ctx->opt.subscore_waves) {
            waves = k.waves;
            per_cu = k.per_cu;
            cached = true;
        }
    if (!cached) {
        const int pin = ctx->opt.subscore_waves > 0 ? std::min(ctx->opt.subscore_waves, SUBP_MAXWAVES) : 0;
        for (int wv = pin ? pin : 4; wv <= (pin ? pin : SUBP_MAXWAVES); wv += 4) {
            const size_t lds = ms_bytes + (size_t)wv * subp_wave_bytes(rt);
            if (lds > SPK_LDS_TOTAL) break;
            int pcu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pcu, kfn, wv * 64, lds) != hipSuccess || pcu < 1) continue;
            if (wv * pcu > waves * per_cu) { waves = wv; per_cu = pcu; }
        }
        SP_REQUIRE(waves > 0, SP_ELIMIT, "paired subflattening score: no workgroup shape fits the LDS (%d taxa)", n);
        shape_cache.push_back({kfn, rt, ms_bytes, ctx->opt.subscore_waves, waves, per_cu});
    }
    const size_t lds_t = ms_bytes + (size_t)waves * subp_wave_bytes(rt);
    const int64_t npairs = pc.poff[pc.nclass];
    if (npairs == 0) return SP_OK;
    const int64_t want_blocks = (npairs + waves - 1) / waves;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want_blocks, (int64_t)dev_cus * per_cu));
    if (m32)
        hipLaunchKernelGGL((k_subscore_pair<true, true>), dim3(blocks), dim3(waves * 64), lds_t, ctx->stream, al->moments.p, n, rt,
                           dtaxa, da, order, pcd, scores_out, status_out);
    else if (al->exact)
        hipLaunchKernelGGL((k_subscore_pair<true, false>), dim3(blocks), dim3(waves * 64), lds_t, ctx->stream, al->moments.p, n,
                           rt, dtaxa, da, order, pcd, scores_out, status_out);
    else
        hipLaunchKernelGGL((k_subscore_pair<false, false>), dim3(blocks), dim3(waves * 64), lds_t, ctx->stream, al->moments.p, n,
                           rt, dtaxa, da, order, pcd, scores_out, status_out);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
