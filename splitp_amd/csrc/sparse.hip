// Sparse route: flattening + split score of one split entirely inside one workgroup (LDS; global memory for big tables).
//
// Replaces, like the dense route, splitp/constructions.py:31-55 + splitp/phylogenetics.py:280-300 (and is
// the device form of the reference's own sparse scorer, phylogenetics.py:303-312: top-4 singular values of
// the sparse flattening + its Frobenius norm).  A flattening of a 100 k-site alignment has ~8 k non-zeros
// in up to 10^6 cells, so instead of materialising the matrix (dense route: scatter -> Gram -> eigen, all
// through HBM) the workgroup keeps the D non-zeros as CSC + CSR lists and runs block subspace iteration on the
// implicit Gram operator, one half product at a time:
//       W = C^T V          (sparse, CSC)          Y = C W            (sparse, CSR)
//   one lane per entry (all four columns), groups laid out by size class: a wave / a 16-lane row / a quad / a lane per
//   group, partial sums combined with DPP;
//   after EACH half product the Gram matrix of the fresh block (v_mfma_f64_4x4x4) gives a Ritz sum - its trace, the
//   input block being orthonormal - and the Cholesky factor that re-orthonormalises the block (Cholesky-QR);
//   the sums converge to the sum of the 4 largest squared singular values of C by (sigma_5 / sigma_4)^2 per half product
//   score = sqrt(max(0, 1 - top4 / trace)),  trace = sum of count^2 (exact integer, k_sparse_meta).
// Block width 4: on these matrices lambda_5..lambda_16 are of one magnitude, so guard vectors 5-8 buy almost
// nothing (rate lambda_5/lambda_4 ~ 2e-3 vs lambda_9/lambda_4), while a 4-wide block halves every array.
// Small row sides (<= 64 ids, i.e. |A| <= 3 taxa) have a long column side whose W would not fit, but their
// Gram matrix does: it is accumulated exactly (integer atomics over the pairs inside every column) and the
// iteration runs on it densely.
//
// Every group is summed in table order (stable counting sort with wave-private counters) and every reduction has a
// fixed tree, the Gram accumulation is integer: results are reproducible bit for bit.  Which group sits where inside a
// size class depends on the arrival order of aggregated atomics - that moves groups around, never a sum.
// A split whose arrays do not fit the 160 KiB of LDS is flagged (status 2) and re-run by the same kernel instantiated on
// a slab of global memory (HBM = true); one that has not converged after SPK_MAXIT products goes to the dense route.
#define SPK_MAIN_TU
#include "sparse_common.h"

// Build one list grouped by `major` (CSC: compact column, CSR: compact row) as a STABLE counting sort of the
// table order: the table is cut into SPK_SORT_WAVES contiguous chunks, wave w owns chunk w and a private row of
// per-group counters (BITS-wide fields packed into 32-bit LDS words); position of an entry = ptr[group] + (entries of
// the group in earlier chunks) + (rank among the wave's own earlier entries).  The last term is the value returned by
// the wave's own LDS atomic add: lanes of one ds_add_rtn instruction that hit the same word are resolved by the LDS in
// a fixed order, and no other wave touches that counter row, so the layout - and with it every later summation
// order - is reproducible run to run (checked by the bitwise-repeat tests).  No occupancy bitmaps, no chunking.
//   perm[nmajor] = groups by size class / descending size, ptrp[nmajor + 1] = entry offsets in THAT order (the entries
//   of perm[idx] are ent[ptrp[idx] .. ptrp[idx + 1])), ent[D] (u32 = minor | count << 16), ptr[nmajor] = start of
//   group m (build-time only); *nwave / *nrow = number of groups in the whole-wave / 16-lane-row size classes.
// size class of a group: 0 empty, 1..4 = 1, 2, 3-4, 5-8 entries (one batch of 1 / 2 / 4 / 8), 5..7 = 2 / 3 / 4 batches of 8,
// 8 = one 16-lane row per group (> SPK_TEAM_MAX entries), 9 = one wave per group (> SPK_ROW_MAX entries)
#define SPK_NCLASS 10
__device__ __forceinline__ int spk_class(int c) {
    if (c > SPK_ROW_MAX) return 9;
    if (c > SPK_TEAM_MAX) return 8;
    if (c > 8) return 4 + ((c + 7) >> 3) - 1;   // 9..16 -> 5, 17..24 -> 6, 25..32 -> 7
    return c > 4 ? 4 : (c > 2 ? 3 : c);
}

#define SPK_MAXQ 20   // entries per lane whose counts pass B fetches up front when they live in global memory
// The counts of a lane's SPK_MAXQ consecutive table entries from global memory, as five 16-byte loads (4-byte aligned:
// global memory takes them).  As SPK_MAXQ predicated 4-byte loads - lanes 68 bytes apart, so every one of them touched 64
// cache lines - they kept the texture addresser busy for ~15 k cycles of a list build (tools/gpu_stamps_lists.sh).  Words
// past the lane's share, or past the table, are loaded and never used: every counts buffer is allocated with SP_COUNTS_PAD
// (128) bytes behind its last entry (common.h; api.hip / hist.hip allocation sites), and a start beyond the table is pulled
// back to its end.
typedef u32 spk_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <typename CT>
__device__ __forceinline__ void spk_prefetch_counts(const CT* cnt, int first, int D, u32 (&cpre)[SPK_MAXQ]) {
    static_assert(SPK_MAXQ % 4 == 0, "whole 16-byte loads");
    static_assert(SPK_MAXQ * 4 <= SP_COUNTS_PAD, "the over-read must stay inside the pad of the counts buffers");
    if constexpr (sizeof(CT) == 4) {
        const u32* cb = reinterpret_cast<const u32*>(cnt) + min(first, D);
#pragma unroll
        for (int t4 = 0; t4 < SPK_MAXQ / 4; ++t4) {
            const spk_u32x4_a4 v = *reinterpret_cast<const spk_u32x4_a4*>(cb + 4 * t4);
            cpre[4 * t4] = v.x; cpre[4 * t4 + 1] = v.y; cpre[4 * t4 + 2] = v.z; cpre[4 * t4 + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < SPK_MAXQ; ++t) cpre[t] = first + t < D ? (u32)cnt[first + t] : 0u;
    }
}

// counts of the q entries lane `lane` walks in pass B (i = first + t): from the LDS / slab copy, or - plain LDS form,
// which keeps no copy - straight from the table in global memory, all loads issued before the first use
template <typename CT>
__device__ __forceinline__ u32 spk_count_at(const CT* cnt, int i) { return (u32)cnt[i]; }

template <bool MAJOR_IS_COL, int BITS, bool PERMUTE, typename CT = unsigned short>
__device__ __forceinline__ void spk_build_list(const u32* pc, const CT* cnt, int D, int nmajor,
                                               unsigned short* ptr, unsigned short* ptrp, u32* ent,
                                               unsigned short* perm, int* nwave, int* nrow, int* nquad, int* used, u32* cw,
                                               int nsort, SpkShared& sh, unsigned short* end_of = nullptr,
                                               int stamp0 = -1) {
#ifdef SPK_STAMPS
#define BSTAMP(k) do { if (stamp0 >= 0) SSTAMP(stamp0 + (k)); } while (0)
#else
#define BSTAMP(k)
#endif
    constexpr int PER = 32 / BITS;                     // counters per word
    constexpr u32 FMASK = (1u << BITS) - 1;
    const int stride = (nmajor + PER - 1) / PER;       // words per wave row
    const int lane = threadIdx.x & 63, w = spk_wave_id();
    // entries per lane: ODD, so that the 64 lanes of a load (stride q words) spread over all LDS banks - an even q puts
    // them on 2 .. 16 banks (q = 16: a 32-way conflict on every load of both passes)
    const int q = ((D + nsort * 64 - 1) / (nsort * 64)) | 1;   // nsort = wave-private counter rows in use (<= SPK_WAVES)
    const int chunk = q * 64;
    const int lo = min(D, w * chunk), hi = min(D, lo + chunk);
    for (int i = threadIdx.x; i < nsort * stride; i += SPK_THREADS) cw[i] = 0;
    if (threadIdx.x < 68) sh.bucket[threadIdx.x] = 0;
    __syncthreads();
    BSTAMP(5);
    u32* myrow = cw + (w < nsort ? w : 0) * stride;   // (waves >= nsort have an empty chunk)
    // lane l walks the contiguous sub-chunk [lo + l*q, lo + (l+1)*q): consecutive table entries share their leading
    // digits (hence often their row or column), so giving them to ONE lane keeps the 64 lanes of an atomic on
    // different counters (measured: 64-way same-word conflicts otherwise)
    // (plain loops: the kernel is VALU-issue bound here, a batched / predicated form measured slower)
    for (int t = 0; t < q; ++t) {                         // pass A: per-chunk group sizes
        const int i = lo + lane * q + t;
        if (i >= hi) continue;
        const u32 v = pc[i];
        const int mj = MAJOR_IS_COL ? (int)(v & 0xFFFF) : (int)(v >> 16);
        atomicAdd(&myrow[mj / PER], 1u << (BITS * (mj % PER)));
    }
    __syncthreads();
    BSTAMP(0);
    // exclusive prefix over the chunks, one thread per packed word (PER groups); totals -> ptr
    for (int q = threadIdx.x; q < stride; q += SPK_THREADS) {
        u32 run[PER];
#pragma unroll
        for (int f = 0; f < PER; ++f) run[f] = 0;
        u32 words[SPK_WAVES];
#pragma unroll
        for (int ww = 0; ww < SPK_WAVES; ++ww) words[ww] = ww < nsort ? spk_aload(cw + ww * stride + q) : 0u;
#pragma unroll
        for (int ww = 0; ww < SPK_WAVES; ++ww) {
            if (ww >= nsort) break;
            u32 outw = 0;
#pragma unroll
            for (int f = 0; f < PER; ++f) {
                outw |= run[f] << (BITS * f);
                run[f] += (words[ww] >> (BITS * f)) & FMASK;
            }
            cw[ww * stride + q] = outw;
        }
#pragma unroll
        for (int f = 0; f < PER; ++f)
            if (q * PER + f < nmajor) ptr[q * PER + f] = (unsigned short)run[f];   // group size, for now
    }
    __syncthreads();
    BSTAMP(1);
    if (PERMUTE) {
        // Size classes (spk_class), largest first: whole-wave groups, 16-lane-row groups, quad groups (4 / 3 / 2 batches of
        // 8 entries), lane groups (5-8, 3-4, 2, 1 entries), empty groups last.  Counting sort with wave-aggregated counters: one ballot
        // per class, lane c of the wave adds the class-c population with ONE atomic (per-lane atomics on a handful of
        // hot counters serialise: measured 11 k cycles).  The order INSIDE a class depends on which wave gets there
        // first; that only moves groups around - every group is still summed in table order, so results do not change.
        const int rounds = (nmajor + SPK_THREADS - 1) / SPK_THREADS;
        // (the ballots of the first SPK_KEEP rounds are kept in registers for the placement pass: 10 ballots a round are
        // ~50 instructions, and every instruction of a loop costs the block 16 cycles)
        constexpr int SPK_KEEP = 4;
        int kcls[SPK_KEEP];
        u32 kmine[SPK_KEEP], krank[SPK_KEEP];
        auto classify = [&](int m, int& cls, u32& mine, u32& rank) {
            cls = m < nmajor ? spk_class(ptr[m]) : -1;
            mine = 0;
            u64 mymask = 0;
#pragma unroll
            for (int k = 0; k < SPK_NCLASS; ++k) {
                const u64 b = __ballot(cls == k);
                if (lane == k) mine = (u32)__popcll(b);
                if (cls == k) mymask = b;
            }
            rank = (u32)__popcll(mymask & ((1ull << lane) - 1));
        };
        for (int r = 0; r < rounds; ++r) {
            int cls;
            u32 mine, rank;
            classify(r * SPK_THREADS + (int)threadIdx.x, cls, mine, rank);
#pragma unroll
            for (int kk = 0; kk < SPK_KEEP; ++kk)
                if (r == kk) {
                    kcls[kk] = cls; kmine[kk] = mine; krank[kk] = rank;
                }
            if (lane < SPK_NCLASS && mine) atomicAdd(&sh.bucket[lane], mine);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 run = 0;
            for (int k = SPK_NCLASS - 1; k >= 0; --k) {
                const u32 c = sh.bucket[k];
                sh.bucket[k] = run;
                run += c;
                if (k == SPK_NCLASS - 1) *nwave = (int)run;
                if (k == SPK_NCLASS - 2) *nrow = (int)run - *nwave;
                if (k == 5) *nquad = (int)run;   // (minus the wave / row groups, below)
                if (k == 1 && used) *used = (int)run;   // everything but the empty groups
            }
            *nquad -= *nwave + *nrow;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SPK_KEEP; ++r) {
            if (r >= rounds) break;
            const int m = r * SPK_THREADS + (int)threadIdx.x;
            u32 base = 0;
            if (lane < SPK_NCLASS && kmine[r]) base = atomicAdd(&sh.bucket[lane], kmine[r]);
            base = __shfl(base, kcls[r] < 0 ? 0 : kcls[r], 64);
            if (kcls[r] >= 0) perm[base + krank[r]] = (unsigned short)m;
        }
        for (int r = SPK_KEEP; r < rounds; ++r) {
            const int m = r * SPK_THREADS + (int)threadIdx.x;
            int cls;
            u32 mine, rank;
            classify(m, cls, mine, rank);
            u32 base = 0;
            if (lane < SPK_NCLASS && mine) base = atomicAdd(&sh.bucket[lane], mine);
            base = __shfl(base, cls < 0 ? 0 : cls, 64);
            if (cls >= 0) perm[base + rank] = (unsigned short)m;
        }
        __syncthreads();
        BSTAMP(2);
        {   // entries are laid out in permutation order: ptrp[idx] .. ptrp[idx + 1] = entries of group perm[idx] ...
            const int per = (nmajor + SPK_THREADS - 1) / SPK_THREADS;
            const int l0 = min(nmajor, (int)threadIdx.x * per), h0 = min(nmajor, l0 + per);
            u32 sum = 0;
            for (int i = l0; i < h0; ++i) sum += ptr[perm[i]];
            u32 tot;
            u32 run = spk_scan(sum, sh, tot);
            for (int i = l0; i < h0; ++i) {
                ptrp[i] = (unsigned short)run;
                run += ptr[perm[i]];
            }
            if (threadIdx.x == 0) ptrp[nmajor] = (unsigned short)tot;
            __syncthreads();
            // ... and ptr[m] becomes the start of group m in that layout (placement below)
            for (int i = threadIdx.x; i < nmajor; i += SPK_THREADS) ptr[perm[i]] = ptrp[i];
            __syncthreads();
        }
    } else {   // group order = index order: ptr = exclusive scan of the sizes (small path: no products on the list)
        const int per = (nmajor + SPK_THREADS - 1) / SPK_THREADS;
        const int l0 = min(nmajor, (int)threadIdx.x * per), h0 = min(nmajor, l0 + per);
        u32 sum = 0, nz = 0;
        for (int i = l0; i < h0; ++i) {
            sum += ptr[i];
            nz += ptr[i] != 0;
        }
        u32 tot, tot_nz;
        spk_scan(nz, sh, tot_nz);
        u32 run = spk_scan(sum, sh, tot);
        for (int i = l0; i < h0; ++i) {
            const u32 c = ptr[i];
            ptr[i] = (unsigned short)run;
            run += c;
        }
        if (threadIdx.x == 0) {
            ptr[nmajor] = (unsigned short)tot;
            if (used) *used = (int)tot_nz;
        }
        __syncthreads();
    }
    BSTAMP(3);
    constexpr bool CNT_GLOBAL = !std::is_same<CT, unsigned short>::value;   // counts read from the table in global memory
    u32 cpre[SPK_MAXQ];
    const bool pre = CNT_GLOBAL && q <= SPK_MAXQ;
    if (pre) spk_prefetch_counts(cnt, lo + lane * q, D, cpre);
    auto place = [&](int i, u32 c) {
        const u32 v = pc[i];
        const int mj = MAJOR_IS_COL ? (int)(v & 0xFFFF) : (int)(v >> 16);
        const int mn = MAJOR_IS_COL ? (int)(v >> 16) : (int)(v & 0xFFFF);
        const u32 old = atomicAdd(&myrow[mj / PER], 1u << (BITS * (mj % PER)));
        const int pos = ptr[mj] + (int)((old >> (BITS * (mj % PER))) & FMASK);
        ent[pos] = (u32)mn | (c << 16);
        if (!PERMUTE && end_of) end_of[pos] = (unsigned short)mj;   // small path: group of every position
    };
    if (pre) {                                            // pass B: placement (same walk as pass A)
#pragma unroll
        for (int t = 0; t < SPK_MAXQ; ++t) {
            const int i = lo + lane * q + t;
            if (t < q && i < hi) place(i, cpre[t]);
        }
    } else {
        for (int t = 0; t < q; ++t) {
            const int i = lo + lane * q + t;
            if (i < hi) place(i, (u32)cnt[i]);
        }
    }
    __syncthreads();
}

// Both lists at once (plain LDS form): waves 0..7 build the CSC list while waves 8..15 build the CSR list - the same
// stable counting sort as spk_build_list, every step executed by both halves side by side on their own list, with 8
// wave-private counter rows each.  A list build is a chain of ~10 short barrier-separated steps that are latency-bound,
// not throughput-bound (two passes over the table, a prefix over the counter rows, the size-class permutation, a scan):
// run one after the other they cost 2 x 24 k cycles of a 130 k-cycle workgroup, side by side ~60 % of that.
struct SpkListOut {
    int nmajor;
    unsigned short* ptr;    // build-time start of every group
    unsigned short* ptrp;   // entry offsets in permutation order (nmajor + 1)
    unsigned short* perm;
    u32* ent;
    u32* cw;                // SPK_HALF_WAVES counter rows
    int* nwave;
    int* nrow;
    int* nquad;
    int* used;
};
#define SPK_HALF_WAVES (SPK_WAVES / 2)
#define SPK_HALF_THREADS (SPK_THREADS / 2)

// exclusive scan of one u32 per thread over the thread's HALF of the block (both halves call it together)
__device__ __forceinline__ u32 spk_scan_half(u32 v, SpkShared& sh, u32& total) {
    const int lane = threadIdx.x & 63, w = spk_wave_id(), h = w / SPK_HALF_WAVES, wh = w % SPK_HALF_WAVES;
    u32 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) sh.scan[w] = x;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SPK_HALF_WAVES; ++i) {
        const u32 sv = sh.scan[h * SPK_HALF_WAVES + i];
        if (i < wh) base += sv;
        tot += sv;
    }
    total = tot;
    return base + x - v;
}

template <typename CT>
__device__ __forceinline__ void spk_build_pair(const u32* pc, const CT* cnt, int D, const SpkListOut& Lc,
                                               const SpkListOut& Lr, SpkShared& sh) {
    constexpr int BITS = 16, PER = 2;
    constexpr u32 FMASK = 0xFFFFu;
    const int lane = threadIdx.x & 63, w = spk_wave_id();
    const int h = w / SPK_HALF_WAVES, wh = w % SPK_HALF_WAVES, th = threadIdx.x % SPK_HALF_THREADS;
    const bool is_col = h == 0;
    // (wave-uniform selects: the half's own list)
    const int nmajor = is_col ? Lc.nmajor : Lr.nmajor;
    unsigned short* const ptr = is_col ? Lc.ptr : Lr.ptr;
    unsigned short* const ptrp = is_col ? Lc.ptrp : Lr.ptrp;
    unsigned short* const perm = is_col ? Lc.perm : Lr.perm;
    u32* const ent = is_col ? Lc.ent : Lr.ent;
    u32* const cw = is_col ? Lc.cw : Lr.cw;
    u32* const bucket = sh.bucket + h * 16;               // (sh.bucket has 68 words: 2 x 10 classes fit)
    const int stride = (nmajor + PER - 1) / PER;
    const int q = ((D + SPK_HALF_WAVES * 64 - 1) / (SPK_HALF_WAVES * 64)) | 1;   // odd: see spk_build_list
    const int chunk = q * 64;
    const int lo = min(D, wh * chunk), hi = min(D, lo + chunk);
    for (int i = th; i < SPK_HALF_WAVES * stride; i += SPK_HALF_THREADS) cw[i] = 0;
    if (th < 16) bucket[th] = 0;
    __syncthreads();
    SSTAMP(42);
    u32* const myrow = cw + wh * stride;
    for (int t = 0; t < q; ++t) {                         // pass A: per-chunk group sizes
        const int i = lo + lane * q + t;
        if (i >= hi) continue;
        const u32 v = pc[i];
        const int mj = is_col ? (int)(v & 0xFFFF) : (int)(v >> 16);
        atomicAdd(&myrow[mj / PER], 1u << (BITS * (mj % PER)));
    }
    // counts of this lane's entries for pass B: issued now, consumed ~10 k cycles later
    constexpr bool CNT_GLOBAL = !std::is_same<CT, unsigned short>::value;
    u32 cpre[SPK_MAXQ];
    const bool pre = CNT_GLOBAL && q <= SPK_MAXQ;
    if (pre) spk_prefetch_counts(cnt, lo + lane * q, D, cpre);
    __syncthreads();
    SSTAMP(43);
    for (int qq = th; qq < stride; qq += SPK_HALF_THREADS) {   // exclusive prefix over the 8 chunks; totals -> ptr
        u32 run0 = 0, run1 = 0;
        u32 words[SPK_HALF_WAVES];
#pragma unroll
        for (int ww = 0; ww < SPK_HALF_WAVES; ++ww) words[ww] = cw[ww * stride + qq];
#pragma unroll
        for (int ww = 0; ww < SPK_HALF_WAVES; ++ww) {
            cw[ww * stride + qq] = run0 | (run1 << 16);
            run0 += words[ww] & FMASK;
            run1 += words[ww] >> 16;
        }
        if (qq * 2 < nmajor) ptr[qq * 2] = (unsigned short)run0;
        if (qq * 2 + 1 < nmajor) ptr[qq * 2 + 1] = (unsigned short)run1;
    }
    __syncthreads();
    SSTAMP(46);
    // size classes (spk_class), largest first; wave-aggregated class counters of the half
    const int rounds = (nmajor + SPK_HALF_THREADS - 1) / SPK_HALF_THREADS;
    constexpr int KEEP = 4;
    int kcls[KEEP];
    u32 kmine[KEEP], krank[KEEP];
    auto classify = [&](int m, int& cls, u32& mine, u32& rank) {
        cls = m < nmajor ? spk_class(ptr[m]) : -1;
        mine = 0;
        u64 mymask = 0;
#pragma unroll
        for (int k = 0; k < SPK_NCLASS; ++k) {
            const u64 b = __ballot(cls == k);
            if (lane == k) mine = (u32)__popcll(b);
            if (cls == k) mymask = b;
        }
        rank = (u32)__popcll(mymask & ((1ull << lane) - 1));
    };
    for (int r = 0; r < rounds; ++r) {
        int cls;
        u32 mine, rank;
        classify(r * SPK_HALF_THREADS + th, cls, mine, rank);
#pragma unroll
        for (int kk = 0; kk < KEEP; ++kk)
            if (r == kk) {
                kcls[kk] = cls; kmine[kk] = mine; krank[kk] = rank;
            }
        if (lane < SPK_NCLASS && mine) atomicAdd(&bucket[lane], mine);
    }
    __syncthreads();
    SSTAMP(47);
    if (th == 0) {
        int* const nwave = is_col ? Lc.nwave : Lr.nwave;
        int* const nrow = is_col ? Lc.nrow : Lr.nrow;
        int* const nquad = is_col ? Lc.nquad : Lr.nquad;
        int* const used = is_col ? Lc.used : Lr.used;
        u32 run = 0;
        for (int k = SPK_NCLASS - 1; k >= 0; --k) {
            const u32 c = bucket[k];
            bucket[k] = run;
            run += c;
            if (k == SPK_NCLASS - 1) *nwave = (int)run;
            if (k == SPK_NCLASS - 2) *nrow = (int)run - *nwave;
            if (k == 5) *nquad = (int)run;
            if (k == 1) *used = (int)run;
        }
        *nquad -= *nwave + *nrow;
    }
    __syncthreads();
    SSTAMP(48);
#pragma unroll
    for (int r = 0; r < KEEP; ++r) {
        if (r >= rounds) break;
        const int m = r * SPK_HALF_THREADS + th;
        u32 base = 0;
        if (lane < SPK_NCLASS && kmine[r]) base = atomicAdd(&bucket[lane], kmine[r]);
        base = __shfl(base, kcls[r] < 0 ? 0 : kcls[r], 64);
        if (kcls[r] >= 0) perm[base + krank[r]] = (unsigned short)m;
    }
    for (int r = KEEP; r < rounds; ++r) {
        const int m = r * SPK_HALF_THREADS + th;
        int cls;
        u32 mine, rank;
        classify(m, cls, mine, rank);
        u32 base = 0;
        if (lane < SPK_NCLASS && mine) base = atomicAdd(&bucket[lane], mine);
        base = __shfl(base, cls < 0 ? 0 : cls, 64);
        if (cls >= 0) perm[base + rank] = (unsigned short)m;
    }
    __syncthreads();
    SSTAMP(49);
    {   // entry offsets in permutation order, then ptr[m] = start of group m in that layout
        const int per = (nmajor + SPK_HALF_THREADS - 1) / SPK_HALF_THREADS;
        const int l0 = min(nmajor, th * per), h0 = min(nmajor, l0 + per);
        u32 sum = 0;
        for (int i = l0; i < h0; ++i) sum += ptr[perm[i]];
        u32 tot;
        u32 run = spk_scan_half(sum, sh, tot);
        for (int i = l0; i < h0; ++i) {
            ptrp[i] = (unsigned short)run;
            run += ptr[perm[i]];
        }
        if (th == 0) ptrp[nmajor] = (unsigned short)tot;
        __syncthreads();
        for (int i = th; i < nmajor; i += SPK_HALF_THREADS) ptr[perm[i]] = ptrp[i];
        __syncthreads();
    }
    SSTAMP(52);
    auto place = [&](int i, u32 c) {                      // pass B: placement (same walk as pass A)
        const u32 v = pc[i];
        const int mj = is_col ? (int)(v & 0xFFFF) : (int)(v >> 16);
        const int mn = is_col ? (int)(v >> 16) : (int)(v & 0xFFFF);
        const u32 old = atomicAdd(&myrow[mj / PER], 1u << (BITS * (mj % PER)));
        const int pos = ptr[mj] + (int)((old >> (BITS * (mj % PER))) & FMASK);
        ent[pos] = (u32)mn | (c << 16);
    };
    if (pre) {
#pragma unroll
        for (int t = 0; t < SPK_MAXQ; ++t) {
            const int i = lo + lane * q + t;
            if (t < q && i < hi) place(i, cpre[t]);
        }
    } else {
        for (int t = 0; t < q; ++t) {
            const int i = lo + lane * q + t;
            if (i < hi) place(i, (u32)cnt[i]);
        }
    }
    __syncthreads();
}

// Gram path (short row side): group the entries by column with ONE shared set of packed 16-bit counters - no stable sort,
// no wave-private rows, all 16 waves at work.  The exact Gram matrix G = C C^T is then accumulated with INTEGER atomics over
// the pairs of entries inside every column, and an integer sum does not depend on the order of the entries inside a column
// nor on the order of the columns: whatever order the atomics below resolve in, G comes out the same bit for bit.  (The
// general path's lists feed floating-point sums and therefore need the stable counting sort above.)
//   entry word  = row | count << 16
//   position word = index of the entry inside its column | (column size - 1) << 8      (column size <= R <= 256)
//   ENT64 (lists in global memory): both in ONE 8-byte word per list position, ent64[pos] = entry | position word << 32 -
//   one scattered 8-byte store per entry instead of a 4-byte and a 2-byte one; otherwise ent[pos] and pinfo[pos].
//   csz / ccur: (nmajor + 2) / 2 words of packed counters each (sizes / fill cursors, zeroed here), gstart: nmajor + 2
//   start offsets (all three build-time only)
// Columns are laid out by SIZE CLASS, largest first (> 64, 17 .. 64, 5 .. 16, <= 4 entries; index order inside a class): the
// pair loop gives every list position its column's size / 2 steps, so the 64 consecutive positions of a wave should belong
// to columns of one size (a wave that mixes a 60-entry column with singletons runs 31 steps for a few lanes' sake).
// *used = number of non-empty columns.  All-global form: the counters may live in global memory (atomics are read back with
// agent-scope loads).
template <typename CT, bool ENT64>
__device__ __forceinline__ void spk_group_cols(const u32* pc, const CT* cnt, int D, int nmajor, u32* csz, u32* ccur,
                                               unsigned short* gstart, u32* ent, unsigned short* pinfo, u64* ent64,
                                               int* used, SpkShared& sh) {
    constexpr bool CNT_GLOBAL = !std::is_same<CT, unsigned short>::value;
    const int words = (nmajor + 2) >> 1;
    // lane-contiguous sub-chunks of the table (consecutive entries often share their column: see spk_build_list), odd
    // length so that the lanes of a load spread over the LDS banks
    const int q = ((D + SPK_THREADS - 1) / SPK_THREADS) | 1;
    const int lo = min(D, (int)threadIdx.x * q), hi = min(D, lo + q);
    for (int i = threadIdx.x; i < words; i += SPK_THREADS) {
        csz[i] = 0;
        ccur[i] = 0;
    }
    __syncthreads();
    SSTAMP(55);
    for (int i = lo; i < hi; ++i) {                        // pass A: column sizes
        const int col = (int)(pc[i] & 0xFFFF);
        atomicAdd(&csz[col >> 1], 1u << (16 * (col & 1)));
    }
    u32 cpre[SPK_MAXQ];
    if (CNT_GLOBAL) {   // counts of this thread's first entries for pass B: issued now, used after the scans
        spk_prefetch_counts(cnt, lo, D, cpre);
    }
    __syncthreads();
    SSTAMP(50);
    {   // start offsets of all four classes from ONE block scan: the class sums travel as four 16-bit fields of a 64-bit
        // word (every class total is <= D <= 65535: no field carries into the next), the count of non-empty columns as a
        // 32-bit word next to it (four scans, one per class, cost 14 - 40 k cycles; this one 4 - 10 k)
        const int pw = (words + SPK_THREADS - 1) / SPK_THREADS;
        const int w0 = min(words, (int)threadIdx.x * pw), w1 = min(words, w0 + pw);
        auto cls_sh = [](u32 n) { return n > 64u ? 0 : (n > 16u ? 16 : (n > 4u ? 32 : 48)); };
        u64 sum = 0;
        u32 nz = 0;
        for (int wi = w0; wi < w1; ++wi) {
            const u32 v = spk_aload(csz + wi);
            const u32 a = v & 0xFFFFu, b = v >> 16;
            sum += ((u64)a << cls_sh(a)) + ((u64)b << cls_sh(b));
            nz += (a != 0) + (b != 0);
        }
        const int lane = threadIdx.x & 63, w = spk_wave_id();
        u64 x = sum;
        u32 xn = nz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 y = __shfl_up(x, d, 64);
            const u32 yn = __shfl_up(xn, d, 64);
            if (lane >= d) {
                x += y;
                xn += yn;
            }
        }
        u64* const wsum = reinterpret_cast<u64*>(sh.red);   // (16 + 16 words of the reduction scratch)
        u32* const wnz = reinterpret_cast<u32*>(sh.red + SPK_WAVES);
        __syncthreads();
        if (lane == 63) {
            wsum[w] = x;
            wnz[w] = xn;
        }
        __syncthreads();
        u64 before = 0, total = 0;
        u32 tot_nz = 0;
#pragma unroll
        for (int i = 0; i < SPK_WAVES; ++i) {
            const u64 sv = wsum[i];
            if (i < w) before += sv;
            total += sv;
            tot_nz += wnz[i];
        }
        const u64 excl = before + x - sum;   // exclusive prefix of this thread, per class field
        // class bases: class 0 starts at 0, class c after the totals of the classes before it
        const u32 t0 = (u32)(total & 0xFFFFu), t1 = (u32)((total >> 16) & 0xFFFFu), t2 = (u32)((total >> 32) & 0xFFFFu);
        u32 run0 = (u32)(excl & 0xFFFFu), run1 = t0 + (u32)((excl >> 16) & 0xFFFFu), run2 = t0 + t1 + (u32)((excl >> 32) & 0xFFFFu),
            run3 = t0 + t1 + t2 + (u32)(excl >> 48);
        auto start_of = [&](u32 n) {
            u32 st;
            if (n > 64u) { st = run0; run0 += n; }
            else if (n > 16u) { st = run1; run1 += n; }
            else if (n > 4u) { st = run2; run2 += n; }
            else { st = run3; run3 += n; }
            return st;
        };
        for (int wi = w0; wi < w1; ++wi) {
            const u32 v = spk_aload(csz + wi);
            gstart[2 * wi] = (unsigned short)start_of(v & 0xFFFFu);
            gstart[2 * wi + 1] = (unsigned short)start_of(v >> 16);
        }
        if (threadIdx.x == 0 && used) *used = (int)tot_nz;
        __syncthreads();
    }
    SSTAMP(51);
    auto place = [&](int i, u32 c) {                       // pass B: placement
        const u32 v = pc[i];
        const int col = (int)(v & 0xFFFF);
        const u32 old = atomicAdd(&ccur[col >> 1], 1u << (16 * (col & 1)));
        const u32 idx = (old >> (16 * (col & 1))) & 0xFFFFu;
        const u32 n = (spk_aload(csz + (col >> 1)) >> (16 * (col & 1))) & 0xFFFFu;
        const int pos = (int)gstart[col] + (int)idx;
        const u32 e = (v >> 16) | (c << 16), pi = idx | ((n - 1u) << 8);
        if (ENT64) {
            ent64[pos] = (u64)e | ((u64)pi << 32);
        } else {
            ent[pos] = e;
            pinfo[pos] = (unsigned short)pi;
        }
    };
    if (CNT_GLOBAL) {
        for (int t0 = 0; t0 < q; t0 += SPK_MAXQ) {
            if (t0 > 0) {
                spk_prefetch_counts(cnt, lo + t0, D, cpre);
            }
#pragma unroll
            for (int t = 0; t < SPK_MAXQ; ++t)
                if (t0 + t < q && lo + t0 + t < hi) place(lo + t0 + t, cpre[t]);
        }
    } else {
        for (int i = lo; i < hi; ++i) place(i, (u32)cnt[i]);
    }
    __syncthreads();
}

// out[m][0..3] = sum over the entries e of major group m of count_e * in[minor_e][0..3].
// One LANE handles an entry for all four columns: one entry word, one address, one count conversion and four gathers
// (two ds_read2_b64) per entry - a quarter of the VALU work of a lane-per-column layout, same LDS traffic.  Groups are
// laid out by size class (spk_build_list / spk_class):
//   one wave per group   (> SPK_ROW_MAX entries)   idx <  nwave
//   one 16-lane row      (> SPK_TEAM_MAX)          idx <  nwave + nrow
//   one quad             (9 .. SPK_TEAM_MAX)       idx <  nwave + nrow + nquad
//   one lane             (<= 8, sorted 5-8, 3-4, 2, 1, 0 entries)
// Partial sums of a quad / row are combined with DPP (no LDS traffic), rows of a wave with two shuffles; every order is
// fixed, so the result is reproducible run to run.  Ends with a barrier.
template <int CTRL>
__device__ __forceinline__ double spk_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#ifndef SPK_UW
#define SPK_UW 4            // entries per batch of the wide classes
#endif
#define SPK_DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define SPK_DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define SPK_DPP_ROW_SHR4 0x114
#define SPK_DPP_ROW_SHR8 0x118

// a[0..3] += count * in[minor][0..3] over entries start, start + step, ... < p1, U entries per batch.  PRED = false:
// the caller guarantees (p1 - start) / step is a multiple of U (no tail).
template <int U, bool PRED>
__device__ __forceinline__ void spk_lane_sum(const u32* ent, int start, int step, int p1, const char* in,
                                             int pitch_bytes, int cs, double (&a)[4]) {
    for (int e = start; e < p1; e += U * step) {
        u32 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ent[e + u * step];   // may run into the next group / the padding: masked below
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (PRED) v[u] = (e + u * step < p1) ? v[u] : 0u;     // 0 = count 0 of row 0
        double x[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double* row = reinterpret_cast<const double*>(in + (v[u] & 0xFFFFu) * pitch_bytes);
            x[u][0] = row[0]; x[u][1] = row[cs]; x[u][2] = row[2 * cs]; x[u][3] = row[3 * cs];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double c = (double)(v[u] >> 16);
            a[0] = fma(c, x[u][0], a[0]);
            a[1] = fma(c, x[u][1], a[1]);
            a[2] = fma(c, x[u][2], a[2]);
            a[3] = fma(c, x[u][3], a[3]);
        }
    }
}

__device__ __forceinline__ void spk_store4(double* out, int cs, const double (&a)[4]) {
    out[0] = a[0]; out[cs] = a[1]; out[2 * cs] = a[2]; out[3 * cs] = a[3];
}

__device__ __forceinline__ void spk_spmm(const unsigned short* ptrp, const u32* ent, int nmajor,
                                         const unsigned short* perm, int nwave, int nrow, int nquad, const double* in_d,
                                         int in_pitch, int in_cs, double* out, int out_pitch, int out_cs,
                                         int stamp_at = -1) {
    const int lane = threadIdx.x & 63, w = spk_wave_id();
    const char* in = reinterpret_cast<const char*>(in_d);
    const int pb = in_pitch * 8;
    for (int idx = w; idx < nwave; idx += SPK_WAVES) {                      // one wave per group
        const int p0 = ptrp[idx], p1 = ptrp[idx + 1], m = perm[idx];
        double a[4] = {0, 0, 0, 0};
        if (p1 - p0 <= 128)   // (uniform: one group per wave)
            spk_lane_sum<2, true>(ent, p0 + lane, 64, p1, in, pb, in_cs, a);
        else
            spk_lane_sum<4, true>(ent, p0 + lane, 64, p1, in, pb, in_cs, a);
#ifndef SPK_NOREDUCE
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[c] += spk_dpp<SPK_DPP_QUAD_XOR1>(a[c]);
            a[c] += spk_dpp<SPK_DPP_QUAD_XOR2>(a[c]);
            a[c] += spk_dpp<SPK_DPP_ROW_SHR4>(a[c]);
            a[c] += spk_dpp<SPK_DPP_ROW_SHR8>(a[c]);     // lanes 12..15 of every row: sum over the row
            a[c] += __shfl_xor(a[c], 16, 64);
            a[c] += __shfl_xor(a[c], 32, 64);
        }
#endif
        if (lane == 63) spk_store4(out + m * out_pitch, out_cs, a);
    }
#ifdef SPK_STAMPS
    if (stamp_at >= 0) SSTAMP(stamp_at + 10);
#endif
    {                                                                       // one 16-lane row per group
        const int row = threadIdx.x >> 4, t = threadIdx.x & 15;
        for (int idx = nwave + row; idx < nwave + nrow; idx += SPK_THREADS / 16) {
            const int p0 = ptrp[idx], p1 = ptrp[idx + 1], m = perm[idx];
            double a[4] = {0, 0, 0, 0};
            spk_lane_sum<SPK_UW, true>(ent, p0 + t, 16, p1, in, pb, in_cs, a);
#ifndef SPK_NOREDUCE
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[c] += spk_dpp<SPK_DPP_QUAD_XOR1>(a[c]);
                a[c] += spk_dpp<SPK_DPP_QUAD_XOR2>(a[c]);
                a[c] += spk_dpp<SPK_DPP_ROW_SHR4>(a[c]);
                a[c] += spk_dpp<SPK_DPP_ROW_SHR8>(a[c]);
            }
#endif
            if (t == 15) spk_store4(out + m * out_pitch, out_cs, a);
        }
    }
#ifdef SPK_STAMPS
    if (stamp_at >= 0) SSTAMP(stamp_at + 12);
#endif
    {                                                                       // one quad per group (9 .. 32 entries)
        const int quad = threadIdx.x >> 2, t = threadIdx.x & 3;
        const int q0 = nwave + nrow;
        for (int idx = q0 + quad; idx < q0 + nquad; idx += SPK_THREADS / 4) {
            const int p0 = ptrp[idx], p1 = ptrp[idx + 1], m = perm[idx];
            double a[4] = {0, 0, 0, 0};
            spk_lane_sum<SPK_UW, true>(ent, p0 + t, 4, p1, in, pb, in_cs, a);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[c] += spk_dpp<SPK_DPP_QUAD_XOR1>(a[c]);
                a[c] += spk_dpp<SPK_DPP_QUAD_XOR2>(a[c]);
            }
            if (t == 0) spk_store4(out + m * out_pitch, out_cs, a);
        }
    }
#ifdef SPK_STAMPS
    if (stamp_at >= 0) SSTAMP(stamp_at);
#endif
    for (int idx = nwave + nrow + nquad + (int)threadIdx.x; idx < nmajor; idx += SPK_THREADS) {   // one lane per group
        const int p0 = ptrp[idx], p1 = ptrp[idx + 1], m = perm[idx], len = p1 - p0;
        double a[4] = {0, 0, 0, 0};
        // the classes are sorted, so the lanes of a wave hold groups of (nearly) one size
        if (__ballot(len > 4))
            spk_lane_sum<SPK_UW, true>(ent, p0, 1, p1, in, pb, in_cs, a);
        else if (__ballot(len > 2))
            spk_lane_sum<4, true>(ent, p0, 1, p1, in, pb, in_cs, a);
        else if (__ballot(len != 1) == 0)
            spk_lane_sum<1, false>(ent, p0, 1, p1, in, pb, in_cs, a);
        else
            spk_lane_sum<2, true>(ent, p0, 1, p1, in, pb, in_cs, a);
        spk_store4(out + m * out_pitch, out_cs, a);
    }
    __syncthreads();
}

// status: bit 0 = iteration cap hit (score written but flagged), bit 1 = not handled here (re-score on the
// dense route), bits 8.. = number of operator applications.

// grid = n_al * S workgroups: block b scores split order[b / n_al] of alignment b % n_al (heaviest splits of every
// alignment first); score / status index = alignment * S + split.
//
// HBM = false: every array of the workgroup lives in its 160 KB of LDS (the fast path, tables up to ~9 k patterns at
// 10 taxa).  HBM = true: the same code with every array in a per-workgroup slab of global memory (L2-resident) and only
// SpkShared in LDS - for the splits the LDS form hands back (status 2, e.g. 12 taxa x 100 k sites: 13.5 k patterns, row and
// column sides of ~1800 ids).  Communication inside the workgroup then goes through global memory: plain stores are
// coherent at workgroup scope after __syncthreads(); words that were updated by ATOMICS (key bitmaps, sort counters, the
// integer Gram) are read back with agent-scope atomic loads (spk_aload), since device atomics are done in L2 past the L1.
// One (alignment, split) item, start to finish, by the whole workgroup.  `slab`: this workgroup's piece of global memory
// (HBM: every array; LISTS_GLOBAL: the two entry lists + group descriptors; else unused).  `lds_cap`: 0, or a pretended
// LDS size for the plain LDS form (test switch: exercises the hand-back chain on small tables).  Returns through any of
// its early exits with scores / status of the item written by thread 0 and returns the status word (the same value in
// every thread: all exit conditions are uniform); the caller barriers before LDS is reused.
template <bool HBM, bool WIDE, bool LISTS_GLOBAL, bool W_GLOBAL = false>
__device__ __forceinline__ int spk_score_one(unsigned char* __restrict__ smem, const AlDesc& ad, int ai, int sid, int n,
                                              const SplitDev& sp, int S,
                                              double* __restrict__ scores_all, int* __restrict__ status_all,
                                              unsigned char* __restrict__ slab, size_t slab_bytes, size_t lds_cap,
                                              int wide_cap) {
    SpkShared& sh = *reinterpret_cast<SpkShared*>(smem);
    // wide fallback block (HBM form only: its LDS holds nothing but SpkShared, so the Jacobi workspace sits behind it)
    constexpr bool wide_on = HBM && WIDE;   // (its own instantiation: the extra live state must not cost the others registers)
    EigShared& esh = *reinterpret_cast<EigShared*>(smem + ((sizeof(SpkShared) + 15) & ~(size_t)15));
    constexpr int NBC = wide_on ? SPK_WB : SPK_NB;
    unsigned char* const base = HBM ? slab : smem;
    // (LISTS_GLOBAL: the LDS form with its two entry lists - written once, read sequentially - in a small slab of
    // global memory, which is what lets a 13 k-pattern table keep its staging arrays, counters and the V / W blocks in LDS)
    static_assert(!(HBM && LISTS_GLOBAL), "LISTS_GLOBAL is a variant of the LDS form");
    // W_GLOBAL: the lists-in-global form with the column-side block W in the slab as well.  A 4|8 split of a 12-taxon
    // table has ~5000 used columns: W alone is the whole LDS, but V (256 rows), the staged table and the sort counters
    // fit - so only the products' gathers of W (L2 hits, thousands in flight) and the Gram / Cholesky-QR passes over W
    // leave the LDS, not the whole build as in the all-global form (331 -> 260 us of one CU per item).
    static_assert(!W_GLOBAL || LISTS_GLOBAL, "W_GLOBAL is a variant of the lists-in-global form");
    const size_t cap = HBM ? slab_bytes : ((lds_cap && !LISTS_GLOBAL) ? lds_cap : (size_t)SPK_LDS_BYTES);
    const u32* __restrict__ keys = ad.keys32;
    const u32* __restrict__ counts = ad.counts;
    const SpkMeta* __restrict__ meta = ad.meta;
    const int64_t D = ad.D;
    double* __restrict__ scores = scores_all + (int64_t)ai * S;
    int* __restrict__ status = status_all + (int64_t)ai * S;
    const int nr = sp.nr, nc = sp.nc, rw = sp.rw, cw = sp.cw;
    if (threadIdx.x < 32) {
        const int t = threadIdx.x < nr + nc ? sp.taxa[threadIdx.x] : 0;
        sh.shifts[threadIdx.x] = 2 * (n - 1 - t);
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + SPK_NTOP) sh.top[threadIdx.x - 64] = meta->top[threadIdx.x - 64];
    if (threadIdx.x == 64 + SPK_NTOP) {
        sh.trace = meta->trace;
        sh.ntop = meta->ntop;
        sh.ntab = meta->pad;
    }
    __syncthreads();  // shifts are read by every wave below
    size_t off = HBM ? 0 : (sizeof(SpkShared) + 15) & ~(size_t)15;
    auto carve = [&](size_t bytes) {
        unsigned char* p = base + off;
        off = (off + bytes + 15) & ~(size_t)15;
        return p;
    };
    // ---- stage the table in LDS once: pc[i] = row id << 16 | col id, cnt[i] = count --------------------------------
    // (every later pass reads LDS; a pass over global memory costs D / SPK_THREADS serialised load latencies)
    // A side of at most 5 taxa keeps its raw base-4 id (<= 1024 ids: the V / W rows of unused ids stay zero); a longer
    // side is compacted to the ids in use with a presence bitmap + popcount ranks.
#ifndef SPK_RAW_MAX
#define SPK_RAW_MAX 5
#endif
    const bool raw_r = nr <= SPK_RAW_MAX, raw_c = nc <= SPK_RAW_MAX;
    const int rwl = raw_r ? 0 : rw;   // bitmap words in use on the row side
    const int Di = (int)D;
    // a-priori bounds on the matrix sizes (the actual sizes of compacted sides are known only after ranking)
    const int kc_cap = raw_c ? (1 << (2 * nc)) : (int)min((long long)D, nc >= 8 ? (long long)D : (1ll << (2 * nc)));
    const int r_cap = raw_r ? (1 << (2 * nr)) : (int)min((long long)D, nr >= 8 ? (long long)D : (1ll << (2 * nr)));
    // Gram path: the row side is short enough for the exact Gram matrix G = C C^T to sit in LDS and for the iteration to run
    // on it densely; then no CSR list is needed and the CSC grouping needs no order (spk_group_cols).
    //   small: up to SPK_SMALL_R (64) row ids, G as fp64 (integer accumulation in 32 or 64 bits)
    //   mid:   up to SPK_MID_R (256) row ids, lists-in-global form only: G as the packed upper triangle of 32-bit integers,
    //          131.6 KB at 256 rows (needs trace < 2^32) - the 4|8 splits of a 12-taxon table, whose column-side block W
    //          (5000 ids x 4 x 8 bytes) does not fit the LDS
    // (a table with split counts - several rows per pattern - pairs the pieces of one cell: such tables take the general path)
    const bool g32 = sh.trace < (1ull << 32);
    const bool small_sure = r_cap <= SPK_SMALL_R && (u32)Di == sh.ntab && !wide_on;
    // (mid: exactly the raw 4-taxon row side, R = SPK_MID_R - its product runs without bounds checks)
    const bool mid_sure = LISTS_GLOBAL && !W_GLOBAL && !small_sure && raw_r && r_cap == SPK_MID_R && (u32)Di == sh.ntab && g32 && !wide_on;
    const bool gpath = small_sure || mid_sure;
    // A column side of 10 and more taxa has too many possible ids for presence bitmaps in LDS (4^10 bits + rank prefixes =
    // 196 KB: the 2|10 splits of a 12-taxon table used to fall through to the all-global form for them alone).  The Gram
    // path needs no ORDER on the column ids, only equal ids for equal columns: an open-addressing hash table over the
    // column keys hands out its slot numbers instead (H >= 1.125 D slots of 4 bytes; unused slots are empty columns).
    // Plain LDS form (the fast kernel): its Gram path keeps round 2's STABLE list build (wave-private 8-bit counter rows,
    // column pointers + column of every position) and leaves the order-free grouping, the hash relabelling and the packed
    // G to the slow kernel's forms.  Measured on config 2: the new grouping makes the 2|8 and 3|7 splits 6.5 % faster when a
    // launch holds one size class only, but the whole mixed launch 2.3 % slower (0.1025 against 0.1002 ms per step, three
    // A/B rounds on one box) - the fast kernel grew from 240 to 292 KB of code and its general path, which decides the
    // makespan of a launch, pays for that in instruction fetch.
    constexpr bool GP_STABLE = !HBM && !LISTS_GLOBAL;
    const bool hash_c = !HBM && !GP_STABLE && gpath && !raw_c && (size_t)cw * 12 > 49152;
    int H = 1024;
    while (H < Di + (Di >> 3) + 16) H <<= 1;
    const int cwl = (raw_c || hash_c) ? 0 : cw;   // bitmap words in use on the column side
    const int W = rwl + cwl;
    // LDS forms keep no copy of the counts in LDS: pass B of the list builds fetches them from the table in global memory
    // (all of a lane's loads issued up front), which frees 2 D bytes for a second set of sort counters
    constexpr bool PLAIN = !HBM && !LISTS_GLOBAL;
    // (general path with the lists in global memory: one list after the other with the counts staged next to the table,
    // 8 + 16 counter rows, measured 89 k cycles for both lists of a 13.5 k-pattern table; counts fetched from global memory
    // with 12 + 16 rows: 100 - 108 k; both lists side by side: 105 - 112 k)
    const bool stage_cnt = HBM || (LISTS_GLOBAL && !gpath);
    const size_t need_build = off + (LISTS_GLOBAL ? 0 : (size_t)(D + 8) * (gpath ? 4 : 8)) + (size_t)(D + 8) * (stage_cnt ? 6 : 4) +
                              (size_t)W * 12 + (hash_c ? (size_t)H * 4 : 0) + 4096 + 256;
    if (D > 65535 || n > 16 || need_build + 2048 > cap) {
        if (threadIdx.x == 0) {
            scores[sid] = 0.0;
            status[sid] = 2;
        }
        SFORM(1);
        return 2;
    }
    (void)kc_cap;
    SSTAMP(0);
    // region A (persistent): CSC list, CSR list.  Region B: staging + bitmaps while building, then V and W (or G).
    // Sizes that depend on R / Kc are carved after the ranks are known.
    const size_t list_bytes = ((size_t)(D + 8) * 4 + 15) & ~(size_t)15;   // + 8: the unpredicated tail reads of the products
    unsigned char* const lslab = LISTS_GLOBAL ? slab : nullptr;
    // (LISTS_GLOBAL: the group descriptors / permutations of both lists - read in sequence by the products - go there too)
    const size_t gdesc_bytes = (((size_t)max((long long)D, 1024ll) + 2) * 2 + 15) & ~(size_t)15;
    const size_t wslab_off = 2 * list_bytes + 4 * gdesc_bytes;   // W_GLOBAL: W sits behind the lists and descriptors
    if (LISTS_GLOBAL && wslab_off + (W_GLOBAL ? ((size_t)max((long long)D, 1024ll) + 8) * SPK_NB * 8 : 0) > slab_bytes) {
        if (threadIdx.x == 0) {
            scores[sid] = 0.0;
            status[sid] = 2;
        }
        SFORM(2);
        return 2;
    }
    // (Gram path with the lists in global memory: ONE list of 8-byte words - entry + position word - over both list slots)
    u32* csc_ent = LISTS_GLOBAL ? reinterpret_cast<u32*>(lslab) : reinterpret_cast<u32*>(carve((size_t)(D + 8) * 4));
    u64* const ent64 = LISTS_GLOBAL ? reinterpret_cast<u64*>(lslab) : nullptr;
    u32* csr_ent = gpath ? nullptr
                              : (LISTS_GLOBAL ? reinterpret_cast<u32*>(lslab + list_bytes)
                                              : reinterpret_cast<u32*>(carve((size_t)(D + 8) * 4)));
    if (threadIdx.x < 8 && !gpath) {   // (the products' unpredicated tail reads)
        csc_ent[D + threadIdx.x] = 0;
        if (csr_ent) csr_ent[D + threadIdx.x] = 0;
    }
    const size_t off_after_lists = off;
    u32* pc = reinterpret_cast<u32*>(carve((size_t)(D + 8) * 4));
    unsigned short* cnt = stage_cnt ? reinterpret_cast<unsigned short*>(carve((size_t)D * 2)) : nullptr;
    u64* bm = reinterpret_cast<u64*>(carve((size_t)W * 8));
    u32* pf = reinterpret_cast<u32*>(carve((size_t)W * 4));
    const int* shifts = sh.shifts;
    for (int i = threadIdx.x; i < W; i += SPK_THREADS) bm[i] = 0;
    // cell = sum_t digit_t(key) << dst_t is a bit permutation of the key: done per key BYTE through four 256-entry
    // tables (4 taxa each) built here - 3 look-ups + 2 ORs per pattern instead of 4 instructions per taxon (the
    // staging loop is instruction-bound: 16 waves share 4 SIMDs)
    u32* lut = reinterpret_cast<u32*>(carve(4 * 256 * 4));
    u32* htab = hash_c ? reinterpret_cast<u32*>(carve((size_t)H * 4)) : nullptr;
    // The table's keys (and counts) come from global memory, SPK_SU loads in flight per thread; the plain LDS form issues
    // its first - for every table it can hold, only - batch HERE, before the look-up tables are built, and consumes it
    // after the barrier: under load a batch waits 4 - 5 k cycles for the L2, and the old 8-load batches took two of them
    // for a 8.2 k-pattern table (34 patterns in the second).
    constexpr int SPK_SU = PLAIN ? 10 : (HBM ? 8 : 14);   // (lists-in-global forms: one batch for a 14 k-pattern table)
    u32 key[SPK_SU], cv[SPK_SU];
    auto load_batch = [&](int base0) {
#pragma unroll
        for (int u = 0; u < SPK_SU; ++u) {
            const int i = base0 + u * SPK_THREADS + (int)threadIdx.x;
            key[u] = i < Di ? keys[i] : 0u;
            if (!PLAIN) cv[u] = (stage_cnt && i < Di) ? counts[i] : 0u;
        }
    };
    if (!HBM) load_batch(0);
    {
        for (int e = threadIdx.x; e < 1024; e += SPK_THREADS) {
            const int ch = e >> 8, v = e & 255;
            u32 out = 0;
            for (int t = 0; t < nr + nc; ++t) {
                const int src = shifts[t];
                const int dst = t < nr ? 2 * (nc + nr - 1 - t) : 2 * (nr + nc - 1 - t);
                if ((src >> 3) == ch) out |= (u32)((v >> (src & 7)) & 3) << dst;
            }
            lut[e] = out;
        }
    }
    const u32 cmask = (nc >= 16) ? 0xFFFFFFFFu : ((1u << (2 * nc)) - 1);
    const bool both_raw = raw_r && raw_c;
    const bool wide_key = n > 12;   // bits 24..31 in use
    __syncthreads();   // bitmaps are zero
    for (int base0 = 0; base0 < Di; base0 += SPK_THREADS * SPK_SU) {
        if (HBM || base0 > 0) load_batch(base0);
#pragma unroll
        for (int u = 0; u < SPK_SU; ++u) {
            const int i = base0 + u * SPK_THREADS + (int)threadIdx.x;
            if (i >= Di) continue;
            u32 cell = lut[key[u] & 255u] | lut[256 + ((key[u] >> 8) & 255u)] | lut[512 + ((key[u] >> 16) & 255u)];
            if (wide_key) cell |= lut[768 + (key[u] >> 24)];
            const u32 r = cell >> (2 * nc), c = cell & cmask;
            pc[i] = both_raw ? ((r << 16) | c) : cell;
            if (!PLAIN && stage_cnt) cnt[i] = (unsigned short)cv[u];
            if (!raw_r) {   // presence bits (a plain read first saves most of the atomics)
                const u64 rb = 1ull << (r & 63);
                if (!(*(volatile u64*)(bm + (r >> 6)) & rb)) atomicOr(bm + (r >> 6), rb);
            }
            if (!raw_c && !hash_c) {
                const u64 cb = 1ull << (c & 63);
                if (!(*(volatile u64*)(bm + rwl + (c >> 6)) & cb)) atomicOr(bm + rwl + (c >> 6), cb);
            }
        }
    }
    __syncthreads();
    if (hash_c)
        for (int i = threadIdx.x; i < H; i += SPK_THREADS) htab[i] = 0;
    int dimsRC[2] = {1 << (2 * (raw_r ? nr : 0)), hash_c ? H : 1 << (2 * (raw_c ? nc : 0))};
    for (int which = 0; which < 2; ++which) {
        if (which ? (raw_c || hash_c) : raw_r) continue;
        const int wbase = which ? rwl : 0, cntw = which ? cwl : rwl;
        const int per = (cntw + SPK_THREADS - 1) / SPK_THREADS;
        const int lo = min(cntw, (int)threadIdx.x * per), hi = min(cntw, lo + per);
        u32 s = 0;
        for (int i = lo; i < hi; ++i) s += __popcll(spk_aload(bm + wbase + i));
        u32 tot;
        u32 run = spk_scan(s, sh, tot);
        for (int i = lo; i < hi; ++i) {
            pf[wbase + i] = run;
            run += __popcll(spk_aload(bm + wbase + i));
        }
        dimsRC[which] = (int)tot;
    }
    __syncthreads();
    SSTAMP(1);
    const int R = dimsRC[0], Kc = dimsRC[1];   // matrix sizes (ids), >= the rows / columns in use on raw sides
    const double trace = (double)sh.trace;
    // min(shape) <= 4: the reference's SVD returns at most 4 values, so it computes 1 - x/x = 0 exactly; an all-zero
    // table gives 0/0 = nan.  (raw sides: the number of ids in use is known after the lists are built, checked there)
    auto degenerate = [&](int used_r, int used_c) {
        if (used_r > 4 && used_c > 4 && trace > 0) return false;
        if (threadIdx.x == 0) {
            scores[sid] = trace > 0 ? 0.0 : __builtin_nan("");
            status[sid] = 0;
        }
        return true;
    };
    if (degenerate(raw_r ? 5 : R, (raw_c || hash_c) ? 5 : Kc)) return 0;
    if (hash_c) {   // column id = slot of the column key in the hash table (empty = 0, key c stored as c + 1; nc <= 15)
        for (int i = threadIdx.x; i < Di; i += SPK_THREADS) {
            const u32 cell = pc[i];
            const u32 r = cell >> (2 * nc), c = cell & cmask;
            u32 slot = (c * 0x9E3779B1u) >> 7;
            slot = (slot ^ (slot >> 11)) & (u32)(H - 1);
            for (;;) {   // (H > D: an empty slot always turns up)
                const u32 old = atomicCAS(&htab[slot], 0u, c + 1u);
                if (old == 0u || old == c + 1u) break;
                slot = (slot + 1u) & (u32)(H - 1);
            }
            pc[i] = ((raw_r ? r : bm_rank(bm, pf, r)) << 16) | slot;
        }
        __syncthreads();
    } else if (!both_raw) {   // compact coordinates in place: pc[i] = rr << 16 | cc (8 look-ups in flight per thread)
        for (int base = 0; base < Di; base += SPK_THREADS * 8) {
            u32 cell[8], rr[8], cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * SPK_THREADS + (int)threadIdx.x;
                cell[u] = i < Di ? pc[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const u32 r = cell[u] >> (2 * nc), c = cell[u] & cmask;
                rr[u] = raw_r ? r : bm_rank(bm, pf, r);
                cc[u] = raw_c ? c : bm_rank(bm + rwl, pf + rwl, c);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * SPK_THREADS + (int)threadIdx.x;
                if (i < Di) pc[i] = (rr[u] << 16) | cc[u];
            }
        }
        __syncthreads();
    }
    const bool small = gpath;        // Gram path (either form of G); everything else is the general path
    const bool mid = mid_sure;       // ... with G as a packed triangle of 32-bit integers
    const int Rp = (R + 3) & ~3;
    // group descriptors + permutations (now that R and Kc are known) are carved top-down from the end of LDS; the key
    // bitmaps are dead: the counting-sort counters start where they were
    size_t top = cap;
    auto carve_top = [&](size_t bytes) {
        top = (top - bytes) & ~(size_t)15;
        return base + top;
    };
    auto carve_desc = [&](int which, size_t bytes) {
        return reinterpret_cast<unsigned short*>(LISTS_GLOBAL ? lslab + 2 * list_bytes + (size_t)which * gdesc_bytes
                                                              : carve_top(bytes));
    };
    unsigned short* desc_c = small ? nullptr : carve_desc(0, (size_t)(Kc + 1) * 2);
    unsigned short* desc_r = small ? nullptr : carve_desc(1, (size_t)(R + 1) * 2);
    unsigned short* perm_c = small ? nullptr : carve_desc(2, (size_t)Kc * 2);
    unsigned short* perm_r = small ? nullptr : carve_desc(3, (size_t)R * 2);
    const size_t build_end = reinterpret_cast<unsigned char*>(bm) - base;
    // V and W are column-major: four arrays of Rp / Kcp doubles.  A lane's four gathers then go to four arrays -
    // measured 9 % faster than four consecutive doubles of one row (fewer LDS bank conflicts), and no row padding.
    // (+ 4: the four columns of one row start on different banks - the dense small-side product reads them together)
    const int Kcp = ((Kc + 3) & ~3) + 4;
    const int Vp = Rp + 4;
    // (W_GLOBAL keeps the column-major layout: row-major [column id][4] - one 32-byte access per gathered row - measured
    // the same, 267 against 259 us per item: the passes over W wait on L2 latency, not on sectors)
    const int v_rs = 1, v_cs = Vp, w_rs = 1, w_cs = Kcp;
    const size_t base_iter = off_after_lists + (size_t)Vp * NBC * 8 + 16;
    const int Gp = R | 1;   // odd row pitch of the dense G: a pitch of 64 doubles puts every row on the same LDS bank
    const size_t gtri_bytes = ((size_t)R * (R + 1) / 2) * 4;   // mid: packed upper triangle, 32-bit cells
    const size_t need_iter = base_iter + (small ? (mid ? gtri_bytes : (size_t)R * Gp * 8) : ((W_GLOBAL ? 0 : (size_t)Kcp * NBC * 8)));
    // Gram path: index inside its column / column size of every list position (for the entry-parallel pair loop below);
    // with the lists in global memory it goes there as well, into one of the four descriptor slots - what stays in LDS is
    // the staged table, the counters and G
    unsigned short* pinfo = (small && !LISTS_GLOBAL && !GP_STABLE) ? carve_desc(2, (size_t)Di * 2) : nullptr;
    // (plain LDS form: column pointers and the column of every list position, see GP_STABLE)
    unsigned short* csc_ptr = (small && GP_STABLE) ? carve_desc(0, (size_t)(Kc + 1) * 2) : nullptr;
    unsigned short* colof = (small && GP_STABLE) ? carve_desc(2, (size_t)Di * 2) : nullptr;
    // general path: wave-private counter rows of 16-bit fields; every wave gets its own row when that fits (shorter
    // chunks per lane), else fewer waves sort.  Gram path: one shared row + the start offsets (spk_group_cols).
    const size_t cw_c1 = small ? (GP_STABLE ? (size_t)((Kc + 3) / 4) * 4   // (8-bit fields: a column has at most R <= 64 entries)
                                            : 2 * (size_t)((Kc + 2) / 2) * 4 + (((size_t)Kc + 2) * 2 + 15 & ~(size_t)15))
                               : (size_t)((Kc + 1) / 2) * 4;
    const size_t cw_r1 = (size_t)((R + 1) / 2) * 4;
    const size_t grp_r_probe = small ? 0 : (((size_t)R + 1) * 2 + 15) & ~(size_t)15;
    // All-global form: the sort counters - the only words of the build that take an atomic per entry - go to the
    // otherwise idle LDS when at least 4 wave rows fit (global atomics made the two list builds 78 % of a 4|8 split of
    // the 12-taxon table).
    const size_t lds_used = ((sizeof(SpkShared) + 15) & ~(size_t)15) + sizeof(EigShared) + 32;
    const size_t lds_free = HBM ? (size_t)SPK_LDS_BYTES - lds_used : 0;
    const int ns_c_lds = HBM ? (small ? (cw_c1 + 16 <= lds_free ? 4 : 0) : (int)min((size_t)SPK_WAVES, lds_free / max(cw_c1, (size_t)4))) : 0;
    const int ns_r_lds = (HBM && !small) ? (int)min((size_t)SPK_WAVES, lds_free / max(cw_r1, (size_t)4)) : 0;
    const bool cwc_lds = ns_c_lds >= 4, cwr_lds = ns_r_lds >= 4;
    auto rows_that_fit = [&](size_t row_bytes, size_t extra) {   // wave-private counter rows: as many of the 16 as fit
        int ns = SPK_WAVES;
        while (ns > 1 && build_end + (size_t)ns * row_bytes + extra + 16 > top) --ns;
        return ns;
    };
    // (the group starts of the sequential CSC build sit behind its counter rows - general path, lists in global memory -
    // and have to fit as well: without the allowance a side whose 16 rows just fitted was refused, hand-back to the all-global form)
    const size_t grp_c_room = (small || (!LISTS_GLOBAL && ((size_t)Kc + 1) * 2 <= ((size_t)D + 8) * 4)) ? 0 : ((((size_t)Kc + 1) * 2 + 15) & ~(size_t)15) + 16;
    const int ns_c = (small && !GP_STABLE) ? 1 : (cwc_lds ? ns_c_lds : rows_that_fit(cw_c1, grp_c_room));
    const int ns_r = cwr_lds ? ns_r_lds
                             : (build_end + SPK_WAVES * cw_r1 + grp_r_probe + 16 <= top ? SPK_WAVES : SPK_SORT_WAVES);
    const size_t cw_c = cwc_lds ? 0 : (size_t)ns_c * cw_c1;                  // bytes taken behind the staging arrays
    const size_t cw_r = (small || cwr_lds) ? 0 : (size_t)ns_r * cw_r1;
    // build-time start of every group (general path): columns - in the not yet written CSR list; rows - right behind
    // the row counters
    const size_t grp_r_bytes = small ? 0 : (((size_t)R + 1) * 2 + 15) & ~(size_t)15;
    // (a raw column side of a tiny table has more ids than the CSR list has bytes: then behind the column counters)
    const bool grp_c_in_csr = !small && !LISTS_GLOBAL && ((size_t)Kc + 1) * 2 <= ((size_t)D + 8) * 4;
    const size_t grp_c_bytes = (small || grp_c_in_csr) ? 0 : (((size_t)Kc + 1) * 2 + 15) & ~(size_t)15;
    if (need_iter > top || build_end + cw_c + grp_c_bytes + 16 > top || build_end + cw_r + grp_r_bytes + 16 > top ||
        Kc > 65535) {
        if (threadIdx.x == 0) {
            scores[sid] = 0.0;
            status[sid] = 2;
        }
        SFORM(3);
        return 2;
    }
    u32* const cw_lds = reinterpret_cast<u32*>(smem + lds_used);
    u32* cwbuf = cwc_lds ? cw_lds : reinterpret_cast<u32*>(base + build_end);
    u32* cwbuf_r = cwr_lds ? cw_lds : reinterpret_cast<u32*>(base + build_end);
    unsigned short* grp_c = grp_c_in_csr ? reinterpret_cast<unsigned short*>(csr_ent)
                                         : reinterpret_cast<unsigned short*>(base + ((build_end + cw_c + 15) & ~(size_t)15));
    unsigned short* grp_r = reinterpret_cast<unsigned short*>(base + ((build_end + cw_r + 15) & ~(size_t)15));
    if (small) {   // Gram path: order-free grouping by column (one shared counter row + start offsets behind the staging arrays)
        if (GP_STABLE) {
            spk_build_list<true, 8, false, u32>(pc, counts, Di, Kc, csc_ptr, nullptr, csc_ent, nullptr, nullptr, nullptr, nullptr,
                                                &sh.used_c, cwbuf, ns_c, sh, colof);
        }
        const size_t gwords = (size_t)((Kc + 2) / 2);
        u32* const csz = cwbuf;
        u32* const ccur = cwbuf + gwords;
        unsigned short* const gstart = reinterpret_cast<unsigned short*>(cwbuf + 2 * gwords);
        if (HBM)
            spk_group_cols<unsigned short, false>(pc, cnt, Di, Kc, csz, ccur, gstart, csc_ent, pinfo, nullptr, &sh.used_c, sh);
        else if (!GP_STABLE)
            spk_group_cols<u32, LISTS_GLOBAL>(pc, counts, Di, Kc, csz, ccur, gstart, csc_ent, pinfo, ent64, &sh.used_c, sh);
        SSTAMP(2);
    } else if (!HBM) {
        // both lists side by side (waves 0..7 / 8..15) when two sets of 8 counter rows + both group-start arrays fit - plain
        // LDS form only: with the lists in global memory (13.5 k patterns: 27 entries per lane and pass) the two halves do
        // not overlap, 105 - 112 k cycles against 89 k for one list after the other (8 + 16 counter rows)
        const size_t pair_c = (size_t)SPK_HALF_WAVES * cw_c1, pair_r = (size_t)SPK_HALF_WAVES * cw_r1;
        const size_t pair_gc = (((size_t)Kc + 1) * 2 + 15) & ~(size_t)15, pair_gr = (((size_t)R + 1) * 2 + 15) & ~(size_t)15;
        const size_t pair_at = (build_end + 15) & ~(size_t)15;
        const bool pair = !LISTS_GLOBAL && pair_at + pair_c + pair_r + pair_gc + pair_gr + 16 <= top;
        if (LISTS_GLOBAL) {
            spk_build_list<true, 16, true>(pc, cnt, Di, Kc, grp_c, desc_c, csc_ent, perm_c, &sh.nw_c, &sh.nr_c, &sh.nq_c, &sh.used_c,
                                           cwbuf, ns_c, sh, nullptr, 50);
            SSTAMP(2);
            spk_build_list<false, 16, true>(pc, cnt, Di, R, grp_r, desc_r, csr_ent, perm_r, &sh.nw_r, &sh.nr_r, &sh.nq_r, &sh.used_r,
                                            cwbuf_r, ns_r, sh);
        } else if (pair) {
            unsigned char* at = base + pair_at;
            SpkListOut Lc{Kc, reinterpret_cast<unsigned short*>(at + pair_c + pair_r), desc_c, perm_c, csc_ent,
                          reinterpret_cast<u32*>(at), &sh.nw_c, &sh.nr_c, &sh.nq_c, &sh.used_c};
            SpkListOut Lr{R, reinterpret_cast<unsigned short*>(at + pair_c + pair_r + pair_gc), desc_r, perm_r, csr_ent,
                          reinterpret_cast<u32*>(at + pair_c), &sh.nw_r, &sh.nr_r, &sh.nq_r, &sh.used_r};
            spk_build_pair<u32>(pc, counts, Di, Lc, Lr, sh);
            SSTAMP(2);
        } else {
            spk_build_list<true, 16, true, u32>(pc, counts, Di, Kc, grp_c, desc_c, csc_ent, perm_c, &sh.nw_c, &sh.nr_c, &sh.nq_c,
                                                &sh.used_c, cwbuf, ns_c, sh, nullptr, 50);
            SSTAMP(2);
            spk_build_list<false, 16, true, u32>(pc, counts, Di, R, grp_r, desc_r, csr_ent, perm_r, &sh.nw_r, &sh.nr_r, &sh.nq_r,
                                                 &sh.used_r, cwbuf_r, ns_r, sh);
        }
    } else {
        spk_build_list<true, 16, true>(pc, cnt, Di, Kc, grp_c, desc_c, csc_ent, perm_c, &sh.nw_c, &sh.nr_c, &sh.nq_c, &sh.used_c,
                                       cwbuf, ns_c, sh, nullptr, 50);
        SSTAMP(2);
        spk_build_list<false, 16, true>(pc, cnt, Di, R, grp_r, desc_r, csr_ent, perm_r, &sh.nw_r, &sh.nr_r, &sh.nq_r, &sh.used_r,
                                        cwbuf_r, ns_r, sh);
    }
    SSTAMP(3);
    // ids in use on raw sides (small path: the used rows are the non-zero diagonal entries of G, checked below)
    if (degenerate(small ? 5 : sh.used_r, sh.used_c)) return 0;
    // ---- start block: unit vectors on the rows of the 4 largest counts (distinct rows) ----------------------------
    // (the dominant singular vectors of a count flattening sit on the few very frequent patterns); chosen by four
    // rounds of a block arg-max over (count, index), deterministic tie-break.
    int top_row[SPK_NB], top_cnt[SPK_NB];
    {
        // rows of the SPK_NTOP most frequent patterns (found once per alignment, k_sparse_meta); the first 4 distinct
        // rows among them, missing ones (fewer than 4 distinct) filled with the lowest unused row ids.  Any 4 strong
        // distinct rows make a good start block; the hash noise below covers the directions they miss.
        int* rows_out = reinterpret_cast<int*>(sh.S);   // [0..3] rows, [4..7] largest count of each
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const int ntop = (int)sh.ntop;
            int myrow = -1, mycnt = 1;
            if (lane < ntop) {
                myrow = (int)(pc[sh.top[lane]] >> 16);
                mycnt = (!PLAIN && stage_cnt) ? (int)cnt[sh.top[lane]] : (int)counts[sh.top[lane]];
            }
            u64 active = __ballot(myrow >= 0);
            // (fully unrolled, constant indices: a runtime-indexed local array would live in scratch = global memory)
            int chosen[SPK_NB] = {-1, -1, -1, -1}, ccnt[SPK_NB] = {1, 1, 1, 1};
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k) {
                if (active) {
                    const int first = __builtin_ctzll(active);
                    const int r = __shfl(myrow, first, 64);
                    chosen[k] = r;
                    ccnt[k] = __shfl(mycnt, first, 64);
                    active &= ~__ballot(myrow == r);
                }
            }
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k) {
                if (chosen[k] < 0) {   // fewer than 4 distinct rows among the candidates: lowest row id not taken yet
                    int r = 0;
#pragma unroll
                    for (int t = 0; t < SPK_NB; ++t)
                        r += (r == chosen[0]) | (r == chosen[1]) | (r == chosen[2]) | (r == chosen[3]);
                    chosen[k] = r;
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < SPK_NB; ++k) {
                    rows_out[k] = chosen[k];
                    rows_out[SPK_NB + k] = ccnt[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPK_NB; ++k) {
            top_row[k] = rows_out[k];
            top_cnt[k] = rows_out[SPK_NB + k];
        }
        __syncthreads();
    }
    SSTAMP(4);
    // V and W / G are laid out over the (now dead) staging area
    off = off_after_lists;
    double* V = reinterpret_cast<double*>(carve((size_t)Vp * NBC * 8));
    double* Wb = (W_GLOBAL && !small) ? reinterpret_cast<double*>(lslab + wslab_off)
                                      : reinterpret_cast<double*>(base + off);   // large: W (Kc x 4);  small: G (R x R)
    // mid path: cell (lo, hi), lo <= hi, of the packed upper triangle (R rows: lo (2R + 1 - lo) is even)
    auto tri = [&](int lo, int hi) { return (int)(__umul24((u32)lo, (u32)(2 * R + 1 - lo)) >> 1) + (hi - lo); };
    if (small) {
        // exact Gram G = C C^T: all pairs of entries inside every column, accumulated with integer LDS atomics (exact,
        // order independent).  One thread per list position i of a column of n entries takes the pairs (i, i + d mod n),
        // d = 0 .. n/2 - every unordered pair once, every thread of a column the same number of steps (walking the
        // rest of the column instead leaves half of the lanes of a long column idle).  32-bit atomics when every entry
        // fits (G[r][r'] <= trace), 64-bit otherwise (small form only).
        u32* G32 = reinterpret_cast<u32*>(Wb);
        unsigned long long* G64 = reinterpret_cast<unsigned long long*>(Wb);
        const int gwords = mid ? (int)(gtri_bytes / 4) : (g32 ? R * Gp : 2 * R * Gp);   // (small: integer G with the same odd pitch as the fp64 one)
        for (int i = threadIdx.x; i < gwords; i += SPK_THREADS) G32[i] = 0;
        __syncthreads();
        SSTAMP(44);
        // (instruction count is what this loop costs - 16 waves share 4 SIMDs, so every instruction of the body is 16
        // cycles of the block: position and size of the entry in its column come in one 16-bit word (spk_group_cols), the
        // partner index walks and wraps by compare-select, the cell is min * Gp + max)
        // (lists in global memory: entry and position word come in one 8-byte word, partners are the low halves; eight
        // partner entries are fetched before the first of their atomics - a dependent L2 round trip per step otherwise)
        auto entry_at = [&](int q) -> u32 { return LISTS_GLOBAL ? (u32)ent64[q] : csc_ent[q]; };
        auto pair_loop = [&](auto add_cell) {
            for (int a = threadIdx.x; a < Di; a += SPK_THREADS) {
                u32 va;
                int info;
                int i, n, p0, pend;
                if (GP_STABLE) {
                    const int col = colof[a];
                    p0 = csc_ptr[col];
                    pend = csc_ptr[col + 1];
                    n = pend - p0;
                    i = a - p0;
                    va = csc_ent[a];
                } else {
                    if (LISTS_GLOBAL) {
                        const u64 w = ent64[a];
                        va = (u32)w;
                        info = (int)(w >> 32);
                    } else {
                        va = csc_ent[a];
                        info = pinfo[a];
                    }
                    i = info & 255;
                    n = (info >> 8) + 1;
                    p0 = a - i;
                    pend = p0 + n;
                }
                const int half = n >> 1;
                const u32 ca = va >> 16;
                const int ra = va & 0xFFFF;
                const int steps = ((n & 1) == 0 && i >= half) ? half : half + 1;   // even n: the pair (i, i + n/2) belongs to i < n/2
                int q = a;
                if (LISTS_GLOBAL) {
                    for (int d = 0; d < steps; d += 8) {
                        u32 vb[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            vb[u] = entry_at(q);      // (d + u >= steps: a valid position of the column, not used)
                            q = q + 1 == pend ? p0 : q + 1;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (d + u >= steps) break;
                            const int rb = vb[u] & 0xFFFF;
                            add_cell(min(ra, rb), max(ra, rb), ca, vb[u] >> 16);
                        }
                    }
                } else {
                    for (int d = 0; d < steps; ++d) {
                        const u32 vb = entry_at(q);
                        q = q + 1 == pend ? p0 : q + 1;
                        const int rb = vb & 0xFFFF;
                        add_cell(min(ra, rb), max(ra, rb), ca, vb >> 16);
                    }
                }
            }
        };
        if (mid)
            pair_loop([&](int lo, int hi, u32 c1, u32 c2) { atomicAdd(&G32[tri(lo, hi)], c1 * c2); });
        else if (g32)
            pair_loop([&](int lo, int hi, u32 c1, u32 c2) { atomicAdd(&G32[lo * Gp + hi], c1 * c2); });
        else
            pair_loop([&](int lo, int hi, u32 c1, u32 c2) {
                atomicAdd(&G64[lo * Gp + hi], (unsigned long long)c1 * (unsigned long long)c2);
            });
        __syncthreads();
        SSTAMP(45);
        if (mid) {   // rows in use = non-zero diagonal cells; G stays packed integers (converted on the fly by the product)
            if (threadIdx.x == 0) sh.used_r = 0;
            __syncthreads();
            const int dr = (int)threadIdx.x < R ? (int)threadIdx.x : 0;
            const u64 nzd = __ballot((int)threadIdx.x < R && G32[tri(dr, dr)] != 0u);
            if ((threadIdx.x & 63) == 0 && nzd) atomicAdd(&sh.used_r, (int)__popcll(nzd));
            __syncthreads();
        } else {
            {   // exact integer -> fp64 in place (same R x Gp layout), through registers: 4- / 8-byte cells overlap
                constexpr int NG = (SPK_SMALL_R * (SPK_SMALL_R + 1) + SPK_THREADS - 1) / SPK_THREADS;
                double gv[NG];
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int i = k * SPK_THREADS + (int)threadIdx.x;
                    gv[k] = i < R * Gp ? (g32 ? (double)spk_aload(G32 + i) : (double)spk_aload(G64 + i)) : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int i = k * SPK_THREADS + (int)threadIdx.x;
                    if (i < R * Gp) Wb[i] = gv[k];
                }
            }
            __syncthreads();
            const float pinv = 1.0f / (float)Gp;   // i / Gp for i < 4160, Gp <= 65: (i + 0.5) / Gp is >= 1/130 away from an integer
            for (int i = threadIdx.x; i < R * Gp; i += SPK_THREADS) {
                const int r = (int)(((float)i + 0.5f) * pinv), c = i - r * Gp;
                if (c < R && r > c) Wb[r * Gp + c] = Wb[c * Gp + r];
            }
            if (threadIdx.x < 64) {   // rows in use = non-zero diagonal entries (R <= 64)
                const u64 nzd = __ballot((int)threadIdx.x < R && Wb[threadIdx.x * Gp + threadIdx.x] != 0.0);
                if (threadIdx.x == 0) sh.used_r = (int)__popcll(nzd);
            }
            __syncthreads();
        }
        if (degenerate(sh.used_r, 5)) return 0;
    }
    // ---- iteration: alternate half products, one Ritz sum per half product -------------------------------------
    //   h odd :  W = C^T V  (V orthonormal)  ->  trace(W^T W) = trace(V^T C C^T V) = Ritz sum of C C^T on span(V)
    //   h even:  Y = C W    (W orthonormal)  ->  trace(Y^T Y) = Ritz sum of C^T C on span(W)
    // Both are Rayleigh-Ritz sums of the same four squared singular values, each better than the last by
    // (sigma_5 / sigma_4)^2; the block is re-orthonormalised by one Cholesky-QR step on the Gram matrix that has just
    // been formed (no eigen-decomposition, no polar factor).  Small row side: dense G instead of the two sparse halves.
    // Start: unit vectors on the 4 rows picked above.  General path: C^T of unit vectors is just those 4 rows of C, so
    // the first half product is a scatter of 4 CSR rows into W (plus 1 % hash noise for the directions they miss).
    double prev_sum = 0, prev_delta = 0, prev_ratio = 1.0, top4 = 0, prev_th5 = 0, prev_d5 = 0, prev_sum8 = 0;
    int wide_settled = 0;
    int it = 0, conv = 0;
    if (small) {
        for (int i = threadIdx.x; i < R; i += SPK_THREADS) {
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k)
                V[i * v_rs + k * v_cs] = 0.02 * spk_hash(i, k) + (top_row[k] == i ? 1.0 : 0.0);
        }
        __syncthreads();
        spk_gram(V, R, v_rs, v_cs, sh);
        spk_chol_factor(sh);
        spk_orth(V, R, v_rs, v_cs, sh);
    } else {
        const double nscale = 0.01 * spk_rsqrt((double)Kc * (1.0 / 3.0));
        double amp[SPK_NB];
#pragma unroll
        for (int k = 0; k < SPK_NB; ++k) amp[k] = nscale * (double)top_cnt[k];
        for (int c = threadIdx.x; c < Kc; c += SPK_THREADS) {   // one 32-bit hash per row, a signed byte of it per column
            u32 h = (u32)c * 0x9E3779B1u + 0x7F4A7C15u;
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k)
                Wb[c * w_rs + k * w_cs] = amp[k] * (double)((float)((int)(h << (8 * k)) >> 24) * (1.0f / 128.0f));
            if (wide_on) {   // columns 4..7 of the wide block: noise only
                u32 g = h * 0x85EBCA77u + 0x165667B1u;
                g ^= g >> 13; g *= 0xC2B2AE3Du; g ^= g >> 16;
#pragma unroll
                for (int k = 0; k < SPK_NB; ++k)
                    Wb[c * w_rs + (SPK_NB + k) * w_cs] = (double)((float)((int)(g << (8 * k)) >> 24) * (1.0f / 128.0f));
            }
        }
        for (int idx = threadIdx.x; idx < R; idx += SPK_THREADS) {   // where do the 4 rows sit in the CSR layout?
            const int m = perm_r[idx];
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k)
                if (m == top_row[k]) sh.bucket[k] = (u32)idx;
        }
        __syncthreads();
        // A count >= 2^16 enters the table as several rows (pieces <= 65535) of ONE cell, and the SUM of the pieces belongs
        // into W.  Rounds 1 - 2 let every piece store its own value - whichever thread came last won, the start block
        // differed from run to run and with it the last bits of such tables' scores (randomised sweep: up to 3.6e-10
        // apart on scores of 1e-5).  Tables with pieces therefore clear the cells first and ADD: the values are integers,
        // so the fp64 atomic sum is exact in any order (the pieces need not even sit next to each other in the list).
        if ((u32)Di != sh.ntab) {   // (uniform)
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k) {
                const int idx = (int)sh.bucket[k];
                const int p0 = desc_r[idx], p1 = desc_r[idx + 1];
                for (int e = p0 + (int)threadIdx.x; e < p1; e += SPK_THREADS) Wb[(csr_ent[e] & 0xFFFFu) * w_rs + k * w_cs] = 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k) {
                const int idx = (int)sh.bucket[k];
                const int p0 = desc_r[idx], p1 = desc_r[idx + 1];
                for (int e = p0 + (int)threadIdx.x; e < p1; e += SPK_THREADS) {
                    const u32 v = csr_ent[e];
                    atomicAdd(&Wb[(v & 0xFFFFu) * w_rs + k * w_cs], (double)(v >> 16));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < SPK_NB; ++k) {
                const int idx = (int)sh.bucket[k];
                const int p0 = desc_r[idx], p1 = desc_r[idx + 1];
                for (int e = p0 + (int)threadIdx.x; e < p1; e += SPK_THREADS) {
                    const u32 v = csr_ent[e];
                    Wb[(v & 0xFFFFu) * w_rs + k * w_cs] = (double)(v >> 16);
                }
            }
        }
        __syncthreads();
        if (wide_on) {
            double t0, t1, t2;
            spk_wide_ritz_orth(Wb, Kc, w_cs, esh, t0, t1, t2);   // here only as an orthonormaliser
        } else {
            spk_gram(Wb, Kc, w_rs, w_cs, sh);
            spk_chol_factor(sh);
            spk_orth(Wb, Kc, w_rs, w_cs, sh);
        }
    }
    SSTAMP(5);
    SSTAMP(6);
    if (small) {
        const u32* const Gt = reinterpret_cast<const u32*>(Wb);
        for (it = 1; it <= SPK_MAXIT; ++it) {
            // Y = G V densely; Ritz sum = trace(V^T Y); Y staged in registers, written over V
            if (it == 2) SSTAMP(56);
            int row = threadIdx.x >> 2, j = threadIdx.x & 3;
            double acc = 0, part = 0;
            if (mid) {
                // R = 256, G packed integers.  Plain fp64 FMAs, not the matrix cores: with only 4 right-hand columns a
                // v_mfma_f64_4x4x4 (256 FMA) occupies the matrix pipe as long as a 16 x 16 x 4 (1024 FMA) - measured
                // 22 k cycles per product, against 4 k cycles for the same 262 k FMA at the vector rate.  Thread = 4 rows x
                // every 16th k: lane (s16 = lane & 15, row group = lane >> 4) of wave w takes rows 16 w + 4 (lane >> 4) .. + 3
                // and k = s16, 16 + s16, ...: the 16 lanes of a group read 16 consecutive cells of a G row and 16 consecutive
                // rows of V per instruction (no bank conflicts), a V row is fetched once per 4 rows of G, and the 16 partial
                // sums of a (row, column) meet in the DPP row reduction of spk_spmm.
                const int lane = threadIdx.x & 63, w = spk_wave_id();
                const int s16 = lane & 15, r0 = 16 * w + 4 * (lane >> 4);
                double a4[4][4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int c = 0; c < 4; ++c) a4[rr][c] = 0.0;
                // cell (r, k) of the packed triangle: rowb[r] + k for k >= r, colb(k) + r for k < r.  A wave's rows are
                // 16 w .. 16 w + 15 and step kk covers k = 16 kk .. 16 kk + 15, so the case is uniform except at kk == w: one
                // add per cell instead of the eight integer operations of tri() (the loop is VALU-issue bound: every
                // instruction costs the block 16 cycles).
                int rowb[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) rowb[rr] = tri(r0 + rr, r0 + rr) - (r0 + rr);
                auto step = [&](int kk, auto which) {
                    const int k = 16 * kk + s16;
                    double v[4], gcell[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = V[k * v_rs + c * v_cs];
                    const int colb = tri(k, k) - k;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        int cell;
                        if (decltype(which)::value == 0) cell = colb + r0 + rr;            // k < r
                        else if (decltype(which)::value == 2) cell = rowb[rr] + k;         // k > r
                        else cell = k >= r0 + rr ? rowb[rr] + k : colb + r0 + rr;
                        gcell[rr] = (double)Gt[cell];
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int c = 0; c < 4; ++c) a4[rr][c] = fma(gcell[rr], v[c], a4[rr][c]);
                };
#pragma unroll 2
                for (int kk = 0; kk < w; ++kk) step(kk, std::integral_constant<int, 0>{});
                step(w, std::integral_constant<int, 1>{});
#pragma unroll 2
                for (int kk = w + 1; kk < SPK_MID_R / 16; ++kk) step(kk, std::integral_constant<int, 2>{});
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        double x = a4[rr][c];
                        x += spk_dpp<SPK_DPP_QUAD_XOR1>(x);
                        x += spk_dpp<SPK_DPP_QUAD_XOR2>(x);
                        x += spk_dpp<SPK_DPP_ROW_SHR4>(x);
                        x += spk_dpp<SPK_DPP_ROW_SHR8>(x);     // lane 15 of every 16-lane row: the whole sum
                        a4[rr][c] = x;
                    }
                part = 0.0;
                if (s16 == 15) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int c = 0; c < 4; ++c) part = fma(a4[rr][c], V[(r0 + rr) * v_rs + c * v_cs], part);
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                if (it == 2) SSTAMP(57);
                __syncthreads();
                if (s16 == 15) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int c = 0; c < 4; ++c) V[(r0 + rr) * v_rs + c * v_cs] = a4[rr][c];
                }
                row = R;   // (the common tail below writes nothing more)
            } else if (row < R) {
                for (int k = 0; k < R; ++k) acc = fma(Wb[row * Gp + k], V[k * v_rs + j * v_cs], acc);
                part = acc * V[row * v_rs + j * v_cs];
            }
            if (!mid) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                if (it == 2) SSTAMP(57);
                __syncthreads();
                if (row < R) V[row * v_rs + j * v_cs] = acc;
            }
            if ((threadIdx.x & 63) == 0) sh.red[spk_wave_id()] = part;
            __syncthreads();
            top4 = 0;
#pragma unroll
            for (int ww = 0; ww < SPK_WAVES; ++ww) top4 += sh.red[ww];
            __syncthreads();   // spk_gram reuses sh.red: nobody may still be summing it (a rare race, caught after 36 rounds)
            spk_gram(V, R, v_rs, v_cs, sh);                  // Y^T Y = V^T G^2 V: its eigenvalues are the squared Ritz values
            if (it == 2) SSTAMP(58);
            // (S holds SQUARED Ritz values here: the cheap bound suffices when rest^2 <= 0.09 of it, i.e. rest <= 0.3 sqrt)
            const double rest_s = trace - top4;
            // (the Gram path's early certified stop - spk_converged<true>: 3rd sum, one measured ratio - only in the slow
            // kernel's forms: in the fast kernel it made the 2|8 and 3|7 splits of config 2 faster one class at a time and the
            // pipelined mixed launch 0.6 % slower, 0.1016 against 0.1010 ms per step, A/B on one box: the plain LDS form keeps
            // round 2's rule)
            constexpr bool EARLY = !GP_STABLE;
            spk_chol_factor(sh, it >= (EARLY ? 3 : 4), rest_s > 0 ? rest_s * rest_s * (1.0 / 0.3) : 0.0);
            if (spk_converged<EARLY>(top4, sh.L[11] > 0 ? sp_fsqrt(sh.L[11]) : 0.0, trace, it, prev_sum, prev_delta, prev_ratio)) {
                conv = 1;
                break;
            }
            if (it == 2) SSTAMP(59);
            spk_orth(V, R, v_rs, v_cs, sh);
            if (it == 2) SSTAMP(60);
        }
    } else {
        SSTAMP(7);
        const int maxhalf = wide_on ? (wide_cap > 0 ? wide_cap : SPK_MAXHALF_WIDE) : SPK_MAXHALF;
        for (it = 2; it <= maxhalf; ++it) {
            // (one call site: the product code is inlined once, the kernel has to stay inside the 64 KB instruction cache;
            // the wide block makes two 4-column passes)
            const bool odd = it & 1;                          // odd: W = C^T V (CSC)   even: Y = C W (CSR)
            double* X = odd ? Wb : V;
            const int rows = odd ? Kc : R, xrs = odd ? w_rs : v_rs, xcs = odd ? w_cs : v_cs;
            const int ics = odd ? v_cs : w_cs;
            for (int cb = 0; cb < NBC; cb += SPK_NB)
                spk_spmm(odd ? desc_c : desc_r, odd ? csc_ent : csr_ent, rows, odd ? perm_c : perm_r, odd ? sh.nw_c : sh.nw_r,
                         odd ? sh.nr_c : sh.nr_r, odd ? sh.nq_c : sh.nq_r, (odd ? V : Wb) + (size_t)cb * ics, odd ? v_rs : w_rs,
                         ics, X + (size_t)cb * xcs, xrs, xcs, it == 3 ? 20 : -1);
            if (wide_on) {
                double th4, sum8;
                double th5, thmin;
                spk_wide_ritz_orth(X, rows, xcs, esh, top4, th4, sum8, &th5, &thmin);
                const int verdict = spk_wide_converged(top4, th4, sum8, trace, it - 1, prev_sum, prev_delta, prev_ratio, th5, prev_th5,
                                                       prev_d5, thmin, wide_settled, prev_sum8);
                if (verdict) {   // 1: certified; 2: nothing will certify this block - flagged (status bit 0), the direct solver's
                    conv = verdict == 1;
                    break;
                }
                continue;
            }
            if (it == 2) SSTAMP(9);
            if (it == 3) SSTAMP(8);
            spk_gram(X, rows, xrs, xcs, sh);
            if (it == 2) SSTAMP(40);
            top4 = (sh.S[0] + sh.S[5]) + (sh.S[10] + sh.S[15]);
            spk_chol_factor(sh, it >= 5, trace - top4);
            // (the first real Ritz sum is that of half product 2)
            if (spk_converged(top4, sh.L[11], trace, it - 1, prev_sum, prev_delta, prev_ratio)) {
                conv = 1;
                break;
            }
            spk_orth(X, rows, xrs, xcs, sh);
            if (it == 2) SSTAMP(41);
        }
    }
    SSTAMP(11);
    // no spectral gap behind the 4th value (conv == 0): the Ritz sum so far is a lower bound of the top-4 sum, so the
    // score written is an upper estimate; status bit 1 hands the split on (wide block / dense route), the wide block
    // - the last resort - flags it with bit 0 instead.
    const int code = conv ? (it << 8) : ((wide_on ? 1 : 2) | (it << 8));
    SFORM(0);
    if (threadIdx.x == 0) {
        const double op = 1.0 - top4 / trace;
        scores[sid] = sqrt(op > 0 ? op : 0.0);
        status[sid] = code;
    }
    return code;
}

// ---- the kernels around spk_score_one ---------------------------------------------------------------------------------
// k_sparse_score: grid = n_al * S workgroups, block b scores the (b / n_al)-th split of the launch order (heaviest splits
// of every alignment first) of alignment b % n_al in the in-LDS form; score / status index = alignment * S + split.
// Block 0 also zeroes the work-queue head of the slow kernel queued behind it on the same stream.
// A workgroup's first loads are all one hop from the kernel arguments: `launch` holds the split descriptors IN launch
// order with the split's own index in `cls` (no order[] -> splits[] chain), and a single alignment's descriptor comes by
// value (no als[] -> keys chain).  Under load a dependent global load costs 2.5 - 5 k cycles, and the staging of a late
// workgroup used to wait for three of them in a row (24 k cycles against 13 k for the first workgroups of a launch).
__global__ __launch_bounds__(SPK_THREADS) void k_sparse_score(const AlDesc* __restrict__ als, const AlDesc al0, int n_al,
                                                              int n, const SplitDev* __restrict__ launch, int S,
                                                              double* __restrict__ scores_all,
                                                              int* __restrict__ status_all, int* __restrict__ queue_head,
                                                              size_t lds_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (queue_head && blockIdx.x == 0 && threadIdx.x == 0) *queue_head = 0;
    const int ai = (int)(blockIdx.x % n_al);
    const SplitDev& sp = launch[blockIdx.x / n_al];
    if (n_al == 1)
        (void)spk_score_one<false, false, false>(smem, al0, 0, sp.cls, n, sp, S, scores_all, status_all, nullptr, 0, lds_cap, 0);
    else
        (void)spk_score_one<false, false, false>(smem, als[ai], ai, sp.cls, n, sp, S, scores_all, status_all, nullptr, 0,
                                                 lds_cap, 0);
}

// k_sparse_slow: the hand-back chain ON THE DEVICE, queued behind k_sparse_score - no host round trip.  A fixed grid of
// persistent workgroups, each with its own slab of global memory for its whole life (a slab never changes CU, so plain
// loads / stores stay coherent through that CU's L1 and its XCD's L2), draws chunks of consecutive items from an atomic
// head and takes the items whose status word asks for it:
//   status == 2 ("does not fit the LDS form", no half product run): the lists-in-global form, then the all-global form;
//   status 2 + half products (the 4-wide block found no certified gap): the 8-wide block - always when `wide_all`, else
//   only when the smaller side is beyond the dense route's EIG_MAXR rows (the synchronous entry point hands the others
//   to the dense route).
// The three forms are real (non-inlined) functions.  Every workgroup ends on its one failed draw; a pass with nothing to
// do costs each of its (few) workgroups one atomic and one status load.
struct SpkSlow {
    int* head;
    unsigned char* slabs;
    size_t slab_stride;              // bytes per workgroup
    size_t slab_l, slab_h, slab_w;   // what the lists-in-global / all-global / wide forms may use of it
    int chunk;                       // items per draw
    int wide_all;
    int wide_cap;                    // half-product cap of the wide block (0 = SPK_MAXHALF_WIDE; tests of SP_ENOCONV lower it)
};

template <bool HBM, bool WIDE, bool LISTS_GLOBAL, bool W_GLOBAL = false>
__device__ __noinline__ int spk_score_slow(const AlDesc* __restrict__ als, int ai, int sid, int n,
                                           const SplitDev* __restrict__ splits, int S, double* __restrict__ scores_all,
                                           int* __restrict__ status_all, unsigned char* __restrict__ slab,
                                           size_t slab_bytes, int wide_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    return spk_score_one<HBM, WIDE, LISTS_GLOBAL, W_GLOBAL>(smem, als[ai], ai, sid, n, splits[sid], S, scores_all, status_all,
                                                            slab, slab_bytes, 0, wide_cap);
}

__global__ __launch_bounds__(SPK_THREADS) void k_sparse_slow(const AlDesc* __restrict__ als, int n_al, int n,
                                                             const SplitDev* __restrict__ splits, int S,
                                                             double* __restrict__ scores_all,
                                                             int* __restrict__ status_all, SpkSlow q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SpkShared& sh = *reinterpret_cast<SpkShared*>(smem);
    unsigned char* const slab = q.slabs + (size_t)blockIdx.x * q.slab_stride;
    const int total = n_al * S;
    const int nchunks = (total + q.chunk - 1) / q.chunk;
    const int lane = threadIdx.x & 63;
    for (;;) {
        __syncthreads();   // the previous item is finished with the LDS
        if (threadIdx.x == 0) sh.qchunk = atomicAdd(q.head, 1);
        __syncthreads();
        const int chunk = sh.qchunk;
        if (chunk >= nchunks) break;   // (uniform: every workgroup ends on its one failed draw)
        const int item = chunk * q.chunk + lane;   // (q.chunk <= 64: one status word per lane, the same in every wave)
        bool want = false;
        int st_mine = 0;
        if (lane < q.chunk && item < total) {
            const int st = st_mine = status_all[item];
            want = st == 2;
            if (!want && (st & 2) && (st >> 8) != 0) {
                want = q.wide_all != 0;
                if (!want) {
                    const SplitDev& sp = splits[item % S];
                    const long long side = 1ll << (2 * min(sp.nr, sp.nc));
                    const long long rmax = min(side, (long long)als[item / S].D);
                    want = ((rmax + 63) & ~63ll) > EIG_MAXR;
                }
            }
        }
        unsigned long long mask = __ballot(want);
        while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int it = chunk * q.chunk + b;
            const int ai = it / S, sid = it % S;
            int st = __shfl(st_mine, b, 64);
            if (st == 2) {
                __syncthreads();
                st = spk_score_slow<false, false, true>(als, ai, sid, n, splits, S, scores_all, status_all, slab, q.slab_l, 0);
            }
            if (st == 2) {   // ... then with W in global memory too
                __syncthreads();
                st = spk_score_slow<false, false, true, true>(als, ai, sid, n, splits, S, scores_all, status_all, slab, q.slab_l,
                                                              0);
            }
            if (st == 2) {
                __syncthreads();
                st = spk_score_slow<true, false, false>(als, ai, sid, n, splits, S, scores_all, status_all, slab, q.slab_h, 0);
            }
            if ((st & 2) && (st >> 8) != 0) {
                bool wide = q.wide_all != 0;
                if (!wide) {
                    const SplitDev& sp = splits[sid];
                    const long long side = 1ll << (2 * min(sp.nr, sp.nc));
                    const long long rmax = min(side, (long long)als[ai].D);
                    wide = ((rmax + 63) & ~63ll) > EIG_MAXR;
                }
                if (wide) {
                    __syncthreads();
                    (void)spk_score_slow<true, true, false>(als, ai, sid, n, splits, S, scores_all, status_all, slab, q.slab_w,
                                                            q.wide_cap);
                }
            }
        }
    }
}

// Once per alignment: keys narrowed to 32 bits, trace = sum count^2, the SPK_NTOP largest counts (descending, ties by
// lowest index).  One 1024-thread block; D <= 65535.
// trace_override != 0: the table is an expanded one (large counts entered in pieces, common.h) and the caller supplies the
// trace of the real counts.
__global__ __launch_bounds__(1024) void k_sparse_meta(const u64* __restrict__ keys, const u32* __restrict__ counts, int D,
                                                      u32* __restrict__ keys32, SpkMeta* __restrict__ meta,
                                                      unsigned long long trace_override, u32 orig_rows) {
    __shared__ unsigned long long red[16];
    __shared__ unsigned long long winner;
    const int lane = threadIdx.x & 63, w = spk_wave_id();
    unsigned long long tr = 0;
    for (int i = threadIdx.x; i < D; i += 1024) {
        keys32[i] = (u32)keys[i];
        const unsigned long long c = counts[i];
        tr += c * c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);
    if (lane == 0) red[w] = tr;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; ++i) t += red[i];
        meta->trace = trace_override ? trace_override : t;
        meta->ntop = (u32)(D < SPK_NTOP ? D : SPK_NTOP);
        meta->pad = orig_rows;
    }
    unsigned long long bound = ~0ull;   // candidates = (count << 32) | (0xFFFFFFFF - index): extracted in descending order
    for (int k = 0; k < SPK_NTOP; ++k) {
        unsigned long long best = 0;
        for (int i = threadIdx.x; i < D; i += 1024) {
            const unsigned long long cand = ((unsigned long long)counts[i] << 32) | (0xFFFFFFFFull - (unsigned)i);
            if (cand < bound && cand > best) best = cand;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(best, d, 64);
            best = o > best ? o : best;
        }
        __syncthreads();
        if (lane == 0) red[w] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long b = 0;
            for (int i = 0; i < 16; ++i) b = red[i] > b ? red[i] : b;
            winner = b;
            meta->top[k] = b ? (u32)(0xFFFFFFFFull - (b & 0xFFFFFFFFull)) : 0u;
        }
        __syncthreads();
        bound = winner ? winner : 0;   // 0: table exhausted (k >= D), the remaining slots are not used (ntop)
        if (!bound) bound = 0;
    }
}

int launch_sparse_meta(sp_ctx* ctx, const u64* keys, const u32* counts, int64_t D, u32* keys32, SpkMeta* meta,
                       unsigned long long trace_override, int64_t orig_rows) {
    hipLaunchKernelGGL(k_sparse_meta, dim3(1), dim3(1024), 0, ctx->stream, keys, counts, (int)D, keys32, meta,
                       trace_override, (u32)orig_rows);
    SP_HIP(hipGetLastError());
    return SP_OK;
}

// Bytes of global memory one workgroup of the lists-in-global form needs (two entry lists + four descriptor arrays).
size_t sparse_list_slab_bytes(int64_t D) {
    const size_t list_bytes = ((size_t)(D + 8) * 4 + 15) & ~(size_t)15;
    const size_t gdesc_bytes = (((size_t)std::max<int64_t>(D, 1024) + 2) * 2 + 15) & ~(size_t)15;
    const size_t w_bytes = ((size_t)std::max<int64_t>(D, 1024) + 8) * SPK_NB * 8;   // the W_GLOBAL variant's block
    return (2 * list_bytes + 4 * gdesc_bytes + w_bytes + 255) & ~(size_t)255;
}

// Bytes of global memory one workgroup of the HBM form needs for a table of D patterns whose bitmaps take `bm_words`
// 64-bit words: the same carve as the LDS form with every size at its a-priori bound (sides <= max(D, 1024) ids).
size_t sparse_slab_bytes(int64_t D, int64_t bm_words, bool wide) {
    const size_t d1 = (size_t)std::max<int64_t>(D, 1024) + 16;
    return (size_t)(D + 8) * 8 + (size_t)D * 6 + (size_t)bm_words * 12 + d1 * (16 + 16 + 2 * (wide ? 96 : 48)) +
           (size_t)D * 2 + 65536;
}

// The whole sparse route for n_al alignments x S splits on the context's stream, no host synchronisation: the in-LDS
// kernel, and queued behind it the slow kernel that finishes - on the device - whatever the first one handed back.
// d_max / bm_words_max: largest table / bitmap size among the items (slab sizing).  wide_all: see k_sparse_slow.
// Afterwards a status word has bit 1 set only for splits left to the dense route (wide_all = false).
// Every item "does not fit the in-LDS form" without running it: status 2, score 0, and the slow kernel's queue head reset.
__global__ void k_sparse_refuse_all(int n_items, double* __restrict__ scores, int* __restrict__ status, int* __restrict__ queue_head) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *queue_head = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
        scores[i] = 0.0;
        status[i] = 2;
    }
}

int launch_sparse_chain(sp_ctx* ctx, const AlDesc* als_dev, const AlDesc& al0, int n_al, int n_taxa,
                        const SplitDev* splits_dev, const SplitDev* launch_dev, int64_t S, double* scores, int* status,
                        int64_t d_max, int64_t bm_words_max, bool wide_all, const SparseFitHint& hint) {
    if (S == 0 || n_al == 0) return SP_OK;
    const int64_t n_items = S * n_al;
    SP_REQUIRE(n_items < ((int64_t)1 << 31), SP_ELIMIT, "sparse route: %lld items in one call (limit 2^31)", (long long)n_items);
    static PerDeviceOnce attr;
    if (attr.need(ctx->device)) {
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sparse_score),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_BYTES));
        SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sparse_slow),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_BYTES));
        attr.done(ctx->device);
    }
    SpkSlow q{};
    q.slab_l = sparse_list_slab_bytes(d_max);
    q.slab_h = (sparse_slab_bytes(d_max, bm_words_max, false) + 255) & ~(size_t)255;
    q.slab_w = (sparse_slab_bytes(d_max, bm_words_max, true) + 255) & ~(size_t)255;
    q.slab_stride = std::max(q.slab_l, std::max(q.slab_h, q.slab_w));
    // Grid of the slow kernel - a matter of speed only, any grid finishes every item.  Tables the in-LDS form cannot
    // hold by size alone (or a test's pretended LDS size): most items WILL come here, so one workgroup per CU draws small
    // chunks (a slow item runs 30 - 500 us: balance).  Otherwise hand-backs are rare (data-dependent fits, gapless
    // spectra) and the common case is an empty pass: a handful of workgroups scan the status words 64 at a time - each
    // workgroup of this kernel has to wait for a CU whose whole LDS is free, which at one per CU costs the pipelined
    // benchmark loop 3.7 % (0.1043 against 0.1006 ms per step).  Huge slabs shrink the grid (8 GB pool at most).
    const bool expect_many = (size_t)(d_max + 8) * 8 + (size_t)d_max * 6 + 12288 > (size_t)SPK_LDS_BYTES || ctx->opt.lds_cap > 0;
    // (one item per draw: a draw is one atomic against 100 us of work, and chunks of 7 left the last workgroups of BASELINE
    // config 5's pass up to 0.7 ms of items while the others had run dry)
    q.chunk = expect_many ? 1 : 64;
    const int64_t nchunks = (n_items + q.chunk - 1) / q.chunk;
    int grid = (int)std::min<int64_t>(ctx->n_cu, nchunks);
    grid = (int)std::max<int64_t>(1, std::min<int64_t>(grid, std::max<int64_t>(16, ((int64_t)8 << 30) / (int64_t)q.slab_stride)));
    SP_CHECK(ctx->slabs.ensure((size_t)grid * q.slab_stride));
    SP_CHECK(ctx->chain.ensure(64));
    q.head = ctx->chain.as<int>();
    q.slabs = ctx->slabs.as<unsigned char>();
    q.wide_all = wide_all ? 1 : 0;
    q.wide_cap = ctx->opt.wide_cap;
    // Can ANY item run in the in-LDS form?  Its first test (spk_score_one, "need_build") is a function of the table size and
    // the split's bitmap words alone, so the host evaluates it for the cheapest split on the smallest table: when even that
    // one does not fit - a 12-taxon 100 k-site table: every split - the launch of one workgroup per item that would only
    // write "status 2" (0.26 ms for the 65 120 items of BASELINE config 5, 1 % of the step) is replaced by a fill kernel.
    bool none_fits = false;
    if (hint.d_min > 0) {
        const size_t off = (sizeof(SpkShared) + 15) & ~(size_t)15, cap = ctx->opt.lds_cap > 0 ? (size_t)ctx->opt.lds_cap : (size_t)SPK_LDS_BYTES;
        const size_t d8 = (size_t)hint.d_min + 8;
        size_t best = ~(size_t)0;
        if (hint.w12_gpath >= 0) best = std::min(best, off + d8 * 8 + (size_t)hint.w12_gpath + 4096 + 256);
        if (hint.w12_general >= 0) best = std::min(best, off + d8 * 12 + (size_t)hint.w12_general + 4096 + 256);
        none_fits = best != ~(size_t)0 && best + 2048 > cap;
    }
    {
        PhaseScope ps(ctx, SP_PHASE_SPARSE);
        if (none_fits)
            hipLaunchKernelGGL(k_sparse_refuse_all, dim3((unsigned)std::min<int64_t>(1024, (n_items + 255) / 256)), dim3(256), 0,
                               ctx->stream, (int)n_items, scores, status, q.head);
        else
            hipLaunchKernelGGL(k_sparse_score, dim3((unsigned)n_items), dim3(SPK_THREADS), SPK_LDS_BYTES, ctx->stream, als_dev,
                               al0, n_al, n_taxa, launch_dev, (int)S, scores, status, q.head, (size_t)ctx->opt.lds_cap);
        SP_HIP(hipGetLastError());
    }
    PhaseScope ps(ctx, SP_PHASE_CHAIN);
    hipLaunchKernelGGL(k_sparse_slow, dim3((unsigned)grid), dim3(SPK_THREADS), SPK_LDS_BYTES, ctx->stream, als_dev, n_al,
                       n_taxa, splits_dev, (int)S, scores, status, q);
    SP_HIP(hipGetLastError());
    return SP_OK;
}


