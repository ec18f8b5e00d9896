// Subflattening path.
//
// Replaces splitp/constructions.py:108-163 (subflattening): an O((3a+1)(3b+1) * D * n) Python loop per
// split.  Every subflattening of an alignment is a sub-block of ONE (3n+1) x (3n+1) signed
// second-moment matrix:
//     w(p)[3t+j] = SIGN[j][digit_t(p)]  (j = A,C,G),  w(p)[3n] = +1
//     M          = sum_p weight(p) * w(p) w(p)^T
//     subflattening(A|B) = M[idx(A) + [3n], idx(B) + [3n]],  idx(S) = [3t+j for t in S for j in 0..2]
// where SIGN is the +-1 table behind the reference's `banned` set (constructions.py:143-147) and the
// label order is that of constructions.py:174-189.  So: one contraction per alignment (k_moments),
// then per split a gather and a tiny symmetric eigenproblem (k_subscore, one wave per split).
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "subflat_common.h"

#define MOM_THREADS 256
#define MOM_CHUNK 256

// sign bits of one pattern: bit (3t+j) set <=> w = -1.  A-row: d in {1,2}; C-row: d in {2,3}; G-row: d odd.
__device__ __forceinline__ void sign_bits(u64 key, int n, u64& lo, u64& hi) {
    u64 l = 0, h = 0;
    for (int t = 0; t < n; ++t) {
        const unsigned d = (unsigned)(key >> (2 * (n - 1 - t))) & 3u;
        const unsigned negC = d >> 1, negG = d & 1u, negA = negC ^ negG;
        const u64 tri = (u64)negA | ((u64)negC << 1) | ((u64)negG << 2);
        const int pos = 3 * t;
        if (pos < 64) {
            l |= tri << pos;
            if (pos > 61) h |= tri >> (64 - pos);
        } else {
            h |= tri << (pos - 64);
        }
    }
    lo = l;
    hi = h;
}

template <typename W, typename ACC>
__global__ __launch_bounds__(MOM_THREADS) void k_moments(const u64* __restrict__ keys, const W* __restrict__ wts,
                                                         int64_t D, int n, ACC* __restrict__ partial) {
    __shared__ u64 s_lo[MOM_CHUNK], s_hi[MOM_CHUNK];
    __shared__ ACC s_w[MOM_CHUNK];
    const int m = 3 * n + 1, mm = m * m;
    ACC* out = partial + (int64_t)blockIdx.x * mm;
    for (int e = threadIdx.x; e < mm; e += MOM_THREADS) out[e] = 0;
    for (int64_t base = (int64_t)blockIdx.x * MOM_CHUNK; base < D; base += (int64_t)gridDim.x * MOM_CHUNK) {
        const int cnt = (int)min((int64_t)MOM_CHUNK, D - base);
        __syncthreads();
        if (threadIdx.x < cnt) {
            u64 lo, hi;
            sign_bits(keys[base + threadIdx.x], n, lo, hi);
            s_lo[threadIdx.x] = lo;
            s_hi[threadIdx.x] = hi;
            s_w[threadIdx.x] = (ACC)wts[base + threadIdx.x];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < mm; e += MOM_THREADS) {
            const int u = e / m, v = e % m;
            if (u > v) continue;  // symmetric: mirrored by the reduce kernel
            const bool uh = u >= 64, vh = v >= 64;
            const int ub = u & 63, vb = v & 63;
            ACC acc = 0;
            for (int p = 0; p < cnt; ++p) {
                const u64 a = uh ? s_hi[p] : s_lo[p];
                const u64 b = vh ? s_hi[p] : s_lo[p];
                const bool neg = ((a >> ub) ^ (b >> vb)) & 1ull;
                acc += neg ? -s_w[p] : s_w[p];
            }
            out[e] += acc;
        }
    }
}

template <typename ACC>
__global__ void k_moments_reduce(const ACC* __restrict__ partial, int parts, int m, ACC* __restrict__ M) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int mm = m * m;
    if (e >= mm) return;
    const int u = e / m, v = e % m;
    const int src = u <= v ? e : v * m + u;
    ACC s = 0;
    for (int g = 0; g < parts; ++g) s += partial[(int64_t)g * mm + src];  // fixed order: reproducible
    M[e] = s;
}

// ---- the same matrix on the fp64 matrix cores (round 4; up to 21 taxa: m <= 64) ------------------------------------------
// M = S^T diag(w) S with S[p][u] = +-1 is a (64 x D) x (D x 64) product.  One wave per chunk of 64 patterns, grid-stride:
// every lane turns ONE pattern into its sign word (sign_bits) and stages {signs, weight} in LDS; 16 steps of
// v_mfma_f64_16x16x4 then take 4 patterns each - lane (fr = lane & 15, fk = lane >> 4) supplies w_p s_p[16 I + fr] as A and
// s_p[16 J + fr] as B operand of tile (I, J), p = 4 step + fk - into the 10 tiles on and above the diagonal, which stay in
// registers over the wave's whole share of the table.  Count tables: every term and every partial sum is an integer below
// 2^53 (at most the number of sites), so the fp64 sums are exact and do not depend on the order; k_moments_reduce64 adds the
// waves' tiles in a fixed order, mirrors them and writes int64 (or double for weighted tables).
// 20 taxa x 1 M sites (259 k patterns): 0.44 + 0.35 ms for k_moments + k_moments_reduce, which spent 8 instructions per
// (pattern, entry) and summed 1024 partial matrices with one thread an entry.
#define MOM64_WAVES 4
template <typename W>
__global__ __launch_bounds__(MOM64_WAVES * 64) void k_moments_mfma(const u64* __restrict__ keys, const W* __restrict__ wts,
                                                                   int64_t D, int n, double* __restrict__ partial) {
    __shared__ u64 s_sig[MOM64_WAVES][64];
    __shared__ double s_wt[MOM64_WAVES][64];
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * MOM64_WAVES + w, nw = (int64_t)gridDim.x * MOM64_WAVES;
    d4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = d4{0, 0, 0, 0};
    for (int64_t base = gw * 64; base < D; base += nw * 64) {
        const int64_t p = base + lane;
        u64 lo = 0, hi = 0;
        double wt = 0.0;   // (patterns past the end: weight 0)
        if (p < D) {
            sign_bits(keys[p], n, lo, hi);
            wt = (double)wts[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the previous chunk's reads are done
        s_sig[w][lane] = lo;
        s_wt[w][lane] = wt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
        for (int step = 0; step < 16; ++step) {
            const u64 sg = s_sig[w][4 * step + fk];
            const double wp = s_wt[w][4 * step + fk];
            double b[4], a[4];
#pragma unroll
            for (int I = 0; I < 4; ++I) {
                b[I] = ((sg >> (16 * I + fr)) & 1ull) ? -1.0 : 1.0;
                a[I] = b[I] * wp;
            }
            int t = 0;
#pragma unroll
            for (int I = 0; I < 4; ++I)
#pragma unroll
                for (int J = I; J < 4; ++J, ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], b[J], acc[t], 0, 0, 0);
        }
    }
    // the wave's tiles: accumulator q of lane (fr, fk) is entry (16 I + fk + 4 q, 16 J + fr)
    double* out = partial + gw * 4096;
    int t = 0;
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = I; J < 4; ++J, ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) out[(16 * I + fk + 4 * q) * 64 + 16 * J + fr] = acc[t][q];
}

// M[u][v] = sum over the waves' tiles, in wave order (fixed: reproducible); tiles below the diagonal are the mirror images.
// 16 entries a workgroup, 16 lanes an entry: every lane adds its 16th of the partial matrices, lane 0 of the entry adds the
// 16 sums in order.
template <typename ACC>
__global__ __launch_bounds__(256) void k_moments_reduce64(const double* __restrict__ partial, int parts, int m,
                                                          ACC* __restrict__ M) {
    __shared__ double s_part[16][17];
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;   // entry of the 64 x 64 layout
    const int u = e >> 6, v = e & 63;
    const int src = (u >> 4) <= (v >> 4) ? e : v * 64 + u;
    double sum = 0.0;
    for (int g = slice; g < parts; g += 16) sum += partial[(int64_t)g * 4096 + src];
    s_part[el][slice] = sum;
    __syncthreads();
    if (slice == 0 && u < m && v < m) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += s_part[el][k];
        M[u * m + v] = (ACC)tot;
    }
}

static int ensure_moments(sp_alignment* al) {
    if (al->moments_ready) return SP_OK;
    sp_ctx* ctx = al->ctx;
    const int m = 3 * al->n_taxa + 1, mm = m * m;
    if (m <= 64 && !ctx->opt.moments_valu) {
        const int dev_cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
        const int64_t chunks = (al->D + 63) / 64;
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(dev_cus, (chunks + MOM64_WAVES - 1) / MOM64_WAVES));
        const int parts = blocks * MOM64_WAVES;
        SP_CHECK(al->moments.ensure((size_t)mm * 8));
        SP_CHECK(ctx->misc.ensure((size_t)parts * 4096 * 8));
        PhaseScope ps(ctx, SP_PHASE_MOMENT);
        if (al->exact) {
            hipLaunchKernelGGL(k_moments_mfma<u32>, dim3(blocks), dim3(MOM64_WAVES * 64), 0, ctx->stream, al->keys.as<u64>(),
                               al->counts.as<u32>(), al->D, al->n_taxa, ctx->misc.as<double>());
            hipLaunchKernelGGL(k_moments_reduce64<long long>, dim3(256), dim3(256), 0, ctx->stream, ctx->misc.as<double>(), parts,
                               m, al->moments.as<long long>());
        } else {
            hipLaunchKernelGGL(k_moments_mfma<double>, dim3(blocks), dim3(MOM64_WAVES * 64), 0, ctx->stream, al->keys.as<u64>(),
                               al->weights.as<double>(), al->D, al->n_taxa, ctx->misc.as<double>());
            hipLaunchKernelGGL(k_moments_reduce64<double>, dim3(256), dim3(256), 0, ctx->stream, ctx->misc.as<double>(), parts, m,
                               al->moments.as<double>());
        }
        SP_HIP(hipGetLastError());
        al->moments_ready = true;
        return SP_OK;
    }
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (al->D + MOM_CHUNK - 1) / MOM_CHUNK));
    SP_CHECK(al->moments.ensure((size_t)mm * 8));
    SP_CHECK(ctx->misc.ensure((size_t)parts * mm * 8));
    PhaseScope ps(ctx, SP_PHASE_MOMENT);
    if (al->exact) {
        hipLaunchKernelGGL((k_moments<u32, long long>), dim3(parts), dim3(MOM_THREADS), 0, ctx->stream,
                           al->keys.as<u64>(), al->counts.as<u32>(), al->D, al->n_taxa, ctx->misc.as<long long>());
        hipLaunchKernelGGL(k_moments_reduce<long long>, dim3((mm + 255) / 256), dim3(256), 0, ctx->stream,
                           ctx->misc.as<long long>(), parts, m, al->moments.as<long long>());
    } else {
        hipLaunchKernelGGL((k_moments<double, double>), dim3(parts), dim3(MOM_THREADS), 0, ctx->stream,
                           al->keys.as<u64>(), al->weights.as<double>(), al->D, al->n_taxa, ctx->misc.as<double>());
        hipLaunchKernelGGL(k_moments_reduce<double>, dim3((mm + 255) / 256), dim3(256), 0, ctx->stream,
                           ctx->misc.as<double>(), parts, m, al->moments.as<double>());
    }
    SP_HIP(hipGetLastError());
    al->moments_ready = true;
    return SP_OK;
}

extern "C" int sp_moment_matrix(sp_alignment* al, int64_t* out_i64, double* out_f64) {
    return sp_guard("sp_moment_matrix", [&]() -> int {
    SP_REQUIRE(al, SP_EINVAL, "alignment is NULL");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    SP_REQUIRE(al->D > 0, SP_EINVAL, "empty pattern table");
    SP_CHECK(ensure_moments(al));
    const int m = 3 * al->n_taxa + 1;
    SP_HIP(hipStreamSynchronize(ctx->stream));
    if (al->exact) {
        SP_REQUIRE(out_i64, SP_EINVAL, "exact alignment: pass out_i64");
        SP_HIP(hipMemcpy(out_i64, al->moments.p, (size_t)m * m * 8, hipMemcpyDeviceToHost));
    } else {
        SP_REQUIRE(out_f64, SP_EINVAL, "weighted alignment: pass out_f64");
        SP_HIP(hipMemcpy(out_f64, al->moments.p, (size_t)m * m * 8, hipMemcpyDeviceToHost));
    }
    return SP_OK;
    });
}


template <bool EXACT>
__global__ void k_subflatten_gather(const void* __restrict__ Mv, int n, double N, const SplitDev* __restrict__ sp_,
                                    double* __restrict__ out) {
    const SplitDev& sp = sp_[0];
    const int m = 3 * n + 1;
    const int R = 3 * sp.nr + 1, C = 3 * sp.nc + 1;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * C) return;
    const int i = e / C, j = e % C;
    const int u = sub_index(sp.taxa, sp.nr, n, i), v = sub_index(sp.taxa + sp.nr, sp.nc, n, j);
    if (EXACT)  // exact integer signed sum, one rounding in the division (the reference accumulates count/N terms)
        out[e] = (double)reinterpret_cast<const long long*>(Mv)[u * m + v] / N;
    else
        out[e] = reinterpret_cast<const double*>(Mv)[u * m + v];
}

extern "C" int sp_subflatten(sp_alignment* al, const int32_t* oa, int a, const int32_t* ob, int b, double* out_host) {
    return sp_guard("sp_subflatten", [&]() -> int {
    SP_REQUIRE(al && oa && ob && out_host, SP_EINVAL, "NULL argument");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    const int n = al->n_taxa;
    SP_REQUIRE(a >= 1 && b >= 1 && a + b == n, SP_EINVAL,
               "subflattening needs a split covering all %d taxa (got %d + %d); the reference raises KeyError "
               "(constructions.py:198)", n, a, b);
    unsigned seen = 0;
    for (int i = 0; i < n; ++i) {
        const int t = i < a ? oa[i] : ob[i - a];
        SP_REQUIRE(t >= 0 && t < n && !(seen & (1u << t)), SP_EINVAL, "bad taxon index %d in split", t);
        seen |= 1u << t;
    }
    SP_REQUIRE(al->D > 0, SP_EINVAL, "empty pattern table");
    SP_CHECK(ensure_moments(al));
    SplitDev sd;
    memset(&sd, 0, sizeof(sd));
    sd.nr = a;
    sd.nc = b;
    for (int i = 0; i < a; ++i) sd.taxa[i] = (int8_t)oa[i];
    for (int i = 0; i < b; ++i) sd.taxa[a + i] = (int8_t)ob[i];
    const int R = 3 * a + 1, C = 3 * b + 1;
    SP_CHECK(ctx->splits.ensure(sizeof(SplitDev)));
    SP_CHECK(ctx->misc2.ensure((size_t)R * C * 8));
    if (ctx->cache) ctx->cache->valid = false;
    SP_HIP(hipMemcpyAsync(ctx->splits.p, &sd, sizeof(sd), hipMemcpyHostToDevice, ctx->stream));
    if (al->exact)
        hipLaunchKernelGGL(k_subflatten_gather<true>, dim3((R * C + 255) / 256), dim3(256), 0, ctx->stream,
                           al->moments.p, n, (double)al->N, ctx->splits.as<SplitDev>(), ctx->misc2.as<double>());
    else
        hipLaunchKernelGGL(k_subflatten_gather<false>, dim3((R * C + 255) / 256), dim3(256), 0, ctx->stream,
                           al->moments.p, n, 1.0, ctx->splits.as<SplitDev>(), ctx->misc2.as<double>());
    SP_HIP(hipGetLastError());
    SP_HIP(hipMemcpyAsync(out_host, ctx->misc2.p, (size_t)R * C * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    return SP_OK;
    });
}

// ---- batched score of subflattenings: one wave per split ------------------------------------------
// The block is at most 49 x 91; its Gram over the smaller side (r <= 49) is formed in LDS and
// diagonalised by parallel-order cyclic Jacobi; score = sqrt(max(0, 1 - top4 / trace)).
#define SUB_WAVES 4


template <bool EXACT>
__global__ __launch_bounds__(SUB_WAVES * 64) void k_subscore(const void* __restrict__ Mv, int n,
                                                              const int8_t* __restrict__ split_taxa,
                                                              const int* __restrict__ split_a, int64_t S, int rmax,
                                                              double* __restrict__ scores, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t sid = (int64_t)blockIdx.x * SUB_WAVES + w;
    if (sid >= S) return;
    const int P = rmax + 1;  // rmax is even; odd pitch
    double* G = sm + (size_t)w * rmax * P;
    const int m = 3 * n + 1;
    const int8_t* taxa = split_taxa + sid * n;
    const int a = split_a[sid], b = n - a;
    // orient: rows = smaller side
    const bool swap = a > b;
    const int8_t* rt = swap ? taxa + a : taxa;
    const int8_t* ct = swap ? taxa : taxa + a;
    const int nr = swap ? b : a, nc = swap ? a : b;
    const int r = 3 * nr + 1, c = 3 * nc + 1;
    const int re = (r + 1) & ~1;  // even size for the round-robin pairing
    auto Mval = [&](int u, int v) -> double {
        return EXACT ? (double)reinterpret_cast<const long long*>(Mv)[u * m + v]
                     : reinterpret_cast<const double*>(Mv)[u * m + v];
    };
    // Gram over the rows
    for (int e = lane; e < re * re; e += 64) {
        const int i = e / re, j = e % re;
        double s = 0;
        if (i < r && j < r && i <= j) {
            const int ui = sub_index(rt, nr, n, i), uj = sub_index(rt, nr, n, j);
            for (int k = 0; k < c; ++k) {
                const int v = sub_index(ct, nc, n, k);
                s += Mval(ui, v) * Mval(uj, v);
            }
        }
        G[i * P + j] = s;
    }
    wave_sync_lds2();
    for (int e = lane; e < re * re; e += 64) {
        const int i = e / re, j = e % re;
        if (i > j) G[i * P + j] = G[j * P + i];
    }
    wave_sync_lds2();
    double tr = 0;
    for (int i = lane; i < r; i += 64) tr += G[i * P + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);
    if (r <= 4 || !(tr > 0)) {
        if (lane == 0) {
            scores[sid] = (tr > 0) ? 0.0 : __builtin_nan("");
            status[sid] = 0;
        }
        return;
    }
    const int half = re / 2;
    int sweep = 0;
    double prev_off = 1e300;
    for (; sweep < 30; ++sweep) {
        double off = 0, dg = 0;
        for (int e = lane; e < re * re; e += 64) {
            const int i = e / re, j = e % re;
            const double v = G[i * P + j];
            if (i == j) dg += v * v; else off += v * v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            off += __shfl_xor(off, d, 64);
            dg += __shfl_xor(dg, d, 64);
        }
        // converged: the off-diagonal mass is below 1e-15 of the diagonal's - or it has reached its rounding floor: the
        // entries of a 22 x 22 ... 49 x 49 block carry ~eps |G| each, their squares sum to more than 1e-30 of the diagonal,
        // and the sweeps ran to the cap of 30 with a spurious "not converged" flag on every split of such a table
        // (randomised sweep of round 2, forced onto this kernel at 14 - 20 taxa; the values were right to 1e-12).  Jacobi
        // converges quadratically, so a sweep that no longer shrinks a small off-diagonal mass tenfold has hit that floor.
        if (!(off > 1e-30 * dg) || (off <= 1e-24 * dg && off > 0.1 * prev_off)) break;
        prev_off = off;
        for (int round = 0; round < re - 1; ++round) {
            // rotation parameters for all pairs first (they read the pre-round matrix)
            // pair pi: (re-1, round) for pi = 0, else ((round+pi) % (re-1), (round-pi) mod (re-1))
            // column phase: items (pair, row)
            // every lane recomputes (c, s) for the pair it is serving; G[p][p], G[q][q], G[p][q] of a
            // pair are only modified by that pair's own rotation, and the column phase only touches
            // them through that rotation, so (c, s) must be computed before any write: stage in LDS.
            double* cs = sm + (size_t)SUB_WAVES * rmax * P + (size_t)w * rmax;  // [half][2]
            for (int pi = lane; pi < half; pi += 64) {
                int p, q;
                if (pi == 0) { p = re - 1; q = round; }
                else { p = (round + pi) % (re - 1); q = (round + (re - 1) - pi) % (re - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double app = G[p * P + p], aqq = G[q * P + q], apq = G[p * P + q];
                double cc = 1.0, ss = 0.0;
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-20 * sqrt(fabs(app * aqq))) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    cc = 1.0 / sqrt(1.0 + t * t);
                    ss = t * cc;
                }
                cs[2 * pi] = cc;
                cs[2 * pi + 1] = ss;
            }
            wave_sync_lds2();
            for (int it = lane; it < half * re; it += 64) {
                const int pi = it / re, i = it % re;
                int p, q;
                if (pi == 0) { p = re - 1; q = round; }
                else { p = (round + pi) % (re - 1); q = (round + (re - 1) - pi) % (re - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double cc = cs[2 * pi], ss = cs[2 * pi + 1];
                const double hp = G[i * P + p], hq = G[i * P + q];
                G[i * P + p] = cc * hp - ss * hq;
                G[i * P + q] = ss * hp + cc * hq;
            }
            wave_sync_lds2();
            for (int it = lane; it < half * re; it += 64) {
                const int pi = it / re, j = it % re;
                int p, q;
                if (pi == 0) { p = re - 1; q = round; }
                else { p = (round + pi) % (re - 1); q = (round + (re - 1) - pi) % (re - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double cc = cs[2 * pi], ss = cs[2 * pi + 1];
                const double hp = G[p * P + j], hq = G[q * P + j];
                G[p * P + j] = cc * hp - ss * hq;
                G[q * P + j] = ss * hp + cc * hq;
            }
            wave_sync_lds2();
        }
    }
    // four largest diagonal entries
    double top = 0;
    {
        double best[4] = {-1e300, -1e300, -1e300, -1e300};
        for (int i = 0; i < r; ++i) {  // every lane scans (uniform); r <= 49
            double v = G[i * P + i];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v > best[k]) { const double t = best[k]; best[k] = v; v = t; }
        }
        top = best[0] + best[1] + best[2] + best[3];
    }
    if (lane == 0) {
        const double op = 1.0 - top / tr;
        scores[sid] = sqrt(op > 0 ? op : 0.0);
        status[sid] = (sweep >= 30 ? 1 : 0) | (sweep << 8);
    }
}


// ---- batched score, fast form (n <= 20: moment matrix <= 61 x 61, smaller side <= 31 rows) ---------------------------
// Same Gram matrix as k_subscore, but the moment matrix is staged once per workgroup in LDS (the Jacobi kernel re-reads
// it from L2 for every product term), the Gram matrix is formed by fp64 MFMA, every wave walks splits in a grid-stride
// loop, and the eigenvalues come from the classical dense symmetric route instead of cyclic Jacobi: Householder
// tridiagonalisation (two lanes per row; ~3 r^2 LDS operations a lane instead of ~14 r^2 for 7 Jacobi sweeps) followed by
// multisection on the Sturm count of the tridiagonal matrix - 4 groups of 16 lanes, one group per wanted eigenvalue, 16
// shifts a pass, 13 passes (17^13 = 9.9e15 > 2^53: the bracket ends below eps * span; a 14th pass only moved noise).  Both steps are backward stable: eigenvalues good to a few eps * lambda_1,
// which is what 1 - top4 / trace needs.  No iteration that could fail to converge.
#define SUBT_MAXWAVES 16  // waves of a workgroup: chosen per launch so that the CU holds as many waves as its LDS allows
#ifndef SUBT_PASSES
#define SUBT_PASSES 13
#endif

// Sums over lanes of a wave, returned to all of them (subt_row0_sum, subt_half_sum below): two quad steps and two row shifts
// on the DPP path (no LDS crossbar), a row_bcast step where two DPP rows are summed, the total through scalar registers.
// Fixed order.
// Per-wave LDS: de (32 pairs {d_i, e2_(i-1)}: diagonal and squared sub-diagonal of the tridiagonal matrix, one 16-byte
// read per step of the Sturm recurrence), G (rmax rows of pitch P = rmax | 1 doubles: an odd pitch keeps a column walk off
// one bank), v, w (32 doubles each), urow (32 x u16: row offset u * m into the staged matrix, < 61 * 61) and vcol
// (64 x u8: column < 61).  Sized for the longest side of the batch, not for 32 rows: 16 taxa (rmax 25) take 6.2 KB a wave
// instead of 9.6.
__host__ __device__ __forceinline__ size_t subt_wave_bytes(int rmax) {
    return ((size_t)rmax * (rmax | 1) * 8 + 4 * 32 * 8 + 32 * 2 + 64 + 15) & ~(size_t)15;
}
// The same recurrence with the table read through the SCALAR cache: every lane of the wave needs the same pair {d_i,
// e2_(i-1)} at step i, and as a broadcast LDS read that is 1 KB of LDS bandwidth a step - with 13 passes a third of the
// kernel's LDS time, the unit that bounds it (profiles/r03_pmc_binding_config4_auto.json: LDS 77 % busy, vector issue 58 %).
// The scaled table is therefore also written to a per-wave slot in global memory (SUBT_SLOT doubles, k_subscore_tri) and
// fetched here with s_load_dwordx8 - two steps a load, straight into scalar registers, which the vector instructions
// take as operands: no LDS, no vector instruction.  The compiler knows nothing of these loads, so each is waited for
// (s_waitcnt lgkmcnt(0), tied to the registers by the "+s" operand) on every path before its registers can be reused.
// cur = {d_I, e2_(I-1), d_(I+1), e2_I}, I odd.
#define SUBT_SLOT 80   // doubles per wave: 32 pairs + the pairs a load past the end touches
typedef double subt_d4 __attribute__((ext_vector_type(4)));
template <int I>
__device__ __forceinline__ void subt_minor_steps_s(int r, const double* gde, double sigma, double pp, double pc, subt_d4 cur,
                                                   unsigned& mask) {
    if constexpr (I < 31) {
        if (I < r) {
            subt_d4 nxt;
            asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(nxt) : "s"(gde), "n"((I + 2) * 16));
            __builtin_amdgcn_sched_barrier(0);   // (the request goes out before this pair of steps, not behind the first)
            double pn = fma(cur.x - sigma, pc, -(cur.y * pp));
            mask = __builtin_amdgcn_alignbit(mask, (unsigned)__double2hiint(pn), 31);   // (mask << 1) | sign
            if ((I & 7) == 0) {
                const int ex = __builtin_amdgcn_frexp_exp(subt_max_abs(pn, pc));
                pn = ldexp(pn, -ex);
                pc = ldexp(pc, -ex);
            }
            double pq = pn;
            if (I + 1 < r) {
                pq = fma(cur.z - sigma, pn, -(cur.w * pc));
                mask = __builtin_amdgcn_alignbit(mask, (unsigned)__double2hiint(pq), 31);
                if (((I + 1) & 7) == 0) {
                    const int ex = __builtin_amdgcn_frexp_exp(subt_max_abs(pq, pn));
                    pq = ldexp(pq, -ex);
                    pn = ldexp(pn, -ex);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nxt));
            if (I + 1 < r) subt_minor_steps_s<I + 2>(r, gde, sigma, pn, pq, nxt, mask);
        }
    }
}
// sum over lanes 0 .. 15 (whatever the others hold)
__device__ __forceinline__ double subt_row0_sum(double x) {
    x += subt_dpp<0xB1>(x);
    x += subt_dpp<0x4E>(x);
    x += subt_dpp<0x114>(x);
    x += subt_dpp<0x118>(x);
    return subt_readlane(x, 15);
}
// sum over lanes 0 .. 31 (whatever the others hold): DPP rows 0 and 1 only
__device__ __forceinline__ double subt_half_sum(double x) {
    x += subt_dpp<0xB1>(x);
    x += subt_dpp<0x4E>(x);
    x += subt_dpp<0x114>(x);
    x += subt_dpp<0x118>(x);
    x += subt_dpp<0x142, 0xA>(x);
    return subt_readlane(x, 31);
}

// M32: count table with fewer than 2^31 sites - every moment fits an int32, the staged matrix takes half the LDS and a
// third workgroup fits the CU (3 waves per SIMD instead of 2: the eigenvalue steps are dependency chains).
#ifdef SUBT_STAMPS
// diagnostic build (tools/gpu_subflat_stamps.sh): s_memtime of wave 0 of workgroup 0 at the phase boundaries of its last split
// (the splits come in ascending size classes: one of the longest)
__device__ long long g_subt_stamps[16];
extern "C" int sp_debug_subt_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_subt_stamps), sizeof(long long) * 16) == hipSuccess ? 0 : 2;
}
#define TSTAMP(i) do { if (blockIdx.x == 0 && w == 0 && lane == 0) g_subt_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i)
#endif

template <bool EXACT, bool M32>
__global__ __launch_bounds__(SUBT_MAXWAVES * 64) void k_subscore_tri(const void* __restrict__ Mv, int n, int rmax,
                                                                     const int8_t* __restrict__ split_taxa,
                                                                     const int* __restrict__ split_a, int64_t S,
                                                                     double* __restrict__ scores, int* __restrict__ status,
                                                                     double* __restrict__ slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    const int m = 3 * n + 1;
    typedef typename std::conditional<M32, int, double>::type MsT;
    MsT* Ms = reinterpret_cast<MsT*>(smem_t);
    // (the wave index through readfirstlane: the compiler then knows that the split, its side lengths and every loop
    // bound below are wave-uniform - scalar registers and scalar branches instead of 64 identical vector lanes)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int P = rmax | 1, nthreads = blockDim.x;
    unsigned char* wbase = smem_t + (((size_t)m * m * sizeof(MsT) + 15) & ~(size_t)15) + (size_t)w * subt_wave_bytes(rmax);
    double* const sde = reinterpret_cast<double*>(wbase);   // [2 i] = d_i, [2 i + 1] = e2_(i-1)
    double* const G = sde + 64;
    double* const sv = G + rmax * P;                         // (behind G: a read past G's last row meets finite numbers)
    double* const swv = sv + 32;
    unsigned short* const urow = reinterpret_cast<unsigned short*>(swv + 32);
    unsigned char* const vcol = reinterpret_cast<unsigned char*>(urow + 32);
    for (int e = threadIdx.x; e < m * m; e += nthreads)
        Ms[e] = EXACT ? (MsT) reinterpret_cast<const long long*>(Mv)[e] : (MsT) reinterpret_cast<const double*>(Mv)[e];
    __syncthreads();
    const int wpb = nthreads >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t sid = (int64_t)blockIdx.x * wpb + w; sid < S; sid += nwaves) {
        TSTAMP(0);
        const int8_t* taxa = split_taxa + sid * n;
        const int a = split_a[sid], b = n - a;
        const bool swap = a > b;   // rows = smaller side
        const int8_t* rt = swap ? taxa + a : taxa;
        const int8_t* ct = swap ? taxa : taxa + a;
        const int nr = swap ? b : a, nc = swap ? a : b;
        const int r = 3 * nr + 1, c = 3 * nc + 1;
        wave_sync_lds2();   // the previous split's reads of the tables are done
        if (lane < r) urow[lane] = (unsigned short)(sub_index(rt, nr, n, lane) * m);
        if (lane < c) vcol[lane] = (unsigned char)sub_index(ct, nc, n, lane);
        wave_sync_lds2();
        TSTAMP(1);
        // Gram over the rows on the matrix cores: G = B B^T, B[i][k] = M[urow_i + vcol_k] gathered straight from the
        // staged moment matrix.  v_mfma_f64_16x16x4: lane (fr = lane & 15, fk = lane >> 4) supplies B[16 I + fr][4 s + fk]
        // as A and B[16 J + fr][4 s + fk] as B operand of tile (I, J); accumulator q holds G[16 I + fk + 4 q][16 J + fr].
        // (count tables: every term and sum is an integer below 2^53, so the result does not depend on the order)
        {
            typedef double d4 __attribute__((ext_vector_type(4)));
            const int fr = lane & 15, fk = lane >> 4;
            const bool two = r > 16;
            const int u0 = fr < r ? (int)urow[fr] : -1, u1 = (two && 16 + fr < r) ? (int)urow[16 + fr] : -1;
            d4 g00 = {0, 0, 0, 0}, g01 = {0, 0, 0, 0}, g11 = {0, 0, 0, 0};
            for (int k0 = 0; k0 < c; k0 += 4) {
                const int k = k0 + fk;
                const int v = k < c ? (int)vcol[k] : -1;
                const double x0 = (v >= 0 && u0 >= 0) ? (double)Ms[u0 + v] : 0.0;
                g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, g00, 0, 0, 0);
                if (two) {
                    const double x1 = (v >= 0 && u1 >= 0) ? (double)Ms[u1 + v] : 0.0;
                    g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, g01, 0, 0, 0);
                    g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, g11, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = fk + 4 * q;   // (G has r <= rmax rows: the padding of the 16 x 16 tiles is not stored)
                if (i < r && fr < r) G[i * P + fr] = g00[q];
                if (two) {
                    if (16 + fr < r) {
                        G[i * P + 16 + fr] = g01[q];
                        G[(16 + fr) * P + i] = g01[q];
                        if (16 + i < r) G[(16 + i) * P + 16 + fr] = g11[q];
                    }
                }
            }
        }
        // The row loops of the tridiagonalisation run whole trips of 2 or 4 entries and may read up to 3 entries past the end
        // of a row, times a zero of v: those entries must be finite.  Past the row lies its padding (zeroed here), the head
        // of the next row (data), the row behind the last one (zeroed here) or, behind G, the vector v.
        if (lane <= r && lane < rmax) {
#pragma unroll
            for (int cpad = 0; cpad < 3; ++cpad) {
                const int col = lane < r ? r + cpad : cpad;
                if (col < P) G[lane * P + col] = 0.0;
            }
        }
        wave_sync_lds2();
        double tr = 0;
        for (int i = lane; i < r; i += 64) tr += G[i * P + i];
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) tr += __shfl_xor(tr, dd, 64);
        if (r <= 4 || !(tr > 0)) {
            if (lane == 0) {
                scores[sid] = (tr > 0) ? 0.0 : __builtin_nan("");
                status[sid] = 0;
            }
            continue;
        }
        TSTAMP(2);
        // ---- Householder tridiagonalisation: d (diagonal), e2 (squared off-diagonal) -----------------------------------
        // step k annihilates column k below the sub-diagonal with H = I - beta v v^T on the trailing block A (L x L):
        // A <- H A H = A - v w^T - w v^T,  p = beta A v,  w = p - (beta v^T p / 2) v.   Lane pair (2i, 2i+1) owns row i.
        // (lane * P and the row offsets of the two row layouts once a split: a 32-bit vector multiply takes four issue slots,
        // and the compiler redid three of them every step)
        const int laneP = lane * P, row2P = (lane & 31) * P + (lane >> 5), row4P = (lane & 15) * P + (lane >> 4);
        for (int k = 0; k < r - 2; ++k) {
            const int L = r - k - 1, o = k + 1;
            const double x = (lane < L) ? G[laneP + (o * P + k)] : 0.0;
            const double sig = subt_half_sum(x * x);          // (L <= 30: lanes 32.. hold 0)
            const double x0 = subt_readlane(x, 0);
            if (lane == 0) sde[2 * k] = G[k * P + k];
            // (rest of the column negligible against its head: nothing to annihilate; alpha and v below have no
            // cancellation, so the tail's own norm is not needed)
            if (!(sig - x0 * x0 > 0)) {   // already tridiagonal in this column
                if (lane == 0) sde[2 * k + 3] = x0 * x0;
                continue;
            }
            // sqrt and reciprocal by the hardware seeds + Newton steps (an IEEE fp64 sqrt and division are ~25 and ~30
            // instructions, and this kernel is VALU-issue bound: 75 % of the issue slots, profiles/r03a_pmc_binding_config4.json)
            // (v_rsq_f64 is good to 2^-24: one Newton step gives 2^-47, the correction of sq = sig * ry squares that again)
            double ry = __builtin_amdgcn_rsq(sig);
            ry = ry * fma(-0.5 * sig * ry, ry, 1.5);
            double sq = sig * ry;
            sq = fma(0.5 * ry, fma(-sq, sq, sig), sq);        // sqrt(sig) to the last bit or two
            const double alpha = __builtin_copysign(sq, -x0);  // -sign(x0) sqrt(sig)   (x0 = -0.0 counts as negative: either sign is a valid reflector)
            const double den = sig - alpha * x0;
            double beta = __builtin_amdgcn_rcp(den);          // 2 / (v^T v)
            beta = fma(fma(-den, beta, 1.0), beta, beta);
            beta = fma(fma(-den, beta, 1.0), beta, beta);
            const double vi = (lane == 0) ? x0 - alpha : x;   // v (lane < L)
            if (lane == 0) sde[2 * k + 3] = sig;              // alpha^2
            if (lane < 32) sv[lane] = vi;                     // (x = 0 from lane L on, and L >= 2)
            wave_sync_lds2();
            // rows of the trailing block: 2 lanes a row, 4 once the block has <= 16 rows (half the trips of both loops).
            // Both loops are unrolled over wave-uniform trips (constant LDS offsets, one scalar test a trip): the product
            // runs whole trips - v_j = w_j = 0 for L <= j < 32 -, the update masks its last partial trip (an entry past the
            // row can be a live entry of the next one).
            // Lanes of a row are 32 (16) lanes apart - row = lane & 31, part = lane >> 5 -, not neighbours: the 16 lanes one
            // LDS cycle serves then read 16 different rows, whose odd pitch puts them on 16 different bank pairs.  With
            // neighbouring lanes on one row, lane (row, 1) met lane (row - 1, 0) on the same bank pair in every access
            // (29 % of the LDS cycles were conflict replays, profiles/r03_pmc_binding_config4_auto.json before this).
            auto rows = [&](auto shc) {
                constexpr int SH = decltype(shc)::value, LPR = 1 << SH, TRIPS = SH == 2 ? 4 : 15, NROW = 64 >> SH;
                const int row = lane & (NROW - 1), par = lane >> (6 - SH);
                const bool act = row < L;
                double* const g = G + (SH == 2 ? row4P : row2P) + (o * P + o);   // this lane's first entry of its row
                const double* const vp = sv + par;
                const double* const wp = swv + par;
                double p = 0, p1 = 0;   // (two chains: the sum is a dependent sequence of up to 15 fused multiply-adds otherwise)
                if (act) {
#pragma unroll
                    for (int tq = 0; tq < TRIPS; ++tq) {
                        if (tq * LPR >= L) break;
                        if (tq & 1) p1 = fma(g[tq * LPR], vp[tq * LPR], p1);
                        else p = fma(g[tq * LPR], vp[tq * LPR], p);
                    }
                }
                p += p1;
                if (SH == 2) p += __shfl_xor(p, 16, 64);   // the partner lanes of the row
                p += __shfl_xor(p, 32, 64);
                p *= beta;
                const double vr = act ? sv[row] : 0.0;
                // (lanes 0 .. NROW - 1 hold one row's term each, the others copies; vr = 0 off the block)
                const double kk = SH == 2 ? subt_row0_sum(vr * p) : subt_half_sum(vr * p);
                const double wr = p - 0.5 * beta * kk * vr;
                if (lane < NROW) swv[row] = wr;               // (p = vr = 0 off the block: wr = 0 there)
                wave_sync_lds2();
                if (act) {
#pragma unroll
                    for (int tq = 0; tq < TRIPS; ++tq) {
                        if (tq * LPR >= L) break;
                        if ((tq + 1) * LPR <= L || par < L - tq * LPR)
                            g[tq * LPR] = fma(-vr, wp[tq * LPR], fma(-wr, vp[tq * LPR], g[tq * LPR]));
                    }
                }
                wave_sync_lds2();
            };
            if (L <= 16) rows(std::integral_constant<int, 2>());
            else rows(std::integral_constant<int, 1>());
        }
        if (lane == 0) {
            const double eo = G[(r - 1) * P + (r - 2)];
            sde[2 * (r - 2)] = G[(r - 2) * P + (r - 2)];
            sde[2 * (r - 1)] = G[(r - 1) * P + (r - 1)];
            sde[2 * (r - 1) + 1] = eo * eo;
        }
        wave_sync_lds2();
        TSTAMP(3);
        // ---- four largest eigenvalues of the tridiagonal matrix by multisection on the Sturm count ---------------------
        // The count of eigenvalues below a shift is the number of sign changes in the sequence of leading principal minors
        // P_0 = 1, P_1 = d_0 - s, P_(i+1) = (d_i - s) P_i - e2_(i-1) P_(i-1): three full-rate fp64 instructions and one
        // integer instruction a step (the sign bits are shifted into a mask; r <= 31 of them).  The usual ratio form
        // q_i = P_(i+1) / P_i needs a reciprocal a step - v_rcp_f64 issues at a quarter of the rate, plus its Newton step
        // and the small-pivot guard: 12 issue slots a step against 4.6 - and exists because the minors over- and underflow.
        // Here the matrix is moved to units in which the Gershgorin interval has length in [0.5, 1) (exact: a power of two),
        // so |d_i - s| <= 1, e2 <= 1 and a minor grows at most 2x a step; squared off-diagonals below 2^-120 are raised to
        // that (an off-diagonal of 2^-60 of the interval: eigenvalues move by less than 2^-59 of it), so no minor of a pair
        // is less than 2^-120 of the pair two steps before, an exact zero is followed by a non-zero of the opposite sign
        // to its predecessor (one sign change whichever sign the zero is given), and rescaling the pair by the exponent of
        // its larger member every 8th step keeps it inside 2^+-500.
        double gl = 1e300, gu = -1e300;
        if (lane < r) {
            const double e2l = lane > 0 ? sde[2 * lane + 1] : 0.0, e2r = lane < r - 1 ? sde[2 * lane + 3] : 0.0;
            const double el = sqrt(e2l), er = sqrt(e2r);
            gl = sde[2 * lane] - el - er;
            gu = sde[2 * lane] + el + er;
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            gl = fmin(gl, __shfl_xor(gl, dd, 64));
            gu = fmax(gu, __shfl_xor(gu, dd, 64));
        }
        // (tr > 0, so gu > 0; an interval of length 0 - a multiple of the identity - is scaled by its position instead)
        const int E = __builtin_amdgcn_frexp_exp(fmax(gu - gl, fmax(fabs(gl), fabs(gu)) * 0x1p-40));
        wave_sync_lds2();   // every lane has read its neighbours' entries
        double* const gde = slots + ((size_t)blockIdx.x * wpb + w) * SUBT_SLOT;   // this wave's slot (subt_minor_steps_s)
        if (lane < r) {
            const double ds = ldexp(sde[2 * lane], -E);
            const double es = lane > 0 ? fmax(ldexp(sde[2 * lane + 1], -2 * E), 0x1p-120) : 0.0;
            sde[2 * lane] = ds;
            subt_d2 pr;
            pr.x = ds;
            pr.y = es;
            reinterpret_cast<subt_d2*>(gde)[lane] = pr;
        }
        // stores acknowledged by L2, then this CU's scalar cache dropped: it may hold the slot's previous contents
        asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        wave_sync_lds2();
        double lo = ldexp(gl, -E) - 0x1p-44, hi = ldexp(gu, -E) + 0x1p-44;   // (per group of 16 lanes)
        const int grp = lane >> 4, t = lane & 15;
        const int want = r - 1 - grp;                     // ascending index of this group's eigenvalue
        const unsigned rmask = (1u << r) - 1u;            // r <= 31
        const double d0 = subt_readlane(sde[0], 0);
        const double tr_s = subt_readlane(ldexp(tr, -E), 0);   // the trace in the scaled units
        int passes = SUBT_PASSES;
        for (int pass = 0; pass < SUBT_PASSES; ++pass) {
            subt_d4 cur;
            asm volatile("s_load_dwordx8 %0, %1, 16" : "=s"(cur) : "s"(gde));
            const double sigma = lo + (hi - lo) * ((double)(t + 1) * (1.0 / 17.0));
            double pp = 1.0, pc = d0 - sigma;
            unsigned mask = (unsigned)__double2hiint(pc) >> 31;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cur));
            subt_minor_steps_s<1>(r, gde, sigma, pp, pc, cur, mask);
            // bit j of mask = sign of P_(r-j), bit r = 0 = sign of P_0: sign changes = eigenvalues below sigma
            const int cnt = __popc((mask ^ (mask >> 1)) & rmask);
            // cnt = eigenvalues below sigma; the wanted one is below sigma iff cnt > want
            const unsigned long long above = __ballot(cnt > want);
            const unsigned int mine = (unsigned int)((above >> (16 * grp)) & 0xFFFFull);
            const int first = mine ? __builtin_ctz(mine) : 16;   // first shift of the group that is above the eigenvalue
            const double s_hi = __shfl(sigma, 16 * grp + (first < 16 ? first : 15), 64);
            const double s_lo = __shfl(sigma, 16 * grp + (first > 0 ? first - 1 : 0), 64);
            const double nlo = first > 0 ? s_lo : lo, nhi = first < 16 ? s_hi : hi;
            lo = nlo;
            hi = nhi;
            // After 11 passes each of the four brackets is 3e-14 of the (unit-length) Gershgorin interval wide, so the sum of
            // their midpoints is good to 4 x 1.5e-14 and 1 - top4 / trace to 6e-14 / tr_s, the trace in the scaled units
            // (<= 1 per row, ~0.1 at worst on count tables); the score sqrt(.) then moves by at most that over 2 score.
            // Where the score is >= 0.05 (1 - top4 / trace >= 2.5e-3: every split that is not nearly tree-like) that is
            // <= 6e-12 in the worst case and ~1e-12 typically - inside the 1e-10 parity bar (SCORE_TOL), and the reason
            // the self-consistency bar between two kernel shapes (PROP_TOL 5e-12) is asserted on builds that skip alike -
            // so the last two passes are skipped there.
            if (pass == SUBT_PASSES - 3) {
                double tq = (t == 0) ? fmax(0.5 * (lo + hi), 0.0) : 0.0;
                tq += __shfl_xor(tq, 16, 64);
                tq += __shfl_xor(tq, 32, 64);
                if (1.0 - subt_readlane(tq, 0) / tr_s >= 2.5e-3) {
                    passes = pass + 1;
                    break;
                }
            }
        }
        TSTAMP(4);
        const double lam = 0.5 * (lo + hi);
        double top = (t == 0) ? fmax(lam, 0.0) : 0.0;
        top += __shfl_xor(top, 16, 64);
        top += __shfl_xor(top, 32, 64);
        if (lane == 0) {
            const double op = 1.0 - top / tr_s;             // (top in the scaled units)
            scores[sid] = sqrt(op > 0 ? op : 0.0);
            status[sid] = passes << 8;
        }
        TSTAMP(5);
    }
}

// Scores of S splits whose taxon lists (int8, split-major) and first-side sizes already sit on the device.
// scores_out / status_out: device buffers of the caller (NULL = the context's own, ctx->scores / ctx->status).
// pc / order (optional): the batch's size classes and, for a list that is not sorted by class, the positions' split indices
// - what the two-splits-a-wave kernel (subflat_pair.hip) needs; without them the one-split-a-wave kernel runs.
static int launch_subscore(sp_alignment* al, const int8_t* dtaxa, const int* da, int64_t S, int kmax,
                           double* scores_out = nullptr, int* status_out = nullptr, const PairClasses* pc = nullptr,
                           const int* order = nullptr);

int run_subflat_route(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t S) {
    sp_ctx* ctx = al->ctx;
    const int n = al->n_taxa;
    std::vector<int8_t> taxa8((size_t)S * n);
    int kmax = 0;
    for (int64_t s = 0; s < S; ++s) {
        const int a = split_a[s];
        SP_REQUIRE(a >= 1 && a < n, SP_EINVAL, "split %lld: side sizes %d | %d", (long long)s, a, n - a);
        unsigned seen = 0;
        for (int i = 0; i < n; ++i) {
            const int t = split_taxa[s * n + i];
            SP_REQUIRE(t >= 0 && t < n && !(seen & (1u << t)), SP_EINVAL, "split %lld: bad taxon index %d",
                       (long long)s, t);
            seen |= 1u << t;
            taxa8[s * n + i] = (int8_t)t;
        }
        kmax = std::max(kmax, std::min(a, n - a));
    }
    // positions sorted by size class (stable): the two-splits-a-wave kernel pairs neighbours of one class
    PairClasses pc;
    memset(&pc, 0, sizeof(pc));   // (padding too: the table is compared bytewise with the context's copy)
    std::vector<int> order((size_t)S);
    {
        std::vector<long long> per(n / 2 + 2, 0);
        for (int64_t s = 0; s < S; ++s) ++per[std::min(split_a[s], n - split_a[s])];
        std::vector<long long> at(n / 2 + 2, 0);
        for (int b = 1; b <= n / 2; ++b) {
            if (!per[b] || pc.nclass >= 16) continue;
            const int c = pc.nclass++;
            pc.rows[c] = 3 * b + 1;
            pc.start[c] = c ? pc.start[c - 1] + pc.count[c - 1] : 0;
            pc.count[c] = per[b];
            pc.poff[c + 1] = pc.poff[c] + (per[b] + 1) / 2;
            at[b] = pc.start[c];
        }
        for (int64_t s = 0; s < S; ++s) order[(size_t)at[std::min(split_a[s], n - split_a[s])]++] = (int)s;
    }
    const size_t taxa_bytes = (taxa8.size() + 15) & ~(size_t)15;
    SP_CHECK(ctx->coords.ensure(taxa_bytes + (size_t)S * 8 + 64));
    int8_t* dtaxa = ctx->coords.as<int8_t>();
    int* da = reinterpret_cast<int*>(dtaxa + taxa_bytes);
    int* dorder = da + S;
    SP_HIP(hipMemcpyAsync(dtaxa, taxa8.data(), taxa8.size(), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(da, split_a, (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(dorder, order.data(), (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));  // taxa8 and order are host temporaries
    return launch_subscore(al, dtaxa, da, S, kmax, nullptr, nullptr, n / 2 <= 16 ? &pc : nullptr, dorder);
}

// ---- every split of the taxa, enumerated on the device ----------------------------------------------------------------
// all_splits (reference splits.py:39-59): size classes ascending; inside a class itertools.combinations order (for the
// balanced class n/2|n/2: taxon 0 plus every (n/2 - 1)-subset of the others); the side holding taxon 0 first, both sides in
// taxon order.  Building those 524 267 Python tuples for 20 taxa costs seconds, encoding them 1.2 s, scoring them 21 ms:
// here thread i un-ranks combination i of its class (combinatorial number system) and writes the taxon list itself.
// Shards (multi-GPU, SURVEY 8e "index-mod-P within each class"): rank r of P enumerates the combinations r, r + P, ... of
// every class; thread `local` writes combination local * P + r at position local.
// Pascal's triangle up to 32, built at compile time: part of the code object (no upload, no host synchronisation a call).
struct BinomTable {
    unsigned long long v[33 * 33];
    constexpr BinomTable() : v{} {
        for (int i = 0; i < 33; ++i)
            for (int j = 0; j <= i; ++j) v[i * 33 + j] = (j == 0 || j == i) ? 1ull : v[(i - 1) * 33 + j - 1] + v[(i - 1) * 33 + j];
    }
};
__device__ const BinomTable g_binom{};

// the size classes of one enumeration (at most 16: sizes 1 .. 16 of <= 32 taxa), passed by value
struct EnumClasses {
    int nclass;
    int b[16];                        // size of the smaller side
    unsigned long long full[16];      // combinations of the class
    unsigned long long offset[17];    // first output position of the class (this shard's counts, prefix sums)
};

__global__ __launch_bounds__(256) void k_enumerate_splits(int n, EnumClasses cl, int8_t* __restrict__ taxa_out,
                                                          int* __restrict__ a_out, unsigned shard_rank, unsigned shard_world) {
    const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= cl.offset[cl.nclass]) return;
    int q = 0;
    while (q + 1 < cl.nclass && g >= cl.offset[q + 1]) ++q;
    const unsigned long long local = g - cl.offset[q];
    const unsigned long long idx = local * shard_world + shard_rank;   // < cl.full[q] by the definition of the shard's count
    const int bal = cl.b[q], even = 2 * bal == n ? 1 : 0;
    const unsigned long long* __restrict__ binom = g_binom.v;
    const int m = even ? n - 1 : n, r = even ? bal - 1 : bal, base = even ? 1 : 0;
    unsigned int member = even ? 1u : 0u;
    unsigned long long x = idx;
    int e = 0;
    for (int slot = 0; slot < r; ++slot) {
        // combinations starting with element e at this slot: C(m - e - 1, r - slot - 1)
        while (true) {
            const unsigned long long c = binom[(m - e - 1) * 33 + (r - slot - 1)];
            if (x < c) break;
            x -= c;
            ++e;
        }
        member |= 1u << (e + base);
        ++e;
    }
    const unsigned int all = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    const unsigned int first = (member & 1u) ? member : (~member & all);   // the side that holds taxon 0
    int8_t* out = taxa_out + g * (unsigned long long)n;
    int pos = 0;
    for (int t = 0; t < n; ++t)
        if (first & (1u << t)) out[pos++] = (int8_t)t;
    const int a = pos;
    for (int t = 0; t < n; ++t)
        if (!(first & (1u << t))) out[pos++] = (int8_t)t;
    a_out[g] = a;
}

static unsigned long long binom_host(int nn, int kk) {
    if (kk < 0 || kk > nn) return 0;
    unsigned long long c = 1;
    for (int i = 1; i <= kk; ++i) c = c * (unsigned long long)(nn - kk + i) / (unsigned long long)i;   // exact: c stays an integer
    return c;
}

// The splits all_splits yields, enumerated on the device into ctx->coords: taxa_out[total][n] (int8: the side holding
// taxon 0, then the other side, both in taxon order) and a_out[total].  `enumerate` = false only counts.  sizes / counts:
// the size classes in order and their populations.
int enumerate_all_splits(sp_ctx* ctx, int n, int trivial, int size, bool enumerate, int shard_rank, int shard_world,
                         int64_t* total_out, const int8_t** dtaxa_out, const int** da_out, std::vector<int>& sizes,
                         std::vector<unsigned long long>& counts) {
    // (counts = this shard's share of every class: combinations shard_rank, shard_rank + shard_world, ...)
    SP_REQUIRE(shard_world >= 1 && shard_rank >= 0 && shard_rank < shard_world, SP_EINVAL, "shard %d of %d", shard_rank,
               shard_world);
    SP_REQUIRE(n >= 2 && n <= 31, SP_ELIMIT, "split enumeration supports 2..31 taxa (got %d)", n);
    SP_REQUIRE(size >= 0 && size <= n / 2, SP_EINVAL, "size %d out of range [0, %d]", size, n / 2);
    sizes.clear();
    counts.clear();
    if (size > 0) sizes.push_back(size);
    else for (int b = trivial ? 1 : 2; b <= n / 2; ++b) sizes.push_back(b);
    int64_t total = 0;
    std::vector<unsigned long long> full;
    for (int b : sizes) {
        const bool even = 2 * b == n;
        const unsigned long long c = even ? binom_host(n - 1, b - 1) : binom_host(n, b);
        full.push_back(c);
        const unsigned long long mine = c > (unsigned long long)shard_rank
                                            ? (c - shard_rank + shard_world - 1) / (unsigned long long)shard_world : 0ull;
        counts.push_back(mine);
        total += (int64_t)mine;
    }
    if (total_out) *total_out = total;
    if (!enumerate || total == 0) return SP_OK;
    SP_REQUIRE(total < ((int64_t)1 << 31), SP_ELIMIT, "%lld splits: more than one call takes", (long long)total);
    SP_REQUIRE(sizes.size() <= 16, SP_ELIMIT, "%d size classes", (int)sizes.size());
    const size_t taxa_bytes = ((size_t)total * n + 15) & ~(size_t)15;
    // The enumeration depends on (n, trivial, size, shard) alone - it is the split LIST, what sp_plan is to the flattening
    // route - so it lives in a buffer of its own and is kept: a repeated call (the steps of a benchmark, the rounds of a
    // study over many alignments) finds it in place and launches nothing.
    const long long key[6] = {n, trivial, size, shard_rank, shard_world, (long long)total};
    int8_t* dtaxa = ctx->enum_buf.as<int8_t>();
    if (ctx->enum_valid && dtaxa && memcmp(ctx->enum_key, key, sizeof(key)) == 0) {
        *dtaxa_out = dtaxa;
        *da_out = reinterpret_cast<int*>(dtaxa + taxa_bytes);
        return SP_OK;
    }
    ctx->enum_valid = false;
    SP_CHECK(ctx->enum_buf.ensure(taxa_bytes + (size_t)total * 4 + 64));
    dtaxa = ctx->enum_buf.as<int8_t>();
    int* da = reinterpret_cast<int*>(dtaxa + taxa_bytes);
    // one launch for all classes, nothing uploaded and no host synchronisation (round 2 launched a kernel a class behind
    // an upload of the binomial table and a stream synchronise: 0.15 ms of a 0.5 ms step at 16 taxa)
    EnumClasses cl = {};
    cl.nclass = (int)sizes.size();
    for (size_t q = 0; q < sizes.size(); ++q) {
        cl.b[q] = sizes[q];
        cl.full[q] = full[q];
        cl.offset[q + 1] = cl.offset[q] + counts[q];
    }
    hipLaunchKernelGGL(k_enumerate_splits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, n, cl, dtaxa, da,
                       (unsigned)shard_rank, (unsigned)shard_world);
    SP_HIP(hipGetLastError());
    memcpy(ctx->enum_key, key, sizeof(key));
    ctx->enum_valid = true;
    *dtaxa_out = dtaxa;
    *da_out = da;
    return SP_OK;
}

// Number of splits all_splits yields, and - when scores are asked for - their subflattening scores in that order.
// scores_out / status_out (device, optional): where the kernel writes instead of the context's buffers.
int run_subflat_all_splits(sp_alignment* al, int trivial, int size, int shard_rank, int shard_world, int64_t* n_out,
                           bool score, double* scores_out, int* status_out) {
    sp_ctx* ctx = al->ctx;
    const int8_t* dtaxa = nullptr;
    const int* da = nullptr;
    std::vector<int> sizes;
    std::vector<unsigned long long> counts;
    int64_t total = 0;
    SP_CHECK(enumerate_all_splits(ctx, al->n_taxa, trivial, size, score, shard_rank, shard_world, &total, &dtaxa, &da, sizes,
                                  counts));
    if (n_out) *n_out = total;
    if (!score || total == 0) return SP_OK;
    int kmax = 0;
    PairClasses pc;
    memset(&pc, 0, sizeof(pc));   // (padding too: the table is compared bytewise with the context's copy)
    for (size_t q = 0; q < sizes.size(); ++q) {
        if (!counts[q]) continue;
        kmax = std::max(kmax, sizes[q]);
        const int c = pc.nclass++;
        pc.rows[c] = 3 * sizes[q] + 1;
        pc.start[c] = c ? pc.start[c - 1] + pc.count[c - 1] : 0;   // (classes without splits take no positions)
        pc.count[c] = (long long)counts[q];
        pc.poff[c + 1] = pc.poff[c] + (pc.count[c] + 1) / 2;
    }
    return launch_subscore(al, dtaxa, da, total, kmax, scores_out, status_out, &pc);
}

static int launch_subscore(sp_alignment* al, const int8_t* dtaxa, const int* da, int64_t S, int kmax, double* scores_out,
                           int* status_out, const PairClasses* pc, const int* order) {
    sp_ctx* ctx = al->ctx;
    const int n = al->n_taxa;
    SP_CHECK(ensure_moments(al));
    const int rmax = (3 * kmax + 1 + 1) & ~1;
    if (!scores_out) {
        SP_CHECK(ctx->scores.ensure((size_t)S * 8));
        scores_out = ctx->scores.as<double>();
    }
    if (!status_out) {
        SP_CHECK(ctx->status.ensure((size_t)S * 4));
        status_out = ctx->status.as<int>();
    }
    PhaseScope ps(ctx, SP_PHASE_SUBSCORE);
    const int mdim = 3 * n + 1;
    if (mdim <= SUBT_MMAX && rmax <= 32 && !ctx->opt.subscore_jacobi) {   // fast form (the option keeps the Jacobi kernel testable)
        const bool m32 = al->exact && al->N < ((int64_t)1 << 31);
        const int rt = 3 * kmax + 1;   // longest row side of the batch
        if (pc && ctx->opt.subscore_pair) return launch_subscore_pair(al, dtaxa, da, order, *pc, rt, scores_out, status_out);
        const size_t ms_bytes = ((size_t)mdim * mdim * (m32 ? 4 : 8) + 15) & ~(size_t)15;
        const int dev_cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
        const void* kfn = m32 ? reinterpret_cast<const void*>(k_subscore_tri<true, true>)
                              : (al->exact ? reinterpret_cast<const void*>(k_subscore_tri<true, false>)
                                           : reinterpret_cast<const void*>(k_subscore_tri<false, false>));
        static PerDeviceOnce attr_t;
        if (attr_t.need(ctx->device)) {
            SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_subscore_tri<true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_TOTAL));
            SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_subscore_tri<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_TOTAL));
            SP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_subscore_tri<false, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SPK_LDS_TOTAL));
            attr_t.done(ctx->device);
        }
        // Workgroup shape: the eigenvalue steps are dependency chains, so what counts is the number of waves resident on a
        // CU, and that is set by the LDS (one staged matrix per workgroup + one work area per wave).  Take the shape that
        // holds the most waves - whole multiples of 4 only (every SIMD the same number: the splits are dealt out statically,
        // so the fullest SIMD sets the time; 2 x 10 waves ran 20 % slower than 4 x 4 at 16 taxa), the smaller workgroup on
        // ties (4 x 4 waves 0.641 ms, 1 x 16 waves 0.654 ms at 16 taxa).  `subscore_waves` pins the workgroup (tests, tuning).
        // (the occupancy queries are host calls of several microseconds each: asked once per (kernel, longest side, matrix
        // bytes, pinned shape) and remembered - a 16-taxon step is 0.3 ms of kernel)
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
        for (int wv = 4; wv <= SUBT_MAXWAVES && !cached; wv += 4) {
            if (ctx->opt.subscore_waves > 0) continue;
            const size_t lds = ms_bytes + (size_t)wv * subt_wave_bytes(rt);
            if (lds > SPK_LDS_TOTAL) break;
            int pc = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, kfn, wv * 64, lds) != hipSuccess || pc < 1) continue;
            if (wv * pc > waves * per_cu) { waves = wv; per_cu = pc; }
        }
        if (ctx->opt.subscore_waves > 0 && !cached) {
            const int wv = std::min(ctx->opt.subscore_waves, SUBT_MAXWAVES);
            const size_t lds = ms_bytes + (size_t)wv * subt_wave_bytes(rt);
            int pc = 0;
            if (lds <= SPK_LDS_TOTAL && hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, kfn, wv * 64, lds) == hipSuccess && pc >= 1) {
                waves = wv;
                per_cu = pc;
            }
        }
        SP_REQUIRE(waves > 0, SP_ELIMIT, "subflattening score: no workgroup shape fits the LDS (%d taxa)", n);
        if (!cached) shape_cache.push_back({kfn, rt, ms_bytes, ctx->opt.subscore_waves, waves, per_cu});
        const size_t lds_t = ms_bytes + (size_t)waves * subt_wave_bytes(rt);
        const int64_t want_blocks = (S + waves - 1) / waves;
        // as many workgroups as are resident at once: persistent waves, grid-stride over the splits
        const unsigned blocks_t = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want_blocks, (int64_t)dev_cus * per_cu));
        SP_CHECK(ctx->eigws.ensure((size_t)blocks_t * waves * SUBT_SLOT * 8));   // per-wave slots of the scaled tridiagonal tables
        if (m32)
            hipLaunchKernelGGL((k_subscore_tri<true, true>), dim3(blocks_t), dim3(waves * 64), lds_t, ctx->stream,
                               al->moments.p, n, rt, dtaxa, da, S, scores_out, status_out, ctx->eigws.as<double>());
        else if (al->exact)
            hipLaunchKernelGGL((k_subscore_tri<true, false>), dim3(blocks_t), dim3(waves * 64), lds_t, ctx->stream,
                               al->moments.p, n, rt, dtaxa, da, S, scores_out, status_out, ctx->eigws.as<double>());
        else
            hipLaunchKernelGGL((k_subscore_tri<false, false>), dim3(blocks_t), dim3(waves * 64), lds_t, ctx->stream,
                               al->moments.p, n, rt, dtaxa, da, S, scores_out, status_out, ctx->eigws.as<double>());
        SP_HIP(hipGetLastError());
        return SP_OK;
    }
    const size_t lds = ((size_t)SUB_WAVES * rmax * (rmax + 1) + (size_t)SUB_WAVES * rmax) * 8;
    const unsigned blocks = (unsigned)((S + SUB_WAVES - 1) / SUB_WAVES);
    if (al->exact)
        hipLaunchKernelGGL(k_subscore<true>, dim3(blocks), dim3(SUB_WAVES * 64), lds, ctx->stream, al->moments.p, n,
                           dtaxa, da, S, rmax, scores_out, status_out);
    else
        hipLaunchKernelGGL(k_subscore<false>, dim3(blocks), dim3(SUB_WAVES * 64), lds, ctx->stream, al->moments.p, n,
                           dtaxa, da, S, rmax, scores_out, status_out);
    SP_HIP(hipGetLastError());
    return SP_OK;
}
