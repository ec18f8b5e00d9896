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

#include "common.h"

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

static int ensure_moments(sp_alignment* al) {
    if (al->moments_ready) return SP_OK;
    sp_ctx* ctx = al->ctx;
    const int m = 3 * al->n_taxa + 1, mm = m * m;
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
}

// idx(S) + [3n] for one split half
__device__ __forceinline__ int sub_index(const int8_t* taxa, int cnt, int n, int i) {
    return i < 3 * cnt ? 3 * taxa[i / 3] + (i % 3) : 3 * n;
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
}

// ---- batched score of subflattenings: one wave per split ------------------------------------------
// The block is at most 49 x 91; its Gram over the smaller side (r <= 49) is formed in LDS and
// diagonalised by parallel-order cyclic Jacobi; score = sqrt(max(0, 1 - top4 / trace)).
#define SUB_WAVES 4

__device__ __forceinline__ void wave_sync_lds2() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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
        if (!(off > 1e-30 * dg)) break;
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
    SP_CHECK(ensure_moments(al));
    const int rmax = (3 * kmax + 1 + 1) & ~1;
    SP_CHECK(ctx->coords.ensure(taxa8.size() + (size_t)S * 4 + 64));
    SP_CHECK(ctx->scores.ensure((size_t)S * 8));
    SP_CHECK(ctx->status.ensure((size_t)S * 4));
    int8_t* dtaxa = ctx->coords.as<int8_t>();
    int* da = reinterpret_cast<int*>(dtaxa + ((taxa8.size() + 15) & ~(size_t)15));
    SP_HIP(hipMemcpyAsync(dtaxa, taxa8.data(), taxa8.size(), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(da, split_a, (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));  // taxa8 is a host temporary
    PhaseScope ps(ctx, SP_PHASE_SUBSCORE);
    const size_t lds = ((size_t)SUB_WAVES * rmax * (rmax + 1) + (size_t)SUB_WAVES * rmax) * 8;
    const unsigned blocks = (unsigned)((S + SUB_WAVES - 1) / SUB_WAVES);
    if (al->exact)
        hipLaunchKernelGGL(k_subscore<true>, dim3(blocks), dim3(SUB_WAVES * 64), lds, ctx->stream, al->moments.p, n,
                           dtaxa, da, S, rmax, ctx->scores.as<double>(), ctx->status.as<int>());
    else
        hipLaunchKernelGGL(k_subscore<false>, dim3(blocks), dim3(SUB_WAVES * 64), lds, ctx->stream, al->moments.p, n,
                           dtaxa, da, S, rmax, ctx->scores.as<double>(), ctx->status.as<int>());
    SP_HIP(hipGetLastError());
    return SP_OK;
}
