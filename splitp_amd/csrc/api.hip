// Host side of libsplitp_hip.so: context / alignment handles, batch planning, the C ABI.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"

// ------------------------------------------------------------------ errors ---------------------
static thread_local char g_err[1024] = "";
void sp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sp_last_error(void) { return g_err; }
extern "C" int sp_abi_version(void) { return SP_ABI_VERSION; }
extern "C" int sp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap) return SP_OK;
    if (p) {
        SP_HIP(hipDeviceSynchronize());
        SP_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t want = std::max(bytes, (size_t)4096);
    want += want / 8;  // slack so slowly growing shapes do not re-allocate every call
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        sp_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return SP_ENOMEM;
    }
    cap = want;
    return SP_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// ------------------------------------------------------------------ timing ---------------------
PhaseScope::PhaseScope(sp_ctx* ctx, int ph) : c(ctx), phase(ph) {
    if (!c->timing) return;
    auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!c->timer.pool.empty()) {
            e = c->timer.pool.back();
            c->timer.pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
        return e;
    };
    e0 = get();
    e1 = get();
    if (e0) (void)hipEventRecord(e0, c->stream);
}
PhaseScope::~PhaseScope() {
    if (!c->timing || !e0 || !e1) return;
    (void)hipEventRecord(e1, c->stream);
    c->timer.pending[phase].push_back({e0, e1});
}

static int drain_timers(sp_ctx* ctx) {
    SP_HIP(hipStreamSynchronize(ctx->stream));
    for (int ph = 0; ph < SP_N_PHASES; ++ph) {
        for (auto& pr : ctx->timer.pending[ph]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                ctx->timer.ms[ph] += ms;
                ctx->timer.launches[ph] += 1;
            }
            ctx->timer.pool.push_back(pr.first);
            ctx->timer.pool.push_back(pr.second);
        }
        ctx->timer.pending[ph].clear();
    }
    return SP_OK;
}

// ------------------------------------------------------------------ context --------------------
extern "C" int sp_ctx_create(int device, void* stream, sp_ctx** out) {
    return sp_guard("sp_ctx_create", [&]() -> int {
    SP_REQUIRE(out, SP_EINVAL, "sp_ctx_create: out is NULL");
    int n = sp_device_count();
    SP_REQUIRE(n > 0, SP_EHIP, "sp_ctx_create: no HIP device visible (this library has no CPU fallback)");
    SP_REQUIRE(device >= 0 && device < n, SP_EINVAL, "sp_ctx_create: device %d out of range (have %d)", device, n);
    SP_HIP(hipSetDevice(device));
    sp_ctx* c = new sp_ctx();
    c->device = device;
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            sp_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return SP_EHIP;
        }
        c->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    // test / tuning switches: the environment is read here, once (later changes: sp_ctx_set_option)
    const char* gm = getenv("SPLITP_GRAM");
    if (gm && strcmp(gm, "f64") == 0) c->gram_mode = 1;
    auto env_flag = [](const char* name) {
        const char* v = getenv(name);
        return v && v[0] && v[0] != '0';
    };
    c->opt.force_big = env_flag("SPLITP_FORCE_BIG");
    c->opt.big_by_keys = env_flag("SPLITP_BIG_BY_KEYS");
    c->opt.subscore_jacobi = env_flag("SPLITP_SUBSCORE_JACOBI");
    if (const char* sw = getenv("SPLITP_SUBSCORE_WAVES")) c->opt.subscore_waves = atoi(sw);
    if (const char* sp = getenv("SPLITP_SUBSCORE_PAIR")) c->opt.subscore_pair = atoi(sp);
    c->opt.divergence_global = env_flag("SPLITP_DIVERGENCE_GLOBAL");
    c->opt.gram_tile64 = env_flag("SPLITP_GRAM_TILE64");
    c->opt.eigen_one_stream = env_flag("SPLITP_EIGEN_ONE_STREAM");
    if (const char* hs = getenv("SPLITP_HIST_SORT")) c->opt.hist_sort = hs[0] == '1' ? 1 : (hs[0] == '0' ? 0 : -1);
    if (const char* lc = getenv("SPLITP_DEBUG_LDS_CAP")) c->opt.lds_cap = atoll(lc);
    if (const char* df = getenv("SPLITP_DIRECT_FINISH")) c->opt.direct_finish = df[0] != '0';
    c->opt.direct_all = env_flag("SPLITP_DIRECT_ALL");
    c->opt.eigen_block16 = env_flag("SPLITP_EIGEN_BLOCK16");
    c->opt.sort_three_launch = env_flag("SPLITP_SORT_THREE_LAUNCH");
    *out = c;
    return SP_OK;
    });
}

static long long* option_slot(sp_ctx* c, const char* name, int** as_int) {
    *as_int = nullptr;
    if (!strcmp(name, "force_big")) *as_int = &c->opt.force_big;
    else if (!strcmp(name, "big_by_keys")) *as_int = &c->opt.big_by_keys;
    else if (!strcmp(name, "subscore_jacobi")) *as_int = &c->opt.subscore_jacobi;
    else if (!strcmp(name, "subscore_waves")) *as_int = &c->opt.subscore_waves;
    else if (!strcmp(name, "subscore_pair")) *as_int = &c->opt.subscore_pair;
    else if (!strcmp(name, "moments_valu")) *as_int = &c->opt.moments_valu;
    else if (!strcmp(name, "divergence_global")) *as_int = &c->opt.divergence_global;
    else if (!strcmp(name, "hist_sort")) *as_int = &c->opt.hist_sort;
    else if (!strcmp(name, "wide_cap")) *as_int = &c->opt.wide_cap;
    else if (!strcmp(name, "gram_tile64")) *as_int = &c->opt.gram_tile64;
    else if (!strcmp(name, "eigen_one_stream")) *as_int = &c->opt.eigen_one_stream;
    else if (!strcmp(name, "direct_finish")) *as_int = &c->opt.direct_finish;
    else if (!strcmp(name, "direct_max_rows")) *as_int = &c->opt.direct_max_rows;
    else if (!strcmp(name, "direct_all")) *as_int = &c->opt.direct_all;
    else if (!strcmp(name, "eigen_block16")) *as_int = &c->opt.eigen_block16;
    else if (!strcmp(name, "sort_three_launch")) *as_int = &c->opt.sort_three_launch;
    else if (!strcmp(name, "sort_digit_bits")) *as_int = &c->opt.sort_digit_bits;
    else if (!strcmp(name, "lds_cap")) return &c->opt.lds_cap;
    return nullptr;
}

extern "C" int sp_ctx_set_option(sp_ctx* c, const char* name, int64_t value) {
    return sp_guard("sp_ctx_set_option", [&]() -> int {
    SP_REQUIRE(c && name, SP_EINVAL, "sp_ctx_set_option: NULL argument");
    int* pi = nullptr;
    long long* pl = option_slot(c, name, &pi);
    SP_REQUIRE(pi || pl, SP_EINVAL, "sp_ctx_set_option: unknown option '%s'", name);
    SP_REQUIRE(value >= -1 && value <= ((int64_t)1 << 30), SP_EINVAL, "sp_ctx_set_option: value %lld out of range",
               (long long)value);
    if (pi) *pi = (int)value; else *pl = (long long)value;
    if (c->cache) c->cache->valid = false;
    return SP_OK;
    });
}

extern "C" int sp_ctx_get_option(sp_ctx* c, const char* name, int64_t* value) {
    return sp_guard("sp_ctx_get_option", [&]() -> int {
    SP_REQUIRE(c && name && value, SP_EINVAL, "sp_ctx_get_option: NULL argument");
    int* pi = nullptr;
    long long* pl = option_slot(c, name, &pi);
    SP_REQUIRE(pi || pl, SP_EINVAL, "sp_ctx_get_option: unknown option '%s'", name);
    *value = pi ? (int64_t)*pi : (int64_t)*pl;
    return SP_OK;
    });
}

extern "C" int sp_ctx_destroy(sp_ctx* c) {
    return sp_guard("sp_ctx_destroy", [&]() -> int {
    if (!c) return SP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int ph = 0; ph < SP_N_PHASES; ++ph)
        for (auto& pr : c->timer.pending[ph]) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (auto e : c->timer.pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&c->splits, &c->bitmaps, &c->coords, &c->dims, &c->mats,  &c->grams,  &c->eigws,
                      &c->scores, &c->status,  &c->misc,   &c->misc2, &c->gram_items, &c->aldescs, &c->slabs, &c->chain,
                      &c->splits_launch, &c->hist_bins, &c->hist_blk, &c->hist_off, &c->enum_buf, &c->pair_dev};
    if (c->cache && c->cache->sparse) (void)sp_plan_release(c->cache->sparse);
    delete c->cache;
    for (auto* b : bufs) b->release();
    for (auto& b : c->big) b.release();
    for (auto& b : c->hist_work) b.release();
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
        (void)hipEventDestroy(c->ev_fork);
        (void)hipEventDestroy(c->ev_join);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SP_OK;
    });
}

extern "C" int sp_ctx_set_stream(sp_ctx* c, void* stream) {
    return sp_guard("sp_ctx_set_stream", [&]() -> int {
    SP_REQUIRE(c, SP_EINVAL, "sp_ctx_set_stream: ctx is NULL");
    SP_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
        c->own_stream = false;
    } else {
        SP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return SP_OK;
    });
}

extern "C" int sp_ctx_set_gram_mode(sp_ctx* c, int mode) {
    return sp_guard("sp_ctx_set_gram_mode", [&]() -> int {
    SP_REQUIRE(c, SP_EINVAL, "ctx is NULL");
    SP_REQUIRE(mode == 0 || mode == 1, SP_EINVAL, "gram mode must be 0 (auto) or 1 (fp64)");
    c->gram_mode = mode;
    if (c->cache) c->cache->valid = false;
    return SP_OK;
    });
}

extern "C" int sp_ctx_synchronize(sp_ctx* c) {
    return sp_guard("sp_ctx_synchronize", [&]() -> int {
    SP_REQUIRE(c, SP_EINVAL, "sp_ctx_synchronize: ctx is NULL");
    SP_HIP(hipStreamSynchronize(c->stream));
    return SP_OK;
    });
}

extern "C" int sp_ctx_enable_timing(sp_ctx* c, int on) {
    return sp_guard("sp_ctx_enable_timing", [&]() -> int {
    SP_REQUIRE(c, SP_EINVAL, "ctx is NULL");
    c->timing = on != 0;
    // Events are created HERE, not lazily inside the first timed launches: a 20-step measurement right after the switch
    // (the driver's bench command) otherwise paid a hipEventCreate x 4 per step - host_us_per_step 63 against 30 us of a
    // long run.  A phase scope takes two events; they return to the pool when the times are read.
    if (c->timing) {
        SP_HIP(hipSetDevice(c->device));
        while (c->timer.pool.size() < 1024) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            c->timer.pool.push_back(e);
        }
    }
    return SP_OK;
    });
}
extern "C" int sp_ctx_reset_timing(sp_ctx* c) {
    return sp_guard("sp_ctx_reset_timing", [&]() -> int {
    SP_REQUIRE(c, SP_EINVAL, "ctx is NULL");
    SP_CHECK(drain_timers(c));
    for (int i = 0; i < SP_N_PHASES; ++i) {
        c->timer.ms[i] = 0;
        c->timer.launches[i] = 0;
    }
    return SP_OK;
    });
}
extern "C" int sp_ctx_phase_times(sp_ctx* c, double* ms, int64_t* launches) {
    return sp_guard("sp_ctx_phase_times", [&]() -> int {
    SP_REQUIRE(c, SP_EINVAL, "ctx is NULL");
    SP_CHECK(drain_timers(c));
    for (int i = 0; i < SP_N_PHASES; ++i) {
        if (ms) ms[i] = c->timer.ms[i];
        if (launches) launches[i] = c->timer.launches[i];
    }
    return SP_OK;
    });
}

// ------------------------------------------------------------------ alignment ------------------
extern "C" int sp_alignment_create(sp_ctx* ctx, const uint64_t* keys, const double* weights, const int64_t* counts,
                                   int64_t D, int n_taxa, int64_t N, sp_alignment** out) {
    return sp_guard("sp_alignment_create", [&]() -> int {
    SP_REQUIRE(ctx && out, SP_EINVAL, "sp_alignment_create: NULL ctx/out");
    SP_REQUIRE(n_taxa >= 2 && n_taxa <= 32, SP_EINVAL, "n_taxa must be in [2, 32], got %d", n_taxa);
    SP_REQUIRE(D >= 0, SP_EINVAL, "D < 0");
    SP_REQUIRE(D == 0 || (keys && (weights || counts)), SP_EINVAL, "keys and weights/counts are required");
    SP_HIP(hipSetDevice(ctx->device));
    const u64 lim = n_taxa == 32 ? ~0ull : ((1ull << (2 * n_taxa)) - 1);
    std::vector<u32> c32;
    std::vector<double> w;
    double sumsq = 0;
    u32 maxc = 0;
    if (counts) {
        c32.resize(D);
        for (int64_t i = 0; i < D; ++i) {
            SP_REQUIRE(counts[i] >= 0 && counts[i] <= 0xFFFFFFFFll, SP_EINVAL, "count %lld out of uint32 range",
                       (long long)counts[i]);
            c32[i] = (u32)counts[i];
            if (c32[i] > maxc) maxc = c32[i];
        }
        SP_REQUIRE(N > 0, SP_EINVAL, "N must be positive when counts are given");
    }
    w.resize(D);
    for (int64_t i = 0; i < D; ++i) {
        SP_REQUIRE(keys[i] <= lim, SP_EINVAL, "pattern key %llu does not fit %d taxa", (unsigned long long)keys[i],
                   n_taxa);
        w[i] = weights ? weights[i] : (double)counts[i] / (double)N;
        sumsq += w[i] * w[i];
    }
    sp_alignment* al = new sp_alignment();
    al->ctx = ctx;
    al->n_taxa = n_taxa;
    al->D = D;
    al->N = counts ? N : 0;
    al->exact = counts != nullptr;
    al->sumsq_w = sumsq;
    al->max_count = maxc;
    int rc = SP_OK;
    const size_t d1 = (size_t)std::max<int64_t>(D, 1);
    if ((rc = al->keys.ensure(d1 * 8)) || (rc = al->weights.ensure(d1 * 8)) || (rc = al->counts.ensure(d1 * 4 + SP_COUNTS_PAD))) {
        sp_alignment_destroy(al);
        return rc;
    }
    if (D > 0) {
        // synchronous copies: the host vectors above die at return
        SP_HIP(hipMemcpy(al->keys.p, keys, D * 8, hipMemcpyHostToDevice));
        SP_HIP(hipMemcpy(al->weights.p, w.data(), D * 8, hipMemcpyHostToDevice));
        if (counts) SP_HIP(hipMemcpy(al->counts.p, c32.data(), D * 4, hipMemcpyHostToDevice));
    }
    *out = al;
    return SP_OK;
    });
}

extern "C" int sp_alignment_destroy(sp_alignment* al) {
    return sp_guard("sp_alignment_destroy", [&]() -> int {
    if (!al) return SP_OK;
    (void)hipSetDevice(al->ctx->device);
    // lanes (other contexts, other streams) may still be reading keys32 / counts / aldesc of this table: wait for the whole
    // device, not only for the creator's stream (sp_plan_release does the same)
    (void)hipDeviceSynchronize();
    al->keys.release();
    al->weights.release();
    al->counts.release();
    al->keys32.release();
    al->spk_meta.release();
    al->aldesc.release();
    al->spk_keys.release();
    al->spk_counts.release();
    al->moments.release();
    delete al;
    return SP_OK;
    });
}

extern "C" int sp_alignment_info(const sp_alignment* al, int64_t* D, int* n_taxa, int64_t* N, int* exact) {
    return sp_guard("sp_alignment_info", [&]() -> int {
    SP_REQUIRE(al, SP_EINVAL, "alignment is NULL");
    if (D) *D = al->D;
    if (n_taxa) *n_taxa = al->n_taxa;
    if (N) *N = al->N;
    if (exact) *exact = al->exact ? 1 : 0;
    return SP_OK;
    });
}

extern "C" int sp_alignment_fetch(sp_alignment* al, uint64_t* keys, double* weights, int64_t* counts) {
    return sp_guard("sp_alignment_fetch", [&]() -> int {
    SP_REQUIRE(al, SP_EINVAL, "alignment is NULL");
    SP_HIP(hipSetDevice(al->ctx->device));
    SP_HIP(hipStreamSynchronize(al->ctx->stream));
    if (al->D == 0) return SP_OK;
    if (keys) SP_HIP(hipMemcpy(keys, al->keys.p, al->D * 8, hipMemcpyDeviceToHost));
    if (weights) SP_HIP(hipMemcpy(weights, al->weights.p, al->D * 8, hipMemcpyDeviceToHost));
    if (counts) {
        SP_REQUIRE(al->exact, SP_EINVAL, "alignment holds no integer counts");
        std::vector<u32> c(al->D);
        SP_HIP(hipMemcpy(c.data(), al->counts.p, al->D * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < al->D; ++i) counts[i] = c[i];
    }
    return SP_OK;
    });
}

// ------------------------------------------------------------------ split validation / planning
static int check_split(int n, const int32_t* oa, int a, const int32_t* ob, int b) {
    SP_REQUIRE(oa && ob, SP_EINVAL, "split order arrays are NULL");
    SP_REQUIRE(a >= 1 && b >= 1 && a + b == n, SP_EINVAL,
               "a split must cover all %d taxa of the table (got %d + %d): the reference overwrites colliding "
               "cells for partial splits (constructions.py:43,:101), which is rejected here",
               n, a, b);
    unsigned seen = 0;
    for (int i = 0; i < a + b; ++i) {
        const int t = i < a ? oa[i] : ob[i - a];
        SP_REQUIRE(t >= 0 && t < n, SP_EINVAL, "taxon index %d out of range [0, %d)", t, n);
        SP_REQUIRE(!(seen & (1u << t)), SP_EINVAL, "taxon %d appears twice in the split", t);
        seen |= 1u << t;
    }
    return SP_OK;
}

// rows_first_small: orient every split so that the smaller side indexes the rows (scoring).
static int limbs_for(const sp_alignment* al) {
    if (!al->exact || al->ctx->gram_mode == 1) return 0;
    if (al->max_count < (1u << 7)) return 1;
    if (al->max_count < (1u << 14)) return 2;
    if (al->max_count < (1u << 21)) return 3;
    return 0;
}

// nl > 0: matrices are nl int8 limb planes (pitch in bytes, multiple of 128) instead of one typed matrix.
static int plan_splits(int n, int64_t D, const int32_t* split_taxa, const int32_t* split_a, int64_t S,
                       bool small_rows, bool want_mats, bool want_gram, Plan& plan, int nl = 0) {
    plan = Plan();
    plan.splits.resize(S);
    for (int64_t s = 0; s < S; ++s) {
        const int a = split_a[s], b = n - a;
        const int32_t* oa = split_taxa + s * n;
        const int32_t* ob = oa + a;
        SP_CHECK(check_split(n, oa, a, ob, b));
        SplitDev& sd = plan.splits[s];
        memset(&sd, 0, sizeof(sd));
        const bool swap = small_rows && a > b;
        sd.nr = swap ? b : a;
        sd.nc = swap ? a : b;
        for (int i = 0; i < sd.nr; ++i) sd.taxa[i] = (int8_t)(swap ? ob[i] : oa[i]);
        for (int i = 0; i < sd.nc; ++i) sd.taxa[sd.nr + i] = (int8_t)(swap ? oa[i] : ob[i]);
        SP_REQUIRE(sd.nr <= 14 && sd.nc <= 14, SP_ELIMIT,
                   "split side of %d taxa: the bitmap compaction of this build supports sides up to 14 taxa",
                   std::max(sd.nr, sd.nc));
        sd.rw = (int32_t)((pow4(sd.nr) + 63) / 64);
        sd.cw = (int32_t)((pow4(sd.nc) + 63) / 64);
        sd.bm_off = (int64_t)plan.bm_words;
        sd.pfx_off = (int64_t)plan.pf_words;
        plan.bm_words += (size_t)sd.rw + sd.cw;
        plan.pf_words += (size_t)sd.rw + sd.cw;
        const int64_t rmax = std::min<int64_t>(pow4(sd.nr), std::max<int64_t>(D, 1));
        const int64_t cmax = std::min<int64_t>(pow4(sd.nc), std::max<int64_t>(D, 1));
        sd.rcap = (int32_t)round_up(rmax, 64);
        sd.pitch = (int32_t)round_up(cmax, nl > 0 ? 128 : 32);
        if (want_mats) {
            sd.mat_off = (int64_t)plan.mat_elems;
            plan.mat_elems += (size_t)sd.rcap * sd.pitch * (nl > 0 ? nl : 1);
        }
        if (want_gram) {
            sd.g_pitch = sd.rcap;
            sd.g_off = (int64_t)plan.g_elems;
            plan.g_elems += (size_t)sd.rcap * sd.rcap;
            sd.ev_off = (int64_t)plan.ev_elems;
            plan.ev_elems += (size_t)sd.rcap * 16;
        }
    }
    if (want_gram) build_gram_items(plan);
    return SP_OK;
}

static GramItem* big_items_ptr(sp_ctx* ctx, const Plan& plan) {
    return reinterpret_cast<GramItem*>(reinterpret_cast<int*>(ctx->gram_items.as<GramItem>() + plan.gram_items.size() +
                                                              plan.row_items.size()) + round_up(plan.order.size(), 2));
}

static int upload_items(sp_ctx* ctx, const Plan& plan) {
    if (plan.gram_items.empty()) return SP_OK;
    const size_t ng = plan.gram_items.size(), nr = plan.row_items.size(), no = plan.order.size();
    const size_t nb = plan.gram_items_big.size();
    SP_CHECK(ctx->gram_items.ensure((ng + nr + nb) * sizeof(GramItem) + round_up(no, 2) * sizeof(int)));
    if (nb)   // layout: 64-tiles | row blocks | order (padded to 8 bytes) | 128-tiles
        SP_HIP(hipMemcpyAsync(big_items_ptr(ctx, plan), plan.gram_items_big.data(), nb * sizeof(GramItem),
                              hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(ctx->gram_items.as<GramItem>() + ng + nr, plan.order.data(), no * sizeof(int),
                          hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(ctx->gram_items.p, plan.gram_items.data(), ng * sizeof(GramItem), hipMemcpyHostToDevice,
                          ctx->stream));
    SP_HIP(hipMemcpyAsync(ctx->gram_items.as<GramItem>() + ng, plan.row_items.data(), nr * sizeof(GramItem),
                          hipMemcpyHostToDevice, ctx->stream));
    return SP_OK;
}

static int upload_plan(sp_ctx* ctx, const Plan& plan, int64_t D) {
    const size_t S = plan.splits.size();
    if (ctx->cache) ctx->cache->valid = false;  // the device copy of any cached plan is overwritten
    SP_CHECK(ctx->splits.ensure(S * sizeof(SplitDev)));
    SP_CHECK(ctx->bitmaps.ensure(plan.bm_words * 8 + plan.pf_words * 4 + 64));
    SP_CHECK(ctx->coords.ensure(S * (size_t)std::max<int64_t>(D, 1) * 8));
    SP_CHECK(ctx->dims.ensure(S * sizeof(int2)));
    SP_HIP(hipMemcpyAsync(ctx->splits.p, plan.splits.data(), S * sizeof(SplitDev), hipMemcpyHostToDevice, ctx->stream));
    SP_CHECK(upload_items(ctx, plan));
    return SP_OK;
}

static u64* bm_ptr(sp_ctx* ctx) { return ctx->bitmaps.as<u64>(); }
static u32* pf_ptr(sp_ctx* ctx, const Plan& plan) { return reinterpret_cast<u32*>(ctx->bitmaps.as<u64>() + plan.bm_words); }
static u32* rr_ptr(sp_ctx* ctx) { return ctx->coords.as<u32>(); }
static u32* cc_ptr(sp_ctx* ctx, size_t S, int64_t D) { return ctx->coords.as<u32>() + S * (size_t)std::max<int64_t>(D, 1); }

// ------------------------------------------------------------------ flattening API ------------
extern "C" int sp_flatten_indices(sp_alignment* al, const int32_t* oa, int a, const int32_t* ob, int b, int64_t* rows,
                                  int64_t* cols) {
    return sp_guard("sp_flatten_indices", [&]() -> int {
    SP_REQUIRE(al && rows && cols, SP_EINVAL, "NULL argument");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    SP_CHECK(check_split(al->n_taxa, oa, a, ob, b));
    if (al->D == 0) return SP_OK;
    SplitDev sd;
    memset(&sd, 0, sizeof(sd));
    sd.nr = a;
    sd.nc = b;
    for (int i = 0; i < a; ++i) sd.taxa[i] = (int8_t)oa[i];
    for (int i = 0; i < b; ++i) sd.taxa[a + i] = (int8_t)ob[i];
    SP_CHECK(ctx->splits.ensure(sizeof(SplitDev)));
    SP_CHECK(ctx->misc.ensure((size_t)al->D * 16));
    if (ctx->cache) ctx->cache->valid = false;
    SP_HIP(hipMemcpyAsync(ctx->splits.p, &sd, sizeof(sd), hipMemcpyHostToDevice, ctx->stream));
    int64_t* drows = ctx->misc.as<int64_t>();
    int64_t* dcols = drows + al->D;
    SP_CHECK(launch_bit_indices(ctx, al->keys.as<u64>(), al->D, al->n_taxa, ctx->splits.as<SplitDev>(), drows, dcols));
    SP_HIP(hipMemcpyAsync(rows, drows, al->D * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipMemcpyAsync(cols, dcols, al->D * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    return SP_OK;
    });
}

extern "C" int sp_flatten_reduced_prepare(sp_alignment* al, const int32_t* oa, int a, const int32_t* ob, int b,
                                          int64_t* R, int64_t* C) {
    return sp_guard("sp_flatten_reduced_prepare", [&]() -> int {
    SP_REQUIRE(al && R && C, SP_EINVAL, "NULL argument");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    al->red_ready = false;
    std::vector<int32_t> taxa(al->n_taxa);
    SP_CHECK(check_split(al->n_taxa, oa, a, ob, b));
    for (int i = 0; i < a; ++i) taxa[i] = oa[i];
    for (int i = 0; i < b; ++i) taxa[a + i] = ob[i];
    if (al->D == 0) {
        *R = *C = 0;
        al->red_R = al->red_C = 0;
        al->red_ready = true;
        return SP_OK;
    }
    Plan plan;
    int32_t aa = a;
    SP_CHECK(plan_splits(al->n_taxa, al->D, taxa.data(), &aa, 1, false, false, false, plan));
    SP_CHECK(upload_plan(ctx, plan, al->D));
    SP_CHECK(launch_reindex(ctx, al->keys.as<u64>(), al->D, al->n_taxa, ctx->splits.as<SplitDev>(), plan.splits,
                            bm_ptr(ctx), pf_ptr(ctx, plan), ctx->dims.as<int2>(), rr_ptr(ctx), cc_ptr(ctx, 1, al->D)));
    int2 d;
    SP_HIP(hipMemcpyAsync(&d, ctx->dims.p, sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    *R = al->red_R = d.x;
    *C = al->red_C = d.y;
    al->red_ready = true;
    return SP_OK;
    });
}

__global__ void k_scatter_exact_shape(int64_t D, const u32* __restrict__ rr, const u32* __restrict__ cc,
                                      const double* __restrict__ vals, double* __restrict__ out, int64_t C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) out[(int64_t)rr[i] * C + cc[i]] = vals[i];
}

extern "C" int sp_flatten_reduced_fetch(sp_alignment* al, double* matrix, int64_t* row_keys, int64_t* col_keys) {
    return sp_guard("sp_flatten_reduced_fetch", [&]() -> int {
    SP_REQUIRE(al && al->red_ready, SP_EINVAL, "sp_flatten_reduced_fetch without a preceding _prepare");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    const int64_t R = al->red_R, C = al->red_C, D = al->D;
    al->red_ready = false;
    if (D == 0 || R == 0 || C == 0) return SP_OK;
    const size_t mbytes = (size_t)R * C * 8;
    SP_CHECK(ctx->misc.ensure(mbytes + (size_t)(R + C) * 8));
    double* dm = ctx->misc.as<double>();
    int64_t* drk = reinterpret_cast<int64_t*>(dm + R * C);
    int64_t* dck = drk + R;
    if (matrix) {
        SP_HIP(hipMemsetAsync(dm, 0, mbytes, ctx->stream));
        hipLaunchKernelGGL(k_scatter_exact_shape, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, ctx->stream, D,
                           rr_ptr(ctx), cc_ptr(ctx, 1, D), al->weights.as<double>(), dm, C);
        SP_HIP(hipGetLastError());
        SP_HIP(hipMemcpyAsync(matrix, dm, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (row_keys || col_keys) {
        SP_CHECK(launch_used_keys(ctx, al->keys.as<u64>(), D, al->n_taxa, ctx->splits.as<SplitDev>(), rr_ptr(ctx),
                                  cc_ptr(ctx, 1, D), drk, dck));
        if (row_keys) SP_HIP(hipMemcpyAsync(row_keys, drk, R * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (col_keys) SP_HIP(hipMemcpyAsync(col_keys, dck, C * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    SP_HIP(hipStreamSynchronize(ctx->stream));
    return SP_OK;
    });
}

extern "C" int sp_flatten_dense_counts(sp_alignment* al, const int32_t* oa, int a, const int32_t* ob, int b,
                                       uint32_t* out_host) {
    return sp_guard("sp_flatten_dense_counts", [&]() -> int {
    SP_REQUIRE(al && out_host, SP_EINVAL, "NULL argument");
    SP_REQUIRE(al->exact, SP_EINVAL, "sp_flatten_dense_counts needs an alignment with integer counts");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    SP_CHECK(check_split(al->n_taxa, oa, a, ob, b));
    SP_REQUIRE(al->n_taxa <= 14, SP_ELIMIT, "dense flattening of %d taxa would need 4^%d cells", al->n_taxa, al->n_taxa);
    SplitDev sd;
    memset(&sd, 0, sizeof(sd));
    sd.nr = a;
    sd.nc = b;
    for (int i = 0; i < a; ++i) sd.taxa[i] = (int8_t)oa[i];
    for (int i = 0; i < b; ++i) sd.taxa[a + i] = (int8_t)ob[i];
    const int64_t cells = pow4(al->n_taxa);
    SP_CHECK(ctx->splits.ensure(sizeof(SplitDev)));
    SP_CHECK(ctx->mats.ensure((size_t)cells * 4));
    if (ctx->cache) ctx->cache->valid = false;
    SP_HIP(hipMemcpyAsync(ctx->splits.p, &sd, sizeof(sd), hipMemcpyHostToDevice, ctx->stream));
    SP_CHECK(launch_dense_scatter(ctx, al->keys.as<u64>(), al->counts.as<u32>(), al->D, al->n_taxa,
                                  ctx->splits.as<SplitDev>(), sd, ctx->mats.as<u32>()));
    SP_HIP(hipMemcpyAsync(out_host, ctx->mats.p, (size_t)cells * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    return SP_OK;
    });
}

// ------------------------------------------------------------------ scoring --------------------
static int run_dense_route(sp_alignment* al, const Plan& plan, bool plan_on_device) {
    sp_ctx* ctx = al->ctx;
    const size_t S = plan.splits.size();
    const int64_t D = al->D;
    if (!plan_on_device) SP_CHECK(upload_plan(ctx, plan, D));
    SP_CHECK(ctx->grams.ensure(plan.g_elems * 8));
    SP_CHECK(ctx->scores.ensure(S * 8));
    SP_CHECK(ctx->status.ensure(S * 4));
    const SplitDev* sdev = ctx->splits.as<SplitDev>();
    int2* dims = ctx->dims.as<int2>();
    SP_CHECK(launch_reindex(ctx, al->keys.as<u64>(), D, al->n_taxa, sdev, plan.splits, bm_ptr(ctx), pf_ptr(ctx, plan),
                            dims, rr_ptr(ctx), cc_ptr(ctx, S, D)));
    const int nl = ctx->cache ? ctx->cache->nl : 0;
    bool g_i32 = false;
    if (al->exact && nl > 0) {
        SP_CHECK(ctx->mats.ensure(plan.mat_elems + 256));
        SP_CHECK(launch_zero_scatter_i8(ctx, nl, sdev, plan.splits, D, dims, rr_ptr(ctx), cc_ptr(ctx, S, D),
                                        al->counts.as<u32>(), ctx->mats.as<uint8_t>()));
        // G entries are bounded by max_i sum_c C[i][c]^2 <= max_count * N: store int32 when that fits
        g_i32 = (unsigned long long)al->max_count * (unsigned long long)al->N < (1ull << 31);
        int64_t kmax = 0;
        for (const SplitDev& sd : plan.splits) kmax = std::max<int64_t>(kmax, sd.pitch);
        if (kmax <= 65536 && !plan.gram_items_big.empty() && ctx->opt.gram_tile64 == 0)
            SP_CHECK(launch_gram_i8_big(ctx, nl, g_i32, sdev, big_items_ptr(ctx, plan), (int64_t)plan.gram_items_big.size(),
                                        dims, ctx->mats.as<uint8_t>(), ctx->grams.p));
        else   // K beyond one int32 chunk (tables of > 65536 patterns): the 64 x 64 form flushes into int64 as it goes
            SP_CHECK(launch_gram_i8(ctx, nl, g_i32, sdev, ctx->gram_items.as<GramItem>(), (int64_t)plan.gram_items.size(),
                                    dims, ctx->mats.as<uint8_t>(), ctx->grams.p));
    } else if (al->exact) {
        SP_CHECK(ctx->mats.ensure(plan.mat_elems * 4));
        SP_CHECK(launch_zero_scatter<u32>(ctx, sdev, plan.splits, D, dims, rr_ptr(ctx), cc_ptr(ctx, S, D),
                                          al->counts.as<u32>(), ctx->mats.as<u32>()));
        SP_CHECK(launch_gram<u32>(ctx, sdev, ctx->gram_items.as<GramItem>(), (int64_t)plan.gram_items.size(), dims,
                                  ctx->mats.as<u32>(), ctx->grams.as<double>()));
    } else {
        SP_CHECK(ctx->mats.ensure(plan.mat_elems * 8));
        SP_CHECK(launch_zero_scatter<double>(ctx, sdev, plan.splits, D, dims, rr_ptr(ctx), cc_ptr(ctx, S, D),
                                             al->weights.as<double>(), ctx->mats.as<double>()));
        SP_CHECK(launch_gram<double>(ctx, sdev, ctx->gram_items.as<GramItem>(), (int64_t)plan.gram_items.size(), dims,
                                     ctx->mats.as<double>(), ctx->grams.as<double>()));
    }
    if (ctx->opt.direct_all && !g_i32) {   // test switch: every split through the direct solver (fp64 G: the caller planned nl = 0)
        std::vector<int2> hd(S);
        SP_HIP(hipMemcpyAsync(hd.data(), dims, S * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(hipStreamSynchronize(ctx->stream));
        int mmax = 0;
        for (size_t k = 0; k < S; ++k) mmax = std::max(mmax, std::min<int>(hd[k].x, plan.splits[k].rcap));
        SP_CHECK(ctx->eigws.ensure(direct_ws_doubles((int64_t)S, mmax) * 8));
        return launch_direct_top4(ctx, sdev, dims, (int64_t)S, mmax, ctx->grams.as<double>(), ctx->eigws.as<double>(), nullptr,
                                  ctx->scores.as<double>(), ctx->status.as<int>());
    }
    const int* order_dev = reinterpret_cast<const int*>(ctx->gram_items.as<GramItem>() + plan.gram_items.size() + plan.row_items.size());
    if (!ctx->opt.eigen_block16)   // certified 4-wide block, one workgroup per split (heaviest first); what it flags is the direct solver's
        return launch_eigen4(ctx, sdev, plan.splits, dims, ctx->grams.p, g_i32, order_dev, ctx->scores.as<double>(),
                             ctx->status.as<int>());
    SP_CHECK(launch_eigen(ctx, sdev, plan.splits, dims, ctx->grams.p, g_i32,
                          ctx->gram_items.as<GramItem>() + plan.gram_items.size(), (int64_t)plan.row_items.size(), order_dev,
                          ctx->scores.as<double>(), ctx->status.as<int>(), (int64_t)plan.n_row_a, (int64_t)plan.n_order_a));
    return SP_OK;
}

// One generic matrix whose fp64 Gram matrix is in ctx->grams (plan.splits[0], dims on the device): the block iteration
// where it fits the eigen kernels (smaller side <= EIG_MAXR rows) and certifies its sum, the direct solver otherwise -
// a matrix without a spectral gap behind its 4th singular value, or with a longer smaller side (up to direct_max_rows).
static int score_single_gram(sp_ctx* ctx, const Plan& plan, int64_t R) {
    const int64_t max_rows = ctx->opt.direct_max_rows > 0 ? ctx->opt.direct_max_rows : 16384;
    bool direct = ctx->opt.direct_all != 0 || plan.splits[0].rcap > EIG_MAXR;
    SP_REQUIRE(!direct || R <= max_rows, SP_ELIMIT,
               "matrix with a smaller side of %lld rows: the eigen kernels hold %d rows in LDS, the direct solver takes %lld",
               (long long)R, EIG_MAXR, (long long)max_rows);
    if (!direct && !ctx->opt.eigen_block16) {
        SP_CHECK(launch_eigen4(ctx, ctx->splits.as<SplitDev>(), plan.splits, ctx->dims.as<int2>(), ctx->grams.p, false, nullptr,
                               ctx->scores.as<double>(), ctx->status.as<int>()));
        if (!ctx->opt.direct_finish) return SP_OK;
        int st = 0;
        SP_HIP(hipMemcpyAsync(&st, ctx->status.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(hipStreamSynchronize(ctx->stream));
        if (!(st & 3)) return SP_OK;
        direct = true;   // (the kernel only reads G)
    }
    if (!direct) {
        SP_CHECK(launch_eigen(ctx, ctx->splits.as<SplitDev>(), plan.splits, ctx->dims.as<int2>(), ctx->grams.p, false,
                              ctx->gram_items.as<GramItem>() + plan.gram_items.size(), (int64_t)plan.row_items.size(),
                              reinterpret_cast<const int*>(ctx->gram_items.as<GramItem>() + plan.gram_items.size() +
                                                           plan.row_items.size()),
                              ctx->scores.as<double>(), ctx->status.as<int>(), (int64_t)plan.n_row_a, (int64_t)plan.n_order_a));
        if (!ctx->opt.direct_finish) return SP_OK;
        int st = 0;
        SP_HIP(hipMemcpyAsync(&st, ctx->status.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(hipStreamSynchronize(ctx->stream));
        if (!(st & 3)) return SP_OK;
        direct = true;   // (the eigen kernels only read G)
    }
    SP_CHECK(ctx->eigws.ensure(direct_ws_doubles(1, R) * 8));
    return launch_direct_top4(ctx, ctx->splits.as<SplitDev>(), ctx->dims.as<int2>(), 1, (int)R, ctx->grams.as<double>(),
                              ctx->eigws.as<double>(), nullptr, ctx->scores.as<double>(), ctx->status.as<int>());
}

// Rows of the table the sparse kernel sees: D, plus one row per further 65535 of every count >= 2^16 (common.h).  The
// first call for an alignment with such counts copies the counts to the host once.
static int sparse_rows(sp_alignment* al, int64_t* rows) {
    if (al->spk_D < 0) {
        if (!al->exact || al->D == 0 || al->max_count < 65536u) {
            al->spk_D = al->D;
        } else {
            std::vector<u32> hc((size_t)al->D);
            SP_HIP(hipStreamSynchronize(al->ctx->stream));
            SP_HIP(hipMemcpy(hc.data(), al->counts.p, (size_t)al->D * 4, hipMemcpyDeviceToHost));
            int64_t extra = 0;
            for (u32 c : hc) extra += c >= 65536u ? (int64_t)((c + 65534u) / 65535u) - 1 : 0;
            al->spk_D = al->D + extra;
        }
    }
    *rows = al->spk_D;
    return SP_OK;
}

// One-time, per alignment: 32-bit keys, trace, largest counts, and the AlDesc the kernels read - built on `ctx`'s stream
// and SYNCHRONISED, so that afterwards the alignment is immutable and any context (lane) of the device may score it.
static int prepare_sparse_table(sp_ctx* ctx, sp_alignment* al) {
    SP_REQUIRE(al->n_taxa <= 16, SP_ELIMIT, "sparse route: at most 16 taxa");
    int64_t rows = 0;
    SP_CHECK(sparse_rows(al, &rows));
    SP_HIP(hipStreamSynchronize(al->ctx->stream));   // the table itself may still be in flight on its creator's stream
    unsigned long long trace = 0;
    const u64* keys = al->keys.as<u64>();
    const u32* counts = al->counts.as<u32>();
    if (rows != al->D) {   // expand on the host (a few dozen patterns at most carry such counts)
        std::vector<u64> hk((size_t)al->D), ek;
        std::vector<u32> hc((size_t)al->D), ec;
        SP_HIP(hipMemcpy(hk.data(), al->keys.p, (size_t)al->D * 8, hipMemcpyDeviceToHost));
        SP_HIP(hipMemcpy(hc.data(), al->counts.p, (size_t)al->D * 4, hipMemcpyDeviceToHost));
        ek.reserve((size_t)rows);
        ec.reserve((size_t)rows);
        for (int64_t i = 0; i < al->D; ++i) {
            u32 c = hc[i];
            trace += (unsigned long long)c * c;
            do {
                const u32 piece = c > 65535u ? 65535u : c;
                ek.push_back(hk[i]);
                ec.push_back(piece);
                c -= piece;
            } while (c > 0);
        }
        SP_REQUIRE((int64_t)ek.size() == rows, SP_EINVAL, "expanded table has %zu rows, expected %lld", ek.size(),
                   (long long)rows);
        SP_CHECK(al->spk_keys.ensure((size_t)rows * 8));
        SP_CHECK(al->spk_counts.ensure((size_t)rows * 4 + SP_COUNTS_PAD));
        SP_HIP(hipMemcpy(al->spk_keys.p, ek.data(), (size_t)rows * 8, hipMemcpyHostToDevice));
        SP_HIP(hipMemcpy(al->spk_counts.p, ec.data(), (size_t)rows * 4, hipMemcpyHostToDevice));
        keys = al->spk_keys.as<u64>();
        counts = al->spk_counts.as<u32>();
    }
    SP_CHECK(al->keys32.ensure((size_t)std::max<int64_t>(rows, 1) * 4));
    SP_CHECK(al->spk_meta.ensure(sizeof(SpkMeta)));
    SP_CHECK(al->aldesc.ensure(sizeof(AlDesc)));
    SP_CHECK(launch_sparse_meta(ctx, keys, counts, rows, al->keys32.as<u32>(), al->spk_meta.as<SpkMeta>(), trace, al->D));
    const AlDesc d{al->keys32.as<u32>(), counts, al->spk_meta.as<SpkMeta>(), rows};
    SP_HIP(hipMemcpyAsync(al->aldesc.p, &d, sizeof(d), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    al->spk_ready = true;
    return SP_OK;
}

static AlDesc host_aldesc(const sp_alignment* al) {
    return AlDesc{al->keys32.as<u32>(), al->spk_D != al->D ? al->spk_counts.as<u32>() : al->counts.as<u32>(),
                  al->spk_meta.as<SpkMeta>(), al->spk_D};
}

// Descriptor array of a multi-alignment call.  One alignment: the alignment's own immutable device copy.  Several: the
// context's buffer, rewritten on the context's stream only when the set differs from the last call's - in stream order
// behind every kernel that read the previous contents (a context's buffer is only ever used on its own stream).
static int aldescs_for(sp_ctx* ctx, sp_alignment* const* als, int n_al, const AlDesc** out) {
    for (int i = 0; i < n_al; ++i)
        if (!als[i]->spk_ready) SP_CHECK(prepare_sparse_table(ctx, als[i]));
    if (n_al == 1) {
        *out = als[0]->aldesc.as<AlDesc>();
        return SP_OK;
    }
    std::vector<AlDesc> d((size_t)n_al);
    for (int i = 0; i < n_al; ++i) d[i] = host_aldesc(als[i]);
    if (!(ctx->aldescs_host.size() == d.size() && ctx->aldescs.p &&
          memcmp(ctx->aldescs_host.data(), d.data(), d.size() * sizeof(AlDesc)) == 0)) {
        SP_CHECK(ctx->aldescs.ensure(d.size() * sizeof(AlDesc)));
        // (pageable source: the runtime stages it before returning, so the vector may be replaced right away)
        SP_HIP(hipMemcpyAsync(ctx->aldescs.p, d.data(), d.size() * sizeof(AlDesc), hipMemcpyHostToDevice, ctx->stream));
        SP_HIP(hipStreamSynchronize(ctx->stream));
        ctx->aldescs_host = d;
    }
    *out = ctx->aldescs.as<AlDesc>();
    return SP_OK;
}

// ------------------------------------------------------------------ plans ----------------------
extern "C" int sp_plan_create(sp_ctx* ctx, int n_taxa, const int32_t* split_taxa, const int32_t* split_a,
                              int64_t n_splits, sp_plan** out) {
    return sp_guard("sp_plan_create", [&]() -> int {
    SP_REQUIRE(ctx && out && (n_splits == 0 || (split_taxa && split_a)), SP_EINVAL, "sp_plan_create: NULL argument");
    SP_REQUIRE(n_taxa >= 2 && n_taxa <= 32 && n_splits >= 0, SP_EINVAL, "sp_plan_create: bad sizes");
    SP_HIP(hipSetDevice(ctx->device));
    Plan hp;
    SP_CHECK(plan_splits(n_taxa, 1, split_taxa, split_a, n_splits, true, false, false, hp, 0));
    sp_plan* pl = new sp_plan();
    pl->device = ctx->device;
    pl->n = n_taxa;
    pl->S = n_splits;
    pl->splits = hp.splits;
    pl->taxa.assign(split_taxa, split_taxa + (size_t)n_splits * n_taxa);
    pl->a.assign(split_a, split_a + n_splits);
    // heaviest first: the larger the smaller side, the longer the workgroup runs (stable: ties keep the list order).
    // (Round 3 tried the measured per-class cost instead - 4|6 before 5|5 on 10 taxa, 56 against 51 us - and the pipelined
    // benchmark got 0.5 % slower, 0.1011 against 0.1006 ms per step: kept as it was.)
    std::vector<int> order((size_t)n_splits);
    for (int64_t i = 0; i < n_splits; ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return pl->splits[x].nr > pl->splits[y].nr; });
    for (const SplitDev& sd : pl->splits) {
        pl->bm_words_max = std::max<int64_t>(pl->bm_words_max, (int64_t)sd.rw + sd.cw);
        pl->fit.add_split(sd.nr, sd.nc, sd.rw, sd.cw);
    }
    int rc = SP_OK;
    const size_t s1 = (size_t)std::max<int64_t>(n_splits, 1);
    if ((rc = pl->splits_dev.ensure(s1 * sizeof(SplitDev))) || (rc = pl->launch_dev.ensure(s1 * sizeof(SplitDev)))) {
        sp_plan_release(pl);
        return rc;
    }
    if (n_splits > 0) {
        std::vector<SplitDev> launch((size_t)n_splits);
        for (int64_t b = 0; b < n_splits; ++b) {
            launch[b] = pl->splits[order[b]];
            launch[b].cls = order[b];
        }
        hipError_t e = hipMemcpy(pl->splits_dev.p, pl->splits.data(), (size_t)n_splits * sizeof(SplitDev), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(pl->launch_dev.p, launch.data(), (size_t)n_splits * sizeof(SplitDev), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            sp_plan_release(pl);
            sp_set_error("sp_plan_create: upload failed: %s", hipGetErrorString(e));
            return SP_EHIP;
        }
    }
    *out = pl;
    return SP_OK;
    });
}

extern "C" int sp_plan_retain(sp_plan* plan) {
    return sp_guard("sp_plan_retain", [&]() -> int {
    SP_REQUIRE(plan, SP_EINVAL, "plan is NULL");
    ++plan->refs;
    return SP_OK;
    });
}

extern "C" int sp_plan_release(sp_plan* plan) {
    return sp_guard("sp_plan_release", [&]() -> int {
    if (!plan) return SP_OK;
    if (--plan->refs > 0) return SP_OK;
    (void)hipSetDevice(plan->device);
    (void)hipDeviceSynchronize();   // kernels of any lane may still be reading it
    plan->splits_dev.release();
    plan->launch_dev.release();
    delete plan;
    return SP_OK;
    });
}

extern "C" int sp_plan_info(const sp_plan* plan, int* n_taxa, int64_t* n_splits) {
    return sp_guard("sp_plan_info", [&]() -> int {
    SP_REQUIRE(plan, SP_EINVAL, "plan is NULL");
    if (n_taxa) *n_taxa = plan->n;
    if (n_splits) *n_splits = plan->S;
    return SP_OK;
    });
}

// The device-side sparse route for n_al alignments x the plan's splits on `ctx`'s stream (no host synchronisation once
// the alignments are prepared).
static int enqueue_sparse_plan(sp_ctx* ctx, sp_alignment* const* als, int n_al, const sp_plan* plan, double* scores,
                               int* status, bool wide_all) {
    int64_t dmax = 0, dmin = INT64_MAX;
    for (int i = 0; i < n_al; ++i) {
        sp_alignment* al = als[i];
        SP_REQUIRE(al && al->ctx->device == ctx->device && al->n_taxa == plan->n, SP_EINVAL,
                   "alignment %d: NULL, on another device, or not of the plan's %d taxa", i, plan->n);
        int64_t srows = 0;
        SP_CHECK(sparse_rows(al, &srows));
        SP_REQUIRE(al->exact && srows <= 65535 && al->D > 0 && al->n_taxa <= 16, SP_ELIMIT,
                   "sparse route needs integer counts, 1..65535 table rows (counts >= 65536 take several) and at most 16 "
                   "taxa (alignment %d: %lld rows, %d taxa, %s)", i, (long long)srows, al->n_taxa,
                   al->exact ? "counts" : "float weights");
        dmax = std::max(dmax, srows);
        dmin = std::min(dmin, srows);
    }
    const AlDesc* descs = nullptr;
    SP_CHECK(aldescs_for(ctx, als, n_al, &descs));
    SparseFitHint hint = plan->fit;
    hint.d_min = n_al > 0 ? dmin : 0;
    return launch_sparse_chain(ctx, descs, host_aldesc(als[0]), n_al, plan->n, plan->splits_dev.as<SplitDev>(),
                               plan->launch_dev.as<SplitDev>(), plan->S, scores, status, dmax, plan->bm_words_max, wide_all, hint);
}

extern "C" int sp_score_plan_steps(sp_ctx* lane, sp_alignment* const* als, int n_al, sp_plan* plan, int n_steps,
                                   void* scores_dev, int64_t scores_step_bytes, void* status_dev, int64_t status_step_bytes) {
    return sp_guard("sp_score_plan_steps", [&]() -> int {
    SP_REQUIRE(lane && als && n_al >= 1 && plan && scores_dev && status_dev && n_steps >= 0, SP_EINVAL,
               "sp_score_plan_steps: NULL argument or negative step count");
    SP_REQUIRE(plan->device == lane->device, SP_EINVAL, "plan and lane are on different devices");
    const int64_t items = (int64_t)n_al * plan->S;
    SP_REQUIRE(n_steps <= 1 || (scores_step_bytes >= items * 8 && status_step_bytes >= items * 4 && scores_step_bytes % 8 == 0 &&
                                status_step_bytes % 4 == 0),
               SP_EINVAL, "sp_score_plan_steps: step strides of %lld / %lld bytes for %lld items per pass", (long long)scores_step_bytes,
               (long long)status_step_bytes, (long long)items);
    SP_HIP(hipSetDevice(lane->device));
    if (plan->S == 0) return SP_OK;
    for (int s = 0; s < n_steps; ++s)
        SP_CHECK(enqueue_sparse_plan(lane, als, n_al, plan, (double*)((char*)scores_dev + (int64_t)s * scores_step_bytes),
                                     (int*)((char*)status_dev + (int64_t)s * status_step_bytes), true));
    return SP_OK;
    });
}

extern "C" int sp_score_plan_async(sp_ctx* lane, sp_alignment* const* als, int n_al, sp_plan* plan, void* scores_dev,
                                   void* status_dev) {
    return sp_guard("sp_score_plan_async", [&]() -> int {
    SP_REQUIRE(lane && als && n_al >= 1 && plan && scores_dev && status_dev, SP_EINVAL, "sp_score_plan_async: NULL argument");
    SP_REQUIRE(plan->device == lane->device, SP_EINVAL, "plan and lane are on different devices");
    SP_HIP(hipSetDevice(lane->device));
    if (plan->S == 0) return SP_OK;
    return enqueue_sparse_plan(lane, als, n_al, plan, (double*)scores_dev, (int*)status_dev, true);
    });
}

// The context's cached plan for a split list (content-keyed): the asynchronous convenience entry points and the
// synchronous sparse route re-plan only when the list changes.
static int cached_sparse_plan(sp_ctx* ctx, int n, const int32_t* split_taxa, const int32_t* split_a, int64_t S,
                              sp_plan** out) {
    if (!ctx->cache) ctx->cache = new PlanCache();
    sp_plan*& pl = ctx->cache->sparse;
    const size_t nt = (size_t)S * n;
    if (pl && pl->n == n && pl->S == S && memcmp(pl->a.data(), split_a, (size_t)S * 4) == 0 &&
        memcmp(pl->taxa.data(), split_taxa, nt * 4) == 0) {
        *out = pl;
        return SP_OK;
    }
    if (pl) {
        (void)sp_plan_release(pl);
        pl = nullptr;
    }
    SP_CHECK(sp_plan_create(ctx, n, split_taxa, split_a, S, &pl));
    *out = pl;
    return SP_OK;
}

// ------------------------------------------------------------------ direct finisher -----------
// Splits the iterative routes could not certify (status bit 0: budget spent or given up; bit 1: handed back) are scored
// by the direct solver (finish.hip): compact matrix of the split -> fp64 Gram over its smaller side -> Householder
// tridiagonalisation + Sturm multisection.  Host-driven (2 m launches, dims fetched once per chunk); the chunks keep the
// compact matrices + Gram matrices of one batch under ~24 GB.  Splits whose smaller side has more compact rows than
// `direct_max_rows` (default 16384: 2 GB of G, ~23 TB of traffic - about 25 s) or a side of more than 14 taxa (bitmap
// compaction) keep their flagged estimate.  scores_dev / status_dev: the S-entry device arrays to patch; st: their
// host copy (patched too).  Returns the number of splits finished in *n_done.
static int finish_flagged(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t S, std::vector<int>& st,
                          double* scores_dev, int* status_dev, int64_t* n_done) {
    sp_ctx* ctx = al->ctx;
    const int n = al->n_taxa;
    const int64_t D = al->D;
    if (n_done) *n_done = 0;
    if (!ctx->opt.direct_finish || D == 0) return SP_OK;
    const int64_t max_rows = ctx->opt.direct_max_rows > 0 ? ctx->opt.direct_max_rows : 16384;
    const size_t elt = al->exact ? 4 : 8;
    struct Cand { int idx; int64_t rcap, pitch; };
    std::vector<Cand> cand;
    for (int64_t i = 0; i < S; ++i) {
        if (!(st[i] & 3)) continue;
        const int a = split_a[i], b = n - a;
        if (a < 1 || b < 1 || std::max(a, b) > 14) continue;
        const int64_t rmax = std::min<int64_t>(pow4(std::min(a, b)), D), cmax = std::min<int64_t>(pow4(std::max(a, b)), D);
        if (rmax > max_rows) continue;
        cand.push_back({(int)i, round_up(rmax, 64), round_up(cmax, 32)});
    }
    if (cand.empty()) return SP_OK;
    if (ctx->cache) ctx->cache->valid = false;   // the plan pools are overwritten
    const size_t budget = (size_t)24 << 30;
    size_t pos = 0;
    int64_t done = 0;
    while (pos < cand.size()) {
        size_t end = pos, bytes = 0;
        while (end < cand.size() && end - pos < 4096) {
            const size_t need = (size_t)cand[end].rcap * cand[end].pitch * elt + (size_t)cand[end].rcap * cand[end].rcap * 8;
            if (end > pos && bytes + need > budget) break;
            bytes += need;
            ++end;
        }
        const int64_t cnt = (int64_t)(end - pos);
        std::vector<int32_t> t2((size_t)cnt * n), a2((size_t)cnt);
        std::vector<int> idx((size_t)cnt);
        for (int64_t k = 0; k < cnt; ++k) {
            const int i = cand[pos + k].idx;
            memcpy(&t2[(size_t)k * n], split_taxa + (size_t)i * n, (size_t)n * 4);
            a2[k] = split_a[i];
            idx[k] = i;
        }
        Plan plan;
        SP_CHECK(plan_splits(n, D, t2.data(), a2.data(), cnt, true, true, true, plan, 0));
        SP_CHECK(upload_plan(ctx, plan, D));
        SP_CHECK(ctx->grams.ensure(plan.g_elems * 8));
        SP_CHECK(ctx->mats.ensure(plan.mat_elems * elt));
        SP_CHECK(ctx->misc2.ensure((size_t)cnt * 4));
        const SplitDev* sdev = ctx->splits.as<SplitDev>();
        int2* dims = ctx->dims.as<int2>();
        SP_HIP(hipMemcpyAsync(ctx->misc2.p, idx.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, ctx->stream));
        SP_CHECK(launch_reindex(ctx, al->keys.as<u64>(), D, n, sdev, plan.splits, bm_ptr(ctx), pf_ptr(ctx, plan), dims,
                                rr_ptr(ctx), cc_ptr(ctx, (size_t)cnt, D)));
        if (al->exact) {
            SP_CHECK(launch_zero_scatter<u32>(ctx, sdev, plan.splits, D, dims, rr_ptr(ctx), cc_ptr(ctx, (size_t)cnt, D),
                                              al->counts.as<u32>(), ctx->mats.as<u32>()));
            SP_CHECK(launch_gram<u32>(ctx, sdev, ctx->gram_items.as<GramItem>(), (int64_t)plan.gram_items.size(), dims,
                                      ctx->mats.as<u32>(), ctx->grams.as<double>()));
        } else {
            SP_CHECK(launch_zero_scatter<double>(ctx, sdev, plan.splits, D, dims, rr_ptr(ctx), cc_ptr(ctx, (size_t)cnt, D),
                                                 al->weights.as<double>(), ctx->mats.as<double>()));
            SP_CHECK(launch_gram<double>(ctx, sdev, ctx->gram_items.as<GramItem>(), (int64_t)plan.gram_items.size(), dims,
                                         ctx->mats.as<double>(), ctx->grams.as<double>()));
        }
        std::vector<int2> hd((size_t)cnt);
        SP_HIP(hipMemcpyAsync(hd.data(), dims, (size_t)cnt * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(hipStreamSynchronize(ctx->stream));   // (idx / t2 / a2 die with this iteration)
        int mmax = 0;
        for (int64_t k = 0; k < cnt; ++k) mmax = std::max(mmax, std::min<int>(hd[k].x, plan.splits[k].rcap));
        SP_CHECK(ctx->eigws.ensure(direct_ws_doubles(cnt, mmax) * 8));
        SP_CHECK(launch_direct_top4(ctx, sdev, dims, cnt, mmax, ctx->grams.as<double>(), ctx->eigws.as<double>(),
                                    ctx->misc2.as<int>(), scores_dev, status_dev));
        SP_HIP(hipStreamSynchronize(ctx->stream));
        for (int64_t k = 0; k < cnt; ++k) st[idx[k]] = 4 | (std::min<int>(hd[k].x, 0x7FFFFF) << 8);
        done += cnt;
        pos = end;
    }
    if (n_done) *n_done = done;
    return SP_OK;
}

// Synchronous sparse route: the device chain (in-LDS kernel -> lists in global memory -> all arrays in global memory ->
// 8-wide fallback block); whatever leaves it flagged is the direct solver's (finish_flagged, at the caller).
static int run_sparse_route(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t S) {
    sp_ctx* ctx = al->ctx;
    sp_plan* plan = nullptr;
    SP_CHECK(cached_sparse_plan(ctx, al->n_taxa, split_taxa, split_a, S, &plan));
    SP_CHECK(ctx->scores.ensure((size_t)S * 8));
    SP_CHECK(ctx->status.ensure((size_t)S * 4));
    return enqueue_sparse_plan(ctx, &al, 1, plan, ctx->scores.as<double>(), ctx->status.as<int>(), true);
}

// ---- every split of the taxa, flattening + score, planned on the device ---------------------------------------------
// subflat.hip's enumeration gives (taxa, a) of every split in all_splits order; k_plan_enumerated turns them into the
// sparse kernel's split descriptors (smaller side = rows) and the heaviest-first launch order, on the device: no split
// list, no plan crosses the boundary.
int enumerate_all_splits(sp_ctx* ctx, int n, int trivial, int size, bool enumerate, int shard_rank, int shard_world,
                         int64_t* total_out, const int8_t** dtaxa_out, const int** da_out, std::vector<int>& sizes,
                         std::vector<unsigned long long>& counts);   // subflat.hip

struct ClassLayout {
    int n_classes;
    int start[17];     // first split of every size class in all_splits order (classes ascending), then the total
    int out_start[17]; // first position of the class in the launch order (largest class first)
};

__global__ __launch_bounds__(256) void k_plan_enumerated(int n, int total, const int8_t* __restrict__ taxa,
                                                         const int* __restrict__ a_arr, ClassLayout cl,
                                                         SplitDev* __restrict__ out, SplitDev* __restrict__ launch) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int a = a_arr[i], b = n - a;
    const bool swap = a > b;
    SplitDev sd;
    memset(&sd, 0, sizeof(sd));
    sd.nr = swap ? b : a;
    sd.nc = swap ? a : b;
    const int8_t* t = taxa + (size_t)i * n;
    for (int k = 0; k < sd.nr; ++k) sd.taxa[k] = swap ? t[a + k] : t[k];
    for (int k = 0; k < sd.nc; ++k) sd.taxa[sd.nr + k] = swap ? t[k] : t[a + k];
    sd.rw = (int)(((1ll << (2 * sd.nr)) + 63) / 64);
    sd.cw = (int)(((1ll << (2 * sd.nc)) + 63) / 64);
    out[i] = sd;
    int q = 0;
    while (q + 1 < cl.n_classes && i >= cl.start[q + 1]) ++q;
    sd.cls = i;
    launch[cl.out_start[q] + (i - cl.start[q])] = sd;
}

static int run_flat_all_splits(sp_alignment* al, int method, int trivial, int size, int shard_rank, int shard_world,
                               int64_t* n_out, bool score) {
    sp_ctx* ctx = al->ctx;
    const int n = al->n_taxa;
    const int8_t* dtaxa = nullptr;
    const int* da = nullptr;
    std::vector<int> sizes;
    std::vector<unsigned long long> counts;
    int64_t total = 0;
    SP_CHECK(enumerate_all_splits(ctx, n, trivial, size, score, shard_rank, shard_world, &total, &dtaxa, &da, sizes, counts));
    if (n_out) *n_out = total;
    if (!score || total == 0) return SP_OK;
    int64_t srows = 0;
    SP_CHECK(sparse_rows(al, &srows));
    const bool sparse_ok = al->exact && srows <= 65535 && n <= 16 && al->D > 0;
    const bool use_sparse = (method == SP_METHOD_FLATTENING_SPARSE || (method == SP_METHOD_FLATTENING && ctx->gram_mode == 0 &&
                                                                       !ctx->opt.force_big)) && sparse_ok;
    // the split list on the host: only for the routes planned there, or for the (rare) dense-route hand-back
    std::vector<int32_t> h_taxa, h_a;
    auto fetch_list = [&]() -> int {
        std::vector<int8_t> t8((size_t)total * n);
        h_a.resize((size_t)total);
        SP_HIP(hipMemcpyAsync(t8.data(), dtaxa, t8.size(), hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(hipMemcpyAsync(h_a.data(), da, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(hipStreamSynchronize(ctx->stream));
        h_taxa.assign(t8.begin(), t8.end());
        return SP_OK;
    };
    if (!use_sparse) {
        SP_CHECK(fetch_list());
        const int rc = sp_score_splits(al, h_taxa.data(), h_a.data(), total, method, nullptr, nullptr, nullptr);
        return rc == SP_ENOCONV ? SP_OK : rc;   // (scores and status are in the context's buffers either way)
    }
    SP_REQUIRE(sizes.size() <= 16, SP_ELIMIT, "%zu size classes", sizes.size());
    ClassLayout cl{};
    cl.n_classes = (int)sizes.size();
    int run = 0;
    for (int q = 0; q < cl.n_classes; ++q) {
        cl.start[q] = run;
        run += (int)counts[q];
    }
    cl.start[cl.n_classes] = run;
    int pos = 0;
    int64_t bmw = 0;
    for (int q = cl.n_classes - 1; q >= 0; --q) {   // largest smaller side first
        cl.out_start[q] = pos;
        pos += (int)counts[q];
        bmw = std::max<int64_t>(bmw, (pow4(sizes[q]) + 63) / 64 + (pow4(n - sizes[q]) + 63) / 64);
    }
    SP_CHECK(ctx->splits.ensure((size_t)total * sizeof(SplitDev)));
    SP_CHECK(ctx->splits_launch.ensure((size_t)total * sizeof(SplitDev)));
    SP_CHECK(ctx->scores.ensure((size_t)total * 8));
    SP_CHECK(ctx->status.ensure((size_t)total * 4));
    if (ctx->cache) ctx->cache->valid = false;   // ctx->splits is rewritten
    hipLaunchKernelGGL(k_plan_enumerated, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, n, (int)total, dtaxa,
                       da, cl, ctx->splits.as<SplitDev>(), ctx->splits_launch.as<SplitDev>());
    SP_HIP(hipGetLastError());
    const AlDesc* descs = nullptr;
    sp_alignment* als1[1] = {al};
    SP_CHECK(aldescs_for(ctx, als1, 1, &descs));
    SP_CHECK(launch_sparse_chain(ctx, descs, host_aldesc(al), 1, n, ctx->splits.as<SplitDev>(),
                                 ctx->splits_launch.as<SplitDev>(), total, ctx->scores.as<double>(), ctx->status.as<int>(),
                                 srows, bmw, true));
    if (!ctx->opt.direct_finish) return SP_OK;
    std::vector<int> st((size_t)total);
    SP_HIP(hipMemcpyAsync(st.data(), ctx->status.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    bool any = false;
    for (int v : st) any |= (v & 3) != 0;
    if (!any) return SP_OK;
    SP_CHECK(fetch_list());   // (rare: the split list crosses the boundary only when the direct solver is needed)
    return finish_flagged(al, h_taxa.data(), h_a.data(), total, st, ctx->scores.as<double>(), ctx->status.as<int>(), nullptr);
}

int run_subflat_route(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t S);  // subflat.hip

// Big-table form of the sparse route (sparse.hip: k_sparse_big): count tables beyond the 65535 rows of the list kernels, for
// the taxon counts the dense route cannot take (12+).  Splits go in chunks so that one segmented sort stays below 2^32
// entries and ~6 GB of work buffers.
static int run_sparse_big_route(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t S) {
    sp_ctx* ctx = al->ctx;
    const int64_t D = al->D;
    const int n = al->n_taxa;
    if (ctx->cache) ctx->cache->valid = false;   // the plan pools are overwritten
    SP_CHECK(ctx->scores.ensure((size_t)S * 8));
    SP_CHECK(ctx->status.ensure((size_t)S * 4));
    int max_side = 0;
    for (int64_t s = 0; s < S; ++s) {
        SP_REQUIRE(split_a[s] >= 1 && split_a[s] < n, SP_EINVAL, "split %lld: side sizes %d | %d", (long long)s, split_a[s],
                   n - split_a[s]);
        max_side = std::max(max_side, std::max(split_a[s], n - split_a[s]));
    }
    const bool by_keys = max_side > 14 || ctx->opt.big_by_keys;   // beyond the bitmap compaction
    const int64_t cap_entries = std::min<int64_t>(((int64_t)1 << 32) - 1, (int64_t)6e9 / (by_keys ? 96 : 64));
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(S, cap_entries / std::max<int64_t>(D, 1)));
    for (int64_t s0 = 0; s0 < S; s0 += chunk) {
        const int64_t cnt = std::min(chunk, S - s0);
        if (by_keys) {
            for (int64_t s = s0; s < s0 + cnt; ++s)
                SP_CHECK(check_split(n, split_taxa + s * n, split_a[s], split_taxa + s * n + split_a[s], n - split_a[s]));
            SP_CHECK(launch_sparse_big_keys(ctx, al->keys.as<u64>(), D, n, split_taxa + s0 * n, split_a + s0, cnt,
                                            al->exact ? al->counts.as<u32>() : nullptr, al->weights.as<double>(), ctx->n_cu,
                                            ctx->scores.as<double>() + s0, ctx->status.as<int>() + s0));
            continue;
        }
        Plan plan;
        SP_CHECK(plan_splits(n, D, split_taxa + s0 * n, split_a + s0, cnt, false, false, false, plan));
        SP_CHECK(upload_plan(ctx, plan, D));
        SP_CHECK(launch_reindex(ctx, al->keys.as<u64>(), D, n, ctx->splits.as<SplitDev>(), plan.splits, bm_ptr(ctx),
                                pf_ptr(ctx, plan), ctx->dims.as<int2>(), rr_ptr(ctx), cc_ptr(ctx, (size_t)cnt, D)));
        SP_CHECK(launch_sparse_big(ctx, D, cnt, rr_ptr(ctx), cc_ptr(ctx, (size_t)cnt, D),
                                   al->exact ? al->counts.as<u32>() : nullptr, al->weights.as<double>(),
                                   ctx->dims.as<int2>(), ctx->n_cu, ctx->scores.as<double>() + s0,
                                   ctx->status.as<int>() + s0));
    }
    return SP_OK;
}

// Mutual-information route (divergence.hip): reindex (compact coordinates per split), marginals, sum.
static int run_divergence_route(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t S) {
    sp_ctx* ctx = al->ctx;
    const int64_t D = al->D;
    if (ctx->cache) ctx->cache->valid = false;   // the plan pools are overwritten
    Plan plan;
    SP_CHECK(plan_splits(al->n_taxa, D, split_taxa, split_a, S, false, false, false, plan));
    SP_CHECK(upload_plan(ctx, plan, D));
    SP_CHECK(ctx->scores.ensure((size_t)S * 8));
    SP_CHECK(ctx->status.ensure((size_t)S * 4));
    SP_CHECK(ctx->misc.ensure((size_t)S * 2 * (size_t)D * 8));
    SP_CHECK(launch_reindex(ctx, al->keys.as<u64>(), D, al->n_taxa, ctx->splits.as<SplitDev>(), plan.splits, bm_ptr(ctx),
                            pf_ptr(ctx, plan), ctx->dims.as<int2>(), rr_ptr(ctx), cc_ptr(ctx, S, D)));
    SP_CHECK(launch_divergence(ctx, al->exact, D, S, rr_ptr(ctx), cc_ptr(ctx, S, D), al->counts.as<u32>(),
                               al->weights.as<double>(), (double)al->N, ctx->misc.as<unsigned long long>(),
                               ctx->scores.as<double>()));
    SP_HIP(hipMemsetAsync(ctx->status.p, 0, (size_t)S * 4, ctx->stream));
    return SP_OK;
}

extern "C" int sp_score_splits(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t n_splits,
                               int method, double* scores_host, void* scores_dev, int32_t* status_host) {
    return sp_guard("sp_score_splits", [&]() -> int {
    SP_REQUIRE(al && split_taxa && split_a, SP_EINVAL, "NULL argument");
    SP_REQUIRE(n_splits >= 0, SP_EINVAL, "n_splits < 0");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    if (n_splits == 0) return SP_OK;
    SP_REQUIRE(al->D > 0, SP_EINVAL, "empty pattern table");
    // tables beyond the list kernels' 65535 rows at taxon counts the dense route cannot take: the big-table form
    bool handled = false;
    if (method == SP_METHOD_FLATTENING) {
        bool want = ctx->opt.force_big != 0;   // (test switch: every table)
        if (!want && al->n_taxa >= 12) {
            if (!al->exact || al->n_taxa > 16) {   // (more than 16 taxa: the list kernels pack keys into 32 bits)
                want = true;            // float weights: the list kernels carry integer counts, the dense route ends at 11 taxa
            } else {
                int64_t srows = 0;
                SP_CHECK(sparse_rows(al, &srows));
                want = srows > 65535;
            }
        }
        if (want) {
            SP_CHECK(run_sparse_big_route(al, split_taxa, split_a, n_splits));
            handled = true;
        }
    }
    if (handled) {
        // (scores and status are in the context's buffers, copied out below)
    } else if (method == SP_METHOD_FLATTENING || method == SP_METHOD_FLATTENING_DENSE ||
        method == SP_METHOD_FLATTENING_SPARSE) {
        int64_t srows = 0;
        SP_CHECK(sparse_rows(al, &srows));
        const bool sparse_ok = al->exact && srows <= 65535 && al->n_taxa <= 16;
        SP_REQUIRE(method != SP_METHOD_FLATTENING_SPARSE || sparse_ok, SP_ELIMIT,
                   "sparse route needs integer counts, at most 65535 table rows (counts >= 65536 take several) and at most 16 taxa");
        const bool use_sparse = method == SP_METHOD_FLATTENING_SPARSE ||
                                (method == SP_METHOD_FLATTENING && sparse_ok && ctx->gram_mode == 0);
        if (use_sparse) {
            SP_CHECK(run_sparse_route(al, split_taxa, split_a, n_splits));
        } else {
            if (!ctx->cache) ctx->cache = new PlanCache();
            PlanCache& pc = *ctx->cache;
            const size_t nt = (size_t)n_splits * al->n_taxa;
            const int nl = ctx->opt.direct_all ? 0 : limbs_for(al);   // (the direct solver reads an fp64 G)
            const bool hit = pc.valid && pc.nl == nl && pc.n == al->n_taxa && pc.D == al->D &&
                             pc.a.size() == (size_t)n_splits &&
                             memcmp(pc.a.data(), split_a, n_splits * 4) == 0 &&
                             memcmp(pc.taxa.data(), split_taxa, nt * 4) == 0;
            if (!hit) {
                pc.valid = false;
                SP_CHECK(plan_splits(al->n_taxa, al->D, split_taxa, split_a, n_splits, true, true, true, pc.plan,
                                     nl > 0 ? nl : 0));
                pc.taxa.assign(split_taxa, split_taxa + nt);
                pc.a.assign(split_a, split_a + n_splits);
                pc.n = al->n_taxa;
                pc.D = al->D;
                pc.nl = nl;
            }
            SP_CHECK(run_dense_route(al, pc.plan, hit));
            pc.valid = true;
        }
    } else if (method == SP_METHOD_SUBFLATTENING) {
        SP_CHECK(run_subflat_route(al, split_taxa, split_a, n_splits));
    } else if (method == SP_METHOD_MUTUAL_INFORMATION) {
        SP_CHECK(run_divergence_route(al, split_taxa, split_a, n_splits));
    } else {
        sp_set_error("unknown method %d", method);
        return SP_EINVAL;
    }
    // Results out.  Scores and status words travel to the host in ONE batch of copies behind one synchronisation; only when
    // a status word is flagged (bit 0 / 1: an iterative flattening route found no certificate) does the direct solver run
    // (status bit 2 afterwards) and the patched results are copied again.
    const bool flat_method = method == SP_METHOD_FLATTENING || method == SP_METHOD_FLATTENING_DENSE || method == SP_METHOD_FLATTENING_SPARSE;
    const bool may_finish = flat_method && ctx->opt.direct_finish != 0 && !ctx->async_results;
    std::vector<int32_t> st_tmp;
    int32_t* st_out = status_host;
    if ((scores_host || may_finish) && !status_host) {   // the return code reports unconverged splits either way
        st_tmp.resize((size_t)n_splits);
        st_out = st_tmp.data();
    }
    if (scores_host) SP_HIP(hipMemcpyAsync(scores_host, ctx->scores.p, n_splits * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (st_out) SP_HIP(hipMemcpyAsync(st_out, ctx->status.p, n_splits * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (scores_host || st_out) SP_HIP(hipStreamSynchronize(ctx->stream));
    if (may_finish && st_out) {
        bool any = false;
        for (int64_t i = 0; i < n_splits && !any; ++i) any = (st_out[i] & 3) != 0;
        if (any) {
            std::vector<int> st(st_out, st_out + n_splits);
            int64_t done = 0;
            SP_CHECK(finish_flagged(al, split_taxa, split_a, n_splits, st, ctx->scores.as<double>(), ctx->status.as<int>(), &done));
            if (done) {
                if (scores_host) SP_HIP(hipMemcpyAsync(scores_host, ctx->scores.p, n_splits * 8, hipMemcpyDeviceToHost, ctx->stream));
                SP_HIP(hipMemcpyAsync(st_out, ctx->status.p, n_splits * 4, hipMemcpyDeviceToHost, ctx->stream));
                SP_HIP(hipStreamSynchronize(ctx->stream));
            }
        }
    }
    if (scores_dev)
        SP_HIP(hipMemcpyAsync(scores_dev, ctx->scores.p, n_splits * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (st_out) {
        int64_t bad = 0, first = -1;
        for (int64_t i = 0; i < n_splits; ++i)
            if (st_out[i] & 3) {
                if (first < 0) first = i;
                ++bad;
            }
        if (bad) {
            sp_set_error("%lld of %lld splits left their eigen-solver without a certificate (first: split %lld) and are beyond "
                         "the direct solver's limits or it is switched off: their scores are upper estimates (status bit 0 / 1)",
                         (long long)bad, (long long)n_splits, (long long)first);
            return SP_ENOCONV;
        }
    }
    return SP_OK;
    });
}

// ABI 4: the host step behind the asynchronous entry points.  scores_host / status_host hold the fetched results of an
// asynchronous pass over (al, these splits); every split whose status word has bit 0 or bit 1 set is re-scored by the
// direct solver and patched in place (status then: bit 2 set, bits 8.. = rows of the solved Gram matrix).
extern "C" int sp_finish_flagged(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t n_splits,
                                 double* scores_host, int32_t* status_host, int64_t* n_finished) {
    return sp_guard("sp_finish_flagged", [&]() -> int {
    SP_REQUIRE(al && split_taxa && split_a && scores_host && status_host, SP_EINVAL, "NULL argument");
    SP_REQUIRE(n_splits >= 0, SP_EINVAL, "n_splits < 0");
    if (n_finished) *n_finished = 0;
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    std::vector<int> st(status_host, status_host + n_splits);
    bool any = false;
    for (int v : st) any |= (v & 3) != 0;
    if (!any) return SP_OK;
    for (int64_t s = 0; s < n_splits; ++s)
        if (st[s] & 3)
            SP_CHECK(check_split(al->n_taxa, split_taxa + s * al->n_taxa, split_a[s], split_taxa + s * al->n_taxa + split_a[s],
                                 al->n_taxa - split_a[s]));
    SP_CHECK(ctx->scores.ensure((size_t)n_splits * 8));
    SP_CHECK(ctx->status.ensure((size_t)n_splits * 4));
    SP_HIP(hipMemcpyAsync(ctx->scores.p, scores_host, (size_t)n_splits * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(ctx->status.p, status_host, (size_t)n_splits * 4, hipMemcpyHostToDevice, ctx->stream));
    int64_t done = 0;
    SP_CHECK(finish_flagged(al, split_taxa, split_a, n_splits, st, ctx->scores.as<double>(), ctx->status.as<int>(), &done));
    SP_HIP(hipMemcpyAsync(scores_host, ctx->scores.p, (size_t)n_splits * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipMemcpyAsync(status_host, ctx->status.p, (size_t)n_splits * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    if (n_finished) *n_finished = done;
    return SP_OK;
    });
}

int run_subflat_all_splits(sp_alignment* al, int trivial, int size, int shard_rank, int shard_world, int64_t* n_out,
                           bool score, double* scores_out, int* status_out);  // subflat.hip

// Every split of the table's taxa in the reference's all_splits order (splits.py:39-59), enumerated on the device.
extern "C" int sp_score_all_splits_shard(sp_alignment* al, int method, int trivial, int size, int shard_rank,
                                         int shard_world, int64_t* n_splits, double* scores_host, void* scores_dev,
                                         int32_t* status_host, void* status_dev) {
    return sp_guard("sp_score_all_splits_shard", [&]() -> int {
    SP_REQUIRE(al, SP_EINVAL, "alignment is NULL");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    const bool score = scores_host || scores_dev || status_host || status_dev;
    SP_REQUIRE(!score || al->D > 0, SP_EINVAL, "empty pattern table");
    int64_t n = 0;
    // where the scores / status words are on the device when the route returns (the subflattening kernel writes straight
    // into the caller's device buffers; the other routes leave them in the context's and they are copied below)
    const void* sc_at = nullptr;
    const void* st_at = nullptr;
    if (method == SP_METHOD_SUBFLATTENING) {
        SP_CHECK(run_subflat_all_splits(al, trivial, size, shard_rank, shard_world, &n, score,
                                        static_cast<double*>(scores_dev), static_cast<int*>(status_dev)));
        sc_at = scores_dev;
        st_at = status_dev;
    } else {
        SP_REQUIRE(method == SP_METHOD_FLATTENING || method == SP_METHOD_FLATTENING_DENSE ||
                   method == SP_METHOD_FLATTENING_SPARSE || method == SP_METHOD_MUTUAL_INFORMATION, SP_EINVAL,
                   "unknown method %d", method);
        SP_CHECK(run_flat_all_splits(al, method, trivial, size, shard_rank, shard_world, &n, score));
    }
    if (n_splits) *n_splits = n;
    if (!score || n == 0) return SP_OK;
    if (!sc_at) {
        sc_at = ctx->scores.p;
        if (scores_dev) SP_HIP(hipMemcpyAsync(scores_dev, sc_at, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (!st_at) {
        st_at = ctx->status.p;
        if (status_dev) SP_HIP(hipMemcpyAsync(status_dev, st_at, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (scores_host) SP_HIP(hipMemcpyAsync(scores_host, sc_at, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (status_host) SP_HIP(hipMemcpyAsync(status_host, st_at, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (scores_host || status_host) SP_HIP(hipStreamSynchronize(ctx->stream));
    if (status_host) {
        int64_t bad = 0;
        for (int64_t i = 0; i < n; ++i) bad += status_host[i] & 1;
        if (bad) {
            sp_set_error("%lld of %lld splits hit the iteration cap of their eigen-solver: their scores are upper estimates "
                         "(status bit 0)", (long long)bad, (long long)n);
            return SP_ENOCONV;
        }
    }
    return SP_OK;
    });
}

extern "C" int sp_score_all_splits(sp_alignment* al, int method, int trivial, int size, int64_t* n_splits,
                                   double* scores_host, void* scores_dev, int32_t* status_host) {
    return sp_score_all_splits_shard(al, method, trivial, size, 0, 1, n_splits, scores_host, scores_dev, status_host, nullptr);
}

// Generic matrix: upload (transposed if needed so the smaller side indexes rows), Gram, eigen.
extern "C" int sp_score_matrix_f64(sp_ctx* ctx, const double* m, int64_t rows, int64_t cols, int64_t ld,
                                   double* score) {
    return sp_guard("sp_score_matrix_f64", [&]() -> int {
    SP_REQUIRE(ctx && m && score, SP_EINVAL, "NULL argument");
    SP_REQUIRE(rows >= 1 && cols >= 1 && ld >= cols, SP_EINVAL, "bad matrix shape %lld x %lld (ld %lld)",
               (long long)rows, (long long)cols, (long long)ld);
    SP_HIP(hipSetDevice(ctx->device));
    const bool tr = rows > cols;
    const int64_t R = tr ? cols : rows, K = tr ? rows : cols;
    if (R <= 4) {
        // reference: 1 - x/x = 0 exactly (nan for the all-zero matrix)
        bool any = false;
        for (int64_t i = 0; i < rows && !any; ++i)
            for (int64_t j = 0; j < cols; ++j)
                if (m[i * ld + j] != 0) {
                    any = true;
                    break;
                }
        *score = any ? 0.0 : NAN;
        return SP_OK;
    }
    Plan plan;
    plan.splits.resize(1);
    SplitDev& sd = plan.splits[0];
    memset(&sd, 0, sizeof(sd));
    sd.rcap = (int32_t)round_up(R, 64);
    sd.pitch = (int32_t)round_up(K, 32);
    sd.g_pitch = sd.rcap;
    SP_CHECK(ctx->splits.ensure(sizeof(SplitDev)));
    SP_CHECK(ctx->dims.ensure(sizeof(int2)));
    SP_CHECK(ctx->mats.ensure((size_t)sd.rcap * sd.pitch * 8));
    SP_CHECK(ctx->grams.ensure((size_t)sd.rcap * sd.rcap * 8));
    SP_CHECK(ctx->scores.ensure(8));
    SP_CHECK(ctx->status.ensure(4));
    // pack on the host into the padded (R_pad x K_pad) layout, transposing if needed
    std::vector<double> packed((size_t)sd.rcap * sd.pitch, 0.0);
    if (!tr) {
        for (int64_t i = 0; i < rows; ++i) memcpy(&packed[(size_t)i * sd.pitch], m + i * ld, cols * 8);
    } else {
        for (int64_t i = 0; i < rows; ++i)
            for (int64_t j = 0; j < cols; ++j) packed[(size_t)j * sd.pitch + i] = m[i * ld + j];
    }
    const int2 d = make_int2((int)R, (int)K);
    SP_HIP(hipMemcpyAsync(ctx->mats.p, packed.data(), packed.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(ctx->dims.p, &d, sizeof(d), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->cache) ctx->cache->valid = false;
    SP_HIP(hipMemcpyAsync(ctx->splits.p, &sd, sizeof(sd), hipMemcpyHostToDevice, ctx->stream));
    build_gram_items(plan);
    SP_CHECK(upload_items(ctx, plan));
    SP_CHECK(launch_gram<double>(ctx, ctx->splits.as<SplitDev>(), ctx->gram_items.as<GramItem>(),
                                 (int64_t)plan.gram_items.size(), ctx->dims.as<int2>(), ctx->mats.as<double>(),
                                 ctx->grams.as<double>()));
    SP_CHECK(score_single_gram(ctx, plan, R));
    SP_HIP(hipMemcpyAsync(score, ctx->scores.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    return SP_OK;
    });
}

// Sparse (COO) matrix: phylogenetics.py:303-312.  The reference asks ARPACK for the top-4 singular
// values and divides by the Frobenius norm; here the non-empty rows / columns are compacted on the
// device, the compact matrix is scattered densely and scored through the same Gram + eigen kernels
// (removing all-zero rows / columns does not change singular values).
extern "C" int sp_score_coo_f64(sp_ctx* ctx, const int64_t* ri, const int64_t* ci, const double* v, int64_t nnz,
                                int64_t rows, int64_t cols, double* score) {
    return sp_guard("sp_score_coo_f64", [&]() -> int {
    SP_REQUIRE(ctx && score, SP_EINVAL, "NULL argument");
    SP_REQUIRE(rows >= 1 && cols >= 1 && nnz >= 0, SP_EINVAL, "bad shape");
    SP_REQUIRE(nnz == 0 || (ri && ci && v), SP_EINVAL, "NULL triplet arrays");
    SP_HIP(hipSetDevice(ctx->device));
    if (nnz == 0) {
        *score = NAN;  // reference: 0/0
        return SP_OK;
    }
    SP_REQUIRE(rows <= ((int64_t)1 << 32) && cols <= ((int64_t)1 << 32), SP_ELIMIT,
               "sparse matrix side %lld exceeds the 2^32 supported by the bitmap compaction",
               (long long)std::max(rows, cols));
    bool any = false;
    for (int64_t i = 0; i < nnz; ++i) {
        SP_REQUIRE(ri[i] >= 0 && ri[i] < rows && ci[i] >= 0 && ci[i] < cols, SP_EINVAL, "triplet %lld out of range",
                   (long long)i);
        any |= v[i] != 0;
    }
    if (!any) {
        *score = NAN;
        return SP_OK;
    }
    // orient so that the side with fewer distinct indices can become the rows: unknown before
    // compaction, so compact first with rows = given rows, then transpose logically if needed.
    Plan plan;
    plan.splits.resize(1);
    SplitDev& sd = plan.splits[0];
    memset(&sd, 0, sizeof(sd));
    sd.rw = (int32_t)((rows + 63) / 64);
    sd.cw = (int32_t)((cols + 63) / 64);
    plan.bm_words = plan.pf_words = (size_t)sd.rw + sd.cw;
    SP_CHECK(ctx->splits.ensure(sizeof(SplitDev)));
    SP_CHECK(ctx->bitmaps.ensure(plan.bm_words * 12 + 64));
    SP_CHECK(ctx->coords.ensure((size_t)nnz * 8));
    SP_CHECK(ctx->dims.ensure(sizeof(int2)));
    SP_CHECK(ctx->misc.ensure((size_t)nnz * 24));
    int64_t* dri = ctx->misc.as<int64_t>();
    int64_t* dci = dri + nnz;
    double* dv = reinterpret_cast<double*>(dci + nnz);
    SP_HIP(hipMemcpyAsync(dri, ri, nnz * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(dci, ci, nnz * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(dv, v, nnz * 8, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->cache) ctx->cache->valid = false;
    SP_HIP(hipMemcpyAsync(ctx->splits.p, &sd, sizeof(sd), hipMemcpyHostToDevice, ctx->stream));
    u32* rr = rr_ptr(ctx);
    u32* cc = rr + nnz;
    SP_CHECK(launch_reindex_coo(ctx, dri, dci, nnz, ctx->splits.as<SplitDev>(), bm_ptr(ctx), pf_ptr(ctx, plan),
                                ctx->dims.as<int2>(), rr, cc));
    int2 d;
    SP_HIP(hipMemcpyAsync(&d, ctx->dims.p, sizeof(d), hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    const bool tr = d.x > d.y;
    const int64_t R = tr ? d.y : d.x, K = tr ? d.x : d.y;
    if (R <= 4) {
        *score = 0.0;
        return SP_OK;
    }
    sd.rcap = (int32_t)round_up(R, 64);
    sd.pitch = (int32_t)round_up(K, 32);
    sd.g_pitch = sd.rcap;
    const int2 d2 = make_int2((int)R, (int)K);
    SP_CHECK(ctx->mats.ensure((size_t)sd.rcap * sd.pitch * 8));
    SP_CHECK(ctx->grams.ensure((size_t)sd.rcap * sd.rcap * 8));
    SP_CHECK(ctx->scores.ensure(8));
    SP_CHECK(ctx->status.ensure(4));
    if (ctx->cache) ctx->cache->valid = false;
    SP_HIP(hipMemcpyAsync(ctx->splits.p, &sd, sizeof(sd), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(hipMemcpyAsync(ctx->dims.p, &d2, sizeof(d2), hipMemcpyHostToDevice, ctx->stream));
    SP_CHECK(launch_zero_scatter<double>(ctx, ctx->splits.as<SplitDev>(), plan.splits, nnz, ctx->dims.as<int2>(),
                                         tr ? cc : rr, tr ? rr : cc, dv, ctx->mats.as<double>()));
    build_gram_items(plan);
    SP_CHECK(upload_items(ctx, plan));
    SP_CHECK(launch_gram<double>(ctx, ctx->splits.as<SplitDev>(), ctx->gram_items.as<GramItem>(),
                                 (int64_t)plan.gram_items.size(), ctx->dims.as<int2>(), ctx->mats.as<double>(),
                                 ctx->grams.as<double>()));
    SP_CHECK(score_single_gram(ctx, plan, R));
    SP_HIP(hipMemcpyAsync(score, ctx->scores.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    return SP_OK;
    });
}

extern "C" int sp_score_splits_async(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a,
                                     int64_t n_splits, int method, void* scores_dev, void* status_dev) {
    return sp_guard("sp_score_splits_async", [&]() -> int {
    SP_REQUIRE(al && split_taxa && split_a && scores_dev && status_dev, SP_EINVAL, "NULL argument");
    SP_REQUIRE(n_splits >= 0, SP_EINVAL, "n_splits < 0");
    sp_ctx* ctx = al->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    if (n_splits == 0) return SP_OK;
    SP_REQUIRE(al->D > 0, SP_EINVAL, "empty pattern table");
    int64_t srows = 0;
    SP_CHECK(sparse_rows(al, &srows));
    const bool sparse_ok = al->exact && srows <= 65535 && al->n_taxa <= 16;
    const bool want_sparse = method == SP_METHOD_FLATTENING_SPARSE ||
                             (method == SP_METHOD_FLATTENING && sparse_ok && ctx->gram_mode == 0 && !ctx->opt.force_big);
    if (want_sparse) {
        sp_plan* plan = nullptr;
        SP_CHECK(cached_sparse_plan(ctx, al->n_taxa, split_taxa, split_a, n_splits, &plan));
        return enqueue_sparse_plan(ctx, &al, 1, plan, (double*)scores_dev, (int*)status_dev, true);
    }
    // Other methods and routes: the synchronous entry point's route with device outputs and WITHOUT its result step - no
    // status fetch, no host synchronisation behind the kernels: a flagged split (status bit 0 / 1: the dense route's 4-wide
    // block found no certificate) stays flagged for the caller's sp_finish_flagged, as on the sparse route.  (Until round 4
    // this fetched the status words to look for flagged splits: one stream synchronisation per call, i.e. the steps of a
    // pipeline over the dense route could not overlap.)
    struct AsyncScope {
        sp_ctx* c;
        explicit AsyncScope(sp_ctx* c_) : c(c_) { c->async_results = true; }
        ~AsyncScope() { c->async_results = false; }
    } scope(ctx);
    int rc = sp_score_splits(al, split_taxa, split_a, n_splits, method, nullptr, scores_dev, nullptr);
    if (rc != SP_OK && rc != SP_ENOCONV) return rc;
    SP_HIP(hipMemcpyAsync(status_dev, ctx->status.p, (size_t)n_splits * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return SP_OK;
    });
}

// Several alignments (same taxa, same split list) in ONE device pass: n_al * n_splits items through the sparse route's
// chain.  scores_dev / status_dev hold n_al * n_splits entries, alignment-major.
extern "C" int sp_score_splits_multi_async(sp_alignment* const* als, int n_al, const int32_t* split_taxa,
                                           const int32_t* split_a, int64_t n_splits, void* scores_dev,
                                           void* status_dev) {
    return sp_guard("sp_score_splits_multi_async", [&]() -> int {
    SP_REQUIRE(als && n_al >= 1 && split_taxa && split_a && scores_dev && status_dev, SP_EINVAL, "NULL argument");
    sp_alignment* al0 = als[0];
    SP_REQUIRE(al0, SP_EINVAL, "NULL alignment");
    sp_ctx* ctx = al0->ctx;
    SP_HIP(hipSetDevice(ctx->device));
    if (n_splits == 0) return SP_OK;
    sp_plan* plan = nullptr;
    SP_CHECK(cached_sparse_plan(ctx, al0->n_taxa, split_taxa, split_a, n_splits, &plan));
    return enqueue_sparse_plan(ctx, als, n_al, plan, (double*)scores_dev, (int*)status_dev, true);
    });
}

// phylogenetics.py:364-373 for a dense row-major matrix on the host (any non-negative matrix; cells equal to 0 are skipped).
extern "C" int sp_divergence_matrix_f64(sp_ctx* ctx, const double* m, int64_t rows, int64_t cols, int64_t ld,
                                        double* out) {
    return sp_guard("sp_divergence_matrix_f64", [&]() -> int {
    SP_REQUIRE(ctx && m && out, SP_EINVAL, "NULL argument");
    SP_REQUIRE(rows >= 1 && cols >= 1 && ld >= cols, SP_EINVAL, "bad matrix shape %lld x %lld (ld %lld)",
               (long long)rows, (long long)cols, (long long)ld);
    SP_HIP(hipSetDevice(ctx->device));
    if (ctx->cache) ctx->cache->valid = false;
    SP_CHECK(ctx->mats.ensure((size_t)rows * cols * 8));
    SP_CHECK(ctx->misc.ensure((size_t)(2 * rows + cols + 1) * 8));
    SP_HIP(hipMemcpy2DAsync(ctx->mats.p, (size_t)cols * 8, m, (size_t)ld * 8, (size_t)cols * 8, (size_t)rows,
                            hipMemcpyHostToDevice, ctx->stream));
    double* scratch = ctx->misc.as<double>();
    double* res = scratch + 2 * rows + cols;
    SP_CHECK(launch_divergence_matrix(ctx, ctx->mats.as<double>(), rows, cols, scratch, res));
    SP_HIP(hipMemcpyAsync(out, res, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(hipStreamSynchronize(ctx->stream));
    return SP_OK;
    });
}
