// Internal declarations shared by the .hip translation units of libsplitp_hip.so.
// (The public C ABI is include/splitp_hip.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/splitp_hip.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// The wave's index in its workgroup as a value the compiler knows to be wave-uniform (threadIdx.x >> 6 is "divergent" to its
// analysis): loops dealt out by wave, their bounds and the addresses built from them then live in scalar registers and
// branch on the scalar unit instead of taking a vector register and an exec-mask sequence each.
__device__ __forceinline__ int sp_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// a / b and sqrt(x) to the last bit or two from the hardware seeds (v_rcp_f64 / v_rsq_f64, good to 2^-24) and Newton steps:
// 8 and 7 instructions.  The IEEE forms the compiler emits for `/` and sqrt() are ~30 and ~25 (v_div_scale / v_div_fmas /
// v_div_fixup around the same seed) - and in a 1024-thread workgroup every instruction of block-wide code costs 16 cycles,
// in a section one wave runs alone while 15 wait at the barrier ~8.  For stop rules, conditioning estimates and start
// scalings; b and x must be normal, finite and non-zero (callers guard with a select: the discarded NaN is harmless).
__device__ __forceinline__ double sp_fdiv(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    y = fma(fma(-b, y, 1.0), y, y);
    y = fma(fma(-b, y, 1.0), y, y);
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}
__device__ __forceinline__ double sp_fsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    const double s = x * r;
    return fma(0.5 * r, fma(-s, s, x), s);
}

void sp_set_error(const char* fmt, ...);

#define SP_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            sp_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return SP_EHIP;                                                                       \
        }                                                                                         \
    } while (0)

#define SP_CHECK(call)          \
    do {                        \
        int r__ = (call);       \
        if (r__ != SP_OK) return r__; \
    } while (0)

#define SP_REQUIRE(cond, code, ...) \
    do {                            \
        if (!(cond)) {              \
            sp_set_error(__VA_ARGS__); \
            return (code);          \
        }                           \
    } while (0)

// No C++ exception crosses the C ABI: every extern "C" body runs inside sp_guard.
#include <exception>
#include <new>
template <typename F>
static inline int sp_guard(const char* fn, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        sp_set_error("%s: out of host memory", fn);
        return SP_ENOMEM;
    } catch (const std::exception& e) {
        sp_set_error("%s: C++ exception: %s", fn, e.what());
        return SP_EHIP;
    } catch (...) {
        sp_set_error("%s: unknown C++ exception", fn);
        return SP_EHIP;
    }
}

// Grow-only device buffer.  Re-allocation synchronises the device (rare: sizes settle after
// the first call of a given shape; 288 GB of HBM means we never need to be frugal).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// Every counts buffer of an alignment (sp_alignment::counts, ::spk_counts) is allocated with this many bytes behind its last
// entry: the sparse kernels fetch a lane's counts as whole 16-byte loads starting anywhere up to the table's end
// (sparse.hip: spk_prefetch_counts reads up to SPK_MAXQ * 4 = 80 bytes past it and ignores them) - a contract of the
// allocation sites, not an accident of DevBuf's growth slack.
#define SP_COUNTS_PAD 128

struct PhaseTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[SP_N_PHASES];
    std::vector<hipEvent_t> pool;
    double ms[SP_N_PHASES] = {0};
    int64_t launches[SP_N_PHASES] = {0};
};

struct PlanCache;
// One alignment as the batched kernels see it (sparse route scores several alignments in one launch).
#define SPK_NTOP 16
// split-independent facts about a pattern table, computed once per alignment on the device (k_sparse_meta)
struct SpkMeta {
    unsigned long long trace;   // sum of count^2 (exact)
    u32 top[SPK_NTOP];          // indices of the SPK_NTOP largest counts, descending (ties: lowest index first)
    u32 ntop;                   // min(SPK_NTOP, D)
    u32 pad;                    // rows of the table before large counts were split into pieces (== D when none was)
};
struct AlDesc {
    const u32* keys32;          // pattern keys narrowed to 32 bits (n_taxa <= 16)
    const u32* counts;
    const SpkMeta* meta;
    int64_t D;
};

// Test / tuning switches of a context.  Defaults come from the environment ONCE, at sp_ctx_create (SPLITP_FORCE_BIG,
// SPLITP_BIG_BY_KEYS, SPLITP_SUBSCORE_JACOBI, SPLITP_DIVERGENCE_GLOBAL, SPLITP_HIST_SORT, SPLITP_DEBUG_LDS_CAP); later
// changes go through sp_ctx_set_option.  They select between kernels that must agree, never a CPU path.
struct CtxOptions {
    int force_big = 0;          // every flattening score on the big-table form
    int big_by_keys = 0;        // ... with its sort-based compaction
    int subscore_jacobi = 0;    // Jacobi kernel for the batched subflattening score
    int moments_valu = 0;       // the round-1 moment kernels (vector units, int64 partial sums) instead of the matrix-core form
    int subscore_pair = 1;      // two splits a wave (subflat_pair.hip) where the batch's classes are known; 0 = one split a wave
    int subscore_waves = 0;     // waves per workgroup of the fast subflattening score kernel (0 = the shape that fills the CU)
    int divergence_global = 0;  // global-memory form of the mutual-information score
    int hist_sort = -1;         // -1 auto, 0 direct bins, 1 sort + run-length encode
    int wide_cap = 0;           // half-product cap of the wide fallback block (0 = built-in 600)
    int gram_tile64 = 0;        // int8 Gram of the dense route on the 64 x 64-tile kernel instead of 128 x 128
    int eigen_one_stream = 0;   // dense route's eigen phase on the context's stream only (no side stream for the short sides)
    long long lds_cap = 0;      // pretend the LDS is this small (plain LDS form of the sparse kernel)
    int direct_finish = 1;      // flagged splits (status bit 0 / 1) end in the direct solver (finish.hip); 0: they stay flagged (SP_ENOCONV)
    int direct_max_rows = 0;    // largest smaller side (compact rows) the direct solver takes (0 = 16384)
    int direct_all = 0;         // dense route / generic matrices: the direct solver instead of the block iteration (test switch)
    int sort_three_launch = 0;  // radix sorts (n > 12 histogram, big-table form) in round 3's three-launches-a-pass form instead of one-sweep
    int sort_digit_bits = 0;    // one-sweep sorts: 9 = 9-bit digits (test switch; measured slower per pass than the pass they save), else 8
    int eigen_block16 = 0;      // dense route / generic matrices: rounds 1 - 3's 16-wide block pipeline (eigen.hip) instead of the
                                // certified 4-wide kernel (eig4.hip) - kept as a cross-check
};

// the size classes of one batch (at most 16); the kernel reads a copy in device memory.  Positions are indices into `order` (or, without one, into
// the split list itself).
struct PairClasses {
    int nclass;
    int rows[16];            // 3 * (smaller side) + 1
    long long start[16];     // first position of the class
    long long count[16];     // its splits
    long long poff[17];      // pairs before the class: a class of c splits makes (c + 1) / 2 pairs
};

struct sp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool timing = false;
    int gram_mode = 0;  // 0 = auto (int8 limbs when the alignment is exact and counts < 128^3), 1 = always fp64 MFMA
    PhaseTimer timer;
    // workspace pools (see DESIGN.md "HBM layout")
    DevBuf splits;     // SplitDev[n_splits]
    DevBuf splits_launch;   // ... and in launch order (device-planned all-splits call)
    DevBuf bitmaps;    // presence bitmaps + rank prefixes
    DevBuf coords;     // compact (row, col) of every (split, pattern)
    DevBuf dims;       // int2 (R, C) per split
    DevBuf mats;       // compact count / weight matrices
    DevBuf grams;      // Gram matrices (fp64)
    DevBuf eigws;      // eigen workspace
    DevBuf scores;     // double per split
    DevBuf status;     // int per split
    DevBuf misc;       // API scratch
    DevBuf misc2;
    DevBuf gram_items; // GramItem[]: Gram tiles, then the row-block items
    DevBuf aldescs;    // AlDesc[] of the current multi-alignment call (this context's stream only)
    DevBuf chain;      // work-queue head of the sparse route's device-side hand-back chain (k_sparse_slow)
    CtxOptions opt;
    DevBuf slabs;      // per-workgroup global-memory slabs of the sparse kernel's HBM form (grow-only)
    DevBuf big[24];    // work buffers of the big-table form (grow-only: a multi-GB hipMalloc / hipFree per call costs more than the kernels)
    std::vector<AlDesc> aldescs_host;
    PlanCache* cache = nullptr;
    int n_cu = 256;
    // dense route, eigen phase: the splits whose smaller side is short (<= EIG_SMALL_ROWS rows: a third of the LDS, three
    // workgroups per CU) run as a pipeline of their own on this internal stream, next to the long sides on `stream`
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // direct-bins histogram (hist.hip), tables of up to 12 taxa: the 4^n-bin array is kept between calls and handed back
    // ALL ZERO by the pass that reads it (no allocation, no memset per alignment); hist_clean = that invariant holds
    DevBuf hist_bins, hist_blk, hist_off;
    bool hist_clean = false;
    DevBuf hist_work[6];   // sort-based histogram: sorted keys (2), runs, run lengths, result words, sort work (grow-only)
    // all-splits enumeration (subflat.hip: enumerate_all_splits) of the last (n, trivial, size, shard): kept between calls
    DevBuf enum_buf;
    long long enum_key[6] = {0, 0, 0, 0, 0, 0};
    bool enum_valid = false;
    // set by sp_score_splits_async around its call of the synchronous entry: results stay on the device, no status fetch,
    // flagged splits are left to the caller's sp_finish_flagged (api.hip)
    bool async_results = false;
    // size classes of the last paired subflattening batch (subflat_pair.hip), host copy and the device copy the kernel reads
    PairClasses pair_host = {};
    DevBuf pair_dev;
    bool pair_valid = false;
};

struct sp_alignment {
    sp_ctx* ctx = nullptr;
    int n_taxa = 0;
    int64_t D = 0;
    int64_t N = 0;
    bool exact = false;
    DevBuf keys;     // u64[D]
    DevBuf weights;  // double[D]
    DevBuf counts;   // u32[D]   (exact only)
    DevBuf keys32;   // u32[D]   sparse route: keys narrowed to 32 bits (n_taxa <= 16), filled on first use
    DevBuf spk_meta; // SpkMeta  sparse route: trace + largest counts, filled on first use
    DevBuf aldesc;   // AlDesc   sparse route: this table as the kernels see it (device copy, written once)
    bool spk_ready = false;   // keys32 / spk_meta / aldesc are complete AND visible to every stream (prepared synchronously)
    // counts >= 2^16 do not fit the 16-bit count field of the sparse kernel's list entries: such a pattern is entered as
    // several table rows with the same key whose counts add up (every product is linear in the entries); spk_D = number
    // of rows of that expanded table (-1: not computed yet), spk_keys / spk_counts hold it when it differs from the table
    int64_t spk_D = -1;
    DevBuf spk_keys;    // u64[spk_D]
    DevBuf spk_counts;  // u32[spk_D], every value < 65536
    double sumsq_w = 0;  // sum of weights^2 (host-computed, informational)
    uint32_t max_count = 0;  // largest count (exact only): decides the number of 7-bit limbs of the int8 Gram
    // cached signed second-moment matrix (subflattening path)
    bool moments_ready = false;
    DevBuf moments;  // int64 or double [(3n+1)^2]
    // state of the last sp_flatten_reduced_prepare
    int64_t red_R = 0, red_C = 0;
    bool red_ready = false;
};

// One candidate split as the kernels see it.
struct SplitDev {
    int32_t nr, nc;      // taxa on the row side / column side
    int8_t taxa[32];     // row-side taxa (most significant first), then column-side taxa
    int64_t bm_off;      // u64-word offset of the row bitmap in the bitmap pool; col bitmap follows
    int32_t rw, cw;      // words in the row / col bitmaps
    int64_t pfx_off;     // u32 offset of the rank-prefix arrays (row prefixes, then col prefixes)
    int64_t mat_off;     // element offset of this split's compact matrix in the matrix pool
    int32_t pitch;       // matrix row pitch in elements
    int32_t rcap;        // allocated rows (multiple of 64)
    int64_t g_off;       // element offset of the Gram matrix in the gram pool
    int32_t g_pitch;     // Gram pitch (= rcap)
    int32_t cls;         // launch-ordered copies (sp_plan::launch_dev): the split's index in the list
    int64_t ev_off;      // element offset of the 16 x rcap iteration blocks (V^T, Y) in the eigen pools
};

struct GramItem {
    int32_t sid;
    int16_t ti, tj;
};

// Host-side launch plan for one batch of splits (cached in the context: scoring the same split
// list again - the benchmark loop, an erickson round replayed - skips planning and the uploads).
struct Plan {
    std::vector<SplitDev> splits;
    std::vector<GramItem> gram_items;   // upper-triangle 64 x 64 tiles, heaviest splits first
    std::vector<GramItem> row_items;    // 64-row blocks (ti = block index) for the G V product
    std::vector<GramItem> gram_items_big;   // upper-triangle 128 x 128 tiles (int8 Gram), longest K first
    std::vector<int> order;             // split ids, heaviest first (block -> split map of the per-split kernels)
    // eigen phase classes: the first n_order_a entries of `order` / n_row_a entries of `row_items` belong to the splits
    // with more than EIG_SMALL_ROWS allocated rows, the rest to the short sides (each part XCD-interleaved by itself)
    size_t n_order_a = 0, n_row_a = 0;
    size_t bm_words = 0, pf_words = 0, mat_elems = 0, g_elems = 0, ev_elems = 0;
};

struct PlanCache {
    sp_plan* sparse = nullptr;   // retained plan of the last sparse-route split list (asynchronous entry points)
    bool valid = false;
    int nl = 0;  // limb count the plan was laid out for (0 = element-typed u32 / f64 matrices, -1 = sparse route)
    int n = 0;
    int64_t D = 0;
    std::vector<int32_t> taxa, a;
    Plan plan;
};

// What the host knows about whether any (alignment, split) item can run in the sparse kernel's in-LDS form: the smallest
// table of the call and, over the split list, the smallest bitmap footprint (12 bytes per word in use) of a Gram-path
// split (row side <= 3 taxa) and of a general-path split; -1 = no such split, d_min <= 0 = unknown (never skip).
struct SparseFitHint {
    int64_t d_min = 0;
    int64_t w12_gpath = -1, w12_general = -1;
    void add_split(int nr, int nc, int64_t rw, int64_t cw) {   // SPK_RAW_MAX = 5: shorter sides keep their raw ids, no bitmap
        const int64_t w12 = 12 * ((nr > 5 ? rw : 0) + (nc > 5 ? cw : 0));
        int64_t& slot = nr <= 3 ? w12_gpath : w12_general;
        slot = slot < 0 ? w12 : (w12 < slot ? w12 : slot);
    }
};

// An immutable, reference-counted candidate-split list on the device (sp_plan_create): split descriptors + the
// heaviest-first launch order.  Uploaded synchronously at creation and never written again, so any number of contexts
// (lanes) of the same device may score with it concurrently.
struct sp_plan {
    std::atomic<int> refs{1};   // (retain / release may come from several host threads, one per lane)
    int device = 0;
    int n = 0;
    int64_t S = 0;
    std::vector<SplitDev> splits;   // host copy
    std::vector<int32_t> taxa, a;   // the list as given (content key of the internal plan cache)
    int64_t bm_words_max = 0;       // largest rw + cw among the splits (slab sizing)
    SparseFitHint fit;              // cheapest split of the list for the in-LDS form (d_min is filled in per call)
    DevBuf splits_dev;   // SplitDev[S], list order (the slow kernel and the dense route index it by split)
    DevBuf launch_dev;   // SplitDev[S], launch order (heaviest first), each with its split index in `cls`
};

// hipFuncSetAttribute (dynamic LDS beyond 64 KB) is a per-DEVICE setting of a function: a process that drives several GPUs
// (node.hip) must set it once on each - a process-wide `static bool` served the first device only.  Thread-safe.
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool need(int device) const { return !((mask.load(std::memory_order_acquire) >> (device & 63)) & 1ull); }
    void done(int device) { mask.fetch_or(1ull << (device & 63), std::memory_order_release); }
};

struct PhaseScope {
    sp_ctx* c;
    int phase;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    PhaseScope(sp_ctx* ctx, int ph);
    ~PhaseScope();
};

#define EIG_SMALL_ROWS 256   // eigen phase: sides up to this many rows form the class that runs three workgroups per CU
#define EIG_MAXR 1024   // rows of the smaller side the dense route's eigen kernels hold in LDS (eig_small.h: 8 waves x 8 tiles x 16)
static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t pow4(int k) { return (int64_t)1 << (2 * k); }

// ---- launchers implemented in the kernel translation units -------------------------------
int launch_reindex(sp_ctx* ctx, const u64* keys, int64_t D, int n_taxa, const SplitDev* splits_dev,
                   const std::vector<SplitDev>& splits, u64* bitmaps, u32* prefixes, int2* dims, u32* rr, u32* cc);
int launch_reindex_coo(sp_ctx* ctx, const int64_t* rows_in, const int64_t* cols_in, int64_t nnz,
                       const SplitDev* split_dev, u64* bitmaps, u32* prefixes, int2* dims, u32* rr, u32* cc);
int launch_bit_indices(sp_ctx* ctx, const u64* keys, int64_t D, int n_taxa, const SplitDev* split_dev,
                       int64_t* rows, int64_t* cols);
template <typename T>
int launch_zero_scatter(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, int64_t D,
                        const int2* dims, const u32* rr, const u32* cc, const T* vals, T* mats);
int launch_used_keys(sp_ctx* ctx, const u64* keys, int64_t D, int n_taxa, const SplitDev* split_dev, const u32* rr,
                     const u32* cc, int64_t* row_keys, int64_t* col_keys);
int launch_dense_scatter(sp_ctx* ctx, const u64* keys, const u32* counts, int64_t D, int n_taxa,
                         const SplitDev* split_dev, const SplitDev& split, u32* out);
int launch_zero_scatter_i8(sp_ctx* ctx, int nl, const SplitDev* splits_dev, const std::vector<SplitDev>& splits,
                           int64_t D, const int2* dims, const u32* rr, const u32* cc, const u32* vals, uint8_t* mats);
int launch_gram_i8(sp_ctx* ctx, int nl, bool g_i32, const SplitDev* splits_dev, const GramItem* items_dev,
                   int64_t n_items, const int2* dims, const uint8_t* mats, void* grams);
int launch_gram_i8_big(sp_ctx* ctx, int nl, bool g_i32, const SplitDev* splits_dev, const GramItem* items_dev,
                       int64_t n_items, const int2* dims, const uint8_t* mats, void* grams);
int launch_divergence(sp_ctx* ctx, bool exact, int64_t D, int64_t S, const u32* rr, const u32* cc, const u32* counts,
                      const double* weights, double n_total, unsigned long long* marg, double* out);
int launch_divergence_matrix(sp_ctx* ctx, const double* m_dev, int64_t rows, int64_t cols, double* scratch, double* out);
size_t sparse_slab_bytes(int64_t D, int64_t bm_words, bool wide = false);
size_t sparse_list_slab_bytes(int64_t D);
int launch_sparse_big_keys(sp_ctx* ctx, const u64* keys, int64_t D, int n, const int32_t* split_taxa, const int32_t* split_a,
                           int64_t S, const u32* counts, const double* weights, int dev_cus, double* scores, int* status);
int launch_sparse_big(sp_ctx* ctx, int64_t D, int64_t S, const u32* rr, const u32* cc, const u32* counts,
                      const double* weights, const int2* dims, int dev_cus, double* scores, int* status);
int launch_sparse_chain(sp_ctx* ctx, const AlDesc* als_dev, const AlDesc& al0, int n_al, int n_taxa,
                        const SplitDev* splits_dev, const SplitDev* launch_dev, int64_t S, double* scores, int* status,
                        int64_t d_max, int64_t bm_words_max, bool wide_all, const SparseFitHint& hint = SparseFitHint());
int launch_sparse_meta(sp_ctx* ctx, const u64* keys, const u32* counts, int64_t D, u32* keys32, SpkMeta* meta,
                       unsigned long long trace_override, int64_t orig_rows);
template <typename T>
int launch_gram(sp_ctx* ctx, const SplitDev* splits_dev, const GramItem* items_dev, int64_t n_items, const int2* dims,
                const T* mats, double* grams);
void build_gram_items(Plan& plan);
int launch_eigen4(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, const int2* dims,
                  const void* grams, bool g_i32, const int* order_dev, double* scores, int* status);   // eig4.hip
size_t direct_ws_doubles(int64_t n_mats, int64_t max_rows);   // finish.hip
int launch_direct_top4(sp_ctx* ctx, const SplitDev* splits_dev, const int2* dims_dev, int64_t n_mats, int max_rows,
                       double* grams, double* ws_dev, const int* out_idx_dev, double* scores, int* status);
int launch_eigen(sp_ctx* ctx, const SplitDev* splits_dev, const std::vector<SplitDev>& splits, const int2* dims,
                 const void* grams, bool g_i32, const GramItem* rowblocks_dev, int64_t n_rowblocks,
                 const int* order_dev, double* scores, int* status, int64_t n_rowblocks_a = -1, int64_t n_splits_a = -1);
