/*
 * splitp_hip.h - C ABI of the MI355X (gfx950) implementation of SplitP's
 * flattening / subflattening / split_score hot path.
 *
 * The reference (js51/SplitP v0.3.2) is pure Python and has NO FFI or plugin interface
 * (SURVEY.md section 8b): the drop-in boundary is its Python call surface
 *     splitp.flattening     (splitp/constructions.py:7)
 *     splitp.subflattening  (splitp/constructions.py:108)
 *     splitp.split_score    (splitp/phylogenetics.py:315)
 * which splitp_amd/ mirrors.  This header is the C ABI underneath that surface: what a
 * maintainer of the reference would bind with ctypes to replace the bodies of those three
 * functions (INTEGRATION.md shows the stub).  Every entry point names the reference code
 * it replaces.
 *
 * Conventions
 *   - every function returns an int status: 0 = SP_OK, otherwise an SP_E* code; the message
 *     is available from sp_last_error() (thread-local).
 *   - plain pointers and sizes only; no C++ exceptions cross the ABI.
 *   - "host" pointers are ordinary process memory (NumPy arrays); "dev" pointers are HIP
 *     device pointers (e.g. torch tensor data_ptr()).  The library owns all other device
 *     memory behind the opaque handles.
 *   - a context is bound to one device and one HIP stream; it is not thread-safe.  All
 *     kernels of a call are enqueued on the context's stream; calls that return host data
 *     synchronise that stream before returning, calls that write only device memory do not.
 *   - taxa are numbered 0..n-1 in the order of the pattern string (taxon 0 = first
 *     character).  A pattern key is the base-4 value of the pattern string with
 *     A,C,G,T = 0,1,2,3 (splitp/constants.py:7-8) and the FIRST character most significant
 *     (splitp/constructions.py:166-171).
 *   - a split is given as two ordered lists of taxon indices (the order in which the
 *     reference iterates split[0] / split[1]: first listed = most significant base-4 digit of
 *     the row / column index, constructions.py:39-40).  The two lists must be disjoint and
 *     together cover all n taxa (the reference silently overwrites colliding cells otherwise,
 *     constructions.py:43,:101 - rejected here with SP_EINVAL).
 */
#ifndef SPLITP_HIP_H
#define SPLITP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_EINVAL 1   /* bad argument */
#define SP_EHIP 2     /* a HIP runtime call failed */
#define SP_ENOMEM 3   /* device or host allocation failed */
#define SP_ELIMIT 4   /* size outside what this build supports (see message) */
#define SP_ENOCONV 5  /* eigen iteration did not converge (score still written, flagged) */

#define SP_ABI_VERSION 4   /* 3: sp_score_plan_steps; 4: sp_finish_flagged, status bit 2 (direct solver) */

typedef struct sp_ctx sp_ctx;             /* device + stream + workspace arena */
typedef struct sp_alignment sp_alignment; /* device-resident pattern table */
typedef struct sp_plan sp_plan;           /* immutable, reference-counted candidate-split list on the device */

/* ---------------------------------------------------------------- context ---------- */
int sp_abi_version(void);
const char* sp_last_error(void);
/* number of visible HIP devices (0 when there is no GPU; never fails) */
int sp_device_count(void);

/* device: HIP ordinal.  stream: a hipStream_t to adopt (e.g. torch's current stream), or
 * NULL to create a private non-blocking stream owned by the context. */
int sp_ctx_create(int device, void* stream, sp_ctx** out);
int sp_ctx_destroy(sp_ctx* ctx);
int sp_ctx_set_stream(sp_ctx* ctx, void* stream);
int sp_ctx_synchronize(sp_ctx* ctx);
/* Test / tuning switches (they select between kernels that must agree - never a CPU path).  Defaults are read from the
 * environment once, at sp_ctx_create (SPLITP_<NAME IN CAPITALS>, SPLITP_DEBUG_LDS_CAP for "lds_cap").  Names:
 *   "force_big" 0/1, "big_by_keys" 0/1, "subscore_jacobi" 0/1, "divergence_global" 0/1, "hist_sort" -1 auto / 0 / 1,
 *   "lds_cap" bytes (0 = off), "wide_cap" half products of the sparse route's last resort (0 = built-in 600),
 *   "gram_tile64" 0/1 (dense route's int8 Gram on 64 x 64 tiles instead of 128 x 128), "eigen_one_stream" 0/1 (dense
 *   route's eigen phase without the internal side stream for the short sides), "subscore_waves" 0 auto / 1..16 (waves per
 *   workgroup of the batched subflattening score: by default the shape that puts the most waves on a CU). */
int sp_ctx_set_option(sp_ctx* ctx, const char* name, int64_t value);
int sp_ctx_get_option(sp_ctx* ctx, const char* name, int64_t* value);
/* Gram-kernel selection of the dense flattening route: 0 = auto (exact integer Gram on the int8 matrix
 * cores, count limbs of 7 bits, when the alignment holds counts < 128^3; fp64 MFMA otherwise),
 * 1 = always the fp64 MFMA kernel.  Both give bit-identical Gram matrices for integer counts. */
int sp_ctx_set_gram_mode(sp_ctx* ctx, int mode);

/* Per-phase device timing.  When enabled, every batched call brackets each phase with
 * hipEvents on the context's stream; sp_ctx_phase_times() then synchronises and returns,
 * for each phase, the accumulated milliseconds and number of launches since the last
 * reset.  Phase ids: SP_PHASE_*. */
#define SP_PHASE_REINDEX 0  /* per-split bit-gather + presence bitmaps + ranks            */
#define SP_PHASE_SCATTER 1  /* zero-fill + scatter into the (compact) count matrix        */
#define SP_PHASE_GRAM 2     /* fp64 MFMA Gram over the smaller side                       */
#define SP_PHASE_EIGEN 3    /* top-4 eigenvalues + score                                  */
#define SP_PHASE_MOMENT 4   /* signed second-moment matrix (subflattening path)           */
#define SP_PHASE_SUBSCORE 5 /* Gram + top-4 eigenvalues of the (3a+1)x(3b+1) blocks        */
#define SP_PHASE_HIST 6     /* site-pattern histogram (alignment columns -> pattern table) */
#define SP_PHASE_DENSE 7    /* full 4^a x 4^b dense scatter (sp_flatten_dense)             */
#define SP_PHASE_SPARSE 8   /* sparse route: one workgroup per split, everything in LDS     */
#define SP_PHASE_DIVERGENCE 9 /* mutual-information score (marginals + sum over the patterns)  */
#define SP_PHASE_CHAIN 10   /* sparse route: the hand-back kernel queued behind the in-LDS kernel (k_sparse_slow) */
#define SP_N_PHASES 11
int sp_ctx_enable_timing(sp_ctx* ctx, int on);
int sp_ctx_reset_timing(sp_ctx* ctx);
int sp_ctx_phase_times(sp_ctx* ctx, double* ms /*[SP_N_PHASES]*/, int64_t* launches /*[SP_N_PHASES]*/);

/* ---------------------------------------------------------------- alignment -------- */
/* Upload a pattern table (the dict the reference passes as `pattern_probabilities`,
 * e.g. from splitp/simulation.py:42-56 or splitp/parsers/fasta.py:66-80).
 *   keys[D]     pattern keys (see above), all distinct
 *   weights[D]  the table's float values (relative frequencies or exact probabilities)
 *   counts[D]   optional (may be NULL): integer site counts with weights[i] == counts[i]/N
 *               exactly; enables the exact-integer Gram / moment path
 *   N           number of sites (sum of counts), ignored when counts == NULL */
int sp_alignment_create(sp_ctx* ctx, const uint64_t* keys, const double* weights, const int64_t* counts,
                        int64_t D, int n_taxa, int64_t N, sp_alignment** out);

/* Build the pattern table on the device from alignment columns: the site-pattern histogram
 * that precedes the path (splitp/parsers/fasta.py:48-63 get_pattern_counts; sites containing a
 * character outside ACGT, either case, are skipped, :54-57).  seqs is n_taxa rows of L ASCII
 * characters (row stride `stride` bytes, host memory).  Up to 12 taxa or so the sites are counted into a direct 4^n bin
 * array; beyond that (4^n > 32 L, or more than 16 taxa) they are radix-sorted and run-length encoded; the table is the
 * same either way (keys ascending).  Up to 31 taxa (at 32 the invalid-site marker would coincide with a pattern). */
int sp_alignment_from_sequences(sp_ctx* ctx, const uint8_t* seqs, int n_taxa, int64_t L, int64_t stride,
                                sp_alignment** out);
/* Same, from already packed site keys resident on the host (one key per site). */
int sp_alignment_from_site_keys(sp_ctx* ctx, const uint64_t* site_keys, int64_t L, int n_taxa, sp_alignment** out);

/* Device simulator + histogram: replaces splitp/simulation.py:9-56 generate_alignment(tree, model, L).
 * The tree is given parents-first: node 0 is the root (parent -1); parent[i] < i; leaf_taxon[i] = taxon index of a leaf,
 * -1 for an internal node; transition[i] = row-major 4 x 4 matrix M of node i's branch with M[new][old] (a state is
 * drawn from column `old`, simulation.py:17-18; ignored for the root, whose state is uniform, :28).  Sites are
 * independent; the same seed gives the same table.  The result is the pattern table of the L simulated sites
 * (counts / L), resident on the device like every other alignment.  2..31 taxa, up to 64 tree nodes. */
int sp_simulate_alignment(sp_ctx* ctx, int n_nodes, const int32_t* parent, const int32_t* leaf_taxon,
                          const double* transition, int n_taxa, int64_t L, uint64_t seed, sp_alignment** out);

/* Frees the table.  An alignment must outlive all work in flight that reads it - also the work other contexts (lanes)
 * enqueued with sp_score_plan_async / sp_score_plan_steps: the call waits for the whole device before it frees. */
int sp_alignment_destroy(sp_alignment* al);
/* D distinct patterns, n taxa, N sites (0 if unknown), exact = 1 when integer counts are held */
int sp_alignment_info(const sp_alignment* al, int64_t* D, int* n_taxa, int64_t* N, int* exact);
/* copy the table back (sorted by key when it was built on the device); any pointer may be NULL */
int sp_alignment_fetch(sp_alignment* al, uint64_t* keys, double* weights, int64_t* counts);

/* ---------------------------------------------------------------- flattening ------- */
/* Row/column index of every pattern for one split - the body of the per-pattern loop of
 * constructions.py:88-93 (sparse/dok format).  rows[D], cols[D] host, table order. */
int sp_flatten_indices(sp_alignment* al, const int32_t* order_a, int a, const int32_t* order_b, int b,
                       int64_t* rows, int64_t* cols);

/* Reduced flattening (constructions.py:31-55): two calls.  _prepare computes the sorted sets
 * of used row / column keys on the device and returns their sizes; _fetch writes the R x C
 * float64 matrix (row-major, zeros elsewhere) and, if non-NULL, the used row/col keys. */
int sp_flatten_reduced_prepare(sp_alignment* al, const int32_t* order_a, int a, const int32_t* order_b, int b,
                               int64_t* R, int64_t* C);
int sp_flatten_reduced_fetch(sp_alignment* al, double* matrix, int64_t* row_keys, int64_t* col_keys);

/* Full 4^a x 4^b count matrix (uint32, row-major) of one split; requires integer counts and
 * 4^(a+b) <= 2^32 cells.  The .todense() of constructions.py:86-102 times N. */
int sp_flatten_dense_counts(sp_alignment* al, const int32_t* order_a, int a, const int32_t* order_b, int b,
                            uint32_t* out_host);

/* ---------------------------------------------------------------- subflattening ---- */
/* (3a+1) x (3b+1) signed-sum matrix of constructions.py:108-163, row-major float64. */
int sp_subflatten(sp_alignment* al, const int32_t* order_a, int a, const int32_t* order_b, int b, double* out_host);
/* the (3n+1)x(3n+1) signed second-moment matrix all subflattenings are sub-blocks of
 * (SURVEY.md appendix A.3); int64 when the alignment is exact (out_i64), else float64. */
int sp_moment_matrix(sp_alignment* al, int64_t* out_i64, double* out_f64);

/* ---------------------------------------------------------------- split score ------ */
/* phylogenetics.py:280-300 for a host matrix (row-major, leading dimension ld):
 * score = sqrt(max(0, 1 - (sum of the 4 largest sigma^2) / (sum of all sigma^2))).
 * min(rows, cols) <= 4 gives exactly 0 (the reference's 1 - x/x).  An all-zero matrix gives NaN
 * (the reference's 0/0). */
int sp_score_matrix_f64(sp_ctx* ctx, const double* m, int64_t rows, int64_t cols, int64_t ld, double* score);
/* phylogenetics.py:303-312 for a sparse matrix given as COO triplets (host). */
int sp_score_coo_f64(sp_ctx* ctx, const int64_t* ri, const int64_t* ci, const double* v, int64_t nnz,
                     int64_t rows, int64_t cols, double* score);

/* phylogenetics.py:364-373 flattening_rank_1_approximation_divergence(matrix): sum over the non-zero cells of
 * f * log(f / (column sum * row sum)), with the marginals of phylogenetics.py:332-341.  Dense row-major host matrix. */
int sp_divergence_matrix_f64(sp_ctx* ctx, const double* m, int64_t rows, int64_t cols, int64_t ld, double* out);

/* Batched: flattening + split_score for many splits of one alignment, everything on the
 * device (the README loop, README.md:36-41, as one call).
 *   split_taxa[n_splits * n]  for split s: order_a (a entries) then order_b (n - a entries)
 *   split_a[n_splits]         a of each split
 *   method                    SP_METHOD_FLATTENING (auto: the sparse in-LDS route when the table holds counts
 *                             < 65536, the dense MFMA route otherwise or for splits the sparse kernel hands
 *                             back), SP_METHOD_FLATTENING_DENSE, SP_METHOD_FLATTENING_SPARSE (error if a split
 *                             cannot be handled there), SP_METHOD_SUBFLATTENING or SP_METHOD_MUTUAL_INFORMATION
 *                             (flattening + flattening_rank_1_approximation_divergence instead of split_score)
 *   scores_host               may be NULL; if given, the stream is synchronised
 *   scores_dev                may be NULL; device buffer of n_splits doubles
 *   status_host               may be NULL; per-split flags (bit 0: no certificate - the score is then an upper
 *                             estimate; bit 2: scored by the direct solver; bits 8..: number of operator
 *                             applications, or the rows of the solved Gram matrix when bit 2 is set).
 * Hand-back chain of SP_METHOD_FLATTENING on a count table (every stage bit-reproducible): in-LDS kernel -> the same
 * kernel with its entry lists in global memory -> with all arrays in global memory (tables beyond ~9 k patterns) ->
 * for a split whose 4-wide block finds no certified spectral gap in 40 half products: the kernel's 8-wide block, which
 * certifies (gap behind the 8th value) or flags -> flagged splits: the direct solver (sp_finish_flagged, run here).
 * Tables with more than 65535 patterns and float-weight tables: the dense route up to 11 taxa, from 12 taxa on the
 * big-table form of the sparse route (segmented sorts + global-memory products, same block iteration and stop rule). */
#define SP_METHOD_FLATTENING 0
#define SP_METHOD_SUBFLATTENING 1
#define SP_METHOD_FLATTENING_DENSE 2
#define SP_METHOD_FLATTENING_SPARSE 3
#define SP_METHOD_MUTUAL_INFORMATION 4 /* KL divergence of the flattening from the product of its marginals
                                        * (phylogenetics.py:364-373, erickson_SVD's Method.mutual_information) */
int sp_score_splits(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t n_splits,
                    int method, double* scores_host, void* scores_dev, int32_t* status_host);

/* Scores of EVERY split of the table's taxa, in the order of the reference's all_splits generator (splits.py:39-59:
 * size classes ascending, itertools.combinations order inside a class, the side holding taxon 0 first; `trivial` adds
 * the 1 | n-1 class, `size` > 0 restricts to one class).  The splits are enumerated on the device (no split list crosses
 * the boundary: the 524 267 splits of 20 taxa take 1.2 s to encode on the host and 21 ms to score).
 * method: any SP_METHOD_*.  SP_METHOD_SUBFLATTENING scores straight from the enumeration; SP_METHOD_FLATTENING on a
 * table the sparse route takes is also PLANNED on the device (split descriptors + launch order built by a kernel from the
 * enumeration); the other routes fetch the enumerated list once and plan on the host.  n_splits receives the number of
 * splits; with all three output pointers NULL the call only counts.  Outputs and SP_ENOCONV as for sp_score_splits. */
int sp_score_all_splits(sp_alignment* al, int method, int trivial, int size, int64_t* n_splits, double* scores_host,
                        void* scores_dev, int32_t* status_host);
/* One shard of the same enumeration for multi-GPU runs ("index-mod-P within each class"): rank shard_rank of
 * shard_world scores the combinations shard_rank, shard_rank + shard_world, ... of EVERY size class - an equal share of
 * every cost class, nothing but two integers crosses the boundary.  Its n_splits results come class by class, in
 * enumeration order inside a class: the j-th result of class q is split (class start) + shard_rank + j * shard_world of
 * all_splits (splitp_amd/batch.py shard_layout un-permutes the all-gathered buffers).  status_dev: optional device
 * buffer for the status words (so that scores and status can be written straight into a collective's send buffer). */
int sp_score_all_splits_shard(sp_alignment* al, int method, int trivial, int size, int shard_rank, int shard_world,
                              int64_t* n_splits, double* scores_host, void* scores_dev, int32_t* status_host,
                              void* status_dev);

/* Asynchronous form for pipelines that keep everything on the device (benchmark loop, multi-GPU all-gather):
 * enqueues the scoring of the splits on the context's stream and returns without any host synchronisation.
 *   scores_dev[n_splits] (double) and status_dev[n_splits] (int32) are device buffers written by the kernels.
 * With SP_METHOD_FLATTENING / _SPARSE on a table the sparse route takes, this is sp_score_plan_async on an internally
 * cached plan: the hand-back chain runs on the device and every score is final (status as described there).  Other
 * methods and tables run the route of sp_score_splits with device outputs and without its result step: nothing is
 * fetched, and a split the dense route's eigen kernel flags (status bit 0 / 1) is left to sp_finish_flagged. */
int sp_score_splits_async(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t n_splits,
                          int method, void* scores_dev, void* status_dev);

/* Several alignments (same taxa, same candidate-split list - e.g. the replicates of a simulation study, BASELINE
 * config 5) in ONE device pass; asynchronous like sp_score_splits_async.  scores_dev / status_dev hold
 * n_al * n_splits entries, alignment-major. */
int sp_score_splits_multi_async(sp_alignment* const* als, int n_al, const int32_t* split_taxa, const int32_t* split_a,
                                int64_t n_splits, void* scores_dev, void* status_dev);

/* ---------------------------------------------------------------- plans and lanes --- */
/* A candidate-split list as a device object: validated, laid out and uploaded ONCE (synchronously), immutable
 * afterwards and reference-counted - the loop of README.md:36-41 over many alignments, or the same alignment scored
 * from several streams, re-uses it without re-planning.  split_taxa / split_a as for sp_score_splits; sides of at
 * most 14 taxa.  The creator holds one reference; sp_plan_release drops one and frees the plan with the last. */
int sp_plan_create(sp_ctx* ctx, int n_taxa, const int32_t* split_taxa, const int32_t* split_a, int64_t n_splits,
                   sp_plan** out);
int sp_plan_retain(sp_plan* plan);
int sp_plan_release(sp_plan* plan);
int sp_plan_info(const sp_plan* plan, int* n_taxa, int64_t* n_splits);

/* Flattening + split score of every split of `plan` for each of the n_al alignments (count tables of the plan's taxa,
 * at most 65535 table rows, at most 16 taxa: the sparse route), enqueued on the stream of `lane` with NO host
 * synchronisation.  `lane` is any context of the alignments' device - it supplies the stream and the work memory, so
 * several lanes (one context per HIP stream) keep several calls in flight; plan and alignments are only read.
 * The whole hand-back chain runs on the device: behind the in-LDS kernel a second kernel of persistent workgroups
 * picks - from the status words the first one left - the splits that did not fit (re-run in the lists-in-global form,
 * then the all-global form) and those whose 4-wide block found no certified spectral gap (8-wide block).  On completion
 * of the stream work every score is certified or flagged: status bit 0 = no certificate (score = upper estimate; finish
 * it with sp_finish_flagged), bits 8.. = operator applications, bit 1 never set.  scores_dev / status_dev: n_al * n_splits entries,
 * alignment-major.  The first call for an alignment prepares its split-independent metadata (synchronously, once). */
int sp_score_plan_async(sp_ctx* lane, sp_alignment* const* als, int n_al, sp_plan* plan, void* scores_dev,
                        void* status_dev);

/* n_steps passes of sp_score_plan_async from ONE host call (ABI 3): pass s - the whole flattening + score + hand-back
 * chain over every (alignment, split) item, nothing cached between passes - writes its n_al * n_splits scores to
 * (char*)scores_dev + s * scores_step_bytes and its status words to (char*)status_dev + s * status_step_bytes.  This is
 * the host loop `for s in range(n_steps): sp_score_plan_async(...)` (the repeated scoring of one table: bootstrap
 * replicates of the reference's README loop, README.md:36-41, or a benchmark's steps) without a Python / ctypes round
 * trip per pass: two kernel launches of host time per pass.  Same argument checks and errors as sp_score_plan_async. */
int sp_score_plan_steps(sp_ctx* lane, sp_alignment* const* als, int n_al, sp_plan* plan, int n_steps, void* scores_dev,
                        int64_t scores_step_bytes, void* status_dev, int64_t status_step_bytes);

/* ---------------------------------------------------------------- direct solver (ABI 4) --- */
/* The iterative routes CERTIFY a score (a spectral gap behind the block bounds what the Ritz sum still lacks) or FLAG it
 * (status bit 0: an upper estimate).  Flagged splits are finished by a direct method - fp64 Gram matrix over the split's
 * smaller side, Householder tridiagonalisation, Sturm-count multisection: no start vector, no stop rule, the counterpart of
 * the LAPACK gesdd call the reference's dense path makes for EVERY matrix (phylogenetics.py:281-285).  The synchronous
 * entry points (sp_score_splits, sp_score_all_splits[_shard], sp_score_matrix_f64, sp_score_coo_f64) run it themselves;
 * this is the host step behind the asynchronous ones: scores_host / status_host are the fetched results of a pass over
 * (al, the n_splits splits); every split whose status word has bit 0 or bit 1 set is re-scored and both arrays are patched
 * in place.  A finished split's status word: bit 2 set, bits 0 / 1 clear, bits 8.. = rows of the solved Gram matrix.
 * Limits: smaller side of at most `direct_max_rows` compact rows (context option, default 16384: 2 GB of Gram matrix; measured 21 ms at 1024 rows, 0.39 s at 4096, i.e. ~25 s at the limit)
 * and sides of at most 14 taxa; a split beyond them keeps its flagged estimate.  n_finished: may be NULL.
 * Context option `direct_finish` = 0 switches the finisher off everywhere (flagged splits then surface as SP_ENOCONV). */
int sp_finish_flagged(sp_alignment* al, const int32_t* split_taxa, const int32_t* split_a, int64_t n_splits,
                      double* scores_host, int32_t* status_host, int64_t* n_finished);

/* ---------------------------------------------------------------- one process, all GPUs of the node (ABI 4) --- */
/* SURVEY 8b / 8e, north_star: "partitioned across the 8 GPUs of one node by sharding the per-tree candidate-split set with an
 * RCCL all-gather of scores over xGMI".  An sp_node owns one context per device and one RCCL communicator per device
 * (ncclCommInitAll; librccl is loaded with dlopen on first use - the library itself links only the HIP runtime).
 * sp_node_score_all_splits replicates the pattern table on every device (a few MB), lets device r enumerate and score the
 * combinations r, r + P, ... of every size class (sp_score_all_splits_shard: an equal share of every cost class, no split
 * list anywhere; one host thread per device), all-gathers the packed (scores, status) shards in ONE collective and returns
 * all scores in the order of the reference's all_splits (splits.py:39-59).  keys / weights / counts / D / n_taxa / N as for
 * sp_alignment_create, method / trivial / size / outputs / SP_ENOCONV as for sp_score_all_splits.  n_devices = 0: every
 * visible device.  n_devices < 0 is a TEST MODE: |n_devices| ranks emulated on device 0, the collective replaced by device
 * copies - the sharding, packing and un-permuting of P > 1 ranks run on a one-GPU box (no RCCL involved).  (bench.py and splitp_amd.batch drive the same partition with one PROCESS per GPU through
 * torch.distributed, as the task's launch contract prescribes; this entry is for hosts that are not Python.) */
typedef struct sp_node sp_node;
int sp_node_create(int n_devices, sp_node** out);
int sp_node_destroy(sp_node* node);
int sp_node_info(const sp_node* node, int* n_devices);
int sp_node_score_all_splits(sp_node* node, const uint64_t* keys, const double* weights, const int64_t* counts, int64_t D,
                             int n_taxa, int64_t N, int method, int trivial, int size, int64_t* n_splits,
                             double* scores_host, int32_t* status_host);

/* ---------------------------------------------------------------- test entry --- */
/* The library's stable segmented LSD radix sort (csrc/radix_sort.h: one-sweep, decoupled look-back) on host arrays, for
 * tests only: n_seg independent segments of seg_len keys each, sorted on the bits [0, end_bit); keys as 64-bit words
 * (key_bytes = 4: narrowed to 32 bits on the device), vals_host / vals_out optional 32-bit values carried with the keys. */
int sp_debug_radix_sort(sp_ctx* ctx, const uint64_t* keys_host, const uint32_t* vals_host, int key_bytes, int64_t seg_len,
                        int64_t n_seg, unsigned end_bit, uint64_t* keys_out, uint32_t* vals_out);

#ifdef __cplusplus
}
#endif
#endif /* SPLITP_HIP_H */
