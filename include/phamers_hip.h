/*
 * phamers_hip.h -- C ABI of libphamers_hip.so, the MI355X (gfx950) k-mer count +
 * phage-score path behind PhaMers' Python hot-path functions.
 *
 * The reference (jondeaton/PhaMers, Python 2.7) has no FFI / plugin interface: its
 * boundary for this path is a set of Python call signatures.  Each entry point below
 * names the reference function (file:line, relative to the PhaMers tree) whose
 * arithmetic it replaces; phamers_amd/{kmer,learning,phamer}.py bind them with ctypes
 * and reproduce the Python-level names, defaults, return shapes and soft-failure
 * behaviour (INTEGRATION.md shows the stub a PhaMers maintainer would add).
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  Every function returns
 *     PHK_OK (0) or a negative PHK_ERR_* code; phk_last_error() gives the text
 *     (thread-local).
 *   - the library never allocates caller-visible memory: outputs are caller-allocated.
 *   - "host API" functions take HOST pointers, are synchronous, and stage through a
 *     per-context device workspace.  "device API" functions (suffix _dev) take DEVICE
 *     pointers (hipMalloc / torch data_ptr), enqueue on the context's HIP stream and
 *     return without synchronising (phk_sync() to wait).
 *   - one context per (process, GPU); calls on one context are serialised on its
 *     stream; distinct contexts may be used from distinct threads.
 *
 * Packed base stream (the device-side sequence format; this library's own layout)
 *   All contigs of a batch are concatenated into ONE stream of T bases with no padding.
 *   offsets[n+1] (uint64, in bases, offsets[0] = 0, offsets[n] = T) delimits contig c as
 *   stream positions [offsets[c], offsets[c+1]).
 *   packed : uint32 words, ceil(T/16)+1 of them (one zero pad word).  Base g is the 2-bit
 *            field at bits [30-2*(g%16), 31-2*(g%16)] of word g/16 -- the FIRST base of a
 *            word sits in the MOST significant bits, so a k-mer read off the word is
 *            already the reference's bin index (first base = most significant base-4
 *            digit, scripts/kmer.py:50).  Codes follow the symbols string: for the
 *            default 'ATGC' A=0 T=1 G=2 C=3 (scripts/kmer.py:28).
 *   mask   : optional validity bits, uint32 words, ceil(T/32)+1 of them; bit 31-(g%32) of
 *            word g/32 is 1 when base g is one of the 4 symbols (case-sensitive) and 0
 *            otherwise (the reference's '-', scripts/kmer.py:190-191).  NULL = all valid.
 *   A window is counted iff all k of its bases lie inside one contig and are valid
 *   (scripts/kmer.py:47-50).
 */
#ifndef PHAMERS_HIP_H
#define PHAMERS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): phk_kmeans_lloyd reports the smallest assignment gap; phk_batch_from_fasta_part; knobs, see phk_set_option */
#define PHK_ABI_VERSION 2

#define PHK_OK 0
#define PHK_ERR_ARG (-1)         /* bad argument (null pointer, bad k, bad shape) */
#define PHK_ERR_HIP (-2)         /* HIP runtime error; text has hipGetErrorString */
#define PHK_ERR_NOMEM (-3)       /* device allocation failed */
#define PHK_ERR_UNSUPPORTED (-4) /* outside the implemented path (e.g. k > PHK_MAX_K) */
#define PHK_ERR_NAN (-5)         /* a query row contains NaN (zero-count contig), see phk_score_* */
#define PHK_ERR_IO (-6)          /* file cannot be opened / read (phk_fasta_read) */

#define PHK_MAX_K 7              /* 4^k uint32 bins must fit one wave's LDS histogram */

#define PHK_METHOD_KNN 1         /* scripts/phamer.py:268-273 */
#define PHK_METHOD_KMEANS 2      /* scripts/phamer.py:240-256 (centroids supplied) */
#define PHK_METHOD_COMBO 3       /* scripts/phamer.py:303-313 */

typedef struct phk_ctx phk_ctx;
typedef struct phk_model phk_model;
typedef struct phk_fasta phk_fasta;
typedef struct phk_batch phk_batch;

/* ---- library / context ------------------------------------------------------------- */
int phk_abi_version(void);
const char *phk_last_error(void);
int phk_device_count(int *count);
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), or
 * NULL to let the context create and own a stream. */
int phk_create(int device_id, void *stream, phk_ctx **out);
int phk_destroy(phk_ctx *ctx);
int phk_sync(phk_ctx *ctx);
/* Tuning / diagnostic knobs of a context.  Each knob takes its initial value from the environment variable
 * PHK_<KEY upper case> ONCE, when the context is created; afterwards it changes only through this call (no
 * launch path reads the environment).  Keys: "count_lanes" ("0" wave-per-contig count kernel only, "f" the slot
 * kernel whatever the batch looks like, "P" / "p" the two-windows-per-add kernel with 1024 / 512 threads where the
 * batch's statistics allow it, "Q" / "q" the same forced), "count_sort" ("0" = no
 * length-bucketed order for ragged batches), "force_exact" ("1" = float64 scoring path for every model), "proposal"
 * ("" default: high parts only at k = 4, the two-part int8 sweep at k = 5, 6 | "f16" split-query kernel | "hi" / "cxf" /
 * "i83": the general-D alternatives), "cx_cfg" ("24" | "28": workgroup shape of the k = 4 sweep), "rerank" ("w" | "g"),
 * "score_batch" (queries per scoring batch), "tail_aside" ("0" = a multi-batch call keeps every batch's queue tail on
 * the main stream), "gen_groups" / "gen_seq" / "i8_insert" (general-D sweep shapes), "ws_fail" (fault injection for
 * the tests: the n-th workspace request from now fails with PHK_ERR_NOMEM; not read from the environment).
 * Unknown key -> PHK_ERR_ARG.  The results of every entry point are the same under every setting; the parity tests
 * use the knobs to cross-check the paths.  (ABI 2 dropped "count_cfg", "slot_threads", "pipeline" and the proposal
 * values "f32" / "cx2" with the kernels behind them.) */
int phk_set_option(phk_ctx *ctx, const char *key, const char *value);

/* device buffers for hosts that do not bring their own allocator */
int phk_malloc(phk_ctx *ctx, uint64_t bytes, void **dptr);
int phk_free(phk_ctx *ctx, void *dptr);
int phk_memcpy_h2d(phk_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes);
int phk_memcpy_d2h(phk_ctx *ctx, void *dst_host, const void *src_dev, uint64_t bytes);

/* ---- host API: counting ------------------------------------------------------------ */
/* kmer.count_string / kmer.count (scripts/kmer.py:32-50, 82-111): n sequences given as
 * concatenated ASCII `bases` + offsets[n+1]; counts[n][4^k] int64 (NumPy's default int, as
 * the reference returns).  symbols: the 4 characters that code 0..3 (the reference's
 * `symbols` argument when it has exactly 4 characters; "ATGC" by default). */
int phk_count_ascii(phk_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n, int k,
                    const char *symbols4, int64_t *counts);

/* kmer.normalize_counts (scripts/kmer.py:209-221): out[r][j] = (double)counts[r][j] /
 * (double)sum_j counts[r][j]; a zero row gives NaN exactly as the reference does. */
int phk_normalize_i64(phk_ctx *ctx, const int64_t *counts, uint64_t n, uint64_t D, double *out);
/* same for an already-float matrix (row sum taken left to right in float64) */
int phk_normalize_f64(phk_ctx *ctx, const double *rows, uint64_t n, uint64_t D, double *out);

/* transform_kmers.transform_kmers (scripts/transform_kmers.py:68-88): out[r][j] = rows[r][perm[j]]; the
 * reverse / complement / reverse-complement count vectors are column permutations (perm built by the
 * host facade). */
int phk_permute_columns_i64(phk_ctx *ctx, const int64_t *rows, uint64_t n, uint64_t D, const uint32_t *perm,
                            int64_t *out);

/* ---- host API: FASTA ingest --------------------------------------------------------- */
/* One multi-threaded pass over a FASTA file (plain, or gzip when the name ends in ".gz") replacing
 * the Biopython passes of kmer.count_file (scripts/kmer.py:124-140) and fileIO.read_fasta /
 * get_fasta_ids / get_fasta_sequences (scripts/fileIO.py:28-94).  Record semantics are Bio.SeqIO's:
 * title = '>' line minus '>' and trailing white space (record.id = its first word); sequence = the
 * following lines with trailing white space, ' ' and '\r' removed.  threads <= 0 = automatic.
 * Returns PHK_ERR_IO when the file cannot be read (count_file then returns (None, None),
 * scripts/kmer.py:126-128). */
int phk_fasta_read(const char *path, int threads, phk_fasta **out);
/* Titles, ids and sequence LENGTHS only (offsets as phk_fasta_read gives them; phk_fasta_data returns NULL for the bases
 * and the handle cannot be counted): what phamer_scorer.screen_by_length needs when the features came from the cache
 * (scripts/phamer.py:144-157 parses the whole file again for the lengths).  One pass over the file, no sequence buffer. */
int phk_fasta_index(const char *path, int threads, phk_fasta **out);
/* The same pass over ONE rank's share of the file (the multi-GPU form of kmer.count_file's loop, scripts/kmer.py:
 * 124-140 with fileIO.get_fasta_ids scripts/fileIO.py:62-77: records are independent, so the file is cut by bytes):
 * the records whose '>' line begins in bytes [byte_lo, byte_hi) of the file (of the decompressed stream for ".gz").
 * Ranges that tile [0, size) give every record to exactly one range wherever the cuts fall (inside a sequence line, a
 * title, at a '>' that does not begin a line); byte_hi past the end = to the end.  Only the range's own bytes are
 * read from a plain file.  phk_fasta_read_part: range `part` of `n_parts` equal byte ranges. */
int phk_fasta_read_range(const char *path, uint64_t byte_lo, uint64_t byte_hi, int threads, phk_fasta **out);
int phk_fasta_read_part(const char *path, uint32_t part, uint32_t n_parts, int threads, phk_fasta **out);
int phk_fasta_shape(const phk_fasta *f, uint64_t *n_records, uint64_t *total_bases, uint64_t *title_bytes);
/* borrowed pointers, valid until phk_fasta_free: concatenated sequence bytes + offsets[n+1] (exactly the
 * arguments of phk_count_ascii), concatenated titles + title_offsets[n+1] */
int phk_fasta_data(const phk_fasta *f, const char **bases, const uint64_t **offsets, const char **titles,
                   const uint64_t **title_offsets);
int phk_fasta_free(phk_fasta *f);

/* FASTA header -> PhaMers id (id_parser.get_id, scripts/id_parser.py:89-100, with get_contig_id :18-30,
 * get_phage_id :71-77, get_bacteria_id :57-68, is_genbank_id :80-86).  *status says what the reference does with
 * the header: PHK_ID_OK (the id is written to id_out, *id_len bytes, no terminator), PHK_ID_INDEX_ERROR (the
 * reference raises IndexError: a header that matches none of its three shapes), PHK_ID_NONE (it returns None). */
#define PHK_ID_OK 0
#define PHK_ID_INDEX_ERROR 1
#define PHK_ID_NONE 2
int phk_parse_id(const char *header, uint64_t header_len, char *id_out, uint64_t cap, uint64_t *id_len, int *status);
/* the ids of a parsed file, computed by the reader's threads from each record.id (what fileIO.get_fasta_ids
 * returns, scripts/fileIO.py:62-77): concatenated ids + id_offsets[n+1] + id_status[n]; borrowed pointers */
int phk_fasta_ids(const phk_fasta *f, const char **ids, const uint64_t **id_offsets, const uint8_t **id_status);
/* the same as a [n][width] zero-padded byte matrix (a NumPy 'S<width>' array) in caller memory */
int phk_fasta_ids_fixed(const phk_fasta *f, uint64_t width, char *out);
/* kmer.count_file's counting loop (scripts/kmer.py:135-139) on a parsed file: counts[n][4^k] int64 */
int phk_count_fasta(phk_ctx *ctx, const phk_fasta *f, int k, const char *symbols4, int64_t *counts);

/* ---- on-disk formats either side of the path (host only) ------------------------------- */
/* fileIO.save_counts (scripts/fileIO.py:169-181): `prefix` (the '# ' header block, NUL-terminated, may be NULL) then
 * one "id,c0,c1,...\n" line per row; counts are uint32 (elem_bytes 4) or int64 (8), [n][D].  ids: concatenated bytes
 * + id_offsets[n+1].  Formatted on all cores; byte-identical to the reference's np.savetxt output. */
int phk_write_counts_csv(const char *path, const char *prefix, const char *ids, const uint64_t *id_offsets,
                         const void *counts, int elem_bytes, uint64_t n, uint64_t D);
/* fileIO.save_phamer_scores (scripts/fileIO.py:241-253): "id, score\n" lines, the score written as
 * str(numpy.float64) writes it (shortest round-trip digits, Python's positional / scientific rule). */
int phk_write_scores_csv(const char *path, const char *prefix, const char *ids, const uint64_t *id_offsets,
                         const double *scores, uint64_t n);
/* the same two writers with the ids given as a NumPy 'U<id_width>' array as it lies in memory ([n][id_width] UCS-4 code
 * points, NUL padded; written as str(id).encode('latin-1', 'replace')): no per-id Python work for 10^6 rows */
int phk_write_counts_csv_ucs4(const char *path, const char *prefix, const uint32_t *ids, uint64_t id_width,
                              const void *counts, int elem_bytes, uint64_t n, uint64_t D);
int phk_write_scores_csv_ucs4(const char *path, const char *prefix, const uint32_t *ids, uint64_t id_width,
                              const double *scores, uint64_t n);
/* one float64 in that notation (NUL-terminated) */
int phk_format_float(double v, char *out, int cap);
/* fileIO.read_feature_file (scripts/fileIO.py:134-166) for files of the shape save_counts writes: '#' comment lines
 * and blank lines skipped, every other line "id,int,int,...".  phk_features_open maps the file and indexes its rows
 * (n rows, D count columns, the longest id in bytes); phk_features_read parses them on all cores into counts[n][D]
 * int64 and ids[n][id_width] (zero padded: a NumPy 'S<id_width>' array).  Anything np.loadtxt would treat differently
 * -- a '#' inside a line, a row with another number of fields, a field that is not a plain decimal integer, a
 * non-ASCII id -- gives PHK_ERR_UNSUPPORTED, and the caller reads the file the reference's way (np.loadtxt). */
typedef struct phk_features phk_features;
int phk_features_open(const char *path, phk_features **out, uint64_t *n, uint64_t *D, uint64_t *id_width);
int phk_features_read(const phk_features *f, int64_t *counts, char *ids, uint64_t id_width);
int phk_features_close(phk_features *f);

/* ---- device-resident contig batches (the facade's data path) ------------------------- */
/* What a PhaMers user calls on a FASTA input -- phamer_scorer.load_data + score_points (scripts/phamer.py:131, 139,
 * 579), kmer.count_file (scripts/kmer.py:114-140) -- with every intermediate kept on the device.  A batch is built
 * from host sequence bytes (uploaded once, in chunks through pinned staging buffers, copies overlapped with the packer) and owns
 * counts[n][4^k] uint32 + row sums in HBM; counts come back only when asked for (the features cache,
 * scripts/phamer.py:132-134), scores are the only per-call download. */
int phk_batch_from_ascii(phk_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n, int k,
                         const char *symbols4, phk_batch **out);
int phk_batch_from_fasta(phk_ctx *ctx, const phk_fasta *f, int k, const char *symbols4, phk_batch **out);
/* FASTA file -> device-resident batch in one call (kmer.count_file as load_data uses it, scripts/kmer.py:114-140,
 * scripts/phamer.py:131): the records are measured in one pass over the file and their sequences then written straight into
 * the upload's pinned staging buffers, chunk by chunk, overlapped with the bus and the packer -- no host copy of the
 * sequences exists.  *index_out: titles, ids and lengths as phk_fasta_index gives them (free with phk_fasta_free). */
int phk_batch_from_fasta_file(phk_ctx *ctx, const char *path, int k, const char *symbols4, int threads,
                              phk_fasta **index_out, phk_batch **out);
/* The same for one rank's share of the file: the records whose '>' line begins in byte range `part` of `n_parts` equal
 * ranges (the ranges of phk_fasta_read_part).  What rank `part` of `python -m phamers_amd.phamer --gpus n_parts` loads
 * (scripts/phamer.py:131 on a shard; contigs are independent, scripts/kmer.py:102-105). */
int phk_batch_from_fasta_part(phk_ctx *ctx, const char *path, uint32_t part, uint32_t n_parts, int k,
                              const char *symbols4, int threads, phk_fasta **index_out, phk_batch **out);
/* A batch from a count matrix on the host -- the features cache read back by fileIO.read_feature_file
 * (scripts/phamer.py:132-136, scripts/fileIO.py:134-166): counts[n][D] int64 row-major, D = 4^k.  The run then scores from
 * the same resident integers as one that counted the FASTA file (the reference normalises the cached counts to float rows
 * first: the scores are equal to rounding, and equal bit for bit to the cold run's here).  total_bases of such a batch is 0.
 * PHK_ERR_UNSUPPORTED: D is not 4^k with k <= PHK_MAX_K, or an entry is negative or >= 2^32. */
int phk_batch_from_counts(phk_ctx *ctx, const int64_t *counts, uint64_t n, uint64_t D, phk_batch **out);
int phk_batch_shape(const phk_batch *b, uint64_t *n, uint64_t *D, uint64_t *total_bases, int *any_invalid);
/* borrowed device pointers (valid until phk_batch_free): counts[n][D] uint32, row sums[n] uint32 */
int phk_batch_device_ptrs(const phk_batch *b, const uint32_t **d_counts, const uint32_t **d_rowsums);
/* kmer.count_file's count matrix as the reference returns it (int64, scripts/kmer.py:130-139) */
int phk_batch_counts_i64(phk_ctx *ctx, const phk_batch *b, int64_t *counts);
/* the same as the device holds it (uint32; half the download, what the features-cache writer takes) */
int phk_batch_counts_u32(phk_ctx *ctx, const phk_batch *b, uint32_t *counts);
/* kmer.normalize_counts of the batch (scripts/kmer.py:209-221; zero row -> NaN), float64 [n][D] to the host */
int phk_batch_normalized(phk_ctx *ctx, const phk_batch *b, double *rows);
/* rows[0..m) of a batch as a new batch: the row filter of phamer_scorer.screen_by_length (scripts/phamer.py:
 * 144-157) as a device gather */
int phk_batch_select(phk_ctx *ctx, const phk_batch *b, const uint64_t *rows, uint64_t m, phk_batch **out);
/* Column sums of a batch, sums[4^k] int64 to the host: kmer.count_directory's per-file np.sum(file_counts, axis=0)
 * (scripts/kmer.py:170-173), which is how a reference matrix row is regenerated from a genome file. */
int phk_batch_column_sums(phk_ctx *ctx, const phk_batch *b, int64_t *sums);
/* transform_kmers.transform_kmers (scripts/transform_kmers.py:68-88) on resident counts: a new batch with
 * out[r][j] = counts[r][table[j]] (table: 4^k host entries, each < 4^k; the reference's tables are not permutations, so
 * the row sums are recomputed on the device). */
int phk_batch_gather_columns(phk_ctx *ctx, const phk_batch *b, const uint32_t *table, phk_batch **out);
/* phamer_scorer.score_points (scripts/phamer.py:177-195) on a batch: scores[n] float64 to the host.  PHK_ERR_NAN
 * when a row has no counted window (the reference's NaN row makes scikit-learn raise). */
int phk_batch_score(phk_ctx *ctx, const phk_model *model, const phk_batch *b, int method, double *scores);
int phk_batch_free(phk_ctx *ctx, phk_batch *b);

/* ---- scoring model ----------------------------------------------------------------- */
/* The training side of phamer_scorer.score_points (scripts/phamer.py:177-195): positive /
 * negative reference rows (float64, normalised, row-major [n][D]); train = vstack(pos, neg),
 * labels = 1 for pos rows, 0 for neg rows (scripts/phamer.py:186-187).  cpos / cneg are the
 * k-means centroids of the two classes (scripts/phamer.py:245-248, learning.get_centroids
 * scripts/learning.py:69-81) and may be NULL / 0 when only PHK_METHOD_KNN will be used.
 * kn = k_neighbors (scripts/phamer.py:79, default 3).  All pointers are HOST pointers; the
 * model keeps device copies. */
int phk_model_create(phk_ctx *ctx, const double *pos, uint64_t n_pos, const double *neg,
                     uint64_t n_neg, const double *cpos, uint64_t n_cpos, const double *cneg,
                     uint64_t n_cneg, uint64_t D, int kn, phk_model **out);
int phk_model_destroy(phk_ctx *ctx, phk_model *model);
/* Cross-validation service (cross_validator.cross_validate, scripts/cross_validate.py:57-101: N folds whose train
 * sets share (N-1)/N of the rows): ONE model holds every reference row; a fold is a column mask over its train rows
 * (mask[n_pos + n_neg] host bytes in vstack(pos, neg) order, non-zero = held out, excluded from the k-NN search;
 * NULL lifts the mask) plus that fold's centroids (same counts as at creation).  Scores then equal those of a model
 * built from the unmasked rows alone: the search is translation invariant, so keeping the full matrix's centring
 * vector changes nothing but the error bounds, which are evaluated for it. */
int phk_model_set_centroids(phk_ctx *ctx, phk_model *model, const double *cpos, uint64_t n_cpos, const double *cneg,
                            uint64_t n_cneg);
int phk_model_set_column_mask(phk_ctx *ctx, phk_model *model, const uint8_t *mask);

/* Deterministic device k-means (opt-in alternative to the scikit-learn fit of scripts/learning.py:131-146,
 * whose centroids depend on the scikit-learn version): k-means++ seeding driven by splitmix64(seed + j),
 * Lloyd sweeps in float64 with index-ordered sums (bit-reproducible), empty clusters re-seeded with the
 * farthest point; stops when no label changes or after max_iter sweeps.  X[n][D] host float64 ->
 * centroids[k][D], labels[n] (may be NULL), number of sweeps (may be NULL).  Cluster c's centroid is the
 * mean of the points labelled c, as learning.get_centroids computes it (scripts/learning.py:69-81). */
int phk_kmeans(phk_ctx *ctx, const double *X, uint64_t n, uint64_t D, uint32_t k, uint64_t seed, int max_iter,
               double *centroids, uint32_t *labels, int *n_iter);
/* The Lloyd iteration of scikit-learn's KMeans.fit (scripts/learning.py:138; _kmeans_single_lloyd) on the device, from
 * GIVEN initial centres: E-step (float64 direct differences, ties to the lower centre), M-step, stop when no label changed
 * or when the summed squared centre shift is <= tol (scikit-learn: 1e-4 x the mean per-feature variance), in which case
 * one more E-step makes the labels match the final centres.  With the host-side k-means++ seeding of scikit-learn
 * (kmeans_plusplus on the mean-centred rows, RandomState(10): phamers_amd/learning.py) the labels -- and so the reference's
 * per-label means, learning.get_centroids -- equal those of KMeans(n_clusters=k, random_state=10).fit(X).
 * X[n][D], init[k][D] host float64 -> labels[n]; centres (may be NULL); sweeps; number of empty clusters met (scikit-learn
 * relocates an empty cluster, this entry does not: a caller that sees n_empty != 0 must use the host fit).
 * min_gap (may be NULL): the closest call any E-step made, min over points and sweeps of (d2_second - d2_best) / d2_second.
 * scikit-learn forms its distances as -2 x.c + |c|^2 in chunked matrix products, this entry by direct differences: the two
 * agree to ~1e-13 relative, so a label can differ only where min_gap is of that order -- the facade takes the host fit
 * below 1e-9 (phamers_amd/learning.py). */
int phk_kmeans_lloyd(phk_ctx *ctx, const double *X, uint64_t n, uint64_t D, uint32_t k, const double *init, double tol,
                     int max_iter, double *centres, uint32_t *labels, int *n_iter, int *n_empty, double *min_gap);

/* ---- host API: scoring ------------------------------------------------------------- */
/* phamer.score_points / phamer_scorer.score_points (scripts/phamer.py:451-468, 177-195) for
 * method in {knn, kmeans, combo}: Q[N][D] float64 host rows -> scores[N] float64.
 *   knn    : 2*(majority label of the kn nearest train rows) - 1   (scripts/learning.py:118-128)
 *   kmeans : tanh((e- - e+)/(e+ + e-)), e+/- = distance to the nearest positive / negative
 *            centroid (scripts/phamer.py:198-210, 250-256; scripts/learning.py:47-66)
 *   combo  : knn + kmeans (scripts/phamer.py:303-313)
 * Returns PHK_ERR_NAN (scores untouched) if any query element is NaN -- the reference's
 * scikit-learn call raises on such input. */
int phk_score(phk_ctx *ctx, const phk_model *model, const double *Q, uint64_t N, int method,
              double *scores);

/* learning.distances (scripts/learning.py:47-56; with np.argmin on its result: learning.closest_to :59-66): the
 * Euclidean distances of every row of Q[N][D] to every row of X[M][D], out[N][M], float64, in the reference's
 * direct-difference form sqrt(sum_d (q_d - x_d)^2).  Host pointers. */
int phk_distances(phk_ctx *ctx, const double *Q, uint64_t N, const double *X, uint64_t M, uint64_t D, double *out);

/* ---- device API -------------------------------------------------------------------- */
/* ASCII -> packed stream (+ mask).  d_any_invalid (one uint32, device) is set non-zero when
 * some base is not one of symbols4; it may be NULL. */
int phk_pack_ascii_dev(phk_ctx *ctx, const char *d_bases, uint64_t total_bases,
                       const char *symbols4, uint32_t *d_packed, uint32_t *d_mask,
                       uint32_t *d_any_invalid);
/* packed stream -> d_counts[n][4^k] uint32 and d_nwin[n] (= row sums = number of counted
 * windows; may be NULL).  d_mask may be NULL (all bases valid). */
int phk_count_dev(phk_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_mask,
                  uint64_t total_bases, const uint64_t *d_offsets, uint64_t n, int k,
                  uint32_t *d_counts, uint32_t *d_nwin);
int phk_normalize_dev(phk_ctx *ctx, const uint32_t *d_counts, uint64_t n, uint64_t D,
                      double *d_out);
/* scores for device-resident float64 rows / for device-resident uint32 count rows (the
 * rows are normalised on the fly exactly as kmer.normalize_counts would).  d_status: one
 * uint32 (device), set to the number of NaN (zero-count) rows seen; their scores are NaN. */
int phk_score_dev(phk_ctx *ctx, const phk_model *model, const double *d_Q, uint64_t N,
                  int method, double *d_scores, uint32_t *d_status);
int phk_score_counts_dev(phk_ctx *ctx, const phk_model *model, const uint32_t *d_counts,
                         uint64_t N, int method, double *d_scores, uint32_t *d_status);
/* the whole hot path on device-resident input: count -> normalise -> score.  d_counts
 * ([n][4^k] uint32) receives the materialised counts. */
int phk_count_score_dev(phk_ctx *ctx, const phk_model *model, const uint32_t *d_packed,
                        const uint32_t *d_mask, uint64_t total_bases, const uint64_t *d_offsets,
                        uint64_t n, int k, int method, uint32_t *d_counts, double *d_scores,
                        uint32_t *d_status);

/* Verification on the device (a full-size batch of counts is tens of GB and is never brought to the host to be
 * checked): d_result[0] = number of rows of d_counts[n][D] whose sum differs from expected_rowsum (not evaluated when
 * expected_rowsum is UINT64_MAX), d_result[1] = number of words in which d_counts differs from d_other (not evaluated
 * when d_other is NULL).  d_result: two uint64 on the device.  A contig of L valid bases has L - k + 1 windows
 * (scripts/kmer.py:47), which is what the full-size tests pass as expected_rowsum. */
int phk_check_counts_dev(phk_ctx *ctx, const uint32_t *d_counts, const uint32_t *d_other, uint64_t n, uint64_t D,
                         uint64_t expected_rowsum, uint64_t *d_result);

/* Diagnostics of the most recent scoring call on this context, summed over its batches (synchronises the stream):
 * how many queries were resolved by the float64 brute-force fallback kernel and how many
 * (query, segment) orderings had to be decided by exact candidate distances rather than by
 * the certified fp32 margin.  Both are 0 on the all-float64 path. */
int phk_score_stats(phk_ctx *ctx, uint64_t *n_fallback, uint64_t *n_exact_resolved);
/* the same and more, out[0 .. n_out): [0] brute-forced queries, [1] orderings decided by exact distances, [2] queries
 * that took the second chance (split-query MFMA pass), [3..6] why the high-parts-only decision stage passed them on:
 * window wider than the refined candidates, window reaching past the lists, refined values too close, centroid
 * leader not certified; general D (k = 5, 6), int8 sweep: [7] rows re-swept alone with all three digits (the two-digit
 * window held more columns than the lists), [8] rows beyond the int8 operand, swept by the f16 count-exact kernel
 * (both are also counted in [2]) */
int phk_score_stats_ex(phk_ctx *ctx, uint64_t *out, int n_out);

/* Diagnostic of the arithmetic the MFMA proposal rests on: chains of v_mfma_f32_32x32x16_f16 on caller tiles.  Per tile
 * acc = C[32][32]; for s < steps: acc = A[s][32][16] (fp16 bits, row-major) x B[s][16][32] + acc, with D[tile][s][32][32] =
 * acc after step s.  Host pointers.  The parity tests use it to assert the per-instruction rounding charge of the
 * certification (2u (|acc_in| + sum |products|)) on adversarial inputs. */
int phk_mfma_f16_probe(phk_ctx *ctx, const uint16_t *A, const uint16_t *B, const float *C, uint64_t n_tiles, uint32_t steps,
                       float *D);

/* seeded synthetic batch generated on the device (phamers_amd/synth.py defines the hash):
 * n contigs of L bases, contig ids first_contig..first_contig+n-1; writes the packed stream,
 * the mask (if d_mask != NULL; required when invalid_ppm > 0) and offsets[n+1]. */
int phk_synth_packed_dev(phk_ctx *ctx, uint64_t seed, uint64_t first_contig, uint64_t n,
                         uint64_t L, uint32_t invalid_ppm, uint32_t *d_packed, uint32_t *d_mask,
                         uint64_t *d_offsets);

/* the ragged, composition-skewed variant (second bench workload): contig boundaries d_offsets[n+1] (device, any
 * lengths, d_offsets[n] = total_bases) are the caller's; every contig draws its own GC fraction
 * 1/2 +- gc_spread_permille / 2000 and bases from one hash per base (phamers_amd/synth.py: synth_ragged_codes). */
int phk_synth_ragged_dev(phk_ctx *ctx, uint64_t seed, uint64_t first_contig, uint64_t n, const uint64_t *d_offsets,
                         uint64_t total_bases, uint32_t gc_spread_permille, uint32_t invalid_ppm, uint32_t *d_packed,
                         uint32_t *d_mask);

/* ---- in-library kernel timing (HIP events on the context's stream) ------------------ */
int phk_profile_enable(phk_ctx *ctx, int on);
int phk_profile_reset(phk_ctx *ctx);
/* number of distinct kernels timed since the last reset */
int phk_profile_count(phk_ctx *ctx, int *count);
/* idx-th kernel: name (NUL-terminated, truncated to cap), total milliseconds, launches */
int phk_profile_get(phk_ctx *ctx, int idx, char *name, int cap, double *total_ms,
                    uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* PHAMERS_HIP_H */
