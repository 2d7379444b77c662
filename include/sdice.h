/*
 * sdice.h -- C ABI of libsplicedice_hip.so: the MI355X (gfx950) hot path of splicedice.
 *
 * The reference (BrooksLabUCSC/splicedice) is pure Python and has NO FFI/plugin
 * interface; its only extension seam is add_cmd(name, add_parser, run_with)
 * (splicedice/__main__.py:17-33).  This header is therefore the boundary a maintainer
 * would bind with ctypes from inside the four hot `run_with` functions; each entry
 * point cites the reference loop it replaces (file:line relative to the reference
 * checkout).  See INTEGRATION.md for the ctypes stub.
 *
 * Conventions
 *  - plain C, every function returns int status (0 = ok, <0 = error);
 *    sdice_last_error() returns a thread-local message; no exceptions cross the ABI.
 *  - host entry points (no suffix): caller owns all host buffers (C-contiguous,
 *    pointer + explicit sizes); the call copies in, runs the HIP kernels, copies out and
 *    is synchronous on return.
 *  - device entry points (`_dev`): all pointers are device pointers obtained from
 *    sdice_dmalloc; work is enqueued on the context's HIP stream and NOT synchronised
 *    (call sdice_sync).  These keep tables resident in HBM between stages.
 *  - a context is bound to one GPU and used from one thread at a time.
 *  - there is no CPU backend: without a usable HIP device sdice_ctx_create fails.
 */
#ifndef SDICE_H
#define SDICE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdice_ctx sdice_ctx;

#define SDICE_OK          0
#define SDICE_ERR_ARG    -1
#define SDICE_ERR_HIP    -2
#define SDICE_ERR_NOMEM  -3
#define SDICE_ERR_STATE  -4
#define SDICE_ERR_COMM   -5

#define SDICE_ABI_VERSION 1

/* ---- lifecycle / errors ---------------------------------------------------------- */
int         sdice_version(void);
const char* sdice_last_error(void);
int         sdice_ctx_create(int device_ordinal, sdice_ctx** out);
int         sdice_ctx_destroy(sdice_ctx* ctx);
int         sdice_sync(sdice_ctx* ctx);
/* synchronise and hand the context's cached device scratch (the per-call arena, kept between calls) back to the
 * driver; a long-lived context calls it after an unusually large job (no reference counterpart: the reference's
 * numpy temporaries die with each call, e.g. pairwise_fisher.py:187-191) */
int         sdice_trim(sdice_ctx* ctx);
/* name: caller buffer of name_cap bytes; any out pointer may be NULL */
int         sdice_device_info(sdice_ctx* ctx, char* name, int name_cap, int* compute_units,
                              int64_t* hbm_bytes);

/* ---- device memory (for the _dev entry points) ------------------------------------ */
int sdice_dmalloc(sdice_ctx* ctx, int64_t bytes, void** dptr);
int sdice_dfree(sdice_ctx* ctx, void* dptr);
int sdice_h2d(sdice_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
int sdice_d2h(sdice_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);
int sdice_dmemset(sdice_ctx* ctx, void* dptr, int value, int64_t bytes);

/* ---- clustering: replaces SPLICEDICE.getClusters (SPLICEDICE.py:230-255), its twin
 *      counts_to_ps.determine_clusters (counts_to_ps.py:16-41) and the junctionIndex
 *      sort (SPLICEDICE.py:96).
 *  in : n junctions as parallel arrays, any order, all distinct (a duplicate is reported as
 *       SDICE_ERR_ARG: the reference holds the junctions in a set / dict, SPLICEDICE.py:156).
 *       chrom_rank = dense rank of the chromosome name under Python string sort,
 *       0 <= left <= right < 2^31, strand 0 = '+', 1 = '-'.
 *  out: row_of[n]   output row of input junction i = its rank in
 *                   (chrom, left, right, strand) order (SPLICEDICE.py:96);
 *       row_ptr[n+1], col[nnz]: per output row, the rows of all junctions on the same
 *                   chromosome+strand whose closed interval overlaps it
 *                   (prior.right >= cur.left, SPLICEDICE.py:250), in the reference's
 *                   list order: earlier-in-sweep overlaps most recent first, then
 *                   later ones in sweep order.
 *  The column list lives in the context until the next sdice_cluster* call and is
 *  fetched with sdice_cluster_col (host) / sdice_cluster_col_dev (device pointer).
 */
int sdice_cluster(sdice_ctx* ctx, int64_t n, const int32_t* chrom_rank, const int32_t* left,
                  const int32_t* right, const int8_t* strand,
                  int32_t* row_of, int64_t* row_ptr, int64_t* nnz);
int sdice_cluster_col(sdice_ctx* ctx, int32_t* col, int64_t capacity);
/* Device variant.  nnz != NULL: synchronous on return (*nnz = list length; invalid or duplicate
 * junctions are reported here).  nnz == NULL: ASYNCHRONOUS -- the whole chain (sort, row order,
 * lists) is enqueued with no host round trip, so that a dependent sdice_ps_dev can be enqueued right
 * behind it; validation results, nnz and the list capacity check are then reported by the next
 * sdice_sync / sdice_cluster_status / sdice_cluster_col_dev(nnz != NULL) (deferred-error semantics,
 * as for asynchronous HIP work).  A chain that failed leaves row_ptr all zero, so dependent launches
 * stay in bounds.  The list buffer holds at least 16 entries per junction; one synchronous call sizes
 * it for denser data. */
int sdice_cluster_dev(sdice_ctx* ctx, int64_t n, const int32_t* d_chrom_rank,
                      const int32_t* d_left, const int32_t* d_right, const int8_t* d_strand,
                      int32_t* d_row_of, int64_t* d_row_ptr, int64_t* nnz /* host out, may be NULL */);
/* resolves a pending asynchronous sdice_cluster_dev (synchronises); nnz / reach (largest
 * |row(neighbour) - row|) may be NULL */
int sdice_cluster_status(sdice_ctx* ctx, int64_t* nnz, int32_t* reach);
int sdice_cluster_col_dev(sdice_ctx* ctx, const int32_t** d_col, int64_t* nnz);

/* ---- PS: replaces SPLICEDICE.calculatePsi (SPLICEDICE.py:297-310), counts_to_ps
 *      writePsValues' arithmetic (counts_to_ps.py:63-68) and pairwise's exclusion
 *      gather (pairwise_fisher.py:158-160).
 *  counts[n,s] int32 row-major, rows in output-row order; CSR as above (any valid CSR
 *  over rows is accepted, e.g. one parsed from an _allClusters.tsv).
 *  excl[n,s] int64 (optional, may be NULL): sum of the count rows of the neighbours.
 *  ps[n,s] float32 (optional, may be NULL) = float32(double(incl) / double(incl+excl));
 *  0/0 -> NaN.
 */
int sdice_ps(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* counts,
             const int64_t* row_ptr, const int32_t* col, int64_t* excl, float* ps);
int sdice_ps_dev(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* d_counts,
                 const int64_t* d_row_ptr, const int32_t* d_col, int64_t* d_excl, float* d_ps);

/* ---- PS of a float64 count table: counts_to_ps.writePsValues (counts_to_ps.py:58-70) as written --
 *      the table is parsed with dtype=float (:50), so fractional / normalised counts are legal there.
 *  counts[n_rows,s] float64 row-major; rows 0..n_out-1 get a result, rows n_out..n_rows-1 are
 *  sources only (overlaps that are named in a cluster list but are not cluster keys themselves);
 *  row_ptr[n_out+1], col[nnz] with 0 <= col < n_rows.
 *  ps[n_out,s] float64 = counts[r] / (counts[r] + counts[col[k0]] + counts[col[k0+1]] + ...), the sum
 *  taken left to right in list order, one IEEE addition at a time; 0/0 -> NaN, x/0 -> inf.
 */
int sdice_ps_f64(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* counts,
                 const int64_t* row_ptr, const int32_t* col, double* ps);
int sdice_ps_f64_dev(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* d_counts,
                     const int64_t* d_row_ptr, const int32_t* d_col, double* d_ps);

/* ---- exclusion sums of a float64 count table: `pairwise` on fractional counts.  pairwise_fisher.py:46-61 parses the
 *      table with dtype=float and :158-160 adds the rows named in the event's cluster, np.sum(counts[mask], axis=0):
 *      one IEEE addition per row in TABLE order; scipy.stats.fisher_exact then truncates the float sums (and the
 *      float inclusion counts) to int64.  Same shapes as sdice_ps_f64; excl[n_out,s] float64 =
 *      counts[col[k0]] + counts[col[k0+1]] + ... left to right (0.0 for an empty list).  The caller lists the rows of
 *      every event in ascending (table) order and truncates.
 */
int sdice_excl_f64(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* counts,
                   const int64_t* row_ptr, const int32_t* col, double* excl);
int sdice_excl_f64_dev(sdice_ctx* ctx, int64_t n_out, int64_t n_rows, int32_t s, const double* d_counts,
                       const int64_t* d_row_ptr, const int32_t* d_col, double* d_excl);

/* --lowCoverageNan (SPLICEDICE.py:307-309): ps[low_idx[i]] = NaN, flat indices r*s+c */
int sdice_mark_low(sdice_ctx* ctx, int64_t n_elems, float* ps, const int64_t* low_idx,
                   int64_t n_low);
int sdice_mark_low_dev(sdice_ctx* ctx, int64_t n_elems, float* d_ps, const int64_t* d_low_idx,
                       int64_t n_low);

/* the `_allPS.tsv` text round trip between quant and compare_sample_sets:
 * f'{x:.3f}' (SPLICEDICE.py:353) re-read as float32 (compareSampleSets.py:202) */
int sdice_quantize3(sdice_ctx* ctx, int64_t n_elems, float* ps_inout);
int sdice_quantize3_dev(sdice_ctx* ctx, int64_t n_elems, float* d_ps_inout);

/* ---- compare_sample_sets: replaces the per-row loop compareSampleSets.py:216-232
 *      (NaN drop, <3 skip, scipy.stats.ranksums, np.median x2, np.mean x2).
 *  ps[n,s] float32; g1[n1], g2[n2] column indices (table order).
 *  Outputs are un-compacted, one slot per row: tested[n] (1 = row was tested),
 *  p[n] float64, med1/med2/mean1/mean2/delta[n] float32 (0 where not tested);
 *  z[n] float64 optional (NULL to skip).  The host compacts in row order.
 */
int sdice_ranksum(sdice_ctx* ctx, int64_t n, int32_t s, const float* ps,
                  const int32_t* g1, int32_t n1, const int32_t* g2, int32_t n2,
                  uint8_t* tested, double* p, double* z, float* med1, float* med2,
                  float* mean1, float* mean2, float* delta);
int sdice_ranksum_dev(sdice_ctx* ctx, int64_t n, int32_t s, const float* d_ps,
                      const int32_t* d_g1, int32_t n1, const int32_t* d_g2, int32_t n2,
                      uint8_t* d_tested, double* d_p, double* d_z, float* d_med1, float* d_med2,
                      float* d_mean1, float* d_mean2, float* d_delta);

/* ---- pairwise: replaces the per-pair loop pairwise_fisher.py:164-179
 *      (scipy.stats.fisher_exact two-sided on [[incl_a, incl_b],[excl_a, excl_b]]).
 *  incl[n,s] int32, excl[n,s] int64 (from sdice_ps); p[n, s(s-1)/2] float64 row-major,
 *  pair order (0,1),(0,2)...(s-2,s-1) (pairwise_fisher.py:142-147).
 */
int sdice_fisher_pairs(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* incl,
                       const int64_t* excl, double* p);
int sdice_fisher_pairs_dev(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* d_incl,
                           const int64_t* d_excl, double* d_p);
/* measurement aid (no reference counterpart): with the context parameter "fisher.count_steps" = 1 the pair kernel
 * counts the lane-steps it issues and those that advanced a live walk inside its support; this returns the two
 * counts of the last sdice_fisher_pairs[_dev] call (bench.py: roofline.valu_f64.useful_lane_frac). */
int sdice_fisher_step_stats(sdice_ctx* ctx, uint64_t* useful, uint64_t* issued);
/* m independent 2x2 tables abcd[m,4] int64 -> p[m] (same kernel math; KAT entry point) */
int sdice_fisher_tables(sdice_ctx* ctx, int64_t m, const int64_t* abcd, double* p);

/* pairwise --chi2: replaces test_method = scipy.stats.chi2_contingency (pairwise_fisher.py:133-136,
 * :179): Yates-corrected 2x2 chi-square, p = chdtrc(1, chi2).  scipy raises ValueError on a zero
 * expected frequency (the reference run aborts); such tables get p = NaN and are counted in *n_bad
 * so that the caller can raise.  Same shapes and pair order as sdice_fisher_pairs. */
int sdice_chi2_pairs(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* incl,
                     const int64_t* excl, double* p, int64_t* n_bad);
int sdice_chi2_pairs_dev(sdice_ctx* ctx, int64_t n, int32_t s, const int32_t* d_incl,
                         const int64_t* d_excl, double* d_p, int64_t* d_n_bad /* device */);

/* ---- Benjamini-Hochberg: replaces statsmodels multipletests(p, method="fdr_bh")[1]
 *      (compareSampleSets.py:235; pairwise_fisher.py:185,190). */
int sdice_bh(sdice_ctx* ctx, int64_t m, const double* p, double* q);
int sdice_bh_dev(sdice_ctx* ctx, int64_t m, const double* d_p, double* d_q);
/* BH over the PRESENT entries of a padded vector: entry i is present when tested[i] != 0 (tested == NULL:
 * when p[i] >= 0); absent entries are left out of the ranking and of m and get q = 0.  The sharded
 * compare gathers the per-junction table of every rank (padded to equal length) and corrects it in
 * place on the device -- the reference compacts the tested rows first (compareSampleSets.py:223-235). */
int sdice_bh_masked_dev(sdice_ctx* ctx, int64_t n, const double* d_p, const uint8_t* d_tested, double* d_q);
/* BH down each of `cols` columns of a row-major [n, cols] table, in place
 * (pairwise_fisher.py:187-191) */
int sdice_bh_columns(sdice_ctx* ctx, int64_t n, int64_t cols, double* p_inout);
int sdice_bh_columns_dev(sdice_ctx* ctx, int64_t n, int64_t cols, double* d_p_inout);
/* the same for a table whose rows are `pitch` elements apart (a column range of a wider table), pitch >= cols */
int sdice_bh_columns_pitched_dev(sdice_ctx* ctx, int64_t n, int64_t cols, int64_t pitch, double* d_p_inout);

/* measurement aid (tools/bench_cli.py): out3 = seconds spent formatting, seconds spent writing, bytes written by
 * sdice_write_table on the calling thread since the last reset */
int sdice_textio_stats(double* out3, int reset);

/* ---- host-side table text I/O (no device, no context): multithreaded, byte-compatible with the
 *      reference's writers  f'{x:.3f}' (SPLICEDICE.py:353, counts_to_ps.py:69), f'{x:.0f}'
 *      (SPLICEDICE.py:340), str(numpy.float64) (pairwise_fisher.py:200)  and with its readers
 *      (text -> float64 -> dtype: compareSampleSets.py:202, pairwise_fisher.py:60, counts_to_ps.py:50).
 *  Tables are  header-line '\n'  then one line per row:  name '\t' v '\t' v ... '\n'.
 *  names: concatenated row names, name_off[n+1] byte offsets into it.
 *  dtype 0 float32, 1 float64, 2 int32;  mode 0 '%.3f', 1 '%.0f', 2 numpy str() shortest repr;
 *  mode | 0x100 appends to an existing file (tables streamed in row slabs). */
int sdice_write_table(const char* path, const char* header /* incl. '\n' */, int64_t n, int32_t s,
                      const char* names, const int64_t* name_off, const void* data, int dtype, int mode,
                      int threads /* 0 = all cores */);
/* Column-major variant: column c is cols[c] (n values) with its own dtype / mode -- the
 * compare_sample_sets output table mixes float32 and float64 numpy-repr columns
 * (compareSampleSets.py:252-270). */
int sdice_write_columns(const char* path, const char* header, int64_t n, const char* names, const int64_t* name_off,
                        int32_t ncols, const void* const* cols, const int32_t* dtypes, const int32_t* modes, int threads);
/* sdice_write_columns with a ready-made text suffix per row (written after the last numeric column): the
 * gene / overlapping / transcript_id columns of an annotated table (compareSampleSets.py:238-264). */
int sdice_write_columns_sfx(const char* path, const char* header, int64_t n, const char* names, const int64_t* name_off,
                            int32_t ncols, const void* const* cols, const int32_t* dtypes, const int32_t* modes,
                            const char* sfx, const int64_t* sfx_off, int threads);
/* The interval scan of the GTF annotation join (compareSampleSets.py:246-252): event e (group ev_group[e], -1 =
 * none; positions ev_a[e], ev_b[e]) matches interval k of its group (grp_ptr CSR over lo[] / hi[]) when
 * lo <= a <= hi or lo <= b <= hi.  out_idx == NULL: fill out_ptr[0..n_events] with the prefix sums of the match
 * counts; otherwise also write the matching interval indices (increasing k per event). */
int sdice_interval_overlaps(int64_t n_events, const int32_t* ev_group, const int64_t* ev_a, const int64_t* ev_b,
                            int32_t n_groups, const int64_t* grp_ptr, const int64_t* lo, const int64_t* hi,
                            int64_t* out_ptr, int64_t* out_idx, int64_t out_cap, int threads);
/* `<prefix>_allClusters.tsv` (SPLICEDICE.py:316-326): name<TAB>comma-joined names of the row's
 * neighbour list (CSR in row indices), one line per junction row. */
int sdice_write_clusters(const char* path, int64_t n, const char* names, const int64_t* name_off,
                         const int64_t* row_ptr, const int32_t* col, int threads);
/* `<prefix>_junctions.bed` (SPLICEDICE.py:316-321): chrom<TAB>left<TAB>right<TAB>chrom:left-right:strand<TAB>0<TAB>strand
 * per junction row; chrom[r] indexes the n_chrom names in chrom_names / chrom_off, strand[r] is the character. */
int sdice_write_junction_bed(const char* path, int64_t n, const char* chrom_names, const int64_t* chrom_off,
                             int32_t n_chrom, const int32_t* chrom, const int32_t* left, const int32_t* right,
                             const char* strand, int threads);
/* sdice_write_table keeps its threads' text buffers between calls (a table written in slabs); this frees them. */
int sdice_textio_trim(void);
/* Row names 'chrom:left-right:strand' (SPLICEDICE.py:312-314) of n junction rows as one byte string + off[n + 1], the
 * form the table writers above take their names in.  *need = bytes the names take; out_cap too small (or out NULL, to
 * ask for the size): SDICE_ERR_ARG with off and *need filled. */
int sdice_junction_names(int64_t n, const char* chrom_names, const int64_t* chrom_off, int32_t n_chrom,
                         const int32_t* chrom, const int32_t* left, const int32_t* right, const char* strand,
                         char* out, int64_t out_cap, int64_t* off, int64_t* need);
typedef struct sdice_table sdice_table;
int sdice_table_open(const char* path, sdice_table** out, int64_t* n, int32_t* s, int64_t* names_bytes,
                     int64_t* header_bytes);
int sdice_table_read(sdice_table* t, char* header, char* names, int64_t* name_off, void* data,
                     int dtype /* 0 float32, 1 float64 */, int threads);
int sdice_table_close(sdice_table* t);

/* ---- host-side junction-file parser (no device, no context): one pass over a sample file of
 *      `splicedice quant` with the reference's per-type admission filters
 *      (SPLICEDICE.getAllJunctions, SPLICEDICE.py:147-228) and every line's key + score for the
 *      count pass (SPLICEDICE.getJunctionCounts, SPLICEDICE.py:257-295).
 *  type: 0 bed / leafcutter, 1 splicedicebed, 2 STAR SJ.out.tab.
 *  Per line: chrom_id (index into the file's chromosome table, first-appearance order), left, right,
 *  strand (0 '+', 1 '-', 2 other), score, admit (1 = passes the admission filters).
 *  sdice_junc_lookup: row of every query junction in rows sorted by (chrom, left, right, strand), -1 if absent. */
typedef struct sdice_juncfile sdice_juncfile;
int sdice_junc_open(const char* path, int type, sdice_juncfile** out, int64_t* n_lines, int32_t* n_chroms,
                    int64_t* chrom_bytes);
int sdice_junc_read(sdice_juncfile* f, int32_t min_length, int32_t max_length, int32_t min_unique,
                    int32_t min_overhang, double min_entropy, int no_multimap, int32_t* chrom_id,
                    int32_t* left, int32_t* right, int8_t* strand, int64_t* score, uint8_t* admit,
                    char* chrom_names, int64_t* chrom_off /* [n_chroms+1] */, int threads);
int sdice_junc_close(sdice_juncfile* f);
int sdice_junc_lookup(int64_t n_rows, const int32_t* row_chrom, const int32_t* row_left,
                      const int32_t* row_right, const int8_t* row_strand, int64_t n_q,
                      const int32_t* q_chrom, const int32_t* q_left, const int32_t* q_right,
                      const int8_t* q_strand, int32_t* row_out, int threads);
/* One sample's column of the count table (SPLICEDICE.getJunctionCounts, SPLICEDICE.py:257-295: later lines of a file
 * overwrite earlier ones, no score filter): col[row] = score for every line whose junction is among the rows, in line
 * order.  col: the sample's int32 [n_rows] stretch of the TRANSPOSED table [sample][row], zeroed by the caller;
 * low (or NULL): set to 1 where a line with score < min_unique hits the row.  A final value outside [0, 2^31) is
 * SDICE_ERR_ARG ("junction counts must be non-negative and below 2**31", the reference's int conversion limit here).
 * Single-threaded: one call per sample, side by side.  sdice_transpose_i32: dst[c][r] = src[r][c] (threads).
 * sdice_host_threads: the default worker count of the host-side calls (hardware threads capped by the cgroup quota). */
int sdice_junc_count_column(int64_t n_rows, const int32_t* row_chrom, const int32_t* row_left,
                            const int32_t* row_right, const int8_t* row_strand, int64_t n_q,
                            const int32_t* q_chrom, const int32_t* q_left, const int32_t* q_right,
                            const int8_t* q_strand, const int64_t* score, int32_t min_unique, int32_t* col,
                            uint8_t* low);
/* Second step of the ingest, per file (the junction set of SPLICEDICE.getAllJunctions, SPLICEDICE.py:147-228):
 * chrom_rank[i] = rank_of_chrom[chrom_id[i]], and the admitted lines' keys, packed order-preservingly as
 * chrom 12 | left 31 | right - left 20 | strand 1 bits, compacted into keys_out (room for n; input of
 * sdice_sort_unique_u64).  *packable = 0 when an admitted junction does not fit the packing. */
int sdice_junc_pack_keys(int64_t n, const int32_t* chrom_id, const int32_t* rank_of_chrom, int32_t n_chroms,
                         const int32_t* left, const int32_t* right, const int8_t* strand, const uint8_t* admit,
                         int32_t* chrom_rank, uint64_t* keys_out, int64_t* n_keys, int32_t* packable);
int sdice_transpose_i32(int64_t rows, int64_t cols, const int32_t* src, int32_t* dst, int threads);
int sdice_host_threads(void);

/* ---- K0: sorted set of 64-bit keys, in place: the junction union of `quant`
 * (`self.junctions.add(...)` over every sample file + `sorted(self.junctions)`, SPLICEDICE.py:147-228,
 * :96) on order-preserving packed keys chrom|left|right-left|strand.  keys_inout holds n keys on
 * entry and the *n_unique distinct keys in ascending order on return. */
int sdice_sort_unique_u64(sdice_ctx* ctx, int64_t n, uint64_t* keys_inout, int64_t* n_unique);

/* ---- K8: `similarity` scoring (similarity.py:25-47).  For every row with sign != 0 (event
 * significant in the comparison table: p <= 0.05 and delta != 0, similarity.py:13-20) and every
 * column with a non-NaN PS: counts[c] += 1; scores[c] += (ps < mid) if sign < 0 else (ps > mid).
 * PS and midpoints are float64: the reference compares Python floats parsed from both tables. */
int sdice_similarity(sdice_ctx* ctx, int64_t n, int32_t s, const double* ps, const double* mid, const int8_t* sign,
                     int64_t* scores, int64_t* counts);
int sdice_similarity_dev(sdice_ctx* ctx, int64_t n, int32_t s, const double* d_ps, const double* d_mid,
                         const int8_t* d_sign, int64_t* d_scores, int64_t* d_counts);

/* ---- K9: per-row np.nanmean / np.nanstd over k selected columns (findOutliers.py:125-135), in the
 * matrix's own dtype (0 float32, 1 float64), bit-identical to numpy: NaNs replaced by zero in place,
 * numpy's pairwise summation, both divisions by the valid count in float64 then rounded.
 * mean / std are arrays of that dtype, n_nan the number of NaNs among the selected columns. */
int sdice_rowstats(sdice_ctx* ctx, int64_t n, int32_t s, const void* data, int dtype, const int32_t* idx, int32_t k,
                   void* mean, void* std_, int32_t* n_nan);
int sdice_rowstats_dev(sdice_ctx* ctx, int64_t n, int32_t s, const void* d_data, int dtype, const int32_t* d_idx,
                       int32_t k, void* d_mean, void* d_std, int32_t* d_nan);

/* ---- junction-axis shard plan (host only, no context): `world` contiguous row ranges cut at clean
 *      positions (no overlap edge crosses the cut) nearest to k*n/world; where none lies within
 *      max_shift_frac * n / world of it the ideal position is kept and the shard gets a read-only halo.
 *      plan[world][4] = own_lo, own_hi (rows whose results the rank produces), ext_lo, ext_hi (rows it holds). */
int sdice_shard_plan(int64_t n, const int64_t* row_ptr, const int32_t* col, int32_t world, double max_shift_frac,
                     int64_t* plan);
/* the same plan from the junction coordinates alone -- rows in output order, i.e. sorted by (chrom rank, left, right,
 * strand) and distinct -- so that no rank has to cluster the whole set before it knows its range: i < j of one
 * (chrom, strand) are joined iff left[j] <= right[i] (SPLICEDICE.py:237-250).  A rank then clusters rows
 * [ext_lo, ext_hi) alone (sdice_cluster*): the lists of its own rows are complete, indices relative to ext_lo. */
int sdice_shard_plan_junctions(int64_t n, const int32_t* chrom_rank, const int32_t* left, const int32_t* right,
                               const int8_t* strand, int32_t world, double max_shift_frac, int64_t* plan);

/* ---- multi-GPU (new; the reference is single-process): one context per rank,
 *      RCCL communicator owned by the context.  id is SDICE_COMM_ID_BYTES opaque
 *      bytes created on rank 0 and distributed by the caller (any channel). */
#define SDICE_COMM_ID_BYTES 128
int sdice_comm_unique_id(sdice_ctx* ctx, void* id_out);
int sdice_comm_init(sdice_ctx* ctx, const void* id, int rank, int world);
int sdice_comm_destroy(sdice_ctx* ctx);
/* collectives beside compute: after sdice_comm_fork the context's collectives run on a second stream, behind everything
 * enqueued on the main stream so far (call it again before a collective whose input has just been produced);
 * sdice_comm_join makes the main stream wait for them and takes collectives back to the main stream (sdice_sync joins) */
int sdice_comm_fork(sdice_ctx* ctx);
int sdice_comm_join(sdice_ctx* ctx);
/* all-gather equal-sized row shards: recv holds world * bytes_per_rank */
int sdice_allgather_dev(sdice_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_rank);
/* All-to-all of equal blocks: block q of d_send (bytes_per_peer each) goes to rank q, block r of
 * d_recv comes from rank r.  Used for the rows->columns transpose that `pairwise`'s default
 * per-pair-column BH (pairwise_fisher.py:187-191) needs when junction rows are sharded. */
int sdice_alltoall_dev(sdice_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes_per_peer);
/* Strided device copy (pack / unpack a column block of a row-major matrix), async on the ctx stream. */
int sdice_copy2d_dev(sdice_ctx* ctx, void* d_dst, int64_t dpitch, const void* d_src, int64_t spitch,
                     int64_t width_bytes, int64_t rows);

/* ---- per-kernel timing with HIP events on the context's stream ---------------------- */
/* on: 0 = off, 1 = every kernel, 2 = only the dominant kernel of each path (ps_tile_kernel,
 * ranksum_*_kernel, fisher_pairs_kernel, rccl_allgather) so that a timed loop is not perturbed */
int sdice_prof_enable(sdice_ctx* ctx, int on);
int sdice_prof_reset(sdice_ctx* ctx);
/* total device time and launch count of kernel `name` since the last reset (syncs) */
int sdice_prof_query(sdice_ctx* ctx, const char* name, int64_t* launches, double* total_ms);
/* newline-separated "name launches total_ms" table into buf */
int sdice_prof_report(sdice_ctx* ctx, char* buf, int cap);
/* stream-level stopwatch (hipEvents on the context stream) */
int sdice_timer_start(sdice_ctx* ctx);
int sdice_timer_stop(sdice_ctx* ctx, double* elapsed_ms);

/* Tuning and test knobs (integer parameters by name); unknown name -> ERR_ARG.  Defaults are the measured optima.
 *   ps.lds_bytes (81920)  ps.threads (1024)  ps.tile_rows (0 = from the LDS budget)  ps.halo_rows (-1 = 16 or the clustering's
 *     reach)  ps.chunk_cols (0 = all columns up to 256, else 128)  ps.xcd_remap (1)  ps.quantize3 (0; 1 = store the '.3f'
 *     round trip of PS)  ps.prio (1 = window loads at raised wave priority)  ps.nt_loads (1 = non-temporal window
 *     loads when the table is not cut into column chunks)  ps.ablate (timing experiments of the ablation instantiations only)
 *   cluster.generic / cluster.legacy (0; 1 = the radix-sort path)  cluster.sample_sort (1)  cluster.bucket_mean (2048)
 *     cluster.spb (0 = 12 samples per bucket up to 2 M junctions, 8 beyond)  cluster.lds_cap (8192; small values force the in-HBM sort: tests)  cluster.ablate
 *     (only in a library built with -DSDICE_CLUSTER_ABLATE=1)
 *   ranksum.variant (0 auto, 1 lane, 2 block, 3 wave, 4 float lane pair, 5 counting)  ranksum.ablate (timing experiments)
 *   fisher.table_max (1 << 20 log-factorials; small values force the lgamma path: tests)  fisher.refill (16 idle lanes before
 *     the next pairs are fetched)  fisher.unroll (8 walk steps per trip: 1, 2, 4, 6 or 8)
 *   bh.columns_path (0 by size, 1 radix, 2 sample sort)  bh.mean (160 values per bucket)  bh.spb (8 samples per bucket)
 *     bh.reg_cap (1024; small values force the in-HBM bucket sort: tests)
 *   sort.rounds (radix-sort scheduling experiment) */
int sdice_set_param(sdice_ctx* ctx, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* SDICE_H */
