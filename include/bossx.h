/* bossx.h — C-ABI of the MI355X-native BOSS-RUNS decision-update engine.
 *
 * The reference (goldman-gp-ebi/BOSS-RUNS v0.4.0) is pure Python and has no FFI for this
 * path: the path sits behind Python method calls on `BossRuns` (boss/runs/core.py:20-224)
 * and the `boss.npz` mask file.  This header is therefore the boundary a maintainer would
 * bind with `ctypes` from those methods (binding stub: INTEGRATION.md).  Every entry point
 * names the reference call site(s) it replaces.
 *
 * Conventions
 *   - plain C types, caller-allocated output buffers, inputs borrowed for the call only;
 *   - every function returns 0 on success or a negative BOSSX_E_* code; the message of the
 *     last failure on a handle is `bossx_last_error(h)`;
 *   - one caller thread per handle; calls are synchronous unless stated otherwise;
 *   - there is NO CPU fallback: without a HIP device `bossx_create` fails.
 */
#ifndef BOSSX_H
#define BOSSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BOSSX_OK 0
#define BOSSX_E_INVALID   (-1)  /* bad argument / call order                                  */
#define BOSSX_E_HIP       (-2)  /* HIP runtime failure (message carries hipGetErrorString)     */
#define BOSSX_E_PARSE     (-3)  /* malformed tag / AS value, CIGAR that does not consume qend - qstart read bases (ValueError) */
#define BOSSX_E_KEY       (-4)  /* read id in PAF not present in the batch, unknown tag type letter (KeyError) */
#define BOSSX_E_RANGE     (-5)  /* PAF line with fewer than 12 columns, mapping outside its contig, non-ACGT read base (IndexError) */
#define BOSSX_E_WINDOW    (-6)  /* move_sum window outside [1, n] (Bottleneck ValueError)      */
#define BOSSX_E_EMPTY     (-7)  /* no non-zero benefit (np.max of empty array: ValueError)     */
#define BOSSX_E_TYPE      (-8)  /* a PAF column the path computes with is not an integer (TypeError: paf.py:103-108 keeps it a str) */
#define BOSSX_E_OVERFLOW  (-10) /* CIGAR run of 2^32 bases or more, AS:f:inf, mapq / AS beyond int64 where several mappings are ranked (OverflowError: numpy / int()) */
#define BOSSX_E_ASSERT    (-9)  /* mapping without cg tag / CIGAR does not span tend - tstart (AssertionError, sequences.py:718,732) */

#define BOSSX_WINDOW        100     /* strategy / downsampling window, reference.py:109,215   */
#define BOSSX_BUCKET        20000   /* activation bucket, reference.py:83                      */
#define BOSSX_MAXCOV        30      /* "maxed" depth, sequences.py:419                         */
#define BOSSX_NCOMP         278256  /* compositions of 5 counts with sum < 30 = C(34,5)        */
#define BOSSX_HIST_BINS     1088    /* |frexp exponent| of x/max: 0..1075, rounded up          */
#define BOSSX_NWIN          11      /* window 4 (S_mu) + 10 read-length pieces                 */

typedef struct bossx_engine bossx_engine;

typedef struct bossx_config {
    int32_t device;          /* HIP device ordinal                                             */
    int32_t nbarcodes;       /* >= 1 (reference.py:34: nbarcodes = len(barcodes) or 1)         */
    int32_t track_entropy;   /* keep the per-site entropy array (sequences.py:450)             */
    int32_t reserved;
    void   *stream;          /* hipStream_t to launch on; NULL = engine-owned stream           */
} bossx_config;

/* ---- lifetime --------------------------------------------------------------------------- */
int  bossx_create(const bossx_config *cfg, bossx_engine **out);
void bossx_destroy(bossx_engine *h);
const char *bossx_last_error(const bossx_engine *h);
const char *bossx_version(void);

/* ---- reference set-up: Reference._load_contigs / Contig.__init__ (reference.py:18-118,
 *      305-338).  Contigs are added in FASTA order, rejected ones included (they only
 *      occupy 4 sites of Reference.n_sites and a bool[1] mask).  `seq` is ASCII, any case;
 *      non-ACGT letters become code 0 (reference.py:46-68).  Contigs shorter than 100 kb
 *      must be filtered by the caller (reference.py:330-331).                               */
#define BOSSX_CONTIG_REJECTED 1   /* reject_refs entry: "ACGT" dummy, bool[1] mask                */
#define BOSSX_CONTIG_REMOTE   2   /* multi-GPU: sites live on another device; `seq` may be NULL,
                                     `length` is required.  The contig keeps its place in the
                                     merged bin / row geometry (core.py:125-155).               */
int bossx_add_contig(bossx_engine *h, const char *name, const char *seq, int64_t length,
                     int32_t flags);
/* Allocate and initialise device state (coverage 0, scores = haploid score0, strat = 1).
 * `score0`/`ent0`: initial fill (Contig is always built with ploidy=1, reference.py:314).   */
int bossx_finalize(bossx_engine *h, double score0, double ent0);

/* Score/entropy tables for every coverage composition with depth < 30, replacing
 * Scoring.init_score_array + the on-demand fill (sequences.py:347-393, 433-448).
 * Entry [rank*4 + ref]; rank = sum_k C(c0+..+c(k-1) + k-1, k), k = 1..5.                     */
int bossx_set_lut(bossx_engine *h, const double *score, const double *entropy, int64_t n);

/* ---- ingestion: Paf.parse_PAF (paf.py:631-672, min_len filter + primary filter),
 *      Paf.choose_best_mapper (paf.py:709-722), CoverageConverter.convert_records +
 *      _parse_cigar (sequences.py:678-794) and Contig.increment_coverage
 *      (reference.py:122-145) for every non-rejected contig.
 *
 * The batch: n_reads reads; `names`/`seqs` are concatenated blobs with n_reads+1 offsets;
 * `barcodes` is NULL (all 0) or one barcode index per read.  `paf` is the PAF text the
 * mapper produced.  Per chosen mapping (one per mapped read, in first-appearance order) the
 * summary arrays receive: read index, contig index in add order (-1: target unknown), strand
 * (0 '+', 1 '-'), tstart, tend, qlen — what ReadStartDist.count_read_starts
 * (readstartdist.py:43-82) and AbundanceTracker (abundance_tracker.py:24-41) consume.
 * Capacity of each summary array must be >= n_reads; *n_rec returns the count.
 *
 * bossx_stage_batch parses and uploads (H2D) only; bossx_ingest_staged launches the
 * coverage-scatter kernel on the staged batch (asynchronous on the engine stream);
 * bossx_ingest_paf = stage + ingest.                                                        */
typedef struct bossx_batch_summary {
    int32_t *read_idx;
    int32_t *contig_idx;
    uint8_t *rev;
    int64_t *tstart;
    int64_t *tend;
    int64_t *qlen;
} bossx_batch_summary;

int bossx_stage_batch(bossx_engine *h, const char *paf, size_t paf_len,
                      const char *names, const int64_t *name_off,
                      const char *seqs, const int64_t *seq_off,
                      const int32_t *barcodes, int32_t n_reads, int32_t min_len,
                      bossx_batch_summary *summary, int32_t *n_rec,
                      int64_t *aligned_bases);
/* Same as bossx_stage_batch for callers that hold the reads as separate strings (a Python
 * dict of str): one pointer + length per read name and per read sequence; the library gathers
 * the sequences into its own pinned staging buffer (no concatenation on the caller's side).    */
int bossx_stage_batch_ptrs(bossx_engine *h, const char *paf, size_t paf_len,
                           const char *const *name_ptrs, const int64_t *name_lens,
                           const char *const *seq_ptrs, const int64_t *seq_lens,
                           const int32_t *barcodes, int32_t n_reads, int32_t min_len,
                           bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases);
int bossx_ingest_staged(bossx_engine *h);
/* How the reads cross PCIe and lie in HBM when some read holds a byte other than A/C/G/T (otherwise: bossx_pack_reads2 below):
 * four bits per base, two bases per byte (low nibble first), every read from a byte boundary on.  Codes: 0-3 = A C G T; 4-8 = '0'..'4' and 9 = '7' (the bytes np.fromstring of sequences.py:766 turns into a
 * base index, a deletion or the padding value: a digit in a read counts like the letter it would be translated to); 15 = any
 * other byte (never counted).  bossx_pack_reads writes the (n + 1) / 2 bytes of ONE read's n bases to `dst` exactly as the
 * staging does (an odd read's last high nibble is 15) and sets *dirty to 1 if some byte is not A/C/G/T.  Host only, no engine:
 * the CPU test-suite holds the vector and the scalar packer to one another with it.                                         */
int bossx_pack_reads(const char *bases, int64_t n, uint8_t *dst, int32_t *dirty);
/* Round 6: a batch in which every read with a mapping holds nothing but A/C/G/T — every batch a basecaller writes — crosses
 * PCIe with TWO bits per base (base i of a read in byte i / 4, bits 2 (i mod 4) up, codes 0-3 as above; every read from a
 * byte boundary on); a batch with any other byte in such a read is packed a second time, as nibbles, before anything reads
 * it (BOSSX_BLOB_NIBBLES=1: always nibbles).  bossx_pack_reads2 writes the (n + 3) / 4 bytes of ONE read as the staging does
 * (unused high bits of the last byte are 0) and sets *dirty to 1 if some byte is not A/C/G/T — the bytes written are then
 * void.  Host only, no engine: vector packer against scalar packer in the CPU tier.                                           */
int bossx_pack_reads2(const char *bases, int64_t n, uint8_t *dst, int32_t *dirty);
/* Mapping choice only: Paf.parse_PAF filters + choose_best_mapper per read, summary arrays as
 * above, nothing staged or ingested.  This is what the simulation's decision step needs from
 * the truncated-read PAF (runs/simulation.py:63-75).  Reads are identified by name only.     */
int bossx_paf_summary(bossx_engine *h, const char *paf, size_t paf_len,
                      const char *const *name_ptrs, const int64_t *name_lens, int32_t n_reads,
                      int32_t min_len, bossx_batch_summary *summary, int32_t *n_rec);
/* The simulation's decision step keeps, per read, either its full-length or its truncated mapping
 * (runs/simulation.py:87-120): copies to `out` the lines of `paf` whose query name (column 1,
 * normalised like PafLine: "007" is "7") is read i of the batch with keep[i] != 0, in file order,
 * newline separated.  `out_cap` >= paf_len + 1 always suffices.  Host only.                    */
int bossx_paf_select_lines(const char *paf, size_t paf_len, const char *const *name_ptrs,
                           const int64_t *name_lens, int32_t n_reads, const uint8_t *keep,
                           char *out, size_t out_cap, size_t *out_len);
/* ReadlengthDist.update + ccl_approx_constant (readlengthdist.py:36-97) on the host, natively: adds
 * the `n_lens` new read lengths (> min_len_exclusive, clipped to hist_len - 1) to the uint16
 * histogram `hist` (the caller's array of hist_len counters; wraps like the reference's), then
 * returns lam (mean length), the longest observed length and the eta - 1 indices where the
 * complementary cumulative length distribution falls to 0.95, 0.85, ... (approx_ccl), all
 * bit-identical to the reference's numpy arithmetic.  `hi_inout` carries the largest index ever
 * incremented between calls.  `observed` = 0 if the histogram is still empty (nothing else is
 * written then).  No device, no engine: this is the host step the move_sum windows wait for.   */
int bossx_rl_update(uint16_t *hist, int64_t hist_len, const int64_t *lens, int64_t n_lens,
                    int64_t min_len_exclusive, int32_t eta, int64_t *hi_inout,
                    double *lam, int64_t *longest_read, int32_t *approx_ccl, int32_t *observed);

/* Host-only check of the PAF front end: no engine, no device.  Parses exactly like
 * bossx_stage_batch_ptrs (same filters, mapping choice, CIGAR walk, error codes; `n_threads`
 * 0 = default) for contigs given as names / lengths / BOSSX_CONTIG_* flags, verifies the tile
 * segments it would upload, and expands the emit runs base by base the way the ingest kernels
 * read them: for emitted base e, out_contig/out_pos = where it lands, out_code = 0..3 (A C G T
 * on the reference strand), 4 (deletion), 255 (not A/C/G/T), out_barcode = barcode index.  The
 * out_* arrays may be NULL (counts and checks only).  Used by the CPU test-suite to hold the
 * native parser to CoverageConverter.convert_records (sequences.py:678-794).                  */
int bossx_host_parse(const char *const *contig_names, const int64_t *contig_lengths,
                     const int32_t *contig_flags, int32_t n_contigs, int32_t nbarcodes,
                     const char *paf, size_t paf_len, const char *const *name_ptrs,
                     const int64_t *name_lens, const char *const *seq_ptrs, const int64_t *seq_lens,
                     const int32_t *barcodes, int32_t n_reads, int32_t min_len, int32_t n_threads,
                     bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases,
                     int32_t *out_contig, int64_t *out_pos, uint8_t *out_code, uint8_t *out_barcode,
                     int64_t out_cap, char *err_buf, size_t err_cap);

/* Staged batches live in numbered slots (default 0) so several batches can be resident in HBM
 * at once; selects the slot the next stage/ingest call uses.                                */
int bossx_select_batch(bossx_engine *h, int32_t slot);
/* Round 6: which queue the staging's kernels (walk, expansion) go to from the next bossx_stage_batch* on.  0 (default): the
 * engine's staging stream — a batch staged AHEAD runs next to the update in flight on the main stream.  1: the main stream —
 * for a batch that bossx_ingest_staged / bossx_update_begin consume right away: the sweep then follows the expansion in one
 * queue instead of behind a wait across queues (~25 us per lone update).  Results do not depend on it.                       */
int bossx_stage_stream(bossx_engine *h, int32_t on_main);
int bossx_ingest_paf(bossx_engine *h, const char *paf, size_t paf_len,
                     const char *names, const int64_t *name_off,
                     const char *seqs, const int64_t *seq_off,
                     const int32_t *barcodes, int32_t n_reads, int32_t min_len,
                     bossx_batch_summary *summary, int32_t *n_rec, int64_t *aligned_bases);

/* ---- per-site sweep: Scoring.update_scores (sequences.py:398-455), Contig.modify_scores /
 *      _find_dropout (reference.py:148-179), the 20-kb sums of Contig.check_buckets
 *      (reference.py:196-199) and the 100-site in-order bin sums of Contig.calc_smu
 *      (reference.py:225-231), fused into one pass.  Asynchronous on the engine stream.    */
int bossx_sweep(bossx_engine *h);

/* Bucket sums of the last sweep: uint64[nbarcodes][n_full_buckets(contig)] (only complete
 * 20-kb buckets; the reference duplicates the last one into the tail bucket,
 * utils.py:206-226).  Synchronises.                                                         */
int bossx_get_bucket_sums(bossx_engine *h, int32_t contig, uint64_t *dst);
/* Current bucket switches uint8[n_buckets(contig)][nbarcodes] (reference layout), decided by
 * the caller as Contig.check_buckets does (reference.py:200-208).                           */
int bossx_set_bucket_switches(bossx_engine *h, int32_t contig, const uint8_t *sw);

/* ---- benefit: Contig.calc_smu + Contig.calc_u (reference.py:215-269) for every
 *      non-rejected contig and barcode, with the exact Bottleneck running-sum recurrence.
 *      windows[0] = mu/100 = 4; windows[1..10] = approx_ccl // 100; mult[10] = the
 *      np.arange(.05,1,.1)[::-1] doubles.  Writes max(additional_benefit) over the merged,
 *      length-adjusted array (sequences.py:588) to *max_benefit (synchronises).             */
int bossx_benefit(bossx_engine *h, const int32_t *windows, const double *mult,
                  double *max_benefit);

/* ---- threshold statistics of Scoring.find_strat_thread (sequences.py:584-629) given the
 *      (global) normaliser.  fhat is passed compact: `fhat_c` = float64[n_windows][2]
 *      (already scaled by the normaliser of readstartdist.py:145-151) plus the index
 *      arithmetic of _expand_fhat / adjust_length.  Outputs (host):
 *      counts int64[BOSSX_HIST_BINS]; f_grid and ubar0 as exact 128-bit fixed point
 *      (value * 2^100), little-endian limbs uint64[bins][2] and uint64[2].                  */
typedef struct bossx_fhat_desc {
    const double *fhat_c;      /* [n_windows][2]                                              */
    int64_t n_windows;
    int64_t rep;               /* 2000 / 100 = 20                                             */
    int64_t target_rs;         /* ReadStartDist.target_size                                   */
    int64_t target;            /* Reference.n_sites // 100                                    */
} bossx_fhat_desc;

/* Read-start counts resident in HBM (ReadStartDist.read_starts merged over the non-rejected
 * contigs, readstartdist.py:24-41): bossx_fhat_reset installs `counts` float64[n_windows][2]
 * (NULL: zeros); bossx_fhat_add adds one start at each flat index of `keys`
 * (count_read_starts / update_read_starts, readstartdist.py:43-82).                            */
int bossx_fhat_reset(bossx_engine *h, const double *counts, int64_t n_windows);
int bossx_fhat_add(bossx_engine *h, const int64_t *keys, int32_t n_keys);
/* Rebuild the posterior of update_f_pointmass (readstartdist.py:86-152) in HBM from the resident
 * counts, asynchronously on the engine's stream (what BOSSX_UPDATE_FHAT_RESIDENT does inside
 * bossx_update); bossx_dist_hist then takes a descriptor whose fhat_c is NULL.                  */
int bossx_fhat_build(bossx_engine *h, int64_t n_windows, int64_t target_rs, double alpha, double den,
                     double expected, double on_target);

int bossx_histogram(bossx_engine *h, double normaliser, const bossx_fhat_desc *fh,
                    int64_t *counts, uint64_t *fgrid_fx, uint64_t *ubar0_fx);

/* ---- strategy: `benefit >= threshold` (sequences.py:648) written through the bucket gate
 *      with the reference's row arithmetic (core.py:125-155).  Asynchronous.                */
int bossx_apply_threshold(bossx_engine *h, double threshold);

/* ---- the whole of BossRuns.update_wrapper (core.py:160-198) in one call: sweep, bucket
 *      switches (decided on the device with `bucket_threshold`, reference.py:200-208), and —
 *      once any bucket of any contig is on — benefit, threshold search and masks, enqueued
 *      back to back with a single synchronisation at the end.  `fhat_c == NULL` runs only the
 *      sweep and the bucket switches.  `strat_all` (may be NULL) receives every non-rejected
 *      contig's mask back to back in add order (bossx_strat_bytes bytes); `contig_on`
 *      (may be NULL) one byte per contig in add order = any(Contig.switched_on).  The optional
 *      statistics arrays are those of bossx_histogram.                                       */
#define BOSSX_UPDATE_SWEEP_DONE   1  /* bossx_update_begin already enqueued sweep + bucket switches */
#define BOSSX_UPDATE_BENEFIT_DONE 2  /* bossx_update_benefit already enqueued the move_sum chain     */
#define BOSSX_UPDATE_STRAT_BITS   4  /* `strat_all` receives the masks packed 8:1 (bossx_get_strat_bits) */
#define BOSSX_UPDATE_STRAT_DELTA  16 /* round 6: the caller has not written into `strat_all` since the previous bossx_update
                                      * filled it (a bossx_host_alloc buffer in byte form): where the engine, too, knows the buffer
                                      * to hold the device's masks byte for byte, only the groups of rows whose masks change are
                                      * written — a few per cent of the 2.2 MB of chr20+21 between two updates.  Without the flag,
                                      * or after anything else wrote the masks (import, a different buffer), every mask is written. */
#define BOSSX_UPDATE_FHAT_RESIDENT 8 /* fhat_c is NULL: the posterior is rebuilt on the device from the resident
                                      * read-start counts (bossx_fhat_reset / bossx_fhat_add) and the fhat_* scalars */
typedef struct bossx_update_params {
    int32_t windows[BOSSX_NWIN];
    int32_t flags;
    double  mult[10];
    double  tc;                 /* ReadlengthDist.time_cost // 100 (sequences.py:581)           */
    double  bucket_threshold;   /* OptionalConfig.bucket_threshold (config.py:51)               */
    const double *fhat_c;       /* as bossx_fhat_desc                                           */
    int64_t n_windows;
    int64_t target_rs;
    /* BOSSX_UPDATE_FHAT_RESIDENT: ReadStartDist.update_f_pointmass (readstartdist.py:86-152) on the
     * device; the O(1) part of the model comes from the host                                     */
    double  fhat_alpha;         /* prior alpha                                                   */
    double  fhat_den;           /* 2 N alpha + sum(counts)                                       */
    double  fhat_expected;      /* point-mass expectation of a window without read starts (:104-117) */
    double  fhat_on_target;     /* ReadStartDist.on_target                                       */
} bossx_update_params;

typedef struct bossx_update_result {
    int32_t updated;            /* 1 if the strategy was recomputed (switched_on)               */
    int32_t any_on;
    int32_t strat_size;         /* argmax + 1 of sequences.py:636                               */
    int32_t n_bins;             /* occupied exponent bins                                       */
    double  threshold;
    double  normaliser;
    double  ubar0;
    /* (best - second best) / best of cs_u / cs_t at the argmax of sequences.py:636.  The engine's sums are exact, the reference's
     * are 12-chunk float sums (sequences.py:609-629): the two agree to ~1e-16 relative, so a margin far above that means both
     * choose the same bin, a margin near it that a flipped choice (threshold off by a factor of two) is possible.  1.0 when
     * there is a single occupied bin.                                                                                       */
    double  argmax_margin;
    int32_t thr_code;           /* exponent bin u of the threshold 2^-u * normaliser when the masks were formed from the exponent
                                 * codes (exact); -1 when they compared doubles (subnormal threshold, u > 254, host-picked)  */
    int32_t reserved;
} bossx_update_result;

/* Optional first half of bossx_update: enqueue the sweep (with the pending batch's increments)
 * and the bucket switches and return immediately, so the caller's host-side bookkeeping
 * (read-length distribution, read starts, f-hat) overlaps with it; then call bossx_update with
 * BOSSX_UPDATE_SWEEP_DONE in `flags`.                                                         */
int bossx_update_begin(bossx_engine *h, double bucket_threshold);
/* Optional second part: enqueue calc_smu/calc_u (gated on the device-side "armed" flag) as soon
 * as the read-length windows are known; asynchronous.                                          */
int bossx_update_benefit(bossx_engine *h, const int32_t *windows, const double *mult);
/* Multi-GPU helpers.  bossx_arm sets the device-side "some strategy is switched on" flag (the
 * decision is global: any contig on any rank, core.py:111), so gated stages run on ranks whose
 * own contigs are still below the bucket threshold.  bossx_get_max waits for the chain and
 * returns this device's max(additional_benefit) (0 if the gated chain did not run).           */
/* Device-resident multi-GPU update: the statistics stay in HBM and are all-reduced in-stream
 * (RCCL through torch.distributed on tensors that alias the buffers below; the engine must
 * have been created on that stream), so an update has no host round trip before its end:
 *     bossx_update_begin                      sweep + bucket switches
 *     all-reduce MAX  ARMED flag              (int32[1])
 *     bossx_update_benefit                    chain (gated on the flag)
 *     all-reduce MAX  NORMALISER              (int64[1]: bit pattern of a non-negative double)
 *     bossx_dist_hist                         histogram with the global normaliser -> LIMBS
 *     all-reduce SUM  LIMBS                   (int64[(BOSSX_HIST_BINS + 1) * 5], exact)
 *     bossx_dist_pick                         global threshold; publishes the block TAILS
 *     all-reduce SUM  TAILS                   (the first n_filt * n_filt * 2 * nb float64, one non-zero
 *                                              contributor per element: exact)
 *     bossx_dist_finish                       masks (halo rows from the tails), D2H, sync
 * Shorter form (one collective fewer): after the chain, bossx_dist_tails publishes the TAILS
 * (they do not depend on the threshold).  The TAILS buffer is followed in memory by the running
 * maximum (BOSSX_PTR_TAILS reports n_filt * n_filt * 2 * nb + 1 doubles): ONE all-reduce MAX over
 * it (every element is non-negative and has one non-zero contributor, so MAX is as exact as SUM)
 * replaces the NORMALISER and TAILS exchanges, reducing the normaliser in place; bossx_dist_pick
 * then only picks.                                                                            */
#define BOSSX_PTR_ARMED      0
#define BOSSX_PTR_NORMALISER 1
#define BOSSX_PTR_LIMBS      2
#define BOSSX_PTR_TAILS      3
int bossx_device_ptr(bossx_engine *h, int32_t which, void **ptr, size_t *bytes);
int bossx_dist_hist(bossx_engine *h, const bossx_fhat_desc *fh);
int bossx_dist_pick(bossx_engine *h, double tc);
int bossx_dist_tails(bossx_engine *h);
int bossx_dist_finish(bossx_engine *h, uint8_t *strat_all, uint8_t *contig_on, bossx_update_result *res);
/* ---- the same update with the collectives issued by the LIBRARY: RCCL (ncclAllReduce) on the engine's
 *      own stream, between its kernels — one call per update, no interpreter between the stages.
 *      bossx_dist_unique_id (rank 0) makes the id every rank passes to bossx_dist_init (share it through
 *      any channel: a file, MPI, torch.distributed); bossx_dist_init creates the communicator for this
 *      engine's device.  bossx_dist_chain (optional) exchanges the "some strategy is on" flag (until it is)
 *      and enqueues the move_sum chain as soon as the read-length windows are known; bossx_dist_update
 *      enqueues whatever is still missing of
 *          sweep + bucket switches | MAX armed | chain | tails | MAX tails + normaliser | f-hat |
 *          histogram | SUM limbs | threshold | masks | D2H
 *      and returns with the masks of the LOCAL contigs and the global threshold.  `up` as for bossx_update
 *      (BOSSX_UPDATE_SWEEP_DONE / _BENEFIT_DONE / _FHAT_RESIDENT honoured; fhat_c or the resident counts
 *      must describe the GLOBAL read starts).  librccl is loaded on first use (dlopen).                 */
#define BOSSX_NCCL_ID_BYTES 128
int bossx_dist_unique_id(uint8_t *id /*[BOSSX_NCCL_ID_BYTES]*/);
int bossx_dist_init(bossx_engine *h, const uint8_t *id, int32_t rank, int32_t world);
int bossx_dist_chain(bossx_engine *h, const int32_t *windows, const double *mult);
int bossx_dist_update(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                      bossx_update_result *res);
/* Collectives issued through this engine's communicator so far (measurement).                         */
/* bossx_dist_update in two calls, like bossx_update_launch / _collect: everything — collectives included — is enqueued
 * by _launch; the caller stages the next batch; _collect (same arguments) waits and fills the results.              */
int bossx_dist_update_launch(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                             bossx_update_result *res);
int bossx_dist_update_collect(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all, uint8_t *contig_on,
                              bossx_update_result *res);
int64_t bossx_dist_collectives(const bossx_engine *h);
/* `bytes` bytes from every rank, in rank order, into recv_all[world * bytes] (host memory on both sides): the exchange of
 * the per-batch summaries when reads are sharded over ranks (the reference has one process and no such step: SURVEY §8e;
 * boss-runs_amd/parallel.py account_batch).  One RCCL all-gather on the engine's stream; counted by bossx_dist_collectives. */
int bossx_dist_allgather(bossx_engine *h, const void *send, void *recv_all, size_t bytes);

/* The benefit chain (calc_smu + calc_u, reference.py:215-269) runs chunk-parallel by default: candidate start
 * values per chunk on the matrix core, stitched, every segment recomputed from its exact start and checked against
 * its successor's (csrc/kernels.hip.inc).  Counters since bossx_finalize:
 *   out[0] launches of the chunk-parallel form        out[1] launches kept on the serial kernel (too many chunks
 *   out[2] chunks the stitch added the plain way               needed plain adds in the launch before: 8 launches pause)
 *   out[3] launches whose check failed (the serial kernel, enqueued behind and gated on the flag, then ran; 3: off) */
int bossx_chain_stats(const bossx_engine *h, int64_t out[4]);
/* The chunk-parallel chain's device-side counters (collected only while the environment holds BOSSX_SPEC_STATS: the kernels then
 * count with atomics), since finalize: out[0] table rows built, [1] rows left standing (inputs unchanged), [2] rows without a table,
 * [3] strided rows built (a chunk whose sum climbs more than four binades: 64 of its residues, csrc/kernels.hip.inc kStridedK),
 * [4] strided rows the stitch looked up chunk by chunk, [5] ... of which the two enclosing candidates ended on different values
 * (the chunk is then evaluated from the exact value), [6] groups stepped through their composed super-row, [7] groups walked chunk
 * by chunk.  Diagnostics and tests; results never depend on them.                                                              */
int bossx_chain_counters(bossx_engine *h, int64_t out[8]);
/* Page-locked host memory for the caller's output buffers — the mask buffer of bossx_update above
 * all: a device-to-host copy into it is a direct DMA, into pageable memory it is staged (2.2 MB of
 * masks at 110 Mb: 0.1 ms less per update; registering pageable memory after the fact measured
 * slower than either).  bossx_host_free needs no engine: the buffer may outlive it.             */
int bossx_host_alloc(bossx_engine *h, size_t bytes, void **ptr);
int bossx_host_free(void *ptr);
/* The concurrent chain (bossx_update_benefit next to the sweep of bossx_update_begin) falls back to
 * the serial schedule inside bossx_update if it times out; callers that consume the chain through
 * the stage-wise / multi-GPU entry points instead switch it off.                               */
int bossx_set_overlap(bossx_engine *h, int32_t on);
int bossx_arm(bossx_engine *h);
int bossx_get_max(bossx_engine *h, double *max_benefit);
int bossx_update(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all,
                 uint8_t *contig_on, bossx_update_result *res, int64_t *counts,
                 uint64_t *fgrid_fx, uint64_t *ubar0_fx);
/* The same update in two calls: _launch enqueues everything (masks and results included) and returns; the caller
 * does host work that needs neither — e.g. stages the NEXT batch into the other slot — and _collect, with the SAME
 * arguments, waits and fills `res`, `contig_on`, `counts` ...  Buffers and `up` must stay valid in between; one
 * update at a time; no other engine call but bossx_select_batch / bossx_stage_batch* between the two.          */
int bossx_update_launch(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all,
                        uint8_t *contig_on, bossx_update_result *res, int64_t *counts,
                        uint64_t *fgrid_fx, uint64_t *ubar0_fx);
int bossx_update_collect(bossx_engine *h, const bossx_update_params *up, uint8_t *strat_all,
                         uint8_t *contig_on, bossx_update_result *res, int64_t *counts,
                         uint64_t *fgrid_fx, uint64_t *ubar0_fx);
int64_t bossx_strat_bytes(const bossx_engine *h);
int64_t bossx_strat_offset(const bossx_engine *h, int32_t contig);

/* Contig.strat as bool bytes, reference layout [length//100][2][nbarcodes] (a rejected
 * contig has the single byte 0, reference.py:116).  Synchronises.                           */
int bossx_get_strat(bossx_engine *h, int32_t contig, uint8_t *dst);

/* All masks (the bossx_strat_bytes buffer, non-rejected contigs back to back in add order)
 * packed 8:1 on the device in np.packbits order (element 8i+j = bit 7-j of byte i): element
 * bossx_strat_offset(c) + (row*2 + strand)*nbarcodes + barcode of the unpacked buffer is
 * Contig.strat[row, strand, barcode].  This is the payload of the bit-packed mask file the
 * readfish-side consumer maps (dynamic_readfish.py:169-210 replacement).
 * `dst` holds bossx_strat_bits_bytes bytes.  Synchronises.                                  */
int64_t bossx_strat_bits_bytes(const bossx_engine *h);
int bossx_get_strat_bits(bossx_engine *h, uint8_t *dst);

/* ---- geometry / introspection ----------------------------------------------------------- */
int32_t bossx_n_contigs(const bossx_engine *h);
int64_t bossx_contig_length(const bossx_engine *h, int32_t contig);
int64_t bossx_n_sites(const bossx_engine *h);       /* Reference.n_sites (reference.py:343)   */
int64_t bossx_merged_bins(const bossx_engine *h);   /* sum over non-rejected of length//100+1 */
/* 1 if the move_sum recurrence runs on the FP64 matrix core (it passed the bit-exactness
 * self-test at finalize), 0 if on the vector ALU.  Results are identical either way.         */
int32_t bossx_matrix_chain(const bossx_engine *h);

/* Parity / checkpoint hooks.  `which`:
 *   0 coverage   uint16[nb][5][L]        (reference holds [L][5][nb])
 *   1 scores     float64[nb][L]          (materialised from site state)
 *   2 entropy    float64[nb][L]
 *   3 scores_ds  float64[nb][L//100+1]
 *   4 benefit    float64[nb][2][L//100+1] (additional_benefit, reference.py:266-269)
 *   5 site state uint8[nb][L] (bit0-1 ref base, bit2 scored, bit3 zeroed by dropout)
 *   6 touched    uint8[L]     (change_mask rows of the pending batch)
 *   7 bucket switches uint8[nb][L//20000+1] (export only)
 *   8 benefit tail float64[nb][2][min(L//100+1, n_filt)] (export only; multi-GPU halo rows)
 *   9 strat      uint8 [L//100][2][nb] (Contig.strat)
 * bossx_import accepts 0, 2, 3 (input of bossx_benefit until the next sweep), 5, 6, 7 (layout [nb][n_buckets]) and 9.                                                          */
int bossx_export(bossx_engine *h, int32_t contig, int32_t which, void *dst, size_t dst_bytes);
int bossx_import(bossx_engine *h, int32_t contig, int32_t which, const void *src, size_t src_bytes);

/* Synthetic preload for benchmarks: Poisson(depth) reads' worth of coverage on the reference
 * base with substitution/deletion noise, generated on the device (no reference analogue).   */
int bossx_preload_coverage(bossx_engine *h, double depth, uint64_t seed);

/* ---- measurement ------------------------------------------------------------------------ */
#define BOSSX_K_INGEST   0
#define BOSSX_K_SWEEP    1
#define BOSSX_K_BENEFIT  2
#define BOSSX_K_HIST     3
#define BOSSX_K_MASK     4
#define BOSSX_K_COUNT    5
/* HIP-event time of the last launch of each kernel (ms) and launch counts; enabling timing
 * brackets each kernel with hipEvents on the engine stream.  `on` = 1: every kernel; 2 + k: kernel k
 * (BOSSX_K_*) alone — every event pair is two marker packets between the kernels of an update,
 * ~10 us of an idle GPU each: a measurement that wants ONE kernel's time inside a timed region
 * asks for that kernel only.  0: off.                                                       */
int bossx_enable_timing(bossx_engine *h, int32_t on);
int bossx_kernel_ms(bossx_engine *h, float *ms_last /*[BOSSX_K_COUNT]*/,
                    double *ms_total /*[BOSSX_K_COUNT]*/, int64_t *launches /*[BOSSX_K_COUNT]*/);
/* Algorithmic bytes of the last launch of each kernel (definition: DESIGN.md).              */
int bossx_kernel_bytes(bossx_engine *h, double *bytes_last /*[BOSSX_K_COUNT]*/);
int bossx_synchronize(bossx_engine *h);

#ifdef __cplusplus
}
#endif
#endif /* BOSSX_H */
