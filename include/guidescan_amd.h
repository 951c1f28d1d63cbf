/*
 * guidescan_amd.h -- C-ABI of the MI355X-native off-target enumerator.
 *
 * This is the drop-in boundary for guidescan's enumerate hot path.  The
 * reference has no FFI: the seam is a C++ template call from the per-guide
 * pipeline into the FM-index.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference
 * tree).  Plain pointers and sizes only; no C++ or torch types; every function
 * returns a gs_status and never throws.
 *
 * Batch-oriented by design: the reference calls inexact_search once per guide
 * per strand from N std::threads (src/guidescan.cxx:240-247); a GPU needs the
 * whole batch, so gs_enumerate takes n guides and returns CSR hit lists in the
 * reference's canonical per-guide order (include/genomics/process.hpp:100-115).
 */
#ifndef GUIDESCAN_AMD_H
#define GUIDESCAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GS_OK = 0,
  GS_ERR_ARG = 1,         /* bad argument (NULL, sizes out of range) */
  GS_ERR_DEVICE = 2,      /* HIP runtime error / no usable device */
  GS_ERR_UNSUPPORTED = 3, /* input outside what the device path implements */
  GS_ERR_NOMEM = 4,
  GS_ERR_IO = 5,
  GS_ERR_FORMAT = 6 /* malformed index file */
} gs_status;

/* Opaque handle: both strand indexes of one genome resident in one GPU's HBM.
 * Replaces the two `genome_index<wt_huff<>,64,8192>` objects built in
 * src/guidescan.cxx:198-211 (include/genomics/index.hpp:23-123).
 * Threads: the reference calls its seam from N std::threads on one const index
 * (src/guidescan.cxx:240-247).  A handle owns its workspace, its lazily built tables and the
 * device buffers its results are left in, so it runs ONE call at a time: every entry point that
 * takes a handle holds the handle's lock for the whole call and other threads' calls wait.  The
 * host-pointer entry points (gs_enumerate, gs_enumerate_general*, gs_score, ...) copy their
 * results out under the lock: any number of threads may call them on one handle and each gets
 * the bytes a single thread would.  The device-pointer entry points leave their results in the
 * handle's buffers "until the next call": a thread that wants to read or score them holds the
 * handle with gs_index_lock ... gs_index_unlock around its calls (the lock is recursive). */
typedef struct gs_index gs_index;

/* One off-target hit, 16 bytes.
 *   pos : signed absolute coordinate exactly as process.hpp:104,111 computes it
 *         (forward-index hit: -SA[row]; reverse-index hit: L-(SA[row]+1)).
 *   key : canonical sort key that also encodes the whole `match` struct
 *         (include/genomics/structures.hpp:33-43):
 *           bits 63..61  match.mismatches
 *           bit  60      0 = found in the forward index, 1 = reverse index
 *           bits 59..1   match.sequence as per-position codes, position 0 first (2L + 3P <= 59 bits,
 *                        from bit 59 down; what a sequence does not use stays zero):
 *                        guide positions (2 bits): 0 = equals the query base
 *                          (upper case), 1..3 = mismatch, the matched base's rank
 *                          among the three other bases in A<C<G<T order (lower case);
 *                        PAM positions (3 bits): A=0 C=1 G=2 N=3 T=4 (upper case).
 *           bit  0       zero
 *         (Sequences of up to 52 bits - 20-mers with a PAM of up to four symbols - leave bits 7..1 zero: the
 *         layout of earlier versions.  23-mers with a four-symbol PAM use 58.)
 *         Ascending key == the reference's order of (distance, index, std::set<match>
 *         ordered by sequence string) because 'A'<'C'<'G'<'N'<'T'<'a'<'c'<'g'<'t'.
 * gs_decode_sequence() rebuilds match.sequence from (guide, key). */
typedef struct {
  int64_t pos;
  uint64_t key;
} gs_hit;

#define GS_KEY_MISMATCHES(key) ((uint32_t)((key) >> 61))
#define GS_KEY_INDEX(key) ((uint32_t)(((key) >> 60) & 1))

/* flags for gs_enumerate */
#define GS_FLAG_PAM_AT_START 1u /* --start, process.hpp:63,84-87 */
/* Walk the whole search tree from the root exactly as the reference does (no prefix-table
 * shortcut).  Same hits; n_ext then equals the reference traversal's node count.  Without it
 * n_ext counts only the extensions actually executed below the table depth. */
#define GS_FLAG_FAITHFUL_WALK 2u
/* Measurement only: run the counting instantiation of the search kernel, which tallies the distinct
 * 64-byte lines each of its load instructions asks for (gs_index_last_counters; with the handle's switch
 * GS_COUNT_SHIFT=7 - gs_index_set_option - 128-byte blocks, what the memory system serves as one random request).
 * Same results, slower; never set in a timed call. */
#define GS_FLAG_COUNT_REQUESTS 4u
/* Also return, per guide, the number of hits BEFORE duplicate sequences are dropped
 * (gs_result_view.raw_hits): the quantity the reference's --threshold filter compares with 1
 * (off_target_counter, process.hpp:25-27 and 66-76 - one count per PAM pattern that matches, so a
 * site two patterns of the list match counts twice).  Saturates at 2^32-1. */
#define GS_FLAG_RAW_COUNTS 8u
/* Do not build new derived tables for this call (the PAM-pair and deep tables, DESIGN.md 4.9-4.10: about
 * 0.3 s and tens of GB at hg38 size, kept on the handle); tables the handle already holds are used.
 * For short jobs: the tables pay for themselves after ~5 guides per 1,000 genome bases (15 M guides
 * at hg38 size).  Same results either way. */
#define GS_FLAG_NO_NEW_TABLES 16u

typedef struct {
  uint64_t n_guides;
  uint64_t n_hits;
  const uint64_t *guide_offsets; /* n_guides+1 entries, host memory owned by the result */
  const gs_hit *hits;            /* n_hits entries, host memory owned by the result */
  /* work counters of SURVEY.md section 8d (properties of the input): */
  uint64_t n_ext;     /* search-tree nodes extended (a3 calls with position>=0 + a4 calls with begin!=end);
                         exact only under GS_FLAG_FAITHFUL_WALK, see there */
  uint64_t n_matches; /* distinct (index, match.sequence) intervals */
  /* device timing of the last call, milliseconds (HIP events on the call's stream) */
  float ms_search;  /* search kernel only */
  float ms_total;   /* prepare + search + order + locate, device side */
  /* Guides outside what the fast path encodes (a guide symbol outside A,C,G,T; a symbol other than
   * A,C,G,T,N in the guide's own PAM; an alt PAM with such a symbol that the genome contains): their
   * hit lists are EMPTY here and guide_flags[i] has bit 0 set - enumerate exactly those guides with
   * gs_enumerate_general (same arguments) and format them with gs_format_guide_ex.  The rest of the
   * batch is unaffected.  guide_flags: n_guides bytes (host memory owned by the result), or NULL
   * when n_unsupported == 0. */
  uint64_t n_unsupported;
  const uint8_t *guide_flags;
  const uint32_t *raw_hits; /* n_guides entries with GS_FLAG_RAW_COUNTS, else NULL */
} gs_result_view;
#define GS_GUIDE_NEEDS_GENERAL 1u

typedef struct gs_result gs_result;

/* ---- index lifecycle ------------------------------------------------------ */

/* Build both strand indexes from the forward genome text (the reference's
 * <fasta>.forward.dna bytes: upper-cased, concatenated, no separators,
 * src/genomics/seq_io.cxx:57-63) and upload them to `device`.
 * Replaces `guidescan index` = sdsl::construct x2 (src/guidescan.cxx:109-179)
 * followed by sdsl::load_from_file x2 (src/guidescan.cxx:198-208).
 * The suffix arrays are built on the GPU. */
gs_status gs_index_build(const uint8_t *text, uint64_t len, int device, gs_index **out);

/* Same, with caller-supplied suffix arrays of text+'\0' and of
 * reverse_complement(text)+'\0' (len+1 uint32 entries each). */
gs_status gs_index_build_with_sa(const uint8_t *text, uint64_t len, const uint32_t *sa_fwd,
                                 const uint32_t *sa_rev, int device, gs_index **out);

/* Import the reference's on-disk index: <prefix>.forward, <prefix>.reverse
 * (sdsl::csa_wt<wt_huff<>,64,8192>::serialize, sdsl/include/sdsl/csa_wt.hpp:372-382).
 * Replaces sdsl::load_from_file (src/guidescan.cxx:198-208).  Both files are turned into text AND suffix array on the
 * device - BWT by wavelet-tree access, LF walks from the SA samples, every step writing text[q - 1] and SA[row] - so no
 * suffix sort runs; .reverse must be the index of reverse_complement(.forward's text) (src/guidescan.cxx:146-157),
 * else GS_ERR_FORMAT.  Without a .reverse file (or with more than 16 distinct symbols) .forward's text is recovered on
 * the host and both strands are built from it. */
gs_status gs_index_open_sdsl(const char *prefix, int device, gs_index **out);
/* The genome text stored in one reference index file (BWT inverted on the host: the importer's host path).
 * *text is malloc'ed: release with gs_free. */
gs_status gs_sdsl_extract_text(const char *index_file, uint8_t **text, uint64_t *len);

/* Native index file: both suffix arrays of a built index (4 bytes per row and strand, with the text's
 * length and a fingerprint), so that a later gs_index_open_sa skips the suffix sort - the part of
 * `guidescan index` worth storing (src/guidescan.cxx:168-175 stores the whole csa_wt; the device
 * layout here is derived data rebuilt from text + suffix arrays in seconds).  `text` as given to
 * gs_index_build.  gs_index_open_sa returns GS_ERR_FORMAT when the file belongs to another text. */
gs_status gs_index_save_sa(gs_index *ix, const uint8_t *text, uint64_t len, const char *path);
gs_status gs_index_open_sa(const uint8_t *text, uint64_t len, const char *path, int device, gs_index **out);

void gs_index_close(gs_index *ix);
/* Work counters of the last gs_enumerate_device call on this handle: [0] extensions executed by the
 * Occ walk, [1] items whose matches overflowed their slots, [2] distinct matches, [4] items seeded
 * from both strands' tables, [5] items seeded one-sided although two-sided seeding was on, [6] guides
 * redone because their matches overflowed the first pass's slots, [7] bit 0: the whole batch was ordered
 * device-wide, bit 1: the redone guides were, bit 2: the overflowing guides' records came out of the arena,
 * bit 3: the device-wide ordering ran as one sort of (sort word, low bits of the first row), bit 4: it had runs
 * to put right afterwards, bit 5: the guides beyond LDS were ordered per guide in tiles (gs_tileorder.hip),
 * bit 6: that form gave up (overlapping PAM patterns, a bucket beyond its slots) and the device-wide one ran
 * (DESIGN.md section 5.3), [13] slots per item of the first pass, [14] / [15] sum and
 * maximum of the per-item match counts; with
 * GS_FLAG_COUNT_REQUESTS also the 64-byte lines requested by the search kernel: [8] prefix-table
 * lines, [9] 16-bit context lines, [10] 32-bit context words, [11] SA/ISA gathers of the search,
 * [12] Occ block lines, [3] lines of the seed recipe lists.  (the bytes the
 * roofline is priced on.)  [7] >> 8: items whose seeds went through PAM-pair tables. */
gs_status gs_index_last_counters(const gs_index *ix, uint64_t out[16]);
/* The last batch's heavy items (DESIGN.md sections 5.1, 5.3): [0] items of which at least one verification pass was
 * handed to other waves, [1] packages reserved in the queue, [2] packages the queue holds, [3] tickets the helping waves
 * drew; [4] guides with an item of more than 2^20 match records, which the per-guide tile ordering leaves to the
 * device-wide ordering - alone, the rest of the batch stays in tiles; [5] launches of the package-running form BEHIND the
 * search launch (the one beside it came too early and left); [6] the form the search ran in: 0 every item with its wave,
 * 1 one launch that publishes and helps, 2 two launches (the plain form publishes, the heavy form beside it runs the
 * packages), 3 the two seeding launches of a batch whose every pattern has its PAM-pair + deep tables and that shares no
 * item (gs_seed.hip); [7] guides of the batch whose own k-mer heads an interval of 8 x share_min rows or more in a strand
 * table: what the form is chosen from, together with the last batch's count of heavy passes. */
gs_status gs_index_last_sharing(const gs_index *ix, uint64_t out[8]);
/* Switches of a handle.  The library's tuning and test switches ("GS_NO_BIDIR", "GS_SHARE_MIN", "GS_DEBUG", ... -
 * DESIGN.md names each where it acts) are a per-handle table: filled from the process environment's GS_* variables
 * ONCE, when the handle is made (gs_index_build / _with_sa / _open_sdsl / _open_sa), and changed only through
 * gs_index_set_option (value NULL removes the entry).  No gs_enumerate* / gs_score* / gs_kmers* call reads the
 * environment: a host that calls setenv from another thread cannot change a running batch.  The reference has no
 * counterpart (its behaviour is fixed at compile time, src/guidescan.cxx:24-27).  gs_index_get_option copies the
 * value into out[cap] (GS_ERR_ARG when the switch is not set or does not fit). */
gs_status gs_index_set_option(gs_index *ix, const char *key, const char *value);
gs_status gs_index_get_option(const gs_index *ix, const char *key, char *out, uint64_t cap);
/* Hold / release the handle's lock (see gs_index above): between the two, calls on this handle
 * from other threads wait, and the device buffers a gs_enumerate_device call returned stay as
 * they are.  Stands where the reference needs nothing (its index is const and its per-call state
 * lives on the caller's stack, include/genomics/index.hpp:102-110). */
gs_status gs_index_lock(gs_index *ix);
gs_status gs_index_unlock(gs_index *ix);
uint64_t gs_index_genome_length(const gs_index *ix); /* sum of chromosome lengths (no sentinel) */
uint64_t gs_index_device_bytes(const gs_index *ix);

/* ---- the hot path ---------------------------------------------------------- */

/* Enumerate all off-targets of n guides.
 *   guides     : n*L ASCII bytes (kmer.sequence), no terminators
 *   guide_pams : n*P ASCII bytes (kmer.pam), P may be 0
 *   alt_pams   : n_alt*P ASCII bytes (-a/--alt-pam), searched before the guide's own PAM
 * Replaces, per guide, process.hpp:51-115: query/PAM preparation, the two
 * genome_index::inexact_search calls (index.hpp:377-398 -> 182-248 -> 125-170),
 * the std::set<match> ordering (process.hpp:21-23) and the resolve() expansion
 * (index.hpp:53-55 -> csa_wt.hpp:333-346).  Bulge budgets are 0 (index.hpp:388-391).
 * Host pointers in, host result out (device work and copies inside). */
gs_status gs_enumerate(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                       const char *guide_pams, uint32_t P, const char *alt_pams, uint32_t n_alt,
                       uint32_t mismatches, uint32_t flags, gs_result **out);

/* Same with the guide/PAM arrays already resident in this device's HBM and the
 * result left in HBM: *d_offsets (n+1 uint64) and *d_hits (gs_hit[]) point into
 * buffers owned by the index handle, valid until the next call on it.
 * `stream` is a hipStream_t (NULL = default stream).  This is what bench.py times. */
gs_status gs_enumerate_device(gs_index *ix, const void *d_guides, uint64_t n, uint32_t L,
                              const void *d_guide_pams, uint32_t P, const char *alt_pams,
                              uint32_t n_alt, uint32_t mismatches, uint32_t flags, void *stream,
                              const void **d_offsets, const void **d_hits, gs_result_view *stats);
/* What the FIRST batch of a shape pays once per handle - the seed recipes, the PAM-pair and deep tables of the PAM patterns
 * (0.3 s and 24-33 GB per strand at hg38 size), the workspace of a batch of n guides - done ahead of the first job: n guides
 * of length L drawn from a generator (not from the genome: few hits), each with the pattern `pam` (P symbols), are enumerated
 * with these alt PAMs, budget and flags and the result is dropped.  Nothing in the reference corresponds (its index is ready
 * when loaded, src/guidescan.cxx:198-211); a service calls this after opening a handle so that a 2,500-guide job
 * (manual/manual.tex:472-475) is not all warm-up.  The batch's own workspace for hits still grows with the first real batch.
 * The call runs a batch on the handle: device results an earlier gs_enumerate_device left in HBM are no longer valid after
 * it; what the handle has learned from the caller's batches (slot sizes, the form of the search) is kept as it was. */
gs_status gs_index_prepare(gs_index *ix, uint64_t n, uint32_t L, const char *pam, uint32_t P, const char *alt_pams,
                           uint32_t n_alt, uint32_t mismatches, uint32_t flags);

/* flags of the last gs_enumerate_device call on this handle (device memory, n bytes, valid until the
 * next call) and how many guides carry GS_GUIDE_NEEDS_GENERAL */
gs_status gs_index_last_guide_flags(const gs_index *ix, const void **d_flags, uint64_t *n_unsupported);
gs_status gs_result_get(const gs_result *r, gs_result_view *view);
void gs_result_free(gs_result *r);

/* Rebuild match.sequence (index.hpp:226,243-244; lower case = mismatch) from the
 * hit key.  `guide`/`L`, `P`, `flags` as passed to gs_enumerate.  out needs L+P+1 bytes. */
gs_status gs_decode_sequence(const char *guide, uint32_t L, uint32_t P, uint32_t flags,
                             uint64_t key, char *out);

/* ---- unit-level entry points (parity tests of SURVEY section 8a rows a6-a8) ---- */

/* Occ(c, i) for c in A,C,G,T at each rows[j] in [0, n]: out[4*j + {0,1,2,3}].
 * Replaces csa_wt::rank_bwt (sdsl/include/sdsl/csa_wt.hpp:270-273). strand 0 = forward index. */
gs_status gs_rank_bwt4(gs_index *ix, int strand, const uint64_t *rows, uint64_t n, uint64_t *out);
/* SA[rows[j]].  Replaces genome_index::resolve (index.hpp:53-55). */
gs_status gs_resolve(gs_index *ix, int strand, const uint64_t *rows, uint64_t n, uint64_t *out);
/* csa.C[csa.char2comp[c]] for c in "ACGT" then 'N' (0 when absent), and csa.size(). */
gs_status gs_index_meta(const gs_index *ix, int strand, uint64_t C_acgtn[5], uint64_t *size);
/* copy the device-resident suffix array back (n = size entries) */
gs_status gs_index_copy_sa(gs_index *ix, int strand, uint32_t *out);
/* The seed plan of k_search as data (host only, no device needed): which depth-k nodes of the
 * reference's search tree (index.hpp:182-248: the variants of a guide's first k consumed symbols with
 * at most m substitutions) a batch shape looks up, and from which strand's table.  Writes the 64-bit
 * recipes (one-sided list | this strand's share | the other strand's share) to out[0..cap) and their
 * counts to counts[3]; astar = NULL: one-sided only.  Recipe: bits 2:0 substitutions n, 5:3 lower bound
 * the other strand's verification applies, 6 reads a rotated copy, 11:7 of which step, then n fields of 7
 * bits from bit 12: step * 4 + digit (the digit-th other base).  n_x = |X|: the leading steps only this
 * strand's table covers; deep: the other strand's recipes index a deep table (steps = guide symbols).
 * gs_debug_choose_thresholds: the cost model's a*(o) for that shape (tests pin both). */
gs_status gs_debug_seed_recipes(uint32_t k, uint32_t L, uint32_t P, uint32_t m, uint32_t n_x, const uint32_t *astar,
                                uint32_t deep, uint64_t *out, uint64_t cap, uint64_t counts[3]);
void gs_debug_choose_thresholds(uint32_t m, uint32_t n_x, uint32_t n_o, uint32_t n_r, double pam_expansions,
                                double verify_a, double verify_b, uint32_t astar[8]);
/* What an item of the two seeding launches (gs_seed.hip) starts from: the 64-byte descriptor k_describe derives from a packed
 * guide record - q (2-bit codes in consumption order), four PAM patterns (3 bits per symbol, 4 = N) - for a batch shape
 * (table depth k, |X| = x_len, the PAM-pair codes of the table slots), as sixteen words: q lo, q hi, pam[4], meta, pidx0,
 * pidxg, qrem_b, bsel_z, bsel_w, qhot, key_a, key_b, guide (host only; tests/test_seed_descriptor.py pins every field). */
gs_status gs_debug_guide_descriptor(uint64_t q, const uint32_t pam[4], uint32_t npams, uint32_t valid, uint32_t L, uint32_t P,
                                    uint32_t k, uint32_t x_len, uint32_t n_pt, const uint32_t code[2], uint32_t out[16]);
/* The tile ordering's plan for one (guide, index) item of `records` match records, as data (host only; tests pin it):
 * out = {buckets the item is dealt into (0: the item is one tile), records a bucket's slot holds, sample words per
 * splitter, records one wave orders, buckets an item may have at most (more: the batch is ordered device-wide)}. */
void gs_debug_tile_plan(uint32_t records, uint32_t out[5]);

/* Self-check of a resident index from the genome text alone (no suffix-array builder involved):
 * the suffix array of `strand` is a permutation of [0, n) (all rows), n_samples evenly spread
 * adjacent row pairs are in suffix order by direct comparison of the text, and the BWT symbol
 * each sampled row's Occ block holds is text[SA[row]-1].  `text` = the forward genome text as given
 * to gs_index_build.  Stands where the reference relies on sdsl::construct being right
 * (sdsl/include/sdsl/construct.hpp:121-166); used by the parity tests at n > 2^31. */
typedef struct {
  uint64_t rows;            /* n = text length + 1 */
  uint64_t not_permutation; /* rows whose value is out of range or was seen before */
  uint64_t sampled;         /* adjacent pairs compared */
  uint64_t out_of_order;    /* pairs with suffix(SA[r]) >= suffix(SA[r+1]) */
  uint64_t undecided;       /* pairs still equal after 65536 comparison steps (N runs are skipped) */
  uint64_t bwt_mismatch;    /* sampled rows whose block symbol differs from the text */
} gs_sa_report;
gs_status gs_index_verify_sa(gs_index *ix, int strand, const uint8_t *text, uint64_t len,
                             uint64_t n_samples, uint64_t seed, gs_sa_report *report);
/* n_samples = GS_VERIFY_ALL_ROWS: EVERY adjacent pair of rows, by the linear-time rule - text[SA[r]] < text[SA[r+1]], or equal
 * symbols and ISA[SA[r]+1] < ISA[SA[r+1]+1] - with ISA checked to be SA's inverse (counted in not_permutation) and the BWT
 * symbol of every row compared with the text: a complete proof that the resident array is the suffix array of `text`
 * (what csa_wt::operator[] presumes, sdsl/include/sdsl/csa_wt.hpp:333-346), one streaming pass (0.4 s per strand at hg38
 * size); `undecided` is 0 by construction.  Needs the inverse suffix array (GS_ERR_UNSUPPORTED on an index built without). */
#define GS_VERIFY_ALL_ROWS (~0ull)

/* ---- text encoders: the step right after the path (host side) ---------------------------- */

typedef struct {
  const char *const *chr_names; /* genome_structure, include/genomics/structures.hpp:45 */
  const uint64_t *chr_lengths;
  uint32_t n_chr;
} gs_genome_structure;

typedef struct { /* the kmer fields the printers use, include/genomics/structures.hpp:10-17 */
  const char *id;
  const char *sequence;
  const char *pam;
  int sense_positive; /* kmers file `sense` column == "+" */
} gs_kmer;

#define GS_TEXT_SAM 0x100u      /* --format sam (default csv) */
#define GS_TEXT_COMPLETE 0x200u /* --mode complete */

/* The database lines of one guide from its hit list (as returned by gs_enumerate, canonical
 * order).  Byte-exact replacement of get_csv_lines / get_sam_lines
 * (include/genomics/printer.hpp:245-300, 302-360) including resolve_absolute
 * (src/genomics/structures.cxx:7-52), CFD and specificity.  `mismatches` and
 * GS_FLAG_PAM_AT_START as passed to gs_enumerate; max_off_targets = -1 for none.
 * *out_text is malloc'ed (NUL terminated); release with gs_free. */
gs_status gs_format_guide(const gs_genome_structure *gs, const gs_kmer *k, const gs_hit *hits,
                          uint64_t n_hits, uint32_t mismatches, uint32_t flags,
                          int64_t max_off_targets, char **out_text, size_t *out_len);
/* Same lines with the guide's specificity supplied by the caller - the float gs_score /
 * gs_score_device computed on the device for the same hits, flags and max_off_targets - so the
 * host formats text only (no CFD arithmetic per hit).  What the C++ host (`guidescan enumerate`)
 * calls. */
gs_status gs_format_guide_scored(const gs_genome_structure *gs, const gs_kmer *k, const gs_hit *hits,
                                 uint64_t n_hits, uint32_t mismatches, uint32_t flags,
                                 int64_t max_off_targets, float specificity, char **out_text,
                                 size_t *out_len);
/* The lines of guides [0, n) of one enumerate result in ONE buffer (same bytes as
 * gs_format_guide_scored guide after guide; rows of guides with skip[g] != 0 left out): what a
 * writer thread calls per contiguous range of a batch.  offsets: n+1 positions into `hits`. */
gs_status gs_format_guides_scored(const gs_genome_structure *gs, const gs_kmer *kmers, uint64_t n,
                                  const uint64_t *offsets, const gs_hit *hits, const float *specificity,
                                  const uint8_t *skip, uint32_t mismatches, uint32_t flags,
                                  int64_t max_off_targets, char **out_text, size_t *out_len);
/* write_sam_header / write_csv_header (include/genomics/printer.hpp:173-187) */
gs_status gs_format_header(const gs_genome_structure *gs, uint32_t flags, char **out_text,
                           size_t *out_len);
void gs_free(void *p);

/* ---- the general path: bulges, symbols outside A,C,G,T, any number of PAMs ---------------------- */

/* One hit of the general path, 48 bytes.  match.sequence can hold '.', lower-case bulge bases, any
 * literal symbol of the guide or of a PAM pattern, and has no fixed length, so it travels as its
 * bytes (byte order == the reference's std::set<match> order). */
typedef struct {
  int64_t pos;
  char seq[32];      /* match.sequence, seq_len bytes, NUL padded */
  uint32_t mismatches;
  uint8_t dna_bulges, rna_bulges;
  uint8_t index;   /* 0 forward index, 1 reverse index */
  uint8_t seq_len;
} gs_hit_ex;

typedef struct gs_result_ex gs_result_ex;

/* Everything genome_index::inexact_search accepts, exactly: replaces both branches of the dispatcher
 * (index.hpp:377-398) - the bulge-aware recursion (:250-375, max_bulge_size = 1 as process.hpp:82-83
 * calls it) when rna_bulges/dna_bulges > 0, else the PAM-aware recursion (:182-248) - for inputs the
 * fast path does not encode: guide symbols outside A,C,G,T (matched literally, else charged a
 * mismatch, :218-247), PAM symbols other than 'N' as literals (:125-170), up to 31 alt PAMs
 * (process.hpp:51-56).  Same set ordering and resolve() expansion; hits per guide in canonical order.
 * A slow path (one node per lane, comparator sort): gs_enumerate reports which guides need it. */
gs_status gs_enumerate_general(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                               const char *guide_pams, uint32_t P, const char *alt_pams, uint32_t n_alt,
                               uint32_t mismatches, uint32_t rna_bulges, uint32_t dna_bulges,
                               uint32_t flags, gs_result_ex **out);
/* The same with alt PAMs of their own lengths: the reference searches every pattern of `-a` next to the
 * guides' PAM whatever its length (process.hpp:51-56 -> index.hpp:182-216: the PAM stage of a pattern ends
 * after its own symbols), so a hit's match.sequence has L + that pattern's length symbols.  alt_pams: the
 * n_alt patterns back to back, alt_lens[j] symbols each (1..8). */
gs_status gs_enumerate_general_pams(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                                    const char *guide_pams, uint32_t P, const char *alt_pams,
                                    const uint32_t *alt_lens, uint32_t n_alt, uint32_t mismatches,
                                    uint32_t rna_bulges, uint32_t dna_bulges, uint32_t flags, gs_result_ex **out);
/* the same function under the name the bulge options were first served by */
gs_status gs_enumerate_bulges(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                              const char *guide_pams, uint32_t P, const char *alt_pams, uint32_t n_alt,
                              uint32_t mismatches, uint32_t rna_bulges, uint32_t dna_bulges,
                              uint32_t flags, gs_result_ex **out);
gs_status gs_result_ex_get(const gs_result_ex *r, uint64_t *n_guides, const uint64_t **guide_offsets,
                           const gs_hit_ex **hits);
/* per guide: hits before duplicate sequences are dropped (what --threshold compares, process.hpp:25-27) */
gs_status gs_result_ex_raw_hits(const gs_result_ex *r, const uint32_t **raw_hits);
void gs_result_ex_free(gs_result_ex *r);
/* match.sequence of a general-path hit as a C string; out needs 33 bytes */
gs_status gs_decode_sequence_ex(const gs_hit_ex *hit, char *out);
/* gs_format_guide for general-path hits (rna_bulges / dna_bulges columns filled in) */
gs_status gs_format_guide_ex(const gs_genome_structure *gs, const gs_kmer *k, const gs_hit_ex *hits,
                             uint64_t n_hits, uint32_t mismatches, uint32_t flags,
                             int64_t max_off_targets, char **out_text, size_t *out_len);

/* CFD score of one hit (include/genomics/printer.hpp:98-113), float semantics preserved. */
float gs_calculate_cfd(const char *sgrna, const char *match_sequence, const char *pam);

/* ---- scoring on the device ---------------------------------- */

/* CFD of every hit and specificity of every guide of an enumerate result, computed in HBM.
 * Replaces calculate_cfd (printer.hpp:98-113) and the aggregation loops of get_csv_lines
 * (printer.hpp:251-297; flags without GS_TEXT_SAM) or off_target_fields (printer.hpp:115-170;
 * GS_TEXT_SAM): float sum of the CFDs in canonical hit order, hits dropped by resolve_absolute
 * (structures.cxx:46-48) left out, --max-off-targets applied per distance exactly as each loop
 * does, `+1` unless a perfect hit with an xGG PAM exists, specificity = 1/sum.  Bit-identical
 * to gs_calculate_cfd / gs_format_guide.
 *   d_guides  : n*L ASCII bytes as passed to gs_enumerate_device
 *   d_offsets, d_hits : what gs_enumerate_device returned (or any CSR hit list in that layout)
 *   flags     : GS_FLAG_PAM_AT_START as passed to gs_enumerate, GS_TEXT_SAM selects the SAM rule
 *   d_cfd     : float[n_hits] or NULL;  d_specificity : float[n] */
gs_status gs_score_device(gs_index *ix, const void *d_guides, uint64_t n, uint32_t L, uint32_t P,
                          uint32_t flags, int64_t max_off_targets, const gs_genome_structure *gs,
                          const void *d_offsets, const void *d_hits, void *stream, void *d_cfd,
                          void *d_specificity);
/* Same with host arrays in and out (copies inside). */
gs_status gs_score(gs_index *ix, const char *guides, uint64_t n, uint32_t L, uint32_t P, uint32_t flags,
                   int64_t max_off_targets, const gs_genome_structure *gs, const uint64_t *offsets,
                   const gs_hit *hits, float *cfd, float *specificity);

/* ---- candidate-guide generation on the device ------------------- */

typedef struct gs_kmers gs_kmers;

/* Every PAM site of ONE chromosome (one FASTA record, any case) in the reference script's order.
 * Replaces find_all_kmers / find_kmers (scripts/generate_kmers.py:70-118): + strand sites per
 * concrete PAM expansion ('N' -> A,C,T,G), then - strand sites per reverse-complemented
 * expansion, each in position order; protospacers with non-ACGT symbols or cut by the record's
 * ends dropped.  `chr` is a host pointer, or a device pointer when chr_on_device != 0.
 * flags: GS_FLAG_PAM_AT_START (--start). */
gs_status gs_kmers_generate(int device, const uint8_t *chr, uint64_t chr_len, int chr_on_device,
                            const char *pam, uint32_t k, uint32_t flags, void *stream, gs_kmers **out);
/* n records: seqs n*k ASCII, pams n*P ASCII (the pattern), positions uint32 (1-based protospacer
 * or PAM start as the script prints it), senses '+'/'-'.  on_device != 0: pointers into HBM, in
 * the layout gs_enumerate_device takes; else host copies owned by the object. */
gs_status gs_kmers_get(gs_kmers *km, int on_device, uint64_t *n, const void **seqs, const void **pams,
                       const void **positions, const void **senses);
void gs_kmers_free(gs_kmers *km);

const char *gs_status_string(gs_status s);
const char *gs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GUIDESCAN_AMD_H */
