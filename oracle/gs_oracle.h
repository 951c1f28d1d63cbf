/*
 * gs_oracle.h -- CPU restatement of guidescan's off-target enumeration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under guidescan-cli_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / baseline.
 *
 * Parity status: PINNED against the reference itself, compiled in
 * place from /root/reference by oracle/Makefile into oracle/_ref/ (g++ only):
 *   - rank_bwt / LF / C / char2comp / resolve_absolute / (reverse_)complement /
 *     CFD tables: against libgs_ref.so (wt_huff, byte_alphabet, structures.cxx,
 *     sequences.cxx, doench.hpp), tests/test_oracle_vs_ref.py;
 *   - inexact_search recursion (plain, N-wildcard PAM, bulge-aware), the per-guide
 *     pipeline (PAM list, set order/dedupe, threshold, expansion, coordinates) and
 *     the CSV/SAM printers incl. calculate_cfd: against gs_ref_enumerate (the
 *     reference's index.hpp / process.hpp / printer.hpp + csa_wt, whole output
 *     files compared), tests/test_oracle_vs_ref_pipeline.py.
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gso_index gso_index;

/* text: len bytes (no sentinel; '\0' is appended internally, as sdsl::construct
 * does, sdsl/include/sdsl/construct.hpp:133-135).  sa_opt: optional suffix array
 * of the len+1 suffixes (uint32), or NULL to build one here. */
gso_index *gso_index_build(const uint8_t *text, uint64_t len, const uint32_t *sa_opt);
/* large inputs: sa is borrowed (not kept, gso_copy_sa unavailable), built with nthreads */
gso_index *gso_index_build_borrow(const uint8_t *text, uint64_t len, const uint32_t *sa, int nthreads);
void gso_index_free(gso_index *);
uint64_t gso_size(const gso_index *);                            /* csa.size() = len+1 */
uint64_t gso_rank_bwt(const gso_index *, uint64_t i, uint8_t c); /* csa_wt.hpp:270-273 */
uint64_t gso_C(const gso_index *, uint8_t c);                    /* csa.C[csa.char2comp[c]] */
uint64_t gso_locate(const gso_index *, uint64_t row);            /* csa[row], csa_wt.hpp:333-346 */
uint8_t gso_bwt(const gso_index *, uint64_t row);
/* copy of the full suffix array (n entries); for tests of the product's builder */
void gso_copy_sa(const gso_index *, uint32_t *out);

typedef struct {
  int64_t pos;          /* signed absolute coordinate, process.hpp:104,111 */
  uint32_t mismatches;  /* match.mismatches */
  uint32_t index;       /* 0 = found in forward index, 1 = reverse index */
  uint64_t sp, ep;      /* SA interval of the match */
  uint64_t row;         /* SA row this hit came from */
  char sequence[48];    /* match.sequence (NUL terminated) */
  uint32_t dna_bulges;  /* match.dna_bulges */
  uint32_t rna_bulges;  /* match.rna_bulges */
} gso_hit;

typedef struct {
  uint64_t n_ext;  /* a3 invocations with position>=0 + a4 invocations with begin!=end */
  uint64_t n_hit;  /* hit positions emitted */
  uint64_t n_rank; /* rank_bwt calls */
} gso_counters;

typedef struct {
  int mismatches;          /* -m */
  int start;               /* --start */
  int n_alt_pams;          /* -a */
  const char *const *alt_pams;
  int64_t max_off_targets; /* --max-off-targets, -1 = none */
  int complete;            /* --mode complete */
  int threshold;           /* -t */
  int rna_bulges;          /* --rna-bulges */
  int dna_bulges;          /* --dna-bulges */
} gso_opts;

/* One guide: process.hpp:35-115.  Returns number of hits, *out malloc'ed in
 * canonical order (distance asc; forward-index matches then reverse-index
 * matches; matches by sequence bytes; rows asc).  Returns -1 if the guide is
 * skipped by the --threshold pre-filter. */
int64_t gso_enumerate(const gso_index *fwd, const gso_index *rev, uint64_t genome_length,
                      const char *seq, const char *pam, const gso_opts *opts,
                      gso_hit **out, gso_counters *ctr);

/* Batch over nthreads std::thread-like workers (pthread), guide i -> thread
 * i % nthreads as src/guidescan.cxx:229-231.  Guides are n fixed-width rows.
 * hit_counts (n entries, may be NULL) receives hits per guide.  Returns total hits. */
int64_t gso_enumerate_batch(const gso_index *fwd, const gso_index *rev, uint64_t genome_length,
                            const char *seqs, int L, const char *pams, int P, uint64_t n,
                            const gso_opts *opts, int nthreads, uint64_t *hit_counts,
                            gso_counters *ctr);

/* structures.cxx:7-52.  Returns chromosome index or -1 (sentinel); *start 1-based, *strand '+'/'-' */
int gso_resolve_absolute(const uint64_t *chr_len, int n_chr, int64_t abs, int seq_len, int pam_len,
                         int64_t *start, char *strand);

float gso_calculate_cfd(const char *sgrna, const char *sequence, const char *pam); /* printer.hpp:98-113 */

/* printer.hpp:201-300 / 302-360.  Return malloc'ed NUL-terminated text. */
char *gso_csv_lines(const char *const *chr_names, const uint64_t *chr_len, int n_chr,
                    const char *id, const char *seq, const char *pam, int dir_positive,
                    const gso_opts *opts, const gso_hit *hits, int64_t n_hits);
char *gso_sam_lines(const char *const *chr_names, const uint64_t *chr_len, int n_chr,
                    const char *id, const char *seq, const char *pam, int dir_positive,
                    const gso_opts *opts, const gso_hit *hits, int64_t n_hits);
void gso_free(void *);

/* brute force: all text positions whose window matches (independent definition of
 * the hit set, not derived from the reference's traversal).  Used to cross-check
 * the restated recursion.  Returns count; out receives start offsets / mismatch counts. */
int64_t gso_bruteforce(const uint8_t *text, uint64_t len, const char *pattern20, int L,
                       const char *pam_pattern, int P, int mismatches,
                       uint64_t *out_pos, uint32_t *out_mm, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
