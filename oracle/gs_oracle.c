/*
 * gs_oracle.c -- CPU restatement of guidescan's off-target enumeration path.
 * TEST INFRASTRUCTURE ONLY (see gs_oracle.h for the parity status of each part).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  The FM-index here is deliberately naive (byte BWT +
 * per-64-row counts + 1-in-64 SA samples): it restates WHAT csa_wt<wt_huff<>,64,8192>
 * answers (Occ, C, LF, SA[i]), not how SDSL stores it.
 */
#define _GNU_SOURCE
#include "gs_oracle.h"
#include "cfd_table.inc"

#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SA_DENS 64 /* src/guidescan.cxx:24-27: csa_wt<wt_huff<>, 64, 8192> */

struct gso_index {
  uint64_t n;          /* text length incl. sentinel */
  uint8_t *bwt;        /* n bytes */
  uint32_t *sa_full;   /* full SA kept for tests (gso_copy_sa) */
  uint32_t *sa_sample; /* SA[0], SA[64], ... (csa_sampling_strategy.hpp:85-99) */
  uint64_t C[257];     /* C[c] = #symbols < c; valid for present symbols */
  uint8_t present[256];
  int sigma;
  uint8_t sym_idx[256];
  uint32_t *ck; /* [(n/64)+1][sigma] counts before row 64*b */
};

/* ---------- suffix array (own builder; prefix doubling over groups) ---------- */
typedef struct {
  uint64_t key;
  uint32_t idx;
} kv_t;
static int kv_cmp(const void *a, const void *b) {
  const kv_t *x = a, *y = b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  return 0;
}

static void build_sa(const uint8_t *t, uint64_t n, uint32_t *sa) {
  kv_t *kv = malloc(sizeof(kv_t) * n);
  uint32_t *rank = malloc(sizeof(uint32_t) * n);
  /* initial order: first 8 bytes, big endian; beyond the end pads with 0 (sentinel
   * is the unique smallest byte so padded keys stay unique) */
  for (uint64_t i = 0; i < n; i++) {
    uint64_t k = 0;
    for (int j = 0; j < 8; j++) k = (k << 8) | (i + j < n ? t[i + j] : 0);
    kv[i].key = k;
    kv[i].idx = (uint32_t)i;
  }
  qsort(kv, n, sizeof(kv_t), kv_cmp);
  /* rank = start of group */
  uint64_t gs = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (i > 0 && kv[i].key != kv[i - 1].key) gs = i;
    sa[i] = kv[i].idx;
    rank[kv[i].idx] = (uint32_t)gs;
  }
  for (uint64_t h = 8;; h *= 2) {
    int any = 0;
    uint64_t i = 0;
    while (i < n) {
      uint64_t b = i, e = i + 1;
      uint32_t r = rank[sa[b]];
      while (e < n && rank[sa[e]] == r) e++;
      if (e - b > 1) {
        any = 1;
        for (uint64_t j = b; j < e; j++) {
          uint32_t s = sa[j];
          kv[j].idx = s;
          kv[j].key = (s + h < n) ? (uint64_t)rank[s + h] + 1 : 0;
        }
        qsort(kv + b, e - b, sizeof(kv_t), kv_cmp);
        /* two-phase: new ranks computed from the sorted keys, written after */
        uint64_t g = b, prev = kv[b].key;
        for (uint64_t j = b; j < e; j++) {
          if (kv[j].key != prev) g = j;
          prev = kv[j].key;
          sa[j] = kv[j].idx;
          kv[j].key = g; /* reuse as new rank */
        }
        for (uint64_t j = b; j < e; j++) rank[kv[j].idx] = (uint32_t)kv[j].key;
      }
      i = e;
    }
    if (!any) break;
  }
  free(kv);
  free(rank);
}

/* parallel helpers for large inputs (cpu_baseline at hg38 size): chunked over threads */
typedef struct {
  gso_index *ix;
  const uint8_t *t;
  const uint32_t *sa;
  uint64_t lo, hi; /* row range, multiple of 64 at lo */
  uint64_t cnt[256];
  uint32_t base[256];
  int phase;
} bjob_t;

static void *build_worker(void *arg) {
  bjob_t *j = arg;
  gso_index *ix = j->ix;
  if (j->phase == 0) {
    memset(j->cnt, 0, sizeof j->cnt);
    for (uint64_t i = j->lo; i < j->hi; i++) {
      uint8_t c = j->sa[i] ? j->t[j->sa[i] - 1] : j->t[ix->n - 1];
      ix->bwt[i] = c;
      j->cnt[c]++;
    }
  } else {
    uint32_t run[256];
    memcpy(run, j->base, sizeof run);
    for (uint64_t i = j->lo; i < j->hi; i++) {
      if (i % 64 == 0)
        for (int c = 0; c < 256; c++)
          if (ix->present[c]) ix->ck[(i / 64) * ix->sigma + ix->sym_idx[c]] = run[c];
      run[ix->bwt[i]]++;
    }
    if (j->hi == ix->n && ix->n % 64 == 0)
      for (int c = 0; c < 256; c++)
        if (ix->present[c]) ix->ck[(ix->n / 64) * ix->sigma + ix->sym_idx[c]] = run[c];
  }
  return NULL;
}

static gso_index *index_build_impl(const uint8_t *text, uint64_t len, const uint32_t *sa_opt,
                                   int keep_sa, int nthreads) {
  gso_index *ix = calloc(1, sizeof(*ix));
  uint64_t n = len + 1;
  ix->n = n;
  uint8_t *t = malloc(n);
  memcpy(t, text, len);
  t[len] = 0; /* construct.hpp:133-135 appends the 0 sentinel */
  uint32_t *sa_own = NULL;
  const uint32_t *sa = sa_opt;
  if (!sa_opt) {
    sa_own = malloc(sizeof(uint32_t) * n);
    build_sa(t, n, sa_own);
    sa = sa_own;
  }
  ix->bwt = malloc(n);
  if (nthreads < 1) nthreads = 1;
  if (n < 1000000) nthreads = 1;
  bjob_t *jobs = calloc(nthreads, sizeof(bjob_t));
  pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
  uint64_t chunk = ((n / nthreads) / 64 + 1) * 64;
  for (int k = 0; k < nthreads; k++) {
    jobs[k].ix = ix;
    jobs[k].t = t;
    jobs[k].sa = sa;
    jobs[k].lo = (uint64_t)k * chunk < n ? (uint64_t)k * chunk : n;
    jobs[k].hi = (uint64_t)(k + 1) * chunk < n ? (uint64_t)(k + 1) * chunk : n;
    if (k == nthreads - 1) jobs[k].hi = n;
    jobs[k].phase = 0;
  }
  for (int k = 0; k < nthreads; k++) pthread_create(&th[k], NULL, build_worker, &jobs[k]);
  for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
  uint64_t cnt[256] = {0};
  for (int k = 0; k < nthreads; k++)
    for (int c = 0; c < 256; c++) cnt[c] += jobs[k].cnt[c]; /* BWT is a permutation of the text */
  /* byte_alphabet: sdsl/lib/csa_alphabet_strategy.cpp:25-55 */
  uint64_t acc = 0;
  for (int c = 0; c < 256; c++) {
    ix->C[c] = acc;
    ix->present[c] = cnt[c] > 0;
    if (cnt[c]) ix->sym_idx[c] = (uint8_t)ix->sigma++;
    acc += cnt[c];
  }
  ix->C[256] = acc;
  uint64_t nb = n / 64 + 1;
  ix->ck = calloc(nb * ix->sigma, sizeof(uint32_t));
  uint32_t run[256] = {0};
  for (int k = 0; k < nthreads; k++) {
    memcpy(jobs[k].base, run, sizeof run);
    for (int c = 0; c < 256; c++) run[c] += (uint32_t)jobs[k].cnt[c];
    jobs[k].phase = 1;
  }
  for (int k = 0; k < nthreads; k++) pthread_create(&th[k], NULL, build_worker, &jobs[k]);
  for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
  free(jobs);
  free(th);
  uint64_t ns = (n + SA_DENS - 1) / SA_DENS;
  ix->sa_sample = malloc(sizeof(uint32_t) * ns);
  for (uint64_t i = 0; i < ns; i++) ix->sa_sample[i] = sa[i * SA_DENS];
  if (keep_sa) {
    if (sa_own) {
      ix->sa_full = sa_own;
      sa_own = NULL;
    } else {
      ix->sa_full = malloc(sizeof(uint32_t) * n);
      memcpy(ix->sa_full, sa, sizeof(uint32_t) * n);
    }
  }
  free(sa_own);
  free(t);
  return ix;
}

gso_index *gso_index_build(const uint8_t *text, uint64_t len, const uint32_t *sa_opt) {
  return index_build_impl(text, len, sa_opt, 1, 1);
}
/* large inputs: borrow the caller's suffix array (not kept), build with nthreads */
gso_index *gso_index_build_borrow(const uint8_t *text, uint64_t len, const uint32_t *sa, int nthreads) {
  return index_build_impl(text, len, sa, 0, nthreads);
}

void gso_index_free(gso_index *ix) {
  if (!ix) return;
  free(ix->bwt);
  free(ix->sa_full);
  free(ix->sa_sample);
  free(ix->ck);
  free(ix);
}
uint64_t gso_size(const gso_index *ix) { return ix->n; }
uint8_t gso_bwt(const gso_index *ix, uint64_t row) { return ix->bwt[row]; }
void gso_copy_sa(const gso_index *ix, uint32_t *out) {
  if (ix->sa_full) memcpy(out, ix->sa_full, 4 * ix->n);
}

/* csa_wt.hpp:270-273 -> wt_pc.hpp:360-384: number of c in BWT[0,i); a symbol that
 * is not in the text ranks 0 (wt_pc.hpp:363-365). */
uint64_t gso_rank_bwt(const gso_index *ix, uint64_t i, uint8_t c) {
  if (!ix->present[c]) return 0;
  uint64_t b = i / 64;
  uint64_t r = ix->ck[b * ix->sigma + ix->sym_idx[c]];
  for (uint64_t j = b * 64; j < i; j++) r += (ix->bwt[j] == c);
  return r;
}
/* csa.C[csa.char2comp[c]] : char2comp of an absent byte is 0 and C[0]=0
 * (csa_alphabet_strategy.cpp:25-55) */
uint64_t gso_C(const gso_index *ix, uint8_t c) { return ix->present[c] ? ix->C[c] : 0; }

/* csa_wt.hpp:333-346 with LF from suffix_array_helper.hpp:337-349 and
 * wt_pc::inverse_select wt_pc.hpp:396-414 */
uint64_t gso_locate(const gso_index *ix, uint64_t i) {
  uint64_t off = 0;
  while (i % SA_DENS != 0) {
    uint8_t c = ix->bwt[i];
    i = ix->C[c] + gso_rank_bwt(ix, i, c);
    ++off;
  }
  uint64_t result = ix->sa_sample[i / SA_DENS];
  if (result + off < ix->n) return result + off;
  return result + off - ix->n;
}

/* ---------- sequences.cxx:14-46 ---------- */
static char complement_c(char c) {
  switch (c) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'a': return 't';
    case 't': return 'a';
    case 'c': return 'g';
    case 'g': return 'c';
    default: return c;
  }
}
static void complement_s(const char *in, char *out) {
  size_t n = strlen(in);
  for (size_t i = 0; i < n; i++) out[i] = complement_c(in[i]);
  out[n] = 0;
}
static void revcomp_s(const char *in, char *out) {
  size_t n = strlen(in);
  for (size_t i = 0; i < n; i++) out[i] = complement_c(in[n - 1 - i]);
  out[n] = 0;
}

/* ---------- search ---------- */
typedef struct {
  char sequence[48];
  uint64_t sp, ep;
  uint32_t mismatches, dna_bulges, rna_bulges;
} omatch;

typedef struct {
  omatch *v;
  size_t n, cap;
} mvec;

typedef struct {
  const gso_index *ix;
  const char *query;
  int qlen;
  const char *const *pams;
  const int *pam_len;
  int npams;
  int mismatches;
  mvec *sets;    /* one per mismatch count (process.hpp:21-23); NULL when counting */
  uint64_t count; /* off_target_counter process.hpp:27-29 */
  gso_counters *ctr;
  int max_rna, max_dna, max_bulge_size; /* bulge budgets (index.hpp:250-375) */
  uint32_t cur_dna, cur_rna;            /* bulge counts of the match being emitted */
} sctx;

static const char SEARCH_ALPHABET[] = "ATCG"; /* index.hpp:31 */

static uint64_t rk(sctx *s, uint64_t i, char c) {
  s->ctr->n_rank++;
  return gso_rank_bwt(s->ix, i, (uint8_t)c);
}

static void emit(sctx *s, uint64_t sp, uint64_t ep, uint32_t k, const char *seq, int seqlen) {
  if (!s->sets) {
    s->count += ep - sp + 1;
    return;
  }
  mvec *mv = &s->sets[k];
  if (mv->n == mv->cap) {
    mv->cap = mv->cap ? mv->cap * 2 : 16;
    mv->v = realloc(mv->v, mv->cap * sizeof(omatch));
  }
  omatch *m = &mv->v[mv->n++];
  memcpy(m->sequence, seq, seqlen);
  m->sequence[seqlen] = 0;
  m->sp = sp;
  m->ep = ep;
  m->mismatches = k;
  m->dna_bulges = s->cur_dna;
  m->rna_bulges = s->cur_rna;
}

/* index.hpp:125-170  basic recursion with 'N' wildcard; used for the PAM with
 * mismatches=0, k=0.  [begin,end) is consumed right to left. */
static void search_basic(sctx *s, const char *begin, const char *end, uint64_t sp, uint64_t ep,
                         char *seq, int seqlen, size_t mismatches, size_t k, uint32_t k_outer) {
  if (begin == end) { /* :134-137 */
    emit(s, sp, ep, k_outer, seq, seqlen);
    return;
  }
  s->ctr->n_ext++;
  char c = *(end - 1);
  uint64_t occ_before = rk(s, sp, c);
  uint64_t occ_within = rk(s, ep + 1, c) - occ_before;
  if (occ_within > 0) { /* :144-149 */
    uint64_t spp = gso_C(s->ix, (uint8_t)c) + occ_before;
    uint64_t epp = spp + occ_within - 1;
    seq[seqlen] = c;
    search_basic(s, begin, end - 1, spp, epp, seq, seqlen + 1, mismatches, k, k_outer);
  }
  size_t cost = 1;
  if (k >= mismatches && c != 'N') return; /* :152 */
  if (c == 'N') cost = 0;
  for (size_t i = 0; i < 4; i++) { /* :155-169 */
    if (SEARCH_ALPHABET[i] == c) continue;
    char a = SEARCH_ALPHABET[i];
    occ_before = rk(s, sp, a);
    occ_within = rk(s, ep + 1, a) - occ_before;
    if (occ_within > 0) {
      uint64_t spp = gso_C(s->ix, (uint8_t)a) + occ_before;
      uint64_t epp = spp + occ_within - 1;
      seq[seqlen] = a;
      search_basic(s, begin, end - 1, spp, epp, seq, seqlen + 1, mismatches, k + cost, k_outer);
    }
  }
}

/* index.hpp:182-248  bulge-unaware PAM-aware recursion */
static void search_pam(sctx *s, long position, uint64_t sp, uint64_t ep, char *seq, int seqlen,
                       size_t k) {
  if (position < 0) { /* :193-216 */
    for (int p = 0; p < s->npams; p++) {
      const char *pam = s->pams[p];
      search_basic(s, pam, pam + s->pam_len[p], sp, ep, seq, seqlen, 0, 0, (uint32_t)k);
    }
    return;
  }
  s->ctr->n_ext++;
  char c = s->query[position];
  uint64_t occ_before = rk(s, sp, c);
  uint64_t occ_within = rk(s, ep + 1, c) - occ_before;
  if (occ_within > 0) { /* :223-228 */
    uint64_t spp = gso_C(s->ix, (uint8_t)c) + occ_before;
    uint64_t epp = spp + occ_within - 1;
    seq[seqlen] = c;
    search_pam(s, position - 1, spp, epp, seq, seqlen + 1, k);
  }
  if (k >= (size_t)s->mismatches) return; /* :230 */
  for (size_t i = 0; i < 4; i++) {        /* :232-247 */
    if (SEARCH_ALPHABET[i] == c) continue;
    char a = SEARCH_ALPHABET[i];
    occ_before = rk(s, sp, a);
    occ_within = rk(s, ep + 1, a) - occ_before;
    if (occ_within > 0) {
      uint64_t spp = gso_C(s->ix, (uint8_t)a) + occ_before;
      uint64_t epp = spp + occ_within - 1;
      seq[seqlen] = (char)tolower(a); /* :243 */
      search_pam(s, position - 1, spp, epp, seq, seqlen + 1, k + 1);
    }
  }
}

/* index.hpp:12-20 */
typedef struct {
  uint64_t mismatches, dna_bulges, rna_bulges;
  int state; /* 0 none, 1 dna, 2 rna */
  uint64_t curr_bulge_size;
} affinity;

/* index.hpp:250-375  bulge-aware recursion */
static void search_bulge(sctx *s, long position, uint64_t sp, uint64_t ep, char *seq, int seqlen,
                         affinity aff) {
  affinity dna_aff = aff; /* :265-273 */
  if ((uint64_t)s->max_dna > aff.dna_bulges) {
    if (aff.state != 1 || dna_aff.curr_bulge_size == (uint64_t)s->max_bulge_size) {
      dna_aff.state = 1;
      dna_aff.curr_bulge_size = 0;
      dna_aff.dna_bulges += 1;
    }
  }
  /* :275-277: `position != query.length() - 1` compares after conversion to unsigned, so
   * position == -1 passes: a DNA bulge may sit between protospacer and PAM */
  if (dna_aff.state == 1 && dna_aff.curr_bulge_size < (uint64_t)s->max_bulge_size &&
      (size_t)position != (size_t)s->qlen - 1) {
    dna_aff.curr_bulge_size += 1;
    for (size_t i = 0; i < 4; i++) { /* :280-294 */
      char a = SEARCH_ALPHABET[i];
      uint64_t occ_before = rk(s, sp, a);
      uint64_t occ_within = rk(s, ep + 1, a) - occ_before;
      if (occ_within > 0) {
        uint64_t spp = gso_C(s->ix, (uint8_t)a) + occ_before;
        uint64_t epp = spp + occ_within - 1;
        seq[seqlen] = (char)tolower(a);
        search_bulge(s, position, spp, epp, seq, seqlen + 1, dna_aff);
      }
    }
  }
  if (position < 0) { /* :297-314 */
    s->cur_dna = (uint32_t)aff.dna_bulges;
    s->cur_rna = (uint32_t)aff.rna_bulges;
    for (int p = 0; p < s->npams; p++) {
      const char *pam = s->pams[p];
      search_basic(s, pam, pam + s->pam_len[p], sp, ep, seq, seqlen, 0, 0, (uint32_t)aff.mismatches);
    }
    s->cur_dna = s->cur_rna = 0;
    return;
  }
  s->ctr->n_ext++;
  char c = s->query[position];
  uint64_t occ_before = rk(s, sp, c);
  uint64_t occ_within = rk(s, ep + 1, c) - occ_before;
  if (occ_within > 0) { /* :321-331 */
    uint64_t spp = gso_C(s->ix, (uint8_t)c) + occ_before;
    uint64_t epp = spp + occ_within - 1;
    affinity aff_orig = aff;
    aff_orig.state = 0;
    seq[seqlen] = c;
    search_bulge(s, position - 1, spp, epp, seq, seqlen + 1, aff_orig);
  }
  if ((uint64_t)s->mismatches > aff.mismatches) { /* :333-356 */
    for (size_t i = 0; i < 4; i++) {
      if (SEARCH_ALPHABET[i] == c) continue;
      char a = SEARCH_ALPHABET[i];
      occ_before = rk(s, sp, a);
      occ_within = rk(s, ep + 1, a) - occ_before;
      if (occ_within > 0) {
        uint64_t spp = gso_C(s->ix, (uint8_t)a) + occ_before;
        uint64_t epp = spp + occ_within - 1;
        affinity am = aff;
        am.state = 0;
        am.mismatches += 1;
        seq[seqlen] = (char)tolower(a);
        search_bulge(s, position - 1, spp, epp, seq, seqlen + 1, am);
      }
    }
  }
  affinity rna_aff = aff; /* :358-366 */
  if ((uint64_t)s->max_rna > aff.rna_bulges) {
    if (aff.state != 2 || rna_aff.curr_bulge_size == (uint64_t)s->max_bulge_size) {
      rna_aff.state = 2;
      rna_aff.curr_bulge_size = 0;
      rna_aff.rna_bulges += 1;
    }
  }
  if (rna_aff.state == 2 && rna_aff.curr_bulge_size < (uint64_t)s->max_bulge_size &&
      (size_t)position != (size_t)s->qlen - 1) { /* :368-374 */
    rna_aff.curr_bulge_size += 1;
    seq[seqlen] = '.';
    search_bulge(s, position - 1, sp, ep, seq, seqlen + 1, rna_aff);
  }
}

/* index.hpp:377-398 dispatcher (bulge budgets 0 only; the bulge-aware variant
 * :250-375 is not restated: no BASELINE config uses it) */
static void inexact_search(const gso_index *ix, const char *query, const char *const *pams,
                           const int *pam_len, int npams, int mismatches, int max_rna, int max_dna,
                           int max_bulge_size, mvec *sets, uint64_t *count, gso_counters *ctr) {
  sctx s = {ix, query, (int)strlen(query), pams, pam_len, npams, mismatches, sets, 0, ctr,
            max_rna, max_dna, max_bulge_size, 0, 0};
  char seq[64];
  if (max_rna == 0 && max_dna == 0) { /* :388-391 performance path */
    search_pam(&s, (long)s.qlen - 1, 0, ix->n - 1, seq, 0, 0);
  } else {
    affinity aff = {0, 0, 0, 0, 0}; /* :394-397 */
    search_bulge(&s, (long)s.qlen - 1, 0, ix->n - 1, seq, 0, aff);
  }
  if (count) *count += s.count;
}

static int omatch_cmp(const void *a, const void *b) { /* structures.hpp:40-42 */
  return strcmp(((const omatch *)a)->sequence, ((const omatch *)b)->sequence);
}
/* std::set<match> semantics: ordered by sequence, first inserted of equal keys kept */
static void set_normalise(mvec *mv) {
  if (mv->n < 2) return;
  /* stable dedupe: mark order of insertion */
  qsort(mv->v, mv->n, sizeof(omatch), omatch_cmp);
  size_t w = 0;
  for (size_t i = 0; i < mv->n; i++) {
    if (w > 0 && strcmp(mv->v[w - 1].sequence, mv->v[i].sequence) == 0) continue;
    mv->v[w++] = mv->v[i];
  }
  mv->n = w;
}

/* process.hpp:35-115 */
int64_t gso_enumerate(const gso_index *fwd, const gso_index *rev, uint64_t genome_length,
                      const char *seq, const char *pam, const gso_opts *o, gso_hit **out,
                      gso_counters *ctr) {
  gso_counters local = {0, 0, 0};
  if (!ctr) ctr = &local;
  /* :51-61 */
  int npams = 0;
  const char *pams[64];
  char pams_c_buf[64][32];
  const char *pams_c[64];
  int pam_len[64];
  if (pam[0] == 0) {
    pams[npams++] = "";
  } else {
    for (int i = 0; i < o->n_alt_pams; i++) pams[npams++] = o->alt_pams[i];
    pams[npams++] = pam;
  }
  for (int i = 0; i < npams; i++) {
    revcomp_s(pams[i], pams_c_buf[i]);
    pams_c[i] = pams_c_buf[i];
    pam_len[i] = (int)strlen(pams[i]);
  }
  char kmer[64];
  if (!o->start)
    revcomp_s(seq, kmer);
  else
    strcpy(kmer, seq); /* :63 */
  const char *const *use_pams = o->start ? pams : pams_c;

  if (o->threshold > 0) { /* :66-76 */
    uint64_t count = 0;
    gso_counters scratch = {0, 0, 0};
    inexact_search(fwd, kmer, use_pams, pam_len, npams, o->threshold, 0, 0, 0, NULL, &count, &scratch);
    if (count > 1) return -1;
    inexact_search(rev, kmer, use_pams, pam_len, npams, o->threshold, 0, 0, 0, NULL, &count, &scratch);
    if (count > 1) return -1;
  }

  int m = o->mismatches;
  mvec *fs = calloc(m + 1, sizeof(mvec)), *rs = calloc(m + 1, sizeof(mvec));
  inexact_search(fwd, kmer, use_pams, pam_len, npams, m, o->rna_bulges, o->dna_bulges, 1, fs, NULL, ctr); /* :82 */
  inexact_search(rev, kmer, use_pams, pam_len, npams, m, o->rna_bulges, o->dna_bulges, 1, rs, NULL, ctr); /* :83 */

  size_t total = 0;
  for (int d = 0; d <= m; d++) {
    set_normalise(&fs[d]);
    set_normalise(&rs[d]);
    for (size_t i = 0; i < fs[d].n; i++) total += fs[d].v[i].ep - fs[d].v[i].sp + 1;
    for (size_t i = 0; i < rs[d].n; i++) total += rs[d].v[i].ep - rs[d].v[i].sp + 1;
  }
  gso_hit *h = malloc(sizeof(gso_hit) * (total ? total : 1));
  size_t w = 0;
  for (int d = 0; d <= m; d++) { /* :100-115 */
    for (size_t i = 0; i < fs[d].n; i++) {
      omatch *mt = &fs[d].v[i];
      for (uint64_t j = mt->sp; j <= mt->ep; j++) {
        uint64_t r = gso_locate(fwd, j);
        h[w].pos = (int64_t)(0 - r); /* -gi_forward.resolve(j) in size_t arithmetic */
        h[w].mismatches = mt->mismatches;
        h[w].index = 0;
        h[w].sp = mt->sp;
        h[w].ep = mt->ep;
        h[w].row = j;
        strcpy(h[w].sequence, mt->sequence);
        h[w].dna_bulges = mt->dna_bulges;
        h[w].rna_bulges = mt->rna_bulges;
        w++;
      }
    }
    for (size_t i = 0; i < rs[d].n; i++) {
      omatch *mt = &rs[d].v[i];
      for (uint64_t j = mt->sp; j <= mt->ep; j++) {
        uint64_t r = gso_locate(rev, j);
        h[w].pos = (int64_t)(genome_length - (r + 1));
        h[w].mismatches = mt->mismatches;
        h[w].index = 1;
        h[w].sp = mt->sp;
        h[w].ep = mt->ep;
        h[w].row = j;
        strcpy(h[w].sequence, mt->sequence);
        h[w].dna_bulges = mt->dna_bulges;
        h[w].rna_bulges = mt->rna_bulges;
        w++;
      }
    }
  }
  for (int d = 0; d <= m; d++) {
    free(fs[d].v);
    free(rs[d].v);
  }
  free(fs);
  free(rs);
  ctr->n_hit += w;
  if (out)
    *out = h;
  else
    free(h);
  return (int64_t)w;
}

/* ---------- batch (src/guidescan.cxx:226-251) ---------- */
typedef struct {
  const gso_index *fwd, *rev;
  uint64_t genome_length;
  const char *seqs, *pams;
  int L, P;
  uint64_t n;
  const gso_opts *opts;
  int tid, nthreads;
  uint64_t *hit_counts;
  gso_counters ctr;
  int64_t total;
} bjob;

static void *batch_worker(void *arg) {
  bjob *j = arg;
  char seq[64], pam[32];
  for (uint64_t i = j->tid; i < j->n; i += j->nthreads) { /* round robin :229-231 */
    memcpy(seq, j->seqs + i * j->L, j->L);
    seq[j->L] = 0;
    memcpy(pam, j->pams + i * j->P, j->P);
    pam[j->P] = 0;
    int64_t r = gso_enumerate(j->fwd, j->rev, j->genome_length, seq, pam, j->opts, NULL, &j->ctr);
    if (j->hit_counts) j->hit_counts[i] = r < 0 ? 0 : (uint64_t)r;
    if (r > 0) j->total += r;
  }
  return NULL;
}

int64_t gso_enumerate_batch(const gso_index *fwd, const gso_index *rev, uint64_t genome_length,
                            const char *seqs, int L, const char *pams, int P, uint64_t n,
                            const gso_opts *opts, int nthreads, uint64_t *hit_counts,
                            gso_counters *ctr) {
  if (nthreads < 1) nthreads = 1;
  pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
  bjob *jobs = calloc(nthreads, sizeof(bjob));
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (bjob){fwd, rev, genome_length, seqs, pams, L, P, n, opts, t, nthreads, hit_counts,
                     {0, 0, 0}, 0};
    pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  int64_t total = 0;
  for (int t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    total += jobs[t].total;
    if (ctr) {
      ctr->n_ext += jobs[t].ctr.n_ext;
      ctr->n_hit += jobs[t].ctr.n_hit;
      ctr->n_rank += jobs[t].ctr.n_rank;
    }
  }
  free(th);
  free(jobs);
  return total;
}

/* ---------- structures.cxx:7-52 ---------- */
int gso_resolve_absolute(const uint64_t *chr_len, int n_chr, int64_t abs, int seq_len, int pam_len,
                         int64_t *start, char *strand) {
  char st = '+';
  if (abs < 0) { /* :10-13 */
    abs = -abs;
    st = '-';
  }
  int c = -1;
  for (int i = 0; i < n_chr; i++) { /* :16-23 */
    if (abs <= (int64_t)(chr_len[i] - 1)) {
      c = i;
      break;
    } else {
      abs -= (int64_t)chr_len[i];
    }
  }
  if (c < 0) return -1; /* assert(c.name != "") compiled out; sentinel chromosome {"" ,0} */
  int64_t start_position, end_position;
  if (st == '+') { /* :29-35 */
    end_position = abs + 1;
    start_position = end_position - (int64_t)seq_len - (int64_t)pam_len + 1;
  } else { /* :36-42 */
    start_position = abs + 1;
    end_position = start_position + seq_len + pam_len - 1;
  }
  if ((start_position < 0) || (end_position > (int64_t)chr_len[c])) return -1; /* :46-48 */
  *start = start_position;
  *strand = st;
  return c;
}

/* ---------- printer.hpp:98-113 ---------- */
static int base_idx(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': case 'U': return 3;
    default: return -1;
  }
}
float gso_calculate_cfd(const char *sgrna, const char *sequence, const char *pam) {
  if (strlen(sgrna) != 20 || strlen(pam) != 3) return 1.0f;
  float cfd = 1.0f;
  for (int i = 0; i < 20; i++) {
    char g = sgrna[i], t = sequence[i];
    if (g != t) {
      if (g == 'T') g = 'U';
      int r = base_idx(g);
      int d = base_idx((char)toupper(complement_c(t)));
      /* a key absent from the std::map yields 0.0 through operator[] */
      double sc = (r >= 0 && d >= 0) ? gso_cfd_mm[(r * 4 + d) * 20 + i] : 0.0;
      cfd = (float)((double)cfd * sc); /* float *= double */
    }
  }
  int b1 = base_idx(pam[1]), b2 = base_idx(pam[2]);
  if (pam[1] == 'U') b1 = -1;
  if (pam[2] == 'U') b2 = -1;
  double ps = (b1 >= 0 && b2 >= 0) ? gso_cfd_pam[b1 * 4 + b2] : 0.0;
  cfd = (float)((double)cfd * ps);
  return cfd;
}

/* ---------- text output ---------- */
typedef struct {
  char *p;
  size_t n, cap;
} sbuf;
static void sb_add(sbuf *b, const char *s) {
  size_t l = strlen(s);
  if (b->n + l + 1 > b->cap) {
    b->cap = (b->n + l + 1) * 2 + 64;
    b->p = realloc(b->p, b->cap);
  }
  memcpy(b->p + b->n, s, l + 1);
  b->n += l;
}
static void sb_addc(sbuf *b, char c) {
  char t[2] = {c, 0};
  sb_add(b, t);
}
static void sb_addi(sbuf *b, long long v) {
  char t[32];
  snprintf(t, sizeof t, "%lld", v);
  sb_add(b, t);
}
static void sb_addf(sbuf *b, float f) { /* std::to_string(float) == "%f" */
  char t[64];
  snprintf(t, sizeof t, "%f", (double)f);
  sb_add(b, t);
}

static void split_by_distance(const gso_hit *hits, int64_t n, int m, int64_t *beg) {
  /* hits are in canonical order: distance ascending */
  int64_t i = 0;
  for (int d = 0; d <= m; d++) {
    beg[d] = i;
    while (i < n && (int)hits[i].mismatches == d) i++;
  }
  beg[m + 1] = i;
}

static void match_pam(const char *match_sequence, char *pam) { /* printer.hpp:139-143 */
  if (strlen(match_sequence) < 20) {
    pam[0] = 0;
  } else {
    size_t l = strlen(match_sequence + 20);
    if (l > 3) l = 3;
    memcpy(pam, match_sequence + 20, l);
    pam[l] = 0;
  }
}

/* printer.hpp:245-300 */
char *gso_csv_lines(const char *const *chr_names, const uint64_t *chr_len, int n_chr,
                    const char *id, const char *seq, const char *pam, int dir_positive,
                    const gso_opts *o, const gso_hit *hits, int64_t n_hits) {
  (void)dir_positive;
  sbuf out = {0}, lines = {0};
  sb_add(&out, "");
  sb_add(&lines, "");
  int m = o->mismatches;
  int64_t beg[16];
  split_by_distance(hits, n_hits, m, beg);
  float cfd_sum = 0.0f;
  int perfect = 0, nomatch = 1;
  char sequence[128];
  if (o->start)
    snprintf(sequence, sizeof sequence, "%s%s", pam, seq);
  else
    snprintf(sequence, sizeof sequence, "%s%s", seq, pam);
  /* remember where each kept line ends so specificity can be appended */
  size_t nlines = 0, lcap = 16;
  size_t *ends = malloc(sizeof(size_t) * lcap);
  for (int d = 0; d <= m; d++) {
    for (int64_t i = 0; i < beg[d + 1] - beg[d]; i++) {
      nomatch = 0;
      if (o->max_off_targets != -1 && i >= o->max_off_targets) break; /* :259 */
      const gso_hit *h = &hits[beg[d] + i];
      char ms[64], mp[8];
      complement_s(h->sequence, ms);
      match_pam(ms, mp);
      if (h->mismatches == 0 && strlen(mp) == 3 && mp[1] == 'G' && mp[2] == 'G') perfect = 1;
      int64_t start;
      char strand;
      int c = gso_resolve_absolute(chr_len, n_chr, h->pos, (int)strlen(seq), (int)strlen(pam),
                                   &start, &strand);
      if (c < 0) continue; /* get_csv_line returns "" :210 */
      sb_add(&lines, id);
      sb_addc(&lines, ',');
      sb_add(&lines, sequence);
      sb_addc(&lines, ',');
      sb_add(&lines, chr_names[c]);
      sb_addc(&lines, ',');
      sb_addi(&lines, start);
      sb_addc(&lines, ',');
      sb_addc(&lines, strand);
      sb_addc(&lines, ',');
      sb_addi(&lines, h->mismatches);
      if (o->complete) {
        sb_addc(&lines, ',');
        sb_add(&lines, ms);
        sb_addc(&lines, ',');
        sb_addi(&lines, h->rna_bulges); /* printer.hpp:235-239 */
        sb_addc(&lines, ',');
        sb_addi(&lines, h->dna_bulges);
      }
      if (nlines == lcap) {
        lcap *= 2;
        ends = realloc(ends, sizeof(size_t) * lcap);
      }
      ends[nlines++] = lines.n;
      cfd_sum += gso_calculate_cfd(seq, ms, mp);
    }
  }
  if (nomatch) { /* :189-199 */
    sb_add(&out, id);
    sb_addc(&out, ',');
    sb_add(&out, sequence);
    sb_add(&out, ",NA,NA,NA,0");
    if (o->complete) sb_add(&out, ",NA,NA,NA");
    sb_add(&out, ",1.0\n");
  } else {
    float specificity = 0.0f;
    if (!perfect) cfd_sum += 1;
    if (cfd_sum > 0) specificity = 1 / cfd_sum;
    size_t prev = 0;
    for (size_t i = 0; i < nlines; i++) {
      char save = lines.p[ends[i]];
      lines.p[ends[i]] = 0;
      sb_add(&out, lines.p + prev);
      lines.p[ends[i]] = save;
      prev = ends[i];
      sb_addc(&out, ',');
      sb_addf(&out, specificity);
      sb_addc(&out, '\n');
    }
  }
  free(ends);
  free(lines.p);
  return out.p;
}

static void hex_le(sbuf *b, uint64_t num) { /* printer.hpp:18-79 */
  static const char *hx = "0123456789abcdef";
  char t[17];
  for (int i = 0; i < 8; i++) {
    unsigned char lo = num % 256;
    num /= 256;
    t[2 * i] = hx[lo / 16];
    t[2 * i + 1] = hx[lo % 16];
  }
  t[16] = 0;
  sb_add(b, t);
}

/* printer.hpp:115-170 and 302-360 */
char *gso_sam_lines(const char *const *chr_names, const uint64_t *chr_len, int n_chr,
                    const char *id, const char *seq, const char *pam, int dir_positive,
                    const gso_opts *o, const gso_hit *hits, int64_t n_hits) {
  sbuf out = {0}, hex = {0};
  sb_add(&out, "");
  sb_add(&hex, "");
  int m = o->mismatches;
  int64_t beg[16];
  split_by_distance(hits, n_hits, m, beg);
  int64_t delim = 0;
  for (int i = 0; i < n_chr; i++) delim += (int64_t)chr_len[i];
  delim = -(delim + 1); /* :90-96 */
  float cfd_sum = 0.0f;
  int perfect = 0;
  for (int d = 0; d <= m; d++) {
    int64_t n_off = 0;
    for (int64_t i = 0; i < beg[d + 1] - beg[d]; i++) {
      if (o->max_off_targets != -1 && n_off >= o->max_off_targets) break; /* :129 */
      const gso_hit *h = &hits[beg[d] + i];
      char ms[64], mp[8];
      complement_s(h->sequence, ms);
      match_pam(ms, mp);
      if (h->mismatches == 0 && strlen(mp) == 3 && mp[1] == 'G' && mp[2] == 'G') perfect = 1;
      int64_t start;
      char strand;
      int c = gso_resolve_absolute(chr_len, n_chr, h->pos, (int)strlen(seq), (int)strlen(pam),
                                   &start, &strand);
      if (c < 0) continue; /* :151 */
      hex_le(&hex, (uint64_t)h->pos);
      cfd_sum += gso_calculate_cfd(seq, ms, mp);
      n_off++;
    }
    hex_le(&hex, (uint64_t)d);
    hex_le(&hex, (uint64_t)delim);
  }
  float specificity = 0.0f;
  if (!perfect) cfd_sum += 1;
  if (cfd_sum > 0) specificity = 1 / cfd_sum;

  char sequence[128], rc[128];
  if (o->start)
    snprintf(sequence, sizeof sequence, "%s%s", pam, seq);
  else
    snprintf(sequence, sizeof sequence, "%s%s", seq, pam);
  revcomp_s(sequence, rc);
  for (int64_t i = beg[0]; i < beg[1]; i++) { /* one line per distance-0 hit :314-357 */
    const gso_hit *h = &hits[i];
    int64_t start = 0;
    char strand = 0;
    int c = gso_resolve_absolute(chr_len, n_chr, h->pos, (int)strlen(seq), (int)strlen(pam),
                                 &start, &strand);
    sb_add(&out, id);
    sb_addc(&out, '\t');
    sb_add(&out, dir_positive ? "0" : "16");
    sb_addc(&out, '\t');
    sb_add(&out, c >= 0 ? chr_names[c] : ""); /* no sentinel check :321-332 */
    sb_addc(&out, '\t');
    sb_addi(&out, c >= 0 ? start : 0);
    sb_add(&out, "\t100\t");
    sb_addi(&out, (long long)strlen(sequence));
    sb_add(&out, "M\t*\t0\t0\t");
    sb_add(&out, dir_positive ? sequence : rc);
    sb_add(&out, "\t*");
    for (int k = 0; k <= m; k++) {
      sb_add(&out, "\tk");
      sb_addi(&out, k);
      sb_add(&out, ":i:");
      sb_addi(&out, beg[k + 1] - beg[k]);
    }
    if (o->complete) {
      sb_add(&out, "\tof:H:");
      sb_add(&out, hex.p);
    }
    sb_add(&out, "\tsp:f:");
    sb_addf(&out, specificity);
    sb_addc(&out, '\n');
  }
  free(hex.p);
  return out.p;
}

void gso_free(void *p) { free(p); }

/* ---------- brute force (independent definition) ---------- */
int64_t gso_bruteforce(const uint8_t *text, uint64_t len, const char *pat, int L, const char *pam,
                       int P, int mismatches, uint64_t *out_pos, uint32_t *out_mm, int64_t cap) {
  /* occurrences of pat (<= mismatches substitutions, text bases limited to ACGT at
   * mismatching positions, exact otherwise) immediately followed by pam where
   * pam 'N' matches any of A,C,G,T,N and other pam chars match themselves */
  int64_t cnt = 0;
  if (len < (uint64_t)(L + P)) return 0;
  for (uint64_t i = 0; i + L + P <= len; i++) {
    int mm = 0, ok = 1;
    for (int j = 0; j < L && ok; j++) {
      uint8_t t = text[i + j];
      if (t != (uint8_t)pat[j]) {
        if (t == 'A' || t == 'C' || t == 'G' || t == 'T')
          mm++;
        else
          ok = 0;
        if (mm > mismatches) ok = 0;
      }
    }
    for (int j = 0; j < P && ok; j++) {
      uint8_t t = text[i + L + j];
      if (pam[j] == 'N') {
        if (!(t == 'A' || t == 'C' || t == 'G' || t == 'T' || t == 'N')) ok = 0;
      } else if (t != (uint8_t)pam[j])
        ok = 0;
    }
    if (ok) {
      if (cnt < cap) {
        out_pos[cnt] = i;
        out_mm[cnt] = (uint32_t)mm;
      }
      cnt++;
    }
  }
  return cnt;
}
