/* gs_common.h -- internal declarations shared by the translation units of libgsamd.so */
#ifndef GS_COMMON_H
#define GS_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "guidescan_amd.h"

/* ---- device-resident FM-index of one strand ---------
 * Occ block = 64 bytes covering 128 BWT rows:
 *   uint32 cnt[4]   occurrences of A,C,G,T in BWT[0, 128*b)
 *   uint64 lo[2]    bit j of word w: low  bit of the 2-bit code of row 128*b+64*w+j
 *   uint64 hi[2]    high bit of the code (A=0 C=1 G=2 T=3)
 *   uint64 ex[2]    1 = row holds a non-ACGT symbol ('\0' sentinel, N, IUPAC) or padding
 * One aligned 64-byte read answers Occ(c, i) for all four bases exactly, including
 * rows whose BWT symbol is not a base (no side table on the fast path). */
/* overflow arena of k_search (gs_search.hip): records per chunk (16 KiB): one atomic per 1,024 matches */
#define ARENA_SHIFT 10u
#define ARENA_CHUNK (1u << ARENA_SHIFT)

#define GS_BLOCK_ROWS 128u
#define GS_BLOCK_SHIFT 7u

/* A PAM-pair table (gs_pairtab.hip): the depth-k prefix table restricted to the rows whose left context
 * has the pair `code` (two 2-bit symbols) at offsets v_rem-2, v_rem-1 and only A,C,G,T nearer than
 * that.  Entry = 8 bytes {first of its rows in the table's own row arrays, rows (6 bits) | filter (26 bits)}:
 * the 16 two-symbol extensions of a (k-2)-mer are ONE 128-byte block.  rows = 63: 63 or more - the row
 * arrays then hold a header slot at `first` (rowid[first] = the count) and the rows behind it.  Filter:
 * with one row its nearest 13 context symbols (2 bits each); with more, for each of the nearest 6
 * context positions the set of symbols its rows show there (4 bits each): a query symbol outside the
 * set is a substitution in every row. */
#define GS_PT_BIG 63u
struct gs_pairtab_dev {
  const uint2 *tab;
  const uint2 *rot;      /* rotated copies of consumption steps rot_first .. k-3 */
  const uint16_t *c16;   /* per row of the table: nearest eight context symbols */
  const uint32_t *ctx;   /* all sixteen */
  const uint32_t *rowid; /* its row in the strand's suffix array */
  uint32_t rot_first;
  uint32_t code;
  /* the deep table of the strand seen from the other strand's items (gs_pairtab.hip), or nullptr */
  const uint4 *deep;
};
struct gs_strand_dev {
  const uint4 *blocks;       /* (n >> 7) + 1 blocks of 4 x uint4 */
  const uint32_t *sa;        /* full suffix array, n entries */
  const uint32_t *run_start; /* first row of each maximal run of 'N' in the BWT */
  const uint32_t *run_cum;   /* run_cum[r] = number of N rows in runs < r ; nruns+1 entries */
  uint32_t n;                /* rows = text length + 1 */
  uint32_t nruns;
  uint32_t C[4]; /* first row of the A, C, G, T ranges */
  uint32_t CN;   /* first row of the N range (undefined when has_n == 0) */
  uint32_t has_n;
  /* prefix interval table: one 16-byte entry per k-mer, indexed by the
   * k-mer with its FIRST text symbol in the LOWEST bits, so the 16 two-symbol left extensions of
   * a (k-2)-mer are contiguous.  Entry = {sp, cnt | flag<<31, mask_lo, mask_hi}:
   * cnt==0: k-mer absent; flag: some row of the interval has a non-ACGT symbol (or the text
   * start) within the 16 symbols preceding its suffix, so it must take the Occ walk;
   * mask (64 bits in z, w): four 16-bit sets; bit 16j + v: some row of the interval is preceded by
   * the symbol pair v (two symbols, 2 bits each, nearest first) at distances mask_off[j],
   * mask_off[j]+1 (0 = the symbol right before the suffix).  The offsets are 0, 2, 4 and - so that
   * for 20-mers with a 3-symbol PAM whose first symbol is a wildcard the fourth pair is the PAM's two
   * fixed symbols - 21-k (7 at k = 14); any offsets are correct, they only set the filter's power.  A site with at
   * most b substitutions among the symbols a seed still has to match leaves at least (pairs - b)
   * of its query pairs intact, so a seed whose interval shows fewer of them cannot reach a hit. */
  const uint4 *ptab;
  uint32_t mask_off; /* 4 bits per pair position */
  /* rotated copies of the table: copy p (p < k-2) has the symbol of
   * consumption step p moved to the lowest index bits, so the three substitutions at step p
   * of an otherwise fixed k-mer are neighbours in one 64-byte line.  Steps rot_first .. k-2, back to back. */
  const uint4 *ptab_rot;
  uint32_t rot_first; /* first consumption step that has a copy (31: none).  The lowest steps are the
                         other strand's PAM steps, never substituted, and the last substituted step of
                         few one-sided seeds: their copies (4.3 GB each at hg38 size) are not built */
  /* preceding context: ctx[r] = the 16 text symbols before suffix SA[r],
   * nearest first, 2 bits each (A,C,G,T = 0..3).  Lets a small interval at depth k be resolved
   * against the rest of the pattern with one 4-byte read per row instead of an Occ walk. */
  const uint32_t *ctx;
  /* the nearest 8 of them (low 16 bits of ctx[r]): the first level of the verification reads
   * these, 8 rows per 16-byte load, and only the few surviving rows touch ctx[] */
  const uint16_t *ctx16;
  /* inverse suffix array (isa[sa[r]] = r), n entries, or nullptr: turns a text position found
   * through the other strand's index into this strand's row (two-sided seeding) */
  const uint32_t *isa;
  /* exception rows: the rows of valid k-mer intervals whose 16-symbol left
   * context holds a symbol outside A,C,G,T or runs off the text start - ctx[] cannot say so (it
   * stores 2 bits per symbol).  Sorted by row; exc_sym[i] = the 16 symbols as nibbles, nearest
   * first: 0..3 A,C,G,T, 4 'N', 5 any other symbol, 6 before the text start.  A few rows per N run:
   * the verification looks a row up here only when its table entry carries the flag. */
  const uint32_t *exc_row;
  const uint64_t *exc_sym;
  uint32_t n_exc;
  /* every symbol outside A,C,G,T (the general path, gs_general.hip): maximal runs of equal such
   * symbols in the BWT, grouped by symbol.  xr_seg[c] = {first run of symbol c, number of runs};
   * xr_start[i] = first row of run i, xr_cum[i] = rows of the symbol in its earlier runs (one entry
   * more per symbol: its total); C256[c] = csa.C[csa.char2comp[c]] for every byte present. */
  const uint32_t *xr_start;
  const uint32_t *xr_cum;
  const uint2 *xr_seg;
  const uint32_t *C256;
};

struct gs_strand {
  gs_strand_dev d{};
  void *blocks = nullptr, *sa = nullptr, *run_start = nullptr, *run_cum = nullptr, *ptab = nullptr, *ctx = nullptr, *ctx16 = nullptr, *ptab_rot = nullptr, *isa = nullptr, *exc_row = nullptr, *exc_sym = nullptr, *xr_start = nullptr, *xr_cum = nullptr, *xr_seg = nullptr, *C256 = nullptr;
  bool has_sym[256] = {false}; /* bytes present in this strand's text */
  /* the rotated table copies this strand may hold (gs_strand_rot_ensure builds them on first use): first step, count (0: none), table depth */
  uint32_t rot_plan_first = 31, rot_plan_n = 0, rot_k = 0;
  uint64_t n = 0;
  uint64_t C_acgtn[5] = {0, 0, 0, 0, 0};
  uint64_t bytes = 0;
};

/* a maximal run of 'N' bytes in the forward text with the bytes around it (host side; used per
 * batch to list the few windows where a PAM 'N' can meet a literal N, index.hpp:139-149) */
#define GS_NRUN_FLANK 40
struct gs_nrun {
  uint64_t start, len;
  uint8_t left[GS_NRUN_FLANK];  /* text[start-40 .. start), 0 beyond the text */
  uint8_t right[GS_NRUN_FLANK]; /* text[start+len .. start+len+40) */
};

struct gs_buffer {
  void *p = nullptr;
  size_t cap = 0;
};

struct gs_pairtab_host {
  bool valid = false;
  uint32_t v_rem = 0, code = 0, rot_first = 31;
  gs_pairtab_dev d[2]{};
  void *mem[2][8] = {{nullptr}, {nullptr}};
  bool deep = false;               /* both strands' deep tables exist (PAM length deep_P) */
  uint32_t deep_P = 0, deep_kb = 0;
  uint64_t bytes = 0, used = 0;
};

struct gs_recipe_set {
  gs_buffer buf;
  uint64_t key[2] = {0, 0};
  bool valid = false;
  uint32_t n_full = 0, n_a = 0, n_b = 0, n_a8 = 0; /* full | a | b | a as read through PAM-pair tables */
  uint32_t a_rot_first = 31; /* lowest consumption step whose rotated copy the PAM-pair list reads (31: none) */
};

struct gs_index {
  /* Switches of the library (tuning experiments, the forms the tests force): the process environment's GS_* variables
   * as they were when the handle was made, then whatever gs_index_set_option changed.  No entry point reads the
   * environment after that (a multithreaded host may setenv at any time): gs_opt() is the only reader. */
  std::map<std::string, std::string> opts;
  /* One batch at a time per handle: the workspace, the lazily built tables and the result buffers belong to the handle,
   * while the reference's seam is called from N host threads on one const index (src/guidescan.cxx:240-247).  Every
   * entry point that takes a handle holds this lock for its whole call (host-pointer entry points copy their
   * results out under it: they are safe from any number of threads); recursive because the host-pointer entry points
   * call the device ones.  gs_index_lock / gs_index_unlock hold it across several device-pointer calls. */
  mutable std::recursive_mutex mtx;
  int device = 0;
  uint64_t genome_length = 0;
  gs_strand strand[2];
  /* per-handle workspace, grown on demand, reused across calls */
  gs_buffer w_guides, w_slots, w_counts, w_nmatch, w_nhits, w_offsets, w_hits, w_misc, w_blocksums,
      w_grec, w_flags, w_raw, w_ovf_list, w_grec2, w_slots2, w_counts2, w_nmatch2, w_nhits2, w_h_off, w_h_tmp,
      /* device-wide ordering of guides with more matches than an LDS sort holds (gs_search.hip) */
      w_b_src, w_b_cnt, w_b_prefix, w_b_recs, w_b_w0, w_b_w0b, w_b_idx, w_b_idxb, w_b_keep, w_b_keeps,
      w_b_rows, w_b_rowss, w_b_redo_pos, w_b_s, w_b_tab,
      w_score, w_score_io, w_score_tmp, /* gs_score.hip: score tables + chromosome prefix sums; host-pointer staging */
      /* overflow arena of k_search (gs_search.hip): records, chunk owners + sequence numbers, chunks per item */
      w_arena, w_arena_meta, w_nchunk,
      /* heavy items shared among waves (gs_search_args::shq): the package queue; counters, flags, shared-item list, sums, directory bases */
      w_shq, w_sh_meta,
      /* per-guide ordering in LDS tiles (gs_tileorder.hip): k_search's per-class counts, the plan's scans, tile
       * descriptors, bucket space, chunk index, partitioned items, class starts, rank tables + flags */
      w_cls, w_desc, w_sched, /* gs_seed.hip: descriptors per guide; work counters, histograms, the two schedules */
      w_t_plan, w_t_tiles, w_t_buckets, w_t_chunkof, w_t_big, w_t_rel, w_t_tab, w_t_excl, w_t_spill, w_b_redo_pos2;
  uint32_t share_backoff = 0; /* batches this handle still runs without sharing after a sharing launch was not resident as a whole */
  bool share_timed_out = false; /* a helping wave gave up waiting for a package (k_search_body): the call fails, gs_enumerate_device redoes the batch without sharing */
  uint32_t opt_share_min = 512, opt_share_max = 2048; /* groups of eight rows: a verification pass of share_min or more is handed out, in packages of at most share_max (0: items are never shared) */
  unsigned long long last_share[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* the last batch: shared items, packages reserved, queue capacity, tickets handed out; [4] guides beyond the tiles' reach, ordered device-wide alone */
  uint64_t shq_packages = 16384; /* packages the next batch's queue holds (1,152 bytes each): grown when a batch reserved more */
  uint64_t arena_chunks = 4096; /* chunks of 1,024 records the next batch's arena holds: grown when a batch needed more */
  /* matches per item the last batch showed, per mismatch budget (slot sizing), and what it was measured on */
  double seen_mean[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  double seen_max[8] = {0};
  uint64_t seen_hpass[8] = {0}, seen_items[8] = {0}; /* verification passes of share_min row groups or more in the last batch at this budget, and its items */
  bool last_raw_valid = false;   /* w_raw holds the raw hit counts of the last batch (GS_FLAG_RAW_COUNTS) */
  uint64_t last_unsupported = 0; /* guides of the last batch flagged GS_GUIDE_NEEDS_GENERAL (w_flags) */
  uint64_t seen_key[8] = {0};
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_tile = nullptr;  /* gs_tileorder_run: behind the copy of the slow-tile counts */
  hipStream_t st_help = nullptr; /* lowest priority: the launch that runs published packages next to the search launch (run_search) */
  hipEvent_t ev_help[2] = {nullptr, nullptr};
  uint32_t *h_pin = nullptr;     /* 256 bytes of page-locked host memory: small results of asynchronous copies that are read behind an event */
  std::atomic<uint64_t> lock_owner{0}; /* gs_index_lock: the thread that holds the handle (0: none), and how many times */
  uint32_t lock_depth = 0;
  uint32_t pt_k = 0; /* depth of the prefix interval tables (0: none) */
  unsigned long long last_counters[16] = {0}; /* k_search's stats array of the last gs_enumerate_device call */
  std::vector<gs_nrun> nruns_text; /* 'N' runs of the forward text */
  gs_buffer w_cand;                /* literal-N candidate windows (device): kept from batch to batch of one shape */
  uint64_t cand_key = ~0ull;       /* (L, P, bucketed or not) w_cand was made for; ~0: nothing kept */
  uint32_t cand_n[2] = {0, 0};     /* windows per strand */
  size_t cand_bidx[2] = {0, 0};    /* words of each strand's bucket index behind them */
  /* seed recipes of the last two (budget, geometry, thresholds) - the CLI's --threshold pass alternates two
   * budgets on one handle: full | a | b per set; rec_cur = the set the last call used */
  gs_recipe_set rec[2];
  uint32_t rec_cur = 0;
  /* PAM pairs whose tables did not fit next to the rest (bit per pair code): not tried again until memory
   * is released (gs_pairtab_forget_nofit) */
  uint32_t pairtab_nofit = 0;
  /* PAM-pair tables (gs_pairtab.hip), built on first use for the pairs a batch's patterns end in */
  gs_pairtab_host pairtab[2];
  bool pairtab_off = false; /* a batch ran out of memory next to them: not built again on this handle */
  bool rot_off = false;     /* the same for the strand tables' rotated copies */
  bool tile_order_off = false; /* a batch had interval records or one sequence at one row twice: the per-guide tile ordering is not tried again for that shape */
  uint64_t tile_order_off_key = 0;
  bool big_long_runs = false; /* a batch showed long runs of one sequence (repeat-rich genome): the device-wide ordering sorts by row, then by word (gs_search.hip) */
};

/* make sure slot `slot` holds the tables of pair `code` at context depth v_rem with rotated copies from
 * step rot_first on (fewer when memory is short: they may take `share` of what is free beyond the
 * reserve); valid stays false when they do not fit */
gs_status gs_pairtab_ensure(gs_index *ix, uint32_t slot, uint32_t v_rem, uint32_t code, uint32_t rot_first, double share,
                            hipStream_t st);
/* add the deep tables (the other strand's side, PAM of three symbols) to a valid slot; p.deep stays false when they do not fit */
gs_status gs_pairtab_ensure_deep(gs_index *ix, uint32_t slot, uint32_t P, uint32_t kb, hipStream_t st);
void gs_pairtab_free(gs_index *ix, uint32_t slot);
/* gs_index.hip: the strand tables' rotated copies, built by the first batch that reads them */
gs_status gs_strand_rot_ensure(gs_index *ix, hipStream_t st);
bool gs_strand_rot_release(gs_index *ix);

/* ---- gs_tileorder.hip: the per-guide ordering of the guides k_order's LDS does not hold ---- */
struct gs_tileorder_in {
  uint32_t n_set;         /* guides of the set */
  const uint32_t *list;   /* device: set guide -> guide of the batch (nullptr: the whole batch) */
  const uint32_t *redo_pos; /* device: guide of the batch -> set guide (with list) */
  const uint32_t *counts; /* device [2 n]: records per item, exact */
  const uint32_t *cls;    /* device [2 n][8]: records per item and mismatch count */
  const uint4 *slots;     /* the main slot array, `cap` per item */
  uint32_t cap;
  const uint4 *arena;     /* overflow chunks and their owners */
  const uint32_t *chunk_item, *chunk_seq;
  uint32_t n_used;
  uint32_t *nhits;        /* device [n]: set to the set guides' hit counts by the plan */
  const uint64_t *offsets; /* device [n + 1]: valid when gs_tileorder_run is called */
  gs_hit *hits;
  uint32_t L, P, m, v_rem;
};
struct gs_tileorder_state {
  uint32_t n_it = 0, n_tiles = 0, n_btiles = 0, n_chunks = 0, n_big = 0, n_deal = 0;
  uint32_t spill_units = 0, spill_cap = 0; /* bucket units kept for the buckets that outgrow their slots; records the spill list holds */
  uint32_t n_excl = 0; /* guides with an item beyond the tiles' reach (ix->w_t_excl lists them): ordered device-wide by the caller */
  uint64_t n_records = 0; /* records of the set (set by gs_tileorder_run) */
};
/* does the sort word (class base + rank of the sequence, then the row) fit 64 bits? */
bool gs_tileorder_fits(uint32_t L, uint32_t P, uint32_t m);
/* plan: tiles, bucket space, chunk index, class starts; sets nhits of the set's guides.  *usable = false: an item is
 * too large for this form (nothing else was changed) */
gs_status gs_tileorder_plan(gs_index *ix, const gs_tileorder_in &in, hipStream_t st, gs_tileorder_state &S, bool *usable);
/* partition + order + write the hits; *violations != 0: an assumption did not hold (multi-row records, a sequence at
 * one row twice, a bucket beyond its space) - the hits of the set are then not valid and the caller orders it the other way */
gs_status gs_tileorder_run(gs_index *ix, const gs_tileorder_in &in, gs_tileorder_state &S, hipStream_t st, uint32_t *violations);

#define GS_HANDLE_LOCK(ix)                               \
  std::unique_lock<std::recursive_mutex> handle_lock__; \
  if (ix) handle_lock__ = std::unique_lock<std::recursive_mutex>((ix)->mtx)

#define GS_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      gs_set_error(std::string(#expr) + ": " + hipGetErrorString(e__));           \
      return GS_ERR_DEVICE;                                                       \
    }                                                                             \
  } while (0)

void gs_set_error(const std::string &s);
/* value of switch `key` on this handle, or nullptr (gs_index::opts) */
const char *gs_opt(const gs_index *ix, const char *key);
void gs_opts_from_env(gs_index *ix);
extern std::atomic<int> gs_debug_any; /* some handle has GS_DEBUG set: the allocator reports large growth */
gs_status gs_reserve(gs_buffer &b, size_t bytes);

/* GPU suffix array: d_text (n bytes incl. sentinel) -> d_sa (n uint32) */
gs_status gs_device_suffix_array(const uint8_t *d_text, uint64_t n, uint32_t *d_sa, hipStream_t st);
/* build the device layout of one strand from device-resident text and SA */
gs_status gs_strand_from_device(const uint8_t *d_text, uint32_t *d_sa_owned, uint64_t n,
                                gs_strand *out, hipStream_t st);
void gs_strand_free(gs_strand *s);
/* gs_verify.hip: rows of a device-resident suffix array that are out of range or repeat a value */
gs_status gs_count_bad_sa_rows(const uint32_t *d_sa, uint64_t n, hipStream_t st, uint64_t *bad);

/* SDSL importer (gs_sdsl_import.cpp): reads <path> into BWT bytes + SA samples */
gs_status gs_sdsl_read(const char *path, std::vector<uint8_t> &bwt, std::vector<uint64_t> &sa_samples,
                       uint64_t C256[257]);

/* gs_suffix.hip: the same suffix array by doubling over the rows whose groups still have company only */
gs_status gs_device_suffix_array_discarding(const uint8_t *d_text, uint64_t n, uint32_t *d_sa, hipStream_t st, bool debug);

/* k_order / k_locate (gs_order.hip): guides one wave takes at a time, a lane each for the ones that need no wave - as
 * many as still leave every wave of the chip a group (the guides of a group that DO need the wave are served one after
 * the other) */
static inline __host__ __device__ uint32_t gs_lane_group(uint32_t n_guides) {
  uint32_t g = 64;
  while (g > 1 && (n_guides + g - 1) / g < 8192u) g >>= 1;
  return g;
}

#endif
