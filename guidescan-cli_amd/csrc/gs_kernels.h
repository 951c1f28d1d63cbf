/*
 * gs_kernels.h -- what the kernels of the enumerate path and the host code that launches them share: launch geometry,
 * argument structs, kernel declarations.  The kernels live in gs_search.hip (k_search, k_prepare), gs_order.hip (order,
 * scan, locate, the overflow arena, the fix-up of shared items), gs_bigorder.hip (the device-wide ordering); the host
 * side is gs_enumerate.hip (the batch pipeline) and gs_recipes.hip (seed recipes).
 */
#pragma once
#include "gs_device.h"

#define SEARCH_WAVES 4 /* waves per workgroup */
#ifndef STACK_ENTRIES
#define STACK_ENTRIES 224 /* 16-byte nodes of the X/G stacks per wave */
#endif
/* verification queue behind the stacks: seeds waiting for their context rows to be read
 * (VQ_CAP descriptors) and the owner markers of one verification pass (128 x uint32) */
#ifndef VQ_CAP
#define VQ_CAP 128
#endif
#ifndef VQ_DRAIN
#define VQ_DRAIN 64 /* verify as soon as this many seeds wait (a seeding step adds at most 64) */
#endif
#define WAVE_LDS_FAST (VQ_CAP + 32 + DTAB + 5) /* 16-byte entries per wave of the table-only variant: 3.9 KiB */
#define WAVE_LDS_ENTRIES (STACK_ENTRIES + WAVE_LDS_FAST) /* the walking variant adds the X/G stacks: 8 KiB */
#define MAX_FANOUT 5      /* children one node can push (A,C,G,T + literal N / 4 PAM copies) */

/* node meta (64 bit):  [63:59] t  [58:56] k  [55] -  [54] fan  [53:52] pam id  [51:0] path */
#define META_T(m) ((uint32_t)((m) >> 59))
#define META_K(m) ((uint32_t)(((m) >> 56) & 7))
#define META_FAN(m) ((uint32_t)(((m) >> 54) & 1))
#define META_PAM(m) ((uint32_t)(((m) >> 52) & 3))
#define PATH_MASK ((1ull << 52) - 1)


struct gs_search_args {
  gs_strand_dev sd[2];
  const gs_guide_rec *guides;
  uint4 *slots;          /* [n_items][cap] match records {key_lo, key_hi, sp, ep} */
  const uint64_t *slot_off; /* optional: item s owns slots [slot_off[s], slot_off[s+1]) instead */
  uint32_t *counts;      /* [n_items] matches found (may exceed cap -> overflow) */
  /* Overflow arena: an item whose matches outgrow its slots continues in chunks of ARENA_CHUNK records
   * taken from one array with an atomic counter (chunk c belongs to item chunk_item[c] and holds its
   * records cap + chunk_seq[c] * ARENA_CHUNK ...), so no item is searched twice; nchunk[item] = {chunks
   * taken, the last one}.  An item is complete when counts <= cap + chunks * ARENA_CHUNK; when the arena
   * runs out the item keeps counting and the host falls back to the exact-size second pass. */
  uint4 *arena;          /* or nullptr */
  uint32_t *arena_next;  /* chunks handed out */
  uint32_t *chunk_item, *chunk_seq;
  uint2 *nchunk;         /* [n_items] */
  uint32_t arena_chunks; /* chunks the arena holds */
  /* with the arena: matches per item and mismatch count, [n_items][8] - what lets the per-guide ordering
   * (gs_tileorder.hip) place an item's records among the other index's without a counting pass */
  uint32_t *cls;
  /* Every loop of an item counts its rounds against max_iter; an item that passes it gives up, raises
   * *err and the wave skips what is left of the queue, so the grid always drains and the call fails with
   * GS_ERR_DEVICE instead of hanging the device (a table damaged in memory, a code-generation fault). */
  uint32_t max_iter;
  uint32_t *err;
  uint32_t *work;        /* work-queue head */
  uint32_t take;         /* items a wave takes per visit to the work counter (>= 1) */
  unsigned long long *stats; /* [0] n_ext, [1] overflow items, [4] two-sided items, [5] one-sided, [8..] request counters */
  uint32_t n_items, L, P, m, cap;
  /* prefix-table seeding (pt_k = 0: walk from the root).  The seeds of an item are listed in
   * RECIPES that do not depend on the guide (gs_build_recipes_*): a recipe is the set of
   * substitutions (consumption step, which of the three other bases) that turns the guide's own
   * k-mer into the seed's, plus the table copy to read it from.  64 bits: [2:0] substitutions n,
   * [5:3] lower bound on the substitutions in X (other strand's seeds), [7:6] 1 = read the rotated
   * copy of step [11:8], then n 7-bit fields 3*step + digit from bit 12.  Lane l of a seeding step
   * takes recipe pos+l: consecutive recipes are laid out so that neighbours share table lines.
   *   rec_full : every depth-k node within m substitutions (one-sided seeding)
   *   rec_a    : this strand's share under two-sided seeding (a < astar(o))
   *   rec_b    : the other strand's share, steps counted as y = guide symbol L-1-y */
  const uint2 *rec_full, *rec_a, *rec_b;
  uint32_t n_rec_full, n_rec_a, n_rec_b;
  /* rec_a for items whose seeds go through PAM-pair tables (8-byte entries: the two-symbol extensions of a
   * variant are one 128-byte block, so the class with one substitution left needs no rotated copy) */
  const uint2 *rec_a8;
  uint32_t n_rec_a8;
  /* PAM-pair tables (gs_pairtab.hip): this strand's seeds of an item whose PAM patterns all end (in
   * consumption order) in one of these pairs of concrete bases are looked up among the rows that
   * have that pair in place - a sixteenth of the genome's rows - instead of all of them */
  gs_pairtab_dev pt[2][2]; /* [slot][strand] */
  uint32_t n_pt;           /* slots in use */
  /* bdeep: every pattern of the batch has a PAM-pair table with a deep table (PAM of three symbols): the
   * other strand's seeds are entries of those - k-2 guide symbols deep, the base under the PAM's N
   * folded in - and X shrinks to the first x_len = L-k+2 guide symbols (else x_len = v_rem) */
  uint32_t bdeep, x_len;
  uint32_t pt_k; /* table depth k; seeds are the depth-k nodes */
  /* context verification: L+P-pt_k (<= 16) symbols remain after the table depth; 0 = disabled */
  uint32_t v_rem;
  uint32_t v_max; /* rows per queued descriptor (<= 1023): larger intervals are verified in pieces */
  uint32_t dbg_skip; /* timing experiments only (GS_DBG_SKIP): 1 = no verification, 2 = no seeds kept */
  /* the counting instantiation tallies distinct aligned blocks of 2^cnt_shift bytes per load instruction: 6 = the
   * 64-byte lines the roofline's bytes are priced on, 7 (GS_COUNT_SHIFT=7) = 128-byte blocks - what the memory
   * system serves as ONE random request (tools/gather_bench: a 128-byte block read by one instruction costs what a
   * 64-byte one does, 4.8 x 10^10 per second at 12-40 GB) */
  uint32_t cnt_shift;
  /* two-sided seeding.  X = the first v_rem consumed guide symbols (only
   * this strand's table covers them), O = the next pt_k - v_rem (both tables), R = the rest of the
   * guide (only the other strand's table, with the PAM).  A site with (a, o, b) substitutions in
   * (X, O, R) is found from THIS strand's table when a < astar(o) and from the OTHER strand's
   * table otherwise; astar holds 4 bits per o (15: this strand takes every a). */
  uint32_t append; /* this pass adds to the matches an earlier pass (other PAM patterns) left in the slots */
  uint32_t bidir, astar;
  /* windows of this strand's text where a literal 'N' lies under the PAM (index.hpp:139-149) and
   * the guide part is plain A,C,G,T: {q lo, q hi, PAM symbols (3 bits each: 0..3, 4 = N), text
   * position of the site}.  The other strand's table cannot see them: its share of them is
   * reported straight from this list. */
  const uint4 *cand[2];
  uint32_t n_cand[2];
  /* an assembly with thousands of N runs: the windows bucketed by each of the first four 5-symbol chunks of their
   * guide part (cand_off[s][1025 c + v] .. [+1] = the places in cand_ids[s] of the windows whose chunk c spells v):
   * within m <= 3 substitutions one of the four chunks is intact, so an item reads the four buckets of its own
   * chunks instead of the whole list; nullptr: the list is scanned in order */
  const uint32_t *cand_off[2], *cand_ids[2];
  /* ---- heavy items shared among waves (table-seeded variants, arena on, one PAM pass) ----------------------
   * One wave owns one item, and on a repeat-rich genome a quarter of the items hold 10^4 .. 10^5 records each: the
   * launch lasted as long as the wave slots that drew two or three of them.  A verification pass (k_search_body::
   * verify) whose queued descriptors cover share_min groups of eight rows or more is not run by the item's wave: the
   * descriptors go to a queue in memory as PACKAGES of at most share_max groups - 64 descriptors, self-contained
   * next to the item number, the side (this strand's table / the other strand's) and the PAM-pair table - and the
   * waves that find the work counter exhausted run them: same code, entered with the queue preloaded and no recipes.
   * A helper's records go to arena chunks of its own (chunk_seq = 0x40000000 | its number among the item's helper
   * chunks, chunk_fill = what it holds), its counts to sh_acc; k_share_scan/dir/fix (below) then close the gaps
   * (records from the item's last chunks into the holes), so everything downstream sees the layout it always saw.
   * Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): packages are whole 128-byte lines written once per
   * launch with write-through (sc1) stores, drained, then an sc1 flag per package; a consumer holds ticket t
   * (one atomicAdd on the head), polls flag t, acquires at agent scope and reads with sc1 loads.  No wave waits for
   * another except a helper for the package of its ticket, whose writer never waits: every wave reaches its exit. */
  uint4 *shq;           /* nullptr: off.  SHQ_PKG uint4 per package: [0] = {item, shared item, side | table << 1 | descriptors << 8, 0} */
  uint32_t *shq_ctl;    /* [0] packages reserved, [32] tickets handed out, [64] waves that left the item phase, [96] shared items */
  uint32_t *shq_ready;  /* per package: written */
  uint32_t *sh_list;    /* shared item -> slot */
  uint32_t *sh_acc;     /* per shared item 16 words: [0] records of helpers, [1] their chunks, [8..15] per mismatch class */
  uint32_t *chunk_fill; /* per chunk of a helper: records it holds */
  uint32_t shq_cap, sh_max, share_min, share_max, n_waves;
  /* every table-seeded instantiation counts the verification passes of share_min row groups or more: what tells the
   * host whether the next batch of this shape is better served by the heavy instantiation */
  uint32_t *hpass;
  /* GS_DEBUG: where the heavy launch's time goes, in ticks of the 100 MHz wall clock (8 x uint64 behind shq_ctl + 104):
   * [0] first wave's start (min), [1] last wave leaving its items (max), [2] last wave's exit (max), [3] sum of the
   * waves' item phases, [4] of their helper episodes, [5] of their waits for a package, [6] episodes */
  uint32_t sh_prof;
  uint32_t helper_only; /* a launch that has no items of its own: its waves run packages until the OTHER launch's n_waves have left their items */
  /* ---- the two-launch form of a batch whose every pattern has its PAM-pair + deep tables (gs_seed.hip) ----
   * desc_a / desc_b: one descriptor per guide (k_describe) in the order each launch takes the guides (by their first / last
   * symbols, or as given: one array); xwork: eight work counters per launch, one per XCD, 128 bytes apart */
  const struct gs_guide_desc *desc_a, *desc_b;
  uint32_t *xwork;
  uint32_t seed_opt; /* bit 0: single-use reads non-temporal; bit 1: seeds without a substitution in X read the plain table */
};
/* what an item of the table-seeded search derives from its guide alone (strand independent): 64 bytes, one scalar load */
struct gs_guide_desc {
  uint32_t q_lo, q_hi; /* gs_guide_rec::q */
  uint32_t pam[4];
  uint32_t meta;       /* [2:0] patterns (0: the guide is not valid), [4:3] PAM-pair table slots its patterns go through, [7:5] pairs the
                          deep tables' masks test, [8 + j] the slot of pattern j's tables, [12 + 4 j +: 4] the bases pattern j's first symbol takes,
                          [28 + s] strand s has a literal-N window within the budget of the guide's symbols */
  uint32_t pidx0;      /* this strand's side: table index of the guide's exact k-mer */
  uint32_t pidxg;      /* the other strand's side: index of the complemented last L - x_len guide symbols */
  uint32_t qrem_b;     /* the complemented first x_len guide symbols, last first */
  uint32_t bsel_z, bsel_w; /* the bit of the guide's pair in each of a deep-table entry's four 16-bit masks */
  uint32_t qhot;       /* the nearest six remaining guide symbols as one-hot nibbles (PAM-pair table filters) */
  uint32_t key_a, key_b; /* scheduling keys: the first min(x_len, 8) symbols as the pair table indexes them; the last min(L - k, 8) as the deep table does */
  uint32_t guide;      /* whose descriptor this is (a launch reads them in its schedule's order) */
};
struct gs_describe_args {
  const gs_guide_rec *guides;
  gs_guide_desc *desc;
  uint32_t *hist; /* [2][65536] or nullptr */
  uint32_t n, L, P, k, x_len, n_pt;
  uint32_t code[2];
  /* the literal-N windows of the batch (gs_search_args::cand ...): a guide none of whose strand's windows lies within m
   * substitutions of its symbols says so in its descriptor (meta bits 28, 29 clear), and its items skip the list */
  const uint4 *cand[2];
  const uint32_t *cand_off[2], *cand_ids[2];
  uint32_t n_cand[2];
  uint32_t m;
};
#define SHQ_PKG 72u /* uint4 per package: header + 64 descriptors, padded to nine 128-byte lines */
#define SH_NONE 0xFFFFFFFFu
#define SH_HELPER_SEQ 0x40000000u
typedef uint32_t __attribute__((address_space(1))) gs_gu32;
typedef unsigned long long __attribute__((address_space(1))) gs_gu64;
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) {
  return __hip_atomic_load((const gs_gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) {
  __hip_atomic_store((gs_gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* 16 bytes as two write-through / L1-bypassing 8-byte accesses */
__device__ __forceinline__ void st16_agent(uint4 *p, const uint4 v) {
  __hip_atomic_store((gs_gu64 *)p, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((gs_gu64 *)p + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint4 ld16_agent(const uint4 *p) {
  const unsigned long long lo = __hip_atomic_load((const gs_gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load((const gs_gu64 *)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
#define DSC_LO 27u  /* descriptor.y bits 29:27: fewest substitutions allowed among the remaining guide symbols */
#define DSC_EXC 30u /* descriptor.y bit 30: the interval holds exception rows (gs_strand_dev::exc_row) */

#define VERIFY_MAX_DEFAULT 1023u
#define DTAB 88u /* per-item substitution table: 4 entries per step of this strand's k-mer (k <= 16) or per
                    guide symbol of the other strand's (k - P <= 21); the two sides seed one after the other */

#ifndef GS_VU
#define GS_VU 4u /* candidate rows per lane whose second-level loads are in flight together (k_search_body::verify) */
#endif
#define SEED_LOW_MAX 128 /* refill the stacks from the prefix table when they hold this few nodes */


/* waves per SIMD each search instantiation is compiled for (the host sizes the persistent grid by them) */
#ifndef GS_WAVES_EU
#define GS_WAVES_EU 5 /* the walking variant: 88 VGPRs */
#endif
#ifndef GS_WAVES_EU_FAST
#define GS_WAVES_EU_FAST 8 /* the table-only variants carry no X/G stack code: <= 64 VGPRs */
#endif
#ifndef GS_WAVES_EU_PD
#define GS_WAVES_EU_PD 8
#endif
#ifndef GS_WAVES_EU_HEAVY
#define GS_WAVES_EU_HEAVY 8
#endif
#ifndef GS_WAVES_EU_SEED
#define GS_WAVES_EU_SEED 8 /* gs_seed.hip */
#endif

/* ---- gs_order.hip ---- */
struct gs_order_args {
  uint4 *slots;           /* in: [n][2][cap] raw ; out: [n][2*cap] ordered unique {key_lo,key_hi,sp,cnt} */
  const uint32_t *counts; /* [2n] */
  uint32_t *nmatch;       /* [n] */
  uint32_t *nhits;        /* [n] */
  unsigned long long *stats; /* [2] total matches */
  uint32_t n, cap;
};
#define ORDER_WAVES 4
#define ORDER_SMALL 128u
#define SCAN_BLOCK 1024
struct gs_locate_args {
  gs_strand_dev sd[2];
  const uint4 *matches; /* [n][2*cap] ordered unique */
  const uint32_t *nmatch;
  const uint64_t *offsets;
  const uint32_t *gmap; /* optional: offsets index of guide g is gmap[g] (redo batch) */
  gs_hit *hits;
  uint64_t genome_length;
  uint32_t n, cap;
  uint32_t v_rem; /* records with key bit 0 set sit v_rem symbols right of the site's start */
};
struct gs_agather_args {
  const uint4 *slots, *arena;
  const uint32_t *counts;                 /* per item of the main pass (exact) */
  const uint32_t *chunk_item, *chunk_seq; /* per chunk */
  const uint32_t *list;                   /* overflow guides */
  const uint32_t *redo_pos;               /* guide -> position in list */
  const uint64_t *dst_off;                /* 2 n_o + 1 offsets, or nullptr: stride cap2 */
  uint4 *dst;
  uint32_t n_o, cap, cap2, n_used;
};
struct gs_share_args {
  const uint32_t *ctl;     /* gs_search_args::shq_ctl */
  const uint32_t *sh_list;
  const uint32_t *sh_acc;
  uint32_t *counts;
  uint2 *nchunk;
  uint32_t *cls;
  const uint32_t *chunk_item;
  uint32_t *chunk_seq;
  const uint32_t *chunk_fill;
  const uint32_t *arena_next;
  uint4 *slots, *arena;
  uint32_t *dbase; /* [sh_max + 1] first directory entry of each shared item */
  uint32_t *dir;   /* the item's chunks in order: the owner's, then the helpers' */
  unsigned long long *stats;
  uint32_t sh_max, cap, arena_chunks;
};
#define SH_MAXSEG 4096u /* slots + chunks of one shared item the fix holds in LDS (4 M records) */
#define SH_SMALLSEG 254u /* ... in its small form (260,000 records) */
__global__ void k_order(gs_order_args a);
__global__ void k_order_wg(gs_order_args a, uint32_t nmax);
__global__ void k_scan_partial(const uint32_t *in, uint64_t *blocksum,
                                                             uint32_t n);
__global__ void k_scan_blocksums(uint64_t *blocksum, uint32_t nb);
__global__ void k_scan_final(const uint32_t *in, const uint64_t *blocksum,
                                                           uint64_t *out, uint32_t n, uint32_t nb);
__global__ void k_locate(gs_locate_args a);
__global__ void k_rank4(gs_strand_dev sd, const uint64_t *rows, uint64_t n, uint64_t *out);
__global__ void k_resolve(gs_strand_dev sd, const uint64_t *rows, uint64_t n, uint64_t *out);
__global__ void k_collect_overflow(const uint32_t *counts, uint32_t n, uint32_t cap, uint32_t *list,
                                   uint32_t *n_list);
__global__ void k_gather_guides(const gs_guide_rec *in, const uint32_t *list, uint32_t n_o,
                                gs_guide_rec *out);
__global__ void k_gather_counts(const uint32_t *counts, const uint32_t *list, uint32_t n_o, uint32_t *out);
__global__ void k_patch_overflow(const uint32_t *list, uint32_t n_o, const uint32_t *nhits2,
                                 uint32_t *nhits);
__global__ void k_arena_gather(gs_agather_args a);
__global__ void k_share_scan(gs_share_args a);
__global__ void k_share_dir(gs_share_args a);
template <uint32_t MAXSEG>
__global__ void k_share_fix(gs_share_args a);
__global__ void k_raw_counts(const uint4 *slots, const uint32_t *counts, uint32_t n, uint32_t cap,
                                                    uint32_t *raw);
__global__ void k_count_stats(const uint32_t *counts, uint32_t n_items, unsigned long long *out);
__global__ void k_need_chunks(const uint32_t *counts, uint32_t n_items, uint32_t cap, uint32_t *out);

/* ---- gs_bigorder.hip ---- */
struct gs_big_src {   /* where the records of one set item live */
  uint64_t off;       /* element offset */
  uint32_t alt;       /* 0: main slot array, 1: the exact-size redo array */
};
struct gs_big2_tab {
  unsigned long long n[32][8]; /* n[a][r] = C(a, r) 3^r: sequences of a positions with r substitutions */
  /* base[mismatches << 1 | index]: the sequences that go before the class's first - every sequence with fewer
   * mismatches on either index, and the class's own on index 0: (mismatches, index, rank) as ONE number, three
   * bits narrower than the three fields side by side (a radix pass less at m = 5 and 6) */
  unsigned long long base[16];
};
struct gs_big2_compact_args {
  const uint4 *slots_main, *slots_alt;
  const gs_big_src *src;
  /* from_arena: the set's records are read where k_search left them - an item's first `cap` records in the
   * main slot array, the rest in its arena chunks - instead of from a contiguous copy */
  const uint4 *arena;
  const uint32_t *chunk_item, *chunk_seq, *counts;
  const uint32_t *list, *redo_pos; /* the set is the overflow list (set guide j = list[j]); nullptr: the whole batch */
  uint32_t cap, n_used, from_arena;
  const unsigned long long *prefix;
  const gs_big2_tab *tab;
  uint4 *recs;
  unsigned long long *W;
  uint32_t *rowkey, *idx;
  unsigned long long pam_mul;
  uint32_t n_items, L, P, rbits;
  /* row_bits != 0: the sort word carries the low row_bits bits of the record's first row (+ row_off) below it:
   * one sort orders the words and - nearly - the rows inside a run of equal words (big_order) */
  uint32_t row_bits;
  uint32_t row_off; /* tests: moves where the rows of a run cross a multiple of 2^row_bits */
};
struct gs_blocate3_args {
  gs_strand_dev sd[2];
  const uint4 *S2;
  const unsigned long long *W;
  const uint32_t *keep;
  const unsigned long long *row_scan;
  const unsigned long long *prefix;
  const uint32_t *gmap;
  const uint64_t *offsets;
  gs_hit *hits;
  uint64_t genome_length, T;
  uint32_t v_rem, gshift;
};
__global__ void k_big_totals(const unsigned long long *prefix, const uint32_t *keep_scan,
                             const unsigned long long *row_scan, uint32_t n_set, uint32_t *nmatch,
                             uint32_t *nhits, uint32_t *err);
__global__ void k_big2_compact(gs_big2_compact_args a);
__global__ void k_big2_counts(const uint32_t *counts, const uint32_t *list, uint32_t n_items, unsigned long long *cnt64);
__global__ void k_big2_gather(const uint4 *recs, const uint32_t *idx, uint64_t T, uint4 *out);
__global__ void k_big2_gather_w(const unsigned long long *W, const uint32_t *idx, uint64_t T, unsigned long long *out);
__global__ void k_big2_runs(const unsigned long long *W, const uint32_t *idx_in, const uint4 *recs,
                                                   uint64_t T, uint32_t short_max, uint32_t *idx_out, uint32_t *long_run);
__global__ void k_max_u64(const unsigned long long *v, uint32_t n, unsigned long long *out);
__global__ void k_big2_comp(const unsigned long long *W, const uint32_t *rowkey, uint64_t T, uint32_t row_bits,
                            uint32_t row_off, unsigned long long *Wc);
__global__ void k_big2_wraps(const uint4 *S2, const unsigned long long *Wc, uint64_t T, uint32_t row_bits, uint32_t *list,
                             uint32_t *n_list);
__global__ void k_big2_fixruns(uint4 *S2, uint4 *tmp, const unsigned long long *Wc, uint64_t T,
                                                      uint32_t row_bits, uint32_t row_off, const uint32_t *list, uint32_t n,
                                                      uint32_t *claimed);
__global__ void k_iota_u32(uint32_t *p, uint64_t n);
__global__ void k_big2_flags(const uint4 *S2, const unsigned long long *W, uint64_t T, uint32_t *keep,
                             unsigned long long *rows, uint32_t wshift);
__global__ void k_big2_locate(gs_blocate3_args a);
__global__ void k_big_sources(const uint32_t *counts_main, const uint32_t *redo_pos, const uint64_t *slot_off2,
                              const uint32_t *counts2, uint32_t n_items, uint32_t cap, gs_big_src *src,
                              unsigned long long *cnt64);
__global__ void k_fill_u32(uint32_t *p, uint32_t v, uint32_t n);
__global__ void k_mark_redo(const uint32_t *list, uint32_t n_o, uint32_t *redo_pos);

/* ---- gs_search.hip ---- */
__global__ void k_search_walk(gs_search_args a);
__global__ void k_search_fast(gs_search_args a);
__global__ void k_search_count(gs_search_args a);
__global__ void k_search_fast_pd(gs_search_args a);
__global__ void k_search_count_pd(gs_search_args a);
__global__ void k_search_heavy(gs_search_args a);
__global__ void k_search_heavy_pd(gs_search_args a);
__global__ void k_search_pub(gs_search_args a);
__global__ void k_search_pub_pd(gs_search_args a);
__global__ void k_prepare(gs_prep_args a);

/* ---- gs_seed.hip ---- */
__global__ void k_seed_b(gs_search_args a);
__global__ void k_seed_a(gs_search_args a);
__global__ void k_seed_count_b(gs_search_args a);
__global__ void k_seed_count_a(gs_search_args a);
/* descriptors (+ schedules when `sorted`) of the guides sa.guides[0 .. ng); *out = sa with desc / sched_* / xwork set */
gs_status gs_seed_describe(gs_index *ix, const gs_search_args &sa, uint32_t ng, bool sorted, hipStream_t st, gs_search_args *out);
gs_status gs_seed_launch(const gs_search_args &sa, uint32_t grid, bool count_req, hipStream_t st);
/* n_heavy[0]: guides of the batch whose own k-mer heads an interval of `thresh` rows or more in a strand table, [1]: the
 * largest such interval (d_out: 8 bytes of scratch) */
gs_status gs_estimate_heavy(gs_index *ix, const gs_guide_rec *guides, uint32_t n, uint32_t thresh, uint32_t *d_out, hipStream_t st,
                            uint32_t n_heavy[2]);

/* ---- gs_recipes.hip (host) ---- */
void gs_choose_astar(uint32_t m, uint32_t nX, uint32_t nO, uint32_t nR, double epam, uint32_t astar[8], double verify_a = 1.5,
                     double verify_b = 1.9);
gs_status gs_recipes_for(gs_index *ix, uint32_t L, uint32_t P, uint32_t m, uint32_t v_rem, const uint32_t *astar, bool deep,
                         hipStream_t st);
