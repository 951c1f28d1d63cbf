/*
 * gs_search.hip -- the enumerate hot path as hand-written CDNA4 (gfx950) kernels.
 *
 * Pipeline per batch (all on one stream, device resident):
 *   k_prepare   ASCII guides/PAMs -> packed query records        (process.hpp:51-63)
 *   k_search    one wavefront per (guide, strand): bounded-Hamming backward search,
 *               DFS stack of SA intervals in LDS, ballot/prefix compaction of live
 *               branches                                          (index.hpp:182-248, 125-170)
 *   k_order     per guide: canonical order + dedupe of matches   (process.hpp:21-23,
 *                                                                  structures.hpp:40-42)
 *   k_scan*     hits-per-guide -> CSR offsets
 *   k_locate    SA gather + coordinate rule                       (process.hpp:100-115,
 *                                                                  csa_wt.hpp:333-346)
 * No MFMA: integer rank/popcount work bound by random 64-byte HBM reads.
 */
#include "gs_device.h"

#include <rocprim/rocprim.hpp>

#include <cmath>

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

/* ---- search: one wavefront per (guide, strand) ----------------------------
 * Two LDS stacks per wave share one 3.5 KiB array (STACK_ENTRIES nodes): X (grows up) holds "single-symbol" nodes -
 * mismatch budget spent (index.hpp:230 returns before the substitution loop) or a fixed PAM
 * base - which need Occ of one base and have at most one child; G (grows down) holds nodes
 * that still branch (k < m), PAM 'N' wildcards and PAM fan-out nodes.  ~89 % of all nodes are
 * X nodes, and an X iteration costs ~1/4 of the instructions of a G one.
 * CNT: count the distinct 64-byte lines every load instruction asks for (bench.py's algorithmic
 * bytes of THIS algorithm); the timed kernel is the CNT = false instantiation. */
#ifndef GS_WAVES_EU
#define GS_WAVES_EU 5 /* the walking variant: 88 VGPRs */
#endif
#ifndef GS_WAVES_EU_FAST
#define GS_WAVES_EU_FAST 8 /* the table-only variants carry no X/G stack code: <= 64 VGPRs */
#endif
/* WALK: the Occ walk (X/G stacks, G fan-out) is compiled in - the reference-order walk from the root
 * and inputs whose remainder does not fit ctx[].  The table-only variant (every interval resolved
 * against the context arrays) needs neither the 3.5 KiB stack array per wave nor that code. */
/* SPEC: the batch's every PAM pattern (three symbols) has its PAM-pair table and deep table, so every item
 * seeds this strand's side through a pair table and the other strand's through a deep table, none is
 * one-sided: the strand tables' side of the seeding (pair masks, rotated copies, PAM expansion) is compiled
 * out together with the wave-uniform state it keeps alive. */
/* Path codes of up to 16 symbols at once (2-bit fields of T = text, Q = query): 0 where they agree, else the
 * text base's place among the three other bases, A<C<G<T, counted from 1 (what the walk writes per
 * substitution, index.hpp:230-247) = T + [T < Q]; no field carries into its neighbour (T < Q <= 3). */
__device__ __forceinline__ uint32_t path_codes16(const uint32_t T, const uint32_t Q) {
  const uint32_t x = T ^ Q;
  const uint32_t ne = (x | (x >> 1)) & 0x55555555u;
  const uint32_t nq = ~T & Q; /* text bit 0 under query bit 1 */
  const uint32_t lt = ((nq >> 1) | ((~x >> 1) & nq)) & 0x55555555u; /* high bits decide, else the low ones */
  return (T + lt) & (ne * 3u);
}
/* the 2-bit fields of x in reverse order (field 0 <-> field 15) */
__device__ __forceinline__ uint32_t rev_fields16(const uint32_t x) {
  const uint32_t r = __brev(x);
  return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
/* a wave-uniform pointer as a scalar register pair of its own: values that arrive together in one wide
 * kernel-argument load otherwise stay one 8- or 16-register tuple, which the register allocator spills and
 * reloads whole (16 v_readlane for one pointer in the seeding loops) */
template <typename T>
__device__ __forceinline__ const T *own_sgprs(const T *p) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  /* (a pointer rebuilt from integers is a flat one to the compiler: say that it is global memory) */
  return (const T *)(const T __attribute__((address_space(1))) *)(((uint64_t)hi << 32) | lo);
}

template <bool CNT, bool WALK, bool SPEC = false, bool HEAVY = false>
__device__ __forceinline__ void k_search_body(const gs_search_args &a, uint4 *stk) {
  constexpr uint32_t STK = WALK ? STACK_ENTRIES : 0u;
  /* Where a path lives in the 64-bit word the search carries next to a seed or a hit.  The walking variant packs
   * the node's step, PAM pattern and fan-out flag above it (node meta, top of the file): 52 path bits, key = path << 8.
   * The table-only variants carry nothing but the mismatch count: 59 path bits - 2L + 3P <= 59 covers 23-mers with
   * a four-symbol PAM (Cas12a) - in the SAME key layout (position 0 at key bits 59:58, bit 0 = the record's
   * row is v_rem symbols into the site): a path of at most 52 bits gives the key it always gave. */
#ifndef GS_X_PB
#define GS_X_PB 7u
#endif
  constexpr uint32_t PB = WALK ? 0u : GS_X_PB;           /* path bias: field shifts are those of the 52-bit layout + PB */
  constexpr uint32_t KSH = (WALK || GS_X_PB == 0u) ? 56u : 61u;        /* mismatch count above the path */
  constexpr uint32_t PSG = 50u + PB, PSP = 49u + PB; /* guide symbol t at PSG - 2t, PAM symbol u at PSP - 2L - 3u */
  constexpr uint64_t PMASK = (1ull << (52u + PB)) - 1ull;
  const uint32_t lane = lane_id();
  unsigned long long n_ext = 0, n_ovf = 0;
  uint32_t n_fail = 0; /* items that needed more overflow chunks than the arena had left */
  bool bailed = false; /* an item of this wave passed the iteration bound */
  uint32_t n_hpass = 0; /* verification passes of at least share_min row groups */
  uint32_t n_two = 0, n_fb = 0, n_pair = 0; /* items seeded from both strands / one-sided although two-sided seeding is on */
  /* request counters (CNT): table lines, ctx16 lines, ctx words, SA/ISA gathers of the search, Occ lines */
  uint32_t c_tab = 0, c_c16 = 0, c_ctx = 0, c_isa = 0, c_occ = 0, c_rec = 0;
  /* distinct 64-byte lines one load instruction asks for: lanes whose line differs from the
   * previous active lane's (the access patterns here are runs of neighbouring lanes) */
  auto count_lines = [&](uint32_t &acc, bool act, const void *p) __attribute__((always_inline)) {
    if constexpr (CNT) {
      const uint32_t line = (uint32_t)((uintptr_t)p >> a.cnt_shift);
      const uint32_t prev = (uint32_t)__shfl_up((int)line, 1);
      const int pact = __shfl_up((int)act, 1);
      const bool fresh = act && (lane == 0u || !pact || prev != line);
      acc += (uint32_t)__popcll(__ballot(fresh));
    }
  };
  const uint32_t L = a.L, P = a.P, m = a.m;
  const uint32_t T_end = L + P;
  const uint32_t reserve = (MAX_FANOUT - 1) * (T_end + 2);
  const uint32_t limit = STACK_ENTRIES - reserve;
  /* a seeding step pushes at most 64 nodes: it runs only while that keeps the stacks within
   * `limit`, so the single-pop DFS of the G iterations always finds its reserve */
  const uint32_t seed_low = limit > WAVE ? (limit - WAVE < SEED_LOW_MAX ? limit - WAVE : SEED_LOW_MAX) : 0u;
  uint4 *vq = stk + STK;                       /* queued seed descriptors */
  uint2 *own2 = (uint2 *)(vq + VQ_CAP);        /* owner markers of a pass, two per lane */
  uint32_t *own = (uint32_t *)own2;
  uint4 *dtab = vq + VQ_CAP + 32;              /* substitution table of the item: {index xor, path lo, path hi, -} */
  /* overflow chunks of the item: {taken, the last one, the one before}; [3] its number among the shared items;
   * [4..11] matches per mismatch count; [12..14] a helper episode's package: {side | table << 1 | descriptors << 8, the
   * shared item, the package's place in the queue} - wave-uniform state that is read at a handful of places lives
   * here and not in scalar registers, which the seeding loops are short of */
  uint32_t *wmisc = (uint32_t *)(dtab + DTAB);
  /* [16..18] the wave's reserve of arena chunks {next, end, chunks the next visit to the counter takes}: one atomic on
   * one word serves ~88 waves per microsecond chip-wide, and a repeat-rich batch of 20,000 guides takes 480,000 chunks -
   * 5.5 ms of a 9.6 ms launch if every chunk were a visit.  A wave takes 1, 2, 4 ... 16 chunks per visit; what it leaves
   * unused stays marked empty (chunk_seq = 0xFFFFFFFF, written before the launch). */
  if (lane == 0) {
    wmisc[16] = 0u;
    wmisc[17] = 0u;
    wmisc[18] = 1u;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

  /* Items are taken from the work counter `take` at a time: one atomic on one word serves about 88 waves per
   * microsecond chip-wide (MI355X_MICROARCH.md, dequeue), so a counter bumped once per item held a launch of
   * 2 M items at 22.9 ms whatever the items did (measured with the seeding switched off: 22.9 of 26.5 ms) */
  uint32_t item_next = 0, item_end = 0;
  const bool sharing = HEAVY && a.shq != nullptr;
#ifdef GS_SH_PROFILE
  unsigned long long *const prof = (HEAVY && a.sh_prof) ? (unsigned long long *)(a.shq_ctl + 104) : nullptr;
#else
  unsigned long long *const prof = nullptr; /* (a build with -DGS_SH_PROFILE times the phases: tools/ab_share_variants.sh) */
#endif
  unsigned long long t_prev = 0, t_help = 0, t_wait = 0, n_epi = 0;
  if (prof != nullptr) {
    t_prev = wall_clock64();
    if (lane == 0) atomicMin(&prof[0], t_prev);
  }
  bool items_done = false; /* the work counter is exhausted: this wave runs packages of shared items until none is left */
  for (;;) {
    if (!items_done && item_next == item_end) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(a.work, a.take);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base >= a.n_items) {
        if (!sharing) break; /* exit condition every wave reaches */
        items_done = true;
        if (lane == 0) atomicAdd(&a.shq_ctl[64], 1u); /* this wave reserves no package any more */
        if (prof != nullptr) {
          const unsigned long long t = wall_clock64();
          if (lane == 0) {
            atomicMax(&prof[1], t);
            atomicAdd(&prof[3], t - t_prev);
          }
          t_prev = t;
        }
      } else {
        item_next = base;
        item_end = base + a.take < a.n_items ? base + a.take : a.n_items;
      }
    }
    /* a helper episode: one package = one verification pass of somebody else's item */
    bool helper = false;
    uint32_t h_item = 0;
    if (items_done) {
      /* Ticket t: package t is this wave's, if it is ever reserved.  The wave waits for its flag; once every wave has
       * left its items no reservation can follow, and a ticket at or beyond the reserved count leaves.  The writer of
       * a reserved package never waits for anything, and the spin is bounded all the same (GS_ERR_DEVICE, no hang). */
      uint32_t got = SH_NONE;
      if (lane == 0) {
        const uint32_t t = atomicAdd(&a.shq_ctl[32], 1u);
        if (t < a.shq_cap) {
          for (uint32_t spins = 0;; ++spins) {
            if (ld_agent(a.shq_ready + t) != 0u) {
              got = t;
              break;
            }
            if ((spins & 3u) == 3u && ld_agent(&a.shq_ctl[64]) >= a.n_waves && t >= ld_agent(&a.shq_ctl[0])) break;
            if (spins > (1u << 20)) { /* ~3 s of sleeping: something is broken; fail the call, drain the grid */
              atomicOr(a.err, 2u);
              break;
            }
            __builtin_amdgcn_s_sleep(100);
          }
        }
      }
      got = __builtin_amdgcn_readfirstlane(got);
      if (prof != nullptr) { /* (t_prev: the end of the item phase or of the last episode) */
        const unsigned long long t = wall_clock64();
        t_wait += t - t_prev;
        t_prev = t;
      }
      if (got == SH_NONE) break; /* exit condition every wave reaches: all items taken, no package left for this ticket */
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const uint4 hd = ld16_agent(a.shq + (size_t)got * SHQ_PKG);
      h_item = __builtin_amdgcn_readfirstlane(hd.x);
      const uint32_t hz = __builtin_amdgcn_readfirstlane(hd.z);
      if ((hz >> 8) == 0u || (hz >> 8) > WAVE || h_item >= a.n_items) continue; /* a filler for a reservation the queue had no room for */
      if (lane == 0) {
        wmisc[12] = hz;
        wmisc[13] = hd.y;
        wmisc[14] = got;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      helper = true;
    }
    const uint32_t item = helper ? h_item : item_next++;
    /* all forward-index items first, then all reverse-index items: at any moment the waves
     * touch one strand's Occ array, which halves the hot footprint (TLB reach) */
    const uint32_t n_guides = a.n_items >> 1;
    const uint32_t strand = item >= n_guides ? 1u : 0u;
    const uint32_t guide = item - strand * n_guides;
    const uint32_t slot = 2u * guide + strand;
    /* the guide record is wave-uniform: keep every field in scalar registers */
    const uint32_t *gp = (const uint32_t *)(a.guides + guide);
    const uint32_t gw0 = __builtin_amdgcn_readfirstlane(gp[0]);
    const uint32_t gw1 = __builtin_amdgcn_readfirstlane(gp[1]);
    const uint64_t gr_q = ((uint64_t)gw1 << 32) | gw0;
    const uint32_t gr_pam0 = __builtin_amdgcn_readfirstlane(gp[2]);
    const uint32_t gr_pam1 = __builtin_amdgcn_readfirstlane(gp[3]);
    const uint32_t gr_pam2 = __builtin_amdgcn_readfirstlane(gp[4]);
    const uint32_t gr_pam3 = __builtin_amdgcn_readfirstlane(gp[5]);
    const uint32_t gr_npams = __builtin_amdgcn_readfirstlane(gp[6]);
    const uint32_t gr_valid = __builtin_amdgcn_readfirstlane(gp[7]);
    if (!gr_valid || bailed) {
      if (!helper) {
        if (lane == 0) a.counts[slot] = 0;
        if (a.arena != nullptr && lane < 8u) a.cls[(size_t)slot * 8u + lane] = 0u;
      }
      continue;
    }
    uint32_t guard_left = a.max_iter; /* rounds this item's loops may still take (every outer step runs a counted inner loop) */
    const gs_strand_dev &sd = a.sd[strand];
    const uint4 *__restrict__ blocks = sd.blocks;
    const uint32_t npams = P ? gr_npams : 1u;
    const bool fanning = P > 0 && npams > 1u; /* a finished 20-mer fans out per PAM pattern */
    uint4 *out = a.slots + (a.slot_off ? (size_t)a.slot_off[slot] : (size_t)slot * a.cap);
    /* (a helper owns no slots: its records go to arena chunks of its own from the first one on) */
    const uint32_t item_cap = helper ? 0u : a.slot_off ? (uint32_t)(a.slot_off[slot + 1] - a.slot_off[slot]) : a.cap;
    uint32_t n_match = 0;
    if (a.append) n_match = __builtin_amdgcn_readfirstlane(a.counts[slot]);
    if (a.arena != nullptr) {
      uint2 nc = make_uint2(0u, 0u);
      if (a.append) nc = a.nchunk[slot];
      if (lane < 4u) wmisc[lane] = lane == 0u ? nc.x : lane == 1u ? nc.y : lane == 2u ? 0u : SH_NONE;
      if (lane >= 4u && lane < 12u) wmisc[lane] = a.append ? a.cls[(size_t)slot * 8u + (lane - 4u)] : 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    uint32_t xs = 0, gs = 0; /* sizes of the X and G stacks */
    /* what this strand's seeds are looked up in and verified against: the strand's own table and
     * context arrays, or (set per item, below) a PAM-pair table and its rows */
    const uint4 *atab = sd.ptab, *arot = sd.ptab_rot;
    const uint2 *atab8 = nullptr, *arot8 = nullptr; /* a PAM-pair table's 8-byte entries */
    uint32_t arot_first = sd.rot_first;
    const uint16_t *a16 = sd.ctx16;
    const uint32_t *actx = sd.ctx, *arow = nullptr;

    /* is a node at step t2 with k2 mismatches (PAM pattern pamid) a single-symbol node? */
    auto is_single = [&](uint32_t t2, uint32_t k2, uint32_t pamid) __attribute__((always_inline)) -> bool {
      if (t2 < L) return k2 == m;
      if (t2 == L && fanning) return false;
      const uint32_t pw =
          pamid == 0 ? gr_pam0 : pamid == 1 ? gr_pam1 : pamid == 2 ? gr_pam2 : gr_pam3;
      return ((pw >> (3u * (t2 - L))) & 7u) < 4u;
    };
    /* route a live child: emit (terminal), push on X or on G */
    auto route = [&](bool live, bool term, bool single, uint32_t csp, uint32_t cep, uint64_t cmeta,
                     uint32_t vflag = 0u) __attribute__((always_inline)) {
      const bool em = live && term;
      if constexpr (WALK) {
        const bool px = live && !term && single;
        const bool pg = live && !term && !single;
        const uint64_t bx = __ballot(px);
        if (bx) {
          if (px) stk[xs + lanes_below(bx)] = make_uint4(csp, cep, (uint32_t)cmeta, (uint32_t)(cmeta >> 32));
          xs += __popcll(bx);
        }
        const uint64_t bg = __ballot(pg);
        if (bg) {
          if (pg)
            stk[STACK_ENTRIES - 1u - (gs + lanes_below(bg))] =
                make_uint4(csp, cep, (uint32_t)cmeta, (uint32_t)(cmeta >> 32));
          gs += __popcll(bg);
        }
      } else {
        (void)single;
      }
      const uint64_t be = __ballot(em);
      if (be) {
        const uint32_t hi = n_match + (uint32_t)__popcll(be); /* one past the last record of this emission */
#ifndef GS_X_NO_CLS
        if (a.arena != nullptr) {
          /* matches per mismatch count: lane d adds this emission's share of class d (one LDS add, distinct words) */
          const uint32_t kk = (uint32_t)((cmeta >> KSH) & 7ull);
          uint32_t add = 0;
          for (uint32_t d = 0; d <= m; ++d) {
            const uint32_t c = (uint32_t)__popcll(__ballot(em && kk == d));
            add = lane == d ? c : add;
          }
          if (add) atomicAdd(&wmisc[4u + lane], add);
        }
#endif
        if (hi > item_cap && a.arena != nullptr) {
          /* the emission reaches beyond the item's slots: take overflow chunks up to its last record
           * (wave-uniform; at most two per emission, almost always none) */
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          uint32_t nch = __builtin_amdgcn_readfirstlane(wmisc[0]);
          const uint32_t need = (hi - item_cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT;
          while (nch < need) {
            uint32_t id = 0;
            if (lane == 0) {
              uint32_t rn = wmisc[16];
              if (rn == wmisc[17]) { /* the reserve is used up: the next one, twice as large (up to 16 chunks) */
                const uint32_t g = wmisc[18];
                rn = atomicAdd(a.arena_next, g);
                wmisc[17] = rn + g;
                wmisc[18] = g < 16u ? 2u * g : 16u;
              }
              id = rn;
              wmisc[16] = rn + 1u;
            }
            id = __builtin_amdgcn_readfirstlane(id);
            if (id >= a.arena_chunks) break; /* arena exhausted: the item goes on counting only */
            if (lane == 0) {
              a.chunk_item[id] = slot;
              if (helper) { /* numbered among the item's helper chunks; what it holds is said when the episode ends */
                a.chunk_seq[id] = SH_HELPER_SEQ | atomicAdd(&a.sh_acc[16u * wmisc[13] + 1u], 1u);
                a.chunk_fill[id] = ARENA_CHUNK;
              } else {
                a.chunk_seq[id] = nch;
              }
              wmisc[2] = wmisc[1];
              wmisc[1] = id;
            }
            nch++;
          }
          if (lane == 0) wmisc[0] = nch;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        if (em) {
          const uint32_t idx = n_match + lanes_below(be);
          /* bit 0: the record is a single row at the table depth whose text position still
           * has to move left by v_rem symbols (k_locate) */
          const uint64_t key = ((uint64_t)((cmeta >> KSH) & 7ull) << 61) | ((uint64_t)strand << 60) |
                               ((cmeta & PMASK) << (8u - PB)) | vflag;
          const uint4 rec = make_uint4((uint32_t)key, (uint32_t)(key >> 32), csp, cep);
          if (idx < item_cap) {
            out[idx] = rec;
          } else if (a.arena != nullptr) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t e = idx - item_cap, sq = e >> ARENA_SHIFT;
            const uint32_t nch = wmisc[0];
            if (sq < nch) {
              const uint32_t id = sq + 1u == nch ? wmisc[1] : wmisc[2];
              a.arena[((size_t)id << ARENA_SHIFT) | (e & (ARENA_CHUNK - 1u))] = rec;
            }
          }
        }
        n_match = hi;
      }
    };


    /* ---- context verification (both seeding directions) --------------------------------
     * Every lane may bring one interval piece [vsp, vsp+vcnt) at the table depth with kk
     * substitutions so far and its path in cmeta.  The v_rem symbols left of each suffix are in
     * ctx[row], nearest first, i.e. in consumption order.
     *   modeB == false (this strand's table): compare the remaining guide symbols under the
     *     remaining budget, then each PAM pattern exactly ('N' = any base, or a literal 'N' of the
     *     text on an exception row).  A hit is the row itself; its text position is SA[row] -
     *     v_rem (key bit 0, k_locate).
     *   modeB == true (other strand's table, PAM and the rest of the guide already consumed):
     *     the remaining symbols are the complemented first v_rem guide symbols, last first;
     *     a row counts when it has at least `lo` substitutions there (the descriptor's lower
     *     bound: fewer belong to this strand's own seeds) and fits the budget.  The hit is
     *     reported as the row of THIS strand's suffix array that starts at the same site (SA of
     *     the other strand -> position -> ISA of this strand), so records look exactly like the
     *     ones the walk produces.
     * Two levels.  ctx16[row] holds the nearest 8 of those symbols in 16 bits: rows are handed
     * out in groups of eight consecutive rows of one seed; lane l of a pass takes groups 2l and
     * 2l+1, finds the owner seed of each (seeds mark their first group, a running max spreads
     * the marks) and reads its eight words with one 16-byte load, so consecutive lanes read
     * consecutive 16-byte pieces (coalesced), the owner lookup is paid once per eight rows, and
     * an interval of the mean size (11.5 rows at hg38 size) lies in 1.3 cache lines instead of
     * the 1.7 of 32-bit words.  The few rows whose visible guide symbols fit the budget are
     * then decided from the full word ctx[row] - and, when the table entry says the interval has
     * exception rows (a symbol outside A,C,G,T within 16 symbols to the left), from the row's
     * entry in the exception list, which holds the true symbols.  Both arrays are padded by one group. */
    uint32_t qrem_b = 0; /* the other strand's side: the complemented first x_len guide symbols, last first */
    auto verify = [&](const bool modeB, const uint32_t take, uint4 *dsrc) __attribute__((always_inline)) {
      const uint32_t k = a.pt_k;
      const gs_strand_dev &sv = modeB ? a.sd[strand ^ 1u] : sd;
      /* the context arrays the rows live in: the other strand's, this strand's, or the rows of a PAM-pair table */
      const uint16_t *const v16 = modeB ? own_sgprs(sv.ctx16) : a16;
      const uint32_t *const vctx = modeB ? own_sgprs(sv.ctx) : actx;
      /* lane l < take brings seed descriptor dsrc[l] = {first row, mismatches so far << 14 |
       * rows << 17 | lower bound << 27 | exceptions << 30, path lo, path hi}; the first group of
       * each seed is added to .y here */
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint4 mine = make_uint4(0u, 0u, 0u, 0u);
      if (lane < take) mine = dsrc[lane];
      const uint32_t vcnt = (mine.y >> 17) & 0x3FFu;
      const uint32_t vgrp = (vcnt + 7u) >> 3;
      const uint32_t incl = wave_incl_sum(vgrp);
      const uint32_t R = __builtin_amdgcn_readlane(incl, WAVE - 1); /* groups of this step */
      if (!R) return;
      const uint32_t excl = incl - vgrp;
      if (!WALK && R >= a.share_min) n_hpass++;
      if constexpr (HEAVY) {
        /* ---- a heavy pass is handed to the waves that have run out of items (gs_search_args::shq) ---- */
        if (sharing && !helper && R >= a.share_min) {
          uint32_t sid = __builtin_amdgcn_readfirstlane(wmisc[3]);
          if (sid == SH_NONE) {
            uint32_t s = 0;
            if (lane == 0) s = atomicAdd(&a.shq_ctl[96], 1u);
            sid = __builtin_amdgcn_readfirstlane(s);
            if (lane == 0) {
              wmisc[3] = sid;
              if (sid < a.sh_max) a.sh_list[sid] = slot;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          }
          if (sid < a.sh_max) {
            /* packages of at most share_max groups: consecutive descriptors (each <= 128 groups <= share_max) */
            const uint32_t pid = lane < take ? excl / a.share_max : 0u;
            const uint32_t np = __builtin_amdgcn_readlane(pid, (int)(take - 1u)) + 1u;
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&a.shq_ctl[0], np);
            qb = __builtin_amdgcn_readfirstlane(qb);
            if (qb + np <= a.shq_cap) {
              const uint32_t ppid = dpp_or_zero<0x138>(pid + 1u); /* the lane below's package + 1 (lane 0: 0) */
              const bool first = lane < take && ppid != pid + 1u;
              const uint64_t bm = __ballot(first);
              const uint64_t upto = (2ull << lane) - 1ull; /* lanes 0 .. lane (lane 63: all) */
              const uint32_t start = 63u - (uint32_t)__builtin_clzll((bm & upto) | 1ull);
              const uint64_t above = bm & ~upto;
              const uint32_t nxt = above ? (uint32_t)__builtin_ctzll(above) : take;
              uint4 *pk = a.shq + (size_t)(qb + pid) * SHQ_PKG;
              if (lane < take) st16_agent(pk + 1u + (lane - start), mine);
              /* the PAM-pair table this strand's seeds are going through (3: the strand's own table) */
              const uint32_t cur_tab = (modeB || arow == nullptr) ? 3u : arow == a.pt[0][strand].rowid ? 0u : 1u;
              if (first) st16_agent(pk, make_uint4(item, sid, (modeB ? 1u : 0u) | (cur_tab << 1) | ((nxt - start) << 8), 0u));
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* every store of the packages has left before their flags do */
              if (first) st_agent(a.shq_ready + qb + pid, 1u);
              return;
            }
            /* no room in the queue (a later batch gets a larger one): empty packages for what was reserved, and the
             * pass runs here */
            if (qb < a.shq_cap && lane < a.shq_cap - qb && lane < np) {
              st16_agent(a.shq + (size_t)(qb + lane) * SHQ_PKG, make_uint4(0u, 0u, 0u, 0u));
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              st_agent(a.shq_ready + qb + lane, 1u);
            }
          }
        }
      }
      /* descriptor.y = first group (14 bits) | mismatches so far << 14 | rows << 17 | ... */
      if (lane < take) dsrc[lane].y = mine.y | excl;
      const uint32_t g = modeB ? a.x_len : L - k; /* guide symbols among the remaining ones */
      const uint32_t gmask = g >= 16u ? 0xFFFFFFFFu : ((1u << (2u * g)) - 1u);
      const uint32_t qrem = modeB ? qrem_b : ((uint32_t)(gr_q >> (2u * k)) & gmask);
      /* first level: the 16-bit words see the nearest g8 <= 8 guide symbols; two rows per dword */
      const uint32_t g8 = g < 8u ? g : 8u;
      const uint32_t gm8 = (1u << (2u * g8)) - 1u;
      const uint32_t q2x = (qrem & gm8) * 0x00010001u, gm2x = gm8 * 0x00010001u;
      for (uint32_t base = 0; base < R; base += 2u * WAVE) {
        own2[lane] = make_uint2(0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (vgrp) {
          if (excl >= base && excl < base + 2u * WAVE) own[excl - base] = lane + 1u;
          if (excl < base && excl + vgrp > base) own[0] = lane + 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint2 mk = own2[lane];
        uint32_t o0 = mk.x, o1 = mk.y > o0 ? mk.y : o0;
        const uint32_t run = wave_incl_max(o1); /* inclusive max-scan over lanes, then shift to exclusive */
        const uint32_t prev = dpp_or_zero<0x138>(run); /* wave_shr:1, lane 0 reads 0 */
        o0 = o0 > prev ? o0 : prev;
        o1 = o1 > prev ? o1 : prev;
        const uint32_t ow[2] = {o0, o1};
        uint4 wq[2], dsc[2];
        uint32_t row0[2], nrow[2], kkv[2], okm[2];
#pragma unroll
        for (uint32_t jj = 0; jj < 2u; ++jj) { /* two independent 16-byte loads in flight */
          const uint32_t grp = base + 2u * lane + jj;
          const bool on = grp < R;
          dsc[jj] = make_uint4(0u, 0u, 0u, 0u);
          if (on) dsc[jj] = dsrc[ow[jj] - 1u];
          const uint32_t r0 = (grp - (dsc[jj].y & 0x3FFFu)) << 3; /* first row of the group in its seed */
          const uint32_t cnt = (dsc[jj].y >> 17) & 0x3FFu;
          row0[jj] = dsc[jj].x + r0;
          nrow[jj] = on ? (cnt - r0 < 8u ? cnt - r0 : 8u) : 0u;
          kkv[jj] = (dsc[jj].y >> 14) & 7u;
          /* bit c set: c substitutions among the visible symbols are acceptable (lower bound ..
           * budget left).  The first level sees g8 of the g symbols: at least lo - (g - g8) of the
           * lo required ones show there. */
          const uint32_t lo = (dsc[jj].y >> DSC_LO) & 7u;
          const uint32_t lo8 = lo > g - g8 ? lo - (g - g8) : 0u;
          okm[jj] = on ? (((2u << (m - kkv[jj])) - 1u) & ~((1u << lo8) - 1u)) : 0u;
          wq[jj] = make_uint4(0u, 0u, 0u, 0u);
          if (on) wq[jj] = load16_a2(v16 + row0[jj]);
        }
        if constexpr (CNT) {
          /* distinct lines of the pass: the lanes' groups in order (2l, 2l+1), a group counted when
           * its first or its last byte lies in a line the group before it did not reach */
          const bool on0 = nrow[0] != 0u, on1 = nrow[1] != 0u;
          const uint32_t f0 = (uint32_t)((uintptr_t)(v16 + row0[0]) >> 6), l0 = (uint32_t)(((uintptr_t)(v16 + row0[0]) + 15u) >> 6);
          const uint32_t f1 = (uint32_t)((uintptr_t)(v16 + row0[1]) >> 6), l1 = (uint32_t)(((uintptr_t)(v16 + row0[1]) + 15u) >> 6);
          const uint32_t mylast = on1 ? l1 : l0;
          const uint32_t prev = (uint32_t)__shfl_up((int)mylast, 1);
          const int pact = __shfl_up((int)(on0 || on1), 1);
          uint32_t add = 0;
          if (on0) add += ((lane == 0u || !pact || prev != f0) ? 1u : 0u) + (l0 != f0 ? 1u : 0u);
          if (on1) add += ((!on0 || l0 != f1) ? 1u : 0u) + (l1 != f1 ? 1u : 0u);
          for (int o = 32; o > 0; o >>= 1) add += (uint32_t)__shfl_xor((int)add, o);
          c_c16 += add;
        }
        uint32_t cm = 0u; /* candidate rows of this lane: bit 8*jj + r */
#pragma unroll
        for (uint32_t jj = 0; jj < 2u; ++jj) {
          const uint32_t wv[4] = {wq[jj].x, wq[jj].y, wq[jj].z, wq[jj].w};
#pragma unroll
          for (uint32_t h = 0; h < 4u; ++h) {
            const uint32_t x = (wv[h] ^ q2x) & gm2x;
            const uint32_t y = (x | (x >> 1)) & 0x55555555u;
            const uint32_t m0 = __popc(y & 0xFFFFu), m1 = __popc(y >> 16);
            cm |= (((okm[jj] >> m0) & 1u) | (((okm[jj] >> m1) & 1u) << 1)) << (8u * jj + 2u * h);
          }
        }
        cm &= ((1u << nrow[0]) - 1u) | (((1u << nrow[1]) - 1u) << 8); /* rows that exist */
        /* (the heavy instantiation takes the form with several rows in flight when some lane has two candidates or more,
         * else the plain one: on a genome without repeat families a pass has a handful of candidates in all) */
        if (HEAVY && __ballot((cm & (cm - 1u)) != 0u) != 0ull) {
          /* second level: the full 16-symbol word decides (rare on a genome without repeat families: a row passes the
           * first level with probability ~0.5 % at budget 1; inside a family nearly every row does).  GS_VU candidate rows
           * per lane and round, their loads issued side by side - the context words, then the rows' numbers in the
           * strand's suffix array (this side) or the site's position and this strand's row for it (the other side): a
           * round of ONE row per lane waited for two or three dependent gathers, and that wait, not bytes or instructions,
           * was the search's time on a repeat-rich genome (13 of 16 ms; 3.5 us per round of 64 rows). */
          while (__ballot(cm != 0u)) {
            uint32_t pkw = 0u;         /* per candidate 8 bits: 0x80 there is one, low bits = its place among the lane's 16 rows */
            uint32_t cw[GS_VU];        /* its context word */
            uint32_t co[GS_VU];        /* its row as the strand's suffix array numbers it (the other side: this strand's row of the site) */
            uint32_t cmeta[GS_VU];     /* bit 0 the guide symbols fit, bits 3:1 substitutions among them, bits 11:4 literal N under PAM symbol u */
  #pragma unroll
            for (uint32_t u = 0; u < GS_VU; ++u) {
              const bool has = cm != 0u;
              const uint32_t pick = has ? (uint32_t)__builtin_ctz(cm) : 0u;
              cm &= cm - 1u;
              pkw |= ((has ? 0x80u : 0u) | pick) << (8u * u);
              const uint32_t row = ((pick >> 3) ? row0[1] : row0[0]) + (pick & 7u);
              cw[u] = 0u;
              if (has) cw[u] = vctx[row];
              if constexpr (CNT) c_ctx += (uint32_t)__popcll(__ballot(has));
            }
  #pragma unroll
            for (uint32_t u = 0; u < GS_VU; ++u) {
              const uint32_t pb = (pkw >> (8u * u)) & 0xFFu, pick = pb & 15u;
              const bool has = (pb & 0x80u) != 0u, hi = (pick >> 3) != 0u;
              const uint32_t dy = hi ? dsc[1].y : dsc[0].y;
              const uint32_t row = (hi ? row0[1] : row0[0]) + (pick & 7u);
              const uint32_t kv = hi ? kkv[1] : kkv[0];
              const uint32_t w = cw[u];
              /* exception rows: the true symbols decide.  Under a guide symbol nothing outside
               * A,C,G,T can match or be substituted (index.hpp:31,230-247); under a PAM 'N' a literal
               * 'N' of the text matches (index.hpp:139-149) */
              uint32_t nmask = 0u; /* bit u: the text holds a literal N under PAM symbol u */
              bool excbad = false;
              const bool fl = has && ((dy >> DSC_EXC) & 1u) != 0u;
              if (__ballot(fl)) {
                if (fl) {
                  uint32_t el = 0, eh = sv.n_exc;
                  while (el < eh) {
                    const uint32_t mid = (el + eh) >> 1;
                    if (sv.exc_row[mid] < row)
                      el = mid + 1;
                    else
                      eh = mid;
                  }
                  if (el < sv.n_exc && sv.exc_row[el] == row) {
                    const uint64_t nb = sv.exc_sym[el];
                    const uint32_t upto = modeB ? g : g + P;
                    for (uint32_t j = 0; j < upto; ++j) {
                      const uint32_t c = (uint32_t)(nb >> (4u * j)) & 15u;
                      if (c > 3u) {
                        if (j < g || c != 4u)
                          excbad = true;
                        else
                          nmask |= 1u << (j - g);
                      }
                    }
                  }
                }
              }
              const uint32_t xf = (w ^ qrem) & gmask;
              const uint32_t mmv = __popc((xf | (xf >> 1)) & 0x55555555u);
              const bool gok = has && !excbad && kv + mmv <= m && mmv >= ((dy >> DSC_LO) & 7u);
              cmeta[u] = (gok ? 1u : 0u) | (mmv << 1) | (nmask << 4);
              co[u] = row;
              if (modeB) {
                if (gok) co[u] = sv.sa[row]; /* site = [pB - v_rem, pB - v_rem + L + P) on the other strand */
                if constexpr (CNT) c_isa += 2u * (uint32_t)__popcll(__ballot(gok));
              } else if (arow != nullptr) {
                if (gok) co[u] = arow[row];
                if constexpr (CNT) c_isa += (uint32_t)__popcll(__ballot(gok));
              }
            }
            if (modeB) {
  #pragma unroll
              for (uint32_t u = 0; u < GS_VU; ++u)
                if (cmeta[u] & 1u) co[u] = sd.isa[(sd.n - 1u) - (co[u] - g) - (L + P)];
            }
            /* the hits, one candidate per lane at a time (ONE copy of the emission code: the candidates move up a place) */
  #pragma unroll 1
            for (uint32_t u = 0; u < GS_VU; ++u) {
              const uint32_t pick = pkw & 15u, cmt = cmeta[0], w = cw[0], orow = co[0];
              const bool gok = (cmt & 1u) != 0u;
              const bool more = __ballot((pkw >> 8) != 0u) != 0ull;
              const bool hi = (pick >> 3) != 0u;
              const uint4 dd = hi ? dsc[1] : dsc[0];
              const uint32_t kv = hi ? kkv[1] : kkv[0], mmv = (cmt >> 1) & 7u, nmask = cmt >> 4;
              pkw >>= 8;
  #pragma unroll
              for (uint32_t v = 0; v + 1u < GS_VU; ++v) {
                cw[v] = cw[v + 1u];
                co[v] = co[v + 1u];
                cmeta[v] = cmeta[v + 1u];
              }
              cmeta[GS_VU - 1u] = 0u;
              if (__ballot(gok)) {
                const uint64_t spath = (((uint64_t)dd.w << 32) | dd.z) & PMASK;
                if (modeB) {
                  /* word symbol j is guide symbol g-1-j, complemented: the bases as this strand reads them against
                   * the guide's own; code j belongs at path bit 50 - 2 (g-1-j) */
                  const uint64_t gpath = (uint64_t)path_codes16(~w & gmask, ~qrem & gmask) << (52u + PB - 2u * g);
                  const uint64_t mmeta = ((uint64_t)(kv + mmv) << KSH) | spath | gpath;
                  route(gok, true, false, orow, orow, mmeta, 0u);
                } else {
                  for (uint32_t pj = 0; pj < npams; ++pj) {
                    const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
                    bool ok = gok;
                    uint64_t ppath = 0;
                    for (uint32_t q = 0; q < P; ++q) {
                      const uint32_t pc = (pw >> (3u * q)) & 7u;
                      const uint32_t tb = (w >> (2u * (g + q))) & 3u;
                      const bool isn = ((nmask >> q) & 1u) != 0u;
                      ok = ok && (isn ? pc == 4u : (pc == 4u || pc == tb));
                      ppath |= (uint64_t)(isn ? 3u : (tb < 3u ? tb : 4u)) << (PSP - 2u * L - 3u * q);
                    }
                    if (!__ballot(ok)) continue;
                    /* word symbol v is guide symbol k+v: code v belongs at path bit 50 - 2 (k+v), the fields in reverse order */
                    const uint64_t gpath = ((uint64_t)rev_fields16(path_codes16(w & gmask, qrem)) << 32) >> (12u - PB + 2u * k);
                    const uint64_t mmeta = ((uint64_t)(kv + mmv) << KSH) | spath | gpath | ppath;
                    route(ok, true, false, orow, orow, mmeta, 1u);
                  }
                }
              }
              if (!more) break;
            }
          }
        } else {
          /* second level, one candidate row per lane per round (rare: a row passes the first
           * level with probability ~0.5 % at budget 1): the full 16-symbol word decides */
          while (__ballot(cm != 0u)) {
            const bool has = cm != 0u;
            const uint32_t pick = has ? (uint32_t)__builtin_ctz(cm) : 0u;
            cm &= cm - 1u;
            const bool hi = (pick >> 3) != 0u;
            const uint4 dd = hi ? dsc[1] : dsc[0];
            const uint32_t row = (hi ? row0[1] : row0[0]) + (pick & 7u);
            const uint32_t kv = hi ? kkv[1] : kkv[0];
            uint32_t w = 0u;
            if (has) w = vctx[row];
            if constexpr (CNT) c_ctx += (uint32_t)__popcll(__ballot(has));
            /* exception rows: the true symbols decide.  Under a guide symbol nothing outside
             * A,C,G,T can match or be substituted (index.hpp:31,230-247); under a PAM 'N' a literal
             * 'N' of the text matches (index.hpp:139-149) */
            uint32_t nmask = 0u; /* bit u: the text holds a literal N under PAM symbol u */
            bool excbad = false;
            const bool fl = has && ((dd.y >> DSC_EXC) & 1u) != 0u;
            if (__ballot(fl)) {
              if (fl) {
                uint32_t el = 0, eh = sv.n_exc;
                while (el < eh) {
                  const uint32_t mid = (el + eh) >> 1;
                  if (sv.exc_row[mid] < row)
                    el = mid + 1;
                  else
                    eh = mid;
                }
                if (el < sv.n_exc && sv.exc_row[el] == row) {
                  const uint64_t nb = sv.exc_sym[el];
                  const uint32_t upto = modeB ? g : g + P;
                  for (uint32_t j = 0; j < upto; ++j) {
                    const uint32_t c = (uint32_t)(nb >> (4u * j)) & 15u;
                    if (c > 3u) {
                      if (j < g || c != 4u)
                        excbad = true;
                      else
                        nmask |= 1u << (j - g);
                    }
                  }
                }
              }
            }
            const uint32_t xf = (w ^ qrem) & gmask;
            const uint32_t mmv = __popc((xf | (xf >> 1)) & 0x55555555u);
            const bool gok = has && !excbad && kv + mmv <= m && mmv >= ((dd.y >> DSC_LO) & 7u);
            if (!__ballot(gok)) continue;
            const uint64_t spath = (((uint64_t)dd.w << 32) | dd.z) & PMASK;
            if (modeB) {
              /* word symbol j is guide symbol g-1-j, complemented: the bases as this strand reads them against
               * the guide's own; code j belongs at path bit 50 - 2 (g-1-j) */
              const uint64_t gpath = (uint64_t)path_codes16(~w & gmask, ~qrem & gmask) << (52u + PB - 2u * g);
              uint32_t rowA = 0;
              if (gok) {
                /* site = [pB - v_rem, pB - v_rem + L + P) on the other strand */
                const uint32_t pB = sv.sa[row];
                const uint32_t sA = (sd.n - 1u) - (pB - g) - (L + P);
                rowA = sd.isa[sA];
              }
              if constexpr (CNT) c_isa += 2u * (uint32_t)__popcll(__ballot(gok));
              const uint64_t mmeta = ((uint64_t)(kv + mmv) << KSH) | spath | gpath;
              route(gok, true, false, rowA, rowA, mmeta, 0u);
              continue;
            }
            uint32_t orow = row; /* the row as the strand's suffix array numbers it */
            if (arow != nullptr) {
              if (gok) orow = arow[row];
              if constexpr (CNT) c_isa += (uint32_t)__popcll(__ballot(gok));
            }
            for (uint32_t pj = 0; pj < npams; ++pj) {
              const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
              bool ok = gok;
              uint64_t ppath = 0;
              for (uint32_t u = 0; u < P; ++u) {
                const uint32_t pc = (pw >> (3u * u)) & 7u;
                const uint32_t tb = (w >> (2u * (g + u))) & 3u;
                const bool isn = ((nmask >> u) & 1u) != 0u;
                ok = ok && (isn ? pc == 4u : (pc == 4u || pc == tb));
                ppath |= (uint64_t)(isn ? 3u : (tb < 3u ? tb : 4u)) << (PSP - 2u * L - 3u * u);
              }
              if (!__ballot(ok)) continue;
              /* word symbol v is guide symbol k+v: code v belongs at path bit 50 - 2 (k+v), the fields in reverse order */
              const uint64_t gpath = ((uint64_t)rev_fields16(path_codes16(w & gmask, qrem)) << 32) >> (12u - PB + 2u * k);
              const uint64_t mmeta = ((uint64_t)(kv + mmv) << KSH) | spath | gpath | ppath;
              route(ok, true, false, orow, orow, mmeta, 1u);
            }
          }
        }
      }
    };

    /* seeding state (wave-uniform): position in the recipe list this strand's seeding reads */
    uint32_t spos = 0;
    const uint2 *rec = a.rec_full;
    uint32_t nrec = a.n_rec_full;
    uint32_t qn = 0; /* seeds waiting in the verification queue */
    const bool seeding = a.pt_k != 0;
    bool seeds_left = seeding;
    /* the substitution table of the item: entry 4 * step + digit = what substituting the digit-th
     * other base at that step does to the k-mer's table index and to the path; digit 3 = nothing */
    auto fill_dtab = [&](const bool sideB) __attribute__((always_inline)) {
      const uint32_t k = a.pt_k, nYb = (SPEC || a.bdeep) ? L - a.x_len : k - P, nst = sideB ? nYb : k;
      for (uint32_t e = lane; e < 4u * nst; e += WAVE) {
        const uint32_t s = e >> 2, d = e & 3u;
        const uint32_t t = sideB ? L - 1u - s : s; /* guide symbol the step consumes */
        const uint32_t qc = (uint32_t)(gr_q >> (2u * t)) & 3u;
        const uint32_t sym = (qc + 1u + d) & 3u;               /* one of the three other bases */
        const uint32_t code = 1u + sym - (sym > qc ? 1u : 0u); /* its rank among them, A<C<G<T */
        const uint64_t pb = (uint64_t)code << (PSG - 2u * t);
        /* the other strand's k-mer holds the complements: complementing both keeps the xor */
        const uint32_t sh = 2u * (sideB ? nYb - 1u - s : k - 1u - s);
        dtab[e] = d == 3u ? make_uint4(0u, 0u, 0u, 0u) : make_uint4((qc ^ sym) << sh, (uint32_t)pb, (uint32_t)(pb >> 32), 0u);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    /* a recipe applied to the exact k-mer index `pidx` / path `path`; returns the entry's address */
    /* kd = symbols the table is indexed by, dbl = log2 of the uint4 per index */
    auto apply_recipe = [&](const uint2 rc, const uint32_t rot_first, const uint32_t kd, uint32_t &pidx, uint64_t &path,
                            bool &in_rot, const bool has_rot = true) __attribute__((always_inline)) -> size_t {
      const uint32_t k = kd, n = rc.x & 7u;
      uint64_t f = (((uint64_t)rc.y << 32) | rc.x) >> 12;
      uint32_t plo = (uint32_t)path, phi = (uint32_t)(path >> 32);
      /* three substitutions per round, their table entries read side by side (entry 3 = step 0, digit 3 = nothing) */
      for (uint32_t i = 0; __ballot(i < n) != 0ull; i += 3u) {
        const uint32_t f0 = (uint32_t)f & 127u, f1 = (uint32_t)(f >> 7) & 127u, f2 = (uint32_t)(f >> 14) & 127u;
        const uint4 e0 = dtab[i < n ? f0 : 3u];
        const uint4 e1 = dtab[i + 1u < n ? f1 : 3u];
        const uint4 e2 = dtab[i + 2u < n ? f2 : 3u];
        f >>= 21;
        pidx ^= e0.x ^ e1.x ^ e2.x;
        plo |= e0.y | e1.y | e2.y;
        phi |= e0.z | e1.z | e2.z;
      }
      path = ((uint64_t)phi << 32) | plo;
      size_t ei = pidx;
      in_rot = false;
      const uint32_t rs = (rc.x >> 7) & 31u;
      if (has_rot && (rc.x & 64u) != 0u && rs >= rot_first) { /* (the deep tables have no rotated copies) */
        /* the copy rotated at step rs: that step's symbol and everything after it swap places, so the
         * recipes that differ only at step rs are neighbours in one 64-byte line */
        const uint32_t sh = 2u * (k - 1u - rs);
        const uint32_t ridx = ((pidx >> (sh + 2u)) << (sh + 2u)) | ((pidx & ((1u << sh) - 1u)) << 2) | ((pidx >> sh) & 3u);
        ei = ((size_t)(rs - rot_first) << (2u * k)) + ridx;
        in_rot = true;
      }
      return ei;
    };
    uint32_t pidx0 = 0; /* table index of the exact k-prefix of the query */
    /* context mask of this strand's seeds (gs_strand_dev::ptab): the query's symbol pairs at the four
     * pair positions after the table depth.  A pair of two guide steps may be broken by a
     * substitution (each substitution breaks at most one pair); a pair of two PAM steps must occur
     * as one of the pairs the PAM patterns allow; a pair straddling guide and PAM is not tested. */
    uint32_t asel_z = 0, asel_w = 0; /* the bit of the query's pair in each of the entry's 16-bit masks (guide pair positions) */
    uint32_t n_gpairs = 0;
    uint32_t pam_pairs = 0;   /* bit j: pair position j lies inside the PAM */
    uint32_t pam16[4] = {0u, 0u, 0u, 0u}; /* PAM pair positions: the 16-bit set of pairs some pattern allows */
    const bool use_mask = seeding && !SPEC;
    if (seeding) {
      for (uint32_t t = 0; t < a.pt_k; ++t)
        pidx0 |= ((uint32_t)(gr_q >> (2u * t)) & 3u) << (2u * (a.pt_k - 1u - t));
      for (uint32_t j = 0; j < 4u && !SPEC; ++j) {
        const uint32_t s0 = a.pt_k + ((sd.mask_off >> (4u * j)) & 15u), s1 = s0 + 1u;
        if (s1 < L) {
          const uint32_t v = (uint32_t)(gr_q >> (2u * s0)) & 15u;
          if (j < 2u)
            asel_z |= 1u << (16u * j + v);
          else
            asel_w |= 1u << (16u * (j - 2u) + v);
          n_gpairs++;
        } else if (s0 >= L && s1 < T_end) {
          pam_pairs |= 1u << j;
          for (uint32_t pj = 0; pj < npams; ++pj) {
            const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
            const uint32_t c0 = (pw >> (3u * (s0 - L))) & 7u, c1 = (pw >> (3u * (s1 - L))) & 7u;
            for (uint32_t v = 0; v < 16u; ++v)
              if ((c0 == 4u || (v & 3u) == c0) && (c1 == 4u || (v >> 2) == c1)) pam16[j] |= 1u << v;
          }
        }
      }
    } else if constexpr (WALK) {
      /* root: whole SA range, nothing consumed (index.hpp:388-391) */
      if (is_single(0, 0, 0)) {
        xs = 1;
        if (lane == 0) stk[0] = make_uint4(0u, sd.n - 1u, 0u, 0u);
      } else {
        gs = 1;
        if (lane == 0) stk[STACK_ENTRIES - 1u] = make_uint4(0u, sd.n - 1u, 0u, 0u);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    /* ---- two-sided seeding: a site with many substitutions among the first v_rem consumed
     * guide symbols (set X) has few in the rest, so it is cheap to enumerate from the other end:
     * on the other strand the same site reads reversed and complemented, and a backward search
     * there consumes the PAM first, then the guide from its last consumed symbol down to X.
     * Its depth-k seeds are PAM expansions x (o substitutions in O, b in R) for the classes
     * (o, b) the batch's plan gives to that side, each verified with the lower bound
     * a >= astar(o) against the other strand's ctx[].  What it cannot see - a literal N under
     * the PAM - is reported from the batch's window list; a PAM pattern with more than two N
     * makes the item one-sided with the full plan. */
    uint32_t pslots = 0; /* PAM-pair tables still to go through (bit per slot) */
    if (a.bidir && seeding) {
      bool fallback = false;
      const gs_strand_dev &sb = a.sd[strand ^ 1u];
      const uint32_t k = a.pt_k, sx = a.x_len, nY = k - P;
      for (uint32_t pj = 0; pj < npams && !SPEC; ++pj) {
        const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
        uint32_t nn = 0;
        for (uint32_t u = 0; u < P; ++u) nn += ((pw >> (3u * u)) & 7u) == 4u;
        if (nn > 2u) fallback = true;
      }
      /* PAM-pair tables: when every pattern of the item ends in a pair of concrete bases that has a
       * table, this strand's seeds go through those tables (one pass of the recipes per table) */
      if (!fallback && a.n_pt != 0u && P >= 2u) {
        bool all = true;
        for (uint32_t pj = 0; pj < npams; ++pj) {
          const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
          const uint32_t c0 = (pw >> (3u * (P - 2u))) & 7u, c1 = (pw >> (3u * (P - 1u))) & 7u;
          const uint32_t code = c0 | (c1 << 2);
          if (c0 > 3u || c1 > 3u)
            all = false;
          else if (code == a.pt[0][strand].code)
            pslots |= 1u;
          else if (a.n_pt > 1u && code == a.pt[1][strand].code)
            pslots |= 2u;
          else
            all = false;
        }
        if (!all) pslots = 0u;
      }
      /* a helper goes through the one table its package names (or none: the other strand's side) */
      if (helper) {
        const uint32_t hz = __builtin_amdgcn_readfirstlane(wmisc[12]);
        pslots = (!(hz & 1u) && ((hz >> 1) & 3u) < 2u) ? 1u << ((hz >> 1) & 3u) : 0u;
      }
      if (!fallback) {
        /* literal-N windows within reach whose (a, o) belongs to the other side - or all of them:
         * the PAM-pair tables hold no row with a symbol outside A,C,G,T next to it */
        const uint32_t ncand = helper ? 0u : a.n_cand[strand]; /* (the item's own wave reports the windows) */
        const uint64_t lmask = (1ull << (2u * L)) - 1ull;
        const uint64_t xmask = (1ull << (2u * sx)) - 1ull, komask = (1ull << (2u * k)) - 1ull;
#ifdef GS_X_NO_BUCKETS
        const uint32_t *cids = nullptr;
#else
        const uint32_t *cids = a.cand_ids[strand];
#endif
        const uint32_t nseg = helper ? 0u : cids != nullptr ? 4u : 1u;
        for (uint32_t sg = 0; sg < nseg; ++sg) {
        uint32_t s0 = 0, s1 = ncand;
        if (cids != nullptr) { /* the bucket of this item's chunk sg */
          const uint32_t *off = a.cand_off[strand] + 1025u * sg + ((uint32_t)(gr_q >> (10u * sg)) & 1023u);
          s0 = __builtin_amdgcn_readfirstlane(off[0]);
          s1 = __builtin_amdgcn_readfirstlane(off[1]);
        }
        for (uint32_t c0 = s0; c0 < s1; c0 += WAVE) {
          bool in = c0 + lane < s1;
          uint4 ce = make_uint4(0u, 0u, 0u, 0u);
          if (in) ce = a.cand[strand][cids != nullptr ? cids[(size_t)sg * ncand + c0 + lane] : c0 + lane];
          const uint64_t cq = ((uint64_t)ce.y << 32) | ce.x;
          const uint64_t x = cq ^ gr_q;
          /* a window that an earlier chunk spells too was reported from that chunk's bucket */
          for (uint32_t e = 0; e < sg; ++e) in = in && ((uint32_t)(x >> (10u * e)) & 1023u) != 0u;
          const uint64_t nz = (x | (x >> 1)) & 0x5555555555555555ull & lmask;
          const uint32_t tot = __popcll(nz), jx = __popcll(nz & xmask), jo = __popcll(nz & komask & ~xmask);
          const bool mine = in && tot <= m && (pslots != 0u || jx >= ((a.astar >> (4u * (jo < 7u ? jo : 7u))) & 15u));
          if (!__ballot(mine)) continue;
          uint64_t gpath = 0;
          for (uint32_t t = 0; t < L; ++t) {
            const uint32_t qc = (uint32_t)(gr_q >> (2u * t)) & 3u, tb = (uint32_t)(cq >> (2u * t)) & 3u;
            const uint32_t code = tb == qc ? 0u : 1u + tb - (tb > qc ? 1u : 0u);
            gpath |= (uint64_t)code << (PSG - 2u * t);
          }
          uint32_t rowA = 0;
          if (mine) rowA = sd.isa[ce.w];
          if constexpr (CNT) c_isa += (uint32_t)__popcll(__ballot(mine));
          for (uint32_t pj = 0; pj < npams; ++pj) {
            const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
            bool ok = mine;
            uint64_t ppath = 0;
            for (uint32_t u = 0; u < P; ++u) {
              const uint32_t pc = (pw >> (3u * u)) & 7u, tb = (ce.z >> (3u * u)) & 7u;
              ok = ok && (tb == 4u ? pc == 4u : (pc == 4u || pc == tb));
              ppath |= (uint64_t)(tb == 4u ? 3u : (tb < 3u ? tb : 4u)) << (PSP - 2u * L - 3u * u);
            }
            route(ok, true, false, rowA, rowA, ((uint64_t)tot << KSH) | gpath | ppath, 0u);
          }
        }
        }
        /* context mask for the other strand's seeds: the symbols it consumes next are the complemented
         * X symbols, last first - all guide symbols: up to four pairs, each broken by at most one
         * of the substitutions the seed's budget leaves for X */
        const uint32_t deep = SPEC ? 1u : a.bdeep;
        const uint32_t boffs = deep ? 0x6420u : sb.mask_off; /* the deep tables' masks: pairs at 0, 2, 4, 6 */
        /* bsel: the bit of the query's pair in each of the entry's four 16-bit masks (z: positions 0, 1; w: 2, 3) */
        uint32_t bsel_z = 0, bsel_w = 0, n_bpairs = 0;
        for (uint32_t j = 0; j < 4u; ++j) {
          const uint32_t o = (boffs >> (4u * j)) & 15u;
          if (o + 1u < sx) { /* both symbols inside X */
            const uint32_t v = (3u - ((uint32_t)(gr_q >> (2u * (sx - 1u - o))) & 3u)) |
                               ((3u - ((uint32_t)(gr_q >> (2u * (sx - 2u - o))) & 3u)) << 2);
            if (j < 2u)
              bsel_z |= 1u << (16u * j + v);
            else
              bsel_w |= 1u << (16u * (j - 2u) + v);
            n_bpairs++;
          }
        }
        for (uint32_t j = 0; j < sx; ++j) qrem_b |= (3u - ((uint32_t)(gr_q >> (2u * (sx - 1u - j))) & 3u)) << (2u * j);
        /* lanes of one pass over the recipes: one per recipe, or - deep tables - four, one per base under the N */
        /* (a helper has no recipes to go through: its queue comes filled and the loop below only drains it) */
        const uint32_t nlanes = helper ? 0u : deep ? 4u * a.n_rec_b : a.n_rec_b;
        if (!helper) fill_dtab(true);
        if (helper) {
          const uint32_t hz = __builtin_amdgcn_readfirstlane(wmisc[12]);
          if (hz & 1u) {
            const uint4 *pkg = a.shq + (size_t)__builtin_amdgcn_readfirstlane(wmisc[14]) * SHQ_PKG;
            if (lane < (hz >> 8)) vq[lane] = ld16_agent(pkg + 1u + lane);
            qn = hz >> 8;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          }
        }
        /* the guide part of the other strand's k-mer: step P+y holds the complement of guide symbol L-1-y */
        const uint32_t nYb = deep ? L - sx : nY;
        uint32_t pidxg = 0;
        for (uint32_t y = 0; y < nYb; ++y)
          pidxg |= (3u - ((uint32_t)(gr_q >> (2u * (L - 1u - y))) & 3u)) << (2u * (nYb - 1u - y));
        /* steps over (PAM pattern bpj, expansion be of its N's, 64 lanes from bc0 of the recipes' lane
         * space), then one last round that only drains the queue: written as one loop so that the
         * verification is instantiated once */
        uint32_t bpj = 0, be = 0, bc0 = 0, bnn = 0, pidxb = 0, bxset = 0;
        const uint4 *bdeep = nullptr; /* the deep table of the pattern's pair */
        uint64_t ppath = 0;
        bool bfinal = nlanes == 0u;
        /* the recipe of the step after this one is on its way while this one's entries are read: the list is
         * the same for every expansion and pattern, a step past its end starts it again */
        uint2 rc_next = make_uint2(0u, 0u);
        if (lane < nlanes) rc_next = a.rec_b[deep ? lane >> 2 : lane];
        for (;;) {
          uint32_t rem = 0, first = 0, jb = 0, lo = 0, eflag = 0;
          uint64_t cmeta = 0;
          if (!bfinal) {
            if (bc0 == 0u) {
              /* the PAM part: step P-1-u holds the complement of PAM symbol u ('N': the expansion's base) */
              const uint32_t pw = bpj == 0 ? gr_pam0 : bpj == 1 ? gr_pam1 : bpj == 2 ? gr_pam2 : gr_pam3;
              bnn = 0;
              for (uint32_t u = 0; u < P; ++u) bnn += ((pw >> (3u * u)) & 7u) == 4u;
              uint32_t ee = be;
              pidxb = pidxg;
              ppath = 0;
              if (deep) {
                /* the table of the pattern's pair; its first symbol picks the bases to take (bit 3 - base) */
                const uint32_t c0 = (pw >> (3u * (P - 2u))) & 7u, c1 = (pw >> (3u * (P - 1u))) & 7u, cn = pw & 7u;
                const uint32_t bslot = (a.n_pt > 1u && (c0 | (c1 << 2)) == a.pt[1][strand].code) ? 1u : 0u;
                bdeep = own_sgprs(a.pt[bslot][strand ^ 1u].deep);
                bxset = cn == 4u ? 15u : 1u << (3u - cn);
                bnn = 0;
                for (uint32_t u = 1; u < P; ++u)
                  ppath |= (uint64_t)(((pw >> (3u * u)) & 7u) < 3u ? ((pw >> (3u * u)) & 7u) : 4u) << (PSP - 2u * L - 3u * u);
              } else {
                for (uint32_t u = 0; u < P; ++u) {
                  const uint32_t pc = (pw >> (3u * u)) & 7u;
                  uint32_t base = pc;
                  if (pc == 4u) {
                    base = ee & 3u;
                    ee >>= 2;
                  }
                  ppath |= (uint64_t)(base < 3u ? base : 4u) << (PSP - 2u * L - 3u * u);
                  pidxb |= (3u - base) << (2u * (k - P + u));
                }
              }
            }
            /* one expansion = the whole recipe list rec_b: classes (o, b) one after the other; inside a
             * class (position mask) x (3^jb digit combinations), the digit of the last consumed
             * substituted symbol running fastest, so the three lanes that differ only there share
             * one 64-byte line of that step's rotated copy (or of the plain table when it is the
             * k-mer's last step) */
            const uint32_t idx = bc0 + lane;
            const uint32_t ri = deep ? idx >> 2 : idx, bx = idx & 3u;
            bool act = idx < nlanes;
            const uint2 rc = act ? rc_next : make_uint2(0u, 0u);
            {
              const uint32_t nidx = (bc0 + WAVE >= nlanes ? 0u : bc0 + WAVE) + lane;
              rc_next = make_uint2(0u, 0u);
              if (nidx < nlanes) rc_next = a.rec_b[deep ? nidx >> 2 : nidx];
            }
            count_lines(c_rec, act, a.rec_b + ri);
            jb = rc.x & 7u;
            lo = (rc.x >> 3) & 7u;
            uint32_t pidx = pidxb;
            uint64_t path = ppath;
            /* deep tables: the line of the recipe's (k-2)-mer holds one entry per base under the N */
            bool brot;
            const size_t bei = apply_recipe(rc, deep ? 31u : sb.rot_first, deep ? nYb : k, pidx, path, brot, !deep);
            const uint4 *ep = deep ? bdeep + (bei << 2) : (brot ? sb.ptab_rot : sb.ptab) + bei;
            if (deep) {
              ep += bx;
              act = act && ((bxset >> bx) & 1u) != 0u;
              path |= (uint64_t)(bx != 0u ? 3u - bx : 4u) << (PSP - 2u * L); /* that base: 3 - bx (T = 3 is coded 4) */
            }
            uint4 ent = make_uint4(0u, 0u, 0u, 0u);
            if (act) ent = *ep;
            count_lines(c_tab, act, ep);
            const uint32_t ecnt = ent.y & 0x7FFFFFFFu, mz = ent.z, mw = ent.w;
            eflag = ent.y >> 31;
            first = ent.x;
            bool live = act && ecnt != 0u && !(a.dbg_skip & 2u);
            /* fewer of the query's pairs to the left of the interval's rows than the budget left for
             * X can break: no row can match */
            const uint32_t bl = m - jb;
            if (!eflag) {
              const uint32_t intact = (uint32_t)__popc(mz & bsel_z) + (uint32_t)__popc(mw & bsel_w);
              if (intact + bl < n_bpairs) live = false;
            }
            cmeta = (WALK ? ((uint64_t)k << 59) | ((uint64_t)jb << 56) : 0ull) | path; /* (the queue keeps the count in .y) */
            rem = (live && !(a.dbg_skip & 1u)) ? ecnt : 0u;
          }
          /* the surviving seeds wait in the queue (it is this phase's alone: one-sided seeding has not
           * started) until a pass can be filled - a pass costs the same instructions for 10 seeds as
           * for 64; intervals larger than a descriptor holds are queued piece by piece */
          for (;;) {
            if (guard_left == 0u) {
              bailed = true;
              break;
            }
            guard_left--;
            const uint64_t bq = __ballot(rem != 0u);
            if (bq && qn + WAVE <= VQ_CAP) {
              const uint32_t rows = rem < a.v_max ? rem : a.v_max;
              if (rem != 0u)
                vq[qn + lanes_below(bq)] = make_uint4(first, (jb << 14) | (rows << 17) | (lo << DSC_LO) | (eflag << DSC_EXC),
                                                      (uint32_t)cmeta, (uint32_t)(cmeta >> 32));
              qn += __popcll(bq);
              first += rows;
              rem -= rows;
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              continue;
            }
            if (qn >= VQ_DRAIN || (bq && qn != 0u) || (bfinal && qn != 0u)) {
              const uint32_t take = qn < WAVE ? qn : WAVE;
              qn -= take;
              verify(true, take, vq + qn);
              continue;
            }
            break;
          }
          if (bfinal || bailed) break;
          bc0 += WAVE;
          if (bc0 >= nlanes) {
            bc0 = 0;
            if (++be >= (1u << (2u * bnn))) {
              be = 0;
              if (++bpj >= npams) bfinal = true;
            }
          }
        }
      }
      if (fallback) {
        if (!helper) n_fb++; /* every seed from this strand */
      } else {
        rec = pslots ? a.rec_a8 : a.rec_a;
        nrec = pslots ? a.n_rec_a8 : a.n_rec_a;
        if (!helper) {
          n_two++;
          if (pslots) n_pair++;
        }
      }
    }
    if (helper) nrec = 0u;
    auto next_pairtab = [&]() __attribute__((always_inline)) {
      const uint32_t s = (pslots & 1u) ? 0u : 1u;
      pslots &= ~(1u << s);
      const gs_pairtab_dev &p = a.pt[s][strand];
      atab8 = own_sgprs(p.tab);
      arot8 = own_sgprs(p.rot);
      arot_first = p.rot_first;
      a16 = own_sgprs(p.c16);
      actx = own_sgprs(p.ctx);
      arow = own_sgprs(p.rowid);
    };
    if (pslots) next_pairtab();
    /* the guide symbols this strand's seeds leave to the context check */
    const uint32_t gA = L - a.pt_k, gmaskA = gA >= 16u ? 0xFFFFFFFFu : ((1u << (2u * gA)) - 1u);
    const uint32_t qremA = seeding ? (uint32_t)(gr_q >> (2u * a.pt_k)) & gmaskA : 0u;
    /* against a PAM-pair table entry's filter: the nearest 13 symbols of a single row; the query's nearest 6
     * as one-hot nibbles against the symbol sets of several rows */
    const uint32_t gmask13 = gA >= 13u ? 0x3FFFFFFu : gmaskA;
    uint32_t qhot = 0u;
    for (uint32_t j = 0; j < 6u && j < gA; ++j) qhot |= 1u << (4u * j + ((qremA >> (2u * j)) & 3u));
    if (seeding && !helper) {
      fill_dtab(false);
      if (nrec == 0u) seeds_left = false;
    }
    if (helper) {
      /* this strand's side of a package: ONE seeding step without recipes, whose queue loop finds the pass ended and
       * the queue filled - the verification below is the one the item's own wave would have run */
      const uint32_t hz = __builtin_amdgcn_readfirstlane(wmisc[12]);
      seeds_left = !(hz & 1u);
      if (!(hz & 1u)) {
        const uint4 *pkg = a.shq + (size_t)__builtin_amdgcn_readfirstlane(wmisc[14]) * SHQ_PKG;
        if (lane < (hz >> 8)) vq[lane] = ld16_agent(pkg + 1u + lane);
        qn = hz >> 8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      }
    }
    /* as on the other side: the next step's recipe is requested a step ahead */
    uint2 rc_ahead = make_uint2(0u, 0u);
    if (seeding && lane < nrec) rc_ahead = rec[lane];

    for (;;) {
      if (bailed) break;
      if constexpr (WALK) { /* the seeding steps count in their queue loop */
        if (guard_left == 0u) {
          bailed = true;
          break;
        }
        guard_left--;
      }
      const uint32_t total = xs + gs;
      if (seeds_left && total <= seed_low) {
        /* ---- seed depth-k nodes from the prefix interval table -------------------------
         * The top of the search tree is input independent and full down to depth ~log4(n):
         * instead of walking it, enumerate every variant of the first k-2 query symbols with
         * j <= m substitutions and, one table entry per lane, the two-symbol extensions the
         * remaining budget allows.  Same node set at depth k as the walk (index.hpp:182-248). */
        const uint32_t k = a.pt_k;
        /* Seeds = the recipes of the list, 64 per step whatever classes they belong to.  Their order
         * (gs_build_recipes_a) keeps table lines shared: the 16 two-symbol extensions of a variant of
         * the first k-2 symbols are 4 lines of the plain table; with one substitution left 4 + 3
         * recipes read one line of the plain table (last symbol runs) and one of the copy rotated at
         * step k-2 (second-last runs); with none left the three recipes that differ in the digit of
         * their last substituted step read one line of that step's rotated copy. */
        const uint32_t l = spos + lane;
        const bool act = l < nrec;
        const uint2 rc = act ? rc_ahead : make_uint2(0u, 0u);
        {
          const uint32_t nl = (spos + WAVE >= nrec ? 0u : spos + WAVE) + lane; /* past the end: the next table's pass */
          rc_ahead = make_uint2(0u, 0u);
          if (nl < nrec) rc_ahead = rec[nl];
        }
        count_lines(c_rec, act, rec + l);
        const uint32_t kk = rc.x & 7u;
        uint32_t pidx = pidx0;
        uint64_t path = 0;
        uint4 ent = make_uint4(0u, 0u, 0u, 0u);
        bool in_rot;
        const size_t ei = apply_recipe(rc, arot_first, a.pt_k, pidx, path, in_rot);
        const uint32_t bl = m - kk; /* budget left (>= 0 by construction) */
        bool hopeless = false;
        uint32_t ecnt, eflag = 0u;
        if (SPEC || arow != nullptr) {
          /* PAM-pair table: 8-byte entry {first row of the table's own arrays, rows (6 bits) | filter} */
          const uint2 *ep = (in_rot ? arot8 : atab8) + ei;
          uint2 e8 = make_uint2(0u, 0u);
          if (act) e8 = *ep;
          count_lines(c_tab, act, ep);
          ent.x = e8.x;
          ecnt = e8.y & 63u;
          const uint32_t filt = e8.y >> 6;
          if (ecnt == 1u) {
            const uint32_t xf = (filt ^ qremA) & gmask13; /* the row's own context symbols */
            hopeless = (uint32_t)__popc((xf | (xf >> 1)) & 0x55555555u) > bl;
          } else if (ecnt > 1u) {
            /* a query symbol none of the rows shows at its position is a substitution in every row */
            hopeless = (uint32_t)__popc(qhot & ~filt) > bl;
            if (ecnt == GS_PT_BIG && !hopeless) { /* 63 rows and more: the count sits in a header slot in front of them */
              ecnt = arow[ent.x];
              ent.x += 1u;
              if constexpr (CNT) c_isa += (uint32_t)__popcll(__ballot(true));
            }
          }
        } else {
          const uint4 *ep = (in_rot ? arot : atab) + ei;
          if (act) ent = *ep;
          count_lines(c_tab, act, ep);
          ecnt = ent.y & 0x7FFFFFFFu;
          eflag = ent.y >> 31;
          /* context mask: drop the seed when fewer of the query's symbol pairs occur to the left of its
           * interval's rows than the remaining budget can break, or none of the PAM's pairs does */
          if (use_mask && eflag == 0u) {
            const uint32_t em[4] = {ent.z & 0xFFFFu, ent.z >> 16, ent.w & 0xFFFFu, ent.w >> 16};
            const uint32_t intact = (uint32_t)__popc(ent.z & asel_z) + (uint32_t)__popc(ent.w & asel_w);
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j)
              if (((pam_pairs >> j) & 1u) && (em[j] & pam16[j]) == 0u) hopeless = true;
            hopeless = hopeless || intact + bl < n_gpairs;
          }
        }
        const bool live = act && ecnt != 0u && !hopeless && !(a.dbg_skip & 2u);
        const uint64_t cmeta = (WALK ? ((uint64_t)k << 59) | ((uint64_t)kk << 56) : 0ull) | path;
        /* every interval is resolved right here against ctx[] (exception rows included, large
         * ones in pieces); without the context arrays the seeds continue as ordinary nodes
         * (k < L: never terminal) */
        const bool ver = live && (!WALK || a.v_rem != 0u);
        if constexpr (WALK) route(live && !ver, false, kk == m, ent.x, ent.x + ecnt - 1u, cmeta);
        spos += WAVE;
        const bool pass_end = spos >= nrec; /* of the recipes through one table */
        /* The verifying seeds wait in the queue until a pass can be filled: about a quarter of
         * a step's 64 lanes survive the context mask, and a pass (prefix sums, owner lookup,
         * row groups) costs the same instructions for 16 seeds as for 64.  Drain from the tail
         * (no shifting): when at least VQ_DRAIN wait, when the next push might not fit, and
         * everything once the seeds are exhausted. */
        uint32_t rem = (ver && !(a.dbg_skip & 1u)) ? ecnt : 0u, first = ent.x;
        for (;;) {
          if (guard_left == 0u) {
            bailed = true;
            break;
          }
          guard_left--;
          const uint64_t bq = __ballot(rem != 0u);
          if (bq && qn + WAVE <= VQ_CAP) {
            const uint32_t rows = rem < a.v_max ? rem : a.v_max;
            if (rem != 0u)
              vq[qn + lanes_below(bq)] = make_uint4(first, (kk << 14) | (rows << 17) | (eflag << DSC_EXC),
                                                    (uint32_t)cmeta, (uint32_t)(cmeta >> 32));
            qn += __popcll(bq);
            first += rows;
            rem -= rows;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            continue;
          }
          if (qn >= VQ_DRAIN || (bq && qn != 0u) || (pass_end && qn != 0u)) {
            const uint32_t take = qn < WAVE ? qn : WAVE;
            qn -= take;
            verify(false, take, vq + qn);
            continue;
          }
          break;
        }
        if (pass_end) {
          if (pslots) { /* the queue is empty: the same recipes through the next PAM-pair table */
            next_pairtab();
            spos = 0;
          } else {
            seeds_left = false;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        continue;
      }
      if (total == 0) break;
      if constexpr (WALK) {
      const uint32_t room = total < limit ? limit - total : 0u;

      if (xs > 0 && (xs >= WAVE || gs == 0 || room < (MAX_FANOUT - 1) * WAVE)) {
        /* ---- X iteration: one symbol, at most one child, stack cannot grow ------------ */
        const uint32_t w = xs < WAVE ? xs : WAVE;
        const bool active = lane < w;
        uint4 nd = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active) nd = stk[xs - 1u - lane];
        xs -= w;
        n_ext += w;
        const uint32_t sp = nd.x, ep = nd.y;
        const uint64_t meta = ((uint64_t)nd.w << 32) | nd.z;
        const uint32_t t = META_T(meta), k = META_K(meta), pamid = META_PAM(meta);
        uint64_t path = meta & PATH_MASK;
        uint32_t c;
        if (t < L) {
          c = (uint32_t)(gr_q >> (2u * t)) & 3u; /* exact child keeps code 0 (upper case) */
        } else {
          const uint32_t pw =
              pamid == 0 ? gr_pam0 : pamid == 1 ? gr_pam1 : pamid == 2 ? gr_pam2 : gr_pam3;
          c = (pw >> (3u * (t - L))) & 3u; /* fixed PAM base (code < 4 by construction) */
          path |= (uint64_t)(c < 3u ? c : 4u) << (PSP - 2u * L - 3u * (t - L));
        }
        uint32_t oa = 0, ob = 0;
        if (active) {
          /* Occ(c, sp) and Occ(c, ep+1): rows before sp in sp's block, rows up to and including
           * ep in ep's block - the same 64-byte line whenever the interval does not straddle */
          oa = occ1(blocks, sp >> GS_BLOCK_SHIFT, sp & (GS_BLOCK_ROWS - 1u), c);
          ob = occ1(blocks, ep >> GS_BLOCK_SHIFT, (ep & (GS_BLOCK_ROWS - 1u)) + 1u, c);
        }
        if constexpr (CNT)
          c_occ += (uint32_t)__popcll(__ballot(active)) +
                   (uint32_t)__popcll(__ballot(active && (sp >> GS_BLOCK_SHIFT) != (ep >> GS_BLOCK_SHIFT)));
        const uint32_t Cc = c == 0 ? sd.C[0] : c == 1 ? sd.C[1] : c == 2 ? sd.C[2] : sd.C[3];
        const uint32_t t2 = t + 1u;
        const uint64_t cmeta =
            ((uint64_t)t2 << 59) | ((uint64_t)k << 56) | ((uint64_t)pamid << 52) | path |
            ((t2 == L && fanning) ? (1ull << 54) : 0ull);
        route(active && ob > oa, t2 == T_end, is_single(t2, k, pamid), Cc + oa, Cc + ob - 1u, cmeta);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        continue;
      }

      /* ---- G iteration: up to four substitution children, PAM wildcard, PAM fan-out ------ */
      uint32_t w = gs < WAVE ? gs : WAVE;
      {
        /* never pop more than the stacks can take children for; w >= 1 keeps a plain DFS
         * going, whose depth (T_end) is covered by `reserve` */
        const uint32_t fit = room / (MAX_FANOUT - 1);
        if (w > fit) w = fit ? fit : 1u;
      }
      const bool active = lane < w;
      uint4 nd = make_uint4(0, 0, 0, 0);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (active) nd = stk[STACK_ENTRIES - gs + lane];
      gs -= w;

      const uint32_t sp = nd.x, ep = nd.y;
      const uint64_t meta = ((uint64_t)nd.w << 32) | nd.z;
      const uint32_t t = META_T(meta), k = META_K(meta);
      const bool fan = META_FAN(meta) != 0;
      const uint32_t pamid = META_PAM(meta);
      const uint64_t path = meta & PATH_MASK;
      const bool ext = active && !fan;
      n_ext += __popcll(__ballot(ext));

      uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
      if (ext) {
        occ4(blocks, sp >> GS_BLOCK_SHIFT, sp & (GS_BLOCK_ROWS - 1u), a0, a1, a2, a3);
        occ4(blocks, ep >> GS_BLOCK_SHIFT, (ep & (GS_BLOCK_ROWS - 1u)) + 1u, b0, b1, b2, b3);
      }
      if constexpr (CNT)
        c_occ += (uint32_t)__popcll(__ballot(ext)) +
                 (uint32_t)__popcll(__ballot(ext && (sp >> GS_BLOCK_SHIFT) != (ep >> GS_BLOCK_SHIFT)));

      /* which symbols may be tried, and what they cost */
      const bool inpam = t >= L;
      uint32_t qc = 0, allow = 0, pc = 0;
      if (!inpam) {
        qc = (uint32_t)(gr_q >> (2u * t)) & 3u;
        allow = (k < m) ? 0xFu : (1u << qc); /* index.hpp:230 */
      } else {
        const uint32_t pw =
            pamid == 0 ? gr_pam0 : pamid == 1 ? gr_pam1 : pamid == 2 ? gr_pam2 : gr_pam3;
        pc = (pw >> (3u * (t - L))) & 7u;
        allow = pc < 4u ? (1u << pc) : 0xFu; /* 'N' tries A,T,C,G at cost 0: index.hpp:151-169 */
      }
      const uint32_t t2 = t + 1u;
      const bool term = (t2 == T_end);
      const bool needfan = (t2 == L) && fanning;
      const uint32_t sh_g = PSG - 2u * t;                                    /* guide step: 2-bit code */
      const uint32_t sh_p = inpam ? PSP - 2u * L - 3u * (t - L) : 0u; /* PAM step: 3-bit code */

#pragma unroll
      for (uint32_t c = 0; c < MAX_FANOUT; ++c) {
        bool live = false, cterm = false, single = false;
        uint32_t csp = 0, cep = 0;
        uint64_t cmeta = 0;
        if (c < 4u) {
          if (fan) {
            /* PAM fan-out: one copy of the finished 20-mer node per PAM pattern (index.hpp:212-214) */
            live = active && c < npams;
            csp = sp;
            cep = ep;
            cmeta = (meta & ~((1ull << 54) | (3ull << 52))) | ((uint64_t)c << 52);
            const uint32_t pw0 = c == 0 ? gr_pam0 : c == 1 ? gr_pam1 : c == 2 ? gr_pam2 : gr_pam3;
            single = (pw0 & 7u) < 4u;
          } else {
            const uint32_t oa = c == 0 ? a0 : c == 1 ? a1 : c == 2 ? a2 : a3;
            const uint32_t ob = c == 0 ? b0 : c == 1 ? b1 : c == 2 ? b2 : b3;
            live = ext && ((allow >> c) & 1u) && ob > oa; /* occ_within > 0 */
            csp = sd.C[c] + oa;
            cep = sd.C[c] + ob - 1u;
            uint64_t p2;
            uint32_t k2 = k;
            if (!inpam) {
              const uint32_t code = (c == qc) ? 0u : 1u + c - (c > qc ? 1u : 0u);
              p2 = path | ((uint64_t)code << sh_g);
              k2 += (c != qc);
            } else {
              const uint32_t code = c < 3u ? c : 4u; /* A=0 C=1 G=2 (N=3) T=4 */
              p2 = path | ((uint64_t)code << sh_p);
            }
            cmeta = ((uint64_t)t2 << 59) | ((uint64_t)k2 << 56) | ((uint64_t)pamid << 52) | p2 |
                    (needfan ? (1ull << 54) : 0ull);
            cterm = term;
            single = is_single(t2, k2, pamid);
          }
        } else {
          /* literal 'N' of the genome under a PAM 'N' (index.hpp:139-149); rare */
          if (sd.has_n && sd.nruns) {
            const bool want = ext && inpam && pc == 4u;
            if (__ballot(want)) {
              if (want) {
                const uint32_t na = occ_n(sd, sp), nb = occ_n(sd, ep + 1u);
                live = nb > na;
                csp = sd.CN + na;
                cep = sd.CN + nb - 1u;
                cmeta = ((uint64_t)t2 << 59) | ((uint64_t)k << 56) | ((uint64_t)pamid << 52) | path |
                        (3ull << sh_p);
                cterm = term;
                single = is_single(t2, k, pamid);
              }
            }
          }
        }
        route(live, cterm, single, csp, cep, cmeta);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      } /* WALK */
    }
    if (helper) {
      /* the episode's records, chunks and class counts join the item's through sh_acc; its last chunk says what it holds
       * (k_share_fix closes the gaps behind the launch) */
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t nch = __builtin_amdgcn_readfirstlane(wmisc[0]), last = __builtin_amdgcn_readfirstlane(wmisc[1]);
      const bool short_of = n_match > (nch << ARENA_SHIFT); /* the arena ran out: counted, not kept (the host searches the batch's overflow again) */
      const uint32_t h_sid = __builtin_amdgcn_readfirstlane(wmisc[13]);
      if (lane == 0 && n_match != 0u) atomicAdd(&a.sh_acc[16u * h_sid], n_match);
      if (lane == 1u && nch != 0u && !short_of) a.chunk_fill[last] = n_match - ((nch - 1u) << ARENA_SHIFT);
      if (lane >= 8u && lane < 16u && wmisc[lane - 4u] != 0u) atomicAdd(&a.sh_acc[16u * h_sid + lane], wmisc[lane - 4u]);
      if (short_of) n_fail++;
      if (prof != nullptr) {
        const unsigned long long t = wall_clock64();
        t_help += t - t_prev;
        t_prev = t;
        n_epi++;
      }
      continue;
    }
    if (lane == 0) a.counts[slot] = n_match;
    if (n_match > item_cap) n_ovf++;
    if (a.arena != nullptr) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t nch = __builtin_amdgcn_readfirstlane(wmisc[0]), last = __builtin_amdgcn_readfirstlane(wmisc[1]);
      /* (a shared item: its number instead of the last chunk - nobody appends to it - for k_share_dir/fix) */
      const uint32_t sid = __builtin_amdgcn_readfirstlane(wmisc[3]);
      if (lane == 0) a.nchunk[slot] = make_uint2(nch, (HEAVY && sid < a.sh_max) ? 0x80000000u | sid : last);
      if (lane < 8u) a.cls[(size_t)slot * 8u + lane] = wmisc[4u + lane];
      if (n_match > item_cap && n_match - item_cap > (nch << ARENA_SHIFT)) n_fail++;
    }
  }
  if (prof != nullptr && lane == 0) {
    atomicMax(&prof[2], wall_clock64());
    atomicAdd(&prof[4], t_help);
    atomicAdd(&prof[5], t_wait);
    atomicAdd(&prof[6], n_epi);
  }
  if (lane == 0) {
    if (n_ext) atomicAdd(&a.stats[0], n_ext);
    if (n_ovf) atomicAdd(&a.stats[1], n_ovf);
    if (n_fail) atomicAdd(&a.stats[6], (unsigned long long)n_fail);
    if (!WALK && n_hpass) atomicAdd(a.hpass, n_hpass);
    if (bailed) atomicOr(a.err, 1u);
    if (n_two) atomicAdd(&a.stats[4], (unsigned long long)n_two);
    if (n_fb) atomicAdd(&a.stats[5], (unsigned long long)n_fb);
    if (n_pair) atomicAdd(&a.stats[7], (unsigned long long)n_pair);
    if constexpr (CNT) {
      atomicAdd(&a.stats[8], (unsigned long long)c_tab);
      atomicAdd(&a.stats[9], (unsigned long long)c_c16);
      atomicAdd(&a.stats[10], (unsigned long long)c_ctx);
      atomicAdd(&a.stats[11], (unsigned long long)c_isa);
      atomicAdd(&a.stats[12], (unsigned long long)c_occ);
      atomicAdd(&a.stats[3], (unsigned long long)c_rec);
    }
  }
}

#define GS_DEF_SEARCH(NAME, CNT, WALK, SPEC, WEU, ...)                                                               \
  __global__ __launch_bounds__(WAVE *SEARCH_WAVES) __attribute__((amdgpu_waves_per_eu(WEU, WEU))) void NAME(          \
      gs_search_args a) {                                                                                             \
    __shared__ uint4 s_stack[SEARCH_WAVES][(WALK) ? WAVE_LDS_ENTRIES : WAVE_LDS_FAST];                                 \
    k_search_body<CNT, WALK, SPEC, ##__VA_ARGS__>(a, s_stack[threadIdx.x / WAVE]);                                     \
  }
GS_DEF_SEARCH(k_search_walk, false, true, false, GS_WAVES_EU)       /* reference-order walk; remainders beyond ctx[] */
GS_DEF_SEARCH(k_search_fast, false, false, false, GS_WAVES_EU_FAST) /* table seeding, any mix of tables */
GS_DEF_SEARCH(k_search_count, true, false, false, GS_WAVES_EU_FAST) /* the same with the request tally (bench.py) */
/* every item through PAM-pair + deep tables: what an NGG / NAG / TTN ... batch runs (the timed kernel of bench.py) */
#ifndef GS_WAVES_EU_PD
#define GS_WAVES_EU_PD 8
#endif
GS_DEF_SEARCH(k_search_fast_pd, false, false, true, GS_WAVES_EU_PD)
GS_DEF_SEARCH(k_search_count_pd, true, false, true, GS_WAVES_EU_PD)
/* HEAVY: the instantiations for a handle whose earlier batches showed items of thousands of records (a repeat-rich genome; m >= 5):
 * heavy verification passes are handed to the waves that ran out of items (gs_search_args::shq) and the second level of the
 * verification keeps GS_VU rows per lane in flight.  The same results; on a genome without such items the plain forms are faster
 * (1 M guides at m <= 3: 21 ms against 28-32 - the second level is rare there and its unrolled form costs instructions and registers). */
#ifndef GS_WAVES_EU_HEAVY
#define GS_WAVES_EU_HEAVY 8
#endif
GS_DEF_SEARCH(k_search_heavy, false, false, false, GS_WAVES_EU_HEAVY, true)
GS_DEF_SEARCH(k_search_heavy_pd, false, false, true, GS_WAVES_EU_HEAVY, true)

/* ---- prepare: ASCII -> packed records (process.hpp:51-63) ------------------ */
__device__ __forceinline__ int base_code(uint8_t c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}


__global__ void k_prepare(gs_prep_args a) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.n) return;
  gs_guide_rec r;
  r.q = 0;
  r.valid = a.force_invalid ? 0u : 1u;
  r.npams = 0;
  for (int j = 0; j < 4; j++) r.pam[j] = 0;
  const uint8_t *s = a.guides + (size_t)g * a.L;
  /* query = reverse_complement(sequence) consumed right to left == complement of the guide left
   * to right (process.hpp:63, index.hpp:218); with --start the guide itself right to left */
  for (uint32_t t = 0; t < a.L; t++) {
    int c = a.start ? base_code(s[a.L - 1 - t]) : base_code(s[t]);
    if (c < 0) {
      r.valid = 0;
      c = 0;
    }
    if (!a.start) c = 3 - c; /* complement in A,C,G,T = 0..3 */
    r.q |= (uint64_t)c << (2 * t);
  }
  if (a.P > 0) {
    const uint8_t *own = a.guide_pams + (size_t)g * a.P;
    /* pams = alt_pams ++ [k.pam] (process.hpp:51-56); this record holds four of them */
    const uint32_t np = a.n_alt + 1;
    for (uint32_t u = 0; u < a.P; u++) /* the guide's own PAM decides validity whatever the chunk */
      if (own[u] != 'N' && base_code(own[u]) < 0) r.valid = 0;
    for (uint32_t j = 4u * a.chunk; j < np && j < 4u * a.chunk + 4u; j++) {
      const uint8_t *p = j < a.n_alt ? a.alt[j] : own;
      uint32_t code = 0;
      for (uint32_t u = 0; u < a.P; u++) {
        uint8_t ch = a.start ? p[a.P - 1 - u] : p[u];
        int c;
        if (ch == 'N') {
          c = 4;
        } else {
          c = base_code(ch);
          if (c < 0)
            c = 0; /* the guide is invalid (own PAM); alt PAMs were checked on the host */
          else if (!a.start)
            c = 3 - c;
        }
        code |= (uint32_t)c << (3 * u);
      }
      r.pam[r.npams++] = code;
    }
  } else {
    r.npams = 1;
  }
  if (a.pair_hist != nullptr && a.P >= 2u) {
    /* which PAM-pair tables would serve this batch: one atomic per wave and distinct pair */
    for (uint32_t j = 0; j < 4u; j++) {
      const bool has = r.valid && j < r.npams;
      const uint32_t c0 = (r.pam[j] >> (3u * (a.P - 2u))) & 7u, c1 = (r.pam[j] >> (3u * (a.P - 1u))) & 7u;
      const uint32_t code = (c0 > 3u || c1 > 3u) ? 16u : (c0 | (c1 << 2));
      uint64_t todo = __ballot(has);
      while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const uint32_t c = (uint32_t)__shfl((int)code, l);
        const uint64_t same = __ballot(has && code == c);
        if ((int)lane_id() == l) atomicAdd(&a.pair_hist[c], (uint32_t)__popcll(same));
        todo &= ~same;
      }
    }
  }
  if (a.chunk == 0) {
    if (!r.valid) atomicAdd(a.n_invalid, 1u);
    if (a.flags) a.flags[g] = r.valid ? 0u : 1u;
  }
  a.out[g] = r;
}

void gs_launch_prepare(const gs_prep_args &pa, hipStream_t st) {
  hipLaunchKernelGGL(k_prepare, dim3((pa.n + 255) / 256), dim3(256), 0, st, pa);
}

/* ---- order: per guide canonical order + dedupe ----------------------------- */
struct gs_order_args {
  uint4 *slots;           /* in: [n][2][cap] raw ; out: [n][2*cap] ordered unique {key_lo,key_hi,sp,cnt} */
  const uint32_t *counts; /* [2n] */
  uint32_t *nmatch;       /* [n] */
  uint32_t *nhits;        /* [n] */
  unsigned long long *stats; /* [2] total matches */
  uint32_t n, cap;
};

/* One wavefront per guide at a time, ORDER_WAVES wavefronts per workgroup, guides dealt to the
 * waves grid-stride (a launch of one single-wave workgroup per guide with one atomic each was
 * latency bound: 12 ms per 1 M guides).  Dynamic LDS per wave: 2*cap uint4 (records) + ORDER_SMALL
 * uint4 (rank-sort output).  Up to ORDER_SMALL records a guide is rank-sorted (M^2/64 compares per
 * lane: 11 at the 26 records of an m = 3 guide); larger guides go through a bitonic network in
 * place (log^2 N / 2 steps of N/128 compare-exchanges per lane: at the 1,440 records of an m = 5
 * guide 2.1 k per lane instead of 32 k).  The loop body has no lane-conditional blocks (DESIGN.md
 * 5b, compiler pitfall): per-guide results are stored by all lanes to the same address. */
#define ORDER_WAVES 4
#define ORDER_SMALL 128u
__global__ __launch_bounds__(WAVE *ORDER_WAVES) void k_order(gs_order_args a) {
  extern __shared__ uint4 s_mem[];
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
  const uint32_t cap = a.cap;
  uint4 *rec = s_mem + (size_t)wave * (2u * cap + ORDER_SMALL);
  uint4 *srt = rec + 2u * cap;
  uint32_t total_out = 0;
  for (uint32_t g = blockIdx.x * nw + wave; g < a.n; g += gridDim.x * nw) {
    const uint32_t c0 = a.counts[2 * g], c1 = a.counts[2 * g + 1];
    if (c0 > cap || c1 > cap) {
      /* more matches than slots: this guide is redone with larger slots (host side); the
       * redo's totals are patched in before the scan */
      a.nmatch[g] = 0;
      a.nhits[g] = 0;
      continue;
    }
    const uint32_t M = c0 + c1;
    uint4 *base = a.slots + (size_t)g * 2 * cap;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t i = lane; i < M; i += WAVE) rec[i] = i < c0 ? base[i] : base[cap + (i - c0)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint4 *sorted = srt;
    if (M <= ORDER_SMALL) {
      /* rank sort: ascending (key, first row, original index) */
      for (uint32_t i = lane; i < M; i += WAVE) {
        const uint4 me = rec[i];
        const uint64_t key = ((uint64_t)me.y << 32) | me.x;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < M; j++) {
          const uint4 o = rec[j];
          const uint64_t ok = ((uint64_t)o.y << 32) | o.x;
          rank += (ok < key) || (ok == key && (o.z < me.z || (o.z == me.z && j < i)));
        }
        srt[rank] = me;
      }
    } else {
      /* bitonic network over N = the next power of two, padded with records that sort last */
      uint32_t N = 2u * ORDER_SMALL;
      while (N < M) N <<= 1;
      for (uint32_t i = M + lane; i < N; i += WAVE) rec[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      for (uint32_t kk = 2; kk <= N; kk <<= 1)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          for (uint32_t t = lane; t < (N >> 1); t += WAVE) {
            const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), hi = lo | j;
            const uint4 A = rec[lo], B = rec[hi];
            const uint64_t ka = ((uint64_t)A.y << 32) | A.x, kb = ((uint64_t)B.y << 32) | B.x;
            const bool gt = ka > kb || (ka == kb && A.z > B.z);
            if (gt == ((lo & kk) == 0u)) {
              rec[lo] = B;
              rec[hi] = A;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
      sorted = rec;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    /* dedupe equal sequences (std::set keeps the first), compact, count hits */
    uint32_t n_out = 0, hits = 0;
    for (uint32_t i0 = 0; i0 < M; i0 += WAVE) {
      const uint32_t i = i0 + lane;
      bool keep = false;
      uint4 me = make_uint4(0, 0, 0, 0);
      if (i < M) {
        me = sorted[i];
        keep = true;
        if (i > 0) {
          const uint4 pv = sorted[i - 1];
          keep = !(pv.x == me.x && pv.y == me.y && pv.z == me.z); /* same sequence, same rows */
        }
      }
      const uint64_t kb = __ballot(keep);
      const uint32_t cnt = keep ? (me.w - me.z + 1u) : 0u;
      if (keep) base[n_out + lanes_below(kb)] = make_uint4(me.x, me.y, me.z, cnt);
      n_out += __popcll(kb);
      /* wave sum of cnt */
      uint32_t s = cnt;
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      hits += s;
    }
    a.nmatch[g] = n_out;
    a.nhits[g] = hits;
    total_out += n_out;
  }
  if (lane == 0 && total_out) atomicAdd(&a.stats[2], (unsigned long long)total_out);
}

/* The same for guides with hundreds to thousands of matches (cap > 128): one 256-thread workgroup
 * per guide at a time, the bitonic network spread over its four waves (a single wave needs 2.1 k
 * compare-exchange rounds for the 1,440 records of an m = 5 guide), LDS sized by the largest guide
 * of the batch (`nmax` records, a power of two) rather than by the slot capacity. */
__global__ __launch_bounds__(256) void k_order_wg(gs_order_args a, uint32_t nmax) {
  extern __shared__ uint4 s_mem[];
  uint4 *rec = s_mem;
  __shared__ uint32_t s_nout, s_hits;
  const uint32_t tid = threadIdx.x, lane = lane_id();
  const uint32_t cap = a.cap;
  uint32_t total_out = 0;
  for (uint32_t g = blockIdx.x; g < a.n; g += gridDim.x) {
    const uint32_t c0 = a.counts[2 * g], c1 = a.counts[2 * g + 1];
    const uint32_t M = c0 + c1;
    if (c0 > cap || c1 > cap || M > nmax) { /* redone with larger slots (host side) */
      if (tid == 0) {
        a.nmatch[g] = 0;
        a.nhits[g] = 0;
      }
      continue;
    }
    uint4 *base = a.slots + (size_t)g * 2 * cap;
    uint32_t N = 64;
    while (N < M) N <<= 1;
    __syncthreads(); /* the previous guide's compaction has finished reading rec[] */
    for (uint32_t i = tid; i < N; i += 256)
      rec[i] = i < c0 ? base[i] : i < M ? base[cap + (i - c0)] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    /* Compare-exchange t touches lo and lo | j.  With j <= 64 the 64 exchanges of a wave stay inside one
     * aligned block of 128 records - the same block for every such j - so a pass needs the workgroup
     * barrier only when it or the pass before it reaches further (j >= 128): 14 barriers instead of
     * 66 at N = 2,048; the other passes order their LDS accesses within the wave. */
    uint32_t j_prev = 128;
    for (uint32_t kk = 2; kk <= N; kk <<= 1)
      for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
        if (j >= 128u || j_prev >= 128u) {
          __syncthreads();
        } else {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        j_prev = j;
        for (uint32_t t = tid; t < (N >> 1); t += 256) {
          const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), hi = lo | j;
          const uint4 A = rec[lo], B = rec[hi];
          const uint64_t ka = ((uint64_t)A.y << 32) | A.x, kb = ((uint64_t)B.y << 32) | B.x;
          const bool gt = ka > kb || (ka == kb && A.z > B.z);
          if (gt == ((lo & kk) == 0u)) {
            rec[lo] = B;
            rec[hi] = A;
          }
        }
      }
    __syncthreads();
    /* dedupe equal sequences (std::set keeps the first), compact, count hits: the first wave alone */
    if (tid < WAVE) {
      uint32_t n_out = 0, hits = 0;
      for (uint32_t i0 = 0; i0 < M; i0 += WAVE) {
        const uint32_t i = i0 + lane;
        bool keep = false;
        uint4 me = make_uint4(0, 0, 0, 0);
        if (i < M) {
          me = rec[i];
          keep = true;
          if (i > 0) {
            const uint4 pv = rec[i - 1];
            keep = !(pv.x == me.x && pv.y == me.y && pv.z == me.z);
          }
        }
        const uint64_t kb = __ballot(keep);
        const uint32_t cnt = keep ? (me.w - me.z + 1u) : 0u;
        if (keep) base[n_out + lanes_below(kb)] = make_uint4(me.x, me.y, me.z, cnt);
        n_out += __popcll(kb);
        uint32_t s = cnt;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        hits += s;
      }
      if (lane == 0) {
        a.nmatch[g] = n_out;
        a.nhits[g] = hits;
      }
      total_out += n_out;
    }
  }
  (void)s_nout;
  (void)s_hits;
  if (tid == 0 && total_out) atomicAdd(&a.stats[2], (unsigned long long)total_out);
}

/* ---- exclusive scan of nhits (uint32) into uint64 offsets ------------------- */
#define SCAN_BLOCK 1024
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partial(const uint32_t *in, uint64_t *blocksum,
                                                             uint32_t n) {
  __shared__ unsigned long long s[SCAN_BLOCK / WAVE];
  const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  unsigned long long v = i < n ? in[i] : 0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane_id() == 0) s[threadIdx.x / WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int j = 0; j < SCAN_BLOCK / WAVE; j++) t += s[j];
    blocksum[blockIdx.x] = t;
  }
}
/* single block: exclusive scan of the block sums in place, total to blocksum[nb] */
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_blocksums(uint64_t *blocksum, uint32_t nb) {
  __shared__ unsigned long long s[SCAN_BLOCK];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    const unsigned long long v = i < nb ? blocksum[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < SCAN_BLOCK; o <<= 1) {
      unsigned long long add = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) blocksum[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += s[SCAN_BLOCK - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) blocksum[nb] = carry;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_final(const uint32_t *in, const uint64_t *blocksum,
                                                           uint64_t *out, uint32_t n, uint32_t nb) {
  __shared__ unsigned long long s[SCAN_BLOCK];
  const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const unsigned long long v = i < n ? in[i] : 0;
  s[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t o = 1; o < SCAN_BLOCK; o <<= 1) {
    unsigned long long add = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  if (i < n) out[i] = blocksum[blockIdx.x] + s[threadIdx.x] - v;
  if (i == 0) out[n] = blocksum[nb];
}

/* ---- locate: SA gather + coordinate rule ----------------------------------- */
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

/* one wavefront per guide; dynamic LDS: (2*cap + 1) uint32 exclusive prefix of match sizes */
__global__ __launch_bounds__(WAVE) void k_locate(gs_locate_args a) {
  extern __shared__ uint32_t s_pre[];
  const uint32_t g = blockIdx.x;
  const uint32_t lane = lane_id();
  if (g >= a.n) return;
  const uint32_t M = a.nmatch[g];
  if (M == 0) return;
  const uint4 *mt = a.matches + (size_t)g * 2 * a.cap;
  uint32_t run = 0;
  for (uint32_t i0 = 0; i0 < M; i0 += WAVE) {
    const uint32_t i = i0 + lane;
    const uint32_t c = i < M ? mt[i].w : 0;
    uint32_t inc = c; /* inclusive wave scan */
    for (int o = 1; o < WAVE; o <<= 1) {
      const uint32_t up = __shfl_up(inc, o);
      if ((int)lane >= o) inc += up;
    }
    if (i < M) s_pre[i] = run + inc - c;
    run += __shfl(inc, WAVE - 1);
  }
  if (lane == 0) s_pre[M] = run;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t H = run;
  gs_hit *out = a.hits + a.offsets[a.gmap ? a.gmap[g] : g];
  for (uint32_t h = lane; h < H; h += WAVE) {
    /* last match j with s_pre[j] <= h */
    uint32_t lo = 0, hi = M;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_pre[mid] <= h)
        lo = mid;
      else
        hi = mid;
    }
    const uint4 m = mt[lo];
    const uint64_t key = ((uint64_t)m.y << 32) | m.x;
    const uint32_t strand = (uint32_t)(key >> 60) & 1u;
    const uint32_t row = m.z + (h - s_pre[lo]);
    const uint64_t sa = (uint64_t)a.sd[strand].sa[row] - ((key & 1ull) ? a.v_rem : 0u);
    gs_hit o;
    /* process.hpp:104 / :111 */
    o.pos = strand == 0 ? -(int64_t)sa : (int64_t)(a.genome_length - (sa + 1ull));
    o.key = key & ~1ull;
    out[h] = o;
  }
}

/* ---- unit kernels ----------------------------------------------------------- */
__global__ void k_rank4(gs_strand_dev sd, const uint64_t *rows, uint64_t n, uint64_t *out) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t a, c, g, t;
  const uint32_t i = (uint32_t)rows[j];
  occ4(sd.blocks, i >> GS_BLOCK_SHIFT, i & (GS_BLOCK_ROWS - 1u), a, c, g, t);
  out[4 * j + 0] = a;
  out[4 * j + 1] = c;
  out[4 * j + 2] = g;
  out[4 * j + 3] = t;
}
__global__ void k_resolve(gs_strand_dev sd, const uint64_t *rows, uint64_t n, uint64_t *out) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  out[j] = sd.sa[rows[j]];
}

/* ---- selective redo of guides whose matches overflowed their slots -------------------- */
__global__ void k_collect_overflow(const uint32_t *counts, uint32_t n, uint32_t cap, uint32_t *list,
                                   uint32_t *n_list) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  if (counts[2 * g] > cap || counts[2 * g + 1] > cap) list[atomicAdd(n_list, 1u)] = g;
}
__global__ void k_gather_guides(const gs_guide_rec *in, const uint32_t *list, uint32_t n_o,
                                gs_guide_rec *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_o) out[i] = in[list[i]];
}
__global__ void k_gather_counts(const uint32_t *counts, const uint32_t *list, uint32_t n_o, uint32_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_o) {
    out[2 * i] = counts[2 * list[i]];
    out[2 * i + 1] = counts[2 * list[i] + 1];
  }
}
__global__ void k_patch_overflow(const uint32_t *list, uint32_t n_o, const uint32_t *nhits2,
                                 uint32_t *nhits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_o) nhits[list[i]] = nhits2[i];
}

/* ---- overflow arena -> contiguous records (instead of a second search pass) ------------------------
 * The guides on the overflow list have their first `cap` records per item in the main slot array and the
 * rest in arena chunks (gs_search_args::arena).  Item j of the list (2 * position + strand) is copied to
 * dst at dst_off[j] (exact-size layout) or at j * cap2 (fixed stride): blocks [0, 2 n_o) copy the slot
 * parts, one block per chunk the rest. */
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
__global__ __launch_bounds__(256) void k_arena_gather(gs_agather_args a) {
  const uint32_t b = blockIdx.x;
  const uint4 *src;
  uint32_t j, first, nrec;
  if (b < 2u * a.n_o) {
    j = b;
    const uint32_t item = 2u * a.list[j >> 1] + (j & 1u);
    const uint32_t c = a.counts[item];
    src = a.slots + (size_t)item * a.cap;
    first = 0;
    nrec = c < a.cap ? c : a.cap;
  } else {
    const uint32_t c = b - 2u * a.n_o;
    if (c >= a.n_used) return;
    if (a.chunk_seq[c] == 0xFFFFFFFFu) return; /* emptied when a shared item's gaps were closed (k_share_fix) */
    const uint32_t item = a.chunk_item[c];
    const uint32_t pos = a.redo_pos[item >> 1];
    if (pos == 0xFFFFFFFFu) return; /* cannot happen: an item with chunks overflowed its slots */
    j = 2u * pos + (item & 1u);
    const uint32_t cnt = a.counts[item], e0 = a.chunk_seq[c] << ARENA_SHIFT;
    if (cnt <= a.cap + e0) return;
    src = a.arena + ((size_t)c << ARENA_SHIFT);
    first = a.cap + e0;
    nrec = cnt - first < ARENA_CHUNK ? cnt - first : ARENA_CHUNK;
  }
  uint4 *dst = a.dst + (a.dst_off ? (size_t)a.dst_off[j] : (size_t)j * a.cap2) + first;
  for (uint32_t i = threadIdx.x; i < nrec; i += blockDim.x) dst[i] = src[i];
}

/* ---- shared items (gs_search_args::shq): the gaps their helpers left are closed -----------------------------
 * Behind k_search a shared item's records lie in its slots (the owner's first `cap`), the owner's chunks (full but
 * the last) and the helpers' chunks (each episode's last one partly filled).  Everything downstream reads an item as
 * "slots, then chunks 0, 1, .. in order, all full but the last": k_share_fix moves the records that lie beyond the
 * item's total into the holes before it (their order inside an item means nothing: the ordering kernels sort by
 * (sequence, row)), renumbers the chunks, drops the emptied ones (chunk_seq = 0xFFFFFFFF) and adds the helpers' counts. */
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
__global__ __launch_bounds__(1024) void k_share_scan(gs_share_args a) {
  __shared__ uint32_t s_w[16], s_carry;
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u), w = tid / WAVE;
  const uint32_t n_sh = a.ctl[96] < a.sh_max ? a.ctl[96] : a.sh_max;
  if (tid == 0) s_carry = 0u;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n_sh + 1u; i0 += 1024u) {
    const uint32_t sid = i0 + tid;
    uint32_t v = 0;
    if (sid < n_sh) v = a.nchunk[a.sh_list[sid]].x + a.sh_acc[16u * sid + 1u];
    const uint32_t incl = wave_incl_sum(v);
    if (lane == WAVE - 1u) s_w[w] = incl;
    __syncthreads();
    uint32_t b = s_carry;
    for (uint32_t u = 0; u < w; ++u) b += s_w[u];
    if (sid <= n_sh) a.dbase[sid] = b + incl - v;
    __syncthreads();
    if (tid == 1023u) s_carry = b + incl;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_share_dir(gs_share_args a) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_used = *a.arena_next < a.arena_chunks ? *a.arena_next : a.arena_chunks;
  if (c >= n_used || a.chunk_seq[c] == 0xFFFFFFFFu) return; /* (reserved by a wave and never used) */
  const uint32_t slot = a.chunk_item[c];
  const uint2 nc = a.nchunk[slot];
  if (!(nc.y >> 31)) return;
  const uint32_t sid = nc.y & 0x7FFFFFFFu;
  const uint32_t n_sh = a.ctl[96] < a.sh_max ? a.ctl[96] : a.sh_max;
  if (sid >= n_sh) return;
  const uint32_t seq = a.chunk_seq[c];
  const uint32_t j = (seq & SH_HELPER_SEQ) ? nc.x + (seq & (SH_HELPER_SEQ - 1u)) : seq;
  const uint32_t d0 = a.dbase[sid];
  if (j < a.dbase[sid + 1u] - d0) a.dir[d0 + j] = c;
}
__global__ __launch_bounds__(256) void k_share_fix(gs_share_args a) {
  /* segment 0 = the slots, segment 1 + j = chunk j of the directory */
  __shared__ uint32_t s_fill[SH_MAXSEG + 1u], s_hole[SH_MAXSEG + 2u], s_mov[SH_MAXSEG + 2u];
  __shared__ uint32_t s_red[3][4], s_tot[3];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u), w = tid / WAVE;
  const uint32_t n_sh = a.ctl[96] < a.sh_max ? a.ctl[96] : a.sh_max;
  for (uint32_t sid = blockIdx.x; sid < n_sh; sid += gridDim.x) {
    const uint32_t slot = a.sh_list[sid];
    const uint32_t own = a.counts[slot], H = a.sh_acc[16u * sid], nho = a.nchunk[slot].x;
    const uint32_t d0 = a.dbase[sid], ns = a.dbase[sid + 1u] - d0;
    const uint32_t *dir = a.dir + d0;
    const uint32_t T = own + H, cap = a.cap;
    const bool own_short = own > cap && own - cap > (nho << ARENA_SHIFT);
    const uint32_t nseg = ns + 1u;
    __syncthreads(); /* (the previous item's tables are no longer read) */
    if (ns > SH_MAXSEG || own_short) {
      /* not in a state to be closed up (or the arena ran out under the owner): the total is exact, the host searches the
       * batch's overflowing guides again - this item among them */
      if (tid == 0) {
        a.counts[slot] = T > cap ? T : cap + 1u;
        atomicAdd(&a.stats[6], 1ull);
        atomicAdd(&a.stats[1], 1ull);
      }
      continue;
    }
    /* what each segment holds */
    uint32_t v_sum = 0;
    for (uint32_t s = tid; s < nseg; s += 256u) {
      uint32_t f;
      if (s == 0u)
        f = own < cap ? own : cap;
      else if (s - 1u < nho)
        f = s < nho ? ARENA_CHUNK : own - cap - ((nho - 1u) << ARENA_SHIFT);
      else
        f = a.chunk_fill[dir[s - 1u]];
      if (f > ARENA_CHUNK && s != 0u) f = ARENA_CHUNK;
      s_fill[s] = f;
      v_sum += f;
    }
    for (int o = 32; o > 0; o >>= 1) v_sum += (uint32_t)__shfl_xor((int)v_sum, o);
    if (lane == 0) s_red[0][w] = v_sum;
    __syncthreads();
    const uint32_t V = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    if (V != T) { /* a helper was short of chunks: as above */
      if (tid == 0) {
        a.counts[slot] = T > cap ? T : cap + 1u;
        atomicAdd(&a.stats[6], 1ull);
        atomicAdd(&a.stats[1], 1ull);
      }
      continue;
    }
    /* holes before place T and records at or beyond it, per segment; thread t takes a run of per segments */
    const uint32_t per = (nseg + 255u) / 256u, s_lo = tid * per, s_hi = s_lo + per < nseg ? s_lo + per : nseg;
    uint32_t h_sum = 0, m_sum = 0;
    for (uint32_t s = s_lo; s < s_hi; ++s) {
      const uint32_t start = s == 0u ? 0u : cap + ((s - 1u) << ARENA_SHIFT), room = s == 0u ? cap : ARENA_CHUNK, f = s_fill[s];
      const uint32_t in = T > start ? (T - start < room ? T - start : room) : 0u; /* places of the segment before T */
      const uint32_t hole = in > f ? in - f : 0u, mov = f > in ? f - in : 0u;
      s_hole[s] = hole;
      s_mov[s] = mov;
      h_sum += hole;
      m_sum += mov;
    }
    const uint32_t hi = wave_incl_sum(h_sum), mi = wave_incl_sum(m_sum);
    if (lane == WAVE - 1u) {
      s_red[1][w] = hi;
      s_red[2][w] = mi;
    }
    __syncthreads();
    uint32_t hb = hi - h_sum, mb = mi - m_sum;
    for (uint32_t u = 0; u < w; ++u) {
      hb += s_red[1][u];
      mb += s_red[2][u];
    }
    if (tid == 255u) {
      s_tot[1] = hb + h_sum;
      s_tot[2] = mb + m_sum;
    }
    for (uint32_t s = s_lo; s < s_hi; ++s) { /* exclusive prefixes in place */
      const uint32_t h = s_hole[s], m = s_mov[s];
      s_hole[s] = hb;
      s_mov[s] = mb;
      hb += h;
      mb += m;
    }
    __syncthreads();
    const uint32_t M = s_tot[2];
    if (tid == 0) {
      s_hole[nseg] = s_tot[1];
      s_mov[nseg] = M;
    }
    __syncthreads();
    if (s_tot[1] == M) {
      for (uint32_t r = tid; r < M; r += 256u) {
        /* mover r: the last segment whose prefix is <= r (segments without movers share a prefix with their successor) */
        uint32_t lo = 0, hi2 = nseg;
        while (hi2 - lo > 1u) {
          const uint32_t mid = (lo + hi2) >> 1;
          if (s_mov[mid] <= r) lo = mid; else hi2 = mid;
        }
        const uint32_t sm = lo, fm = s_fill[sm];
        const uint32_t startm = cap + ((sm - 1u) << ARENA_SHIFT); /* (segment 0 holds no mover unless T = 0: then M = 0) */
        const uint32_t inm = T > startm ? (T - startm < fm ? T - startm : fm) : 0u;
        const uint4 *src = a.arena + (((size_t)dir[sm - 1u] << ARENA_SHIFT) + inm + (r - s_mov[sm]));
        lo = 0, hi2 = nseg;
        while (hi2 - lo > 1u) {
          const uint32_t mid = (lo + hi2) >> 1;
          if (s_hole[mid] <= r) lo = mid; else hi2 = mid;
        }
        const uint32_t sh = lo, off = s_fill[sh] + (r - s_hole[sh]);
        uint4 *dst = sh == 0u ? a.slots + ((size_t)slot * cap + off) : a.arena + (((size_t)dir[sh - 1u] << ARENA_SHIFT) + off);
        *dst = *src;
      }
    }
    const uint32_t nf = T > cap ? (T - cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT : 0u;
    for (uint32_t j = tid; j < ns; j += 256u) a.chunk_seq[dir[j]] = j < nf ? j : 0xFFFFFFFFu;
    if (tid < 8u) a.cls[(size_t)slot * 8u + tid] += a.sh_acc[16u * sid + 8u + tid];
    if (tid == 0) {
      a.counts[slot] = T;
      a.nchunk[slot] = make_uint2(nf, nf ? dir[nf - 1u] : 0u);
      if (own <= cap && T > cap) atomicAdd(&a.stats[1], 1ull);
      if (s_tot[1] != M) atomicAdd(&a.stats[6], 1ull); /* (cannot happen: V = T makes holes and movers equal) */
    }
  }
}

/* ---- guides with more matches than an LDS sort can hold: repeat-derived guides at any budget,
 * every guide at <= 6 mismatches on a genome of this size (~5,400 matches per item).  Their match
 * records are compacted into one array (item order = guide order), ordered by two stable
 * device-wide radix sorts - (key low bits, first row), then (guide, key high bits) - made unique,
 * scanned, and located one thread per record.  No per-guide atomics, no comparator sort. ---- */
struct gs_big_src {   /* where the records of one set item live */
  uint64_t off;       /* element offset */
  uint32_t alt;       /* 0: main slot array, 1: the exact-size redo array */
};
/* one workgroup per set item: copy its records to the compact array and build the two sort words */
__global__ __launch_bounds__(256) void k_big_compact(const uint4 *slots_main, const uint4 *slots_alt,
                                                     const gs_big_src *src, const unsigned long long *prefix,
                                                     uint32_t n_items, uint4 *recs, unsigned long long *w0,
                                                     unsigned long long *w1, uint32_t *idx) {
  const uint32_t item = blockIdx.x;
  if (item >= n_items) return;
  const unsigned long long b = prefix[item], e = prefix[item + 1];
  const gs_big_src s = src[item];
  const uint4 *in = (s.alt ? slots_alt : slots_main) + s.off;
  const unsigned long long g = item >> 1;
  for (unsigned long long r = b + threadIdx.x; r < e; r += blockDim.x) {
    const uint4 v = in[r - b];
    recs[r] = v;
    w0[r] = ((unsigned long long)(v.x >> 8) << 32) | v.z; /* key bits 31:8, first row */
    w1[r] = (g << 32) | v.y;                                /* guide, key bits 63:32 */
    idx[r] = (uint32_t)r;
  }
}
__global__ void k_big_gather_w1(const unsigned long long *w1, const uint32_t *idx, uint64_t T,
                                unsigned long long *out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < T) out[r] = w1[idx[r]];
}
/* keep[r] = 1 when sorted record r starts a new (guide, key, first row); rows[r] = its row count */
__global__ void k_big_flags(const uint4 *recs, const uint32_t *idx, const unsigned long long *w1s, uint64_t T,
                            uint32_t *keep, unsigned long long *rows) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T) return;
  const uint4 c = recs[idx[r]];
  bool f = true;
  if (r > 0 && (w1s[r - 1] >> 32) == (w1s[r] >> 32)) {
    const uint4 p = recs[idx[r - 1]];
    f = !(p.x == c.x && p.y == c.y && p.z == c.z);
  }
  keep[r] = f ? 1u : 0u;
  rows[r] = f ? (unsigned long long)(c.w - c.z + 1u) : 0ull;
}
/* per guide of the set: unique matches and hits from the two scans (guide g owns the sorted
 * positions [prefix[2g], prefix[2g+2]): the compact array is in guide order and the sort keeps it) */
__global__ void k_big_totals(const unsigned long long *prefix, const uint32_t *keep_scan,
                             const unsigned long long *row_scan, uint32_t n_set, uint32_t *nmatch,
                             uint32_t *nhits, uint32_t *err) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_set) return;
  const unsigned long long b = prefix[2 * g], e = prefix[2 * g + 2];
  nmatch[g] = keep_scan[e] - keep_scan[b];
  const unsigned long long h = row_scan[e] - row_scan[b];
  if (h >> 32) atomicOr(err, 1u); /* more than 2^32 hits for one guide */
  nhits[g] = (uint32_t)h;
}
struct gs_blocate2_args {
  gs_strand_dev sd[2];
  const uint4 *recs;
  const uint32_t *idx;
  const unsigned long long *w1s;
  const uint32_t *keep;
  const unsigned long long *row_scan;
  const unsigned long long *prefix;
  const uint32_t *gmap; /* set position -> guide of the batch (nullptr: identity) */
  const uint64_t *offsets;
  gs_hit *hits;
  uint64_t genome_length, T;
  uint32_t v_rem;
};
__global__ void k_big_locate(gs_blocate2_args a) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.T || !a.keep[r]) return;
  const uint4 m = a.recs[a.idx[r]];
  const uint32_t g = (uint32_t)(a.w1s[r] >> 32);
  const uint64_t key = ((uint64_t)m.y << 32) | m.x;
  const uint32_t strand = (uint32_t)(key >> 60) & 1u;
  gs_hit *out = a.hits + a.offsets[a.gmap ? a.gmap[g] : g] + (a.row_scan[r] - a.row_scan[a.prefix[2 * g]]);
  const uint32_t cnt = m.w - m.z + 1u;
  for (uint32_t h = 0; h < cnt; ++h) {
    const uint64_t sa = (uint64_t)a.sd[strand].sa[m.z + h] - ((key & 1ull) ? a.v_rem : 0u);
    gs_hit o;
    o.pos = strand == 0 ? -(int64_t)sa : (int64_t)(a.genome_length - (sa + 1ull));
    o.key = key & ~1ull;
    out[h] = o;
  }
}

/* ---- the same ordering with ONE radix sort (the form that runs whenever its sort word fits 64 bits) ----
 * What orders a guide's records is (mismatches, index, match.sequence, row).  match.sequence travels in
 * the key as per-position codes (52 bits), but among the sequences with j substitutions in L positions
 * and P PAM symbols there are only C(L,j) 3^j 5^P of them: their lexicographic RANK (combinatorial number
 * system, position 0 most significant as in the key) orders them exactly and needs 22 bits at L = 20,
 * j <= 3, P = 3 instead of 52.  Sort word W = guide of the set | (mismatches, index, rank) as one number -
 * the class's base (gs_big2_tab::base: all sequences of the classes before it) + rank: 36 bits for 8 k guides.  Two stable sorts: by first row (32-bit keys, four passes over 8-byte pairs), then by W
 * (five passes over 12-byte pairs) - nine passes and 184 bytes moved per record where sorting the raw key
 * took thirteen passes and 312 bytes.  (Sorting by W alone and ordering the rows inside each run of equal W
 * afterwards was tried: on a repeat-rich genome a third of the records sit in runs of 10^4 and more - the
 * family's consensus sequence - and the run-by-run passes cost more than the row sort does.) */
struct gs_big2_tab {
  unsigned long long n[32][8]; /* n[a][r] = C(a, r) 3^r: sequences of a positions with r substitutions */
  /* base[mismatches << 1 | index]: the sequences that go before the class's first - every sequence with fewer
   * mismatches on either index, and the class's own on index 0: (mismatches, index, rank) as ONE number, three
   * bits narrower than the three fields side by side (a radix pass less at m = 5 and 6) */
  unsigned long long base[16];
};
__device__ __forceinline__ unsigned long long big2_rank(const unsigned long long key, const uint32_t L, const uint32_t P,
                                                         const unsigned long long *nt /* [32][8] in LDS */,
                                                         const unsigned long long pam_mul) {
  const unsigned long long path = key >> 8;
  uint32_t j = 0;
  for (uint32_t t = 0; t < L; t++) j += ((path >> (50u - 2u * t)) & 3ull) != 0ull;
  if (j > 7u) j = 7u;
  uint32_t r = j;
  unsigned long long rank = 0;
  for (uint32_t t = 0; t < L && r != 0u; t++) {
    const uint32_t c = (uint32_t)(path >> (50u - 2u * t)) & 3u;
    if (c) {
      const uint32_t a = L - 1u - t; /* positions behind this one */
      /* smaller sequences with the same prefix: a 0 here (r substitutions behind), or one of the c-1 lower codes */
      rank += nt[a * 8u + r] + (unsigned long long)(c - 1u) * nt[a * 8u + r - 1u];
      r--;
    }
  }
  unsigned long long pr = 0;
  for (uint32_t u = 0; u < P; u++) {
    const uint32_t c = (uint32_t)(path >> (49u - 2u * L - 3u * u)) & 7u;
    pr = pr * 5ull + (c < 4u ? c : 4u);
  }
  return rank * pam_mul + pr;
}
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
/* one workgroup per set item: copy its records to the compact array and build their sort words */
__global__ __launch_bounds__(256) void k_big2_compact(gs_big2_compact_args a) {
  __shared__ unsigned long long nt[32 * 8];
  __shared__ unsigned long long bs[16];
  for (uint32_t i = threadIdx.x; i < 32u * 8u; i += blockDim.x) nt[i] = a.tab->n[i >> 3][i & 7u];
  if (threadIdx.x < 16u) bs[threadIdx.x] = a.tab->base[threadIdx.x];
  __syncthreads();
  const uint4 *in;
  unsigned long long b, e, g;
  if (!a.from_arena) {
    const uint32_t item = blockIdx.x;
    if (item >= a.n_items) return;
    b = a.prefix[item];
    e = a.prefix[item + 1];
    const gs_big_src s = a.src[item];
    in = (s.alt ? a.slots_alt : a.slots_main) + s.off;
    g = item >> 1;
  } else if (blockIdx.x < a.n_items) {
    const uint32_t sb = blockIdx.x; /* item of the set -> item of the batch */
    const uint32_t item = a.list ? 2u * a.list[sb >> 1] + (sb & 1u) : sb;
    const uint32_t c = a.counts[item];
    in = a.slots_main + (size_t)item * a.cap;
    b = a.prefix[sb];
    e = b + (c < a.cap ? c : a.cap);
    g = sb >> 1;
  } else {
    const uint32_t c = blockIdx.x - a.n_items;
    if (c >= a.n_used || a.chunk_seq[c] == 0xFFFFFFFFu) return; /* (emptied by k_share_fix) */
    const uint32_t item = a.chunk_item[c];
    uint32_t sb = item;
    if (a.list) {
      const uint32_t pos = a.redo_pos[item >> 1];
      if (pos == 0xFFFFFFFFu) return;
      sb = 2u * pos + (item & 1u);
    }
    const uint32_t cnt = a.counts[item], e0 = a.chunk_seq[c] << ARENA_SHIFT;
    if (cnt <= a.cap + e0) return;
    in = a.arena + ((size_t)c << ARENA_SHIFT);
    b = a.prefix[sb] + a.cap + e0;
    const uint32_t left = cnt - a.cap - e0;
    e = b + (left < ARENA_CHUNK ? left : ARENA_CHUNK);
    g = sb >> 1;
  }
  for (unsigned long long r = b + threadIdx.x; r < e; r += blockDim.x) {
    const uint4 v = in[r - b];
    const unsigned long long key = ((unsigned long long)v.y << 32) | v.x;
    a.recs[r] = v;
    const unsigned long long w = (g << (4u + a.rbits)) | (bs[(uint32_t)(key >> 60) & 15u] + big2_rank(key, a.L, a.P, nt, a.pam_mul));
    if (a.row_bits) {
      a.W[r] = (w << a.row_bits) | (((unsigned long long)v.z + a.row_off) & ((1ull << a.row_bits) - 1ull));
    } else {
      a.W[r] = w;
      a.rowkey[r] = v.z;
    }
    a.idx[r] = (uint32_t)r;
  }
}
/* records per item of the set when they are read from the slots and the arena (the main pass counted exactly) */
__global__ void k_big2_counts(const uint32_t *counts, const uint32_t *list, uint32_t n_items, unsigned long long *cnt64) {
  const uint32_t sb = blockIdx.x * blockDim.x + threadIdx.x;
  if (sb >= n_items) return;
  cnt64[sb] = counts[list ? 2u * list[sb >> 1] + (sb & 1u) : sb];
}
__global__ void k_big2_gather(const uint4 *recs, const uint32_t *idx, uint64_t T, uint4 *out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < T) out[r] = recs[idx[r]];
}
__global__ void k_big2_gather_w(const unsigned long long *W, const uint32_t *idx, uint64_t T, unsigned long long *out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < T) out[r] = W[idx[r]];
}
/* After a sort by W alone: put the rows inside each run of equal W in order.  Nearly every run holds one
 * record (a sequence found at one row: every record of a batch on a repeat-free genome), a few hold several
 * (the same sequence at several rows): thread r finds its run by looking at most `short_max` words either
 * way, counts the records that go before it (smaller first row, ties by position) and stores its record's
 * position there.  A run longer than that raises *long_run: the batch is then ordered by the two sorts
 * (rows, then W) instead, and so are the handle's later batches - a repeat-rich genome has runs of 10^4. */
__global__ __launch_bounds__(256) void k_big2_runs(const unsigned long long *W, const uint32_t *idx_in, const uint4 *recs,
                                                   uint64_t T, uint32_t short_max, uint32_t *idx_out, uint32_t *long_run) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T) return;
  const unsigned long long w = W[r];
  const uint32_t mine = idx_in[r];
  if (!(r > 0 && W[r - 1] == w) && !(r + 1 < T && W[r + 1] == w)) { /* a run of one */
    idx_out[r] = mine;
    return;
  }
  uint64_t s = r, e = r + 1;
  while (s > 0 && r - s < short_max && W[s - 1] == w) s--;
  while (e < T && e - r < short_max && W[e] == w) e++;
  if ((s > 0 && W[s - 1] == w) || (e < T && W[e] == w) || e - s > short_max) {
    *long_run = 1u;
    idx_out[r] = mine;
    return;
  }
  const uint32_t myrow = recs[mine].z;
  uint64_t rank = 0;
  for (uint64_t j = s; j < e; j++) {
    const uint32_t z = recs[idx_in[j]].z;
    rank += (z < myrow || (z == myrow && j < r)) ? 1u : 0u;
  }
  idx_out[s + rank] = mine;
}
/* largest of n 64-bit counts (a grid-stride loop, one atomic per wave) */
__global__ void k_max_u64(const unsigned long long *v, uint32_t n, unsigned long long *out) {
  unsigned long long mx = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    mx = v[i] > mx ? v[i] : mx;
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long x = __shfl_xor(mx, o);
    mx = x > mx ? x : mx;
  }
  if (lane_id() == 0 && mx) atomicMax(out, mx);
}
/* the composite word of records whose plain words and first rows are already there (the batch that shows a
 * handle its first long run) */
__global__ void k_big2_comp(const unsigned long long *W, const uint32_t *rowkey, uint64_t T, uint32_t row_bits,
                            uint32_t row_off, unsigned long long *Wc) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T) return;
  Wc[r] = (W[r] << row_bits) | (((unsigned long long)rowkey[r] + row_off) & ((1ull << row_bits) - 1ull));
}
/* After ONE sort by (word << row_bits | low row_bits bits of the row): inside a run of equal words the rows are in
 * order by their low bits.  They are scattered over the suffix array interval of the run's k-mer (this strand's
 * hits carry the row of the suffix v_rem symbols into the site), so wherever that interval reaches across a
 * multiple of 2^row_bits the run is out of order: the full rows show a descent.  Every descent goes on a list;
 * k_big2_fixruns then orders each such run by (row >> row_bits), stably. */
__global__ void k_big2_wraps(const uint4 *S2, const unsigned long long *Wc, uint64_t T, uint32_t row_bits, uint32_t *list,
                             uint32_t *n_list) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0 || r >= T) return;
  if ((Wc[r] >> row_bits) == (Wc[r - 1] >> row_bits) && S2[r].z < S2[r - 1].z) list[atomicAdd(n_list, 1u)] = (uint32_t)r;
}
/* One workgroup per listed descent; the first to claim the run (claimed[] = zeros) puts it in order: the records
 * are in order by the low bits of row + row_off, so a stable partition by the high part - one pass per value it
 * takes between its least and its greatest, two nearly always - finishes the job.  tmp = the unordered records'
 * array (read for the last time by the gather), used at the run's own positions. */
__global__ __launch_bounds__(256) void k_big2_fixruns(uint4 *S2, uint4 *tmp, const unsigned long long *Wc, uint64_t T,
                                                      uint32_t row_bits, uint32_t row_off, const uint32_t *list, uint32_t n,
                                                      uint32_t *claimed) {
  __shared__ unsigned long long s_b[2];
  __shared__ uint32_t s_take, s_lo, s_hi, s_w[4];
  const uint32_t tid = threadIdx.x, wave = tid / WAVE, lane = lane_id();
  for (uint32_t d = blockIdx.x; d < n; d += gridDim.x) {
    const uint64_t s = list[d];
    if (tid == 0) {
      const unsigned long long w = Wc[s] >> row_bits;
      uint64_t lo = 0, hi = s; /* first position of the word: in [0, s] */
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((Wc[mid] >> row_bits) < w)
          lo = mid + 1;
        else
          hi = mid;
      }
      s_b[0] = lo;
      s_take = atomicExch(&claimed[lo], 1u) == 0u ? 1u : 0u;
      lo = s + 1; /* one past its last position: in (s, T] */
      hi = T;
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((Wc[mid] >> row_bits) <= w)
          lo = mid + 1;
        else
          hi = mid;
      }
      s_b[1] = lo;
      s_lo = 0xFFFFFFFFu;
      s_hi = 0u;
    }
    __syncthreads();
    const uint64_t start = s_b[0], end = s_b[1], len = end - start;
    const bool mine = s_take != 0u;
    __syncthreads(); /* (thread 0 writes these again in the next round) */
    if (!mine) continue; /* wave-uniform and workgroup-uniform: another workgroup has the run */
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint64_t i = tid; i < len; i += blockDim.x) {
      const uint32_t h = (uint32_t)(((unsigned long long)S2[start + i].z + row_off) >> row_bits);
      lo = h < lo ? h : lo;
      hi = h > hi ? h : hi;
    }
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
    __syncthreads();
    const uint32_t hmin = s_lo, hmax = s_hi;
    __syncthreads(); /* every thread has read them: thread 0 writes s_lo again as the loop's first statement */
    uint64_t base = 0;
    /* one pass per value the high part TAKES (the next one is found during the pass), not per integer between
     * the least and the greatest: with few row bits a run's rows can span thousands of multiples of 2^row_bits */
    for (uint32_t v = hmin;;) {
      if (tid == 0) s_lo = 0xFFFFFFFFu; /* least high part above v */
      __syncthreads();
      uint32_t nxt = 0xFFFFFFFFu;
      for (uint64_t c = 0; c < len; c += blockDim.x) {
        const uint64_t i = c + tid;
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        bool f = false;
        if (i < len) {
          rec = S2[start + i];
          const uint32_t h = (uint32_t)(((unsigned long long)rec.z + row_off) >> row_bits);
          f = h == v;
          if (h > v && h < nxt) nxt = h;
        }
        const uint64_t b = __ballot(f);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t q = 0; q < blockDim.x / WAVE; ++q) {
          if (q < wave) before += s_w[q];
          total += s_w[q];
        }
        if (f) tmp[start + base + before + lanes_below(b)] = rec;
        base += total;
        __syncthreads();
      }
      if (nxt != 0xFFFFFFFFu) atomicMin(&s_lo, nxt);
      __syncthreads();
      const uint32_t nv = s_lo;
      __syncthreads();
      if (nv == 0xFFFFFFFFu || v == hmax) break;
      v = nv;
    }
    __threadfence();
    __syncthreads();
    for (uint64_t i = tid; i < len; i += blockDim.x) S2[start + i] = tmp[start + i];
    __syncthreads();
  }
}
__global__ void k_iota_u32(uint32_t *p, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}
/* keep[r] = 1 when ordered record r starts a new (guide, key, first row); rows[r] = its row count */
__global__ void k_big2_flags(const uint4 *S2, const unsigned long long *W, uint64_t T, uint32_t *keep,
                             unsigned long long *rows, uint32_t wshift) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T) return;
  const uint4 c = S2[r];
  bool f = true;
  if (r > 0 && (W[r - 1] >> wshift) == (W[r] >> wshift)) {
    const uint4 p = S2[r - 1];
    f = !(p.x == c.x && p.y == c.y && p.z == c.z);
  }
  keep[r] = f ? 1u : 0u;
  rows[r] = f ? (unsigned long long)(c.w - c.z + 1u) : 0ull;
}
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
__global__ void k_big2_locate(gs_blocate3_args a) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.T || !a.keep[r]) return;
  const uint4 m = a.S2[r];
  const uint32_t g = (uint32_t)(a.W[r] >> a.gshift);
  const uint64_t key = ((uint64_t)m.y << 32) | m.x;
  const uint32_t strand = (uint32_t)(key >> 60) & 1u;
  gs_hit *out = a.hits + a.offsets[a.gmap ? a.gmap[g] : g] + (a.row_scan[r] - a.row_scan[a.prefix[2 * g]]);
  const uint32_t cnt = m.w - m.z + 1u;
  for (uint32_t h = 0; h < cnt; ++h) {
    const uint64_t sa = (uint64_t)a.sd[strand].sa[m.z + h] - ((key & 1ull) ? a.v_rem : 0u);
    gs_hit o;
    o.pos = strand == 0 ? -(int64_t)sa : (int64_t)(a.genome_length - (sa + 1ull));
    o.key = key & ~1ull;
    out[h] = o;
  }
}

/* sources of the set items: the main slot array, or - for guides on the redo list - the exact-size array */
__global__ void k_big_sources(const uint32_t *counts_main, const uint32_t *redo_pos, const uint64_t *slot_off2,
                              const uint32_t *counts2, uint32_t n_items, uint32_t cap, gs_big_src *src,
                              unsigned long long *cnt64) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= n_items) return;
  gs_big_src s;
  uint32_t c;
  /* no main array: the set IS the redo list, set guide j = redo position j */
  const uint32_t rp = counts_main ? (redo_pos ? redo_pos[item >> 1] : 0xFFFFFFFFu) : (item >> 1);
  if (rp != 0xFFFFFFFFu) {
    const uint32_t it2 = 2u * rp + (item & 1u);
    s.off = slot_off2[it2];
    s.alt = 1u;
    c = counts2[it2];
  } else {
    s.off = (uint64_t)item * cap;
    s.alt = 0u;
    c = counts_main[item];
  }
  src[item] = s;
  cnt64[item] = c;
}
__global__ void k_fill_u32(uint32_t *p, uint32_t v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_mark_redo(const uint32_t *list, uint32_t n_o, uint32_t *redo_pos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_o) redo_pos[list[i]] = i;
}
/* hits of a guide BEFORE the per-distance sets drop duplicate sequences: what the reference's
 * threshold filter counts (off_target_counter, process.hpp:25-27: ep - sp + 1 per callback, one
 * callback per PAM pattern that matches).  One wavefront per guide over its raw match records;
 * a guide whose matches overflowed its slots has far more than the filter's bound: saturated. */
__global__ __launch_bounds__(256) void k_raw_counts(const uint4 *slots, const uint32_t *counts, uint32_t n, uint32_t cap,
                                                    uint32_t *raw) {
  const uint32_t g = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE, lane = lane_id();
  if (g >= n) return;
  const uint32_t c0 = counts[2 * g], c1 = counts[2 * g + 1];
  unsigned long long s = 0;
  if (c0 > cap || c1 > cap) {
    s = 0xFFFFFFFFull;
  } else {
    const uint4 *base = slots + (size_t)g * 2 * cap;
    for (uint32_t i = lane; i < c0; i += WAVE) s += base[i].w - base[i].z + 1u;
    for (uint32_t i = lane; i < c1; i += WAVE) s += base[cap + i].w - base[cap + i].z + 1u;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  }
  if (lane == 0) raw[g] = s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s;
}
/* sum and maximum of the per-item match counts (slot sizing of the next batch) */
__global__ void k_count_stats(const uint32_t *counts, uint32_t n_items, unsigned long long *out) {
  /* a grid-stride loop: one pair of atomics per wave of a grid of at most 1,024 workgroups (one per 64 items
   * was 62 k atomics on two words at 2 M items: 0.65 ms of a 23 ms step) */
  unsigned long long v = 0, mx = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long c = counts[i];
    v += c;
    mx = c > mx ? c : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    v += __shfl_xor(v, o);
    const unsigned long long x = __shfl_xor(mx, o);
    mx = x > mx ? x : mx;
  }
  if (lane_id() == 0) {
    if (v) atomicAdd(&out[0], v);
    atomicMax(&out[1], mx);
  }
}
/* arena chunks the items' records beyond their slots take (the exact counts are known even when the arena ran out) */
__global__ void k_need_chunks(const uint32_t *counts, uint32_t n_items, uint32_t cap, uint32_t *out) {
  uint32_t v = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = counts[i];
    if (c > cap) v += (c - cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane_id() == 0 && v) atomicAdd(out, v);
}

/* ---- host side of the pipeline ---------------------------------------------- */
/* slots per (guide, strand) of the first pass.  Up to three mismatches: 64 and the overflow redo
 * takes the tail.  Beyond: from the mean count the previous batch at this budget showed on this
 * index (Poisson-like on a repeat-free genome: mean + 8 sigma), else from the expected count of a
 * uniform genome: sites x sum_k C(L,k) 3^k / 4^L x PAM share.  Whatever does not fit is redone
 * with exact sizes, so a wrong guess costs time, not hits. */
static uint32_t choose_cap(const gs_index *ix, uint32_t m, uint32_t L, uint32_t P, uint32_t n_alt, uint32_t flags) {
  if (m <= 3) return 64;
  double mean = -1, seen_max = 0;
  const uint64_t key = ((uint64_t)L << 32) | ((uint64_t)P << 16) | (n_alt << 8) | (flags & GS_FLAG_PAM_AT_START);
  if (m < 8 && ix->seen_mean[m] >= 0 && ix->seen_key[m] == key) {
    mean = ix->seen_mean[m];
    seen_max = ix->seen_max[m];
  }
  if (mean < 0) {
    double v = 0, c = 1;
    for (uint32_t k = 0; k <= m && k <= L; k++) {
      v += c;
      c = c * 3.0 * (L - k) / (k + 1);
    }
    for (uint32_t i = 0; i < L; i++) v /= 4.0;
    mean = v * (double)ix->strand[0].n * (n_alt + 1) / (P >= 2 ? 16.0 : P == 1 ? 4.0 : 1.0) * 1.3;
  }
  /* counts spread wider than Poisson (base composition of the guide): half again the mean on
   * top, and the largest count the last batch showed unless a repeat-derived guide made it huge */
  double want = 1.5 * mean + 8.0 * sqrt(mean > 1 ? mean : 1) + 64;
  if (seen_max > want) want = seen_max * 1.05 < 3.0 * mean + 64 ? seen_max * 1.05 : 3.0 * mean + 64;
  if (const char *e = gs_opt(ix, "GS_SLOT_CAP")) want = atof(e);
  uint32_t cap = 64;
  while (cap < want && cap < 256) cap <<= 1;
  if (want > 256) cap = (uint32_t)((want + 255) / 256) * 256;
  if (cap > (1u << 20)) cap = 1u << 20;
  return cap;
}

int gs_num_cus(int device) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return 256;
  return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
}

/* Which strand's table finds a site with (a, o, b) substitutions in (X, O, R)?  This strand's
 * table covers X and O: a class of its seeds is a pair (a, o), and verifying such a seed against
 * ctx[] finds every b the budget leaves.  The other strand's table covers O, R and the PAM: its
 * classes are pairs (o, b), each finding every a.  For a fixed o the cells (a, b), a + b <= m - o,
 * must each be covered by row a or by column b; the staircase shape makes every minimal cover
 * "rows a < a*, columns b <= m - o - a*", so the plan is one threshold a*(o) per o.
 * Cost of a class = its seeds x (table line share + chance to survive the context mask x a
 * verification pass), the chances being those of an hg38-sized table (11.5 rows per k-mer); through a
 * PAM-pair table (gs_pairtab.hip) a seed of this strand rarely verifies at all (verify_a ~ 0.2). */
static void gs_choose_astar(uint32_t m, uint32_t nX, uint32_t nO, uint32_t nR, double epam, uint32_t astar[8],
                            double verify_a = 1.5, double verify_b = 1.9) {
  auto binom3 = [](uint32_t n, uint32_t j) -> double { /* C(n, j) 3^j */
    if (j > n) return 0.0;
    double v = 1;
    for (uint32_t i = 0; i < j; i++) v = v * (n - i) / (i + 1) * 3.0;
    return v;
  };
  static const double pass[4] = {0.17, 0.86, 1.0, 1.0};
  auto seed_cost = [&](uint32_t budget_left, double verify) -> double {
    return 0.35 + pass[budget_left < 3 ? budget_left : 3] * verify;
  };
  for (uint32_t o = 0; o < 8; o++) {
    astar[o] = 15;
    if (o > m || o > nO) continue;
    const uint32_t M = m - o;
    double best = -1;
    for (uint32_t as = 0; as <= M + 1; as++) {
      double c = 0;
      for (uint32_t a = 0; a < as && a <= M; a++) c += binom3(nX, a) * seed_cost(M - a, verify_a);
      if (as <= M) {
        if (as > nX) continue; /* the other side would need more substitutions in X than X holds */
        for (uint32_t b = 0; b + as <= M; b++) c += epam * binom3(nR, b) * seed_cost(M - b, verify_b);
      }
      if (best < 0 || c < best) {
        best = c;
        astar[o] = as <= M ? as : 15;
      }
    }
  }
  /* k_search sizes a class's two-symbol extension by the largest o it may reach: keep the
   * thresholds non-increasing in o so that "allowed at o" implies "allowed below o" */
  for (uint32_t o = 1; o < 8; o++)
    if (astar[o] > astar[o - 1]) astar[o] = astar[o - 1];
}

/* ---- seed recipes (gs_search_args::rec_*) -------------------------------------------------------
 * The depth-k seeds of an item are the same set of substitution patterns for every guide: which
 * steps are substituted, by which of the three other bases (a digit relative to the guide's own
 * symbol), read from which copy of the table.  The lists are written once per (budget, geometry,
 * thresholds) and kept on the handle; a seeding step of k_search hands recipe pos + lane to lane. */
static inline uint64_t recipe_word(uint32_t n, uint32_t lo, bool rot, uint32_t rs, const uint32_t *fields) {
  uint64_t w = (uint64_t)n | ((uint64_t)lo << 3) | (rot ? (1ull << 6) | ((uint64_t)rs << 7) : 0ull);
  for (uint32_t i = 0; i < 7; i++) w |= (uint64_t)(i < n ? fields[i] : 3u) << (12 + 7 * i);
  return w;
}
/* this strand's seeds: variants of the first k-2 steps with j substitutions (ax of them among the first
 * nX steps, set X) x the two-symbol extensions the budget allows; two-sided (astar != nullptr): only
 * what has ax < astar[substitutions outside X] */
static void build_recipes_a(std::vector<uint64_t> &out, uint32_t k, uint32_t m, uint32_t nX, const uint32_t *astar, bool rot,
                            bool pair8 = false) {
  const uint32_t kp = k - 2, xmask = nX >= 32 ? 0xFFFFFFFFu : (1u << nX) - 1u;
  auto mine = [&](uint32_t ax, uint32_t o) { return !astar || (o < 8 && ax < astar[o]); };
  const uint32_t jmax = std::min(std::min(m, kp), 7u);
  for (uint32_t j = 0; j <= jmax; j++)
    for (uint32_t mk = 0; mk < (1u << kp); mk++) {
      if ((uint32_t)__builtin_popcount(mk) != j) continue;
      const uint32_t ax = (uint32_t)__builtin_popcount(mk & xmask), o0 = j - ax;
      if (!mine(ax, o0)) continue;
      if (astar && nX > kp) {
        /* X reaches into the two-symbol extension (27-symbol sites at k = 14: X = steps 0 .. 12): a substitution at step
         * k-2 counts for X, one at step k-1 for O - every (e2, e1) is taken or left by itself, from the plain table */
        uint32_t steps[8], ns = 0;
        for (uint32_t t = 0; t < kp; t++)
          if ((mk >> t) & 1u) steps[ns++] = t;
        uint32_t ndig = 1;
        for (uint32_t i = 0; i < j; i++) ndig *= 3;
        for (uint32_t dc = 0; dc < ndig; dc++) {
          uint32_t f[10], x = dc;
          for (uint32_t i = j; i-- > 0;) {
            f[i] = (steps[i] << 2) | (x % 3);
            x /= 3;
          }
          for (uint32_t e2 = 0; e2 < 4; e2++)
            for (uint32_t e1 = 0; e1 < 4; e1++) {
              uint32_t n = j;
              if (e2) f[n++] = ((k - 2) << 2) | (e2 - 1);
              if (e1) f[n++] = ((k - 1) << 2) | (e1 - 1);
              if (n > m || n > 7 || !mine(ax + (e2 ? 1u : 0u), o0 + (e1 ? 1u : 0u))) continue;
              out.push_back(recipe_word(n, 0, false, 0, f));
            }
        }
        continue;
      }
      uint32_t eb = 0; /* substitutions the extension may add */
      while (eb < 2 && j + eb + 1 <= m && mine(ax, o0 + eb + 1)) eb++;
      uint32_t steps[8], ns = 0, plast = 0;
      for (uint32_t t = 0; t < kp; t++)
        if ((mk >> t) & 1u) steps[ns++] = plast = t;
      uint32_t ndig = 1;
      for (uint32_t i = 0; i < j; i++) ndig *= 3;
      for (uint32_t dc = 0; dc < ndig; dc++) {
        uint32_t f[8], x = dc;
        for (uint32_t i = j; i-- > 0;) { /* the last substituted step's digit runs fastest */
          f[i] = (steps[i] << 2) | (x % 3);
          x /= 3;
        }
        auto emit = [&](uint32_t e2, uint32_t e1, bool r, uint32_t rs) {
          uint32_t n = j;
          if (e2) f[n++] = ((k - 2) << 2) | (e2 - 1);
          if (e1) f[n++] = ((k - 1) << 2) | (e1 - 1);
          if (n > m || n > 7 || !mine(ax, o0 + (n - j))) return;
          out.push_back(recipe_word(n, 0, r, rs, f));
        };
        if (eb >= 2) { /* 16 neighbours of the plain table: 4 lines */
          for (uint32_t e2 = 0; e2 < 4; e2++)
            for (uint32_t e1 = 0; e1 < 4; e1++) emit(e2, e1, false, 0);
        } else if (eb == 1) { /* one line of the plain table + one of the copy rotated at step k-2 */
          for (uint32_t e1 = 0; e1 < 4; e1++) emit(0, e1, false, 0);
          /* a PAM-pair table's 8-byte entries: the three are in the same 128-byte block as the four */
          for (uint32_t e2 = 1; e2 < 4; e2++) emit(e2, 0, rot && !pair8, k - 2);
        } else {
          emit(0, 0, rot && j >= 1, plast);
        }
      }
    }
}
/* the other strand's seeds under two-sided seeding: classes (o substitutions in O, b in R) with
 * astar[o] + o + b <= m; step y consumes guide symbol L-1-y: R = y in [0, L-k), O = y in [L-k, k-P).
 * The recipe carries lo = astar[o], the least number of substitutions its rows need inside X. */
static void build_recipes_b(std::vector<uint64_t> &out, uint32_t k, uint32_t L, uint32_t P, uint32_t m, uint32_t nX,
                            const uint32_t *astar, bool rot, bool deep) {
  /* deep tables (gs_pairtab.hip): k-2 guide symbols index the table, the copies are numbered by guide symbol */
  const uint32_t nO = k - nX, nR = L - k, ylo = L - k, nY = L - nX, step0 = deep ? 0 : P, kd = deep ? nY : k;
  for (uint32_t o = 0; o <= m && o <= nO && o < 8; o++)
    for (uint32_t b = 0; b <= nR && astar[o] + o + b <= m; b++) {
      if (astar[o] > nX || o + b > 7) continue;
      const uint32_t jb = o + b;
      uint32_t ndig = 1;
      for (uint32_t i = 0; i < jb; i++) ndig *= 3;
      for (uint32_t mo = 0; mo < (1u << nO); mo++) {
        if ((uint32_t)__builtin_popcount(mo) != o) continue;
        for (uint32_t mr = 0; mr < (1u << nR); mr++) {
          if ((uint32_t)__builtin_popcount(mr) != b) continue;
          const uint32_t mk = (mo << ylo) | mr;
          uint32_t ys[8], ns = 0, ymax = 0;
          for (uint32_t y = 0; y < nY; y++)
            if ((mk >> y) & 1u) ys[ns++] = ymax = y;
          const uint32_t slast = step0 + ymax; /* consumption step of the last substituted symbol */
          const bool r = rot && !deep && jb >= 1 && slast + 2 <= kd; /* a deep table's line is one index */
          for (uint32_t dc = 0; dc < ndig; dc++) {
            uint32_t f[8], x = dc;
            for (uint32_t i = jb; i-- > 0;) {
              f[i] = (ys[i] << 2) | (x % 3);
              x /= 3;
            }
            out.push_back(recipe_word(jb, astar[o] > 7 ? 7u : astar[o], r, slast, f));
          }
        }
      }
    }
}
extern "C" gs_status gs_debug_seed_recipes(uint32_t k, uint32_t L, uint32_t P, uint32_t m, uint32_t n_x,
                                           const uint32_t *astar, uint32_t deep, uint64_t *out, uint64_t cap,
                                           uint64_t counts[3]) {
  if (k < 4 || k > 16 || L < k || L > 31 || m > 7 || n_x + 1 > k || !counts) return GS_ERR_ARG;
  try {
    std::vector<uint64_t> all;
    build_recipes_a(all, k, m, n_x, nullptr, true);
    counts[0] = all.size();
    counts[1] = counts[2] = 0;
    if (astar) {
      build_recipes_a(all, k, m, n_x, astar, true);
      counts[1] = all.size() - counts[0];
      build_recipes_b(all, k, L, P, m, n_x, astar, true, deep != 0);
      counts[2] = all.size() - counts[0] - counts[1];
    }
    for (uint64_t i = 0; i < all.size() && i < cap && out; i++) out[i] = all[i];
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
  return GS_OK;
}
extern "C" void gs_debug_choose_thresholds(uint32_t m, uint32_t n_x, uint32_t n_o, uint32_t n_r, double pam_expansions,
                                           double verify_a, double verify_b, uint32_t astar[8]) {
  gs_choose_astar(m, n_x, n_o, n_r, pam_expansions, astar, verify_a, verify_b);
}

static gs_status gs_recipes_for(gs_index *ix, uint32_t L, uint32_t P, uint32_t m, uint32_t v_rem, const uint32_t *astar,
                                bool deep, hipStream_t st) {
  const uint32_t k = ix->pt_k;
  const bool rot = true; /* the recipes name the copy that would share lines; a table without it reads its plain copy */
  uint64_t key[2] = {((uint64_t)L << 48) | ((uint64_t)P << 40) | ((uint64_t)m << 32) | ((uint64_t)k << 24) |
                         ((uint64_t)v_rem << 16) | (deep ? 4u : 0u) | (rot ? 2u : 0u) | (astar ? 1u : 0u),
                     0};
  if (astar)
    for (uint32_t o = 0; o < 8; o++) key[1] |= (uint64_t)(astar[o] > 15 ? 15u : astar[o]) << (4 * o);
  for (uint32_t i = 0; i < 2; i++)
    if (ix->rec[i].valid && ix->rec[i].key[0] == key[0] && ix->rec[i].key[1] == key[1]) {
      ix->rec_cur = i;
      return GS_OK;
    }
  std::vector<uint64_t> all;
  build_recipes_a(all, k, m, v_rem, nullptr, rot);
  const size_t n_full = all.size();
  size_t n_a = 0, n_b = 0;
  size_t n_a8 = 0;
  if (astar) {
    build_recipes_a(all, k, m, v_rem, astar, rot);
    n_a = all.size() - n_full;
    build_recipes_b(all, k, L, P, m, v_rem, astar, rot, deep);
    n_b = all.size() - n_full - n_a;
    build_recipes_a(all, k, m, v_rem, astar, rot, true); /* this strand's share read through PAM-pair tables */
    n_a8 = all.size() - n_full - n_a - n_b;
  }
  if (all.size() >= (1ull << 31)) {
    gs_set_error("seed plan too large for this mismatch budget");
    return GS_ERR_UNSUPPORTED;
  }
  /* into the set the last call did not use (or an empty one) */
  const uint32_t slot = !ix->rec[ix->rec_cur].valid ? ix->rec_cur : ix->rec_cur ^ 1u;
  gs_recipe_set &R = ix->rec[slot];
  R.valid = false;
  gs_status rc = gs_reserve(R.buf, 8 * all.size() + 64);
  if (rc != GS_OK) return rc;
  GS_HIP(hipMemcpyAsync(R.buf.p, all.data(), 8 * all.size(), hipMemcpyHostToDevice, st));
  GS_HIP(hipStreamSynchronize(st)); /* `all` is a local */
  R.n_full = (uint32_t)n_full;
  R.n_a = (uint32_t)n_a;
  R.n_b = (uint32_t)n_b;
  R.n_a8 = (uint32_t)n_a8;
  R.a_rot_first = 31; /* of the list read through PAM-pair tables */
  for (size_t i = n_full + n_a + n_b; i < all.size(); i++)
    if (all[i] & 64u) R.a_rot_first = std::min(R.a_rot_first, (uint32_t)(all[i] >> 7) & 31u);
  R.key[0] = key[0];
  R.key[1] = key[1];
  R.valid = true;
  ix->rec_cur = slot;
  if (gs_opt(ix, "GS_DEBUG"))
    fprintf(stderr, "[gs] seed recipes: %zu one-sided, %zu + %zu two-sided (%.1f MB)\n", n_full, n_a, n_b, 8e-6 * all.size());
  return GS_OK;
}


static gs_status enumerate_device_impl(gs_index *ix, const void *d_guides, uint64_t n, uint32_t L,
                                       const void *d_guide_pams, uint32_t P, const char *alt_pams,
                                       uint32_t n_alt, uint32_t mismatches, uint32_t flags,
                                       void *stream, const void **d_offsets, const void **d_hits,
                                       gs_result_view *stats);
extern "C" gs_status gs_enumerate_device(gs_index *ix, const void *d_guides, uint64_t n, uint32_t L,
                                         const void *d_guide_pams, uint32_t P, const char *alt_pams,
                                         uint32_t n_alt, uint32_t mismatches, uint32_t flags,
                                         void *stream, const void **d_offsets, const void **d_hits,
                                         gs_result_view *stats) {
  GS_HANDLE_LOCK(ix);
  try { /* the plans and lists built per batch live in std containers: nothing may throw across the C boundary */
    gs_status rc = enumerate_device_impl(ix, d_guides, n, L, d_guide_pams, P, alt_pams, n_alt, mismatches, flags, stream,
                                         d_offsets, d_hits, stats);
    /* the batch's workspace did not fit.  First what earlier batches left on the handle and this one may not need goes -
     * a batch ordered device-wide leaves tens of bytes per record in a dozen arrays that a batch ordered in tiles never
     * touches, and the other way round (10^9 records: 70 GB either way) - and the batch is redone: every workspace
     * buffer grows again on demand.  Then the derived tables, one kind at a time: the strand tables' rotated copies,
     * then the PAM-pair tables. */
    if (rc == GS_ERR_NOMEM && ix) {
      (void)hipGetLastError();
      size_t freed = 0;
      for (gs_buffer *b : {&ix->w_b_src, &ix->w_b_cnt, &ix->w_b_prefix, &ix->w_b_recs, &ix->w_b_w0, &ix->w_b_w0b, &ix->w_b_w1, &ix->w_b_idx,
                           &ix->w_b_idxb, &ix->w_b_keep, &ix->w_b_keeps, &ix->w_b_rows, &ix->w_b_rowss, &ix->w_b_s, &ix->w_slots2, &ix->w_h_tmp,
                           &ix->w_t_buckets, &ix->w_t_tiles, &ix->w_t_chunkof, &ix->w_t_big, &ix->w_hits, &ix->w_score_tmp, &ix->w_score_io,
                           &ix->w_arena, &ix->w_shq, &ix->w_slots}) {
        if (b->p) {
          freed += b->cap;
          (void)hipFree(b->p);
        }
        b->p = nullptr;
        b->cap = 0;
      }
      if (freed > ((size_t)1 << 30)) {
        if (gs_opt(ix, "GS_DEBUG")) fprintf(stderr, "[gs] out of device memory: %.1f GB of workspace released, batch redone\n", 1e-9 * (double)freed);
        rc = enumerate_device_impl(ix, d_guides, n, L, d_guide_pams, P, alt_pams, n_alt, mismatches, flags, stream, d_offsets, d_hits, stats);
      }
    }
    if (rc == GS_ERR_NOMEM && ix && gs_strand_rot_release(ix)) {
      (void)hipGetLastError();
      ix->rot_off = true;
      ix->pairtab_nofit = 0; /* 86 GB came back: a pair that did not fit may now */
      if (gs_opt(ix, "GS_DEBUG")) fprintf(stderr, "[gs] out of device memory: rotated table copies dropped, batch redone without them\n");
      rc = enumerate_device_impl(ix, d_guides, n, L, d_guide_pams, P, alt_pams, n_alt, mismatches, flags, stream, d_offsets,
                                 d_hits, stats);
    }
    if (rc == GS_ERR_NOMEM && ix && (ix->pairtab[0].valid || ix->pairtab[1].valid)) {
      (void)hipGetLastError();
      gs_pairtab_free(ix, 0);
      gs_pairtab_free(ix, 1);
      ix->pairtab_off = true;
      if (gs_opt(ix, "GS_DEBUG")) fprintf(stderr, "[gs] out of device memory: PAM-pair tables dropped, batch redone without them\n");
      rc = enumerate_device_impl(ix, d_guides, n, L, d_guide_pams, P, alt_pams, n_alt, mismatches, flags, stream, d_offsets,
                                 d_hits, stats);
    }
    return rc;
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}
static gs_status enumerate_device_impl(gs_index *ix, const void *d_guides, uint64_t n, uint32_t L,
                                       const void *d_guide_pams, uint32_t P, const char *alt_pams,
                                       uint32_t n_alt, uint32_t mismatches, uint32_t flags,
                                       void *stream, const void **d_offsets, const void **d_hits,
                                       gs_result_view *stats) {
  if (!ix || (!d_guides && n) || (P && !d_guide_pams && n) || (n_alt && !alt_pams))
    return GS_ERR_ARG;
  if (n >= (1ull << 31)) return GS_ERR_ARG;
  if (L < 1 || L > 31 || P > 8 || 2 * L + 3 * P > 59 || mismatches > 7 || n_alt > 31) {
    gs_set_error("device path supports 1<=L<=31, P<=8, 2L+3P<=59, mismatches<=7, <=31 alt PAMs");
    return GS_ERR_UNSUPPORTED;
  }
  const bool wide_key = 2 * L + 3 * P > 52; /* beyond what the walking kernel and the device-wide ordering carry */
  hipStream_t st = (hipStream_t)stream;
  GS_HIP(hipSetDevice(ix->device));
  ix->last_unsupported = 0;
  for (int i = 0; i < 4; i++)
    if (!ix->ev[i]) GS_HIP(hipEventCreate(&ix->ev[i]));

  const uint32_t n32 = (uint32_t)n;
  uint32_t cap = choose_cap(ix, mismatches, L, P, P ? n_alt : 0, flags);
  gs_status rc;
  /* misc: [0..15] uint64 stats ; then work counter / invalid counter */
  if ((rc = gs_reserve(ix->w_misc, 512)) != GS_OK) return rc;
  /* PAM list = alt PAMs ++ the guide's own (process.hpp:51-56).  An alt PAM with a symbol outside
   * A,C,G,T,N is a literal (index.hpp:130-137): it can only match if the genome holds that symbol -
   * then the whole batch belongs to the general path - and is dropped otherwise. */
  std::string alt_kept;
  bool force_general = false;
  if (P)
    for (uint32_t j = 0; j < n_alt; j++) {
      bool plain = true, possible = true;
      for (uint32_t u = 0; u < P; u++) {
        const uint8_t c = (uint8_t)alt_pams[j * P + u];
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') continue;
        plain = false;
        if (!ix->strand[0].has_sym[c] && !ix->strand[1].has_sym[c]) possible = false;
      }
      if (plain)
        alt_kept.append(alt_pams + j * P, P);
      else if (possible)
        force_general = true;
    }
  const uint32_t n_alt_f = P ? (uint32_t)(alt_kept.size() / P) : 0u; /* alt PAMs of the fast path */
  /* a guide record holds four PAM patterns: longer lists are searched in chunks that append to the
   * same match slots (k_order merges them and drops sequences found twice, as the std::set does) */
  const uint32_t n_chunks = (n_alt_f + 1 + 3) / 4;
  if ((rc = gs_reserve(ix->w_grec, sizeof(gs_guide_rec) * (n + 1) * n_chunks)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_flags, n + 16)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_counts, sizeof(uint32_t) * (2 * n + 2))) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_nmatch, sizeof(uint32_t) * (n + 1))) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_nhits, sizeof(uint32_t) * (n + 1))) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_offsets, sizeof(uint64_t) * (n + 2))) != GS_OK) return rc;
  const uint32_t nb = (n32 + SCAN_BLOCK - 1) / SCAN_BLOCK;
  if ((rc = gs_reserve(ix->w_blocksums, sizeof(uint64_t) * (nb + 2))) != GS_OK) return rc;

  unsigned long long *d_stats = (unsigned long long *)ix->w_misc.p;
  uint32_t *d_work = (uint32_t *)((char *)ix->w_misc.p + 128);
  uint32_t *d_invalid = d_work + 1;

  GS_HIP(hipEventRecord(ix->ev[0], st));
  GS_HIP(hipMemsetAsync(ix->w_misc.p, 0, 512, st));
  if (n == 0) {
    GS_HIP(hipMemsetAsync(ix->w_offsets.p, 0, sizeof(uint64_t), st));
    GS_HIP(hipStreamSynchronize(st));
    if (d_offsets) *d_offsets = ix->w_offsets.p;
    if (d_hits) *d_hits = ix->w_hits.p;
    if (stats) {
      memset(stats, 0, sizeof(*stats));
    }
    return GS_OK;
  }
  for (uint32_t c = 0; c < n_chunks; c++) {
    gs_prep_args pa;
    memset(&pa, 0, sizeof(pa));
    pa.guides = (const uint8_t *)d_guides;
    pa.guide_pams = (const uint8_t *)d_guide_pams;
    for (uint32_t j = 0; j < n_alt_f; j++)
      for (uint32_t u = 0; u < P; u++) pa.alt[j][u] = (uint8_t)alt_kept[j * P + u];
    pa.out = (gs_guide_rec *)ix->w_grec.p + (size_t)c * n;
    pa.n_invalid = d_invalid;
    pa.flags = (uint8_t *)ix->w_flags.p;
    pa.n = n32;
    pa.L = L;
    pa.P = P;
    pa.n_alt = n_alt_f; /* empty guide PAM drops the alt PAMs: process.hpp:52-53 */
    pa.start = (flags & GS_FLAG_PAM_AT_START) ? 1 : 0;
    pa.chunk = c;
    pa.force_invalid = force_general ? 1u : 0u;
    pa.pair_hist = (uint32_t *)((char *)ix->w_misc.p + 256);
    hipLaunchKernelGGL(k_prepare, dim3((n32 + 255) / 256), dim3(256), 0, st, pa);
  }
  /* guides the fast path does not encode get empty hit lists and a flag; the batch goes on */
  uint32_t h_invalid = 0, h_pairs[17] = {0};
  GS_HIP(hipMemcpyAsync(&h_invalid, d_invalid, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(h_pairs, (char *)ix->w_misc.p + 256, sizeof(h_pairs), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  ix->last_unsupported = h_invalid;
  const uint32_t n_alt_given = n_alt;
  (void)n_alt_given;
  n_alt = n_alt_f;
  alt_pams = alt_kept.data();

  const int cus = gs_num_cus(ix->device);
  float ms_search = 0.f;
  /* context verification is possible when what remains after the table depth fits ctx[] */
  uint32_t v_rem = 0;
  if (ix->pt_k >= 4 && ix->pt_k + 1 <= L && !(flags & GS_FLAG_FAITHFUL_WALK) && ix->strand[0].ctx &&
      ix->strand[1].ctx && ix->strand[0].ctx16 && ix->strand[1].ctx16 && L + P - ix->pt_k <= 16)
    v_rem = L + P - ix->pt_k;
  uint32_t *d_nlist = d_work + 2;
  if (wide_key && v_rem == 0) {
    gs_set_error("match sequences beyond 52 key bits (2L+3P > 52) need the table-seeded search: this index's prefix table is too "
                 "shallow for them (or the reference-order walk was asked for) - gs_enumerate_general carries such sequences as bytes");
    return GS_ERR_UNSUPPORTED;
  }

  /* two-sided seeding (k_search): possible when set X (the first consumed guide symbols, which only
   * this strand's table covers) lies inside the recipes' positions, the PAM fits the table depth and
   * both inverse suffix arrays exist */
  bool bidir = false, deep = false;
  uint32_t astar_packed = 0xFFFFFFFFu, astar[8] = {15, 15, 15, 15, 15, 15, 15, 15};
  uint32_t n_cand[2] = {0, 0};
  const uint4 *d_cand[2] = {nullptr, nullptr};
  const uint32_t *d_cand_off[2] = {nullptr, nullptr}, *d_cand_ids[2] = {nullptr, nullptr};
  uint32_t x_len = v_rem;
  uint32_t n_pt = 0, pt_slot[2] = {0, 0};
  const bool table_seeding = ix->pt_k >= 4 && ix->pt_k + 1 <= L && !(flags & GS_FLAG_FAITHFUL_WALK);
  if (table_seeding && mismatches > 7) {
    gs_set_error("more than 7 mismatches");
    return GS_ERR_UNSUPPORTED;
  }
  const bool two_ok = v_rem != 0 && mismatches >= 1 && v_rem + 1 <= ix->pt_k && P + 1 <= ix->pt_k && ix->pt_k - P <= 21 &&
                      L <= 31 && ix->strand[0].isa && ix->strand[1].isa && !gs_opt(ix, "GS_NO_BIDIR");
  /* the pairs of bases the batch's patterns end in (k_prepare's tally), most frequent first */
  uint32_t want[2] = {16, 16}, n_codes = 0;
  for (uint32_t c = 0; c < 16; c++) {
    if (!h_pairs[c]) continue;
    n_codes++;
    if (ix->pairtab_nofit & (1u << c)) continue; /* its tables did not fit on this handle: not tried again */
    if (want[0] == 16 || h_pairs[c] > h_pairs[want[0]]) {
      want[1] = want[0];
      want[0] = c;
    } else if (want[1] == 16 || h_pairs[c] > h_pairs[want[1]]) {
      want[1] = c;
    }
  }
  const uint32_t max_pt = gs_opt(ix, "GS_PAIRTABS") ? std::min(2u, (uint32_t)atol(gs_opt(ix, "GS_PAIRTABS"))) : 2u;
  const bool pairable = two_ok && P >= 2 && v_rem >= 2 && n_codes >= 1 && !ix->pairtab_off && !gs_opt(ix, "GS_NO_PAIRTAB");
  /* deep tables for the other strand's side: every pattern of the batch must have its PAM-pair table */
  uint32_t deep_kb = ix->pt_k - 2; /* guide symbols a deep table is indexed by */
  if (const char *e = gs_opt(ix, "GS_DEEP_SYMBOLS")) deep_kb = (uint32_t)atoi(e);
  bool try_deep = pairable && P == 3 && h_pairs[16] == 0 && n_codes <= max_pt && deep_kb + P >= ix->pt_k && deep_kb <= 14 &&
                  deep_kb + 2 <= L && L <= deep_kb + 16 && L - deep_kb + 2 <= ix->pt_k && !gs_opt(ix, "GS_NO_DEEP");
  for (int attempt = 0; attempt < 2; attempt++) {
    deep = try_deep;
    bidir = false;
    n_pt = 0;
    x_len = deep ? L - deep_kb : v_rem;
    if (two_ok) {
      const uint32_t k = ix->pt_k, m = mismatches;
      const uint32_t nX = x_len, nO = k - x_len, nR = L - k; /* |X|, |O|, |R| */
      /* PAM expansions the other strand enumerates per item (its table holds concrete bases only;
       * a deep table folds the N in: one pass per pattern) */
      double epam = 0;
      {
        const uint32_t np = P ? n_alt + 1 : 1;
        for (uint32_t j = 0; j < np; j++) {
          double e = 1;
          for (uint32_t u = 0; u < P && !deep; u++) {
            const char c = j < n_alt ? alt_pams[j * P + u] : 'N'; /* the guides' own PAM: taken as one wildcard pattern */
            if (c == 'N' && (j < n_alt || u == 0)) e *= 4;
          }
          epam += e;
        }
      }
      gs_choose_astar(m, nX, nO, nR, epam, astar, pairable ? 0.4 : 1.5, deep ? 1.6 : 1.9);
      if (const char *e = gs_opt(ix, "GS_ASTAR")) { /* experiments: "2,2,1,1" */
        uint32_t o = 0;
        for (const char *p = e; *p && o < 8; o++) {
          astar[o] = (uint32_t)strtoul(p, (char **)&p, 10);
          if (*p == ',') p++;
        }
      }
      bool any_b = false;
      for (uint32_t o = 0; o <= m && o <= nO && o < 8; o++) any_b = any_b || astar[o] + o <= m;
      if (any_b) {
        bidir = true;
        astar_packed = 0;
        for (uint32_t o = 0; o < 8; o++) astar_packed |= (astar[o] > 15 ? 15u : astar[o]) << (4 * o);
      }
    }
    deep = deep && bidir;
    /* the seed recipes of this (budget, geometry, thresholds): built once per handle and kept */
    if (table_seeding && (rc = gs_recipes_for(ix, L, P, mismatches, x_len, bidir ? astar : nullptr, deep, st)) != GS_OK) return rc;
    /* PAM-pair tables for the (at most two) pairs of bases most patterns of this batch end in */
    if (bidir && pairable) {
      const uint32_t n_want = (want[0] < 16 ? 1u : 0u) + (max_pt > 1 && want[1] < 16 ? 1u : 0u);
      for (int round = 0; round < 2; round++) {
        /* round 0: a slot that already holds a pair stays, a missing one takes what is free; when the
         * second pair does not fit next to a first one built with all its copies, round 1 frees both
         * and gives each half of the room (fewer rotated copies each, but both patterns served) */
        n_pt = 0;
        bool taken[2] = {false, false};
        for (uint32_t i = 0; i < max_pt; i++) {
          if (want[i] == 16) continue;
          for (uint32_t s = 0; s < 2; s++)
            if (!taken[s] && ix->pairtab[s].valid && ix->pairtab[s].code == want[i] && ix->pairtab[s].v_rem == v_rem) {
              taken[s] = true;
              break;
            }
        }
        uint32_t to_build = 0;
        for (uint32_t i = 0; i < max_pt; i++) {
          if (want[i] == 16) continue;
          bool have = false;
          for (uint32_t j = 0; j < 2; j++)
            have = have || (ix->pairtab[j].valid && ix->pairtab[j].code == want[i] && ix->pairtab[j].v_rem == v_rem);
          to_build += have ? 0u : 1u;
        }
        for (uint32_t i = 0; i < max_pt; i++) {
          if (want[i] == 16) continue;
          uint32_t s = 2;
          bool have = false;
          for (uint32_t j = 0; j < 2; j++)
            if (ix->pairtab[j].valid && ix->pairtab[j].code == want[i] && ix->pairtab[j].v_rem == v_rem) {
              s = j;
              have = true;
            }
          if (s == 2)
            for (uint32_t j = 0; j < 2; j++)
              if (!taken[j]) {
                s = j;
                taken[j] = true;
                break;
              }
          if (s == 2) continue;
          const bool frozen = (flags & GS_FLAG_NO_NEW_TABLES) != 0; /* use what the handle holds, build nothing */
          if (!have && frozen) continue;
          if ((rc = gs_pairtab_ensure(ix, s, v_rem, want[i], frozen ? 31u : ix->rec[ix->rec_cur].a_rot_first,
                                      have ? 1.0 : 1.0 / (double)to_build, st)) != GS_OK)
            return rc;
          if (!have && to_build) to_build--;
          if (ix->pairtab[s].valid && deep && !(frozen && !ix->pairtab[s].deep) &&
              (rc = gs_pairtab_ensure_deep(ix, s, P, deep_kb, st)) != GS_OK)
            return rc;
          if (ix->pairtab[s].valid) pt_slot[n_pt++] = s;
        }
        if (n_pt == n_want || n_want < 2 || (flags & GS_FLAG_NO_NEW_TABLES)) break;
        /* a pair whose tables did not fit: remembered on the handle, so that later batches do not free and
         * rebuild the first pair's tables every call for nothing (cleared when memory is given back) */
        auto mark_missing = [&]() {
          for (uint32_t i = 0; i < max_pt; i++) {
            if (want[i] == 16) continue;
            bool have = false;
            for (uint32_t j = 0; j < 2; j++)
              have = have || (ix->pairtab[j].valid && ix->pairtab[j].code == want[i] && ix->pairtab[j].v_rem == v_rem);
            if (!have) ix->pairtab_nofit |= 1u << want[i];
          }
        };
        if (round == 1) {
          mark_missing();
          break;
        }
        /* round 1 frees a valid first table only when two tables without any rotated copy are known to fit */
        {
          size_t free_b = 0, total_b = 0;
          GS_HIP(hipMemGetInfo(&free_b, &total_b));
          double reserve = 64e9;
          if (const char *e = gs_opt(ix, "GS_PAIRTAB_RESERVE_GB")) reserve = atof(e) * 1e9;
          if (reserve > 0.25 * (double)total_b) reserve = 0.25 * (double)total_b;
          double room = (double)free_b + (double)ix->pairtab[0].bytes + (double)ix->pairtab[1].bytes - reserve;
          if (const char *e = gs_opt(ix, "GS_INDEX_BUDGET_GB"))
            room = std::min(room, atof(e) * 1e9 - (double)(ix->strand[0].bytes + ix->strand[1].bytes));
          const double one = 2.0 * 8.0 * (double)(1ull << (2 * ix->pt_k)) + 10.0 * 1.5 * ((double)ix->strand[0].n + (double)ix->strand[1].n) / 16.0 +
                             8.0 * (double)(1ull << (2 * ix->pt_k)) + 64e6;
          if (2.0 * one > room) {
            mark_missing();
            break;
          }
        }
        gs_pairtab_free(ix, 0);
        gs_pairtab_free(ix, 1);
      }
    }
    if (!try_deep) break;
    bool all_deep = deep && n_pt == n_codes;
    for (uint32_t i = 0; i < n_pt; i++) all_deep = all_deep && ix->pairtab[pt_slot[i]].deep;
    if (all_deep) break;
    try_deep = false; /* not every pattern has its deep table: plan again with the strand tables on that side */
  }
  /* the strand tables' rotated copies: read by this strand's seeds of items without PAM-pair tables, by the
   * other strand's seeds unless the deep tables take them, by one-sided items - built now if any of that
   * can happen in this batch (a batch whose every pattern has its pair + deep tables reads none) */
  if (table_seeding && !(bidir && deep && n_pt != 0 && n_pt == n_codes && h_pairs[16] == 0))
    if ((rc = gs_strand_rot_ensure(ix, st)) != GS_OK) return rc;
  if (bidir) {
    /* windows where a literal 'N' of the genome lies under the PAM (index.hpp:139-149) and the
     * guide part is plain A,C,G,T: the other strand's table cannot hold them (its k-mers spell the
     * PAM), so its share of them is reported from this list.  Window of strand s, left to right:
     * P PAM symbols (last consumed first), then the guide symbols L-1 .. 0.  Entry = {q lo, q hi,
     * PAM symbols in consumption order (3 bits each, 4 = N), position of the site in the strand's text}. */
    std::vector<uint4> cand[2];
    const uint32_t W = L + P;
    const uint64_t len = ix->genome_length;
    auto code = [](uint8_t c) -> int { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; };
    for (const gs_nrun &r : ix->nruns_text) {
      auto at = [&](int64_t pos) -> uint8_t { /* forward text around the run */
        if (pos < 0 || (uint64_t)pos >= len) return 0;
        if ((uint64_t)pos < r.start) return r.start - pos <= GS_NRUN_FLANK ? r.left[GS_NRUN_FLANK - (r.start - pos)] : 0;
        if ((uint64_t)pos < r.start + r.len) return 'N';
        const uint64_t o = pos - (r.start + r.len);
        return o < GS_NRUN_FLANK ? r.right[o] : 0;
      };
      const int64_t s0 = (int64_t)r.start, e0 = (int64_t)(r.start + r.len);
      /* forward strand: the run's tail under the window's first P symbols; text offset o < P holds
       * the PAM symbol of consumption step P-1-o */
      for (int64_t i = e0 - (int64_t)P; i < e0; i++) {
        if (i < 0 || (uint64_t)i + W > len) continue;
        bool ok = true;
        uint64_t q = 0;
        uint32_t pc = 0;
        for (uint32_t o = 0; o < W && ok; o++) {
          const uint8_t c = at(i + o);
          if (o < P) {
            ok = c == 'N' || code(c) >= 0;
            if (ok) pc |= (uint32_t)(c == 'N' ? 4 : code(c)) << (3u * (P - 1u - o));
          } else {
            const int cc = code(c);
            ok = cc >= 0;
            if (ok) q |= (uint64_t)cc << (2u * (L - 1u - (o - P)));
          }
        }
        if (ok) cand[0].push_back(make_uint4((uint32_t)q, (uint32_t)(q >> 32), pc, (uint32_t)i));
      }
      /* reverse strand: its window is the forward window read backwards and complemented, so the
       * run's head lies under the forward window's last P symbols; guide symbol t sits at forward
       * offset t, complemented; PAM step u at forward offset L+u, complemented */
      for (int64_t j = s0 + 1 - (int64_t)W; j <= s0 + (int64_t)P - (int64_t)W; j++) {
        if (j < 0 || (uint64_t)j + W > len) continue;
        bool ok = true;
        uint64_t q = 0;
        uint32_t pc = 0;
        for (uint32_t o = 0; o < W && ok; o++) {
          const uint8_t c = at(j + o);
          if (o >= L) {
            ok = c == 'N' || code(c) >= 0;
            if (ok) pc |= (uint32_t)(c == 'N' ? 4 : 3 - code(c)) << (3u * (o - L));
          } else {
            const int cc = code(c);
            ok = cc >= 0;
            if (ok) q |= (uint64_t)(3 - cc) << (2u * o);
          }
        }
        if (ok) cand[1].push_back(make_uint4((uint32_t)q, (uint32_t)(q >> 32), pc, (uint32_t)(len - ((uint64_t)j + W))));
      }
    }
    n_cand[0] = (uint32_t)cand[0].size();
    n_cand[1] = (uint32_t)cand[1].size();
    if (n_cand[0] + n_cand[1]) {
      /* behind the windows: per strand with many of them, the bucket index (4 x 1025 offsets, 4 x n places) */
      std::vector<uint32_t> bidx[2];
      for (uint32_t s = 0; s < 2; s++) {
        uint32_t from = 256;
        if (const char *e = gs_opt(ix, "GS_CAND_BUCKETS_FROM")) from = (uint32_t)atol(e);
        if (n_cand[s] <= from || mismatches > 3 || L < 20 || gs_opt(ix, "GS_NO_CAND_BUCKETS")) continue;
        const uint32_t nc = n_cand[s];
        bidx[s].assign(4u * 1025u + 4u * (size_t)nc, 0u);
        for (uint32_t c = 0; c < 4; c++) {
          uint32_t *off = bidx[s].data() + 1025u * c, *ids = bidx[s].data() + 4u * 1025u + (size_t)c * nc;
          auto val = [&](uint32_t i) { return (uint32_t)((((uint64_t)cand[s][i].y << 32) | cand[s][i].x) >> (10u * c)) & 1023u; };
          for (uint32_t i = 0; i < nc; i++) off[val(i) + 1u]++;
          for (uint32_t v = 0; v < 1024; v++) off[v + 1u] += off[v];
          std::vector<uint32_t> cur(off, off + 1024);
          for (uint32_t i = 0; i < nc; i++) ids[cur[val(i)]++] = i;
        }
      }
      const size_t b_w = 16 * (size_t)(n_cand[0] + n_cand[1]);
      if ((rc = gs_reserve(ix->w_cand, b_w + 4 * (bidx[0].size() + bidx[1].size()) + 16)) != GS_OK) return rc;
      uint4 *dc = (uint4 *)ix->w_cand.p;
      if (n_cand[0]) GS_HIP(hipMemcpy(dc, cand[0].data(), 16 * (size_t)n_cand[0], hipMemcpyHostToDevice));
      if (n_cand[1]) GS_HIP(hipMemcpy(dc + n_cand[0], cand[1].data(), 16 * (size_t)n_cand[1], hipMemcpyHostToDevice));
      d_cand[0] = dc;
      d_cand[1] = dc + n_cand[0];
      uint32_t *di = (uint32_t *)((char *)ix->w_cand.p + b_w);
      for (uint32_t s = 0; s < 2; s++) {
        if (bidx[s].empty()) continue;
        GS_HIP(hipMemcpy(di, bidx[s].data(), 4 * bidx[s].size(), hipMemcpyHostToDevice));
        d_cand_off[s] = di;
        d_cand_ids[s] = di + 4u * 1025u; /* chunk c's places: from c * n_cand[s] on */
        di += bidx[s].size();
      }
    }
    if (gs_opt(ix, "GS_DEBUG"))
      fprintf(stderr, "[gs] two-sided seeding: astar %u,%u,%u,%u,%u,%u,%u,%u over |X|=%u |O|=%u |R|=%u, "
              "literal-N windows %u + %u%s, PAM-pair tables %u%s\n", astar[0], astar[1], astar[2], astar[3], astar[4], astar[5],
              astar[6], astar[7], x_len, ix->pt_k - x_len, L - ix->pt_k, n_cand[0], n_cand[1],
              d_cand_off[0] || d_cand_off[1] ? " (bucketed by 5-symbol chunks)" : "", n_pt, deep ? " with deep tables" : "");
  }


  const bool count_req = (flags & GS_FLAG_COUNT_REQUESTS) != 0;
  /* overflow arena of the main pass (gs_search_args::arena): sized from what earlier batches on this
   * handle needed; a batch that needs more falls back to the exact-size second pass and leaves a larger
   * arena to the next one */
  uint32_t arena_chunks = 0;
  {
    uint64_t want = ix->arena_chunks;
    if (const char *e = gs_opt(ix, "GS_ARENA_CHUNKS")) want = (uint64_t)atoll(e);
    if (gs_opt(ix, "GS_NO_ARENA")) want = 0;
    if (want > (1ull << 21)) want = 1ull << 21; /* 32 GB of records */
    if (want) {
      auto reserve_arena = [&]() {
        return gs_reserve(ix->w_arena, sizeof(uint4) * (want << ARENA_SHIFT)) == GS_OK &&
               gs_reserve(ix->w_arena_meta, 16 * want + 64) == GS_OK && gs_reserve(ix->w_nchunk, sizeof(uint2) * (2 * n + 2)) == GS_OK &&
               gs_reserve(ix->w_cls, 32 * (2 * n + 2)) == GS_OK;
      };
      if (!reserve_arena()) {
        /* no room: give back what only the paths without the arena use (the exact-size array of a second
         * pass, the ordered copy that otherwise lives in the arena, the raw-key sort word) and try again */
        (void)hipGetLastError();
        for (gs_buffer *b : {&ix->w_slots2, &ix->w_b_s, &ix->w_b_w1}) {
          if (b->p) hipFree(b->p);
          b->p = nullptr;
          b->cap = 0;
        }
        if (!reserve_arena()) {
          (void)hipGetLastError();
          want = 0; /* the second pass serves the overflowing guides */
        }
      }
    }
    arena_chunks = (uint32_t)want;
  }
  uint32_t *d_arena_next = d_work + 4;
  uint64_t arena_fail = 0; /* items of the main pass the arena had no chunk left for */
  uint64_t arena_raw = 0;  /* chunks its waves reserved (theirs, their helpers' partly filled ones, reserves not used up) */
  auto run_search = [&](const gs_guide_rec *guides, uint32_t ng, uint4 *slots, uint32_t *counts,
                        uint32_t cap_, unsigned long long h_stats[2],
                        const uint64_t *slot_off = nullptr, bool with_arena = false) -> gs_status {
    GS_HIP(hipMemsetAsync(ix->w_misc.p, 0, 16, st)); /* n_ext, overflow items */
    GS_HIP(hipMemsetAsync(d_stats + 6, 0, 8, st));   /* items the arena failed */
    GS_HIP(hipMemsetAsync(d_work, 0, 4, st));
    GS_HIP(hipMemsetAsync(d_work + 5, 0, 4, st));
    GS_HIP(hipMemsetAsync(d_work + 8, 0, 4, st));
    if (with_arena) {
      GS_HIP(hipMemsetAsync(d_arena_next, 0, 4, st));
      /* every chunk empty until a wave says whose it is: waves reserve several per visit to the counter (k_search) */
      GS_HIP(hipMemsetAsync((uint32_t *)ix->w_arena_meta.p + arena_chunks, 0xFF, 4 * (size_t)arena_chunks, st));
      GS_HIP(hipMemsetAsync(ix->w_arena_meta.p, 0, 4 * (size_t)arena_chunks, st));
    }
    gs_search_args sa;
    memset(&sa, 0, sizeof(sa));
    if (with_arena) ix->last_share[0] = ix->last_share[1] = ix->last_share[2] = ix->last_share[3] = ix->last_share[4] = 0; /* (of the main pass: a redo shares nothing) */
    sa.sd[0] = ix->strand[0].d;
    sa.sd[1] = ix->strand[1].d;
    sa.slots = slots;
    sa.slot_off = slot_off;
    sa.counts = counts;
    sa.work = d_work;
    sa.stats = d_stats;
    sa.n_items = 2 * ng;
    sa.L = L;
    sa.P = P;
    sa.m = mismatches;
    sa.cap = cap_;
    if (with_arena) {
      sa.arena = (uint4 *)ix->w_arena.p;
      sa.arena_next = d_arena_next;
      sa.chunk_item = (uint32_t *)ix->w_arena_meta.p;
      sa.chunk_seq = sa.chunk_item + arena_chunks;
      sa.nchunk = (uint2 *)ix->w_nchunk.p;
      sa.cls = (uint32_t *)ix->w_cls.p;
      sa.arena_chunks = arena_chunks;
      sa.chunk_fill = sa.chunk_item + 2 * (size_t)arena_chunks;
    }
    /* items per visit to the work counter: enough to keep the counter far from its ~88 visits per microsecond,
     * few enough that every resident wave still gets several visits (balance at the tail) */
    {
      const uint64_t waves = (uint64_t)cus * 32u;
      uint64_t take = (2ull * ng) / (waves * 64u); /* 2 M items: 3 (23.8 ms against 26.4 one at a time; 8: 24.4, 64: 26.2) */
      take = take < 1 ? 1 : take > 4 ? 4 : take;
      if (const char *e = gs_opt(ix, "GS_SEARCH_TAKE")) take = (uint64_t)std::max(1l, atol(e));
      sa.take = (uint32_t)take;
    }
    sa.max_iter = gs_opt(ix, "GS_SEARCH_MAX_ITER") ? (uint32_t)atol(gs_opt(ix, "GS_SEARCH_MAX_ITER")) : (1u << 26);
    sa.err = d_work + 5;
    sa.hpass = d_work + 8;
    sa.v_max = VERIFY_MAX_DEFAULT;
    if (const char *e = gs_opt(ix, "GS_VERIFY_MAX")) {
      const long v = atol(e);
      sa.v_max = v < 1 ? 1u : v > 1023 ? 1023u : (uint32_t)v;
    }
    sa.dbg_skip = gs_opt(ix, "GS_DBG_SKIP") ? (uint32_t)atol(gs_opt(ix, "GS_DBG_SKIP")) : 0u;
    sa.cnt_shift = gs_opt(ix, "GS_COUNT_SHIFT") ? (uint32_t)std::min(12l, std::max(4l, atol(gs_opt(ix, "GS_COUNT_SHIFT")))) : 6u;
    sa.astar = 0xFFFFFFFFu;
    if (ix->pt_k >= 4 && ix->pt_k + 1 <= L && !(flags & GS_FLAG_FAITHFUL_WALK)) {
      /* seeds = depth-pt_k nodes: variants of the first pt_k-2 query symbols with j <= m
       * substitutions x the two-symbol extensions the remaining budget allows */
      sa.pt_k = ix->pt_k;
      sa.v_rem = v_rem;
      sa.x_len = x_len;
      sa.bdeep = deep ? 1u : 0u;
      const gs_recipe_set &R = ix->rec[ix->rec_cur];
      sa.rec_full = (const uint2 *)R.buf.p;
      sa.n_rec_full = R.n_full;
      if (bidir) {
        sa.bidir = 1;
        sa.astar = astar_packed;
        sa.rec_a = sa.rec_full + R.n_full;
        sa.n_rec_a = R.n_a;
        sa.rec_b = sa.rec_a + R.n_a;
        sa.n_rec_b = R.n_b;
        sa.rec_a8 = sa.rec_b + R.n_b;
        sa.n_rec_a8 = R.n_a8;
        sa.n_pt = n_pt;
        for (uint32_t i = 0; i < n_pt; i++) {
          sa.pt[i][0] = ix->pairtab[pt_slot[i]].d[0];
          sa.pt[i][1] = ix->pairtab[pt_slot[i]].d[1];
        }
        sa.cand[0] = d_cand[0];
        sa.cand[1] = d_cand[1];
        sa.n_cand[0] = n_cand[0];
        sa.n_cand[1] = n_cand[1];
        for (uint32_t s = 0; s < 2; s++) {
          sa.cand_off[s] = d_cand_off[s];
          sa.cand_ids[s] = d_cand_ids[s];
        }
      }
    }
    /* persistent waves pulling (guide, strand) items: as many 4-wave workgroups per CU as their
     * LDS (verification queue 2.5 KiB + substitution table 1.4 KiB per wave, + 3.5 KiB of stacks in
     * the walking variant) and the registers (8 waves per SIMD = 8 workgroups per CU) allow */
    const bool walk = sa.pt_k == 0 || sa.v_rem == 0;
    const size_t dyn = 0;
    const size_t lds_wg = sizeof(uint4) * (walk ? WAVE_LDS_ENTRIES : WAVE_LDS_FAST) * SEARCH_WAVES;
    uint32_t per_cu = (uint32_t)(160u * 1024u / lds_wg);
    /* every item through PAM-pair + deep tables (no pattern ends in an N, each has its tables): the kernel
     * without the strand tables' side of the seeding */
    const bool spec = !walk && sa.bidir && sa.bdeep && n_pt != 0 && n_pt == n_codes && h_pairs[16] == 0 && !gs_opt(ix, "GS_NO_SPEC");
    /* heavy items shared among waves (gs_search_args::shq): table-seeded kernels with the arena, one PAM pass */
    uint32_t *d_shctl = nullptr;
    uint32_t share_min = ix->opt_share_min, share_max = ix->opt_share_max;
    if (const char *e = gs_opt(ix, "GS_SHARE_MIN")) share_min = (uint32_t)std::max(0l, atol(e));
    if (const char *e = gs_opt(ix, "GS_SHARE_MAX")) share_max = (uint32_t)std::max(128l, atol(e));
    sa.share_min = share_min ? share_min : 0xFFFFFFFFu; /* (every instantiation counts the passes that large: gs_search_args::hpass) */
    /* the heavy instantiation: asked for (GS_HEAVY=1 / 0), or the last batch of this shape on this handle showed heavy
     * verification passes - one per sixteen items, or any at all in a batch of at most 64 items per wave slot of the chip
     * (beyond that the heavy items spread over the waves by themselves, and the plain instantiation is the faster one
     * where such passes are rare: 1 M guides at m <= 3 on a genome without repeat families, 22 ms against 32 - the
     * heavy form's second level spills registers; m <= 6, which has none: 61 against 73-80) */
    bool heavy = with_arena && !walk && n_chunks == 1 && !count_req && share_min != 0 && mismatches < 8 &&
                 ix->seen_key[mismatches] == (((uint64_t)L << 32) | ((uint64_t)P << 16) | (n_alt << 8) | (flags & GS_FLAG_PAM_AT_START)) &&
                 ix->seen_hpass[mismatches] != 0 &&
                 (16.0 * (double)ix->seen_hpass[mismatches] >= (double)ix->seen_items[mismatches] || 2 * (uint64_t)ng <= 64ull * (uint64_t)cus * 32u);
    if (const char *e = gs_opt(ix, "GS_HEAVY")) heavy = atol(e) != 0 && with_arena && !walk && n_chunks == 1 && !count_req && share_min != 0;
    const uint32_t weu = walk ? GS_WAVES_EU : heavy ? GS_WAVES_EU_HEAVY : spec ? GS_WAVES_EU_PD : GS_WAVES_EU_FAST;
    if (per_cu > weu) per_cu = weu; /* 4 SIMDs x weu waves = weu four-wave workgroups per CU */
    uint32_t grid = (uint32_t)cus * per_cu;
    const uint32_t need = (2 * ng + SEARCH_WAVES - 1) / SEARCH_WAVES;
    if (grid > need) grid = need;
    if (gs_opt(ix, "GS_DEBUG")) {
      int occ = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, walk ? k_search_walk : k_search_fast, WAVE * SEARCH_WAVES, dyn);
      fprintf(stderr, "[gs] k_search (%s): grid %u x %u threads, LDS %zu B per workgroup, %d workgroups per CU resident\n",
              walk ? "walk" : "table", grid, WAVE * SEARCH_WAVES, lds_wg, occ);
    }
    if (heavy) {
      uint64_t qcap = ix->shq_packages;
      if (const char *e = gs_opt(ix, "GS_SHARE_QUEUE")) qcap = (uint64_t)std::max(1ll, atoll(e));
      if (qcap > (1ull << 20)) qcap = 1ull << 20; /* 1.2 GB of packages */
      const uint32_t sh_max = std::min<uint32_t>(2 * ng, 1u << 18);
      const size_t meta = 512 + 4 * (size_t)qcap + 4 * (size_t)sh_max + 64 * (size_t)sh_max;
      if (gs_reserve(ix->w_shq, 16 * (size_t)SHQ_PKG * qcap) == GS_OK && gs_reserve(ix->w_sh_meta, meta + 4 * ((size_t)sh_max + 2)) == GS_OK) {
        d_shctl = (uint32_t *)ix->w_sh_meta.p;
        sa.shq = (uint4 *)ix->w_shq.p;
        sa.shq_ctl = d_shctl;
        sa.shq_ready = d_shctl + 128;
        sa.sh_list = sa.shq_ready + qcap;
        sa.sh_acc = sa.sh_list + sh_max;
        sa.shq_cap = (uint32_t)qcap;
        sa.sh_max = sh_max;
        sa.share_min = share_min;
        sa.share_max = std::max(128u, share_max);
        sa.n_waves = grid * SEARCH_WAVES;
        sa.sh_prof = gs_opt(ix, "GS_DEBUG") ? 1u : 0u;
        GS_HIP(hipMemsetAsync(d_shctl, 0, meta, st));
        if (sa.sh_prof) GS_HIP(hipMemsetAsync(d_shctl + 104, 0xFF, 8, st)); /* the minimum's start value */
      } else {
        (void)hipGetLastError(); /* no room for the queue: every item stays with its wave */
      }
    }
    GS_HIP(hipEventRecord(ix->ev[1], st));
    for (uint32_t c = 0; c < n_chunks; c++) { /* four PAM patterns per pass, appending to the same slots */
      sa.guides = guides + (size_t)c * ng;
      sa.append = c ? 1u : 0u;
      if (c) GS_HIP(hipMemsetAsync(d_work, 0, 4, st));
      if (walk)
        hipLaunchKernelGGL(k_search_walk, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (spec && count_req)
        hipLaunchKernelGGL(k_search_count_pd, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (spec && sa.shq != nullptr)
        hipLaunchKernelGGL(k_search_heavy_pd, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (sa.shq != nullptr)
        hipLaunchKernelGGL(k_search_heavy, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (spec)
        hipLaunchKernelGGL(k_search_fast_pd, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (count_req)
        hipLaunchKernelGGL(k_search_count, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else
        hipLaunchKernelGGL(k_search_fast, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
    }
    GS_HIP(hipEventRecord(ix->ev[2], st));
    unsigned long long h7[22] = {0}; /* the stats and, behind them, the work words */
    uint32_t h_ctl[128] = {0};
    GS_HIP(hipMemcpyAsync(h7, d_stats, sizeof(h7), hipMemcpyDeviceToHost, st));
    if (d_shctl) GS_HIP(hipMemcpyAsync(h_ctl, d_shctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    GS_HIP(hipGetLastError());
    if (d_shctl && sa.sh_prof) {
      const unsigned long long *pr = (const unsigned long long *)(h_ctl + 104);
      const double us = 0.01, nw = (double)sa.n_waves;
      fprintf(stderr, "[gs] heavy launch: %u waves; the last wave left its items after %.0f us, the last exit after %.0f us; per wave: items %.0f us, "
              "helper episodes %.0f us (%.1f episodes), waiting for a package %.0f us; shared items %u, packages %u (queue %u)\n",
              sa.n_waves, us * (double)(pr[1] - pr[0]), us * (double)(pr[2] - pr[0]), us * (double)pr[3] / nw, us * (double)pr[4] / nw,
              (double)pr[6] / nw, us * (double)pr[5] / nw, h_ctl[96], h_ctl[0], sa.shq_cap);
    }
    if (d_shctl) {
      ix->last_share[0] = std::min(h_ctl[96], sa.sh_max); /* shared items */
      ix->last_share[1] = h_ctl[0];                       /* packages reserved */
      ix->last_share[2] = sa.shq_cap;
      ix->last_share[3] = h_ctl[32];                      /* tickets handed out */
      if (!gs_opt(ix, "GS_SHARE_QUEUE") && (uint64_t)h_ctl[0] + h_ctl[0] / 4 + 64 > ix->shq_packages) ix->shq_packages = (uint64_t)h_ctl[0] + h_ctl[0] / 4 + 64;
      if (h_ctl[96] != 0u && ((const uint32_t *)(h7 + 16))[5] == 0u) {
        /* close the gaps the helpers left (k_share_fix), then read the counters again: it may add overflowing items */
        gs_share_args fa;
        memset(&fa, 0, sizeof(fa));
        fa.ctl = d_shctl;
        fa.sh_list = sa.sh_list;
        fa.sh_acc = sa.sh_acc;
        fa.counts = counts;
        fa.nchunk = sa.nchunk;
        fa.cls = sa.cls;
        fa.chunk_item = sa.chunk_item;
        fa.chunk_seq = sa.chunk_seq;
        fa.chunk_fill = sa.chunk_fill;
        fa.arena_next = d_arena_next;
        fa.slots = slots;
        fa.arena = sa.arena;
        fa.dbase = sa.sh_acc + 16 * (size_t)sa.sh_max;
        fa.dir = sa.chunk_item + 3 * (size_t)arena_chunks;
        fa.stats = d_stats;
        fa.sh_max = sa.sh_max;
        fa.cap = cap_;
        fa.arena_chunks = arena_chunks;
        const uint32_t n_sh = (uint32_t)ix->last_share[0];
        hipLaunchKernelGGL(k_share_scan, dim3(1), dim3(1024), 0, st, fa);
        hipLaunchKernelGGL(k_share_dir, dim3((arena_chunks + 255) / 256), dim3(256), 0, st, fa);
        hipLaunchKernelGGL(k_share_fix, dim3(std::min<uint32_t>(n_sh, (uint32_t)cus * 8u)), dim3(256), 0, st, fa);
        GS_HIP(hipEventRecord(ix->ev[2], st));
        GS_HIP(hipMemcpyAsync(h7, d_stats, sizeof(h7), hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        GS_HIP(hipGetLastError());
      }
    }
    h_stats[0] = h7[0];
    h_stats[1] = h7[1];
    if (with_arena) arena_fail = h7[6];
    if (with_arena) arena_raw = ((const uint32_t *)(h7 + 16))[4];
    if (with_arena && mismatches < 8) { /* the main pass: heavy verification passes per item, for the next batch's choice */
      ix->seen_hpass[mismatches] = ((const uint32_t *)(h7 + 16))[8];
      ix->seen_items[mismatches] = 2 * (uint64_t)ng;
    }
    if (((const uint32_t *)(h7 + 16))[5] != 0u) {
      gs_set_error("internal: an item of the search passed its iteration bound (GS_SEARCH_MAX_ITER)");
      return GS_ERR_DEVICE;
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, ix->ev[1], ix->ev[2]);
    ms_search += ms;
    return GS_OK;
  };
  auto run_order = [&](uint4 *slots, const uint32_t *counts, uint32_t *nmatch, uint32_t *nhits,
                       uint32_t ng, uint32_t cap_, uint32_t max_item) -> gs_status {
    gs_order_args oa;
    oa.slots = slots;
    oa.counts = counts;
    oa.nmatch = nmatch;
    oa.nhits = nhits;
    oa.stats = d_stats;
    oa.n = ng;
    oa.cap = cap_;
    if (cap_ > 128) {
      /* LDS for the largest guide of this pass (2 x the largest item count, as a power of two) */
      uint32_t nmax = 256;
      while (nmax < 2u * max_item && nmax < 2u * cap_) nmax <<= 1;
      const size_t lds = sizeof(uint4) * (size_t)nmax;
      if (lds > 64 * 1024)
        GS_HIP(hipFuncSetAttribute((const void *)k_order_wg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      uint32_t grid = ng;
      const uint32_t gmax = (uint32_t)cus * (uint32_t)(lds > 80 * 1024 ? 1 : lds > 40 * 1024 ? 3 : 6);
      if (grid > gmax) grid = gmax;
      if (grid == 0) grid = 1;
      hipLaunchKernelGGL(k_order_wg, dim3(grid), dim3(256), lds, st, oa, nmax);
      return GS_OK;
    }
    /* four waves per workgroup (4 KiB per wave at cap 64); eight workgroups per CU resident, twice
     * that many launched so the tail balances */
    const uint32_t ow = ORDER_WAVES;
    const size_t lds = sizeof(uint4) * (2 * (size_t)cap_ + ORDER_SMALL) * ow;
    uint32_t grid = (ng + ow - 1) / ow;
    const uint32_t gmax = (uint32_t)cus * 16u;
    if (grid > gmax) grid = gmax;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_order, dim3(grid), dim3(WAVE * ow), lds, st, oa);
    return GS_OK;
  };
  auto run_locate = [&](const uint4 *matches, const uint32_t *nmatch, const uint32_t *gmap, uint32_t ng,
                        uint32_t cap_) {
    gs_locate_args la;
    la.sd[0] = ix->strand[0].d;
    la.sd[1] = ix->strand[1].d;
    la.matches = matches;
    la.nmatch = nmatch;
    la.offsets = (const uint64_t *)ix->w_offsets.p;
    la.gmap = gmap;
    la.hits = (gs_hit *)ix->w_hits.p;
    la.genome_length = ix->genome_length;
    la.n = ng;
    la.cap = cap_;
    la.v_rem = v_rem;
    const size_t lds = sizeof(uint32_t) * (2 * (size_t)cap_ + 1);
    hipLaunchKernelGGL(k_locate, dim3(ng), dim3(WAVE), lds, st, la);
  };

  /* ---- guides whose match count exceeds what k_order sorts in LDS: `n_set`
   * guides whose items' records lie in the main slot array (stride cap) or, for guides on the redo
   * list, in the exact-size array slots2/slot_off2.  Leaves nmatch/nhits per set guide and the
   * sorted arrays k_big_locate reads once the CSR offsets exist. */
  uint64_t big_T = 0;
  bool big_used = false, big_v2 = false;
  void *big_s2 = nullptr; /* the records in final order (one-word form) */
  const unsigned long long *big_wfinal = nullptr; /* and their sort words */
  uint32_t big_gshift = 0;
  bool big_comp = false;    /* the ordering ran as one sort by (word, low bits of the row) */
  uint32_t big_fixed = 0;   /* descents it found inside runs (k_big2_fixruns) */
  /* the device-wide ordering runs in its one-word form (k_big2_*) when the sort word fits 64 bits */
  unsigned long long big_pam_mul = 1, big_n_max = 1;
  gs_big2_tab big_tab;
  uint32_t big_rbits = 1;
  {
    for (uint32_t a = 0; a < 32; a++)
      for (uint32_t r = 0; r < 8; r++) {
        unsigned long long v = 0;
        if (r <= a) {
          double c = 1;
          for (uint32_t i = 0; i < r; i++) c = c * (double)(a - i) / (double)(i + 1);
          v = (unsigned long long)(c + 0.5);
          for (uint32_t i = 0; i < r; i++) v *= 3ull;
        }
        big_tab.n[a][r] = v;
      }
    for (uint32_t j = 0; j <= mismatches && j <= L && j < 8; j++) big_n_max = std::max(big_n_max, big_tab.n[L][j]);
    for (uint32_t u = 0; u < P; u++) big_pam_mul *= 5ull;
    /* (mismatches, index, rank) as one number below the guide: `4 + big_rbits` bits hold the count of all classes */
    unsigned long long cum = 0;
    for (uint32_t j = 0; j < 8; j++) {
      const unsigned long long nj = j <= mismatches && j <= L ? big_tab.n[L][j] * big_pam_mul : 0ull;
      big_tab.base[2 * j] = cum;
      big_tab.base[2 * j + 1] = cum + nj;
      cum += 2ull * nj;
    }
    uint32_t cbits = 4;
    while (cbits < 63 && ((cum - 1ull) >> cbits) != 0ull) cbits++;
    big_rbits = cbits - 4;
    (void)big_n_max;
  }
  auto big_fits_v2 = [&](uint32_t n_set) -> bool {
    uint32_t gbits = 1;
    while ((1ull << gbits) < n_set) gbits++;
    return gbits + 4 + big_rbits <= 64 && !gs_opt(ix, "GS_BIG_ORDER_V1");
  };
  /* arena_list != nullptr or arena_all: the set's records are read from the main slots and the overflow
   * arena (the set = the guides of arena_list, or the whole batch), not from a contiguous copy */
  auto big_order = [&](uint32_t n_set, const uint32_t *counts_main, uint32_t cap_, const uint32_t *redo_pos,
                       const uint64_t *slot_off2, const uint32_t *counts2, uint32_t *nmatch_out,
                       uint32_t *nhits_out, bool from_arena = false, const uint32_t *arena_list = nullptr,
                       uint32_t n_used = 0, const uint32_t *arena_redo_pos = nullptr) -> gs_status {
    gs_status r2;
    const uint32_t n_it = 2 * n_set;
    if ((r2 = gs_reserve(ix->w_b_src, sizeof(gs_big_src) * ((size_t)n_it + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_cnt, 8 * ((size_t)n_it + 2))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_prefix, 8 * ((size_t)n_it + 2))) != GS_OK) return r2;
    if (from_arena)
      hipLaunchKernelGGL(k_big2_counts, dim3((n_it + 255) / 256), dim3(256), 0, st, (const uint32_t *)ix->w_counts.p, arena_list,
                         n_it, (unsigned long long *)ix->w_b_cnt.p);
    else
      hipLaunchKernelGGL(k_big_sources, dim3((n_it + 255) / 256), dim3(256), 0, st, counts_main, redo_pos, slot_off2,
                         counts2, n_it, cap_, (gs_big_src *)ix->w_b_src.p, (unsigned long long *)ix->w_b_cnt.p);
    GS_HIP(hipMemsetAsync((unsigned long long *)ix->w_b_cnt.p + n_it, 0, 8, st));
    size_t tb = 0;
    GS_HIP(rocprim::exclusive_scan(nullptr, tb, (unsigned long long *)ix->w_b_cnt.p,
                                   (unsigned long long *)ix->w_b_prefix.p, 0ull, (size_t)n_it + 1,
                                   rocprim::plus<unsigned long long>(), st));
    if ((r2 = gs_reserve(ix->w_h_tmp, tb + 16)) != GS_OK) return r2;
    size_t tbs = ix->w_h_tmp.cap;
    GS_HIP(rocprim::exclusive_scan(ix->w_h_tmp.p, tbs, (unsigned long long *)ix->w_b_cnt.p,
                                   (unsigned long long *)ix->w_b_prefix.p, 0ull, (size_t)n_it + 1,
                                   rocprim::plus<unsigned long long>(), st));
    unsigned long long T = 0;
    GS_HIP(hipMemcpyAsync(&T, (unsigned long long *)ix->w_b_prefix.p + n_it, 8, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    if (T >= (1ull << 32) - 2) {
      gs_set_error("more than 2^32 match records in one batch: use smaller batches at this mismatch budget");
      return GS_ERR_UNSUPPORTED;
    }
    big_T = T;
    big_used = true;
    big_v2 = big_fits_v2(n_set);
    if (from_arena && !big_v2) {
      gs_set_error("internal: device-wide ordering from the arena without its one-word form");
      return GS_ERR_DEVICE;
    }
    if ((r2 = gs_reserve(ix->w_b_recs, 16 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_w0, 8 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_w0b, 8 * (T + 1))) != GS_OK) return r2;
    if (!big_v2 && (r2 = gs_reserve(ix->w_b_w1, 8 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_idx, 4 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_idxb, 4 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_keep, 4 * (T + 2))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_keeps, 4 * (T + 2))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_rows, 8 * (T + 2))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_rowss, 8 * (T + 2))) != GS_OK) return r2;
    uint4 *recs = (uint4 *)ix->w_b_recs.p;
    unsigned long long *w0 = (unsigned long long *)ix->w_b_w0.p, *w0b = (unsigned long long *)ix->w_b_w0b.p,
                       *w1 = (unsigned long long *)ix->w_b_w1.p;
    uint32_t *idx = (uint32_t *)ix->w_b_idx.p, *idxb = (uint32_t *)ix->w_b_idxb.p;
    uint32_t gbits = 1;
    while ((1ull << gbits) < n_set) gbits++;
    const uint32_t rbits = big_rbits;
    const unsigned long long pam_mul = big_pam_mul;
    big_gshift = 4 + rbits;
    if (T && big_v2) {
      /* the records in final order go where the arena's chunks were (read for the last time by the
       * compaction) when they fit there: 16 bytes per record less next to a 220 GB index */
      const bool s2_in_arena = from_arena && ix->w_arena.cap >= 16 * (T + 1);
      if (!s2_in_arena && (r2 = gs_reserve(ix->w_b_s, 16 * (T + 1))) != GS_OK) return r2;
      big_s2 = s2_in_arena ? ix->w_arena.p : ix->w_b_s.p;
      if ((r2 = gs_reserve(ix->w_b_tab, sizeof(gs_big2_tab))) != GS_OK) return r2;
      GS_HIP(hipMemcpy(ix->w_b_tab.p, &big_tab, sizeof(big_tab), hipMemcpyHostToDevice));
      unsigned long long *W = w0, *Wb = w0b;
      uint4 *S2 = (uint4 *)big_s2;
      uint32_t *rk = (uint32_t *)ix->w_b_keep.p, *rkb = (uint32_t *)ix->w_b_keeps.p; /* free until the flags are written */
      gs_big2_compact_args ca;
      memset(&ca, 0, sizeof(ca));
      ca.slots_main = (const uint4 *)ix->w_slots.p;
      ca.slots_alt = (const uint4 *)ix->w_slots2.p;
      ca.src = (const gs_big_src *)ix->w_b_src.p;
      if (from_arena) {
        ca.from_arena = 1;
        ca.arena = (const uint4 *)ix->w_arena.p;
        ca.chunk_item = (const uint32_t *)ix->w_arena_meta.p;
        ca.chunk_seq = ca.chunk_item + arena_chunks;
        ca.counts = (const uint32_t *)ix->w_counts.p;
        ca.list = arena_list;
        ca.redo_pos = arena_redo_pos ? arena_redo_pos : (const uint32_t *)ix->w_b_redo_pos.p;
        ca.cap = cap;
        ca.n_used = n_used;
      }
      ca.prefix = (const unsigned long long *)ix->w_b_prefix.p;
      ca.tab = (const gs_big2_tab *)ix->w_b_tab.p;
      ca.recs = recs;
      ca.W = W;
      ca.rowkey = rk;
      ca.idx = idx;
      ca.pam_mul = pam_mul;
      ca.n_items = n_it;
      ca.L = L;
      ca.P = P;
      ca.rbits = rbits;
      /* Long runs of one sequence (a repeat-rich genome; the handle remembers having seen one): ONE sort by
       * (word << b | low b bits of the first row), b = what 64 bits leave, instead of a sort by row and a
       * stable one by word; the runs it leaves out of order (k_big2_wraps) are put right one by one
       * (k_big2_fixruns).  b < 32 needs no run longer than 2^b (a run is no longer than the largest item):
       * two rows of a run may then differ by a multiple of 2^b only through the high part.  0: not usable. */
      const uint32_t wbits = gbits + 4 + rbits;
      auto composite_bits = [&]() -> uint32_t {
        if (gs_opt(ix, "GS_BIG2_NO_COMPOSITE") || wbits >= 64) return 0u;
        uint32_t rb = 64 - wbits > 32 ? 32u : 64u - wbits;
        if (const char *e = gs_opt(ix, "GS_BIG2_ROWBITS")) return std::min<uint32_t>(rb, (uint32_t)std::max(1l, atol(e)));
        /* the runs to put right afterwards multiply as the row bits shrink (hg38 size, 20 k repeat-rich guides:
         * 243-548 per batch at 25 bits, 4.3 x 10^5 at 17 and 121 ms against the two sorts' 77): below 22 bits -
         * sort words beyond 42 - the two sorts serve */
        return rb >= 22 ? rb : 0u;
      };
      const bool two_from_start = ix->big_long_runs || gs_opt(ix, "GS_BIG2_TWO_SORTS");
      uint32_t rowb = two_from_start ? composite_bits() : 0u;
      ca.row_bits = rowb;
      ca.row_off = gs_opt(ix, "GS_BIG2_ROWOFF") ? (uint32_t)atol(gs_opt(ix, "GS_BIG2_ROWOFF")) : 0u;
      hipLaunchKernelGGL(k_big2_compact, dim3(n_it + (from_arena ? n_used : 0u)), dim3(256), 0, st, ca);
      size_t s1 = 0, s2 = 0, s3 = 0;
      GS_HIP(rocprim::radix_sort_pairs(nullptr, s1, rk, rkb, idx, idxb, (size_t)T, 0, 32, st));
      GS_HIP(rocprim::radix_sort_pairs(nullptr, s2, Wb, W, idxb, idx, (size_t)T, 0, wbits, st));
      GS_HIP(rocprim::radix_sort_pairs(nullptr, s3, Wb, W, idxb, idx, (size_t)T, 0, 64, st));
      if ((r2 = gs_reserve(ix->w_h_tmp, std::max(std::max(s1, s2), s3) + 16)) != GS_OK) return r2;
      const unsigned gT = (unsigned)((T + 255) / 256);
      const unsigned long long *W_final = nullptr;
      const uint32_t *idx_final = nullptr;
      uint32_t wshift = 0;
      bool comp_in_wb = false; /* the composite words were built from the plain ones, into Wb */
      /* One sort by W and the rows put in order inside its (short, rare) runs - unless this handle has seen a
       * batch with long runs of one sequence (a repeat-rich genome): then, and for the batch that shows the
       * first such run, two stable sorts: by first row, then by W. */
      uint32_t short_max = 32;
      if (const char *e = gs_opt(ix, "GS_BIG2_SHORT")) short_max = (uint32_t)std::max(1l, atol(e));
      if (!two_from_start) {
        tbs = ix->w_h_tmp.cap;
        GS_HIP(rocprim::radix_sort_pairs(ix->w_h_tmp.p, tbs, W, Wb, idx, idxb, (size_t)T, 0, wbits, st));
        uint32_t *d_long = d_work + 6;
        GS_HIP(hipMemsetAsync(d_long, 0, 4, st));
        hipLaunchKernelGGL(k_big2_runs, dim3(gT), dim3(256), 0, st, (const unsigned long long *)Wb, (const uint32_t *)idxb,
                           (const uint4 *)recs, T, short_max, idx, d_long);
        uint32_t h_long = 0;
        GS_HIP(hipMemcpyAsync(&h_long, d_long, 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        if (!h_long) {
          W_final = Wb;
          idx_final = idx;
        } else {
          ix->big_long_runs = true;
          hipLaunchKernelGGL(k_iota_u32, dim3(gT), dim3(256), 0, st, idx, T);
          rowb = composite_bits();
          if (rowb) { /* the plain words and the rows are there: the composite words go where the failed order was */
            hipLaunchKernelGGL(k_big2_comp, dim3(gT), dim3(256), 0, st, (const unsigned long long *)W, (const uint32_t *)rk, T, rowb,
                               ca.row_off, Wb);
            comp_in_wb = true;
          }
        }
      }
      if (!W_final && rowb) {
        unsigned long long *src = comp_in_wb ? Wb : W, *dst = comp_in_wb ? W : Wb;
        tbs = ix->w_h_tmp.cap;
        GS_HIP(rocprim::radix_sort_pairs(ix->w_h_tmp.p, tbs, src, dst, idx, idxb, (size_t)T, 0, wbits + rowb, st));
        W_final = dst;
        idx_final = idxb;
        wshift = rowb;
        big_comp = true;
      }
      if (!W_final) {
        tbs = ix->w_h_tmp.cap;
        GS_HIP(rocprim::radix_sort_pairs(ix->w_h_tmp.p, tbs, rk, rkb, idx, idxb, (size_t)T, 0, 32, st));
        hipLaunchKernelGGL(k_big2_gather_w, dim3(gT), dim3(256), 0, st, (const unsigned long long *)W, (const uint32_t *)idxb, T, Wb);
        tbs = ix->w_h_tmp.cap;
        GS_HIP(rocprim::radix_sort_pairs(ix->w_h_tmp.p, tbs, Wb, W, idxb, idx, (size_t)T, 0, wbits, st));
        W_final = W;
        idx_final = idx;
      }
      big_wfinal = W_final;
      big_gshift = 4 + rbits + wshift;
      /* W_final = the sort words in final order, idx_final = where each record sits in recs */
      hipLaunchKernelGGL(k_big2_gather, dim3(gT), dim3(256), 0, st, (const uint4 *)recs, idx_final, T, S2);
      if (wshift && (wshift < 32 || ca.row_off != 0)) {
        /* the descents go into the array the flags' row counts are written to afterwards, the claims into the
         * flags' own; each run that shows one is put in order through the unordered records' array */
        uint32_t *d_n = d_work + 6, h_n = 0, *list = (uint32_t *)ix->w_b_rows.p;
        GS_HIP(hipMemsetAsync(d_n, 0, 4, st));
        hipLaunchKernelGGL(k_big2_wraps, dim3(gT), dim3(256), 0, st, (const uint4 *)S2, W_final, T, wshift, list, d_n);
        GS_HIP(hipMemcpyAsync(&h_n, d_n, 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        if (h_n) {
          GS_HIP(hipMemsetAsync(ix->w_b_keep.p, 0, 4 * (size_t)(T + 1), st));
          hipLaunchKernelGGL(k_big2_fixruns, dim3(std::min<uint32_t>(h_n, 8192u)), dim3(256), 0, st, S2, recs, W_final, T, wshift,
                             ca.row_off, (const uint32_t *)list, h_n, (uint32_t *)ix->w_b_keep.p);
        }
        big_fixed += h_n;
        if (gs_opt(ix, "GS_DEBUG"))
          fprintf(stderr, "[gs] composite ordering: %llu records, word bits %u, row bits %u, %u descents inside runs\n", T, wbits,
                  wshift, h_n);
      }
      hipLaunchKernelGGL(k_big2_flags, dim3(gT), dim3(256), 0, st, (const uint4 *)S2, W_final, T,
                         (uint32_t *)ix->w_b_keep.p, (unsigned long long *)ix->w_b_rows.p, wshift);
    } else if (T) {
      hipLaunchKernelGGL(k_big_compact, dim3(n_it), dim3(256), 0, st, (const uint4 *)ix->w_slots.p,
                         (const uint4 *)ix->w_slots2.p, (const gs_big_src *)ix->w_b_src.p,
                         (const unsigned long long *)ix->w_b_prefix.p, n_it, recs, w0, w1, idx);
      /* (key bits 31:8, first row): 56 bits; then, stable, (guide, key bits 63:32) */
      uint32_t gbits = 1;
      while ((1ull << gbits) < n_set) gbits++;
      size_t s1 = 0, s2 = 0, s3 = 0, s4 = 0;
      GS_HIP(rocprim::radix_sort_pairs(nullptr, s1, w0, w0b, idx, idxb, (size_t)T, 0, 56, st));
      GS_HIP(rocprim::radix_sort_pairs(nullptr, s2, w0b, w0, idxb, idx, (size_t)T, 0, 32 + gbits, st));
      GS_HIP(rocprim::exclusive_scan(nullptr, s3, (uint32_t *)ix->w_b_keep.p, (uint32_t *)ix->w_b_keeps.p, 0u,
                                     (size_t)T + 1, rocprim::plus<uint32_t>(), st));
      GS_HIP(rocprim::exclusive_scan(nullptr, s4, (unsigned long long *)ix->w_b_rows.p,
                                     (unsigned long long *)ix->w_b_rowss.p, 0ull, (size_t)T + 1,
                                     rocprim::plus<unsigned long long>(), st));
      size_t need = s1 > s2 ? s1 : s2;
      if (s3 > need) need = s3;
      if (s4 > need) need = s4;
      if ((r2 = gs_reserve(ix->w_h_tmp, need + 16)) != GS_OK) return r2;
      tbs = ix->w_h_tmp.cap;
      GS_HIP(rocprim::radix_sort_pairs(ix->w_h_tmp.p, tbs, w0, w0b, idx, idxb, (size_t)T, 0, 56, st));
      const unsigned gT = (unsigned)((T + 255) / 256);
      hipLaunchKernelGGL(k_big_gather_w1, dim3(gT), dim3(256), 0, st, (const unsigned long long *)w1,
                         (const uint32_t *)idxb, T, w0b); /* w0b now holds w1 in the first sort's order */
      tbs = ix->w_h_tmp.cap;
      GS_HIP(rocprim::radix_sort_pairs(ix->w_h_tmp.p, tbs, w0b, w0, idxb, idx, (size_t)T, 0, 32 + gbits, st));
      /* w0 = sorted (guide, key high), idx = final order */
      hipLaunchKernelGGL(k_big_flags, dim3(gT), dim3(256), 0, st, (const uint4 *)recs, (const uint32_t *)idx,
                         (const unsigned long long *)w0, T, (uint32_t *)ix->w_b_keep.p,
                         (unsigned long long *)ix->w_b_rows.p);
    }
    GS_HIP(hipMemsetAsync((uint32_t *)ix->w_b_keep.p + T, 0, 4, st));
    GS_HIP(hipMemsetAsync((unsigned long long *)ix->w_b_rows.p + T, 0, 8, st));
    {
      size_t s3 = 0, s4 = 0;
      GS_HIP(rocprim::exclusive_scan(nullptr, s3, (uint32_t *)ix->w_b_keep.p, (uint32_t *)ix->w_b_keeps.p, 0u,
                                     (size_t)T + 1, rocprim::plus<uint32_t>(), st));
      GS_HIP(rocprim::exclusive_scan(nullptr, s4, (unsigned long long *)ix->w_b_rows.p,
                                     (unsigned long long *)ix->w_b_rowss.p, 0ull, (size_t)T + 1,
                                     rocprim::plus<unsigned long long>(), st));
      if ((r2 = gs_reserve(ix->w_h_tmp, (s3 > s4 ? s3 : s4) + 16)) != GS_OK) return r2;
      tbs = ix->w_h_tmp.cap;
      GS_HIP(rocprim::exclusive_scan(ix->w_h_tmp.p, tbs, (uint32_t *)ix->w_b_keep.p, (uint32_t *)ix->w_b_keeps.p,
                                     0u, (size_t)T + 1, rocprim::plus<uint32_t>(), st));
      tbs = ix->w_h_tmp.cap;
      GS_HIP(rocprim::exclusive_scan(ix->w_h_tmp.p, tbs, (unsigned long long *)ix->w_b_rows.p,
                                     (unsigned long long *)ix->w_b_rowss.p, 0ull, (size_t)T + 1,
                                     rocprim::plus<unsigned long long>(), st));
    }
    uint32_t *d_err = d_work + 3;
    hipLaunchKernelGGL(k_big_totals, dim3((n_set + 255) / 256), dim3(256), 0, st,
                       (const unsigned long long *)ix->w_b_prefix.p, (const uint32_t *)ix->w_b_keeps.p,
                       (const unsigned long long *)ix->w_b_rowss.p, n_set, nmatch_out, nhits_out, d_err);
    uint32_t h_err = 0, h_uq = 0;
    GS_HIP(hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipMemcpyAsync(&h_uq, (uint32_t *)ix->w_b_keeps.p + T, 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    if (h_err) {
      gs_set_error("more than 2^32 hits for one guide");
      return GS_ERR_UNSUPPORTED;
    }
    /* matches counter: these guides were skipped by (or never went through) k_order */
    unsigned long long cur = 0;
    GS_HIP(hipMemcpy(&cur, d_stats + 2, 8, hipMemcpyDeviceToHost));
    cur += h_uq;
    GS_HIP(hipMemcpy(d_stats + 2, &cur, 8, hipMemcpyHostToDevice));
    return GS_OK;
  };
  auto big_locate = [&](const uint32_t *gmap) {
    if (!big_T) return;
    if (big_v2) {
      gs_blocate3_args la;
      la.sd[0] = ix->strand[0].d;
      la.sd[1] = ix->strand[1].d;
      la.S2 = (const uint4 *)big_s2;
      la.W = big_wfinal;
      la.keep = (const uint32_t *)ix->w_b_keep.p;
      la.row_scan = (const unsigned long long *)ix->w_b_rowss.p;
      la.prefix = (const unsigned long long *)ix->w_b_prefix.p;
      la.gmap = gmap;
      la.offsets = (const uint64_t *)ix->w_offsets.p;
      la.hits = (gs_hit *)ix->w_hits.p;
      la.genome_length = ix->genome_length;
      la.T = big_T;
      la.v_rem = v_rem;
      la.gshift = big_gshift;
      hipLaunchKernelGGL(k_big2_locate, dim3((unsigned)((big_T + 255) / 256)), dim3(256), 0, st, la);
      return;
    }
    gs_blocate2_args la;
    la.sd[0] = ix->strand[0].d;
    la.sd[1] = ix->strand[1].d;
    la.recs = (const uint4 *)ix->w_b_recs.p;
    la.idx = (const uint32_t *)ix->w_b_idx.p;
    la.w1s = (const unsigned long long *)ix->w_b_w0.p;
    la.keep = (const uint32_t *)ix->w_b_keep.p;
    la.row_scan = (const unsigned long long *)ix->w_b_rowss.p;
    la.prefix = (const unsigned long long *)ix->w_b_prefix.p;
    la.gmap = gmap;
    la.offsets = (const uint64_t *)ix->w_offsets.p;
    la.hits = (gs_hit *)ix->w_hits.p;
    la.genome_length = ix->genome_length;
    la.T = big_T;
    la.v_rem = v_rem;
    hipLaunchKernelGGL(k_big_locate, dim3((unsigned)((big_T + 255) / 256)), dim3(256), 0, st, la);
  };
  /* exact-size second pass of the guides on the redo list (their counts2 are exact) */
  auto redo_exact = [&](uint32_t n_o) -> gs_status {
    std::vector<uint32_t> c2(2 * (size_t)n_o);
    GS_HIP(hipMemcpy(c2.data(), ix->w_counts2.p, 8 * (size_t)n_o, hipMemcpyDeviceToHost));
    std::vector<uint64_t> h_slot_off(2 * (size_t)n_o + 1, 0);
    for (size_t i = 0; i < 2 * (size_t)n_o; i++) h_slot_off[i + 1] = h_slot_off[i] + c2[i];
    const uint64_t T = h_slot_off.back();
    gs_status r2;
    if ((r2 = gs_reserve(ix->w_slots2, sizeof(uint4) * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_h_off, 8 * h_slot_off.size())) != GS_OK) return r2;
    GS_HIP(hipMemcpyAsync(ix->w_h_off.p, h_slot_off.data(), 8 * h_slot_off.size(), hipMemcpyHostToDevice, st));
    GS_HIP(hipStreamSynchronize(st)); /* h_slot_off is a local */
    unsigned long long h2[2] = {0, 0};
    if ((r2 = run_search((const gs_guide_rec *)ix->w_grec2.p, n_o, (uint4 *)ix->w_slots2.p,
                         (uint32_t *)ix->w_counts2.p, 0, h2, (const uint64_t *)ix->w_h_off.p)) != GS_OK)
      return r2;
    if (h2[1] != 0) {
      gs_set_error("internal: exact-size redo overflowed");
      return GS_ERR_DEVICE;
    }
    return GS_OK;
  };

  /* ---- main pass ---- */
  const uint32_t LDS_CAP_MAX = 4096; /* k_order_wg: 2 * cap records of 16 bytes in LDS */
  /* every guide through the device-wide sort: slots beyond what LDS orders, and - measured at hg38 size,
   * m <= 5: 96.8 ms per 100 k guides against 105.5 - from 1,024 slots on, where the bitonic network over
   * 16-byte records in LDS costs more than nine radix passes (m <= 4, 512 slots: 32.9 against 35.1, LDS kept) */
  uint32_t wide_from = 1024;
  if (const char *e = gs_opt(ix, "GS_ORDER_WIDE_FROM")) wide_from = (uint32_t)atol(e);
  const bool big_batch = cap > LDS_CAP_MAX || (cap >= wide_from && (wide_key ? gs_tileorder_fits(L, P, mismatches) : big_fits_v2(n32)));
  if ((rc = gs_reserve(ix->w_slots, sizeof(uint4) * (size_t)cap * 2 * n)) != GS_OK) return rc;
  unsigned long long h_stats[2] = {0, 0};
  if ((rc = run_search((const gs_guide_rec *)ix->w_grec.p, n32, (uint4 *)ix->w_slots.p,
                       (uint32_t *)ix->w_counts.p, cap, h_stats, nullptr, arena_chunks != 0)) != GS_OK)
    return rc;
  if (arena_chunks != 0 && arena_fail != 0 && !gs_opt(ix, "GS_ARENA_CHUNKS") && n_chunks == 1) {
    /* The arena ran out: a handle's first batch on a repeat-rich genome (the arena starts at 64 MB and is sized from
     * what earlier batches needed).  The counts are exact all the same, so the arena this batch needs is known: it is
     * made that large and the main pass runs once more - a second k_search (tens of ms) instead of the exact-size second
     * pass of the overflowing guides and, for them, the device-wide ordering (half a second at 5 x 10^8 records); the
     * per-guide tile ordering then serves this batch like every later one, and allocates its workspace now. */
    uint32_t *d_need = d_work + 10, h_need = 0;
    GS_HIP(hipMemsetAsync(d_need, 0, 4, st));
    hipLaunchKernelGGL(k_need_chunks, dim3(std::min<uint32_t>((2 * n32 + 255) / 256, 1024u)), dim3(256), 0, st,
                       (const uint32_t *)ix->w_counts.p, 2 * n32, cap, d_need);
    GS_HIP(hipMemcpyAsync(&h_need, d_need, 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    uint64_t want = (uint64_t)h_need + h_need / 4 + (uint64_t)cus * 32u * 16u + 64; /* (+ the waves' reserves) */
    if (want > (1ull << 21)) want = 1ull << 21;
    if (want > arena_chunks && gs_reserve(ix->w_arena, sizeof(uint4) * (want << ARENA_SHIFT)) == GS_OK &&
        gs_reserve(ix->w_arena_meta, 16 * want + 64) == GS_OK) {
      if (gs_opt(ix, "GS_DEBUG"))
        fprintf(stderr, "[gs] the arena ran out (%u chunks, %u needed): main pass run again with %llu\n", arena_chunks, h_need, (unsigned long long)want);
      arena_chunks = (uint32_t)want;
      ix->arena_chunks = want;
      arena_fail = 0;
      if ((rc = run_search((const gs_guide_rec *)ix->w_grec.p, n32, (uint4 *)ix->w_slots.p, (uint32_t *)ix->w_counts.p, cap, h_stats, nullptr,
                           true)) != GS_OK)
        return rc;
    } else {
      (void)hipGetLastError();
    }
  }
  if (stats) stats->n_ext = h_stats[0];
  GS_HIP(hipMemsetAsync(d_stats + 2, 0, 8, st)); /* match counter */
  unsigned long long h_cstat[2] = {0, 0}; /* sum and maximum of this batch's exact per-item counts */
  {
    GS_HIP(hipMemsetAsync(d_stats + 14, 0, 16, st));
    hipLaunchKernelGGL(k_count_stats, dim3(std::min<uint32_t>((2 * n32 + 255) / 256, 1024u)), dim3(256), 0, st,
                       (const uint32_t *)ix->w_counts.p, 2 * n32, d_stats + 14);
    if (cap > 128) { /* sizes k_order_wg's LDS; the small-slot path does not wait for it */
      GS_HIP(hipMemcpyAsync(h_cstat, d_stats + 14, 16, hipMemcpyDeviceToHost, st));
      GS_HIP(hipStreamSynchronize(st));
    }
  }
  ix->last_raw_valid = false;
  if (flags & GS_FLAG_RAW_COUNTS) { /* before k_order replaces the raw records by the unique ones */
    if ((rc = gs_reserve(ix->w_raw, 4 * ((size_t)n + 1))) != GS_OK) return rc;
    hipLaunchKernelGGL(k_raw_counts, dim3((n32 + 3) / 4), dim3(256), 0, st, (const uint4 *)ix->w_slots.p,
                       (const uint32_t *)ix->w_counts.p, n32, cap, (uint32_t *)ix->w_raw.p);
    ix->last_raw_valid = true;
  }
  if (!big_batch)
    if ((rc = run_order((uint4 *)ix->w_slots.p, (const uint32_t *)ix->w_counts.p,
                        (uint32_t *)ix->w_nmatch.p, (uint32_t *)ix->w_nhits.p, n32, cap,
                        (uint32_t)(h_cstat[1] < cap ? h_cstat[1] : cap))) != GS_OK)
      return rc;

  /* ---- redo only the guides whose matches did not fit their slots ---- */
  uint32_t n_o = 0, cap2 = cap, n_used = 0;
  bool redo_big = false, arena_direct = false; /* arena_direct: the ordering reads the slots and the arena themselves */
  bool lds_redo = false, ovf_arena_ok = false; /* the overflowing guides fit k_order_wg's LDS; their records beyond the slots are in the arena */
  std::vector<uint32_t> ovf_c2;                /* exact counts of the overflowing guides' items */
  auto arena_gather = [&](const uint64_t *dst_off, uint32_t cap2_) {
    gs_agather_args ga;
    ga.slots = (const uint4 *)ix->w_slots.p;
    ga.arena = (const uint4 *)ix->w_arena.p;
    ga.counts = (const uint32_t *)ix->w_counts.p;
    ga.chunk_item = (const uint32_t *)ix->w_arena_meta.p;
    ga.chunk_seq = ga.chunk_item + arena_chunks;
    ga.list = (const uint32_t *)ix->w_ovf_list.p;
    ga.redo_pos = (const uint32_t *)ix->w_b_redo_pos.p;
    ga.dst_off = dst_off;
    ga.dst = (uint4 *)ix->w_slots2.p;
    ga.n_o = n_o;
    ga.cap = cap;
    ga.cap2 = cap2_;
    ga.n_used = n_used;
    hipLaunchKernelGGL(k_arena_gather, dim3(2u * n_o + n_used), dim3(256), 0, st, ga);
  };
  auto arena_gather_exact = [&](const uint64_t *dst_off) { arena_gather(dst_off, 0u); };
  if (h_stats[1] != 0) {
    if ((rc = gs_reserve(ix->w_ovf_list, sizeof(uint32_t) * (n + 1))) != GS_OK) return rc;
    GS_HIP(hipMemsetAsync(d_nlist, 0, 4, st));
    hipLaunchKernelGGL(k_collect_overflow, dim3((n32 + 255) / 256), dim3(256), 0, st,
                       (const uint32_t *)ix->w_counts.p, n32, cap, (uint32_t *)ix->w_ovf_list.p, d_nlist);
    GS_HIP(hipMemcpyAsync(&n_o, d_nlist, 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    if ((rc = gs_reserve(ix->w_grec2, sizeof(gs_guide_rec) * (size_t)n_o * n_chunks)) != GS_OK) return rc;
    if ((rc = gs_reserve(ix->w_counts2, sizeof(uint32_t) * 2 * (size_t)n_o)) != GS_OK) return rc;
    if ((rc = gs_reserve(ix->w_nmatch2, sizeof(uint32_t) * (size_t)n_o)) != GS_OK) return rc;
    if ((rc = gs_reserve(ix->w_nhits2, sizeof(uint32_t) * (size_t)n_o)) != GS_OK) return rc;
    for (uint32_t c = 0; c < n_chunks; c++)
      hipLaunchKernelGGL(k_gather_guides, dim3((n_o + 255) / 256), dim3(256), 0, st,
                         (const gs_guide_rec *)ix->w_grec.p + (size_t)c * n, (const uint32_t *)ix->w_ovf_list.p, n_o,
                         (gs_guide_rec *)ix->w_grec2.p + (size_t)c * n_o);
    /* the main pass counted every item's matches exactly, also beyond its slots */
    hipLaunchKernelGGL(k_gather_counts, dim3((n_o + 255) / 256), dim3(256), 0, st,
                       (const uint32_t *)ix->w_counts.p, (const uint32_t *)ix->w_ovf_list.p, n_o,
                       (uint32_t *)ix->w_counts2.p);
    uint32_t need_cap = 0;
    uint64_t need_chunks = 0;
    std::vector<uint32_t> c2(2 * (size_t)n_o);
    GS_HIP(hipMemcpyAsync(c2.data(), ix->w_counts2.p, 8 * (size_t)n_o, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    for (uint32_t c : c2) {
      need_cap = c > need_cap ? c : need_cap;
      if (c > cap) need_chunks += (c - cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT;
    }
    /* the overflowing items' records beyond their slots are in the arena - unless it ran out (or is off):
     * then these guides are searched once more with slots of the exact sizes, and the next batch gets
     * the arena this one would have needed */
    const bool arena_ok = arena_chunks != 0 && arena_fail == 0;
    if (arena_raw > need_chunks) need_chunks = arena_raw; /* (what the waves reserved: a later batch should find as much) */
    if (arena_chunks != 0 && !gs_opt(ix, "GS_ARENA_CHUNKS") && need_chunks + need_chunks / 4 + 64 > ix->arena_chunks)
      ix->arena_chunks = need_chunks + need_chunks / 4 + 64;
    if (arena_ok) {
      GS_HIP(hipMemcpyAsync(&n_used, d_arena_next, 4, hipMemcpyDeviceToHost, st));
      if ((rc = gs_reserve(ix->w_b_redo_pos, 4 * ((size_t)n + 1))) != GS_OK) return rc;
      hipLaunchKernelGGL(k_fill_u32, dim3((n32 + 255) / 256), dim3(256), 0, st, (uint32_t *)ix->w_b_redo_pos.p,
                         0xFFFFFFFFu, n32);
      hipLaunchKernelGGL(k_mark_redo, dim3((n_o + 255) / 256), dim3(256), 0, st, (const uint32_t *)ix->w_ovf_list.p,
                         n_o, (uint32_t *)ix->w_b_redo_pos.p);
      GS_HIP(hipStreamSynchronize(st));
      if (n_used > arena_chunks) n_used = arena_chunks;
    }
    lds_redo = !big_batch && need_cap <= LDS_CAP_MAX;
    if (lds_redo) {
      /* slots every one of these guides fits, ordered in LDS */
      cap2 = 128;
      while (cap2 < need_cap) cap2 <<= 1;
      if ((rc = gs_reserve(ix->w_slots2, sizeof(uint4) * (size_t)cap2 * 2 * n_o)) != GS_OK) return rc;
      if (arena_ok) {
        arena_gather(nullptr, cap2);
      } else {
        unsigned long long h2[2] = {0, 0};
        if ((rc = run_search((const gs_guide_rec *)ix->w_grec2.p, n_o, (uint4 *)ix->w_slots2.p,
                             (uint32_t *)ix->w_counts2.p, cap2, h2)) != GS_OK)
          return rc;
        if (h2[1] != 0) {
          gs_set_error("internal: redo pass overflowed slots sized from exact counts");
          return GS_ERR_DEVICE;
        }
      }
      if ((rc = run_order((uint4 *)ix->w_slots2.p, (const uint32_t *)ix->w_counts2.p,
                          (uint32_t *)ix->w_nmatch2.p, (uint32_t *)ix->w_nhits2.p, n_o, cap2, need_cap)) != GS_OK)
        return rc;
      hipLaunchKernelGGL(k_patch_overflow, dim3((n_o + 255) / 256), dim3(256), 0, st,
                         (const uint32_t *)ix->w_ovf_list.p, n_o, (const uint32_t *)ix->w_nhits2.p,
                         (uint32_t *)ix->w_nhits.p);
    }
    ovf_arena_ok = arena_ok;
    ovf_c2.swap(c2);
  }
  /* ---- the set that LDS does not order: the overflowing guides beyond k_order_wg's reach, or - from 1,024 slots
   * per item on - the whole batch.  Per guide in LDS tiles (gs_tileorder.hip) when k_search counted the classes
   * (arena on) and the sort word fits; the device-wide ordering otherwise, and whenever a tile reports that one
   * of its assumptions did not hold (then everything from the ordering on is done again that way). ---- */
  const bool set_exists = big_batch || (n_o != 0 && !lds_redo);
  /* (the walking kernel's records are intervals; a batch shape that showed overlapping PAM patterns is remembered) */
  uint64_t tile_key = 1469598103934665603ull;
  {
    auto mix = [&](uint64_t v) { tile_key = (tile_key ^ v) * 1099511628211ull; };
    mix(L);
    mix(P);
    mix(n_alt);
    mix(flags & (GS_FLAG_PAM_AT_START | GS_FLAG_FAITHFUL_WALK));
    for (uint32_t i = 0; i < n_alt * P; i++) mix((uint8_t)alt_pams[i]);
  }
  bool tile = set_exists && arena_chunks != 0 && (n_o == 0 || ovf_arena_ok) && v_rem != 0 && gs_tileorder_fits(L, P, mismatches) &&
              !(ix->tile_order_off && ix->tile_order_off_key == tile_key) && !gs_opt(ix, "GS_NO_TILE_ORDER");
  bool tile_used = false, tile_fell_back = false;
  uint32_t guides_left_out = 0; /* guides with an item beyond the tiles' reach, ordered device-wide by themselves */
  const uint32_t TO_F_DUP_HOST = 2u; /* (gs_tileorder.hip's TO_F_DUP: one sequence at one row twice) */
  uint64_t total = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    gs_tileorder_in ti;
    gs_tileorder_state ts;
    memset(&ti, 0, sizeof(ti));
    if (set_exists && tile) {
      if (n_o) {
        if ((rc = gs_reserve(ix->w_b_redo_pos, 4 * ((size_t)n + 1))) != GS_OK) return rc;
        hipLaunchKernelGGL(k_fill_u32, dim3((n32 + 255) / 256), dim3(256), 0, st, (uint32_t *)ix->w_b_redo_pos.p, 0xFFFFFFFFu, n32);
        hipLaunchKernelGGL(k_mark_redo, dim3((n_o + 255) / 256), dim3(256), 0, st, (const uint32_t *)ix->w_ovf_list.p, n_o,
                           (uint32_t *)ix->w_b_redo_pos.p);
      }
      ti.n_set = big_batch ? n32 : n_o;
      ti.list = big_batch ? nullptr : (const uint32_t *)ix->w_ovf_list.p;
      ti.redo_pos = (const uint32_t *)ix->w_b_redo_pos.p;
      ti.counts = (const uint32_t *)ix->w_counts.p;
      ti.cls = (const uint32_t *)ix->w_cls.p;
      ti.slots = (const uint4 *)ix->w_slots.p;
      ti.cap = cap;
      ti.arena = (const uint4 *)ix->w_arena.p;
      ti.chunk_item = (const uint32_t *)ix->w_arena_meta.p;
      ti.chunk_seq = ti.chunk_item + arena_chunks;
      ti.n_used = n_used;
      ti.nhits = (uint32_t *)ix->w_nhits.p;
      ti.L = L;
      ti.P = P;
      ti.m = mismatches;
      ti.v_rem = v_rem;
      bool usable = false;
      if ((rc = gs_tileorder_plan(ix, ti, st, ts, &usable)) != GS_OK) return rc;
      if (!usable) tile = false;
    }
    if (set_exists && !tile && wide_key) {
      gs_set_error("a guide with more matches than LDS orders and a match sequence beyond 52 key bits: the device-wide ordering "
                   "does not carry such keys and the per-guide tile ordering could not take the batch (gs_enumerate_general does)");
      return GS_ERR_UNSUPPORTED;
    }
    if (set_exists && !tile) {
      if (!big_batch) {
        if (ovf_arena_ok && big_fits_v2(n_o)) {
          arena_direct = true; /* no copy at all: the ordering's first kernel reads slots and chunks */
        } else if (ovf_arena_ok) {
          /* the exact-size array the second pass would have filled, filled by copies */
          std::vector<uint64_t> h_slot_off(2 * (size_t)n_o + 1, 0);
          for (size_t i = 0; i < 2 * (size_t)n_o; i++) h_slot_off[i + 1] = h_slot_off[i] + ovf_c2[i];
          if ((rc = gs_reserve(ix->w_slots2, sizeof(uint4) * (h_slot_off.back() + 1))) != GS_OK) return rc;
          if ((rc = gs_reserve(ix->w_h_off, 8 * h_slot_off.size())) != GS_OK) return rc;
          GS_HIP(hipMemcpyAsync(ix->w_h_off.p, h_slot_off.data(), 8 * h_slot_off.size(), hipMemcpyHostToDevice, st));
          GS_HIP(hipStreamSynchronize(st)); /* h_slot_off is a local */
          arena_gather_exact((const uint64_t *)ix->w_h_off.p);
        } else if ((rc = redo_exact(n_o)) != GS_OK) {
          return rc;
        }
        redo_big = true;
        /* the redo list alone goes through the device-wide sort */
        if ((rc = big_order(n_o, nullptr, 0, nullptr, (const uint64_t *)ix->w_h_off.p,
                            (const uint32_t *)ix->w_counts2.p, (uint32_t *)ix->w_nmatch2.p,
                            (uint32_t *)ix->w_nhits2.p, arena_direct, (const uint32_t *)ix->w_ovf_list.p, n_used)) != GS_OK)
          return rc;
        hipLaunchKernelGGL(k_patch_overflow, dim3((n_o + 255) / 256), dim3(256), 0, st,
                           (const uint32_t *)ix->w_ovf_list.p, n_o, (const uint32_t *)ix->w_nhits2.p,
                           (uint32_t *)ix->w_nhits.p);
      } else {
        /* every guide: records from the main slots, or from the arena / the exact-size array for redo guides */
        const uint32_t *redo_pos = nullptr;
        if (n_o) {
          if (ovf_arena_ok && big_fits_v2(n32)) {
            arena_direct = true;
          } else if (ovf_arena_ok) {
            std::vector<uint64_t> h_slot_off(2 * (size_t)n_o + 1, 0);
            for (size_t i = 0; i < 2 * (size_t)n_o; i++) h_slot_off[i + 1] = h_slot_off[i] + ovf_c2[i];
            if ((rc = gs_reserve(ix->w_slots2, sizeof(uint4) * (h_slot_off.back() + 1))) != GS_OK) return rc;
            if ((rc = gs_reserve(ix->w_h_off, 8 * h_slot_off.size())) != GS_OK) return rc;
            GS_HIP(hipMemcpyAsync(ix->w_h_off.p, h_slot_off.data(), 8 * h_slot_off.size(), hipMemcpyHostToDevice, st));
            GS_HIP(hipStreamSynchronize(st));
            arena_gather_exact((const uint64_t *)ix->w_h_off.p);
          } else if ((rc = redo_exact(n_o)) != GS_OK) {
            return rc;
          }
          redo_big = true;
          if ((rc = gs_reserve(ix->w_b_redo_pos, 4 * ((size_t)n + 1))) != GS_OK) return rc;
          hipLaunchKernelGGL(k_fill_u32, dim3((n32 + 255) / 256), dim3(256), 0, st, (uint32_t *)ix->w_b_redo_pos.p,
                             0xFFFFFFFFu, n32);
          hipLaunchKernelGGL(k_mark_redo, dim3((n_o + 255) / 256), dim3(256), 0, st, (const uint32_t *)ix->w_ovf_list.p,
                             n_o, (uint32_t *)ix->w_b_redo_pos.p);
          redo_pos = (const uint32_t *)ix->w_b_redo_pos.p;
        }
        if ((rc = big_order(n32, (const uint32_t *)ix->w_counts.p, cap, redo_pos, (const uint64_t *)ix->w_h_off.p,
                            (const uint32_t *)ix->w_counts2.p, (uint32_t *)ix->w_nmatch.p,
                            (uint32_t *)ix->w_nhits.p, arena_direct, nullptr, n_used)) != GS_OK)
          return rc;
      }
    }

    hipLaunchKernelGGL(k_scan_partial, dim3(nb), dim3(SCAN_BLOCK), 0, st,
                       (const uint32_t *)ix->w_nhits.p, (uint64_t *)ix->w_blocksums.p, n32);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(SCAN_BLOCK), 0, st,
                       (uint64_t *)ix->w_blocksums.p, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(SCAN_BLOCK), 0, st,
                       (const uint32_t *)ix->w_nhits.p, (const uint64_t *)ix->w_blocksums.p,
                       (uint64_t *)ix->w_offsets.p, n32, nb);
    total = 0;
    GS_HIP(hipMemcpyAsync(&total, (uint64_t *)ix->w_offsets.p + n, 8, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    if ((rc = gs_reserve(ix->w_hits, sizeof(gs_hit) * (total + 1))) != GS_OK) return rc;
    if (!big_batch) {
      run_locate((const uint4 *)ix->w_slots.p, (const uint32_t *)ix->w_nmatch.p, nullptr, n32, cap);
      if (n_o && lds_redo)
        run_locate((const uint4 *)ix->w_slots2.p, (const uint32_t *)ix->w_nmatch2.p,
                   (const uint32_t *)ix->w_ovf_list.p, n_o, cap2);
    }
    if (!set_exists) break;
    if (!tile) {
      big_locate(big_batch ? nullptr : (const uint32_t *)ix->w_ovf_list.p);
      break;
    }
    ti.offsets = (const uint64_t *)ix->w_offsets.p;
    ti.hits = (gs_hit *)ix->w_hits.p;
    uint32_t viol = 0;
    if ((rc = gs_tileorder_run(ix, ti, ts, st, &viol)) != GS_OK) return rc;
    if (!viol && ts.n_excl != 0) {
      /* guides with an item beyond the tiles' reach (10^6 records: a guide inside the largest repeat family of a genome):
       * these alone through the device-wide ordering, their records read where k_search left them; the hit list has
       * their places already (as many hits as records: checked - a difference means one sequence at one row twice, and
       * the batch is then ordered device-wide as a whole, like any batch whose tiles meet that) */
      if (wide_key || !big_fits_v2(ts.n_excl)) {
        gs_set_error("a guide with more than 10^6 match records per index and a match sequence beyond 52 key bits: the device-wide "
                     "ordering does not carry such keys (gs_enumerate_general does)");
        return GS_ERR_UNSUPPORTED;
      }
      const uint32_t n_x = ts.n_excl;
      const uint32_t *xlist = (const uint32_t *)ix->w_t_excl.p;
      if ((rc = gs_reserve(ix->w_b_redo_pos2, 4 * ((size_t)n + 1))) != GS_OK) return rc;
      if ((rc = gs_reserve(ix->w_nmatch2, sizeof(uint32_t) * (size_t)std::max(n_x, n_o))) != GS_OK) return rc;
      if ((rc = gs_reserve(ix->w_nhits2, sizeof(uint32_t) * (size_t)std::max(n_x, n_o))) != GS_OK) return rc;
      hipLaunchKernelGGL(k_fill_u32, dim3((n32 + 255) / 256), dim3(256), 0, st, (uint32_t *)ix->w_b_redo_pos2.p, 0xFFFFFFFFu, n32);
      hipLaunchKernelGGL(k_mark_redo, dim3((n_x + 255) / 256), dim3(256), 0, st, xlist, n_x, (uint32_t *)ix->w_b_redo_pos2.p);
      if (n_used == 0) { /* (the chunks in use, when no earlier step asked for them) */
        GS_HIP(hipMemcpyAsync(&n_used, d_arena_next, 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        if (n_used > arena_chunks) n_used = arena_chunks;
      }
      if ((rc = big_order(n_x, nullptr, 0, nullptr, nullptr, nullptr, (uint32_t *)ix->w_nmatch2.p, (uint32_t *)ix->w_nhits2.p, true, xlist,
                          n_used, (const uint32_t *)ix->w_b_redo_pos2.p)) != GS_OK)
        return rc;
      std::vector<uint32_t> hx(n_x), lx(n_x), cx(2 * (size_t)n_x);
      GS_HIP(hipMemcpy(hx.data(), ix->w_nhits2.p, 4 * (size_t)n_x, hipMemcpyDeviceToHost));
      GS_HIP(hipMemcpy(lx.data(), xlist, 4 * (size_t)n_x, hipMemcpyDeviceToHost));
      bool same = true;
      for (uint32_t j = 0; j < n_x && same; j++) {
        GS_HIP(hipMemcpy(&cx[2 * j], (const uint32_t *)ix->w_counts.p + 2 * (size_t)lx[j], 8, hipMemcpyDeviceToHost));
        same = (uint64_t)hx[j] == (uint64_t)cx[2 * j] + cx[2 * j + 1];
      }
      if (same) {
        big_locate(xlist);
        guides_left_out = n_x;
        ix->last_share[4] = n_x;
      } else {
        viol = TO_F_DUP_HOST;
      }
    }
    if (!viol) {
      tile_used = true;
      /* matches counter: these guides were skipped by (or never went through) k_order */
      unsigned long long cur = 0;
      GS_HIP(hipMemcpy(&cur, d_stats + 2, 8, hipMemcpyDeviceToHost));
      cur += ts.n_records;
      GS_HIP(hipMemcpy(d_stats + 2, &cur, 8, hipMemcpyHostToDevice));
      break;
    }
    if (gs_opt(ix, "GS_DEBUG")) fprintf(stderr, "[gs] per-guide tile ordering gave up (flags %u): device-wide ordering instead\n", viol);
    tile = false;
    tile_fell_back = true;
    /* overlapping PAM patterns or interval records are a property of the batch's shape: later batches of this handle skip the attempt */
    if (viol & 3u) {
      ix->tile_order_off = true;
      ix->tile_order_off_key = tile_key;
    }
  }
  GS_HIP(hipEventRecord(ix->ev[3], st));
  unsigned long long h_stats3[16] = {0};
  GS_HIP(hipMemcpyAsync(h_stats3, d_stats, sizeof(h_stats3), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  if (bidir && gs_opt(ix, "GS_DEBUG"))
    fprintf(stderr, "[gs] items: seeded from both strands %llu, one-sided (PAM with more than two N) %llu; slots %u per item, "
            "%u guides redone%s%s\n", h_stats3[4], h_stats3[5], cap, n_o, big_batch ? " (whole batch through the wide ordering)" : "",
            tile_used ? " (per guide in LDS tiles)" : "");
  if (guides_left_out && gs_opt(ix, "GS_DEBUG"))
    fprintf(stderr, "[gs] %u guide(s) with an item beyond the tiles' reach ordered device-wide by themselves\n", guides_left_out);
  h_stats3[6] = n_o;
  h_stats3[7] = (h_stats3[7] << 8) | (big_batch ? 1u : 0u) | (redo_big ? 2u : 0u) |
                (n_o && arena_chunks != 0 && arena_fail == 0 ? 4u : 0u) | /* bit 2: the overflowing guides came out of the arena, no second pass */
                (big_comp ? 8u : 0u) | (big_fixed ? 16u : 0u) |
                (tile_used ? 32u : 0u) | (tile_fell_back ? 64u : 0u);    /* bits 5, 6: ordered per guide in LDS tiles; that form gave up and the device-wide one ran */           /* bits 3, 4: ordered by one sort of (word, row bits); runs put right afterwards */ /* items through PAM-pair tables above the flags */
  h_stats3[13] = cap;
  memcpy(ix->last_counters, h_stats3, sizeof(h_stats3));
  /* matches per item seen at this budget: sizes the slots of the next batch */
  if (mismatches < 8 && n32) {
    ix->seen_mean[mismatches] = (double)h_stats3[14] / (2.0 * n32);
    ix->seen_max[mismatches] = (double)h_stats3[15];
    ix->seen_key[mismatches] = ((uint64_t)L << 32) | ((uint64_t)P << 16) | (n_alt << 8) | (flags & GS_FLAG_PAM_AT_START);
  }
  GS_HIP(hipGetLastError());
  if (d_offsets) *d_offsets = ix->w_offsets.p;
  if (d_hits) *d_hits = ix->w_hits.p;
  if (stats) {
    stats->n_guides = n;
    stats->n_hits = total;
    stats->guide_offsets = nullptr;
    stats->hits = nullptr;
    stats->n_matches = h_stats3[2];
    stats->ms_search = ms_search;
    float ms = 0.f;
    hipEventElapsedTime(&ms, ix->ev[0], ix->ev[3]);
    stats->ms_total = ms;
  }
  return GS_OK;
}

extern "C" gs_status gs_rank_bwt4(gs_index *ix, int strand, const uint64_t *rows, uint64_t n,
                                  uint64_t *out) {
  GS_HANDLE_LOCK(ix);
  if (!ix || strand < 0 || strand > 1 || (n && (!rows || !out))) return GS_ERR_ARG;
  for (uint64_t j = 0; j < n; j++)
    if (rows[j] > ix->strand[strand].n) return GS_ERR_ARG;
  GS_HIP(hipSetDevice(ix->device));
  uint64_t *d_rows = nullptr, *d_out = nullptr;
  if (n == 0) return GS_OK;
  GS_HIP(hipMalloc(&d_rows, 8 * n));
  GS_HIP(hipMalloc(&d_out, 32 * n));
  GS_HIP(hipMemcpy(d_rows, rows, 8 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_rank4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ix->strand[strand].d,
                     d_rows, n, d_out);
  GS_HIP(hipMemcpy(out, d_out, 32 * n, hipMemcpyDeviceToHost));
  hipFree(d_rows);
  hipFree(d_out);
  return GS_OK;
}

extern "C" gs_status gs_resolve(gs_index *ix, int strand, const uint64_t *rows, uint64_t n,
                                uint64_t *out) {
  GS_HANDLE_LOCK(ix);
  if (!ix || strand < 0 || strand > 1 || (n && (!rows || !out))) return GS_ERR_ARG;
  for (uint64_t j = 0; j < n; j++)
    if (rows[j] >= ix->strand[strand].n) return GS_ERR_ARG;
  GS_HIP(hipSetDevice(ix->device));
  uint64_t *d_rows = nullptr, *d_out = nullptr;
  if (n == 0) return GS_OK;
  GS_HIP(hipMalloc(&d_rows, 8 * n));
  GS_HIP(hipMalloc(&d_out, 8 * n));
  GS_HIP(hipMemcpy(d_rows, rows, 8 * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0,
                     ix->strand[strand].d, d_rows, n, d_out);
  GS_HIP(hipMemcpy(out, d_out, 8 * n, hipMemcpyDeviceToHost));
  hipFree(d_rows);
  hipFree(d_out);
  return GS_OK;
}
