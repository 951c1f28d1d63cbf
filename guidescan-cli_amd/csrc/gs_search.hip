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

#include "gs_kernels.h"

/* ---- search: one wavefront per (guide, strand) ----------------------------
 * Two LDS stacks per wave share one 3.5 KiB array (STACK_ENTRIES nodes): X (grows up) holds "single-symbol" nodes -
 * mismatch budget spent (index.hpp:230 returns before the substitution loop) or a fixed PAM
 * base - which need Occ of one base and have at most one child; G (grows down) holds nodes
 * that still branch (k < m), PAM 'N' wildcards and PAM fan-out nodes.  ~89 % of all nodes are
 * X nodes, and an X iteration costs ~1/4 of the instructions of a G one.
 * CNT: count the distinct 64-byte lines every load instruction asks for (bench.py's algorithmic
 * bytes of THIS algorithm); the timed kernel is the CNT = false instantiation. */
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

/* The publishing of a heavy pass, out of line (SHARE = 2: the plain form's loops keep their registers; the pass is rare there).
 * Everything it needs comes by value - a reference to the kernel's argument struct would put the struct on the stack.
 * Returns true when the pass went into the queue (the caller's pass is done), false when the queue had no room. */
struct gs_pub_args {
  uint4 *shq;
  uint32_t *shq_ctl, *shq_ready, *sh_list;
  uint32_t shq_cap, sh_max, share_max;
};
__device__ __noinline__ bool gs_publish_pass(const gs_pub_args q, uint32_t *wmisc, const uint32_t take, const uint4 mine, const uint32_t excl,
                                             const uint32_t slot, const uint32_t item, const uint32_t side_tab) {
  const uint32_t lane = lane_id();
  uint32_t sid = __builtin_amdgcn_readfirstlane(wmisc[3]);
  if (sid == SH_NONE) {
    uint32_t s = 0;
    if (lane == 0) s = atomicAdd(&q.shq_ctl[96], 1u);
    sid = __builtin_amdgcn_readfirstlane(s);
    if (lane == 0) {
      wmisc[3] = sid;
      if (sid < q.sh_max) q.sh_list[sid] = slot;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  if (sid >= q.sh_max) return false;
  const uint32_t pid = lane < take ? excl / q.share_max : 0u;
  const uint32_t np = __builtin_amdgcn_readlane(pid, (int)(take - 1u)) + 1u;
  uint32_t qb = 0;
  if (lane == 0) qb = atomicAdd(&q.shq_ctl[0], np);
  qb = __builtin_amdgcn_readfirstlane(qb);
  if (qb + np <= q.shq_cap) {
    const uint32_t ppid = dpp_or_zero<0x138>(pid + 1u);
    const bool first = lane < take && ppid != pid + 1u;
    const uint64_t bm = __ballot(first);
    const uint64_t upto = (2ull << lane) - 1ull;
    const uint32_t start = 63u - (uint32_t)__builtin_clzll((bm & upto) | 1ull);
    const uint64_t above = bm & ~upto;
    const uint32_t nxt = above ? (uint32_t)__builtin_ctzll(above) : take;
    uint4 *pk = q.shq + (size_t)(qb + pid) * SHQ_PKG;
    if (lane < take) st16_agent(pk + 1u + (lane - start), mine);
    if (first) st16_agent(pk, make_uint4(item, sid, side_tab | ((nxt - start) << 8), 0u));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (first) st_agent(q.shq_ready + qb + pid, 1u);
    return true;
  }
  if (qb < q.shq_cap && lane < q.shq_cap - qb && lane < np) {
    st16_agent(q.shq + (size_t)(qb + lane) * SHQ_PKG, make_uint4(0u, 0u, 0u, 0u));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_agent(q.shq_ready + qb + lane, 1u);
  }
  return false;
}

/* SHARE: 0 - every item stays with its wave; 1 (HEAVY) - heavy verification passes are published as packages AND the waves
 * that ran out of items run them (one launch: a repeat-rich batch); 2 - published only: the waves leave when the items are
 * taken, and the packages are run by a launch of the HEAVY form that has no items of its own (gs_search_args::helper_only)
 * on a second, low-priority stream: it fills the slots the first launch's waves leave behind */
template <bool CNT, bool WALK, bool SPEC = false, int SHARE = 0>
__device__ __forceinline__ void k_search_body(const gs_search_args &a, uint4 *stk) {
  constexpr bool HEAVY = SHARE == 1, PUB = SHARE != 0;
  /* launch-wide facts every item tests, named once: as expressions on the argument struct at each use they cost the headline
   * launch 2.4 ms of 24.4 (335 -> 325 scalar values in spill lanes; profiles/r05_ab_compile_variants.txt, one box: the library
   * before this line 24.38 ms, with it 22.02).  The same facts as template constants (a form of the kernel for a batch's main
   * pass: nothing to append to, slots of one size, the arena on) allocate worse again: 23.01. */
  const bool k_append = a.append != 0u;
  const uint64_t *const k_slot_off = a.slot_off;
  const bool k_arena = a.arena != nullptr;
  const uint32_t k_cap = a.cap, k_n_items = a.n_items; /* (these two: 23.0 -> 22.7 ms on one box) */
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
  const bool sharing = PUB && a.shq != nullptr;
  if constexpr (SHARE == 2) { /* the launch with the items is resident: the launch without may wait for it */
    if (sharing && lane == 0) st_agent(&a.shq_ctl[65], 1u);
  }
  if constexpr (HEAVY) {
    /* a launch without items that got onto the chip BEFORE the one it serves must not hold its slots (its waves would wait
     * for waves that cannot start): it leaves, and the host runs what it left undone behind the other launch */
    if (sharing && a.helper_only && __builtin_amdgcn_readfirstlane(ld_agent(&a.shq_ctl[65])) == 0u) return;
  }
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
      if (HEAVY && a.helper_only) { /* (a launch without items: no visit to the counter - 8,192 of them are 0.1 ms on one word) */
        base = k_n_items;
      } else {
        if (lane == 0) base = atomicAdd(a.work, a.take);
        base = __builtin_amdgcn_readfirstlane(base);
      }
      if (base >= k_n_items) {
        if (!sharing) break; /* exit condition every wave reaches */
        if (lane == 0 && !a.helper_only) atomicAdd(&a.shq_ctl[64], 1u); /* this wave reserves no package any more */
        if constexpr (!HEAVY) break; /* (published only: the packages belong to the other launch) */
        items_done = true;
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
        item_end = base + a.take < k_n_items ? base + a.take : k_n_items;
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
      /* (no ticket is drawn when none can be served any more - every wave has left its items, every package reserved has its
       * ticket: the last visit of each of 8,192 waves to that one word was 0.1 ms at the end of every launch) */
      if (lane == 0 && !(ld_agent(&a.shq_ctl[64]) >= a.n_waves && ld_agent(&a.shq_ctl[32]) >= ld_agent(&a.shq_ctl[0]))) {
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
      if ((hz >> 8) == 0u || (hz >> 8) > WAVE || h_item >= k_n_items) continue; /* a filler for a reservation the queue had no room for */
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
    const uint32_t n_guides = k_n_items >> 1;
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
        if (k_arena && lane < 8u) a.cls[(size_t)slot * 8u + lane] = 0u;
      }
      continue;
    }
    uint32_t guard_left = a.max_iter; /* rounds this item's loops may still take (every outer step runs a counted inner loop) */
    const gs_strand_dev &sd = a.sd[strand];
    const uint4 *__restrict__ blocks = sd.blocks;
    const uint32_t npams = P ? gr_npams : 1u;
    const bool fanning = P > 0 && npams > 1u; /* a finished 20-mer fans out per PAM pattern */
    uint4 *out = a.slots + (k_slot_off ? (size_t)k_slot_off[slot] : (size_t)slot * k_cap);
    /* (a helper owns no slots: its records go to arena chunks of its own from the first one on) */
    const uint32_t item_cap = helper ? 0u : k_slot_off ? (uint32_t)(k_slot_off[slot + 1] - k_slot_off[slot]) : k_cap;
    uint32_t n_match = 0;
    if (k_append) n_match = __builtin_amdgcn_readfirstlane(a.counts[slot]);
    if (k_arena) {
      uint2 nc = make_uint2(0u, 0u);
      if (k_append) nc = a.nchunk[slot];
      if (lane < 4u) wmisc[lane] = lane == 0u ? nc.x : lane == 1u ? nc.y : lane == 2u ? 0u : SH_NONE;
      if (lane >= 4u && lane < 12u) wmisc[lane] = k_append ? a.cls[(size_t)slot * 8u + (lane - 4u)] : 0u;
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
        if (k_arena) {
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
        if (hi > item_cap && k_arena) {
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
          } else if (k_arena) {
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
      if (!WALK && !helper && R >= a.share_min) n_hpass++; /* (an item's own passes: what the next batch's choice of form counts - a helper's packages are those passes again) */
      if constexpr (SHARE == 2) {
        if (sharing && __builtin_expect(R >= a.share_min, 0)) {
          gs_pub_args q;
          q.shq = a.shq;
          q.shq_ctl = a.shq_ctl;
          q.shq_ready = a.shq_ready;
          q.sh_list = a.sh_list;
          q.shq_cap = a.shq_cap;
          q.sh_max = a.sh_max;
          q.share_max = a.share_max;
          const uint32_t cur_tab = (modeB || arow == nullptr) ? 3u : arow == a.pt[0][strand].rowid ? 0u : 1u;
          if (gs_publish_pass(q, wmisc, take, mine, excl, slot, item, (modeB ? 1u : 0u) | (cur_tab << 1))) return;
        }
      }
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
    if (k_arena) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t nch = __builtin_amdgcn_readfirstlane(wmisc[0]), last = __builtin_amdgcn_readfirstlane(wmisc[1]);
      /* (a shared item: its number instead of the last chunk - nobody appends to it - for k_share_dir/fix) */
      const uint32_t sid = __builtin_amdgcn_readfirstlane(wmisc[3]);
      if (lane == 0) a.nchunk[slot] = make_uint2(nch, (PUB && sid < a.sh_max) ? 0x80000000u | sid : last);
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
GS_DEF_SEARCH(k_search_fast_pd, false, false, true, GS_WAVES_EU_PD)
GS_DEF_SEARCH(k_search_count_pd, true, false, true, GS_WAVES_EU_PD)
/* HEAVY: the instantiations for a handle whose earlier batches showed items of thousands of records (a repeat-rich genome; m >= 5):
 * heavy verification passes are handed to the waves that ran out of items (gs_search_args::shq) and the second level of the
 * verification keeps GS_VU rows per lane in flight.  The same results; on a genome without such items the plain forms are faster
 * (1 M guides at m <= 3: 21 ms against 28-32 - the second level is rare there and its unrolled form costs instructions and registers). */
GS_DEF_SEARCH(k_search_heavy, false, false, false, GS_WAVES_EU_HEAVY, 1)
GS_DEF_SEARCH(k_search_heavy_pd, false, false, true, GS_WAVES_EU_HEAVY, 1)
/* the plain forms that publish their heavy passes and leave (SHARE = 2) */
GS_DEF_SEARCH(k_search_pub, false, false, false, GS_WAVES_EU_FAST, 2)
GS_DEF_SEARCH(k_search_pub_pd, false, false, true, GS_WAVES_EU_PD, 2)

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
  __shared__ uint32_t s_pair[17];
  if (threadIdx.x < 17u) s_pair[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t g0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inside = g0 < a.n;
  const uint32_t g = inside ? g0 : a.n - 1u; /* (the lanes beyond the batch redo its last guide and store nothing) */
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
  if (inside) {
    if (a.chunk == 0) {
      if (!r.valid) atomicAdd(a.n_invalid, 1u);
      if (a.flags) a.flags[g] = r.valid ? 0u : 1u;
    }
    a.out[g] = r;
  }
  if (a.pair_hist != nullptr && a.P >= 2u) {
    /* which PAM-pair tables would serve this batch: one LDS atomic per wave and distinct pair, then one atomic per
     * WORKGROUP and pair to memory - a batch's patterns end in one or two pairs, and one word of memory takes ~90
     * atomics per microsecond: one per wave of 64 guides was 0.18 of this kernel's 0.19 ms at a million guides */
    for (uint32_t j = 0; j < 4u; j++) {
      const bool has = inside && r.valid && j < r.npams;
      const uint32_t c0 = (r.pam[j] >> (3u * (a.P - 2u))) & 7u, c1 = (r.pam[j] >> (3u * (a.P - 1u))) & 7u;
      const uint32_t code = (c0 > 3u || c1 > 3u) ? 16u : (c0 | (c1 << 2));
      uint64_t todo = __ballot(has);
      while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const uint32_t c = (uint32_t)__shfl((int)code, l);
        const uint64_t same = __ballot(has && code == c);
        if ((int)lane_id() == l) atomicAdd(&s_pair[c], (uint32_t)__popcll(same));
        todo &= ~same;
      }
    }
    __syncthreads();
    if (threadIdx.x < 17u && s_pair[threadIdx.x]) atomicAdd(&a.pair_hist[threadIdx.x], s_pair[threadIdx.x]);
  }
}

void gs_launch_prepare(const gs_prep_args &pa, hipStream_t st) {
  hipLaunchKernelGGL(k_prepare, dim3((pa.n + 1023) / 1024), dim3(1024), 0, st, pa);
}
