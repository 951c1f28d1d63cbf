/*
 * gs_seed.hip -- the table-seeded search of a batch whose every PAM pattern has its PAM-pair and deep tables (NGG, NAG,
 * TTN + --start ...: what k_search_fast_pd served), as TWO launches over the same (guide, strand) items:
 *
 *   k_describe   per guide, once: everything an item derives from its guide alone - table indices of its exact k-mers,
 *                the context-mask selection words, which tables its patterns go through - as one 64-byte descriptor,
 *                and the two keys the items are scheduled by
 *   k_sched_*    counting sorts of the guides by those keys
 *   k_seed_b     the other strand's seeds (deep tables, index.hpp:182-248 seen from the far end) + the literal-N
 *                window list (index.hpp:139-149); items in the order of their last guide symbols
 *   k_seed_a     this strand's seeds (PAM-pair tables), appending to the same match slots; items in the order of their
 *                first guide symbols
 *
 * Why two launches.  k_search_body spent 32 % of the headline launch before an item's first seed (profiles/
 * r05_k_search_decomposition.txt: ~30 dependent round trips, 834 scalar instructions of loops over the guide's symbols)
 * and kept 326 scalar values in spill lanes.  Here an item starts from one descriptor, each launch holds one side's
 * state, and - the reason for the ORDER - an item's probes are not spread over the whole table: a seed without a
 * substitution in R (the last L-k guide symbols) reads the 4^(k-2-(L-k)) lines of the deep table that its R names
 * (154 of an item's 172 lines at m <= 3, a 256-KB piece), a seed without one in X (the first symbols) the 4^(k-|X|)
 * entries of the pair table that its X names (a 32-KB piece).  Items that share R (or X) and run at the same time on
 * the same XCD find those lines in its L2: each XCD takes a contiguous piece of the sorted order (8 work counters,
 * HW_REG_XCC_ID), and steals from the others' when its own runs dry.
 *
 * A seed's path (its substitutions as match.sequence spells them) is not carried through the queue: a hit is rare
 * (13 per guide at m <= 3 against 1,838 probes per item) and rebuilds it from the recipe's number.
 *
 * Results are those of k_search_fast_pd record for record (slot layout, arena chunks, class counts); which form a
 * handle runs is the host's choice (run_search, GS_SEED_FORM).
 */
#include "gs_kernels.h"

#define SEED_VQ VQ_CAP
#define SEED_LDS (SEED_VQ + 32 + 24 + 6) /* uint4 per wave: queue, owner markers, substitution table (96 words), state (24 words) */

__device__ __forceinline__ uint32_t seed_xcc_id() {
  /* HW_REG_XCC_ID (20), bits 3:0: the XCD this wave runs on (MI355X_MICROARCH.md, workgroup dispatch) */
  return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;
}
template <typename T>
__device__ __forceinline__ const T *seed_sgprs(const T *p) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return (const T *)(const T __attribute__((address_space(1))) *)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t seed_codes16(const uint32_t T, const uint32_t Q) { /* gs_search.hip: path_codes16 */
  const uint32_t x = T ^ Q;
  const uint32_t ne = (x | (x >> 1)) & 0x55555555u;
  const uint32_t nq = ~T & Q;
  const uint32_t lt = ((nq >> 1) | ((~x >> 1) & nq)) & 0x55555555u;
  return (T + lt) & (ne * 3u);
}
__device__ __forceinline__ uint32_t seed_rev16(const uint32_t x) {
  const uint32_t r = __brev(x);
  return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}

/* ---- per guide: the descriptor (process.hpp:51-63 gave the record; this is what k_search_body computed per item) ----
 * One function for the device (k_describe) and the host (gs_debug_guide_descriptor: tests pin every field against a
 * restatement in numpy, tests/test_seed_descriptor.py). */
struct gs_describe_shape {
  uint32_t L, P, k, x_len, n_pt;
  uint32_t code[2];
};
__host__ __device__ inline gs_guide_desc gs_describe_one(const gs_guide_rec &r, const gs_describe_shape &a, uint32_t g) {
  gs_guide_desc d;
  const uint64_t q = r.q;
  const uint32_t L = a.L, P = a.P, k = a.k, sx = a.x_len, nYb = L - sx;
  auto sym = [&](uint32_t t) -> uint32_t { return (uint32_t)(q >> (2u * t)) & 3u; };
  d.q_lo = (uint32_t)q;
  d.q_hi = (uint32_t)(q >> 32);
  for (uint32_t j = 0; j < 4; j++) d.pam[j] = r.pam[j];
  const uint32_t npams = r.valid ? (r.npams < 4u ? r.npams : 4u) : 0u;
  uint32_t pslots = 0, bits = 0;
  for (uint32_t pj = 0; pj < npams; pj++) {
    const uint32_t pw = r.pam[pj];
    const uint32_t c0 = (pw >> (3u * (P - 2u))) & 7u, c1 = (pw >> (3u * (P - 1u))) & 7u, cn = pw & 7u;
    const uint32_t code = c0 | (c1 << 2);
    const uint32_t bslot = (a.n_pt > 1u && code == a.code[1]) ? 1u : 0u;
    pslots |= 1u << bslot;
    bits |= bslot << (8u + pj);
    bits |= (cn == 4u ? 15u : 1u << (3u - cn)) << (12u + 4u * pj);
  }
  uint32_t bsel_z = 0, bsel_w = 0, n_bpairs = 0;
  for (uint32_t j = 0; j < 4u; ++j) { /* the deep tables' masks: pairs at context offsets 0, 2, 4, 6 */
    const uint32_t o = 2u * j;
    if (o + 1u < sx) {
      const uint32_t v = (3u - sym(sx - 1u - o)) | ((3u - sym(sx - 2u - o)) << 2);
      if (j < 2u)
        bsel_z |= 1u << (16u * j + v);
      else
        bsel_w |= 1u << (16u * (j - 2u) + v);
      n_bpairs++;
    }
  }
  d.meta = npams | (pslots << 3) | (n_bpairs << 5) | bits;
  uint32_t pidx0 = 0;
  for (uint32_t t = 0; t < k; ++t) pidx0 |= sym(t) << (2u * (k - 1u - t));
  d.pidx0 = pidx0;
  uint32_t pidxg = 0;
  for (uint32_t y = 0; y < nYb; ++y) pidxg |= (3u - sym(L - 1u - y)) << (2u * (nYb - 1u - y));
  d.pidxg = pidxg;
  uint32_t qrem_b = 0;
  for (uint32_t j = 0; j < sx; ++j) qrem_b |= (3u - sym(sx - 1u - j)) << (2u * j);
  d.qrem_b = qrem_b;
  d.bsel_z = bsel_z;
  d.bsel_w = bsel_w;
  const uint32_t gA = L - k, gmaskA = gA >= 16u ? 0xFFFFFFFFu : ((1u << (2u * gA)) - 1u);
  const uint32_t qremA = (uint32_t)(q >> (2u * k)) & gmaskA;
  uint32_t qhot = 0;
  for (uint32_t j = 0; j < 6u && j < gA; ++j) qhot |= 1u << (4u * j + ((qremA >> (2u * j)) & 3u));
  d.qhot = qhot;
  /* scheduling keys: X as the pair table's index spells it (its highest symbols), R as the deep table's */
  const uint32_t xa = sx < 8u ? sx : 8u, rb = (L - k) < 8u ? (L - k) : 8u;
  d.key_a = xa ? pidx0 >> (2u * (k - xa)) : 0u;
  d.key_b = (rb && rb <= nYb) ? pidxg >> (2u * (nYb - rb)) : 0u;
  d.guide = g;
  return d;
}
__global__ void k_describe(gs_describe_args a) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.n) return;
  gs_describe_shape sh;
  sh.L = a.L;
  sh.P = a.P;
  sh.k = a.k;
  sh.x_len = a.x_len;
  sh.n_pt = a.n_pt;
  sh.code[0] = a.code[0];
  sh.code[1] = a.code[1];
  gs_guide_desc d = gs_describe_one(a.guides[g], sh, g);
  /* literal-N windows within reach (index.hpp:139-149; what k_seed_b reports straight from the list): does the strand have
   * one at all for this guide?  Almost never - and then the item does not read the list (four dependent reads and 150
   * vector instructions per item, 12 % of k_seed_b).  Through the 5-symbol buckets where the batch has them. */
  if ((d.meta & 7u) != 0u) {
    const uint64_t q = ((uint64_t)d.q_hi << 32) | d.q_lo, lmask = (1ull << (2u * a.L)) - 1ull;
    for (uint32_t s = 0; s < 2u; ++s) {
      bool any = false;
      const uint32_t nc = a.n_cand[s];
      if (a.cand_ids[s] != nullptr) {
        for (uint32_t sg = 0; sg < 4u && !any; ++sg) {
          const uint32_t *off = a.cand_off[s] + 1025u * sg + ((uint32_t)(q >> (10u * sg)) & 1023u);
          for (uint32_t c = off[0]; c < off[1] && !any; ++c) {
            const uint4 ce = a.cand[s][a.cand_ids[s][(size_t)sg * nc + c]];
            const uint64_t x = ((((uint64_t)ce.y << 32) | ce.x) ^ q);
            any = (uint32_t)__popcll((x | (x >> 1)) & 0x5555555555555555ull & lmask) <= a.m;
          }
        }
      } else {
        for (uint32_t c = 0; c < nc && !any; ++c) {
          const uint4 ce = a.cand[s][c];
          const uint64_t x = ((((uint64_t)ce.y << 32) | ce.x) ^ q);
          any = (uint32_t)__popcll((x | (x >> 1)) & 0x5555555555555555ull & lmask) <= a.m;
        }
      }
      if (any) d.meta |= 1u << (28u + s);
    }
  }
  a.desc[g] = d;
  if (a.hist != nullptr) {
    atomicAdd(&a.hist[d.key_a], 1u);
    atomicAdd(&a.hist[65536u + d.key_b], 1u);
  }
}
/* host only: the descriptor of one packed guide record for a batch shape (k = table depth, x_len = |X|, the pair codes of
 * the table slots), as sixteen words */
extern "C" gs_status gs_debug_guide_descriptor(uint64_t q, const uint32_t pam[4], uint32_t npams, uint32_t valid, uint32_t L, uint32_t P,
                                               uint32_t k, uint32_t x_len, uint32_t n_pt, const uint32_t code[2], uint32_t out[16]) {
  if (!pam || !code || !out || L < 1 || L > 31 || k < 4 || k > 16 || k > L || x_len < 1 || x_len >= L || P < 2 || P > 8) return GS_ERR_ARG;
  gs_guide_rec r;
  r.q = q;
  for (int j = 0; j < 4; j++) r.pam[j] = pam[j];
  r.npams = npams;
  r.valid = valid;
  gs_describe_shape sh;
  sh.L = L;
  sh.P = P;
  sh.k = k;
  sh.x_len = x_len;
  sh.n_pt = n_pt;
  sh.code[0] = code[0];
  sh.code[1] = code[1];
  const gs_guide_desc d = gs_describe_one(r, sh, 0u);
  static_assert(sizeof(gs_guide_desc) == 64, "one descriptor is sixteen words");
  memcpy(out, &d, 64);
  return GS_OK;
}
/* exclusive scans of the two histograms (65,536 bins each): one workgroup per histogram */
__global__ void __launch_bounds__(1024) k_sched_scan(uint32_t *hist) {
  __shared__ uint32_t part[1024];
  uint32_t *h = hist + 65536u * blockIdx.x;
  const uint32_t t = threadIdx.x;
  uint32_t s = 0;
  for (uint32_t i = 0; i < 64u; i++) s += h[64u * t + i];
  part[t] = s;
  __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1) {
    const uint32_t v = t >= o ? part[t - o] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - s;
  for (uint32_t i = 0; i < 64u; i++) {
    const uint32_t c = h[64u * t + i];
    h[64u * t + i] = run;
    run += c;
  }
}
__global__ void k_sched_scatter(const gs_guide_desc *desc, uint32_t n, uint32_t *cursor, gs_guide_desc *desc_a, gs_guide_desc *desc_b) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const gs_guide_desc d = desc[g];
  desc_a[atomicAdd(&cursor[d.key_a], 1u)] = d;
  desc_b[atomicAdd(&cursor[65536u + d.key_b], 1u)] = d;
}

/* ---- which form of the search this batch needs, from the batch itself: guides whose own k-mer - the first k symbols
 * they consume, unsubstituted - heads an interval of `thresh` rows or more in either strand's table.  Such a guide sits in
 * a repeat family: its seeds verify 10^4 .. 10^6 rows, the passes the sharing forms hand to idle waves (run_search).  One
 * table read per guide and strand; the handle used to learn this from the PREVIOUS batch's count of heavy passes, so the
 * first batch of a shape - and any batch unlike its predecessor - met its giant items with one wave each. ---- */
__global__ void k_estimate_heavy(const gs_guide_rec *guides, uint32_t n, const uint4 *ptab0, const uint4 *ptab1, uint32_t k,
                                 uint32_t thresh, uint32_t *out) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  bool heavy = false;
  uint32_t most = 0;
  if (g < n && guides[g].valid) {
    const uint64_t q = guides[g].q;
    uint32_t pidx = 0;
    for (uint32_t t = 0; t < k; ++t) pidx |= ((uint32_t)(q >> (2u * t)) & 3u) << (2u * (k - 1u - t));
    const uint32_t c0 = ptab0[pidx].y & 0x7FFFFFFFu, c1 = ptab1[pidx].y & 0x7FFFFFFFu;
    heavy = c0 >= thresh || c1 >= thresh;
    most = c0 > c1 ? c0 : c1;
  }
  const uint64_t b = __ballot(heavy);
  if (b) {
    if (lane_id() == 0) atomicAdd(out, (uint32_t)__popcll(b));
    if (heavy) atomicMax(out + 1, most); /* the largest interval a guide's own k-mer heads */
  }
}
gs_status gs_estimate_heavy(gs_index *ix, const gs_guide_rec *guides, uint32_t n, uint32_t thresh, uint32_t *d_out, hipStream_t st,
                            uint32_t n_heavy[2]) {
  n_heavy[0] = n_heavy[1] = 0;
  if (!ix->pt_k || !ix->strand[0].ptab || !ix->strand[1].ptab || !n) return GS_OK;
  GS_HIP(hipMemsetAsync(d_out, 0, 8, st));
  hipLaunchKernelGGL(k_estimate_heavy, dim3((n + 255) / 256), dim3(256), 0, st, guides, n, (const uint4 *)ix->strand[0].ptab,
                     (const uint4 *)ix->strand[1].ptab, ix->pt_k, thresh, d_out);
  GS_HIP(hipMemcpyAsync(n_heavy, d_out, 8, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  return GS_OK;
}

/* ---- the seeding launches ------------------------------------------------------------------------------------------
 * SIDE 0: the other strand's seeds + the window list; SIDE 1: this strand's seeds, appending.  CNT: the request tally. */
template <bool CNT, int SIDE>
__device__ __forceinline__ void k_seed_body(const gs_search_args &a, uint4 *lds) {
  constexpr bool modeB = SIDE == 0;
  constexpr uint32_t PB = 7u, KSH = 61u, PSG = 50u + PB, PSP = 49u + PB;
  constexpr uint64_t PMASK = (1ull << (52u + PB)) - 1ull;
  const uint32_t lane = lane_id();
  uint4 *const vq = lds;                                   /* queued seeds: {first row, meta, seed number, -} */
  uint2 *const own2 = (uint2 *)(vq + SEED_VQ);              /* owner markers of a pass, two per lane */
  uint32_t *const own = (uint32_t *)own2;
  uint32_t *const dtab = (uint32_t *)(vq + SEED_VQ + 32);   /* substitution table: entry 4 step + digit = its xor on the table index */
  uint32_t *const wmisc = dtab + 96;                        /* [0..2] overflow chunks {taken, last, the one before}, [4..11] matches per class, [16..18] chunk reserve */
  const uint32_t L = a.L, P = a.P, m = a.m, k = a.pt_k, sx = a.x_len;
  const bool k_arena = a.arena != nullptr;
  const bool k_append = modeB ? a.append != 0u : true;
  const uint64_t *const k_slot_off = a.slot_off;
  const uint32_t k_cap = a.cap, n_items = a.n_items, ng = n_items >> 1;
  uint32_t *const xwork = a.xwork + (modeB ? 0u : 256u);
  unsigned long long n_ovf = 0;
  uint32_t n_fail = 0, n_hpass = 0, n_two = 0;
  bool bailed = false;
  uint32_t c_tab = 0, c_c16 = 0, c_ctx = 0, c_isa = 0, c_rec = 0;
  auto count_lines = [&](uint32_t &acc, bool act, const void *p) __attribute__((always_inline)) {
    if constexpr (CNT) {
      const uint32_t line = (uint32_t)((uintptr_t)p >> a.cnt_shift);
      const uint32_t prev = (uint32_t)__shfl_up((int)line, 1);
      const int pact = __shfl_up((int)act, 1);
      const bool fresh = act && (lane == 0u || !pact || prev != line);
      acc += (uint32_t)__popcll(__ballot(fresh));
    }
  };
  if (lane == 0) {
    wmisc[16] = 0u;
    wmisc[17] = 0u;
    wmisc[18] = 1u;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

  /* items: each XCD draws from its own eighth of the schedule, then from the others'.  The draw for the NEXT visit is issued
   * when a visit's last item starts (the counter's answer comes from the memory side of the fabric: ~2 us that nothing waits for
   * any more), and the descriptors lie in the schedule's order, so an item is its position and one scalar load */
  const gs_guide_desc *const descs = modeB ? a.desc_b : a.desc_a;
  uint32_t cur = seed_xcc_id(), tried = 0;
  uint32_t item_next = 0, item_end = 0;
  uint32_t pend = 0xFFFFFFFFu; /* lane 0: the answer of the draw that is on its way */
  bool pending = false;
  for (;;) {
    if (item_next == item_end) {
      bool none = false;
      for (;;) {
        const uint32_t lo = (uint32_t)(((uint64_t)cur * n_items) >> 3), hi = (uint32_t)(((uint64_t)(cur + 1u) * n_items) >> 3);
        if (!pending) {
          pend = 0xFFFFFFFFu;
          /* (a piece that is not this wave's own is looked at before it is drawn from: at the end of a launch every wave visits
           * every counter, and a load does not queue behind the others as an atomic does) */
          if (lane == 0 && (tried == 0u || ld_agent(&xwork[32u * cur]) < hi - lo)) pend = atomicAdd(&xwork[32u * cur], a.take);
        }
        pending = false;
        const uint32_t base = __builtin_amdgcn_readfirstlane(pend);
        if (base < hi - lo) {
          item_next = lo + base;
          item_end = (hi - lo) - base < a.take ? hi : item_next + a.take;
          break;
        }
        cur = (cur + 1u) & 7u;
        if (++tried == 8u) {
          none = true;
          break;
        }
      }
      if (none) break; /* exit condition every wave reaches: all eight pieces are taken */
    }
    const uint32_t pos = item_next++;
    if (item_next == item_end) { /* the visit's last item: the next draw goes out now */
      pend = 0xFFFFFFFFu;
      if (lane == 0) pend = atomicAdd(&xwork[32u * cur], a.take);
      pending = true;
    }
    const uint32_t strand = pos >= ng ? 1u : 0u;
    /* the descriptor: sixteen wave-uniform words, one scalar load */
    typedef uint32_t seed_u32x16 __attribute__((ext_vector_type(16)));
    const seed_u32x16 dp = *(const seed_u32x16 __attribute__((address_space(4))) *)(uintptr_t)(descs + (pos - strand * ng));
    const uint32_t guide = dp[15];
    const uint32_t slot = 2u * guide + strand;
    const uint32_t gw0 = dp[0], gw1 = dp[1];
    const uint64_t gr_q = ((uint64_t)gw1 << 32) | gw0;
    const uint32_t gr_pam0 = dp[2], gr_pam1 = dp[3], gr_pam2 = dp[4], gr_pam3 = dp[5];
    const uint32_t meta = dp[6];
    const uint32_t npams = meta & 7u;
    if (npams == 0u || bailed) {
      if (modeB && !a.append) {
        if (lane == 0) a.counts[slot] = 0;
        if (k_arena && lane < 8u) a.cls[(size_t)slot * 8u + lane] = 0u;
        if (k_arena && lane == 8u) a.nchunk[slot] = make_uint2(0u, 0u);
      }
      continue;
    }
    uint32_t guard_left = a.max_iter;
    const gs_strand_dev &sd = a.sd[strand];
    const gs_strand_dev &sv = a.sd[strand ^ 1u];
    uint4 *out = a.slots + (k_slot_off ? (size_t)k_slot_off[slot] : (size_t)slot * k_cap);
    const uint32_t item_cap = k_slot_off ? (uint32_t)(k_slot_off[slot + 1] - k_slot_off[slot]) : k_cap;
    uint32_t n_match = 0;
    if (k_append) n_match = __builtin_amdgcn_readfirstlane(a.counts[slot]);
    if (k_arena) {
      uint2 nc = make_uint2(0u, 0u);
      if (k_append) nc = a.nchunk[slot];
      if (lane < 3u) wmisc[lane] = lane == 0u ? nc.x : lane == 1u ? nc.y : 0u;
      if (lane >= 4u && lane < 12u) wmisc[lane] = k_append ? a.cls[(size_t)slot * 8u + (lane - 4u)] : 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }

    /* a record of the item: bit 0 of the key = the row sits v_rem symbols into the site (k_locate) */
    auto emit = [&](const bool em, const uint32_t row, const uint64_t cmeta, const uint32_t vflag) __attribute__((always_inline)) {
      const uint64_t be = __ballot(em);
      if (!be) return;
      const uint32_t hi = n_match + (uint32_t)__popcll(be);
      if (k_arena) {
        const uint32_t kk = (uint32_t)((cmeta >> KSH) & 7ull);
        uint32_t add = 0;
        for (uint32_t d = 0; d <= m; ++d) {
          const uint32_t c = (uint32_t)__popcll(__ballot(em && kk == d));
          add = lane == d ? c : add;
        }
        if (add) atomicAdd(&wmisc[4u + lane], add);
      }
      if (hi > item_cap && k_arena) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t nch = __builtin_amdgcn_readfirstlane(wmisc[0]);
        const uint32_t need = (hi - item_cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT;
        while (nch < need) {
          uint32_t id = 0;
          if (lane == 0) {
            uint32_t rn = wmisc[16];
            if (rn == wmisc[17]) {
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
            a.chunk_seq[id] = nch;
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
        const uint64_t key = ((uint64_t)((cmeta >> KSH) & 7ull) << 61) | ((uint64_t)strand << 60) | ((cmeta & PMASK) << (8u - PB)) | vflag;
        const uint4 rec = make_uint4((uint32_t)key, (uint32_t)(key >> 32), row, row);
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
    };

    const uint2 *const recs = modeB ? a.rec_b : a.rec_a8;
    const uint32_t nrec = modeB ? a.n_rec_b : a.n_rec_a8;
    /* the path of seed `sid` (a hit's lanes only): the substitutions of its recipe as codes at their guide positions
     * (index.hpp:230-247: the text base's place among the three other bases), on the other strand's side also the
     * pattern's fixed PAM symbols and the base the deep table's entry stands for under the N */
    auto seed_path = [&](const uint32_t sid) __attribute__((always_inline)) -> uint64_t {
      const uint2 rc = recs[sid & 0xFFFFFFu];
      const uint32_t n = rc.x & 7u;
      uint64_t f = (((uint64_t)rc.y << 32) | rc.x) >> 12;
      uint64_t path = 0;
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t fld = (uint32_t)f & 127u, s = fld >> 2, d = fld & 3u;
        f >>= 7;
        const uint32_t t = modeB ? L - 1u - s : s;
        const uint32_t qc = (uint32_t)(gr_q >> (2u * t)) & 3u;
        const uint32_t sym = (qc + 1u + d) & 3u;
        path |= (uint64_t)(1u + sym - (sym > qc ? 1u : 0u)) << (PSG - 2u * t);
      }
      if (modeB) {
        const uint32_t bx = (sid >> 24) & 3u, pj = (sid >> 26) & 3u;
        const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
        path |= (uint64_t)(bx != 0u ? 3u - bx : 4u) << (PSP - 2u * L);
        for (uint32_t u = 1; u < P; ++u) path |= (uint64_t)(((pw >> (3u * u)) & 7u) < 3u ? ((pw >> (3u * u)) & 7u) : 4u) << (PSP - 2u * L - 3u * u);
      }
      return path;
    };

    /* what the rows of this side's seeds are verified against */
    const uint16_t *v16 = modeB ? seed_sgprs(sv.ctx16) : nullptr;
    const uint32_t *vctx = modeB ? seed_sgprs(sv.ctx) : nullptr;
    const uint32_t *arow = nullptr; /* SIDE 1: a PAM-pair table's rows as the strand's suffix array numbers them */
    const uint32_t g = modeB ? sx : L - k; /* guide symbols among the remaining ones */
    const uint32_t gmask = g >= 16u ? 0xFFFFFFFFu : ((1u << (2u * g)) - 1u);
    const uint32_t qrem = modeB ? dp[9] : ((uint32_t)(gr_q >> (2u * k)) & gmask);

    /* ---- a row that passed both levels: its place in THIS strand's suffix array - through the other strand's SA and this
     * strand's ISA, or a PAM-pair table's row numbers -, its seed's path from the recipe, its record.  (Parking the hits in
     * LDS and resolving them 64 at a time, so that a pass does not wait for these dependent reads, changed nothing: 16.4 ms
     * against 16.3, profiles/r06_ab_seed_variants.txt - the waves that wait are covered by the SIMD's other seven) ---- */
    auto resolve = [&](const bool on, const uint32_t e_x, const uint32_t e_y, const uint32_t w, const uint32_t e_w) __attribute__((always_inline)) {
      const uint32_t n = (uint32_t)__popcll(__ballot(on));
      const uint4 e = make_uint4(e_x, e_y, w, e_w);
      if constexpr (modeB) {
        /* word symbol j is guide symbol g-1-j, complemented; the site's row on THIS strand through SA -> ISA */
        uint64_t mmeta = 0;
        uint32_t rowA = 0;
        if (on) {
          const uint32_t pB = sv.sa[e.x];
          const uint64_t sp = seed_path(e.y);
          rowA = sd.isa[(sd.n - 1u) - (pB - g) - (L + P)];
          const uint64_t gpath = (uint64_t)seed_codes16(~w & gmask, ~qrem & gmask) << (52u + PB - 2u * g);
          mmeta = ((uint64_t)e.w << KSH) | sp | gpath;
        }
        if constexpr (CNT) c_isa += 2u * n;
        emit(on, rowA, mmeta, 0u);
      } else {
        uint32_t orow = 0;
        uint64_t base_meta = 0;
        if (on) {
          orow = arow[e.x];
          const uint64_t gpath = ((uint64_t)seed_rev16(seed_codes16(w & gmask, qrem)) << 32) >> (12u - PB + 2u * k);
          base_meta = ((uint64_t)e.w << KSH) | seed_path(e.y) | gpath;
        }
        if constexpr (CNT) c_isa += n;
        for (uint32_t pj = 0; pj < npams; ++pj) {
          const uint32_t pw = pj == 0 ? gr_pam0 : pj == 1 ? gr_pam1 : pj == 2 ? gr_pam2 : gr_pam3;
          bool ok = on;
          uint64_t ppath = 0;
          for (uint32_t u = 0; u < P; ++u) {
            const uint32_t pc = (pw >> (3u * u)) & 7u;
            const uint32_t tb = (w >> (2u * (g + u))) & 3u;
            ok = ok && (pc == 4u || pc == tb);
            ppath |= (uint64_t)(tb < 3u ? tb : 4u) << (PSP - 2u * L - 3u * u);
          }
          if (!__ballot(ok)) continue;
          emit(ok, orow, base_meta | ppath, 1u);
        }
      }
    };

    /* ---- context verification of `take` queued seeds (k_search_body::verify, the plain form) ---- */
    auto verify = [&](const uint32_t take, uint4 *dsrc) __attribute__((always_inline)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint4 mine = make_uint4(0u, 0u, 0u, 0u);
      if (lane < take) mine = dsrc[lane];
      const uint32_t vcnt = (mine.y >> 17) & 0x3FFu;
      const uint32_t vgrp = (vcnt + 7u) >> 3;
      const uint32_t incl = wave_incl_sum(vgrp);
      const uint32_t R = __builtin_amdgcn_readlane(incl, WAVE - 1);
      if (!R) return;
      const uint32_t excl = incl - vgrp;
      if (R >= a.share_min) n_hpass++;
      if (lane < take) dsrc[lane].y = mine.y | excl;
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
        const uint32_t run = wave_incl_max(o1);
        const uint32_t prev = dpp_or_zero<0x138>(run);
        o0 = o0 > prev ? o0 : prev;
        o1 = o1 > prev ? o1 : prev;
        const uint32_t ow[2] = {o0, o1};
        uint4 wq[2], dsc[2];
        uint32_t row0[2], nrow[2], kkv[2], okm[2];
#pragma unroll
        for (uint32_t jj = 0; jj < 2u; ++jj) {
          const uint32_t grp = base + 2u * lane + jj;
          const bool on = grp < R;
          dsc[jj] = make_uint4(0u, 0u, 0u, 0u);
          if (on) dsc[jj] = dsrc[ow[jj] - 1u];
          const uint32_t r0 = (grp - (dsc[jj].y & 0x3FFFu)) << 3;
          const uint32_t cnt = (dsc[jj].y >> 17) & 0x3FFu;
          row0[jj] = dsc[jj].x + r0;
          nrow[jj] = on ? (cnt - r0 < 8u ? cnt - r0 : 8u) : 0u;
          kkv[jj] = (dsc[jj].y >> 14) & 7u;
          const uint32_t lo = (dsc[jj].y >> DSC_LO) & 7u;
          const uint32_t lo8 = lo > g - g8 ? lo - (g - g8) : 0u;
          okm[jj] = on ? (((2u << (m - kkv[jj])) - 1u) & ~((1u << lo8) - 1u)) : 0u;
          wq[jj] = make_uint4(0u, 0u, 0u, 0u);
          if (on) wq[jj] = load16_a2(v16 + row0[jj]);
        }
        if constexpr (CNT) {
          const bool on0 = nrow[0] != 0u, on1 = nrow[1] != 0u;
          const uint32_t f0 = (uint32_t)((uintptr_t)(v16 + row0[0]) >> 6), l0 = (uint32_t)(((uintptr_t)(v16 + row0[0]) + 15u) >> 6);
          const uint32_t f1 = (uint32_t)((uintptr_t)(v16 + row0[1]) >> 6), l1 = (uint32_t)(((uintptr_t)(v16 + row0[1]) + 15u) >> 6);
          const uint32_t mylast = on1 ? l1 : l0;
          const uint32_t pl = (uint32_t)__shfl_up((int)mylast, 1);
          const int pact = __shfl_up((int)(on0 || on1), 1);
          uint32_t add = 0;
          if (on0) add += ((lane == 0u || !pact || pl != f0) ? 1u : 0u) + (l0 != f0 ? 1u : 0u);
          if (on1) add += ((!on0 || l0 != f1) ? 1u : 0u) + (l1 != f1 ? 1u : 0u);
          for (int o = 32; o > 0; o >>= 1) add += (uint32_t)__shfl_xor((int)add, o);
          c_c16 += add;
        }
        uint32_t cm = 0u; /* candidate rows of this lane: bit 8 jj + r */
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
        cm &= ((1u << nrow[0]) - 1u) | (((1u << nrow[1]) - 1u) << 8);
        /* second level, one candidate row per lane per round: the full 16-symbol word decides */
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
          /* exception rows (the other strand's side only: a PAM-pair table holds no row with a symbol outside A,C,G,T
           * nearby): the true symbols decide; under a guide symbol nothing but a base can match or be substituted */
          bool excbad = false;
          if constexpr (modeB) {
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
                  for (uint32_t j = 0; j < g; ++j)
                    if (((uint32_t)(nb >> (4u * j)) & 15u) > 3u) excbad = true;
                }
              }
            }
          }
          const uint32_t xf = (w ^ qrem) & gmask;
          const uint32_t mmv = __popc((xf | (xf >> 1)) & 0x55555555u);
          const bool gok = has && !excbad && kv + mmv <= m && mmv >= ((dd.y >> DSC_LO) & 7u);
          const uint64_t bh = __ballot(gok);
          if (!bh) continue;
          resolve(gok, row, dd.z, w, kv + mmv);
        }
      }
    };

    /* the seeds of a step wait in the queue until a pass can be filled (a pass costs the same instructions for 10
     * seeds as for 64); intervals larger than a descriptor holds are queued piece by piece */
    uint32_t qn = 0;
    const uint32_t v_max = a.v_max;
    auto queue_step = [&](uint32_t rem, uint32_t first, const uint32_t ymeta, const uint32_t sid, const bool drain_all) __attribute__((always_inline)) {
      for (;;) {
        if (guard_left == 0u) {
          bailed = true;
          break;
        }
        guard_left--;
        const uint64_t bq = __ballot(rem != 0u);
        if (bq && qn + WAVE <= SEED_VQ) {
          const uint32_t rows = rem < v_max ? rem : v_max;
          if (rem != 0u) {
            uint32_t *e = (uint32_t *)(vq + qn + lanes_below(bq));
            *(uint2 *)e = make_uint2(first, ymeta | (rows << 17));
            e[2] = sid;
          }
          qn += __popcll(bq);
          first += rows;
          rem -= rows;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          continue;
        }
        if (qn >= VQ_DRAIN || (bq && qn != 0u) || (drain_all && qn != 0u)) {
          const uint32_t take = qn < WAVE ? qn : WAVE;
          qn -= take;
          verify(take, vq + qn);
          continue;
        }
        break;
      }
    };
    /* entry 4 step + digit: what substituting the digit-th other base at that step does to the table index (3: nothing) */
    auto fill_dtab = [&]() __attribute__((always_inline)) {
      const uint32_t nst = modeB ? L - sx : k;
      for (uint32_t e = lane; e < 4u * nst; e += WAVE) {
        const uint32_t s = e >> 2, d = e & 3u;
        const uint32_t t = modeB ? L - 1u - s : s;
        const uint32_t qc = (uint32_t)(gr_q >> (2u * t)) & 3u;
        const uint32_t sym = (qc + 1u + d) & 3u;
        dtab[e] = d == 3u ? 0u : (qc ^ sym) << (2u * (nst - 1u - s)); /* (the other strand's k-mer holds the complements: the xor stays) */
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    /* the recipe's substitutions applied to the exact index.  A recipe's unused fields hold 3 = "step 0, digit 3" = nothing
     * (gs_recipes.hip: recipe_word), so the first three fields are read whatever the recipe's count, side by side; the other
     * four only by a batch whose budget allows more than three substitutions, in a round some lane needs */
    auto apply_recipe = [&](const uint2 rc, uint32_t pidx) __attribute__((always_inline)) -> uint32_t {
      const uint32_t w0 = __builtin_amdgcn_alignbit(rc.y, rc.x, 12); /* fields 0..2: bits 12..32 of the word */
      pidx ^= dtab[w0 & 127u] ^ dtab[(w0 >> 7) & 127u] ^ dtab[(w0 >> 14) & 127u];
      if (m > 3u && __ballot((rc.x & 7u) > 3u) != 0ull) {
        const uint32_t w1 = rc.y >> 1; /* fields 3..6: bits 33..60 */
        pidx ^= dtab[w1 & 127u] ^ dtab[(w1 >> 7) & 127u] ^ dtab[(w1 >> 14) & 127u] ^ dtab[(w1 >> 21) & 127u];
      }
      return pidx;
    };
    const uint32_t keep = (a.dbg_skip & 3u) ? 0u : 0xFFFFFFFFu; /* (timing experiments: GS_DBG_SKIP = 1, 2 keep no seed) */

    if constexpr (modeB) {
      /* ---- literal-N windows within reach (index.hpp:139-149): the tables hold no row with a symbol outside A,C,G,T
       * next to it, so every such site of the item comes from this list (k_search_body, two-sided seeding) ---- */
      {
        const uint32_t ncand = a.n_cand[strand];
        const uint64_t lmask = (1ull << (2u * L)) - 1ull;
        const uint32_t *cids = a.cand_ids[strand];
        /* (k_describe has looked: the strand has a window within the budget of this guide's symbols, or the list is not read) */
        const uint32_t nseg = !((meta >> (28u + strand)) & 1u) ? 0u : cids != nullptr ? 4u : 1u;
        for (uint32_t sg = 0; sg < nseg; ++sg) {
          uint32_t s0 = 0, s1 = ncand;
          if (cids != nullptr) {
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
            for (uint32_t e = 0; e < sg; ++e) in = in && ((uint32_t)(x >> (10u * e)) & 1023u) != 0u;
            const uint64_t nz = (x | (x >> 1)) & 0x5555555555555555ull & lmask;
            const uint32_t tot = __popcll(nz);
            const bool mine = in && tot <= m;
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
              emit(ok, rowA, ((uint64_t)tot << KSH) | gpath | ppath, 0u);
            }
          }
        }
      }
      /* ---- the other strand's seeds: per pattern one pass over (recipe, base under the N), four lanes per recipe reading the
       * four entries of one deep-table line ---- */
      const uint32_t pidxg = dp[8], bsel_z = dp[10], bsel_w = dp[11], n_bpairs = (meta >> 5) & 7u;
      const uint32_t nlanes = 4u * nrec;
      fill_dtab();
      uint2 rc_next = make_uint2(0u, 0u);
      if (lane < nlanes) rc_next = recs[lane >> 2];
      for (uint32_t bpj = 0; bpj < npams && !bailed; ++bpj) {
        const uint32_t bslot = (meta >> (8u + bpj)) & 1u, bxset = (meta >> (12u + 4u * bpj)) & 15u;
        const uint4 *bdeep = seed_sgprs(a.pt[bslot][strand ^ 1u].deep);
        for (uint32_t bc0 = 0; bc0 < nlanes; bc0 += WAVE) {
          const uint32_t idx = bc0 + lane, ri = idx >> 2, bx = idx & 3u;
          const bool act = idx < nlanes && ((bxset >> bx) & 1u) != 0u;
          const uint2 rc = rc_next; /* (lanes beyond the list hold zeros: they load nothing) */
          {
            const uint32_t nidx = (bc0 + WAVE >= nlanes ? 0u : bc0 + WAVE) + lane;
            rc_next = make_uint2(0u, 0u);
            if (nidx < nlanes) rc_next = recs[nidx >> 2];
          }
          count_lines(c_rec, idx < nlanes, recs + ri);
          const uint32_t jb = rc.x & 7u, lo = (rc.x >> 3) & 7u;
          const uint32_t pidx = apply_recipe(rc, pidxg);
          const uint4 *ep = bdeep + ((size_t)pidx << 2) + bx;
          uint4 ent = make_uint4(0u, 0u, 0u, 0u);
          if (act) ent = *ep;
          count_lines(c_tab, act, ep);
          /* fewer of the query's pairs left of the rows than the budget for X can break: no row can match (an entry with
           * exception rows skips the test) */
          const uint32_t need = n_bpairs > m - jb ? n_bpairs - (m - jb) : 0u;
          const uint32_t intact = (uint32_t)__popc(ent.z & bsel_z) + (uint32_t)__popc(ent.w & bsel_w);
          const uint32_t y = ((int32_t)ent.y < 0 || intact >= need) ? ent.y & keep : 0u;
          queue_step(y & 0x7FFFFFFFu, ent.x, (jb << 14) | (lo << DSC_LO) | ((y >> 31) << DSC_EXC), ri | (bx << 24) | (bpj << 26),
                     bc0 + WAVE >= nlanes && bpj + 1u == npams);
          if (bailed) break;
        }
      }
    } else {
      /* ---- this strand's seeds through the PAM-pair tables of the item's patterns (one pass of the recipes per table) ---- */
      const uint32_t pidx0 = dp[7], qhot = dp[12];
      const uint32_t gmask13 = g >= 13u ? 0x3FFFFFFu : gmask;
      uint32_t pslots = (meta >> 3) & 3u;
      fill_dtab();
      uint2 rc_ahead = make_uint2(0u, 0u);
      if (lane < nrec) rc_ahead = recs[lane];
      while (pslots != 0u && !bailed) {
        const uint32_t s = (pslots & 1u) ? 0u : 1u;
        pslots &= ~(1u << s);
        const gs_pairtab_dev &p = a.pt[s][strand];
        const uint2 *atab8 = seed_sgprs(p.tab), *arot8 = seed_sgprs(p.rot);
        const uint32_t arot_first = p.rot_first;
        v16 = seed_sgprs(p.c16);
        vctx = seed_sgprs(p.ctx);
        arow = seed_sgprs(p.rowid);
        for (uint32_t spos = 0; spos < nrec; spos += WAVE) {
          const uint32_t l = spos + lane;
          const bool act = l < nrec;
          const uint2 rc = rc_ahead; /* (lanes beyond the list hold zeros: they load nothing) */
          {
            const uint32_t nl = (spos + WAVE >= nrec ? 0u : spos + WAVE) + lane; /* past the end: the next table's pass */
            rc_ahead = make_uint2(0u, 0u);
            if (nl < nrec) rc_ahead = recs[nl];
          }
          count_lines(c_rec, act, recs + l);
          const uint32_t kk = rc.x & 7u;
          const uint32_t pidx = apply_recipe(rc, pidx0);
          const uint2 *ep = atab8 + pidx;
          const uint32_t rs = (rc.x >> 7) & 31u;
          const bool rot = (rc.x & 64u) != 0u && rs >= arot_first;
          if (__ballot(rot) != 0ull) {
            if (rot) {
              /* the copy rotated at step rs: that step's symbol and everything after it swap places */
              const uint32_t sh = 2u * (k - 1u - rs);
              const uint32_t ridx = ((pidx >> (sh + 2u)) << (sh + 2u)) | ((pidx & ((1u << sh) - 1u)) << 2) | ((pidx >> sh) & 3u);
              ep = arot8 + (((size_t)(rs - arot_first) << (2u * k)) + ridx);
            }
          }
          uint2 e8 = make_uint2(0u, 0u);
          if (act) e8 = *ep;
          count_lines(c_tab, act, ep);
          uint32_t first = e8.x, ecnt = e8.y & 63u;
          const uint32_t filt = e8.y >> 6, bl = m - kk;
          /* one row: its own context symbols against the query's; several: a query symbol none of the rows shows at its
           * position is a substitution in every row (an empty entry keeps nothing whatever this says) */
          const uint32_t xf = (filt ^ qrem) & gmask13;
          const uint32_t tw = ecnt == 1u ? ((xf | (xf >> 1)) & 0x55555555u) : (qhot & ~filt);
          const bool hopeless = (uint32_t)__popc(tw) > bl;
          if (__ballot(ecnt == GS_PT_BIG && !hopeless) != 0ull) { /* 63 rows and more: the count sits in a header slot in front of them */
            if (ecnt == GS_PT_BIG && !hopeless) {
              ecnt = arow[first];
              first += 1u;
            }
            if constexpr (CNT) c_isa += (uint32_t)__popcll(__ballot(true));
          }
          const uint32_t rem = hopeless ? 0u : ecnt & keep;
          queue_step(rem, first, kk << 14, l, spos + WAVE >= nrec);
          if (bailed) break;
        }
      }
      n_two++;
    }

    if (lane == 0) a.counts[slot] = n_match;
    if (!modeB && n_match > item_cap) n_ovf++;
    if (k_arena) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t nch = __builtin_amdgcn_readfirstlane(wmisc[0]), last = __builtin_amdgcn_readfirstlane(wmisc[1]);
      if (lane == 0) a.nchunk[slot] = make_uint2(nch, last);
      if (lane < 8u) a.cls[(size_t)slot * 8u + lane] = wmisc[4u + lane];
      if (!modeB && n_match > item_cap && n_match - item_cap > (nch << ARENA_SHIFT)) n_fail++;
    }
  }
  if (lane == 0) {
    if (n_ovf) atomicAdd(&a.stats[1], n_ovf);
    if (n_fail) atomicAdd(&a.stats[6], (unsigned long long)n_fail);
    if (n_hpass) atomicAdd(a.hpass, n_hpass);
    if (bailed) atomicOr(a.err, 1u);
    if (n_two) {
      atomicAdd(&a.stats[4], (unsigned long long)n_two);
      atomicAdd(&a.stats[7], (unsigned long long)n_two);
    }
    if constexpr (CNT) {
      atomicAdd(&a.stats[8], (unsigned long long)c_tab);
      atomicAdd(&a.stats[9], (unsigned long long)c_c16);
      atomicAdd(&a.stats[10], (unsigned long long)c_ctx);
      atomicAdd(&a.stats[11], (unsigned long long)c_isa);
      atomicAdd(&a.stats[3], (unsigned long long)c_rec);
    }
  }
}

#define GS_DEF_SEED(NAME, CNT, SIDE)                                                                                \
  __global__ __launch_bounds__(WAVE *SEARCH_WAVES) __attribute__((amdgpu_waves_per_eu(GS_WAVES_EU_SEED, GS_WAVES_EU_SEED))) void NAME( \
      gs_search_args a) {                                                                                           \
    __shared__ uint4 s_lds[SEARCH_WAVES][SEED_LDS];                                                                 \
    k_seed_body<CNT, SIDE>(a, s_lds[threadIdx.x / WAVE]);                                                           \
  }
GS_DEF_SEED(k_seed_b, false, 0)
GS_DEF_SEED(k_seed_a, false, 1)
GS_DEF_SEED(k_seed_count_b, true, 0)
GS_DEF_SEED(k_seed_count_a, true, 1)

/* ---- host: the descriptor pre-pass + the schedules, then the two launches (run_search's SPEC form) ---- */
gs_status gs_seed_describe(gs_index *ix, const gs_search_args &sa, uint32_t ng, bool sorted, hipStream_t st, gs_search_args *out) {
  gs_status rc;
  if ((rc = gs_reserve(ix->w_desc, sizeof(gs_guide_desc) * ((size_t)ng + 1) * (sorted ? 3u : 1u))) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_sched, 4 * (2 * 65536 + 512 + 16))) != GS_OK) return rc;
  uint32_t *base = (uint32_t *)ix->w_sched.p;
  uint32_t *xwork = base, *hist = base + 512;
  gs_guide_desc *desc = (gs_guide_desc *)ix->w_desc.p, *desc_a = desc + ng, *desc_b = desc_a + ng;
  gs_describe_args da;
  memset(&da, 0, sizeof(da));
  da.guides = sa.guides;
  da.desc = desc;
  da.n = ng;
  da.L = sa.L;
  da.P = sa.P;
  da.k = sa.pt_k;
  da.x_len = sa.x_len;
  da.n_pt = sa.n_pt;
  da.code[0] = sa.pt[0][0].code;
  da.code[1] = sa.n_pt > 1 ? sa.pt[1][0].code : 0xFFFFFFFFu;
  da.hist = sorted ? hist : nullptr;
  da.m = sa.m;
  for (uint32_t s = 0; s < 2; s++) {
    da.cand[s] = sa.cand[s];
    da.cand_off[s] = sa.cand_off[s];
    da.cand_ids[s] = sa.cand_ids[s];
    da.n_cand[s] = sa.n_cand[s];
  }
  if (sorted) GS_HIP(hipMemsetAsync(hist, 0, 4 * 2 * 65536, st));
  hipLaunchKernelGGL(k_describe, dim3((ng + 255) / 256), dim3(256), 0, st, da);
  if (sorted) {
    hipLaunchKernelGGL(k_sched_scan, dim3(2), dim3(1024), 0, st, hist);
    hipLaunchKernelGGL(k_sched_scatter, dim3((ng + 255) / 256), dim3(256), 0, st, (const gs_guide_desc *)desc, ng, hist, desc_a, desc_b);
  }
  *out = sa;
  out->desc_a = sorted ? desc_a : desc;
  out->desc_b = sorted ? desc_b : desc;
  out->xwork = xwork;
  return GS_OK;
}
gs_status gs_seed_launch(const gs_search_args &sa, uint32_t grid, bool count_req, hipStream_t st) {
  GS_HIP(hipMemsetAsync(sa.xwork, 0, 4 * 512, st));
  if (count_req) {
    hipLaunchKernelGGL(k_seed_count_b, dim3(grid), dim3(WAVE * SEARCH_WAVES), 0, st, sa);
    hipLaunchKernelGGL(k_seed_count_a, dim3(grid), dim3(WAVE * SEARCH_WAVES), 0, st, sa);
  } else {
    hipLaunchKernelGGL(k_seed_b, dim3(grid), dim3(WAVE * SEARCH_WAVES), 0, st, sa);
    hipLaunchKernelGGL(k_seed_a, dim3(grid), dim3(WAVE * SEARCH_WAVES), 0, st, sa);
  }
  return GS_OK;
}
