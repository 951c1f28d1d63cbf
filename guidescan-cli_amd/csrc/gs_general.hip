/*
 * gs_general.hip -- the general search path: everything genome_index::inexact_search accepts that the
 * fast path (gs_search.hip) does not encode.
 *
 *   - guide symbols outside A,C,G,T: matched literally against the genome, otherwise charged a
 *     mismatch like any other (index.hpp:218-247); PAM symbols other than 'N': literal only
 *     (index.hpp:125-170 with zero mismatches); any number of alt PAMs (process.hpp:51-56);
 *   - RNA / DNA bulges (index.hpp:250-375, max_bulge_size = 1 as process.hpp:82-83 calls it).
 *
 * One wavefront per (guide, strand) walks the reference's recursion from the root with an LDS stack
 * of 48-byte nodes {sp, ep, state, match.sequence so far as raw bytes}.  Symbols are bytes: Occ of
 * A,C,G,T comes from the 64-byte blocks, Occ of anything else from the per-symbol run lists of the
 * BWT (gs_strand_dev::xr_*).  Matches are 48-byte records ordered by one device-wide comparator sort
 * on (guide, distance, index, sequence bytes, row) - std::string order is byte order -, made unique
 * per (guide, distance, index, sequence) like the per-distance std::set, then located.
 * Untuned on purpose (one node per lane per step, generic sort): it serves the inputs the fast path
 * refuses, per guide, so that no input of the reference aborts a batch.
 */
#include "gs_device.h"

#include <rocprim/rocprim.hpp>

#define GSTACK 1024 /* 48-byte nodes per wave: 48 KB, three single-wave workgroups per CU.  (512 nodes, six per CU: the room a pop of
                        64 nodes needs - 11 children each - is never there, pops shrink to 17 lanes and the batch takes 3 x as long) */
#define GFAN 12     /* children one node can push: 4 DNA-bulge + exact + 4 substitutions + RNA bulge + PAM hop (+1) */

/* state word: t[5:0] mm[8:6] dna[11:9] rna[14:12] bulge_type[16:15] curr[17] slen[23:18] pamid[28:24] inpam[29] hop[30] */
#define GM_T(m) ((m)&63u)
#define GM_MM(m) (((m) >> 6) & 7u)
#define GM_DNA(m) (((m) >> 9) & 7u)
#define GM_RNA(m) (((m) >> 12) & 7u)
#define GM_STATE(m) (((m) >> 15) & 3u)
#define GM_CURR(m) (((m) >> 17) & 1u)
#define GM_SLEN(m) (((m) >> 18) & 63u)
#define GM_PAMID(m) (((m) >> 24) & 31u)
#define GM_INPAM(m) (((m) >> 29) & 1u)
#define GM_HOP(m) (((m) >> 30) & 1u)
__device__ __forceinline__ uint32_t gm_make(uint32_t t, uint32_t mm, uint32_t dna, uint32_t rna, uint32_t state,
                                            uint32_t curr, uint32_t slen, uint32_t pamid, uint32_t inpam,
                                            uint32_t hop) {
  return t | (mm << 6) | (dna << 9) | (rna << 12) | (state << 15) | (curr << 17) | (slen << 18) | (pamid << 24) |
         (inpam << 29) | (hop << 30);
}

struct gs_gen_guide { /* one guide of the general path, prepared on the host */
  uint8_t q[32];      /* query bytes in consumption order (process.hpp:63, index.hpp:218) */
  uint8_t pam[8];     /* the guide's own PAM in consumption order */
};
struct gs_grec { /* one match, 48 bytes */
  uint32_t seq[8]; /* match.sequence, bytes packed big-endian: word order == std::string order */
  uint32_t sp, ep;
  uint32_t meta; /* mm[2:0] dna[5:3] rna[8:6] index[9] slen[15:10] */
  uint32_t g;
};
struct gs_grec_less {
  __host__ __device__ bool operator()(const gs_grec &a, const gs_grec &b) const {
    if (a.g != b.g) return a.g < b.g;
    const uint32_t ma = a.meta & 7u, mb = b.meta & 7u; /* off_targets_bwt[m.mismatches] */
    if (ma != mb) return ma < mb;
    const uint32_t ia = (a.meta >> 9) & 1u, ib = (b.meta >> 9) & 1u; /* forward index first */
    if (ia != ib) return ia < ib;
    for (int i = 0; i < 8; i++)
      if (a.seq[i] != b.seq[i]) return a.seq[i] < b.seq[i];
    return a.sp < b.sp;
  }
};

struct gs_gsearch_args {
  gs_strand_dev sd[2];
  const gs_gen_guide *guides;
  gs_grec *recs;             /* item s writes at recs[slot_off[s] ...]; nullptr = count only */
  const uint64_t *slot_off;
  /* slot_off == nullptr with recs: ONE pass - records go to recs[] in emission order through the counter pool_next
   * (the device-wide sort that follows orders by guide first, so an item's records need not be neighbours);
   * records beyond pool_cap are counted, not written: the host then runs the pass again with room for all */
  unsigned long long *pool_next;
  unsigned long long pool_cap;
  uint32_t *counts;
  uint32_t *work;   /* [0] work-queue head, [1] error flag (iteration bound hit) */
  uint8_t alt[32][8]; /* alt PAM patterns in consumption order */
  uint8_t plen[40];   /* symbols of pattern j (alt PAMs, then the guides' own at n_alt): the reference searches
                         alt PAMs of any length next to the guides' PAM (process.hpp:51-56) */
  uint32_t p_max;     /* the longest of them */
  uint32_t n_items, L, P, m, n_alt, max_rna, max_dna;
  uint32_t max_iter; /* per-item iteration bound */
};

__device__ __forceinline__ void gseq_append(uint32_t (&s)[8], uint32_t slen, uint32_t byte) {
  if (slen < 32u) s[slen >> 2] |= byte << (8u * (3u - (slen & 3u)));
}
__device__ __forceinline__ uint32_t glower(uint32_t b) { return b | 0x20u; } /* A,C,G,T -> a,c,g,t */

__global__ __launch_bounds__(WAVE) void k_search_general(gs_gsearch_args a) {
  __shared__ uint4 s_stack[GSTACK * 3];
  const uint32_t lane = lane_id();
  uint4 *stk = s_stack;
  const uint32_t L = a.L, P = a.P, m = a.m;
  const uint32_t npams = P ? a.n_alt + 1u : 1u;
  const uint32_t reserve = (GFAN - 1) * (L + a.p_max + a.max_dna + npams + 4u);
  const uint32_t limit = GSTACK > reserve ? GSTACK - reserve : 1u;
  const uint32_t BASES[4] = {'A', 'C', 'G', 'T'};
  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(a.work, 1u);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= a.n_items) break;
    const uint32_t n_guides = a.n_items >> 1;
    const uint32_t strand = item >= n_guides ? 1u : 0u;
    const uint32_t guide = item - strand * n_guides;
    const uint32_t slot = 2u * guide + strand;
    const gs_gen_guide *gg = a.guides + guide;
    const gs_strand_dev &sd = a.sd[strand];
    const uint4 *__restrict__ blocks = sd.blocks;
    const bool pooled = a.recs != nullptr && a.slot_off == nullptr;
    gs_grec *out = a.recs && !pooled ? a.recs + a.slot_off[slot] : a.recs;
    const uint32_t item_cap = a.recs && !pooled ? (uint32_t)(a.slot_off[slot + 1] - a.slot_off[slot]) : 0u;
    uint32_t n_match = 0, size = 1;
    if (lane == 0) {
      stk[0] = make_uint4(0u, sd.n - 1u, 0u, 0u);
      stk[1] = make_uint4(0u, 0u, 0u, 0u);
      stk[2] = make_uint4(0u, 0u, 0u, 0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    auto route = [&](bool live, bool term, uint32_t csp, uint32_t cep, const uint32_t (&sq)[8],
                     uint32_t cmeta) __attribute__((always_inline)) {
      const bool pu = live && !term, em = live && term;
      const uint64_t bp = __ballot(pu);
      if (bp) {
        if (pu) {
          const uint32_t at = 3u * (size + lanes_below(bp));
          stk[at] = make_uint4(csp, cep, cmeta, 0u);
          stk[at + 1u] = make_uint4(sq[0], sq[1], sq[2], sq[3]);
          stk[at + 2u] = make_uint4(sq[4], sq[5], sq[6], sq[7]);
        }
        size += __popcll(bp);
      }
      const uint64_t be = __ballot(em);
      if (be) {
        unsigned long long pbase = 0;
        if (pooled) { /* one atomic per emission of the wave */
          if (lane == (uint32_t)__builtin_ctzll(be)) pbase = atomicAdd(a.pool_next, (unsigned long long)__popcll(be));
          pbase = ((unsigned long long)__shfl((int)(pbase >> 32), (int)__builtin_ctzll(be)) << 32) |
                  (uint32_t)__shfl((int)(uint32_t)pbase, (int)__builtin_ctzll(be));
        }
        if (em) {
          const unsigned long long idx = pooled ? pbase + lanes_below(be) : (unsigned long long)(n_match + lanes_below(be));
          if (idx < (pooled ? a.pool_cap : (unsigned long long)item_cap)) {
            gs_grec r;
            for (int i = 0; i < 8; i++) r.seq[i] = sq[i];
            r.sp = csp;
            r.ep = cep;
            r.meta = GM_MM(cmeta) | (GM_DNA(cmeta) << 3) | (GM_RNA(cmeta) << 6) | (strand << 9) |
                     (GM_SLEN(cmeta) << 10);
            r.g = guide;
            out[idx] = r;
          }
        }
        n_match += __popcll(be);
      }
    };

    /* every wave must drain: past the iteration bound the item gives up loudly (error flag)
     * instead of spinning.  The exit and the tail below are free of lane-conditional blocks
     * (compiler pitfall). */
    uint32_t guard = 0;
    bool bail = false;
    while (size > 0 && !bail) {
      bail = ++guard > a.max_iter;
      uint32_t w = size < WAVE ? size : WAVE;
      const uint32_t room = size < limit ? limit - size : 0u;
      const uint32_t fit = room / (GFAN - 1);
      if (w > fit) w = fit ? fit : 1u;
      const bool active = lane < w;
      uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (active) {
        n0 = stk[3u * (size - 1u - lane)];
        n1 = stk[3u * (size - 1u - lane) + 1u];
        n2 = stk[3u * (size - 1u - lane) + 2u];
      }
      size -= w;
      const uint32_t sp = n0.x, ep = n0.y, meta = n0.z;
      const uint32_t sq[8] = {n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w};
      const uint32_t t = GM_T(meta), mm = GM_MM(meta), dna = GM_DNA(meta), rna = GM_RNA(meta);
      const uint32_t state = GM_STATE(meta), curr = GM_CURR(meta), slen = GM_SLEN(meta);
      const uint32_t pamid = GM_PAMID(meta);
      const bool inpam = GM_INPAM(meta) != 0u, hop = GM_HOP(meta) != 0u;
      uint32_t oa[4] = {0, 0, 0, 0}, ob[4] = {0, 0, 0, 0};
      if (active && !hop) {
        occ4(blocks, sp >> GS_BLOCK_SHIFT, sp & (GS_BLOCK_ROWS - 1u), oa[0], oa[1], oa[2], oa[3]);
        occ4(blocks, ep >> GS_BLOCK_SHIFT, (ep & (GS_BLOCK_ROWS - 1u)) + 1u, ob[0], ob[1], ob[2], ob[3]);
      }
      /* ---- the hop into the PAM stage (position < 0, index.hpp:193-216 / 297-314): one search per
       * pattern, alt PAMs first, the guide's own last (process.hpp:51-56); a chain of hop nodes
       * keeps the fan-out at two whatever the number of patterns */
      {
        const bool h = active && hop;
        route(h && P != 0u, false, sp, ep, sq, gm_make(L, mm, dna, rna, state, curr, slen, pamid, 1u, 0u));
        route(h && P != 0u && pamid + 1u < npams, false, sp, ep, sq,
              gm_make(L, mm, dna, rna, state, curr, slen, pamid + 1u, 0u, 1u));
        route(h && P == 0u, true, sp, ep, sq, meta); /* empty PAM: the finished guide is a match */
      }
      const bool guide_node = active && !hop && !inpam;
      /* ---- DNA bulge (index.hpp:265-295): opens before the terminal check, never at the first step */
      uint32_t d_dna = dna, d_state = state, d_curr = curr;
      if (a.max_dna > dna && (state != 1u || curr == 1u)) {
        d_state = 1u;
        d_curr = 0u;
        d_dna = dna + 1u;
      }
      const bool dna_ok = guide_node && d_state == 1u && d_curr < 1u && t != 0u;
      /* ---- RNA bulge (index.hpp:358-374): only with guide symbols left */
      uint32_t r_rna = rna, r_state = state, r_curr = curr;
      if (a.max_rna > rna && (state != 2u || curr == 1u)) {
        r_state = 2u;
        r_curr = 0u;
        r_rna = rna + 1u;
      }
      const bool rna_ok = guide_node && t < L && r_state == 2u && r_curr < 1u && t != 0u;
      const bool guide_step = guide_node && t < L;
      /* the query symbol of this step: a guide byte, or a byte of the PAM pattern */
      uint32_t qc = 0;
      if (guide_step) qc = gg->q[t];
      if (active && inpam) qc = pamid < a.n_alt ? a.alt[pamid][t - L] : gg->pam[t - L];
      const bool wild = active && inpam && qc == 'N'; /* PAM 'N': literal N, then A,T,C,G at cost 0 (index.hpp:139-169) */
      const uint32_t T_end = L + a.plen[pamid < 40u ? pamid : 39u]; /* where this pattern's PAM stage ends */
#pragma unroll
      for (uint32_t c = 0; c < 4u; ++c) {
        const uint32_t csp = sd.C[c] + oa[c], cep = sd.C[c] + ob[c] - 1u;
        const bool present = ob[c] > oa[c];
        /* DNA bulge child: genome base c consumed, guide position unchanged, lower case */
        {
          uint32_t s2[8] = {sq[0], sq[1], sq[2], sq[3], sq[4], sq[5], sq[6], sq[7]};
          gseq_append(s2, slen, glower(BASES[c]));
          route(dna_ok && present, false, csp, cep, s2, gm_make(t, mm, d_dna, rna, 1u, 1u, slen + 1u, 0u, 0u, 0u));
        }
        /* consuming child with base c: guide step (exact when c is the query byte, else a
         * substitution if the budget allows) or PAM step (the pattern's own base, or any under 'N') */
        {
          const bool exact = qc == BASES[c];
          uint32_t s2[8] = {sq[0], sq[1], sq[2], sq[3], sq[4], sq[5], sq[6], sq[7]};
          /* one route() for both kinds of step: it ballots, so it must not sit in a divergent branch */
          const bool live = inpam ? (active && present && (exact || wild))
                                  : (guide_step && present && (exact || mm < m));
          gseq_append(s2, slen, (inpam || exact) ? BASES[c] : glower(BASES[c]));
          const uint32_t cm = inpam ? gm_make(t + 1u, mm, dna, rna, state, curr, slen + 1u, pamid, 1u, 0u)
                                    : gm_make(t + 1u, mm + (exact ? 0u : 1u), dna, rna, 0u, curr, slen + 1u, 0u, 0u, 0u);
          route(live, inpam && t + 1u == T_end, csp, cep, s2, cm);
        }
      }
      /* the query byte itself when it is not a base: a literal match against the genome
       * (guide: index.hpp:218-228; PAM: :139-149 for 'N', :130-137 for any other symbol) */
      {
        const bool lit = (guide_step || (active && inpam)) && qc != 'A' && qc != 'C' && qc != 'G' && qc != 'T';
        if (__ballot(lit)) {
          bool live = false;
          uint32_t csp = 0, cep = 0;
          if (lit) {
            const uint32_t na = occ_sym(sd, qc, sp), nb = occ_sym(sd, qc, ep + 1u);
            live = nb > na;
            csp = sd.C256[qc & 255u] + na;
            cep = sd.C256[qc & 255u] + nb - 1u;
          }
          uint32_t s2[8] = {sq[0], sq[1], sq[2], sq[3], sq[4], sq[5], sq[6], sq[7]};
          gseq_append(s2, slen, qc);
          const uint32_t cm = inpam ? gm_make(t + 1u, mm, dna, rna, state, curr, slen + 1u, pamid, 1u, 0u)
                                    : gm_make(t + 1u, mm, dna, rna, 0u, curr, slen + 1u, 0u, 0u, 0u);
          route(live, inpam && t + 1u == T_end, csp, cep, s2, cm);
        }
      }
      /* the guide is consumed: hand over to the PAM stage */
      route(guide_node && t == L, false, sp, ep, sq, gm_make(L, mm, dna, rna, state, curr, slen, 0u, 0u, 1u));
      /* RNA bulge child: a guide symbol skipped, interval unchanged, '.' recorded */
      {
        uint32_t s2[8] = {sq[0], sq[1], sq[2], sq[3], sq[4], sq[5], sq[6], sq[7]};
        gseq_append(s2, slen, '.');
        route(rna_ok, false, sp, ep, s2, gm_make(t + 1u, mm, dna, r_rna, 2u, 1u, slen + 1u, 0u, 0u, 0u));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    a.counts[slot] = n_match; /* same address, same value from every lane */
    if (bail) atomicOr(&a.work[1], 1u);
  }
}

/* flag[r] = 1 when sorted record r starts a new (guide, distance, index, sequence) */
__global__ void k_gen_flags(const gs_grec *srt, uint64_t T, uint32_t *flag) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T) return;
  bool f = true;
  if (r > 0) {
    const gs_grec &p = srt[r - 1], &c = srt[r];
    /* std::set<match> is keyed on the sequence alone (structures.hpp:40-42), per distance */
    f = !(p.g == c.g && (p.meta & 7u) == (c.meta & 7u) && ((p.meta >> 9) & 1u) == ((c.meta >> 9) & 1u));
    if (!f)
      for (int i = 0; i < 8; i++) f = f || p.seq[i] != c.seq[i];
  }
  flag[r] = f ? 1u : 0u;
}
__global__ void k_gen_compact(const gs_grec *srt, const uint32_t *flag, const uint32_t *pos, uint64_t T,
                              gs_grec *uq, unsigned long long *cnt64, uint32_t *nmatch,
                              unsigned long long *nhits64) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T || !flag[r]) return;
  const gs_grec h = srt[r];
  const uint32_t c = h.ep - h.sp + 1u;
  uq[pos[r]] = h;
  cnt64[pos[r]] = c;
  atomicAdd(&nmatch[h.g], 1u);
  atomicAdd(&nhits64[h.g], (unsigned long long)c);
}
struct gs_glocate_args {
  gs_strand_dev sd[2];
  const gs_grec *uq;
  const unsigned long long *hit_scan;
  const unsigned long long *guide_first;
  const uint64_t *offsets;
  gs_hit_ex *hits;
  uint64_t genome_length;
  uint32_t n_uq;
};
__global__ __launch_bounds__(WAVE) void k_gen_locate(gs_glocate_args a) {
  const uint32_t r = blockIdx.x;
  if (r >= a.n_uq) return;
  const gs_grec m = a.uq[r];
  const uint32_t strand = (m.meta >> 9) & 1u;
  gs_hit_ex *out = a.hits + a.offsets[m.g] + (a.hit_scan[r] - a.guide_first[m.g]);
  const uint32_t cnt = m.ep - m.sp + 1u;
  for (uint32_t h = lane_id(); h < cnt; h += WAVE) {
    const uint64_t sa = a.sd[strand].sa[m.sp + h];
    gs_hit_ex o;
    o.pos = strand == 0 ? -(int64_t)sa : (int64_t)(a.genome_length - (sa + 1ull));
    for (int i = 0; i < 32; i++) o.seq[i] = (char)((m.seq[i >> 2] >> (8 * (3 - (i & 3)))) & 255u);
    o.mismatches = m.meta & 7u;
    o.dna_bulges = (uint8_t)((m.meta >> 3) & 7u);
    o.rna_bulges = (uint8_t)((m.meta >> 6) & 7u);
    o.index = (uint8_t)strand;
    o.seq_len = (uint8_t)((m.meta >> 10) & 63u);
    out[h] = o;
  }
}
__global__ void k_gen_first(const unsigned long long *hit_scan, const uint64_t *first_rec, uint32_t n,
                            unsigned long long *guide_first) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) guide_first[g] = hit_scan[first_rec[g]];
}

/* hits per guide before the sets drop duplicate sequences (off_target_counter, process.hpp:25-27) */
__global__ void k_gen_raw(const gs_grec *recs, uint64_t T, unsigned long long *raw) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < T) atomicAdd(&raw[recs[r].g], (unsigned long long)(recs[r].ep - recs[r].sp + 1u));
}

struct gs_result_ex {
  std::vector<uint64_t> offsets;
  std::vector<gs_hit_ex> hits;
  std::vector<uint32_t> raw;
};

#define GS_TRY(expr)                 \
  do {                               \
    gs_status rc__ = (expr);         \
    if (rc__ != GS_OK) return rc__;  \
  } while (0)

/* genomics::complement (src/genomics/sequences.cxx:14-26): bases and lower-case bases, anything else unchanged */
static uint8_t gen_comp(uint8_t c) {
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

static gs_status enumerate_general(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                                   const char *guide_pams, uint32_t P, const char *alt_pams, uint32_t n_alt,
                                   uint32_t mismatches, uint32_t rna_bulges, uint32_t dna_bulges, uint32_t flags,
                                   gs_result_ex **out, const uint32_t *alt_lens = nullptr) {
  if (!ix || !out || (n && !guides) || (n && P && !guide_pams) || (n_alt && !alt_pams)) return GS_ERR_ARG;
  uint32_t p_max = P;
  for (uint32_t j = 0; alt_lens && j < n_alt && j < 32; j++) {
    if (alt_lens[j] < 1 || alt_lens[j] > 8) {
      gs_set_error("alt PAMs of 1 to 8 symbols");
      return GS_ERR_UNSUPPORTED;
    }
    p_max = std::max(p_max, alt_lens[j]);
  }
  if (L < 1 || L > 31 || P > 8 || mismatches > 7 || n_alt > 31 || rna_bulges > 3 || dna_bulges > 3 ||
      L + dna_bulges + p_max > 32 || n >= (1ull << 30)) {
    gs_set_error("general path supports L<=31, P<=8, mismatches<=7, <=31 alt PAMs, <=3 bulges of each kind, "
                 "L+dna_bulges+P<=32");
    return GS_ERR_UNSUPPORTED;
  }
  if (!ix->strand[0].xr_seg || !ix->strand[1].xr_seg) {
    gs_set_error("index lacks the per-symbol run lists");
    return GS_ERR_UNSUPPORTED;
  }
  GS_HIP(hipSetDevice(ix->device));
  hipStream_t st = nullptr;
  const uint32_t n32 = (uint32_t)n;
  const bool start = (flags & GS_FLAG_PAM_AT_START) != 0;
  gs_result_ex *res = new (std::nothrow) gs_result_ex();
  if (!res) return GS_ERR_NOMEM;
  struct res_guard {
    gs_result_ex *p;
    ~res_guard() { delete p; }
  } guard{res};
  res->offsets.assign(n + 1, 0);
  res->raw.assign(n, 0);
  if (n == 0) {
    guard.p = nullptr;
    *out = res;
    return GS_OK;
  }
  /* per-call device buffers (rare mode: no workspace reuse) */
  struct dbuf {
    void *p = nullptr;
    ~dbuf() {
      if (p) hipFree(p);
    }
    gs_status get(size_t b) { return hipMalloc(&p, b ? b : 16) == hipSuccess ? GS_OK : GS_ERR_NOMEM; }
  };
  dbuf d_g, d_cnt, d_misc, d_off, d_a, d_b, d_flag, d_pos, d_uq, d_c64, d_scan, d_nm, d_nh, d_first, d_frec, d_goff,
      d_hits, d_tmp;
  /* query = reverse_complement(sequence) consumed right to left == complement of the guide left to
   * right (process.hpp:63, index.hpp:218); with --start the guide itself right to left; PAMs likewise */
  std::vector<gs_gen_guide> hg(n);
  for (uint64_t g = 0; g < n; g++) {
    memset(&hg[g], 0, sizeof(gs_gen_guide));
    const uint8_t *s = (const uint8_t *)guides + g * L;
    for (uint32_t t = 0; t < L; t++) hg[g].q[t] = start ? s[L - 1 - t] : gen_comp(s[t]);
    const uint8_t *p = (const uint8_t *)guide_pams + g * P;
    for (uint32_t u = 0; u < P; u++) hg[g].pam[u] = start ? p[P - 1 - u] : gen_comp(p[u]);
  }
  gs_gsearch_args sa;
  memset(&sa, 0, sizeof(sa));
  {
    size_t at = 0; /* alt PAMs back to back: P symbols each, or alt_lens[j] */
    for (uint32_t j = 0; j < (P ? n_alt : 0u); j++) {
      const uint32_t pl = alt_lens ? alt_lens[j] : P;
      const uint8_t *p = (const uint8_t *)alt_pams + at;
      for (uint32_t u = 0; u < pl; u++) sa.alt[j][u] = start ? p[pl - 1 - u] : gen_comp(p[u]);
      sa.plen[j] = (uint8_t)pl;
      at += pl;
    }
    sa.plen[P ? n_alt : 0u] = (uint8_t)P;
    sa.p_max = p_max;
  }
  GS_TRY(d_g.get(sizeof(gs_gen_guide) * n));
  GS_TRY(d_cnt.get(8 * n));
  GS_TRY(d_misc.get(64));
  GS_HIP(hipMemcpy(d_g.p, hg.data(), sizeof(gs_gen_guide) * n, hipMemcpyHostToDevice));
  GS_HIP(hipMemset(d_misc.p, 0, 64));
  sa.sd[0] = ix->strand[0].d;
  sa.sd[1] = ix->strand[1].d;
  sa.guides = (const gs_gen_guide *)d_g.p;
  sa.recs = nullptr;
  sa.slot_off = nullptr;
  sa.counts = (uint32_t *)d_cnt.p;
  sa.work = (uint32_t *)d_misc.p;
  sa.n_items = 2 * n32;
  sa.L = L;
  sa.P = P;
  sa.m = mismatches;
  sa.n_alt = P ? n_alt : 0; /* empty guide PAM drops the alt PAMs: process.hpp:52-53 */
  sa.max_rna = rna_bulges;
  sa.max_dna = dna_bulges;
  sa.max_iter = gs_opt(ix, "GS_BULGE_MAX_ITER") ? (uint32_t)atol(gs_opt(ix, "GS_BULGE_MAX_ITER")) : (1u << 26);
  const uint32_t grid_max = (uint32_t)gs_num_cus(ix->device) * 3u; /* 48 KB of LDS per single-wave workgroup */
  uint32_t grid = 2 * n32;
  if (grid > grid_max) grid = grid_max;
  /* ONE search pass: records go to a pool through an atomic counter (the sort below orders by guide first).  The pool
   * is sized by a guess - 256 records per guide - and what does not fit is only counted: the pass then runs again with
   * room for all (the first version always searched twice: a counting pass, then a filling pass at exact offsets) */
  uint64_t T = 0, cap = std::max<uint64_t>((uint64_t)n * 256u, 1u << 16);
  if (const char *e = gs_opt(ix, "GS_GENERAL_POOL")) cap = (uint64_t)std::max(1ll, atoll(e));
  {
    /* the guess may take a quarter of what is free next to a resident index (12 KB per guide is 12 GB for 10^6 guides);
     * a smaller pool only means that a batch with many hits is searched a second time with room for exactly its records */
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const uint64_t fit = (uint64_t)(free_b / 4) / sizeof(gs_grec);
      if (cap > fit) cap = std::max<uint64_t>(fit, 1u << 12);
    } else {
      (void)hipGetLastError();
    }
  }
  for (int attempt = 0; attempt < 2; attempt++) {
    if (d_a.p) {
      hipFree(d_a.p);
      d_a.p = nullptr;
    }
    if (attempt == 0 && d_a.get(sizeof(gs_grec) * cap) != GS_OK) {
      (void)hipGetLastError(); /* no room for the guess: count with a small pool, then allocate what the batch needs */
      cap = 1u << 12;
    }
    if (!d_a.p) GS_TRY(d_a.get(sizeof(gs_grec) * cap));
    GS_HIP(hipMemset(d_misc.p, 0, 64));
    sa.recs = (gs_grec *)d_a.p;
    sa.slot_off = nullptr;
    sa.pool_next = (unsigned long long *)((char *)d_misc.p + 16);
    sa.pool_cap = cap;
    hipLaunchKernelGGL(k_search_general, dim3(grid), dim3(WAVE), 0, st, sa);
    uint32_t h_misc[8] = {0};
    GS_HIP(hipMemcpy(h_misc, d_misc.p, 32, hipMemcpyDeviceToHost));
    if (h_misc[1]) {
      gs_set_error("internal: general search exceeded its iteration bound");
      return GS_ERR_DEVICE;
    }
    T = ((uint64_t)h_misc[5] << 32) | h_misc[4];
    if (T <= cap) break;
    if (attempt == 1) {
      gs_set_error("internal: general search found more records the second time");
      return GS_ERR_DEVICE;
    }
    cap = T;
  }
  if (T >= (1ull << 31)) {
    gs_set_error("more than 2^31 match records in one batch of the general path: use smaller batches");
    return GS_ERR_UNSUPPORTED;
  }
  if (T == 0) {
    guard.p = nullptr;
    *out = res;
    return GS_OK;
  }
  GS_TRY(d_b.get(sizeof(gs_grec) * T));
  {
    dbuf d_raw;
    GS_TRY(d_raw.get(8 * n));
    GS_HIP(hipMemsetAsync(d_raw.p, 0, 8 * n, st));
    hipLaunchKernelGGL(k_gen_raw, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, (const gs_grec *)d_a.p, T,
                       (unsigned long long *)d_raw.p);
    std::vector<unsigned long long> hr(n);
    GS_HIP(hipMemcpy(hr.data(), d_raw.p, 8 * n, hipMemcpyDeviceToHost));
    for (uint64_t g = 0; g < n; g++) res->raw[g] = hr[g] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)hr[g];
  }
  /* canonical order: (guide, distance, index, sequence, row) */
  size_t tb = 0, tb2 = 0, tb3 = 0;
  GS_TRY(d_flag.get(4 * T));
  GS_TRY(d_pos.get(4 * T));
  GS_TRY(d_c64.get(8 * (T + 1)));
  GS_TRY(d_scan.get(8 * (T + 1)));
  GS_HIP(rocprim::merge_sort(nullptr, tb, (gs_grec *)d_a.p, (gs_grec *)d_b.p, (size_t)T, gs_grec_less(), st));
  GS_HIP(rocprim::exclusive_scan(nullptr, tb2, (uint32_t *)d_flag.p, (uint32_t *)d_pos.p, 0u, (size_t)T,
                                 rocprim::plus<uint32_t>(), st));
  GS_HIP(rocprim::exclusive_scan(nullptr, tb3, (unsigned long long *)d_c64.p, (unsigned long long *)d_scan.p,
                                 0ull, (size_t)T + 1, rocprim::plus<unsigned long long>(), st));
  if (tb2 > tb) tb = tb2;
  if (tb3 > tb) tb = tb3;
  GS_TRY(d_tmp.get(tb + 16));
  size_t tbs = tb;
  GS_HIP(rocprim::merge_sort(d_tmp.p, tbs, (gs_grec *)d_a.p, (gs_grec *)d_b.p, (size_t)T, gs_grec_less(), st));
  const unsigned gT = (unsigned)((T + 255) / 256);
  hipLaunchKernelGGL(k_gen_flags, dim3(gT), dim3(256), 0, st, (const gs_grec *)d_b.p, T, (uint32_t *)d_flag.p);
  tbs = tb;
  GS_HIP(rocprim::exclusive_scan(d_tmp.p, tbs, (uint32_t *)d_flag.p, (uint32_t *)d_pos.p, 0u, (size_t)T,
                                 rocprim::plus<uint32_t>(), st));
  GS_TRY(d_uq.get(sizeof(gs_grec) * T));
  GS_TRY(d_nm.get(4 * n));
  GS_TRY(d_nh.get(8 * n));
  GS_HIP(hipMemsetAsync(d_nm.p, 0, 4 * n, st));
  GS_HIP(hipMemsetAsync(d_nh.p, 0, 8 * n, st));
  GS_HIP(hipMemsetAsync(d_c64.p, 0, 8 * (T + 1), st));
  hipLaunchKernelGGL(k_gen_compact, dim3(gT), dim3(256), 0, st, (const gs_grec *)d_b.p,
                     (const uint32_t *)d_flag.p, (const uint32_t *)d_pos.p, T, (gs_grec *)d_uq.p,
                     (unsigned long long *)d_c64.p, (uint32_t *)d_nm.p, (unsigned long long *)d_nh.p);
  std::vector<uint32_t> nm(n);
  std::vector<unsigned long long> nh(n);
  GS_HIP(hipMemcpy(nm.data(), d_nm.p, 4 * n, hipMemcpyDeviceToHost));
  GS_HIP(hipMemcpy(nh.data(), d_nh.p, 8 * n, hipMemcpyDeviceToHost));
  std::vector<uint64_t> first_rec(n + 1, 0);
  uint64_t nuq = 0;
  for (size_t g = 0; g < n; g++) {
    first_rec[g] = nuq;
    nuq += nm[g];
    res->offsets[g + 1] = res->offsets[g] + nh[g];
  }
  first_rec[n] = nuq;
  const uint64_t H = res->offsets[n];
  tbs = tb;
  GS_HIP(rocprim::exclusive_scan(d_tmp.p, tbs, (unsigned long long *)d_c64.p, (unsigned long long *)d_scan.p,
                                 0ull, (size_t)nuq + 1, rocprim::plus<unsigned long long>(), st));
  GS_TRY(d_frec.get(8 * (n + 1)));
  GS_TRY(d_first.get(8 * n));
  GS_TRY(d_goff.get(8 * (n + 1)));
  GS_TRY(d_hits.get(sizeof(gs_hit_ex) * (H + 1)));
  GS_HIP(hipMemcpy(d_frec.p, first_rec.data(), 8 * (n + 1), hipMemcpyHostToDevice));
  GS_HIP(hipMemcpy(d_goff.p, res->offsets.data(), 8 * (n + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_gen_first, dim3((n32 + 255) / 256), dim3(256), 0, st,
                     (const unsigned long long *)d_scan.p, (const uint64_t *)d_frec.p, n32,
                     (unsigned long long *)d_first.p);
  gs_glocate_args la;
  la.sd[0] = ix->strand[0].d;
  la.sd[1] = ix->strand[1].d;
  la.uq = (const gs_grec *)d_uq.p;
  la.hit_scan = (const unsigned long long *)d_scan.p;
  la.guide_first = (const unsigned long long *)d_first.p;
  la.offsets = (const uint64_t *)d_goff.p;
  la.hits = (gs_hit_ex *)d_hits.p;
  la.genome_length = ix->genome_length;
  la.n_uq = (uint32_t)nuq;
  if (nuq) hipLaunchKernelGGL(k_gen_locate, dim3((unsigned)nuq), dim3(WAVE), 0, st, la);
  res->hits.resize(H);
  if (H) GS_HIP(hipMemcpy(res->hits.data(), d_hits.p, sizeof(gs_hit_ex) * H, hipMemcpyDeviceToHost));
  GS_HIP(hipDeviceSynchronize());
  GS_HIP(hipGetLastError());
  guard.p = nullptr;
  *out = res;
  return GS_OK;
}

extern "C" gs_status gs_enumerate_general(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                                          const char *guide_pams, uint32_t P, const char *alt_pams, uint32_t n_alt,
                                          uint32_t mismatches, uint32_t rna_bulges, uint32_t dna_bulges,
                                          uint32_t flags, gs_result_ex **out) {
  GS_HANDLE_LOCK(ix);
  try {
    return enumerate_general(ix, guides, n, L, guide_pams, P, alt_pams, n_alt, mismatches, rna_bulges, dna_bulges,
                             flags, out);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}
/* the same with alt PAMs of their own lengths (alt_pams: the patterns back to back, alt_lens[j] symbols each) */
extern "C" gs_status gs_enumerate_general_pams(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                                               const char *guide_pams, uint32_t P, const char *alt_pams,
                                               const uint32_t *alt_lens, uint32_t n_alt, uint32_t mismatches,
                                               uint32_t rna_bulges, uint32_t dna_bulges, uint32_t flags,
                                               gs_result_ex **out) {
  GS_HANDLE_LOCK(ix);
  if (n_alt && !alt_lens) return GS_ERR_ARG;
  try {
    return enumerate_general(ix, guides, n, L, guide_pams, P, alt_pams, n_alt, mismatches, rna_bulges, dna_bulges,
                             flags, out, alt_lens);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}
/* the name the bulge options were first served under */
extern "C" gs_status gs_enumerate_bulges(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                                         const char *guide_pams, uint32_t P, const char *alt_pams,
                                         uint32_t n_alt, uint32_t mismatches, uint32_t rna_bulges,
                                         uint32_t dna_bulges, uint32_t flags, gs_result_ex **out) {
  return gs_enumerate_general(ix, guides, n, L, guide_pams, P, alt_pams, n_alt, mismatches, rna_bulges, dna_bulges,
                              flags, out);
}
extern "C" gs_status gs_result_ex_get(const gs_result_ex *r, uint64_t *n_guides, const uint64_t **offsets,
                                      const gs_hit_ex **hits) {
  if (!r) return GS_ERR_ARG;
  if (n_guides) *n_guides = r->offsets.size() - 1;
  if (offsets) *offsets = r->offsets.data();
  if (hits) *hits = r->hits.data();
  return GS_OK;
}
extern "C" gs_status gs_result_ex_raw_hits(const gs_result_ex *r, const uint32_t **raw_hits) {
  if (!r || !raw_hits) return GS_ERR_ARG;
  *raw_hits = r->raw.data();
  return GS_OK;
}
extern "C" void gs_result_ex_free(gs_result_ex *r) { delete r; }
