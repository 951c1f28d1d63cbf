/*
 * gs_general.hip -- the general search path (bulge-aware recursion and guides/PAMs with symbols outside
 * A,C,G,T).  Split from gs_search.hip; see the section comment below.
 */
#include "gs_device.h"

#include <rocprim/rocprim.hpp>

/* =====================================================================================
 * Bulge-aware search (include/genomics/index.hpp:250-375).
 * A separate, general path: walk from the root with the affinity state of index.hpp:12-20
 * in every node, sequences as 4-bit codes in a 128-bit key (they contain '.', lower-case
 * bulge bases and vary in length), records ordered by one device-wide comparator sort.
 * Not tuned: no BASELINE config uses bulges; it exists so that the CLI covers the option.
 * ===================================================================================== */
#define BSTACK 512 /* 32-byte nodes per wave */
#ifndef BWAVES
#define BWAVES 2
#endif
#define BFAN 10

/* meta: t[5:0] mm[8:6] dna[11:9] rna[14:12] state[16:15] curr[17] slen[23:18] pamid[25:24] inpam[26] */
#define BM_T(m) ((m)&63u)
#define BM_MM(m) (((m) >> 6) & 7u)
#define BM_DNA(m) (((m) >> 9) & 7u)
#define BM_RNA(m) (((m) >> 12) & 7u)
#define BM_STATE(m) (((m) >> 15) & 3u)
#define BM_CURR(m) (((m) >> 17) & 1u)
#define BM_SLEN(m) (((m) >> 18) & 63u)
#define BM_PAMID(m) (((m) >> 24) & 3u)
#define BM_INPAM(m) (((m) >> 26) & 1u)
__device__ __forceinline__ uint32_t bm_make(uint32_t t, uint32_t mm, uint32_t dna, uint32_t rna,
                                            uint32_t state, uint32_t curr, uint32_t slen, uint32_t pamid,
                                            uint32_t inpam) {
  return t | (mm << 6) | (dna << 9) | (rna << 12) | (state << 15) | (curr << 17) | (slen << 18) |
         (pamid << 24) | (inpam << 26);
}
/* sequence symbols as 4-bit codes that sort like their ASCII bytes: 0 pad < '.' < A C G N T < a c g t */
__device__ __forceinline__ void seq_append(uint64_t &hi, uint64_t &lo, uint32_t slen, uint32_t code) {
  if (slen < 16u)
    hi |= (uint64_t)code << (60u - 4u * slen);
  else if (slen < 32u)
    lo |= (uint64_t)code << (60u - 4u * (slen - 16u));
}
__device__ __forceinline__ uint32_t code_upper(uint32_t b) { return b < 3u ? 2u + b : 6u; } /* A C G T */
__device__ __forceinline__ uint32_t code_lower(uint32_t b) { return 7u + b; }

struct gs_brec { /* one match of the bulge path, 32 bytes */
  uint64_t key_hi, key_lo;
  uint32_t sp, ep;
  uint32_t meta; /* mm[2:0] dna[5:3] rna[8:6] index[9] slen[15:10] */
  uint32_t g;
};
struct gs_brec_less {
  __host__ __device__ bool operator()(const gs_brec &a, const gs_brec &b) const {
    if (a.g != b.g) return a.g < b.g;
    const uint32_t ma = a.meta & 7u, mb = b.meta & 7u; /* distance */
    if (ma != mb) return ma < mb;
    const uint32_t ia = (a.meta >> 9) & 1u, ib = (b.meta >> 9) & 1u; /* forward index first */
    if (ia != ib) return ia < ib;
    if (a.key_hi != b.key_hi) return a.key_hi < b.key_hi;
    if (a.key_lo != b.key_lo) return a.key_lo < b.key_lo;
    return a.sp < b.sp;
  }
};

struct gs_bsearch_args {
  gs_strand_dev sd[2];
  const gs_guide_rec *guides;
  gs_brec *recs;             /* item s writes at recs[slot_off[s] ...]; nullptr = count only */
  const uint64_t *slot_off;
  uint32_t *counts;
  uint32_t *work;   /* [0] work-queue head, [1] error flag (iteration bound hit) */
  uint32_t n_items, L, P, m, max_rna, max_dna;
  uint32_t max_iter; /* per-item iteration bound */
};

__global__ __launch_bounds__(WAVE *BWAVES) void k_search_bulge(gs_bsearch_args a) {
  __shared__ uint4 s_bstack[BWAVES][BSTACK * 2];
  const uint32_t wave = threadIdx.x / WAVE;
  const uint32_t lane = lane_id();
  uint4 *stk = s_bstack[wave];
  const uint32_t L = a.L, P = a.P, m = a.m;
  const uint32_t T_end = L + P;
  const uint32_t reserve = (BFAN - 1) * (T_end + a.max_dna + 3u);
  const uint32_t limit = BSTACK - reserve;
  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(a.work, 1u);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= a.n_items) break;
    const uint32_t n_guides = a.n_items >> 1;
    const uint32_t strand = item >= n_guides ? 1u : 0u;
    const uint32_t guide = item - strand * n_guides;
    const uint32_t slot = 2u * guide + strand;
    const uint32_t *gp = (const uint32_t *)(a.guides + guide);
    /* readfirstlane returns int: go through uint32_t or bit 31 of the low word sign-extends */
    const uint32_t gw0 = __builtin_amdgcn_readfirstlane(gp[0]);
    const uint32_t gw1 = __builtin_amdgcn_readfirstlane(gp[1]);
    const uint64_t gr_q = ((uint64_t)gw1 << 32) | gw0;
    const uint32_t gr_pam0 = __builtin_amdgcn_readfirstlane(gp[2]);
    const uint32_t gr_pam1 = __builtin_amdgcn_readfirstlane(gp[3]);
    const uint32_t gr_pam2 = __builtin_amdgcn_readfirstlane(gp[4]);
    const uint32_t gr_pam3 = __builtin_amdgcn_readfirstlane(gp[5]);
    const uint32_t npams = P ? __builtin_amdgcn_readfirstlane(gp[6]) : 1u;
    const gs_strand_dev &sd = a.sd[strand];
    const uint4 *__restrict__ blocks = sd.blocks;
    gs_brec *out = a.recs ? a.recs + a.slot_off[slot] : nullptr;
    const uint32_t item_cap = a.recs ? (uint32_t)(a.slot_off[slot + 1] - a.slot_off[slot]) : 0u;
    uint32_t n_match = 0, size = 1;
    if (lane == 0) {
      stk[0] = make_uint4(0u, sd.n - 1u, 0u, 0u);
      stk[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    auto route = [&](bool live, bool term, uint32_t csp, uint32_t cep, uint64_t chi, uint64_t clo,
                     uint32_t cmeta) __attribute__((always_inline)) {
      const bool pu = live && !term, em = live && term;
      const uint64_t bp = __ballot(pu);
      if (bp) {
        if (pu) {
          const uint32_t at = 2u * (size + lanes_below(bp));
          stk[at] = make_uint4(csp, cep, cmeta, 0u);
          stk[at + 1u] = make_uint4((uint32_t)chi, (uint32_t)(chi >> 32), (uint32_t)clo, (uint32_t)(clo >> 32));
        }
        size += __popcll(bp);
      }
      const uint64_t be = __ballot(em);
      if (be) {
        if (em) {
          const uint32_t idx = n_match + lanes_below(be);
          if (idx < item_cap) {
            gs_brec r;
            r.key_hi = chi;
            r.key_lo = clo;
            r.sp = csp;
            r.ep = cep;
            r.meta = BM_MM(cmeta) | (BM_DNA(cmeta) << 3) | (BM_RNA(cmeta) << 6) | (strand << 9) |
                     (BM_SLEN(cmeta) << 10);
            r.g = guide;
            out[idx] = r;
          }
        }
        n_match += __popcll(be);
      }
    };

    /* every wave must drain: past the iteration bound the item gives up loudly (error flag)
     * instead of spinning.  The exit and the tail below are deliberately free of
     * lane-conditional blocks: with an `if (lane == 0)` at both ends of the item loop the
     * compiler threaded lane 0 and lanes 1..63 through the back edge separately, so that the
     * readfirstlane of the next item ran on a partial wave. */
    uint32_t guard = 0;
    bool bail = false;
    while (size > 0 && !bail) {
      bail = ++guard > a.max_iter;
      uint32_t w = size < WAVE ? size : WAVE;
      const uint32_t room = size < limit ? limit - size : 0u;
      const uint32_t fit = room / (BFAN - 1);
      if (w > fit) w = fit ? fit : 1u;
      const bool active = lane < w;
      uint4 n0 = make_uint4(0, 0, 0, 0), n1 = make_uint4(0, 0, 0, 0);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (active) {
        n0 = stk[2u * (size - 1u - lane)];
        n1 = stk[2u * (size - 1u - lane) + 1u];
      }
      size -= w;
      const uint32_t sp = n0.x, ep = n0.y, meta = n0.z;
      const uint64_t shi = ((uint64_t)n1.y << 32) | n1.x, slo = ((uint64_t)n1.w << 32) | n1.z;
      const uint32_t t = BM_T(meta), mm = BM_MM(meta), dna = BM_DNA(meta), rna = BM_RNA(meta);
      const uint32_t state = BM_STATE(meta), curr = BM_CURR(meta), slen = BM_SLEN(meta);
      const uint32_t pamid = BM_PAMID(meta);
      const bool inpam = BM_INPAM(meta) != 0u;
      uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
      if (active) {
        occ4(blocks, sp >> GS_BLOCK_SHIFT, sp & (GS_BLOCK_ROWS - 1u), a0, a1, a2, a3);
        occ4(blocks, ep >> GS_BLOCK_SHIFT, (ep & (GS_BLOCK_ROWS - 1u)) + 1u, b0, b1, b2, b3);
      }
      /* ---- DNA bulge (index.hpp:265-295): opens before the terminal check, never at the
       * first step (position == len-1) */
      uint32_t d_dna = dna, d_state = state, d_curr = curr;
      if (a.max_dna > dna && (state != 1u || curr == 1u)) {
        d_state = 1u;
        d_curr = 0u;
        d_dna = dna + 1u;
      }
      const bool dna_ok = active && !inpam && d_state == 1u && d_curr < 1u && t != 0u;
      /* ---- RNA bulge (index.hpp:358-374): only with guide symbols left */
      uint32_t r_rna = rna, r_state = state, r_curr = curr;
      if (a.max_rna > rna && (state != 2u || curr == 1u)) {
        r_state = 2u;
        r_curr = 0u;
        r_rna = rna + 1u;
      }
      const bool rna_ok = active && !inpam && t < L && r_state == 2u && r_curr < 1u && t != 0u;
      const bool terminal = active && !inpam && t == L; /* position < 0 (index.hpp:297-314) */
      const bool guide_step = active && !inpam && t < L;
      uint32_t qc = 0, allow = 0, pc = 0;
      if (guide_step) {
        qc = (uint32_t)(gr_q >> (2u * t)) & 3u;
        allow = mm < m ? 0xFu : (1u << qc);
      } else if (active && inpam) {
        const uint32_t pw =
            pamid == 0 ? gr_pam0 : pamid == 1 ? gr_pam1 : pamid == 2 ? gr_pam2 : gr_pam3;
        pc = (pw >> (3u * (t - L))) & 7u;
        allow = pc < 4u ? (1u << pc) : 0xFu;
      }
#pragma unroll
      for (uint32_t c = 0; c < 4u; ++c) {
        const uint32_t oa = c == 0 ? a0 : c == 1 ? a1 : c == 2 ? a2 : a3;
        const uint32_t ob = c == 0 ? b0 : c == 1 ? b1 : c == 2 ? b2 : b3;
        const uint32_t csp = sd.C[c] + oa, cep = sd.C[c] + ob - 1u;
        /* DNA bulge child: genome base c consumed, guide position unchanged, lower case */
        {
          uint64_t hi = shi, lo = slo;
          seq_append(hi, lo, slen, code_lower(c));
          route(dna_ok && ob > oa, false, csp, cep, hi, lo,
                bm_make(t, mm, d_dna, rna, 1u, 1u, slen + 1u, 0u, 0u));
        }
        /* consuming child: guide step (exact / substitution) or PAM step */
        {
          const bool live = (guide_step || (active && inpam)) && ((allow >> c) & 1u) && ob > oa;
          uint64_t hi = shi, lo = slo;
          uint32_t cm;
          if (!inpam) {
            const bool sub = c != qc;
            seq_append(hi, lo, slen, sub ? code_lower(c) : code_upper(c));
            cm = bm_make(t + 1u, mm + (sub ? 1u : 0u), dna, rna, 0u, curr, slen + 1u, 0u, 0u);
          } else {
            seq_append(hi, lo, slen, code_upper(c));
            cm = bm_make(t + 1u, mm, dna, rna, state, curr, slen + 1u, pamid, 1u);
          }
          route(live, inpam && t + 1u == T_end, csp, cep, hi, lo, cm);
        }
        /* terminal stage: one PAM search per pattern (index.hpp:310-312); empty PAM = a match */
        route(terminal && c < npams, P == 0u, sp, ep, shi, slo,
              bm_make(t, mm, dna, rna, state, curr, slen, c, 1u));
      }
      /* RNA bulge child: a guide symbol skipped, interval unchanged, '.' recorded */
      {
        uint64_t hi = shi, lo = slo;
        seq_append(hi, lo, slen, 1u);
        route(rna_ok, false, sp, ep, hi, lo, bm_make(t + 1u, mm, dna, r_rna, 2u, 1u, slen + 1u, 0u, 0u));
      }
      /* literal 'N' of the genome under a PAM 'N' (index.hpp:139-149) */
      if (sd.has_n && sd.nruns) {
        const bool want = active && inpam && pc == 4u;
        if (__ballot(want)) {
          bool live = false;
          uint32_t csp = 0, cep = 0;
          if (want) {
            const uint32_t na = occ_n(sd, sp), nb = occ_n(sd, ep + 1u);
            live = nb > na;
            csp = sd.CN + na;
            cep = sd.CN + nb - 1u;
          }
          uint64_t hi = shi, lo = slo;
          seq_append(hi, lo, slen, 5u);
          route(live, t + 1u == T_end, csp, cep, hi, lo,
                bm_make(t + 1u, mm, dna, rna, state, curr, slen + 1u, pamid, 1u));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    a.counts[slot] = n_match; /* same address, same value from every lane */
    if (bail) atomicOr(&a.work[1], 1u);
  }
}

/* flag[r] = 1 when sorted record r starts a new (guide, distance, index, sequence, rows) */
__global__ void k_bulge_flags(const gs_brec *srt, uint64_t T, uint32_t *flag) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T) return;
  bool f = true;
  if (r > 0) {
    const gs_brec &p = srt[r - 1], &c = srt[r];
    /* std::set<match> is keyed on the sequence alone (structures.hpp:40-42), per distance */
    f = !(p.g == c.g && (p.meta & 7u) == (c.meta & 7u) && ((p.meta >> 9) & 1u) == ((c.meta >> 9) & 1u) &&
          p.key_hi == c.key_hi && p.key_lo == c.key_lo);
  }
  flag[r] = f ? 1u : 0u;
}
__global__ void k_bulge_compact(const gs_brec *srt, const uint32_t *flag, const uint32_t *pos, uint64_t T,
                                gs_brec *uq, unsigned long long *cnt64, uint32_t *nmatch,
                                unsigned long long *nhits64) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= T || !flag[r]) return;
  const gs_brec h = srt[r];
  const uint32_t c = h.ep - h.sp + 1u;
  uq[pos[r]] = h;
  cnt64[pos[r]] = c;
  atomicAdd(&nmatch[h.g], 1u);
  atomicAdd(&nhits64[h.g], (unsigned long long)c);
}
struct gs_blocate_args {
  gs_strand_dev sd[2];
  const gs_brec *uq;
  const unsigned long long *hit_scan;
  const unsigned long long *guide_first;
  const uint64_t *offsets;
  gs_hit_ex *hits;
  uint64_t genome_length;
  uint32_t n_uq;
};
__global__ __launch_bounds__(WAVE) void k_bulge_locate(gs_blocate_args a) {
  const uint32_t r = blockIdx.x;
  if (r >= a.n_uq) return;
  const gs_brec m = a.uq[r];
  const uint32_t strand = (m.meta >> 9) & 1u;
  gs_hit_ex *out = a.hits + a.offsets[m.g] + (a.hit_scan[r] - a.guide_first[m.g]);
  const uint32_t cnt = m.ep - m.sp + 1u;
  for (uint32_t h = lane_id(); h < cnt; h += WAVE) {
    const uint64_t sa = a.sd[strand].sa[m.sp + h];
    gs_hit_ex o;
    o.pos = strand == 0 ? -(int64_t)sa : (int64_t)(a.genome_length - (sa + 1ull));
    o.key_hi = m.key_hi;
    o.key_lo = m.key_lo;
    o.mismatches = m.meta & 7u;
    o.dna_bulges = (uint8_t)((m.meta >> 3) & 7u);
    o.rna_bulges = (uint8_t)((m.meta >> 6) & 7u);
    o.index = (uint8_t)strand;
    o.seq_len = (uint8_t)((m.meta >> 10) & 63u);
    out[h] = o;
  }
}
__global__ void k_bulge_first(const unsigned long long *hit_scan, const uint64_t *first_rec, uint32_t n,
                              unsigned long long *guide_first) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) guide_first[g] = hit_scan[first_rec[g]];
}

struct gs_result_ex {
  std::vector<uint64_t> offsets;
  std::vector<gs_hit_ex> hits;
};

#define GS_TRY(expr)                 \
  do {                               \
    gs_status rc__ = (expr);         \
    if (rc__ != GS_OK) return rc__;  \
  } while (0)

extern "C" gs_status gs_enumerate_bulges(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                                         const char *guide_pams, uint32_t P, const char *alt_pams,
                                         uint32_t n_alt, uint32_t mismatches, uint32_t rna_bulges,
                                         uint32_t dna_bulges, uint32_t flags, gs_result_ex **out) {
  if (!ix || !out || (n && !guides) || (n && P && !guide_pams) || (n_alt && !alt_pams)) return GS_ERR_ARG;
  if (L < 1 || L > 31 || P > 8 || mismatches > 7 || n_alt > 3 || rna_bulges > 3 || dna_bulges > 3 ||
      L + dna_bulges + P > 32 || n >= (1ull << 30)) {
    gs_set_error("bulge path supports L<=31, P<=8, mismatches<=7, <=3 bulges of each kind, "
                 "L+dna_bulges+P<=32");
    return GS_ERR_UNSUPPORTED;
  }
  GS_HIP(hipSetDevice(ix->device));
  hipStream_t st = nullptr;
  const uint32_t n32 = (uint32_t)n;
  const bool dbg = getenv("GS_DEBUG") != nullptr;
#define BDBG(msg) do { if (dbg) { hipDeviceSynchronize(); fprintf(stderr, "[gs] bulge: %s\n", msg); fflush(stderr); } } while (0)
  gs_result_ex *res = new gs_result_ex();
  res->offsets.assign(n + 1, 0);
  if (n == 0) {
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
  dbuf d_g, d_rec, d_cnt, d_misc, d_off, d_a, d_b, d_flag, d_pos, d_uq, d_c64, d_scan, d_nm, d_nh, d_first,
      d_frec, d_goff, d_hits, d_tmp;
  GS_TRY(d_g.get(n * (size_t)(L + P) + 16));
  GS_TRY(d_rec.get(sizeof(gs_guide_rec) * n));
  GS_TRY(d_cnt.get(8 * n));
  GS_TRY(d_misc.get(64));
  GS_HIP(hipMemcpy(d_g.p, guides, n * (size_t)L, hipMemcpyHostToDevice));
  if (P) GS_HIP(hipMemcpy((char *)d_g.p + n * (size_t)L, guide_pams, n * (size_t)P, hipMemcpyHostToDevice));
  GS_HIP(hipMemset(d_misc.p, 0, 64));
  {
    gs_prep_args pa;
    memset(&pa, 0, sizeof(pa));
    pa.guides = (const uint8_t *)d_g.p;
    pa.guide_pams = (const uint8_t *)d_g.p + n * (size_t)L;
    for (uint32_t j = 0; j < n_alt; j++)
      for (uint32_t u = 0; u < P; u++) pa.alt[j][u] = (uint8_t)alt_pams[j * P + u];
    pa.out = (gs_guide_rec *)d_rec.p;
    pa.n_invalid = (uint32_t *)d_misc.p + 8;
    pa.n = n32;
    pa.L = L;
    pa.P = P;
    pa.n_alt = P ? n_alt : 0;
    pa.start = (flags & GS_FLAG_PAM_AT_START) ? 1 : 0;
    gs_launch_prepare(pa, st);
    uint32_t inv = 0;
    GS_HIP(hipMemcpy(&inv, (uint32_t *)d_misc.p + 8, 4, hipMemcpyDeviceToHost));
    if (inv) {
      gs_set_error("guide or PAM contains a symbol outside A,C,G,T (PAM: +N)");
      return GS_ERR_UNSUPPORTED;
    }
  }
  BDBG("prepared");
  gs_bsearch_args sa;
  sa.sd[0] = ix->strand[0].d;
  sa.sd[1] = ix->strand[1].d;
  sa.guides = (const gs_guide_rec *)d_rec.p;
  sa.recs = nullptr;
  sa.slot_off = nullptr;
  sa.counts = (uint32_t *)d_cnt.p;
  sa.work = (uint32_t *)d_misc.p;
  sa.n_items = 2 * n32;
  sa.L = L;
  sa.P = P;
  sa.m = mismatches;
  sa.max_rna = rna_bulges;
  sa.max_dna = dna_bulges;
  sa.max_iter = getenv("GS_BULGE_MAX_ITER") ? (uint32_t)atol(getenv("GS_BULGE_MAX_ITER")) : (1u << 26);
  const uint32_t grid_max = (uint32_t)gs_num_cus(ix->device) * 4u;
  uint32_t grid = (2 * n32 + BWAVES - 1) / BWAVES;
  if (grid > grid_max) grid = grid_max;
  /* pass 1: count matches per (guide, strand) */
  hipLaunchKernelGGL(k_search_bulge, dim3(grid), dim3(WAVE * BWAVES), 0, st, sa);
  BDBG("pass 1 done");
  std::vector<uint32_t> cnt(2 * n);
  GS_HIP(hipMemcpy(cnt.data(), d_cnt.p, 8 * n, hipMemcpyDeviceToHost));
  {
    uint32_t flag = 0;
    GS_HIP(hipMemcpy(&flag, (uint32_t *)d_misc.p + 1, 4, hipMemcpyDeviceToHost));
    if (flag) {
      gs_set_error("internal: bulge search exceeded its iteration bound");
      return GS_ERR_DEVICE;
    }
    if (getenv("GS_DEBUG")) {
      uint64_t tot = 0;
      for (auto c : cnt) tot += c;
      fprintf(stderr, "[gs] bulge pass 1: %llu match records for %llu guides; counts:", (unsigned long long)tot,
              (unsigned long long)n);
      for (size_t i = 0; i < cnt.size() && i < 16; i++) fprintf(stderr, " %u", cnt[i]);
      fprintf(stderr, "\n");
    }
  }
  std::vector<uint64_t> soff(2 * n + 1, 0);
  for (size_t i = 0; i < 2 * n; i++) soff[i + 1] = soff[i] + cnt[i];
  const uint64_t T = soff.back();
  if (T >= (1ull << 31)) {
    gs_set_error("more than 2^31 match records in one batch: use smaller batches with bulges");
    return GS_ERR_UNSUPPORTED;
  }
  if (T == 0) {
    *out = res;
    return GS_OK;
  }
  GS_TRY(d_off.get(8 * soff.size()));
  GS_TRY(d_a.get(sizeof(gs_brec) * T));
  GS_TRY(d_b.get(sizeof(gs_brec) * T));
  GS_HIP(hipMemcpy(d_off.p, soff.data(), 8 * soff.size(), hipMemcpyHostToDevice));
  /* pass 2: fill at exact offsets */
  GS_HIP(hipMemset(d_misc.p, 0, 8));
  sa.recs = (gs_brec *)d_a.p;
  sa.slot_off = (const uint64_t *)d_off.p;
  hipLaunchKernelGGL(k_search_bulge, dim3(grid), dim3(WAVE * BWAVES), 0, st, sa);
  BDBG("pass 2 done");
  /* canonical order: (guide, distance, index, sequence, row) */
  size_t tb = 0, tb2 = 0, tb3 = 0;
  GS_TRY(d_flag.get(4 * T));
  GS_TRY(d_pos.get(4 * T));
  GS_TRY(d_c64.get(8 * (T + 1)));
  GS_TRY(d_scan.get(8 * (T + 1)));
  GS_HIP(rocprim::merge_sort(nullptr, tb, (gs_brec *)d_a.p, (gs_brec *)d_b.p, (size_t)T, gs_brec_less(), st));
  GS_HIP(rocprim::exclusive_scan(nullptr, tb2, (uint32_t *)d_flag.p, (uint32_t *)d_pos.p, 0u, (size_t)T,
                                 rocprim::plus<uint32_t>(), st));
  GS_HIP(rocprim::exclusive_scan(nullptr, tb3, (unsigned long long *)d_c64.p, (unsigned long long *)d_scan.p,
                                 0ull, (size_t)T + 1, rocprim::plus<unsigned long long>(), st));
  if (tb2 > tb) tb = tb2;
  if (tb3 > tb) tb = tb3;
  GS_TRY(d_tmp.get(tb + 16));
  size_t tbs = tb;
  GS_HIP(rocprim::merge_sort(d_tmp.p, tbs, (gs_brec *)d_a.p, (gs_brec *)d_b.p, (size_t)T, gs_brec_less(), st));
  BDBG("sorted");
  const unsigned gT = (unsigned)((T + 255) / 256);
  hipLaunchKernelGGL(k_bulge_flags, dim3(gT), dim3(256), 0, st, (const gs_brec *)d_b.p, T, (uint32_t *)d_flag.p);
  tbs = tb;
  GS_HIP(rocprim::exclusive_scan(d_tmp.p, tbs, (uint32_t *)d_flag.p, (uint32_t *)d_pos.p, 0u, (size_t)T,
                                 rocprim::plus<uint32_t>(), st));
  GS_TRY(d_uq.get(sizeof(gs_brec) * T));
  GS_TRY(d_nm.get(4 * n));
  GS_TRY(d_nh.get(8 * n));
  GS_HIP(hipMemsetAsync(d_nm.p, 0, 4 * n, st));
  GS_HIP(hipMemsetAsync(d_nh.p, 0, 8 * n, st));
  GS_HIP(hipMemsetAsync(d_c64.p, 0, 8 * (T + 1), st));
  hipLaunchKernelGGL(k_bulge_compact, dim3(gT), dim3(256), 0, st, (const gs_brec *)d_b.p,
                     (const uint32_t *)d_flag.p, (const uint32_t *)d_pos.p, T, (gs_brec *)d_uq.p,
                     (unsigned long long *)d_c64.p, (uint32_t *)d_nm.p, (unsigned long long *)d_nh.p);
  BDBG("compacted");
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
  hipLaunchKernelGGL(k_bulge_first, dim3((n32 + 255) / 256), dim3(256), 0, st,
                     (const unsigned long long *)d_scan.p, (const uint64_t *)d_frec.p, n32,
                     (unsigned long long *)d_first.p);
  gs_blocate_args la;
  la.sd[0] = ix->strand[0].d;
  la.sd[1] = ix->strand[1].d;
  la.uq = (const gs_brec *)d_uq.p;
  la.hit_scan = (const unsigned long long *)d_scan.p;
  la.guide_first = (const unsigned long long *)d_first.p;
  la.offsets = (const uint64_t *)d_goff.p;
  la.hits = (gs_hit_ex *)d_hits.p;
  la.genome_length = ix->genome_length;
  la.n_uq = (uint32_t)nuq;
  if (nuq) hipLaunchKernelGGL(k_bulge_locate, dim3((unsigned)nuq), dim3(WAVE), 0, st, la);
  BDBG("located");
  res->hits.resize(H);
  if (H) GS_HIP(hipMemcpy(res->hits.data(), d_hits.p, sizeof(gs_hit_ex) * H, hipMemcpyDeviceToHost));
  GS_HIP(hipDeviceSynchronize());
  GS_HIP(hipGetLastError());
  *out = res;
  return GS_OK;
}
extern "C" gs_status gs_result_ex_get(const gs_result_ex *r, uint64_t *n_guides, const uint64_t **offsets,
                                      const gs_hit_ex **hits) {
  if (!r) return GS_ERR_ARG;
  if (n_guides) *n_guides = r->offsets.size() - 1;
  if (offsets) *offsets = r->offsets.data();
  if (hits) *hits = r->hits.data();
  return GS_OK;
}
extern "C" void gs_result_ex_free(gs_result_ex *r) { delete r; }

