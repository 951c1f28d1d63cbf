/*
 * gs_enumerate.hip -- the batch pipeline on the host: gs_enumerate_device and what it launches, in order (k_prepare,
 * k_search, the orderings, scans, locate), workspace and arena policy, the fall-backs.  Kernels: gs_search.hip, gs_order.hip,
 * gs_bigorder.hip, gs_tileorder.hip.
 */
#include "gs_kernels.h"

/* gs_seed.hip (launched here behind k_prepare; gs_estimate_heavy is its stand-alone form) */
__global__ void k_estimate_heavy(const gs_guide_rec *guides, uint32_t n, const uint4 *ptab0, const uint4 *ptab1, uint32_t k,
                                 uint32_t thresh, uint32_t *out);

#include <rocprim/rocprim.hpp>

#include <cmath>

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
  /* asked once per device: hipGetDeviceProperties fills a kilobyte-sized struct through the driver every time it is called,
   * and every enumerate and score call wants this one number */
  static std::atomic<int> cached[64];
  if (device >= 0 && device < 64) {
    const int c = cached[device].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) {
    (void)hipGetLastError();
    return 256;
  }
  const int n = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  if (device >= 0 && device < 64) cached[device].store(n, std::memory_order_relaxed);
  return n;
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
    if (rc == GS_ERR_DEVICE && ix && ix->share_timed_out) { /* (run_search: the sharing's bounded wait ran out) */
      ix->share_timed_out = false;
      /* this handle shares nothing for a while - whatever kept the launch off the chip may still be there -, whatever the
       * tuning switches say (GS_SHARE_MIN, GS_HEAVY, GS_SPLIT_SHARE): run_search looks at the back-off after it has read them */
      ix->share_backoff = 64;
      if (gs_opt(ix, "GS_DEBUG")) fprintf(stderr, "[gs] sharing timed out: batch redone with every item on its own wave\n");
      rc = enumerate_device_impl(ix, d_guides, n, L, d_guide_pams, P, alt_pams, n_alt, mismatches, flags, stream, d_offsets, d_hits, stats);
    }
    /* the batch's workspace did not fit.  First what earlier batches left on the handle and this one may not need goes -
     * a batch ordered device-wide leaves tens of bytes per record in a dozen arrays that a batch ordered in tiles never
     * touches, and the other way round (10^9 records: 70 GB either way) - and the batch is redone: every workspace
     * buffer grows again on demand.  Then the derived tables, one kind at a time: the strand tables' rotated copies,
     * then the PAM-pair tables. */
    if (rc == GS_ERR_NOMEM && ix) {
      (void)hipGetLastError();
      size_t freed = 0;
      for (gs_buffer *b : {&ix->w_b_src, &ix->w_b_cnt, &ix->w_b_prefix, &ix->w_b_recs, &ix->w_b_w0, &ix->w_b_w0b, &ix->w_b_idx,
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
    gs_launch_prepare(pa, st); /* (workgroups of 1,024: one atomic per workgroup and PAM pair) */
  }
  /* run_search's form estimate (k_estimate_heavy, gs_seed.hip: the guides whose own k-mer heads a giant interval) is
   * launched here, behind the records it reads, so that its two words come back with this stage's readback instead of
   * costing the step a host round trip of their own (25-70 us on this pool's hosts) */
  uint32_t pre_est[2] = {0, 0}, pre_est_thresh = 0;
  {
    uint32_t smin = ix->opt_share_min;
    if (const char *e = gs_opt(ix, "GS_SHARE_MIN")) smin = (uint32_t)std::max(0l, atol(e));
    if (n_chunks == 1 && smin != 0 && smin < (1u << 28) && ix->pt_k && ix->strand[0].ptab && ix->strand[1].ptab && !gs_opt(ix, "GS_NO_FORM_ESTIMATE")) {
      pre_est_thresh = 8u * smin;
      hipLaunchKernelGGL(k_estimate_heavy, dim3((n32 + 255) / 256), dim3(256), 0, st, (const gs_guide_rec *)ix->w_grec.p, n32,
                         (const uint4 *)ix->strand[0].ptab, (const uint4 *)ix->strand[1].ptab, ix->pt_k, pre_est_thresh, d_work + 10);
    }
  }
  /* guides the fast path does not encode get empty hit lists and a flag; the batch goes on */
  uint32_t h_invalid = 0, h_pairs[17] = {0};
  GS_HIP(hipMemcpyAsync(&h_invalid, d_invalid, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(h_pairs, (char *)ix->w_misc.p + 256, sizeof(h_pairs), hipMemcpyDeviceToHost, st));
  if (pre_est_thresh) GS_HIP(hipMemcpyAsync(pre_est, d_work + 10, 8, hipMemcpyDeviceToHost, st));
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
    /* The list depends on the text's N runs and on (L, P, whether it is bucketed) only - not on the batch's guides or
     * patterns: the handle keeps the last one it uploaded (a batch of the same shape finds it in place: the host's
     * pass over the runs and three blocking copies were 0.1 ms of every 17 ms step). */
    uint32_t cand_from = 256;
    if (const char *e = gs_opt(ix, "GS_CAND_BUCKETS_FROM")) cand_from = (uint32_t)atol(e);
    const bool cand_buckets_ok = !(mismatches > 3 || L < 20 || gs_opt(ix, "GS_NO_CAND_BUCKETS"));
    const uint64_t cand_key = (uint64_t)L | ((uint64_t)P << 8) | ((uint64_t)(cand_buckets_ok ? 1u : 0u) << 16) | ((uint64_t)cand_from << 32);
    const bool cand_hit = ix->cand_key == cand_key && ix->w_cand.p != nullptr;
    std::vector<uint4> cand[2];
    std::vector<uint32_t> bidx[2];
    const uint32_t W = L + P;
    const uint64_t len = ix->genome_length;
    auto code = [](uint8_t c) -> int { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; };
    if (!cand_hit)
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
    size_t n_bidx[2] = {0, 0};
    if (cand_hit) {
      n_cand[0] = ix->cand_n[0];
      n_cand[1] = ix->cand_n[1];
      n_bidx[0] = ix->cand_bidx[0];
      n_bidx[1] = ix->cand_bidx[1];
    } else {
      n_cand[0] = (uint32_t)cand[0].size();
      n_cand[1] = (uint32_t)cand[1].size();
      /* behind the windows: per strand with many of them, the bucket index (4 x 1025 offsets, 4 x n places) */
      for (uint32_t s = 0; s < 2; s++) {
        if (n_cand[s] <= cand_from || !cand_buckets_ok) continue;
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
        n_bidx[s] = bidx[s].size();
      }
      ix->cand_key = ~0ull; /* (valid again once everything below is in place) */
    }
    if (n_cand[0] + n_cand[1]) {
      const size_t b_w = 16 * (size_t)(n_cand[0] + n_cand[1]);
      if (!cand_hit && (rc = gs_reserve(ix->w_cand, b_w + 4 * (n_bidx[0] + n_bidx[1]) + 16)) != GS_OK) return rc;
      uint4 *dc = (uint4 *)ix->w_cand.p;
      if (!cand_hit) {
        if (n_cand[0]) GS_HIP(hipMemcpy(dc, cand[0].data(), 16 * (size_t)n_cand[0], hipMemcpyHostToDevice));
        if (n_cand[1]) GS_HIP(hipMemcpy(dc + n_cand[0], cand[1].data(), 16 * (size_t)n_cand[1], hipMemcpyHostToDevice));
      }
      d_cand[0] = dc;
      d_cand[1] = dc + n_cand[0];
      uint32_t *di = (uint32_t *)((char *)ix->w_cand.p + b_w);
      for (uint32_t s = 0; s < 2; s++) {
        if (n_bidx[s] == 0) continue;
        if (!cand_hit) GS_HIP(hipMemcpy(di, bidx[s].data(), 4 * n_bidx[s], hipMemcpyHostToDevice));
        d_cand_off[s] = di;
        d_cand_ids[s] = di + 4u * 1025u; /* chunk c's places: from c * n_cand[s] on */
        di += n_bidx[s];
      }
    }
    if (!cand_hit) {
      ix->cand_n[0] = n_cand[0];
      ix->cand_n[1] = n_cand[1];
      ix->cand_bidx[0] = n_bidx[0];
      ix->cand_bidx[1] = n_bidx[1];
      ix->cand_key = n_cand[0] + n_cand[1] ? cand_key : ~0ull; /* (no windows: nothing to keep, nothing to upload) */
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
        for (gs_buffer *b : {&ix->w_slots2, &ix->w_b_s}) {
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
    GS_HIP(hipMemsetAsync(d_work + 9, 0x80, 4, st)); /* (0x80808080: what a launch without items finds in its work counter) */
    if (with_arena) {
      GS_HIP(hipMemsetAsync(d_arena_next, 0, 4, st));
      /* every chunk empty until a wave says whose it is: waves reserve several per visit to the counter (k_search) */
      GS_HIP(hipMemsetAsync((uint32_t *)ix->w_arena_meta.p + arena_chunks, 0xFF, 4 * (size_t)arena_chunks, st));
      GS_HIP(hipMemsetAsync(ix->w_arena_meta.p, 0, 4 * (size_t)arena_chunks, st));
    }
    gs_search_args sa;
    memset(&sa, 0, sizeof(sa));
    if (with_arena) ix->last_share[0] = ix->last_share[1] = ix->last_share[2] = ix->last_share[3] = ix->last_share[4] = ix->last_share[5] = ix->last_share[6] = 0; /* (of the main pass: a redo shares nothing) */
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
    if (sa.dbg_skip & 4u) sa.n_rec_full = sa.n_rec_a = sa.n_rec_b = sa.n_rec_a8 = 0u; /* (experiments: no recipe at all - what an item costs before its first seed) */
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
    /* Three forms.  Plain: every item with its wave.  Heavy, one launch (GS_HEAVY=1): heavy verification passes are
     * published and the waves that ran out of items run them - the second level of the verification four rows per lane; twice the
     * code, registers in scratch: 32 ms against 22 on 1 M guides of a genome without repeat families.  Split (GS_SPLIT_SHARE=2): the
     * plain form publishes and leaves (+4 % on that batch), the heavy form - no items of its own - runs the packages in a launch
     * of the lowest priority beside it, taking the slots the first launch's waves leave.  By itself a handle picks from what the
     * last batch of the same shape showed (gs_search_args::hpass): no heavy pass - plain; one per sixteen items, or any in a
     * batch of at most 64 items per wave slot - heavy (a repeat-rich batch: 9.3 ms per 20,000 guides against 9.7 split, 16.2
     * plain); fewer - split (1 M light guides + 8 of an Alu-like family, 650,000 hits each: 26.7 ms against 33.6 plain, 32.7 heavy). */
    /* (a handle whose sharing launch was not resident as a whole backs off for 64 batches, then tries again) */
    if (with_arena && ix->share_backoff != 0) ix->share_backoff--;
    const bool share_ok = with_arena && !walk && n_chunks == 1 && !count_req && share_min != 0 && ix->share_backoff == 0;
    /* which form: from THIS batch - the guides whose own k-mer heads an interval of 8 x share_min rows or more in a strand
     * table sit in a repeat family, their items are the ones with heavy passes (k_estimate_heavy: one table read per guide
     * and strand, 20 us) -, and from what the last batch of the same shape counted (a guide a substitution away from a
     * family's consensus has heavy passes without a heavy k-mer of its own) */
    uint32_t est[2] = {0, 0};
    if (share_ok && with_arena && !gs_opt(ix, "GS_NO_FORM_ESTIMATE")) {
      if (pre_est_thresh == 8u * share_min && guides == (const gs_guide_rec *)ix->w_grec.p && ng == n32) {
        est[0] = pre_est[0]; /* (the main pass: estimated behind k_prepare) */
        est[1] = pre_est[1];
      } else {
        const gs_status er = gs_estimate_heavy(ix, guides, ng, 8u * share_min, d_work + 10, st, est);
        if (er != GS_OK) return er;
      }
      if (gs_opt(ix, "GS_DEBUG")) fprintf(stderr, "[gs] form estimate: %u guides with a heavy k-mer of their own, the largest interval %u rows\n", est[0], est[1]);
    }
    const uint32_t est_heavy = est[0];
    const bool seen_last = mismatches < 8 &&
                           ix->seen_key[mismatches] == (((uint64_t)L << 32) | ((uint64_t)P << 16) | (n_alt << 8) | (flags & GS_FLAG_PAM_AT_START)) &&
                           ix->seen_hpass[mismatches] != 0;
    /* heavy items expected in this batch (each guide counts for two items; the last batch's count scaled to this batch's size) */
    const double hp_last = seen_last && ix->seen_items[mismatches] ? (double)ix->seen_hpass[mismatches] * (2.0 * ng) / (double)ix->seen_items[mismatches] : 0.0;
    const double hp = std::max(2.0 * (double)est_heavy, hp_last);
    const bool seen = hp >= 1.0;
    const bool dense = seen && (16.0 * hp >= 2.0 * (double)ng || 2 * (uint64_t)ng <= 64ull * (uint64_t)cus * 32u);
    ix->last_share[7] = est_heavy; /* (gs_index_last_sharing: guides of the batch with a heavy k-mer of their own) */
    bool heavy = share_ok && dense;
    /* few heavy items in a large batch: the two launches pay only when an item is too large to hide behind the rest of the
     * batch on one wave - the publishing form runs k_search's one-launch kernel, a quarter slower than the two seeding launches
     * on the light guides (1 M light guides + 8 guides of 650,000 hits, the largest k-mer interval 145,000 rows: every item
     * with its wave 24.9 ms, two launches 29.4, profiles/r06_mixed_batch.txt).  From 2^19 rows under one k-mer on (GS_SPLIT_FROM) */
    const uint32_t split_from = gs_opt(ix, "GS_SPLIT_FROM") ? (uint32_t)atol(gs_opt(ix, "GS_SPLIT_FROM")) : (1u << 19);
    uint32_t split = share_ok && seen && !dense && est[1] >= split_from ? 2u : 0u; /* 1: the second launch behind the first; 3: before it (tests) */
    if (const char *e = gs_opt(ix, "GS_HEAVY")) {
      heavy = atol(e) != 0 && share_ok;
      split = 0u;
    }
    if (const char *e = gs_opt(ix, "GS_SPLIT_SHARE")) {
      split = share_ok ? (uint32_t)std::min(3l, std::max(0l, atol(e))) : 0u;
      if (split) heavy = false;
    }
    const uint32_t weu = walk ? GS_WAVES_EU : heavy ? GS_WAVES_EU_HEAVY : spec ? GS_WAVES_EU_PD : GS_WAVES_EU_FAST;
    if (per_cu > weu) per_cu = weu; /* 4 SIMDs x weu waves = weu four-wave workgroups per CU */
    /* The sharing forms need their whole grid on the chip at once (a helper waits for a package only a resident wave can
     * write): never more workgroups than the runtime says the heavy kernel gets per CU - registers, LDS AND its scratch. */
    uint32_t heavy_per_cu = std::min<uint32_t>((uint32_t)(160u * 1024u / lds_wg), GS_WAVES_EU_HEAVY);
    if (heavy || split) {
      int occ = 0;
      hipError_t oe = spec ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_heavy_pd, WAVE * SEARCH_WAVES, dyn)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_search_heavy, WAVE * SEARCH_WAVES, dyn);
      if (oe != hipSuccess) (void)hipGetLastError();
      if (oe == hipSuccess && occ > 0 && (uint32_t)occ < heavy_per_cu) heavy_per_cu = (uint32_t)occ;
      if (heavy && per_cu > heavy_per_cu) per_cu = heavy_per_cu;
    }
    uint32_t grid = (uint32_t)cus * per_cu;
    const uint32_t need = (2 * ng + SEARCH_WAVES - 1) / SEARCH_WAVES;
    if (grid > need) grid = need;
    if (gs_opt(ix, "GS_DEBUG")) {
      int occ = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, walk ? k_search_walk : k_search_fast, WAVE * SEARCH_WAVES, dyn);
      fprintf(stderr, "[gs] k_search (%s): grid %u x %u threads, LDS %zu B per workgroup, %d workgroups per CU resident\n",
              walk ? "walk" : "table", grid, WAVE * SEARCH_WAVES, lds_wg, occ);
    }
    if (heavy || split) {
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
    /* the form of a batch whose every pattern has its tables and that shares no item: 0 - k_search_fast_pd (one launch, every item
     * sets itself up), 1 - two launches from descriptors (gs_seed.hip), 2 - ... with the guides scheduled by their symbols */
    /* (budgets beyond four substitutions stay with the one launch: a hit is no longer rare there - 10^4 per guide at m <= 6 -
     * and rebuilding every hit's path from its recipe costs what the descriptors save: 65.0 against 63.1 ms per 20,000
     * guides at m <= 6, 2.6 against 2.6 at m <= 4, profiles/r06_seed_forms_m6.txt) */
    uint32_t seed_form = (spec && sa.shq == nullptr && !heavy && !split && mismatches <= 4u) ? 2u : 0u;
    if (const char *e = gs_opt(ix, "GS_SEED_FORM")) seed_form = seed_form ? (uint32_t)std::min(2l, std::max(0l, atol(e))) : 0u;
    const uint32_t seed_sort_from = gs_opt(ix, "GS_SEED_SORT_FROM") ? (uint32_t)atol(gs_opt(ix, "GS_SEED_SORT_FROM")) : 4096u;
    uint32_t seed_grid = 0;
    if (seed_form) {
      const size_t lds_seed = sizeof(uint4) * (VQ_CAP + 32 + 24 + 6) * SEARCH_WAVES;
      seed_grid = (uint32_t)cus * std::min<uint32_t>((uint32_t)(160u * 1024u / lds_seed), GS_WAVES_EU_SEED);
      if (seed_grid > need) seed_grid = need;
    }
    gs_search_args sh_args;
    uint32_t sh_grid = 0;
    auto launch_helpers = [&](hipStream_t hs) {
      if (spec)
        hipLaunchKernelGGL(k_search_heavy_pd, dim3(sh_grid), dim3(WAVE * SEARCH_WAVES), dyn, hs, sh_args);
      else
        hipLaunchKernelGGL(k_search_heavy, dim3(sh_grid), dim3(WAVE * SEARCH_WAVES), dyn, hs, sh_args);
    };
    if (with_arena) ix->last_share[6] = sa.shq == nullptr ? 0u : heavy ? 1u : 2u;
    GS_HIP(hipEventRecord(ix->ev[1], st));
    for (uint32_t c = 0; c < n_chunks; c++) { /* four PAM patterns per pass, appending to the same slots */
      sa.guides = guides + (size_t)c * ng;
      sa.append = c ? 1u : 0u;
      if (c) GS_HIP(hipMemsetAsync(d_work, 0, 4, st));
      if (walk)
        hipLaunchKernelGGL(k_search_walk, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (spec && count_req && !seed_form)
        hipLaunchKernelGGL(k_search_count_pd, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (split && sa.shq != nullptr) {
        /* the launch without items: on a stream of the lowest priority beside the search launch (its workgroups get the
         * slots the search launch's waves leave), or behind it on the same stream */
        sh_args = sa;
        sh_args.helper_only = 1u;
        sh_args.work = d_work + 9; /* a counter that is past the items from the start */
        sh_grid = (uint32_t)cus * heavy_per_cu;
        bool side = split == 2u;
        if (side && !ix->st_help) {
          int lo = 0, hi = 0;
          hipStream_t hs = nullptr;
          bool ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
                    hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, lo) == hipSuccess;
          for (hipEvent_t &e : ix->ev_help) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
          if (ok) {
            ix->st_help = hs;
          } else { /* no second stream to be had: the second launch goes behind the first on the caller's */
            (void)hipGetLastError();
            if (hs) (void)hipStreamDestroy(hs);
            for (hipEvent_t &e : ix->ev_help) {
              if (e) (void)hipEventDestroy(e);
              e = nullptr;
            }
            side = false;
          }
        }
        if (side) { /* the queue's control words are zeroed: the other stream may start */
          GS_HIP(hipEventRecord(ix->ev_help[0], st));
          GS_HIP(hipStreamWaitEvent(ix->st_help, ix->ev_help[0], 0));
        }
        if (split == 3u) launch_helpers(st); /* (tests: a launch that comes too early leaves at once, everything is left for the one behind) */
        if (spec)
          hipLaunchKernelGGL(k_search_pub_pd, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
        else
          hipLaunchKernelGGL(k_search_pub, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
        if (split != 3u) launch_helpers(side ? ix->st_help : st);
        if (side) {
          GS_HIP(hipEventRecord(ix->ev_help[1], ix->st_help));
          GS_HIP(hipStreamWaitEvent(st, ix->ev_help[1], 0));
        }
      } else if (spec && sa.shq != nullptr)
        hipLaunchKernelGGL(k_search_heavy_pd, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (sa.shq != nullptr)
        hipLaunchKernelGGL(k_search_heavy, dim3(grid), dim3(WAVE * SEARCH_WAVES), dyn, st, sa);
      else if (seed_form) {
        /* the two-launch form (gs_seed.hip): descriptors per guide, the guides scheduled by their last / first symbols,
         * the other strand's seeds + the window list, then this strand's seeds appending */
        gs_search_args sb;
        gs_status r2 = gs_seed_describe(ix, sa, ng, seed_form >= 2u && ng >= seed_sort_from, st, &sb);
        /* one item per visit: each XCD has its own counter (a sixteenth of the visits one word took), and the items that share
         * a piece of a table are then in flight together */
        sb.seed_opt = gs_opt(ix, "GS_SEED_OPT") ? (uint32_t)atol(gs_opt(ix, "GS_SEED_OPT")) : 0u;
        sb.take = gs_opt(ix, "GS_SEED_TAKE") ? (uint32_t)std::max(1l, atol(gs_opt(ix, "GS_SEED_TAKE"))) : 1u;
        if (r2 == GS_OK) r2 = gs_seed_launch(sb, seed_grid, count_req, st);
        if (r2 != GS_OK) return r2;
        if (with_arena) ix->last_share[6] = 3u; /* (gs_index_last_sharing: the form of the main pass) */
      } else if (spec)
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
    /* (packages reserved beyond the queue's size were run by their owners: what the helpers could draw is the smaller) */
    if (d_shctl && split && sh_grid != 0u && h_ctl[32] < std::min(h_ctl[0], sa.shq_cap) && ((const uint32_t *)(h7 + 16))[5] == 0u) {
      /* packages nobody ran: the launch without items was on the chip before the one it serves and left (k_search_body's
       * first lines) - again, behind it */
      launch_helpers(st);
      GS_HIP(hipEventRecord(ix->ev[2], st));
      GS_HIP(hipMemcpyAsync(h7, d_stats, sizeof(h7), hipMemcpyDeviceToHost, st));
      GS_HIP(hipMemcpyAsync(h_ctl, d_shctl, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
      GS_HIP(hipStreamSynchronize(st));
      GS_HIP(hipGetLastError());
      ix->last_share[5]++;
    }
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
        hipLaunchKernelGGL((k_share_fix<SH_SMALLSEG>), dim3(n_sh), dim3(256), 0, st, fa); /* a workgroup per item */
        hipLaunchKernelGGL((k_share_fix<SH_MAXSEG>), dim3(std::min<uint32_t>(n_sh, (uint32_t)cus * 3u)), dim3(256), 0, st, fa);
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
    if (sa.shq != nullptr && gs_opt(ix, "GS_DBG_SHARE_TIMEOUT")) ((uint32_t *)(h7 + 16))[5] |= 2u; /* (tests: as if a helping wave had given up) */
    if (((const uint32_t *)(h7 + 16))[5] != 0u) {
      if ((((const uint32_t *)(h7 + 16))[5] & 2u) != 0u && sa.shq != nullptr) {
        /* a wave waited ~3 s for a package that was reserved and never written: the launch's waves were not all on the
         * chip together (another process on the device, a profiler holding CUs).  Not a wrong result - none is returned -
         * and not the end of the handle: the caller redoes the batch with every item on its own wave. */
        ix->share_timed_out = true;
        gs_set_error("internal: a wave gave up waiting for a shared verification pass (the launch was not resident as a whole)");
        return GS_ERR_DEVICE;
      }
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
    const uint32_t gsz = gs_lane_group(ng); /* guides a wave takes at a time: a lane each where one record needs no order */
    uint32_t grid = ((ng + gsz - 1) / gsz + ow - 1) / ow;
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
    const uint32_t gsz = gs_lane_group(ng);
    hipLaunchKernelGGL(k_locate, dim3((ng + gsz - 1) / gsz), dim3(WAVE), lds, st, la);
  };

  /* ---- guides whose match count exceeds what k_order sorts in LDS: `n_set`
   * guides whose items' records lie in the main slot array (stride cap) or, for guides on the redo
   * list, in the exact-size array slots2/slot_off2.  Leaves nmatch/nhits per set guide and the
   * sorted arrays k_big_locate reads once the CSR offsets exist. */
  uint64_t big_T = 0;
  bool big_used = false;
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
    return gbits + 4 + big_rbits <= 64;
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
    if (!big_fits_v2(n_set)) {
      gs_set_error("the device-wide ordering's sort word (guide, class, sequence rank) does not fit 64 bits: use smaller batches");
      return GS_ERR_UNSUPPORTED;
    }
    if ((r2 = gs_reserve(ix->w_b_recs, 16 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_w0, 8 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_w0b, 8 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_idx, 4 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_idxb, 4 * (T + 1))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_keep, 4 * (T + 2))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_keeps, 4 * (T + 2))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_rows, 8 * (T + 2))) != GS_OK) return r2;
    if ((r2 = gs_reserve(ix->w_b_rowss, 8 * (T + 2))) != GS_OK) return r2;
    uint4 *recs = (uint4 *)ix->w_b_recs.p;
    unsigned long long *w0 = (unsigned long long *)ix->w_b_w0.p, *w0b = (unsigned long long *)ix->w_b_w0b.p;
    uint32_t *idx = (uint32_t *)ix->w_b_idx.p, *idxb = (uint32_t *)ix->w_b_idxb.p;
    uint32_t gbits = 1;
    while ((1ull << gbits) < n_set) gbits++;
    const uint32_t rbits = big_rbits;
    const unsigned long long pam_mul = big_pam_mul;
    big_gshift = 4 + rbits;
    if (T) {
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
    hipLaunchKernelGGL(k_count_stats, dim3(std::min<uint32_t>((2 * n32 + 1023) / 1024, 256u)), dim3(256), 0, st,
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

/* ---- gs_index_prepare: the first batch's one-off work ahead of the first job ------------------------------------ */
__global__ void k_prepare_fill(uint8_t *guides, uint8_t *pams, uint64_t n, uint32_t L, uint32_t P, uint4 pat /* <= 8 symbols in x, y */) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t h = (i + 1u) * 0x9E3779B97F4A7C15ull;
  for (uint32_t t = 0; t < L; ++t) {
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    guides[i * L + t] = "ACGT"[(h >> 61) & 3u];
  }
  for (uint32_t u = 0; u < P; ++u) pams[i * P + u] = (uint8_t)((u < 4u ? pat.x >> (8u * u) : pat.y >> (8u * (u - 4u))) & 0xFFu);
}
extern "C" gs_status gs_index_prepare(gs_index *ix, uint64_t n, uint32_t L, const char *pam, uint32_t P, const char *alt_pams,
                                      uint32_t n_alt, uint32_t mismatches, uint32_t flags) {
  GS_HANDLE_LOCK(ix);
  if (!ix || (P && !pam) || (n_alt && !alt_pams) || L < 1 || L > 31 || P > 8 || n >= (1ull << 31)) return GS_ERR_ARG;
  if (n == 0) return GS_OK;
  GS_HIP(hipSetDevice(ix->device));
  uint8_t *d_g = nullptr, *d_p = nullptr;
  GS_HIP(hipMalloc(&d_g, n * L));
  if (hipMalloc(&d_p, n * (P ? P : 1u)) != hipSuccess) {
    (void)hipFree(d_g);
    (void)hipGetLastError();
    return GS_ERR_NOMEM;
  }
  uint4 pat = make_uint4(0u, 0u, 0u, 0u);
  for (uint32_t u = 0; u < P; ++u) (u < 4u ? pat.x : pat.y) |= (uint32_t)(uint8_t)pam[u] << (8u * (u & 3u));
  hipLaunchKernelGGL(k_prepare_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_g, d_p, n, L, P, pat);
  const void *off = nullptr, *hits = nullptr;
  gs_result_view v;
  /* what the handle learns from its batches (slot sizing, the search's form, the queue of shared passes) must come from the
   * caller's guides, not from this synthetic few-hit batch: saved and put back.  The call does overwrite the device
   * buffers an earlier gs_enumerate_device left its results in (include/guidescan_amd.h says so). */
  double s_mean[8], s_max[8];
  uint64_t s_hp[8], s_it[8], s_key[8];
  for (int i = 0; i < 8; i++) {
    s_mean[i] = ix->seen_mean[i];
    s_max[i] = ix->seen_max[i];
    s_hp[i] = ix->seen_hpass[i];
    s_it[i] = ix->seen_items[i];
    s_key[i] = ix->seen_key[i];
  }
  const uint64_t s_pk = ix->shq_packages;
  const gs_status rc = gs_enumerate_device(ix, d_g, n, L, d_p, P, alt_pams, n_alt, mismatches, flags & ~GS_FLAG_COUNT_REQUESTS, nullptr, &off, &hits, &v);
  for (int i = 0; i < 8; i++) {
    ix->seen_mean[i] = s_mean[i];
    ix->seen_max[i] = s_max[i];
    ix->seen_hpass[i] = s_hp[i];
    ix->seen_items[i] = s_it[i];
    ix->seen_key[i] = s_key[i];
  }
  ix->shq_packages = s_pk;
  (void)hipDeviceSynchronize();
  (void)hipFree(d_g);
  (void)hipFree(d_p);
  return rc;
}
