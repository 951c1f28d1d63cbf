/*
 * gs_score.hip -- CFD per hit and specificity per guide on the device (SURVEY.md section 8a
 * row a10): calculate_cfd (include/genomics/printer.hpp:98-113) and the aggregation of
 * get_csv_lines (printer.hpp:251-297) / off_target_fields (printer.hpp:115-170), with the
 * reference's arithmetic kept operation for operation - float accumulator, double table,
 * `cfd = float(double(cfd) * score)`, float sum in canonical hit order, one float division -
 * so the results are bit-identical to the host path (gs_calculate_cfd / gs_format_guide).
 *
 * One wavefront per guide at a time: the lanes decode and score 64 hits in parallel, then the
 * wave adds the 64 scores ONE AFTER THE OTHER in hit order (float addition does not commute
 * with reordering; the reference sums sequentially), applying the --max-off-targets and
 * chromosome-boundary rules on the way.  Streaming: 16 B read + 4 B written per hit.
 */
#include "gs_common.h"
#include "cfd_table.h"

#define WAVE 64
#define SCORE_WAVES 4

struct gs_score_args {
  const uint8_t *guides;   /* n*L ASCII (A,C,G,T) */
  const uint64_t *offsets; /* n+1 */
  const gs_hit *hits;
  const uint64_t *chr_cum; /* n_chr+1 cumulative chromosome lengths */
  const double *tab;       /* 320 mismatch scores + 16 PAM scores */
  float *cfd;              /* per hit, or nullptr */
  float *spec;             /* per guide */
  long long max_off;       /* -1 = none */
  uint32_t n, L, P, n_chr, start, sam;
};

__device__ __forceinline__ int sc_bidx(uint32_t c) {
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}
__device__ __forceinline__ uint32_t sc_comp_upper(uint32_t c) { /* sequences.cxx:14-26 on upper case */
  return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
}

/* match.sequence[i] as (upper-case symbol, is_lower): the decoding of gs_decode_sequence */
__device__ __forceinline__ uint32_t sc_seq_at(const uint8_t *guide, uint32_t L, uint32_t start,
                                              uint64_t path, uint32_t i, bool &lower) {
  lower = false;
  if (i < L) {
    const uint32_t qc = start ? guide[L - 1u - i] : sc_comp_upper(guide[i]);
    const uint32_t code = (uint32_t)(path >> (57u - 2u * i)) & 3u;
    if (code == 0u) return qc;
    const int q = sc_bidx(qc);
    int a = (int)code - 1;
    if (a >= q) a++;
    lower = true; /* index.hpp:243 */
    return a == 0 ? 'A' : a == 1 ? 'C' : a == 2 ? 'G' : 'T';
  }
  const uint32_t code = (uint32_t)(path >> (56u - 2u * L - 3u * (i - L))) & 7u;
  return code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : code == 3 ? 'N' : 'T';
}

/* src/genomics/structures.cxx:7-52: true when the hit is dropped at a chromosome boundary */
__device__ __forceinline__ bool sc_sentinel(const uint64_t *cum, uint32_t n_chr, long long pos, uint32_t L,
                                            uint32_t P) {
  const bool minus = pos < 0;
  const unsigned long long ab = (unsigned long long)(minus ? -pos : pos);
  if (n_chr == 0u || ab >= cum[n_chr]) return true;
  uint32_t lo = 0, hi = n_chr; /* first c with cum[c+1] > ab */
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (cum[mid + 1u] > ab)
      hi = mid;
    else
      lo = mid + 1u;
  }
  const long long off = (long long)(ab - cum[lo]);
  const long long len = (long long)(cum[lo + 1u] - cum[lo]);
  long long s, e;
  if (!minus) {
    e = off + 1;
    s = e - (long long)L - (long long)P + 1;
  } else {
    s = off + 1;
    e = s + (long long)L + (long long)P - 1;
  }
  return s < 0 || e > len; /* :46-48 */
}

/* Two kernels.  k_score_hits: one thread per hit - its CFD (printer.hpp:98-113) and the three facts the
 * aggregation needs (distance, dropped at a chromosome boundary, perfect xGG hit); every hit is
 * independent, so guides with 10^5 hits spread over the whole chip.  k_score_sum: one wavefront per guide
 * walks its hits in order - only the float additions are sequential (they are not associative and the
 * reference adds hit by hit) and --max-off-targets' per-distance counters. */
__global__ __launch_bounds__(256) void k_score_hits(gs_score_args a, uint64_t n_hits, float *cf, uint8_t *info) {
  __shared__ double s_tab[336];
  for (uint32_t i = threadIdx.x; i < 336u; i += blockDim.x) s_tab[i] = a.tab[i];
  __syncthreads();
  const uint32_t L = a.L, P = a.P, slen = L + P;
  /* pam = match_sequence.substr(20, 3) when the sequence has at least 20 symbols */
  const uint32_t pam_len = slen < 20u ? 0u : (slen - 20u < 3u ? slen - 20u : 3u);
  const bool scored = L == 20u && pam_len == 3u; /* printer.hpp:99 */
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < n_hits; h += (uint64_t)gridDim.x * blockDim.x) {
    /* the guide of hit h: last g with offsets[g] <= h.  One binary search per wavefront (for its first hit,
     * on scalar loads), then a short walk: 64 consecutive hits belong to one guide on a repeat-rich
     * batch and to five on a batch with 13 hits per guide */
    const uint64_t h_lane0 = h - (threadIdx.x & (WAVE - 1u));
    const uint64_t h_first = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(h_lane0 >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)h_lane0); /* wave-uniform: scalar loads below */
    uint32_t lo = 0, hi = a.n;
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (a.offsets[mid] <= h_first)
        lo = mid;
      else
        hi = mid;
    }
    while (lo + 1u < a.n && a.offsets[lo + 1u] <= h) lo++;
    const uint8_t *gd = a.guides + (size_t)lo * L;
    const gs_hit hit = a.hits[h];
    const uint64_t path = (hit.key >> 1) & ((1ull << 59) - 1ull); /* key bits 59:1: position 0 at the top */
    const uint32_t d = (uint32_t)(hit.key >> 61);
    float c = 1.0f;
    uint32_t pgg = 0u;
    bool lw;
    if (scored) {
      /* positions in order (each product is rounded to float before the next, printer.hpp:104-109).  A
       * position enters when match_sequence[i] differs from the guide's symbol, case-sensitively: always
       * where the hit has a substitution (lower case); elsewhere never when the guide was searched as
       * given, while with --start the two strings are compared as the reference compares them */
      uint64_t todo = a.start ? 0xFFFFFull : 0ull;
      if (!a.start)
        for (uint32_t i = 0; i < 20u; ++i) todo |= (uint64_t)(((path >> (57u - 2u * i)) & 3ull) != 0ull) << i;
      while (todo) {
        const uint32_t i = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t su = sc_seq_at(gd, L, a.start, path, i, lw);
        const uint32_t mu = sc_comp_upper(su); /* match_sequence[i] = complement(sequence[i]), case kept */
        const uint32_t gc = gd[i];
        if (lw || gc != mu) {
          const int r = sc_bidx(gc);  /* 'T' is looked up as 'U': same slot */
          const int dd = sc_bidx(su); /* toupper(complement(match_sequence[i])) == sequence[i] */
          const double sc = (r >= 0 && dd >= 0) ? s_tab[(r * 4 + dd) * 20 + (int)i] : 0.0;
          c = (float)((double)c * sc);
        }
      }
      const int b1 = sc_bidx(sc_comp_upper(sc_seq_at(gd, L, a.start, path, 21u, lw)));
      const int b2 = sc_bidx(sc_comp_upper(sc_seq_at(gd, L, a.start, path, 22u, lw)));
      const double ps = (b1 >= 0 && b2 >= 0) ? s_tab[320 + b1 * 4 + b2] : 0.0;
      c = (float)((double)c * ps);
    }
    if (d == 0u && pam_len == 3u) { /* perfect NGG-style hit, printer.hpp:145-146 / :262 */
      bool l1, l2;
      const uint32_t p1 = sc_comp_upper(sc_seq_at(gd, L, a.start, path, 21u, l1));
      const uint32_t p2 = sc_comp_upper(sc_seq_at(gd, L, a.start, path, 22u, l2));
      pgg = (!l1 && !l2 && p1 == 'G' && p2 == 'G') ? 1u : 0u;
    }
    const uint32_t sent = sc_sentinel(a.chr_cum, a.n_chr, (long long)hit.pos, L, P) ? 1u : 0u;
    cf[h] = c;
    info[h] = (uint8_t)(d | (sent << 3) | (pgg << 4));
  }
}

__global__ __launch_bounds__(WAVE *SCORE_WAVES) void k_score_sum(gs_score_args a, const float *cf, const uint8_t *info) {
  const uint32_t lane = threadIdx.x & (WAVE - 1u);
  const uint32_t wave = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
  for (uint32_t g = blockIdx.x * nw + wave; g < a.n; g += gridDim.x * nw) {
    const uint64_t hb = a.offsets[g], he = a.offsets[g + 1u];
    float sum = 0.0f;
    uint32_t perfect = 0u, cur_d = 0xFFFFFFFFu;
    unsigned long long raw = 0, kept = 0; /* hits / hits that counted so far at distance cur_d */
    const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull;
    /* the next block's loads are in flight while this block's additions run (a guide with 4 x 10^5 hits is
     * 6,900 blocks on one wavefront: their load latency, not the additions, set its time) */
    float c_next = 0.0f;
    uint32_t inf_next = 8u;
    if (hb + lane < he) {
      c_next = cf[hb + lane];
      inf_next = info[hb + lane];
    }
    for (uint64_t h0 = hb; h0 < he; h0 += WAVE) {
      const uint64_t h = h0 + lane;
      const bool valid = h < he;
      const float c = c_next;
      const uint32_t inf = inf_next;
      c_next = 0.0f;
      inf_next = 8u;
      if (h + WAVE < he) {
        c_next = cf[h + WAVE];
        inf_next = info[h + WAVE];
      }
      uint32_t d = 8u, sent = 0u, pgg = 0u;
      if (valid) {
        d = inf & 7u;
        sent = (inf >> 3) & 1u;
        pgg = (inf >> 4) & 1u;
      }
      /* --max-off-targets: a hit is passed over when `max_off` hits of its distance came before it - CSV:
       * counted on the raw index (printer.hpp:259); SAM: on the hits that counted (:129), i.e. the ones not
       * dropped at a chromosome boundary (while below the bound every such hit counts).  Hits are ordered by
       * distance, so both counts are prefix counts inside the distance class: per lane from ballots, across
       * the 64-hit blocks through (cur_d, raw, kept). */
      bool skip = false;
      if (a.max_off != -1) {
        uint64_t same = 0, same_ok = 0;
        for (uint32_t dv = 0; dv < 8u; ++dv) {
          const uint64_t b1 = __ballot(valid && d == dv), b2 = __ballot(valid && d == dv && !sent);
          if (d == dv) {
            same = b1;
            same_ok = b2;
          }
        }
        const unsigned long long before = (a.sam ? (unsigned long long)__popcll(same_ok & below) : (unsigned long long)__popcll(same & below)) +
                                          (d == cur_d ? (a.sam ? kept : raw) : 0ull);
        skip = valid && before >= (unsigned long long)a.max_off;
        /* carry: the class of the block's last hit */
        const uint32_t nv = (uint32_t)__popcll(__ballot(valid));
        const uint32_t d_last = (uint32_t)__shfl((int)d, (int)(nv - 1u));
        const unsigned long long n_last = __popcll(__ballot(valid && d == d_last)),
                                 ok_last = __popcll(__ballot(valid && d == d_last && !sent));
        if (d_last == cur_d) {
          raw += n_last;
          kept += ok_last;
        } else {
          cur_d = d_last;
          raw = n_last;
          kept = ok_last;
        }
      }
      if (__ballot(valid && !skip && pgg != 0u)) perfect = 1u;
      /* the sum itself runs hit by hit (float addition is not associative and the reference adds in
       * order): hits that do not count add +0, which leaves every partial sum as it is */
      const uint32_t abits = __float_as_uint((valid && !skip && !sent) ? c : 0.0f);
#pragma unroll
      for (int i = 0; i < WAVE; ++i) sum += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)abits, i));
    }
    if (!perfect) sum += 1.0f;
    float sp = 0.0f;
    if (sum > 0.0f) sp = __fdiv_rn(1.0f, sum);
    a.spec[g] = sp; /* all lanes, same address (no lane-conditional tail in the guide loop) */
  }
}

extern "C" gs_status gs_score_device(gs_index *ix, const void *d_guides, uint64_t n, uint32_t L, uint32_t P,
                                     uint32_t flags, int64_t max_off_targets, const gs_genome_structure *gs,
                                     const void *d_offsets, const void *d_hits, void *stream, void *d_cfd,
                                     void *d_specificity) {
  GS_HANDLE_LOCK(ix);
  if (!ix || !gs || (n && (!d_guides || !d_offsets || !d_specificity)) || (gs->n_chr && !gs->chr_lengths))
    return GS_ERR_ARG;
  if (n >= (1ull << 31) || max_off_targets < -1) return GS_ERR_ARG;
  if (L < 1 || L > 31 || P > 8 || 2 * L + 3 * P > 59) {
    gs_set_error("device path supports 1<=L<=31, P<=8, 2L+3P<=59");
    return GS_ERR_UNSUPPORTED;
  }
  if (n == 0) return GS_OK;
  hipStream_t st = (hipStream_t)stream;
  GS_HIP(hipSetDevice(ix->device));
  gs_status rc;
  const size_t cum_bytes = 8 * ((size_t)gs->n_chr + 1);
  if ((rc = gs_reserve(ix->w_score, 336 * sizeof(double) + cum_bytes)) != GS_OK) return rc;
  std::vector<uint64_t> host(336 + (size_t)gs->n_chr + 1);
  memcpy(host.data(), gs_cfd_mm, 320 * sizeof(double));
  memcpy(host.data() + 320, gs_cfd_pam, 16 * sizeof(double));
  uint64_t acc = 0;
  host[336] = 0;
  for (uint32_t i = 0; i < gs->n_chr; i++) {
    acc += gs->chr_lengths[i];
    host[337 + i] = acc;
  }
  GS_HIP(hipMemcpyAsync(ix->w_score.p, host.data(), 8 * host.size(), hipMemcpyHostToDevice, st));
  GS_HIP(hipStreamSynchronize(st)); /* `host` is a local */
  gs_score_args a;
  a.guides = (const uint8_t *)d_guides;
  a.offsets = (const uint64_t *)d_offsets;
  a.hits = (const gs_hit *)d_hits;
  a.tab = (const double *)ix->w_score.p;
  a.chr_cum = (const uint64_t *)ix->w_score.p + 336;
  a.cfd = (float *)d_cfd;
  a.spec = (float *)d_specificity;
  a.max_off = (long long)max_off_targets;
  a.n = (uint32_t)n;
  a.L = L;
  a.P = P;
  a.n_chr = gs->n_chr;
  a.start = (flags & GS_FLAG_PAM_AT_START) ? 1u : 0u;
  a.sam = (flags & GS_TEXT_SAM) ? 1u : 0u;
  hipDeviceProp_t prop;
  int cus = 256;
  if (hipGetDeviceProperties(&prop, ix->device) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  /* hits of the batch (the last offset), their CFDs (the caller's array or one of the handle's) and facts */
  uint64_t n_hits = 0;
  GS_HIP(hipMemcpyAsync(&n_hits, (const uint64_t *)d_offsets + n, 8, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  if ((rc = gs_reserve(ix->w_score_tmp, (d_cfd ? 0 : 4 * n_hits) + n_hits + 64)) != GS_OK) return rc;
  float *cf = d_cfd ? (float *)d_cfd : (float *)ix->w_score_tmp.p;
  uint8_t *info = (uint8_t *)ix->w_score_tmp.p + (d_cfd ? 0 : 4 * n_hits);
  if (n_hits) {
    uint64_t gh = (n_hits + 255) / 256;
    if (gh > (uint64_t)cus * 32u) gh = (uint64_t)cus * 32u;
    hipLaunchKernelGGL(k_score_hits, dim3((uint32_t)gh), dim3(256), 0, st, a, n_hits, cf, info);
  }
  uint32_t grid = (uint32_t)((n + SCORE_WAVES - 1) / SCORE_WAVES);
  if (grid > (uint32_t)cus * 16u) grid = (uint32_t)cus * 16u;
  hipLaunchKernelGGL(k_score_sum, dim3(grid), dim3(WAVE * SCORE_WAVES), 0, st, a, (const float *)cf, (const uint8_t *)info);
  GS_HIP(hipStreamSynchronize(st));
  GS_HIP(hipGetLastError());
  return GS_OK;
}

extern "C" gs_status gs_score(gs_index *ix, const char *guides, uint64_t n, uint32_t L, uint32_t P,
                              uint32_t flags, int64_t max_off_targets, const gs_genome_structure *gs,
                              const uint64_t *offsets, const gs_hit *hits, float *cfd, float *specificity) {
  GS_HANDLE_LOCK(ix);
  if (!ix || !gs || (n && (!guides || !offsets || !specificity))) return GS_ERR_ARG;
  if (n == 0) return GS_OK;
  const uint64_t nh = offsets[n];
  if (nh && !hits) return GS_ERR_ARG;
  GS_HIP(hipSetDevice(ix->device));
  const size_t b_g = ((size_t)n * L + 15) & ~(size_t)15, b_o = (8 * ((size_t)n + 1) + 15) & ~(size_t)15,
               b_h = sizeof(gs_hit) * (size_t)nh, b_c = (4 * (size_t)nh + 15) & ~(size_t)15, b_s = 4 * (size_t)n;
  gs_status rc = gs_reserve(ix->w_score_io, b_g + b_o + b_h + b_c + b_s + 64);
  if (rc != GS_OK) return rc;
  char *p = (char *)ix->w_score_io.p;
  char *d_g = p, *d_o = d_g + b_g, *d_h = d_o + b_o, *d_c = d_h + b_h, *d_s = d_c + b_c;
  GS_HIP(hipMemcpy(d_g, guides, (size_t)n * L, hipMemcpyHostToDevice));
  GS_HIP(hipMemcpy(d_o, offsets, 8 * ((size_t)n + 1), hipMemcpyHostToDevice));
  if (nh) GS_HIP(hipMemcpy(d_h, hits, b_h, hipMemcpyHostToDevice));
  rc = gs_score_device(ix, d_g, n, L, P, flags, max_off_targets, gs, d_o, d_h, nullptr, cfd ? d_c : nullptr,
                       d_s);
  if (rc != GS_OK) return rc;
  if (cfd && nh) GS_HIP(hipMemcpy(cfd, d_c, 4 * (size_t)nh, hipMemcpyDeviceToHost));
  GS_HIP(hipMemcpy(specificity, d_s, b_s, hipMemcpyDeviceToHost));
  return GS_OK;
}
