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

#include <algorithm>

#define WAVE 64
#define SCORE_WAVES 4
int gs_num_cus(int device); /* gs_enumerate.hip: asked of the driver once per device */

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
  /* k_score_sum's work counter hands out tickets: the first p0 tickets are one guide each (the order is heaviest first),
   * every later one `take` guides - a counter bumped once per guide serves ~88 waves per microsecond chip-wide and held a
   * batch of 10^6 light guides at 11 ms whatever their hits (19 ms per 1 M guides with 2 x 10^6 hits; now 2 ms) */
  uint32_t p0, take;
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

/* Three steps.  k_score_hits: one thread per hit - its CFD (printer.hpp:98-113) and the three facts the aggregation
 * needs (distance, dropped at a chromosome boundary, perfect xGG hit); every hit is independent, so guides with 10^5
 * hits spread over the whole chip.  k_score_classes / k_score_place: the guides in order of falling hit count (by the
 * count's binary logarithm), so that the longest chains start first.  k_score_sum: one wavefront per guide adds its
 * scores in order.  (One LANE per guide - 64 chains per wave, each lane streaming its own hits - was built and is
 * 14 x slower: a lane's next loads wait a memory latency per four hits, 220 ms for the heaviest guide alone.) */
#define SC_CHUNK 4096u  /* hits a wave of k_score_hits takes at a time: one search for its first guide, then a walk */
#define SC_MAXCHR 2048u /* chromosomes whose prefix sums the block keeps in LDS (more: read from memory) */
#define SC_BINS 4096u
struct gs_score_geo {
  const uint16_t *bin_chr; /* [SC_BINS] chromosome that holds position bin << bin_shift | 0x8000 when the bin reaches into the next one */
  uint32_t bin_shift;
};
/* src/genomics/structures.cxx:7-52 with the chromosome found through a table of 4,096 bins (one LDS read; a bin with a
 * chromosome boundary inside walks on from there) instead of a binary search per hit */
__device__ __forceinline__ bool sc_sentinel_lds(const uint64_t *s_cum, const uint16_t *s_bin, const uint32_t bin_shift, const uint32_t n_chr,
                                                const long long pos, const uint32_t L, const uint32_t P) {
  const bool minus = pos < 0;
  const unsigned long long ab = (unsigned long long)(minus ? -pos : pos);
  if (n_chr == 0u || ab >= s_cum[n_chr]) return true;
  const uint32_t e = s_bin[ab >> bin_shift];
  uint32_t c = e & 0x7FFFu;
  if (e & 0x8000u)
    while (s_cum[c + 1u] <= ab) c++;
  const long long off = (long long)(ab - s_cum[c]);
  const long long len = (long long)(s_cum[c + 1u] - s_cum[c]);
  long long s1, e1;
  if (!minus) {
    e1 = off + 1;
    s1 = e1 - (long long)L - (long long)P + 1;
  } else {
    s1 = off + 1;
    e1 = s1 + (long long)L + (long long)P - 1;
  }
  return s1 < 0 || e1 > len; /* :46-48 */
}
/* cf: every hit's CFD (the caller's array; or nullptr); info: the fact bytes (--max-off-targets: k_score_sum_maxoff reads
 * them) or nullptr; cfm: the score as it enters the guide's sum (+0 for a hit dropped at a chromosome boundary) and perfect[g]:
 * the guide has a perfect xGG hit - what k_score_sum reads */
__global__ __launch_bounds__(256) void k_score_hits(gs_score_args a, gs_score_geo geo, uint64_t n_hits, float *cf, uint8_t *info, float *cfm,
                                                    uint32_t *perfect) {
  __shared__ double s_tab[336];
  __shared__ uint64_t s_cum[SC_MAXCHR + 1u];
  __shared__ uint16_t s_bin[SC_BINS];
  const bool geo_lds = a.n_chr <= SC_MAXCHR;
  for (uint32_t i = threadIdx.x; i < 336u; i += blockDim.x) s_tab[i] = a.tab[i];
  if (geo_lds) {
    for (uint32_t i = threadIdx.x; i <= a.n_chr; i += blockDim.x) s_cum[i] = a.chr_cum[i];
    for (uint32_t i = threadIdx.x; i < SC_BINS; i += blockDim.x) s_bin[i] = geo.bin_chr[i];
  }
  __syncthreads();
  const uint32_t L = a.L, P = a.P, slen = L + P, lane = threadIdx.x & (WAVE - 1u);
  /* pam = match_sequence.substr(20, 3) when the sequence has at least 20 symbols */
  const uint32_t pam_len = slen < 20u ? 0u : (slen - 20u < 3u ? slen - 20u : 3u);
  const bool scored = L == 20u && pam_len == 3u; /* printer.hpp:99 */
  const uint64_t n_chunks = (n_hits + SC_CHUNK - 1u) / SC_CHUNK;
  const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE, n_waves = (uint64_t)gridDim.x * blockDim.x / WAVE;
  for (uint64_t ch = wave0; ch < n_chunks; ch += n_waves) {
    /* the guide of the chunk's first hit: last g with offsets[g] <= h (wave-uniform: scalar loads), then the lanes walk */
    const uint64_t h_first = ch * SC_CHUNK;
    uint32_t lo = 0, hi = a.n;
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (a.offsets[mid] <= h_first)
        lo = mid;
      else
        hi = mid;
    }
    uint32_t g = lo;
    const uint64_t h_end = h_first + SC_CHUNK < n_hits ? h_first + SC_CHUNK : n_hits;
    for (uint64_t h = h_first + lane; h < h_end; h += WAVE) {
      while (g + 1u < a.n && a.offsets[g + 1u] <= h) g++;
      const uint8_t *gd = a.guides + (size_t)g * L;
      const gs_hit hit = a.hits[h];
      const uint64_t path = (hit.key >> 1) & ((1ull << 59) - 1ull); /* key bits 59:1: position 0 at the top */
      const uint32_t d = (uint32_t)(hit.key >> 61);
      float c = 1.0f;
      uint32_t pgg = 0u;
      bool lw;
      if (scored && !a.start) {
        /* The guide searched as given: a position enters the product exactly where the hit has a substitution (its
         * two-bit field of the path is not zero: lower case in match.sequence, index.hpp:243), in position order, each
         * product rounded to float before the next (printer.hpp:104-109).  On base indices (A,C,G,T = 0..3) instead of
         * characters: the guide's base r = sequence[i]; the text's base under it is the (code - 1)-th of the three others
         * of its complement, and toupper(complement(match_sequence[i])) is that base itself: d. */
        const uint64_t f = path >> 19; /* position i at bits 39 - 2 i : 38 - 2 i */
        uint64_t nz = (f | (f >> 1)) & 0x5555555555ull;
        while (nz) {
          const uint32_t hb = 63u - (uint32_t)__builtin_clzll(nz);
          nz &= ~(1ull << hb);
          const uint32_t i = (38u - hb) >> 1;
          const int r = sc_bidx(gd[i]);
          const int q = 3 - r; /* the query base: the guide's complement (process.hpp:63) */
          int dd = (int)((f >> hb) & 3ull) - 1;
          if (dd >= q) dd++;
          const double sc = r >= 0 ? s_tab[(r * 4 + dd) * 20 + (int)i] : 0.0;
          c = (float)((double)c * sc);
        }
        /* PAM symbols 1 and 2 of match.sequence (codes A,C,G,N,T = 0..4), complemented: the score's d-side bases */
        const uint32_t c1 = (uint32_t)(path >> (56u - 2u * L - 3u)) & 7u, c2 = (uint32_t)(path >> (56u - 2u * L - 6u)) & 7u;
        const int b1 = c1 == 3u ? -1 : c1 >= 4u ? 0 : 3 - (int)c1, b2 = c2 == 3u ? -1 : c2 >= 4u ? 0 : 3 - (int)c2;
        const double ps = (b1 >= 0 && b2 >= 0) ? s_tab[320 + b1 * 4 + b2] : 0.0;
        c = (float)((double)c * ps);
        pgg = (d == 0u && c1 == 1u && c2 == 1u) ? 1u : 0u; /* perfect xGG hit (printer.hpp:145-146 / :262): match_sequence ends in GG */
      } else {
      if (scored) {
        /* --start: the two strings are compared position by position as the reference compares them, case-sensitively */
        uint64_t todo = 0xFFFFFull;
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
      }
      const bool sent = geo_lds ? sc_sentinel_lds(s_cum, s_bin, geo.bin_shift, a.n_chr, (long long)hit.pos, L, P)
                                : sc_sentinel(a.chr_cum, a.n_chr, (long long)hit.pos, L, P);
      if (cf != nullptr) cf[h] = c;
      if (info != nullptr) info[h] = (uint8_t)(d | ((sent ? 1u : 0u) << 3) | (pgg << 4));
      if (cfm != nullptr) {
        cfm[h] = sent ? 0.0f : c;
        if (pgg) perfect[g] = 1u; /* (every writer stores the same word: a repeat family's guide has thousands of perfect hits - an atomic per hit made this kernel 12 ms instead of 5) */
      }
    }
  }
}

/* guides by the binary logarithm of their hit count: cls[c] = guides of class c (hits in [2^c, 2^(c+1)), class 0 also
 * the guides without hits), then each guide's place in the order of falling classes */
__global__ __launch_bounds__(256) void k_score_classes(const uint64_t *offsets, uint32_t n, uint32_t *cls) {
  __shared__ uint32_t s_c[40];
  if (threadIdx.x < 40u) s_c[threadIdx.x] = 0u;
  __syncthreads();
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
    const uint64_t c = offsets[g + 1u] - offsets[g];
    atomicAdd(&s_c[63u - (uint32_t)__builtin_clzll(c | 1ull)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 40u && s_c[threadIdx.x]) atomicAdd(&cls[threadIdx.x], s_c[threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_score_place(const uint64_t *offsets, uint32_t n, const uint32_t *cls, uint32_t *cursor, uint32_t *order) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = g < n;
  uint32_t k = 0;
  if (on) {
    const uint64_t c = offsets[g + 1u] - offsets[g];
    k = 63u - (uint32_t)__builtin_clzll(c | 1ull);
  }
  /* one atomic per wave and class: a batch of 10^6 guides with a handful of hits each is three or four classes - an atomic
   * per guide on those few words took 6 of the 9 ms the scoring of such a batch took */
  const uint32_t lane = threadIdx.x & (WAVE - 1u);
  uint32_t pos = 0;
  uint64_t todo = __ballot(on);
  while (todo) {
    const int l = __ffsll((long long)todo) - 1;
    const uint32_t kk = (uint32_t)__shfl((int)k, l);
    const uint64_t same = __ballot(on && k == kk);
    uint32_t b0 = 0;
    if ((int)lane == l) b0 = atomicAdd(&cursor[kk], (uint32_t)__popcll(same));
    b0 = (uint32_t)__shfl((int)b0, l);
    if (on && k == kk) pos = b0 + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  if (!on) return;
  uint32_t base = 0;
  for (uint32_t j = 39u; j > k; --j) base += cls[j]; /* the classes of more hits go first */
  order[base + pos] = g;
}
/* --max-off-targets given: one wavefront per guide, 64 hits per round, the per-distance counters from ballots and the
 * additions as a chain of v_readlane / v_add pairs (the form every batch took until round 5) */
__global__ __launch_bounds__(WAVE *SCORE_WAVES) void k_score_sum_maxoff(gs_score_args a, const float *cf, const uint8_t *info) {
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

/* The sum of a guide's CFDs, hit by hit in order (float addition is not associative and the reference adds sequentially:
 * printer.hpp:251-297 / :115-170).  One wavefront per guide at a time, guides in order of falling hit count from a
 * counter.  The scores arrive as they enter the sum (k_score_hits wrote +0 for what does not count: it leaves every
 * partial sum as it is): 512 per round, eight neighbouring ones per lane, written to LDS; then the chain runs from LDS,
 * four scores per broadcast read, on the first 16 lanes only (a broadcast to all 64 moves 256 bytes of LDS traffic per
 * hit): 5 instructions per four hits where the v_readlane / v_add pairs of the first form took 8, and the loads of the next
 * TWO rounds are in flight meanwhile.  What bounds the kernel is the heaviest guide's own chain (4.4 x 10^5 dependent
 * additions on the repeat-rich batch).  Tried and dropped: one LANE per guide, each lane streaming its own hits (a lane's
 * next loads wait a memory latency per four hits: 220 ms for the heaviest guide alone); the scores through the scalar
 * cache, eight per s_load_dwordx8, one v_add_f32 with a scalar operand per hit (a wave keeps 256 bytes in flight where a
 * memory latency needs 4 KB: 6.6 ms against 3.9). */
#define SC_ROUND 512u
#define SC_PER 8u /* hits per lane and round */
typedef float sc_f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(WAVE *SCORE_WAVES) void k_score_sum(gs_score_args a, const float *cfm, const uint32_t *perfect, const uint32_t *order,
                                                                 uint32_t *next) {
  __shared__ float s_buf[SCORE_WAVES][SC_ROUND];
  const uint32_t lane = threadIdx.x & (WAVE - 1u);
  float *buf = s_buf[threadIdx.x / WAVE];
  for (;;) {
    uint32_t tk = 0;
    if (lane == 0) tk = atomicAdd(next, 1u);
    tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    /* (64-bit: tickets far beyond the last guide - every wave draws one more at the end - must not wrap) */
    const uint64_t tb = tk < a.p0 ? (uint64_t)tk : (uint64_t)a.p0 + (uint64_t)(tk - a.p0) * a.take;
    if (tb >= a.n) break;
    const uint32_t t_lo = (uint32_t)tb;
    const uint64_t te = tb + (tk < a.p0 ? 1u : a.take);
    const uint32_t t_hi = te < a.n ? (uint32_t)te : a.n;
    for (uint32_t t = t_lo; t < t_hi; ++t) {
    const uint32_t g = order[t];
    const uint64_t hb = a.offsets[g], len = a.offsets[g + 1u] - hb;
    const float *pc = cfm + hb + SC_PER * lane;
    float sum = 0.0f;
    /* (no load is guarded - a test per load made the compiler wait for each before the next, eight memory latencies per
     * round: what lies beyond the guide's last hit, the next guide's scores or the 8 KB behind the array, is replaced by
     * +0 when the round is written to LDS) */
    sc_f4u cA0 = *(const sc_f4u *)pc, cA1 = *(const sc_f4u *)(pc + 4), cB0 = *(const sc_f4u *)(pc + SC_ROUND), cB1 = *(const sc_f4u *)(pc + SC_ROUND + 4);
    for (uint64_t base = 0; base < len; base += SC_ROUND) {
      float c[SC_PER];
      const sc_f4u n0 = *(const sc_f4u *)(pc + base + 2u * SC_ROUND), n1 = *(const sc_f4u *)(pc + base + 2u * SC_ROUND + 4);
      const uint64_t q0 = base + SC_PER * lane;
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) {
        c[u] = q0 + u < len ? cA0[u] : 0.0f;
        c[4u + u] = q0 + 4u + u < len ? cA1[u] : 0.0f;
      }
      cA0 = cB0;
      cA1 = cB1;
      cB0 = n0;
      cB1 = n1;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* (the chain of the round before has read the buffer) */
      *(float4 *)(buf + SC_PER * lane) = make_float4(c[0], c[1], c[2], c[3]);
      *(float4 *)(buf + SC_PER * lane + 4u) = make_float4(c[4], c[5], c[6], c[7]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      /* the chain: eight broadcast reads (32 scores) are on their way while the 32 before them are added - a read that
       * waited for the additions before it cost 35 cycles per hit.  Places beyond the guide's last hit hold +0 (every
       * lane wrote its eight): rounds of 32 need no test. */
      const uint32_t left = len - base < SC_ROUND ? (uint32_t)(len - base) : SC_ROUND;
      if (lane < 16u) {
        float4 va[8], vb[8];
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) va[q] = *(const float4 *)(buf + 4u * q);
        for (uint32_t j = 0; j < left; j += 64u) {
#pragma unroll
          for (uint32_t q = 0; q < 8u; ++q) vb[q] = *(const float4 *)(buf + ((j + 32u + 4u * q) & (SC_ROUND - 1u)));
#pragma unroll
          for (uint32_t q = 0; q < 8u; ++q) {
            sum += va[q].x;
            sum += va[q].y;
            sum += va[q].z;
            sum += va[q].w;
          }
          if (j + 32u >= left) break;
#pragma unroll
          for (uint32_t q = 0; q < 8u; ++q) va[q] = *(const float4 *)(buf + ((j + 64u + 4u * q) & (SC_ROUND - 1u)));
#pragma unroll
          for (uint32_t q = 0; q < 8u; ++q) {
            sum += vb[q].x;
            sum += vb[q].y;
            sum += vb[q].z;
            sum += vb[q].w;
          }
        }
      }
    }
    sum = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sum)));
    if (!perfect[g]) sum += 1.0f;
    float sp = 0.0f;
    if (sum > 0.0f) sp = __fdiv_rn(1.0f, sum);
    a.spec[g] = sp; /* all lanes, same address */
    }
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
  const size_t bins_at = 336 * sizeof(double) + cum_bytes; /* behind the prefix sums: the bin table (uint16 x SC_BINS, as 1,024 uint64), 40 + 40 class words, the guide counter, the order */
  if ((rc = gs_reserve(ix->w_score, bins_at + 2 * SC_BINS + 4 * 96 + 4 * ((size_t)n + 1))) != GS_OK) return rc;
  std::vector<uint64_t> host(336 + (size_t)gs->n_chr + 1 + SC_BINS / 4);
  memcpy(host.data(), gs_cfd_mm, 320 * sizeof(double));
  memcpy(host.data() + 320, gs_cfd_pam, 16 * sizeof(double));
  uint64_t acc = 0;
  host[336] = 0;
  for (uint32_t i = 0; i < gs->n_chr; i++) {
    acc += gs->chr_lengths[i];
    host[337 + i] = acc;
  }
  /* the chromosome of every 4,096th-of-the-genome bin's first position; 0x8000: the bin reaches into the next chromosome */
  uint32_t bin_shift = 0;
  while ((acc >> bin_shift) >= SC_BINS) bin_shift++;
  {
    uint16_t *bins = (uint16_t *)(host.data() + 337 + gs->n_chr);
    uint32_t c = 0;
    for (uint32_t b = 0; b < SC_BINS; b++) {
      const uint64_t lo = (uint64_t)b << bin_shift, hi = (((uint64_t)b + 1) << bin_shift) - 1;
      while (c + 1 < gs->n_chr && host[337 + c] <= lo) c++;
      uint16_t e = (uint16_t)(c < 0x7FFFu ? c : 0x7FFFu);
      if (gs->n_chr && hi >= host[337 + c]) e |= 0x8000u;
      bins[b] = e;
    }
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
  const int cus = gs_num_cus(ix->device);
  /* hits of the batch (the last offset), their CFDs (the caller's array or one of the handle's) and facts */
  uint64_t n_hits = 0;
  GS_HIP(hipMemcpyAsync(&n_hits, (const uint64_t *)d_offsets + n, 8, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  /* --max-off-targets: every CFD (the caller's array or one of the handle's) and the fact bytes; else the scores as they
   * enter the sums (+ 8 KB: the last guide's unguarded loads reach up to three rounds beyond them) and a word per guide */
  const bool with_max = max_off_targets != -1;
  if ((rc = gs_reserve(ix->w_score_tmp, with_max ? (d_cfd ? 0 : 4 * n_hits) + n_hits + 64 : 4 * n_hits + 8192 + 128 + 4 * (n + 1))) != GS_OK) return rc;
  float *cf = d_cfd ? (float *)d_cfd : with_max ? (float *)ix->w_score_tmp.p : nullptr;
  uint8_t *info = with_max ? (uint8_t *)ix->w_score_tmp.p + (d_cfd ? 0 : 4 * n_hits) : nullptr;
  float *cfm = with_max ? nullptr : (float *)ix->w_score_tmp.p;
  uint32_t *d_perfect = with_max ? nullptr : (uint32_t *)((char *)ix->w_score_tmp.p + ((4 * n_hits + 8192 + 63) & ~(size_t)63));
  if (!with_max) GS_HIP(hipMemsetAsync(d_perfect, 0, 4 * n, st));
  gs_score_geo geo;
  geo.bin_chr = (const uint16_t *)((const char *)ix->w_score.p + bins_at);
  geo.bin_shift = bin_shift;
  uint32_t *d_cls = (uint32_t *)((char *)ix->w_score.p + bins_at + 2 * SC_BINS), *d_cursor = d_cls + 40, *d_next = d_cls + 80, *d_order = d_cls + 96;
  GS_HIP(hipMemsetAsync(d_cls, 0, 4 * 96, st));
  if (n_hits) {
    uint64_t gh = ((n_hits + SC_CHUNK - 1) / SC_CHUNK + 3) / 4; /* four waves per workgroup, a chunk per wave and visit */
    if (gh > (uint64_t)cus * 8u) gh = (uint64_t)cus * 8u;
    hipLaunchKernelGGL(k_score_hits, dim3((uint32_t)gh), dim3(256), 0, st, a, geo, n_hits, cf, info, cfm, d_perfect);
  }
  const uint32_t n32 = (uint32_t)n;
  hipLaunchKernelGGL(k_score_classes, dim3(std::min<uint32_t>((n32 + 255) / 256, (uint32_t)cus * 4u)), dim3(256), 0, st,
                     (const uint64_t *)d_offsets, n32, d_cls);
  hipLaunchKernelGGL(k_score_place, dim3((n32 + 255) / 256), dim3(256), 0, st, (const uint64_t *)d_offsets, n32, (const uint32_t *)d_cls,
                     d_cursor, d_order);
  if (a.max_off != -1) {
    uint32_t grid = (uint32_t)((n + SCORE_WAVES - 1) / SCORE_WAVES);
    if (grid > (uint32_t)cus * 16u) grid = (uint32_t)cus * 16u;
    hipLaunchKernelGGL(k_score_sum_maxoff, dim3(grid), dim3(WAVE * SCORE_WAVES), 0, st, a, (const float *)cf, (const uint8_t *)info);
  } else {
    uint32_t grid = (uint32_t)((n + SCORE_WAVES - 1) / SCORE_WAVES);
    if (grid > (uint32_t)cus * 8u) grid = (uint32_t)cus * 8u;
    {
      const uint64_t waves = (uint64_t)grid * SCORE_WAVES;
      a.p0 = (uint32_t)std::min<uint64_t>(n, 4u * waves); /* the heaviest guides one at a time: they set the balance */
      const uint64_t rest = n - a.p0;
      a.take = (uint32_t)std::min<uint64_t>(64u, std::max<uint64_t>(1u, rest / (waves * 8u)));
    }
    hipLaunchKernelGGL(k_score_sum, dim3(grid), dim3(WAVE * SCORE_WAVES), 0, st, a, (const float *)cfm, (const uint32_t *)d_perfect,
                       (const uint32_t *)d_order, d_next);
  }
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
