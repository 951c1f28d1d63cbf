/*
 * gs_tileorder.hip -- canonical order of guides with more matches than k_order holds in LDS, per guide.
 *
 * What orders a guide's hits is (mismatches, index, match.sequence, row) (process.hpp:100-115,
 * structures.hpp:33-43).  An item of the search is one (guide, index): inside it the order is
 * (mismatches, sequence, row), i.e. one 64-bit word K = (class base + lexicographic rank of the sequence) << 32 | row
 * (the rank of gs_search.hip's big2_rank without the index bit: 32 bits at L = 20, P = 3, m <= 6).
 * Instead of eight device-wide radix passes over (guide | rank | row bits) and five more passes to gather, flag,
 * scan and locate, every item is ordered by itself:
 *
 *   k_to_plan       one workgroup: per set item the tiles, bucket space and arena chunks it needs; four scans
 *   k_to_fill       tile descriptors, the chunk index of every item, per guide where each (mismatches, index)
 *                   class starts in its hit list (from k_search's per-class counts), hits per guide
 *   k_to_partition  items beyond one tile: a sample of their records ordered in LDS gives splitters; one
 *                   streaming pass deals the records into buckets of at most TO_TILE records (sample sort:
 *                   the full word decides, so a run of 10^4 equal sequences is split by row)
 *   k_to_sort       one workgroup per tile (a small item where k_search left it, or a bucket): K of every
 *                   record, a merge sort in LDS (eight words per thread ordered in registers, then merge-path
 *                   rounds), then the hits - suffix array gather + coordinate rule (process.hpp:104, 111) -
 *                   straight to their final place; 512 threads for tiles of up to 4,096 records, 128 for
 *                   those of up to 1,024
 *
 * Two passes of HBM traffic for partitioned items (16 B read + 16 B written, twice), one for the rest.
 * The path writes final hits on two assumptions it checks as it goes: every record is a single row, and no
 * (sequence, row) occurs twice in an item (overlapping PAM patterns).  A tile that sees otherwise - or a bucket
 * that outgrows its space - raises a flag; the caller then orders the batch with the device-wide form.
 */
#include "gs_device.h"

#include <algorithm>

#define TO_NT 512u                 /* threads of the sort and partition workgroups */
#define TO_NW (TO_NT / WAVE)       /* 8 waves */
#define TO_KPT 8u                  /* keys per thread */
#define TO_TILE (TO_NT * TO_KPT)   /* 4,096 records per tile */
#define TO_NBMAX 1024u             /* buckets of one item */
#define TO_DIRECT 0x80000000u

/* flags raised by the kernels (gs_tileorder_run) */
#define TO_F_MULTIROW 1u
#define TO_F_DUP 2u
#define TO_F_BUCKET 4u
#define TO_F_BIG 8u
#define TO_F_CLS 16u

struct gs_to_tab {
  unsigned long long n[32][8]; /* n[a][r] = C(a, r) 3^r */
  unsigned long long base[8];  /* sequences of the classes with fewer mismatches (one index) */
  unsigned long long pam_mul;
};

/* rank of a key's match.sequence among the sequences of its mismatch class: position 0 most significant, as
 * the key's own bits order them (the same number gs_search.hip's big2_rank computes) */
__device__ __forceinline__ unsigned long long to_rank(const unsigned long long key, const uint32_t L, const uint32_t P,
                                                       const unsigned long long *nt, const unsigned long long pam_mul) {
  const unsigned long long path = (key >> 1) & ((1ull << 59) - 1ull); /* key bits 59:1 */
  /* the guide's L two-bit fields, position 0 in the highest: only the substituted ones (at most seven) count */
  const unsigned long long gb = path >> (59u - 2u * L);
  unsigned long long nz = (gb | (gb >> 1)) & 0x5555555555555555ull & ((1ull << (2u * L)) - 1ull);
  uint32_t r = (uint32_t)__popcll(nz);
  if (r > 7u) r = 7u;
  unsigned long long rank = 0;
  while (nz != 0ull && r != 0u) {
    const uint32_t hb = 63u - (uint32_t)__builtin_clzll(nz); /* = 2 x the positions behind this one */
    const uint32_t c = (uint32_t)(gb >> hb) & 3u, a = hb >> 1;
    /* smaller sequences with the same prefix: a 0 here (r substitutions behind), or one of the c-1 lower codes */
    rank += nt[a * 8u + r] + (unsigned long long)(c - 1u) * nt[a * 8u + r - 1u];
    r--;
    nz &= ~(1ull << hb);
  }
  unsigned long long pr = 0;
  for (uint32_t u = 0; u < P; u++) {
    const uint32_t c = (uint32_t)(path >> (56u - 2u * L - 3u * u)) & 7u;
    pr = pr * 5ull + (c < 4u ? c : 4u);
  }
  return rank * pam_mul + pr;
}
__device__ __forceinline__ unsigned long long to_word(const uint4 rec, const uint32_t L, const uint32_t P, const unsigned long long *nt,
                                                       const unsigned long long *bs, const unsigned long long pam_mul) {
  const unsigned long long key = ((unsigned long long)rec.y << 32) | rec.x;
  const unsigned long long w = bs[(uint32_t)(key >> 61)] + to_rank(key, L, P, nt, pam_mul);
  return (w << 32) | rec.z;
}

/* ---- merge sort of n <= TO_TILE 64-bit words in LDS by the whole workgroup --------------------------------------
 * keys (+ idx: a 16-bit payload that moves with its key), TO_LDS entries each: word e lives at TO_AT(e) - one pad
 * entry after every eight, so that threads reading their own eight consecutive words hit different banks (blocked
 * accesses at a 64-byte stride were a 32-way bank conflict: 65 % of the LDS cycles of the first version).
 * Thread t owns places 8t .. 8t+7: it orders its eight words in registers (Batcher's network, 19 exchanges), then
 * log2(n / 8) rounds merge neighbouring runs pairwise - the thread finds where its eight places of the merged run
 * begin in either input (merge path: a binary search along its diagonal) and merges eight words from there, from LDS
 * into registers; all write back after a barrier.  About 30 instructions per word and round whatever the words' bits - the radix passes this replaced (one ballot per key bit
 * to rank a wave's keys stably) cost 90 per word and pass, eight passes on a repeat-rich tile.
 * Places n .. 8 * ceil(n / 8) hold words of all ones. */
#define TO_AT(e) ((e) + ((e) >> 3))
#define TO_LDS (TO_TILE + TO_TILE / 8u)
#define TO_CE(i, j)                                  \
  if (k[i] > k[j]) {                                 \
    const unsigned long long tk = k[i];              \
    k[i] = k[j];                                     \
    k[j] = tk;                                       \
    if constexpr (WITH_IDX) {                        \
      const uint32_t tv = v[i];                      \
      v[i] = v[j];                                   \
      v[j] = tv;                                     \
    }                                                \
  }
template <bool WITH_IDX>
__device__ __forceinline__ void to_msort(unsigned long long *keys, uint16_t *idx, const uint32_t n) { /* (any workgroup size: thread t owns places 8t ..) */
  const uint32_t t = threadIdx.x;
  const uint32_t nthr = (n + TO_KPT - 1u) / TO_KPT, ntot = nthr * TO_KPT;
  const bool on = t < nthr;
  const uint32_t mine = t * (TO_KPT + 1u); /* = TO_AT(8 t) */
  unsigned long long k[TO_KPT];
  uint32_t v[TO_KPT];
#pragma unroll
  for (uint32_t j = 0; j < TO_KPT; ++j) {
    const uint32_t e = t * TO_KPT + j;
    k[j] = on && e < n ? keys[mine + j] : ~0ull;
    v[j] = 0u;
    if constexpr (WITH_IDX) v[j] = on && e < n ? idx[mine + j] : 0u;
  }
  TO_CE(0, 1) TO_CE(2, 3) TO_CE(4, 5) TO_CE(6, 7)
  TO_CE(0, 2) TO_CE(1, 3) TO_CE(4, 6) TO_CE(5, 7)
  TO_CE(1, 2) TO_CE(5, 6)
  TO_CE(0, 4) TO_CE(1, 5) TO_CE(2, 6) TO_CE(3, 7)
  TO_CE(2, 4) TO_CE(3, 5)
  TO_CE(1, 2) TO_CE(3, 4) TO_CE(5, 6)
  if (on) {
#pragma unroll
    for (uint32_t j = 0; j < TO_KPT; ++j) {
      keys[mine + j] = k[j];
      if constexpr (WITH_IDX) idx[mine + j] = (uint16_t)v[j];
    }
  }
  __syncthreads();
  for (uint32_t width = TO_KPT; width < ntot; width <<= 1) {
    if (on) {
      const uint32_t base = (t * TO_KPT) & ~(2u * width - 1u);
      const uint32_t d = t * TO_KPT - base; /* this thread's first place on the merged run */
      const uint32_t a0 = base, a1 = base + width < ntot ? base + width : ntot;
      const uint32_t b1 = base + 2u * width < ntot ? base + 2u * width : ntot;
      const uint32_t la = a1 - a0, lb = b1 - a1;
      uint32_t lo = d > lb ? d - lb : 0u, hi = d < la ? d : la;
      while (lo < hi) { /* how many of the first d merged words come from A */
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[TO_AT(a0 + mid)] <= keys[TO_AT(a1 + d - 1u - mid)])
          lo = mid + 1u;
        else
          hi = mid;
      }
      /* eight words of the merged run from there, one at a time (a version that loaded the eight words that follow
       * in either input at once and took the eight smallest through a bitonic network had no read waiting for a
       * compare, and 85 % more instructions: the kernel was instruction bound with it) */
      uint32_t i = lo, j = d - lo;
      uint32_t pa = TO_AT(a0 + i), pb = TO_AT(a1 + j);
      unsigned long long ka = i < la ? keys[pa] : ~0ull, kb = j < lb ? keys[pb] : ~0ull;
#pragma unroll
      for (uint32_t o = 0; o < TO_KPT; ++o) {
        const bool ta = j >= lb || (i < la && ka <= kb);
        k[o] = ta ? ka : kb;
        if constexpr (WITH_IDX) v[o] = idx[ta ? pa : pb];
        if (ta) {
          ++i;
          pa = TO_AT(a0 + i);
          ka = i < la ? keys[pa] : ~0ull;
        } else {
          ++j;
          pb = TO_AT(a1 + j);
          kb = j < lb ? keys[pb] : ~0ull;
        }
      }
    }
    __syncthreads();
    if (on) {
#pragma unroll
      for (uint32_t o = 0; o < TO_KPT; ++o) {
        keys[mine + o] = k[o];
        if constexpr (WITH_IDX) idx[mine + o] = (uint16_t)v[o];
      }
    }
    __syncthreads();
  }
}

/* ---- the plan ------------------------------------------------------------------------------------------- */
struct gs_to_plan_args {
  const uint32_t *counts; /* per item of the batch: records (exact) */
  const uint32_t *list;   /* set guide -> guide of the batch, or nullptr: the whole batch */
  uint32_t n_it, cap;
  uint32_t *tbase, *bbase, *cbase, *gbase; /* [n_it + 1] first tile / first bucket slot / first chunk-index entry / place on the list of partitioned items */
  uint32_t *flags;
};
/* buckets of an item of c > TO_TILE records and the records each is meant to hold: the fewer samples per
 * splitter a large item can afford (TO_TILE samples in all), the more room its buckets get */
__device__ __host__ __forceinline__ uint32_t to_buckets(const uint32_t c) {
  if (c <= TO_TILE) return 0u;
  const uint32_t target = c <= 64u * (TO_TILE / 2u) ? TO_TILE / 2u : c <= 128u * (TO_TILE / 3u) ? TO_TILE / 3u : c <= 256u * (TO_TILE / 4u) ? TO_TILE / 4u
                          : c <= 512u * (TO_TILE / 6u) ? TO_TILE / 6u : TO_TILE / 8u;
  return (c + target - 1u) / target;
}
__global__ __launch_bounds__(1024) void k_to_plan(gs_to_plan_args a) {
  __shared__ uint32_t s_w[4][16], s_carry[4];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u), w = tid / WAVE;
  if (tid < 4u) s_carry[tid] = 0u;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < a.n_it + 1u; i0 += 1024u) {
    const uint32_t sb = i0 + tid;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (sb < a.n_it) {
      const uint32_t item = a.list ? 2u * a.list[sb >> 1] + (sb & 1u) : sb;
      const uint32_t c = a.counts[item];
      uint32_t nb = to_buckets(c);
      if (nb > TO_NBMAX) { /* an item beyond half a million records: the device-wide form orders this batch */
        atomicOr(a.flags, TO_F_BIG);
        nb = 0u;
      }
      v[0] = c == 0u ? 0u : nb ? nb : 1u;
      v[1] = nb;
      v[2] = c > a.cap ? (c - a.cap + ARENA_CHUNK - 1u) / ARENA_CHUNK : 0u;
      v[3] = nb ? 1u : 0u;
    }
    uint32_t ex[4];
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) {
      const uint32_t incl = wave_incl_sum(v[q]);
      if (lane == WAVE - 1u) s_w[q][w] = incl;
      ex[q] = incl - v[q];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) {
      uint32_t b = s_carry[q];
      for (uint32_t u = 0; u < w; ++u) b += s_w[q][u];
      ex[q] += b;
    }
    if (sb <= a.n_it) {
      a.tbase[sb] = ex[0];
      a.bbase[sb] = ex[1];
      a.cbase[sb] = ex[2];
      a.gbase[sb] = ex[3];
    }
    __syncthreads();
    if (tid == 1023u) {
#pragma unroll
      for (uint32_t q = 0; q < 4u; ++q) s_carry[q] = ex[q] + v[q];
    }
    __syncthreads();
  }
}

/* ---- tile descriptors, chunk index, class starts ------------------------------------------------------------- */
struct gs_to_fill_args {
  const uint32_t *counts, *cls, *list;
  uint32_t n_it, cap;
  const uint32_t *tbase, *bbase, *gbase;
  uint4 *tiles;     /* {set item | TO_DIRECT, bucket slot, records, records of the item in the buckets before} */
  uint32_t *biglist; /* set items that are partitioned */
  uint32_t *rel;    /* [n_set][16]: where class (mismatches d, index s) of the guide starts in its hit list, less the
                       records of item s in classes before d: a record's place = rel[2d + s] + its rank in the item */
  uint32_t *nhits;  /* of the batch: hits per guide */
  uint32_t *flags;
};
__global__ __launch_bounds__(256) void k_to_fill(gs_to_fill_args a) {
  const uint32_t sb = blockIdx.x * blockDim.x + threadIdx.x;
  if (sb >= a.n_it) return;
  const uint32_t g = a.list ? a.list[sb >> 1] : (sb >> 1);
  const uint32_t item = 2u * g + (sb & 1u);
  const uint32_t c = a.counts[item];
  const uint32_t nb = to_buckets(c), tb = a.tbase[sb];
  if (c != 0u && nb == 0u) a.tiles[tb] = make_uint4(sb | TO_DIRECT, 0u, c, 0u);
  if (nb != 0u && nb <= TO_NBMAX) {
    const uint32_t bb = a.bbase[sb];
    for (uint32_t b = 0; b < nb; ++b) a.tiles[tb + b] = make_uint4(sb, bb + b, 0u, 0u);
    a.biglist[a.gbase[sb]] = sb;
  }
  if ((sb & 1u) == 0u) {
    const uint32_t *c0 = a.cls + (size_t)item * 8u, *c1 = c0 + 8u;
    uint32_t p0 = 0, p1 = 0;
    for (uint32_t d = 0; d < 8u; ++d) {
      a.rel[(size_t)(sb >> 1) * 16u + 2u * d] = p1; /* index 0: the other index's records of the classes before */
      p0 += c0[d];
      a.rel[(size_t)(sb >> 1) * 16u + 2u * d + 1u] = p0; /* index 1: index 0's records up to and including this class */
      p1 += c1[d];
    }
    const uint32_t cb = a.counts[item + 1u];
    if (p0 != c || p1 != cb) atomicOr(a.flags, TO_F_CLS); /* (cannot happen: k_search counts both) */
    a.nhits[g] = c + cb;
  }
}
/* records of the set (flags[2..3] as one 64-bit count) */
__global__ __launch_bounds__(256) void k_to_total(const uint32_t *counts, const uint32_t *list, uint32_t n_it, unsigned long long *out) {
  unsigned long long v = 0;
  for (uint64_t sb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sb < n_it; sb += (uint64_t)gridDim.x * blockDim.x)
    v += counts[list ? 2u * list[sb >> 1] + ((uint32_t)sb & 1u) : (uint32_t)sb];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane_id() == 0 && v) atomicAdd(out, v);
}
struct gs_to_chunk_args {
  const uint32_t *counts, *chunk_item, *chunk_seq, *redo_pos, *cbase;
  uint32_t *chunk_of;
  uint32_t n_used, cap, by_list;
};
__global__ __launch_bounds__(256) void k_to_chunks(gs_to_chunk_args a) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.n_used) return;
  const uint32_t item = a.chunk_item[c];
  uint32_t sb = item;
  if (a.by_list) {
    const uint32_t pos = a.redo_pos[item >> 1];
    if (pos == 0xFFFFFFFFu) return;
    sb = 2u * pos + (item & 1u);
  }
  const uint32_t seq = a.chunk_seq[c];
  if (a.counts[item] > a.cap + (seq << ARENA_SHIFT)) a.chunk_of[a.cbase[sb] + seq] = c;
}

/* record i of an item where k_search left it: its slots, then its arena chunks in order */
struct gs_to_src {
  const uint4 *slots, *arena;
  const uint32_t *chunk_of;
  uint32_t cap;
};
__device__ __forceinline__ const uint4 *to_addr(const gs_to_src &s, const uint32_t item, const uint32_t cb, const uint32_t i) {
  if (i < s.cap) return s.slots + (size_t)item * s.cap + i;
  const uint32_t e = i - s.cap;
  return s.arena + (((size_t)s.chunk_of[cb + (e >> ARENA_SHIFT)] << ARENA_SHIFT) | (e & (ARENA_CHUNK - 1u)));
}

struct gs_to_run_args {
  gs_to_src src;
  const uint32_t *counts, *list;
  const uint32_t *tbase, *bbase, *cbase;
  const uint32_t *biglist;
  uint4 *tiles;
  uint4 *buckets;
  const gs_to_tab *tab;
  const uint32_t *rel;
  const uint64_t *offsets;
  gs_hit *hits;
  const uint32_t *sa[2];
  uint64_t genome_length;
  uint32_t *flags;
  uint32_t L, P, v_rem;
  uint32_t sample_per; /* 0, or GS_TILE_SAMPLE_PER (tests) */
};

/* ---- items beyond one tile: splitters from a sample, then one streaming pass into buckets ---------------------- */
__global__ __launch_bounds__(TO_NT) void k_to_partition(gs_to_run_args a) {
  __shared__ unsigned long long s_keys[TO_LDS];
  __shared__ unsigned long long s_spl[TO_NBMAX];
  __shared__ unsigned long long s_nt[32 * 8], s_bs[8];
  __shared__ uint32_t s_cur[TO_NBMAX];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u);
  for (uint32_t i = tid; i < 32u * 8u; i += TO_NT) s_nt[i] = a.tab->n[i >> 3][i & 7u];
  if (tid < 8u) s_bs[tid] = a.tab->base[tid];
  const unsigned long long pam_mul = a.tab->pam_mul;
  const uint32_t sb = a.biglist[blockIdx.x];
  const uint32_t g = a.list ? a.list[sb >> 1] : (sb >> 1);
  const uint32_t item = 2u * g + (sb & 1u);
  const uint32_t c = a.counts[item], nb = to_buckets(c);
  const uint32_t bb = a.bbase[sb], tb = a.tbase[sb], cb = a.cbase[sb];
  /* stratified sample: one record from each of `ns` equal stretches of the item (emission order is seed by seed,
   * so a stretch is a few neighbouring sequences: no worse than independent draws) */
  uint32_t per = TO_TILE / nb < 64u ? TO_TILE / nb : 64u; /* samples per bucket */
  if (a.sample_per != 0u && a.sample_per < per) per = a.sample_per; /* tests: a sample too small to keep the buckets within their slots */
  const uint32_t slots = a.sample_per == 1u ? TO_TILE / 4u : TO_TILE; /* ... and, at one word per splitter, a quarter of the slots: certain to overflow */
  const uint32_t ns = per * nb;
  __syncthreads();
  bool multi = false;
  for (uint32_t j = tid; j < ns; j += TO_NT) {
    const uint32_t lo = (uint32_t)(((unsigned long long)j * c) / ns), hi = (uint32_t)(((unsigned long long)(j + 1u) * c) / ns);
    uint32_t h = j * 2654435761u ^ (sb * 40503u + 0x9E3779B9u);
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    const uint32_t pos = lo + h % (hi - lo); /* hi > lo: c > TO_TILE >= ns */
    const uint4 rec = *to_addr(a.src, item, cb, pos);
    s_keys[TO_AT(j)] = to_word(rec, a.L, a.P, s_nt, s_bs, pam_mul);
  }
  __syncthreads();
  to_msort<false>(s_keys, nullptr, ns);
  unsigned long long spl_mine[2] = {0ull, 0ull};
  for (uint32_t b = tid, q = 0; b + 1u < nb; b += TO_NT, ++q) spl_mine[q] = s_keys[TO_AT((b + 1u) * per - 1u)];
  __syncthreads();
  for (uint32_t b = tid, q = 0; b + 1u < nb; b += TO_NT, ++q) s_spl[b] = spl_mine[q];
  for (uint32_t b = tid; b < nb; b += TO_NT) s_cur[b] = 0u;
  __syncthreads();
  uint4 *out = a.buckets + (size_t)bb * TO_TILE;
  const uint32_t c_pad = ((c + TO_NT - 1u) / TO_NT) * TO_NT;
  for (uint32_t i0 = tid; i0 < c_pad; i0 += 4u * TO_NT) {
    uint4 rec[4];
    bool on[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) { /* four loads in flight */
      const uint32_t i = i0 + u * TO_NT;
      on[u] = i < c;
      rec[u] = on[u] ? *to_addr(a.src, item, cb, i) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      if (i0 - tid + u * TO_NT >= c_pad) break; /* workgroup-uniform */
      uint32_t b = 0, kw = 0;
      if (on[u]) {
        multi = multi || rec[u].z != rec[u].w;
        const unsigned long long K = to_word(rec[u], a.L, a.P, s_nt, s_bs, pam_mul);
        kw = (uint32_t)(K >> 32);
        uint32_t lo = 0, hi = nb - 1u; /* splitters below K */
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_spl[mid] < K)
            lo = mid + 1u;
          else
            hi = mid;
        }
        b = lo;
      }
      /* one LDS atomic per wave when its records go to one bucket (the stretches of a long run), else one per lane */
      const unsigned long long act = __ballot(on[u]);
      if (act == 0ull) continue;
      const uint32_t b0 = (uint32_t)__shfl((int)b, (int)__builtin_ctzll(act));
      uint32_t pos;
      if (__ballot(on[u] && b != b0) == 0ull) {
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(act)) base = atomicAdd(&s_cur[b0], (uint32_t)__popcll(act));
        base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(act));
        pos = base + lanes_below(act);
      } else {
        pos = on[u] ? atomicAdd(&s_cur[b], 1u) : 0u;
      }
      /* a bucket record carries its sequence word where the (equal) last row was: the tile does not rank it again */
      if (on[u] && pos < slots) out[(size_t)b * TO_TILE + pos] = make_uint4(rec[u].x, rec[u].y, rec[u].z, kw);
    }
  }
  if (multi) atomicOr(a.flags, TO_F_MULTIROW);
  __syncthreads();
  /* records per bucket and before it (the first wave: nb <= 1,024 = 16 per lane) */
  if (tid < WAVE) {
    const uint32_t each = (nb + WAVE - 1u) / WAVE;
    uint32_t s = 0;
    bool over = false;
    for (uint32_t q = 0; q < each; ++q) {
      const uint32_t b = tid * each + q;
      if (b < nb) {
        s += s_cur[b];
        over = over || s_cur[b] > slots;
      }
    }
    uint32_t run = wave_incl_sum(s) - s;
    for (uint32_t q = 0; q < each; ++q) {
      const uint32_t b = tid * each + q;
      if (b < nb) {
        const uint32_t n = s_cur[b];
        a.tiles[tb + b].z = n < slots ? n : slots;
        a.tiles[tb + b].w = run;
        run += n;
      }
    }
    if (over) atomicOr(a.flags, TO_F_BUCKET);
  }
}

/* ---- one tile: order its records in LDS and write its hits -------------------------------------------------------- */
/* NT threads, tiles of NT x 8 records at most: 512 for the tiles proper, 128 for the many small ones (an m <= 5 batch is
 * 2 x 10^5 items of ~700 records: in a workgroup of 512 a merge round keeps 90 threads busy and seven waves wait at its
 * barriers; 12 KB of LDS instead of 47: twelve workgroups per CU).  Both are launched over the whole tile list, each
 * takes its size class (lo < n <= NT x 8). */
template <uint32_t NT>
__global__ __launch_bounds__(NT) void k_to_sort(gs_to_run_args a, const uint32_t lo) {
  constexpr uint32_t CAP = NT * TO_KPT, LDSN = CAP + CAP / 8u;
  __shared__ unsigned long long s_keys[LDSN];
  __shared__ unsigned long long s_nt[32 * 8], s_bs[8];
  __shared__ uint16_t s_idx[LDSN];
  __shared__ uint32_t s_rel[8];
  __shared__ unsigned long long s_first;

  const uint32_t tid = threadIdx.x;
  /* one workgroup per tile (persistent workgroups looping over the tiles were tried: the loop took the kernel from 71
   * to 156 registers, one workgroup per CU instead of three, 22 ms instead of 9) */
  const uint4 t = a.tiles[blockIdx.x];
  const uint32_t n = t.z;
  if (n <= lo || n > CAP) return;
#ifdef TO_PROFILE
  const unsigned long long tp0 = wall_clock64();
#endif
  const bool direct = (t.x & TO_DIRECT) != 0u;
  const uint32_t sb = t.x & ~TO_DIRECT;
  const uint32_t gset = sb >> 1, strand = sb & 1u;
  const uint32_t g = a.list ? a.list[gset] : gset;
  const uint32_t item = 2u * g + strand;
  const uint4 *bucket = a.buckets + (size_t)t.y * TO_TILE;
  /* every global read the tile needs is asked for before anything waits: the records (eight per thread), where the
   * guide's classes start, the guide's first hit */
  uint4 rec[TO_KPT];
  if (direct) {
    const uint32_t cb = a.cbase[sb];
#pragma unroll
    for (uint32_t u = 0; u < TO_KPT; ++u) {
      const uint32_t i = tid + u * NT;
      rec[u] = i < n ? *to_addr(a.src, item, cb, i) : make_uint4(0u, 0u, 0u, 0u);
    }
  } else {
#pragma unroll
    for (uint32_t u = 0; u < TO_KPT; ++u) {
      const uint32_t i = tid + u * NT;
      rec[u] = i < n ? bucket[i] : make_uint4(0u, 0u, 0u, 0u); /* {key, row, sequence word}: k_to_partition ranked it */
    }
  }
  if (tid < 8u) s_rel[tid] = a.rel[(size_t)gset * 16u + 2u * tid + strand];
  if (tid == 8u) s_first = a.offsets[g] + t.w;
  bool multi = false;
  uint2 kx[TO_KPT]; /* the records' keys, in load order: they wait in registers while the words are ordered */
  if (direct) {
    for (uint32_t i = tid; i < 32u * 8u; i += NT) s_nt[i] = a.tab->n[i >> 3][i & 7u];
    if (tid < 8u) s_bs[tid] = a.tab->base[tid];
    const unsigned long long pam_mul = a.tab->pam_mul;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < TO_KPT; ++u) {
      const uint32_t i = tid + u * NT;
      kx[u] = make_uint2(rec[u].x, rec[u].y);
      if (i < n) {
        multi = multi || rec[u].z != rec[u].w;
        s_keys[TO_AT(i)] = to_word(rec[u], a.L, a.P, s_nt, s_bs, pam_mul);
        s_idx[TO_AT(i)] = (uint16_t)i;
      }
    }
  } else {
#pragma unroll
    for (uint32_t u = 0; u < TO_KPT; ++u) {
      const uint32_t i = tid + u * NT;
      kx[u] = make_uint2(rec[u].x, rec[u].y);
      if (i < n) {
        s_keys[TO_AT(i)] = ((unsigned long long)rec[u].w << 32) | rec[u].z;
        s_idx[TO_AT(i)] = (uint16_t)i;
      }
    }
  }
  if (multi) atomicOr(a.flags, TO_F_MULTIROW);
  __syncthreads();
#ifdef TO_PROFILE
  const unsigned long long tp1 = wall_clock64();
#endif
  to_msort<true>(s_keys, s_idx, n);
#ifdef TO_PROFILE
  const unsigned long long tp2 = wall_clock64();
#endif
  /* the hits: place = the guide's first + the class's start + rank in the item (process.hpp:100-115).  Consecutive
   * lanes take consecutive places; the ordered words and the records they came from are read into registers, then
   * the keys - still in registers in load order - take the words' place in LDS and are picked up from there: the
   * only random global request left per hit is the suffix array's (the first version read each key again from the
   * bucket: 10^9 random requests per batch, the rate the memory system serves them at was the kernel's bound) */
  const uint32_t *sa = a.sa[strand];
  gs_hit *out = a.hits + s_first;
  bool dup = false;
  uint32_t row[TO_KPT], from[TO_KPT], sav[TO_KPT];
#pragma unroll
  for (uint32_t u = 0; u < TO_KPT; ++u) {
    const uint32_t r = tid + u * NT;
    row[u] = from[u] = sav[u] = 0u;
    if (r < n) {
      const unsigned long long K = s_keys[TO_AT(r)];
      from[u] = s_idx[TO_AT(r)];
      dup = dup || (r > 0u && s_keys[TO_AT(r - 1u)] == K);
      row[u] = (uint32_t)K;
      sav[u] = sa[row[u]];
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t u = 0; u < TO_KPT; ++u) {
    const uint32_t i = tid + u * NT;
    if (i < n) s_keys[TO_AT(i)] = ((unsigned long long)kx[u].y << 32) | kx[u].x;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t u = 0; u < TO_KPT; ++u) {
    const uint32_t r = tid + u * NT;
    if (r < n) {
      const unsigned long long key = s_keys[TO_AT(from[u])];
      const uint64_t sap = (uint64_t)sav[u] - ((key & 1ull) ? a.v_rem : 0u);
      gs_hit o;
      o.pos = strand == 0u ? -(int64_t)sap : (int64_t)(a.genome_length - (sap + 1ull));
      o.key = key & ~1ull;
      out[s_rel[(uint32_t)(key >> 61)] + r] = o;
    }
  }
  if (dup) atomicOr(a.flags, TO_F_DUP);
#ifdef TO_PROFILE
  __syncthreads();
  if (tid == 0u) { /* 100 MHz ticks per phase: load, order, write; tiles; records */
    unsigned long long *dbg = (unsigned long long *)(a.flags + 4);
    const unsigned long long tp3 = wall_clock64();
    atomicAdd(&dbg[0], tp1 - tp0);
    atomicAdd(&dbg[1], tp2 - tp1);
    atomicAdd(&dbg[2], tp3 - tp2);
    atomicAdd(&dbg[3], 1ull);
    atomicAdd(&dbg[4], (unsigned long long)n);
  }
#endif
}

/* ---- host side ------------------------------------------------------------------------------------------------ */
static bool to_make_tab(uint32_t L, uint32_t P, uint32_t m, gs_to_tab &tab) {
  memset(&tab, 0, sizeof(tab));
  for (uint32_t a = 0; a < 32; a++)
    for (uint32_t r = 0; r < 8; r++) {
      unsigned long long v = 0;
      if (r <= a) {
        double c = 1;
        for (uint32_t i = 0; i < r; i++) c = c * (double)(a - i) / (double)(i + 1);
        v = (unsigned long long)(c + 0.5);
        for (uint32_t i = 0; i < r; i++) v *= 3ull;
      }
      tab.n[a][r] = v;
    }
  tab.pam_mul = 1;
  for (uint32_t u = 0; u < P; u++) tab.pam_mul *= 5ull;
  long double cum = 0;
  unsigned long long c64 = 0;
  for (uint32_t j = 0; j < 8; j++) {
    tab.base[j] = c64;
    if (j <= m && j <= L) {
      cum += (long double)tab.n[L][j] * (long double)tab.pam_mul;
      c64 += tab.n[L][j] * tab.pam_mul;
    }
  }
  return L >= 1 && 2 * L + 3 * P <= 59 && m <= 7 && cum < 4294967295.0L; /* the words stay below 2^32 - 1: all ones marks padding */
}
bool gs_tileorder_fits(uint32_t L, uint32_t P, uint32_t m) {
  gs_to_tab tab;
  return to_make_tab(L, P, m, tab);
}

gs_status gs_tileorder_plan(gs_index *ix, const gs_tileorder_in &in, hipStream_t st, gs_tileorder_state &S, bool *usable) {
  *usable = false;
  gs_to_tab tab;
  if (!to_make_tab(in.L, in.P, in.m, tab)) return GS_OK;
  gs_status rc;
  const uint32_t n_it = 2u * in.n_set;
  S = gs_tileorder_state();
  S.n_it = n_it;
  if ((rc = gs_reserve(ix->w_t_plan, 4 * 4 * ((size_t)n_it + 1) + 64)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_tab, sizeof(gs_to_tab) + 64)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_rel, 64 * ((size_t)in.n_set + 1))) != GS_OK) return rc;
  uint32_t *tbase = (uint32_t *)ix->w_t_plan.p, *bbase = tbase + (n_it + 1), *cbase = bbase + (n_it + 1), *gbase = cbase + (n_it + 1);
  uint32_t *d_flags = (uint32_t *)((char *)ix->w_t_tab.p + sizeof(gs_to_tab));
  GS_HIP(hipMemcpyAsync(ix->w_t_tab.p, &tab, sizeof(tab), hipMemcpyHostToDevice, st));
  GS_HIP(hipMemsetAsync(d_flags, 0, 64, st));
  gs_to_plan_args pa;
  pa.counts = in.counts;
  pa.list = in.list;
  pa.n_it = n_it;
  pa.cap = in.cap;
  pa.tbase = tbase;
  pa.bbase = bbase;
  pa.cbase = cbase;
  pa.gbase = gbase;
  pa.flags = d_flags;
  hipLaunchKernelGGL(k_to_plan, dim3(1), dim3(1024), 0, st, pa);
  uint32_t tot[4] = {0, 0, 0, 0}, h_flags = 0;
  GS_HIP(hipMemcpyAsync(&tot[0], tbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&tot[1], bbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&tot[2], cbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&tot[3], gbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&h_flags, d_flags, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st)); /* (tab is a local, too) */
  if (h_flags) return GS_OK;
  S.n_tiles = tot[0];
  S.n_btiles = tot[1];
  S.n_chunks = tot[2];
  S.n_big = tot[3];
  if ((rc = gs_reserve(ix->w_t_tiles, 16 * ((size_t)S.n_tiles + 1))) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_buckets, 16 * (size_t)TO_TILE * S.n_btiles + 16)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_chunkof, 4 * ((size_t)S.n_chunks + 1))) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_big, 4 * ((size_t)S.n_big + 1))) != GS_OK) return rc;
  gs_to_fill_args fa;
  fa.counts = in.counts;
  fa.cls = in.cls;
  fa.list = in.list;
  fa.n_it = n_it;
  fa.cap = in.cap;
  fa.tbase = tbase;
  fa.bbase = bbase;
  fa.gbase = gbase;
  fa.tiles = (uint4 *)ix->w_t_tiles.p;
  fa.biglist = (uint32_t *)ix->w_t_big.p;
  fa.rel = (uint32_t *)ix->w_t_rel.p;
  fa.nhits = in.nhits;
  fa.flags = d_flags;
  hipLaunchKernelGGL(k_to_fill, dim3((n_it + 255) / 256), dim3(256), 0, st, fa);
  hipLaunchKernelGGL(k_to_total, dim3(std::min<uint32_t>((n_it + 255) / 256, 512u)), dim3(256), 0, st, in.counts, in.list, n_it,
                     (unsigned long long *)(d_flags + 2));
  if (in.n_used) {
    gs_to_chunk_args ca;
    ca.counts = in.counts;
    ca.chunk_item = in.chunk_item;
    ca.chunk_seq = in.chunk_seq;
    ca.redo_pos = in.redo_pos;
    ca.cbase = cbase;
    ca.chunk_of = (uint32_t *)ix->w_t_chunkof.p;
    ca.n_used = in.n_used;
    ca.cap = in.cap;
    ca.by_list = in.list ? 1u : 0u;
    hipLaunchKernelGGL(k_to_chunks, dim3((in.n_used + 255) / 256), dim3(256), 0, st, ca);
  }
  *usable = true;
  return GS_OK;
}

gs_status gs_tileorder_run(gs_index *ix, const gs_tileorder_in &in, gs_tileorder_state &S, hipStream_t st, uint32_t *violations) {
  *violations = 0;
  const uint32_t n_it = S.n_it;
  uint32_t *tbase = (uint32_t *)ix->w_t_plan.p, *bbase = tbase + (n_it + 1), *cbase = bbase + (n_it + 1);
  uint32_t *d_flags = (uint32_t *)((char *)ix->w_t_tab.p + sizeof(gs_to_tab));
  gs_to_run_args ra;
  ra.src.slots = in.slots;
  ra.src.arena = in.arena;
  ra.src.chunk_of = (const uint32_t *)ix->w_t_chunkof.p;
  ra.src.cap = in.cap;
  ra.counts = in.counts;
  ra.list = in.list;
  ra.tbase = tbase;
  ra.bbase = bbase;
  ra.cbase = cbase;
  ra.biglist = (const uint32_t *)ix->w_t_big.p;
  ra.tiles = (uint4 *)ix->w_t_tiles.p;
  ra.buckets = (uint4 *)ix->w_t_buckets.p;
  ra.tab = (const gs_to_tab *)ix->w_t_tab.p;
  ra.rel = (const uint32_t *)ix->w_t_rel.p;
  ra.offsets = in.offsets;
  ra.hits = in.hits;
  ra.sa[0] = ix->strand[0].d.sa;
  ra.sa[1] = ix->strand[1].d.sa;
  ra.genome_length = ix->genome_length;
  ra.flags = d_flags;
  ra.L = in.L;
  ra.P = in.P;
  ra.v_rem = in.v_rem;
  ra.sample_per = getenv("GS_TILE_SAMPLE_PER") ? (uint32_t)std::max(1l, atol(getenv("GS_TILE_SAMPLE_PER"))) : 0u;
  if (S.n_big) hipLaunchKernelGGL(k_to_partition, dim3(S.n_big), dim3(TO_NT), 0, st, ra);
  if (S.n_tiles) {
    hipLaunchKernelGGL(k_to_sort<128u>, dim3(S.n_tiles), dim3(128), 0, st, ra, 0u);
    hipLaunchKernelGGL(k_to_sort<TO_NT>, dim3(S.n_tiles), dim3(TO_NT), 0, st, ra, 128u * TO_KPT);
  }
  uint32_t h[16] = {0};
  GS_HIP(hipMemcpyAsync(h, d_flags, 64, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
#ifdef TO_PROFILE
  {
    const unsigned long long *d = (const unsigned long long *)(h + 4);
    fprintf(stderr, "[gs] k_to_sort phases (wall-clock ticks per tile, 100 MHz): load %.0f, order %.0f, write %.0f; %llu tiles, %.0f records each\n",
            (double)d[0] / d[3], (double)d[1] / d[3], (double)d[2] / d[3], d[3], (double)d[4] / d[3]);
  }
#endif
  if (getenv("GS_DEBUG"))
    fprintf(stderr, "[gs] tile ordering: %u items, %u of them partitioned into %u buckets, %u tiles\n", S.n_it, S.n_big, S.n_btiles, S.n_tiles);
  *violations = h[0];
  S.n_records = ((uint64_t)h[3] << 32) | h[2];
  GS_HIP(hipGetLastError());
  return GS_OK;
}
