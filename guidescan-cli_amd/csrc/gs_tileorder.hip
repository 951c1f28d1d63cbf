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
 *   k_to_plan       one workgroup: per set item the tiles, bucket space, arena chunks and dealing stretches it needs;
 *                   five scans
 *   k_to_fill       tile descriptors, a descriptor per dealing stretch, per guide where each (mismatches, index)
 *                   class starts in its hit list (from k_search's per-class counts), hits per guide
 *   k_to_splitters  items of more than 4,096 records: a sample of their records ordered in LDS gives splitters
 *   k_to_deal       ... one streaming pass, a workgroup per 2,048 records, deals the records into buckets that aim at
 *                   400 (sample sort: the full word decides, so a run of 10^4 equal sequences is split by row)
 *   k_to_bucketsum  ... records before each bucket; the buckets beyond 512 records go on the workgroup kernels' lists
 *   k_to_wsort      ONE WAVE per tile of at most 512 records (a small item where k_search left it, or a bucket): K of
 *                   every record, a bitonic network over the wave's registers (DPP), then the hits - suffix array
 *                   gather + coordinate rule (process.hpp:104, 111) - straight to their final place
 *   k_to_sort       one workgroup per tile of 513 .. 4,096 records: a merge sort in LDS (eight words per thread ordered
 *                   in registers, then merge-path rounds); 128 threads for tiles of up to 1,024 records, 512 beyond
 *
 * Two passes of HBM traffic for dealt items (16 B read + 16 B written, twice), one for the rest.
 * The path writes final hits on two assumptions it checks as it goes: every record is a single row, and no
 * (sequence, row) occurs twice in an item (overlapping PAM patterns).  A tile that sees otherwise - or a bucket
 * that outgrows its space - raises a flag; the caller then orders the batch with the device-wide form.
 */
#include "gs_device.h"

#include <algorithm>

#define TO_NT 512u                 /* threads of the workgroup-tile and dealing kernels */
#define TO_NW (TO_NT / WAVE)       /* 8 waves */
#define TO_KPT 8u                  /* keys per thread */
#define TO_TILE (TO_NT * TO_KPT)   /* 4,096 records: the largest tile (one workgroup); items beyond are dealt into buckets */
#define TO_NBMAX 1024u             /* buckets of one item */
#define TO_BU 128u                 /* bucket slots are handed out in units of 128 records */
#define TO_SNT 1024u               /* threads of k_to_splitters */
#define TO_SAMPLE (TO_SNT * TO_KPT) /* 8,192: the largest sample of an item */
#define TO_DEAL 2048u              /* records one workgroup of k_to_deal deals into buckets */
#define TO_WTILE 512u              /* records one wave orders in registers (k_to_wsort) */
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
 * the key's own bits order them (the same number gs_search.hip's big2_rank computes).  32-bit arithmetic: the path is
 * only taken when every word of the batch is below 2^32 - 1 (to_make_tab), so are the table entries a record within
 * the batch's mismatch limit can reach - the kernels keep the table's low words. */
__device__ __forceinline__ uint32_t to_rank(const unsigned long long key, const uint32_t L, const uint32_t P, const uint32_t *nt,
                                            const uint32_t pam_mul) {
  const unsigned long long path = (key >> 1) & ((1ull << 59) - 1ull); /* key bits 59:1 */
  /* the guide's L two-bit fields, position 0 in the highest: only the substituted ones (at most seven) count */
  const unsigned long long gb = path >> (59u - 2u * L);
  unsigned long long nz = (gb | (gb >> 1)) & 0x5555555555555555ull & ((1ull << (2u * L)) - 1ull);
  uint32_t r = (uint32_t)__popcll(nz);
  if (r > 7u) r = 7u;
  uint32_t rank = 0;
  while (nz != 0ull && r != 0u) {
    const uint32_t hb = 63u - (uint32_t)__builtin_clzll(nz); /* = 2 x the positions behind this one */
    const uint32_t c = (uint32_t)(gb >> hb) & 3u, a = hb >> 1;
    /* smaller sequences with the same prefix: a 0 here (r substitutions behind), or one of the c-1 lower codes */
    const uint32_t n0 = nt[a * 8u + r - 1u];
    rank += nt[a * 8u + r] + (c == 2u ? n0 : c == 3u ? 2u * n0 : 0u);
    r--;
    nz &= ~(1ull << hb);
  }
  uint32_t pr = 0;
  for (uint32_t u = 0; u < P; u++) {
    const uint32_t c = (uint32_t)(path >> (56u - 2u * L - 3u * u)) & 7u;
    pr = pr * 5u + (c < 4u ? c : 4u);
  }
  return rank * pam_mul + pr;
}
__device__ __forceinline__ unsigned long long to_word(const uint4 rec, const uint32_t L, const uint32_t P, const uint32_t *nt,
                                                       const uint32_t *bs, const uint32_t pam_mul) {
  const unsigned long long key = ((unsigned long long)rec.y << 32) | rec.x;
  const uint32_t w = bs[rec.y >> 29] + to_rank(key, L, P, nt, pam_mul);
  return ((unsigned long long)w << 32) | rec.z;
}
/* four records at once, step by step: the rank loop reads its table through a chain of LDS reads, each waiting for
 * the one before - four chains side by side hide three quarters of that wait (the dealing pass was bound by it) */
__device__ __forceinline__ void to_word4(const uint4 (&rec)[4], const uint32_t L, const uint32_t P, const uint32_t *nt, const uint32_t *bs,
                                         const uint32_t pam_mul, unsigned long long (&K)[4]) {
  unsigned long long gb[4], nz[4], path[4];
  uint32_t r[4], rank[4];
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) {
    const unsigned long long key = ((unsigned long long)rec[u].y << 32) | rec[u].x;
    path[u] = (key >> 1) & ((1ull << 59) - 1ull);
    gb[u] = path[u] >> (59u - 2u * L);
    nz[u] = (gb[u] | (gb[u] >> 1)) & 0x5555555555555555ull & ((1ull << (2u * L)) - 1ull);
    r[u] = (uint32_t)__popcll(nz[u]);
    if (r[u] > 7u) r[u] = 7u;
    rank[u] = 0u;
  }
  for (;;) {
    bool any = false;
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const bool go = nz[u] != 0ull && r[u] != 0u;
      const uint32_t hb = go ? 63u - (uint32_t)__builtin_clzll(nz[u]) : 0u;
      const uint32_t c = (uint32_t)(gb[u] >> hb) & 3u, a = hb >> 1;
      const uint32_t n1 = nt[a * 8u + (go ? r[u] : 1u)], n0 = nt[a * 8u + (go ? r[u] : 1u) - 1u];
      if (go) {
        rank[u] += n1 + (c == 2u ? n0 : c == 3u ? 2u * n0 : 0u);
        r[u]--;
        nz[u] &= ~(1ull << hb);
      }
      any = any || go;
    }
    if (!any) break;
  }
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) {
    uint32_t pr = 0;
    for (uint32_t q = 0; q < P; q++) {
      const uint32_t c = (uint32_t)(path[u] >> (56u - 2u * L - 3u * q)) & 7u;
      pr = pr * 5u + (c < 4u ? c : 4u);
    }
    const uint32_t w = bs[rec[u].y >> 29] + rank[u] * pam_mul + pr;
    K[u] = ((unsigned long long)w << 32) | rec[u].z;
  }
}

/* ---- merge sort of n <= TO_TILE 64-bit words in LDS by the whole workgroup --------------------------------------
 * keys (+ idx: a 16-bit payload that moves with its key), n + n / 8 entries each: word e lives at TO_AT(e) - one pad
 * entry after every eight, so that threads reading their own eight consecutive words hit different banks (blocked
 * accesses at a 64-byte stride were a 32-way bank conflict: 65 % of the LDS cycles of the first version).
 * Thread t owns places 8t .. 8t+7: it orders its eight words in registers (Batcher's network, 19 exchanges), then
 * log2(n / 8) rounds merge neighbouring runs pairwise - the thread finds where its eight places of the merged run
 * begin in either input (merge path: a binary search along its diagonal) and merges eight words from there, from LDS
 * into registers; all write back after a barrier.  About 30 instructions per word and round whatever the words' bits - the radix passes this replaced (one ballot per key bit
 * to rank a wave's keys stably) cost 90 per word and pass, eight passes on a repeat-rich tile.
 * Places n .. 8 * ceil(n / 8) hold words of all ones. */
#define TO_AT(e) ((e) + ((e) >> 3))
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
  uint32_t *tbase, *bbase, *cbase, *gbase, *dbase; /* [n_it + 1] first tile / first bucket slot / first chunk-index entry / place on the list of dealt
                                                      items / first stretch of TO_DEAL records the dealing kernel takes */
  uint32_t *flags;
};
/* buckets of an item of c > TO_TILE records: how many, and the slot each gets (su units of TO_BU records).  The
 * buckets aim below the 512 records one wave orders in registers (k_to_wsort); what a splitter's luck makes larger is a
 * tile of the workgroup kernels.  An item affords TO_SAMPLE samples in all: the fewer per splitter, the wider a bucket's
 * size spreads (a gamma distribution of that shape) and the more room its slot has over the aim - every row keeps the
 * chance of one bucket outgrowing its slot below 10^-10 (32 samples: 2.9 x the aim; 16: 4; 8: 6.2).  No more than 32
 * samples per splitter: ordering the sample is itself a sort of ns words per item (64 per splitter, 4,096 for an item of
 * 24,000 records, cost a third of what ordering the records costs). */
__device__ __host__ __forceinline__ uint32_t to_buckets(const uint32_t c, uint32_t &su) {
  su = 0u;
  if (c <= TO_TILE) return 0u;
  /* A bucket's slot is 1.5-1.8 x its aim (until round 5: 2.9-6.2 x, so that no bucket ever outgrew it - 3.7 x the
   * records in bucket space, 29 GB on the repeat-rich batch and 60 GB at 10^9 records).  What a bucket receives beyond
   * its slot goes to a spill list; a bucket that did is moved, whole, to a slot of its real size behind the planned
   * ones (k_to_bucketsum, k_to_respill): one bucket in a thousand at 32 samples per splitter, a few in a hundred at 8. */
  uint32_t target;
  if (c <= 256u * 400u)
    target = 400u, su = 4u; /* 32 samples per splitter: one bucket in fifteen grows past its 512 */
  else if (c <= 512u * 320u)
    target = 320u, su = 4u; /* 16 */
  else if (c <= 1024u * 288u)
    target = 288u, su = 4u; /* 8 */
  else
    target = 1024u, su = 12u; /* 8 (to a million records: 1,024 buckets, ordered by the workgroup kernels - the 1,024-thread form takes 8,192) */
  return (c + target - 1u) / target;
}
/* An item beyond TO_NBMAX buckets of 1,024 records (a guide inside the largest repeat family of a genome: 10^6 records
 * and more) is not dealt: its GUIDE - both items - is left out of the tiles and goes, alone, through the device-wide
 * ordering (gs_search.hip, big_order on the list k_to_fill writes); the rest of the batch stays here. */
#define TO_ITEM_MAX (TO_NBMAX * 1024u)
__device__ __forceinline__ bool to_left_out(const uint32_t *counts, const uint32_t item) {
  return counts[item] > TO_ITEM_MAX || counts[item ^ 1u] > TO_ITEM_MAX;
}
__global__ __launch_bounds__(1024) void k_to_plan(gs_to_plan_args a) {
  __shared__ uint32_t s_w[5][16], s_carry[5];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u), w = tid / WAVE;
  if (tid < 5u) s_carry[tid] = 0u;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < a.n_it + 1u; i0 += 1024u) {
    const uint32_t sb = i0 + tid;
    uint32_t v[5] = {0u, 0u, 0u, 0u, 0u};
    if (sb < a.n_it) {
      const uint32_t item = a.list ? 2u * a.list[sb >> 1] + (sb & 1u) : sb;
      const uint32_t c = to_left_out(a.counts, item) ? 0u : a.counts[item]; /* (a guide that is left out has no tiles, buckets or chunks here) */
      uint32_t su;
      const uint32_t nb = to_buckets(c, su);
      v[0] = c == 0u ? 0u : nb ? nb : 1u;
      v[1] = nb * su;
      v[2] = c > a.cap ? (c - a.cap + ARENA_CHUNK - 1u) / ARENA_CHUNK : 0u;
      v[3] = nb ? 1u : 0u;
      v[4] = nb ? (c + TO_DEAL - 1u) / TO_DEAL : 0u;
    }
    uint32_t ex[5];
#pragma unroll
    for (uint32_t q = 0; q < 5u; ++q) {
      const uint32_t incl = wave_incl_sum(v[q]);
      if (lane == WAVE - 1u) s_w[q][w] = incl;
      ex[q] = incl - v[q];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < 5u; ++q) {
      uint32_t b = s_carry[q];
      for (uint32_t u = 0; u < w; ++u) b += s_w[q][u];
      ex[q] += b;
    }
    if (sb <= a.n_it) {
      a.tbase[sb] = ex[0];
      a.bbase[sb] = ex[1];
      a.cbase[sb] = ex[2];
      a.gbase[sb] = ex[3];
      a.dbase[sb] = ex[4];
    }
    __syncthreads();
    if (tid == 1023u) {
#pragma unroll
      for (uint32_t q = 0; q < 5u; ++q) s_carry[q] = ex[q] + v[q];
    }
    __syncthreads();
  }
}

/* ---- tile descriptors, chunk index, class starts ------------------------------------------------------------- */
struct gs_to_fill_args {
  const uint32_t *counts, *cls, *list;
  uint32_t n_it, cap;
  const uint32_t *tbase, *bbase, *gbase, *dbase;
  uint32_t *slow[3], *slow_n; /* tiles beyond one wave's 512 records: lists (to 1,024 records; to 4,096; beyond) and their lengths */
  uint4 *dealmap;    /* per stretch of TO_DEAL records, two words: {set item, first record, records of the item, first tile},
                        {first bucket unit, first chunk-index entry, item of the batch, 0} - all k_to_deal needs to start */
  const uint32_t *cbase;
  uint4 *tiles;     /* {set item | TO_DIRECT, bucket slot (unit), records, records of the item in the buckets before} */
  uint32_t *biglist; /* set items that are dealt into buckets */
  uint32_t *rel;    /* [n_set][16]: where class (mismatches d, index s) of the guide starts in its hit list, less the
                       records of item s in classes before d: a record's place = rel[2d + s] + its rank in the item */
  uint32_t *nhits;  /* of the batch: hits per guide */
  uint32_t *flags;
  uint32_t *excl;   /* guides (of the batch) that are left out: flags[40] of them */
};
__global__ __launch_bounds__(256) void k_to_fill(gs_to_fill_args a) {
  const uint32_t sb = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = sb < a.n_it;
  const uint32_t g = in ? (a.list ? a.list[sb >> 1] : (sb >> 1)) : 0u;
  const uint32_t item = 2u * g + (sb & 1u);
  const bool left_out = in && to_left_out(a.counts, item);
  const uint32_t c_all = in ? a.counts[item] : 0u, c = left_out ? 0u : c_all;
  uint32_t su;
  const uint32_t nb = to_buckets(c, su), tb = in ? a.tbase[sb] : 0u;
  if (c != 0u && nb == 0u) a.tiles[tb] = make_uint4(sb | TO_DIRECT, 0u, c, 0u);
  /* an item of 513 .. 4,096 records is a tile of the workgroup kernels: a place on its list (one atomic per wave and
   * list - an m <= 5 batch has 10^5 of them) */
  for (uint32_t k = 0; k < 2u; ++k) {
    const bool mine = nb == 0u && c > TO_WTILE && (c > 128u * TO_KPT) == (k == 1u); /* (an item is a tile up to 4,096 records) */
    const unsigned long long m = __ballot(mine);
    if (m == 0ull) continue;
    uint32_t base = 0;
    if (lane_id() == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&a.slow_n[k], (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(m));
    if (mine) a.slow[k][base + lanes_below(m)] = tb;
  }
  if (!in) return;
  if (nb != 0u && nb <= TO_NBMAX) {
    const uint32_t bb = a.bbase[sb];
    for (uint32_t b = 0; b < nb; ++b) a.tiles[tb + b] = make_uint4(sb, bb + b * su, 0u, 0u);
    a.biglist[a.gbase[sb]] = sb;
    const uint32_t d0 = a.dbase[sb], nd = (c + TO_DEAL - 1u) / TO_DEAL;
    const uint32_t cbi = a.cbase[sb];
    for (uint32_t q = 0; q < nd; ++q) {
      a.dealmap[2u * (d0 + q)] = make_uint4(sb, q * TO_DEAL, c, tb);
      a.dealmap[2u * (d0 + q) + 1u] = make_uint4(bb, cbi, item, 0u);
    }
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
    if (p0 != c_all || p1 != cb) atomicOr(a.flags, TO_F_CLS); /* (cannot happen: k_search counts both) */
    a.nhits[g] = c_all + cb;
    if (left_out) a.excl[atomicAdd(a.flags + 40, 1u)] = g; /* the device-wide ordering takes this guide (its hits: as many as its records - the host checks) */
  }
}
/* records of the set (flags[2..3] as one 64-bit count) */
__global__ __launch_bounds__(256) void k_to_total(const uint32_t *counts, const uint32_t *list, uint32_t n_it, unsigned long long *out) {
  unsigned long long v = 0;
  for (uint64_t sb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sb < n_it; sb += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t item = list ? 2u * list[sb >> 1] + ((uint32_t)sb & 1u) : (uint32_t)sb;
    if (!to_left_out(counts, item)) v += counts[item];
  }
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
  if (c >= a.n_used || a.chunk_seq[c] == 0xFFFFFFFFu) return; /* (emptied when a shared item's gaps were closed: gs_search.hip, k_share_fix) */
  const uint32_t item = a.chunk_item[c];
  uint32_t sb = item;
  if (a.by_list) {
    const uint32_t pos = a.redo_pos[item >> 1];
    if (pos == 0xFFFFFFFFu) return;
    sb = 2u * pos + (item & 1u);
  }
  const uint32_t seq = a.chunk_seq[c];
  if (to_left_out(a.counts, item)) return;
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
  const uint4 *dealmap;
  unsigned long long *spl; /* [n_tiles]: entry tbase + b = splitter b of the item */
  uint32_t *slow[3], *slow_n;
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
  /* what a bucket received beyond its slot: the records and their tiles (flags[41] of them, spill_cap at most); a
   * bucket that overflowed moves to spill_unit0 + ... units behind the planned slots (flags[42] units handed out, of
   * spill_units), fill2[tile] = how much of the new slot is written; jobs (flags[43]): {tile, old slot unit, records to copy} */
  uint4 *spill_rec;
  uint32_t *spill_tile, *fill2;
  uint4 *jobs;
  uint32_t spill_cap, spill_unit0, spill_units;
  uint32_t sample_per; /* 0, or GS_TILE_SAMPLE_PER (tests) */
  uint32_t big_from;   /* tiles of more records go to the 1,024-thread kernel: TO_TILE (tests: GS_TILE_BIG_FROM) */
};

/* ---- items beyond one tile: splitters from a sample, then one streaming pass into buckets ----------------------
 * Three kernels.  k_to_splitters: one workgroup per item orders a sample of it and leaves the splitters in memory.
 * k_to_deal: one workgroup per stretch of TO_DEAL records - counts its records per bucket in LDS, reserves that many
 * places in each bucket with one atomic on the bucket's tile descriptor (tiles[].z is the cursor), writes the records.
 * k_to_bucketsum: records of the item before each bucket.  (One workgroup per ITEM doing all of it - the first form -
 * left the chip to 10,482 workgroups on the repeat-rich batch, 46,000 records each and the largest 219,515: 280 us of
 * every item's 336 went into its own stream loop, twenty-two passes of chained LDS reads with three workgroups per
 * CU to hide them, and the kernel ended when the largest item did.) */
__device__ __forceinline__ uint32_t to_sample_per(const gs_to_run_args &a, const uint32_t nb) {
  uint32_t per = TO_SAMPLE / nb < 32u ? TO_SAMPLE / nb : 32u; /* samples per bucket */
  if (a.sample_per != 0u && a.sample_per < per) per = a.sample_per; /* tests: a sample too small to keep the buckets within their slots */
  return per;
}
__device__ __forceinline__ uint32_t to_slots(const gs_to_run_args &a, const uint32_t su) {
  return a.sample_per == 1u ? su * TO_BU / 4u : su * TO_BU; /* tests: at one word per splitter, a quarter of the slots - certain to overflow */
}
/* SNT threads order a sample of up to SNT x 8 words: 1,024 for the large items, 128 for the items whose sample is at most
 * 1,024 words (an m <= 6 batch has 27,000 items of ~5,000 records, 450 sample words each: in workgroups of 1,024, two per
 * CU, their samples took 1.0 ms; 9 KB of LDS instead of 74).  Both are launched over the list, each takes its own. */
template <uint32_t SNT>
__global__ __launch_bounds__(SNT) void k_to_splitters(gs_to_run_args a) {
  __shared__ unsigned long long s_keys[SNT * TO_KPT + SNT];
  __shared__ uint32_t s_nt[32 * 8], s_bs[8];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 32u * 8u; i += SNT) s_nt[i] = (uint32_t)a.tab->n[i >> 3][i & 7u];
  if (tid < 8u) s_bs[tid] = (uint32_t)a.tab->base[tid];
  const uint32_t pam_mul = (uint32_t)a.tab->pam_mul;
  const uint32_t sb = a.biglist[blockIdx.x];
  const uint32_t g = a.list ? a.list[sb >> 1] : (sb >> 1);
  const uint32_t item = 2u * g + (sb & 1u);
  uint32_t su;
  const uint32_t c = a.counts[item], nb = to_buckets(c, su);
  const uint32_t tb = a.tbase[sb], cb = a.cbase[sb];
  /* stratified sample: one record from each of `ns` equal stretches of the item (emission order is seed by seed,
   * so a stretch is a few neighbouring sequences: no worse than independent draws) */
  const uint32_t per = to_sample_per(a, nb), ns = per * nb;
  if ((ns <= 128u * TO_KPT) != (SNT == 128u)) return; /* (workgroup-uniform) */
  __syncthreads();
  {
    uint4 srec[TO_KPT]; /* (ns <= TO_SAMPLE: eight per thread, all asked for before the first is ranked) */
#pragma unroll
    for (uint32_t q = 0; q < TO_KPT; ++q) {
      const uint32_t j = tid + q * SNT;
      srec[q] = make_uint4(0u, 0u, 0u, 0u);
      if (j < ns) {
        const uint32_t lo = (uint32_t)(((unsigned long long)j * c) / ns), hi = (uint32_t)(((unsigned long long)(j + 1u) * c) / ns);
        uint32_t h = j * 2654435761u ^ (sb * 40503u + 0x9E3779B9u);
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        const uint32_t pos = lo + h % (hi - lo); /* hi > lo: c > TO_TILE, ns <= 32 (c / 400 + 1) */
        srec[q] = *to_addr(a.src, item, cb, pos);
      }
    }
#pragma unroll
    for (uint32_t q = 0; q < TO_KPT; ++q) {
      const uint32_t j = tid + q * SNT;
      if (j < ns) s_keys[TO_AT(j)] = to_word(srec[q], a.L, a.P, s_nt, s_bs, pam_mul);
    }
  }
  __syncthreads();
  to_msort<false>(s_keys, nullptr, ns);
  for (uint32_t b = tid; b + 1u < nb; b += SNT) a.spl[tb + b] = s_keys[TO_AT((b + 1u) * per - 1u)];
}

__global__ __launch_bounds__(TO_NT) void k_to_deal(gs_to_run_args a) {
  __shared__ unsigned long long s_spl[TO_NBMAX];
  __shared__ uint32_t s_nt[32 * 8], s_bs[8];
  __shared__ uint32_t s_cnt[TO_NBMAX];
  __shared__ uint32_t s_chunk[4];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u);
  for (uint32_t i = tid; i < 32u * 8u; i += TO_NT) s_nt[i] = (uint32_t)a.tab->n[i >> 3][i & 7u];
  if (tid < 8u) s_bs[tid] = (uint32_t)a.tab->base[tid];
  const uint32_t pam_mul = (uint32_t)a.tab->pam_mul;
  const uint4 d0 = a.dealmap[2u * blockIdx.x], d1 = a.dealmap[2u * blockIdx.x + 1u]; /* (one read, then everything else at once) */
  const uint32_t i_lo = d0.y, c = d0.z, tb = d0.w, bb = d1.x, cb = d1.y, item = d1.z;
  uint32_t su;
  const uint32_t nb = to_buckets(c, su);
  const uint32_t cap = a.src.cap;
  /* the arena chunks the stretch touches (TO_DEAL / ARENA_CHUNK + 1 at most), by number from the first */
  const uint32_t e_lo = i_lo > cap ? (i_lo - cap) >> ARENA_SHIFT : 0u;
  const uint32_t nch = c > cap ? (c - cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT : 0u;
  if (tid < 4u && e_lo + tid < nch) s_chunk[tid] = a.src.chunk_of[cb + e_lo + tid];
  for (uint32_t b = tid; b < nb; b += TO_NT) {
    s_cnt[b] = 0u;
    if (b + 1u < nb) s_spl[b] = a.spl[tb + b];
  }
  const uint32_t slots0 = su * TO_BU, slots = to_slots(a, su);
  const uint32_t nsteps = 32u - (uint32_t)__builtin_clz(nb - 1u); /* probes of a search among nb - 1 splitters (nb >= 2) */
  const uint4 *slots_of = a.src.slots + (size_t)item * cap;
  __syncthreads();
  uint4 rec[4];
  bool on[4];
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) {
    const uint32_t i = i_lo + tid + u * TO_NT;
    on[u] = i < c;
    rec[u] = make_uint4(0u, 0u, 0u, 0u);
    if (on[u])
      rec[u] = i < cap ? slots_of[i] : a.src.arena[((size_t)s_chunk[((i - cap) >> ARENA_SHIFT) - e_lo] << ARENA_SHIFT) | ((i - cap) & (ARENA_CHUNK - 1u))];
  }
  /* the four records go through every stage together: their words, then the splitters below each (a search of
   * `nsteps` probes, the same for all) - four chains of LDS reads side by side */
  unsigned long long K[4];
  to_word4(rec, a.L, a.P, s_nt, s_bs, pam_mul, K);
  uint32_t blo[4], bhi[4], lp[4];
  bool multi = false;
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) {
    multi = multi || (on[u] && rec[u].z != rec[u].w);
    blo[u] = 0u;
    bhi[u] = nb - 1u;
  }
  for (uint32_t st = 0; st < nsteps; ++st) {
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t mid = (blo[u] + bhi[u]) >> 1;
      const bool below = s_spl[mid < nb - 1u ? mid : 0u] < K[u];
      if (blo[u] < bhi[u]) {
        if (below)
          blo[u] = mid + 1u;
        else
          bhi[u] = mid;
      }
    }
  }
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) {
    /* place among the workgroup's records of the bucket: one LDS atomic per wave when its records go to one bucket
     * (the stretches of a long run), else one per lane */
    const unsigned long long act = __ballot(on[u]);
    lp[u] = 0u;
    if (act == 0ull) continue;
    const uint32_t b0 = (uint32_t)__shfl((int)blo[u], (int)__builtin_ctzll(act));
    if (__ballot(on[u] && blo[u] != b0) == 0ull) {
      uint32_t base = 0;
      if (lane == (uint32_t)__builtin_ctzll(act)) base = atomicAdd(&s_cnt[b0], (uint32_t)__popcll(act));
      lp[u] = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(act)) + lanes_below(act);
    } else if (on[u]) {
      lp[u] = atomicAdd(&s_cnt[blo[u]], 1u);
    }
  }
  if (multi) atomicOr(a.flags, TO_F_MULTIROW);
  __syncthreads();
  for (uint32_t b = tid; b < nb; b += TO_NT) { /* that many places in the bucket, from its cursor */
    const uint32_t n = s_cnt[b];
    s_cnt[b] = n ? atomicAdd(&a.tiles[tb + b].z, n) : 0u;
  }
  __syncthreads();
  /* (staged through LDS first, so that neighbouring threads write neighbouring records of a bucket - sixteen in a row on
   * average -, the kernel took 5.2 ms instead of 4.2: three more barriers, and 256 bytes are still a short row) */
  uint4 *out = a.buckets + (size_t)bb * TO_BU;
#pragma unroll
  for (uint32_t u = 0; u < 4u; ++u) {
    const uint32_t pos = s_cnt[blo[u]] + lp[u];
    /* a bucket record carries its sequence word where the (equal) last row was: the tile does not rank it again */
    const uint4 br = make_uint4(rec[u].x, rec[u].y, rec[u].z, (uint32_t)(K[u] >> 32));
    if (on[u] && pos < slots) {
      out[(size_t)blo[u] * slots0 + pos] = br;
    } else if (on[u]) { /* beyond the bucket's slot: to the spill list (k_to_respill gives the bucket a slot of its real size) */
      const uint32_t e = atomicAdd(a.flags + 41, 1u);
      if (e < a.spill_cap) {
        a.spill_rec[e] = br;
        a.spill_tile[e] = tb + blo[u];
      } else {
        atomicOr(a.flags, TO_F_BUCKET);
      }
    }
  }
}

/* records per bucket (cut to the slot; more raise TO_F_BUCKET) and before it: one wave per item, nb <= 1,024 = 16 per lane */
__global__ __launch_bounds__(256) void k_to_bucketsum(gs_to_run_args a, const uint32_t n_big) {
  const uint32_t w = (blockIdx.x * 256u + threadIdx.x) / WAVE, lane = threadIdx.x & (WAVE - 1u);
  if (w >= n_big) return;
  const uint32_t sb = a.biglist[w];
  const uint32_t g = a.list ? a.list[sb >> 1] : (sb >> 1);
  uint32_t su;
  const uint32_t nb = to_buckets(a.counts[2u * g + (sb & 1u)], su);
  const uint32_t tb = a.tbase[sb], slots = to_slots(a, su);
  const uint32_t each = (nb + WAVE - 1u) / WAVE;
  uint32_t s = 0, cl[3] = {0u, 0u, 0u};
  bool over = false;
  for (uint32_t q = 0; q < each; ++q) {
    const uint32_t b = lane * each + q;
    if (b < nb) s += a.tiles[tb + b].z;
  }
  uint32_t run = wave_incl_sum(s) - s;
  for (uint32_t q = 0; q < each; ++q) {
    const uint32_t b = lane * each + q;
    if (b < nb) {
      const uint32_t n = a.tiles[tb + b].z;
      a.tiles[tb + b].w = run;
      run += n;
      if (n > slots) {
        /* the bucket outgrew its slot: a slot of its real size behind the planned ones; what the old slot holds is
         * copied there and the spilled records follow (k_to_respill) */
        const uint32_t units = (n + TO_BU - 1u) / TO_BU, u0 = atomicAdd(a.flags + 42, units);
        if (u0 + units <= a.spill_units && n <= 1024u * TO_KPT) {
          a.jobs[atomicAdd(a.flags + 43, 1u)] = make_uint4(tb + b, a.tiles[tb + b].y, slots, 0u);
          a.tiles[tb + b].y = a.spill_unit0 + u0;
          a.fill2[tb + b] = slots;
        } else {
          over = true; /* (no room left, or beyond what one workgroup orders: the caller orders the batch the other way) */
          a.tiles[tb + b].z = slots;
        }
      }
    }
    /* (one bucket in fifteen goes on a list of the workgroup kernels: counted here, placed below) */
    const uint32_t n2 = b < nb ? a.tiles[tb + b].z : 0u;
    if (n2 > TO_WTILE) cl[n2 > 128u * TO_KPT ? (n2 > a.big_from ? 2u : 1u) : 0u]++;
  }
  /* a place on its list for every such bucket: ONE atomic per wave and list (three words serve every item of the batch: one
   * atomic per bucket round and list - 50,000 on the repeat-rich batch - was 0.2 ms of waiting for them) */
  uint32_t at[3];
#pragma unroll
  for (uint32_t k = 0; k < 3u; ++k) {
    const uint32_t incl = wave_incl_sum(cl[k]);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    uint32_t base = 0;
    if (tot != 0u) {
      if (lane == 0u) base = atomicAdd(&a.slow_n[k], tot);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    }
    at[k] = base + incl - cl[k];
  }
  for (uint32_t q = 0; q < each; ++q) {
    const uint32_t b = lane * each + q;
    const uint32_t n2 = b < nb ? a.tiles[tb + b].z : 0u;
    if (n2 > TO_WTILE) {
      const uint32_t k = n2 > 128u * TO_KPT ? (n2 > a.big_from ? 2u : 1u) : 0u;
      const uint32_t pos = k == 0u ? at[0]++ : k == 1u ? at[1]++ : at[2]++;
      uint32_t *const dst = k == 0u ? a.slow[0] : k == 1u ? a.slow[1] : a.slow[2];
      dst[pos] = tb + b;
    }
  }
  if (over) atomicOr(a.flags, TO_F_BUCKET);
}

/* buckets that outgrew their slots: (0) what the old slot holds goes to the new one, a workgroup per bucket and visit;
 * (1) the spilled records follow, each to the next free place of its bucket's new slot */
__global__ __launch_bounds__(256) void k_to_respill(gs_to_run_args a, const uint32_t phase) {
  if (phase == 0u) {
    const uint32_t nj = a.flags[43];
    for (uint32_t j = blockIdx.x; j < nj; j += gridDim.x) {
      const uint4 job = a.jobs[j];
      const uint4 *src = a.buckets + (size_t)job.y * TO_BU;
      uint4 *dst = a.buckets + (size_t)a.tiles[job.x].y * TO_BU;
      for (uint32_t i = threadIdx.x; i < job.z; i += 256u) dst[i] = src[i];
    }
    return;
  }
  const uint32_t ne = a.flags[41] < a.spill_cap ? a.flags[41] : a.spill_cap;
  for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < ne; e += gridDim.x * 256u) {
    const uint32_t t = a.spill_tile[e];
    const uint4 tile = a.tiles[t];
    const uint32_t pos = atomicAdd(&a.fill2[t], 1u);
    if (pos < tile.z) a.buckets[(size_t)tile.y * TO_BU + pos] = a.spill_rec[e];
  }
}

/* ---- one tile: order its records in LDS and write its hits -------------------------------------------------------- */
/* NT threads, tiles of NT x 8 records at most: 512 for the tiles proper, 128 for the many small ones (an m <= 5 batch is
 * 2 x 10^5 items of ~700 records: in a workgroup of 512 a merge round keeps 90 threads busy and seven waves wait at its
 * barriers; 12 KB of LDS instead of 47: twelve workgroups per CU).  Since k_to_wsort they take the tiles of more than 512
 * records only, each from its list (k_to_fill and k_to_bucketsum write them; launched over all tiles to pick their own,
 * 1.7 x 10^6 workgroups that leave at once cost 1.5 ms). */
static constexpr size_t to_sort_lds(uint32_t nt) { /* words + s_first, tables and class starts, 16-bit places */
  return (size_t)(nt * TO_KPT + nt) * 8u + 8u + (32u * 8u + 8u + 8u) * 4u + (size_t)(nt * TO_KPT + nt) * 2u;
}
template <uint32_t NT>
__global__ __launch_bounds__(NT) void k_to_sort(gs_to_run_args a, const uint32_t *list) {
  constexpr uint32_t CAP = NT * TO_KPT, LDSN = CAP + CAP / 8u;
  /* dynamic LDS (to_sort_lds(NT) bytes; the 1,024-thread form needs 94 KB - beyond what a kernel may declare) */
  extern __shared__ unsigned long long to_sort_smem[];
  unsigned long long *s_keys = to_sort_smem;
  unsigned long long &s_first = to_sort_smem[LDSN];
  uint32_t *s_nt = (uint32_t *)(to_sort_smem + LDSN + 1u), *s_bs = s_nt + 32u * 8u, *s_rel = s_bs + 8u;
  uint16_t *s_idx = (uint16_t *)(s_rel + 8u);

  const uint32_t tid = threadIdx.x;
  /* one workgroup per tile (persistent workgroups looping over the tiles were tried: the loop took the kernel from 71
   * to 156 registers, one workgroup per CU instead of three, 22 ms instead of 9) */
  const uint4 t = a.tiles[list[blockIdx.x]];
  const uint32_t n = t.z;
  if (n > CAP) return; /* (cannot happen: the lists are made by size) */
#ifdef TO_PROFILE
  const unsigned long long tp0 = wall_clock64();
#endif
  const bool direct = (t.x & TO_DIRECT) != 0u;
  const uint32_t sb = t.x & ~TO_DIRECT;
  const uint32_t gset = sb >> 1, strand = sb & 1u;
  const uint32_t g = a.list ? a.list[gset] : gset;
  const uint32_t item = 2u * g + strand;
  const uint4 *bucket = a.buckets + (size_t)t.y * TO_BU;
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
      rec[u] = i < n ? bucket[i] : make_uint4(0u, 0u, 0u, 0u); /* {key, row, sequence word}: k_to_deal ranked it */
    }
  }
  if (tid < 8u) s_rel[tid] = a.rel[(size_t)gset * 16u + 2u * tid + strand];
  if (tid == 8u) s_first = a.offsets[g] + t.w;
  bool multi = false;
  uint2 kx[TO_KPT]; /* the records' keys, in load order: they wait in registers while the words are ordered */
  if (direct) {
    for (uint32_t i = tid; i < 32u * 8u; i += NT) s_nt[i] = (uint32_t)a.tab->n[i >> 3][i & 7u];
    if (tid < 8u) s_bs[tid] = (uint32_t)a.tab->base[tid];
    const uint32_t pam_mul = (uint32_t)a.tab->pam_mul;
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

/* ---- one tile of at most 512 records per WAVE: ordered in registers ------------------------------------------------
 * Lane l holds places 8l .. 8l+7 of the tile (word, key: four registers per record, 32 in all).  A bitonic network
 * over the 512 places: partners whose places differ in bits 0..2 are the lane's own registers, partners whose places
 * differ in bit 3 + b sit in lane l ^ 2^b and arrive by DPP (b = 0, 1, 3: quad_perm, row_ror:8; b = 2: row_shl:4 and
 * row_shr:4 under bank masks), ds_swizzle (b = 4) or ds_bpermute (b = 5).  21 of the 45 steps cross lanes, eight
 * independent records each - no LDS array, no barrier, no read that waits for a compare: the workgroup form above
 * spends its time in the merge rounds' chains of dependent LDS reads (nine rounds of up to twenty) with three tiles
 * resident per CU; this one keeps eight tiles per SIMD going.  Padding places hold words of all ones. */
template <uint32_t D>
__device__ __forceinline__ uint32_t to_xl(const uint32_t v) { /* v of lane ^ D */
  if constexpr (D == 1u)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); /* quad_perm [1,0,3,2] */
  else if constexpr (D == 2u)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); /* quad_perm [2,3,0,1] */
  else if constexpr (D == 4u) {
    const int r = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);   /* row_shl:4 into lanes 0-3, 8-11 of a row */
    return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xF, 0xA, false); /* row_shr:4 into lanes 4-7, 12-15 */
  } else if constexpr (D == 8u)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true); /* row_ror:8 */
  else if constexpr (D == 16u)
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); /* bit mode: and 0x1F, or 0, xor 0x10 */
  else
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane_id() ^ 32u) << 2), (int)v);
}
struct to_wregs {
  unsigned long long K[8];
  uint32_t x[8], y[8];
};
/* places i < j of the lane: the smaller word to i when asc, the larger when not (one compare and a mask flip: equal
 * words then trade places when not asc, which changes nothing) */
#define TO_WCE(i, j)                                              \
  {                                                               \
    const bool sw = (r.K[i] > r.K[j]) != desc;                    \
    const unsigned long long tk = sw ? r.K[j] : r.K[i];           \
    r.K[j] = sw ? r.K[i] : r.K[j];                                \
    r.K[i] = tk;                                                  \
    if constexpr (!PACKED) { /* (the record's place in load order travels with its word) */ \
      const uint32_t tx = sw ? r.x[j] : r.x[i];                   \
      r.x[j] = sw ? r.x[i] : r.x[j];                              \
      r.x[i] = tx;                                                \
    }                                                             \
  }
template <uint32_t D, bool PACKED>
__device__ __forceinline__ void to_wcross(to_wregs &r, const bool asc) {
  /* the lane keeps the smaller of the two words when it is the lower lane of an ascending pair (or the upper of a
   * descending one).  One compare: of two EQUAL words the lane that keeps the larger takes its partner's record and the
   * other keeps its own - one record twice, one lost.  Equal words are padding (identical records) or the duplicate
   * (sequence, row) that TO_F_DUP reports from the words, which stay what they were: the batch is then ordered again. */
  const bool keep_max = ((lane_id() & D) == 0u) != asc;
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) {
    const uint32_t pl = to_xl<D>((uint32_t)r.K[u]), ph = to_xl<D>((uint32_t)(r.K[u] >> 32));
    const unsigned long long pK = ((unsigned long long)ph << 32) | pl;
    const bool take = (pK < r.K[u]) != keep_max;
    if constexpr (!PACKED) {
      const uint32_t px = to_xl<D>(r.x[u]);
      r.x[u] = take ? px : r.x[u];
    }
    r.K[u] = take ? pK : r.K[u];
  }
}
template <bool PACKED>
__device__ __forceinline__ void to_wlocal(to_wregs &r, const bool asc) { /* the steps at distances 4, 2, 1 */
  const bool desc = !asc;
  TO_WCE(0, 4) TO_WCE(1, 5) TO_WCE(2, 6) TO_WCE(3, 7)
  TO_WCE(0, 2) TO_WCE(1, 3) TO_WCE(4, 6) TO_WCE(5, 7)
  TO_WCE(0, 1) TO_WCE(2, 3) TO_WCE(4, 5) TO_WCE(6, 7)
}
template <bool PACKED>
__device__ __forceinline__ void to_wsort(to_wregs &r) {
  const uint32_t lane = lane_id();
  {
    const bool desc = (lane & 1u) != 0u; /* the lane's eight: Batcher's network, up in even lanes and down in odd ones */
    TO_WCE(0, 1) TO_WCE(2, 3) TO_WCE(4, 5) TO_WCE(6, 7)
    TO_WCE(0, 2) TO_WCE(1, 3) TO_WCE(4, 6) TO_WCE(5, 7)
    TO_WCE(1, 2) TO_WCE(5, 6)
    TO_WCE(0, 4) TO_WCE(1, 5) TO_WCE(2, 6) TO_WCE(3, 7)
    TO_WCE(2, 4) TO_WCE(3, 5)
    TO_WCE(1, 2) TO_WCE(3, 4) TO_WCE(5, 6)
  }
  bool asc = (lane & 2u) == 0u; /* runs of 16 */
  to_wcross<1u, PACKED>(r, asc);
  to_wlocal<PACKED>(r, asc);
  asc = (lane & 4u) == 0u; /* 32 */
  to_wcross<2u, PACKED>(r, asc);
  to_wcross<1u, PACKED>(r, asc);
  to_wlocal<PACKED>(r, asc);
  asc = (lane & 8u) == 0u; /* 64 */
  to_wcross<4u, PACKED>(r, asc);
  to_wcross<2u, PACKED>(r, asc);
  to_wcross<1u, PACKED>(r, asc);
  to_wlocal<PACKED>(r, asc);
  asc = (lane & 16u) == 0u; /* 128 */
  to_wcross<8u, PACKED>(r, asc);
  to_wcross<4u, PACKED>(r, asc);
  to_wcross<2u, PACKED>(r, asc);
  to_wcross<1u, PACKED>(r, asc);
  to_wlocal<PACKED>(r, asc);
  asc = (lane & 32u) == 0u; /* 256 */
  to_wcross<16u, PACKED>(r, asc);
  to_wcross<8u, PACKED>(r, asc);
  to_wcross<4u, PACKED>(r, asc);
  to_wcross<2u, PACKED>(r, asc);
  to_wcross<1u, PACKED>(r, asc);
  to_wlocal<PACKED>(r, asc);
  asc = true; /* 512 */
  to_wcross<32u, PACKED>(r, asc);
  to_wcross<16u, PACKED>(r, asc);
  to_wcross<8u, PACKED>(r, asc);
  to_wcross<4u, PACKED>(r, asc);
  to_wcross<2u, PACKED>(r, asc);
  to_wcross<1u, PACKED>(r, asc);
  to_wlocal<PACKED>(r, asc);
}
#define TO_WNW 4u /* waves (tiles) per workgroup */
#define TO_WPACK_BITS 23u /* PACKED: sequence words below 2^23 (20-mers + NGG to three mismatches: 4.07 x 10^6) */
/* The keys wait in LDS (4 KB per wave) while their words are ordered and are picked up by the record's place in load
 * order.  PACKED: word << 41 | row << 9 | that place - nothing travels with the 64 bits: five instructions per record and
 * crossing step (two moves, a compare, two selects).  Otherwise the place is a third register: seven (with the key's
 * two words travelling instead, the first form: nine - 7.9 ms against 6.3 on the repeat-rich batch). */
template <bool PACKED>
__global__ __launch_bounds__(TO_WNW *WAVE) void k_to_wsort(gs_to_run_args a, const uint32_t n_tiles) {
  __shared__ uint32_t s_nt[32 * 8], s_bs[8];
  __shared__ uint2 s_key[TO_WNW * TO_WTILE]; /* 4 KB per wave: the keys while their words are ordered, then the way to the stores' layout */
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u);
  const uint32_t tile = blockIdx.x * TO_WNW + tid / WAVE;
  uint4 t = make_uint4(0u, 0u, 0u, 0u);
  if (tile < n_tiles) t = a.tiles[tile];
  for (uint32_t i = tid; i < 32u * 8u; i += TO_WNW * WAVE) s_nt[i] = (uint32_t)a.tab->n[i >> 3][i & 7u];
  if (tid < 8u) s_bs[tid] = (uint32_t)a.tab->base[tid];
  const uint32_t pam_mul = (uint32_t)a.tab->pam_mul;
  __syncthreads(); /* (the only one: from here on each wave is by itself) */
  const uint32_t n = t.z;
  if (n == 0u || n > TO_WTILE) return;
  const bool direct = (t.x & TO_DIRECT) != 0u;
  const uint32_t sb = t.x & ~TO_DIRECT;
  const uint32_t gset = sb >> 1, strand = sb & 1u;
  const uint32_t g = a.list ? a.list[gset] : gset;
  uint4 rec[8];
  if (direct) {
    const uint32_t item = 2u * g + strand, cb = a.cbase[sb];
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
      const uint32_t i = lane + u * WAVE;
      rec[u] = i < n ? *to_addr(a.src, item, cb, i) : make_uint4(0u, 0u, 0u, 0u);
    }
  } else {
    const uint4 *bucket = a.buckets + (size_t)t.y * TO_BU;
#pragma unroll
    for (uint32_t u = 0; u < 8u; ++u) {
      const uint32_t i = lane + u * WAVE;
      rec[u] = i < n ? bucket[i] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  const uint32_t relv = lane < 8u ? a.rel[(size_t)gset * 16u + 2u * lane + strand] : 0u;
  const uint64_t first = a.offsets[g] + t.w;
  uint2 *keys = s_key + (tid / WAVE) * TO_WTILE;
  to_wregs r;
  bool multi = false;
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) {
    const uint32_t i = lane + u * WAVE;
    const bool on = i < n;
    r.x[u] = PACKED ? rec[u].x : i; /* (x unused when PACKED, y always, until the keys come back - but left unset, the
                                       compiler's allocation of the kernel goes from 62 registers to 82) */
    r.y[u] = rec[u].y;
    if (direct) {
      multi = multi || (on && rec[u].z != rec[u].w);
      r.K[u] = on ? to_word(rec[u], a.L, a.P, s_nt, s_bs, pam_mul) : ~0ull;
    } else {
      r.K[u] = on ? ((unsigned long long)rec[u].w << 32) | rec[u].z : ~0ull; /* k_to_deal ranked it */
    }
    keys[i] = make_uint2(rec[u].x, rec[u].y);
    if constexpr (PACKED) {
      if (on) r.K[u] = ((r.K[u] >> 32) << 41) | ((r.K[u] & 0xFFFFFFFFull) << 9) | i;
    }
  }
  if (multi) atomicOr(a.flags, TO_F_MULTIROW);
  to_wsort<PACKED>(r);
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) { /* (the ordered records, in registers, before anything below starts: left to itself the compiler
                                         sinks the network's last exchanges into the code that follows and keeps both sides of each) */
    if constexpr (PACKED)
      asm volatile("" : "+v"(r.K[u]));
    else
      asm volatile("" : "+v"(r.K[u]), "+v"(r.x[u]));
  }
  /* place 8 lane + u of the tile is in register u of lane `lane` */
  constexpr uint32_t SH = PACKED ? 9u : 0u; /* word and row from here up */
  const uint32_t p_lo = (uint32_t)__shfl_up((int)(uint32_t)r.K[7], 1), p_hi = (uint32_t)__shfl_up((int)(uint32_t)(r.K[7] >> 32), 1);
  bool dup = lane != 0u && 8u * lane < n && ((((unsigned long long)p_hi << 32) | p_lo) >> SH) == (r.K[0] >> SH);
  uint32_t row[8];
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) {
    if (u != 0u) dup = dup || (8u * lane + u < n && (r.K[u] >> SH) == (r.K[u - 1u] >> SH));
    row[u] = (uint32_t)(r.K[u] >> SH);
    const uint2 k = keys[(PACKED ? (uint32_t)r.K[u] : r.x[u]) & (TO_WTILE - 1u)];
    r.x[u] = k.x;
    r.y[u] = k.y;
  }
  /* ... and goes to register q >> 6 of lane q & 63, through the wave's 4 KB of LDS (rows and low key words, then the
   * high words), so that a store instruction writes 64 neighbouring hits: with each lane writing its own eight - 64
   * sixteen-byte pieces 128 bytes apart per instruction - the kernel took 6.3 ms instead of 4.6 */
  uint32_t *T = (uint32_t *)keys;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  *(uint4 *)(T + 8u * lane) = make_uint4(row[0], row[1], row[2], row[3]);
  *(uint4 *)(T + 8u * lane + 4u) = make_uint4(row[4], row[5], row[6], row[7]);
  *(uint4 *)(T + TO_WTILE + 8u * lane) = make_uint4(r.x[0], r.x[1], r.x[2], r.x[3]);
  *(uint4 *)(T + TO_WTILE + 8u * lane + 4u) = make_uint4(r.x[4], r.x[5], r.x[6], r.x[7]);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) {
    row[u] = T[u * WAVE + lane];
    r.x[u] = T[TO_WTILE + u * WAVE + lane];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  *(uint4 *)(T + 8u * lane) = make_uint4(r.y[0], r.y[1], r.y[2], r.y[3]);
  *(uint4 *)(T + 8u * lane + 4u) = make_uint4(r.y[4], r.y[5], r.y[6], r.y[7]);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) r.y[u] = T[u * WAVE + lane];
  const uint32_t *sa = a.sa[strand];
  gs_hit *out = a.hits + first + lane;
  uint32_t sav[8];
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) sav[u] = sa[u * WAVE + lane < n ? row[u] : 0u]; /* eight gathers in flight (padding reads entry 0) */
#pragma unroll
  for (uint32_t u = 0; u < 8u; ++u) {
    const uint32_t rel = (uint32_t)__shfl((int)relv, (int)(r.y[u] >> 29));
    const uint64_t sap = (uint64_t)sav[u] - ((r.x[u] & 1u) ? a.v_rem : 0u);
    gs_hit o;
    o.pos = strand == 0u ? -(int64_t)sap : (int64_t)(a.genome_length - (sap + 1ull));
    o.key = (((unsigned long long)r.y[u] << 32) | r.x[u]) & ~1ull;
    if (u * WAVE + lane < n) out[rel + u * WAVE] = o;
  }
  if (dup) atomicOr(a.flags, TO_F_DUP);
}

/* ---- host side ------------------------------------------------------------------------------------------------ */
static bool to_make_tab(uint32_t L, uint32_t P, uint32_t m, gs_to_tab &tab, unsigned long long *words = nullptr) {
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
  if (words) *words = c64; /* (exact when the function returns true) */
  return L >= 1 && 2 * L + 3 * P <= 59 && m <= 7 && cum < 4294967295.0L; /* the words stay below 2^32 - 1: all ones marks padding */
}
extern "C" void gs_debug_tile_plan(uint32_t records, uint32_t out[5]) {
  uint32_t su = 0;
  const uint32_t nb = to_buckets(records, su);
  out[0] = nb;
  out[1] = su * TO_BU;
  out[2] = nb ? (TO_SAMPLE / nb < 32u ? TO_SAMPLE / nb : 32u) : 0u;
  out[3] = TO_WTILE;
  out[4] = TO_NBMAX;
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
  if ((rc = gs_reserve(ix->w_t_plan, 5 * 4 * ((size_t)n_it + 1) + 64)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_tab, sizeof(gs_to_tab) + 192)) != GS_OK) return rc; /* the table, flags (128 bytes), list lengths */
  if ((rc = gs_reserve(ix->w_t_rel, 64 * ((size_t)in.n_set + 1))) != GS_OK) return rc;
  uint32_t *tbase = (uint32_t *)ix->w_t_plan.p, *bbase = tbase + (n_it + 1), *cbase = bbase + (n_it + 1), *gbase = cbase + (n_it + 1);
  uint32_t *dbase = gbase + (n_it + 1);
  uint32_t *d_flags = (uint32_t *)((char *)ix->w_t_tab.p + sizeof(gs_to_tab));
  GS_HIP(hipMemcpyAsync(ix->w_t_tab.p, &tab, sizeof(tab), hipMemcpyHostToDevice, st));
  GS_HIP(hipMemsetAsync(d_flags, 0, 192, st));
  gs_to_plan_args pa;
  pa.counts = in.counts;
  pa.list = in.list;
  pa.n_it = n_it;
  pa.cap = in.cap;
  pa.tbase = tbase;
  pa.bbase = bbase;
  pa.cbase = cbase;
  pa.gbase = gbase;
  pa.dbase = dbase;
  pa.flags = d_flags;
  hipLaunchKernelGGL(k_to_plan, dim3(1), dim3(1024), 0, st, pa);
  uint32_t tot[5] = {0, 0, 0, 0, 0}, h_flags = 0;
  GS_HIP(hipMemcpyAsync(&tot[0], tbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&tot[1], bbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&tot[2], cbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&tot[3], gbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&tot[4], dbase + n_it, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&h_flags, d_flags, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st)); /* (tab is a local, too) */
  if (h_flags) return GS_OK;
  S.n_tiles = tot[0];
  S.n_btiles = tot[1];
  S.n_chunks = tot[2];
  S.n_big = tot[3];
  S.n_deal = tot[4];
  if ((rc = gs_reserve(ix->w_t_tiles, (16 + 8) * ((size_t)S.n_tiles + 1))) != GS_OK) return rc; /* descriptors, then the splitters */
  /* bucket space: the planned slots, an eighth more for the buckets that outgrow theirs (k_to_respill), the spill list
   * (an eighth of the dealt records: 20 bytes each), a word per tile and a job per moved bucket */
  S.spill_units = S.n_btiles / 8u + 2048u;
  S.spill_cap = S.n_deal * (TO_DEAL / 8u) + 65536u;
  if (gs_opt(ix, "GS_TILE_NO_SPILL")) S.spill_units = S.spill_cap = 0u; /* (tests: a bucket beyond its slot then sends the batch to the device-wide ordering) */
  if ((rc = gs_reserve(ix->w_t_buckets, 16 * (size_t)TO_BU * ((size_t)S.n_btiles + S.spill_units) + 16)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_spill, 20 * (size_t)S.spill_cap + (4 + 16) * ((size_t)S.n_tiles + 1) + 64)) != GS_OK) return rc;
  if ((rc = gs_reserve(ix->w_t_chunkof, 4 * ((size_t)S.n_chunks + 1) + 12 * ((size_t)S.n_tiles + 1))) != GS_OK) return rc; /* the chunk index, then the three lists */
  const size_t deal_at = (4 * ((size_t)S.n_big + 1) + 31) & ~(size_t)31; /* the list, then the deal map */
  if ((rc = gs_reserve(ix->w_t_big, deal_at + 32 * ((size_t)S.n_deal + 1))) != GS_OK) return rc;
  gs_to_fill_args fa;
  fa.counts = in.counts;
  fa.cls = in.cls;
  fa.list = in.list;
  fa.n_it = n_it;
  fa.cap = in.cap;
  fa.tbase = tbase;
  fa.bbase = bbase;
  fa.gbase = gbase;
  fa.dbase = dbase;
  fa.dealmap = (uint4 *)((char *)ix->w_t_big.p + deal_at);
  fa.slow[0] = (uint32_t *)ix->w_t_chunkof.p + (S.n_chunks + 1);
  fa.slow[1] = fa.slow[0] + (S.n_tiles + 1);
  fa.slow[2] = fa.slow[1] + (S.n_tiles + 1);
  fa.slow_n = d_flags + 32;
  fa.cbase = cbase;
  fa.tiles = (uint4 *)ix->w_t_tiles.p;
  fa.biglist = (uint32_t *)ix->w_t_big.p;
  fa.rel = (uint32_t *)ix->w_t_rel.p;
  fa.nhits = in.nhits;
  fa.flags = d_flags;
  if ((rc = gs_reserve(ix->w_t_excl, 4 * ((size_t)in.n_set + 1))) != GS_OK) return rc;
  fa.excl = (uint32_t *)ix->w_t_excl.p;
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
  ra.dealmap = (const uint4 *)((const char *)ix->w_t_big.p + ((4 * ((size_t)S.n_big + 1) + 31) & ~(size_t)31));
  ra.spl = (unsigned long long *)((char *)ix->w_t_tiles.p + 16 * ((size_t)S.n_tiles + 1));
  ra.slow[0] = (uint32_t *)ix->w_t_chunkof.p + (S.n_chunks + 1);
  ra.slow[1] = ra.slow[0] + (S.n_tiles + 1);
  ra.slow[2] = ra.slow[1] + (S.n_tiles + 1);
  ra.slow_n = d_flags + 32;
  ra.tiles = (uint4 *)ix->w_t_tiles.p;
  ra.buckets = (uint4 *)ix->w_t_buckets.p;
  ra.spill_rec = (uint4 *)ix->w_t_spill.p;
  ra.jobs = ra.spill_rec + S.spill_cap;
  ra.spill_tile = (uint32_t *)(ra.jobs + (S.n_tiles + 1));
  ra.fill2 = ra.spill_tile + S.spill_cap;
  ra.spill_cap = S.spill_cap;
  ra.spill_unit0 = S.n_btiles;
  ra.spill_units = S.spill_units;
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
  ra.sample_per = gs_opt(ix, "GS_TILE_SAMPLE_PER") ? (uint32_t)std::max(1l, atol(gs_opt(ix, "GS_TILE_SAMPLE_PER"))) : 0u;
  ra.big_from = gs_opt(ix, "GS_TILE_BIG_FROM") ? (uint32_t)std::max(1024l, atol(gs_opt(ix, "GS_TILE_BIG_FROM"))) : TO_TILE;
  if (S.n_big) {
    hipLaunchKernelGGL(k_to_splitters<128u>, dim3(S.n_big), dim3(128), 0, st, ra);
    hipLaunchKernelGGL(k_to_splitters<TO_SNT>, dim3(S.n_big), dim3(TO_SNT), 0, st, ra);
    hipLaunchKernelGGL(k_to_deal, dim3(S.n_deal), dim3(TO_NT), 0, st, ra);
    hipLaunchKernelGGL(k_to_bucketsum, dim3((S.n_big + 3u) / 4u), dim3(256), 0, st, ra, S.n_big);
    if (S.spill_cap != 0u && S.n_tiles != 0u) {
      hipLaunchKernelGGL(k_to_respill, dim3(std::min<uint32_t>(S.n_tiles, 2048u)), dim3(256), 0, st, ra, 0u);
      hipLaunchKernelGGL(k_to_respill, dim3(std::min<uint32_t>((S.spill_cap + 255u) / 256u, 2048u)), dim3(256), 0, st, ra, 1u);
    }
  }
  if (S.n_tiles) {
    /* the tiles one wave cannot take: how many is known once the buckets are counted - the host waits for that number
     * alone (an event behind its copy) while k_to_wsort runs */
    /* (the event and the page-locked words the copy lands in belong to the handle) */
    if (!ix->ev_tile) GS_HIP(hipEventCreateWithFlags(&ix->ev_tile, hipEventDisableTiming));
    if (!ix->h_pin) GS_HIP(hipHostMalloc((void **)&ix->h_pin, 256, hipHostMallocDefault));
    uint32_t *n_slow = ix->h_pin;
    n_slow[0] = n_slow[1] = n_slow[2] = 0u;
    GS_HIP(hipMemcpyAsync(n_slow, ra.slow_n, 12, hipMemcpyDeviceToHost, st));
    GS_HIP(hipEventRecord(ix->ev_tile, st));
    gs_to_tab tab;
    unsigned long long words = ~0ull;
    to_make_tab(in.L, in.P, in.m, tab, &words);
    if (words < (1ull << TO_WPACK_BITS) && !gs_opt(ix, "GS_TILE_NO_PACK"))
      hipLaunchKernelGGL(k_to_wsort<true>, dim3((S.n_tiles + TO_WNW - 1u) / TO_WNW), dim3(TO_WNW * WAVE), 0, st, ra, S.n_tiles);
    else
      hipLaunchKernelGGL(k_to_wsort<false>, dim3((S.n_tiles + TO_WNW - 1u) / TO_WNW), dim3(TO_WNW * WAVE), 0, st, ra, S.n_tiles);
    GS_HIP(hipEventSynchronize(ix->ev_tile));
    if (gs_opt(ix, "GS_DEBUG"))
      fprintf(stderr, "[gs] tile ordering: tiles beyond one wave: %u of up to 1024 records, %u of up to 4096, %u beyond\n", n_slow[0], n_slow[1], n_slow[2]);
    if (n_slow[0]) hipLaunchKernelGGL(k_to_sort<128u>, dim3(n_slow[0]), dim3(128), to_sort_lds(128u), st, ra, (const uint32_t *)ra.slow[0]);
    if (n_slow[1]) hipLaunchKernelGGL(k_to_sort<TO_NT>, dim3(n_slow[1]), dim3(TO_NT), to_sort_lds(TO_NT), st, ra, (const uint32_t *)ra.slow[1]);
    if (n_slow[2]) { /* buckets of an item beyond 3 x 10^5 records (they aim at 1,024) that came out beyond 4,096 */
      GS_HIP(hipFuncSetAttribute((const void *)k_to_sort<1024u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)to_sort_lds(1024u)));
      hipLaunchKernelGGL(k_to_sort<1024u>, dim3(n_slow[2]), dim3(1024), to_sort_lds(1024u), st, ra, (const uint32_t *)ra.slow[2]);
    }
  }
  uint32_t h[48] = {0};
  GS_HIP(hipMemcpyAsync(h, d_flags, sizeof(h), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
#ifdef TO_PROFILE
  {
    const unsigned long long *d = (const unsigned long long *)(h + 4);
    fprintf(stderr, "[gs] k_to_sort phases (wall-clock ticks per tile, 100 MHz): load %.0f, order %.0f, write %.0f; %llu tiles, %.0f records each\n",
            (double)d[0] / d[3], (double)d[1] / d[3], (double)d[2] / d[3], d[3], (double)d[4] / d[3]);
  }
#endif
  if (gs_opt(ix, "GS_DEBUG"))
    fprintf(stderr, "[gs] tile ordering: %u items, %u of them dealt into %u bucket units of 128 records, %u tiles; %u records beyond their bucket's slot, "
            "%u buckets moved to %u units of the %u kept for that\n", S.n_it, S.n_big, S.n_btiles, S.n_tiles, h[41], h[43], h[42], S.spill_units);
  *violations = h[0];
  S.n_records = ((uint64_t)h[3] << 32) | h[2];
  S.n_excl = h[40]; /* guides left to the device-wide ordering (ix->w_t_excl) */
  GS_HIP(hipGetLastError());
  return GS_OK;
}
