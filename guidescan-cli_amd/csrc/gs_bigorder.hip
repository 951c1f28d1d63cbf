/*
 * gs_bigorder.hip -- the device-wide ordering of match records: what orders a batch the per-guide tile ordering
 * (gs_tileorder.hip) does not take.  Kernels only; the host side is big_order / big_locate in gs_enumerate.hip.
 */
#include "gs_kernels.h"

/* ---- guides with more matches than an LDS sort can hold: repeat-derived guides at any budget,
 * every guide at <= 6 mismatches on a genome of this size (~5,400 matches per item).  Their match
 * records are compacted into one array (item order = guide order), ordered by device-wide radix
 * sorts of one 64-bit word per record (below), made unique, scanned, and located one thread per record.
 * No per-guide atomics, no comparator sort.  (Round 2's form - the raw key's bits in two words, thirteen
 * passes - served sort words beyond 64 bits until round 5: no shape the path accepts has one below
 * 2^24 guides per set, and such a batch is refused with "use smaller batches".) ---- */
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
