/*
 * gs_order.hip -- what follows k_search in a batch: per-guide canonical order + dedupe in LDS (k_order), hits per guide ->
 * CSR offsets (k_scan_*), the suffix array gather + coordinate rule (k_locate), the selective redo of overflowed guides, the
 * overflow arena made contiguous (k_arena_gather), the gaps shared items leave closed (k_share_*), raw counts for --threshold.
 */
#include "gs_kernels.h"

/* ---- order: per guide canonical order + dedupe ----------------------------- */

/* k_order's guides of at most SEGW records, 64 / SEGW of them at a time: a lane per record - the rank sort's compares run
 * over the segment's records in LDS (`srt`: 64 records in, 64 out), the std::set's dedupe over the segment's share of one
 * ballot.  lc0, lc1: the lanes' own guides' counts (guide g0 + lane); todo: the lanes whose guides are served here.
 * Returns the records kept (wave-uniform). */
template <uint32_t SEGW>
__device__ __forceinline__ uint32_t order_segments(const gs_order_args &a, uint32_t g0, uint32_t lc0, uint32_t lc1, uint64_t todo,
                                                   uint4 *srt) {
  const uint32_t lane = lane_id(), cap = a.cap;
  const uint32_t seg = lane / SEGW, sub = lane % SEGW;
  uint4 *qin = srt + seg * SEGW, *qout = srt + WAVE + seg * SEGW;
  uint32_t total = 0;
  while (todo) {
    uint32_t tsel = WAVE;
#pragma unroll
    for (uint32_t s4 = 0; s4 < WAVE / SEGW; s4++) {
      const uint32_t t = todo ? (uint32_t)__ffsll((unsigned long long)todo) - 1u : (uint32_t)WAVE;
      todo &= todo - 1ull;
      tsel = seg == s4 ? t : tsel;
    }
    const bool act = tsel < WAVE;
    const uint32_t src = act ? tsel : lane;
    const uint32_t c0 = (uint32_t)__shfl((int)lc0, (int)src), c1 = (uint32_t)__shfl((int)lc1, (int)src);
    const uint32_t M = act ? c0 + c1 : 0u;
    const uint32_t g = g0 + src;
    uint4 *base = a.slots + (size_t)g * 2 * cap;
    const bool has = sub < M;
    uint4 me = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (has) me = sub < c0 ? base[sub] : base[cap + (sub - c0)];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    qin[sub] = me;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint64_t key = ((uint64_t)me.y << 32) | me.x;
    uint32_t rank = 0; /* ascending (key, first row, original index) */
    for (uint32_t j = 0; j < SEGW; j++) {
      if (__ballot(j < M) == 0ull) break;
      const uint4 o = qin[j];
      const uint64_t ok = ((uint64_t)o.y << 32) | o.x;
      rank += (j < M && ((ok < key) || (ok == key && (o.z < me.z || (o.z == me.z && j < sub))))) ? 1u : 0u;
    }
    if (has) qout[rank] = me;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    /* dedupe equal sequences (std::set keeps the first), compact, count hits */
    bool keep = false;
    uint4 sm = make_uint4(0, 0, 0, 0);
    if (has) {
      sm = qout[sub];
      keep = true;
      if (sub > 0) {
        const uint4 pv = qout[sub - 1u];
        keep = !(pv.x == sm.x && pv.y == sm.y && pv.z == sm.z); /* same sequence, same rows */
      }
    }
    const uint64_t kb = __ballot(keep);
    const uint32_t kseg = (uint32_t)(kb >> (seg * SEGW)) & (uint32_t)((1ull << SEGW) - 1ull);
    const uint32_t cnt = keep ? (sm.w - sm.z + 1u) : 0u;
    if (keep) base[__popc(kseg & (uint32_t)((1ull << sub) - 1ull))] = make_uint4(sm.x, sm.y, sm.z, cnt);
    uint32_t hs = cnt;
    for (int o = SEGW / 2; o > 0; o >>= 1) hs += __shfl_xor(hs, o);
    if (act && sub == 0u) {
      a.nmatch[g] = (uint32_t)__popc(kseg);
      a.nhits[g] = hs;
    }
    total += (uint32_t)__popcll(kb);
  }
  return total;
}

/* A wavefront takes a GROUP of guides at a time (gs_lane_group: 64 on a batch of a million), ORDER_WAVES wavefronts per
 * workgroup, groups dealt to the waves grid-stride (a launch of one single-wave workgroup per guide with one atomic each
 * was latency bound: 12 ms per 1 M guides).  Round 6: a guide with at most ONE record - nearly every guide of an m <= 3
 * batch on a genome without repeat families - needs no order: lane l of the wave serves guide l of the group by itself
 * (64 count reads, 64 record reads and 64 stores in flight instead of one guide's chain of five dependent steps:
 * what a batch of guides that occur once gains); guides of 2 .. 16 records - the headline's batch: 13 per guide - are
 * served FOUR at a time, sixteen lanes each, those of 17 .. 32 two at a time (order_segments); the guides of the group
 * with more records by the whole wave, one after the other, as before (0.33 -> 0.2 ms per 1 M guides).  Dynamic LDS per wave: 2*cap uint4 (records) + ORDER_SMALL
 * uint4 (rank-sort output).  Up to ORDER_SMALL records a guide is rank-sorted (M^2/64 compares per
 * lane: 11 at the 26 records of an m = 3 guide); larger guides go through a bitonic network in
 * place (log^2 N / 2 steps of N/128 compare-exchanges per lane: at the 1,440 records of an m = 5
 * guide 2.1 k per lane instead of 32 k).  The wave's loop body has no lane-conditional blocks (DESIGN.md
 * 5b, compiler pitfall): per-guide results are stored by all lanes to the same address. */
__global__ __launch_bounds__(WAVE *ORDER_WAVES) void k_order(gs_order_args a) {
  extern __shared__ uint4 s_mem[];
  __shared__ uint32_t s_total;
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
  const uint32_t cap = a.cap;
  uint4 *rec = s_mem + (size_t)wave * (2u * cap + ORDER_SMALL);
  uint4 *srt = rec + 2u * cap;
  uint32_t total_out = 0;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  const uint32_t gsz = gs_lane_group(a.n), n_groups = (a.n + gsz - 1) / gsz;
  for (uint32_t G = blockIdx.x * nw + wave; G < n_groups; G += gridDim.x * nw) {
    /* the lanes' own guides: none or one record, or more records than slots (redone with larger slots, host side; the
     * redo's totals are patched in before the scan) */
    const uint32_t gl = G * gsz + lane;
    const bool mine = lane < gsz && gl < a.n;
    uint32_t lc0 = 0, lc1 = 0;
    if (mine) {
      const uint2 c = ((const uint2 *)a.counts)[gl];
      lc0 = c.x;
      lc1 = c.y;
    }
    const bool ovf = lc0 > cap || lc1 > cap;
    const bool one = mine && !ovf && lc0 + lc1 == 1u;
    if (mine && (ovf || lc0 + lc1 <= 1u)) {
      uint32_t h = 0;
      if (one) {
        uint4 *b = a.slots + (size_t)gl * 2 * cap;
        const uint4 me = b[lc0 ? 0u : cap];
        h = me.w - me.z + 1u;
        b[0] = make_uint4(me.x, me.y, me.z, h);
      }
      a.nmatch[gl] = one ? 1u : 0u;
      a.nhits[gl] = h;
    }
    total_out += (uint32_t)__popcll(__ballot(one));
    /* guides of 2 .. 16 records FOUR at a time, sixteen lanes each; of 17 .. 32 two at a time */
    total_out += order_segments<16>(a, G * gsz, lc0, lc1, __ballot(mine && !ovf && lc0 + lc1 > 1u && lc0 + lc1 <= 16u), srt);
    total_out += order_segments<32>(a, G * gsz, lc0, lc1, __ballot(mine && !ovf && lc0 + lc1 > 16u && lc0 + lc1 <= 32u), srt);
    uint64_t todo = __ballot(mine && !ovf && lc0 + lc1 > 32u);
    while (todo) {
      const uint32_t tl = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
      todo &= todo - 1ull;
      const uint32_t g = G * gsz + tl;
      const uint32_t c0 = (uint32_t)__shfl((int)lc0, (int)tl), c1 = (uint32_t)__shfl((int)lc1, (int)tl);
      const uint32_t M = c0 + c1;
      uint4 *base = a.slots + (size_t)g * 2 * cap;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (uint32_t i = lane; i < M; i += WAVE) rec[i] = i < c0 ? base[i] : base[cap + (i - c0)];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint4 *sorted = srt;
      if (M <= ORDER_SMALL) {
        /* rank sort: ascending (key, first row, original index) */
        for (uint32_t i = lane; i < M; i += WAVE) {
          const uint4 me = rec[i];
          const uint64_t key = ((uint64_t)me.y << 32) | me.x;
          uint32_t rank = 0;
          for (uint32_t j = 0; j < M; j++) {
            const uint4 o = rec[j];
            const uint64_t ok = ((uint64_t)o.y << 32) | o.x;
            rank += (ok < key) || (ok == key && (o.z < me.z || (o.z == me.z && j < i)));
          }
          srt[rank] = me;
        }
      } else {
        /* bitonic network over N = the next power of two, padded with records that sort last */
        uint32_t N = 2u * ORDER_SMALL;
        while (N < M) N <<= 1;
        for (uint32_t i = M + lane; i < N; i += WAVE) rec[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (uint32_t kk = 2; kk <= N; kk <<= 1)
          for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t t = lane; t < (N >> 1); t += WAVE) {
              const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), hi = lo | j;
              const uint4 A = rec[lo], B = rec[hi];
              const uint64_t ka = ((uint64_t)A.y << 32) | A.x, kb = ((uint64_t)B.y << 32) | B.x;
              const bool gt = ka > kb || (ka == kb && A.z > B.z);
              if (gt == ((lo & kk) == 0u)) {
                rec[lo] = B;
                rec[hi] = A;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          }
        sorted = rec;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      /* dedupe equal sequences (std::set keeps the first), compact, count hits */
      uint32_t n_out = 0, hits = 0;
      for (uint32_t i0 = 0; i0 < M; i0 += WAVE) {
        const uint32_t i = i0 + lane;
        bool keep = false;
        uint4 me = make_uint4(0, 0, 0, 0);
        if (i < M) {
          me = sorted[i];
          keep = true;
          if (i > 0) {
            const uint4 pv = sorted[i - 1];
            keep = !(pv.x == me.x && pv.y == me.y && pv.z == me.z); /* same sequence, same rows */
          }
        }
        const uint64_t kb = __ballot(keep);
        const uint32_t cnt = keep ? (me.w - me.z + 1u) : 0u;
        if (keep) base[n_out + lanes_below(kb)] = make_uint4(me.x, me.y, me.z, cnt);
        n_out += __popcll(kb);
        /* wave sum of cnt */
        uint32_t s = cnt;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        hits += s;
      }
      a.nmatch[g] = n_out;
      a.nhits[g] = hits;
      total_out += n_out;
    }
  }
  /* one atomic per workgroup (one per wave was 16,000 on one word of memory: 0.18 ms behind the last guide) */
  if (lane == 0 && total_out) atomicAdd(&s_total, total_out);
  __syncthreads();
  if (threadIdx.x == 0 && s_total) atomicAdd(&a.stats[2], (unsigned long long)s_total);
}

/* The same for guides with hundreds to thousands of matches (cap > 128): one 256-thread workgroup
 * per guide at a time, the bitonic network spread over its four waves (a single wave needs 2.1 k
 * compare-exchange rounds for the 1,440 records of an m = 5 guide), LDS sized by the largest guide
 * of the batch (`nmax` records, a power of two) rather than by the slot capacity. */
__global__ __launch_bounds__(256) void k_order_wg(gs_order_args a, uint32_t nmax) {
  extern __shared__ uint4 s_mem[];
  uint4 *rec = s_mem;
  __shared__ uint32_t s_nout, s_hits;
  const uint32_t tid = threadIdx.x, lane = lane_id();
  const uint32_t cap = a.cap;
  uint32_t total_out = 0;
  for (uint32_t g = blockIdx.x; g < a.n; g += gridDim.x) {
    const uint32_t c0 = a.counts[2 * g], c1 = a.counts[2 * g + 1];
    const uint32_t M = c0 + c1;
    if (c0 > cap || c1 > cap || M > nmax) { /* redone with larger slots (host side) */
      if (tid == 0) {
        a.nmatch[g] = 0;
        a.nhits[g] = 0;
      }
      continue;
    }
    uint4 *base = a.slots + (size_t)g * 2 * cap;
    uint32_t N = 64;
    while (N < M) N <<= 1;
    __syncthreads(); /* the previous guide's compaction has finished reading rec[] */
    for (uint32_t i = tid; i < N; i += 256)
      rec[i] = i < c0 ? base[i] : i < M ? base[cap + (i - c0)] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    /* Compare-exchange t touches lo and lo | j.  With j <= 64 the 64 exchanges of a wave stay inside one
     * aligned block of 128 records - the same block for every such j - so a pass needs the workgroup
     * barrier only when it or the pass before it reaches further (j >= 128): 14 barriers instead of
     * 66 at N = 2,048; the other passes order their LDS accesses within the wave. */
    uint32_t j_prev = 128;
    for (uint32_t kk = 2; kk <= N; kk <<= 1)
      for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
        if (j >= 128u || j_prev >= 128u) {
          __syncthreads();
        } else {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        j_prev = j;
        for (uint32_t t = tid; t < (N >> 1); t += 256) {
          const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), hi = lo | j;
          const uint4 A = rec[lo], B = rec[hi];
          const uint64_t ka = ((uint64_t)A.y << 32) | A.x, kb = ((uint64_t)B.y << 32) | B.x;
          const bool gt = ka > kb || (ka == kb && A.z > B.z);
          if (gt == ((lo & kk) == 0u)) {
            rec[lo] = B;
            rec[hi] = A;
          }
        }
      }
    __syncthreads();
    /* dedupe equal sequences (std::set keeps the first), compact, count hits: the first wave alone */
    if (tid < WAVE) {
      uint32_t n_out = 0, hits = 0;
      for (uint32_t i0 = 0; i0 < M; i0 += WAVE) {
        const uint32_t i = i0 + lane;
        bool keep = false;
        uint4 me = make_uint4(0, 0, 0, 0);
        if (i < M) {
          me = rec[i];
          keep = true;
          if (i > 0) {
            const uint4 pv = rec[i - 1];
            keep = !(pv.x == me.x && pv.y == me.y && pv.z == me.z);
          }
        }
        const uint64_t kb = __ballot(keep);
        const uint32_t cnt = keep ? (me.w - me.z + 1u) : 0u;
        if (keep) base[n_out + lanes_below(kb)] = make_uint4(me.x, me.y, me.z, cnt);
        n_out += __popcll(kb);
        uint32_t s = cnt;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        hits += s;
      }
      if (lane == 0) {
        a.nmatch[g] = n_out;
        a.nhits[g] = hits;
      }
      total_out += n_out;
    }
  }
  (void)s_nout;
  (void)s_hits;
  if (tid == 0 && total_out) atomicAdd(&a.stats[2], (unsigned long long)total_out);
}

/* ---- exclusive scan of nhits (uint32) into uint64 offsets ------------------- */
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partial(const uint32_t *in, uint64_t *blocksum,
                                                             uint32_t n) {
  __shared__ unsigned long long s[SCAN_BLOCK / WAVE];
  const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  unsigned long long v = i < n ? in[i] : 0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane_id() == 0) s[threadIdx.x / WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int j = 0; j < SCAN_BLOCK / WAVE; j++) t += s[j];
    blocksum[blockIdx.x] = t;
  }
}
/* single block: exclusive scan of the block sums in place, total to blocksum[nb] */
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_blocksums(uint64_t *blocksum, uint32_t nb) {
  __shared__ unsigned long long s[SCAN_BLOCK];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    const unsigned long long v = i < nb ? blocksum[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < SCAN_BLOCK; o <<= 1) {
      unsigned long long add = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) blocksum[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += s[SCAN_BLOCK - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) blocksum[nb] = carry;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_final(const uint32_t *in, const uint64_t *blocksum,
                                                           uint64_t *out, uint32_t n, uint32_t nb) {
  __shared__ unsigned long long s[SCAN_BLOCK];
  const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  const unsigned long long v = i < n ? in[i] : 0;
  s[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t o = 1; o < SCAN_BLOCK; o <<= 1) {
    unsigned long long add = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  if (i < n) out[i] = blocksum[blockIdx.x] + s[threadIdx.x] - v;
  if (i == 0) out[n] = blocksum[nb];
}

/* ---- locate: SA gather + coordinate rule ----------------------------------- */

/* One wavefront per GROUP of guides (gs_lane_group: 64 on a batch of a million; one single-wave workgroup per group);
 * dynamic LDS: (2*cap + 1) uint32 exclusive prefix of match sizes.  Round 6: the guides whose every record is ONE row
 * (hits == records: nearly every guide of a genome without repeat families) are located TOGETHER, a lane per hit of the
 * group - lane -> (guide, record) through the prefix of the guides' hit counts in LDS - so 64 suffix array gathers are in
 * flight whatever a guide holds (a workgroup per guide had 13 of 64 lanes at work on the headline's batch, and a million
 * workgroups to dispatch: 0.36 ms per 1 M guides); the other guides of the group are served by the whole wave one after
 * the other, as before. */
__global__ __launch_bounds__(WAVE) void k_locate(gs_locate_args a) {
  extern __shared__ uint32_t s_pre[];
  __shared__ uint32_t s_fl[WAVE + 1];
  __shared__ uint64_t s_off[WAVE];
  const uint32_t lane = lane_id();
  const uint32_t gsz = gs_lane_group(a.n);
  const uint32_t gl = blockIdx.x * gsz + lane;
  const bool mine = lane < gsz && gl < a.n;
  const uint32_t lM = mine ? a.nmatch[gl] : 0u;
  uint64_t off = 0;
  uint32_t lH = 0;
  if (lM != 0u) {
    const uint32_t oi = a.gmap ? a.gmap[gl] : gl;
    off = a.offsets[oi];
    lH = (uint32_t)(a.offsets[oi + 1] - off);
  }
  const bool flat = lM != 0u && lH == lM; /* one row per record */
  {
    const uint32_t c = flat ? lM : 0u;
    uint32_t inc = c; /* inclusive wave scan */
    for (int o = 1; o < WAVE; o <<= 1) {
      const uint32_t up = __shfl_up(inc, o);
      if ((int)lane >= o) inc += up;
    }
    s_fl[lane] = inc - c;
    s_off[lane] = off;
    if (lane == WAVE - 1) s_fl[WAVE] = inc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t T = s_fl[WAVE];
    for (uint32_t f = lane; f < T; f += WAVE) {
      uint32_t lo = 0, hi = WAVE; /* the last guide j of the group with s_fl[j] <= f: the one that has hit f */
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_fl[mid] <= f)
          lo = mid;
        else
          hi = mid;
      }
      const uint32_t r = f - s_fl[lo];
      const uint4 m = a.matches[(size_t)(blockIdx.x * gsz + lo) * 2 * a.cap + r];
      const uint64_t key = ((uint64_t)m.y << 32) | m.x;
      const uint32_t strand = (uint32_t)(key >> 60) & 1u;
      const uint64_t sa = (uint64_t)a.sd[strand].sa[m.z] - ((key & 1ull) ? a.v_rem : 0u);
      gs_hit o;
      /* process.hpp:104 / :111 */
      o.pos = strand == 0 ? -(int64_t)sa : (int64_t)(a.genome_length - (sa + 1ull));
      o.key = key & ~1ull;
      a.hits[s_off[lo] + r] = o;
    }
  }
  uint64_t todo = __ballot(lM != 0u && !flat);
  while (todo) {
    const uint32_t tl = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
    todo &= todo - 1ull;
    const uint32_t g = blockIdx.x * gsz + tl;
    const uint32_t M = (uint32_t)__shfl((int)lM, (int)tl);
    const uint4 *mt = a.matches + (size_t)g * 2 * a.cap;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); /* the guide before this one has read its prefix */
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t run = 0;
    for (uint32_t i0 = 0; i0 < M; i0 += WAVE) {
      const uint32_t i = i0 + lane;
      const uint32_t c = i < M ? mt[i].w : 0;
      uint32_t inc = c; /* inclusive wave scan */
      for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o);
        if ((int)lane >= o) inc += up;
      }
      if (i < M) s_pre[i] = run + inc - c;
      run += __shfl(inc, WAVE - 1);
    }
    if (lane == 0) s_pre[M] = run;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t H = run;
    gs_hit *out = a.hits + a.offsets[a.gmap ? a.gmap[g] : g];
    for (uint32_t h = lane; h < H; h += WAVE) {
      /* last match j with s_pre[j] <= h */
      uint32_t lo = 0, hi = M;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pre[mid] <= h)
          lo = mid;
        else
          hi = mid;
      }
      const uint4 m = mt[lo];
      const uint64_t key = ((uint64_t)m.y << 32) | m.x;
      const uint32_t strand = (uint32_t)(key >> 60) & 1u;
      const uint32_t row = m.z + (h - s_pre[lo]);
      const uint64_t sa = (uint64_t)a.sd[strand].sa[row] - ((key & 1ull) ? a.v_rem : 0u);
      gs_hit o;
      /* process.hpp:104 / :111 */
      o.pos = strand == 0 ? -(int64_t)sa : (int64_t)(a.genome_length - (sa + 1ull));
      o.key = key & ~1ull;
      out[h] = o;
    }
  }
}

/* ---- unit kernels ----------------------------------------------------------- */
__global__ void k_rank4(gs_strand_dev sd, const uint64_t *rows, uint64_t n, uint64_t *out) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t a, c, g, t;
  const uint32_t i = (uint32_t)rows[j];
  occ4(sd.blocks, i >> GS_BLOCK_SHIFT, i & (GS_BLOCK_ROWS - 1u), a, c, g, t);
  out[4 * j + 0] = a;
  out[4 * j + 1] = c;
  out[4 * j + 2] = g;
  out[4 * j + 3] = t;
}
__global__ void k_resolve(gs_strand_dev sd, const uint64_t *rows, uint64_t n, uint64_t *out) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  out[j] = sd.sa[rows[j]];
}

/* ---- selective redo of guides whose matches overflowed their slots -------------------- */
__global__ void k_collect_overflow(const uint32_t *counts, uint32_t n, uint32_t cap, uint32_t *list,
                                   uint32_t *n_list) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  if (counts[2 * g] > cap || counts[2 * g + 1] > cap) list[atomicAdd(n_list, 1u)] = g;
}
__global__ void k_gather_guides(const gs_guide_rec *in, const uint32_t *list, uint32_t n_o,
                                gs_guide_rec *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_o) out[i] = in[list[i]];
}
__global__ void k_gather_counts(const uint32_t *counts, const uint32_t *list, uint32_t n_o, uint32_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_o) {
    out[2 * i] = counts[2 * list[i]];
    out[2 * i + 1] = counts[2 * list[i] + 1];
  }
}
__global__ void k_patch_overflow(const uint32_t *list, uint32_t n_o, const uint32_t *nhits2,
                                 uint32_t *nhits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_o) nhits[list[i]] = nhits2[i];
}

/* ---- overflow arena -> contiguous records (instead of a second search pass) ------------------------
 * The guides on the overflow list have their first `cap` records per item in the main slot array and the
 * rest in arena chunks (gs_search_args::arena).  Item j of the list (2 * position + strand) is copied to
 * dst at dst_off[j] (exact-size layout) or at j * cap2 (fixed stride): blocks [0, 2 n_o) copy the slot
 * parts, one block per chunk the rest. */
__global__ __launch_bounds__(256) void k_arena_gather(gs_agather_args a) {
  const uint32_t b = blockIdx.x;
  const uint4 *src;
  uint32_t j, first, nrec;
  if (b < 2u * a.n_o) {
    j = b;
    const uint32_t item = 2u * a.list[j >> 1] + (j & 1u);
    const uint32_t c = a.counts[item];
    src = a.slots + (size_t)item * a.cap;
    first = 0;
    nrec = c < a.cap ? c : a.cap;
  } else {
    const uint32_t c = b - 2u * a.n_o;
    if (c >= a.n_used) return;
    if (a.chunk_seq[c] == 0xFFFFFFFFu) return; /* emptied when a shared item's gaps were closed (k_share_fix) */
    const uint32_t item = a.chunk_item[c];
    const uint32_t pos = a.redo_pos[item >> 1];
    if (pos == 0xFFFFFFFFu) return; /* cannot happen: an item with chunks overflowed its slots */
    j = 2u * pos + (item & 1u);
    const uint32_t cnt = a.counts[item], e0 = a.chunk_seq[c] << ARENA_SHIFT;
    if (cnt <= a.cap + e0) return;
    src = a.arena + ((size_t)c << ARENA_SHIFT);
    first = a.cap + e0;
    nrec = cnt - first < ARENA_CHUNK ? cnt - first : ARENA_CHUNK;
  }
  uint4 *dst = a.dst + (a.dst_off ? (size_t)a.dst_off[j] : (size_t)j * a.cap2) + first;
  for (uint32_t i = threadIdx.x; i < nrec; i += blockDim.x) dst[i] = src[i];
}

/* ---- shared items (gs_search_args::shq): the gaps their helpers left are closed -----------------------------
 * Behind k_search a shared item's records lie in its slots (the owner's first `cap`), the owner's chunks (full but
 * the last) and the helpers' chunks (each episode's last one partly filled).  Everything downstream reads an item as
 * "slots, then chunks 0, 1, .. in order, all full but the last": k_share_fix moves the records that lie beyond the
 * item's total into the holes before it (their order inside an item means nothing: the ordering kernels sort by
 * (sequence, row)), renumbers the chunks, drops the emptied ones (chunk_seq = 0xFFFFFFFF) and adds the helpers' counts. */
__global__ __launch_bounds__(1024) void k_share_scan(gs_share_args a) {
  __shared__ uint32_t s_w[16], s_carry;
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u), w = tid / WAVE;
  const uint32_t n_sh = a.ctl[96] < a.sh_max ? a.ctl[96] : a.sh_max;
  if (tid == 0) s_carry = 0u;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n_sh + 1u; i0 += 1024u) {
    const uint32_t sid = i0 + tid;
    uint32_t v = 0;
    if (sid < n_sh) v = a.nchunk[a.sh_list[sid]].x + a.sh_acc[16u * sid + 1u];
    const uint32_t incl = wave_incl_sum(v);
    if (lane == WAVE - 1u) s_w[w] = incl;
    __syncthreads();
    uint32_t b = s_carry;
    for (uint32_t u = 0; u < w; ++u) b += s_w[u];
    if (sid <= n_sh) a.dbase[sid] = b + incl - v;
    __syncthreads();
    if (tid == 1023u) s_carry = b + incl;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_share_dir(gs_share_args a) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_used = *a.arena_next < a.arena_chunks ? *a.arena_next : a.arena_chunks;
  if (c >= n_used || a.chunk_seq[c] == 0xFFFFFFFFu) return; /* (reserved by a wave and never used) */
  const uint32_t slot = a.chunk_item[c];
  const uint2 nc = a.nchunk[slot];
  if (!(nc.y >> 31)) return;
  const uint32_t sid = nc.y & 0x7FFFFFFFu;
  const uint32_t n_sh = a.ctl[96] < a.sh_max ? a.ctl[96] : a.sh_max;
  if (sid >= n_sh) return;
  const uint32_t seq = a.chunk_seq[c];
  const uint32_t j = (seq & SH_HELPER_SEQ) ? nc.x + (seq & (SH_HELPER_SEQ - 1u)) : seq;
  const uint32_t d0 = a.dbase[sid];
  if (j < a.dbase[sid + 1u] - d0) a.dir[d0 + j] = c;
}
/* MAXSEG: the segments the workgroup's tables hold.  Two launches: items of at most SH_SMALLSEG chunks - nearly all - with 3 KB of LDS
 * (eight workgroups per CU), the others with SH_MAXSEG (49 KB: three per CU, which kept the whole kernel at 0.64 ms per 10,800 items). */
template <uint32_t MAXSEG>
__global__ __launch_bounds__(256) void k_share_fix(gs_share_args a) {
  /* segment 0 = the slots, segment 1 + j = chunk j of the directory */
  __shared__ uint32_t s_fill[MAXSEG + 1u], s_hole[MAXSEG + 2u], s_mov[MAXSEG + 2u];
  __shared__ uint32_t s_red[3][4], s_tot[3];
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u), w = tid / WAVE;
  const uint32_t n_sh = a.ctl[96] < a.sh_max ? a.ctl[96] : a.sh_max;
  for (uint32_t sid = blockIdx.x; sid < n_sh; sid += gridDim.x) {
    const uint32_t slot = a.sh_list[sid];
    const uint32_t own = a.counts[slot], H = a.sh_acc[16u * sid], nho = a.nchunk[slot].x;
    const uint32_t d0 = a.dbase[sid], ns = a.dbase[sid + 1u] - d0;
    const uint32_t *dir = a.dir + d0;
    const uint32_t T = own + H, cap = a.cap;
    const bool own_short = own > cap && own - cap > (nho << ARENA_SHIFT);
    const uint32_t nseg = ns + 1u;
    __syncthreads(); /* (the previous item's tables are no longer read) */
    if ((ns > SH_SMALLSEG) != (MAXSEG > SH_SMALLSEG)) continue; /* (the other launch's item; workgroup-uniform) */
    if (ns > SH_MAXSEG || own_short) {
      /* not in a state to be closed up (or the arena ran out under the owner): the total is exact, the host searches the
       * batch's overflowing guides again - this item among them */
      if (tid == 0) {
        a.counts[slot] = T > cap ? T : cap + 1u;
        atomicAdd(&a.stats[6], 1ull);
        atomicAdd(&a.stats[1], 1ull);
      }
      continue;
    }
    /* what each segment holds */
    uint32_t v_sum = 0;
    for (uint32_t s = tid; s < nseg; s += 256u) {
      uint32_t f;
      if (s == 0u)
        f = own < cap ? own : cap;
      else if (s - 1u < nho)
        f = s < nho ? ARENA_CHUNK : own - cap - ((nho - 1u) << ARENA_SHIFT);
      else
        f = a.chunk_fill[dir[s - 1u]];
      if (f > ARENA_CHUNK && s != 0u) f = ARENA_CHUNK;
      s_fill[s] = f;
      v_sum += f;
    }
    for (int o = 32; o > 0; o >>= 1) v_sum += (uint32_t)__shfl_xor((int)v_sum, o);
    if (lane == 0) s_red[0][w] = v_sum;
    __syncthreads();
    const uint32_t V = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    if (V != T) { /* a helper was short of chunks: as above */
      if (tid == 0) {
        a.counts[slot] = T > cap ? T : cap + 1u;
        atomicAdd(&a.stats[6], 1ull);
        atomicAdd(&a.stats[1], 1ull);
      }
      continue;
    }
    /* holes before place T and records at or beyond it, per segment; thread t takes a run of per segments */
    const uint32_t per = (nseg + 255u) / 256u, s_lo = tid * per, s_hi = s_lo + per < nseg ? s_lo + per : nseg;
    uint32_t h_sum = 0, m_sum = 0;
    for (uint32_t s = s_lo; s < s_hi; ++s) {
      const uint32_t start = s == 0u ? 0u : cap + ((s - 1u) << ARENA_SHIFT), room = s == 0u ? cap : ARENA_CHUNK, f = s_fill[s];
      const uint32_t in = T > start ? (T - start < room ? T - start : room) : 0u; /* places of the segment before T */
      const uint32_t hole = in > f ? in - f : 0u, mov = f > in ? f - in : 0u;
      s_hole[s] = hole;
      s_mov[s] = mov;
      h_sum += hole;
      m_sum += mov;
    }
    const uint32_t hi = wave_incl_sum(h_sum), mi = wave_incl_sum(m_sum);
    if (lane == WAVE - 1u) {
      s_red[1][w] = hi;
      s_red[2][w] = mi;
    }
    __syncthreads();
    uint32_t hb = hi - h_sum, mb = mi - m_sum;
    for (uint32_t u = 0; u < w; ++u) {
      hb += s_red[1][u];
      mb += s_red[2][u];
    }
    if (tid == 255u) {
      s_tot[1] = hb + h_sum;
      s_tot[2] = mb + m_sum;
    }
    for (uint32_t s = s_lo; s < s_hi; ++s) { /* exclusive prefixes in place */
      const uint32_t h = s_hole[s], m = s_mov[s];
      s_hole[s] = hb;
      s_mov[s] = mb;
      hb += h;
      mb += m;
    }
    __syncthreads();
    const uint32_t M = s_tot[2];
    if (tid == 0) {
      s_hole[nseg] = s_tot[1];
      s_mov[nseg] = M;
    }
    __syncthreads();
    if (s_tot[1] == M) {
      for (uint32_t r = tid; r < M; r += 256u) {
        /* mover r: the last segment whose prefix is <= r (segments without movers share a prefix with their successor) */
        uint32_t lo = 0, hi2 = nseg;
        while (hi2 - lo > 1u) {
          const uint32_t mid = (lo + hi2) >> 1;
          if (s_mov[mid] <= r) lo = mid; else hi2 = mid;
        }
        const uint32_t sm = lo, fm = s_fill[sm];
        const uint32_t startm = cap + ((sm - 1u) << ARENA_SHIFT); /* (segment 0 holds no mover unless T = 0: then M = 0) */
        const uint32_t inm = T > startm ? (T - startm < fm ? T - startm : fm) : 0u;
        const uint4 *src = a.arena + (((size_t)dir[sm - 1u] << ARENA_SHIFT) + inm + (r - s_mov[sm]));
        lo = 0, hi2 = nseg;
        while (hi2 - lo > 1u) {
          const uint32_t mid = (lo + hi2) >> 1;
          if (s_hole[mid] <= r) lo = mid; else hi2 = mid;
        }
        const uint32_t sh = lo, off = s_fill[sh] + (r - s_hole[sh]);
        uint4 *dst = sh == 0u ? a.slots + ((size_t)slot * cap + off) : a.arena + (((size_t)dir[sh - 1u] << ARENA_SHIFT) + off);
        *dst = *src;
      }
    }
    const uint32_t nf = T > cap ? (T - cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT : 0u;
    for (uint32_t j = tid; j < ns; j += 256u) a.chunk_seq[dir[j]] = j < nf ? j : 0xFFFFFFFFu;
    if (tid < 8u) a.cls[(size_t)slot * 8u + tid] += a.sh_acc[16u * sid + 8u + tid];
    if (tid == 0) {
      a.counts[slot] = T;
      a.nchunk[slot] = make_uint2(nf, nf ? dir[nf - 1u] : 0u);
      if (own <= cap && T > cap) atomicAdd(&a.stats[1], 1ull);
      if (s_tot[1] != M) atomicAdd(&a.stats[6], 1ull); /* (cannot happen: V = T makes holes and movers equal) */
    }
  }
}
template __global__ void k_share_fix<SH_SMALLSEG>(gs_share_args a);
template __global__ void k_share_fix<SH_MAXSEG>(gs_share_args a);

/* hits of a guide BEFORE the per-distance sets drop duplicate sequences: what the reference's
 * threshold filter counts (off_target_counter, process.hpp:25-27: ep - sp + 1 per callback, one
 * callback per PAM pattern that matches).  One wavefront per guide over its raw match records;
 * a guide whose matches overflowed its slots has far more than the filter's bound: saturated. */
__global__ __launch_bounds__(256) void k_raw_counts(const uint4 *slots, const uint32_t *counts, uint32_t n, uint32_t cap,
                                                    uint32_t *raw) {
  const uint32_t g = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE, lane = lane_id();
  if (g >= n) return;
  const uint32_t c0 = counts[2 * g], c1 = counts[2 * g + 1];
  unsigned long long s = 0;
  if (c0 > cap || c1 > cap) {
    s = 0xFFFFFFFFull;
  } else {
    const uint4 *base = slots + (size_t)g * 2 * cap;
    for (uint32_t i = lane; i < c0; i += WAVE) s += base[i].w - base[i].z + 1u;
    for (uint32_t i = lane; i < c1; i += WAVE) s += base[cap + i].w - base[cap + i].z + 1u;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  }
  if (lane == 0) raw[g] = s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s;
}
/* sum and maximum of the per-item match counts (slot sizing of the next batch) */
__global__ void k_count_stats(const uint32_t *counts, uint32_t n_items, unsigned long long *out) {
  /* a grid-stride loop over 16-byte words of four counts, ONE pair of atomics per workgroup of a grid of at most 256:
   * two words of memory take ~90 atomics per microsecond, so the count of them is what this kernel costs (one pair per
   * 64 items was 0.65 ms of a 23 ms step at 2 M items; one per wave of 1,024 workgroups still 0.1 ms; now 0.01) */
  __shared__ unsigned long long s_v[16], s_m[16];
  unsigned long long v = 0, mx = 0;
  const uint32_t n4 = n_items >> 2;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 c = ((const uint4 *)counts)[i];
    v += (unsigned long long)c.x + c.y + c.z + c.w;
    const uint32_t m01 = c.x > c.y ? c.x : c.y, m23 = c.z > c.w ? c.z : c.w, m4 = m01 > m23 ? m01 : m23;
    mx = m4 > mx ? m4 : mx;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n_items & 3u)) {
    const unsigned long long c = counts[4u * n4 + threadIdx.x];
    v += c;
    mx = c > mx ? c : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    v += __shfl_xor(v, o);
    const unsigned long long x = __shfl_xor(mx, o);
    mx = x > mx ? x : mx;
  }
  if (lane_id() == 0) {
    s_v[threadIdx.x / WAVE] = v;
    s_m[threadIdx.x / WAVE] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t w = 1; w < blockDim.x / WAVE; w++) {
      v += s_v[w];
      mx = s_m[w] > mx ? s_m[w] : mx;
    }
    if (v) atomicAdd(&out[0], v);
    if (mx) atomicMax(&out[1], mx);
  }
}
/* arena chunks the items' records beyond their slots take (the exact counts are known even when the arena ran out) */
__global__ void k_need_chunks(const uint32_t *counts, uint32_t n_items, uint32_t cap, uint32_t *out) {
  uint32_t v = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = counts[i];
    if (c > cap) v += (c - cap + ARENA_CHUNK - 1u) >> ARENA_SHIFT;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane_id() == 0 && v) atomicAdd(out, v);
}
