/*
 * gs_index.hip -- builds the device-resident FM-index layout.
 *
 * Replaces `guidescan index` (src/guidescan.cxx:109-179: sdsl::construct = divsufsort
 * SA -> BWT -> wt_huff -> 1-in-64 SA samples) with a GPU builder: suffix array by
 * prefix doubling over rocPRIM radix sorts, then BWT -> 64-byte Occ blocks + the full
 * uint32 suffix array kept in HBM (288 GB makes the reference's sampling unnecessary).
 * Offline step, not on the enumerate hot path; rocPRIM is used for the plain sorts/scans.
 */
#include "gs_common.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <functional>
#include <system_error>
#include <thread>
#include <cstdlib>
#include <cstring>

/* BWT symbol class of a text byte: 0..3 = A,C,G,T ; 4 = N ; 5 = anything else (incl. '\0') */
__device__ __forceinline__ uint32_t sym_class(uint8_t c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    case 'N': return 4;
    default: return 5;
  }
}
__device__ __forceinline__ uint8_t bwt_at(const uint8_t *text, const uint32_t *sa, uint64_t n,
                                          uint64_t row) {
  const uint32_t s = sa[row];
  return s ? text[s - 1] : text[n - 1];
}

/* one thread per 32-row word: builds lo/hi/ex words and per-word A,C,G,T counts */
__global__ void k_build_words(const uint8_t *text, const uint32_t *sa, uint64_t n, uint64_t nwords,
                              uint4 *blocks, uint32_t *wcount /* [4][nwords] */) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint32_t lo = 0, hi = 0, ex = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const uint64_t base = w * 32;
  for (uint32_t j = 0; j < 32; j++) {
    const uint64_t row = base + j;
    uint32_t cls = 5;
    if (row < n) cls = sym_class(bwt_at(text, sa, n, row));
    if (cls < 4) {
      lo |= (cls & 1u) << j;
      hi |= (cls >> 1) << j;
      c0 += cls == 0;
      c1 += cls == 1;
      c2 += cls == 2;
      c3 += cls == 3;
    } else {
      ex |= 1u << j;
    }
  }
  /* block = 4 x uint4: [cnt][lo x4][hi x4][ex x4]; word w%4 of block w/4 */
  uint32_t *b = (uint32_t *)(blocks + (w >> 2) * 4);
  const uint32_t k = (uint32_t)(w & 3);
  b[4 + k] = lo;
  b[8 + k] = hi;
  b[12 + k] = ex;
  wcount[0 * nwords + w] = c0;
  wcount[1 * nwords + w] = c1;
  wcount[2 * nwords + w] = c2;
  wcount[3 * nwords + w] = c3;
}
/* after an exclusive scan of wcount: header of block b = scanned value at word 4b */
__global__ void k_write_headers(uint4 *blocks, const uint32_t *wscan, uint64_t nwords,
                                uint64_t nblocks) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  uint4 h;
  h.x = wscan[0 * nwords + 4 * b];
  h.y = wscan[1 * nwords + 4 * b];
  h.z = wscan[2 * nwords + 4 * b];
  h.w = wscan[3 * nwords + 4 * b];
  blocks[4 * b] = h;
}
__global__ void k_histogram(const uint8_t *text, uint64_t n, unsigned long long *hist) {
  __shared__ unsigned int s[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s[i] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&s[text[i]], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x)
    if (s[i]) atomicAdd(&hist[i], (unsigned long long)s[i]);
}
/* boundaries of maximal runs of equal symbols outside A,C,G,T (and not the sentinel) in the BWT:
 * starts as row << 8 | symbol, (exclusive) ends as rows */
__global__ void k_x_runs(const uint8_t *text, const uint32_t *sa, uint64_t n, unsigned long long *starts,
                         uint32_t *ends, uint32_t *counters, uint32_t cap) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint8_t cur = bwt_at(text, sa, n, row);
  if (cur == 0 || sym_class(cur) < 4) return;
  const bool first = row == 0 || bwt_at(text, sa, n, row - 1) != cur;
  const bool last = row == n - 1 || bwt_at(text, sa, n, row + 1) != cur;
  if (first) {
    const uint32_t k = atomicAdd(&counters[0], 1u);
    if (k < cap) starts[k] = ((unsigned long long)row << 8) | cur;
  }
  if (last) {
    const uint32_t k = atomicAdd(&counters[1], 1u);
    if (k < cap) ends[k] = (uint32_t)(row + 1);
  }
}
__global__ void k_revcomp(const uint8_t *in, uint8_t *out, uint64_t len) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  uint8_t c = in[len - 1 - i];
  switch (c) { /* src/genomics/sequences.cxx:14-26 */
    case 'A': c = 'T'; break;
    case 'T': c = 'A'; break;
    case 'C': c = 'G'; break;
    case 'G': c = 'C'; break;
    case 'a': c = 't'; break;
    case 't': c = 'a'; break;
    case 'c': c = 'g'; break;
    case 'g': c = 'c'; break;
    default: break;
  }
  out[i] = c;
}

static inline unsigned nblk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

void gs_strand_free(gs_strand *s) {
  if (s->blocks) hipFree(s->blocks);
  if (s->sa) hipFree(s->sa);
  if (s->run_start) hipFree(s->run_start);
  if (s->run_cum) hipFree(s->run_cum);
  if (s->ptab) hipFree(s->ptab);
  if (s->ctx) hipFree(s->ctx);
  if (s->ctx16) hipFree(s->ctx16);
  if (s->ptab_rot) hipFree(s->ptab_rot);
  if (s->isa) hipFree(s->isa);
  if (s->exc_row) hipFree(s->exc_row);
  if (s->exc_sym) hipFree(s->exc_sym);
  if (s->xr_start) hipFree(s->xr_start);
  if (s->xr_cum) hipFree(s->xr_cum);
  if (s->xr_seg) hipFree(s->xr_seg);
  if (s->C256) hipFree(s->C256);
  *s = gs_strand();
}

gs_status gs_strand_from_device(const uint8_t *d_text, uint32_t *d_sa_owned, uint64_t n,
                                gs_strand *out, hipStream_t st) {
  if (n < 2 || n >= (1ull << 32) - 256) {
    gs_set_error("text length must be in [1, 2^32-258]");
    return GS_ERR_UNSUPPORTED;
  }
  const uint64_t nblocks = (n >> GS_BLOCK_SHIFT) + 1;
  const uint64_t nwords = nblocks * 4;
  uint4 *blocks = nullptr;
  uint32_t *wcount = nullptr, *wscan = nullptr;
  unsigned long long *d_hist = nullptr;
  GS_HIP(hipMalloc(&blocks, nblocks * 64));
  GS_HIP(hipMalloc(&wcount, nwords * 16));
  GS_HIP(hipMalloc(&wscan, nwords * 16));
  GS_HIP(hipMalloc(&d_hist, 256 * 8));
  GS_HIP(hipMemsetAsync(d_hist, 0, 256 * 8, st));
  hipLaunchKernelGGL(k_build_words, dim3(nblk(nwords, 256)), dim3(256), 0, st, d_text, d_sa_owned, n,
                     nwords, blocks, wcount);
  {
    size_t tmp_bytes = 0;
    GS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, wcount, wscan, 0u, nwords,
                                   rocprim::plus<uint32_t>(), st));
    void *tmp = nullptr;
    GS_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    for (int c = 0; c < 4; c++)
      GS_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, wcount + c * nwords, wscan + c * nwords, 0u,
                                     nwords, rocprim::plus<uint32_t>(), st));
    GS_HIP(hipStreamSynchronize(st));
    hipFree(tmp);
  }
  hipLaunchKernelGGL(k_write_headers, dim3(nblk(nblocks, 256)), dim3(256), 0, st, blocks, wscan,
                     nwords, nblocks);
  hipLaunchKernelGGL(k_histogram, dim3(1024), dim3(256), 0, st, d_text, n, d_hist);
  unsigned long long hist[256];
  GS_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  hipFree(wcount);
  hipFree(wscan);
  hipFree(d_hist);

  /* byte_alphabet semantics: C[c] = number of text symbols smaller than c
   * (sdsl/lib/csa_alphabet_strategy.cpp:25-55) */
  uint64_t Cc[257];
  uint64_t acc = 0;
  for (int c = 0; c < 256; c++) {
    Cc[c] = acc;
    acc += hist[c];
  }
  Cc[256] = acc;
  const char bases[5] = {'A', 'C', 'G', 'T', 'N'};
  for (int k = 0; k < 5; k++) out->C_acgtn[k] = hist[(uint8_t)bases[k]] ? Cc[(uint8_t)bases[k]] : 0;

  /* runs of the symbols outside A,C,G,T in the BWT */
  uint32_t *d_cnt = nullptr, *d_ends = nullptr;
  unsigned long long *d_starts = nullptr;
  uint32_t cap = 1u << 20;
  std::vector<unsigned long long> xstarts;
  std::vector<uint32_t> xends;
  for (;;) {
    GS_HIP(hipMalloc(&d_cnt, 8));
    GS_HIP(hipMalloc(&d_starts, 8ull * cap));
    GS_HIP(hipMalloc(&d_ends, 4ull * cap));
    GS_HIP(hipMemsetAsync(d_cnt, 0, 8, st));
    hipLaunchKernelGGL(k_x_runs, dim3(nblk(n, 256)), dim3(256), 0, st, d_text, d_sa_owned, n, d_starts,
                       d_ends, d_cnt, cap);
    uint32_t hc[2];
    GS_HIP(hipMemcpyAsync(hc, d_cnt, 8, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    if (hc[0] <= cap && hc[1] <= cap) {
      xstarts.resize(hc[0]);
      xends.resize(hc[1]);
      if (hc[0]) GS_HIP(hipMemcpy(xstarts.data(), d_starts, 8ull * hc[0], hipMemcpyDeviceToHost));
      if (hc[1]) GS_HIP(hipMemcpy(xends.data(), d_ends, 4ull * hc[1], hipMemcpyDeviceToHost));
      hipFree(d_cnt);
      hipFree(d_starts);
      hipFree(d_ends);
      break;
    }
    cap = std::max(hc[0], hc[1]);
    hipFree(d_cnt);
    hipFree(d_starts);
    hipFree(d_ends);
  }
  if (xstarts.size() != xends.size()) {
    gs_set_error("internal: run boundaries do not pair up");
    return GS_ERR_DEVICE;
  }
  /* runs are disjoint: the i-th start and the i-th end, both in row order, belong together */
  std::sort(xstarts.begin(), xstarts.end());
  std::sort(xends.begin(), xends.end());
  std::vector<uint32_t> starts, ends; /* the runs of 'N' (the fast path's literal-N rule) */
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> by_sym(256);
  for (size_t r = 0; r < xstarts.size(); r++) {
    const uint32_t row = (uint32_t)(xstarts[r] >> 8), sym = (uint32_t)(xstarts[r] & 255u);
    by_sym[sym].push_back({row, xends[r]});
    if (sym == 'N') {
      starts.push_back(row);
      ends.push_back(xends[r]);
    }
  }
  std::vector<uint32_t> cum(starts.size() + 1, 0);
  for (size_t r = 0; r < starts.size(); r++) cum[r + 1] = cum[r] + (ends[r] - starts[r]);
  {
    std::vector<uint32_t> xs, xc, c256(256, 0);
    std::vector<uint2> seg(256, make_uint2(0u, 0u));
    for (int c = 0; c < 256; c++) {
      c256[c] = hist[c] ? (uint32_t)Cc[c] : 0u; /* char2comp of an absent byte is 0 and C[0] = 0 */
      out->has_sym[c] = hist[c] != 0;
      if (by_sym[c].empty()) continue;
      seg[c] = make_uint2((uint32_t)xs.size(), (uint32_t)by_sym[c].size());
      uint32_t acc = 0;
      for (auto &pr : by_sym[c]) {
        xs.push_back(pr.first);
        xc.push_back(acc);
        acc += pr.second - pr.first;
      }
      xs.push_back(0xFFFFFFFFu); /* keeps xr_cum[i + 1] valid for the symbol's last run */
      xc.push_back(acc);
    }
    GS_HIP(hipMalloc(&out->xr_seg, sizeof(uint2) * 256));
    GS_HIP(hipMalloc(&out->C256, 4 * 256));
    GS_HIP(hipMemcpy(out->xr_seg, seg.data(), sizeof(uint2) * 256, hipMemcpyHostToDevice));
    GS_HIP(hipMemcpy(out->C256, c256.data(), 4 * 256, hipMemcpyHostToDevice));
    if (!xs.empty()) {
      GS_HIP(hipMalloc(&out->xr_start, 4 * xs.size()));
      GS_HIP(hipMalloc(&out->xr_cum, 4 * xc.size()));
      GS_HIP(hipMemcpy(out->xr_start, xs.data(), 4 * xs.size(), hipMemcpyHostToDevice));
      GS_HIP(hipMemcpy(out->xr_cum, xc.data(), 4 * xc.size(), hipMemcpyHostToDevice));
    }
    out->d.xr_start = (const uint32_t *)out->xr_start;
    out->d.xr_cum = (const uint32_t *)out->xr_cum;
    out->d.xr_seg = (const uint2 *)out->xr_seg;
    out->d.C256 = (const uint32_t *)out->C256;
  }

  out->blocks = blocks;
  out->sa = d_sa_owned;
  out->n = n;
  out->bytes = nblocks * 64 + n * 4;
  if (!starts.empty()) {
    GS_HIP(hipMalloc(&out->run_start, 4 * starts.size()));
    GS_HIP(hipMalloc(&out->run_cum, 4 * cum.size()));
    GS_HIP(hipMemcpy(out->run_start, starts.data(), 4 * starts.size(), hipMemcpyHostToDevice));
    GS_HIP(hipMemcpy(out->run_cum, cum.data(), 4 * cum.size(), hipMemcpyHostToDevice));
  }
  gs_strand_dev &d = out->d;
  d.blocks = blocks;
  d.sa = d_sa_owned;
  d.run_start = (const uint32_t *)out->run_start;
  d.run_cum = (const uint32_t *)out->run_cum;
  d.n = (uint32_t)n;
  d.nruns = (uint32_t)starts.size();
  for (int k = 0; k < 4; k++) d.C[k] = (uint32_t)Cc[(uint8_t)bases[k]];
  d.CN = (uint32_t)Cc[(uint8_t)'N'];
  d.has_n = hist[(uint8_t)'N'] ? 1u : 0u;
  return GS_OK;
}

/* ---------------- GPU suffix array: prefix doubling ------------------------- */
struct sa_flag_op {
  /* max-scan of group starts */
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

__global__ void k_sa_init_keys(const uint8_t *text, uint64_t n, const uint8_t *dense /*256*/,
                               uint32_t bits, uint32_t k0, uint64_t *keys, uint32_t *idx) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
  for (uint32_t j = 0; j < k0; j++) {
    const uint64_t p = i + j;
    const uint64_t c = p < n ? dense[text[p]] : 0; /* past the sentinel: pad with the smallest */
    key = (key << bits) | c;
  }
  keys[i] = key;
  idx[i] = (uint32_t)i;
}
/* head[i] = i if key[i] != key[i-1] else 0 */
__global__ void k_sa_heads(const uint64_t *keys, uint64_t n, uint32_t *head) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || keys[i] != keys[i - 1]) ? (uint32_t)i : 0u;
}
__global__ void k_sa_scatter_rank(const uint32_t *sa, const uint32_t *grp, uint64_t n, uint32_t *rank) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rank[sa[i]] = grp[i];
}
__global__ void k_sa_pair_keys(const uint32_t *sa, const uint32_t *rank, uint64_t n, uint64_t h,
                               uint32_t nbits, uint64_t *keys) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t s = sa[i];
  const uint64_t r1 = rank[s];
  const uint64_t r2 = (s + h < n) ? (uint64_t)rank[s + h] : 0ull;
  keys[i] = (r1 << nbits) | r2;
}
__global__ void k_sa_count_heads(const uint32_t *head, uint64_t n, unsigned long long *cnt) {
  /* grid-stride count of group heads; one atomic per workgroup */
  __shared__ unsigned int s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  unsigned int local = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    local += (i == 0 || head[i] != 0);
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(&s_sum, local);
  __syncthreads();
  if (threadIdx.x == 0 && s_sum) atomicAdd(cnt, (unsigned long long)s_sum);
}

gs_status gs_device_suffix_array(const uint8_t *d_text, uint64_t n, uint32_t *d_sa, hipStream_t st) {
  /* dense alphabet */
  unsigned long long *d_hist = nullptr;
  GS_HIP(hipMalloc(&d_hist, 256 * 8 + 8));
  GS_HIP(hipMemsetAsync(d_hist, 0, 256 * 8 + 8, st));
  hipLaunchKernelGGL(k_histogram, dim3(1024), dim3(256), 0, st, d_text, n, d_hist);
  unsigned long long hist[256];
  GS_HIP(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  uint8_t dense[256];
  uint32_t sigma = 0;
  for (int c = 0; c < 256; c++) dense[c] = hist[c] ? (uint8_t)sigma++ : 0;
  uint32_t bits = 1;
  while ((1u << bits) < sigma) bits++;
  const uint32_t k0 = 64 / bits;
  uint32_t nbits = 1;
  while ((1ull << nbits) < n) nbits++;

  uint8_t *d_dense = nullptr;
  uint64_t *keys_a = nullptr, *keys_b = nullptr;
  uint32_t *idx_a = nullptr, *rank = nullptr, *head = nullptr;
  unsigned long long *d_cnt = d_hist + 256;
  GS_HIP(hipMalloc(&d_dense, 256));
  GS_HIP(hipMemcpy(d_dense, dense, 256, hipMemcpyHostToDevice));
  GS_HIP(hipMalloc(&keys_a, 8 * n));
  GS_HIP(hipMalloc(&keys_b, 8 * n));
  GS_HIP(hipMalloc(&idx_a, 4 * n));
  GS_HIP(hipMalloc(&rank, 4 * n));
  GS_HIP(hipMalloc(&head, 4 * n));
  const unsigned g = nblk(n, 256);
  hipLaunchKernelGGL(k_sa_init_keys, dim3(g), dim3(256), 0, st, d_text, n, d_dense, bits, k0, keys_a,
                     idx_a);
  size_t sort_bytes = 0, scan_bytes = 0;
  GS_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_a, keys_b, idx_a, d_sa, n, 0, 64, st));
  GS_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, head, head, n, sa_flag_op(), st));
  void *tmp = nullptr;
  const size_t tmp_bytes = std::max(sort_bytes, scan_bytes);
  GS_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  size_t sb = tmp_bytes;
  GS_HIP(rocprim::radix_sort_pairs(tmp, sb, keys_a, keys_b, idx_a, d_sa, n, 0, bits * k0, st));
  gs_status rc = GS_OK;
  for (uint64_t h = k0;; h *= 2) {
    /* keys_b sorted, d_sa = suffixes in that order */
    hipLaunchKernelGGL(k_sa_heads, dim3(g), dim3(256), 0, st, keys_b, n, head);
    GS_HIP(hipMemsetAsync(d_cnt, 0, 8, st));
    hipLaunchKernelGGL(k_sa_count_heads, dim3(g < 2048 ? g : 2048), dim3(256), 0, st, head, n, d_cnt);
    unsigned long long groups = 0;
    GS_HIP(hipMemcpyAsync(&groups, d_cnt, 8, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    if (groups == n) break; /* every suffix has a distinct rank: sorted */
    if (h >= n) {
      gs_set_error("internal: suffix array doubling did not converge");
      rc = GS_ERR_DEVICE;
      break;
    }
    sb = tmp_bytes;
    GS_HIP(rocprim::inclusive_scan(tmp, sb, head, head, n, sa_flag_op(), st));
    hipLaunchKernelGGL(k_sa_scatter_rank, dim3(g), dim3(256), 0, st, d_sa, head, n, rank);
    hipLaunchKernelGGL(k_sa_pair_keys, dim3(g), dim3(256), 0, st, d_sa, rank, n, h, nbits, keys_a);
    GS_HIP(hipMemcpyAsync(idx_a, d_sa, 4 * n, hipMemcpyDeviceToDevice, st));
    sb = tmp_bytes;
    GS_HIP(rocprim::radix_sort_pairs(tmp, sb, keys_a, keys_b, idx_a, d_sa, n, 0, 2 * nbits, st));
  }
  GS_HIP(hipStreamSynchronize(st));
  hipFree(tmp);
  hipFree(d_dense);
  hipFree(keys_a);
  hipFree(keys_b);
  hipFree(idx_a);
  hipFree(rank);
  hipFree(head);
  hipFree(d_hist);
  return rc;
}

/* ---------------- prefix interval table ---------------------------------------- */
__device__ __forceinline__ bool kmer_code(const uint8_t *text, uint64_t n, uint64_t p, uint32_t k,
                                          uint32_t &code) {
  if (p + k > n - 1) return false; /* would run into the sentinel */
  uint32_t c = 0;
  for (uint32_t i = 0; i < k; i++) {
    const uint32_t cls = sym_class(text[p + i]);
    if (cls > 3) return false;
    c |= cls << (2 * i); /* first text symbol of the k-mer in the lowest bits */
  }
  code = c;
  return true;
}
__global__ void k_ptab_build(const uint8_t *text, const uint32_t *sa, uint64_t n, uint32_t k,
                             uint4 *tab) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  uint32_t c = 0, cp = 0, cn = 0;
  if (!kmer_code(text, n, sa[r], k, c)) return;
  const bool vp = r > 0 && kmer_code(text, n, sa[r - 1], k, cp);
  const bool vn = r + 1 < n && kmer_code(text, n, sa[r + 1], k, cn);
  if (!vp || cp != c) tab[c].x = (uint32_t)r;       /* first row of the k-mer's interval */
  if (!vn || cn != c) tab[c].y = (uint32_t)(r + 1); /* one past its last row */
}

/* (the builder's choices are switches of the handle like every other: the environment's GS_* variables as they were when
 * the handle was made, gs_opt() the only reader - no getenv on any path of the library) */
static uint32_t choose_prefix_k(const gs_index *ix, uint64_t n) {
  if (const char *e = gs_opt(ix, "GS_PREFIX_K")) return (uint32_t)atoi(e);
  /* deepest level at which k-mers still average 2+ rows; capped so the table stays <= 4 GiB.
   * (Until the two-level context check the rule was 8+ rows; a chr1-sized genome then got k = 12,
   * one short of what two-sided seeding needs for 20+3-mers, and ran 6 times slower than at 13:
   * 166 vs 27 ms per 1 M guides.  Fewer rows per interval also sharpen the context mask.) */
  uint32_t k = 0;
  uint64_t v = n / 2;
  while (v >= 4) {
    v >>= 2;
    k++;
  }
  if (k > 14) k = 14;
  if (k < 4) k = 0;
  return k;
}

/* end -> count */
__global__ void k_ptab_finish(uint4 *tab, uint64_t entries) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= entries) return;
  uint4 e = tab[i];
  e.y = e.y ? e.y - e.x : 0u;
  tab[i] = e;
}
/* ctx[r] = 16 symbols preceding suffix SA[r] (nearest first); rows whose window holds a
 * non-ACGT symbol or runs off the text start flag their k-mer's table entry */
__global__ void k_ctx_build(const uint8_t *text, const uint32_t *sa, uint64_t n, uint32_t k,
                            uint32_t *ctx, uint16_t *ctx16, uint4 *tab, uint32_t *exc_count,
                            uint32_t *exc_row, uint64_t *exc_sym, uint32_t exc_cap, uint32_t mask_off) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t p = sa[r];
  uint32_t w = 0;
  uint64_t nib = 0;
  bool exc = false;
  for (uint32_t j = 1; j <= 16; j++) {
    uint32_t cls = 6; /* before the text start */
    if (p >= j) cls = sym_class(text[p - j]);
    nib |= (uint64_t)cls << (4 * (j - 1));
    if (cls > 3) {
      exc = true;
      cls = 0;
    }
    w |= cls << (2 * (j - 1));
  }
  ctx[r] = w;
  ctx16[r] = (uint16_t)w;
  if (tab) {
    uint32_t c;
    if (kmer_code(text, n, p, k, c)) {
      if (exc) {
        atomicOr(&tab[c].y, 0x80000000u);
        const uint32_t at = atomicAdd(exc_count, 1u);
        if (at < exc_cap) {
          exc_row[at] = (uint32_t)r;
          exc_sym[at] = nib;
        }
      }
      /* which symbol pairs occur to the left of this k-mer's rows: four 16-bit sets, pair j =
       * the symbols 2j and 2j+1 before the suffix (nearest first), bit = 4 bits of the pair */
      const uint32_t o0 = mask_off & 15u, o1 = (mask_off >> 4) & 15u, o2 = (mask_off >> 8) & 15u, o3 = (mask_off >> 12) & 15u;
      atomicOr(&tab[c].z, (1u << ((w >> (2u * o0)) & 15u)) | (1u << (16u + ((w >> (2u * o1)) & 15u))));
      atomicOr(&tab[c].w, (1u << ((w >> (2u * o2)) & 15u)) | (1u << (16u + ((w >> (2u * o3)) & 15u))));
    }
  }
}

/* rot[slot][perm_p(i)] = tab[i]: the field of consumption step p (bits 2(k-1-p)) moves to bits 1:0 */
__global__ void k_rot_copy(const uint4 *tab, uint4 *rot, uint32_t k, uint32_t p, uint32_t slot) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> (2 * k)) return;
  const uint32_t sh = 2u * (k - 1u - p);
  const uint64_t hi = i >> (sh + 2u), lo = i & ((1ull << sh) - 1ull), f = (i >> sh) & 3ull;
  const uint64_t j = (hi << (sh + 2u)) | (lo << 2) | f;
  rot[((uint64_t)slot << (2 * k)) + j] = tab[i];
}

__global__ void k_isa_build(const uint32_t *sa, uint64_t n, uint32_t *isa) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) isa[sa[r]] = (uint32_t)r;
}

static gs_status build_ptab(const gs_index *ix, const uint8_t *d_text, gs_strand *s, uint32_t k, hipStream_t st) {
  if (!k) return GS_OK;
  const uint64_t entries = 1ull << (2 * k);
  const size_t bytes = sizeof(uint4) * entries;
  uint4 *tab = nullptr;
  GS_HIP(hipMalloc(&tab, bytes));
  GS_HIP(hipMemsetAsync(tab, 0, bytes, st));
  hipLaunchKernelGGL(k_ptab_build, dim3(nblk(s->n, 256)), dim3(256), 0, st, d_text,
                     (const uint32_t *)s->sa, s->n, k, tab);
  hipLaunchKernelGGL(k_ptab_finish, dim3(nblk(entries, 256)), dim3(256), 0, st, tab, entries);
  uint32_t *ctx = nullptr;
  uint16_t *ctx16 = nullptr;
  if (!gs_opt(ix, "GS_NO_CTX")) {
    GS_HIP(hipMalloc(&ctx, 4 * s->n + 16));
    GS_HIP(hipMalloc(&ctx16, 2 * s->n + 32)); /* one row group of padding (k_search reads groups of eight) */
    /* pair positions of the context mask (gs_strand_dev::mask_off) */
    uint32_t off3 = k < 21 ? 21u - k : 6u;
    if (off3 < 6) off3 = 6;
    if (off3 > 14) off3 = 14;
    uint32_t mask_off = 0u | (2u << 4) | (4u << 8) | (off3 << 12);
    if (const char *e = gs_opt(ix, "GS_MASK_OFFSETS")) { /* experiments: "0,2,4,7" */
      mask_off = 0;
      for (uint32_t j = 0; j < 4 && *e; j++) {
        const unsigned long v = strtoul(e, (char **)&e, 10);
        mask_off |= (uint32_t)(v > 14 ? 14 : v) << (4 * j);
        if (*e == ',') e++;
      }
    }
    s->d.mask_off = mask_off;
    /* exception rows: counted in a first pass when the first guess is too small */
    uint32_t *d_cnt = nullptr, *d_er = nullptr;
    uint64_t *d_es = nullptr;
    uint32_t cap = 1u << 16, h_cnt = 0;
    GS_HIP(hipMalloc(&d_cnt, 4));
    for (int pass = 0; pass < 2; pass++) {
      GS_HIP(hipMalloc(&d_er, 4 * (size_t)cap));
      GS_HIP(hipMalloc(&d_es, 8 * (size_t)cap));
      GS_HIP(hipMemsetAsync(d_cnt, 0, 4, st));
      hipLaunchKernelGGL(k_ctx_build, dim3(nblk(s->n, 256)), dim3(256), 0, st, d_text,
                         (const uint32_t *)s->sa, s->n, k, ctx, ctx16, tab, d_cnt, d_er, d_es, cap, mask_off);
      GS_HIP(hipMemcpyAsync(&h_cnt, d_cnt, 4, hipMemcpyDeviceToHost, st));
      GS_HIP(hipStreamSynchronize(st));
      if (h_cnt <= cap) break;
      hipFree(d_er);
      hipFree(d_es);
      d_er = nullptr;
      d_es = nullptr;
      cap = h_cnt; /* flags and masks are set by atomicOr: a second pass changes nothing there */
    }
    hipFree(d_cnt);
    if (h_cnt) {
      std::vector<uint32_t> er(h_cnt);
      std::vector<uint64_t> es(h_cnt);
      GS_HIP(hipMemcpy(er.data(), d_er, 4 * (size_t)h_cnt, hipMemcpyDeviceToHost));
      GS_HIP(hipMemcpy(es.data(), d_es, 8 * (size_t)h_cnt, hipMemcpyDeviceToHost));
      std::vector<uint32_t> ord(h_cnt);
      for (uint32_t i = 0; i < h_cnt; i++) ord[i] = i;
      std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return er[a] < er[b]; });
      std::vector<uint32_t> er2(h_cnt);
      std::vector<uint64_t> es2(h_cnt);
      for (uint32_t i = 0; i < h_cnt; i++) {
        er2[i] = er[ord[i]];
        es2[i] = es[ord[i]];
      }
      GS_HIP(hipMemcpy(d_er, er2.data(), 4 * (size_t)h_cnt, hipMemcpyHostToDevice));
      GS_HIP(hipMemcpy(d_es, es2.data(), 8 * (size_t)h_cnt, hipMemcpyHostToDevice));
      s->exc_row = d_er;
      s->exc_sym = d_es;
      s->d.exc_row = d_er;
      s->d.exc_sym = d_es;
      s->d.n_exc = h_cnt;
      s->bytes += 12 * (size_t)h_cnt;
    } else {
      hipFree(d_er);
      hipFree(d_es);
    }
    s->bytes += 6 * s->n;
  }
  /* optional structures, HBM capacity spent to cut random requests: the
   * rotated table copies (k-1 tables: 56 GB per strand at k = 14) and the inverse suffix array
   * (4n: what two-sided seeding needs).  GS_INDEX_BUDGET_GB caps the whole index (both strands):
   * the copies go first, then the inverse suffix array; an allocation that fails is skipped too. */
  double budget = 1e30;
  if (const char *e = gs_opt(ix, "GS_INDEX_BUDGET_GB")) budget = atof(e) * 1e9 / 2.0; /* per strand */
  const double base_bytes = (double)s->bytes + (double)bytes; /* blocks, SA, context arrays, table */
  uint32_t rot_first = 3; /* see gs_strand_dev::rot_first */
  if (const char *e = gs_opt(ix, "GS_ROT_FIRST")) rot_first = (uint32_t)atoi(e);
  const uint32_t nrot = k >= 4 && rot_first + 1 < k ? k - 1 - rot_first : 0;
  const double rot_bytes = (double)bytes * nrot, isa_bytes = 4.0 * (double)s->n;
  uint32_t *isa = nullptr;
  if (ctx && !gs_opt(ix, "GS_NO_BIDIR") && !gs_opt(ix, "GS_NO_ISA") && base_bytes + isa_bytes <= budget) {
    if (hipMalloc(&isa, 4 * s->n) == hipSuccess) {
      hipLaunchKernelGGL(k_isa_build, dim3(nblk(s->n, 256)), dim3(256), 0, st, (const uint32_t *)s->sa, s->n, isa);
      s->bytes += 4 * s->n;
    } else {
      isa = nullptr;
      (void)hipGetLastError();
    }
  }
  /* the rotated copies are derived data written in milliseconds: built by the first batch that reads them
   * (gs_strand_rot_ensure) - a batch served by PAM-pair and deep tables never does, and their 86 GB at hg38
   * size are better left to those tables and to the workspace */
  s->rot_plan_first = rot_first;
  s->rot_plan_n = (ctx && nrot && !gs_opt(ix, "GS_NO_ROT") && base_bytes + (isa ? isa_bytes : 0.0) + rot_bytes <= budget) ? nrot : 0;
  s->rot_k = k;
  GS_HIP(hipStreamSynchronize(st));
  s->isa = isa;
  s->d.isa = isa;
  s->ptab_rot = nullptr;
  s->d.ptab_rot = nullptr;
  s->d.rot_first = 31u;
  s->ptab = tab;
  s->d.ptab = tab;
  s->ctx = ctx;
  s->d.ctx = ctx;
  s->ctx16 = ctx16;
  s->d.ctx16 = ctx16;
  s->bytes += bytes;
  return GS_OK;
}

/* rotated copies of both strand tables (gs_strand_dev::ptab_rot), built when a batch reads them: one copy per
 * step rot_first..k-2 - steps up to k-3 serve the budget-0 variants (their last substituted step), step k-2
 * the budget-1 variants (substitutions of the second-last symbol next to each other).  Kept afterwards;
 * released when a batch runs short of memory (gs_strand_rot_release).  No room (beyond `reserve` bytes left
 * for the batch), GS_NO_ROT or GS_INDEX_BUDGET_GB: the plain tables serve every class. */
gs_status gs_strand_rot_ensure(gs_index *ix, hipStream_t st) {
  for (int s = 0; s < 2; s++) {
    gs_strand &S = ix->strand[s];
    if (S.ptab_rot || !S.rot_plan_n || !S.ptab || ix->rot_off) continue;
    const uint64_t entries = 1ull << (2 * S.rot_k);
    const size_t bytes = sizeof(uint4) * entries * S.rot_plan_n;
    size_t free_b = 0, total_b = 0;
    GS_HIP(hipMemGetInfo(&free_b, &total_b));
    double reserve = 32e9; /* they are dropped first when a batch runs short, and written again in 50 ms */
    if (reserve > 0.125 * (double)total_b) reserve = 0.125 * (double)total_b;
    uint4 *rot = nullptr;
    if ((double)bytes + reserve > (double)free_b || hipMalloc(&rot, bytes) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    for (uint32_t p = 0; p < S.rot_plan_n; p++)
      hipLaunchKernelGGL(k_rot_copy, dim3(nblk(entries, 256)), dim3(256), 0, st, (const uint4 *)S.ptab, rot, S.rot_k,
                         S.rot_plan_first + p, p);
    GS_HIP(hipStreamSynchronize(st));
    S.ptab_rot = rot;
    S.d.ptab_rot = rot;
    S.d.rot_first = S.rot_plan_first;
    S.bytes += bytes;
    if (gs_opt(ix, "GS_DEBUG"))
      fprintf(stderr, "[gs] strand %d: %u rotated table copies built (%.1f GB)\n", s, S.rot_plan_n, 1e-9 * (double)bytes);
  }
  return GS_OK;
}
bool gs_strand_rot_release(gs_index *ix) {
  bool any = false;
  for (int s = 0; s < 2; s++) {
    gs_strand &S = ix->strand[s];
    if (!S.ptab_rot) continue;
    hipFree(S.ptab_rot);
    S.bytes -= sizeof(uint4) * (1ull << (2 * S.rot_k)) * S.rot_plan_n;
    S.ptab_rot = nullptr;
    S.d.ptab_rot = nullptr;
    S.d.rot_first = 31u;
    any = true;
  }
  return any;
}

/* maximal runs of 'N' in the forward text, with 40 bytes of text on either side */
static void scan_n_runs(const uint8_t *text, uint64_t len, std::vector<gs_nrun> &out) {
  out.clear();
  uint64_t pos = 0;
  while (pos < len) {
    const uint8_t *hit = (const uint8_t *)memchr(text + pos, 'N', len - pos);
    if (!hit) break;
    gs_nrun r;
    r.start = (uint64_t)(hit - text);
    uint64_t e = r.start;
    while (e < len && text[e] == 'N') e++;
    r.len = e - r.start;
    for (uint64_t j = 0; j < GS_NRUN_FLANK; j++) {
      r.left[j] = r.start >= GS_NRUN_FLANK - j ? text[r.start - (GS_NRUN_FLANK - j)] : 0;
      r.right[j] = e + j < len ? text[e + j] : 0;
    }
    out.push_back(r);
    pos = e;
  }
}

/* ---------------- C-ABI: index lifecycle -------------------------------------- */
static gs_status build_common(const uint8_t *text, uint64_t len, const uint32_t *sa_fwd,
                              const uint32_t *sa_rev, int device, gs_index **out, gs_status bad_sa = GS_ERR_ARG, bool sa_on_device = false) {
  if (!text || !out || len < 1) return GS_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    gs_set_error("no such HIP device");
    return GS_ERR_DEVICE;
  }
  GS_HIP(hipSetDevice(device));
  const uint64_t n = len + 1;
  gs_index *ix = new (std::nothrow) gs_index();
  if (!ix) return GS_ERR_NOMEM;
  ix->device = device;
  ix->genome_length = len;
  gs_opts_from_env(ix);
  hipStream_t st = nullptr;
  uint8_t *d_fwd = nullptr, *d_rev = nullptr;
  /* every early return below (GS_HIP) releases what was allocated so far */
  struct cleanup_t {
    gs_index *&ix;
    uint8_t *&a, *&b;
    uint32_t *sa = nullptr; /* a suffix array not yet owned by a strand */
    bool armed = true;
    ~cleanup_t() {
      if (a) hipFree(a);
      if (b) hipFree(b);
      if (!armed) return;
      if (sa) hipFree(sa);
      if (ix) gs_index_close(ix);
    }
  } cleanup{ix, d_fwd, d_rev};
  GS_HIP(hipMalloc(&d_fwd, n));
  GS_HIP(hipMalloc(&d_rev, n));
  GS_HIP(hipMemcpy(d_fwd, text, len, hipMemcpyHostToDevice));
  GS_HIP(hipMemset(d_fwd + len, 0, 1)); /* sentinel: sdsl/include/sdsl/construct.hpp:133-135 */
  hipLaunchKernelGGL(k_revcomp, dim3(nblk(len, 256)), dim3(256), 0, st, d_fwd, d_rev, len);
  GS_HIP(hipMemset(d_rev + len, 0, 1));
  gs_status rc = GS_OK;
  const uint32_t pk = choose_prefix_k(ix, n);
  for (int s = 0; s < 2 && rc == GS_OK; s++) {
    uint32_t *d_sa = nullptr;
    GS_HIP(hipMalloc(&d_sa, 4 * n));
    cleanup.sa = d_sa;
    const uint32_t *given = s == 0 ? sa_fwd : sa_rev;
    const uint8_t *d_t = s == 0 ? d_fwd : d_rev;
    if (given) {
      GS_HIP(hipMemcpy(d_sa, given, 4 * n, sa_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
      /* a suffix array from outside (a file, the caller): every value in range and none twice, or no
       * table is derived from it - the builders read text[sa[r] - j] without looking */
      uint64_t bad = 0;
      if ((rc = gs_count_bad_sa_rows(d_sa, n, st, &bad)) == GS_OK && bad) {
        gs_set_error("the suffix array given for the " + std::string(s ? "reverse" : "forward") + " strand is not a permutation of its rows (" +
                     std::to_string(bad) + " bad rows)");
        rc = bad_sa;
      }
      if (rc != GS_OK) continue;
    } else {
      /* (gs_suffix.hip: doubling that leaves sorted suffixes alone; GS_SA_PLAIN: the first builder, every row every round) */
      rc = gs_opt(ix, "GS_SA_PLAIN") ? gs_device_suffix_array(d_t, n, d_sa, st)
                                     : gs_device_suffix_array_discarding(d_t, n, d_sa, st, gs_opt(ix, "GS_DEBUG") != nullptr);
    }
    if (rc == GS_OK) rc = gs_strand_from_device(d_t, d_sa, n, &ix->strand[s], st);
    if (ix->strand[s].sa == d_sa) cleanup.sa = nullptr; /* the strand owns it now (also when it failed later) */
    if (rc == GS_OK) rc = build_ptab(ix, d_t, &ix->strand[s], pk, st);
  }
  if (rc == GS_OK) ix->pt_k = pk; /* depth of the prefix interval tables; the seed recipes are written per batch shape (gs_search.hip) */
  if (rc == GS_OK) scan_n_runs(text, len, ix->nruns_text);
  if (rc != GS_OK) return rc; /* cleanup releases everything */
  cleanup.armed = false;
  *out = ix;
  return GS_OK;
}

extern "C" gs_status gs_index_build(const uint8_t *text, uint64_t len, int device, gs_index **out) {
  try {
    return build_common(text, len, nullptr, nullptr, device, out);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}
/* suffix arrays that are in device memory already (the importer of the reference's index files, gs_sdsl_import.hip) */
gs_status gs_build_from_device_sa(const uint8_t *text, uint64_t len, const uint32_t *d_sa_fwd, const uint32_t *d_sa_rev, int device, gs_index **out) {
  if (!d_sa_fwd || !d_sa_rev) return GS_ERR_ARG;
  try {
    return build_common(text, len, d_sa_fwd, d_sa_rev, device, out, GS_ERR_FORMAT, true);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}
extern "C" gs_status gs_index_build_with_sa(const uint8_t *text, uint64_t len, const uint32_t *sa_fwd,
                                            const uint32_t *sa_rev, int device, gs_index **out) {
  if (!sa_fwd || !sa_rev) return GS_ERR_ARG;
  try {
    return build_common(text, len, sa_fwd, sa_rev, device, out);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}

/* ---- native index file: the two suffix arrays, kept next to the genome text ------------------
 * `guidescan index` of the reference stores what `enumerate` loads (src/guidescan.cxx:168-175).
 * The device layout is derived data (210 GB at hg38 size, rebuilt from text + suffix arrays in a few
 * seconds); what is worth storing is the result of the sort: 4 bytes per row and strand. */
struct gs_sa_header {
  char magic[8]; /* "GSAMDSA2" ("GSAMDSA1": the same without payload checksums, still read) */
  uint64_t n;    /* rows per strand = text length + 1 */
  uint64_t text_hash;
  uint64_t sa_hash[2]; /* checksum of each strand's payload (sa_hash_update over its n entries) */
  uint64_t reserved[3];
};
/* running checksum of suffix-array entries, two at a time; chunks chain through h */
static uint64_t sa_hash_update(uint64_t h, const uint32_t *p, size_t m) {
  size_t i = 0;
  for (; i + 2 <= m; i += 2) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 32;
  }
  if (i < m) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}
static uint64_t text_fingerprint(const uint8_t *text, uint64_t len) {
  /* every byte of the text, eight at a time (a suffix array of another text would give wrong hits
   * silently, so no sampling): ~0.5 s at 3.1 GB */
  uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
  uint64_t i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t w;
    memcpy(&w, text + i, 8);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 32;
  }
  for (; i < len; i++) h = (h ^ text[i]) * 1099511628211ull;
  return h;
}
extern "C" gs_status gs_index_save_sa(gs_index *ix, const uint8_t *text, uint64_t len, const char *path) {
  GS_HANDLE_LOCK(ix);
  if (!ix || !text || !path || len != ix->genome_length) return GS_ERR_ARG;
  GS_HIP(hipSetDevice(ix->device));
  FILE *f = fopen(path, "wb");
  if (!f) {
    gs_set_error(std::string("cannot write ") + path);
    return GS_ERR_IO;
  }
  gs_sa_header h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "GSAMDSA2", 8);
  h.n = ix->strand[0].n;
  h.text_hash = text_fingerprint(text, len);
  h.sa_hash[0] = h.sa_hash[1] = 0x9E3779B97F4A7C15ull;
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
  const size_t chunk = 64u << 20; /* entries per copy */
  std::vector<uint32_t> buf;
  try {
    buf.resize(chunk);
  } catch (const std::bad_alloc &) {
    fclose(f);
    return GS_ERR_NOMEM;
  }
  for (int s = 0; s < 2 && ok; s++)
    for (uint64_t at = 0; at < h.n && ok; at += chunk) {
      const size_t m = (size_t)std::min<uint64_t>(chunk, h.n - at);
      if (hipMemcpy(buf.data(), (const uint32_t *)ix->strand[s].sa + at, 4 * m, hipMemcpyDeviceToHost) != hipSuccess) {
        fclose(f);
        gs_set_error("copying the suffix array back failed");
        return GS_ERR_DEVICE;
      }
      ok = fwrite(buf.data(), 4, m, f) == m;
      h.sa_hash[s] = sa_hash_update(h.sa_hash[s], buf.data(), m);
    }
  /* the header again, now with the payload checksums */
  ok = ok && fseek(f, 0, SEEK_SET) == 0 && fwrite(&h, sizeof(h), 1, f) == 1;
  ok = fclose(f) == 0 && ok;
  if (!ok) {
    gs_set_error(std::string("short write to ") + path);
    return GS_ERR_IO;
  }
  return GS_OK;
}
extern "C" gs_status gs_index_open_sa(const uint8_t *text, uint64_t len, const char *path, int device,
                                      gs_index **out) {
  if (!text || !path || !out || len < 1) return GS_ERR_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) {
    gs_set_error(std::string("cannot read ") + path);
    return GS_ERR_IO;
  }
  gs_sa_header h;
  const uint64_t n = len + 1;
  bool ok = fread(&h, sizeof(h), 1, f) == 1 && (!memcmp(h.magic, "GSAMDSA2", 8) || !memcmp(h.magic, "GSAMDSA1", 8)) &&
            h.n == n && h.text_hash == text_fingerprint(text, len);
  const bool summed = ok && h.magic[7] == '2';
  std::vector<uint32_t> sa[2];
  try {
    for (int s = 0; s < 2 && ok; s++) {
      sa[s].resize(n);
      ok = fread(sa[s].data(), 4, n, f) == n;
    }
  } catch (const std::bad_alloc &) {
    fclose(f);
    return GS_ERR_NOMEM;
  }
  fclose(f);
  if (ok && summed) {
    /* a damaged payload of the right length must not reach the builders: both strands' checksums, side by side */
    uint64_t got[2] = {0, 0};
    auto sum = [&](int s) { got[s] = sa_hash_update(0x9E3779B97F4A7C15ull, sa[s].data(), (size_t)n); };
    try {
      std::thread t1(sum, 1);
      sum(0);
      t1.join();
    } catch (const std::system_error &) { /* no second thread to be had */
      sum(0);
      sum(1);
    }
    ok = got[0] == h.sa_hash[0] && got[1] == h.sa_hash[1];
  }
  if (!ok) {
    gs_set_error(std::string(path) + " is not the suffix-array file of this genome text (or is damaged)");
    return GS_ERR_FORMAT;
  }
  /* files without checksums (and any file, again): build_common checks that each array is a permutation */
  return build_common(text, len, sa[0].data(), sa[1].data(), device, out, GS_ERR_FORMAT);
}

extern "C" void gs_index_close(gs_index *ix) {
  if (!ix) return;
  hipSetDevice(ix->device);
  gs_strand_free(&ix->strand[0]);
  gs_strand_free(&ix->strand[1]);
  gs_buffer *bufs[] = {&ix->w_guides, &ix->w_slots, &ix->w_counts, &ix->w_nmatch, &ix->w_nhits,
                       &ix->w_offsets, &ix->w_hits, &ix->w_misc, &ix->w_blocksums, &ix->w_grec, &ix->w_flags, &ix->w_raw,
                       &ix->w_ovf_list, &ix->w_grec2, &ix->w_slots2, &ix->w_counts2, &ix->w_nmatch2,
                       &ix->w_nhits2, &ix->w_h_off, &ix->w_h_tmp, &ix->w_b_src, &ix->w_b_cnt, &ix->w_b_prefix,
                       &ix->w_b_recs, &ix->w_b_w0, &ix->w_b_w0b, &ix->w_b_idx, &ix->w_b_idxb,
                       &ix->w_b_keep, &ix->w_b_keeps, &ix->w_b_rows, &ix->w_b_rowss, &ix->w_b_redo_pos, &ix->w_b_s, &ix->w_b_tab,
                       &ix->w_cand, &ix->rec[0].buf, &ix->rec[1].buf, &ix->w_score, &ix->w_score_io, &ix->w_score_tmp, &ix->w_arena, &ix->w_arena_meta, &ix->w_nchunk, &ix->w_shq, &ix->w_sh_meta,
                       &ix->w_cls, &ix->w_desc, &ix->w_sched, &ix->w_t_plan, &ix->w_t_tiles, &ix->w_t_buckets, &ix->w_t_chunkof, &ix->w_t_big, &ix->w_t_rel, &ix->w_t_tab, &ix->w_t_excl, &ix->w_t_spill, &ix->w_b_redo_pos2};
  for (gs_buffer *b : bufs)
    if (b->p) hipFree(b->p);
  for (int i = 0; i < 4; i++)
    if (ix->ev[i]) hipEventDestroy(ix->ev[i]);
  if (ix->ev_tile) hipEventDestroy(ix->ev_tile);
  for (hipEvent_t e : ix->ev_help)
    if (e) hipEventDestroy(e);
  if (ix->st_help) hipStreamDestroy(ix->st_help);
  if (ix->h_pin) hipHostFree(ix->h_pin);
  gs_pairtab_free(ix, 0);
  gs_pairtab_free(ix, 1);
  delete ix;
}
/* hold the handle across several device-pointer calls (enumerate, score, copies of the results they leave in HBM):
 * other threads' calls on the handle wait */
static uint64_t this_thread_tag() { return (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1ull; }
extern "C" gs_status gs_index_lock(gs_index *ix) {
  if (!ix) return GS_ERR_ARG;
  ix->mtx.lock();
  ix->lock_owner.store(this_thread_tag());
  ix->lock_depth++;
  return GS_OK;
}
/* (unlocking a std::recursive_mutex the thread does not own is undefined behaviour: a thread that does not hold the
 * handle through gs_index_lock gets GS_ERR_ARG instead) */
extern "C" gs_status gs_index_unlock(gs_index *ix) {
  if (!ix) return GS_ERR_ARG;
  if (ix->lock_owner.load() != this_thread_tag() || ix->lock_depth == 0) return GS_ERR_ARG;
  if (--ix->lock_depth == 0) ix->lock_owner.store(0);
  ix->mtx.unlock();
  return GS_OK;
}
extern "C" gs_status gs_index_last_guide_flags(const gs_index *ix, const void **d_flags, uint64_t *n_unsupported) {
  GS_HANDLE_LOCK(ix);
  if (!ix) return GS_ERR_ARG;
  if (d_flags) *d_flags = ix->w_flags.p;
  if (n_unsupported) *n_unsupported = ix->last_unsupported;
  return GS_OK;
}
extern "C" gs_status gs_index_last_counters(const gs_index *ix, uint64_t out[16]) {
  GS_HANDLE_LOCK(ix);
  if (!ix || !out) return GS_ERR_ARG;
  for (int i = 0; i < 16; i++) out[i] = ix->last_counters[i];
  return GS_OK;
}
extern "C" uint64_t gs_index_genome_length(const gs_index *ix) { return ix ? ix->genome_length : 0; }
extern "C" uint64_t gs_index_device_bytes(const gs_index *ix) {
  return ix ? ix->strand[0].bytes + ix->strand[1].bytes + ix->pairtab[0].bytes + ix->pairtab[1].bytes : 0;
}
extern "C" gs_status gs_index_meta(const gs_index *ix, int strand, uint64_t C_acgtn[5],
                                   uint64_t *size) {
  if (!ix || strand < 0 || strand > 1) return GS_ERR_ARG;
  if (C_acgtn)
    for (int k = 0; k < 5; k++) C_acgtn[k] = ix->strand[strand].C_acgtn[k];
  if (size) *size = ix->strand[strand].n;
  return GS_OK;
}
extern "C" gs_status gs_index_copy_sa(gs_index *ix, int strand, uint32_t *out) {
  GS_HANDLE_LOCK(ix);
  if (!ix || strand < 0 || strand > 1 || !out) return GS_ERR_ARG;
  GS_HIP(hipSetDevice(ix->device));
  GS_HIP(hipMemcpy(out, ix->strand[strand].sa, 4 * ix->strand[strand].n, hipMemcpyDeviceToHost));
  return GS_OK;
}

gs_status gs_reserve(gs_buffer &b, size_t bytes) {
  if (b.cap >= bytes && b.p) return GS_OK;
  if (bytes > ((size_t)1 << 30) && gs_debug_any.load(std::memory_order_relaxed))
    fprintf(stderr, "[gs] workspace buffer grows from %.2f to %.2f GiB\n", (double)b.cap / (1 << 30), (double)bytes / (1 << 30));
  const bool again = b.p != nullptr && b.cap > ((size_t)1 << 30);
  if (b.p) hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  /* room to grow without another allocation: a quarter on top, a sixteenth for buffers beyond 1 GiB
   * (the workspace of a repeat-rich batch is tens of GB next to a 220 GB index) - and an eighth when such a buffer grows
   * a second time: batches of one kind differ by several per cent, and freeing and allocating 30 GB takes a second */
  size_t want = bytes + (bytes > ((size_t)1 << 30) ? (again ? bytes / 8 : bytes / 16) : bytes / 4) + 256;
  if (hipMalloc(&b.p, want) != hipSuccess) {
    (void)hipGetLastError(); /* or the next call that reports the last error (rocPRIM does) fails with this one */
    if (hipMalloc(&b.p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      gs_set_error("out of device memory");
      return GS_ERR_NOMEM;
    }
    want = bytes;
  }
  b.cap = want;
  return GS_OK;
}
