/*
 * gs_suffix.hip -- the suffix array of one strand by prefix doubling that leaves sorted suffixes alone.
 *
 * What it replaces: sdsl::construct's suffix sort behind `guidescan index` (src/guidescan.cxx:109-179: divsufsort on
 * the host, about an hour at hg38 size), and - since round 6 - this library's own first builder (gs_index.hip:
 * gs_device_suffix_array), which doubled over ALL n rows every round: 18 rounds of 0.47 s per strand at hg38 size,
 * although after the first sort (21 symbols per key at six distinct bytes) 98.7 % of the rows are alone in their
 * group and will never move again - what keeps the rounds coming are the rows inside runs of N (a run of 2^22 N needs
 * 18 doublings): 1.2 % of bench.py's hg38-sized text, 5 % of the real assembly.
 *
 * The rule (Larsson & Sadakane's discarding, in its sort-everything-that-is-left form): a suffix whose group has one
 * member has its final row.  After each round only the rows of groups with two or more members stay in play
 * (`pos`: their rows, ascending); a round gathers (rank[s], rank[s + h]) for the suffixes at those rows, sorts the
 * pairs - the first word keeps every group in its own stretch of rows -, writes the suffixes back to the same rows in
 * the new order, gives every new group the row of its first member as its rank, and keeps the rows of the groups that
 * still have company.  rank[] of a suffix out of play is its row, which is what a later comparison needs.
 *
 * The result is THE suffix array of the text (there is only one): gs_index_verify_sa proves it row by row from the
 * text alone in the full-size tests, and tests/test_gpu_suffix_array.py compares it with the first builder's.
 * GS_SA_PLAIN=1 on the handle takes the first builder.  Offline step, not on the enumerate hot path; rocPRIM for the
 * plain sorts and scans.
 */
#include "gs_common.h"

#include <rocprim/rocprim.hpp>

static inline unsigned sx_blk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

struct sx_max_op { /* max-scan of group starts */
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

__global__ void k_sx_histogram(const uint8_t *text, uint64_t n, unsigned long long *hist) {
  __shared__ unsigned int s[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s[i] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&s[text[i]], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x)
    if (s[i]) atomicAdd(&hist[i], (unsigned long long)s[i]);
}
/* the first key of suffix i: its first k0 symbols in the text's dense alphabet, `bits` each (past the sentinel: the
 * smallest symbol - such a suffix holds the sentinel, which makes it unique whatever follows) */
__global__ void k_sx_init_keys(const uint8_t *text, uint64_t n, const uint8_t *dense /*256*/, uint32_t bits, uint32_t k0,
                               uint64_t *keys, uint32_t *idx) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
  for (uint32_t j = 0; j < k0; j++) {
    const uint64_t p = i + j;
    key = (key << bits) | (uint64_t)(p < n ? dense[text[p]] : 0);
  }
  keys[i] = key;
  idx[i] = (uint32_t)i;
}
/* head[i] = at[i] (its row; all rows: i itself) where the sorted key differs from the one before, else 0 */
__global__ void k_sx_heads(const uint64_t *keys, const uint32_t *at, uint64_t n, uint32_t *head) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || keys[i] != keys[i - 1]) ? (at ? at[i] : (uint32_t)i) : 0u;
}
__global__ void k_sx_scatter_rank(const uint32_t *sa, const uint32_t *grp, uint64_t n, uint32_t *rank) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rank[sa[i]] = grp[i];
}
/* flag[i] = 1: element i's group (grp: equal for the members of a group, which are neighbours) has company */
__global__ void k_sx_company(const uint32_t *grp, uint64_t n, uint32_t *flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = grp[i];
  flag[i] = ((i > 0 && grp[i - 1] == g) || (i + 1 < n && grp[i + 1] == g)) ? 1u : 0u;
}
/* the rows that stay in play, ascending: out[idx[i]] = at[i] (all rows: i) where flag[i] */
__global__ void k_sx_compact(const uint32_t *flag, const uint32_t *idx, const uint32_t *at, uint64_t n, uint32_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) out[idx[i]] = at ? at[i] : (uint32_t)i;
}
/* (rank of the suffix at row pos[j], rank of the suffix h symbols further) and the suffix itself */
__global__ void k_sx_pair_keys(const uint32_t *pos, const uint32_t *sa, const uint32_t *rank, uint64_t n, uint64_t h, uint32_t nbits,
                               uint64_t n_act, uint64_t *keys, uint32_t *vals) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_act) return;
  const uint64_t s = sa[pos[j]];
  const uint64_t r1 = rank[s];
  const uint64_t r2 = (s + h < n) ? (uint64_t)rank[s + h] : 0ull;
  keys[j] = (r1 << nbits) | r2;
  vals[j] = (uint32_t)s;
}
/* the sorted suffixes back to the same rows, each with the first row of its new group as its rank */
__global__ void k_sx_apply(const uint32_t *vals, const uint32_t *ngrp, const uint32_t *pos, uint64_t n_act, uint32_t *sa, uint32_t *rank) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_act) return;
  const uint32_t s = vals[j];
  sa[pos[j]] = s;
  rank[s] = ngrp[j];
}

namespace {
struct sx_buffers { /* everything the builder allocates: released on every way out */
  void *p[16] = {nullptr};
  int n = 0;
  template <typename T>
  hipError_t get(T **out, size_t bytes) {
    void *q = nullptr;
    *out = nullptr;
    if (n >= 16) return hipErrorOutOfMemory; /* (the list is sized for what the builder allocates: 13 at most) */
    const hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
    if (e == hipSuccess) p[n++] = q;
    *out = (T *)q;
    return e;
  }
  void drop(void *q) {
    for (int i = 0; i < n; i++)
      if (p[i] == q) {
        hipFree(q);
        p[i] = p[--n];
        return;
      }
  }
  ~sx_buffers() {
    for (int i = 0; i < n; i++) hipFree(p[i]);
  }
};
}  // namespace

gs_status gs_device_suffix_array_discarding(const uint8_t *d_text, uint64_t n, uint32_t *d_sa, hipStream_t st, bool debug) {
  sx_buffers B;
  /* dense alphabet: as many symbols per first key as 64 bits hold */
  unsigned long long *d_hist = nullptr;
  GS_HIP(B.get(&d_hist, 256 * 8));
  GS_HIP(hipMemsetAsync(d_hist, 0, 256 * 8, st));
  hipLaunchKernelGGL(k_sx_histogram, dim3(1024), dim3(256), 0, st, d_text, n, d_hist);
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
  GS_HIP(B.get(&d_dense, 256));
  GS_HIP(hipMemcpy(d_dense, dense, 256, hipMemcpyHostToDevice));
  GS_HIP(B.get(&keys_a, 8 * n));
  GS_HIP(B.get(&keys_b, 8 * n));
  GS_HIP(B.get(&idx_a, 4 * n));
  GS_HIP(B.get(&rank, 4 * n));
  GS_HIP(B.get(&head, 4 * n));
  const unsigned g = sx_blk(n, 256);
  hipLaunchKernelGGL(k_sx_init_keys, dim3(g), dim3(256), 0, st, d_text, n, d_dense, bits, k0, keys_a, idx_a);
  size_t sort_bytes = 0, scan_bytes = 0, xscan_bytes = 0;
  GS_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_a, keys_b, idx_a, d_sa, n, 0, 64, st));
  GS_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, head, head, n, sx_max_op(), st));
  GS_HIP(rocprim::exclusive_scan(nullptr, xscan_bytes, head, head, 0u, n, rocprim::plus<uint32_t>(), st));
  void *tmp = nullptr;
  const size_t tmp_bytes = std::max(sort_bytes, std::max(scan_bytes, xscan_bytes));
  GS_HIP(B.get(&tmp, tmp_bytes));
  size_t sb = tmp_bytes;
  /* round 0, all rows: sort by the first k0 symbols; groups, ranks */
  GS_HIP(rocprim::radix_sort_pairs(tmp, sb, keys_a, keys_b, idx_a, d_sa, n, 0, bits * k0, st));
  hipLaunchKernelGGL(k_sx_heads, dim3(g), dim3(256), 0, st, keys_b, (const uint32_t *)nullptr, n, head);
  sb = tmp_bytes;
  GS_HIP(rocprim::inclusive_scan(tmp, sb, head, head, n, sx_max_op(), st));
  hipLaunchKernelGGL(k_sx_scatter_rank, dim3(g), dim3(256), 0, st, d_sa, head, n, rank);
  /* the rows in play: flag -> idx_a, places -> keys_a (as 32-bit words) */
  uint32_t *flag = idx_a, *place = (uint32_t *)keys_a;
  hipLaunchKernelGGL(k_sx_company, dim3(g), dim3(256), 0, st, head, n, flag);
  sb = tmp_bytes;
  GS_HIP(rocprim::exclusive_scan(tmp, sb, flag, place, 0u, n, rocprim::plus<uint32_t>(), st));
  uint32_t h_last[2] = {0, 0};
  GS_HIP(hipMemcpyAsync(&h_last[0], place + (n - 1), 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&h_last[1], flag + (n - 1), 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  uint64_t n_act = (uint64_t)h_last[0] + h_last[1];
  if (debug) fprintf(stderr, "[gs] suffix array: %llu rows, %u symbols per first key, %llu rows in groups with company\n", (unsigned long long)n, k0, (unsigned long long)n_act);
  if (n_act == 0) return GS_OK;
  uint32_t *pos = nullptr, *pos2 = nullptr;
  GS_HIP(B.get(&pos, 4 * n_act));
  hipLaunchKernelGGL(k_sx_compact, dim3(g), dim3(256), 0, st, flag, place, (const uint32_t *)nullptr, n, pos);
  GS_HIP(hipStreamSynchronize(st));
  /* the arrays of the rows in play take the place of the arrays of all rows */
  B.drop(keys_a);
  B.drop(keys_b);
  B.drop(idx_a);
  B.drop(head);
  uint64_t *keys = nullptr, *keys2 = nullptr;
  uint32_t *vals = nullptr, *vals2 = nullptr, *ngrp = nullptr, *cflag = nullptr, *cplace = nullptr;
  GS_HIP(B.get(&keys, 8 * n_act));
  GS_HIP(B.get(&keys2, 8 * n_act));
  GS_HIP(B.get(&vals, 4 * n_act));
  GS_HIP(B.get(&vals2, 4 * n_act));
  GS_HIP(B.get(&ngrp, 4 * n_act));
  GS_HIP(B.get(&cflag, 4 * n_act));
  GS_HIP(B.get(&cplace, 4 * n_act));
  GS_HIP(B.get(&pos2, 4 * n_act));
  uint32_t rounds = 0;
  for (uint64_t h = k0; n_act != 0; h *= 2) {
    if (h >= 2 * n) {
      gs_set_error("internal: suffix array doubling did not converge");
      return GS_ERR_DEVICE;
    }
    const unsigned ga = sx_blk(n_act, 256);
    hipLaunchKernelGGL(k_sx_pair_keys, dim3(ga), dim3(256), 0, st, pos, d_sa, rank, n, h, nbits, n_act, keys, vals);
    sb = tmp_bytes;
    GS_HIP(rocprim::radix_sort_pairs(tmp, sb, keys, keys2, vals, vals2, n_act, 0, 2 * nbits, st));
    hipLaunchKernelGGL(k_sx_heads, dim3(ga), dim3(256), 0, st, keys2, pos, n_act, ngrp);
    sb = tmp_bytes;
    GS_HIP(rocprim::inclusive_scan(tmp, sb, ngrp, ngrp, n_act, sx_max_op(), st));
    hipLaunchKernelGGL(k_sx_apply, dim3(ga), dim3(256), 0, st, vals2, ngrp, pos, n_act, d_sa, rank);
    hipLaunchKernelGGL(k_sx_company, dim3(ga), dim3(256), 0, st, ngrp, n_act, cflag);
    sb = tmp_bytes;
    GS_HIP(rocprim::exclusive_scan(tmp, sb, cflag, cplace, 0u, n_act, rocprim::plus<uint32_t>(), st));
    GS_HIP(hipMemcpyAsync(&h_last[0], cplace + (n_act - 1), 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipMemcpyAsync(&h_last[1], cflag + (n_act - 1), 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_sx_compact, dim3(ga), dim3(256), 0, st, cflag, cplace, pos, n_act, pos2);
    GS_HIP(hipStreamSynchronize(st));
    n_act = (uint64_t)h_last[0] + h_last[1];
    std::swap(pos, pos2);
    rounds++;
    if (debug) fprintf(stderr, "[gs] suffix array: round %u (h = %llu): %llu rows still in play\n", rounds, (unsigned long long)h, (unsigned long long)n_act);
  }
  GS_HIP(hipGetLastError());
  return GS_OK;
}
