/*
 * gs_verify.hip -- checks of a device-resident index that use nothing but the genome text and the
 * stored arrays (no suffix-array builder, no oracle): what the parity tests run at sizes where no
 * CPU construction is at hand (n > 2^31).
 *
 *  - the suffix array is a permutation of [0, n): every value in range, none twice (bitmap, all rows)
 *  - adjacent rows are in suffix order: text[sa[r]..] < text[sa[r+1]..] by direct comparison of the
 *    text, on sampled rows (runs of 'N' megabases long are skipped through the run list)
 *  - the BWT symbol held in the Occ block of a sampled row is text[sa[r]-1]
 * The reference's counterpart is sdsl's own construction (sdsl/include/sdsl/construct_sa.hpp via
 * divsufsort), checked there by sdsl/test/csa_byte_test.cpp on small texts.
 */
#include "gs_device.h"

#include <algorithm>

struct gs_vrun {
  uint64_t start, end; /* [start, end) is a maximal run of 'N' in this strand's text */
};

__global__ void k_v_revcomp(const uint8_t *in, uint8_t *out, uint64_t len) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  uint8_t c = in[len - 1 - i];
  c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c; /* sequences.cxx:14-26 (text is upper case) */
  out[i] = c;
}

__global__ void k_v_permutation(const uint32_t *sa, uint64_t n, uint32_t *bitmap, unsigned long long *bad) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t v = sa[r];
  bool b = v >= n;
  if (!b) {
    const uint32_t bit = 1u << (v & 31u);
    b = (atomicOr(&bitmap[v >> 5], bit) & bit) != 0u; /* seen before */
  }
  if (b) atomicAdd(bad, 1ull);
}

__device__ __forceinline__ uint64_t v_hash(uint64_t x) { /* splitmix64 */
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* symbols of the run holding position p that lie at or after p (0: p is not inside a run) */
__device__ __forceinline__ uint64_t v_run_left(const gs_vrun *runs, uint32_t nruns, uint64_t p) {
  uint32_t lo = 0, hi = nruns; /* last run with start <= p */
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (runs[mid].start <= p)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (!lo) return 0;
  const gs_vrun r = runs[lo - 1];
  return p < r.end ? r.end - p : 0;
}

/* rows of a device-resident suffix array that are out of range or repeat a value (0: a permutation of [0, n)).
 * What a suffix array that did not come from this library's own sort is put through before any table is
 * derived from it (gs_index_open_sa, gs_index_build_with_sa): the builders read text[sa[r] - j] unchecked. */
gs_status gs_count_bad_sa_rows(const uint32_t *d_sa, uint64_t n, hipStream_t st, uint64_t *bad) {
  uint32_t *bitmap = nullptr;
  unsigned long long *d_bad = nullptr;
  const size_t words = (size_t)((n + 31) / 32);
  if (hipMalloc(&bitmap, 4 * words + 16) != hipSuccess) { /* the bitmap, then the 8-byte aligned counter */
    (void)hipGetLastError();
    return GS_ERR_NOMEM;
  }
  d_bad = (unsigned long long *)(bitmap + ((words + 1) & ~(size_t)1));
  hipError_t e = hipMemsetAsync(bitmap, 0, 4 * words + 16, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_v_permutation, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_sa, n, bitmap, d_bad);
    unsigned long long h = 0;
    e = hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    *bad = h;
  }
  hipFree(bitmap);
  if (e != hipSuccess) {
    gs_set_error(std::string("suffix-array check: ") + hipGetErrorString(e));
    return GS_ERR_DEVICE;
  }
  return GS_OK;
}

struct gs_vorder_args {
  const uint8_t *text; /* n bytes: the strand's text and the 0 sentinel */
  const uint32_t *sa;
  const uint4 *blocks;
  const gs_vrun *runs;
  unsigned long long *out; /* [0] out of order, [1] undecided, [2] BWT symbol differs */
  uint64_t n, samples, seed, max_steps;
  uint32_t nruns;
};

__global__ void k_v_order(gs_vorder_args a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.samples) return;
  /* samples cover the array evenly: stride + jitter inside the stride */
  const uint64_t stride = (a.n - 1) / a.samples ? (a.n - 1) / a.samples : 1;
  uint64_t r = i * stride + v_hash(i ^ a.seed) % stride;
  if (r + 1 >= a.n) r = a.n - 2;
  const uint64_t pa = a.sa[r], pb = a.sa[r + 1];
  /* values out of range (what the verifier is there to diagnose) must not be used as text offsets */
  if (pa >= a.n || pb >= a.n || pa == pb) {
    atomicAdd(&a.out[0], 1ull);
    return;
  }
  /* BWT symbol of row r in the Occ block against the text (k_build_words, gs_index.hip) */
  {
    const uint32_t *b = (const uint32_t *)(a.blocks + (r >> GS_BLOCK_SHIFT) * 4);
    const uint32_t w = (uint32_t)(r & 127u) >> 5, j = (uint32_t)r & 31u;
    const uint32_t lo = (b[4 + w] >> j) & 1u, hi = (b[8 + w] >> j) & 1u, ex = (b[12 + w] >> j) & 1u;
    const uint8_t c = pa ? a.text[pa - 1] : a.text[a.n - 1];
    const int cls = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
    const bool ok = cls < 0 ? ex == 1u : (ex == 0u && (lo | (hi << 1)) == (uint32_t)cls);
    if (!ok) atomicAdd(&a.out[2], 1ull);
  }
  uint64_t x = pa, y = pb;
  for (uint64_t step = 0; step < a.max_steps; ++step) {
    /* the sentinel is unique and smallest: a suffix that reaches it first is the smaller one */
    const uint8_t cx = a.text[x], cy = a.text[y];
    if (cx != cy) {
      if (cx > cy) atomicAdd(&a.out[0], 1ull);
      return;
    }
    if (cx == 0) { /* both at the sentinel: the same suffix twice */
      atomicAdd(&a.out[0], 1ull);
      return;
    }
    uint64_t adv = 1;
    if (cx == 'N' && a.nruns) {
      const uint64_t lx = v_run_left(a.runs, a.nruns, x), ly = v_run_left(a.runs, a.nruns, y);
      const uint64_t m = lx < ly ? lx : ly;
      if (m > 1) adv = m;
    }
    x += adv;
    y += adv;
  }
  atomicAdd(&a.out[1], 1ull);
}

/* ---- every row (n_samples = GS_VERIFY_ALL_ROWS): the linear-time check of a suffix array against its text.  With SA a
 * permutation of [0, n) (k_v_permutation) and ISA its inverse (isa[sa[r]] == r, checked here), SA is the suffix array iff
 * for every r: text[sa[r]] < text[sa[r+1]], or the two symbols are equal and isa[sa[r]+1] < isa[sa[r+1]+1] - the order of
 * two suffixes that start alike is the order of what follows them, which the array itself states (induction over the
 * length of the common prefix; the sentinel is unique and smallest).  One streaming pass, five random reads per row;
 * nothing undecided, no sampling.  What sdsl's own tests do for csa_wt on small texts (sdsl/test/csa_byte_test.cpp), and
 * what csa_wt::operator[] (sdsl/include/sdsl/csa_wt.hpp:333-346) presumes. ---- */
struct gs_vfull_args {
  const uint8_t *text;
  const uint32_t *sa, *isa;
  const uint4 *blocks;
  unsigned long long *out; /* [0] out of order, [2] BWT symbol differs, [3] isa is not the inverse */
  uint64_t n;
};
__global__ void k_v_full(gs_vfull_args a) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n) return;
  const uint64_t pa = a.sa[r];
  if (pa >= a.n) { /* (counted by k_v_permutation; not to be used as an offset) */
    atomicAdd(&a.out[0], 1ull);
    return;
  }
  if (a.isa[pa] != (uint32_t)r) atomicAdd(&a.out[3], 1ull);
  {
    const uint32_t *b = (const uint32_t *)(a.blocks + (r >> GS_BLOCK_SHIFT) * 4);
    const uint32_t w = (uint32_t)(r & 127u) >> 5, j = (uint32_t)r & 31u;
    const uint32_t lo = (b[4 + w] >> j) & 1u, hi = (b[8 + w] >> j) & 1u, ex = (b[12 + w] >> j) & 1u;
    const uint8_t c = pa ? a.text[pa - 1] : a.text[a.n - 1];
    const int cls = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
    const bool ok = cls < 0 ? ex == 1u : (ex == 0u && (lo | (hi << 1)) == (uint32_t)cls);
    if (!ok) atomicAdd(&a.out[2], 1ull);
  }
  if (r + 1 >= a.n) return;
  const uint64_t pb = a.sa[r + 1];
  if (pb >= a.n || pb == pa) {
    atomicAdd(&a.out[0], 1ull);
    return;
  }
  const uint8_t ca = a.text[pa], cb = a.text[pb];
  bool bad = ca > cb;
  if (ca == cb) /* (equal and not the sentinel: neither suffix is the last one, pa + 1 and pb + 1 are positions) */
    bad = ca == 0 || pa + 1 >= a.n || pb + 1 >= a.n || !(a.isa[pa + 1] < a.isa[pb + 1]);
  if (bad) atomicAdd(&a.out[0], 1ull);
}

extern "C" gs_status gs_index_verify_sa(gs_index *ix, int strand, const uint8_t *text, uint64_t len,
                                        uint64_t n_samples, uint64_t seed, gs_sa_report *rep) {
  GS_HANDLE_LOCK(ix);
  if (!ix || strand < 0 || strand > 1 || !text || !rep || len != ix->genome_length) return GS_ERR_ARG;
  GS_HIP(hipSetDevice(ix->device));
  const gs_strand &s = ix->strand[strand];
  const uint64_t n = s.n;
  memset(rep, 0, sizeof(*rep));
  rep->rows = n;
  struct dbuf {
    void *p = nullptr;
    ~dbuf() {
      if (p) hipFree(p);
    }
  } d_text, d_tmp, d_bitmap, d_out, d_runs;
  GS_HIP(hipMalloc(&d_text.p, n));
  if (strand == 0) {
    GS_HIP(hipMemcpy(d_text.p, text, len, hipMemcpyHostToDevice));
  } else {
    GS_HIP(hipMalloc(&d_tmp.p, len));
    GS_HIP(hipMemcpy(d_tmp.p, text, len, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_v_revcomp, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, 0, (const uint8_t *)d_tmp.p,
                       (uint8_t *)d_text.p, len);
  }
  GS_HIP(hipMemset((uint8_t *)d_text.p + len, 0, 1));
  /* N runs of this strand's text: the forward runs, mirrored for the reverse strand */
  std::vector<gs_vrun> runs;
  for (const gs_nrun &r : ix->nruns_text) {
    gs_vrun v;
    if (strand == 0) {
      v.start = r.start;
      v.end = r.start + r.len;
    } else {
      v.start = len - (r.start + r.len);
      v.end = len - r.start;
    }
    runs.push_back(v);
  }
  std::sort(runs.begin(), runs.end(), [](const gs_vrun &a, const gs_vrun &b) { return a.start < b.start; });
  if (!runs.empty()) {
    GS_HIP(hipMalloc(&d_runs.p, sizeof(gs_vrun) * runs.size()));
    GS_HIP(hipMemcpy(d_runs.p, runs.data(), sizeof(gs_vrun) * runs.size(), hipMemcpyHostToDevice));
  }
  const size_t words = (size_t)((n + 31) / 32);
  GS_HIP(hipMalloc(&d_bitmap.p, 4 * words));
  GS_HIP(hipMemset(d_bitmap.p, 0, 4 * words));
  GS_HIP(hipMalloc(&d_out.p, 64));
  GS_HIP(hipMemset(d_out.p, 0, 64));
  unsigned long long *out = (unsigned long long *)d_out.p;
  hipLaunchKernelGGL(k_v_permutation, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint32_t *)s.sa, n,
                     (uint32_t *)d_bitmap.p, out + 3);
  if (n_samples == GS_VERIFY_ALL_ROWS) {
    if (!s.isa) {
      gs_set_error("gs_index_verify_sa: the check of every row reads the inverse suffix array, which this index was built without");
      return GS_ERR_UNSUPPORTED;
    }
    gs_vfull_args a;
    a.text = (const uint8_t *)d_text.p;
    a.sa = (const uint32_t *)s.sa;
    a.isa = (const uint32_t *)s.isa;
    a.blocks = (const uint4 *)s.blocks;
    a.out = out;
    a.n = n;
    hipLaunchKernelGGL(k_v_full, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a);
    n_samples = n - 1;
    unsigned long long h[4] = {0, 0, 0, 0};
    GS_HIP(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    GS_HIP(hipGetLastError());
    rep->sampled = n_samples;
    rep->out_of_order = h[0];
    rep->undecided = 0;
    rep->bwt_mismatch = h[2];
    rep->not_permutation = h[3]; /* repeated / out-of-range values (the bitmap) + rows whose isa entry is not their own */
    return GS_OK;
  }
  if (n_samples > n - 1) n_samples = n - 1;
  if (n_samples) {
    gs_vorder_args a;
    a.text = (const uint8_t *)d_text.p;
    a.sa = (const uint32_t *)s.sa;
    a.blocks = (const uint4 *)s.blocks;
    a.runs = (const gs_vrun *)d_runs.p;
    a.out = out;
    a.n = n;
    a.samples = n_samples;
    a.seed = seed;
    a.max_steps = 1u << 16;
    a.nruns = (uint32_t)runs.size();
    hipLaunchKernelGGL(k_v_order, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, 0, a);
  }
  unsigned long long h[4] = {0, 0, 0, 0};
  GS_HIP(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  GS_HIP(hipGetLastError());
  rep->sampled = n_samples;
  rep->out_of_order = h[0];
  rep->undecided = h[1];
  rep->bwt_mismatch = h[2];
  rep->not_permutation = h[3];
  return GS_OK;
}
