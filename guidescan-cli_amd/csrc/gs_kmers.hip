/*
 * gs_kmers.hip -- candidate-guide generation on the device: the
 * scan of one chromosome for every PAM site, in the order the reference's helper script emits
 * them (scripts/generate_kmers.py:70-118): for each concrete expansion of the PAM pattern
 * ('N' -> A,C,T,G, first N first) every occurrence on the + strand in position order, then for
 * each reverse-complemented expansion every occurrence on the - strand; protospacers with a
 * symbol outside A,C,G,T or cut by a chromosome end are dropped; positions are 1-based.
 *
 * Three streaming kernels and two library primitives (this is the step BEFORE the hot path):
 *   k_km_count   sites per 256-position block                       (1 B read per position)
 *   (rocprim exclusive scan of the block counts)
 *   k_km_keys    key = bucket << 32 | position, compacted in position order
 *   (rocprim radix sort of the keys: bucket, then position)
 *   k_km_emit    one thread per site: protospacer (reverse-complemented on the - strand),
 *                position, sense, PAM pattern.
 * The output arrays stay in HBM in exactly the layout gs_enumerate_device takes.
 */
#include "gs_common.h"

#include <rocprim/rocprim.hpp>

#define KM_BLOCK 256

struct gs_kmers {
  int device = 0;
  uint64_t n = 0;
  uint32_t k = 0, P = 0;
  void *d_seqs = nullptr, *d_pams = nullptr, *d_pos = nullptr, *d_sense = nullptr;
  std::vector<uint8_t> h_seqs, h_pams, h_sense;
  std::vector<uint32_t> h_pos;
  bool have_host = false;
};

struct gs_km_args {
  const uint8_t *chr;
  const uint16_t *lut; /* 4^P entries: low byte = 1 + id of the matching + strand expansion, high byte: - strand */
  uint64_t len;
  uint32_t k, P, n_exp, start;
};

__device__ __forceinline__ int km_code(uint32_t c) {
  if (c >= 'a' && c <= 'z') c -= 32u; /* chrm.upper(), scripts/generate_kmers.py:103 */
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}
__device__ __forceinline__ bool km_acgt(const uint8_t *p, uint32_t k) {
  bool ok = true;
  for (uint32_t j = 0; j < k; ++j) ok = ok && km_code(p[j]) >= 0;
  return ok;
}
/* the sites whose PAM starts at idx: bit 0 = + strand (bucket fid-1), bit 1 = - strand (bucket rid-1) */
__device__ __forceinline__ uint32_t km_sites(const gs_km_args &a, uint64_t idx, uint32_t &fid, uint32_t &rid) {
  fid = rid = 0;
  if (idx + a.P > a.len) return 0u;
  uint32_t w = 0;
  for (uint32_t j = 0; j < a.P; ++j) {
    const int c = km_code(a.chr[idx + j]);
    if (c < 0) return 0u;
    w = (w << 2) | (uint32_t)c;
  }
  const uint32_t e = a.lut[w];
  uint32_t m = 0;
  /* scripts/generate_kmers.py:79-93: protospacer before the PAM (+ strand with the PAM at the
   * end, - strand with the PAM at the start) or after it */
  const bool has_before = idx >= a.k;               /* position = idx - k >= 0 */
  const bool has_after = idx + a.P + a.k <= a.len;  /* len(kmer) == k */
  if (e & 0xFFu) {
    const bool before = !a.start;
    if (before ? (has_before && km_acgt(a.chr + idx - a.k, a.k)) : (has_after && km_acgt(a.chr + idx + a.P, a.k))) {
      m |= 1u;
      fid = e & 0xFFu;
    }
  }
  if (e >> 8) {
    const bool before = a.start != 0u;
    if (before ? (has_before && km_acgt(a.chr + idx - a.k, a.k)) : (has_after && km_acgt(a.chr + idx + a.P, a.k))) {
      m |= 2u;
      rid = e >> 8;
    }
  }
  return m;
}

__global__ __launch_bounds__(KM_BLOCK) void k_km_count(gs_km_args a, uint32_t *block_cnt) {
  __shared__ uint32_t s[KM_BLOCK / 64];
  const uint64_t idx = (uint64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
  uint32_t f, r;
  const uint32_t m = idx < a.len ? km_sites(a, idx, f, r) : 0u;
  uint32_t c = (m & 1u) + (m >> 1);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63u) == 0u) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(KM_BLOCK) void k_km_keys(gs_km_args a, const uint32_t *block_off, uint64_t *keys) {
  __shared__ uint32_t s[KM_BLOCK / 64];
  const uint64_t idx = (uint64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t f, r;
  const uint32_t m = idx < a.len ? km_sites(a, idx, f, r) : 0u;
  const uint32_t c = (m & 1u) + (m >> 1);
  uint32_t inc = c; /* inclusive wave scan */
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o);
    if ((int)lane >= o) inc += up;
  }
  if (lane == 63u) s[wv] = inc;
  __syncthreads();
  uint32_t base = block_off[blockIdx.x];
  for (uint32_t w = 0; w < wv; ++w) base += s[w];
  uint32_t at = base + inc - c;
  /* inside one position the + strand site comes first; the sort separates the buckets anyway */
  if (m & 1u) keys[at++] = ((uint64_t)(f - 1u) << 32) | idx;
  if (m & 2u) keys[at] = ((uint64_t)(a.n_exp + r - 1u) << 32) | idx;
}

struct gs_km_emit_args {
  const uint8_t *chr;
  const uint64_t *keys;
  uint8_t *seqs, *pams, *sense;
  uint32_t *pos;
  uint64_t n;
  uint32_t k, P, n_exp, start;
  uint8_t pam[8];
};
__global__ __launch_bounds__(KM_BLOCK) void k_km_emit(gs_km_emit_args a) {
  const uint64_t j = (uint64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
  if (j >= a.n) return;
  const uint64_t key = a.keys[j];
  const uint64_t idx = key & 0xFFFFFFFFull;
  const bool minus = (uint32_t)(key >> 32) >= a.n_exp;
  const bool before = minus ? (a.start != 0u) : (a.start == 0u);
  const uint64_t s0 = before ? idx - a.k : idx + a.P;
  uint8_t *o = a.seqs + j * a.k;
  for (uint32_t t = 0; t < a.k; ++t) {
    uint32_t c = minus ? a.chr[s0 + (a.k - 1u - t)] : a.chr[s0 + t];
    if (c >= 'a' && c <= 'z') c -= 32u;
    if (minus) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C'; /* revcom, :52-53 */
    o[t] = (uint8_t)c;
  }
  for (uint32_t u = 0; u < a.P; ++u) a.pams[j * a.P + u] = a.pam[u]; /* the pattern, :110,117 */
  a.pos[j] = (uint32_t)((before ? idx - a.k : idx) + 1ull); /* 1-based, :99 */
  a.sense[j] = minus ? '-' : '+';
}

static void km_release(gs_kmers *km) {
  if (!km) return;
  hipSetDevice(km->device);
  for (void *p : {km->d_seqs, km->d_pams, km->d_pos, km->d_sense})
    if (p) hipFree(p);
  delete km;
}

extern "C" gs_status gs_kmers_generate(int device, const uint8_t *chr, uint64_t chr_len, int chr_on_device,
                                       const char *pam, uint32_t k, uint32_t flags, void *stream,
                                       gs_kmers **out) {
  if (!out || (chr_len && !chr) || !pam) return GS_ERR_ARG;
  const uint32_t P = (uint32_t)strlen(pam);
  if (k < 1 || k > 64 || P < 1 || P > 8 || chr_len >= (1ull << 32) - 64) {
    gs_set_error("kmer generation supports 1<=k<=64, 1<=P<=8, chromosomes shorter than 2^32");
    return GS_ERR_UNSUPPORTED;
  }
  /* generate_pam_set (scripts/generate_kmers.py:55-68): breadth-first replacement of the first
   * N by A,C,T,G == all N positions left to right as digits in the order A,C,T,G */
  uint32_t npos[8], nn = 0;
  for (uint32_t u = 0; u < P; u++) {
    const char c = pam[u];
    if (c == 'N')
      npos[nn++] = u;
    else if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
      gs_set_error("PAM symbols outside A,C,G,T,N are not expanded by the reference script and not implemented here");
      return GS_ERR_UNSUPPORTED;
    }
  }
  if (nn > 3) {
    gs_set_error("more than three N in the PAM pattern");
    return GS_ERR_UNSUPPORTED;
  }
  const uint32_t n_exp = 1u << (2 * nn);
  auto code = [](char c) -> uint32_t { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; };
  static const char NUCS[4] = {'A', 'C', 'T', 'G'}; /* :49 */
  std::vector<uint16_t> lut((size_t)1 << (2 * P), 0);
  for (uint32_t e = 0; e < n_exp; e++) {
    char p[9];
    memcpy(p, pam, P);
    for (uint32_t i = 0; i < nn; i++) p[npos[i]] = NUCS[(e >> (2 * (nn - 1 - i))) & 3u];
    uint32_t wf = 0, wr = 0;
    for (uint32_t u = 0; u < P; u++) wf = (wf << 2) | code(p[u]);
    for (uint32_t u = 0; u < P; u++) wr = (wr << 2) | (3u - code(p[P - 1 - u])); /* revcom(p) */
    lut[wf] = (uint16_t)((lut[wf] & 0xFF00u) | (e + 1u));
    lut[wr] = (uint16_t)((lut[wr] & 0x00FFu) | ((e + 1u) << 8));
  }

  hipStream_t st = (hipStream_t)stream;
  GS_HIP(hipSetDevice(device));
  gs_kmers *km = new gs_kmers();
  km->device = device;
  km->k = k;
  km->P = P;
  if (chr_len == 0) {
    *out = km;
    return GS_OK;
  }
  const uint32_t nb = (uint32_t)((chr_len + KM_BLOCK - 1) / KM_BLOCK);
  void *d_chr_own = nullptr, *d_lut = nullptr, *d_cnt = nullptr, *d_off = nullptr, *d_keys = nullptr,
       *d_keys2 = nullptr, *d_tmp = nullptr;
  auto cleanup = [&]() {
    for (void *p : {d_chr_own, d_lut, d_cnt, d_off, d_keys, d_keys2, d_tmp})
      if (p) hipFree(p);
  };
#define KM_HIP(expr)                                                        \
  do {                                                                      \
    hipError_t e__ = (expr);                                                \
    if (e__ != hipSuccess) {                                                \
      gs_set_error(std::string(#expr) + ": " + hipGetErrorString(e__));     \
      cleanup();                                                            \
      km_release(km);                                                       \
      return GS_ERR_DEVICE;                                                 \
    }                                                                       \
  } while (0)
  const uint8_t *d_chr = chr;
  if (!chr_on_device) {
    KM_HIP(hipMalloc(&d_chr_own, chr_len));
    KM_HIP(hipMemcpyAsync(d_chr_own, chr, chr_len, hipMemcpyHostToDevice, st));
    d_chr = (const uint8_t *)d_chr_own;
  }
  KM_HIP(hipMalloc(&d_lut, 2 * lut.size()));
  KM_HIP(hipMemcpyAsync(d_lut, lut.data(), 2 * lut.size(), hipMemcpyHostToDevice, st));
  KM_HIP(hipMalloc(&d_cnt, 4 * ((size_t)nb + 1)));
  KM_HIP(hipMalloc(&d_off, 4 * ((size_t)nb + 1)));
  KM_HIP(hipMemsetAsync(d_cnt, 0, 4 * ((size_t)nb + 1), st));
  gs_km_args a;
  a.chr = d_chr;
  a.lut = (const uint16_t *)d_lut;
  a.len = chr_len;
  a.k = k;
  a.P = P;
  a.n_exp = n_exp;
  a.start = (flags & GS_FLAG_PAM_AT_START) ? 1u : 0u;
  hipLaunchKernelGGL(k_km_count, dim3(nb), dim3(KM_BLOCK), 0, st, a, (uint32_t *)d_cnt);
  size_t tb = 0;
  KM_HIP(rocprim::exclusive_scan(nullptr, tb, (uint32_t *)d_cnt, (uint32_t *)d_off, 0u, (size_t)nb + 1,
                                 rocprim::plus<uint32_t>(), st));
  KM_HIP(hipMalloc(&d_tmp, tb + 16));
  KM_HIP(rocprim::exclusive_scan(d_tmp, tb, (uint32_t *)d_cnt, (uint32_t *)d_off, 0u, (size_t)nb + 1,
                                 rocprim::plus<uint32_t>(), st));
  uint32_t total = 0;
  KM_HIP(hipMemcpyAsync(&total, (uint32_t *)d_off + nb, 4, hipMemcpyDeviceToHost, st));
  KM_HIP(hipStreamSynchronize(st));
  km->n = total;
  if (total) {
    KM_HIP(hipMalloc(&d_keys, 8 * (size_t)total));
    KM_HIP(hipMalloc(&d_keys2, 8 * (size_t)total));
    hipLaunchKernelGGL(k_km_keys, dim3(nb), dim3(KM_BLOCK), 0, st, a, (const uint32_t *)d_off, (uint64_t *)d_keys);
    /* bucket (1 + 2*nn bits above bit 32), then position: the whole key, so the order does not
     * lean on the sort being stable */
    const unsigned end_bit = 32u + 1u + 2u * nn;
    size_t sb = 0;
    KM_HIP(rocprim::radix_sort_keys(nullptr, sb, (uint64_t *)d_keys, (uint64_t *)d_keys2, (size_t)total, 0u,
                                    end_bit, st));
    if (sb > tb) {
      hipFree(d_tmp);
      d_tmp = nullptr;
      KM_HIP(hipMalloc(&d_tmp, sb + 16));
    }
    KM_HIP(rocprim::radix_sort_keys(d_tmp, sb, (uint64_t *)d_keys, (uint64_t *)d_keys2, (size_t)total, 0u,
                                    end_bit, st));
    KM_HIP(hipMalloc(&km->d_seqs, (size_t)total * k));
    KM_HIP(hipMalloc(&km->d_pams, (size_t)total * P));
    KM_HIP(hipMalloc(&km->d_pos, 4 * (size_t)total));
    KM_HIP(hipMalloc(&km->d_sense, (size_t)total));
    gs_km_emit_args ea;
    ea.chr = d_chr;
    ea.keys = (const uint64_t *)d_keys2;
    ea.seqs = (uint8_t *)km->d_seqs;
    ea.pams = (uint8_t *)km->d_pams;
    ea.sense = (uint8_t *)km->d_sense;
    ea.pos = (uint32_t *)km->d_pos;
    ea.n = total;
    ea.k = k;
    ea.P = P;
    ea.n_exp = n_exp;
    ea.start = a.start;
    memset(ea.pam, 0, sizeof(ea.pam));
    memcpy(ea.pam, pam, P);
    hipLaunchKernelGGL(k_km_emit, dim3((unsigned)(((size_t)total + KM_BLOCK - 1) / KM_BLOCK)), dim3(KM_BLOCK), 0,
                       st, ea);
  }
  KM_HIP(hipStreamSynchronize(st));
  KM_HIP(hipGetLastError());
#undef KM_HIP
  cleanup();
  *out = km;
  return GS_OK;
}

extern "C" gs_status gs_kmers_get(gs_kmers *km, int on_device, uint64_t *n, const void **seqs,
                                  const void **pams, const void **positions, const void **senses) {
  if (!km) return GS_ERR_ARG;
  if (n) *n = km->n;
  if (on_device) {
    if (seqs) *seqs = km->d_seqs;
    if (pams) *pams = km->d_pams;
    if (positions) *positions = km->d_pos;
    if (senses) *senses = km->d_sense;
    return GS_OK;
  }
  if (!km->have_host) {
    GS_HIP(hipSetDevice(km->device));
    km->h_seqs.resize((size_t)km->n * km->k);
    km->h_pams.resize((size_t)km->n * km->P);
    km->h_pos.resize((size_t)km->n);
    km->h_sense.resize((size_t)km->n);
    if (km->n) {
      GS_HIP(hipMemcpy(km->h_seqs.data(), km->d_seqs, km->h_seqs.size(), hipMemcpyDeviceToHost));
      GS_HIP(hipMemcpy(km->h_pams.data(), km->d_pams, km->h_pams.size(), hipMemcpyDeviceToHost));
      GS_HIP(hipMemcpy(km->h_pos.data(), km->d_pos, 4 * km->h_pos.size(), hipMemcpyDeviceToHost));
      GS_HIP(hipMemcpy(km->h_sense.data(), km->d_sense, km->h_sense.size(), hipMemcpyDeviceToHost));
    }
    km->have_host = true;
  }
  if (seqs) *seqs = km->h_seqs.data();
  if (pams) *pams = km->h_pams.data();
  if (positions) *positions = km->h_pos.data();
  if (senses) *senses = km->h_sense.data();
  return GS_OK;
}

extern "C" void gs_kmers_free(gs_kmers *km) { km_release(km); }
