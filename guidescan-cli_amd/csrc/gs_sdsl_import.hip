/*
 * gs_sdsl_import.hip -- importer for the reference's on-disk index:
 * <prefix>.forward = sdsl::csa_wt<wt_huff<>,64,8192>::serialize
 * (sdsl/include/sdsl/csa_wt.hpp:372-382).  Host code only.
 *
 * The file holds the BWT as a Huffman-shaped wavelet tree plus 1-in-64 SA samples.  The
 * importer rebuilds the BWT by expanding the tree, inverts it to the genome text with
 * independent LF walks between consecutive SA samples (threads), and hands the text to the
 * normal GPU builder - the device layout is derived from the text, never from SDSL's layout.
 */
#include "gs_common.h"

#include <rocprim/rocprim.hpp>

#include <new>
#include <stdexcept>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

struct reader {
  const uint8_t *p, *end;
  bool ok = true;
  template <class T>
  T get() {
    T v{};
    if ((size_t)(end - p) < sizeof(T)) {
      ok = false;
      return v;
    }
    memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  const uint8_t *take(size_t n) {
    if ((size_t)(end - p) < n) {
      ok = false;
      return nullptr;
    }
    const uint8_t *q = p;
    p += n;
    return q;
  }
};

/* int_vector<w>: u64 size_in_bits [, u8 width if w == 0], then ((bits+63)>>6) u64 words
 * (sdsl/include/sdsl/int_vector.hpp:416-419, 593-610, 1563-1595) */
struct ivec {
  uint64_t bits = 0;
  uint8_t width = 0;
  const uint64_t *words = nullptr;
  uint64_t size() const { return width ? bits / width : 0; }
  uint64_t at(uint64_t i) const { /* little-endian bit packing */
    const uint64_t b = i * width;
    const uint64_t w = b >> 6, o = b & 63;
    uint64_t v = words[w] >> o;
    if (o + width > 64) v |= words[w + 1] << (64 - o);
    return width == 64 ? v : v & ((1ull << width) - 1);
  }
};
bool read_ivec(reader &r, uint8_t fixed_width, ivec &v) {
  v.bits = r.get<uint64_t>();
  v.width = fixed_width ? fixed_width : r.get<uint8_t>();
  const uint64_t nwords = (v.bits + 63) >> 6;
  v.words = (const uint64_t *)r.take(nwords * 8);
  return r.ok;
}
/* select_support_mcl::load (select_support_mcl.hpp:464-493); contents unused by the path */
bool skip_select(reader &r) {
  const uint64_t arg_cnt = r.get<uint64_t>();
  if (!r.ok) return false;
  if (!arg_cnt) return true;
  const uint64_t sb = (arg_cnt + 4095) >> 12;
  ivec tmp, mini_or_long;
  if (!read_ivec(r, 0, tmp)) return false;          /* superblock */
  if (!read_ivec(r, 1, mini_or_long)) return false; /* helper bit vector */
  for (uint64_t i = 0; i < sb; i++)
    if (!read_ivec(r, 0, tmp)) return false; /* one int_vector<0> per superblock either way */
  return true;
}

struct wt_node { /* wt_helper.hpp:109-127 */
  uint64_t bv_pos, bv_pos_rank;
  uint16_t parent, child[2];
};

struct parsed {
  uint64_t n = 0, sigma = 0;
  ivec bv, sa_samples;
  std::vector<wt_node> nodes;
  std::vector<uint64_t> ones_before; /* per 64-bit word of bv */
};

bool bit(const ivec &bv, uint64_t i) { return (bv.words[i >> 6] >> (i & 63)) & 1; }
uint64_t rank1(const parsed &P, uint64_t i) { /* ones in bv[0,i) */
  uint64_t r = P.ones_before[i >> 6];
  if (i & 63) r += __builtin_popcountll(P.bv.words[i >> 6] & ((1ull << (i & 63)) - 1));
  return r;
}

/* the symbol sequence stored in node v (size symbols).  The file is user input: every node and
 * every bit range is checked before it is followed (validate_tree), so a foreign or damaged file
 * ends in GS_ERR_FORMAT, not in an out-of-bounds read or an endless recursion. */
bool validate_tree(const parsed &P) {
  const size_t nn = P.nodes.size();
  for (size_t v = 0; v < nn; v++) {
    const wt_node &nd = P.nodes[v];
    if (nd.child[0] == 0xFFFF || nd.child[1] == 0xFFFF) {
      if (nd.child[0] != nd.child[1]) return false; /* a leaf has no child at all */
      continue;
    }
    /* children come after their parent in the node array (BFS order): no cycles, depth <= nodes */
    if (nd.child[0] >= nn || nd.child[1] >= nn || nd.child[0] <= v || nd.child[1] <= v) return false;
    if (nd.bv_pos > P.bv.bits) return false;
  }
  return true;
}
bool expand(const parsed &P, uint16_t v, uint64_t size, std::vector<uint8_t> &out, uint32_t depth = 0) {
  if (v >= P.nodes.size() || depth > 256) return false;
  const wt_node &nd = P.nodes[v];
  out.resize(size);
  if (nd.child[0] == 0xFFFF) { /* leaf: bv_pos_rank holds the symbol */
    std::fill(out.begin(), out.end(), (uint8_t)nd.bv_pos_rank);
    return true;
  }
  if (nd.bv_pos > P.bv.bits || size > P.bv.bits - nd.bv_pos) return false; /* the node's bits lie inside bv */
  const uint64_t ones = rank1(P, nd.bv_pos + size) - rank1(P, nd.bv_pos);
  std::vector<uint8_t> left, right;
  if (!expand(P, nd.child[0], size - ones, left, depth + 1) || !expand(P, nd.child[1], ones, right, depth + 1))
    return false;
  uint64_t l = 0, r = 0;
  for (uint64_t j = 0; j < size; j++) out[j] = bit(P.bv, nd.bv_pos + j) ? right[r++] : left[l++];
  return true;
}

}  // namespace

/* the file read and parsed: P points into buf.  with_ones: the prefix popcounts of the wavelet tree's bits, which the host
 * expansion reads (the device path makes its own) */
static gs_status parse_index_file(const char *path, std::vector<uint8_t> &buf, parsed &P, bool with_ones) {
  FILE *f = fopen(path, "rb");
  if (!f) {
    gs_set_error(std::string("cannot open ") + path);
    return GS_ERR_IO;
  }
  fseek(f, 0, SEEK_END);
  const long fsz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (fsz < 0) {
    fclose(f);
    gs_set_error(std::string("cannot size ") + path);
    return GS_ERR_IO;
  }
  buf.resize((size_t)fsz);
  if (fread(buf.data(), 1, buf.size(), f) != buf.size()) {
    fclose(f);
    gs_set_error("short read");
    return GS_ERR_IO;
  }
  fclose(f);
  reader r{buf.data(), buf.data() + buf.size()};
  /* wt_pc::serialize wt_pc.hpp:656-671 */
  P.n = r.get<uint64_t>();
  P.sigma = r.get<uint64_t>();
  ivec rank_blocks;
  if (!read_ivec(r, 1, P.bv) || !read_ivec(r, 64, rank_blocks) || !skip_select(r) || !skip_select(r)) {
    gs_set_error("truncated wavelet tree");
    return GS_ERR_FORMAT;
  }
  const uint64_t n_nodes = r.get<uint64_t>();
  if (!r.ok || n_nodes == 0 || n_nodes > 65535) {
    gs_set_error("bad wavelet tree node count");
    return GS_ERR_FORMAT;
  }
  P.nodes.resize(n_nodes);
  for (auto &nd : P.nodes) {
    nd.bv_pos = r.get<uint64_t>();
    nd.bv_pos_rank = r.get<uint64_t>();
    nd.parent = r.get<uint16_t>();
    nd.child[0] = r.get<uint16_t>();
    nd.child[1] = r.get<uint16_t>();
  }
  r.take(256 * 2); /* c_to_leaf */
  r.take(256 * 8); /* path */
  ivec isa_samples, char2comp, comp2char, Cv;
  if (!read_ivec(r, 0, P.sa_samples) || !read_ivec(r, 0, isa_samples) || !read_ivec(r, 8, char2comp) ||
      !read_ivec(r, 8, comp2char) || !read_ivec(r, 64, Cv)) {
    gs_set_error("truncated sample/alphabet section");
    return GS_ERR_FORMAT;
  }
  const uint16_t sigma16 = r.get<uint16_t>();
  if (!r.ok || r.p != r.end || sigma16 != P.sigma) {
    gs_set_error("index file has trailing bytes or an inconsistent alphabet (not a "
                 "csa_wt<wt_huff<>,64,8192> file?)");
    return GS_ERR_FORMAT;
  }
  const uint64_t n = P.n;
  if (n < 2 || P.sa_samples.size() != (n + 63) / 64) {
    gs_set_error("SA sample count does not match density 64");
    return GS_ERR_FORMAT;
  }
  if (P.bv.bits > (uint64_t)buf.size() * 8 || !validate_tree(P)) {
    gs_set_error("inconsistent wavelet tree (not a csa_wt<wt_huff<>,64,8192> file?)");
    return GS_ERR_FORMAT;
  }
  if (with_ones) { /* prefix popcounts of the concatenated node bit vectors */
    const uint64_t nw = (P.bv.bits + 63) >> 6;
    P.ones_before.resize(nw + 1);
    uint64_t acc = 0;
    for (uint64_t w = 0; w < nw; w++) {
      P.ones_before[w] = acc;
      acc += __builtin_popcountll(P.bv.words[w]);
    }
    P.ones_before[nw] = acc;
  }
  return GS_OK;
}

gs_status gs_sdsl_read_text(const char *path, std::vector<uint8_t> &text) {
  std::vector<uint8_t> buf;
  parsed P;
  const gs_status prc = parse_index_file(path, buf, P, true);
  if (prc != GS_OK) return prc;
  const uint64_t n = P.n;
  uint64_t acc = 0;

  std::vector<uint8_t> bwt;
  if (!expand(P, 0, n, bwt)) {
    gs_set_error("inconsistent wavelet tree (not a csa_wt<wt_huff<>,64,8192> file?)");
    return GS_ERR_FORMAT;
  }

  /* LF support: C[] and per-64-row checkpoints of the symbols present */
  uint64_t cnt[256] = {0};
  for (uint64_t i = 0; i < n; i++) cnt[bwt[i]]++;
  uint64_t C[256];
  int dense[256], sigma = 0;
  acc = 0;
  for (int c = 0; c < 256; c++) {
    C[c] = acc;
    acc += cnt[c];
    dense[c] = cnt[c] ? sigma++ : -1;
  }
  const uint64_t nb = n / 64 + 1;
  std::vector<uint32_t> ck(nb * sigma);
  {
    std::vector<uint32_t> run(sigma, 0);
    for (uint64_t i = 0; i < n; i++) {
      if ((i & 63) == 0) memcpy(&ck[(i >> 6) * sigma], run.data(), 4 * sigma);
      run[dense[bwt[i]]]++;
    }
    if ((n & 63) == 0) memcpy(&ck[(n >> 6) * sigma], run.data(), 4 * sigma);
  }
  auto lf = [&](uint64_t row) -> uint64_t {
    const uint8_t c = bwt[row];
    uint64_t rk = ck[(row >> 6) * sigma + dense[c]];
    for (uint64_t j = row & ~63ull; j < row; j++) rk += bwt[j] == c;
    return C[c] + rk;
  };
  /* sampled rows sorted by text position; walk the gap below each sample */
  const uint64_t ns = P.sa_samples.size();
  std::vector<std::pair<uint64_t, uint64_t>> smp(ns); /* (text position, row) */
  for (uint64_t j = 0; j < ns; j++) smp[j] = {P.sa_samples.at(j), j * 64};
  std::sort(smp.begin(), smp.end());
  if (smp.back().first != n - 1) { /* row 0 (the sentinel suffix) is always sampled */
    gs_set_error("SA samples lack the sentinel suffix");
    return GS_ERR_FORMAT;
  }
  text.assign(n - 1, 0);
  unsigned nt = std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (ns < 4096) nt = 1;
  std::vector<std::thread> th;
  std::vector<int> bad(nt, 0);
  for (unsigned t = 0; t < nt; t++) {
    th.emplace_back([&, t]() {
      const uint64_t lo = ns * t / nt, hi = ns * (t + 1) / nt;
      for (uint64_t s = lo; s < hi; s++) {
        const uint64_t p = smp[s].first;
        const uint64_t stop = s ? smp[s - 1].first : 0; /* write text[stop .. p-1] */
        uint64_t row = smp[s].second;
        for (uint64_t q = p; q > stop; q--) {
          const uint8_t c = bwt[row];
          if (c == 0) {
            bad[t] = 1;
            break;
          }
          text[q - 1] = c;
          row = lf(row);
        }
      }
    });
  }
  for (auto &x : th) x.join();
  for (int b : bad)
    if (b) {
      gs_set_error("BWT inversion met the sentinel inside the text");
      return GS_ERR_FORMAT;
    }
  return GS_OK;
}

extern "C" gs_status gs_sdsl_extract_text(const char *index_file, uint8_t **text, uint64_t *len) {
  if (!index_file || !text || !len) return GS_ERR_ARG;
  try { /* the parser works in std containers sized by the file's own fields */
    std::vector<uint8_t> t;
    gs_status rc = gs_sdsl_read_text(index_file, t);
    if (rc != GS_OK) return rc;
    uint8_t *p = (uint8_t *)malloc(t.size() ? t.size() : 1);
    if (!p) return GS_ERR_NOMEM;
    memcpy(p, t.data(), t.size());
    *text = p;
    *len = t.size();
    return GS_OK;
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  } catch (const std::length_error &) {
    gs_set_error("index file declares an impossible size");
    return GS_ERR_FORMAT;
  }
}

/* ---- the same on the device: text AND suffix array of one strand from its index file ---------------------------------
 * The host path above expands the wavelet tree level by level in std::vectors, builds checkpoints and walks LF between
 * the samples on the host's threads, and the index is then built from the text alone - the suffix sort again: 118 s at
 * hg38 size.  Here the file's own content is used whole: BWT[i] for every row by a wavelet-tree access (two or three
 * ranks over the node bit vectors, csa_wt.hpp:270-273 -> wt_pc.hpp:360-384), per-64-row symbol counts, the samples
 * (SA[0], SA[64], ... packed at their width, csa_sampling_strategy.hpp:85-99) ordered by text position, and one thread
 * per sample walking LF down to the sample before it: every step knows its row AND its text position, so it writes
 * text[q - 1] = BWT[row] and SA[row] = q.  No sort: the reference's suffix arrays are taken from the reference's files. */
#define IMP_MAXNODE 64u
#define IMP_MAXSIGMA 16u
struct imp_tree {
  uint64_t bv_pos[IMP_MAXNODE], rank_at[IMP_MAXNODE]; /* first bit of the node, ones before it */
  uint16_t child0[IMP_MAXNODE], child1[IMP_MAXNODE];
  uint8_t sym[IMP_MAXNODE]; /* a leaf's symbol */
};
__device__ __forceinline__ uint64_t imp_rank1(const uint64_t *words, const uint64_t *ones, uint64_t i) {
  uint64_t r = ones[i >> 6];
  if (i & 63) r += (uint64_t)__popcll(words[i >> 6] & ((1ull << (i & 63)) - 1ull));
  return r;
}
__global__ __launch_bounds__(256) void k_imp_access(const uint64_t *words, const uint64_t *ones, imp_tree T, uint64_t n, uint8_t *bwt) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t v = 0;
    uint64_t pos = i;
    for (uint32_t d = 0; d < IMP_MAXNODE && T.child0[v] != 0xFFFFu; ++d) {
      const uint64_t at = T.bv_pos[v] + pos;
      const bool b = (words[at >> 6] >> (at & 63)) & 1ull;
      const uint64_t r1 = imp_rank1(words, ones, at) - T.rank_at[v];
      pos = b ? r1 : pos - r1;
      v = b ? T.child1[v] : T.child0[v];
    }
    bwt[i] = T.sym[v];
  }
}
__global__ __launch_bounds__(256) void k_imp_hist(const uint8_t *bwt, uint64_t n, unsigned long long *hist) {
  __shared__ uint32_t s_h[256];
  s_h[threadIdx.x] = 0u;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&s_h[bwt[i]], 1u);
  __syncthreads();
  if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}
struct imp_dense {
  uint8_t of[256]; /* symbol -> its place among the symbols present (0xFF: absent) */
};
/* cnt[d * nb + c] = rows of symbol d in chunk c (64 rows) */
__global__ __launch_bounds__(256) void k_imp_chunks(const uint8_t *bwt, uint64_t n, imp_dense D, uint32_t sigma, uint64_t nb, uint32_t *cnt) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nb) return;
  uint32_t k[IMP_MAXSIGMA];
#pragma unroll
  for (uint32_t d = 0; d < IMP_MAXSIGMA; ++d) k[d] = 0u;
  const uint64_t lo = c * 64u, hi = lo + 64u < n ? lo + 64u : n;
  for (uint64_t i = lo; i < hi; ++i) {
    const uint32_t d = D.of[bwt[i]];
#pragma unroll
    for (uint32_t e = 0; e < IMP_MAXSIGMA; ++e) k[e] += d == e ? 1u : 0u;
  }
#pragma unroll
  for (uint32_t d = 0; d < IMP_MAXSIGMA; ++d)
    if (d < sigma) cnt[(uint64_t)d * nb + c] = k[d];
}
__global__ __launch_bounds__(256) void k_imp_samples(const uint64_t *packed, uint32_t width, uint64_t ns, uint32_t *pos, uint32_t *idx) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  const uint64_t b = j * width, w = b >> 6, o = b & 63;
  uint64_t v = packed[w] >> o;
  if (o + width > 64) v |= packed[w + 1] << (64 - o);
  if (width < 64) v &= (1ull << width) - 1ull;
  pos[j] = (uint32_t)v;
  idx[j] = (uint32_t)j;
}
struct imp_lf {
  const uint8_t *bwt;
  const uint32_t *ck; /* ck[d * nb + c] = rows of symbol d before chunk c */
  uint64_t nb;
  uint32_t C[IMP_MAXSIGMA]; /* first row of each present symbol */
  imp_dense D;
};
__global__ __launch_bounds__(256) void k_imp_invert(imp_lf L, const uint32_t *pos, const uint32_t *idx, uint64_t ns, uint8_t *text, uint32_t *sa,
                                                    uint32_t *bad) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  const uint32_t p = pos[s], stop = s ? pos[s - 1] : 0u;
  uint64_t row = 64ull * idx[s];
  sa[row] = p;
  if (s && p == stop) atomicOr(bad, 2u); /* two samples at one text position */
  for (uint32_t q = p; q > stop; --q) {
    const uint32_t c = L.bwt[row];
    if (c == 0u) { /* the sentinel inside the text */
      atomicOr(bad, 1u);
      break;
    }
    text[q - 1u] = (uint8_t)c;
    /* LF: rows of c before this one (csa_wt's rank_bwt), from the chunk's count and the chunk's bytes below the row */
    const uint32_t d = L.D.of[c];
    uint32_t rk = L.ck[(uint64_t)d * L.nb + (row >> 6)];
    const uint64_t base = row & ~63ull;
    const uint32_t upto = (uint32_t)(row & 63u);
    const uint32_t *w32 = (const uint32_t *)(L.bwt + base); /* (chunks are 64-byte aligned: the array is) */
    const uint32_t cc = c * 0x01010101u;
    for (uint32_t j = 0; 4u * j < upto; ++j) {
      uint32_t x = w32[j] ^ cc;
      if (4u * j + 4u > upto) x |= 0xFFFFFFFFu << (8u * (upto - 4u * j)); /* bytes at and beyond the row: never equal */
      const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); /* 0x80 per zero byte */
      rk += (uint32_t)__popc(t);
    }
    row = (uint64_t)L.C[d] + rk;
    /* (at q - 1 = stop the row is the sample's below, whose own thread writes it - but nobody samples below the first:
     * its walk ends at the row of the suffix at text position 0, whose BWT symbol is the sentinel) */
    if (q - 1u > stop || s == 0) sa[row] = q - 1u;
  }
}
__global__ __launch_bounds__(256) void k_imp_cmp_revcomp(const uint8_t *fwd, const uint8_t *rev, uint64_t len, unsigned long long *diff) {
  unsigned long long v = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t c = fwd[len - 1u - i];
    const uint8_t rc = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'a' ? 't' : c == 't' ? 'a' : c == 'c' ? 'g' : c == 'g' ? 'c' : c;
    v += rev[i] != rc;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63u) == 0u && v) atomicAdd(diff, v);
}

namespace {
struct dbuf { /* a device allocation that frees itself */
  void *p = nullptr;
  ~dbuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t get(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  void *release() {
    void *q = p;
    p = nullptr;
    return q;
  }
};
}  // namespace
#define IMP_HIP(expr)                                                         \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess) {                                                  \
      gs_set_error(std::string(#expr) + ": " + hipGetErrorString(e__));       \
      return e__ == hipErrorOutOfMemory ? GS_ERR_NOMEM : GS_ERR_DEVICE;       \
    }                                                                         \
  } while (0)

/* one strand's file -> its text (n bytes: the genome and the sentinel) and suffix array (n rows) in device memory.
 * GS_ERR_UNSUPPORTED: a file this path does not take (more than 16 distinct symbols, 2^32 rows) - the caller uses the host path */
static gs_status sdsl_strand_on_device(const char *path, uint8_t **d_text_out, uint32_t **d_sa_out, uint64_t *n_out) {
  std::vector<uint8_t> buf;
  parsed P;
  gs_status rc = parse_index_file(path, buf, P, true);
  if (rc != GS_OK) return rc;
  const uint64_t n = P.n;
  if (n >= (1ull << 32) - 256 || P.nodes.size() > IMP_MAXNODE) return GS_ERR_UNSUPPORTED;
  /* the tree: every node's size (the root holds all n symbols) and where its bits lie, checked against the bit vector */
  imp_tree T;
  memset(&T, 0, sizeof(T));
  std::vector<uint64_t> size(P.nodes.size(), 0);
  size[0] = n;
  for (size_t v = 0; v < P.nodes.size(); v++) { /* (children come after their parent: validate_tree) */
    const wt_node &nd = P.nodes[v];
    T.child0[v] = nd.child[0];
    T.child1[v] = nd.child[1];
    if (nd.child[0] == 0xFFFF) {
      T.sym[v] = (uint8_t)nd.bv_pos_rank;
      continue;
    }
    if (nd.bv_pos > P.bv.bits || size[v] > P.bv.bits - nd.bv_pos) {
      gs_set_error("inconsistent wavelet tree (a node's bits lie outside the bit vector)");
      return GS_ERR_FORMAT;
    }
    T.bv_pos[v] = nd.bv_pos;
    T.rank_at[v] = rank1(P, nd.bv_pos);
    const uint64_t ones = rank1(P, nd.bv_pos + size[v]) - T.rank_at[v];
    size[nd.child[0]] += size[v] - ones;
    size[nd.child[1]] += ones;
  }
  const uint64_t nw = (P.bv.bits + 63) >> 6, ns = P.sa_samples.size(), nb = n / 64 + 1;
  const uint32_t width = P.sa_samples.width;
  if (width == 0 || width > 64) {
    gs_set_error("bad SA sample width");
    return GS_ERR_FORMAT;
  }
  dbuf d_words, d_ones, d_bwt, d_hist;
  IMP_HIP(d_words.get(8 * (nw + 2)));
  IMP_HIP(d_ones.get(8 * (nw + 2)));
  IMP_HIP(d_bwt.get(nb * 64 + 64)); /* whole chunks, zero beyond the last row */
  IMP_HIP(d_hist.get(256 * 8));
  IMP_HIP(hipMemset(d_words.p, 0, 8 * (nw + 2)));
  IMP_HIP(hipMemcpy(d_words.p, P.bv.words, 8 * nw, hipMemcpyHostToDevice));
  IMP_HIP(hipMemcpy(d_ones.p, P.ones_before.data(), 8 * (nw + 1), hipMemcpyHostToDevice));
  IMP_HIP(hipMemset(d_bwt.p, 0, nb * 64 + 64));
  IMP_HIP(hipMemset(d_hist.p, 0, 256 * 8));
  const unsigned grid = 256u * 32u;
  hipLaunchKernelGGL(k_imp_access, dim3(grid), dim3(256), 0, 0, (const uint64_t *)d_words.p, (const uint64_t *)d_ones.p, T, n, (uint8_t *)d_bwt.p);
  hipLaunchKernelGGL(k_imp_hist, dim3(grid), dim3(256), 0, 0, (const uint8_t *)d_bwt.p, n, (unsigned long long *)d_hist.p);
  unsigned long long hist[256];
  IMP_HIP(hipMemcpy(hist, d_hist.p, sizeof(hist), hipMemcpyDeviceToHost));
  (void)hipFree(d_words.release());
  (void)hipFree(d_ones.release());
  imp_lf L;
  memset(&L, 0, sizeof(L));
  memset(L.D.of, 0xFF, sizeof(L.D.of));
  uint32_t sigma = 0;
  uint64_t acc = 0;
  for (int c = 0; c < 256; c++) {
    if (!hist[c]) continue;
    if (sigma == IMP_MAXSIGMA) return GS_ERR_UNSUPPORTED;
    L.D.of[c] = (uint8_t)sigma;
    L.C[sigma++] = (uint32_t)acc;
    acc += hist[c];
  }
  if (acc != n || hist[0] != 1) {
    gs_set_error("the wavelet tree does not hold one sentinel and n symbols");
    return GS_ERR_FORMAT;
  }
  /* rows of each symbol before every chunk of 64 */
  dbuf d_cnt, d_ck, d_tmp;
  IMP_HIP(d_cnt.get(4 * (size_t)sigma * nb));
  IMP_HIP(d_ck.get(4 * (size_t)sigma * nb));
  hipLaunchKernelGGL(k_imp_chunks, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, (const uint8_t *)d_bwt.p, n, L.D, sigma, nb, (uint32_t *)d_cnt.p);
  size_t tb = 0;
  IMP_HIP(rocprim::exclusive_scan(nullptr, tb, (uint32_t *)d_cnt.p, (uint32_t *)d_ck.p, 0u, (size_t)nb, rocprim::plus<uint32_t>(), 0));
  IMP_HIP(d_tmp.get(tb + 16));
  for (uint32_t d = 0; d < sigma; d++) {
    size_t t2 = tb;
    IMP_HIP(rocprim::exclusive_scan(d_tmp.p, t2, (uint32_t *)d_cnt.p + (size_t)d * nb, (uint32_t *)d_ck.p + (size_t)d * nb, 0u, (size_t)nb,
                                    rocprim::plus<uint32_t>(), 0));
  }
  (void)hipFree(d_cnt.release());
  /* the samples by text position */
  dbuf d_packed, d_pos, d_idx, d_pos2, d_idx2, d_sorttmp;
  const uint64_t pw = (P.sa_samples.bits + 63) >> 6;
  IMP_HIP(d_packed.get(8 * (pw + 2)));
  IMP_HIP(hipMemset(d_packed.p, 0, 8 * (pw + 2)));
  IMP_HIP(hipMemcpy(d_packed.p, P.sa_samples.words, 8 * pw, hipMemcpyHostToDevice));
  IMP_HIP(d_pos.get(4 * ns));
  IMP_HIP(d_idx.get(4 * ns));
  IMP_HIP(d_pos2.get(4 * ns));
  IMP_HIP(d_idx2.get(4 * ns));
  hipLaunchKernelGGL(k_imp_samples, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, 0, (const uint64_t *)d_packed.p, width, ns, (uint32_t *)d_pos.p,
                     (uint32_t *)d_idx.p);
  size_t sb = 0;
  IMP_HIP(rocprim::radix_sort_pairs(nullptr, sb, (uint32_t *)d_pos.p, (uint32_t *)d_pos2.p, (uint32_t *)d_idx.p, (uint32_t *)d_idx2.p, (size_t)ns, 0, 32, 0));
  IMP_HIP(d_sorttmp.get(sb + 16));
  IMP_HIP(rocprim::radix_sort_pairs(d_sorttmp.p, sb, (uint32_t *)d_pos.p, (uint32_t *)d_pos2.p, (uint32_t *)d_idx.p, (uint32_t *)d_idx2.p, (size_t)ns, 0, 32, 0));
  uint32_t last = 0;
  IMP_HIP(hipMemcpy(&last, (uint32_t *)d_pos2.p + (ns - 1), 4, hipMemcpyDeviceToHost));
  if ((uint64_t)last != n - 1) { /* row 0 (the sentinel suffix) is always sampled */
    gs_set_error("SA samples lack the sentinel suffix");
    return GS_ERR_FORMAT;
  }
  /* the walks */
  dbuf d_text, d_sa, d_bad;
  IMP_HIP(d_text.get(n + 64));
  IMP_HIP(d_sa.get(4 * (n + 64)));
  IMP_HIP(d_bad.get(16));
  IMP_HIP(hipMemset(d_text.p, 0, n + 64));
  IMP_HIP(hipMemset(d_sa.p, 0xFF, 4 * (n + 64)));
  IMP_HIP(hipMemset(d_bad.p, 0, 16));
  L.bwt = (const uint8_t *)d_bwt.p;
  L.ck = (const uint32_t *)d_ck.p;
  L.nb = nb;
  hipLaunchKernelGGL(k_imp_invert, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, 0, L, (const uint32_t *)d_pos2.p, (const uint32_t *)d_idx2.p, ns,
                     (uint8_t *)d_text.p, (uint32_t *)d_sa.p, (uint32_t *)d_bad.p);
  uint32_t bad = 0;
  IMP_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
  IMP_HIP(hipGetLastError());
  if (bad) {
    gs_set_error(bad & 1u ? "BWT inversion met the sentinel inside the text" : "two SA samples at one text position");
    return GS_ERR_FORMAT;
  }
  *d_text_out = (uint8_t *)d_text.release();
  *d_sa_out = (uint32_t *)d_sa.release();
  *n_out = n;
  return GS_OK;
}

gs_status gs_build_from_device_sa(const uint8_t *text, uint64_t len, const uint32_t *d_sa_fwd, const uint32_t *d_sa_rev, int device, gs_index **out);

/* both strands' files -> the index, without a suffix sort; GS_ERR_UNSUPPORTED / GS_ERR_IO: take the host path */
static gs_status open_sdsl_on_device(const char *prefix, int device, gs_index **out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    gs_set_error("no such HIP device");
    return GS_ERR_DEVICE;
  }
  IMP_HIP(hipSetDevice(device));
  struct strand_t {
    uint8_t *text = nullptr;
    uint32_t *sa = nullptr;
    uint64_t n = 0;
    ~strand_t() {
      if (text) (void)hipFree(text);
      if (sa) (void)hipFree(sa);
    }
  } f, r;
  gs_status rc = sdsl_strand_on_device((std::string(prefix) + ".forward").c_str(), &f.text, &f.sa, &f.n);
  if (rc != GS_OK) return rc;
  rc = sdsl_strand_on_device((std::string(prefix) + ".reverse").c_str(), &r.text, &r.sa, &r.n);
  if (rc != GS_OK) return rc == GS_ERR_IO ? GS_ERR_UNSUPPORTED : rc; /* (no .reverse file: the host path rebuilds that strand from the text) */
  if (r.n != f.n) {
    gs_set_error(".forward and .reverse hold texts of different lengths");
    return GS_ERR_FORMAT;
  }
  /* the reverse index is the FM-index of reverse_complement(forward text) (src/guidescan.cxx:146-157): it must be */
  const uint64_t len = f.n - 1;
  dbuf d_diff;
  IMP_HIP(d_diff.get(16));
  IMP_HIP(hipMemset(d_diff.p, 0, 16));
  hipLaunchKernelGGL(k_imp_cmp_revcomp, dim3(256u * 16u), dim3(256), 0, 0, (const uint8_t *)f.text, (const uint8_t *)r.text, len, (unsigned long long *)d_diff.p);
  unsigned long long diff = 0;
  IMP_HIP(hipMemcpy(&diff, d_diff.p, 8, hipMemcpyDeviceToHost));
  if (diff) {
    gs_set_error(".reverse is not the index of the reverse complement of .forward's text");
    return GS_ERR_FORMAT;
  }
  (void)hipFree(r.text);
  r.text = nullptr;
  std::vector<uint8_t> text(len);
  IMP_HIP(hipMemcpy(text.data(), f.text, len, hipMemcpyDeviceToHost));
  (void)hipFree(f.text);
  f.text = nullptr;
  rc = gs_build_from_device_sa(text.data(), len, f.sa, r.sa, device, out);
  if (rc != GS_OK) return rc;
  /* the arrays came from the file's samples, not from a sort of this text: neighbouring rows are compared by their
   * suffixes in the text at 2^16 places per strand (a damaged sample moves a whole walk of 64 rows: a permutation still,
   * in the wrong order) */
  for (int strand = 0; strand < 2; strand++) {
    gs_sa_report rep;
    rc = gs_index_verify_sa(*out, strand, text.data(), len, 65536, 0x5D51ull + (uint64_t)strand, &rep);
    if (rc == GS_OK && (rep.not_permutation || rep.out_of_order || rep.bwt_mismatch)) {
      gs_set_error("the index file's suffix array samples do not order its text");
      rc = GS_ERR_FORMAT;
    }
    if (rc != GS_OK) {
      gs_index_close(*out);
      *out = nullptr;
      return rc;
    }
  }
  return GS_OK;
}

extern "C" gs_status gs_index_open_sdsl(const char *prefix, int device, gs_index **out) {
  if (!prefix || !out) return GS_ERR_ARG;
  try {
    /* text and suffix arrays of both strands straight from the files, on the device (no sort); what that path does not
     * take - no .reverse file, more than 16 distinct symbols - goes through the host path: .forward's text, both strands
     * built from it */
    {
      const gs_status drc = open_sdsl_on_device(prefix, device, out);
      if (drc != GS_ERR_UNSUPPORTED) return drc;
      (void)hipGetLastError();
    }
    std::vector<uint8_t> fwd;
    gs_status rc = gs_sdsl_read_text((std::string(prefix) + ".forward").c_str(), fwd);
    if (rc != GS_OK) return rc;
    /* the reverse index is the FM-index of reverse_complement(forward text)
     * (src/guidescan.cxx:146-157); it is rebuilt from the text rather than imported */
    return gs_index_build(fwd.data(), fwd.size(), device, out);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  } catch (const std::length_error &) {
    gs_set_error("index file declares an impossible size");
    return GS_ERR_FORMAT;
  }
}
