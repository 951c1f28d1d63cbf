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

gs_status gs_sdsl_read_text(const char *path, std::vector<uint8_t> &text) {
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
  std::vector<uint8_t> buf((size_t)fsz);
  if (fread(buf.data(), 1, buf.size(), f) != buf.size()) {
    fclose(f);
    gs_set_error("short read");
    return GS_ERR_IO;
  }
  fclose(f);
  reader r{buf.data(), buf.data() + buf.size()};
  parsed P;
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
  /* prefix popcounts of the concatenated node bit vectors */
  const uint64_t nw = (P.bv.bits + 63) >> 6;
  P.ones_before.resize(nw + 1);
  uint64_t acc = 0;
  for (uint64_t w = 0; w < nw; w++) {
    P.ones_before[w] = acc;
    acc += __builtin_popcountll(P.bv.words[w]);
  }
  P.ones_before[nw] = acc;

  std::vector<uint8_t> bwt;
  if (P.bv.bits > (uint64_t)buf.size() * 8 || !validate_tree(P) || !expand(P, 0, n, bwt)) {
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

extern "C" gs_status gs_index_open_sdsl(const char *prefix, int device, gs_index **out) {
  if (!prefix || !out) return GS_ERR_ARG;
  try {
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
