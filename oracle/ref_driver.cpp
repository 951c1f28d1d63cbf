/*
 * ref_driver.cpp -- thin extern "C" shim over the REFERENCE'S OWN SOURCES,
 * compiled in place from /root/reference by oracle/Makefile into oracle/_ref/.
 * TEST INFRASTRUCTURE ONLY; never linked into the product.
 *
 * What of the reference can be compiled here without its build system:
 *   sdsl::wt_huff<> (= wt_pc + rank_support_v + _byte_tree): rank, inverse_select,
 *       serialize/load            -- the whole of csa_wt::rank_bwt (csa_wt.hpp:270-273)
 *   sdsl::byte_alphabet           -- C / char2comp (csa_alphabet_strategy.cpp:25-55)
 *   sdsl::int_vector<0> I/O       -- the container format of sa/isa samples
 *   genomics::resolve_absolute    -- src/genomics/structures.cxx
 *   genomics::(reverse_)complement-- src/genomics/sequences.cxx
 *   genomics::mm_scores/pam_scores-- include/genomics/doench.hpp
 * This library stays free of any build trick: it includes only headers that do
 * not reach construct_sa.hpp -> divsufsort.h (cmake-generated).  The CSA itself,
 * genomics/index.hpp, process.hpp and printer.hpp are compiled by the second
 * target, ref_enumerate.cpp (include guards of the construction headers
 * pre-defined; see there).
 *
 * The glue that csa_wt adds on top of these parts (operator[] = LF walk to the
 * next 1-in-64 SA sample, csa_wt.hpp:333-346; serialize order csa_wt.hpp:372-382)
 * is re-stated below around the real wt_huff / int_vector objects so tests can
 * (a) compare locate() and (b) produce an index file in the reference's on-disk
 * layout for the product's SDSL importer.
 */
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include <sdsl/wt_huff.hpp>
#include <sdsl/csa_alphabet_strategy.hpp>
#include <sdsl/int_vector.hpp>
#include <sdsl/int_vector_buffer.hpp>
#include <sdsl/io.hpp>

#include "genomics/structures.hpp"
#include "genomics/sequences.hpp"
#include "genomics/doench.hpp"

namespace {
struct ref_index {
  sdsl::wt_huff<> wt;
  sdsl::byte_alphabet alphabet;
  sdsl::int_vector<0> sa_samples;
  sdsl::int_vector<0> isa_samples;
  uint64_t n;
};
}  // namespace

extern "C" {

/* bwt: n bytes; sa: n entries.  tmp: a writable scratch file path. */
void *ref_index_build(const uint8_t *bwt, const uint32_t *sa, uint64_t n, const char *tmp) {
  ref_index *r = new ref_index();
  r->n = n;
  {
    sdsl::int_vector<8> v(n);
    for (uint64_t i = 0; i < n; i++) v[i] = bwt[i];
    sdsl::store_to_file(v, tmp);
  }
  {
    /* same two constructions as csa_wt(cache_config&) csa_wt.hpp:300-316 */
    sdsl::int_vector_buffer<8> buf(tmp);
    sdsl::byte_alphabet a(buf, n);
    r->alphabet.swap(a);
  }
  {
    sdsl::int_vector_buffer<8> buf(tmp);
    sdsl::wt_huff<> w(buf, n);
    r->wt.swap(w);
  }
  std::remove(tmp);
  /* _sa_order_sampling csa_sampling_strategy.hpp:85-99: width hi(n)+1, every 64th row */
  r->sa_samples.width(sdsl::bits::hi(n) + 1);
  r->sa_samples.resize((n + 63) / 64);
  for (uint64_t i = 0, j = 0; i < n; i += 64) r->sa_samples[j++] = sa[i];
  /* _isa_sampling csa_sampling_strategy.hpp:626-642: ISA[0], ISA[8192], ... */
  r->isa_samples.width(sdsl::bits::hi(n) + 1);
  if (n >= 1) {
    r->isa_samples.resize((n - 1) / 8192 + 1);
    for (uint64_t i = 0; i < n; i++)
      if (sa[i] % 8192 == 0) r->isa_samples[sa[i] / 8192] = i;
  }
  return r;
}
/* same, from the text (n-1 bytes, the sentinel is implied): BWT[i] = T[SA[i]-1] */
void *ref_index_build_text(const uint8_t *text, const uint32_t *sa, uint64_t n, const char *tmp) {
  std::vector<uint8_t> bwt(n);
  for (uint64_t i = 0; i < n; i++) bwt[i] = sa[i] ? text[sa[i] - 1] : 0;
  return ref_index_build(bwt.data(), sa, n, tmp);
}
void ref_index_free(void *p) { delete (ref_index *)p; }

uint64_t ref_rank_bwt(void *p, uint64_t i, uint8_t c) { return ((ref_index *)p)->wt.rank(i, c); }
uint64_t ref_C(void *p, uint8_t c) {
  ref_index *r = (ref_index *)p;
  return r->alphabet.C[r->alphabet.char2comp[c]];
}
uint32_t ref_sigma(void *p) { return ((ref_index *)p)->alphabet.sigma; }
uint8_t ref_char2comp(void *p, uint8_t c) { return ((ref_index *)p)->alphabet.char2comp[c]; }
void ref_inverse_select(void *p, uint64_t i, uint64_t *rank, uint8_t *c) {
  auto rc = ((ref_index *)p)->wt.inverse_select(i);
  *rank = rc.first;
  *c = rc.second;
}
/* csa_wt::operator[] csa_wt.hpp:333-346 restated over the real wt / alphabet / samples */
uint64_t ref_locate(void *p, uint64_t i) {
  ref_index *r = (ref_index *)p;
  uint64_t off = 0;
  while (i % 64 != 0) {
    auto rc = r->wt.inverse_select(i); /* suffix_array_helper.hpp:337-349 */
    i = r->alphabet.C[r->alphabet.char2comp[rc.second]] + rc.first;
    ++off;
  }
  uint64_t result = r->sa_samples[i / 64];
  if (result + off < r->n) return result + off;
  return result + off - r->n;
}
/* csa_wt::serialize order csa_wt.hpp:372-382 using each part's own serialize() */
int ref_write_index_file(void *p, const char *path) {
  ref_index *r = (ref_index *)p;
  std::ofstream out(path, std::ios::binary | std::ios::trunc);
  if (!out) return 1;
  r->wt.serialize(out);
  r->sa_samples.serialize(out);
  r->isa_samples.serialize(out);
  r->alphabet.serialize(out);
  return out.good() ? 0 : 1;
}

/* genomics::resolve_absolute structures.cxx:7-52 */
int ref_resolve_absolute(const uint64_t *chr_len, int n_chr, int64_t abs, int seq_len, int pam_len,
                         int64_t *start, char *strand) {
  genomics::genome_structure gs;
  for (int i = 0; i < n_chr; i++) gs.push_back({std::to_string(i + 1), chr_len[i]});
  genomics::kmer k;
  k.sequence = std::string(seq_len, 'A');
  k.pam = std::string(pam_len, 'G');
  genomics::coordinates c;
  std::string st;
  std::tie(c, st) = genomics::resolve_absolute(gs, abs, k);
  if (c.chr.name == "") return -1;
  *start = (int64_t)c.offset;
  *strand = st[0];
  return std::stoi(c.chr.name) - 1;
}
void ref_reverse_complement(const char *in, char *out) {
  std::string s = genomics::reverse_complement(std::string(in));
  std::strcpy(out, s.c_str());
}
void ref_complement(const char *in, char *out) {
  std::string s = genomics::complement(std::string(in));
  std::strcpy(out, s.c_str());
}
/* doench.hpp tables; returns -1 when the key is absent */
double ref_mm_score(char r, char d, int pos1) {
  std::string key = std::string("r") + r + ":d" + d + "," + std::to_string(pos1);
  auto it = genomics::mm_scores.find(key);
  return it == genomics::mm_scores.end() ? -1.0 : it->second;
}
double ref_pam_score(char a, char b) {
  std::string key;
  key += a;
  key += b;
  auto it = genomics::pam_scores.find(key);
  return it == genomics::pam_scores.end() ? -1.0 : it->second;
}
}
