/*
 * gs_text.hip -- host-side text encoders for the hit lists the device returns: the step right
 * after the hot path.  Byte-exact restatement of the reference's
 * CSV and SAM database formats (include/genomics/printer.hpp:173-360) and of
 * resolve_absolute (src/genomics/structures.cxx:7-52).  Host code only; no kernels.
 */
#include "gs_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifndef GS_GUIDESCAN_VERSION
#define GS_GUIDESCAN_VERSION "2.0.0" /* the database format version this writer is compatible with */
#endif

namespace {

char comp(char c) { /* src/genomics/sequences.cxx:14-26 */
  switch (c) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'a': return 't';
    case 't': return 'a';
    case 'c': return 'g';
    case 'g': return 'c';
    default: return c;
  }
}
std::string complement(const std::string &s) {
  std::string o(s);
  for (auto &c : o) c = comp(c);
  return o;
}
std::string reverse_complement(const std::string &s) {
  std::string o(s.rbegin(), s.rend());
  for (auto &c : o) c = comp(c);
  return o;
}

/* src/genomics/structures.cxx:7-52.  Returns chromosome index or -1 for the sentinel. */
int resolve_absolute(const gs_genome_structure *gs, int64_t abs, size_t seq_len, size_t pam_len,
                     int64_t *start, char *strand) {
  char st = '+';
  if (abs < 0) {
    abs = -abs;
    st = '-';
  }
  int c = -1;
  for (uint32_t i = 0; i < gs->n_chr; i++) {
    if (abs <= (int64_t)(gs->chr_lengths[i] - 1)) {
      c = (int)i;
      break;
    }
    abs -= (int64_t)gs->chr_lengths[i];
  }
  if (c < 0) return -1;
  int64_t s, e;
  if (st == '+') { /* abs = 0-based inclusive end of the site */
    e = abs + 1;
    s = e - (int64_t)seq_len - (int64_t)pam_len + 1;
  } else { /* abs = 0-based start of the site */
    s = abs + 1;
    e = s + (int64_t)seq_len + (int64_t)pam_len - 1;
  }
  if (s < 0 || e > (int64_t)gs->chr_lengths[c]) return -1; /* :46-48, note s<0 not s<1 */
  *start = s;
  *strand = st;
  return c;
}

void hex_le(std::string &out, uint64_t v) { /* printer.hpp:18-79 */
  static const char *hx = "0123456789abcdef";
  for (int i = 0; i < 8; i++) {
    const unsigned b = (unsigned)(v & 0xff);
    v >>= 8;
    out += hx[b >> 4];
    out += hx[b & 15];
  }
}
std::string f2s(float f) { /* std::to_string(float) */
  char t[64];
  snprintf(t, sizeof t, "%f", (double)f);
  return t;
}

struct decoded_hit {
  int64_t pos;
  uint32_t mismatches;
  std::string sequence;       /* match.sequence */
  std::string match_sequence; /* complement(match.sequence), printer.hpp:232,264 */
  std::string pam;            /* printer.hpp:139-143 */
  uint32_t rna_bulges = 0, dna_bulges = 0;
};

}  // namespace

/* spec: the guide's specificity when the caller already has it (gs_score / gs_score_device:
 * the same float, computed on the device); nullptr = compute CFD and the sum here */
static gs_status format_decoded(const gs_genome_structure *gs, const gs_kmer *k,
                                const std::vector<std::vector<decoded_hit>> &off, uint32_t mismatches,
                                uint32_t flags, int64_t max_off_targets, char **out_text, size_t *out_len,
                                const float *spec = nullptr) {
  const bool start = flags & GS_FLAG_PAM_AT_START;
  const bool sam = flags & GS_TEXT_SAM;
  const bool complete = flags & GS_TEXT_COMPLETE;
  const std::string seq(k->sequence), pam(k->pam);
  const uint32_t L = (uint32_t)seq.size(), P = (uint32_t)pam.size();
  const std::string sequence = start ? pam + seq : seq + pam;
  std::string out;
  if (!sam) {
    /* printer.hpp:245-300 */
    float cfd_sum = 0.0f;
    bool perfect = false, none = true;
    std::vector<std::string> lines;
    for (uint32_t d = 0; d <= mismatches; d++) {
      for (int64_t i = 0; i < (int64_t)off[d].size(); i++) {
        none = false;
        if (max_off_targets != -1 && i >= max_off_targets) break; /* raw index, :259 */
        const decoded_hit &h = off[d][(size_t)i];
        if (h.mismatches == 0 && h.pam.size() == 3 && h.pam.compare(1, 2, "GG") == 0) perfect = true;
        int64_t s;
        char st;
        const int c = resolve_absolute(gs, h.pos, L, P, &s, &st);
        if (c < 0) continue; /* boundary sentinel: no row, no CFD (:280-283) */
        std::string line(k->id);
        line += ',';
        line += sequence;
        line += ',';
        line += gs->chr_names[c];
        line += ',';
        line += std::to_string(s);
        line += ',';
        line += st;
        line += ',';
        line += std::to_string(h.mismatches);
        if (complete) {
          line += ',';
          line += h.match_sequence;
          line += ',';
          line += std::to_string(h.rna_bulges); /* printer.hpp:235-239 */
          line += ',';
          line += std::to_string(h.dna_bulges);
        }
        lines.push_back(std::move(line));
        if (!spec) cfd_sum += gs_calculate_cfd(k->sequence, h.match_sequence.c_str(), h.pam.c_str());
      }
    }
    if (none) { /* :189-199 */
      out += k->id;
      out += ',';
      out += sequence;
      out += ",NA,NA,NA,0";
      if (complete) out += ",NA,NA,NA";
      out += ",1.0\n";
    } else {
      float specificity = 0.0f;
      if (!perfect) cfd_sum += 1;
      if (cfd_sum > 0) specificity = 1 / cfd_sum;
      if (spec) specificity = *spec;
      const std::string sp = f2s(specificity);
      for (const auto &l : lines) {
        out += l;
        out += ',';
        out += sp;
        out += '\n';
      }
    }
  } else {
    /* off_target_fields printer.hpp:115-170 */
    int64_t delim = 0;
    for (uint32_t i = 0; i < gs->n_chr; i++) delim += (int64_t)gs->chr_lengths[i];
    delim = -(delim + 1);
    std::string hex;
    float cfd_sum = 0.0f;
    bool perfect = false;
    for (uint32_t d = 0; d <= mismatches; d++) {
      int64_t n_off = 0;
      for (const decoded_hit &h : off[d]) {
        if (max_off_targets != -1 && n_off >= max_off_targets) break; /* kept hits, :129 */
        if (h.mismatches == 0 && h.pam.size() == 3 && h.pam.compare(1, 2, "GG") == 0) perfect = true;
        int64_t s;
        char st;
        if (resolve_absolute(gs, h.pos, L, P, &s, &st) < 0) continue;
        hex_le(hex, (uint64_t)h.pos);
        if (!spec) cfd_sum += gs_calculate_cfd(k->sequence, h.match_sequence.c_str(), h.pam.c_str());
        n_off++;
      }
      hex_le(hex, (uint64_t)d);
      hex_le(hex, (uint64_t)delim);
    }
    float specificity = 0.0f;
    if (!perfect) cfd_sum += 1;
    if (cfd_sum > 0) specificity = 1 / cfd_sum;
    if (spec) specificity = *spec;
    /* one line per distance-0 hit, printer.hpp:314-357 */
    for (const decoded_hit &h : off[0]) {
      int64_t s = 0;
      char st = 0;
      const int c = resolve_absolute(gs, h.pos, L, P, &s, &st);
      out += k->id;
      out += '\t';
      out += k->sense_positive ? "0" : "16";
      out += '\t';
      if (c >= 0) out += gs->chr_names[c]; /* no sentinel check in the reference: empty RNAME */
      out += '\t';
      out += std::to_string(c >= 0 ? s : 0);
      out += "\t100\t";
      out += std::to_string(sequence.size());
      out += "M\t*\t0\t0\t";
      out += k->sense_positive ? sequence : reverse_complement(sequence);
      out += "\t*";
      for (uint32_t d = 0; d <= mismatches; d++) {
        out += "\tk";
        out += std::to_string(d);
        out += ":i:";
        out += std::to_string(off[d].size());
      }
      if (complete) {
        out += "\tof:H:";
        out += hex;
      }
      out += "\tsp:f:";
      out += f2s(specificity);
      out += '\n';
    }
  }
  char *p = (char *)malloc(out.size() + 1);
  if (!p) return GS_ERR_NOMEM;
  memcpy(p, out.data(), out.size());
  p[out.size()] = 0;
  *out_text = p;
  if (out_len) *out_len = out.size();
  return GS_OK;
}

static gs_status format_guide_impl(const gs_genome_structure *gs, const gs_kmer *k,
                                   const gs_hit *hits, uint64_t n_hits, uint32_t mismatches,
                                   uint32_t flags, int64_t max_off_targets, char **out_text,
                                   size_t *out_len, const float *spec) {
  if (!gs || !k || !k->id || !k->sequence || !k->pam || (n_hits && !hits) || !out_text)
    return GS_ERR_ARG;
  const uint32_t L = (uint32_t)strlen(k->sequence), P = (uint32_t)strlen(k->pam);
  /* split by distance; hits arrive in canonical order (distance ascending) */
  std::vector<std::vector<decoded_hit>> off(mismatches + 1);
  std::vector<char> buf(L + P + 1);
  for (uint64_t h = 0; h < n_hits; h++) {
    const uint32_t d = GS_KEY_MISMATCHES(hits[h].key);
    if (d > mismatches) return GS_ERR_ARG;
    gs_status rc = gs_decode_sequence(k->sequence, L, P, flags & GS_FLAG_PAM_AT_START, hits[h].key,
                                      buf.data());
    if (rc != GS_OK) return rc;
    decoded_hit dh;
    dh.pos = hits[h].pos;
    dh.mismatches = d;
    dh.sequence = buf.data();
    dh.match_sequence = complement(dh.sequence);
    dh.pam = dh.match_sequence.size() < 20 ? std::string() : dh.match_sequence.substr(20, 3);
    off[d].push_back(std::move(dh));
  }
  return format_decoded(gs, k, off, mismatches, flags, max_off_targets, out_text, out_len, spec);
}

extern "C" gs_status gs_format_guide(const gs_genome_structure *gs, const gs_kmer *k,
                                     const gs_hit *hits, uint64_t n_hits, uint32_t mismatches,
                                     uint32_t flags, int64_t max_off_targets, char **out_text,
                                     size_t *out_len) {
  return format_guide_impl(gs, k, hits, n_hits, mismatches, flags, max_off_targets, out_text, out_len, nullptr);
}

extern "C" gs_status gs_format_guide_scored(const gs_genome_structure *gs, const gs_kmer *k,
                                            const gs_hit *hits, uint64_t n_hits, uint32_t mismatches,
                                            uint32_t flags, int64_t max_off_targets, float specificity,
                                            char **out_text, size_t *out_len) {
  return format_guide_impl(gs, k, hits, n_hits, mismatches, flags, max_off_targets, out_text, out_len,
                           &specificity);
}

/* ---- the CSV rows of many guides in one buffer: what the writer of `guidescan enumerate` calls.
 * Same bytes as gs_format_guide_scored guide after guide (get_csv_lines, printer.hpp:245-300), but
 * no per-hit strings: the 13 M lines of a 1 M-guide batch are written straight into one growing
 * buffer, the match sequence decoded from the hit key in place. */
namespace {
struct outbuf {
  char *p = nullptr;
  size_t len = 0, cap = 0;
  bool ok = true;
  bool need(size_t k) {
    if (len + k <= cap) return true;
    size_t nc = cap ? cap + cap / 2 : (1u << 16);
    while (nc < len + k) nc += nc / 2;
    char *q = (char *)realloc(p, nc);
    if (!q) {
      ok = false;
      return false;
    }
    p = q;
    cap = nc;
    return true;
  }
  void put(const char *s, size_t k) {
    memcpy(p + len, s, k);
    len += k;
  }
  void ch(char c) { p[len++] = c; }
  void u64(uint64_t v) {
    char t[24];
    int k = 0;
    do {
      t[k++] = (char)('0' + v % 10);
      v /= 10;
    } while (v);
    while (k) p[len++] = t[--k];
  }
};
}  // namespace

static gs_status format_csv_fast(const gs_genome_structure *gs, const uint64_t *chr_end /* prefix sums */,
                                 const gs_kmer *k, const gs_hit *hits, uint64_t n_hits, uint32_t mismatches,
                                 uint32_t flags, int64_t max_off_targets, float spec, outbuf &o) {
  const bool start = flags & GS_FLAG_PAM_AT_START, complete = flags & GS_TEXT_COMPLETE;
  const size_t L = strlen(k->sequence), P = strlen(k->pam), idl = strlen(k->id);
  if (L < 1 || 2 * L + 3 * P > 59) return GS_ERR_ARG;
  if (!o.need(idl + L + P + 64)) return GS_ERR_NOMEM;
  if (n_hits == 0) { /* printer.hpp:189-199 */
    o.put(k->id, idl);
    o.ch(',');
    if (start) o.put(k->pam, P);
    o.put(k->sequence, L);
    if (!start) o.put(k->pam, P);
    o.put(",NA,NA,NA,0", 11);
    if (complete) o.put(",NA,NA,NA", 9);
    o.put(",1.0\n", 5);
    return GS_OK;
  }
  char sp[64];
  const int spl = snprintf(sp, sizeof sp, "%f", (double)spec); /* std::to_string(float) */
  static const char B[4] = {'A', 'C', 'G', 'T'};
  static const char PBC[5] = {'T', 'G', 'C', 'N', 'A'}; /* complement of A,C,G,N,T */
  uint32_t cur_d = 0xFFFFFFFFu;
  int64_t idx_in_d = 0;
  for (uint64_t h = 0; h < n_hits; h++) {
    const uint64_t key = hits[h].key;
    const uint32_t d = GS_KEY_MISMATCHES(key);
    if (d > mismatches) return GS_ERR_ARG;
    if (d != cur_d) {
      cur_d = d;
      idx_in_d = 0;
    }
    const int64_t i = idx_in_d++;
    if (max_off_targets != -1 && i >= max_off_targets) continue; /* raw index, :259 (break of that distance) */
    /* resolve_absolute, src/genomics/structures.cxx:7-52 */
    int64_t abs = hits[h].pos;
    char st = '+';
    if (abs < 0) {
      abs = -abs;
      st = '-';
    }
    uint32_t lo = 0, hi = gs->n_chr; /* first chromosome whose end exceeds abs */
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((uint64_t)abs < chr_end[mid + 1])
        hi = mid;
      else
        lo = mid + 1;
    }
    if (lo >= gs->n_chr) continue;
    const int64_t in_chr = abs - (int64_t)chr_end[lo];
    int64_t s, e;
    if (st == '+') {
      e = in_chr + 1;
      s = e - (int64_t)L - (int64_t)P + 1;
    } else {
      s = in_chr + 1;
      e = s + (int64_t)L + (int64_t)P - 1;
    }
    if (s < 0 || e > (int64_t)gs->chr_lengths[lo]) continue; /* boundary sentinel: no row (:280-283) */
    const size_t cl = strlen(gs->chr_names[lo]);
    if (!o.need(idl + 2 * (L + P) + cl + spl + 64)) return GS_ERR_NOMEM;
    o.put(k->id, idl);
    o.ch(',');
    if (start) o.put(k->pam, P);
    o.put(k->sequence, L);
    if (!start) o.put(k->pam, P);
    o.ch(',');
    o.put(gs->chr_names[lo], cl);
    o.ch(',');
    o.u64((uint64_t)s);
    o.ch(',');
    o.ch(st);
    o.ch(',');
    o.u64(d);
    if (complete) {
      o.ch(',');
      /* complement(match.sequence), decoded from the key (gs_decode_sequence + printer.hpp:232,264) */
      const uint64_t path = (key >> 1) & ((1ull << 59) - 1); /* key bits 59:1 */
      for (uint32_t t = 0; t < L; t++) {
        const char gq = start ? k->sequence[L - 1 - t] : k->sequence[t]; /* guide base under this step */
        const char qc = start ? gq : comp(gq);                            /* the query char (index.hpp:218) */
        const uint32_t code = (uint32_t)(path >> (57 - 2 * t)) & 3u;
        if (code == 0) {
          o.ch(comp(qc));
        } else {
          const int q = qc == 'A' ? 0 : qc == 'C' ? 1 : qc == 'G' ? 2 : qc == 'T' ? 3 : -1;
          if (q < 0) return GS_ERR_ARG;
          int a = (int)code - 1;
          if (a >= q) a++;
          o.ch((char)(comp(B[a]) | 0x20)); /* lower case marks the mismatch (index.hpp:243) */
        }
      }
      for (uint32_t u = 0; u < P; u++) {
        const uint32_t code = (uint32_t)(path >> (56 - 2 * L - 3 * u)) & 7u;
        if (code > 4) return GS_ERR_ARG;
        o.ch(PBC[code]);
      }
      o.put(",0,0", 4); /* rna_bulges, dna_bulges */
    }
    o.ch(',');
    o.put(sp, (size_t)spl);
    o.ch('\n');
  }
  return GS_OK;
}

extern "C" gs_status gs_format_guides_scored(const gs_genome_structure *gs, const gs_kmer *kmers, uint64_t n,
                                             const uint64_t *offsets, const gs_hit *hits, const float *specificity,
                                             const uint8_t *skip, uint32_t mismatches, uint32_t flags,
                                             int64_t max_off_targets, char **out_text, size_t *out_len) {
  if (!gs || (n && (!kmers || !offsets || !specificity)) || !out_text) return GS_ERR_ARG;
  try {
    std::vector<uint64_t> chr_end(gs->n_chr + 1, 0);
    for (uint32_t i = 0; i < gs->n_chr; i++) chr_end[i + 1] = chr_end[i] + gs->chr_lengths[i];
    outbuf o;
    if (!o.need(n ? (size_t)((offsets[n] - offsets[0]) * 96 + n * 16 + 64) : 64)) return GS_ERR_NOMEM;
    for (uint64_t g = 0; g < n; g++) {
      if (skip && skip[g]) continue;
      const gs_kmer *k = kmers + g;
      if (!k->id || !k->sequence || !k->pam) {
        free(o.p);
        return GS_ERR_ARG;
      }
      const uint64_t b = offsets[g], e = offsets[g + 1];
      gs_status rc;
      if (flags & GS_TEXT_SAM) { /* the SAM writer keeps the general encoder */
        char *tx = nullptr;
        size_t tl = 0;
        rc = format_guide_impl(gs, k, hits + b, e - b, mismatches, flags, max_off_targets, &tx, &tl, &specificity[g]);
        if (rc == GS_OK) {
          if (o.need(tl + 1)) o.put(tx, tl);
          else rc = GS_ERR_NOMEM;
          free(tx);
        }
      } else {
        rc = format_csv_fast(gs, chr_end.data(), k, hits + b, e - b, mismatches, flags, max_off_targets,
                             specificity[g], o);
      }
      if (rc != GS_OK) {
        free(o.p);
        return rc;
      }
    }
    if (!o.need(1)) {
      free(o.p);
      return GS_ERR_NOMEM;
    }
    o.p[o.len] = 0;
    *out_text = o.p;
    if (out_len) *out_len = o.len;
    return GS_OK;
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}

extern "C" gs_status gs_decode_sequence_ex(const gs_hit_ex *hit, char *out) {
  if (!hit || !out || hit->seq_len > 32) return GS_ERR_ARG;
  memcpy(out, hit->seq, hit->seq_len);
  out[hit->seq_len] = 0;
  return GS_OK;
}

extern "C" gs_status gs_format_guide_ex(const gs_genome_structure *gs, const gs_kmer *k,
                                        const gs_hit_ex *hits, uint64_t n_hits, uint32_t mismatches,
                                        uint32_t flags, int64_t max_off_targets, char **out_text,
                                        size_t *out_len) {
  if (!gs || !k || !k->id || !k->sequence || !k->pam || (n_hits && !hits) || !out_text)
    return GS_ERR_ARG;
  std::vector<std::vector<decoded_hit>> off(mismatches + 1);
  char buf[40];
  for (uint64_t h = 0; h < n_hits; h++) {
    if (hits[h].mismatches > mismatches) return GS_ERR_ARG;
    gs_status rc = gs_decode_sequence_ex(&hits[h], buf);
    if (rc != GS_OK) return rc;
    decoded_hit dh;
    dh.pos = hits[h].pos;
    dh.mismatches = hits[h].mismatches;
    dh.sequence = buf;
    dh.match_sequence = complement(dh.sequence);
    /* printer.hpp:139-143: substr(20, 3) whatever the bulges did to the alignment */
    dh.pam = dh.match_sequence.size() < 20 ? std::string() : dh.match_sequence.substr(20, 3);
    dh.rna_bulges = hits[h].rna_bulges;
    dh.dna_bulges = hits[h].dna_bulges;
    off[dh.mismatches].push_back(std::move(dh));
  }
  return format_decoded(gs, k, off, mismatches, flags, max_off_targets, out_text, out_len);
}

extern "C" gs_status gs_format_header(const gs_genome_structure *gs, uint32_t flags, char **out_text,
                                      size_t *out_len) {
  if (!gs || !out_text) return GS_ERR_ARG;
  std::string out;
  if (flags & GS_TEXT_SAM) { /* printer.hpp:173-179 */
    out += "@HD\tVN:1.0\tSO:unknown\n";
    out += "@PG\tID:Guidescan\tVN:" GS_GUIDESCAN_VERSION "\n";
    for (uint32_t i = 0; i < gs->n_chr; i++) {
      out += "@SQ\tSN:";
      out += gs->chr_names[i];
      out += "\tLN:";
      out += std::to_string(gs->chr_lengths[i]);
      out += '\n';
    }
  } else { /* printer.hpp:181-187 */
    out += "id,sequence,match_chrm,match_position,match_strand,match_distance";
    if (flags & GS_TEXT_COMPLETE) out += ",match_sequence,rna_bulges,dna_bulges";
    out += ",specificity\n";
  }
  char *p = (char *)malloc(out.size() + 1);
  if (!p) return GS_ERR_NOMEM;
  memcpy(p, out.data(), out.size());
  p[out.size()] = 0;
  *out_text = p;
  if (out_len) *out_len = out.size();
  return GS_OK;
}

extern "C" void gs_free(void *p) { free(p); }
