/*
 * bam_writer.hpp -- SAM text -> BAM records in BGZF blocks (zlib), for `guidescan enumerate --format bam`.
 *
 * The reference writes SAM text only (include/genomics/printer.hpp:302-360) and its manual sends users to
 * `samtools view -b` for the BAM database (manual/manual.tex:581-582).  This header does that step in the
 * host: the SAM lines come from the same encoder as `--format sam` (gs_format_guides_scored), so a BAM
 * file decodes to exactly the SAM file.  Layout per the SAM/BAM specification (SAMv1 section 4): header
 * block {magic "BAM\1", header text, reference names and lengths}, one alignment block per line
 * (refID, 0-based pos, bin from reg2bin, mapq, flag, read name, CIGAR ops, 4-bit sequence, quality 0xFF for
 * '*', typed tags: integers in the smallest type that holds them, sp:f as float32, of:H as a hex string),
 * all inside BGZF blocks of at most 64 KiB, closed by the empty end-of-file block.
 */
#ifndef GS_BAM_WRITER_HPP
#define GS_BAM_WRITER_HPP

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace bam {

inline void put32(std::string &o, uint32_t v) {
  char b[4] = {(char)(v & 255), (char)((v >> 8) & 255), (char)((v >> 16) & 255), (char)(v >> 24)};
  o.append(b, 4);
}
inline void put16(std::string &o, uint32_t v) {
  char b[2] = {(char)(v & 255), (char)((v >> 8) & 255)};
  o.append(b, 2);
}

/* BGZF: `raw` cut into blocks of at most 0xff00 bytes, each a gzip member with the BC extra field */
inline bool bgzf_append(const std::string &raw, std::string &out) {
  const size_t BLOCK = 0xff00;
  for (size_t at = 0; at < raw.size(); at += BLOCK) {
    const size_t n = raw.size() - at < BLOCK ? raw.size() - at : BLOCK;
    unsigned char buf[0x10000 + 64];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = (Bytef *)(raw.data() + at);
    zs.avail_in = (uInt)n;
    zs.next_out = buf;
    zs.avail_out = sizeof buf;
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen + 26 > 0x10000) return false;
    const uint32_t bsize = (uint32_t)(clen + 25); /* total block size - 1 */
    static const unsigned char head[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    out.append((const char *)head, 16);
    put16(out, bsize);
    out.append((const char *)buf, clen);
    put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(raw.data() + at), (uInt)n));
    put32(out, (uint32_t)n);
  }
  return true;
}
inline void bgzf_eof(std::string &out) {
  static const unsigned char eof[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  out.append((const char *)eof, 28);
}

/* uncompressed BAM header from the SAM header text (@SQ lines give the references) */
inline std::string header(const std::string &sam_header, const std::vector<std::string> &names,
                          const std::vector<uint64_t> &lengths) {
  std::string o("BAM\1", 4);
  put32(o, (uint32_t)sam_header.size());
  o += sam_header;
  put32(o, (uint32_t)names.size());
  for (size_t i = 0; i < names.size(); i++) {
    put32(o, (uint32_t)names[i].size() + 1);
    o.append(names[i].c_str(), names[i].size() + 1);
    put32(o, (uint32_t)lengths[i]);
  }
  return o;
}

inline uint32_t reg2bin(int64_t beg, int64_t end) { /* SAMv1 section 5.3 */
  --end;
  if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

/* one SAM line (no newline) -> one BAM alignment block appended to `o`; false on a malformed line */
inline bool record(const char *line, size_t len, const std::map<std::string, int32_t> &refid, std::string &o) {
  std::vector<std::pair<const char *, size_t>> f;
  size_t at = 0;
  while (at <= len) {
    const char *tab = (const char *)memchr(line + at, '\t', len - at);
    const size_t e = tab ? (size_t)(tab - line) : len;
    f.push_back({line + at, e - at});
    at = e + 1;
    if (!tab) break;
  }
  if (f.size() < 11) return false;
  auto str = [&](size_t i) { return std::string(f[i].first, f[i].second); };
  const std::string qname = str(0), rname = str(2), cigar = str(5), rnext = str(6), seq = str(9), qual = str(10);
  const uint32_t flag = (uint32_t)strtoul(str(1).c_str(), nullptr, 10);
  const int64_t pos = strtoll(str(3).c_str(), nullptr, 10) - 1; /* 0-based; -1 when '0' */
  const uint32_t mapq = (uint32_t)strtoul(str(4).c_str(), nullptr, 10);
  const int64_t pnext = strtoll(str(7).c_str(), nullptr, 10) - 1;
  const int32_t tlen = (int32_t)strtol(str(8).c_str(), nullptr, 10);
  int32_t rid = -1, nrid = -1;
  /* the reference writes an EMPTY reference name (and position 0) for a hit dropped at a chromosome boundary
   * (structures.cxx:46-48 -> printer.hpp:330): BAM stores a reference index, so that line carries -1 = '*' */
  if (rname != "*" && !rname.empty()) {
    auto it = refid.find(rname);
    if (it == refid.end()) return false;
    rid = it->second;
  }
  if (rnext == "=")
    nrid = rid;
  else if (rnext != "*") {
    auto it = refid.find(rnext);
    if (it == refid.end()) return false;
    nrid = it->second;
  }
  /* CIGAR */
  std::vector<uint32_t> ops;
  int64_t ref_len = 0;
  if (cigar != "*") {
    static const char *OPS = "MIDNSHP=X";
    uint64_t num = 0;
    bool any = false;
    for (char c : cigar) {
      if (c >= '0' && c <= '9') {
        num = num * 10 + (uint64_t)(c - '0');
        any = true;
        continue;
      }
      const char *p = strchr(OPS, c);
      if (!p || !any) return false;
      const uint32_t op = (uint32_t)(p - OPS);
      ops.push_back((uint32_t)(num << 4) | op);
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += (int64_t)num;
      num = 0;
      any = false;
    }
  }
  const bool has_seq = seq != "*";
  const uint32_t l_seq = has_seq ? (uint32_t)seq.size() : 0u;
  /* what BAM's fixed-width fields cannot hold is an error, not a truncated byte: l_read_name counts the NUL in
   * eight bits, MAPQ is one byte */
  if (qname.size() > 254 || mapq > 255) return false;
  std::string body;
  put32(body, (uint32_t)rid);
  put32(body, (uint32_t)(int32_t)pos);
  body.push_back((char)(qname.size() + 1));
  body.push_back((char)mapq);
  put16(body, pos < 0 ? 4680u : reg2bin(pos, pos + (ref_len > 0 ? ref_len : 1)));
  put16(body, (uint32_t)ops.size());
  put16(body, flag);
  put32(body, l_seq);
  put32(body, (uint32_t)nrid);
  put32(body, (uint32_t)(int32_t)pnext);
  put32(body, (uint32_t)tlen);
  body.append(qname.c_str(), qname.size() + 1);
  for (uint32_t op : ops) put32(body, op);
  if (has_seq) {
    static const char *CODE = "=ACMGRSVTWYHKDBN";
    for (uint32_t i = 0; i < l_seq; i += 2) {
      auto code = [&](char c) -> uint32_t {
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        const char *p = strchr(CODE, c);
        return p && c ? (uint32_t)(p - CODE) : 15u;
      };
      const uint32_t hi = code(seq[i]), lo = i + 1 < l_seq ? code(seq[i + 1]) : 0u;
      body.push_back((char)((hi << 4) | lo));
    }
    if (qual == "*")
      body.append(l_seq, (char)0xFF);
    else
      for (uint32_t i = 0; i < l_seq; i++) body.push_back((char)(i < qual.size() ? qual[i] - 33 : 0xFF));
  }
  /* tags TAG:TYPE:VALUE */
  for (size_t i = 11; i < f.size(); i++) {
    if (f[i].second < 5 || f[i].first[2] != ':' || f[i].first[4] != ':') return false;
    const char *t = f[i].first;
    const std::string val(t + 5, f[i].second - 5);
    body.append(t, 2);
    switch (t[3]) {
      case 'i': {
        const long long v = strtoll(val.c_str(), nullptr, 10);
        if (v < -2147483648ll || v > 4294967295ll) return false; /* no integer tag type is wider than 32 bits */
        if (v >= 0 && v <= 255) {
          body.push_back('C');
          body.push_back((char)v);
        } else if (v >= -128 && v < 0) {
          body.push_back('c');
          body.push_back((char)v);
        } else if (v >= 0 && v <= 65535) {
          body.push_back('S');
          put16(body, (uint32_t)v);
        } else if (v >= -32768 && v < 0) {
          body.push_back('s');
          put16(body, (uint32_t)v);
        } else if (v >= 0) {
          body.push_back('I');
          put32(body, (uint32_t)v);
        } else {
          body.push_back('i');
          put32(body, (uint32_t)v);
        }
        break;
      }
      case 'f': {
        const float fv = strtof(val.c_str(), nullptr);
        uint32_t bits;
        memcpy(&bits, &fv, 4);
        body.push_back('f');
        put32(body, bits);
        break;
      }
      case 'A':
        body.push_back('A');
        body.push_back(val.empty() ? ' ' : val[0]);
        break;
      case 'Z':
      case 'H':
        body.push_back(t[3]);
        body.append(val.c_str(), val.size() + 1);
        break;
      default:
        return false;
    }
  }
  put32(o, (uint32_t)body.size());
  o += body;
  return true;
}

/* every line of a block of SAM text (lines end in '\n'; header lines '@...' are skipped) */
inline bool records(const char *text, size_t len, const std::map<std::string, int32_t> &refid, std::string &o) {
  size_t at = 0;
  while (at < len) {
    const char *nl = (const char *)memchr(text + at, '\n', len - at);
    const size_t e = nl ? (size_t)(nl - text) : len;
    if (e > at && text[at] != '@' && !record(text + at, e - at, refid, o)) return false;
    at = e + 1;
  }
  return true;
}

}  // namespace bam
#endif
