/*
 * guidescan_main.cpp -- C++ host that keeps guidescan's `index` / `enumerate` command line
 * (src/guidescan.cxx:28-95, 316-358) and database output, and calls the MI355X path through the
 * C-ABI (include/guidescan_amd.h).  Control plane only: no search logic lives here.
 *
 *   guidescan index  [--index PREFIX] [--store-sa] GENOME.fa
 *       writes PREFIX.gs (chromosome names/lengths, src/genomics/seq_io.cxx:112-122) and
 *       PREFIX.dna (= the reference's <fasta>.forward.dna: upper-cased concatenated sequence,
 *       seq_io.cxx:57-63).  The FM-index itself is built on the GPU when `enumerate` starts
 *       (~25 s at hg38 size, about what the reference needs to load its index files).
 *   guidescan enumerate PREFIX -f KMERS.csv -o OUT [-m 3] [-a PAM ...] [--format csv|sam|bam]
 *       [--mode succinct|complete] [--max-off-targets N] [--start] [--device D] [--gpus N] [--batch-size B]
 *       --gpus N: one index per device (devices D .. D+N-1), one host thread per device pulling batches
 *       from a shared queue, output written in input order (src/guidescan.cxx:226-251 is the
 *       reference's fan-out over threads).  On every device the search of batch i+1 overlaps the text
 *       formatting of batch i.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "guidescan_amd.h"
#include "bam_writer.hpp"

namespace {

struct genome_structure {
  std::vector<std::string> names;
  std::vector<uint64_t> lengths;
};

std::string trim_ws(const std::string &s) { /* seq_io.cxx:47-55: both ends, isspace */
  size_t b = 0, e = s.size();
  while (b < e && isspace((unsigned char)s[b])) b++;
  while (e > b && isspace((unsigned char)s[e - 1])) e--;
  return s.substr(b, e - b);
}

/* seq_io.cxx:57-63 (sequence) and :74-110 (structure) in one pass */
bool parse_fasta(const std::string &path, std::string &text, genome_structure &gs) {
  std::ifstream in(path);
  if (!in) return false;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') {
      const size_t sp = line.find(' ');
      gs.names.push_back(line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1));
      gs.lengths.push_back(0);
      continue;
    }
    if (!gs.lengths.empty()) gs.lengths.back() += line.length(); /* untrimmed: seq_io.cxx:100-104 */
    std::string t = trim_ws(line);
    for (auto &c : t) c = (char)toupper((unsigned char)c);
    text += t;
  }
  return true;
}

bool write_gs(const std::string &path, const genome_structure &gs) {
  std::ofstream out(path);
  if (!out) return false;
  for (size_t i = 0; i < gs.names.size(); i++) out << gs.names[i] << "\n" << gs.lengths[i] << "\n";
  return (bool)out;
}
bool read_gs(const std::string &path, genome_structure &gs, std::string &err) { /* seq_io.cxx:124-144 */
  std::ifstream in(path);
  if (!in) {
    err = "No genome structure file " + path;
    return false;
  }
  std::string name, len;
  while (std::getline(in, name) && std::getline(in, len)) {
    char *end = nullptr;
    const unsigned long long v = strtoull(len.c_str(), &end, 10);
    if (end == len.c_str() || (*end && *end != '\r')) {
      err = "malformed genome structure file " + path + ": length '" + len + "' of " + name;
      return false;
    }
    gs.names.push_back(name);
    gs.lengths.push_back(v);
  }
  return true;
}

struct kmer_row {
  std::string id, sequence, pam, chromosome, sense;
  long long position;
};

std::string trim_field(const std::string &s) { /* include/csv.hpp:1110-1116: ' ' and '\t' */
  size_t b = 0, e = s.size();
  while (b < e && (s[b] == ' ' || s[b] == '\t')) b++;
  while (e > b && (s[e - 1] == ' ' || s[e - 1] == '\t')) e--;
  return s.substr(b, e - b);
}
std::vector<std::string> split_csv(const std::string &line) {
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    const size_t c = line.find(',', b);
    out.push_back(trim_field(line.substr(b, c == std::string::npos ? std::string::npos : c - b)));
    if (c == std::string::npos) break;
    b = c + 1;
  }
  return out;
}
/* src/genomics/kmer.cxx:9-25 */
bool read_kmers(const std::string &path, std::vector<kmer_row> &rows, std::string &err) {
  std::ifstream in(path);
  if (!in) {
    err = "cannot open kmers file";
    return false;
  }
  std::string line;
  if (!std::getline(in, line)) {
    err = "empty kmers file";
    return false;
  }
  if (!line.empty() && line.back() == '\r') line.pop_back();
  const std::vector<std::string> header = split_csv(line);
  const char *want[6] = {"id", "sequence", "pam", "chromosome", "position", "sense"};
  int col[6];
  for (int i = 0; i < 6; i++) {
    col[i] = -1;
    for (size_t j = 0; j < header.size(); j++)
      if (header[j] == want[i]) col[i] = (int)j;
    if (col[i] < 0) {
      err = std::string("kmers file lacks column ") + want[i];
      return false;
    }
  }
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) continue;
    const std::vector<std::string> f = split_csv(line);
    if (f.size() < header.size()) {
      err = "kmers row with too few columns: " + line;
      return false;
    }
    kmer_row r;
    r.id = f[col[0]];
    r.sequence = f[col[1]];
    r.pam = f[col[2]];
    r.chromosome = f[col[3]];
    char *end = nullptr;
    r.position = strtoll(f[col[4]].c_str(), &end, 10);
    if (end == f[col[4]].c_str() || *end) {
      err = "kmers position is not an integer: " + f[col[4]];
      return false;
    }
    r.sense = f[col[5]];
    rows.push_back(std::move(r));
  }
  return true;
}

int usage() {
  std::cerr << "usage: guidescan index [--index PREFIX] [--store-sa [--device D]] GENOME.fa\n"
               "       guidescan enumerate PREFIX -f KMERS -o OUT [-m N] [-a PAM]... [--format csv|sam|bam]\n"
               "                 [--mode succinct|complete] [--max-off-targets N] [--start]\n"
               "                 [--rna-bulges N] [--dna-bulges N] [-t THRESHOLD] [-n FORMAT_THREADS]\n"
               "                 [--device D] [--gpus N] [--batch-size B]\n";
  return 2;
}

int do_index(int argc, char **argv) {
  std::string fasta, prefix;
  bool store_sa = false;
  int device = 0;
  for (int i = 0; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--index" && i + 1 < argc)
      prefix = argv[++i];
    else if (a == "--store-sa")
      store_sa = true;
    else if (a == "--device" && i + 1 < argc)
      device = atoi(argv[++i]);
    else if (!a.empty() && a[0] != '-')
      fasta = a;
    else
      return usage();
  }
  if (fasta.empty()) return usage();
  if (prefix.empty()) prefix = fasta + ".index"; /* src/guidescan.cxx:112-117 */
  std::string text;
  genome_structure gs;
  if (!parse_fasta(fasta, text, gs)) {
    std::cerr << "error: cannot read " << fasta << "\n";
    return 1;
  }
  if (!write_gs(prefix + ".gs", gs)) {
    std::cerr << "error: cannot write " << prefix << ".gs\n";
    return 1;
  }
  std::ofstream dna(prefix + ".dna", std::ios::binary);
  dna.write(text.data(), (std::streamsize)text.size());
  if (!dna) {
    std::cerr << "error: cannot write " << prefix << ".dna\n";
    return 1;
  }
  std::cout << "Wrote " << prefix << ".gs and " << prefix << ".dna (" << text.size() << " bases, "
            << gs.names.size() << " sequences)\n";
  if (store_sa) {
    /* the part of the index worth storing: both suffix arrays (built on the GPU now), so that
     * `enumerate` skips the sort; 8 bytes per base on disk */
    gs_index *ix = nullptr;
    gs_status rc = gs_index_build((const uint8_t *)text.data(), text.size(), device, &ix);
    if (rc == GS_OK) rc = gs_index_save_sa(ix, (const uint8_t *)text.data(), text.size(), (prefix + ".sa").c_str());
    if (ix) gs_index_close(ix);
    if (rc != GS_OK) {
      std::cerr << "error: " << gs_status_string(rc) << "\n";
      return 1;
    }
    std::cout << "Wrote " << prefix << ".sa\n";
  }
  return 0;
}

/* One batch of kmers with equal (L, P), as it moves through the pipeline: a device thread searches
 * and scores it, a formatting task turns the hit lists into text, the main thread writes the text
 * in input order. */
struct text_part { /* the lines of one contiguous range of a batch */
  char *p = nullptr; /* a buffer of the library (gs_free), or */
  size_t n = 0;
  std::string s;     /* lines gathered guide by guide */
};
struct batch {
  size_t lo = 0, hi = 0;
  std::vector<text_part> parts;
  std::string seqs, pams;
  gs_result *res = nullptr;
  gs_result_ex *resx = nullptr;       /* general path: every guide (bulges) or the flagged ones */
  std::vector<uint32_t> gen_of;       /* guide -> its position in resx, or ~0u */
  std::vector<float> spec;
  std::vector<char> skip;
  std::string error;
  bool handed = false; /* a device thread took it (set under enumerate_job::mtx) */
  bool ready = false;
};

struct enumerate_job {
  std::vector<kmer_row> kmers;
  std::vector<batch> batches;
  genome_structure gs;
  gs_genome_structure cgs{};
  std::vector<const char *> names;
  std::string alts;
  std::vector<uint32_t> alt_lens; /* symbols of each alt PAM: any length next to the guides' PAM (process.hpp:51-56) */
  uint32_t n_alt = 0;
  uint32_t mismatches = 3, rna = 0, dna = 0, tflags = 0, sflags = 0;
  long long max_off = -1, threshold = -1;
  unsigned fmt_threads = 1;
  std::mutex mtx;
  std::condition_variable cv;
  size_t next_batch = 0;   /* work queue of the device threads */
  size_t in_flight = 0;    /* searched but not yet written: bounds the host memory held by results */
  size_t max_in_flight = 2;
  double s_device = 0, s_format = 0, s_write = 0; /* seconds spent per stage (stages overlap) */
  /* --format bam: the SAM lines of the encoder, turned into BAM records and BGZF blocks by the formatting
   * threads (bam_writer.hpp; the reference leaves that step to `samtools view -b`, manual/manual.tex:581-582) */
  bool bam = false;
  std::map<std::string, int32_t> refid;
};

/* a part's SAM text -> BGZF-compressed BAM records (in place: the text is released) */
static bool part_to_bam(const enumerate_job &job, text_part &part) {
  std::string raw;
  bool ok = true;
  if (part.p) ok = bam::records(part.p, part.n, job.refid, raw);
  if (ok && !part.s.empty()) ok = bam::records(part.s.data(), part.s.size(), job.refid, raw);
  if (part.p) gs_free(part.p);
  part.p = nullptr;
  part.n = 0;
  part.s.clear();
  return ok && bam::bgzf_append(raw, part.s);
}

/* text of one batch from its hit lists: contiguous guide ranges formatted in parallel */
static void format_batch(enumerate_job &job, batch &b) {
  const size_t n = b.hi - b.lo;
  gs_result_view v;
  memset(&v, 0, sizeof v);
  if (b.res) gs_result_get(b.res, &v);
  const uint64_t *xoff = nullptr;
  const gs_hit_ex *xhits = nullptr;
  if (b.resx) gs_result_ex_get(b.resx, nullptr, &xoff, &xhits);
  unsigned nt = job.fmt_threads;
  if (nt < 1) nt = 1;
  if (nt > n) nt = (unsigned)n;
  b.parts.assign(nt, text_part());
  std::vector<gs_status> prc(nt, GS_OK);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt; t++) {
    pool.emplace_back([&, t]() {
      const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
      text_part &part = b.parts[t];
      if (b.gen_of.empty() && !b.resx) {
        /* the whole range in one buffer, rows written in place (no per-hit strings) */
        std::vector<gs_kmer> ck(hi - lo);
        for (size_t g = lo; g < hi; g++) {
          const kmer_row &k = job.kmers[b.lo + g];
          ck[g - lo] = gs_kmer{k.id.c_str(), k.sequence.c_str(), k.pam.c_str(), k.sense == "+" ? 1 : 0};
        }
        prc[t] = gs_format_guides_scored(&job.cgs, ck.data(), hi - lo, v.guide_offsets + lo, v.hits, b.spec.data() + lo,
                                         b.skip.empty() ? nullptr : (const uint8_t *)b.skip.data() + lo, job.mismatches,
                                         job.tflags | job.sflags, job.max_off, &part.p, &part.n);
        if (prc[t] == GS_OK && job.bam && !part_to_bam(job, part)) prc[t] = GS_ERR_FORMAT;
        return;
      }
      char *tx = nullptr;
      size_t tl = 0;
      for (size_t g = lo; g < hi; g++) {
        if (!b.skip.empty() && b.skip[g]) continue;
        const kmer_row &k = job.kmers[b.lo + g];
        gs_kmer ck{k.id.c_str(), k.sequence.c_str(), k.pam.c_str(), k.sense == "+" ? 1 : 0};
        gs_status r;
        const uint32_t gx = b.gen_of.empty() ? ~0u : b.gen_of[g];
        if (gx == ~0u) {
          const uint64_t s0 = v.guide_offsets[g], s1 = v.guide_offsets[g + 1];
          r = gs_format_guide_scored(&job.cgs, &ck, v.hits + s0, s1 - s0, job.mismatches, job.tflags | job.sflags,
                                     job.max_off, b.spec[g], &tx, &tl);
        } else {
          const uint64_t s0 = xoff[gx], s1 = xoff[gx + 1];
          r = gs_format_guide_ex(&job.cgs, &ck, xhits + s0, s1 - s0, job.mismatches, job.tflags | job.sflags,
                                 job.max_off, &tx, &tl);
        }
        if (r != GS_OK) {
          prc[t] = r;
          return;
        }
        part.s.append(tx, tl);
        gs_free(tx);
      }
      if (job.bam && !part_to_bam(job, part)) prc[t] = GS_ERR_FORMAT;
    });
  }
  for (auto &th : pool) th.join();
  for (unsigned t = 0; t < nt; t++)
    if (prc[t] != GS_OK && b.error.empty()) b.error = gs_status_string(prc[t]);
  if (b.res) gs_result_free(b.res);
  if (b.resx) gs_result_ex_free(b.resx);
  b.res = nullptr;
  b.resx = nullptr;
  b.spec = std::vector<float>();
}

/* the device side of one batch: threshold filter, search (fast path; general path for the guides it
 * flags, or for all of them with bulges), scoring */
static std::string search_batch_as(enumerate_job &job, gs_index *ix, batch &b, bool all_general, gs_status &rc);
static std::string search_batch(enumerate_job &job, gs_index *ix, batch &b) {
  const uint32_t L = (uint32_t)job.kmers[b.lo].sequence.size(), P = (uint32_t)job.kmers[b.lo].pam.size();
  /* Match sequences beyond the fast path's key: up to 59 bits (23-mers with a four-symbol PAM: 58) the table-seeded
   * kernels carry them; the reference-order walk (small genomes whose table is too shallow for the context arrays)
   * stops at 52 and says GS_ERR_UNSUPPORTED - then, and beyond 59 bits, the general path carries sequences as bytes */
  const uint32_t bits = 2 * L + 3 * P;
  gs_status rc = GS_OK;
  std::string err = search_batch_as(job, ix, b, bits > 59, rc);
  if (!err.empty() && rc == GS_ERR_UNSUPPORTED && bits > 52 && bits <= 59) {
    if (b.res) gs_result_free(b.res);
    b.res = nullptr;
    b.skip.clear();
    err = search_batch_as(job, ix, b, true, rc);
  }
  return err;
}
static std::string search_batch_as(enumerate_job &job, gs_index *ix, batch &b, bool all_general, gs_status &rc) {
  const size_t n = b.hi - b.lo;
  const uint32_t L = (uint32_t)job.kmers[b.lo].sequence.size(), P = (uint32_t)job.kmers[b.lo].pam.size();
  const uint32_t n_alt = P ? job.n_alt : 0;
  const bool bulges = job.rna > 0 || job.dna > 0;
  /* an alt PAM shorter or longer than the batch's PAM: the fixed-width fast path does not take it; the
   * general path searches every pattern at its own length, as the reference does */
  bool mixed = false;
  for (uint32_t j = 0; j < n_alt; j++) mixed = mixed || job.alt_lens[j] != P;
  if (all_general) mixed = true;
  /* --threshold t (process.hpp:66-76): a guide with more than one hit within t mismatches (both
   * indexes, bulges off; counted per PAM pattern, before duplicate sequences collapse) is dropped
   * before the real search */
  if (job.threshold > 0 && mixed) {
    gs_result_ex *cx = nullptr;
    rc = gs_enumerate_general_pams(ix, b.seqs.data(), n, L, b.pams.data(), P, job.alts.data(), job.alt_lens.data(), n_alt,
                                   (uint32_t)job.threshold, 0, 0, job.sflags, &cx);
    if (rc != GS_OK) return gs_status_string(rc);
    const uint32_t *xraw = nullptr;
    gs_result_ex_raw_hits(cx, &xraw);
    b.skip.assign(n, 0);
    for (size_t g = 0; g < n; g++) b.skip[g] = xraw[g] > 1;
    gs_result_ex_free(cx);
  } else if (job.threshold > 0) {
    gs_result *cres = nullptr;
    rc = gs_enumerate(ix, b.seqs.data(), n, L, b.pams.data(), P, job.alts.data(), n_alt, (uint32_t)job.threshold,
                      job.sflags | GS_FLAG_RAW_COUNTS, &cres);
    if (rc != GS_OK) return gs_status_string(rc);
    gs_result_view cv;
    gs_result_get(cres, &cv);
    b.skip.assign(n, 0);
    /* raw counts: a site that two PAM patterns of the list match counts twice, as off_target_counter does */
    for (size_t g = 0; g < n; g++) b.skip[g] = cv.raw_hits[g] > 1;
    /* guides the fast path cannot count: through the general path (rare, exact) */
    if (cv.n_unsupported) {
      std::string s2, p2;
      std::vector<size_t> idx;
      for (size_t g = 0; g < n; g++)
        if (cv.guide_flags[g] & GS_GUIDE_NEEDS_GENERAL) {
          idx.push_back(g);
          s2.append(b.seqs, g * L, L);
          p2.append(b.pams, g * P, P);
        }
      gs_result_ex *cx = nullptr;
      rc = gs_enumerate_general(ix, s2.data(), idx.size(), L, p2.data(), P, job.alts.data(), n_alt,
                                (uint32_t)job.threshold, 0, 0, job.sflags, &cx);
      if (rc != GS_OK) {
        gs_result_free(cres);
        return gs_status_string(rc);
      }
      const uint32_t *xraw = nullptr;
      gs_result_ex_raw_hits(cx, &xraw);
      for (size_t j = 0; j < idx.size(); j++) b.skip[idx[j]] = xraw[j] > 1;
      gs_result_ex_free(cx);
    }
    gs_result_free(cres);
  }
  if (bulges || mixed) {
    /* bulge-aware search: index.hpp:250-375 behind gs_enumerate_general; alt PAMs of other lengths: the same entry */
    rc = gs_enumerate_general_pams(ix, b.seqs.data(), n, L, b.pams.data(), P, job.alts.data(), job.alt_lens.data(), n_alt,
                                   job.mismatches, job.rna, job.dna, job.sflags, &b.resx);
    if (rc != GS_OK) return gs_status_string(rc);
    b.gen_of.resize(n);
    for (size_t g = 0; g < n; g++) b.gen_of[g] = (uint32_t)g;
    return "";
  }
  rc = gs_enumerate(ix, b.seqs.data(), n, L, b.pams.data(), P, job.alts.data(), n_alt, job.mismatches, job.sflags,
                    &b.res);
  if (rc != GS_OK) return gs_status_string(rc);
  gs_result_view v;
  gs_result_get(b.res, &v);
  if (v.n_unsupported) {
    /* guides with symbols the fast path does not encode (index.hpp:218-247): the general path, for
     * them alone; the rest of the batch is untouched */
    std::string s2, p2;
    b.gen_of.assign(n, ~0u);
    uint32_t k = 0;
    for (size_t g = 0; g < n; g++)
      if (v.guide_flags[g] & GS_GUIDE_NEEDS_GENERAL) {
        b.gen_of[g] = k++;
        s2.append(b.seqs, g * L, L);
        p2.append(b.pams, g * P, P);
      }
    rc = gs_enumerate_general(ix, s2.data(), k, L, p2.data(), P, job.alts.data(), n_alt, job.mismatches, 0, 0,
                              job.sflags, &b.resx);
    if (rc != GS_OK) return gs_status_string(rc);
  }
  /* specificity of every guide of the batch on the device (printer.hpp:98-170, 251-297 behind
   * gs_score): the formatting threads only print */
  b.spec.resize(n);
  rc = gs_score(ix, b.seqs.data(), n, L, P, job.tflags | job.sflags, job.max_off, &job.cgs, v.guide_offsets, v.hits,
                nullptr, b.spec.data());
  if (rc != GS_OK) return gs_status_string(rc);
  return "";
}

int do_enumerate(int argc, char **argv) {
  std::string prefix, kmers_file, output, format = "csv", mode = "complete";
  std::vector<std::string> alt_pams;
  long long mismatches = 3, max_off = -1, threshold = -1, rna = 0, dna = 0;
  int device = 0, gpus = 1;
  size_t batch_size = 0;
  unsigned fmt_threads = 0;
  bool start = false;
  for (int i = 0; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&](const char *what) -> const char * {
      if (i + 1 >= argc) {
        std::cerr << "error: " << what << " needs a value\n";
        exit(2);
      }
      return argv[++i];
    };
    if (a == "-f" || a == "--kmers-file") kmers_file = need("-f");
    else if (a == "-o" || a == "--output") output = need("-o");
    else if (a == "-m" || a == "--mismatches") mismatches = atoll(need("-m"));
    else if (a == "-a" || a == "--alt-pam") alt_pams.push_back(need("-a"));
    else if (a == "-n" || a == "--threads") fmt_threads = (unsigned)atoi(need("-n")); /* text formatting threads; the search runs on the GPU */
    else if (a == "-t" || a == "--threshold") threshold = atoll(need("-t"));
    else if (a == "--rna-bulges") rna = atoll(need("--rna-bulges"));
    else if (a == "--dna-bulges") dna = atoll(need("--dna-bulges"));
    else if (a == "--max-off-targets") max_off = atoll(need("--max-off-targets"));
    else if (a == "--format") format = need("--format");
    else if (a == "--mode") mode = need("--mode");
    else if (a == "--start") start = true;
    else if (a == "--device") device = atoi(need("--device"));
    else if (a == "--gpus") gpus = atoi(need("--gpus"));
    else if (a == "--batch-size") batch_size = (size_t)atoll(need("--batch-size"));
    else if (!a.empty() && a[0] != '-' && prefix.empty()) prefix = a;
    else return usage();
  }
  if (prefix.empty() || kmers_file.empty() || output.empty()) return usage();
  if ((format != "csv" && format != "sam" && format != "bam") || (mode != "succinct" && mode != "complete")) return usage();
  if (gpus < 1 || mismatches < 0 || rna < 0 || dna < 0) return usage();
  enumerate_job job;
  std::string err;
  if (!read_gs(prefix + ".gs", job.gs, err)) {
    std::cerr << "error: " << err << "\n";
    return 1;
  }
  /* PREFIX.dna (this tool's `index`) or, for indices made by the reference, PREFIX.forward */
  std::string text;
  bool from_sdsl = false;
  {
    std::ifstream dna_in(prefix + ".dna", std::ios::binary | std::ios::ate);
    if (dna_in) {
      text.resize((size_t)dna_in.tellg());
      dna_in.seekg(0);
      dna_in.read(&text[0], (std::streamsize)text.size());
    } else if (std::ifstream(prefix + ".forward")) {
      from_sdsl = true;
    } else {
      std::cerr << "error: neither " << prefix << ".dna nor " << prefix << ".forward exists\n";
      return 1;
    }
  }
  if (!read_kmers(kmers_file, job.kmers, err)) {
    std::cerr << "error: " << err << "\n";
    return 1;
  }
  std::cout << "Read in " << job.kmers.size() << " kmer(s).\n";

  /* one index per device, built side by side (src/guidescan.cxx:226-251 fans the guides out over
   * threads that share one index; here every GPU holds its own copy in HBM) */
  auto t0 = std::chrono::steady_clock::now();
  /* GS_CLI_SAME_DEVICE=1: every worker builds its index on `device` itself - the fan-out, the batch queue
   * and the ordered writer run with N workers on a box with one GPU (tests) */
  const int dev_step = getenv("GS_CLI_SAME_DEVICE") ? 0 : 1;
  std::vector<gs_index *> ix((size_t)gpus, nullptr);
  {
    std::vector<gs_status> brc((size_t)gpus, GS_OK);
    std::vector<std::string> bmsg((size_t)gpus);
    std::vector<std::thread> bt;
    for (int d = 0; d < gpus; d++)
      bt.emplace_back([&, d]() {
        if (from_sdsl) {
          brc[d] = gs_index_open_sdsl(prefix.c_str(), device + d * dev_step, &ix[d]);
        } else {
          /* stored suffix arrays (guidescan index --store-sa) skip the sort; a file that does not
           * belong to this text is ignored */
          brc[d] = GS_ERR_IO;
          if (std::ifstream(prefix + ".sa"))
            brc[d] = gs_index_open_sa((const uint8_t *)text.data(), text.size(), (prefix + ".sa").c_str(), device + d * dev_step, &ix[d]);
          if (brc[d] == GS_ERR_IO || brc[d] == GS_ERR_FORMAT)
            brc[d] = gs_index_build((const uint8_t *)text.data(), text.size(), device + d * dev_step, &ix[d]);
        }
        if (brc[d] != GS_OK) bmsg[d] = gs_status_string(brc[d]);
      });
    for (auto &th : bt) th.join();
    for (int d = 0; d < gpus; d++)
      if (brc[d] != GS_OK) {
        std::cerr << "error: device " << device + d << ": " << bmsg[d] << "\n";
        for (gs_index *p : ix) gs_index_close(p);
        return 1;
      }
  }
  text = std::string();
  std::cout << "Built the forward and reverse index on " << gpus << " device(s) from " << device << " in "
            << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s\n";

  const int fd = open(output.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) {
    std::cerr << "error: cannot write " << output << "\n";
    return 1;
  }
  uint64_t file_off = 0;
  bool write_ok = true;
  auto pwrite_all = [&](const char *p, size_t n, uint64_t at) -> bool {
    while (n) {
      const ssize_t w = pwrite(fd, p, n, (off_t)at);
      if (w <= 0) return false;
      p += w;
      n -= (size_t)w;
      at += (uint64_t)w;
    }
    return true;
  };
  for (auto &n : job.gs.names) job.names.push_back(n.c_str());
  job.cgs = gs_genome_structure{job.names.data(), job.gs.lengths.data(), (uint32_t)job.names.size()};
  job.tflags = (format != "csv" ? GS_TEXT_SAM : 0u) | (mode == "complete" ? GS_TEXT_COMPLETE : 0u);
  job.bam = format == "bam";
  for (size_t i = 0; i < job.gs.names.size(); i++) job.refid[job.gs.names[i]] = (int32_t)i;
  job.sflags = start ? GS_FLAG_PAM_AT_START : 0u;
  {
    /* the PAM-pair and deep tables cost ~0.23 s per device at hg38 size (the strand tables' rotated copies, which a
     * job without them builds instead: ~0.1 s) and - since the seeding launches read them (gs_seed.hip) - save ~85 ms
     * per million guides in batches of 2^17 (3.5 against 14 ms per batch): jobs below ~1.9 M guides per device at
     * that size go without (the library's default is to build them).  Measured with bench.py's e2e row, 1 M guides:
     * 0.35 s without the tables, 0.45 s with them - the first batch's build also keeps the writer waiting.  (Until
     * round 6 the saving was 15 ms per million and the bar stood at 15 M guides.) */
    uint64_t glen = 0;
    for (uint64_t l : job.gs.lengths) glen += l;
    if ((double)job.kmers.size() / (double)gpus < 6e-4 * (double)glen) job.sflags |= GS_FLAG_NO_NEW_TABLES;
  }
  job.mismatches = (uint32_t)mismatches;
  job.rna = (uint32_t)rna;
  job.dna = (uint32_t)dna;
  job.max_off = max_off;
  job.threshold = threshold;
  job.fmt_threads = fmt_threads ? fmt_threads : std::max(1u, std::thread::hardware_concurrency());
  job.max_in_flight = 2 * (size_t)gpus;
  char *txt = nullptr;
  size_t len = 0;
  gs_format_header(&job.cgs, job.tflags, &txt, &len);
  if (job.bam) { /* the header block: magic, the SAM header text, the reference list */
    std::string hz;
    write_ok = bam::bgzf_append(bam::header(std::string(txt, len), job.gs.names, job.gs.lengths), hz) &&
               pwrite_all(hz.data(), hz.size(), file_off);
    file_off += hz.size();
  } else {
    write_ok = pwrite_all(txt, len, file_off);
    file_off += len;
  }
  gs_free(txt);

  /* batches of equal (L, P) in input order: the device call takes fixed-width rows.  The hit lists
   * of a batch live in HBM and on the host until it is written: ~13 hits per guide at <= 3
   * mismatches, ~1.4e3 at 5, ~1.1e4 at 6 on a human-sized genome, so deeper searches take smaller batches */
  /* At <= 3 mismatches a batch of 2^17 guides: the search is 3 ms of device time per batch at hg38 size (a million guides in
   * one batch: 17.5 ms, in eight: 24), but its 1.1 GB of text took 0.22 s to format and 0.19 s to write BEHIND the search when
   * the set was one batch - three stages that overlap only from batch to batch (bench.py's e2e_cli row: 0.60 -> 0.4 s). */
  if (!batch_size) batch_size = mismatches <= 3 ? (1u << 17) : mismatches == 4 ? (1u << 18) : mismatches == 5 ? (1u << 16) : (1u << 13);
  /* searched, being formatted, being written: three batches in flight per device where a batch's hit lists and text are
   * small (m <= 4: ~0.2 GB), two where they are gigabytes */
  if (mismatches <= 4) job.max_in_flight = 3 * (size_t)gpus;
  for (size_t done = 0; done < job.kmers.size();) {
    const size_t L = job.kmers[done].sequence.size(), P = job.kmers[done].pam.size();
    batch b;
    b.lo = done;
    size_t end = done;
    while (end < job.kmers.size() && end - done < batch_size && job.kmers[end].sequence.size() == L &&
           job.kmers[end].pam.size() == P)
      end++;
    b.hi = end;
    job.batches.push_back(std::move(b));
    done = end;
  }
  /* alt PAMs are searched whatever their length, next to each guide's own PAM (process.hpp:51-56): a batch
   * whose PAM length they all share goes through the fast path, any other through the general path */
  for (auto &a : alt_pams) {
    if (a.empty() || a.size() > 8) {
      std::cerr << "error: alt PAM " << a << ": 1 to 8 symbols\n";
      for (gs_index *p : ix) gs_index_close(p);
      return 1;
    }
    job.alts += a;
    job.alt_lens.push_back((uint32_t)a.size());
    job.n_alt++;
  }

  t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> formatters(job.batches.size());
  std::vector<std::thread> devs;
  for (int d = 0; d < gpus; d++)
    devs.emplace_back([&, d]() {
      for (;;) {
        size_t bi;
        {
          std::unique_lock<std::mutex> lk(job.mtx);
          job.cv.wait(lk, [&] { return job.in_flight < job.max_in_flight || job.next_batch >= job.batches.size(); });
          if (job.next_batch >= job.batches.size()) return;
          bi = job.next_batch++;
          job.batches[bi].handed = true;
          job.in_flight++;
        }
        batch &b = job.batches[bi];
        for (size_t g = b.lo; g < b.hi; g++) {
          b.seqs += job.kmers[g].sequence;
          b.pams += job.kmers[g].pam;
        }
        const auto ts = std::chrono::steady_clock::now();
        b.error = search_batch(job, ix[d], b);
        {
          std::lock_guard<std::mutex> lk(job.mtx);
          job.s_device += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
        }
        /* text formatting of this batch overlaps the device work of the next one.  The thread object is
         * stored under the mutex the formatter takes before it sets `ready`: the writer joins
         * formatters[bi] only after it has seen `ready`, i.e. after this assignment is complete */
        std::lock_guard<std::mutex> lk(job.mtx);
        formatters[bi] = std::thread([&job, &b]() {
          const auto tf = std::chrono::steady_clock::now();
          if (b.error.empty()) format_batch(job, b);
          std::lock_guard<std::mutex> lk2(job.mtx);
          job.s_format += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count();
          b.ready = true;
          job.cv.notify_all();
        });
      }
    });
  int rcode = 0;
  for (size_t bi = 0; bi < job.batches.size(); bi++) {
    batch &b = job.batches[bi];
    {
      /* after an error no more batches are handed out (next_batch is moved to the end below): a batch
       * that no device thread took will never become ready, and neither will any behind it */
      std::unique_lock<std::mutex> lk(job.mtx);
      job.cv.wait(lk, [&] { return b.ready || (!b.handed && job.next_batch >= job.batches.size()); });
      if (!b.ready) break;
    }
    formatters[bi].join();
    if (!b.error.empty()) {
      if (!rcode) std::cerr << "error: " << b.error << "\n";
      rcode = 1;
    } else if (!rcode) { /* after a failed batch nothing more is written: rows behind a hole are not a database */
      const auto tw = std::chrono::steady_clock::now();
      /* one writer: page-cache writes to one file serialise on the inode anyway (eight pwrite
       * threads were slower on tmpfs, 1.6 s against 1.1 s for 4.6 GB) */
      for (const text_part &pt : b.parts) {
        if (pt.p && !pwrite_all(pt.p, pt.n, file_off)) write_ok = false;
        file_off += pt.n;
        if (!pt.s.empty() && !pwrite_all(pt.s.data(), pt.s.size(), file_off)) write_ok = false;
        file_off += pt.s.size();
      }
      job.s_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
    }
    for (text_part &pt : b.parts) gs_free(pt.p);
    b.parts = std::vector<text_part>();
    b.seqs = std::string();
    b.pams = std::string();
    {
      std::lock_guard<std::mutex> lk(job.mtx);
      job.in_flight--;
      if (rcode) job.next_batch = job.batches.size(); /* stop handing out work */
      job.cv.notify_all();
    }
  }
  for (auto &th : devs) th.join();
  if (job.bam && !rcode) { /* the empty block that ends a BGZF file */
    std::string eof;
    bam::bgzf_eof(eof);
    if (!pwrite_all(eof.data(), eof.size(), file_off)) write_ok = false;
    file_off += eof.size();
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout << "Processed " << job.kmers.size() << " kmers in " << secs << " seconds.\n";
  std::cout << "Stages (overlapping): device " << job.s_device << " s, text formatting " << job.s_format
            << " s, file writes " << job.s_write << " s\n";
  for (gs_index *p : ix) gs_index_close(p);
  /* only a regular file is ever removed: -o /dev/stdout, a FIFO or a device node stays (written through pwrite they
   * fail with ESPIPE, and unlinking them would delete the node itself) */
  struct stat fst;
  const bool regular = fstat(fd, &fst) == 0 && S_ISREG(fst.st_mode);
  if (close(fd) != 0) write_ok = false;
  if (!write_ok) std::cerr << "error: short write to " << output << "\n";
  if (rcode || !write_ok) {
    /* a run that failed leaves no file that looks like a database (CSV/SAM rows up to the failed batch, a BAM
     * without its end-of-file block) */
    if (regular && unlink(output.c_str()) == 0) std::cerr << "error: " << output << " removed (incomplete)\n";
  }
  return (write_ok && !rcode) ? 0 : 1;
}

/* guidescan sam2bam IN.sam OUT.bam: the encoder of `enumerate --format bam` on a SAM file (what the reference's
 * manual does with `samtools view -b`); the references come from the file's @SQ lines */
int do_sam2bam(int argc, char **argv) {
  if (argc != 2) return usage();
  std::ifstream in(argv[0], std::ios::binary);
  if (!in) {
    std::cerr << "error: cannot read " << argv[0] << "\n";
    return 1;
  }
  std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::string head;
  std::vector<std::string> names;
  std::vector<uint64_t> lengths;
  std::map<std::string, int32_t> refid;
  size_t at = 0;
  while (at < text.size() && text[at] == '@') {
    size_t e = text.find('\n', at);
    e = e == std::string::npos ? text.size() : e + 1;
    const std::string line = text.substr(at, e - at);
    if (line.compare(0, 3, "@SQ") == 0) {
      std::string sn;
      uint64_t ln = 0;
      std::stringstream ss(line);
      std::string f;
      while (std::getline(ss, f, '\t')) {
        while (!f.empty() && (f.back() == '\n' || f.back() == '\r')) f.pop_back();
        if (f.compare(0, 3, "SN:") == 0) sn = f.substr(3);
        if (f.compare(0, 3, "LN:") == 0) ln = strtoull(f.c_str() + 3, nullptr, 10);
      }
      refid[sn] = (int32_t)names.size();
      names.push_back(sn);
      lengths.push_back(ln);
    }
    head += line;
    at = e;
  }
  std::string out, raw;
  if (!bam::bgzf_append(bam::header(head, names, lengths), out) || !bam::records(text.data() + at, text.size() - at, refid, raw) ||
      !bam::bgzf_append(raw, out)) {
    std::cerr << "error: malformed SAM line in " << argv[0] << "\n";
    return 1;
  }
  bam::bgzf_eof(out);
  std::ofstream o(argv[1], std::ios::binary);
  o.write(out.data(), (std::streamsize)out.size());
  return o.good() ? 0 : 1;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc >= 2 && (!strcmp(argv[1], "--version") || !strcmp(argv[1], "-v"))) {
    std::cout << "2.0.0 (" << gs_version() << ")\n";
    return 0;
  }
  if (argc < 2) return usage();
  if (!strcmp(argv[1], "sam2bam")) return do_sam2bam(argc - 2, argv + 2);
  if (!strcmp(argv[1], "index")) return do_index(argc - 2, argv + 2);
  if (!strcmp(argv[1], "enumerate")) return do_enumerate(argc - 2, argv + 2);
  return usage();
}
