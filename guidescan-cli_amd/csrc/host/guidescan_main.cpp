/*
 * guidescan_main.cpp -- C++ host that keeps guidescan's `index` / `enumerate` command line
 * (src/guidescan.cxx:28-95, 316-358) and database output, and calls the MI355X path through the
 * C-ABI (include/guidescan_amd.h).  Control plane only: no search logic lives here.
 *
 *   guidescan index  [--index PREFIX] GENOME.fa
 *       writes PREFIX.gs (chromosome names/lengths, src/genomics/seq_io.cxx:112-122) and
 *       PREFIX.dna (= the reference's <fasta>.forward.dna: upper-cased concatenated sequence,
 *       seq_io.cxx:57-63).  The FM-index itself is built on the GPU when `enumerate` starts
 *       (~25 s at hg38 size, about what the reference needs to load its index files).
 *   guidescan enumerate PREFIX -f KMERS.csv -o OUT [-m 3] [-a PAM ...] [--format csv|sam]
 *       [--mode succinct|complete] [--max-off-targets N] [--start] [--device D] [--batch-size B]
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "guidescan_amd.h"

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
bool read_gs(const std::string &path, genome_structure &gs) { /* seq_io.cxx:124-144 */
  std::ifstream in(path);
  if (!in) return false;
  std::string name, len;
  while (std::getline(in, name) && std::getline(in, len)) {
    gs.names.push_back(name);
    gs.lengths.push_back(std::stoull(len));
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
  std::cerr << "usage: guidescan index [--index PREFIX] GENOME.fa\n"
               "       guidescan enumerate PREFIX -f KMERS -o OUT [-m N] [-a PAM]... [--format csv|sam]\n"
               "                 [--mode succinct|complete] [--max-off-targets N] [--start]\n"
               "                 [--device D] [--batch-size B]\n";
  return 2;
}

int do_index(int argc, char **argv) {
  std::string fasta, prefix;
  for (int i = 0; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--index" && i + 1 < argc)
      prefix = argv[++i];
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
  return 0;
}

int do_enumerate(int argc, char **argv) {
  std::string prefix, kmers_file, output, format = "csv", mode = "complete";
  std::vector<std::string> alt_pams;
  long long mismatches = 3, max_off = -1, threshold = -1, rna = 0, dna = 0;
  int device = 0;
  size_t batch_size = 1u << 20;
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
    else if (a == "--batch-size") batch_size = (size_t)atoll(need("--batch-size"));
    else if (!a.empty() && a[0] != '-' && prefix.empty()) prefix = a;
    else return usage();
  }
  if (prefix.empty() || kmers_file.empty() || output.empty()) return usage();
  if ((format != "csv" && format != "sam") || (mode != "succinct" && mode != "complete")) return usage();
  const bool bulges = rna > 0 || dna > 0;
  genome_structure gs;
  if (!read_gs(prefix + ".gs", gs)) {
    std::cerr << "error: No genome structure file " << prefix << ".gs\n";
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

  std::vector<kmer_row> kmers;
  std::string err;
  if (!read_kmers(kmers_file, kmers, err)) {
    std::cerr << "error: " << err << "\n";
    return 1;
  }
  std::cout << "Read in " << kmers.size() << " kmer(s).\n";

  auto t0 = std::chrono::steady_clock::now();
  gs_index *ix = nullptr;
  gs_status rc = from_sdsl ? gs_index_open_sdsl(prefix.c_str(), device, &ix)
                           : gs_index_build((const uint8_t *)text.data(), text.size(), device, &ix);
  if (rc != GS_OK) {
    std::cerr << "error: " << gs_status_string(rc) << "\n";
    return 1;
  }
  std::cout << "Built the forward and reverse index on device " << device << " in "
            << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s\n";

  std::ofstream out(output, std::ios::binary);
  if (!out) {
    std::cerr << "error: cannot write " << output << "\n";
    return 1;
  }
  std::vector<const char *> names;
  for (auto &n : gs.names) names.push_back(n.c_str());
  gs_genome_structure cgs{names.data(), gs.lengths.data(), (uint32_t)names.size()};
  const uint32_t tflags = (format == "sam" ? GS_TEXT_SAM : 0u) | (mode == "complete" ? GS_TEXT_COMPLETE : 0u);
  const uint32_t sflags = start ? GS_FLAG_PAM_AT_START : 0u;
  char *txt = nullptr;
  size_t len = 0;
  gs_format_header(&cgs, tflags, &txt, &len);
  out.write(txt, (std::streamsize)len);
  gs_free(txt);

  /* batches of equal (L, P) in input order: the device call takes fixed-width rows */
  t0 = std::chrono::steady_clock::now();
  size_t done = 0;
  while (done < kmers.size()) {
    const size_t L = kmers[done].sequence.size(), P = kmers[done].pam.size();
    size_t end = done;
    std::string seqs, pams, alts;
    while (end < kmers.size() && end - done < batch_size && kmers[end].sequence.size() == L &&
           kmers[end].pam.size() == P) {
      seqs += kmers[end].sequence;
      pams += kmers[end].pam;
      end++;
    }
    uint32_t n_alt = 0;
    if (P > 0)
      for (auto &a : alt_pams) {
        if (a.size() != P) {
          std::cerr << "error: alt PAM " << a << " differs in length from the guides' PAM\n";
          return 1;
        }
        alts += a;
        n_alt++;
      }
    /* --threshold t (process.hpp:66-76): a guide with more than one site within t mismatches
     * (both indexes, bulges off) is dropped before the real search */
    std::vector<char> skip(end - done, 0);
    if (threshold > 0) {
      gs_result *cres = nullptr;
      rc = gs_enumerate(ix, seqs.data(), end - done, (uint32_t)L, pams.data(), (uint32_t)P, alts.data(),
                        n_alt, (uint32_t)threshold, sflags, &cres);
      if (rc != GS_OK) {
        std::cerr << "error: " << gs_status_string(rc) << "\n";
        return 1;
      }
      gs_result_view cv;
      gs_result_get(cres, &cv);
      for (size_t g = 0; g < end - done; g++)
        skip[g] = cv.guide_offsets[g + 1] - cv.guide_offsets[g] > 1;
      gs_result_free(cres);
    }
    gs_result *res = nullptr;
    gs_result_ex *resx = nullptr;
    gs_result_view v;
    memset(&v, 0, sizeof v);
    const uint64_t *xoff = nullptr;
    const gs_hit_ex *xhits = nullptr;
    if (!bulges) {
      rc = gs_enumerate(ix, seqs.data(), end - done, (uint32_t)L, pams.data(), (uint32_t)P, alts.data(),
                        n_alt, (uint32_t)mismatches, sflags, &res);
    } else {
      /* bulge-aware search: index.hpp:250-375 behind gs_enumerate_bulges */
      rc = gs_enumerate_bulges(ix, seqs.data(), end - done, (uint32_t)L, pams.data(), (uint32_t)P,
                               alts.data(), n_alt, (uint32_t)mismatches, (uint32_t)rna, (uint32_t)dna,
                               sflags, &resx);
    }
    if (rc != GS_OK) {
      std::cerr << "error: " << gs_status_string(rc) << "\n";
      return 1;
    }
    /* specificity of every guide of the batch on the device (printer.hpp:98-170, 251-297 behind
     * gs_score): the formatting threads below only print */
    std::vector<float> spec;
    if (!bulges) {
      gs_result_get(res, &v);
      spec.resize(end - done);
      rc = gs_score(ix, seqs.data(), end - done, (uint32_t)L, (uint32_t)P, tflags | sflags, max_off, &cgs,
                    v.guide_offsets, v.hits, nullptr, spec.data());
      if (rc != GS_OK) {
        std::cerr << "error: " << gs_status_string(rc) << "\n";
        return 1;
      }
    } else {
      gs_result_ex_get(resx, nullptr, &xoff, &xhits);
    }
    /* format in parallel over contiguous guide ranges, write in input order (-n 1 order) */
    unsigned nt = fmt_threads ? fmt_threads : std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > end - done) nt = (unsigned)(end - done);
    std::vector<std::string> parts(nt);
    std::vector<gs_status> prc(nt, GS_OK);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++) {
      pool.emplace_back([&, t]() {
        const size_t lo = done + (end - done) * t / nt, hi = done + (end - done) * (t + 1) / nt;
        char *tx = nullptr;
        size_t tl = 0;
        for (size_t g = lo; g < hi; g++) {
          if (skip[g - done]) continue;
          const kmer_row &k = kmers[g];
          gs_kmer ck{k.id.c_str(), k.sequence.c_str(), k.pam.c_str(), k.sense == "+" ? 1 : 0};
          gs_status r;
          if (!bulges) {
            const uint64_t b = v.guide_offsets[g - done], e = v.guide_offsets[g - done + 1];
            r = gs_format_guide_scored(&cgs, &ck, v.hits + b, e - b, (uint32_t)mismatches, tflags | sflags,
                                       max_off, spec[g - done], &tx, &tl);
          } else {
            const uint64_t b = xoff[g - done], e = xoff[g - done + 1];
            r = gs_format_guide_ex(&cgs, &ck, xhits + b, e - b, (uint32_t)mismatches, tflags | sflags,
                                   max_off, &tx, &tl);
          }
          if (r != GS_OK) {
            prc[t] = r;
            return;
          }
          parts[t].append(tx, tl);
          gs_free(tx);
        }
      });
    }
    for (auto &th : pool) th.join();
    for (unsigned t = 0; t < nt; t++) {
      if (prc[t] != GS_OK) {
        std::cerr << "error: " << gs_status_string(prc[t]) << "\n";
        return 1;
      }
      out.write(parts[t].data(), (std::streamsize)parts[t].size());
    }
    if (res) gs_result_free(res);
    if (resx) gs_result_ex_free(resx);
    done = end;
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout << "Processed " << kmers.size() << " kmers in " << secs << " seconds.\n";
  gs_index_close(ix);
  return out ? 0 : 1;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc >= 2 && (!strcmp(argv[1], "--version") || !strcmp(argv[1], "-v"))) {
    std::cout << "2.0.0 (" << gs_version() << ")\n";
    return 0;
  }
  if (argc < 2) return usage();
  if (!strcmp(argv[1], "index")) return do_index(argc - 2, argv + 2);
  if (!strcmp(argv[1], "enumerate")) return do_enumerate(argc - 2, argv + 2);
  return usage();
}
