/*
 * ref_enumerate.cpp -- command-line shim around the REFERENCE'S OWN enumerate
 * pipeline, compiled in place from /root/reference by oracle/Makefile into
 * oracle/_ref/gs_ref_enumerate.  TEST INFRASTRUCTURE ONLY; never linked into or
 * called by the product.
 *
 * Everything that computes is the reference's code, included or compiled where
 * it lies: genomics::genome_index::inexact_search (include/genomics/index.hpp),
 * genomics::process_kmers_to_stream (include/genomics/process.hpp), the CSV/SAM
 * printers (include/genomics/printer.hpp), kmers_file_producer
 * (src/genomics/kmer.cxx), seq_io::load_from_file (src/genomics/seq_io.cxx) and
 * sdsl::csa_wt<wt_huff<>,64,8192> + sdsl::load_from_file.  This file only does
 * what do_enumerate_cmd (src/guidescan.cxx:181-258) does around them -- fill
 * enumerate_cmd_options from argv, load the three index files, write the
 * header, start the worker thread(s) -- because src/guidescan.cxx itself also holds the
 * `index` and `download` commands, which need sdsl::construct (divsufsort.h,
 * cmake-generated) and libcurl headers.
 *
 * How the headers compile without the generated divsufsort.h: oracle/Makefile
 * passes -DINCLUDED_SDSL_CONSTRUCT_SA -DINCLUDED_SDSL_CONSTRUCT (the include
 * guards of sdsl/construct_sa.hpp and sdsl/construct.hpp, so the two
 * construction-side headers are skipped) and -fpermissive (two never-instantiated
 * members of csa_sampling_strategy.hpp name construct()/construct_isa()).
 * Nothing is declared or defined here in their place; the query side
 * (load, rank_bwt, C, operator[]) is untouched.
 */
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <list>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <sdsl/suffix_arrays.hpp>

#include "genomics/index.hpp"
#include "genomics/kmer.hpp"
#include "genomics/printer.hpp"
#include "genomics/process.hpp"
#include "genomics/seq_io.hpp"
#include "genomics/structures.hpp"
#include "guidescan.hpp"

/* the reference's index type (src/guidescan.cxx:24-27) */
typedef sdsl::wt_huff<> t_wt;
const uint32_t t_sa_dens = 64;
const uint32_t t_isa_dens = 8192;

int main(int argc, char **argv) {
  if (argc < 12) {
    std::cerr << "usage: gs_ref_enumerate INDEX_PREFIX KMERS OUT csv|sam complete|succinct "
                 "MISMATCHES RNA_BULGES DNA_BULGES THRESHOLD MAX_OFF_TARGETS START [ALT_PAM...]\n";
    return 2;
  }
  enumerate_cmd_options opts;
  opts.index_file_prefix = argv[1];
  opts.kmers_file = argv[2];
  opts.database_file = argv[3];
  opts.out_format = argv[4];
  opts.out_mode = argv[5];
  opts.mismatches = std::strtoul(argv[6], nullptr, 10);
  opts.rna_bulges = std::strtoul(argv[7], nullptr, 10);
  opts.dna_bulges = std::strtoul(argv[8], nullptr, 10);
  opts.threshold = std::atoi(argv[9]);
  opts.max_off_targets = std::atoll(argv[10]);
  opts.start = std::atoi(argv[11]) != 0;
  opts.nthreads = 1;
  for (int i = 12; i < argc; i++) opts.alt_pams.push_back(argv[i]);
  spdlog::set_level(spdlog::level::off);

  genomics::genome_structure gs;
  if (!genomics::seq_io::load_from_file(gs, opts.index_file_prefix + ".gs")) return 1;
  sdsl::csa_wt<t_wt, t_sa_dens, t_isa_dens> forward_fm_index, reverse_fm_index;
  if (!sdsl::load_from_file(forward_fm_index, opts.index_file_prefix + ".forward")) return 1;
  if (!sdsl::load_from_file(reverse_fm_index, opts.index_file_prefix + ".reverse")) return 1;
  genomics::genome_index<t_wt, t_sa_dens, t_isa_dens> gi_forward(forward_fm_index, gs);
  genomics::genome_index<t_wt, t_sa_dens, t_isa_dens> gi_reverse(reverse_fm_index, gs);

  std::ofstream output(opts.database_file);
  const bool complete = opts.out_mode == "complete";
  if (opts.out_format == "sam")
    genomics::write_sam_header(output, gi_forward.gs);
  else
    genomics::write_csv_header(output, complete);

  /* kmers dealt round-robin to the threads (src/guidescan.cxx:226-231); one thread unless
   * GS_REF_THREADS says otherwise (bench.py's cpu_baseline leg; row order then varies) */
  if (const char *e = std::getenv("GS_REF_THREADS")) opts.nthreads = std::max(1, std::atoi(e));
  genomics::kmers_file_producer kmer_p(opts.kmers_file);
  std::vector<std::vector<genomics::kmer>> kmers(opts.nthreads);
  genomics::kmer k;
  size_t kmer_count = 0;
  for (; kmer_p.get_next_kmer(k); kmer_count++) kmers[kmer_count % opts.nthreads].push_back(k);

  std::mutex output_mtx;
  std::atomic<uint64_t> done(0);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> threads;
  for (size_t i = 0; i < opts.nthreads; i++)
    threads.emplace_back(genomics::process_kmers_to_stream<t_wt, t_sa_dens, t_isa_dens>, std::cref(gi_forward),
                         std::cref(gi_reverse), std::cref(opts), std::cref(kmers[i]), std::ref(output),
                         std::ref(output_mtx), std::ref(done), kmer_count, t0, complete);
  for (auto &t : threads) t.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cerr << "gs_ref_enumerate: " << kmer_count << " kmers in " << secs << " s on " << opts.nthreads
            << " thread(s)\n";
  return 0;
}
