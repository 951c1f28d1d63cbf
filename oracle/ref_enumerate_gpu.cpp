/*
 * ref_enumerate_gpu.cpp -- the reference's enumerate command with its search replaced by the
 * C-ABI library: the proof that integration/process_gpu.hpp compiles
 * against the reference's own headers and produces the reference's files.  Built by oracle/Makefile
 * into oracle/_ref/gs_ref_enumerate_gpu; TEST INFRASTRUCTURE (the product does not link it).
 *
 * Around the stub this file does what do_enumerate_cmd does (src/guidescan.cxx:181-258): options
 * from argv, <prefix>.gs, the header, kmers grouped into batches of equal (L, P) in input order.
 * The index is opened from the reference's OWN index files through gs_index_open_sdsl.  Same build
 * switches as ref_enumerate.cpp (construction headers skipped; nothing defined in their place).
 */
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <list>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include <sdsl/suffix_arrays.hpp>

#include "process_gpu.hpp"
#include "genomics/seq_io.hpp"

typedef sdsl::wt_huff<> t_wt;
const uint32_t t_sa_dens = 64;
const uint32_t t_isa_dens = 8192;

int main(int argc, char **argv) {
  if (argc < 12) {
    std::cerr << "usage: gs_ref_enumerate_gpu INDEX_PREFIX KMERS OUT csv|sam complete|succinct "
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

  genomics::genome_structure gs;
  if (!genomics::seq_io::load_from_file(gs, opts.index_file_prefix + ".gs")) return 1;
  sdsl::csa_wt<t_wt, t_sa_dens, t_isa_dens> no_csa; /* the printers only read gi.gs */
  genomics::genome_index<t_wt, t_sa_dens, t_isa_dens> gi_forward(no_csa, gs);

  gs_index *gpu = nullptr;
  gs_status rc = gs_index_open_sdsl(opts.index_file_prefix.c_str(), 0, &gpu);
  if (rc != GS_OK) {
    std::cerr << "error: " << gs_status_string(rc) << "\n";
    return 1;
  }

  std::ofstream output(opts.database_file);
  const bool complete = opts.out_mode == "complete";
  if (opts.out_format == "sam")
    genomics::write_sam_header(output, gi_forward.gs);
  else
    genomics::write_csv_header(output, complete);

  genomics::kmers_file_producer kmer_p(opts.kmers_file);
  std::vector<genomics::kmer> batch;
  genomics::kmer k;
  auto flush = [&]() {
    genomics::process_kmers_to_stream_gpu<t_wt, t_sa_dens, t_isa_dens>(gpu, gi_forward, opts, batch, output, complete);
    batch.clear();
  };
  while (kmer_p.get_next_kmer(k)) {
    if (!batch.empty() && (k.sequence.size() != batch[0].sequence.size() || k.pam.size() != batch[0].pam.size()))
      flush();
    batch.push_back(k);
  }
  flush();
  gs_index_close(gpu);
  return output ? 0 : 1;
}
