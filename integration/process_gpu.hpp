/*
 * process_gpu.hpp -- the reference-side binding of libgsamd.so: what a
 * maintainer of guidescan-cli adds as include/genomics/process_gpu.hpp.  A batch variant of
 * process_kmers_to_stream (include/genomics/process.hpp:35-158) that keeps the reference's own
 * printers untouched: the search, the set ordering and the resolve() expansion run on the GPU
 * behind the C-ABI (include/guidescan_amd.h), everything from process.hpp:117 on is the
 * reference's code.
 *
 * This file is compiled - against the reference's headers where they lie, with the oracle/Makefile
 * recipe - into oracle/_ref/gs_ref_enumerate_gpu (shim: oracle/ref_enumerate_gpu.cpp) and run by
 * tests/test_integration_stub_gpu.py against oracle/_ref/gs_ref_enumerate.
 */
#ifndef GENOMICS_PROCESS_GPU_H
#define GENOMICS_PROCESS_GPU_H

#include <ostream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "guidescan_amd.h"

#include "genomics/index.hpp"
#include "genomics/kmer.hpp"
#include "genomics/printer.hpp"
#include "genomics/structures.hpp"
#include "guidescan.hpp"

namespace genomics {
  /* gi_forward is only used for its genome_structure (the printers read gi.gs); the search runs on
   * the GPU handle.  One call per batch of kmers with equal sequence and PAM lengths. */
  template <class t_wt, uint32_t t_dens, uint32_t t_inv_dens>
  void process_kmers_to_stream_gpu(gs_index* gpu,
                                   const genome_index<t_wt, t_dens, t_inv_dens>& gi_forward,
                                   const enumerate_cmd_options& opts,
                                   const std::vector<kmer>& kmers,
                                   std::ostream& output, bool complete) {
    if (kmers.empty()) return;
    const uint32_t L = kmers[0].sequence.size(), P = kmers[0].pam.size();
    const uint32_t flags = opts.start ? GS_FLAG_PAM_AT_START : 0;
    std::string seqs, pams, alts;
    for (const auto& k : kmers) { seqs += k.sequence; pams += k.pam; }
    for (const auto& a : opts.alt_pams) alts += a;
    const uint32_t n_alt = P ? opts.alt_pams.size() : 0;   // process.hpp:51-56: no PAM, no alt PAMs

    // --threshold (process.hpp:66-76): a counting call of the same entry point at t mismatches
    std::vector<char> skip(kmers.size(), 0);
    if (opts.threshold > 0) {
      gs_result* cres = nullptr;
      gs_status rc = gs_enumerate(gpu, seqs.data(), kmers.size(), L, pams.data(), P, alts.data(), n_alt,
                                  opts.threshold, flags | GS_FLAG_RAW_COUNTS, &cres);
      if (rc != GS_OK) throw std::runtime_error(std::string("threshold pass: ") + gs_status_string(rc));
      gs_result_view cv; gs_result_get(cres, &cv);
      // off_target_counter (process.hpp:25-27) counts per PAM pattern, before the sets drop duplicates
      for (size_t g = 0; g < kmers.size(); g++) skip[g] = cv.raw_hits[g] > 1;
      gs_result_free(cres);
    }

    gs_result* res = nullptr;
    gs_status rc = gs_enumerate(gpu, seqs.data(), kmers.size(), L, pams.data(), P, alts.data(), n_alt,
                                opts.mismatches, flags, &res);
    if (rc != GS_OK) throw std::runtime_error(gs_status_string(rc));
    gs_result_view v; gs_result_get(res, &v);
    std::vector<char> buf(L + P + 1);
    for (size_t g = 0; g < kmers.size(); g++) {
      if (skip[g]) continue;
      // the container the reference's printers take (process.hpp:100); hits arrive in its order
      std::vector<std::vector<std::tuple<int64_t, match>>> off_targets(opts.mismatches + 1);
      for (uint64_t h = v.guide_offsets[g]; h < v.guide_offsets[g + 1]; h++) {
        const gs_hit& hit = v.hits[h];
        gs_decode_sequence(kmers[g].sequence.data(), L, P, flags, hit.key, buf.data());
        match m = {std::string(buf.data()), 0, 0, GS_KEY_MISMATCHES(hit.key), 0, 0};
        off_targets[m.mismatches].push_back(std::make_tuple(hit.pos, m));
      }
      output << (opts.out_format == "csv"
                   ? get_csv_lines(gi_forward, kmers[g], opts.start, opts.max_off_targets, off_targets, complete)
                   : get_sam_lines(gi_forward, kmers[g], opts.start, opts.max_off_targets, off_targets, complete));
    }
    gs_result_free(res);
  }
}

#endif /* GENOMICS_PROCESS_GPU_H */
