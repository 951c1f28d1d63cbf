/*
 * process_gpu.hpp -- the reference-side binding of libgsamd.so: what a
 * maintainer of guidescan-cli adds as include/genomics/process_gpu.hpp.  A batch variant of
 * process_kmers_to_stream (include/genomics/process.hpp:35-158) that keeps the reference's own
 * printers untouched: the search, the set ordering and the resolve() expansion run on the GPU
 * behind the C-ABI (include/guidescan_amd.h), everything from process.hpp:117 on is the
 * reference's code.
 *
 * Every input the reference's command accepts is served:
 *   - plain batches (A,C,G,T guides, PAM patterns over A,C,G,T,N, no bulges): gs_enumerate, the fast path;
 *   - guides the fast path flags (a symbol outside A,C,G,T in the guide or its own PAM: matched literally,
 *     index.hpp:125-170, 218-247) are enumerated again, alone, by gs_enumerate_general and take their hits from
 *     there - the fast path leaves them an empty list and GS_GUIDE_NEEDS_GENERAL in gs_result_view.guide_flags;
 *   - --rna-bulges / --dna-bulges (index.hpp:250-375): the whole batch through gs_enumerate_bulges;
 *   - alt PAMs whose length differs from the guides' PAM (process.hpp:51-56 takes any): the whole batch
 *     through gs_enumerate_general_pams;
 *   - --threshold (process.hpp:66-76): a counting call of the same entry points at t mismatches, no bulges.
 *
 * This file is compiled - against the reference's headers where they lie, with the oracle/Makefile
 * recipe - into oracle/_ref/gs_ref_enumerate_gpu (shim: oracle/ref_enumerate_gpu.cpp) and run by
 * tests/test_integration_stub_gpu.py against oracle/_ref/gs_ref_enumerate.
 */
#ifndef GENOMICS_PROCESS_GPU_H
#define GENOMICS_PROCESS_GPU_H

#include <ostream>
#include <stdexcept>
#include <memory>
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
  namespace gpu_detail {
    typedef std::vector<std::vector<std::tuple<int64_t, match>>> off_target_lists; /* process.hpp:100 */

    inline void check(gs_status rc, const char* what) {
      if (rc != GS_OK) throw std::runtime_error(std::string(what) + ": " + gs_status_string(rc));
    }

    /* the general path for `which` guides of the batch (all of them when which is empty): hit lists in the
     * reference's container, and - raw != nullptr - the hits counted before duplicate sequences collapse */
    inline void enumerate_general(gs_index* gpu, const std::vector<kmer>& kmers, const std::vector<size_t>& which,
                                  uint32_t L, uint32_t P, const enumerate_cmd_options& opts, uint32_t mismatches,
                                  uint32_t rna, uint32_t dna, uint32_t flags,
                                  std::vector<off_target_lists>* lists, std::vector<uint32_t>* raw) {
      std::vector<size_t> idx = which;
      if (idx.empty())
        for (size_t g = 0; g < kmers.size(); g++) idx.push_back(g);
      std::string seqs, pams, alts;
      std::vector<uint32_t> alt_lens;
      for (size_t g : idx) { seqs += kmers[g].sequence; pams += kmers[g].pam; }
      if (P)
        for (const auto& a : opts.alt_pams) { alts += a; alt_lens.push_back(a.size()); }
      gs_result_ex* res = nullptr;
      check(gs_enumerate_general_pams(gpu, seqs.data(), idx.size(), L, pams.data(), P, alts.data(), alt_lens.data(),
                                      alt_lens.size(), mismatches, rna, dna, flags, &res), "general path");
      std::unique_ptr<gs_result_ex, void (*)(gs_result_ex*)> hold(res, gs_result_ex_free); /* (freed on a throw below, too) */
      uint64_t n = 0;
      const uint64_t* off = nullptr;
      const gs_hit_ex* hits = nullptr;
      gs_result_ex_get(res, &n, &off, &hits);
      if (raw) {
        const uint32_t* rh = nullptr;
        check(gs_result_ex_raw_hits(res, &rh), "raw hit counts");
        for (size_t j = 0; j < idx.size(); j++) (*raw)[idx[j]] = rh[j];
      }
      if (lists) {
        char buf[33];
        for (size_t j = 0; j < idx.size(); j++) {
          off_target_lists& l = (*lists)[idx[j]];
          l.assign(mismatches + 1, {});
          for (uint64_t h = off[j]; h < off[j + 1]; h++) {
            gs_decode_sequence_ex(&hits[h], buf);
            match m = {std::string(buf), 0, 0, hits[h].mismatches, hits[h].dna_bulges, hits[h].rna_bulges};
            l[m.mismatches].push_back(std::make_tuple(hits[h].pos, m));
          }
        }
      }
    }

    /* the fast path for the whole batch; the guides it flags are listed in `flagged` and keep empty lists.  false: the
     * fast path does not take this batch shape (GS_ERR_UNSUPPORTED: e.g. a match sequence beyond 52 key bits on an index
     * whose table is too shallow for it) - the general path does */
    inline bool enumerate_fast(gs_index* gpu, const std::vector<kmer>& kmers, uint32_t L, uint32_t P,
                               const enumerate_cmd_options& opts, uint32_t mismatches, uint32_t flags,
                               std::vector<off_target_lists>* lists, std::vector<uint32_t>* raw,
                               std::vector<size_t>& flagged) {
      std::string seqs, pams, alts;
      for (const auto& k : kmers) { seqs += k.sequence; pams += k.pam; }
      for (const auto& a : opts.alt_pams) alts += a;
      const uint32_t n_alt = P ? opts.alt_pams.size() : 0;   // process.hpp:51-56: no PAM, no alt PAMs
      gs_result* res = nullptr;
      const gs_status rc = gs_enumerate(gpu, seqs.data(), kmers.size(), L, pams.data(), P, alts.data(), n_alt, mismatches,
                                        flags | (raw ? GS_FLAG_RAW_COUNTS : 0), &res);
      if (rc == GS_ERR_UNSUPPORTED) return false;
      check(rc, "fast path");
      std::unique_ptr<gs_result, void (*)(gs_result*)> hold(res, gs_result_free);
      gs_result_view v;
      gs_result_get(res, &v);
      std::vector<char> buf(L + P + 1);
      for (size_t g = 0; g < kmers.size(); g++) {
        if (v.n_unsupported && v.guide_flags && (v.guide_flags[g] & GS_GUIDE_NEEDS_GENERAL)) {
          flagged.push_back(g);
          continue;
        }
        if (raw) (*raw)[g] = v.raw_hits[g];
        if (!lists) continue;
        off_target_lists& l = (*lists)[g];
        l.assign(mismatches + 1, {});
        for (uint64_t h = v.guide_offsets[g]; h < v.guide_offsets[g + 1]; h++) {
          const gs_hit& hit = v.hits[h];
          gs_decode_sequence(kmers[g].sequence.data(), L, P, flags, hit.key, buf.data());
          match m = {std::string(buf.data()), 0, 0, GS_KEY_MISMATCHES(hit.key), 0, 0};
          l[m.mismatches].push_back(std::make_tuple(hit.pos, m));
        }
      }
      return true;
    }

    /* one search of the batch: fast path + the general path for what it flags, or the general path for all */
    inline void enumerate(gs_index* gpu, const std::vector<kmer>& kmers, uint32_t L, uint32_t P,
                          const enumerate_cmd_options& opts, uint32_t mismatches, uint32_t rna, uint32_t dna,
                          uint32_t flags, std::vector<off_target_lists>* lists, std::vector<uint32_t>* raw) {
      /* the fast path carries match sequences of up to 59 key bits (23-mers with a four-symbol PAM: 58); where a batch
       * needs what only 52 bits allow (a shallow table, the device-wide ordering) it says GS_ERR_UNSUPPORTED and the
       * general path - 100 x slower - takes the batch, as in host/guidescan_main.cpp */
      bool general = rna != 0 || dna != 0 || 2 * L + 3 * P > 59;
      if (P)
        for (const auto& a : opts.alt_pams) general = general || a.size() != P;
      std::vector<size_t> flagged;
      if (!general && !enumerate_fast(gpu, kmers, L, P, opts, mismatches, flags, lists, raw, flagged)) general = true;
      if (general) {
        enumerate_general(gpu, kmers, {}, L, P, opts, mismatches, rna, dna, flags, lists, raw);
        return;
      }
      if (!flagged.empty()) enumerate_general(gpu, kmers, flagged, L, P, opts, mismatches, 0, 0, flags, lists, raw);
    }
  }

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

    // --threshold (process.hpp:66-76): a counting search at t mismatches, no bulges; off_target_counter
    // (process.hpp:25-27) counts per PAM pattern, before the sets drop duplicates
    std::vector<char> skip(kmers.size(), 0);
    if (opts.threshold > 0) {
      std::vector<uint32_t> raw(kmers.size(), 0);
      gpu_detail::enumerate(gpu, kmers, L, P, opts, opts.threshold, 0, 0, flags, nullptr, &raw);
      for (size_t g = 0; g < kmers.size(); g++) skip[g] = raw[g] > 1;
    }

    std::vector<gpu_detail::off_target_lists> lists(kmers.size());
    gpu_detail::enumerate(gpu, kmers, L, P, opts, opts.mismatches, opts.rna_bulges, opts.dna_bulges, flags, &lists,
                          nullptr);
    for (size_t g = 0; g < kmers.size(); g++) {
      if (skip[g]) continue;
      // the container the reference's printers take (process.hpp:100); hits arrive in its order
      output << (opts.out_format == "csv"
                   ? get_csv_lines(gi_forward, kmers[g], opts.start, opts.max_off_targets, lists[g], complete)
                   : get_sam_lines(gi_forward, kmers[g], opts.start, opts.max_off_targets, lists[g], complete));
    }
  }
}

#endif /* GENOMICS_PROCESS_GPU_H */
