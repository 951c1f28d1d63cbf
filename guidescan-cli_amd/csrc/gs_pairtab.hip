/*
 * gs_pairtab.hip -- PAM-pair tables: the depth-k prefix table of a strand restricted to the rows that
 * can host a site of a PAM ending in a given pair of bases.
 *
 * This strand's seeds of k_search (gs_search.hip) are depth-k intervals of all rows whose suffix starts
 * with a variant of the guide's first k consumed symbols; what decides a row is the v_rem symbols to
 * its left: the rest of the guide, then the PAM (index.hpp:182-248 walk the same symbols one by
 * one).  The last two consumed PAM symbols of NGG, NAG, TTTN (--start) ... are concrete bases, so
 * only the rows with exactly that pair at context offsets v_rem-2, v_rem-1 can hold a site: one row in
 * sixteen.  Collecting those rows per k-mer gives a table of the same shape whose intervals hold
 * 0.7 rows instead of 11.5 at hg38 size: half the entries are empty (the seed dies with the table
 * read), a third hold one row whose context symbols sit in the entry itself (no second read), the
 * rest are filtered by per-position symbol sets over their few rows.  Entries are 8 bytes
 * (gs_common.h, gs_pairtab_dev): the 16 two-symbol extensions of a variant are one 128-byte block.  Rows with a symbol outside A,C,G,T among
 * the nearest v_rem are left out: k_search reports those sites from the literal-N window list.
 * Derived data, built on the device from the strand's table and ctx[] on first use (~0.1 s at hg38
 * size), 4.3 GB per table copy + 10 bytes per selected row.
 */
#include "gs_device.h"

#include <algorithm>
#include <rocprim/rocprim.hpp>

struct pt_args {
  const uint4 *tab;
  const uint32_t *ctx;
  const uint32_t *exc_row;
  const uint64_t *exc_sym;
  uint32_t n_exc;
  uint32_t v_rem, code, mask_off;
  uint64_t entries;
  /* outputs */
  uint32_t *count;       /* per k-mer: selected rows */
  const uint32_t *start; /* exclusive sums of count[] */
  uint2 *out;
  uint16_t *c16;
  uint32_t *octx, *rowid;
};

/* does row r (context word w) belong to the pair's table? */
__device__ __forceinline__ bool pt_selected(const pt_args &a, uint32_t r, uint32_t w, bool flagged) {
  if (((w >> (2u * (a.v_rem - 2u))) & 15u) != a.code) return false;
  if (flagged) { /* the k-mer has exception rows: is this one, and is the symbol near enough to matter? */
    uint32_t lo = 0, hi = a.n_exc;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (a.exc_row[mid] < r)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < a.n_exc && a.exc_row[lo] == r) {
      const uint64_t nb = a.exc_sym[lo];
      for (uint32_t j = 0; j < a.v_rem; j++)
        if (((nb >> (4u * j)) & 15u) > 3u) return false;
    }
  }
  return true;
}

__global__ void k_pt_count(pt_args a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.entries) return;
  const uint4 e = a.tab[i];
  const uint32_t cnt = e.y & 0x7FFFFFFFu;
  const bool flagged = (e.y >> 31) != 0u;
  uint32_t c = 0;
  for (uint32_t j = 0; j < cnt; j++) c += pt_selected(a, e.x + j, a.ctx[e.x + j], flagged) ? 1u : 0u;
  a.count[i] = c + (c >= GS_PT_BIG ? 1u : 0u); /* slots in the row arrays: a header slot in front of 63 rows and more */
}

__global__ void k_pt_fill(pt_args a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.entries) return;
  const uint32_t slots = a.count[i];
  uint32_t at = a.start[i];
  uint2 o = make_uint2(at, 0u);
  if (slots) {
    const uint4 e = a.tab[i];
    const uint32_t cnt = e.y & 0x7FFFFFFFu;
    const bool flagged = (e.y >> 31) != 0u;
    const bool big = slots > GS_PT_BIG; /* 63 rows and more: header slot + rows */
    const uint32_t mine = big ? slots - 1u : slots;
    if (big) {
      a.c16[at] = 0;
      a.octx[at] = 0;
      a.rowid[at] = mine;
      at++;
    }
    uint32_t c = 0, sets = 0, lw = 0;
    for (uint32_t j = 0; j < cnt && c < mine; j++) {
      const uint32_t r = e.x + j, w = a.ctx[r];
      if (!pt_selected(a, r, w, flagged)) continue;
      a.c16[at + c] = (uint16_t)w;
      a.octx[at + c] = w;
      a.rowid[at + c] = r;
      for (uint32_t p = 0; p < 6u; p++) sets |= 1u << (4u * p + ((w >> (2u * p)) & 3u));
      lw = w;
      c++;
    }
    o.y = (big ? GS_PT_BIG : mine) | ((mine == 1u ? (lw & 0x3FFFFFFu) : sets) << 6);
  }
  a.out[i] = o;
}

/* rot[slot][perm_p(i)] = tab[i]: the field of consumption step p moves to bits 1:0 (gs_index.hip k_rot_copy) */
__global__ void k_pt_rot(const uint2 *tab, uint2 *rot, uint32_t k, uint32_t p, uint32_t slot) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >> (2 * k)) return;
  const uint32_t sh = 2u * (k - 1u - p);
  const uint64_t hi = i >> (sh + 2u), lo = i & ((1ull << sh) - 1ull), f = (i >> sh) & 3ull;
  rot[((uint64_t)slot << (2 * k)) + ((hi << (sh + 2u)) | (lo << 2) | f)] = tab[i];
}

/* ---- the other strand's side: tables deeper by the PAM's free symbol ----------------------------------
 * Under two-sided seeding the sites with many substitutions among a guide's first consumed symbols
 * come from the OTHER strand's table, whose backward search consumes the PAM first: k-mer = P PAM
 * symbols + k-P guide symbols, one lookup per base a PAM 'N' can stand for.  For a PAM of three symbols
 * that ends (as this strand consumes it) in the pair `code`, the deep table is indexed by k-2 guide
 * symbols and, in the lowest two bits, the base under the N: entry = the interval of the rows that
 * start with those k+1 symbols - a quarter of the rows of a depth-k interval - in the strand table's
 * own format {first row, rows | flag << 31, pair masks over these rows (context offsets 0, 2, 4, 6)}.
 * The four entries of one (k-2)-mer are one 64-byte line; with 2.9 rows behind a mask instead of
 * 11.5 few seeds survive it.  The rows are rows of the strand's own suffix array: verification reads
 * the strand's ctx16[] / ctx[] as before.  4^(k-2) x 64 bytes: 1.07 GB per strand at k = 14. */
struct pb_args {
  gs_strand_dev sd;
  uint32_t k, P, code; /* code as this strand's consumption sees the pair: first | second << 2 */
  uint32_t kb;         /* guide symbols of the index: k-P .. 14 */
  uint64_t entries;    /* 4^kb */
  uint4 *out;          /* 4 per entry */
};

__global__ void k_pb_build(pb_args a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.entries) return;
  const uint32_t k = a.k, P = a.P, kb = a.kb;
  /* the first k consumed symbols: the pair (second consumed PAM symbol of this strand first), the base
   * under the N, then the k-P leading symbols of the entry */
  const uint32_t c0 = a.code & 3u, c1 = (a.code >> 2) & 3u;
  uint32_t head = ((3u - c1) << (2u * (k - 1u))) | ((3u - c0) << (2u * (k - 2u)));
  for (uint32_t y = 0; y < k - P; y++) head |= (uint32_t)((i >> (2u * (kb - 1u - y))) & 3u) << (2u * (k - 1u - P - y));
  for (uint32_t x = 0; x < 4u; x++) {
    const uint4 e = a.sd.ptab[head | (x << (2u * (k - 3u)))];
    uint32_t lo = e.x, cnt = e.y & 0x7FFFFFFFu;
    bool blind = (e.y >> 31) != 0u; /* exception rows somewhere in the depth-k interval: verify, do not filter */
    for (uint32_t y = k - P; y < kb && cnt; y++) { /* the symbols the depth-k table does not reach */
      const uint32_t c = (uint32_t)((i >> (2u * (kb - 1u - y))) & 3u);
      const uint32_t hi = lo + cnt - 1u;
      const uint32_t oa = occ1(a.sd.blocks, lo >> GS_BLOCK_SHIFT, lo & (GS_BLOCK_ROWS - 1u), c);
      const uint32_t ob = occ1(a.sd.blocks, hi >> GS_BLOCK_SHIFT, (hi & (GS_BLOCK_ROWS - 1u)) + 1u, c);
      lo = a.sd.C[c] + oa;
      cnt = ob - oa;
    }
    if (cnt > 4096u) blind = true; /* too many rows to scan here */
    uint32_t mz = 0, mw = 0;
    if (cnt && !blind)
      for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t w = a.sd.ctx[lo + j];
        mz |= (1u << (w & 15u)) | (1u << (16u + ((w >> 4) & 15u)));
        mw |= (1u << ((w >> 8) & 15u)) | (1u << (16u + ((w >> 12) & 15u)));
      }
    a.out[4 * i + x] = make_uint4(lo, cnt | (blind && cnt ? 0x80000000u : 0u), mz, mw);
  }
}

void gs_pairtab_free(gs_index *ix, uint32_t slot) {
  gs_pairtab_host &p = ix->pairtab[slot];
  for (int s = 0; s < 2; s++)
    for (int j = 0; j < 8; j++) {
      if (p.mem[s][j]) hipFree(p.mem[s][j]);
      p.mem[s][j] = nullptr;
    }
  p.valid = false;
  p.deep = false;
  p.d[0] = gs_pairtab_dev{};
  p.d[1] = gs_pairtab_dev{};
  p.bytes = 0;
}

static gs_status build_one(gs_index *ix, gs_pairtab_host &p, int s, uint32_t k, uint32_t rot_from, hipStream_t st) {
  const gs_strand &S = ix->strand[s];
  const uint64_t entries = 1ull << (2 * k);
  pt_args a;
  memset(&a, 0, sizeof(a));
  a.tab = (const uint4 *)S.ptab;
  a.ctx = (const uint32_t *)S.ctx;
  a.exc_row = S.d.exc_row;
  a.exc_sym = S.d.exc_sym;
  a.n_exc = S.d.n_exc;
  a.v_rem = p.v_rem;
  a.code = p.code;
  a.mask_off = S.d.mask_off;
  a.entries = entries;
  uint32_t *d_count = nullptr, *d_start = nullptr;
  void *d_tmp = nullptr;
  struct cleanup_t {
    uint32_t *&c, *&s;
    void *&t;
    ~cleanup_t() {
      if (c) hipFree(c);
      if (s) hipFree(s);
      if (t) hipFree(t);
    }
  } cleanup{d_count, d_start, d_tmp};
  if (hipMalloc(&d_count, 4 * entries) != hipSuccess || hipMalloc(&d_start, 4 * entries) != hipSuccess) {
    (void)hipGetLastError();
    return GS_ERR_NOMEM;
  }
  a.count = d_count;
  const uint32_t nb = (uint32_t)((entries + 255) / 256);
  hipLaunchKernelGGL(k_pt_count, dim3(nb), dim3(256), 0, st, a);
  size_t tmp_bytes = 0;
  GS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_count, d_start, 0u, entries, rocprim::plus<uint32_t>(), st));
  if (hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16) != hipSuccess) {
    (void)hipGetLastError();
    return GS_ERR_NOMEM;
  }
  GS_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_count, d_start, 0u, entries, rocprim::plus<uint32_t>(), st));
  uint32_t last_start = 0, last_count = 0;
  GS_HIP(hipMemcpyAsync(&last_start, d_start + entries - 1, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipMemcpyAsync(&last_count, d_count + entries - 1, 4, hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  const uint64_t rows = (uint64_t)last_start + last_count;
  void **m = p.mem[s];
  const uint32_t nrot = rot_from + 2 < k ? k - 2 - rot_from : 0; /* steps rot_from .. k-3 (step k-2's variants sit in the plain table's 128-byte blocks) */
  if (hipMalloc(&m[0], sizeof(uint2) * entries) != hipSuccess || hipMalloc(&m[2], 2 * rows + 64) != hipSuccess ||
      hipMalloc(&m[3], 4 * rows + 16) != hipSuccess || hipMalloc(&m[4], 4 * rows + 16) != hipSuccess ||
      (nrot && hipMalloc(&m[1], sizeof(uint2) * entries * nrot) != hipSuccess)) {
    (void)hipGetLastError();
    return GS_ERR_NOMEM;
  }
  GS_HIP(hipMemsetAsync(m[2], 0, 2 * rows + 64, st)); /* k_search reads whole groups of eight */
  a.start = d_start;
  a.out = (uint2 *)m[0];
  a.c16 = (uint16_t *)m[2];
  a.octx = (uint32_t *)m[3];
  a.rowid = (uint32_t *)m[4];
  hipLaunchKernelGGL(k_pt_fill, dim3(nb), dim3(256), 0, st, a);
  for (uint32_t j = 0; j < nrot; j++)
    hipLaunchKernelGGL(k_pt_rot, dim3(nb), dim3(256), 0, st, (const uint2 *)m[0], (uint2 *)m[1], k, rot_from + j, j);
  GS_HIP(hipStreamSynchronize(st));
  GS_HIP(hipGetLastError());
  gs_pairtab_dev &d = p.d[s];
  d.tab = (const uint2 *)m[0];
  d.rot = (const uint2 *)m[1];
  d.c16 = (const uint16_t *)m[2];
  d.ctx = (const uint32_t *)m[3];
  d.rowid = (const uint32_t *)m[4];
  d.rot_first = nrot ? rot_from : 31u;
  d.code = p.code;
  p.bytes += sizeof(uint2) * entries * (1 + nrot) + 10 * rows;
  if (gs_opt(ix, "GS_DEBUG"))
    fprintf(stderr, "[gs] PAM-pair table: strand %d, pair %u at context depth %u: %llu of %llu rows, %u rotated copies, %.2f GB\n",
            s, p.code, p.v_rem, (unsigned long long)rows, (unsigned long long)S.n, nrot,
            1e-9 * (double)(sizeof(uint2) * entries * (1 + nrot) + 10 * rows));
  return GS_OK;
}

/* the deep table of strand s for the pair (built after the pair table proper; optional) */
static gs_status build_deep(gs_index *ix, gs_pairtab_host &p, int s, uint32_t k, uint32_t P, uint32_t kb, hipStream_t st) {
  const gs_strand &S = ix->strand[s];
  const uint64_t entries = 1ull << (2 * kb);
  void **m = p.mem[s];
  if (hipMalloc(&m[5], 64 * entries) != hipSuccess) {
    (void)hipGetLastError();
    return GS_ERR_NOMEM;
  }
  pb_args a;
  memset(&a, 0, sizeof(a));
  a.sd = S.d;
  a.k = k;
  a.P = P;
  a.kb = kb;
  a.code = p.code;
  a.entries = entries;
  a.out = (uint4 *)m[5];
  hipLaunchKernelGGL(k_pb_build, dim3((uint32_t)((entries + 255) / 256)), dim3(256), 0, st, a);
  GS_HIP(hipStreamSynchronize(st));
  GS_HIP(hipGetLastError());
  p.d[s].deep = (const uint4 *)m[5];
  p.bytes += 64 * entries;
  if (gs_opt(ix, "GS_DEBUG"))
    fprintf(stderr, "[gs] deep table: strand %d, pair %u: %llu lines of four entries, %.2f GB\n", s, p.code,
            (unsigned long long)entries, 1e-9 * (double)(64 * entries));
  return GS_OK;
}

gs_status gs_pairtab_ensure(gs_index *ix, uint32_t slot, uint32_t v_rem, uint32_t code, uint32_t rot_first, double share,
                            hipStream_t st) {
  gs_pairtab_host &p = ix->pairtab[slot];
  const uint32_t k = ix->pt_k;
  if (rot_first > 31) rot_first = 31;
  const uint32_t rot_wanted = rot_first;
  if (p.valid && p.v_rem == v_rem && p.code == code && p.rot_first <= rot_wanted) return GS_OK;
  gs_pairtab_free(ix, slot);
  if (!k || v_rem < 2 || v_rem > 16 || !ix->strand[0].ctx || !ix->strand[1].ctx) return GS_OK;
  /* fit into what is free, keeping room for the batch workspace: drop rotated copies first */
  size_t free_b = 0, total_b = 0;
  GS_HIP(hipMemGetInfo(&free_b, &total_b));
  const double entry_bytes = 8.0 * (double)(1ull << (2 * k));
  double reserve = 64e9; /* slots, sort buffers and hits of a large batch at a high budget */
  if (const char *e = gs_opt(ix, "GS_PAIRTAB_RESERVE_GB")) reserve = atof(e) * 1e9;
  if (reserve > 0.25 * (double)total_b) reserve = 0.25 * (double)total_b;
  const double rows_bytes = 10.0 * ((double)ix->strand[0].n + (double)ix->strand[1].n) / 16.0 * 1.5;
  const double tmp_bytes = 8.0 * (double)(1ull << (2 * k)) + 64e6;
  double room = (double)free_b - reserve;
  if (const char *e = gs_opt(ix, "GS_INDEX_BUDGET_GB")) { /* the cap on the whole index covers its derived tables too */
    const double left = atof(e) * 1e9 - (double)gs_index_device_bytes(ix);
    if (left < room) room = left;
  }
  for (;;) {
    const uint32_t nrot = rot_first + 2 < k ? k - 2 - rot_first : 0;
    const double need = 2.0 * entry_bytes * (1 + nrot) + rows_bytes + tmp_bytes;
    if (need <= room * share) break; /* share < 1: another pair's tables are still to come */
    if (nrot == 0) {
      if (gs_opt(ix, "GS_DEBUG")) fprintf(stderr, "[gs] PAM-pair table %u: not enough free memory (%.1f GB), skipped\n", code, 1e-9 * (double)free_b);
      return GS_OK;
    }
    rot_first = nrot == 1 ? 31 : rot_first + 1;
  }
  p.v_rem = v_rem;
  p.code = code;
  p.rot_first = rot_wanted;
  for (int s = 0; s < 2; s++) {
    const gs_status rc = build_one(ix, p, s, k, rot_first, st);
    if (rc == GS_ERR_NOMEM) { /* the estimate was off: go on without this table */
      gs_pairtab_free(ix, slot);
      return GS_OK;
    }
    if (rc != GS_OK) {
      gs_pairtab_free(ix, slot);
      return rc;
    }
  }
  p.valid = true;
  return GS_OK;
}

gs_status gs_pairtab_ensure_deep(gs_index *ix, uint32_t slot, uint32_t P, uint32_t kb, hipStream_t st) {
  gs_pairtab_host &p = ix->pairtab[slot];
  const uint32_t k = ix->pt_k;
  if (!p.valid || P != 3 || k < 6 || kb + P < k || kb > 14) return GS_OK;
  if (p.deep && p.deep_P == P && p.deep_kb == kb) return GS_OK;
  auto drop = [&]() {
    for (int s = 0; s < 2; s++) {
      if (p.mem[s][5]) hipFree(p.mem[s][5]);
      p.mem[s][5] = nullptr;
      p.d[s].deep = nullptr;
    }
    p.deep = false;
  };
  drop();
  size_t free_b = 0, total_b = 0;
  GS_HIP(hipMemGetInfo(&free_b, &total_b));
  double reserve = 56e9;
  if (const char *e = gs_opt(ix, "GS_PAIRTAB_RESERVE_GB")) reserve = atof(e) * 1e9;
  if (reserve > 0.25 * (double)total_b) reserve = 0.25 * (double)total_b;
  if (2.0 * 64.0 * (double)(1ull << (2 * kb)) + reserve > (double)free_b) return GS_OK;
  for (int s = 0; s < 2; s++) {
    const gs_status rc = build_deep(ix, p, s, k, P, kb, st);
    if (rc != GS_OK) {
      drop();
      return rc == GS_ERR_NOMEM ? GS_OK : rc;
    }
  }
  p.deep = true;
  p.deep_P = P;
  p.deep_kb = kb;
  return GS_OK;
}
