/* gs_device.h -- device-side helpers shared by the kernels of libgsamd.so: Occ from the 64-byte
 * blocks, the run lists of the rare symbols, wave64 ballot/DPP primitives, the packed guide record. */
#ifndef GS_DEVICE_H
#define GS_DEVICE_H

#include "gs_common.h"

#define WAVE 64

struct gs_guide_rec {
  uint64_t q;      /* 2-bit codes of the query in consumption order: step t at bits [2t+1:2t] */
  uint32_t pam[4]; /* per PAM pattern: 3-bit codes in consumption order (0-3 ACGT, 4 = N wildcard) */
  uint32_t npams;
  uint32_t valid;
};

/* prefix mask of r_j = clamp(r - 32*j, 0, 32) low bits, r in [0,128] */
__device__ __forceinline__ uint32_t word_mask(uint32_t r, uint32_t j) {
  int rj = (int)r - (int)(32u * j);
  rj = rj < 0 ? 0 : rj > 32 ? 32 : rj; /* v_med3_i32 */
  return (uint32_t)((0xFFFFFFFFull << rj) >> 32);
}

/* ---- Occ for all four bases: rows [128*blk, 128*blk + r) of one 64-byte block, r in [0,128] */
__device__ __forceinline__ void occ4(const uint4 *__restrict__ blocks, uint32_t blk, uint32_t r,
                                     uint32_t &oA, uint32_t &oC, uint32_t &oG, uint32_t &oT) {
  const uint4 *p = blocks + ((size_t)blk << 2);
  const uint4 cnt = p[0];
  const uint4 lo = p[1];
  const uint4 hi = p[2];
  const uint4 ex = p[3];
  const uint32_t v0 = ~ex.x & word_mask(r, 0), v1 = ~ex.y & word_mask(r, 1),
                 v2 = ~ex.z & word_mask(r, 2), v3 = ~ex.w & word_mask(r, 3);
  oA = cnt.x + __popc(~lo.x & ~hi.x & v0) + __popc(~lo.y & ~hi.y & v1) +
       __popc(~lo.z & ~hi.z & v2) + __popc(~lo.w & ~hi.w & v3);
  oC = cnt.y + __popc(lo.x & ~hi.x & v0) + __popc(lo.y & ~hi.y & v1) +
       __popc(lo.z & ~hi.z & v2) + __popc(lo.w & ~hi.w & v3);
  oG = cnt.z + __popc(~lo.x & hi.x & v0) + __popc(~lo.y & hi.y & v1) +
       __popc(~lo.z & hi.z & v2) + __popc(~lo.w & hi.w & v3);
  oT = cnt.w + __popc(lo.x & hi.x & v0) + __popc(lo.y & hi.y & v1) + __popc(lo.z & hi.z & v2) +
       __popc(lo.w & hi.w & v3);
}
/* ---- Occ for ONE base c (the common case: mismatch budget spent, or a fixed PAM base) */
__device__ __forceinline__ uint32_t occ1(const uint4 *__restrict__ blocks, uint32_t blk, uint32_t r,
                                         uint32_t c) {
  const uint4 *p = blocks + ((size_t)blk << 2);
  const uint4 cnt = p[0];
  const uint4 lo = p[1];
  const uint4 hi = p[2];
  const uint4 ex = p[3];
  const uint32_t base = c == 0 ? cnt.x : c == 1 ? cnt.y : c == 2 ? cnt.z : cnt.w;
  const uint32_t fl = (c & 1u) ? 0u : 0xFFFFFFFFu; /* flip planes so that "matches c" == 1&1 */
  const uint32_t fh = (c & 2u) ? 0u : 0xFFFFFFFFu;
  return base + __popc((lo.x ^ fl) & (hi.x ^ fh) & ~ex.x & word_mask(r, 0)) +
         __popc((lo.y ^ fl) & (hi.y ^ fh) & ~ex.y & word_mask(r, 1)) +
         __popc((lo.z ^ fl) & (hi.z ^ fh) & ~ex.z & word_mask(r, 2)) +
         __popc((lo.w ^ fl) & (hi.w ^ fh) & ~ex.w & word_mask(r, 3));
}

/* number of BWT rows < i holding a literal 'N' (only the PAM's N can ask: index.hpp:139-149) */
__device__ __forceinline__ uint32_t occ_n(const gs_strand_dev &sd, uint32_t i) {
  uint32_t lo = 0, hi = sd.nruns; /* last run with start < i */
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (sd.run_start[mid] < i)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo == 0) return 0;
  uint32_t r = lo - 1;
  uint32_t len = sd.run_cum[r + 1] - sd.run_cum[r];
  uint32_t d = i - sd.run_start[r];
  return sd.run_cum[r] + (d < len ? d : len);
}

/* number of BWT rows < i holding symbol c, c outside A,C,G,T (general path: a literal query or PAM
 * symbol, index.hpp:139-149, 218-228), from the per-symbol run lists */
__device__ __forceinline__ uint32_t occ_sym(const gs_strand_dev &sd, uint32_t c, uint32_t i) {
  const uint2 seg = sd.xr_seg[c & 255u];
  if (!seg.y) return 0;
  const uint32_t *st = sd.xr_start + seg.x, *cu = sd.xr_cum + seg.x;
  uint32_t lo = 0, hi = seg.y; /* last run with start < i */
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (st[mid] < i)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo == 0) return 0;
  const uint32_t r = lo - 1;
  const uint32_t len = cu[r + 1] - cu[r];
  const uint32_t d = i - st[r];
  return cu[r] + (d < len ? d : len);
}

/* 16 bytes from a 2-byte aligned address (global memory takes unaligned dwordx4 loads) */
typedef uint32_t gs_u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ uint4 load16_a2(const uint16_t *p) {
  const gs_u32x4_a2 v = *(const gs_u32x4_a2 *)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t ballot) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

/* ---- wave64 inclusive scans in seven DPP instructions (row_shr 1,2,3 of the input, row_shr 4
 * and 8 of the partial result under bank masks, then row_bcast 15 and 31 under row masks):
 * a __shfl_up ladder costs six LDS-crossbar round trips and about thirty instructions ---- */
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) { /* lanes without a source read 0 */
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
  uint32_t v = x + dpp_or_zero<0x111>(x) + dpp_or_zero<0x112>(x) + dpp_or_zero<0x113>(x);
  v += dpp_or_zero<0x114, 0xF, 0xE>(v);
  v += dpp_or_zero<0x118, 0xF, 0xC>(v);
  v += dpp_or_zero<0x142, 0xA>(v);
  v += dpp_or_zero<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) {
  uint32_t v = umax32(umax32(x, dpp_or_zero<0x111>(x)), umax32(dpp_or_zero<0x112>(x), dpp_or_zero<0x113>(x)));
  v = umax32(v, dpp_or_zero<0x114, 0xF, 0xE>(v));
  v = umax32(v, dpp_or_zero<0x118, 0xF, 0xC>(v));
  v = umax32(v, dpp_or_zero<0x142, 0xA>(v));
  v = umax32(v, dpp_or_zero<0x143, 0xC>(v));
  return v;
}


struct gs_prep_args {
  const uint8_t *guides;     /* n*L */
  const uint8_t *guide_pams; /* n*P */
  uint8_t alt[32][8];        /* alt PAM patterns (ASCII) */
  gs_guide_rec *out;         /* records of this chunk of the PAM list */
  uint32_t *n_invalid;
  uint8_t *flags;            /* per guide: GS_GUIDE_NEEDS_GENERAL (chunk 0 writes them) */
  uint32_t n, L, P, n_alt, start;
  uint32_t chunk;            /* this launch packs patterns 4*chunk .. 4*chunk+3 of alt_pams ++ [k.pam] */
  uint32_t force_invalid;    /* an alt PAM needs the general path: every guide does */
  uint32_t *pair_hist;       /* or nullptr: [17] patterns of valid guides by the pair of bases they end in (consumption
                                order; code = first | second << 2), [16] = patterns that end in an 'N' */
};

/* k_prepare (gs_search.hip): ASCII guides/PAMs -> packed records, process.hpp:51-63 */
void gs_launch_prepare(const gs_prep_args &pa, hipStream_t st);
int gs_num_cus(int device);

#endif
