/*
 * gs_host.hip -- host-side half of the C-ABI: the host-pointer enumerate wrapper, hit
 * decoding, CFD, status strings.  No kernels here.
 */
#include "gs_common.h"
#include "cfd_table.h"

#include <cctype>
#include <cstring>
#include <mutex>

static thread_local std::string g_last_error;
void gs_set_error(const std::string &s) { g_last_error = s; }

/* ---- switches of a handle (gs_index::opts) ---- */
extern char **environ;
std::atomic<int> gs_debug_any{0};
const char *gs_opt(const gs_index *ix, const char *key) {
  if (!ix) return nullptr;
  const auto it = ix->opts.find(key);
  return it == ix->opts.end() ? nullptr : it->second.c_str();
}
void gs_opts_from_env(gs_index *ix) {
  /* the one place that looks at the environment: when a handle is made */
  for (char **e = environ; e && *e; ++e) {
    if (strncmp(*e, "GS_", 3) != 0) continue;
    const char *eq = strchr(*e, '=');
    if (!eq) continue;
    ix->opts[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
  }
  if (ix->opts.count("GS_DEBUG")) gs_debug_any.store(1);
}
extern "C" gs_status gs_index_set_option(gs_index *ix, const char *key, const char *value) {
  GS_HANDLE_LOCK(ix);
  if (!ix || !key || strncmp(key, "GS_", 3) != 0) return GS_ERR_ARG;
  try {
    if (value)
      ix->opts[key] = value;
    else
      ix->opts.erase(key);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
  if (value && strcmp(key, "GS_DEBUG") == 0) gs_debug_any.store(1);
  return GS_OK;
}
extern "C" gs_status gs_index_get_option(const gs_index *ix, const char *key, char *out, uint64_t cap) {
  GS_HANDLE_LOCK(ix);
  if (!ix || !key || !out || cap == 0) return GS_ERR_ARG;
  const char *v = gs_opt(ix, key);
  if (!v) {
    out[0] = 0;
    return GS_ERR_ARG; /* not set */
  }
  const size_t n = strlen(v);
  if (n + 1 > cap) return GS_ERR_ARG;
  memcpy(out, v, n + 1);
  return GS_OK;
}
extern "C" gs_status gs_index_last_sharing(const gs_index *ix, uint64_t out[8]) {
  GS_HANDLE_LOCK(ix);
  if (!ix || !out) return GS_ERR_ARG;
  for (int i = 0; i < 8; i++) out[i] = ix->last_share[i];
  return GS_OK;
}

extern "C" const char *gs_status_string(gs_status s) {
  switch (s) {
    case GS_OK: return "ok";
    case GS_ERR_ARG: return "bad argument";
    case GS_ERR_DEVICE: return g_last_error.empty() ? "device error" : g_last_error.c_str();
    case GS_ERR_UNSUPPORTED:
      return g_last_error.empty() ? "unsupported input" : g_last_error.c_str();
    case GS_ERR_NOMEM: return "out of memory";
    case GS_ERR_IO: return g_last_error.empty() ? "i/o error" : g_last_error.c_str();
    case GS_ERR_FORMAT: return g_last_error.empty() ? "malformed index file" : g_last_error.c_str();
  }
  return "unknown";
}
extern "C" const char *gs_version(void) { return "guidescan-amd 0.2 (gfx950)"; }

/* Page-locked host buffers for the results of the host-pointer entry point, kept in a small
 * process-wide pool: a 1 M-guide batch returns ~215 MB of hits, and a fresh pageable buffer costs
 * more (page faults, staged copy) than the copy itself.  Portable: any device may fill them. */
struct gs_pinned {
  void *p = nullptr;
  size_t cap = 0;
};
static std::mutex g_pin_mtx;
static std::vector<gs_pinned> g_pin_free;
static gs_pinned pin_acquire(size_t bytes) {
  if (!bytes) bytes = 16;
  {
    std::lock_guard<std::mutex> lk(g_pin_mtx);
    int best = -1;
    for (size_t i = 0; i < g_pin_free.size(); i++)
      if (g_pin_free[i].cap >= bytes && (best < 0 || g_pin_free[i].cap < g_pin_free[best].cap)) best = (int)i;
    if (best >= 0) {
      gs_pinned b = g_pin_free[best];
      g_pin_free.erase(g_pin_free.begin() + best);
      return b;
    }
  }
  gs_pinned b;
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&b.p, want, hipHostMallocPortable) == hipSuccess) {
    b.cap = want;
  } else {
    (void)hipGetLastError();
    b.p = nullptr;
    if (hipHostMalloc(&b.p, bytes, hipHostMallocPortable) == hipSuccess) b.cap = bytes;
    else (void)hipGetLastError();
  }
  return b;
}
static void pin_release(gs_pinned b) {
  if (!b.p) return;
  std::lock_guard<std::mutex> lk(g_pin_mtx);
  g_pin_free.push_back(b);
  while (g_pin_free.size() > 8) { /* keep the pool small: drop the smallest buffer */
    size_t k = 0;
    for (size_t i = 1; i < g_pin_free.size(); i++)
      if (g_pin_free[i].cap < g_pin_free[k].cap) k = i;
    hipHostFree(g_pin_free[k].p);
    g_pin_free.erase(g_pin_free.begin() + k);
  }
}

struct gs_result {
  gs_pinned offsets, hits, flags, raw;
  gs_result_view view{};
  ~gs_result() {
    pin_release(offsets);
    pin_release(hits);
    pin_release(flags);
    pin_release(raw);
  }
};

static gs_status enumerate_host(gs_index *ix, const char *guides, uint64_t n, uint32_t L, const char *guide_pams,
                                uint32_t P, const char *alt_pams, uint32_t n_alt, uint32_t mismatches, uint32_t flags,
                                gs_result **out) {
  if (!ix || !out || (n && !guides) || (n && P && !guide_pams)) return GS_ERR_ARG;
  GS_HIP(hipSetDevice(ix->device));
  gs_status rc = gs_reserve(ix->w_guides, n * (size_t)(L + P) + 16);
  if (rc != GS_OK) return rc;
  char *d_g = (char *)ix->w_guides.p;
  char *d_p = d_g + n * (size_t)L;
  if (n) {
    GS_HIP(hipMemcpy(d_g, guides, n * (size_t)L, hipMemcpyHostToDevice));
    if (P) GS_HIP(hipMemcpy(d_p, guide_pams, n * (size_t)P, hipMemcpyHostToDevice));
  }
  const void *d_off = nullptr, *d_hits = nullptr;
  gs_result *r = new gs_result();
  struct guard_t {
    gs_result *p;
    ~guard_t() { delete p; }
  } guard{r};
  rc = gs_enumerate_device(ix, d_g, n, L, d_p, P, alt_pams, n_alt, mismatches, flags, nullptr, &d_off,
                           &d_hits, &r->view);
  if (rc != GS_OK) return rc;
  r->offsets = pin_acquire(8 * (n + 1));
  r->hits = pin_acquire(sizeof(gs_hit) * r->view.n_hits);
  if (!r->offsets.p || !r->hits.p) return GS_ERR_NOMEM;
  GS_HIP(hipMemcpy(r->offsets.p, d_off, 8 * (n + 1), hipMemcpyDeviceToHost));
  if (r->view.n_hits)
    GS_HIP(hipMemcpy(r->hits.p, d_hits, sizeof(gs_hit) * r->view.n_hits, hipMemcpyDeviceToHost));
  r->view.n_guides = n;
  r->view.guide_offsets = (const uint64_t *)r->offsets.p;
  r->view.hits = (const gs_hit *)r->hits.p;
  r->view.n_unsupported = ix->last_unsupported;
  r->view.guide_flags = nullptr;
  if (ix->last_unsupported) {
    r->flags = pin_acquire(n);
    if (!r->flags.p) return GS_ERR_NOMEM;
    GS_HIP(hipMemcpy(r->flags.p, ix->w_flags.p, n, hipMemcpyDeviceToHost));
    r->view.guide_flags = (const uint8_t *)r->flags.p;
  }
  r->view.raw_hits = nullptr;
  if ((flags & GS_FLAG_RAW_COUNTS) && ix->last_raw_valid && n) {
    r->raw = pin_acquire(4 * n);
    if (!r->raw.p) return GS_ERR_NOMEM;
    GS_HIP(hipMemcpy(r->raw.p, ix->w_raw.p, 4 * n, hipMemcpyDeviceToHost));
    r->view.raw_hits = (const uint32_t *)r->raw.p;
  }
  guard.p = nullptr;
  *out = r;
  return GS_OK;
}
extern "C" gs_status gs_enumerate(gs_index *ix, const char *guides, uint64_t n, uint32_t L,
                                  const char *guide_pams, uint32_t P, const char *alt_pams,
                                  uint32_t n_alt, uint32_t mismatches, uint32_t flags,
                                  gs_result **out) {
  GS_HANDLE_LOCK(ix);
  try { /* nothing may throw across the C boundary */
    return enumerate_host(ix, guides, n, L, guide_pams, P, alt_pams, n_alt, mismatches, flags, out);
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
}

extern "C" gs_status gs_result_get(const gs_result *r, gs_result_view *view) {
  if (!r || !view) return GS_ERR_ARG;
  *view = r->view;
  return GS_OK;
}
extern "C" void gs_result_free(gs_result *r) { delete r; }

static char comp(char c) {
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

extern "C" gs_status gs_decode_sequence(const char *guide, uint32_t L, uint32_t P, uint32_t flags,
                                        uint64_t key, char *out) {
  if (!guide || !out || L < 1 || 2 * L + 3 * P > 59) return GS_ERR_ARG;
  const uint64_t path = (key >> 1) & ((1ull << 59) - 1); /* key bits 59:1, position 0 at the top */
  const bool start = flags & GS_FLAG_PAM_AT_START;
  static const char B[4] = {'A', 'C', 'G', 'T'};
  for (uint32_t t = 0; t < L; t++) {
    /* query char consumed at step t (process.hpp:63, index.hpp:218) */
    const char qc = start ? guide[L - 1 - t] : comp(guide[t]);
    const uint32_t code = (uint32_t)(path >> (57 - 2 * t)) & 3u;
    if (code == 0) {
      out[t] = qc;
    } else {
      /* code-1 = rank among the three bases other than qc, in A<C<G<T order */
      int q = qc == 'A' ? 0 : qc == 'C' ? 1 : qc == 'G' ? 2 : qc == 'T' ? 3 : -1;
      if (q < 0) return GS_ERR_ARG;
      int a = (int)code - 1;
      if (a >= q) a++;
      out[t] = (char)tolower(B[a]); /* index.hpp:243 */
    }
  }
  static const char PB[5] = {'A', 'C', 'G', 'N', 'T'};
  for (uint32_t u = 0; u < P; u++) {
    const uint32_t code = (uint32_t)(path >> (56 - 2 * L - 3 * u)) & 7u;
    if (code > 4) return GS_ERR_ARG;
    out[L + u] = PB[code];
  }
  out[L + P] = 0;
  return GS_OK;
}

static int bidx(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}
/* include/genomics/printer.hpp:98-113 with the std::map tables flattened (immutable,
 * so the reference's operator[] data race, SURVEY 5.2, cannot happen here) */
extern "C" float gs_calculate_cfd(const char *sgrna, const char *seq, const char *pam) {
  if (!sgrna || !seq || !pam) return 1.0f;
  if (strlen(sgrna) != 20 || strlen(pam) != 3) return 1.0f;
  float cfd = 1.0f;
  for (int i = 0; i < 20; i++) {
    const char g = sgrna[i], t = seq[i];
    if (g != t) {
      const int r = bidx(g); /* T is looked up as U: same slot */
      const int d = bidx((char)toupper(comp(t)));
      const double sc = (r >= 0 && d >= 0) ? gs_cfd_mm[(r * 4 + d) * 20 + i] : 0.0;
      cfd = (float)((double)cfd * sc);
    }
  }
  const int b1 = bidx(pam[1]), b2 = bidx(pam[2]);
  const double ps = (b1 >= 0 && b2 >= 0) ? gs_cfd_pam[b1 * 4 + b2] : 0.0;
  return (float)((double)cfd * ps);
}

