/*
 * gs_recipes.hip -- host side of the table seeding: which strand finds which class of sites (thresholds a*(o)) and the
 * seed recipes k_search reads (gs_search_args::rec_*).  No kernels.
 */
#include "gs_kernels.h"

#include <algorithm>
#include <cmath>
#include <vector>

/* Which strand's table finds a site with (a, o, b) substitutions in (X, O, R)?  This strand's
 * table covers X and O: a class of its seeds is a pair (a, o), and verifying such a seed against
 * ctx[] finds every b the budget leaves.  The other strand's table covers O, R and the PAM: its
 * classes are pairs (o, b), each finding every a.  For a fixed o the cells (a, b), a + b <= m - o,
 * must each be covered by row a or by column b; the staircase shape makes every minimal cover
 * "rows a < a*, columns b <= m - o - a*", so the plan is one threshold a*(o) per o.
 * Cost of a class = its seeds x (table line share + chance to survive the context mask x a
 * verification pass), the chances being those of an hg38-sized table (11.5 rows per k-mer); through a
 * PAM-pair table (gs_pairtab.hip) a seed of this strand rarely verifies at all (verify_a ~ 0.2). */
void gs_choose_astar(uint32_t m, uint32_t nX, uint32_t nO, uint32_t nR, double epam, uint32_t astar[8], double verify_a,
                     double verify_b) {
  auto binom3 = [](uint32_t n, uint32_t j) -> double { /* C(n, j) 3^j */
    if (j > n) return 0.0;
    double v = 1;
    for (uint32_t i = 0; i < j; i++) v = v * (n - i) / (i + 1) * 3.0;
    return v;
  };
  static const double pass[4] = {0.17, 0.86, 1.0, 1.0};
  auto seed_cost = [&](uint32_t budget_left, double verify) -> double {
    return 0.35 + pass[budget_left < 3 ? budget_left : 3] * verify;
  };
  for (uint32_t o = 0; o < 8; o++) {
    astar[o] = 15;
    if (o > m || o > nO) continue;
    const uint32_t M = m - o;
    double best = -1;
    for (uint32_t as = 0; as <= M + 1; as++) {
      double c = 0;
      for (uint32_t a = 0; a < as && a <= M; a++) c += binom3(nX, a) * seed_cost(M - a, verify_a);
      if (as <= M) {
        if (as > nX) continue; /* the other side would need more substitutions in X than X holds */
        for (uint32_t b = 0; b + as <= M; b++) c += epam * binom3(nR, b) * seed_cost(M - b, verify_b);
      }
      if (best < 0 || c < best) {
        best = c;
        astar[o] = as <= M ? as : 15;
      }
    }
  }
  /* k_search sizes a class's two-symbol extension by the largest o it may reach: keep the
   * thresholds non-increasing in o so that "allowed at o" implies "allowed below o" */
  for (uint32_t o = 1; o < 8; o++)
    if (astar[o] > astar[o - 1]) astar[o] = astar[o - 1];
}

/* ---- seed recipes (gs_search_args::rec_*) -------------------------------------------------------
 * The depth-k seeds of an item are the same set of substitution patterns for every guide: which
 * steps are substituted, by which of the three other bases (a digit relative to the guide's own
 * symbol), read from which copy of the table.  The lists are written once per (budget, geometry,
 * thresholds) and kept on the handle; a seeding step of k_search hands recipe pos + lane to lane. */
static inline uint64_t recipe_word(uint32_t n, uint32_t lo, bool rot, uint32_t rs, const uint32_t *fields) {
  uint64_t w = (uint64_t)n | ((uint64_t)lo << 3) | (rot ? (1ull << 6) | ((uint64_t)rs << 7) : 0ull);
  for (uint32_t i = 0; i < 7; i++) w |= (uint64_t)(i < n ? fields[i] : 3u) << (12 + 7 * i);
  return w;
}
/* this strand's seeds: variants of the first k-2 steps with j substitutions (ax of them among the first
 * nX steps, set X) x the two-symbol extensions the budget allows; two-sided (astar != nullptr): only
 * what has ax < astar[substitutions outside X] */
static void build_recipes_a(std::vector<uint64_t> &out, uint32_t k, uint32_t m, uint32_t nX, const uint32_t *astar, bool rot,
                            bool pair8 = false) {
  const uint32_t kp = k - 2, xmask = nX >= 32 ? 0xFFFFFFFFu : (1u << nX) - 1u;
  auto mine = [&](uint32_t ax, uint32_t o) { return !astar || (o < 8 && ax < astar[o]); };
  const uint32_t jmax = std::min(std::min(m, kp), 7u);
  for (uint32_t j = 0; j <= jmax; j++)
    for (uint32_t mk = 0; mk < (1u << kp); mk++) {
      if ((uint32_t)__builtin_popcount(mk) != j) continue;
      const uint32_t ax = (uint32_t)__builtin_popcount(mk & xmask), o0 = j - ax;
      if (!mine(ax, o0)) continue;
      if (astar && nX > kp) {
        /* X reaches into the two-symbol extension (27-symbol sites at k = 14: X = steps 0 .. 12): a substitution at step
         * k-2 counts for X, one at step k-1 for O - every (e2, e1) is taken or left by itself, from the plain table */
        uint32_t steps[8], ns = 0;
        for (uint32_t t = 0; t < kp; t++)
          if ((mk >> t) & 1u) steps[ns++] = t;
        uint32_t ndig = 1;
        for (uint32_t i = 0; i < j; i++) ndig *= 3;
        for (uint32_t dc = 0; dc < ndig; dc++) {
          uint32_t f[10], x = dc;
          for (uint32_t i = j; i-- > 0;) {
            f[i] = (steps[i] << 2) | (x % 3);
            x /= 3;
          }
          for (uint32_t e2 = 0; e2 < 4; e2++)
            for (uint32_t e1 = 0; e1 < 4; e1++) {
              uint32_t n = j;
              if (e2) f[n++] = ((k - 2) << 2) | (e2 - 1);
              if (e1) f[n++] = ((k - 1) << 2) | (e1 - 1);
              if (n > m || n > 7 || !mine(ax + (e2 ? 1u : 0u), o0 + (e1 ? 1u : 0u))) continue;
              out.push_back(recipe_word(n, 0, false, 0, f));
            }
        }
        continue;
      }
      uint32_t eb = 0; /* substitutions the extension may add */
      while (eb < 2 && j + eb + 1 <= m && mine(ax, o0 + eb + 1)) eb++;
      uint32_t steps[8], ns = 0, plast = 0;
      for (uint32_t t = 0; t < kp; t++)
        if ((mk >> t) & 1u) steps[ns++] = plast = t;
      uint32_t ndig = 1;
      for (uint32_t i = 0; i < j; i++) ndig *= 3;
      for (uint32_t dc = 0; dc < ndig; dc++) {
        uint32_t f[8], x = dc;
        for (uint32_t i = j; i-- > 0;) { /* the last substituted step's digit runs fastest */
          f[i] = (steps[i] << 2) | (x % 3);
          x /= 3;
        }
        auto emit = [&](uint32_t e2, uint32_t e1, bool r, uint32_t rs) {
          uint32_t n = j;
          if (e2) f[n++] = ((k - 2) << 2) | (e2 - 1);
          if (e1) f[n++] = ((k - 1) << 2) | (e1 - 1);
          if (n > m || n > 7 || !mine(ax, o0 + (n - j))) return;
          out.push_back(recipe_word(n, 0, r, rs, f));
        };
        if (eb >= 2) { /* 16 neighbours of the plain table: 4 lines */
          for (uint32_t e2 = 0; e2 < 4; e2++)
            for (uint32_t e1 = 0; e1 < 4; e1++) emit(e2, e1, false, 0);
        } else if (eb == 1) { /* one line of the plain table + one of the copy rotated at step k-2 */
          for (uint32_t e1 = 0; e1 < 4; e1++) emit(0, e1, false, 0);
          /* a PAM-pair table's 8-byte entries: the three are in the same 128-byte block as the four */
          for (uint32_t e2 = 1; e2 < 4; e2++) emit(e2, 0, rot && !pair8, k - 2);
        } else {
          emit(0, 0, rot && j >= 1, plast);
        }
      }
    }
}
/* the other strand's seeds under two-sided seeding: classes (o substitutions in O, b in R) with
 * astar[o] + o + b <= m; step y consumes guide symbol L-1-y: R = y in [0, L-k), O = y in [L-k, k-P).
 * The recipe carries lo = astar[o], the least number of substitutions its rows need inside X. */
static void build_recipes_b(std::vector<uint64_t> &out, uint32_t k, uint32_t L, uint32_t P, uint32_t m, uint32_t nX,
                            const uint32_t *astar, bool rot, bool deep) {
  /* deep tables (gs_pairtab.hip): k-2 guide symbols index the table, the copies are numbered by guide symbol */
  const uint32_t nO = k - nX, nR = L - k, ylo = L - k, nY = L - nX, step0 = deep ? 0 : P, kd = deep ? nY : k;
  for (uint32_t o = 0; o <= m && o <= nO && o < 8; o++)
    for (uint32_t b = 0; b <= nR && astar[o] + o + b <= m; b++) {
      if (astar[o] > nX || o + b > 7) continue;
      const uint32_t jb = o + b;
      uint32_t ndig = 1;
      for (uint32_t i = 0; i < jb; i++) ndig *= 3;
      for (uint32_t mo = 0; mo < (1u << nO); mo++) {
        if ((uint32_t)__builtin_popcount(mo) != o) continue;
        for (uint32_t mr = 0; mr < (1u << nR); mr++) {
          if ((uint32_t)__builtin_popcount(mr) != b) continue;
          const uint32_t mk = (mo << ylo) | mr;
          uint32_t ys[8], ns = 0, ymax = 0;
          for (uint32_t y = 0; y < nY; y++)
            if ((mk >> y) & 1u) ys[ns++] = ymax = y;
          const uint32_t slast = step0 + ymax; /* consumption step of the last substituted symbol */
          const bool r = rot && !deep && jb >= 1 && slast + 2 <= kd; /* a deep table's line is one index */
          for (uint32_t dc = 0; dc < ndig; dc++) {
            uint32_t f[8], x = dc;
            for (uint32_t i = jb; i-- > 0;) {
              f[i] = (ys[i] << 2) | (x % 3);
              x /= 3;
            }
            out.push_back(recipe_word(jb, astar[o] > 7 ? 7u : astar[o], r, slast, f));
          }
        }
      }
    }
}
extern "C" gs_status gs_debug_seed_recipes(uint32_t k, uint32_t L, uint32_t P, uint32_t m, uint32_t n_x,
                                           const uint32_t *astar, uint32_t deep, uint64_t *out, uint64_t cap,
                                           uint64_t counts[3]) {
  if (k < 4 || k > 16 || L < k || L > 31 || m > 7 || n_x + 1 > k || !counts) return GS_ERR_ARG;
  try {
    std::vector<uint64_t> all;
    build_recipes_a(all, k, m, n_x, nullptr, true);
    counts[0] = all.size();
    counts[1] = counts[2] = 0;
    if (astar) {
      build_recipes_a(all, k, m, n_x, astar, true);
      counts[1] = all.size() - counts[0];
      build_recipes_b(all, k, L, P, m, n_x, astar, true, deep != 0);
      counts[2] = all.size() - counts[0] - counts[1];
    }
    for (uint64_t i = 0; i < all.size() && i < cap && out; i++) out[i] = all[i];
  } catch (const std::bad_alloc &) {
    return GS_ERR_NOMEM;
  }
  return GS_OK;
}
extern "C" void gs_debug_choose_thresholds(uint32_t m, uint32_t n_x, uint32_t n_o, uint32_t n_r, double pam_expansions,
                                           double verify_a, double verify_b, uint32_t astar[8]) {
  gs_choose_astar(m, n_x, n_o, n_r, pam_expansions, astar, verify_a, verify_b);
}

gs_status gs_recipes_for(gs_index *ix, uint32_t L, uint32_t P, uint32_t m, uint32_t v_rem, const uint32_t *astar,
                                bool deep, hipStream_t st) {
  const uint32_t k = ix->pt_k;
  const bool rot = true; /* the recipes name the copy that would share lines; a table without it reads its plain copy */
  uint64_t key[2] = {((uint64_t)L << 48) | ((uint64_t)P << 40) | ((uint64_t)m << 32) | ((uint64_t)k << 24) |
                         ((uint64_t)v_rem << 16) | (deep ? 4u : 0u) | (rot ? 2u : 0u) | (astar ? 1u : 0u),
                     0};
  if (astar)
    for (uint32_t o = 0; o < 8; o++) key[1] |= (uint64_t)(astar[o] > 15 ? 15u : astar[o]) << (4 * o);
  for (uint32_t i = 0; i < 2; i++)
    if (ix->rec[i].valid && ix->rec[i].key[0] == key[0] && ix->rec[i].key[1] == key[1]) {
      ix->rec_cur = i;
      return GS_OK;
    }
  std::vector<uint64_t> all;
  build_recipes_a(all, k, m, v_rem, nullptr, rot);
  const size_t n_full = all.size();
  size_t n_a = 0, n_b = 0;
  size_t n_a8 = 0;
  if (astar) {
    build_recipes_a(all, k, m, v_rem, astar, rot);
    n_a = all.size() - n_full;
    build_recipes_b(all, k, L, P, m, v_rem, astar, rot, deep);
    n_b = all.size() - n_full - n_a;
    build_recipes_a(all, k, m, v_rem, astar, rot, true); /* this strand's share read through PAM-pair tables */
    n_a8 = all.size() - n_full - n_a - n_b;
  }
  if (all.size() >= (1ull << 31)) {
    gs_set_error("seed plan too large for this mismatch budget");
    return GS_ERR_UNSUPPORTED;
  }
  /* into the set the last call did not use (or an empty one) */
  const uint32_t slot = !ix->rec[ix->rec_cur].valid ? ix->rec_cur : ix->rec_cur ^ 1u;
  gs_recipe_set &R = ix->rec[slot];
  R.valid = false;
  gs_status rc = gs_reserve(R.buf, 8 * all.size() + 64);
  if (rc != GS_OK) return rc;
  GS_HIP(hipMemcpyAsync(R.buf.p, all.data(), 8 * all.size(), hipMemcpyHostToDevice, st));
  GS_HIP(hipStreamSynchronize(st)); /* `all` is a local */
  R.n_full = (uint32_t)n_full;
  R.n_a = (uint32_t)n_a;
  R.n_b = (uint32_t)n_b;
  R.n_a8 = (uint32_t)n_a8;
  R.a_rot_first = 31; /* of the list read through PAM-pair tables */
  for (size_t i = n_full + n_a + n_b; i < all.size(); i++)
    if (all[i] & 64u) R.a_rot_first = std::min(R.a_rot_first, (uint32_t)(all[i] >> 7) & 31u);
  R.key[0] = key[0];
  R.key[1] = key[1];
  R.valid = true;
  ix->rec_cur = slot;
  if (gs_opt(ix, "GS_DEBUG"))
    fprintf(stderr, "[gs] seed recipes: %zu one-sided, %zu + %zu two-sided (%.1f MB)\n", n_full, n_a, n_b, 8e-6 * all.size());
  return GS_OK;
}

