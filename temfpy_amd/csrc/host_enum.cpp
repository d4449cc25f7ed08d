// Host side of the sweep: the integer / combinatorial work between the dense kernels.
//
//  tmf_cut_vectors   best-first enumeration of occupation patterns and their ordering
//                    (schmidt_utils.py:211-324, :99-185; slater.py:662-689)
//  tmf_site_prepare  merged bra leg, always/sometimes/never split with anticommutation
//                    signs, charge-sector matching and the index lists of every minor
//                    (slater.py:1023-1067, :760-825, :1132-1141, :857-864)
//
// Occupation patterns are 128-bit masks instead of boolean arrays; everything else keeps the
// reference's ordering rules so that integer outputs are identical.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/temfpy_hip.h"

namespace tmf {
void set_error(const char* fmt, ...);
}

namespace {

struct Mask {
  uint64_t lo, hi;
  bool get(int i) const { return i < 64 ? (lo >> i) & 1 : (hi >> (i - 64)) & 1; }
  void flip(int i) {
    if (i < 64) lo ^= (1ull << i);
    else hi ^= (1ull << (i - 64));
  }
  int count() const { return __builtin_popcountll(lo) + __builtin_popcountll(hi); }
};

// numpy's pairwise summation (the order np.sum uses on a contiguous double array)
double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

}  // namespace

extern "C" int tmf_cut_vectors(const double* e, int k, int filled_left, int64_t chi_max, double svd_min,
                               double degeneracy_tol, const int64_t* sectors, int n_sectors, int64_t chi_cap,
                               uint64_t* sets_out, double* lam_raw, int32_t* q_left, int64_t* chi_out,
                               int64_t* n_checked_out) {
  if (k < 0 || k > 128) {
    tmf::set_error("tmf_cut_vectors: %d entangled orbitals (supported: 0..128)", k);
    return TMF_E_LIMIT;
  }
  if (!(svd_min > 0.0 && svd_min < 1.0) || !(degeneracy_tol > 0.0)) {
    tmf::set_error("tmf_cut_vectors: svd_min must be in (0,1) and degeneracy_tol positive");
    return TMF_E_ARG;
  }
  auto in_sector = [&](int q) {
    if (!sectors) return true;
    for (int i = 0; i < n_sectors; ++i)
      if (sectors[i] == q) return true;
    return false;
  };
  const double max_logval = -log(svd_min) + degeneracy_tol;  // schmidt_utils.py:96

  std::vector<double> sums;
  std::vector<Mask> sets;
  int64_t n_checked = 1;

  if (k == 0) {  // schmidt_utils.py:268-271
    if (in_sector(filled_left)) {
      sums.push_back(0.0);
      sets.push_back(Mask{0, 0});
    }
  } else {
    std::vector<double> a(k), av(k), neg;
    for (int i = 0; i < k; ++i) {
      a[i] = log((1.0 - e[i]) / e[i]) / 2.0;  // slater.py:428, :663
      av[i] = fabs(a[i]);
    }
    Mask min_set{0, 0};
    for (int i = 0; i < k; ++i)
      if (a[i] < 0) {
        neg.push_back(a[i]);
        min_set.flip(i);
      }
    const double min_sum = np_pairwise_sum(neg.data(), (int)neg.size());
    if (in_sector(filled_left + min_set.count())) {
      sums.push_back(min_sum);
      sets.push_back(min_set);
    }
    std::vector<int> idx(k);
    for (int i = 0; i < k; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return av[x] < av[y]; });

    // binary heap of 16-byte items (sum, seq, payload index); the occupation masks live in a side array,
    // so sifting moves 16 bytes instead of 40 (the heap operations are ~all of this function's time)
    struct Item {
      double sum;
      uint32_t seq, id;
    };
    auto greater = [](const Item& a, const Item& b) { return a.sum != b.sum ? a.sum > b.sum : a.seq > b.seq; };
    struct Payload {
      Mask set;
      int i;
    };
    std::vector<Item> heap;
    std::vector<Payload> pay;
    const size_t guess = chi_max > 0 ? (size_t)(2 * chi_max + 8) : 4096;
    heap.reserve(guess), pay.reserve(guess), sums.reserve(guess / 2 + 2), sets.reserve(guess / 2 + 2);
    uint32_t seq = 0;
    auto push = [&](double sum, int i, const Mask& set) {
      heap.push_back(Item{sum, seq++, (uint32_t)pay.size()});
      pay.push_back(Payload{set, i});
      std::push_heap(heap.begin(), heap.end(), greater);
    };
    {
      Mask s = min_set;
      s.flip(idx[0]);
      push(min_sum + av[idx[0]], 0, s);
    }
    auto keep_generating = [&]() {  // schmidt_utils.py:99-138
      if (chi_max > 0 && (int64_t)sums.size() > chi_max) return false;
      if (!sums.empty() && sums.back() - sums.front() > max_logval) return false;
      return true;
    };
    while (!heap.empty() && keep_generating()) {
      ++n_checked;
      std::pop_heap(heap.begin(), heap.end(), greater);
      const Item it = heap.back();
      heap.pop_back();
      const Payload nd = pay[it.id];
      if (in_sector(filled_left + nd.set.count())) {
        sums.push_back(it.sum);
        sets.push_back(nd.set);
      }
      if (nd.i < k - 1) {
        Mask c1 = nd.set;
        c1.flip(idx[nd.i + 1]);
        double s = it.sum + av[idx[nd.i + 1]];
        push(s, nd.i + 1, c1);
        Mask c2 = c1;
        c2.flip(idx[nd.i]);
        s = s - av[idx[nd.i]];
        push(s, nd.i + 1, c2);
      }
    }
  }
  if (n_checked_out) *n_checked_out = n_checked;

  // truncate (schmidt_utils.py:140-185)
  int64_t cut = 0;
  const int64_t n = (int64_t)sums.size();
  if (n > 0) {
    const double lim = -log(svd_min);
    for (int64_t i = 0; i < n; ++i) {
      bool ok = true;
      if (chi_max > 0 && i >= chi_max) ok = false;
      if (!(sums[i] - sums[0] < lim)) ok = false;
      if (i + 1 < n && !((sums[i + 1] - sums[i]) > degeneracy_tol)) ok = false;
      if (ok) cut = i + 1;
    }
    if (cut == 0) {
      // numpy: np.nonzero(good)[0][-1] raises IndexError here; report it as an argument error
      tmf::set_error("tmf_cut_vectors: no admissible truncation point (degenerate multiplet wider than chi_max)");
      return TMF_E_ARG;
    }
  }
  *chi_out = cut;
  if (cut > chi_cap) {
    tmf::set_error("tmf_cut_vectors: %lld vectors exceed the output capacity %lld", (long long)cut,
                   (long long)chi_cap);
    return TMF_E_LIMIT;
  }
  // stable sort by left charge (slater.py:672-678), Schmidt values (slater.py:489)
  std::vector<int64_t> ord(cut);
  {
    int start[130] = {0};
    for (int64_t i = 0; i < cut; ++i) ++start[sets[i].count() + 1];
    for (int b = 1; b < 130; ++b) start[b] += start[b - 1];
    for (int64_t i = 0; i < cut; ++i) ord[start[sets[i].count()]++] = i;
  }
  // Occupation patterns of EQUAL weight (particle-hole symmetric spectra, identical spin species: sums that differ by
  // rounding only) come out of the heap in an order set by the last bits of the eigenvalues - in the reference as well
  // (threshold events).  Two sweeps over different matrices that contain the same cut (C_to_MPS of the
  // short chain and C_to_iMPS, src/examples/iMPS.py:27-38) must still number them alike, so inside a charge sector runs of
  // patterns whose sums agree to 1e-10 are put in ascending order of their masks.  TMF_TIE_ORDER=0 keeps the heap's order.
  // "Agree" is meant up to what the eigenvalues themselves carry: e is known to ~4e-15 absolutely, so
  // a_i = ln((1 - e_i) / e_i) / 2 to da_i = 2e-15 / min(e_i, 1 - e_i), and two patterns are tied when their sums differ by
  // less than the da of the orbitals in which they differ (an orbital of weight 1e-10 and its particle-hole partner give
  // partner patterns whose weights differ by 5e-6 relatively - by rounding alone, and differently in every sweep).
  static const bool tie_order = !(getenv("TMF_TIE_ORDER") && atoi(getenv("TMF_TIE_ORDER")) == 0);
  double da[128];
  for (int i = 0; i < k; ++i) da[i] = 2e-15 / std::max(std::min(e[i], 1.0 - e[i]), 1e-300);
  auto tied = [&](int64_t x, int64_t y) {
    const Mask& p = sets[x];
    const Mask& q = sets[y];
    if (p.count() != q.count()) return false;
    uint64_t dl = p.lo ^ q.lo, dh = p.hi ^ q.hi;
    double tol = 1e-10 * std::max(1.0, fabs(sums[x]));
    for (; dl; dl &= dl - 1) tol += da[__builtin_ctzll(dl)];
    for (; dh; dh &= dh - 1) tol += da[64 + __builtin_ctzll(dh)];
    return fabs(sums[y] - sums[x]) <= tol;
  };
  if (tie_order)
    for (int64_t a = 0; a < cut;) {
      int64_t b = a;
      while (b + 1 < cut && tied(ord[b], ord[b + 1])) ++b;
      if (b > a)
        std::sort(ord.begin() + a, ord.begin() + b + 1, [&](int64_t x, int64_t y) {
          return sets[x].hi != sets[y].hi ? sets[x].hi < sets[y].hi : sets[x].lo < sets[y].lo;
        });
      a = b + 1;
    }
  double tab[2][128];
  for (int i = 0; i < k; ++i) tab[0][i] = 1.0 - e[i], tab[1][i] = e[i];
  for (int64_t r = 0; r < cut; ++r) {
    const Mask& s = sets[ord[r]];
    sets_out[2 * r] = s.lo;
    sets_out[2 * r + 1] = s.hi;
    q_left[r] = filled_left + s.count();
    double prod = 1.0;
    for (int i = 0; i < k; ++i) prod *= tab[s.get(i)][i];
    lam_raw[r] = sqrt(prod);
  }
  return TMF_OK;
}

// -------------------------------------------------------------------------------------------
namespace {

struct Side {
  int k, nf, chi;
  const uint64_t* sets;
  bool right;
  // occupation of entangled orbital j (side order) in Schmidt vector alpha
  bool occ(int alpha, int j) const {
    Mask m{sets[2 * alpha], sets[2 * alpha + 1]};
    return right ? !m.get(k - 1 - j) : m.get(j);
  }
  int ent_count(int alpha) const {
    Mask m{sets[2 * alpha], sets[2 * alpha + 1]};
    return right ? k - m.count() : m.count();
  }
};

}  // namespace

extern "C" int tmf_site_prepare(const tmf_site_in* in, const uint64_t* sets_b, const int32_t* q_b,
                                const uint64_t* sets_k, const int32_t* q_k, int32_t* row_sel, int8_t* row_sign,
                                int32_t* col_sel, int8_t* col_sign, int32_t* bra_p, int32_t* bra_alpha,
                                tmf_sector* sectors, int32_t sector_cap, uint8_t* idx_pool, int64_t idx_cap,
                                tmf_site_out* out) {
  (void)q_b;
  const bool right = (in->mode & 1) != 0;
  const bool phys = !(in->mode & 2);   // bit 1: overlap of two bases of the SAME orbitals, no physical leg (slater.py:1023-1024)
  Side B{in->k_b, in->nf_b, in->chi_b, sets_b, right};
  Side K{in->k_k, in->nf_k, in->chi_k, sets_k, right};
  if (B.chi <= 0 || K.chi <= 0) {
    tmf::set_error("tmf_site_prepare: empty Schmidt basis");
    return TMF_E_ARG;
  }

  // ---- classify entangled orbitals: 0 never, 1 sometimes, 2 always --------------------------
  auto classify = [](const Side& s, std::vector<int>& cls) {
    // AND / OR of all occupation masks: an orbital is always (never) occupied iff its bit is set in
    // every (no) mask - mirrored for the right side, where occupation is the complement
    uint64_t and_lo = ~0ull, and_hi = ~0ull, or_lo = 0, or_hi = 0;
    for (int a = 0; a < s.chi; ++a) {
      and_lo &= s.sets[2 * a], and_hi &= s.sets[2 * a + 1];
      or_lo |= s.sets[2 * a], or_hi |= s.sets[2 * a + 1];
    }
    const Mask all{and_lo, and_hi}, any{or_lo, or_hi};
    cls.assign(s.k, 0);
    for (int j = 0; j < s.k; ++j) {
      const int bit = s.right ? s.k - 1 - j : j;
      const bool always = s.right ? !any.get(bit) : all.get(bit);
      const bool never = s.right ? all.get(bit) : !any.get(bit);
      cls[j] = never ? 0 : (always ? 2 : 1);
    }
  };
  std::vector<int> cb, ck;
  classify(B, cb);
  classify(K, ck);

  // orbital lists in the reference's column order; entry = device column (-1: physical)
  struct Orb {
    int src;    // device column: entangled j -> j ; filled f -> k + f ; physical -> -1
    int ent;    // entangled index or -1
    int sign;
  };
  auto build = [&](const Side& s, const std::vector<int>& cls, bool with_phys, std::vector<Orb>& always,
                   std::vector<Orb>& some) {
    always.clear();
    some.clear();
    if (!right) {  // reference order: filled, entangled, [physical]
      for (int f = 0; f < s.nf; ++f) always.push_back(Orb{s.k + f, -1, 1});
      for (int j = 0; j < s.k; ++j)
        if (cls[j] == 2) always.push_back(Orb{j, j, 1});
      const int ka = (int)always.size();
      int n_always_before = s.nf;
      for (int j = 0; j < s.k; ++j) {
        if (cls[j] == 2) ++n_always_before;
        if (cls[j] == 1) some.push_back(Orb{j, j, ((ka - n_always_before) & 1) ? -1 : 1});  // slater.py:813
      }
      if (with_phys) some.push_back(Orb{-1, -1, 1});  // last column: no always orbital to its right
    } else {  // reference order: [physical], entangled, filled
      if (with_phys) some.push_back(Orb{-1, -1, 1});  // first column: no always orbital to its left
      int n_always_before = 0;
      for (int j = 0; j < s.k; ++j) {
        if (cls[j] == 2) {
          always.push_back(Orb{j, j, 1});
          ++n_always_before;
        }
        if (cls[j] == 1) some.push_back(Orb{j, j, (n_always_before & 1) ? -1 : 1});  // slater.py:820
      }
      for (int f = 0; f < s.nf; ++f) always.push_back(Orb{s.k + f, -1, 1});
    }
  };
  std::vector<Orb> ab, sb_, ak, sk_;
  build(B, cb, phys, ab, sb_);
  build(K, ck, false, ak, sk_);
  const int kb = (int)ab.size(), kk = (int)ak.size();
  const int k = std::min(kb, kk);  // slater.py:1069

  // W rows: [always block (k)] then the S rows in the reference's trimmed order
  std::vector<Orb> rows, cols;
  std::vector<Orb> srows, scols;  // S-row / S-col orbitals (positions used by the minors)
  auto assemble = [&](const std::vector<Orb>& always, const std::vector<Orb>& some, std::vector<Orb>& all,
                      std::vector<Orb>& s_part) {
    const int ka = (int)always.size();
    all.clear();
    s_part.clear();
    if (!right) {  // O[:k,:k] is the always block; O[k:] = remaining always, then sometimes
      for (int i = 0; i < k; ++i) all.push_back(always[i]);
      for (int i = k; i < ka; ++i) s_part.push_back(always[i]);
      for (auto& o : some) s_part.push_back(o);
    } else {  // O[-k:,-k:] is the always block; O[:-k] = sometimes, then the first ka-k always
      for (int i = ka - k; i < ka; ++i) all.push_back(always[i]);
      for (auto& o : some) s_part.push_back(o);
      for (int i = 0; i < ka - k; ++i) s_part.push_back(always[i]);
    }
    for (auto& o : s_part) all.push_back(o);
  };
  assemble(ab, sb_, rows, srows);
  assemble(ak, sk_, cols, scols);
  // Order inside the always block (free: the Schur complement does not depend on it, det(A) only through the parity,
  // which goes into the sign of the first row).  Filled orbital f of either side is column f of the same random block
  // pushed through the side's projector and orthonormalised in column order, so <filled_b f | filled_k f> is the
  // dominant entry of its row and column.  Those pairs go on the diagonal of the leading part; the orbitals without a
  // partner (always-occupied entangled ones, surplus filled ones) come last: they repair the rank the paired part
  // lacks when the two sides have different numbers of filled orbitals, which only shows in its last pivots.  The
  // block-local pivoting of tmf_lu_diag_batched relies on this (lu_schur.hip); the fully pivoted kernels do not care.
  {
    auto filled_index = [&](const Orb& o, int k_side) { return (o.ent < 0 && o.src >= 0) ? o.src - k_side : -1; };
    const int nf_max = std::max(B.nf, K.nf);
    std::vector<char> in_rows((size_t)nf_max + 1, 0), paired((size_t)nf_max + 1, 0);
    for (int i = 0; i < k; ++i) {
      const int f = filled_index(rows[i], B.k);
      if (f >= 0) in_rows[(size_t)f] = 1;
    }
    for (int i = 0; i < k; ++i) {
      const int f = filled_index(cols[i], K.k);
      if (f >= 0 && f <= nf_max && in_rows[(size_t)f]) paired[(size_t)f] = 1;
    }
    auto reorder = [&](std::vector<Orb>& v, int k_side) -> int {   // returns the parity of the permutation
      std::vector<Orb> rest, pairs;   // filled orbitals appear in ascending f on both sides
      long inversions = 0;
      for (int i = 0; i < k; ++i) {
        const int f = filled_index(v[i], k_side);
        if (f >= 0 && paired[(size_t)f]) pairs.push_back(v[i]), inversions += (long)rest.size();
        else rest.push_back(v[i]);
      }
      for (size_t i = 0; i < pairs.size(); ++i) v[i] = pairs[i];
      for (size_t i = 0; i < rest.size(); ++i) v[pairs.size() + i] = rest[i];
      return (int)(inversions & 1);
    };
    const int parity = reorder(rows, B.k) ^ reorder(cols, K.k);
    if (parity && k > 0) rows[0].sign = -rows[0].sign;
  }
  const int mb = (int)rows.size(), mk = (int)cols.size();
  const int sb = mb - k, sk = mk - k;
  if (sb > 255 || sk > 255) {
    tmf::set_error("tmf_site_prepare: %d x %d sometimes-matrix exceeds the 8-bit index lists", sb, sk);
    return TMF_E_LIMIT;
  }
  for (int i = 0; i < mb; ++i) row_sel[i] = rows[i].src, row_sign[i] = (int8_t)rows[i].sign;
  for (int i = 0; i < mk; ++i) col_sel[i] = cols[i].src, col_sign[i] = (int8_t)cols[i].sign;

  // ---- merged bra leg: (p, alpha) sorted stably by the charge to the left (slater.py:1053-1058)
  const int nb2 = phys ? 2 * B.chi : B.chi;
  std::vector<int> cnt(nb2), perm(nb2);
  for (int r = 0; r < nb2; ++r) {
    const int p = r >= B.chi, a = r % B.chi;
    cnt[r] = B.nf + B.ent_count(a) + p;  // particles in the bra orbitals incl. the physical one
    perm[r] = r;
  }
  {  // stable counting sort by particle number (ascending on the left, descending on the right)
    int cmin = cnt[0], cmax = cnt[0];
    for (int r = 1; r < nb2; ++r) cmin = std::min(cmin, cnt[r]), cmax = std::max(cmax, cnt[r]);
    std::vector<int> start(cmax - cmin + 2, 0);
    for (int r = 0; r < nb2; ++r) ++start[(right ? cmax - cnt[r] : cnt[r] - cmin) + 1];
    for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
    for (int r = 0; r < nb2; ++r) perm[start[right ? cmax - cnt[r] : cnt[r] - cmin]++] = r;
  }
  for (int r = 0; r < nb2; ++r) {
    bra_p[r] = perm[r] >= B.chi;
    bra_alpha[r] = perm[r] % B.chi;
  }

  // ---- charge sectors of the ket leg and the matching bra rows (slater.py:1132-1141) ----------
  auto occupied = [&](const Side& s, const Orb& o, int alpha, int p) -> bool {
    if (o.src < 0) return p != 0;
    if (o.ent < 0) return true;  // filled
    return s.occ(alpha, o.ent);
  };
  // Occupancy of the S-rows / S-columns of one Schmidt vector as a 64-bit word (bit i = S position i):
  // constant bits for filled orbitals, the physical bit, and one shift per entangled position.
  struct SMap {
    uint64_t fixed = 0;      // filled orbitals: always occupied
    int phys = -1;           // S position of the physical orbital
    int n = 0;
    uint8_t pos[64], bit[64], inv[64];
  };
  auto make_map = [&](const Side& s, const std::vector<Orb>& part, SMap& mp) -> bool {
    if (part.size() > 64) return false;
    for (size_t i = 0; i < part.size(); ++i) {
      const Orb& o = part[i];
      if (o.src < 0) mp.phys = (int)i;
      else if (o.ent < 0) mp.fixed |= 1ull << i;
      else {
        mp.pos[mp.n] = (uint8_t)i;
        mp.bit[mp.n] = (uint8_t)(s.right ? s.k - 1 - o.ent : o.ent);
        mp.inv[mp.n] = s.right ? 1 : 0;
        ++mp.n;
      }
    }
    return true;
  };
  auto occ_word = [](const SMap& mp, const uint64_t* set2, int p) -> uint64_t {
    uint64_t wd = mp.fixed;
    if (mp.phys >= 0 && p) wd |= 1ull << mp.phys;
    const uint64_t lo = set2[0], hi = set2[1];
    for (int t = 0; t < mp.n; ++t) {
      const int b = mp.bit[t];
      const uint64_t v = ((b < 64 ? lo >> b : hi >> (b - 64)) & 1ull) ^ mp.inv[t];
      wd |= v << mp.pos[t];
    }
    return wd;
  };
  SMap map_b, map_k;
  const bool fast = make_map(B, srows, map_b) && make_map(K, scols, map_k);
  int nsec = 0;
  int64_t idx_used = 0, out_elems = 0;
  int c0 = 0;
  while (c0 < K.chi) {
    int c1 = c0;
    while (c1 < K.chi && q_k[c1] == q_k[c0]) ++c1;
    const int kcnt = K.nf + K.ent_count(c0);
    // bra rows with the same particle number (contiguous in the sorted order)
    int r0 = -1, r1 = -1;
    for (int r = 0; r < nb2; ++r)
      if (cnt[perm[r]] == kcnt) {
        if (r0 < 0) r0 = r;
        r1 = r + 1;
      }
    if (r0 >= 0) {
      if (nsec >= sector_cap) {
        tmf::set_error("tmf_site_prepare: more than %d charge sectors", sector_cap);
        return TMF_E_LIMIT;
      }
      // order of the minors: occupied S-columns of the first ket row
      int n = 0;
      if (fast) n = __builtin_popcountll(occ_word(map_k, K.sets + 2 * (size_t)c0, 0));
      else
        for (int i = 0; i < sk; ++i) n += occupied(K, scols[i], c0, 0);
      const int64_t need = (int64_t)(r1 - r0 + c1 - c0) * n;
      if (idx_used + need > idx_cap) {
        tmf::set_error("tmf_site_prepare: index pool too small (%lld > %lld)", (long long)(idx_used + need),
                       (long long)idx_cap);
        return TMF_E_LIMIT;
      }
      tmf_sector& S = sectors[nsec++];
      S.q = q_k[c0];
      S.r0 = r0, S.r1 = r1, S.c0 = c0, S.c1 = c1, S.n = n;
      S.bra_off = idx_used;
      for (int r = r0; r < r1; ++r) {
        int m = 0;
        if (fast) {
          uint64_t wd = occ_word(map_b, B.sets + 2 * (size_t)bra_alpha[r], bra_p[r]);
          m = __builtin_popcountll(wd);
          if (m == n)
            for (int q = 0; q < n; ++q, wd &= wd - 1) idx_pool[idx_used + q] = (uint8_t)__builtin_ctzll(wd);
        } else {
          for (int i = 0; i < sb; ++i)
            if (occupied(B, srows[i], bra_alpha[r], bra_p[r])) {
              if (m < n) idx_pool[idx_used + m] = (uint8_t)i;
              ++m;
            }
        }
        if (m != n) {  // slater.py:847-855
          tmf::set_error("tmf_site_prepare: bra row %d has %d particles in the minor, ket sector has %d", r, m, n);
          return TMF_E_ARG;
        }
        idx_used += n;
      }
      S.ket_off = idx_used;
      for (int c = c0; c < c1; ++c) {
        int m = 0;
        if (fast) {
          uint64_t wd = occ_word(map_k, K.sets + 2 * (size_t)c, 0);
          m = __builtin_popcountll(wd);
          if (m == n)
            for (int q = 0; q < n; ++q, wd &= wd - 1) idx_pool[idx_used + q] = (uint8_t)__builtin_ctzll(wd);
        } else {
          for (int i = 0; i < sk; ++i)
            if (occupied(K, scols[i], c, 0)) {
              if (m < n) idx_pool[idx_used + m] = (uint8_t)i;
              ++m;
            }
        }
        if (m != n) {
          tmf::set_error("tmf_site_prepare: ket row %d has %d particles in the minor, sector has %d", c, m, n);
          return TMF_E_ARG;
        }
        idx_used += n;
      }
      S.out_off = out_elems;
      out_elems += (int64_t)(r1 - r0) * (c1 - c0);
    }
    c0 = c1;
  }
  out->mb = mb, out->mk = mk, out->k_always = k, out->sb = sb, out->sk = sk, out->n_sectors = nsec;
  out->idx_bytes = idx_used;
  out->out_elems = out_elems;
  return TMF_OK;
}

// -------------------------------------------------------------------------------------------
// Batched, multi-threaded variants: one call for all cuts / all sites of a sweep.
// -------------------------------------------------------------------------------------------
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace {

// Worker threads of the host phase, kept between calls: a site range of a multi-GPU conversion has ~40 cuts, i.e. ~0.35 ms of
// enumeration on 16 threads, and starting 15 threads per call cost more than that (measured: 1.1 ms per call, on the
// critical path of a shard).  One job at a time (callers are serialised by the job mutex); the caller works too.
class HostPool {
 public:
  static HostPool& get() {
    static HostPool* p = new HostPool();   // never destroyed: workers may outlive static destructors at exit
    return *p;
  }
  // runs fn(worker_index) on `want` threads (the caller is one of them)
  void run(int want, const std::function<void()>& fn) {
    std::lock_guard<std::mutex> job(job_mutex_);
    const int helpers = want - 1;
    grow(helpers);
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      limit_ = helpers;
      pending_ = helpers;
      ++generation_;
    }
    if (helpers > 0) cv_.notify_all();
    fn();
    if (helpers > 0) {
      std::unique_lock<std::mutex> lk(m_);
      done_.wait(lk, [&] { return pending_ == 0; });
    }
    fn_ = nullptr;
  }

 private:
  void grow(int helpers) {
    std::lock_guard<std::mutex> lk(m_);
    while ((int)threads_.size() < helpers) {
      const int id = (int)threads_.size();
      threads_.emplace_back([this, id] { loop(id); });
      threads_.back().detach();
    }
  }
  void loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void()>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (id < limit_) fn = fn_;
      }
      if (fn) {
        (*fn)();
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::mutex job_mutex_, m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;
  const std::function<void()>* fn_ = nullptr;
  uint64_t generation_ = 0;
  int limit_ = 0, pending_ = 0;
};

template <typename F>
int parallel_for(int n, int nthreads, F&& fn) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > n) nthreads = n > 0 ? n : 1;
  if (nthreads > 256) nthreads = 256;
  std::atomic<int> next(0), err(0);
  // The error text is thread-local (tmf_last_error): the first failing worker's message is copied out under
  // the same compare-exchange that records its status, and re-set on the CALLING thread before returning.
  char first_msg[512] = "";
  const std::function<void()> work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      const int st = fn(i);
      if (st != 0) {
        int z = 0;
        if (err.compare_exchange_strong(z, st)) snprintf(first_msg, sizeof(first_msg), "item %d: %s", i, tmf_last_error());
      }
    }
  };
  if (nthreads == 1) work();
  else HostPool::get().run(nthreads, work);
  if (err.load() != 0) tmf::set_error("%s", first_msg);
  return err.load();
}
}  // namespace

// generic form for the host drivers (csrc/sweep.cpp): fn(i, arg) for i in [0, n) on the library's worker threads
extern "C" int tmf_host_parallel_for(int n, int nthreads, int (*fn)(int, void*), void* arg) {
  return parallel_for(n, nthreads, [&](int i) { return fn(i, arg); });
}

extern "C" int tmf_cut_vectors_batch(int ncuts, const double* e_pool, const int64_t* e_off, const int32_t* k,
                                     const int32_t* filled_left, int64_t chi_max, double svd_min,
                                     double degeneracy_tol, const int64_t* sectors, int n_sectors, int64_t cap,
                                     uint64_t* sets, double* lam_raw, int32_t* q_left, int64_t* chi,
                                     int64_t* n_checked, int nthreads) {
  return parallel_for(ncuts, nthreads, [&](int i) {
    return tmf_cut_vectors(e_pool + e_off[i], k[i], filled_left[i], chi_max, svd_min, degeneracy_tol, sectors,
                           n_sectors, cap, sets + (size_t)i * cap * 2, lam_raw + (size_t)i * cap,
                           q_left + (size_t)i * cap, chi + i, n_checked + i);
  });
}

extern "C" int tmf_site_prepare_batch(int nsites, const tmf_site_job* jobs, const uint64_t* sets,
                                      const int32_t* q_left, const int64_t* chi, int64_t cap, int32_t* row_sel,
                                      int8_t* row_sign, int32_t* col_sel, int8_t* col_sign, int32_t* bra_p,
                                      int32_t* bra_alpha, tmf_sector* sectors, uint8_t* idx_pool,
                                      tmf_site_out* outs, int nthreads) {
  return parallel_for(nsites, nthreads, [&](int i) {
    const tmf_site_job& j = jobs[i];
    tmf_site_in in;
    in.mode = j.mode;
    in.k_b = j.k_b, in.nf_b = j.nf_b, in.chi_b = (int32_t)chi[j.cut_b];
    in.k_k = j.k_k, in.nf_k = j.nf_k, in.chi_k = (int32_t)chi[j.cut_k];
    in.pad = 0;
    return tmf_site_prepare(&in, sets + (size_t)j.cut_b * cap * 2, q_left + (size_t)j.cut_b * cap,
                            sets + (size_t)j.cut_k * cap * 2, q_left + (size_t)j.cut_k * cap, row_sel + j.row_off,
                            row_sign + j.row_off, col_sel + j.col_off, col_sign + j.col_off, bra_p + j.bra_off,
                            bra_alpha + j.bra_off, sectors + j.sec_off, j.sec_cap, idx_pool + j.idx_off, j.idx_cap,
                            outs + i);
  });
}


// -------------------------------------------------------------------------------------------
// Tile descriptors of the pivoted-exchange determinant kernel (det_ppt.hip) for all sites of a sweep:
// one tile = one charge sector (or a range of its bra rows when the sector has more than
// `pairs_per_tile` pairs).  Sectors the kernel does not take (order 0 or > 32, more than 64 rows or
// columns) are listed in `rest` for the caller's general path.  Tiles are returned largest first.
// -------------------------------------------------------------------------------------------
extern "C" int64_t tmf_det_tiles_build(int nsites, const tmf_site_job* jobs, const tmf_site_out* outs,
                                       const tmf_sector* sectors, const tmf_det_site* sites, int elem_bytes,
                                       int64_t pairs_per_tile, tmf_det_desc* tiles, int64_t tile_cap, int64_t* rest,
                                       int64_t rest_cap, int64_t* n_rest, int32_t* lds_max, double* flops_n3,
                                       int64_t* n_pairs) {
  struct Ref {
    int site, sec;
    int32_t a0, a1;
    int64_t pairs;
  };
  std::vector<Ref> refs;
  int64_t nr = 0, pairs_tot = 0;
  double fl = 0.0;
  int lmax = 0;
  auto a16 = [](int64_t x) { return (x + 15) & ~(int64_t)15; };
  for (int i = 0; i < nsites; ++i) {
    const tmf_site_out& o = outs[i];
    for (int q = 0; q < o.n_sectors; ++q) {
      const tmf_sector& sc_ = sectors[jobs[i].sec_off + q];
      const int64_t nsb = sc_.r1 - sc_.r0, nsk = sc_.c1 - sc_.c0;
      if (sc_.n < 1 || sc_.n > 32 || o.sb > 64 || o.sk > 64) {
        if (nr < rest_cap) rest[nr] = ((int64_t)i << 32) | (int64_t)q;
        ++nr;
        continue;
      }
      int64_t ta = (pairs_per_tile + nsk - 1) / nsk;
      if (ta < 1) ta = 1;
      if (ta > nsb) ta = nsb;
      for (int64_t a0 = 0; a0 < nsb; a0 += ta) {
        const int64_t a1 = std::min(nsb, a0 + ta);
        refs.push_back(Ref{i, q, (int32_t)a0, (int32_t)a1, (a1 - a0) * nsk});
        const int64_t scr = std::max<int64_t>(264, (int64_t)sc_.n * sc_.n);
        // (+ the class tables of the pair phase: 12 bytes per bra set, 3 per ket set; layout in csrc/det_ppt.hip)
        const int64_t lds = a16(a16((int64_t)o.sb * o.sk * elem_bytes) + (nsk + (a1 - a0)) * 8) + 4 * (scr * elem_bytes + 288) +
                            a16(12 * (a1 - a0) + 3 * nsk + 16);
        lmax = std::max<int64_t>(lmax, lds);
      }
      pairs_tot += nsb * nsk;
      fl += (double)(nsb * nsk) * (double)sc_.n * sc_.n * sc_.n;
    }
  }
  *n_rest = nr;
  *lds_max = lmax;
  *flops_n3 = fl;
  *n_pairs = pairs_tot;
  const int64_t nt = (int64_t)refs.size();
  if (!tiles || nt > tile_cap || nr > rest_cap) return nt;   // counting pass (or buffers too small)
  std::stable_sort(refs.begin(), refs.end(), [](const Ref& x, const Ref& y) { return x.pairs > y.pairs; });
  for (int64_t t = 0; t < nt; ++t) {
    const Ref& r = refs[t];
    const tmf_site_out& o = outs[r.site];
    const tmf_sector& sc_ = sectors[jobs[r.site].sec_off + r.sec];
    const tmf_det_site& st = sites[r.site];
    tmf_det_desc& d = tiles[t];
    d.S = st.S, d.scale = st.scale;
    d.bra_idx = st.idx_base + (uint64_t)sc_.bra_off;
    d.ket_idx = st.idx_base + (uint64_t)sc_.ket_off;
    d.out = st.out_base + (uint64_t)sc_.out_off * (uint64_t)elem_bytes;
    d.sb = o.sb, d.sk = o.sk, d.lds = st.lds, d.n = sc_.n;
    d.nsb = sc_.r1 - sc_.r0, d.nsk = sc_.c1 - sc_.c0, d.a0 = r.a0, d.a1 = r.a1;
  }
  return nt;
}
