// Sweep level of the Slater -> MPS conversion: the host driver in C++.
//
// The reference walks the chain site by site (slater.py:1300-1346) and calls LAPACK per cut.  All L+1 cuts
// and all L sites are independent given C, so every stage here is ONE batched launch over all of them
// (descriptor arrays, variable sizes) with two host round trips: the entangled eigenvalues come down for
// the best-first enumeration (integer work on host threads, overlapped with the filled-basis launches), the
// index lists go up for the determinant stage.
//
// Per cut side (block A = C_LL or C_RR, off-diagonal block F = C_LR or C_RL), using that C is a projector
// (A - A^2 = F F^H: entangled orbitals = left singular vectors of F with sigma^2 = e (1 - e), the same set
// as slater.py:350):
//   E1  Y = F Omega                 running sums over the nested blocks of all cuts (nested.hip)
//   E2  Q = qr(Y)                   Householder slab kernel (or blocked Gram-Schmidt)
//   E3  B^H = F^H Q ; R^H           GEMM + slab QR (R only)
//   E4  R^H = U diag(sigma) V^H     one-sided Jacobi in LDS, left vectors only
//   E5  U0 = Q U[:, sigma^2 >= thr] GEMM
//   E6  T = U0^H A U0 ; T X = X e   2 GEMMs + Jacobi (Rayleigh-Ritz)
//   E7  U_E = U0 X                  GEMM
//   F   filled basis: orthonormalise (1 - U_E U_E^H) A Omega_f ; centre right orbitals = C_RL v_L
// Per site:
//   S1  O = V_bra^H V_ket           MFMA GEMM (slater.py:1071)
//   S2  W = signed gather of O      [always | sometimes] + physical orbital
//   S3  det_always, Schur complement (slater.py:1077-1090)
//   S4  all minors of all sectors   (slater.py:828-869)
// Self-check (testing.py:131-177): reconstruction deviations of the centre cut (recon.hip).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <functional>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include "common.hpp"

namespace tmf {
namespace {

using i64 = int64_t;
using u64 = uint64_t;

constexpr int PANEL_W = 16;
constexpr int N_STAGES = 16;
const char* kStageNames[N_STAGES] = {"upload",        "E_entangled",      "host_classify",   "F_filled",
                                     "S_overlap_gemm", "host_enumerate",   "host_site_prepare", "S_overlap_schur",
                                     "S_determinants", "host_result",      "download",        "total",
                                     "host_wait_eigenvalues", "host_wait_threads", "", ""};
enum Stage { ST_UPLOAD, ST_E, ST_CLASSIFY, ST_F, ST_S1, ST_ENUM, ST_SITEPREP, ST_SCHUR, ST_DET, ST_RESULT, ST_DOWNLOAD,
             ST_TOTAL, ST_WAIT_E, ST_WAIT_T };

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline i64 cdiv(i64 a, i64 b) { return (a + b - 1) / b; }
inline i64 a16(i64 x) { return (x + 15) & ~(i64)15; }

#define TMF_TRY(expr)            \
  do {                           \
    const int st__ = (expr);     \
    if (st__ != TMF_OK) return st__; \
  } while (0)
#define HIP_TRY(expr) TMF_TRY(check_hip((expr), #expr))

// ---- memory -----------------------------------------------------------------------------------------
// Temporaries of one sweep: bump allocation out of large device blocks that are kept between sweeps.
struct DeviceArena {
  struct Chunk {
    char* p;
    size_t cap, off;
  };
  std::vector<Chunk> chunks;
  size_t cur = 0, total = 0;
  int alloc(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (; cur < chunks.size(); ++cur) {
      Chunk& c = chunks[cur];
      if (c.off + bytes <= c.cap) {
        *out = c.p + c.off;
        c.off += bytes;
        return TMF_OK;
      }
    }
    Chunk c;
    c.cap = std::max(bytes, (size_t)1 << 30);
    c.off = bytes;
    HIP_TRY(hipMalloc((void**)&c.p, c.cap));
    total += c.cap;
    chunks.push_back(c);
    cur = chunks.size() - 1;
    *out = c.p;
    return TMF_OK;
  }
  void reset() {
    for (auto& c : chunks) c.off = 0;
    cur = 0;
  }
  void release() {
    for (auto& c : chunks) (void)hipFree(c.p);
    chunks.clear();
    cur = total = 0;
  }
};

// Page-locked staging memory with a mirrored device block: descriptor uploads are a host memcpy plus an
// asynchronous copy on the launch stream (a pageable copy blocks the host and drains the stream).
struct StagingArena {
  char *h = nullptr, *d = nullptr;
  size_t cap = 0, off = 0, flushed = 0;  // [flushed, off): written on the host, not yet copied to the device
  std::vector<std::pair<char*, char*>> retired;  // blocks still referenced by queued copies of this sweep
  int reserve(size_t bytes) {
    size_t o = (off + 255) & ~(size_t)255;
    if (h == nullptr || o + bytes > cap) {
      if (h != nullptr) retired.emplace_back(h, d);
      const size_t ncap = std::max(std::max(2 * cap, 2 * bytes), (size_t)64 << 20);
      HIP_TRY(hipHostMalloc((void**)&h, ncap, hipHostMallocDefault));
      HIP_TRY(hipMalloc((void**)&d, ncap));
      cap = ncap;
      o = 0;
      flushed = 0;
    }
    off = o;
    return TMF_OK;
  }
  void reset() {  // only after the launch stream has drained
    for (auto& r : retired) {
      (void)hipHostFree(r.first);
      (void)hipFree(r.second);
    }
    retired.clear();
    off = flushed = 0;
  }
  void release() {
    reset();
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
    h = d = nullptr;
    cap = off = 0;
  }
};

struct PinnedBuf {  // grow-only page-locked host buffer
  char* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return TMF_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t ncap = std::max(bytes + bytes / 4, (size_t)4096);
    HIP_TRY(hipHostMalloc((void**)&p, ncap, hipHostMallocDefault));
    memset(p, 0, ncap);  // fault the pages in now, not inside the worker threads
    cap = ncap;
    return TMF_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Output block of one conversion (tensors + det_always): stays alive until its download has finished, so
// two or three of them rotate while downloads overlap the next conversion.
struct OutSlot {
  char* d = nullptr;
  char* d_chk = nullptr;  // 8 doubles: deviations of the self-check
  size_t cap = 0;
  hipEvent_t done = nullptr;
  bool busy = false;      // a download is (or may be) running
  bool reserved = false;  // taken by the sweep in progress
  i64 ticket = -1;
  bool waited = true;     // tmf_sweep_wait has returned this ticket's self-check deviations
  PinnedBuf checks;  // 8 doubles
  int n_checks = 0;
  int det_off_elems = 0;
};

// Self-check deviations of a finished download whose output block is reused before anybody waited for it.
struct FinishedTicket {
  i64 ticket;
  int n;
  double v[8];
};

struct KernelEvent {
  hipEvent_t e0, e1;
  double flops;
  i64 n;
  int kind, order;
};

}  // namespace
}  // namespace tmf

using namespace tmf;

struct tmf_ctx {
  int device = 0;
  hipStream_t s_main = nullptr, s_down = nullptr, s_up = nullptr;
  // Two sets of per-sweep memory, used alternately: the next sweep is enqueued behind the running one without
  // waiting for its kernels (only for the sweep before that, which used the same set) - no idle GPU between
  // conversions when the caller downloads asynchronously.
  DeviceArena dev_set[2];
  StagingArena stage_set[2];
  std::vector<hipEvent_t> event_pool_set[2];
  hipEvent_t set_done[2] = {nullptr, nullptr};
  bool set_used[2] = {false, false};
  int cur = 0;
  std::vector<OutSlot> slots;
  std::vector<FinishedTicket> finished;   // the last few of them (ring)
  i64 next_ticket = 1;
  PinnedBuf fetch;     // eigenvalues / counts coming down
  PinnedBuf lu_stats;  // verdict of the block-local elimination, written by the kernel itself: 2 x {min pivot, max inverse, flag}
  PinnedBuf pool_pin;  // index lists going up
  PinnedBuf c_pin;     // C going up
  size_t events_used = 0;

  // ---- state of the current sweep ----
  tmf_sweep_params par{};
  std::vector<i64> sectors;
  bool have_sectors = false;
  int dtype = TMF_F64;
  i64 el = 8, L = 0, oc = 0, s_lo = 0, s_hi = 0;
  bool cplx = false, begun = false, have_sites = false, have_out = false;
  double cutoff = 0, thr2 = 0;
  std::vector<double> diag;
  char *d_Crm = nullptr, *d_C = nullptr;
  // cut-side problems
  i64 ncs = 0;
  std::vector<i64> cs_b, cs_side, n, m, ld1;
  std::vector<u64> blk, off;
  std::vector<char> doE;
  bool has_centre = false;
  i64 centre_L = -1, centre_R = -1;
  // entangled stage
  int P = 0;
  std::vector<i64> p, oS;
  std::vector<u64> UEp;
  std::vector<double> h_sig, h_e;
  std::vector<int32_t> h_cnt;
  int range_iterations = 0;
  double range_floor = 0;
  // classification
  std::vector<i64> k, nf, ent0;
  std::vector<std::vector<double>> e_side;
  i64 n_fermion = 0;
  // host phase outputs
  std::vector<i64> my_cuts, cpos, e_off, c_chi, c_chk;
  std::vector<int32_t> kk_cut, nfl, nfr;
  std::vector<double> e_pool, c_lam;
  std::vector<u64> c_sets;
  std::vector<int32_t> c_q;
  i64 cap = 0, ncut = 0, ns = 0;
  std::vector<tmf_site_job> jobs;
  std::vector<tmf_site_out> souts;
  std::vector<int32_t> row_sel, col_sel, bra_p, bra_alpha, mode;
  std::vector<int8_t> row_sign, col_sign;
  std::vector<tmf_sector> sec_buf;
  std::vector<i64> chi_b, chi_k, out_off, nsec;
  i64 out_tot = 0, sc_tot = 0, br_tot = 0, ix_tot = 0;
  // device results
  OutSlot* slot = nullptr;
  char *d_out = nullptr, *d_det = nullptr, *d_chk = nullptr;
  hipEvent_t chk_done = nullptr;   // the self-check of this sweep has run (it is launched on the upload stream)
  int n_checks = 0;
  std::vector<int32_t*> sweep_counters;  // device arrays of Jacobi sweep counts
  std::vector<i64> sweep_counts_n;
  double lu_min_pivot = 0.0, lu_max_inverse = 0.0, lu_inverse_cap = 100.0;
  i64 lu_fallbacks = 0;
  bool conditional = false;      // inside a tmf_launch_condition scope
  bool lu_pending[2] = {false, false};   // verdict of the sweep on memory set i not yet folded into the numbers above
  // timings
  double stage_ms[N_STAGES] = {0};
  double t_begin = 0;
  std::vector<KernelEvent> gemm_events, det_events;
  i64 n_det = 0;
};

namespace tmf {
namespace {

// Summary of the block-local elimination of the sweep that ran on memory set `set` (written by diag_verdict_kernel into
// page-locked memory; valid once that sweep has finished on the device).
static void fold_lu_verdict(tmf_ctx& c, int set) {
  if (!c.lu_pending[set] || c.lu_stats.p == nullptr) return;
  const double* v = (const double*)c.lu_stats.p + 3 * set;
  c.lu_min_pivot = v[0], c.lu_max_inverse = v[1];
  if (v[2] != 0.0) c.lu_fallbacks += 1;
  c.lu_pending[set] = false;
}

struct Sweep {
  tmf_ctx& c;
  explicit Sweep(tmf_ctx& ctx) : c(ctx) {}

  // ------------------------------------------------------------------ plumbing
  // Stream operations of a stage are DEFERRED: the helpers below build their descriptors, put them into the page-locked
  // staging block (`up`) and queue the launch; `run_deferred` copies everything staged so far in ONE transfer and then
  // issues the queued operations in order.  Issued one by one, every descriptor array was its own 5 us copy kernel on the
  // launch stream - 74 of them in the conversion of a 38-site shard, 0.37 ms of its critical path
  // (profiles/r02/shard_8way_rank3_kernel_stats.csv).
  std::vector<std::function<int()>> dq;
#define LATER(expr) dq.emplace_back([=]() -> int { return (expr); })
#define LATER_HIP(expr) dq.emplace_back([=]() -> int { return check_hip((expr), #expr); })
  int flush_uploads() {
    StagingArena& a = c.stage_set[c.cur];
    if (a.off > a.flushed) {
      HIP_TRY(hipMemcpyAsync(a.d + a.flushed, a.h + a.flushed, a.off - a.flushed, hipMemcpyHostToDevice, c.s_main));
      a.flushed = a.off;
    }
    return TMF_OK;
  }
  int run_deferred() {
    int st = flush_uploads();
    for (auto& f : dq) {
      if (st != TMF_OK) break;
      st = f();
    }
    dq.clear();
    return st;
  }
  int dalloc(i64 count, i64 elem, void** out, bool zero = false) {
    const size_t bytes = (size_t)std::max<i64>(count, 1) * (size_t)elem;
    TMF_TRY(c.dev_set[c.cur].alloc(bytes, out));
    if (zero) {
      void* p_ = *out;
      LATER_HIP(hipMemsetAsync(p_, 0, bytes, c.s_main));
    }
    return TMF_OK;
  }
  int alloc_el(i64 count, u64* out, bool zero = false) {  // elements of the sweep's dtype
    void* p = nullptr;
    TMF_TRY(dalloc(count, c.el, &p, zero));
    *out = (u64)p;
    return TMF_OK;
  }
  int up(const void* host, size_t bytes, u64* dev) {
    if (bytes == 0) bytes = 1;
    StagingArena& a = c.stage_set[c.cur];
    if (a.h == nullptr || ((a.off + 255) & ~(size_t)255) + bytes > a.cap) TMF_TRY(flush_uploads());   // (a new block follows)
    TMF_TRY(a.reserve(bytes));
    memcpy(a.h + a.off, host, bytes);
    *dev = (u64)(a.d + a.off);
    a.off += bytes;
    return TMF_OK;
  }
  template <typename T>
  int up_vec(const std::vector<T>& v, u64* dev) {
    return up(v.data(), v.size() * sizeof(T), dev);
  }
  int new_event(hipEvent_t* e) {
    std::vector<hipEvent_t>& pool = c.event_pool_set[c.cur];
    if (c.events_used == pool.size()) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      pool.push_back(ev);
    }
    *e = pool[c.events_used++];
    return TMF_OK;
  }
  void tick(int stage, double t0) { c.stage_ms[stage] += now_ms() - t0; }
  bool timing() const { return (c.par.flags & TMF_SWEEP_TIME_KERNELS) != 0; }

  // ------------------------------------------------------------------ batched ops
  struct Gemm {
    std::vector<tmf_gemm_desc> d;
    void add(u64 A, u64 B, u64 C, i64 M, i64 N, i64 K, i64 lda, i64 ldb, i64 ldc) {
      if (M <= 0 || N <= 0) return;
      tmf_gemm_desc g;
      g.A = A, g.B = B, g.C = C;
      g.M = (int32_t)M, g.N = (int32_t)N, g.K = (int32_t)K;
      g.lda = (int32_t)std::max<i64>(lda, 1), g.ldb = (int32_t)std::max<i64>(ldb, 1), g.ldc = (int32_t)ldc;
      d.push_back(g);
    }
  };
  // C = alpha op(A) B + beta C over a list of problems (tile table: longest contractions first)
  int gemm(int opA, double alpha, double beta, const Gemm& g, hipStream_t on = nullptr) {
    if (on == nullptr) on = c.s_main;
    const size_t np = g.d.size();
    if (np == 0) return TMF_OK;
    int maxN = 0;
    for (auto& x : g.d) maxN = std::max(maxN, x.N);
    const int tn = maxN <= 16 ? 16 : 64;
    struct Tile {
      int32_t prob, tm, tnn, z;
    };
    std::vector<Tile> tiles;
    for (size_t i = 0; i < np; ++i) {
      const i64 tm = cdiv(g.d[i].M, 64), tnn = cdiv(g.d[i].N, tn);
      for (i64 t = 0; t < tm * tnn; ++t) tiles.push_back(Tile{(int32_t)i, (int32_t)(t % tm), (int32_t)(t / tm), 0});
    }
    std::stable_sort(tiles.begin(), tiles.end(), [&](const Tile& a, const Tile& b) { return g.d[a.prob].K > g.d[b.prob].K; });
    u64 dd, dt;
    TMF_TRY(up(g.d.data(), np * sizeof(tmf_gemm_desc), &dd));
    TMF_TRY(up(tiles.data(), tiles.size() * sizeof(Tile), &dt));
    KernelEvent ev{};
    const bool timed = timing() && !c.conditional;   // launches of a conditional scope may return at once: not GEMM work
    if (timed) {
      TMF_TRY(new_event(&ev.e0));
      TMF_TRY(new_event(&ev.e1));
      LATER_HIP(hipEventRecord(ev.e0, on));
    }
    const int ntile = (int)tiles.size();
    LATER(tmf_gemm_batched(c.dtype, opA, alpha, beta, (const tmf_gemm_desc*)dd, (const int32_t*)dt, ntile, tn, on));
    if (timed) {
      LATER_HIP(hipEventRecord(ev.e1, on));
      double fl = 0;
      for (auto& x : g.d) fl += (double)x.M * x.N * x.K;
      ev.flops = fl * (c.cplx ? 8.0 : 2.0);
      ev.kind = tn == 16 ? 2 : (opA ? 1 : 0);
      c.gemm_events.push_back(ev);
    }
    return TMF_OK;
  }

  struct Slab {
    u64 base;
    i64 rows, ld, c0, c1;
    u64 scratch;
  };
  // blocked classical Gram-Schmidt with re-orthogonalisation (tmf_bcgs_batched)
  // `on`: the stream the kernels run on.  Descriptors (and the zeroing of the norms) always go through the launch
  // stream; another stream waits for them by event.
  int bcgs(std::vector<Slab> s, int passes, bool cholqr, bool wide = false, hipStream_t on = nullptr, bool fused = false) {
    if (on == nullptr) on = c.s_main;
    s.erase(std::remove_if(s.begin(), s.end(), [](const Slab& x) { return !(x.rows > 0 && x.c1 > x.c0); }), s.end());
    if (s.empty()) return TMF_OK;
    std::stable_sort(s.begin(), s.end(), [](const Slab& a, const Slab& b) { return a.rows > b.rows; });
    i64 span_tot = 0;
    for (auto& x : s) span_tot += x.c1 - x.c0;
    void* d_nrm;
    TMF_TRY(dalloc(span_tot, 8, &d_nrm, true));
    std::vector<tmf_norms_desc> nd(s.size());
    std::vector<tmf_bcgs_desc> bd(s.size());
    i64 noff = 0;
    for (size_t i = 0; i < s.size(); ++i) {
      const u64 nrm = (u64)d_nrm + 8 * noff;
      nd[i].src = s[i].base + (u64)(s[i].c0 * s[i].ld * c.el);
      nd[i].out = nrm;
      nd[i].n = (int32_t)s[i].rows, nd[i].c = (int32_t)(s[i].c1 - s[i].c0), nd[i].lds_ = (int32_t)s[i].ld, nd[i].pad = 0;
      bd[i].base = s[i].base, bd[i].scratch = s[i].scratch, bd[i].norms = nrm;
      bd[i].rows = (int32_t)s[i].rows, bd[i].ld = (int32_t)s[i].ld, bd[i].c_begin = (int32_t)s[i].c0, bd[i].c_end = (int32_t)s[i].c1;
      noff += s[i].c1 - s[i].c0;
    }
    u64 t_nd, t_bd;
    TMF_TRY(up_vec(nd, &t_nd));
    TMF_TRY(up_vec(bd, &t_bd));
    const i64 wb = tmf_bcgs_work_bytes(bd.data(), (int)bd.size());
    void* d_work;
    TMF_TRY(dalloc(wb, 1, &d_work));
    if (on != c.s_main) {
      hipEvent_t ev;
      TMF_TRY(new_event(&ev));
      LATER_HIP(hipEventRecord(ev, c.s_main));
      LATER_HIP(hipStreamWaitEvent(on, ev, 0));
    }
    const int nslab = (int)s.size();
    LATER(tmf_column_norms_batched(c.dtype, (const tmf_norms_desc*)t_nd, nslab, on));
    auto bdp = std::make_shared<std::vector<tmf_bcgs_desc>>(std::move(bd));   // host copy of the records, read at launch time
    const int bflags = (cholqr ? 1 : 0) | (wide ? 2 : 0) | (fused ? 4 : 0);
    LATER(tmf_bcgs_batched(c.dtype, (const tmf_bcgs_desc*)t_bd, bdp->data(), (int)bdp->size(), passes, bflags, d_work, wb, on));
    return TMF_OK;
  }

  struct HSlab {
    size_t idx;  // position in the caller's arrays
    u64 base;
    i64 rows, ld, cols;
    u64 r;
    i64 r_ld;
  };
  // Householder QR of tall slabs.  r_only: only R^H into s.r (slab destroyed).  Otherwise the orthonormal factor
  // is built in a scratch whose address is written to out[s.idx] (leading dimension = rows).
  int house_slab(std::vector<HSlab> s, bool r_only, std::vector<u64>* out) {
    s.erase(std::remove_if(s.begin(), s.end(), [](const HSlab& x) { return !(x.rows > 0 && x.cols > 0); }), s.end());
    if (s.empty()) return TMF_OK;
    std::stable_sort(s.begin(), s.end(), [](const HSlab& a, const HSlab& b) { return a.rows > b.rows; });
    std::vector<tmf_slab_desc> d(s.size());
    i64 maxn = 0, maxc = 0;
    for (auto& x : s) maxn = std::max(maxn, x.rows), maxc = std::max(maxc, x.cols);
    if (r_only) {
      for (size_t i = 0; i < s.size(); ++i) {
        d[i].A = s[i].base, d[i].Q = 0, d[i].R = s[i].r;
        d[i].n = (int32_t)s[i].rows, d[i].c = (int32_t)s[i].cols, d[i].lda = (int32_t)s[i].ld, d[i].ldq = 1;
        d[i].ldr = (int32_t)s[i].r_ld, d[i].flags = 1 | 4;
      }
    } else {
      i64 tot = 0;
      std::vector<i64> o(s.size());
      for (size_t i = 0; i < s.size(); ++i) {
        o[i] = tot;
        tot += (s[i].rows * s[i].cols + 1) & ~(i64)1;
      }
      u64 d_q;
      TMF_TRY(alloc_el(tot + 2, &d_q));
      for (size_t i = 0; i < s.size(); ++i) {
        d[i].A = s[i].base, d[i].Q = d_q + (u64)(o[i] * c.el), d[i].R = 0;
        d[i].n = (int32_t)s[i].rows, d[i].c = (int32_t)s[i].cols, d[i].lda = (int32_t)s[i].ld, d[i].ldq = (int32_t)s[i].rows;
        d[i].ldr = 1, d[i].flags = 2;
        (*out)[s[i].idx] = d[i].Q;
      }
    }
    u64 t_d;
    TMF_TRY(up_vec(d, &t_d));
    const int nd_ = (int)d.size();
    LATER(tmf_house_slab_batched(c.dtype, (const tmf_slab_desc*)t_d, nd_, (int)maxn, (int)maxc, c.s_main));
    return TMF_OK;
  }

  // Householder QR of slabs of any width in place (global-memory kernel, one workgroup per slab): the range finders wider
  // than the 64 columns of the slab kernel.  Orthogonal for any rank - the blocked Gram-Schmidt that used to run here lost
  // an entangled orbital on a spinful chain whose two species decouple exactly (rounding noise has no component outside
  // span(Q)): tests/soak/soak_small.py seed 30023, 80 instead of 81 orbitals at one cut, weak eigenvalues
  // off by 1 - 5 %.
  int house_general(const std::vector<Slab>& s_in) {
    // Slabs whose rows fit the panel kernel (1024 complex / 2048 real, LDS of 8 reflectors): factored there with the Q's
    // left for tmf_house_form_q_batched, which forms them in place (no scratch of the size of the slabs) - about a third of
    // the time of the global-memory kernel below at n x 256 (76 ms per launch for the 2 046 cuts of a range-12 chain).
    {
      i64 mm = 0, mn = 0, tau_tot = 0;
      for (const Slab& x : s_in)
        if (x.rows > 0 && x.c1 > 0) mm = std::max(mm, x.rows), mn = std::max(mn, x.c1), tau_tot += (x.c1 + 1) & ~(i64)1;
      if (mm == 0) return TMF_OK;
      const i64 row_cap = c.cplx ? 1024 : 2048;
      const size_t lds = ((size_t)mm * 8 + (size_t)mn + 4) * (size_t)c.el + 64;
      static const bool allow = !(getenv("TMF_WIDE_QR") && std::string(getenv("TMF_WIDE_QR")) == "global");
      if (allow && mm <= row_cap && lds <= 150 * 1024) {
        u64 d_tau;
        TMF_TRY(alloc_el(tau_tot + 2, &d_tau));
        std::vector<tmf_slab_desc> d;
        i64 o = 0;
        for (const Slab& x : s_in) {
          if (!(x.rows > 0 && x.c1 > 0)) continue;
          tmf_slab_desc q{};
          q.A = x.base, q.Q = d_tau + (u64)(o * c.el), q.R = 0;
          q.n = (int32_t)x.rows, q.c = (int32_t)x.c1, q.lda = (int32_t)x.ld, q.ldq = 1, q.ldr = 1, q.flags = 8;
          d.push_back(q);
          o += (x.c1 + 1) & ~(i64)1;
        }
        std::stable_sort(d.begin(), d.end(), [](const tmf_slab_desc& a, const tmf_slab_desc& b) { return (i64)a.n * a.c > (i64)b.n * b.c; });
        u64 t_d;
        TMF_TRY(up_vec(d, &t_d));
        const int nd_ = (int)d.size(), mm_ = (int)mm, mn_ = (int)mn;
        LATER(tmf_house_slab_batched(c.dtype, (const tmf_slab_desc*)t_d, nd_, mm_, mn_, c.s_main));
        LATER(tmf_house_form_q_batched(c.dtype, (const tmf_slab_desc*)t_d, nd_, mm_, mn_, c.s_main));
        return TMF_OK;
      }
    }
    std::vector<tmf_qr_desc> d;
    i64 maxm = 0, maxn = 0;
    for (const Slab& x : s_in) {
      if (!(x.rows > 0 && x.c1 > 0)) continue;
      tmf_qr_desc q{};
      q.A = x.base, q.R = 0, q.m = (int32_t)x.rows, q.n = (int32_t)x.c1, q.lda = (int32_t)x.ld, q.ldr = 1, q.flags = 0;
      d.push_back(q);
      maxm = std::max(maxm, x.rows), maxn = std::max(maxn, x.c1);
    }
    if (d.empty()) return TMF_OK;
    std::stable_sort(d.begin(), d.end(), [](const tmf_qr_desc& a, const tmf_qr_desc& b) { return (i64)a.m * a.n > (i64)b.m * b.n; });
    u64 t_d;
    TMF_TRY(up_vec(d, &t_d));
    const int nd_ = (int)d.size();
    LATER(tmf_house_qr_batched(c.dtype, (const tmf_qr_desc*)t_d, nd_, (int)maxm, (int)maxn, c.s_main));
    return TMF_OK;
  }

  // One-sided Jacobi per problem (p > 0 only).  left_only: U receives the normalised left singular vectors.
  int jacobi(const std::vector<u64>& X, const std::vector<u64>& V, const std::vector<u64>& s, const std::vector<u64>* count,
             double thresh2, const std::vector<i64>& p, bool left_only) {
    std::vector<tmf_jacobi_desc> d;
    i64 maxp = 0;
    for (size_t i = 0; i < p.size(); ++i) maxp = std::max(maxp, p[i]);
    if (maxp == 0) return TMF_OK;
    const bool big = maxp > 64;
    const bool outU = left_only || big;
    std::vector<i64> pp;
    for (size_t i = 0; i < p.size(); ++i) {
      if (p[i] <= 0) continue;
      tmf_jacobi_desc j{};
      j.X = X[i];
      (outU ? j.U : j.V) = V[i];
      j.s = s[i];
      j.count = count ? (*count)[i] : 0;
      j.thresh2 = thresh2;
      j.p = (int32_t)p[i], j.ldx = (int32_t)std::max<i64>(p[i], 1);
      j.ldv = outU ? 0 : (int32_t)std::max<i64>(p[i], 1);
      j.ldu = outU ? (int32_t)std::max<i64>(p[i], 1) : 1;
      d.push_back(j);
      pp.push_back(p[i]);
    }
    void* d_sw;
    TMF_TRY(dalloc((i64)d.size(), 4, &d_sw, true));
    c.sweep_counters.push_back((int32_t*)d_sw);
    c.sweep_counts_n.push_back((i64)d.size());
    if (big) {
      if (!left_only) {  // workspace for the accumulated rotations
        i64 tot = 0;
        for (auto q : pp) tot += q * q;
        u64 d_ws;
        TMF_TRY(alloc_el(tot, &d_ws));
        i64 o = 0;
        for (size_t i = 0; i < d.size(); ++i) {
          d[i].V = d_ws + (u64)(o * c.el), d[i].ldv = (int32_t)pp[i];
          o += pp[i] * pp[i];
        }
      }
      u64 dd;
      TMF_TRY(up_vec(d, &dd));
      const int nd_ = (int)d.size();
      LATER(tmf_jacobi_block_batched(c.dtype, left_only ? 0 : 1, (const tmf_jacobi_desc*)dd, nd_, (int)maxp, (int32_t*)d_sw, c.s_main));
      return TMF_OK;
    }
    u64 dd;
    TMF_TRY(up_vec(d, &dd));
    const int nd_ = (int)d.size();
    if (left_only) LATER(tmf_svd_left_batched(c.dtype, (const tmf_jacobi_desc*)dd, nd_, (int)maxp, (int32_t*)d_sw, c.s_main));
    else LATER(tmf_jacobi_batched(c.dtype, (const tmf_jacobi_desc*)dd, nd_, (int)maxp, (int32_t*)d_sw, c.s_main));
    return TMF_OK;
  }

  // Y_i = A_i Omega (kind 'A') or F_i Omega (kind 'F') for all cut sides at once through running sums
  int nested(char kind, u64 Omp, i64 ldo, const std::vector<u64>& dest, const std::vector<i64>& ncol) {
    const i64 D = c.L;
    tmf_nested_desc descs[2];
    int nd = 0, maxc_all = 0;
    for (int sd = 0; sd < 2; ++sd) {
      std::vector<u64> tab_d(D + 1, 0);
      std::vector<int32_t> tab_n(D + 1, 0), tab_l(D + 1, 1);
      i64 xlo = D + 1, xhi = -1, maxc = 0;
      for (i64 i = 0; i < c.ncs; ++i) {
        if (c.cs_side[i] != sd || ncol[i] <= 0) continue;
        const i64 x = c.cs_b[i];
        tab_d[x] = dest[i], tab_n[x] = (int32_t)ncol[i], tab_l[x] = (int32_t)c.ld1[i];
        xlo = std::min(xlo, x), xhi = std::max(xhi, x), maxc = std::max(maxc, ncol[i]);
      }
      if (xhi < 0) continue;
      u64 t_d, t_n, t_l;
      TMF_TRY(up_vec(tab_d, &t_d));
      TMF_TRY(up_vec(tab_n, &t_n));
      TMF_TRY(up_vec(tab_l, &t_l));
      const bool suffix = (kind == 'A') ? (sd == 1) : (sd == 0);
      tmf_nested_desc& q = descs[nd++];
      q.C = (u64)c.d_C, q.Omega = Omp, q.dest = t_d, q.ncol = t_n, q.ld = t_l;
      q.D = (int32_t)D, q.ldc = (int32_t)D, q.ldo = (int32_t)ldo, q.suffix = suffix ? 1 : 0, q.rows_ge = sd;
      q.x_lo = (int32_t)xlo, q.x_hi = (int32_t)xhi, q.maxc = (int32_t)maxc;
      maxc_all = std::max<int>(maxc_all, (int)maxc);
    }
    if (nd == 0) return TMF_OK;
    u64 t_desc;
    TMF_TRY(up(descs, nd * sizeof(tmf_nested_desc), &t_desc));
    LATER(tmf_nested_products_batched(c.dtype, (const tmf_nested_desc*)t_desc, nd, (int)D, maxc_all, c.s_main));
    return TMF_OK;
  }

  // normalised column copies (tmf_normalise_columns_batched)
  int colcopy(std::vector<tmf_colnorm_desc>& d) {
    d.erase(std::remove_if(d.begin(), d.end(), [](const tmf_colnorm_desc& x) { return !(x.n > 0 && x.c > 0); }), d.end());
    if (d.empty()) return TMF_OK;
    u64 dd;
    TMF_TRY(up_vec(d, &dd));
    const int nd_ = (int)d.size();
    LATER(tmf_normalise_columns_batched(c.dtype, (const tmf_colnorm_desc*)dd, nd_, c.s_main));
    return TMF_OK;
  }

  // ------------------------------------------------------------------ begin
  int begin(const void* C, const tmf_sweep_params* par) {
    HIP_TRY(hipSetDevice(c.device));
    c.t_begin = now_ms();
    memset(c.stage_ms, 0, sizeof(c.stage_ms));
    // the other memory set: free once the sweep before the previous one has finished on the GPU
    c.cur ^= 1;
    if (c.set_used[c.cur]) HIP_TRY(hipEventSynchronize(c.set_done[c.cur]));
    fold_lu_verdict(c, c.cur);
    c.dev_set[c.cur].reset();
    c.stage_set[c.cur].reset();
    c.events_used = 0;
    c.gemm_events.clear();
    c.det_events.clear();
    c.sweep_counters.clear();
    c.sweep_counts_n.clear();
    c.begun = c.have_sites = c.have_out = false;
    c.chk_done = nullptr;
    if (c.slot) c.slot->reserved = false;  // (a sweep that was never downloaded)
    c.slot = nullptr;
    c.par = *par;
    c.have_sectors = par->sectors != nullptr;
    c.sectors.clear();
    if (c.have_sectors) c.sectors.assign(par->sectors, par->sectors + par->n_sectors);
    c.sectors.push_back(0);  // never empty: data() stays a valid non-null pointer for "no sector allowed"
    c.par.sectors = nullptr;
    const i64 L = par->L;
    if (L <= 0 || par->site_lo < 0 || par->site_hi > L || par->site_lo >= par->site_hi || par->ortho_center < 0 ||
        par->ortho_center > L || !(par->svd_min > 0 && par->svd_min < 1) || !(par->degeneracy_tol > 0)) {
      set_error("tmf_sweep_begin: bad parameters (L = %lld, sites [%lld, %lld), centre %lld)", (long long)L,
                (long long)par->site_lo, (long long)par->site_hi, (long long)par->ortho_center);
      return TMF_E_ARG;
    }
    c.L = L, c.oc = par->ortho_center, c.s_lo = par->site_lo, c.s_hi = par->site_hi;
    c.cplx = par->is_complex != 0;
    c.dtype = c.cplx ? TMF_C128 : TMF_F64;
    c.el = c.cplx ? 16 : 8;
    c.cutoff = par->svd_min * par->svd_min;  // slater.py:318
    c.thr2 = c.cutoff * (1.0 - c.cutoff);

    double t0 = now_ms();
    const size_t cbytes = (size_t)L * L * c.el;
    void* p;
    TMF_TRY(c.dev_set[c.cur].alloc(cbytes, &p));
    c.d_C = (char*)p;
    c.diag.assign(L, 0.0);
    if (par->flags & TMF_SWEEP_C_ON_DEVICE) {
      c.d_Crm = (char*)C;
      // the diagonal (particle counts per cut) comes down with a strided copy
      TMF_TRY(c.fetch.ensure((size_t)L * 16));
      // (on the upload stream: the launch stream may still be busy with the previous sweep)
      HIP_TRY(hipMemcpy2DAsync(c.fetch.p, c.el, C, (size_t)(L + 1) * c.el, c.el, L, hipMemcpyDeviceToHost, c.s_up));
      HIP_TRY(hipStreamSynchronize(c.s_up));
      for (i64 i = 0; i < L; ++i) c.diag[i] = ((const double*)c.fetch.p)[i * (c.cplx ? 2 : 1)];
    } else {
      TMF_TRY(c.dev_set[c.cur].alloc(cbytes, &p));
      c.d_Crm = (char*)p;
      TMF_TRY(c.c_pin.ensure(cbytes));
      memcpy(c.c_pin.p, C, cbytes);
      HIP_TRY(hipMemcpyAsync(c.d_Crm, c.c_pin.p, cbytes, hipMemcpyHostToDevice, c.s_main));
      const double* h = (const double*)C;
      for (i64 i = 0; i < L; ++i) c.diag[i] = h[(size_t)i * (L + 1) * (c.cplx ? 2 : 1)];
    }
    TMF_TRY(tmf_transpose(c.dtype, c.d_Crm, c.d_C, (int)L, c.s_main));
    tick(ST_UPLOAD, t0);

    // ---- cut-side problems ----
    std::vector<char> need((size_t)(L + 1) * 2, 0);
    const i64 oc = c.oc;
    for (i64 i = c.s_lo; i < c.s_hi; ++i) {
      const int sd = i < oc ? 0 : 1;
      need[(size_t)i * 2 + sd] = 1, need[(size_t)(i + 1) * 2 + sd] = 1;
    }
    if (need[(size_t)oc * 2 + 1] || (c.s_lo <= oc && oc <= c.s_hi))
      need[(size_t)oc * 2 + 0] = need[(size_t)oc * 2 + 1] = 1;  // the centre's right orbitals are paired with its left ones
    c.cs_b.clear(), c.cs_side.clear();
    for (i64 b = 0; b <= L; ++b)
      for (int sd = 0; sd < 2; ++sd)
        if (need[(size_t)b * 2 + sd]) c.cs_b.push_back(b), c.cs_side.push_back(sd);
    c.ncs = (i64)c.cs_b.size();
    c.n.resize(c.ncs), c.m.resize(c.ncs), c.ld1.resize(c.ncs), c.blk.resize(c.ncs), c.off.resize(c.ncs), c.doE.resize(c.ncs);
    c.has_centre = need[(size_t)oc * 2 + 0] != 0;
    c.centre_L = c.centre_R = -1;
    for (i64 i = 0; i < c.ncs; ++i) {
      const i64 b = c.cs_b[i];
      const bool left = c.cs_side[i] == 0;
      c.n[i] = left ? b : L - b;
      c.m[i] = L - c.n[i];
      c.ld1[i] = std::max<i64>(c.n[i], 1);
      c.blk[i] = (u64)c.d_C + (u64)((left ? 0 : (b + b * L)) * c.el);  // A = C_LL or C_RR
      c.off[i] = (u64)c.d_C + (u64)((left ? b * L : b) * c.el);        // F (n x m)
      c.doE[i] = (c.n[i] > 0 && c.m[i] > 0) ? 1 : 0;
      if (c.has_centre && b == oc) (left ? c.centre_L : c.centre_R) = i;
    }
    // the right side of the centre cut is paired with the left side through C_RL (block_svd, slater.py:407)
    if (c.has_centre) c.doE[c.centre_R] = 0;
    c.begun = true;
    return TMF_OK;
  }

  // ------------------------------------------------------------------ entangled stage (E1-E7)
  int entangled(int P, int iterations, double* worst_out, int32_t* sat_out, int32_t* weak_out, int32_t* sweeps_out,
                i64* bad_cut) {
    if (!c.begun) {
      set_error("tmf_sweep_entangled: no sweep in progress");
      return TMF_E_ARG;
    }
    const double t0 = now_ms();
    const i64 L = c.L, el = c.el, ncs = c.ncs;
    c.P = P;
    u64 d_Om;
    TMF_TRY(alloc_el(L * P, &d_Om));
    LATER(tmf_fill_normal(c.dtype, (void*)d_Om, L * P, 0x5EED1, c.s_main));
    std::vector<i64>& p = c.p;
    p.assign(ncs, 0);
    std::vector<char> full(ncs, 0);
    for (i64 i = 0; i < ncs; ++i) {
      const i64 mn = std::min(c.n[i], c.m[i]);
      p[i] = c.doE[i] ? std::min<i64>(P, mn) : 0;
      full[i] = (c.doE[i] && p[i] == P && P < mn) ? 1 : 0;  // cuts the range finder truncates
    }
    std::vector<i64> oY(ncs), oB(ncs), oR(ncs);
    c.oS.assign(ncs, 0);
    i64 tY = 0, tB = 0, tR = 0, tS = 0;
    for (i64 i = 0; i < ncs; ++i) {
      oY[i] = tY, oB[i] = tB, oR[i] = tR, c.oS[i] = tS;
      tY += c.n[i] * p[i], tB += c.m[i] * p[i], tR += p[i] * p[i], tS += p[i];
    }
    u64 d_Y, d_U0, d_W1, d_Bt, d_Q2, d_R, d_Z, d_T, d_X, d_scr;
    TMF_TRY(alloc_el(tY, &d_Y)); TMF_TRY(alloc_el(tY, &d_U0)); TMF_TRY(alloc_el(tY, &d_W1));
    TMF_TRY(alloc_el(tB, &d_Bt)); TMF_TRY(alloc_el(tB, &d_Q2));
    TMF_TRY(alloc_el(tR, &d_R)); TMF_TRY(alloc_el(tR, &d_Z)); TMF_TRY(alloc_el(tR, &d_T)); TMF_TRY(alloc_el(tR, &d_X));
    void *d_sig, *d_e, *d_cnt;
    TMF_TRY(dalloc(tS, 8, &d_sig, true)); TMF_TRY(dalloc(tS, 8, &d_e, true)); TMF_TRY(dalloc(ncs, 4, &d_cnt, true));
    TMF_TRY(alloc_el(ncs * P * PANEL_W, &d_scr));
    std::vector<u64> Yp(ncs), U0p(ncs), W1p(ncs), Btp(ncs), Q2p(ncs), Rp(ncs), Zp(ncs), Tp(ncs), Xp(ncs), sigp(ncs), ep(ncs),
        cntp(ncs), scrp(ncs), omp(ncs);
    for (i64 i = 0; i < ncs; ++i) {
      Yp[i] = d_Y + (u64)(oY[i] * el), U0p[i] = d_U0 + (u64)(oY[i] * el), W1p[i] = d_W1 + (u64)(oY[i] * el);
      Btp[i] = d_Bt + (u64)(oB[i] * el), Q2p[i] = d_Q2 + (u64)(oB[i] * el);
      Rp[i] = d_R + (u64)(oR[i] * el), Zp[i] = d_Z + (u64)(oR[i] * el), Tp[i] = d_T + (u64)(oR[i] * el), Xp[i] = d_X + (u64)(oR[i] * el);
      sigp[i] = (u64)d_sig + (u64)(c.oS[i] * 8), ep[i] = (u64)d_e + (u64)(c.oS[i] * 8);
      cntp[i] = (u64)d_cnt + (u64)(i * 4);
      scrp[i] = d_scr + (u64)(i * P * PANEL_W * el);
      omp[i] = d_Om + (u64)((c.cs_side[i] == 0 ? c.cs_b[i] : 0) * el);  // rows of Omega on the other side
    }
    // E1: Y = F Omega
    TMF_TRY(nested('F', d_Om, L, Yp, p));
    // E2: Q = qr(Y)
    const i64 slab_rows = c.cplx ? 1024 : 2048;  // row limit of the slab kernel (registers)
    i64 maxp = 0, maxm = 0;
    for (i64 i = 0; i < ncs; ++i)
      if (c.doE[i]) maxp = std::max(maxp, p[i]), maxm = std::max(maxm, c.m[i]);
    const bool house_ok = !(c.par.flags & TMF_SWEEP_RANGE_BCGS) && maxp <= 64;
    const bool house = house_ok && maxm <= slab_rows;  // the longer slabs: F^H Q is m x p
    auto rqr = [&](std::vector<u64>& ptr, const std::vector<i64>& rows) -> int {
      i64 maxr = 0;
      for (i64 i = 0; i < ncs; ++i)
        if (c.doE[i]) maxr = std::max(maxr, rows[i]);
      if (house_ok && maxr <= slab_rows) {  // Q stays in the buffer the kernel builds it in (no copy back)
        std::vector<HSlab> s;
        for (i64 i = 0; i < ncs; ++i)
          if (c.doE[i]) s.push_back(HSlab{(size_t)i, ptr[i], rows[i], rows[i], p[i], 0, 0});
        return house_slab(s, false, &ptr);
      }
      std::vector<Slab> s;
      for (i64 i = 0; i < ncs; ++i)
        if (c.doE[i]) s.push_back(Slab{ptr[i], rows[i], rows[i], 0, p[i], scrp[i]});
      if (!(c.par.flags & TMF_SWEEP_RANGE_BCGS)) return house_general(s);
      return bcgs(s, 3, false);
    };
    TMF_TRY(rqr(Yp, c.n));
    // E3: B^H = F^H Q (m x p); R^H.  With `iterations`: one round of orthogonal (subspace) iteration first - the
    // range-finder error of a direction is ~ sigma_(p+1) / sigma_i; Y <- F orth(F^H Q) cubes that ratio.
    bool r_from_kernel = false;
    for (int it = 0; it <= iterations; ++it) {
      Gemm g;
      for (i64 i = 0; i < ncs; ++i) g.add(c.off[i], Yp[i], Btp[i], c.m[i], p[i], c.n[i], L, c.ld1[i], std::max<i64>(c.m[i], 1));
      TMF_TRY(gemm(1, 1.0, 0.0, g));
      if (house && iterations == 0) {
        // only R^H = (F^H Q)^H Q2 is needed below: factor B^H in place and take R^H from the kernel
        LATER_HIP(hipMemsetAsync((void*)d_R, 0, (size_t)std::max<i64>(tR, 1) * el, c.s_main));
        std::vector<HSlab> s;
        for (i64 i = 0; i < ncs; ++i)
          if (c.doE[i]) s.push_back(HSlab{(size_t)i, Btp[i], c.m[i], std::max<i64>(c.m[i], 1), p[i], Rp[i], std::max<i64>(p[i], 1)});
        TMF_TRY(house_slab(s, true, nullptr));
        r_from_kernel = true;
        break;
      }
      LATER_HIP(hipMemcpyAsync((void*)d_Q2, (void*)d_Bt, (size_t)std::max<i64>(tB, 1) * el, hipMemcpyDeviceToDevice, c.s_main));
      for (i64 i = 0; i < ncs; ++i) Q2p[i] = d_Q2 + (u64)(oB[i] * el);
      TMF_TRY(rqr(Q2p, c.m));
      if (it < iterations) {
        Gemm g2;
        for (i64 i = 0; i < ncs; ++i) g2.add(c.off[i], Q2p[i], Yp[i], c.n[i], p[i], c.m[i], L, std::max<i64>(c.m[i], 1), c.ld1[i]);
        TMF_TRY(gemm(0, 1.0, 0.0, g2));
        TMF_TRY(rqr(Yp, c.n));
      }
    }
    if (!r_from_kernel) {
      Gemm g;
      for (i64 i = 0; i < ncs; ++i)
        g.add(Btp[i], Q2p[i], Rp[i], p[i], p[i], c.m[i], std::max<i64>(c.m[i], 1), std::max<i64>(c.m[i], 1), std::max<i64>(p[i], 1));
      TMF_TRY(gemm(1, 1.0, 0.0, g));
    }
    // E4: Jacobi SVD of R^H: left singular vectors Z, sigma; columns below the threshold zeroed
    TMF_TRY(jacobi(Rp, Zp, sigp, &cntp, c.thr2, p, true));
    {  // E5: U0 = Q Z
      Gemm g;
      for (i64 i = 0; i < ncs; ++i) g.add(Yp[i], Zp[i], U0p[i], c.n[i], p[i], p[i], c.ld1[i], std::max<i64>(p[i], 1), c.ld1[i]);
      TMF_TRY(gemm(0, 1.0, 0.0, g));
    }
    {  // E6: T = U0^H (A U0), Jacobi eigen-decomposition
      Gemm g1, g2;
      for (i64 i = 0; i < ncs; ++i) {
        g1.add(c.blk[i], U0p[i], W1p[i], c.n[i], p[i], c.n[i], L, c.ld1[i], c.ld1[i]);
        g2.add(U0p[i], W1p[i], Tp[i], p[i], p[i], c.n[i], c.ld1[i], c.ld1[i], std::max<i64>(p[i], 1));
      }
      TMF_TRY(gemm(0, 1.0, 0.0, g1));
      TMF_TRY(gemm(1, 1.0, 0.0, g2));
      TMF_TRY(jacobi(Tp, Xp, ep, nullptr, 0.0, p, false));
    }
    {  // E7: U_E = U0 X (reuses the Y buffer; Q is no longer needed)
      Gemm g;
      for (i64 i = 0; i < ncs; ++i) g.add(U0p[i], Xp[i], Yp[i], c.n[i], p[i], p[i], c.ld1[i], std::max<i64>(p[i], 1), c.ld1[i]);
      TMF_TRY(gemm(0, 1.0, 0.0, g));
    }
    c.UEp = Yp;
    // ---- host round trip 1: singular values, counts, Ritz values, sweep counts ----
    size_t nsw = 0;
    for (auto q : c.sweep_counts_n) nsw += (size_t)q;
    const size_t b_sig = (size_t)std::max<i64>(tS, 1) * 8, b_cnt = (size_t)ncs * 4;
    TMF_TRY(c.fetch.ensure(2 * b_sig + b_cnt + nsw * 4 + 64));
    char* h = c.fetch.p;
    // (written by a kernel into the page-locked buffer: a DMA copy here waits for its turn between the 64 MB pieces of the
    // previous conversion's tensor download, several ms per conversion)
    LATER(tmf_export_words(h, (const void*)d_sig, (int64_t)b_sig, c.s_main));
    LATER(tmf_export_words(h + b_sig, (const void*)d_e, (int64_t)b_sig, c.s_main));
    LATER(tmf_export_words(h + 2 * b_sig, (const void*)d_cnt, (int64_t)b_cnt, c.s_main));
    size_t o = 2 * b_sig + ((b_cnt + 7) & ~(size_t)7);
    for (size_t j = 0; j < c.sweep_counters.size(); ++j) {
      const int32_t* src_ = c.sweep_counters[j];
      const int64_t nb_ = (int64_t)c.sweep_counts_n[j] * 4;
      char* dst_ = h + o;
      LATER(tmf_export_words(dst_, src_, nb_, c.s_main));
      o += (size_t)c.sweep_counts_n[j] * 4;
    }
    TMF_TRY(run_deferred());
    tick(ST_E, t0);
    const double tw = now_ms();
    HIP_TRY(hipStreamSynchronize(c.s_main));
    tick(ST_WAIT_E, tw);
    c.h_sig.assign((const double*)h, (const double*)h + std::max<i64>(tS, 1));
    c.h_e.assign((const double*)(h + b_sig), (const double*)(h + b_sig) + std::max<i64>(tS, 1));
    c.h_cnt.assign((const int32_t*)(h + 2 * b_sig), (const int32_t*)(h + 2 * b_sig) + ncs);
    int32_t maxsw = 0;
    {
      const int32_t* sw = (const int32_t*)(h + 2 * b_sig + ((b_cnt + 7) & ~(size_t)7));
      for (size_t j = 0; j < nsw; ++j) maxsw = std::max(maxsw, sw[j]);
    }
    c.sweep_counters.clear();
    c.sweep_counts_n.clear();
    double worst = 0.0;
    int32_t sat = 0, weak = 0;
    i64 bad = -1;
    for (i64 i = 0; i < ncs; ++i) {
      if (!full[i]) continue;
      const double sP = c.h_sig[c.oS[i] + P - 1];
      worst = std::max(worst, sP);
      if (c.h_cnt[i] >= p[i]) {
        sat = 1;
        if (bad < 0) bad = c.cs_b[i];
      }
      if (iterations > 0 && sP > 4.6e-4 * sqrt(c.thr2)) {  // (s_P / sqrt(thr2))^3 > 1e-10 even after the iteration
        weak = 1;
        bad = c.cs_b[i];
      }
    }
    c.range_floor = worst;
    c.range_iterations = iterations;
    *worst_out = worst, *sat_out = sat, *weak_out = weak, *sweeps_out = maxsw;
    if (bad_cut) *bad_cut = bad;
    return TMF_OK;
  }

  // ------------------------------------------------------------------ classification (host round trip 1)
  void classify() {
    const double t0 = now_ms();
    const i64 ncs = c.ncs, L = c.L;
    std::vector<double> csum(L + 1, 0.0);
    for (i64 i = 0; i < L; ++i) csum[i + 1] = csum[i] + c.diag[i];
    c.n_fermion = (i64)nearbyint(csum[L]);  // slater.py:414
    c.k.assign(ncs, 0), c.nf.assign(ncs, 0), c.ent0.assign(ncs, 0);
    c.e_side.assign(ncs, {});
    std::vector<double> esum(ncs, 0.0);
    for (i64 i = 0; i < ncs; ++i) {
      if (!c.doE[i]) continue;
      const i64 cnt = c.h_cnt[i];
      const double* e = c.h_e.data() + c.oS[i];
      i64 x_hi = 0, x_lo = 0;
      for (i64 j = 0; j < cnt; ++j) {
        if (e[j] >= 1.0 - c.cutoff) ++x_hi;  // kept but 'filled' by slater.py:350
        if (e[j] < c.cutoff) ++x_lo;
      }
      c.ent0[i] = x_hi;
      c.k[i] = cnt - x_hi - x_lo;
      c.e_side[i].assign(e + x_hi, e + x_hi + c.k[i]);
      double s = 0;
      for (double v : c.e_side[i]) s += v;
      esum[i] = s;
    }
    if (c.has_centre) {  // right-side eigenvalues are 1 - e_L reversed (slater.py:386 convention)
      const i64 cl = c.centre_L, cr = c.centre_R;
      c.k[cr] = c.k[cl];
      c.e_side[cr].resize(c.k[cl]);
      double s = 0;
      for (i64 j = 0; j < c.k[cl]; ++j) c.e_side[cr][j] = 1.0 - c.e_side[cl][c.k[cl] - 1 - j], s += c.e_side[cr][j];
      esum[cr] = s;
    }
    for (i64 i = 0; i < ncs; ++i) {
      const double tr = c.cs_side[i] == 0 ? csum[c.cs_b[i]] : csum[L] - csum[c.cs_b[i]];
      const i64 v = (i64)nearbyint(tr - esum[i]);
      c.nf[i] = std::min(std::max<i64>(v, 0), c.n[i] - c.k[i]);
    }
    tick(ST_CLASSIFY, t0);
  }

  // ------------------------------------------------------------------ host phase: enumeration + site preparation
  int host_phase() {
    double t0 = now_ms();
    const i64 L = c.L, ncs = c.ncs;
    std::vector<i64> side_idx((size_t)(L + 1) * 2, -1);
    for (i64 i = 0; i < ncs; ++i) side_idx[(size_t)c.cs_b[i] * 2 + c.cs_side[i]] = i;
    c.my_cuts.clear();
    for (i64 i = 0; i < ncs; ++i)
      if (c.my_cuts.empty() || c.my_cuts.back() != c.cs_b[i]) c.my_cuts.push_back(c.cs_b[i]);
    const i64 ncut = c.ncut = (i64)c.my_cuts.size();
    c.cpos.assign(L + 1, -1);
    for (i64 j = 0; j < ncut; ++j) c.cpos[c.my_cuts[j]] = j;
    c.kk_cut.assign(ncut, 0), c.nfl.assign(ncut, 0), c.nfr.assign(ncut, 0), c.e_off.assign(ncut, 0);
    std::vector<i64> src(ncut);
    std::vector<char> hasL(ncut);
    i64 tot = 0;
    for (i64 j = 0; j < ncut; ++j) {
      const i64 iL = side_idx[(size_t)c.my_cuts[j] * 2], iR = side_idx[(size_t)c.my_cuts[j] * 2 + 1];
      hasL[j] = iL >= 0;
      src[j] = hasL[j] ? iL : iR;  // the side whose eigenvalues define e_left
      c.kk_cut[j] = (int32_t)c.k[src[j]];
      const i64 nf_src = c.nf[src[j]];
      const i64 other = c.n_fermion - c.kk_cut[j] - nf_src;  // slater.py:167 / :172
      c.nfl[j] = (int32_t)(hasL[j] ? nf_src : (iR >= 0 ? other : 0));
      c.nfr[j] = (int32_t)(hasL[j] ? (iR >= 0 ? c.nf[iR] : other) : nf_src);
      c.e_off[j] = tot;
      tot += c.kk_cut[j];
    }
    // left eigenvalues of every cut, flat: from the left block as they are, from the right block as 1 - e reversed
    c.e_pool.assign(tot + 1, 0.0);
    for (i64 j = 0; j < ncut; ++j) {
      const std::vector<double>& es = c.e_side[src[j]];
      const i64 kk = c.kk_cut[j];
      for (i64 t = 0; t < kk; ++t) c.e_pool[c.e_off[j] + t] = hasL[j] ? es[t] : 1.0 - es[kk - 1 - t];
    }
    i64 cap = c.par.chi_max > 0 ? c.par.chi_max + 1 : 4096;
    const int threads = std::max(1, c.par.host_threads);
    for (;;) {
      // (no re-zeroing: every kept entry is written by the enumeration; the pages are faulted in by the first sweep)
      c.c_sets.resize((size_t)ncut * cap * 2), c.c_lam.resize((size_t)ncut * cap), c.c_q.resize((size_t)ncut * cap);
      c.c_chi.assign(ncut, 0), c.c_chk.assign(ncut, 0);
      const int st = tmf_cut_vectors_batch((int)ncut, c.e_pool.data(), c.e_off.data(), c.kk_cut.data(), c.nfl.data(),
                                           c.par.chi_max > 0 ? c.par.chi_max : 0, c.par.svd_min, c.par.degeneracy_tol,
                                           c.have_sectors ? c.sectors.data() : nullptr,
                                           c.have_sectors ? (int)c.sectors.size() - 1 : 0, cap, c.c_sets.data(), c.c_lam.data(), c.c_q.data(), c.c_chi.data(),
                                           c.c_chk.data(), threads);
      const i64 mx = *std::max_element(c.c_chi.begin(), c.c_chi.end());
      if (st == TMF_E_LIMIT && c.par.chi_max <= 0 && mx > cap) {
        cap = mx + 1;  // unlimited chi: grow the per-cut capacity and redo
        continue;
      }
      TMF_TRY(st);
      break;
    }
    c.cap = cap;
    for (i64 j = 0; j < ncut; ++j)
      if (c.c_chi[j] == 0) {
        set_error("No Schmidt vectors left after filtering by `trunc_par.sectors`!");  // slater.py:668
        return TMF_E_ARG;
      }
    tick(ST_ENUM, t0);

    // ---- per-site integer preparation ----
    t0 = now_ms();
    const i64 ns = c.ns = c.s_hi - c.s_lo;
    c.jobs.assign(ns, tmf_site_job{});
    c.souts.assign(ns, tmf_site_out{});
    c.mode.assign(ns, 0), c.chi_b.assign(ns, 0), c.chi_k.assign(ns, 0);
    i64 rs = 0, cs = 0, br = 0, sc = 0, ix = 0;
    for (i64 j = 0; j < ns; ++j) {
      const i64 site = c.s_lo + j;
      const int md = site >= c.oc ? 1 : 0;
      const i64 bb = md == 0 ? site : site + 1, kb = md == 0 ? site + 1 : site;
      const i64 ib = side_idx[(size_t)bb * 2 + md], ik = side_idx[(size_t)kb * 2 + md];
      const i64 cb = c.cpos[bb], ck = c.cpos[kb];
      c.mode[j] = md;
      c.chi_b[j] = c.c_chi[cb], c.chi_k[j] = c.c_chi[ck];
      tmf_site_job& q = c.jobs[j];
      q.mode = md, q.cut_b = (int32_t)cb, q.cut_k = (int32_t)ck;
      q.k_b = (int32_t)c.k[ib], q.nf_b = (int32_t)c.nf[ib], q.k_k = (int32_t)c.k[ik], q.nf_k = (int32_t)c.nf[ik];
      const i64 mb_cap = c.k[ib] + c.nf[ib] + 1, mk_cap = std::max<i64>(c.k[ik] + c.nf[ik], 1), sec_cap = c.k[ik] + 2;
      const i64 n_bound = std::min<i64>(255, c.k[ik] + std::max<i64>(0, c.nf[ik] + c.k[ik] - c.nf[ib]) + 1);
      const i64 idx_cap = (2 * c.chi_b[j] + c.chi_k[j]) * n_bound + 16;
      q.sec_cap = (int32_t)sec_cap;
      q.row_off = rs, q.col_off = cs, q.bra_off = br, q.sec_off = sc, q.idx_off = ix, q.idx_cap = idx_cap;
      rs += mb_cap, cs += mk_cap, br += 2 * c.chi_b[j], sc += sec_cap, ix += idx_cap;
    }
    c.sc_tot = sc, c.br_tot = br, c.ix_tot = ix;
    c.row_sel.assign(rs + 1, 0), c.row_sign.assign(rs + 1, 0), c.col_sel.assign(cs + 1, 0), c.col_sign.assign(cs + 1, 0);
    c.bra_p.resize(br + 1), c.bra_alpha.resize(br + 1);   // every merged row is written by the site preparation
    c.sec_buf.assign(sc + 1, tmf_sector{});
    TMF_TRY(c.pool_pin.ensure((size_t)ix + 1));  // uploaded straight from page-locked memory
    TMF_TRY(tmf_site_prepare_batch((int)ns, c.jobs.data(), c.c_sets.data(), c.c_q.data(), c.c_chi.data(), cap, c.row_sel.data(),
                                   c.row_sign.data(), c.col_sel.data(), c.col_sign.data(), c.bra_p.data(), c.bra_alpha.data(),
                                   c.sec_buf.data(), (uint8_t*)c.pool_pin.p, c.souts.data(), threads));
    tick(ST_SITEPREP, t0);
    return TMF_OK;
  }

  static constexpr double GAUGE_GROUP_TOL = 1e-9;   // eigenvalues closer than this form one group of the canonical gauge
  // new column order of the left orbitals of the centre cut (empty: unchanged), see filled_stage
  static std::vector<i64> group_order(const std::vector<double>& e, double tol) {
    const i64 k = (i64)e.size();
    std::vector<i64> perm(k);
    std::iota(perm.begin(), perm.end(), 0);
    bool moved = false;
    for (i64 a = 0; a < k;) {
      i64 b = a;
      while (b + 1 < k && !(fabs(e[b + 1] - e[b]) > tol)) ++b;      // utils.py:71
      // insertion sort by descending e (1 - e); values equal to rounding (exact multiplets) keep their order
      for (i64 x = a + 1; x <= b; ++x)
        for (i64 y = x; y > a; --y) {
          const double s1 = e[perm[y]] * (1.0 - e[perm[y]]), s0 = e[perm[y - 1]] * (1.0 - e[perm[y - 1]]);
          if (!(s1 > s0 * (1.0 + 1e-13))) break;
          std::swap(perm[y], perm[y - 1]), moved = true;
        }
      a = b + 1;
    }
    if (!moved) perm.clear();
    return perm;
  }

  // ------------------------------------------------------------------ F: orbital matrices V = [U_E (k) | Q_f (nf)]
  std::vector<i64> ncolV;
  std::vector<u64> Vp;
  int filled_stage() {
    const double t0 = now_ms();
    const i64 ncs = c.ncs, L = c.L, el = c.el;
    ncolV.assign(ncs, 0);
    Vp.assign(ncs, 0);
    i64 tV = 0;
    std::vector<i64> oV(ncs);
    for (i64 i = 0; i < ncs; ++i) {
      ncolV[i] = c.k[i] + c.nf[i];
      oV[i] = tV;
      tV += c.n[i] * ncolV[i];
    }
    u64 d_V;
    TMF_TRY(alloc_el(tV, &d_V));
    for (i64 i = 0; i < ncs; ++i) Vp[i] = d_V + (u64)(oV[i] * el);
    {  // canonical gauge of the entangled Ritz vectors (tmf_gauge_desc): phases, and the basis inside groups of eigenvalues
       // that C does not tell apart, do not depend on the sweep that computed them
      std::vector<tmf_gauge_desc> gd;
      for (i64 i = 0; i < ncs; ++i) {
        if (!c.doE[i] || c.k[i] <= 0) continue;
        std::vector<int32_t> start((size_t)c.k[i]);
        const std::vector<double>& e = c.e_side[i];
        for (i64 j = 0; j < c.k[i]; ++j) {
          // (a group = eigenvalues C does not tell apart: rounding noise, or 1e-9 of their distance from 0 / 1)
          const double w = j > 0 ? std::min(std::min(e[(size_t)j], 1.0 - e[(size_t)j]), std::min(e[(size_t)j - 1], 1.0 - e[(size_t)j - 1])) : 0.0;
          start[(size_t)j] = (j > 0 && !(fabs(e[(size_t)j] - e[(size_t)j - 1]) > 1e-13 + GAUGE_GROUP_TOL * w)) ? start[(size_t)j - 1] : (int32_t)j;
        }
        tmf_gauge_desc q{};
        TMF_TRY(up_vec(start, &q.start));
        q.V = c.UEp[i] + (u64)(c.ent0[i] * c.ld1[i] * el);
        q.n = (int32_t)c.n[i], q.k = (int32_t)c.k[i], q.ld = (int32_t)c.ld1[i], q.from_top = c.cs_side[i] == 1 ? 1 : 0;
        gd.push_back(q);
      }
      if (!gd.empty()) {
        u64 dd;
        TMF_TRY(up_vec(gd, &dd));
        const int nd_ = (int)gd.size();
        int gmax_n = 1, gmax_k = 1;
        for (const tmf_gauge_desc& q : gd) gmax_n = std::max(gmax_n, (int)q.n), gmax_k = std::max(gmax_k, (int)q.k);
        LATER(tmf_canonical_gauge_batched(c.dtype, (const tmf_gauge_desc*)dd, nd_, gmax_n, gmax_k, c.s_main));
      }
    }
    {  // entangled columns (renormalised copy)
      std::vector<tmf_colnorm_desc> d;
      for (i64 i = 0; i < ncs; ++i) {
        if (!c.doE[i]) continue;
        tmf_colnorm_desc q{};
        q.src = c.UEp[i] + (u64)(c.ent0[i] * c.ld1[i] * el), q.dst = Vp[i];
        q.n = (int32_t)c.n[i], q.c = (int32_t)c.k[i], q.lds_ = (int32_t)c.ld1[i], q.ldd = (int32_t)c.ld1[i];
        // Left orbitals of the centre cut inside a group of eigenvalues closer than degeneracy_tol: the reference takes the
        // SVD of v_L^H C_LR v_R per group (utils.py:66-94, called at slater.py:407) and keeps e as it is.  For paired
        // orbitals that block is diagonal with entries sqrt(e (1 - e)), so its SVD is the permutation that sorts them in
        // descending order: orbitals that are only NEARLY degenerate change places, the right partners follow (they are
        // computed from these columns below).
        std::vector<i64> perm;
        if (c.has_centre && i == c.centre_L) perm = group_order(c.e_side[i], c.par.degeneracy_tol);
        if (perm.empty()) {
          d.push_back(q);
          continue;
        }
        q.c = 1;
        for (i64 j = 0; j < c.k[i]; ++j) {
          tmf_colnorm_desc qj = q;
          qj.src = q.src + (u64)(perm[j] * c.ld1[i] * el), qj.dst = q.dst + (u64)(j * c.ld1[i] * el);
          d.push_back(qj);
        }
      }
      TMF_TRY(colcopy(d));
    }
    // centre-right: C_RL U_E(left), reversed, odd columns flipped (slater.py:407-410).  A chain of seven small launches on
    // ONE cut (0.34 ms): it runs on the upload stream, next to the products of the filled bases of all cuts, and the launch
    // stream waits for it before the Gram-Schmidt that reads these columns.
    hipEvent_t centre_done = nullptr;
    if (c.has_centre && c.k[c.centre_L] > 0 && c.n[c.centre_R] > 0) {
      const i64 cl = c.centre_L, cr = c.centre_R;
      const i64 kc = c.k[cl], nR = c.n[cr], ldR = c.ld1[cr];
      u64 d_pair, d_T, d_scrc;
      TMF_TRY(alloc_el(nR * kc, &d_pair));
      const bool side = !(c.par.flags & TMF_SWEEP_ONE_STREAM);
      hipStream_t cs_ = side ? c.s_up : c.s_main;
      if (side) {
        hipEvent_t ev;
        TMF_TRY(new_event(&ev));
        LATER_HIP(hipEventRecord(ev, c.s_main));
        LATER_HIP(hipStreamWaitEvent(c.s_up, ev, 0));
      }
      Gemm g;
      g.add(c.off[cr], Vp[cl], d_pair, nR, kc, c.m[cr], L, c.ld1[cl], ldR);
      TMF_TRY(gemm(0, 1.0, 0.0, g, cs_));
      // The partners of weak orbitals (sigma -> 1e-6) carry errors ~1e-7 from the division by sigma, so they are
      // Gram-Schmidt orthonormalised in order of DECREASING sigma: strong partners stay as computed, weak ones are
      // corrected against them.  Column permutations are k x k (signed) permutation GEMMs.
      const std::vector<double>& eL = c.e_side[cl];
      std::vector<i64> order(kc);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](i64 a, i64 b) { return eL[a] * (1.0 - eL[a]) > eL[b] * (1.0 - eL[b]); });
      const size_t w = c.cplx ? 2 : 1;
      std::vector<double> Pm((size_t)kc * kc * w, 0.0), Sm((size_t)kc * kc * w, 0.0);  // column-major
      for (i64 a = 0; a < kc; ++a) {
        const i64 j = order[a];
        Pm[(size_t)(a * kc + j) * w] = 1.0;                                             // T[:, a] = pair[:, order[a]]
        const i64 col = kc - 1 - j;
        Sm[(size_t)(col * kc + a) * w] = (col & 1) ? -1.0 : 1.0;                         // slater.py:410 reversal + signs
      }
      u64 t_P, t_S;
      TMF_TRY(up_vec(Pm, &t_P));
      TMF_TRY(up_vec(Sm, &t_S));
      TMF_TRY(alloc_el(nR * kc, &d_T));
      Gemm g1;
      g1.add(d_pair, t_P, d_T, nR, kc, kc, ldR, kc, ldR);
      TMF_TRY(gemm(0, 1.0, 0.0, g1, cs_));
      TMF_TRY(alloc_el((kc + 1) * PANEL_W, &d_scrc));
      TMF_TRY(bcgs({Slab{d_T, nR, ldR, 0, kc, d_scrc}}, 2, false, false, cs_));
      Gemm g2;
      g2.add(d_T, t_S, Vp[cr], nR, kc, kc, ldR, kc, ldR);
      TMF_TRY(gemm(0, 1.0, 0.0, g2, cs_));
      if (side) {
        TMF_TRY(new_event(&centre_done));
        LATER_HIP(hipEventRecord(centre_done, c.s_up));
      }
    }
    auto join_centre = [&]() {
      if (centre_done) {
        hipEvent_t ev = centre_done;
        LATER_HIP(hipStreamWaitEvent(c.s_main, ev, 0));
        centre_done = nullptr;
      }
    };
    // filled: Y = A Omega_f, projected off U_E and orthonormalised
    i64 maxnf = 0, maxcol = 0;
    for (i64 i = 0; i < ncs; ++i) maxnf = std::max(maxnf, c.nf[i]), maxcol = std::max(maxcol, ncolV[i]);
    if (maxnf > 0) {
      u64 d_OmF, d_scr2;
      TMF_TRY(alloc_el(L * maxnf, &d_OmF));
      LATER(tmf_fill_normal(c.dtype, (void*)d_OmF, L * maxnf, 0xF111ED, c.s_main));
      std::vector<u64> Vf(ncs);
      for (i64 i = 0; i < ncs; ++i) Vf[i] = Vp[i] + (u64)(c.k[i] * c.ld1[i] * el);
      // one multiplication by A: the filled space has eigenvalue >= 1 - 1e-12, everything that is not projected off
      // with U_E below has eigenvalue <= 1e-12 (the reference's own cutoff)
      TMF_TRY(nested('A', d_OmF, L, Vf, c.nf));
      // 64-column outer blocks in the Gram-Schmidt (a quarter of the re-reads of the earlier columns): the coefficient
      // scratch of a slab is (columns) x 64
      const bool wide = !(c.par.flags & TMF_SWEEP_NARROW_BCGS);
      const i64 per = (maxcol + 1) * (wide ? 64 : PANEL_W);
      TMF_TRY(alloc_el(per * ncs, &d_scr2));
      // The Gram-Schmidt of the filled bases is a chain of ~110 short dependent launches (per 16-column panel: descriptors,
      // coefficients, update, two Gram / Cholesky rounds; 20 - 35 us each, every one over all slabs) - bound by the latency
      // of one launch, not by the GPU.  The slabs of the left and of the right blocks are independent: their two chains run
      // on two streams at once (the upload stream is idle until the index lists go up).
      std::vector<Slab> s[2];
      for (i64 i = 0; i < ncs; ++i)
        if (c.nf[i] > 0) s[c.cs_side[i]].push_back(Slab{Vp[i], c.n[i], c.ld1[i], c.k[i], ncolV[i], d_scr2 + (u64)(i * per * el)});
      const int passes = (c.par.flags & TMF_SWEEP_TWO_PASSES) ? 2 : 1;
      const bool cholqr = !(c.par.flags & TMF_SWEEP_NO_CHOLQR);
      const bool two_streams = !s[0].empty() && !s[1].empty() && !(c.par.flags & TMF_SWEEP_ONE_STREAM);
      const bool fused = !(c.par.flags & TMF_SWEEP_UNFUSED_BCGS);
      join_centre();
      if (two_streams) {
        TMF_TRY(bcgs(s[1], passes, cholqr, wide, c.s_up, fused));
        TMF_TRY(bcgs(s[0], passes, cholqr, wide, nullptr, fused));
        hipEvent_t ev;
        TMF_TRY(new_event(&ev));
        LATER_HIP(hipEventRecord(ev, c.s_up));
        LATER_HIP(hipStreamWaitEvent(c.s_main, ev, 0));
      } else {
        s[0].insert(s[0].end(), s[1].begin(), s[1].end());
        TMF_TRY(bcgs(s[0], passes, cholqr, wide, nullptr, fused));
      }
    }
    join_centre();
    // self-check of the centre cut (testing.py:131-177; slater.py:419-420 runs it only there)
    c.n_checks = 0;
    c.d_chk = nullptr;
    if ((c.par.flags & TMF_SWEEP_CHECKS) && c.has_centre && c.doE[c.centre_L]) {
      const i64 iL = c.centre_L, iR = c.centre_R;
      const i64 qL = ncolV[iL], qR = ncolV[iR], kc = c.k[iL];
      std::vector<double> w;  // wL | wR | sv
      for (double v : c.e_side[iL]) w.push_back(v);
      for (i64 j = 0; j < c.nf[iL]; ++j) w.push_back(1.0);
      const size_t oR_ = w.size();
      for (double v : c.e_side[iR]) w.push_back(v);
      for (i64 j = 0; j < c.nf[iR]; ++j) w.push_back(1.0);
      const size_t oS_ = w.size();
      for (i64 j = 0; j < kc; ++j) {
        const double e = c.e_side[iL][j];
        w.push_back(sqrt(e * (1.0 - e)) * (((kc - 1 - j) & 1) ? -1.0 : 1.0));  // slater.py:266-268
      }
      w.push_back(0.0);
      u64 t_w;
      TMF_TRY(up_vec(w, &t_w));
      // the five deviations live in the output block (read by the download stream after this sweep's memory set
      // may have been handed to the sweep after next)
      TMF_TRY(acquire_slot(0));
      void* chk = c.slot->d_chk;
      // (0.19 ms on ONE cut: on the upload stream, beside the overlap products of all sites; the download and the reuse of
      // this memory set wait for it)
      const bool chk_side = !(c.par.flags & TMF_SWEEP_ONE_STREAM);
      hipStream_t ks_ = chk_side ? c.s_up : c.s_main;
      if (chk_side) {
        hipEvent_t ev;
        TMF_TRY(new_event(&ev));
        LATER_HIP(hipEventRecord(ev, c.s_main));
        LATER_HIP(hipStreamWaitEvent(c.s_up, ev, 0));
      }
      LATER_HIP(hipMemsetAsync(chk, 0, 64, ks_));
      c.d_chk = (char*)chk;
      tmf_recon_desc d[5];
      memset(d, 0, sizeof(d));
      auto item = [&](int i, u64 T, u64 X, u64 Y, u64 wp, i64 rows, i64 cols, i64 q, i64 inner, i64 ldt, i64 ldx, i64 ldy, int md,
                      int yrev) {
        d[i].T = T, d[i].X = X, d[i].Y = Y, d[i].w = wp, d[i].out = (u64)chk + 8 * i;
        d[i].rows = (int32_t)rows, d[i].cols = (int32_t)cols, d[i].q = (int32_t)q, d[i].inner = (int32_t)inner;
        d[i].ldt = (int32_t)ldt, d[i].ldx = (int32_t)ldx, d[i].ldy = (int32_t)ldy, d[i].mode = md, d[i].y_reverse = yrev;
      };
      item(0, 0, Vp[iL], Vp[iL], 0, qL, qL, 0, c.n[iL], 1, c.ld1[iL], c.ld1[iL], 1, 0);                       // vL is not unitary
      item(1, c.blk[iL], Vp[iL], Vp[iL], t_w, c.n[iL], c.n[iL], qL, 0, L, c.ld1[iL], c.ld1[iL], 0, 0);        // vL does not diagonalise C_LL
      item(2, 0, Vp[iR], Vp[iR], 0, qR, qR, 0, c.n[iR], 1, c.ld1[iR], c.ld1[iR], 1, 0);                       // vR is not unitary
      item(3, c.blk[iR], Vp[iR], Vp[iR], t_w + 8 * oR_, c.n[iR], c.n[iR], qR, 0, L, c.ld1[iR], c.ld1[iR], 0, 0);  // vR ... C_RR
      item(4, c.off[iL], Vp[iL], Vp[iR], t_w + 8 * oS_, c.n[iL], c.n[iR], kc, 0, L, c.ld1[iL], c.ld1[iR], 0, 1);  // SVD of C_LR
      std::vector<int32_t> tiles;
      for (int i = 0; i < 5; ++i) {
        const i64 tr = cdiv(d[i].rows, 64), tc = cdiv(d[i].cols, 64);
        for (i64 t = 0; t < tr * tc; ++t) {
          tiles.push_back(i), tiles.push_back((int32_t)(t / std::max<i64>(tc, 1))), tiles.push_back((int32_t)(t % std::max<i64>(tc, 1)));
        }
      }
      if (!tiles.empty()) {
        u64 t_d, t_t;
        TMF_TRY(up(d, sizeof(d), &t_d));
        TMF_TRY(up_vec(tiles, &t_t));
        const int ntile = (int)(tiles.size() / 3);
        LATER(tmf_recon_error_batched(c.dtype, (const tmf_recon_desc*)t_d, (const int32_t*)t_t, ntile, ks_));
      }
      if (chk_side) {
        TMF_TRY(new_event(&c.chk_done));
        hipEvent_t ev = c.chk_done;
        LATER_HIP(hipEventRecord(ev, c.s_up));
      }
      c.n_checks = 5;
    }
    tick(ST_F, t0);
    return TMF_OK;
  }

  // ------------------------------------------------------------------ S1: O = V_bra^H V_ket (needs the orbital matrices only)
  std::vector<i64> e_ib, e_ik, cb_, ck_;
  std::vector<u64> Op, physp;
  int overlap_stage() {
    const double t0 = now_ms();
    const i64 ns = c.s_hi - c.s_lo, L = c.L, el = c.el;
    std::vector<i64> side_idx((size_t)(L + 1) * 2, -1);
    for (i64 i = 0; i < c.ncs; ++i) side_idx[(size_t)c.cs_b[i] * 2 + c.cs_side[i]] = i;
    e_ib.resize(ns), e_ik.resize(ns), cb_.resize(ns), ck_.resize(ns), Op.resize(ns), physp.resize(ns);
    i64 tO = 0;
    std::vector<i64> oO(ns);
    for (i64 j = 0; j < ns; ++j) {
      const i64 site = c.s_lo + j;
      const int md = site >= c.oc ? 1 : 0;
      e_ib[j] = side_idx[(size_t)(md == 0 ? site : site + 1) * 2 + md];
      e_ik[j] = side_idx[(size_t)(md == 0 ? site + 1 : site) * 2 + md];
      cb_[j] = ncolV[e_ib[j]], ck_[j] = ncolV[e_ik[j]];
      oO[j] = tO;
      tO += cb_[j] * ck_[j];
    }
    u64 d_O;
    TMF_TRY(alloc_el(tO, &d_O));
    Gemm g;
    for (i64 j = 0; j < ns; ++j) {
      const int md = (c.s_lo + j) >= c.oc ? 1 : 0;
      const i64 nb_rows = c.n[e_ib[j]];  // contraction length = bra orbitals
      Op[j] = d_O + (u64)(oO[j] * el);
      const u64 Vk_sub = Vp[e_ik[j]] + (u64)((md == 1 ? 1 : 0) * el);  // right mode: physical orbital is row 0 of the ket block
      physp[j] = Vp[e_ik[j]] + (u64)((md == 1 ? 0 : nb_rows) * el);
      g.add(Vp[e_ib[j]], Vk_sub, Op[j], cb_[j], ck_[j], nb_rows, c.ld1[e_ib[j]], c.ld1[e_ik[j]], std::max<i64>(cb_[j], 1));
    }
    TMF_TRY(gemm(1, 1.0, 0.0, g));
    tick(ST_S1, t0);
    return TMF_OK;
  }

  // The output block of this sweep (tensors, det_always, self-check deviations).  Called once per sweep with 0 bytes
  // when only the check buffer is needed yet, and again with the real size once the host phase knows it.
  int acquire_slot(size_t bytes) {
    OutSlot* s = c.slot;
    if (s == nullptr) {
      for (auto& x : c.slots) {
        if (x.busy && hipEventQuery(x.done) == hipSuccess) x.busy = false;
        if (!x.busy && !x.reserved && (s == nullptr || x.cap > s->cap)) s = &x;
      }
      if (s == nullptr) {
        c.slots.emplace_back();
        s = &c.slots.back();
        HIP_TRY(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
        HIP_TRY(hipMalloc((void**)&s->d_chk, 64));
      }
      if (s->ticket >= 0 && !s->waited && s->checks.p) {   // its download is done (not busy): keep what tmf_sweep_wait reports
        FinishedTicket f{s->ticket, s->n_checks, {0}};
        memcpy(f.v, s->checks.p, sizeof(double) * (size_t)std::min(s->n_checks, 8));
        if (c.finished.size() >= 64) c.finished.erase(c.finished.begin());
        c.finished.push_back(f);
      }
      s->reserved = true;
      s->ticket = -1;
      s->waited = true;
      TMF_TRY(s->checks.ensure(64));
      c.slot = s;
    }
    if (s->cap < bytes) {
      if (s->d) HIP_TRY(hipFree(s->d));
      s->d = nullptr, s->cap = 0;
      const size_t ncap = bytes + bytes / 8 + 4096;
      HIP_TRY(hipMalloc((void**)&s->d, ncap));
      s->cap = ncap;
    }
    return TMF_OK;
  }

  // ------------------------------------------------------------------ S2-S4 after the host phase
  int site_stage() {
    double t0 = now_ms();
    const i64 ns = c.ns, el = c.el;
    // the index pool was written into pinned memory by the site preparation; it goes up on a side stream while the
    // gather and the LU run
    void* t_pool;
    TMF_TRY(dalloc(c.ix_tot + 1, 1, &t_pool));
    hipEvent_t ev_main, ev_pool;
    TMF_TRY(new_event(&ev_main));
    TMF_TRY(new_event(&ev_pool));
    {
      const void* pool_src = c.pool_pin.p;
      const size_t pool_bytes = (size_t)c.ix_tot + 1;
      LATER_HIP(hipEventRecord(ev_main, c.s_main));
      LATER_HIP(hipStreamWaitEvent(c.s_up, ev_main, 0));
      LATER_HIP(hipMemcpyAsync(t_pool, pool_src, pool_bytes, hipMemcpyHostToDevice, c.s_up));
      LATER_HIP(hipEventRecord(ev_pool, c.s_up));
    }

    std::vector<i64> mb(ns), mk(ns), ka(ns), sbv(ns), skv(ns);
    i64 tW = 0, maxmb = 0;
    std::vector<i64> oW(ns);
    c.out_off.assign(ns, 0), c.nsec.assign(ns, 0);
    i64 out_tot = 0;
    for (i64 j = 0; j < ns; ++j) {
      const tmf_site_out& o = c.souts[j];
      mb[j] = o.mb, mk[j] = o.mk, ka[j] = o.k_always, sbv[j] = o.sb, skv[j] = o.sk;
      oW[j] = tW;
      tW += mb[j] * mk[j];
      maxmb = std::max(maxmb, mb[j]);
      c.out_off[j] = out_tot;
      out_tot += o.out_elems;
      c.nsec[j] = o.n_sectors;
    }
    c.out_tot = out_tot;
    u64 d_W;
    TMF_TRY(alloc_el(tW, &d_W));
    // output block: tensors followed by det_always per site
    TMF_TRY(acquire_slot((size_t)(out_tot + ns + 2) * el));
    c.d_out = c.slot->d;
    c.d_det = c.slot->d + (size_t)out_tot * el;
    u64 t_rs, t_cs, t_rg, t_cg;
    TMF_TRY(up_vec(c.row_sel, &t_rs)); TMF_TRY(up_vec(c.col_sel, &t_cs)); TMF_TRY(up_vec(c.row_sign, &t_rg)); TMF_TRY(up_vec(c.col_sign, &t_cg));
    std::vector<tmf_gather_desc> gd(ns);
    std::vector<tmf_schur_desc> sd(ns);
    std::vector<u64> Wp(ns), detp(ns), Sp(ns);
    for (i64 j = 0; j < ns; ++j) {
      Wp[j] = d_W + (u64)(oW[j] * el);
      detp[j] = (u64)c.d_det + (u64)(j * el);
      tmf_gather_desc& g = gd[j];
      g.src = Op[j], g.dst = Wp[j];
      g.row_sel = t_rs + (u64)(c.jobs[j].row_off * 4), g.col_sel = t_cs + (u64)(c.jobs[j].col_off * 4);
      g.row_sign = t_rg + (u64)c.jobs[j].row_off, g.col_sign = t_cg + (u64)c.jobs[j].col_off;
      g.phys = physp[j];
      g.rows = (int32_t)mb[j], g.cols = (int32_t)mk[j];
      g.lds_ = (int32_t)std::max<i64>(cb_[j], 1), g.ldd = (int32_t)std::max<i64>(mb[j], 1), g.ldp = (int32_t)c.ld1[e_ik[j]], g.pad = 0;
      tmf_schur_desc& s = sd[j];
      s.W = Wp[j], s.S = 0, s.det = detp[j];
      s.mb = (int32_t)mb[j], s.mk = (int32_t)mk[j], s.k = (int32_t)ka[j], s.ldw = (int32_t)std::max<i64>(mb[j], 1), s.lds = 1, s.pad = 0;
      Sp[j] = Wp[j] + (u64)((ka[j] + ka[j] * std::max<i64>(mb[j], 1)) * el);
    }
    u64 t_gd, t_sd = 0;
    TMF_TRY(up_vec(gd, &t_gd));
    LATER(tmf_gather_signed_batched(c.dtype, (const tmf_gather_desc*)t_gd, (int)ns, c.s_main));
    // Sites ordered by the size of their always-block so that the sites still active at outer step j0 are a prefix of
    // the descriptor arrays of the blocked methods.
    constexpr i64 WB = 64;
    std::vector<i64> order(ns);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](i64 a, i64 b) { return ka[a] > ka[b]; });
    const i64 kmax = ns > 0 ? ka[order[0]] : 0;
    // Fully pivoted blocked LU over several launches: 64 always-columns per outer step, the rank-64 trailing update of
    // all sites as one batched MFMA GEMM (see lu_schur.hip).
    auto lu_pivoted = [&]() -> int {
      i64 piv_tot = 0, t_tot = 0;
      for (i64 j = 0; j < ns; ++j) piv_tot += ka[j], t_tot += ka[j] > 0 ? WB * mk[j] : 0;
      void* d_piv;
      u64 d_T;
      TMF_TRY(dalloc(piv_tot, 4, &d_piv));
      TMF_TRY(alloc_el(t_tot, &d_T));
      std::vector<tmf_lublock_desc> ld(ns);
      i64 po = 0, to = 0;
      for (i64 r = 0; r < ns; ++r) {
        const i64 j = order[r];
        tmf_lublock_desc& q = ld[r];
        q.W = Wp[j], q.det = detp[j], q.piv = (u64)d_piv + (u64)(po * 4), q.T = d_T + (u64)(to * el);
        q.mb = (int32_t)mb[j], q.mk = (int32_t)mk[j], q.k = (int32_t)ka[j], q.ldw = (int32_t)std::max<i64>(mb[j], 1);
        po += ka[j], to += ka[j] > 0 ? WB * mk[j] : 0;
      }
      u64 t_ld;
      TMF_TRY(up_vec(ld, &t_ld));
      for (i64 j0 = 0; j0 < std::max<i64>(kmax, 1); j0 += WB) {
        i64 nact = 0, act_mb = 0, max_cols = 0;
        while (nact < ns && ka[order[nact]] > j0) {
          const i64 j = order[nact], cend = std::min(ka[j], j0 + WB);
          act_mb = std::max(act_mb, mb[j]), max_cols = std::max(max_cols, mk[j] - cend);
          ++nact;
        }
        // (j0 = 0 runs over all sites: a site without always-block gets det = 1)
        LATER(tmf_lu_block_batched(c.dtype, (const tmf_lublock_desc*)t_ld, (int)(j0 == 0 ? ns : nact), (int)j0, (int)WB,
                                   (int)(j0 == 0 ? maxmb : act_mb), c.s_main));
        if (nact == 0) break;
        LATER(tmf_lu_trsm_batched(c.dtype, (const tmf_lublock_desc*)t_ld, (int)nact, (int)j0, (int)WB, (int)max_cols, c.s_main));
        Gemm g;
        for (i64 r = 0; r < nact; ++r) {
          const i64 j = order[r], cend = std::min(ka[j], j0 + WB), ldw = std::max<i64>(mb[j], 1);
          g.add(Wp[j] + (u64)((cend + j0 * ldw) * el), ld[r].T, Wp[j] + (u64)((cend + cend * ldw) * el), mb[j] - cend, mk[j] - cend,
                cend - j0, ldw, WB, ldw);
        }
        TMF_TRY(gemm(0, -1.0, 1.0, g));
      }
      return TMF_OK;
    };
    // Block-local pivoting: row search inside the 64 x 64 diagonal block only, everything else GEMMs with the explicit
    // inverse of the block (lu_schur.hip).  A one-workgroup kernel turns the per-site statistics into a device flag; the
    // fully pivoted method is enqueued right behind it under tmf_launch_condition(flag) and returns at once when the
    // flag is clear.  The host never waits for the verdict (a host synchronisation here serialises the tensor download of
    // this conversion with the next conversion: 53 instead of 29 ms per step, measured); it reads the summary from
    // page-locked memory when the memory set is reused or the statistics are asked for.
    auto lu_local = [&]() -> int {
      i64 n_blk = 0, t_tot = 0;
      for (i64 j = 0; j < ns; ++j)
        if (ka[j] > 0) ++n_blk, t_tot += WB * mk[j];
      u64 d_inv, d_X;
      TMF_TRY(alloc_el(n_blk * WB * WB, &d_inv));
      TMF_TRY(alloc_el(t_tot, &d_X));
      std::vector<tmf_diaginv_desc> ld(ns);
      std::vector<u64> Xp(ns);
      i64 io = 0, to = 0;
      for (i64 r = 0; r < ns; ++r) {
        const i64 j = order[r];
        tmf_diaginv_desc& q = ld[r];
        q.W = Wp[j], q.det = detp[j], q.inv = d_inv + (u64)(io * el);
        Xp[r] = d_X + (u64)(to * el);
        q.mb = (int32_t)mb[j], q.mk = (int32_t)mk[j], q.k = (int32_t)ka[j], q.ldw = (int32_t)std::max<i64>(mb[j], 1);
        if (ka[j] > 0) io += WB * WB, to += WB * mk[j];
      }
      u64 t_ld;
      void *t_minp, *t_flag;
      TMF_TRY(up_vec(ld, &t_ld));
      TMF_TRY(dalloc(2 * ns + 2, 8, &t_minp));
      TMF_TRY(dalloc(2, 4, &t_flag));
      for (i64 step = 0; step * WB < std::max<i64>(kmax, 1); ++step) {
        i64 nact = 0;
        while (nact < ns && ka[order[nact]] > step * WB) ++nact;   // cdiv(k, 64) blocks whichever end they are counted from
        LATER(tmf_diag_inverse_batched(c.dtype, (const tmf_diaginv_desc*)t_ld, (int)(step == 0 ? ns : nact), (int)step, t_minp, c.s_main));
        if (nact == 0) break;
        Gemm g1, g2;
        for (i64 r = 0; r < nact; ++r) {
          const i64 j = order[r], nb0 = (ka[j] - 1) % WB + 1;
          const i64 j0 = step == 0 ? 0 : nb0 + (step - 1) * WB, cend = step == 0 ? nb0 : j0 + WB, nb = cend - j0;
          const i64 ldw = std::max<i64>(mb[j], 1), rows2 = mb[j] - cend, cols2 = mk[j] - cend;
          if (rows2 <= 0 || cols2 <= 0) continue;   // nothing behind the block
          g1.add(ld[r].inv, Wp[j] + (u64)((j0 + cend * ldw) * el), Xp[r], nb, cols2, nb, WB, ldw, WB);          // X = D^-1 A12
          g2.add(Wp[j] + (u64)((cend + j0 * ldw) * el), Xp[r], Wp[j] + (u64)((cend + cend * ldw) * el), rows2, cols2, nb, ldw, WB, ldw);
        }
        TMF_TRY(gemm(0, 1.0, 0.0, g1));
        TMF_TRY(gemm(0, -1.0, 1.0, g2));
      }
      // verdict -> device flag + summary in page-locked host memory (written by the kernel: no DMA copy on this stream)
      TMF_TRY(c.lu_stats.ensure(64));
      double* summary = (double*)c.lu_stats.p + 3 * c.cur;
      {
        const double cap_ = c.lu_inverse_cap;
        const int force_ = (c.par.flags & TMF_SWEEP_LU_FORCE_FALLBACK) ? 1 : 0;
        LATER(tmf_diag_inverse_verdict(t_minp, (int)ns, cap_, force_, (int32_t*)t_flag, summary, c.s_main));
      }
      c.lu_pending[c.cur] = true;
      if (getenv("TMF_LU_DEBUG")) {   // diagnostic (tools/lu_debug.py): per-site statistics, with a synchronisation
        TMF_TRY(run_deferred());
        std::vector<double> h((size_t)(2 * ns + 2));
        HIP_TRY(hipMemcpyAsync(h.data(), t_minp, (size_t)ns * 16, hipMemcpyDeviceToHost, c.s_main));
        HIP_TRY(hipStreamSynchronize(c.s_main));
        for (i64 r = 0; r < ns; ++r)
          if (h[2 * r] < 1e-4 || !(h[2 * r + 1] < 16.0))
            fprintf(stderr, "lu site %lld mode %d: k %lld mb %lld mk %lld nf_b %d nf_k %d min pivot %.3e max |D^-1| %.3e\n",
                    (long long)order[r], (int)c.jobs[order[r]].mode, (long long)ka[order[r]], (long long)mb[order[r]],
                    (long long)mk[order[r]], (int)c.jobs[order[r]].nf_b, (int)c.jobs[order[r]].nf_k, std::sqrt(h[2 * r]),
                    std::sqrt(h[2 * r + 1]));
      }
      // the fallback: gather again (the elimination has overwritten W), fully pivoted blocked LU - all of it conditional
      dq.emplace_back([=]() -> int {
        tmf_launch_condition((const int32_t*)t_flag);
        return TMF_OK;
      });
      c.conditional = true;
      LATER(tmf_gather_signed_batched(c.dtype, (const tmf_gather_desc*)t_gd, (int)ns, c.s_main));
      const int st = lu_pivoted();
      c.conditional = false;
      dq.emplace_back([]() -> int {
        tmf_launch_condition(nullptr);
        return TMF_OK;
      });
      return st;
    };
    if (c.par.flags & TMF_SWEEP_LU_SINGLE) {  // A/B switch: the one-workgroup-per-site kernel
      TMF_TRY(up_vec(sd, &t_sd));
      LATER(tmf_lu_schur_batched(c.dtype, (const tmf_schur_desc*)t_sd, (int)ns, (int)maxmb, c.s_main));
    } else if (c.par.flags & TMF_SWEEP_LU_PIVOTED) {
      TMF_TRY(lu_pivoted());
    } else {
      TMF_TRY(lu_local());
    }
    tick(ST_SCHUR, t0);

    // ---- S4: all minors ----
    t0 = now_ms();
    LATER_HIP(hipStreamWaitEvent(c.s_main, ev_pool, 0));  // the determinant kernels read the pool
    const double flop_per_det = c.cplx ? (8.0 / 3.0) : (2.0 / 3.0);  // LU of an n x n complex / real matrix
    c.n_det = 0;
    std::vector<i64> rest_keys;
    bool have_rest_keys = false;
    const bool use_ppt = !(c.par.flags & (TMF_SWEEP_DET_REDUCED | TMF_SWEEP_DET_DIRECT));
    if (use_ppt) {
      std::vector<tmf_det_site> ds(ns);
      for (i64 j = 0; j < ns; ++j) {
        ds[j].S = Sp[j], ds[j].scale = detp[j], ds[j].lds = (int32_t)std::max<i64>(mb[j], 1), ds[j].pad = 0;
        ds[j].idx_base = (u64)t_pool + (u64)c.jobs[j].idx_off;
        ds[j].out_base = (u64)c.d_out + (u64)(c.out_off[j] * el);
      }
      i64 n_rest = 0, npairs = 0;
      int32_t lds_max = 0;
      double fl3 = 0;
      i64 nt = tmf_det_tiles_build((int)ns, c.jobs.data(), c.souts.data(), c.sec_buf.data(), ds.data(), (int)el, 16384, nullptr, 0,
                                   nullptr, 0, &n_rest, &lds_max, &fl3, &npairs);
      std::vector<tmf_det_desc> tiles((size_t)std::max<i64>(nt, 1));
      rest_keys.assign((size_t)std::max<i64>(n_rest, 1), 0);
      nt = tmf_det_tiles_build((int)ns, c.jobs.data(), c.souts.data(), c.sec_buf.data(), ds.data(), (int)el, 16384, tiles.data(), nt,
                               rest_keys.data(), (i64)rest_keys.size(), &n_rest, &lds_max, &fl3, &npairs);
      rest_keys.resize((size_t)n_rest);
      have_rest_keys = true;
      if (nt > 0) {
        u64 t_dd;
        TMF_TRY(up(tiles.data(), (size_t)nt * sizeof(tmf_det_desc), &t_dd));
        KernelEvent ev{};
        if (timing()) {
          TMF_TRY(new_event(&ev.e0));
          TMF_TRY(new_event(&ev.e1));
          LATER_HIP(hipEventRecord(ev.e0, c.s_main));
        }
        // mask width of the launch: 32 bits when every sector fits them AND no sector has more than 2048 ket sets (the
        // 32-bit kernel would take those on its unclassed path, which sends order-5 pairs to the queue; the 64-bit kernel
        // uses the closed form of the classed phase there: pair by pair the same arithmetic whichever kernel runs)
        i64 widest = 0;
        for (i64 j = 0; j < ns; ++j) widest = std::max(widest, std::max(sbv[j], skv[j]));
        for (size_t q = 0; q + 1 < c.sec_buf.size(); ++q)
          if (c.sec_buf[q].c1 - c.sec_buf[q].c0 > 2048) widest = std::max<i64>(widest, 33);
        static const int force_bits = getenv("TMF_PPT_MASK_BITS") ? atoi(getenv("TMF_PPT_MASK_BITS")) : 0;   // (A/B switch)
        const int mbits = force_bits == 64 ? 64 : (widest <= 32 ? 32 : 64);
        LATER(tmf_det_ppt_batched_w(c.dtype, (const tmf_det_desc*)t_dd, (int)nt, lds_max, mbits, c.s_main));
        if (timing()) {
          LATER_HIP(hipEventRecord(ev.e1, c.s_main));
          ev.flops = fl3 * flop_per_det, ev.n = npairs, ev.kind = 0, ev.order = 0;
          c.det_events.push_back(ev);
        }
        c.n_det += npairs;
      }
    }
    // general path: every sector (A/B switches) or the ones the exchange kernel does not take
    struct Rest {
      i64 site, loc;
    };
    std::vector<Rest> rs;
    if (!have_rest_keys) {
      for (i64 j = 0; j < ns; ++j)
        for (i64 q = 0; q < c.nsec[j]; ++q) rs.push_back(Rest{j, q});
    } else {
      for (i64 key : rest_keys) rs.push_back(Rest{key >> 32, key & 0xFFFFFFFF});
    }
    if (!rs.empty()) {
      struct TileX {
        tmf_det_desc d;
        int cls;
        bool red;
        i64 lneed, pairs, nq;
      };
      std::vector<TileX> tx;
      const bool force_direct = (c.par.flags & TMF_SWEEP_DET_DIRECT) != 0;
      static const bool force_global = getenv("TMF_DET_GLOBAL") && atoi(getenv("TMF_DET_GLOBAL")) != 0;   // (test switch: class 255 for every sector of the general path)
      for (const Rest& r : rs) {
        const tmf_sector& sec = c.sec_buf[c.jobs[r.site].sec_off + r.loc];
        const i64 nq = sec.n, nsb = sec.r1 - sec.r0, nsk = sec.c1 - sec.c0, j = r.site;
        const int cls = nq <= 32 ? (int)nq : 64;  // exact order for n <= 32 (templated kernels), generic above
        const i64 ta = std::min(std::max<i64>(cdiv(4096, nsk), 1), nsb);
        const i64 ntile = cdiv(nsb, ta);
        const u64 pbase = (u64)t_pool + (u64)c.jobs[j].idx_off;
        for (i64 t = 0; t < ntile; ++t) {
          TileX x{};
          x.d.S = Sp[j], x.d.scale = detp[j];
          x.d.bra_idx = pbase + (u64)sec.bra_off, x.d.ket_idx = pbase + (u64)sec.ket_off;
          x.d.out = (u64)c.d_out + (u64)((c.out_off[j] + sec.out_off) * el);
          x.d.sb = (int32_t)sbv[j], x.d.sk = (int32_t)skv[j], x.d.lds = (int32_t)std::max<i64>(mb[j], 1);
          x.d.n = (int32_t)nq, x.d.nsb = (int32_t)nsb, x.d.nsk = (int32_t)nsk;
          x.d.a0 = (int32_t)(t * ta), x.d.a1 = (int32_t)std::min(nsb, t * ta + ta);
          x.cls = cls, x.nq = nq;
          // LDS per workgroup (det_gather.hip): M, index lists, then per wave the gathered rows M[rows(a), :] (n*sk) +
          // 64 scratch elements; class 64 instead holds one n x n minor
          const i64 gpw = nq <= 8 ? 8 : (nq <= 16 ? 4 : 2);
          i64 lneed = a16(sbv[j] * skv[j] * el) + a16(nsk * nq) + a16(ta * nq) +
                      (cls == 64 ? nq * nq * el : 4 * ((nq | 1) * skv[j] + gpw * (nq + 1)) * el);
          x.pairs = (i64)(x.d.a1 - x.d.a0) * nsk;
          // reduced-minor kernel (one Gauss-Jordan per bra row-set) whenever the sometimes-matrix has <= 64 columns
          // and 1 <= n <= 32; the direct kernel covers the rest
          bool red = cls >= 1 && cls <= 32 && skv[j] <= 64 && !force_direct;
          const i64 lred = a16(sbv[j] * skv[j] * el) + a16(nsk * nq) + a16(nsk * 8) + a16(ta * nq) +
                           4 * (((nq | 1) * skv[j] + 264) * el + 576);  // _native.reduced_det_lds - 16
          red = red && (lred + 16 <= 160 * 1024);
          x.red = red;
          x.lneed = red ? lred : lneed;
          if (x.lneed > 160 * 1024 || nq > 64 || force_global) {
            // the sometimes-matrix (or the minor) does not fit next to the index lists: class 255 reads the matrix from
            // global memory and keeps only the minor in LDS (slow, general)
            x.cls = 255, x.red = false, x.lneed = a16(nq * nq * el) + 64;
            if (x.lneed > 158 * 1024) {
              set_error("minors of order %lld exceed the LDS of a CU (%lld B)", (long long)nq, (long long)x.lneed);
              return TMF_E_LIMIT;
            }
          }
          tx.push_back(x);
        }
      }
      // one launch per (class, kernel), the heaviest first; inside a launch the biggest tiles first
      struct Group {
        int cls;
        bool red;
        i64 pairs;
      };
      std::vector<Group> groups;
      for (auto& x : tx) {
        auto it = std::find_if(groups.begin(), groups.end(), [&](const Group& g) { return g.cls == x.cls && g.red == x.red; });
        if (it == groups.end()) groups.push_back(Group{x.cls, x.red, x.pairs});
        else it->pairs += x.pairs;
      }
      std::sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) {
        if (a.cls != b.cls) return a.cls < b.cls;
        return (int)a.red < (int)b.red;
      });
      std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.pairs > b.pairs; });
      for (const Group& gq : groups) {
        std::vector<const TileX*> sel;
        for (auto& x : tx)
          if (x.cls == gq.cls && x.red == gq.red) sel.push_back(&x);
        std::stable_sort(sel.begin(), sel.end(), [](const TileX* a, const TileX* b) {
          return a->pairs * (a->nq + 1) * (a->nq + 1) > b->pairs * (b->nq + 1) * (b->nq + 1);
        });
        std::vector<tmf_det_desc> dd(sel.size());
        i64 lmax = 0, pairs = 0;
        double fl = 0;
        for (size_t i = 0; i < sel.size(); ++i) {
          dd[i] = sel[i]->d;
          lmax = std::max(lmax, sel[i]->lneed);
          pairs += sel[i]->pairs;
          fl += (double)sel[i]->pairs * (double)sel[i]->nq * (double)sel[i]->nq * (double)sel[i]->nq;
        }
        u64 t_dd;
        TMF_TRY(up_vec(dd, &t_dd));
        KernelEvent ev{};
        if (timing()) {
          TMF_TRY(new_event(&ev.e0));
          TMF_TRY(new_event(&ev.e1));
          LATER_HIP(hipEventRecord(ev.e0, c.s_main));
        }
        {
          const bool red_ = gq.red;
          const int cls_ = gq.cls, ndd_ = (int)dd.size(), lds_ = (int)lmax + 16;
          LATER((red_ ? tmf_det_reduced_batched : tmf_det_gather_batched)(c.dtype, cls_, (const tmf_det_desc*)t_dd, ndd_, lds_, c.s_main));
        }
        if (timing()) {
          LATER_HIP(hipEventRecord(ev.e1, c.s_main));
          ev.flops = fl * flop_per_det, ev.n = pairs, ev.kind = gq.red ? 1 : 2, ev.order = gq.cls;
          c.det_events.push_back(ev);
        }
        c.n_det += pairs;
      }
    }
    {
      if (c.chk_done) {
        hipEvent_t ev = c.chk_done;
        LATER_HIP(hipStreamWaitEvent(c.s_main, ev, 0));   // (long finished: the memory set is not reused under the self-check)
      }
      hipEvent_t done_ = c.set_done[c.cur];
      LATER_HIP(hipEventRecord(done_, c.s_main));  // every kernel of this sweep is enqueued
    }
    TMF_TRY(run_deferred());
    c.set_used[c.cur] = true;
    tick(ST_DET, t0);
    return TMF_OK;
  }

  int sites(tmf_sweep_dims* dims) {
    if (!c.begun || c.h_cnt.empty()) {
      set_error("tmf_sweep_sites: call tmf_sweep_begin and tmf_sweep_entangled first");
      return TMF_E_ARG;
    }
    classify();
    TMF_TRY(filled_stage());
    TMF_TRY(overlap_stage());
    TMF_TRY(run_deferred());  // one descriptor transfer, then every launch of the two stages
    TMF_TRY(host_phase());  // integer work on host threads while the GPU runs the filled-basis launches
    TMF_TRY(site_stage());
    c.have_sites = c.have_out = true;
    dims->ncut = c.ncut, dims->cap = c.cap, dims->ns = c.ns, dims->sec_tot = c.sc_tot, dims->bra_tot = c.br_tot;
    dims->e_tot = (i64)c.e_pool.size() - 1, dims->out_elems = c.out_tot, dims->elem_bytes = c.el;
    return TMF_OK;
  }

  // ------------------------------------------------------------------ result
  int download(const tmf_sweep_ptrs* q, int want_tensors, i64* ticket) {
    if (!c.have_sites) {
      set_error("tmf_sweep_download: no finished sweep");
      return TMF_E_ARG;
    }
    double t0 = now_ms();
    auto cp = [](void* dst, const void* src, size_t bytes) {
      if (dst && bytes) memcpy(dst, src, bytes);
    };
    const i64 ns = c.ns, ncut = c.ncut;
    cp(q->my_cuts, c.my_cuts.data(), (size_t)ncut * 8);
    cp(q->c_sets, c.c_sets.data(), c.c_sets.size() * 8);
    cp(q->c_lam, c.c_lam.data(), c.c_lam.size() * 8);
    cp(q->c_q, c.c_q.data(), c.c_q.size() * 4);
    cp(q->c_chi, c.c_chi.data(), (size_t)ncut * 8);
    cp(q->c_chk, c.c_chk.data(), (size_t)ncut * 8);
    cp(q->e_pool, c.e_pool.data(), c.e_pool.size() * 8);
    cp(q->e_off, c.e_off.data(), (size_t)ncut * 8);
    cp(q->kk_cut, c.kk_cut.data(), (size_t)ncut * 4);
    cp(q->nfl, c.nfl.data(), (size_t)ncut * 4);
    cp(q->nfr, c.nfr.data(), (size_t)ncut * 4);
    cp(q->mode, c.mode.data(), (size_t)ns * 4);
    if (q->sec_off)
      for (i64 j = 0; j < ns; ++j) ((i64*)q->sec_off)[j] = c.jobs[j].sec_off;
    if (q->bra_off)
      for (i64 j = 0; j < ns; ++j) ((i64*)q->bra_off)[j] = c.jobs[j].bra_off;
    cp(q->nsec, c.nsec.data(), (size_t)ns * 8);
    cp(q->sectors, c.sec_buf.data(), c.sec_buf.size() * sizeof(tmf_sector));
    cp(q->out_off, c.out_off.data(), (size_t)ns * 8);
    cp(q->chi_b, c.chi_b.data(), (size_t)ns * 8);
    cp(q->chi_k, c.chi_k.data(), (size_t)ns * 8);
    cp(q->bra_p, c.bra_p.data(), c.bra_p.size() * 4);
    cp(q->bra_alpha, c.bra_alpha.data(), c.bra_alpha.size() * 4);
    tick(ST_RESULT, t0);
    t0 = now_ms();
    // tensors: asynchronous DMA on the download stream (the launch stream is free for the next conversion; the output
    // block stays reserved until the copy has finished)
    OutSlot* s = c.slot;
    hipEvent_t ev;
    TMF_TRY(new_event(&ev));
    HIP_TRY(hipEventRecord(ev, c.s_main));
    HIP_TRY(hipStreamWaitEvent(c.s_down, ev, 0));
    s->n_checks = c.n_checks;
    if (c.n_checks > 0) HIP_TRY(hipMemcpyAsync(s->checks.p, c.d_chk, (size_t)c.n_checks * 8, hipMemcpyDeviceToHost, c.s_down));
    if (q->det) HIP_TRY(hipMemcpyAsync(q->det, c.d_det, (size_t)ns * c.el, hipMemcpyDeviceToHost, c.s_down));
    if (want_tensors && q->out) {
      const size_t total = (size_t)c.out_tot * c.el, chunk = (size_t)64 << 20;
      for (size_t o = 0; o < total; o += chunk)  // in pieces: small copies of other streams get a turn
        HIP_TRY(hipMemcpyAsync((char*)q->out + o, c.d_out + o, std::min(chunk, total - o), hipMemcpyDeviceToHost, c.s_down));
    }
    HIP_TRY(hipEventRecord(s->done, c.s_down));
    s->busy = true;
    s->reserved = false;
    c.slot = nullptr;
    c.have_sites = false;  // one download per sweep
    s->ticket = c.next_ticket++;
    s->waited = false;
    *ticket = s->ticket;
    tick(ST_DOWNLOAD, t0);
    c.stage_ms[ST_TOTAL] = now_ms() - c.t_begin;
    return TMF_OK;
  }
};

#include "sweep_pf.inc"

}  // namespace
}  // namespace tmf

// ====================================================================================== C ABI
extern "C" int tmf_ctx_create(int device, tmf_ctx** out) {
  int ndev = 0;
  TMF_TRY(check_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount"));
  if (device < 0 || device >= ndev) {
    set_error("tmf_ctx_create: device %d of %d", device, ndev);
    return TMF_E_ARG;
  }
  HIP_TRY(hipSetDevice(device));
  tmf_ctx* c = new tmf_ctx();
  c->device = device;
  int st = check_hip(hipStreamCreateWithFlags(&c->s_main, hipStreamNonBlocking), "hipStreamCreate");
  if (st == TMF_OK) st = check_hip(hipStreamCreateWithFlags(&c->s_down, hipStreamNonBlocking), "hipStreamCreate");
  if (st == TMF_OK) st = check_hip(hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking), "hipStreamCreate");
  for (int i = 0; i < 2 && st == TMF_OK; ++i)
    st = check_hip(hipEventCreateWithFlags(&c->set_done[i], hipEventDisableTiming), "hipEventCreate");
  if (st != TMF_OK) {
    delete c;
    return st;
  }
  c->slots.reserve(16);  // OutSlot addresses stay valid
  *out = c;
  return TMF_OK;
}

extern "C" void tmf_ctx_destroy(tmf_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& s : c->slots) {
    if (s.d) (void)hipFree(s.d);
    if (s.d_chk) (void)hipFree(s.d_chk);
    if (s.done) (void)hipEventDestroy(s.done);
    s.checks.release();
  }
  for (int i = 0; i < 2; ++i) {
    for (auto e : c->event_pool_set[i]) (void)hipEventDestroy(e);
    if (c->set_done[i]) (void)hipEventDestroy(c->set_done[i]);
    c->dev_set[i].release();
    c->stage_set[i].release();
  }
  c->fetch.release(), c->lu_stats.release(), c->pool_pin.release(), c->c_pin.release();
  if (c->s_main) (void)hipStreamDestroy(c->s_main);
  if (c->s_down) (void)hipStreamDestroy(c->s_down);
  if (c->s_up) (void)hipStreamDestroy(c->s_up);
  delete c;
}

extern "C" int tmf_sweep_begin(tmf_ctx* ctx, const void* C, const tmf_sweep_params* par) {
  if (!ctx || !C || !par) {
    set_error("tmf_sweep_begin: null argument");
    return TMF_E_ARG;
  }
  return Sweep(*ctx).begin(C, par);
}

extern "C" int tmf_sweep_entangled(tmf_ctx* ctx, int P, int iterations, double* smallest_sigma, int32_t* saturated, int32_t* weak,
                                   int32_t* max_sweeps, int64_t* bad_cut) {
  if (!ctx || P < 1 || iterations < 0) {
    set_error("tmf_sweep_entangled: bad argument");
    return TMF_E_ARG;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return Sweep(*ctx).entangled(P, iterations, smallest_sigma, saturated, weak, max_sweeps, bad_cut);
}

extern "C" int tmf_sweep_sites(tmf_ctx* ctx, tmf_sweep_dims* dims) {
  if (!ctx || !dims) {
    set_error("tmf_sweep_sites: null argument");
    return TMF_E_ARG;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return Sweep(*ctx).sites(dims);
}

extern "C" int tmf_sweep_download(tmf_ctx* ctx, const tmf_sweep_ptrs* dst, int want_tensors, int64_t* ticket) {
  if (!ctx || !dst || !ticket) {
    set_error("tmf_sweep_download: null argument");
    return TMF_E_ARG;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  return Sweep(*ctx).download(dst, want_tensors, ticket);
}

static OutSlot* find_ticket(tmf_ctx* ctx, int64_t ticket) {
  for (auto& s : ctx->slots)
    if (s.ticket == ticket) return &s;
  return nullptr;
}

extern "C" int tmf_sweep_query(tmf_ctx* ctx, int64_t ticket) {
  OutSlot* s = ctx ? find_ticket(ctx, ticket) : nullptr;
  if (!s) return 1;  // its block has been reused: that download finished long ago
  const hipError_t e = hipEventQuery(s->done);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  return check_hip(e, "hipEventQuery");
}

extern "C" int tmf_sweep_wait(tmf_ctx* ctx, int64_t ticket, double* checks, int32_t* n_checks) {
  if (n_checks) *n_checks = 0;
  OutSlot* s = ctx ? find_ticket(ctx, ticket) : nullptr;
  if (!s) {   // the output block has been reused: the download finished long ago, its deviations were put aside
    if (ctx && checks && n_checks)
      for (const FinishedTicket& f : ctx->finished)
        if (f.ticket == ticket) {
          *n_checks = f.n;
          memcpy(checks, f.v, sizeof(double) * (size_t)f.n);
        }
    return TMF_OK;
  }
  HIP_TRY(hipEventSynchronize(s->done));
  s->busy = false;
  s->waited = true;
  if (checks && n_checks) {
    *n_checks = s->n_checks;
    memcpy(checks, s->checks.p, (size_t)s->n_checks * 8);
  }
  return TMF_OK;
}

extern "C" const char* tmf_sweep_stage_name(int i) { return (i >= 0 && i < N_STAGES) ? kStageNames[i] : ""; }

extern "C" int tmf_sweep_device_out(tmf_ctx* ctx, uint64_t* d_out, int64_t* elems) {
  if (!ctx || !ctx->have_out) {
    set_error("tmf_sweep_device_out: no finished sweep");
    return TMF_E_ARG;
  }
  *d_out = (uint64_t)ctx->d_out, *elems = ctx->out_tot;
  return TMF_OK;
}

extern "C" int tmf_sweep_info_get(tmf_ctx* ctx, tmf_sweep_info* o) {
  if (!ctx || !o) {
    set_error("tmf_sweep_info_get: null argument");
    return TMF_E_ARG;
  }
  memset(o, 0, sizeof(*o));
  memcpy(o->stage_ms, ctx->stage_ms, sizeof(double) * N_STAGES);
  o->range_width = ctx->P, o->range_iterations = ctx->range_iterations, o->range_floor = ctx->range_floor;
  o->n_fermion = ctx->n_fermion, o->device_bytes = (int64_t)(ctx->dev_set[0].total + ctx->dev_set[1].total);
  o->n_det = ctx->n_det;
  if (ctx->lu_pending[0] || ctx->lu_pending[1]) {
    // Only the verdicts of sweeps that have FINISHED on the device (no wait: with an asynchronous download the caller asks
    // for the stage timers while the site stage is still running, and a synchronisation here kept the next conversion's
    // host phases - and, on another context, its kernels - from overlapping it; a verdict still on its way is folded by
    // the next call or by the sweep after next).
    HIP_TRY(hipSetDevice(ctx->device));
    for (int set : {1 - ctx->cur, ctx->cur})                                   // older sweep first
      if (ctx->lu_pending[set] && ctx->set_used[set] && hipEventQuery(ctx->set_done[set]) == hipSuccess) fold_lu_verdict(*ctx, set);
  }
  o->lu_min_pivot = ctx->lu_min_pivot, o->lu_max_inverse = ctx->lu_max_inverse, o->lu_fallbacks = ctx->lu_fallbacks;
  const int64_t n_det_all = ctx->n_det;
  (void)n_det_all;
  if (!ctx->gemm_events.empty() || !ctx->det_events.empty()) {
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->s_main));
    for (auto& e : ctx->gemm_events) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e.e0, e.e1));
      o->gemm_ms += ms, o->gemm_flops += e.flops;
      o->gemm_split_ms[e.kind] += ms, o->gemm_split_flops[e.kind] += e.flops, o->gemm_split_launches[e.kind] += 1;
    }
    o->n_gemm_launches = (int64_t)ctx->gemm_events.size();
    for (auto& e : ctx->det_events) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e.e0, e.e1));
      o->det_all_ms += ms;
      if (ms > o->det_ms) o->det_ms = ms, o->det_flops = e.flops, o->det_kind = e.kind, o->det_order = e.order, o->n_det = e.n;
    }
  }
  return TMF_OK;
}

// ---- one call: whole conversion, result owned by the library -------------------------------------------
struct tmf_result {
  tmf_ctx* ctx;
  tmf_sweep_dims dims;
  int64_t L, s_lo, s_hi, oc;
  int cplx;
  PinnedBuf buf;
  tmf_sweep_ptrs p;
  std::vector<int64_t> cpos;
  double checks[8];
  int32_t n_checks;
};

extern "C" int tmf_slater_sweep(tmf_ctx* ctx, const void* C, const tmf_sweep_params* par, double range_floor_tol,
                                tmf_result** out) {
  if (!ctx || !C || !par || !out) {
    set_error("tmf_slater_sweep: null argument");
    return TMF_E_ARG;
  }
  if (!(range_floor_tol > 0)) range_floor_tol = 1e-11;
  TMF_TRY(tmf_sweep_begin(ctx, C, par));
  // Adequacy of the range finder is CHECKED, not assumed (see temfpy_amd/engine.py: entangled_stage_adaptive): the
  // state error is ~ the smallest captured singular value; above the tolerance one subspace iteration cubes the
  // ratio; a saturated or still too weak cut moves on to the next width.
  const int ladder[3] = {64, 128, 256};
  bool done = false;
  for (int li = 0; li < 3 && !done; ++li) {
    double worst;
    int32_t sat, weak, sweeps;
    int64_t bad;
    TMF_TRY(tmf_sweep_entangled(ctx, ladder[li], 0, &worst, &sat, &weak, &sweeps, &bad));
    if (sweeps >= 60) {
      set_error("Jacobi iteration did not converge in 60 sweeps");
      return TMF_E_HIP;
    }
    if (worst > range_floor_tol) {
      TMF_TRY(tmf_sweep_entangled(ctx, ladder[li], 1, &worst, &sat, &weak, &sweeps, &bad));
      if (sweeps >= 60) {
        set_error("Jacobi iteration did not converge in 60 sweeps");
        return TMF_E_HIP;
      }
      if (weak) continue;
    }
    if (sat) continue;
    done = true;
  }
  if (!done) {
    set_error("entanglement rank beyond the widest range finder (256 columns)");
    return TMF_E_LIMIT;
  }
  tmf_result* r = new tmf_result();
  r->ctx = ctx;
  int st = tmf_sweep_sites(ctx, &r->dims);
  if (st != TMF_OK) {
    delete r;
    return st;
  }
  const tmf_sweep_dims& d = r->dims;
  r->L = par->L, r->s_lo = par->site_lo, r->s_hi = par->site_hi, r->oc = par->ortho_center, r->cplx = par->is_complex;
  // one page-locked block for everything, 4096-byte aligned pieces
  size_t off = 0;
  auto place = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 4095) & ~(size_t)4095;
    return o;
  };
  const size_t el = (size_t)d.elem_bytes;
  size_t o[23];
  const size_t sz[23] = {(size_t)d.ncut * 8, (size_t)d.ncut * d.cap * 16, (size_t)d.ncut * d.cap * 8, (size_t)d.ncut * d.cap * 4,
                         (size_t)d.ncut * 8, (size_t)d.ncut * 8, (size_t)(d.e_tot + 1) * 8, (size_t)d.ncut * 8, (size_t)d.ncut * 4,
                         (size_t)d.ncut * 4, (size_t)d.ncut * 4, (size_t)d.ns * 4, (size_t)d.ns * 8, (size_t)d.ns * 8,
                         (size_t)(d.sec_tot + 1) * sizeof(tmf_sector), (size_t)d.ns * 8, (size_t)d.ns * 8, (size_t)d.ns * 8,
                         (size_t)d.ns * 8, (size_t)(d.bra_tot + 1) * 4, (size_t)(d.bra_tot + 1) * 4, (size_t)d.ns * el,
                         (size_t)d.out_elems * el};
  for (int i = 0; i < 23; ++i) o[i] = place(sz[i]);
  st = r->buf.ensure(off + 4096);
  if (st != TMF_OK) {
    delete r;
    return st;
  }
  void** pp = (void**)&r->p;
  for (int i = 0; i < 23; ++i) pp[i] = r->buf.p + o[i];
  int64_t ticket = 0;
  st = tmf_sweep_download(ctx, &r->p, 1, &ticket);
  if (st == TMF_OK) st = tmf_sweep_wait(ctx, ticket, r->checks, &r->n_checks);
  if (st != TMF_OK) {
    r->buf.release();
    delete r;
    return st;
  }
  r->cpos.assign((size_t)r->L + 1, -1);
  for (int64_t j = 0; j < d.ncut; ++j) r->cpos[((const int64_t*)r->p.my_cuts)[j]] = j;
  // singular always-block: the reference fails in numpy.linalg.inv (slater.py:1079 / :1086)
  const double* det = (const double*)r->p.det;
  for (int64_t j = 0; j < d.ns; ++j) {
    const double re = det[j * (r->cplx ? 2 : 1)], im = r->cplx ? det[2 * j + 1] : 0.0;
    if (!(std::isfinite(re) && std::isfinite(im)) || (re == 0.0 && im == 0.0)) {
      r->buf.release();
      delete r;
      set_error("Singular matrix");
      return TMF_E_ARG;
    }
  }
  *out = r;
  return TMF_OK;
}

extern "C" int tmf_result_dims(const tmf_result* r, tmf_sweep_dims* dims, int64_t* L, int64_t* site_lo, int64_t* site_hi) {
  if (!r) return TMF_E_ARG;
  if (dims) *dims = r->dims;
  if (L) *L = r->L;
  if (site_lo) *site_lo = r->s_lo;
  if (site_hi) *site_hi = r->s_hi;
  return TMF_OK;
}

extern "C" int tmf_result_bond(const tmf_result* r, int64_t b, tmf_bond_view* o) {
  if (!r || !o || b < 0 || b > r->L || r->cpos[(size_t)b] < 0) {
    set_error("tmf_result_bond: bond %lld is not held by this site range", (long long)b);
    return TMF_E_ARG;
  }
  const int64_t j = r->cpos[(size_t)b], cap = r->dims.cap;
  o->x = b, o->chi = ((const int64_t*)r->p.c_chi)[j], o->k = ((const int32_t*)r->p.kk_cut)[j];
  o->n_filled_left = ((const int32_t*)r->p.nfl)[j], o->n_filled_right = ((const int32_t*)r->p.nfr)[j];
  o->n_checked = ((const int64_t*)r->p.c_chk)[j];
  o->e = (const double*)r->p.e_pool + ((const int64_t*)r->p.e_off)[j];
  o->masks = (const uint64_t*)r->p.c_sets + (size_t)j * cap * 2;
  o->lam_raw = (const double*)r->p.c_lam + (size_t)j * cap;
  o->q_left = (const int32_t*)r->p.c_q + (size_t)j * cap;
  return TMF_OK;
}

extern "C" int tmf_result_site(const tmf_result* r, int64_t i, tmf_site_view* o) {
  if (!r || !o || i < r->s_lo || i >= r->s_hi) {
    set_error("tmf_result_site: site %lld is outside this site range", (long long)i);
    return TMF_E_ARG;
  }
  const int64_t j = i - r->s_lo;
  o->site = i, o->chi_bra = ((const int64_t*)r->p.chi_b)[j], o->chi_ket = ((const int64_t*)r->p.chi_k)[j];
  o->n_blocks = ((const int64_t*)r->p.nsec)[j];
  o->mode = ((const int32_t*)r->p.mode)[j], o->pad = 0;
  const double* det = (const double*)r->p.det;
  o->det_always[0] = det[j * (r->cplx ? 2 : 1)], o->det_always[1] = r->cplx ? det[2 * j + 1] : 0.0;
  const int64_t bo = ((const int64_t*)r->p.bra_off)[j];
  o->bra_p = (const int32_t*)r->p.bra_p + bo, o->bra_alpha = (const int32_t*)r->p.bra_alpha + bo;
  return TMF_OK;
}

extern "C" int tmf_result_block(const tmf_result* r, int64_t i, int64_t jb, tmf_block_view* o) {
  if (!r || !o || i < r->s_lo || i >= r->s_hi) {
    set_error("tmf_result_block: site %lld is outside this site range", (long long)i);
    return TMF_E_ARG;
  }
  const int64_t j = i - r->s_lo;
  if (jb < 0 || jb >= ((const int64_t*)r->p.nsec)[j]) {
    set_error("tmf_result_block: site %lld has %lld blocks", (long long)i, (long long)((const int64_t*)r->p.nsec)[j]);
    return TMF_E_ARG;
  }
  const tmf_sector& s = ((const tmf_sector*)r->p.sectors)[((const int64_t*)r->p.sec_off)[j] + jb];
  o->q = s.q, o->r0 = s.r0, o->r1 = s.r1, o->c0 = s.c0, o->c1 = s.c1, o->n = s.n;
  o->data = (const char*)r->p.out + (size_t)(((const int64_t*)r->p.out_off)[j] + s.out_off) * (size_t)r->dims.elem_bytes;
  return TMF_OK;
}

extern "C" int tmf_result_checks(const tmf_result* r, double* checks, int32_t* n_checks) {
  if (!r || !checks || !n_checks) return TMF_E_ARG;
  *n_checks = r->n_checks;
  memcpy(checks, r->checks, sizeof(double) * (size_t)r->n_checks);
  return TMF_OK;
}

extern "C" void tmf_result_free(tmf_result* r) {
  if (!r) return;
  r->buf.release();
  delete r;
}

#include "sweep_pf_abi.inc"
