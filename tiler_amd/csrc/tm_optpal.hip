// tm_optpal.hip -- A11: OptimizePalettes (tilingencoder.pas:4309-4432) on the host.  P x PaletteSize colours only, so it
// stays CPU code: each palette's colour slots are permuted to maximise the cross-palette
// per-slot spread, searched with the reference's Powell/Brent minimiser (powell.pas, a scipy port) over 15 rank
// variables, sweeping all palettes until the mean objective stops improving.  Double arithmetic, sequential: the same
// libm sqrt and the same evaluation order as the reference, including powell.pas's dynamic-array aliasing of the
// direction set (direc[n-1] := direc1 shares storage in FreePascal).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

// the objective and the line functions are template parameters (no std::function indirection: the whole search inlines)

double sign_of(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }

struct Bracketed { double xa, xb, xc; };

template <class Fn1> Bracketed bracket(const Fn1 &f, double xa, double xb) {  // Bracket, powell.pas:56-147
  const double gold = (1 + std::sqrt(5.0)) / 2, small = 1e-21, grow_limit = 110;
  double fa = f(xa), fb = f(xb);
  if (fa < fb) { std::swap(xa, xb); std::swap(fa, fb); }
  double xc = xb + gold * (xb - xa), fc = f(xc);
  int iter = 0;
  while (fc < fb) {
    const double tmp1 = (xb - xa) * (fb - fc), tmp2 = (xb - xc) * (fb - fa), val = tmp2 - tmp1;
    const double denom = std::fabs(val) < small ? 2 * small : 2 * val;
    double w = xb - ((xb - xc) * tmp2 - (xb - xa) * tmp1) / denom;
    const double wlim = xb + grow_limit * (xc - xb);
    if (iter > 1000) break;  // the reference raises here; bounded objectives never get this far
    ++iter;
    double fw = 0;
    if ((w - xc) * (xb - w) > 0) {
      fw = f(w);
      if (fw < fc) { xa = xb; xb = w; fa = fb; fb = fw; break; }
      if (fw > fb) { xc = w; fc = fw; break; }
      w = xc + gold * (xc - xb);
      fw = f(w);
    } else if ((w - wlim) * (wlim - xc) >= 0) {
      w = wlim;
      fw = f(w);
    } else if ((w - wlim) * (xc - w) > 0) {
      fw = f(w);
      if (fw < fc) { xb = xc; xc = w; w = xc + gold * (xc - xb); fb = fc; fc = fw; fw = f(w); }
    } else {
      w = xc + gold * (xc - xb);
      fw = f(w);
    }
    xa = xb; xb = xc; xc = w;
    fa = fb; fb = fc; fc = fw;
  }
  if (xa > xc) { std::swap(xa, xc); std::swap(fa, fc); }
  return {xa, xb, xc};
}

struct LineMin { double x, fx; };

template <class Fn1> LineMin brent(const Fn1 &f, double xtol, int maxiter) {  // Brent + BrentHelper, powell.pas:149-266, bracket seed (0, 1)
  const double cg = (3 - std::sqrt(5.0)) / 2;
  const Bracketed br = bracket(f, 0, 1);
  double a = br.xa, x = br.xb, b = br.xc, fx = f(x);
  if (a > b) std::swap(a, b);
  double w = x, v = x, fw = fx, fv = fx, deltax = 0, rat = 0;
  for (int iter = 0; iter < maxiter; ++iter) {
    const double xmid = 0.5 * (a + b);
    if (std::fabs(x - xmid) <= 2 * xtol - 0.5 * (b - a)) break;
    if (std::fabs(deltax) <= xtol) {
      deltax = x >= xmid ? a - x : b - x;
      rat = cg * deltax;
    } else {
      const double tmp1 = (x - w) * (fx - fv);
      double tmp2 = (x - v) * (fx - fw);
      double p = (x - v) * tmp2 - (x - w) * tmp1;
      tmp2 = 2 * (tmp2 - tmp1);
      if (tmp2 > 0) p = -p;
      tmp2 = std::fabs(tmp2);
      const double dx_temp = deltax;
      deltax = rat;
      if (p > tmp2 * (a - x) && p < tmp2 * (b - x) && std::fabs(p) < std::fabs(0.5 * tmp2 * dx_temp)) {
        rat = p / tmp2;
        const double u = x + rat;
        if (u - a < xtol || b - u < xtol) rat = sign_of(xmid - x) * xtol;
      } else {
        deltax = x >= xmid ? a - x : b - x;
        rat = cg * deltax;
      }
    }
    const double u = std::fabs(rat) > xtol ? x + rat : x + sign_of(rat) * xtol;
    const double fu = f(u);
    if (fu > fx) {
      if (u < x) a = u; else b = u;
      if (fu <= fw || w == x) { v = w; w = u; fv = fw; fw = fu; }
      else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    } else {
      if (u >= x) a = x; else b = x;
      v = w; w = x; x = u;
      fv = fw; fw = fx; fx = fu;
    }
  }
  return {x, fx};
}

template <class FnN> double linesearch(const FnN &f, std::vector<double> &p, double *xi, double xtol, std::vector<double> &ray) {  // LinesearchPowell, powell.pas:285-314
  const size_t n = p.size();
  double sos = 0;
  for (size_t i = 0; i < n; i++) sos += xi[i] * xi[i];
  const double sqsos = std::sqrt(sos);
  double atol = 1.0;
  if (sqsos != 0) atol = 5 * xtol / sqsos;
  atol = std::min(0.1, atol);
  const LineMin m = brent([&](double t) {
    for (size_t i = 0; i < n; i++) ray[i] = p[i] + t * xi[i];
    return f(ray);
  }, atol, 100);
  for (size_t i = 0; i < n; i++) { xi[i] = xi[i] * m.x; p[i] = p[i] + xi[i]; }
  return m.fx;
}

template <class FnN> double powell_minimize(const FnN &f, std::vector<double> &x, double scale, double xtol, double ftol, int maxiter) {
  // PowellMinimize, powell.pas:316-384; rows are pointers because the reference's rows alias after a replacement
  const size_t n = x.size();
  std::vector<double> store((n + 1) * n, 0.0), tmp(n), x1(x), ray(n);
  std::vector<double *> direc(n);
  for (size_t i = 0; i < n; i++) { direc[i] = &store[i * n]; direc[i][i] = scale; }
  double *direc1 = &store[n * n];
  double fval = f(x);
  for (int iter = 0;;) {
    const double fx = fval;
    double delta = 0;
    size_t bigind = 0;
    for (size_t i = 0; i < n; i++) {
      const double before = fval;
      fval = linesearch(f, x, direc[i], xtol, ray);
      if (before - fval > delta) { delta = before - fval; bigind = i; }
    }
    ++iter;
    if (fx - fval <= ftol || iter >= maxiter) break;
    for (size_t i = 0; i < n; i++) { direc1[i] = x[i] - x1[i]; tmp[i] = x[i] + direc1[i]; x1[i] = x[i]; }
    const double fx2 = f(tmp);
    if (fx > fx2) {
      double t = 2 * (fx + fx2 - 2 * fval), temp = fx - fval - delta;
      t = t * temp * temp;
      temp = fx - fx2;
      t = t - delta * temp * temp;
      if (t < 0) {
        fval = linesearch(f, x, direc1, xtol, ray);
        direc[bigind] = direc[n - 1];
        direc[n - 1] = direc1;
      }
    }
  }
  return fval;
}

// Helper threads that outlive the call: the palettes of one sweep are independent tasks (DoPal runs under
// ProcThreadPool.DoParallelLocalProc in the reference, tilingencoder.pas:4415), but threads made per call lose to their own creation
// cost inside a process that already runs other runtimes' threads -- so they are made once, sleep on a condition variable between
// sweeps and are joined when the library goes away.
class HelperPool {
 public:
  static HelperPool &get() { static HelperPool p; return p; }
  template <class F> void run(int ntasks, const F &f) {  // f(task) for task in [0, ntasks); the caller works too
    if (ntasks <= 1 || workers_.empty()) { for (int t = 0; t < ntasks; t++) f(t); return; }
    std::lock_guard<std::mutex> one_job(run_m_);  // encoders on other threads take their turn
    std::function<void(int)> job = f;
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job; next_ = 0; total_ = ntasks; pending_ = ntasks; generation_.fetch_add(1, std::memory_order_release);
    }
    cv_work_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr; total_ = 0;
  }

 private:
  HelperPool() {
    const char *env = getenv("TM_HOST_THREADS");
    int n = env ? atoi(env) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    for (int i = 0; i < n - 1; i++) workers_.emplace_back([this] { loop(); });
  }
  ~HelperPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; stop_flag_.store(true); }
    cv_work_.notify_all();
    for (auto &t : workers_) t.join();
  }
  void drain() {  // tasks are claimed under the lock (they run for tens of microseconds: the lock is noise)
    for (;;) {
      int t;
      const std::function<void(int)> *j;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (next_ >= total_) return;
        t = next_++;
        j = job_;
      }
      (*j)(t);
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) cv_done_.notify_all();
    }
  }
  void loop() {
    unsigned seen = 0;
    for (;;) {
      // A search is several sweeps back to back, each a job of its own: a helper that went to sleep on the condition variable after every job
      // paid a futex wake-up per sweep (and now and then a scheduler's delay of milliseconds, which the caller then waited out at the end of the
      // sweep: PreparePalettes' 0.5 ms host phase measured 5 ms about once in six steps).  So a helper first looks for the next job for ~150 us.
      const auto t0 = std::chrono::steady_clock::now();
      while (generation_.load(std::memory_order_acquire) == seen && !stop_flag_.load(std::memory_order_acquire) &&
             std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(150)) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return stop_ || generation_.load(std::memory_order_relaxed) != seen; });
        if (stop_) return;
        seen = generation_.load(std::memory_order_relaxed);
      }
      drain();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_, run_m_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(int)> *job_ = nullptr;
  int next_ = 0, total_ = 0, pending_ = 0;
  std::atomic<unsigned> generation_{0};
  std::atomic<bool> stop_flag_{false};
  bool stop_ = false;
};

}  // namespace

int optimize_palettes_host(std::vector<int32_t> &pals, int pal_count, int pal_size, int *sweeps_out) {
  TM_CHECK(pal_size >= 2 && pal_size <= 64 && pal_count >= 1 && (size_t)pal_count * pal_size == pals.size(), TM_E_INVAL,
           "optimize_palettes: bad shape");
  std::vector<int32_t> newpal(pals.size());
  std::vector<double> fbest(pal_count);
  uint64_t mean[3] = {0, 0, 0};
  for (int32_t c : pals) { mean[0] += (uint32_t)c & 0xff; mean[1] += ((uint32_t)c >> 8) & 0xff; mean[2] += ((uint32_t)c >> 16) & 0xff; }
  for (auto &m : mean) m /= (uint64_t)pal_size;  // "mean of all palette colors", divided by PaletteSize (4401-4403)
  int sweeps = 0;
  double fsum = 0, prev = 0;
  do {
    prev = std::max(fsum, prev);
    ++sweeps;
    // DoPal (4315-4375): every palette against the others as they stood before the sweep -- independent tasks (4415)
    HelperPool::get().run(pal_count, [&](int a) {
      uint64_t acc[3][64] = {};
      for (int p = 0; p < pal_count; p++)
        if (p != a)
          for (int i = 0; i < pal_size; i++) {
            const uint32_t c = (uint32_t)pals[(size_t)p * pal_size + i];
            acc[0][i] += c & 0xff; acc[1][i] += (c >> 8) & 0xff; acc[2][i] += (c >> 16) & 0xff;
          }
      auto objective = [&](const std::vector<double> &x) {  // PowellOP, 4265-4307
        struct Item { int count, index; } perm[64];
        perm[0] = {0, 0};
        for (int i = 1; i < pal_size; i++) perm[i] = {(int)llrint(x[i - 1] * 1000), i};
        for (int i = 1; i < pal_size; i++) {  // insertion sort by (count, index): a total order, so any sort gives this result
          const Item it = perm[i];
          int j = i - 1;
          while (j >= 0 && (perm[j].count != it.count ? perm[j].count > it.count : perm[j].index > it.index)) { perm[j + 1] = perm[j]; j--; }
          perm[j + 1] = it;
        }
        uint64_t sd[3] = {0, 0, 0};
        for (int i = 0; i < pal_size; i++) {
          const uint32_t c = (uint32_t)pals[(size_t)a * pal_size + perm[i].index];
          newpal[(size_t)a * pal_size + i] = (int32_t)c;
          const uint64_t ch[3] = {c & 0xff, (c >> 8) & 0xff, (c >> 16) & 0xff};
          for (int k = 0; k < 3; k++) { const uint64_t d = acc[k][i] + ch[k] - mean[k]; sd[k] += d * d; }  // UInt64 wrap == signed square
        }
        return -((299 * std::sqrt((double)sd[0] / pal_size) + 587 * std::sqrt((double)sd[1] / pal_size) +
                  114 * std::sqrt((double)sd[2] / pal_size)) / 1000);
      };
      std::vector<double> x(pal_size - 1);
      for (int i = 1; i < pal_size; i++) x[i - 1] = i;
      powell_minimize(objective, x, 1.0, 1.0, 1.0, 2147483647);
      fbest[a] = -objective(x);
    });
    fsum = 0;
    for (double v : fbest) fsum += v;
    pals = newpal;
    fsum /= pal_count;
  } while (!(fsum <= prev));
  if (sweeps_out) *sweeps_out = sweeps;
  return TM_OK;
}

}  // namespace tmx

extern "C" int tm_optimize_palettes_host(int32_t *palettes, int pal_count, int pal_size, int *sweeps) {
  if (!palettes) { tmx::set_error("null palettes"); return TM_E_INVAL; }
  std::vector<int32_t> p(palettes, palettes + (size_t)pal_count * pal_size);
  const int rc = tmx::optimize_palettes_host(p, pal_count, pal_size, sweeps);
  if (rc == TM_OK) std::copy(p.begin(), p.end(), palettes);
  return rc;
}
