// rf_host.h -- host-side table preparation shared by the C-ABI library and the
// CPU emulator (plain C++, no HIP).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "rf_core.h"

namespace rf {

struct SigmaTableHost {
  std::vector<double> xt, st, sl;
  std::vector<int> bin;
  double x0 = 0, inv_dx = 0;
};

// log10k / sigma are the float64 tables the Python host computes exactly as
// powertools.py:153-154.  Slopes use the interpolator's own expression
// (y_hi - y_lo) / (x_hi - x_lo); `bin` is a uniform acceleration grid in x.
inline void build_sigma_table(const double* log10k, const double* sigma, int n, SigmaTableHost& t, int nbins = 2048) {
  t.xt.assign(log10k, log10k + n);
  t.st.assign(sigma, sigma + n);
  t.sl.resize(n > 1 ? n - 1 : 1);
  for (int j = 0; j + 1 < n; ++j) t.sl[j] = (t.st[j + 1] - t.st[j]) / (t.xt[j + 1] - t.xt[j]);
  t.x0 = t.xt[0];
  const double span = t.xt[n - 1] - t.xt[0];
  t.inv_dx = span > 0 ? nbins / span : 0.0;
  t.bin.resize(nbins);
  int j = 0;
  for (int b = 0; b < nbins; ++b) {
    const double edge = t.x0 + (t.inv_dx > 0 ? b / t.inv_dx : 0.0);
    while (j + 1 < n - 1 && t.xt[j + 1] <= edge) ++j;
    t.bin[b] = j;
  }
}

// Per-bin records of the fast float32 sigma lookup over [xlo, xhi] (the grid's own log10 k range,
// padded).  Tries the smallest power-of-two bin count >= 128 with at most one knot per bin (so that
// the records fit LDS when possible); false if even 65536 bins do not separate the knots -- the
// caller then uses the exact kernel.
inline bool build_fast_records(const SigmaTableHost& t, double xlo, double xhi, std::vector<FastRec>& rec,
                               double& x0, double& dx) {
  const int n = (int)t.xt.size();
  // cells outside the table get sigma = 0 (powertools.py:155-157): when the requested range reaches a
  // table edge, one all-zero guard bin is put beyond it (x below / above clamps into it)
  const bool guard_lo = xlo <= t.xt[0], guard_hi = xhi >= t.xt[n - 1];
  xlo = std::max(xlo, t.xt[0]);
  xhi = std::min(xhi, t.xt[n - 1]);
  if (!(xhi > xlo)) return false;
  auto sigma_at = [&](double x, int j) { return t.sl[j] * (x - t.xt[j]) + t.st[j]; };
  for (int nbins = 126; nbins <= 65536; nbins = (nbins + 2) * 2 - 2) {   // +2 guard bins stay within 2^k
    dx = (xhi - xlo) / nbins;
    rec.assign(nbins, FastRec());
    bool ok = true;
    int j = 0;
    for (int b = 0; b < nbins && ok; ++b) {
      const double e0 = xlo + b * dx, e1 = xlo + (b + 1) * dx;
      while (j + 1 < n - 1 && t.xt[j + 1] <= e0) ++j;          // interval containing e0
      FastRec& r = rec[b];
      r.v0 = (float)sigma_at(e0, j);
      r.sa = (float)(t.sl[j] * dx);
      r.fs = 2.0f;
      r.ds = 0.0f;
      if (j + 1 < n - 1 && t.xt[j + 1] < e1) {                   // a knot inside the bin
        r.fs = (float)((t.xt[j + 1] - e0) / dx);
        r.ds = (float)((t.sl[j + 1] - t.sl[j]) * dx);
        if (j + 2 < n - 1 && t.xt[j + 2] < e1) ok = false;       // a second one: need finer bins
      }
    }
    if (ok) {
      x0 = xlo;
      if (guard_lo) { rec.insert(rec.begin(), FastRec{0.0f, 0.0f, 2.0f, 0.0f}); x0 -= dx; }
      if (guard_hi) rec.push_back(FastRec{0.0f, 0.0f, 2.0f, 0.0f});
      return true;
    }
  }
  return false;
}

// exp(+2 pi i q / n), q in [0, n), evaluated in double
template <typename T> inline std::vector<cplx<T>> make_twiddles(int n) {
  std::vector<cplx<T>> w(n);
  for (int q = 0; q < n; ++q) {
    // exact symmetries keep the table clean: reduce to the first octant
    const double a = 2.0 * M_PI * (double)q / (double)n;
    w[q].x = (T)std::cos(a);
    w[q].y = (T)std::sin(a);
  }
  if (n % 4 == 0) {  // exact values on the axes
    w[0] = mk<T>(1, 0); w[n / 4] = mk<T>(0, 1); w[n / 2] = mk<T>(-1, 0); w[3 * n / 4] = mk<T>(0, -1);
  } else if (n % 2 == 0) {
    w[0] = mk<T>(1, 0); w[n / 2] = mk<T>(-1, 0);
  }
  return w;
}

}  // namespace rf
