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

// Per-bin records of the fast float32 sigma lookup, restricted to [xlo, xhi] (the
// grid's own log10 k range, padded).  Returns false if no bin count up to 65536
// separates the knots (more than one knot per bin) -- the caller then uses the
// exact kernel.
inline bool build_fast_records(const SigmaTableHost& t, double xlo, double xhi, std::vector<FastRec>& rec,
                               float& x0, float& inv_dx, float& xmin, float& xmax) {
  const int n = (int)t.xt.size();
  xlo = std::max(xlo, t.xt[0]);
  xhi = std::min(xhi, t.xt[n - 1]);
  if (!(xhi > xlo)) return false;
  for (int nbins = 512; nbins <= 65536; nbins *= 2) {
    const double dx = (xhi - xlo) / nbins;
    rec.assign(nbins, FastRec());
    bool ok = true;
    int j = 0;
    for (int b = 0; b < nbins && ok; ++b) {
      const double e0 = xlo + b * dx, e1 = xlo + (b + 1) * dx;
      while (j + 1 < n - 1 && t.xt[j + 1] <= e0) ++j;          // interval containing e0
      FastRec& r = rec[b];
      const int ja = j;
      int jb = j;
      r.xs = INFINITY;
      if (ja + 1 < n - 1 && t.xt[ja + 1] < e1) {                 // a knot inside the bin
        jb = ja + 1;
        r.xs = (float)t.xt[jb];
        if (jb + 1 < n - 1 && t.xt[jb + 1] < e1) ok = false;     // a second one: need finer bins
      }
      r.xa = (float)t.xt[ja]; r.sa = (float)t.st[ja]; r.sla = (float)t.sl[ja];
      r.xb = (float)t.xt[jb]; r.sb = (float)t.st[jb]; r.slb = (float)t.sl[jb];
      r.pad = 0.0f;
    }
    if (ok) {
      x0 = (float)xlo; inv_dx = (float)(1.0 / dx);
      xmin = (float)t.xt[0]; xmax = (float)t.xt[n - 1];
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
