// rf_host.h -- host-side table preparation shared by the C-ABI library and the
// CPU emulator (plain C++, no HIP).
#pragma once
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
