/* CPU oracle, plain C: the compiled-lasso ADMM sweep exactly as the reference executes it,
 * unrolled.  TEST INFRASTRUCTURE: used only by tests/ (checked against the
 * generic numpy oracle) and by bench.py's cpu_baseline leg; never linked into the product.
 *
 * Reference path restated (all fp64, single thread like the reference,
 * tools/run_benchmarks.sh:15-17):
 *   driver        src/epsilon/algorithms/prox_admm.cc:131-169 (Gauss-Seidel sweep)
 *   SUM_SQUARE    src/epsilon/prox/sum_square.cc:31-33 -> vector/block_cholesky.cc:135-137:
 *                 forward-sub  t = A v        (dgemv N, linear/dense_matrix_impl.cc:63)
 *                 block scale  w = Minv (b-t) (dgemv on the cached explicit inverse)
 *                 back-sub     x0 = v + 2 A^T w (dgemv T)
 *   NORM_1        src/epsilon/prox/scaled_zone.cc:90-101 with alpha=beta=1, M=C=0
 *   residuals     src/epsilon/algorithms/prox_admm.cc:178-217
 * with A_ = [I, -I], b_ empty, so y0 = x0, y1 = -x1.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads for the three mat-vecs (1 = the reference's own configuration; bench.py also times
 * all host cores).  The arithmetic per output element is the same chain of operations for any
 * thread count - rows of y (N form) and columns (T form) are dealt to the threads whole - so the
 * iterates do not depend on it. */
static int g_threads = 1;
void oracle_set_threads(int t) { g_threads = t < 1 ? 1 : t; }
int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void gemv_n(int m, int n, const double* A, const double* x, double* y) {
  /* each thread owns a block of rows and streams all columns through it */
#pragma omp parallel num_threads(g_threads)
  {
    int nt = 1, id = 0;
#ifdef _OPENMP
    nt = omp_get_num_threads();
    id = omp_get_thread_num();
#endif
    const int per = ((m + nt - 1) / nt + 7) / 8 * 8;
    const int i0 = id * per < m ? id * per : m;
    const int i1 = i0 + per < m ? i0 + per : m;
    if (i1 > i0) {
      memset(y + i0, 0, sizeof(double) * (size_t)(i1 - i0));
      for (int j = 0; j < n; ++j) {
        const double xj = x[j];
        const double* a = A + (size_t)j * m;
        for (int i = i0; i < i1; ++i) y[i] += a[i] * xj;
      }
    }
  }
}

static void gemv_t(int m, int n, const double* A, const double* x, double* y) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int j = 0; j < n; ++j) {
    const double* a = A + (size_t)j * m;
    double s = 0;
    for (int i = 0; i < m; ++i) s += a[i] * x[i];
    y[j] = s;
  }
}

/* G = A A^T (m x m, both triangles) for column-major A (m x n): the contraction the reference
 * hands to dgemm_ at Init (linear/linear_map_multiply.cc:14-37).  Plain cache-blocked C - 64 x 64
 * blocks of G accumulated over all columns - used by bench.py to time the CPU's Init on a
 * reduced instance; threads as in the mat-vecs. */
void gram_aat(int m, int n, const double* A, double* G) {
  const int B = 64;
  const int nb = (m + B - 1) / B;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic)
  for (int t = 0; t < nb * nb; ++t) {
    const int bi = t / nb, bj = t % nb;
    if (bj > bi) continue;
    const int i0 = bi * B, j0 = bj * B;
    const int ni = m - i0 < B ? m - i0 : B, nj = m - j0 < B ? m - j0 : B;
    double acc[64][64];
    for (int q = 0; q < nj; ++q)
      for (int p = 0; p < ni; ++p) acc[q][p] = 0;
    for (int k = 0; k < n; ++k) {
      const double* ai = A + (size_t)k * m + i0;
      const double* aj = A + (size_t)k * m + j0;
      for (int q = 0; q < nj; ++q) {
        const double v = aj[q];
        for (int p = 0; p < ni; ++p) acc[q][p] += ai[p] * v;
      }
    }
    for (int q = 0; q < nj; ++q)
      for (int p = 0; p < ni; ++p) {
        G[(size_t)(j0 + q) * m + i0 + p] = acc[q][p];
        G[(size_t)(i0 + p) * m + j0 + q] = acc[q][p];
      }
  }
}

static double nrm2(int n, const double* x) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += x[i] * x[i];
  return sqrt(s);
}

/* Runs sweeps iter0 .. iter0+k-1 (stops early when OPTIMAL at an epoch check).
 * State x0,x1,u,y0,y1 (n each) is updated in place; Minv = (I + 2 A A^T)^-1 (m x m).
 * resid = {r_norm, s_norm, eps_pri, eps_dual}; returns the number of sweeps executed and
 * sets *optimal. */
int lasso_admm_run(int m, int n, const double* A, const double* Minv, const double* b,
                   double lam, double* x0, double* x1, double* u, double* y0, double* y1,
                   int iter0, int k, double abs_tol, double rel_tol, int epoch, double* resid,
                   int* optimal) {
  double* t = (double*)malloc(sizeof(double) * (size_t)m);
  double* w = (double*)malloc(sizeof(double) * (size_t)m);
  double* g = (double*)malloc(sizeof(double) * (size_t)n);
  double* y1_prev = (double*)malloc(sizeof(double) * (size_t)n);
  int done = 0;
  *optimal = 0;
  for (int it = iter0; it < iter0 + k; ++it) {
    memcpy(y1_prev, y1, sizeof(double) * (size_t)n);
    for (int j = 0; j < n; ++j) u[j] = (u[j] - y0[j]) - y1[j];
    /* term 0: sum_square */
    for (int j = 0; j < n; ++j) u[j] += y0[j];
    gemv_n(m, n, A, u, t);
    for (int i = 0; i < m; ++i) t[i] = b[i] - t[i];
    gemv_t(m, m, Minv, t, w); /* Minv symmetric */
    gemv_t(m, n, A, w, g);
    for (int j = 0; j < n; ++j) {
      x0[j] = u[j] + 2.0 * g[j];
      y0[j] = x0[j];
      u[j] -= y0[j];
    }
    /* term 1: norm_1, v = -u */
    for (int j = 0; j < n; ++j) {
      u[j] += y1[j];
      const double v = -u[j];
      double x;
      if (fabs(v) <= 0) x = v;
      else if (v > lam) x = v - lam;
      else if (v < -lam) x = v + lam;
      else x = 0;
      x1[j] = x;
      y1[j] = -x;
      u[j] -= y1[j];
    }
    ++done;
    if (it % epoch == 0) {
      double r2 = 0, s2 = 0;
      for (int j = 0; j < n; ++j) {
        const double d = x0[j] - x1[j];
        const double e = y1[j] - y1_prev[j];
        r2 += d * d;
        s2 += e * e;
      }
      const double nx0 = nrm2(n, x0), nx1 = nrm2(n, x1), nu = nrm2(n, u);
      resid[0] = sqrt(r2);
      resid[1] = sqrt(s2);
      resid[2] = abs_tol * sqrt((double)n) + rel_tol * (nx0 > nx1 ? nx0 : nx1);
      resid[3] = abs_tol * sqrt(2.0 * n) + rel_tol * sqrt(2.0) * nu;
      if (resid[0] <= resid[2] && resid[1] <= resid[3]) {
        *optimal = 1;
        break;
      }
    }
  }
  free(t);
  free(w);
  free(g);
  free(y1_prev);
  return done;
}

/* Exact 1-D total-variation prox (Johnson's DP; same restatement as
 * oracle/epsilon_oracle.py:tv1d_prox, fp64) for sizes where python loops are too slow. */
void tv1d_prox(int n, const double* y, double lam, double* beta) {
  if (n == 0) return;
  if (n == 1 || lam == 0) {
    memcpy(beta, y, sizeof(double) * (size_t)n);
    return;
  }
  double* x = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  double* a = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  double* b = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  double* tm = (double*)malloc(sizeof(double) * (size_t)(n - 1));
  double* tp = (double*)malloc(sizeof(double) * (size_t)(n - 1));
  tm[0] = -lam + y[0];
  tp[0] = lam + y[0];
  long l = n - 1, r = n;
  x[l] = tm[0];
  x[r] = tp[0];
  a[l] = 1;
  b[l] = -y[0] + lam;
  a[r] = -1;
  b[r] = y[0] + lam;
  double afirst = 1, bfirst = -lam - y[1], alast = -1, blast = -lam + y[1];
  for (long k = 1; k < n - 1; ++k) {
    double alo = afirst, blo = bfirst;
    long lo;
    for (lo = l; lo <= r; ++lo) {
      if (alo * x[lo] + blo > -lam) break;
      alo += a[lo];
      blo += b[lo];
    }
    tm[k] = (-lam - blo) / alo;
    l = lo - 1;
    x[l] = tm[k];
    double ahi = alast, bhi = blast;
    long hi;
    for (hi = r; hi >= l; --hi) {
      if (-ahi * x[hi] - bhi < lam) break;
      ahi += a[hi];
      bhi += b[hi];
    }
    tp[k] = (lam + bhi) / (-ahi);
    r = hi + 1;
    x[r] = tp[k];
    a[l] = alo;
    b[l] = blo + lam;
    a[r] = ahi;
    b[r] = bhi + lam;
    afirst = 1;
    bfirst = -lam - y[k + 1];
    alast = -1;
    blast = -lam + y[k + 1];
  }
  double alo = afirst, blo = bfirst;
  for (long lo = l; lo <= r; ++lo) {
    if (alo * x[lo] + blo > 0) break;
    alo += a[lo];
    blo += b[lo];
  }
  beta[n - 1] = -blo / alo;
  for (long k = n - 2; k >= 0; --k) {
    if (beta[k + 1] > tp[k]) beta[k] = tp[k];
    else if (beta[k + 1] < tm[k]) beta[k] = tm[k];
    else beta[k] = beta[k + 1];
  }
  free(x);
  free(a);
  free(b);
  free(tm);
  free(tp);
}
