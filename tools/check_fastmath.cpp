// Host check of csrc/glrm_fastmath.hpp against libm (long double where it matters).  g++ -O2 -ffp-contract=off tools/check_fastmath.cpp -o /tmp/cfm && /tmp/cfm
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <initializer_list>
#include "../lowrankmodels.jl_amd/csrc/glrm_fastmath.hpp"

static uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double unif(uint64_t i) { return ((double)(mix(i) >> 11) + 0.5) / 9007199254740992.0; }

int main() {
  const int N = 4000000;
  double e_exp = 0, e_l1p = 0, e_L = 0, e_dL = 0, e_Lref = 0, a_Lref = 0;
  long nonzero_where_ref_zero = 0;
  for (int i = 0; i < N; ++i) {
    const double x = -745.0 + 1454.0 * unif(i);
    const long double ex = expl((long double)x);
    const double err = (double)fabsl(((long double)glrm::fm_exp(x) - ex) / ex);
    if (ex > 1e-300L && err > e_exp) e_exp = err;
    const double xs = -40.0 + 80.0 * unif(i + 77777777);   // the range logistic arguments live in
    const long double es = expl((long double)xs);
    const double errs = (double)fabsl(((long double)glrm::fm_exp(xs) - es) / es);
    if (errs > e_exp) e_exp = errs;
    const double t = unif(i + 123456789);
    const double w1 = 1.0 + t * (i & 1 ? 1.0 : ldexp(1.0, -(int)(mix(i) % 50)));            // [1, 2], dense near 1
    const double w2 = ldexp(1.0 + t, (int)(mix(i + 5) % 1023));                              // the whole range above 1
    for (double w : {w1, w2}) {
      const long double l1 = logl((long double)w);
      if (l1 > 0) { const double err2 = (double)fabsl(((long double)glrm::fm_log_ge1(w) - l1) / l1); if (err2 > e_l1p) e_l1p = err2; }
    }
    // logistic: u in [-45, 45], both labels
    const double u = -45.0 + 90.0 * unif(i + 424242), aa = (i & 1) ? 1.0 : -1.0, sc = 0.7;
    double L, dL;
    glrm::fm_logistic<true>(sc, aa, u, L, dL);
    const long double zz = (long double)aa * u;
    const long double dLx = -aa * sc / (1.0L + expl(zz));
    const double r2 = (double)fabsl((dL - dLx) / dLx);
    if (r2 > e_dL) e_dL = r2;
    const double Lref = sc * log(1 + exp(-aa * u)), dLref = -aa * sc / (1 + exp(aa * u)); // the reference's literal formulas in double
    const double ad = fabs(L - Lref);
    if (ad > a_Lref) a_Lref = ad;
    if (Lref > 0) { const double rr = ad / Lref; if (rr > e_Lref) e_Lref = rr; } else if (L != 0) ++nonzero_where_ref_zero;
    const double r1 = dLref != 0 ? fabs((dL - dLref) / dLref) : 0; if (r1 > e_L) e_L = r1;
  }
  // fm_log on (0, Inf): the arguments of MultinomialOrdinalLoss are 1 - e and e_hi - e_lo in (0, 1); the whole exponent range incl. denormals
  double e_log = 0, a_log1 = 0;
  for (int i = 0; i < N; ++i) {
    const double t = unif(i + 31337);
    const double x1 = t;                                                        // (0, 1) uniform
    const double x2 = ldexp(0.5 + 0.5 * t, -(int)(mix(i + 9) % 1074));          // down to the denormals
    const double x3 = 1.0 - ldexp(t, -(int)(mix(i + 3) % 40));                  // just below 1
    for (double x : {x1, x2, x3}) {
      if (!(x > 0)) continue;
      const long double l = logl((long double)x);
      const double v = glrm::fm_log(x);
      if (l != 0) { const double r = (double)fabsl(((long double)v - l) / l); if (r > e_log) e_log = r; }
      const double ad = (double)fabsl((long double)v - l); if (x > 0.25 && ad > a_log1) a_log1 = ad;
    }
  }
  printf("fm_log on (0, 1] incl. denormals: max rel err %.3g; max abs err on (0.25, 1): %.3g; log(0) %g log(-1) %g log(Inf) %g log(NaN) %g log(1) %g log(4.9e-324) %.17g (libm %.17g)\n",
         e_log, a_log1, glrm::fm_log(0.0), glrm::fm_log(-1.0), glrm::fm_log(INFINITY), glrm::fm_log(NAN), glrm::fm_log(1.0), glrm::fm_log(4.9e-324), log(4.9e-324));
  double L, dL;
  glrm::fm_logistic<true>(1.0, 1.0, NAN, L, dL);
  printf("max rel err vs long double: exp %.3g  log (w >= 1) %.3g  logistic derivative %.3g\n", e_exp, e_l1p, e_dL);
  printf("logistic vs the reference's literal double formulas: loss max abs %.3g, max rel %.3g (nonzero where the reference is exactly 0: %ld); derivative max rel %.3g; NaN in -> %g %g\n", a_Lref, e_Lref, nonzero_where_ref_zero, e_L, L, dL);
  glrm::fm_logistic<true>(1.0, 1.0, -800.0, L, dL);
  printf("z = -800 (exp overflows in the reference): L %g (reference Inf) dL %g (reference -1)\n", L, dL);
  glrm::fm_logistic<true>(1.0, 1.0, 800.0, L, dL);
  printf("z = +800: L %g (reference 0) dL %g (reference -0)\n", L, dL);
  glrm::fm_logistic<true>(1.0, 1.0, 40.0, L, dL);
  printf("z = +40: L %g (reference %g)\n", L, log(1 + exp(-40.0)));
  printf("exp(800) %g exp(-800) %g exp(0) %.17g exp(1) %.17g\n", glrm::fm_exp(800), glrm::fm_exp(-800), glrm::fm_exp(0.0), glrm::fm_exp(1.0));
  glrm::fm_logistic<true>(1.0, 1.0, 1.0, L, dL);
  printf("logistic(1, true) %.17g (reference KAT 0.31326168751822286)  grad %.17g\n", L, dL);
  glrm::fm_logistic<true>(1.0, -1.0, 1.0, L, dL);
  printf("logistic(1, false) %.17g (reference KAT 1.3132616875182228)\n", L);
  return 0;
}
