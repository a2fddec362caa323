// stats.hpp -- host-side statistics of SNPmer calling (src/kmer_comp.rs:555-593).
//
// Third-party arithmetic the reference reaches through crates that are NOT in its tree; restated
// from their published algorithms (parity unpinned):
//   * statrs 0.16.1  Binomial::cdf(x) = beta_reg(n-x, x+1, 1-p); beta_reg = regularized incomplete
//     beta by the modified-Lentz continued fraction (the Numerical-Recipes / Math.NET form statrs uses).
//   * fishers_exact 1.0.1  two-tailed p = htslib kfunc.c `kt_fisher_exact` (hypergeometric walk from
//     both tails, terms < (1+1e-8) * p_observed).
// The test-side CPU checker deliberately uses a DIFFERENT formulation (exact pmf sums) so the two
// cross-check each other; tests compare both against scipy.
#pragma once
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace savont {

// ln Gamma(x) of the SAME libm call, remembered for the integer arguments the two tests below pass (counts + 1): a table of 2^18 doubles filled on
// demand.  The arguments of a 100k-read step are a few thousand distinct integers asked for ~10^6 times; std::lgamma is ~80 ns, the table a load.
// The slots are relaxed atomics of the value's bit pattern (concurrent fills store the same bits); lgamma_r keeps the call off the global `signgam`.
inline double lgamma_memo(double x) {
    constexpr size_t N = (size_t)1 << 18;
    constexpr uint64_t UNSET = ~0ull;                               // a NaN pattern no ln Gamma takes
    static std::atomic<uint64_t>* table = [] { auto* t = new std::atomic<uint64_t>[N]; for (size_t i = 0; i < N; i++) t[i].store(UNSET, std::memory_order_relaxed); return t; }();
    int sign = 0;
    if (x >= 1.0 && x < (double)N) {
        const size_t i = (size_t)x;
        if ((double)i == x) {
            uint64_t b = table[i].load(std::memory_order_relaxed);
            double v;
            if (b == UNSET) { v = ::lgamma_r(x, &sign); memcpy(&b, &v, 8); table[i].store(b, std::memory_order_relaxed); }
            else memcpy(&v, &b, 8);
            return v;
        }
    }
    return ::lgamma_r(x, &sign);
}

inline double beta_reg(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double bt = std::exp(lgamma_memo(a + b) - lgamma_memo(a) - lgamma_memo(b) + a * std::log(x) + b * std::log1p(-x));
    const bool symm = x >= (a + 1.0) / (a + b + 2.0);
    const double eps = 1.1102230246251565e-16;
    const double fpmin = 2.2250738585072014e-308 / eps;
    double aa = a, bb = b, xx = x;
    if (symm) { xx = 1.0 - x; aa = b; bb = a; }
    const double qab = aa + bb, qap = aa + 1.0, qam = aa - 1.0;
    double c = 1.0, d = 1.0 - qab * xx / qap;
    if (std::fabs(d) < fpmin) d = fpmin;
    d = 1.0 / d;
    double h = d;
    for (int m = 1, m2 = 2; m <= 140; m++, m2 += 2) {
        double aam = m * (bb - m) * xx / ((qam + m2) * (aa + m2));
        d = 1.0 + aam * d; if (std::fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aam / c; if (std::fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d; h = h * d * c;
        aam = -(aa + m) * (qab + m) * xx / ((aa + m2) * (qap + m2));
        d = 1.0 + aam * d; if (std::fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aam / c; if (std::fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) <= eps) break;
    }
    return symm ? 1.0 - bt * h / aa : bt * h / aa;
}

// src/utils.rs:37-49: 1 - Binomial(p, n).cdf(k)
inline double binomial_test(uint64_t n, uint64_t k, double p) {
    double cdf = (k >= n) ? 1.0 : beta_reg((double)(n - k), (double)k + 1.0, 1.0 - p);
    return 1.0 - cdf;
}

// ---- htslib kfunc.c restatement -----------------------------------------------------------------
namespace detail {
inline double lbinom(int64_t n, int64_t k) {
    if (k == 0 || n == k) return 0;
    return lgamma_memo((double)n + 1) - lgamma_memo((double)k + 1) - lgamma_memo((double)(n - k) + 1);
}
inline double hypergeo(int64_t n11, int64_t n1_, int64_t n_1, int64_t n) {
    return std::exp(lbinom(n1_, n11) + lbinom(n - n1_, n_1 - n11) - lbinom(n, n_1));
}
struct HgAcc { int64_t n11, n1_, n_1, n; double p; };
inline double hypergeo_acc(int64_t n11, int64_t n1_, int64_t n_1, int64_t n, HgAcc& aux) {
    if (n1_ || n_1 || n) { aux.n11 = n11; aux.n1_ = n1_; aux.n_1 = n_1; aux.n = n; }
    else {
        if (n11 % 11 && n11 + aux.n - aux.n1_ - aux.n_1) {
            if (n11 == aux.n11 + 1) {
                aux.p *= (double)(aux.n1_ - aux.n11) / n11 * (aux.n_1 - aux.n11) / (n11 + aux.n - aux.n1_ - aux.n_1);
                aux.n11 = n11; return aux.p;
            }
            if (n11 == aux.n11 - 1) {
                aux.p *= (double)aux.n11 / (aux.n1_ - n11) * (aux.n11 + aux.n - aux.n1_ - aux.n_1) / (aux.n_1 - n11);
                aux.n11 = n11; return aux.p;
            }
        }
        aux.n11 = n11;
    }
    aux.p = hypergeo(aux.n11, aux.n1_, aux.n_1, aux.n);
    return aux.p;
}
}  // namespace detail

// table [n11 n12 / n21 n22] as passed at src/kmer_comp.rs:575-579
inline double fisher_two_tail(int64_t n11, int64_t n12, int64_t n21, int64_t n22) {
    using namespace detail;
    const int64_t n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    const int64_t mx = (n_1 < n1_) ? n_1 : n1_;
    int64_t mn = n1_ + n_1 - n; if (mn < 0) mn = 0;
    if (mn == mx) return 1.0;
    HgAcc aux{0, 0, 0, 0, 0.0};
    const double q = hypergeo_acc(n11, n1_, n_1, n, aux);
    double p = hypergeo_acc(mn, 0, 0, 0, aux), left = 0.0, right = 0.0;
    int64_t i, j;
    for (i = mn + 1; p < 0.99999999 * q && i <= mx; ++i) { left += p; p = hypergeo_acc(i, 0, 0, 0, aux); }
    --i;
    if (p < 1.00000001 * q) left += p; else --i;
    p = hypergeo_acc(mx, 0, 0, 0, aux);
    for (j = mx - 1; p < 0.99999999 * q && j >= 0; --j) { right += p; p = hypergeo_acc(j, 0, 0, 0, aux); }
    ++j;
    if (p < 1.00000001 * q) right += p; else ++j;
    double two = left + right;
    if (two > 1.0) two = 1.0;
    return two;
}

}  // namespace savont
