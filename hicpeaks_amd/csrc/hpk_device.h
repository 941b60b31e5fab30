// Device helpers shared by the two kernel translation units (hpk_kernels.hip: the product; hpk_testkernels.hip: the tests'
// independent checks and dense debug outputs).  Included INSIDE each unit's anonymous namespace: every unit gets its own copy.
// No include guard on purpose.

// a band descriptor's pointer field (global address space, see HPK_GP) as an ordinary pointer
template <class T> __device__ __forceinline__ T* gptr(HPK_GP(T) p) { return (T*)p; }

// balanced value of pixel (rr, cc) on diagonal k (formed on chip in weight mode): (raw * w_r) * w_c, NaN -> 0
__device__ __forceinline__ double balanced_of(float raw, double wr, double wc) {
    double b = ((double)raw * wr) * wc;
    return (b == b) ? b : 0.0;
}

// explicit local-expected sums for pixels whose window is clipped by the matrix ends (callers.py:50-96 padding)
// (returned by value: reference parameters of a call that is not inlined live in scratch memory, and the scoring kernel
// then stored and re-loaded its expected sums around the - rare - call in every work item)
__device__ __noinline__ double2 edge_expected(const int32_t* __restrict__ m, int wi, const double* __restrict__ IR, int r,
                                              int c, int n, int num, int mw) {
    double ek = 0.0, ey = 0.0;
    for (int di = -wi; di <= wi; ++di) {
        for (int dj = -wi; dj <= wi; ++dj) {
            if (di == 0 || dj == 0) continue;
            const int adi = di < 0 ? -di : di, adj = dj < 0 ? -dj : dj;
            const int rho = adi > adj ? adi : adj;
            const int mm = m[rho];
            if (mm == 0) continue;
            const int rr = r + di, cc = c + dj, kk = cc - rr;
            if (rr < 0 || cc >= n || kk < mw || kk >= num) continue;
            const double v = (double)mm * IR[kk];
            ek += v;
            if (di > 0 && dj < 0) ey += v;
        }
    }
    return make_double2(ek, ey);
}


// ------------------------------------------------------------------ Poisson
// dpois by the saddle-point form (C. Loader, "Fast and accurate computation of binomial probabilities", 2000):
// pmf(x; lam) = exp(-stirlerr(x) - bd0(x, lam)) / sqrt(2 pi x).  sfe[0..31] = stirlerr(n) for small n (host,
// long double).
__device__ __forceinline__ double stirlerr(double x, const double* __restrict__ sfe) {
    if (x < 32.0) return sfe[(int)x];
    const double x2 = x * x;
    return (0.083333333333333333333 - (0.00277777777777777777778 - (0.00079365079365079365079365 -
            (0.000595238095238095238095238 - 0.0008417508417508417508417508 / x2) / x2) / x2) / x2) / x;
}
__device__ __forceinline__ double bd0(double x, double np) {
    if (fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        if (fabs(s) < DBL_MIN) return s;
        double ej = 2.0 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; ++j) {
            ej *= v;
            const double s1 = s + ej / (double)((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * log(x / np) + np - x;
}
__device__ __forceinline__ double dpois(double x, double lam, const double* __restrict__ sfe) {
    if (x == 0.0) return exp(-lam);
    return exp(-stirlerr(x, sfe) - bd0(x, lam)) / sqrt(6.283185307179586476925286766559 * x);
}
// 1 - cdf(k; lam) formed like the reference forms it (1 - pdtr): through the cdf rounded to f64, so that the
// far tail quantises to multiples of 2^-53 and reaches exactly 0.
// sigcap: the scoring kernel only asks "is p <= sig, and if so what is it".  For an integer k below lambda >= 1 the
// survival is at least P(X >= 2; lambda = 1) = 1 - 2/e = 0.264 (the minimum over lambda >= 1 sits just above lambda = 1 with
// k = 1), so with sig <= 0.25 such a pixel cannot pass and its lower sum - half of bhfdr's per-pixel series - is not
// formed: any value above sig will do (1.0).  The table kernel and the test helper pass sigcap = 1: every value exact.
__device__ __noinline__ double poisson_sf(double k, double lam, const double* __restrict__ sfe, double sigcap) {
    if (!(lam > 0.0)) return 0.0;
    if (k < 0.0) return 1.0;
    k = floor(k);
    if (sigcap <= 0.25 && k < lam && lam >= 1.0) return 1.0;
    double cdf;
    if (k < lam) {                         // lower sum, terms shrink going down from k
        // (a pmf that underflowed to 0 stays 0 all the way down: without the t > 0 test such a lane - k thousands below a
        // large lambda - walked every term to j = 0, and the table kernel spent 2 ms in a few hundred such waves)
        double t = dpois(k, lam, sfe), sum = t, j = k;
        while (j > 0.0 && t > 0.0) {
            t *= j / lam; j -= 1.0; sum += t;
            if (t < sum * 1e-18) break;
        }
        cdf = sum < 1.0 ? sum : 1.0;
    } else {                               // upper tail, terms shrink going up from k + 1
        double j = k + 1.0, t = dpois(j, lam, sfe), sum = t;
        for (int it = 0; it < 100000 && t > 0.0; ++it) {        // (t = 0: the tail is 0 to the last bit, no term can change it)
            j += 1.0; t *= lam / j; sum += t;
            if (t < sum * 1e-18) break;
        }
        cdf = 1.0 - sum;
    }
    return 1.0 - cdf;
}


// local expected of a pixel at its resolving step: table value in the interior; within maxww of the first rows or the
// last columns the window is clipped by the matrix end (callers.py:50-96 padding) and the value comes from the edge
// tables, indexed by the distance to that end; only a pixel clipped on both sides (chromosomes shorter than the
// band) takes the explicit loop.
//   etab [(s * 2 + fl) * (D + 1) + d]
//   eedge[(((side * W + e) * nsteps + s) * 2 + fl) * (D + 1) + d]    side 0: e = r < W,  side 1: e = n - 1 - c < W
__device__ __forceinline__ void local_expected(const HpkDevPlan* __restrict__ plan, const double* __restrict__ etab,
                                               const double* __restrict__ eedge, const double* __restrict__ IR, int step,
                                               int r, int c, int d, int n, int num, int mw, int D, int W, double& EK,
                                               double& EY) {
    const bool top = r < W, right = c >= n - W;
    if (!top && !right) {
        EK = etab[(int64_t)(step * 2) * (D + 1) + d];
        EY = etab[(int64_t)(step * 2 + 1) * (D + 1) + d];
    } else if (top != right) {
        const int side = top ? 0 : 1, e = top ? r : n - 1 - c;
        const int64_t o = ((int64_t)((side * W + e) * plan->nsteps + step) * 2) * (D + 1) + d;
        EK = eedge[o];
        EY = eedge[o + (D + 1)];
    } else {
        const double2 ee = edge_expected(plan->steps[step].m, plan->steps[step].wi, IR, r, c, n, num, mw);
        EK = ee.x; EY = ee.y;
    }
}

