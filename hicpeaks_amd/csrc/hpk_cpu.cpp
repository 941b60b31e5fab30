// Back-end #0 of the C ABI: the path of hicpeaks/callers.py:44-590 on host threads (hpk_cpu.h).  Dense-band form of the as-coded
// algorithm: per candidate the Chebyshev rings of its window are added cell by cell from a zero-padded copy
// of its row block in true matrix coordinates - no summed-area table, so no cancellation and nothing to redo exactly -, weighted
// by the plan's ring multiplicities (hpk_plan.cpp: the reference's add-only incremental update, callers.py:175-198).  Two passes
// over the candidates: the first finds every candidate's resolving step per peak width and counts them (callers.py:203-217), the
// freeze decision is replayed on the counts (callers.py:219-229 / 505-511), the second forms sums, corrected expected, lambda
// chunk and Poisson p for the candidates resolved at executed steps (callers.py:238-271 / 517-540).  Benjamini-Hochberg, the
// result arrays and everything after them are the host half the device path uses (hpk_api.cpp).
#include "hpk_cpu.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

namespace {

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <class F>
void parallel_for(int64_t n, int threads, F&& f) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
    std::atomic<int64_t> next{0};
    auto work = [&](int t) {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n) break;
            f(i, t);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread& t : pool) t.join();
}

// ---- Poisson (hpk_device.h restated for the host: C. Loader's saddle-point pmf, the cdf summed from the mode's side)
double stirlerr(double x, const double* sfe) {
    if (x < 32.0) return sfe[(int)x];
    const double x2 = x * x;
    return (0.083333333333333333333 - (0.00277777777777777777778 - (0.00079365079365079365079365 -
            (0.000595238095238095238095238 - 0.0008417508417508417508417508 / x2) / x2) / x2) / x2) / x;
}
double bd0(double x, double np) {
    if (std::fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        if (std::fabs(s) < 2.2250738585072014e-308) return s;
        double ej = 2.0 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; ++j) {
            ej *= v;
            const double s1 = s + ej / (double)((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * std::log(x / np) + np - x;
}
double dpois(double x, double lam, const double* sfe) {
    if (x == 0.0) return std::exp(-lam);
    return std::exp(-stirlerr(x, sfe) - bd0(x, lam)) / std::sqrt(6.283185307179586476925286766559 * x);
}

struct Rings {          // what one candidate's window holds, ring by ring (filled as far as a step asks)
    double yraw[HPK_MAX_W + 1], ybal[HPK_MAX_W + 1], kbal[HPK_MAX_W + 1];
    int nraw = 0, nbal = 0;     // rings 1 .. n are filled
};

// A row block in true matrix coordinates, zero-padded: local row y <-> matrix row r0 - W + y, local column x <-> matrix column
// r0 - W + x.  raw: stored diagonals 0 .. num - 1 (callers.py:57-64); bal: diagonals min(ww) .. num - 1 (callers.py:84-96).
struct Block {
    int r0, r1, rows, cols;
    std::vector<float> raw;
    std::vector<double> bal;
};

}  // namespace

double hpk_cpu_poisson_sf(double k, double lam, const double* sfe, double sigcap) {
    if (!(lam > 0.0)) return 0.0;
    if (k < 0.0) return 1.0;
    k = std::floor(k);
    if (sigcap <= 0.25 && k < lam && lam >= 1.0) return 1.0;
    double cdf;
    if (k < lam) {
        double t = dpois(k, lam, sfe), sum = t, j = k;
        while (j > 0.0 && t > 0.0) {
            t *= j / lam; j -= 1.0; sum += t;
            if (t < sum * 1e-18) break;
        }
        cdf = sum < 1.0 ? sum : 1.0;
    } else {
        double j = k + 1.0, t = dpois(j, lam, sfe), sum = t;
        for (int it = 0; it < 100000 && t > 0.0; ++it) {
            j += 1.0; t *= lam / j; sum += t;
            if (t < sum * 1e-18) break;
        }
        cdf = 1.0 - sum;
    }
    return 1.0 - cdf;
}

void hpk_cpu_build_tables(HpkCpuTables& t, int threads) {
    const int32_t total = t.off[HPK_NB_TAB + 1];
    t.ptab.assign((size_t)total, 0.0);
    parallel_for(HPK_NB_TAB, threads, [&](int64_t i, int) {
        const int ch = (int)i + 1;
        const double lam = t.bounds[ch - 1];
        for (int32_t k = 0; k < t.off[ch + 1] - t.off[ch]; ++k)
            t.ptab[(size_t)t.off[ch] + k] = hpk_cpu_poisson_sf((double)k, lam, t.sfe.data(), 1.0);
    });
    t.built = true;
}

int hpk_cpu_band(const hpk_band& in, const hpk_params& prm, const HpkDevPlan& plan, const HpkCpuTables& tabs, int threads,
                 HpkCpuOut& out, std::string& err) {
    const int n = in.n, num = in.num;
    const int64_t ld = in.ld;
    const int W = plan.W, mw = plan.mw, D = plan.D, nsteps = plan.nsteps, nslots = plan.nslots;
    const bool bh = plan.mode == HPK_MODE_BHFDR;
    const int nsets = bh ? 1 : 2 * plan.npairs;
    const double sig = prm.sig;
    if (!in.raw || (!in.weight && !in.balanced)) { err = "raw and one of weight / balanced are needed"; return HPK_ERR_INVALID; }
    if (!in.IR && !in.weight) { err = "IR can only be derived from weights"; return HPK_ERR_INVALID; }
    const float* raw = in.raw;
    const double* wgt = in.weight;
    const double* balf = in.balanced;
    auto balanced_at = [&](int r, int k) -> double {        // diagonals min(ww) .. num - 1, inside the matrix (callers by construction)
        double b;
        if (balf) b = balf[(int64_t)r * ld + k];
        else b = ((double)raw[(int64_t)r * ld + k] * wgt[r]) * wgt[r + k];     // (raw * w_r) * w_c, as numpy forms it (pyHICCUPS:150-152)
        return b == b ? b : 0.0;                            // NaN -> 0 (pyHICCUPS:157)
    };
    const double t_begin = now_ms();

    // ---- scripts/pyHICCUPS:149-166 where the caller did not do it: IR[d] = mean of the balanced diagonal without the stored pixels
    // of masked bins, biases = 1 / weight (0 where the weight is 0 or NaN)
    const double* IR = in.IR;
    if (!IR) {
        out.IR.assign((size_t)num, 0.0);
        parallel_for(std::max(0, std::min(num, n) - mw), threads, [&](int64_t i, int) {
            const int d = mw + (int)i;
            double sum = 0.0;
            int64_t good = 0;
            for (int r = 0; r + d < n; ++r) {
                const double cnt = (double)raw[(int64_t)r * ld + d];
                if (cnt == 0.0) { ++good; continue; }       // an unstored pixel counts as 0, masked bin or not
                const double v = (cnt * wgt[r]) * wgt[r + d];
                if (v == v) { sum += v; ++good; }
            }
            out.IR[d] = good ? sum / (double)good : NAN;
        });
        IR = out.IR.data();
    }
    const double* b1 = in.bias1;
    const double* b2 = in.bias2;
    if (!b1 || !b2) {
        if (!wgt) { err = "biases can only be derived from weights"; return HPK_ERR_INVALID; }
        out.b1.assign((size_t)n, 0.0);
        for (int r = 0; r < n; ++r) { const double w = wgt[r]; out.b1[r] = (w == w && w != 0.0) ? 1.0 / w : 0.0; }
        if (!b1) b1 = out.b1.data();
        if (!b2) b2 = out.b1.data();
    }

    // ---- local-expected sums of the interior, per step and diagonal (callers.py:66-72 + 175-198 on EM)
    std::vector<double> etab((size_t)nsteps * 2 * (D + 1));
    hpk_build_etab(&plan, IR, num, etab.data());
    auto expected = [&](int s, int r, int c, double& EK, double& EY) {
        const int d = c - r;
        if (r >= W && c < n - W) {
            EK = etab[(size_t)(s * 2) * (D + 1) + d];
            EY = etab[(size_t)(s * 2 + 1) * (D + 1) + d];
            return;
        }
        // a window clipped by a matrix end: its cells one by one (the zero padding of callers.py:50-96)
        const int32_t* m = plan.steps[s].m;
        const int wi = plan.steps[s].wi;
        double ek = 0.0, ey = 0.0;
        for (int di = -wi; di <= wi; ++di)
            for (int dj = -wi; dj <= wi; ++dj) {
                if (di == 0 || dj == 0) continue;
                const int rho = std::max(std::abs(di), std::abs(dj));
                const int mm = m[rho];
                if (mm == 0) continue;
                const int rr = r + di, cc = c + dj, kk = cc - rr;
                if (rr < 0 || cc >= n || kk < mw || kk >= num) continue;
                const double v = (double)mm * IR[kk];
                ek += v;
                if (di > 0 && dj < 0) ey += v;
            }
        EK = ek; EY = ey;
    };

    // ---- row blocks
    const int TRB = 32;
    const int nblocks = (n + TRB - 1) / TRB;
    const int cols = TRB + D + 2 * W + 1;
    auto build_block = [&](int rb, Block& B) {
        B.r0 = rb * TRB; B.r1 = std::min(n, B.r0 + TRB);
        B.rows = TRB + 2 * W; B.cols = cols;
        B.raw.assign((size_t)B.rows * cols, 0.f);
        B.bal.assign((size_t)B.rows * cols, 0.0);
        const int clo = B.r0 - W;
        for (int y = 0; y < B.rows; ++y) {
            const int rr = B.r0 - W + y;
            if (rr < 0 || rr >= n) continue;
            for (int k = 0; k < num && rr + k < n; ++k) {
                const int x = rr + k - clo;
                if (x < 0) continue;
                if (x >= cols) break;
                const float cnt = raw[(int64_t)rr * ld + k];
                B.raw[(size_t)y * cols + x] = cnt;
                if (k >= mw && (balf || cnt != 0.f)) B.bal[(size_t)y * cols + x] = balanced_at(rr, k);
            }
        }
    };
    // ring rho of the window of local pixel (y, x): lower-left part of the raw counts / of the balanced values, all four quadrants
    // of the balanced values (the centre row and column belong to no ring: callers.py:179)
    auto ring_raw = [&](const Block& B, int y, int x, int rho) -> double {
        const float* p = B.raw.data();
        double s = 0.0;
        for (int di = 1; di <= rho; ++di) s += (double)p[(size_t)(y + di) * cols + (x - rho)];
        for (int dj = -rho + 1; dj <= -1; ++dj) s += (double)p[(size_t)(y + rho) * cols + (x + dj)];
        return s;
    };
    auto ring_bal = [&](const Block& B, int y, int x, int rho, double& yy, double& kk) {
        const double* p = B.bal.data();
        double ll = 0.0, others = 0.0;
        for (int di = 1; di <= rho; ++di) {
            ll += p[(size_t)(y + di) * cols + (x - rho)];
            others += p[(size_t)(y + di) * cols + (x + rho)] + p[(size_t)(y - di) * cols + (x - rho)] + p[(size_t)(y - di) * cols + (x + rho)];
        }
        for (int dj = 1; dj <= rho - 1; ++dj) {
            ll += p[(size_t)(y + rho) * cols + (x - dj)];
            others += p[(size_t)(y + rho) * cols + (x + dj)] + p[(size_t)(y - rho) * cols + (x - dj)] + p[(size_t)(y - rho) * cols + (x + dj)];
        }
        yy = ll; kk = ll + others;
    };
    // the widest ring a step reads
    std::vector<int> rmax_reads(nsteps, 0), rmax_sums(nsteps, 0);
    for (int s = 0; s < nsteps; ++s)
        for (int rho = 1; rho <= W; ++rho) {
            if (plan.steps[s].mr[rho]) rmax_reads[s] = rho;
            if (plan.steps[s].m[rho]) rmax_sums[s] = rho;
        }

    // ---- pass 1: the step at which each candidate resolves, per peak width (callers.py:203-217, 487-499)
    std::vector<std::vector<uint8_t>> stepidx((size_t)nblocks);
    const int nthreads = std::max(1, threads);
    std::vector<std::vector<unsigned long long>> hist_t((size_t)nthreads, std::vector<unsigned long long>(HPK_MAX_STEPS + 1, 0ull));
    out.rowlive.assign((size_t)n, 0);
    parallel_for(nblocks, nthreads, [&](int64_t rb, int t) {
        Block B;
        build_block((int)rb, B);
        std::vector<uint8_t>& si = stepidx[(size_t)rb];
        unsigned long long* hist = hist_t[(size_t)t].data();
        for (int r = B.r0; r < B.r1; ++r) {
            const int y = r - B.r0 + W;
            // gap rows (callers.py:238): a row of the balanced upper band without a non-zero value
            bool live = false;
            for (int k = mw; k < num && r + k < n && !live; ++k) live = B.bal[(size_t)y * cols + (r + k - (B.r0 - W))] != 0.0;
            out.rowlive[r] = live ? 1 : 0;
            for (int k = mw; k <= D && k < num && r + k < n; ++k) {
                const int x = r + k - (B.r0 - W);
                if (B.raw[(size_t)y * cols + x] == 0.f) continue;
                ++hist[HPK_MAX_STEPS];
                Rings R;
                for (int q = 0; q < nslots; ++q) {
                    uint8_t found = 0xff;
                    for (int s = 0; s < nsteps; ++s) {
                        const HpkDevStep& st = plan.steps[s];
                        if (st.slot != q) continue;
                        while (R.nraw < rmax_reads[s]) { ++R.nraw; R.yraw[R.nraw] = ring_raw(B, y, x, R.nraw); }
                        double reads = 0.0;
                        for (int rho = 1; rho <= rmax_reads[s]; ++rho) reads += (double)st.mr[rho] * R.yraw[rho];
                        if (reads >= (double)plan.min_reads) { found = (uint8_t)s; break; }
                    }
                    si.push_back(found);
                    if (found != 0xff) ++hist[found];
                }
            }
        }
    });
    for (int i = 0; i <= HPK_MAX_STEPS; ++i) {
        out.hist[i] = 0ull;
        for (int t = 0; t < nthreads; ++t) out.hist[i] += hist_t[(size_t)t][i];
    }
    // ---- the freeze decision (callers.py:208-229; bhfdr: the break at 505-511)
    {
        const long long total = (long long)out.hist[HPK_MAX_STEPS];
        long long unres[HPK_KSLOTS];
        for (int q = 0; q < HPK_KSLOTS; ++q) unres[q] = total;
        int fw = plan.W, e = 0;
        for (int s = 0; s < nsteps; ++s) {
            const HpkDevStep& st = plan.steps[s];
            if (st.wi > fw) { out.exec[s] = 0; continue; }
            out.exec[s] = 1;
            const long long before = unres[st.slot];
            if (before == 0 && e == 0) e = s + 1;
            const long long now = (long long)out.hist[s];
            const double vr = before ? (double)now / (double)before : 0.0;
            unres[st.slot] = before - now;
            const double lr = total ? (double)unres[st.slot] / (double)total : 0.0;
            if ((bh || st.wi >= plan.maxw) && (vr < 0.3 || lr < 0.03)) fw = st.wi;
        }
        for (int s = nsteps; s < HPK_MAX_STEPS; ++s) out.exec[s] = 0;
        out.frozen = fw;
        out.err = e;
    }
    out.band_px = 0;
    for (int d = mw; d <= std::min(D, num - 1); ++d) out.band_px += std::max(n - d, 0);
    out.ms_sums = now_ms() - t_begin;
    const double t_score = now_ms();
    out.emax.assign((size_t)nsets, 0ull);
    out.fam_m.assign((size_t)nsets * (HPK_NB + 1), 0u);
    out.fam_f.assign((size_t)nsets * (HPK_NB + 1), 0u);
    out.surv.clear();
    if (out.err != 0 || (prm.flags & HPK_FLAG_NO_SCORE)) { out.ms_score = 0.0; return HPK_OK; }

    // ---- pass 2: sums at the resolving step, corrected expected, lambda chunk, Poisson p (callers.py:238-271, 517-540)
    struct Local {
        std::vector<unsigned long long> emax;
        std::vector<uint32_t> fam_m, fam_f;
        std::vector<HpkSurv> surv;
    };
    std::vector<Local> loc((size_t)nthreads);
    for (Local& l : loc) { l.emax.assign((size_t)nsets, 0ull); l.fam_m.assign((size_t)nsets * (HPK_NB + 1), 0u); l.fam_f.assign((size_t)nsets * (HPK_NB + 1), 0u); }
    const double* bounds = tabs.bounds.data();
    const double* sfe = tabs.sfe.data();
    const int npairs = bh ? 1 : plan.npairs;
    parallel_for(nblocks, nthreads, [&](int64_t rb, int t) {
        Block B;
        build_block((int)rb, B);
        Local& L = loc[(size_t)t];
        const uint8_t* si = stepidx[(size_t)rb].data();
        for (int r = B.r0; r < B.r1; ++r) {
            const int y = r - B.r0 + W;
            for (int k = mw; k <= D && k < num && r + k < n; ++k) {
                const int x = r + k - (B.r0 - W);
                const float rawpix = B.raw[(size_t)y * cols + x];
                if (rawpix == 0.f) continue;
                const uint8_t* mine = si;
                si += nslots;
                const int c = r + k, d = k;
                Rings R;
                double SKq[HPK_KSLOTS], SYq[HPK_KSLOTS], EKq[HPK_KSLOTS], EYq[HPK_KSLOTS];
                bool done[HPK_KSLOTS] = {false, false, false, false};
                for (int pj = 0; pj < npairs; ++pj) {
                    const int q = bh ? 0 : plan.pair_slot[pj];
                    const int wi0 = bh ? plan.steps[0].wi : plan.pair_wi[pj];
                    const int s = mine[q];
                    if (s == 0xff || plan.steps[s].wi > out.frozen || d < wi0) continue;
                    if (!done[q]) {
                        const HpkDevStep& st = plan.steps[s];
                        while (R.nbal < rmax_sums[s]) { ++R.nbal; ring_bal(B, y, x, R.nbal, R.ybal[R.nbal], R.kbal[R.nbal]); }
                        double sk = 0.0, sy = 0.0;
                        for (int rho = 1; rho <= rmax_sums[s]; ++rho)
                            if (st.m[rho]) { sk += (double)st.m[rho] * R.kbal[rho]; sy += (double)st.m[rho] * R.ybal[rho]; }
                        SKq[q] = sk; SYq[q] = sy;
                        expected(s, r, c, EKq[q], EYq[q]);
                        done[q] = true;
                    }
                    // callers.py:244-249: E = ((IR[d] * (bS / bE)) * B1[x]) * B2[y] where bE != 0
                    const double eK = (EKq[q] != 0.0) ? ((IR[d] * (SKq[q] / EKq[q])) * b1[r]) * b2[c] : 0.0;
                    const double eY = (EYq[q] != 0.0) ? ((IR[d] * (SYq[q] / EYq[q])) * b1[r]) * b2[c] : 0.0;
                    const double O = (double)rawpix;
                    for (int fl = 0; fl < (bh ? 1 : 2); ++fl) {
                        const int set = bh ? 0 : pj * 2 + fl;
                        const double E = fl ? eY : eK;
                        if (!(E > 0.0)) continue;                               // callers.py:250
                        unsigned long long ebits;
                        std::memcpy(&ebits, &E, 8);
                        if (ebits > L.emax[(size_t)set]) L.emax[(size_t)set] = ebits;
                        int chunk;
                        double p;
                        if (bh) {
                            chunk = 1;                                          // one family (callers.py:545)
                            p = hpk_cpu_poisson_sf(O, E, sfe, sig);             // callers.py:536-540
                        } else {
                            // lambda chunks: strict on both sides (callers.py:38) - an E on a boundary belongs to none
                            const int pos = (int)(std::lower_bound(bounds, bounds + HPK_NB, E) - bounds);
                            if (pos < HPK_NB && bounds[pos] == E) chunk = 0;
                            else chunk = pos < HPK_NB ? pos + 1 : 0;
                            p = 1.0;
                            if (chunk) {
                                if (chunk <= HPK_NB_TAB) {
                                    const int kO = (int)O, len = tabs.off[chunk + 1] - tabs.off[chunk];
                                    p = kO < len ? tabs.ptab[(size_t)tabs.off[chunk] + kO] : 0.0;
                                } else p = hpk_cpu_poisson_sf(O, bounds[chunk - 1], sfe, sig);      // callers.py:268-270
                            }
                        }
                        ++L.fam_m[(size_t)set * (HPK_NB + 1) + chunk];
                        if (chunk && p <= sig) {
                            ++L.fam_f[(size_t)set * (HPK_NB + 1) + chunk];
                            HpkSurv rec;
                            rec.x = r; rec.y = c; rec.O = rawpix; rec.set = (uint8_t)set; rec.chunk = (uint8_t)chunk;
                            rec.flag = (uint8_t)((fl == 0 && eY == 0.0) ? 1 : 0);           // callers.py:330
                            rec.pad = 0; rec.E = E; rec.p = p; rec.bal = B.bal[(size_t)y * cols + x];
                            L.surv.push_back(rec);
                        }
                    }
                }
            }
        }
    });
    for (const Local& l : loc) {
        for (int s2 = 0; s2 < nsets; ++s2) out.emax[(size_t)s2] = std::max(out.emax[(size_t)s2], l.emax[(size_t)s2]);
        for (size_t i = 0; i < out.fam_m.size(); ++i) { out.fam_m[i] += l.fam_m[i]; out.fam_f[i] += l.fam_f[i]; }
        out.surv.insert(out.surv.end(), l.surv.begin(), l.surv.end());
    }
    out.ms_score = now_ms() - t_score;
    return HPK_OK;
}
