// HIP kernels for gfx950 (MI355X / CDNA4).  Wave = 64 lanes; one stencil workgroup owns the whole 160 KiB
// LDS of a CU.
//
//   hpk_stencil   donut (K) + lower-left (Y) local sums and adaptive widening   callers.py:132-232, 440-513
//   hpk_freeze    frozen_w / break decision from the resolve histogram            callers.py:208-229, 505-511
//   hpk_score     corrected expected -> lambda chunk -> Poisson p -> survivors    callers.py:238-271, 517-540
//   hpk_gap       zero rows of the balanced band                                  callers.py:238, 557
//   hpk_ptab      Poisson survival table for the chunk bounds                     callers.py:268-270
//   hpk_brute     independent explicit-window check (tests)
//
// Stencil design.  The tile is built in true matrix coordinates (r, c): an output tile of TR x TC pixels
// plus a halo of maxww (+1 row/column for the prefix origin) is read from band storage - rows are
// contiguous in c, so every wave reads 128 consecutive floats of one band row - and turned into a
// summed-area table (SAT) of 16-byte cells {f64 balanced, u32 raw, u32 valid-raw} in LDS.  Each wave owns
// RPW consecutive rows x 128 columns (two cells per lane): the prefix along a row is an in-register DPP
// scan, the prefix down the columns is a running sum in registers, so the SAT is written to LDS exactly
// once and never read back during construction.  With the SAT every quadrant box of the (p, w) window is
// four cell reads (one ds_read_b128 each), independent of w.  u32 sums wrap but their differences are
// exact; the valid-raw sum tells an all-zero balanced box (exact 0, as the reference's CSR adds give)
// from floating-point residue of the f64 SAT.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "hpk_kernels.h"

namespace {

constexpr int LC = HPK_LC;
constexpr int LR = HPK_LR;

struct __attribute__((aligned(16))) Cell {
    double c;        // balanced
    unsigned r;      // raw count
    unsigned v;      // raw count where balanced != 0
};

// ------------------------------------------------------------------ wave64 DPP scan (gfx9 DPP controls)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWMASK, 0xf, true);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void scan_step(double& c, unsigned& r, unsigned& v) {
    c += dpp_f64<CTRL, ROWMASK>(c);
    r += dpp_u32<CTRL, ROWMASK>(r);
    v += dpp_u32<CTRL, ROWMASK>(v);
}
// inclusive prefix over the 64 lanes, then shifted right by one lane (= exclusive prefix)
__device__ __forceinline__ void wave_exclusive_scan(double& c, unsigned& r, unsigned& v) {
    scan_step<DPP_ROW_SHR1, 0xf>(c, r, v);
    scan_step<DPP_ROW_SHR2, 0xf>(c, r, v);
    scan_step<DPP_ROW_SHR4, 0xf>(c, r, v);
    scan_step<DPP_ROW_SHR8, 0xf>(c, r, v);
    scan_step<DPP_ROW_BCAST15, 0xa>(c, r, v);
    scan_step<DPP_ROW_BCAST31, 0xc>(c, r, v);
    c = dpp_f64<DPP_WAVE_SHR1, 0xf>(c);
    r = dpp_u32<DPP_WAVE_SHR1, 0xf>(r);
    v = dpp_u32<DPP_WAVE_SHR1, 0xf>(v);
}

// ------------------------------------------------------------------ box sums on the SAT
// Four off-cross quadrants (donut support) and the lower-left quadrant at Chebyshev radius rho around
// SAT cell (Y, X).  pixc / pixv: the pixel's own balanced value / valid count; sYXm1 = S(Y, X-1).
__device__ __forceinline__ void box_ky(const Cell* __restrict__ S, int Y, int X, int rho, double pixc, unsigned pixv,
                                       const Cell& sYXm1, double& kc, unsigned& kv, double& yc, unsigned& yv) {
    const int t = Y - rho - 1, b = Y + rho, xl = X - rho - 1, xr = X + rho;
    const Cell tl = S[t * LC + xl], tm1 = S[t * LC + X - 1], tm = S[t * LC + X], tr = S[t * LC + xr];
    const Cell bl = S[b * LC + xl], bm1 = S[b * LC + X - 1], bm = S[b * LC + X], br = S[b * LC + xr];
    const Cell ml0 = S[(Y - 1) * LC + xl], ml1 = S[Y * LC + xl], mr0 = S[(Y - 1) * LC + xr], mr1 = S[Y * LC + xr];
    const double top = (tl.c - tm1.c) + (tm.c - tr.c);
    const double bot = (br.c - bm.c) + (bm1.c - bl.c);
    const double mid = (ml1.c - ml0.c) + (mr0.c - mr1.c);
    kc = ((top + bot) + mid) + pixc;
    kv = tl.v - tm1.v + tm.v - tr.v + br.v - bm.v + bm1.v - bl.v + ml1.v - ml0.v + mr0.v - mr1.v + pixv;
    yc = (bm1.c - bl.c) - (sYXm1.c - ml1.c);
    yv = bm1.v - bl.v - sYXm1.v + ml1.v;
}

__device__ __forceinline__ unsigned reads_box(const Cell* __restrict__ S, int Y, int X, int rho, unsigned sYXm1r) {
    const int b = Y + rho, xl = X - rho - 1;
    return S[b * LC + X - 1].r - sYXm1r - S[b * LC + xl].r + S[Y * LC + xl].r;
}

// balanced value of pixel (rr, cc) on diagonal k (formed on chip in weight mode): (raw * w_r) * w_c, NaN -> 0
__device__ __forceinline__ double balanced_of(float raw, double wr, double wc) {
    double b = ((double)raw * wr) * wc;
    return (b == b) ? b : 0.0;
}

// explicit local-expected sums for pixels whose window is clipped by the matrix ends (callers.py:50-96 padding)
__device__ __noinline__ void edge_expected(const HpkDevStep& st, const double* __restrict__ IR, int r, int c, int n,
                                           int num, int mw, double& EK, double& EY) {
    double ek = 0.0, ey = 0.0;
    const int wi = st.wi;
    for (int di = -wi; di <= wi; ++di) {
        for (int dj = -wi; dj <= wi; ++dj) {
            if (di == 0 || dj == 0) continue;
            const int adi = di < 0 ? -di : di, adj = dj < 0 ? -dj : dj;
            const int rho = adi > adj ? adi : adj;
            const int m = st.m[rho];
            if (m == 0) continue;
            const int rr = r + di, cc = c + dj, kk = cc - rr;
            if (rr < 0 || cc >= n || kk < mw || kk >= num) continue;
            const double v = (double)m * IR[kk];
            ek += v;
            if (di > 0 && dj < 0) ey += v;
        }
    }
    EK = ek;
    EY = ey;
}

// ------------------------------------------------------------------ stencil
template <int NW, bool BALF64, bool SUMS>
__global__ void __launch_bounds__(NW * 64) hpk_stencil(HpkStencilArgs a) {
    constexpr int RPW = LR / NW;
    static_assert(RPW * NW == LR, "rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Cell* S = reinterpret_cast<Cell*>(smem);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    // XCD-aware tile order: block b runs on XCD b % 8 (observed), give each XCD a contiguous run of tiles so
    // that the halo rows/columns shared by neighbouring tiles are served by one L2.
    const int tid = (int)(blockIdx.x & 7) * a.chunk + (int)(blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= (unsigned)a.chunk || tid >= a.ntiles) return;
    const int rb = tid / a.J, cj = tid - rb * a.J;
    const int r0 = rb * a.TR;
    const int c0 = r0 + a.mw + cj * a.TC;
    const int W = a.W, n = a.n, num = a.num, mw = a.mw, D = a.D;
    if (c0 >= n || (mw + cj * a.TC - (a.TR - 1)) > D) return;   // no band pixel inside the matrix

    // ---- phase 1: read this wave's RPW rows x 128 columns, form balanced values, column totals
    const int xx0 = 2 * lane;
    const int cc0 = c0 - W - 1 + xx0;
    const int rr0 = r0 - W - 1 + wave * RPW;
    float rawv[RPW][2];
    double balv[RPW][2];
    double wc[2] = {0.0, 0.0};
    if (!BALF64) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cc = cc0 + e;
            wc[e] = (cc >= 0 && cc < n) ? a.weight[cc] : 0.0;
        }
    }
    double tc[2] = {0.0, 0.0};
    unsigned tr[2] = {0u, 0u}, tv[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int rr = rr0 + j;
        const bool rowok = rr >= 0 && rr < n;
        double wr = 0.0;
        if (!BALF64) wr = rowok ? a.weight[rr] : 0.0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cc = cc0 + e;
            const int k = cc - rr;
            const bool inb = rowok && cc < n && k >= 0 && k < num;
            float rv = 0.f;
            double bv = 0.0;
            if (inb) {
                const int64_t off = (int64_t)rr * a.ld + k;
                rv = a.raw[off];
                if (k >= mw) {
                    if (BALF64) { bv = a.bal[off]; bv = (bv == bv) ? bv : 0.0; }
                    else bv = balanced_of(rv, wr, wc[e]);
                }
            }
            rawv[j][e] = rv;
            balv[j][e] = bv;
            const unsigned ru = (unsigned)rv;
            tc[e] += bv;
            tr[e] += ru;
            tv[e] += (bv != 0.0) ? ru : 0u;
        }
    }
    // column totals of this wave's row segment -> LDS (aliases the SAT; consumed before the SAT is written)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        Cell t; t.c = tc[e]; t.r = tr[e]; t.v = tv[e];
        S[wave * LC + xx0 + e] = t;
    }
    __syncthreads();
    double ac[2] = {0.0, 0.0};           // running column sums of row-prefixed values = SAT of the rows above
    unsigned ar[2] = {0u, 0u}, av[2] = {0u, 0u};
    for (int w2 = 0; w2 < wave; ++w2) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const Cell t = S[w2 * LC + xx0 + e];
            ac[e] += t.c; ar[e] += t.r; av[e] += t.v;
        }
    }
    __syncthreads();
    {   // prefix of the segment base along the row
        double pc = ac[0] + ac[1]; unsigned pr = ar[0] + ar[1], pv = av[0] + av[1];
        const double l1c = pc; const unsigned l1r = pr, l1v = pv;
        wave_exclusive_scan(pc, pr, pv);
        ac[0] = pc + ac[0]; ar[0] = pr + ar[0]; av[0] = pv + av[0];
        ac[1] = pc + l1c;   ar[1] = pr + l1r;   av[1] = pv + l1v;
    }
    // ---- phase 2: row prefix by DPP scan, column prefix by running sums, one LDS write per cell
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const unsigned r0u = (unsigned)rawv[j][0], r1u = (unsigned)rawv[j][1];
        const double c0v = balv[j][0], c1v = balv[j][1];
        const unsigned v0 = (c0v != 0.0) ? r0u : 0u, v1 = (c1v != 0.0) ? r1u : 0u;
        const double l1c = c0v + c1v; const unsigned l1r = r0u + r1u, l1v = v0 + v1;
        double pc = l1c; unsigned pr = l1r, pv = l1v;
        wave_exclusive_scan(pc, pr, pv);
        ac[0] += pc + c0v; ar[0] += pr + r0u; av[0] += pv + v0;
        ac[1] += pc + l1c; ar[1] += pr + l1r; av[1] += pv + l1v;
        Cell o0; o0.c = ac[0]; o0.r = ar[0]; o0.v = av[0];
        Cell o1; o1.c = ac[1]; o1.r = ar[1]; o1.v = av[1];
        S[(wave * RPW + j) * LC + xx0] = o0;
        S[(wave * RPW + j) * LC + xx0 + 1] = o1;
    }
    __syncthreads();

    // ---- phase 3: every band pixel of the tile
    const HpkDevPlan* __restrict__ plan = a.plan;
    const int nsteps = plan->nsteps, nslots = plan->nslots, min_reads = plan->min_reads;
    const unsigned alldone = (1u << nslots) - 1u;
    unsigned myhist = 0u, mycand = 0u;
    const int64_t slot_stride = (int64_t)n * a.ldo;

    for (int y = wave; y < a.TR; y += NW) {
        const int r = r0 + y;
        if (r >= n) break;
        const int Y = y + W + 1;
        for (int xb = 0; xb < a.TC; xb += 64) {
            const int x = xb + lane;
            const int c = c0 + x;
            const int d = c - r;
            const bool inband = x < a.TC && c < n && d >= mw && d <= D && d < num;
            const int X = x + W + 1;
            float rawpix = 0.f;
            if (inband) rawpix = a.raw[(int64_t)r * a.ld + d];
            const bool cand = inband && rawpix != 0.f;
            double eKs[HPK_KSLOTS], eYs[HPK_KSLOTS];
            double4 sums[SUMS ? HPK_KSLOTS : 1];
            unsigned wres = 0u;          // 8 bits per slot
#pragma unroll
            for (int q = 0; q < HPK_KSLOTS; ++q) { eKs[q] = 0.0; eYs[q] = 0.0; }
            if (SUMS) {
#pragma unroll
                for (int q = 0; q < HPK_KSLOTS; ++q) sums[q] = make_double4(0.0, 0.0, 0.0, 0.0);
            }
            const unsigned long long candmask = __ballot(cand);
            if (lane == 0) mycand += (unsigned)__popcll(candmask);
            if (candmask != 0ull) {
                double pixc = 0.0, ir = 0.0, b1r = 0.0, b2c = 0.0;
                unsigned pixv = 0u;
                Cell sYXm1; sYXm1.c = 0.0; sYXm1.r = 0u; sYXm1.v = 0u;
                if (cand) {
                    if (BALF64) { pixc = a.bal[(int64_t)r * a.ld + d]; pixc = (pixc == pixc) ? pixc : 0.0; }
                    else pixc = balanced_of(rawpix, a.weight[r], a.weight[c]);
                    pixv = (pixc != 0.0) ? (unsigned)rawpix : 0u;
                    ir = a.IR[d];
                    b1r = a.b1[r];
                    b2c = a.b2[c];
                    sYXm1 = S[Y * LC + X - 1];
                }
                const bool edge = (r < W) || (c >= n - W);
                unsigned done = cand ? 0u : alldone;
                int cur_rid = -1;
                unsigned reads = 0u;
                for (int s = 0; s < nsteps; ++s) {
                    const HpkDevStep& st = plan->steps[s];
                    const bool need = ((done >> st.slot) & 1u) == 0u;
                    if (__ballot(need) == 0ull) continue;
                    if (st.reads_id != cur_rid) {
                        cur_rid = st.reads_id;
                        if (done != alldone) {
                            unsigned acc = 0u;
                            for (int j = 0; j < st.nrt; ++j)
                                acc += (unsigned)st.rt_coef[j] * reads_box(S, Y, X, st.rt_rho[j], sYXm1.r);
                            reads = acc;
                        }
                    }
                    const bool hit = need && (reads >= (unsigned)min_reads);
                    const unsigned long long hitmask = __ballot(hit);
                    if (lane == s) myhist += (unsigned)__popcll(hitmask);
                    if (hit) {
                        double SK = 0.0, SY = 0.0;
                        unsigned VK = 0u, VY = 0u;
                        for (int j = 0; j < st.nkt; ++j) {
                            double kc, yc; unsigned kv, yv;
                            box_ky(S, Y, X, st.kt_rho[j], pixc, pixv, sYXm1, kc, kv, yc, yv);
                            const double cf = (double)st.kt_coef[j];
                            SK += cf * kc; SY += cf * yc;
                            VK += (unsigned)st.kt_coef[j] * kv; VY += (unsigned)st.kt_coef[j] * yv;
                        }
                        if (VK == 0u) SK = 0.0;          // every contributing balanced value is 0: exact 0
                        if (VY == 0u) SY = 0.0;
                        double EK, EY;
                        if (!edge) {
                            EK = a.etab[(int64_t)(s * 2) * (D + 1) + d];
                            EY = a.etab[(int64_t)(s * 2 + 1) * (D + 1) + d];
                        } else {
                            edge_expected(st, a.IR, r, c, n, num, mw, EK, EY);
                        }
                        // callers.py:244-249: E = ((IR[d] * (bS / bE)) * B1[x]) * B2[y] where bE != 0
                        const double eK = (EK != 0.0) ? ((ir * (SK / EK)) * b1r) * b2c : 0.0;
                        const double eY = (EY != 0.0) ? ((ir * (SY / EY)) * b1r) * b2c : 0.0;
#pragma unroll
                        for (int q = 0; q < HPK_KSLOTS; ++q) {
                            if (q == st.slot) {
                                eKs[q] = eK; eYs[q] = eY;
                                if (SUMS) sums[q] = make_double4(SK, EK, SY, EY);
                            }
                        }
                        wres |= (unsigned)st.wi << (8 * st.slot);
                        done |= 1u << st.slot;
                    }
                    if (__ballot(done != alldone) == 0ull) break;
                }
            }
            if (inband) {
                const int64_t o = (int64_t)r * a.ldo + d;
#pragma unroll
                for (int q = 0; q < HPK_KSLOTS; ++q) {
                    if (q < nslots) {
                        a.outE[q * slot_stride + o] = make_double2(eKs[q], eYs[q]);
                        a.outW[q * slot_stride + o] = (uint8_t)((wres >> (8 * q)) & 0xffu);
                        if (SUMS) a.outS[q * slot_stride + o] = sums[q];
                    }
                }
            }
        }
    }
    if (lane < nsteps && myhist) atomicAdd(&a.hist[lane], (unsigned long long)myhist);
    if (lane == 0 && mycand) atomicAdd(&a.hist[HPK_HIST_NCAND], (unsigned long long)mycand);
}

// ------------------------------------------------------------------ freeze (one thread)
__global__ void hpk_freeze(const HpkDevPlan* __restrict__ plan, const unsigned long long* __restrict__ hist,
                           int32_t* frozen, int32_t* executed, int32_t* err) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long total = (long long)hist[HPK_HIST_NCAND];
    long long unres[HPK_KSLOTS];
    for (int q = 0; q < HPK_KSLOTS; ++q) unres[q] = total;
    int fw = plan->W;
    int e = 0;
    for (int s = 0; s < plan->nsteps; ++s) {
        const HpkDevStep& st = plan->steps[s];
        if (st.wi > fw) { executed[s] = 0; continue; }                 // callers.py:133-134 / break at 505-511
        executed[s] = 1;
        const long long before = unres[st.slot];
        if (before == 0 && e == 0) e = s + 1;                           // the reference raises here
        const long long now = (long long)hist[s];
        const double vr = before ? (double)now / (double)before : 0.0;  // callers.py:208 / 492
        unres[st.slot] = before - now;
        const double lr = total ? (double)unres[st.slot] / (double)total : 0.0;   // callers.py:219 / 501
        const bool widest = (plan->mode == HPK_MODE_BHFDR) || (st.wi >= plan->maxw);
        if (widest && (vr < 0.3 || lr < 0.03)) fw = st.wi;              // callers.py:223-229
    }
    *frozen = fw;
    *err = e;
}

// ------------------------------------------------------------------ gap rows
__global__ void __launch_bounds__(256) hpk_gap(const float* __restrict__ raw, const double* __restrict__ bal,
                                               const double* __restrict__ weight, int n, int num, int64_t ld,
                                               int mw, uint8_t* __restrict__ gap) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    bool any = false;
    const double wr = weight ? weight[r] : 0.0;
    const int kend = (n - r < num) ? n - r : num;
    for (int k = mw + lane; k < kend; k += 64) {
        double b;
        if (bal) { b = bal[(int64_t)r * ld + k]; b = (b == b) ? b : 0.0; }
        else b = balanced_of(raw[(int64_t)r * ld + k], wr, weight[r + k]);
        any = any || (b != 0.0);
    }
    const unsigned long long m = __ballot(any);
    if (lane == 0) gap[r] = (m == 0ull) ? 1 : 0;
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
__device__ double poisson_sf(double k, double lam, const double* __restrict__ sfe) {
    if (!(lam > 0.0)) return 0.0;
    if (k < 0.0) return 1.0;
    k = floor(k);
    double cdf;
    if (k < lam) {                         // lower sum, terms shrink going down from k
        double t = dpois(k, lam, sfe), sum = t, j = k;
        while (j > 0.0) {
            t *= j / lam; j -= 1.0; sum += t;
            if (t < sum * 1e-18) break;
        }
        cdf = sum < 1.0 ? sum : 1.0;
    } else {                               // upper tail, terms shrink going up from k + 1
        double j = k + 1.0, t = dpois(j, lam, sfe), sum = t;
        for (int it = 0; it < 100000; ++it) {
            j += 1.0; t *= lam / j; sum += t;
            if (t < sum * 1e-18) break;
        }
        cdf = 1.0 - sum;
    }
    return 1.0 - cdf;
}

__global__ void __launch_bounds__(256) hpk_ptab(const double* __restrict__ bounds, const int32_t* __restrict__ off,
                                                const double* __restrict__ sfe, double* __restrict__ ptab, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int ch = 1;
    while (ch < HPK_NB_TAB && i >= off[ch + 1]) ++ch;      // off[ch] .. off[ch+1]-1 belong to chunk ch (1-based)
    const double k = (double)(i - off[ch]);
    ptab[i] = poisson_sf(k, bounds[ch - 1], sfe);
}

__global__ void __launch_bounds__(256) hpk_poisson_sf_k(const double* __restrict__ k, const double* __restrict__ lam,
                                                        const double* __restrict__ sfe, double* __restrict__ out,
                                                        int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = poisson_sf(k[i], lam[i], sfe);
}

// ------------------------------------------------------------------ scoring
// One thread per band pixel (row = blockIdx.x, diagonals mw + blockIdx.y * 256 + threadIdx.x).
__global__ void __launch_bounds__(256) hpk_score(HpkScoreArgs a) {
    __shared__ unsigned int lhist[2 * HPK_MAX_PAIRS][HPK_NB + 1];
    __shared__ unsigned long long lemax[2 * HPK_MAX_PAIRS];
    __shared__ unsigned int lvalid[2 * HPK_MAX_PAIRS];
    const HpkDevPlan* __restrict__ plan = a.plan;
    const int mode = plan->mode;
    const int npairs = plan->npairs;
    const int nsets = (mode == HPK_MODE_BHFDR) ? 1 : 2 * npairs;
    for (int i = threadIdx.x; i < nsets * (HPK_NB + 1); i += blockDim.x) (&lhist[0][0])[i] = 0u;
    if (threadIdx.x < 2 * HPK_MAX_PAIRS) { lemax[threadIdx.x] = 0ull; lvalid[threadIdx.x] = 0u; }
    __syncthreads();

    const int r = blockIdx.x;
    const int d = a.mw + blockIdx.y * 256 + threadIdx.x;
    const int c = r + d;
    const int lane = threadIdx.x & 63;
    const bool inband = d <= a.D && d < a.num && c < a.n;
    float rawpix = 0.f;
    if (inband) rawpix = a.raw[(int64_t)r * a.ld + d];
    const bool cand = inband && rawpix != 0.f;
    const int frozen = *a.frozen;
    const int64_t slot_stride = (int64_t)a.n * a.ldo;
    const int64_t o = (int64_t)r * a.ldo + d;
    const double O = (double)rawpix;

    for (int pj = 0; pj < npairs; ++pj) {
        const int slot = plan->pair_slot[pj];
        const int wi0 = plan->pair_wi[pj];
        bool ok = cand && d >= wi0;                                   // callers.py:244
        double2 e2 = make_double2(0.0, 0.0);
        if (ok) {
            const unsigned w = a.outW[slot * slot_stride + o];
            ok = (w != 0u) && ((int)w <= frozen);                     // resolved at an executed step
            if (ok) e2 = a.outE[slot * slot_stride + o];
        }
        const int nfl = (mode == HPK_MODE_BHFDR) ? 1 : 2;
        for (int fl = 0; fl < nfl; ++fl) {
            const int set = (mode == HPK_MODE_BHFDR) ? 0 : pj * 2 + fl;
            const double E = fl ? e2.y : e2.x;
            const bool valid = ok && (E > 0.0);                       // callers.py:250
            int chunk = 0;
            double p = 1.0;
            if (valid) {
                atomicAdd(&lvalid[set], 1u);
                atomicMax(&lemax[set], (unsigned long long)__double_as_longlong(E));
                if (mode == HPK_MODE_BHFDR) {
                    chunk = 1;
                    p = poisson_sf(O, E, a.sfe);                      // callers.py:536-540
                } else {
                    // smallest i with E < bounds[i-1]; membership is strict on both sides (callers.py:38)
                    int lo = 0, hi = HPK_NB;                          // search in bounds[0..HPK_NB)
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (E < a.bounds[mid]) hi = mid; else lo = mid + 1; }
                    if (lo < HPK_NB && !(lo > 0 && E == a.bounds[lo - 1])) {
                        chunk = lo + 1;
                        if (chunk <= HPK_NB_TAB) {
                            const int base = a.ptab_off[chunk], len = a.ptab_off[chunk + 1] - base;
                            const long long kO = (long long)O;
                            p = (kO < len) ? a.ptab[base + (int)kO] : 0.0;
                        } else {
                            p = poisson_sf(O, a.bounds[chunk - 1], a.sfe);   // callers.py:268-270
                        }
                    }
                }
                if (chunk) atomicAdd(&lhist[set][chunk], 1u);
            }
            const bool surv = valid && chunk != 0 && p <= a.sig;      // only these can reach q <= sig
            const unsigned long long sm = __ballot(surv);
            if (sm != 0ull) {
                unsigned long long base = 0ull;
                if (lane == 0) base = atomicAdd(a.nsurv, (unsigned long long)__popcll(sm));
                base = __shfl(base, 0);
                if (surv) {
                    const unsigned long long idx = base + (unsigned long long)__popcll(sm & ((1ull << lane) - 1ull));
                    if ((int64_t)idx < a.cap) {
                        double b;
                        if (a.bal) { b = a.bal[(int64_t)r * a.ld + d]; b = (b == b) ? b : 0.0; }
                        else b = balanced_of(rawpix, a.weight[r], a.weight[c]);
                        a.sx[idx] = r; a.sy[idx] = c; a.sset[idx] = (uint8_t)set; a.schunk[idx] = (uint8_t)chunk;
                        a.sflag[idx] = (fl == 0 && e2.y == 0.0) ? 1 : 0;      // callers.py:330
                        a.sO[idx] = rawpix; a.sE[idx] = E; a.sp[idx] = p; a.sbal[idx] = b;
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nsets * (HPK_NB + 1); i += blockDim.x) {
        const unsigned v = (&lhist[0][0])[i];
        if (v) atomicAdd(&a.chunk_hist[i], v);
    }
    if (threadIdx.x < nsets) {
        if (lvalid[threadIdx.x]) atomicAdd(&a.nvalid[threadIdx.x], (unsigned long long)lvalid[threadIdx.x]);
        if (lemax[threadIdx.x]) atomicMax(&a.emax_bits[threadIdx.x], lemax[threadIdx.x]);
    }
}

// ------------------------------------------------------------------ brute-force check (tests only)
__global__ void __launch_bounds__(64) hpk_brute(HpkBruteArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.count) return;
    const HpkDevStep& st = a.plan->steps[a.step];
    const int r = a.rows[i], c = a.cols[i];
    const int n = a.n, num = a.num, mw = a.plan->mw, W = a.plan->W;
    double SK = 0.0, EK = 0.0, SY = 0.0, EY = 0.0, RD = 0.0;
    for (int di = -W; di <= W; ++di) {
        for (int dj = -W; dj <= W; ++dj) {
            if (di == 0 || dj == 0) continue;
            const int adi = di < 0 ? -di : di, adj = dj < 0 ? -dj : dj;
            const int rho = adi > adj ? adi : adj;
            const int m = st.m[rho], mr = st.mr[rho];
            if (m == 0 && mr == 0) continue;
            const int rr = r + di, cc = c + dj, kk = cc - rr;
            if (rr < 0 || rr >= n || cc < 0 || cc >= n || kk < 0 || kk >= num) continue;
            const float rv = a.raw[(int64_t)rr * a.ld + kk];
            const bool ll = di > 0 && dj < 0;
            if (ll) RD += (double)mr * (double)rv;
            if (kk < mw) continue;
            double b;
            if (a.bal) { b = a.bal[(int64_t)rr * a.ld + kk]; b = (b == b) ? b : 0.0; }
            else b = balanced_of(rv, a.weight[rr], a.weight[cc]);
            const double x = a.IR[kk];
            SK += (double)m * b; EK += (double)m * x;
            if (ll) { SY += (double)m * b; EY += (double)m * x; }
        }
    }
    double* o = a.out + i * 5;
    o[0] = SK; o[1] = EK; o[2] = SY; o[3] = EY; o[4] = RD;
}

}  // namespace

// ------------------------------------------------------------------ launchers
int hpk_stencil_lds_bytes() { return LR * LC * (int)sizeof(Cell); }

template <int NW, bool BALF64, bool SUMS>
static void launch_stencil_t(const HpkStencilArgs& a, hipStream_t st) {
    auto kern = hpk_stencil<NW, BALF64, SUMS>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  hpk_stencil_lds_bytes());
        attr_done = true;
    }
    const int grid = a.chunk * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), hpk_stencil_lds_bytes(), st, a);
}

void hpk_launch_stencil(const HpkStencilArgs& a, bool balf64, bool sums, hipStream_t st) {
    constexpr int NW = 16;
    if (balf64) { if (sums) launch_stencil_t<NW, true, true>(a, st); else launch_stencil_t<NW, true, false>(a, st); }
    else        { if (sums) launch_stencil_t<NW, false, true>(a, st); else launch_stencil_t<NW, false, false>(a, st); }
}

void hpk_launch_freeze(const HpkDevPlan* plan, const unsigned long long* hist, int32_t* frozen, int32_t* executed,
                       int32_t* err, hipStream_t st) {
    hipLaunchKernelGGL(hpk_freeze, dim3(1), dim3(64), 0, st, plan, hist, frozen, executed, err);
}

void hpk_launch_gap(const float* raw, const double* bal, const double* weight, int32_t n, int32_t num, int64_t ld,
                    int32_t mw, uint8_t* gap, hipStream_t st) {
    hipLaunchKernelGGL(hpk_gap, dim3((n + 3) / 4), dim3(256), 0, st, raw, bal, weight, n, num, ld, mw, gap);
}

void hpk_launch_score(const HpkScoreArgs& a, hipStream_t st) {
    const int wd = a.D - a.mw + 1;
    if (wd <= 0 || a.n <= 0) return;
    hipLaunchKernelGGL(hpk_score, dim3(a.n, (wd + 255) / 256), dim3(256), 0, st, a);
}

void hpk_launch_ptab(const double* bounds, const int32_t* off, const double* sfe, double* ptab, int32_t total,
                     hipStream_t st) {
    hipLaunchKernelGGL(hpk_ptab, dim3((total + 255) / 256), dim3(256), 0, st, bounds, off, sfe, ptab, total);
}

void hpk_launch_poisson_sf(const double* k, const double* lam, const double* sfe, double* out, int64_t count,
                           hipStream_t st) {
    hipLaunchKernelGGL(hpk_poisson_sf_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, k, lam, sfe, out, count);
}

void hpk_launch_brute(const HpkBruteArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(hpk_brute, dim3((unsigned)((a.count + 63) / 64)), dim3(64), 0, st, a);
}
