// Kernels that only the tests and the dense debug outputs use - kept out of the product's translation unit:
//   hpk_dense          records -> dense (E_K, E_Y), resolving width, the four sums per slot and pixel (HPK_FLAG_DENSE_*)
//   hpk_probe          the production records of sampled pixels (hpk_probe_sums)
//   hpk_brute          independent explicit-window sums, no summed-area table, no tiles (hpk_bruteforce_sums)
//   hpk_poisson_sf_k   the scoring kernel's Poisson survival function at given (k, lambda) (hpk_poisson_sf)
// Built into libhpk.so with -DHPK_TEST_KERNELS (the Makefile's default); without it the four entry points return HPK_ERR_INVALID.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <stdlib.h>
#define HPK_KERNEL_TU
#include "hpk_kernels.h"

namespace {
#include "hpk_device.h"

__global__ void __launch_bounds__(256) hpk_poisson_sf_k(const double* __restrict__ k, const double* __restrict__ lam,
                                                        const double* __restrict__ sfe, double* __restrict__ out,
                                                        int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = poisson_sf(k[i], lam[i], sfe, 1.0);
}


// ------------------------------------------------------------------ dense debug outputs (tests: HPK_FLAG_DENSE_*)
// Expands the stencil output to (E_K, E_Y), resolving width and (bS_K, bE_K, bS_Y, bE_Y) per slot and pixel.
__global__ void __launch_bounds__(256) hpk_dense(HpkDenseArgs a) {
    const HpkDevPlan* __restrict__ plan = a.plan;
    const int tile = blockIdx.x;
    const int cnt = (int)a.tile_cnt[tile];
    const int rb = tile / a.J, cj = tile - rb * a.J;
    const int r0 = rb * a.TR, c0 = r0 + a.mw + cj * a.TC;
    const int64_t slot_stride = (int64_t)a.n * a.ldo;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int64_t ri = (int64_t)tile * a.tilecap + i;
        const unsigned ent = a.rec_ent[ri];
        const int r = r0 + (int)HPK_ENT_Y(ent);
        const int c = c0 + (int)HPK_ENT_X(ent);
        const int d = c - r;
        const int64_t o = (int64_t)r * a.ldo + d;
        for (int q = 0; q < plan->nslots; ++q) {
            const int stp = (int)a.rec_W[q * a.rec_stride + ri];
            double2 e = make_double2(0.0, 0.0);
            double4 sm = make_double4(0.0, 0.0, 0.0, 0.0);
            uint8_t w = 0;
            if (stp != 0) {
                const double2 s2 = a.rec_S[q * a.rec_stride + ri];
                double EK, EY;
                local_expected(plan, a.etab, a.eedge, a.IR, stp - 1, r, c, d, a.n, a.num, a.mw, a.D, plan->W, EK, EY);
                const double ir = a.IR[d], b1r = a.b1[r], b2c = a.b2[c];
                e.x = (EK != 0.0) ? ((ir * (s2.x / EK)) * b1r) * b2c : 0.0;
                e.y = (EY != 0.0) ? ((ir * (s2.y / EY)) * b1r) * b2c : 0.0;
                sm = make_double4(s2.x, EK, s2.y, EY);
                w = (uint8_t)plan->steps[stp - 1].wi;
            }
            a.dE[q * slot_stride + o] = e;
            a.dW[q * slot_stride + o] = w;
            if (a.dS) a.dS[q * slot_stride + o] = sm;
        }
    }
}

// ------------------------------------------------------------------ record look-up at sampled pixels (tests)
// One wave per query pixel: finds the pixel's record in its tile's region (scan of the entries), then reports per
// slot (bS_K, bE_K, bS_Y, bE_Y, resolving width); width -1: the pixel is not a candidate (zero count or off the band).
__global__ void __launch_bounds__(64) hpk_probe(HpkDenseArgs a, const int32_t* __restrict__ rows, const int32_t* __restrict__ cols,
                                                int64_t count, double* __restrict__ out) {
    const HpkDevPlan* __restrict__ plan = a.plan;
    const int64_t qi = blockIdx.x;
    if (qi >= count) return;
    const int lane = threadIdx.x;
    const int r = rows[qi], c = cols[qi], d = c - r;
    const int nslots = plan->nslots;
    double* o = out + qi * (int64_t)nslots * 5;
    long long found = -1;
    if (r >= 0 && r < a.n && c < a.n && d >= a.mw && d <= a.D) {
        const int rb = r / a.TR, r0 = rb * a.TR, cj = (c - r0 - a.mw) / a.TC;
        const int tile = rb * a.J + cj;
        const int x = c - (r0 + a.mw + cj * a.TC), y = r - r0;
        const unsigned key = (unsigned)x | ((unsigned)y << HPK_ENT_YSHIFT);
        const int cnt = (int)a.tile_cnt[tile];
        for (int i0 = 0; i0 < cnt; i0 += 64) {
            const int i = i0 + lane;
            const bool hit = i < cnt && (a.rec_ent[(int64_t)tile * a.tilecap + i] & ((1u << HPK_ENT_CNT_SHIFT) - 1u)) == key;
            const unsigned long long m = __ballot(hit);
            if (m) { found = (long long)tile * a.tilecap + i0 + (__ffsll((long long)m) - 1); break; }
        }
    }
    if (lane != 0) return;
    for (int q = 0; q < nslots; ++q) {
        double* oq = o + q * 5;
        oq[0] = oq[1] = oq[2] = oq[3] = 0.0;
        oq[4] = found < 0 ? -1.0 : 0.0;
        if (found < 0) continue;
        const int stp = (int)a.rec_W[q * a.rec_stride + found];
        if (stp == 0) continue;
        const double2 s2 = a.rec_S[q * a.rec_stride + found];
        double EK, EY;
        local_expected(plan, a.etab, a.eedge, a.IR, stp - 1, r, c, d, a.n, a.num, a.mw, a.D, plan->W, EK, EY);
        oq[0] = s2.x; oq[1] = EK; oq[2] = s2.y; oq[3] = EY; oq[4] = (double)plan->steps[stp - 1].wi;
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

void hpk_launch_dense(const HpkDenseArgs& a, hipStream_t st) {
    if (a.ntiles <= 0) return;
    hipLaunchKernelGGL(hpk_dense, dim3(a.ntiles), dim3(256), 0, st, a);
}

void hpk_launch_probe(const HpkDenseArgs& a, const int32_t* rows, const int32_t* cols, int64_t count, double* out, hipStream_t st) {
    if (count <= 0) return;
    hipLaunchKernelGGL(hpk_probe, dim3((unsigned)count), dim3(64), 0, st, a, rows, cols, count, out);
}


void hpk_launch_poisson_sf(const double* k, const double* lam, const double* sfe, double* out, int64_t count,
                           hipStream_t st) {
    hipLaunchKernelGGL(hpk_poisson_sf_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, k, lam, sfe, out, count);
}


void hpk_launch_brute(const HpkBruteArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(hpk_brute, dim3((unsigned)((a.count + 63) / 64)), dim3(64), 0, st, a);
}
