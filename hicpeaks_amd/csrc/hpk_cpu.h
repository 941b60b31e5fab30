// Back-end #0 of the C ABI (include/hpk.h: hpk_create(-1, ...)): the same path on host
// cores - plain C++ on threads, the as-coded algorithm of hicpeaks/callers.py:44-590 in dense-band form (ring multiplicities of
// hpk_plan.cpp, window cells added one by one as the reference's CSR adds do).  It exists to be measured beside the GPU path
// (bench.py: cpu_baseline.kind = "native") and to put the parity ladder within reach of a machine without a GPU; it is never
// selected implicitly - a context is a CPU context only when the caller asked for device -1 - and shares nothing with oracle/.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/hpk.h"
#include "hpk_kernels.h"
#include "hpk_plan.h"

// Poisson survival tables of the lambda chunks (chunk ch: lambda = bounds[ch - 1], entries k = 0 .. len - 1; beyond: exactly 0), as
// hpk_ptab builds them on the device; filled once per set of bounds.
struct HpkCpuTables {
    std::vector<double> bounds, sfe;
    std::vector<int32_t> off;           // [HPK_NB_TAB + 2]
    std::vector<double> ptab;
    bool built = false;
};
void hpk_cpu_build_tables(HpkCpuTables& t, int threads);
// 1 - cdf(k; lam) as the kernels form it (hpk_device.h: poisson_sf), on the host
double hpk_cpu_poisson_sf(double k, double lam, const double* sfe, double sigcap);

// What the device path leaves in a band's counter block and survivor regions, for the host half both back-ends share
// (hpk_api.cpp: assemble_sets).
struct HpkCpuOut {
    unsigned long long hist[HPK_MAX_STEPS + 1];     // candidates resolved per step | candidates
    int32_t frozen = 0, err = 0;
    int32_t exec[HPK_MAX_STEPS];
    std::vector<unsigned long long> emax;           // [nsets] bit pattern of the largest E
    std::vector<uint32_t> fam_m, fam_f;             // [nsets][HPK_NB + 1] tests per family | of those p <= sig
    std::vector<uint8_t> rowlive;                   // [n] the row holds a non-zero balanced value (gap = !rowlive)
    std::vector<HpkSurv> surv;                      // every pixel with p <= sig
    std::vector<double> IR, b1, b2;                 // as derived, when the band came without them
    int64_t band_px = 0;
    double ms_sums = 0.0, ms_score = 0.0;
};
// One chromosome (host arrays).  Returns HPK_OK or a status with `err` set; an empty step (callers.py:203-208) is reported through
// out.err like the device path's counter block.
int hpk_cpu_band(const hpk_band& in, const hpk_params& prm, const HpkDevPlan& plan, const HpkCpuTables& tabs, int threads,
                 HpkCpuOut& out, std::string& err);
