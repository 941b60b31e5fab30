// Host-side widening plan: the as-coded step sequence of hicpeaks/callers.py:15-23 + 132-201 (hiccups)
// and callers.py:440-485 (bhfdr) reduced to per-ring multiplicities, plus the box-term form the
// kernels consume.  Shared by host (.cpp) and device (.hip) code: plain structs only.
#pragma once
#include <stdint.h>
#include "../../include/hpk.h"

#define HPK_KSLOTS 4            // distinct peak widths the stencil kernel keeps in registers
#define HPK_NB     128          // lambda-chunk boundaries 2^((i-1)/3), i = 1..HPK_NB
#define HPK_NB_TAB 46           // chunks served from the device-built Poisson table (rv <= 2^15)
#define HPK_PK_CAP 1023u        // raw counts enter the stencil's packed SAT plane capped at max(this, min_local_reads): HpkDevPlan::pk_cap
#define HPK_PK_BITS 21          // width of the capped-count field of the packed plane: a box of (2 maxww + 1)^2 capped cells must fit

struct HpkDevStep {
    int32_t slot;               // output slot = index of pi among the distinct peak widths
    int32_t pi, wi;
    int32_t reads_id;           // steps with equal id see the same Reads matrix
    int32_t nkt;                // donut / lower-left:  sum_j kt_coef[j] * Box(kt_rho[j])
    int32_t kt_rho[HPK_MAX_W + 1];
    int32_t kt_coef[HPK_MAX_W + 1];
    int32_t nrt;                // Reads:               sum_j rt_coef[j] * BoxLL(rt_rho[j])
    int32_t rt_rho[HPK_MAX_W + 1];
    int32_t rt_coef[HPK_MAX_W + 1];
    int32_t m[HPK_MAX_W + 1];   // cumulative ring multiplicity after this step (edge path, E tables)
    int32_t mr[HPK_MAX_W + 1];  // same for Reads
};

struct HpkDevPlan {
    int32_t mode;
    int32_t nsteps;
    int32_t nslots;
    int32_t W;                  // maxww
    int32_t mw;                 // min(ww)
    int32_t maxw;               // max(ww)
    int32_t D;                  // maxapart / res
    int32_t min_reads;
    int32_t slot_pi[HPK_MAX_PAIRS];
    int32_t npairs;
    int32_t pair_slot[HPK_MAX_PAIRS];
    int32_t pair_wi[HPK_MAX_PAIRS];
    HpkDevStep steps[HPK_MAX_STEPS];
    // The same steps packed for the stencil kernel, which keeps step s in lane s of 7 VGPRs and reads it back
    // with v_readlane (no memory access inside the per-pixel loop):
    //   [0]    slot | wi << 4 | reads_id << 10 | nrt << 16 | nkt << 20 | rho_min << 24 (smallest ring with m > 0)
    //   [1..2] up to HPK_PK_RT Reads box terms, 16 bits each: rho | (int8 coef) << 8
    //   [3..6] up to HPK_PK_KT donut box terms, same encoding
    uint32_t packed[HPK_MAX_STEPS][8];
    // "Simple Reads" plans (every reference configuration with pairs listed in increasing order): the Reads matrix
    // of a step of width wi is the lower-left rings p0+1 .. wi, so a pixel's first sufficient width w* does not
    // depend on the slot and the resolving step of slot q is step_of[q][max(w*, slot_wfirst[q])].
    int32_t simple_reads;
    int32_t reads_p0;
    int32_t wmin;                              // smallest step width
    int32_t slot_wfirst[HPK_KSLOTS];
    uint8_t step_of[HPK_KSLOTS][32];           // 0xff = no such step
    // Local-expected stencils for the device-side table build: the window of step s holds ecoef[s][fl][t] cells
    // (with multiplicity) at diagonal offset delta = t - 2W, so bE[s][fl](d) = sum_t ecoef * IR[d + t - 2W].
    int16_t ecoef[HPK_MAX_STEPS][2][4 * HPK_MAX_W + 1];
    // "Textbook" plans - one peak width, every step (p, w) the plain donut Box(w) - Box(p) with steps at consecutive
    // widths from wmin on (every single-pair run of hiccups() and every bhfdr() run): the peak width p, else -1.  The
    // stencil then needs no per-candidate plan look-up: step = w* - wmin, sums = box(w*) - box(p).
    int32_t single_p;
    // Every step's first (innermost) donut box has this radius and a non-zero coefficient, else 0.  True of every plan
    // with min(pw) >= 1: the innermost ring is min(pw) + 1 wide whatever the step.  The stencil then forms that box once
    // per candidate and reuses it in every slot.
    int32_t first_rho;
    // Counts enter the packed plane (and the record entries) capped here: max(HPK_PK_CAP, min_reads).  Only
    // Reads >= min_reads is ever asked of them, and a cell capped at or above the threshold decides that like the true count.
    int32_t pk_cap;
};
#define HPK_PK_RT 4
#define HPK_PK_KT 8

// Returns HPK_OK or a negative status; msg (>= 256 bytes) receives the reason.
int hpk_build_plan(const hpk_params* prm, HpkDevPlan* plan, char* msg);

// Interior local-expected sums: etab[(s * 2 + fl) * (D + 1) + d] = sum over the window cells of step s
// (with multiplicity) of IR[d + dj - di]; valid for pixels at least W bins away from both matrix ends.
void hpk_build_etab(const HpkDevPlan* plan, const double* IR, int32_t num, double* etab);   // host version (tests)
