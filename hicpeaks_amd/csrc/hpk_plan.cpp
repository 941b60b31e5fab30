// Widening plan builder (host).  See hpk_plan.h.
//
// The reference walks a (2w+1)^2 window cell by cell and adds one shifted copy of the band per cell
// (hicpeaks/callers.py:147-198).  Which cells are visited at a step, and with which sign, depends only on
// the cell's Chebyshev radius rho = max(|di|, |dj|) ("bgloc", callers.py:149/176), on whether it lies on
// the centre cross and on whether it lies in the lower-left quadrant - so the accumulated matrices are
// sum_rho m_rho * ring_rho with small integer multiplicities m_rho.  This file replays the reference's
// control flow on rings instead of cells and converts the multiplicities to box terms
// sum_rho m_rho * ring_rho = sum_rho (m_rho - m_{rho+1}) * Box_rho.
#include "hpk_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct RawStep { int p, w; };

void box_terms(const int* m, int W, int32_t* n_out, int32_t* rho_out, int32_t* coef_out) {
    int n = 0;
    for (int rho = 1; rho <= W; ++rho) {
        int next = (rho < W) ? m[rho + 1] : 0;
        int c = m[rho] - next;
        if (c != 0) { rho_out[n] = rho; coef_out[n] = c; ++n; }
    }
    *n_out = n;
}

}  // namespace

int hpk_build_plan(const hpk_params* prm, HpkDevPlan* plan, char* msg) {
    std::memset(plan, 0, sizeof(*plan));
    msg[0] = 0;
    const int W = prm->maxww;
    if (prm->npairs < 1 || prm->npairs > HPK_MAX_PAIRS) { std::snprintf(msg, 256, "npairs must be 1..%d", HPK_MAX_PAIRS); return HPK_ERR_INVALID; }
    if (W < 1 || W > HPK_MAX_W) { std::snprintf(msg, 256, "maxww must be 1..%d", HPK_MAX_W); return HPK_ERR_INVALID; }
    if (prm->res <= 0 || prm->maxapart < 0) { std::snprintf(msg, 256, "res/maxapart invalid"); return HPK_ERR_INVALID; }
    if (prm->mode != HPK_MODE_HICCUPS && prm->mode != HPK_MODE_BHFDR) { std::snprintf(msg, 256, "unknown mode"); return HPK_ERR_INVALID; }
    if (prm->mode == HPK_MODE_BHFDR && prm->npairs != 1) { std::snprintf(msg, 256, "bhfdr takes one (pw, ww) pair"); return HPK_ERR_INVALID; }
    int mw = prm->ww[0], maxw = prm->ww[0], minp = prm->pw[0];
    for (int i = 0; i < prm->npairs; ++i) {
        if (prm->pw[i] < 0 || prm->ww[i] < 1) { std::snprintf(msg, 256, "pw >= 0 and ww >= 1 required"); return HPK_ERR_INVALID; }
        mw = std::min(mw, prm->ww[i]);
        maxw = std::max(maxw, prm->ww[i]);
        minp = std::min(minp, prm->pw[i]);
    }
    plan->mode = prm->mode;
    plan->W = W;
    plan->mw = mw;
    plan->maxw = maxw;
    plan->D = (int32_t)(prm->maxapart / prm->res);
    plan->min_reads = (prm->mode == HPK_MODE_BHFDR) ? 16 : prm->min_local_reads;   // callers.py:490
    plan->pk_cap = std::max<int32_t>((int32_t)HPK_PK_CAP, plan->min_reads);
    {   // a box of (2 maxww + 1)^2 capped counts must fit the packed plane's field
        const int64_t cells = (int64_t)(2 * W + 1) * (2 * W + 1);
        const int64_t most = ((int64_t)1 << HPK_PK_BITS) - 1;
        if (plan->min_reads < 0 || cells * plan->pk_cap > most) {
            std::snprintf(msg, 256, "min_local_reads = %d with maxww = %d: the stencil sums capped counts in %d bits, at most %lld is supported at this maxww",
                          plan->min_reads, W, HPK_PK_BITS, (long long)(most / cells));
            return HPK_ERR_INVALID;
        }
    }
    plan->npairs = prm->npairs;

    // output slots: distinct peak widths in order of appearance (the bSV / bEV / RefIdx dict keys, callers.py:107-119)
    for (int i = 0; i < prm->npairs; ++i) {
        int s = -1;
        for (int j = 0; j < plan->nslots; ++j) if (plan->slot_pi[j] == prm->pw[i]) s = j;
        if (s < 0) { s = plan->nslots++; plan->slot_pi[s] = prm->pw[i]; }
        plan->pair_slot[i] = s;
        plan->pair_wi[i] = prm->ww[i];
    }
    if (plan->nslots > HPK_KSLOTS) { std::snprintf(msg, 256, "at most %d distinct peak widths", HPK_KSLOTS); return HPK_ERR_INVALID; }

    // step order
    std::vector<RawStep> order;
    if (prm->mode == HPK_MODE_HICCUPS) {                               // pw_ww_pairs, callers.py:15-23
        for (int i = 0; i < prm->npairs; ++i)
            for (int w = prm->ww[i]; w <= W; ++w) order.push_back({prm->pw[i], w});
        std::stable_sort(order.begin(), order.end(), [](const RawStep& a, const RawStep& b) {
            return a.w != b.w ? a.w < b.w : a.p < b.p; });
    } else {                                                            // callers.py:440
        for (int w = prm->ww[0]; w <= W; ++w) order.push_back({prm->pw[0], w});
    }
    if ((int)order.size() > HPK_MAX_STEPS) { std::snprintf(msg, 256, "plan has %zu steps, max %d", order.size(), HPK_MAX_STEPS); return HPK_ERR_INVALID; }
    plan->nsteps = (int32_t)order.size();

    int mK[HPK_MAX_W + 2] = {0};
    int mR[HPK_MAX_W + 2] = {0};
    bool limit = false;
    int last_pi = 0, last_wi = 0;
    int reads_id = -1;
    for (int s = 0; s < plan->nsteps; ++s) {
        const int pi = order[s].p, wi = order[s].w;
        bool reads_changed = false;
        for (int rho = 1; rho <= wi; ++rho) {
            if (prm->mode == HPK_MODE_HICCUPS) {
                if (limit && (((rho <= last_wi) && (rho > std::max(pi, last_pi))) || (rho <= std::min(pi, last_pi))))
                    continue;                                           // callers.py:150-152
                if (rho > pi) {                                         // outside P1 (callers.py:138, 179)
                    bool plus = (!limit) || rho > last_wi || (rho > pi && rho <= last_pi);   // callers.py:180, 187
                    mK[rho] += plus ? 1 : -1;
                    if ((!limit) || (pi == minp && rho > last_wi)) { mR[rho] += 1; reads_changed = true; }  // 197
                }
            } else {
                if (limit && rho < wi) continue;                        // callers.py:455
                if (rho > pi) { mK[rho] += 1; mR[rho] += 1; reads_changed = true; }   // 481-485
            }
        }
        limit = true;
        last_pi = pi;
        last_wi = wi;
        if (reads_changed || reads_id < 0) ++reads_id;

        HpkDevStep& st = plan->steps[s];
        st.pi = pi;
        st.wi = wi;
        st.reads_id = reads_id;
        st.slot = -1;
        for (int j = 0; j < plan->nslots; ++j) if (plan->slot_pi[j] == pi) st.slot = j;
        for (int rho = 0; rho <= W; ++rho) {
            if (mK[rho] < 0 || mR[rho] < 0) {
                std::snprintf(msg, 256, "step (%d,%d): negative ring multiplicity at radius %d", pi, wi, rho);
                return HPK_ERR_PLAN;
            }
            st.m[rho] = mK[rho];
            st.mr[rho] = mR[rho];
        }
        box_terms(mK, W, &st.nkt, st.kt_rho, st.kt_coef);
        box_terms(mR, W, &st.nrt, st.rt_rho, st.rt_coef);
        if (st.nkt > HPK_PK_KT || st.nrt > HPK_PK_RT || reads_id > 63) {
            std::snprintf(msg, 256, "step (%d,%d) needs %d donut / %d Reads box terms; the kernel packs %d / %d", pi, wi,
                          st.nkt, st.nrt, HPK_PK_KT, HPK_PK_RT);
            return HPK_ERR_PLAN;
        }
        uint32_t* pk = plan->packed[s];
        int rho_min = 31;                       // smallest ring with a non-zero multiplicity
        for (int rho = W; rho >= 1; --rho) if (mK[rho] > 0) rho_min = rho;
        pk[0] = (uint32_t)st.slot | (uint32_t)wi << 4 | (uint32_t)reads_id << 10 | (uint32_t)st.nrt << 16 |
                (uint32_t)st.nkt << 20 | (uint32_t)rho_min << 24;
        for (int j = 0; j < st.nrt; ++j) {
            if (st.rt_coef[j] < -128 || st.rt_coef[j] > 127) { std::snprintf(msg, 256, "box coefficient out of range"); return HPK_ERR_PLAN; }
            pk[1 + j / 2] |= ((uint32_t)st.rt_rho[j] | ((uint32_t)(uint8_t)(int8_t)st.rt_coef[j]) << 8) << (16 * (j & 1));
        }
        for (int j = 0; j < st.nkt; ++j) {
            if (st.kt_coef[j] < -128 || st.kt_coef[j] > 127) { std::snprintf(msg, 256, "box coefficient out of range"); return HPK_ERR_PLAN; }
            pk[3 + j / 2] |= ((uint32_t)st.kt_rho[j] | ((uint32_t)(uint8_t)(int8_t)st.kt_coef[j]) << 8) << (16 * (j & 1));
        }
    }

    // simple-Reads detection
    std::memset(plan->step_of, 0xff, sizeof(plan->step_of));
    bool simple = plan->nsteps > 0;
    int p0 = -1;
    int lastw[HPK_KSLOTS];
    for (int q = 0; q < HPK_KSLOTS; ++q) { lastw[q] = -1; plan->slot_wfirst[q] = 0; }
    plan->wmin = plan->nsteps ? plan->steps[0].wi : 0;
    for (int s = 0; s < plan->nsteps && simple; ++s) {
        const HpkDevStep& st = plan->steps[s];
        int lo = -1;
        for (int rho = 1; rho <= W; ++rho) if (st.mr[rho]) { lo = rho; break; }
        if (lo < 0) { simple = false; break; }
        if (p0 < 0) p0 = lo - 1;
        for (int rho = 0; rho <= W; ++rho)
            if (st.mr[rho] != ((rho > p0 && rho <= st.wi) ? 1 : 0)) simple = false;
        if (st.wi <= lastw[st.slot]) simple = false;               // widths must grow within a slot
        if (lastw[st.slot] < 0) plan->slot_wfirst[st.slot] = st.wi;
        lastw[st.slot] = st.wi;
        if (st.wi < plan->wmin) simple = false;                     // plan order is by width
        if (simple) plan->step_of[st.slot][st.wi] = (uint8_t)s;
    }
    // every width from a slot's first one up to W must have a step (so that max(w*, wfirst) always maps)
    for (int q = 0; q < plan->nslots && simple; ++q)
        for (int w = plan->slot_wfirst[q]; w <= W; ++w) if (plan->step_of[q][w] == 0xff) simple = false;
    plan->simple_reads = simple ? 1 : 0;
    plan->reads_p0 = simple ? p0 : 0;

    // textbook single-pair plans
    plan->single_p = -1;
    if (simple && plan->nslots == 1 && plan->nsteps == W - plan->wmin + 1) {
        const int p = plan->slot_pi[0];
        bool ok = true;
        for (int s = 0; s < plan->nsteps && ok; ++s) {
            const HpkDevStep& st = plan->steps[s];
            ok = st.wi == plan->wmin + s && st.wi > p;
            if (p > 0) ok = ok && st.nkt == 2 && st.kt_rho[0] == p && st.kt_coef[0] == -1 && st.kt_rho[1] == st.wi && st.kt_coef[1] == 1;
            else ok = ok && st.nkt == 1 && st.kt_rho[0] == st.wi && st.kt_coef[0] == 1;
        }
        if (ok) plan->single_p = p;
    }

    plan->first_rho = 0;
    if (plan->nsteps > 0 && plan->steps[0].nkt > 0) {
        int fr = plan->steps[0].kt_rho[0];
        for (int s = 0; s < plan->nsteps; ++s)
            if (plan->steps[s].nkt < 1 || plan->steps[s].kt_rho[0] != fr) fr = 0;
        plan->first_rho = fr;
    }

    // cell counts per diagonal offset for the local-expected tables
    for (int s = 0; s < plan->nsteps; ++s) {
        const HpkDevStep& st = plan->steps[s];
        for (int rho = 1; rho <= W; ++rho) {
            const int m = st.m[rho];
            if (m == 0) continue;
            for (int di = -rho; di <= rho; ++di)
                for (int dj = -rho; dj <= rho; ++dj) {
                    if (std::max(std::abs(di), std::abs(dj)) != rho || di == 0 || dj == 0) continue;
                    plan->ecoef[s][0][dj - di + 2 * W] += (int16_t)m;
                    if (di > 0 && dj < 0) plan->ecoef[s][1][dj - di + 2 * W] += (int16_t)m;
                }
        }
    }
    return HPK_OK;
}

void hpk_build_etab(const HpkDevPlan* plan, const double* IR, int32_t num, double* etab) {
    const int W = plan->W, D = plan->D, mw = plan->mw;
    const int span = 4 * W + 1;                 // delta = dj - di in [-2W, 2W]
    for (int s = 0; s < plan->nsteps; ++s)
        for (int fl = 0; fl < 2; ++fl) {
            double* t = etab + (size_t)(s * 2 + fl) * (D + 1);
            for (int d = 0; d <= D; ++d) {
                double acc = 0.0;
                for (int k = 0; k < span; ++k) {
                    const int kk = d + k - 2 * W;
                    const int cf = plan->ecoef[s][fl][k];
                    if (cf == 0 || kk < mw || kk >= num) continue;
                    acc += (double)cf * IR[kk];
                }
                t[d] = acc;
            }
        }
}
