// Kernel argument blocks and launchers (implemented in hpk_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hpk_plan.h"

#define HPK_LC 128                      // SAT columns per tile (two cells per lane)
#define HPK_LR 80                       // SAT rows per tile: 80 * 128 * 12 B = 120 KiB + 32 KiB of candidate lists
#ifndef HPK_UNIT
#define HPK_UNIT 128                    // records per scoring work unit (128 measured best of 128 | 256 | 512) (a tile has at most 64 x 127 records: <= 61 units of 128)
#endif
#define HPK_SCH 64                      // survivor slots a scoring wave reserves at a time (one batch always fits)
#define HPK_SCH_LOG2 6
#define HPK_HIST_NCAND HPK_MAX_STEPS    // hist[HPK_MAX_STEPS] counts the candidates
#define HPK_ACC_STRIDE 16                // u64 words between the chromosome's resolve totals (HpkStencilArgs::hist_acc): one per 128 bytes
#ifndef HPK_NWAVES
#define HPK_NWAVES 16                   // waves per stencil workgroup (16 or 8)
#endif
#define HPK_ROWS_PER_WAVE (64 / HPK_NWAVES)                  // output-tile rows a wave walks in phase 3 (tile rows <= 64)
// record entry of a candidate: x (7 bits) | y << 7 (6 bits: row of the output tile) | capped raw count << 13
#define HPK_LISTCAP (HPK_ROWS_PER_WAVE * 128)               // candidate ids per wave and tile

struct HpkStencilArgs {
    const float*  raw;
    const double* bal;                  // f64 band or nullptr
    const double* weight;               // f64[n] or nullptr
    const HpkDevPlan* plan;
    // Output: compact candidate records, one region of `tilecap` records per tile.  Waves reserve their share of a
    // region with one atomicAdd on tile_cnt[tile] (distinct addresses per tile: no serialisation).
    //   rec_ent[tile * tilecap + i]                    x | row of the output tile << 7 | min(raw, HPK_PK_CAP) << 13
    //   rec_S[slot * rec_stride + tile * tilecap + i]  (bS_K, bS_Y) at the resolving step
    //   rec_W[slot * rec_stride + tile * tilecap + i]  resolving step + 1, 0 = unresolved
    unsigned* rec_ent;
    double2* rec_S;
    uint8_t* rec_W;
    unsigned* tile_cnt;
    uint2* units;                       // scoring work list: one entry per 256 records of a tile (appended at tile end)
    unsigned* nunits;
    int32_t tilecap;
    int64_t rec_stride;
    uint8_t* gap;                       // [n] preset to 0; set for rows with a non-zero balanced value (gap = !flag)
    unsigned long long* hist;           // [HPK_MAX_STEPS + 1] totals, written by hpk_freeze
    unsigned* hist_part;                // [grid][HPK_MAX_STEPS + 1] per-workgroup resolve counts, [..][HPK_MAX_STEPS] = candidates
    unsigned long long* hist_acc;       // or (scoring follows): [(HPK_MAX_STEPS + 1) * HPK_ACC_STRIDE] zeroed totals the workgroups add to
    unsigned* ticket;                   // zeroed; the workgroup that draws grid - 1 runs the freeze (nullptr: hpk_freeze follows)
    int32_t* frozen;                    // outputs of the freeze: frozen_w, executed[nsteps], first empty step + 1 or 0
    int32_t* executed;
    int32_t* err;
    double risk;                        // box sums below risk x (largest table entry of the window) are redone exactly
    int32_t n, num;
    int64_t ld, ldo;
    int32_t W, mw, D;
    int32_t TR, TC;                     // output tile = (HPK_LR - 2W - 1) x (HPK_LC - 2W - 1)
    int32_t J;                          // column chunks per row block
    int32_t ntiles, chunk;              // chunk = ceil(ntiles / 8): tiles handed to one XCD
    int32_t grid;                       // persistent workgroups (multiple of 8, one per CU)
    int32_t single;                     // the plan is a textbook single-pair plan (HpkDevPlan::single_p >= 0)
    int32_t order;                      // tile order within an XCD's run: 0 row-major, 1 column chunks rotated per row block
    int32_t wguess;                     // hpk_stencil_s: records only for candidates whose first sufficient width is <= wguess (255: all)
    unsigned long long* clk;            // -DHPK_PHASE_CLOCK builds: [grid][waves][8] cycle sums per phase, or nullptr
    int32_t dbg_stop;                   // profiling ablation (HPK_DBG_STOP): 1 stop after the loads, 2 after the SAT,
                                        // 4 no candidates (loads + SAT + zero stores), 5 search without box sums
};

// One pixel that can still end with q <= sig (40 bytes).
struct HpkSurv {
    int32_t x, y;
    float O;
    uint8_t set, chunk, flag, pad;
    double E, p, bal;
};
#define HPK_NFAM (2 * HPK_MAX_PAIRS * (HPK_NB + 1))     // (set, chunk) families
#define HPK_HSHIFT 1                                     // hpk_score's p-value histogram: bins a factor 4 wide (8 bins reach sig 2^-14; fewer, fuller bins to flush)
#define HPK_TIGHTEN_MAX 16                               // counter arrays per family: exact rounds, or the bins of the one-pass histogram
#define HPK_NREG 64                     // independent survivor regions (one reservation counter each, 256 B apart)
#define HPK_REG_STRIDE 32               // counters are u64[HPK_NREG * HPK_REG_STRIDE]

struct HpkScoreArgs {
    const float*  raw;
    const double* bal;
    const double* weight;
    const unsigned* rec_ent;            // candidate records written by hpk_stencil
    const double2* rec_S;
    const uint8_t* rec_W;
    const unsigned* tile_cnt;
    const uint2* units;                 // work list {tile, unit | records of the tile << 8}, appended by hpk_stencil
    const unsigned* nunits;
    int32_t tilecap;
    int64_t rec_stride;
    int32_t ntiles, TR, TC, J, W;       // tile geometry
    const HpkDevPlan* plan;
    const double* etab;                 // [nsteps][2][D + 1]
    const double* eedge;                // [2][W][nsteps][2][D + 1] windows clipped by the first rows / last columns
    const double* IR;
    const double* b1;
    const double* b2;
    int32_t* frozen;                    // device scalar: written by the freeze kernel, or by this kernel's first workgroup (hist_acc)
    const unsigned long long* hist_acc; // the stencil's resolve totals: every workgroup replays the freeze decision on them (nullptr: read `frozen`)
    unsigned long long* hist_out;       // with hist_acc: totals, executed flags and error for the host, written by the first workgroup
    int32_t* executed;
    int32_t* err;
    const double* bounds;               // [HPK_NB] chunk upper bounds
    const double* ptab;                 // Poisson survival table
    const int32_t* ptab_off;            // [HPK_NB_TAB + 2]
    const double* sfe;                  // stirlerr(0..31)
    double sig;
    int32_t n, num;
    int64_t ld, ldo;
    int32_t mw, D;
    // outputs
    unsigned int* fam_m;                // [HPK_NFAM] tests per family (chunk sizes, callers.py:266)
    unsigned int* fam_f;                // [HPK_NFAM] of those, p <= sig
    unsigned long long* emax_bits;      // [nsets]
    unsigned long long* nvalid;         // [nsets]
    unsigned long long* nsurv;          // [HPK_NREG * HPK_REG_STRIDE] reserved slots per region
    int64_t cap;                        // survivor capacity per region (multiple of 256)
    HpkSurv* surv;
    unsigned* chunk_used;               // [HPK_NREG * cap / HPK_SCH] filled slots per chunk
    unsigned int* hist;                 // [nsets * (HPK_NB + 1)][hbins] p-values <= sig by family and log2 bin (zeroed), or unused
    int32_t hbins;                      // bins per family of `hist`, 0 = no histogram (hpk_thr_hist / counting rounds do the cut)
    int32_t nsets_half;                 // (pw, ww) pairs of the call: the launcher sizes the histogram's LDS with it
    int32_t dbg;                        // unused (the ablation switches of round 1 cost scalar work in the hot loop)
};

struct HpkDenseArgs {
    const unsigned* rec_ent; const double2* rec_S; const uint8_t* rec_W; const unsigned* tile_cnt;
    int32_t tilecap; int64_t rec_stride; int32_t ntiles, TR, TC, J;
    const HpkDevPlan* plan; const double* etab; const double* eedge;
    const double* IR; const double* b1; const double* b2;
    int32_t n, num; int64_t ldo; int32_t mw, D;
    double2* dE; uint8_t* dW; double4* dS;
};

struct HpkBruteArgs {
    const float* raw; const double* bal; const double* weight; const double* IR;
    const HpkDevPlan* plan;
    int32_t n, num; int64_t ld; int32_t step;
    const int32_t* rows; const int32_t* cols; int64_t count;
    double* out;                        // [count][5]
};

int  hpk_stencil_lds_bytes();
void hpk_launch_stencil(const HpkStencilArgs& a, bool balf64, bool simple, hipStream_t st);
bool hpk_stencil_s_applies(const HpkStencilArgs& a, bool simple);   // the second-generation kernel takes this launch
void hpk_launch_dense(const HpkDenseArgs& a, hipStream_t st);
void hpk_launch_probe(const HpkDenseArgs& a, const int32_t* rows, const int32_t* cols, int64_t count, double* out, hipStream_t st);
void hpk_launch_freeze_tot(const HpkDevPlan* plan, const unsigned long long* hist_acc, unsigned long long* hist,
                           int32_t* frozen, int32_t* executed, int32_t* err, hipStream_t st);
void hpk_launch_freeze(const HpkDevPlan* plan, unsigned long long* hist, const unsigned* hist_part, int nparts,
                       int32_t* frozen, int32_t* executed, int32_t* err, hipStream_t st);
void hpk_launch_prep(const float* raw, const double* weight, int n, int num, int64_t ld, int mw, double* psum, unsigned* pnan,
                     double* IR, double* bias, hipStream_t st);
void hpk_launch_etab(const HpkDevPlan* plan, int nsteps, int D, int W, const double* IR, int n, int num, double* etab,
                     double* eedge, void* zero, size_t zero_bytes, hipStream_t st);
void hpk_launch_gap(const float* raw, const double* bal, const double* weight, int32_t n, int32_t num,
                    int64_t ld, int32_t mw, uint8_t* gap, hipStream_t st);
void hpk_launch_score(const HpkScoreArgs& a, bool bhfdr, int cus, hipStream_t st);
// Benjamini-Hochberg cut tightening on the survivor list: thr[f] <- sig * #{p <= thr[f]} / m[f], `rounds` times,
// then compaction of the records with p <= thr[f] into `out` (count in *nout).
int  hpk_thr_hist_bins(int nsets);       // bins per family of the one-pass tightening (rounds < 0)
int  hpk_score_hist_bins(int nsets);     // bins per family of the histogram hpk_score keeps (rounds <= -100)
void hpk_launch_tighten(const HpkSurv* surv, const unsigned long long* nsurv, int64_t cap, const unsigned* chunk_used,
                        const unsigned int* fam_m, const unsigned int* fam_f, unsigned int* fam_cnt, double sig, int rounds,
                        int nsets, HpkSurv* out_head, unsigned long long inl, HpkSurv* out_rest, unsigned long long* nout,
                        const double* bal, const double* weight, int64_t ld, int cus, hipStream_t st);
void hpk_launch_publish(const void* src, void* dst_host_mapped, size_t bytes, hipStream_t st);
void hpk_launch_publish_head(const void* src, void* dst_host_mapped, const size_t seg[3][2], size_t inl_begin,
                             const unsigned long long* nout, unsigned inl, unsigned recbytes, hipStream_t st);
void hpk_launch_ptab(const double* bounds, const int32_t* off, const double* sfe, double* ptab, int32_t total,
                     hipStream_t st);
void hpk_launch_poisson_sf(const double* k, const double* lam, const double* sfe, double* out, int64_t count,
                           hipStream_t st);
void hpk_launch_brute(const HpkBruteArgs& a, hipStream_t st);
