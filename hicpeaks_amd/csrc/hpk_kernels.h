// Kernel argument blocks and launchers (implemented in hpk_kernels.hip).
//
// Since round 3 every production kernel works on a *batch* of chromosomes: the host describes each band of the batch
// in one HpkBandDesc (device memory), the kernels take the descriptor array and find their band by a grid coordinate
// (prep, expected tables, scoring, cut, publish) or - the stencil - walk the tiles of all bands in one persistent
// launch.  A single chromosome is a batch of one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hpk_plan.h"

#define HPK_LC 160                      // SAT columns per tile: a table row per DPP row of 16 lanes, ten consecutive cells per lane
#define HPK_LR 64                       // SAT rows per tile (four per wave): 64 * 160 * 12 B = 120 KiB + 30 KiB of candidate list
#ifndef HPK_UNIT
#define HPK_UNIT 128                    // records per scoring work unit (128 measured best of 128 | 256 | 512) (a tile has at most 64 x 127 records: <= 61 units of 128)
#endif
#define HPK_SCH 64                      // survivor slots a scoring wave reserves at a time (one batch always fits)
#define HPK_SCH_LOG2 6
#define HPK_HIST_NCAND HPK_MAX_STEPS    // hist[HPK_MAX_STEPS] counts the candidates
#define HPK_HREP 16                      // copies of a chromosome's resolve totals (HpkBandDesc::hist_acc): workgroup g adds to copy g % HPK_HREP
#define HPK_HREP_OF(g) ((g) & (HPK_HREP - 1))
#define HPK_ACC_STRIDE 80                // u64 words from one copy to the next: HPK_MAX_STEPS + 1 totals, whole 128-byte lines
#ifndef HPK_NWAVES
#define HPK_NWAVES 16                   // waves per stencil workgroup (16 or 8)
#endif
// record entry of a candidate: x (8 bits) | y << 8 (6 bits: row of the output tile) | capped raw count << 14 (<= pk_cap < 2^13)
#define HPK_ENT_X(e) ((e) & 255u)
#define HPK_ENT_YSHIFT 8
#define HPK_ENT_Y(e) (((e) >> HPK_ENT_YSHIFT) & 63u)
#define HPK_ENT_CNT_SHIFT 14
#define HPK_HACC 72                     // words of one of hpk_stencil_s's two LDS buffers of resolve counts: 64 widths / steps, the candidates
#define HPK_TLIST 7680                  // entries of hpk_stencil_s's tile-wide candidate list: TR * TC must fit

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

// Counter block of one band ("small"): fixed part, byte offsets.  Host and device share the layout; the head of the
// block (counters | row flags | first compacted survivors) is what hpk_publish copies to pinned host memory.
#define HPK_OFF_HIST    0                                                   // u64[65] resolve totals + candidates, for the host
#define HPK_OFF_FROZEN  (HPK_OFF_HIST + 8 * (HPK_MAX_STEPS + 1))            // i32
#define HPK_OFF_ERR     (HPK_OFF_FROZEN + 8)                                // i32
#define HPK_OFF_EXEC    (HPK_OFF_ERR + 8)                                   // i32[64]
#define HPK_OFF_NUNITS  (HPK_OFF_EXEC + 4 * HPK_MAX_STEPS)                  // u32 (+ HPK_OFF_BCLASS): survives the overflow rerun
#define HPK_OFF_BCLASS  (HPK_OFF_NUNITS + 4)                                // u32: 0x10000 | depth class << 8 | the record bound hpk_band_class gave the band (0: none)
#define HPK_OFF_LEAN    (HPK_OFF_NUNITS + 8)                                // u32[4]: tiles built without the f64 plane | of those, computed once more in full | candidates summed cell by cell | -
#define HPK_OFF_NSURV   (HPK_OFF_NUNITS + 24)                               // u64[HPK_NREG * HPK_REG_STRIDE] ... from here on reset by the rerun
#define HPK_OFF_NVALID  (HPK_OFF_NSURV + 8 * HPK_NREG * HPK_REG_STRIDE)     // u64[16]
#define HPK_OFF_EMAX    (HPK_OFF_NVALID + 8 * 2 * HPK_MAX_PAIRS)            // u64[16]
#define HPK_OFF_NOUT    (HPK_OFF_EMAX + 8 * 2 * HPK_MAX_PAIRS)              // u64
#define HPK_OFF_SPECFAIL (HPK_OFF_NOUT + 8)                                 // u32 (+ pad): a family's cut lies above the bound its survivors were written to
#define HPK_OFF_TBIN    (HPK_OFF_SPECFAIL + 8)                              // u8[HPK_NFAM]: histogram bin of every family's cut (hpk_thr_compact)
#define HPK_OFF_FAM_M   (HPK_OFF_TBIN + (HPK_NFAM + 15) / 16 * 16)          // u32[HPK_NFAM]
#define HPK_OFF_FAM_F   (HPK_OFF_FAM_M + 4 * HPK_NFAM)                      // u32[HPK_NFAM]
#define HPK_SMALL_BYTES (HPK_OFF_FAM_F + 4 * HPK_NFAM)
#define HPK_HEAD_INLINE 4096            // compacted survivors that travel to the host with the counters

// Pointers inside a band descriptor are declared in the global address space for device code: a pointer read out of
// memory is otherwise a generic pointer to the compiler, which then issues flat_load / flat_store (counted on vmcnt
// *and* lgkmcnt - every LDS wait would also wait for the record stores and the prefetch).  The layout is the same on
// both sides (8-byte pointers); kernels turn a field into an ordinary pointer with gptr().
#if defined(__HIP_DEVICE_COMPILE__) && defined(HPK_KERNEL_TU)      // (hpk_kernels.hip defines it; hipcc's device pass also parses the host-only files)
#define HPK_GP(T) __attribute__((address_space(1))) T*
#else
#define HPK_GP(T) T*
#endif

// Tile geometry of hpk_stencil_s under a halo of Wh widths.  Output tile: what the halo leaves of the 64 x 160 table, as many
// rows as the tile-wide candidate list holds (HPK_TLIST entries; 6 bits of the record entry).  The tiles of a row block reach
// the last stored diagonal D + maxww (gap rows, callers.py:238) through the last tile's right halo: with a halo below maxww
// the chunks themselves have to go further (Dg).  One function for the host (sizes, the batch's default) and the device
// (hpk_band_class: every band's own halo).
struct HpkGeo { int32_t W, Dg, TR, TC, J, tilecap; };
__host__ __device__ inline HpkGeo hpk_geo_of(int Wh, int planW, int D, int mw, int tr_cap) {
    HpkGeo g;
    g.W = Wh; g.Dg = D + (planW > Wh ? planW - Wh : 0);
    g.TC = HPK_LC - 2 * Wh - 1;
    int tr = HPK_LR - 2 * Wh - 1;
    tr = tr < tr_cap ? tr : tr_cap;
    tr = tr < HPK_TLIST / g.TC ? tr : HPK_TLIST / g.TC;
    g.TR = tr;
    g.J = (g.TR + g.Dg - mw + g.TC - 1) / g.TC;
    g.tilecap = g.TR * g.TC;
    return g;
}

// One band of a batch as the kernels see it (device memory, written by the host before the launches).
struct HpkBandDesc {
    // ---- what the stencil reads per tile (first 128 bytes)
    HPK_GP(const float) raw;                  // [n][ld] raw counts
    HPK_GP(const double) weight;               // f64[n] or nullptr
    HPK_GP(const double) bal;                  // f64 band or nullptr
    // compact candidate records, one region of `tilecap` records per tile
    //   rec_ent[tile * tilecap + i]                    x | row of the output tile << 8 | min(raw, pk_cap) << 14
    //   rec_S[slot * rec_stride + tile * tilecap + i]  (bS_K, bS_Y) at the resolving step
    //   rec_W[slot * rec_stride + tile * tilecap + i]  resolving step + 1, 0 = unresolved
    HPK_GP(unsigned) rec_ent;
    HPK_GP(double2) rec_S;
    HPK_GP(uint8_t) rec_W;
    int64_t rec_stride;                 // ntiles * tilecap
    int64_t ld;
    HPK_GP(unsigned) tile_cnt;                 // [ntiles] records per tile
    HPK_GP(uint2) units;                       // scoring work list {row block << 8 | column chunk, unit | records of the tile << 8}, appended at tile end
    HPK_GP(unsigned char) small;               // the band's counter block (HPK_OFF_*)
    HPK_GP(uint8_t) gap;                       // [n] preset to 0; set for rows with a non-zero balanced value (gap = !flag)
    HPK_GP(unsigned long long) hist_acc;       // [HPK_HREP][HPK_ACC_STRIDE] zeroed resolve totals the stencil workgroups add to (step s: word s, the candidates: word HPK_MAX_STEPS)
    int32_t n, num;
    int32_t ntiles, chunk;              // chunk = ceil(ntiles / 8): tiles handed to one XCD
    int32_t k0;                         // first index of this band in an XCD's run over the batch (sum of the chunks before)
    int32_t wguess;                     // records only for candidates whose first sufficient width is <= wguess (255: all)
    // ---- prep, expected tables, scoring, cut, publish
    HPK_GP(double) IR;                         // [num] (input, or derived by hpk_ir_*)
    HPK_GP(double) b1;
    HPK_GP(double) b2;
    HPK_GP(double) etab;                       // [nsteps][2][D + 1]
    HPK_GP(double) eedge;                      // [2][W][nsteps][2][D + 1] windows clipped by the first rows / last columns
    HPK_GP(double) psum;                       // hpk_ir_partial: [nparts][num]
    HPK_GP(unsigned) pnan;
    HPK_GP(HpkSurv) surv;                      // [HPK_NREG][cap]
    HPK_GP(HpkSurv) surv2;                     // compacted survivors beyond the inline head
    HPK_GP(unsigned) chunk_used;               // [HPK_NREG * cap / HPK_SCH] filled slots per chunk
    HPK_GP(unsigned) cnt;                      // [HPK_NFAM][HPK_TIGHTEN_MAX] tightening counters / p-value histogram
    HPK_GP(unsigned char) head_host;           // mapped pinned memory the band's head is published to
    int64_t cap;                        // survivor capacity per region (multiple of 256)
    uint64_t zero_bytes;                // bytes of `small` the table kernel zero-fills (multiple of 16)
    uint32_t off_rowlive, off_inl;      // offsets of the row flags / inline survivors inside `small`
    int32_t derive;                     // 1: IR and biases are derived on the device from raw + weight, 2: IR only, 0: given
    int32_t score_wgs;                  // scoring workgroups that take part for this band (the rest of the grid row exits)
    int32_t lean_cj;                    // tiles of column chunks >= lean_cj are built without the f64 plane (hpk_band_class; >= J: none)
    // the band's own tile geometry (hpk_geo_of): the batch's, or - hpk_band_class - the one of the band's own record bound
    int32_t W;                          // halo of the tiles = widest width the search looks at (<= the plan's maxww, see Dg)
    int32_t Dg;                         // last diagonal the tiles must cover for the gap rows: D + (maxww - W)
    int32_t TR, TC;                     // output tile
    int32_t J;                          // column chunks per row block
    int32_t tilecap;                    // records per tile region (TR x TC)
    // which bins have a non-zero weight (NaN counts as zero), one bit per bin from bit HPK_WNZ_LEAD on, zeros around (hpk_band_class
    // writes it, hpk_stencil_lean reads ten columns' bits and a row's per lane): offset inside `small`, 0 = none
    uint32_t off_wnz;
    uint32_t pad_;
};
#define HPK_WNZ_LEAD 64                 // zero bits before bin 0 (tiles reach maxww + 1 bins beyond the matrix' ends)
#define HPK_WNZ_WORDS(n) (((n) + 31) / 32 + 2 + 12)

// Geometry and parameters common to all bands of a batch (kernel argument of the stencil).
struct HpkStencilArgs {
    const HpkDevPlan* plan;
    double risk;                        // box sums below risk x (largest table entry of the window) are redone exactly
    int32_t nbands;
    int32_t mw, D;                      // (the tile geometry is the band's: HpkBandDesc::W ... tilecap)
    int32_t grid;                       // persistent workgroups (multiple of 8, one per CU)
    int32_t single;                     // the plan is a textbook single-pair plan (HpkDevPlan::single_p >= 0)
    int32_t order;                      // tile order within an XCD's run: 0 row-major, 1 column chunks rotated per row block
    int32_t generic;                    // the plan's Reads matrix is not monotone in the width: steps walked in plan order (general plans only)
    int32_t dbg_stop;                   // profiling ablation: 1 stop after the loads, 2 after the SAT, 4 no candidates, 5 search without box sums
    unsigned long long* clk;            // -DHPK_PHASE_CLOCK builds: [grid][waves][8] cycle sums per phase, or nullptr
    int32_t lean_max;                   // lean tiles (hpk_stencil_lean): most candidates summed cell by cell before the tile is handed to hpk_stencil_s; 0: no lean tiles
    int32_t pad_;
    unsigned* redoq;                    // {tiles queued, tiles taken, -, -, (band, row block << 8 | column chunk) ...}: the tiles hpk_stencil_lean gave up, or nullptr
};

struct HpkScoreArgs {
    const HpkDevPlan* plan;             // (tile geometry: the band's, HpkBandDesc)
    const double* bounds;               // [HPK_NB] chunk upper bounds
    const double* ptab;                 // Poisson survival table
    const int32_t* ptab_off;            // [HPK_NB_TAB + 2]
    const double* sfe;                  // stirlerr(0..31)
    double sig;
    int32_t mw, D;
    int32_t hbins;                      // bins per family of the p-value histogram (HpkBandDesc::cnt), 0 = none (hpk_thr_hist / counting rounds do the cut)
    int32_t nsets_half;                 // (pw, ww) pairs of the call: the launcher sizes the histogram's LDS with it
    int32_t gridx;                      // workgroups per band
    const uint8_t* kmin;                // [HPK_NFAM] or nullptr: survivor records only for p-values in histogram bin >= kmin[family]
    const int32_t* kcrit;               // hiccups: [HPK_NB_TAB + 2] or nullptr: per chunk of the Poisson table the smallest count with p <= sig (hpk_kcrit);
                                        // bhfdr: [HPK_KCL_N] or nullptr: the same per cell of a grid over lambda (hpk_kcrit_lam)
    unsigned long long* clk;            // -DHPK_PHASE_CLOCK builds: [8] accumulators of hpk_score (prologue, loop, epilogue ticks; waves; items; longest wave), else nullptr
};

// bhfdr's critical counts: lambda in [2^HPK_KCL_E0, 2^(HPK_KCL_E0 + HPK_KCL_N / 16)) on a grid of 16 cells per octave - cell =
// the top 16 bits of the double (sign 0, exponent, four mantissa bits) minus HPK_KCL_G0; a cell's entry belongs to its lower edge
#define HPK_KCL_N 512
#define HPK_KCL_E0 (-16)
#define HPK_KCL_G0 ((1023 + HPK_KCL_E0) << 4)

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

// hpk_stencil_s walks the whole batch in one launch (bands within its addressing limits, a halo of at least 4)
bool hpk_stencil_s_applies(const HpkGeo& g, int64_t max_ld, int32_t max_n);
// hpk_stencil_lean over the lean column chunks (a.lean_max > 0), then hpk_stencil_s over the others and the tiles the first gave up
void hpk_launch_stencil_batch(const HpkStencilArgs& a, const HpkBandDesc* d_bands, bool balf64, int cus, hipStream_t st);
void hpk_launch_dense(const HpkDenseArgs& a, hipStream_t st);
void hpk_launch_probe(const HpkDenseArgs& a, const int32_t* rows, const int32_t* cols, int64_t count, double* out, hipStream_t st);
void hpk_launch_freeze_tot(const HpkDevPlan* plan, const HpkBandDesc* d_bands, int nbands, hipStream_t st);
// IR / biases of the bands with `derive` set (scripts/pyHICCUPS:149-166); max_n / max_num: largest band of the batch
void hpk_launch_prep(const HpkBandDesc* d_bands, int nbands, int max_n, int max_num, int mw, hipStream_t st);
// expected tables + zero-fill of the counter blocks; max_zero: largest zero_bytes of the batch
void hpk_launch_etab(const HpkDevPlan* plan, const HpkBandDesc* d_bands, int nbands, int nsteps, int D, int W, size_t max_zero, hipStream_t st);
void hpk_launch_gap(const float* raw, const double* bal, const double* weight, int32_t n, int32_t num,
                    int64_t ld, int32_t mw, uint8_t* gap, hipStream_t st);
int  hpk_score_grid(bool bhfdr, int npairs, int hbins, int cus);            // resident workgroups of the scoring kernel
void hpk_launch_score(const HpkScoreArgs& a, const HpkBandDesc* d_bands, int nbands, bool bhfdr, hipStream_t st);
// Benjamini-Hochberg cut tightening on the survivor lists: thr[f] <- sig * #{p <= thr[f]} / m[f], `rounds` times,
// then compaction of the records with p <= thr[f] (count in the band's HPK_OFF_NOUT).
int  hpk_thr_hist_bins(int nsets);       // bins per family of the one-pass tightening (rounds < 0)
int  hpk_score_hist_bins(int nsets, bool bhfdr);     // bins per family of the histogram hpk_score keeps (rounds <= -100): 8 | 4, bhfdr 64 fine ones
void hpk_launch_tighten(const HpkBandDesc* d_bands, int nbands, double sig, int rounds, int nsets, const uint8_t* kmin, hipStream_t st);
// result heads -> mapped pinned host memory: full = the whole head (no scoring ran), otherwise the stretches a chromosome fills
void hpk_launch_publish(const HpkBandDesc* d_bands, int nbands, int nsets, bool full, size_t max_head_bytes, hipStream_t st);
void hpk_launch_ptab(const double* bounds, const int32_t* off, const double* sfe, double* ptab, int32_t total,
                     hipStream_t st);
void hpk_launch_kcrit(const double* ptab, const int32_t* off, double sig, int32_t* kcrit, hipStream_t st);
void hpk_launch_kcrit_lam(const double* sfe, double sig, int32_t* kcrit, hipStream_t st);
// Record bound per chromosome by depth class (hpk_band_class): class = quarter octave of the band's mean count per pixel (every
// 64th row sampled), bound = table[class] + margin (table: the width chromosomes of that class froze at, -1 unknown) clamped to
// [wmin, wg_all]; written into the descriptor's wguess and into the counter block (HPK_OFF_BCLASS).  Runs before the stencil.
#define HPK_NCLASS 64
// ... and, same kernel, same sampled rows: from which column chunk on a band's tiles are expected to hold (next to) no candidate
// that resolves within the band's bound - the mean Reads of the chunk's nearest pixels is at most lean_frac x min_local_reads -
// and are built without their f64 plane (HpkBandDesc::lean_cj, hpk_stencil_s).
struct HpkClassArgs {
    const signed char* table;           // [HPK_NCLASS] or nullptr: no depth classes, every band keeps the batch's bound
    int32_t mw, D, wg_all, margin, wmin;
    int32_t lean;                       // 0: no lean tiles (lean_cj = J)
    int32_t halo;                       // 1: every band's tiles laid out for its own bound's halo (hpk_geo_of); 0: the batch's geometry stays
    int32_t planW, tr_cap;              // hpk_geo_of's inputs
    int32_t p0, minr;                   // Reads = lower-left rings p0 + 1 .. w against min_local_reads
    float lean_frac;
    int32_t lean_share;                 // least share (percent) of a band's column chunks that must be lean for the band to have lean tiles at all
    int32_t own;                        // 1: every band's bound is the wguess its descriptor carries (the second pass of spec_halo = 2: its frozen width), no table
};
void hpk_launch_band_class(HpkBandDesc* d_bands, int nbands, const HpkClassArgs& a, hipStream_t st);
void hpk_launch_poisson_sf(const double* k, const double* lam, const double* sfe, double* out, int64_t count,
                           hipStream_t st);
void hpk_launch_brute(const HpkBruteArgs& a, hipStream_t st);
// pixels of one chromosome -> zero-filled band (info[0] += stored pixels, info[1] = 1 on a bin outside [0, n))
void hpk_launch_coo_scatter(const int64_t* bin1, const int64_t* bin2, const void* count, int count_f64, int64_t nnz, int n, int num,
                            int64_t ld, float* raw, unsigned long long* info, hipStream_t st);
