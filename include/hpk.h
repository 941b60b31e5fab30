/*
 * hpk.h - C ABI of libhpk.so: the MI355X (gfx950) HiCCUPS / BH-FDR scoring core.
 *
 * This is the drop-in boundary for the hot path of XiaoTaoWang/HiCPeaks 0.3.9.  The reference has no
 * FFI of its own (it is pure Python); the seam it offers is two functions,
 *
 *     hicpeaks/callers.py:44-46   hiccups(M, cM, B1, B2, IR, chromLen, Diags, cDiags, num, chrom, pw, ww, ...)
 *     hicpeaks/callers.py:364-365 bhfdr  (M, cM, B1, B2, IR, chromLen, Diags, cDiags, num, chrom, pw, ww, ...)
 *
 * called once per chromosome from scripts/pyHICCUPS:170-173 and scripts/pyBHFDR:143-144.  Everything those
 * functions do between "inputs arrive" and "q <= sig pixels are known" (callers.py:50-287 and 367-553:
 * zero-padded band, donut / lower-left box sums, adaptive widening, corrected expected, lambda-chunk
 * Poisson p-values, Benjamini-Hochberg q-values) happens behind hpk_score_band(); the gap filter,
 * donut/LL combination, clustering and text output (callers.py:289-362, 555-590, scripts/pyHICCUPS:200-210)
 * stay in the host language above this ABI (hicpeaks_amd/callers.py).
 *
 * Conventions: plain C, no exceptions across the boundary.  Every entry point returns HPK_OK or a
 * negative hpk_status; hpk_last_error() gives the message.  The caller owns all inputs (never written),
 * the library owns everything reachable from hpk_result until hpk_result_free().  One hpk_ctx per device
 * and per host thread; calls block until the result is complete (ctypes releases the GIL around them).
 * There is no CPU fallback: hpk_create(device >= 0) fails when no gfx950 device is usable (device -1 *asks* for back-end #0,
 * the same path on host threads: see hpk_create).
 *
 * Band layout (both inputs and dense outputs): element (r, k) of a [n][ld] row-major array is matrix
 * pixel (row r, column r + k); entries with r + k >= n are ignored.
 */
#ifndef HPK_H
#define HPK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPK_ABI_VERSION 3
#define HPK_MAX_BATCH   256    /* chromosomes per hpk_submit_batch */
#define HPK_MAX_PAIRS   8      /* (pw, ww) pairs per call */
#define HPK_MAX_W       20     /* largest supported maxww (reference keyword default, callers.py:45) */
#define HPK_MAX_STEPS   64     /* widening steps in one plan (lane-indexed histogram) */

typedef enum {
    HPK_OK = 0,
    HPK_ERR_INVALID = -1,      /* bad argument: null pointers, shapes / sizes / option values out of range, inputs that
                                  do not go together.  (Negative balancing weights / balanced values are NOT an error:
                                  they are kept with their sign, as callers.py:78 keeps them -
                                  tests/test_gpu_parity.py::test_negative_balanced_values_are_kept.) */
    HPK_ERR_HIP = -2,          /* HIP runtime error (message has the call and hipError name) */
    HPK_ERR_NO_DEVICE = -3,    /* no gfx950 device / extension unusable: there is no CPU path */
    HPK_ERR_EMPTY_STEP = -4,   /* a widening step was entered with no unresolved candidate for its peak
                                  width - the reference raises here (callers.py:203-208, 487-492) */
    HPK_ERR_PLAN = -5,         /* (pw, ww, maxww) give a negative ring multiplicity; not representable */
    HPK_ERR_NOMEM = -6,
    HPK_ERR_BUSY = -7          /* every lane of the context holds a chromosome in flight */
} hpk_status;

typedef enum {
    HPK_MODE_HICCUPS = 0,      /* callers.py:44-362: donut (K) + lower-left (Y), lambda chunks */
    HPK_MODE_BHFDR = 1         /* callers.py:364-590: donut only, per-pixel Poisson, one BH */
} hpk_mode;

enum {
    HPK_FLAG_DENSE_E = 1,      /* copy the dense per-slot local expected (E_K, E_Y) and resolving width back */
    HPK_FLAG_DENSE_SUMS = 2,   /* debug: also the four raw sums (bS_K, bE_K, bS_Y, bE_Y) at every candidate */
    HPK_FLAG_NO_SCORE = 4,     /* stop after the stencil (bench: time the donut kernel alone) */
    HPK_FLAG_PHASE_TIMING = 8, /* also time upload / freeze / scoring / tightening (five more events on the compute
                                  stream, ~3 us of idle each); ms_d2h, ms_host_bh, ms_total are always filled */
    HPK_FLAG_NO_STENCIL_TIMING = 16  /* leave out the two events around the stencil launch (~6 us of idle per call):
                                  ms_stencil = 0.  Production callers set it; bench.py brackets every fourth launch */
};

/* hiccups(): pw/ww lists, maxww, sig, maxapart, res, min_local_reads (callers.py:44-46);
 * bhfdr(): one pair, min_local_reads is ignored (hard-coded 16, callers.py:490). */
typedef struct {
    int32_t mode;
    int32_t npairs;
    int32_t pw[HPK_MAX_PAIRS];
    int32_t ww[HPK_MAX_PAIRS];
    int32_t maxww;
    int32_t min_local_reads;   /* the stencil compares Reads on counts capped at max(1023, this) (exact for the threshold) and
                                  sums a box of them in 21 bits: at most (2^21 - 1) / (2 maxww + 1)^2, i.e. 4755 at maxww = 10,
                                  1247 at maxww = 20;
                                  larger values return HPK_ERR_INVALID.  The reference's defaults are 25 / 16 */
    int64_t maxapart;
    int64_t res;
    double  sig;
    int32_t flags;
    int32_t reserved;
} hpk_params;

/* One intra-chromosomal band.  Replaces (M, cM, B1, B2, IR, chromLen, Diags, cDiags, num) of callers.py:44.
 * Either `balanced` (f64 band, NaN already zeroed = cDiags, scripts/pyHICCUPS:149-158) or `weight`
 * (f64[n], NaN/0 = masked bin; balanced is then formed on chip as (raw * w[r]) * w[c]) must be given. */
typedef struct {
    int32_t n;                 /* chromLen */
    int32_t num;               /* stored diagonals: maxapart/res + maxww + 1 (scripts/pyHICCUPS:146) */
    int64_t ld;                /* row pitch of raw / balanced in elements, >= num */
    const float*  raw;         /* [n][ld] raw counts (exact below 2^24) = Diags */
    const double* balanced;    /* [n][ld] or NULL */
    const double* weight;      /* [n] or NULL */
    const double* IR;          /* [num] 1-D expected per diagonal, 0 below min(ww) (scripts/pyHICCUPS:150-156) */
    const double* bias1;       /* [n] B1 (callers.py:249) */
    const double* bias2;       /* [n] B2 */
                               /* IR = bias1 = bias2 = NULL (needs `weight`): the library derives them on the device the
                                  way scripts/pyHICCUPS:149-166 does - IR[d] = mean of the balanced diagonal with stored
                                  pixels of masked bins left out, biases = 1 / weight (0 where the weight is 0 / NaN).
                                  IR = NULL with bias1 / bias2 given: only IR is derived (a divisive weight column: the
                                  caller hands over weight = 1 / column and the biases the reference forms from the column) */
    int32_t on_device;         /* non-zero: all pointers above are device pointers on the ctx's device */
    int32_t reserved;
} hpk_band;

/* One scored (pair, filter) set = one pass of callers.py:242-287. */
typedef struct {
    int32_t pair;              /* index into params.pw/ww */
    int32_t fl;                /* 0 = 'K' donut, 1 = 'Y' lower-left */
    int64_t nvalid;            /* pixels with E > 0 (callers.py:257) */
    int32_t numbin;            /* lambda chunks (callers.py:30); 0 for bhfdr */
    int32_t reserved;
    double  emax;
    int64_t begin, end;        /* slice of the survivor arrays: pixels with q <= sig (callers.py:279) */
    const uint32_t* chunk_tests;   /* [129] Poisson models per lambda chunk i = 1..numbin (family sizes of the BH step,
                                      callers.py:266); entry 0 unused; bhfdr: entry 1 = all of them */
    const uint32_t* chunk_below;   /* [129] of those, p <= sig */
} hpk_set;

typedef struct {
    /* widening log (callers.py:203-232) */
    int32_t nsteps;
    int32_t step_pi[HPK_MAX_STEPS];
    int32_t step_wi[HPK_MAX_STEPS];
    int32_t step_executed[HPK_MAX_STEPS];
    int64_t step_resolved[HPK_MAX_STEPS];   /* candidates resolved at the step; 0 for steps not executed (beyond frozen_w)
                                               unless the call asked for every candidate (dense outputs, HPK_FLAG_NO_SCORE) */
    int32_t frozen_w;
    int32_t nslots;            /* distinct peak widths */
    int32_t slot_pi[HPK_MAX_PAIRS];
    int64_t ncand;             /* nonzero pixels with min(ww) <= d <= maxapart/res (callers.py:101-104) */

    int32_t nsets;
    hpk_set sets[2 * HPK_MAX_PAIRS];

    /* survivors, row-major within each set */
    int64_t nsig;
    const int32_t* x;
    const int32_t* y;
    const double*  O;          /* raw count */
    const double*  bal;        /* balanced value at the pixel (cM[x, y]) */
    const double*  E;          /* corrected expected */
    const double*  p;
    const double*  q;
    const uint8_t* other_zero; /* K sets: 1 if the lower-left corrected expected is 0 there (callers.py:330) */

    const uint8_t* gap;        /* [n] 1 = row of the balanced band sums to 0 (callers.py:238) */

    /* optional dense outputs (HPK_FLAG_DENSE_*): [nslots][n][dense_ld] */
    int64_t dense_ld;
    const double*  dense_E;    /* [..][2] = (E_K, E_Y), 0 where unresolved */
    const uint8_t* dense_w;    /* resolving donut width, 0 = never resolved */
    const double*  dense_sums; /* [..][4] = (bS_K, bE_K, bS_Y, bE_Y) */

    /* timing, milliseconds (HIP events on the ctx stream; wall for host parts) */
    float ms_h2d, ms_stencil, ms_freeze, ms_score, ms_tighten, ms_gap, ms_d2h, ms_host_bh, ms_total;
    int32_t stencil_kernel;    /* stencil kernel that ran: always 2: the one stencil generation there is */
    int32_t record_bound;      /* the stencil wrote records for candidates resolved up to this width (255: all of them;
                                  HPK_FLAG_DENSE_*, HPK_FLAG_NO_SCORE); see hpk_submit_band */
    int32_t redone;            /* bit 0: the widening froze beyond the bound taken from the previous chromosome and the
                                  chromosome was computed once more with every resolved candidate; bit 1: the
                                  Benjamini-Hochberg cut of a family lay above the bound the survivor records were written to
                                  (hpk_set_option spec_surv) and scoring + cut ran once more with a record for every p <= sig */
    int64_t nsurv_sig;         /* pixels with p <= sig (before the BH cut is tightened on the device) */
    int64_t nsurv_cut;         /* of those, how many were copied back for the final Benjamini-Hochberg step */
    int64_t stencil_tiles;
    int64_t band_px;           /* pixels with min(ww) <= d <= maxapart/res inside the matrix */
    int32_t batch_bands;       /* chromosomes that shared this one's kernel launches (hpk_submit_batch; 1 otherwise).  The
                                  ms_* kernel times of a batch are split over its chromosomes by band pixels: their sum over
                                  the batch is the launch's duration */
    int32_t halo_w;            /* halo of the stencil's tiles = widest width its search looked at: maxww, or - second-generation
                                  kernel, hpk_set_option spec_halo = 1 (default) - the record bound.  Box sums are differences of
                                  table entries summed from the tile's corner, so two runs of one chromosome under different halos
                                  agree to ~1e-13 relative in E / p / q (coordinates, counts and the widening log exactly) */
    /* ABI v3 */
    int32_t lean_tiles;        /* tiles built without their f64 plane (far from the diagonal next to no candidate resolves within the
                                  record bound: hpk_set_option "lean"); the few candidates of such a tile that do count get their
                                  sums cell by cell from the band */
    int32_t lean_redone;       /* of those, tiles that met more such candidates than "lean_max" and were computed once more in full */
    int64_t lean_explicit;     /* candidates whose sums were formed cell by cell in a lean tile */
} hpk_result;

typedef struct hpk_ctx hpk_ctx;

/* device >= 0: HIP device ordinal.  Fails with HPK_ERR_NO_DEVICE when it is not a gfx950 GPU - there is no fallback.
 * device == -1: back-end #0, the same path as plain C++ on host threads (hpk_cpu.cpp; option "cpu_threads",
 * HPK_CPU_THREADS; default: every core the process may run on).  It exists to be measured beside the GPU path (bench.py:
 * cpu_baseline.kind = "native") and to run the parity ladder where there is no GPU; a context is a CPU context only when the
 * caller asks for one.  It takes host bands (on_device = 0), single chromosomes or batches (scored one after the other when the
 * job is collected), has no history (every chromosome as if it were the first: record_bound 255, halo_w = maxww, stencil_kernel 0)
 * and none of the device path's debug entry points (dense outputs, probes, hpk_devband_create: HPK_ERR_INVALID). */
int  hpk_create(int device, hpk_ctx** out);
void hpk_destroy(hpk_ctx* ctx);
const char* hpk_last_error(const hpk_ctx* ctx);      /* ctx may be NULL: last create() failure */
int  hpk_abi_version(void);

/* The whole path for one chromosome.  *out is allocated by the library. */
int  hpk_score_band(hpk_ctx* ctx, const hpk_band* band, const hpk_params* params, hpk_result** out);
void hpk_result_free(hpk_result* res);

/* The same path split in two, so that the caller's loop over chromosomes (scripts/pyHICCUPS:192-198 maps worker()
 * over the chromosomes one after the other) can keep the GPU busy: hpk_submit_band enqueues the uploads, every
 * kernel and the download of the result head on one of the context's hpk_pipeline_depth() lanes (each with its own
 * HIP stream and workspaces) and returns at once; hpk_collect waits for that lane, runs the Benjamini-Hochberg step
 * on the host and hands out the result.  With one chromosome submitted ahead, the host half of chromosome i and the
 * upload of chromosome i + 1 overlap the kernels.  hpk_score_band == submit + collect.
 * Host input arrays (band->on_device == 0) must stay valid until the job is collected.  hpk_collect always
 * consumes the job, also on error.  HPK_ERR_BUSY: no free lane.
 * A context remembers the width at which the widening froze (callers.py:223-229) in the chromosome it collected last
 * for the same parameters, and the next submission's stencil writes candidate records up to that width only - wider
 * ones are dropped by the scoring rules anyway (callers.py:133-134).  With option spec_halo (default 1) the launch also
 * stops the search for the first sufficient width at the bound and lays its tiles out for the bound's halo instead of
 * maxww's (hpk_result::halo_w: larger output tiles, fewer of them).  A chromosome that freezes later is noticed at
 * collection and computed once more in full, under the plan's own tiles (hpk_result::redone).  The pixels reported, their
 * counts and the widening log never depend on the bound; E / p / q are bit-identical under spec_halo = 0 and equal to
 * rounding (~1e-13 relative: the box sums are differences of table entries summed from other tile corners) otherwise. */
typedef struct hpk_job hpk_job;
int  hpk_pipeline_depth(void);
int  hpk_submit_band(hpk_ctx* ctx, const hpk_band* band, const hpk_params* params, hpk_job** job);
int  hpk_collect(hpk_ctx* ctx, hpk_job* job, hpk_result** out);

/* The loop over chromosomes itself (scripts/pyHICCUPS:192-198 maps worker() over them; every statistic of the path is
 * per chromosome) as ONE set of kernel launches: `nbands` chromosomes scored with the same parameters - the stencil walks
 * the tiles of all of them in one persistent launch, the expected tables, the scoring, the Benjamini-Hochberg cut and the
 * copy-back are one launch each over the batch.  Results are the ones hpk_score_band gives chromosome by chromosome, bit
 * for bit (under the same record bound, see hpk_submit_band).  1 <= nbands <= HPK_MAX_BATCH; all bands carry the same kind of input (all `balanced` or all `weight`); the
 * HPK_FLAG_DENSE_* outputs are for single chromosomes.  A batch occupies one lane (see hpk_submit_band).
 * hpk_collect_batch always consumes the job.  outs[i] receives chromosome i's result (NULL where status[i] != HPK_OK, e.g.
 * HPK_ERR_EMPTY_STEP for a chromosome on which the reference raises); errmsg, if not NULL, is [nbands][errmsg_len] chars
 * and receives the message of every failed chromosome.  The return value is HPK_OK when the batch itself ran. */
int  hpk_submit_batch(hpk_ctx* ctx, const hpk_band* bands, int32_t nbands, const hpk_params* params, hpk_job** job);
int  hpk_collect_batch(hpk_ctx* ctx, hpk_job* job, hpk_result** outs, int32_t* status, char* errmsg, int32_t errmsg_len);

/* Tuning and test switches of a context (the HPK_<NAME> environment variables are read once, by hpk_create; nothing
 * between hpk_submit_* and its return looks at the environment).  Names: "rounds" (-2: the scoring kernel keeps the
 * p-value histogram the cut is derived from [default], -1: a histogram pass of its own, 0..4: exact counting rounds),
 * "surv_cap" (survivor slots per region, 0 = sized from the band; tests force the overflow rerun with it), "spec" (0: no
 * record bound from earlier chromosomes), "spec_margin" (widths added to the bound), "spec_force" (>= 0: this bound;
 * tests), "spec_class" (1 [default]: under the batch's bound every chromosome writes records only up to the width chromosomes of
 * its depth class - quarter octaves of the mean count per band pixel, sorted on the device - froze at last; verified like the
 * batch's bound, hpk_result::record_bound is the chromosome's own; 0: one bound per batch), "class_force" (tests), "spec_surv" (0: a survivor record for every p <= sig; 1 [default]: only up to the histogram bin the families' cuts fell
 * into in the chromosomes before, minus "spec_surv_margin" bins - verified, hpk_result::redone bit 1), "spec_surv_force" (tests),
 * "host_threads" (threads of a batch's host half), "spec_halo" (1 [default]: a chromosome's tiles are laid out for the halo of the bound it inherits; 0: tiles always
 * under maxww's halo - runs of one chromosome are then bit-identical whatever the bound; 2: as 1, and a chromosome whose halo was not the one of the
 * width its OWN widening froze at is computed once more under that one, lean tiles included (hpk_result::redone bit 0): E / p / q are a function of the
 * chromosome alone - what the command lines run under), "risk_log2" (exact-fallback threshold 2^-x), "tile_order", "gap_kernel" (1: gap rows by the row kernel),
 * "score_div" (tiles per scoring workgroup of a batch), "surv_div" (survivor capacity of a chromosome: band pixels x sets / surv_div records with
 * p <= sig, default 6; a chromosome with more is scored once more with room), "dbg_stop" (profiling ablation), "grid_cap" (tests: at most this many stencil workgroups, 0 = one
 * per CU - every workgroup then crosses every band of a batch and walks long runs of tiles), "lean" (1 [default]: tiles of the column chunks
 * whose mean Reads stays below "lean_frac_pct" % of min_local_reads - sampled per chromosome on the device - are built without their
 * f64 plane; up to "lean_max" candidates of such a tile that resolve within the bound get their sums cell by cell, a tile with more
 * is computed once more in full; weight input only, off under spec_halo = 0), "kcrit" (1 [default]: the scoring kernel forms a
 * p-value only where the pixel's count reaches the critical count - the smallest one with p <= sig - of its lambda chunk (hiccups)
 * or of its lambda's cell on a grid of 16 per octave (bhfdr); 0: every p-value is formed; same results), "reset_hints" (forget the bounds
 * learnt from the chromosomes collected so far).  Returns HPK_ERR_INVALID for an
 * unknown name or a value out of range. */
int  hpk_set_option(hpk_ctx* ctx, const char* name, int64_t value);

/* Host-only helpers (no device needed): the widening plan of callers.py:15-23 + 132-201 as ring
 * multiplicities.  mult is [HPK_MAX_STEPS][HPK_MAX_W + 1]; returns the number of steps or a status. */
int  hpk_plan_rings(const hpk_params* params, int32_t* step_pi, int32_t* step_wi,
                    int32_t* mult_K, int32_t* mult_reads);
/* lambda-chunk boundaries 2^((i-1)/3), i = 1..count (callers.py:33-37) as the library computes them. */
int  hpk_chunk_bounds(double* bounds, int32_t count);
/* Replace the boundaries (count must be 128) - the Python layer hands over numpy's own np.power(2, (i-1)/3.)
 * values so that chunk membership is decided on the very numbers the reference compares with. */
int  hpk_set_chunk_bounds(hpk_ctx* ctx, const double* bounds, int32_t count);

/* Device helpers used by tests and bench. */
int  hpk_device_info(hpk_ctx* ctx, char* name, int32_t name_len, int32_t* cus, int64_t* hbm_bytes);
/* Poisson survival 1 - cdf(k; lam) exactly as the scoring kernel evaluates it (callers.py:268-270, 536-540). */
int  hpk_poisson_sf(hpk_ctx* ctx, const double* k, const double* lam, double* out, int64_t count);
/* Independent brute-force check kernel: explicit (2w+1)^2 window sums at `count` sampled pixels
 * (no summed-area table); out is [count][5] = (bS_K, bE_K, bS_Y, bE_Y, Reads) for one step. */
int  hpk_bruteforce_sums(hpk_ctx* ctx, const hpk_band* band, const hpk_params* params, int32_t step,
                         const int32_t* rows, const int32_t* cols, int64_t count, double* out);

/* Device-side band builder (row F2, the GPU counterpart of hpk_band_from_coo): the pixels of one chromosome travel as they
 * are - 20 bytes per stored pixel instead of 4 bytes per band cell, a fifth to a twentieth of the dense band - and are
 * scatter-added into a zero-filled band raw[n][ld] (ld = num rounded up to 64) in device memory owned by the library;
 * `weight` (f64[n], host) and, if not NULL, `bias` (f64[n], host: see hpk_band, IR-only derivation) are uploaded with
 * them.  *band is filled in for hpk_submit_band / hpk_submit_batch (on_device = 1, IR = NULL: derived on the device).  The
 * memory stays valid until hpk_devband_free (after the job that used it was collected) or hpk_destroy, which frees the bands
 * still alive (a hpk_devband must not be freed after its context).  Counts are integers below
 * 2^24: the f32 sums are exact whatever the order of the adds.  Returns the number of stored pixels, or a negative
 * status (HPK_ERR_INVALID: a bin outside [0, n)). */
typedef struct hpk_devband hpk_devband;
int64_t hpk_devband_create(hpk_ctx* ctx, const int64_t* bin1, const int64_t* bin2, const void* count, int32_t count_is_f64,
                           int64_t nnz, int32_t n, int32_t num, const double* weight, const double* bias,
                           hpk_devband** out, hpk_band* band);
void hpk_devband_free(hpk_ctx* ctx, hpk_devband* b);

/* The production kernels' own sums at `count` sampled pixels without the dense debug outputs (which need
 * [nslots][n][D+1] arrays - gigabytes at 5 kb / 1 kb resolution): runs the stencil, looks the pixels up in the
 * candidate records; out is [count][nslots][5] = (bS_K, bE_K, bS_Y, bE_Y, resolving width); width 0 = never
 * resolved, -1 = not a candidate (zero count or outside min(ww) <= d <= maxapart/res). */
int  hpk_probe_sums(hpk_ctx* ctx, const hpk_band* band, const hpk_params* params, const int32_t* rows,
                    const int32_t* cols, int64_t count, double* out);

/* Host-only band builder (no device needed): counterpart of `Diags = [H.diagonal(i) for i in range(num)]`
 * (scripts/pyHICCUPS:147), which scans all stored pixels once per diagonal - O(num * nnz).  Here one O(nnz) pass
 * scatter-adds the pixels (bin1[t], bin2[t], count[t]) of one chromosome (bins relative to its first bin; every pixel
 * listed once, in either orientation - cooler's pixel table lists the upper triangle) into the dense
 * upper band raw[n][ld], raw[r][k] = count of (r, r + k), k < num.  `raw` must be zero-filled by the caller.
 * count_f64 != 0: counts are double, otherwise int32.  Returns the number of pixels stored, or a negative status. */
int64_t hpk_band_from_coo(const int64_t* bin1, const int64_t* bin2, const void* count, int32_t count_f64, int64_t nnz,
                          int32_t n, int32_t num, int64_t ld, float* raw);

/* Host-only helper of the cooler reader (no device needed; counterpart of the chunk filter pipeline behind
 * `Lib.matrix(balance=False).fetch(key)`, scripts/pyHICCUPS:142): `nchunks` consecutive chunks of a one-dimensional HDF5 data set,
 * as stored (deflate, behind the byte-shuffle filter if `shuffle`; src[i] / src_len[i]: chunk first_chunk + i, which holds the
 * elements [(first_chunk + i) * chunk_elems, ...)), are inflated, un-shuffled and widened on `threads` threads (0: all cores)
 * into out[0 .. stop - start) = elements [start, stop) minus `bias` (bin ids relative to a chromosome's first bin) as int64
 * (out_f64 = 0), f64 (1) or int32 (2: signed columns of up to four bytes, unsigned ones of up to two).  elem_size 1 / 2 / 4 / 8; kind 0 signed,
 * 1 unsigned integers, 2 floats; little-endian.  HPK_ERR_INVALID: a chunk that does not inflate to whole elements. */
int  hpk_decode_chunks(const void* const* src, const uint64_t* src_len, int64_t nchunks, int64_t first_chunk, int64_t chunk_elems,
                       int32_t elem_size, int32_t kind, int32_t shuffle, int64_t start, int64_t stop, void* out, int32_t out_f64,
                       int64_t bias, int32_t threads);
/* The same with the chunks read by the decoding threads themselves: pread(fd, ..., file_off[i]) of src_len[i] bytes each (the
 * chunks' addresses in the file, as H5Dget_chunk_info_by_coord reports them) - the HDF5 library, which is not thread-safe, is
 * then only asked where the chunks are. */
int  hpk_decode_chunks_fd(int32_t fd, const uint64_t* file_off, const uint64_t* src_len, int64_t nchunks, int64_t first_chunk,
                          int64_t chunk_elems, int32_t elem_size, int32_t kind, int32_t shuffle, int64_t start, int64_t stop,
                          void* out, int32_t out_f64, int64_t bias, int32_t threads);

/* The pixels of a chromosome's rows that belong to its intra-chromosomal map (counterpart of the selection inside
 * `Lib.matrix(balance=False, as_pixels=True, join=False).fetch(key)`, scripts/pyHICCUPS:142): of `n` pixels (bin1, bin2, count;
 * bins relative to the chromosome's first bin, count_size 4 or 8 bytes per count) those with 0 <= bin2 < nbins and, if `square`
 * (both triangles stored), bin2 >= bin1, in order, into out1 / out2 / outc (no overlap with the input) on `threads` threads.
 * Returns the number kept; if that is n, or if all three outputs are null (count only), nothing is written.  Host only. */
int64_t hpk_compact_pixels(const int64_t* bin1, const int64_t* bin2, const void* count, int32_t count_size, int64_t n, int64_t nbins,
                           int32_t square, int64_t* out1, int64_t* out2, void* outc, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* HPK_H */
