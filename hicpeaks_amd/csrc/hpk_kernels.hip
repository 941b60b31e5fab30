// HIP kernels for gfx950 (MI355X / CDNA4).  Wave = 64 lanes; one stencil workgroup owns the LDS of a CU
// (158 of 160 KiB).
//
//   hpk_ir_partial / hpk_ir_final   1-D expected IR[d] and biases from raw + weights   scripts/pyHICCUPS:149-166
//   hpk_etab_edge   local-expected tables (interior + clipped windows), zero-fill duty     callers.py:66-72, 175-198
//   hpk_stencil_s   donut (K) + lower-left (Y) local sums, adaptive widening, gap rows,    callers.py:132-232, 440-513,
//                   scoring work list                                                      238
//   hpk_freeze      frozen_w / break decision from the resolve histogram                   callers.py:208-229, 505-511
//   hpk_score       corrected expected -> lambda chunk -> Poisson p -> survivors           callers.py:238-271, 517-540
//   hpk_thr_count / hpk_thr_compact   Benjamini-Hochberg cut on the survivor list          callers.py:273-279, 545-553
//   hpk_publish     result head -> mapped pinned host memory
//   hpk_ptab        Poisson survival table for the chunk bounds                            callers.py:268-270
//   hpk_gap         gap rows of callers that hand over more diagonals than the band          callers.py:238
//   (hpk_brute, hpk_dense, hpk_probe, hpk_poisson_sf_k: hpk_testkernels.hip)
//
// Stencil design.  The tile is built in true matrix coordinates (r, c): an output tile of TR x TC pixels
// plus a halo of maxww (+1 row/column for the prefix origin) is read from band storage - rows are
// contiguous in c - and turned into a summed-area table (SAT) of 12-byte cells {f64 balanced, u32 capped raw
// count | valid flag << 21} in LDS, 64 rows x 160 columns.  A table row belongs to one DPP row of 16 lanes, ten
// consecutive cells per lane, four table rows per wave: the prefix along a row is nine adds inside the lane and a
// four-step scan over 16 lanes, written straight to LDS; the prefix down the columns is a second pass through LDS
// (a thread per column and chunk of rows).  With the SAT every quadrant box of the (p, w) window is
// four cell reads, independent of w.  u32 sums wrap but their differences are exact; the valid count
// tells an all-zero balanced box (exact 0, as the reference's CSR adds give) from floating-point residue
// of the f64 SAT.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <stdlib.h>
#include <type_traits>
#define HPK_KERNEL_TU
#include "hpk_kernels.h"

namespace {
#include "hpk_device.h"

constexpr int LC = HPK_LC;
constexpr int LR = HPK_LR;

// The SAT is kept as two planes so that every access pattern of the evaluation phase is bank-conflict free
// (lanes walk along x): f64 balanced (8-byte stride, ds_read_b64) and one u32 plane (ds_read_b32) that carries two
// exact integer sums side by side:
//   bits  0..20  raw count capped at HPK_PK_CAP.  Only "Reads >= min_local_reads" is ever asked of it: with
//                min_local_reads <= HPK_PK_CAP a capped cell decides the comparison the same way the true count
//                does, and a box of (2 maxww + 1)^2 <= 1681 cells stays below 2^21;
//   bits 21..31  number of cells with a non-zero balanced value (a box holds < 2^11 of them).
// Sums wrap mod 2^32, box differences are exact in both fields.  Sizes: 80 KiB + 40 KiB; the rest of the LDS holds
// the per-wave candidate lists.
constexpr unsigned PK_SHIFT = HPK_PK_BITS, PK_MASK = (1u << PK_SHIFT) - 1u;
static_assert((2 * HPK_MAX_W + 1) * (2 * HPK_MAX_W + 1) * HPK_PK_CAP < (1u << PK_SHIFT), "capped raw box sum must fit its field");
static_assert((2 * HPK_MAX_W + 1) * (2 * HPK_MAX_W + 1) < (1u << (32 - PK_SHIFT)), "valid count of a box must fit its field");
// ------------------------------------------------------------------ wave64 DPP scan (gfx9 DPP controls)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

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
// Exclusive prefix over the 16 lanes of every DPP row (a table row of hpk_stencil_s): four row_shr steps and a shift; lanes
// without a source read 0.  f64 has no DPP form on gfx9: two moves and an add per step.
__device__ __forceinline__ double row16_exclusive_scan(double c) {
    c += dpp_f64<DPP_ROW_SHR1, 0xf>(c);
    c += dpp_f64<DPP_ROW_SHR2, 0xf>(c);
    c += dpp_f64<DPP_ROW_SHR4, 0xf>(c);
    c += dpp_f64<DPP_ROW_SHR8, 0xf>(c);
    return dpp_f64<DPP_ROW_SHR1, 0xf>(c);
}
__device__ __forceinline__ unsigned row16_exclusive_scan(unsigned r) {
    r += dpp_u32<DPP_ROW_SHR1, 0xf>(r);
    r += dpp_u32<DPP_ROW_SHR2, 0xf>(r);
    r += dpp_u32<DPP_ROW_SHR4, 0xf>(r);
    r += dpp_u32<DPP_ROW_SHR8, 0xf>(r);
    return dpp_u32<DPP_ROW_SHR1, 0xf>(r);
}
// Inclusive prefix over the 64 lanes of the wave (the lanes' candidate counts)
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned r) {
    r += dpp_u32<DPP_ROW_SHR1, 0xf>(r);
    r += dpp_u32<DPP_ROW_SHR2, 0xf>(r);
    r += dpp_u32<DPP_ROW_SHR4, 0xf>(r);
    r += dpp_u32<DPP_ROW_SHR8, 0xf>(r);
    r += dpp_u32<DPP_ROW_BCAST15, 0xa>(r);
    r += dpp_u32<DPP_ROW_BCAST31, 0xc>(r);
    return r;
}

// Explicit window sums of one pixel at one step (rare path of the stencil): taken when the summed-area table cannot
// deliver ~1e-11 - the box sum is a difference of f64 prefix sums, so its absolute error is that of the largest
// corner; a window of small values in a tile that also holds values hundreds of thousands of times larger (a badly
// balanced bin, a count outlier) would come out with the large values' rounding noise.  The reference adds the
// window cells themselves (callers.py:175-198) and has no such failure mode.  The whole wave works on one pixel: the
// (2w + 1)^2 window cells are dealt to the 64 lanes (independent loads, one memory latency), then a wave reduction.
// (r, c, m, Wm) are wave-uniform.  Returns (bS_K, bS_Y) in every lane.
__device__ __noinline__ double2 explicit_sums_wave(const float* __restrict__ raw, const double* __restrict__ bal,
                                                   const double* __restrict__ weight, const int32_t* __restrict__ m, int Wm,
                                                   int r, int c, int n, int num, int64_t ld, int mw, int lane) {
    while (Wm > 1 && m[Wm] == 0) --Wm;                  // the step's widest ring
    const int side = 2 * Wm + 1, cells = side * side;
    double sk = 0.0, sy = 0.0;
    for (int idx = lane; idx < cells; idx += 64) {
        const int di = idx / side - Wm, dj = idx - (di + Wm) * side - Wm;
        const int adi = di < 0 ? -di : di, adj = dj < 0 ? -dj : dj;
        const int rr = r + di, cc = c + dj, kk = cc - rr;
        const bool in = di != 0 && dj != 0 && rr >= 0 && rr < n && cc >= 0 && cc < n && kk >= mw && kk < num;
        if (!in) continue;
        const int mm = m[adi > adj ? adi : adj];
        double b;
        if (bal) { b = bal[(int64_t)rr * ld + kk]; b = (b == b) ? b : 0.0; }
        else b = balanced_of(raw[(int64_t)rr * ld + kk], weight[rr], weight[cc]);
        const double v = (double)mm * b;
        sk += v;
        if (di > 0 && dj < 0) sy += v;
    }
    for (int off = 32; off > 0; off >>= 1) { sk += __shfl_xor(sk, off); sy += __shfl_xor(sy, off); }
    return make_double2(sk, sy);
}

// Per-phase cycle accounting (-DHPK_PHASE_CLOCK builds only; scripts/measure/gpu_phase_clock.sh): every wave sums s_memtime
// deltas per phase of the tile loop and leaves them in a.clk[(workgroup * NW + wave) * 8 + phase].
#if defined(HPK_PHASE_CLOCK) && defined(HPK_WG_LIFE)
// (-DHPK_PHASE_CLOCK -DHPK_WG_LIFE: no marks in the tile loop - a wave's first and last s_memrealtime in slots 6 / 7, the phase slots
//  left at 1: when the workgroups of a launch start and end, scripts/measure/wg_life.py)
#define HPK_CLK_DECL unsigned long long ck0 = 1, ck1 = 0, ck2 = 0, ck3 = 0, ck4 = 0, ck5 = 0, ck6 = __builtin_amdgcn_s_memrealtime(), ck7 = 0;
#define HPK_CLK(v)
#elif defined(HPK_PHASE_CLOCK)
#define HPK_CLK_DECL unsigned long long ck0 = 0, ck1 = 0, ck2 = 0, ck3 = 0, ck4 = 0, ck5 = 0, ck6 = 0, ck7 = 0, ckt = __builtin_readcyclecounter();
#define HPK_CLK(v) { const unsigned long long t__ = __builtin_readcyclecounter(); v += t__ - ckt; ckt = t__; }
#else
#define HPK_CLK_DECL
#define HPK_CLK(v)
#endif
#ifndef HPK_DYN_BATCH
#define HPK_DYN_BATCH 1                 // hpk_stencil_s: the batches of a tile beyond the waves' first are dealt dynamically
#endif
#ifndef HPK_SCORE_WPE
#define HPK_SCORE_WPE 6                 // hpk_score: waves per SIMD the register allocation is held to (80 VGPRs; the Emax registers would have made it 83 -> 88 allocated -> 5)
#endif
#ifndef HPK_SCORE_ODD_MASK
#define HPK_SCORE_ODD_MASK 0            // hpk_score: "this batch needs the general chunk rules" as lane masks ORed on the scalar side
#endif
#ifndef HPK_SCORE_REUSE
#define HPK_SCORE_REUSE 1               // hpk_score, several pairs: the candidate's own loads stay for the pairs of a batch
#endif
// -DHPK_CLK_P1 (with HPK_PHASE_CLOCK): the top of the tile - from the last barrier to the rows' arrival - booked under slot 4
// instead of slot 0.  (A mark is an s_memtime and a wait for it, sixteen waves at a time behind every barrier: the build runs
// ~20 % slower than the plain one and the slots carry a few hundred ticks of that each - shares below ~5 % of a tile, and
// what sits between two marks close together, are the marks' own.  Round 6 read a 26 % "band switch" and a 29 % "wait for
// rows" out of them that timing the plain build with either removed did not confirm: profiles/r06_stencil_band_switch.txt.)

// A chromosome's resolve totals (HpkBandDesc::hist_acc): HPK_HREP copies, a stencil workgroup adds to the one its index picks -
// 256 workgroups leave a band within microseconds of each other, and device-wide adds to one word run one after the other -,
// the readers add the copies up.
__device__ __forceinline__ unsigned long long hist_total(const unsigned long long* __restrict__ acc, int i) {
    unsigned long long v = 0ull;
#pragma unroll
    for (int r = 0; r < HPK_HREP; ++r) v += acc[r * HPK_ACC_STRIDE + i];
    return v;
}

// ------------------------------------------------------------------ the stencil kernel
// hpk_stencil_s, per tile (1024 threads = 16 waves, one persistent workgroup per CU):
//   * the next tile's band elements are requested a tile ahead: ten consecutive elements per lane in three buffer loads
//     (elements outside the matrix or the stored diagonals are zeroed where they are used), the lane's row weight, and -
//     160 threads - the tile's column weights, which go through a small LDS table;
//   * phase 1 (rows): balanced values and packed cells of the lane's ten cells, the row prefix of both planes (nine adds
//     in the lane, a scan over the 16 lanes of the DPP row that holds the table row), written straight to the tables; the
//     candidates - cells with a count inside a range of cells that the lane works out once per tile - go to ONE tile-wide
//     list (the wave's slice is reserved with an LDS atomic whose return is consumed after the f64 prefix);
//   * phase 2 (columns): a thread per column and chunk of rows sums its chunk in registers, parks the chunk's total and,
//     behind a barrier, writes its cells back with the totals of the chunks above;
//   * the table's row direction is reversed (entry (Y, X) = sum over rows <= Y and columns >= X): next to the main
//     diagonal the counts fall by two orders of magnitude from left to right, and the large values must not sit under the
//     far pixels' boxes (the exact fallback below is left for genuine outliers);
//   * phase 3: batches of 64 candidates dealt round-robin to the waves; the Reads boxes that decide most candidates are
//     read together, the resolve histogram is kept per *width* (converted to steps once per workgroup), the plan sits in
//     LDS; the scoring work list's append of tile i is finished while tile i + 1 ends.
// The code is written branch-free where the compiler would otherwise wrap every cell in its own exec-mask branch
// (selects on the inputs instead of ifs around the arithmetic).
// (HPK_TLIST, hpk_kernels.h: the tile-wide candidate list holds TR * TC entries - 50 x 151 at a halo of 4, 51 x 147 at 6)
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ldbuf_f32(rsrc_t r, unsigned off, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, (int)soff, 0));
}
__device__ __forceinline__ double ldbuf_f64(rsrc_t r, unsigned off, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, (int)soff, 0));
}
// the lane mask of a predicate as the compiler holds it (HIP's __ballot goes through an int: v_cndmask + v_cmp per call)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// The same two on LDS byte addresses (phase 3 of hpk_stencil_s): per read one add of an offset to the cell's own address
// instead of index arithmetic plus a scale, neighbours through the instruction's immediate.  pb = address of P(Y, X) in
// the packed plane, cb = in the f64 plane (both include the workgroup's LDS base).
typedef __attribute__((address_space(3))) const unsigned lds_cu32_t;
typedef __attribute__((address_space(3))) const double lds_cf64_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
__device__ __forceinline__ unsigned lds_u32(unsigned addr) { return *(lds_cu32_t*)addr; }
__device__ __forceinline__ unsigned lds_u16(unsigned addr) { return (unsigned)*(__attribute__((address_space(3))) const unsigned short*)addr; }
__device__ __forceinline__ void lds_st_u16(unsigned addr, unsigned v) { *(__attribute__((address_space(3))) unsigned short*)addr = (unsigned short)v; }
__device__ __forceinline__ void lds_st_u32(unsigned addr, unsigned v) { *(__attribute__((address_space(3))) unsigned*)addr = v; }
// (volatile: two ds_read_b64 take 2 LDS cycles each, the ds_read2_b64 the compiler would pair them into takes 8)
__device__ __forceinline__ double lds_f64(unsigned addr) { return *(volatile lds_cf64_t*)addr; }
__device__ __forceinline__ unsigned reads_box_b(unsigned pb, int rho, unsigned sr) {
    const unsigned dn = (unsigned)rho * (unsigned)(LC * 4), lf = (unsigned)rho * 4u;       // (rho is uniform: scalar offsets)
    return (lds_u32(pb + (dn - lf)) - lds_u32(pb + dn) - lds_u32(pb - lf) + sr) & PK_MASK;
}
__device__ __forceinline__ void box_ky_b(unsigned cb, int rho, double pixc, double sc, double& kc, double& yc) {
    constexpr unsigned PITCH = LC * 8;                               // row pitch of the f64 plane
    unsigned r8 = (unsigned)rho << 3, rL = (unsigned)rho * PITCH;
    asm volatile("" : "+v"(r8), "+v"(rL));                           // (sums of the two, not multiplies of rho)
    const unsigned up = cb - PITCH;                                  // row Y - 1
    const unsigned u = rL + r8, d = rL - r8, at = up - rL, ab = cb + rL, al = up - r8, ar = up + r8;
    const double tl = lds_f64(up - u), tm = lds_f64(at), tm1 = lds_f64(at + 8), tr = lds_f64(up - d + 8);
    const double bl = lds_f64(cb + d), bm = lds_f64(ab), bm1 = lds_f64(ab + 8), br = lds_f64(cb + u + 8);
    const double ml0 = lds_f64(al), ml1 = lds_f64(al + PITCH), mr0 = lds_f64(ar + 8), mr1 = lds_f64(ar + 8 + PITCH);
    const double bot = (bl - bm) + (bm1 - br);          // rows <= Y + rho, columns [X - rho, X - 1] and [X + 1, X + rho]
    const double top = (tl - tm) + (tm1 - tr);          // rows <= Y - rho - 1, same columns
    const double mid = (ml1 - mr1) - (ml0 - mr0);       // row Y, columns [X - rho, X + rho]
    kc = ((bot - top) - mid) + pixc;
    yc = (bl - bm) - (ml1 - sc);
}
// The box coefficients of a step add up to 0 whenever its innermost box has a radius (Box(w*) - Box(p) for a single
// pair): the pixel's own value then cancels in the donut sum and is neither read nor added.  kc: the donut sum without
// it; yc as in box_ky_b - with P(Y, X), so that a lower-left box over empty rows stays an exact 0 (its four corners are
// pairwise the same numbers); big: the largest table entry the sums touch (what their rounding noise scales with).
__device__ __forceinline__ void box_ky_d(unsigned cb, int rho, double sc, double& kc, double& yc, double& big) {
    constexpr unsigned PITCH = LC * 8;
    unsigned r8 = (unsigned)rho << 3, rL = (unsigned)rho * PITCH;
    asm volatile("" : "+v"(r8), "+v"(rL));
    const unsigned up = cb - PITCH;
    const unsigned u = rL + r8, d = rL - r8, at = up - rL, ab = cb + rL, al = up - r8, ar = up + r8;
    const double tl = lds_f64(up - u), tm = lds_f64(at), tm1 = lds_f64(at + 8), tr = lds_f64(up - d + 8);
    const double bl = lds_f64(cb + d), bm = lds_f64(ab), bm1 = lds_f64(ab + 8), br = lds_f64(cb + u + 8);
    const double ml0 = lds_f64(al), ml1 = lds_f64(al + PITCH), mr0 = lds_f64(ar + 8), mr1 = lds_f64(ar + 8 + PITCH);
    const double bot = (bl - bm) + (bm1 - br);
    const double top = (tl - tm) + (tm1 - tr);
    const double mid = (ml1 - mr1) - (ml0 - mr0);
    kc = (bot - top) - mid;
    yc = (bl - bm) - (ml1 - sc);
    big = bl;
}
__device__ __noinline__ unsigned long long box_ky_valid_m(const unsigned* __restrict__ Pv, int base, int rho, unsigned pixv, unsigned sv) {
    const int t = base - (rho + 1) * LC, b = base + rho * LC, m0 = base - LC;
    const unsigned tl = Pv[t - rho], tm = Pv[t], tm1 = Pv[t + 1], tr = Pv[t + rho + 1];
    const unsigned bl = Pv[b - rho], bm = Pv[b], bm1 = Pv[b + 1], br = Pv[b + rho + 1];
    const unsigned ml1 = Pv[base - rho], ml0 = Pv[m0 - rho], mr1 = Pv[base + rho + 1], mr0 = Pv[m0 + rho + 1];
    const unsigned kv = (bl - bm + bm1 - br) - (tl - tm + tm1 - tr) - (ml1 - mr1 - ml0 + mr0) + pixv;
    const unsigned yv = bl - bm - ml1 + sv;
    return (unsigned long long)(kv >> PK_SHIFT) | (unsigned long long)(yv >> PK_SHIFT) << 32;
}

// What a lane holds of the next tile while the current one is worked on.  Tile geometry of hpk_stencil_s: SAT row Y <->
// matrix row r0 - W - 1 + Y (row 0 is the table's origin row); SAT column X <-> matrix column c0 - W + X (columns run the
// other way: X = LC - 1 is the origin side).  Output pixel (y, x) of the tile sits at (Y, X) = (y + W + 1, x + W).
// A table row belongs to one DPP row of 16 lanes: lane l of wave w holds the cells X = 159 - 10 (l % 16) - e, e = 0..9, of
// SAT row Y = 4 w + l / 16 - ten consecutive band elements, fetched as they lie in memory (mem[i] <-> e = 9 - i).  The row
// prefix is then nine adds inside the lane and a scan over 16 lanes (row_shr steps only), for four table rows at a time.
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4u32 ldbuf_v4(rsrc_t r, unsigned off) { return __builtin_bit_cast(v4u32, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); }
__device__ __forceinline__ v2u32 ldbuf_v2(rsrc_t r, unsigned off) { return __builtin_bit_cast(v2u32, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }

template <bool BALF64>
struct TileRegsS {
    unsigned raw[10];                   // f32 bit patterns, memory order
    double bal[BALF64 ? 10 : 1];        // f64 input mode: balanced values as given, memory order
    double wrow;                        // weight mode: weight of the lane's table row
    double wcol;                        // weight mode, threads 0..LC-1: weight of SAT column X = thread (goes to the LDS table)
};

template <bool BALF64>
__device__ __forceinline__ void tile_load_s(const HpkStencilArgs& a, const HpkBandDesc* __restrict__ bd, int rb, int cj, int wave,
                                            int lane, TileRegsS<BALF64>& t) {
    const int bn = bd->n;
    const int64_t bld = bd->ld;
    const int r0 = rb * bd->TR;
    const int rt0 = r0 - bd->W - 1;                        // matrix row of SAT row 0 (negative in the first row block)
    // The tile's buffer: one row more than the tile has on either side, where the band has them - a wide load of the tile's
    // first row may start in the row before (negative diagonals), one of its last row may run on into the next.
    const int rb0 = rt0 > 1 ? rt0 - 1 : 0;
    int rows = bn - rb0;
    rows = rows > LR + 2 ? LR + 2 : rows;
    const unsigned ldu = (unsigned)bld;
    const int koff = a.mw + cj * bd->TC + 1;               // diagonal of SAT cell (0, 0): (c0 - W) - (r0 - W - 1)
    const int Y = 4 * wave + (lane >> 4);
    const int XL = (LC - 10) - 10 * (lane & 15);           // the lane's lowest column (cell e = 9)
    const rsrc_t rraw = make_rsrc(gptr(bd->raw) + (int64_t)rb0 * bld, (unsigned)rows * ldu * 4u);
    if (!BALF64) {
        const rsrc_t rw = make_rsrc(gptr(bd->weight), (unsigned)bn * 8u);
        // (a masked bin's weight is NaN, scripts/pyHICCUPS:163-166; it is turned into 0 where it is used - once per table
        // column and once per lane - so that the products of its pixels are zeros without a NaN test per cell)
        if (wave < 3) t.wcol = ldbuf_f64(rw, (unsigned)(rt0 + koff + wave * 64 + lane) * 8u, 0u);    // columns < 0 or >= n read 0
        t.wrow = ldbuf_f64(rw, (unsigned)(rt0 + Y) * 8u, 0u);
    }
    // Ten consecutive elements of band row rt0 + Y from diagonal koff + XL - Y on.  Elements outside the matrix or the stored
    // diagonals are zeroed where they are used (phase 1 knows the row's limits); a start below the row (negative diagonal)
    // reads the tail of the row before or - first row of the buffer, rows above the matrix - out of bounds (0); rows at or
    // beyond the matrix end are out of bounds.  Wide loads: a stored element never shares one with bytes beyond the band
    // (rows of 16 elements or more; narrower bands take ten loads).
    // (matrix row 0 has no row before: the lanes whose ten elements straddle the start of the band take single loads too)
    const int flat = (rt0 + Y - rb0) * (int)ldu + (koff + XL - Y);
    const unsigned off = (unsigned)flat * 4u;
    const bool wide = ldu >= 16u && !(flat < 0 && flat > -10);
    if (wide) {
        const v4u32 q0 = ldbuf_v4(rraw, off), q1 = ldbuf_v4(rraw, off + 16u);
        const v2u32 q2 = ldbuf_v2(rraw, off + 32u);
        t.raw[0] = q0.x; t.raw[1] = q0.y; t.raw[2] = q0.z; t.raw[3] = q0.w;
        t.raw[4] = q1.x; t.raw[5] = q1.y; t.raw[6] = q1.z; t.raw[7] = q1.w;
        t.raw[8] = q2.x; t.raw[9] = q2.y;
    } else {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            unsigned o1 = off + 4u * (unsigned)i;
            asm volatile("" : "+v"(o1));        // (opaque: the compiler would merge the ten into wide loads again)
            t.raw[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rraw, (int)o1, 0, 0);
        }
    }
    if (BALF64) {
        const rsrc_t rbal = make_rsrc(gptr(bd->bal) + (int64_t)rb0 * bld, (unsigned)rows * ldu * 8u);
        if (wide) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const v4u32 d = ldbuf_v4(rbal, off * 2u + 16u * (unsigned)i);
                t.bal[2 * i] = __hiloint2double((int)d.y, (int)d.x);
                t.bal[2 * i + 1] = __hiloint2double((int)d.w, (int)d.z);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                unsigned o1 = off * 2u + 8u * (unsigned)i;
                asm volatile("" : "+v"(o1));
                t.bal[i] = ldbuf_f64(rbal, o1, 0u);
            }
        }
    }
}

// Persistent tile walk over a batch of bands, without divisions inside a band.  XCD x owns, of every band b, the
// contiguous run of tiles [x * chunk_b, (x + 1) * chunk_b); the runs of the batch's bands laid end to end form the XCD's
// index space, and the XCD's workgroups walk it with the stride of the workgroups per XCD - straight across the band
// boundaries, so that a batch has one ramp-up and one tail whatever the number of bands.  Inside a band, (row block, column
// chunk) advance by the stride's quotient and remainder by J; a band switch recomputes them with a division (rare).
// A band's tiles are walked by two kernels: MODE 0 (hpk_stencil_s) takes the column chunks below the band's lean_cj, MODE 1
// (hpk_stencil_lean) the chunks from lean_cj on; the tiles the lean kernel could not finish go through hpk_stencil_s once more,
// in a small launch of their own that walks the redo queue instead (QueueWalk).
#define HPK_BW_REDO 0x80000000u         // bword: a tile out of the redo queue (its candidates were counted by the lean kernel)
template <int MODE>
struct TileWalk {
    int k, band, kb, chunkb, ntb;       // index in the XCD's run; current band, its first index, its chunk and tile count
    int J, clo;                         // column chunks of the current band this walk takes, the first of them (every band has its own tile geometry)
    int rbk, ck, rm;                    // row block, column chunk before rotation, row block mod J
    int dk, dr, dc, drm;                // per step: index stride, its quotient and remainder by J, quotient mod J
    int nt;                             // tiles handed out so far (see bword)
    bool done;
    __device__ __forceinline__ void locate(const HpkStencilArgs& a, const HpkBandDesc* __restrict__ bands, bool fresh) {
        const int xcd = (int)(blockIdx.x & 7);
        for (;;) {
            while (k >= kb + chunkb) {
                kb += chunkb;
                ++band;
                if (band >= a.nbands) { done = true; return; }
                const HpkBandDesc* __restrict__ nb = bands + __builtin_amdgcn_readfirstlane(band);
                const int Jr = nb->J, lc = nb->lean_cj;
                const int jf = lc < Jr ? (lc > 0 ? lc : 0) : Jr;       // chunks that are not lean
                J = MODE == 0 ? jf : Jr - jf;
                clo = MODE == 0 ? 0 : jf;
                ntb = (nb->ntiles / Jr) * J;
                chunkb = (ntb + 7) / 8;
                if (J > 0) { dr = dk / J; dc = dk - dr * J; drm = dr % J; }
                fresh = true;
            }
            // (which eighth of a band an XCD takes turns with the band: the eighths are not equally heavy - the last one ends in the
            //  matrix's corner - and a batch evens that out over its bands: round 6, scripts/measure/wg_life.py)
            const int t = ((xcd + band) & 7) * chunkb + (k - kb);
            if (t < ntb) {
                if (fresh) { rbk = t / J; ck = t - rbk * J; rm = rbk % J; }
                return;
            }
            // (the last XCD's run of a band can be shorter than the others': on to the next band)
            k += (kb + chunkb - k + dk - 1) / dk * dk;
        }
    }
    __device__ __forceinline__ void init(const HpkStencilArgs& a, const HpkBandDesc* __restrict__ bands) {
        dk = (int)(gridDim.x >> 3);
        J = 1; clo = 0; dr = dk; dc = 0; drm = 0;
        k = (int)(blockIdx.x >> 3);
        band = -1; kb = 0; chunkb = 0; ntb = 0;
        rbk = 0; ck = 0; rm = 0; nt = 0;
        done = false;
        locate(a, bands, true);
    }
    // Column chunk of the tile: rotated by the row block, and by the number of this walk's strides below the row block -
    // when the stride is a multiple of J (J = 4 on 32 workgroups per XCD) the row block's own rotation stands still, and a
    // workgroup would meet one density class only.  (A function of the row block alone: the J tiles of a row block still
    // take the J chunks.  Two scalar divisions per tile, in the one wave that walks.)
    __device__ __forceinline__ int cj(const HpkStencilArgs& a) const {
        if (a.order == 0) return clo + ck;
        const int c = ck + rm + (rbk / (dr > 0 ? dr : 1)) % J;
        return clo + (c >= J ? (c >= 2 * J ? c - 2 * J : c - J) : c);
    }
    __device__ __forceinline__ void step(const HpkStencilArgs& a, const HpkBandDesc* __restrict__ bands) {
        k += dk; rbk += dr; ck += dc; rm += drm; nt += 1;
        if (ck >= J) { ck -= J; rbk += 1; rm += 1; }
        if (rm >= J) rm -= J;
        if (rm >= J) rm -= J;
        locate(a, bands, false);
    }
    // what wave 0 publishes for the others: row block << 8 | column chunk, ~0 = no more tiles
    __device__ __forceinline__ unsigned word(const HpkStencilArgs& a) const { return done ? ~0u : ((unsigned)rbk << 8 | (unsigned)cj(a)); }
    // ... and the band's index with a segment number on top that steps every 8192 tiles of the walk: the workgroup flushes
    // its resolve counts whenever this word changes, i.e. at every band boundary and - long runs of tiles on small grids -
    // before a 16-bit counter field can wrap (a wave runs at most 7 batches of a tile, a lane counts at most one
    // candidate per batch), at no cost in the fifteen waves that do not walk
    __device__ __forceinline__ unsigned bword() const { return (unsigned)band | (((unsigned)(nt >> 13) & 0x7fffu) << 16); }
};
// The same interface over the redo queue {tiles queued, tiles taken, -, -, (band, row block << 8 | column chunk) ...}: workgroup
// g takes the entries g, g + grid, ... (the queue is complete when the launch starts: hpk_stencil_lean ran before it).
struct QueueWalk {
    unsigned qi, qn, qband, qword;
    int rbk;
    bool done;
    __device__ __forceinline__ void fetch(const HpkStencilArgs& a) {
        done = qi >= qn;
        if (!done) { qband = a.redoq[4 + 2 * qi]; qword = a.redoq[5 + 2 * qi]; }
        rbk = (int)(qword >> 8);
    }
    __device__ __forceinline__ void init(const HpkStencilArgs& a, const HpkBandDesc* __restrict__) {
        qn = a.redoq ? a.redoq[0] : 0u;
        qi = blockIdx.x; qband = 0u; qword = 0u;
        fetch(a);
    }
    __device__ __forceinline__ void step(const HpkStencilArgs& a, const HpkBandDesc* __restrict__) { qi += gridDim.x; fetch(a); }
    __device__ __forceinline__ int cj(const HpkStencilArgs&) const { return (int)(qword & 255u); }
    __device__ __forceinline__ unsigned word(const HpkStencilArgs&) const { return done ? ~0u : qword; }
    __device__ __forceinline__ unsigned bword() const { return qband | HPK_BW_REDO; }
};

// The walk's state parked in LDS between the steps of the one wave that walks (words; the loads come back as scalars)
template <class T>
__device__ __forceinline__ void walk_park(unsigned* __restrict__ l, const T& t, int lane) {
    static_assert(sizeof(T) % 4 == 0, "walk state in words");
    unsigned w[sizeof(T) / 4];
    __builtin_memcpy(w, &t, sizeof(T));
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < (int)(sizeof(T) / 4); ++i) l[i] = w[i];
    }
}
template <class T>
__device__ __forceinline__ void walk_take(const unsigned* __restrict__ l, T& t) {
    unsigned w[sizeof(T) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) w[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)l[i]);
    __builtin_memcpy(&t, w, sizeof(T));
}

template <bool BALF64, bool SINGLE, bool QUEUE>
__global__ void __launch_bounds__(1024) hpk_stencil_s(HpkStencilArgs a, const HpkBandDesc* __restrict__ bands) {
    constexpr int NW = 16;
    static_assert(LR == 4 * NW && LC == 160, "tile geometry: four table rows per wave, ten cells per lane");
    // the queue pass: a workgroup without a queued tile leaves before the prologue (the queue is empty as a rule, and 64
    // workgroups setting themselves up for nothing took 50 us of every batch)
    if (QUEUE) { if (!a.redoq || blockIdx.x >= a.redoq[0]) return; }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* __restrict__ Sc = reinterpret_cast<double*>(smem);
    unsigned* __restrict__ Sp = reinterpret_cast<unsigned*>(smem + (size_t)LR * LC * 8);
    unsigned* __restrict__ lst = reinterpret_cast<unsigned*>(smem + (size_t)LR * LC * 12);      // [HPK_TLIST] tile-wide candidate list
    unsigned* __restrict__ tcount = lst + HPK_TLIST;       // [2] entries in the list, tiles alternate
    // (the dynamic LDS of a kernel without static LDS starts at address 0: the addresses phase 3 reads by are immediates)
    constexpr unsigned lds0 = 0u;
    if ((unsigned)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();
    // the widening plan as the batches read it: per step 8 words {w0 (HpkDevPlan::packed[0]), four words of box terms},
    // then step_of[slot][width] as bytes
    unsigned* __restrict__ pl = tcount + 32;                          // [HPK_MAX_STEPS][8]
    unsigned char* __restrict__ stepof = reinterpret_cast<unsigned char*>(pl + HPK_MAX_STEPS * 8);   // [HPK_KSLOTS][32]
    double* __restrict__ wct = reinterpret_cast<double*>(stepof + HPK_KSLOTS * 32);      // [LC] column weights of the tile (NaN -> 0)
    double* __restrict__ ctot = wct + LC;                              // [3][LC] phase 2: totals of the f64 plane's row chunks
    unsigned* __restrict__ utot = reinterpret_cast<unsigned*>(ctot + 3 * LC);            // [LC] ... of the packed plane's first chunk

    unsigned* __restrict__ twl = utot + LC;                      // [16] the tile walk's state between wave 0's steps
    // resolve counts of the band (segment) the walk left, summed over the waves: [2][HPK_HACC] - lane-indexed widths / steps, then
    // the candidates -, two buffers in turn; per step its width index | its slot's first width << 8 (what turns widths into steps)
    unsigned* __restrict__ hacc = twl + 16;
    unsigned* __restrict__ hstep = hacc + 2 * HPK_HACC;          // [HPK_MAX_STEPS]

    const int lane_k = threadIdx.x & 63;
    const int wave_k = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int mw = a.mw;

    const HpkDevPlan* __restrict__ plan = a.plan;
    const int nsteps = plan->nsteps, nslots = plan->nslots;
    int maxnkt = 0;
    {
        int pk0 = 0;
        if (lane_k < nsteps) pk0 = (int)plan->packed[lane_k][0];
        if (wave_k == 0) {
            if (lane_k < nsteps) {
                const uint32_t* pk = plan->packed[lane_k];
                pl[lane_k * 8 + 0] = pk[0]; pl[lane_k * 8 + 1] = pk[3]; pl[lane_k * 8 + 2] = pk[4]; pl[lane_k * 8 + 3] = pk[5];
                pl[lane_k * 8 + 4] = pk[6];
                pl[lane_k * 8 + 5] = pk[1]; pl[lane_k * 8 + 6] = pk[2];          // (general plans: the Reads box terms of the step)
            }
            stepof[lane_k] = plan->step_of[lane_k >> 5][lane_k & 31];
            stepof[64 + lane_k] = plan->step_of[2 + (lane_k >> 5)][lane_k & 31];
            if (lane_k < 2) { tcount[lane_k] = 0u; tcount[4 + lane_k] = 0u; tcount[6 + lane_k] = 0u; }     // list entries | records written | batches dealt, tiles alternate
        }
        if (wave_k == 1) {
            unsigned hs = 0u;
            if (lane_k < nsteps) { const HpkDevStep& st = plan->steps[lane_k]; hs = (unsigned)st.wi | ((unsigned)plan->slot_wfirst[st.slot] << 8); }
            hstep[lane_k] = hs;
            hacc[lane_k] = 0u; hacc[64 + lane_k] = 0u;
            if (lane_k < 2 * HPK_HACC - 128) hacc[128 + lane_k] = 0u;
        }
        for (int s = 0; s < nsteps; ++s) {
            const int k = (__builtin_amdgcn_readlane(pk0, s) >> 20) & 15;
            maxnkt = k > maxnkt ? k : maxnkt;
        }
    }
    int wf_q[HPK_KSLOTS];
#pragma unroll
    for (int q = 0; q < HPK_KSLOTS; ++q) wf_q[q] = __builtin_amdgcn_readfirstlane(plan->slot_wfirst[q]);
    const int nslots_p = __builtin_amdgcn_readfirstlane(nslots);
    const int minr_p = __builtin_amdgcn_readfirstlane(plan->min_reads), p0_p = __builtin_amdgcn_readfirstlane(plan->reads_p0);
    const int wmin_p = __builtin_amdgcn_readfirstlane(plan->wmin);
    const int sp_p = __builtin_amdgcn_readfirstlane(plan->single_p);     // SINGLE: the peak width
    const unsigned pkcap_p = (unsigned)__builtin_amdgcn_readfirstlane(plan->pk_cap);
    const int fr_p = SINGLE ? 0 : __builtin_amdgcn_readfirstlane(plan->first_rho);   // general plans: the box every step starts with
    // Plans whose Reads matrix is not monotone in the width (pairs listed in decreasing order, callers.py:15-23 sorts the steps
    // by width then peak width): no "first sufficient width" - the steps are walked in plan order, per slot the first one whose
    // Reads reach min_local_reads resolves (callers.py:203-217), and the resolve histogram is kept per step.
    const bool generic_p = !SINGLE && a.generic != 0;
    // a halo wider than the plan's maxww (maxww < 4: the tiles keep a halo of 4): widths beyond maxww have no step
    const int planw_p = __builtin_amdgcn_readfirstlane(plan->W);
    // Resolve histogram of the band the workgroup is in: flushed into the band's totals when the walk enters the next
    // band (and after the last tile), see the top of the tile loop.
    unsigned myhist = 0u;                 // lane w: candidates whose first sufficient width is w
    unsigned long long hpack0 = 0ull, hpack1 = 0ull;      // this lane's candidates by width min(ww) + k: 16 bits each, k = 0..3 | 4..7
    unsigned mycand = 0u;
    // the per-lane width counts, summed over the wave, into lane min(ww) + k of myhist
    auto fold_hpack = [&]() {
#pragma unroll
        for (int k = 0; k < (BALF64 ? 0 : 8); ++k) {
            unsigned v = (unsigned)((k < 4 ? hpack0 : hpack1) >> (16 * (k & 3))) & 0xffffu;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) v += (unsigned)__shfl_xor((int)v, m);
            if (lane_k == wmin_p + k) myhist += v;
        }
        hpack0 = 0ull; hpack1 = 0ull;
    };
    // scoring work list: the append of a tile is completed one tile later (the atomic's return is not waited for)
    int pend_tid = -1, pend_band = 0;
    unsigned pend_c = 0u, pend_off = 0u, pend_rc = 0u;      // (pend_rc: the tile as the work list names it, row block << 8 | column chunk)

    HPK_CLK_DECL
    TileRegsS<BALF64> nxt;
    int par = 0;                        // which of the two list counters the current tile uses
    // The tile walk is the same scalar arithmetic in every wave, and all sixteen would queue for the one scalar unit
    // with it at the top of every tile: wave 0 alone walks, one tile ahead, and publishes the next tile through LDS
    // (tseq[2] / tband[2], alternating: row block << 8 | column chunk (~0 = no more tiles) and the band's index).
    unsigned* __restrict__ tseq = tcount + 8;
    unsigned* __restrict__ tband = tcount + 12;
    // (QUEUE: the launch over the tiles hpk_stencil_lean gave up)
    typedef typename std::conditional<QUEUE, QueueWalk, TileWalk<0>>::type walk_t;
    bool have;
    int rb, cj;
    unsigned bw;                        // band (low 16 bits) and flush segment / redo mark of the current tile
    {
        walk_t tw;
        tw.init(a, bands);
        have = !tw.done; rb = tw.rbk; cj = tw.cj(a); bw = tw.bword();
        if (wave_k == 0) {
            tw.step(a, bands);
            if (lane_k == 0) { tseq[0] = tw.word(a); tband[0] = tw.bword(); }
            walk_park(twl, tw, lane_k);
        }
    }
    __syncthreads();                    // plan, counters and the second tile in LDS
    unsigned tnext = lds_u32(lds0 + (unsigned)((unsigned char*)tseq - smem));
    unsigned bnext = lds_u32(lds0 + (unsigned)((unsigned char*)tband - smem));
    int tpar = 0;                       // which tseq word holds the tile after the current one
    // ---- the resolve counts of a band (segment) go to that band's totals when the walk leaves it.  Every wave folds its lanes'
    // width counts and adds them to the LDS buffer of the turn - no barrier: the walk is already in the next band's first tile,
    // whose rows were requested a tile ahead like any other's.  Behind that tile's first barrier wave 15 - idle in phase 2 -
    // turns the widths into steps (per step s of slot q and width w: the candidates whose first sufficient width is w - w above
    // the slot's first width - or at most w - at it) and adds them to the chromosome's totals, one word per cache line, without
    // waiting for them; every scoring workgroup replays the freeze decision on the totals.  No ticket, no fences, no tail.
    auto flush_hist = [&](int p) {
        fold_hpack();
        atomicAdd(&hacc[p * HPK_HACC + lane_k], myhist);
        if (wave_k == 0 && lane_k == 0) atomicAdd(&hacc[p * HPK_HACC + 64], mycand);
        myhist = 0u; mycand = 0u;
    };
    // (wave 15, behind a barrier that follows the flush)
    auto publish_hist = [&](const HpkBandDesc* __restrict__ hb, int p) {
        unsigned* __restrict__ hw = hacc + p * HPK_HACC;
        unsigned long long* __restrict__ acc = gptr(hb->hist_acc) + HPK_HREP_OF(blockIdx.x) * HPK_ACC_STRIDE;
        unsigned out = 0u;
        if (lane_k < nsteps) {
            const unsigned hs = hstep[lane_k];
            const int wi = (int)(hs & 0xffu), wf = (int)(hs >> 8);
            if (generic_p) out = hw[lane_k];                    // (counted per step)
            else if (wi > wf) out = hw[wi & 63];
            else for (int w = 0; w <= wf && w < 64; ++w) out += hw[w];
        }
        const unsigned nc = hw[64];
        if (out) atomicAdd(&acc[lane_k], (unsigned long long)out);
        if (lane_k == 0 && nc) atomicAdd(&acc[HPK_MAX_STEPS], (unsigned long long)nc);
        hw[lane_k] = 0u;
        if (lane_k == 0) hw[64] = 0u;
    };
    int hp = 0;                         // the buffer of the next flush
    int pub = -1;                       // a flushed band waiting for wave 15: band << 1 | buffer
    bool fetched = false;               // the tile on top of the loop has its rows on the way (every tile but the workgroup's first)
    int hband = -1;                     // band whose resolve counts are pending in myhist / hpack / mycand
#pragma unroll 1
    while (have) {
    // ---- one band (segment): what the tile loop needs of it - pointers, sizes, the record bound - does not change below
    const unsigned cbw = bw;
    const int band = (int)(cbw & 0xffffu);
    const HpkBandDesc* __restrict__ bd = bands + band;
    const int n = bd->n, bnum = bd->num;
    const int Dm = a.D < bd->num - 1 ? a.D : bd->num - 1;   // last diagonal that holds band pixels
    // Records are written for candidates whose first sufficient width is at most the band's wguess, packed (a tile's
    // record i is no longer its list entry i): the widening stops at a width that only the whole chromosome's histogram
    // decides (frozen_w, freeze_replay), wider candidates and unresolved ones are dropped by the scoring kernel anyway,
    // and the caller knows a bound from the chromosomes before (hpk_api.cpp; 255 = every candidate, as the dense outputs want).
    const int wg_p = bd->wguess;
    // the band's tile geometry (hpk_geo_of)
    const int W = bd->W, TR = bd->TR, TC = bd->TC, J_p = bd->J, Dg_p = bd->Dg, tilecap_p = bd->tilecap;
    // a tile out of the redo queue (hpk_stencil_lean gave it up): its candidates and their resolve counts are in the band's totals
    const bool is_redo = QUEUE;
    // the workgroup's first tile: nobody asked for its rows yet (all others, across band boundaries too: a tile ahead)
    if (!fetched) tile_load_s<BALF64>(a, bd, rb, cj, wave_k, lane_k, nxt);
    if (hband >= 0) { flush_hist(hp); pub = hband << 1 | hp; hp ^= 1; }
    hband = band;
    if (!BALF64 && !fetched) {          // the first tile's column weights (the tiles after it: behind their predecessor's tables)
        const int tix = wave_k * 64 + lane_k;
        if (tix < LC) wct[tix] = nxt.wcol == nxt.wcol ? nxt.wcol : 0.0;
        __syncthreads();
    }
#pragma unroll 1
    do {
    // Everything below that depends only on (wave, lane) is the same for every tile, and the compiler would hoist it
    // out of the tile loop - list-entry templates, row flags, compare constants: 100+ SGPRs and a dozen VGPRs that
    // then spill to scratch, whose reloads (vmcnt(0)) also wait for the prefetch.  Opaque copies keep it in here.
    int wave = wave_k, lane = lane_k;
    asm volatile("" : "+s"(wave));
    asm volatile("" : "+v"(lane));
    const int tid = rb * J_p + cj;
    const int r0 = rb * TR;
    const int c0 = r0 + mw + cj * TC;
    // (a.Dg, not a.D: with a halo below maxww the stored diagonals beyond D - read for the gap rows only, callers.py:238 -
    // reach further than the last candidates' tile sees)
    const bool empty_tile = c0 >= n || (mw + cj * TC - (TR - 1)) > Dg_p;      // no stored pixel inside the matrix
    const int tn = __builtin_amdgcn_readfirstlane((int)tnext);
    const unsigned bw_next = (unsigned)__builtin_amdgcn_readfirstlane((int)bnext);
    const bool have_next = tn != -1;
    const bool pre_next = have_next;                            // the next tile's rows are prefetched, whatever its band
    const HpkBandDesc* __restrict__ bd_next = bands + (bw_next & 0xffffu);
    const int rb_next = (int)((unsigned)tn >> 8), cj_next = tn & 255;
    if (wave == 0) {                    // the tile after the next one, for everybody's next round
        walk_t tw;
        walk_take(twl, tw);
        tw.step(a, bands);
        if (lane == 0) { tseq[tpar ^ 1] = tw.word(a); tband[tpar ^ 1] = tw.bword(); }
        walk_park(twl, tw, lane);
    }
    tpar ^= 1;
    if (empty_tile) {
        if (pre_next) {
            tile_load_s<BALF64>(a, bd_next, rb_next, cj_next, wave, lane, nxt);
            if (!BALF64 && wave < 3) {
                const int tix = wave * 64 + lane;
                if (tix < LC) wct[tix] = nxt.wcol == nxt.wcol ? nxt.wcol : 0.0;
            }
        }
        fetched = pre_next;
        have = have_next; rb = rb_next; cj = cj_next; bw = bw_next;
        __syncthreads();                // (rare: the far end of the chromosome) wave 0's word before it is read
        if (pub >= 0) { if (wave == NW - 1) publish_hist(bands + (pub >> 1), pub & 1); pub = -1; }
        tnext = lds_u32(lds0 + (unsigned)((unsigned char*)tseq - smem) + (unsigned)tpar * 4u);
        bnext = lds_u32(lds0 + (unsigned)((unsigned char*)tband - smem) + (unsigned)tpar * 4u);
        continue;
    }
    unsigned* __restrict__ tcnt = tcount + par;
#if defined(HPK_CLK_P1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (phase clocks: the wait for the prefetched rows on its own, booked under slot 4)
    HPK_CLK(ck4)
#endif
    // ---- phase 1 (rows): balanced values and packed cells of the lane's ten cells, the candidates among them, and the row
    // prefix of both planes - nine adds inside the lane, a scan over the 16 lanes of the DPP row that holds the table row -
    // written straight to the tables.  Four table rows per wave, all 64 in one pass.
    {
    const int Y = 4 * wave + (lane >> 4);                               // the lane's table row
    const int XH = (LC - 1) - 10 * (lane & 15);                         // SAT column of the lane's cell e = 0 (cell e: XH - e)
    const int rr = r0 - W - 1 + Y;                                      // its matrix row
    const int y = Y - (W + 1);                                          // its row of the output tile
    const int kH = mw + cj * TC + 1 + XH - Y;                           // diagonal of cell e = 0 (cell e: kH - e)
    const int xo = XH - W;                                              // output column of cell e = 0 (cell e: xo - e)
    // the row's stored elements: diagonals [0, lim) - inside the stored diagonals and left of the matrix end (column r + k < n);
    // rows above the matrix or beyond its end have none
    int lim = n - rr;
    lim = lim < bnum ? lim : bnum;
    lim = rr >= 0 ? lim : 0;
    lim = lim > 0 ? lim : 0;
    // Candidates: cells with a count, inside the tile's columns (0 <= xo - e < TC), on a row of the output tile (rows at or
    // beyond n read 0) and on a diagonal that holds band pixels (0 <= kH - e - mw <= Dm - mw): a range of e per lane -
    // bit 9 - e of cmask - times the cells' own "count != 0" bits.
    unsigned cmask;
    {
        const int A = kH - mw;
        int elo = A - (Dm - mw), ehi = A < xo ? A : xo;
        elo = elo > xo - TC + 1 ? elo : xo - TC + 1;
        elo = elo > 0 ? elo : 0;
        ehi = ehi < 9 ? ehi : 9;
        const unsigned m = ((2u << (9 - elo)) - 1u) & ~((1u << (9 - ehi)) - 1u);        // bits 9 - ehi .. 9 - elo
        cmask = ((unsigned)y < (unsigned)TR && elo <= ehi) ? m : 0u;
    }
    // weights of masked bins (NaN, scripts/pyHICCUPS:163-166) count as 0: their pixels' balanced values are the zeros the
    // reference turns its NaNs into (pyHICCUPS:157), everything else - negative weights included - is the reference's product.
    // Column weights: the tile's LDS table (NaNs already 0), memory order like the prefetched elements.
    double wr = 0.0, wcm[10];
    if (!BALF64) {
        wr = nxt.wrow == nxt.wrow ? nxt.wrow : 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double2 w2 = *reinterpret_cast<const double2*>(&wct[XH - 9 + 2 * i]);
            wcm[2 * i] = w2.x; wcm[2 * i + 1] = w2.y;
        }
    }
    const unsigned span = lim > mw ? (unsigned)(lim - mw) : 0u;         // f64 input: balanced values exist on diagonals [mw, lim)
    double bv[10];
    unsigned pk[10];
    unsigned cm = 0u;                                                   // cells with a count: bit 9 - e
    // Tiles inside the band - every cell of every lane of the wave a stored element with a balanced value - skip the two
    // range selects per cell.
#define HPK_CELLS(MASKED)                                                                                                      \
    _Pragma("unroll") for (int e = 0; e < 10; ++e) {                                                                           \
        const int i = 9 - e;                                                                                                   \
        const int k = kH - e, km = k - mw;                              /* diagonal, diagonal - min(ww) */                     \
        unsigned rb32 = nxt.raw[i];                                                                                            \
        if (MASKED) rb32 = (unsigned)k < (unsigned)lim ? rb32 : 0u;                                                            \
        float rv = __uint_as_float(rb32);                                                                                      \
        const unsigned ru = (unsigned)rv;                                                                                      \
        const unsigned rc = ru < pkcap_p ? ru : pkcap_p;                                                                       \
        if (BALF64) {                                                                                                          \
            /* as given: the caller zeroed the NaNs (hpk.h), signs are kept (callers.py:78) */                                 \
            bv[e] = (!(MASKED) || (unsigned)km < span) ? nxt.bal[i] : 0.0;                                                     \
        } else {                                                                                                               \
            if (MASKED) {                                                                                                      \
                rv = km >= 0 ? rv : 0.f;                                /* balanced values exist from diagonal min(ww) on */   \
                asm volatile("" : "+v"(rv));                            /* (select on the f32, not on the converted f64) */    \
            }                                                                                                                  \
            /* (raw * w_r) * w_c with NaN weights as 0: NaN -> 0, signs are kept.  Two rounded products, as numpy forms */      \
            /* them (scripts/pyHICCUPS:150-152): not to be contracted with the prefix adds that follow */                       \
            bv[e] = ((double)rv * wr) * wcm[i];                                                                                \
            asm volatile("" : "+v"(bv[e]));                                                                                    \
        }                                                                                                                      \
        pk[e] = rc | (bv[e] != 0.0 ? 1u << PK_SHIFT : 0u);                                                                     \
        /* cm = 2 cm + (count != 0): a compare and an add with carry */                                                        \
        asm("v_cmp_ne_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(cm) : "v"(ru) : "vcc");             \
    }
    {
        const bool inner = (kH - 9 >= mw) & (kH < lim);
        if (ballot64(!inner) == 0ull) { HPK_CELLS(false) }
        else { HPK_CELLS(true) }
    }
#undef HPK_CELLS
    cm &= cmask;
    // the lane's slice of the tile-wide list: behind the candidates of the lanes before it, in the wave's slice
    const unsigned cnt = (unsigned)__popc(cm);
    const unsigned inc = wave_inclusive_scan(cnt);
    const unsigned nrow = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
    unsigned slot = 0u;
    {
        // hipcc's atomic optimiser would wait for the returned value on the spot; issued by hand, the LDS round
        // trip runs beside the f64 prefix.  All lanes would add the same count into the same word: lane 0 only, by
        // narrowing exec around the instruction (every lane is active here) instead of a branch on a lane mask.
        const unsigned addr = (unsigned)(size_t)tcnt;
        asm volatile("s_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %1, %2\n\ts_mov_b64 exec, -1"
                     : "=&v"(slot) : "v"(addr), "v"(nrow) : "memory");
    }
    // row prefix of the f64 plane (cell e = 0 first: from the origin side)
#pragma unroll
    for (int e = 1; e < 10; ++e) bv[e] += bv[e - 1];
    {
        const double ex = row16_exclusive_scan(bv[9]);
        double* __restrict__ dst = &Sc[Y * LC + XH - 9];
#pragma unroll
        for (int i = 0; i < 5; ++i)
            *reinterpret_cast<double2*>(&dst[2 * i]) = make_double2(bv[9 - 2 * i] + ex, bv[8 - 2 * i] + ex);
    }
    if (nrow != 0u) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(slot) :: "memory");
        // entries in the order of the rows, inside a row by descending column
        const unsigned at0 = (unsigned)__builtin_amdgcn_readfirstlane((int)slot) + (inc - cnt);
        const unsigned ebase = (unsigned)xo | ((unsigned)y << HPK_ENT_YSHIFT);
#pragma unroll
        for (int e = 0; e < 10; ++e) {
            const bool cd = ((cm >> (9 - e)) & 1u) != 0u;
            const unsigned at = at0 + (unsigned)__popc(cm >> (10 - e));        // (e = 0: cm has ten bits)
            if (cd) lds_st_u32(lds0 + (unsigned)(LR * LC * 12) + at * 4u, (ebase - (unsigned)e) | ((pk[e] & PK_MASK) << HPK_ENT_CNT_SHIFT));
        }
    }
    // ... and of the packed plane
#pragma unroll
    for (int e = 1; e < 10; ++e) pk[e] += pk[e - 1];
    {
        const unsigned ex = row16_exclusive_scan(pk[9]);
        unsigned* __restrict__ dst = &Sp[Y * LC + XH - 9];
#pragma unroll
        for (int i = 0; i < 5; ++i)
            *reinterpret_cast<uint2*>(&dst[2 * i]) = make_uint2(pk[9 - 2 * i] + ex, pk[8 - 2 * i] + ex);
    }
    }
    HPK_CLK(ck0)
    // The next tile's rows start moving now, from every wave.
    if (pre_next) tile_load_s<BALF64>(a, bd_next, rb_next, cj_next, wave, lane, nxt);
    fetched = pre_next;
    __syncthreads();
    if (pub >= 0) { if (wave == NW - 1) publish_hist(bands + (pub >> 1), pub & 1); pub = -1; }      // (the band before: see flush_hist)
    HPK_CLK(ck1)
    // ---- phase 2 (columns): the tables hold row prefixes; the prefix down the columns runs through LDS.  f64 plane: waves
    // 0-9, a thread per column and chunk of 16 rows; packed plane: waves 10-14, chunks of 32 rows.  A thread sums its chunk in
    // registers, parks the chunk's total, and - behind a barrier - writes its cells back with the totals of the chunks above.
    unsigned creg[32];
    {
        if (wave < 10) {
            const int tix = wave * 64 + lane;
            const int ch = tix >= 2 * LC ? (tix >= 3 * LC ? 3 : 2) : (tix >= LC ? 1 : 0), col = tix - ch * LC;
            double v[16];
            // (single ds_read_b64: the ds_read2_b64 the compiler pairs two rows into takes 8 LDS cycles per KiB, two singles 4 -
            // MI355X_MICROARCH.md; round 6: -1.7 % of the kernel on the default workload)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lds_f64(lds0 + (unsigned)((16 * ch) * LC + col) * 8u + (unsigned)i * (unsigned)(LC * 8));
#pragma unroll
            for (int i = 1; i < 16; ++i) v[i] += v[i - 1];
            if (ch < 3) ctot[ch * LC + col] = v[15];
#pragma unroll
            for (int i = 0; i < 16; ++i) { creg[2 * i] = (unsigned)__double2loint(v[i]); creg[2 * i + 1] = (unsigned)__double2hiint(v[i]); }
        } else {
            // (wave 15 has no chunk: it reads along with wave 14 and writes nothing - every path defines the registers)
            const int tix = ((wave < 15 ? wave : 14) - 10) * 64 + lane;
            const int ch = tix >= LC ? 1 : 0, col = tix - ch * LC;
            const unsigned* __restrict__ src = &Sp[(32 * ch) * LC + col];
#pragma unroll
            for (int i = 0; i < 32; ++i) creg[i] = src[i * LC];
#pragma unroll
            for (int i = 1; i < 32; ++i) creg[i] += creg[i - 1];
            if (ch == 0 && wave < 15) utot[col] = creg[31];
        }
    }
    __syncthreads();
    HPK_CLK(ck2)
    {
        if (wave < 10) {
            const int tix = wave * 64 + lane;
            const int ch = tix >= 2 * LC ? (tix >= 3 * LC ? 3 : 2) : (tix >= LC ? 1 : 0), col = tix - ch * LC;
            // (plain sums in chunk order)
            double base = ch >= 1 ? ctot[col] : 0.0;
            base += ch >= 2 ? ctot[LC + col] : 0.0;
            base += ch >= 3 ? ctot[2 * LC + col] : 0.0;
            double* __restrict__ dst = &Sc[(16 * ch) * LC + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[i * LC] = __hiloint2double((int)creg[2 * i + 1], (int)creg[2 * i]) + base;
        } else if (wave < 15) {
            const int tix = (wave - 10) * 64 + lane;
            const int ch = tix >= LC ? 1 : 0, col = tix - ch * LC;
            const unsigned base = ch == 1 ? utot[col] : 0u;
            unsigned* __restrict__ dst = &Sp[(32 * ch) * LC + col];
#pragma unroll
            for (int i = 0; i < 32; ++i) dst[i * LC] = creg[i] + base;
        }
    }
    __syncthreads();
    // the next tile's column weights are in: into the LDS table (nobody reads it before the barrier that ends this tile)
    if (!BALF64 && pre_next && wave < 3) {
        const int tix = wave * 64 + lane;
        if (tix < LC) wct[tix] = nxt.wcol == nxt.wcol ? nxt.wcol : 0.0;
    }
    HPK_CLK(ck3)
    const int total = a.dbg_stop == 2 ? 0 : (int)*tcnt;         // (profiling ablation 2: the tables only, no batches)
    // gap rows (callers.py:238): rows of the tile's columns (the last tile of a row block: up to the end of its halo)
    // without a non-zero balanced value - exact on the valid-count field
    const int tx = wave * 64 + lane;             // (not threadIdx.x: its address arithmetic would be hoisted and spilled)
    if (tx < TR && r0 + tx < n) {
        const bool last = (cj == J_p - 1) || (c0 + TC >= n) || (mw + (cj + 1) * TC - (TR - 1)) > Dg_p;
        const int Y = tx + W + 1;
        unsigned rs = Sp[Y * LC + W] - Sp[(Y - 1) * LC + W];
        const int xe = last ? W : W + TC;        // last: nothing is taken off (the two reads below cancel)
        rs -= last ? 0u : Sp[Y * LC + xe] - Sp[(Y - 1) * LC + xe];
        if ((rs >> PK_SHIFT) != 0u) gptr(bd->gap)[r0 + tx] = 1;
    }
    if (tx == 0) { tcount[par ^ 1] = 0u; tcount[4 + (par ^ 1)] = 0u; tcount[6 + (par ^ 1)] = 0u; }    // the next tile's counters (its atomics start after the barrier below)
    // ---- phase 3: batches of 64 candidates, dealt round-robin to the waves
    const int64_t tbase = (int64_t)tid * tilecap_p;
    unsigned* __restrict__ ent_t = gptr(bd->rec_ent) + tbase;
    HPK_CLK(ck4)
#if HPK_DYN_BATCH
    // the waves' first batches are their own; the others are dealt as the waves come free (a counter in LDS: batches differ -
    // widths in between, sums redone exactly, records or none - and the tile's last barrier waits for the slowest wave)
    auto next_batch = [&]() {
        unsigned v = 0u;
        if (lane == 0) v = atomicAdd(&tcount[6 + par], 1u);
        return NW + __builtin_amdgcn_readfirstlane((int)v);
    };
#pragma unroll 1
    for (int b = wave; b * 64 < total; b = next_batch()) {
#else
#pragma unroll 1
    for (int b = wave; b * 64 < total; b += NW) {
#endif
#if defined(HPK_PHASE_CLOCK) && !defined(HPK_WG_LIFE)
        ck7 += 1ull;
#endif
        const int i = b * 64 + lane;
        const bool cand = i < total;
        const unsigned id = lst[cand ? i : 0];
        const int x = (int)HPK_ENT_X(id);
        const int y = (int)HPK_ENT_Y(id);
        const int base = (y + W + 1) * LC + W + x;
        // one round of reads: P(Y, X) of both planes, the pixel's own value, the three Reads boxes that decide most
        // candidates (p0: subtracted from all, narrowest, widest), the largest corner of the widest window
        const unsigned cb = lds0 + (unsigned)base * 8u, pb = lds0 + (unsigned)(LR * LC * 8) + (unsigned)base * 4u;
        const unsigned sr = lds_u32(pb);
        const bool diffp = SINGLE && sp_p > 0;                      // Box(w*) - Box(p): see box_ky_d
        const double sc = lds_f64(cb);
        double pixc = 0.0, amax = 0.0;
        if (SINGLE && !diffp) {
            pixc = (sc - lds_f64(cb + 8)) - (lds_f64(cb - LC * 8) - lds_f64(cb - LC * 8 + 8));
            amax = lds_f64(cb + (unsigned)(W * (LC - 1) * 8));
        }
        int wstar = 255;
        unsigned sstar = 0xffffffffu;           // general plans: resolving step per slot, 8 bits each (0xff = none)
        if (generic_p) {
            const unsigned alldone = (1u << nslots_p) - 1u;
            unsigned done = cand ? 0u : alldone;
            int cur_rid = -1;
            unsigned reads = 0u;
#pragma unroll 1
            for (int s2 = 0; s2 < nsteps; ++s2) {
                const unsigned w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)pl[s2 * 8]);
                const int slot = (int)(w0 & 3u), rid = (int)((w0 >> 10) & 63u), nrt = (int)((w0 >> 16) & 15u);
                const bool need = ((done >> slot) & 1u) == 0u;
                if (ballot64(need) == 0ull) continue;
                if (rid != cur_rid) {                   // the step's Reads matrix: sum of its lower-left box terms
                    cur_rid = rid;
                    const unsigned long long rt = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)pl[s2 * 8 + 5]) |
                                                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)pl[s2 * 8 + 6]) << 32;
                    unsigned acc = 0u;
                    for (int j = 0; j < nrt; ++j) {
                        const unsigned t = (unsigned)(rt >> (16 * j)) & 0xffffu;
                        acc += (unsigned)(int)(signed char)(t >> 8) * reads_box_b(pb, (int)(t & 0xffu), sr);
                    }
                    reads = acc & PK_MASK;
                }
                const bool hit = need & (reads >= (unsigned)minr_p);
                const unsigned long long hm = ballot64(hit);
                if (lane == s2) myhist += (unsigned)__popcll(hm);
                if (hit) {
                    sstar = (sstar & ~(0xffu << (8 * slot))) | ((unsigned)s2 << (8 * slot));
                    done |= 1u << slot;
                }
                if (ballot64(done != alldone) == 0ull) break;
            }
            wstar = (cand & (sstar != 0xffffffffu)) ? wmin_p : 255;     // (resolved in some slot: gets a record whatever the bound)
        } else {
        const unsigned b0 = (p0_p > 0) ? reads_box_b(pb, p0_p, sr) : 0u;
        const unsigned bf = reads_box_b(pb, wmin_p, sr);
        const unsigned bl = reads_box_b(pb, W, sr);
        wstar = (cand & (bl - b0 >= (unsigned)minr_p)) ? W : wstar;
        wstar = (cand & (bf - b0 >= (unsigned)minr_p)) ? wmin_p : wstar;
        // lanes that pass the widest but not the narrowest box: all widths in between, four at a time (Reads is monotone)
        if (W - wmin_p > 1 && ballot64(wstar == W) != 0ull) {
#pragma unroll 1
            for (int wa = wmin_p + 1; wa < W; wa += 4) {
                // widths wa .. wa + 3 from three addresses (those at or beyond W read rows the list follows: not used)
                const unsigned dn = (unsigned)wa * (unsigned)(LC * 4), lf = (unsigned)wa * 4u;
                unsigned a1 = pb + dn, a2 = pb + (dn - lf), a3 = pb - (lf + 12u);
                asm volatile("" : "+v"(a1), "+v"(a2), "+v"(a3));     // (whole addresses: the steps below fit the reads' immediates)
                unsigned rd[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    rd[t] = (lds_u32(a2 + t * (LC * 4 - 4)) - lds_u32(a1 + t * (LC * 4)) - lds_u32(a3 + (3 - t) * 4) + sr) & PK_MASK;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    wstar = ((wa + t < W) & (wstar == W) & (rd[t] - b0 >= (unsigned)minr_p)) ? wa + t : wstar;
            }
        }
        // resolve histogram by width.  The first eight widths are counted per lane, 16 bits each in two registers
        // (a lane sees at most one candidate per batch; the walk leaves a band's segment before 65 535 batches of this
        // wave: TileWalk::bword), and added up over the wave when the walk leaves the band; wider ones (maxww >= min(ww) + 8)
        // by one ballot per width.
        // (The f64-input variants are out of registers - their prefetch holds 30 instead of 16 - and keep the ballots.)
        if (!BALF64) {
            // (a tile out of the redo queue does not count: the lean kernel did)
            const unsigned off = is_redo ? 255u : (unsigned)(wstar - wmin_p);     // 255 - min(ww) >= 8 for "no sufficient width"
            const unsigned long long inc = 1ull << ((off & 3u) * 16u);
            hpack0 += off < 4u ? inc : 0ull;
            hpack1 += (off - 4u) < 4u ? inc : 0ull;
            if (W - wmin_p >= 8 && ballot64((off >= 8u) & (off != 255u) & (wstar != 255)) != 0ull) {
#pragma unroll 1
                for (int w = wmin_p + 8; w <= W; ++w) {
                    const unsigned c = (unsigned)__popcll(ballot64(wstar == w));
                    if (lane == w) myhist += c;
                }
            }
        } else {
            const unsigned long long mf = ballot64(wstar == wmin_p);
            if (lane == wmin_p) myhist += (unsigned)__popcll(mf);
            if (ballot64((wstar != wmin_p) & (wstar != 255)) != 0ull) {
#pragma unroll 1
                for (int w = wmin_p + 1; w <= W; ++w) {
                    const unsigned c = (unsigned)__popcll(ballot64(wstar == w));
                    if (lane == w) myhist += c;
                }
            }
        }
        if (W > planw_p) wstar = wstar > planw_p ? 255 : wstar;        // (a halo beyond maxww: no step at those widths)
        }   // monotone Reads
        // The candidates that get a record: a batch without any is done here; the others take their (packed) places in the
        // tile's record region from the tile's counter - one LDS atomic per batch, lane 0 (see phase 1), whose return is
        // waited for after the box sums.
        const bool live = cand & (wstar <= wg_p);
        const unsigned long long lm = ballot64(live);
        if (lm == 0ull) continue;
        unsigned rslot = 0u;
        {
            const unsigned addr = (unsigned)(size_t)(tcount + 4 + par);
            const unsigned nl = (unsigned)__popcll(lm);
            asm volatile("s_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %1, %2\n\ts_mov_b64 exec, -1"
                         : "=&v"(rslot) : "v"(addr), "v"(nl) : "memory");
        }
        wstar = live ? wstar : 255;                 // (candidates beyond the bound count as unresolved from here on)
        // general plans: the innermost box is the same in every step of every slot - formed once per candidate
        double kc0 = 0.0, yc0 = 0.0, big0 = 0.0;
        if (!SINGLE && fr_p > 0) box_ky_d(cb, fr_p, sc, kc0, yc0, big0);
        // ---- sums at the resolving step, once per slot
        unsigned ri = 0u;
#pragma unroll 1
        for (int q = 0; q < (SINGLE ? 1 : nslots_p); ++q) {
            int sq;
            if (SINGLE) sq = wstar == 255 ? 0xff : wstar - wmin_p;      // textbook plan: one step per width, in order
            else {
                const int wf = (q == 0) ? wf_q[0] : (q == 1) ? wf_q[1] : (q == 2) ? wf_q[2] : wf_q[3];
                const int wq = wstar > wf ? wstar : wf;
                sq = (int)stepof[q * 32 + (wq & 31)];
                sq = ((wstar == 255) | (wq > wg_p)) ? 0xff : sq;      // (a slot whose own first width lies beyond the bound)
                if (generic_p) sq = wstar == 255 ? 0xff : (int)((sstar >> (8 * q)) & 0xffu);
            }
            const bool act = sq != 0xff;
            double SK = 0.0, SY = 0.0;
            if (ballot64(act) != 0ull) {
                unsigned w0 = 0u, k0 = 0u, k1 = 0u, k2 = 0u, k3 = 0u;
                int nkt = 0, rho_min;
                if (SINGLE) {
                    // Box(w*) - Box(p): the outer radius per lane, the inner one the same for all
                    double kcw, ycw, kcp = 0.0, ycp = 0.0;
                    if (diffp) {
                        double bigp;
                        box_ky_d(cb, act ? wstar : 1, sc, kcw, ycw, amax);
                        box_ky_d(cb, sp_p, sc, kcp, ycp, bigp);
                    } else box_ky_b(cb, act ? wstar : 1, pixc, sc, kcw, ycw);
                    SK = act ? kcw - kcp : 0.0;
                    SY = act ? ycw - ycp : 0.0;
                    rho_min = sp_p + 1;
                } else {
                    double cfs = 0.0;
                    amax = 0.0;
                    const int src = act ? sq : 0;
                    const uint4 pw4 = *reinterpret_cast<const uint4*>(&pl[src * 8]);
                    w0 = pw4.x; k0 = pw4.y; k1 = pw4.z; k2 = pw4.w;
                    if (maxnkt > 6) k3 = pl[src * 8 + 4];
                    nkt = act ? (int)((w0 >> 20) & 15u) : 0;
                    int j0 = 0;
                    if (fr_p > 0) {                         // term 0 of every step: the shared box
                        const double cf = nkt > 0 ? (double)(int)(signed char)((k0 >> 8) & 0xffu) : 0.0;
                        SK = cf * kc0; SY = cf * yc0;
                        cfs = cf;
                        amax = nkt > 0 ? big0 : 0.0;
                        j0 = 1;
                    }
#pragma unroll 1
                    for (int j = j0; j < maxnkt; ++j) {
                        const bool on = j < nkt;
                        if (ballot64(on) == 0ull) break;
                        const unsigned kw = (j < 2) ? k0 : (j < 4) ? k1 : (j < 6) ? k2 : k3;
                        const unsigned t = (kw >> (16 * (j & 1))) & 0xffffu;
                        const int rho = on ? (int)(t & 0xffu) : 1;      // idle lanes read a harmless box
                        const double cf = on ? (double)(int)(signed char)(t >> 8) : 0.0;
                        double kc, yc, big;
                        box_ky_d(cb, rho, sc, kc, yc, big);
                        SK += cf * kc; SY += cf * yc;
                        cfs += cf;
                        amax = on ? fmax(amax, big) : amax;
                    }
                    // (box_ky_d) a step whose coefficients do not cancel still needs the pixel's own value
                    if (ballot64(cfs != 0.0) != 0ull)
                        SK += cfs * ((sc - lds_f64(cb + 8)) - (lds_f64(cb - LC * 8) - lds_f64(cb - LC * 8 + 8)));
                    rho_min = (int)((w0 >> 24) & 31u);
                }
                // lower-left support off the band: exact 0 (see hpk_stencil)
                const int d = c0 + x - (r0 + y);
                SY = (d - rho_min - 1 < mw) ? 0.0 : SY;
                // Sums that are small against the largest corner of the window's table entries carry that corner's
                // rounding noise (relative error of the sum ~ 1e-15 x corner / sum): below a.risk of it they are redone
                // exactly - 0 when no contributing cell is non-zero (valid-count plane), otherwise by adding the window
                // cells themselves (a.risk = 2^-12: worst-case relative error of what stays on the table ~ 5e-14 x 2^12 = 2e-10,
                // typically two orders below).
                const double thr = amax * a.risk;
                const bool risky = act & ((SK < thr) | ((SY < thr) & (SY != 0.0)));
                if (ballot64(risky) != 0ull) {
#if defined(HPK_PHASE_CLOCK) && !defined(HPK_WG_LIFE)
                    ck6 += (unsigned long long)__popcll(ballot64(risky && SK > 0.0)) << 40;
#endif
                    if (risky) {
                        const unsigned pv = sr - Sp[base + 1] - Sp[base - LC] + Sp[base - LC + 1];
                        unsigned VK = 0u, VY = 0u;
                        if (SINGLE) {
                            const unsigned long long vw = box_ky_valid_m(Sp, base, wstar, pv, sr);
                            const unsigned long long vp = sp_p > 0 ? box_ky_valid_m(Sp, base, sp_p, pv, sr) : 0ull;
                            VK = (unsigned)vw - (unsigned)vp;
                            VY = (unsigned)(vw >> 32) - (unsigned)(vp >> 32);
                        } else {
#pragma unroll 1
                            for (int j = 0; j < nkt; ++j) {
                                const unsigned kw = (j < 2) ? k0 : (j < 4) ? k1 : (j < 6) ? k2 : k3;
                                const unsigned t = (kw >> (16 * (j & 1))) & 0xffffu;
                                const unsigned long long kyv = box_ky_valid_m(Sp, base, (int)(t & 0xffu), pv, sr);
                                VK += (unsigned)(int)(signed char)(t >> 8) * (unsigned)kyv;
                                VY += (unsigned)(int)(signed char)(t >> 8) * (unsigned)(kyv >> 32);
                            }
                        }
                        if (VK == 0u) { SK = 0.0; SY = 0.0; }
                        else if (VY == 0u) SY = 0.0;
                    }
                    // what is left has non-zero cells: one pixel at a time, the whole wave on it
                    unsigned long long todo = ballot64(risky && SK != 0.0 && ((SK < thr) | ((SY < thr) & (SY != 0.0))));
#if defined(HPK_PHASE_CLOCK) && !defined(HPK_WG_LIFE)
                    ck7 += (unsigned long long)__popcll(todo) << 40;
#endif
                    while (todo != 0ull) {
                        const int src = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        const int er = __builtin_amdgcn_readlane(r0 + y, src), ec = __builtin_amdgcn_readlane(c0 + x, src);
                        const int es = __builtin_amdgcn_readlane(sq, src);
                        const double2 ex = explicit_sums_wave(gptr(bd->raw), gptr(bd->bal), gptr(bd->weight), plan->steps[es].m, W, er, ec, n, bd->num, bd->ld, mw, lane);
                        if (lane == src) { SK = ex.x; SY = (SY == 0.0) ? 0.0 : ex.y; }   // SY == 0: exact by construction or no non-zero cell
                    }
                }
            }
            if (q == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rslot) :: "memory");
                ri = (unsigned)__builtin_amdgcn_readfirstlane((int)rslot) +
                     __builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
#ifdef HPK_ABLATE
                if (live && a.dbg_stop != 3) ent_t[ri] = id;
#else
                if (live) ent_t[ri] = id;
#endif
            }
#ifdef HPK_ABLATE
            if (live && a.dbg_stop != 3) {           // (ablation 3: everything but the record stores)
#else
            if (live) {
#endif
                const int64_t o = q * bd->rec_stride + tbase + ri;
                gptr(bd->rec_S)[o] = make_double2(act ? SK : 0.0, act ? SY : 0.0);
                gptr(bd->rec_W)[o] = act ? (uint8_t)(sq + 1) : (uint8_t)0;
            }
        }
    }
    HPK_CLK(ck5)
    // (written by wave 0 at the top of this round, three barriers ago: in flight across the barrier below)
    tnext = lds_u32(lds0 + (unsigned)((unsigned char*)tseq - smem) + (unsigned)tpar * 4u);
    bnext = lds_u32(lds0 + (unsigned)((unsigned char*)tband - smem) + (unsigned)tpar * 4u);
    __syncthreads();                 // every wave is done with this tile's SAT and list
    HPK_CLK(ck6)
    if (wave == 0) {
        // scoring work list: one entry per HPK_UNIT records.  The slot reservation (a returning atomic on one global
        // counter) of this tile is only consumed when the next tile ends.
        if (pend_tid >= 0) {
            const unsigned nu = (pend_c + (unsigned)HPK_UNIT - 1u) / (unsigned)HPK_UNIT;
            const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane((int)pend_off);
            if ((unsigned)lane < nu) gptr(bands[pend_band].units)[off + lane] = make_uint2(pend_rc, (unsigned)lane | (pend_c << 8));
        }
        pend_tid = -1;
        const unsigned nrec = (unsigned)__builtin_amdgcn_readfirstlane((int)tcount[4 + par]);    // records of this tile
        if (nrec > 0u) {
            pend_tid = tid;
            pend_rc = (unsigned)rb << 8 | (unsigned)cj;
            pend_band = band;
            pend_c = nrec;
            if (lane == 0) pend_off = atomicAdd(reinterpret_cast<unsigned*>(gptr(bd->small) + HPK_OFF_NUNITS), (pend_c + (unsigned)HPK_UNIT - 1u) / (unsigned)HPK_UNIT);
        }
        if (lane == 0) { gptr(bd->tile_cnt)[tid] = nrec; mycand += is_redo ? 0u : (unsigned)total; }
    }
    have = have_next; rb = rb_next; cj = cj_next; bw = bw_next;
    par ^= 1;
    } while (have && bw == cbw);   // tile loop of the band
    }   // bands
    if (hband >= 0) {
        flush_hist(hp);
        __syncthreads();
        if (wave_k == NW - 1) publish_hist(bands + hband, hp);
    }
    const int lane = lane_k, wave = wave_k;
    if (wave == 0 && pend_tid >= 0) {
        const unsigned nu = (pend_c + (unsigned)HPK_UNIT - 1u) / (unsigned)HPK_UNIT;
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane((int)pend_off);
        if ((unsigned)lane < nu) gptr(bands[pend_band].units)[off + lane] = make_uint2(pend_rc, (unsigned)lane | (pend_c << 8));
    }
#ifdef HPK_PHASE_CLOCK
    if (a.clk && lane == 0) {
        unsigned long long* o = a.clk + ((size_t)blockIdx.x * NW + wave) * 8;
#ifdef HPK_WG_LIFE
        ck7 = __builtin_amdgcn_s_memrealtime();
#endif
        o[0] = ck0; o[1] = ck1; o[2] = ck2; o[3] = ck3; o[4] = ck4; o[5] = ck5; o[6] = ck6; o[7] = ck7;
    }
#endif
}

// The lean kernel's loads of a tile: the lane's ten band elements (as tile_load_s), the non-zero flags of its ten column weights
// (bit i: column XH - 9 + i) and of its row weight, out of the band's bit mask (HpkBandDesc::off_wnz).
__device__ __forceinline__ void tile_load_lean(const HpkStencilArgs& a, const HpkBandDesc* __restrict__ bd, int rb, int cj, int wave, int lane,
                                               unsigned (&raw)[10], unsigned& wbits, bool& wr_nz) {
    const int bn = bd->n;
    const int64_t bld = bd->ld;
    const int W = bd->W;
    const int r0 = rb * bd->TR;
    const int rt0 = r0 - W - 1;
    const int rb0 = rt0 > 1 ? rt0 - 1 : 0;
    int rows = bn - rb0;
    rows = rows > LR + 2 ? LR + 2 : rows;
    const unsigned ldu = (unsigned)bld;
    const int koff = a.mw + cj * bd->TC + 1;
    const int Y = 4 * wave + (lane >> 4);
    const int XL = (LC - 10) - 10 * (lane & 15);
    const rsrc_t rraw = make_rsrc(gptr(bd->raw) + (int64_t)rb0 * bld, (unsigned)rows * ldu * 4u);
    const int flat = (rt0 + Y - rb0) * (int)ldu + (koff + XL - Y);
    const unsigned off = (unsigned)flat * 4u;
    const bool wide = ldu >= 16u && !(flat < 0 && flat > -10);
    if (wide) {
        const v4u32 q0 = ldbuf_v4(rraw, off), q1 = ldbuf_v4(rraw, off + 16u);
        const v2u32 q2 = ldbuf_v2(rraw, off + 32u);
        raw[0] = q0.x; raw[1] = q0.y; raw[2] = q0.z; raw[3] = q0.w;
        raw[4] = q1.x; raw[5] = q1.y; raw[6] = q1.z; raw[7] = q1.w;
        raw[8] = q2.x; raw[9] = q2.y;
    } else {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            unsigned o1 = off + 4u * (unsigned)i;
            asm volatile("" : "+v"(o1));
            raw[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rraw, (int)o1, 0, 0);
        }
    }
    const unsigned* __restrict__ wnz = reinterpret_cast<const unsigned*>(gptr(bd->small) + bd->off_wnz);
    const int gc = r0 + a.mw + cj * bd->TC - W + XL + HPK_WNZ_LEAD;      // bit of the lane's lowest column (matrix column c0 - W + X)
    const unsigned w0 = wnz[gc >> 5], w1 = wnz[(gc >> 5) + 1];
    wbits = (unsigned)((((unsigned long long)w1 << 32) | (unsigned long long)w0) >> (gc & 31));
    const int gr = rt0 + Y + HPK_WNZ_LEAD;
    wr_nz = ((wnz[gr >> 5] >> (gr & 31)) & 1u) != 0u;
}

// The same loads a tile ahead, without registers: buffer loads that write straight to LDS (`buffer_load ... lds`: the data of lane l
// lands at the wave's base + 16 l for 16 or 12 bytes per lane, + 4 l for 4: scripts/measure/ubench/lds_dma.hip).  A stage holds, per lane, the four pieces of its ten elements (4 + 4 + 1 + 1) and the three mask
// words (HPK_LST_*: byte offsets of the parts, each [16 waves][64 lanes]); two stages, so that a tile's loads are requested when the
// tile before it starts.  (wave-uniform LDS address: lds0-relative byte offset `stage`.)
#define HPK_LST_A 0
#define HPK_LST_B 16384
#define HPK_LST_C 32768
#define HPK_LST_C2 36864
#define HPK_LST_D 40960
#define HPK_LST_E 45056
#define HPK_LST_F 49152
#define HPK_LST_BYTES 53248
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void tile_issue_lean(const HpkStencilArgs& a, const HpkBandDesc* __restrict__ bd, int rb, int cj, int wave, int lane,
                                                unsigned char* stage) {
    const int bn = bd->n;
    const int64_t bld = bd->ld;
    const int W = bd->W;
    const int r0 = rb * bd->TR;
    const int rt0 = r0 - W - 1;
    const int rb0 = rt0 > 1 ? rt0 - 1 : 0;
    int rows = bn - rb0;
    rows = rows > LR + 2 ? LR + 2 : rows;
    const unsigned ldu = (unsigned)bld;
    const int koff = a.mw + cj * bd->TC + 1;
    const int Y = 4 * wave + (lane >> 4);
    const int XL = (LC - 10) - 10 * (lane & 15);
    const rsrc_t rraw = make_rsrc(gptr(bd->raw) + (int64_t)rb0 * bld, (unsigned)rows * ldu * 4u);
    const int flat = (rt0 + Y - rb0) * (int)ldu + (koff + XL - Y);
    const unsigned off = (unsigned)flat * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rraw, (lds_void_t*)(stage + HPK_LST_A + wave * 1024), 16, (int)off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rraw, (lds_void_t*)(stage + HPK_LST_B + wave * 1024), 16, (int)(off + 16u), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rraw, (lds_void_t*)(stage + HPK_LST_C + wave * 256), 4, (int)(off + 32u), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rraw, (lds_void_t*)(stage + HPK_LST_C2 + wave * 256), 4, (int)(off + 36u), 0, 0, 0);
    const rsrc_t rw = make_rsrc(gptr(bd->small) + bd->off_wnz, (unsigned)HPK_WNZ_WORDS(bn) * 4u);
    const int gc = r0 + a.mw + cj * bd->TC - W + XL + HPK_WNZ_LEAD;
    const int gr = rt0 + Y + HPK_WNZ_LEAD;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(stage + HPK_LST_D + wave * 256), 4, (int)((unsigned)(gc >> 5) * 4u), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(stage + HPK_LST_E + wave * 256), 4, (int)((unsigned)(gc >> 5) * 4u + 4u), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(stage + HPK_LST_F + wave * 256), 4, (int)((unsigned)(gr >> 5) * 4u), 0, 0, 0);
}
// ... and what the lane finds in a stage (the wave's own loads: waited for with vmcnt, no barrier).  Lanes whose ten elements would
// straddle the start of the band (tile_load_s: single loads) and bands narrower than 16 diagonals take the plain loads instead.
__device__ __forceinline__ void tile_take_lean(const HpkStencilArgs& a, const HpkBandDesc* __restrict__ bd, int rb, int cj, int wave, int lane,
                                               const unsigned char* stage, unsigned (&raw)[10], unsigned& wbits, bool& wr_nz) {
    const int W = bd->W;
    const int r0 = rb * bd->TR;
    const int rt0 = r0 - W - 1;
    const int rb0 = rt0 > 1 ? rt0 - 1 : 0;
    const unsigned ldu = (unsigned)bd->ld;
    const int koff = a.mw + cj * bd->TC + 1;
    const int Y = 4 * wave + (lane >> 4);
    const int XL = (LC - 10) - 10 * (lane & 15);
    const int flat = (rt0 + Y - rb0) * (int)ldu + (koff + XL - Y);
    const bool wide = ldu >= 16u && !(flat < 0 && flat > -10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ballot64(!wide) != 0ull) { tile_load_lean(a, bd, rb, cj, wave, lane, raw, wbits, wr_nz); return; }
    const int t = wave * 64 + lane;
    const uint4 q0 = *reinterpret_cast<const uint4*>(stage + HPK_LST_A + t * 16);
    const uint4 q1 = *reinterpret_cast<const uint4*>(stage + HPK_LST_B + t * 16);
    raw[0] = q0.x; raw[1] = q0.y; raw[2] = q0.z; raw[3] = q0.w;
    raw[4] = q1.x; raw[5] = q1.y; raw[6] = q1.z; raw[7] = q1.w;
    raw[8] = *reinterpret_cast<const unsigned*>(stage + HPK_LST_C + t * 4);
    raw[9] = *reinterpret_cast<const unsigned*>(stage + HPK_LST_C2 + t * 4);
    const unsigned w0 = *reinterpret_cast<const unsigned*>(stage + HPK_LST_D + t * 4), w1 = *reinterpret_cast<const unsigned*>(stage + HPK_LST_E + t * 4);
    const unsigned rw = *reinterpret_cast<const unsigned*>(stage + HPK_LST_F + t * 4);
    const int gc = r0 + a.mw + cj * bd->TC - W + XL + HPK_WNZ_LEAD;
    const int gr = rt0 + Y + HPK_WNZ_LEAD;
    wbits = (unsigned)((((unsigned long long)w1 << 32) | (unsigned long long)w0) >> (gc & 31));
    wr_nz = ((rw >> (gr & 31)) & 1u) != 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (in registers before the stage is handed to the tile after next)
}

// ------------------------------------------------------------------ the stencil kernel of the lean column chunks
// hpk_stencil_lean (weight input, plans with a monotone Reads matrix): far from the diagonal a tile holds candidates but next to
// none that resolves within the band's bound, and its f64 plane would be built for nothing.  The tiles of the column chunks
// from HpkBandDesc::lean_cj on (hpk_band_class: where the mean Reads of the chunk's nearest pixels stays far below
// min_local_reads) are this kernel's: it builds the packed plane only - no conversions, products, f64 scan, a third of the table
// traffic; the valid flags from "count, row weight and column weight all non-zero" (hpk_band_class rules out weights small
// enough for a product to underflow) - keeps no candidate list (every lane goes through its own candidates and tests the widest
// Reads box, what decides "resolves within the bound at all"), and the few candidates that pass get first sufficient width,
// resolve count and record as in hpk_stencil_s, with their sums formed cell by cell from the band (explicit_sums_wave).  A tile
// with more than a.lean_max candidates that count is given up: it goes to the redo queue, and hpk_stencil_s - which runs after
// this kernel - computes it in full.
template <bool SINGLE>
__global__ void __launch_bounds__(1024) hpk_stencil_lean(HpkStencilArgs a, const HpkBandDesc* __restrict__ bands) {
    constexpr int NW = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* __restrict__ Sp = reinterpret_cast<unsigned*>(smem);                      // [LR][LC] packed plane
    unsigned* __restrict__ utot3 = Sp + LR * LC;                                        // [3][LC] totals of the 16-row chunks
    unsigned* __restrict__ tcount = utot3 + 3 * LC;                                     // [32] counters, walk words, column-weight mask
    unsigned char* __restrict__ stepof = reinterpret_cast<unsigned char*>(tcount + 32);   // [HPK_KSLOTS][32]
    unsigned char* __restrict__ stage0 = stepof + HPK_KSLOTS * 32;                        // [2][HPK_LST_BYTES] the next tiles' rows on their way (tile_issue_lean)
    unsigned* __restrict__ twl = reinterpret_cast<unsigned*>(stage0 + 2 * HPK_LST_BYTES);     // [16] the tile walk's state between wave 0's steps
    // (the dynamic LDS of a kernel without static LDS starts at address 0: addresses formed from lds0 are immediates)
    constexpr unsigned lds0 = 0u;
    if ((unsigned)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();
    unsigned* __restrict__ tflag = tcount + 16;                       // [2] the tile met more candidates that count than lean_max
    unsigned* __restrict__ tseq = tcount + 8;
    unsigned* __restrict__ tband = tcount + 12;
    const int lane_k = threadIdx.x & 63;
    const int wave_k = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int mw = a.mw;
    const HpkDevPlan* __restrict__ plan = a.plan;
    const int nsteps = plan->nsteps, nslots = plan->nslots;
    if (wave_k == 0) {
        stepof[lane_k] = plan->step_of[lane_k >> 5][lane_k & 31];
        stepof[64 + lane_k] = plan->step_of[2 + (lane_k >> 5)][lane_k & 31];
        if (lane_k < 32) tcount[lane_k] = 0u;
    }
    int wf_q[HPK_KSLOTS];
#pragma unroll
    for (int q = 0; q < HPK_KSLOTS; ++q) wf_q[q] = __builtin_amdgcn_readfirstlane(plan->slot_wfirst[q]);
    const int nslots_p = __builtin_amdgcn_readfirstlane(nslots);
    const int minr_p = __builtin_amdgcn_readfirstlane(plan->min_reads), p0_p = __builtin_amdgcn_readfirstlane(plan->reads_p0);
    const int wmin_p = __builtin_amdgcn_readfirstlane(plan->wmin);
    const unsigned pkcap_p = (unsigned)__builtin_amdgcn_readfirstlane(plan->pk_cap);
    const int planw_p = __builtin_amdgcn_readfirstlane(plan->W);
    unsigned myhist = 0u;                 // lane w: candidates whose first sufficient width is w
    unsigned long long hpack0 = 0ull, hpack1 = 0ull;      // this lane's candidates by width min(ww) + k: 16 bits each, k = 0..3 | 4..7
    unsigned mycand = 0u, mylean = 0u, myexpl = 0u;       // candidates | tiles, of those given up << 16 (wave 0) | candidates summed cell by cell (per wave)
    auto fold_hpack = [&]() {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned v = (unsigned)((k < 4 ? hpack0 : hpack1) >> (16 * (k & 3))) & 0xffffu;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) v += (unsigned)__shfl_xor((int)v, m);
            if (lane_k == wmin_p + k) myhist += v;
        }
        hpack0 = 0ull; hpack1 = 0ull;
    };
    int pend_tid = -1, pend_band = 0;
    unsigned pend_c = 0u, pend_off = 0u, pend_rc = 0u;      // (pend_rc: the tile as the work list names it, row block << 8 | column chunk)
    int sp = 0;                         // the stage that holds the current tile's rows
    int par = 0;
    bool have;
    int rb, cj;
    unsigned bw;
    {
        TileWalk<1> tw;
        tw.init(a, bands);
        have = !tw.done; rb = tw.rbk; cj = tw.cj(a); bw = tw.bword();
        if (wave_k == 0) {
            tw.step(a, bands);
            if (lane_k == 0) { tseq[0] = tw.word(a); tband[0] = tw.bword(); }
            walk_park(twl, tw, lane_k);
        }
    }
    __syncthreads();
    unsigned tnext = lds_u32(lds0 + (unsigned)((unsigned char*)tseq - smem));
    unsigned bnext = lds_u32(lds0 + (unsigned)((unsigned char*)tband - smem));
    int tpar = 0;
    // the resolve counts of a band (segment) go to that band's totals when the walk leaves it (see hpk_stencil_s)
    auto flush_hist = [&](const HpkBandDesc* __restrict__ hb) {
        fold_hpack();
        unsigned* red = reinterpret_cast<unsigned*>(smem);
        int tix = wave_k * 64 + lane_k;
        asm volatile("" : "+v"(tix));
        __syncthreads();
        red[tix] = myhist;
        if (lane_k == 0) red[NW * 64 + wave_k] = mycand;
        __syncthreads();
        if (tix < 64) {
            unsigned tot = 0u;
            for (int w2 = 0; w2 < NW; ++w2) tot += red[w2 * 64 + tix];
            red[(NW + 1) * 64 + tix] = tot;
        }
        __syncthreads();
        if (tix <= HPK_MAX_STEPS) {
            unsigned out = 0u;
            const unsigned* hw = red + (NW + 1) * 64;
            if (tix < nsteps) {
                const HpkDevStep& st = plan->steps[tix];
                const int wf = plan->slot_wfirst[st.slot];
                if (st.wi > wf) out = hw[st.wi];
                else for (int w = 0; w <= wf && w < 64; ++w) out += hw[w];
            } else if (tix == HPK_MAX_STEPS) out = red[NW * 64];
            if (out) atomicAdd(&gptr(hb->hist_acc)[HPK_HREP_OF(blockIdx.x) * HPK_ACC_STRIDE + tix], (unsigned long long)out);
        }
        {
            unsigned* lc = reinterpret_cast<unsigned*>(gptr(hb->small) + HPK_OFF_LEAN);
            if (tix == 0 && mylean) { atomicAdd(&lc[0], mylean & 0xffffu); if (mylean >> 16) atomicAdd(&lc[1], mylean >> 16); }
            if (lane_k == 0 && myexpl) atomicAdd(&lc[2], myexpl);
            mylean = 0u; myexpl = 0u;
        }
        __syncthreads();
        myhist = 0u; mycand = 0u;
    };
    int hband = -1;
#pragma unroll 1
    while (have) {
    const unsigned cbw = bw;
    const int band = (int)(cbw & 0xffffu);
    const HpkBandDesc* __restrict__ bd = bands + band;
    const int n = bd->n, bnum = bd->num;
    const int Dm = a.D < bd->num - 1 ? a.D : bd->num - 1;
    const int wg_p = bd->wguess;
    const int W = bd->W, TR = bd->TR, TC = bd->TC, J_p = bd->J, Dg_p = bd->Dg, tilecap_p = bd->tilecap;
    tile_issue_lean(a, bd, rb, cj, wave_k, lane_k, stage0 + sp * HPK_LST_BYTES);     // the band's first tile: nothing is prefetched across a boundary
    if (hband >= 0) flush_hist(bands + hband);
    hband = band;
#pragma unroll 1
    do {
    int wave = wave_k, lane = lane_k;
    asm volatile("" : "+s"(wave));
    asm volatile("" : "+v"(lane));
    const int tid = rb * J_p + cj;
    const int r0 = rb * TR;
    const int c0 = r0 + mw + cj * TC;
    const bool empty_tile = c0 >= n || (mw + cj * TC - (TR - 1)) > Dg_p;
    const int tn = __builtin_amdgcn_readfirstlane((int)tnext);
    const unsigned bw_next = (unsigned)__builtin_amdgcn_readfirstlane((int)bnext);
    const bool have_next = tn != -1;
    const bool pre_next = have_next && bw_next == cbw;         // the next tile is this band's: its rows are prefetched
    const int rb_next = (int)((unsigned)tn >> 8), cj_next = tn & 255;
    if (wave == 0) {
        TileWalk<1> tw;
        walk_take(twl, tw);
        tw.step(a, bands);
        if (lane == 0) { tseq[tpar ^ 1] = tw.word(a); tband[tpar ^ 1] = tw.bword(); }
        walk_park(twl, tw, lane);
    }
    tpar ^= 1;
    if (empty_tile) {
        // (its rows were requested like any tile's: waited for before their stage is written again)
        if (pre_next) tile_issue_lean(a, bd, rb_next, cj_next, wave, lane, stage0 + (sp ^ 1) * HPK_LST_BYTES);
        sp ^= 1;
        have = have_next; rb = rb_next; cj = cj_next; bw = bw_next;
        __syncthreads();
        tnext = lds_u32(lds0 + (unsigned)((unsigned char*)tseq - smem) + (unsigned)tpar * 4u);
        bnext = lds_u32(lds0 + (unsigned)((unsigned char*)tband - smem) + (unsigned)tpar * 4u);
        continue;
    }
    unsigned* __restrict__ tcnt = tcount + par;
    // the tile's rows out of their stage, and the next tile's requested at once into the other one (a whole tile of lead)
    unsigned raw[10], wbits;
    bool wr_nz;
    tile_take_lean(a, bd, rb, cj, wave, lane, stage0 + sp * HPK_LST_BYTES, raw, wbits, wr_nz);
    if (pre_next) tile_issue_lean(a, bd, rb_next, cj_next, wave, lane, stage0 + (sp ^ 1) * HPK_LST_BYTES);
    sp ^= 1;
    // ---- phase 1 (rows): capped counts and valid flags of the lane's ten cells, the candidates among them (a bit mask the lane
    // keeps), the row prefix of the packed plane (see hpk_stencil_s: a table row per DPP row of 16 lanes)
    unsigned cm;
    const int Y = 4 * wave + (lane >> 4);
    const int XH = (LC - 1) - 10 * (lane & 15);
    {
    const int rr = r0 - W - 1 + Y;
    const int y = Y - (W + 1);
    const int kH = mw + cj * TC + 1 + XH - Y;
    const int xo = XH - W;
    int lim = n - rr;
    lim = lim < bnum ? lim : bnum;
    lim = rr >= 0 ? lim : 0;
    lim = lim > 0 ? lim : 0;
    unsigned cmask;
    {
        const int A = kH - mw;
        int elo = A - (Dm - mw), ehi = A < xo ? A : xo;
        elo = elo > xo - TC + 1 ? elo : xo - TC + 1;
        elo = elo > 0 ? elo : 0;
        ehi = ehi < 9 ? ehi : 9;
        const unsigned m = ((2u << (9 - elo)) - 1u) & ~((1u << (9 - ehi)) - 1u);
        cmask = ((unsigned)y < (unsigned)TR && elo <= ehi) ? m : 0u;
    }
    unsigned pk[10];
    cm = 0u;
#define HPK_CELLS_LEAN(MASKED)                                                                                                 \
    _Pragma("unroll") for (int e = 0; e < 10; ++e) {                                                                           \
        const int i = 9 - e;                                                                                                   \
        const int k = kH - e;                                                                                                  \
        unsigned rb32 = raw[i];                                                                                                \
        if (MASKED) rb32 = (unsigned)k < (unsigned)lim ? rb32 : 0u;                                                            \
        const unsigned ru = (unsigned)__uint_as_float(rb32);                                                                   \
        pk[e] = ru < pkcap_p ? ru : pkcap_p;                                                                                   \
        asm("v_cmp_ne_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(cm) : "v"(ru) : "vcc");             \
    }
    {
        const bool inner = (kH - 9 >= mw) & (kH < lim);
        const bool all_inner = ballot64(!inner) == 0ull;
        if (all_inner) { HPK_CELLS_LEAN(false) }
        else { HPK_CELLS_LEAN(true) }
        // a cell's balanced value is non-zero where its count, its row weight and its column weight are (masked cells: and its
        // diagonal is at least min(ww))
        unsigned vm = cm & wbits;
        vm = wr_nz ? vm : 0u;
        if (!all_inner) {
            const int A = kH - mw;
            const unsigned km = A >= 9 ? 0x3ffu : (A < 0 ? 0u : (0x3ffu & ~((1u << (9 - A)) - 1u)));
            vm &= km;
        }
#pragma unroll
        for (int e = 0; e < 10; ++e) pk[e] |= ((vm >> (9 - e)) & 1u) << PK_SHIFT;
    }
#undef HPK_CELLS_LEAN
    cm &= cmask;
    {   // the tile's candidates are counted (the band's total), not listed
        const unsigned cnt = (unsigned)__popc(cm);
        const unsigned nrow = (unsigned)__builtin_amdgcn_readlane((int)wave_inclusive_scan(cnt), 63);
        const unsigned addr = (unsigned)(size_t)tcnt;
        asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(addr), "v"(nrow) : "memory");
    }
#pragma unroll
    for (int e = 1; e < 10; ++e) pk[e] += pk[e - 1];
    {
        const unsigned ex = row16_exclusive_scan(pk[9]);
        unsigned* __restrict__ dst = &Sp[Y * LC + XH - 9];
#pragma unroll
        for (int i = 0; i < 5; ++i)
            *reinterpret_cast<uint2*>(&dst[2 * i]) = make_uint2(pk[9 - 2 * i] + ex, pk[8 - 2 * i] + ex);
    }
    }
    __syncthreads();
    // ---- phase 2 (columns): waves 0-9, a thread per column and chunk of 16 rows
    {
        unsigned creg[16];
        const int tix = wave * 64 + lane;
        const int ch = tix >= 2 * LC ? (tix >= 3 * LC ? 3 : 2) : (tix >= LC ? 1 : 0), col = tix - ch * LC;
        if (wave < 10) {
            const unsigned* __restrict__ src = &Sp[(16 * ch) * LC + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) creg[i] = src[i * LC];
#pragma unroll
            for (int i = 1; i < 16; ++i) creg[i] += creg[i - 1];
            if (ch < 3) utot3[ch * LC + col] = creg[15];
        }
        __syncthreads();
        if (wave < 10) {
            unsigned base = ch >= 1 ? utot3[col] : 0u;
            base += ch >= 2 ? utot3[LC + col] : 0u;
            base += ch >= 3 ? utot3[2 * LC + col] : 0u;
            unsigned* __restrict__ dst = &Sp[(16 * ch) * LC + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[i * LC] = creg[i] + base;
        }
    }
    __syncthreads();
    // gap rows (callers.py:238), exact on the valid-count field
    const int tx = wave * 64 + lane;
    if (tx < TR && r0 + tx < n) {
        const bool last = (cj == J_p - 1) || (c0 + TC >= n) || (mw + (cj + 1) * TC - (TR - 1)) > Dg_p;
        const int Yg = tx + W + 1;
        unsigned rs = Sp[Yg * LC + W] - Sp[(Yg - 1) * LC + W];
        const int xe = last ? W : W + TC;
        rs -= last ? 0u : Sp[Yg * LC + xe] - Sp[(Yg - 1) * LC + xe];
        if ((rs >> PK_SHIFT) != 0u) gptr(bd->gap)[r0 + tx] = 1;
    }
    if (tx == 0) { tcount[par ^ 1] = 0u; tcount[4 + (par ^ 1)] = 0u; tflag[par ^ 1] = 0u; }
    // ---- the candidates: every lane goes through its own (bit 9 - e of cm) and tests the widest Reads box; the few that pass are
    // a batch of their own - first sufficient width, resolve count, record, sums cell by cell
    const int64_t tbase = (int64_t)tid * tilecap_p;
    unsigned* __restrict__ ent_t = gptr(bd->rec_ent) + tbase;
    if (a.dbg_stop != 2) {
        const unsigned prow = lds0 + (unsigned)(Y * LC + XH) * 4u;          // P(Y, XH); cell e: - 4 e
        unsigned bits = cm;
#pragma unroll 1
        while (ballot64(bits != 0u) != 0ull) {
            const bool on = bits != 0u;
            const int bpos = on ? 31 - __clz((int)bits) : 0;
            bits = on ? bits & ~(1u << bpos) : 0u;
            const int e = 9 - bpos;
            const unsigned pb = prow - 4u * (unsigned)e;
            const unsigned sr = lds_u32(pb);
            const unsigned dnW = (unsigned)W * (unsigned)(LC * 4), lfW = (unsigned)W * 4u;
            const unsigned bl = (lds_u32(pb + (dnW - lfW)) - lds_u32(pb + dnW) - lds_u32(pb - lfW) + sr) & PK_MASK;
            // (Reads = widest box - box p0 >= min_local_reads needs the widest box alone to reach it)
            if (ballot64(on & (bl >= (unsigned)minr_p)) == 0ull) continue;
            const unsigned b0 = (p0_p > 0) ? ((lds_u32(pb + (unsigned)p0_p * (unsigned)(LC * 4 - 4)) - lds_u32(pb + (unsigned)p0_p * (unsigned)(LC * 4)) - lds_u32(pb - (unsigned)p0_p * 4u) + sr) & PK_MASK) : 0u;
            const bool cand = on & (bl - b0 >= (unsigned)minr_p);
            if (ballot64(cand) == 0ull) continue;
            // ---- one batch (see hpk_stencil_s, phase 3)
            const unsigned cntc = (sr - lds_u32(pb + 4u) - lds_u32(pb - (unsigned)(LC * 4)) + lds_u32(pb - (unsigned)(LC * 4) + 4u)) & PK_MASK;
            const int x = XH - e - W, y = Y - (W + 1);
            const unsigned id = (unsigned)x | ((unsigned)y << HPK_ENT_YSHIFT) | (cntc << HPK_ENT_CNT_SHIFT);
            int wstar = cand ? W : 255;
            {
                const unsigned dnf = (unsigned)wmin_p * (unsigned)(LC * 4), lff = (unsigned)wmin_p * 4u;
                const unsigned bf = (lds_u32(pb + (dnf - lff)) - lds_u32(pb + dnf) - lds_u32(pb - lff) + sr) & PK_MASK;
                wstar = (cand & (bf - b0 >= (unsigned)minr_p)) ? wmin_p : wstar;
            }
            if (W - wmin_p > 1 && ballot64(wstar == W) != 0ull) {
#pragma unroll 1
                for (int wa = wmin_p + 1; wa < W; ++wa) {
                    const unsigned dn = (unsigned)wa * (unsigned)(LC * 4), lf = (unsigned)wa * 4u;
                    const unsigned rd = (lds_u32(pb + (dn - lf)) - lds_u32(pb + dn) - lds_u32(pb - lf) + sr) & PK_MASK;
                    wstar = ((wstar == W) & (rd - b0 >= (unsigned)minr_p)) ? wa : wstar;
                }
            }
            {   // resolve histogram by width (16-bit fields per lane for the first eight widths, a ballot per width beyond)
                const unsigned off = (unsigned)(wstar - wmin_p);
                const unsigned long long inc = 1ull << ((off & 3u) * 16u);
                hpack0 += off < 4u ? inc : 0ull;
                hpack1 += (off - 4u) < 4u ? inc : 0ull;
                if (W - wmin_p >= 8 && ballot64((off >= 8u) & (wstar != 255)) != 0ull) {
#pragma unroll 1
                    for (int w = wmin_p + 8; w <= W; ++w) {
                        const unsigned c = (unsigned)__popcll(ballot64(wstar == w));
                        if (lane == w) myhist += c;
                    }
                }
            }
            if (W > planw_p) wstar = wstar > planw_p ? 255 : wstar;
            const bool live = cand & (wstar <= wg_p);
            const unsigned long long lm = ballot64(live);
            if (lm == 0ull) continue;
            // the tile's record region: up to a.lean_max candidates; a batch that would go beyond gives the tile up
            const unsigned nl = (unsigned)__popcll(lm);
            unsigned before = 0u;
            if (lane == 0) before = atomicAdd(&tcount[4 + par], nl);
            before = (unsigned)__builtin_amdgcn_readfirstlane((int)before);
            if (before + nl > (unsigned)a.lean_max) {
                if (lane == 0) tflag[par] = 1u;
                continue;
            }
            myexpl += nl;
            wstar = live ? wstar : 255;
            const unsigned ri = before + __builtin_amdgcn_mbcnt_hi((unsigned)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)lm, 0u));
            if (live) ent_t[ri] = id;
#pragma unroll 1
            for (int q = 0; q < (SINGLE ? 1 : nslots_p); ++q) {
                int sq;
                if (SINGLE) sq = wstar == 255 ? 0xff : wstar - wmin_p;
                else {
                    const int wf = (q == 0) ? wf_q[0] : (q == 1) ? wf_q[1] : (q == 2) ? wf_q[2] : wf_q[3];
                    const int wq = wstar > wf ? wstar : wf;
                    sq = (int)stepof[q * 32 + (wq & 31)];
                    sq = ((wstar == 255) | (wq > wg_p)) ? 0xff : sq;
                }
                const bool act = sq != 0xff;
                double SK = 0.0, SY = 0.0;
                unsigned long long todo = ballot64(act);
                while (todo != 0ull) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const int er = __builtin_amdgcn_readlane(r0 + y, src), ec = __builtin_amdgcn_readlane(c0 + x, src);
                    const int es = __builtin_amdgcn_readlane(sq, src);
                    const double2 ex = explicit_sums_wave(gptr(bd->raw), nullptr, gptr(bd->weight), plan->steps[es].m, W, er, ec, n, bd->num, bd->ld, mw, lane);
                    if (lane == src) { SK = ex.x; SY = ex.y; }
                }
                if (live) {
                    const int64_t o = q * bd->rec_stride + tbase + ri;
                    gptr(bd->rec_S)[o] = make_double2(act ? SK : 0.0, act ? SY : 0.0);
                    gptr(bd->rec_W)[o] = act ? (uint8_t)(sq + 1) : (uint8_t)0;
                }
            }
        }
    }
    tnext = lds_u32(lds0 + (unsigned)((unsigned char*)tseq - smem) + (unsigned)tpar * 4u);
    bnext = lds_u32(lds0 + (unsigned)((unsigned char*)tband - smem) + (unsigned)tpar * 4u);
    __syncthreads();                 // every wave is done with this tile's table
    if (wave == 0) {
        if (pend_tid >= 0) {
            const unsigned nu = (pend_c + (unsigned)HPK_UNIT - 1u) / (unsigned)HPK_UNIT;
            const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane((int)pend_off);
            if ((unsigned)lane < nu) gptr(bands[pend_band].units)[off + lane] = make_uint2(pend_rc, (unsigned)lane | (pend_c << 8));
        }
        pend_tid = -1;
        const bool given_up = __builtin_amdgcn_readfirstlane((int)tflag[par]) != 0;
        if (given_up) {
            // more candidates that count than the cell-by-cell path takes: hpk_stencil_s computes the tile in full (redo queue)
            if (lane == 0) {
                const unsigned qi = atomicAdd(&a.redoq[0], 1u);
                a.redoq[4 + 2 * qi] = (unsigned)band;
                a.redoq[5 + 2 * qi] = (unsigned)rb << 8 | (unsigned)cj;
            }
        } else {
            const unsigned nrec = (unsigned)__builtin_amdgcn_readfirstlane((int)tcount[4 + par]);
            if (nrec > 0u) {
                pend_tid = tid;
                pend_rc = (unsigned)rb << 8 | (unsigned)cj;
                pend_band = band;
                pend_c = nrec;
                if (lane == 0) pend_off = atomicAdd(reinterpret_cast<unsigned*>(gptr(bd->small) + HPK_OFF_NUNITS), (pend_c + (unsigned)HPK_UNIT - 1u) / (unsigned)HPK_UNIT);
            }
            if (lane == 0) gptr(bd->tile_cnt)[tid] = nrec;
        }
        mycand += *tcnt;
        mylean += given_up ? 0x10001u : 1u;
    }
    have = have_next; rb = rb_next; cj = cj_next; bw = bw_next;
    par ^= 1;
    } while (have && bw == cbw);
    }
    if (hband >= 0) flush_hist(bands + hband);
    if (wave_k == 0 && pend_tid >= 0) {
        const unsigned nu = (pend_c + (unsigned)HPK_UNIT - 1u) / (unsigned)HPK_UNIT;
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane((int)pend_off);
        if ((unsigned)lane_k < nu) gptr(bands[pend_band].units)[off + lane_k] = make_uint2(pend_rc, (unsigned)lane_k | (pend_c << 8));
    }
}

// ------------------------------------------------------------------ band from the pixel table (hpk_devband_create)
// One pixel (bin1, bin2, count) per thread, scatter-added into the zero-filled band raw[n][ld]: counts are integers
// below 2^24, so the f32 sums are exact whatever order the adds come in.  A bin outside [0, n) sets the error flag.
__global__ void __launch_bounds__(256) hpk_coo_scatter(const int64_t* __restrict__ bin1, const int64_t* __restrict__ bin2,
                                                       const void* __restrict__ count, int count_f64, int64_t nnz, int n, int num,
                                                       int64_t ld, float* __restrict__ raw, unsigned long long* __restrict__ info) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long stored = 0ull;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += stride) {
        const int64_t i = bin1[t], j = bin2[t];
        const int64_t a = i < j ? i : j, b = i < j ? j : i;
        if (a < 0 || b >= n) { info[1] = 1ull; continue; }
        const int64_t k = b - a;
        if (k >= num) continue;                             // beyond the band
        const float v = count_f64 ? (float)static_cast<const double*>(count)[t] : (float)static_cast<const int32_t*>(count)[t];
        atomicAdd(&raw[a * ld + k], v);
        ++stored;
    }
    for (int off = 32; off > 0; off >>= 1) stored += __shfl_down(stored, off);
    if ((threadIdx.x & 63) == 0 && stored) atomicAdd(&info[0], stored);
}

// ------------------------------------------------------------------ 1-D expected IR[d] and biases (scripts/pyHICCUPS:149-166)
// IR[d] = mean over diagonal d of the balanced values, where *stored* (non-zero) pixels in masked bins are NaN and
// are left out of both sum and count, while unstored pixels count as 0 even in masked bins.  Deterministic
// two-level reduction: workgroup b sums rows [b * HPK_IR_ROWS, ...) per diagonal (lanes = diagonals, coalesced),
// hpk_ir_final adds the partials of one diagonal in a fixed order (eight threads, then their eight sums).  blockIdx.z = band of the batch.
#define HPK_IR_ROWS 128                             // rows per partial sum (hpk_api.cpp sizes psum / pnan for 32: an upper bound)
#define HPK_IR_KPT 8                                // diagonals per thread: a workgroup covers 2048 diagonals of its rows
// A workgroup sums 128 whole band rows (their stored diagonals in chunks of 2048: every row is read as one contiguous
// stretch, 8 KB at num = 2011) - thread t takes the diagonals t, t + 256, ... and keeps their sums in registers; the
// column weights of the window, weight[rbeg + k + j], sit in LDS.  (Round 2's version gave a workgroup 256 diagonals of
// 32 rows - 1 KB per row, the rest of the row read by seven other workgroups at other times - and fetched the column
// weight of every pixel from the cache: 1.7 TB/s, 2.9 ms of a whole genome @5 kb's 7.6; 16-byte loads - four consecutive
// diagonals per thread - were tried on top and were no faster.)  Sums run over the rows in ascending order.
__global__ void __launch_bounds__(256) hpk_ir_partial(const HpkBandDesc* __restrict__ bands, int mw) {
    const HpkBandDesc* __restrict__ bd = bands + blockIdx.z;
    if (!bd->derive) return;
    const int n = bd->n, num = bd->num;
    const int64_t ld = bd->ld;
    const int rbeg = blockIdx.x * HPK_IR_ROWS;
    const int k0 = mw + (int)blockIdx.y * (256 * HPK_IR_KPT);
    if (rbeg >= n || k0 >= num) return;
    const int kend = (k0 + 256 * HPK_IR_KPT < num) ? k0 + 256 * HPK_IR_KPT : num;
    const float* __restrict__ raw = gptr(bd->raw);
    const double* __restrict__ weight = gptr(bd->weight);
    double* __restrict__ psum = gptr(bd->psum);
    unsigned* __restrict__ pnan = gptr(bd->pnan);
    __shared__ double wrow[HPK_IR_ROWS];
    __shared__ double wwin[256 * HPK_IR_KPT + HPK_IR_ROWS];      // weight[rbeg + k0 + i]
    if ((int)threadIdx.x < HPK_IR_ROWS) wrow[threadIdx.x] = (rbeg + (int)threadIdx.x < n) ? weight[rbeg + threadIdx.x] : 0.0;
    for (int i = threadIdx.x; i < kend - k0 + HPK_IR_ROWS; i += 256) {
        const int c = rbeg + k0 + i;
        wwin[i] = c < n ? weight[c] : 0.0;
    }
    __syncthreads();
    double s[HPK_IR_KPT];
    unsigned nn[HPK_IR_KPT];
#pragma unroll
    for (int q = 0; q < HPK_IR_KPT; ++q) { s[q] = 0.0; nn[q] = 0u; }
    const int kt = k0 + (int)threadIdx.x;
    const float* __restrict__ src = raw + (int64_t)rbeg * ld + kt;
    if (rbeg + HPK_IR_ROWS + kend <= n && kend - k0 == 256 * HPK_IR_KPT) {
        // every row of the group holds every diagonal of the chunk: no tests
#pragma unroll 4
        for (int j = 0; j < HPK_IR_ROWS; ++j) {
            const double wr = wrow[j];
#pragma unroll
            for (int q = 0; q < HPK_IR_KPT; ++q) {
                const float cnt = src[(int64_t)j * ld + 256 * q];
                const double b = ((double)cnt * wr) * wwin[(int)threadIdx.x + 256 * q + j];
                if (cnt != 0.f) { if (b == b) s[q] += b; else ++nn[q]; }
            }
        }
    } else {
#pragma unroll 2
        for (int j = 0; j < HPK_IR_ROWS; ++j) {
            const double wr = wrow[j];
#pragma unroll
            for (int q = 0; q < HPK_IR_KPT; ++q) {
                const int k = kt + 256 * q;
                // rows of this group that still have column r + k inside the matrix: j < n - k - rbeg
                const bool in = k < kend && rbeg + j + k < n;
                const float cnt = in ? src[(int64_t)j * ld + 256 * q] : 0.f;
                const double b = ((double)cnt * wr) * wwin[in ? (int)threadIdx.x + 256 * q + j : 0];
                if (cnt != 0.f) { if (b == b) s[q] += b; else ++nn[q]; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < HPK_IR_KPT; ++q) {
        const int k = kt + 256 * q;
        if (k < kend) {
            psum[(int64_t)blockIdx.x * num + k] = s[q];
            pnan[(int64_t)blockIdx.x * num + k] = nn[q];
        }
    }
}
// Workgroups beyond the diagonals turn the weights into biases (scripts/pyHICCUPS:163-166) - one launch less.
__global__ void __launch_bounds__(256) hpk_ir_final(const HpkBandDesc* __restrict__ bands, int mw, int nirb) {
    const HpkBandDesc* __restrict__ bd = bands + blockIdx.y;
    if (!bd->derive) return;
    const int n = bd->n, num = bd->num;
    if ((int)blockIdx.x >= nirb) {
        const int i = ((int)blockIdx.x - nirb) * 256 + threadIdx.x;
        if (i < n && bd->derive == 1) {         // (derive == 2: the caller gave the biases, only IR is derived)
            const double w = gptr(bd->weight)[i];
            gptr(bd->b1)[i] = (w == 0.0 || w != w) ? 0.0 : 1.0 / w;
        }
        return;
    }
    // 32 consecutive diagonals per workgroup (the partials of one row group are read as 256 contiguous bytes), eight
    // threads per diagonal that each walk every eighth row group; their sums are added in a fixed order
    const int kl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kl;
    const int nparts = (n + HPK_IR_ROWS - 1) / HPK_IR_ROWS;
    const double* __restrict__ psum = gptr(bd->psum);
    const unsigned* __restrict__ pnan = gptr(bd->pnan);
    double s = 0.0;
    unsigned long long nn = 0ull;
    if (k >= mw && k < num) for (int p = pl; p < nparts; p += 8) { s += psum[(int64_t)p * num + k]; nn += pnan[(int64_t)p * num + k]; }
    __shared__ double ls[8][32];
    __shared__ unsigned long long ln[8][32];
    ls[pl][kl] = s; ln[pl][kl] = nn;
    __syncthreads();
    if (pl != 0 || k >= num) return;
    for (int q = 1; q < 8; ++q) { s += ls[q][kl]; nn += ln[q][kl]; }
    const long long denom = (long long)(n - k) - (long long)nn;
    gptr(bd->IR)[k] = (k >= mw && n - k > 0) ? s / (double)denom : 0.0;       // 0/0 -> NaN like numpy's mean of an empty slice
}
// ------------------------------------------------------------------ record bound by depth class, lean column chunks
// The width at which a chromosome's widening freezes grows with its depth; a context that serves chromosomes of several
// samples would otherwise write every band's records up to the deepest sample's width (hpk_api.cpp: the bound is verified at
// collection whatever it was).  One workgroup per band: the counts of every 64th row, summed per diagonal (integers: the
// same sums whatever the order of the adds); their mean over the band's pixels, its quarter octave = the depth class.
// From the same sums: the column chunks whose tiles the stencil builds without their f64 plane (hpk_stencil_s, lean tiles).  A
// candidate counts if its Reads - the lower-left rings p0 + 1 .. w of the raw counts, w up to the band's bound - reach
// min_local_reads (callers.py:197-217); for the pixels of chunk c nearest to the diagonal the mean of that sum follows from
// the per-diagonal means, and chunks from the first one on after which it stays below lean_frac x min_local_reads are lean
// (a prediction, verified tile by tile: a lean tile with more than a handful of candidates that count is computed once more).
// Lean tiles take "count, row weight and column weight non-zero" for "balanced value non-zero": not with weights so small
// that a product of two underflows (none in practice; checked here).
#define HPK_CLS_KMAX 4096                           // diagonals profiled (beyond: the last profiled one stands for the rest)
__global__ void __launch_bounds__(1024) hpk_band_class(HpkBandDesc* __restrict__ bands, HpkClassArgs a) {
    HpkBandDesc* bd = bands + blockIdx.x;
    const int n = bd->n, num = bd->num;
    const int64_t ld = bd->ld;
    const int mw = a.mw;
    const int Dm = a.D < num - 1 ? a.D : num - 1;
    const float* __restrict__ raw = gptr(bd->raw);
    __shared__ unsigned long long S[HPK_CLS_KMAX];  // per diagonal: sum of the sampled rows' counts
    __shared__ unsigned long long tot;
    __shared__ int lean_from, tiny_w, wg_s;
    const int t = (int)threadIdx.x;
    // every 64th row; chromosomes beyond 32 768 bins: every 128th, 192nd ... (at most 512 sampled rows)
    const int RS = 64 * ((n + 32767) / 32768);
    const int KP = num < HPK_CLS_KMAX ? num : HPK_CLS_KMAX;
    for (int k = t; k < KP; k += 1024) S[k] = 0ull;
    if (t == 0) { tot = 0ull; lean_from = 0; tiny_w = 0; }
    __syncthreads();
    {
        // thread -> (diagonal, group of sampled rows): consecutive threads read consecutive diagonals of one row
        const int KPr = (KP + 63) & ~63;
        const int G = KPr <= 1024 ? 1024 / KPr : 1;
        for (int idx = t; idx < KPr * G; idx += 1024) {
            const int g = idx / KPr, k = idx - g * KPr;
            if (k >= KP) continue;
            unsigned long long acc = 0ull;
            for (int r = RS * g; r + k < n; r += RS * G) acc += (unsigned long long)raw[(int64_t)r * ld + k];
            if (acc) atomicAdd(&S[k], acc);
            if (k >= mw && k <= Dm && acc) atomicAdd(&tot, acc);
        }
        if (Dm >= HPK_CLS_KMAX) {       // (bands wider than the profile: the class still sees every band diagonal)
            unsigned long long acc = 0ull;
            for (int k = HPK_CLS_KMAX + t; k <= Dm; k += 1024)
                for (int r = 0; r + k < n; r += RS) acc += (unsigned long long)raw[(int64_t)r * ld + k];
            if (acc) atomicAdd(&tot, acc);
        }
        if (a.lean && gptr(bd->weight)) {
            const double* __restrict__ w = gptr(bd->weight);
            unsigned* __restrict__ wnz = bd->off_wnz ? reinterpret_cast<unsigned*>(gptr(bd->small) + bd->off_wnz) : nullptr;
            bool tiny = false;
            for (int i0 = 0; i0 < n; i0 += 1024) {          // (whole waves: the bits of 64 bins are one ballot)
                const int i = i0 + t;
                // (on the bit pattern: NaN is above the infinities' 0x7ff0..., and a comparison of fabs(w) with itself is folded away)
                const unsigned long long ub = i < n ? ((unsigned long long)__double_as_longlong(w[i]) & 0x7fffffffffffffffull) : 0ull;
                const bool isw = ub != 0ull && ub <= 0x7ff0000000000000ull;         // a weight: non-zero, not NaN
                tiny = tiny || (isw && ub < 0x20b0000000000000ull);                 // |w| < ~2^-500 (1e-150)
                const unsigned long long nz = __ballot(isw);
                if (wnz && (t & 63) == 0 && i < n) {
                    const int wd = (i + HPK_WNZ_LEAD) >> 5;
                    wnz[wd] = (unsigned)nz; wnz[wd + 1] = (unsigned)(nz >> 32);
                }
            }
            if (tiny) tiny_w = 1;
        }
    }
    __syncthreads();
    if (t == 0) {
        long long cells = 0;
        for (int r = 0; r < n; r += RS) {
            const int kmax = Dm < n - 1 - r ? Dm : n - 1 - r;
            cells += kmax >= mw ? kmax - mw + 1 : 0;
        }
        int cls = 0;
        if (cells > 0 && tot > 0ull) {
            cls = (int)floor(4.0 * log2((double)tot / (double)cells)) + 32;
            cls = cls < 0 ? 0 : (cls > HPK_NCLASS - 1 ? HPK_NCLASS - 1 : cls);
        }
        int wg = a.own ? bd->wguess : a.wg_all;
        if (a.table || a.own) {
            if (!a.own) {
                const int tb = (int)a.table[cls];
                if (tb >= 0) {
                    wg = tb + a.margin;
                    wg = wg < a.wmin ? a.wmin : wg;
                    wg = wg < a.wg_all ? wg : a.wg_all;
                }
            }
            bd->wguess = wg;
            // ... and the band's tiles are laid out for that bound's halo (at least 4, at least the plan's narrowest width)
            int halo = bd->W;
            if (a.halo) {
                int Wh = wg > a.wmin ? wg : a.wmin;
                Wh = Wh > 4 ? Wh : 4;
                if (Wh < bd->W) {
                    const HpkGeo g = hpk_geo_of(Wh, a.planW, a.D, mw, a.tr_cap);
                    bd->W = g.W; bd->Dg = g.Dg; bd->TR = g.TR; bd->TC = g.TC; bd->J = g.J; bd->tilecap = g.tilecap;
                    bd->ntiles = ((n + g.TR - 1) / g.TR) * g.J;
                    bd->chunk = (bd->ntiles + 7) / 8;
                    bd->rec_stride = (int64_t)bd->ntiles * g.tilecap;
                    halo = Wh;
                }
            }
            *reinterpret_cast<unsigned*>(gptr(bd->small) + HPK_OFF_BCLASS) = 0x10000u | ((unsigned)halo << 20) | ((unsigned)cls << 8) | (unsigned)wg;
        }
        wg_s = wg;
    }
    __syncthreads();
    const int gTR = bd->TR, gTC = bd->TC, gJ = bd->J;      // (thread 0's writes: behind the barrier)
    int lean_cj = 0x7fffffff;
    if (a.lean && !tiny_w && gptr(bd->weight) && bd->off_wnz) {
        // (at the width of the band's halo - its bound, or min(ww) / 4 where those are wider: which tiles are lean is then a function of
        // the band and its tile layout alone, what spec_halo = 2 needs)
        int wg = wg_s > a.wmin ? wg_s : a.wmin;
        wg = wg > 4 ? wg : 4;
        // mean count of a pixel on diagonal k: S[k] over the sampled rows that hold it
        auto mu = [&](int k) -> double {
            if (k < 0) return 0.0;                  // below the main diagonal: not stored, not counted (callers.py:50-96)
            k = k < KP ? k : KP - 1;
            const int rows = (n - k + RS - 1) / RS;
            return rows > 0 ? (double)S[k] / (double)rows : 0.0;
        };
        for (int c = t; c < gJ; c += 1024) {
            int kd = mw + c * gTC - (gTR - 1);       // the chunk's pixels nearest to the diagonal
            kd = kd < mw ? mw : kd;
            double lam = 0.0;                        // mean Reads there at the band's bound: lower-left cells (i, j), i, j = 1 .. wg, without the p0 x p0 corner
            for (int i = 1; i <= wg; ++i)
                for (int jj = 1; jj <= wg; ++jj)
                    if (i > a.p0 || jj > a.p0) lam += mu(kd - i - jj);
            if (!(lam <= (double)a.lean_frac * (double)a.minr)) atomicMax(&lean_from, c + 1);
        }
        __syncthreads();
        lean_cj = lean_from;
        // The lean kernel pays where most of a band is far field (5 kb, 1 kb: 12 of 15 chunks); with one or two lean chunks of
        // four - 10 kb - hpk_stencil_s does the same tiles as fast itself (their candidates are denser, and it walks them in
        // batches of 64): a band whose lean share is below a.lean_share stays with it entirely.
        if ((gJ - lean_cj) * 100 < gJ * a.lean_share) lean_cj = 0x7fffffff;
    }
    if (t == 0) bd->lean_cj = lean_cj;
}

// ------------------------------------------------------------------ freeze
// The reference's frozen_w / break logic on the chromosome's totals (callers.py:208-229, 505-511), one thread.
// hist[s] = candidates resolved at step s, hist[HPK_HIST_NCAND] = all candidates; executed may be nullptr.
__device__ __forceinline__ void freeze_replay(const HpkDevPlan* __restrict__ plan, const unsigned long long* hist, const int* swi,
                                              const int* sslot, int32_t* executed, int& fw_out, int& err_out) {
    const long long total = (long long)hist[HPK_HIST_NCAND];
    long long unres[HPK_KSLOTS];
    for (int q = 0; q < HPK_KSLOTS; ++q) unres[q] = total;
    int fw = plan->W;
    int e = 0;
    const int nsteps = plan->nsteps, maxw = plan->maxw;
    const bool bh = plan->mode == HPK_MODE_BHFDR;
    for (int s = 0; s < nsteps; ++s) {
        struct { int wi, slot; } st = {swi[s], sslot[s]};
        if (st.wi > fw) { if (executed) executed[s] = 0; continue; }   // callers.py:133-134 / break at 505-511
        if (executed) executed[s] = 1;
        const long long before = unres[st.slot];
        if (before == 0 && e == 0) e = s + 1;                           // the reference raises here
        const long long now = (long long)hist[s];
        const double vr = before ? (double)now / (double)before : 0.0;  // callers.py:208 / 492
        unres[st.slot] = before - now;
        const double lr = total ? (double)unres[st.slot] / (double)total : 0.0;   // callers.py:219 / 501
        const bool widest = bh || (st.wi >= maxw);
        if (widest && (vr < 0.3 || lr < 0.03)) fw = st.wi;              // callers.py:223-229
    }
    fw_out = fw;
    err_out = e;
}
// The decision for launches that no scoring follows (probes, HPK_FLAG_NO_SCORE; the scoring kernel replays it in its own
// prologue): one workgroup per band on the totals the stencil workgroups added up (HpkBandDesc::hist_acc).
__global__ void __launch_bounds__(128) hpk_freeze_tot(const HpkDevPlan* __restrict__ plan, const HpkBandDesc* __restrict__ bands) {
    const HpkBandDesc* __restrict__ bd = bands + blockIdx.x;
    __shared__ unsigned long long hist[HPK_MAX_STEPS + 1];
    __shared__ int swi[HPK_MAX_STEPS], sslot[HPK_MAX_STEPS];
    unsigned char* small = gptr(bd->small);
    if ((int)threadIdx.x < plan->nsteps) { swi[threadIdx.x] = plan->steps[threadIdx.x].wi; sslot[threadIdx.x] = plan->steps[threadIdx.x].slot; }
    if (threadIdx.x <= HPK_MAX_STEPS) {
        hist[threadIdx.x] = hist_total(gptr(bd->hist_acc), (int)threadIdx.x);
        reinterpret_cast<unsigned long long*>(small + HPK_OFF_HIST)[threadIdx.x] = hist[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int fw, e;
    freeze_replay(plan, hist, swi, sslot, reinterpret_cast<int32_t*>(small + HPK_OFF_EXEC), fw, e);
    *reinterpret_cast<int32_t*>(small + HPK_OFF_FROZEN) = fw;
    *reinterpret_cast<int32_t*>(small + HPK_OFF_ERR) = e;
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
    if (lane == 0) gap[r] = (m == 0ull) ? 0 : 1;        // same "row has signal" flag as the stencil kernel writes
}

__global__ void __launch_bounds__(256) hpk_ptab(const double* __restrict__ bounds, const int32_t* __restrict__ off,
                                                const double* __restrict__ sfe, double* __restrict__ ptab, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int ch = 1;
    while (ch < HPK_NB_TAB && i >= off[ch + 1]) ++ch;      // off[ch] .. off[ch+1]-1 belong to chunk ch (1-based)
    const double k = (double)(i - off[ch]);
    ptab[i] = poisson_sf(k, bounds[ch - 1], sfe, 1.0);
}

// Critical counts of the table's chunks for one sig: kcrit[ch] = the smallest count k whose table entry 1 - cdf(k; chunk's upper
// bound) is <= sig (the entries fall with k; beyond the table they are 0), so that hpk_score looks a p-value up only where it can
// be <= sig.  sig >= 1: 0 (everything is looked up).
__global__ void __launch_bounds__(64) hpk_kcrit(const double* __restrict__ ptab, const int32_t* __restrict__ off, double sig, int32_t* __restrict__ kcrit) {
    const int ch = threadIdx.x + 1;
    if (threadIdx.x == 0) { kcrit[0] = 0; kcrit[HPK_NB_TAB + 1] = 0; }
    if (ch > HPK_NB_TAB) return;
    const int base = off[ch], len = off[ch + 1] - base;
    // (bisection: the entries fall with k - a walk from 0 was up to 33 000 dependent loads on one thread, 5.8 ms of the first call with a sig)
    int lo = -1, hi = len;                   // entry lo > sig (or lo = -1), entry hi <= sig (or hi = len: none is)
    while (hi - lo > 1) {
        const int mid = lo + (hi - lo) / 2;
        if (ptab[base + mid] <= sig) hi = mid; else lo = mid;
    }
    kcrit[ch] = hi;
}

// bhfdr scores every pixel at its own lambda = E, so its critical counts sit on a grid over lambda (HPK_KCL_*): kcrit[g] = the
// smallest count k with poisson_sf(k; lower edge of cell g) <= sig (1 + 1e-9).  The survival function grows with lambda, so a
// pixel of cell g whose count is below kcrit[g] has p > sig whatever its lambda inside the cell (the margin is nine orders of
// magnitude above the series' rounding) and keeps the placeholder 1 without the series being formed.  Bisection: p falls with k.
__global__ void __launch_bounds__(64) hpk_kcrit_lam(const double* __restrict__ sfe, double sig, int32_t* __restrict__ kcrit) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= HPK_KCL_N) return;
    const double lam = __longlong_as_double((long long)((unsigned long long)(g + HPK_KCL_G0) << 48));
    const double cap = sig * (1.0 + 1e-9);
    int lo = 0, hi = (int)(lam + 40.0 * sqrt(lam) + 64.0);          // p(hi) is 0 to the last bit
    if (poisson_sf((double)lo, lam, sfe, 1.0) <= cap) { kcrit[g] = 0; return; }
    while (hi - lo > 1) {                   // p(lo) > cap >= p(hi)
        const int mid = lo + (hi - lo) / 2;
        if (poisson_sf((double)mid, lam, sfe, 1.0) <= cap) hi = mid; else lo = mid;
    }
    kcrit[g] = hi;
}

// Fine p-value bins (bhfdr: one family of millions of tests per chromosome, whose cut the eight factor-4 bins of the
// lambda-chunk families bracket only within a factor ~20: 55 000 records copied back for 3 300 pixels): four bins per
// octave of x = sig / p, edges at 2^e x {1, 1.25, 1.5, 1.75}, the last bin open.  Any binning that is monotone in the
// rounded quotient keeps the cut's bound an over-count (thr_table_hist) and the survivors' bound safe (hpk_thr_compact).
#define HPK_FINE_BINS 64
__device__ __forceinline__ int fine_bin(double x, int nbins) {
    const unsigned long long xb = (unsigned long long)__double_as_longlong(x);
    int k = ((int)((xb >> 52) & 0x7ff) - 1023) * 4 + (int)((xb >> 50) & 3ull);
    k = (x == x) ? k : nbins - 1;                 // (0 / 0)
    return k < 0 ? 0 : (k > nbins - 1 ? nbins - 1 : k);
}
// ------------------------------------------------------------------ scoring
// Edge tables.  Clipping by one matrix end removes whole window rows (top: di < -e) or whole window columns (right:
// dj > e) whatever the diagonal, so the clipped window is again a 1-D stencil over IR: tap t = dj - di + 2wi carries
// the summed multiplicity of the surviving cells on that anti-diagonal.  One workgroup per (band, side, e, step) and 256
// diagonals: the taps are formed once in LDS (<= 2wi+1 cells each), then every thread runs <= 4wi+1 multiply-adds on an
// LDS-staged IR window.
// blockIdx.y >= 2 W nsteps: the unclipped (interior) table of step y - 2 W nsteps, same taps without clipping.
// blockIdx.y >= ntab: zero-fill duty (the band's counter block; saves a memset launch).  blockIdx.z = band of the batch.
#define HPK_ET_THREADS 256
__global__ void __launch_bounds__(HPK_ET_THREADS) hpk_etab_edge(const HpkDevPlan* __restrict__ plan, const HpkBandDesc* __restrict__ bands,
                                                                int ntab) {
    const HpkBandDesc* __restrict__ bd = bands + blockIdx.z;
    if ((int)blockIdx.y >= ntab) {
        const unsigned long long i = ((unsigned long long)(blockIdx.y - ntab) * gridDim.x + blockIdx.x) * HPK_ET_THREADS + threadIdx.x;
        if (i < bd->zero_bytes / 16) reinterpret_cast<uint4*>(gptr(bd->small))[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const double* __restrict__ IR = gptr(bd->IR);
    const int num = bd->num;
    const int D = plan->D, W = plan->W, mw = plan->mw, ns = plan->nsteps;
    const int d = blockIdx.x * HPK_ET_THREADS + threadIdx.x;
    int t = blockIdx.y;
    const bool interior = t >= 2 * W * ns;
    if (interior) t -= 2 * W * ns;
    const int s = t % ns; t /= ns;
    const int e = interior ? 0 : t % W, side = interior ? 2 : t / W;
    const HpkDevStep& st = plan->steps[s];
    const int wi = st.wi;
    __shared__ int lm[HPK_MAX_W + 1];                    // ring multiplicities of this step
    __shared__ double tapK[4 * HPK_MAX_W + 1], tapY[4 * HPK_MAX_W + 1];
    __shared__ double lir[HPK_ET_THREADS + 4 * HPK_MAX_W + 2];       // IR[d0 - 2wi .. d0 + 255 + 2wi] (0 outside [mw, num))
    if ((int)threadIdx.x <= HPK_MAX_W) lm[threadIdx.x] = st.m[threadIdx.x];
    const int d0 = blockIdx.x * HPK_ET_THREADS, kbase = d0 - 2 * wi;
    for (int i = threadIdx.x; i < HPK_ET_THREADS + 4 * wi + 1; i += HPK_ET_THREADS) {
        const int kk = kbase + i;
        lir[i] = (kk >= mw && kk < num) ? IR[kk] : 0.0;
    }
    __syncthreads();
    for (int tp = threadIdx.x; tp <= 4 * wi; tp += HPK_ET_THREADS) {
        const int off = tp - 2 * wi;                     // dj - di
        int ck = 0, cy = 0;
        for (int di = -wi; di <= wi; ++di) {
            const int dj = di + off;
            if (di == 0 || dj == 0 || dj < -wi || dj > wi) continue;
            if (side == 0 ? (e + di < 0) : (side == 1 && dj > e)) continue;   // rows above / columns right of the matrix
            const int adi = di < 0 ? -di : di, adj = dj < 0 ? -dj : dj;
            const int m = lm[adi > adj ? adi : adj];
            ck += m;
            if (di > 0 && dj < 0) cy += m;
        }
        tapK[tp] = (double)ck;
        tapY[tp] = (double)cy;
    }
    __syncthreads();
    if (d > D) return;
    double EK = 0.0, EY = 0.0;
    for (int tp = 0; tp <= 4 * wi; ++tp) {
        // (a diagonal without a single unmasked pixel has IR = NaN - far corners of short chromosomes: it may only reach
        // the sums of windows that really hold one of its cells, as in the reference's cell-by-cell adds; 0 x NaN is NaN)
        const double v = lir[(int)threadIdx.x + tp];
        const double ck = tapK[tp], cy = tapY[tp];
        EK += ck != 0.0 ? ck * v : 0.0;
        EY += cy != 0.0 ? cy * v : 0.0;
    }
    if (interior) {
        gptr(bd->etab)[(int64_t)(s * 2) * (D + 1) + d] = EK;
        gptr(bd->etab)[(int64_t)(s * 2 + 1) * (D + 1) + d] = EY;
        return;
    }
    const int64_t o = ((int64_t)((side * W + e) * ns + s) * 2) * (D + 1) + d;
    gptr(bd->eedge)[o] = EK;
    gptr(bd->eedge)[o + (D + 1)] = EY;
}

// Persistent blocks: block b walks rows b, b + gridDim.x, ...; the per-family counters live in LDS for the block's
// whole life and are flushed once.  Chunk boundaries sit in LDS; the chunk of E is 3 * exponent(E) plus two
// comparisons against the reference's own boundary values.
// blockIdx.y = band of the batch; a band's units are walked by the first `score_wgs` workgroups of its grid row.
#define HPK_BQ 128                      // hpk_score, bhfdr: entries of a wave's ring of pending pixels (< 64 left behind + 64 new)
template <bool BH, bool ONE>  // BH: bhfdr (one set, per-pixel lambda = E); otherwise hiccups (lambda chunks); ONE: a single (pw, ww) pair
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HPK_SCORE_WPE, HPK_SCORE_WPE))) hpk_score(HpkScoreArgs a, const HpkBandDesc* __restrict__ bands) {
    const HpkBandDesc* __restrict__ bd = bands + blockIdx.y;
    const int nwg = bd->score_wgs;
    if ((int)blockIdx.x >= nwg) return;
#ifdef HPK_PHASE_CLOCK
    const unsigned long long sck0 = __builtin_readcyclecounter();
    unsigned long long sck_items = 0ull;
#endif
    // per-pair counters: a single-pair launch (every hiccups() run with one (pw, ww), every bhfdr()) needs two sets only
    constexpr int NSETS_LDS = ONE ? 2 : 2 * HPK_MAX_PAIRS;
    __shared__ unsigned int lm[NSETS_LDS][HPK_NB + 1];
    __shared__ unsigned int lf[NSETS_LDS][HPK_NB + 1];
    __shared__ double lbounds[HPK_NB];
    __shared__ unsigned long long lemax[NSETS_LDS];
    __shared__ int lstepw[HPK_MAX_STEPS];
    __shared__ int lpair_slot[HPK_MAX_PAIRS], lpair_wi[HPK_MAX_PAIRS];
    __shared__ int lptoff[HPK_NB_TAB + 2];
    __shared__ int lkcrit[HPK_NB_TAB + 2];      // per chunk of the table: the smallest count whose p is <= sig (hpk_kcrit)
    // bhfdr: critical counts on the grid over lambda (hpk_kcrit_lam), and per wave a ring of the pixels whose series is still to
    // be formed - the few whose count reaches their cell's critical count wait here until a wave-full of them is there, so
    // that the series (the bulk of this kernel's arithmetic in this mode) runs on full waves, once per ~ten batches
    __shared__ int lkcl[BH ? HPK_KCL_N : 1];
    __shared__ double qE[BH ? 4 * HPK_BQ : 1];
    __shared__ float qO[BH ? 4 * HPK_BQ : 1];
    __shared__ int qr[BH ? 4 * HPK_BQ : 1], qc[BH ? 4 * HPK_BQ : 1];
    // Survivor records are only written for p-values at or below a per-family bound, given as a histogram bin (p <= sig
    // 4^-kmin): the cut of the chromosomes before lies orders of magnitude below sig, and 99 % of the p <= sig records
    // used to be written only for the compaction to drop them.  The histogram and the family counts still see every
    // p <= sig; hpk_thr_compact checks that the cut it derives lies inside the bound (HPK_OFF_SPECFAIL otherwise: the
    // host scores the chromosome once more without a bound).
    __shared__ unsigned char lkmin[NSETS_LDS * (HPK_NB + 1)];
    // p-values at or below sig by family and log bin (bin k: sig 4^-(k+1) < p <= sig 4^-k, the last one open towards 0):
    // what the Benjamini-Hochberg cut is derived from (hpk_thr_compact).  Families of the table's chunks are counted here,
    // [nsets][HPK_NB_TAB + 1][a.hbins], the few beyond it straight in global memory.
    extern __shared__ unsigned int lhist[];
    const HpkDevPlan* __restrict__ plan = a.plan;
    // Arguments that only rare paths need (re-read of a capped count, Poisson beyond the table, chunk bookkeeping) are
    // fetched from the kernel-argument segment where they are used instead of being held in SGPRs for the whole
    // kernel: with ~45 scalars live the compiler spilled ~80 of them to VGPR lanes, and the spill traffic
    // (v_readlane / v_writelane / s_nop) was 15 % of the instruction stream of this issue-bound kernel.
    const volatile HpkScoreArgs* ka = (const volatile HpkScoreArgs*)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
    const volatile HpkBandDesc* kb = bands + blockIdx.y;       // the band's rarely used fields, likewise
    unsigned char* const small = gptr(bd->small);
    const int64_t b_cap = bd->cap, b_rec_stride = bd->rec_stride;
    const unsigned* __restrict__ b_rec_ent = gptr(bd->rec_ent);
    const uint8_t* __restrict__ b_rec_W = gptr(bd->rec_W);
    const double2* __restrict__ b_rec_S = gptr(bd->rec_S);
    const uint2* __restrict__ b_units = gptr(bd->units);
    const double* __restrict__ b_IR = gptr(bd->IR);
    const double* __restrict__ b_b1 = gptr(bd->b1);
    const double* __restrict__ b_b2 = gptr(bd->b2);
    const double* __restrict__ b_etab = gptr(bd->etab);
    const double* __restrict__ b_eedge = gptr(bd->eedge);
    const int b_n = bd->n;
    const int b_J = bd->J, b_TR = bd->TR, b_TC = bd->TC, b_tilecap = bd->tilecap;      // the band's tile geometry
    HpkSurv* __restrict__ b_surv = gptr(bd->surv);
    unsigned long long* __restrict__ b_nsurv = reinterpret_cast<unsigned long long*>(small + HPK_OFF_NSURV);
    const int npairs = ONE ? 1 : plan->npairs;
    const int W = plan->W;
    const int nsets = BH ? 1 : 2 * npairs;
    for (int i = threadIdx.x; i < nsets * (HPK_NB + 1); i += blockDim.x) { (&lm[0][0])[i] = 0u; (&lf[0][0])[i] = 0u; }
    {
        const uint8_t* km = const_cast<const uint8_t*>(ka->kmin);
        for (int i = threadIdx.x; i < nsets * (HPK_NB + 1); i += blockDim.x) lkmin[i] = km ? km[i] : (unsigned char)0;
    }
    const int hbins = a.hbins;
    for (int i = threadIdx.x; i < nsets * (HPK_NB_TAB + 1) * hbins; i += blockDim.x) lhist[i] = 0u;
    const int sig_e = (int)((unsigned long long)__double_as_longlong(a.sig) >> 52);                  // sig > 0, normal
    const unsigned long long sig_m = (unsigned long long)__double_as_longlong(a.sig) & 0xfffffffffffffull;
    if (threadIdx.x < HPK_NB) lbounds[threadIdx.x] = const_cast<const double*>(ka->bounds)[threadIdx.x];
    if (threadIdx.x < NSETS_LDS) lemax[threadIdx.x] = 0ull;
    if (threadIdx.x < HPK_MAX_STEPS) lstepw[threadIdx.x] = (threadIdx.x < plan->nsteps) ? plan->steps[threadIdx.x].wi : 0;
    if (threadIdx.x < HPK_MAX_PAIRS) { lpair_slot[threadIdx.x] = plan->pair_slot[threadIdx.x]; lpair_wi[threadIdx.x] = plan->pair_wi[threadIdx.x]; }
    if (threadIdx.x < HPK_NB_TAB + 2) {
        lptoff[threadIdx.x] = const_cast<const int32_t*>(ka->ptab_off)[threadIdx.x];
        const int32_t* kc = BH ? nullptr : const_cast<const int32_t*>(ka->kcrit);
        lkcrit[threadIdx.x] = kc ? kc[threadIdx.x] : 0;          // (no table of critical counts: every p-value is looked up)
    }
    if (BH) {
        const int32_t* kc = const_cast<const int32_t*>(ka->kcrit);
        for (int i = threadIdx.x; i < HPK_KCL_N; i += blockDim.x) lkcl[i] = kc ? kc[i] : 0;      // (none: every series is formed)
    }
    // The width the widening froze at.  The stencil's workgroups added their resolve counts into the chromosome's totals;
    // every workgroup here replays the reference's decision on them (a handful of steps, one thread), the first one
    // also leaves totals, executed flags and the "empty step" verdict for the host.  (It used to be the tail of the
    // stencil kernel: drain, release, ticket, acquire, replay by the last workgroup - 5 us of every launch.)
    __shared__ unsigned long long lhtot[HPK_MAX_STEPS + 1];
    __shared__ int lslot[HPK_MAX_STEPS];
    __shared__ int lfrozen;
    if (threadIdx.x <= HPK_MAX_STEPS) lhtot[threadIdx.x] = hist_total(gptr(bd->hist_acc), (int)threadIdx.x);
    if (threadIdx.x < HPK_MAX_STEPS) lslot[threadIdx.x] = (threadIdx.x < plan->nsteps) ? plan->steps[threadIdx.x].slot : 0;
    __syncthreads();
    {
        if (threadIdx.x == 0) {
            int fw, e;
            freeze_replay(plan, lhtot, lstepw, lslot, blockIdx.x == 0 ? reinterpret_cast<int32_t*>(small + HPK_OFF_EXEC) : nullptr, fw, e);
            lfrozen = fw;
            if (blockIdx.x == 0) { *reinterpret_cast<int32_t*>(small + HPK_OFF_FROZEN) = fw; *reinterpret_cast<int32_t*>(small + HPK_OFF_ERR) = e; }
        }
        if (blockIdx.x == 0 && threadIdx.x <= HPK_MAX_STEPS) reinterpret_cast<unsigned long long*>(small + HPK_OFF_HIST)[threadIdx.x] = lhtot[threadIdx.x];
        __syncthreads();
    }

    const int lane = threadIdx.x & 63;
#ifdef HPK_PHASE_CLOCK
    const unsigned long long sck1 = __builtin_readcyclecounter();
#endif
    const int nsteps_u = plan->nsteps;
    const int frozen = lfrozen;
    const unsigned pkcap = (unsigned)plan->pk_cap;
    // Survivor slots are reserved from the global counter HPK_SCH records at a time per wave (same-address atomics
    // run at ~90 per microsecond device-wide); how many slots of a chunk were filled goes to chunk_used[].
    unsigned long long wbase = 0ull;
    unsigned wused = HPK_SCH;               // "no chunk yet"
    bool have_chunk = false;
    const int region = __builtin_amdgcn_readfirstlane((int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) % HPK_NREG));
    const int64_t rbase = (int64_t)region * b_cap;      // this wave's survivor region
    // Work unit = 4 consecutive 64-record batches of one tile's record region; units are dealt round-robin to all
    // waves of the grid (tiles differ a lot in candidate count), reads are coalesced.
    const unsigned nunits = *reinterpret_cast<const unsigned*>(small + HPK_OFF_NUNITS);     // work list appended by hpk_stencil: non-empty units only
    const unsigned gw = (unsigned)__builtin_amdgcn_readfirstlane((int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6)));    // wave-uniform, and known to be
    const unsigned nwv = (unsigned)(((int64_t)nwg * blockDim.x) >> 6);
    // One batch ahead.  While a batch is scored the records of the next one - of the same work unit or of the wave's
    // next unit - are on their way, and as soon as this batch's Poisson-table reads are issued the next batch's second
    // round of loads (IR, biases, the local-expected table entries of its first pair: all addressed from its record) is
    // requested behind them.  A batch then waits for one memory round trip, the table's, where it used to wait for
    // three in a row (records at the start of a unit, second round, table).
    // Several (pw, ww) pairs: a work item is one pair of one batch (pair after pair, then the unit's next batch), each
    // reading its own slot's record - the pairs share nothing but the entry, and one stream of single-pair items keeps
    // every item on the one-ahead schedule (a pair loop inside the batch exposed two round trips per pair).
    struct Geo { int r0, c0, cnt, i0, iend, pj, wi0; int64_t tbase0, sl; };
    const unsigned tstride = 2u * (unsigned)(a.D + 1);          // table offsets fit 32 bits: <= 2 * 20 * 64 * 2 * (D + 1) entries
    unsigned ri_b = 0u, ent_b = 0u;
    int stp_b = 0;
    double2 s2_b = make_double2(0.0, 0.0);
    double ir_b = 0.0, b1_b = 0.0, b2_b = 0.0, EK_b = 0.0, EY_b = 0.0;
    auto decode = [&](const uint2 un, Geo& g) {
        // (the tile by row block << 8 | column chunk: no division by the band's own J per unit)
        const int rb = (int)(un.x >> 8), cj = (int)(un.x & 255u), ub = (int)(un.y & 255u);
        const int tile = rb * b_J + cj;
        g.cnt = (int)(un.y >> 8);
        g.r0 = rb * b_TR;
        g.c0 = g.r0 + a.mw + cj * b_TC;
        g.i0 = ub * HPK_UNIT;
        g.iend = (g.i0 + HPK_UNIT < g.cnt) ? g.i0 + HPK_UNIT : g.cnt;
        g.tbase0 = (int64_t)tile * b_tilecap;
    };
    auto set_pair = [&](Geo& g, const int pj) {
        g.pj = pj;
        g.wi0 = __builtin_amdgcn_readfirstlane(lpair_wi[pj]);
        g.sl = (int64_t)__builtin_amdgcn_readfirstlane(lpair_slot[pj]) * b_rec_stride;
    };
    // first round: entry, first slot's step and sums (idle lanes read the tile's first record: always allocated, never used)
    // (several pairs: the items of one batch follow each other, pair after pair - the entry, and with it the pixel's 1-D expected
    // and biases in the second round, are loaded for the first pair and stay for the others)
    auto issue_records = [&](const Geo& g) {
        ri_b = (g.i0 + lane < g.cnt) ? (unsigned)(g.i0 + lane) : 0u;
        if (ONE || !HPK_SCORE_REUSE || g.pj == 0) ent_b = (b_rec_ent + g.tbase0)[ri_b];
        stp_b = (int)(b_rec_W + g.tbase0 + g.sl)[ri_b];
        s2_b = (b_rec_S + g.tbase0 + g.sl)[ri_b];
    };
    // second round of the batch whose records are in (ent_b, stp_b); stp_b becomes the step that counts (0: none)
    auto issue_round2 = [&](const Geo& g) {
        // (a lane beyond the unit's records holds the tile's first one - a pixel like any other, whose loads are in range: only
        //  its step is taken away, below)
        const bool cn = g.i0 + lane < g.cnt;
        const unsigned e = ent_b;
        const int r = g.r0 + (int)HPK_ENT_Y(e), c = g.c0 + (int)HPK_ENT_X(e), d = c - r;
        if (ONE || !HPK_SCORE_REUSE || g.pj == 0) {
            ir_b = b_IR[(unsigned)d];
            b2_b = b_b2[(unsigned)c];
            b1_b = b_b1[(unsigned)r];
        }
        // (windows clipped by one matrix end - pixels within maxww of it - take their expected sums from the edge tables: rare, and
        //  only in tiles that reach into the first maxww rows or the last maxww columns - the unit's scalars say so)
        const double* __restrict__ tab = b_etab;
        unsigned tbase = 0u;
        if (g.r0 < W || g.c0 + b_TC > b_n - W) {
            const bool top = r < W, right = c >= b_n - W;
            if (__ballot(top != right) != 0ull) {
                tab = (top != right) ? b_eedge : b_etab;
                tbase = (top != right) ? (unsigned)(((top ? 0 : 1) * W + (top ? r : b_n - 1 - c)) * nsteps_u) * tstride : 0u;
            }
        }
        int stp = cn ? stp_b : 0;
        const int stepw = lstepw[stp > 0 ? stp - 1 : 0];
        asm volatile("" : "+v"(stp));
        stp = d >= g.wi0 ? stp : 0;
        asm volatile("" : "+v"(stp));
        stp = stepw <= frozen ? stp : 0;
        asm volatile("" : "+v"(stp));
        const unsigned srow = (unsigned)(stp > 1 ? stp - 1 : 0);
        const unsigned to = tbase + srow * tstride + (unsigned)d;
        EK_b = tab[to];
        EY_b = tab[to + (unsigned)(a.D + 1)];
        stp_b = stp;
    };
    // The survivors of a wave's pixels (wave-uniform call: surv somewhere): family count, histogram bin, record.
    auto survivors = [&](const bool surv, const int set, const int chunk, const int flag, const int r, const int c, const float rawpix,
                         const double E, const double p) {
        bool wr = surv;
        if (surv) {
            atomicAdd(&lf[set][chunk], 1u);
            if (hbins) {
                // floor(log2(sig / p)) from the two exponents and a mantissa compare (p = 0, subnormal: last bin)
                const unsigned long long pb = (unsigned long long)__double_as_longlong(p);
                int k = sig_e - (int)(pb >> 52) - ((pb & 0xfffffffffffffull) > sig_m ? 1 : 0);
                k >>= HPK_HSHIFT;                                 // bins a factor 2^(2^HPK_HSHIFT) wide
                k = (pb >> 52) == 0ull ? hbins - 1 : k;
                k = k < 0 ? 0 : (k > hbins - 1 ? hbins - 1 : k);
                if (BH && hbins > 16) k = fine_bin(a.sig / p, hbins);     // (bhfdr: four bins per octave)
                if (chunk <= HPK_NB_TAB) atomicAdd(&lhist[(set * (HPK_NB_TAB + 1) + chunk) * hbins + k], 1u);
                else atomicAdd(&gptr(kb->cnt)[(set * (HPK_NB + 1) + chunk) * hbins + k], 1u);
                wr = k >= (int)lkmin[set * (HPK_NB + 1) + chunk];
            }
        }
        const unsigned long long wm = __ballot(wr);
        if (wm == 0ull) return;
        // per-wave reservation of survivor slots
        const unsigned scnt = (unsigned)__popcll(wm);
        if (wused + scnt > HPK_SCH) {            // wave-uniform: retire the chunk, take a new one
            if (have_chunk && lane == 0 && (int64_t)wbase < b_cap) gptr(kb->chunk_used)[(rbase + (int64_t)wbase) / HPK_SCH] = wused;
            have_chunk = true;
            unsigned long long nb = 0ull;
            if (lane == 0) nb = atomicAdd(&b_nsurv[region * HPK_REG_STRIDE], (unsigned long long)HPK_SCH);
            wbase = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)nb) |
                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(nb >> 32)) << 32;
            wused = 0u;
        }
        const unsigned long long basei = wbase + wused;
        wused += scnt;
        if (wr) {
            const unsigned long long idx = basei + (unsigned long long)__popcll(wm & ((1ull << lane) - 1ull));
            if ((int64_t)idx < b_cap) {
                HpkSurv rec;
                rec.x = r; rec.y = c; rec.O = rawpix; rec.set = (uint8_t)set; rec.chunk = (uint8_t)chunk;
                rec.flag = (uint8_t)flag;                        // callers.py:330
                rec.pad = 0; rec.E = E; rec.p = p; rec.bal = 0.0;       // balanced value: filled by hpk_thr_compact
                b_surv[rbase + (int64_t)idx] = rec;
            }
        }
    };
    // bhfdr: the wave's ring of pending pixels (qhead / qcount are wave-uniform)
    const int qbase = BH ? (int)(threadIdx.x >> 6) * HPK_BQ : 0;
    int qhead = 0, qcount = 0;
    auto drain = [&](const int nq) {                   // the series of the ring's first nq <= 64 pixels, their survivors
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool act = lane < nq;
        const int qi = qbase + ((qhead + lane) & (HPK_BQ - 1));
        const double E = act ? qE[qi] : 1.0;
        const float o = act ? qO[qi] : 0.0f;
        const int rr = act ? qr[qi] : 0, cc = act ? qc[qi] : 0;
        double p = 1.0;
        if (act) p = poisson_sf((double)o, E, const_cast<const double*>(ka->sfe), a.sig);      // callers.py:536-540
        const bool surv = act && p <= a.sig;
        if (__ballot(surv) != 0ull) survivors(surv, 0, 1, rr < 0 ? 1 : 0, rr & 0x7fffffff, cc, o, E, p);
        qhead = (qhead + nq) & (HPK_BQ - 1);
        qcount -= nq;
        __builtin_amdgcn_wave_barrier();
    };
    // Emax of a set (E > 0: a double's order; a negative or NaN E never wins a max against a non-negative one).  One pair (two
    // sets) or bhfdr (one): a running maximum per lane - one v_max_f64 per set and item -, folded over the wave and into the
    // block's LDS word when the wave runs out of work.  Several pairs: the set changes with the item (see below).
    double em0 = 0.0, em1 = 0.0;
    const bool may_both = b_n < 2 * W + a.D + 2;        // (r < W, c >= n - W and c - r <= D need n < 2 W + D)
    if (gw < nunits) {
        Geo gn;
        decode(b_units[gw], gn);
        set_pair(gn, 0);
        unsigned u = gw;
        uint2 un_next = (u + nwv < nunits) ? b_units[u + nwv] : make_uint2(0u, 0u);
        issue_records(gn);
        issue_round2(gn);
        bool more = true;
        while (more) {
#ifdef HPK_PHASE_CLOCK
            ++sck_items;
#endif
            const Geo g = gn;
            // (lanes beyond the unit's records hold the tile's first entry and step 0 - issue_round2 -: nothing of theirs is looked at)
            const unsigned ent = ent_b;
            const int stp0 = stp_b;
            const double2 s20 = s2_b;
            const double ir = ir_b, b1r = b1_b, b2c = b2_b, EK0 = EK_b, EY0 = EY_b;
            // This item's pixel: only where it is needed (a survivor, a capped count, bhfdr's ring) - the common path never looks
            const int g_r0 = g.r0, g_c0 = g.c0;
            auto pix_r = [&]() { return g_r0 + (int)HPK_ENT_Y(ent); };
            auto pix_c = [&]() { return g_c0 + (int)HPK_ENT_X(ent); };
            const int r_bh = BH ? pix_r() : 0, c_bh = BH ? pix_c() : 0;     // (bhfdr parks 3-8 % of its pixels in the ring: most items need them)
            // ... and - short chromosomes only: a window is clipped by both matrix ends where r < W and c >= n - W, with c - r <= D -
            // the explicit expected sums of such windows: a call, placed ahead of the next item's loads so that nothing is in
            // flight (and nothing has to be parked in scratch) across it.
            double EK = EK0, EY = EY0;
            if (may_both) {
                const int r = pix_r(), c = pix_c();
                const bool bo = (stp0 != 0) && r < W && c >= b_n - W;
                if (__ballot(bo) != 0ull) {
                    if (bo) {
                        const double2 ee = edge_expected(plan->steps[stp0 - 1].m, plan->steps[stp0 - 1].wi, b_IR, r, c, b_n, kb->num, a.mw);
                        EK = ee.x; EY = ee.y;
                    }
                }
            }
            // the batch after this one: the same unit's next 64 records or the first of the wave's next unit
            bool next_batch = true;
            if (!ONE) {
                next_batch = gn.pj + 1 >= npairs;
                set_pair(gn, next_batch ? 0 : gn.pj + 1);
            }
            if (next_batch) {
                gn.i0 += 64;
                if (gn.i0 >= gn.iend) {
                    u += nwv;
                    more = u < nunits;
                    if (more) {
                        decode(un_next, gn);
                        if (u + nwv < nunits) un_next = b_units[u + nwv];
                    }
                }
            }
            if (more) issue_records(gn);
            // nothing in this item counts (every record resolved beyond the width the widening froze at): next
            if (__ballot(stp0 != 0) == 0ull) {
                if (more) issue_round2(gn);
                continue;
            }
            // the pixel's count: the entry's field - the SAT holds counts capped at HPK_PK_CAP: those are re-read, behind one ballot -
            // as the integer the table is indexed by; as a float / double only where a survivor, the ring or the series wants it
            const unsigned cntu = ent >> HPK_ENT_CNT_SHIFT;
            int kO = (int)cntu;
            float rawre = 0.f;
            if (__ballot(cntu >= pkcap) != 0ull) {
                if (stp0 != 0 && cntu >= pkcap) { const int r = pix_r(); rawre = gptr(kb->raw)[(int64_t)r * kb->ld + (pix_c() - r)]; kO = (int)(double)rawre; }
            }
            auto rawpix_of = [&]() { return (stp0 != 0 && cntu >= pkcap) ? rawre : (float)cntu; };
            {
                // The scalar unit is this kernel's busiest one (lane masks ANDed and ORed, exec saved and restored around
                // every divergent if): conditions are folded into the values - a record that does not count turns into
                // step 0, expected sum 0, E = 0 (issue_round2) - so that each decision is one compare feeding one select.
                const int pj = ONE ? 0 : g.pj;
                const int stp = stp0;                   // resolved at an executed step, far enough from the diagonal, or 0
                const double2 s2 = s20;
                EK = stp != 0 ? EK : 0.0;
                EY = stp != 0 ? EY : 0.0;
                // callers.py:244-249: E = ((IR[d] * (bS / bE)) * B1[x]) * B2[y] where bE != 0
                const double eK = (EK != 0.0) ? ((ir * (s2.x / EK)) * b1r) * b2c : 0.0;
                const double eY = (EY != 0.0) ? ((ir * (s2.y / EY)) * b1r) * b2c : 0.0;
                constexpr int nfl = BH ? 1 : 2;
                int chunk2[2] = {0, 0};
                double p2[2] = {1.0, 1.0};
                if (BH) {
                    if (more) issue_round2(gn);                 // (bhfdr: one pair; everything below is arithmetic and LDS)
                    // callers.py:517-540.  The family's size counts every valid pixel; only p <= sig is ever looked at beyond that, and
                    // a pixel whose count stays below the critical count of its lambda's cell (hpk_kcrit_lam) has p > sig: it keeps
                    // the placeholder.  The others - a few per batch - queue up for their series (drain).
                    const bool valid = eK > 0.0;
                    const unsigned long long vm = __ballot(valid);
                    if (vm != 0ull) {
                        asm("v_max_f64 %0, %0, %1" : "+v"(em0) : "v"(eK));
                        if (lane == 0) atomicAdd(&lm[0][1], (unsigned)__popcll(vm));            // one family: chunk 1
                        const unsigned gi = (unsigned)((int)((unsigned long long)__double_as_longlong(eK) >> 48) - HPK_KCL_G0);
                        const int kc = (valid && gi < (unsigned)HPK_KCL_N) ? lkcl[gi] : 0;        // (outside the grid: formed)
                        const bool need = valid && kO >= kc;
                        const unsigned long long nm = __ballot(need);
                        if (nm != 0ull) {
                            if (need) {
                                const int qi = qbase + ((qhead + qcount + (int)__popcll(nm & ((1ull << lane) - 1ull))) & (HPK_BQ - 1));
                                qE[qi] = eK; qO[qi] = rawpix_of(); qr[qi] = r_bh | (eY == 0.0 ? (int)0x80000000 : 0); qc[qi] = c_bh;
                            }
                            qcount += (int)__popcll(nm);
                            if (qcount >= 64) drain(64);
                        }
                    }
                    continue;
                } else {
                    // Chunk of E: boundaries lbounds[i] = 2^(i/3); membership is strict on both sides (callers.py:38), so E
                    // sitting on a boundary belongs to no chunk.  E in [2^k, 2^(k+1)) has lbounds[3k .. 3k+2] at or below
                    // it; E < 1 is chunk 1.  The common case - below 2^15 (the Poisson table's range) and not on a boundary -
                    // is straight-line code for both expected values, whose two table reads then travel together; one
                    // ballot sends the rest of a batch through the general rules.
                    unsigned odd = 0u;
                    unsigned long long oddm = 0ull;
                    int len2[2];
                    unsigned at2[2];
                    bool crit2[2];
#pragma unroll
                    for (int fl = 0; fl < 2; ++fl) {
                        const double E = fl ? eY : eK;
                        const int ex = (int)((unsigned long long)__double_as_longlong(E) >> 52) - 1023;     // E = 0: -1023
                        int i0b = 3 * ex;
                        i0b = i0b < 0 ? 0 : i0b;
                        i0b = i0b > HPK_NB - 3 ? HPK_NB - 3 : i0b;              // clamped: every lane reads inside the table
                        const double bq = lbounds[i0b], bA = lbounds[i0b + 1], bB = lbounds[i0b + 2];
                        int ch = 3 * ex + 2 + (E >= bA ? 1 : 0) + (E >= bB ? 1 : 0);
                        ch = ex >= 0 ? ch : 1;
                        asm volatile("" : "+v"(ch));
                        ch = E > 0.0 ? ch : 0;
#if HPK_SCORE_ODD_MASK
                        oddm |= __ballot((ex >= 15) | (E == bq) | (E == bA) | (E == bB));     // lambda beyond the table, or on a boundary
#else
                        asm volatile("" : "+v"(odd));
                        odd = ex >= 15 ? 1u : odd;                             // lambda beyond the table
                        asm volatile("" : "+v"(odd));
                        odd = E == bq ? 1u : odd;
                        asm volatile("" : "+v"(odd));
                        odd = E == bA ? 1u : odd;
                        asm volatile("" : "+v"(odd));
                        odd = E == bB ? 1u : odd;
                        asm volatile("" : "+v"(odd));
#endif
                        chunk2[fl] = ch;
                        const int ct = ch > 1 ? ch : 1;
                        const int base = lptoff[ct];
                        len2[fl] = lptoff[ct + 1] - base;
                        at2[fl] = (unsigned)(base + (kO < len2[fl] ? kO : 0));
                        crit2[fl] = kO >= lkcrit[ct];
                    }
                    if ((HPK_SCORE_ODD_MASK ? oddm : __ballot(odd != 0u)) == 0ull) {
                        // Only p <= sig is ever looked at (the family sizes count every valid pixel), p falls with the count, and
                        // sig is one number per call: a pixel whose count stays below its chunk's critical count - the smallest
                        // one whose table entry is <= sig (hpk_kcrit) - keeps the placeholder 1, and the table - a third
                        // dependent round trip to memory per batch - is read for the one pixel in a thousand that can survive.
                        if (more) issue_round2(gn);
                        if (__ballot(crit2[0] | crit2[1]) != 0ull) {
                            if (crit2[0]) p2[0] = kO < len2[0] ? a.ptab[at2[0]] : 0.0;
                            if (crit2[1]) p2[1] = kO < len2[1] ? a.ptab[at2[1]] : 0.0;
                        }
                    } else {
#pragma unroll 1
                        for (int fl = 0; fl < 2; ++fl) {
                            const double E = fl ? eY : eK;
                            const bool valid = E > 0.0;
                            const int e3 = 3 * ((int)((unsigned long long)__double_as_longlong(E) >> 52) - 1023);
                            const bool ge1 = valid && E >= 1.0;
                            const bool big = e3 + 2 >= HPK_NB;
                            const int i0b = (ge1 && !big) ? e3 : 0;
                            const double bq = lbounds[i0b], bA = lbounds[i0b + 1], bB = lbounds[i0b + 2];
                            const int lo = ge1 ? (big ? HPK_NB : e3 + 1 + (E >= bA ? 1 : 0) + (E >= bB ? 1 : 0)) : 0;
                            const bool onb = ge1 && !big && (E == bq || E == bA || E == bB);
                            const bool inch = valid && lo < HPK_NB && !onb;
                            const int chunk = inch ? lo + 1 : 0;
                            const bool tabd = inch && chunk <= HPK_NB_TAB;
                            const int ct = tabd ? chunk : 1;
                            const int base = lptoff[ct], len = lptoff[ct + 1] - base;
                            double p = 1.0;
                            if (tabd) p = (kO < len) ? a.ptab[(unsigned)(base + kO)] : 0.0;
                            const bool rare = inch && !tabd;                      // lambda > 2^15: beyond the table
                            if (__ballot(rare) != 0ull) {
                                if (rare) p = poisson_sf((double)rawpix_of(), lbounds[chunk - 1], const_cast<const double*>(ka->sfe), a.sig);   // callers.py:268-270
                            }
                            if (fl == 0) { chunk2[0] = chunk; p2[0] = p; } else { chunk2[1] = chunk; p2[1] = p; }
                        }
                        if (more) issue_round2(gn);
                    }
                }
#pragma unroll
                for (int fl = 0; fl < nfl; ++fl) {
                    const int set = BH ? 0 : pj * 2 + fl;
                    const double E = fl ? eY : eK;
                    const bool valid = E > 0.0;                               // callers.py:250 (E = 0 where the record does not count)
                    const int chunk = chunk2[fl];
                    const double p = p2[fl];
                    // only p <= sig can reach q <= sig; a pixel without a chunk keeps p = 1 (callers.py:259-260); chunk != 0 implies E > 0
                    const bool surv = (p <= a.sig) & (chunk != 0);
                    const unsigned long long vm = __ballot(valid);
                    const unsigned long long sm = __ballot(surv);
                    if (vm != 0ull) {
                        // Emax of the set: E > 0, so its bit pattern orders like its value (E = 0: never above).  The
                        // block's running maximum settles after a few batches; only lanes that beat it touch the LDS atomic.
                        // (a negative E - negative balanced values - is not valid and must not enter: its sign bit would win)
                        if (ONE) {
                            if (fl == 0) asm("v_max_f64 %0, %0, %1" : "+v"(em0) : "v"(E)); else asm("v_max_f64 %0, %0, %1" : "+v"(em1) : "v"(E));
                        } else {
                            const unsigned long long ebits = valid ? (unsigned long long)__double_as_longlong(E) : 0ull;
                            const bool beats = ebits > lemax[set];
                            if (__ballot(beats) != 0ull) { if (beats) atomicMax(&lemax[set], ebits); }
                        }
                        // tests per family; family 0 of a set collects its valid pixels without a chunk (the host adds the
                        // families up to the set's number of valid pixels).  (One LDS atomic per lane, onto the handful of words a
                        // wave's chunks share: served one lane at a time, and still the cheapest form - one add of the lanes' count
                        // per distinct chunk, a leader's chunk at a time, costs the kernel 5 % with one round and 17 % with three:
                        // it is bound by the instructions it issues, and the LDS pipe has nothing else to do here.  Round 5,
                        // profiles/r05_score_ab.txt.)
                        if (valid) atomicAdd(&lm[set][chunk], 1u);
                    }
                    if (sm != 0ull) survivors(surv, set, chunk, (fl == 0 && eY == 0.0) ? 1 : 0, pix_r(), pix_c(), rawpix_of(), E, p);
                }
            }
        }
        if (BH) { while (qcount > 0) drain(qcount < 64 ? qcount : 64); }
        if (ONE) {
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) {
                em0 = fmax(em0, __shfl_xor(em0, m));
                if (!BH) em1 = fmax(em1, __shfl_xor(em1, m));
            }
            if (lane == 0) {
                if (em0 > 0.0) atomicMax(&lemax[0], (unsigned long long)__double_as_longlong(em0));
                if (!BH && em1 > 0.0) atomicMax(&lemax[1], (unsigned long long)__double_as_longlong(em1));
            }
        }
    }
    if (have_chunk && lane == 0 && (int64_t)wbase < b_cap) gptr(kb->chunk_used)[(rbase + (int64_t)wbase) / HPK_SCH] = wused;
#ifdef HPK_PHASE_CLOCK
    const unsigned long long sck2 = __builtin_readcyclecounter();
#endif
    __syncthreads();
    for (int i = threadIdx.x; i < nsets * (HPK_NB + 1); i += blockDim.x) {
        const unsigned v = (&lm[0][0])[i], f = (&lf[0][0])[i];
        if (v) atomicAdd(&reinterpret_cast<unsigned int*>(small + HPK_OFF_FAM_M)[i], v);
        if (f) atomicAdd(&reinterpret_cast<unsigned int*>(small + HPK_OFF_FAM_F)[i], f);
    }
    if (threadIdx.x < nsets) {
        if (lemax[threadIdx.x]) atomicMax(&reinterpret_cast<unsigned long long*>(small + HPK_OFF_EMAX)[threadIdx.x], lemax[threadIdx.x]);
    }
    for (int i = threadIdx.x; i < nsets * (HPK_NB_TAB + 1) * hbins; i += blockDim.x) {
        const unsigned v = lhist[i];
        if (v) {
            const int k = i % hbins, fc = i / hbins, ch = fc % (HPK_NB_TAB + 1), st = fc / (HPK_NB_TAB + 1);
            atomicAdd(&gptr(kb->cnt)[(st * (HPK_NB + 1) + ch) * hbins + k], v);
        }
    }
#ifdef HPK_PHASE_CLOCK
    if (a.clk && lane == 0) {
        const unsigned long long sck3 = __builtin_readcyclecounter();
        atomicAdd(&a.clk[0], sck1 - sck0); atomicAdd(&a.clk[1], sck2 - sck1); atomicAdd(&a.clk[2], sck3 - sck2);
        atomicAdd(&a.clk[3], 1ull); atomicAdd(&a.clk[4], sck_items); atomicMax(&a.clk[5], sck3 - sck0);
        atomicMin(&a.clk[6], sck0); atomicMax(&a.clk[7], sck3);
    }
#endif
}

// ------------------------------------------------------------------ BH cut tightening on the survivor list
// Benjamini-Hochberg rejects the k* smallest p-values of a family of m tests, k* = max{k : p_(k) <= sig k / m}.
// With F(t) = #{p <= t}: k* <= F(sig), hence p_(k*) <= sig F(sig) / m =: T1 <= sig, then p_(k*) <= sig F(T1) / m, ...
// Every p-value above the current bound is irrelevant both for the rejection set and for the q-values of the
// rejected ones (their step-up terms exceed sig), so the list can be cut to p <= T before it leaves the device.
// Round k (k = 0 .. rounds-1) counts F(T_k) into cnt[k][.], where T_0 = sig f / m (f = #{p <= sig}, counted by the
// scoring kernel) and T_k = sig cnt[k-1] / m.  Every workgroup derives the bounds it needs from the previous rounds'
// counters in its prologue, so a round is one launch and the last bound is applied by the compaction itself.
__device__ __forceinline__ void thr_table(double* lthr, const unsigned int* __restrict__ fam_m, const unsigned int* __restrict__ fam_f,
                                          const unsigned int* __restrict__ cnt, int round, double sig, int nfam) {
    for (int i = threadIdx.x; i < nfam; i += blockDim.x) {
        const unsigned m = fam_m[i];
        double t = 0.0;
        if (m) {
            t = fmin(sig, sig * ((double)fam_f[i] / (double)m) * (1.0 + 1e-9));
            for (int k = 0; k < round; ++k) t = fmin(t, sig * ((double)cnt[(size_t)k * HPK_NFAM + i] / (double)m) * (1.0 + 1e-9));
        }
        lthr[i] = t;
    }
}
// One-pass variant of the tightening rounds: instead of counting F(T_k) exactly, one launch per round, ONE launch
// histograms the p-values of every family below T_0 = sig F(sig) / m on a log2 scale (bin k: T_0 2^-(k+1) < p <= T_0 2^-k,
// the last bin open towards 0), and the compaction derives its bound from the histogram: T <- sig C(T) / m with C(T) the
// count up to the first bin edge at or above T - an over-count, so the bound stays above the true Benjamini-Hochberg cut
// (every p-value it drops has a step-up term above sig) while coming within a factor 2 of the exact fixed point.
__device__ __forceinline__ double thr_t0(unsigned m, unsigned f, double sig) {
    return m ? fmin(sig, sig * ((double)f / (double)m) * (1.0 + 1e-9)) : 0.0;
}
// The survivor-list kernels: blockIdx.z = band of the batch, blockIdx.y = survivor region.
#define HPK_THR_BAND_ARGS                                                                                              \
    const HpkBandDesc* __restrict__ bd = bands + blockIdx.z;                                                          \
    unsigned char* const small = gptr(bd->small);                                                                    \
    const HpkSurv* __restrict__ surv = gptr(bd->surv);                                                                \
    const unsigned long long* __restrict__ nsurv = reinterpret_cast<const unsigned long long*>(small + HPK_OFF_NSURV); \
    const int64_t cap = bd->cap;                                                                                       \
    const unsigned* __restrict__ chunk_used = gptr(bd->chunk_used);                                                    \
    const unsigned int* __restrict__ fam_m = reinterpret_cast<const unsigned int*>(small + HPK_OFF_FAM_M);             \
    const unsigned int* __restrict__ fam_f = reinterpret_cast<const unsigned int*>(small + HPK_OFF_FAM_F);
__global__ void __launch_bounds__(256) hpk_thr_hist(const HpkBandDesc* __restrict__ bands, int nbins, double sig, int nfam) {
    HPK_THR_BAND_ARGS
    unsigned int* __restrict__ hist = gptr(bd->cnt);
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    double* lt0 = reinterpret_cast<double*>(hsm);                        // [nfam]
    unsigned* lh = reinterpret_cast<unsigned*>(hsm + (size_t)nfam * 8);   // [nfam][nbins]
    const int reg = blockIdx.y;
    int64_t n = (int64_t)nsurv[reg * HPK_REG_STRIDE]; if (n > cap) n = cap;
    if ((int64_t)blockIdx.x * blockDim.x >= n) return;
    for (int i = threadIdx.x; i < nfam; i += blockDim.x) lt0[i] = thr_t0(fam_m[i], fam_f[i], sig);
    for (int i = threadIdx.x; i < nfam * nbins; i += blockDim.x) lh[i] = 0u;
    __syncthreads();
    const int64_t rb = (int64_t)reg * cap;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if ((unsigned)(i & (HPK_SCH - 1)) >= chunk_used[(rb + i) >> HPK_SCH_LOG2]) continue;
        const HpkSurv& rec = surv[rb + i];
        const int f = (int)rec.set * (HPK_NB + 1) + (int)rec.chunk;
        const double p = rec.p, t0 = lt0[f];
        if (p <= t0) {
            int k = nbins - 1;
            if (p > 0.0) {                 // exponent of t0 / p: the quotient never rounds below a power of two it reaches
                k = (int)((__double_as_longlong(t0 / p) >> 52) & 0x7ff) - 1023;
                k = k < 0 ? 0 : (k > nbins - 1 ? nbins - 1 : k);
            }
            atomicAdd(&lh[f * nbins + k], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nfam * nbins; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
// bound per family from the histogram (see hpk_thr_hist; absolute: the bins of hpk_score, edges sig 2^-k instead of T_0 2^-k)
// (fine bins - bhfdr: one family of millions of tests, 64 bins - take a wave per family with tests, a lane per bin: the thread per
// family below walked 64 bins of global memory per round of its fixed-point iteration, one dependent load after the other, in
// every workgroup of the compaction - 0.6 ms per 64 chromosomes)
__device__ __forceinline__ void thr_table_hist_fine(double* lthr, const unsigned int* __restrict__ fam_m, const unsigned int* __restrict__ fam_f,
                                                    const unsigned int* __restrict__ hist, int nbins, double sig, int nfam) {
    __shared__ int nact, act[HPK_NFAM];
    if (threadIdx.x == 0) nact = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nfam; i += blockDim.x) {
        const unsigned m = fam_m[i];
        const double t0 = thr_t0(m, fam_f[i], sig);
        lthr[i] = t0;
        if (m && t0 > 0.0) act[atomicAdd(&nact, 1)] = i;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int q = wave; q < nact; q += nw) {
        const int i = act[q];
        const double m = (double)fam_m[i];
        // suf = p-values in the bins from this lane's on (whole numbers below 2^53: any order of adding them is exact)
        double suf = lane < nbins ? (double)hist[(size_t)i * nbins + lane] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double o = __shfl_down(suf, d);
            suf += lane + d < 64 ? o : 0.0;
        }
        double t = lthr[i];
        for (int it = 0; it < 2 * nbins; ++it) {
            const int k = fine_bin(sig / (t * (1.0 + 1e-12)), nbins);
            const double c = __shfl(suf, k);
            const double tn = fmin(t, sig * (c / m) * (1.0 + 1e-9));
            if (!(tn < t)) break;
            t = tn;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) lthr[i] = t;
    }
}
__device__ __forceinline__ void thr_table_hist(double* lthr, const unsigned int* __restrict__ fam_m, const unsigned int* __restrict__ fam_f,
                                               const unsigned int* __restrict__ hist, int nbins, double sig, int nfam, bool absolute) {
    if (absolute && nbins > 16) { thr_table_hist_fine(lthr, fam_m, fam_f, hist, nbins, sig, nfam); return; }
    for (int i = threadIdx.x; i < nfam; i += blockDim.x) {
        const unsigned m = fam_m[i];
        const double t0 = thr_t0(m, fam_f[i], sig);
        const double ref = absolute ? sig : t0;
        double t = t0;
        if (m && t0 > 0.0) {
            for (int it = 0; it < 2 * nbins; ++it) {
                // largest k whose edge ref 2^-k is still >= t (with a margin for the rounding of the quotient)
                int k;
                if (absolute && nbins > 16) k = fine_bin(ref / (t * (1.0 + 1e-12)), nbins);
                else {
                    k = (int)((__double_as_longlong(ref / (t * (1.0 + 1e-12))) >> 52) & 0x7ff) - 1023;
                    if (absolute) k >>= HPK_HSHIFT;
                    k = k < 0 ? 0 : (k > nbins - 1 ? nbins - 1 : k);
                }
                unsigned long long c = 0ull;
                for (int kk = k; kk < nbins; ++kk) c += hist[(size_t)i * nbins + kk];
                const double tn = fmin(t, sig * ((double)c / (double)m) * (1.0 + 1e-9));
                if (!(tn < t)) break;
                t = tn;
            }
        }
        lthr[i] = t;
    }
}
__global__ void __launch_bounds__(256) hpk_thr_count(const HpkBandDesc* __restrict__ bands, int round, double sig, int nfam) {
    HPK_THR_BAND_ARGS
    unsigned int* __restrict__ cnt = gptr(bd->cnt);
    __shared__ unsigned int lc[HPK_NFAM];
    __shared__ double lthr[HPK_NFAM];
    const int reg = blockIdx.y;             // one grid row per survivor region
    int64_t n = (int64_t)nsurv[reg * HPK_REG_STRIDE]; if (n > cap) n = cap;
    if ((int64_t)blockIdx.x * blockDim.x >= n) return;
    for (int i = threadIdx.x; i < nfam; i += blockDim.x) lc[i] = 0u;
    thr_table(lthr, fam_m, fam_f, cnt, round, sig, nfam);
    __syncthreads();
    {
        const int64_t rb = (int64_t)reg * cap;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            if ((unsigned)(i & (HPK_SCH - 1)) >= chunk_used[(rb + i) >> HPK_SCH_LOG2]) continue;
            const HpkSurv& rec = surv[rb + i];
            const int f = (int)rec.set * (HPK_NB + 1) + (int)rec.chunk;
            if (rec.p <= lthr[f]) atomicAdd(&lc[f], 1u);
        }
    }
    __syncthreads();
    unsigned int* out = cnt + (size_t)round * HPK_NFAM;
    for (int i = threadIdx.x; i < nfam; i += blockDim.x) if (lc[i]) atomicAdd(&out[i], lc[i]);
}
// The first `inl` survivors of the cut go to out_head (which travels to the host together with the counters), the rest
// to out_rest.
__global__ void __launch_bounds__(256) hpk_thr_compact(const HpkBandDesc* __restrict__ bands, int rounds, double sig, int nfam,
                                                       const uint8_t* __restrict__ kmin) {
    HPK_THR_BAND_ARGS
    const unsigned int* __restrict__ cnt = gptr(bd->cnt);
    HpkSurv* __restrict__ out_head = reinterpret_cast<HpkSurv*>(small + bd->off_inl);
    const unsigned long long inl = HPK_HEAD_INLINE;
    HpkSurv* __restrict__ out_rest = gptr(bd->surv2);
    unsigned long long* __restrict__ nout = reinterpret_cast<unsigned long long*>(small + HPK_OFF_NOUT);
    const double* __restrict__ bal = gptr(bd->bal);
    const double* __restrict__ weight = gptr(bd->weight);
    const int64_t ld = bd->ld;
    __shared__ double lthr[HPK_NFAM];
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int reg = blockIdx.y;
    int64_t n = (int64_t)nsurv[reg * HPK_REG_STRIDE]; if (n > cap) n = cap;
    // the band's first workgroup also reports, per family, the histogram bin its cut falls into (what the next chromosomes'
    // scoring bounds its survivor records with) and whether a cut lies above the bound this chromosome's were written to
    const bool lead = blockIdx.x == 0 && blockIdx.y == 0 && rounds <= -100;
    if (!lead && (int64_t)blockIdx.x * blockDim.x >= n) return;
    if (rounds <= -100) thr_table_hist(lthr, fam_m, fam_f, cnt, -rounds - 100, sig, nfam, true);
    else if (rounds < 0) thr_table_hist(lthr, fam_m, fam_f, cnt, -rounds, sig, nfam, false);
    else thr_table(lthr, fam_m, fam_f, cnt, rounds, sig, nfam);
    __syncthreads();
    if (lead) {
        const int hb = -rounds - 100;
        const int sig_e = (int)((unsigned long long)__double_as_longlong(sig) >> 52);
        const unsigned long long sig_m = (unsigned long long)__double_as_longlong(sig) & 0xfffffffffffffull;
        for (int i = threadIdx.x; i < nfam; i += blockDim.x) {
            const unsigned long long tb = (unsigned long long)__double_as_longlong(lthr[i]);
            int k = sig_e - (int)(tb >> 52) - ((tb & 0xfffffffffffffull) > sig_m ? 1 : 0);      // as hpk_score bins a p-value
            k >>= HPK_HSHIFT;
            k = (tb >> 52) == 0ull ? hb - 1 : k;
            k = k < 0 ? 0 : (k > hb - 1 ? hb - 1 : k);
            if (hb > 16) k = fine_bin(sig / lthr[i], hb);
            if (fam_f[i] == 0u) k = hb - 1;                 // no p-value at or below sig: nothing to keep
            small[HPK_OFF_TBIN + i] = (unsigned char)k;
            if (kmin && fam_f[i] != 0u && k < (int)kmin[i]) *reinterpret_cast<unsigned*>(small + HPK_OFF_SPECFAIL) = 1u;
        }
        if ((int64_t)blockIdx.x * blockDim.x >= n) return;
    }
    const int64_t rb = (int64_t)reg * cap;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += stride) {
        const int64_t i = i0 + threadIdx.x;
        bool keep = false;
        // (the record travels as four 8-byte words - records are 40 bytes apart - and a double: a local HpkSurv would live
        // in scratch memory)
        uint2 q0 = make_uint2(0u, 0u), q1 = q0, q2 = q0, q3 = q0;
        double b = 0.0;
        if (i < n) {
            if ((unsigned)(i & (HPK_SCH - 1)) < chunk_used[(rb + i) >> HPK_SCH_LOG2]) {
                const HpkSurv& src = surv[rb + i];
                keep = src.p <= lthr[(int)src.set * (HPK_NB + 1) + (int)src.chunk];
                if (keep) {                               // the other 30 bytes only for the few that stay
                    const uint2* s2 = reinterpret_cast<const uint2*>(&src);
                    q0 = s2[0]; q1 = s2[1]; q2 = s2[2]; q3 = s2[3];       // x, y | O, set chunk flag pad | E | p
                    const int rx = (int)q0.x, ry = (int)q0.y;
                    // the pixel's balanced value (reported with the call, callers.py:254-256), for these few only
                    if (bal) { b = bal[(int64_t)rx * ld + (ry - rx)]; b = (b == b) ? b : 0.0; }
                    else b = balanced_of(__uint_as_float(q1.x), weight[rx], weight[ry]);
                }
            }
        }
        const unsigned long long km = __ballot(keep);
        if (km == 0ull) continue;
        unsigned long long basei = 0ull;
        if (lane == 0) basei = atomicAdd(nout, (unsigned long long)__popcll(km));
        basei = __shfl(basei, 0);
        if (keep) {
            const unsigned long long o = basei + (unsigned long long)__popcll(km & ((1ull << lane) - 1ull));
            HpkSurv* dst = (o < inl) ? out_head + o : out_rest + (o - inl);
            uint2* d2 = reinterpret_cast<uint2*>(dst);
            d2[0] = q0; d2[1] = q1; d2[2] = q2; d2[3] = q3;
            dst->bal = b;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------ launchers
int hpk_stencil_s_lds_bytes() { return LR * LC * 12 + HPK_TLIST * 4 + 128 + HPK_MAX_STEPS * 32 + HPK_KSLOTS * 32 + LC * 8 * 4 + LC * 4 + 64 + 2 * HPK_HACC * 4 + HPK_MAX_STEPS * 4; }

template <bool BALF64, bool SINGLE, bool QUEUE = false>
static void launch_stencil_s_t(const HpkStencilArgs& a, const HpkBandDesc* d_bands, hipStream_t st) {
    auto kern = hpk_stencil_s<BALF64, SINGLE, QUEUE>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  hpk_stencil_s_lds_bytes());
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.grid), dim3(1024), hpk_stencil_s_lds_bytes(), st, a, d_bands);
}

// The buffer-addressing limits of hpk_stencil_s: 32-bit byte offsets inside one tile's rows, weights addressed from element 0;
// a halo of at least 4 (plans with maxww < 4 keep a halo of 4: widths beyond maxww have no step).
bool hpk_stencil_s_applies(const HpkGeo& g, int64_t max_ld, int32_t max_n) {
    return max_ld <= (int64_t)(1 << 21) && max_n < (1 << 27) && g.W >= 4 && g.TR * g.TC <= HPK_TLIST;
}

int hpk_stencil_lean_lds_bytes() { return LR * LC * 4 + 3 * LC * 4 + 128 + HPK_KSLOTS * 32 + 2 * HPK_LST_BYTES + 64; }

template <bool SINGLE>
static void launch_stencil_lean_t(const HpkStencilArgs& a, const HpkBandDesc* d_bands, int cus, hipStream_t st) {
    auto kern = hpk_stencil_lean<SINGLE>;
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 1024, hpk_stencil_lean_lds_bytes()) != hipSuccess || nb <= 0) nb = 1;
        per_cu = nb > 2 ? 2 : nb;
    }
    HpkStencilArgs al = a;
    al.grid = std::max(8, (cus * per_cu) / 8 * 8);          // persistent: the workgroups resident at once
    hipLaunchKernelGGL(kern, dim3(al.grid), dim3(1024), hpk_stencil_lean_lds_bytes(), st, al, d_bands);
}

void hpk_launch_stencil_batch(const HpkStencilArgs& a, const HpkBandDesc* d_bands, bool balf64, int cus, hipStream_t st) {
    const bool single = a.single != 0;
    const bool lean = !balf64 && a.lean_max > 0 && a.redoq && !a.generic;
    if (lean) {
        if (single) launch_stencil_lean_t<true>(a, d_bands, cus, st); else launch_stencil_lean_t<false>(a, d_bands, cus, st);
    }
    if (balf64) { if (single) launch_stencil_s_t<true, true>(a, d_bands, st); else launch_stencil_s_t<true, false>(a, d_bands, st); }
    else        { if (single) launch_stencil_s_t<false, true>(a, d_bands, st); else launch_stencil_s_t<false, false>(a, d_bands, st); }
    if (lean) {
        // the tiles the lean kernel gave up (none, as a rule: the launch finds its queue empty), in full
        HpkStencilArgs aq = a;
        aq.grid = std::min(a.grid, 64);
        if (single) launch_stencil_s_t<false, true, true>(aq, d_bands, st); else launch_stencil_s_t<false, false, true>(aq, d_bands, st);
    }
}

void hpk_launch_freeze_tot(const HpkDevPlan* plan, const HpkBandDesc* d_bands, int nbands, hipStream_t st) {
    hipLaunchKernelGGL(hpk_freeze_tot, dim3(nbands), dim3(128), 0, st, plan, d_bands);
}

void hpk_launch_gap(const float* raw, const double* bal, const double* weight, int32_t n, int32_t num, int64_t ld,
                    int32_t mw, uint8_t* gap, hipStream_t st) {
    hipLaunchKernelGGL(hpk_gap, dim3((n + 3) / 4), dim3(256), 0, st, raw, bal, weight, n, num, ld, mw, gap);
}

// Persistent grid: a single chromosome gets exactly the workgroups that are resident at once (occupancy x CUs), so that
// no second round of workgroups pays the prologue again (measured: 0.105 -> 0.095 ms against twice as many).
template <bool BH, bool ONE>
static int score_grid(int cus, size_t lds) {
    static int per_cu = 0;
    static size_t per_cu_lds = ~(size_t)0;
    if (per_cu == 0 || per_cu_lds != lds) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hpk_score<BH, ONE>, 256, lds) != hipSuccess || nb <= 0) nb = 4;
        per_cu = nb; per_cu_lds = lds;
    }
    return cus * per_cu;
}
// bins per family of the p-value histogram hpk_score keeps (0 = none: HpkScoreArgs::hbins)
int hpk_score_hist_bins(int nsets, bool bhfdr) { return bhfdr ? HPK_FINE_BINS : (nsets <= 6 ? 8 : 4); }
static size_t score_lds(bool bhfdr, int npairs, int hbins) {
    const int nsets = bhfdr ? 1 : 2 * npairs;
    return (size_t)nsets * (HPK_NB_TAB + 1) * (size_t)hbins * 4;
}
int hpk_score_grid(bool bhfdr, int npairs, int hbins, int cus) {
    const size_t lds = score_lds(bhfdr, npairs, hbins);
    if (bhfdr) return score_grid<true, true>(cus, lds);
    if (npairs == 1) return score_grid<false, true>(cus, lds);
    return score_grid<false, false>(cus, lds);
}
void hpk_launch_score(const HpkScoreArgs& a, const HpkBandDesc* d_bands, int nbands, bool bhfdr, hipStream_t st) {
    if (nbands <= 0 || a.gridx <= 0) return;
    const size_t lds = score_lds(bhfdr, a.nsets_half, a.hbins);
    const dim3 grid(a.gridx, nbands);
    if (bhfdr) hipLaunchKernelGGL((hpk_score<true, true>), grid, dim3(256), lds, st, a, d_bands);
    else if (a.nsets_half == 1) hipLaunchKernelGGL((hpk_score<false, true>), grid, dim3(256), lds, st, a, d_bands);
    else hipLaunchKernelGGL((hpk_score<false, false>), grid, dim3(256), lds, st, a, d_bands);
}

int hpk_thr_hist_bins(int nsets) { return nsets * (HPK_NB + 1) <= 1032 ? 16 : 8; }   // LDS of hpk_thr_hist <= 83 KB

void hpk_launch_tighten(const HpkBandDesc* d_bands, int nbands, double sig, int rounds, int nsets, const uint8_t* kmin, hipStream_t st) {
    if (rounds > HPK_TIGHTEN_MAX) rounds = HPK_TIGHTEN_MAX;
    const int nfam = nsets * (HPK_NB + 1);          // families in use: (set, chunk)
    const dim3 grid(8, HPK_NREG, nbands);
    if (rounds <= -100) {       // the histogram came with the scoring kernel: only the compaction is left
        // (hiccups' survivor lists are short - a few hundred records per region once the bound of the chromosomes before is in,
        // a thousand without - and 32 768 workgroups per 64 chromosomes, nearly all of them leaving at once, took 0.18 ms; bhfdr's
        // one family keeps tens of thousands per region)
        const dim3 gridc(nsets == 1 ? 8 : 2, HPK_NREG, nbands);
        hipLaunchKernelGGL(hpk_thr_compact, gridc, dim3(256), 0, st, d_bands, rounds, sig, nfam, kmin);
        return;
    }
    const uint8_t* none = nullptr;
    if (rounds < 0) {           // one histogram pass instead of the counting rounds (cnt = [nfam][nbins], zeroed)
        const int nbins = hpk_thr_hist_bins(nsets);
        const size_t lds = (size_t)nfam * 8 + (size_t)nfam * nbins * 4;
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hpk_thr_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_done = true;
        }
        hipLaunchKernelGGL(hpk_thr_hist, grid, dim3(256), lds, st, d_bands, nbins, sig, nfam);
        hipLaunchKernelGGL(hpk_thr_compact, grid, dim3(256), 0, st, d_bands, -nbins, sig, nfam, none);
        return;
    }
    for (int r = 0; r < rounds; ++r)
        hipLaunchKernelGGL(hpk_thr_count, grid, dim3(256), 0, st, d_bands, r, sig, nfam);
    hipLaunchKernelGGL(hpk_thr_compact, grid, dim3(256), 0, st, d_bands, rounds, sig, nfam, none);
}

void hpk_launch_prep(const HpkBandDesc* d_bands, int nbands, int max_n, int max_num, int mw, hipStream_t st) {
    const int nparts = (max_n + HPK_IR_ROWS - 1) / HPK_IR_ROWS;        // hpk_api.cpp sizes psum / pnan with the same constant
    const int nirb = (max_num + 31) / 32;
    hipLaunchKernelGGL(hpk_ir_partial, dim3(nparts, (max_num - mw + 256 * HPK_IR_KPT - 1) / (256 * HPK_IR_KPT), nbands), dim3(256), 0, st, d_bands, mw);
    hipLaunchKernelGGL(hpk_ir_final, dim3(nirb + (max_n + 255) / 256, nbands), dim3(256), 0, st, d_bands, mw, nirb);
}

void hpk_launch_etab(const HpkDevPlan* plan, const HpkBandDesc* d_bands, int nbands, int nsteps, int D, int W, size_t max_zero,
                     hipStream_t st) {
    const int gx = (D + HPK_ET_THREADS) / HPK_ET_THREADS;
    const int ntab = 2 * W * nsteps + nsteps;
    // rows of workgroups beyond the tables zero-fill the bands' counter blocks: gx * 256 threads x 16 B per row
    const int nzero = (int)((max_zero / 16 + (size_t)gx * HPK_ET_THREADS - 1) / ((size_t)gx * HPK_ET_THREADS));
    hipLaunchKernelGGL(hpk_etab_edge, dim3(gx, ntab + nzero, nbands), dim3(HPK_ET_THREADS), 0, st, plan, d_bands, ntab);
}

// Result heads -> pinned host memory by a kernel (the copy engine costs ~15 us of start-up latency per chromosome).  A
// head is 200 KB of which a chromosome fills a third - family counters of the sets in use, one flag per row, the
// survivors that made the cut - and the PCIe write is what the launch takes: only the filled stretches travel (four
// [begin, end) in 16-byte units; the last one ends after min(nout, HPK_HEAD_INLINE) records).  blockIdx.y = band.
__global__ void __launch_bounds__(256) hpk_publish(const HpkBandDesc* __restrict__ bands, int nsets, int full) {
    const HpkBandDesc* __restrict__ bd = bands + blockIdx.y;
    const unsigned char* const small = gptr(bd->small);
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(small);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(gptr(bd->head_host));
    unsigned b[4], e[4];
    const unsigned head_end = (bd->off_inl + (unsigned)(HPK_HEAD_INLINE * sizeof(HpkSurv))) / 16u;
    if (full) { b[0] = 0u; e[0] = head_end; b[1] = e[1] = b[2] = e[2] = b[3] = e[3] = 0u; }
    else {
        const unsigned nfam_b = 4u * (unsigned)nsets * (HPK_NB + 1);
        const unsigned long long no = *reinterpret_cast<const unsigned long long*>(small + HPK_OFF_NOUT);
        b[0] = 0u; e[0] = (HPK_OFF_FAM_M + nfam_b + 15u) / 16u;
        b[1] = HPK_OFF_FAM_F / 16u; e[1] = (HPK_OFF_FAM_F + nfam_b + 15u) / 16u;
        b[2] = bd->off_rowlive / 16u; e[2] = (bd->off_rowlive + (unsigned)bd->n + 15u) / 16u;
        b[3] = bd->off_inl / 16u;
        e[3] = b[3] + (unsigned)(((no < HPK_HEAD_INLINE ? no : (unsigned long long)HPK_HEAD_INLINE) * sizeof(HpkSurv) + 15ull) / 16ull);
    }
    for (unsigned i0 = blockIdx.x * 256u + threadIdx.x; ; i0 += gridDim.x * 256u) {
        unsigned i = i0;                    // thread index -> unit: the stretches back to back
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned len = e[k] - b[k];
            if (!hit && i < len) { dst[b[k] + i] = src[b[k] + i]; hit = true; }
            i -= len;
        }
        if (!hit) break;
    }
}
void hpk_launch_publish(const HpkBandDesc* d_bands, int nbands, int nsets, bool full, size_t max_head_bytes, hipStream_t st) {
    // enough workgroups per band for the whole head in one sweep when little is filled; the loop covers the rest
    const unsigned units = (unsigned)((max_head_bytes + 15) / 16);
    unsigned gx = (units / (full ? 1u : 3u) + 255u) / 256u;
    gx = gx < 1u ? 1u : gx;
    hipLaunchKernelGGL(hpk_publish, dim3(gx, nbands), dim3(256), 0, st, d_bands, nsets, full ? 1 : 0);
}

void hpk_launch_kcrit(const double* ptab, const int32_t* off, double sig, int32_t* kcrit, hipStream_t st) {
    static_assert(HPK_NB_TAB <= 64, "one thread per chunk of the Poisson table");
    hipLaunchKernelGGL(hpk_kcrit, dim3(1), dim3(64), 0, st, ptab, off, sig, kcrit);
}

void hpk_launch_kcrit_lam(const double* sfe, double sig, int32_t* kcrit, hipStream_t st) {
    hipLaunchKernelGGL(hpk_kcrit_lam, dim3(HPK_KCL_N / 64), dim3(64), 0, st, sfe, sig, kcrit);
}

void hpk_launch_ptab(const double* bounds, const int32_t* off, const double* sfe, double* ptab, int32_t total,
                     hipStream_t st) {
    hipLaunchKernelGGL(hpk_ptab, dim3((total + 255) / 256), dim3(256), 0, st, bounds, off, sfe, ptab, total);
}

void hpk_launch_band_class(HpkBandDesc* d_bands, int nbands, const HpkClassArgs& a, hipStream_t st) {
    if (nbands > 0) hipLaunchKernelGGL(hpk_band_class, dim3(nbands), dim3(1024), 0, st, d_bands, a);
}

void hpk_launch_coo_scatter(const int64_t* bin1, const int64_t* bin2, const void* count, int count_f64, int64_t nnz, int n, int num,
                            int64_t ld, float* raw, unsigned long long* info, hipStream_t st) {
    if (nnz <= 0) return;
    const int64_t blocks = (nnz + 255) / 256;
    hipLaunchKernelGGL(hpk_coo_scatter, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, bin1, bin2, count, count_f64, nnz,
                       n, num, ld, raw, info);
}

