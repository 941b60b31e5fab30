// libhpk.so entry points (include/hpk.h): context, staging, the per-chromosome pipeline and the
// Benjamini-Hochberg step on the compacted survivors.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

#include "../../include/hpk.h"
#include "hpk_kernels.h"
#include "hpk_plan.h"
#include "hpk_cpu.h"

#ifdef HPK_TEST_KERNELS
#define HPK_NEED_TEST_KERNELS(ctx) do {} while (0)
#else   // product-only build (make TESTK=): the tests' check kernels and the dense debug outputs are not in the library
#define HPK_NEED_TEST_KERNELS(ctx) return fail(ctx, HPK_ERR_INVALID, "libhpk.so was built without its test kernels (HPK_TEST_KERNELS)")
void hpk_launch_dense(const HpkDenseArgs&, hipStream_t) {}
void hpk_launch_probe(const HpkDenseArgs&, const int32_t*, const int32_t*, int64_t, double*, hipStream_t) {}
void hpk_launch_brute(const HpkBruteArgs&, hipStream_t) {}
void hpk_launch_poisson_sf(const double*, const double*, const double*, double*, int64_t, hipStream_t) {}
#endif

namespace {

std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        // (an eighth of room to grow: a batch's workspaces follow its bands' tile geometries, which move by a few percent with the bounds
        //  the bands inherit - bounded at 256 MB, chr1 @5 kb re-allocated its 20 GB buffers every few calls at ~25 ms per GB, 0.31 -> 53 ms
        //  per chromosome on a box whose driver had pages to clear: profiles/r06_fresh_context.txt)
        size_t want = bytes + bytes / 8 + 256;
        static const bool prof = std::getenv("HPK_ALLOC_PROF") != nullptr;      // (what a fresh context's first call spends in hipMalloc)
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, want);
        if (prof) std::fprintf(stderr, "[hpk alloc] %.3f GB in %.1f ms\n", want / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host memory handed out in pieces and taken back all at once (a lane's survivor lists beyond the inline heads): copies
// into pageable memory are staged by the runtime, through kernels that queue behind the next batch's persistent workgroups - a
// collection then waited one or two batch times for a few megabytes (maps with structure: tens of thousands of survivors per
// chromosome; the bench line came out at 0.37 or 0.74 ms per chromosome by how the copies happened to fall).
struct PinnedArena {
    std::vector<std::pair<char*, size_t>> chunks;
    size_t cur = 0, used = 0;
    void reset() { cur = 0; used = 0; }
    void* take(size_t bytes) {
        bytes = (bytes + 63) & ~(size_t)63;
        while (cur < chunks.size() && used + bytes > chunks[cur].second) { ++cur; used = 0; }
        if (cur == chunks.size()) {
            const size_t cap = std::max(bytes, (size_t)16 << 20);
            void* p = nullptr;
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
            chunks.emplace_back(static_cast<char*>(p), cap);
            used = 0;
        }
        void* r = chunks[cur].first + used;
        used += bytes;
        return r;
    }
    void release() { for (auto& ch : chunks) (void)hipHostFree(ch.first); chunks.clear(); reset(); }
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// One lane = everything a batch of chromosomes in flight owns: upload stream, events, device workspaces (pooled over
// the batch's bands), the pinned landing area of the result heads and the cached device copy of the widening plan.  Two
// lanes let the host half of batch i (Benjamini-Hochberg, result assembly, the caller's Python) and the upload of batch
// i + 1 overlap the kernels.  The stencil, the scoring and the cut of all lanes run in submission order on the context's
// one compute stream: the stencil fills the whole chip (one 160 KiB-LDS workgroup per CU), so running two batches' big
// kernels side by side would only time-slice them.  What a batch needs *before* its stencil - uploads, IR / biases, the
// expected tables and the zero-fill of its counters - runs on the lane's side stream, beside the scoring / cut kernels
// of the batch before.  (The cut and the copy of the result head were tried on the side stream too: they then wait
// behind the next stencil, the host collects a result later and submits the batch after next later.)
#define HPK_LANES 2
struct Lane {
    hipStream_t up = nullptr;           // uploads of host inputs run beside the kernels of the batch before
    hipEvent_t ev[8];                   // phase marks on the compute stream
    hipEvent_t ev_up = nullptr, ev_done = nullptr;
    int nev = 0;
    bool busy = false;
    // workspaces (grow only), pooled over the bands of a batch
    DevBuf raw, bal, weight, IR, b1, b2, plan, etab, eedge, recE, recS, recW, dE, dW, dS, small, surv, surv2, survx, survx2, cux, psum, pnan, units, desc, kmin, classtab, redoq;
    void* h_head = nullptr;             // pinned: per band counters | row flags | first survivors
    size_t h_head_cap = 0;
    void* h_desc = nullptr;             // pinned staging of the band descriptors
    size_t h_desc_cap = 0;
    PinnedArena h_rest;                 // pinned: the survivors beyond the inline heads of the batch being collected
    // the device copy of the widening plan is reused while the parameters do not change
    hpk_params plan_key;
    bool plan_valid = false;
    HpkDevPlan plan_host;
    void release() {
        DevBuf* all[] = {&raw, &bal, &weight, &IR, &b1, &b2, &plan, &etab, &eedge, &recE, &recS, &recW, &dE, &dW, &dS, &small,
                         &surv, &surv2, &survx, &survx2, &cux, &psum, &pnan, &units, &desc, &kmin, &classtab, &redoq};
        for (DevBuf* b : all) b->release();
        if (h_head) { (void)hipHostFree(h_head); h_head = nullptr; h_head_cap = 0; }
        if (h_desc) { (void)hipHostFree(h_desc); h_desc = nullptr; h_desc_cap = 0; }
        h_rest.release();
        for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
        nev = 0;
        if (ev_up) { (void)hipEventDestroy(ev_up); ev_up = nullptr; }
        if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
        if (up) { (void)hipStreamDestroy(up); up = nullptr; }
    }
};

// Tuning and test switches: read from the environment once (hpk_create), changed through hpk_set_option; the submit /
// collect paths only ever look at this struct.
struct Options {
    int rounds = -2;            // -2: the scoring kernel keeps the p-value histogram of the cut; -1: hpk_thr_hist; 0..4 counting rounds
    int64_t surv_cap = 0;       // survivor slots per region (0: sized from the band)
    int spec = 1;               // record bound from the chromosomes collected before
    int spec_margin = 0;
    int spec_force = -1;
    int tr_cap = 127;           // most rows of an hpk_stencil_s output tile (A/B: 64 = the first-generation kernel's limit)
    int spec_class = 1;         // record bound per chromosome by depth class (hpk_band_class) under the batch's bound
    int spec_halo = 1;          // tiles laid out for the record bound's halo instead of maxww's (hpk_stencil_s launches); 2: ... and a chromosome whose
                                // halo was not the one of its OWN frozen width is computed once more under that one (its values then depend on it alone)
    int risk_log2 = 12;
    int tile_order = 1;
    int gap_kernel = 0;
    int score_div = 6;          // tiles per scoring workgroup of a batch (4 ... 128 measured: profiles/r05_score_div.txt)
    int dbg_stop = 0;
    int surv_div = 6;           // survivor capacity of a chromosome: band pixels x sets / surv_div records (p <= sig; more: scored once more with room)
    int grid_cap = 0;           // tests: at most this many stencil workgroups (0 = one per CU) - long walks, many bands per workgroup
    int host_prof = 0;
    int spec_surv = 1;          // survivor records only up to the cut's histogram bin of the chromosomes before (minus spec_surv_margin bins)
    int spec_surv_margin = 2;
    int spec_surv_force = -1;   // tests: this bin for every family (too narrow a bound: scored once more)
    int host_threads = 8;       // threads of a batch's host half (hpk_collect_batch), at most one per four chromosomes (hpk_create: up to 16 by the host's cores)
    int kcrit = 1;              // hpk_score forms a p-value only where the count reaches the critical count of its chunk (hiccups) / of its lambda's cell (bhfdr)
    int side_serial = 0;        // measurement (profiles/r06_side_stream.txt): the tables / class kernels of a batch on the compute stream, behind the batch before, instead of beside its scoring on the lane's side stream
    int lean = 1;               // tiles of the column chunks hpk_band_class expects no resolving candidate in are built without their f64 plane
    int lean_max = 24;          // ... candidates of such a tile that do count and get their sums cell by cell; more: the tile is computed once more
    int lean_share_pct = 50;    // ... and a band has lean tiles only if at least this share of its column chunks is lean (below: hpk_stencil_s does them as fast)
    int lean_frac_pct = 35;     // ... a chunk is lean when the mean Reads of its nearest pixels is at most this share of min_local_reads
};

struct hpk_ctx {
    int device = -1;                    // HIP device ordinal; -1: back-end #0, the path on host threads (hpk_cpu.cpp) - only ever by request
    int cpu_threads = 1;                // ... its threads (HPK_CPU_THREADS / option "cpu_threads"; default: every core the process may use)
    HpkCpuTables cpu_tabs;              // ... and its Poisson tables
    hipStream_t stream = nullptr;       // compute stream: every kernel and the result downloads
    std::string err;
    char name[128] = {0};
    int cus = 0;
    size_t hbm = 0;
    Options opt;
    // constant tables
    std::vector<double> h_bounds;
    std::vector<int32_t> h_off;
    std::vector<double> h_sfe;
    DevBuf d_bounds, d_off, d_sfe, d_ptab, d_kcrit, d_kclam;
    double kcrit_sig = -1.0;            // the sig d_kcrit was built for (hpk_kcrit; rebuilt with the tables)
    double kclam_sig = -1.0;            // ... and d_kclam (bhfdr: hpk_kcrit_lam)
    bool tables_dirty = true;           // the Poisson table is (re)built before the first launch that needs it
    Lane lane[HPK_LANES];
    DevBuf tmpA, tmpB, tmpC, tmpD;
    // hpk_devband_create: a stream of its own, staging buffers for the pixel table, and the bands handed back (kept for reuse)
    hipStream_t aux = nullptr;
    DevBuf cooA, cooB, cooC;
    std::vector<std::pair<size_t, void*>> pool;
    std::vector<struct hpk_devband*> live_bands;    // handed out by hpk_devband_create and not freed yet: hpk_destroy frees them
    // The widths the widening froze at (freeze_replay) in the chromosomes collected last with these parameters: the next
    // stencil writes records up to the widest of them only (HpkBandDesc::wguess); a chromosome that freezes later is
    // redone in full.
    hpk_params hint_key;
    int hint_w[4] = {-1, -1, -1, -1};
    int hint_n = 0;
    // ... and per depth class (hpk_band_class) the width the last chromosome of the class froze at (-1: none seen): a band's own
    // record bound under the batch's
    signed char class_w[HPK_NCLASS];
    signed char class_w1[HPK_NCLASS];   // ... and the one before it: the band's bound is the wider of the two
    long long spec_reruns = 0;
    // ... and the histogram bins their Benjamini-Hochberg cuts fell into, per family (HPK_OFF_TBIN): the smallest bin of the
    // last collections, minus a margin, bounds the survivor records the next scoring launches write (HpkScoreArgs::kmin)
    uint8_t hint_bin[4][HPK_NFAM];
    int hint_bn = 0;
    long long surv_rescored = 0;
};

namespace {

int fail(hpk_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(ctx, (e__ == hipErrorOutOfMemory) ? HPK_ERR_NOMEM : HPK_ERR_HIP, "%s -> %s", #call, \
                        hipGetErrorName(e__));                                                         \
    } while (0)

void fill_bounds(std::vector<double>& b) {
    b.resize(HPK_NB);
    for (int i = 1; i <= HPK_NB; ++i) b[i - 1] = std::pow(2.0, (double)(i - 1) / 3.0);   // callers.py:36-37
}

// Table length per chunk: beyond lam + t with t^2 / (2 (lam + t/3)) >= 40 the Poisson tail is < 2^-57, i.e.
// 1 - cdf rounds to exactly 0.
void fill_offsets(const std::vector<double>& b, std::vector<int32_t>& off) {
    off.assign(HPK_NB_TAB + 2, 0);
    int32_t total = 0;
    for (int ch = 1; ch <= HPK_NB_TAB; ++ch) {
        const double lam = b[ch - 1];
        const double t = (26.7 + std::sqrt(26.7 * 26.7 + 320.0 * lam)) / 2.0;
        off[ch] = total;
        total += (int32_t)std::ceil(lam + t) + 4;
    }
    off[HPK_NB_TAB + 1] = total;
}

void fill_sfe(std::vector<double>& sfe) {
    sfe.resize(32);
    sfe[0] = 0.0;
    const long double half_log_2pi = 0.5L * logl(2.0L * acosl(-1.0L));
    for (int n = 1; n < 32; ++n) {
        const long double x = (long double)n;
        sfe[n] = (double)(lgammal(x + 1.0L) - (x + 0.5L) * logl(x) + x - half_log_2pi);
    }
}

// chunk bounds, table offsets, stirlerr values and the Poisson table on the device; built before the first launch that
// needs them and again after hpk_set_chunk_bounds (the Python layer replaces the bounds right after hpk_create: one build)
int upload_tables(hpk_ctx* c) {
    if (!c->tables_dirty) return HPK_OK;
    fill_offsets(c->h_bounds, c->h_off);
    const int32_t total = c->h_off[HPK_NB_TAB + 1];
    HIPCHK(c, c->d_bounds.reserve(sizeof(double) * HPK_NB));
    HIPCHK(c, c->d_off.reserve(sizeof(int32_t) * (HPK_NB_TAB + 2)));
    HIPCHK(c, c->d_sfe.reserve(sizeof(double) * 32));
    HIPCHK(c, c->d_ptab.reserve(sizeof(double) * (size_t)total));
    HIPCHK(c, hipMemcpyAsync(c->d_bounds.p, c->h_bounds.data(), sizeof(double) * HPK_NB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_off.p, c->h_off.data(), sizeof(int32_t) * (HPK_NB_TAB + 2), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_sfe.p, c->h_sfe.data(), sizeof(double) * 32, hipMemcpyHostToDevice, c->stream));
    hpk_launch_ptab(c->d_bounds.as<double>(), c->d_off.as<int32_t>(), c->d_sfe.as<double>(), c->d_ptab.as<double>(), total, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tables_dirty = false;
    c->kcrit_sig = -1.0;
    c->kclam_sig = -1.0;
    return HPK_OK;
}

int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}

struct ResultBox {
    hpk_result pub;          // must stay first
    std::vector<int32_t> x, y;
    std::vector<double> O, bal, E, p, q;
    std::vector<uint8_t> oz, gap, denseW;
    std::vector<double> denseE, denseS;
    std::vector<uint32_t> fam;      // [2][nsets][HPK_NB + 1]: tests per chunk, of those p <= sig
};

struct Surv { int32_t x, y; uint8_t set, chunk, flag, keep; float O; double E, p, bal, q; };
static_assert(sizeof(HpkSurv) == 40, "survivor record layout");

// Benjamini-Hochberg on the p <= sig subset of one family of m tests (statsmodels fdr_bh): the subset holds
// the m' smallest p-values, so their ranks and step-up q-values are those of the full family.
// `fam` arrives sorted by p.
void bh_family(Surv* const* fam, size_t k, unsigned long long m, double sig, bool use_reject_mask) {
    double running = INFINITY;
    long rejectmax = -1;
    for (size_t j = k; j-- > 0;) {
        const double ecdf = (double)(j + 1) / (double)m;
        const double qraw = fam[j]->p / ecdf;
        running = std::min(running, qraw);
        fam[j]->q = running > 1.0 ? 1.0 : running;
        if (use_reject_mask && rejectmax < 0 && fam[j]->p <= ecdf * sig) rejectmax = (long)j;
    }
    // hiccups keeps q <= sig (callers.py:279); bhfdr keeps statsmodels' step-up mask (callers.py:546)
    for (size_t j = 0; j < k; ++j)
        fam[j]->keep = use_reject_mask ? ((long)j <= rejectmax) : (fam[j]->q <= sig);
}

}  // namespace

extern "C" {

int hpk_abi_version(void) { return HPK_ABI_VERSION; }

const char* hpk_last_error(const hpk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hpk_create(int device, hpk_ctx** out) {
    if (!out) return fail(nullptr, HPK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (device == -1) {
        // Back-end #0 ("device -1 = CPU"): no HIP call is made.  Never a fallback - a caller gets it by asking for it.
        hpk_ctx* c = new hpk_ctx();
        { std::memset(c->class_w, -1, sizeof(c->class_w)); std::memset(c->class_w1, -1, sizeof(c->class_w1)); }
        c->device = -1;
        c->cus = 0;
        c->hbm = 0;
        int nt = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
        { cpu_set_t set; if (sched_getaffinity(0, sizeof(set), &set) == 0) nt = CPU_COUNT(&set); }
#endif
        c->cpu_threads = std::max(1, std::min(1024, env_int("HPK_CPU_THREADS", std::max(1, nt))));
        std::snprintf(c->name, sizeof(c->name), "host threads (back-end #0, %d threads)", c->cpu_threads);
        fill_bounds(c->h_bounds);
        fill_sfe(c->h_sfe);
        c->tables_dirty = true;
        *out = c;
        return HPK_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, HPK_ERR_NO_DEVICE, "no HIP device (hipGetDeviceCount -> %s, count %d); libhpk has no CPU path",
                    hipGetErrorName(e), count);
    if (device < 0 || device >= count) return fail(nullptr, HPK_ERR_INVALID, "device %d out of range (count %d)", device, count);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return fail(nullptr, HPK_ERR_HIP, "hipGetDeviceProperties -> %s", hipGetErrorName(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HPK_ERR_NO_DEVICE, "device %d is %s; libhpk is built for gfx950 only", device, prop.gcnArchName);
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(nullptr, HPK_ERR_HIP, "hipSetDevice -> %s", hipGetErrorName(e));
    hpk_ctx* c = new hpk_ctx();
    { std::memset(c->class_w, -1, sizeof(c->class_w)); std::memset(c->class_w1, -1, sizeof(c->class_w1)); }
    c->device = device;
    std::snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    c->cus = prop.multiProcessorCount;
    c->hbm = prop.totalGlobalMem;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int l = 0; l < HPK_LANES && e == hipSuccess; ++l) {
        Lane& L = c->lane[l];
        e = hipStreamCreateWithFlags(&L.up, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&L.ev_up, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming);
        for (int i = 0; i < 8 && e == hipSuccess; ++i) { e = hipEventCreate(&L.ev[i]); if (e == hipSuccess) L.nev++; }
    }
    if (e != hipSuccess) {
        hpk_destroy(c);
        return fail(nullptr, HPK_ERR_HIP, "stream / event creation -> %s", hipGetErrorName(e));
    }
    fill_bounds(c->h_bounds);
    fill_sfe(c->h_sfe);
    // the environment is read here and nowhere else (A/B scripts); tests use hpk_set_option
    Options& o = c->opt;
    o.rounds = std::min(4, env_int("HPK_ROUNDS", o.rounds));
    o.spec = env_int("HPK_SPEC", o.spec);
    o.spec_margin = env_int("HPK_SPEC_MARGIN", o.spec_margin);
    o.spec_halo = env_int("HPK_SPEC_HALO", o.spec_halo);
    o.spec_class = env_int("HPK_SPEC_CLASS", o.spec_class) ? 1 : 0;
    o.spec_force = env_int("HPK_SPEC_FORCE", o.spec_force);     // (measurements: a record bound of one's choosing)
    // (up to 16 where the host has the cores: maps with structure leave tens of thousands of survivors per chromosome, a few
    // milliseconds of sorting each, and the host half of a batch has to stay below its 20 ms of kernels)
    o.host_threads = std::max(2, std::min(16, (int)std::thread::hardware_concurrency() / 2));
    o.host_threads = std::max(1, std::min(64, env_int("HPK_HOST_THREADS", o.host_threads)));
    o.spec_surv = env_int("HPK_SPEC_SURV", o.spec_surv) ? 1 : 0;
    o.spec_surv_margin = std::max(0, env_int("HPK_SPEC_SURV_MARGIN", o.spec_surv_margin));
    o.tr_cap = std::max(16, std::min(127, env_int("HPK_TR_CAP", o.tr_cap)));
    o.risk_log2 = env_int("HPK_RISK_LOG2", o.risk_log2);
    o.tile_order = env_int("HPK_TILE_ORDER", o.tile_order);
    o.gap_kernel = env_int("HPK_GAP_KERNEL", o.gap_kernel);
    o.score_div = std::max(1, env_int("HPK_SCORE_DIV", o.score_div));
    o.dbg_stop = env_int("HPK_DBG_STOP", o.dbg_stop);
    o.surv_div = std::max(1, std::min(1024, env_int("HPK_SURV_DIV", o.surv_div)));
    o.host_prof = env_int("HPK_HOST_PROF", o.host_prof);
    o.kcrit = env_int("HPK_KCRIT", o.kcrit) ? 1 : 0;
    o.side_serial = env_int("HPK_SIDE_SERIAL", o.side_serial) ? 1 : 0;
    o.lean = env_int("HPK_LEAN", o.lean) ? 1 : 0;
    o.lean_max = std::max(0, std::min(4096, env_int("HPK_LEAN_MAX", o.lean_max)));
    o.lean_frac_pct = std::max(0, std::min(400, env_int("HPK_LEAN_FRAC", o.lean_frac_pct)));
    o.lean_share_pct = std::max(0, std::min(100, env_int("HPK_LEAN_SHARE", o.lean_share_pct)));
    *out = c;
    return HPK_OK;
}

int hpk_set_option(hpk_ctx* c, const char* name, int64_t v) {
    if (!c || !name) return HPK_ERR_INVALID;
    Options& o = c->opt;
    const std::string k(name);
    if (k == "rounds" && v >= -2 && v <= 4) o.rounds = (int)v;
    else if (k == "surv_cap" && v >= 0) o.surv_cap = v;
    else if (k == "spec" && (v == 0 || v == 1)) o.spec = (int)v;
    else if (k == "spec_margin" && v >= 0 && v <= HPK_MAX_W) o.spec_margin = (int)v;
    else if (k == "spec_halo" && v >= 0 && v <= 2) o.spec_halo = (int)v;
    else if (k == "host_threads" && v >= 1 && v <= 64) o.host_threads = (int)v;
    else if (k == "cpu_threads" && v >= 1 && v <= 1024) c->cpu_threads = (int)v;      // back-end #0 (device -1)
    else if (k == "spec_surv" && (v == 0 || v == 1)) o.spec_surv = (int)v;
    else if (k == "spec_surv_margin" && v >= 0 && v <= 16) o.spec_surv_margin = (int)v;
    else if (k == "spec_surv_force" && v >= -1 && v <= 255) o.spec_surv_force = (int)v;      // (clamped to the last bin)
    else if (k == "spec_force" && v >= -1 && v <= 255) o.spec_force = (int)v;     // (clamped to maxww where it is used)
    else if (k == "risk_log2" && v >= 0 && v <= 60) o.risk_log2 = (int)v;
    else if (k == "tile_order" && (v == 0 || v == 1)) o.tile_order = (int)v;
    else if (k == "kcrit" && (v == 0 || v == 1)) o.kcrit = (int)v;
    else if (k == "gap_kernel" && (v == 0 || v == 1)) o.gap_kernel = (int)v;
    else if (k == "score_div" && v >= 1 && v <= 4096) o.score_div = (int)v;
    else if (k == "dbg_stop" && v >= 0 && v <= 16) o.dbg_stop = (int)v;
    else if (k == "grid_cap" && v >= 0 && v <= 4096) o.grid_cap = (int)v;
    else if (k == "surv_div" && v >= 1 && v <= 1024) o.surv_div = (int)v;
    else if (k == "spec_class" && (v == 0 || v == 1)) o.spec_class = (int)v;
    else if (k == "lean" && (v == 0 || v == 1)) o.lean = (int)v;
    else if (k == "lean_max" && v >= 0 && v <= 4096) o.lean_max = (int)v;
    else if (k == "lean_share_pct" && v >= 0 && v <= 100) o.lean_share_pct = (int)v;
    else if (k == "lean_frac_pct" && v >= 0 && v <= 100000) o.lean_frac_pct = (int)v;       // (tests: a large share makes every chunk lean)
    else if (k == "class_force" && v >= -1 && v <= 127) { std::memset(c->class_w, (int)v, sizeof(c->class_w)); std::memset(c->class_w1, (int)v, sizeof(c->class_w1)); }     // tests: every depth class claims this width
    else if (k == "reset_hints") { c->hint_n = 0; c->hint_bn = 0; { std::memset(c->class_w, -1, sizeof(c->class_w)); std::memset(c->class_w1, -1, sizeof(c->class_w1)); } }     // forget the bounds learnt from the chromosomes collected so far
    else return fail(c, HPK_ERR_INVALID, "unknown option or value out of range: %s = %lld", name, (long long)v);
    return HPK_OK;
}

void hpk_destroy(hpk_ctx* c) {
    if (!c) return;
    if (c->device < 0) { delete c; return; }
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    DevBuf* all[] = {&c->d_bounds, &c->d_off, &c->d_sfe, &c->d_ptab, &c->d_kcrit, &c->d_kclam, &c->tmpA, &c->tmpB, &c->tmpC, &c->tmpD, &c->cooA, &c->cooB, &c->cooC};
    for (DevBuf* b : all) b->release();
    while (!c->live_bands.empty()) hpk_devband_free(c, c->live_bands.back());      // (bands their owner did not free: into the pool, freed below)
    for (auto& e : c->pool) (void)hipFree(e.second);
    c->pool.clear();
    if (c->aux) { (void)hipStreamDestroy(c->aux); c->aux = nullptr; }
    for (int l = 0; l < HPK_LANES; ++l) c->lane[l].release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int hpk_set_chunk_bounds(hpk_ctx* c, const double* bounds, int32_t count) {
    if (!c || !bounds || count != HPK_NB) return fail(c, HPK_ERR_INVALID, "need %d bounds", HPK_NB);
    if (c->device >= 0) (void)hipSetDevice(c->device);
    for (int i = 0; i < HPK_NB; ++i) {
        if (!(bounds[i] > 0.0) || (i && !(bounds[i] > bounds[i - 1]))) return fail(c, HPK_ERR_INVALID, "bounds must increase");
        c->h_bounds[i] = bounds[i];
    }
    c->tables_dirty = true;
    return HPK_OK;
}

int hpk_chunk_bounds(double* bounds, int32_t count) {
    if (!bounds || count < 1 || count > HPK_NB) return HPK_ERR_INVALID;
    std::vector<double> b;
    fill_bounds(b);
    std::memcpy(bounds, b.data(), sizeof(double) * count);
    return HPK_OK;
}

int hpk_plan_rings(const hpk_params* params, int32_t* step_pi, int32_t* step_wi, int32_t* mult_K, int32_t* mult_reads) {
    if (!params) return HPK_ERR_INVALID;
    HpkDevPlan plan;
    char msg[256];
    int rc = hpk_build_plan(params, &plan, msg);
    if (rc != HPK_OK) { g_create_error = msg; return rc; }
    for (int s = 0; s < plan.nsteps; ++s) {
        if (step_pi) step_pi[s] = plan.steps[s].pi;
        if (step_wi) step_wi[s] = plan.steps[s].wi;
        for (int r = 0; r <= HPK_MAX_W; ++r) {
            if (mult_K) mult_K[s * (HPK_MAX_W + 1) + r] = plan.steps[s].m[r];
            if (mult_reads) mult_reads[s * (HPK_MAX_W + 1) + r] = plan.steps[s].mr[r];
        }
    }
    return plan.nsteps;
}

int hpk_device_info(hpk_ctx* c, char* name, int32_t name_len, int32_t* cus, int64_t* hbm_bytes) {
    if (!c) return HPK_ERR_INVALID;
    if (name && name_len > 0) { std::strncpy(name, c->name, name_len - 1); name[name_len - 1] = 0; }
    if (cus) *cus = c->cus;
    if (hbm_bytes) *hbm_bytes = (int64_t)c->hbm;
    return HPK_OK;
}

int hpk_poisson_sf(hpk_ctx* c, const double* k, const double* lam, double* out, int64_t count) {
    if (c && c->device >= 0) { HPK_NEED_TEST_KERNELS(c); }
    if (!c || !k || !lam || !out || count < 0) return fail(c, HPK_ERR_INVALID, "bad arguments");
    if (count == 0) return HPK_OK;
    if (c->device < 0) {        // back-end #0: the same series on the host
        for (int64_t i = 0; i < count; ++i) out[i] = hpk_cpu_poisson_sf(k[i], lam[i], c->h_sfe.data(), 1.0);
        return HPK_OK;
    }
    (void)hipSetDevice(c->device);
    { const int rc = upload_tables(c); if (rc != HPK_OK) return rc; }
    const size_t bytes = sizeof(double) * (size_t)count;
    HIPCHK(c, c->tmpA.reserve(bytes));
    HIPCHK(c, c->tmpB.reserve(bytes));
    HIPCHK(c, c->tmpC.reserve(bytes));
    HIPCHK(c, hipMemcpyAsync(c->tmpA.p, k, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->tmpB.p, lam, bytes, hipMemcpyHostToDevice, c->stream));
    hpk_launch_poisson_sf(c->tmpA.as<double>(), c->tmpB.as<double>(), c->d_sfe.as<double>(), c->tmpC.as<double>(), count, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->tmpC.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HPK_OK;
}

}  // extern "C"


// ---------------------------------------------------------------------------- the per-batch pipeline
namespace {

struct Staged {
    const float* raw = nullptr;
    const double* bal = nullptr;
    const double* weight = nullptr;
    double* IR = nullptr;
    double* b1 = nullptr;
    double* b2 = nullptr;
};

int check_band(hpk_ctx* c, const hpk_band* band) {
    if (!band || band->n <= 0 || band->num <= 0 || band->ld < band->num) return fail(c, HPK_ERR_INVALID, "bad band shape");
    if (!band->raw) return fail(c, HPK_ERR_INVALID, "raw pointer required");
    if (!band->balanced && !band->weight) return fail(c, HPK_ERR_INVALID, "either balanced or weight must be given");
    const bool derive = !band->IR;
    if (derive && !band->weight) return fail(c, HPK_ERR_INVALID, "IR / biases can only be derived on the device from `weight`");
    if ((band->bias1 == nullptr) != (band->bias2 == nullptr)) return fail(c, HPK_ERR_INVALID, "bias1 and bias2 go together");
    if (!derive && !band->bias1) return fail(c, HPK_ERR_INVALID, "IR needs bias1 and bias2 (only IR itself, or all three, may be left to the device)");
    return HPK_OK;
}

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" {

void hpk_result_free(hpk_result* res) {
    if (res) delete reinterpret_cast<ResultBox*>(res);
}

}  // extern "C"

// One chromosome of a batch in flight: where its slices of the lane's pooled workspaces are.
struct BandSlot {
    hpk_band in;
    Staged st;
    HpkBandDesc d;                      // host copy of the descriptor
    int32_t n = 0, num = 0, ntiles = 0;
    int32_t ntiles_full = 0;            // tiles under the plan's own halo (the geometry of a chromosome computed once more)
    int64_t ld = 0, cap = 0, band_px = 0, ldo = 0;
    size_t off_rowlive = 0, off_inl = 0, head_bytes = 0, off_hacc = 0, off_tc = 0, off_cnt = 0, off_cu = 0, zero_bytes = 0;
    size_t small_off = 0, head_off = 0; // inside Lane::small / Lane::h_head
    size_t dense_elems = 0;
    bool redone = false, overflowed = false, finished = false, rescored = false;
    int cls = -1;                       // depth class hpk_band_class put the band in (-1: not classified)
    bool canon_done = false;            // spec_halo = 2: computed once more under the halo of its own frozen width
    bool rest_fetched = false;          // the survivors beyond the inline head already sit in `rest` (overflow rerun)
    HpkSurv* rest = nullptr;            // ... in the lane's pinned arena (Lane::h_rest)
    int status = HPK_OK;
    std::string err;
    ResultBox* box = nullptr;
};

struct hpk_job {
    hpk_ctx* ctx = nullptr;
    int lane = -1;                      // (-1: a job of back-end #0 - computed when it is collected, on host threads)
    HpkDevPlan* cpu_plan = nullptr;
    hpk_params prm, key;
    std::vector<BandSlot> bands;
    HpkStencilArgs sa;
    HpkScoreArgs sc;
    HpkGeo gs, gf;                      // tile geometry of the batch's launches (a band may lay its tiles out for a narrower halo of its own:
                                        // hpk_band_class) | of a chromosome computed once more on its own: the plan's
    int nsets = 0, rounds_eff = 0, gmax = 0, tr_cap = 127;
    bool sums = false, dense = false, do_score = true, phases = false, simple = false, use_s = false, balf64 = false,
         time_stencil = true, use_class = false, use_lean = false,
         canon = false;                 // spec_halo = 2: every chromosome ends up under the halo of its own frozen width (collect_impl)
    HpkClassArgs cargs;                 // hpk_band_class as launched for the batch (use_class || use_lean): the second pass of spec_halo = 2 runs it again, per chromosome
    size_t max_zero = 0, max_head = 0;
    signed char class_tab[HPK_NCLASS];  // the depth classes' widths as uploaded for this batch (source of an asynchronous copy: lives with the job)
    double t_begin = 0.0;
    uint8_t kmin_host[HPK_NFAM];        // the survivor bound of the batch's scoring launches (HpkScoreArgs::kmin), if any
    ~hpk_job() { for (BandSlot& b : bands) delete b.box; delete cpu_plan; }
};

namespace {

HpkBandDesc* lane_desc(Lane& L, const hpk_job* j, int b, bool solo) {
    return L.desc.as<HpkBandDesc>() + (solo ? (int)j->bands.size() + b : b);
}

// the kernels of bands [b0, b0 + nbl) of a job whose counter blocks are zero: stencil (+ the freeze decision), scoring,
// cut, copy-back.  solo: band b0 alone, through its second descriptor (records for every resolved candidate).
int launch_compute(hpk_ctx* c, hpk_job* j, int b0, int nbl, bool solo, bool with_stencil, bool all_survivors = false, bool solo_lean = false) {
    Lane& L = c->lane[j->lane];
    const HpkDevPlan& plan = L.plan_host;
    const HpkBandDesc* dd = lane_desc(L, j, b0, solo);
    if (with_stencil) {
        if (j->time_stencil) (void)hipEventRecord(L.ev[1], c->stream);
        {
            HpkStencilArgs sa = j->sa;
            sa.nbands = nbl;
            int kall = 0;
            for (int b = b0; b < b0 + nbl; ++b) kall += j->bands[b].d.chunk;
            sa.grid = std::max(8, std::min((c->cus / 8) * 8, kall * 8));
            if (c->opt.grid_cap > 0) sa.grid = std::max(8, std::min(sa.grid, c->opt.grid_cap / 8 * 8));
            if (solo && !solo_lean) { sa.lean_max = 0; sa.redoq = nullptr; }      // (a chromosome on its own again: every tile in full;
            else if (sa.redoq) HIPCHK(c, hipMemsetAsync(sa.redoq, 0, 16, c->stream));    //  solo_lean: the second pass of spec_halo = 2, like a batch of one)
            hpk_launch_stencil_batch(sa, dd, j->balf64, c->cus, c->stream);
            HIPCHK(c, hipGetLastError());
        }
        if (j->time_stencil) (void)hipEventRecord(L.ev[2], c->stream);
        if (!j->do_score) {         // no scoring kernel to replay the freeze decision: a one-workgroup kernel per band
            hpk_launch_freeze_tot(L.plan.as<HpkDevPlan>(), dd, nbl, c->stream);
            HIPCHK(c, hipGetLastError());
        }
        if (j->phases) (void)hipEventRecord(L.ev[3], c->stream);
    }
    if (j->do_score) {
        HpkScoreArgs sc = j->sc;
        if (all_survivors) sc.kmin = nullptr;       // (a chromosome whose cut lay above the bound of its survivor records)
        if (solo) sc.kcrit = nullptr;               // (a later submission may have rebuilt the critical counts for another sig: look every p up)
        sc.gridx = 0;           // the widest row of scoring workgroups among the bands launched (HpkBandDesc::score_wgs)
        for (int b = b0; b < b0 + nbl; ++b) sc.gridx = std::max(sc.gridx, solo ? j->gmax : j->bands[b].d.score_wgs);
        hpk_launch_score(sc, dd, nbl, plan.mode == HPK_MODE_BHFDR, c->stream);
        HIPCHK(c, hipGetLastError());
        // (a call whose stencil is timed also times its scoring kernel: one more event, behind the stencil's second one)
        if (j->phases || (j->time_stencil && with_stencil)) (void)hipEventRecord(L.ev[4], c->stream);
        hpk_launch_tighten(dd, nbl, j->prm.sig, j->rounds_eff, j->nsets, sc.kmin, c->stream);
        HIPCHK(c, hipGetLastError());
    } else if (j->phases) (void)hipEventRecord(L.ev[4], c->stream);
    if (j->phases) (void)hipEventRecord(L.ev[5], c->stream);
    // gap rows are produced by the stencil kernel (hpk_gap remains as an independent check for the tests)
    // ... as long as the band stops where the command lines stop it (num = D + maxww + 1, scripts/pyHICCUPS:146): the
    // tiles see the diagonals up to about D + maxww.  A caller of the drop-in hiccups() / bhfdr() may hand over more
    // diagonals, and callers.py:238 sums all of them: then the row kernel, which reads every stored diagonal, decides.
    if (with_stencil) {
        for (int b = b0; b < b0 + nbl; ++b) {
            const BandSlot& s = j->bands[b];
            if (c->opt.gap_kernel || s.num > plan.D + plan.W + 1) {
                hpk_launch_gap(s.st.raw, s.st.bal, s.st.weight, s.n, s.num, s.ld, plan.mw, s.d.gap, c->stream);
                HIPCHK(c, hipGetLastError());
            }
        }
    }
    if (j->phases) (void)hipEventRecord(L.ev[6], c->stream);
    // counters, row flags and the first survivors of every band into pinned memory, by a kernel: pinned memory is
    // device-visible, and even one pinned copy per band costs ~15 us of copy-engine start-up
    hpk_launch_publish(dd, nbl, j->nsets, !j->do_score, j->max_head, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(L.ev_done, c->stream));
    return HPK_OK;
}

int submit_impl(hpk_ctx* c, hpk_job* j, const hpk_band* bands, int nb, const hpk_params* prm) {
    Lane& L = c->lane[j->lane];
    const Options& opt = c->opt;
    hpk_params key = *prm;
    key.flags = 0; key.reserved = 0;
    for (int i = key.npairs > 0 ? key.npairs : 0; i < HPK_MAX_PAIRS; ++i) { key.pw[i] = 0; key.ww[i] = 0; }
    const bool plan_hit = L.plan_valid && std::memcmp(&key, &L.plan_key, sizeof(key)) == 0;
    int rc;
    if (!plan_hit) {
        char msg[256];
        L.plan_valid = false;
        rc = hpk_build_plan(prm, &L.plan_host, msg);
        if (rc != HPK_OK) return fail(c, rc, "%s", msg);
    }
    rc = upload_tables(c);
    if (rc != HPK_OK) return rc;
    const HpkDevPlan& plan = L.plan_host;
    const int W = plan.W, mw = plan.mw, D = plan.D;
    if (D < mw) return fail(c, HPK_ERR_INVALID, "maxapart / res (%d) is below min(ww) (%d)", D, mw);
    // Tile geometry under a halo of Wh widths.  Output tile: what the halo leaves of the SAT tile, as many rows as the kernel's
    // tile-wide candidate list holds (HPK_TLIST entries; 7 bits of the record entry).  The tiles of a row block reach the
    // last stored diagonal D + maxww (gap rows, callers.py:238) through the last tile's right halo: with a halo below
    // maxww the chunks themselves have to go further (Dg).  The halo is at least 4 (plans with maxww < 4: the widths beyond
    // maxww have no step, the kernel leaves them out).
    // (rows beyond 64 pay on single-pair plans - chr1 @10 kb: 66 rows at a halo of 6, -4.5 % - and cost 2 % on the
    // three-slot union plan, measured with HPK_TR_CAP)
    const int tr_cap_s = plan.single_p >= 0 ? opt.tr_cap : std::min(opt.tr_cap, 64);
    auto geo_of = [&](int Wh) { return hpk_geo_of(Wh, W, D, mw, tr_cap_s); };
    auto upt_of = [](const HpkGeo& g) { return ((size_t)g.tilecap + HPK_UNIT - 1) / HPK_UNIT; };      // at most ceil(tilecap / HPK_UNIT) units per tile
    const HpkGeo GF = geo_of(std::max(W, 4));
    j->prm = *prm; j->key = key;
    j->nsets = (plan.mode == HPK_MODE_BHFDR) ? 1 : 2 * plan.npairs;
    j->sums = (prm->flags & HPK_FLAG_DENSE_SUMS) != 0;
    j->dense = j->sums || (prm->flags & HPK_FLAG_DENSE_E) != 0;
    j->do_score = (prm->flags & HPK_FLAG_NO_SCORE) == 0;
    j->phases = (prm->flags & HPK_FLAG_PHASE_TIMING) != 0;
    j->time_stencil = j->phases || !(prm->flags & HPK_FLAG_NO_STENCIL_TIMING);
    j->balf64 = bands[0].balanced != nullptr;
    j->simple = plan.simple_reads != 0;
    const bool dense = j->dense, sums = j->sums;
    if (dense) { HPK_NEED_TEST_KERNELS(c); }
    if (dense && nb != 1) return fail(c, HPK_ERR_INVALID, "the dense outputs (HPK_FLAG_DENSE_*) are for single chromosomes");
    for (int b = 0; b < nb; ++b)
        if ((bands[b].balanced != nullptr) != j->balf64)
            return fail(c, HPK_ERR_INVALID, "the bands of a batch must carry the same kind of input (all `balanced` or all `weight`)");

    // ---- which candidates get a record, and the halo that goes with it.  Dense outputs and probes want every candidate;
    // the scoring kernel needs those resolved up to the width the widening freezes at, which the chromosomes collected
    // last with the same parameters tell within a step or so (option spec = 0: no guess, every resolved candidate;
    // spec_margin: widths added to it).  Widths beyond that bound are then not looked at at all: the search stops at it,
    // the resolve counts beyond it stay 0 - the freeze decision up to the bound does not read them - and the tiles carry
    // the bound's halo instead of maxww's (larger output tiles, fewer of them; option spec_halo = 0: the plan's own).  A
    // chromosome that does not freeze by the bound is computed once more under the plan's geometry (collect_impl).
    int wg_all = 255;
    if (j->do_score && !dense) {
        wg_all = W;
        if (opt.spec && c->hint_n > 0 && std::memcmp(&key, &c->hint_key, sizeof(key)) == 0) {
            int hw = -1;
            for (int i = 0; i < c->hint_n; ++i) hw = std::max(hw, c->hint_w[i]);
            // ... and no narrower than the widest width a depth class is known to freeze at: hpk_band_class gives a band its class'
            // width only up to the batch's bound, and a chromosome of a deeper sample than the last collections' would run under a
            // bound it is known to exceed (computed once more, every time).  Every band lays its tiles out for its own bound, so a
            // wide batch bound costs the shallow ones nothing.
            if (opt.spec_class)
                for (int i = 0; i < HPK_NCLASS; ++i) hw = std::max(hw, (int)std::max(c->class_w[i], c->class_w1[i]));
            if (hw >= 0) wg_all = std::min(W, hw + opt.spec_margin);
        }
        if (opt.spec_force >= 0) wg_all = std::min(W, opt.spec_force);        // tests: a bound that is too narrow
        if (!j->simple) wg_all = W;     // (plans without a monotone Reads matrix: no width to bound the records by)
    }
    int64_t max_ld = 0;
    int32_t max_n = 0, max_num = 0, max_dn = 0, max_dnum = 0;
    for (int b = 0; b < nb; ++b) {
        max_ld = std::max<int64_t>(max_ld, bands[b].ld); max_n = std::max(max_n, bands[b].n); max_num = std::max(max_num, bands[b].num);
    }
    j->use_s = true;
    if (!hpk_stencil_s_applies(GF, max_ld, max_n))
        return fail(c, HPK_ERR_INVALID, "band outside the stencil's addressing limits (n < 2^27, ld <= 2^21)");
    HpkGeo GS = GF;
    if (j->simple && opt.spec_halo && wg_all < W) {
        const int Wh = std::min(W, std::max(std::max(wg_all, (int)plan.wmin), 4));
        const HpkGeo g = geo_of(Wh);
        if (Wh < W && hpk_stencil_s_applies(g, max_ld, max_n)) GS = g;
    }
    j->gs = GS; j->gf = GF; j->tr_cap = tr_cap_s;
    // Under the batch's bound every band may get a narrower one of its own (hpk_band_class, below), and with it its own halo and tile
    // geometry - decided on the device, where the bands are: the record regions and work lists are sized for the worst of the
    // geometries a band can end up with (halos from the plan's narrowest width to the batch's).
    const bool class_job = opt.spec && opt.spec_class && j->simple && j->do_score && !dense && wg_all < W && wg_all > (int)plan.wmin;
    const bool band_halo = class_job && opt.spec_halo && GS.W > std::max((int)plan.wmin, 4);
    // Lean tiles: hpk_band_class tells, per band, from which column chunk on the tiles hold (next to) no candidate that resolves within
    // the band's bound; those are hpk_stencil_lean's, built without their f64 plane.  Weight input, a monotone Reads matrix, records
    // under a bound (no dense outputs), and not under spec_halo = 0, whose runs promise bit-identical values whatever the
    // context scored before (which tiles are lean depends on the bound, and their few sums are formed cell by cell).
    const bool lean_job = opt.lean && opt.lean_max > 0 && opt.spec_halo != 0 && j->simple && j->do_score && !dense && !j->balf64 && wg_all != 255;
    std::vector<HpkGeo> geos{GS, GF};
    if (band_halo) for (int Wh = std::max((int)plan.wmin, 4); Wh < GS.W; ++Wh) geos.push_back(geo_of(Wh));
    // (spec_halo = 2: a chromosome may be computed once more under the halo of any width it can freeze at)
    j->canon = opt.spec_halo == 2 && j->simple && j->do_score && !dense && wg_all != 255;
    if (j->canon) for (int Wh = std::max((int)plan.wmin, 4); Wh < GF.W; ++Wh) geos.push_back(geo_of(Wh));
    const int hbins = (opt.rounds <= -2) ? hpk_score_hist_bins((plan.mode == HPK_MODE_BHFDR) ? 1 : 2 * plan.npairs, plan.mode == HPK_MODE_BHFDR) : 0;
    // ---- geometry, sizes and offsets of every band's slices
    const int TR = GS.TR, TC = GS.TC, J_ = GS.J, tilecap = GS.tilecap;
    (void)TC;
    size_t upt = 0;
    for (const HpkGeo& g : geos) upt = std::max(upt, upt_of(g));
    const size_t etab_el = std::max<size_t>((size_t)plan.nsteps * 2 * (D + 1), 1);
    const size_t eedge_el = std::max<size_t>((size_t)2 * W * plan.nsteps * 2 * (D + 1), 1);
    j->rounds_eff = (opt.rounds <= -2) ? -100 - hbins : opt.rounds;
    j->gmax = j->do_score ? hpk_score_grid(plan.mode == HPK_MODE_BHFDR, plan.npairs, hbins, c->cus) : 0;
    j->bands.resize(nb);
    size_t tot_rec = 0, tot_recS = 0, tot_units = 0, tot_small = 0, tot_surv = 0, tot_head = 0, tot_ps = 0, tot_ir = 0, tot_n = 0,
           tot_rawel = 0, tot_hn = 0;
    std::vector<size_t> off_rec(nb), off_recS(nb), off_units(nb), off_surv(nb), off_ps(nb), off_ir(nb), off_n(nb), off_rawel(nb),
                        off_hn(nb);
    bool any_derive = false;
    int k0 = 0;
    for (int b = 0; b < nb; ++b) {
        BandSlot& s = j->bands[b];
        s.in = bands[b];
        const int n = bands[b].n, num = bands[b].num;
        s.n = n; s.num = num; s.ld = bands[b].ld;
        s.ntiles = ((n + TR - 1) / TR) * J_;
        s.ntiles_full = ((n + GF.TR - 1) / GF.TR) * GF.J;
        size_t tiles_max = 0, rec_max = 0;
        for (const HpkGeo& g : geos) {
            const size_t nt = (size_t)((n + g.TR - 1) / g.TR) * g.J;
            tiles_max = std::max(tiles_max, nt);
            rec_max = std::max(rec_max, nt * (size_t)g.tilecap);
        }
        int64_t band_px = 0;                    // pixels with mw <= d <= D inside the matrix
        for (int d = mw; d <= std::min(D, num - 1); ++d) if (n - d > 0) band_px += n - d;
        s.band_px = band_px;
        const bool derive = !bands[b].IR;
        any_derive = any_derive || derive;
        // scoring workgroups of the band: a single chromosome gets the resident grid; in a batch the bands share it
        int wgs = j->gmax;
        if (nb > 1) wgs = std::max(16, std::min(j->gmax, (s.ntiles + opt.score_div - 1) / opt.score_div));
        // survivor capacity per region; every scoring wave may hold one partly filled chunk of HPK_SCH records.  (band pixels x sets /
        // surv_div: Poisson bands hold 0.4 % (depth 15) to 6.3 % (depth 150) of their band pixels as p <= 0.1 records over both sets,
        // the default 6 leaves room for 33 %; a chromosome with more is scored once more with room for everything - below.)
        int64_t cap = (std::max<int64_t>(1 << 16, band_px * j->nsets / opt.surv_div) + (int64_t)std::max(wgs, 1) * 4 * HPK_SCH * 2) / HPK_NREG;
        if (opt.surv_cap > 0) cap = std::max<int64_t>(256, opt.surv_cap);       // tests: force the overflow rerun
        cap = (cap + 255) / 256 * 256;
        s.cap = cap;
        // One block per band, zero-filled by the table kernel:
        //   head (published to the host):  counters | row-has-signal flags [n] | first HEAD_INLINE compacted survivors
        //   scratch:                       resolve totals | per-tile record counts | tightening counters | chunk fill counts
        s.off_rowlive = up256(HPK_SMALL_BYTES);
        s.off_inl = up256(s.off_rowlive + (size_t)n);
        s.head_bytes = s.off_inl + sizeof(HpkSurv) * HPK_HEAD_INLINE;
        s.off_hacc = up256(s.head_bytes);
        s.off_tc = up256(s.off_hacc + 8 * (size_t)HPK_HREP * HPK_ACC_STRIDE);
        s.off_cnt = up256(s.off_tc + sizeof(unsigned) * tiles_max);
        s.off_cu = up256(s.off_cnt + sizeof(unsigned) * HPK_NFAM * HPK_TIGHTEN_MAX);
        const size_t off_wnz = up256(s.off_cu + sizeof(unsigned) * (size_t)(cap / HPK_SCH * HPK_NREG + 1));
        s.zero_bytes = (off_wnz + (lean_job ? sizeof(unsigned) * (size_t)HPK_WNZ_WORDS(n) : 0) + 4095) / 4096 * 4096;
        s.small_off = tot_small; tot_small += s.zero_bytes;
        s.head_off = tot_head; tot_head += up256(s.head_bytes);
        off_rec[b] = tot_rec; tot_rec += rec_max;
        off_recS[b] = tot_recS; tot_recS += rec_max * plan.nslots;
        off_units[b] = tot_units; tot_units += tiles_max * upt + 16;
        off_surv[b] = tot_surv; tot_surv += (size_t)cap * HPK_NREG;
        off_ir[b] = tot_ir; off_n[b] = tot_n;
        if (derive || !bands[b].on_device) { tot_ir += ((size_t)num + 31) / 32 * 32; tot_n += ((size_t)n + 31) / 32 * 32; }
        off_ps[b] = tot_ps;
        if (derive) {
            tot_ps += (size_t)((n + 31) / 32) * num;       // room for one partial per 32 rows (hpk_ir_partial takes 128 per partial)
            max_dn = std::max(max_dn, n); max_dnum = std::max(max_dnum, num);
        }
        off_rawel[b] = tot_rawel; off_hn[b] = tot_hn;
        if (!bands[b].on_device) { tot_rawel += (size_t)n * (size_t)s.ld; tot_hn += ((size_t)n + 31) / 32 * 32; }
        j->max_zero = std::max(j->max_zero, s.zero_bytes);
        j->max_head = std::max(j->max_head, s.head_bytes);
        s.ldo = ((int64_t)(D + 1) + 31) / 32 * 32;
        s.dense_elems = (size_t)plan.nslots * (size_t)n * (size_t)s.ldo;
        s.box = new ResultBox();
        std::memset(&s.box->pub, 0, sizeof(s.box->pub));
        // descriptor: geometry
        HpkBandDesc& d = s.d;
        std::memset(&d, 0, sizeof(d));
        d.n = n; d.num = num; d.ld = s.ld; d.ntiles = s.ntiles; d.chunk = (s.ntiles + 7) / 8; d.k0 = k0;
        k0 += d.chunk;
        d.rec_stride = (int64_t)s.ntiles * tilecap;
        d.cap = cap; d.zero_bytes = s.zero_bytes; d.off_rowlive = (uint32_t)s.off_rowlive; d.off_inl = (uint32_t)s.off_inl;
        d.derive = derive ? (bands[b].bias1 ? 2 : 1) : 0;       // 1: IR and biases, 2: IR only
        d.lean_cj = 0x7fffffff;                                 // (no lean tiles unless hpk_band_class says where)
        d.off_wnz = lean_job ? (uint32_t)off_wnz : 0u;
        d.W = GS.W; d.Dg = GS.Dg; d.TR = GS.TR; d.TC = GS.TC; d.J = GS.J; d.tilecap = GS.tilecap;
        d.score_wgs = wgs;
    }

    // ---- workspaces
    HIPCHK(c, L.recE.reserve(sizeof(unsigned) * tot_rec));
    HIPCHK(c, L.recS.reserve(sizeof(double2) * tot_recS));
    HIPCHK(c, L.recW.reserve(tot_recS));
    HIPCHK(c, L.units.reserve(sizeof(uint2) * tot_units));
    HIPCHK(c, L.small.reserve(tot_small));
    HIPCHK(c, L.etab.reserve(sizeof(double) * etab_el * nb));
    HIPCHK(c, L.eedge.reserve(sizeof(double) * eedge_el * nb));
    HIPCHK(c, L.desc.reserve(sizeof(HpkBandDesc) * 2 * (size_t)nb));
    if (j->do_score) {
        HIPCHK(c, L.surv.reserve(sizeof(HpkSurv) * tot_surv));
        HIPCHK(c, L.surv2.reserve(sizeof(HpkSurv) * tot_surv));
    }
    if (tot_ir) { HIPCHK(c, L.IR.reserve(sizeof(double) * tot_ir)); HIPCHK(c, L.b1.reserve(sizeof(double) * tot_n)); HIPCHK(c, L.b2.reserve(sizeof(double) * tot_n)); }
    if (tot_ps) { HIPCHK(c, L.psum.reserve(sizeof(double) * tot_ps)); HIPCHK(c, L.pnan.reserve(sizeof(unsigned) * tot_ps)); }
    if (tot_rawel) {
        HIPCHK(c, L.raw.reserve(sizeof(float) * tot_rawel));
        if (j->balf64) HIPCHK(c, L.bal.reserve(sizeof(double) * tot_rawel));
        HIPCHK(c, L.weight.reserve(sizeof(double) * tot_hn));
    }
    if (dense) {
        const size_t de = j->bands[0].dense_elems;
        HIPCHK(c, L.dE.reserve(sizeof(double2) * de));
        HIPCHK(c, L.dW.reserve(de));
        if (sums) HIPCHK(c, L.dS.reserve(sizeof(double4) * de));
    }
    if (L.h_head_cap < tot_head) {
        if (L.h_head) (void)hipHostFree(L.h_head);
        L.h_head = nullptr; L.h_head_cap = 0;
        HIPCHK(c, hipHostMalloc(&L.h_head, tot_head + tot_head / 4, hipHostMallocMapped));
        L.h_head_cap = tot_head + tot_head / 4;
    }
    const size_t desc_bytes = sizeof(HpkBandDesc) * 2 * (size_t)nb;
    if (L.h_desc_cap < desc_bytes) {
        if (L.h_desc) (void)hipHostFree(L.h_desc);
        L.h_desc = nullptr; L.h_desc_cap = 0;
        HIPCHK(c, hipHostMalloc(&L.h_desc, desc_bytes * 2, hipHostMallocDefault));
        L.h_desc_cap = desc_bytes * 2;
    }

    // ---- inputs: host arrays are uploaded on the lane's side stream; device arrays are used where they are
    if (j->phases) (void)hipEventRecord(L.ev[0], c->stream);
    for (int b = 0; b < nb; ++b) {
        BandSlot& s = j->bands[b];
        const hpk_band& in = s.in;
        const size_t n = (size_t)s.n, num = (size_t)s.num, ld = (size_t)s.ld;
        Staged& st = s.st;
        const bool derive = s.d.derive != 0;
        if (in.on_device) {
            st.raw = in.raw; st.bal = in.balanced; st.weight = in.weight;
            st.IR = const_cast<double*>(in.IR); st.b1 = const_cast<double*>(in.bias1); st.b2 = const_cast<double*>(in.bias2);
        } else {
            float* draw = L.raw.as<float>() + off_rawel[b];
            HIPCHK(c, hipMemcpyAsync(draw, in.raw, sizeof(float) * n * ld, hipMemcpyHostToDevice, L.up));
            st.raw = draw;
            if (in.balanced) {
                double* dbal = L.bal.as<double>() + off_rawel[b];
                HIPCHK(c, hipMemcpyAsync(dbal, in.balanced, sizeof(double) * n * ld, hipMemcpyHostToDevice, L.up));
                st.bal = dbal;
            }
            if (in.weight) {
                double* dw = L.weight.as<double>() + off_hn[b];
                HIPCHK(c, hipMemcpyAsync(dw, in.weight, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
                st.weight = dw;
            }
            if (!derive) {
                st.IR = L.IR.as<double>() + off_ir[b];
                HIPCHK(c, hipMemcpyAsync(st.IR, in.IR, sizeof(double) * num, hipMemcpyHostToDevice, L.up));
                st.b1 = L.b1.as<double>() + off_n[b];
                HIPCHK(c, hipMemcpyAsync(st.b1, in.bias1, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
                if (in.bias2 == in.bias1) st.b2 = st.b1;
                else {
                    st.b2 = L.b2.as<double>() + off_n[b];
                    HIPCHK(c, hipMemcpyAsync(st.b2, in.bias2, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
                }
            }
        }
        if (derive) {      // IR / biases from raw + weight on the device (scripts/pyHICCUPS:149-166): hpk_launch_prep below
            st.IR = L.IR.as<double>() + off_ir[b];
            if (!in.bias1) {
                st.b1 = L.b1.as<double>() + off_n[b];
                st.b2 = st.b1;
            } else if (!in.on_device) {     // IR only: the caller's biases (e.g. of a divisive weight column) are uploaded
                st.b1 = L.b1.as<double>() + off_n[b];
                HIPCHK(c, hipMemcpyAsync(st.b1, in.bias1, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
                if (in.bias2 == in.bias1) st.b2 = st.b1;
                else {
                    st.b2 = L.b2.as<double>() + off_n[b];
                    HIPCHK(c, hipMemcpyAsync(st.b2, in.bias2, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
                }
            }
        }
        // descriptor: pointers
        HpkBandDesc& d = s.d;
        unsigned char* small = L.small.as<unsigned char>() + s.small_off;
        d.raw = st.raw; d.weight = st.weight; d.bal = st.bal;
        d.rec_ent = L.recE.as<unsigned>() + off_rec[b];
        d.rec_S = L.recS.as<double2>() + off_recS[b];
        d.rec_W = L.recW.as<uint8_t>() + off_recS[b];
        d.tile_cnt = reinterpret_cast<unsigned*>(small + s.off_tc);
        d.units = L.units.as<uint2>() + off_units[b];
        d.small = small;
        d.gap = small + s.off_rowlive;
        d.hist_acc = reinterpret_cast<unsigned long long*>(small + s.off_hacc);
        d.IR = st.IR; d.b1 = st.b1; d.b2 = st.b2;
        d.etab = L.etab.as<double>() + etab_el * b;
        d.eedge = L.eedge.as<double>() + eedge_el * b;
        if (derive) { d.psum = L.psum.as<double>() + off_ps[b]; d.pnan = L.pnan.as<unsigned>() + off_ps[b]; }
        if (j->do_score) { d.surv = L.surv.as<HpkSurv>() + off_surv[b]; d.surv2 = L.surv2.as<HpkSurv>() + off_surv[b]; }
        d.chunk_used = reinterpret_cast<unsigned*>(small + s.off_cu);
        d.cnt = reinterpret_cast<unsigned*>(small + s.off_cnt);
        d.head_host = static_cast<unsigned char*>(L.h_head) + s.head_off;
        d.wguess = wg_all;
    }
    {   // descriptors: [0, nb) as the batch runs them, [nb, 2 nb) for a chromosome computed once more on its own
        HpkBandDesc* hd = static_cast<HpkBandDesc*>(L.h_desc);
        for (int b = 0; b < nb; ++b) {
            hd[b] = j->bands[b].d;
            hd[nb + b] = j->bands[b].d;
            hd[nb + b].k0 = 0;
            if (hd[nb + b].wguess != 255) hd[nb + b].wguess = W;
            hd[nb + b].ntiles = j->bands[b].ntiles_full;
            hd[nb + b].chunk = (j->bands[b].ntiles_full + 7) / 8;
            hd[nb + b].rec_stride = (int64_t)j->bands[b].ntiles_full * GF.tilecap;
            hd[nb + b].W = GF.W; hd[nb + b].Dg = GF.Dg; hd[nb + b].TR = GF.TR; hd[nb + b].TC = GF.TC; hd[nb + b].J = GF.J;
            hd[nb + b].tilecap = GF.tilecap;
            hd[nb + b].score_wgs = j->gmax;
        }
        HIPCHK(c, hipMemcpyAsync(L.desc.p, hd, desc_bytes, hipMemcpyHostToDevice, L.up));
    }
    if (!plan_hit) {
        HIPCHK(c, L.plan.reserve(sizeof(HpkDevPlan)));
        HIPCHK(c, hipMemcpyAsync(L.plan.p, &L.plan_host, sizeof(HpkDevPlan), hipMemcpyHostToDevice, L.up));
        L.plan_key = key;
        L.plan_valid = true;
    }
    // The derivation of IR / biases and the expected tables (the same launch zero-fills the counter blocks) run on the
    // lane's side stream like the uploads: for the batch submitted one ahead this executes beside the scoring and cut
    // kernels of the batch before (those leave registers and LDS free; the stencil does not); the compute stream waits
    // for the lot.
    hipStream_t side = L.up;
    if (opt.side_serial) {      // (A/B: the same kernels in line on the compute stream, behind the uploads above)
        HIPCHK(c, hipEventRecord(L.ev_up, L.up));
        HIPCHK(c, hipStreamWaitEvent(c->stream, L.ev_up, 0));
        side = c->stream;
    }
    if (any_derive) {
        hpk_launch_prep(L.desc.as<HpkBandDesc>(), nb, max_dn, max_dnum, mw, side);
        HIPCHK(c, hipGetLastError());
    }
    hpk_launch_etab(L.plan.as<HpkDevPlan>(), L.desc.as<HpkBandDesc>(), nb, plan.nsteps, D, W, j->max_zero, side);
    HIPCHK(c, hipGetLastError());
    // Record bound per chromosome: the batch's bound comes from the widest freeze of the last collections; a band of a shallower
    // sample freezes earlier, and what it writes beyond its own width is read and dropped by the scoring kernel.  hpk_band_class
    // sorts the bands into depth classes on the device and gives each the width its class froze at last (verified at collection
    // like the batch's bound: a chromosome that froze later is computed once more).
    j->use_class = class_job;
    j->use_lean = lean_job;
    if (j->use_class || j->use_lean) {
        HpkClassArgs ca;
        std::memset(&ca, 0, sizeof(ca));
        if (j->use_class) {
            signed char* tab = j->class_tab;
            if (c->hint_n > 0 && std::memcmp(&key, &c->hint_key, sizeof(key)) == 0)
                for (int i = 0; i < HPK_NCLASS; ++i) tab[i] = std::max(c->class_w[i], c->class_w1[i]);
            else std::memset(tab, -1, HPK_NCLASS);
            HIPCHK(c, L.classtab.reserve(HPK_NCLASS));
            HIPCHK(c, hipMemcpyAsync(L.classtab.p, tab, HPK_NCLASS, hipMemcpyHostToDevice, side));
            ca.table = L.classtab.as<signed char>();
        }
        ca.mw = mw; ca.D = D; ca.wg_all = wg_all; ca.margin = opt.spec_margin; ca.wmin = (int)plan.wmin;
        ca.lean = j->use_lean ? 1 : 0;
        ca.halo = band_halo ? 1 : 0; ca.planW = W; ca.tr_cap = tr_cap_s;
        ca.p0 = plan.reads_p0; ca.minr = plan.min_reads;
        ca.lean_frac = (float)opt.lean_frac_pct / 100.f;
        ca.lean_share = opt.lean_share_pct;
        j->cargs = ca;
        hpk_launch_band_class(L.desc.as<HpkBandDesc>(), nb, ca, side);
        HIPCHK(c, hipGetLastError());
        if (opt.host_prof >= 2 && j->use_lean) {      // debug: the non-zero-weight mask of band 0 against its weights
            HIPCHK(c, hipStreamSynchronize(L.up));
            const BandSlot& s0 = j->bands[0];
            std::vector<unsigned> m(HPK_WNZ_WORDS(s0.n));
            std::vector<double> w(s0.n);
            HIPCHK(c, hipMemcpy(m.data(), s0.d.small + s0.d.off_wnz, m.size() * 4, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(w.data(), s0.st.weight, w.size() * 8, hipMemcpyDeviceToHost));
            int bad = 0, nan = 0;
            for (int i = 0; i < s0.n; ++i) {
                const int bit = (m[(i + HPK_WNZ_LEAD) >> 5] >> ((i + HPK_WNZ_LEAD) & 31)) & 1;
                const int want = (w[i] == w[i] && w[i] != 0.0) ? 1 : 0;
                nan += w[i] != w[i];
                if (bit != want && bad++ < 8) std::fprintf(stderr, "[hpk dbg] wnz bin %d bit %d want %d w %g\n", i, bit, want, w[i]);
            }
            HpkBandDesc dd;
            HIPCHK(c, hipMemcpy(&dd, L.desc.as<HpkBandDesc>(), sizeof(dd), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[hpk dbg] wnz: %d bins, %d NaN weights, %d wrong bits; lean_cj %d J %d off_wnz %u\n", s0.n, nan, bad, dd.lean_cj, dd.J, dd.off_wnz);
        }
    }
    HIPCHK(c, hipEventRecord(L.ev_up, L.up));
    HIPCHK(c, hipStreamWaitEvent(c->stream, L.ev_up, 0));
    if (dense) {   // pixels outside the band are never written by the kernel
        const size_t de = j->bands[0].dense_elems;
        HIPCHK(c, hipMemsetAsync(L.dE.p, 0, sizeof(double2) * de, c->stream));
        HIPCHK(c, hipMemsetAsync(L.dW.p, 0, de, c->stream));
        if (sums) HIPCHK(c, hipMemsetAsync(L.dS.p, 0, sizeof(double4) * de, c->stream));
    }

    // ---- common kernel arguments
    HpkStencilArgs& sa = j->sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.plan = L.plan.as<HpkDevPlan>();
    sa.risk = std::ldexp(1.0, -opt.risk_log2);
    sa.nbands = nb;
    sa.mw = mw; sa.D = D;
    sa.single = plan.single_p >= 0 ? 1 : 0;
    sa.generic = j->simple ? 0 : 1;
    sa.order = opt.tile_order;
    sa.dbg_stop = opt.dbg_stop;
    sa.clk = nullptr;
    sa.lean_max = j->use_lean ? opt.lean_max : 0;
    sa.redoq = nullptr;
    if (j->use_lean) {          // the tiles the lean kernel gives up: at most all of them
        size_t tiles_all = 0;
        for (int b = 0; b < nb; ++b) tiles_all += (size_t)std::max(j->bands[b].ntiles, j->bands[b].ntiles_full) * 2;
        HIPCHK(c, L.redoq.reserve(16 + 8 * tiles_all));
        sa.redoq = L.redoq.as<unsigned>();
    }
#ifdef HPK_PHASE_CLOCK
    if (std::getenv("HPK_CLK_DUMP")) {
        HIPCHK(c, c->tmpD.reserve(sizeof(unsigned long long) * (8 * HPK_NWAVES * 1024 + 8)));
        HIPCHK(c, hipMemsetAsync(c->tmpD.p, 0, sizeof(unsigned long long) * 8 * HPK_NWAVES * 1024, c->stream));
        sa.clk = c->tmpD.as<unsigned long long>();
    }
#endif
    HpkScoreArgs& sc = j->sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.plan = sa.plan;
    sc.bounds = c->d_bounds.as<double>(); sc.ptab = c->d_ptab.as<double>();
    sc.ptab_off = c->d_off.as<int32_t>(); sc.sfe = c->d_sfe.as<double>(); sc.sig = prm->sig;
    sc.mw = mw; sc.D = D;
    if (!opt.kcrit) {
    } else if (plan.mode == HPK_MODE_HICCUPS && !(prm->sig >= 1.0)) {      // the critical counts of this sig (cached: one tiny launch per change)
        if (c->kcrit_sig != prm->sig) {
            HIPCHK(c, c->d_kcrit.reserve(sizeof(int32_t) * (HPK_NB_TAB + 2)));
            hpk_launch_kcrit(c->d_ptab.as<double>(), c->d_off.as<int32_t>(), prm->sig, c->d_kcrit.as<int32_t>(), c->stream);
            HIPCHK(c, hipGetLastError());
            c->kcrit_sig = prm->sig;
        }
        sc.kcrit = c->d_kcrit.as<int32_t>();
    } else if (plan.mode == HPK_MODE_BHFDR && prm->sig < 0.25) {   // bhfdr: critical counts on a grid over lambda, likewise
        if (c->kclam_sig != prm->sig) {
            HIPCHK(c, c->d_kclam.reserve(sizeof(int32_t) * HPK_KCL_N));
            hpk_launch_kcrit_lam(c->d_sfe.as<double>(), prm->sig, c->d_kclam.as<int32_t>(), c->stream);
            HIPCHK(c, hipGetLastError());
            c->kclam_sig = prm->sig;
        }
        sc.kcrit = c->d_kclam.as<int32_t>();
    }
#ifdef HPK_PHASE_CLOCK
    if (std::getenv("HPK_CLK_DUMP")) {          // hpk_score's accumulators: behind the stencil's region of tmpD
        sc.clk = c->tmpD.as<unsigned long long>() + (size_t)8 * HPK_NWAVES * 1024;
        HIPCHK(c, hipMemsetAsync(sc.clk, 0, 8 * 8, c->stream));
        HIPCHK(c, hipMemsetAsync(sc.clk + 6, 0xff, 8, c->stream));
    }
#endif
    sc.hbins = hbins; sc.nsets_half = plan.npairs;
    // Survivor records: p <= sig is what can reach q <= sig, but the Benjamini-Hochberg cut of a family lies orders of
    // magnitude below sig (sig x rejections / tests), and the chromosomes collected last with these parameters say in which
    // histogram bin.  The launch writes records from that bin (minus a margin) on; hpk_thr_compact verifies.
    sc.kmin = nullptr;
    if (hbins > 0 && opt.spec_surv && j->do_score && !dense) {
        uint8_t* km = j->kmin_host;
        bool have = false;
        if (opt.spec_surv_force >= 0) {
            std::memset(km, std::min(opt.spec_surv_force, hbins - 1), HPK_NFAM);
            have = true;
        } else if (c->hint_bn > 0 && c->hint_n > 0 && std::memcmp(&key, &c->hint_key, sizeof(key)) == 0) {
            for (int f = 0; f < HPK_NFAM; ++f) {
                int kb = 255;
                for (int i = 0; i < c->hint_bn; ++i) kb = std::min<int>(kb, c->hint_bin[i][f]);
                // (margin in factor-4 bins; bhfdr's fine bins are four to the octave)
                km[f] = (uint8_t)std::max(0, std::min(kb, hbins - 1) - opt.spec_surv_margin * (hbins > 16 ? 8 : 1));
            }
            have = true;
        }
        if (have) {
            HIPCHK(c, L.kmin.reserve(HPK_NFAM));
            HIPCHK(c, hipMemcpyAsync(L.kmin.p, km, HPK_NFAM, hipMemcpyHostToDevice, c->stream));
            sc.kmin = L.kmin.as<uint8_t>();
            if (opt.host_prof) std::fprintf(stderr, "[hpk host] survivor bound: bins %d %d %d ... (hbins %d)\n", km[1], km[2], km[HPK_NB + 2], hbins);
        }
    }
    return launch_compute(c, j, 0, nb, false, true);
}

// The host half both back-ends share: lambda chunks (callers.py:30) + Benjamini-Hochberg per family (callers.py:273 / 545) on the
// pixels with p <= sig that came back, and the result arrays.  h_chist / h_famf: tests per (set, chunk) family and how many of them
// have p <= sig; h_emax: bit pattern of the largest E per set.
void assemble_sets(const HpkDevPlan& plan, const hpk_params* prm, int nsets, const unsigned int* h_chist, const unsigned int* h_famf,
                   const unsigned long long* h_emax, std::vector<Surv>& sv, ResultBox* box, int host_prof, double t_d2h1) {
    hpk_result& R = box->pub;
    std::vector<std::vector<Surv*>> kept(nsets);
    // families = (set, chunk): one sort by (family, p) on flat 16-byte keys (p >= 0, so its bit pattern orders
    // like its value), then Benjamini-Hochberg on each run
    struct Key { uint32_t fam, idx; uint64_t pbits; };
    std::vector<Key> keys(sv.size());
    for (size_t i = 0; i < sv.size(); ++i) {
        uint64_t bits;
        std::memcpy(&bits, &sv[i].p, 8);
        keys[i] = Key{(uint32_t)sv[i].set << 8 | sv[i].chunk, (uint32_t)i, bits};
    }
    // (dealt to their families first - a counting pass over at most 256 x 256 family codes, a few hundred in use -, then every
    // family sorted by p on its own: tens of thousands of survivors per chromosome on maps with structure, and one sort over all
    // of them by (family, p) was 40 % of this half)
    {
        std::vector<uint32_t> start;
        uint32_t maxf = 0;
        for (const Key& k : keys) maxf = std::max(maxf, k.fam);
        start.assign((size_t)maxf + 2, 0u);
        for (const Key& k : keys) ++start[(size_t)k.fam + 1];
        for (size_t f = 1; f < start.size(); ++f) start[f] += start[f - 1];
        std::vector<Key> dealt(keys.size());
        std::vector<uint32_t> at(start.begin(), start.end() - 1);
        for (const Key& k : keys) dealt[at[k.fam]++] = k;
        for (size_t f = 0; f + 1 < start.size(); ++f)
            if (start[f + 1] - start[f] > 1)
                std::sort(dealt.begin() + start[f], dealt.begin() + start[f + 1], [](const Key& a, const Key& b2) {
                    return a.pbits != b2.pbits ? a.pbits < b2.pbits : a.idx < b2.idx; });
        keys.swap(dealt);
    }
    std::vector<Surv*> order(sv.size());
    for (size_t i = 0; i < sv.size(); ++i) order[i] = &sv[keys[i].idx];
    const double t_h1 = now_ms();
    std::vector<int> numbins(nsets, 0);
    box->fam.resize((size_t)2 * nsets * (HPK_NB + 1));
    std::memcpy(box->fam.data(), h_chist, sizeof(uint32_t) * (size_t)nsets * (HPK_NB + 1));
    std::memcpy(box->fam.data() + (size_t)nsets * (HPK_NB + 1), h_famf, sizeof(uint32_t) * (size_t)nsets * (HPK_NB + 1));
    for (int t = 0; t < nsets; ++t) {
        hpk_set& hs = R.sets[t];
        hs.pair = (plan.mode == HPK_MODE_BHFDR) ? 0 : t / 2;
        hs.fl = (plan.mode == HPK_MODE_BHFDR) ? 0 : t % 2;
        // pixels with E > 0: the families of the set added up (family 0 = those without a chunk, not a family of tests)
        hs.nvalid = 0;
        for (int ch = 0; ch <= HPK_NB; ++ch) hs.nvalid += (int64_t)h_chist[(size_t)t * (HPK_NB + 1) + ch];
        box->fam[(size_t)t * (HPK_NB + 1)] = 0u;
        double emax = 0.0;
        std::memcpy(&emax, &h_emax[t], 8);
        hs.emax = emax;
        int numbin = 0;
        if (plan.mode == HPK_MODE_HICCUPS && hs.nvalid > 0 && emax > 0.0) {
            const double nbd = std::ceil(std::log(emax) / std::log(2.0) * 3.0 + 1.0);     // callers.py:30
            numbin = !(nbd >= 0) ? 0 : (nbd > HPK_NB ? HPK_NB : (int)nbd);
        }
        hs.numbin = numbin;
        hs.chunk_tests = box->fam.data() + (size_t)t * (HPK_NB + 1);
        hs.chunk_below = box->fam.data() + (size_t)(nsets + t) * (HPK_NB + 1);
        // chunks beyond numbin do not exist for the reference (their pixels keep p = q = 1, callers.py:259-260)
        for (int ch = ((plan.mode == HPK_MODE_BHFDR) ? 1 : numbin) + 1; ch <= HPK_NB; ++ch) {
            box->fam[(size_t)t * (HPK_NB + 1) + ch] = 0u;
            box->fam[(size_t)(nsets + t) * (HPK_NB + 1) + ch] = 0u;
        }
        numbins[t] = (plan.mode == HPK_MODE_BHFDR) ? 1 : numbin;
    }
    for (size_t i = 0; i < order.size();) {
        size_t e = i;
        while (e < order.size() && order[e]->set == order[i]->set && order[e]->chunk == order[i]->chunk) ++e;
        const int t = order[i]->set, ch = order[i]->chunk;
        if (t < nsets && ch >= 1 && ch <= numbins[t]) {        // chunks beyond numbin keep p = q = 1 (callers.py:259-260)
            bh_family(order.data() + i, e - i, h_chist[(size_t)t * (HPK_NB + 1) + ch], prm->sig, plan.mode == HPK_MODE_BHFDR);
            for (size_t u = i; u < e; ++u) if (order[u]->keep) kept[t].push_back(order[u]);
        }
        i = e;
    }
    const double t_h2 = now_ms();
    for (int t = 0; t < nsets; ++t) {           // by (x, y): on flat keys, the records themselves are not touched by the sort
        std::vector<std::pair<uint64_t, Surv*>> xy(kept[t].size());
        for (size_t i = 0; i < xy.size(); ++i) xy[i] = {(uint64_t)(uint32_t)kept[t][i]->x << 32 | (uint32_t)kept[t][i]->y, kept[t][i]};
        std::sort(xy.begin(), xy.end(), [](const std::pair<uint64_t, Surv*>& a, const std::pair<uint64_t, Surv*>& b2) { return a.first < b2.first; });
        for (size_t i = 0; i < xy.size(); ++i) kept[t][i] = xy[i].second;
    }
    if (host_prof) std::fprintf(stderr, "[hpk host] n=%zu sort=%.3f bh=%.3f\n", sv.size(), t_h1 - t_d2h1, t_h2 - t_h1);
    size_t total = 0;
    for (int t = 0; t < nsets; ++t) total += kept[t].size();
    box->x.reserve(total); box->y.reserve(total); box->O.reserve(total); box->bal.reserve(total);
    box->E.reserve(total); box->p.reserve(total); box->q.reserve(total); box->oz.reserve(total);
    for (int t = 0; t < nsets; ++t) {
        R.sets[t].begin = (int64_t)box->x.size();
        for (const Surv* p : kept[t]) {
            box->x.push_back(p->x); box->y.push_back(p->y); box->O.push_back((double)p->O); box->bal.push_back(p->bal);
            box->E.push_back(p->E); box->p.push_back(p->p); box->q.push_back(p->q); box->oz.push_back(p->flag);
        }
        R.sets[t].end = (int64_t)box->x.size();
    }
    R.nsig = (int64_t)total;
    R.x = box->x.data(); R.y = box->y.data(); R.O = box->O.data(); R.bal = box->bal.data();
    R.E = box->E.data(); R.p = box->p.data(); R.q = box->q.data(); R.other_zero = box->oz.data();
}

// the host half of one band whose head has landed: widening log, lambda chunks, Benjamini-Hochberg, result arrays
// (`err`: where a message goes when several chromosomes are finished at once - hpk_ctx::err is the calling thread's)
int finish_band(hpk_ctx* c, hpk_job* j, int b, std::string* err = nullptr) {
    Lane& L = c->lane[j->lane];
    const HpkDevPlan& plan = L.plan_host;
    const hpk_params* prm = &j->prm;
    BandSlot& s = j->bands[b];
    const int n = s.n, nsets = j->nsets, mw = plan.mw, D = plan.D;
    const bool sums = j->sums, dense = j->dense, do_score = j->do_score;
    ResultBox* box = s.box;
    hpk_result& R = box->pub;
    const unsigned char* hsmall = static_cast<const unsigned char*>(L.h_head) + s.head_off;
    box->gap.resize(n);
    for (int r = 0; r < n; ++r) box->gap[r] = hsmall[s.off_rowlive + r] ? 0 : 1;
    R.band_px = s.band_px;
    R.stencil_tiles = s.ntiles;
    R.stencil_kernel = 2;
    R.batch_bands = (int32_t)j->bands.size();

    const unsigned long long* h_hist = reinterpret_cast<const unsigned long long*>(hsmall + HPK_OFF_HIST);
    const int32_t h_frozen = *reinterpret_cast<const int32_t*>(hsmall + HPK_OFF_FROZEN);
    const int32_t h_err = *reinterpret_cast<const int32_t*>(hsmall + HPK_OFF_ERR);
    R.record_bound = s.d.wguess;
    R.halo_w = s.d.W;
    R.redone = (s.redone ? 1 : 0) | (s.rescored ? 2 : 0);
    {
        const unsigned* lc = reinterpret_cast<const unsigned*>(hsmall + HPK_OFF_LEAN);
        R.lean_tiles = (int32_t)lc[0]; R.lean_redone = (int32_t)lc[1]; R.lean_explicit = (int64_t)lc[2];
    }
    const int32_t* h_exec = reinterpret_cast<const int32_t*>(hsmall + HPK_OFF_EXEC);
    const unsigned long long h_nsurv = *reinterpret_cast<const unsigned long long*>(hsmall + HPK_OFF_NOUT);
    const unsigned long long* h_emax = reinterpret_cast<const unsigned long long*>(hsmall + HPK_OFF_EMAX);
    const unsigned int* h_chist = reinterpret_cast<const unsigned int*>(hsmall + HPK_OFF_FAM_M);

    R.nsteps = plan.nsteps;
    for (int t = 0; t < plan.nsteps; ++t) {
        R.step_pi[t] = plan.steps[t].pi;
        R.step_wi[t] = plan.steps[t].wi;
        R.step_executed[t] = h_exec[t];
        // (steps the widening never runs - beyond frozen_w, callers.py:133-134 - have no count in the reference; here they
        // are counted only when every candidate gets a record: dense outputs, probes, HPK_FLAG_NO_SCORE)
        R.step_resolved[t] = (h_exec[t] || s.d.wguess == 255) ? (int64_t)h_hist[t] : 0;
    }
    R.frozen_w = h_frozen;
    R.nslots = plan.nslots;
    for (int q = 0; q < plan.nslots; ++q) R.slot_pi[q] = plan.slot_pi[q];
    R.ncand = (int64_t)h_hist[HPK_HIST_NCAND];
    R.nsurv_sig = 0;
    if (do_score)
        for (int i = 0; i < nsets * (HPK_NB + 1); ++i)      // (only the families of the sets in use are copied back)
            R.nsurv_sig += reinterpret_cast<const unsigned int*>(hsmall + HPK_OFF_FAM_F)[i];
    R.nsurv_cut = (int64_t)h_nsurv;
    R.gap = box->gap.data();
    if (h_err != 0 && c->opt.dbg_stop == 0) {
        const HpkDevStep& st = plan.steps[h_err - 1];
        char msg[256];
        std::snprintf(msg, sizeof(msg), "step (%d,%d) entered with no unresolved candidate (of %lld); the reference raises here "
                      "(hicpeaks/callers.py:203-208)", st.pi, st.wi, (long long)R.ncand);
        if (err) { *err = msg; return HPK_ERR_EMPTY_STEP; }
        return fail(c, HPK_ERR_EMPTY_STEP, "%s", msg);
    }

    const double t_d2h0 = now_ms();
    std::vector<Surv> sv;
    if (do_score && h_nsurv) {
        const size_t ns = (size_t)h_nsurv;
        const HpkSurv* head = reinterpret_cast<const HpkSurv*>(hsmall + s.off_inl);
        if (ns > HPK_HEAD_INLINE && !s.rest_fetched) {
            s.rest = static_cast<HpkSurv*>(L.h_rest.take(sizeof(HpkSurv) * (ns - HPK_HEAD_INLINE)));
            if (!s.rest) return fail(c, HPK_ERR_HIP, "pinned memory for %zu survivor records", ns - HPK_HEAD_INLINE);
            // beyond the inlined head: fetched on the lane's own (idle) copy stream, so the next batch's kernels on the
            // compute stream are not waited for
            HIPCHK(c, hipMemcpyAsync(s.rest, s.d.surv2, sizeof(HpkSurv) * (ns - HPK_HEAD_INLINE), hipMemcpyDeviceToHost, L.up));
            HIPCHK(c, hipStreamSynchronize(L.up));
        }
        auto rec_at = [&](size_t i) -> const HpkSurv& { return i < HPK_HEAD_INLINE ? head[i] : s.rest[i - HPK_HEAD_INLINE]; };
        sv.resize(ns);
        for (size_t i = 0; i < ns; ++i) {
            const HpkSurv& rc_ = rec_at(i);
            sv[i] = Surv{rc_.x, rc_.y, rc_.set, rc_.chunk, rc_.flag, 0, rc_.O, rc_.E, rc_.p, rc_.bal, 1.0};
        }
    }
    if (dense) {
        const HpkBandDesc& d = s.d;
        HpkDenseArgs da;
        da.rec_ent = d.rec_ent; da.rec_S = d.rec_S; da.rec_W = d.rec_W; da.tile_cnt = d.tile_cnt;
        da.tilecap = d.tilecap; da.rec_stride = d.rec_stride; da.ntiles = s.ntiles; da.TR = d.TR; da.TC = d.TC; da.J = d.J;
        da.plan = j->sa.plan; da.etab = d.etab; da.eedge = d.eedge;
        da.IR = d.IR; da.b1 = d.b1; da.b2 = d.b2; da.n = n; da.num = s.num; da.ldo = s.ldo; da.mw = mw; da.D = D;
        da.dE = L.dE.as<double2>(); da.dW = L.dW.as<uint8_t>(); da.dS = sums ? L.dS.as<double4>() : nullptr;
        hpk_launch_dense(da, c->stream);
        HIPCHK(c, hipGetLastError());
        const size_t dense_elems = s.dense_elems;
        box->denseE.resize(dense_elems * 2);
        box->denseW.resize(dense_elems);
        HIPCHK(c, hipMemcpyAsync(box->denseE.data(), L.dE.p, sizeof(double2) * dense_elems, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(box->denseW.data(), L.dW.p, dense_elems, hipMemcpyDeviceToHost, c->stream));
        if (sums) {
            box->denseS.resize(dense_elems * 4);
            HIPCHK(c, hipMemcpyAsync(box->denseS.data(), L.dS.p, sizeof(double4) * dense_elems, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        R.dense_ld = s.ldo;
        R.dense_E = box->denseE.data();
        R.dense_w = box->denseW.data();
        R.dense_sums = sums ? box->denseS.data() : nullptr;
    }
    const double t_d2h1 = now_ms();

    // ---- lambda chunks (callers.py:30) + Benjamini-Hochberg per family (callers.py:273 / 545)
    R.nsets = do_score ? nsets : 0;
    if (do_score)
        assemble_sets(plan, prm, nsets, h_chist, reinterpret_cast<const unsigned int*>(hsmall + HPK_OFF_FAM_F), h_emax, sv, box,
                      c->opt.host_prof, t_d2h1);
    const double t_end = now_ms();
    if (c->opt.host_prof) std::fprintf(stderr, "[hpk host] total host_bh=%.3f d2h=%.3f\n", t_end - t_d2h1, t_d2h1 - t_d2h0);
    R.ms_d2h = (float)(t_d2h1 - t_d2h0);
    R.ms_host_bh = (float)(t_end - t_d2h1);
    return HPK_OK;
}

// Waits for the batch, recomputes the chromosomes whose record bound was too narrow or whose survivor regions
// overflowed, and runs the host half of every chromosome.  Per-band outcomes go to the slots' status / err.
int collect_impl(hpk_ctx* c, hpk_job* j) {
    Lane& L = c->lane[j->lane];
    const HpkDevPlan& plan = L.plan_host;
    const int nb = (int)j->bands.size();
    HIPCHK(c, hipEventSynchronize(L.ev_done));           // this batch only; the next one keeps running
    L.h_rest.reset();                                    // (the lane's previous batch has its survivors in its results by now)
    std::vector<int> canon_run;          // spec_halo = 2: the chromosomes of a pass that are computed once more under their own layout
    for (int pass = 0; j->do_score; ++pass) {
        bool again = false;
        canon_run.clear();
        for (int b = 0; b < nb; ++b) {
            BandSlot& s = j->bands[b];
            if (s.status != HPK_OK) continue;
            const unsigned char* hsmall = static_cast<const unsigned char*>(L.h_head) + s.head_off;
            const int32_t fz = *reinterpret_cast<const int32_t*>(hsmall + HPK_OFF_FROZEN);
            const int32_t er = *reinterpret_cast<const int32_t*>(hsmall + HPK_OFF_ERR);
            if (j->use_class && !s.redone) {        // the bound the band ran under (hpk_band_class; gone after a full recomputation)
                const unsigned v = *reinterpret_cast<const unsigned*>(hsmall + HPK_OFF_BCLASS);
                if (v & 0x10000u) {
                    s.cls = (int)((v >> 8) & 255u); s.d.wguess = (int32_t)(v & 255u);
                    const int halo = (int)((v >> 20) & 31u);           // the band's own halo, and the tile geometry that goes with it
                    if (halo >= 4 && halo != s.d.W) {
                        const HpkGeo g = hpk_geo_of(halo, plan.W, plan.D, plan.mw, j->tr_cap);
                        s.d.W = g.W; s.d.Dg = g.Dg; s.d.TR = g.TR; s.d.TC = g.TC; s.d.J = g.J; s.d.tilecap = g.tilecap;
                        s.ntiles = ((s.n + g.TR - 1) / g.TR) * g.J;
                        s.d.ntiles = s.ntiles; s.d.chunk = (s.ntiles + 7) / 8; s.d.rec_stride = (int64_t)s.ntiles * g.tilecap;
                    }
                }
            }
            // once more from the band, on its own, under the plan's geometry and through the two-kernel path, with a record for
            // every resolved candidate (the band's second descriptor, prepared at submission)
            auto full_redo = [&](bool all_survivors) -> int {
                s.d.wguess = plan.W;
                s.d.ntiles = s.ntiles_full; s.d.chunk = (s.ntiles_full + 7) / 8;
                s.d.rec_stride = (int64_t)s.ntiles_full * j->gf.tilecap;
                s.d.W = j->gf.W; s.d.Dg = j->gf.Dg; s.d.TR = j->gf.TR; s.d.TC = j->gf.TC; s.d.J = j->gf.J; s.d.tilecap = j->gf.tilecap;
                s.ntiles = s.ntiles_full;
                s.redone = true;
                c->spec_reruns += 1;
                HIPCHK(c, hipMemsetAsync(s.d.small, 0, s.zero_bytes, c->stream));
                return launch_compute(c, j, b, 1, true, true, all_survivors);
            };
            if (er == 0 && s.d.wguess < plan.W && fz > s.d.wguess) {
                // the widening froze later than the bound the stencil wrote records up to: once more, with every resolved candidate
                int rc = full_redo(false);
                if (rc != HPK_OK) return rc;
                again = true;
                continue;
            }
            if (j->canon && er == 0 && !s.canon_done) {
                // (after the rule above: the frozen width is the chromosome's true one from here on)
                // Values that depend on the chromosome alone: the tiles' halo decides the rounding of the box sums, so the halo has to
                // be a function of the chromosome - the one of the width its own widening froze at (which no halo changes) -, not of
                // what the context scored before.  A chromosome that ran under another one is computed once more, on its own, under
                // that one, with a record for every candidate resolved up to that width.
                const int hc = std::min((int)j->gf.W, std::max(std::max((int)fz, (int)plan.wmin), 4));
                if (s.d.W != hc) {
                    const HpkGeo g = hpk_geo_of(hc, plan.W, plan.D, plan.mw, j->tr_cap);
                    s.d.wguess = hc >= j->gf.W ? plan.W : std::min((int)plan.W, std::max((int)fz, (int)plan.wmin));
                    s.d.W = g.W; s.d.Dg = g.Dg; s.d.TR = g.TR; s.d.TC = g.TC; s.d.J = g.J; s.d.tilecap = g.tilecap;
                    s.ntiles = ((s.n + g.TR - 1) / g.TR) * g.J;
                    s.d.ntiles = s.ntiles; s.d.chunk = (s.ntiles + 7) / 8; s.d.rec_stride = (int64_t)s.ntiles * g.tilecap;
                    s.d.lean_cj = 0x7fffffff;
                    s.redone = true; s.canon_done = true;
                    c->spec_reruns += 1;
                    HpkBandDesc* hd = static_cast<HpkBandDesc*>(L.h_desc) + nb + b;
                    *hd = s.d;
                    hd->k0 = 0; hd->score_wgs = j->gmax;
                    HIPCHK(c, hipMemsetAsync(s.d.small, 0, s.zero_bytes, c->stream));
                    // (launched below, together with the pass's other chromosomes of this kind: a context without history sends a
                    // whole batch through here, and twenty-three launches of one took three times the batch's own pass - round 6)
                    canon_run.push_back(b);
                    again = true;
                    continue;
                }
            }
            if (c->opt.host_prof) std::fprintf(stderr, "[hpk host] band %d specfail %u tbin %d %d %d\n", b, *reinterpret_cast<const unsigned*>(hsmall + HPK_OFF_SPECFAIL), hsmall[HPK_OFF_TBIN + 1], hsmall[HPK_OFF_TBIN + 2], hsmall[HPK_OFF_TBIN + HPK_NB + 2]);
            if (*reinterpret_cast<const unsigned*>(hsmall + HPK_OFF_SPECFAIL) != 0u && !s.rescored) {
                // the cut of some family lies above the bound the survivor records were written to: scoring and cut once
                // more, with a record for every p <= sig
                s.rescored = true;
                c->surv_rescored += 1;
                HpkBandDesc* hd = static_cast<HpkBandDesc*>(L.h_desc) + nb + b;
                *hd = s.d;
                hd->k0 = 0;             // (the band's own scoring grid: its survivor regions were sized for that many waves)
                HIPCHK(c, hipMemcpyAsync(lane_desc(L, j, b, true), hd, sizeof(HpkBandDesc), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipMemsetAsync(s.d.chunk_used, 0, sizeof(unsigned) * (size_t)(s.cap / HPK_SCH * HPK_NREG + 1), c->stream));
                HIPCHK(c, hipMemsetAsync(s.d.small + HPK_OFF_NSURV, 0, HPK_SMALL_BYTES - HPK_OFF_NSURV, c->stream));
                HIPCHK(c, hipMemsetAsync(s.d.small + s.off_cnt, 0, sizeof(unsigned) * HPK_NFAM * HPK_TIGHTEN_MAX, c->stream));
                int rc = launch_compute(c, j, b, 1, true, false, true);
                if (rc != HPK_OK) return rc;
                again = true;
                continue;
            }
            unsigned long long ns = 0;              // fullest region
            for (int rg = 0; rg < HPK_NREG; ++rg)
                ns = std::max(ns, reinterpret_cast<const unsigned long long*>(hsmall + HPK_OFF_NSURV)[rg * HPK_REG_STRIDE]);
            if ((int64_t)ns <= s.cap) continue;
            if (s.overflowed) { s.status = fail(c, HPK_ERR_NOMEM, "survivor buffer overflow"); s.err = c->err; continue; }
            // a survivor region overflowed: the scoring once more with room for everything, on buffers of its own; one
            // band at a time (they share the lane's spare buffers), its survivors beyond the inline head are fetched at once
            s.overflowed = true;
            s.cap = ((int64_t)ns * 2 + 1024 + 255) / 256 * 256;
            const size_t cu_bytes = sizeof(unsigned) * (size_t)(s.cap / HPK_SCH * HPK_NREG + 1);
            HIPCHK(c, L.survx.reserve(sizeof(HpkSurv) * (size_t)s.cap * HPK_NREG));
            HIPCHK(c, L.survx2.reserve(sizeof(HpkSurv) * (size_t)s.cap * HPK_NREG));
            HIPCHK(c, L.cux.reserve(cu_bytes));
            s.d.cap = s.cap; s.d.surv = L.survx.as<HpkSurv>(); s.d.surv2 = L.survx2.as<HpkSurv>(); s.d.chunk_used = L.cux.as<unsigned>();
            HpkBandDesc* hd = static_cast<HpkBandDesc*>(L.h_desc) + nb + b;
            *hd = s.d;
            hd->k0 = 0; hd->score_wgs = j->gmax;
            HIPCHK(c, hipMemcpyAsync(lane_desc(L, j, b, true), hd, sizeof(HpkBandDesc), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemsetAsync(L.cux.p, 0, cu_bytes, c->stream));
            HIPCHK(c, hipMemsetAsync(s.d.small + HPK_OFF_NSURV, 0, HPK_SMALL_BYTES - HPK_OFF_NSURV, c->stream));
            HIPCHK(c, hipMemsetAsync(s.d.small + s.off_cnt, 0, sizeof(unsigned) * HPK_NFAM * HPK_TIGHTEN_MAX, c->stream));
            int rc = launch_compute(c, j, b, 1, true, false, s.rescored);
            if (rc != HPK_OK) return rc;
            HIPCHK(c, hipEventSynchronize(L.ev_done));
            const unsigned long long nout = *reinterpret_cast<const unsigned long long*>(hsmall + HPK_OFF_NOUT);
            if (nout > HPK_HEAD_INLINE) {
                s.rest = static_cast<HpkSurv*>(L.h_rest.take(sizeof(HpkSurv) * (nout - HPK_HEAD_INLINE)));
                if (!s.rest) return fail(c, HPK_ERR_HIP, "pinned memory for %llu survivor records", nout - HPK_HEAD_INLINE);
                HIPCHK(c, hipMemcpyAsync(s.rest, s.d.surv2, sizeof(HpkSurv) * (nout - HPK_HEAD_INLINE), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
            s.rest_fetched = true;
            again = true;       // (checked once more: a second overflow is an error)
        }
        // the second passes of spec_halo = 2, a run of neighbouring chromosomes per set of launches (their second descriptors lie
        // side by side): which of a chromosome's tiles are lean is its own affair too - hpk_band_class once more with every band's
        // bound taken from its descriptor (its frozen width; the layout is set above) -, then the lean kernel, hpk_stencil_s and the
        // queue pass like any batch
        for (size_t i = 0; i < canon_run.size();) {
            size_t e = i + 1;
            while (e < canon_run.size() && canon_run[e] == canon_run[e - 1] + 1) ++e;
            const int b0 = canon_run[i], nbl = (int)(e - i);
            HIPCHK(c, hipMemcpyAsync(lane_desc(L, j, b0, true), static_cast<HpkBandDesc*>(L.h_desc) + nb + b0, sizeof(HpkBandDesc) * (size_t)nbl,
                                     hipMemcpyHostToDevice, c->stream));
            if (j->use_lean) {
                HpkClassArgs ca = j->cargs;
                ca.table = nullptr; ca.own = 1;
                ca.margin = 0; ca.halo = 0; ca.lean = 1;
                hpk_launch_band_class(lane_desc(L, j, b0, true), nbl, ca, c->stream);
                HIPCHK(c, hipGetLastError());
            }
            const int rc = launch_compute(c, j, b0, nbl, true, true, false, j->use_lean);
            if (rc != HPK_OK) return rc;
            i = e;
        }
        if (!again) break;
        HIPCHK(c, hipEventSynchronize(L.ev_done));
        if (pass >= 6) break;           // (each of the three reruns fires at most once per chromosome)
    }
#ifdef HPK_PHASE_CLOCK
    if (const char* path = std::getenv("HPK_CLK_DUMP")) {       // [grid][waves][8] u64, overwritten by every batch
        std::vector<unsigned long long> h((size_t)8 * HPK_NWAVES * 1024);
        HIPCHK(c, hipMemcpy(h.data(), c->tmpD.p, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(path, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
        unsigned long long s8[8];
        HIPCHK(c, hipMemcpy(s8, c->tmpD.as<unsigned long long>() + (size_t)8 * HPK_NWAVES * 1024, sizeof(s8), hipMemcpyDeviceToHost));
        if (s8[3]) std::fprintf(stderr, "hpk_score clock (s_memtime ticks): waves %llu items %llu; per wave prologue %.1f loop %.1f epilogue %.1f, longest %llu; first start to last end %llu\n",
                                s8[3], s8[4], (double)s8[0] / s8[3], (double)s8[1] / s8[3], (double)s8[2] / s8[3], s8[5], s8[7] - s8[6]);
    }
#endif
    // kernel times: events around the batch's launches, split over its chromosomes by band pixels
    float ms_st = 0.f, ms_h2d = 0.f, ms_fr = 0.f, ms_sc = 0.f, ms_ti = 0.f, ms_gap = 0.f;
    float ms = 0.f;
    if (j->time_stencil && hipEventElapsedTime(&ms, L.ev[1], L.ev[2]) == hipSuccess) ms_st = ms;
    if (j->time_stencil && !j->phases && j->do_score && hipEventElapsedTime(&ms, L.ev[2], L.ev[4]) == hipSuccess) ms_sc = ms;
    if (j->phases) {
        if (hipEventElapsedTime(&ms, L.ev[0], L.ev[1]) == hipSuccess) ms_h2d = ms;
        if (hipEventElapsedTime(&ms, L.ev[2], L.ev[3]) == hipSuccess) ms_fr = ms;
        if (hipEventElapsedTime(&ms, L.ev[3], L.ev[4]) == hipSuccess) ms_sc = ms;
        if (hipEventElapsedTime(&ms, L.ev[4], L.ev[5]) == hipSuccess) ms_ti = ms;
        if (hipEventElapsedTime(&ms, L.ev[5], L.ev[6]) == hipSuccess) ms_gap = ms;
    }
    double px_all = 0.0;
    for (const BandSlot& s : j->bands) px_all += (double)std::max<int64_t>(s.band_px, 1);
    int fz_max = -1;
    std::vector<std::pair<int, int>> cls_seen;      // (depth class, width the chromosome froze at) of the batch's chromosomes
    // The host half of a chromosome (sorting its survivors, Benjamini-Hochberg, result arrays: ~0.06 ms) is independent of
    // the others': a large batch spreads it over a few threads - the kernels of the next batch take 0.1 ms per chromosome,
    // and the host half must stay well below that.  HIP is only called from this thread: survivors beyond the inline heads
    // are fetched first; dense outputs belong to single chromosomes and stay serial.
    const int nthreads = (!j->dense && nb >= 8) ? std::min(c->opt.host_threads, nb / 4) : 1;
    if (nthreads > 1) {
        bool fetched = false;
        for (int b = 0; b < nb; ++b) {
            BandSlot& s = j->bands[b];
            if (s.status != HPK_OK || !j->do_score || s.rest_fetched) continue;
            const unsigned char* hsmall = static_cast<const unsigned char*>(L.h_head) + s.head_off;
            const size_t ns = (size_t)*reinterpret_cast<const unsigned long long*>(hsmall + HPK_OFF_NOUT);
            if (ns > HPK_HEAD_INLINE) {
                s.rest = static_cast<HpkSurv*>(L.h_rest.take(sizeof(HpkSurv) * (ns - HPK_HEAD_INLINE)));
                if (!s.rest) return fail(c, HPK_ERR_HIP, "pinned memory for %zu survivor records", ns - HPK_HEAD_INLINE);
                HIPCHK(c, hipMemcpyAsync(s.rest, s.d.surv2, sizeof(HpkSurv) * (ns - HPK_HEAD_INLINE), hipMemcpyDeviceToHost, L.up));
                s.rest_fetched = true;
                fetched = true;
            }
        }
        if (fetched) HIPCHK(c, hipStreamSynchronize(L.up));
        std::atomic<int> next{0};
        auto work = [&]() {
            for (;;) {
                const int b = next.fetch_add(1);
                if (b >= nb) break;
                BandSlot& s = j->bands[b];
                if (s.status != HPK_OK) continue;
                std::string msg;
                const int rc = finish_band(c, j, b, &msg);
                if (rc != HPK_OK) { s.status = rc; s.err = msg; }
                else s.finished = true;
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
        work();
        for (std::thread& t : pool) t.join();
    }
    for (int b = 0; b < nb; ++b) {
        BandSlot& s = j->bands[b];
        if (s.status != HPK_OK) continue;
        if (!s.finished) {
            const int rc = finish_band(c, j, b);
            if (rc != HPK_OK) { s.status = rc; s.err = c->err; continue; }
        }
        hpk_result& R = s.box->pub;
        const float share = (float)((double)std::max<int64_t>(s.band_px, 1) / px_all);
        R.ms_stencil = ms_st * share; R.ms_h2d = ms_h2d * share; R.ms_freeze = ms_fr * share; R.ms_score = ms_sc * share;
        R.ms_tighten = ms_ti * share; R.ms_gap = ms_gap * share;
        R.ms_total = (float)(now_ms() - j->t_begin);
        if (j->do_score) fz_max = std::max(fz_max, (int)R.frozen_w);
        if (j->use_class && s.cls >= 0 && s.cls < HPK_NCLASS && R.frozen_w >= 0) cls_seen.emplace_back(s.cls, (int)R.frozen_w);
    }
    if (fz_max >= 0) {          // the next stencils' record bound: the widest freeze of the last few collections
        if (c->hint_n == 0 || std::memcmp(&j->key, &c->hint_key, sizeof(j->key)) != 0) { c->hint_n = 0; c->hint_bn = 0; c->hint_key = j->key; { std::memset(c->class_w, -1, sizeof(c->class_w)); std::memset(c->class_w1, -1, sizeof(c->class_w1)); } }
        for (const auto& cf : cls_seen) { c->class_w1[cf.first] = c->class_w[cf.first]; c->class_w[cf.first] = (signed char)std::min(cf.second, 127); }
        if (j->rounds_eff <= -100) {        // ... and the bins of the families' cuts (smallest over the batch)
            uint8_t bins[HPK_NFAM];
            std::memset(bins, 255, sizeof(bins));
            bool any = false;
            for (int b = 0; b < nb; ++b) {
                const BandSlot& s = j->bands[b];
                if (s.status != HPK_OK) continue;
                const unsigned char* tb = static_cast<const unsigned char*>(L.h_head) + s.head_off + HPK_OFF_TBIN;
                for (int f = 0; f < HPK_NFAM; ++f) bins[f] = std::min(bins[f], tb[f]);
                any = true;
            }
            if (any) {
                if (c->hint_bn == 4) { std::memmove(c->hint_bin[0], c->hint_bin[1], 3 * HPK_NFAM); c->hint_bn = 3; }
                std::memcpy(c->hint_bin[c->hint_bn++], bins, HPK_NFAM);
            }
        }
        if (c->hint_n < 4) c->hint_w[c->hint_n++] = fz_max;
        else { for (int i = 0; i < 3; ++i) c->hint_w[i] = c->hint_w[i + 1]; c->hint_w[3] = fz_max; }
    }
    return HPK_OK;
}

// ---------------------------------------------------------------------------- back-end #0 (device -1): submit = remember, collect = compute
int cpu_submit(hpk_ctx* c, hpk_job* j, const hpk_band* bands, int nb, const hpk_params* prm) {
    if (prm->flags & (HPK_FLAG_DENSE_E | HPK_FLAG_DENSE_SUMS))
        return fail(c, HPK_ERR_INVALID, "the dense debug outputs (HPK_FLAG_DENSE_*) are the device path's");
    for (int b = 0; b < nb; ++b)
        if (bands[b].on_device) return fail(c, HPK_ERR_INVALID, "back-end #0 takes host arrays (on_device = 0)");
    j->cpu_plan = new HpkDevPlan();
    char msg[256];
    const int rc = hpk_build_plan(prm, j->cpu_plan, msg);
    if (rc != HPK_OK) return fail(c, rc, "%s", msg);
    if (j->cpu_plan->D < j->cpu_plan->mw) return fail(c, HPK_ERR_INVALID, "maxapart / res (%d) is below min(ww) (%d)", j->cpu_plan->D, j->cpu_plan->mw);
    j->prm = *prm;
    j->nsets = (j->cpu_plan->mode == HPK_MODE_BHFDR) ? 1 : 2 * j->cpu_plan->npairs;
    j->do_score = (prm->flags & HPK_FLAG_NO_SCORE) == 0;
    j->bands.resize((size_t)nb);
    for (int b = 0; b < nb; ++b) { j->bands[b].in = bands[b]; j->bands[b].n = bands[b].n; j->bands[b].num = bands[b].num; }
    return HPK_OK;
}

int cpu_collect(hpk_ctx* c, hpk_job* j) {
    const HpkDevPlan& plan = *j->cpu_plan;
    if (c->tables_dirty || !c->cpu_tabs.built) {
        c->cpu_tabs.bounds = c->h_bounds;
        c->cpu_tabs.sfe = c->h_sfe;
        fill_offsets(c->h_bounds, c->cpu_tabs.off);
        hpk_cpu_build_tables(c->cpu_tabs, c->cpu_threads);
        c->tables_dirty = false;
    }
    const int nb = (int)j->bands.size();
    for (int b = 0; b < nb; ++b) {
        BandSlot& s = j->bands[b];
        HpkCpuOut o;
        std::string msg;
        const double t0 = now_ms();
        int rc = hpk_cpu_band(s.in, j->prm, plan, c->cpu_tabs, c->cpu_threads, o, msg);
        if (rc != HPK_OK) { s.status = rc; s.err = msg; continue; }
        ResultBox* box = new ResultBox();
        std::memset(&box->pub, 0, sizeof(hpk_result));
        s.box = box;
        hpk_result& R = box->pub;
        box->gap.resize((size_t)s.n);
        for (int r = 0; r < s.n; ++r) box->gap[r] = o.rowlive[r] ? 0 : 1;
        R.gap = box->gap.data();
        R.band_px = o.band_px;
        R.stencil_kernel = 0;               // (no device kernel ran)
        R.batch_bands = nb;
        R.record_bound = 255; R.halo_w = plan.W;
        R.nsteps = plan.nsteps;
        for (int t = 0; t < plan.nsteps; ++t) {
            R.step_pi[t] = plan.steps[t].pi; R.step_wi[t] = plan.steps[t].wi;
            R.step_executed[t] = o.exec[t];
            R.step_resolved[t] = (o.exec[t] || !j->do_score) ? (int64_t)o.hist[t] : 0;
        }
        R.frozen_w = o.frozen;
        R.nslots = plan.nslots;
        for (int q = 0; q < plan.nslots; ++q) R.slot_pi[q] = plan.slot_pi[q];
        R.ncand = (int64_t)o.hist[HPK_HIST_NCAND];
        R.nsurv_sig = 0;
        for (uint32_t v : o.fam_f) R.nsurv_sig += v;
        R.nsurv_cut = (int64_t)o.surv.size();
        if (o.err != 0) {
            const HpkDevStep& st = plan.steps[o.err - 1];
            char m2[256];
            std::snprintf(m2, sizeof(m2), "step (%d,%d) entered with no unresolved candidate (of %lld); the reference raises here "
                          "(hicpeaks/callers.py:203-208)", st.pi, st.wi, (long long)R.ncand);
            s.status = HPK_ERR_EMPTY_STEP; s.err = m2;
            delete s.box; s.box = nullptr;
            continue;
        }
        R.nsets = j->do_score ? j->nsets : 0;
        if (j->do_score) {
            std::vector<Surv> sv(o.surv.size());
            for (size_t i = 0; i < sv.size(); ++i) {
                const HpkSurv& rc_ = o.surv[i];
                sv[i] = Surv{rc_.x, rc_.y, rc_.set, rc_.chunk, rc_.flag, 0, rc_.O, rc_.E, rc_.p, rc_.bal, 1.0};
            }
            const double t1 = now_ms();
            assemble_sets(plan, &j->prm, j->nsets, o.fam_m.data(), o.fam_f.data(), o.emax.data(), sv, box, c->opt.host_prof, t1);
            R.ms_host_bh = (float)(now_ms() - t1);
        }
        R.ms_stencil = (float)o.ms_sums; R.ms_score = (float)o.ms_score;
        R.ms_total = (float)(now_ms() - t0);
    }
    return HPK_OK;
}

int acquire_lane(hpk_ctx* c) {
    for (int l = 0; l < HPK_LANES; ++l) if (!c->lane[l].busy) return l;
    return -1;
}

}  // namespace

extern "C" {

int hpk_pipeline_depth(void) { return HPK_LANES; }

int hpk_submit_batch(hpk_ctx* c, const hpk_band* bands, int32_t nbands, const hpk_params* prm, hpk_job** job) {
    if (!c) return HPK_ERR_INVALID;
    if (!job || !prm || !bands) return fail(c, HPK_ERR_INVALID, "bands / params / job is NULL");
    *job = nullptr;
    if (nbands < 1 || nbands > HPK_MAX_BATCH) return fail(c, HPK_ERR_INVALID, "a batch holds 1..%d chromosomes", HPK_MAX_BATCH);
    for (int b = 0; b < nbands; ++b) {
        const int rc = check_band(c, bands + b);
        if (rc != HPK_OK) return rc;
    }
    if (c->device < 0) {        // back-end #0
        hpk_job* j = new hpk_job();
        j->ctx = c; j->lane = -1; j->t_begin = now_ms();
        const int rc = cpu_submit(c, j, bands, nbands, prm);
        if (rc != HPK_OK) { delete j; return rc; }
        *job = j;
        return HPK_OK;
    }
    const int lane = acquire_lane(c);
    if (lane < 0) return fail(c, HPK_ERR_BUSY, "all %d lanes hold a batch in flight: collect one first", HPK_LANES);
    (void)hipSetDevice(c->device);
    hpk_job* j = new hpk_job();
    j->ctx = c; j->lane = lane; j->t_begin = now_ms();
    const int rc = submit_impl(c, j, bands, nbands, prm);
    if (rc != HPK_OK) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->lane[lane].up);
        delete j;
        return rc;
    }
    c->lane[lane].busy = true;
    *job = j;
    return HPK_OK;
}

int hpk_collect_batch(hpk_ctx* c, hpk_job* job, hpk_result** outs, int32_t* status, char* errmsg, int32_t errmsg_len) {
    if (!c || !job || job->ctx != c) return c ? fail(c, HPK_ERR_INVALID, "job does not belong to this context") : HPK_ERR_INVALID;
    const int nb = (int)job->bands.size();
    if (outs) for (int b = 0; b < nb; ++b) outs[b] = nullptr;
    int rc;
    if (c->device < 0) rc = cpu_collect(c, job);
    else {
        (void)hipSetDevice(c->device);
        rc = collect_impl(c, job);
        if (rc != HPK_OK) (void)hipStreamSynchronize(c->stream);
        c->lane[job->lane].busy = false;
    }
    if (rc == HPK_OK) {
        for (int b = 0; b < nb; ++b) {
            BandSlot& s = job->bands[b];
            if (status) status[b] = s.status;
            if (errmsg && errmsg_len > 0) {
                std::strncpy(errmsg + (size_t)b * errmsg_len, s.err.c_str(), errmsg_len - 1);
                errmsg[(size_t)b * errmsg_len + errmsg_len - 1] = 0;
            }
            if (s.status == HPK_OK && outs) { outs[b] = &s.box->pub; s.box = nullptr; }
            if (s.status != HPK_OK) c->err = s.err;
        }
    }
    delete job;
    return rc;
}

int hpk_submit_band(hpk_ctx* c, const hpk_band* band, const hpk_params* prm, hpk_job** job) {
    return hpk_submit_batch(c, band, 1, prm, job);
}

int hpk_collect(hpk_ctx* c, hpk_job* job, hpk_result** out) {
    if (!c || !job || job->ctx != c) return c ? fail(c, HPK_ERR_INVALID, "job does not belong to this context") : HPK_ERR_INVALID;
    if (job->bands.size() != 1) {
        (void)hpk_collect_batch(c, job, nullptr, nullptr, nullptr, 0);
        return fail(c, HPK_ERR_INVALID, "hpk_collect takes single-chromosome jobs; use hpk_collect_batch");
    }
    if (out) *out = nullptr;
    hpk_result* res = nullptr;
    int32_t st = HPK_OK;
    const int rc = hpk_collect_batch(c, job, &res, &st, nullptr, 0);
    if (rc != HPK_OK) return rc;
    if (st != HPK_OK) return st;
    if (out) *out = res; else hpk_result_free(res);
    return HPK_OK;
}

int hpk_score_band(hpk_ctx* c, const hpk_band* band, const hpk_params* prm, hpk_result** out) {
    if (!c) return HPK_ERR_INVALID;
    if (!out || !prm) return fail(c, HPK_ERR_INVALID, "params / out is NULL");
    *out = nullptr;
    hpk_job* job = nullptr;
    int rc = hpk_submit_band(c, band, prm, &job);
    if (rc != HPK_OK) return rc;
    return hpk_collect(c, job, out);
}

namespace {
// pixels [t0, t1) into the band; returns the pixels stored, -1 on a bin outside the chromosome; *sorted is cleared when the rows
// (smaller bin of a pixel) do not come in non-decreasing order
int64_t band_scatter(const int64_t* bin1, const int64_t* bin2, const int32_t* ci, const double* cd, int64_t t0, int64_t t1,
                     int32_t n, int32_t num, int64_t ld, float* raw, bool* sorted) {
    int64_t stored = 0, last = -1;
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t a = bin1[t] < bin2[t] ? bin1[t] : bin2[t], b = bin1[t] < bin2[t] ? bin2[t] : bin1[t];
        if (a < 0 || b >= n) return -1;                     // a bin outside the chromosome: the caller's offsets are off
        if (a < last && sorted) *sorted = false;
        last = a;
        const int64_t k = b - a;
        if (k >= num) continue;                             // beyond the band
        raw[a * ld + k] += cd ? (float)cd[t] : (float)ci[t];
        ++stored;
    }
    return stored;
}
}  // namespace

// One O(nnz) scatter.  A large pixel table whose rows come in order (cooler's pixel table is sorted by bin1, bin2) is cut at row
// changes into one stretch per thread: the stretches write disjoint band rows, repeats of a cell stay in one stretch.  Whether the
// table is in row order (and every bin inside the chromosome) is found out first, by the same threads reading only; a table
// that is not goes to one thread (repeats may then meet anywhere).  HPK_COO_MIN_PER_THREAD: pixels per thread below which the
// table is not split (default 4 Mi; tests).
int64_t hpk_band_from_coo(const int64_t* bin1, const int64_t* bin2, const void* count, int32_t count_f64, int64_t nnz,
                          int32_t n, int32_t num, int64_t ld, float* raw) {
    if (!bin1 || !bin2 || !count || !raw || nnz < 0 || n <= 0 || num <= 0 || ld < num) return HPK_ERR_INVALID;
    const int32_t* ci = count_f64 ? nullptr : static_cast<const int32_t*>(count);
    const double* cd = count_f64 ? static_cast<const double*>(count) : nullptr;
    int64_t per_thread = 4 << 20;
    if (const char* e = std::getenv("HPK_COO_MIN_PER_THREAD")) per_thread = std::max<int64_t>(1, std::atoll(e));
    const int nt = (int)std::min<int64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), nnz / per_thread);
    if (nt > 1) {
        std::vector<int64_t> cut(nt + 1);
        cut[0] = 0; cut[nt] = nnz;
        auto row = [&](int64_t t) { return bin1[t] < bin2[t] ? bin1[t] : bin2[t]; };
        for (int i = 1; i < nt; ++i) {
            int64_t t = std::max(nnz * i / nt, cut[i - 1]);
            while (t < nnz && t > 0 && row(t) == row(t - 1)) ++t;          // to the next row change
            cut[i] = t;
        }
        // read-only: every stretch in row order, starting at or behind the row the stretch before ended on, every bin inside
        std::vector<char> state(nt, 0);             // 0 in order, 1 not in row order, 2 a bin outside the chromosome
        auto check = [&](int i) {
            int64_t last = cut[i] > 0 ? row(cut[i] - 1) : -1;
            for (int64_t t = cut[i]; t < cut[i + 1]; ++t) {
                const int64_t a = bin1[t] < bin2[t] ? bin1[t] : bin2[t], b = bin1[t] < bin2[t] ? bin2[t] : bin1[t];
                if (a < 0 || b >= n) { state[i] = 2; return; }
                if (a < last) state[i] = 1;
                last = a;
            }
        };
        {
            std::vector<std::thread> pool;
            for (int i = 1; i < nt; ++i) pool.emplace_back(check, i);
            check(0);
            for (std::thread& t : pool) t.join();
        }
        bool sorted = true;
        for (int i = 0; i < nt; ++i) { if (state[i] == 2) return HPK_ERR_INVALID; sorted = sorted && state[i] == 0; }
        if (sorted) {
            std::vector<int64_t> got(nt, 0);
            std::vector<std::thread> pool;
            auto work = [&](int i) { got[i] = band_scatter(bin1, bin2, ci, cd, cut[i], cut[i + 1], n, num, ld, raw, nullptr); };
            for (int i = 1; i < nt; ++i) pool.emplace_back(work, i);
            work(0);
            for (std::thread& t : pool) t.join();
            int64_t stored = 0;
            for (int i = 0; i < nt; ++i) { if (got[i] < 0) return HPK_ERR_INVALID; stored += got[i]; }
            return stored;
        }
    }
    const int64_t stored = band_scatter(bin1, bin2, ci, cd, 0, nnz, n, num, ld, raw, nullptr);
    return stored < 0 ? HPK_ERR_INVALID : stored;
}

struct hpk_devband { float* raw = nullptr; double* weight = nullptr; double* bias = nullptr; size_t raw_bytes = 0, w_bytes = 0; };

namespace {
// Device memory of the band builder: freed blocks are kept and handed out again (hipFree synchronises the device, which
// would stall the batch in flight; a genome's chromosomes come largest first, so a freed band fits the next ones).
hipError_t pool_alloc(hpk_ctx* c, size_t bytes, void** p, size_t* got) {
    size_t best = (size_t)-1;
    for (size_t i = 0; i < c->pool.size(); ++i)
        if (c->pool[i].first >= bytes && (best == (size_t)-1 || c->pool[i].first < c->pool[best].first)) best = i;
    if (best != (size_t)-1 && c->pool[best].first <= 2 * bytes + (1 << 20)) {
        *p = c->pool[best].second; *got = c->pool[best].first;
        c->pool.erase(c->pool.begin() + best);
        return hipSuccess;
    }
    *got = bytes;
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && !c->pool.empty()) {     // the blocks kept for reuse go back to the device, then once more
        (void)hipGetLastError();
        for (auto& b : c->pool) (void)hipFree(b.second);
        c->pool.clear();
        e = hipMalloc(p, bytes);
    }
    return e;
}
void pool_free(hpk_ctx* c, void* p, size_t bytes) {
    if (!p) return;
    if (c && c->pool.size() < 64) c->pool.emplace_back(bytes, p);
    else (void)hipFree(p);
}
}  // namespace

int64_t hpk_devband_create(hpk_ctx* c, const int64_t* bin1, const int64_t* bin2, const void* count, int32_t count_f64, int64_t nnz,
                           int32_t n, int32_t num, const double* weight, const double* bias, hpk_devband** out, hpk_band* band) {
    if (!c) return HPK_ERR_INVALID;
    if (c->device < 0) return fail(c, HPK_ERR_INVALID, "hpk_devband_create builds a band in device memory: back-end #0 takes host bands (hpk_band_from_coo)");
    if (!out || !band || !weight || nnz < 0 || n <= 0 || num <= 0 || (nnz > 0 && (!bin1 || !bin2 || !count)))
        return fail(c, HPK_ERR_INVALID, "hpk_devband_create: bad arguments");
    *out = nullptr;
    (void)hipSetDevice(c->device);
    if (!c->aux && hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess)
        return fail(c, HPK_ERR_HIP, "hpk_devband_create: stream creation failed");
    const int64_t ld = ((int64_t)num + 63) / 64 * 64;
    hpk_devband* b = new hpk_devband();
    auto run = [&]() -> int64_t {
        // (on a stream of its own: the batches in flight on the compute stream are not waited for)
        HIPCHK(c, pool_alloc(c, sizeof(float) * (size_t)n * (size_t)ld, reinterpret_cast<void**>(&b->raw), &b->raw_bytes));
        HIPCHK(c, pool_alloc(c, sizeof(double) * (size_t)n * (bias ? 2 : 1), reinterpret_cast<void**>(&b->weight), &b->w_bytes));
        if (bias) b->bias = b->weight + n;
        HIPCHK(c, hipMemsetAsync(b->raw, 0, sizeof(float) * (size_t)n * (size_t)ld, c->aux));
        HIPCHK(c, hipMemcpyAsync(b->weight, weight, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->aux));
        if (bias) HIPCHK(c, hipMemcpyAsync(b->bias, bias, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->aux));
        unsigned long long info[2] = {0ull, 0ull};
        if (nnz > 0) {
            const size_t cb = ((count_f64 ? 8 : 4) * (size_t)nnz + 7) / 8 * 8;
            HIPCHK(c, c->cooA.reserve(8 * (size_t)nnz));
            HIPCHK(c, c->cooB.reserve(8 * (size_t)nnz));
            HIPCHK(c, c->cooC.reserve(cb + 16));
            unsigned long long* d_info = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->cooC.p) + cb);
            HIPCHK(c, hipMemcpyAsync(c->cooA.p, bin1, 8 * (size_t)nnz, hipMemcpyHostToDevice, c->aux));
            HIPCHK(c, hipMemcpyAsync(c->cooB.p, bin2, 8 * (size_t)nnz, hipMemcpyHostToDevice, c->aux));
            HIPCHK(c, hipMemcpyAsync(c->cooC.p, count, (count_f64 ? 8 : 4) * (size_t)nnz, hipMemcpyHostToDevice, c->aux));
            HIPCHK(c, hipMemsetAsync(d_info, 0, 16, c->aux));
            hpk_launch_coo_scatter(c->cooA.as<int64_t>(), c->cooB.as<int64_t>(), c->cooC.p, count_f64 ? 1 : 0, nnz, n, num, ld, b->raw,
                                   d_info, c->aux);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(info, d_info, 16, hipMemcpyDeviceToHost, c->aux));
        }
        HIPCHK(c, hipStreamSynchronize(c->aux));
        if (info[1]) return fail(c, HPK_ERR_INVALID, "hpk_devband_create: a bin outside [0, %d) (are the bins relative to the chromosome's first bin?)", n);
        return (int64_t)info[0];
    };
    const int64_t rc = run();
    if (rc < 0) { hpk_devband_free(c, b); return rc; }
    std::memset(band, 0, sizeof(*band));
    band->n = n; band->num = num; band->ld = ld;
    band->raw = b->raw; band->weight = b->weight;
    band->bias1 = b->bias; band->bias2 = b->bias;
    band->on_device = 1;
    c->live_bands.push_back(b);
    *out = b;
    return rc;
}

void hpk_devband_free(hpk_ctx* c, hpk_devband* b) {
    if (!b) return;
    if (c) {
        if (c->device >= 0) (void)hipSetDevice(c->device);
        c->live_bands.erase(std::remove(c->live_bands.begin(), c->live_bands.end(), b), c->live_bands.end());
    }
    pool_free(c, b->raw, b->raw_bytes);
    pool_free(c, b->weight, b->w_bytes);
    delete b;
}

int hpk_probe_sums(hpk_ctx* c, const hpk_band* band, const hpk_params* prm, const int32_t* rows, const int32_t* cols,
                   int64_t count, double* out) {
    if (c) { HPK_NEED_TEST_KERNELS(c); }
    if (!c) return HPK_ERR_INVALID;
    if (c->device < 0) return fail(c, HPK_ERR_INVALID, "a check kernel of the device path: not on back-end #0");
    if (!prm || !rows || !cols || !out || count < 0) return fail(c, HPK_ERR_INVALID, "bad arguments");
    hpk_params p2 = *prm;
    p2.flags = HPK_FLAG_NO_SCORE;               // stencil + freeze only; the records stay in the lane's workspaces
    hpk_job* job = nullptr;
    int rc = hpk_submit_band(c, band, &p2, &job);
    if (rc != HPK_OK) return rc;
    Lane& L = c->lane[job->lane];
    const BandSlot& s = job->bands[0];
    const HpkBandDesc& d = s.d;
    const HpkDevPlan& plan = L.plan_host;
    auto run = [&]() -> int {
        if (count == 0) return HPK_OK;
        HIPCHK(c, c->tmpA.reserve(4 * (size_t)count));
        HIPCHK(c, c->tmpB.reserve(4 * (size_t)count));
        HIPCHK(c, c->tmpC.reserve(8 * 5 * (size_t)count * plan.nslots));
        HIPCHK(c, hipMemcpyAsync(c->tmpA.p, rows, 4 * (size_t)count, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->tmpB.p, cols, 4 * (size_t)count, hipMemcpyHostToDevice, c->stream));
        HpkDenseArgs da;
        std::memset(&da, 0, sizeof(da));
        da.rec_ent = d.rec_ent; da.rec_S = d.rec_S; da.rec_W = d.rec_W; da.tile_cnt = d.tile_cnt;
        da.tilecap = d.tilecap; da.rec_stride = d.rec_stride; da.ntiles = s.ntiles; da.TR = d.TR; da.TC = d.TC; da.J = d.J;
        da.plan = job->sa.plan; da.etab = d.etab; da.eedge = d.eedge;
        da.IR = d.IR; da.b1 = d.b1; da.b2 = d.b2; da.n = s.n; da.num = s.num; da.ldo = s.ldo;
        da.mw = plan.mw; da.D = plan.D;
        hpk_launch_probe(da, c->tmpA.as<int32_t>(), c->tmpB.as<int32_t>(), count, c->tmpC.as<double>(), c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(out, c->tmpC.p, 8 * 5 * (size_t)count * plan.nslots, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return HPK_OK;
    };
    rc = run();
    const std::string keep = c->err;
    (void)hpk_collect(c, job, nullptr);          // gives the lane back (an "empty step" verdict is not this call's business)
    if (rc != HPK_OK) c->err = keep;
    return rc;
}

int hpk_bruteforce_sums(hpk_ctx* c, const hpk_band* band, const hpk_params* prm, int32_t step, const int32_t* rows,
                        const int32_t* cols, int64_t count, double* out) {
    if (c) { HPK_NEED_TEST_KERNELS(c); }
    if (!c) return HPK_ERR_INVALID;
    if (c->device < 0) return fail(c, HPK_ERR_INVALID, "a check kernel of the device path: not on back-end #0");
    if (!prm || !rows || !cols || !out || count < 0) return fail(c, HPK_ERR_INVALID, "bad arguments");
    int rc = check_band(c, band);
    if (rc != HPK_OK) return rc;
    if (count == 0) return HPK_OK;
    (void)hipSetDevice(c->device);
    HpkDevPlan plan;
    char msg[256];
    rc = hpk_build_plan(prm, &plan, msg);
    if (rc != HPK_OK) return fail(c, rc, "%s", msg);
    if (step < 0 || step >= plan.nsteps) return fail(c, HPK_ERR_INVALID, "step out of range");
    if (c->lane[0].busy) return fail(c, HPK_ERR_BUSY, "lane 0 holds a chromosome in flight");
    Lane& L = c->lane[0];
    // inputs as the pipeline stages them (uploads and the derivation of IR / biases on the lane's side stream)
    const size_t n = (size_t)band->n, num = (size_t)band->num, ld = (size_t)band->ld;
    const bool derive = !band->IR;
    Staged in;
    HpkBandDesc d;
    std::memset(&d, 0, sizeof(d));
    if (band->on_device) {
        in.raw = band->raw; in.bal = band->balanced; in.weight = band->weight;
        in.IR = const_cast<double*>(band->IR);
    } else {
        HIPCHK(c, L.raw.reserve(sizeof(float) * n * ld));
        HIPCHK(c, hipMemcpyAsync(L.raw.p, band->raw, sizeof(float) * n * ld, hipMemcpyHostToDevice, L.up));
        in.raw = L.raw.as<float>();
        if (band->balanced) {
            HIPCHK(c, L.bal.reserve(sizeof(double) * n * ld));
            HIPCHK(c, hipMemcpyAsync(L.bal.p, band->balanced, sizeof(double) * n * ld, hipMemcpyHostToDevice, L.up));
            in.bal = L.bal.as<double>();
        }
        if (band->weight) {
            HIPCHK(c, L.weight.reserve(sizeof(double) * n));
            HIPCHK(c, hipMemcpyAsync(L.weight.p, band->weight, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
            in.weight = L.weight.as<double>();
        }
        if (!derive) {
            HIPCHK(c, L.IR.reserve(sizeof(double) * num));
            HIPCHK(c, hipMemcpyAsync(L.IR.p, band->IR, sizeof(double) * num, hipMemcpyHostToDevice, L.up));
            in.IR = L.IR.as<double>();
        }
    }
    if (derive) {
        const size_t nparts = (n + 31) / 32;
        HIPCHK(c, L.IR.reserve(sizeof(double) * num));
        HIPCHK(c, L.b1.reserve(sizeof(double) * n));
        HIPCHK(c, L.psum.reserve(sizeof(double) * nparts * num));
        HIPCHK(c, L.pnan.reserve(sizeof(unsigned) * nparts * num));
        HIPCHK(c, L.desc.reserve(sizeof(HpkBandDesc)));
        d.raw = in.raw; d.weight = in.weight; d.n = band->n; d.num = band->num; d.ld = band->ld; d.derive = 1;
        d.IR = L.IR.as<double>(); d.b1 = L.b1.as<double>(); d.b2 = d.b1; d.psum = L.psum.as<double>(); d.pnan = L.pnan.as<unsigned>();
        HIPCHK(c, hipMemcpyAsync(L.desc.p, &d, sizeof(d), hipMemcpyHostToDevice, L.up));
        HIPCHK(c, hipStreamSynchronize(L.up));              // (`d` lives on this stack frame)
        hpk_launch_prep(L.desc.as<HpkBandDesc>(), 1, band->n, band->num, plan.mw, L.up);
        HIPCHK(c, hipGetLastError());
        in.IR = L.IR.as<double>();
    }
    // the brute-force kernel runs on the compute stream: it has to wait for what the side stream staged
    HIPCHK(c, hipEventRecord(L.ev_up, L.up));
    HIPCHK(c, hipStreamWaitEvent(c->stream, L.ev_up, 0));
    L.plan_valid = false;
    HIPCHK(c, L.plan.reserve(sizeof(HpkDevPlan)));
    HIPCHK(c, hipMemcpyAsync(L.plan.p, &plan, sizeof(plan), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, c->tmpA.reserve(4 * (size_t)count));
    HIPCHK(c, c->tmpB.reserve(4 * (size_t)count));
    HIPCHK(c, c->tmpC.reserve(8 * 5 * (size_t)count));
    HIPCHK(c, hipMemcpyAsync(c->tmpA.p, rows, 4 * (size_t)count, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->tmpB.p, cols, 4 * (size_t)count, hipMemcpyHostToDevice, c->stream));
    HpkBruteArgs a;
    a.raw = in.raw; a.bal = in.bal; a.weight = in.weight; a.IR = in.IR; a.plan = L.plan.as<HpkDevPlan>();
    a.n = band->n; a.num = band->num; a.ld = band->ld; a.step = step;
    a.rows = c->tmpA.as<int32_t>(); a.cols = c->tmpB.as<int32_t>(); a.count = count; a.out = c->tmpC.as<double>();
    hpk_launch_brute(a, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->tmpC.p, 8 * 5 * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HPK_OK;
}

}  // extern "C"
