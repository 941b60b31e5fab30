// libhpk.so entry points (include/hpk.h): context, staging, the per-chromosome pipeline and the
// Benjamini-Hochberg step on the compacted survivors.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/hpk.h"
#include "hpk_kernels.h"
#include "hpk_plan.h"

namespace {

std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// One lane = everything a chromosome in flight owns: upload stream, events, device workspaces, the pinned landing
// area of its result head and the cached device copy of its widening plan.  Two lanes let the host half of
// chromosome i (Benjamini-Hochberg, result assembly, the caller's Python) and the upload of chromosome i + 1 overlap
// the kernels.  The stencil, the scoring and the cut of all lanes run in submission order on the context's one compute
// stream: the stencil fills the whole chip (one 160 KiB-LDS workgroup per CU), so running two chromosomes' big kernels
// side by side would only time-slice them.  What a chromosome needs *before* its stencil - uploads, IR / biases, the
// expected tables and the zero-fill of its counters - runs on the lane's side stream, beside the scoring / cut kernels
// of the chromosome before.  (The cut and the copy of the result head were tried on the side stream too: they then
// wait behind the next chromosome's stencil, the host collects a result later and submits the chromosome after next
// later - 0.203 -> 0.228 ms per chromosome.)
#define HPK_LANES 2
struct Lane {
    hipStream_t up = nullptr;           // uploads of host inputs run beside the kernels of the chromosome before
    hipEvent_t ev[8];                   // phase marks on the compute stream
    hipEvent_t ev_up = nullptr, ev_done = nullptr;
    int nev = 0;
    bool busy = false;
    // workspaces (grow only)
    DevBuf raw, bal, weight, IR, b1, b2, plan, etab, eedge, recE, recS, recW, dE, dW, dS, small, surv, surv2, chunkused, psum, pnan, units;
    void* h_head = nullptr;             // pinned: counters | row flags | first survivors
    size_t h_head_cap = 0;
    // the device copy of the widening plan is reused while the parameters do not change
    hpk_params plan_key;
    bool plan_valid = false;
    HpkDevPlan plan_host;
    void release() {
        DevBuf* all[] = {&raw, &bal, &weight, &IR, &b1, &b2, &plan, &etab, &eedge, &recE, &recS, &recW, &dE, &dW, &dS, &small,
                         &surv, &surv2, &chunkused, &psum, &pnan, &units};
        for (DevBuf* b : all) b->release();
        if (h_head) { (void)hipHostFree(h_head); h_head = nullptr; h_head_cap = 0; }
        for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
        nev = 0;
        if (ev_up) { (void)hipEventDestroy(ev_up); ev_up = nullptr; }
        if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
        if (up) { (void)hipStreamDestroy(up); up = nullptr; }
    }
};

struct hpk_ctx {
    int device = -1;
    hipStream_t stream = nullptr;       // compute stream: every kernel and the result downloads
    std::string err;
    char name[128] = {0};
    int cus = 0;
    size_t hbm = 0;
    // constant tables
    std::vector<double> h_bounds;
    std::vector<int32_t> h_off;
    std::vector<double> h_sfe;
    DevBuf d_bounds, d_off, d_sfe, d_ptab;
    Lane lane[HPK_LANES];
    DevBuf tmpA, tmpB, tmpC, tmpD;
    // The width the widening froze at (freeze_body) in the last chromosome collected with these parameters: the next
    // stencil writes records up to it only (HpkStencilArgs::wguess); a chromosome that freezes later is redone in full.
    hpk_params hint_key;
    int hint_w = -1;
    long long spec_reruns = 0;
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

int upload_tables(hpk_ctx* c) {
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
    return HPK_OK;
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
    int rc = upload_tables(c);
    if (rc != HPK_OK) { g_create_error = c->err; hpk_destroy(c); return rc; }
    *out = c;
    return HPK_OK;
}

void hpk_destroy(hpk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    DevBuf* all[] = {&c->d_bounds, &c->d_off, &c->d_sfe, &c->d_ptab, &c->tmpA, &c->tmpB, &c->tmpC, &c->tmpD};
    for (DevBuf* b : all) b->release();
    for (int l = 0; l < HPK_LANES; ++l) c->lane[l].release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int hpk_set_chunk_bounds(hpk_ctx* c, const double* bounds, int32_t count) {
    if (!c || !bounds || count != HPK_NB) return fail(c, HPK_ERR_INVALID, "need %d bounds", HPK_NB);
    (void)hipSetDevice(c->device);
    for (int i = 0; i < HPK_NB; ++i) {
        if (!(bounds[i] > 0.0) || (i && !(bounds[i] > bounds[i - 1]))) return fail(c, HPK_ERR_INVALID, "bounds must increase");
        c->h_bounds[i] = bounds[i];
    }
    return upload_tables(c);
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
    if (!c || !k || !lam || !out || count < 0) return fail(c, HPK_ERR_INVALID, "bad arguments");
    if (count == 0) return HPK_OK;
    (void)hipSetDevice(c->device);
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

// ---------------------------------------------------------------------------- staging shared by score / brute
namespace {

struct Staged {
    const float* raw = nullptr;
    const double* bal = nullptr;
    const double* weight = nullptr;
    const double* IR = nullptr;
    const double* b1 = nullptr;
    const double* b2 = nullptr;
};

int stage_inputs(hpk_ctx* c, Lane& L, const hpk_band* band, int mw, Staged* s) {
    const size_t n = (size_t)band->n, num = (size_t)band->num, ld = (size_t)band->ld;
    const bool derive = !band->IR;          // IR / biases from raw + weight on the device (scripts/pyHICCUPS:149-166)
    if (band->on_device) {
        s->raw = band->raw; s->bal = band->balanced; s->weight = band->weight; s->IR = band->IR; s->b1 = band->bias1; s->b2 = band->bias2;
    } else {
        HIPCHK(c, L.raw.reserve(sizeof(float) * n * ld));
        HIPCHK(c, hipMemcpyAsync(L.raw.p, band->raw, sizeof(float) * n * ld, hipMemcpyHostToDevice, L.up));
        s->raw = L.raw.as<float>();
        if (band->balanced) {
            HIPCHK(c, L.bal.reserve(sizeof(double) * n * ld));
            HIPCHK(c, hipMemcpyAsync(L.bal.p, band->balanced, sizeof(double) * n * ld, hipMemcpyHostToDevice, L.up));
            s->bal = L.bal.as<double>();
        }
        if (band->weight) {
            HIPCHK(c, L.weight.reserve(sizeof(double) * n));
            HIPCHK(c, hipMemcpyAsync(L.weight.p, band->weight, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
            s->weight = L.weight.as<double>();
        }
        if (!derive) {
            HIPCHK(c, L.IR.reserve(sizeof(double) * num));
            HIPCHK(c, hipMemcpyAsync(L.IR.p, band->IR, sizeof(double) * num, hipMemcpyHostToDevice, L.up));
            s->IR = L.IR.as<double>();
            HIPCHK(c, L.b1.reserve(sizeof(double) * n));
            HIPCHK(c, hipMemcpyAsync(L.b1.p, band->bias1, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
            s->b1 = L.b1.as<double>();
            if (band->bias2 == band->bias1) s->b2 = s->b1;
            else {
                HIPCHK(c, L.b2.reserve(sizeof(double) * n));
                HIPCHK(c, hipMemcpyAsync(L.b2.p, band->bias2, sizeof(double) * n, hipMemcpyHostToDevice, L.up));
                s->b2 = L.b2.as<double>();
            }
        }
    }
    if (derive) {
        // The derivation runs on the lane's side stream too: it only needs the inputs, so for the chromosome submitted
        // one ahead it executes beside the scoring / cut kernels of the chromosome before (those leave registers and
        // LDS free; the stencil does not).
        const size_t nparts = (n + 31) / 32;        // HPK_IR_ROWS rows per partial (hpk_launch_prep)
        HIPCHK(c, L.IR.reserve(sizeof(double) * num));
        HIPCHK(c, L.b1.reserve(sizeof(double) * n));
        HIPCHK(c, L.psum.reserve(sizeof(double) * nparts * num));
        HIPCHK(c, L.pnan.reserve(sizeof(unsigned) * nparts * num));
        hpk_launch_prep(s->raw, s->weight, (int)n, (int)num, (int64_t)ld, mw, L.psum.as<double>(), L.pnan.as<unsigned>(),
                        L.IR.as<double>(), L.b1.as<double>(), L.up);
        HIPCHK(c, hipGetLastError());
        s->IR = L.IR.as<double>();
        s->b1 = L.b1.as<double>();
        s->b2 = s->b1;
    }
    // (submit_impl adds the expected tables to the side stream, then lets the compute stream wait for all of it)
    return HPK_OK;
}

int check_band(hpk_ctx* c, const hpk_band* band) {
    if (!band || band->n <= 0 || band->num <= 0 || band->ld < band->num) return fail(c, HPK_ERR_INVALID, "bad band shape");
    if (!band->raw) return fail(c, HPK_ERR_INVALID, "raw pointer required");
    if (!band->balanced && !band->weight) return fail(c, HPK_ERR_INVALID, "either balanced or weight must be given");
    const bool derive = !band->IR && !band->bias1 && !band->bias2;
    if (derive && !band->weight) return fail(c, HPK_ERR_INVALID, "IR / biases can only be derived on the device from `weight`");
    if (!derive && (!band->IR || !band->bias1 || !band->bias2)) return fail(c, HPK_ERR_INVALID, "IR, bias1 and bias2 go together");
    return HPK_OK;
}

// small device scratch block layout (bytes)
constexpr size_t OFF_HIST = 0;                                                   // u64[65]
constexpr size_t OFF_FROZEN = OFF_HIST + 8 * (HPK_MAX_STEPS + 1);               // i32
constexpr size_t OFF_ERR = OFF_FROZEN + 8;                                       // i32
constexpr size_t OFF_EXEC = OFF_ERR + 8;                                         // i32[64]
constexpr size_t OFF_NUNITS = OFF_EXEC + 4 * HPK_MAX_STEPS;                      // u32 (+ pad): survives the overflow rerun
constexpr size_t OFF_NSURV = OFF_NUNITS + 8;                                     // u64 ... everything from here is reset by the rerun
constexpr size_t OFF_NVALID = OFF_NSURV + 8 * HPK_NREG * HPK_REG_STRIDE;          // u64[16]
constexpr size_t OFF_EMAX = OFF_NVALID + 8 * 2 * HPK_MAX_PAIRS;                  // u64[16]
constexpr size_t OFF_NOUT = OFF_EMAX + 8 * 2 * HPK_MAX_PAIRS;                    // u64
constexpr size_t OFF_FAM_M = OFF_NOUT + 8;                                       // u32[HPK_NFAM]
constexpr size_t OFF_FAM_F = OFF_FAM_M + 4 * HPK_NFAM;                           // u32[HPK_NFAM]
constexpr size_t SMALL_BYTES = OFF_FAM_F + 4 * HPK_NFAM;
constexpr size_t HEAD_INLINE = 4096;         // compacted survivors that travel to the host with the counters

}  // namespace

extern "C" {

void hpk_result_free(hpk_result* res) {
    if (res) delete reinterpret_cast<ResultBox*>(res);
}

}  // extern "C"

// One chromosome in flight.
struct hpk_job {
    hpk_ctx* ctx = nullptr;
    int lane = -1;
    hpk_params prm;
    int32_t n = 0, num = 0;
    int64_t ld = 0;
    Staged in;
    HpkStencilArgs sa;
    size_t off_rowlive = 0, off_inl = 0, head_bytes = 0, off_cnt = 0, off_cu = 0, dense_elems = 0;
    int64_t cap = 0, band_px = 0, ldo = 0;
    size_t zero_bytes = 0;
    hpk_params key;
    int nsets = 0, TR = 0, TC = 0, rounds = 2;
    bool sums = false, dense = false, do_score = true, phases = false, simple = false, time_stencil = true, redone = false;
    double t_begin = 0.0;
    ResultBox* box = nullptr;
    ~hpk_job() { delete box; }
};

namespace {

// scoring + BH-cut tightening + the head copy, on the job's lane (first pass and overflow rerun)
int launch_scoring(hpk_ctx* c, hpk_job* j, int attempt) {
    Lane& L = c->lane[j->lane];
    const HpkDevPlan& plan = L.plan_host;
    unsigned char* small = L.small.as<unsigned char>();
    unsigned long long* d_nsurv = reinterpret_cast<unsigned long long*>(small + OFF_NSURV);
    unsigned long long* d_nout = reinterpret_cast<unsigned long long*>(small + OFF_NOUT);
    unsigned int* d_fam_m = reinterpret_cast<unsigned int*>(small + OFF_FAM_M);
    unsigned int* d_fam_f = reinterpret_cast<unsigned int*>(small + OFF_FAM_F);
    unsigned int* d_cnt = reinterpret_cast<unsigned int*>(small + j->off_cnt);
    unsigned* d_chunkused = reinterpret_cast<unsigned*>(small + j->off_cu);
    const HpkStencilArgs& sa = j->sa;
    if (j->do_score) {
        const int64_t cap = j->cap;
        HIPCHK(c, L.surv.reserve(sizeof(HpkSurv) * (size_t)cap * HPK_NREG));
        HIPCHK(c, L.surv2.reserve(sizeof(HpkSurv) * (size_t)cap * HPK_NREG));
        if (attempt > 0) {       // overflow rerun: the chunk table no longer fits the zero block
            const size_t cu_bytes = sizeof(unsigned) * (size_t)(cap / HPK_SCH * HPK_NREG + 1);
            HIPCHK(c, L.chunkused.reserve(cu_bytes));
            HIPCHK(c, hipMemsetAsync(L.chunkused.p, 0, cu_bytes, c->stream));
            HIPCHK(c, hipMemsetAsync(small + OFF_NSURV, 0, SMALL_BYTES - OFF_NSURV, c->stream));
            HIPCHK(c, hipMemsetAsync(small + j->off_cnt, 0, sizeof(unsigned) * HPK_NFAM * HPK_TIGHTEN_MAX, c->stream));
            d_chunkused = L.chunkused.as<unsigned>();
        }
        HpkScoreArgs sc;
        std::memset(&sc, 0, sizeof(sc));
        sc.raw = j->in.raw; sc.bal = j->in.bal; sc.weight = j->in.weight; sc.plan = sa.plan;
        sc.etab = L.etab.as<double>(); sc.eedge = L.eedge.as<double>(); sc.IR = j->in.IR; sc.b1 = j->in.b1; sc.b2 = j->in.b2;
        sc.frozen = reinterpret_cast<int32_t*>(small + OFF_FROZEN);
        sc.hist_acc = sa.hist_acc;
        sc.hist_out = reinterpret_cast<unsigned long long*>(small + OFF_HIST);
        sc.executed = reinterpret_cast<int32_t*>(small + OFF_EXEC);
        sc.err = reinterpret_cast<int32_t*>(small + OFF_ERR);
        sc.bounds = c->d_bounds.as<double>(); sc.ptab = c->d_ptab.as<double>();
        sc.ptab_off = c->d_off.as<int32_t>(); sc.sfe = c->d_sfe.as<double>(); sc.sig = j->prm.sig;
        sc.n = j->n; sc.num = j->num; sc.ld = j->ld; sc.ldo = j->ldo; sc.mw = plan.mw; sc.D = plan.D;
        sc.rec_ent = sa.rec_ent; sc.rec_S = sa.rec_S; sc.rec_W = sa.rec_W; sc.tile_cnt = sa.tile_cnt;
        sc.units = L.units.as<uint2>(); sc.nunits = reinterpret_cast<const unsigned*>(small + OFF_NUNITS);
        sc.tilecap = sa.tilecap; sc.rec_stride = sa.rec_stride; sc.ntiles = sa.ntiles;
        sc.TR = j->TR; sc.TC = j->TC; sc.J = sa.J; sc.W = plan.W;
        sc.fam_m = d_fam_m; sc.fam_f = d_fam_f;
        sc.emax_bits = reinterpret_cast<unsigned long long*>(small + OFF_EMAX);
        sc.nvalid = reinterpret_cast<unsigned long long*>(small + OFF_NVALID);
        sc.nsurv = d_nsurv;
        // default: the scoring kernel keeps the p-value histogram the cut is derived from (no separate pass over the
        // survivors); HPK_ROUNDS = -1: hpk_thr_hist, >= 0: exact counting rounds
        int rounds = j->rounds;
        sc.hist = d_cnt; sc.hbins = 0; sc.nsets_half = plan.npairs;
        if (rounds <= -2) { sc.hbins = hpk_score_hist_bins(j->nsets); rounds = -100 - sc.hbins; }
        sc.cap = cap; sc.surv = L.surv.as<HpkSurv>(); sc.chunk_used = d_chunkused;
        hpk_launch_score(sc, plan.mode == HPK_MODE_BHFDR, c->cus, c->stream);
        HIPCHK(c, hipGetLastError());
        if (j->phases) (void)hipEventRecord(L.ev[4], c->stream);
        hpk_launch_tighten(sc.surv, d_nsurv, cap, sc.chunk_used, d_fam_m, d_fam_f, d_cnt, j->prm.sig, rounds, j->nsets,
                           reinterpret_cast<HpkSurv*>(small + j->off_inl), HEAD_INLINE, L.surv2.as<HpkSurv>(), d_nout,
                           j->in.bal, j->in.weight, j->ld, c->cus, c->stream);
        HIPCHK(c, hipGetLastError());
    } else {
        if (j->phases) (void)hipEventRecord(L.ev[4], c->stream);
    }
    if (j->phases) (void)hipEventRecord(L.ev[5], c->stream);
    // gap rows are produced by the stencil kernel (hpk_gap remains as an independent check for the tests)
    // ... as long as the band stops where the command lines stop it (num = D + maxww + 1, scripts/pyHICCUPS:146): the
    // tiles see the diagonals up to about D + maxww.  A caller of the drop-in hiccups() / bhfdr() may hand over more
    // diagonals, and callers.py:238 sums all of them: then the row kernel, which reads every stored diagonal, decides.
    if (attempt == 0 && (std::getenv("HPK_GAP_KERNEL") || j->num > plan.D + plan.W + 1)) {
        hpk_launch_gap(j->in.raw, j->in.bal, j->in.weight, j->n, j->num, j->ld, plan.mw, small + j->off_rowlive, c->stream);
        HIPCHK(c, hipGetLastError());
    }
    if (j->phases) (void)hipEventRecord(L.ev[6], c->stream);
    // counters, row flags and the first survivors in one copy into pinned memory
    if (std::getenv("HPK_HEAD_COPY")) {
        HIPCHK(c, hipMemcpyAsync(L.h_head, small, j->head_bytes, hipMemcpyDeviceToHost, c->stream));
    } else {        // pinned memory is device-visible: a small kernel writes it (head_bytes is a multiple of 16)
        if (j->do_score && !std::getenv("HPK_PUBLISH_ALL")) {
            // the stretches a chromosome fills: counters up to the families in use, their F(sig) counts, the row flags,
            // and as many inline survivors as the cut left
            const size_t nfam_b = sizeof(unsigned) * (size_t)j->nsets * (HPK_NB + 1);
            const size_t seg[3][2] = {{0, OFF_FAM_M + nfam_b}, {OFF_FAM_F, OFF_FAM_F + nfam_b}, {j->off_rowlive, j->off_rowlive + (size_t)j->n}};
            hpk_launch_publish_head(small, L.h_head, seg, j->off_inl, d_nout, (unsigned)HEAD_INLINE, (unsigned)sizeof(HpkSurv), c->stream);
        } else hpk_launch_publish(small, L.h_head, j->head_bytes, c->stream);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(L.ev_done, c->stream));
    return HPK_OK;
}

// stencil (+ the freeze decision) of a staged chromosome whose counter block is zero
int launch_stencil_stage(hpk_ctx* c, hpk_job* j) {
    Lane& L = c->lane[j->lane];
    const HpkStencilArgs& sa = j->sa;
    unsigned char* small = L.small.as<unsigned char>();
    if (j->time_stencil) (void)hipEventRecord(L.ev[1], c->stream);
    hpk_launch_stencil(sa, j->in.bal != nullptr, j->simple, c->stream);
    HIPCHK(c, hipGetLastError());
    if (j->time_stencil) (void)hipEventRecord(L.ev[2], c->stream);
    if (sa.hist_acc) {      // totals: the scoring kernel decides in its prologue; without scoring, a one-workgroup kernel
        if (!j->do_score) {
            hpk_launch_freeze_tot(sa.plan, sa.hist_acc, reinterpret_cast<unsigned long long*>(small + OFF_HIST),
                                  reinterpret_cast<int32_t*>(small + OFF_FROZEN), reinterpret_cast<int32_t*>(small + OFF_EXEC),
                                  reinterpret_cast<int32_t*>(small + OFF_ERR), c->stream);
            HIPCHK(c, hipGetLastError());
        }
    } else if (!sa.ticket) {       // HPK_FREEZE_KERNEL: the decision as a kernel of its own instead of the last stencil workgroup
        hpk_launch_freeze(sa.plan, reinterpret_cast<unsigned long long*>(small + OFF_HIST), sa.hist_part, sa.grid,
                          reinterpret_cast<int32_t*>(small + OFF_FROZEN), reinterpret_cast<int32_t*>(small + OFF_EXEC),
                          reinterpret_cast<int32_t*>(small + OFF_ERR), c->stream);
        HIPCHK(c, hipGetLastError());
    }
    if (j->phases) (void)hipEventRecord(L.ev[3], c->stream);
    return HPK_OK;
}

int submit_impl(hpk_ctx* c, hpk_job* j, const hpk_band* band, const hpk_params* prm) {
    Lane& L = c->lane[j->lane];
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
    const HpkDevPlan& plan = L.plan_host;
    const int W = plan.W, mw = plan.mw, D = plan.D;
    const int n = band->n, num = band->num;
    // output tile: what the halo leaves of the SAT tile, at most 4 rows per stencil wave (row slot = 2 bits of the record
    // entry, HPK_LISTCAP ids per wave)
    const int TR = std::min(HPK_LR - 2 * W - 1, HPK_ROWS_PER_WAVE * HPK_NWAVES), TC = HPK_LC - 2 * W - 1;
    static_assert(HPK_ROWS_PER_WAVE * (HPK_LC - 1) <= HPK_LISTCAP, "a wave's candidate list must hold its tile rows");
    if (D < mw) return fail(c, HPK_ERR_INVALID, "maxapart / res (%d) is below min(ww) (%d)", D, mw);
    j->prm = *prm; j->n = n; j->num = num; j->ld = band->ld; j->TR = TR; j->TC = TC;
    j->nsets = (plan.mode == HPK_MODE_BHFDR) ? 1 : 2 * plan.npairs;
    j->sums = (prm->flags & HPK_FLAG_DENSE_SUMS) != 0;
    j->dense = j->sums || (prm->flags & HPK_FLAG_DENSE_E) != 0;
    j->do_score = (prm->flags & HPK_FLAG_NO_SCORE) == 0;
    j->phases = (prm->flags & HPK_FLAG_PHASE_TIMING) != 0;
    j->rounds = std::getenv("HPK_ROUNDS") ? std::atoi(std::getenv("HPK_ROUNDS")) : -2;     // -2: histogram kept by hpk_score, -1: hpk_thr_hist
    if (j->rounds > 4) j->rounds = 4;
    const bool sums = j->sums, dense = j->dense;

    j->box = new ResultBox();
    std::memset(&j->box->pub, 0, sizeof(j->box->pub));

    // ---- inputs
    if (j->phases) (void)hipEventRecord(L.ev[0], c->stream);
    rc = stage_inputs(c, L, band, plan.mw, &j->in);
    if (rc != HPK_OK) return rc;
    const Staged& in = j->in;
    if (!plan_hit) {
        HIPCHK(c, L.plan.reserve(sizeof(HpkDevPlan)));
        HIPCHK(c, hipMemcpyAsync(L.plan.p, &L.plan_host, sizeof(HpkDevPlan), hipMemcpyHostToDevice, L.up));
        L.plan_key = key;
        L.plan_valid = true;
    }
    HIPCHK(c, L.etab.reserve(sizeof(double) * std::max<size_t>((size_t)plan.nsteps * 2 * (D + 1), 1)));
    HIPCHK(c, L.eedge.reserve(sizeof(double) * std::max<size_t>((size_t)2 * W * plan.nsteps * 2 * (D + 1), 1)));
    const int64_t ldo = ((int64_t)(D + 1) + 31) / 32 * 32;
    const size_t dense_elems = (size_t)plan.nslots * (size_t)n * (size_t)ldo;
    j->ldo = ldo; j->dense_elems = dense_elems;
    if (dense) {
        HIPCHK(c, L.dE.reserve(sizeof(double2) * dense_elems));
        HIPCHK(c, L.dW.reserve(dense_elems));
        if (sums) HIPCHK(c, L.dS.reserve(sizeof(double4) * dense_elems));
    }
    // One block, zero-filled by a single memset:
    //   head (one D2H copy):  counters | row-has-signal flags [n] | first HEAD_INLINE compacted survivors
    //   scratch:              per-workgroup histograms | per-tile record counts | tightening counters | chunk fill counts
    const int J_ = (TR + D - mw + TC - 1) / TC;
    const int ntiles_ = ((n + TR - 1) / TR) * J_;
    const int grid_ = std::max(8, std::min((c->cus / 8) * 8, ((ntiles_ + 7) / 8) * 8));
    int64_t band_px = 0;                    // pixels with mw <= d <= D inside the matrix
    for (int d = mw; d <= std::min(D, num - 1); ++d) if (n - d > 0) band_px += n - d;
    j->band_px = band_px;
    // survivor capacity per region; every scoring wave may hold one partly filled chunk of HPK_SCH records
    int64_t cap = (std::max<int64_t>(1 << 16, band_px * j->nsets / 6) + (int64_t)c->cus * 8 * 4 * HPK_SCH * 2) / HPK_NREG;
    if (const char* e = std::getenv("HPK_SURV_CAP")) cap = std::max<int64_t>(256, std::atoll(e));     // tests: force the overflow rerun
    cap = (cap + 255) / 256 * 256;
    j->cap = cap;
    auto up256 = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t off_rowlive = up256(SMALL_BYTES);
    const size_t off_inl = up256(off_rowlive + (size_t)n);
    const size_t head_bytes = off_inl + sizeof(HpkSurv) * HEAD_INLINE;
    const size_t off_hp = up256(head_bytes);
    const size_t off_tc = up256(off_hp + sizeof(unsigned) * (size_t)grid_ * (HPK_MAX_STEPS + 1));
    const size_t off_cnt = up256(off_tc + sizeof(unsigned) * (size_t)ntiles_);
    const size_t off_cu = up256(off_cnt + sizeof(unsigned) * HPK_NFAM * HPK_TIGHTEN_MAX);
    const size_t zero_bytes = (off_cu + sizeof(unsigned) * (size_t)(cap / HPK_SCH * HPK_NREG + 1) + 4095) / 4096 * 4096;
    j->off_rowlive = off_rowlive; j->off_inl = off_inl; j->head_bytes = head_bytes; j->off_cnt = off_cnt; j->off_cu = off_cu;
    j->zero_bytes = zero_bytes; j->key = key;
    HIPCHK(c, L.small.reserve(zero_bytes));
    // expected tables of this chromosome; the same launch zero-fills the block (extra workgroups).  On the lane's side
    // stream, like the uploads and the derivation of IR / biases: for the chromosome submitted one ahead this runs beside
    // the scoring and cut kernels of the chromosome before; the compute stream waits for the lot.
    static const bool etab_main = std::getenv("HPK_ETAB_MAIN") != nullptr;        // A/B: everything on the compute stream
    if (etab_main) {
        HIPCHK(c, hipEventRecord(L.ev_up, L.up));
        HIPCHK(c, hipStreamWaitEvent(c->stream, L.ev_up, 0));
    }
    hpk_launch_etab(L.plan.as<HpkDevPlan>(), plan.nsteps, D, W, in.IR, n, num, L.etab.as<double>(), L.eedge.as<double>(),
                    L.small.p, zero_bytes, etab_main ? c->stream : L.up);
    HIPCHK(c, hipGetLastError());
    if (!etab_main) {
        HIPCHK(c, hipEventRecord(L.ev_up, L.up));
        HIPCHK(c, hipStreamWaitEvent(c->stream, L.ev_up, 0));
    }
    if (L.h_head_cap < head_bytes) {
        if (L.h_head) (void)hipHostFree(L.h_head);
        L.h_head = nullptr; L.h_head_cap = 0;
        HIPCHK(c, hipHostMalloc(&L.h_head, head_bytes + head_bytes / 4, hipHostMallocMapped));
        L.h_head_cap = head_bytes + head_bytes / 4;
    }
    if (dense) {   // pixels outside the band are never written by the kernel
        HIPCHK(c, hipMemsetAsync(L.dE.p, 0, sizeof(double2) * dense_elems, c->stream));
        HIPCHK(c, hipMemsetAsync(L.dW.p, 0, dense_elems, c->stream));
        if (sums) HIPCHK(c, hipMemsetAsync(L.dS.p, 0, sizeof(double4) * dense_elems, c->stream));
    }

    unsigned char* small = L.small.as<unsigned char>();
    unsigned long long* d_hist = reinterpret_cast<unsigned long long*>(small + OFF_HIST);
    int32_t* d_frozen = reinterpret_cast<int32_t*>(small + OFF_FROZEN);
    int32_t* d_err = reinterpret_cast<int32_t*>(small + OFF_ERR);
    int32_t* d_exec = reinterpret_cast<int32_t*>(small + OFF_EXEC);

    // ---- stencil
    HpkStencilArgs& sa = j->sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.raw = in.raw; sa.bal = in.bal; sa.weight = in.weight;
    sa.plan = L.plan.as<HpkDevPlan>();
    sa.hist = d_hist;
    sa.n = n; sa.num = num; sa.ld = band->ld; sa.ldo = ldo; sa.W = W; sa.mw = mw; sa.D = D; sa.TR = TR; sa.TC = TC;
    sa.J = J_;
    sa.ntiles = ntiles_;
    sa.chunk = (sa.ntiles + 7) / 8;
    { const char* e = std::getenv("HPK_DBG_STOP"); sa.dbg_stop = e ? std::atoi(e) : 0; }
    { const char* e = std::getenv("HPK_TILE_ORDER"); sa.order = e ? std::atoi(e) : 1; }
    sa.grid = std::max(8, std::min((c->cus / 8) * 8, ((sa.chunk + 0) * 8)));
    sa.hist_part = reinterpret_cast<unsigned*>(small + off_hp);
    sa.tilecap = TR * TC;
    sa.rec_stride = (int64_t)sa.ntiles * sa.tilecap;
    HIPCHK(c, L.recE.reserve(sizeof(unsigned) * (size_t)sa.rec_stride));
    HIPCHK(c, L.recS.reserve(sizeof(double2) * (size_t)sa.rec_stride * plan.nslots));
    HIPCHK(c, L.recW.reserve((size_t)sa.rec_stride * plan.nslots));
    sa.rec_ent = L.recE.as<unsigned>(); sa.rec_S = L.recS.as<double2>(); sa.rec_W = L.recW.as<uint8_t>();
    sa.tile_cnt = reinterpret_cast<unsigned*>(small + off_tc);
    {   // at most ceil(tilecap / HPK_UNIT) units per tile
        const size_t upt = ((size_t)sa.tilecap + HPK_UNIT - 1) / HPK_UNIT;
        HIPCHK(c, L.units.reserve(sizeof(uint2) * (size_t)sa.ntiles * upt + 16));
        sa.units = L.units.as<uint2>();
        sa.nunits = reinterpret_cast<unsigned*>(small + OFF_NUNITS);
    }
    sa.gap = small + off_rowlive;
    sa.clk = nullptr;
#ifdef HPK_PHASE_CLOCK
    if (std::getenv("HPK_CLK_DUMP")) {
        HIPCHK(c, c->tmpD.reserve(sizeof(unsigned long long) * 8 * HPK_NWAVES * 1024));
        HIPCHK(c, hipMemsetAsync(c->tmpD.p, 0, sizeof(unsigned long long) * 8 * HPK_NWAVES * 1024, c->stream));
        sa.clk = c->tmpD.as<unsigned long long>();
    }
#endif
    sa.ticket = std::getenv("HPK_FREEZE_KERNEL") ? nullptr : reinterpret_cast<unsigned*>(small + OFF_NUNITS + 4);
    // default: the workgroups add their resolve counts into totals (the zeroed partial-count area, one word per 128 bytes)
    // and the freeze decision is replayed where it is needed; HPK_FREEZE_TICKET=1: by the stencil's last workgroup as before
    sa.hist_acc = nullptr;
    const char* ft = std::getenv("HPK_FREEZE_TICKET");
    if (!(ft && std::atoi(ft) != 0) && !std::getenv("HPK_FREEZE_KERNEL") &&
        sizeof(unsigned) * (size_t)grid_ * (HPK_MAX_STEPS + 1) >= 8 * (size_t)(HPK_MAX_STEPS + 1) * HPK_ACC_STRIDE) {
        sa.hist_acc = reinterpret_cast<unsigned long long*>(small + off_hp);
        sa.ticket = nullptr;
    }
    sa.frozen = d_frozen; sa.executed = d_exec; sa.err = d_err;
    sa.single = plan.single_p >= 0 ? 1 : 0;
    { const char* e = std::getenv("HPK_RISK_LOG2"); sa.risk = std::ldexp(1.0, e ? -std::atoi(e) : -12); }
    j->time_stencil = j->phases || !(prm->flags & HPK_FLAG_NO_STENCIL_TIMING);
    j->simple = plan.simple_reads != 0 && !std::getenv("HPK_GENERIC_SEARCH");
    // Which candidates get a record.  Dense outputs and probes want every candidate; the scoring kernel needs those
    // resolved up to the width the widening freezes at, which the chromosome collected last with the same parameters
    // tells within a step or so (HPK_SPEC=0: no guess, every resolved candidate; HPK_SPEC_MARGIN: widths added to it).
    sa.wguess = 255;
    if (j->do_score && !dense) {
        static const bool spec = !(std::getenv("HPK_SPEC") && std::atoi(std::getenv("HPK_SPEC")) == 0);
        static const int margin = std::getenv("HPK_SPEC_MARGIN") ? std::atoi(std::getenv("HPK_SPEC_MARGIN")) : 0;
        sa.wguess = W;
        if (spec && c->hint_w >= 0 && std::memcmp(&key, &c->hint_key, sizeof(key)) == 0) sa.wguess = std::min(W, c->hint_w + margin);
        if (const char* e = std::getenv("HPK_SPEC_FORCE")) sa.wguess = std::min(W, std::max(0, std::atoi(e)));     // tests: a bound that is too narrow
    }
    rc = launch_stencil_stage(c, j);
    if (rc != HPK_OK) return rc;
    return launch_scoring(c, j, 0);
}

int collect_impl(hpk_ctx* c, hpk_job* j, hpk_result** out) {
    Lane& L = c->lane[j->lane];
    const HpkDevPlan& plan = L.plan_host;
    const hpk_params* prm = &j->prm;
    const int n = j->n, num = j->num, nsets = j->nsets, mw = plan.mw, D = plan.D, TR = j->TR, TC = j->TC;
    const bool sums = j->sums, dense = j->dense, do_score = j->do_score;
    const size_t off_rowlive = j->off_rowlive, off_inl = j->off_inl, dense_elems = j->dense_elems;
    const int64_t ldo = j->ldo;
    const HpkStencilArgs& sa = j->sa;
    const Staged& in = j->in;
    ResultBox* box = j->box;
    hpk_result& R = box->pub;
    (void)num; (void)mw;
    const unsigned char* hsmall = static_cast<const unsigned char*>(L.h_head);
    for (int attempt = 0;; ++attempt) {
        HIPCHK(c, hipEventSynchronize(L.ev_done));       // this chromosome only; the next one keeps running
        {   // the widening froze later than the bound the stencil wrote records up to: once more, with every resolved candidate
            const int32_t fz = *reinterpret_cast<const int32_t*>(hsmall + OFF_FROZEN);
            const int32_t er = *reinterpret_cast<const int32_t*>(hsmall + OFF_ERR);
            if (do_score && er == 0 && j->sa.wguess < plan.W && fz > j->sa.wguess) {
                j->sa.wguess = plan.W;
                j->redone = true;
                c->spec_reruns += 1;
                HIPCHK(c, hipMemsetAsync(L.small.p, 0, j->zero_bytes, c->stream));
                int rc = launch_stencil_stage(c, j);
                if (rc == HPK_OK) rc = launch_scoring(c, j, 0);
                if (rc != HPK_OK) return rc;
                attempt = -1;
                continue;
            }
        }
        unsigned long long ns = 0;              // fullest region
        for (int rg = 0; rg < HPK_NREG; ++rg)
            ns = std::max(ns, reinterpret_cast<const unsigned long long*>(hsmall + OFF_NSURV)[rg * HPK_REG_STRIDE]);
        if (!do_score || (int64_t)ns <= j->cap) break;
        if (attempt == 1) return fail(c, HPK_ERR_NOMEM, "survivor buffer overflow");
        j->cap = ((int64_t)ns * 2 + 1024 + 255) / 256 * 256;      // rerun the scoring with room for everything
        int rc = launch_scoring(c, j, 1);
        if (rc != HPK_OK) return rc;
    }
#ifdef HPK_PHASE_CLOCK
    if (const char* path = std::getenv("HPK_CLK_DUMP")) {       // [grid][waves][8] u64, overwritten by every chromosome
        std::vector<unsigned long long> h((size_t)8 * HPK_NWAVES * sa.grid);
        HIPCHK(c, hipMemcpy(h.data(), c->tmpD.p, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(path, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
    }
#endif
    box->gap.resize(n);
    for (int r = 0; r < n; ++r) box->gap[r] = hsmall[off_rowlive + r] ? 0 : 1;
    R.band_px = j->band_px;
    R.stencil_tiles = sa.ntiles;
    R.stencil_kernel = hpk_stencil_s_applies(sa, j->simple) ? 2 : 1;

    // ---- results to host
    const unsigned long long* h_hist = reinterpret_cast<const unsigned long long*>(hsmall + OFF_HIST);
    const int32_t h_frozen = *reinterpret_cast<const int32_t*>(hsmall + OFF_FROZEN);
    const int32_t h_err = *reinterpret_cast<const int32_t*>(hsmall + OFF_ERR);
    if (do_score && h_err == 0) { c->hint_w = h_frozen; c->hint_key = j->key; }      // the next stencil's record bound
    R.record_bound = sa.wguess;
    R.redone = j->redone ? 1 : 0;
    const int32_t* h_exec = reinterpret_cast<const int32_t*>(hsmall + OFF_EXEC);
    const unsigned long long h_nsurv = *reinterpret_cast<const unsigned long long*>(hsmall + OFF_NOUT);
    const unsigned long long* h_emax = reinterpret_cast<const unsigned long long*>(hsmall + OFF_EMAX);
    const unsigned int* h_chist = reinterpret_cast<const unsigned int*>(hsmall + OFF_FAM_M);

    R.nsteps = plan.nsteps;
    for (int s = 0; s < plan.nsteps; ++s) {
        R.step_pi[s] = plan.steps[s].pi;
        R.step_wi[s] = plan.steps[s].wi;
        R.step_executed[s] = h_exec[s];
        R.step_resolved[s] = (int64_t)h_hist[s];
    }
    R.frozen_w = h_frozen;
    R.nslots = plan.nslots;
    for (int q = 0; q < plan.nslots; ++q) R.slot_pi[q] = plan.slot_pi[q];
    R.ncand = (int64_t)h_hist[HPK_HIST_NCAND];
    R.nsurv_sig = 0;
    for (int i = 0; i < nsets * (HPK_NB + 1); ++i)      // (only the families of the sets in use are copied back)
        R.nsurv_sig += reinterpret_cast<const unsigned int*>(hsmall + OFF_FAM_F)[i];
    R.nsurv_cut = (int64_t)h_nsurv;
    R.gap = box->gap.data();
    if (h_err != 0 && sa.dbg_stop == 0) {
        const HpkDevStep& st = plan.steps[h_err - 1];
        return fail(c, HPK_ERR_EMPTY_STEP, "step (%d,%d) entered with no unresolved candidate (of %lld); the reference raises here "
                    "(hicpeaks/callers.py:203-208)", st.pi, st.wi, (long long)R.ncand);
    }

    const double t_d2h0 = now_ms();
    std::vector<Surv> sv;
    if (do_score && h_nsurv) {
        const size_t ns = (size_t)h_nsurv;
        const HpkSurv* head = reinterpret_cast<const HpkSurv*>(hsmall + off_inl);
        std::vector<HpkSurv> rest;
        if (ns > HEAD_INLINE) {
            rest.resize(ns - HEAD_INLINE);
            // beyond the inlined head: fetched on the lane's own (idle) copy stream, so the next chromosome's kernels
            // on the compute stream are not waited for
            HIPCHK(c, hipMemcpyAsync(rest.data(), L.surv2.p, sizeof(HpkSurv) * (ns - HEAD_INLINE), hipMemcpyDeviceToHost, L.up));
            HIPCHK(c, hipStreamSynchronize(L.up));
        }
        auto rec_at = [&](size_t i) -> const HpkSurv& { return i < HEAD_INLINE ? head[i] : rest[i - HEAD_INLINE]; };
        sv.resize(ns);
        for (size_t i = 0; i < ns; ++i) {
            const HpkSurv& rc_ = rec_at(i);
            sv[i] = Surv{rc_.x, rc_.y, rc_.set, rc_.chunk, rc_.flag, 0, rc_.O, rc_.E, rc_.p, rc_.bal, 1.0};
        }
    }
    if (dense) {
        HpkDenseArgs da;
        da.rec_ent = sa.rec_ent; da.rec_S = sa.rec_S; da.rec_W = sa.rec_W; da.tile_cnt = sa.tile_cnt;
        da.tilecap = sa.tilecap; da.rec_stride = sa.rec_stride; da.ntiles = sa.ntiles; da.TR = TR; da.TC = TC; da.J = sa.J;
        da.plan = sa.plan; da.etab = L.etab.as<double>(); da.eedge = L.eedge.as<double>();
        da.IR = in.IR; da.b1 = in.b1; da.b2 = in.b2; da.n = n; da.num = num; da.ldo = ldo; da.mw = mw; da.D = D;
        da.dE = L.dE.as<double2>(); da.dW = L.dW.as<uint8_t>(); da.dS = sums ? L.dS.as<double4>() : nullptr;
        hpk_launch_dense(da, c->stream);
        HIPCHK(c, hipGetLastError());
        box->denseE.resize(dense_elems * 2);
        box->denseW.resize(dense_elems);
        HIPCHK(c, hipMemcpyAsync(box->denseE.data(), L.dE.p, sizeof(double2) * dense_elems, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(box->denseW.data(), L.dW.p, dense_elems, hipMemcpyDeviceToHost, c->stream));
        if (sums) {
            box->denseS.resize(dense_elems * 4);
            HIPCHK(c, hipMemcpyAsync(box->denseS.data(), L.dS.p, sizeof(double4) * dense_elems, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        R.dense_ld = ldo;
        R.dense_E = box->denseE.data();
        R.dense_w = box->denseW.data();
        R.dense_sums = sums ? box->denseS.data() : nullptr;
    }
    const double t_d2h1 = now_ms();

    // ---- lambda chunks (callers.py:30) + Benjamini-Hochberg per family (callers.py:273 / 545)
    R.nsets = do_score ? nsets : 0;
    std::vector<std::vector<Surv*>> kept(nsets);
    if (do_score) {
        // families = (set, chunk): one sort by (family, p) on flat 16-byte keys (p >= 0, so its bit pattern orders
        // like its value), then Benjamini-Hochberg on each run
        struct Key { uint32_t fam, idx; uint64_t pbits; };
        std::vector<Key> keys(sv.size());
        for (size_t i = 0; i < sv.size(); ++i) {
            uint64_t b;
            std::memcpy(&b, &sv[i].p, 8);
            keys[i] = Key{(uint32_t)sv[i].set << 8 | sv[i].chunk, (uint32_t)i, b};
        }
        std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
            return a.fam != b.fam ? a.fam < b.fam : a.pbits < b.pbits; });
        std::vector<Surv*> order(sv.size());
        for (size_t i = 0; i < sv.size(); ++i) order[i] = &sv[keys[i].idx];
        const double t_h1 = now_ms();
        std::vector<int> numbins(nsets, 0);
        box->fam.resize((size_t)2 * nsets * (HPK_NB + 1));
        std::memcpy(box->fam.data(), h_chist, sizeof(uint32_t) * (size_t)nsets * (HPK_NB + 1));
        std::memcpy(box->fam.data() + (size_t)nsets * (HPK_NB + 1), hsmall + OFF_FAM_F, sizeof(uint32_t) * (size_t)nsets * (HPK_NB + 1));
        for (int s = 0; s < nsets; ++s) {
            hpk_set& hs = R.sets[s];
            hs.pair = (plan.mode == HPK_MODE_BHFDR) ? 0 : s / 2;
            hs.fl = (plan.mode == HPK_MODE_BHFDR) ? 0 : s % 2;
            // pixels with E > 0: the families of the set added up (family 0 = those without a chunk, not a family of tests)
            hs.nvalid = 0;
            for (int ch = 0; ch <= HPK_NB; ++ch) hs.nvalid += (int64_t)h_chist[(size_t)s * (HPK_NB + 1) + ch];
            box->fam[(size_t)s * (HPK_NB + 1)] = 0u;
            double emax = 0.0;
            std::memcpy(&emax, &h_emax[s], 8);
            hs.emax = emax;
            int numbin = 0;
            if (plan.mode == HPK_MODE_HICCUPS && hs.nvalid > 0) {
                const double nb = std::ceil(std::log(emax) / std::log(2.0) * 3.0 + 1.0);     // callers.py:30
                numbin = nb < 0 ? 0 : (nb > HPK_NB ? HPK_NB : (int)nb);
            }
            hs.numbin = numbin;
            hs.chunk_tests = box->fam.data() + (size_t)s * (HPK_NB + 1);
            hs.chunk_below = box->fam.data() + (size_t)(nsets + s) * (HPK_NB + 1);
            // chunks beyond numbin do not exist for the reference (their pixels keep p = q = 1, callers.py:259-260)
            for (int ch = ((plan.mode == HPK_MODE_BHFDR) ? 1 : numbin) + 1; ch <= HPK_NB; ++ch) {
                box->fam[(size_t)s * (HPK_NB + 1) + ch] = 0u;
                box->fam[(size_t)(nsets + s) * (HPK_NB + 1) + ch] = 0u;
            }
            numbins[s] = (plan.mode == HPK_MODE_BHFDR) ? 1 : numbin;
        }
        for (size_t i = 0; i < order.size();) {
            size_t j = i;
            while (j < order.size() && order[j]->set == order[i]->set && order[j]->chunk == order[i]->chunk) ++j;
            const int s = order[i]->set, ch = order[i]->chunk;
            if (s < nsets && ch >= 1 && ch <= numbins[s]) {        // chunks beyond numbin keep p = q = 1 (callers.py:259-260)
                bh_family(order.data() + i, j - i, h_chist[(size_t)s * (HPK_NB + 1) + ch], prm->sig, plan.mode == HPK_MODE_BHFDR);
                for (size_t t = i; t < j; ++t) if (order[t]->keep) kept[s].push_back(order[t]);
            }
            i = j;
        }
        const double t_h2 = now_ms();
        for (int s = 0; s < nsets; ++s)
            std::sort(kept[s].begin(), kept[s].end(), [](const Surv* a, const Surv* b) {
                return a->x != b->x ? a->x < b->x : a->y < b->y; });
        if (std::getenv("HPK_HOST_PROF")) std::fprintf(stderr, "[hpk host] n=%zu sort=%.3f bh=%.3f\n", sv.size(), t_h1 - t_d2h1, t_h2 - t_h1);
        size_t total = 0;
        for (int s = 0; s < nsets; ++s) total += kept[s].size();
        box->x.reserve(total); box->y.reserve(total); box->O.reserve(total); box->bal.reserve(total);
        box->E.reserve(total); box->p.reserve(total); box->q.reserve(total); box->oz.reserve(total);
        for (int s = 0; s < nsets; ++s) {
            R.sets[s].begin = (int64_t)box->x.size();
            for (const Surv* p : kept[s]) {
                box->x.push_back(p->x); box->y.push_back(p->y); box->O.push_back((double)p->O); box->bal.push_back(p->bal);
                box->E.push_back(p->E); box->p.push_back(p->p); box->q.push_back(p->q); box->oz.push_back(p->flag);
            }
            R.sets[s].end = (int64_t)box->x.size();
        }
        R.nsig = (int64_t)total;
        R.x = box->x.data(); R.y = box->y.data(); R.O = box->O.data(); R.bal = box->bal.data();
        R.E = box->E.data(); R.p = box->p.data(); R.q = box->q.data(); R.other_zero = box->oz.data();
    }
    const double t_end = now_ms();
    if (std::getenv("HPK_HOST_PROF")) std::fprintf(stderr, "[hpk host] total host_bh=%.3f d2h=%.3f\n", t_end - t_d2h1, t_d2h1 - t_d2h0);

    float ms = 0.f;
    if (j->time_stencil && hipEventElapsedTime(&ms, L.ev[1], L.ev[2]) == hipSuccess) R.ms_stencil = ms;
    if (j->phases) {
        if (hipEventElapsedTime(&ms, L.ev[0], L.ev[1]) == hipSuccess) R.ms_h2d = ms;
        if (hipEventElapsedTime(&ms, L.ev[2], L.ev[3]) == hipSuccess) R.ms_freeze = ms;
        if (hipEventElapsedTime(&ms, L.ev[3], L.ev[4]) == hipSuccess) R.ms_score = ms;
        if (hipEventElapsedTime(&ms, L.ev[4], L.ev[5]) == hipSuccess) R.ms_tighten = ms;
        if (hipEventElapsedTime(&ms, L.ev[5], L.ev[6]) == hipSuccess) R.ms_gap = ms;
    }
    R.ms_d2h = (float)(t_d2h1 - t_d2h0);
    R.ms_host_bh = (float)(t_end - t_d2h1);

    R.ms_total = (float)(t_end - j->t_begin);
    j->box = nullptr;
    *out = &box->pub;
    return HPK_OK;
}

int acquire_lane(hpk_ctx* c) {
    for (int l = 0; l < HPK_LANES; ++l) if (!c->lane[l].busy) return l;
    return -1;
}

}  // namespace

extern "C" {

int hpk_pipeline_depth(void) { return HPK_LANES; }

int hpk_submit_band(hpk_ctx* c, const hpk_band* band, const hpk_params* prm, hpk_job** job) {
    if (!c) return HPK_ERR_INVALID;
    if (!job || !prm) return fail(c, HPK_ERR_INVALID, "params / job is NULL");
    *job = nullptr;
    int rc = check_band(c, band);
    if (rc != HPK_OK) return rc;
    const int lane = acquire_lane(c);
    if (lane < 0) return fail(c, HPK_ERR_BUSY, "all %d lanes hold a chromosome in flight: collect one first", HPK_LANES);
    (void)hipSetDevice(c->device);
    hpk_job* j = new hpk_job();
    j->ctx = c; j->lane = lane; j->t_begin = now_ms();
    rc = submit_impl(c, j, band, prm);
    if (rc != HPK_OK) {
        (void)hipStreamSynchronize(c->stream);
        delete j;
        return rc;
    }
    c->lane[lane].busy = true;
    *job = j;
    return HPK_OK;
}

int hpk_collect(hpk_ctx* c, hpk_job* job, hpk_result** out) {
    if (!c || !job || job->ctx != c) return c ? fail(c, HPK_ERR_INVALID, "job does not belong to this context") : HPK_ERR_INVALID;
    if (out) *out = nullptr;
    (void)hipSetDevice(c->device);
    hpk_result* res = nullptr;
    int rc = collect_impl(c, job, &res);
    if (rc != HPK_OK) (void)hipStreamSynchronize(c->stream);
    c->lane[job->lane].busy = false;
    delete job;
    if (rc != HPK_OK) return rc;
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

int64_t hpk_band_from_coo(const int64_t* bin1, const int64_t* bin2, const void* count, int32_t count_f64, int64_t nnz,
                          int32_t n, int32_t num, int64_t ld, float* raw) {
    if (!bin1 || !bin2 || !count || !raw || nnz < 0 || n <= 0 || num <= 0 || ld < num) return HPK_ERR_INVALID;
    const int32_t* ci = static_cast<const int32_t*>(count);
    const double* cd = static_cast<const double*>(count);
    int64_t stored = 0;
    for (int64_t t = 0; t < nnz; ++t) {
        const int64_t a = bin1[t] < bin2[t] ? bin1[t] : bin2[t], b = bin1[t] < bin2[t] ? bin2[t] : bin1[t];
        if (a < 0 || b >= n) continue;
        const int64_t k = b - a;
        if (k >= num) continue;                             // beyond the band
        raw[a * ld + k] += count_f64 ? (float)cd[t] : (float)ci[t];
        ++stored;
    }
    return stored;
}

int hpk_probe_sums(hpk_ctx* c, const hpk_band* band, const hpk_params* prm, const int32_t* rows, const int32_t* cols,
                   int64_t count, double* out) {
    if (!c) return HPK_ERR_INVALID;
    if (!prm || !rows || !cols || !out || count < 0) return fail(c, HPK_ERR_INVALID, "bad arguments");
    hpk_params p2 = *prm;
    p2.flags = HPK_FLAG_NO_SCORE;               // stencil + freeze only; the records stay in the lane's workspaces
    hpk_job* job = nullptr;
    int rc = hpk_submit_band(c, band, &p2, &job);
    if (rc != HPK_OK) return rc;
    Lane& L = c->lane[job->lane];
    const HpkStencilArgs& sa = job->sa;
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
        da.rec_ent = sa.rec_ent; da.rec_S = sa.rec_S; da.rec_W = sa.rec_W; da.tile_cnt = sa.tile_cnt;
        da.tilecap = sa.tilecap; da.rec_stride = sa.rec_stride; da.ntiles = sa.ntiles; da.TR = job->TR; da.TC = job->TC; da.J = sa.J;
        da.plan = sa.plan; da.etab = L.etab.as<double>(); da.eedge = L.eedge.as<double>();
        da.IR = job->in.IR; da.b1 = job->in.b1; da.b2 = job->in.b2; da.n = job->n; da.num = job->num; da.ldo = job->ldo;
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
    if (!c) return HPK_ERR_INVALID;
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
    Staged in;
    rc = stage_inputs(c, L, band, plan.mw, &in);
    if (rc != HPK_OK) return rc;
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
