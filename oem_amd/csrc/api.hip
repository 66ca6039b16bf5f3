// api.hip -- C ABI of liboemgpu: contexts, workspace, argument checks, and the host-side driver that the
// reference keeps in src/oem_dense.cpp:30-309, src/oem_xtx.cpp:29-219 and src/oem_big.cpp:30-258
// (lambda bookkeeping, result packing, DataStd::recover).  All arithmetic of the hot path runs in the HIP
// kernels of gram.hip / path_small.hip / path_large.hip; there is no CPU fallback.
#include "ctx.hpp"

#include <atomic>
#include <unistd.h>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace oemgpu {

// ---------------------------------------------------------------- the environment switches (switches.hpp), parsed once
static std::mutex g_sw_mu;
static std::atomic<const Switches *> g_sw{nullptr};
static unsigned g_sw_generation = 0;
static const Switches *sw_parse()
{
    Switches *t = new Switches();                     // (a reload leaks the old table -- a few hundred bytes, tests only: readers may still hold it)
    auto read = [](const char *name, Switch &v) {
        const char *e = getenv(name);
        v.set = e != nullptr;
        if (e) { v.num = atoll(e); strncpy(v.str, e, sizeof v.str - 1); v.str[sizeof v.str - 1] = 0; }
    };
#define OEM_SW_READ(name) read(#name, t->name);
    OEM_SWITCH_TABLE(OEM_SW_READ)
#undef OEM_SW_READ
    t->generation = ++g_sw_generation;
    return t;
}
const Switches &sw()
{
    const Switches *t = g_sw.load(std::memory_order_acquire);
    if (__builtin_expect(t != nullptr, 1)) return *t;
    std::lock_guard<std::mutex> lk(g_sw_mu);
    t = g_sw.load(std::memory_order_acquire);
    if (!t) { t = sw_parse(); g_sw.store(t, std::memory_order_release); }
    return *t;
}
void sw_reload()
{
    std::lock_guard<std::mutex> lk(g_sw_mu);
    g_sw.store(sw_parse(), std::memory_order_release);
}

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// The cooperating-workgroup engine (path_coop.hip) spins on its partners, so every workgroup of every such kernel in flight must
// be resident at once: concurrent callers (xval.oem's fold threads, user threads) queue here for CU slots.
// The count is per DEVICE: launches on different GPUs share no CUs and must not queue behind each other (the penalty split of
// solve_summed and xval.oem over devices run one such kernel per device side by side -- ADVICE r2).
static const int COOP_MAX_DEV = 64;
static std::mutex g_coop_mu;
static std::condition_variable g_coop_cv;
static int g_coop_in_flight[COOP_MAX_DEV] = {0};
static int g_xcd_load[COOP_MAX_DEV][8] = {{0}};                    // CUs of every XCD booked by persistent launches in flight (per device, under g_coop_mu)
static std::atomic<unsigned> g_xcd_next{(unsigned)getpid()};      // (processes that share a GPU start at different XCDs more often than not)
// CU slots of the persistent engines: all workgroups of all such kernels in flight must be resident at once, so concurrent callers queue.
// Booked per device AND per XCD: a device-scope launch of `want` workgroups is dealt round the XCDs by the dispatcher
// (ceil(want / 8) CUs of each); a one-XCD launch (path_coop.hip, q <= 512) puts the W workgroups of instance y on XCD (base + y) mod 8
// and gets the base for which the fullest XCD it touches stays emptiest -- and waits while no base leaves every XCD within its CUs.
struct CoopSlots {
    int n = 0, dev = 0;
    int add[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void take(int device, int want, int capacity)
    {
        dev = (device >= 0 && device < COOP_MAX_DEV) ? device : COOP_MAX_DEV - 1;
        std::unique_lock<std::mutex> lk(g_coop_mu);
        g_coop_cv.wait(lk, [&] { return g_coop_in_flight[dev] == 0 || g_coop_in_flight[dev] + want <= capacity; });
        g_coop_in_flight[dev] += want; n += want;
        for (int x = 0; x < 8; ++x) { const int a = (want + 7) / 8; add[x] += a; g_xcd_load[dev][x] += a; }
    }
    // `ninst` instances of W workgroups, each on ONE XCD of a device with num_cu CUs; returns the XCD of instance 0
    int take_local(int device, int W, int ninst, int num_cu, int capacity)
    {
        dev = (device >= 0 && device < COOP_MAX_DEV) ? device : COOP_MAX_DEV - 1;
        const int per = num_cu / 8, want = W * ninst, start = (int)(g_xcd_next.fetch_add(1u, std::memory_order_relaxed) & 7u);
        int best = -1;
        auto fits = [&](bool must) {
            best = -1;
            int best_peak = 0;
            for (int k = 0; k < 8; ++k) {
                const int base = (start + k) & 7;
                int peak = 0;
                for (int x = 0; x < 8; ++x) {
                    const int cnt = ninst / 8 + (((x - base) & 7) < ninst % 8 ? 1 : 0);       // instances y with (base + y) mod 8 == x
                    if (cnt) { const int l = g_xcd_load[dev][x] + cnt * W; if (l > peak) peak = l; }
                }
                if (best < 0 || peak < best_peak) { best = base; best_peak = peak; }
            }
            return must || best_peak <= per;
        };
        std::unique_lock<std::mutex> lk(g_coop_mu);
        g_coop_cv.wait(lk, [&] { return g_coop_in_flight[dev] == 0 ? fits(true) : (g_coop_in_flight[dev] + want <= capacity && fits(false)); });
        g_coop_in_flight[dev] += want; n += want;
        for (int x = 0; x < 8; ++x) {
            const int a = (ninst / 8 + (((x - best) & 7) < ninst % 8 ? 1 : 0)) * W;
            add[x] += a; g_xcd_load[dev][x] += a;
        }
        return best;
    }
    void release()
    {
        if (n) {
            { std::lock_guard<std::mutex> lk(g_coop_mu); g_coop_in_flight[dev] -= n; for (int x = 0; x < 8; ++x) { g_xcd_load[dev][x] -= add[x]; add[x] = 0; } }
            n = 0;
            g_coop_cv.notify_all();
        }
    }
    ~CoopSlots() { release(); }
};

// (kernel, device) -> largest dynamic-LDS limit set so far
int lds_limit_once(const void *fn, size_t bytes)
{
    struct Key { const void *fn; int dev; size_t bytes; };
    static std::mutex mu;
    static std::vector<Key> seen;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Key &k : seen) if (k.fn == fn && k.dev == dev && k.bytes >= bytes) return 0;
    }
    OEM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    std::lock_guard<std::mutex> lk(mu);
    for (Key &k : seen) if (k.fn == fn && k.dev == dev) { if (k.bytes < bytes) k.bytes = bytes; return 0; }
    seen.push_back(Key{fn, dev, bytes});
    return 0;
}

static thread_local int (*g_poll)(void *) = nullptr;
static thread_local void *g_poll_arg = nullptr;
bool caller_interrupted() { return g_poll && g_poll(g_poll_arg) != 0; }
struct PollScope {          // the engines poll the caller's interrupt between batches of iterations, on the calling thread only
    PollScope(const oemgpu_opts *o) { g_poll = o->interrupt; g_poll_arg = o->interrupt_arg; }
    ~PollScope() { g_poll = nullptr; g_poll_arg = nullptr; }
};

}  // namespace oemgpu

using namespace oemgpu;

namespace oemgpu {

int ctx_reserve(oemgpu_ctx *c, size_t bytes)
{
    if (bytes <= c->ws_bytes) return 0;
    if (c->ws) { OEM_HIP(hipStreamSynchronize(c->stream)); OEM_HIP(hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
    bytes = bytes + bytes / 8 + (1 << 20);
    OEM_HIP(hipMalloc((void **)&c->ws, bytes)); ++g_alloc_count;
    c->ws_bytes = bytes;
    return 0;
}

int ctx_grow(oemgpu_ctx *c, char **buf, size_t *have, size_t bytes)
{
    if (bytes <= *have) return 0;
    if (*buf) { OEM_HIP(hipStreamSynchronize(c->stream)); OEM_HIP(hipFree(*buf)); *buf = nullptr; *have = 0; }
    OEM_HIP(hipMalloc((void **)buf, bytes)); ++g_alloc_count;
    *have = bytes;
    return 0;
}

int set_device(const oemgpu_ctx *c) { OEM_HIP(hipSetDevice(c->device)); return 0; }

// ---------------------------------------------------------------- process-wide context cache
static std::mutex g_cache_mu;
static std::vector<oemgpu_ctx *> g_cache;

oemgpu_ctx *ctx_acquire(int device)
{
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (oemgpu_ctx *c : g_cache)
            if (c->device == device && !c->busy) { c->busy = true; (void)hipSetDevice(device); return c; }
    }
    oemgpu_ctx *c = oemgpu_create(device, nullptr);
    if (!c) return nullptr;
    c->cached = true; c->busy = true;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache.push_back(c);
    return c;
}

// A cached context keeps its grow-only buffers between calls (no allocation in the steady state) -- up to a bound: the copy of the
// host rows (c->xres: up to half of HBM for one large fit) xval.oem's fold-ordered copy (c->aux) and the packed triangle of a q > 4096 Gram (c->pack_buf) are freed on release when they
// exceed OEMGPU_CACHE_KEEP_BYTES (default: an eighth of the device's memory, 36 GB on MI355X), so that one large oem() / big.oem()
// call does not starve torch or other users of the GPU for the rest of the session.  Streams, pinned staging lanes
// and the workspace stay.  The caller has synchronised the context's stream.
void ctx_release(oemgpu_ctx *c)
{
    if (!c) return;
    size_t keep = c->hbm_total / 8;
    if (sw().OEMGPU_CACHE_KEEP_BYTES.set && sw().OEMGPU_CACHE_KEEP_BYTES.num >= 0) keep = (size_t)sw().OEMGPU_CACHE_KEEP_BYTES.num;
    if (c->xres_bytes > keep || c->aux_bytes > keep || c->pack_bytes > keep) {
        (void)hipSetDevice(c->device);
        if (c->xres_bytes > keep) { (void)hipFree(c->xres); c->xres = nullptr; c->xres_bytes = 0; }
        if (c->aux_bytes > keep) { (void)hipFree(c->aux); c->aux = nullptr; c->aux_bytes = 0; }
        if (c->pack_bytes > keep) { (void)hipFree(c->pack_buf); c->pack_buf = nullptr; c->pack_bytes = 0; }
    }
    std::lock_guard<std::mutex> lk(g_cache_mu);
    c->busy = false;
}

}  // namespace oemgpu

namespace {

int ctx_pinned_in(oemgpu_ctx *c, size_t bytes)
{
    if (bytes <= c->pinned_in_bytes) return 0;
    if (c->pinned_in) { OEM_HIP(hipStreamSynchronize(c->stream)); OEM_HIP(hipHostFree(c->pinned_in)); c->pinned_in = nullptr; c->pinned_in_bytes = 0; }
    bytes = (bytes + 4095) / 4096 * 4096;
    OEM_HIP(hipHostMalloc((void **)&c->pinned_in, bytes, hipHostMallocDefault)); ++g_alloc_count;
    c->pinned_in_bytes = bytes;
    return 0;
}

int ctx_pinned(oemgpu_ctx *c, size_t bytes)
{
    if (bytes <= c->pinned_bytes) return 0;
    if (c->pinned) { OEM_HIP(hipStreamSynchronize(c->stream)); OEM_HIP(hipHostFree(c->pinned)); c->pinned = nullptr; c->pinned_bytes = 0; }
    bytes = bytes + bytes / 8 + 4096;
    OEM_HIP(hipHostMalloc((void **)&c->pinned, bytes, hipHostMallocDefault)); ++g_alloc_count;
    c->pinned_bytes = bytes;
    return 0;
}

struct Timer {
    oemgpu_ctx *c; int id;
    Timer(oemgpu_ctx *c_, int id_) : c(c_), id(id_)
    {
        if (c->timing) { (void)hipEventRecord(c->ev[2 * id], c->stream); c->ev_used[id] = true; }
    }
    ~Timer() { if (c->timing) (void)hipEventRecord(c->ev[2 * id + 1], c->stream); }
};

// ---------------------------------------------------------------- argument checks (the R front ends stop() on these)
int nl_of(const oemgpu_opts *o) { return (o->lambda_user && o->nlambda_user > 0) ? o->nlambda_user : o->nlambda; }

int check_opts(const oemgpu_opts *o, int p, int ngroupvars_expected)
{
    if (!o) { set_error("opts is NULL"); return OEMGPU_ERR_ARG; }
    if (o->npen < 1 || !o->penalty) { set_error("at least one penalty is required"); return OEMGPU_ERR_ARG; }
    bool any_grp = false;
    for (int k = 0; k < o->npen; ++k) {
        if (o->penalty[k] < 0 || o->penalty[k] >= OEMGPU_NPENALTIES) { set_error("unknown penalty code %d", o->penalty[k]); return OEMGPU_ERR_ARG; }
        any_grp |= pen_is_grp(o->penalty[k]);
    }
    if (p < 2) { set_error("x must have at least two columns"); return OEMGPU_ERR_ARG; }                 // ref R/oem.R:226-229
    if (nl_of(o) < 1) { set_error("nlambda must be a positive integer"); return OEMGPU_ERR_ARG; }       // ref R/oem.R:361-364
    if (!(o->lambda_user && o->nlambda_user > 0) && !(o->lambda_min_ratio > 0.0 && o->lambda_min_ratio < 1.0)) {
        set_error("lambda.min.ratio must be between 0 and 1"); return OEMGPU_ERR_ARG;                    // ref R/oem.R:356-359
    }
    if (o->maxit <= 0) { set_error("maxit and irls.maxit should be positive"); return OEMGPU_ERR_ARG; }  // ref R/oem.R:427-430
    if (o->tol < 0) { set_error("tol and irls.tol should be nonnegative"); return OEMGPU_ERR_ARG; }
    if (!o->penalty_factor) { set_error("penalty.factor must have same length as number of columns in x"); return OEMGPU_ERR_ARG; }
    if (any_grp) {
        if (!o->groups || o->ngroupvars != ngroupvars_expected) {
            set_error("If any group penalty is used groups must have same length as number of columns in x");   // ref R/oem.R:288-290
            return OEMGPU_ERR_ARG;
        }
        if (!o->unique_groups || o->ngroups < 1) { set_error("unique_groups is empty"); return OEMGPU_ERR_ARG; }
        if (o->n_group_weights > 0 && o->n_group_weights != o->ngroups) {
            set_error("group.weights must have same length as the number of groups"); return OEMGPU_ERR_ARG;   // ref R/oem.R:313-315
        }
    }
    return 0;
}

// ---------------------------------------------------------------- host-built parameter blob
struct Blob {
    std::vector<char> h;
    size_t add(const void *src, size_t bytes)
    {
        size_t o = (h.size() + 15) / 16 * 16;
        h.resize(o + bytes);
        if (bytes) memcpy(h.data() + o, src, bytes);
        return o;
    }
};

}  // namespace oemgpu
#ifdef OEM_HOST_TIMING
#include <chrono>
namespace { struct HostT { double acc[8] = {0}; long n = 0; ~HostT() { if (n) fprintf(stderr, "host timing over %ld calls (us): entry->moments enqueued %.2f | ->path enqueued %.2f | ->copy enqueued %.2f | ->synced %.2f | ->unpacked/returned %.2f\n", n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n); } } g_ht;
  double g_tp[6]; inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); } }
#define HT(i) g_tp[i] = now_us()
#else
#define HT(i) do { } while (0)
#endif
namespace oemgpu {
// group bookkeeping of get_group_indexes (ref src/oem_dense.h:421-456).  q = dimension of beta;
// nscan = number of leading beta positions whose group entry is looked at (oemBig: nvars of nvars+1, quirk Q17).
struct Groups {
    std::vector<int> gid, gstart, gidx, gzero;
    std::vector<double> gw;
};
void build_groups(const oemgpu_opts *o, int q, int nscan, Groups &G)
{
    G.gid.assign(q, -1);
    G.gstart.assign(o->ngroups + 1, 0);
    G.gzero.assign(o->ngroups > 0 ? o->ngroups : 1, 0);
    G.gw.assign(o->ngroups > 0 ? o->ngroups : 1, 0.0);
    G.gidx.clear();
    // the reference matches every variable against every unique value (ref src/oem_dense.h:421-456): ngroups x nvars comparisons -- 2.5e9
    // of them for 25,000 groups over 100,000 columns, seconds of host time per call.  Same members in the same order from a value -> index
    // map and a counting sort, whenever the unique values ARE unique (R's sort(unique(groups)), R/oem.R:292); else the reference's loops.
    const int nv = nscan < o->ngroupvars ? nscan : o->ngroupvars;
    std::unordered_map<int, int> index;
    bool distinct = true;
    for (int g = 0; g < o->ngroups && distinct; ++g) distinct = index.emplace(o->unique_groups[g], g).second;
    if (distinct && o->ngroups > 0) {
        std::vector<int> cnt(o->ngroups + 1, 0), gv(nv > 0 ? nv : 1, -1);
        for (int v = 0; v < nv; ++v) {
            const auto it = index.find(o->groups[v]);
            if (it != index.end()) { gv[v] = it->second; ++cnt[it->second + 1]; }
        }
        for (int g = 0; g < o->ngroups; ++g) { cnt[g + 1] += cnt[g]; G.gstart[g] = cnt[g]; G.gzero[g] = o->unique_groups[g] == 0; }
        G.gidx.assign(cnt[o->ngroups], 0);
        std::vector<int> at(cnt.begin(), cnt.end() - 1);
        for (int v = 0; v < nv; ++v) if (gv[v] >= 0) { G.gidx[at[gv[v]]++] = v; G.gid[v] = gv[v]; }
    } else {
        for (int g = 0; g < o->ngroups; ++g) {
            G.gstart[g] = (int)G.gidx.size();
            for (int v = 0; v < nv; ++v)
                if (o->groups[v] == o->unique_groups[g]) { G.gidx.push_back(v); G.gid[v] = g; }
            G.gzero[g] = o->unique_groups[g] == 0;
        }
    }
    if (o->ngroups > 0) G.gstart[o->ngroups] = (int)G.gidx.size();
    for (int g = 0; g < o->ngroups; ++g)
        G.gw[g] = (o->n_group_weights < 1) ? std::sqrt((double)(G.gstart[g + 1] - G.gstart[g])) : o->group_weights[g];
    if (G.gidx.empty()) G.gidx.push_back(0);
}

// ---------------------------------------------------------------- the driver behind all entry points
// xx (q x q), xy (q), stats already on the device (in the workspace).  sem: OEMGPU_SEM_*, or 2 for oem.xtx.
enum { SEM_XTX = 2, SEM_SPARSE = 4 };        // OEMGPU_SEM_XVAL = 3 (oemgpu.h): oemBig's algebra with xval.oem's lambda_zero, groups and loss;
                                              // SEM_SPARSE: oemSparse (the intercept slot rescaled in place through `scale_factor`, groups = q entries, slot 0 the intercept's group 0)

// nbatch > 1: that many independent problems (instance b at xx + b * bstride, ... ; outputs of instance b at beta + b * npen * nl *
// rows, lambda_out / niter / loss + b * npen * nl, d_out[b]) solved by ONE launch, one workgroup (set) each; q <= SMALL_P_MAX only.
// Weighted oemDense with nobs <= nvars (weighted.hip): d comes from one matrix, the iteration runs on another and the loss belongs to
// the first again -- d handed over (launch-per-iteration Gram engine only), the loss as a pass of its own over (loss_xx, loss_xy, loss_stats).
struct PathExtras { double d_fixed = 0.0; const double *loss_xx = nullptr, *loss_xy = nullptr, *loss_stats = nullptr; };

// ---- PLAN: everything run_paths decides before it touches the device, as a pure host function of sizes and options (VERDICT r4:
// ten path engines chosen inside one 350-line function; the class of bug ADVICE r3 found -- a launch that rejects the workspace its
// own caller sized -- now has a CPU test: oemgpu_selftest_plan, tests/test_host_api.py sweeps it over q, penalty families and options).
struct PlanIn {
    int num_cu = 256;
    int p = 0, q = 0, sem = 0, intercept = 0, nbatch = 1;
    const oemgpu_opts *o = nullptr;
    bool has_scale = false;          // oem.xtx's scale.factor / oemSparse's in-place rescale of the intercept slot
    bool launches_only = false;      // d handed over (PathExtras::d_fixed): the launch-per-iteration Gram engine only
    bool loss_ext = false;           // the loss of ANOTHER Gram (PathExtras::loss_xx)
    int wide_n = 0;                  // > 0: the p >= n iteration through the standardised X itself (WideArgs::n rows), no Gram matrix
};
// Which XCD does workgroup i of a launch run on?  The one-XCD form of the cooperating engine (path_coop.hip) assumes "i mod 8" up to a
// rotation; this asks the hardware once per context (64 workgroups report HW_REG_XCC_ID), and every such launch proves its own
// placement again before it relies on it.
__global__ void xcc_probe_kernel(int *out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(id & 0xfu);
}

static bool ctx_xcd_layout_ok(oemgpu_ctx *c)
{
    if (c->xcd_layout == 0) {
        c->xcd_layout = -1;
        int *dv = nullptr, h[64];
        if (hipMalloc((void **)&dv, sizeof h) == hipSuccess) {          // (once per context, outside the workspace a call in progress has carved)
            hipLaunchKernelGGL(xcc_probe_kernel, dim3(64), dim3(64), 0, c->stream, dv);
            if (hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
                hipMemcpy(h, dv, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
                bool ok = true;
                unsigned seen = 0;
                for (int i = 0; i < 64; ++i) ok = ok && h[i] == h[i & 7];
                for (int i = 0; i < 8; ++i) seen |= 1u << (h[i] & 15);
                if (ok && __builtin_popcount(seen) == 8) c->xcd_layout = 1;
            }
            (void)hipFree(dv);
        }
    }
    return c->xcd_layout == 1;
}

struct PathPlan {
    int engine = OEMGPU_ENGINE_NONE;                 // of the first attempt
    bool any_grp = false, loss_on = false, loss_post = false;
    bool small = false, coop = false, symc = false, rowcoop = false, symcoop = false, pen_split = false;
    bool one_xcd = false;                            // path_coop.hip with all workgroups of an instance on ONE XCD (if the device's layout allows: run_paths)
    bool wcoop = false, wres = false, wstream = false, cut = false;
    int lan = 0, wg_n = 0, wsets = 1, wsg = 0, grcpw = 0;
    size_t work_d = 0, sym_off_d = 0;                // doubles of `work` per instance; where path_symcoop.hip's exchange area starts in it
    size_t out_stride = 0, out_bytes = 0, nb = 0, nk = 0;
    Groups G;
    std::vector<int> cst, grs, grg, grw;
    int grp_head = 0;                                // group operators in the head of the packed-triangle pairs (PathArgs::grp_head: 1 .. 3)
    std::vector<int> grun;                           // ... [2 q]
    std::vector<double> gwc;                         // ... [q]
    SymcoopPlan *symplan = nullptr;                  // (thread-local cache: a pure function of q, the CUs and the group runs)
    PathArgs ap;                                     // the scalar fields of the kernels' arguments (the eligibility rules read them)
    size_t frame_bytes() const { const int np = pen_split ? ap.npen : 1; Bump t; t.take(out_stride * ap.nbatch); t.take(work_d * sizeof(double) * ap.nbatch * np); return t.off; }
};

static int plan_paths(const PlanIn &in, PathPlan &P)
{
    const oemgpu_opts *o = in.o;
    const int q = in.q, p = in.p, sem = in.sem, nbatch = in.nbatch, num_cu = in.num_cu;
    const int nl = nl_of(o), npen = o->npen;
    const bool wide = in.wide_n > 0, launches_only = in.launches_only;
    for (int k = 0; k < npen; ++k) P.any_grp |= pen_is_grp(o->penalty[k]);
    const bool any_grp = P.any_grp;
    oemgpu_opts og = *o;
    if (!any_grp) og.ngroups = 0;
    Groups &G = P.G;
    build_groups(&og, q, sem == OEMGPU_SEM_BIG ? p : q, G);      // quirk Q17 is oemBig's alone (ref src/oem_xval_dense.h:636 and src/oem_sparse.h:465 scan all groups.size() slots)
    // p >= n on the cooperating engine with group penalties: where every group is a run of neighbouring columns the columns are dealt
    // to the workgroups in whole groups (<= 4 CW columns each), so that no group's norm needs a value from another workgroup
    std::vector<int> &cst = P.cst;
    if (wide && og.ngroups > 0 && !sw().OEM_WCOOP_NO_ALIGN.set) {
        const int cpg = path_wcoop_cpg(in.wide_n), gmax = path_wcoop_max_workgroups(in.wide_n, q);
        bool ok = cpg > 0 && gmax > 0;
        cst.push_back(0);
        int fill = 0;
        for (int j = 0; j < q && ok;) {
            int len = 1;
            const int g = G.gid[j];
            if (g >= 0) {
                len = G.gstart[g + 1] - G.gstart[g];
                for (int k = 0; k < len && ok; ++k) ok = G.gidx[G.gstart[g] + k] == j + k;      // a run of neighbouring columns starting here
            }
            if (len > cpg) ok = false;
            if (fill + len > cpg) { cst.push_back(j); fill = 0; }
            fill += len; j += len;
        }
        cst.push_back(q);
        if (!ok || (int)cst.size() - 1 > gmax || (int)cst.size() - 1 > num_cu * 3 / 4) cst.clear();
    }
    // p >= n beyond the persistent engines, group penalties: where every group is a run of neighbouring columns (<= WIDE_GRUN_MAX of
    // them) the runs are dealt to the workgroups of the fused group kernel (path_large.hip: wide_groups_kernel) in whole runs
    std::vector<int> &grs = P.grs, &grg = P.grg, &grw = P.grw;
    if (wide && og.ngroups > 0) {
        bool ok = true;
        int maxrun = 1;
        for (int j = 0; j < q && ok;) {
            const int g = G.gid[j];
            if (g < 0) { ok = false; break; }
            const int len = G.gstart[g + 1] - G.gstart[g];
            for (int k = 0; k < len && ok; ++k) ok = G.gidx[G.gstart[g] + k] == j + k;
            if (len > WIDE_GRUN_MAX) ok = false;
            grs.push_back(j); grg.push_back(g);
            if (len > maxrun) maxrun = len;
            j += len;
        }
        if (ok) {
            grs.push_back(q);
            const int Wt = wide_workgroups(in.wide_n, q), target = (q + Wt - 1) / Wt + maxrun - 1;      // (whole runs: never more than Wt workgroups)
            int fill = 0;
            grw.push_back(0);
            for (int r = 0; r + 1 < (int)grs.size(); ++r) {
                const int len = grs[r + 1] - grs[r];
                if (fill > 0 && fill + len > target) { grw.push_back(r); if (fill > P.grcpw) P.grcpw = fill; fill = 0; }
                fill += len;
            }
            grw.push_back((int)grs.size() - 1);
            if (fill > P.grcpw) P.grcpw = fill;
        } else { grs.clear(); grg.clear(); }
    }
    // 1024 < q <= 4096, element-wise penalties: the lower triangle of XX in the registers of <= 3/4 of the CUs (path_symcoop.hip);
    // the plan is a pure function of (q, CUs): kept from call to call (until the switches are read again)
    static thread_local SymcoopPlan symplan_plain, symplan_runs;
    static thread_local int symplan_q = 0, symplan_gmax = 0;
    static thread_local unsigned plans_gen = 0;
    static thread_local std::vector<int> runs_key;
    static thread_local int runs_q = 0, runs_gmax = 0;
    if (plans_gen != sw().generation) { symplan_q = 0; runs_q = 0; runs_key.clear(); plans_gen = sw().generation; }
    SymcoopPlan *symplan_p = &symplan_plain;
    if (!wide && !launches_only && nbatch == 1 && q > 1024 && q <= 4096 && !sw().OEM_NO_SYMCOOP.set && !sw().OEM_NO_COOP.set) {
        const int gmax = num_cu * 3 / 4 < WCOOP_GMAX ? num_cu * 3 / 4 : WCOOP_GMAX;
        if (any_grp) {
            // group operators: every group must be a run of neighbouring coordinates -- the owners' slices are cut there (and inside runs of more than 32)
            std::vector<int> rs;
            bool ok = true;
            for (int j = 0; j < q && ok;) {
                const int g = G.gid[j];
                int len = 1;
                if (g >= 0) {
                    len = G.gstart[g + 1] - G.gstart[g];
                    for (int k = 0; k < len && ok; ++k) ok = G.gidx[G.gstart[g] + k] == j + k;
                }
                rs.push_back(j);
                j += len;
            }
            rs.push_back(q);
            symplan_p = &symplan_runs;
            // (the partition of the runs over the owners is a dynamic programme of some milliseconds: kept while q, the CUs and the runs are the same)
            if (!ok) symplan_runs = SymcoopPlan();
            else if (runs_q != q || runs_gmax != gmax || runs_key != rs) {
                if (!symcoop_plan(q, gmax, symplan_runs, rs.data(), (int)rs.size() - 1)) symplan_runs = SymcoopPlan();
                runs_key = rs; runs_q = q; runs_gmax = gmax;
            }
            if (!ok) { runs_key.clear(); runs_q = 0; }
        } else if (symplan_q != q || symplan_gmax != gmax) {
            if (!symcoop_plan(q, gmax, symplan_plain)) symplan_plain = SymcoopPlan();
            symplan_q = q; symplan_gmax = gmax;
        }
        P.symc = !symplan_p->tab.empty();
    }
    P.symplan = symplan_p;
    const SymcoopPlan &symplan = *symplan_p;

    // ---- outputs region (device) : beta | lambda | loss | d[2] | niter | stats copy
    P.nb = (size_t)npen * nl * q; P.nk = (size_t)npen * nl;
    const size_t out_doubles = P.nb + 2 * P.nk + D_OUT_LEN + (size_t)stats_len(p);
    P.out_bytes = out_doubles * sizeof(double) + P.nk * sizeof(int32_t);
    P.out_stride = (P.out_bytes + 255) / 256 * 256;
    P.loss_on = (sem == OEMGPU_SEM_DENSE || sem == OEMGPU_SEM_XVAL || sem == SEM_SPARSE) && o->compute_loss != 0;
    // several instances (xval.oem's K + 1 fits) on the cooperating engine: only if all their workgroup sets are resident at once
    // wide: the p >= n iteration through the standardised X itself (there is no Gram matrix)
    P.coop = !wide && !launches_only && path_coop_eligible(q, in.has_scale, P.loss_on && !(in.has_scale && nbatch == 1), og.ngroups, nbatch) &&
             (nbatch == 1 || path_coop_workgroups(q) * nbatch <= num_cu * 3 / 4);
    P.small = !wide && !launches_only && q <= SMALL_P_MAX && !P.coop;
    if (nbatch > 1 && !P.small && !P.coop) { set_error("internal: batched paths need p <= %d", SMALL_P_MAX); return OEMGPU_ERR_INTERNAL; }
    // Lanczos step cap: q up to 288 (the whole Krylov space: the recurrence stops by itself when the top Ritz value has settled, and
    // a spectrum that needs more than 128 steps gets them -- ADVICE r1); the large-p engines keep their own caps (256 / 512)
    // (more steps than rows: without re-orthogonalisation a clustered spectrum does not exhaust the Krylov space in q steps -- a 9 x 9
    // standardised sparse Gram was 1.1e-7 short after 9 -- but the top Ritz value keeps converging; the stagnation test ends it)
    int lan = 2 * q < 32 ? 32 : (2 * q < 288 ? 2 * q : 288);
    if (sw().OEMGPU_LANCZOS_CAP.set) { const int k = (int)sw().OEMGPU_LANCZOS_CAP.num; if (k >= 2 && k < lan) lan = k; }     // test knob: reach the cap
    P.lan = lan;
    size_t work_d = P.small ? path_small_xchg_bytes() / 8 : path_large_work_doubles(q, lan);
    if (P.coop && work_d < path_coop_xchg_bytes() / 8) work_d = path_coop_xchg_bytes() / 8;
    P.sym_off_d = (work_d + 31) / 32 * 32;             // the exchange area of path_symcoop.hip behind the launch-per-iteration engines' workspace (the fallback needs both)
    if (P.symc) work_d = P.sym_off_d + (symcoop_work_bytes(symplan) + 7) / 8;
    P.work_d = work_d;
    // one workgroup (set) per penalty: they are independent cold starts (the cooperating sets must all be resident: <= half the CUs)
    P.pen_split = npen > 1 && (P.small || (P.coop && path_coop_workgroups(q) * npen * nbatch <= num_cu / 2));
    // q <= 512: the <= 16 cooperating workgroups of an instance fit ONE XCD (an eighth of the CUs), where an exchange through the local L2
    // costs 0.65 us instead of 1.05 (path_coop.hip: LOCAL); instance y goes to XCD (base + y) mod 8, so the instances that share an XCD
    // must fit it together
    {
        const int ninst = nbatch * (P.pen_split ? npen : 1), per_xcd = (ninst + 7) / 8;
        P.one_xcd = P.coop && q <= 512 && num_cu >= 64 && num_cu % 8 == 0 && !sw().OEM_NO_ONE_XCD.set && path_coop_workgroups(q) * per_xcd <= num_cu / 8;
    }

    // ---- the scalar part of the kernels' arguments
    PathArgs &a = P.ap;
    memset(&a, 0, sizeof a);
    a.p = q; a.npen = npen; a.nl = nl; a.user_lambda = o->lambda_user && o->nlambda_user > 0; a.maxit = o->maxit;
    a.accelerate = (sem == OEMGPU_SEM_DENSE) ? (o->accelerate != 0) : 0;           // only oemDense accelerates (quirk Q12)
    a.compute_loss = P.loss_on ? 1 : 0;
    // oemSparse with an intercept beyond the single-workgroup kernels: the engines rescale the member in place before the product their
    // loss would come from, so the loss is a pass of its own behind them (sparse.hip: gram_loss_kernel)
    // (weighted oemDense, nobs <= nvars: the loss of ANOTHER Gram, see PathExtras)
    P.loss_post = (P.loss_on && in.loss_ext) || (P.loss_on && in.has_scale && !P.small && !wide && nbatch == 1);
    if (P.loss_post) a.compute_loss = 0;
    a.ngroups = og.ngroups; a.lanczos_steps = lan; a.yscale = (sem == OEMGPU_SEM_DENSE);
    const bool biglike = sem == OEMGPU_SEM_BIG || sem == OEMGPU_SEM_XVAL || sem == SEM_SPARSE;
    a.lmax_from = (sem == OEMGPU_SEM_XVAL || sem == SEM_SPARSE) ? ((biglike && in.intercept) ? 1 : 0) : 0;              // ref src/oem_xval_dense.h:1025-1032
    a.alpha = o->alpha; a.gamma = o->gamma; a.tau = o->tau; a.tol = o->tol; a.lambda_min_ratio = o->lambda_min_ratio;
    a.sinv = in.has_scale ? reinterpret_cast<const double *>(8) : nullptr;      // (a marker for the eligibility rules; run_paths sets the pointer)
    a.pen_split = P.pen_split; a.pen_lo = 0; a.pen_hi = npen;
    a.nbatch = nbatch;

    // the launch engines at q > 1024 (the packed triangle): groups that are runs of <= 96 neighbouring coordinates
    // -- the group operators run in the head of the (head, product) pairs (path_large.hip), one launch fewer per iteration
    // and no single-workgroup update kernel (q = 8,192: 65 -> 52 us per iteration)
    if (!wide && any_grp && q > 1024 && nbatch == 1 && (int)G.gidx.size() <= q) {
        bool ok = true;
        int longest = 0;
        for (int g = 0; g < og.ngroups && ok; ++g) {
            const int m0 = G.gstart[g], len = G.gstart[g + 1] - m0;
            ok = len <= 96;
            if (len > longest) longest = len;
            for (int k = 0; k < len && ok; ++k) ok = G.gidx[m0 + k] == G.gidx[m0] + k && G.gid[G.gidx[m0 + k]] == g;
        }
        if (ok) {
            P.grp_head = longest <= 32 ? 1 : (longest <= 64 ? 2 : 3);      // blocks of 32 coordinates on either side of a head workgroup's own
            P.grun.assign(2 * (size_t)q, 0); P.gwc.assign(q, 0.0);
            for (int j = 0; j < q; ++j) {
                const int g = G.gid[j];
                if (g < 0) { P.grun[2 * j] = 1; P.grun[2 * j + 1] = 0; continue; }
                const int gs = G.gidx[G.gstart[g]], ge = gs + (G.gstart[g + 1] - G.gstart[g]);
                P.grun[2 * j] = gs; P.grun[2 * j + 1] = ge | (G.gzero[g] ? (1 << 30) : 0);
                P.gwc[j] = G.gw[g];
            }
        }
    }

    // 1024 < q <= 2048, element-wise penalties: the row-split form with ONE exchange per iteration, else the symmetric one
    P.rowcoop = P.symc && path_rowcoop_eligible(a, any_grp) && path_rowcoop_workgroups(q) <= num_cu * 3 / 4 &&
                symcoop_work_bytes(symplan) >= path_rowcoop_xchg_bytes();
    P.symcoop = P.symc && (P.rowcoop || path_symcoop_eligible(a, any_grp, symplan.runs));
    if (wide) {
        WideArgs wd;                                   // (sizes only: the eligibility rules read the layout and n)
        wd.xs = nullptr; wd.ys = nullptr; wd.lay = wide_layout(in.wide_n); wd.n = in.wide_n; wd.scratch = nullptr;
        // p >= n with Xs small enough for the register files of <= 192 CUs: one persistent launch (path_wcoop.hip)
        // (all of a set's workgroups must be resident at once: never more of them than three quarters of this device's CUs)
        const bool wcoop0 = path_wcoop_eligible(a, wd) && path_wcoop_workgroups(wd.n, q) <= num_cu * 3 / 4;
        // (... unless the few extra workgroups of the cut partition cost a whole penalty set its place beside the others)
        P.cut = wcoop0 && !cst.empty() && path_wcoop_sets(wd.n, q, npen, num_cu, (int)cst.size() - 1) >= 1 &&
                path_wcoop_sets(wd.n, q, npen, num_cu, (int)cst.size() - 1) >= path_wcoop_sets(wd.n, q, npen, num_cu, 0);
        P.wg_n = P.cut ? (int)cst.size() - 1 : (wcoop0 ? path_wcoop_workgroups(wd.n, q) : 0);
        P.wsets = wcoop0 ? path_wcoop_sets(wd.n, q, npen, num_cu, P.wg_n) : 1;
        const bool wres_forced = sw().OEM_WRES.set && path_wres_eligible(a, wd, num_cu - 8);      // (tests: also where the vector registers would do)
        P.wcoop = wcoop0 && P.wsets >= 1 && !wres_forced;          // (0 sets: the exchange scratch cannot hold this partition -- the launches of run_path_wide take the call)
        // ... where the vector registers alone cannot hold Xs: more column sets of every wave in the ACCUMULATOR file (path_wres_kernel: Xs up to
        // ~11 M entries; 500 x 20,000 on 209 CUs).  That is more than three quarters of the CUs: every CU but a few, and alone on the device.
        P.wres = !P.wcoop && path_wres_eligible(a, wd, num_cu - 8);
        // ... and where it does not fit: the same persistent launch re-reading its column tiles every iteration (path_wstream_kernel)
        P.wsg = num_cu * 3 / 4 < WCOOP_GMAX ? num_cu * 3 / 4 : WCOOP_GMAX;
        P.wstream = !P.wcoop && !P.wres && (path_wcoop_workgroups(wd.n, q) > WCOOP_GMAX || sw().OEM_WSTREAM.set) && path_wstream_eligible(a, wd, P.wsg);      // (where it pays: measured there)
    }
    P.engine = P.rowcoop ? OEMGPU_ENGINE_ROWCOOP : P.symcoop ? OEMGPU_ENGINE_SYMCOOP : P.wres ? OEMGPU_ENGINE_WRES : P.wcoop ? OEMGPU_ENGINE_WCOOP
               : P.wstream ? OEMGPU_ENGINE_WSTREAM : wide ? OEMGPU_ENGINE_WLAUNCHES : P.small ? OEMGPU_ENGINE_ROWS : P.coop ? OEMGPU_ENGINE_COOP : OEMGPU_ENGINE_LAUNCHES;
    return 0;
}

size_t paths_ws_bytes(int p, int q, const oemgpu_opts *o, int nbatch = 1);

// ---- scattered groups at 1024 < q <= 4096: the register-resident engine (path_symcoop.hip) cuts the owners' slices at group boundaries (and
// inside runs of more than 32), so it takes group penalties only where every group is a RUN of neighbouring coordinates; other layouts went to the
// launch-per-iteration engines (q = 3,000: 22 against 6-8 us per iteration -- the reference has no such cliff between group layouts, ref
// src/oem_dense.h:421-456).  The OEM iteration does not care where a coordinate sits: u = A beta + XY, coordinate- / group-wise operators, a stop
// rule over all coordinates, d an eigenvalue.  So groups that are not runs are MADE runs: coordinates reordered group by group (groups in
// the order of their first member, members in their own order -- the order the reference sums their squares in), XX, XY, the column
// constants, penalty factors and scale factors permuted alike, the path solved there, the coefficients put back.  One gather of q^2
// doubles per call.
__global__ __launch_bounds__(256) void permute_sym_kernel(const double *__restrict__ xx, const double *__restrict__ xy, const double *__restrict__ st, int q, int p_stats,
                                                           const int *__restrict__ perm, double *__restrict__ xx2, double *__restrict__ xy2, double *__restrict__ st2)
{
    const int j = blockIdx.x, pj = perm[j];
    for (int i = threadIdx.x; i < q; i += blockDim.x) xx2[(size_t)j * q + i] = xx[(size_t)pj * q + perm[i]];
    if (threadIdx.x == 0) {
        xy2[j] = xy[pj];
        if (j < p_stats) { st2[4 + j] = st[4 + pj]; st2[4 + p_stats + j] = st[4 + p_stats + pj]; }      // meanX | scaleX (stats layout, common.hpp)
    }
    if (j == 0) {
        for (int k = threadIdx.x; k < 4; k += blockDim.x) st2[k] = st[k];
        for (int k = 4 + 2 * p_stats + threadIdx.x; k < stats_len(p_stats); k += blockDim.x) st2[k] = st[k];
    }
}

// new position -> old position, or empty when the groups need no reordering / cannot be made runs
static std::vector<int> group_run_permutation(const oemgpu_opts *o, int q, int max_len)
{
    std::vector<int> perm;
    if (o->ngroups <= 0 || o->ngroupvars != q) return perm;
    oemgpu_opts og = *o;
    Groups G;
    build_groups(&og, q, q, G);
    bool runs = true;
    int longest = 0;
    for (int g = 0; g < o->ngroups; ++g) {
        const int len = G.gstart[g + 1] - G.gstart[g];
        if (len > longest) longest = len;
        for (int k = 1; k < len && runs; ++k) runs = G.gidx[G.gstart[g] + k] == G.gidx[G.gstart[g]] + k;
    }
    // (groups of more than 32 members: the register engine sums their norms over several owners, path_symcoop.hip -- max_len = q;
    //  the launches' head form beyond 4096 takes runs of <= 96: max_len = 96, nothing to gain otherwise)
    if (runs || longest > max_len) return perm;
    perm.reserve(q);
    std::vector<char> done(o->ngroups, 0);
    for (int j = 0; j < q; ++j) {
        const int g = G.gid[j];
        if (g < 0) perm.push_back(j);
        else if (!done[g]) { done[g] = 1; for (int m = G.gstart[g]; m < G.gstart[g + 1]; ++m) perm.push_back(G.gidx[m]); }
    }
    if ((int)perm.size() != q) perm.clear();            // (a variable listed in two groups: not ours to untangle)
    return perm;
}

static thread_local bool g_in_permuted_call = false;

int run_paths(oemgpu_ctx *c, Bump &B, const double *xx, const double *xy, const double *stats, int p, int q, int sem,
              int standardize, int intercept, const oemgpu_opts *o, const double *scale_factor,
              double *beta, double *lambda_out, int32_t *niter, double *loss, double *d_out,
              int nbatch = 1, size_t bstride = 0, bool shared_lmax = false, const WideArgs *wide = nullptr, const double *lmax_xy_dev = nullptr,
              const PathExtras *ex = nullptr)
{
    const bool launches_only = ex && ex->d_fixed > 0.0;
    const int nl = nl_of(o), npen = o->npen;
    const bool user = o->lambda_user && o->nlambda_user > 0;
    // (beyond 4096 the same reordering where it buys the launches their head form -- every group <= 96 members, PathArgs::grp_head)
    const bool perm_reg = q > 1024 && q <= 4096 && !sw().OEM_NO_SYMCOOP.set && !sw().OEM_NO_COOP.set && !sw().OEM_SYMCOOP_NO_GENERAL.set;
    // (... and below 4096 when the register engine is switched off: the launches are what runs)
    const bool perm_large = q > 1024 && !perm_reg && !sw().OEM_NO_SYM.set && !sw().OEM_NO_FUSED.set;
    if (!g_in_permuted_call && (perm_reg || perm_large) && nbatch == 1 && !wide && !ex && !lmax_xy_dev && xx && (sem == OEMGPU_SEM_DENSE || sem == SEM_XTX) &&
        o->ngroups > 0) {
        bool any_group_penalty = false;
        for (int k = 0; k < npen; ++k) any_group_penalty |= pen_is_grp(o->penalty[k]);
        const std::vector<int> perm = any_group_penalty ? group_run_permutation(o, q, perm_reg ? q : 96) : std::vector<int>();
        if (!perm.empty()) {
            // the permuted problem in a buffer of its own (xx2 | xy2 | stats2 | perm), the options with their per-coordinate arrays permuted
            const size_t nd = (size_t)q * q + q + stats_len(p) + 8;
            if (ctx_grow(c, &c->perm_buf, &c->perm_bytes, nd * sizeof(double) + (size_t)q * sizeof(int) + 256)) return OEMGPU_ERR_HIP;
            double *xx2 = (double *)c->perm_buf, *xy2 = xx2 + (size_t)q * q, *st2 = xy2 + q;
            int *perm_dev = (int *)(st2 + stats_len(p) + 8);
            OEM_HIP(hipMemcpyAsync(perm_dev, perm.data(), sizeof(int) * q, hipMemcpyHostToDevice, c->stream));
            OEM_HIP(hipStreamSynchronize(c->stream));                  // (perm is a pageable vector of this frame)
            hipLaunchKernelGGL(permute_sym_kernel, dim3(q), dim3(256), 0, c->stream, xx, xy, stats, q, p, perm_dev, xx2, xy2, st2);
            OEM_HIP(hipGetLastError());
            std::vector<int32_t> g2(q);
            std::vector<double> pf2(p), sf2(scale_factor ? q : 0);
            for (int j = 0; j < q; ++j) { g2[j] = o->groups[perm[j]]; pf2[j] = o->penalty_factor[perm[j]]; if (scale_factor) sf2[j] = scale_factor[perm[j]]; }
            oemgpu_opts o2 = *o;
            o2.groups = g2.data(); o2.penalty_factor = pf2.data();
            g_in_permuted_call = true;
            const int rc = run_paths(c, B, xx2, xy2, st2, p, q, sem, standardize, intercept, &o2, scale_factor ? sf2.data() : nullptr, beta, lambda_out, niter, loss, d_out,
                                     nbatch, bstride, shared_lmax, wide, lmax_xy_dev, ex);
            g_in_permuted_call = false;
            if (rc) return rc;
            // the coefficients back to where the caller's variables are (row 0 of oem()'s result is the intercept)
            const int rows = (sem == SEM_XTX) ? p : p + 1, off = rows - q;
            std::vector<double> row(rows);
            for (size_t ki = 0; ki < (size_t)npen * nl; ++ki) {
                double *ob = beta + ki * rows;
                memcpy(row.data(), ob, sizeof(double) * rows);
                for (int j = 0; j < q; ++j) ob[off + perm[j]] = row[off + j];
            }
            return 0;
        }
    }
    PlanIn pin;
    pin.num_cu = c->num_cu; pin.p = p; pin.q = q; pin.sem = sem; pin.intercept = intercept; pin.nbatch = nbatch; pin.o = o;
    pin.has_scale = scale_factor != nullptr; pin.launches_only = launches_only; pin.loss_ext = ex && ex->loss_xx; pin.wide_n = wide ? wide->n : 0;
    PathPlan P;
    if (int prc = plan_paths(pin, P)) return prc;
    const bool any_grp = P.any_grp, symc = P.symc, coop = P.coop, small = P.small, pen_split = P.pen_split, loss_post = P.loss_post;
    const bool loss_ext = P.loss_on && pin.loss_ext;
    const Groups &G = P.G;
    const std::vector<int> &cst = P.cst, &grs = P.grs, &grg = P.grg, &grw = P.grw;
    const int grcpw = P.grcpw;
    SymcoopPlan &symplan = *P.symplan;
    const size_t nb = P.nb, nk = P.nk, out_bytes = P.out_bytes, out_stride = P.out_stride, work_d = P.work_d, sym_off_d = P.sym_off_d;
    (void)any_grp;

    // ---- parameter blob
    Blob bl;
    std::vector<double> pf(q, 0.0), sinv;
    const bool biglike = sem == OEMGPU_SEM_BIG || sem == OEMGPU_SEM_XVAL || sem == SEM_SPARSE;
    const int off = (biglike && intercept) ? 1 : 0;       // leading 0 for the intercept, ref src/oem_big.cpp:105-113, src/oem_xval_dense.cpp:139-146
    for (int j = 0; j < p; ++j) pf[j + off] = o->penalty_factor[j];
    if (scale_factor) { sinv.resize(q); for (int j = 0; j < q; ++j) sinv[j] = 1.0 / scale_factor[j]; }
    const size_t o_pen = bl.add(o->penalty, sizeof(int32_t) * npen);
    const size_t o_lam = user ? bl.add(o->lambda_user, sizeof(double) * (size_t)npen * nl) : 0;
    const size_t o_pf = bl.add(pf.data(), sizeof(double) * q);
    const size_t o_sinv = scale_factor ? bl.add(sinv.data(), sizeof(double) * q) : 0;
    const size_t o_gid = bl.add(G.gid.data(), sizeof(int) * q);
    const size_t o_gst = bl.add(G.gstart.data(), sizeof(int) * G.gstart.size());
    const size_t o_gix = bl.add(G.gidx.data(), sizeof(int) * G.gidx.size());
    const size_t o_gz = bl.add(G.gzero.data(), sizeof(int) * G.gzero.size());
    const size_t o_gw = bl.add(G.gw.data(), sizeof(double) * G.gw.size());
    const size_t o_cst = cst.empty() ? 0 : bl.add(cst.data(), sizeof(int) * cst.size());
    const size_t o_grs = grw.empty() ? 0 : bl.add(grs.data(), sizeof(int) * grs.size());
    const size_t o_grg = grw.empty() ? 0 : bl.add(grg.data(), sizeof(int) * grg.size());
    const size_t o_grw = grw.empty() ? 0 : bl.add(grw.data(), sizeof(int) * grw.size());
    const size_t o_symp = symc ? bl.add(symplan.tab.data(), sizeof(int) * symplan.tab.size()) : 0;
    const size_t o_grun = P.grp_head ? bl.add(P.grun.data(), sizeof(int) * P.grun.size()) : 0;
    const size_t o_gwc = P.grp_head ? bl.add(P.gwc.data(), sizeof(double) * P.gwc.size()) : 0;

    // the outputs come first: when the caller's frame ends with `stats` (both callers), stats | outputs is one
    // contiguous range and one device-to-host copy returns both
    const size_t a_out = B.take(out_stride * nbatch), a_blob = B.take(bl.h.size()),
                 a_work = B.take(work_d * sizeof(double) * nbatch * (pen_split ? npen : 1));
    // the workspace may be re-allocated by ctx_reserve: xx/xy/stats are offsets into it, so recompute after
    const size_t off_xx = xx ? (const char *)xx - c->ws : 0, off_xy = (const char *)xy - c->ws, off_st = (const char *)stats - c->ws;
    if (B.off > c->ws_bytes) { set_error("internal: workspace under-reserved (%zu > %zu)", B.off, c->ws_bytes); return OEMGPU_ERR_INTERNAL; }
    if (xx) xx = (const double *)(c->ws + off_xx);
    xy = (const double *)(c->ws + off_xy); stats = (const double *)(c->ws + off_st);
    (void)a_blob;                                     // (the frame keeps its slot; the blob itself lives in a buffer of its own)
    // its own grow-only buffer, not the workspace: the Gram partials of the NEXT call overlay this frame, and a blob that survives
    // between calls is what lets run_paths skip an identical upload
    if (bl.h.size() + 256 > c->blob_bytes) c->blob_dev = nullptr;      // a re-allocation may land on the old address
    if (ctx_grow(c, &c->blob_buf, &c->blob_bytes, bl.h.size() + 256)) return OEMGPU_ERR_HIP;
    char *dblob = c->blob_buf;
    double *dout = (double *)(c->ws + a_out);
    if (ctx_pinned_in(c, bl.h.size())) return OEMGPU_ERR_HIP;       // the previous call ended with a stream sync: the buffer is free
    // Repeated solves with the same options (a lambda path re-fitted on new data, bench.py's loop) find the parameter blob already on
    // the device: same bytes at the same workspace address as the last upload of this context => no copy (a 2 KB H2D copy is a
    // 4 us node plus a boundary on a 570 us chain).
    if (!(c->blob_dev == dblob && c->blob_len == bl.h.size() && memcmp(c->pinned_in, bl.h.data(), bl.h.size()) == 0)) {
        memcpy(c->pinned_in, bl.h.data(), bl.h.size());
        OEM_HIP(hipMemcpyAsync(dblob, c->pinned_in, bl.h.size(), hipMemcpyHostToDevice, c->stream));
        c->blob_dev = dblob; c->blob_len = bl.h.size();
    }
    // Every output the host reads below is written by the path kernels, so the region is not cleared.  OEM_POISON_OUT=1
    // fills it with NaN bit patterns first: the GPU suite run that way proves nothing depends on stale contents.
    const bool poison = sw().OEM_POISON_OUT.set;
    if (poison) { OEM_HIP(hipMemsetAsync(dout, 0xFF, out_stride * nbatch, c->stream)); c->blob_dev = nullptr; }

    WideArgs wide_loc;
    if (wide && !grw.empty()) {
        wide_loc = *wide;
        wide_loc.grun_start = (const int *)(dblob + o_grs); wide_loc.grun_gid = (const int *)(dblob + o_grg); wide_loc.grun_wg = (const int *)(dblob + o_grw);
        wide_loc.grun_W = (int)grw.size() - 1; wide_loc.grun_cpw = grcpw;
        wide = &wide_loc;
    }
    PathArgs a = P.ap;                                   // the scalars are the plan's; the pointers:
    a.d_fixed = launches_only ? ex->d_fixed : 0.0;
    a.xx = xx; a.xy = xy; a.stats = stats;
    a.penalty = (const int *)(dblob + o_pen);
    a.lambda_user = user ? (const double *)(dblob + o_lam) : nullptr;
    a.pf = (const double *)(dblob + o_pf);
    a.sinv = scale_factor ? (const double *)(dblob + o_sinv) : nullptr;
    a.gid = (const int *)(dblob + o_gid); a.gstart = (const int *)(dblob + o_gst); a.gidx = (const int *)(dblob + o_gix);
    a.gzero = (const int *)(dblob + o_gz); a.gw = (const double *)(dblob + o_gw);
    a.grp_head = P.grp_head;
    a.grun = P.grp_head ? (const int *)(dblob + o_grun) : nullptr; a.gwc = P.grp_head ? (const double *)(dblob + o_gwc) : nullptr;
    a.beta = dout; a.lambda_out = dout + nb; a.loss = a.lambda_out + nk; a.d_out = a.loss + nk;
    double *dstats = a.d_out + D_OUT_LEN;
    a.niter = (int *)(dstats + stats_len(p));
    a.work = (double *)(c->ws + a_work);
    a.lmax_xy = (nbatch > 1 && shared_lmax) ? xy : lmax_xy_dev;    // instance 0's X'Y; or the caller's (big.oem with p >= n: the scaled X'y)
    a.bs_xx = a.bs_xy = a.bs_stats = (long long)bstride; a.bs_out = (long long)out_stride; a.bs_work = (long long)work_d;

    const size_t st_gap = (size_t)((const char *)dout - (const char *)stats);
    // Results straight into pinned host memory (row-split kernel, one instance): that kernel only ever STORES to its outputs, so they
    // may live in host memory the device writes over PCIe while it runs (a few hundred bytes per lambda); the device-to-host copy
    // node behind the kernel -- a launch boundary and a DMA set-up with the GPU idle -- disappears.  OEM_NO_ZERO_COPY=1: the copy.
    a.stats_out = nullptr; a.stats_n = 0;
    const bool zero_copy = small && nbatch == 1 && !poison && path_small_takes_rows(a) && !sw().OEM_NO_ZERO_COPY.set;
    const bool joined = !zero_copy && nbatch == 1 && (const char *)stats < (const char *)dout && st_gap == ((size_t)stats_len(p) * 8 + 255) / 256 * 256;
    const size_t back_bytes = nbatch > 1 ? out_stride * nbatch : out_bytes + (joined ? st_gap : 0);
    if (ctx_pinned(c, back_bytes > 16384 ? back_bytes : 16384)) return OEMGPU_ERR_HIP;
    if (zero_copy) {
        double *hout = (double *)c->pinned;
        a.beta = hout; a.lambda_out = hout + nb; a.loss = a.lambda_out + nk; a.d_out = a.loss + nk;
        a.stats_out = a.d_out + D_OUT_LEN; a.stats_n = stats_len(p);
        a.niter = (int *)(a.stats_out + stats_len(p));
    }
    const bool rowcoop = P.rowcoop, symcoop = P.symcoop, wcoop = P.wcoop, wres = P.wres, wstream = P.wstream;
    const int wg_n = P.wg_n, wsets = P.wsets, wsg = P.wsg;
    const int *wcst = P.cut ? (const int *)(dblob + o_cst) : nullptr;
    // The persistent engines spin on their partners: every workgroup of every such kernel in flight must be resident at once, so
    // concurrent callers queue for CU slots (held until the stream has been synchronised below)
    CoopSlots slots;
    // (one XCD: an instance's <= 16 workgroups compete for the 32 CUs of ITS XCD -- booked per XCD, the first XCD chosen where the load is lowest)
    // (a reload of the switches -- tests -- also forgets a timed-out engine's back-off and a refused placement)
    if (c->sw_generation != sw().generation) { c->sw_generation = sw().generation; c->persistent_backoff = 0; c->persistent_skip = 0; if (c->xcd_layout < 0) c->xcd_layout = 0; }
    bool local = P.one_xcd && ctx_xcd_layout_ok(c);
    const int coop_ninst = (pen_split ? npen : 1) * nbatch;
    int local_base = 0;
    if (coop) {
        if (local) local_base = slots.take_local(c->device, path_coop_workgroups(q), coop_ninst, c->num_cu, c->num_cu * 3 / 4);
        else slots.take(c->device, path_coop_workgroups(q) * coop_ninst, c->num_cu * 3 / 4);
    }
    c->last_placement = 0;
    if (symcoop) slots.take(c->device, rowcoop ? path_rowcoop_workgroups(q) : symplan.G, c->num_cu * 3 / 4);
    if (wcoop) slots.take(c->device, wg_n * wsets, c->num_cu * 3 / 4);
    if (wres) slots.take(c->device, path_wres_workgroups(wide->n, q) > c->num_cu * 3 / 4 ? c->num_cu : path_wres_workgroups(wide->n, q), c->num_cu * 3 / 4);
    if (wstream) slots.take(c->device, wsg, c->num_cu * 3 / 4);
    // The persistent engines (p >= n; 208 < p <= 1024) need all their workgroups resident at once.  If somebody else holds the CUs (another process on a
    // shared GPU) their exchanges time out after about a second and poison the result: the call is then made again on the
    // launch-per-iteration engines, which wait for nobody.
    // A persistent launch runs the whole penalty x lambda path (config 4: 23 ms; p >= n at maxit: seconds) and the reference polls for user
    // interrupts every third lambda (ref src/oem_dense.cpp:235-238): with an interrupt callback the kernel gets an abort word in
    // host-coherent memory, and the host waits for the launch by polling the stream AND the callback (on the calling thread).
    const bool abortable = o->interrupt != nullptr && (wcoop || wres || wstream || symcoop || coop);
    if (abortable) {
        if (!c->abort_host) {
            OEM_HIP(hipHostMalloc((void **)&c->abort_host, 64, hipHostMallocMapped)); ++g_alloc_count;
            OEM_HIP(hipHostGetDevicePointer((void **)&c->abort_dev, c->abort_host, 0));
        }
        __atomic_store_n(c->abort_host, 0, __ATOMIC_SEQ_CST);
    }
    const bool any_persistent = wcoop || wres || wstream || symcoop || (coop && nbatch == 1);
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // (ctx.hpp: a caller on a shared GPU does not pay the second of a timeout on every call -- VERDICT r4)
    const bool skip_persistent = any_persistent && c->persistent_skip > 0 && now_s() < c->persistent_skip_until;
    if (skip_persistent) --c->persistent_skip;
    for (int attempt = skip_persistent ? 1 : 0;; ++attempt) {
        const bool persistent = any_persistent && attempt == 0;
        a.abort_word = (abortable && (persistent || (coop && attempt == 0))) ? c->abort_dev : nullptr;
        a.one_xcd = (coop && attempt == 0 && local) ? (sw().OEM_FAKE_XCD_MISMATCH.set ? 2 : 1) : 0;
        a.xcd_base = a.one_xcd ? local_base : 0;          // (CoopSlots::take_local: where this device's XCDs are emptiest)
        {
            Timer t(c, OEMGPU_T_EIGPATH);
            PollScope poll(o);
            int rc;
            c->last_engine = (persistent && rowcoop) ? OEMGPU_ENGINE_ROWCOOP : (persistent && symcoop) ? OEMGPU_ENGINE_SYMCOOP : (persistent && wres) ? OEMGPU_ENGINE_WRES
                             : (persistent && wcoop) ? OEMGPU_ENGINE_WCOOP : (persistent && wstream) ? OEMGPU_ENGINE_WSTREAM : persistent ? OEMGPU_ENGINE_COOP
                             : wide ? OEMGPU_ENGINE_WLAUNCHES : small ? OEMGPU_ENGINE_ROWS : (coop && attempt == 0) ? OEMGPU_ENGINE_COOP : OEMGPU_ENGINE_LAUNCHES;
            if (persistent && rowcoop) rc = launch_path_rowcoop(c->stream, a, a.work + sym_off_d);
            else if (persistent && symcoop) rc = launch_path_symcoop(c->stream, a, symplan, (const int *)(dblob + o_symp), a.work + sym_off_d);
            else if (persistent && wres) rc = launch_path_wres(c->stream, a, *wide);
            else if (persistent) rc = wcoop ? launch_path_wcoop(c->stream, a, *wide, wsets, wcst, wg_n) : wstream ? launch_path_wstream(c->stream, a, *wide, wsg) : launch_path_coop(c->stream, a);
            else if (wide) rc = run_path_wide(c->stream, a, *wide, (double *)c->pinned);
            else if (small) rc = launch_path_small(c->stream, a);
            else if (coop && attempt == 0) rc = launch_path_coop(c->stream, a);                 // (several instances: no second engine takes them)
            else {
                PathArgs al = a;                                  // (after a timed-out cooperating launch: one instance, its penalties in turn)
                al.pen_split = 0; al.pen_lo = 0; al.pen_hi = npen;
                if (const size_t pkd = sympk_doubles(q)) {         // q > 4096: the products stream a packed copy of the lower triangle
                    if (ctx_grow(c, &c->pack_buf, &c->pack_bytes, pkd * sizeof(double))) return OEMGPU_ERR_HIP;
                    al.sympk = (double *)c->pack_buf;
                }
                rc = run_path_large(c->stream, al, (double *)c->pinned);
            }
            if (rc) return rc;
            if (loss_post && (rc = launch_gram_loss(c->stream, loss_ext ? ex->loss_xx : xx, loss_ext ? ex->loss_xy : xy, loss_ext ? ex->loss_stats : stats, q,
                                                    a.beta, a.sinv, a.niter, a.loss, (int)nk))) return rc;
        }
        HT(2);
        if (!zero_copy) {
            if (!joined && nbatch == 1) OEM_HIP(hipMemcpyAsync(dstats, stats, sizeof(double) * stats_len(p), hipMemcpyDeviceToDevice, c->stream));      // (stats outside the frame: the permuted problem)
            else if (!joined) OEM_HIP(hipMemcpy2DAsync(dstats, out_stride, stats, bstride * sizeof(double), sizeof(double) * stats_len(p), nbatch,
                                                       hipMemcpyDeviceToDevice, c->stream));
            OEM_HIP(hipMemcpyAsync(c->pinned, joined ? (const void *)stats : (const void *)dout, back_bytes, hipMemcpyDeviceToHost, c->stream));
        }
        HT(3);
        if (a.abort_word) {
            bool asked = false;
            const auto t0 = std::chrono::steady_clock::now();
            auto last = t0;
            for (;;) {
                const hipError_t qe = hipStreamQuery(c->stream);
                if (qe == hipSuccess) break;
                if (qe != hipErrorNotReady) OEM_HIP(qe);
                const auto now = std::chrono::steady_clock::now();
                if (!asked && now - last >= std::chrono::milliseconds(1)) {          // the callback at most once per millisecond
                    last = now;
                    if (o->interrupt(o->interrupt_arg)) { __atomic_store_n(c->abort_host, 1, __ATOMIC_SEQ_CST); asked = true; }
                }
                if (now - t0 < std::chrono::milliseconds(2)) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
            OEM_HIP(hipStreamSynchronize(c->stream));
            if (asked) { set_error("interrupted by the caller"); return OEMGPU_ERR_INTERRUPTED; }
        }
        OEM_HIP(hipStreamSynchronize(c->stream));
        HT(4);
        if (a.one_xcd) {
            // the launch's own proof of placement (path_coop.hip): poison value 2 in any instance = its workgroups were NOT on one XCD and
            // nothing was computed; this context does not ask again, and the call is made again with the exchange at device scope
            // Poison 3 = a partner never came to the proof: the instance's workgroups were not co-resident on ITS XCD (somebody this process does
            // not know of holds CUs there) -- and, for several instances (which have no second engine), a later exchange that timed out the same
            // way: the call is made again ONCE with the workgroups anywhere on the device; no back-off, the context keeps asking.
            bool refused = false, crowded = false;
            for (int bi = 0; bi < nbatch; ++bi) {
                const double *hdb = (const double *)((const char *)c->pinned + (joined ? st_gap : 0) + (size_t)bi * out_stride) + nb + 2 * nk;
                refused = refused || hdb[6] == 2.0;
                crowded = crowded || hdb[6] == 3.0 || (nbatch > 1 && hdb[6] == 1.0);
            }
            c->last_placement = refused ? 2 : crowded ? 3 : 1;
            if (refused) c->xcd_layout = -1;
            if (refused || crowded) {
                local = false; --attempt;
                slots.release();
                slots.take(c->device, path_coop_workgroups(q) * coop_ninst, c->num_cu * 3 / 4);
                continue;
            }
        }
        if (persistent) {
            const double *hd0 = (const double *)((const char *)c->pinned + (joined ? st_gap : 0)) + nb + 2 * nk;
            if (hd0[6] != 0.0) {
                ++c->persistent_fallbacks;
                c->persistent_backoff = c->persistent_backoff < 4 ? 4 : (c->persistent_backoff < 64 ? 2 * c->persistent_backoff : 64);
                c->persistent_skip = c->persistent_backoff; c->persistent_skip_until = now_s() + 30.0;
                continue;
            }
            c->persistent_backoff = 0; c->persistent_skip = 0;
        }
        break;
    }

    // ---- unpack (ref src/oem_dense.cpp:249-294, src/DataStd.h:269-293, src/oem_big.h:880-897, src/oem_big.cpp:213-220)
    const int flag = (standardize ? 1 : 0) + 2 * (intercept ? 1 : 0);
    const int rows = (sem == SEM_XTX) ? p : p + 1;
    for (int bi = 0; bi < nbatch; ++bi) {
        const double *hb = (const double *)((const char *)c->pinned + (joined ? st_gap : 0) + (size_t)bi * out_stride), *hl = hb + nb,
                     *hloss = hl + nk, *hd = hloss + nk, *hs = joined ? (const double *)c->pinned : hd + D_OUT_LEN;
        const int32_t *hn = (const int32_t *)(hd + D_OUT_LEN + stats_len(p));
        double *beta_b = beta + (size_t)bi * nk * rows, *lambda_b = lambda_out + (size_t)bi * nk, *loss_b = loss + (size_t)bi * nk;
        int32_t *niter_b = niter + (size_t)bi * nk;
        d_out[bi] = hd[0];
        // poison slot (common.hpp): cleared by the launchers of the multi-workgroup engines, written 0 by the others
        if (hd[6] != 0.0) { set_error("cooperating workgroups lost each other (exchange timeout)"); return OEMGPU_ERR_INTERNAL; }
        c->diag[0] = hd[2]; c->diag[1] = hd[3];
        if (bi == 0) { c->eig_steps = (int)hd[4]; c->eig_capped = hd[5] != 0.0; }
        c->shifted = hs[stats_shift_flag(p)] != 0.0;
        c->shift_advised = hs[stats_shift_flag(p) + 1] != 0.0;
        const double meany = hs[0], scaley = hs[1];
        const double *meanx = hs + 4, *scalex = hs + 4 + p;
        for (int k = 0; k < npen; ++k) {
            const int nlam = (o->penalty[k] == OEMGPU_OLS) ? 1 : nl;
            for (int i = 0; i < nl; ++i) {
                const size_t ki = (size_t)k * nl + i;
                lambda_b[ki] = hl[ki];
                niter_b[ki] = (i < nlam) ? hn[ki] : 0;
                loss_b[ki] = (i < nlam && (a.compute_loss || loss_post)) ? hloss[ki] : 1e99;
                double *ob = beta_b + ki * rows;
                const double *b = hb + ki * q;
                if (i >= nlam) { for (int j = 0; j < rows; ++j) ob[j] = 0.0; continue; }
                if (sem == SEM_XTX) {
                    for (int j = 0; j < p; ++j) ob[j] = b[j];
                } else if (biglike) {
                    ob[0] = intercept ? b[0] : 0.0;
                    for (int j = 0; j < p; ++j) ob[j + 1] = b[j + off] * (standardize ? scalex[j] : 1.0);
                } else {
                    // DataStd::recover (ref src/DataStd.h:269-293), one loop per flag: the tests inside the loop kept it scalar
                    // (12 us of host time per config-1 solve with the GPU idle; the same operations in the same order now)
                    double s = 0.0;
                    switch (flag) {
                    case 0: for (int j = 0; j < p; ++j) ob[j + 1] = b[j]; break;
                    case 1: for (int j = 0; j < p; ++j) ob[j + 1] = b[j] / scalex[j] * scaley; break;
                    case 2: for (int j = 0; j < p; ++j) { const double cf = b[j] * scaley; ob[j + 1] = cf; } break;
                    default: for (int j = 0; j < p; ++j) { const double cf = b[j] / scalex[j] * scaley; ob[j + 1] = cf; } break;
                    }
                    if (flag & 2) for (int j = 0; j < p; ++j) s += ob[j + 1] * meanx[j];
                    ob[0] = (flag & 2) ? meany - s : 0.0;
                }
            }
        }
    }
    return 0;
}

size_t paths_ws_bytes(int p, int q, const oemgpu_opts *o, int nbatch)
{
    const int nl = nl_of(o);
    if (nbatch > 1) return (size_t)nbatch * (paths_ws_bytes(p, q, o) + 1024) + (q >= path_coop_min_q(true) ? (size_t)nbatch * o->npen * path_coop_xchg_bytes() : 0);
    const size_t splits = (q <= 1024 && o->npen > 1) ? (size_t)o->npen : 1;
    size_t b = 0;
    b += (size_t)o->npen * 4 + (size_t)o->npen * nl * 8 + (size_t)q * (8 + 8 + 4) + (size_t)(o->ngroups + 2) * 16 +
         (size_t)(o->ngroupvars + q + 2) * 4 + (q > 1024 ? (size_t)q * 16 : 0) + 4096;      // (+ the per-coordinate group runs and weights of PathArgs::grp_head)
    b += ((size_t)o->npen * nl * (q + 3) + 4 + stats_len(p)) * 8 + 4096;
    // (from the cooperating engine's smallest q on -- 209, below the single-workgroup limit of 288 -- a call that is not small keeps the
    // launch-per-iteration engines' workspace as its fallback: oemgpu_selftest_plan found 209 <= q <= 288 with several penalties sized for
    // the small engine's exchange alone, 18 KB short at q = 209 x 8 penalties and covered only by ctx_reserve's slack)
    size_t wk = (q > SMALL_P_MAX || q >= path_coop_min_q(true)) ? path_large_work_doubles(q, 128) * 8 : path_small_xchg_bytes();
    if (wk < path_small_xchg_bytes()) wk = path_small_xchg_bytes();
    if (q >= path_coop_min_q(true) && q <= 1024 && wk < path_coop_xchg_bytes()) wk = path_coop_xchg_bytes();
    b += wk * splits + 4096;
    b += (size_t)q * 12 + 64;                                          // the runs of the fused group kernel (p >= n)
    if (q > 1024 && q <= 4096) b += symcoop_xchg_bytes_max(q) + (size_t)(80 + WCOOP_GMAX * 96) * 4 + (size_t)q * 8 + 1024;      // path_symcoop.hip's exchange area and plan (+ the fragment table of split groups)
    return b;
}

// out[i] = ((parts[0][i] + parts[1][i]) + parts[2][i]) + ...: the shards' moment buffers added in shard (= rank = device) order -- the order
// hoststream.hip adds them in, so the one-process-per-GPU form (all-gather + this kernel) and the in-library form return the same bits
__global__ __launch_bounds__(256) void sum_in_order_kernel(const double *__restrict__ parts, int nparts, size_t len, double *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
        double s = parts[i];
        for (int r = 1; r < nparts; ++r) s += parts[(size_t)r * len + i];
        out[i] = s;
    }
}

__global__ void accumulate_kernel(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// moments of one device-resident shard into `moments` (overwrite) ; tpart/vpart scratch from the workspace
int shard_moments(oemgpu_ctx *c, const GramPlan &pl, const double *x, int64_t n, int64_t ld, const double *y,
                  const double *sums, double *tpart, double *vpart, double *moments)
{
    {                       // no clear: moments_reduce_kernel writes all (p+2)^2 entries
        Timer t(c, OEMGPU_T_GRAMK);
        int rc = launch_gram(c->stream, pl, x, n, ld, y, sums, tpart, vpart);
        if (rc) return rc;
    }
    return launch_moments_reduce(c->stream, pl, tpart, vpart, moments);
}

// self-test aid: `blocks` workgroups that each take a whole CU (150 KB of LDS) and spin for `ms` milliseconds of device time
__global__ __launch_bounds__(256) void hold_cus_kernel(unsigned long long ticks, int *sink)
{
    extern __shared__ int hold_lds[];
    hold_lds[threadIdx.x] = (int)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (hold_lds[threadIdx.x] == -1 && sink) *sink = 1;
}
}  // namespace

// =====================================================================================================
#pragma GCC visibility push(default)
extern "C" {

void oemgpu_reload_switches(void) { sw_reload(); }

int oemgpu_selftest_plan(int32_t p, int32_t q, int32_t semantics, int32_t intercept, const oemgpu_opts *o, int32_t has_scale, int32_t nbatch,
                         int64_t wide_n, int32_t num_cu, int32_t *engine, int64_t *frame_bytes, int64_t *reserved_bytes,
                         int64_t *scratch_need_doubles, int64_t *scratch_have_doubles)
{
    if (!o || !engine || !frame_bytes || !reserved_bytes || !scratch_need_doubles || !scratch_have_doubles) { set_error("selftest_plan: NULL argument"); return OEMGPU_ERR_ARG; }
    if (int rc = check_opts(o, p, semantics == OEMGPU_SEM_DENSE || semantics == SEM_XTX ? p : q)) return rc;
    PlanIn in;
    in.num_cu = num_cu; in.p = p; in.q = q; in.sem = semantics; in.intercept = intercept; in.nbatch = nbatch; in.o = o;
    in.has_scale = has_scale != 0; in.wide_n = (int)wide_n;
    PathPlan P;
    if (int rc = plan_paths(in, P)) return rc;
    *engine = P.engine | (P.one_xcd ? 256 : 0) | (P.grp_head << 9);      // (bit 8: OEMGPU_ENGINE_COOP planned with every instance on one XCD; bits 9-10: PathArgs::grp_head, what the launches would do with the groups)
    // the frame run_paths carves (outputs | blob slot | work) against what every caller reserves for it (paths_ws_bytes); the blob
    // slot is bounded by the reservation's first two terms, which this adds back as the groups' and lambdas' actual sizes
    const int nl = nl_of(o);
    size_t blob = (size_t)o->npen * 4 + ((o->lambda_user && o->nlambda_user > 0) ? (size_t)o->npen * nl * 8 : 0) + (size_t)q * (8 + 8 + 4) +
                  P.G.gstart.size() * 4 + P.G.gidx.size() * 4 + P.G.gzero.size() * 4 + P.G.gw.size() * 8 + P.cst.size() * 4 +
                  (P.grw.empty() ? 0 : (P.grs.size() + P.grg.size() + P.grw.size()) * 4) + (P.symc ? P.symplan->tab.size() * 4 : 0) +
                  (P.grp_head ? P.grun.size() * 4 + P.gwc.size() * 8 : 0) + 18 * 16;
    *frame_bytes = (int64_t)(P.frame_bytes() + (blob + 255) / 256 * 256);
    *reserved_bytes = (int64_t)(paths_ws_bytes(p, q, o, nbatch) + 4096);
    // p >= n: the exchange buffers of the persistent engine chosen against the scratch every caller allocates (wide_scratch_doubles)
    *scratch_need_doubles = 0; *scratch_have_doubles = 0;
    if (wide_n > 0) {
        *scratch_have_doubles = (int64_t)wide_scratch_doubles((int)wide_n, q);
        if (P.wres) *scratch_need_doubles = (int64_t)path_wres_xchg_doubles((int)wide_n, q);
        else if (P.wstream) *scratch_need_doubles = (int64_t)path_wstream_xchg_doubles((int)wide_n);
        else if (P.wcoop) *scratch_need_doubles = (int64_t)path_wcoop_launch_doubles((int)wide_n, q, P.wg_n, P.wsets);
    }
    return 0;
}
const char *oemgpu_switch_names(void)
{
#define OEM_SW_NAME(name) #name " "
    return OEM_SWITCH_TABLE(OEM_SW_NAME);
#undef OEM_SW_NAME
}

const char *oemgpu_last_error(void) { return g_err; }
const char *oemgpu_version(void) { return "oemgpu 0.1 (gfx950)"; }

int oemgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

oemgpu_ctx *oemgpu_create(int32_t device, void *stream)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { set_error("no HIP device available (liboemgpu has no CPU fallback)"); return nullptr; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) { set_error("device %d out of range (%d devices)", device, n); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_error("hipGetDeviceProperties failed"); return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; liboemgpu is built for gfx950 only", device, prop.gcnArchName);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return nullptr; }
    oemgpu_ctx *c = new oemgpu_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->hbm_total = prop.totalGlobalMem;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return nullptr; }
        c->own_stream = true; ++g_alloc_count;
    }
    for (int i = 0; i < OEMGPU_NTIMERS; ++i) { c->ev_used[i] = false; c->ms[i] = 0.0; }
    return c;
}

void oemgpu_destroy(oemgpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->ws) (void)hipFree(c->ws);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_in) (void)hipHostFree(c->pinned_in);
    if (c->aux) (void)hipFree(c->aux);
    if (c->xres) (void)hipFree(c->xres);
    if (c->blob_buf) (void)hipFree(c->blob_buf);
    if (c->pack_buf) (void)hipFree(c->pack_buf);
    if (c->perm_buf) (void)hipFree(c->perm_buf);
    if (c->abort_host) (void)hipHostFree(c->abort_host);
    if (c->acc) (void)hipFree(c->acc);
    for (oemgpu_lane &l : c->lanes) {
        for (int k = 0; k < 2; ++k) {
            if (l.slot[k]) (void)hipHostFree(l.slot[k]);
            if (l.slot_ev[k]) (void)hipEventDestroy(l.slot_ev[k]);
            if (l.blk_ev[k]) (void)hipEventDestroy(l.blk_ev[k]);
        }
        if (l.s) (void)hipStreamDestroy(l.s);
    }
    for (int k = 0; k < 2; ++k) if (c->done_ev[k]) (void)hipEventDestroy(c->done_ev[k]);
    if (c->xfer_ev) (void)hipEventDestroy(c->xfer_ev);
    if (c->xfer_host) (void)hipHostFree(c->xfer_host);
    for (oemgpu_ctx *k : c->kids) oemgpu_destroy(k);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->ev_made) for (int i = 0; i < 2 * OEMGPU_NTIMERS; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void oemgpu_release_cache(void)
{
    std::vector<oemgpu_ctx *> drop;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        std::vector<oemgpu_ctx *> keep;
        for (oemgpu_ctx *c : g_cache) (c->busy ? keep : drop).push_back(c);
        g_cache.swap(keep);
    }
    for (oemgpu_ctx *c : drop) oemgpu_destroy(c);
}

int oemgpu_synchronize(oemgpu_ctx *c)
{
    if (!c) { set_error("ctx is NULL"); return OEMGPU_ERR_ARG; }
    OEM_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int oemgpu_selftest_gram_plan(int64_t n, int32_t p, int32_t num_cu, int64_t *out)
{
    if (n < 1 || p < 1 || num_cu < 1 || !out) { set_error("selftest_gram_plan: bad argument"); return OEMGPU_ERR_ARG; }
    const GramPlan pl = gram_plan(n, p, num_cu);
    const int64_t n8 = pl.n8, n6 = pl.n6, n4 = pl.n4;
    out[0] = pl.ntc; out[1] = n8; out[2] = n6; out[3] = pl.nchunk; out[4] = pl.steps;
    out[5] = 64 * (n8 * (n8 - 1) / 2) + 48 * n8 * n6 + 32 * n8 * n4 + 24 * n6 * n4 + 36 * n8 + 24 * n6 + 12 * n4;
    out[7] = n4;
    if (pl.wd) { out[1] = out[2] = out[7] = -1; out[5] = pl.wd == 4 ? 136 * pl.wd_units + 128 * pl.wd_units * (pl.wd_units - 1) : 80; }      // (one eight-wave workgroup per row chunk, the triangle of 16 tile columns: gram_wd.hip)
    out[6] = (int64_t)pl.ntc * (pl.ntc + 1) / 2;
    // the scratch sized for "any row count up to n" holds the plans of smaller row counts (folds, row tiles)
    const GramPlan bd = gram_plan_bound(n, p, num_cu);
    for (int64_t d = 1; d <= 4096 && n / d >= 1; d = d < 16 ? d + 1 : d * 2) {
        for (int64_t m : {n / d, n / d + 1, n - n / (d + 1)}) {
            if (m < 1 || m > n) continue;
            const GramPlan q = gram_plan(m, p, num_cu);
            if (q.tpart_doubles > bd.tpart_doubles || q.vpart_doubles > bd.vpart_doubles) {
                set_error("selftest_gram_plan: the plan of %lld rows (%d chunks) does not fit the bound for %lld (%d chunks)", (long long)m, q.nchunk, (long long)n, bd.nchunk);
                return OEMGPU_ERR_INTERNAL;
            }
        }
    }
    return 0;
}

int oemgpu_selftest_hold_cus(oemgpu_ctx *c, int32_t blocks, double ms)
{
    if (!c || blocks < 1 || blocks > 4096 || !(ms > 0.0) || ms > 30000.0) { set_error("hold_cus: bad argument"); return OEMGPU_ERR_ARG; }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const int lds = 150 * 1024;
    OEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hold_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(hold_cus_kernel, dim3(blocks), dim3(256), lds, c->stream, (unsigned long long)(ms * 1e5), (int *)nullptr);      // s_memrealtime: 100 MHz
    OEM_HIP(hipGetLastError());
    return 0;
}

int oemgpu_sum_in_order_dev(oemgpu_ctx *c, const double *parts_dev, int32_t nparts, int64_t len, double *out_dev)
{
    if (!c || !parts_dev || !out_dev || nparts < 1 || len < 1) { set_error("sum_in_order: bad argument"); return OEMGPU_ERR_ARG; }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const int blocks = (int)std::min<int64_t>((len + 255) / 256, 2048);
    hipLaunchKernelGGL(sum_in_order_kernel, dim3(blocks), dim3(256), 0, c->stream, parts_dev, (int)nparts, (size_t)len, out_dev);
    OEM_HIP(hipGetLastError());
    return 0;
}

int oemgpu_selftest_group_permutation(const oemgpu_opts *o, int32_t q, int32_t *perm)
{
    if (!o || !perm || q < 1) { set_error("selftest_group_permutation: bad argument"); return OEMGPU_ERR_ARG; }
    const std::vector<int> pm = group_run_permutation(o, q, q);
    for (size_t j = 0; j < pm.size(); ++j) perm[j] = pm[j];
    return (int)pm.size();
}

int oemgpu_selftest_symcoop_owners(int32_t q, int32_t num_cu, const int32_t *runs, int32_t nruns, int32_t *owner_c0, int32_t *owner_n, int32_t *frag,
                                   int32_t *nowners, int32_t *split)
{
    if (q <= 1024 || q > 4096 || num_cu < 8 || !runs || nruns < 1 || !owner_c0 || !owner_n || !frag || !nowners || !split || runs[0] != 0 || runs[nruns] != q) {
        set_error("selftest_symcoop_owners: bad argument (1024 < q <= 4096, runs[0] = 0, runs[nruns] = q)"); return OEMGPU_ERR_ARG;
    }
    const int gmax = num_cu * 3 / 4 < WCOOP_GMAX ? num_cu * 3 / 4 : WCOOP_GMAX;
    int G = 0, sp = 0;
    if (!symcoop_plan_owners(q, gmax, runs, nruns, owner_c0, owner_n, frag, &G, &sp)) { *nowners = 0; *split = 0; return 0; }      // (no plan: the launches take the call)
    *nowners = G; *split = sp;
    return 0;
}

int oemgpu_selftest_coop_slots(int32_t num_cu, int32_t W, int32_t ninst, int32_t calls, int32_t *bases, int32_t *peak)
{
    if (num_cu < 8 || num_cu % 8 || W < 1 || ninst < 1 || calls < 1 || calls > 64 || !bases || !peak) { set_error("selftest_coop_slots: bad argument"); return OEMGPU_ERR_ARG; }
    if ((long long)W * ninst * calls > num_cu * 3 / 4) { set_error("selftest_coop_slots: %d calls of %d x %d workgroups would wait for each other", calls, ninst, W); return OEMGPU_ERR_ARG; }
    std::vector<CoopSlots> held(calls);
    const int dev = COOP_MAX_DEV - 2;                    // (a device index no GPU of this process has: the book of a device of its own)
    *peak = 0;
    for (int k = 0; k < calls; ++k) bases[k] = held[k].take_local(dev, W, ninst, num_cu, num_cu * 3 / 4);
    { std::lock_guard<std::mutex> lk(g_coop_mu); for (int x = 0; x < 8; ++x) if (g_xcd_load[dev][x] > *peak) *peak = g_xcd_load[dev][x]; }
    return 0;
}

int oemgpu_selftest_sympk_gemv(oemgpu_ctx *c, const double *xx_dev, int32_t q, const double *vec_dev, double *out_dev, int32_t reps, double *us_per_product)
{
    if (!c || !xx_dev || !vec_dev || !out_dev || q <= 4096 || reps < 0 || reps > 100000) { set_error("selftest_sympk_gemv: bad argument (q > 4096)"); return OEMGPU_ERR_ARG; }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    if (ctx_grow(c, &c->pack_buf, &c->pack_bytes, sympk_doubles(q) * sizeof(double))) return OEMGPU_ERR_HIP;
    return sympk_gemv_probe(c->stream, xx_dev, q, (double *)c->pack_buf, vec_dev, out_dev, reps, us_per_product);
}

int oemgpu_set_timing(oemgpu_ctx *c, int32_t on)
{
    if (!c) { set_error("ctx is NULL"); return OEMGPU_ERR_ARG; }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    if (on && !c->ev_made) {
        for (int i = 0; i < 2 * OEMGPU_NTIMERS; ++i) OEM_HIP(hipEventCreate(&c->ev[i]));
        c->ev_made = true;
    }
    c->timing = on != 0;
    return 0;
}

int oemgpu_last_timings(oemgpu_ctx *c, double *ms)
{
    if (!c || !ms) { set_error("NULL argument"); return OEMGPU_ERR_ARG; }
    OEM_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < OEMGPU_NTIMERS; ++i) {
        ms[i] = 0.0;
        if (c->ev_made && c->ev_used[i]) {
            float f = 0.f;
            if (hipEventElapsedTime(&f, c->ev[2 * i], c->ev[2 * i + 1]) == hipSuccess) ms[i] = f;
        }
    }
    ms[OEMGPU_T_PATHCYC] = c->diag[0];
    ms[OEMGPU_T_PATHTICKS] = c->diag[1];
    return 0;
}

// ---------------------------------------------------------------------------------------------- staged interface
int oemgpu_shift_sums_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                          double *sums_dev)
{
    if (!c || !x_dev || !y_dev || !sums_dev || n < 1 || p < 1 || ld < n) { set_error("shift_sums: bad argument"); return OEMGPU_ERR_ARG; }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    Timer t(c, OEMGPU_T_SHIFT);
    return launch_shift_sums(c->stream, x_dev, n, ld, p, y_dev, sums_dev);
}

int oemgpu_moments_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                       const double *sums_dev, double *moments_dev)
{
    if (!c || !x_dev || !y_dev || !moments_dev || n < 1 || p < 1 || ld < n) { set_error("moments: bad argument"); return OEMGPU_ERR_ARG; }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const GramPlan pl = gram_plan(n, p, c->num_cu);
    Bump B;
    const size_t a_t = B.take(pl.tpart_doubles * 8), a_v = B.take(pl.vpart_doubles * 8);
    if (ctx_reserve(c, B.off)) return OEMGPU_ERR_HIP;
    Timer t(c, OEMGPU_T_MOMENTS);
    return shard_moments(c, pl, x_dev, n, ld, y_dev, sums_dev, (double *)(c->ws + a_t), (double *)(c->ws + a_v), moments_dev);
}

static int run_paths_parts(oemgpu_ctx *c, Bump B, const double *xx, const double *xy, const double *st, int p, int q, int sem, int standardize, int intercept,
                           const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d,
                           const WideArgs *wd, const double *lmax_xy);

// wpatch (device, or NULL): the constants of DataStd's WEIGHTED standardisation (weighted.hip).  The moments are then those of the
// standardised, sqrt(w)-scaled copy: finalize takes them as they are (flag 0) and the constants go into `stats` for the lambda
// grid (scaleY) and for recover(), which run under the caller's flags.
static int solve_moments_impl(oemgpu_ctx *c, const double *moments_dev, const double *sums_dev, int32_t p,
                              int32_t semantics, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                              double *beta, double *lambda_out, int32_t *niter, double *loss, double *d, const double *wpatch,
                              const PathExtras *ex = nullptr, const double *lmax_xy = nullptr, size_t extra_ws = 0, double *big_xy_std = nullptr);

int oemgpu_solve_moments_dev(oemgpu_ctx *c, const double *moments_dev, const double *sums_dev, int32_t p,
                             int32_t semantics, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                             double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    return solve_moments_impl(c, moments_dev, sums_dev, p, semantics, standardize, intercept, o, beta, lambda_out, niter, loss, d, nullptr);
}

static int solve_moments_impl(oemgpu_ctx *c, const double *moments_dev, const double *sums_dev, int32_t p,
                              int32_t semantics, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                              double *beta, double *lambda_out, int32_t *niter, double *loss, double *d, const double *wpatch,
                              const PathExtras *ex, const double *lmax_xy, size_t extra_ws, double *big_xy_std)
{
    if (!c || !moments_dev || !beta || !lambda_out || !niter || !loss || !d) { set_error("solve_moments: NULL argument"); return OEMGPU_ERR_ARG; }
    if (semantics != OEMGPU_SEM_DENSE && semantics != OEMGPU_SEM_BIG && semantics != OEMGPU_SEM_XVAL) { set_error("unknown semantics %d", semantics); return OEMGPU_ERR_ARG; }
    const int q = p + ((semantics != OEMGPU_SEM_DENSE && intercept) ? 1 : 0);
    int rc = check_opts(o, p, q);
    if (rc) return rc;
    if (semantics == OEMGPU_SEM_BIG && o->compute_loss) {
        set_error("compute.loss is not available for big.oem: the reference expression is ill-formed (src/oem_big.h:899-921)");
        return OEMGPU_ERR_UNSUPPORTED;
    }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    Bump B;
    const size_t a_xx = B.take((size_t)q * q * 8), a_xy = B.take((size_t)q * 8), a_st = B.take((size_t)stats_len(p) * 8);
    // moments/sums may live in the workspace (fit_dense_dev): keep them below this frame
    size_t base = 0;
    if ((const char *)moments_dev >= c->ws && (const char *)moments_dev < c->ws + c->ws_bytes)
        base = ((const char *)moments_dev - c->ws) + (size_t)oemgpu_moments_len(p) * 8;
    if (sums_dev && (const char *)sums_dev >= c->ws && (const char *)sums_dev < c->ws + c->ws_bytes) {
        size_t e = ((const char *)sums_dev - c->ws) + (size_t)oemgpu_sums_len(p) * 8;
        if (e > base) base = e;
    }
    base = (base + 255) / 256 * 256;
    const size_t need = base + B.off + paths_ws_bytes(p, q, o) + extra_ws + 4096;
    if (need > c->ws_bytes) {
        if (base != 0) { set_error("internal: workspace frame too small"); return OEMGPU_ERR_INTERNAL; }
        if (ctx_reserve(c, need)) return OEMGPU_ERR_HIP;
    }
    Bump B2; B2.off = base + B.off;
    double *xx = (double *)(c->ws + base + a_xx), *xy = (double *)(c->ws + base + a_xy), *st = (double *)(c->ws + base + a_st);
    {
        Timer t(c, OEMGPU_T_FINAL);
        // big_xy_std (big.oem / sparse x with nobs <= nvars on the Gram, fit_big_gram_dev): X'X / n and X'y / n of the data as they are,
        // the column scales only in `stats` (returned coefficients) and in big_xy_std = xy colsq_inv (lambda_zero)
        const bool plain = wpatch || big_xy_std;
        rc = launch_finalize(c->stream, moments_dev, sums_dev, p, semantics, plain ? 0 : standardize, plain ? 0 : intercept, xx, xy, st);
        if (!rc && wpatch) rc = launch_weighted_patch_stats(c->stream, wpatch, p, st);
        if (!rc && big_xy_std) { rc = launch_big_gram_scales(c->stream, xx, xy, p, standardize, st, big_xy_std); lmax_xy = big_xy_std; }
        if (rc) return rc;
    }
    if (!ex) return run_paths_parts(c, B2, xx, xy, st, p, q, semantics, standardize, intercept, o, beta, lambda_out, niter, loss, d, nullptr, lmax_xy);
    return run_paths(c, B2, xx, xy, st, p, q, semantics, standardize, intercept, o, nullptr, beta, lambda_out, niter, loss, d,
                     1, 0, false, nullptr, lmax_xy, ex);
}

// p >= n.  The reference switches to its two-GEMV iteration when nobs <= nvars (ref src/oem_dense.h:476-482); here that form is
// taken where it pays: the Gram form (same iteration, p^2 doubles read per iteration and held in memory) wins while p fits the
// register / cooperating engines (p <= 1024: ~1 us per iteration, no launch per iteration) and while 2 n p >= p^2; beyond, the
// wide engine reads 8 n p bytes per iteration and needs no p x p matrix (p = 20,000: 80 MB of Xs instead of 3.2 GB).
// OEM_WIDE=1 forces it wherever it can run (tests), OEM_NO_WIDE=1 switches it off.
static bool wide_pays(int64_t n, int32_t p, const oemgpu_opts *o)
{
    if (n > p || n > WIDE_MAX_N) return false;
    if (sw().OEM_NO_WIDE.set) return false;
    if (sw().OEM_WIDE.set) return true;
    // round 4: one element-wise penalty at 1024 < p <= 2048 -- the Gram form on the row-split engine (path_rowcoop_kernel: ONE exchange
    // per iteration, ~3.4 us) beats the persistent wide engine's two (500 x 2,000: 4.0 us).  Several penalties stay here: the wide
    // engine runs them side by side in workgroup sets of their own.
    if (p > 1024 && p <= 2048 && o && o->npen == 1 && !pen_is_grp(o->penalty[0]) && !o->accelerate && !o->compute_loss &&
        !sw().OEM_NO_ROWCOOP.set && !sw().OEM_NO_SYMCOOP.set && !sw().OEM_NO_COOP.set) return false;
    if (p <= 1024) return false;                  // (the register engines run the Gram form at 1-3 us per iteration)
    // where Xs fits the registers of the cooperating engine (path_wcoop.hip) that one launch beats everything else at these sizes
    // (n = 900, p = 1500: 5.6 against 11.6 us per iteration on the launch-per-iteration Gram engines)
    const int g = path_wcoop_workgroups((int)n, p);
    const bool wc = n <= 1024 && g >= 1 && g <= WCOOP_GMAX && !sw().OEM_NO_WCOOP.set;
    // round 4: up to p = 4096 the Gram lives in the register files of the chip (path_symcoop.hip: 5-7 us per iteration, no launch per
    // iteration) -- that beats the wide engine's launches (1,000 x 4,000: ~20 us per iteration) wherever the persistent wide engine
    // does not take the call
    if (p <= 4096 && !sw().OEM_NO_SYMCOOP.set && !sw().OEM_NO_COOP.set) return wc;
    if (2 * n < p) return true;
    // n <= p < 2n: the Gram is the smaller matrix
    return wc;
}

// p >= n, several penalties, some of them group penalties: one group penalty sends the WHOLE call to the engines that know group
// operators, and where the standardised X only fits the chip's registers WITH the accumulator file (path_wres_kernel: element-wise
// operators only) that is the launch-per-iteration engine re-reading Xs every iteration (500 x 20,000: 26-33 us per iteration
// against 7).  Penalties are independent cold starts (ref src/oem_dense.cpp:206-246), so the call is made in two parts there --
// the element-wise penalties, then the group penalties -- and the results go back into the caller's order.  (The eigenvalue step
// runs twice; both engines hold d to 1e-10 and the first part's is reported.)  OEM_NO_PENALTY_SPLIT=1: one call.
// The same for n > p with 1024 < p <= 2048: the element-wise penalties take the one-exchange row-split engine (path_rowcoop_kernel, ~3 us
// per iteration), the group penalties the symmetric engine's general form (~5).
static int run_paths_parts(oemgpu_ctx *c, Bump B, const double *xx, const double *xy, const double *st, int p, int q, int sem, int standardize, int intercept,
                           const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d,
                           const WideArgs *wd, const double *lmax_xy)
{
    std::vector<int> ie, ig;
    for (int k = 0; k < o->npen; ++k) (pen_is_grp(o->penalty[k]) ? ig : ie).push_back(k);
    bool split = !ie.empty() && !ig.empty() && !o->accelerate && !sw().OEM_NO_PENALTY_SPLIT.set;
    if (split && wd) {
        const int gw = path_wres_workgroups(wd->n, p);
        const bool wcoop_fits = path_wcoop_workgroups(wd->n, p) >= 1 && path_wcoop_workgroups(wd->n, p) <= (c->num_cu * 3 / 4 < WCOOP_GMAX ? c->num_cu * 3 / 4 : WCOOP_GMAX);
        split = !wcoop_fits && gw >= 1 && gw <= c->num_cu - 8 && path_wres_xchg_doubles(wd->n, p) > 0 && !sw().OEM_NO_WRES.set && !sw().OEM_NO_WCOOP.set;
    } else if (split)
        split = q > 1024 && q <= 2048 && sem == OEMGPU_SEM_DENSE && path_rowcoop_workgroups(q) <= c->num_cu * 3 / 4 &&
                !sw().OEM_NO_ROWCOOP.set && !sw().OEM_NO_SYMCOOP.set && !sw().OEM_NO_COOP.set;
    if (!split)
        return run_paths(c, B, xx, xy, st, p, q, sem, standardize, intercept, o, nullptr, beta, lambda_out, niter, loss, d, 1, 0, false, wd, lmax_xy);
    const int nl = nl_of(o), rows = p + 1;
    const bool user = o->lambda_user && o->nlambda_user > 0;
    bool first = true;
    for (const std::vector<int> *part : {&ie, &ig}) {
        const int m = (int)part->size();
        std::vector<int32_t> pen(m);
        std::vector<double> lam(user ? (size_t)m * nl : 0);
        for (int k = 0; k < m; ++k) {
            pen[k] = o->penalty[(*part)[k]];
            if (user) memcpy(&lam[(size_t)k * nl], o->lambda_user + (size_t)(*part)[k] * nl, sizeof(double) * nl);
        }
        oemgpu_opts os = *o;
        os.npen = m; os.penalty = pen.data(); os.lambda_user = user ? lam.data() : nullptr;
        std::vector<double> b((size_t)m * nl * rows), lo((size_t)m * nl), ls((size_t)m * nl);
        std::vector<int32_t> ni((size_t)m * nl);
        double dd = 0.0;
        Bump Bp = B;                                             // (each part takes the same frame behind xy / stats)
        const int rc = run_paths(c, Bp, xx, xy, st, p, q, sem, standardize, intercept, &os, nullptr, b.data(), lo.data(), ni.data(), ls.data(), &dd,
                                 1, 0, false, wd, lmax_xy);
        if (rc) return rc;
        for (int k = 0; k < m; ++k) {
            const size_t dst = (size_t)(*part)[k] * nl, src = (size_t)k * nl;
            memcpy(beta + dst * rows, &b[src * rows], sizeof(double) * (size_t)nl * rows);
            memcpy(lambda_out + dst, &lo[src], sizeof(double) * nl);
            memcpy(loss + dst, &ls[src], sizeof(double) * nl);
            memcpy(niter + dst, &ni[src], sizeof(int32_t) * nl);
        }
        if (first) *d = dd;
        first = false;
    }
    return 0;
}

static int fit_dense_wide_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                              int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                              double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const WideLayout lay = wide_layout(n);
    Bump X;
    const size_t a_xs = X.take(sizeof(double) * (size_t)lay.rows() * p), a_ys = X.take(sizeof(double) * (size_t)lay.rows()),
                 a_sc = X.take(sizeof(double) * wide_scratch_doubles((int)n, p));
    if (ctx_grow(c, &c->aux, &c->aux_bytes, X.off)) return OEMGPU_ERR_HIP;
    Bump B;
    const size_t a_xy = B.take((size_t)p * 8), a_st = B.take((size_t)stats_len(p) * 8);       // stats last: run_paths returns it with the outputs
    if (ctx_reserve(c, B.off + paths_ws_bytes(p, p, o) + 4096)) return OEMGPU_ERR_HIP;
    double *xs = (double *)(c->aux + a_xs), *ys = (double *)(c->aux + a_ys);
    double *xy = (double *)(c->ws + a_xy), *st = (double *)(c->ws + a_st);
    int rc;
    {
        Timer t(c, OEMGPU_T_MOMENTS);             // the stage that reads X: DataStd on the data (the standardised copy, X'Y / n)
        rc = launch_wide_standardize(c->stream, x_dev, n, ld, p, y_dev, standardize, intercept, lay, xs, ys, xy, st);
        if (rc) return rc;
    }
    WideArgs wd;
    wd.xs = xs; wd.ys = ys; wd.lay = lay; wd.n = (int)n; wd.scratch = (double *)(c->aux + a_sc);
    return run_paths_parts(c, B, nullptr, xy, st, p, p, OEMGPU_SEM_DENSE, standardize, intercept, o, beta, lambda_out, niter, loss, d, &wd, nullptr);
}

// oemSparse::get_loss with nobs <= nvars (ref src/oem_sparse.h:932-941): the residual of the RETURNED coefficients on the data as they are
static int big_wide_sparse_loss(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev, const oemgpu_opts *o,
                                const double *beta, const int32_t *niter, double *loss)
{
    const size_t nk = (size_t)o->npen * nl_of(o), nchunk = (size_t)((n + 2047) / 2048);
    Bump L;
    const size_t a_b = L.take(nk * (size_t)(p + 1) * 8), a_p = L.take(nk * nchunk * 8), a_l = L.take(nk * 8);
    if (ctx_reserve(c, L.off + 4096)) return OEMGPU_ERR_HIP;          // (the paths are done: their frame may be overlaid)
    double *bd = (double *)(c->ws + a_b), *ld_ = (double *)(c->ws + a_l);
    OEM_HIP(hipMemcpyAsync(bd, beta, nk * (size_t)(p + 1) * 8, hipMemcpyHostToDevice, c->stream));
    int rc;
    if ((rc = launch_resid_loss(c->stream, x_dev, n, ld, p, y_dev, bd, p + 1, (int)nk, (double *)(c->ws + a_p), ld_))) return rc;
    OEM_HIP(hipMemcpyAsync(loss, ld_, nk * 8, hipMemcpyDeviceToHost, c->stream));
    OEM_HIP(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < nk; ++k) if (niter[k] == 0) loss[k] = 1e99;
    return 0;
}

// big.oem / oem() on a sparse x with nobs <= nvars and NO intercept (ref src/oem_big.h:537-541, 568-584, 743-764, 880-897;
// src/oem_sparse.h:607-612, 638-647): the iteration u = X'(Y - X beta)/n + d beta on the data AS THEY ARE, d from X X'/n, no y scaling;
// with standardize only lambda_zero (max |x_j'y| colsq_inv_j / n) and the returned coefficients (beta colsq_inv) carry the column
// scales -- what the reference does.  The wide engine on the DataStd-flag-0 copy is exactly that iteration.  (With an intercept the
// reference multiplies the n x nvars map by a vector of nvars + 1 entries: refused by the callers.)
static int fit_big_wide_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev, int32_t standardize,
                            const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d, bool sparse_loss = false)
{
    if (set_device(c)) return OEMGPU_ERR_HIP;
    if (n < 2) { set_error("big.oem / sparse x with p >= n: at least two rows"); return OEMGPU_ERR_UNSUPPORTED; }
    int rc;
    if (!wide_pays(n, p, o)) {
        // ... on the Gram where the two-product form does not pay (p <= 1024: the register engines; n <= p < 2 n beyond the persistent
        // engines; more than WIDE_MAX_N rows, where the wide engine ends) -- as oemgpu_fit_dense_dev does for oemDense: the iteration
        // u = X'(Y - X b)/n + d b is (dI - X'X/n) b + X'Y/n, and X X'/n and X'X/n share their non-zero spectrum
        const GramPlan pl = gram_plan(n, p, c->num_cu);
        Bump B;
        const size_t a_xs2 = B.take((size_t)p * 8), a_mom = B.take((size_t)oemgpu_moments_len(p) * 8);
        const size_t frame = B.off;
        const size_t a_t = B.take(pl.tpart_doubles * 8), a_v = B.take(pl.vpart_doubles * 8);
        size_t need = B.off;
        const size_t need2 = frame + ((size_t)p * p + p + stats_len(p)) * 8 + 1024 + paths_ws_bytes(p, p, o) + 4096;
        if (need2 > need) need = need2;
        if (ctx_reserve(c, need)) return OEMGPU_ERR_HIP;
        double *mom = (double *)(c->ws + a_mom);
        {
            Timer t(c, OEMGPU_T_MOMENTS);
            rc = shard_moments(c, pl, x_dev, n, ld, y_dev, nullptr, (double *)(c->ws + a_t), (double *)(c->ws + a_v), mom);
            if (rc) return rc;
        }
        oemgpu_opts og = *o;
        og.compute_loss = 0;                          // (a sparse x: the loss is a pass of its own below; big.oem never has one)
        rc = solve_moments_impl(c, mom, nullptr, p, OEMGPU_SEM_BIG, standardize, 0, &og, beta, lambda_out, niter, loss, d, nullptr, nullptr, nullptr, 0,
                                (double *)(c->ws + a_xs2));
        return (rc || !sparse_loss) ? rc : big_wide_sparse_loss(c, x_dev, n, ld, p, y_dev, o, beta, niter, loss);
    }
    const WideLayout lay = wide_layout(n);
    Bump X;
    const size_t a_xs = X.take(sizeof(double) * (size_t)lay.rows() * p), a_ys = X.take(sizeof(double) * (size_t)lay.rows()),
                 a_sc = X.take(sizeof(double) * wide_scratch_doubles((int)n, p));
    if (ctx_grow(c, &c->aux, &c->aux_bytes, X.off)) return OEMGPU_ERR_HIP;
    Bump B;
    const size_t a_xs2 = B.take((size_t)p * 8), a_xy = B.take((size_t)p * 8), a_st = B.take((size_t)stats_len(p) * 8);       // stats last (run_paths)
    if (ctx_reserve(c, B.off + paths_ws_bytes(p, p, o) + 4096)) return OEMGPU_ERR_HIP;
    double *xs = (double *)(c->aux + a_xs), *ys = (double *)(c->aux + a_ys);
    double *xy = (double *)(c->ws + a_xy), *st = (double *)(c->ws + a_st), *xy_std = (double *)(c->ws + a_xs2);
    {
        Timer t(c, OEMGPU_T_MOMENTS);
        rc = launch_wide_standardize(c->stream, x_dev, n, ld, p, y_dev, 0, 0, lay, xs, ys, xy, st);
        if (!rc && standardize) rc = launch_big_wide_scales(c->stream, x_dev, n, ld, p, xy, st, xy_std);
        if (rc) return rc;
    }
    WideArgs wd;
    wd.xs = xs; wd.ys = ys; wd.lay = lay; wd.n = (int)n; wd.scratch = (double *)(c->aux + a_sc);
    rc = run_paths_parts(c, B, nullptr, xy, st, p, p, OEMGPU_SEM_BIG, standardize, 0, o, beta, lambda_out, niter, loss, d, &wd, standardize ? xy_std : nullptr);
    return (rc || !sparse_loss) ? rc : big_wide_sparse_loss(c, x_dev, n, ld, p, y_dev, o, beta, niter, loss);
}

// the same from one contiguous host matrix (n rows, column-major, ld = n)
static int fit_big_wide_host(const double *x, int64_t n, int32_t p, const double *y, int32_t standardize, const oemgpu_opts *o,
                             double *beta, double *lambda_out, int32_t *niter, double *loss, double *d, bool sparse_loss = false)
{
    oemgpu_ctx *c = ctx_acquire(o->device);
    if (!c) return OEMGPU_ERR_NO_DEVICE;
    double *xd = nullptr, *yd = nullptr;
    int64_t ld = 0;
    int rc = host_upload_resident(c, x, n, p, y, o, &xd, &ld, &yd, 0, true);
    if (!rc) rc = fit_big_wide_dev(c, xd, n, ld, p, yd, standardize, o, beta, lambda_out, niter, loss, d, sparse_loss);
    (void)hipStreamSynchronize(c->stream);
    ctx_release(c);
    return rc;
}

int oemgpu_fit_dense_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                         int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                         double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    HT(0);
    if (!c || !x_dev || !y_dev) { set_error("fit_dense: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = check_opts(o, p, p);
    if (rc) return rc;
    if (n < 1 || ld < n) { set_error("fit_dense: bad n / ld"); return OEMGPU_ERR_ARG; }
    // p >= n (ref src/oem_dense.h:476-482,513-521: d from XXt / n, u = X'(Y - X b)/n + d b): the reference's own form where it
    // pays (wide_pays), else the same iteration on the Gram: the non-zero spectra of XXt and XtX coincide and
    // X'(Y - X b)/n + d b = (dI - X'X/n) b + X'Y/n.
    if (wide_pays(n, p, o)) return fit_dense_wide_dev(c, x_dev, n, ld, p, y_dev, standardize, intercept, o, beta, lambda_out, niter, loss, d);
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const GramPlan pl = gram_plan(n, p, c->num_cu);
    Bump B;
    const size_t a_sums = B.take((size_t)oemgpu_sums_len(p) * 8), a_mom = B.take((size_t)oemgpu_moments_len(p) * 8);
    const size_t frame = B.off;
    const size_t a_t = B.take(pl.tpart_doubles * 8), a_v = B.take(pl.vpart_doubles * 8);
    size_t need = B.off;
    const size_t need2 = frame + ((size_t)p * p + p + stats_len(p)) * 8 + 1024 + paths_ws_bytes(p, p, o) + 4096;
    if (need2 > need) need = need2;
    if (ctx_reserve(c, need)) return OEMGPU_ERR_HIP;
    double *sums = (double *)(c->ws + a_sums), *mom = (double *)(c->ws + a_mom);
    // Moments about 0 first: that is what the shift predicate yields unless some column has |mean| > 16 sd, and finalize tells
    // (from the full-data moments) when it was the wrong guess.  The sample pass and the shifted pass then run only for such data.
    {
        Timer t(c, OEMGPU_T_MOMENTS);
        rc = shard_moments(c, pl, x_dev, n, ld, y_dev, nullptr, (double *)(c->ws + a_t), (double *)(c->ws + a_v), mom);
        if (rc) return rc;
    }
    HT(1);
    rc = oemgpu_solve_moments_dev(c, mom, nullptr, p, OEMGPU_SEM_DENSE, standardize, intercept, o, beta, lambda_out, niter, loss, d);
#ifdef OEM_HOST_TIMING
    HT(5); g_ht.acc[0] += g_tp[1] - g_tp[0]; g_ht.acc[1] += g_tp[2] - g_tp[1]; g_ht.acc[2] += g_tp[3] - g_tp[2]; g_ht.acc[3] += g_tp[4] - g_tp[3]; g_ht.acc[4] += g_tp[5] - g_tp[4]; ++g_ht.n;
#endif
    if (rc || !c->shift_advised) return rc;
    {
        Timer t(c, OEMGPU_T_SHIFT);
        rc = launch_shift_sums(c->stream, x_dev, n, ld, p, y_dev, sums);
        if (rc) return rc;
    }
    {
        Timer t(c, OEMGPU_T_MOMENTS);
        rc = shard_moments(c, pl, x_dev, n, ld, y_dev, sums, (double *)(c->ws + a_t), (double *)(c->ws + a_v), mom);
        if (rc) return rc;
    }
    return oemgpu_solve_moments_dev(c, mom, sums, p, OEMGPU_SEM_DENSE, standardize, intercept, o, beta, lambda_out, niter, loss, d);
}

// oemDense with observation weights (weighted.hip has the algebra and the reference lines).  n > p.
static int fit_dense_weighted_impl(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev, const double *w_dev,
                                   const double *w_host, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                                   double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!c || !x_dev || !y_dev || (!w_dev && !w_host)) { set_error("fit_dense_weighted: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = check_opts(o, p, p);
    if (rc) return rc;
    if (n < 1 || ld < n) { set_error("fit_dense_weighted: bad n / ld"); return OEMGPU_ERR_ARG; }
    const bool wide = n <= p;                             // the XWXt branch (ref src/oem_dense.h:466-471, 513-517)
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const int flag = (standardize ? 1 : 0) + 2 * (intercept ? 1 : 0);
    const int64_t ldz = (n + 63) / 64 * 64;
    Bump Z;
    const size_t a_z = Z.take(sizeof(double) * (size_t)ldz * p), a_yz = Z.take(sizeof(double) * (size_t)ldz), a_ws = Z.take(sizeof(double) * (size_t)(2 + 2 * p)),
                 a_w = Z.take(sizeof(double) * (size_t)n);
    // nobs <= nvars: the first pass's X'WX / n, X'(Yw) / n and constants are kept -- d, lambda_zero and the loss belong to them
    const size_t a_xx1 = wide ? Z.take(sizeof(double) * (size_t)p * p) : 0, a_xy1 = wide ? Z.take(sizeof(double) * (size_t)p) : 0,
                 a_st1 = wide ? Z.take(sizeof(double) * (size_t)stats_len(p)) : 0;
    if (ctx_grow(c, &c->aux, &c->aux_bytes, Z.off)) return OEMGPU_ERR_HIP;
    double *z = (double *)(c->aux + a_z), *yz = (double *)(c->aux + a_yz), *wsd = (double *)(c->aux + a_ws);
    if (w_host) {
        for (int64_t i = 0; i < n; ++i)
            if (!(w_host[i] >= 0.0) || !std::isfinite(w_host[i])) { set_error("fit_dense_weighted: weights must be finite and >= 0"); return OEMGPU_ERR_ARG; }
        OEM_HIP(hipMemcpyAsync(c->aux + a_w, w_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        w_dev = (const double *)(c->aux + a_w);
    }
    const GramPlan pl = gram_plan(n, p, c->num_cu);
    Bump B;
    const size_t a_mom = B.take((size_t)oemgpu_moments_len(p) * 8);
    const size_t frame = B.off;
    const size_t a_t = B.take(pl.tpart_doubles * 8), a_v = B.take(pl.vpart_doubles * 8);
    size_t need = B.off;
    const size_t large_ws = wide ? path_large_work_doubles(p, 128) * 8 + 4096 : 0;      // (the launch-per-iteration engine at any p)
    const size_t need2 = frame + ((size_t)p * p + p + stats_len(p)) * 8 + 1024 + paths_ws_bytes(p, p, o) + large_ws + 4096;
    if (need2 > need) need = need2;
    if (ctx_reserve(c, need)) return OEMGPU_ERR_HIP;
    double *mom = (double *)(c->ws + a_mom);
    {
        Timer t(c, OEMGPU_T_MOMENTS);
        rc = launch_weighted_stats(c->stream, x_dev, n, ld, p, y_dev, w_dev, flag, wsd);
        if (!rc) rc = launch_weighted_apply(c->stream, x_dev, n, ld, p, y_dev, w_dev, flag, wsd, 0, z, ldz, yz);
        if (!rc) rc = shard_moments(c, pl, z, n, ldz, yz, nullptr, (double *)(c->ws + a_t), (double *)(c->ws + a_v), mom);
        if (rc) return rc;
    }
    if (!wide) return solve_moments_impl(c, mom, nullptr, p, OEMGPU_SEM_DENSE, standardize, intercept, o, beta, lambda_out, niter, loss, d, wsd);
    // ---- nobs <= nvars.  The reference takes d from (sqrt(w) Xs)(sqrt(w) Xs)'/n -- the non-zero spectrum of the Z'Z / n just formed --
    // and lambda_zero from XY = Xs'(Ys w)/n, but iterates u = Xs'((Ys - Xs beta) w^2)/n + d beta: the Gram form on Z2 = diag(w) Xs.
    double *xx1 = (double *)(c->aux + a_xx1), *xy1 = (double *)(c->aux + a_xy1), *st1 = (double *)(c->aux + a_st1);
    rc = launch_finalize(c->stream, mom, nullptr, p, OEMGPU_SEM_DENSE, 0, 0, xx1, xy1, st1);
    if (!rc) rc = launch_weighted_patch_stats(c->stream, wsd, p, st1);
    double lam1 = 0.0;
    if (!rc) rc = oemgpu_eig_max_dev(c, xx1, p, &lam1);              // (uses the workspace from its start: nothing of this call lives there now)
    if (rc) return rc;
    if (ctx_reserve(c, need)) return OEMGPU_ERR_HIP;
    mom = (double *)(c->ws + a_mom);
    {
        Timer t(c, OEMGPU_T_MOMENTS);
        rc = launch_weighted_apply(c->stream, x_dev, n, ld, p, y_dev, w_dev, flag, wsd, 1, z, ldz, yz);
        if (!rc) rc = shard_moments(c, pl, z, n, ldz, yz, nullptr, (double *)(c->ws + a_t), (double *)(c->ws + a_v), mom);
        if (rc) return rc;
    }
    PathExtras ex;
    ex.d_fixed = lam1 * 1.005;                                        // ref src/oem_dense.h:498
    ex.loss_xx = xx1; ex.loss_xy = xy1; ex.loss_stats = st1;          // get_loss = sum w (Ys - Xs beta)^2 (ref :759-770): the FIRST Gram's identity
    return solve_moments_impl(c, mom, nullptr, p, OEMGPU_SEM_DENSE, standardize, intercept, o, beta, lambda_out, niter, loss, d, wsd, &ex, xy1, large_ws);
}

int oemgpu_fit_dense_weighted_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev, const double *w_dev,
                                  int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                                  double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    return fit_dense_weighted_impl(c, x_dev, n, ld, p, y_dev, w_dev, nullptr, standardize, intercept, o, beta, lambda_out, niter, loss, d);
}

// host buffers in, as `.Call("oem_fit_dense", x, y, family, penalty, weights, ...)` hands them over (ref src/oem_dense.cpp:30-75)
int oemgpu_fit_dense_weighted(const double *x, int64_t n, int32_t p, const double *y, const double *weights, int32_t standardize, int32_t intercept,
                              const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!x || !y || !weights || !o || !beta || !lambda_out || !niter || !loss || !d) { set_error("fit_dense_weighted: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = check_opts(o, p, p);
    if (rc) return rc;
    if (n < 1) { set_error("fit_dense_weighted: bad n"); return OEMGPU_ERR_ARG; }
    oemgpu_ctx *c = ctx_acquire(o->device);
    if (!c) return OEMGPU_ERR_NO_DEVICE;
    double *xd = nullptr, *yd = nullptr;
    int64_t ld = 0;
    rc = host_upload_resident(c, x, n, p, y, o, &xd, &ld, &yd, 0, true);
    if (!rc) rc = fit_dense_weighted_impl(c, xd, n, ld, p, yd, nullptr, weights, standardize, intercept, o, beta, lambda_out, niter, loss, d);
    (void)hipStreamSynchronize(c->stream);
    ctx_release(c);
    return rc;
}

int oemgpu_fit_xtx_dev(oemgpu_ctx *c, const double *xtx_dev, const double *xty_dev, int32_t p, const double *scale_factor,
                       const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!c || !xtx_dev || !xty_dev || !beta || !lambda_out || !niter || !loss || !d) { set_error("fit_xtx: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = check_opts(o, p, p);
    if (rc) return rc;
    if (set_device(c)) return OEMGPU_ERR_HIP;
    Bump B;
    const size_t a_xx = B.take((size_t)p * p * 8), a_xy = B.take((size_t)p * 8), a_sf = B.take((size_t)p * 8),
                 a_st = B.take((size_t)stats_len(p) * 8);      // stats last: run_paths returns it with the outputs
    if (ctx_reserve(c, B.off + paths_ws_bytes(p, p, o) + 4096)) return OEMGPU_ERR_HIP;
    double *xx = (double *)(c->ws + a_xx), *xy = (double *)(c->ws + a_xy), *st = (double *)(c->ws + a_st), *sf = (double *)(c->ws + a_sf);
    std::vector<double> sinv;
    if (scale_factor) {
        sinv.resize(p);
        for (int j = 0; j < p; ++j) sinv[j] = 1 / scale_factor[j];                // ref src/oem_xtx.h:527-529
        OEM_HIP(hipMemcpyAsync(sf, sinv.data(), sizeof(double) * p, hipMemcpyHostToDevice, c->stream));
        OEM_HIP(hipStreamSynchronize(c->stream));
    }
    {
        Timer t(c, OEMGPU_T_FINAL);
        rc = launch_xtx_prepare(c->stream, xtx_dev, xty_dev, scale_factor ? sf : nullptr, p, xx, xy, st);
        if (rc) return rc;
    }
    return run_paths(c, B, xx, xy, st, p, p, SEM_XTX, 0, 0, o, scale_factor, beta, lambda_out, niter, loss, d);
}

int oemgpu_last_shift_in_effect(oemgpu_ctx *c) { return c ? c->shifted : -1; }
int oemgpu_last_shift_advised(oemgpu_ctx *c) { return c ? c->shift_advised : -1; }

int oemgpu_eig_max_dev(oemgpu_ctx *c, const double *a_dev, int32_t p, double *lambda_max)
{
    if (!c || !a_dev || !lambda_max || p < 1) { set_error("eig_max: bad argument"); return OEMGPU_ERR_ARG; }
    if (set_device(c)) return OEMGPU_ERR_HIP;
    // run the engines with zero penalties: they stop after the eigenvalue step
    Bump B;
    const size_t a_z = B.take((size_t)(p + 8) * 8), a_o = B.take(256);
    const int lan = 2 * p < 32 ? 32 : (2 * p < 288 ? 2 * p : 288);
    size_t work_d = p <= SMALL_P_MAX ? path_small_xchg_bytes() / 8 : path_large_work_doubles(p, lan);
    const bool coop = path_coop_eligible(p, false, false, 0, 1);
    if (coop && work_d < path_coop_xchg_bytes() / 8) work_d = path_coop_xchg_bytes() / 8;
    const size_t a_w = B.take(work_d * 8);
    if (ctx_reserve(c, B.off)) return OEMGPU_ERR_HIP;
    if (ctx_pinned(c, 16384)) return OEMGPU_ERR_HIP;
    OEM_HIP(hipMemsetAsync(c->ws + a_z, 0, (size_t)(p + 8) * 8, c->stream));
    PathArgs a;
    memset(&a, 0, sizeof a);
    a.p = p; a.npen = 0; a.nl = 1; a.maxit = 1; a.lanczos_steps = lan; a.lambda_min_ratio = 0.5;
    a.xx = a_dev; a.xy = (const double *)(c->ws + a_z); a.pf = a.xy; a.stats = a.xy;
    a.d_out = (double *)(c->ws + a_o);
    a.work = (double *)(c->ws + a_w);
    a.pen_lo = 0; a.pen_hi = 0;
    CoopSlots slots;
    if (coop) slots.take(c->device, path_coop_workgroups(p), c->num_cu * 3 / 4);
    int rc = p <= SMALL_P_MAX ? launch_path_small(c->stream, a) : (coop ? launch_path_coop(c->stream, a) : run_path_large(c->stream, a, (double *)c->pinned));
    if (rc) return rc;
    double h[D_OUT_LEN];
    OEM_HIP(hipMemcpyAsync(h, a.d_out, sizeof h, hipMemcpyDeviceToHost, c->stream));
    OEM_HIP(hipStreamSynchronize(c->stream));
    if (h[6] != 0.0) { set_error("cooperating workgroups lost each other (exchange timeout)"); return OEMGPU_ERR_INTERNAL; }
    *lambda_max = h[1];
    c->eig_steps = (int)h[4]; c->eig_capped = h[5] != 0.0;
    return 0;
}

int oemgpu_last_path_engine(oemgpu_ctx *c, int32_t *engine, int32_t *persistent_fallbacks)
{
    if (!c) return -1;
    if (engine) *engine = c->last_engine;
    if (persistent_fallbacks) *persistent_fallbacks = c->persistent_fallbacks;
    return 0;
}

int oemgpu_last_placement(oemgpu_ctx *c) { return c ? c->last_placement : -1; }

int oemgpu_last_eigen_info(oemgpu_ctx *c, int32_t *steps, int32_t *capped)
{
    if (!c) return -1;
    if (steps) *steps = c->eig_steps;
    if (capped) *capped = c->eig_capped ? 1 : 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------- drop-in entry points
// Host buffers in, host buffers out.  Contexts come from the process-wide cache and own every device / pinned buffer the
// calls need (grow-only), so a repeated call creates no stream, allocates nothing and frees nothing.
int oemgpu_fit_dense(const double *x, int64_t n, int32_t p, const double *y, int32_t standardize, int32_t intercept,
                     const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!x || !y || !o || !beta || !lambda_out || !niter || !loss || !d) { set_error("fit_dense: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = check_opts(o, p, p);
    if (rc) return rc;
    if (n < 1) { set_error("fit_dense: bad n"); return OEMGPU_ERR_ARG; }
    // p >= n (ref src/oem_dense.h:476-482,513-521): where the reference's two-GEMV form pays (wide_pays) the rows go up once and the
    // wide engine iterates through the standardised copy -- no (p+2)^2 moment buffer; otherwise the same iteration on the Gram.
    if (wide_pays(n, p, o) && o->ngpus <= 1) {
        oemgpu_ctx *c = ctx_acquire(o->device);
        if (!c) return OEMGPU_ERR_NO_DEVICE;
        double *xd = nullptr, *yd = nullptr;
        int64_t ld = 0;
        rc = host_upload_resident(c, x, n, p, y, o, &xd, &ld, &yd, 0, true);      // ld = n: whole columns go up as linear copies
        if (!rc) rc = fit_dense_wide_dev(c, xd, n, ld, p, yd, standardize, intercept, o, beta, lambda_out, niter, loss, d);
        (void)hipStreamSynchronize(c->stream);
        ctx_release(c);
        return rc;
    }
    return host_fit_dense(x, n, p, y, standardize, intercept, o, beta, lambda_out, niter, loss, d);
}

int oemgpu_fit_xtx(const double *xtx, const double *xty, int32_t p, const double *scale_factor, const oemgpu_opts *o,
                   double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!xtx || !xty || !o) { set_error("fit_xtx: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = check_opts(o, p, p);
    if (rc) return rc;
    oemgpu_ctx *c = ctx_acquire(o->device);
    if (!c) return OEMGPU_ERR_NO_DEVICE;
    const size_t nb = sizeof(double) * (size_t)p * p;
    rc = ctx_grow(c, &c->xres, &c->xres_bytes, nb + sizeof(double) * (size_t)p + 256);
    double *ad = (double *)c->xres, *bd = (double *)(c->xres + (nb + 255) / 256 * 256);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(ad, xtx, nb, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(bd, xty, sizeof(double) * (size_t)p, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { set_error("upload of xtx failed: %s", hipGetErrorString(e)); rc = OEMGPU_ERR_HIP; }
    }
    if (!rc) rc = oemgpu_fit_xtx_dev(c, ad, bd, p, scale_factor, o, beta, lambda_out, niter, loss, d);
    (void)hipStreamSynchronize(c->stream);
    ctx_release(c);
    return rc;
}

int oemgpu_fit_big(const double *const *x_shards, const int64_t *n_shard, int32_t nshards, int32_t p,
                   const double *const *y_shards, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                   double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!x_shards || !n_shard || !y_shards || !o || nshards < 1 || !beta || !lambda_out || !niter || !loss || !d) {
        set_error("fit_big: bad argument"); return OEMGPU_ERR_ARG;
    }
    const int q = p + (intercept ? 1 : 0);
    int rc = check_opts(o, p, q);
    if (rc) return rc;
    if (o->compute_loss) {
        set_error("compute.loss is not available for big.oem: the reference expression is ill-formed (src/oem_big.h:899-921)");
        return OEMGPU_ERR_UNSUPPORTED;
    }
    int64_t n = 0;
    for (int s = 0; s < nshards; ++s) {
        if (n_shard[s] < 0 || (n_shard[s] > 0 && (!x_shards[s] || !y_shards[s]))) { set_error("fit_big: bad shard %d", s); return OEMGPU_ERR_ARG; }
        n += n_shard[s];
    }
    if (n <= q) {
        // nobs <= nvars + intercept: the XXt branch (ref src/oem_big.h:537-541, 568-584)
        if (intercept) {
            set_error("big.oem with p >= n and an intercept: the reference multiplies the n x p map by a vector of p + 1 entries (src/oem_big.h:568-584) -- "
                      "nothing well-formed to reproduce; intercept = FALSE is served");
            return OEMGPU_ERR_UNSUPPORTED;
        }
        std::vector<double> xc, yc;                                     // the shards as one matrix (n <= p rows)
        try { xc.resize((size_t)n * p); yc.resize((size_t)n); }
        catch (const std::bad_alloc &) { set_error("big.oem with p >= n: no host memory for the %lld x %d matrix", (long long)n, p); return OEMGPU_ERR_ARG; }
        int64_t r0 = 0;
        for (int s = 0; s < nshards; ++s) {
            const int64_t ns = n_shard[s];
            for (int j = 0; j < p && ns > 0; ++j) memcpy(xc.data() + (size_t)j * n + r0, x_shards[s] + (size_t)j * ns, sizeof(double) * (size_t)ns);
            if (ns > 0) memcpy(yc.data() + r0, y_shards[s], sizeof(double) * (size_t)ns);
            r0 += ns;
        }
        return fit_big_wide_host(xc.data(), n, p, yc.data(), standardize, o, beta, lambda_out, niter, loss, d);
    }
    return host_fit_big(x_shards, n_shard, nshards, p, y_shards, standardize, intercept, o, beta, lambda_out, niter, loss, d);
}

// ---------------------------------------------------------------------------------------------- oem() on a sparse X
// oemSparse's Gram is oemBig's with the intercept column holding `intval` instead of 1 (ref src/oem_sparse.h:577-593):
// XX_sparse = D XX_big D, XY_sparse = D XY_big, D = diag(intval, 1, ..., 1).
static __global__ void scale_intercept_kernel(double *__restrict__ xx, double *__restrict__ xy, int q, double intval)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= q) return;
    if (j == 0) { xx[0] *= intval * intval; xy[0] *= intval; }
    else { xx[(size_t)j * q] *= intval; xx[j] *= intval; }
}

// rows [r0, r1) of a compressed-column matrix into a zeroed dense column-major tile (row indices increase inside a column:
// one lower_bound per workgroup finds where the tile's part of the column starts)
static __global__ void csc_densify_kernel(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx, const double *__restrict__ val,
                                          int64_t r0, int64_t r1, int64_t ld, double *__restrict__ xd)
{
    __shared__ int64_t first;
    const int j = blockIdx.y;
    const int64_t lo0 = colptr[j], hi0 = colptr[j + 1];
    if (threadIdx.x == 0) {
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rowidx[mid] < r0) lo = mid + 1; else hi = mid; }
        first = lo;
    }
    __syncthreads();
    const int64_t k = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < hi0) { const int64_t r = rowidx[k]; if (r < r1) xd[(size_t)j * ld + (r - r0)] = val[k]; }
}

// nbatch moment buffers (instance b at moments + b * mstride, all OUTSIDE the context workspace), one finalize each, then ONE
// launch that walks all their paths side by side (q <= SMALL_P_MAX).  Outputs as in run_paths.
// cs != nullptr: observation weights -- the moments are those of the p + 1 data columns [sqrt(w) | sqrt(w) X] ((p + 3)^2 each) and
// cs holds the unweighted column sums of squares and the row count of every instance (launch_finalize_weighted).
static int solve_moments_batch(oemgpu_ctx *c, const double *moments, size_t mstride, int nbatch, int32_t p, int32_t semantics,
                               int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                               double *beta, double *lambda_out, int32_t *niter, double *loss, double *d, bool shared_lmax,
                               const double *cs = nullptr, size_t cstride = 0)
{
    const int q = p + ((semantics != OEMGPU_SEM_DENSE && intercept) ? 1 : 0);
    Bump B;
    const size_t per = ((size_t)q * q * 8 + 255) / 256 * 256 + ((size_t)q * 8 + 255) / 256 * 256 + ((size_t)stats_len(p) * 8 + 255) / 256 * 256;
    const size_t a_0 = B.take(per * nbatch);
    if (ctx_reserve(c, B.off + paths_ws_bytes(p, q, o, nbatch) + 4096)) return OEMGPU_ERR_HIP;
    const size_t o_xy = ((size_t)q * q * 8 + 255) / 256 * 256, o_st = o_xy + ((size_t)q * 8 + 255) / 256 * 256;
    for (int b = 0; b < nbatch; ++b) {
        char *f = c->ws + a_0 + per * b;
        int rc = cs ? launch_finalize_weighted(c->stream, moments + mstride * b, cs + cstride * b, p, standardize, intercept, (double *)f,
                                               (double *)(f + o_xy), (double *)(f + o_st))
                    : launch_finalize(c->stream, moments + mstride * b, nullptr, p, semantics, standardize, intercept, (double *)f,
                                      (double *)(f + o_xy), (double *)(f + o_st));
        if (rc) return rc;
    }
    char *f0 = c->ws + a_0;
    return run_paths(c, B, (const double *)f0, (const double *)(f0 + o_xy), (const double *)(f0 + o_st), p, q, semantics, standardize,
                     intercept, o, nullptr, beta, lambda_out, niter, loss, d, nbatch, per / 8, shared_lmax);
}

// ---------------------------------------------------------------------------------------------- xval.oem
static int any_grp_count(const oemgpu_opts *o)
{
    for (int k = 0; k < o->npen; ++k) if (pen_is_grp(o->penalty[k])) return o->ngroups;
    return 0;
}

static int ctx_aux(oemgpu_ctx *c, size_t bytes)
{
    if (bytes <= c->aux_bytes) return 0;
    if (c->aux) { OEM_HIP(hipStreamSynchronize(c->stream)); OEM_HIP(hipFree(c->aux)); c->aux = nullptr; c->aux_bytes = 0; }
    OEM_HIP(hipMalloc((void **)&c->aux, bytes)); ++g_alloc_count;
    c->aux_bytes = bytes;
    return 0;
}

// The call in three phases over one buffer layout in oemgpu_ctx::aux (a pure function of n, p, K, npen, nl and "weighted", so
// the phases may also be separate C-ABI calls with a collective between them: row shards on several GPUs, oemgpu_xval_*_dev):
//   prepare  rows into fold order, per-fold moments of the LOCAL rows           (additive over row shards)
//   solve    fold sums, the full-data fit and the K left-out-fold fits           (replicated; needs the moments of ALL rows)
//   cverr    per-observation error of the LOCAL rows, merged to (count, mean, M2) per (penalty, lambda)
struct XvalLay {
    int64_t n, ldp;
    int p, pm, K, npen, nl, nwg;
    bool weighted;
    size_t mlen, cslen;
    GramPlan plmax;
    size_t a_cs, a_cnt, a_fn, a_bad, a_pos, a_xp, a_yp, a_mf, a_mc, a_ms, a_t, a_v, a_b, a_part, a_out, a_peer, total;
};
static XvalLay xval_layout(oemgpu_ctx *c, int64_t n, int p, int K, int npen, int nl, bool weighted)
{
    XvalLay L;
    L.n = n; L.p = p; L.K = K; L.npen = npen; L.nl = nl; L.weighted = weighted;
    L.ldp = (n + 16 * (int64_t)K + 15) / 16 * 16;
    // observation weights: the fold-ordered copy gets a leading column sqrt(w) and everything is scaled by sqrt(w), so the moment
    // kernels see pm = p + 1 data columns and their Gram IS X'WX with its intercept border (ref src/oem_xval_dense.h:486-623)
    L.pm = p + (weighted ? 1 : 0);
    L.mlen = (size_t)oemgpu_moments_len(L.pm); L.cslen = (size_t)p + 1;
    L.nwg = cv_wg_per_fold(n, K, npen, c->num_cu);
    L.plmax = gram_plan_bound(n, L.pm, c->num_cu);             // holds the moment plan of every fold
    Bump A;
    L.a_cs = A.take(sizeof(double) * L.cslen * (2 * (size_t)K + 1));      // per fold, then all folds / all but fold ff
    L.a_cnt = A.take(fold_layout_ints(n, K) * sizeof(int)); L.a_fn = A.take(sizeof(int64_t) * 2 * K); L.a_bad = A.take(256);
    L.a_pos = A.take(sizeof(int) * (size_t)n); L.a_xp = A.take(sizeof(double) * (size_t)L.ldp * L.pm);
    L.a_yp = A.take(sizeof(double) * (size_t)L.ldp); L.a_mf = A.take(sizeof(double) * L.mlen * K);
    L.a_mc = A.take(sizeof(double) * L.mlen); L.a_ms = A.take(sizeof(double) * L.mlen * (K + 1));
    L.a_t = A.take(L.plmax.tpart_doubles * 8); L.a_v = A.take(L.plmax.vpart_doubles * 8);
    L.a_b = A.take(sizeof(double) * (size_t)K * npen * nl * (p + 1));
    L.a_part = A.take(sizeof(double) * cv_part_doubles(L.nwg, K, npen, nl)); L.a_out = A.take(sizeof(double) * 3 * (size_t)npen * nl);
    L.a_peer = A.take(sizeof(double) * (L.mlen + L.cslen) * K);            // another device's fold moments on their way into the sum
    L.total = A.off;
    return L;
}

static int xval_check(oemgpu_ctx *c, const void *x_dev, const void *y_dev, const void *foldid_dev, int64_t n, int64_t ld, int p, int K,
                      int intercept, int type_measure, bool weighted, const oemgpu_opts *o)
{
    if (!o) { set_error("xval_dense: NULL argument"); return OEMGPU_ERR_ARG; }
    (void)c; (void)x_dev; (void)y_dev; (void)foldid_dev;
    if (intercept >= 0) {                                          // (< 0: a phase that does not solve; the options were checked by the one that does)
        int rc = check_opts(o, p, p + (intercept ? 1 : 0));
        if (rc) return rc;
    } else if (!o->penalty || o->npen < 1) { set_error("xval_dense: no penalty"); return OEMGPU_ERR_ARG; }
    if (K < 2 || K > 512) { set_error("xval_dense: nfolds must be in 2..512"); return OEMGPU_ERR_ARG; }
    if (type_measure != 0 && type_measure != 1) { set_error("xval_dense: type_measure must be 0 (mse) or 1 (mae)"); return OEMGPU_ERR_ARG; }
    if (n < 1 || ld < n) { set_error("xval_dense: bad n / ld"); return OEMGPU_ERR_ARG; }
    if (n + 16 * (int64_t)K >= (int64_t)1 << 31) { set_error("xval_dense: n too large for 32-bit row positions"); return OEMGPU_ERR_UNSUPPORTED; }
    if (weighted && o->compute_loss) {
        set_error("compute.loss with observation weights: the reference's loss is the unweighted residual sum (src/oem_xval_dense.h:1122-1145), "
                  "which the weighted Gram does not hold");
        return OEMGPU_ERR_UNSUPPORTED;
    }
    return 0;
}

// phase 1: hf <- the local fold sizes [K] and fold starts [K]; mfold (and csq) in aux <- the local rows' per-fold moments
static int xval_prepare(oemgpu_ctx *c, const XvalLay &L, const double *x_dev, int64_t ld, const double *y_dev, const double *w_dev,
                        const int32_t *foldid_dev, std::vector<int64_t> &hf)
{
    const int64_t n = L.n;
    const int K = L.K, p = L.p;
    char *ax = c->aux;
    int *blockcnt = (int *)(ax + L.a_cnt), *bad = (int *)(ax + L.a_bad), *pos = (int *)(ax + L.a_pos);
    int64_t *fold_n = (int64_t *)(ax + L.a_fn), *fold_start = fold_n + K;
    double *xp = (double *)(ax + L.a_xp), *yp = (double *)(ax + L.a_yp), *mfold = (double *)(ax + L.a_mf);
    // ---- rows into fold order
    int rc = launch_fold_layout(c->stream, foldid_dev, n, K, blockcnt, fold_n, fold_start, pos, bad);
    if (rc) return rc;
    hf.assign(2 * K, 0);
    int hbad = 0;
    OEM_HIP(hipMemcpyAsync(hf.data(), fold_n, sizeof(int64_t) * 2 * K, hipMemcpyDeviceToHost, c->stream));
    OEM_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    double *csq = (double *)(ax + L.a_cs);
    if (w_dev) {
        rc = launch_gather_rows(c->stream, w_dev, n, n, 1, y_dev, pos, xp, L.ldp, yp);               // column 0 <- w
        if (!rc) rc = launch_gather_rows(c->stream, x_dev, n, ld, p, y_dev, pos, xp + L.ldp, L.ldp, yp);
        if (!rc) rc = launch_weight_scale(c->stream, xp, L.ldp, yp, p, K, fold_start, fold_n, csq);
    } else rc = launch_gather_rows(c->stream, x_dev, n, ld, p, y_dev, pos, xp, L.ldp, yp);
    if (rc) return rc;
    OEM_HIP(hipStreamSynchronize(c->stream));
    if (hbad) { set_error("xval_dense: foldid must hold values in 1..nfolds"); return OEMGPU_ERR_ARG; }
    // ---- per-fold moments about 0 (ref src/oem_xval_dense.h:358-484), one MFMA pass per fold segment
    for (int k = 0; k < K; ++k) {
        const int64_t nk = hf[k], st = hf[K + k];
        if (nk == 0) { OEM_HIP(hipMemsetAsync(mfold + L.mlen * k, 0, sizeof(double) * L.mlen, c->stream)); continue; }
        const GramPlan pl = gram_plan(nk, L.pm, c->num_cu);
        if (pl.tpart_doubles > L.plmax.tpart_doubles || pl.vpart_doubles > L.plmax.vpart_doubles) {
            set_error("internal: fold plan larger than its scratch"); return OEMGPU_ERR_INTERNAL;
        }
        rc = shard_moments(c, pl, xp + st, nk, L.ldp, yp + st, nullptr, (double *)(ax + L.a_t), (double *)(ax + L.a_v), mfold + L.mlen * k);
        if (rc) return rc;
    }
    return 0;
}

// phase 2: mfold (and csq) in aux hold the per-fold moments of ALL rows; fold_tot[K] their fold sizes, n_tot the row count.
// The full-data fit (ff = 0) and one fit per left-out fold on the lambdas of the first (ref src/oem_xval_dense.cpp:213-340);
// the fold coefficients end in aux (bdev) for phase 3.
static int xval_solve(oemgpu_ctx *c, const XvalLay &L, const int64_t *fold_tot, int64_t n_tot, int32_t standardize, int32_t intercept,
                      const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    const int K = L.K, p = L.p, npen = L.npen, nl = L.nl, q = p + (intercept ? 1 : 0);
    const size_t mlen = L.mlen, cslen = L.cslen;
    const bool weighted = L.weighted;
    char *ax = c->aux;
    double *mfold = (double *)(ax + L.a_mf), *bdev = (double *)(ax + L.a_b);
    double *csq = (double *)(ax + L.a_cs), *cssum = csq + cslen * K;
    if (n_tot <= p) { set_error("dimension of x larger than number of observations"); return OEMGPU_ERR_UNSUPPORTED; }   // ref src/oem_xval_dense.h:690-731
    for (int k = 0; k < K; ++k)
        if (n_tot - fold_tot[k] <= p) { set_error("dimension of x larger than number of observations"); return OEMGPU_ERR_UNSUPPORTED; }  // ref :849-852
    int rc = 0;
    const size_t blen = (size_t)npen * nl * (p + 1), nk2 = (size_t)npen * nl;
    std::vector<double> hb(blen * K);
    double *msum = (double *)(ax + L.a_ms);                      // [K + 1]: all folds, then all but fold ff
    for (int ff = 0; ff <= K; ++ff) {
        rc = launch_fold_sum(c->stream, mfold, K, mlen, ff, msum + mlen * ff);
        if (!rc && weighted) rc = launch_fold_sum(c->stream, csq, K, cslen, ff, cssum + cslen * ff);
        if (rc) return rc;
    }
    const bool coop_batch = q > SMALL_P_MAX && path_coop_eligible(q, false, o->compute_loss != 0, any_grp_count(o), K + 1) &&
                            path_coop_workgroups(q) * (K + 1) <= c->num_cu * 3 / 4;
    if (q <= SMALL_P_MAX || coop_batch) {
        // The K + 1 fits are independent chains of tiny dependent steps: ONE launch, one workgroup (set) per fit.  (K streams
        // would run four at a time: the hardware queues are few.)  The folds' lambda grid is the full fit's, which a fold's
        // kernel derives from the full-data X'Y itself (PathArgs::lmax_xy) instead of waiting for the full fit to end.
        std::vector<double> ab(blen * (K + 1)), al(nk2 * (K + 1)), aloss(nk2 * (K + 1)), ad(K + 1);
        std::vector<int32_t> an(nk2 * (K + 1));
        rc = solve_moments_batch(c, msum, mlen, K + 1, p, OEMGPU_SEM_XVAL, standardize, intercept, o, ab.data(), al.data(), an.data(),
                                 aloss.data(), ad.data(), true, weighted ? cssum : nullptr, cslen);
        if (rc) return rc;
        memcpy(beta, ab.data(), sizeof(double) * blen);
        memcpy(lambda_out, al.data(), sizeof(double) * nk2);
        memcpy(niter, an.data(), sizeof(int32_t) * nk2);
        memcpy(loss, aloss.data(), sizeof(double) * nk2);       // the reference reports the loss of the full fit only (ref :296-301)
        *d = ad[0];
        memcpy(hb.data(), ab.data() + blen, sizeof(double) * blen * K);
    } else if (weighted) {
        // weighted and beyond one launch's size: the K + 1 fits one after the other on this context (each a cooperating-workgroup
        // or launch-per-iteration solve of its own), the folds on the full fit's lambdas
        rc = solve_moments_batch(c, msum, mlen, 1, p, OEMGPU_SEM_XVAL, standardize, intercept, o, beta, lambda_out, niter, loss, d, false,
                                 cssum, cslen);
        if (rc) return rc;
        oemgpu_opts of = *o;
        of.lambda_user = lambda_out; of.nlambda_user = nl; of.compute_loss = 0;
        std::vector<double> hl(nk2), hloss(nk2);
        std::vector<int32_t> hn(nk2);
        double hd = 0.0;
        for (int ff = 1; ff <= K; ++ff) {
            rc = solve_moments_batch(c, msum + mlen * ff, mlen, 1, p, OEMGPU_SEM_XVAL, standardize, intercept, &of, hb.data() + blen * (ff - 1),
                                     hl.data(), hn.data(), hloss.data(), &hd, false, cssum + cslen * ff, cslen);
            if (rc) return rc;
        }
    } else {
        rc = oemgpu_solve_moments_dev(c, msum, nullptr, p, OEMGPU_SEM_XVAL, standardize, intercept, o, beta, lambda_out, niter, loss, d);
        if (rc) return rc;
        oemgpu_opts of = *o;
        of.lambda_user = lambda_out; of.nlambda_user = nl; of.compute_loss = 0;
        // The fold fits run on K worker threads: they must never call back into the caller (R's API is single-threaded, and
        // R_CheckStack on a foreign stack raises a spurious interrupt -- ADVICE r2).  The caller is polled on THIS thread only:
        // by the full-data fit above, before the fold threads start and after they have joined.
        of.interrupt = nullptr; of.interrupt_arg = nullptr;
        if (o->interrupt && o->interrupt(o->interrupt_arg)) { set_error("interrupted by the caller"); return OEMGPU_ERR_INTERRUPTED; }
        std::vector<double> hl(nk2 * K), hloss(nk2 * K), hd(K);
        std::vector<int32_t> hn(nk2 * K);
        msum += mlen;                                            // the folds' sums
        while ((int)c->kids.size() < K) {
            oemgpu_ctx *kid = oemgpu_create(c->device, nullptr);
            if (!kid) return OEMGPU_ERR_HIP;
            c->kids.push_back(kid);
        }
        if (!c->fork_ev) OEM_HIP(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
        OEM_HIP(hipEventRecord(c->fork_ev, c->stream));
        for (int ff = 1; ff <= K; ++ff) OEM_HIP(hipStreamWaitEvent(c->kids[ff - 1]->stream, c->fork_ev, 0));
        {
            std::vector<int> rcs(K, 0);
            std::vector<std::string> errs(K);
            std::vector<std::thread> th;
            th.reserve(K);
            for (int ff = 1; ff <= K; ++ff)
                th.emplace_back([&, ff]() {
                    const int i = ff - 1;
                    rcs[i] = oemgpu_solve_moments_dev(c->kids[i], msum + mlen * i, nullptr, p, OEMGPU_SEM_XVAL, standardize, intercept, &of,
                                                      hb.data() + blen * i, hl.data() + nk2 * i, hn.data() + nk2 * i, hloss.data() + nk2 * i,
                                                      &hd[i]);
                    if (rcs[i]) errs[i] = oemgpu_last_error();          // the message lives in this thread
                });
            for (auto &t : th) t.join();
            for (int i = 0; i < K; ++i)
                if (rcs[i]) { set_error("fold %d: %s", i + 1, errs[i].c_str()); return rcs[i]; }
            if (o->interrupt && o->interrupt(o->interrupt_arg)) { set_error("interrupted by the caller"); return OEMGPU_ERR_INTERRUPTED; }
        }
    }
    OEM_HIP(hipMemcpyAsync(bdev, hb.data(), sizeof(double) * blen * K, hipMemcpyHostToDevice, c->stream));
    OEM_HIP(hipStreamSynchronize(c->stream));                   // hb is a local
    return 0;
}

// phase 3: per-observation error of every LOCAL row under the fit that left its fold out (ref src/oem_xval_dense.cpp:343-461).
// triples == nullptr: cvm / cvsd of these rows (they are all rows); else triples[npen][nl][3] <- (count, mean, M2 = sum (v - mean)^2),
// which the caller merges over the row shards (oemgpu_xval_merge).
static int xval_cverr(oemgpu_ctx *c, const XvalLay &L, int32_t type_measure, const oemgpu_opts *o, double *cvm, double *cvsd, double *triples)
{
    const int K = L.K, p = L.p, npen = L.npen, nl = L.nl;
    char *ax = c->aux;
    int64_t *fold_n = (int64_t *)(ax + L.a_fn), *fold_start = fold_n + K;
    double *xp = (double *)(ax + L.a_xp), *yp = (double *)(ax + L.a_yp), *bdev = (double *)(ax + L.a_b);
    double *part = (double *)(ax + L.a_part), *cvout = (double *)(ax + L.a_out);
    int rc = launch_cv_error(c->stream, xp, L.ldp, yp, fold_start, fold_n, K, p, bdev, npen, nl, type_measure, L.weighted ? 1 : 0, L.nwg, (double)L.n,
                             part, cvout, triples != nullptr);
    if (rc) return rc;
    const int per = triples ? 3 : 2;
    std::vector<double> hc((size_t)per * npen * nl);
    OEM_HIP(hipMemcpyAsync(hc.data(), cvout, sizeof(double) * hc.size(), hipMemcpyDeviceToHost, c->stream));
    OEM_HIP(hipStreamSynchronize(c->stream));
    if (triples) { memcpy(triples, hc.data(), sizeof(double) * hc.size()); return 0; }
    for (int k = 0; k < npen; ++k) {
        const int nlam = (o->penalty[k] == OEMGPU_OLS) ? 1 : nl;
        for (int i = 0; i < nl; ++i) {
            const size_t ki = (size_t)k * nl + i;
            cvm[ki] = i < nlam ? hc[2 * ki] : 0.0;
            cvsd[ki] = i < nlam ? hc[2 * ki + 1] : 0.0;
        }
    }
    return 0;
}

int oemgpu_xval_dense_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                          const double *w_dev, const int32_t *foldid_dev, int32_t nfolds, int32_t standardize, int32_t intercept,
                          int32_t type_measure, const oemgpu_opts *o,
                          double *beta, double *lambda_out, int32_t *niter, double *loss, double *d, double *cvm, double *cvsd)
{
    if (!c || !x_dev || !y_dev || !foldid_dev || !beta || !lambda_out || !niter || !loss || !d || !cvm || !cvsd) {
        set_error("xval_dense: NULL argument"); return OEMGPU_ERR_ARG;
    }
    const int K = nfolds;
    int rc = xval_check(c, x_dev, y_dev, foldid_dev, n, ld, p, K, intercept, type_measure, w_dev != nullptr, o);
    if (rc) return rc;
    if (n <= p) { set_error("dimension of x larger than number of observations"); return OEMGPU_ERR_UNSUPPORTED; }   // ref src/oem_xval_dense.h:690-731
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const XvalLay L = xval_layout(c, n, p, K, o->npen, nl_of(o), w_dev != nullptr);
    if (ctx_aux(c, L.total)) return OEMGPU_ERR_HIP;
    std::vector<int64_t> hf;
    rc = xval_prepare(c, L, x_dev, ld, y_dev, w_dev, foldid_dev, hf);
    if (!rc) rc = xval_solve(c, L, hf.data(), n, standardize, intercept, o, beta, lambda_out, niter, loss, d);
    if (!rc) rc = xval_cverr(c, L, type_measure, o, cvm, cvsd, nullptr);
    return rc;
}

// ---- the same three phases as separate calls, for row shards on several GPUs (one process per GPU; the caller sums the fold
// moments and the fold sizes over the ranks between phases 1 and 2 and merges the error triples after phase 3).  All three calls
// of one fit take the same (n_local, p, nfolds, weighted, o): the buffer layout in the context is a function of those.
int64_t oemgpu_xval_moments_len(int32_t p, int32_t nfolds, int32_t weighted)
{
    return (int64_t)nfolds * (oemgpu_moments_len(p + (weighted ? 1 : 0)) + (weighted ? p + 1 : 0));
}

int oemgpu_xval_fold_moments_dev(oemgpu_ctx *c, const double *x_dev, int64_t n, int64_t ld, int32_t p, const double *y_dev,
                                 const double *w_dev, const int32_t *foldid_dev, int32_t nfolds, const oemgpu_opts *o,
                                 double *fold_moments_dev, int64_t *fold_n)
{
    if (!c || !x_dev || !y_dev || !foldid_dev || !fold_moments_dev || !fold_n) { set_error("xval_fold_moments: NULL argument"); return OEMGPU_ERR_ARG; }
    const int K = nfolds;
    int rc = xval_check(c, x_dev, y_dev, foldid_dev, n, ld, p, K, -1, 0, w_dev != nullptr, o);
    if (rc) return rc;
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const XvalLay L = xval_layout(c, n, p, K, o->npen, nl_of(o), w_dev != nullptr);
    if (ctx_aux(c, L.total)) return OEMGPU_ERR_HIP;
    std::vector<int64_t> hf;
    rc = xval_prepare(c, L, x_dev, ld, y_dev, w_dev, foldid_dev, hf);
    if (rc) return rc;
    for (int k = 0; k < K; ++k) fold_n[k] = hf[k];
    OEM_HIP(hipMemcpyAsync(fold_moments_dev, c->aux + L.a_mf, sizeof(double) * L.mlen * K, hipMemcpyDeviceToDevice, c->stream));
    if (w_dev) OEM_HIP(hipMemcpyAsync(fold_moments_dev + L.mlen * K, c->aux + L.a_cs, sizeof(double) * L.cslen * K, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int oemgpu_xval_solve_folds_dev(oemgpu_ctx *c, const double *fold_moments_dev, const int64_t *fold_n_total, int64_t n_local, int32_t p,
                                int32_t nfolds, int32_t weighted, int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                                double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!c || !fold_moments_dev || !fold_n_total || !beta || !lambda_out || !niter || !loss || !d) { set_error("xval_solve_folds: NULL argument"); return OEMGPU_ERR_ARG; }
    const int K = nfolds;
    int rc = xval_check(c, fold_moments_dev, fold_moments_dev, fold_moments_dev, n_local, n_local, p, K, intercept, 0, weighted != 0, o);
    if (rc) return rc;
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const XvalLay L = xval_layout(c, n_local, p, K, o->npen, nl_of(o), weighted != 0);
    if (c->aux_bytes < L.total) { set_error("xval_solve_folds: call oemgpu_xval_fold_moments_dev with the same arguments first"); return OEMGPU_ERR_ARG; }
    OEM_HIP(hipMemcpyAsync(c->aux + L.a_mf, fold_moments_dev, sizeof(double) * L.mlen * K, hipMemcpyDeviceToDevice, c->stream));
    if (weighted) OEM_HIP(hipMemcpyAsync(c->aux + L.a_cs, fold_moments_dev + L.mlen * K, sizeof(double) * L.cslen * K, hipMemcpyDeviceToDevice, c->stream));
    int64_t n_tot = 0;
    for (int k = 0; k < K; ++k) n_tot += fold_n_total[k];
    return xval_solve(c, L, fold_n_total, n_tot, standardize, intercept, o, beta, lambda_out, niter, loss, d);
}

int oemgpu_xval_cv_triples_dev(oemgpu_ctx *c, int64_t n_local, int32_t p, int32_t nfolds, int32_t weighted, int32_t type_measure,
                               const oemgpu_opts *o, double *triples)
{
    if (!c || !o || !triples) { set_error("xval_cv_triples: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = xval_check(c, triples, triples, triples, n_local, n_local, p, nfolds, -1, type_measure, weighted != 0, o);
    if (rc) return rc;
    if (set_device(c)) return OEMGPU_ERR_HIP;
    const XvalLay L = xval_layout(c, n_local, p, nfolds, o->npen, nl_of(o), weighted != 0);
    if (c->aux_bytes < L.total) { set_error("xval_cv_triples: call the first two phases with the same arguments first"); return OEMGPU_ERR_ARG; }
    return xval_cverr(c, L, type_measure, o, nullptr, nullptr, triples);
}

// (count, mean, M2) of the union of `nsets` row sets per (penalty, lambda) by Chan, Golub & LeVeque's update, in set order; then
// cvm = mean, cvsd = sqrt(M2 / (n - 1)) / sqrt(n)  (ref src/oem_xval_dense.cpp:452-461).  Pure host arithmetic.
int oemgpu_xval_merge(const double *triples, int32_t nsets, const oemgpu_opts *o, double *cvm, double *cvsd)
{
    if (!triples || !o || !cvm || !cvsd || nsets < 1) { set_error("xval_merge: bad argument"); return OEMGPU_ERR_ARG; }
    const int npen = o->npen, nl = nl_of(o);
    const size_t nk = (size_t)npen * nl;
    for (int k = 0; k < npen; ++k) {
        const int nlam = (o->penalty[k] == OEMGPU_OLS) ? 1 : nl;
        for (int i = 0; i < nl; ++i) {
            const size_t ki = (size_t)k * nl + i;
            double na = 0.0, ma = 0.0, qa = 0.0;
            for (int s = 0; s < nsets; ++s) {
                const double *t = triples + ((size_t)s * nk + ki) * 3;
                const double nb = t[0], mb = t[1], qb = t[2];
                if (!(nb > 0.0)) continue;
                if (na > 0.0) {
                    const double nn = na + nb, dl = mb - ma;
                    ma += dl * (nb / nn);
                    qa += qb + dl * dl * (na * nb / nn);
                    na = nn;
                } else { na = nb; ma = mb; qa = qb; }
            }
            cvm[ki] = i < nlam ? ma : 0.0;
            cvsd[ki] = (i < nlam && na > 1.0) ? std::sqrt((qa < 0.0 ? 0.0 : qa) / (na - 1.0)) / std::sqrt(na) : 0.0;
        }
    }
    return 0;
}

// rows [r0, r1) of the host data resident on the context's device: x (leading dimension *ld), y, foldid and the weights
static int xval_upload_rows(oemgpu_ctx *c, const double *x, int64_t n, int32_t p, const double *y, const double *weights, const int32_t *foldid,
                            int64_t r0, int64_t r1, const oemgpu_opts *o, double **xd, int64_t *ld, double **yd, double **wd, int32_t **fd)
{
    const int64_t nr = r1 - r0;
    int rc = host_upload_resident(c, x + r0, nr, p, y + r0, o, xd, ld, yd, n);   // staged through the pinned lanes, leaves room for foldid behind y
    if (rc) return rc;
    *fd = (int32_t *)(*yd + ((nr + 2 + 31) / 32 * 32));
    *wd = nullptr;
    hipError_t e = hipMemcpyAsync(*fd, foldid + r0, sizeof(int32_t) * (size_t)nr, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("upload of foldid failed: %s", hipGetErrorString(e)); return OEMGPU_ERR_HIP; }
    if (weights) {                                               // the rows' weights in the accumulator buffer of the host path (grow-only)
        rc = ctx_grow(c, &c->acc, &c->acc_bytes, sizeof(double) * (size_t)(nr + 2));
        if (rc) return rc;
        *wd = (double *)c->acc;
        if (hipMemcpyAsync(*wd, weights + r0, sizeof(double) * (size_t)nr, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            set_error("upload of the weights failed"); return OEMGPU_ERR_HIP;
        }
    }
    return 0;
}

// opts.ngpus > 1: the three phases of the call (xval_prepare / xval_solve / xval_cverr) with the rows split over the devices like
// oemgpu_fit_dense does (floor(n / G) each, the remainder on the last), one host thread per device for the two row-bound phases:
// the fold moments are handed to the first device by peer copies and added in device order, the K + 1 fits run there, the fold
// coefficients go back to every device, and the error triples of the devices' rows merge on the host (oemgpu_xval_merge).
static int xval_dense_devices(const std::vector<int> &dev, const double *x, int64_t n, int32_t p, const double *y, const double *weights,
                              const int32_t *foldid, int32_t K, int32_t standardize, int32_t intercept, int32_t type_measure,
                              const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d,
                              double *cvm, double *cvsd)
{
    const int G = (int)dev.size(), npen = o->npen, nl = nl_of(o);
    struct Part {
        oemgpu_ctx *c = nullptr;
        int64_t r0 = 0, r1 = 0, ld = 0;
        double *xd = nullptr, *yd = nullptr, *wd = nullptr;
        int32_t *fd = nullptr;
        XvalLay L;
        std::vector<int64_t> hf;
        std::vector<double> tri;
        int rc = 0;
        std::string err;
    };
    std::vector<Part> P(G);
    auto release_all = [&]() { for (Part &q : P) if (q.c) { (void)hipSetDevice(q.c->device); (void)hipStreamSynchronize(q.c->stream); ctx_release(q.c); q.c = nullptr; } };
    for (int g = 0; g < G; ++g) {
        oemgpu_row_split(n, G, g, &P[g].r0, &P[g].r1);
        if (P[g].r1 - P[g].r0 < 1) { release_all(); set_error("xval_dense: fewer rows than devices"); return OEMGPU_ERR_ARG; }
        P[g].c = ctx_acquire(dev[g]);
        if (!P[g].c) { release_all(); return OEMGPU_ERR_NO_DEVICE; }
    }
    auto each = [&](auto &&work) -> int {                           // work(g) on one host thread per device; the first failure, in device order
        std::vector<std::thread> th;
        auto run = [&](int g) {
            P[g].rc = set_device(P[g].c) ? OEMGPU_ERR_HIP : work(g);
            if (P[g].rc) P[g].err = oemgpu_last_error();                // the message lives in the worker's thread
        };
        for (int g = 1; g < G; ++g) th.emplace_back(run, g);
        run(0);
        for (auto &t : th) t.join();
        for (int g = 0; g < G; ++g) if (P[g].rc) { set_error("device %d: %s", dev[g], P[g].err.c_str()); return P[g].rc; }
        return 0;
    };
    // ---- phase 1: rows up, into fold order, per-fold moments
    int rc = each([&](int g) -> int {
        Part &q = P[g];
        int r = xval_upload_rows(q.c, x, n, p, y, weights, foldid, q.r0, q.r1, o, &q.xd, &q.ld, &q.yd, &q.wd, &q.fd);
        if (r) return r;
        q.L = xval_layout(q.c, q.r1 - q.r0, p, K, npen, nl, weights != nullptr);
        if (ctx_aux(q.c, q.L.total)) return OEMGPU_ERR_HIP;
        return xval_prepare(q.c, q.L, q.xd, q.ld, q.yd, q.wd, q.fd, q.hf);
    });
    if (rc) { release_all(); return rc; }
    // ---- the fold moments of devices 1 .. G-1 into device 0's, in device order (bitwise reproducible); fold sizes on the host
    std::vector<int64_t> fold_tot(K, 0);
    for (int g = 0; g < G; ++g) for (int k = 0; k < K; ++k) fold_tot[k] += P[g].hf[k];
    Part &Z = P[0];
    const size_t mlenK = Z.L.mlen * K, cslenK = Z.L.cslen * K;
    double *peer = (double *)(Z.c->aux + Z.L.a_peer);
    for (int g = 1; g < G && !rc; ++g) {
        Part &q = P[g];
        // the staging area is reused for every peer: the add of peer g-1 must have read it before peer g overwrites it
        if (set_device(Z.c) || hipEventRecord(Z.c->xfer_ev, Z.c->stream) != hipSuccess || set_device(q.c) ||
            hipStreamWaitEvent(q.c->stream, Z.c->xfer_ev, 0) != hipSuccess) { set_error("xval_dense: event ordering failed"); rc = OEMGPU_ERR_HIP; break; }
        rc = host_hand_over(Z.c, peer, q.c, (const double *)(q.c->aux + q.L.a_mf), mlenK);
        if (!rc && weights) rc = host_hand_over(Z.c, peer + mlenK, q.c, (const double *)(q.c->aux + q.L.a_cs), cslenK);
        if (!rc) rc = set_device(Z.c) ? OEMGPU_ERR_HIP : host_add_into(Z.c, (double *)(Z.c->aux + Z.L.a_mf), peer, mlenK);
        if (!rc && weights) rc = host_add_into(Z.c, (double *)(Z.c->aux + Z.L.a_cs), peer + mlenK, cslenK);
    }
    // ---- phase 2 on the first device: the K + 1 fits; the fold coefficients back to every device
    if (!rc) rc = set_device(Z.c) ? OEMGPU_ERR_HIP : xval_solve(Z.c, Z.L, fold_tot.data(), n, standardize, intercept, o, beta, lambda_out, niter, loss, d);
    const size_t blenK = (size_t)K * npen * nl * (p + 1);
    for (int g = 1; g < G && !rc; ++g)
        rc = host_hand_over(P[g].c, (double *)(P[g].c->aux + P[g].L.a_b), Z.c, (const double *)(Z.c->aux + Z.L.a_b), blenK);
    // ---- phase 3: the error triples of every device's rows, merged in device order
    if (!rc) rc = each([&](int g) -> int {
        P[g].tri.assign((size_t)3 * npen * nl, 0.0);
        return xval_cverr(P[g].c, P[g].L, type_measure, o, nullptr, nullptr, P[g].tri.data());
    });
    if (!rc) {
        std::vector<double> all;
        for (int g = 0; g < G; ++g) all.insert(all.end(), P[g].tri.begin(), P[g].tri.end());
        rc = oemgpu_xval_merge(all.data(), G, o, cvm, cvsd);
    }
    release_all();
    return rc;
}

int oemgpu_xval_dense(const double *x, int64_t n, int32_t p, const double *y, const double *weights, const int32_t *foldid, int32_t nfolds,
                      int32_t standardize, int32_t intercept, int32_t type_measure, const oemgpu_opts *o,
                      double *beta, double *lambda_out, int32_t *niter, double *loss, double *d, double *cvm, double *cvsd)
{
    if (!x || !y || !foldid || !o) { set_error("xval_dense: NULL argument"); return OEMGPU_ERR_ARG; }
    int rc = check_opts(o, p, p + (intercept ? 1 : 0));
    if (rc) return rc;
    if (n < 1) { set_error("xval_dense: bad n"); return OEMGPU_ERR_ARG; }
    if (o->ngpus > 1) {
        if (!beta || !lambda_out || !niter || !loss || !d || !cvm || !cvsd) { set_error("xval_dense: NULL argument"); return OEMGPU_ERR_ARG; }
        std::vector<int> dev;
        if ((rc = host_device_list(o, dev)) != 0) return rc;
        rc = xval_check(nullptr, x, y, foldid, n, n, p, nfolds, intercept, type_measure, weights != nullptr, o);
        if (rc) return rc;
        if (n <= p) { set_error("dimension of x larger than number of observations"); return OEMGPU_ERR_UNSUPPORTED; }
        return xval_dense_devices(dev, x, n, p, y, weights, foldid, nfolds, standardize, intercept, type_measure, o, beta, lambda_out,
                                  niter, loss, d, cvm, cvsd);
    }
    oemgpu_ctx *c = ctx_acquire(o->device);
    if (!c) return OEMGPU_ERR_NO_DEVICE;
    double *xd = nullptr, *yd = nullptr, *wd = nullptr;
    int32_t *fd = nullptr;
    int64_t ld = 0;
    rc = xval_upload_rows(c, x, n, p, y, weights, foldid, 0, n, o, &xd, &ld, &yd, &wd, &fd);
    if (!rc) rc = oemgpu_xval_dense_dev(c, xd, n, ld, p, yd, wd, fd, nfolds, standardize, intercept, type_measure, o, beta, lambda_out,
                                        niter, loss, d, cvm, cvsd);
    (void)hipStreamSynchronize(c->stream);
    ctx_release(c);
    return rc;
}

int oemgpu_fit_sparse(int64_t n, int32_t p, const int64_t *colptr, const int32_t *rowidx, const double *values, const double *y,
                      int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                      double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    if (!colptr || !y || !o || !beta || !lambda_out || !niter || !loss || !d) { set_error("fit_sparse: NULL argument"); return OEMGPU_ERR_ARG; }
    const int q = p + (intercept ? 1 : 0);
    int rc = check_opts(o, p, q);             // R prepends the unpenalised group 0 for the intercept slot (ref R/oem.R:296-338); oemSparse scans all groups.size() = q slots (ref src/oem_sparse.h:465)
    if (rc) return rc;
    if (n <= p) {
        // nobs <= nvars: the XXt branch (ref src/oem_sparse.h:607-612, 638-647) is oemBig's, line for line
        if (intercept) {
            set_error("a sparse x with p >= n and an intercept: the reference multiplies the n x p map by a vector of p + 1 entries (src/oem_sparse.h:638-647) -- "
                      "nothing well-formed to reproduce; intercept = FALSE is served");
            return OEMGPU_ERR_UNSUPPORTED;
        }
        const int64_t nnz0 = colptr[p];
        if (nnz0 < 0 || (nnz0 > 0 && (!rowidx || !values))) { set_error("fit_sparse: bad compressed-column arrays"); return OEMGPU_ERR_ARG; }
        std::vector<double> xc;                                         // n <= p rows: the dense copy the wide engine reads
        try { xc.assign((size_t)n * p, 0.0); }
        catch (const std::bad_alloc &) { set_error("sparse x with p >= n: no host memory for the dense %lld x %d copy", (long long)n, p); return OEMGPU_ERR_ARG; }
        for (int j = 0; j < p; ++j)
            for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
                if (rowidx[k] < 0 || rowidx[k] >= n) { set_error("fit_sparse: row index out of range"); return OEMGPU_ERR_ARG; }
                xc[(size_t)j * n + rowidx[k]] = values[k];
            }
        return fit_big_wide_host(xc.data(), n, p, y, standardize, o, beta, lambda_out, niter, loss, d, o->compute_loss != 0);
    }
    const int64_t nnz = colptr[p];
    if (nnz < 0 || (nnz > 0 && (!rowidx || !values))) { set_error("fit_sparse: bad compressed-column arrays"); return OEMGPU_ERR_ARG; }
    // rows per staging tile: the dense tile is capped at 2 GiB, whatever n is
    int64_t rcrows = (int64_t)(2147483648.0 / (8.0 * p)) / 64 * 64;
    if (sw().OEM_SPARSE_TILE_ROWS.set) { const long long t = sw().OEM_SPARSE_TILE_ROWS.num / 64 * 64; if (t >= 64) rcrows = t; }   // test knob: several tiles on small data
    if (rcrows < 64) rcrows = 64;
    if (rcrows > n) rcrows = n;
    const int64_t ld = (rcrows + 1) / 2 * 2;
    // intval = sqrt(mean(diag(XX)) / n) with XX the (standardised) Gram before the division by n (ref src/oem_sparse.h:493-508, 577-578)
    double intval = 1.0;
    int64_t maxcol = 0;
    {
        double xxdiag = 0.0;
        for (int j = 0; j < p; ++j) {
            if (colptr[j + 1] < colptr[j]) { set_error("fit_sparse: colptr must be non-decreasing"); return OEMGPU_ERR_ARG; }
            if (colptr[j + 1] - colptr[j] > maxcol) maxcol = colptr[j + 1] - colptr[j];
            double ss = 0.0;
            for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) ss += values[k] * values[k];
            double cs = ss / ((double)n - 1.0);
            if (cs == 0.0) cs = 1.0;
            xxdiag += standardize ? ss / cs : ss;
        }
        xxdiag /= (double)p;
        if (intercept) intval = std::sqrt(xxdiag / (double)n);
    }
    oemgpu_ctx *c = ctx_acquire(o->device);
    if (!c) return OEMGPU_ERR_NO_DEVICE;
    // Two ways to the moment buffer.  Compressed columns (sparse.hip): p * nnz / 2 LDS gathers, the cost follows the non-zeros --
    // taken up to 2 % density, where it beats the dense pass (n p^2 MFMA flops whatever the density) several times over.
    // Denser: zero-filled row tiles through the FP64-MFMA kernels.  OEM_SPARSE_GRAM=csc|dense forces one (tests compare them).
    bool use_csc = csc_moments_fits(p) && (double)nnz <= 0.02 * (double)n * (double)p && n < ((int64_t)1 << 31);
    if (sw().OEM_SPARSE_GRAM.set) {
        const char *ev = sw().OEM_SPARSE_GRAM.str;
        if (!strcmp(ev, "csc") && csc_moments_fits(p)) use_csc = true;
        if (!strcmp(ev, "dense")) use_csc = false;
    }
    double *xd = nullptr, *yd = nullptr, *vd = nullptr;
    int64_t *cd = nullptr;
    int32_t *rd = nullptr;
    void *cwork = nullptr;
    hipError_t e = hipSuccess;
    {                                                       // every staging buffer out of the context's grow-only input buffer
        Bump S;
        const size_t a_x = S.take(use_csc ? csc_moments_work_bytes(n, p) : sizeof(double) * (size_t)ld * p),
                     a_y = S.take(sizeof(double) * (size_t)(n + 2)),
                     a_c = S.take(sizeof(int64_t) * (size_t)(p + 1)), a_r = S.take(sizeof(int32_t) * (size_t)(nnz + 1)),
                     a_v = S.take(sizeof(double) * (size_t)(nnz + 1));
        if (ctx_grow(c, &c->xres, &c->xres_bytes, S.off)) { ctx_release(c); return OEMGPU_ERR_HIP; }
        xd = (double *)(c->xres + a_x); cwork = c->xres + a_x; yd = (double *)(c->xres + a_y); cd = (int64_t *)(c->xres + a_c);
        rd = (int32_t *)(c->xres + a_r); vd = (double *)(c->xres + a_v);
    }
    if (e == hipSuccess && !use_csc) e = hipMemsetAsync(xd, 0, sizeof(double) * (size_t)ld * p, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(yd, y, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(cd, colptr, sizeof(int64_t) * (size_t)(p + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(rd, rowidx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(vd, values, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("fit_sparse: device staging failed: %s", hipGetErrorString(e)); rc = OEMGPU_ERR_HIP; }
    // The Gram of a sparse X through the dense FP64-MFMA pass: row tiles of at most 2 GiB are zeroed, filled with the tile's
    // non-zeros (one 8-byte store each) and read back by the moment kernels; the tiles' moments are added in row order.  At the
    // densities of the reference's examples (1 %) the whole x is one tile of a few hundred MB and a fraction of a millisecond.  The
    // matrix pipe does not care about zeros: the cost is that of a dense n x p pass whatever the density, with no dense copy of x
    // beyond one tile.  (A compressed-column Gram kernel would win below ~0.1 % density; it is not built.)
    if (!rc) {
        const GramPlan plmax = gram_plan_bound(rcrows, p, c->num_cu);
        Bump B;
        const size_t mlen = (size_t)oemgpu_moments_len(p);
        const size_t a_mom = B.take(mlen * 8), a_tmp = B.take(mlen * 8), a_t = B.take(plmax.tpart_doubles * 8), a_v = B.take(plmax.vpart_doubles * 8);
        const size_t a_xx = B.take((size_t)q * q * 8), a_xy = B.take((size_t)q * 8), a_st = B.take((size_t)stats_len(p) * 8);
        rc = ctx_reserve(c, B.off + paths_ws_bytes(p, q, o) + 4096) ? OEMGPU_ERR_HIP : 0;
        double *mom = (double *)(c->ws + a_mom), *mtmp = (double *)(c->ws + a_tmp);
        if (!rc && hipMemsetAsync(mom, 0, mlen * 8, c->stream) != hipSuccess) rc = OEMGPU_ERR_HIP;
        if (!rc && use_csc) rc = launch_csc_moments(c->stream, cd, rd, vd, yd, n, p, cwork, mom);
        for (int64_t r0 = 0; r0 < n && !rc && !use_csc; r0 += rcrows) {
            const int64_t r1 = r0 + rcrows < n ? r0 + rcrows : n, nr = r1 - r0;
            if (hipMemsetAsync(xd, 0, sizeof(double) * (size_t)ld * p, c->stream) != hipSuccess) { set_error("fit_sparse: memset failed"); rc = OEMGPU_ERR_HIP; break; }
            if (nnz > 0) {
                hipLaunchKernelGGL(csc_densify_kernel, dim3((unsigned)((maxcol + 255) / 256), p), dim3(256), 0, c->stream, cd, rd, vd, r0, r1, ld, xd);
                if (hipGetLastError() != hipSuccess) { set_error("fit_sparse: densify launch failed"); rc = OEMGPU_ERR_HIP; break; }
            }
            const GramPlan pl = gram_plan(nr, p, c->num_cu);
            if (pl.tpart_doubles > plmax.tpart_doubles || pl.vpart_doubles > plmax.vpart_doubles) { set_error("internal: tile plan larger than its scratch"); rc = OEMGPU_ERR_INTERNAL; break; }
            rc = shard_moments(c, pl, xd, nr, ld, yd + r0, nullptr, (double *)(c->ws + a_t), (double *)(c->ws + a_v), mtmp);
            if (!rc) hipLaunchKernelGGL(accumulate_kernel, dim3(64), dim3(256), 0, c->stream, mom, mtmp, mlen);
        }
        double *xx = (double *)(c->ws + a_xx), *xy = (double *)(c->ws + a_xy), *st = (double *)(c->ws + a_st);
        if (!rc) rc = launch_finalize(c->stream, (double *)(c->ws + a_mom), nullptr, p, OEMGPU_SEM_BIG, standardize, intercept, xx, xy, st);
        std::vector<double> sf;
        if (!rc && intercept) {
            hipLaunchKernelGGL(scale_intercept_kernel, dim3((q + 255) / 256), dim3(256), 0, c->stream, xx, xy, q, intval);
            sf.assign(q, 1.0); sf[0] = 1.0 / intval;              // get_beta multiplies the intercept slot by intval, in place (ref :897-900)
        }
        if (!rc) rc = run_paths(c, B, xx, xy, st, p, q, SEM_SPARSE, standardize, intercept, o, intercept ? sf.data() : nullptr, beta,
                                lambda_out, niter, loss, d);
    }
    (void)hipStreamSynchronize(c->stream);
    ctx_release(c);
    return rc;
}

}  // extern "C"
#pragma GCC visibility pop
