// hoststream.hip -- the host-resident entry points: what `.Call("oem_fit_dense")` / `.Call("oem_fit_big")` hand over is a
// pageable host matrix, so the call is bounded by how fast its rows reach HBM, and by how many PCIe links carry them.
//
//   rows      the reference's own row blocks: floor(n / G) rows per device, the remainder on the last
//             (ref src/oem_dense.h:328,343 across threads; src/oem_big.h:329-358 across slices)
//   staging   per device T host threads ("lanes"): pageable rows -> memcpy -> a pinned bounce slot (two per lane) ->
//             hipMemcpy(2D)Async on the lane's own copy stream.  The memcpy of slot k+1 overlaps the DMA of slot k, and T
//             lanes keep the link busy while each thread copies at DRAM speed.
//   blocks    a device's rows are cut into row blocks (<= 256 MiB); the MFMA moment pass over block b runs on the
//             context's compute stream while the lanes stage block b+1.  Block moments are added in block order.
//             When the slice fits in HBM it stays resident (a shifted redo then re-reads HBM, not the host); otherwise two
//             block buffers are recycled (events order "moment pass finished reading" before "next DMA overwrites").
//   sum       the G moment buffers ((p+2)^2 fp64: c1 83 KB, c5 532 KB) are copied to the first device (peer copy over
//             xGMI) and added in device order -- bitwise reproducible, no atomics -- then ONE solve; with several
//             penalties on the launch-per-iteration engines (p + intercept > 288) the penalties, which are independent
//             cold starts (ref src/oem_dense.cpp:206-246), are dealt round-robin to the devices instead.
//   steady state: contexts (streams, events, workspace, pinned slots, block buffers) live in the process-wide cache:
//             a repeated call allocates nothing (oemgpu_last_host_stats()[7] == 0).
#include "ctx.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace oemgpu {

std::atomic<long> g_alloc_count{0};                  // device / pinned allocations and stream / event creations, process-wide
static thread_local double g_host_stats[OEMGPU_NHOSTSTATS] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

namespace {

typedef std::chrono::steady_clock Clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

__global__ void add_into_kernel(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

struct HostPiece {            // rows [0, rows) of a column-major host matrix with column stride ldx, and their responses
    const double *x; int64_t ldx; const double *y; int64_t rows;
};

struct Block {                // one row block of one piece and where it lives on the device
    int piece; int64_t r0, nr;
    double *xd; int64_t ld; double *yd;
};

struct Unit {                 // one bounce-slot load: columns [j0, j1) x rows [s0, s1) of a block (j0 == -1: the y rows)
    int j0, j1; int64_t s0, s1;
};

class Barrier {               // reusable barrier for the lanes of one device (C++17: no std::barrier)
    std::mutex mu; std::condition_variable cv; int n, waiting = 0; long gen = 0;
public:
    explicit Barrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

size_t env_size(const Switch &s, size_t dflt) { return (s.set && s.num > 0) ? (size_t)s.num : dflt; }

int lanes_prepare(oemgpu_ctx *c, int T, size_t slot_bytes)
{
    if (set_device(c)) return OEMGPU_ERR_HIP;
    if (c->slot_bytes != slot_bytes) {                  // slot size changed (test knob): drop the old slots
        for (oemgpu_lane &l : c->lanes)
            for (int k = 0; k < 2; ++k) if (l.slot[k]) { (void)hipHostFree(l.slot[k]); l.slot[k] = nullptr; }
        c->slot_bytes = slot_bytes;
    }
    if ((int)c->lanes.size() < T) c->lanes.resize(T);
    for (int t = 0; t < T; ++t) {
        oemgpu_lane &l = c->lanes[t];
        if (!l.s) { OEM_HIP(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking)); ++g_alloc_count; }
        for (int k = 0; k < 2; ++k) {
            if (!l.slot[k]) { OEM_HIP(hipHostMalloc((void **)&l.slot[k], slot_bytes, hipHostMallocDefault)); ++g_alloc_count; }
            if (!l.slot_ev[k]) { OEM_HIP(hipEventCreateWithFlags(&l.slot_ev[k], hipEventDisableTiming)); ++g_alloc_count; }
            if (!l.blk_ev[k]) { OEM_HIP(hipEventCreateWithFlags(&l.blk_ev[k], hipEventDisableTiming)); ++g_alloc_count; }
            l.slot_used[k] = false;
        }
    }
    for (int k = 0; k < 2; ++k)
        if (!c->done_ev[k]) { OEM_HIP(hipEventCreateWithFlags(&c->done_ev[k], hipEventDisableTiming)); ++g_alloc_count; }
    if (!c->xfer_ev) { OEM_HIP(hipEventCreateWithFlags(&c->xfer_ev, hipEventDisableTiming)); ++g_alloc_count; }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// One device's share of the rows: layout, staging, block moments.
struct DevJob {
    oemgpu_ctx *c = nullptr;
    std::vector<HostPiece> pieces;
    int p = 0;
    int T = 1;
    size_t slot_bytes = 0;
    int64_t nrows = 0;
    bool resident = true;
    bool contiguous = false;         // one piece laid out as ONE matrix (host_upload_resident): blocks share ld
    bool tight = false;              // ... with ld = the row count exactly (no padding to 16 rows): a unit of whole columns of a
                                     // contiguous host matrix is then ONE linear copy (the p >= n upload: 100,000 columns of 200 rows
                                     // as a pitched 2-D copy took seconds)
    std::vector<Block> blocks;
    // accumulators (c->acc): moments | block moments | sums | block sums | peer staging
    double *msum = nullptr, *mtmp = nullptr, *ssum = nullptr, *stmp = nullptr, *peer = nullptr;
    // per-pass state
    bool upload = true, want_sums = false, want_moments = true;
    const double *shift_sums = nullptr;          // device pointer: moments about the shift these sums define; nullptr: about 0
    const oemgpu_opts *o = nullptr;
    bool poll = false;                           // this job's lane 0 runs on the CALLING thread: the only one that may poll o->interrupt
    std::atomic<int> *stop = nullptr;            // shared by the devices of one call: somebody failed or the caller interrupted
    std::atomic<int> abort{0};                   // OEMGPU_ERR_* of the first failure (or INTERRUPTED)
    std::string err;
    std::mutex err_mu;
    size_t bytes_staged = 0;

    void fail(int rc, const char *msg)
    {
        std::lock_guard<std::mutex> lk(err_mu);
        if (!abort.load()) { err = msg; abort.store(rc); }
        if (stop) stop->store(1);
    }
    bool stopped() const { return abort.load() != 0 || (stop && stop->load() != 0); }
};

#define JOB_HIP(J, call)                                                                              \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            char b__[400];                                                                            \
            snprintf(b__, sizeof b__, "%s failed: %s (device %d)", #call, hipGetErrorString(e__), (J).c->device); \
            (J).fail(OEMGPU_ERR_HIP, b__);                                                            \
            return;                                                                                   \
        }                                                                                             \
    } while (0)

// rows per block: a block is at most `block_bytes`, a multiple of 64 rows, and one column of it fits a bounce slot
int64_t block_rows(int p, size_t block_bytes, size_t slot_bytes)
{
    int64_t br = (int64_t)(block_bytes / (8 * (size_t)p));
    const int64_t cap = (int64_t)(slot_bytes / 8);
    if (br > cap) br = cap;
    br = br / 64 * 64;
    return br < 64 ? 64 : br;
}

// layout of the job's rows on its device; reserves c->xres / c->acc
int job_layout(DevJob &J, size_t resident_cap)
{
    oemgpu_ctx *c = J.c;
    const int p = J.p;
    const size_t block_bytes = env_size(sw().OEMGPU_BLOCK_BYTES, (size_t)256 << 20);
    const int64_t BR = block_rows(p, block_bytes, J.slot_bytes);
    J.blocks.clear();
    J.nrows = 0;
    int64_t ldsum = 0, maxnr = 0;
    for (size_t k = 0; k < J.pieces.size(); ++k) {
        const int64_t rows = J.pieces[k].rows;
        J.nrows += rows;
        for (int64_t r0 = 0; r0 < rows; r0 += BR) {
            Block b; b.piece = (int)k; b.r0 = r0; b.nr = rows - r0 < BR ? rows - r0 : BR;
            b.xd = nullptr; b.ld = 0; b.yd = nullptr;
            J.blocks.push_back(b);
            if (b.nr > maxnr) maxnr = b.nr;
        }
        ldsum += J.tight ? rows : (rows + 15) / 16 * 16;
    }
    const size_t res_bytes = sizeof(double) * ((size_t)ldsum * p + (size_t)ldsum + 64);
    J.resident = J.contiguous || res_bytes <= resident_cap;
    const int64_t ldb = (maxnr + 15) / 16 * 16;
    Bump X;
    size_t a_x[2], a_y[2];
    if (J.resident) { a_x[0] = X.take(sizeof(double) * (size_t)ldsum * p); a_y[0] = X.take(sizeof(double) * ((size_t)ldsum + 2) + sizeof(int32_t) * ((size_t)ldsum + 64)); a_x[1] = a_y[1] = 0; }
    else for (int k = 0; k < 2; ++k) { a_x[k] = X.take(sizeof(double) * (size_t)ldb * p); a_y[k] = X.take(sizeof(double) * ((size_t)ldb + 2)); }
    if (ctx_grow(c, &c->xres, &c->xres_bytes, X.off)) return OEMGPU_ERR_HIP;
    int64_t roff = 0;
    int lastpiece = -1;
    int64_t piece_base = 0;
    for (size_t i = 0; i < J.blocks.size(); ++i) {
        Block &b = J.blocks[i];
        if (J.resident) {
            if (b.piece != lastpiece) { piece_base = roff; roff += J.tight ? J.pieces[b.piece].rows : (J.pieces[b.piece].rows + 15) / 16 * 16; lastpiece = b.piece; }
            b.ld = ldsum;
            b.xd = (double *)(c->xres + a_x[0]) + piece_base + b.r0;
            b.yd = (double *)(c->xres + a_y[0]) + piece_base + b.r0;
        } else {
            b.ld = ldb;
            b.xd = (double *)(c->xres + a_x[i & 1]);
            b.yd = (double *)(c->xres + a_y[i & 1]);
        }
    }
    const size_t mlen = (size_t)oemgpu_moments_len(p), slen = (size_t)oemgpu_sums_len(p);
    Bump A;
    const size_t a_ms = A.take(mlen * 8), a_mt = A.take(mlen * 8), a_ss = A.take(slen * 8), a_st = A.take(slen * 8),
                 a_peer = A.take((mlen > slen ? mlen : slen) * 8);
    if (ctx_grow(c, &c->acc, &c->acc_bytes, A.off)) return OEMGPU_ERR_HIP;
    J.msum = (double *)(c->acc + a_ms); J.mtmp = (double *)(c->acc + a_mt);
    J.ssum = (double *)(c->acc + a_ss); J.stmp = (double *)(c->acc + a_st); J.peer = (double *)(c->acc + a_peer);
    return 0;
}

void block_units(const DevJob &J, const Block &b, std::vector<Unit> &u)
{
    u.clear();
    const int64_t seg = (int64_t)(J.slot_bytes / 8) < b.nr ? (int64_t)(J.slot_bytes / 8) : b.nr;      // rows per unit
    int cols = (int)(J.slot_bytes / (8 * (size_t)seg));
    if (cols < 1) cols = 1;
    // at least ~2 units per lane and block, so that every lane has something to overlap
    const int want = (J.p + 2 * J.T - 1) / (2 * J.T);
    if (cols > want) cols = want < 1 ? 1 : want;
    for (int64_t s0 = 0; s0 < b.nr; s0 += seg) {
        const int64_t s1 = s0 + seg < b.nr ? s0 + seg : b.nr;
        for (int j0 = 0; j0 < J.p; j0 += cols) { Unit t; t.j0 = j0; t.j1 = j0 + cols < J.p ? j0 + cols : J.p; t.s0 = s0; t.s1 = s1; u.push_back(t); }
        Unit ty; ty.j0 = -1; ty.j1 = 0; ty.s0 = s0; ty.s1 = s1; u.push_back(ty);
    }
}

// lane t of the device's pipeline (t == 0 also drives the compute stream)
void lane_main(DevJob &J, Barrier &bar, int t)
{
    oemgpu_ctx *c = J.c;
    if (hipSetDevice(c->device) != hipSuccess) J.fail(OEMGPU_ERR_HIP, "hipSetDevice failed in a staging lane");
    oemgpu_lane &L = c->lanes[t];
    std::vector<Unit> units;
    const size_t mlen = (size_t)oemgpu_moments_len(J.p), slen = (size_t)oemgpu_sums_len(J.p);
    int slot = 0;
    size_t staged = 0;
    auto stage_block = [&](size_t bi) {
        const Block &b = J.blocks[bi];
        const HostPiece &P = J.pieces[b.piece];
        if (!J.resident && bi >= 2) JOB_HIP(J, hipStreamWaitEvent(L.s, c->done_ev[bi & 1], 0));   // block bi-2's pass has read the buffer
        block_units(J, b, units);
        for (size_t ui = (size_t)t; ui < units.size(); ui += (size_t)J.T) {
            const Unit &u = units[ui];
            const int64_t w = u.s1 - u.s0;
            if (L.slot_used[slot]) JOB_HIP(J, hipEventSynchronize(L.slot_ev[slot]));
            char *sl = L.slot[slot];
            if (u.j0 < 0) {
                memcpy(sl, P.y + b.r0 + u.s0, sizeof(double) * (size_t)w);
                JOB_HIP(J, hipMemcpyAsync(b.yd + u.s0, sl, sizeof(double) * (size_t)w, hipMemcpyHostToDevice, L.s));
                staged += sizeof(double) * (size_t)w;
            } else {
                const int nc = u.j1 - u.j0;
                for (int j = 0; j < nc; ++j)
                    memcpy(sl + sizeof(double) * (size_t)w * j, P.x + (size_t)(u.j0 + j) * P.ldx + b.r0 + u.s0, sizeof(double) * (size_t)w);
                double *dst = b.xd + (size_t)u.j0 * b.ld + u.s0;
                if (nc == 1 || w == b.ld) JOB_HIP(J, hipMemcpyAsync(dst, sl, sizeof(double) * (size_t)w * nc, hipMemcpyHostToDevice, L.s));   // whole columns, no pitch
                else JOB_HIP(J, hipMemcpy2DAsync(dst, sizeof(double) * (size_t)b.ld, sl, sizeof(double) * (size_t)w, sizeof(double) * (size_t)w,
                                                 (size_t)nc, hipMemcpyHostToDevice, L.s));
                staged += sizeof(double) * (size_t)w * nc;
            }
            JOB_HIP(J, hipEventRecord(L.slot_ev[slot], L.s));
            L.slot_used[slot] = true;
            slot ^= 1;
        }
        JOB_HIP(J, hipEventRecord(L.blk_ev[bi & 1], L.s));
    };
    auto pass_block = [&](size_t bi) {            // lane 0: the moment pass over block bi on the compute stream
        const Block &b = J.blocks[bi];
        if (J.upload) for (int k = 0; k < J.T; ++k) JOB_HIP(J, hipStreamWaitEvent(c->stream, c->lanes[k].blk_ev[bi & 1], 0));
        if (J.want_sums) {
            if (launch_shift_sums(c->stream, b.xd, b.nr, b.ld, J.p, b.yd, bi == 0 ? J.ssum : J.stmp)) { J.fail(OEMGPU_ERR_HIP, oemgpu_last_error()); return; }
            if (bi > 0) hipLaunchKernelGGL(add_into_kernel, dim3(8), dim3(256), 0, c->stream, J.ssum, J.stmp, slen);
        }
        if (J.want_moments) {
            const int rc = oemgpu_moments_dev(c, b.xd, b.nr, b.ld, J.p, b.yd, J.shift_sums, bi == 0 ? J.msum : J.mtmp);
            if (rc) { J.fail(rc, oemgpu_last_error()); return; }
            if (bi > 0) hipLaunchKernelGGL(add_into_kernel, dim3(64), dim3(256), 0, c->stream, J.msum, J.mtmp, mlen);
        }
        if (!J.resident) JOB_HIP(J, hipEventRecord(c->done_ev[bi & 1], c->stream));
        if (J.poll && J.o && J.o->interrupt && J.o->interrupt(J.o->interrupt_arg)) J.fail(OEMGPU_ERR_INTERRUPTED, "interrupted by the caller");
    };
    for (size_t bi = 0; bi < J.blocks.size(); ++bi) {
        if (J.upload && !J.stopped()) stage_block(bi);
        if (J.T > 1) bar.wait();                  // every lane's copies of block bi are enqueued, their block events recorded
        if (t == 0 && !J.stopped()) pass_block(bi);
    }
    if (J.upload) {                               // the slots are free again when the call returns
        for (int k = 0; k < 2; ++k) if (L.slot_used[k]) { (void)hipEventSynchronize(L.slot_ev[k]); L.slot_used[k] = false; }
        std::lock_guard<std::mutex> lk(J.err_mu);
        J.bytes_staged += staged;
    }
}

// one pass over the job's blocks: (upload) + (sample sums) + (moments about 0 or about `shift_sums`)
void run_pass(DevJob &J, bool upload, bool want_sums, bool want_moments, const double *shift_sums)
{
    J.upload = upload; J.want_sums = want_sums; J.want_moments = want_moments; J.shift_sums = shift_sums;
    if (J.blocks.empty()) {                       // a device without rows (n < G): zero moments
        if (hipSetDevice(J.c->device) != hipSuccess) { J.fail(OEMGPU_ERR_HIP, "hipSetDevice failed"); return; }
        (void)hipMemsetAsync(J.msum, 0, sizeof(double) * (size_t)oemgpu_moments_len(J.p), J.c->stream);
        (void)hipMemsetAsync(J.ssum, 0, sizeof(double) * (size_t)oemgpu_sums_len(J.p), J.c->stream);
        return;
    }
    const int T = upload ? J.T : 1;
    const int keepT = J.T;
    J.T = T;
    Barrier bar(T);
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(lane_main, std::ref(J), std::ref(bar), t);
    lane_main(J, bar, 0);
    for (auto &x : th) x.join();
    J.T = keepT;
}

// Can `to` read `from`'s memory directly (xGMI / PCIe peer access)?  Asked once per ordered pair and cached; peer access is enabled
// on first use.  OEMGPU_NO_PEER=1 answers "no" for every pair -- the same device included -- so the host-staged route below can be
// exercised on a one-GPU box.
bool peers(int to, int from)
{
    static std::mutex mu;
    static signed char known[64][64];                    // 0 unknown, 1 yes, -1 no
    const bool never = sw().OEMGPU_NO_PEER.set;
    if (never) return false;
    if (to == from) return true;
    if (to < 0 || from < 0 || to >= 64 || from >= 64) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (known[to][from] == 0) {
        int can = 0;
        bool ok = hipDeviceCanAccessPeer(&can, to, from) == hipSuccess && can != 0;
        if (ok) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            if (hipSetDevice(to) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(from, 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                (void)hipGetLastError();                 // "already enabled" is not an error of ours
            } else ok = false;
            (void)hipSetDevice(cur);
        }
        known[to][from] = ok ? 1 : -1;
    }
    return known[to][from] > 0;
}

std::atomic<long> g_host_staged{0};                      // hand-overs that went through the host (oemgpu_last_host_stats()[8])

// dst (on device cd) <- src (on device cs), ordered after everything on `from` so far and before anything later on `to`.
// Peers: one hipMemcpyPeerAsync over xGMI on from's stream.  Not peers (hipDeviceCanAccessPeer says no, or OEMGPU_NO_PEER):
// device -> pinned host buffer of `to` -> device, each leg followed by a stream synchronisation (the buffers are a few hundred
// KB once per call; the bounce buffer is then free again whatever the caller does next).
int hand_over(oemgpu_ctx *to, double *dst, oemgpu_ctx *from, const double *src, size_t doubles)
{
    OEM_HIP(hipSetDevice(from->device));
    if (!peers(to->device, from->device)) {
        if (doubles * 8 > to->xfer_host_bytes) {
            if (to->xfer_host) { OEM_HIP(hipHostFree(to->xfer_host)); to->xfer_host = nullptr; to->xfer_host_bytes = 0; }
            const size_t want = (doubles * 8 + 4095) / 4096 * 4096;
            OEM_HIP(hipHostMalloc((void **)&to->xfer_host, want, hipHostMallocPortable)); ++g_alloc_count;
            to->xfer_host_bytes = want;
        }
        OEM_HIP(hipMemcpyAsync(to->xfer_host, src, doubles * 8, hipMemcpyDeviceToHost, from->stream));
        OEM_HIP(hipStreamSynchronize(from->stream));
        OEM_HIP(hipSetDevice(to->device));
        OEM_HIP(hipMemcpyAsync(dst, to->xfer_host, doubles * 8, hipMemcpyHostToDevice, to->stream));
        OEM_HIP(hipStreamSynchronize(to->stream));
        ++g_host_staged;
        return 0;
    }
    if (to->device == from->device) OEM_HIP(hipMemcpyAsync(dst, src, doubles * 8, hipMemcpyDeviceToDevice, from->stream));
    else OEM_HIP(hipMemcpyPeerAsync(dst, to->device, src, from->device, doubles * 8, from->stream));
    OEM_HIP(hipEventRecord(from->xfer_ev, from->stream));
    OEM_HIP(hipSetDevice(to->device));
    OEM_HIP(hipStreamWaitEvent(to->stream, from->xfer_ev, 0));
    return 0;
}

// buf_0 <- buf_0 + buf_1 + ... + buf_{G-1} on device 0, in device order (bitwise reproducible)
int sum_on_first(std::vector<DevJob> &J, bool sums)
{
    const int G = (int)J.size();
    const size_t len = sums ? (size_t)oemgpu_sums_len(J[0].p) : (size_t)oemgpu_moments_len(J[0].p);
    for (int g = 1; g < G; ++g) {
        double *src = sums ? J[g].ssum : J[g].msum, *dst0 = sums ? J[0].ssum : J[0].msum;
        // the staging buffer is reused for every peer: the add of peer g-1 must have read it before peer g overwrites it
        OEM_HIP(hipSetDevice(J[0].c->device));
        OEM_HIP(hipEventRecord(J[0].c->xfer_ev, J[0].c->stream));
        OEM_HIP(hipSetDevice(J[g].c->device));
        OEM_HIP(hipStreamWaitEvent(J[g].c->stream, J[0].c->xfer_ev, 0));
        int rc = hand_over(J[0].c, J[0].peer, J[g].c, src, len);
        if (rc) return rc;
        hipLaunchKernelGGL(add_into_kernel, dim3(64), dim3(256), 0, J[0].c->stream, dst0, J[0].peer, len);
        OEM_HIP(hipGetLastError());
    }
    return 0;
}

int broadcast_from_first(std::vector<DevJob> &J, bool sums)
{
    const size_t len = sums ? (size_t)oemgpu_sums_len(J[0].p) : (size_t)oemgpu_moments_len(J[0].p);
    for (size_t g = 1; g < J.size(); ++g) {
        int rc = hand_over(J[g].c, sums ? J[g].ssum : J[g].msum, J[0].c, sums ? J[0].ssum : J[0].msum, len);
        if (rc) return rc;
    }
    return 0;
}

int device_list(const oemgpu_opts *o, std::vector<int> &dev)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available (liboemgpu has no CPU fallback)"); return OEMGPU_ERR_NO_DEVICE; }
    const int G = o->ngpus > 1 ? o->ngpus : 1;
    dev.resize(G);
    int first = o->device;
    if (first < 0) { if (G > 1 || hipGetDevice(&first) != hipSuccess) first = 0; }
    for (int g = 0; g < G; ++g) {
        dev[g] = (G > 1 && o->devices) ? o->devices[g] : first + g;
        if (dev[g] < 0 || dev[g] >= ndev) { set_error("device %d out of range (%d devices)", dev[g], ndev); return OEMGPU_ERR_ARG; }
    }
    return 0;
}

struct SubOpts {              // the penalties k = g, g + G, ... of a call, as a call of their own
    oemgpu_opts o;
    std::vector<int32_t> pen, idx;
    std::vector<double> lam, beta, lambda_out, loss;
    std::vector<int32_t> niter;
    double d = 0.0;
};

void make_subopts(const oemgpu_opts *o, int g, int G, int rows, SubOpts &S)
{
    const int nl = (o->lambda_user && o->nlambda_user > 0) ? o->nlambda_user : o->nlambda;
    S.o = *o;
    for (int k = g; k < o->npen; k += G) { S.idx.push_back(k); S.pen.push_back(o->penalty[k]); }
    const int m = (int)S.idx.size();
    S.o.npen = m; S.o.penalty = S.pen.data();
    if (o->lambda_user && o->nlambda_user > 0) {
        S.lam.resize((size_t)m * nl);
        for (int i = 0; i < m; ++i) memcpy(S.lam.data() + (size_t)i * nl, o->lambda_user + (size_t)S.idx[i] * nl, sizeof(double) * nl);
        S.o.lambda_user = S.lam.data();
    }
    S.beta.assign((size_t)m * nl * rows, 0.0); S.lambda_out.assign((size_t)m * nl, 0.0); S.loss.assign((size_t)m * nl, 0.0);
    S.niter.assign((size_t)m * nl, 0);
}

// One solve of the summed moments (on J[0]), or -- several penalties on the launch-per-iteration engines, several devices --
// the penalties dealt round-robin to the devices, every device solving its own from a copy of the moments.
int solve_summed(std::vector<DevJob> &J, bool with_sums, int p, int sem, int standardize, int intercept, const oemgpu_opts *o,
                 double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    const int G = (int)J.size();
    const int q = p + ((sem != OEMGPU_SEM_DENSE && intercept) ? 1 : 0);
    const bool split = G > 1 && o->npen > 1 && q > SMALL_P_MAX && !sw().OEMGPU_NO_PENALTY_SPLIT.set;
    if (!split) {
        OEM_HIP(hipSetDevice(J[0].c->device));
        return oemgpu_solve_moments_dev(J[0].c, J[0].msum, with_sums ? J[0].ssum : nullptr, p, sem, standardize, intercept, o, beta, lambda_out,
                                        niter, loss, d);
    }
    int rc = broadcast_from_first(J, false);
    if (!rc && with_sums) rc = broadcast_from_first(J, true);
    if (rc) return rc;
    const int nl = (o->lambda_user && o->nlambda_user > 0) ? o->nlambda_user : o->nlambda;
    const int rows = p + 1;
    const int used = G < o->npen ? G : o->npen;
    std::vector<SubOpts> S(used);
    std::vector<int> rcs(used, 0);
    std::vector<std::string> errs(used);
    std::vector<std::thread> th;
    auto solve_one = [&](int g) {
        if (hipSetDevice(J[g].c->device) != hipSuccess) { rcs[g] = OEMGPU_ERR_HIP; errs[g] = "hipSetDevice failed"; return; }
        rcs[g] = oemgpu_solve_moments_dev(J[g].c, J[g].msum, with_sums ? J[g].ssum : nullptr, p, sem, standardize, intercept, &S[g].o,
                                          S[g].beta.data(), S[g].lambda_out.data(), S[g].niter.data(), S[g].loss.data(), &S[g].d);
        if (rcs[g]) errs[g] = oemgpu_last_error();
    };
    for (int g = 0; g < used; ++g) {
        make_subopts(o, g, G, rows, S[g]);
        if (g > 0) S[g].o.interrupt = nullptr;      // only the calling thread may poll the caller (R's API is single-threaded)
    }
    for (int g = 1; g < used; ++g) th.emplace_back(solve_one, g);
    solve_one(0);
    for (auto &t : th) t.join();
    for (int g = 0; g < used; ++g) if (rcs[g]) { set_error("device %d: %s", J[g].c->device, errs[g].c_str()); return rcs[g]; }
    for (int g = 0; g < used; ++g)
        for (size_t i = 0; i < S[g].idx.size(); ++i) {
            const size_t k = (size_t)S[g].idx[i];
            memcpy(beta + k * nl * rows, S[g].beta.data() + i * nl * rows, sizeof(double) * (size_t)nl * rows);
            memcpy(lambda_out + k * nl, S[g].lambda_out.data() + i * nl, sizeof(double) * nl);
            memcpy(loss + k * nl, S[g].loss.data() + i * nl, sizeof(double) * nl);
            memcpy(niter + k * nl, S[g].niter.data() + i * nl, sizeof(int32_t) * nl);
        }
    *d = S[0].d;          // every device saw the same moments: d and the shift verdict (J[0].c->shift_advised) agree
    return 0;
}

// the driver behind oemgpu_fit_dense / oemgpu_fit_big: `rows` = the concatenated pieces, split over the devices
int host_fit(const std::vector<HostPiece> &all, int64_t n, int32_t p, int sem, int32_t standardize, int32_t intercept,
             const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    const Clock::time_point t_call = Clock::now();
    const long allocs0 = g_alloc_count.load(), staged0 = g_host_staged.load();
    for (double &v : g_host_stats) v = 0.0;
    std::vector<int> dev;
    int rc = device_list(o, dev);
    if (rc) return rc;
    const int G = (int)dev.size();
    int T = o->upload_threads > 0 ? o->upload_threads : (int)env_size(sw().OEMGPU_UPLOAD_THREADS, 8);
    if (T > 64) T = 64;
    const size_t slot_bytes = env_size(sw().OEMGPU_SLOT_BYTES, (size_t)4 << 20) / 4096 * 4096 + 4096;
    std::vector<DevJob> J(G);
    std::atomic<int> stop{0};
    std::vector<oemgpu_ctx *> held;
    auto release_all = [&]() { for (oemgpu_ctx *c : held) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); ctx_release(c); } };
    // ---- rows of device g: [g * floor(n / G), ...), the remainder on the last (ref src/oem_dense.h:328,343), as sub-ranges of the pieces
    for (int g = 0; g < G; ++g) {
        int64_t r0, r1;
        oemgpu_row_split(n, G, g, &r0, &r1);
        int64_t base = 0;
        for (const HostPiece &P : all) {
            const int64_t lo = r0 > base ? r0 : base, hi = r1 < base + P.rows ? r1 : base + P.rows;
            if (hi > lo) { HostPiece s; s.x = P.x + (lo - base); s.ldx = P.ldx; s.y = P.y + (lo - base); s.rows = hi - lo; J[g].pieces.push_back(s); }
            base += P.rows;
        }
        oemgpu_ctx *c = ctx_acquire(dev[g]);
        if (!c) { release_all(); return OEMGPU_ERR_NO_DEVICE; }
        held.push_back(c);
        J[g].c = c; J[g].p = p; J[g].T = T; J[g].slot_bytes = slot_bytes; J[g].o = o; J[g].stop = &stop; J[g].poll = g == 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { free_b = total_b = (size_t)64 << 30; }
        size_t cap = env_size(sw().OEMGPU_RESIDENT_BYTES, total_b / 2);
        if (cap > free_b + c->xres_bytes) cap = free_b + c->xres_bytes;
        if ((rc = lanes_prepare(c, T, slot_bytes)) != 0 || (rc = job_layout(J[g], cap)) != 0) { release_all(); return rc; }
    }
    // ---- pass 1 on every device side by side: upload + moments about 0 (+ the sample sums the shifted redo would need)
    const Clock::time_point t_up = Clock::now();
    const bool dense = sem == OEMGPU_SEM_DENSE;
    {
        std::vector<std::thread> th;
        for (int g = 1; g < G; ++g) th.emplace_back(run_pass, std::ref(J[g]), true, dense, true, (const double *)nullptr);
        run_pass(J[0], true, dense, true, nullptr);
        for (auto &t : th) t.join();
    }
    auto first_error = [&]() -> int {
        for (int g = 0; g < G; ++g) if (J[g].abort.load()) { set_error("%s", J[g].err.c_str()); return J[g].abort.load(); }
        if (stop.load()) { set_error("stopped"); return OEMGPU_ERR_INTERNAL; }
        return 0;
    };
    if ((rc = first_error()) != 0) { release_all(); return rc; }
    if ((rc = sum_on_first(J, false)) != 0) { release_all(); return rc; }
    double up_ms = ms_since(t_up);
    // ---- solve; a verdict "some column has |mean| > 16 sd" (dense semantics only) redoes the passes about the sample mean
    const Clock::time_point t_solve = Clock::now();
    rc = solve_summed(J, false, p, sem, standardize, intercept, o, beta, lambda_out, niter, loss, d);
    double solve_ms = ms_since(t_solve);
    if (!rc && dense && J[0].c->shift_advised) {
        const Clock::time_point t2 = Clock::now();
        rc = sum_on_first(J, true);
        if (!rc) rc = broadcast_from_first(J, true);
        if (!rc) {
            std::vector<std::thread> th;
            for (int g = 1; g < G; ++g) th.emplace_back(run_pass, std::ref(J[g]), !J[g].resident, false, true, (const double *)J[g].ssum);
            run_pass(J[0], !J[0].resident, false, true, J[0].ssum);
            for (auto &t : th) t.join();
            rc = first_error();
        }
        if (!rc) rc = sum_on_first(J, false);
        up_ms += ms_since(t2);
        const Clock::time_point t3 = Clock::now();
        if (!rc) rc = solve_summed(J, true, p, sem, standardize, intercept, o, beta, lambda_out, niter, loss, d);
        solve_ms += ms_since(t3);
    }
    size_t staged = 0, nblocks = 0;
    bool resident = true;
    for (int g = 0; g < G; ++g) { staged += J[g].bytes_staged; nblocks += J[g].blocks.size(); resident = resident && J[g].resident; }
    release_all();
    g_host_stats[0] = ms_since(t_call); g_host_stats[1] = up_ms; g_host_stats[2] = solve_ms; g_host_stats[3] = (double)staged;
    g_host_stats[4] = G; g_host_stats[5] = (double)nblocks; g_host_stats[6] = resident ? 1.0 : 0.0;
    g_host_stats[7] = (double)(g_alloc_count.load() - allocs0);
    g_host_stats[8] = (double)(g_host_staged.load() - staged0);
    return rc;
}

}  // namespace

int host_fit_dense(const double *x, int64_t n, int32_t p, const double *y, int32_t standardize, int32_t intercept,
                   const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    std::vector<HostPiece> all(1);
    all[0].x = x; all[0].ldx = n; all[0].y = y; all[0].rows = n;
    return host_fit(all, n, p, OEMGPU_SEM_DENSE, standardize, intercept, o, beta, lambda_out, niter, loss, d);
}

int host_fit_big(const double *const *x_shards, const int64_t *n_shard, int32_t nshards, int32_t p, const double *const *y_shards,
                 int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                 double *beta, double *lambda_out, int32_t *niter, double *loss, double *d)
{
    // The reference walks row slices serially (ref src/oem_big.h:329-358).  Moments are taken about 0, like the reference's own
    // sums: oemBig never centres (ref src/oem_big.h:757-763, 469-545), so there is no cancellation for a shift to prevent.
    std::vector<HostPiece> all;
    int64_t n = 0;
    for (int s = 0; s < nshards; ++s) {
        if (n_shard[s] == 0) continue;
        HostPiece P; P.x = x_shards[s]; P.ldx = n_shard[s]; P.y = y_shards[s]; P.rows = n_shard[s];
        all.push_back(P);
        n += n_shard[s];
    }
    return host_fit(all, n, p, OEMGPU_SEM_BIG, standardize, intercept, o, beta, lambda_out, niter, loss, d);
}

int host_device_list(const oemgpu_opts *o, std::vector<int> &dev) { return device_list(o, dev); }
int host_hand_over(oemgpu_ctx *to, double *dst, oemgpu_ctx *from, const double *src, size_t doubles) { return hand_over(to, dst, from, src, doubles); }
int host_add_into(oemgpu_ctx *c, double *dst, const double *src, size_t doubles)
{
    hipLaunchKernelGGL(add_into_kernel, dim3(64), dim3(256), 0, c->stream, dst, src, doubles);
    OEM_HIP(hipGetLastError());
    return 0;
}

int host_upload_resident(oemgpu_ctx *c, const double *x, int64_t n, int32_t p, const double *y, const oemgpu_opts *o,
                         double **x_dev, int64_t *ld, double **y_dev, int64_t ldx, bool tight)
{
    DevJob J;
    J.c = c; J.p = p; J.o = o; J.contiguous = true; J.tight = tight;
    J.T = (o && o->upload_threads > 0) ? o->upload_threads : (int)env_size(sw().OEMGPU_UPLOAD_THREADS, 8);
    if (J.T > 64) J.T = 64;
    J.slot_bytes = env_size(sw().OEMGPU_SLOT_BYTES, (size_t)4 << 20) / 4096 * 4096 + 4096;
    HostPiece P; P.x = x; P.ldx = ldx > 0 ? ldx : n; P.y = y; P.rows = n;
    J.pieces.push_back(P);
    int rc = lanes_prepare(c, J.T, J.slot_bytes);
    if (!rc) rc = job_layout(J, (size_t)-1);
    if (rc) return rc;
    run_pass(J, true, false, false, nullptr);
    if (J.abort.load()) { set_error("%s", J.err.c_str()); return J.abort.load(); }
    // the compute stream has waited for every lane's block events inside the pass; nothing else to order
    *x_dev = J.blocks.empty() ? (double *)c->xres : J.blocks[0].xd;
    *ld = J.blocks.empty() ? n : J.blocks[0].ld;
    *y_dev = J.blocks.empty() ? (double *)c->xres : J.blocks[0].yd;
    return 0;
}

}  // namespace oemgpu

// =====================================================================================================
#pragma GCC visibility push(default)
extern "C" {

void oemgpu_row_split(int64_t n, int32_t G, int32_t g, int64_t *r0, int64_t *r1)
{
    if (G < 1) G = 1;
    const int64_t base = n / G;
    if (r0) *r0 = base * g;
    if (r1) *r1 = (g + 1 < G) ? base * (g + 1) : n;
}

int oemgpu_last_host_stats(double *out)
{
    if (!out) { oemgpu::set_error("NULL argument"); return OEMGPU_ERR_ARG; }
    for (int i = 0; i < OEMGPU_NHOSTSTATS; ++i) out[i] = oemgpu::g_host_stats[i];
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop
