// ctx.hpp -- the context behind the C ABI (internal to liboemgpu; api.hip owns the functions, hoststream.hip uses them).
#pragma once

#include "common.hpp"

#include <atomic>
#include <cstddef>
#include <vector>

// one host staging lane of the upload pipeline (hoststream.hip): a copy stream, two pinned bounce slots, their events
struct oemgpu_lane {
    hipStream_t s = nullptr;
    char *slot[2] = {nullptr, nullptr};
    hipEvent_t slot_ev[2] = {nullptr, nullptr};
    bool slot_used[2] = {false, false};
    hipEvent_t blk_ev[2] = {nullptr, nullptr};     // "my copies of block b are enqueued up to here" (b & 1)
};

struct oemgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    size_t hbm_total = 0;        // bytes of device memory (oemgpu_create)
    char *ws = nullptr;          // device workspace (grow-only)
    size_t ws_bytes = 0;
    char *pinned = nullptr;      // pinned host staging for the results
    size_t pinned_bytes = 0;
    char *pinned_in = nullptr;   // pinned host staging for the parameter blob (a pageable source makes the copy block the host)
    size_t pinned_in_bytes = 0;
    bool timing = false;
    hipEvent_t ev[2 * OEMGPU_NTIMERS];
    bool ev_made = false;
    bool ev_used[OEMGPU_NTIMERS];
    double ms[OEMGPU_NTIMERS];
    double diag[2] = {0.0, 0.0};   // path kernel: shader cycles, 100 MHz ticks
    int eig_steps = 0;             // the last eigenvalue step: Lanczos steps taken ...
    int last_engine = 0;           // OEMGPU_ENGINE_* of the most recent penalty x lambda path (oemgpu_last_path_engine)
    int persistent_fallbacks = 0;  // calls of a persistent p >= n engine that timed out (CUs held by somebody else) and were made again with launches
    bool eig_capped = false;       // ... and whether the step cap ended it (oemgpu_last_eigen_info)
    int shifted = 0;               // the last solve read its moments as accumulated about the provisional shift
    int shift_advised = 0;         // the last solve was given moments about 0 whose columns have |mean| >> sd
    char *aux = nullptr;           // xval.oem: fold-ordered copy of X, fold moments, fold coefficients (grow-only)
    size_t aux_bytes = 0;
    std::vector<oemgpu_ctx *> kids;   // xval.oem: one child context (stream, workspace, staging) per concurrent fold fit
    hipEvent_t fork_ev = nullptr;
    // ---- host-resident inputs (hoststream.hip): everything grow-only, so repeated calls allocate nothing
    char *xres = nullptr;          // device copy of the host rows: the whole slice when it fits, else two block buffers
    size_t xres_bytes = 0;
    char *acc = nullptr;           // moment / sample-sum accumulators and the peers' buffers for the in-order sum
    size_t acc_bytes = 0;
    std::vector<oemgpu_lane> lanes;
    size_t slot_bytes = 0;
    hipEvent_t done_ev[2] = {nullptr, nullptr};   // "the moment pass over block buffer k has finished reading it"
    hipEvent_t xfer_ev = nullptr;                 // cross-device hand-over of the moment buffers
    char *xfer_host = nullptr;                    // pinned bounce buffer of a hand-over INTO this context's device when the two devices
    size_t xfer_host_bytes = 0;                   // are not peers (hoststream.hip: hand_over); grow-only
    char *blob_buf = nullptr;      // the parameter blob of run_paths (grow-only, outside the workspace: it survives between calls)
    size_t blob_bytes = 0;
    char *pack_buf = nullptr;      // q > 4096: the packed lower triangle of XX and its products' partial vectors (PathArgs::sympk; grow-only,
    size_t pack_bytes = 0;         // released with the cache's other big buffers when it exceeds OEMGPU_CACHE_KEEP_BYTES)
    char *perm_buf = nullptr;      // 1024 < q <= 4096 with scattered groups: XX, XY and the column constants reordered so that every group is a run
    size_t perm_bytes = 0;         // of neighbours (api.hip: group_run_permutation; grow-only)
    const char *blob_dev = nullptr;   // where the last parameter blob was uploaded (run_paths skips an identical upload)
    size_t blob_len = 0;
    // A persistent engine that timed out (somebody else holds the CUs) is not tried again at once: the next `persistent_skip` calls
    // that would take one -- for at most 30 s -- go straight to the launch-per-iteration engines; the count doubles (4 .. 64) with every
    // further timeout and starts over after a persistent launch that came back (or when the switches are read again).
    int persistent_backoff = 0, persistent_skip = 0;
    // path_coop.hip's one-XCD form: 0 not probed yet, 1 workgroup ids go round 8 XCDs (blockIdx % 8), -1 they do not (or a launch's own
    // proof of placement failed once: never again on this context); what the last path launch did (0 n/a, 1 one XCD, 2 asked for and refused by the proof,
    // 3 asked for but not co-resident on its XCD: made again at device scope)
    int xcd_layout = 0, last_placement = 0;
    double persistent_skip_until = 0.0;        // steady-clock seconds
    unsigned sw_generation = 0;                // Switches::generation this state belongs to
    int *abort_host = nullptr;     // the abort word of the persistent path engines (PathArgs::abort_word): one int of host-coherent pinned memory,
    int *abort_dev = nullptr;      // mapped into the device; allocated by the first call that has an interrupt callback
    bool cached = false;           // owned by the process-wide cache (oemgpu_release_cache frees it)
    bool busy = false;
};

namespace oemgpu {

extern std::atomic<long> g_alloc_count;   // device / pinned allocations, stream / event creations (oemgpu_last_host_stats()[7])

struct Bump {           // carve-out of a context buffer, 256-byte granules
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; }
};

int ctx_reserve(oemgpu_ctx *c, size_t bytes);           // c->ws
int ctx_grow(oemgpu_ctx *c, char **buf, size_t *have, size_t bytes);   // any grow-only device buffer of the context
int set_device(const oemgpu_ctx *c);

// A context from the process-wide cache (created on first use, kept afterwards: no stream / workspace / pinned-memory
// churn in the steady state of repeated host-level calls).  Contexts are checked out, so concurrent callers never share one.
oemgpu_ctx *ctx_acquire(int device);
void ctx_release(oemgpu_ctx *c);

// host-resident entry points (hoststream.hip)
int host_fit_dense(const double *x, int64_t n, int32_t p, const double *y, int32_t standardize, int32_t intercept,
                   const oemgpu_opts *o, double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);
int host_fit_big(const double *const *x_shards, const int64_t *n_shard, int32_t nshards, int32_t p, const double *const *y_shards,
                 int32_t standardize, int32_t intercept, const oemgpu_opts *o,
                 double *beta, double *lambda_out, int32_t *niter, double *loss, double *d);
// the rows of one host matrix resident on the context's device (c->xres: column-major, leading dimension *ld, y behind it at *y_dev)
// (ldx: column stride of the host matrix, 0 = n -- a row slice of a bigger matrix has ldx > n)
int host_upload_resident(oemgpu_ctx *c, const double *x, int64_t n, int32_t p, const double *y, const oemgpu_opts *o,
                         double **x_dev, int64_t *ld, double **y_dev, int64_t ldx = 0, bool tight = false);   // tight: ld = n, no padding
// pieces of the in-library multi-device machinery that xval.oem's host entry point shares with host_fit
int host_device_list(const oemgpu_opts *o, std::vector<int> &dev);                       // opts.ngpus / devices -> ordinals
int host_hand_over(oemgpu_ctx *to, double *dst, oemgpu_ctx *from, const double *src, size_t doubles);   // (peer) copy on from's stream, to's stream waits for it
int host_add_into(oemgpu_ctx *c, double *dst, const double *src, size_t doubles);      // dst += src on c's stream

}  // namespace oemgpu
