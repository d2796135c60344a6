// C ABI of libplume_hip.so (include/plume_hip.h): context, HBM workspace, chunked pipelines, stage timing.
// Host code only; every computation on the data path is a gfx950 kernel from plume_kernels.hip.  There is no CPU
// fallback of any kind in this library.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <cctype>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <cerrno>
#include <sys/random.h>
#include <string>
#include <thread>
#include <algorithm>
#include <memory>
#include <vector>

#include "../../include/plume_hip.h"
#include "plume_agg_launch.h"
#include "plume_host_logic.h"
#include "plume_launch.h"

using namespace plume;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e__ = (expr);                                                                                  \
        if (e__ != hipSuccess) return fail(PLUME_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

constexpr int kMaxSubBatches = 64;


struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(PLUME_ERR_HIP, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// page-locked host scratch (hipHostMalloc): the source of uploads the library itself prepares.  A pageable source would make the runtime stage the copy and move it with a
// blit KERNEL, which then queues behind the saturating multi-scalar kernel and stalls the upload stream (seen in the round-4 pipeline trace: tests/gpu_debug/e2e_trace.py)
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail(PLUME_ERR_HIP, std::string("hipHostMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct StageTimer {
    std::vector<const char*> names;
    std::vector<hipEvent_t> ev;   // ev[0] start, ev[i+1] after stage i
    size_t used = 0;
    uint64_t runs = 0;            // calls that started a timeline here, i.e. put work on this context (Route)
    bool on = false;              // plume_set_stage_timing: a timing event between two kernels of a stream costs ~6 us of idle GPU (rocprofv3 trace of 2^16-item calls, round 5:
                                  // kernels without an event between them follow each other within 0.2 us), five or six per call -- 2 % of a 2^16-item verify, so only on request
    void begin(hipStream_t st) { runs++; names.clear(); used = 0; if (on) mark(st); }
    void mark(hipStream_t st) {
        if (used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev.push_back(e); }
        (void)hipEventRecord(ev[used++], st);
    }
    void stage(const char* name, hipStream_t st) { if (!on) return; names.push_back(name); mark(st); }
    void destroy() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); }
};

// staging for one piece of a host-pointer call
struct HostSlot {
    DevBuf msgs, off, in[6], out[7];
    PinnedBuf relbuf;                                                 // piece-relative message offsets (source of an upload: lives until the slot is reused), page-locked
    uint64_t* rel = nullptr;
    hipEvent_t ready = nullptr, computed = nullptr, drained = nullptr;   // uploads landed / kernels finished / downloads landed
    bool in_flight = false;
};

// one worker thread per shard of a multi-device context: it binds the shard's device once and then runs the jobs handed to it
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false, ready = false;
    int rc = 0;
    int numa_node = -1;      // the node the thread was bound to (bind_thread_to_device_node), -1 = left alone
    std::string err;
};

// The generator's fixed tables (the verifier's 2^23-row window table, 1 GiB; the signer's comb, 252 MiB; 18 ms to build both) depend on nothing but G: every context of a
// process on the same device shares ONE read-only copy, freed with the last context (a second batch in flight -- a second context -- then costs its workspace only).
// Round 4: each table is built by the first call that needs it (need_gtab: verify, verify_non_zk; need_gcomb: sign, SEC1-DER export, the aggregate check), so a
// verify-only process never holds the comb, a sign-only one never the 1 GiB window table, and plume_init itself allocates neither.
struct FixedTables {
    std::mutex m;                 // the tables of ONE device: their build (18 ms, synchronous) holds this lock, not the process-wide one -- the shards of a multi-device context
                                  // build theirs side by side (the CPU pipeline harness showed eight shards' first call queueing behind one lock: tests/hostsim)
    DevBuf gtab, gcomb, gscan;
    int refs = 0;                 // under g_fixed_mutex
    bool gtab_built = false, gcomb_built = false, gscan_built = false;
};
static std::mutex g_fixed_mutex;                 // the map and the reference counts.  Lock order: g_fixed_mutex, then a device's FixedTables::m
static std::map<int, FixedTables> g_fixed;

struct plume_ctx {
    int device = 0;
    hipStream_t stream = nullptr, up = nullptr, down = nullptr;   // kernels / host->HBM / HBM->host
    hipStream_t side = nullptr;                                   // aggregate check: the generator term and the upper windows' reduction run beside the main stream
    hipStream_t pre = nullptr;                                    // overlapped verify / sign: the stages BEFORE the multi-scalar kernel of sub-batch k+1 run here, beside that kernel of sub-batch k
    hipEvent_t pre_begin = nullptr;                               // ... the caller's stream has reached the call (inputs are there, the workspace is free)
    std::vector<hipEvent_t> pre_ready;                            // ... sub-batch k's window tables are built
    const char* last_msm_kernel = nullptr;                         // the multi-scalar kernel the last verify call on this context launched (plume_last_msm_kernel)
    bool last_msm_sampled = false;                                 // ... and whether that launch sampled its clocks (plume_last_msm_clock: stage timing was on for the call)
    std::vector<size_t> redo_counters;                            // word offsets (in `redo`) of the last verify call's redo counters, one per sub-batch (plume_last_redo_tasks)
    int sub_batches = 1;                                          // device-resident verify / sign: number of sub-batches; 1 = strictly serial launch order (the default: measured on the MI355X, r03, kernels of two
                                                                  // streams sharing the CUs cost MORE than the table kernel's idle issue slots give back -- 21.85 ms serial vs 22.1-22.4 ms for 2..16 sub-batches, LABNOTES.md §6)
    size_t overlap_min = (size_t)1 << 17;                         // ... batches below this many items always run serial (a sub-batch must still fill the chip)
    hipEvent_t agg_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // terms ready / upper bucket sums ready / generator term ready / upper windows reduced
    hipEvent_t ws_free = nullptr;     // recorded behind the last kernel of every device-resident call: the next call's stream waits on it before it
    bool ws_used = false;             // touches the per-context workspace, so calls on DIFFERENT streams of one context cannot race on that scratch
    hipStream_t ws_stream = nullptr;  // the stream ws_free was last recorded on
    size_t chunk = (size_t)1 << 20;
    size_t host_piece = (size_t)1 << 19;                            // host-pointer calls: largest pipelined piece
    size_t host_first_piece = (size_t)1 << 16;                      // ... the first piece (its upload is the only one no kernel hides); pieces then grow 3x per step
    size_t host_tail_piece = (size_t)1 << 16;                       // ... the last piece of calls with large outputs (its download is the only one no kernel hides)
    size_t host_register_min = 0;                                   // host-pointer calls: page-lock caller arrays of at least this many bytes for the call (0 = never)
    // multi-device parent (plume_init_multi): the shards are complete single-device contexts, one worker thread each; a parent owns no GPU state
    std::vector<plume_ctx*> shards;
    std::vector<Worker*> workers;
    HostSlot slot[4];                                              // host-pointer calls: staging slots (two for the one-lane pipeline, four when two lanes take the pieces in turn)
    plume_ctx* host_lane = nullptr;                                // ... the second lane of the host-pointer pipeline: a complete single-device context (workspace, streams), created on first use
    size_t msm_pair_max = (size_t)1 << 14;                          // verify calls (slices) of at most this many items run k_verify_msm_pair (env PLUME_MSM_PAIR_MAX; 0: never)
    bool split_scalars = true;                                      // ... and the scalar stage in that kernel's idle role (env PLUME_SPLIT_SCALARS=0: a launch of its own, A/B)
    size_t ingest_split_max = (size_t)1 << 16;                     // verify calls (slices) of at most this many items run the ingest stage with two lanes per item (latency-bound there)
    int sign_uniform = 1;                                          // plume_set_sign_uniform: the signer's schedule (level 0, 1, 2).  Default 1 since round 5: no branch on a digit of sk or r
                                                                   // (k256's multiplication is constant-time, rust-k256/src/randomizedsigner.rs:51-70; measured price +2.5 % per signature)
    int host_lanes = 2;                                            // ... 1 = every piece on the context itself (rounds 1-3), 2 = pieces alternate between the context and host_lane
    int host_sign_lanes = 2;                                       // ... the signer's host-pointer call: lanes it may use (env PLUME_HOST_SIGN_LANES).  2 since round 6, with uniform 2^16-item pieces (plume_host_logic.h)
    bool host_lane_failed = false;                                 // ... the second lane could not be created once (out of memory): do not retry on every call
    int jobs_per_lane = kTableJobsPerLane;
    bool jobs_per_lane_forced = false;
    FixedTables* fixed = nullptr;                                 // this device's shared generator tables (g_fixed)
    // batches in flight (plume_set_in_flight): device-resident calls are dealt out in turn to this context and to `lanes`, complete single-device contexts of their own
    // (workspace, streams, events; the fixed tables are shared), so that calls the caller issues on DIFFERENT streams run side by side instead of queueing for one workspace
    std::vector<plume_ctx*> lanes;
    size_t lane_next = 0;
    size_t in_flight_min = 0;                                      // verify / sign calls of fewer items are not dealt out to the lanes (env PLUME_IN_FLIGHT_MIN; 0, the default: every call is)
    plume_ctx* lane_last = nullptr;                               // the lane the last device-resident call went to (plume_last_stage_times, plume_last_redo_tasks)
    DevBuf bases, jobflags, itemflags, tab, tabscr, res, resinf, res2, res2inf, pkaff, sink, redo, digs, eq1fall, eq1k, clk;
    int eq1_short = 1;                                             // verify calls that give R: equation 1 in its short form (plume_eis.h).  0 = long form always (A/B), 2 = test: every item takes the fallback
    size_t eq1_short_min = (size_t)1 << 16;                        // ... for calls of at least this many items.  Round 6 sweep on one box, interleaved (profiles/r06_eq1_threshold.txt), now that the
                                                                   // scalar stage -- half-GCD included -- of calls up to 2^16 runs in role B of the two-role ingest kernel: 2^16 items 1.385 -> 1.343 ms
                                                                   // (-3.0 %), 2^17 -4.4 %, 2^18 -4.1 %; 2^15 +6.3 % (0.928 -> 0.987: one wavefront per SIMD, the chain's latency is the
                                                                   // kernel's time and equation 2's chain is the long one either way), <= 2^14 +35 % (the half chains of k_verify_msm_pair
                                                                   // need the long form).  Round 5's threshold, before the scalar stage moved: 2^17
    DevBuf dec[4], preflags;
    DevBuf agg[15];      // aggregate check (plume_aggregate.h): haff, scal, flags, gs, hash_ok, counters, count, sort tiles, sorted, bsum, bsuminf, red, redinf, ssum, perm + its histogram
    DevBuf agg_record;   // the running record of a host-pointer aggregate call (pieces of one batch)
    DevBuf dslots, dminid, dmyslot, dcount, dblockcnt;   // nullifier-set post-processing (plume_dedup.h)   // SEC1 ingest: decompressed 64-byte records + per-item reject flags
    StageTimer timer;
};

// Every per-context tunable a derived context (an in-flight lane, the host pipeline's second lane, a shard) has to share with the context it serves: ONE place, so that a new
// knob cannot reach some derived contexts and miss others (round 4's host lane did not inherit sign_uniform: VERDICT r4).
static void inherit_tunables(plume_ctx* to, const plume_ctx* from) {
    to->chunk = from->chunk; to->sub_batches = from->sub_batches; to->overlap_min = from->overlap_min; to->sign_uniform = from->sign_uniform;
    to->ingest_split_max = from->ingest_split_max; to->split_scalars = from->split_scalars; to->msm_pair_max = from->msm_pair_max;
    to->jobs_per_lane = from->jobs_per_lane; to->jobs_per_lane_forced = from->jobs_per_lane_forced;
    to->host_piece = from->host_piece; to->host_first_piece = from->host_first_piece; to->host_tail_piece = from->host_tail_piece; to->host_register_min = from->host_register_min;
    to->host_lanes = from->host_lanes; to->host_sign_lanes = from->host_sign_lanes; to->eq1_short = from->eq1_short; to->eq1_short_min = from->eq1_short_min;
    to->timer.on = from->timer.on;
}
// after a setter changed `ctx`: hand the change to every context derived from it (shards and their derived contexts, in-flight lanes, the host pipeline's second lane)
static void propagate_tunables(plume_ctx* ctx) {
    for (plume_ctx* sh : ctx->shards) { inherit_tunables(sh, ctx); propagate_tunables(sh); }
    for (plume_ctx* l : ctx->lanes) inherit_tunables(l, ctx);
    if (ctx->host_lane) { inherit_tunables(ctx->host_lane, ctx); ctx->host_lane->sub_batches = 1; }   // the pieces of a host-pointer call are never cut again
}

static int bind(plume_ctx* ctx) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (!ctx->shards.empty()) return fail(PLUME_ERR_ARG, "device-resident entry points need a single-device context (plume_init): device pointers belong to one GPU");
    HIPCHK(hipSetDevice(ctx->device));
    return 0;
}
// The lane a device-resident call runs on -- the context itself, or one of its in-flight lanes in turn -- and the stream it runs on.  The stream is resolved against the
// context the CALLER holds before the call is dealt out: NULL means "that context's own stream" whichever lane serves the call, so successive NULL-stream calls stay
// ordered against each other (a sign followed by a verify of its outputs) exactly as they are without lanes.  (Round 3 resolved NULL after routing, i.e. to the lane's
// private stream, which changes from call to call: ADVICE r3.)  A lane's workspace is guarded by its ws_free event, so running it on a stream it does not own is safe.
struct Route {
    plume_ctx* held;           // the context the caller holds
    plume_ctx* lane;           // the lane that serves this call
    hipStream_t st;
    size_t next0 = 0;
    plume_ctx* last0 = nullptr;
    uint64_t epoch0 = 0;
    Route(plume_ctx* ctx, void* stream, size_t n = (size_t)-1) : held(ctx), lane(ctx) {
        st = stream ? (hipStream_t)stream : (ctx ? ctx->stream : nullptr);
        if (!ctx) return;
        next0 = ctx->lane_next; last0 = ctx->lane_last;
        if (!ctx->lanes.empty() && n >= ctx->in_flight_min) {                 // (a small call stays on the first lane: plume_set_in_flight)
            const size_t k = ctx->lanes.size() + 1, i = ctx->lane_next++ % k;
            lane = i == 0 ? ctx : ctx->lanes[i - 1];
        }
        ctx->lane_last = lane;
        epoch0 = lane->timer.runs;
    }
    Route(const Route&) = delete;
    Route& operator=(const Route&) = delete;
    // A call that is refused before it starts its lane's stage timeline (an argument check, n above the chunk size, an empty batch) never happened as far as the lanes are concerned:
    // it neither takes a turn nor becomes "the last device-resident call" that plume_last_stage_times / plume_last_redo_tasks report.  (Found by the CPU pipeline harness,
    // tests/hostsim: a refused call moved lane_last to a lane that had run nothing, and the stage times read afterwards -- and the wait they imply -- were another call's.)
    ~Route() { if (held && lane->timer.runs == epoch0) { held->lane_next = next0; held->lane_last = last0; } }
};
// Workspace ordering for the device-resident entry points: every call leaves ws_free behind its last kernel, and the next call's stream
// waits on it first.  Calls on one stream are ordered anyway; this makes calls on DIFFERENT streams of one context safe too.
static int ws_acquire(plume_ctx* ctx, hipStream_t st) {
    // (the previous call on the SAME stream is ordered by the stream itself: no wait packet between two calls of a stream of small calls -- each one is a few us of idle GPU)
    // The comparison is of stream HANDLES: a caller who destroys a stream and gets the same handle value back for a new one while this context's last call on the old
    // one is still running would skip a wait it needs.  The header therefore asks that a caller-provided stream stay alive until the context's last call on it has
    // finished (include/plume_hip.h, plume_init); the library's own streams live as long as the context.
    if (ctx->ws_used && ctx->ws_stream != st) HIPCHK(hipStreamWaitEvent(st, ctx->ws_free, 0));
    return 0;
}
static int ws_release(plume_ctx* ctx, hipStream_t st) {
    HIPCHK(hipEventRecord(ctx->ws_free, st));
    ctx->ws_used = true;
    ctx->ws_stream = st;
    return 0;
}
// Holds the workspace from a successful ws_acquire to the end of the call.  A call that fails half way (an allocation, a launch) must still leave ws_free behind
// whatever it has already enqueued -- on the caller's stream and on the context's side stream -- or the next call, on another stream, would run into it.
struct WsHold {
    plume_ctx* ctx;
    hipStream_t st;
    bool released = false;
    WsHold(plume_ctx* c, hipStream_t s) : ctx(c), st(s) {}
    WsHold(const WsHold&) = delete;
    WsHold& operator=(const WsHold&) = delete;
    int release() { released = true; return ws_release(ctx, st); }
    ~WsHold() {
        if (released) return;
        if (ctx->side) (void)hipStreamSynchronize(ctx->side);      // failure path only: side-stream work the caller's stream never joined
        if (ctx->pre) (void)hipStreamSynchronize(ctx->pre);
        (void)hipEventRecord(ctx->ws_free, st);
        ctx->ws_used = true;
        ctx->ws_stream = st;
    }
};

extern "C" const char* plume_last_error(void) { return g_err.c_str(); }
#ifndef PLUME_BUILD_ID
#define PLUME_BUILD_ID "unknown"
#endif
extern "C" const char* plume_version(void) { return "plume_hip 0.6 gfx950 build=" PLUME_BUILD_ID; }

static void destroy_single(plume_ctx* ctx) {
    for (plume_ctx* l : ctx->lanes) destroy_single(l);
    ctx->lanes.clear();
    if (ctx->host_lane) { destroy_single(ctx->host_lane); ctx->host_lane = nullptr; }
    (void)hipSetDevice(ctx->device);
    // The context's workspace, events and the shared tables may still be in use by kernels queued on CALLER streams (device-resident calls run on whatever stream they were
    // given).  Every such call leaves ws_free behind its last kernel and waits for its predecessor's first (ws_acquire / ws_release), so the LAST ws_free covers them all: wait
    // for that event and for the context's own streams -- not for the whole device, which would also wait for every other framework's work in the process.
    if (ctx->ws_used && ctx->ws_free) (void)hipEventSynchronize(ctx->ws_free);
    for (hipStream_t q : {ctx->stream, ctx->up, ctx->down, ctx->side, ctx->pre}) if (q) (void)hipStreamSynchronize(q);
    for (DevBuf* b : {&ctx->bases, &ctx->jobflags, &ctx->itemflags, &ctx->tab, &ctx->tabscr, &ctx->res, &ctx->resinf, &ctx->res2, &ctx->res2inf, &ctx->pkaff,
                      &ctx->sink, &ctx->redo, &ctx->digs, &ctx->eq1fall, &ctx->eq1k, &ctx->clk, &ctx->dec[0], &ctx->dec[1], &ctx->dec[2], &ctx->dec[3], &ctx->preflags, &ctx->agg_record, &ctx->dslots, &ctx->dminid, &ctx->dmyslot, &ctx->dcount, &ctx->dblockcnt})
        b->release();
    for (DevBuf& b : ctx->agg) b.release();
    if (ctx->fixed) {
        std::lock_guard<std::mutex> lk(g_fixed_mutex);
        std::lock_guard<std::mutex> lk2(ctx->fixed->m);
        if (--ctx->fixed->refs == 0) {           // every context that could read the tables has waited for its own last call on its way here: nothing is still running
            ctx->fixed->gtab.release(); ctx->fixed->gcomb.release(); ctx->fixed->gscan.release(); ctx->fixed->gtab_built = ctx->fixed->gcomb_built = ctx->fixed->gscan_built = false;
        }
        ctx->fixed = nullptr;
    }
    for (HostSlot& sl : ctx->slot) {
        sl.msgs.release(); sl.off.release(); sl.relbuf.release(); sl.rel = nullptr;
        for (DevBuf& b : sl.in) b.release();
        for (DevBuf& b : sl.out) b.release();
        for (hipEvent_t e : {sl.ready, sl.computed, sl.drained}) if (e) (void)hipEventDestroy(e);
    }
    ctx->timer.destroy();
    if (ctx->ws_free) (void)hipEventDestroy(ctx->ws_free);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->up) (void)hipStreamDestroy(ctx->up);
    if (ctx->down) (void)hipStreamDestroy(ctx->down);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->pre) (void)hipStreamDestroy(ctx->pre);
    if (ctx->pre_begin) (void)hipEventDestroy(ctx->pre_begin);
    for (hipEvent_t e : ctx->pre_ready) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->agg_ev) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

// everything plume_init does after the context object exists; any failure leaves a context that destroy_single() can release
static int init_single(plume_ctx* ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    if (const char* e = std::getenv("PLUME_JOBS_PER_LANE")) { int v = std::atoi(e); if (v >= 1 && v <= 64) { ctx->jobs_per_lane = v; ctx->jobs_per_lane_forced = true; } }   // tuning knob
    if (const char* e = std::getenv("PLUME_HOST_PIECE")) { long v = std::atol(e); if (v >= 1) ctx->host_piece = (size_t)v; }   // tuning knob
    if (const char* e = std::getenv("PLUME_HOST_FIRST_PIECE")) { long v = std::atol(e); if (v >= 1) ctx->host_first_piece = (size_t)v; }   // tuning knob
    if (const char* e = std::getenv("PLUME_HOST_TAIL_PIECE")) { long v = std::atol(e); if (v >= 1) ctx->host_tail_piece = (size_t)v; }   // tuning knob
    if (const char* e = std::getenv("PLUME_HOST_REGISTER_MIN")) { long v = std::atol(e); if (v >= 0) ctx->host_register_min = (size_t)v; }   // tuning knob
    if (const char* e = std::getenv("PLUME_MSM_PAIR_MAX")) { long v = std::atol(e); if (v >= 0) ctx->msm_pair_max = (size_t)v; }   // tuning / A-B knob (0: never)
    if (const char* e = std::getenv("PLUME_SPLIT_SCALARS")) ctx->split_scalars = std::atoi(e) != 0;   // A/B knob
    if (const char* e = std::getenv("PLUME_INGEST_SPLIT_MAX")) { long v = std::atol(e); if (v >= 0) ctx->ingest_split_max = (size_t)v; }   // tuning knob (0: never)
    if (const char* e = std::getenv("PLUME_SIGN_UNIFORM")) ctx->sign_uniform = std::min(2, std::max(0, std::atoi(e)));   // default of new contexts (plume_set_sign_uniform); 0 opts out of the uniform schedule
    if (const char* e = std::getenv("PLUME_EQ1_SHORT")) { int v = std::atoi(e); if (v >= 0 && v <= 3) ctx->eq1_short = v; }   // A/B and test knob (plume_eis.h)
    if (const char* e = std::getenv("PLUME_EQ1_SHORT_MIN")) { long v = std::atol(e); if (v >= 0) ctx->eq1_short_min = (size_t)v; }   // tuning knob
    if (const char* e = std::getenv("PLUME_IN_FLIGHT_MIN")) { long v = std::atol(e); if (v >= 0) ctx->in_flight_min = (size_t)v; }   // tuning knob (plume_set_in_flight)
    if (const char* e = std::getenv("PLUME_HOST_SIGN_LANES")) { int v = std::atoi(e); if (v == 1 || v == 2) ctx->host_sign_lanes = v; }
    if (const char* e = std::getenv("PLUME_STAGE_TIMES")) ctx->timer.on = std::atoi(e) != 0;   // default of new contexts (plume_set_stage_timing)
    if (const char* e = std::getenv("PLUME_HOST_LANES")) { int v = std::atoi(e); if (v == 1 || v == 2) ctx->host_lanes = v; }   // 1 = the one-lane host-pointer pipeline of rounds 1-3 (A/B)
    // The runtime multiplexes a process's streams onto a few hardware queues PER PRIORITY LEVEL (GPU_MAX_HW_QUEUES, 4 by default), and two streams that share a queue wait for
    // each other's packets -- event records and waits included.  Seen in the round-5 pipeline timelines (PLUME_HOST_TRACE): with two lanes, a finished piece's download waited
    // for the OTHER lane's kernels (6 ms) because `down` shared a queue with that lane's stream.  So the copy streams live at high priority -- their own pool of queues, away
    // from every compute stream of the process (the priority itself is irrelevant to them: copies run on the copy engines) -- and the side / pre streams that most contexts
    // never use are created on first use (side_stream / pre_stream) instead of taking queue slots from the streams that matter.
    int prio_least = 0, prio_greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithPriority(&ctx->up, hipStreamNonBlocking, prio_greatest));
    HIPCHK(hipStreamCreateWithPriority(&ctx->down, hipStreamNonBlocking, prio_greatest));
    HIPCHK(hipEventCreateWithFlags(&ctx->pre_begin, hipEventDisableTiming));
    if (const char* e = std::getenv("PLUME_SUB_BATCHES")) { int v = std::atoi(e); if (v >= 1 && v <= kMaxSubBatches) ctx->sub_batches = v; }   // tuning knob
    if (const char* e = std::getenv("PLUME_SERIAL")) { if (std::atoi(e) != 0) ctx->sub_batches = 1; }                                  // ... and PLUME_SERIAL=1 wins over it: strictly serial launch order (per-kernel measurements)
    if (const char* e = std::getenv("PLUME_OVERLAP_MIN")) { long v = std::atol(e); if (v >= 1) ctx->overlap_min = (size_t)v; }         // tuning knob
    for (hipEvent_t& e : ctx->agg_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ws_free, hipEventDisableTiming));
    for (HostSlot& sl : ctx->slot) {
        HIPCHK(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.drained, hipEventDisableTiming));
    }
    if (ctx->sink.ensure((size_t)PLUME_FIXED_BASES * 2 * PLUME_FE_WORDS * 4 + 512)) return PLUME_ERR_HIP;
    {
        std::lock_guard<std::mutex> lk(g_fixed_mutex);
        FixedTables& ft = g_fixed[ctx->device];
        ft.refs++;
        ctx->fixed = &ft;
    }
    return 0;
}
// The generator's tables, built once per device and process on the first call that needs them, one entry per lane (k_fixed_bases + k_fixed_table), under the lock: contexts
// that meet here side by side (plume_init_multi's shards, a second batch in flight) wait for the first one's build.  The build is synchronous (18 ms for both, once).
static int need_fixed(plume_ctx* ctx, bool gtab, bool gcomb, bool gscan = false) {
    FixedTables& ft = *ctx->fixed;
    std::lock_guard<std::mutex> lk(ft.m);
    const bool bt = gtab && !ft.gtab_built, bc = gcomb && !ft.gcomb_built, bs = gscan && !ft.gscan_built;
    if (!bt && !bc && !bs) return 0;
    if ((bt && ft.gtab.ensure((size_t)PLUME_GTAB_WORDS * 4)) || (bc && ft.gcomb.ensure((size_t)PLUME_COMB_WORDS * 4)) || (bs && ft.gscan.ensure((size_t)PLUME_GSCAN_WORDS * 4))) {
        if (bt) ft.gtab.release();
        if (bc) ft.gcomb.release();
        if (bs) ft.gscan.release();
        return PLUME_ERR_HIP;
    }
    launch_fixed_tables(bt ? ft.gtab.as<uint32_t>() : nullptr, bc ? ft.gcomb.as<uint32_t>() : nullptr, bs ? ft.gscan.as<uint32_t>() : nullptr, ctx->sink.as<uint32_t>(), ctx->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        if (bt) ft.gtab.release();
        if (bc) ft.gcomb.release();
        if (bs) ft.gscan.release();
        return fail(PLUME_ERR_HIP, std::string("fixed tables: ") + hipGetErrorString(e));
    }
    if (bt) ft.gtab_built = true;
    if (bc) ft.gcomb_built = true;
    if (bs) ft.gscan_built = true;
    return 0;
}
static int need_gtab(plume_ctx* ctx) { return need_fixed(ctx, true, false); }
static int need_gcomb(plume_ctx* ctx) { return need_fixed(ctx, false, true); }
// the signer's (and the SEC1-DER export's) table of G: the comb, or at level 2 of the uniform schedule the small scanned table
static int need_sign_tables(plume_ctx* ctx) { return ctx->sign_uniform == 2 ? need_fixed(ctx, false, false, true) : need_gcomb(ctx); }

static int check_device(int device_id) {
    if (device_id < 0) return fail(PLUME_ERR_ARG, "negative device id");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(PLUME_ERR_NODEV, "no HIP device visible (this library has no CPU fallback)");
    if (device_id >= ndev) return fail(PLUME_ERR_ARG, "device id out of range");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PLUME_ERR_NODEV, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    return 0;
}

extern "C" int plume_init(plume_ctx** out, int device_id) {
    if (!out) return fail(PLUME_ERR_ARG, "plume_init: bad argument");
    *out = nullptr;
    if (int rc = check_device(device_id)) return rc;
    plume_ctx* ctx = new plume_ctx();
    ctx->device = device_id;
    if (int rc = init_single(ctx)) { const std::string keep = g_err; destroy_single(ctx); g_err = keep; return rc; }
    *out = ctx;
    return 0;
}

// ------------------------------------------------------------------------------------------- multi-device contexts
// A shard's worker thread issues every copy between the caller's arrays and its GPU; on a two-socket node it should run on the socket the GPU hangs off (SURVEY.md §8e
// names host-side placement as the scaling risk).  The device's PCI address gives the NUMA node through sysfs; the thread's affinity becomes that node's CPUs, cut to the
// CPUs the process may use.  Anything missing on the way (no sysfs, node -1, a container cpuset that excludes the node) leaves the thread where it was.
// PLUME_NO_AFFINITY=1 switches it off.  Returns the node, or -1.
static int bind_thread_to_device_node(int device) {
    if (const char* e = std::getenv("PLUME_NO_AFFINITY")) { if (std::atoi(e) != 0) return -1; }
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* c = bus; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    int node = -1;
    {
        const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) return -1;
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node < 0) return -1;
    char list[4096] = {0};
    {
        const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) return -1;
        const bool got = std::fgets(list, (int)sizeof list, f) != nullptr;
        std::fclose(f);
        if (!got) return -1;
    }
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return -1;
    int any = 0;
    for (const char* q = list; *q && *q != '\n';) {                        // "0-15,128-143"
        char* end = nullptr;
        long lo = std::strtol(q, &end, 10), hi = lo;
        if (end == q) break;
        if (*end == '-') { q = end + 1; hi = std::strtol(q, &end, 10); }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &want); any++; }
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',' ) break;
    }
    if (!any) return -1;
    return sched_setaffinity(0, sizeof want, &want) == 0 ? node : -1;
}

static void worker_main(Worker* w, int device) {
    (void)hipSetDevice(device);
    const int node = bind_thread_to_device_node(device);
    std::unique_lock<std::mutex> lk(w->m);
    w->numa_node = node; w->ready = true;
    w->cv.notify_all();
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        g_err.clear();
        const int rc = job();
        lk.lock();
        w->rc = rc; w->err = g_err; w->done = true;
        w->cv.notify_all();
    }
}
static void worker_post(Worker* w, std::function<int()> job) {
    std::lock_guard<std::mutex> lk(w->m);
    w->job = std::move(job); w->has_job = true; w->done = false;
    w->cv.notify_all();
}
static int worker_wait(Worker* w, std::string& err) {
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->done; });
    err = w->err;
    return w->rc;
}
static void destroy_multi(plume_ctx* ctx) {
    for (Worker* w : ctx->workers) {
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    for (plume_ctx* sh : ctx->shards) destroy_single(sh);
    delete ctx;
}

// Run f(shard context, lo, hi) for the contiguous even split [floor(d*n/g), floor((d+1)*n/g)) of [0, n) on every shard's worker thread
//  and wait for all of them; the first failure (in shard order) is the call's result.
template <class F>
static int for_shards(plume_ctx* ctx, size_t n, F f) {
    const size_t g = ctx->shards.size();
    for (size_t d = 0; d < g; d++) {
        size_t lo, hi;
        plume_host::shard_bounds(n, d, g, lo, hi);
        plume_ctx* sh = ctx->shards[d];
        worker_post(ctx->workers[d], [=]() -> int { return lo < hi ? f(sh, lo, hi) : 0; });
    }
    int rc = 0;
    std::string err, first;
    for (size_t d = 0; d < g; d++) {
        const int r = worker_wait(ctx->workers[d], err);
        if (r != 0 && rc == 0) { rc = r; first = "shard " + std::to_string(d) + " (device " + std::to_string(ctx->shards[d]->device) + "): " + err; }
    }
    return rc ? fail(rc, first) : 0;
}

extern "C" int plume_init_multi(plume_ctx** out, const int* device_ids, int n_devices) {
    if (!out || !device_ids || n_devices <= 0 || n_devices > 64) return fail(PLUME_ERR_ARG, "plume_init_multi: bad argument");
    *out = nullptr;
    for (int d = 0; d < n_devices; d++) if (int rc = check_device(device_ids[d])) return rc;
    plume_ctx* ctx = new plume_ctx();
    ctx->device = device_ids[0];
    // the shards are independent contexts on (usually) different devices: create them side by side (streams, events, the generator's tables), one thread each
    std::vector<int> rcs((size_t)n_devices, 0);
    std::vector<std::string> errs((size_t)n_devices);
    std::vector<std::thread> builders;
    for (int d = 0; d < n_devices; d++) {
        plume_ctx* sh = new plume_ctx();
        sh->device = device_ids[d];
        ctx->shards.push_back(sh);
    }
    for (int d = 0; d < n_devices; d++)
        builders.emplace_back([&, d]() { g_err.clear(); rcs[(size_t)d] = init_single(ctx->shards[(size_t)d]); errs[(size_t)d] = g_err; });
    for (std::thread& b : builders) b.join();
    for (int d = 0; d < n_devices; d++)
        if (rcs[(size_t)d]) { const std::string keep = "shard " + std::to_string(d) + " (device " + std::to_string(device_ids[d]) + "): " + errs[(size_t)d]; const int rc = rcs[(size_t)d]; destroy_multi(ctx); g_err = keep; return rc; }
    inherit_tunables(ctx, ctx->shards[0]);   // the parent owns no GPU state but is what the setters are called on: it starts from what its shards read from the environment
    for (int d = 0; d < n_devices; d++) {
        Worker* w = new Worker();
        ctx->workers.push_back(w);
        w->th = std::thread(worker_main, w, device_ids[d]);
    }
    for (Worker* w : ctx->workers) { std::unique_lock<std::mutex> lk(w->m); w->cv.wait(lk, [&] { return w->ready; }); }   // bound to their devices (and NUMA nodes) before the first call
    *out = ctx;
    return 0;
}

// the NUMA node shard d's worker thread was bound to, -1 = not bound (single-device context, unknown node, PLUME_NO_AFFINITY)
extern "C" int plume_shard_numa_node(const plume_ctx* ctx, int shard) {
    if (!ctx || shard < 0 || (size_t)shard >= ctx->workers.size()) return -1;
    return ctx->workers[(size_t)shard]->numa_node;
}
extern "C" int plume_num_shards(const plume_ctx* ctx) { return ctx ? (ctx->shards.empty() ? 1 : (int)ctx->shards.size()) : 0; }

extern "C" void plume_destroy(plume_ctx* ctx) {
    if (!ctx) return;
    if (!ctx->shards.empty()) destroy_multi(ctx); else destroy_single(ctx);
}

// ------------------------------------------------------------------------------------------------ pinned host memory
extern "C" void* plume_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 1;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { fail(PLUME_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
extern "C" void plume_host_free(void* p) { if (p) (void)hipHostFree(p); }
extern "C" int plume_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return fail(PLUME_ERR_ARG, "plume_host_register: bad argument");
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return 0;
}
extern "C" int plume_host_unregister(void* p) {
    if (!p) return fail(PLUME_ERR_ARG, "plume_host_unregister: bad argument");
    HIPCHK(hipHostUnregister(p));
    return 0;
}

extern "C" int plume_set_chunk(plume_ctx* ctx, size_t max_items) {
    if (!ctx || max_items == 0 || max_items > ((size_t)1 << 26)) return fail(PLUME_ERR_ARG, "plume_set_chunk: bad argument");
    ctx->chunk = max_items;
    propagate_tunables(ctx);
    return 0;
}

// The signer's uniform schedule (opt-in): the two kernels that walk the digits of sk and r (comb multiplication by G, windowed multiplication by H) then execute the same
// instruction sequence whatever the digits are -- no zero-digit skip, no sign branch, no "accumulator is still the identity" case: every slot adds (a zero digit adds row 1 to
// a copy that a masked select drops), the sign is a masked select, and the accumulator starts at an offset point that comes off at the end.  Outputs are bit-identical.
// Level 1 leaves one thing secret-dependent: the ADDRESS of the table row a slot gathers.  Level 2 removes that too, the way k256 does (it scans its 16-entry table with
// conditional moves): every slot reads all 3 rows of its table and keeps one by masked selects (msm_add_uniform), and because the 18-bit comb of G cannot be scanned
// (131072 rows per window) the multiplications by G use a 52-window x 16-row table of their own (PLUME_GSCAN_*, 52 additions instead of 15).  Cost on the MI355X: DESIGN.md §9.
extern "C" int plume_set_sign_uniform(plume_ctx* ctx, int level) {
    if (!ctx) return fail(PLUME_ERR_ARG, "plume_set_sign_uniform: null context");
    if (level < 0 || level > 2) return fail(PLUME_ERR_ARG, "plume_set_sign_uniform: the level is 0, 1 or 2");
    ctx->sign_uniform = level;
    propagate_tunables(ctx);
    return 0;
}

// The verifier's first equation where the call gives R (V1 verify, verify_non_zk): 1 = short form for calls of at least eq1_short_min items (csrc/plume_eis.h; the
// default), 0 = long form always, 3 = short form whatever the size, 2 = test mode: the scalar stage files every item as "long form" and the checked chain of the redo
// launch runs them all.
extern "C" int plume_set_eq1_short(plume_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > 3) return fail(PLUME_ERR_ARG, "plume_set_eq1_short: 0, 1, 2 or 3");
    ctx->eq1_short = mode;
    propagate_tunables(ctx);
    return 0;
}
extern "C" int plume_get_sign_uniform(const plume_ctx* ctx) {
    if (!ctx) return fail(PLUME_ERR_ARG, "plume_get_sign_uniform: null context");
    return ctx->sign_uniform;
}
// the mode plume_set_eq1_short left (or the environment's default) and, through min_items when given, the smallest CALL that takes the short form in mode 1.  The rule is
// applied to the call's n, not to the slices plume_set_sub_batches may cut it into.
extern "C" int plume_get_eq1_short(const plume_ctx* ctx, size_t* min_items) {
    if (!ctx) return fail(PLUME_ERR_ARG, "plume_get_eq1_short: null context");
    if (min_items) *min_items = ctx->eq1_short_min;
    return ctx->eq1_short;
}
// Which multi-scalar kernel the last verify call served by this context launched: "k_verify_msm" (both equations in the long form), "k_verify_msm_s" (equation 1 in the
// short form) or "k_verify_msm_pair" (half chains, small calls); NULL before the first verify.  With batches in flight: the lane of the last device-resident call; for a
// multi-device context: the first shard.  So that a measurement NAMES the kernel it looks up counters for from what ran, not from the environment's defaults.
extern "C" const char* plume_last_msm_kernel(const plume_ctx* ctx) {
    if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
    if (ctx && ctx->lane_last) ctx = ctx->lane_last;
    return ctx ? ctx->last_msm_kernel : nullptr;
}

// device-resident verify / sign: how many sub-batches a call is cut into (verify_device); 1 = strictly serial launch order
extern "C" int plume_set_sub_batches(plume_ctx* ctx, int sub_batches) {
    if (!ctx || sub_batches < 1 || sub_batches > kMaxSubBatches) return fail(PLUME_ERR_ARG, "plume_set_sub_batches: bad argument");
    ctx->sub_batches = sub_batches;
    propagate_tunables(ctx);
    return 0;
}

// Batches in flight: with k > 1 the device-resident calls of this context go in turn to k lanes -- the context itself and k - 1 further single-device contexts of its own
// (each with its workspace, streams and events; the generator's fixed tables are shared) -- so that calls the caller issues on DIFFERENT streams run side by side instead
// of queueing for one workspace.  Whether they DO run side by side is the runtime's business: it multiplexes a process's streams onto GPU_MAX_HW_QUEUES (4 by default)
// hardware queues per priority level, and two caller streams that land on one queue run their kernels one after the other -- round 6's kernel trace of two torch streams
// showed every kernel of both on one queue, and round 5's "two small calls side by side gain nothing" was that.  With GPU_MAX_HW_QUEUES=8 in the process's environment
// (profiles/r06_hw_queues_small_calls.txt, one box, two lanes against one): 2^10-item verifies -49 % per call, 2^12 -31 %, 2^14 -26 %, 2^16 -13.6 % (1.17 instead of 1.36 ms),
// 2^17 -6.5 %, 2^18 -2.3 %, 2^20 within the spread; 2^20 signs -3 %; a third lane gains nothing more.  (Giving the second lane a high-priority stream of its own, tied to the
// caller's by events, does not do it: the queues then differ, but the command processor runs a high-priority queue's kernels ALONE -- strict alternation in the trace.)
// Results do not depend on any of it.  Calls on ONE stream stay in that stream's order whatever k is.  Default 1.
extern "C" int plume_set_in_flight(plume_ctx* ctx, int batches) {
    if (!ctx || batches < 1 || batches > 4) return fail(PLUME_ERR_ARG, "plume_set_in_flight: bad argument");
    if (!ctx->shards.empty()) return fail(PLUME_ERR_ARG, "plume_set_in_flight: a multi-device context runs its shards side by side already");
    while ((int)ctx->lanes.size() + 1 > batches) { destroy_single(ctx->lanes.back()); ctx->lanes.pop_back(); }
    while ((int)ctx->lanes.size() + 1 < batches) {
        plume_ctx* l = new plume_ctx();
        l->device = ctx->device;
        if (int rc = init_single(l)) { const std::string keep = g_err; destroy_single(l); g_err = keep; return rc; }
        inherit_tunables(l, ctx);
        ctx->lanes.push_back(l);
    }
    ctx->lane_next = 0; ctx->lane_last = nullptr;
    return 0;
}

// per-stage timing events inside the device pipelines (plume_last_stage_times); off by default
extern "C" int plume_set_stage_timing(plume_ctx* ctx, int on) {
    if (!ctx) return fail(PLUME_ERR_ARG, "plume_set_stage_timing: null context");
    ctx->timer.on = on != 0;
    propagate_tunables(ctx);
    return 0;
}

extern "C" int plume_set_host_piece(plume_ctx* ctx, size_t items) {
    if (!ctx || items == 0 || items > ((size_t)1 << 26)) return fail(PLUME_ERR_ARG, "plume_set_host_piece: bad argument");
    ctx->host_piece = items;
    propagate_tunables(ctx);
    return 0;
}
extern "C" int plume_set_host_first_piece(plume_ctx* ctx, size_t items) {
    if (!ctx || items == 0 || items > ((size_t)1 << 26)) return fail(PLUME_ERR_ARG, "plume_set_host_first_piece: bad argument");
    ctx->host_first_piece = items;
    propagate_tunables(ctx);
    return 0;
}
extern "C" int plume_set_host_tail_piece(plume_ctx* ctx, size_t items) {
    if (!ctx || items == 0 || items > ((size_t)1 << 26)) return fail(PLUME_ERR_ARG, "plume_set_host_tail_piece: bad argument");
    ctx->host_tail_piece = items;
    propagate_tunables(ctx);
    return 0;
}
extern "C" int plume_set_host_lanes(plume_ctx* ctx, int lanes) {
    if (!ctx || (lanes != 1 && lanes != 2)) return fail(PLUME_ERR_ARG, "plume_set_host_lanes: 1 or 2");
    ctx->host_lanes = lanes;
    propagate_tunables(ctx);
    return 0;
}
extern "C" int plume_set_host_register_min(plume_ctx* ctx, size_t bytes) {
    if (!ctx) return fail(PLUME_ERR_ARG, "plume_set_host_register_min: bad argument");
    ctx->host_register_min = bytes;
    propagate_tunables(ctx);
    return 0;
}

// table jobs per lane: more jobs share one inversion, but small batches need the lanes (>= ~4 workgroups per CU first)
static int pick_jobs_per_lane(const plume_ctx* ctx, size_t njobs, bool kinds_of_three) {
    if (ctx->jobs_per_lane_forced) return ctx->jobs_per_lane;
    size_t l = njobs / ((size_t)256 * 4 * kBlock);
    if (l > (size_t)ctx->jobs_per_lane) l = (size_t)ctx->jobs_per_lane;
    if (kinds_of_three) { l = (l / 3) * 3; if (l < 3) l = 3; }   // keep pk / H / nullifier kinds aligned across a wavefront
    if (l < 1) l = 1;
    return (int)l;
}

// nthrees: how many of the jobs, from the front, come as (pk, H, nullifier) triples (the verifier: 3 per item; its short first equation appends one more job per item behind
// them; the signer: 0): the jobs per lane are then a multiple of three, so that the kinds line up across a wavefront (affine and Jacobian bases take different paths).
static size_t table_stage_scratch(const plume_ctx* ctx, size_t njobs, size_t nthrees) { return tables_scratch_bytes(njobs, pick_jobs_per_lane(ctx, njobs, nthrees != 0)); }
static void table_stage(plume_ctx* ctx, uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, size_t nthrees, hipStream_t st) {
    launch_tables(tab, bases, jobflags, njobs, pick_jobs_per_lane(ctx, njobs, nthrees != 0), ctx->tabscr.as<uint32_t>(), st);
}

// ------------------------------------------------------------------------------------------ device pipelines
// How a device-resident call of n items is cut into sub-batches: the stages in front of the multi-scalar kernel (validation + hash_to_curve, window tables) of sub-batch
// k+1 run on ctx->pre beside the multi-scalar kernel of sub-batch k on the caller's stream.  The multi-scalar kernel saturates the vector ALUs but leaves HBM idle, the
// table kernel is bound by the critical path of its workgroups and leaves half of the issue slots idle (LABNOTES.md §5): side by side they fill each other's gaps, one
// after the other they cannot.  Slices are multiples of 1024 items (whole workgroups, aligned records).  One sub-batch = the strictly serial order.
static std::vector<size_t> sub_batch_bounds(const plume_ctx* ctx, size_t n) { return plume_host::sub_batch_bounds(n, ctx->sub_batches, ctx->overlap_min); }
static int pre_events(plume_ctx* ctx, size_t k) {
    while (ctx->pre_ready.size() < k) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pre_ready.push_back(e);
    }
    return 0;
}

static int side_stream(plume_ctx* ctx) { if (!ctx->side) HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking)); return 0; }
static int pre_stream(plume_ctx* ctx) { if (!ctx->pre) HIPCHK(hipStreamCreateWithFlags(&ctx->pre, hipStreamNonBlocking)); return 0; }

static int verify_device(plume_ctx* ctx, int version, int mode, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk, const uint8_t* nul,
                         const uint8_t* c, const uint8_t* s, const uint8_t* rpt, const uint8_t* hr, uint8_t* ok, hipStream_t st,
                         const uint8_t* preflags = nullptr, bool continue_timer = false, const uint8_t* rpt33 = nullptr, const uint8_t* hr33 = nullptr) {
    if (n == 0) return 0;
    if (n > ctx->chunk) return fail(PLUME_ERR_ARG, "n exceeds the chunk size (plume_set_chunk)");
    if (int rc = need_gtab(ctx)) return rc;
    // Equation 1 in its short form (plume_eis.h) whenever the call GIVES r_point as a 64-byte record -- V1 verify and verify_non_zk; the SEC1 ingest keeps R compressed
    // (a square root per item would cost more than the form saves) and V2 verify has no R: the long form.  One more table job per item (R), the comb of G.
    const bool eq1short = ctx->eq1_short != 0 && (n >= ctx->eq1_short_min || ctx->eq1_short >= 2) && rpt != nullptr && rpt33 == nullptr && (version == 1 || mode == PLUME_MODE_NON_ZK);
    if (eq1short) { if (int rc = need_gcomb(ctx)) return rc; }
    const size_t J = eq1short ? 4 : 3;
    if (!continue_timer) { if (int rc = ws_acquire(ctx, st)) return rc; }     // continue_timer: the caller (SEC1 ingest) holds the workspace already
    std::unique_ptr<WsHold> hold(continue_timer ? nullptr : new WsHold(ctx, st));
    const std::vector<size_t> cut = sub_batch_bounds(ctx, n);
    const size_t nsub = cut.size() - 1;
    const bool overlapped = nsub > 1;
    size_t scr_bytes = 0;
    for (size_t k = 0; k < nsub; k++) scr_bytes = std::max(scr_bytes, table_stage_scratch(ctx, J * (cut[k + 1] - cut[k]), 3 * (cut[k + 1] - cut[k])));
    if (ctx->bases.ensure((size_t)PLUME_BASE_WORDS * 4 * J * n) || ctx->jobflags.ensure(J * n) || ctx->itemflags.ensure(n) || ctx->tab.ensure((size_t)PLUME_TAB_WORDS * 4 * J * n) ||
        ctx->tabscr.ensure(scr_bytes) ||
        ctx->res.ensure((size_t)PLUME_JAC_WORDS * 4 * 2 * n) || ctx->resinf.ensure(2 * n) || ctx->redo.ensure((2 * n + nsub) * 4) || ctx->digs.ensure((size_t)PLUME_VDIG_ROWS * n) ||
        (eq1short && (ctx->eq1fall.ensure(n) || ctx->eq1k.ensure(32 * n))))
        return PLUME_ERR_HIP;
    if (overlapped) { if (int rc = pre_events(ctx, nsub)) return rc; if (int rc = pre_stream(ctx)) return rc; }
    StageTimer& t = ctx->timer;
    if (!continue_timer) t.begin(st);
    hipStream_t pre = overlapped ? ctx->pre : st;
    if (overlapped) {
        HIPCHK(hipEventRecord(ctx->pre_begin, st));                           // everything the caller's stream holds in front of this call (inputs, the SEC1 decompression, the workspace's last user)
        HIPCHK(hipStreamWaitEvent(pre, ctx->pre_begin, 0));
    }
    for (size_t k = 0; k < nsub; k++) {
        const size_t lo = cut[k], cnt = cut[k + 1] - cut[k];
        VerifyArgs a;                                                         // the slice [lo, lo + cnt) as a batch of its own: every array and every scratch region starts at the slice
        memset(&a, 0, sizeof a);
        a.version = version; a.mode = mode; a.n = (uint32_t)cnt; a.msgs = msgs; a.msg_off = msg_off + lo; a.msgs_bytes = msgs_bytes;
        a.pk = pk + 64 * lo; a.nul = nul + 64 * lo; a.c = c + 32 * lo; a.s = s + 32 * lo; a.rpt = rpt ? rpt + 64 * lo : nullptr; a.hr = hr ? hr + 64 * lo : nullptr; a.ok = ok + lo;
        a.preflags = preflags ? preflags + lo : nullptr; a.rpt33 = rpt33 ? rpt33 + 33 * lo : nullptr; a.hr33 = hr33 ? hr33 + 33 * lo : nullptr;
        a.bases = ctx->bases.as<uint32_t>() + (size_t)PLUME_BASE_WORDS * J * lo; a.jobflags = ctx->jobflags.as<uint8_t>() + J * lo; a.itemflags = ctx->itemflags.as<uint8_t>() + lo;
        a.tab = ctx->tab.as<uint32_t>() + (size_t)PLUME_TAB_WORDS * J * lo; a.res = ctx->res.as<uint32_t>() + (size_t)PLUME_JAC_WORDS * 2 * lo; a.resinf = ctx->resinf.as<uint8_t>() + 2 * lo;
        a.gtab = ctx->fixed->gtab.as<uint32_t>();
        a.redo = ctx->redo.as<uint32_t>() + 2 * lo + k;                      // the slice's redo list: counter + up to 2 * cnt tasks
        a.digs = ctx->digs.as<int8_t>() + (size_t)PLUME_VDIG_ROWS * lo;        // the slice's digit rows (row-major over the slice's cnt items)
        if (eq1short) {
            a.eq1fall = ctx->eq1fall.as<uint8_t>() + lo; a.eq1k = ctx->eq1k.as<uint32_t>() + 8 * lo; a.gcomb = ctx->fixed->gcomb.as<uint32_t>();
            a.eq1force = ctx->eq1_short == 2 ? 1 : 0;
        }
        if (k == 0) ctx->redo_counters.clear();
        ctx->redo_counters.push_back(2 * lo + k);
        const bool two_roles = cnt <= ctx->ingest_split_max;
        a.msm_pair = (!eq1short && cnt <= ctx->msm_pair_max) ? 1 : 0;             // a few thousand items: a chain's latency is the kernel's time, so each long-form chain runs as two halves on two lanes
        a.scalars_in_ingest = two_roles && ctx->split_scalars ? 1 : 0;             // the small-batch ingest kernel runs the scalar stage in its idle role: one launch less
        launch_verify_ingest(a, pre, two_roles); if (!overlapped) t.stage(a.scalars_in_ingest ? "verify_ingest_h2c+scalars" : "verify_ingest_h2c", st);
        if (!a.scalars_in_ingest) { launch_verify_scalars(a, pre); if (!overlapped) t.stage("verify_scalars", st); }
        table_stage(ctx, a.tab, a.bases, a.jobflags, J * cnt, 3 * cnt, pre); if (!overlapped) t.stage("tables", st);   // the table kernels of all sub-batches follow one another on one stream: one scratch
        if (overlapped) {
            HIPCHK(hipEventRecord(ctx->pre_ready[k], pre));
            HIPCHK(hipStreamWaitEvent(st, ctx->pre_ready[k], 0));
        }
        ctx->last_msm_kernel = verify_msm_kernel_name(a);
        if (k == 0) ctx->last_msm_sampled = false;
        if (t.on && !overlapped && k == 0) {                                   // stage timing: the multi-scalar kernel also samples its clocks (plume_last_msm_clock)
            if (ctx->clk.ensure(16)) return PLUME_ERR_HIP;
            HIPCHK(hipMemsetAsync(ctx->clk.p, 0, 16, st));
            a.clk = ctx->clk.as<unsigned long long>();
            ctx->last_msm_sampled = true;
        }
        launch_verify_msm(a, st); if (!overlapped) t.stage("verify_msm", st);
        if (version == 2 && mode == PLUME_MODE_VERIFY) { launch_normalize(a.res, a.resinf, 2 * cnt, st); if (!overlapped) t.stage("to_affine", st); }   // V2 hashes the computed R', Hr'
        launch_verify_finalize(a, st); if (!overlapped) t.stage("verify_finalize", st);
    }
    if (overlapped) t.stage("verify_overlapped", st);                         // per-stage times exist in the serial mode only (plume_set_sub_batches(ctx, 1) / PLUME_SERIAL=1)
    HIPCHK(hipGetLastError());
    return hold ? hold->release() : 0;
}

static int sign_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* sk, const uint8_t* r,
                       const uint8_t* pk_in, uint8_t* pk, uint8_t* nul, uint8_t* c, uint8_t* s, uint8_t* rpt, uint8_t* hr, uint8_t* status, uint8_t* h_out,
                       hipStream_t st, bool out33 = false) {
    if (n == 0) return 0;
    if (n > ctx->chunk) return fail(PLUME_ERR_ARG, "n exceeds the chunk size (plume_set_chunk)");
    if (int rc = need_sign_tables(ctx)) return rc;
    if (int rc = ws_acquire(ctx, st)) return rc;
    WsHold hold(ctx, st);
    const std::vector<size_t> cut = sub_batch_bounds(ctx, n);
    const size_t nsub = cut.size() - 1;
    const bool overlapped = nsub > 1;
    size_t scr_bytes = 0;
    constexpr size_t SK = PLUME_SIGN_K;                                          // table jobs per item: H and its shifted copies 2^(j PLUME_SIGN_BITS) H (the signer's chains are PLUME_SIGN_BITS doublings long)
    for (size_t k = 0; k < nsub; k++) scr_bytes = std::max(scr_bytes, table_stage_scratch(ctx, SK * (cut[k + 1] - cut[k]), 0));
    if (ctx->bases.ensure((size_t)PLUME_BASE_WORDS * 4 * SK * n) || ctx->jobflags.ensure(SK * n) || ctx->itemflags.ensure(n) || ctx->tab.ensure((size_t)PLUME_TAB_WORDS * 4 * SK * n) ||
        ctx->tabscr.ensure(scr_bytes) ||
        ctx->res.ensure((size_t)PLUME_JAC_WORDS * 4 * 2 * n) || ctx->resinf.ensure(2 * n) || ctx->res2.ensure((size_t)PLUME_JAC_WORDS * 4 * 2 * n) || ctx->res2inf.ensure(2 * n) || ctx->pkaff.ensure((size_t)2 * PLUME_FE_WORDS * 4 * n))
        return PLUME_ERR_HIP;
    if (overlapped) { if (int rc = pre_events(ctx, nsub)) return rc; if (int rc = pre_stream(ctx)) return rc; }
    StageTimer& t = ctx->timer;
    t.begin(st);
    hipStream_t pre = overlapped ? ctx->pre : st;                               // the stages in front of the H multiplications of sub-batch k+1 run beside those of sub-batch k (verify_device)
    if (overlapped) {
        HIPCHK(hipEventRecord(ctx->pre_begin, st));
        HIPCHK(hipStreamWaitEvent(pre, ctx->pre_begin, 0));
    }
    const size_t P = out33 ? 33 : 64;
    for (size_t k = 0; k < nsub; k++) {
        const size_t lo = cut[k], cnt = cut[k + 1] - cut[k];
        SignArgs a;
        a.version = version; a.n = (uint32_t)cnt; a.msgs = msgs; a.msg_off = msg_off + lo; a.msgs_bytes = msgs_bytes; a.sk = sk + 32 * lo; a.r = r + 32 * lo; a.pk_in = pk_in ? pk_in + 64 * lo : nullptr;
        a.pk = pk ? pk + P * lo : nullptr; a.nul = nul + P * lo; a.c = c + 32 * lo; a.s = s + 32 * lo; a.rpt = rpt + P * lo; a.hr = hr + P * lo; a.status = status + lo;
        a.h_out = h_out ? h_out + 64 * lo : nullptr; a.out33 = out33 ? 1 : 0;
        a.gres = ctx->res.as<uint32_t>() + (size_t)PLUME_JAC_WORDS * 2 * lo; a.gresinf = ctx->resinf.as<uint8_t>() + 2 * lo; a.bases = ctx->bases.as<uint32_t>() + (size_t)PLUME_BASE_WORDS * SK * lo;
        a.jobflags = ctx->jobflags.as<uint8_t>() + SK * lo; a.itemflags = ctx->itemflags.as<uint8_t>() + lo; a.pkaff = ctx->pkaff.as<uint32_t>() + (size_t)2 * PLUME_FE_WORDS * lo;
        a.tab = ctx->tab.as<uint32_t>() + (size_t)PLUME_TAB_WORDS * SK * lo; a.hres = ctx->res2.as<uint32_t>() + (size_t)PLUME_JAC_WORDS * 2 * lo; a.hresinf = ctx->res2inf.as<uint8_t>() + 2 * lo;
        a.gcomb = ctx->fixed->gcomb.as<uint32_t>(); a.gscan = ctx->fixed->gscan.as<uint32_t>(); a.uniform = ctx->sign_uniform;
        launch_sign_gmul(a, pre); if (!overlapped) t.stage("sign_gmul", st);
        launch_normalize(a.gres, a.gresinf, 2 * cnt, pre); if (!overlapped) t.stage("to_affine_g", st);
        launch_sign_h2c(a, pre); if (!overlapped) t.stage("sign_h2c", st);
        launch_sign_hdbl(a, pre); if (!overlapped) t.stage("sign_hdbl", st);
        table_stage(ctx, a.tab, a.bases, a.jobflags, SK * cnt, 0, pre); if (!overlapped) t.stage("tables", st);
        if (overlapped) {
            HIPCHK(hipEventRecord(ctx->pre_ready[k], pre));
            HIPCHK(hipStreamWaitEvent(st, ctx->pre_ready[k], 0));
        }
        launch_sign_hmul(a, st); if (!overlapped) t.stage("sign_hmul", st);
        launch_normalize(a.hres, a.hresinf, 2 * cnt, st); if (!overlapped) t.stage("to_affine_h", st);
        launch_sign_final(a, st); if (!overlapped) t.stage("sign_final", st);
    }
    if (overlapped) t.stage("sign_overlapped", st);
    // the reference zeroizes secrets: wipe the device-side images derived from sk / r
    HIPCHK(hipMemsetAsync(ctx->res.p, 0, (size_t)PLUME_JAC_WORDS * 4 * 2 * n, st));
    HIPCHK(hipGetLastError());
    return hold.release();
}

static int args_ok(int version, size_t n, const void* msgs, const void* off) {
    if (version != 1 && version != 2) return fail(PLUME_ERR_ARG, "version must be 1 or 2");
    if (n > 0 && (!msgs || !off)) return fail(PLUME_ERR_ARG, "null message buffers");
    if (n > 0xFFFFFFF0u) return fail(PLUME_ERR_ARG, "n too large");
    return 0;
}

extern "C" int plume_verify_batch_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                         const uint8_t* pk, const uint8_t* nullifier, const uint8_t* c, const uint8_t* s, const uint8_t* r_point,
                                         const uint8_t* hashed_to_curve_r, uint8_t* ok, void* stream) {
    Route rt_(ctx, stream, n); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk || !nullifier || !c || !s || !ok)) return fail(PLUME_ERR_ARG, "null array");
    if (n && version == 1 && (!r_point || !hashed_to_curve_r)) return fail(PLUME_ERR_ARG, "V1 needs r_point and hashed_to_curve_r");
    return verify_device(ctx, version, PLUME_MODE_VERIFY, n, msgs, msg_off, msgs_bytes, pk, nullifier, c, s, version == 1 ? r_point : nullptr,
                         version == 1 ? hashed_to_curve_r : nullptr, ok, st_);
}

// plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78): same pipeline, PLUME_MODE_NON_ZK
extern "C" int plume_verify_non_zk_batch_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                                const uint8_t* pk, const uint8_t* nullifier, const uint8_t* s, const uint8_t* r_point,
                                                const uint8_t* hashed_to_curve_r, const uint8_t* digest_private, uint8_t* ok, void* stream) {
    Route rt_(ctx, stream, n); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk || !nullifier || !s || !r_point || !hashed_to_curve_r || !digest_private || !ok)) return fail(PLUME_ERR_ARG, "null array");
    return verify_device(ctx, version, PLUME_MODE_NON_ZK, n, msgs, msg_off, msgs_bytes, pk, nullifier, digest_private, s, r_point, hashed_to_curve_r, ok,
                         st_);
}

// SEC1-compressed ingest: decompress on the GPU into the context's 64-byte staging arrays, then the normal pipeline
static int verify_sec1_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk33, const uint8_t* nul33,
                              const uint8_t* c, const uint8_t* s, const uint8_t* r33, const uint8_t* hr33, uint8_t* ok, hipStream_t st) {
    if (n == 0) return 0;
    if (n > ctx->chunk) return fail(PLUME_ERR_ARG, "n exceeds the chunk size (plume_set_chunk)");
    if (int rc = ws_acquire(ctx, st)) return rc;
    WsHold hold(ctx, st);
    // only pk and the nullifier are decompressed (they become bases of scalar multiplications); V1's r_point and hashed_to_curve_r stay
    // in their 33-byte form and are compared / hashed as x + parity by the finalize stage
    const int npts = 2;
    for (int k = 0; k < npts; k++) if (ctx->dec[k].ensure(64 * n)) return PLUME_ERR_HIP;
    if (ctx->preflags.ensure(n)) return PLUME_ERR_HIP;
    DecompressArgs d;
    d.n = (uint32_t)n; d.npts = npts;
    d.in[0] = pk33; d.in[1] = nul33; d.in[2] = r33; d.in[3] = hr33;
    for (int k = 0; k < 4; k++) d.out[k] = ctx->dec[k].as<uint8_t>();
    d.preflags = ctx->preflags.as<uint8_t>();
    ctx->timer.begin(st);
    launch_decompress(d, st); ctx->timer.stage("sec1_decompress", st);
    if (int rc = verify_device(ctx, version, PLUME_MODE_VERIFY, n, msgs, msg_off, msgs_bytes, d.out[0], d.out[1], c, s, nullptr, nullptr, ok, st, d.preflags, true,
                         version == 1 ? r33 : nullptr, version == 1 ? hr33 : nullptr)) return rc;
    return hold.release();
}

extern "C" int plume_verify_batch_sec1_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                              const uint8_t* pk33, const uint8_t* nullifier33, const uint8_t* c, const uint8_t* s, const uint8_t* r_point33,
                                              const uint8_t* hashed_to_curve_r33, uint8_t* ok, void* stream) {
    Route rt_(ctx, stream, n); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk33 || !nullifier33 || !c || !s || !ok)) return fail(PLUME_ERR_ARG, "null array");
    if (n && version == 1 && (!r_point33 || !hashed_to_curve_r33)) return fail(PLUME_ERR_ARG, "V1 needs r_point and hashed_to_curve_r");
    return verify_sec1_device(ctx, version, n, msgs, msg_off, msgs_bytes, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, ok, st_);
}

extern "C" int plume_sign_batch_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* sk,
                                       const uint8_t* r, const uint8_t* pk_in, uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point,
                                       uint8_t* hashed_to_curve_r, uint8_t* status, void* stream) {
    Route rt_(ctx, stream, n); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!sk || !r || !nullifier || !c || !s || !r_point || !hashed_to_curve_r || !status)) return fail(PLUME_ERR_ARG, "null array");
    return sign_device(ctx, version, n, msgs, msg_off, msgs_bytes, sk, r, pk_in, pk, nullifier, c, s, r_point, hashed_to_curve_r, status, nullptr,
                       st_);
}

extern "C" int plume_sign_batch_sec1_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* sk,
                                            const uint8_t* r, const uint8_t* pk_in, uint8_t* pk33, uint8_t* nullifier33, uint8_t* c, uint8_t* s, uint8_t* r_point33,
                                            uint8_t* hashed_to_curve_r33, uint8_t* status, void* stream) {
    Route rt_(ctx, stream, n); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!sk || !r || !nullifier33 || !c || !s || !r_point33 || !hashed_to_curve_r33 || !status)) return fail(PLUME_ERR_ARG, "null array");
    return sign_device(ctx, version, n, msgs, msg_off, msgs_bytes, sk, r, pk_in, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, status, nullptr,
                       st_, true);
}

// unrouted bodies: the host-pointer paths call these on the context itself (host calls never go to a lane: host_pipeline), the extern "C" wrappers route first
static int h2c_only_device(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk, uint8_t* h_out, hipStream_t st) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(1, n, msgs, msg_off)) return rc;
    if (n && !h_out) return fail(PLUME_ERR_ARG, "null array");
    if (n == 0) return 0;
    H2cArgs a; a.n = (uint32_t)n; a.msgs = msgs; a.msg_off = msg_off; a.msgs_bytes = msgs_bytes; a.pk = pk; a.h_out = h_out;
    ctx->timer.begin(st);
    launch_h2c_only(a, st); ctx->timer.stage("h2c_only", st);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int plume_hash_to_curve_batch_device(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk,
                                                uint8_t* h_out, void* stream) {
    Route rt_(ctx, stream); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    return h2c_only_device(ctx, n, msgs, msg_off, msgs_bytes, pk, h_out, st_);
}

// ------------------------------------------------------------------------------ circuit witness hints
static int h2c_inter_device(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk, int registers, uint8_t* u,
                            uint8_t* mapped, uint8_t* q, uint8_t* h, uint8_t* hints, void* stream) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(1, n, msgs, msg_off)) return rc;
    if (registers != 0 && registers != 1) return fail(PLUME_ERR_ARG, "registers must be 0 or 1");
    if (n == 0) return 0;
    H2cInterArgs a;
    a.n = (uint32_t)n; a.msgs = msgs; a.msg_off = msg_off; a.msgs_bytes = msgs_bytes; a.pk = pk; a.registers = registers; a.u = u; a.mapped = mapped; a.q = q; a.h = h; a.hints = hints;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->timer.begin(st);
    launch_h2c_intermediates(a, st); ctx->timer.stage("h2c_intermediates", st);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int plume_h2c_intermediates_batch_device(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk,
                                                    int registers, uint8_t* u, uint8_t* mapped, uint8_t* q, uint8_t* h, void* stream) {
    return h2c_inter_device(ctx, n, msgs, msg_off, msgs_bytes, pk, registers, u, mapped, q, h, nullptr, stream);
}
// the square-root hints of the two maps (UNPINNED definitions, include/plume_hip.h): n x 192 bytes
extern "C" int plume_h2c_hints_batch_device(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk, int registers,
                                            uint8_t* hints, void* stream) {
    if (n && !hints) return fail(PLUME_ERR_ARG, "null array");
    return h2c_inter_device(ctx, n, msgs, msg_off, msgs_bytes, pk, registers, nullptr, nullptr, nullptr, nullptr, hints, stream);
}
static int der_device(plume_ctx* ctx, size_t n, const uint8_t* scalars, uint8_t* der109, uint8_t* status, hipStream_t st) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!scalars || !der109 || !status)) return fail(PLUME_ERR_ARG, "null array");
    if (n > 0xFFFFFFF0u) return fail(PLUME_ERR_ARG, "n too large");
    if (n == 0) return 0;
    if (int rc = need_sign_tables(ctx)) return rc;
    if (int rc = ws_acquire(ctx, st)) return rc;          // no workspace is touched, but the kernel reads the shared table of G: the call joins the ws_free chain that plume_destroy waits on
    WsHold hold(ctx, st);
    DerArgs a; a.n = (uint32_t)n; a.scalars = scalars; a.der = der109; a.status = status; a.gcomb = ctx->fixed->gcomb.as<uint32_t>(); a.gscan = ctx->fixed->gscan.as<uint32_t>(); a.uniform = ctx->sign_uniform;
    ctx->timer.begin(st);
    launch_scalars_der(a, st); ctx->timer.stage("scalars_to_sec1_der", st);
    HIPCHK(hipGetLastError());
    return hold.release();
}
extern "C" int plume_scalars_to_sec1_der_batch_device(plume_ctx* ctx, size_t n, const uint8_t* scalars, uint8_t* der109, uint8_t* status, void* stream) {
    Route rt_(ctx, stream); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    return der_device(ctx, n, scalars, der109, status, st_);
}
extern "C" int plume_registers_from_be_device(plume_ctx* ctx, size_t nvalues, const uint8_t* be32, uint64_t* registers, void* stream) {
    Route rt_(ctx, stream); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (nvalues && (!be32 || !registers)) return fail(PLUME_ERR_ARG, "null array");
    if (nvalues == 0) return 0;
    launch_registers_from_be((uint8_t*)registers, be32, nvalues, st_);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------ aggregate check (plume_aggregate.h)
// Window width.  Only widths that divide 256 keep EVERY window's digits spread over all its buckets (a scalar below 2^255 in 15-bit windows, say, leaves
// a top window holding nothing but the Booth carry: one bucket would receive half of all terms, one lane would add them up); 16 bits from ~2^15 items
// (the bucket count, 2^19, is then what the fixed reduction cost buys), 8 bits below, 4 for a handful.  The pre-filter is built for large batches.
static int agg_window_bits(size_t n) { return n >= ((size_t)1 << 15) ? 16 : n >= 64 ? 8 : 4; }
// carry: device pointer to the record of the pieces before this one (or null); result: device pointer, PLUME_AGG_RESULT_BYTES
static int aggregate_device(plume_ctx* ctx, int version, int mode, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk, const uint8_t* nul,
                            const uint8_t* c, const uint8_t* s, const uint8_t* rpt, const uint8_t* hr, const uint8_t seed[32], uint64_t index_base, uint8_t* hash_ok,
                            const uint8_t* carry, uint8_t* result, hipStream_t st) {
    if (n > ctx->chunk) return fail(PLUME_ERR_ARG, "n exceeds the chunk size (plume_set_chunk)");
    if (int rc = need_gcomb(ctx)) return rc;
    // positions in the sorted pair array are 32-bit: 64 (term, window) pairs per item (69 for 8-bit windows) must stay far below 2^32, and the pair array itself (4 bytes
    // per pair) below what one pass should hold; larger batches go through the host-pointer form, which cuts them into pieces and carries the running record
    if (n > ((size_t)1 << 24)) return fail(PLUME_ERR_ARG, "plume_aggregate_check_device: at most 2^24 items per call (the host-pointer form cuts larger batches into pieces)");
    if (int rc = ws_acquire(ctx, st)) return rc;
    WsHold hold(ctx, st);
    AggArgs a;
    memset(&a, 0, sizeof a);
    a.version = version; a.mode = mode; a.n = (uint32_t)n;
    a.W = agg_window_bits(n);
    a.nw_long = (256 + a.W - 1) / a.W; a.nw_short = (128 + a.W - 1) / a.W;
    a.nbuckets = 1u << (a.W - 1); a.nkeys = (uint32_t)a.nw_long * a.nbuckets;
    a.index_base = index_base;
    memcpy(a.seed, seed, 32);
    // the windows go in two groups: the upper half (only the three long scalars reach it) is summed first, and while the lower half is summed on the
    // caller's stream the upper half's reduction -- a chain of small kernels ending in up to W*(nw-1) serial doublings -- runs beside it on ctx->side, as
    // does the generator term; the caller's stream picks both up before the last kernel
    const uint32_t nlo = (uint32_t)a.nw_long / 2, nhi = (uint32_t)a.nw_long - nlo;
    const size_t npairs = n * (size_t)(3 * a.nw_long + 2 * a.nw_short), nred = agg_reduce_points(a, nhi), sw = agg_scalar_sum_words(n ? n : 1);
    DevBuf* B = ctx->agg;
    if (ctx->bases.ensure((size_t)PLUME_BASE_WORDS * 4 * 3 * n + 16) || ctx->jobflags.ensure(3 * n + 16) || ctx->itemflags.ensure(n + 16) ||
        B[0].ensure(64 * n + 16) || B[1].ensure((size_t)PLUME_AGG_TERMS * 32 * n + 16) || B[2].ensure(2 * n + 16) || B[3].ensure(32 * n + 32) || B[4].ensure(n + 16) ||
        B[5].ensure(16 + 4 * (size_t)kAggScanLanes) || B[6].ensure(((size_t)a.nkeys + 1) * 4) || B[7].ensure(agg_sort_tile_words(a) * 4) || B[8].ensure(npairs * 4 + 16) ||
        B[9].ensure((size_t)PLUME_JAC_WORDS * 4 * a.nkeys) || B[10].ensure(a.nkeys) || B[11].ensure(4 * (size_t)PLUME_JAC_WORDS * 4 * nred) || B[12].ensure(4 * nred) ||
        B[13].ensure((2 * sw + PLUME_JAC_WORDS + 4) * 4) || B[14].ensure(((size_t)a.nkeys + 2 * (size_t)kAggPermBlocks * PLUME_AGG_LEN_BINS) * 4))
        return PLUME_ERR_HIP;
    a.pk = pk; a.nul = nul; a.c = c; a.s = s; a.rpt = rpt; a.hr = hr;
    a.bases = ctx->bases.as<uint32_t>(); a.jobflags = ctx->jobflags.as<uint8_t>(); a.itemflags = ctx->itemflags.as<uint8_t>();
    a.haff = B[0].as<uint8_t>(); a.scal = B[1].as<uint32_t>(); a.tlive = B[2].as<uint8_t>(); a.tneg = B[2].as<uint8_t>() + n; a.gs = B[3].as<uint32_t>();
    a.hash_ok = hash_ok ? hash_ok : B[4].as<uint8_t>(); a.nbad = B[5].as<uint32_t>();
    a.count = B[6].as<uint32_t>(); a.sorted = B[8].as<uint32_t>(); a.bsum = B[9].as<uint32_t>(); a.bsuminf = B[10].as<uint8_t>();
    a.gcomb = ctx->fixed->gcomb.as<uint32_t>(); a.result = result;
    uint32_t* red[4]; uint8_t* rinf[4];                     // lower group: 0, 1; upper group: 2, 3
    for (int k = 0; k < 4; k++) { red[k] = B[11].as<uint32_t>() + (size_t)k * PLUME_JAC_WORDS * nred; rinf[k] = B[12].as<uint8_t>() + (size_t)k * nred; }
    uint32_t* gpt = B[13].as<uint32_t>() + 2 * sw; uint8_t* gptinf = (uint8_t*)(gpt + PLUME_JAC_WORDS);
    uint32_t* perm = B[14].as<uint32_t>(); uint32_t* hist = perm + a.nkeys;
    if (int rc = side_stream(ctx)) return rc;
    hipStream_t side = ctx->side;
    StageTimer& t = ctx->timer;
    t.begin(st);
    HIPCHK(hipMemsetAsync(a.nbad, 0, 16, st));
    HIPCHK(hipMemsetAsync(a.count, 0, ((size_t)a.nkeys + 1) * 4, st));
    if (n) {
        VerifyArgs v;
        memset(&v, 0, sizeof v);
        v.version = version; v.mode = mode; v.n = (uint32_t)n; v.msgs = msgs; v.msg_off = msg_off; v.msgs_bytes = msgs_bytes; v.pk = pk; v.nul = nul; v.c = c; v.s = s; v.rpt = rpt; v.hr = hr;
        v.bases = ctx->bases.as<uint32_t>(); v.jobflags = ctx->jobflags.as<uint8_t>(); v.itemflags = ctx->itemflags.as<uint8_t>();
        launch_verify_ingest(v, st); t.stage("verify_ingest_h2c", st);
        launch_agg_normalize_h(a, st); t.stage("agg_h_affine", st);
        launch_agg_item_terms(a, st); t.stage("agg_item_terms", st);
    } else {
        HIPCHK(hipMemsetAsync(a.gs, 0, 32, st));
    }
    HIPCHK(hipEventRecord(ctx->agg_ev[0], st));
    HIPCHK(hipStreamWaitEvent(side, ctx->agg_ev[0], 0));
    const uint32_t* gsum = launch_agg_scalar_sum(a.gs, n ? n : 1, B[13].as<uint32_t>(), B[13].as<uint32_t>() + sw, side);
    launch_agg_gterm(a, gsum, gpt, gptinf, side);
    HIPCHK(hipEventRecord(ctx->agg_ev[2], side));
    if (n) { launch_agg_sort(a, B[7].as<uint32_t>(), B[5].as<uint32_t>() + 4, st); t.stage("agg_bucket_sort", st); }
    launch_agg_bucket_sum(a, nlo, nhi, perm, hist, st); t.stage("agg_bucket_sum_upper", st);
    HIPCHK(hipEventRecord(ctx->agg_ev[1], st));
    HIPCHK(hipStreamWaitEvent(side, ctx->agg_ev[1], 0));
    const int chi = launch_agg_reduce(a, nlo, nhi, red[2], rinf[2], red[3], rinf[3], side);
    HIPCHK(hipEventRecord(ctx->agg_ev[3], side));
    launch_agg_bucket_sum(a, 0, nlo, perm, hist + (size_t)kAggPermBlocks * PLUME_AGG_LEN_BINS, st); t.stage("agg_bucket_sum_lower", st);
    const int clo = launch_agg_reduce(a, 0, nlo, red[0], rinf[0], red[1], rinf[1], st);
    HIPCHK(hipStreamWaitEvent(st, ctx->agg_ev[2], 0));
    HIPCHK(hipStreamWaitEvent(st, ctx->agg_ev[3], 0));
    launch_agg_final(a, red[clo], rinf[clo], nlo, red[2 + chi], rinf[2 + chi], nhi, gpt, gptinf, carry, st); t.stage("agg_reduce", st);
    HIPCHK(hipGetLastError());
    return hold.release();
}

static int agg_args_ok(int version, int mode, size_t n, const void* msgs, const void* off, const void* seed) {
    if (int rc = args_ok(version, n, msgs, off)) return rc;
    if (mode != PLUME_MODE_VERIFY && mode != PLUME_MODE_NON_ZK) return fail(PLUME_ERR_ARG, "mode must be 0 (verify) or 1 (verify_non_zk)");
    if (mode == PLUME_MODE_VERIFY && version != 1) return fail(PLUME_ERR_ARG, "the aggregate check needs the GIVEN r_point / hashed_to_curve_r: V1 verify, or verify_non_zk");
    (void)seed;                                                     // NULL = the library draws 32 bytes from the OS (agg_seed)
    return 0;
}
// the coefficients a_i, b_i are only as unpredictable as the seed: a caller that has no fresh randomness of its own passes NULL and gets 32 bytes of the OS generator
// Drawn with getrandom(2) and checked: nothing here can throw across the C ABI (std::random_device may), and a short read is an error, not a weaker seed.
static bool os_random(void* out, size_t len) {
    uint8_t* p = (uint8_t*)out;
    while (len) {
        const ssize_t k = getrandom(p, len, 0);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        p += k; len -= (size_t)k;
    }
    return true;
}
static const uint8_t* agg_seed(const uint8_t* seed, uint8_t drawn[32]) {      // nullptr: the OS generator failed
    if (seed) return seed;
    return os_random(drawn, 32) ? drawn : nullptr;
}

extern "C" int plume_aggregate_check_device(plume_ctx* ctx, int version, int mode, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk,
                                            const uint8_t* nullifier, const uint8_t* c, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                                            const uint8_t seed[32], uint64_t index_base, uint8_t* hash_ok, uint8_t* result, void* stream) {
    Route rt_(ctx, stream); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (int rc = agg_args_ok(version, mode, n, msgs, msg_off, seed)) return rc;
    if (!result || (n && (!pk || !nullifier || !c || !s || !r_point || !hashed_to_curve_r))) return fail(PLUME_ERR_ARG, "null array");
    uint8_t drawn[32];
    seed = agg_seed(seed, drawn);
    if (!seed) return fail(PLUME_ERR_HIP, "getrandom failed: no seed for the aggregate check's coefficients");
    return aggregate_device(ctx, version, mode, n, msgs, msg_off, msgs_bytes, pk, nullifier, c, s, r_point, hashed_to_curve_r, seed, index_base, hash_ok, nullptr, result, st_);
}

// ------------------------------------------------------------------------------ nullifier-set post-processing
static int dedup_device(plume_ctx* ctx, size_t n, const uint8_t* nul, const uint8_t* live, const uint64_t* ids, uint8_t* first, uint64_t* n_unique_dev, hipStream_t st) {
    if (n == 0) { if (n_unique_dev) HIPCHK(hipMemsetAsync(n_unique_dev, 0, 8, st)); return 0; }
    if (n > ((size_t)1 << 30)) return fail(PLUME_ERR_ARG, "n too large");
    DedupArgs a;
    a.n = (uint32_t)n; a.nul = nul; a.live = live; a.ids = ids; a.first = first;
    const uint32_t m = dedup_table_size(a.n);
    a.mask = m - 1;
    if (!os_random(a.key, sizeof a.key)) return fail(PLUME_ERR_HIP, "getrandom failed: no hash key for the nullifier table");   // fresh hash key per call (plume_dedup.h)
    a.key[1] |= 1u;
    if (ctx->dslots.ensure((size_t)m * 4) || ctx->dminid.ensure((size_t)m * 8) || ctx->dmyslot.ensure(n * 4) || ctx->dcount.ensure(8) || ctx->dblockcnt.ensure(dedup_blockcnt_bytes(n))) return PLUME_ERR_HIP;
    a.slots = ctx->dslots.as<uint32_t>(); a.minid = ctx->dminid.as<unsigned long long>(); a.myslot = ctx->dmyslot.as<uint32_t>();
    a.n_unique = ctx->dcount.as<unsigned long long>(); a.blockcnt = ctx->dblockcnt.as<uint32_t>();
    if (int rc = ws_acquire(ctx, st)) return rc;
    WsHold hold(ctx, st);
    ctx->timer.begin(st);
    launch_dedup(a, st); ctx->timer.stage("nullifier_first_occurrence", st);
    HIPCHK(hipGetLastError());
    if (n_unique_dev) HIPCHK(hipMemcpyAsync(n_unique_dev, ctx->dcount.p, 8, hipMemcpyDeviceToDevice, st));
    return hold.release();
}
extern "C" int plume_nullifier_first_occurrence_device(plume_ctx* ctx, size_t n, const uint8_t* nullifier, const uint8_t* live, const uint64_t* ids, uint8_t* first,
                                                       uint64_t* n_unique, void* stream) {
    Route rt_(ctx, stream); ctx = rt_.lane; const hipStream_t st_ = rt_.st;
    if (int rc = bind(ctx)) return rc;
    if (n && (!nullifier || !first)) return fail(PLUME_ERR_ARG, "null array");
    return dedup_device(ctx, n, nullifier, live, ids, first, n_unique, st_);
}
// host-pointer form: one pass (every record has to be resident to be compared), staged through slot 0
extern "C" int plume_nullifier_first_occurrence(plume_ctx* ctx, size_t n, const uint8_t* nullifier, const uint8_t* live, const uint64_t* ids, uint8_t* first,
                                                uint64_t* n_unique) {
    if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];   // every record has to meet every other: one device holds the whole set
    if (int rc = bind(ctx)) return rc;
    if (n && (!nullifier || !first)) return fail(PLUME_ERR_ARG, "null array");
    if (n == 0) { if (n_unique) *n_unique = 0; return 0; }
    HostSlot& sl = ctx->slot[0];
    hipStream_t st = ctx->stream;
    if (sl.in[0].ensure(64 * n) || sl.in[1].ensure(n) || sl.in[2].ensure(8 * n) || sl.out[0].ensure(n) || sl.out[1].ensure(8)) return PLUME_ERR_HIP;
    HIPCHK(hipMemcpyAsync(sl.in[0].p, nullifier, 64 * n, hipMemcpyHostToDevice, st));
    if (live) HIPCHK(hipMemcpyAsync(sl.in[1].p, live, n, hipMemcpyHostToDevice, st));
    if (ids) HIPCHK(hipMemcpyAsync(sl.in[2].p, ids, 8 * n, hipMemcpyHostToDevice, st));
    if (int rc = dedup_device(ctx, n, sl.in[0].as<uint8_t>(), live ? sl.in[1].as<uint8_t>() : nullptr, ids ? sl.in[2].as<uint64_t>() : nullptr, sl.out[0].as<uint8_t>(),
                              sl.out[1].as<uint64_t>(), st))
        return rc;
    HIPCHK(hipMemcpyAsync(first, sl.out[0].p, n, hipMemcpyDeviceToHost, st));
    uint64_t cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, sl.out[1].p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_unique) *n_unique = cnt;
    return 0;
}

// ------------------------------------------------------------------------------------- host-pointer pipelines
// The batch is cut into pieces (ctx->host_piece items, the first one ctx->host_first_piece, at most ctx->chunk).  Piece k+1 is staged
// into HBM on the upload stream while piece k computes on the context stream and piece k-1 drains on the download stream; two staging
// slots alternate.  Caller memory that is page-locked (plume_host_alloc / plume_host_register, or any hipHostMalloc'ed / registered
// range) is read and written by the copy engines directly and every copy is asynchronous; with pageable caller memory the copies block
// the calling thread, which is why piece k-1 is drained only AFTER piece k has been submitted: the thread then waits on work that is
// already behind it in the queue.
static int stage_msgs(plume_ctx* ctx, HostSlot& sl, const uint8_t* msgs, const uint64_t* off, size_t i0, size_t cnt) {
    if (sl.relbuf.ensure((cnt + 1) * 8)) return PLUME_ERR_HIP;
    uint64_t* rel = sl.rel = (uint64_t*)sl.relbuf.p;
    const uint64_t base = off[i0];
    switch (plume_host::rebase_offsets(off, i0, cnt, rel)) {
        case 0: break;
        case 1: return fail(PLUME_ERR_ARG, "msg_off is not non-decreasing");
        default: return fail(PLUME_ERR_ARG, "message bytes per pass exceed 4 GiB");
    }
    if (sl.msgs.ensure((size_t)rel[cnt] + 16) || sl.off.ensure((cnt + 1) * 8)) return PLUME_ERR_HIP;
    if (rel[cnt]) HIPCHK(hipMemcpyAsync(sl.msgs.p, msgs + base, (size_t)rel[cnt], hipMemcpyHostToDevice, ctx->up));
    HIPCHK(hipMemcpyAsync(sl.off.p, rel, (cnt + 1) * 8, hipMemcpyHostToDevice, ctx->up));
    return 0;
}
static int h2d(plume_ctx* ctx, DevBuf& b, const uint8_t* src, size_t bytes) {
    if (b.ensure(bytes)) return PLUME_ERR_HIP;
    HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->up));
    return 0;
}
static int d2h(plume_ctx* ctx, uint8_t* dst, const DevBuf& b, size_t bytes) {
    HIPCHK(hipMemcpyAsync(dst, b.p, bytes, hipMemcpyDeviceToHost, ctx->down));
    return 0;
}
static void quiesce(plume_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->up);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->down);
    if (ctx->host_lane) (void)hipStreamSynchronize(ctx->host_lane->stream);
    for (HostSlot& sl : ctx->slot) sl.in_flight = false;
}

// Page-locks caller arrays for the duration of one host-pointer call when ctx->host_register_min asks for it (arrays that are already
// pinned, or too small, are left alone; a failed registration just leaves that array pageable).
struct ScopedPins {
    std::vector<void*> pinned;
    void add(const plume_ctx* ctx, const void* p, size_t bytes) {
        if (!p || !ctx->host_register_min || bytes < ctx->host_register_min) return;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type != hipMemoryTypeUnregistered) return;   // already known to the runtime
        (void)hipGetLastError();
        if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess) pinned.push_back(const_cast<void*>(p));
        else (void)hipGetLastError();
    }
    ~ScopedPins() { for (void* p : pinned) (void)hipHostUnregister(p); }
};

// up(slot, i0, cnt): enqueue the uploads of one piece on ctx->up;  run(slot, cnt): enqueue its kernels on ctx->stream;
// down(slot, i0, cnt): enqueue the downloads on ctx->down.  Any error drains all three streams before it is returned, so no
// copy is left in flight on the caller's memory.
// Piece schedule: the first piece is small (nothing hides its upload), every following piece may be up to three times the one before it (an
// upload runs at ~7 ns per item, the kernels at ~23 ns per item, so piece k+1's upload still hides behind piece k's kernels) up to the
// largest piece; calls with large outputs (the signer: 320 bytes out per item) also end on a small piece, whose download nothing hides.
static std::vector<size_t> piece_schedule(const plume_ctx* ctx, size_t n, bool out_heavy, int lanes) {
    plume_host::PieceKnobs kn{ctx->chunk, ctx->host_piece, ctx->host_first_piece, ctx->host_tail_piece};
    kn.out_lanes = lanes;
    return plume_host::piece_schedule(kn, n, out_heavy, std::getenv("PLUME_HOST_SCHEDULE"));     // the rules and the experiment knob: plume_host_logic.h
}

// The second lane of the host-pointer pipeline (round 4).  Rounds 1-3 ran every piece of a call on the context's one workspace and stream: the kernels of piece k+1 queued
// behind those of piece k, the small first pieces ran at small-batch efficiency (a 2^16-item verify alone reaches 0.70 of the 2^20 rate) and the whole call delivered 0.85 of
// the device-resident rate.  Now the pieces alternate between the context and a lane of its own kind (workspace, streams; the generator tables are shared): the ingest and
// table stages of piece k+1 run beside the multi-scalar kernel of piece k exactly as two device-resident batches in flight do, which needs four staging slots (k-2 is still
// downloading from its slot when k+2 wants to upload).
static plume_ctx* host_lane_of(plume_ctx* ctx) {
    if (ctx->host_lane) return ctx->host_lane;
    if (ctx->host_lane_failed) return nullptr;
    plume_ctx* l = new plume_ctx();
    l->device = ctx->device;
    if (init_single(l)) {
        // the call goes on with one lane and succeeds: it must not leave this failure's text behind as "the last error", nor try the allocation again on every later call
        destroy_single(l);
        g_err.clear();
        ctx->host_lane_failed = true;
        return nullptr;
    }
    ctx->host_lane = l;
    propagate_tunables(ctx);
    return l;
}

// page-locked (hipHostMalloc / hipHostRegister) or device-accessible memory?  Pageable caller arrays make the runtime stage every copy and move it with blit KERNELS, which
// queue behind the compute kernels: with two lanes keeping the machine saturated such a call got slower (26.7 vs 23.5 ms per 2^20 verifies), so it stays on one lane.
static bool is_page_locked(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// PLUME_HOST_TRACE=1: every host-pointer call prints, per piece, when its uploads, kernels and downloads started and ended on the GPU's clock (timing events on the three
// streams, read after the call) -- the pipeline's timeline WITHOUT a profiler attached (rocprofv3's memory-copy trace turns the downloads into blit kernels and stretches the
// call: round 4 chased that artefact, tests/gpu_debug/d2h_probe.hip + d2h_in_library.py show the copies of an unprofiled run on the copy engines).
struct PieceTrace { size_t cnt; int lane; hipEvent_t ev[6]; };    // up begin / end, run begin / end, down begin / end
static bool host_trace_on() { static const bool on = [] { const char* e = std::getenv("PLUME_HOST_TRACE"); return e && std::atoi(e) != 0; }(); return on; }

template <class Up, class Run, class Down>
static int host_pipeline(plume_ctx* ctx, size_t n, bool out_heavy, Up up, Run run, Down down, bool may_use_two_lanes = false) {
    const std::vector<size_t> sched = piece_schedule(ctx, n, out_heavy, (may_use_two_lanes && ctx->host_lanes > 1) ? 2 : 1);
    plume_ctx* lane2 = (may_use_two_lanes && ctx->host_lanes > 1 && sched.size() > 1) ? host_lane_of(ctx) : nullptr;
    const size_t nslots = lane2 ? 4 : 2;
    ctx->lane_last = nullptr;            // host-pointer calls report the stage timer of the lane their last piece ran on (set below)
    const bool trace = host_trace_on();
    std::vector<PieceTrace> tr;
    auto mark = [&](size_t k, int which, hipStream_t st) { if (trace) (void)hipEventRecord(tr[k].ev[which], st); };
    if (trace) {
        tr.resize(sched.size());
        for (size_t k = 0; k < sched.size(); k++) { tr[k].cnt = sched[k]; tr[k].lane = (lane2 && (k & 1)) ? 1 : 0; for (hipEvent_t& e : tr[k].ev) (void)hipEventCreate(&e); }
    }
    struct { HostSlot* sl = nullptr; size_t i0 = 0, cnt = 0, k = 0; } prev;
    auto drain = [&]() -> int {
        HIPCHK(hipStreamWaitEvent(ctx->down, prev.sl->computed, 0));
        mark(prev.k, 4, ctx->down);
        if (int rc = down(*prev.sl, prev.i0, prev.cnt)) return rc;
        HIPCHK(hipEventRecord(prev.sl->drained, ctx->down));
        mark(prev.k, 5, ctx->down);
        return 0;
    };
    auto body = [&]() -> int {
        size_t i0 = 0;
        for (size_t k = 0; k < sched.size(); k++) {
            const size_t cnt = sched[k];
            HostSlot& sl = ctx->slot[k % nslots];
            plume_ctx* on = (lane2 && (k & 1)) ? lane2 : ctx;
            if (sl.in_flight) { HIPCHK(hipEventSynchronize(sl.drained)); sl.in_flight = false; }   // the piece that used this slot last has left it
            mark(k, 0, ctx->up);
            if (int rc = up(sl, i0, cnt)) return rc;
            HIPCHK(hipEventRecord(sl.ready, ctx->up));
            mark(k, 1, ctx->up);
            HIPCHK(hipStreamWaitEvent(on->stream, sl.ready, 0));
            mark(k, 2, on->stream);
            if (int rc = run(sl, cnt, on)) return rc;
            HIPCHK(hipEventRecord(sl.computed, on->stream));
            mark(k, 3, on->stream);
            sl.in_flight = true;
            ctx->lane_last = on == ctx ? nullptr : on;
            if (prev.sl) { if (int rc = drain()) return rc; }
            prev.sl = &sl; prev.i0 = i0; prev.cnt = cnt; prev.k = k;
            i0 += cnt;
        }
        if (prev.sl) { if (int rc = drain()) return rc; }
        return 0;
    };
    const int rc = body();
    quiesce(ctx);
    if (trace) {
        if (rc == 0) {
            std::string out = "plume_host_trace: " + std::to_string(n) + " items in " + std::to_string(sched.size()) + " pieces" + (lane2 ? ", two lanes" : ", one lane") + "  (ms since the first upload began)\n";
            for (size_t k = 0; k < tr.size(); k++) {
                float t[6] = {0, 0, 0, 0, 0, 0};
                for (int j = 0; j < 6; j++) (void)hipEventElapsedTime(&t[j], tr[0].ev[0], tr[k].ev[j]);
                char line[256];
                std::snprintf(line, sizeof line, "  piece %2zu lane %d %8zu items | up %7.3f -%7.3f | run %7.3f -%7.3f (%6.3f) | down %7.3f -%7.3f (%6.3f)\n", k, tr[k].lane, tr[k].cnt, t[0], t[1], t[2], t[3],
                              t[3] - t[2], t[4], t[5], t[5] - t[4]);
                out += line;
            }
            std::fputs(out.c_str(), stderr);
        }
        (void)hipGetLastError();
        for (PieceTrace& p : tr) for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
    }
    if (rc == 0) HIPCHK(hipGetLastError());
    return rc;
}

// one single-device context, items [0, n) of the arrays given (a shard of a multi-device call passes offset pointers; msg_off keeps the
// caller's absolute offsets into the same msgs buffer)
static int verify_host(plume_ctx* ctx, int version, int mode, bool sec1, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nullifier,
                       const uint8_t* c, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r, uint8_t* ok) {
    HIPCHK(hipSetDevice(ctx->device));
    const bool pts = version == 1 || mode == PLUME_MODE_NON_ZK;
    const size_t P = sec1 ? 33 : 64;
    ScopedPins pins;
    if (n) {
        pins.add(ctx, msgs + msg_off[0], (size_t)(msg_off[n] - msg_off[0]));
        pins.add(ctx, pk, P * n); pins.add(ctx, nullifier, P * n); pins.add(ctx, c, 32 * n); pins.add(ctx, s, 32 * n);
        if (pts) { pins.add(ctx, r_point, P * n); pins.add(ctx, hashed_to_curve_r, P * n); }
    }
    // two lanes only when EVERY array of the call is page-locked: one pageable array is enough to send its copies through blit kernels, which queue behind the saturating
    // kernels of two lanes and make the call slower than one lane would be (round 4 looked at pk and nullifier only)
    bool all_locked = n > 0;
    for (const void* a : {(const void*)pk, (const void*)nullifier, (const void*)c, (const void*)s, (const void*)ok, pts ? (const void*)r_point : (const void*)pk,
                          pts ? (const void*)hashed_to_curve_r : (const void*)pk, (n && msg_off[n] > msg_off[0]) ? (const void*)(msgs + msg_off[0]) : (const void*)pk})
        all_locked = all_locked && is_page_locked(a);
    return host_pipeline(
        ctx, n, false,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int rc = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[0], pk + P * i0, P * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[1], nullifier + P * i0, P * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[2], c + 32 * i0, 32 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[3], s + 32 * i0, 32 * cnt)) return rc;
            if (pts) {
                if (int rc = h2d(ctx, sl.in[4], r_point + P * i0, P * cnt)) return rc;
                if (int rc = h2d(ctx, sl.in[5], hashed_to_curve_r + P * i0, P * cnt)) return rc;
            }
            return sl.out[0].ensure(cnt);
        },
        [&](HostSlot& sl, size_t cnt, plume_ctx* on) -> int {
            const uint8_t *rp = pts ? sl.in[4].as<uint8_t>() : nullptr, *hp = pts ? sl.in[5].as<uint8_t>() : nullptr;
            if (sec1)
                return verify_sec1_device(on, version, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), (size_t)sl.rel[cnt], sl.in[0].as<uint8_t>(), sl.in[1].as<uint8_t>(),
                                          sl.in[2].as<uint8_t>(), sl.in[3].as<uint8_t>(), rp, hp, sl.out[0].as<uint8_t>(), on->stream);
            return verify_device(on, version, mode, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), (size_t)sl.rel[cnt], sl.in[0].as<uint8_t>(), sl.in[1].as<uint8_t>(),
                                 sl.in[2].as<uint8_t>(), sl.in[3].as<uint8_t>(), rp, hp, sl.out[0].as<uint8_t>(), on->stream);
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int { return d2h(ctx, ok + i0, sl.out[0], cnt); }, all_locked);
}

static int verify_host_any(plume_ctx* ctx, int version, int mode, bool sec1, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nullifier,
                           const uint8_t* c, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r, uint8_t* ok) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk || !nullifier || !c || !s || !ok)) return fail(PLUME_ERR_ARG, "null array");
    const bool pts = version == 1 || mode == PLUME_MODE_NON_ZK;
    if (n && pts && (!r_point || !hashed_to_curve_r)) return fail(PLUME_ERR_ARG, "r_point and hashed_to_curve_r are required");
    if (n == 0) return 0;
    const size_t P = sec1 ? 33 : 64;
    if (ctx->shards.empty()) return verify_host(ctx, version, mode, sec1, n, msgs, msg_off, pk, nullifier, c, s, pts ? r_point : nullptr, pts ? hashed_to_curve_r : nullptr, ok);
    return for_shards(ctx, n, [=](plume_ctx* sh, size_t lo, size_t hi) -> int {
        return verify_host(sh, version, mode, sec1, hi - lo, msgs, msg_off + lo, pk + P * lo, nullifier + P * lo, c + 32 * lo, s + 32 * lo, pts ? r_point + P * lo : nullptr,
                           pts ? hashed_to_curve_r + P * lo : nullptr, ok + lo);
    });
}

extern "C" int plume_verify_batch(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk,
                                  const uint8_t* nullifier, const uint8_t* c, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                                  uint8_t* ok) {
    return verify_host_any(ctx, version, PLUME_MODE_VERIFY, false, n, msgs, msg_off, pk, nullifier, c, s, r_point, hashed_to_curve_r, ok);
}
extern "C" int plume_verify_batch_sec1(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk33,
                                       const uint8_t* nullifier33, const uint8_t* c, const uint8_t* s, const uint8_t* r_point33,
                                       const uint8_t* hashed_to_curve_r33, uint8_t* ok) {
    return verify_host_any(ctx, version, PLUME_MODE_VERIFY, true, n, msgs, msg_off, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, ok);
}
extern "C" int plume_verify_non_zk_batch(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk,
                                         const uint8_t* nullifier, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                                         const uint8_t* digest_private, uint8_t* ok) {
    return verify_host_any(ctx, version, PLUME_MODE_NON_ZK, false, n, msgs, msg_off, pk, nullifier, digest_private, s, r_point, hashed_to_curve_r, ok);
}

// aggregate check over host arrays: the pieces of the batch run through the same three-stream pipeline; every piece adds its sum to the running
// record kept in HBM (coefficients are indexed by the item's position in the WHOLE batch, so the record does not depend on how the batch was cut)
static int aggregate_host(plume_ctx* ctx, int version, int mode, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nullifier,
                          const uint8_t* c, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r, const uint8_t* seed, uint64_t index_base,
                          uint8_t* hash_ok, uint8_t* record) {
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->agg_record.ensure(PLUME_AGG_RESULT_BYTES)) return PLUME_ERR_HIP;
    uint8_t* rec = ctx->agg_record.as<uint8_t>();
    ScopedPins pins;
    pins.add(ctx, msgs + msg_off[0], (size_t)(msg_off[n] - msg_off[0]));
    pins.add(ctx, pk, 64 * n); pins.add(ctx, nullifier, 64 * n); pins.add(ctx, c, 32 * n); pins.add(ctx, s, 32 * n); pins.add(ctx, r_point, 64 * n); pins.add(ctx, hashed_to_curve_r, 64 * n);
    size_t done = 0;
    int rc = host_pipeline(
        ctx, n, false,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int r = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return r;
            if (int r = h2d(ctx, sl.in[0], pk + 64 * i0, 64 * cnt)) return r;
            if (int r = h2d(ctx, sl.in[1], nullifier + 64 * i0, 64 * cnt)) return r;
            if (int r = h2d(ctx, sl.in[2], c + 32 * i0, 32 * cnt)) return r;
            if (int r = h2d(ctx, sl.in[3], s + 32 * i0, 32 * cnt)) return r;
            if (int r = h2d(ctx, sl.in[4], r_point + 64 * i0, 64 * cnt)) return r;
            if (int r = h2d(ctx, sl.in[5], hashed_to_curve_r + 64 * i0, 64 * cnt)) return r;
            return sl.out[0].ensure(cnt);
        },
        [&](HostSlot& sl, size_t cnt, plume_ctx* on) -> int {
            const int r = aggregate_device(ctx, version, mode, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), (size_t)sl.rel[cnt], sl.in[0].as<uint8_t>(), sl.in[1].as<uint8_t>(),
                                           sl.in[2].as<uint8_t>(), sl.in[3].as<uint8_t>(), sl.in[4].as<uint8_t>(), sl.in[5].as<uint8_t>(), seed, index_base + done, sl.out[0].as<uint8_t>(),
                                           done ? rec : nullptr, rec, ctx->stream);
            done += cnt;
            return r;
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int { return hash_ok ? d2h(ctx, hash_ok + i0, sl.out[0], cnt) : 0; });
    if (rc) return rc;
    HIPCHK(hipMemcpy(record, rec, PLUME_AGG_RESULT_BYTES, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int plume_aggregate_check(plume_ctx* ctx, int version, int mode, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nullifier,
                                     const uint8_t* c, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r, const uint8_t seed[32], uint8_t* hash_ok,
                                     uint8_t* result) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (int rc = agg_args_ok(version, mode, n, msgs, msg_off, seed)) return rc;
    if (!result || (n && (!pk || !nullifier || !c || !s || !r_point || !hashed_to_curve_r))) return fail(PLUME_ERR_ARG, "null array");
    if (n == 0) { memset(result, 0, PLUME_AGG_RESULT_BYTES); result[0] = result[1] = 1; return 0; }
    uint8_t drawn[32];
    seed = agg_seed(seed, drawn);                                   // one seed for all pieces and shards of the call
    if (!seed) return fail(PLUME_ERR_HIP, "getrandom failed: no seed for the aggregate check's coefficients");
    if (ctx->shards.empty()) return aggregate_host(ctx, version, mode, n, msgs, msg_off, pk, nullifier, c, s, r_point, hashed_to_curve_r, seed, 0, hash_ok, result);
    const size_t g = ctx->shards.size();
    std::vector<uint8_t> records(PLUME_AGG_RESULT_BYTES * g, 0);       // an empty shard leaves zeros: the identity, no bad item
    uint8_t* recs = records.data();
    plume_ctx* const* shards = ctx->shards.data();
    int rc = for_shards(ctx, n, [=](plume_ctx* sh, size_t lo, size_t hi) -> int {
        size_t d = 0;
        while (shards[d] != sh) d++;
        return aggregate_host(sh, version, mode, hi - lo, msgs, msg_off + lo, pk + 64 * lo, nullifier + 64 * lo, c + 32 * lo, s + 32 * lo, r_point + 64 * lo,
                              hashed_to_curve_r + 64 * lo, seed, (uint64_t)lo, hash_ok ? hash_ok + lo : nullptr, recs + PLUME_AGG_RESULT_BYTES * d);
    });
    if (rc) return rc;
    // the shards' records are added up on the first shard's GPU
    plume_ctx* sh0 = ctx->shards[0];
    HIPCHK(hipSetDevice(sh0->device));
    if (sh0->agg_record.ensure(PLUME_AGG_RESULT_BYTES) || sh0->sink.ensure(PLUME_AGG_RESULT_BYTES * g)) return PLUME_ERR_HIP;
    HIPCHK(hipMemcpyAsync(sh0->sink.p, recs, PLUME_AGG_RESULT_BYTES * g, hipMemcpyHostToDevice, sh0->stream));
    launch_agg_combine(sh0->sink.as<uint8_t>(), (uint32_t)g, sh0->agg_record.as<uint8_t>(), sh0->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(result, sh0->agg_record.p, PLUME_AGG_RESULT_BYTES, hipMemcpyDeviceToHost, sh0->stream));
    HIPCHK(hipStreamSynchronize(sh0->stream));
    return 0;
}

static int sign_host(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                     const uint8_t* pk_in, uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point, uint8_t* hashed_to_curve_r,
                     uint8_t* status, const size_t P /* bytes per output point record: 64, or 33 for SEC1 */) {
    HIPCHK(hipSetDevice(ctx->device));
    ScopedPins pins;
    if (n) {
        pins.add(ctx, msgs + msg_off[0], (size_t)(msg_off[n] - msg_off[0]));
        pins.add(ctx, sk, 32 * n); pins.add(ctx, r, 32 * n); pins.add(ctx, pk_in, 64 * n);
        pins.add(ctx, pk, P * n); pins.add(ctx, nullifier, P * n); pins.add(ctx, c, 32 * n); pins.add(ctx, s, 32 * n); pins.add(ctx, r_point, P * n); pins.add(ctx, hashed_to_curve_r, P * n);
    }
    // two lanes (round 5) under the same condition as the verifier: every array of the call page-locked
    bool sign_two_lanes = n > 0 && ctx->host_sign_lanes > 1;
    for (const void* a : {(const void*)sk, (const void*)r, (const void*)nullifier, (const void*)c, (const void*)s, (const void*)r_point, (const void*)hashed_to_curve_r, (const void*)status,
                          pk ? (const void*)pk : (const void*)sk, pk_in ? (const void*)pk_in : (const void*)sk, (n && msg_off[n] > msg_off[0]) ? (const void*)(msgs + msg_off[0]) : (const void*)sk})
        sign_two_lanes = sign_two_lanes && is_page_locked(a);
    return host_pipeline(
        ctx, n, true,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int rc = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[0], sk + 32 * i0, 32 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[1], r + 32 * i0, 32 * cnt)) return rc;
            if (pk_in) { if (int rc = h2d(ctx, sl.in[2], pk_in + 64 * i0, 64 * cnt)) return rc; }
            return sl.out[0].ensure(P * cnt) || sl.out[1].ensure(P * cnt) || sl.out[2].ensure(32 * cnt) || sl.out[3].ensure(32 * cnt) || sl.out[4].ensure(P * cnt) ||
                           sl.out[5].ensure(P * cnt) || sl.out[6].ensure(cnt)
                       ? PLUME_ERR_HIP
                       : 0;
        },
        [&](HostSlot& sl, size_t cnt, plume_ctx* on) -> int {
            if (int rc = sign_device(on, version, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), (size_t)sl.rel[cnt], sl.in[0].as<uint8_t>(), sl.in[1].as<uint8_t>(),
                                     pk_in ? sl.in[2].as<uint8_t>() : nullptr, sl.out[0].as<uint8_t>(), sl.out[1].as<uint8_t>(), sl.out[2].as<uint8_t>(),
                                     sl.out[3].as<uint8_t>(), sl.out[4].as<uint8_t>(), sl.out[5].as<uint8_t>(), sl.out[6].as<uint8_t>(), nullptr, on->stream, P == 33))
                return rc;
            // wipe the staged secrets before the slot is reused or freed
            HIPCHK(hipMemsetAsync(sl.in[0].p, 0, 32 * cnt, on->stream));
            HIPCHK(hipMemsetAsync(sl.in[1].p, 0, 32 * cnt, on->stream));
            return 0;
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (pk) { if (int rc = d2h(ctx, pk + P * i0, sl.out[0], P * cnt)) return rc; }
            if (int rc = d2h(ctx, nullifier + P * i0, sl.out[1], P * cnt)) return rc;
            if (int rc = d2h(ctx, c + 32 * i0, sl.out[2], 32 * cnt)) return rc;
            if (int rc = d2h(ctx, s + 32 * i0, sl.out[3], 32 * cnt)) return rc;
            if (int rc = d2h(ctx, r_point + P * i0, sl.out[4], P * cnt)) return rc;
            if (int rc = d2h(ctx, hashed_to_curve_r + P * i0, sl.out[5], P * cnt)) return rc;
            return d2h(ctx, status + i0, sl.out[6], cnt);
        }, sign_two_lanes);
}
static int sign_host_any(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                         const uint8_t* pk_in, uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point, uint8_t* hashed_to_curve_r,
                         uint8_t* status, const size_t P) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!sk || !r || !nullifier || !c || !s || !r_point || !hashed_to_curve_r || !status)) return fail(PLUME_ERR_ARG, "null array");
    if (n == 0) return 0;
    if (ctx->shards.empty()) return sign_host(ctx, version, n, msgs, msg_off, sk, r, pk_in, pk, nullifier, c, s, r_point, hashed_to_curve_r, status, P);
    return for_shards(ctx, n, [=](plume_ctx* sh, size_t lo, size_t hi) -> int {
        return sign_host(sh, version, hi - lo, msgs, msg_off + lo, sk + 32 * lo, r + 32 * lo, pk_in ? pk_in + 64 * lo : nullptr, pk ? pk + P * lo : nullptr, nullifier + P * lo,
                         c + 32 * lo, s + 32 * lo, r_point + P * lo, hashed_to_curve_r + P * lo, status + lo, P);
    });
}

extern "C" int plume_sign_batch(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                                const uint8_t* pk_in, uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point, uint8_t* hashed_to_curve_r,
                                uint8_t* status) {
    return sign_host_any(ctx, version, n, msgs, msg_off, sk, r, pk_in, pk, nullifier, c, s, r_point, hashed_to_curve_r, status, 64);
}
extern "C" int plume_sign_batch_sec1(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                                     const uint8_t* pk_in, uint8_t* pk33, uint8_t* nullifier33, uint8_t* c, uint8_t* s, uint8_t* r_point33,
                                     uint8_t* hashed_to_curve_r33, uint8_t* status) {
    return sign_host_any(ctx, version, n, msgs, msg_off, sk, r, pk_in, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, status, 33);
}

static int h2c_host(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, uint8_t* h_out) {
    HIPCHK(hipSetDevice(ctx->device));
    return host_pipeline(
        ctx, n, false,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int rc = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return rc;
            if (pk) { if (int rc = h2d(ctx, sl.in[0], pk + 64 * i0, 64 * cnt)) return rc; }
            return sl.out[0].ensure(64 * cnt);
        },
        [&](HostSlot& sl, size_t cnt, plume_ctx* on) -> int {
            return h2c_only_device(ctx, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), (size_t)sl.rel[cnt], pk ? sl.in[0].as<uint8_t>() : nullptr, sl.out[0].as<uint8_t>(), ctx->stream);
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int { return d2h(ctx, h_out + 64 * i0, sl.out[0], 64 * cnt); });
}
extern "C" int plume_hash_to_curve_batch(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, uint8_t* h_out) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (int rc = args_ok(1, n, msgs, msg_off)) return rc;
    if (n && !h_out) return fail(PLUME_ERR_ARG, "null array");
    if (n == 0) return 0;
    if (ctx->shards.empty()) return h2c_host(ctx, n, msgs, msg_off, pk, h_out);
    return for_shards(ctx, n, [=](plume_ctx* sh, size_t lo, size_t hi) -> int { return h2c_host(sh, hi - lo, msgs, msg_off + lo, pk ? pk + 64 * lo : nullptr, h_out + 64 * lo); });
}

static int h2c_inter_host(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, int registers, uint8_t* u, uint8_t* mapped, uint8_t* q,
                          uint8_t* h, uint8_t* hints) {
    HIPCHK(hipSetDevice(ctx->device));
    return host_pipeline(
        ctx, n, true,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int rc = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return rc;
            if (pk) { if (int rc = h2d(ctx, sl.in[0], pk + 64 * i0, 64 * cnt)) return rc; }
            return sl.out[0].ensure(64 * cnt) || sl.out[1].ensure(128 * cnt) || sl.out[2].ensure(128 * cnt) || sl.out[3].ensure(64 * cnt) || sl.out[4].ensure(192 * cnt) ? PLUME_ERR_HIP : 0;
        },
        [&](HostSlot& sl, size_t cnt, plume_ctx* on) -> int {
            return h2c_inter_device(ctx, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), (size_t)sl.rel[cnt], pk ? sl.in[0].as<uint8_t>() : nullptr, registers,
                                    u ? sl.out[0].as<uint8_t>() : nullptr, mapped ? sl.out[1].as<uint8_t>() : nullptr, q ? sl.out[2].as<uint8_t>() : nullptr,
                                    h ? sl.out[3].as<uint8_t>() : nullptr, hints ? sl.out[4].as<uint8_t>() : nullptr, ctx->stream);
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (u) { if (int rc = d2h(ctx, u + 64 * i0, sl.out[0], 64 * cnt)) return rc; }
            if (mapped) { if (int rc = d2h(ctx, mapped + 128 * i0, sl.out[1], 128 * cnt)) return rc; }
            if (q) { if (int rc = d2h(ctx, q + 128 * i0, sl.out[2], 128 * cnt)) return rc; }
            if (h) { if (int rc = d2h(ctx, h + 64 * i0, sl.out[3], 64 * cnt)) return rc; }
            if (hints) { if (int rc = d2h(ctx, hints + 192 * i0, sl.out[4], 192 * cnt)) return rc; }
            return 0;
        });
}
static int h2c_inter_any(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, int registers, uint8_t* u, uint8_t* mapped, uint8_t* q, uint8_t* h,
                         uint8_t* hints) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (int rc = args_ok(1, n, msgs, msg_off)) return rc;
    if (registers != 0 && registers != 1) return fail(PLUME_ERR_ARG, "registers must be 0 or 1");
    if (n == 0) return 0;
    if (ctx->shards.empty()) return h2c_inter_host(ctx, n, msgs, msg_off, pk, registers, u, mapped, q, h, hints);
    return for_shards(ctx, n, [=](plume_ctx* sh, size_t lo, size_t hi) -> int {
        return h2c_inter_host(sh, hi - lo, msgs, msg_off + lo, pk ? pk + 64 * lo : nullptr, registers, u ? u + 64 * lo : nullptr, mapped ? mapped + 128 * lo : nullptr,
                              q ? q + 128 * lo : nullptr, h ? h + 64 * lo : nullptr, hints ? hints + 192 * lo : nullptr);
    });
}
extern "C" int plume_h2c_intermediates_batch(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, int registers, uint8_t* u,
                                             uint8_t* mapped, uint8_t* q, uint8_t* h) {
    return h2c_inter_any(ctx, n, msgs, msg_off, pk, registers, u, mapped, q, h, nullptr);
}
extern "C" int plume_h2c_hints_batch(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, int registers, uint8_t* hints) {
    if (n && !hints) return fail(PLUME_ERR_ARG, "null array");
    return h2c_inter_any(ctx, n, msgs, msg_off, pk, registers, nullptr, nullptr, nullptr, nullptr, hints);
}
static int der_host(plume_ctx* ctx, size_t n, const uint8_t* scalars, uint8_t* der109, uint8_t* status) {
    HIPCHK(hipSetDevice(ctx->device));
    HostSlot& sl = ctx->slot[0];
    hipStream_t st = ctx->stream;
    for (size_t i0 = 0; i0 < n; i0 += ctx->chunk) {
        const size_t cnt = n - i0 < ctx->chunk ? n - i0 : ctx->chunk;
        if (sl.in[0].ensure(32 * cnt) || sl.out[0].ensure(PLUME_DER_LEN * cnt) || sl.out[1].ensure(cnt)) return PLUME_ERR_HIP;
        HIPCHK(hipMemcpyAsync(sl.in[0].p, scalars + 32 * i0, 32 * cnt, hipMemcpyHostToDevice, st));
        if (int rc = der_device(ctx, cnt, sl.in[0].as<uint8_t>(), sl.out[0].as<uint8_t>(), sl.out[1].as<uint8_t>(), st)) return rc;
        HIPCHK(hipMemsetAsync(sl.in[0].p, 0, 32 * cnt, st));                       // the scalars may be secret keys: wipe the staged copy
        HIPCHK(hipMemcpyAsync(der109 + PLUME_DER_LEN * i0, sl.out[0].p, PLUME_DER_LEN * cnt, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(status + i0, sl.out[1].p, cnt, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemsetAsync(sl.out[0].p, 0, PLUME_DER_LEN * cnt, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return 0;
}
extern "C" int plume_scalars_to_sec1_der_batch(plume_ctx* ctx, size_t n, const uint8_t* scalars, uint8_t* der109, uint8_t* status) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (n && (!scalars || !der109 || !status)) return fail(PLUME_ERR_ARG, "null array");
    if (n == 0) return 0;
    if (ctx->shards.empty()) return der_host(ctx, n, scalars, der109, status);
    return for_shards(ctx, n, [=](plume_ctx* sh, size_t lo, size_t hi) -> int { return der_host(sh, hi - lo, scalars + 32 * lo, der109 + PLUME_DER_LEN * lo, status + lo); });
}
// The STRUCTURE half of SecretKey::from_sec1_der for the fixed 109-byte form above (what the wasm layer emits): ok[i] = 1 iff the record has that exact shape and
// its scalar is in [1, n-1].  The embedded public key is NOT compared with scalar * G here -- the reference does compare it and returns Err on a mismatch:
// plume_sec1_der_to_scalars_checked below is the function with the reference's semantics.  Host memory; no context needed.
extern "C" int plume_sec1_der_to_scalars(size_t n, const uint8_t* der109, uint8_t* scalars, uint8_t* ok) {
    if (n && (!der109 || !scalars || !ok)) return fail(PLUME_ERR_ARG, "null array");
    static_assert(plume_host::kDerLen == PLUME_DER_LEN, "record length");
    plume_host::sec1_der_to_scalars(n, der109, scalars, ok);
    return 0;
}
// SecretKey::from_sec1_der as the reference performs it (elliptic-curve's TryFrom<EcPrivateKey>: the embedded public key must be scalar * G, or the result is Err):
// the structure check above, then the records are rebuilt from the scalars on the GPU (plume_scalars_to_sec1_der_batch: one comb multiplication each) and compared
// byte for byte -- a record whose public key is not its scalar's gets ok = 0 and a zeroed scalar
extern "C" int plume_sec1_der_to_scalars_checked(plume_ctx* ctx, size_t n, const uint8_t* der109, uint8_t* scalars, uint8_t* ok) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    if (int rc = plume_sec1_der_to_scalars(n, der109, scalars, ok)) return rc;
    if (n == 0) return 0;
    std::vector<uint8_t> again(PLUME_DER_LEN * n), st(n);
    const int rc = plume_scalars_to_sec1_der_batch(ctx, n, scalars, again.data(), st.data());
    if (rc == 0)
        for (size_t i = 0; i < n; i++)
            if (ok[i] && (st[i] != 0 || std::memcmp(again.data() + PLUME_DER_LEN * i, der109 + PLUME_DER_LEN * i, PLUME_DER_LEN) != 0)) { ok[i] = 0; std::memset(scalars + 32 * i, 0, 32); }
    volatile uint8_t* w = again.data();                                  // the records hold secret scalars: wipe the host copy
    for (size_t k = 0; k < again.size(); k++) w[k] = 0;
    return rc;
}
// host form of the register packing: a byte reversal, done on the host (no reason to cross PCIe for it)
extern "C" int plume_registers_from_be(size_t nvalues, const uint8_t* be32, uint64_t* registers) {
    if (nvalues && (!be32 || !registers)) return fail(PLUME_ERR_ARG, "null array");
    plume_host::registers_from_be(nvalues, be32, registers);
    return 0;
}

// ------------------------------------------------------------------------------------------------ measurement
extern "C" int plume_last_stage_times(plume_ctx* ctx, const char** names, float* ms, int cap) {
    if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];   // a multi-device context reports its first shard
    if (ctx && ctx->lane_last) ctx = ctx->lane_last;           // ... a context with batches in flight the lane of its last device-resident call
    if (int rc = bind(ctx)) return rc;
    StageTimer& t = ctx->timer;
    if (!t.on) return fail(PLUME_ERR_ARG, "stage timing is off: plume_set_stage_timing(ctx, 1) (or env PLUME_STAGE_TIMES=1) before the call to be timed");
    const int ns = (int)t.names.size();
    if (t.used < (size_t)ns + 1) return fail(PLUME_ERR_ARG, "no timed call recorded");
    for (int i = 0; i < ns && i < cap; i++) {
        HIPCHK(hipEventSynchronize(t.ev[i + 1]));
        float v = 0;
        HIPCHK(hipEventElapsedTime(&v, t.ev[i], t.ev[i + 1]));
        if (names) names[i] = t.names[i];
        if (ms) ms[i] = v;
    }
    return ns;
}

// measurement / test hook: how many multi-scalar tasks of the last verify call on this context met p == +-q in an unchecked addition and were redone by the second launch
// (0 for honest batches; crafted items -- pk = +-k G with small k and s = +-c -- file one or two tasks each).  Synchronises with the device.
// The shader clock the multi-scalar kernel of the last verify call ran at, in GHz: one workgroup in 32 adds the cycles and the constant-rate wall-clock ticks it lived for to
// two counters; their ratio times the wall clock's rate is the clock.  Needs stage timing on before the call (PLUME_ERR_ARG otherwise).  Synchronises with the device.
extern "C" int plume_last_msm_clock(plume_ctx* ctx, double* ghz) {
    if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
    if (ctx && ctx->lane_last) ctx = ctx->lane_last;
    if (int rc = bind(ctx)) return rc;
    if (!ghz) return fail(PLUME_ERR_ARG, "null argument");
    if (!ctx->last_msm_sampled || !ctx->clk.p || !ctx->last_msm_kernel) return fail(PLUME_ERR_ARG, "plume_last_msm_clock: no verify call with stage timing on (plume_set_stage_timing) has run on this context");
    HIPCHK(hipDeviceSynchronize());
    unsigned long long c[2] = {0, 0};
    HIPCHK(hipMemcpy(c, ctx->clk.p, 16, hipMemcpyDeviceToHost));
    int khz = 0;
    HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (c[1] == 0 || khz <= 0) return fail(PLUME_ERR_ARG, "plume_last_msm_clock: the last call sampled nothing");
    *ghz = (double)c[0] / (double)c[1] * (double)khz * 1e-6;
    return 0;
}
extern "C" int plume_last_redo_tasks(plume_ctx* ctx, uint64_t* count) {
    if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
    if (ctx && ctx->lane_last) ctx = ctx->lane_last;
    if (int rc = bind(ctx)) return rc;
    if (!count) return fail(PLUME_ERR_ARG, "null argument");
    HIPCHK(hipDeviceSynchronize());
    uint64_t total = 0;
    for (size_t off : ctx->redo_counters) {
        uint32_t c = 0;
        HIPCHK(hipMemcpy(&c, ctx->redo.as<uint32_t>() + off, 4, hipMemcpyDeviceToHost));
        total += c;
    }
    *count = total;
    return 0;
}

static thread_local uint64_t g_microbench_cycles = 0;
static thread_local float g_microbench_ms = 0;
// s_memtime ticks spent by wave 0 in the last microbenchmark kernel, and that kernel's duration in ms
extern "C" double plume_microbench_last_ticks(float* ms) { if (ms) *ms = g_microbench_ms; return (double)g_microbench_cycles; }

extern "C" double plume_microbench(plume_ctx* ctx, int kind, int iters) {
    if (ctx && !ctx->shards.empty()) ctx = ctx->shards[0];
    if (bind(ctx)) return -1.0;
    if (iters <= 0 || kind < 0 || kind > 9) { fail(PLUME_ERR_ARG, "plume_microbench: bad argument"); return -1.0; }
    if (ctx->sink.ensure(128 * 4)) return -1.0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return -1.0;
    const int blocks = prop.multiProcessorCount * 8;  // 32 waves per CU
    if (kind == 9) {
        // the table-gather probe runs over the context's window-table buffer, which must dwarf the caches (a 2^20-item verify leaves 3 GiB there)
        if (ctx->tab.cap < ((size_t)1 << 30)) { fail(PLUME_ERR_ARG, "plume_microbench(9): run a verify of >= 2^19 items on this context first (the probe gathers from its window-table buffer)"); return -1.0; }
        const uint32_t nrows = (uint32_t)std::min<size_t>(ctx->tab.cap / 128, 0xFFFFFFF0u);
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
        (void)hipEventRecord(e0, ctx->stream);
        launch_gather_probe(ctx->tab.as<uint32_t>(), nrows, iters, ctx->sink.as<uint32_t>(), blocks, ctx->stream);
        (void)hipEventRecord(e1, ctx->stream);
        if (hipEventSynchronize(e1) != hipSuccess) return -1.0;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        g_microbench_ms = ms;
        return (double)iters * (double)blocks * kBlock / ((double)ms * 1e-3);   // gathers (of 80 bytes from one 128-byte row) per second
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    launch_microbench(kind, 16, ctx->sink.as<uint32_t>(), blocks, ctx->stream);  // warm-up
    (void)hipEventRecord(e0, ctx->stream);
    launch_microbench(kind, iters, ctx->sink.as<uint32_t>(), blocks, ctx->stream);
    (void)hipEventRecord(e1, ctx->stream);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.0;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double per_lane = (kind == 5 || kind == 6) ? 2.0 * iters : 8.0 * iters;   // ops per lane
    uint32_t cyc[2] = {0, 0};
    if (hipMemcpy(cyc, ctx->sink.as<uint32_t>() + 64, 8, hipMemcpyDeviceToHost) == hipSuccess)
        g_microbench_cycles = ((uint64_t)cyc[1] << 32) | cyc[0];
    g_microbench_ms = ms;
    return per_lane * (double)blocks * kBlock / ((double)ms * 1e-3);
}
