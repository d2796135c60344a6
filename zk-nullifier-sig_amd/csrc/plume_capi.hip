// C ABI of libplume_hip.so (include/plume_hip.h): context, HBM workspace, chunked pipelines, stage timing.
// Host code only; every computation on the data path is a gfx950 kernel from plume_kernels.hip.  There is no CPU
// fallback of any kind in this library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/plume_hip.h"
#include "plume_launch.h"

using namespace plume;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e__ = (expr);                                                                                  \
        if (e__ != hipSuccess) return fail(PLUME_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

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

struct StageTimer {
    std::vector<const char*> names;
    std::vector<hipEvent_t> ev;   // ev[0] start, ev[i+1] after stage i
    size_t used = 0;
    void begin(hipStream_t st) { names.clear(); used = 0; mark(st); }
    void mark(hipStream_t st) {
        if (used == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev.push_back(e); }
        (void)hipEventRecord(ev[used++], st);
    }
    void stage(const char* name, hipStream_t st) { names.push_back(name); mark(st); }
    void destroy() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); }
};

// staging for one piece of a host-pointer call
struct HostSlot {
    DevBuf msgs, off, in[6], out[7];
    std::vector<uint64_t> rel;                                        // piece-relative message offsets (source of an upload: lives until the slot is reused)
    hipEvent_t ready = nullptr, computed = nullptr, drained = nullptr;   // uploads landed / kernels finished / downloads landed
    bool in_flight = false;
};

struct plume_ctx {
    int device = 0;
    hipStream_t stream = nullptr, up = nullptr, down = nullptr;   // kernels / host->HBM / HBM->host
    size_t chunk = (size_t)1 << 20;
    size_t host_piece = (size_t)1 << 18;                            // host-pointer calls: items per pipelined piece
    HostSlot slot[2];
    int jobs_per_lane = kTableJobsPerLane;
    bool jobs_per_lane_forced = false;
    DevBuf gcomb, gtab, bases, jobflags, itemflags, tab, tabscr, res, resinf, res2, res2inf, pkaff, sink;
    DevBuf dec[4], preflags;
    DevBuf dslots, dminid, dmyslot, dcount, dblockcnt;   // nullifier-set post-processing (plume_dedup.h)   // SEC1 ingest: decompressed 64-byte records + per-item reject flags
    StageTimer timer;
};

static int bind(plume_ctx* ctx) {
    if (!ctx) return fail(PLUME_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(ctx->device));
    return 0;
}

extern "C" const char* plume_last_error(void) { return g_err.c_str(); }
extern "C" const char* plume_version(void) { return "plume_hip 0.1 gfx950"; }

extern "C" int plume_init(plume_ctx** out, int device_id) {
    if (!out || device_id < 0) return fail(PLUME_ERR_ARG, "plume_init: bad argument");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(PLUME_ERR_NODEV, "no HIP device visible (this library has no CPU fallback)");
    if (device_id >= ndev) return fail(PLUME_ERR_ARG, "device id out of range");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PLUME_ERR_NODEV, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    plume_ctx* ctx = new plume_ctx();
    ctx->device = device_id;
    if (const char* e = std::getenv("PLUME_JOBS_PER_LANE")) { int v = std::atoi(e); if (v >= 1 && v <= 64) { ctx->jobs_per_lane = v; ctx->jobs_per_lane_forced = true; } }   // tuning knob
    HIPCHK(hipSetDevice(device_id));
    if (const char* e = std::getenv("PLUME_HOST_PIECE")) { long v = std::atol(e); if (v >= 1) ctx->host_piece = (size_t)v; }   // tuning knob
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->up, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->down, hipStreamNonBlocking));
    for (HostSlot& sl : ctx->slot) {
        HIPCHK(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.drained, hipEventDisableTiming));
    }
    // generator wide window table (1..128)*G: one lane, once
    if (ctx->gtab.ensure((size_t)PLUME_GTAB_WORDS * 4) || ctx->gcomb.ensure((size_t)PLUME_COMB_WORDS * 4) || ctx->bases.ensure(PLUME_JAC_WORDS * 4 * PLUME_COMB_WINDOWS) || ctx->jobflags.ensure(64) ||
        ctx->tabscr.ensure((size_t)(PLUME_COMB_WINDOWS * PLUME_COMB_ENTRIES > PLUME_GTAB_ENTRIES ? PLUME_COMB_WINDOWS * PLUME_COMB_ENTRIES : PLUME_GTAB_ENTRIES) * PLUME_TAB_SCR_WORDS * 4)) { delete ctx; return PLUME_ERR_HIP; }
    uint32_t hb[PLUME_JAC_WORDS];
    {
        jac g; g.x = fe_gx(); g.y = fe_gy(); g.z = fe_small(1); g.inf = 0;
        st_jac_soa(hb, 1, 0, g);
    }
    uint8_t flag = PLUME_JOB_OK | PLUME_JOB_AFFINE;
    HIPCHK(hipMemcpyAsync(ctx->bases.p, hb, sizeof hb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->jobflags.p, &flag, 1, hipMemcpyHostToDevice, ctx->stream));
    launch_gtab(ctx->gtab.as<uint32_t>(), ctx->bases.as<uint32_t>(), ctx->jobflags.as<uint8_t>(), ctx->tabscr.as<uint32_t>(), ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    launch_gcomb(ctx->gcomb.as<uint32_t>(), ctx->bases.as<uint32_t>(), ctx->jobflags.as<uint8_t>(), ctx->tabscr.as<uint32_t>(), ctx->stream);   // fixed-base comb for the signer
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out = ctx;
    return 0;
}

extern "C" void plume_destroy(plume_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->up) (void)hipStreamSynchronize(ctx->up);
    if (ctx->down) (void)hipStreamSynchronize(ctx->down);
    for (DevBuf* b : {&ctx->gcomb, &ctx->gtab, &ctx->bases, &ctx->jobflags, &ctx->itemflags, &ctx->tab, &ctx->tabscr, &ctx->res, &ctx->resinf, &ctx->res2, &ctx->res2inf, &ctx->pkaff,
                      &ctx->sink, &ctx->dec[0], &ctx->dec[1], &ctx->dec[2], &ctx->dec[3], &ctx->preflags, &ctx->dslots, &ctx->dminid, &ctx->dmyslot, &ctx->dcount, &ctx->dblockcnt})
        b->release();
    for (HostSlot& sl : ctx->slot) {
        sl.msgs.release(); sl.off.release();
        for (DevBuf& b : sl.in) b.release();
        for (DevBuf& b : sl.out) b.release();
        for (hipEvent_t e : {sl.ready, sl.computed, sl.drained}) if (e) (void)hipEventDestroy(e);
    }
    ctx->timer.destroy();
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->up) (void)hipStreamDestroy(ctx->up);
    if (ctx->down) (void)hipStreamDestroy(ctx->down);
    delete ctx;
}

extern "C" int plume_set_chunk(plume_ctx* ctx, size_t max_items) {
    if (!ctx || max_items == 0 || max_items > ((size_t)1 << 26)) return fail(PLUME_ERR_ARG, "plume_set_chunk: bad argument");
    ctx->chunk = max_items;
    return 0;
}

extern "C" int plume_set_host_piece(plume_ctx* ctx, size_t items) {
    if (!ctx || items == 0 || items > ((size_t)1 << 26)) return fail(PLUME_ERR_ARG, "plume_set_host_piece: bad argument");
    ctx->host_piece = items;
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

// ------------------------------------------------------------------------------------------ device pipelines
static int verify_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nul,
                         const uint8_t* c, const uint8_t* s, const uint8_t* rpt, const uint8_t* hr, uint8_t* ok, hipStream_t st,
                         const uint8_t* preflags = nullptr, bool continue_timer = false, const uint8_t* rpt33 = nullptr, const uint8_t* hr33 = nullptr) {
    if (n == 0) return 0;
    if (n > ctx->chunk) return fail(PLUME_ERR_ARG, "n exceeds the chunk size (plume_set_chunk)");
    const int jpl = pick_jobs_per_lane(ctx, 3 * n, true);
    if (ctx->bases.ensure((size_t)PLUME_JAC_WORDS * 4 * 3 * n) || ctx->jobflags.ensure(3 * n) || ctx->itemflags.ensure(n) || ctx->tab.ensure((size_t)PLUME_TAB_WORDS * 4 * 3 * n) ||
        ctx->tabscr.ensure(tables_scratch_bytes(3 * n, jpl)) ||
        ctx->res.ensure((size_t)PLUME_JAC_WORDS * 4 * 2 * n) || ctx->resinf.ensure(2 * n))
        return PLUME_ERR_HIP;
    VerifyArgs a;
    a.version = version; a.n = (uint32_t)n; a.msgs = msgs; a.msg_off = msg_off; a.pk = pk; a.nul = nul; a.c = c; a.s = s; a.rpt = rpt; a.hr = hr; a.ok = ok; a.preflags = preflags; a.rpt33 = rpt33; a.hr33 = hr33;
    a.bases = ctx->bases.as<uint32_t>(); a.jobflags = ctx->jobflags.as<uint8_t>(); a.itemflags = ctx->itemflags.as<uint8_t>();
    a.tab = ctx->tab.as<uint32_t>(); a.res = ctx->res.as<uint32_t>(); a.resinf = ctx->resinf.as<uint8_t>(); a.gtab = ctx->gtab.as<uint32_t>();
    StageTimer& t = ctx->timer;
    if (!continue_timer) t.begin(st);
    launch_verify_ingest(a, st); t.stage("verify_ingest_h2c", st);
    launch_tables(a.tab, a.bases, a.jobflags, 3 * n, jpl, ctx->tabscr.as<uint32_t>(), st); t.stage("tables", st);
    launch_verify_msm(a, st); t.stage("verify_msm", st);
    if (version == 2) { launch_normalize(a.res, a.resinf, 2 * n, st); t.stage("to_affine", st); }   // V2 hashes the computed R', Hr'
    launch_verify_finalize(a, st); t.stage("verify_finalize", st);
    HIPCHK(hipGetLastError());
    return 0;
}

static int sign_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                       const uint8_t* pk_in, uint8_t* pk, uint8_t* nul, uint8_t* c, uint8_t* s, uint8_t* rpt, uint8_t* hr, uint8_t* status, uint8_t* h_out,
                       hipStream_t st, bool out33 = false) {
    if (n == 0) return 0;
    if (n > ctx->chunk) return fail(PLUME_ERR_ARG, "n exceeds the chunk size (plume_set_chunk)");
    const int jpl = pick_jobs_per_lane(ctx, n, false);
    if (ctx->bases.ensure((size_t)PLUME_JAC_WORDS * 4 * n) || ctx->jobflags.ensure(n) || ctx->itemflags.ensure(n) || ctx->tab.ensure((size_t)PLUME_TAB_WORDS * 4 * n) ||
        ctx->tabscr.ensure(tables_scratch_bytes(n, jpl)) ||
        ctx->res.ensure((size_t)PLUME_JAC_WORDS * 4 * 2 * n) || ctx->resinf.ensure(2 * n) || ctx->res2.ensure((size_t)PLUME_JAC_WORDS * 4 * 2 * n) || ctx->res2inf.ensure(2 * n) || ctx->pkaff.ensure((size_t)2 * PLUME_FE_WORDS * 4 * n))
        return PLUME_ERR_HIP;
    SignArgs a;
    a.version = version; a.n = (uint32_t)n; a.msgs = msgs; a.msg_off = msg_off; a.sk = sk; a.r = r; a.pk_in = pk_in;
    a.pk = pk; a.nul = nul; a.c = c; a.s = s; a.rpt = rpt; a.hr = hr; a.status = status; a.h_out = h_out; a.out33 = out33 ? 1 : 0;
    a.gres = ctx->res.as<uint32_t>(); a.gresinf = ctx->resinf.as<uint8_t>(); a.bases = ctx->bases.as<uint32_t>(); a.jobflags = ctx->jobflags.as<uint8_t>();
    a.itemflags = ctx->itemflags.as<uint8_t>(); a.pkaff = ctx->pkaff.as<uint32_t>(); a.tab = ctx->tab.as<uint32_t>();
    a.hres = ctx->res2.as<uint32_t>(); a.hresinf = ctx->res2inf.as<uint8_t>(); a.gtab = ctx->gtab.as<uint32_t>(); a.gcomb = ctx->gcomb.as<uint32_t>();
    StageTimer& t = ctx->timer;
    t.begin(st);
    launch_sign_gmul(a, st); t.stage("sign_gmul", st);
    launch_normalize(a.gres, a.gresinf, 2 * n, st); t.stage("to_affine_g", st);
    launch_sign_h2c(a, st); t.stage("sign_h2c", st);
    launch_tables(a.tab, a.bases, a.jobflags, n, jpl, ctx->tabscr.as<uint32_t>(), st); t.stage("tables", st);
    launch_sign_hmul(a, st); t.stage("sign_hmul", st);
    launch_normalize(a.hres, a.hresinf, 2 * n, st); t.stage("to_affine_h", st);
    launch_sign_final(a, st); t.stage("sign_final", st);
    // the reference zeroizes secrets: wipe the device-side images derived from sk / r
    HIPCHK(hipMemsetAsync(ctx->res.p, 0, (size_t)PLUME_JAC_WORDS * 4 * 2 * n, st));
    HIPCHK(hipGetLastError());
    return 0;
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
    (void)msgs_bytes;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk || !nullifier || !c || !s || !ok)) return fail(PLUME_ERR_ARG, "null array");
    if (n && version == 1 && (!r_point || !hashed_to_curve_r)) return fail(PLUME_ERR_ARG, "V1 needs r_point and hashed_to_curve_r");
    return verify_device(ctx, version, n, msgs, msg_off, pk, nullifier, c, s, version == 1 ? r_point : nullptr, version == 1 ? hashed_to_curve_r : nullptr, ok,
                         stream ? (hipStream_t)stream : ctx->stream);
}

// SEC1-compressed ingest: decompress on the GPU into the context's 64-byte staging arrays, then the normal pipeline
static int verify_sec1_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk33, const uint8_t* nul33,
                              const uint8_t* c, const uint8_t* s, const uint8_t* r33, const uint8_t* hr33, uint8_t* ok, hipStream_t st) {
    if (n == 0) return 0;
    if (n > ctx->chunk) return fail(PLUME_ERR_ARG, "n exceeds the chunk size (plume_set_chunk)");
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
    return verify_device(ctx, version, n, msgs, msg_off, d.out[0], d.out[1], c, s, nullptr, nullptr, ok, st, d.preflags, true, version == 1 ? r33 : nullptr,
                         version == 1 ? hr33 : nullptr);
}

extern "C" int plume_verify_batch_sec1_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                              const uint8_t* pk33, const uint8_t* nullifier33, const uint8_t* c, const uint8_t* s, const uint8_t* r_point33,
                                              const uint8_t* hashed_to_curve_r33, uint8_t* ok, void* stream) {
    (void)msgs_bytes;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk33 || !nullifier33 || !c || !s || !ok)) return fail(PLUME_ERR_ARG, "null array");
    if (n && version == 1 && (!r_point33 || !hashed_to_curve_r33)) return fail(PLUME_ERR_ARG, "V1 needs r_point and hashed_to_curve_r");
    return verify_sec1_device(ctx, version, n, msgs, msg_off, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, ok, stream ? (hipStream_t)stream : ctx->stream);
}

extern "C" int plume_sign_batch_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* sk,
                                       const uint8_t* r, const uint8_t* pk_in, uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point,
                                       uint8_t* hashed_to_curve_r, uint8_t* status, void* stream) {
    (void)msgs_bytes;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!sk || !r || !nullifier || !c || !s || !r_point || !hashed_to_curve_r || !status)) return fail(PLUME_ERR_ARG, "null array");
    return sign_device(ctx, version, n, msgs, msg_off, sk, r, pk_in, pk, nullifier, c, s, r_point, hashed_to_curve_r, status, nullptr,
                       stream ? (hipStream_t)stream : ctx->stream);
}

extern "C" int plume_sign_batch_sec1_device(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* sk,
                                            const uint8_t* r, const uint8_t* pk_in, uint8_t* pk33, uint8_t* nullifier33, uint8_t* c, uint8_t* s, uint8_t* r_point33,
                                            uint8_t* hashed_to_curve_r33, uint8_t* status, void* stream) {
    (void)msgs_bytes;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!sk || !r || !nullifier33 || !c || !s || !r_point33 || !hashed_to_curve_r33 || !status)) return fail(PLUME_ERR_ARG, "null array");
    return sign_device(ctx, version, n, msgs, msg_off, sk, r, pk_in, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, status, nullptr,
                       stream ? (hipStream_t)stream : ctx->stream, true);
}

extern "C" int plume_hash_to_curve_batch_device(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk,
                                                uint8_t* h_out, void* stream) {
    (void)msgs_bytes;
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(1, n, msgs, msg_off)) return rc;
    if (n && !h_out) return fail(PLUME_ERR_ARG, "null array");
    if (n == 0) return 0;
    H2cArgs a; a.n = (uint32_t)n; a.msgs = msgs; a.msg_off = msg_off; a.pk = pk; a.h_out = h_out;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ctx->timer.begin(st);
    launch_h2c_only(a, st); ctx->timer.stage("h2c_only", st);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------ nullifier-set post-processing
static int dedup_device(plume_ctx* ctx, size_t n, const uint8_t* nul, const uint8_t* live, const uint64_t* ids, uint8_t* first, uint64_t* n_unique_dev, hipStream_t st) {
    if (n == 0) { if (n_unique_dev) HIPCHK(hipMemsetAsync(n_unique_dev, 0, 8, st)); return 0; }
    if (n > ((size_t)1 << 30)) return fail(PLUME_ERR_ARG, "n too large");
    DedupArgs a;
    a.n = (uint32_t)n; a.nul = nul; a.live = live; a.ids = ids; a.first = first;
    const uint32_t m = dedup_table_size(a.n);
    a.mask = m - 1;
    if (ctx->dslots.ensure((size_t)m * 4) || ctx->dminid.ensure((size_t)m * 8) || ctx->dmyslot.ensure(n * 4) || ctx->dcount.ensure(8) || ctx->dblockcnt.ensure(dedup_blockcnt_bytes(n))) return PLUME_ERR_HIP;
    a.slots = ctx->dslots.as<uint32_t>(); a.minid = ctx->dminid.as<unsigned long long>(); a.myslot = ctx->dmyslot.as<uint32_t>();
    a.n_unique = ctx->dcount.as<unsigned long long>(); a.blockcnt = ctx->dblockcnt.as<uint32_t>();
    ctx->timer.begin(st);
    launch_dedup(a, st); ctx->timer.stage("nullifier_first_occurrence", st);
    HIPCHK(hipGetLastError());
    if (n_unique_dev) HIPCHK(hipMemcpyAsync(n_unique_dev, ctx->dcount.p, 8, hipMemcpyDeviceToDevice, st));
    return 0;
}
extern "C" int plume_nullifier_first_occurrence_device(plume_ctx* ctx, size_t n, const uint8_t* nullifier, const uint8_t* live, const uint64_t* ids, uint8_t* first,
                                                       uint64_t* n_unique, void* stream) {
    if (int rc = bind(ctx)) return rc;
    if (n && (!nullifier || !first)) return fail(PLUME_ERR_ARG, "null array");
    return dedup_device(ctx, n, nullifier, live, ids, first, n_unique, stream ? (hipStream_t)stream : ctx->stream);
}
// host-pointer form: one pass (every record has to be resident to be compared), staged through slot 0
extern "C" int plume_nullifier_first_occurrence(plume_ctx* ctx, size_t n, const uint8_t* nullifier, const uint8_t* live, const uint64_t* ids, uint8_t* first,
                                                uint64_t* n_unique) {
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
// The batch is cut into pieces (ctx->host_piece items, at most ctx->chunk).  Piece k+1 is staged into HBM on the upload
// stream while piece k computes on the context stream and piece k-1 drains on the download stream; two staging slots
// alternate.  With pageable caller memory the copies block the calling thread, which is why piece k-1 is drained only
// AFTER piece k has been submitted: the thread then waits on work that is already behind it in the queue.
static int stage_msgs(plume_ctx* ctx, HostSlot& sl, const uint8_t* msgs, const uint64_t* off, size_t i0, size_t cnt) {
    std::vector<uint64_t>& rel = sl.rel;
    rel.resize(cnt + 1);
    const uint64_t base = off[i0];
    for (size_t k = 0; k <= cnt; k++) {
        if (off[i0 + k] < base || (k && off[i0 + k] < off[i0 + k - 1])) return fail(PLUME_ERR_ARG, "msg_off is not non-decreasing");
        rel[k] = off[i0 + k] - base;
    }
    if (rel[cnt] > 0xFFFFFF00ull) return fail(PLUME_ERR_ARG, "message bytes per pass exceed 4 GiB");
    if (sl.msgs.ensure((size_t)rel[cnt] + 16) || sl.off.ensure((cnt + 1) * 8)) return PLUME_ERR_HIP;
    if (rel[cnt]) HIPCHK(hipMemcpyAsync(sl.msgs.p, msgs + base, (size_t)rel[cnt], hipMemcpyHostToDevice, ctx->up));
    HIPCHK(hipMemcpyAsync(sl.off.p, rel.data(), (cnt + 1) * 8, hipMemcpyHostToDevice, ctx->up));
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
    ctx->slot[0].in_flight = ctx->slot[1].in_flight = false;
}

// up(slot, i0, cnt): enqueue the uploads of one piece on ctx->up;  run(slot, cnt): enqueue its kernels on ctx->stream;
// down(slot, i0, cnt): enqueue the downloads on ctx->down.  Any error drains all three streams before it is returned, so no
// copy is left in flight on the caller's memory.
template <class Up, class Run, class Down>
static int host_pipeline(plume_ctx* ctx, size_t n, Up up, Run run, Down down) {
    const size_t piece = ctx->host_piece < ctx->chunk ? ctx->host_piece : ctx->chunk;
    struct { HostSlot* sl = nullptr; size_t i0 = 0, cnt = 0; } prev;
    auto drain = [&]() -> int {
        HIPCHK(hipStreamWaitEvent(ctx->down, prev.sl->computed, 0));
        if (int rc = down(*prev.sl, prev.i0, prev.cnt)) return rc;
        HIPCHK(hipEventRecord(prev.sl->drained, ctx->down));
        return 0;
    };
    auto body = [&]() -> int {
        size_t k = 0;
        for (size_t i0 = 0; i0 < n; i0 += piece, k++) {
            const size_t cnt = n - i0 < piece ? n - i0 : piece;
            HostSlot& sl = ctx->slot[k & 1];
            if (sl.in_flight) { HIPCHK(hipEventSynchronize(sl.drained)); sl.in_flight = false; }   // piece k-2 has left this slot
            if (int rc = up(sl, i0, cnt)) return rc;
            HIPCHK(hipEventRecord(sl.ready, ctx->up));
            HIPCHK(hipStreamWaitEvent(ctx->stream, sl.ready, 0));
            if (int rc = run(sl, cnt)) return rc;
            HIPCHK(hipEventRecord(sl.computed, ctx->stream));
            sl.in_flight = true;
            if (prev.sl) { if (int rc = drain()) return rc; }
            prev.sl = &sl; prev.i0 = i0; prev.cnt = cnt;
        }
        if (prev.sl) { if (int rc = drain()) return rc; }
        return 0;
    };
    const int rc = body();
    quiesce(ctx);
    if (rc == 0) HIPCHK(hipGetLastError());
    return rc;
}

extern "C" int plume_verify_batch(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk,
                                  const uint8_t* nullifier, const uint8_t* c, const uint8_t* s, const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                                  uint8_t* ok) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk || !nullifier || !c || !s || !ok)) return fail(PLUME_ERR_ARG, "null array");
    if (n && version == 1 && (!r_point || !hashed_to_curve_r)) return fail(PLUME_ERR_ARG, "V1 needs r_point and hashed_to_curve_r");
    const bool v1 = version == 1;
    return host_pipeline(
        ctx, n,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int rc = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[0], pk + 64 * i0, 64 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[1], nullifier + 64 * i0, 64 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[2], c + 32 * i0, 32 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[3], s + 32 * i0, 32 * cnt)) return rc;
            if (v1) {
                if (int rc = h2d(ctx, sl.in[4], r_point + 64 * i0, 64 * cnt)) return rc;
                if (int rc = h2d(ctx, sl.in[5], hashed_to_curve_r + 64 * i0, 64 * cnt)) return rc;
            }
            return sl.out[0].ensure(cnt);
        },
        [&](HostSlot& sl, size_t cnt) -> int {
            return verify_device(ctx, version, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), sl.in[0].as<uint8_t>(), sl.in[1].as<uint8_t>(), sl.in[2].as<uint8_t>(),
                                 sl.in[3].as<uint8_t>(), v1 ? sl.in[4].as<uint8_t>() : nullptr, v1 ? sl.in[5].as<uint8_t>() : nullptr, sl.out[0].as<uint8_t>(),
                                 ctx->stream);
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int { return d2h(ctx, ok + i0, sl.out[0], cnt); });
}

extern "C" int plume_verify_batch_sec1(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk33,
                                       const uint8_t* nullifier33, const uint8_t* c, const uint8_t* s, const uint8_t* r_point33,
                                       const uint8_t* hashed_to_curve_r33, uint8_t* ok) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!pk33 || !nullifier33 || !c || !s || !ok)) return fail(PLUME_ERR_ARG, "null array");
    if (n && version == 1 && (!r_point33 || !hashed_to_curve_r33)) return fail(PLUME_ERR_ARG, "V1 needs r_point and hashed_to_curve_r");
    const bool v1 = version == 1;
    return host_pipeline(
        ctx, n,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int rc = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[0], pk33 + 33 * i0, 33 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[1], nullifier33 + 33 * i0, 33 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[2], c + 32 * i0, 32 * cnt)) return rc;
            if (int rc = h2d(ctx, sl.in[3], s + 32 * i0, 32 * cnt)) return rc;
            if (v1) {
                if (int rc = h2d(ctx, sl.in[4], r_point33 + 33 * i0, 33 * cnt)) return rc;
                if (int rc = h2d(ctx, sl.in[5], hashed_to_curve_r33 + 33 * i0, 33 * cnt)) return rc;
            }
            return sl.out[0].ensure(cnt);
        },
        [&](HostSlot& sl, size_t cnt) -> int {
            return verify_sec1_device(ctx, version, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), sl.in[0].as<uint8_t>(), sl.in[1].as<uint8_t>(),
                                      sl.in[2].as<uint8_t>(), sl.in[3].as<uint8_t>(), v1 ? sl.in[4].as<uint8_t>() : nullptr, v1 ? sl.in[5].as<uint8_t>() : nullptr,
                                      sl.out[0].as<uint8_t>(), ctx->stream);
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int { return d2h(ctx, ok + i0, sl.out[0], cnt); });
}

static int sign_host(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                     const uint8_t* pk_in, uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point, uint8_t* hashed_to_curve_r,
                     uint8_t* status, const size_t P /* bytes per output point record: 64, or 33 for SEC1 */) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(version, n, msgs, msg_off)) return rc;
    if (n && (!sk || !r || !nullifier || !c || !s || !r_point || !hashed_to_curve_r || !status)) return fail(PLUME_ERR_ARG, "null array");
    return host_pipeline(
        ctx, n,
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
        [&](HostSlot& sl, size_t cnt) -> int {
            if (int rc = sign_device(ctx, version, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), sl.in[0].as<uint8_t>(), sl.in[1].as<uint8_t>(),
                                     pk_in ? sl.in[2].as<uint8_t>() : nullptr, sl.out[0].as<uint8_t>(), sl.out[1].as<uint8_t>(), sl.out[2].as<uint8_t>(),
                                     sl.out[3].as<uint8_t>(), sl.out[4].as<uint8_t>(), sl.out[5].as<uint8_t>(), sl.out[6].as<uint8_t>(), nullptr, ctx->stream, P == 33))
                return rc;
            // wipe the staged secrets before the slot is reused or freed
            HIPCHK(hipMemsetAsync(sl.in[0].p, 0, 32 * cnt, ctx->stream));
            HIPCHK(hipMemsetAsync(sl.in[1].p, 0, 32 * cnt, ctx->stream));
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
        });
}

extern "C" int plume_sign_batch(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                                const uint8_t* pk_in, uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point, uint8_t* hashed_to_curve_r,
                                uint8_t* status) {
    return sign_host(ctx, version, n, msgs, msg_off, sk, r, pk_in, pk, nullifier, c, s, r_point, hashed_to_curve_r, status, 64);
}
extern "C" int plume_sign_batch_sec1(plume_ctx* ctx, int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r,
                                     const uint8_t* pk_in, uint8_t* pk33, uint8_t* nullifier33, uint8_t* c, uint8_t* s, uint8_t* r_point33,
                                     uint8_t* hashed_to_curve_r33, uint8_t* status) {
    return sign_host(ctx, version, n, msgs, msg_off, sk, r, pk_in, pk33, nullifier33, c, s, r_point33, hashed_to_curve_r33, status, 33);
}

extern "C" int plume_hash_to_curve_batch(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, uint8_t* h_out) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = args_ok(1, n, msgs, msg_off)) return rc;
    if (n && !h_out) return fail(PLUME_ERR_ARG, "null array");
    return host_pipeline(
        ctx, n,
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int {
            if (int rc = stage_msgs(ctx, sl, msgs, msg_off, i0, cnt)) return rc;
            if (pk) { if (int rc = h2d(ctx, sl.in[0], pk + 64 * i0, 64 * cnt)) return rc; }
            return sl.out[0].ensure(64 * cnt);
        },
        [&](HostSlot& sl, size_t cnt) -> int {
            return plume_hash_to_curve_batch_device(ctx, cnt, sl.msgs.as<uint8_t>(), sl.off.as<uint64_t>(), (size_t)sl.rel[cnt], pk ? sl.in[0].as<uint8_t>() : nullptr,
                                                    sl.out[0].as<uint8_t>(), ctx->stream);
        },
        [&](HostSlot& sl, size_t i0, size_t cnt) -> int { return d2h(ctx, h_out + 64 * i0, sl.out[0], 64 * cnt); });
}

// ------------------------------------------------------------------------------------------------ measurement
extern "C" int plume_last_stage_times(plume_ctx* ctx, const char** names, float* ms, int cap) {
    if (int rc = bind(ctx)) return rc;
    StageTimer& t = ctx->timer;
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

static thread_local uint64_t g_microbench_cycles = 0;
static thread_local float g_microbench_ms = 0;
// s_memtime ticks spent by wave 0 in the last microbenchmark kernel, and that kernel's duration in ms
extern "C" double plume_microbench_last_ticks(float* ms) { if (ms) *ms = g_microbench_ms; return (double)g_microbench_cycles; }

extern "C" double plume_microbench(plume_ctx* ctx, int kind, int iters) {
    if (bind(ctx)) return -1.0;
    if (iters <= 0 || kind < 0 || kind > 8) { fail(PLUME_ERR_ARG, "plume_microbench: bad argument"); return -1.0; }
    if (ctx->sink.ensure(128 * 4)) return -1.0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return -1.0;
    const int blocks = prop.multiProcessorCount * 8;  // 32 waves per CU
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
