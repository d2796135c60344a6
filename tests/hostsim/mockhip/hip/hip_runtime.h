// TEST INFRASTRUCTURE ONLY.  A CPU-only stand-in for the part of the HIP runtime API that the HOST side of libplume_hip.so uses (csrc/plume_capi.hip: contexts, worker
// threads, streams, events, staging slots, the piece pipeline), so that exactly that host code can be compiled with g++ and run under AddressSanitizer, UBSan and
// ThreadSanitizer without a GPU (tests/hostsim/Makefile, tests/test_sanitizers.py).  Nothing here is part of the product; the product library links the real runtime and
// refuses to start without a gfx950 device.
//
// What is modelled, and why it is not simply "do everything at once":
//  * device memory is heap memory (handed out DIRTY, like hipMalloc's), "kernels" are the per-lane bodies in plain loops (tests/hostsim/host_launch.cpp);
//  * a stream is a QUEUE.  Asynchronous copies, memsets, kernels, event records and event waits are only queued when the library issues them; they run when somebody has to
//    wait for them (hipStreamSynchronize, hipEventSynchronize, a blocking copy, hipFree ...).  The default scheduler is LAZY: it runs only what the waited-for stream or event
//    transitively needs, in dependency order -- so everything the library did NOT order (a download that forgot to wait for its kernels, a slot reused before its upload was
//    consumed, a call that returns while a copy on the caller's arrays is still queued) happens in the worst order and shows up as a wrong result or a sanitizer report.
//    PLUME_MOCK_SCHED=random:<seed> additionally runs ready work of other streams at random points; PLUME_MOCK_SCHED=eager runs everything as early as possible.
//  * copies between a stream and PAGEABLE host memory keep the runtime's semantics: an upload's source is consumed before the call returns (snapshot), a download blocks until
//    it is done.  Page-locked memory (hipHostMalloc / hipHostRegister) is read and written when the copy RUNS, and must still be registered then.
//  * one lock per device: the shards' worker threads (one device each) run side by side, as on the real machine.
// PLUME_MOCK_DEVICES (default 8) devices report gfx950.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotSupported = 801 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };

namespace mockhip {
struct Event;
struct Stream;
struct Op {
    int kind;                       // 0 work, 1 record, 2 wait
    std::function<void()> fn;
    Event* ev;
    uint64_t ver;
};
struct Stream { int device; int prio; std::deque<Op> q; };
struct Event { int device; uint64_t recorded = 0, done = 0; };
struct Device {
    std::recursive_mutex m;
    std::vector<Stream*> streams;
    Stream* null_stream = nullptr;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    long ops_run = 0, ops_run_by_other_stream = 0;
};
struct Range { size_t bytes; hipMemoryType type; };
struct State {
    std::mutex m;                                        // the address map and the counters
    std::map<const void*, Range> ranges;                 // allocations and registrations by base address
    long device_allocs = 0, host_allocs = 0, streams = 0, events = 0;
    long alloc_calls = 0, fail_alloc_at = -1;            // fault injection: the fail_alloc_at-th allocation (hipMalloc / hipHostMalloc, counted from 0) from now on fails once
    size_t fail_big_bytes = 0; long fail_big_skip = 0;   // ... or the (skip+1)-th allocation of at least fail_big_bytes bytes fails once (the generator's tables)
    Device dev[16];
    int sched = 0;                                       // 0 lazy, 1 eager, 2 random
    State() {
        if (const char* e = std::getenv("PLUME_MOCK_SCHED")) {
            if (!std::strncmp(e, "eager", 5)) sched = 1;
            else if (!std::strncmp(e, "random", 6)) {
                sched = 2;
                const uint64_t seed = e[6] == ':' ? std::strtoull(e + 7, nullptr, 10) : 1;
                for (int d = 0; d < 16; d++) dev[d].rng = (seed + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)d * 0xD1B54A32D192ED03ull;
            }
        }
    }
};
inline State& st() { static State s; return s; }
inline int& current_device() { static thread_local int d = 0; return d; }
inline int device_count() { const char* e = std::getenv("PLUME_MOCK_DEVICES"); const int v = e ? std::atoi(e) : 8; return v > 16 ? 16 : v > 0 ? v : 0; }
[[noreturn]] inline void die(const char* what) { std::fprintf(stderr, "mock HIP runtime: %s\n", what); std::abort(); }
inline bool lookup(const void* p, Range& out, size_t need = 1) {
    State& s = st();
    std::lock_guard<std::mutex> lk(s.m);
    auto it = s.ranges.upper_bound(p);
    if (it == s.ranges.begin()) return false;
    --it;
    const size_t off = (size_t)((const char*)p - (const char*)it->first);
    if (off >= it->second.bytes || need > it->second.bytes - off) return false;
    out = it->second;
    return true;
}
// the driver's fault injection: make the k-th allocation from now fail (k < 0: none); returns how many allocations were made since the last call
inline long fail_allocation(long k) { State& s = st(); std::lock_guard<std::mutex> lk(s.m); const long made = s.alloc_calls; s.alloc_calls = 0; s.fail_alloc_at = k; return made; }
inline void fail_allocation_of_at_least(size_t bytes, long skip) { State& s = st(); std::lock_guard<std::mutex> lk(s.m); s.fail_big_bytes = bytes; s.fail_big_skip = skip; }
inline bool alloc_fails(size_t bytes) {
    State& s = st(); std::lock_guard<std::mutex> lk(s.m);
    bool f = s.fail_alloc_at >= 0 && s.alloc_calls == s.fail_alloc_at;
    s.alloc_calls++;
    if (f) s.fail_alloc_at = -1;
    if (!f && s.fail_big_bytes && bytes >= s.fail_big_bytes) { if (s.fail_big_skip-- == 0) { f = true; s.fail_big_bytes = 0; } }
    return f;
}
inline long outstanding(int what) { State& s = st(); std::lock_guard<std::mutex> lk(s.m); return what == 0 ? s.device_allocs : what == 1 ? s.host_allocs : what == 2 ? s.streams : s.events; }
inline long ops_run(int device, bool by_other) { Device& d = st().dev[device]; std::lock_guard<std::recursive_mutex> lk(d.m); return by_other ? d.ops_run_by_other_stream : d.ops_run; }

inline Stream* resolve(Stream* s) {                     // the null stream: one more independent queue per device (the library's own streams are all non-blocking)
    if (s) return s;
    Device& d = st().dev[current_device()];
    std::lock_guard<std::recursive_mutex> lk(d.m);
    if (!d.null_stream) { d.null_stream = new Stream{current_device(), 0, {}}; d.streams.push_back(d.null_stream); }
    return d.null_stream;
}
inline bool runnable(const Op& o) { return o.kind != 2 || o.ev->done >= o.ver; }
// run exactly one queued operation that brings stream s's head closer to running (the head itself, or the head of the chain of streams it waits for).  Device lock held.
inline void step(Device& d, Stream* s, Stream* on_behalf, int depth = 0) {
    if (depth > 64) die("event wait cycle (or a wait for a record that was never queued)");
    if (s->q.empty()) die("internal: step on an empty stream");
    Op& head = s->q.front();
    if (!runnable(head)) {
        Stream* src = nullptr;
        uint64_t best = ~0ull;
        for (Stream* t : d.streams)
            for (const Op& o : t->q)
                if (o.kind == 1 && o.ev == head.ev && o.ver >= head.ver && o.ver < best) { best = o.ver; src = t; }
        if (!src) die("a stream waits for an event record that is in no queue of its device");
        if (src == s) die("a stream waits for an event recorded later on the same stream");
        step(d, src, on_behalf, depth + 1);
        return;
    }
    Op o = std::move(head);
    s->q.pop_front();
    d.ops_run++;
    if (s != on_behalf) d.ops_run_by_other_stream++;
    if (o.kind == 0) o.fn();
    else if (o.kind == 1) { if (o.ev->done < o.ver) o.ev->done = o.ver; }
}
inline uint64_t next_rand(Device& d) { d.rng ^= d.rng << 13; d.rng ^= d.rng >> 7; d.rng ^= d.rng << 17; return d.rng; }
inline void random_extras(Device& d) {                   // random scheduler: sometimes a ready operation of ANY stream of the device runs first
    if (st().sched != 2) return;
    for (int k = 0; k < 3; k++) {
        if (next_rand(d) & 1u) return;
        std::vector<Stream*> ready;
        for (Stream* t : d.streams) if (!t->q.empty() && runnable(t->q.front())) ready.push_back(t);
        if (ready.empty()) return;
        Stream* t = ready[next_rand(d) % ready.size()];
        step(d, t, t);
    }
}
inline void drain_stream(Stream* s) {
    Device& d = st().dev[s->device];
    std::lock_guard<std::recursive_mutex> lk(d.m);
    while (!s->q.empty()) { random_extras(d); if (!s->q.empty()) step(d, s, s); }
}
inline void drain_event(Event* e) {
    Device& d = st().dev[e->device];
    std::lock_guard<std::recursive_mutex> lk(d.m);
    while (e->done < e->recorded) {
        Stream* src = nullptr;
        uint64_t best = ~0ull;
        for (Stream* t : d.streams)
            for (const Op& o : t->q)
                if (o.kind == 1 && o.ev == e && o.ver > e->done && o.ver < best) { best = o.ver; src = t; }
        if (!src) die("an event is waited for whose record is in no queue");
        random_extras(d);
        if (e->done < e->recorded) step(d, src, src);
    }
}
inline void drain_device(int device) {
    Device& d = st().dev[device];
    std::lock_guard<std::recursive_mutex> lk(d.m);
    for (;;) {
        Stream* busy = nullptr;
        for (Stream* t : d.streams) if (!t->q.empty()) { busy = t; break; }
        if (!busy) return;
        step(d, busy, busy);
    }
}
inline void eager_pass(Device& d) {                      // eager scheduler: after every enqueue, everything that can run does
    if (st().sched != 1) return;
    for (bool any = true; any;) {
        any = false;
        for (Stream* t : d.streams) while (!t->q.empty() && runnable(t->q.front())) { step(d, t, t); any = true; }
    }
}
inline void enqueue(Stream* s, Op&& o) {
    s = resolve(s);
    Device& d = st().dev[s->device];
    std::lock_guard<std::recursive_mutex> lk(d.m);
    s->q.push_back(std::move(o));
    eager_pass(d);
}
// a kernel launch of the host build: fn runs when the stream gets there
inline void launch(Stream* s, std::function<void()> fn) { enqueue(s, Op{0, std::move(fn), nullptr, 0}); }
}  // namespace mockhip

typedef mockhip::Stream* hipStream_t;
typedef mockhip::Event* hipEvent_t;

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (mock)" : "error (mock HIP runtime)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = mockhip::device_count(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= mockhip::device_count()) return hipErrorInvalidDevice; mockhip::current_device() = d; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d) {
    if (d < 0 || d >= mockhip::device_count()) return hipErrorInvalidDevice;
    std::memset(p, 0, sizeof *p);
    const char* arch = std::getenv("PLUME_MOCK_ARCH");                      // (a machine with another GPU: the library must refuse it)
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "%s:sramecc+:xnack-", arch ? arch : "gfx950");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
inline hipError_t hipDeviceGetPCIBusId(char* out, int len, int d) {          // an address no machine has: the NUMA lookup behind it then finds no sysfs entry
    if (d < 0 || d >= mockhip::device_count()) return hipErrorInvalidDevice;
    std::snprintf(out, (size_t)len, "FFFF:%02X:00.0", 0xE0 + d);
    return hipSuccess;
}
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }     // (kHz: the constant-rate wall clock)
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 1; *greatest = -1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { mockhip::drain_device(mockhip::current_device()); return hipSuccess; }

inline hipError_t hipMalloc(void** p, size_t bytes) {
    if (mockhip::alloc_fails(bytes)) return hipErrorOutOfMemory;
    void* q = std::malloc(bytes ? bytes : 1);
    if (!q) return hipErrorOutOfMemory;
    std::memset(q, 0xA5, bytes);                          // device memory comes back dirty
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); s.ranges[q] = {bytes ? bytes : 1, hipMemoryTypeDevice}; s.device_allocs++; }
    *p = q;
    return hipSuccess;
}
inline hipError_t hipFree(void* p) {                      // (the real call waits for the device's queued work too)
    if (!p) return hipSuccess;
    mockhip::drain_device(mockhip::current_device());
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); if (!s.ranges.erase(p)) return hipErrorInvalidValue; s.device_allocs--; }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    if (mockhip::alloc_fails(bytes)) return hipErrorOutOfMemory;
    void* q = std::malloc(bytes ? bytes : 1);
    if (!q) return hipErrorOutOfMemory;
    std::memset(q, 0x5A, bytes);
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); s.ranges[q] = {bytes ? bytes : 1, hipMemoryTypeHost}; s.host_allocs++; }
    *p = q;
    return hipSuccess;
}
inline hipError_t hipHostFree(void* p) {
    if (!p) return hipSuccess;
    mockhip::drain_device(mockhip::current_device());
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); if (!s.ranges.erase(p)) return hipErrorInvalidValue; s.host_allocs--; }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostRegister(void* p, size_t bytes, unsigned) {
    mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m);
    if (!p || !bytes || s.ranges.count(p)) return hipErrorInvalidValue;
    s.ranges[p] = {bytes, hipMemoryTypeHost};
    return hipSuccess;
}
inline hipError_t hipHostUnregister(void* p) {
    mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m);
    return s.ranges.erase(p) ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    mockhip::Range r;
    a->device = -1;
    if (!mockhip::lookup(p, r)) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }    // (ROCm 6+: pageable memory is reported, not refused)
    a->type = r.type; a->device = 0;
    return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
    if (!n) return hipSuccess;
    s = mockhip::resolve(s);
    mockhip::Range r;
    if (kind == hipMemcpyHostToDevice && !mockhip::lookup(src, r, n)) {          // pageable source: consumed before the call returns
        auto snap = std::make_shared<std::vector<unsigned char>>((const unsigned char*)src, (const unsigned char*)src + n);
        mockhip::launch(s, [dst, snap] { std::memcpy(dst, snap->data(), snap->size()); });
        return hipSuccess;
    }
    if (kind == hipMemcpyDeviceToHost && !mockhip::lookup(dst, r, n)) {          // pageable destination: the call blocks until the copy is done
        mockhip::launch(s, [dst, src, n] { std::memcpy(dst, src, n); });
        mockhip::drain_stream(s);
        return hipSuccess;
    }
    const void* host = kind == hipMemcpyHostToDevice ? src : kind == hipMemcpyDeviceToHost ? dst : nullptr;
    mockhip::launch(s, [dst, src, n, host] {
        mockhip::Range rr;
        if (host && !mockhip::lookup(host, rr, n)) mockhip::die("an asynchronous copy ran after its page-locked host range was unregistered or freed");
        std::memmove(dst, src, n);
    });
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t s) { if (n) mockhip::launch(s, [d, v, n] { std::memset(d, v, n); }); return hipSuccess; }

inline hipError_t hipStreamCreateWithPriority(hipStream_t* out, unsigned, int prio) {
    mockhip::Device& d = mockhip::st().dev[mockhip::current_device()];
    mockhip::Stream* s = new mockhip::Stream{mockhip::current_device(), prio, {}};
    { std::lock_guard<std::recursive_mutex> lk(d.m); d.streams.push_back(s); }
    { mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.streams++; }
    *out = s;
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) { return hipStreamCreateWithPriority(s, f, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) {      // (the real call lets queued work finish)
    if (!s) return hipErrorInvalidValue;
    mockhip::drain_stream(s);
    mockhip::Device& d = mockhip::st().dev[s->device];
    { std::lock_guard<std::recursive_mutex> lk(d.m); for (size_t k = 0; k < d.streams.size(); k++) if (d.streams[k] == s) { d.streams.erase(d.streams.begin() + (long)k); break; } }
    delete s;
    mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.streams--;
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s) { mockhip::drain_stream(mockhip::resolve(s)); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    *e = new mockhip::Event{mockhip::current_device()};
    { mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.events++; }
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    s = mockhip::resolve(s);
    mockhip::Device& d = mockhip::st().dev[s->device];
    std::lock_guard<std::recursive_mutex> lk(d.m);
    if (e->device != s->device) mockhip::die("event recorded on a stream of another device");
    e->recorded++;
    mockhip::enqueue(s, mockhip::Op{1, nullptr, e, e->recorded});
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    s = mockhip::resolve(s);
    mockhip::Device& d = mockhip::st().dev[s->device];
    std::lock_guard<std::recursive_mutex> lk(d.m);
    if (e->device != s->device) mockhip::die("cross-device event wait (this library has none)");
    if (e->recorded == 0 || e->done >= e->recorded) return hipSuccess;       // never recorded, or already complete: nothing to wait for
    mockhip::enqueue(s, mockhip::Op{2, nullptr, e, e->recorded});
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) { mockhip::drain_event(e); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) {        // (the real call defers the release until the record has completed)
    if (!e) return hipErrorInvalidValue;
    mockhip::drain_event(e);
    {   // a wait on this event may still sit in a queue (already satisfied): it must not read the event after it is gone
        mockhip::Device& d = mockhip::st().dev[e->device];
        std::lock_guard<std::recursive_mutex> lk(d.m);
        for (mockhip::Stream* t : d.streams)
            for (mockhip::Op& o : t->q) if (o.kind == 2 && o.ev == e) { o.kind = 0; o.fn = [] {}; o.ev = nullptr; }
    }
    delete e;
    mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.events--;
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    if (a->done < a->recorded || b->done < b->recorded) return hipErrorInvalidValue;       // (hipErrorNotReady in the real runtime)
    *ms = 0.0f;
    return hipSuccess;
}
