// TEST INFRASTRUCTURE ONLY.  A synchronous, CPU-only stand-in for the part of the HIP runtime API that the HOST side of libplume_hip.so uses (csrc/plume_capi.hip: contexts,
// worker threads, streams, events, staging slots, the piece pipeline), so that exactly that host code can be compiled with g++ and run under AddressSanitizer, UBSan and
// ThreadSanitizer without a GPU (tests/hostsim/Makefile, tests/test_sanitizers.py).  "Device memory" is heap memory, every copy / memset / "kernel" (tests/hostsim/
// host_launch.cpp: the per-lane bodies in plain loops) runs at once in the calling thread, so streams and events are trivially ordered; what stays real is everything the
// sanitizers are there for: buffer sizes and offsets, object lifetimes, the shards' threads and what they share.  PLUME_MOCK_DEVICES (default 8) devices report gfx950.
// Nothing here is part of the product; the product library links the real runtime and refuses to start without a gfx950 device.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotSupported = 801 };
struct mock_hip_stream { int device, priority; };
struct mock_hip_event { int dummy; };
typedef mock_hip_stream* hipStream_t;
typedef mock_hip_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };

namespace mockhip {
struct Range { size_t bytes; hipMemoryType type; };
struct State {
    std::mutex m;
    std::map<const void*, Range> ranges;                 // allocations and registrations by base address
    long device_allocs = 0, host_allocs = 0, streams = 0, events = 0;
};
inline State& st() { static State s; return s; }
inline int& current_device() { static thread_local int d = 0; return d; }
inline int device_count() { const char* e = std::getenv("PLUME_MOCK_DEVICES"); const int v = e ? std::atoi(e) : 8; return v > 0 ? v : 0; }
inline bool lookup(const void* p, Range& out) {
    State& s = st();
    std::lock_guard<std::mutex> lk(s.m);
    auto it = s.ranges.upper_bound(p);
    if (it == s.ranges.begin()) return false;
    --it;
    if ((const char*)p >= (const char*)it->first + it->second.bytes) return false;
    out = it->second;
    return true;
}
// the harness asks at its end whether the library gave everything back
inline long outstanding(int what) { State& s = st(); std::lock_guard<std::mutex> lk(s.m); return what == 0 ? s.device_allocs : what == 1 ? s.host_allocs : what == 2 ? s.streams : s.events; }
}  // namespace mockhip

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (mock)" : "error (mock HIP runtime)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = mockhip::device_count(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= mockhip::device_count()) return hipErrorInvalidDevice; mockhip::current_device() = d; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d) {
    if (d < 0 || d >= mockhip::device_count()) return hipErrorInvalidDevice;
    std::memset(p, 0, sizeof *p);
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
inline hipError_t hipDeviceGetPCIBusId(char*, int, int) { return hipErrorNotSupported; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 1; *greatest = -1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

inline hipError_t hipMalloc(void** p, size_t bytes) {
    void* q = std::malloc(bytes ? bytes : 1);
    if (!q) return hipErrorOutOfMemory;
    std::memset(q, 0xA5, bytes);                          // device memory comes back dirty
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); s.ranges[q] = {bytes, hipMemoryTypeDevice}; s.device_allocs++; }
    *p = q;
    return hipSuccess;
}
inline hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); if (!s.ranges.erase(p)) return hipErrorInvalidValue; s.device_allocs--; }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    void* q = std::malloc(bytes ? bytes : 1);
    if (!q) return hipErrorOutOfMemory;
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); s.ranges[q] = {bytes, hipMemoryTypeHost}; s.host_allocs++; }
    *p = q;
    return hipSuccess;
}
inline hipError_t hipHostFree(void* p) {
    if (!p) return hipSuccess;
    { mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m); if (!s.ranges.erase(p)) return hipErrorInvalidValue; s.host_allocs--; }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostRegister(void* p, size_t bytes, unsigned) {
    mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m);
    if (s.ranges.count(p)) return hipErrorInvalidValue;
    s.ranges[p] = {bytes, hipMemoryTypeHost};
    return hipSuccess;
}
inline hipError_t hipHostUnregister(void* p) {
    mockhip::State& s = mockhip::st(); std::lock_guard<std::mutex> lk(s.m);
    return s.ranges.erase(p) ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    mockhip::Range r;
    if (!mockhip::lookup(p, r)) { a->type = hipMemoryTypeUnregistered; a->device = -1; return hipErrorInvalidValue; }    // (the real runtime reports pageable memory this way too)
    a->type = r.type; a->device = 0;
    return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }

inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int prio) {
    *s = new mock_hip_stream{mockhip::current_device(), prio};
    { mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.streams++; }
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) { return hipStreamCreateWithPriority(s, f, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.streams--; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    *e = new mock_hip_event{0};
    { mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.events++; }
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; mockhip::State& g = mockhip::st(); std::lock_guard<std::mutex> lk(g.m); g.events--; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
