// d2h_probe (round 5): how do a signer piece's downloads (seven arrays, 320 B per item) get to page-locked host memory while a saturating kernel runs?
//   ./d2h_probe log            one copy of each kind, to be run with AMD_LOG_LEVEL=4 (which path does the runtime take: SDMA "HSA Copy ..." or a blit kernel?)
//   ./d2h_probe                timings: every variant alone and beside a busy kernel that holds the CUs the way k_sign_hmul does (VALU-saturating, 33 KiB LDS per workgroup)
// Variants: hipMemcpyAsync per array (idle stream / after hipStreamWaitEvent / destination hipHostMalloc'ed or hipHostRegister'ed), ONE packed copy,
// an export KERNEL (grid G x 256 lanes, 16-byte stores into the mapped host arrays) on a stream of low / normal / high priority.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_busy(uint32_t* sink, int iters) {
    extern __shared__ uint32_t lds[];
    uint64_t a0 = threadIdx.x + 1, a1 = blockIdx.x + 3, a2 = 7, a3 = 11;
    uint32_t m = threadIdx.x * 2654435761u + 12345u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a0 = (uint64_t)m * (uint32_t)a0 + a1; a1 = (uint64_t)m * (uint32_t)a1 + a2; a2 = (uint64_t)m * (uint32_t)a2 + a3; a3 = (uint64_t)m * (uint32_t)a3 + a0;
        }
    }
    lds[threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3);
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 255] == 0x12345678u) sink[0] = lds[threadIdx.x];
}

struct ExpArgs { int narr; const uint8_t* src[8]; uint8_t* dst[8]; unsigned long long bytes[8]; };
__global__ void __launch_bounds__(256) k_export(ExpArgs a) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (int k = 0; k < a.narr; k++) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4* s = (const v4*)a.src[k];
        v4* d = (v4*)a.dst[k];
        const size_t q = a.bytes[k] / 16;
        for (size_t i = tid; i < q; i += nth) { v4 v = __builtin_nontemporal_load(s + i); __builtin_nontemporal_store(v, d + i); }
        for (size_t i = q * 16 + tid; i < a.bytes[k]; i += nth) a.dst[k][i] = a.src[k][i];
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const bool logmode = argc > 1 && !strcmp(argv[1], "log");
    const size_t n = logmode ? (1 << 14) : (1 << 19);
    const size_t w[7] = {64, 64, 32, 32, 64, 64, 1};
    size_t total = 0;
    for (size_t x : w) total += x * n;
    uint8_t *dsrc[7], *hm[7], *hr[7], *dpack, *hpack;
    std::vector<void*> raw(7);
    for (int k = 0; k < 7; k++) {
        CHK(hipMalloc(&dsrc[k], w[k] * n));
        CHK(hipMemset(dsrc[k], k + 1, w[k] * n));
        CHK(hipHostMalloc(&hm[k], w[k] * n, hipHostMallocDefault));
        raw[k] = aligned_alloc(4096, (w[k] * n + 4095) & ~(size_t)4095);
        memset(raw[k], 0, w[k] * n);
        CHK(hipHostRegister(raw[k], w[k] * n, hipHostRegisterDefault));
        hr[k] = (uint8_t*)raw[k];
    }
    CHK(hipMalloc(&dpack, total));
    CHK(hipHostMalloc(&hpack, total, hipHostMallocDefault));
    uint32_t* sink;
    CHK(hipMalloc(&sink, 64));
    int lo = 0, hi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    hipStream_t comp, down, plo, phi;
    CHK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
    CHK(hipStreamCreateWithPriority(&plo, hipStreamNonBlocking, lo));
    CHK(hipStreamCreateWithPriority(&phi, hipStreamNonBlocking, hi));
    hipEvent_t ev, b0, b1;
    CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CHK(hipEventCreate(&b0));
    CHK(hipEventCreate(&b1));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int busy_blocks = prop.multiProcessorCount * 4 * 4;   // four waves of workgroups at four resident workgroups per CU
    int busy_iters = 1;
    auto busy = [&](hipStream_t s) { hipLaunchKernelGGL(k_busy, dim3(busy_blocks), dim3(256), 33 * 1024, s, sink, busy_iters); };

    auto exp_args = [&](uint8_t** dst) {
        ExpArgs a; a.narr = 7;
        for (int k = 0; k < 7; k++) { a.src[k] = dsrc[k]; void* dp = nullptr; CHK(hipHostGetDevicePointer(&dp, dst[k], 0)); a.dst[k] = (uint8_t*)dp; a.bytes[k] = w[k] * n; }
        return a;
    };

    if (logmode) {
        fprintf(stderr, "=== A: hipMemcpyAsync D2H -> hipHostMalloc, idle stream\n");
        CHK(hipMemcpyAsync(hm[0], dsrc[0], w[0] * n, hipMemcpyDeviceToHost, down)); CHK(hipStreamSynchronize(down));
        fprintf(stderr, "=== B: same, after a kernel on another stream + hipStreamWaitEvent\n");
        busy(comp); CHK(hipEventRecord(ev, comp)); CHK(hipStreamWaitEvent(down, ev, 0));
        CHK(hipMemcpyAsync(hm[0], dsrc[0], w[0] * n, hipMemcpyDeviceToHost, down)); CHK(hipStreamSynchronize(down));
        fprintf(stderr, "=== C: -> hipHostRegister'ed memory, idle stream\n");
        CHK(hipMemcpyAsync(hr[0], dsrc[0], w[0] * n, hipMemcpyDeviceToHost, down)); CHK(hipStreamSynchronize(down));
        fprintf(stderr, "=== D: H2D from hipHostMalloc, idle stream\n");
        CHK(hipMemcpyAsync(dsrc[0], hm[0], w[0] * n, hipMemcpyHostToDevice, down)); CHK(hipStreamSynchronize(down));
        fprintf(stderr, "=== E: D2H while an H2D is in flight on another stream\n");
        CHK(hipMemcpyAsync(dsrc[1], hm[1], w[1] * n, hipMemcpyHostToDevice, comp));
        CHK(hipMemcpyAsync(hm[0], dsrc[0], w[0] * n, hipMemcpyDeviceToHost, down)); CHK(hipDeviceSynchronize());
        fprintf(stderr, "=== F: seven D2H back to back on one stream\n");
        for (int k = 0; k < 7; k++) CHK(hipMemcpyAsync(hm[k], dsrc[k], w[k] * n, hipMemcpyDeviceToHost, down));
        CHK(hipStreamSynchronize(down));
        fprintf(stderr, "=== G: hipMemcpyDtoHAsync\n");
        CHK(hipMemcpyDtoHAsync(hm[0], (hipDeviceptr_t)dsrc[0], w[0] * n, down)); CHK(hipStreamSynchronize(down));
        fprintf(stderr, "=== done\n");
        return 0;
    }

    // calibrate the busy kernel to ~20 ms
    busy(comp); CHK(hipStreamSynchronize(comp));
    for (int it = 0; it < 6; it++) {
        CHK(hipEventRecord(b0, comp)); busy(comp); CHK(hipEventRecord(b1, comp)); CHK(hipEventSynchronize(b1));
        float ms; CHK(hipEventElapsedTime(&ms, b0, b1));
        if (ms > 15 && ms < 30) break;
        busy_iters = (int)(busy_iters * 20.0 / (ms > 0.01 ? ms : 0.01)) + 1;
    }
    CHK(hipEventRecord(b0, comp)); busy(comp); CHK(hipEventRecord(b1, comp)); CHK(hipEventSynchronize(b1));
    float busy_alone; CHK(hipEventElapsedTime(&busy_alone, b0, b1));
    printf("busy kernel alone: %.3f ms (%d blocks, iters %d); piece = %zu items, %zu MB down\n", busy_alone, busy_blocks, busy_iters, n, total >> 20);

    struct Variant { const char* name; std::function<void()> go; hipStream_t st; };
    ExpArgs am = exp_args(hm), ar = exp_args(hr);
    ExpArgs apack; apack.narr = 1; apack.src[0] = dpack; apack.dst[0] = hpack; apack.bytes[0] = total;
    auto expk = [&](const ExpArgs& a, int grid, hipStream_t s) { hipLaunchKernelGGL(k_export, dim3(grid), dim3(256), 0, s, a); };
    std::vector<Variant> vs = {
        {"memcpyAsync x7 -> hipHostMalloc", [&] { for (int k = 0; k < 7; k++) CHK(hipMemcpyAsync(hm[k], dsrc[k], w[k] * n, hipMemcpyDeviceToHost, down)); }, down},
        {"memcpyAsync x7 -> hipHostRegister", [&] { for (int k = 0; k < 7; k++) CHK(hipMemcpyAsync(hr[k], dsrc[k], w[k] * n, hipMemcpyDeviceToHost, down)); }, down},
        {"memcpyAsync x1 packed", [&] { CHK(hipMemcpyAsync(hpack, dpack, total, hipMemcpyDeviceToHost, down)); }, down},
        {"memcpyAsync x7 on a HIGH-priority stream", [&] { for (int k = 0; k < 7; k++) CHK(hipMemcpyAsync(hm[k], dsrc[k], w[k] * n, hipMemcpyDeviceToHost, phi)); }, phi},
        {"export kernel grid 16, normal", [&] { expk(am, 16, down); }, down},
        {"export kernel grid 64, normal", [&] { expk(am, 64, down); }, down},
        {"export kernel grid 256, normal", [&] { expk(am, 256, down); }, down},
        {"export kernel grid 1024, normal", [&] { expk(am, 1024, down); }, down},
        {"export kernel grid 64, LOW prio", [&] { expk(am, 64, plo); }, plo},
        {"export kernel grid 64, HIGH prio", [&] { expk(am, 64, phi); }, phi},
        {"export kernel grid 256, HIGH prio", [&] { expk(am, 256, phi); }, phi},
        {"export kernel grid 256, LOW prio", [&] { expk(am, 256, plo); }, plo},
        {"export kernel grid 256 -> hipHostRegister", [&] { expk(ar, 256, down); }, down},
        {"export kernel grid 256 packed", [&] { expk(apack, 256, down); }, down},
    };
    printf("%-44s %10s %10s | beside busy: %10s %10s %10s\n", "variant", "alone ms", "GB/s", "copy ms", "GB/s", "busy ms");
    for (auto& v : vs) {
        for (int k = 0; k < 7; k++) memset(hm[k], 0, 64);
        v.go(); CHK(hipStreamSynchronize(v.st));
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) { double t0 = now(); v.go(); CHK(hipStreamSynchronize(v.st)); double t = now() - t0; if (t < best) best = t; }
        // beside the busy kernel: start it, give it 1 ms to occupy the chip, then the copy
        double cbest = 1e9; float bbest = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(b0, comp)); busy(comp); busy(comp); CHK(hipEventRecord(b1, comp));
            double t0 = now();
            while (now() - t0 < 1e-3) {}
            double t1 = now(); v.go(); CHK(hipStreamSynchronize(v.st)); double t = now() - t1;
            CHK(hipEventSynchronize(b1));
            float ms; CHK(hipEventElapsedTime(&ms, b0, b1));
            if (t < cbest) cbest = t;
            if (ms < bbest) bbest = ms;
        }
        printf("%-44s %10.3f %10.1f | %23.3f %10.1f %10.3f (2 launches; alone %.3f)\n", v.name, best * 1e3, total / best / 1e9, cbest * 1e3, total / cbest / 1e9, bbest, 2 * busy_alone);
        fflush(stdout);
    }
    // correctness of the export kernel
    bool ok = true;
    for (int k = 0; k < 7; k++) for (size_t i = 0; i < w[k] * n; i += 4099) ok = ok && hm[k][i] == k + 1 && hr[k][i] == k + 1;
    printf("export contents %s\n", ok ? "ok" : "WRONG");
    return ok ? 0 : 1;
}
