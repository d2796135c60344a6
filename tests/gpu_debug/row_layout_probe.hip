// Round 3 probe: what does HBM give a table PASS, i.e. a stream that reads three 128-byte rows and writes four per job, when
//   (A) rows are laid out job-major, tab[job][row] (the shipped layout), each lane walking CNT consecutive jobs -- a wavefront instruction touches 64 lines CNT KiB apart;
//   (B) rows are laid out row-major, tab[row][job], lane l of a wavefront walking jobs base + l, base + 64 + l, ... -- a wavefront instruction touches 64 consecutive lines.
// No arithmetic: the rows are only summed into a word that is stored.  Prints GB/s (read + written) for both, for 16-byte-per-lane accesses (what the passes issue).
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 row_layout_probe.hip -o row_layout_probe && ./row_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

constexpr int kBlock = 256, ROWS = 8, CNT = 6;
template <bool ROWMAJOR>
__device__ __forceinline__ uint4* row_ptr(uint4* tab, size_t njobs, size_t job, int e) {
    return ROWMAJOR ? tab + ((size_t)e * njobs + job) * 8 : tab + (job * ROWS + (size_t)e) * 8;      // 8 quads of 16 B per 128-byte row
}
template <bool ROWMAJOR>
__global__ __launch_bounds__(kBlock) void k_pass(uint4* tab, size_t njobs, uint32_t* sink) {
    const size_t lane = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t wave = lane / 64, l = lane % 64;
    uint32_t acc = 0;
    for (int jj = 0; jj < CNT; jj++) {
        const size_t job = ROWMAJOR ? (wave * CNT + (size_t)jj) * 64 + l : lane * CNT + (size_t)jj;
        if (job >= njobs) break;
        uint4 v[15];
        const int rd[3] = {0, 2, 3};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint4* p = row_ptr<ROWMAJOR>(tab, njobs, job, rd[r]);
#pragma unroll
            for (int q = 0; q < 5; q++) v[r * 5 + q] = p[q < 4 ? q : 6];          // x (2 quads), y (2 quads), the ninth limbs' quad
        }
#pragma unroll
        for (int k = 0; k < 15; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
#pragma unroll
        for (int r = 4; r < 8; r++) {
            uint4* p = row_ptr<ROWMAJOR>(tab, njobs, job, r);
#pragma unroll
            for (int q = 0; q < 8; q++) p[q] = make_uint4(acc + q, acc ^ r, (uint32_t)job, (uint32_t)q);   // whole rows (the passes write whole lines through an LDS transpose)
        }
    }
    if (acc == 0x12345678u) sink[lane & 63] = acc;
}

// (C) the shipped layout, but every access COOPERATIVE: instruction k of a row gather has lane l fetch quad (l % 8) of the row that lane 8k + l / 8 needs, so that each
// instruction touches 8 whole lines (8 lanes per line) instead of 64 lines with 16 bytes each -- what the passes' LDS-transposed row STORES already do; the loads would
// need the same transpose through LDS to hand every lane its own row (not done here: only the bytes moved matter to the probe)
__global__ __launch_bounds__(kBlock) void k_pass_coop(uint4* tab, size_t njobs, uint32_t* sink) {
    const size_t lane = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t wave = lane / 64, l = lane % 64;
    uint32_t acc = 0;
    for (int jj = 0; jj < CNT; jj++) {
        const int rd[3] = {0, 2, 3};
        uint4 v[24];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const size_t owner = wave * 64 + 8 * (size_t)k + l / 8, job = owner * CNT + (size_t)jj;      // the lane whose row this instruction's 8 lanes fetch
                v[r * 8 + k] = job < njobs ? row_ptr<false>(tab, njobs, job, rd[r])[l % 8] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
        for (int k = 0; k < 24; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
#pragma unroll
        for (int r = 4; r < 8; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const size_t owner = wave * 64 + 8 * (size_t)k + l / 8, job = owner * CNT + (size_t)jj;
                if (job < njobs) row_ptr<false>(tab, njobs, job, r)[l % 8] = make_uint4(acc + k, acc ^ r, (uint32_t)job, (uint32_t)l);
            }
    }
    if (acc == 0x12345678u) sink[lane & 63] = acc;
}

int main() {
    const size_t njobs = (size_t)3 << 20;
    uint4* tab; uint32_t* sink;
    if (hipMalloc(&tab, njobs * ROWS * 128) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(tab, 1, njobs * ROWS * 128);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lanes = (njobs + CNT - 1) / CNT;
    const unsigned blocks = (unsigned)((lanes + kBlock - 1) / kBlock);
    const double bytes = (double)njobs * (3 * 128 + 4 * 128);          // lines touched: 3 rows read (both halves of each line are touched), 4 rows written
    for (int layout = 0; layout < 3; layout++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipEventRecord(e0);
            if (layout == 2) hipLaunchKernelGGL(k_pass_coop, dim3(blocks), dim3(kBlock), 0, 0, tab, njobs, sink);
            else if (layout) hipLaunchKernelGGL(k_pass<true>, dim3(blocks), dim3(kBlock), 0, 0, tab, njobs, sink);
            else hipLaunchKernelGGL(k_pass<false>, dim3(blocks), dim3(kBlock), 0, 0, tab, njobs, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: %.3f ms for %.2f GB of lines read + written = %.2f TB/s\n", layout == 2 ? "tab[job][row], cooperative whole-line accesses" : layout ? "tab[row][job], interleaved lanes" : "tab[job][row], consecutive jobs per lane", best, bytes / 1e9, bytes / best / 1e9);
    }
    return 0;
}
