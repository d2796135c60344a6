// gfx950 (CDNA4 / MI355X) kernels of the batch PLUME engine.  Each kernel is a thin index-mapping wrapper over a
// per-lane body in plume_stages.h; the heavy integer work (8x32-bit-limb Fp arithmetic through v_mad_u64_u32) is
// all in those headers.  Geometry: 256-thread workgroups (4 wavefronts), one item / job-group / task per lane,
// grids of n/256 workgroups (>> 256 CUs at the batch sizes this engine is built for).
//
// LDS use: the multi-scalar kernels stage the generator's window table (1 KiB, shared by every lane of the
// workgroup) and each lane's signed window digits (4 x 33 bytes, lane-strided so a wave's accesses to one digit
// row fall in consecutive bytes).  The per-lane tables of the variable bases live in HBM: 1 KiB per base is far
// more than LDS can hold at any useful occupancy, and their traffic (~17 KiB per verify) is negligible next to
// the ~10^6 VALU instructions per item.
#include "plume_launch.h"

namespace plume {


__device__ __forceinline__ void stage_gtab(uint32_t* s_gtab, const uint32_t* gtab) {
    for (int w = threadIdx.x; w < PLUME_TAB_WORDS; w += kBlock) s_gtab[w] = gtab[w];
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_verify_ingest(VerifyArgs a) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < a.n) verify_ingest_h2c(a, i);
}

__global__ __launch_bounds__(kBlock) void k_tables(uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, int L) {
    size_t lane = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t j0 = lane * (size_t)L;
    if (j0 < njobs) {
        size_t rem = njobs - j0;
        table_build(tab, bases, jobflags, njobs, j0, (int)(rem < (size_t)L ? rem : (size_t)L));
    }
}

// blocks [0, nb): equation 1 (s*G - c*pk); blocks [nb, 2nb): equation 2 (s*H - c*nullifier) — the role is uniform
// per workgroup so the generator-table-in-LDS path never diverges inside a wavefront
__global__ __launch_bounds__(kBlock) void k_verify_msm(VerifyArgs a) {
    __shared__ uint32_t s_gtab[PLUME_TAB_WORDS];
    __shared__ int8_t s_dig[4 * PLUME_NDIG * kBlock];
    stage_gtab(s_gtab, a.gtab);
    const uint32_t nb = (a.n + kBlock - 1) / kBlock;
    const uint32_t eq = blockIdx.x >= nb ? 1u : 0u;
    const uint32_t i = (eq ? blockIdx.x - nb : blockIdx.x) * kBlock + threadIdx.x;
    if (i < a.n) verify_msm(a, i, eq, s_gtab, s_dig + threadIdx.x, kBlock);
}

__global__ __launch_bounds__(kBlock) void k_verify_finalize(VerifyArgs a) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < a.n) verify_finalize(a, i);
}

__global__ __launch_bounds__(kBlock) void k_sign_gmul(SignArgs a) {
    __shared__ uint32_t s_gtab[PLUME_TAB_WORDS];
    __shared__ int8_t s_dig[2 * PLUME_NDIG * kBlock];
    stage_gtab(s_gtab, a.gtab);
    const uint32_t nb = (a.n + kBlock - 1) / kBlock;
    const uint32_t which = blockIdx.x >= nb ? 1u : 0u;
    const uint32_t i = (which ? blockIdx.x - nb : blockIdx.x) * kBlock + threadIdx.x;
    if (i < a.n) sign_gmul(a, i, which, s_gtab, s_dig + threadIdx.x, kBlock);
}

__global__ __launch_bounds__(kBlock) void k_sign_h2c(SignArgs a) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < a.n) sign_h2c(a, i);
}

__global__ __launch_bounds__(kBlock) void k_sign_hmul(SignArgs a) {
    __shared__ int8_t s_dig[2 * PLUME_NDIG * kBlock];
    const uint32_t nb = (a.n + kBlock - 1) / kBlock;
    const uint32_t which = blockIdx.x >= nb ? 1u : 0u;
    const uint32_t i = (which ? blockIdx.x - nb : blockIdx.x) * kBlock + threadIdx.x;
    if (i < a.n) sign_hmul(a, i, which, s_dig + threadIdx.x, kBlock);
}

__global__ __launch_bounds__(kBlock) void k_sign_final(SignArgs a) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < a.n) sign_final(a, i);
}

__global__ __launch_bounds__(kBlock) void k_h2c_only(H2cArgs a) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < a.n) h2c_only(a, i);
}

// ------------------------------------------------------------------------------------------ microbenchmarks
// 8 independent accumulator chains per lane so the measured figure is issue throughput, not latency.
__global__ __launch_bounds__(kBlock) void k_microbench(int kind, int iters, uint32_t* sink) {
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    uint32_t a = tid * 2654435761u + 12345u, b = tid ^ 0x9E3779B9u;
    uint32_t out = 0;
    if (kind == 0) {         // v_mad_u64_u32
        uint64_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = tid + j;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = (uint64_t)(uint32_t)(a + j) * (uint32_t)(acc[j] >> 7 | 1u) + acc[j];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) out ^= (uint32_t)acc[j] ^ (uint32_t)(acc[j] >> 32);
    } else if (kind == 1) {  // v_add_co_u32 / v_addc_co_u32 pairs (64-bit add through the carry chain)
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { lo[j] = a + j; hi[j] = b + j; }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) { uint32_t c = 0; lo[j] = addc(lo[j], a, c); hi[j] = addc(hi[j], b, c); }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) out ^= lo[j] ^ hi[j];
    } else if (kind == 2) {  // v_mul_lo_u32
        uint32_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = a + j;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = acc[j] * (b | 1u);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) out ^= acc[j];
    } else if (kind == 3) {  // v_fma_f64
        double acc[8];
        const double m = 1.0 + (double)(a & 0xFF) * 1e-9, c = (double)(b & 0xFF) * 1e-9;
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = (double)j;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = __builtin_fma(acc[j], m, c);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) out ^= (uint32_t)(long long)acc[j];
    } else if (kind == 4) {  // v_add_u32 (plain full-rate VALU reference)
        uint32_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = a + j;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = (acc[j] + b) ^ a;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) out ^= acc[j];
    } else if (kind == 5) {  // one Fp multiplication (the unit of the roofline accounting), 2 independent chains
        fe x, y, z, w;
#pragma unroll
        for (int j = 0; j < 8; j++) { x.v[j] = a + j; y.v[j] = b * (j + 1); z.v[j] = b + j; w.v[j] = a * (j + 3); }
        for (int it = 0; it < iters; it++) { fe_mul(x, x, y); fe_mul(z, z, w); }
#pragma unroll
        for (int j = 0; j < 8; j++) out ^= x.v[j] ^ z.v[j];
    } else {                 // one Fp squaring
        fe x, z;
#pragma unroll
        for (int j = 0; j < 8; j++) { x.v[j] = a + j; z.v[j] = b + j; }
        for (int it = 0; it < iters; it++) { fe_sqr(x, x); fe_sqr(z, z); }
#pragma unroll
        for (int j = 0; j < 8; j++) out ^= x.v[j] ^ z.v[j];
    }
    if (out == 0x12345678u) sink[tid & 63] = out;  // keep the chains live without measurable traffic
}

// ------------------------------------------------------------------------------------------------ launchers
static inline unsigned nblocks(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

void launch_verify_ingest(const VerifyArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_verify_ingest, dim3(nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_tables(uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, int L, hipStream_t st) {
    size_t lanes = (njobs + L - 1) / L;
    hipLaunchKernelGGL(k_tables, dim3(nblocks(lanes)), dim3(kBlock), 0, st, tab, bases, jobflags, njobs, L);
}
void launch_verify_msm(const VerifyArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_verify_msm, dim3(2 * nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_verify_finalize(const VerifyArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_verify_finalize, dim3(nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_sign_gmul(const SignArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_sign_gmul, dim3(2 * nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_sign_h2c(const SignArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_sign_h2c, dim3(nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_sign_hmul(const SignArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_sign_hmul, dim3(2 * nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_sign_final(const SignArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_sign_final, dim3(nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_h2c_only(const H2cArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_h2c_only, dim3(nblocks(a.n)), dim3(kBlock), 0, st, a); }
void launch_microbench(int kind, int iters, uint32_t* sink, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_microbench, dim3(blocks), dim3(kBlock), 0, st, kind, iters, sink);
}

}  // namespace plume
