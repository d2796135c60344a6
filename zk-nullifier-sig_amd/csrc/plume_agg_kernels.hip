// gfx950 kernels of the aggregate random-linear-combination check (plume_aggregate.h): thin index-mapping wrappers, 256-thread workgroups.
#include "plume_agg_launch.h"

#define PLUME_AGG_BOUNDS(W) __launch_bounds__(kBlock, W)

namespace plume {

__global__ PLUME_AGG_BOUNDS(2) void k_agg_normalize_h(AggArgs a, size_t nlanes) {
    const size_t lane = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane < nlanes) agg_normalize_h(a, lane, nlanes);
}
__global__ PLUME_AGG_BOUNDS(2) void k_agg_item_terms(AggArgs a) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < a.n) agg_item_terms(a, i);
}
// workgroup (window, tile): the window's bins live in LDS (up to 2^15 words = 128 KiB; a gfx950 workgroup may hold all 160 KiB)
template <bool SCATTER>
__global__ __launch_bounds__(kAggSortThreads) void k_agg_tile_pairs(AggArgs a, uint32_t tile, uint32_t ntiles, uint32_t* tiles) {
    __shared__ uint32_t bins[1u << 15];
    const uint32_t j = blockIdx.x / ntiles, b = blockIdx.x - j * ntiles;
    uint32_t* mine = tiles + (size_t)blockIdx.x * a.nbuckets;
    for (uint32_t e = threadIdx.x; e < a.nbuckets; e += kAggSortThreads) bins[e] = SCATTER ? mine[e] : 0u;
    __syncthreads();
    agg_tile_pairs<SCATTER>(a, j, b, tile, threadIdx.x, kAggSortThreads, bins);
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < a.nbuckets; e += kAggSortThreads) mine[e] = bins[e];
    }
}
__global__ __launch_bounds__(kBlock) void k_agg_tile_totals(AggArgs a, const uint32_t* tiles, uint32_t ntiles) {
    const uint32_t key = blockIdx.x * kBlock + threadIdx.x;
    if (key < a.nkeys) agg_tile_totals(a, tiles, ntiles, key);
}
__global__ __launch_bounds__(kBlock) void k_agg_tile_offsets(AggArgs a, uint32_t* tiles, uint32_t ntiles) {
    const uint32_t key = blockIdx.x * kBlock + threadIdx.x;
    if (key < a.nkeys) agg_tile_offsets(a, tiles, ntiles, key);
}
// exclusive scan of count[0 .. nkeys]: kAggScanLanes lanes own a contiguous range each (range sums -> part), one workgroup scans part, the lanes write back
__global__ __launch_bounds__(kBlock) void k_agg_scan_sums(AggArgs a, uint32_t* part) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t < kAggScanLanes) agg_scan_phase0(a.count, a.nkeys + 1, t, kAggScanLanes, part);
}
__global__ __launch_bounds__(kAggScanTop) void k_agg_scan_top(uint32_t* part) {
    __shared__ uint32_t top[kAggScanTop];
    agg_scan_phase0(part, kAggScanLanes, threadIdx.x, kAggScanTop, top);
    __syncthreads();
    if (threadIdx.x == 0) agg_scan_mid(kAggScanTop, top);
    __syncthreads();
    agg_scan_phase1(part, kAggScanLanes, threadIdx.x, kAggScanTop, top);
}
__global__ __launch_bounds__(kBlock) void k_agg_scan_write(AggArgs a, const uint32_t* part) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t < kAggScanLanes) agg_scan_phase1(a.count, a.nkeys + 1, t, kAggScanLanes, part);
}
template <bool SCATTER>
__global__ __launch_bounds__(kAggPermThreads) void k_agg_perm_pairs(AggArgs a, uint32_t k0, uint32_t k1, uint32_t* hist, uint32_t* perm) {
    __shared__ uint32_t bins[PLUME_AGG_LEN_BINS];
    uint32_t* mine = hist + (size_t)blockIdx.x * PLUME_AGG_LEN_BINS;
    for (uint32_t e = threadIdx.x; e < PLUME_AGG_LEN_BINS; e += kAggPermThreads) bins[e] = SCATTER ? mine[e] : 0u;
    __syncthreads();
    agg_perm_pairs<SCATTER>(a, k0, k1, blockIdx.x, gridDim.x, threadIdx.x, kAggPermThreads, bins, perm);
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < PLUME_AGG_LEN_BINS; e += kAggPermThreads) mine[e] = bins[e];
    }
}
__global__ __launch_bounds__(PLUME_AGG_LEN_BINS) void k_agg_perm_offsets(uint32_t* hist, uint32_t nblk) {
    __shared__ uint32_t part[PLUME_AGG_LEN_BINS];
    agg_perm_phase0(hist, nblk, threadIdx.x, part);
    __syncthreads();
    if (threadIdx.x == 0) agg_perm_mid(part);
    __syncthreads();
    agg_perm_phase1(hist, nblk, threadIdx.x, part);
}
__global__ PLUME_AGG_BOUNDS(3) void k_agg_bucket_sum(AggArgs a, const uint32_t* perm, uint32_t k0, uint32_t k1) {
    const uint32_t lane = k0 + blockIdx.x * kBlock + threadIdx.x;
    if (lane < k1) agg_bucket_sum(a, perm[lane]);
}
__global__ PLUME_AGG_BOUNDS(2) void k_agg_chunk_reduce(AggArgs a, uint32_t chunk, uint32_t nchunks, uint32_t* out, uint8_t* outinf, uint32_t j0, uint32_t nwin) {
    const uint32_t lane = blockIdx.x * 64 + threadIdx.x;
    if (lane < nwin * nchunks) agg_chunk_reduce(a, j0 + lane / nchunks, lane % nchunks, chunk, nchunks, out, outinf, j0, nwin);
}
__global__ PLUME_AGG_BOUNDS(2) void k_agg_group_sum(const uint32_t* in, const uint8_t* ininf, uint32_t m_in, uint32_t g, uint32_t* out, uint8_t* outinf, uint32_t m_out, uint32_t nw) {
    const uint32_t lane = blockIdx.x * 64 + threadIdx.x;
    if (lane < nw * m_out) agg_group_sum(in, ininf, m_in, g, out, outinf, m_out, nw, lane / m_out, lane % m_out);
}
__global__ PLUME_AGG_BOUNDS(2) void k_agg_window_shift(AggArgs a, uint32_t* pts, uint8_t* inf, uint32_t j0, uint32_t nwin) {
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k < nwin) agg_window_shift(a, pts, inf, j0 + k, j0, nwin);
}
__global__ __launch_bounds__(kBlock) void k_agg_scalar_sum(const uint32_t* in, size_t nin, uint32_t* out, size_t nout) {
    const size_t lane = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane < nout) agg_scalar_sum(in, nin, out, nout, lane);
}
__global__ PLUME_AGG_BOUNDS(1) void k_agg_gterm(AggArgs a, const uint32_t* gsum, uint32_t* gout, uint8_t* goutinf) {
    if (blockIdx.x == 0 && threadIdx.x == 0) agg_gterm(a, gsum, gout, goutinf);
}
__global__ PLUME_AGG_BOUNDS(1) void k_agg_final(AggArgs a, const uint32_t* lo, const uint8_t* loinf, uint32_t nlo, const uint32_t* hi, const uint8_t* hiinf, uint32_t nhi, const uint32_t* gpt,
                                                const uint8_t* gptinf, const uint8_t* carry) {
    if (blockIdx.x == 0 && threadIdx.x == 0) agg_final(a, lo, loinf, nlo, hi, hiinf, nhi, gpt, gptinf, carry);
}
__global__ PLUME_AGG_BOUNDS(1) void k_agg_combine(const uint8_t* records, uint32_t m, uint8_t* result) {
    if (blockIdx.x == 0 && threadIdx.x == 0) agg_combine(records, m, result);
}

static inline unsigned nblk(size_t n, unsigned b = kBlock) { return (unsigned)((n + b - 1) / b); }

void launch_agg_normalize_h(const AggArgs& a, hipStream_t st) {
    const size_t nlanes = ((size_t)a.n + PLUME_AGG_NORM_K - 1) / PLUME_AGG_NORM_K;
    hipLaunchKernelGGL(k_agg_normalize_h, dim3(nblk(nlanes)), dim3(kBlock), 0, st, a, nlanes);
}
void launch_agg_item_terms(const AggArgs& a, hipStream_t st) { hipLaunchKernelGGL(k_agg_item_terms, dim3(nblk(a.n)), dim3(kBlock), 0, st, a); }
void agg_sort_tiling(const AggArgs& a, uint32_t& tile, uint32_t& ntiles) {
    ntiles = (a.n + kAggTileItems - 1) / kAggTileItems;
    if (ntiles < 1) ntiles = 1;
    if (ntiles > 64) ntiles = 64;
    tile = (a.n + ntiles - 1) / ntiles;
    if (tile < 1) tile = 1;
}
size_t agg_sort_tile_words(const AggArgs& a) {
    uint32_t tile, ntiles;
    agg_sort_tiling(a, tile, ntiles);
    return (size_t)a.nw_long * ntiles * a.nbuckets;
}
void launch_agg_sort(const AggArgs& a, uint32_t* tiles, uint32_t* scanpart, hipStream_t st) {
    uint32_t tile, ntiles;
    agg_sort_tiling(a, tile, ntiles);
    const unsigned nb = (unsigned)a.nw_long * ntiles;
    hipLaunchKernelGGL(k_agg_tile_pairs<false>, dim3(nb), dim3(kAggSortThreads), 0, st, a, tile, ntiles, tiles);
    hipLaunchKernelGGL(k_agg_tile_totals, dim3(nblk(a.nkeys)), dim3(kBlock), 0, st, a, tiles, ntiles);
    hipLaunchKernelGGL(k_agg_scan_sums, dim3(nblk(kAggScanLanes)), dim3(kBlock), 0, st, a, scanpart);
    hipLaunchKernelGGL(k_agg_scan_top, dim3(1), dim3(kAggScanTop), 0, st, scanpart);
    hipLaunchKernelGGL(k_agg_scan_write, dim3(nblk(kAggScanLanes)), dim3(kBlock), 0, st, a, scanpart);
    hipLaunchKernelGGL(k_agg_tile_offsets, dim3(nblk(a.nkeys)), dim3(kBlock), 0, st, a, tiles, ntiles);
    hipLaunchKernelGGL(k_agg_tile_pairs<true>, dim3(nb), dim3(kAggSortThreads), 0, st, a, tile, ntiles, tiles);
}
// the keys of windows [j0, j0 + nwin): order them by run length, then sum the runs.  perm: nkeys words; hist: kAggPermBlocks x PLUME_AGG_LEN_BINS words
void launch_agg_bucket_sum(const AggArgs& a, uint32_t j0, uint32_t nwin, uint32_t* perm, uint32_t* hist, hipStream_t st) {
    const uint32_t k0 = j0 * a.nbuckets, k1 = (j0 + nwin) * a.nbuckets;
    hipLaunchKernelGGL(k_agg_perm_pairs<false>, dim3(kAggPermBlocks), dim3(kAggPermThreads), 0, st, a, k0, k1, hist, perm);
    hipLaunchKernelGGL(k_agg_perm_offsets, dim3(1), dim3(PLUME_AGG_LEN_BINS), 0, st, hist, (uint32_t)kAggPermBlocks);
    hipLaunchKernelGGL(k_agg_perm_pairs<true>, dim3(kAggPermBlocks), dim3(kAggPermThreads), 0, st, a, k0, k1, hist, perm);
    hipLaunchKernelGGL(k_agg_bucket_sum, dim3(nblk(k1 - k0)), dim3(kBlock), 0, st, a, perm, k0, k1);
}
size_t agg_reduce_points(const AggArgs& a, uint32_t nwin) {
    const uint32_t chunk = a.nbuckets < PLUME_AGG_CHUNK ? a.nbuckets : PLUME_AGG_CHUNK;
    return (size_t)nwin * ((a.nbuckets + chunk - 1) / chunk);
}
// bucket sums of windows [j0, j0 + nwin) -> one shifted sum per window, left in the array the return value names (0: red0, 1: red1); both hold
// agg_reduce_points(a, nwin) points
int launch_agg_reduce(const AggArgs& a, uint32_t j0, uint32_t nwin, uint32_t* red0, uint8_t* inf0, uint32_t* red1, uint8_t* inf1, hipStream_t st) {
    const uint32_t chunk = a.nbuckets < PLUME_AGG_CHUNK ? a.nbuckets : PLUME_AGG_CHUNK;
    uint32_t m = (a.nbuckets + chunk - 1) / chunk;
    hipLaunchKernelGGL(k_agg_chunk_reduce, dim3(nblk((size_t)nwin * m, 64)), dim3(64), 0, st, a, chunk, m, red0, inf0, j0, nwin);
    int cur = 0;
    while (m > 1) {
        const uint32_t g = m < PLUME_AGG_GROUP ? m : PLUME_AGG_GROUP, mo = (m + g - 1) / g;
        hipLaunchKernelGGL(k_agg_group_sum, dim3(nblk((size_t)nwin * mo, 64)), dim3(64), 0, st, cur ? red1 : red0, cur ? inf1 : inf0, m, g, cur ? red0 : red1, cur ? inf0 : inf1, mo, nwin);
        cur ^= 1;
        m = mo;
    }
    hipLaunchKernelGGL(k_agg_window_shift, dim3(nblk(nwin, 64)), dim3(64), 0, st, a, cur ? red1 : red0, cur ? inf1 : inf0, j0, nwin);
    return cur;
}
size_t agg_scalar_sum_words(size_t n) { return 8 * ((n + PLUME_AGG_SUM_K - 1) / PLUME_AGG_SUM_K); }
// sum of the n values of gs (SoA) -> 8 words at the pointer returned (inside s0 or s1; each agg_scalar_sum_words(n) words)
const uint32_t* launch_agg_scalar_sum(const uint32_t* gs, size_t n, uint32_t* s0, uint32_t* s1, hipStream_t st) {
    const uint32_t* in = gs;
    size_t nin = n;
    int cur = 0;
    do {
        const size_t nout = (nin + PLUME_AGG_SUM_K - 1) / PLUME_AGG_SUM_K;
        uint32_t* out = cur ? s1 : s0;
        hipLaunchKernelGGL(k_agg_scalar_sum, dim3(nblk(nout)), dim3(kBlock), 0, st, in, nin, out, nout);
        in = out; nin = nout; cur ^= 1;
    } while (nin > 1);
    return in;
}
void launch_agg_gterm(const AggArgs& a, const uint32_t* gsum, uint32_t* gout, uint8_t* goutinf, hipStream_t st) { hipLaunchKernelGGL(k_agg_gterm, dim3(1), dim3(64), 0, st, a, gsum, gout, goutinf); }
void launch_agg_final(const AggArgs& a, const uint32_t* lo, const uint8_t* loinf, uint32_t nlo, const uint32_t* hi, const uint8_t* hiinf, uint32_t nhi, const uint32_t* gpt, const uint8_t* gptinf,
                      const uint8_t* carry, hipStream_t st) {
    hipLaunchKernelGGL(k_agg_final, dim3(1), dim3(64), 0, st, a, lo, loinf, nlo, hi, hiinf, nhi, gpt, gptinf, carry);
}
void launch_agg_combine(const uint8_t* records, uint32_t m, uint8_t* result, hipStream_t st) { hipLaunchKernelGGL(k_agg_combine, dim3(1), dim3(64), 0, st, records, m, result); }

}  // namespace plume
