// Host-visible launchers of the aggregate-check kernels (plume_agg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "plume_aggregate.h"
#include "plume_launch.h"

namespace plume {

constexpr uint32_t kAggScanLanes = 16384;  // lanes of the offsets scan (a contiguous range of keys each)
constexpr int kAggScanTop = 128;           // lanes of the workgroup that scans their range sums
constexpr int kAggSortThreads = 1024;     // lanes of a sort workgroup (one per CU: its bins fill most of the LDS)
constexpr uint32_t kAggTileItems = 32768;  // items per sort tile

void launch_agg_normalize_h(const AggArgs& a, hipStream_t st);
void launch_agg_item_terms(const AggArgs& a, hipStream_t st);
size_t agg_sort_tile_words(const AggArgs& a);                      // words of the per-(window, tile, bucket) array
void launch_agg_sort(const AggArgs& a, uint32_t* tiles, uint32_t* scanpart /* kAggScanLanes words */, hipStream_t st);   // count per tile, totals, scan, offsets, place
constexpr int kAggPermBlocks = 128, kAggPermThreads = 1024;   // workgroups x lanes of the run-length ordering
// windows [j0, j0 + nwin): order their keys by run length, then sum the runs.  perm: nkeys words; hist: kAggPermBlocks x 256 words
void launch_agg_bucket_sum(const AggArgs& a, uint32_t j0, uint32_t nwin, uint32_t* perm, uint32_t* hist, hipStream_t st);
size_t agg_reduce_points(const AggArgs& a, uint32_t nwin);          // points per reduction array of a group of nwin windows
int launch_agg_reduce(const AggArgs& a, uint32_t j0, uint32_t nwin, uint32_t* red0, uint8_t* inf0, uint32_t* red1, uint8_t* inf1, hipStream_t st);
size_t agg_scalar_sum_words(size_t n);
const uint32_t* launch_agg_scalar_sum(const uint32_t* gs, size_t n, uint32_t* s0, uint32_t* s1, hipStream_t st);
void launch_agg_gterm(const AggArgs& a, const uint32_t* gsum, uint32_t* gout, uint8_t* goutinf, hipStream_t st);
void launch_agg_final(const AggArgs& a, const uint32_t* lo, const uint8_t* loinf, uint32_t nlo, const uint32_t* hi, const uint8_t* hiinf, uint32_t nhi, const uint32_t* gpt, const uint8_t* gptinf,
                      const uint8_t* carry, hipStream_t st);
void launch_agg_combine(const uint8_t* records, uint32_t m, uint8_t* result, hipStream_t st);

}  // namespace plume
