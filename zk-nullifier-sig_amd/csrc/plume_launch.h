// Host-visible launchers of the gfx950 kernels (defined in the k_*.hip translation units).
#pragma once
#include <hip/hip_runtime.h>

#include "plume_dedup.h"
#include "plume_stages.h"

namespace plume {

constexpr int kBlock = 256;              // threads per workgroup: four wavefronts
constexpr int kTableJobsPerLane = 6;   // (measured r02, one box, final layout: 6 -> 2.11 ms, 9 -> 2.25, 12 -> 2.30, 18 -> 2.9, 24 -> 3.3 per 2^20 verifies; the Jacobian builder it replaced: 3.1 on boxes of that speed) multiple of 3: job kinds (pk, H, nullifier) then line up across the lanes of a wavefront

void launch_verify_scalars(const VerifyArgs& a, hipStream_t st);                           // window digits of s and c; the short first equation's coefficients (plume_eis.h)
void launch_verify_ingest(const VerifyArgs& a, hipStream_t st, bool two_roles = false);   // two_roles: the small-batch form, two lanes per item (k_verify_ingest_split), which runs the scalar stage too when a.scalars_in_ingest is set: no launch_verify_scalars then
// the window tables of njobs bases (plume_ec.h: rows P, theta P, 2P): pass A, one batched inversion, pass B.  scr: tables_scratch_bytes(njobs, jobs_per_lane) bytes
size_t tables_scratch_bytes(size_t njobs, int jobs_per_lane);
void launch_tables(uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, int jobs_per_lane, uint32_t* scr, hipStream_t st);
void launch_verify_msm(const VerifyArgs& a, hipStream_t st);
const char* verify_msm_kernel_name(const VerifyArgs& a);      // the kernel launch_verify_msm picks for this call
void launch_verify_finalize(const VerifyArgs& a, hipStream_t st);
void launch_sign_gmul(const SignArgs& a, hipStream_t st);
void launch_sign_h2c(const SignArgs& a, hipStream_t st);
void launch_sign_hdbl(const SignArgs& a, hipStream_t st);     // 2^64 H per item (after launch_sign_h2c, before the table stage over 2n jobs)
void launch_sign_hmul(const SignArgs& a, hipStream_t st);
void launch_sign_final(const SignArgs& a, hipStream_t st);
void launch_normalize(uint32_t* pts, const uint8_t* inf, size_t npts, hipStream_t st);
void launch_decompress(const DecompressArgs& a, hipStream_t st);
void launch_h2c_only(const H2cArgs& a, hipStream_t st);
void launch_h2c_intermediates(const H2cInterArgs& a, hipStream_t st);
void launch_scalars_der(const DerArgs& a, hipStream_t st);
void launch_registers_from_be(uint8_t* out, const uint8_t* in, size_t nvalues, hipStream_t st);
// the generator's fixed tables (once per context): gtab = (1..2^(GW-1)) * G, gcomb = the signer's doubling-free comb; base18: scratch for the window bases
#define PLUME_FIXED_BASES (1 + PLUME_COMB_WINDOWS + PLUME_GSCAN_WINDOWS)
void launch_fixed_tables(uint32_t* gtab, uint32_t* gcomb, uint32_t* gscan, uint32_t* base18 /* PLUME_FIXED_BASES x 18 words */, hipStream_t st);
size_t dedup_blockcnt_bytes(size_t n);                    // size of DedupArgs::blockcnt
void launch_dedup(const DedupArgs& a, hipStream_t st);   // clear, insert, mark, sum (plume_dedup.h)
void launch_microbench(int kind, int iters, uint32_t* sink, int blocks, hipStream_t st);
void launch_gather_probe(const uint32_t* tab, uint32_t nrows, int iters, uint32_t* sink, int blocks, hipStream_t st);   // HBM-counter calibration (plume_microbench kind 9)

}  // namespace plume
