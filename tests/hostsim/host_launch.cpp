// TEST INFRASTRUCTURE ONLY: the launchers of csrc/plume_launch.h and csrc/plume_agg_launch.h for the CPU build of the library's host side (tests/hostsim/Makefile).
// Each "launch" queues, on the mock runtime's stream (mockhip/hip/hip_runtime.h), a plain loop over the SAME grid and the same (workgroup, lane) -> item mapping as the
// kernel of that name in csrc/plume_kernels.hip / plume_agg_kernels.hip, calling the same per-lane bodies (csrc/plume_stages.h, plume_ec.h, plume_aggregate.h, plume_dedup.h)
// on the same buffers -- so the workspace sizes, offsets and lifetimes the host code computes are exercised for real under the sanitizers.  Workgroup-shared memory is a
// local array per simulated workgroup; a workgroup barrier is the end of a loop over its lanes.  What this cannot show (wavefront lock step, LDS banking, occupancy) is the
// GPU tests' business.
#include <algorithm>
#include <cstring>
#include <vector>

#include "plume_agg_launch.h"
#include "plume_launch.h"

namespace plume {

static inline unsigned nblocks(size_t n, unsigned b = kBlock) { return (unsigned)((n + b - 1) / b); }
// one workgroup-wide loop nest: f(workgroup, lane)
template <class F>
static void grid(unsigned blocks, unsigned threads, F f) { for (unsigned b = 0; b < blocks; b++) for (unsigned t = 0; t < threads; t++) f(b, t); }

void launch_verify_scalars(const VerifyArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] {
        a.redo[0] = 0;
        grid(nblocks(a.n), kBlock, [&](unsigned b, unsigned t) { const uint32_t i = b * kBlock + t; if (i < a.n) verify_scalars(a, i); });
    });
}
void launch_verify_ingest(const VerifyArgs& a0, hipStream_t st, bool two_roles) {
    if (!two_roles) {
        mockhip::launch(st, [a = a0] { grid(nblocks(a.n), kBlock, [&](unsigned b, unsigned t) { const uint32_t i = b * kBlock + t; if (i < a.n) verify_ingest_h2c(a, i); }); });
        return;
    }
    mockhip::launch(st, [a = a0] {                            // k_verify_ingest_split: roles A and B of 128 items per workgroup, meeting at two barriers
        constexpr uint32_t H = kBlock / 2;
        const unsigned nb = (a.n + H - 1) / H;
        if (a.scalars_in_ingest) a.redo[0] = 0;
        std::vector<ingest_xch> x(H);
        std::vector<ingest_a_state> stt(H);
        for (unsigned blk = 0; blk < nb; blk++) {
            const uint32_t cnt = std::min<uint32_t>(H, a.n - blk * H);
            for (uint32_t l = 0; l < cnt; l++) verify_ingest_b1(a, blk * H + l, x[l]);
            for (uint32_t l = 0; l < cnt; l++) verify_ingest_a1(a, blk * H + l, x[l], stt[l]);
            for (uint32_t l = 0; l < cnt; l++) verify_ingest_b2(x[l]);
            for (uint32_t l = 0; l < cnt; l++) verify_ingest_a2(a, blk * H + l, x[l], stt[l]);
            for (uint32_t l = 0; l < cnt; l++) verify_ingest_a3(a, blk * H + l, x[l], stt[l]);
            if (a.scalars_in_ingest) for (uint32_t l = 0; l < cnt; l++) verify_scalars(a, blk * H + l);          // role B's last duty: the scalar stage
        }
    });
}

constexpr int kTabBlock = 128;
constexpr int kTabInvK = 8;
static size_t tables_park_bytes(size_t njobs, int L) {
    const size_t lanes = (njobs + L - 1) / L;
    return (size_t)nblocks(lanes) * kBlock * (size_t)L * PLUME_TAB_SCR_WORDS * 4;
}
size_t tables_scratch_bytes(size_t njobs, int L) {
    const size_t lanes = (njobs + L - 1) / L;
    return tables_park_bytes(njobs, L) + (size_t)nblocks(lanes) * kBlock * (PLUME_FE_WORDS * 4 + 1) + 16;
}
void launch_tables(uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, int L, uint32_t* scr, hipStream_t st) {
    const size_t lanes = (njobs + L - 1) / L;
    const unsigned blocks = nblocks(lanes) * (kBlock / kTabBlock);
    const size_t nl = (size_t)blocks * kTabBlock, T = (nl + kTabInvK - 1) / kTabInvK;
    uint32_t* carry = scr + tables_park_bytes(njobs, L) / 4;
    uint8_t* guardf = reinterpret_cast<uint8_t*>(carry + nl * PLUME_FE_WORDS);
    auto span = [=](size_t lane, size_t& j0, int& cnt) { j0 = lane * (size_t)L; cnt = j0 < njobs ? (int)(njobs - j0 < (size_t)L ? njobs - j0 : (size_t)L) : 0; };
    mockhip::launch(st, [=] {
        grid(blocks, kTabBlock, [&](unsigned b, unsigned t) {
            const size_t lane = (size_t)b * kTabBlock + t;
            size_t j0; int cnt; span(lane, j0, cnt);
            uint32_t* myscr = scr + (size_t)b * ((size_t)L * PLUME_TAB_SCR_WORDS * kTabBlock);
            fe c; bool g = false;
            tab_pass_a(bases, jobflags, njobs, j0, cnt, myscr, (size_t)kTabBlock, t, c, g);
            st_fe_soa(carry, nl, lane, c); guardf[lane] = g ? 1 : 0;
        });
    });
    mockhip::launch(st, [=] { grid(nblocks(T), kBlock, [&](unsigned b, unsigned t) { const size_t k = (size_t)b * kBlock + t; if (k < T) tab_invert_group<kTabInvK>(carry, nl, T, k); }); });
    mockhip::launch(st, [=] {
        const DirectRowSinkSync sink;
        grid(blocks, kTabBlock, [&](unsigned b, unsigned t) {
            const size_t lane = (size_t)b * kTabBlock + t;
            size_t j0; int cnt; span(lane, j0, cnt);
            const uint32_t* myscr = scr + (size_t)b * ((size_t)L * PLUME_TAB_SCR_WORDS * kTabBlock);
            fe c; ld_fe_soa(c, carry, nl, lane);
            tab_pass_b(tab, bases, jobflags, njobs, j0, cnt, myscr, (size_t)kTabBlock, t, c, guardf[lane] != 0, sink);
        });
    });
}

#define PLUME_MSM_DIG_ROWS (2 * PLUME_NDIG + PLUME_NPOS)
const char* verify_msm_kernel_name(const VerifyArgs& a) { return a.msm_pair && !verify_eq1_short(a) ? "k_verify_msm_pair" : verify_eq1_short(a) ? "k_verify_msm_s" : "k_verify_msm"; }
void launch_verify_msm(const VerifyArgs& a0, hipStream_t st) {
    if (a0.msm_pair && !verify_eq1_short(a0)) mockhip::launch(st, [a = a0] {       // k_verify_msm_pair: 128 tasks of one equation per workgroup, halves 1 first (they park their sums), then halves 0 and the join
        constexpr uint32_t H = kBlock / 2;
        std::vector<int8_t> s_dig((size_t)PLUME_MSM_DIG_ROWS * kBlock);
        std::vector<jac> parked(H);
        std::vector<uint8_t> okb(H);
        const uint32_t nb = (a.n + H - 1) / H;
        for (uint32_t b = 0; b < 2 * nb; b++) {
            const uint32_t eq = b >= nb ? 1u : 0u, blk = eq ? b - nb : b;
            for (uint32_t l = 0; l < H && blk * H + l < a.n; l++) okb[l] = verify_msm_half(a, blk * H + l, eq, 1, a.gtab, s_dig.data() + H + l, kBlock, parked[l]) ? 1 : 0;
            for (uint32_t l = 0; l < H && blk * H + l < a.n; l++) {
                jac acc;
                const bool ok = verify_msm_half(a, blk * H + l, eq, 0, a.gtab, s_dig.data() + l, kBlock, acc);
                verify_msm_join(a, blk * H + l, eq, acc, ok, parked[l], okb[l] != 0);
            }
        }
    });
    else mockhip::launch(st, [a = a0] {
        std::vector<int8_t> s_dig((size_t)PLUME_MSM_DIG_ROWS * kBlock);
        const uint32_t nb = nblocks(a.n);
        grid(2 * nb, kBlock, [&](unsigned b, unsigned t) {
            const uint32_t eq = b >= nb ? 1u : 0u, i = (eq ? b - nb : b) * kBlock + t;
            if (i >= a.n) return;
            if (verify_eq1_short(a)) verify_msm<false, 1>(a, i, eq, a.gtab, s_dig.data() + t, kBlock);
            else verify_msm<false, 0>(a, i, eq, a.gtab, s_dig.data() + t, kBlock);
        });
    });
    mockhip::launch(st, [a = a0] {                            // k_verify_msm_redo: workgroups of 64 lanes, grid-stride over the filed tasks
        constexpr int R = 64;
        std::vector<int8_t> s_dig((size_t)PLUME_MSM_DIG_ROWS * R);
        const uint32_t count = a.redo[0];
        for (uint32_t k = 0; k < count; k++) { const uint32_t t = a.redo[1 + k]; verify_msm<true>(a, t >> 1, t & 1u, a.gtab, s_dig.data() + (k % R), R); }
    });
}
void launch_verify_finalize(const VerifyArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] { grid(nblocks(a.n), kBlock, [&](unsigned b, unsigned t) { const uint32_t i = b * kBlock + t; if (i < a.n) verify_finalize(a, i); }); });
}

void launch_sign_gmul(const SignArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] {
        const uint32_t nb = nblocks(a.n);
        grid(2 * nb, kBlock, [&](unsigned b, unsigned t) {
            const uint32_t which = b >= nb ? 1u : 0u, i = (which ? b - nb : b) * kBlock + t;
            if (i >= a.n) return;
            if (a.uniform == 2) sign_gmul<2>(a, i, which); else if (a.uniform) sign_gmul<1>(a, i, which); else sign_gmul(a, i, which);
        });
    });
}
void launch_sign_h2c(const SignArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] { grid(nblocks(a.n), kBlock, [&](unsigned b, unsigned t) { const uint32_t i = b * kBlock + t; if (i < a.n) sign_h2c(a, i); }); });
}
void launch_sign_hdbl(const SignArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] { grid(nblocks(a.n), kBlock, [&](unsigned b, unsigned t) { const uint32_t i = b * kBlock + t; if (i < a.n) sign_hdbl(a, i); }); });
}
void launch_sign_hmul(const SignArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] {
        std::vector<int8_t> s_dig((size_t)PLUME_SIGN_K * PLUME_NPOSK * kBlock);
        const uint32_t nb = nblocks(a.n);
        grid(2 * nb, kBlock, [&](unsigned b, unsigned t) {
            int8_t* dg = s_dig.data() + t;
            if (a.uniform == 2) {                              // level 2: the two tasks of an item in adjacent lanes
                const uint32_t which = t & 1u, i = (b * kBlock + t) >> 1;
                if (i < a.n) sign_hmul<2>(a, i, which, dg, kBlock);
                return;
            }
            const uint32_t which = b >= nb ? 1u : 0u, i = (which ? b - nb : b) * kBlock + t;
            if (i >= a.n) return;
            if (a.uniform) sign_hmul<1>(a, i, which, dg, kBlock); else sign_hmul(a, i, which, dg, kBlock);
        });
    });
}
void launch_sign_final(const SignArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] { grid(nblocks(a.n), kBlock, [&](unsigned b, unsigned t) { const uint32_t i = b * kBlock + t; if (i < a.n) sign_final(a, i); }); });
}
void launch_normalize(uint32_t* pts, const uint8_t* inf, size_t npts, hipStream_t st) {
    const size_t nlanes = (npts + PLUME_NORM_K - 1) / PLUME_NORM_K;
    mockhip::launch(st, [=] { for (size_t lane = 0; lane < nlanes; lane++) normalize_points(pts, inf, npts, lane, nlanes); });
}
void launch_decompress(const DecompressArgs& a0, hipStream_t st) { mockhip::launch(st, [a = a0] { for (uint32_t i = 0; i < a.n; i++) decompress_item(a, i); }); }
void launch_h2c_only(const H2cArgs& a0, hipStream_t st) { mockhip::launch(st, [a = a0] { for (uint32_t i = 0; i < a.n; i++) h2c_only(a, i); }); }
void launch_h2c_intermediates(const H2cInterArgs& a0, hipStream_t st) { mockhip::launch(st, [a = a0] { for (uint32_t i = 0; i < a.n; i++) h2c_intermediates(a, i); }); }
void launch_scalars_der(const DerArgs& a0, hipStream_t st) { mockhip::launch(st, [a = a0] { for (uint32_t i = 0; i < a.n; i++) scalar_to_sec1_der(a, i); }); }
void launch_registers_from_be(uint8_t* out, const uint8_t* in, size_t nvalues, hipStream_t st) {
    mockhip::launch(st, [=] { for (size_t k = 0; k < nvalues; k++) registers_from_be(out, in, k); });
}
void launch_fixed_tables(uint32_t* gtab, uint32_t* gcomb, uint32_t* gscan, uint32_t* base18, hipStream_t st) {
    uint32_t* cb = base18 + 2 * PLUME_FE_WORDS;
    uint32_t* sb = cb + (size_t)PLUME_COMB_WINDOWS * 2 * PLUME_FE_WORDS;
    auto build = [st](uint32_t* rows, uint32_t* bases, uint32_t entries, uint32_t nwin, uint32_t W) {
        mockhip::launch(st, [=] { for (uint32_t w = 0; w < nwin; w++) fixed_window_base(bases + (size_t)w * 2 * PLUME_FE_WORDS, W * w); });
        mockhip::launch(st, [=] { for (size_t lane = 0; lane < (size_t)entries * nwin; lane++) fixed_table_lane(rows, bases, entries, lane); });
    };
    if (gscan) build(gscan, sb, PLUME_GSCAN_ENTRIES, PLUME_GSCAN_WINDOWS, PLUME_GSCAN_W);
    if (gtab) build(gtab, base18, PLUME_GTAB_ENTRIES, 1u, 0u);
    if (gcomb) build(gcomb, cb, PLUME_COMB_ENTRIES, PLUME_COMB_WINDOWS, PLUME_COMB_W);
}
size_t dedup_blockcnt_bytes(size_t n) { return (size_t)nblocks(n) * 4; }
void launch_dedup(const DedupArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] { for (uint32_t s = 0; s <= a.mask; s++) { dedup_clear(a, s); if (s == 0xFFFFFFFFu) break; } });
    mockhip::launch(st, [a = a0] { for (uint32_t i = 0; i < a.n; i++) dedup_insert(a, i); });
    mockhip::launch(st, [a = a0] {
        for (unsigned b = 0; b < nblocks(a.n); b++) {
            uint32_t c = 0;
            for (unsigned t = 0; t < (unsigned)kBlock; t++) { const uint32_t i = b * kBlock + t; if (i < a.n && dedup_mark(a, i)) c++; }
            a.blockcnt[b] = c;
        }
    });
    mockhip::launch(st, [a = a0] { unsigned long long c = 0; for (unsigned b = 0; b < nblocks(a.n); b++) c += a.blockcnt[b]; *a.n_unique = c; });
}
// the issue-rate and HBM-calibration probes measure a GPU: here they only honour their output contract
void launch_microbench(int, int, uint32_t* sink, int, hipStream_t st) { mockhip::launch(st, [=] { sink[64] = 1; sink[65] = 0; }); }
void launch_gather_probe(const uint32_t*, uint32_t, int, uint32_t*, int, hipStream_t) {}

// ------------------------------------------------------------------------------------------------ the aggregate check (plume_agg_kernels.hip)
void launch_agg_normalize_h(const AggArgs& a0, hipStream_t st) {
    mockhip::launch(st, [a = a0] { const size_t nlanes = ((size_t)a.n + PLUME_AGG_NORM_K - 1) / PLUME_AGG_NORM_K; for (size_t lane = 0; lane < nlanes; lane++) agg_normalize_h(a, lane, nlanes); });
}
void launch_agg_item_terms(const AggArgs& a0, hipStream_t st) { mockhip::launch(st, [a = a0] { for (uint32_t i = 0; i < a.n; i++) agg_item_terms(a, i); }); }
static void agg_sort_tiling(const AggArgs& a, uint32_t& tile, uint32_t& ntiles) {
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
void launch_agg_sort(const AggArgs& a0, uint32_t* tiles, uint32_t* scanpart, hipStream_t st) {
    uint32_t tile, ntiles;
    agg_sort_tiling(a0, tile, ntiles);
    const unsigned nb = (unsigned)a0.nw_long * ntiles;
    constexpr uint32_t SL = 8;                              // cooperating lanes per simulated sort workgroup (any partition of the tile's items is valid)
    auto pairs = [=](const AggArgs& a, bool scatter) {
        std::vector<uint32_t> bins(a.nbuckets);
        for (unsigned blk = 0; blk < nb; blk++) {
            uint32_t* mine = tiles + (size_t)blk * a.nbuckets;
            for (uint32_t e = 0; e < a.nbuckets; e++) bins[e] = scatter ? mine[e] : 0u;
            for (uint32_t tid = 0; tid < SL; tid++) { if (scatter) agg_tile_pairs<true>(a, blk / ntiles, blk % ntiles, tile, tid, SL, bins.data()); else agg_tile_pairs<false>(a, blk / ntiles, blk % ntiles, tile, tid, SL, bins.data()); }
            if (!scatter) std::copy(bins.begin(), bins.end(), mine);
        }
    };
    mockhip::launch(st, [=, a = a0] { pairs(a, false); });
    mockhip::launch(st, [=, a = a0] { for (uint32_t key = 0; key < a.nkeys; key++) agg_tile_totals(a, tiles, ntiles, key); });
    mockhip::launch(st, [=, a = a0] { for (uint32_t t = 0; t < kAggScanLanes; t++) agg_scan_phase0(a.count, a.nkeys + 1, t, kAggScanLanes, scanpart); });
    mockhip::launch(st, [=] {
        uint32_t top[kAggScanTop];
        for (uint32_t t = 0; t < (uint32_t)kAggScanTop; t++) agg_scan_phase0(scanpart, kAggScanLanes, t, kAggScanTop, top);
        agg_scan_mid(kAggScanTop, top);
        for (uint32_t t = 0; t < (uint32_t)kAggScanTop; t++) agg_scan_phase1(scanpart, kAggScanLanes, t, kAggScanTop, top);
    });
    mockhip::launch(st, [=, a = a0] { for (uint32_t t = 0; t < kAggScanLanes; t++) agg_scan_phase1(a.count, a.nkeys + 1, t, kAggScanLanes, scanpart); });
    mockhip::launch(st, [=, a = a0] { for (uint32_t key = 0; key < a.nkeys; key++) agg_tile_offsets(a, tiles, ntiles, key); });
    mockhip::launch(st, [=, a = a0] { pairs(a, true); });
}
void launch_agg_bucket_sum(const AggArgs& a0, uint32_t j0, uint32_t nwin, uint32_t* perm, uint32_t* hist, hipStream_t st) {
    const uint32_t k0 = j0 * a0.nbuckets, k1 = (j0 + nwin) * a0.nbuckets;
    constexpr uint32_t PL = 4;                              // cooperating lanes per simulated ordering workgroup
    auto pairs = [=](const AggArgs& a, bool scatter) {
        uint32_t bins[PLUME_AGG_LEN_BINS];
        for (uint32_t blk = 0; blk < (uint32_t)kAggPermBlocks; blk++) {
            uint32_t* mine = hist + (size_t)blk * PLUME_AGG_LEN_BINS;
            for (uint32_t e = 0; e < PLUME_AGG_LEN_BINS; e++) bins[e] = scatter ? mine[e] : 0u;
            for (uint32_t tid = 0; tid < PL; tid++) { if (scatter) agg_perm_pairs<true>(a, k0, k1, blk, kAggPermBlocks, tid, PL, bins, perm); else agg_perm_pairs<false>(a, k0, k1, blk, kAggPermBlocks, tid, PL, bins, perm); }
            if (!scatter) std::copy(bins, bins + PLUME_AGG_LEN_BINS, mine);
        }
    };
    mockhip::launch(st, [=, a = a0] { pairs(a, false); });
    mockhip::launch(st, [=] {
        uint32_t part[PLUME_AGG_LEN_BINS];
        for (uint32_t bin = 0; bin < PLUME_AGG_LEN_BINS; bin++) agg_perm_phase0(hist, kAggPermBlocks, bin, part);
        agg_perm_mid(part);
        for (uint32_t bin = 0; bin < PLUME_AGG_LEN_BINS; bin++) agg_perm_phase1(hist, kAggPermBlocks, bin, part);
    });
    mockhip::launch(st, [=, a = a0] { pairs(a, true); });
    mockhip::launch(st, [=, a = a0] { for (uint32_t lane = k0; lane < k1; lane++) agg_bucket_sum(a, perm[lane]); });
}
size_t agg_reduce_points(const AggArgs& a, uint32_t nwin) {
    const uint32_t chunk = a.nbuckets < PLUME_AGG_CHUNK ? a.nbuckets : PLUME_AGG_CHUNK;
    return (size_t)nwin * ((a.nbuckets + chunk - 1) / chunk);
}
int launch_agg_reduce(const AggArgs& a0, uint32_t j0, uint32_t nwin, uint32_t* red0, uint8_t* inf0, uint32_t* red1, uint8_t* inf1, hipStream_t st) {
    const uint32_t chunk = a0.nbuckets < PLUME_AGG_CHUNK ? a0.nbuckets : PLUME_AGG_CHUNK;
    uint32_t m = (a0.nbuckets + chunk - 1) / chunk;
    mockhip::launch(st, [=, a = a0] { for (uint32_t lane = 0; lane < nwin * m; lane++) agg_chunk_reduce(a, j0 + lane / m, lane % m, chunk, m, red0, inf0, j0, nwin); });
    int cur = 0;
    while (m > 1) {
        const uint32_t g = m < PLUME_AGG_GROUP ? m : PLUME_AGG_GROUP, mo = (m + g - 1) / g;
        const uint32_t* in = cur ? red1 : red0; const uint8_t* ininf = cur ? inf1 : inf0;
        uint32_t* out = cur ? red0 : red1; uint8_t* outinf = cur ? inf0 : inf1;
        mockhip::launch(st, [=] { for (uint32_t lane = 0; lane < nwin * mo; lane++) agg_group_sum(in, ininf, m, g, out, outinf, mo, nwin, lane / mo, lane % mo); });
        cur ^= 1;
        m = mo;
    }
    uint32_t* pts = cur ? red1 : red0; uint8_t* inf = cur ? inf1 : inf0;
    mockhip::launch(st, [=, a = a0] { for (uint32_t k = 0; k < nwin; k++) agg_window_shift(a, pts, inf, j0 + k, j0, nwin); });
    return cur;
}
size_t agg_scalar_sum_words(size_t n) { return 8 * ((n + PLUME_AGG_SUM_K - 1) / PLUME_AGG_SUM_K); }
const uint32_t* launch_agg_scalar_sum(const uint32_t* gs, size_t n, uint32_t* s0, uint32_t* s1, hipStream_t st) {
    const uint32_t* in = gs;
    size_t nin = n;
    int cur = 0;
    do {
        const size_t nout = (nin + PLUME_AGG_SUM_K - 1) / PLUME_AGG_SUM_K;
        uint32_t* out = cur ? s1 : s0;
        mockhip::launch(st, [=] { for (size_t lane = 0; lane < nout; lane++) agg_scalar_sum(in, nin, out, nout, lane); });
        in = out; nin = nout; cur ^= 1;
    } while (nin > 1);
    return in;
}
void launch_agg_gterm(const AggArgs& a0, const uint32_t* gsum, uint32_t* gout, uint8_t* goutinf, hipStream_t st) { mockhip::launch(st, [=, a = a0] { agg_gterm(a, gsum, gout, goutinf); }); }
void launch_agg_final(const AggArgs& a0, const uint32_t* lo, const uint8_t* loinf, uint32_t nlo, const uint32_t* hi, const uint8_t* hiinf, uint32_t nhi, const uint32_t* gpt, const uint8_t* gptinf,
                      const uint8_t* carry, hipStream_t st) {
    mockhip::launch(st, [=, a = a0] { agg_final(a, lo, loinf, nlo, hi, hiinf, nhi, gpt, gptinf, carry); });
}
void launch_agg_combine(const uint8_t* records, uint32_t m, uint8_t* result, hipStream_t st) { mockhip::launch(st, [=] { agg_combine(records, m, result); }); }

}  // namespace plume
