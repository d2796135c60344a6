/* plume_hip.h — C ABI of the MI355X batch PLUME sign/verify engine (libplume_hip.so).
 *
 * This is the drop-in boundary for the hot path of plume-sig/zk-nullifier-sig.  The reference has no FFI of
 * its own: its public Rust API is the boundary.  Each entry point below replaces, for a whole
 * batch at once, the Rust item named beside it; a `plume_rustcrypto`-compatible façade binds these with
 * `extern "C"` (INTEGRATION.md shows the stub).
 *
 * Data formats (all arrays are structure-of-arrays, caller-owned, never retained after return):
 *   point   64 bytes  affine x || y, big-endian; all-zero = the identity (k256 AffinePoint::IDENTITY)
 *   scalar  32 bytes  big-endian
 *   msgs    packed message bytes + (n+1) u64 offsets: message i = msgs[msg_off[i] .. msg_off[i+1])
 *   arrays of 32/64-byte records must be 4-byte aligned (16 recommended); msgs may have any alignment
 *
 * Return codes: 0 ok, -1 bad argument, -2 HIP runtime error (text via plume_last_error), -3 no usable GPU.
 * Nothing throws or unwinds across this boundary.  There is NO CPU fallback: without a gfx950 device
 * plume_init fails with -3.
 *
 * Threading: one plume_ctx is used by one host thread at a time; distinct contexts are independent and may be used
 * concurrently from different threads, also on the same GPU.  Multi-GPU, two ways: (a) one process holding all the
 * devices creates ONE context with plume_init_multi and passes whole batches to the host-pointer entry points, the
 * library shards them; (b) one process per GPU (torch.distributed / MPI ranks) each creates a plume_init context for
 * its device and passes its own contiguous shard.  The path has no cross-device exchange, so neither way involves a
 * collective.
 */
#ifndef PLUME_HIP_H
#define PLUME_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plume_ctx plume_ctx;

#define PLUME_OK 0
#define PLUME_ERR_ARG (-1)
#define PLUME_ERR_HIP (-2)
#define PLUME_ERR_NODEV (-3)

/* sign status bits (per item) */
#define PLUME_STATUS_C_NOT_CANONICAL 1 /* SHA-256 digest was 0 or >= n: k256's sign panics here (rust-k256/src/randomizedsigner.rs:90-91), \
                                          arkworks' reduces (rust-arkworks/src/lib.rs:257); c is emitted reduced mod n */
#define PLUME_STATUS_BAD_SCALAR 2      /* sk or r outside [1, n-1] (NonZeroScalar / SecretKey invariant), a supplied pk not on the curve, or \
                                          (device-resident calls) message offsets that decrease or reach past msgs_bytes */
#define PLUME_STATUS_IDENTITY 4        /* H == identity (randomizedsigner.rs:61) or s == 0 (randomizedsigner.rs:95) */

/* Create a context bound to HIP device `device_id` (>= 0): streams, events and a small scratch area; about 1.5 ms.  The generator's fixed tables are built by the first call
 * that needs them, ONCE per device and process -- (1..2^23)*G for the verifier's 24-bit windows (1 GiB, first verify / verify_non_zk call) and the signer's doubling-free
 * comb, 15 windows of 2^17 rows (252 MiB, first sign / SEC1-DER export / aggregate check) -- about 18 ms for both, synchronously inside that call.  They are read-only and
 * shared by every context of the process on that device; the last context to go frees them.  A verify-only process never holds the comb, a sign-only one never the 1 GiB table.
 * plume_destroy (and plume_set_in_flight when it takes lanes away) waits for the context's OWN work before it releases anything -- the last device-resident call it was
 * given, on whatever stream (every such call leaves an event behind its last kernel and waits for its predecessor's), and its private streams -- and does not call
 * hipDeviceSynchronize.  It then frees its workspace with hipFree, which the HIP runtime may itself implement with a device-wide wait: a caller that must not stall
 * other streams should destroy contexts at a quiet point.  A caller-provided stream must stay alive until the context's last call on it has finished. */
int plume_init(plume_ctx** out, int device_id);
/* Create a multi-device context: one shard (a complete single-device context with its own streams and workspace, driven by its own
 * worker thread) per entry of device_ids (SURVEY.md §8b sketch, §8e).  Every HOST-POINTER entry point below then splits its batch
 * evenly and contiguously -- shard d of g gets items [floor(d*n/g), floor((d+1)*n/g)) of every array -- runs the shards concurrently
 * and writes each shard's results into its slice of the caller's output arrays; there is no cross-device exchange and no collective.
 * A device id may repeat (two shards on one GPU overlap one's copies with the other's kernels).  The *_device entry points need
 * a single-device context (device pointers belong to one GPU) and return PLUME_ERR_ARG on a multi-device one;
 * plume_nullifier_first_occurrence runs on the first shard (every record has to meet every other). */
int plume_init_multi(plume_ctx** out, const int* device_ids, int n_devices);
/* 1 for plume_init contexts, n_devices for plume_init_multi contexts */
int plume_num_shards(const plume_ctx* ctx);
/* Each shard's worker thread (it issues all copies between the caller's arrays and its GPU) is bound to the CPUs of the NUMA node its GPU hangs off, found through the
 * device's PCI address in sysfs and cut to the CPUs the process may use; where that cannot be determined the thread is left alone (PLUME_NO_AFFINITY=1 switches the
 * binding off).  Returns the node shard `shard` was bound to, or -1.  Callers should allocate / page-lock their arrays on the same nodes. */
int plume_shard_numa_node(const plume_ctx* ctx, int shard);
void plume_destroy(plume_ctx* ctx);
/* Last error text of this thread (valid until the next failing call on the thread). */
const char* plume_last_error(void);
/* Library / build information: "plume_hip <major.minor> gfx950 build=<hash of the device sources>".  0.5 (round 5): plume_set_stage_timing, stage events off by default; 0.4 (round 5): plume_get_sign_uniform, plume_set_host_lanes, plume_set_eq1_short;
 * the signer defaults to uniform level 1; the generator tables are built by the first call that needs them; stream = NULL means the stream of the context the caller
 * holds; plume_destroy waits for the context's own work only (its last call on any stream and its private streams), not for the whole device. */
const char* plume_version(void);
/* Upper bound on items processed per internal pass (workspace is ~3.9 KB per in-flight item). Default 1<<20. */
int plume_set_chunk(plume_ctx* ctx, size_t max_items_per_pass);
/* Device-resident verify / sign calls of >= 2^17 items can be cut into `sub_batches` slices: validation + hash_to_curve and the window tables of
 * slice k+1 then run on a second stream of the context beside the multi-scalar kernel of slice k.  Results do not depend on it.  Default 1 =
 * strictly serial launch order on the caller's stream, which is also the mode in which plume_last_stage_times reports one time per kernel:
 * on the MI355X the overlapped order measured 1-3 % SLOWER than the serial one (round 3, LABNOTES.md §6: kernels of two streams sharing the
 * compute units cost more than the table kernel's idle issue slots give back), so the knob is an experiment's record, not a recommendation.
 * Env PLUME_SUB_BATCHES=k sets the default of new contexts, PLUME_SERIAL=1 forces 1 (and wins when both are set). */
int plume_set_sub_batches(plume_ctx* ctx, int sub_batches);
/* Batches in flight (default 1).  A context runs its device-resident calls one at a time: they share its workspace, so calls issued on different streams queue.  With
 * batches = 2 the calls go in turn to two lanes of the context (each with a workspace, streams and events of its own; the generator's fixed tables are shared), and calls the
 * caller issues on DIFFERENT streams can run side by side: every kernel of a 2^20 batch fills the chip, yet the memory-bound table passes and the ramps / tails of one batch's
 * kernels fit beside the issue-bound multi-scalar kernel of the other (about 1 % per batch), and SMALL calls leave the chip mostly empty: two of them side by side take much less
 * than one after the other -- on the MI355X, two lanes against one (profiles/r06_hw_queues_small_calls.txt): 2^10-item verifies -49 % per call, 2^12 -31 %, 2^14 -26 %,
 * 2^16 -13.6 % (1.17 instead of 1.36 ms per call), 2^17 -6.5 %, 2^18 -2.3 %; 2^20 signs -3 %; a third lane gains nothing more (1..4 accepted).
 * THE CALLER'S STREAMS MUST SIT ON DIFFERENT HARDWARE QUEUES for any of this: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware
 * queues per priority level, and streams that share a queue run their kernels one after the other (round 6's kernel trace of two torch streams: every kernel of both on one
 * queue, no gain at any size).  Set GPU_MAX_HW_QUEUES=8 in the environment of the process before its first HIP call (the Python package and bench.py do; the figures above are
 * with it).  Do NOT give the two callers' streams different priorities instead: the command processor runs a high-priority queue's kernels alone.
 * Results do not depend on any of it; calls on one stream keep that stream's order -- including NULL, which always means the stream of the context the caller holds,
 * whichever lane serves the call (so a sign followed by a verify of its outputs, both with stream = NULL, stay ordered).  Costs a second per-batch workspace.  Env
 * PLUME_IN_FLIGHT_MIN=<items> keeps smaller calls on the first lane (default 0: none).  Single-device contexts only (a multi-device context
 * already runs its shards side by side).  plume_last_stage_times / plume_last_redo_tasks then report the lane of the last device-resident call. */
int plume_set_in_flight(plume_ctx* ctx, int batches);
/* The signer's schedule.  k256's scalar multiplication is constant-time (call sites rust-k256/src/randomizedsigner.rs:51-70), so the DEFAULT here is level 1
 * since round 5 (library 0.4; rounds 1-4 defaulted to level 0).  Env PLUME_SIGN_UNIFORM=<level> sets the default of new contexts; plume_set_sign_uniform(ctx, 0) opts out.
 *   level 0: fastest.  NOT uniform: zero digits of sk and r are skipped, digit signs are branches, table rows are gathered at digit-dependent addresses -- the instruction
 *            trace and the memory addresses of k_sign_gmul / k_sign_hmul depend on the secrets.
 *   level 1: (default) the two kernels that walk those digits (sk*G, r*G by the comb; sk*H, r*H by windows; also the comb of the SEC1-DER export, whose scalars are secret
 *            keys) run the same instructions for every digit value: every slot adds (a zero digit adds row 1 to a copy that a masked select drops), signs are masked selects,
 *            the accumulator starts at a fixed offset point that is subtracted at the end.  UNIFORM: control flow, instruction count.  NOT uniform: the ADDRESS of the
 *            table row each slot gathers (cache / HBM-channel timing).  Price on the MI355X: +2.5 % per signature.
 *   level 2: level 1, and no address is derived from a digit: every slot reads all 3 rows of its table (P, theta P, 2P: csrc/plume_ec.h) and keeps one by masked selects (what k256 does with its
 *            16-entry tables); the multiplications by G then use a 52-window x 16-row table instead of the 18-bit comb (52 additions instead of 15).  UNIFORM: control flow,
 *            instruction count, every memory address.  Price: +47 %.
 * At every level: the scalar side (reduction mod n, GLV split, digit recoding, s = r + sk*c) is select-based; the range checks of sk and r are not (an out-of-range scalar
 * is a status bit, not a secret worth protecting); nothing is claimed about power / EM channels.  Outputs are bit-identical at every level.  Returns PLUME_ERR_ARG for a
 * level outside 0..2.  Measurements: DESIGN.md §9. */
int plume_set_sign_uniform(plume_ctx* ctx, int level);
/* the level this context signs at (0, 1, 2), or a negative error code */
int plume_get_sign_uniform(const plume_ctx* ctx);
/* The verifier's first equation, R' = s G - c pk compared with the given r_point (rust-k256/src/lib.rs:101,115-121), for calls that GIVE r_point as a 64-byte record (V1
 * verify, verify_non_zk).  It is an identity check, so it may be multiplied by any tau != 0: with (tau, upsilon) from a half-GCD of c in the Eisenstein integers
 * (tau c = upsilon mod n through lambda; all four coefficients of about 64 bits -- csrc/plume_eis.h) the GPU checks  k G - upsilon pk - (tau - 1) R == R,  k = tau s mod n:
 * 66 doublings instead of 128 on that equation, the generator's term from the signer's comb, one more window table per item (R).  Verdicts are identical by construction:
 * the equivalence is exact, not probabilistic, and the pair is re-checked per item (tau c == upsilon mod n) before it is used -- a failure, which cannot happen, would send
 * the item to the long form.
 *   mode 1 (default): short form for calls of at least PLUME_EQ1_SHORT_MIN items (plume_get_eq1_short reports the threshold)
 *   mode 3: short form for calls of any size          mode 0: long form always
 *   mode 2: test mode -- every item takes the scalar stage's fallback, i.e. the long form inside the checked chain of the redo launch
 * Env PLUME_EQ1_SHORT, PLUME_EQ1_SHORT_MIN set the defaults of new contexts.  Measured on the MI355X, 2^20 V1 verifies (round 5, three-row tables): k_verify_msm -7 %,
 * the step -2.5 % (the fourth table and the half-GCD take the rest back: DESIGN.md). */
int plume_set_eq1_short(plume_ctx* ctx, int mode);
/* the mode in force (0..3, or a negative error code); min_items, when not NULL, receives the smallest call (items) that takes the short form in mode 1.  The rule looks at
 * the CALL's n, also when plume_set_sub_batches cuts the call into slices. */
int plume_get_eq1_short(const plume_ctx* ctx, size_t* min_items);
/* Environment knobs read when a context is created (tuning and A/B runs; results never depend on them):
 *   PLUME_SUB_BATCHES, PLUME_SERIAL, PLUME_OVERLAP_MIN   sub-batch overlap of the device-resident calls (plume_set_sub_batches)
 *   PLUME_HOST_PIECE, PLUME_HOST_FIRST_PIECE, PLUME_HOST_TAIL_PIECE, PLUME_HOST_REGISTER_MIN, PLUME_HOST_LANES (1 | 2), PLUME_HOST_SCHEDULE (explicit piece list, read per call)
 *                                                        the host-pointer pipeline (plume_set_host_*)
 *   PLUME_INGEST_SPLIT_MAX   verify calls of at most this many items run the ingest stage with two lanes per item (default 65536; 0: never)
 *   PLUME_MSM_PAIR_MAX       verify calls of at most this many items run every long-form chain of the multi-scalar stage as two half chains on two lanes, joined by one checked
 *                            addition (default 16384; 0: never): on a machine a small call leaves empty, the kernel's time is one chain's latency
 *   PLUME_JOBS_PER_LANE      jobs per lane of the table passes (default: 3..6 by batch size)
 *   PLUME_SIGN_UNIFORM       default level of plume_set_sign_uniform (0, 1, 2; default 1)
 *   PLUME_EQ1_SHORT, PLUME_EQ1_SHORT_MIN   plume_set_eq1_short's mode and the smallest call that takes the short form (default 1, 65536)
 *   PLUME_NO_AFFINITY        multi-device contexts: leave the shard threads' CPU affinity alone */
/* Host-pointer calls only: a call is cut into pieces; piece k+1 uploads while piece k computes and piece k-1 downloads (an upload, a download and the compute streams,
 * four staging slots), so only the first upload and the last download are exposed.  The first piece is small (default 1<<16 items), each following piece up to three times
 * the one before, up to the largest piece (default 1<<19, capped by the chunk size); calls with large outputs (the signer: 320 bytes down per item) taper instead -- a first
 * piece of twice the tail piece (default 1<<16), a body of pieces of at most half the largest, then 3, 2 and 1 tail pieces -- so that the last, unhidden download is short.
 * The pieces of a VERIFY call alternate between the context and a second lane of its own (workspace + stream, created on the first such call) when EVERY array of the call
 * is page-locked.  The signer under the same condition (round 6, now that two lanes' kernels really run side by side -- plume_set_in_flight's note on hardware queues): uniform
 * pieces of the tail size dealt to the two lanes in turn (twice the tail size beyond 32 of them); 2^18..2^20 signs gain 2-7 %, 2^22 0.7 % over the tapered one-lane
 * schedule, which pageable callers and PLUME_HOST_SIGN_LANES=1 still get.  The copy streams are created at high priority: the runtime multiplexes a process's streams onto
 * a few hardware queues per priority level, and a copy stream that shares a queue with a compute stream waits for its kernels (round 5: a finished piece's download started
 * 6 ms late).  Env PLUME_HOST_TRACE=1 prints every call's timeline to stderr (per piece: upload, kernels, download on the GPU's clock; no profiler needed -- and none should be
 * attached: rocprofv3's memory-copy trace turns the downloads into blit kernels).  2^20 items from page-locked arrays on the MI355X: verify 19.3-20.5 ms, sign 17.9-18.7 ms
 * over the boxes met (round 5: 20.4-21.1 and 18.7-19.1), 0.89-0.94 of the device-resident rates.  Results do not depend on any of this. */
int plume_set_host_piece(plume_ctx* ctx, size_t largest_piece_items);
int plume_set_host_first_piece(plume_ctx* ctx, size_t items);
int plume_set_host_tail_piece(plume_ctx* ctx, size_t items);
/* 1 = every piece of a host-pointer call runs on the context itself, 2 (default) = pieces alternate between the context and a second lane (env PLUME_HOST_LANES) */
int plume_set_host_lanes(plume_ctx* ctx, int lanes);
/* Host-pointer calls: caller arrays of at least `bytes` bytes that are not page-locked yet are registered (hipHostRegister) for the
 * duration of the call; 0 (default) = never.  Registration costs more than one batch's copies save: callers that reuse their buffers
 * should page-lock them ONCE with the helpers below instead. */
int plume_set_host_register_min(plume_ctx* ctx, size_t bytes);

/* ---- page-locked host memory ----------------------------------------------------------------------------------
 * The host-pointer entry points accept any memory.  When the caller's arrays are page-locked the GPU's copy engines read and
 * write them directly and every transfer overlaps the kernels of the neighbouring pieces; pageable arrays go through the
 * runtime's staging copies and block the calling thread.  plume_host_alloc returns page-locked memory (NULL on failure);
 * plume_host_register page-locks memory the caller already owns (e.g. a Rust Vec's buffer) until plume_host_unregister. */
void* plume_host_alloc(size_t bytes);
void plume_host_free(void* p);
int plume_host_register(void* p, size_t bytes);
int plume_host_unregister(void* p);

/* ---- PlumeSignature::verify, batched  (rust-k256/src/lib.rs:93-145) -------------------------------------
 * version 1: V1 (v1specific = Some{r_point, hashed_to_curve_r}); version 2: V2 (r_point, hashed_to_curve_r NULL).
 * ok[i] = 1 iff the reference's verify() returns true for item i, else 0.  Inputs the Rust types cannot even
 * represent (c or s outside [1, n-1], coordinates >= p, points off the curve) give ok = 0.
 * Host-pointer form: copies in, runs, copies ok[] out, returns when done. */
int plume_verify_batch(plume_ctx* ctx, int version, size_t n,
                       const uint8_t* msgs, const uint64_t* msg_off,
                       const uint8_t* pk, const uint8_t* nullifier, const uint8_t* c, const uint8_t* s,
                       const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                       uint8_t* ok);

/* ---- plume_arkworks' verify_non_zk, batched  (rust-arkworks/src/tests.rs:28-78; the verification the arkworks crate's tests and the
 * circuit's witness checks use) -------------------------------------------------------------------------------------------------
 * Inputs as the reference takes them: pk, message, PlumeSignaturePublic{s, nullifier}, PlumeSignaturePrivate{hashed_to_curve_r,
 * r_point, digest_private}.  Differences from plume_verify_batch: the challenge c' = SHA256(..) mod n is hashed from the GIVEN
 * r_point / hashed_to_curve_r (6 encodings for V1, 3 for V2: compute_c_v1 / compute_c_v2, rust-arkworks/src/lib.rs:120-163); BOTH
 * equations s*G - digest_private*pk == r_point and s*H - digest_private*nullifier == hashed_to_curve_r are checked for V1 AND V2;
 * s and digest_private are Fr elements, so zero is a value (>= n cannot be represented: ok = 0).
 * ok[i] = 1 Ok(true), 0 Ok(false), 2 Err(HashToCurveError) (pk is the identity, rust-arkworks/src/lib.rs:99-101). */
int plume_verify_non_zk_batch(plume_ctx* ctx, int version, size_t n,
                              const uint8_t* msgs, const uint64_t* msg_off,
                              const uint8_t* pk, const uint8_t* nullifier, const uint8_t* s,
                              const uint8_t* r_point, const uint8_t* hashed_to_curve_r, const uint8_t* digest_private,
                              uint8_t* ok);

/* ---- verify with SEC1-compressed points (33-byte records) --------------------------------------------------
 * The wire format of the reference's serde / wasm layer (javascript/src/lib.rs:95-118,147-184; sec1_affine,
 * rust-arkworks/src/lib.rs:76-88): 02|03 || x big-endian, or a record whose first byte is 00 for the identity (the
 * remaining 32 bytes are ignored).  Decompression (y = sqrt(x^3 + 7), parity from the tag) and on-curve validation run
 * on the GPU; a record that would fail to deserialize in the reference (other tag, x >= p, x not on the curve) gives
 * ok = 0.  Everything else as plume_verify_batch. */
int plume_verify_batch_sec1(plume_ctx* ctx, int version, size_t n,
                            const uint8_t* msgs, const uint64_t* msg_off,
                            const uint8_t* pk33, const uint8_t* nullifier33, const uint8_t* c, const uint8_t* s,
                            const uint8_t* r_point33, const uint8_t* hashed_to_curve_r33,
                            uint8_t* ok);

/* ---- PlumeSigner::try_sign_with_rng / PlumeSignature::sign_v1|sign_v2, batched, nonce supplied -----------
 * (rust-k256/src/randomizedsigner.rs:43-112, rust-k256/src/lib.rs:149-156; the RNG stays on the host: r[i] is
 * the 32 bytes the reference would draw, cf. the mock RNG in rust-k256/tests/signing.rs:23-44).
 * pk_in == NULL : pk = sk*G is derived (plume_rustcrypto shape).
 * pk_in != NULL : plume_arkworks::sign_with_r shape (rust-arkworks/src/lib.rs:229-278): pk supplied, not recomputed.
 * Outputs: pk (may be NULL), nullifier, c (= digest mod n), s, r_point, hashed_to_curve_r (always written, V1 and
 * V2), status[i] bit set as above.  version selects the c-hash (V1: G,pk,H,nul,R,Hr; V2: nul,R,Hr). */
int plume_sign_batch(plume_ctx* ctx, int version, size_t n,
                     const uint8_t* msgs, const uint64_t* msg_off,
                     const uint8_t* sk, const uint8_t* r, const uint8_t* pk_in,
                     uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s,
                     uint8_t* r_point, uint8_t* hashed_to_curve_r, uint8_t* status);

/* ---- hash_to_curve(m, pk), batched  (rust-k256/src/utils.rs:11-20) ----------------------------------------
 * h_out[i] = h2c(msg_i || SEC1c(pk_i)) with DST rust-k256/src/lib.rs:61.  pk == NULL hashes the raw message
 * bytes (Secp256k1::hash_from_bytes(&[msg], &[DST]); rust-k256/tests/verification.rs:148-156) for KAT pinning. */
int plume_hash_to_curve_batch(plume_ctx* ctx, size_t n,
                              const uint8_t* msgs, const uint64_t* msg_off,
                              const uint8_t* pk, uint8_t* h_out);

/* ---- circuit witness hints from the GPU hash_to_curve   -----------------------------------------------
 * The circom verifier (circuits/circom/verify_nullifier.circom:14-31,152-162) takes the hash_to_curve intermediates as
 * precomputed inputs; the GPU computes them anyway.  Per item, for h2c(msg_i || SEC1c(pk_i)) (pk == NULL: the raw message bytes):
 *   u       n x 64  : u0 | u1                                     hash_to_field (RFC 9380 §5.2)
 *   mapped  n x 128 : q0_x_mapped | q0_y_mapped | q1_x_mapped | q1_y_mapped     simplified-SWU outputs, affine, on the isogenous curve E'
 *   q       n x 128 : Q0.x | Q0.y | Q1.x | Q1.y                   after the 3-isogeny, on secp256k1 (identity = zeros)
 *   h       n x 64  : H = Q0 + Q1
 * Any output may be NULL.  registers = 0: every value is 32 big-endian bytes; registers = 1: every value is the circuit's 4 x 64-bit
 * little-endian registers (circuits/circom/utils.ts:11-17 scalarToCircuitValue), i.e. out + 32*k is a uint64_t[4].
 * q*_gx1_sqrt, q*_gx2_sqrt, q*_y_pos come from plume_h2c_hints_batch below (UNPINNED definitions).
 * An invalid pk (or malformed message offsets) zeroes the item's outputs. */
int plume_h2c_intermediates_batch(plume_ctx* ctx, size_t n,
                                  const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk,
                                  int registers, uint8_t* u, uint8_t* mapped, uint8_t* q, uint8_t* h);
/* The remaining hash_to_curve inputs of the circuit: q{0,1}_gx1_sqrt, q{0,1}_gx2_sqrt, q{0,1}_y_pos (verify_nullifier.circom:21-23,27-29).
 * UNPINNED: the reference obtains them from secp256k1_hash_to_curve_circom's ts/generate_inputs (circuits/circom/test/v1.test.ts:5,38-40), a package that is not in
 * the reference tree, and holds no vector of them -- so nothing here can be checked against the reference.  They are DEFINED as follows from RFC 9380 F.2 / F.2.1.2,
 * for each of the two maps k = 0, 1 with u = u_k, Z = -11, E': y^2 = x^3 + A'x + B':
 *     x1 = (-B'/A')(1 + 1/(Z^2 u^4 + Z u^2))   (B'/(Z A') when the denominator vanishes),   gx1 = x1^3 + A' x1 + B'
 *     x2 = Z u^2 x1,                                                                        gx2 = x2^3 + A' x2 + B'  ( = (Z u^2)^3 gx1: exactly one of gx1, gx2 is a square)
 *     qk_gx1_sqrt = the EVEN root r (r mod 2 = 0, 0 <= r < p) of  r^2 = gx1  if gx1 is a square,  of  r^2 = Z gx1  if it is not
 *     qk_gx2_sqrt = the EVEN root of  r^2 = gx2  if gx2 is a square,  of  r^2 = Z gx2  if it is not
 *                   (the second form is sqrt_ratio's return value for a non-residue, RFC 9380 F.2.1.2: a witness that gx is NOT a square, Z being a non-residue)
 *     qk_y_pos    = the root y of  y^2 = (gx1 if it is a square, else gx2)  with sgn0(y) = sgn0(u): the y the map returns, i.e. qk_y_mapped
 * A consumer whose generator fixes these choices differently (the other root; 1 in place of the non-residue witness) derives its values from these with a negation /
 * a constant.  hints: n x 192 bytes, q0_gx1_sqrt | q0_gx2_sqrt | q0_y_pos | q1_gx1_sqrt | q1_gx2_sqrt | q1_y_pos, each 32 bytes (registers as above).
 * tests: the algebraic definitions above against the Python oracle's big-integer restatement (oracle/plume_oracle.py h2c_hints), incl. RFC 9380 J.8.1's messages. */
int plume_h2c_hints_batch(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, int registers, uint8_t* hints);
/* 32-byte big-endian values (c, s, pk.x, pk.y, nullifier.x, ... of a signature) -> 4 x 64-bit little-endian registers each
 * (circuits/circom/utils.ts:11-17, verify_nullifier.circom:380-385).  Host memory; no context needed. */
int plume_registers_from_be(size_t nvalues, const uint8_t* be32, uint64_t* registers);

/* ---- SEC1-DER scalar marshalling  (the wasm layer's wire format for `s` and `digest_private`) --------------
 * der109[i] = SecretKey::from(scalar_i).to_sec1_der() (javascript/src/lib.rs:98-110): the 109-byte RFC 5915 ECPrivateKey
 *   30 6b 02 01 01 04 20 <scalar> a1 44 03 42 00 04 <x> <y>      with (x, y) = scalar * G
 * The generator multiplications run on the GPU (doubling-free comb).  status[i] = 0, or PLUME_STATUS_BAD_SCALAR with an all-zero
 * record when the scalar is outside [1, n-1] (no SecretKey holds it). */
int plume_scalars_to_sec1_der_batch(plume_ctx* ctx, size_t n, const uint8_t* scalars, uint8_t* der109, uint8_t* status);
/* SecretKey::from_sec1_der for that fixed form, with the reference's semantics (elliptic-curve's TryFrom<EcPrivateKey> validates the embedded public key): ok[i] = 1
 * iff record i has exactly this structure, a scalar in [1, n-1], AND a public-key field equal to scalar * G (recomputed on the GPU by the comb and compared byte for
 * byte); scalars[i] = the 32-byte scalar then, zeros otherwise. */
int plume_sec1_der_to_scalars_checked(plume_ctx* ctx, size_t n, const uint8_t* der109, uint8_t* scalars, uint8_t* ok);
/* The structure half alone (host memory, no context, no GPU): shape + scalar range.  The public-key field is NOT checked against the scalar, so a record with a
 * tampered public key passes here where the reference returns Err -- use the _checked form wherever the record comes from outside. */
int plume_sec1_der_to_scalars(size_t n, const uint8_t* der109, uint8_t* scalars, uint8_t* ok);

/* plume_sign_batch with the point outputs as 33-byte SEC1-compressed records (02|03 || x; identity = 00 followed by 32 zero
 * bytes) -- the wire format of the reference's serde / wasm layer (javascript/src/lib.rs:95-118).  pk_in stays 64-byte affine. */
int plume_sign_batch_sec1(plume_ctx* ctx, int version, size_t n,
                          const uint8_t* msgs, const uint64_t* msg_off,
                          const uint8_t* sk, const uint8_t* r, const uint8_t* pk_in,
                          uint8_t* pk33, uint8_t* nullifier33, uint8_t* c, uint8_t* s,
                          uint8_t* r_point33, uint8_t* hashed_to_curve_r33, uint8_t* status);

/* ---- nullifier-set post-processing: first occurrences   --------------------------------
 * PLUME exists so that an application can accept ONE nullifier per (pk, message) (reference README.md:5; the field
 * rust-k256/src/lib.rs:72-73); after verifying a batch the application has to find repeated nullifiers.
 *   nullifier : n x 64 B (x||y big-endian, all-zero = identity), as passed to / produced by the calls above
 *   live      : optional n bytes; 0 = the item takes no part (e.g. ok[i] == 0).  NULL = all items
 *   ids       : optional n distinct 64-bit ids deciding which of several equal nullifiers is "first" (smallest id); NULL = the
 *               position i.  Used when the arrays are one shard of a larger set (multi-GPU exchange)
 *   first     : out, n bytes: 1 iff the item is live and no live item with the same 64-byte nullifier has a smaller id
 *   n_unique  : out, optional: number of 1s in `first`
 * Deterministic whatever the execution order.  n <= 2^30. */
int plume_nullifier_first_occurrence(plume_ctx* ctx, size_t n, const uint8_t* nullifier, const uint8_t* live,
                                     const uint64_t* ids, uint8_t* first, uint64_t* n_unique);

/* ---- aggregate pre-filter: one random-linear-combination check of the whole batch   ---------
 * DIFFERENT SEMANTICS from the verify calls: all-or-nothing and probabilistic -- a fast way to learn that EVERY item of a batch would verify,
 * never a replacement for the per-item `ok`.  It applies where r_point and hashed_to_curve_r are GIVEN: mode 0 = PlumeSignature::verify of V1
 * signatures (rust-k256/src/lib.rs:93-135; version must be 1), mode 1 = verify_non_zk, V1 or V2 (rust-arkworks/src/tests.rs:28-78; `c` is
 * digest_private).  Per item the inputs are validated and the challenge hash is checked EXACTLY (hash_ok); the two group equations
 * s*G - c*pk == r_point and s*H - c*nullifier == hashed_to_curve_r are checked only in aggregate:
 *     A = sum_i a_i (s_i G - c_i pk_i - R_i) + b_i (s_i H_i - c_i nul_i - Hr_i) == identity,
 * one multi-scalar multiplication over 5n points (bucket method), with 127-bit coefficients a_i | b_i = SHA256(seed || be64(i)).  The
 * producer of the batch must not be able to predict `seed` (draw 32 fresh random bytes per call): a batch holding a false equation then
 * passes with probability <= 2^-126.  When the check fails, run the per-item verify to find the culprits.
 *   seed    : 32 bytes, HOST pointer in both forms; NULL = the library draws 32 bytes from the OS generator for this call (callers without fresh randomness of
 *             their own should pass NULL rather than anything constant, reused or known to the signers: the check is only as sound as the seed is unpredictable)
 *   hash_ok : out, optional, n bytes: 1 = the item's inputs are values of the reference's types and c equals the hash
 *   result  : out, PLUME_AGG_RESULT_BYTES = 72 bytes:
 *               [0] all_ok (n_bad == 0 and A is the identity)   [1] A is the identity   [2..4) zero   [4..8) n_bad, u32 little-endian
 *               (items with hash_ok == 0; they take no part in A when their inputs are not representable)
 *               [8..72) A as an affine point (x||y big-endian, zeros = identity) -- a pure function of the inputs and the seed, whatever
 *               the piece / shard cut: tests compare it with the oracle's
 * The multi-device context shards the batch; the shards' sums are added on the first shard's GPU. */
#define PLUME_AGG_RESULT_BYTES 72
int plume_aggregate_check(plume_ctx* ctx, int version, int mode, size_t n,
                          const uint8_t* msgs, const uint64_t* msg_off,
                          const uint8_t* pk, const uint8_t* nullifier, const uint8_t* c, const uint8_t* s,
                          const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                          const uint8_t seed[32], uint8_t* hash_ok, uint8_t* result);

/* ---- device-resident forms -------------------------------------------------------------------------------
 * Same semantics, but every data pointer is a DEVICE pointer on the context's GPU and the work is enqueued on
 * `stream` (a hipStream_t passed as void*; NULL = the context's own stream) without synchronising: the caller
 * orders and waits (e.g. torch.cuda streams/events).  n must not exceed the chunk size (plume_set_chunk).
 * msgs_bytes = the size of the msgs buffer: the offsets live on the device, so the kernels check them -- an item whose offsets
 * decrease or reach past msgs_bytes is rejected (verify: ok = 0; sign: PLUME_STATUS_BAD_SCALAR; hash_to_curve: identity) and its
 * lane never reads msgs.
 * Streams: the calls of one context share its workspace.  Each call first makes its stream wait (hipStreamWaitEvent) for the
 * previous call's last kernel, so calls issued on DIFFERENT streams of one context are safe (they serialise on the workspace);
 * outputs are ready when the stream the call was issued on reaches the end of the call.  One host thread per context at a time. */
int plume_verify_batch_device(plume_ctx* ctx, int version, size_t n,
                              const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                              const uint8_t* pk, const uint8_t* nullifier, const uint8_t* c, const uint8_t* s,
                              const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                              uint8_t* ok, void* stream);
int plume_verify_non_zk_batch_device(plume_ctx* ctx, int version, size_t n,
                                     const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                     const uint8_t* pk, const uint8_t* nullifier, const uint8_t* s,
                                     const uint8_t* r_point, const uint8_t* hashed_to_curve_r, const uint8_t* digest_private,
                                     uint8_t* ok, void* stream);
int plume_verify_batch_sec1_device(plume_ctx* ctx, int version, size_t n,
                                   const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                   const uint8_t* pk33, const uint8_t* nullifier33, const uint8_t* c, const uint8_t* s,
                                   const uint8_t* r_point33, const uint8_t* hashed_to_curve_r33,
                                   uint8_t* ok, void* stream);
int plume_sign_batch_device(plume_ctx* ctx, int version, size_t n,
                            const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                            const uint8_t* sk, const uint8_t* r, const uint8_t* pk_in,
                            uint8_t* pk, uint8_t* nullifier, uint8_t* c, uint8_t* s,
                            uint8_t* r_point, uint8_t* hashed_to_curve_r, uint8_t* status, void* stream);
int plume_hash_to_curve_batch_device(plume_ctx* ctx, size_t n,
                                     const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                     const uint8_t* pk, uint8_t* h_out, void* stream);
int plume_h2c_intermediates_batch_device(plume_ctx* ctx, size_t n,
                                         const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk,
                                         int registers, uint8_t* u, uint8_t* mapped, uint8_t* q, uint8_t* h, void* stream);
int plume_h2c_hints_batch_device(plume_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes, const uint8_t* pk, int registers,
                                 uint8_t* hints, void* stream);
int plume_scalars_to_sec1_der_batch_device(plume_ctx* ctx, size_t n, const uint8_t* scalars, uint8_t* der109, uint8_t* status, void* stream);
int plume_registers_from_be_device(plume_ctx* ctx, size_t nvalues, const uint8_t* be32, uint64_t* registers, void* stream);
int plume_sign_batch_sec1_device(plume_ctx* ctx, int version, size_t n,
                                 const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                 const uint8_t* sk, const uint8_t* r, const uint8_t* pk_in,
                                 uint8_t* pk33, uint8_t* nullifier33, uint8_t* c, uint8_t* s,
                                 uint8_t* r_point33, uint8_t* hashed_to_curve_r33, uint8_t* status, void* stream);
/* n_unique, when not NULL, is a DEVICE pointer to one uint64_t */
int plume_nullifier_first_occurrence_device(plume_ctx* ctx, size_t n, const uint8_t* nullifier, const uint8_t* live,
                                            const uint64_t* ids, uint8_t* first, uint64_t* n_unique, void* stream);

/* index_base: coefficient index of item 0 (pieces of one batch checked by separate calls must use disjoint index ranges); hash_ok (optional) and
 * result are DEVICE pointers, seed is a HOST pointer */
int plume_aggregate_check_device(plume_ctx* ctx, int version, int mode, size_t n,
                                 const uint8_t* msgs, const uint64_t* msg_off, size_t msgs_bytes,
                                 const uint8_t* pk, const uint8_t* nullifier, const uint8_t* c, const uint8_t* s,
                                 const uint8_t* r_point, const uint8_t* hashed_to_curve_r,
                                 const uint8_t seed[32], uint64_t index_base, uint8_t* hash_ok, uint8_t* result, void* stream);

/* ---- measurement hooks (bench.py) -------------------------------------------------------------------------
 * Per-stage device time of the most recent *_device call on this context, measured with HIP events recorded on
 * the stream the kernels were launched on.  Fills up to `cap` entries: names[i] (static strings) and ms[i];
 * returns the number of stages, or a negative error.  Synchronises the recorded events. */
int plume_last_stage_times(plume_ctx* ctx, const char** names, float* ms, int cap);
/* The events behind plume_last_stage_times are recorded only on request (library 0.5; always before): a timing event between two kernels of a stream leaves the GPU idle
 * for about 6 us (rocprofv3 kernel trace of back-to-back 2^16-item verifies: 6.1 +- 0.2 us at each of the five stage boundaries, 0.0-0.2 us between kernels with no event
 * in between) -- 2 % of a 2^16-item call, 0.2 % of a 2^20-item one.  on = 1 before the call to be timed; env PLUME_STAGE_TIMES=1 sets the default of new contexts.  With
 * timing off plume_last_stage_times returns PLUME_ERR_ARG. */
int plume_set_stage_timing(plume_ctx* ctx, int on);
/* Measurement / test hook: the number of multi-scalar tasks (two per item) of the last verify call on this context whose unchecked addition chain met p == +-q and that
 * the second, dense launch redid with checked additions (k_verify_msm_redo).  Honest batches: 0.  Crafted items (pk = +-k G for small k with s = +-c, ...) file one or two
 * tasks each: that is all they cost -- their wavefront neighbours no longer wait for them.  Counts the last device-resident call (for a host-pointer call: its last piece;
 * for a multi-device context: the first shard).  Synchronises with the device. */
int plume_last_redo_tasks(plume_ctx* ctx, uint64_t* count);
/* Measurement hook: the multi-scalar kernel the last verify call served by this context launched -- "k_verify_msm" (both equations in the long form), "k_verify_msm_s"
 * (equation 1 in the short form) or "k_verify_msm_pair" (half chains: small calls); NULL before the first verify.  Reports the same call as plume_last_stage_times.  A static string. */
const char* plume_last_msm_kernel(const plume_ctx* ctx);
/* Measurement hook: the shader clock (GHz) the multi-scalar kernel of the last verify call on this context ran at, sampled INSIDE that kernel -- one workgroup in 32 adds the
 * shader-clock cycles and the constant-rate wall-clock ticks it lived for to two counters.  kernel time x this clock = the kernel's duration in cycles, a figure that does not
 * move with the box's power / thermal state the way the time does.  Only when stage timing was on (plume_set_stage_timing) for THAT call; PLUME_ERR_ARG otherwise (an earlier call's sample is never handed out). */
int plume_last_msm_clock(plume_ctx* ctx, double* ghz);
/* VALU issue-rate microbenchmark (32 waves per CU, 8 independent chains per lane, `iters` x 8 instructions per lane):
 * returns operations per second chip-wide (<= 0 on error).  kind: 0 v_mad_u64_u32, 1 v_addc_co_u32, 2 v_mul_lo_u32,
 * 3 v_mad_u32_u24, 4 v_add_u32, 5 one Fp multiplication, 6 one Fp squaring, 7 v_fma_f64, 8 v_lshl_add_u64.
 * kind 9 is the calibration probe of the HBM counters: 2048 x CUs lanes each gather, `iters` times, the five 16-byte quads of a table addition from a
 * pseudo-random 128-byte row of the context's window-table buffer (needs a verify of >= 2^19 items on the context first); returns gathers per second.  Run under
 * `rocprofv3 --pmc FETCH_SIZE` it tells what the counter reports per gather of this access pattern (profiles/README.md). */
double plume_microbench(plume_ctx* ctx, int kind, int iters);
/* s_memtime ticks that workgroup 0 spent inside the last microbenchmark kernel (and its event duration in ms). */
double plume_microbench_last_ticks(float* ms);

#ifdef __cplusplus
}
#endif
#endif /* PLUME_HIP_H */
