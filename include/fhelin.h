/*
 * fhelin.h — C ABI of the MI355X-native RNS-CKKS evaluation engine (libfhelin_amd.so).
 *
 * This is the drop-in boundary underneath the reference's `class FHEController`
 * (reference src/FHEController.h:22-161).  The reference binds OpenFHE's C++ objects directly
 * (`CryptoContext<DCRTPoly> context`, `Ciphertext<DCRTPoly>`, `Plaintext`; src/FHEController.h:19-23);
 * a maintainer replaces those with the opaque handles below (see INTEGRATION.md and the
 * source-compatible shim include/FHEController.h).  Every entry point cites the reference call site
 * it stands in for.  All functions return 0 on success or an FHELIN_ERR_* code; the message of the
 * last failure on the calling thread is available from fhelin_last_error().
 *
 * Conventions
 *  - plain C types only: pointers, sizes, integers, doubles.  No torch / HIP types in signatures
 *    (a hipStream_t travels as void*).
 *  - residue data is uint64_t, limb-major: poly[limb][N]; ciphertext = poly 0 then poly 1 (then 2).
 *  - limb ids: Q limbs 0..L, special (P) limbs L+1..L+k.
 *  - device work is asynchronous on the context's stream; fhelin_sync() waits for it (and first issues the deferred
 *    bootstraps / polynomial evaluations whose results nobody has read yet, see fhelin_bootstrap_batch).
 *  - there is NO CPU fallback: on a host-only context (device < 0) every evaluation entry point
 *    fails with FHELIN_ERR_NO_DEVICE.
 */
#ifndef FHELIN_H
#define FHELIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHELIN_OK 0
#define FHELIN_ERR_ARG 1
#define FHELIN_ERR_NO_DEVICE 2
#define FHELIN_ERR_HIP 3
#define FHELIN_ERR_STATE 4
#define FHELIN_ERR_KEY 5
#define FHELIN_ERR_INTERNAL 6

typedef struct fhelin_ctx fhelin_ctx;   /* CryptoContext<DCRTPoly>            (FHEController.h:23)  */
typedef struct fhelin_ct fhelin_ct;     /* Ctxt = Ciphertext<DCRTPoly>        (FHEController.h:20)  */
typedef struct fhelin_pt fhelin_pt;     /* Ptxt = Plaintext                   (FHEController.h:19)  */

/* CCParams<CryptoContextCKKSRNS> as set in generate_context (FHEController.cpp:4-35). */
typedef struct fhelin_params {
    int32_t log_n;         /* SetRingDim(1 << log_n)                 :12-13 */
    int32_t n_q;           /* multiplicative depth + 1               :31-35 */
    int32_t first_bits;    /* SetFirstModSize(55)                    :25    */
    int32_t scale_bits;    /* SetScalingModSize(52), FLEXIBLEAUTO    :18-24 */
    int32_t n_p;           /* special primes of HYBRID key switching (OpenFHE-internal); < 0: OpenFHE's rule
                              ceil(bits of the widest digit / special_bits), read it back with fhelin_ctx_info */
    int32_t special_bits;  /* 60                                                       */
    int32_t dnum;          /* SetNumLargeDigits(4)                   :11    */
    int32_t log_slots;     /* SetBatchSize(1 << 14)                  :6,14  */
    int32_t hamming;       /* SPARSE_TERNARY secret weight           :8     */
    int32_t device;        /* HIP device ordinal; < 0 = host-only parameter context */
    uint64_t seed;         /* 0: the client-side generator (ChaCha20) is keyed with 256 bits of OS entropy — the default and the
                              only secure choice; != 0: deterministic 64-bit TEST seed (reproducible tests / benchmarks) */
} fhelin_params;

const char* fhelin_last_error(void);
const char* fhelin_version(void);

/* ---- context (GenCryptoContext + Enable(...), FHEController.cpp:37-45) ------------------------ */
int fhelin_ctx_create(const fhelin_params* p, fhelin_ctx** out);
/* the same with an explicit 256-bit secret seed (p->seed ignored): a client re-creating its own keys from its secret-key
 * file, as FHEController::load_context does from ../keys/secret-key.txt (FHEController.cpp:208-214) */
int fhelin_ctx_create_seeded(const fhelin_params* p, const uint8_t* seed32, fhelin_ctx** out);
/* the 256-bit secret seed all key material of this context derives from — SECRET: belongs in the client's secret-key
 * file (FHEController.cpp:80-86 writes secret-key.txt), never next to the public context */
int fhelin_ctx_secret_seed(const fhelin_ctx* c, uint8_t* out32);
/* one 64-byte ChaCha20 block of the client-side generator (known-answer test hook, RFC 8439 section 2.3.2) */
int fhelin_prng_block(const uint8_t* seed32, uint64_t counter, uint64_t stream, uint8_t* out64);
void fhelin_ctx_destroy(fhelin_ctx* c);
int fhelin_ctx_info(const fhelin_ctx* c, fhelin_params* out, int32_t* alpha, int32_t* has_device);
int fhelin_ctx_moduli(const fhelin_ctx* c, uint64_t* out, int32_t cap);          /* Q then P */
int fhelin_ctx_roots(const fhelin_ctx* c, uint64_t* out, int32_t cap);           /* psi per limb */
int fhelin_ctx_scaling_factors(const fhelin_ctx* c, double* out, int32_t cap);   /* real Delta per level */
int fhelin_ctx_set_stream(fhelin_ctx* c, void* hip_stream);                      /* adopt a caller stream */
/* Deferred rows (default on; FHELIN_LAZY_ROWS=0 or on = 0 here: eager).  fhelin_fc_matmul_pt and fhelin_fc_unwrapExpanded
 * return handles whose rows are evaluated when first read (together with the other rows of the same call the reading
 * operation takes) and never if nobody reads them: the reference's drivers compute whole row sets and then use one row
 * (src/main.cpp:183,:196; :416-424).  A forced row holds exactly the residues eager evaluation gives. */
int fhelin_ctx_set_lazy_rows(fhelin_ctx* c, int32_t on);
/* evaluate EXACTLY the deferred rows among v[0..n) now, in one batched call per producing call (a rank of a row-sharded run
 * evaluates the rows it owns and nothing else); handles that are not deferred are left alone */
int fhelin_ct_force(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n);
/* Level plan (profile-guided; off unless asked for).  The reference's drivers are straight-line programs (src/main.cpp:145-475):
 * which ciphertext meets which, and how many limbs every call consumes, does not depend on the data.  A pass run with
 * fhelin_level_plan_begin(ctx, 1) RECORDS, for every handle this boundary gives out, the handles the producing call read
 * and the result's limb count; fhelin_level_plan_end then derives, back to front from the terminals (the input of a
 * bootstrap needs two limbs, as does a decryption / export), the fewest limbs every value may have, and from that the limb
 * count each SOURCE of the pass - the k-th fhelin_encrypt / fhelin_encrypt_batch vector / fhelin_bootstrap call, in call
 * order - should start with.  A later pass of the same program run with fhelin_level_plan_begin(ctx, 2) APPLIES the plan:
 * encryptions are made at the planned level and bootstraps raise to fewer limbs, so that limbs nothing downstream reads
 * (the reference carries e.g. 20 through the V projection, src/main.cpp:212-215, and 4 from the GELU bootstraps to the
 * pooler's, :354-420) are not dragged through the key switches in between.  Values are unchanged up to the noise; a pass
 * that does not follow the recorded program fails loudly at the first bootstrap / decryption that is short of limbs.
 * get/set move the plan (one int32 per source: limbs to start with, -1 = as asked) so that a driver process can keep it
 * next to its keys.  fhelin_ct_import / _import_device values are never lowered; exporting is a terminal. */
int fhelin_level_plan_begin(fhelin_ctx* c, int32_t mode);            /* 0 off, 1 record, 2 apply; resets the source counter */
int fhelin_level_plan_seek(fhelin_ctx* c, int32_t source);           /* apply: the next source is the source-th of the program
                                                                        (a server picking up after the client's encryptions) */
int fhelin_level_plan_tell(fhelin_ctx* c, int32_t* mode, int32_t* source);   /* the mode and the index the NEXT source call will take: a
                                                                        driver that leaves out sources other processes run (rows and
                                                                        bootstraps of one sample sharded over GPUs) reads the position,
                                                                        seeks past the call it skips and back for the ones it runs */
int fhelin_level_plan_end(fhelin_ctx* c, int32_t* n_sources);        /* record: derive the plan; back to mode 0 */
int fhelin_level_plan_get(fhelin_ctx* c, int32_t* target, int32_t cap, int32_t* n);   /* *n = length; fills min(cap, *n) */
int fhelin_level_plan_set(fhelin_ctx* c, const int32_t* target, int32_t n);
int fhelin_sync(fhelin_ctx* c);
/* Lanes: a context owns a few extra HIP streams ("lanes" 1..n, FHELIN_LANES, default 2) besides its main stream (lane 0), each with
 * its own arena of device memory.  After fhelin_ctx_set_lane(c, k) every call launches on lane k and allocates there, until the next
 * set_lane; deferred operations (fhelin_bootstrap, ...) issued under a lane are evaluated on that lane.  Independent work - the two
 * halves of a batch of samples, src/main.cpp:145-475 once per sample - issued alternately under two lanes runs CONCURRENTLY on the
 * GPU from ONE host thread and ONE context (one key set, one plaintext cache): the tail of one lane's launch is filled by the
 * other's.  lanes_fork: lanes 1..n wait for everything the main stream has been given so far (the inputs); lanes_join: the main
 * stream waits for every lane (before decrypting / exporting under lane 0).  A value must be consumed under the lane that produced it,
 * after a join, or after fhelin_ctx_lane_wait(producer's lane) under the consuming lane (a value of one branch of a circuit meeting
 * another branch's); results are bit-identical to lane 0 (scheduling only). */
int fhelin_ctx_set_lane(fhelin_ctx* c, int32_t lane);
int fhelin_ctx_lane_wait(fhelin_ctx* c, int32_t from_lane);   /* the current lane's stream waits for everything issued under from_lane so far */
/* mark: remember the present point of the current lane's stream; wait_mark: the current lane waits for from_lane's last mark (not for what
 * was issued there after it).  A scheduler that starts a new branch of a circuit on a free lane holds it back until the other lane's
 * last BULK call (a row loop that fills the GPU by itself) has drained: the new branch then runs beside the small launches that follow
 * it there, instead of beside the bulk call, where it would gain nothing. */
int fhelin_ctx_lane_mark(fhelin_ctx* c);
int fhelin_ctx_lane_wait_mark(fhelin_ctx* c, int32_t from_lane);
int fhelin_ctx_lanes_fork(fhelin_ctx* c);
int fhelin_ctx_lanes_join(fhelin_ctx* c);
/* give the device memory the context's caching pool holds but does not use back to the driver (another context / process
 * on the same GPU can then have it); synchronises first */
int fhelin_ctx_trim(fhelin_ctx* c);
/* HIP-event timer on the context's stream (bench.py measures kernel time with these) */
int fhelin_timer_start(fhelin_ctx* c);
int fhelin_timer_stop(fhelin_ctx* c, float* ms);

/* ---- raw device buffers + residue-polynomial kernels (parity tests and bench call these) ------- */
int fhelin_dev_alloc(fhelin_ctx* c, size_t bytes, void** out);
int fhelin_dev_free(fhelin_ctx* c, void* p);
int fhelin_dev_upload(fhelin_ctx* c, void* dst, const void* src, size_t bytes);
int fhelin_dev_download(fhelin_ctx* c, void* dst, const void* src, size_t bytes);

/* K1: in-place negacyclic NTT/INTT of nvec limb vectors d_data[v][N]; vector v uses limb
 * limb_first + (v % limb_count).  Stands in for DCRTPoly::SetFormat inside OpenFHE. */
int fhelin_ntt(fhelin_ctx* c, uint64_t* d_data, int32_t nvec, int32_t limb_first, int32_t limb_count, int32_t inverse);

/* host-side operation counters since the last reset: [0] limb-NTTs, [1] key switches, [2] sum of live limbs over
 * key switches, [3] rescales, [4] ct x pt products, [5] bootstraps, [6] plaintext encodes, and with cap >= 9:
 * [7] sum of live limbs over rescales, [8] sum of live limbs over ct x pt products, and with cap >= 12 the growth of the
 * device pool: [9] blocks obtained from hipMalloc, [10] their bytes, [11] host nanoseconds spent inside hipMalloc, and with
 * cap >= 16: [12] bytes the pool holds from the driver now, [13] / [14] high-water marks of bytes in use / held since the last
 * reset, [15] out-of-memory events (everything idle handed back: a device-wide synchronisation each) */
int fhelin_stats(fhelin_ctx* c, uint64_t* out, int32_t cap, int32_t reset);
/* host-side self-test of the device memory arena (slabs, best fit, coalescing) against a pretend device of device_bytes: n_ops random
 * allocations / frees in the engine's size mix; checks that blocks never overlap, lie inside a slab, that freeing everything leaves one
 * free range per slab and that a trim returns every slab.  out (cap >= 6): [0] peak bytes in use, [1] peak bytes held, [2] slabs
 * obtained, [3] out-of-memory events, [4] coalesced-to-one-range-per-slab (1/0), [5] trim-returned-everything (1/0).  No GPU needed. */
int fhelin_debug_pool_selftest(uint64_t seed, int32_t n_ops, uint64_t device_bytes, uint64_t* out, int32_t cap);


/* ---- keys (client side; sampling on the host, polynomial arithmetic on the GPU) ---------------- */
int fhelin_keygen(fhelin_ctx* c);                       /* context->KeyGen()                  FHEController.cpp:47  */
int fhelin_gen_relin_key(fhelin_ctx* c);                /* context->EvalMultKeyGen(sk)        :49                   */
int fhelin_gen_rotation_keys(fhelin_ctx* c, const int32_t* indices, int32_t n);  /* EvalRotateKeyGen  :248       */
int fhelin_gen_conj_key(fhelin_ctx* c);
/* raw key material, [L+1+k][N] (secret, NTT form) and [dnum][2][L+1+k][N] (switching keys): parity tests
 * hand the same arrays to the oracle.  kind: 0 = relinearisation key, 1 = rotation key for `index`, 2 = conjugation key. */
int fhelin_secret_export(fhelin_ctx* c, uint64_t* out, size_t cap_words);
int fhelin_secret_import(fhelin_ctx* c, const uint64_t* in, size_t words);
int fhelin_key_export(fhelin_ctx* c, int32_t kind, int32_t index, uint64_t* out, size_t cap_words);
int fhelin_key_import(fhelin_ctx* c, int32_t kind, int32_t index, const uint64_t* in, size_t words);

/* ---- plaintexts: context->MakeCKKSPackedPlaintext(vec, 1, level, nullptr, slots)  :353,:368 ---- */
int fhelin_encode(fhelin_ctx* c, const double* vals, int32_t n, int32_t level, int32_t slots, fhelin_pt** out);
void fhelin_pt_free(fhelin_pt* p);
/* the residues [ell][N] (NTT form) this plaintext multiplies / adds with at `ell` live limbs and real scaling factor
 * scale_hi + scale_lo (the library keeps scales as 80-bit long double: two doubles carry one exactly; scale_hi <= 0:
 * the context's Delta of that level) — exactly the encoding EvalMult(ct,pt) / EvalAdd(ct,pt) use for a ciphertext of that
 * shape (parity tests hand them to the oracle's dyadic functions).  ell = L + 1 + k (all limbs of the chain and the special
 * limbs; explicit scale required): the encoding over the full key basis, [L + 1 + k][N], as fhelin_hoisted_dot folds it into keys */
int fhelin_pt_export(fhelin_ctx* c, const fhelin_pt* p, int32_t ell, double scale_hi, double scale_lo, uint64_t* out,
                     size_t cap_words);

/* ---- ciphertexts ----------------------------------------------------------------------------- */
int fhelin_encrypt(fhelin_ctx* c, const fhelin_pt* p, fhelin_ct** out);                 /* context->Encrypt   :380,:384 */
/* n_vec inputs at once (the driver's 194 read_expanded_input calls per sample, src/main.cpp:159-173; FHEController.cpp:623-650):
 * vals [n_vec][n_per] real slot values -> n_vec fresh ciphertexts at `level`.  Encoding (special FFT in fp64, scaling,
 * rounding), the sampling of the encryption randomness (ChaCha20 on the GPU, keyed from the client's generator) and the
 * dyadic combination run as batched kernels. */
int fhelin_encrypt_batch(fhelin_ctx* c, const double* vals, int32_t n_vec, int32_t n_per, int32_t level, int32_t slots, fhelin_ct** outs);
/* The client side of ONE sample on the device: what the reference does in NumPy before the server sees anything
 * (src/python/dimReduce.py:141-160: x_in = [cls; emb + pos/3], X_E = E[:, :S+1] x_in + b_E, X_F likewise) followed by the 64 + S + 1
 * read_expanded_input calls of the driver (src/main.cpp:159-173, src/FHEController.cpp:623-650) - gather / projection / packing /
 * encoding / sampling / encryption as GPU kernels, fp64 with the operation order of the NumPy statement (sequential sums, no FMA).
 * Give the S token embeddings as emb [S][128], or token ids tokens [S] into table [vocab][128].  pos [>= S][128], cls [128],
 * E_w / F_w [32][w_cols] row-major (w_cols >= S + 1), E_b / F_b [32].  outs: 64 + S + 1 handles in the order of the driver's reads:
 * the 32 E-projected rows, the 32 F-projected rows, then the S + 1 tokens (CLS first).  Every output is a source of the level plan.
 * proj_out (optional, (S + 1 + 64) * 128 doubles): x_in rows then the projected rows as the device computed them (parity tests). */
int fhelin_client_ingest(fhelin_ctx* c, const double* emb, const int32_t* tokens, const double* table, int32_t vocab, int32_t S,
                         const double* cls, const double* pos, const double* E_w, const double* E_b, const double* F_w, const double* F_b,
                         int32_t w_cols, int32_t level, fhelin_ct** outs, double* proj_out);
/* 1: the special FFT of CKKS encoding runs on the host (the original encoder, the reference the device encoder is compared
 * with bit for bit); 0 (default): on the GPU.  Both produce identical residues. */
int fhelin_ctx_set_host_encode(fhelin_ctx* c, int32_t on);
/* test hook: n_poly polynomials of N centred coefficients straight from the device sampler (kind 0: rounded Gaussian
 * sigma 3.19, 1: uniform ternary) */
int fhelin_debug_sample(fhelin_ctx* c, int32_t kind, int32_t n_poly, int64_t* out, size_t cap);
int fhelin_decrypt(fhelin_ctx* c, const fhelin_ct* ct, double* out, int32_t slots);     /* context->Decrypt + GetRealPackedValue :387-404 */
int fhelin_ct_import(fhelin_ctx* c, const uint64_t* limbs, int32_t npoly, int32_t ell, int32_t deg, double scale,
                     int32_t slots, fhelin_ct** out);
int fhelin_ct_export(fhelin_ctx* c, const fhelin_ct* ct, uint64_t* out, size_t cap_words);
/* the same with DEVICE buffers (a torch / RCCL tensor): ciphertext rows exchanged between the GPUs of a node travel
 * without a host round trip.  The scale is the library's 80-bit value as hi + lo doubles. */
int fhelin_ct_export_device(fhelin_ctx* c, const fhelin_ct* ct, uint64_t* d_out, size_t cap_words);
int fhelin_ct_import_device(fhelin_ctx* c, const uint64_t* d_limbs, int32_t npoly, int32_t ell, int32_t deg, double scale_hi,
                            double scale_lo, int32_t slots, fhelin_ct** out);
int fhelin_ct_scale(const fhelin_ct* ct, double* scale_hi, double* scale_lo);
int fhelin_ct_info(const fhelin_ct* ct, int32_t* npoly, int32_t* ell, int32_t* level, int32_t* deg, double* scale,
                   int32_t* slots);                                                     /* ct->GetLevel() / GetSlots()  */
int fhelin_ct_clone(fhelin_ctx* c, const fhelin_ct* ct, fhelin_ct** out);                /* ct->Clone()  (main.cpp:223) */
void fhelin_ct_free(fhelin_ct* ct);

/* ---- leveled evaluation (FLEXIBLEAUTO bookkeeping included) ------------------------------------ */
int fhelin_add(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b, fhelin_ct** out);          /* EvalAdd(ct,ct)   :410 */
int fhelin_sub(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b, fhelin_ct** out);
int fhelin_negate(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out);
int fhelin_add_plain(fhelin_ctx* c, const fhelin_ct* a, const fhelin_pt* p, fhelin_ct** out);    /* EvalAdd(ct,pt)   :414 */
int fhelin_mult_plain(fhelin_ctx* c, const fhelin_ct* a, const fhelin_pt* p, fhelin_ct** out);   /* EvalMult(ct,pt)  :427 */
int fhelin_mult(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b, fhelin_ct** out);         /* EvalMult(ct,ct)  :431 */
int fhelin_rotate(fhelin_ctx* c, const fhelin_ct* a, int32_t index, fhelin_ct** out);            /* EvalRotate       :435,:833,:843 */
/* hoisted rotations: outs[i] = EvalRotate(a, indices[i]) for all i with ONE decomposition of a (OpenFHE's
 * EvalFastRotationPrecompute + EvalFastRotation pair); the results are bit-identical to n calls of fhelin_rotate */
int fhelin_rotate_many(fhelin_ctx* c, const fhelin_ct* a, const int32_t* indices, int32_t n, fhelin_ct** outs);
/* outs[i] = EvalRotate(v[i], indices[i]): rows of identical (level, degree, scale) share one batched key switch */
int fhelin_rotate_each(fhelin_ctx* c, const fhelin_ct* const* v, const int32_t* indices, int32_t n, fhelin_ct** outs);
/* outs[i] = v[i] + sum_r EvalRotate(v[i], indices[r]) (1 <= n_rot <= 7) with one decomposition and ONE ModDown per
 * row: the rotated terms are accumulated in the extended basis QP.  Two steps of the reference's rotsum loop (:829-837),
 * x += rot(x, s); x += rot(x, 2s), are the call {s, 2s, 3s}.  Needs the rotation keys of all indices. */
int fhelin_rotate_sum(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const int32_t* indices, int32_t n_rot, fhelin_ct** outs);
/* outs[i] = v[i] * pts[0] + sum_r EvalRotate(v[i], indices[r]) * pts[r + 1] (1 <= n_rot <= 7; pts: n_rot + 1 plaintexts) with one
 * decomposition and ONE ModDown per row ("double hoisting"): the plaintext products are taken in the extended basis QP, through
 * rotation keys with the plaintext folded in (built once per (plaintext, index, scale) and cached: one full key copy each - meant
 * for few plaintexts shared by many rows).  The first step of matmulRElarge (src/FHEController.cpp:915-944 re-associated, DESIGN.md
 * §7) is the call {128, 256, 384} over all rows.  Result: noise degree + 1, scale x the level's plaintext scale.  The plaintext
 * encodings over the full key basis are fhelin_pt_export(p, L + 1 + k, scale).
 * rescale != 0: the result rescaled, with ModDown and rescale as ONE basis conversion (P and the top limb dropped together: one rounding
 * instead of two, 2 ell fewer transforms per row): one limb fewer, the input's noise degree, scale / q_top. */
int fhelin_hoisted_dot(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const fhelin_pt* const* pts, const int32_t* indices,
                       int32_t n_rot, int32_t rescale, fhelin_ct** outs);
/* out = sum_i EvalRotate(v[i], indices[i]) (index 0 = plain addend): the giant steps of EvalBootstrap's linear
 * transforms (:445) — one ModUp per term, inner products accumulated in QP, ONE ModDown per group of <= 7 terms */
int fhelin_rotate_each_sum(fhelin_ctx* c, const fhelin_ct* const* v, const int32_t* indices, int32_t n, fhelin_ct** out);
int fhelin_rescale(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out);                          /* ModReduce (implicit in :427/:431) */
/* the leaf operations over n independent ciphertexts (the rows of the reference's matmul loops, :872,:888,:904,:918,
 * :949,:963,:985,:1001) in one call: same results as n single calls, one launch set per chunk of rows of equal shape */
int fhelin_rotate_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t index, fhelin_ct** outs);
int fhelin_rescale_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** outs);
int fhelin_mult_plain_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const fhelin_pt* p, fhelin_ct** outs);
int fhelin_mult_batch(fhelin_ctx* c, const fhelin_ct* const* a, const fhelin_ct* const* b, int32_t n, fhelin_ct** outs);
int fhelin_add_batch(fhelin_ctx* c, const fhelin_ct* const* a, const fhelin_ct* const* b, int32_t n, fhelin_ct** outs);
int fhelin_level_reduce(fhelin_ctx* c, const fhelin_ct* a, int32_t new_ell, fhelin_ct** out);

/* ---- the same residue functions without scale/level bookkeeping (bit-exact parity vs oracle/) -- */
int fhelin_raw_rescale(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out);                      /* K5            */
int fhelin_raw_rotate(fhelin_ctx* c, const fhelin_ct* a, int32_t index, fhelin_ct** out);        /* K4 + K6-K8    */
int fhelin_raw_mult_relin(fhelin_ctx* c, const fhelin_ct* a, const fhelin_ct* b, fhelin_ct** out);/* K2 + K6-K8   */
/* K9 ModRaise, first step of EvalBootstrap (:445): a ciphertext with ONE limb -> new_ell limbs (centred lift) */
int fhelin_raw_modraise(fhelin_ctx* c, const fhelin_ct* a, int32_t new_ell, fhelin_ct** out);
/* the decryption phase c0 + c1 s (+ c2 s^2) on every live limb, as a 1-component handle (context->Decrypt :389 before
 * decoding; client side: needs the secret key) */
int fhelin_raw_phase(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out);

/* ---- FHEController composite circuit ops (callers of the hot path) --------
 * One entry point per reference method; `vector<Ctxt>` travels as (array of handles, count); outputs are
 * written to caller-provided handle arrays.  A NULL bias means `bias == nullptr` in the reference. */
int fhelin_fc_mult_const(fhelin_ctx* c, const fhelin_ct* a, double d, fhelin_ct** out);                 /* mult(ct,double) :421 */
/* kind 0 mask_block(from=a,to=b,v) :1207 | 1 mask_heads(v) :1221 | 2 mask_heads_128(v) :1235 |
 *      3 mask_mod_n(n=a,padding=b) :1249,:1262 | 4 mask_first_n(n=a,v) :1275 */
int fhelin_fc_mask(fhelin_ctx* c, const fhelin_ct* a, int32_t kind, int32_t x, int32_t y, double v, fhelin_ct** out);
int fhelin_fc_rotsum(fhelin_ctx* c, const fhelin_ct* a, int32_t slots, int32_t padding, fhelin_ct** out);  /* rotsum :829, rotsum_padded :839 */
int fhelin_fc_repeat(fhelin_ctx* c, const fhelin_ct* a, int32_t slots, int32_t padding, fhelin_ct** out);  /* repeat :849,:859 */
int fhelin_fc_add_many(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out);            /* add(vector) :417 */
/* matmulRE :869,:885 (slots=row_size, padding) and matmulCR :982 (slots=128, padding=1), plaintext weight */
int fhelin_fc_matmul_pt(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_pt* w, const fhelin_pt* bias,
                        int32_t slots, int32_t padding, fhelin_ct** outs);
/* matmulRE(ct weight) :901, matmulCR(ct) :946 (64,1), matmulCR_128 :960,:974 (128,1) */
int fhelin_fc_matmul_ct(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_ct* w, int32_t slots,
                        int32_t padding, fhelin_ct** outs);
int fhelin_fc_matmulRElarge(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_pt* const* weights,
                            int32_t nw, const fhelin_pt* bias, double mask_val, fhelin_ct** outs);      /* :915 */
/* rows: n x 4 handles, row-major */
int fhelin_fc_matmulCRlarge(fhelin_ctx* c, const fhelin_ct* const* rows, int32_t n, const fhelin_pt* const* weights,
                            const fhelin_pt* bias, fhelin_ct** outs);                                   /* :998 */
int fhelin_fc_matmulScores(fhelin_ctx* c, const fhelin_ct* const* queries, int32_t n, const fhelin_ct* key, fhelin_ct** out); /* :1028,:1050 */
int fhelin_fc_wrapUpRepeated(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out);       /* :1060 */
int fhelin_fc_wrapUpExpanded(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out);       /* :1070 */
int fhelin_fc_unwrapExpanded(fhelin_ctx* c, const fhelin_ct* a, int32_t inputs_num, fhelin_ct** outs);    /* :1086 */
int fhelin_fc_unwrapScoresExpanded(fhelin_ctx* c, const fhelin_ct* a, int32_t inputs_num, fhelin_ct** outs); /* :1125 */
int fhelin_fc_unwrap_512_in_4_128(fhelin_ctx* c, const fhelin_ct* a, int32_t index, fhelin_ct** outs4);   /* :1142 */
/* outs: input_number x 4 handles, row-major */
int fhelin_fc_unwrapRepeatedLarge(fhelin_ctx* c, const fhelin_ct* const* containers, int32_t nc, int32_t input_number,
                                  fhelin_ct** outs);                                                    /* :1102 */
/* the tokens [first, first + count) only (rows partitioned over the GPUs of a node): outs = count x 4 handles */
int fhelin_fc_unwrapRepeatedLarge_range(fhelin_ctx* c, const fhelin_ct* const* containers, int32_t nc, int32_t input_number,
                                        int32_t first, int32_t count, fhelin_ct** outs);
/* outs must hold ceil(n/32) handles; *n_out receives the count */
int fhelin_fc_generate_containers(fhelin_ctx* c, const fhelin_ct* const* inputs, int32_t n, const fhelin_pt* bias,
                                  fhelin_ct** outs, int32_t* n_out);                                    /* :1164 */
int fhelin_fc_wrap_containers(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t inputs_number, fhelin_ct** out); /* :1193 */

/* ---- the same FHEController calls on a BATCH OF B SAMPLES through one context (BASELINE config 4's per-GPU unit: "the batch of
 * independent input ciphertexts"; a sample is one run of the driver, src/main.cpp:145-475, and samples never meet).  Arrays are
 * sample-major: v[x * n + i] = row i of sample x; all samples of a call have one row count.  Sample x's outputs hold EXACTLY the
 * residues the single-sample entry point gives on sample x's inputs - rows of a batched key switch are independent - but the small
 * launches of one pass (a single query row, a container tail, one wrapped ciphertext: single-digit row counts that cannot fill 256
 * CUs) now carry the rows of every sample: one key set, one plaintext cache, one launch set.  The row loops proper
 * (fhelin_fc_matmul_pt, _matmulRElarge, _matmulCRlarge, fhelin_*_batch, fhelin_bootstrap_batch, fhelin_eval_chebyshev_batch) take the
 * samples' rows concatenated as they are; deferred rows of several samples that are read by ONE call are evaluated together. */
int fhelin_fcb_matmulScores(fhelin_ctx* c, const fhelin_ct* const* queries, int32_t n, const fhelin_ct* const* keys, int32_t B,
                            fhelin_ct** outs);                                                        /* :1028,:1050; outs[B] */
/* matmulRE / matmulCR with a ciphertext weight PER ROW (sample x's rows against sample x's wrapped keys / values, :901,:946,:960) */
int fhelin_fcb_matmul_ct(fhelin_ctx* c, const fhelin_ct* const* rows, const fhelin_ct* const* ws, int32_t n, int32_t slots, int32_t padding,
                         fhelin_ct** outs);
int fhelin_fcb_wrapUpRepeated(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t B, fhelin_ct** outs);       /* :1060; outs[B] */
int fhelin_fcb_wrapUpExpanded(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t B, fhelin_ct** outs);       /* :1070; outs[B] */
int fhelin_fcb_unwrapExpanded(fhelin_ctx* c, const fhelin_ct* const* cs, int32_t B, int32_t inputs_num, fhelin_ct** outs); /* :1086; outs[B * inputs_num] */
int fhelin_fcb_unwrapRepeatedLarge(fhelin_ctx* c, const fhelin_ct* const* containers, int32_t nc, int32_t B, int32_t input_number,
                                   fhelin_ct** outs);                                                 /* :1102; outs[B * input_number * 4] */
/* outs: B x ceil(n/32) handles; *n_out = containers per sample */
int fhelin_fcb_generate_containers(fhelin_ctx* c, const fhelin_ct* const* inputs, int32_t n, int32_t B, const fhelin_pt* bias,
                                   fhelin_ct** outs, int32_t* n_out);                                 /* :1164 */
/* rotsum (:829) / repeat (:849, repeat != 0) over n independent ciphertexts: one batched key switch per tree step */
int fhelin_fc_rotsum_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t slots, int32_t padding, int32_t repeat, fhelin_ct** outs);
int fhelin_add_plain_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, const fhelin_pt* p, fhelin_ct** outs);  /* EvalAdd(ct,pt) per row */
/* EvalPoly (:1291) on n ciphertexts, EvalMultMany (:1297) on B operand lists of n handles: one batched relinearisation per round */
int fhelin_eval_poly_batch(fhelin_ctx* c, const fhelin_ct* const* xs, int32_t n, const double* coeffs, int32_t n_coeffs, fhelin_ct** outs);
int fhelin_mult_many_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, int32_t B, fhelin_ct** outs);

/* ---- polynomial evaluation (ADVANCEDSHE) -------------------------------------------------------- */
int fhelin_mult_real(fhelin_ctx* c, const fhelin_ct* a, double k, fhelin_ct** out);              /* EvalMult(ct, double)            */
int fhelin_add_real(fhelin_ctx* c, const fhelin_ct* a, double k, fhelin_ct** out);               /* EvalAdd(ct, double)             */
int fhelin_mult_many(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** out);      /* EvalMultMany       :1297        */
/* sum_k coeffs[k] * v[k] + c0 (OpenFHE's EvalLinearWSum, the inner sums of EvalPoly / EvalChebyshevSeries :1291,:1319-1335):
 * ciphertexts of one (level, degree 1, scale) are combined in a single kernel pass; same residues as the chain of
 * EvalMult(ct,double) / EvalAdd / EvalAdd(ct,double) */
int fhelin_lincomb(fhelin_ctx* c, const fhelin_ct* const* v, const double* coeffs, int32_t n, double c0, fhelin_ct** out);
/* power-basis polynomial sum_i coeffs[i] x^i                                                      EvalPoly           :1291        */
int fhelin_eval_poly(fhelin_ctx* c, const fhelin_ct* x, const double* coeffs, int32_t n, fhelin_ct** out);
/* Chebyshev series coeffs[0]/2 + sum_{k>=1} coeffs[k] T_k(u), u = (2x-(a+b))/(b-a)     EvalChebyshevFunction :1319-1335
 * (the shim computes the coefficients from the C++ lambda exactly like EvalChebyshevCoefficients) */
int fhelin_eval_chebyshev(fhelin_ctx* c, const fhelin_ct* x, const double* coeffs, int32_t n, double a, double b, fhelin_ct** out);
/* the same series on n ciphertexts at once (EvalChebyshevFunction in a driver's loop over independent ciphertexts,
 * src/main.cpp:354-358): every multiplication round is ONE batched relinearisation over all of them; residues identical to n calls */
int fhelin_eval_chebyshev_batch(fhelin_ctx* c, const fhelin_ct* const* xs, int32_t n, const double* coeffs, int32_t n_coeffs, double a,
                                double b, fhelin_ct** outs);
/* CKKS bootstrapping: EvalBootstrapSetup/KeyGen :238-239 and EvalBootstrap :445 */
int fhelin_bootstrap_setup(fhelin_ctx* c, int32_t level_budget_enc, int32_t level_budget_dec, int32_t slots);
int fhelin_bootstrap(fhelin_ctx* c, const fhelin_ct* a, fhelin_ct** out);
/* EvalBootstrap on n independent ciphertexts (the reference's loop over the GELU containers, src/main.cpp:354-358; the two
 * halves of affine-1, :313-314): every launch of the pipeline carries all of them.  outs[i] holds exactly the residues of
 * fhelin_bootstrap(v[i]).  fhelin_bootstrap / fhelin_eval_chebyshev calls issued back to back on independent ciphertexts are
 * batched the same way by themselves: their results are evaluated when first read (FHELIN_LAZY_HEAVY=0: at the call). */
int fhelin_bootstrap_batch(fhelin_ctx* c, const fhelin_ct* const* v, int32_t n, fhelin_ct** outs);
/* approximation parameters (before setup): |I| bound K, double-angle count R, cosine-fit degree, message correction 2^-c */
int fhelin_bootstrap_config(fhelin_ctx* c, int32_t K, int32_t R, int32_t cheb_degree, int32_t correction);
/* test hook: stop after 1 = ModRaise(+SubSum), 2 = CoeffsToSlots (real part), 3 = approximate mod (real part) */
int fhelin_bootstrap_partial(fhelin_ctx* c, const fhelin_ct* a, int32_t stage, fhelin_ct** out);
/* fhelin_bootstrap raising to L+1-drop limbs only: what a level plan (fhelin_level_plan_*) asks of the k-th bootstrap of a
 * recorded program, here with the number given by the caller */
int fhelin_bootstrap_drop(fhelin_ctx* c, const fhelin_ct* a, int32_t drop, fhelin_ct** out);
/* ---- read-only views of the bootstrapping set-up: the residue-level oracle (oracle/residue_boot.py, tests only) composes
 * EvalBootstrap (:445) from the same linear stages, Chebyshev coefficients and keys and must reach the same residues.
 * describe: out = {packed, slots, K, R, cheb_degree, correction, depth, n_c2s, n_s2c, then per stage (CoeffsToSlots stages
 * first): stage_slots, n_terms, n_terms x (giant, baby)}; *n = words needed (fills min(cap, *n)).
 * diag: the plaintext diagonal of term `term` of stage `stage` (which: 0 CoeffsToSlots, 1 SlotsToCoeffs) as a handle for
 * fhelin_pt_export.  cheb: the cosine-fit coefficients EvalMod evaluates.  Key kind 2 of fhelin_key_export = conjugation key. */
int fhelin_bootstrap_describe(fhelin_ctx* c, int32_t* out, int32_t cap, int32_t* n);
int fhelin_bootstrap_diag(fhelin_ctx* c, int32_t which, int32_t stage, int32_t term, fhelin_pt** out);
int fhelin_bootstrap_cheb(fhelin_ctx* c, double* out, int32_t cap, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* FHELIN_H */
