/*
 * rzk.h — C ABI of the MI355X (gfx950) polynomial-ring backend for ring-zk.
 *
 * The reference crate (AlvinHon/ring-zk, /root/reference) has no plugin / FFI seam: all ring
 * arithmetic of its protocols flows through four crate-private methods of `Mat`
 * (src/mat.rs:95-178), one direct `Polynomial::mul` (src/prove/linear.rs:94) and the norm
 * predicates of `Params` (src/params.rs:102-118).  This header is the boundary a thin Rust shim
 * binds at exactly those places (INTEGRATION.md shows the `extern "C"` block and the replaced
 * function bodies).  Each entry point below cites the reference code it replaces.
 *
 * Conventions
 *   - A polynomial is N int64 coefficients, each the centred representative in
 *     [-(q-1)/2, (q-1)/2] (what ZqI64<Q> stores; src/params.rs:122-126); outputs always are.  The reference's
 *     type guarantees this for its own values (ZqI64::from reduces, src/params.rs:126); data from elsewhere
 *     (the wire, another library) may not be, so EVERY coefficient a kernel loads is tested while it is loaded
 *     (all 64 bits: k*2^32 + s is never read as s).  A violation
 *       - in a verifier-side entry point (rzk_*_verify_batch, rzk_commitment_verify_batch) clears the verdict of
 *         the offending proof (accept / ok = 0) and the call succeeds: a malformed proof is a rejected proof;
 *       - anywhere else (Mat primitives, commit / response phases) makes the call fail with RZK_E_ARG (the
 *         host-pointer variants at return, the *_dev variants at the next rzk_ctx_synchronize /
 *         rzk_ctx_check_inputs); per-proof ok flags of the offending proofs are cleared as well, outputs of
 *         those proofs are unspecified.  rzk_canonicalize_batch reduces foreign data first (= ZqI64::from);
 *         rzk_wire_mat_decode rejects out-of-range coefficients when given q.
 *   - Slabs are dense row-major: [batch][row][N].  A "batch" is B independent proofs that share
 *     only the commitment key.
 *   - Every function returns 0 (RZK_OK) or a negative status; nothing aborts.  The Rust shim turns
 *     a non-zero status into panic!, preserving the reference's assert!/assert_eq! behaviour on
 *     shape mismatch (src/mat.rs:103,129-130,154-155; src/commit.rs:95; src/prove/sum.rs:105).
 *   - `*_dev` variants take device pointers and are asynchronous on the context's HIP stream;
 *     the variants without the suffix take host pointers, copy in/out and are synchronous.
 *   - One context per (GPU, host thread); calls on a context are serialised on its stream.
 *   - The library never falls back to a CPU path: without a usable HIP device rzk_ctx_create fails.
 */
#ifndef RZK_H
#define RZK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RZK_OK 0
#define RZK_E_ARG (-1)         /* bad argument / shape mismatch (reference: assert panic)        */
#define RZK_E_HIP (-2)         /* HIP runtime error, see rzk_last_error                           */
#define RZK_E_STATE (-3)       /* e.g. key not loaded                                             */
#define RZK_E_UNSUPPORTED (-4) /* ring degree / modulus / shape outside what the kernels support */

typedef struct rzk_ctx rzk_ctx;

/* Version of this C ABI: bumped whenever an existing signature changes (rzk_wire_mat_decode gained `q` in 2,
 * version 3 added the entry points marked "v3").  A binding checks rzk_abi_version() == RZK_ABI_VERSION after loading
 * the library, so that a stale or variant .so fails at load time instead of reading shifted arguments. */
#define RZK_ABI_VERSION 3u
uint32_t rzk_abi_version(void);

/* Which block of the commitment key a matrix-vector product uses (src/commit.rs:19-25). */
#define RZK_KEY_A1 0 /* a1: n x k        (rows 0..n-1 of [a1;a2])  */
#define RZK_KEY_A2 1 /* a2: l x k        (rows n..n+l-1)           */
#define RZK_KEY_A 2  /* [a1;a2]: (n+l) x k, as built at src/commit.rs:109-114 */

/* ---- context ---------------------------------------------------------------------------------- */
/* Mirrors Params<ZqI64<Q>> (src/params.rs:18-36) + const generics N and Q.
 * q: ring modulus Q, NOT Params.q: odd, 1073692673 < q <= 4294606851 (above the first auxiliary NTT prime,
 *    (q-1)/2 below twice the third; the default 3515337053, src/params.rs:121, lies inside), else RZK_E_UNSUPPORTED.
 * N: power of two in [4, 2048].  512 / 1024 / 2048 run the NTT kernels (the BASELINE sizes); 4 .. 256
 * (the sizes of the reference's own tests: src/mat.rs:241 N=4, tests/test.rs:8 N=16) run a schoolbook
 * kernel, same results, for drop-in completeness.  Requires k > n >= 1, l >= 1, n + l <= k
 * (src/params.rs:26-31).
 * device: HIP device ordinal. */
int rzk_ctx_create(rzk_ctx** out, int64_t q, uint32_t N, uint32_t n, uint32_t k, uint32_t l,
                   uint32_t kappa, uint64_t b, int device);
void rzk_ctx_destroy(rzk_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. PyTorch's current stream); NULL selects HIP's default
 * (null) stream.  rzk_ctx_use_own_stream goes back to the private stream created with the context. */
int rzk_ctx_set_stream(rzk_ctx* ctx, void* hip_stream);
int rzk_ctx_use_own_stream(rzk_ctx* ctx);
/* v3.  Trusted-producer mode, default OFF.  With on != 0 the caller vouches that every coefficient later calls on this
 * context load is canonical — written by this library on this device (commitments, t, responses, sampler outputs), or
 * validated before (rzk_canonicalize_batch, rzk_wire_mat_decode with q) — and the kernels skip the per-coefficient
 * range test (3 vector instructions per loaded coefficient).  Norm predicates, the norm measurements that fix the
 * prime count, and therefore the results on canonical data are unchanged; on non-canonical data the results are
 * unspecified.  A verifier that receives z, t, c from elsewhere leaves this off (reference: ZqI64::from,
 * src/params.rs:126, makes every value canonical by construction).  Synchronises the stream. */
int rzk_ctx_trust_device_outputs(rzk_ctx* ctx, int on);
/* Waits for the context's stream.  Also the point where the asynchronous *_dev calls report non-canonical input
 * coefficients (see "Conventions"): RZK_E_ARG once, then the condition is cleared.  rzk_ctx_check_inputs is the
 * same call under the name a caller uses when it only wants that verdict. */
int rzk_ctx_synchronize(rzk_ctx* ctx);
int rzk_ctx_check_inputs(rzk_ctx* ctx);
/* Ordering rule for mixed use: a host-pointer call reports only its OWN input faults — it clears the sticky condition
 * when it starts.  A caller of *_dev entry points that wants their verdict must therefore ask for it
 * (rzk_ctx_check_inputs / rzk_ctx_synchronize) before its next host-pointer call on the same context. */
/* Message of the last failure on this context; with ctx == NULL, of the last failed rzk_ctx_create. */
const char* rzk_last_error(const rzk_ctx* ctx);
/* sigma, 4*sigma*floor(sqrt N), 2*sigma*floor(sqrt N)   (src/params.rs:94-98,104,114) */
uint64_t rzk_sigma(const rzk_ctx* ctx);
uint64_t rzk_commit_bound(const rzk_ctx* ctx);
uint64_t rzk_verify_bound(const rzk_ctx* ctx);

/* ---- key ----------------------------------------------------------------------------------------- */
/* CommitmentKey (src/commit.rs:19-25): a = [a1;a2], (n+l)*k polynomials row-major, exactly the
 * matrix commit() assembles at src/commit.rs:109-114 (identity / zero blocks included).  The key is
 * transformed once into the NTT domain of the three auxiliary primes and kept resident in HBM;
 * entries equal to 0 or 1 are recognised and skipped / turned into plain additions. */
int rzk_key_load(rzk_ctx* ctx, const int64_t* a_host);
int rzk_key_load_dev(rzk_ctx* ctx, const int64_t* a_dev);
/* CommitmentKey::new (src/commit.rs:33-60) with the device-side sampler: a1 = [I_n | U], a2 = [0 | I_l | U],
 * U uniform over [-(q-1)/2, (q-1)/2] (src/params.rs:126), drawn from (seed); the key is loaded and, when
 * a_host_out != NULL, also returned ([n+l][k][N], the public key material).  Statistical parity (see samplers). */
int rzk_key_generate(rzk_ctx* ctx, uint64_t seed, int64_t* a_host_out);

/* ---- ring / Mat primitives (the Mat seam) ------------------------------------------------------------ */
/* Polynomial::mul (src/prove/linear.rs:94; src/mat.rs:110,176): out[i] = a[i] * b[i], count polys */
int rzk_polymul_batch(rzk_ctx* ctx, const int64_t* a, const int64_t* b, int64_t* out, size_t count);
int rzk_polymul_batch_dev(rzk_ctx* ctx, const int64_t* a, const int64_t* b, int64_t* out, size_t count);
/* Mat::dot with the key on the left (src/mat.rs:95-115; call sites src/commit.rs:125,
 * src/prove/open.rs:97,171 ...): out[b] = KEY(which) . v[b] (+ addend[b] when addend != NULL, the
 * `.add(&z)` of src/commit.rs:125).  v: [B][k][N]; out, addend: [B][rows(which)][N]. */
int rzk_matvec_batch(rzk_ctx* ctx, int which, const int64_t* v, const int64_t* addend, int64_t* out,
                     size_t B);
int rzk_matvec_batch_dev(rzk_ctx* ctx, int which, const int64_t* v, const int64_t* addend,
                         int64_t* out, size_t B);
/* Mat::componentwise_mul (src/mat.rs:168-178): out[b][i] = m[b][i] * p[b].  m,out: [B][rows][N]; p: [B][N] */
int rzk_cmul_batch(rzk_ctx* ctx, const int64_t* m, uint32_t rows, const int64_t* p, int64_t* out, size_t B);
int rzk_cmul_batch_dev(rzk_ctx* ctx, const int64_t* m, uint32_t rows, const int64_t* p, int64_t* out,
                       size_t B);
/* Any int64 coefficient -> the centred representative in [-(q-1)/2, (q-1)/2] (what ZqI64::from does,
 * src/params.rs:126).  Every other entry point rejects non-canonical inputs (see Conventions); data that did not
 * pass through a ZqI64 goes through this first.  in, out: [count][N], may alias. */
int rzk_canonicalize_batch(rzk_ctx* ctx, const int64_t* in, int64_t* out, size_t count);
int rzk_canonicalize_batch_dev(rzk_ctx* ctx, const int64_t* in, int64_t* out, size_t count);
/* Mat::add / Mat::sub (src/mat.rs:122-140, 147-165) on `count` polynomials */
int rzk_add_batch(rzk_ctx* ctx, const int64_t* a, const int64_t* b, int64_t* out, size_t count);
int rzk_add_batch_dev(rzk_ctx* ctx, const int64_t* a, const int64_t* b, int64_t* out, size_t count);
int rzk_sub_batch(rzk_ctx* ctx, const int64_t* a, const int64_t* b, int64_t* out, size_t count);
int rzk_sub_batch_dev(rzk_ctx* ctx, const int64_t* a, const int64_t* b, int64_t* out, size_t count);
/* Params::check_{commit,verify}_constraint (src/params.rs:102-118 -> src/polynomial.rs:60-73):
 * ok[b] = 1 iff every one of the `rows` polynomials of proof b has floor(sqrt(sum c^2)) <= bound */
int rzk_norm2_le_batch(rzk_ctx* ctx, const int64_t* v, uint32_t rows, uint64_t bound, uint8_t* ok, size_t B);
int rzk_norm2_le_batch_dev(rzk_ctx* ctx, const int64_t* v, uint32_t rows, uint64_t bound, uint8_t* ok,
                           size_t B);
/* derived Mat PartialEq (src/mat.rs:11; `lhs == rhs` at src/prove/open.rs:173): eq[b] = all rows equal */
int rzk_eq_batch(rzk_ctx* ctx, const int64_t* a, const int64_t* b, uint32_t rows, uint8_t* eq, size_t B);
int rzk_eq_batch_dev(rzk_ctx* ctx, const int64_t* a, const int64_t* b, uint32_t rows, uint8_t* eq, size_t B);

/* ---- batched transforms over one auxiliary prime (no reference counterpart) ---------- */
/* in/out: `count` residue polynomials of N uint32 in [0,p).  Forward output / inverse input use the
 * library's NTT-domain layout (rzk_ntt_layout_index).  prime: 0..2. */
int rzk_ntt_forward_batch(rzk_ctx* ctx, int prime, const uint32_t* in, uint32_t* out, size_t count);
int rzk_ntt_forward_batch_dev(rzk_ctx* ctx, int prime, const uint32_t* in, uint32_t* out, size_t count);
int rzk_ntt_inverse_batch(rzk_ctx* ctx, int prime, const uint32_t* in, uint32_t* out, size_t count);
int rzk_ntt_inverse_batch_dev(rzk_ctx* ctx, int prime, const uint32_t* in, uint32_t* out, size_t count);
uint32_t rzk_ntt_prime(int prime);
/* primitive 2N-th root of unity used for ring degree N */
uint32_t rzk_ntt_psi(int prime, uint32_t N);
/* position, in the NTT-domain layout, of element j of the standard bit-reversed-order transform */
uint32_t rzk_ntt_layout_index(uint32_t N, uint32_t j);

/* ---- commitment scheme (src/commit.rs) ------------------------------------------------------------------ */
/* CommitmentKey::commit (commit.rs:88-128) with the randomness supplied by the caller:
 * c = [a1;a2].r + [0_n ; x] (commit.rs:109-125); ok[b] = check_commit_constraint(r_b) — the reference
 * resamples r until it holds (commit.rs:98-107), here the caller resamples the proofs with ok == 0.
 * x:[B][l][N] r:[B][k][N] -> c:[B][n+l][N] ok:[B] (ok may be NULL) */
int rzk_commit_batch(rzk_ctx* ctx, const int64_t* x, const int64_t* r, int64_t* c, uint8_t* ok, size_t B);
int rzk_commit_batch_dev(rzk_ctx* ctx, const int64_t* x, const int64_t* r, int64_t* c, uint8_t* ok, size_t B);
/* Commitment::verify (commit.rs:173-210) of the opening (x, r, f):
 * ok[b] = check_commit_constraint(r) && ( f == NULL ?  [a1;a2].r + [0_n;x] == c
 *                                                   :  c (.) f == [a1;a2].r + [0_n;x] (.) f )
 * c:[B][n+l][N] x:[B][l][N] r:[B][k][N] f:[B][N] or NULL (Opening::f = None) -> ok:[B] */
int rzk_commitment_verify_batch(rzk_ctx* ctx, const int64_t* c, const int64_t* x, const int64_t* r,
                                const int64_t* f, uint8_t* ok, size_t B);
int rzk_commitment_verify_batch_dev(rzk_ctx* ctx, const int64_t* c, const int64_t* x, const int64_t* r,
                                    const int64_t* f, uint8_t* ok, size_t B);

/* ---- OpenProof phases (src/prove/open.rs) --------------------------------------------------------- */
/* OpenProofProver::commit (open.rs:80-103) with the randomness supplied by the caller:
 * c = [a1;a2].r + [0;x] (commit.rs:125), t = a1.y (open.rs:97); ok[b] = check_commit_constraint(r)
 * (commit.rs:98-107: the reference resamples r until it holds; here the caller does).
 * x:[B][l][N] r,y:[B][k][N] -> c:[B][n+l][N] t:[B][n][N] ok:[B] */
int rzk_open_commit_batch(rzk_ctx* ctx, const int64_t* x, const int64_t* r, const int64_t* y,
                          int64_t* c, int64_t* t, uint8_t* ok, size_t B);
int rzk_open_commit_batch_dev(rzk_ctx* ctx, const int64_t* x, const int64_t* r, const int64_t* y,
                              int64_t* c, int64_t* t, uint8_t* ok, size_t B);
/* OpenProofProver::create_response (open.rs:107-117): z = y + r (.) d.   d:[B][N] z:[B][k][N] */
int rzk_open_response_batch(rzk_ctx* ctx, const int64_t* y, const int64_t* r, const int64_t* d,
                            int64_t* z, size_t B);
int rzk_open_response_batch_dev(rzk_ctx* ctx, const int64_t* y, const int64_t* r, const int64_t* d,
                                int64_t* z, size_t B);
/* OpenProofVerifier::verify (open.rs:162-174) on the full commitment c (c1 split as
 * Commitment::c1_c2 does, commit.rs:213-218; requires n == l):
 * accept[b] = check_verify_constraint(z) && a1.z == t + c1 (.) d */
int rzk_open_verify_batch(rzk_ctx* ctx, const int64_t* z, const int64_t* t, const int64_t* c,
                          const int64_t* d, uint8_t* accept, size_t B);
int rzk_open_verify_batch_dev(rzk_ctx* ctx, const int64_t* z, const int64_t* t, const int64_t* c,
                              const int64_t* d, uint8_t* accept, size_t B);

/* ---- LinearProof phases (src/prove/linear.rs) -------------------------------------------------------- */
/* commit (linear.rs:82-140): gx = g*x; cp = commit(gx; rp); c = commit(x; r); t = a1.y; tp = a1.yp;
 * u = (a2.y) (.) g - a2.yp.   ok[b]: bit0 = constraint(r), bit1 = constraint(rp).
 * g:[B][N] x:[B][l][N] r,rp,y,yp:[B][k][N] -> c,cp:[B][n+l][N] t,tp:[B][n][N] u:[B][l][N] */
int rzk_linear_commit_batch(rzk_ctx* ctx, const int64_t* g, const int64_t* x, const int64_t* r,
                            const int64_t* rp, const int64_t* y, const int64_t* yp, int64_t* c,
                            int64_t* cp, int64_t* t, int64_t* tp, int64_t* u, uint8_t* ok, size_t B);
int rzk_linear_commit_batch_dev(rzk_ctx* ctx, const int64_t* g, const int64_t* x, const int64_t* r,
                                const int64_t* rp, const int64_t* y, const int64_t* yp, int64_t* c,
                                int64_t* cp, int64_t* t, int64_t* tp, int64_t* u, uint8_t* ok, size_t B);
/* create_response (linear.rs:144-158): z = y + r(.)d, zp = yp + rp(.)d */
int rzk_linear_response_batch(rzk_ctx* ctx, const int64_t* y, const int64_t* yp, const int64_t* r,
                              const int64_t* rp, const int64_t* d, int64_t* z, int64_t* zp, size_t B);
int rzk_linear_response_batch_dev(rzk_ctx* ctx, const int64_t* y, const int64_t* yp, const int64_t* r,
                                  const int64_t* rp, const int64_t* d, int64_t* z, int64_t* zp, size_t B);
/* verify (linear.rs:213-250) */
int rzk_linear_verify_batch(rzk_ctx* ctx, const int64_t* z, const int64_t* zp, const int64_t* c,
                            const int64_t* cp, const int64_t* g, const int64_t* t, const int64_t* tp,
                            const int64_t* u, const int64_t* d, uint8_t* accept, size_t B);
int rzk_linear_verify_batch_dev(rzk_ctx* ctx, const int64_t* z, const int64_t* zp, const int64_t* c,
                                const int64_t* cp, const int64_t* g, const int64_t* t, const int64_t* tp,
                                const int64_t* u, const int64_t* d, uint8_t* accept, size_t B);

/* ---- SumProof phases (src/prove/sum.rs), V summands per proof ----------------------------------------- */
/* commit (sum.rs:99-178).  gs:[B][V][N] xs:[B][V][l][N] rs,ys:[B][V][k][N] rp,yp:[B][k][N]
 * -> cs:[B][V][n+l][N] cp:[B][n+l][N] ts:[B][V][n][N] tp:[B][n][N] u:[B][l][N]; ok[b] = all constraints */
int rzk_sum_commit_batch(rzk_ctx* ctx, uint32_t V, const int64_t* gs, const int64_t* xs,
                         const int64_t* rs, const int64_t* rp, const int64_t* ys, const int64_t* yp,
                         int64_t* cs, int64_t* cp, int64_t* ts, int64_t* tp, int64_t* u, uint8_t* ok,
                         size_t B);
int rzk_sum_commit_batch_dev(rzk_ctx* ctx, uint32_t V, const int64_t* gs, const int64_t* xs,
                             const int64_t* rs, const int64_t* rp, const int64_t* ys, const int64_t* yp,
                             int64_t* cs, int64_t* cp, int64_t* ts, int64_t* tp, int64_t* u,
                             uint8_t* ok, size_t B);
/* create_response (sum.rs:182-200) */
int rzk_sum_response_batch(rzk_ctx* ctx, uint32_t V, const int64_t* ys, const int64_t* yp,
                           const int64_t* rs, const int64_t* rp, const int64_t* d, int64_t* zs,
                           int64_t* zp, size_t B);
int rzk_sum_response_batch_dev(rzk_ctx* ctx, uint32_t V, const int64_t* ys, const int64_t* yp,
                               const int64_t* rs, const int64_t* rp, const int64_t* d, int64_t* zs,
                               int64_t* zp, size_t B);
/* verify (sum.rs:257-320) */
int rzk_sum_verify_batch(rzk_ctx* ctx, uint32_t V, const int64_t* zs, const int64_t* zp,
                         const int64_t* cs, const int64_t* cp, const int64_t* gs, const int64_t* ts,
                         const int64_t* tp, const int64_t* u, const int64_t* d, uint8_t* accept,
                         size_t B);
int rzk_sum_verify_batch_dev(rzk_ctx* ctx, uint32_t V, const int64_t* zs, const int64_t* zp,
                             const int64_t* cs, const int64_t* cp, const int64_t* gs, const int64_t* ts,
                             const int64_t* tp, const int64_t* u, const int64_t* d, uint8_t* accept,
                             size_t B);

/* ---- instrumentation (bench.py) ------------------------------------------------------------------------- */
/* Times `iters` launches of the batched forward NTT kernel with HIP events on the context stream and
 * returns the average kernel duration in microseconds (negative status on error). */
double rzk_bench_ntt_forward_dev(rzk_ctx* ctx, int prime, const uint32_t* in, uint32_t* out,
                                 size_t count, int iters);
/* ---- device-side samplers ----------------------------------------------------------------------------------- */
/* The distributions of the reference's host RNG helpers, drawn in HBM by a counter-based generator
 * (Philox4x32-10; output = f(seed, stream, polynomial index, position), reproducible and order-independent).
 * Parity with the reference is statistical, not bit-for-bit (the reference draws from rand's thread RNG).
 * count = number of polynomials written to out:[count][N].
 *   uniform   random_polynomial_within (src/polynomial.rs:14-25): coefficients uniform in [-bound, bound],
 *             1 <= bound <= (q-1)/2  (commit randomness r: bound = b, commit.rs:101; key / message: (q-1)/2)
 *   gauss     random_polynomial_in_normal_distribution (src/polynomial.rs:28-44): (i64) N(0, sigma), truncated
 *             toward zero like I::from_f64; y of the provers: sigma = rzk_sigma(ctx) (open.rs:88-94).  Box-Muller on a
 *             64-bit uniform (tail to 9.4 sigma); evaluated in single precision for sigma < 2^19 (sample error < 0.1
 *             before the truncation: every parameter set of the reference), in double precision up to 2^26
 *   challenge random_polynomial_from_challenge_set (src/challenge_space.rs:12-33): exactly kappa coefficients
 *             +-1 (kappa of the context) at a uniformly random subset of positions, zeros elsewhere */
int rzk_sample_uniform_dev(rzk_ctx* ctx, uint64_t seed, uint32_t stream, uint64_t bound, int64_t* out, size_t count);
int rzk_sample_gauss_dev(rzk_ctx* ctx, uint64_t seed, uint32_t stream, double sigma, int64_t* out, size_t count);
int rzk_sample_challenge_dev(rzk_ctx* ctx, uint64_t seed, uint32_t stream, int64_t* out, size_t count);

/* ---- wire format (host only) ---------------------------------------------------------------------------- */
/* bincode layout of the reference's Mat<I,N> (serde derive at src/mat.rs:11-14; bincode default options as in
 * the reference's own test src/mat.rs:424-438: little-endian, u64 length prefixes):
 *     u64 rows ; rows x { u64 cols ; cols x { u64 len ; len x coefficient } }
 * with every polynomial in the crate's trimmed form (no trailing zero coefficients) <-> dense slabs
 * [rows][cols][N] of int64.  The protocol messages (src/commit.rs:134, src/prove/open.rs:180-228, ...) are
 * plain concatenations of their Mat fields in declaration order.  coef_bytes: 8 (i64) or 4 (i32, the width
 * of the reference's test vector); the width ZqI64 uses on the wire is set by the third-party ring crate and
 * is not pinned by any file of the reference.
 * encode: writes rzk_wire_mat_size() bytes (returned through *written; RZK_E_ARG if cap is too small or a
 *         coefficient does not fit coef_bytes).
 * decode: slab may be NULL to measure / validate only; fails on ragged rows, polynomials longer than N, truncated
 *         input and, with q > 0, on any coefficient outside the centred range [-(q-1)/2, (q-1)/2] of a ZqI64
 *         (src/params.rs:122-127) — q = 0 decodes plain integers, as the reference's i32 test vector needs;
 *         *consumed = bytes read, so consecutive fields of a message can be decoded in turn. */
size_t rzk_wire_mat_size(const int64_t* slab, uint32_t rows, uint32_t cols, uint32_t N, uint32_t coef_bytes);
int rzk_wire_mat_encode(const int64_t* slab, uint32_t rows, uint32_t cols, uint32_t N, uint32_t coef_bytes,
                        uint8_t* out, size_t cap, size_t* written);
int rzk_wire_mat_decode(const uint8_t* in, size_t len, uint32_t N, uint32_t coef_bytes, int64_t q, uint32_t* rows,
                        uint32_t* cols, int64_t* slab, size_t slab_polys, size_t* consumed);

/* HIP-event timing of the last phase call's dominant kernel is exposed through these counters:
 * accumulated microseconds and launch count of the row kernel since the last reset. */
/* diagnostic: copies the row kernels' per-wave scratch lines to the host (tools/wave_timeline.py; *total = its size) */
int rzk_debug_read_scratch(rzk_ctx* ctx, void* dst, size_t bytes, size_t* total);
int rzk_prof_reset(rzk_ctx* ctx);
int rzk_prof_enable(rzk_ctx* ctx, int on);
int rzk_prof_read(rzk_ctx* ctx, double* row_kernel_us, uint64_t* row_kernel_launches);
/* launches recorded since the last reset / read (host counter, no synchronisation), and their individual
 * durations in launch order (synchronises; does not clear) */
uint64_t rzk_prof_count(const rzk_ctx* ctx);
int rzk_prof_read_all(rzk_ctx* ctx, double* us, size_t cap, size_t* count);
/* v3.  Which kernel every recorded launch ran and its algorithmic bytes (8 N x (distinct polynomials the row program
 * reads + rows it stores) x batch entries), one "<kernel>\t<bytes>\n" line per launch in launch order; *needed = length
 * of the full text.  bench.py names the dominant kernel of every configuration with this. */
int rzk_prof_read_kernels(rzk_ctx* ctx, char* buf, size_t cap, size_t* needed);

/* ---- environment variables (read once by rzk_ctx_create) ----------------------------------------------------------
 * Testing / tuning switches, NOT part of the API contract: every setting computes the same results; they select which
 * kernel evaluates a row program so that the test-suite can reach every kernel at small shapes and so that A/B
 * measurements need no rebuild.  Defaults are what the measurements in DESIGN.md §6 chose.
 *   RZK_SHIFT=0            challenge products through transforms instead of signed rotations (default 1; N <= 1024)
 *   RZK_PAIRS=0            unit_kernel: no pairing of rows that share their last operand (default 1)
 *   RZK_UPT=<u>            unit_kernel: units of a proof per wavefront task (default: all once batch >= 16 x CUs, else 1)
 *   RZK_VEC_ROWS=0         programs with vector x vector products through unit_kernel instead of row_kernel (default 1)
 *   RZK_ROW_GROUPS=0       no row groups (row_group_kernel) for key blocks with n > 1 (default 1)
 *   RZK_GROUP_MAX=<g>      rows per group, 1 .. 4 (default 4 at N <= 1024, 1 at N = 2048)
 *   RZK_BLOCK_MIN_LOGN=<L> row blocks (row_block_kernel) from ring degree 2^L on (default 11; 12 = never, 10 = also N = 1024)
 *   RZK_SLOT_SHARE_MIN=<x> shared-operand path when (operand transforms) / (distinct operands) >= x (default 2.0; 0 = never)
 *   RZK_PAIR_POLY=0        N = 2048: one wavefront per polynomial instead of two (default 1)
 *   RZK_OIMG=0             Sum proof: the rows of D = sum_i g_i v_i transform v_{i,c} themselves (default 1: they read the transforms the
 *                          a1.v_i key products of the same call left behind, where those ran on the group / block kernels)
 *   RZK_LIN_E=0            Linear verifier: the reference's grouping (a2.z)(.)g - a2.z' == (c2(.)g - c2')(.)d + u with its two products by g
 *                          (default 1: g(.)(a2.z - c2(.)d) - (a2.z' - c2'(.)d) - u == 0, one product)
 *   RZK_DKEY=0|1|2         the scalar multipliers g / g_i of the Linear / Sum proofs transformed once per proof and call into the
 *                          resident key's form: never / when at least four rows of the call multiply by each (default) / always
 *   RZK_SUM_D=0|1          Sum proof: sum_i g_i (a2.v_i) - a2.v' row by row (0) or as a2.(sum_i g_i v_i - v') (1); default: whichever
 *                          needs fewer transforms for the loaded key and V
 *   RZK_PRESET_IN_KERNEL=0 verdict flags always initialised by a fill launch (default 1: by the unit kernels themselves where one
 *                          team evaluates a whole batch entry)
 *   RZK_UNIT_IO=0|1        key-product programs without vector operands: unit_io_kernel (operands read once, per-prime sums
 *                          parked on chip) instead of unit_kernel (default 1 at N = 512, 0 otherwise)
 * RZK_LIB (ring_zk_amd/_lib.py, Python only) loads another build of the library for A/B runs. */

#ifdef __cplusplus
}
#endif
#endif /* RZK_H */
