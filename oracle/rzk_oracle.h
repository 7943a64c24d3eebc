/*
 * rzk_oracle.h — CPU restatement ("oracle") of the ring-zk hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline — never as a fallback for the HIP path.
 *
 * What it restates (citations are file:line under /root/reference, the AlvinHon/ring-zk crate):
 *   - Polynomial<ZqI64<Q>,N> x, +, - : third-party crate poly-ring-xnp1 ^0.3 (Cargo.toml:18), whose
 *     source is NOT present in the reference tree.  The arithmetic is restated from its published
 *     definition: R_q = Z_q[X]/(X^N+1), c_t = sum_{i+j=t} a_i b_j - sum_{i+j=t+N} a_i b_j (mod q),
 *     coefficients kept as centred representatives in [-(q-1)/2, (q-1)/2] (params.rs:122-126).
 *   - Mat::{dot,add,sub,componentwise_mul}            src/mat.rs:95-178
 *   - norm_2                                           src/polynomial.rs:60-73
 *   - Params::{standard_deviation,check_*_constraint}  src/params.rs:94-118
 *   - CommitmentKey::{new,commit}, Commitment::{verify,c1_c2}   src/commit.rs:33-60,88-128,173-218
 *   - Open / Linear / Sum provers + verifiers          src/prove/{open,linear,sum}.rs
 *
 * PARITY STATUS: "parity unpinned" for the numeric value of a product mod q — no test, fixture or
 * golden vector in the reference pins one, and the reference cannot be built here
 * (no cargo/rustc, dependencies not vendored).  Pinned against the reference's own KATs:
 * norm_2([1,-2,3,-4]) = 5 (polynomial.rs:111-115), sigma(1024) = 21780 (params.rs:145-150), and the
 * Mat index-structure tests (mat.rs:243-268, 389-406) — see tests/test_oracle.py.
 *
 * All polynomials are dense arrays of N int64 coefficients (centred residues); matrices are
 * row-major arrays of polynomials ([row][col][N]).
 */
#ifndef RZK_ORACLE_H
#define RZK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar helpers -------------------------------------------------------------------- */
/* canonical centred representative of v mod q, q odd: result in [-(q-1)/2, (q-1)/2] */
int64_t rzko_center(int64_t v, int64_t q);
/* floor(sqrt(x)) on 64-bit (num::integer::Roots::sqrt on usize, params.rs:97,104,114) */
uint64_t rzko_isqrt_u64(uint64_t x);

/* ---- ring element ops (row a1..a3 of SURVEY §8a) ----------------------------------------- */
void rzko_poly_mul(int64_t q, uint32_t N, const int64_t* a, const int64_t* b, int64_t* out);
void rzko_poly_add(int64_t q, uint32_t N, const int64_t* a, const int64_t* b, int64_t* out);
void rzko_poly_sub(int64_t q, uint32_t N, const int64_t* a, const int64_t* b, int64_t* out);
int rzko_poly_eq(uint32_t N, const int64_t* a, const int64_t* b);

/* ---- Mat ops (src/mat.rs) ----------------------------------------------------------------- */
/* (m x n) . (n x p) -> (m x p), literal triple loop of mat.rs:106-113 */
void rzko_mat_dot(int64_t q, uint32_t N, uint32_t m, uint32_t n, uint32_t p, const int64_t* A,
                  const int64_t* B, int64_t* out);
void rzko_mat_add(int64_t q, uint32_t N, uint32_t m, uint32_t n, const int64_t* A, const int64_t* B,
                  int64_t* out);
void rzko_mat_sub(int64_t q, uint32_t N, uint32_t m, uint32_t n, const int64_t* A, const int64_t* B,
                  int64_t* out);
/* every entry times one polynomial, mat.rs:168-178 */
void rzko_mat_cmul(int64_t q, uint32_t N, uint32_t m, uint32_t n, const int64_t* A,
                   const int64_t* elem, int64_t* out);

/* ---- norms and constraints (src/polynomial.rs:60-73, src/params.rs:94-118) ------------------ */
/* floor(sqrt(sum c_i^2)), saturating at UINT64_MAX (cannot happen for |c| < 2^62, N <= 2^16) */
uint64_t rzko_norm2(uint32_t N, const int64_t* p);
uint64_t rzko_norm1(uint32_t N, const int64_t* p);
uint64_t rzko_norm_inf(uint32_t N, const int64_t* p);
/* sigma = b * (11*kappa) * floor(sqrt(k*N))   (params.rs:94-98) */
uint64_t rzko_sigma(uint64_t b, uint64_t kappa, uint64_t k, uint64_t N);
/* 4*sigma*floor(sqrt(N)) and 2*sigma*floor(sqrt(N))  (params.rs:104,114) */
uint64_t rzko_commit_bound(uint64_t b, uint64_t kappa, uint64_t k, uint64_t N);
uint64_t rzko_verify_bound(uint64_t b, uint64_t kappa, uint64_t k, uint64_t N);
/* all `count` polynomials have norm_2 <= bound (params.rs:105-107 / 115-117) */
int rzko_check_norm(uint32_t N, uint32_t count, const int64_t* polys, uint64_t bound);

/* ---- protocol parameters -------------------------------------------------------------------- */
typedef struct {
  int64_t q;      /* ring modulus Q (odd prime), NOT Params.q which is the sampling bound Q/2 */
  uint32_t N;     /* ring degree, power of two */
  uint32_t n, k, l;
  uint32_t kappa;
  uint64_t b;
} rzko_params;

/* CommitmentKey::new (commit.rs:33-60): A = [a1 ; a2] as built at commit.rs:109-114,
 * a1 = [I_n | a1'] (n x k), a2 = [0_{l x n} | I_l | a2'] (l x k).
 * a1p: n*(k-n) polys, a2p: l*(k-n-l) polys (row-major).  out: (n+l)*k polys. */
void rzko_key_build(const rzko_params* P, const int64_t* a1p, const int64_t* a2p, int64_t* A);

/* CommitmentKey::commit (commit.rs:88-128) with r supplied: c = A.r + [0_n ; x].
 * x: l polys, r: k polys, c: (n+l) polys.  Returns check_commit_constraint(r). */
int rzko_commit(const rzko_params* P, const int64_t* A, const int64_t* x, const int64_t* r,
                int64_t* c);
/* Commitment::verify with f = None (commit.rs:173-210) */
int rzko_commitment_verify(const rzko_params* P, const int64_t* A, const int64_t* c,
                           const int64_t* x, const int64_t* r);
/* commit.rs:173-210 with the optional scalar f (NULL = None) */
int rzko_commitment_verify_f(const rzko_params* P, const int64_t* A, const int64_t* c,
                             const int64_t* x, const int64_t* r, const int64_t* f);

/* ---- OpenProof (src/prove/open.rs) ------------------------------------------------------------ */
/* commit (open.rs:80-103): c = commit(x; r), t = a1.y.  t: n polys.  Returns constraint(r). */
int rzko_open_commit(const rzko_params* P, const int64_t* A, const int64_t* x, const int64_t* r,
                     const int64_t* y, int64_t* c, int64_t* t);
/* create_response (open.rs:107-117): z = y + r (.) d.  z: k polys */
void rzko_open_response(const rzko_params* P, const int64_t* y, const int64_t* r, const int64_t* d,
                        int64_t* z);
/* verify (open.rs:162-174) on the full commitment c (c1 taken per Commitment::c1_c2). */
int rzko_open_verify(const rzko_params* P, const int64_t* A, const int64_t* z, const int64_t* t,
                     const int64_t* c, const int64_t* d);

/* ---- LinearProof (src/prove/linear.rs) ---------------------------------------------------------- */
/* commit (linear.rs:82-140).  r for c (commit of x), rp for cp (commit of g*x).
 * Outputs: c, cp ((n+l) polys each), t, tp (n polys each), u (l polys).
 * Returns bit0 = constraint(r), bit1 = constraint(rp). */
int rzko_linear_commit(const rzko_params* P, const int64_t* A, const int64_t* g, const int64_t* x,
                       const int64_t* r, const int64_t* rp, const int64_t* y, const int64_t* yp,
                       int64_t* c, int64_t* cp, int64_t* t, int64_t* tp, int64_t* u);
/* create_response (linear.rs:144-158) */
void rzko_linear_response(const rzko_params* P, const int64_t* y, const int64_t* yp,
                          const int64_t* r, const int64_t* rp, const int64_t* d, int64_t* z,
                          int64_t* zp);
/* verify (linear.rs:213-250) */
int rzko_linear_verify(const rzko_params* P, const int64_t* A, const int64_t* z, const int64_t* zp,
                       const int64_t* c, const int64_t* cp, const int64_t* g, const int64_t* t,
                       const int64_t* tp, const int64_t* u, const int64_t* d);

/* ---- SumProof (src/prove/sum.rs) ------------------------------------------------------------------ */
/* commit (sum.rs:99-178), V summands.  gs: V polys; xs: V*l polys; rs: V*k polys (r of each x_i);
 * rp: k polys (r of x' = sum g_i x_i); ys: V*k polys; yp: k polys.
 * Outputs: cs: V*(n+l) polys, cp: (n+l), ts: V*n, tp: n, u: l.
 * Returns 1 iff every commit constraint holds. */
int rzko_sum_commit(const rzko_params* P, uint32_t V, const int64_t* A, const int64_t* gs,
                    const int64_t* xs, const int64_t* rs, const int64_t* rp, const int64_t* ys,
                    const int64_t* yp, int64_t* cs, int64_t* cp, int64_t* ts, int64_t* tp,
                    int64_t* u);
/* create_response (sum.rs:182-200) */
void rzko_sum_response(const rzko_params* P, uint32_t V, const int64_t* ys, const int64_t* yp,
                       const int64_t* rs, const int64_t* rp, const int64_t* d, int64_t* zs,
                       int64_t* zp);
/* verify (sum.rs:257-320) */
int rzko_sum_verify(const rzko_params* P, uint32_t V, const int64_t* A, const int64_t* zs,
                    const int64_t* zp, const int64_t* cs, const int64_t* cp, const int64_t* gs,
                    const int64_t* ts, const int64_t* tp, const int64_t* u, const int64_t* d);

/* ---- auxiliary-prime NTT (checker for the device NTT kernels; not in the reference) --------- */
/* In-place negacyclic forward NTT mod p: natural order in, bit-reversed order out (Cooley-Tukey,
 * merged psi twist).  psi = primitive 2N-th root of unity mod p.  Values in [0,p). */
void rzko_ntt_forward(uint32_t p, uint32_t psi, uint32_t N, uint32_t* a);
void rzko_ntt_forward_batch(uint32_t p, uint32_t psi, uint32_t N, uint64_t count, uint32_t* data, int threads);
/* inverse of the above: bit-reversed in, natural out, scaled by N^-1 */
void rzko_ntt_inverse(uint32_t p, uint32_t psi, uint32_t N, uint32_t* a);
uint32_t rzko_powmod(uint32_t base, uint64_t e, uint32_t p);

/* ---- batch drivers used only by bench.py's cpu_baseline leg ------------------------------------ */
/* One full OpenProof cycle (commit -> response -> verify) for `B` independent proofs, schoolbook
 * multiply, literal Mat::dot loop order, OpenMP over proofs with `threads` threads (0 = all).
 * Inputs laid out [B][..].  Returns the number of accepted proofs. */
int64_t rzko_open_cycle_batch(const rzko_params* P, uint32_t B, const int64_t* A, const int64_t* x,
                              const int64_t* r, const int64_t* y, const int64_t* d, int threads);
int rzko_hw_threads(void);

#ifdef __cplusplus
}
#endif
#endif
