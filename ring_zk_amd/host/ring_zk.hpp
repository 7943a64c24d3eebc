// ring_zk.hpp — C++ host mirror of the reference's public API over the C ABI (include/rzk.h).
//
// The reference is a Rust crate (no toolchain for it in this image), so this header plays the role of
// the Rust shim of INTEGRATION.md: the same types and method names as src/lib.rs:5-24 —
// Params, CommitmentKey, Commitment, Opening, {Open,Linear,Sum}Proof{Prover,Verifier} — with every ring
// operation delegated to librzk_hip.so.  Sampling (random_polynomial_within, the rounded Gaussian,
// the challenge set; src/polynomial.rs:14-44, src/challenge_space.rs:12-33) stays on the host, as in
// the reference.  Shape mismatches throw std::runtime_error where the reference panics.
//
// Single-proof calls (batch = 1) go through the host-pointer entry points; that is the drop-in
// behaviour, not the fast path — batches use the *_batch entry points directly.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/rzk.h"

namespace ring_zk {

using Poly = std::vector<int64_t>;     // N dense centred coefficients (Polynomial<ZqI64<Q>, N>)
using PolyVec = std::vector<Poly>;     // Vec<Polynomial> / an (m x 1) Mat
using Rng = std::mt19937_64;

constexpr int64_t kDefaultQ = 3515337053LL;   // ZqI64<3515337053>, src/params.rs:121

// src/params.rs:18-36 (+ the ring modulus, which the reference carries as the const generic of ZqI64)
struct Params {
  int64_t modulus = kDefaultQ;
  int64_t q = kDefaultQ / 2;   // sampling bound of key / message coefficients (params.rs:126)
  uint64_t b = 1;
  size_t n = 1, k = 3, l = 1, kappa = 36;

  static uint64_t isqrt(uint64_t x) {
    uint64_t r = (uint64_t)std::sqrt((long double)x);
    while (r * r > x) --r;
    while ((r + 1) * (r + 1) <= x) ++r;
    return r;
  }
  // params.rs:94-98
  uint64_t standard_deviation(size_t deg_n) const { return b * (11 * kappa) * isqrt(k * deg_n); }

  // ZqI64::from: the centred representative mod the ring modulus
  int64_t center(int64_t v) const {
    int64_t r = v % modulus;
    const int64_t half = (modulus - 1) / 2;
    if (r > half) r -= modulus;
    if (r < -half) r += modulus;
    return r;
  }
  // prepare_scalar (params.rs:89-91): integers -> one polynomial (coefficients reduced into Z_q)
  template <size_t N>
  Poly prepare_scalar(const std::vector<int64_t>& scalar) const {
    static_assert(N != 0 && (N & (N - 1)) == 0, "N must be a power of two (params.rs:86-87)");
    if (scalar.size() > N) throw std::runtime_error("prepare_scalar: more than N coefficients");
    Poly p;
    for (int64_t v : scalar) p.push_back(center(v));
    return p;
  }
  // prepare_value (params.rs:67-78): l integer vectors -> the message polynomials; panics unless value.len() == l
  template <size_t N>
  PolyVec prepare_value(const std::vector<std::vector<int64_t>>& value) const {
    if (value.size() != l) throw std::runtime_error("prepare_value: value.len() != l (params.rs:71)");
    PolyVec out;
    for (const auto& v : value) out.push_back(prepare_scalar<N>(v));
    return out;
  }
};

inline void flatten(const PolyVec& v, size_t N, std::vector<int64_t>& out) {
  for (const Poly& p : v) {
    if (p.size() > N) throw std::runtime_error("polynomial longer than the ring degree");
    const size_t start = out.size();
    out.insert(out.end(), p.begin(), p.end());
    out.resize(start + N, 0);   // trimmed representation (mat.rs:430-434) -> dense slab
  }
}
inline Poly dense(const Poly& p, size_t N) {   // zero-pad a trimmed polynomial to N coefficients
  if (p.size() > N) throw std::runtime_error("polynomial longer than the ring degree");
  Poly d(p);
  d.resize(N, 0);
  return d;
}
inline PolyVec unflatten(const std::vector<int64_t>& flat, size_t N) {
  PolyVec out;
  for (size_t i = 0; i + N <= flat.size(); i += N) out.emplace_back(flat.begin() + i, flat.begin() + i + N);
  return out;
}

// one rzk_ctx per (Params, N); shared by key, provers and verifiers
template <size_t N>
class Backend {
 public:
  explicit Backend(const Params& p, int device = 0) : params(p) {
    if (rzk_ctx_create(&ctx_, p.modulus, (uint32_t)N, (uint32_t)p.n, (uint32_t)p.k, (uint32_t)p.l,
                       (uint32_t)p.kappa, p.b, device) != RZK_OK)
      throw std::runtime_error(std::string("rzk_ctx_create: ") + rzk_last_error(nullptr));
  }
  ~Backend() { rzk_ctx_destroy(ctx_); }
  Backend(const Backend&) = delete;
  Backend& operator=(const Backend&) = delete;
  rzk_ctx* ctx() const { return ctx_; }
  void check(int st) const {
    if (st != RZK_OK) throw std::runtime_error(std::string("rzk: ") + rzk_last_error(ctx_));   // reference: panic!
  }
  Params params;

 private:
  rzk_ctx* ctx_ = nullptr;
};

// ---- Mat (src/mat.rs:11-238): m x n matrix of polynomials; arithmetic goes through the C ABI ------------------
template <size_t N>
class Mat {
 public:
  std::vector<PolyVec> polynomials;   // rows of polynomials (trimmed or dense), as in mat.rs:16

  Mat() = default;
  explicit Mat(std::vector<PolyVec> rows) : polynomials(std::move(rows)) {}
  static Mat from_element(size_t m, size_t n, const Poly& element) {            // mat.rs:24-31
    return Mat(std::vector<PolyVec>(m, PolyVec(n, element)));
  }
  static Mat diag(size_t m, size_t n, const Poly& element) {                    // mat.rs:33-44
    Mat out(std::vector<PolyVec>(m, PolyVec(n, Poly{})));
    for (size_t i = 0; i < m && i < n; ++i) out.polynomials[i][i] = element;
    return out;
  }
  static Mat from_vec(const PolyVec& v) {                                       // mat.rs:46-50
    Mat out;
    for (const Poly& p : v) out.polynomials.push_back(PolyVec{p});
    return out;
  }
  PolyVec one_d_mat_to_vec() const {                                            // mat.rs:56-64
    PolyVec v;
    for (const PolyVec& row : polynomials) {
      if (row.size() != 1) throw std::runtime_error("Matrix dimension is not (m x 1)");
      v.push_back(row[0]);
    }
    return v;
  }
  std::pair<size_t, size_t> dim() const {                                       // mat.rs:79-87
    return {polynomials.size(), polynomials.empty() ? 0 : polynomials[0].size()};
  }
  // dot (mat.rs:95-115): (m x n) . (n x p); every entry is sum_k a_ik * b_kj, evaluated as n batched products
  Mat dot(const Mat& other, const Backend<N>& be) const {
    const auto [m, n] = dim();
    const auto [n2, p] = other.dim();
    if (n != n2) throw std::runtime_error("Mat::dot: dimension mismatch (mat.rs:103)");
    Mat out(std::vector<PolyVec>(m, PolyVec(p, Poly(N, 0))));
    std::vector<int64_t> lhs, rhs, prod(m * p * N), acc(m * p * N, 0);
    for (size_t kk = 0; kk < n; ++kk) {
      lhs.clear();
      rhs.clear();
      for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; j < p; ++j) {
          flatten(PolyVec{polynomials[i][kk]}, N, lhs);
          flatten(PolyVec{other.polynomials[kk][j]}, N, rhs);
        }
      be.check(rzk_polymul_batch(be.ctx(), lhs.data(), rhs.data(), prod.data(), m * p));
      be.check(rzk_add_batch(be.ctx(), acc.data(), prod.data(), acc.data(), m * p));
    }
    const PolyVec flat = unflatten(acc, N);
    for (size_t i = 0; i < m; ++i)
      for (size_t j = 0; j < p; ++j) out.polynomials[i][j] = flat[i * p + j];
    return out;
  }
  Mat add(const Mat& other, const Backend<N>& be) const { return addsub(other, be, false); }   // mat.rs:122-140
  Mat sub(const Mat& other, const Backend<N>& be) const { return addsub(other, be, true); }    // mat.rs:147-165
  Mat componentwise_mul(const Poly& element, const Backend<N>& be) const {      // mat.rs:168-178
    const auto [m, n] = dim();
    std::vector<int64_t> a, e, out(m * n * N);
    for (const PolyVec& row : polynomials) flatten(row, N, a);
    flatten(PolyVec{element}, N, e);
    if (m * n) be.check(rzk_cmul_batch(be.ctx(), a.data(), (uint32_t)(m * n), e.data(), out.data(), 1));
    return reshape(out, m, n);
  }
  void extend_rows(const Mat& other) {                                          // mat.rs:186-196
    if (dim().second != other.dim().second) throw std::runtime_error("Mat::extend_rows: column counts differ");
    polynomials.insert(polynomials.end(), other.polynomials.begin(), other.polynomials.end());
  }
  std::pair<Mat, Mat> split_rows(size_t r) const {                              // mat.rs:203-213: (m - r) x n and r x n
    const size_t m = dim().first;
    if (r > m) throw std::runtime_error("Mat::split_rows: r > rows");
    return {Mat(std::vector<PolyVec>(polynomials.begin(), polynomials.begin() + (m - r))),
            Mat(std::vector<PolyVec>(polynomials.begin() + (m - r), polynomials.end()))};
  }
  void extend_cols(const Mat& other) {                                          // mat.rs:221-233
    if (dim().first != other.dim().first) throw std::runtime_error("Mat::extend_cols: row counts differ");
    for (size_t i = 0; i < polynomials.size(); ++i)
      polynomials[i].insert(polynomials[i].end(), other.polynomials[i].begin(), other.polynomials[i].end());
  }
  // derived PartialEq (mat.rs:11) on canonical forms: trailing zeros do not matter
  bool operator==(const Mat& o) const {
    if (dim() != o.dim()) return false;
    for (size_t i = 0; i < polynomials.size(); ++i)
      for (size_t j = 0; j < polynomials[i].size(); ++j)
        if (dense(polynomials[i][j], N) != dense(o.polynomials[i][j], N)) return false;
    return true;
  }

 private:
  static Mat reshape(const std::vector<int64_t>& flat, size_t m, size_t n) {
    const PolyVec v = unflatten(flat, N);
    Mat out(std::vector<PolyVec>(m, PolyVec(n)));
    for (size_t i = 0; i < m; ++i)
      for (size_t j = 0; j < n; ++j) out.polynomials[i][j] = v[i * n + j];
    return out;
  }
  Mat addsub(const Mat& other, const Backend<N>& be, bool subtract) const {
    if (dim() != other.dim()) throw std::runtime_error("Mat::add/sub: dimension mismatch (mat.rs:129-130, 154-155)");
    const auto [m, n] = dim();
    std::vector<int64_t> a, b, out(m * n * N);
    for (const PolyVec& row : polynomials) flatten(row, N, a);
    for (const PolyVec& row : other.polynomials) flatten(row, N, b);
    if (m * n)
      be.check(subtract ? rzk_sub_batch(be.ctx(), a.data(), b.data(), out.data(), m * n)
                        : rzk_add_batch(be.ctx(), a.data(), b.data(), out.data(), m * n));
    return reshape(out, m, n);
  }
};

// ---- samplers (host side, as in the reference) -------------------------------------------------------------
template <size_t N>
Poly random_polynomial_within(Rng& rng, int64_t bound) {   // polynomial.rs:14-25
  std::uniform_int_distribution<int64_t> d(-bound, bound);
  Poly p(N);
  for (auto& c : p) c = d(rng);
  return p;
}
template <size_t N>
Poly random_polynomial_in_normal_distribution(Rng& rng, double mean, double std_dev) {   // polynomial.rs:28-44
  std::normal_distribution<double> d(mean, std_dev);
  Poly p(N);
  for (auto& c : p) c = (int64_t)d(rng);
  return p;
}
template <size_t N>
Poly random_polynomial_from_challenge_set(Rng& rng, size_t kappa) {   // challenge_space.rs:12-33
  Poly p(N, 0);
  for (size_t i = 0; i < std::min(kappa, N); ++i) p[i] = (rng() & 1) ? 1 : -1;
  std::shuffle(p.begin(), p.end(), rng);
  return p;
}

// Set difference of the challenge space (challenge_space.rs:39-54, unused by the protocols): c - c' with c != c'
template <size_t N>
Poly random_polynomial_from_challenge_set_difference(Rng& rng, size_t kappa) {
  const Poly c1 = random_polynomial_from_challenge_set<N>(rng, kappa);
  for (;;) {
    const Poly c2 = random_polynomial_from_challenge_set<N>(rng, kappa);
    if (c1 != c2) {
      Poly d(N);
      for (size_t i = 0; i < N; ++i) d[i] = c1[i] - c2[i];
      return d;
    }
  }
}

// ---- norms (src/polynomial.rs:46-88): exact integers; norm_2 is the floor of the square root ---------------------
inline unsigned __int128 norm_1(const Poly& p) {
  unsigned __int128 s = 0;
  for (int64_t c : p) s += (unsigned __int128)(c < 0 ? -(__int128)c : (__int128)c);
  return s;
}
inline uint64_t norm_2(const Poly& p) {
  unsigned __int128 s = 0;
  for (int64_t c : p) s += (unsigned __int128)((__int128)c * c);
  // floor(sqrt(s)) for s < 2^128: Newton from above on 128-bit integers
  if (s == 0) return 0;
  unsigned __int128 x = (unsigned __int128)1 << 64, y = (x + s / x) >> 1;
  while (y < x) {
    x = y;
    y = (x + s / x) >> 1;
  }
  return (uint64_t)x;
}
inline uint64_t norm_infinity(const Poly& p) {
  uint64_t m = 0;
  for (int64_t c : p) m = std::max<uint64_t>(m, (uint64_t)(c < 0 ? -(__int128)c : (__int128)c));
  return m;
}

// ---- commitment scheme (src/commit.rs) ----------------------------------------------------------------------
template <size_t N>
struct Opening {
  PolyVec x, r;
  Poly f;         // empty = None (what CommitmentKey::commit produces, commit.rs:127); else the scalar of a relaxed opening
};

template <size_t N>
class CommitmentKey {
 public:
  // CommitmentKey::new (commit.rs:33-60); loads [a1;a2] into the backend (resident, NTT domain)
  CommitmentKey(Rng& rng, std::shared_ptr<Backend<N>> be) : be_(std::move(be)) {
    const Params& P = be_->params;
    const Poly zero(N, 0);
    Poly one(N, 0);
    one[0] = 1;
    for (size_t i = 0; i < P.n; ++i) {          // a1 = [I_n | a1']
      for (size_t j = 0; j < P.n; ++j) a.push_back(i == j ? one : zero);
      for (size_t j = P.n; j < P.k; ++j) a.push_back(random_polynomial_within<N>(rng, P.q));
    }
    for (size_t i = 0; i < P.l; ++i) {          // a2 = [0 | I_l | a2']
      for (size_t j = 0; j < P.n; ++j) a.push_back(zero);
      for (size_t j = 0; j < P.l; ++j) a.push_back(i == j ? one : zero);
      for (size_t j = P.n + P.l; j < P.k; ++j) a.push_back(random_polynomial_within<N>(rng, P.q));
    }
    std::vector<int64_t> flat;
    flatten(a, N, flat);
    be_->check(rzk_key_load(be_->ctx(), flat.data()));
  }

  // commit (commit.rs:88-128): resample r until check_commit_constraint holds, c = [a1;a2].r + [0;x]
  std::pair<Opening<N>, PolyVec> commit(Rng& rng, const PolyVec& x) const {
    const Params& P = be_->params;
    if (x.size() != P.l) throw std::runtime_error("commit: x.len() != l (commit.rs:95)");
    std::vector<int64_t> xf, cf((P.n + P.l) * N);
    flatten(x, N, xf);
    PolyVec r;
    std::vector<int64_t> rf;
    for (;;) {   // commit.rs:98-107: resample r until the constraint holds; the product rides along
      r.clear();
      for (size_t i = 0; i < P.k; ++i) r.push_back(random_polynomial_within<N>(rng, (int64_t)P.b));
      rf.clear();
      flatten(r, N, rf);
      uint8_t ok = 0;
      be_->check(rzk_commit_batch(be_->ctx(), xf.data(), rf.data(), cf.data(), &ok, 1));   // a.dot(&r).add(&z)
      if (ok) break;
    }
    return {Opening<N>{x, r, {}}, unflatten(cf, N)};
  }
  const std::shared_ptr<Backend<N>>& backend() const { return be_; }
  PolyVec a;   // (n+l)*k polynomials, row-major

 private:
  std::shared_ptr<Backend<N>> be_;
};

// Commitment::verify (commit.rs:173-210), both branches of the optional scalar f
template <size_t N>
bool commitment_verify(const PolyVec& c, const Opening<N>& o, const CommitmentKey<N>& ck) {
  const auto& be = ck.backend();
  std::vector<int64_t> rf, xf, cf;
  flatten(o.r, N, rf);
  flatten(o.x, N, xf);
  flatten(c, N, cf);
  Poly f = o.f;
  if (!f.empty()) f.resize(N, 0);   // trimmed representation -> dense
  uint8_t ok = 0;
  be->check(rzk_commitment_verify_batch(be->ctx(), cf.data(), xf.data(), rf.data(), f.empty() ? nullptr : f.data(), &ok, 1));
  return ok != 0;
}

// ---- OpenProof (src/prove/open.rs) ----------------------------------------------------------------------------
template <size_t N>
struct OpenProofResponseContext { Opening<N> opening; PolyVec y; };
template <size_t N>
struct OpenProofCommitment { PolyVec c, t; };
template <size_t N>
struct OpenProofVerificationContext { PolyVec c, t; Poly d; };
struct OpenProofChallenge { Poly d; };
struct OpenProofResponse { PolyVec z; };

template <size_t N>
class OpenProofProver {
 public:
  explicit OpenProofProver(const CommitmentKey<N>& ck) : ck_(ck) {}
  // open.rs:80-103
  std::pair<OpenProofResponseContext<N>, OpenProofCommitment<N>> commit(Rng& rng, const PolyVec& x) const {
    const auto& be = ck_.backend();
    const Params& P = be->params;
    auto oc = ck_.commit(rng, x);
    PolyVec y;
    for (size_t i = 0; i < P.k; ++i)
      y.push_back(random_polynomial_in_normal_distribution<N>(rng, 0.0, (double)P.standard_deviation(N)));
    std::vector<int64_t> yf, tf(P.n * N);
    flatten(y, N, yf);
    be->check(rzk_matvec_batch(be->ctx(), RZK_KEY_A1, yf.data(), nullptr, tf.data(), 1));   // t = a1.dot(&y)
    return {OpenProofResponseContext<N>{oc.first, y}, OpenProofCommitment<N>{oc.second, unflatten(tf, N)}};
  }
  // open.rs:107-117: z = y + r (.) d
  OpenProofResponse create_response(const OpenProofResponseContext<N>& ctx, const OpenProofChallenge& ch) const {
    const auto& be = ck_.backend();
    std::vector<int64_t> yf, rf, zf(be->params.k * N);
    flatten(ctx.y, N, yf);
    flatten(ctx.opening.r, N, rf);
    be->check(rzk_open_response_batch(be->ctx(), yf.data(), rf.data(), ch.d.data(), zf.data(), 1));
    return {unflatten(zf, N)};
  }

 private:
  const CommitmentKey<N>& ck_;
};

template <size_t N>
class OpenProofVerifier {
 public:
  explicit OpenProofVerifier(const CommitmentKey<N>& ck) : ck_(ck) {}
  // open.rs:143-158
  std::pair<OpenProofVerificationContext<N>, OpenProofChallenge> generate_challenge(
      Rng& rng, const OpenProofCommitment<N>& cm) const {
    Poly d = random_polynomial_from_challenge_set<N>(rng, ck_.backend()->params.kappa);
    return {OpenProofVerificationContext<N>{cm.c, cm.t, d}, OpenProofChallenge{d}};
  }
  // open.rs:162-174
  bool verify(const OpenProofResponse& resp, const OpenProofVerificationContext<N>& v) const {
    const auto& be = ck_.backend();
    std::vector<int64_t> zf, tf, cf;
    flatten(resp.z, N, zf);
    flatten(v.t, N, tf);
    flatten(v.c, N, cf);
    if (zf.size() != be->params.k * N) throw std::runtime_error("verify: response has the wrong shape");
    uint8_t acc = 0;
    be->check(rzk_open_verify_batch(be->ctx(), zf.data(), tf.data(), cf.data(), v.d.data(), &acc, 1));
    return acc != 0;
  }

 private:
  const CommitmentKey<N>& ck_;
};

// ---- LinearProof (src/prove/linear.rs) ------------------------------------------------------------------------------
template <size_t N>
struct LinearProofResponseContext { Opening<N> opening, opening_p; PolyVec y, yp; };
template <size_t N>
struct LinearProofCommitment { PolyVec c, cp; Poly g; PolyVec t, tp, u; };
template <size_t N>
struct LinearProofVerificationContext { LinearProofCommitment<N> cm; Poly d; };
struct LinearProofChallenge { Poly d; };
struct LinearProofResponse { PolyVec z, zp; };

template <size_t N>
class LinearProofProver {
 public:
  explicit LinearProofProver(const CommitmentKey<N>& ck) : ck_(ck) {}
  // linear.rs:82-140
  std::pair<LinearProofResponseContext<N>, LinearProofCommitment<N>> commit(Rng& rng, const Poly& g_in,
                                                                          const PolyVec& x) const {
    const Poly g = dense(g_in, N);
    const auto& be = ck_.backend();
    const Params& P = be->params;
    if (x.size() != P.l) throw std::runtime_error("commit: x.len() != l");
    std::vector<int64_t> xf, gxf(P.l * N);
    flatten(x, N, xf);
    be->check(rzk_cmul_batch(be->ctx(), xf.data(), (uint32_t)P.l, g.data(), gxf.data(), 1));   // gx = x_i * g
    auto ocp = ck_.commit(rng, unflatten(gxf, N));
    auto oc = ck_.commit(rng, x);
    PolyVec y, yp;
    const double sd = (double)P.standard_deviation(N);
    for (size_t i = 0; i < P.k; ++i) y.push_back(random_polynomial_in_normal_distribution<N>(rng, 0.0, sd));
    for (size_t i = 0; i < P.k; ++i) yp.push_back(random_polynomial_in_normal_distribution<N>(rng, 0.0, sd));
    std::vector<int64_t> yf, ypf, tf(P.n * N), tpf(P.n * N), a2y(P.l * N), a2yp(P.l * N), uf(P.l * N);
    flatten(y, N, yf);
    flatten(yp, N, ypf);
    be->check(rzk_matvec_batch(be->ctx(), RZK_KEY_A1, yf.data(), nullptr, tf.data(), 1));
    be->check(rzk_matvec_batch(be->ctx(), RZK_KEY_A1, ypf.data(), nullptr, tpf.data(), 1));
    // u = (a2.y) (.) g - a2.yp  (linear.rs:124-129), composed from the Mat primitives
    be->check(rzk_matvec_batch(be->ctx(), RZK_KEY_A2, yf.data(), nullptr, a2y.data(), 1));
    be->check(rzk_cmul_batch(be->ctx(), a2y.data(), (uint32_t)P.l, g.data(), uf.data(), 1));
    be->check(rzk_matvec_batch(be->ctx(), RZK_KEY_A2, ypf.data(), nullptr, a2yp.data(), 1));
    be->check(rzk_sub_batch(be->ctx(), uf.data(), a2yp.data(), uf.data(), P.l));
    return {LinearProofResponseContext<N>{oc.first, ocp.first, y, yp},
            LinearProofCommitment<N>{oc.second, ocp.second, g, unflatten(tf, N), unflatten(tpf, N), unflatten(uf, N)}};
  }
  // linear.rs:144-158
  LinearProofResponse create_response(const LinearProofResponseContext<N>& ctx, const LinearProofChallenge& ch) const {
    const auto& be = ck_.backend();
    const size_t kN = be->params.k * N;
    std::vector<int64_t> yf, ypf, rf, rpf, zf(kN), zpf(kN);
    flatten(ctx.y, N, yf);
    flatten(ctx.yp, N, ypf);
    flatten(ctx.opening.r, N, rf);
    flatten(ctx.opening_p.r, N, rpf);
    be->check(rzk_linear_response_batch(be->ctx(), yf.data(), ypf.data(), rf.data(), rpf.data(), ch.d.data(),
                                        zf.data(), zpf.data(), 1));
    return {unflatten(zf, N), unflatten(zpf, N)};
  }

 private:
  const CommitmentKey<N>& ck_;
};

template <size_t N>
class LinearProofVerifier {
 public:
  explicit LinearProofVerifier(const CommitmentKey<N>& ck) : ck_(ck) {}
  std::pair<LinearProofVerificationContext<N>, LinearProofChallenge> generate_challenge(
      Rng& rng, const LinearProofCommitment<N>& cm) const {   // linear.rs:184-209
    Poly d = random_polynomial_from_challenge_set<N>(rng, ck_.backend()->params.kappa);
    return {LinearProofVerificationContext<N>{cm, d}, LinearProofChallenge{d}};
  }
  bool verify(const LinearProofResponse& r, const LinearProofVerificationContext<N>& v) const {   // linear.rs:213-250
    const auto& be = ck_.backend();
    std::vector<int64_t> zf, zpf, cf, cpf, tf, tpf, uf;
    flatten(r.z, N, zf);
    flatten(r.zp, N, zpf);
    flatten(v.cm.c, N, cf);
    flatten(v.cm.cp, N, cpf);
    flatten(v.cm.t, N, tf);
    flatten(v.cm.tp, N, tpf);
    flatten(v.cm.u, N, uf);
    uint8_t acc = 0;
    be->check(rzk_linear_verify_batch(be->ctx(), zf.data(), zpf.data(), cf.data(), cpf.data(), v.cm.g.data(), tf.data(),
                                      tpf.data(), uf.data(), v.d.data(), &acc, 1));
    return acc != 0;
  }

 private:
  const CommitmentKey<N>& ck_;
};

// ---- SumProof (src/prove/sum.rs) -------------------------------------------------------------------------------------
template <size_t N>
struct SumProofResponseContext { std::vector<Opening<N>> openings; Opening<N> opening_p; PolyVec yp; std::vector<PolyVec> ys; };
template <size_t N>
struct SumProofCommitment { PolyVec cp; std::vector<PolyVec> cs; PolyVec gs, tp; std::vector<PolyVec> ts; PolyVec u; };
template <size_t N>
struct SumProofVerificationContext { SumProofCommitment<N> cm; Poly d; };
struct SumProofChallenge { Poly d; };
struct SumProofResponse { PolyVec zp; std::vector<PolyVec> zs; };

template <size_t N>
class SumProofProver {
 public:
  explicit SumProofProver(const CommitmentKey<N>& ck) : ck_(ck) {}
  // sum.rs:99-178.  The commitments use the host sampler per commitment (like the reference); the
  // algebra of xp, ts, tp, u is one call into the fused batch entry point with the sampled randomness.
  std::pair<SumProofResponseContext<N>, SumProofCommitment<N>> commit(Rng& rng, const PolyVec& gs,
                                                                    const std::vector<PolyVec>& xs) const {
    const auto& be = ck_.backend();
    const Params& P = be->params;
    if (gs.empty() || gs.size() != xs.size()) throw std::runtime_error("commit: gs empty or gs.len() != xs.len() (sum.rs:105)");
    const size_t V = gs.size();
    for (const Poly& gi : gs)
      if (gi.size() > N) throw std::runtime_error("polynomial longer than the ring degree");
    auto sample_r = [&]() {
      for (;;) {
        PolyVec r;
        for (size_t i = 0; i < P.k; ++i) r.push_back(random_polynomial_within<N>(rng, (int64_t)P.b));
        std::vector<int64_t> rf;
        flatten(r, N, rf);
        uint8_t ok = 0;
        be->check(rzk_norm2_le_batch(be->ctx(), rf.data(), (uint32_t)P.k, rzk_commit_bound(be->ctx()), &ok, 1));
        if (ok) return r;
      }
    };
    PolyVec rp = sample_r();
    std::vector<PolyVec> rs;
    for (size_t i = 0; i < V; ++i) rs.push_back(sample_r());
    const double sd = (double)P.standard_deviation(N);
    std::vector<PolyVec> ys(V);
    for (auto& y : ys)
      for (size_t i = 0; i < P.k; ++i) y.push_back(random_polynomial_in_normal_distribution<N>(rng, 0.0, sd));
    PolyVec yp;
    for (size_t i = 0; i < P.k; ++i) yp.push_back(random_polynomial_in_normal_distribution<N>(rng, 0.0, sd));
    std::vector<int64_t> gsf, xsf, rsf, rpf, ysf, ypf;
    flatten(gs, N, gsf);
    for (const auto& x : xs) {
      if (x.size() != P.l) throw std::runtime_error("commit: x_i.len() != l");
      flatten(x, N, xsf);
    }
    for (const auto& r : rs) flatten(r, N, rsf);
    flatten(rp, N, rpf);
    for (const auto& y : ys) flatten(y, N, ysf);
    flatten(yp, N, ypf);
    const size_t nl = P.n + P.l;
    std::vector<int64_t> csf(V * nl * N), cpf(nl * N), tsf(V * P.n * N), tpf(P.n * N), uf(P.l * N);
    uint8_t ok = 0;
    be->check(rzk_sum_commit_batch(be->ctx(), (uint32_t)V, gsf.data(), xsf.data(), rsf.data(), rpf.data(), ysf.data(),
                                   ypf.data(), csf.data(), cpf.data(), tsf.data(), tpf.data(), uf.data(), &ok, 1));
    SumProofResponseContext<N> rc;
    SumProofCommitment<N> cm;
    // xp is not revealed by the fused call; recompute it for the opening of cp from the Mat primitives
    std::vector<int64_t> xpf(P.l * N, 0), tmp(P.l * N);
    for (size_t i = 0; i < V; ++i) {
      be->check(rzk_cmul_batch(be->ctx(), xsf.data() + i * P.l * N, (uint32_t)P.l, gsf.data() + i * N, tmp.data(), 1));
      if (i == 0) xpf = tmp; else be->check(rzk_add_batch(be->ctx(), xpf.data(), tmp.data(), xpf.data(), P.l));
    }
    rc.opening_p = Opening<N>{unflatten(xpf, N), rp};
    for (size_t i = 0; i < V; ++i) rc.openings.push_back(Opening<N>{xs[i], rs[i]});
    rc.yp = yp;
    rc.ys = ys;
    cm.cp = unflatten(cpf, N);
    cm.gs = gs;
    cm.tp = unflatten(tpf, N);
    cm.u = unflatten(uf, N);
    for (size_t i = 0; i < V; ++i) {
      cm.cs.push_back(unflatten(std::vector<int64_t>(csf.begin() + i * nl * N, csf.begin() + (i + 1) * nl * N), N));
      cm.ts.push_back(unflatten(std::vector<int64_t>(tsf.begin() + i * P.n * N, tsf.begin() + (i + 1) * P.n * N), N));
    }
    return {rc, cm};
  }
  SumProofResponse create_response(const SumProofResponseContext<N>& ctx, const SumProofChallenge& ch) const {   // sum.rs:182-200
    const auto& be = ck_.backend();
    const Params& P = be->params;
    const size_t V = ctx.ys.size();
    std::vector<int64_t> ysf, ypf, rsf, rpf, zsf(V * P.k * N), zpf(P.k * N);
    for (const auto& y : ctx.ys) flatten(y, N, ysf);
    flatten(ctx.yp, N, ypf);
    for (const auto& o : ctx.openings) flatten(o.r, N, rsf);
    flatten(ctx.opening_p.r, N, rpf);
    be->check(rzk_sum_response_batch(be->ctx(), (uint32_t)V, ysf.data(), ypf.data(), rsf.data(), rpf.data(), ch.d.data(),
                                     zsf.data(), zpf.data(), 1));
    SumProofResponse r;
    r.zp = unflatten(zpf, N);
    for (size_t i = 0; i < V; ++i)
      r.zs.push_back(unflatten(std::vector<int64_t>(zsf.begin() + i * P.k * N, zsf.begin() + (i + 1) * P.k * N), N));
    return r;
  }

 private:
  const CommitmentKey<N>& ck_;
};

template <size_t N>
class SumProofVerifier {
 public:
  explicit SumProofVerifier(const CommitmentKey<N>& ck) : ck_(ck) {}
  std::pair<SumProofVerificationContext<N>, SumProofChallenge> generate_challenge(Rng& rng,
                                                                                  const SumProofCommitment<N>& cm) const {
    Poly d = random_polynomial_from_challenge_set<N>(rng, ck_.backend()->params.kappa);   // sum.rs:228-253
    return {SumProofVerificationContext<N>{cm, d}, SumProofChallenge{d}};
  }
  bool verify(const SumProofResponse& r, const SumProofVerificationContext<N>& v) const {   // sum.rs:257-320
    const auto& be = ck_.backend();
    const size_t V = r.zs.size();
    // sum.rs:273 rejects only when BOTH lengths differ (Q4); the dense ABI needs equal lengths, so
    // any length mismatch is a rejection here.
    if (V == 0 || V != v.cm.ts.size() || V != v.cm.cs.size() || V != v.cm.gs.size()) return false;
    std::vector<int64_t> zsf, zpf, csf, cpf, gsf, tsf, tpf, uf;
    for (const auto& z : r.zs) flatten(z, N, zsf);
    flatten(r.zp, N, zpf);
    for (const auto& c : v.cm.cs) flatten(c, N, csf);
    flatten(v.cm.cp, N, cpf);
    flatten(v.cm.gs, N, gsf);
    for (const auto& t : v.cm.ts) flatten(t, N, tsf);
    flatten(v.cm.tp, N, tpf);
    flatten(v.cm.u, N, uf);
    uint8_t acc = 0;
    be->check(rzk_sum_verify_batch(be->ctx(), (uint32_t)V, zsf.data(), zpf.data(), csf.data(), cpf.data(), gsf.data(),
                                   tsf.data(), tpf.data(), uf.data(), v.d.data(), &acc, 1));
    return acc != 0;
  }

 private:
  const CommitmentKey<N>& ck_;
};

}  // namespace ring_zk
