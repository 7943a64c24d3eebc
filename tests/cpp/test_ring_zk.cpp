// C++ mirror of the reference's integration tests (tests/test.rs:11-93): for each proof type, random
// (ck, x[, g]) over many iterations with a fresh key each time; assert Commitment::verify and the
// protocol's verify return true; plus the negative checks the reference has in its doctests
// (commit.rs:169-170: a mismatched opening is rejected) and a tampered response per proof type.
// Runs on the GPU through ring_zk_amd/host/ring_zk.hpp -> librzk_hip.so.
//
// build: g++ -O2 -std=c++17 tests/cpp/test_ring_zk.cpp -Lring_zk_amd -lrzk_hip -Wl,-rpath,$PWD/ring_zk_amd -o /tmp/test_ring_zk
#include <cstdio>
#include <cstdlib>

#include "../../ring_zk_amd/host/ring_zk.hpp"

using namespace ring_zk;

#ifndef TEST_N
#define TEST_N 512
#endif
constexpr size_t N = TEST_N;   // the reference uses N = 16 (tests/test.rs:8); pytest builds this for 16 and 512

#define REQUIRE(cond)                                                       \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                         \
    }                                                                       \
  } while (0)

// tests/test.rs:95-99: coefficients uniform in [-bound, bound], random length 1..=N (shorter = trimmed)
static Poly random_value(Rng& rng, int64_t bound) {
  Poly p = random_polynomial_within<N>(rng, bound);
  std::uniform_int_distribution<size_t> len(1, N);
  p.resize(len(rng));
  return p;
}

// Params::default() (tests/test.rs), or the shape given in TEST_SHAPE = "n,k,l" (larger shapes drive the
// row-group / row-block / shared-operand kernels through the same reference-style flows)
static Params shape_params() {
  Params p;
  if (const char* e = std::getenv("TEST_SHAPE")) {
    unsigned n = 0, k = 0, l = 0;
    if (std::sscanf(e, "%u,%u,%u", &n, &k, &l) == 3) {
      p.n = n;
      p.k = k;
      p.l = l;
    }
  }
  return p;
}
static PolyVec random_message(Rng& rng, const Params& p) {
  PolyVec x;
  for (size_t i = 0; i < p.l; ++i) x.push_back(random_value(rng, p.q));
  return x;
}

static void test_open_proof(int iters) {
  Rng rng(1);
  Params params = shape_params();
  auto be = std::make_shared<Backend<N>>(params);
  for (int it = 0; it < iters; ++it) {
    CommitmentKey<N> ck(rng, be);
    PolyVec x = random_message(rng, params);
    OpenProofProver<N> prover(ck);
    OpenProofVerifier<N> verifier(ck);
    auto [rctx, cm] = prover.commit(rng, x);
    REQUIRE(commitment_verify(cm.c, rctx.opening, ck));
    auto [vctx, ch] = verifier.generate_challenge(rng, cm);
    auto resp = prover.create_response(rctx, ch);
    REQUIRE(verifier.verify(resp, vctx));
    if (it < 3) {
      auto bad = resp;
      bad.z[1][it] += 1;
      REQUIRE(!verifier.verify(bad, vctx));
      Opening<N> wrong = rctx.opening;          // commit.rs:169-170: mismatched opening is rejected
      wrong.x[0][0] += wrong.x[0][0] > 0 ? -1 : 1;
      REQUIRE(!commitment_verify(cm.c, wrong, ck));
      // relaxed opening (commit.rs:199-206): f*c == a.(f*r) + f*[0;x] with f = 2 and r' = 2r
      Opening<N> relaxed = rctx.opening;
      relaxed.f = Poly{2};
      REQUIRE(!commitment_verify(cm.c, relaxed, ck));   // r not scaled yet
      for (auto& poly : relaxed.r)
        for (auto& coef : poly) coef *= 2;
      REQUIRE(commitment_verify(cm.c, relaxed, ck));
    }
  }
  std::printf("test_open_proof ok (%d iterations, N=%zu)\n", iters, N);
}

static void test_linear_proof(int iters) {
  Rng rng(2);
  Params params = shape_params();
  auto be = std::make_shared<Backend<N>>(params);
  for (int it = 0; it < iters; ++it) {
    CommitmentKey<N> ck(rng, be);
    PolyVec x = random_message(rng, params);
    Poly g = random_value(rng, params.q);
    g.resize(N, 0);
    LinearProofProver<N> prover(ck);
    LinearProofVerifier<N> verifier(ck);
    auto [rctx, cm] = prover.commit(rng, g, x);
    REQUIRE(commitment_verify(cm.c, rctx.opening, ck));
    REQUIRE(commitment_verify(cm.cp, rctx.opening_p, ck));
    auto [vctx, ch] = verifier.generate_challenge(rng, cm);
    auto resp = prover.create_response(rctx, ch);
    REQUIRE(verifier.verify(resp, vctx));
    if (it < 3) {
      auto bad = resp;
      bad.zp[2][7 % N] -= 1;
      REQUIRE(!verifier.verify(bad, vctx));
    }
  }
  std::printf("test_linear_proof ok (%d iterations)\n", iters);
}

static void test_sum_proof(int iters) {
  Rng rng(3);
  Params params = shape_params();
  constexpr size_t VL = 4;   // tests/test.rs:65
  auto be = std::make_shared<Backend<N>>(params);
  for (int it = 0; it < iters; ++it) {
    CommitmentKey<N> ck(rng, be);
    std::vector<PolyVec> xs;
    PolyVec gs;
    for (size_t i = 0; i < VL; ++i) xs.push_back(random_message(rng, params));
    for (size_t i = 0; i < VL; ++i) {
      Poly g = random_value(rng, params.q);
      g.resize(N, 0);
      gs.push_back(g);
    }
    SumProofProver<N> prover(ck);
    SumProofVerifier<N> verifier(ck);
    auto [rctx, cm] = prover.commit(rng, gs, xs);
    REQUIRE(commitment_verify(cm.cp, rctx.opening_p, ck));
    for (size_t i = 0; i < VL; ++i) REQUIRE(commitment_verify(cm.cs[i], rctx.openings[i], ck));
    auto [vctx, ch] = verifier.generate_challenge(rng, cm);
    auto resp = prover.create_response(rctx, ch);
    REQUIRE(verifier.verify(resp, vctx));
    if (it < 3) {
      auto bad = resp;
      bad.zs[VL - 1][0][3] += 2;
      REQUIRE(!verifier.verify(bad, vctx));
      auto shorter = resp;
      shorter.zs.pop_back();
      REQUIRE(!verifier.verify(shorter, vctx));
    }
  }
  std::printf("test_sum_proof ok (%d iterations, VL=%zu)\n", iters, VL);
}

static void test_panics() {
  Rng rng(4);
  Params params;
  auto be = std::make_shared<Backend<N>>(params);
  CommitmentKey<N> ck(rng, be);
  OpenProofProver<N> prover(ck);
  bool threw = false;
  try {
    prover.commit(rng, PolyVec{});   // commit.rs:95: assert_eq!(l, x.len())
  } catch (const std::runtime_error&) {
    threw = true;
  }
  REQUIRE(threw);
  threw = false;
  try {
    Params bad;
    bad.n = 3;   // k > n violated
    Backend<N> b2(bad);
  } catch (const std::runtime_error&) {
    threw = true;
  }
  REQUIRE(threw);
  std::printf("test_panics ok\n");
}

// src/mat.rs:241-422: the reference's Mat tests with their literal polynomials (ring degree 4 there; the
// polynomials have degree <= 2, and the expected products are the golden vectors of tests/golden/golden.json)
static void test_mat() {
  Params params;
  Backend<N> be(params);
  const Poly a00{1, 2, 3}, a01{4, 5, 6}, b00{1, 2}, b10{3, 4};
  const Mat<N> a({{a00, a01}});            // 1 x 2
  const Mat<N> b({{b00}, {b10}});          // 2 x 1
  const Mat<N> c = a.dot(b, be);           // mat.rs:243-268
  REQUIRE(c.dim() == std::make_pair(size_t(1), size_t(1)));
  if (N == 4) REQUIRE(c == Mat<N>({{Poly{13, 35, 45, 30}}}));                    // no wrap-around
  else REQUIRE(c == Mat<N>({{Poly{13, 35, 45, 30}}}));                           // degree 3 < N: same integers
  const Mat<N> col({{a00}, {a01}});        // 2 x 1
  REQUIRE(col.add(col, be) == Mat<N>({{Poly{2, 4, 6}}, {Poly{8, 10, 12}}}));     // mat.rs:271-297
  REQUIRE(col.sub(col, be) == Mat<N>({{Poly{}}, {Poly{}}}));                     // mat.rs:300-326
  Mat<N> ext = Mat<N>::from_vec({a00});
  ext.extend_rows(Mat<N>::from_vec({a01}));                                      // mat.rs:329-355
  REQUIRE(ext == col);
  Mat<N> wide({{a00}});
  wide.extend_cols(Mat<N>({{a01}}));                                             // mat.rs:358-386
  REQUIRE(wide == a);
  const Mat<N> sq = col.componentwise_mul(a00, be);                              // mat.rs:389-406
  if (N == 4) REQUIRE(sq == Mat<N>({{Poly{-8, 4, 10, 12}}, {Poly{-14, 13, 28, 27}}}));   // x^4 = -1
  else REQUIRE(sq == Mat<N>({{Poly{1, 4, 10, 12, 9}}, {Poly{4, 13, 28, 27, 18}}}));      // plain products for N > 4
  const auto [top, bottom] = col.split_rows(1);                                  // mat.rs:409-422
  REQUIRE(top == Mat<N>({{a00}}) && bottom == Mat<N>({{a01}}));
  REQUIRE(Mat<N>::from_element(2, 3, a00).dim() == std::make_pair(size_t(2), size_t(3)));
  REQUIRE(Mat<N>::diag(2, 2, a00).polynomials[0][1].empty());
  REQUIRE(col.one_d_mat_to_vec().size() == 2);
  bool threw = false;
  try {
    (void)a.dot(a, be);                    // 1x2 . 1x2: mat.rs:103 panics
  } catch (const std::runtime_error&) {
    threw = true;
  }
  REQUIRE(threw);
  std::printf("test_mat ok\n");
}

// src/polynomial.rs:92-132 and src/challenge_space.rs:56-82: samplers and norms
static void test_polynomial_and_challenge_space() {
  Rng rng(5);
  const Poly w = random_polynomial_within<4>(rng, 10);                      // polynomial.rs:98-105
  for (int64_t c : w) REQUIRE(-10 <= c && c <= 10);
  const Poly p{1, -2, 3, -4};
  REQUIRE(norm_1(p) == 10 && norm_2(p) == 5 && norm_infinity(p) == 4);      // polynomial.rs:107-123
  REQUIRE(norm_2(Poly{3, 4}) == 5 && norm_2(Poly{}) == 0 && norm_2(Poly{1, 1}) == 1);
  const int64_t half = (Params().modulus - 1) / 2;
  REQUIRE(norm_2(Poly(2048, half)) == 79542997364ull);                     // floor(sqrt(2048 * half^2)): sum of squares 2^72.4, needs 128 bits
  const Poly g = random_polynomial_in_normal_distribution<256>(rng, 0.0, 1000.0);   // polynomial.rs:125-131
  REQUIRE(g.size() == 256 && norm_infinity(g) < 8000);
  const size_t kappa = 60;                                                   // challenge_space.rs:60-68
  const Poly c = random_polynomial_from_challenge_set<256>(rng, kappa);
  REQUIRE(norm_1(c) == kappa && norm_infinity(c) == 1);
  const Poly dd = random_polynomial_from_challenge_set_difference<256>(rng, kappa);   // challenge_space.rs:70-80
  for (int64_t v : dd) REQUIRE(v >= -2 && v <= 2);
  std::printf("test_polynomial_and_challenge_space ok\n");
}

// src/params.rs:145-168: standard_deviation KAT, prepare_scalar / prepare_value
static void test_params() {
  Params params;
  REQUIRE(params.standard_deviation(1024) == 21780);   // params.rs:149
  const Poly p = params.prepare_scalar<4>({1, 2, 3, 4});
  REQUIRE(p.size() == 4 && p[3] == 4);                   // deg() == 3 (params.rs:157)
  const PolyVec v = params.prepare_value<4>({{1, 2, 3, 4}});
  REQUIRE(v.size() == 1 && v[0].size() == 4);            // params.rs:165-166
  REQUIRE(params.prepare_scalar<4>({params.modulus + 5})[0] == 5);       // Into<ZqI64>: reduced mod Q
  REQUIRE(params.prepare_scalar<4>({(params.modulus - 1) / 2 + 1})[0] == -(params.modulus - 1) / 2);
  bool threw = false;
  try {
    (void)params.prepare_value<4>({{1}, {2}});           // value.len() != l panics (params.rs:71)
  } catch (const std::runtime_error&) {
    threw = true;
  }
  REQUIRE(threw);
  std::printf("test_params ok\n");
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 100;   // tests/test.rs: 100 iterations each
  test_params();
  test_polynomial_and_challenge_space();
  test_mat();
  if (const char* only = std::getenv("TEST_ONLY")) {   // "mat": the Mat / Params unit tests alone (N = 4, mat.rs:241)
    if (std::string(only) == "mat") {
      std::printf("all ok\n");
      return 0;
    }
  }
  test_open_proof(iters);
  test_linear_proof(iters);
  test_sum_proof(iters);
  test_panics();
  std::printf("all ok\n");
  return 0;
}
