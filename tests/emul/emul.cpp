// CPU lane emulator for ring_zk_amd/csrc/rzk_core.h (TEST INFRASTRUCTURE — not part of the product).
// Replays the 64 lanes of one wavefront phase by phase, with the wave-private LDS buffer as a plain
// array, so the register/LDS index arithmetic and the modular arithmetic of the HIP kernels can be
// checked bit-for-bit against the oracle in a container without a GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../ring_zk_amd/csrc/rzk_core.h"
#include "../../ring_zk_amd/csrc/rzk_tables.h"
#include "../../ring_zk_amd/csrc/rzk_rng.h"

using namespace rzk;

namespace {

struct Tables {
  std::vector<uint32_t> fwd[kMaxPrimes], inv[kMaxPrimes];
  Tables() {
    for (int i = 0; i < kMaxPrimes; ++i) host::make_twiddles(i, fwd[i], inv[i]);
  }
};
const Tables& tables() {
  static Tables t;
  return t;
}

// A team of 2^LL threads (LL = 6: one wavefront, LL = 7: two wavefronts sharing the slab) replayed thread by thread;
// every phase boundary is a team-wide barrier in the kernels, here simply the end of a loop over the threads.
template <int LOGN, int LL = 6>
struct Wave {
  using G = Geo<LOGN, LL>;
  static constexpr int T = G::LANES;
  uint32_t x[T][G::E];
  uint32_t lds[G::LDS_WORDS];

  void forward(const uint32_t* tw, const PrimeConsts& pc) {   // x in phase-1 layout -> phase-3 layout
    for (int l = 0; l < T; ++l) { fwd_phase1<LOGN, LL>(x[l], tw, pc); lds_put_p1<LOGN, LL>(x[l], l, lds); }
    for (int l = 0; l < T; ++l) { lds_get_p2<LOGN, LL>(x[l], l, lds); fwd_phase2<LOGN, LL>(x[l], l, tw, pc); }
    for (int l = 0; l < T; ++l) lds_put_p2<LOGN, LL>(x[l], l, lds);
    for (int l = 0; l < T; ++l) { lds_get_p3<LOGN, LL>(x[l], l, lds); fwd_phase3<LOGN, LL>(x[l], l, tw, pc); }
  }
  void inverse(const uint32_t* tw, const PrimeConsts& pc) {   // phase-3 layout -> phase-1 layout
    for (int l = 0; l < T; ++l) { inv_phase3<LOGN, LL>(x[l], l, tw, pc); lds_put_p3<LOGN, LL>(x[l], l, lds); }
    for (int l = 0; l < T; ++l) { lds_get_p2<LOGN, LL>(x[l], l, lds); inv_phase2<LOGN, LL>(x[l], l, tw, pc); }
    for (int l = 0; l < T; ++l) lds_put_p2<LOGN, LL>(x[l], l, lds);
    for (int l = 0; l < T; ++l) { lds_get_p1<LOGN, LL>(x[l], l, lds); inv_phase1<LOGN, LL>(x[l], tw, pc); }
  }
};

template <int LOGN, int LL = 6>
int ntt_fwd(int pi, const uint32_t* in, uint32_t* out_std, uint32_t* out_mem) {
  using G = Geo<LOGN, LL>;
  PrimeConsts pc = host::make_prime_consts(pi, G::N);
  static Wave<LOGN, LL> w;
  for (int l = 0; l < G::LANES; ++l)
    for (int e = 0; e < G::E; ++e) w.x[l][e] = in[G::j_p1(l, e)];
  w.forward(tables().fwd[pi].data(), pc);
  for (int l = 0; l < G::LANES; ++l)
    for (int c = 0; c < G::E; ++c) {
      uint32_t v = csub(csub(w.x[l][c], pc.twop), pc.p);
      out_std[G::j_p3(l, c)] = v;
      out_mem[G::mem_p3(l, c)] = v;
    }
  // the 16-byte group index must address the same words
  for (int l = 0; l < G::LANES; ++l)
    for (int g = 0; g < G::E / 4; ++g)
      if (G::key4(l, g) * 4 != G::mem_p3(l, 4 * g) || G::mem_p3(l, 4 * g + 3) != G::mem_p3(l, 4 * g) + 3) return -3;
  return 0;
}

template <int LOGN, int LL = 6>
int ntt_inv(int pi, const uint32_t* in_std, uint32_t* out) {
  using G = Geo<LOGN, LL>;
  PrimeConsts pc = host::make_prime_consts(pi, G::N);
  static Wave<LOGN, LL> w;
  for (int l = 0; l < G::LANES; ++l)
    for (int c = 0; c < G::E; ++c) w.x[l][c] = in_std[G::j_p3(l, c)];
  w.inverse(tables().inv[pi].data(), pc);
  for (int l = 0; l < G::LANES; ++l)
    for (int e = 0; e < G::E; ++e)
      out[G::j_p1(l, e)] = csub(mont_lazy(w.x[l][e], pc.ninv_r, pc.p, pc.npinv), pc.p);
  return 0;
}

// out = a * b in Z_q[X]/(X^N+1), centred; exact via np primes
template <int LOGN>
int polymul(int np, uint64_t q, const int64_t* a, const int64_t* b, int64_t* out) {
  using G = Geo<LOGN>;
  CrtConsts C;
  if (!host::make_crt_consts(q, C)) return -1;
  PrimeConsts pc[kMaxPrimes];
  for (int i = 0; i < kMaxPrimes; ++i) pc[i] = host::make_prime_consts(i, G::N);
  static Wave<LOGN> wa, wb;
  static uint32_t res[kMaxPrimes][64][G::E];
  std::memset(res, 0, sizeof(res));
  for (int pi = 0; pi < np; ++pi) {
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < G::E; ++e) {
        wa.x[l][e] = lift((int32_t)a[G::j_p1(l, e)], pc[pi]);
        wb.x[l][e] = lift((int32_t)b[G::j_p1(l, e)], pc[pi]);
      }
    wa.forward(tables().fwd[pi].data(), pc[pi]);
    wb.forward(tables().fwd[pi].data(), pc[pi]);
    for (int l = 0; l < 64; ++l)
      for (int c = 0; c < G::E; ++c) {
        uint32_t bs = mont_lazy(wb.x[l][c], pc[pi].ninv_r2, pc[pi].p, pc[pi].npinv);  // b^ * N^-1 * R
        wa.x[l][c] = mac_add(0, wa.x[l][c], csub(bs, pc[pi].p), pc[pi]);
      }
    wa.inverse(tables().inv[pi].data(), pc[pi]);
    std::memcpy(res[pi], wa.x, sizeof(wa.x));
  }
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < G::E; ++e) {
      // both reconstruction forms must agree: sign-test form and the incremental offset form the kernels use
      const int64_t v1 = crt_center(res[0][l][e], res[1][l][e], res[2][l][e], np, pc, C);
      uint32_t stA = crt_fold0(res[0][l][e], np, pc, C), stB = 0;
      if (np >= 2) crt_fold1(res[1][l][e], np, pc, C, stA, stB);
      if (np >= 3) crt_fold2(res[2][l][e], pc, C, stA, stB);
      const int64_t v2 = crt_finish(stA, np, C);
      if (v1 != v2) return -2;
      // ... and the sign-test forms of the one- and two-prime rows (unit_kernel)
      if (np == 1 && center_from_zq(crt1_zq(res[0][l][e], pc, C), C) != v1) return -4;
      if (np == 2 && center_from_zq(crt2_zq(res[1][l][e], crt2_digit0(res[0][l][e], pc), pc, C), C) != v1) return -5;
      out[G::j_p1(l, e)] = v2;
    }
  return 0;
}

// out = d * v in Z_q[X]/(X^N+1), centred, by the shift-add scheme of the challenge products (ShiftGeo):
// passes = 1: whole values, 2: two 16-bit halves; outputs accumulated half a lane at a time as in the kernels.
template <int LOGN, bool PAIR, int LL = 6>
int shift_product(int passes, uint64_t q, const int64_t* d, const int64_t* v, int64_t* out) {
  using S = ShiftGeo<LOGN, PAIR, LL>;
  constexpr int H = S::E / 2;
  constexpr int LANES = S::LANES;
  CrtConsts C;
  if (!host::make_crt_consts(q, C)) return -1;
  static int32_t ext[S::WORDS];
  static int32_t vr[LANES][S::E];
  static uint32_t tr[LANES][S::E];
  for (int l = 0; l < LANES; ++l)
    for (int i = 0; i < S::E; ++i) vr[l][i] = (int32_t)v[S::j(l, i)];
  for (int pass = 0; pass < passes; ++pass) {
    const int part = passes == 2 ? (pass == 0 ? SHIFT_LOW16 : SHIFT_HIGH16) : SHIFT_WHOLE;
    for (int l = 0; l < LANES; ++l) shift_fill<LOGN, PAIR, LL>(vr[l], l, ext, part);
    for (int l = 0; l < LANES; ++l)
      for (int half = 0; half < 2; ++half) {
        int64_t acc[H] = {0};
        for (int s = 0; s < S::N; ++s) {
          const int32_t coef = (int32_t)d[s];
          if (coef != 0) shift_accum<LOGN, PAIR, int64_t, 0, H, LL>(acc, l, s, coef, ext + half * (S::N / 2));
        }
        for (int i = 0; i < H; ++i) {
          const uint32_t u = zq_from_i64(acc[i], C);
          uint32_t& t = tr[l][half * H + i];
          t = pass == 0 ? u : addq(t, montq_u(u, C.r48q, C), C.q);
        }
      }
  }
  for (int l = 0; l < LANES; ++l)
    for (int i = 0; i < S::E; ++i) out[S::j(l, i)] = center_from_zq(tr[l][i], C);
  return 0;
}

}  // namespace

extern "C" {
int emul_shift_product(int logn, int pair, int passes, uint64_t q, const int64_t* d, const int64_t* v, int64_t* out) {
  if (logn == 1011)   // N = 2048, two-wavefront team
    return pair ? shift_product<11, true, 7>(passes, q, d, v, out) : shift_product<11, false, 7>(passes, q, d, v, out);
  switch (logn) {
    case 9: return pair ? shift_product<9, true>(passes, q, d, v, out) : shift_product<9, false>(passes, q, d, v, out);
    case 10: return pair ? shift_product<10, true>(passes, q, d, v, out) : shift_product<10, false>(passes, q, d, v, out);
    case 11: return pair ? shift_product<11, true>(passes, q, d, v, out) : shift_product<11, false>(passes, q, d, v, out);
  }
  return -1;
}
int emul_ntt_fwd(int logn, int pi, const uint32_t* in, uint32_t* out_std, uint32_t* out_mem) {
  if (logn == 1011) return ntt_fwd<11, 7>(pi, in, out_std, out_mem);   // N = 2048, two-wavefront team
  switch (logn) {
    case 9: return ntt_fwd<9>(pi, in, out_std, out_mem);
    case 10: return ntt_fwd<10>(pi, in, out_std, out_mem);
    case 11: return ntt_fwd<11>(pi, in, out_std, out_mem);
  }
  return -1;
}
int emul_ntt_inv(int logn, int pi, const uint32_t* in_std, uint32_t* out) {
  if (logn == 1011) return ntt_inv<11, 7>(pi, in_std, out);
  switch (logn) {
    case 9: return ntt_inv<9>(pi, in_std, out);
    case 10: return ntt_inv<10>(pi, in_std, out);
    case 11: return ntt_inv<11>(pi, in_std, out);
  }
  return -1;
}
int emul_polymul(int logn, int np, uint64_t q, const int64_t* a, const int64_t* b, int64_t* out) {
  switch (logn) {
    case 9: return polymul<9>(np, q, a, b, out);
    case 10: return polymul<10>(np, q, a, b, out);
    case 11: return polymul<11>(np, q, a, b, out);
  }
  return -1;
}
// CPU statement of what the device-side samplers draw for polynomial `poly` of a call (rzk_kernels.hip: one Philox block
// per coefficient pair; Floyd's subset algorithm for the challenge): the kernels must give exactly these values.
void emul_sample_uniform(uint64_t seed, uint32_t stream, uint64_t poly, uint32_t N, uint32_t bound, int64_t* out) {
  const uint32_t range = 2u * bound + 1u;
  for (uint32_t blk = 0; blk < N / 2; ++blk) {
    const Philox4 a = sampler_block(seed, stream, poly, blk);
    out[2 * blk] = (int64_t)uniform_below(a.v[0], a.v[1], range) - (int64_t)bound;
    out[2 * blk + 1] = (int64_t)uniform_below(a.v[2], a.v[3], range) - (int64_t)bound;
  }
}
void emul_sample_challenge(uint64_t seed, uint32_t stream, uint64_t poly, uint32_t N, uint32_t kappa, int64_t* out) {
  for (uint32_t i = 0; i < N; ++i) out[i] = 0;
  const uint32_t kap = kappa < N ? kappa : N;
  Philox4 r{};
  for (uint32_t t = 0; t < kap; ++t) {   // Floyd: a uniform kap-subset of [0, N)
    if ((t & 1) == 0) r = sampler_block(seed, stream, poly, t >> 1);
    const uint32_t j = N - kap + t;
    const uint32_t w0 = r.v[(t & 1) * 2], w1 = r.v[(t & 1) * 2 + 1];
    const uint32_t pick = uniform_below(w0, w1 & ~1u, j + 1);
    const uint32_t pos = out[pick] ? j : pick;
    out[pos] = (w1 & 1u) ? 1 : -1;
  }
}
void emul_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  const Philox4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
uint32_t emul_uniform_below(uint32_t hi, uint32_t lo, uint32_t range) { return uniform_below(hi, lo, range); }
uint32_t emul_prime(int pi) { return kPrimes[pi]; }
uint32_t emul_psi(int pi, uint32_t N) { return host::psi_for(pi, N); }
double emul_capacity(int np) { return host::crt_capacity(np); }
}
