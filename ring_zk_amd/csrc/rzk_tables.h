// rzk_tables.h — host-side generation of the constants the kernels consume: per-prime Montgomery
// constants, bit-reversed twiddle tables, CRT / mod-q constants and the static operand-size bounds
// that decide how many auxiliary primes a product needs.  Plain C++ (no HIP), header-only, used by
// the library (rzk_api.hip) and by the CPU lane emulator under tests/emul/.
#pragma once
#include <cstdint>
#include <vector>

#include "rzk_core.h"

namespace rzk {
namespace host {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((u128)a * b % m); }
inline uint64_t powmod(uint64_t b, uint64_t e, uint64_t m) {
  uint64_t r = 1 % m;
  b %= m;
  while (e) {
    if (e & 1) r = mulmod(r, b, m);
    b = mulmod(b, b, m);
    e >>= 1;
  }
  return r;
}
inline uint64_t invmod_prime(uint64_t a, uint64_t p) { return powmod(a, p - 2, p); }
inline uint32_t inv_u32(uint32_t odd) {   // odd^{-1} mod 2^32 (Newton)
  uint32_t x = odd;                       // correct to 3 bits
  for (int i = 0; i < 5; ++i) x *= 2u - odd * x;
  return x;
}
inline uint32_t bitrev(uint32_t v, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

// Montgomery form of v mod p
inline uint32_t to_mont(uint64_t v, uint32_t p) { return (uint32_t)(((u128)(v % p) << 32) % p); }

inline PrimeConsts make_prime_consts(int pi, uint32_t N) {
  PrimeConsts c{};
  const uint32_t p = kPrimes[pi];
  c.p = p;
  c.twop = 2 * p;
  c.npinv = 0u - inv_u32(p);
  c.r2 = (uint32_t)((((u128)1) << 64) % p);
  const uint64_t ninv = invmod_prime(N % p, p);
  c.ninv_r = to_mont(ninv, p);
  c.ninv_r2 = (uint32_t)((u128)c.ninv_r * ((((u128)1) << 32) % p) % p);
  return c;
}

// primitive 2*kTableLen-th root of unity of prime pi
inline uint32_t root_of_unity(int pi) {
  const uint32_t p = kPrimes[pi];
  return (uint32_t)powmod(kPrimeGenerators[pi], (p - 1) / (2u * kTableLen), p);
}
// psi for ring degree N (primitive 2N-th root), consistent with the nested table
inline uint32_t psi_for(int pi, uint32_t N) {
  return (uint32_t)powmod(root_of_unity(pi), kTableLen / N, kPrimes[pi]);
}

// tw[j] = psi_max^{bitrev_12(j)} * R mod p, j < 4096.  The first N entries of this table are exactly
// the table for ring degree N (psi_N = psi_max^(4096/N)), so one table per prime serves every N.
inline void make_twiddles(int pi, std::vector<uint32_t>& fwd, std::vector<uint32_t>& inv) {
  const uint32_t p = kPrimes[pi];
  const uint64_t psi = root_of_unity(pi);
  const uint64_t psi_inv = invmod_prime(psi, p);
  fwd.resize(kTableLen);
  inv.resize(kTableLen);
  std::vector<uint64_t> pw(kTableLen), pwi(kTableLen);
  pw[0] = pwi[0] = 1;
  for (int i = 1; i < kTableLen; ++i) {
    pw[i] = mulmod(pw[i - 1], psi, p);
    pwi[i] = mulmod(pwi[i - 1], psi_inv, p);
  }
  for (int j = 0; j < kTableLen; ++j) {
    const uint32_t e = bitrev((uint32_t)j, kTableLog);
    fwd[j] = to_mont(pw[e], p);
    inv[j] = to_mont(pwi[e], p);
  }
}

// returns false when q is unsupported (even, or too large for the single-add lift)
inline bool make_crt_consts(uint64_t q, CrtConsts& C) {
  if ((q & 1) == 0 || q < 3 || q >= (1ull << 32)) return false;
  if ((q - 1) / 2 >= 2ull * kPrimes[kMaxPrimes - 1]) return false;
  if (q <= kPrimes[0]) return false;   // the reconstruction keeps digits below p0 as residues mod q
  const uint64_t p0 = kPrimes[0], p1 = kPrimes[1], p2 = kPrimes[2];
  C.q = (uint32_t)q;
  C.qinv = inv_u32((uint32_t)q);
  C.qhalf = (uint32_t)((q - 1) / 2);
  C.inv01_r = to_mont(invmod_prime(p0 % p1, p1), (uint32_t)p1);
  C.p0_mod_p2_r = to_mont(p0 % p2, (uint32_t)p2);
  C.inv012_r = to_mont(invmod_prime(mulmod(p0 % p2, p1 % p2, p2), p2), (uint32_t)p2);
  C.c1 = (uint32_t)(((u128)(p0 % q) << 32) % q);
  C.c2 = (uint32_t)(((u128)mulmod(p0 % q, p1 % q, q) << 32) % q);
  C.pmodq[0] = 1 % q;
  C.pmodq[1] = (uint32_t)(p0 % q);
  C.pmodq[2] = (uint32_t)mulmod(p0 % q, p1 % q, q);
  C.pmodq[3] = (uint32_t)mulmod(C.pmodq[2], p2 % q, q);
  C.half1 = (uint32_t)((p0 + 1) / 2);
  C.half2 = (p0 * p1 + 1) / 2;
  const u128 P3 = (u128)p0 * p1 * p2;
  const u128 h3 = (P3 + 1) / 2;
  C.half3_d2 = (uint32_t)(h3 / (p0 * p1));
  C.half3_lo = (uint64_t)(h3 % (p0 * p1));
  for (int np = 0; np <= kMaxPrimes; ++np) {
    u128 P = 1;
    for (int i = 0; i < np; ++i) P *= kPrimes[i];
    const u128 H = (P - 1) / 2;
    for (int i = 0; i < 4; ++i) C.hmod[np][i] = i < kMaxPrimes ? (uint32_t)(H % kPrimes[i]) : 0;
    C.hmodq[np] = (uint32_t)(H % q);
  }
  C.r2q = (uint32_t)((((u128)1) << 64) % q);
  C.r48q = (uint32_t)((((u128)1) << 48) % q);
  return true;
}

// Largest magnitude an exact integer result may have for reconstruction from np primes:
// (P_np - 1)/2, returned as a double rounded DOWN with a safety margin of 2^-40 relative.
inline double crt_capacity(int np) {
  long double P = 1;
  for (int i = 0; i < np; ++i) P *= (long double)kPrimes[i];
  return (double)((P - 1) / 2 * (1.0L - 1e-12L));
}

}  // namespace host
}  // namespace rzk
