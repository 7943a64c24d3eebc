// rzk_rng.h — counter-based random numbers for the device-side samplers (Philox4x32-10, Salmon et al.,
// "Parallel random numbers: as easy as 1, 2, 3", SC'11).  Host + device; the known-answer vectors of the
// Random123 distribution are checked on the CPU by tests/test_emul_core.py.
//
// The samplers replace the reference's host RNG calls (src/polynomial.rs:14-44, src/challenge_space.rs:12-33,
// driven by rand's thread RNG) by a stateless generator: output = f(seed, stream, polynomial, counter), so a
// run is reproducible from its seed and every polynomial can be drawn independently by any wavefront.
// Parity with the reference is statistical (same distributions), not bit-for-bit.
#pragma once
#include <stdint.h>

#include "rzk_core.h"

namespace rzk {

struct Philox4 {
  uint32_t v[4];
};

RZK_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

// Philox4x32-10: counter c[4], key k[2]
RZK_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// Counter layout of the samplers: (block within the polynomial, polynomial index lo, hi, stream id); key = seed.
RZK_HD Philox4 sampler_block(uint64_t seed, uint32_t stream, uint64_t poly, uint32_t block) {
  return philox4x32_10(block, (uint32_t)poly, (uint32_t)(poly >> 32), stream, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// uniform integer in [0, range) from 64 random bits (multiply-shift; bias <= range / 2^64)
RZK_HD uint32_t uniform_below(uint32_t hi, uint32_t lo, uint32_t range) {
  const uint64_t t = (uint64_t)lo * range;
  const uint64_t u = (uint64_t)hi * range + (t >> 32);   // top 64 bits of the 96-bit product (hi:lo) * range
  return (uint32_t)(u >> 32);
}

}  // namespace rzk
