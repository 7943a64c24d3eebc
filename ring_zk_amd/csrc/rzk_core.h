// rzk_core.h — lane-level arithmetic, register/LDS geometry and the wave NTT of the MI355X backend.
//
// One 64-lane wavefront transforms one residue polynomial of N = 2^LOGN coefficients
// (LOGN = 9, 10, 11; E = N/64 coefficients per lane).  The transform is a negacyclic
// Cooley-Tukey NTT (forward, natural -> bit-reversed order) / Gentleman-Sande inverse over an
// auxiliary 30-bit prime p = 1 (mod 8192), with 32-bit Montgomery multiplication (R = 2^32, one
// v_mad_u64_u32 + v_mul_lo_u32 + v_mad_u64_u32) and Harvey lazy reduction (values in [0,4p)
// forward, [0,2p) inverse).  It runs in three register phases separated by two transpositions
// through a wave-private LDS buffer:
//
//   phase 1  lane holds a[e*64 + lane]               stages 0 .. LE-1       (twiddles wave-uniform)
//   phase 2  lane holds a[hi*64 + r*2^(6-LE) + lo]   stages LE .. 2LE-1     (twiddles per lane)
//   phase 3  lane holds a[lane*E + c]                stages 2LE .. LOGN-1   (twiddles per lane)
//
// Every function here is written over an explicit (lane, register array, LDS pointer) so that the
// very same code is compiled (a) by hipcc into the gfx950 kernels and (b) by g++ into the CPU
// lane emulator under tests/emul/, which replays the 64 lanes phase by phase and lets the index
// arithmetic be checked bit-for-bit against the oracle without a GPU.  The emulator is test
// infrastructure; the product library contains only the HIP build.
//
// The reference has no counterpart for any of this: its ring multiply lives in the third-party
// crate poly-ring-xnp1 (Cargo.toml:18) and is called at src/mat.rs:110, src/mat.rs:176 and
// src/prove/linear.rs:94.  Because Q = 3515337053 = 5 (mod 8) has no 2N-th root of unity
// (SURVEY.md §0 fact 3), products are computed exactly over Z via CRT on up to three auxiliary
// primes and then reduced to the centred representative mod Q.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RZK_HD __host__ __device__ __forceinline__
#else
#define RZK_HD inline
#endif

// Optional scheduling fence for the device build (tuning knob, off by default): VALU instructions may
// not be moved across it (loads, LDS reads and scalar instructions may); placed after every
// RZK_BFLY_GROUP butterflies.
#ifndef RZK_BFLY_GROUP
#define RZK_BFLY_GROUP 0   // 0 = no fence (measured: fences do not lower the register count)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define RZK_SCHED_GROUP(count)                                                          \
  do {                                                                                  \
    if (RZK_BFLY_GROUP > 0 && ((count) % (RZK_BFLY_GROUP > 0 ? RZK_BFLY_GROUP : 1)) == 0) \
      __builtin_amdgcn_sched_barrier(0x4 | 0x20 | 0x100);                               \
  } while (0)
#else
#define RZK_SCHED_GROUP(count) do { } while (0)
#endif

namespace rzk {

constexpr int kMaxPrimes = 3;
constexpr int kKeyImages = kMaxPrimes;   // resident images of a key entry in the NTT domain: one per auxiliary prime
constexpr int kTableLog = 12;            // twiddle tables cover N <= 4096
constexpr int kTableLen = 1 << kTableLog;

// The three auxiliary primes: largest primes below 2^30 with p = 1 (mod 8192).
// p < 2^30 keeps lazy values (< 4p) inside 32 bits; p > (Q-1)/4 lets a centred input x be lifted
// to the lazy range with a single add (x + 2p in [0,4p)).
constexpr uint32_t kPrimes[kMaxPrimes] = {1073692673u, 1073668097u, 1073651713u};
constexpr uint32_t kPrimeGenerators[kMaxPrimes] = {3u, 3u, 10u};

struct PrimeConsts {   // per auxiliary prime
  uint32_t p;
  uint32_t twop;
  uint32_t npinv;      // -p^{-1} mod 2^32
  uint32_t r2;         // 2^64 mod p   (mont(x, r2) = x * R : conversion into Montgomery form)
  uint32_t ninv_r;     // N^{-1} * R   mod p : mont(x, ninv_r)  = x * N^{-1}
  uint32_t ninv_r2;    // N^{-1} * R^2 mod p : mont(x, ninv_r2) = x * N^{-1} * R
  uint32_t pad0, pad1;
};

// ---- Montgomery arithmetic, R = 2^32 -------------------------------------------------------------
// x < 2^32, w < p < 2^30.  Returns x*w*R^{-1} mod p as a lazy value in [0, 2p).
RZK_HD uint32_t mont_lazy(uint32_t x, uint32_t w, uint32_t p, uint32_t npinv) {
  uint64_t t = (uint64_t)x * w;
  uint32_t m = (uint32_t)t * npinv;
  uint64_t u = (uint64_t)m * p + t;   // low 32 bits are zero; < 2^63
  return (uint32_t)(u >> 32);
}
// unsigned conditional subtract: v in [0, 2m) -> [0, m)
RZK_HD uint32_t csub(uint32_t v, uint32_t m) {
#if !defined(RZK_CSUB_MIN)
  // subtract with borrow-out, then select: v_sub_co_u32 + v_cndmask_b32 (measured 2-4 % faster per launch on gfx950 than
  // the v_sub_u32 + v_min_u32 form below, whose v_min_u32 issues at ~4.3 cycles against ~2.8 for these two)
  uint32_t d;
  const bool borrow = __builtin_sub_overflow(v, m, &d);
  return borrow ? v : d;
#else
  uint32_t d = v - m;
  return d < v ? d : v;   // d wraps above v exactly when v < m
#endif
}

// Cooley-Tukey butterfly, Harvey lazy: X, Y in [0,4p) -> [0,4p).  w in Montgomery form.
RZK_HD void bfly_fwd(uint32_t& X, uint32_t& Y, uint32_t w, const PrimeConsts& pc) {
  uint32_t x = csub(X, pc.twop);
  uint32_t v = mont_lazy(Y, w, pc.p, pc.npinv);
  X = x + v;
  Y = x - v + pc.twop;
}
// Gentleman-Sande butterfly, lazy: X, Y in [0,2p) -> [0,2p).  w in Montgomery form.
RZK_HD void bfly_inv(uint32_t& X, uint32_t& Y, uint32_t w, const PrimeConsts& pc) {
  uint32_t s = X + Y;
  uint32_t d = X - Y + pc.twop;
  X = csub(s, pc.twop);
  Y = mont_lazy(d, w, pc.p, pc.npinv);
}

// ---- geometry ------------------------------------------------------------------------------------
// A TEAM of 2^LL threads transforms one polynomial: LL = 6 is one wavefront (every size), LL = 7 two wavefronts that
// share one LDS slab (N = 2048: 16 coefficients per thread instead of 32 — the register footprint, and with it the
// occupancy, of the N = 1024 kernels).  `lane` below is the thread's index inside its team (0 .. 2^LL - 1).
template <int LOGN, int LL = 6>
struct Geo {
  static_assert(LL == 6 || LL == 7, "teams of one or two wavefronts");
  static_assert(LOGN >= 9 && LOGN <= 11, "N = 512, 1024, 2048");
  static constexpr int LOGLANES = LL;
  static constexpr int LANES = 1 << LL;
  static constexpr int N = 1 << LOGN;
  static constexpr int LE = LOGN - LL;     // local bits per phase
  static constexpr int E = 1 << LE;        // coefficients per thread
  static constexpr int R3 = LOGN - 2 * LE; // stages left for phase 3
  static constexpr int LOSH = LL - LE;     // low thread bits of phase 2
  static_assert(LE >= 3 && R3 >= 0 && R3 <= LE && LOSH >= 0, "three register phases must cover log2 N stages");
  // LDS word address of coefficient j: one pad word per 32 keeps all three access patterns
  // (phase-1 rows, phase-2 strided, phase-3 blocked) bank-conflict free for ds_*_b32
  // (N = 1024, 2048, both team sizes) or 2-way (N = 512); see tools/lds_conflicts.py.
  static constexpr int LDS_WORDS = N + N / 32;
  RZK_HD static int lds_addr(int j) { return j + (j >> 5); }
  // phase-2 thread split (which thread bits select the LE-bit "hi" digit); chosen per shape for conflict-free LDS:
  //   one wavefront, N <= 1024: hi = lane & (E-1), lo = lane >> LE ;  N = 2048: hi = lane >> 1, lo = lane & 1
  //   two wavefronts (N = 2048): hi = (t >> 2) & 15, lo = (t & 3) | (wave << 2)
  static constexpr bool P2_HI_LOW = (LOGN <= 10);
  RZK_HD static int p2_hi(int lane) {
    if (LL == 7) return (lane >> 2) & (E - 1);
    return P2_HI_LOW ? (lane & (E - 1)) : (lane >> LOSH);
  }
  RZK_HD static int p2_lo(int lane) {
    if (LL == 7) return (lane & 3) | ((lane >> 6) << 2);
    return P2_HI_LOW ? (lane >> LE) : (lane & ((1 << LOSH) - 1));
  }
  // coefficient index held by (lane, reg) in each phase
  RZK_HD static int j_p1(int lane, int e) { return e * LANES + lane; }
  RZK_HD static int j_p2(int lane, int r) { return p2_hi(lane) * LANES + (r << LOSH) + p2_lo(lane); }
  RZK_HD static int j_p3(int lane, int c) { return lane * E + c; }
  // "RZK NTT layout": word offset, inside one N-word residue polynomial in global memory, of the
  // phase-3 register c of `lane`; groups of 4 registers form one 16-byte coalesced access.  The layout is
  // defined by the ONE-wavefront geometry of the ring degree (what the resident key and the batched transforms
  // use); a two-wavefront team addresses the same words through its own (lane, c) -> position map.
  static constexpr int LOGE64 = LOGN - 6;
  RZK_HD static int mem_p3(int lane, int c) {
    const unsigned j = (unsigned)(lane * E + c), l64 = j >> LOGE64, c64 = j & ((1u << LOGE64) - 1u);
    return (int)((c64 >> 2) * 256u + l64 * 4u + (c64 & 3u));
  }
  // index, in 16-byte units, of the group g (registers 4g .. 4g+3) of `lane` in that layout
  RZK_HD static int key4(int lane, int g) { return mem_p3(lane, 4 * g) >> 2; }
  // ... and in a buffer that only this team's threads read back (parked sums, Garner state lines): any bijection
  // will do, lane-consecutive 16-byte slots coalesce best
  RZK_HD static int own4(int lane, int g) { return g * LANES + lane; }
};

// ---- LDS transpositions -----------------------------------------------------------------------------
// Every access is written as (per-lane base) + (compile-time offset), so the compiler folds the offset
// into the DS instruction and needs ONE address register per pattern.  With j = base_j + const and the
// pad term j >> 5, the split is exact because the lane part and the constant part never carry into
// each other below bit 5 (W = LANES + LANES/32 is the padded length of one phase-1 row):
//   phase 1: j = e*LANES + lane          -> addr = [lane + (lane>>5)] + e*W
//   phase 2: j = hi*LANES + (r<<LOSH)+lo -> addr = [hi*W + lo] + (r<<LOSH) + ((r<<LOSH)>>5)     (lo < 2^LOSH)
//   phase 3: j = lane*E + c              -> addr = [lane*E + ((lane*E)>>5)] + c                  ((lane*E & 31) + c < 32)
template <int LOGN, int LL = 6>
struct LdsMap {
  using G = Geo<LOGN, LL>;
  static constexpr int W = G::LANES + G::LANES / 32;
  RZK_HD static int base_p1(int lane) { return lane + (lane >> 5); }
  static constexpr int off_p1(int e) { return e * W; }
  RZK_HD static int base_p2(int lane) { return G::p2_hi(lane) * W + G::p2_lo(lane); }
  static constexpr int off_p2(int r) { return (r << G::LOSH) + ((r << G::LOSH) >> 5); }
  RZK_HD static int base_p3(int lane) { return lane * G::E + ((lane * G::E) >> 5); }
  static constexpr int off_p3(int c) { return c; }
};
template <int LOGN, int LL = 6>
RZK_HD void lds_put_p1(const uint32_t* x, int lane, uint32_t* lds) {
  uint32_t* p = lds + LdsMap<LOGN, LL>::base_p1(lane);
#pragma unroll
  for (int e = 0; e < Geo<LOGN, LL>::E; ++e) p[LdsMap<LOGN, LL>::off_p1(e)] = x[e];
}
template <int LOGN, int LL = 6>
RZK_HD void lds_get_p1(uint32_t* x, int lane, const uint32_t* lds) {
  const uint32_t* p = lds + LdsMap<LOGN, LL>::base_p1(lane);
#pragma unroll
  for (int e = 0; e < Geo<LOGN, LL>::E; ++e) x[e] = p[LdsMap<LOGN, LL>::off_p1(e)];
}
template <int LOGN, int LL = 6>
RZK_HD void lds_put_p2(const uint32_t* x, int lane, uint32_t* lds) {
  uint32_t* p = lds + LdsMap<LOGN, LL>::base_p2(lane);
#pragma unroll
  for (int r = 0; r < Geo<LOGN, LL>::E; ++r) p[LdsMap<LOGN, LL>::off_p2(r)] = x[r];
}
template <int LOGN, int LL = 6>
RZK_HD void lds_get_p2(uint32_t* x, int lane, const uint32_t* lds) {
  const uint32_t* p = lds + LdsMap<LOGN, LL>::base_p2(lane);
#pragma unroll
  for (int r = 0; r < Geo<LOGN, LL>::E; ++r) x[r] = p[LdsMap<LOGN, LL>::off_p2(r)];
}
template <int LOGN, int LL = 6>
RZK_HD void lds_put_p3(const uint32_t* x, int lane, uint32_t* lds) {
  uint32_t* p = lds + LdsMap<LOGN, LL>::base_p3(lane);
#pragma unroll
  for (int c = 0; c < Geo<LOGN, LL>::E; ++c) p[LdsMap<LOGN, LL>::off_p3(c)] = x[c];
}
template <int LOGN, int LL = 6>
RZK_HD void lds_get_p3(uint32_t* x, int lane, const uint32_t* lds) {
  const uint32_t* p = lds + LdsMap<LOGN, LL>::base_p3(lane);
#pragma unroll
  for (int c = 0; c < Geo<LOGN, LL>::E; ++c) x[c] = p[LdsMap<LOGN, LL>::off_p3(c)];
}

// ---- forward transform, register phases ----------------------------------------------------------------
// tw: Montgomery-form table, tw[m + i] = psi^{bitrev(m+i)} * R mod p (first N entries used).
template <int LOGN, int LL = 6>
RZK_HD void fwd_phase1(uint32_t* x, const uint32_t* tw, const PrimeConsts& pc) {
  using G = Geo<LOGN, LL>;
  int nb = 0;
  (void)nb;
#pragma unroll
  for (int s = 0; s < G::LE; ++s) {
    const int half = G::E >> (s + 1);
#pragma unroll
    for (int e = 0; e < G::E; ++e) {
      if (e & half) continue;
      const uint32_t w = tw[(1 << s) + (e >> (G::LE - s))];   // wave-uniform
      bfly_fwd(x[e], x[e + half], w, pc);
      ++nb;
      RZK_SCHED_GROUP(nb);
    }
  }
}
template <int LOGN, int LL = 6>
RZK_HD void fwd_phase2(uint32_t* x, int lane, const uint32_t* tw, const PrimeConsts& pc) {
  using G = Geo<LOGN, LL>;
  int nb = 0;
  (void)nb;
  const int hi = G::p2_hi(lane);
#pragma unroll
  for (int sp = 0; sp < G::LE; ++sp) {
    const int half = G::E >> (sp + 1);
    const uint32_t* twl = tw + (1 << (G::LE + sp)) + (hi << sp);   // 2^sp contiguous entries
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
      if (r & half) continue;
      bfly_fwd(x[r], x[r + half], twl[r >> (G::LE - sp)], pc);
      ++nb;
      RZK_SCHED_GROUP(nb);
    }
  }
}
template <int LOGN, int LL = 6>
RZK_HD void fwd_phase3(uint32_t* x, int lane, const uint32_t* tw, const PrimeConsts& pc) {
  using G = Geo<LOGN, LL>;
  int nb = 0;
  (void)nb;
#pragma unroll
  for (int sq = 0; sq < G::R3; ++sq) {
    const int sh = G::R3 - sq;            // idx = 2^s + (j >> sh)
    const int half = 1 << (sh - 1);
    const uint32_t* twl = tw + (1 << (2 * G::LE + sq)) + lane * (G::E >> sh);
#pragma unroll
    for (int c = 0; c < G::E; ++c) {
      if (c & half) continue;
      bfly_fwd(x[c], x[c + half], twl[c >> sh], pc);
      ++nb;
      RZK_SCHED_GROUP(nb);
    }
  }
}

// ---- inverse transform, register phases (mirror image; tw = psi^{-bitrev} * R) -------------------------------
template <int LOGN, int LL = 6>
RZK_HD void inv_phase3(uint32_t* x, int lane, const uint32_t* tw, const PrimeConsts& pc) {
  using G = Geo<LOGN, LL>;
  int nb = 0;
  (void)nb;
#pragma unroll
  for (int sq = G::R3 - 1; sq >= 0; --sq) {
    const int sh = G::R3 - sq;
    const int half = 1 << (sh - 1);
    const uint32_t* twl = tw + (1 << (2 * G::LE + sq)) + lane * (G::E >> sh);
#pragma unroll
    for (int c = 0; c < G::E; ++c) {
      if (c & half) continue;
      bfly_inv(x[c], x[c + half], twl[c >> sh], pc);
      ++nb;
      RZK_SCHED_GROUP(nb);
    }
  }
}
template <int LOGN, int LL = 6>
RZK_HD void inv_phase2(uint32_t* x, int lane, const uint32_t* tw, const PrimeConsts& pc) {
  using G = Geo<LOGN, LL>;
  int nb = 0;
  (void)nb;
  const int hi = G::p2_hi(lane);
#pragma unroll
  for (int sp = G::LE - 1; sp >= 0; --sp) {
    const int half = G::E >> (sp + 1);
    const uint32_t* twl = tw + (1 << (G::LE + sp)) + (hi << sp);
#pragma unroll
    for (int r = 0; r < G::E; ++r) {
      if (r & half) continue;
      bfly_inv(x[r], x[r + half], twl[r >> (G::LE - sp)], pc);
      ++nb;
      RZK_SCHED_GROUP(nb);
    }
  }
}
template <int LOGN, int LL = 6>
RZK_HD void inv_phase1(uint32_t* x, const uint32_t* tw, const PrimeConsts& pc) {
  using G = Geo<LOGN, LL>;
  int nb = 0;
  (void)nb;
#pragma unroll
  for (int s = G::LE - 1; s >= 0; --s) {
    const int half = G::E >> (s + 1);
#pragma unroll
    for (int e = 0; e < G::E; ++e) {
      if (e & half) continue;
      const uint32_t w = tw[(1 << s) + (e >> (G::LE - s))];
      bfly_inv(x[e], x[e + half], w, pc);
      ++nb;
      RZK_SCHED_GROUP(nb);
    }
  }
}

// ---- lifting centred coefficients into a prime field, pointwise products -----------------------------------
// centred coefficient (|v| <= (Q-1)/2 < 2p) -> lazy residue in [0,4p)
RZK_HD uint32_t lift(int32_t v, const PrimeConsts& pc) { return (uint32_t)v + pc.twop; }

// acc in [0,2p), x < 2^32, k in Montgomery form (< p): acc +/- x*k  -> [0,2p)
RZK_HD uint32_t mac_add(uint32_t acc, uint32_t x, uint32_t k, const PrimeConsts& pc) {
  return csub(acc + mont_lazy(x, k, pc.p, pc.npinv), pc.twop);
}
RZK_HD uint32_t mac_sub(uint32_t acc, uint32_t x, uint32_t k, const PrimeConsts& pc) {
  return csub(acc + pc.twop - mont_lazy(x, k, pc.p, pc.npinv), pc.twop);
}

// ---- CRT reconstruction and reduction to the centred representative mod q ------------------------------------
struct CrtConsts {
  uint32_t q, qinv;        // ring modulus (odd, < 2^32) and q^{-1} mod 2^32
  uint32_t qhalf;          // (q-1)/2
  uint32_t inv01_r;        // p0^{-1} mod p1, Montgomery form mod p1
  uint32_t p0_mod_p2_r;    // p0 mod p2, Montgomery form mod p2
  uint32_t inv012_r;       // (p0*p1)^{-1} mod p2, Montgomery form mod p2
  uint32_t c1;             // p0      * 2^32 mod q
  uint32_t c2;             // p0 * p1 * 2^32 mod q
  uint32_t pmodq[4];       // [np] = (p0*..*p_{np-1}) mod q
  uint32_t half1;          // (p0+1)/2
  uint32_t half3_d2;       // (P3+1)/2 = half3_lo + half3_d2 * (p0*p1)
  uint64_t half2;          // (p0*p1+1)/2
  uint64_t half3_lo;
  // offset form: X' = X + H_np with H_np = (P_np - 1)/2 is non-negative, so no sign test is needed
  uint32_t hmod[4][4];     // hmod[np][i] = H_np mod p_i
  uint32_t hmodq[4];       // H_np mod q
  uint32_t r2q;            // 2^64 mod q   (montq_u(a, r2q) = a * 2^32 mod q)
  uint32_t r48q;           // 2^48 mod q   (montq_u(a, r48q) = a * 2^16 mod q)
};

// a*c*2^{-32} mod q as a signed value in (-q, q);  a, c < 2^32
RZK_HD int64_t montq(uint32_t a, uint32_t c, const CrtConsts& C) {
  uint64_t t = (uint64_t)a * c;
  uint32_t m = (uint32_t)t * C.qinv;
  uint32_t h = (uint32_t)(((uint64_t)m * C.q) >> 32);
  return (int64_t)(uint32_t)(t >> 32) - (int64_t)h;
}
// s -> centred representative, for |s| < (2*ROUNDS - 1) * q / 2 ... conservatively |s| < ROUNDS*q
template <int ROUNDS>
RZK_HD int64_t center_rounds(int64_t s, const CrtConsts& C) {
  const int64_t q = C.q, h = C.qhalf;
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    s = s > h ? s - q : s;
    s = s < -h ? s + q : s;
  }
  return s;
}
// Residues r[i] in [0,2p_i) (lazy, as left by the inverse transform) of an integer X with
// |X| < P_np/2  ->  centred representative of X mod q, as a signed value in [-(q-1)/2, (q-1)/2].
RZK_HD int64_t crt_center(uint32_t r0, uint32_t r1, uint32_t r2, int np, const PrimeConsts* pc,
                          const CrtConsts& C) {
  const uint32_t d0 = csub(r0, pc[0].p);
  int64_t s = d0;
  bool neg;
  if (np == 1) {
    neg = d0 >= C.half1;
  } else {
    // d1 = (r1 - d0) * p0^{-1} mod p1
    const uint32_t t1 = r1 + pc[1].twop - d0;   // r1 < 2p1, d0 < p0 < 2p1  ->  in (0, 4p1)
    const uint32_t d1 = csub(mont_lazy(t1, C.inv01_r, pc[1].p, pc[1].npinv), pc[1].p);
    const uint64_t lo = (uint64_t)d1 * pc[0].p + d0;   // X mod p0p1
    s += montq(d1, C.c1, C);
    if (np == 2) {
      neg = lo >= C.half2;
    } else {
      // d2 = (r2 - (d0 + d1*p0)) * (p0p1)^{-1} mod p2
      const uint32_t m1 = mont_lazy(d1, C.p0_mod_p2_r, pc[2].p, pc[2].npinv);   // d1*p0 mod p2, [0,2p2)
      const uint32_t low2 = csub(csub(d0, pc[2].p) + m1, pc[2].twop);           // (d0 + d1 p0) mod p2, [0,2p2)
      const uint32_t t2 = r2 + pc[2].twop - low2;                               // (0, 4p2)
      const uint32_t d2 = csub(mont_lazy(t2, C.inv012_r, pc[2].p, pc[2].npinv), pc[2].p);
      s += montq(d2, C.c2, C);
      neg = (d2 > C.half3_d2) || (d2 == C.half3_d2 && lo >= C.half3_lo);
    }
  }
  if (neg) s -= C.pmodq[np];
  return center_rounds<3>(s, C);
}

// ---- incremental (Garner) reconstruction in offset form -----------------------------------------------------
// The kernels process the primes one after the other and fold each residue polynomial into a running
// state as soon as its inverse transform is done, so only two 32-bit words per coefficient stay live:
//   after prime 0:  stA = d0                     (first mixed-radix digit of X' = X + H_np)
//   after prime 1:  stA = X' mod q (np == 2)  or (d0 + d1 p0) mod q,  stB = (d0 + d1 p0) mod p2 (np == 3)
//   after prime 2:  stA = X' mod q
// r is the lazy residue in [0,2p_i) left by the inverse transform.  Requires p0 < q.
RZK_HD uint32_t crt_fold0(uint32_t r, int np, const PrimeConsts* pc, const CrtConsts& C) {
  return csub(csub(r + C.hmod[np][0], pc[0].twop), pc[0].p);
}
// ---- one- and two-prime results in SIGN-TEST form (no offset) -----------------------------------------------------
// Rows that need at most two primes (key x ternary, key x Gaussian: the commitments, t, the verifier relation) skip
// the offset: d0 = r0 mod p0 is one conditional subtract instead of add + two, the second digit needs no H terms, and
// the sign of X comes from one 64-bit compare of X' = d0 + d1 p0 against (P+1)/2.  About a third fewer vector
// instructions per coefficient than crt_fold0 / crt_digit1 / crt_finish_zq on this path (three-prime rows keep the
// offset form: its third step would need the 64-bit low part again).  Requires p0 < q.
RZK_HD uint32_t crt1_zq(uint32_t r0, const PrimeConsts* pc, const CrtConsts& C) {   // np = 1: X mod q in [0,q)
  const uint32_t d0 = csub(r0, pc[0].p);
  return d0 >= C.half1 ? d0 + (C.q - pc[0].p) : d0;   // X = d0 - p0 < 0  ->  X + q
}
RZK_HD uint32_t crt2_digit0(uint32_t r0, const PrimeConsts* pc) { return csub(r0, pc[0].p); }
RZK_HD uint32_t crt2_zq(uint32_t r1, uint32_t d0, const PrimeConsts* pc, const CrtConsts& C);   // (defined below addq / montq_u)
// ---- 32-bit arithmetic mod q (q may exceed 2^31, so sums can wrap 32 bits; handled by comparing first)
RZK_HD uint32_t addq(uint32_t a, uint32_t b, uint32_t q) {   // a, b in [0,q)
  const uint32_t nb = q - b;
  const uint32_t d = a - nb;
  return a < nb ? d + q : d;
}
RZK_HD uint32_t subq(uint32_t a, uint32_t b, uint32_t q) {   // a, b in [0,q)
  const uint32_t d = a - b;
  return a < b ? d + q : d;
}
// a*c*2^{-32} mod q in [0,q);  a, c < 2^32
RZK_HD uint32_t montq_u(uint32_t a, uint32_t c, const CrtConsts& C) {
  const uint64_t t = (uint64_t)a * c;
  const uint32_t m = (uint32_t)t * C.qinv;
  const uint32_t h = (uint32_t)(((uint64_t)m * C.q) >> 32);
  const uint32_t hi = (uint32_t)(t >> 32);
  const uint32_t d = hi - h;
  return hi < h ? d + C.q : d;
}
// centred coefficient (|a| <= (q-1)/2) -> [0,q)
RZK_HD uint32_t zq_from_centered(int32_t a, uint32_t q) { return (uint32_t)a + (a < 0 ? q : 0u); }
// [0,q) -> centred representative as int64
RZK_HD int64_t center_from_zq(uint32_t u, const CrtConsts& C) {
  const uint32_t lo = u > C.qhalf ? u - C.q : u;   // |.| <= (q-1)/2 < 2^31: the 32-bit pattern is the int32 value
  return (int64_t)(int32_t)lo;
}
RZK_HD uint32_t crt2_zq(uint32_t r1, uint32_t d0, const PrimeConsts* pc, const CrtConsts& C) {   // np = 2: X mod q in [0,q)
  const uint32_t t1 = r1 + pc[1].twop - d0;                                  // r1 < 2 p1, d0 < p0 < 2 p1  ->  (0, 4 p1)
  const uint32_t d1 = csub(mont_lazy(t1, C.inv01_r, pc[1].p, pc[1].npinv), pc[1].p);
  const uint32_t va = addq(d0, montq_u(d1, C.c1, C), C.q);                   // (d0 + d1 p0) mod q
  const uint64_t x01 = (uint64_t)d1 * pc[0].p + d0;                          // X mod p0 p1, exact
  return subq(va, x01 >= C.half2 ? C.pmodq[2] : 0u, C.q);                    // negative X: minus P mod q
}
// second digit d1 = (X' mod p1 - d0) * p0^{-1} mod p1 from the prime-1 residue r and d0
RZK_HD uint32_t crt_digit1(uint32_t r, uint32_t d0, int np, const PrimeConsts* pc, const CrtConsts& C) {
  const uint32_t a1 = csub(csub(r + C.hmod[np][1], pc[1].twop), pc[1].p);
  const uint32_t t1 = a1 + pc[1].p - csub(d0, pc[1].p);                      // (0, 2p1)
  return csub(mont_lazy(t1, C.inv01_r, pc[1].p, pc[1].npinv), pc[1].p);
}
// (d0 + d1 p0) mod q in [0,q)
RZK_HD uint32_t crt_value01_modq(uint32_t d0, uint32_t d1, const CrtConsts& C) {
  return addq(d0, montq_u(d1, C.c1, C), C.q);   // d0 < p0 < q
}
// (d0 + d1 p0) mod p2, lazy [0,2p2) — only needed when a third prime follows
RZK_HD uint32_t crt_value01_modp2(uint32_t d0, uint32_t d1, const PrimeConsts* pc, const CrtConsts& C) {
  const uint32_t m1 = mont_lazy(d1, C.p0_mod_p2_r, pc[2].p, pc[2].npinv);    // d1*p0 mod p2, [0,2p2)
  return csub(csub(d0, pc[2].p) + m1, pc[2].twop);
}
RZK_HD void crt_fold1(uint32_t r, int np, const PrimeConsts* pc, const CrtConsts& C, uint32_t& stA,
                      uint32_t& stB) {
  const uint32_t d0 = stA;
  const uint32_t d1 = crt_digit1(r, d0, np, pc, C);
  stA = crt_value01_modq(d0, d1, C);
  if (np == 3) stB = crt_value01_modp2(d0, d1, pc, C);
}
RZK_HD void crt_fold2(uint32_t r, const PrimeConsts* pc, const CrtConsts& C, uint32_t& stA, uint32_t stB) {
  const uint32_t a2 = csub(csub(r + C.hmod[3][2], pc[2].twop), pc[2].p);
  const uint32_t t2 = a2 + pc[2].twop - stB;                                 // (0, 3p2)
  const uint32_t d2 = csub(mont_lazy(t2, C.inv012_r, pc[2].p, pc[2].npinv), pc[2].p);
  stA = addq(stA, montq_u(d2, C.c2, C), C.q);
}
// X' mod q -> X mod q in [0,q)   (X = X' - H_np)
RZK_HD uint32_t crt_finish_zq(uint32_t stA, int np, const CrtConsts& C) { return subq(stA, C.hmodq[np], C.q); }
// ... and its centred representative
RZK_HD int64_t crt_finish(uint32_t stA, int np, const CrtConsts& C) {
  return center_from_zq(crt_finish_zq(stA, np, C), C);
}


// ---- shift-add products: sparse multiplier times arbitrary polynomial ------------------------------------------
// The challenges of the protocols have kappa coefficients +-1 and zeros elsewhere
// (reference src/challenge_space.rs:12-33), so d (*) v = sum_s d_s X^s v is kappa signed negacyclic
// rotations of v.  One wavefront keeps the "extended" image  ext[t] = -v[t] (t < N), v[t-N] (t >= N)
// in LDS (2N words), where (X^s v)[j] = ext[j - s + N] needs neither a wrap test nor a sign flip; the
// lanes of a wave read consecutive words (conflict-free) at compile-time offsets from one base address.
// Any multiplier is handled exactly (the cost grows with its number of non-zero coefficients).
// Two register layouts: PAIR (coefficients g*128 + 2*lane + {0,1}: 16-byte global accesses, the stand-alone
// kernel) and the phase-1 layout of the transforms (e*64 + lane: what the row kernel's epilogue uses).
template <int LOGN, bool PAIR = true, int LL = 6>
struct ShiftGeo {
  static constexpr int N = 1 << LOGN;
  static constexpr int LANES = 1 << LL;     // threads of the team (one or two wavefronts)
  static constexpr int E = N / LANES;       // outputs per thread
  static constexpr int G = E / 2;           // pairs per thread (PAIR layout)
  static constexpr int WORDS = 2 * N;       // LDS words per team
  RZK_HD static int lane_base(int lane) { return PAIR ? 2 * lane : lane; }
  static constexpr int off(int i) { return PAIR ? (i >> 1) * 2 * LANES + (i & 1) : i * LANES; }
  RZK_HD static int j(int lane, int i) { return lane_base(lane) + off(i); }   // coefficient of register i
};
// Teams of two wavefronts cannot walk the multiplier with ballot / readlane (each wavefront sees half of it): its
// non-zero coefficients are compacted into a list {position, value} behind the image, kShiftListCap entries per
// round (one round for a challenge; any multiplier is still handled exactly, in ceil(nonzeros / cap) rounds).
constexpr int kShiftListCap = 256;
constexpr int kShiftListWords = 2 * kShiftListCap + 4;   // entries, then the two wavefronts' counts

// what of v goes into the image: the value itself, or one 16-bit half (two passes keep 64-bit sums exact
// for multipliers of any size)
enum : int { SHIFT_WHOLE = 0, SHIFT_LOW16 = 1, SHIFT_HIGH16 = 2 };
RZK_HD int32_t shift_part(int32_t v, int part) {
  return part == SHIFT_WHOLE ? v : (part == SHIFT_LOW16 ? (int32_t)((uint32_t)v & 0xffffu) : (v >> 16));
}

template <int LOGN, bool PAIR, int LL = 6>
RZK_HD void shift_fill(const int32_t* v, int lane, int32_t* ext, int part) {
  using S = ShiftGeo<LOGN, PAIR, LL>;
  int32_t* base = ext + S::lane_base(lane);
#pragma unroll
  for (int i = 0; i < S::E; ++i) {
    const int32_t a = shift_part(v[i], part);
    base[S::N + S::off(i)] = a;
    base[S::off(i)] = -a;
  }
}

// acc[i - I0] += coef * (X^s v)[j(lane, i)] for the registers I0 .. I0+IN-1 of one lane (T = int64_t: one
// v_mad_i64_i32 per output).
template <int LOGN, bool PAIR, typename T, int I0, int IN, int LL = 6>
RZK_HD void shift_accum(T* acc, int lane, int s, int32_t coef, const int32_t* ext) {
  using S = ShiftGeo<LOGN, PAIR, LL>;
  const int32_t* base = ext + (S::lane_base(lane) + S::N - s);
#pragma unroll
  for (int i = 0; i < IN; ++i) acc[i] += (T)coef * (T)base[S::off(I0 + i)];
}
// exact integer -> [0,q), |a| < 2^62
RZK_HD uint32_t zq_from_i64(int64_t a, const CrtConsts& C) {
  const int32_t hi = (int32_t)(a >> 32);          // |hi| < 2^30 < q
  uint32_t lo = (uint32_t)a;                      // < 2^32 < 4q
  lo = lo >= C.q ? lo - C.q : lo;
  lo = lo >= C.q ? lo - C.q : lo;
  lo = lo >= C.q ? lo - C.q : lo;
  return addq(montq_u(zq_from_centered(hi, C.q), C.r2q, C), lo, C.q);
}

}  // namespace rzk
