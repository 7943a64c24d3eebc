// rzk_kernels.hip — gfx950 (MI355X) kernels of the ring-zk polynomial-ring backend.
//
// Execution model: one 64-lane wavefront owns one polynomial-sized unit of work (one output row of
// one proof, or one transform); workgroups are 4 independent wavefronts (no workgroup barriers),
// each with a private LDS slab of N + N/32 words for the two transpositions of the wave NTT
// (rzk_core.h).  Global accesses are coalesced: coefficient slabs are read/written with lane-
// consecutive 8-byte accesses (512 B per wave instruction), NTT-domain data (resident key, transform
// output) with 16-byte accesses (1 KiB per wave instruction).  No MFMA: the work is 32-bit integer
// modular arithmetic (v_mad_u64_u32 / v_mul_lo_u32), bounded by HBM traffic and integer VALU rate.
#include <hip/hip_runtime.h>

#include "rzk_core.h"
#include "rzk_dev.h"
#include "rzk_rng.h"

namespace rzk {

// Order LDS traffic of the lanes of one wavefront (write phase -> read phase).  A wavefront issues
// its LDS instructions in program order, so no s_barrier is needed; the fences only stop the
// compiler from moving LDS accesses across the phase boundary.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- streaming accesses -------------------------------------------------------------------------------------------
// Coefficient slabs are read once or twice and written once per launch: RZK_NT_LD / RZK_NT_ST select the non-temporal
// cache policy for them (so that they do not push the resident key, the twiddles and the teams' scratch lines out of
// L2).  Tuning knobs; see DESIGN.md §6 for what was measured.
#ifndef RZK_NT_LD
#define RZK_NT_LD 0
#endif
#ifndef RZK_NT_ST
#define RZK_NT_ST 1   // measured (Open N=1024, A/B of prebuilt libraries): stores nt +1.5 % (response 78.3 -> 76.0 us); loads nt -1.5 %
#endif
template <class Tp>
__device__ __forceinline__ Tp ld_stream(const Tp* p) {
#if RZK_NT_LD
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ longlong2 ld_stream(const longlong2* p) {
#if RZK_NT_LD
  typedef long long v2ll __attribute__((ext_vector_type(2)));
  const v2ll t = __builtin_nontemporal_load(reinterpret_cast<const v2ll*>(p));
  longlong2 r;
  r.x = t.x, r.y = t.y;
  return r;
#else
  return *p;
#endif
}
template <class Tp>
__device__ __forceinline__ void st_stream(Tp* p, Tp v) {
#if RZK_NT_ST
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st_stream(int4* p, int4 v) {
#if RZK_NT_ST
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i t;
  t.x = v.x, t.y = v.y, t.z = v.z, t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<v4i*>(p));
#else
  *p = v;
#endif
}

// ---- wave reductions ------------------------------------------------------------------------------------
// Butterfly inside the 16-lane rows with DPP operand modifiers (xor 1, xor 2, half-row mirror, row mirror), then the
// two row broadcasts of GFX9 (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): six VALU instructions
// with the lane exchange folded into the arithmetic, the total in lane 63, handed out as a wave-uniform scalar by
// v_readlane.  No LDS traffic (the ds_bpermute form of __shfl_xor costs an LDS round trip per step, which a
// wave that runs alone on its SIMD cannot hide).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {   // lanes without a source read 0 (the identity of +, max)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, dpp_u32<CTRL, ROW_MASK>(__builtin_bit_cast(uint32_t, v)));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppBcast15 = 0x142,
              kDppBcast31 = 0x143;
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {   // caller guarantees the total fits 32 bits
  v += dpp_u32<kDppXor1>(v);
  v += dpp_u32<kDppXor2>(v);
  v += dpp_u32<kDppHalfMirror>(v);
  v += dpp_u32<kDppMirror>(v);
  v += dpp_u32<kDppBcast15, 0xa>(v);
  v += dpp_u32<kDppBcast31, 0xc>(v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_sum_f32(float v) {   // v >= 0 in every lane
  v += dpp_f32<kDppXor1>(v);
  v += dpp_f32<kDppXor2>(v);
  v += dpp_f32<kDppHalfMirror>(v);
  v += dpp_f32<kDppMirror>(v);
  v += dpp_f32<kDppBcast15, 0xa>(v);
  v += dpp_f32<kDppBcast31, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  uint32_t o;
  o = dpp_u32<kDppXor1>(v), v = o > v ? o : v;
  o = dpp_u32<kDppXor2>(v), v = o > v ? o : v;
  o = dpp_u32<kDppHalfMirror>(v), v = o > v ? o : v;
  o = dpp_u32<kDppMirror>(v), v = o > v ? o : v;
  o = dpp_u32<kDppBcast15, 0xa>(v), v = o > v ? o : v;
  o = dpp_u32<kDppBcast31, 0xc>(v), v = o > v ? o : v;
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// exact 64-bit total of per-lane values below 2^56: three 24-bit digits, each summed in 32 bits (64 * 2^24 = 2^30)
__device__ __forceinline__ uint64_t wave_sum_u56(uint64_t v) {
  const uint32_t d0 = wave_sum_u32((uint32_t)v & 0xffffffu);
  const uint32_t d1 = wave_sum_u32((uint32_t)(v >> 24) & 0xffffffu);
  const uint32_t d2 = wave_sum_u32((uint32_t)(v >> 48));
  return (uint64_t)d0 + ((uint64_t)d1 << 24) + ((uint64_t)d2 << 48);
}
// any 64-bit per-lane values (the exact norm kernels): four 16-bit digits
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
  uint64_t tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) tot += (uint64_t)wave_sum_u32((uint32_t)(v >> (16 * i)) & 0xffffu) << (16 * i);
  return tot;
}

// ---- norms ------------------------------------------------------------------------------------------------------
// How many auxiliary primes an exact product needs follows from |a (*) b|_inf <= |a|_2 |b|_2 (Cauchy-Schwarz), so the
// only thing measured per operand is S = sum c^2 — in FLOAT while the coefficients are loaded (v_cvt_f32_i32 +
// v_fma_f32 per coefficient, both full-rate), reduced with wave_sum_f32.  Rounding: the conversion is correct to
// 2^-24, the square to 2^-23, every accumulation step to 2^-24 of the running sum, at most 32 + 6 steps: the float
// total is within a factor (1 +- 2^-18) of S.  kNormSlack = 2^-17 covers that with room.
//   * prime count: S_up = S_float * (1 + kNormSlack) >= S; the bound only has to be safe, never tight.
//   * norm predicate (Params::check_*_constraint, sum c^2 < L with L <= 2^48): decided by the float total whenever
//     it is outside [L (1 - slack), L (1 + slack)], and by exact integer arithmetic (lane_sum_sq_exact) inside, so the
//     verdict is exact for every input: the boundary cases of the tests (flip exactly at (bound+1)^2) take that path.
constexpr float kNormSlack = 0x1p-17f;
template <int E>
__device__ __forceinline__ float lane_sum_sq_f32(const int32_t* v) {   // this lane's share of sum v^2
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const float f = (float)v[e];
    ss = __builtin_fmaf(f, f, ss);
  }
  return ss;
}
// this lane's share of the exact sum of min(|v|, 2^24)^2, saturated at 2^48: the team total equals sum v^2 whenever
// that is below 2^48
template <int E>
__device__ __forceinline__ uint64_t lane_sum_sq_exact(const int32_t* v) {
  uint64_t sq = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t u = (uint32_t)v[e];
    uint32_t a = v[e] < 0 ? 0u - u : u;   // magnitude in unsigned arithmetic (INT32_MIN included)
    a = a < (1u << 24) ? a : (1u << 24);
    sq += (uint64_t)a * a;
  }
  return sq < (1ull << 48) ? sq : (1ull << 48);
}
// Wave-uniform floats are kept in scalar registers: the bounds below live through whole prime passes, where every
// vector register counts (gfx9 has no scalar float ALU, so the arithmetic itself runs on the VALU; v_readfirstlane
// brings the result back).
__device__ __forceinline__ float uniform_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// ---- teams ------------------------------------------------------------------------------------------------------
// Who transforms one polynomial together (Geo<LOGN, LL>, rzk_core.h) and how its threads meet:
//   WaveTeam  one wavefront: the lanes run in lockstep, a "barrier" only stops the compiler from moving LDS accesses
//             across a phase boundary (wave_sync); sums are DPP reductions.
//   PairTeam  two wavefronts that ARE the workgroup (128 threads, N = 2048): s_barrier at the phase boundaries; a
//             sum is two wave reductions exchanged through two LDS words, added in the same order by both waves, so
//             that both take bit-identical decisions (prime counts, exact-path switches) and never part ways
//             before a barrier.
#ifndef RZK_ROW_ROTATE
#define RZK_ROW_ROTATE 1   // row_kernel: rotate the row index per trip when the task stride is a multiple of the row count
#endif
#ifndef RZK_WAVE_TPB
#define RZK_WAVE_TPB 4   // one-wavefront teams per workgroup (experiment: 1 lets a CU hold 19 instead of 16 teams of unit_kernel's 8.1 KB)
#endif
#ifndef RZK_UNIT_MIN_WAVES
#define RZK_UNIT_MIN_WAVES 1   // waves per SIMD unit_kernel<.., false, ..> of one-wavefront teams is compiled for (experiment: 5)
#endif
struct WaveTeam {
  static constexpr int LL = 6;
  static constexpr int kTeamsPerBlock = RZK_WAVE_TPB;
  __device__ __forceinline__ static void sync() { wave_sync(); }
  __device__ __forceinline__ static float sum_f32(float v) { return wave_sum_f32(v); }
  __device__ __forceinline__ static uint64_t sum_u56(uint64_t v) { return wave_sum_u56(v); }
  __device__ __forceinline__ static uint32_t max_u32(uint32_t v) { return wave_max_u32(v); }
};
struct PairTeam {
  static constexpr int LL = 7;
  static constexpr int kTeamsPerBlock = 1;
  __device__ __forceinline__ static void sync() { __syncthreads(); }
  __device__ __forceinline__ static float sum_f32(float v) {
    __shared__ float xf[2];
    const float w = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) xf[(threadIdx.x >> 6) & 1] = w;
    __syncthreads();
    const float tot = xf[0] + xf[1];
    __syncthreads();   // the words are free again
    return uniform_f32(tot);
  }
  __device__ __forceinline__ static uint32_t max_u32(uint32_t v) {
    __shared__ uint32_t xm[2];
    const uint32_t w = wave_max_u32(v);
    if ((threadIdx.x & 63) == 0) xm[(threadIdx.x >> 6) & 1] = w;
    __syncthreads();
    const uint32_t tot = xm[0] > xm[1] ? xm[0] : xm[1];
    __syncthreads();
    return tot;
  }
  __device__ __forceinline__ static uint64_t sum_u56(uint64_t v) {
    __shared__ uint64_t xq[2];
    const uint64_t w = wave_sum_u56(v);
    if ((threadIdx.x & 63) == 0) xq[(threadIdx.x >> 6) & 1] = w;
    __syncthreads();
    const uint64_t tot = xq[0] + xq[1];
    __syncthreads();
    return tot;
  }
};

// BlockPairTeam: two-wavefront teams INSIDE a larger workgroup (row_block_kernel at N = 2048: eight pairs around the
// staged operand transforms).  s_barrier would stop all sixteen waves, so a pair meets through an LDS word of its own:
// the first lane of each wave adds 1 and learns from the returned value which meeting this is — an even old value
// means "I am first": wait until the word has passed old + 2; odd means the partner is already there.  The word only
// grows, so no per-wave generation state is needed; the LDS unit executes one wavefront's instructions in order, so the
// arrive is behind that wave's slab writes and the poll in front of its slab reads (release / acquire at workgroup
// scope keep the compiler honest about it).  The words are cleared once per workgroup (init).
struct BlockPairTeam {
  static constexpr int LL = 7;
  static constexpr int kMaxPairs = 8;
  __device__ __forceinline__ static uint32_t* words() {
    __shared__ uint32_t w[kMaxPairs * 4];   // per pair: meeting counter, pad, two exchange words
    return w + ((threadIdx.x >> 7) & (kMaxPairs - 1)) * 4;
  }
  __device__ __forceinline__ static void init() {
    if ((threadIdx.x & 127) == 0) words()[0] = 0;
    __syncthreads();
  }
  __device__ __forceinline__ static void sync() {
    uint32_t* c = words();
    uint32_t old = 0;
    if ((threadIdx.x & 63) == 0) old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
    const uint32_t target = (old | 1u) + 1u;
    while ((int32_t)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ static float sum_f32(float v) {
    float* xf = reinterpret_cast<float*>(words() + 2);
    const float w = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) xf[(threadIdx.x >> 6) & 1] = w;
    sync();
    const float tot = xf[0] + xf[1];
    sync();   // the words are free again
    return uniform_f32(tot);
  }
  __device__ __forceinline__ static uint32_t max_u32(uint32_t v) {
    uint32_t* xw = words() + 2;
    const uint32_t w = wave_max_u32(v);
    if ((threadIdx.x & 63) == 0) xw[(threadIdx.x >> 6) & 1] = w;
    sync();
    const uint32_t tot = xw[0] > xw[1] ? xw[0] : xw[1];
    sync();
    return tot;
  }
  __device__ __forceinline__ static uint64_t sum_u56(uint64_t v) {   // (rare path: two 28-bit halves through the two words)
    const uint64_t w = wave_sum_u56(v);
    uint32_t* xw = words() + 2;
    uint64_t tot = 0;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      if ((threadIdx.x & 63) == 0) xw[(threadIdx.x >> 6) & 1] = (uint32_t)(w >> (28 * h)) & 0xfffffffu;
      sync();
      tot += ((uint64_t)xw[0] + xw[1]) << (28 * h);
      sync();
    }
    return tot;
  }
};

template <int LOGN, class TM = WaveTeam>
__device__ __forceinline__ void wave_fwd(uint32_t* x, int lane, uint32_t* lds, const uint32_t* __restrict__ tw,
                                         const PrimeConsts& pc) {
  constexpr int LL = TM::LL;
  fwd_phase1<LOGN, LL>(x, tw, pc);
  lds_put_p1<LOGN, LL>(x, lane, lds);
  TM::sync();
  lds_get_p2<LOGN, LL>(x, lane, lds);
  TM::sync();
  fwd_phase2<LOGN, LL>(x, lane, tw, pc);
  lds_put_p2<LOGN, LL>(x, lane, lds);
  TM::sync();
  lds_get_p3<LOGN, LL>(x, lane, lds);
  TM::sync();
  fwd_phase3<LOGN, LL>(x, lane, tw, pc);
}

template <int LOGN, class TM = WaveTeam>
__device__ __forceinline__ void wave_inv(uint32_t* x, int lane, uint32_t* lds, const uint32_t* __restrict__ tw,
                                         const PrimeConsts& pc) {
  constexpr int LL = TM::LL;
  inv_phase3<LOGN, LL>(x, lane, tw, pc);
  lds_put_p3<LOGN, LL>(x, lane, lds);
  TM::sync();
  lds_get_p2<LOGN, LL>(x, lane, lds);
  TM::sync();
  inv_phase2<LOGN, LL>(x, lane, tw, pc);
  lds_put_p2<LOGN, LL>(x, lane, lds);
  TM::sync();
  lds_get_p1<LOGN, LL>(x, lane, lds);
  TM::sync();
  inv_phase1<LOGN, LL>(x, tw, pc);
}

// sum v^2 < limit ?  (limit <= 2^48; ss = the team's float total of the same registers)
template <int E, class TM = WaveTeam>
__device__ __forceinline__ bool norm_below(const int32_t* v, float ss, uint64_t limit) {
  const double s = (double)ss, lim = (double)limit;
  if (s * (1.0 + 2.0 * (double)kNormSlack) < lim) return true;
  if (s * (1.0 - 2.0 * (double)kNormSlack) >= lim) return false;
  return TM::sum_u56(lane_sum_sq_exact<E>(v)) < limit;
}
// upper bound of |.|_2 from the float total
__device__ __forceinline__ float norm2_upper(float ss) {
  return uniform_f32(__builtin_sqrtf(ss * (1.0f + kNormSlack)) * (1.0f + 0x1p-20f));
}
// bound += a * b on wave-uniform non-negative floats (each step is correct to 2^-24; primes_for adds the margin)
__device__ __forceinline__ float bound_fma(float a, float b, float bound) { return uniform_f32(__builtin_fmaf(a, b, bound)); }

// Verdict of a failed norm predicate.  One-bit flags (two_bit == false): the byte is cleared with a plain store
// (idempotent, any number of rows may do it).  Two-bit flags: bit 0 or bit 1 is cleared with an agent-scope
// atomic AND on the aligned word that holds the byte, because rows of one proof on different XCDs may clear
// different bits (the host only enables this when the flag array is word aligned and a multiple of 4 long).
__device__ __forceinline__ void fail_check(uint8_t* flag, bool two_bit, bool second) {
  if (!two_bit) {
    *flag = 0;
    return;
  }
  const uintptr_t a = reinterpret_cast<uintptr_t>(flag);
  const uint32_t bit = (second ? 2u : 1u) << (8u * (uint32_t)(a & 3u));
  __hip_atomic_fetch_and(reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3), ~bit, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
}

// ---- canonical-input test ---------------------------------------------------------------------------------
// A coefficient at the boundary is the centred representative a ZqI64 holds (src/params.rs:122-127): an int64 in
// [-(q-1)/2, (q-1)/2].  The arithmetic below only uses the low word, so every load also proves that the word it
// drops carries no information: with h = (q-1)/2, c is canonical  <=>  (uint64)(c + h) <= q - 1.  The 64-bit add
// is one v_lshl_add_u64; its high word is OR-ed into `bad`, its low word max-ed into `mx` (or, where the 1-norm
// pass already has max |lo|, that is compared with h instead).  A kappa*2^32 + s coefficient is therefore never
// read as s: the proof's verdict flag is cleared and / or the context's sticky input-error word is set.
__device__ __forceinline__ int32_t canon_lo(int64_t c, uint32_t qhalf, uint32_t& bad) {
  bad |= (uint32_t)(((uint64_t)c + qhalf) >> 32);
  return (int32_t)c;
}
__device__ __forceinline__ int32_t canon_lo_mx(int64_t c, uint32_t qhalf, uint32_t& bad, uint32_t& mx) {
  const uint64_t s = (uint64_t)c + qhalf;
  bad |= (uint32_t)(s >> 32);
  const uint32_t lo = (uint32_t)s;
  mx = lo > mx ? lo : mx;
  return (int32_t)c;
}
// the same for a 16-byte load of two coefficients
__device__ __forceinline__ void canon_pair(const longlong2 t, uint32_t qhalf, uint32_t& bad, uint32_t& mx, int32_t& lo0,
                                           int32_t& lo1) {
  lo0 = canon_lo_mx(t.x, qhalf, bad, mx);
  lo1 = canon_lo_mx(t.y, qhalf, bad, mx);
}
// wave-uniform verdict of the per-lane accumulators (mx holds max (lo + h) mod 2^32, canonical <=> <= 2h)
__device__ __forceinline__ bool canon_fail(uint32_t bad, uint32_t mx, uint32_t qhalf) {
  return __any((bad != 0) | (mx > 2u * qhalf)) != 0;
}
// A non-canonical coefficient was loaded for proof `bo`: clear its verdict (all bits) and raise the sticky word.
__device__ __forceinline__ void input_fault(const Operands& ops, uint8_t* flags, uint32_t bo, int lane) {
  if ((lane & 63) != 0) return;   // the first lane of the wavefront that saw the fault (teams of two report per wave)
  if (flags) {
    if (ops.pad) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(flags + bo);
      __hip_atomic_fetch_and(reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3), ~(0xffu << (8u * (uint32_t)(a & 3u))),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      flags[bo] = 0;
    }
  }
  if (ops.bad) *ops.bad = 1u;
}

__device__ __forceinline__ const int64_t* operand_ptr(const Operands& ops, uint32_t op, uint32_t off,
                                                       uint32_t b, uint32_t bo, int n_coef) {
  uint32_t idx = ops.outer[op] ? bo : b;
#ifdef RZK_EXPERIMENT_ALIAS   // diagnostic builds only: every batch entry uses the data of entry (index mod 64), so
#ifndef RZK_EXPERIMENT_ALIAS_OPS
#define RZK_EXPERIMENT_ALIAS_OPS 0xffffffffu   // bit i: operand i of the row program is aliased (e.g. Open commit: 7 = x, r, y read; 24 = c, t written)
#endif
  if ((RZK_EXPERIMENT_ALIAS_OPS >> op) & 1u)
    idx &= (uint32_t)(RZK_EXPERIMENT_ALIAS - 1);   // (a power of two: 64 keeps operands and results in L2, 1024 in the Infinity Cache) — what the launch would cost without its HBM traffic
#endif
  return ops.base[op] + ((uint64_t)idx * ops.stride[op] + off) * (uint64_t)n_coef;
}

#ifndef RZK_EPI_CHUNK
#define RZK_EPI_CHUNK 16  // coefficients per lane handled together in the epilogue (4 was slower: fewer loads in flight)
#endif
// Opaque copy of the lane id inside the loops: stops the compiler from hoisting every lane-dependent
// address out of the loops (where they sit in dozens of VGPRs) at the price of recomputing them per
// term.  At N = 1024 it costs the transform-only rows ~5 % (123 -> 87 VGPRs, but LDS already caps the kernel
// at 4 waves per SIMD), so there it is used only (template flag OPQ) by the rows that start with a shift term,
// which it keeps below 128 VGPRs (144 -> 101); at N = 2048 it takes the kernel from 254 VGPRs (1 wave per
// SIMD) to ~125 (4 waves).
#ifndef RZK_OPAQUE_LANE_MIN_LOGN
#define RZK_OPAQUE_LANE_MIN_LOGN 11
#endif
#define RZK_OPAQUE(v)                                                     \
  do {                                                                    \
    if (LOGN >= RZK_OPAQUE_LANE_MIN_LOGN || OPQ) asm volatile("" : "+v"(v)); \
  } while (0)

// Load one coefficient polynomial (coalesced phase-1 layout) and lift it into prime field `pc`.
// measure (the first prime pass): nrm2 = an upper bound of the polynomial's 2-norm (wave-uniform), and — check — the
// fused norm predicate sum c^2 < limit, exact (norm_below); unless `trusted`, the same pass proves that every
// coefficient is canonical (canon_lo_mx).  Later passes re-read the low words only.
// How the measure pass is laid out (MODE): the arithmetic is the same, the register footprint is not.
//   LL_FUSED   every coefficient tested, squared and lifted as it arrives, all E loads in flight (unit_kernel)
//   LL_HALVES  the same in two rolled halves: E/2 sixty-four-bit coefficients in flight
//   LL_L1INF   all E loads in flight, low words into an int array first; sum v^2 bounded by |v|_1 |v|_inf
enum : int { LL_FUSED = 0, LL_HALVES = 1, LL_L1INF = 2 };
// row_kernel keeps its running sum (and, in a vector x vector term, the first operand's transform) in registers across
// the load.  Measured on the Sum (4,9,4) / Linear configurations (A/B of prebuilt libraries, round 3): LL_L1INF for
// both operands 224-225 k / 4.35 M proofs/s, LL_HALVES for both 221 k / 4.26 M, LL_FUSED spills (128 VGPRs + 156
// bytes of scratch: 213 k / 4.10 M).  Teams of two (N = 2048) stay spill-free only with LL_HALVES.
#ifndef RZK_ROW_MODE_B
#define RZK_ROW_MODE_B (TM::LL == 7 ? LL_HALVES : LL_L1INF)   // row_kernel, a term's first operand
#endif
#ifndef RZK_ROW_MODE_A
#define RZK_ROW_MODE_A (TM::LL == 7 ? LL_HALVES : LL_L1INF)   // ... second operand of a vector x vector term
#endif
template <int LOGN, class TM = WaveTeam, int MODE = LL_FUSED>
__device__ __forceinline__ void load_lift(uint32_t* x, const int64_t* __restrict__ src, int lane, const PrimeConsts& pc,
                                          bool measure, float& nrm2, bool check, uint64_t limit, bool& below,
                                          uint32_t qhalf, bool trusted, bool& fault) {
  using G = Geo<LOGN, TM::LL>;
  if (measure) {
    // one pass: the 64-bit coefficient is tested, squared into the float sum and lifted as soon as it arrives, so that
    // only the lifted residues stay in registers (no second copy of the polynomial)
    float part = 0.f;
    if (MODE == LL_L1INF) {
      int32_t v[G::E];
      uint32_t bad = 0;
#pragma unroll
      for (int e = 0; e < G::E; ++e) v[e] = trusted ? (int32_t)ld_stream(src + G::j_p1(lane, e)) : canon_lo(ld_stream(src + G::j_p1(lane, e)), qhalf, bad);
      uint64_t sum = 0;
      uint32_t mxa = 0;
#pragma unroll
      for (int e = 0; e < G::E; e += 2) {
        const uint32_t u0 = (uint32_t)v[e], u1 = (uint32_t)v[e + 1];
        const uint32_t a0 = v[e] < 0 ? 0u - u0 : u0;
        const uint32_t a1 = v[e + 1] < 0 ? 0u - u1 : u1;
        sum += (uint64_t)a0 + a1;
        mxa = a0 > mxa ? a0 : mxa;
        mxa = a1 > mxa ? a1 : mxa;
      }
      const float l1 = (float)TM::sum_u56(sum) * (1.0f + 0x1p-20f);
      const uint32_t wmx = TM::max_u32(mxa);
      if (!trusted) fault = fault || __any(bad != 0) || wmx > qhalf;
      // sum v^2 <= |v|_1 |v|_inf; handed on as if every thread of the team carried an equal share
      part = l1 * (float)wmx * (1.0f + 0x1p-20f) * (1.0f / (float)G::LANES);
      if (check) {   // the exact predicate needs the exact sum: float squares of the same registers
        float sq = lane_sum_sq_f32<G::E>(v);
        const float ssq = TM::sum_f32(sq);
        const double sd = (double)ssq, lim = (double)limit;
        if (sd * (1.0 + 2.0 * (double)kNormSlack) < lim) below = true;
        else if (sd * (1.0 - 2.0 * (double)kNormSlack) >= lim) below = false;
        else below = TM::sum_u56(lane_sum_sq_exact<G::E>(v)) < limit;
      }
#pragma unroll
      for (int e = 0; e < G::E; ++e) x[e] = lift(v[e], pc);
    } else if (MODE == LL_HALVES) {
      // two rolled halves: E/2 sixty-four-bit coefficients in flight instead of E.  For the kernels that keep a running
      // sum in registers across the load (row_kernel, the group and slot kernels) this is what fits 128 VGPRs without
      // spilling (row_kernel<10>: 116 VGPRs against 128 + 156 bytes of scratch); unit_kernel, with nothing else live,
      // is better off with all loads in flight at once (116 against 132 VGPRs).
      uint32_t bad = 0, mx = 0;
      constexpr int H = G::E / 2;
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        uint32_t y[H];
#pragma unroll
        for (int e2 = 0; e2 < H; ++e2) {
          const int64_t c = ld_stream(src + (size_t)(h * H + e2) * G::LANES + lane);
          const int32_t v = trusted ? (int32_t)c : canon_lo_mx(c, qhalf, bad, mx);
          const float f = (float)v;
          part = __builtin_fmaf(f, f, part);
          y[e2] = lift(v, pc);
        }
#pragma unroll
        for (int e2 = 0; e2 < H; ++e2) {
          x[e2] = h == 0 ? y[e2] : x[e2];
          x[H + e2] = h == 1 ? y[e2] : x[H + e2];
        }
      }
      if (!trusted) fault = fault || canon_fail(bad, mx, qhalf);
    } else if (trusted) {
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        const int32_t v = (int32_t)ld_stream(src + G::j_p1(lane, e));
        const float f = (float)v;
        part = __builtin_fmaf(f, f, part);
        x[e] = lift(v, pc);
      }
    } else {
      uint32_t bad = 0, mx = 0;
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        const int32_t v = canon_lo_mx(ld_stream(src + G::j_p1(lane, e)), qhalf, bad, mx);
        const float f = (float)v;
        part = __builtin_fmaf(f, f, part);
        x[e] = lift(v, pc);
      }
      fault = fault || canon_fail(bad, mx, qhalf);
    }
    const float ss = TM::sum_f32(part);
    nrm2 = norm2_upper(ss);
    if (check && MODE != LL_L1INF) {
      const double sd = (double)ss, lim = (double)limit;
      if (sd * (1.0 + 2.0 * (double)kNormSlack) < lim) {
        below = true;
      } else if (sd * (1.0 - 2.0 * (double)kNormSlack) >= lim) {
        below = false;
      } else {   // inside the rounding band of the limit: exact integers, from the lifted residues (v = x - 2p)
        int32_t v[G::E];
#pragma unroll
        for (int e = 0; e < G::E; ++e) v[e] = (int32_t)(x[e] - pc.twop);
        below = TM::sum_u56(lane_sum_sq_exact<G::E>(v)) < limit;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < G::E; ++e) x[e] = lift((int32_t)ld_stream(src + G::j_p1(lane, e)), pc);
  }
}

// =============================================================================================
// Row-program kernels: the fused product / accumulate / reduce pipeline of every protocol phase.
//
// A wavefront owns polynomial-sized pieces of one proof (rzk_dev.h).  Control flow is wave-uniform and scalar (the
// wave index is read with readfirstlane).  Primes are processed one after the other; each inverse transform is folded
// at once into the running Garner state (rzk_core.h, crt_fold*), so no state occupies registers during the transforms.
// The operands' norms are measured while they are loaded for the first prime, which fixes how many primes (1..3) the
// exact result needs; the same pass proves that every coefficient is canonical.  Which kernel runs a program is
// decided once per (program, shape) in rzk_api.cpp:
//
//   unit_kernel       key-product programs (the default): one wavefront per proof walks the program's units — single
//                     rows, or pairs of rows that share their last operand; sums parked in LDS, Garner words in
//                     per-wave global scratch lines.
//   row_kernel        programs with vector x vector products: one wavefront per row, sum in registers, Garner word A
//                     in LDS.
//   shift_row_kernel  rows whose products all have the sparse challenge as multiplier: rotations, no transform.
//   row_group_kernel  (N <= 1024) / row_block_kernel (N = 2048): key blocks with n > 1, operands transformed once for
//                     several rows.
//   fwd_slots_kernel  + row_slots_kernel: when many rows of a proof use the same operands (sums over V summands at large
//                     shapes), every distinct operand ("slot") is transformed ONCE per proof into a workspace in HBM,
//                     and the rows only multiply-accumulate the stored transforms; rows that need more primes than
//                     were stored fall back to in-wave transforms for the missing primes, so results stay exact.
// =============================================================================================
#ifndef RZK_ROW_MIN_WAVES
#define RZK_ROW_MIN_WAVES 1   // minimum waves per SIMD the row kernels are compiled for (register budget)
#endif

// ---- challenge products as signed rotations (ShiftGeo, rzk_core.h): shared by shift_row_kernel and the
// shift terms of row_kernel ---------------------------------------------------------------------------------

template <int LOGN, int LL = 6>
__device__ __forceinline__ void load_pairs(int32_t* v, const int64_t* __restrict__ src, int lane, uint32_t qhalf,
                                           uint32_t& bad, uint32_t& mx, bool trusted) {
  using S = ShiftGeo<LOGN, true, LL>;
  const longlong2* __restrict__ p = reinterpret_cast<const longlong2*>(src);
  if (trusted) {
#pragma unroll
    for (int g = 0; g < S::G; ++g) {
      const longlong2 t = ld_stream(p + g * S::LANES + lane);
      v[2 * g] = (int32_t)t.x, v[2 * g + 1] = (int32_t)t.y;
    }
  } else {
#pragma unroll
    for (int g = 0; g < S::G; ++g) canon_pair(ld_stream(p + g * S::LANES + lane), qhalf, bad, mx, v[2 * g], v[2 * g + 1]);   // coefficients g*2*LANES + 2*lane, +1
  }
}

#ifndef RZK_SHIFT_H
#define RZK_SHIFT_H 8   // outputs of a lane accumulated per scan over the multiplier's non-zeros (N = 2048 response rows, round 3: 160 us; 4 -> 186, 16 -> 181)
#endif
#ifndef RZK_SHIFT_H_MEM
#define RZK_SHIFT_H_MEM 16   // ... for the rotation terms inside the row kernels (sums go to the wave's scratch line)
#endif
#ifndef RZK_SHIFT_H_MEM_PAIR
#define RZK_SHIFT_H_MEM_PAIR 8   // ... of a two-wavefront team (16 measured slower: verify at N = 2048 189 vs 184 us, 40 vs 8 bytes of scratch)
#endif
// walk the non-zero coefficients of the multiplier (registers a[], lane-distributed in layout PAIR) and add
// the rotations into IN outputs of every lane; `ext` already points at the first of them
template <int LOGN, bool PAIR, int IN>
__device__ __forceinline__ void shift_scan(int64_t* acc, const int32_t* a, int lane, const int32_t* ext) {
  using S = ShiftGeo<LOGN, PAIR>;
#pragma unroll
  for (int i = 0; i < S::E; ++i) {
    uint64_t mask = __ballot(a[i] != 0);
    while (mask) {
      const int l = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int32_t coef = __builtin_amdgcn_readlane(a[i], l);
      const int s = S::off(i) + (PAIR ? 2 * l : l);
      int ln = lane;
      asm volatile("" : "+v"(ln));   // keeps the 16 per-register base addresses from being hoisted into VGPRs
      shift_accum<LOGN, PAIR, int64_t, 0, IN>(acc, ln, s, coef, ext);
    }
  }
}
// (Taking two non-zeros per trip, or sixteen outputs per scan, to keep more LDS reads in flight was measured
// slower: the extra registers cost a wave per SIMD.)

// ---- teams of two wavefronts: the multiplier's non-zeros as a list in LDS (kShiftListCap entries per round) ----
// number of non-zero coefficients this WAVEFRONT holds
template <int E>
__device__ __forceinline__ uint32_t shift_count_nonzeros(const int32_t* a) {
  uint32_t cnt = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) cnt += (uint32_t)__builtin_popcountll(__ballot(a[i] != 0));
  return cnt;
}
// entries [r0, r0 + cap) of the team's list; `base` = entries of the wavefronts before this one
template <int LOGN, bool PAIR, int LL>
__device__ __forceinline__ void shift_list_write(const int32_t* a, int lane, uint32_t base, uint32_t r0, int32_t* list) {
  using S = ShiftGeo<LOGN, PAIR, LL>;
  uint32_t run = base - r0;   // (mod 2^32: entries before the window wrap to huge indices and are skipped)
  int2* ent = reinterpret_cast<int2*>(list);
#pragma unroll
  for (int i = 0; i < S::E; ++i) {
    const uint64_t m = __ballot(a[i] != 0);
    const uint32_t idx = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (a[i] != 0 && idx < (uint32_t)kShiftListCap) ent[idx] = make_int2(S::j(lane, i), a[i]);
    run += (uint32_t)__builtin_popcountll(m);
  }
}
// add the rotations of the list's first `nent` entries into IN outputs of every thread
template <int LOGN, bool PAIR, int LL, int IN>
__device__ __forceinline__ void shift_scan_list(int64_t* acc, const int32_t* list, uint32_t nent, int lane, const int32_t* ext) {
  const int2* ent = reinterpret_cast<const int2*>(list);
  const int l64 = lane & 63;
#pragma unroll 1
  for (uint32_t e0 = 0; e0 < nent; e0 += 64) {
    const uint32_t m = nent - e0 < 64u ? nent - e0 : 64u;
    const int2 mine = (uint32_t)l64 < m ? ent[e0 + l64] : make_int2(0, 0);   // 64 entries per trip, one per lane
#pragma unroll 1
    for (uint32_t e = 0; e < m; ++e) {
      const int s = __builtin_amdgcn_readlane(mine.x, (int)e);
      const int32_t coef = __builtin_amdgcn_readlane(mine.y, (int)e);
      int ln = lane;
      asm volatile("" : "+v"(ln));
      shift_accum<LOGN, PAIR, int64_t, 0, IN, LL>(acc, ln, s, coef, ext);
    }
  }
}

// Build the wave's 2N-word extended image of v (ShiftGeo, rzk_core.h) straight from global memory, in two rolled
// halves so that only E/2 sixty-four-bit coefficients are in flight at a time.  measure: this is the first fill —
// it also proves that v is canonical (canon_lo) and returns max |v| over the lane's coefficients.
template <int LOGN, bool PAIR, int LL = 6>
__device__ __forceinline__ void shift_fill_from(const int64_t* __restrict__ pv, int lane, int32_t* ext, int part,
                                                bool measure, bool canon, uint32_t qhalf, uint32_t& bad, uint32_t& mx,
                                                uint32_t& maxabs) {
  using S = ShiftGeo<LOGN, PAIR, LL>;
  constexpr int H = S::E / 2;               // registers per half; off(h*H + i) = off(i) + h * H * LANES in both layouts
  constexpr int HOFF = H * S::LANES;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    int32_t vh[H];
    if (PAIR) {
      const longlong2* __restrict__ p = reinterpret_cast<const longlong2*>(pv) + (size_t)h * (H / 2) * S::LANES;
#pragma unroll
      for (int g = 0; g < H / 2; ++g) {
        const longlong2 t = ld_stream(p + g * S::LANES + lane);   // coefficients (h*H/2 + g)*2*LANES + 2*lane, +1
        if (canon) {
          canon_pair(t, qhalf, bad, mx, vh[2 * g], vh[2 * g + 1]);
        } else {
          vh[2 * g] = (int32_t)t.x;
          vh[2 * g + 1] = (int32_t)t.y;
        }
      }
    } else {
      const int64_t* __restrict__ p = pv + (size_t)h * H * S::LANES;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const int64_t c = p[i * S::LANES + lane];
        vh[i] = canon ? canon_lo_mx(c, qhalf, bad, mx) : (int32_t)c;
      }
    }
    if (measure) {
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const uint32_t uu = (uint32_t)vh[i];
        const uint32_t vv = vh[i] < 0 ? 0u - uu : uu;
        maxabs = vv > maxabs ? vv : maxabs;
      }
    }
    int32_t* base = ext + S::lane_base(lane) + h * HOFF;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const int32_t a = shift_part(vh[i], part);
      base[S::N + S::off(i)] = a;
      base[S::off(i)] = -a;
    }
  }
}

// res[] (in [0,q)) +/-= (a (*) v) mod q for one product term; a[] holds the multiplier's low words in layout
// PAIR, pv points at the other operand.  ext: the team's 2N-word LDS image (teams of two: followed by the
// kShiftListWords words of the non-zero list).  Team-uniform control flow.
// TO_MEM: res is a per-team line in global memory indexed by coefficient (each thread touches only its own
// coefficients) and `fresh` says that it holds nothing yet; otherwise res are the thread's E registers.
// Sums are exact 64-bit integers (v_mad_i64_i32) as long as |a|_1 |v|_inf < 2^62; beyond that v goes in as
// two 16-bit halves.  Eight of a thread's outputs are accumulated at a time (register budget).
// fault: set when v holds a non-canonical coefficient (the caller tests `a`).
template <int LOGN, bool PAIR, bool TO_MEM, class TM = WaveTeam>
__device__ __forceinline__ void shift_product(uint32_t* res, bool fresh, bool minus, const int32_t* a,
                                              const int64_t* __restrict__ pv, int lane_in, int32_t* ext,
                                              const DevTables& T, bool& fault, bool trusted) {
  constexpr int LL = TM::LL;
  int lane = lane_in;
  if (LL != 6) asm volatile("" : "+v"(lane));   // per call: keeps the thread's 64-bit line / image addresses out of the kernel prologue
  using S = ShiftGeo<LOGN, PAIR, LL>;
  constexpr int E = S::E;
  constexpr int HW = TO_MEM ? (LL == 6 ? RZK_SHIFT_H_MEM : RZK_SHIFT_H_MEM_PAIR) : RZK_SHIFT_H;   // (the in-kernel rotation terms run with nothing else live)
  constexpr int H = HW < E ? HW : E;   // outputs per scan; chunk c covers registers c*H .. c*H+H-1
  constexpr int NCH = E / H;
  constexpr bool LIST = LL != 6;
  const uint32_t q = T.crt.q, qhalf = T.crt.qhalf;
  // optimistic first fill with the whole values; it also measures v
  uint32_t vbad = 0, vmx = 0, maxv = 0;
  TM::sync();   // earlier reads of the image are done before it is overwritten
  shift_fill_from<LOGN, PAIR, LL>(pv, lane, ext, SHIFT_WHOLE, true, !trusted, qhalf, vbad, vmx, maxv);
  if (!trusted) fault = fault || canon_fail(vbad, vmx, qhalf);
  uint64_t suma = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t ua = (uint32_t)a[i];
    suma += a[i] < 0 ? 0u - ua : ua;
  }
  const double bound = (double)TM::sum_u56(suma) * (double)TM::max_u32(maxv);   // |exact product|_inf (E * 2^31 < 2^56 per lane)
  const int npass = __builtin_amdgcn_readfirstlane(bound < 4.0e18 ? 1 : 2);        // 4.0e18 < 2^62
  // teams of two: where this wavefront's non-zeros go in the list, and how many there are in all
  int32_t* list = ext + S::WORDS;
  uint32_t lbase = 0, ltotal = 1;   // (one wavefront: a single "round", the multiplier is walked in registers)
  if (LIST) {
    const uint32_t mine = shift_count_nonzeros<E>(a);
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(lane >> 6);
    if ((lane & 63) == 0) list[2 * kShiftListCap + w] = (int32_t)mine;
    TM::sync();   // (also orders the image's fill before the first scan)
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane(list[2 * kShiftListCap]);
    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readfirstlane(list[2 * kShiftListCap + 1]);
    lbase = w ? c0 : 0u;
    ltotal = c0 + c1;
  }
  bool started = false;   // the TO_MEM line holds this product's partial sums
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    if (npass == 2) {   // (never for a sparse +-1 challenge) the image is rebuilt from 16-bit halves
      uint32_t u0 = 0, u1 = 0, u2 = 0;
      TM::sync();
      shift_fill_from<LOGN, PAIR, LL>(pv, lane, ext, pass == 0 ? SHIFT_LOW16 : SHIFT_HIGH16, false, false, qhalf, u0, u1, u2);
    }
    if (!LIST || npass == 2) TM::sync();
#pragma unroll 1
    for (uint32_t r0 = 0; r0 < (ltotal ? ltotal : 1u); r0 += LIST ? (uint32_t)kShiftListCap : 1u) {   // (a zero multiplier still initialises the sums)
      uint32_t nent = 0;
      if (LIST) {
        if (r0 || pass) TM::sync();   // the previous round's scans are over
        shift_list_write<LOGN, PAIR, LL>(a, lane, lbase, r0, list);
        TM::sync();
        nent = ltotal - r0 < (uint32_t)kShiftListCap ? ltotal - r0 : (uint32_t)kShiftListCap;   // (0 when there is no non-zero at all)
      }
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
        int64_t acc[H];
#pragma unroll
        for (int i = 0; i < H; ++i) acc[i] = 0;
        // off(c*H + i) = off(i) + LANES H c in both layouts
        if (LIST) shift_scan_list<LOGN, PAIR, LL, H>(acc, list, nent, lane, ext + ch * (S::LANES * H));
        else shift_scan<LOGN, PAIR, H>(acc, a, lane, ext + ch * (S::LANES * H));
#pragma unroll
        for (int i = 0; i < H; ++i) {
          uint32_t u = zq_from_i64(acc[i], T.crt);
          if (pass) u = montq_u(u, T.crt.r48q, T.crt);   // high halves carry the weight 2^16
          if (TO_MEM) {
            uint32_t* slot = res + S::j(lane, i) + ch * (S::LANES * H);
            const uint32_t cur = (fresh && !started) ? 0u : *slot;
            *slot = minus ? subq(cur, u, q) : addq(cur, u, q);
          } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {   // register index c*H + i, selected without dynamic indexing
              const uint32_t cur = res[c * H + i];
              const uint32_t nw = minus ? subq(cur, u, q) : addq(cur, u, q);
              res[c * H + i] = c == ch ? nw : cur;
            }
          }
        }
      }
      started = true;
    }
  }
}

// Producer side of Operands::oimg: the transform x (prime pi) of operand (op, off) of batch entry b, as it leaves wave_fwd
template <int LOGN, class TM>
__device__ __forceinline__ void store_operand_image(const uint32_t* x, const Operands& ops, uint32_t op, uint32_t off, uint32_t b,
                                                    int pi, int lane, float nrm2, bool first) {
  using G = Geo<LOGN, TM::LL>;
  if (!ops.oimg || op != ops.oimg_op || off >= 32u) return;
  const int ci = ops.oimg_col[off];
  if (ci < 0) return;
  const size_t oslot = (size_t)b * ops.oimg_n + (uint32_t)ci;
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(ops.oimg + (oslot * kKeyImages + pi) * G::N);
#pragma unroll
  for (int g = 0; g < G::E / 4; ++g) dst[G::key4(lane, g)] = make_uint4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
  if (lane == 0) {
    if (first) ops.oimg_l2[oslot] = (double)nrm2;
    ops.oimg_np[oslot] = (uint8_t)(pi + 1);   // primes 0 .. pi are there (the passes run in this order)
  }
}

// acc +/- (term) for prime `pi`, transforming the term's operands in the wave.
template <int LOGN, bool HAS_VEC, bool OPQ = false, class TM = WaveTeam, bool DD = false>
__device__ __forceinline__ void term_direct(uint32_t* acc, const Term tm, const Operands& ops, uint32_t b,
                                            uint32_t bo, int lane, uint32_t* lds, const uint32_t* __restrict__ twf,
                                            const PrimeConsts& pc, int pi, const uint32_t* __restrict__ key_ntt,
                                            const double* __restrict__ key_l2, bool first, float& bound,
                                            uint8_t* __restrict__ flags, uint32_t qhalf) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  // optional opaque copy of the lane id (RZK_OPAQUE): stops hoisting of lane-dependent addresses
  int ln = lane;
  RZK_OPAQUE(ln);
  uint32_t x[E];
  float nb = 0.f;
  bool below = true, fault = false;
  const bool chk = first && (tm.kind & (TERM_CHECK | TERM_CHECK2));
  const bool trusted = ops.trusted != 0;
  // TERM_DD: the operand's transform under this prime may already lie in the call's operand images
  bool from_image = false;
  size_t oslot = 0;
  if (DD && (tm.kind & TERM_KIND_MASK) == TERM_DD && ops.oimg) {   // (DD is a template flag: as a run-time test in every row kernel it changed the compiler's load scheduling of the ordinary rows — Linear -2 %)
    const uint32_t summand = tm.b_off / ops.oimg_k, col = tm.b_off - summand * ops.oimg_k;
    const int ci = col < 32u ? ops.oimg_col[col] : -1;
    if (ci >= 0) {
      oslot = ((size_t)bo * ops.oimg_group + summand) * ops.oimg_n + (uint32_t)ci;
      from_image = (int)ops.oimg_np[oslot] > pi;
    }
  }
  if (DD) from_image = __builtin_amdgcn_readfirstlane((int)from_image) != 0;
  if (DD && from_image) {
    const uint4* __restrict__ ip = reinterpret_cast<const uint4*>(ops.oimg + (oslot * kKeyImages + pi) * N);
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      const uint4 iv = ip[G::key4(ln, g)];
      x[4 * g] = iv.x, x[4 * g + 1] = iv.y, x[4 * g + 2] = iv.z, x[4 * g + 3] = iv.w;
    }
    nb = (float)ops.oimg_l2[oslot];
  } else {
    load_lift<LOGN, TM, RZK_ROW_MODE_B>(x, operand_ptr(ops, tm.b_op, tm.b_off, b, bo, N), ln, pc, first, nb, chk, ops.norm_limit, below, qhalf, trusted, fault);
    if (chk && !below && (lane & 63) == 0) fail_check(flags + bo, ops.pad != 0, (tm.kind & TERM_CHECK2) != 0);
    wave_fwd<LOGN, TM>(x, ln, lds, twf, pc);
  }
  if (HAS_VEC && (tm.kind & TERM_KIND_MASK) == TERM_VEC) {
    // product of two per-proof polynomials: fold N^-1 and the Montgomery factor into one of them
    uint32_t xb[E];
#pragma unroll
    for (int c = 0; c < E; ++c) xb[c] = csub(mont_lazy(x[c], pc.ninv_r2, pc.p, pc.npinv), pc.p);
    float na = 0.f;
    bool unused_below = true;
    load_lift<LOGN, TM, RZK_ROW_MODE_A>(x, operand_ptr(ops, tm.a_op, tm.a_off, b, bo, N), ln, pc, first, na, false, 0, unused_below, qhalf, trusted, fault);
    wave_fwd<LOGN, TM>(x, ln, lds, twf, pc);
    if (first) bound = bound_fma(na, nb, bound);   // |a (*) b|_inf <= |a|_2 |b|_2
    if (tm.sign >= 0) {
#pragma unroll
      for (int c = 0; c < E; ++c) acc[c] = mac_add(acc[c], x[c], xb[c], pc);
    } else {
#pragma unroll
      for (int c = 0; c < E; ++c) acc[c] = mac_sub(acc[c], x[c], xb[c], pc);
    }
  } else {
    // resident key entry, or (TERM_DKEY) one of the batch entry's own multiplier images — same form, same use
    const bool dk = (tm.kind & TERM_KIND_MASK) == TERM_DKEY || (tm.kind & TERM_KIND_MASK) == TERM_DD;
    const size_t image = dk ? (size_t)bo * ops.dkey_n + tm.a_off : (size_t)tm.a_off;
    if (first) bound = bound_fma((float)(dk ? ops.dkey_l2[image] : key_l2[image]), nb, bound);
    const uint4* __restrict__ kp = reinterpret_cast<const uint4*>((dk ? ops.dkey_img : key_ntt) + (image * kKeyImages + pi) * N);
    if (tm.sign >= 0) {
#pragma unroll
      for (int g = 0; g < E / 4; ++g) {
        const uint4 kv = kp[G::key4(ln, g)];
        acc[4 * g + 0] = mac_add(acc[4 * g + 0], x[4 * g + 0], kv.x, pc);
        acc[4 * g + 1] = mac_add(acc[4 * g + 1], x[4 * g + 1], kv.y, pc);
        acc[4 * g + 2] = mac_add(acc[4 * g + 2], x[4 * g + 2], kv.z, pc);
        acc[4 * g + 3] = mac_add(acc[4 * g + 3], x[4 * g + 3], kv.w, pc);
      }
    } else {
#pragma unroll
      for (int g = 0; g < E / 4; ++g) {
        const uint4 kv = kp[G::key4(ln, g)];
        acc[4 * g + 0] = mac_sub(acc[4 * g + 0], x[4 * g + 0], kv.x, pc);
        acc[4 * g + 1] = mac_sub(acc[4 * g + 1], x[4 * g + 1], kv.y, pc);
        acc[4 * g + 2] = mac_sub(acc[4 * g + 2], x[4 * g + 2], kv.z, pc);
        acc[4 * g + 3] = mac_sub(acc[4 * g + 3], x[4 * g + 3], kv.w, pc);
      }
    }
  }
  if (fault) input_fault(ops, flags, bo, lane);
}

// inverse transform of the prime-`pi` accumulator and fold into the Garner state: word A in LDS, word B
// (third prime only) in the per-wave global scratch line.  acc is clobbered.
template <int LOGN, bool OPQ = false, class TM = WaveTeam>
__device__ __forceinline__ void inverse_and_fold(int pi, int np, uint32_t* acc, int lane, uint32_t* lds,
                                                 const uint32_t* __restrict__ twi, const PrimeConsts& pc,
                                                 uint32_t* st_lds, uint32_t* __restrict__ st_glb, const DevTables& T) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  int li = lane;
  RZK_OPAQUE(li);
  wave_inv<LOGN, TM>(acc, li, lds, twi, pc);
  if (pi == 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) st_lds[G::j_p1(li, e)] = crt_fold0(acc[e], np, T.pc, T.crt);
  } else if (pi == 1) {
    uint32_t d0[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      d0[e] = st_lds[G::j_p1(li, e)];
      acc[e] = crt_digit1(acc[e], d0[e], np, T.pc, T.crt);
    }
    if (np == 3) {
#pragma unroll
      for (int e = 0; e < E; ++e) st_glb[G::j_p1(li, e)] = crt_value01_modp2(d0[e], acc[e], T.pc, T.crt);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) st_lds[G::j_p1(li, e)] = crt_value01_modq(d0[e], acc[e], T.crt);
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint32_t a = st_lds[G::j_p1(li, e)];
      crt_fold2(acc[e], T.pc, T.crt, a, st_glb[G::j_p1(li, e)]);
      st_lds[G::j_p1(li, e)] = a;
    }
  }
}

// Checked additions (ADD_CHECK / ADD_CHECK2: the host marks them only among the first four additions of a row): the
// fused norm predicate sum c^2 < limit of the polynomial an addition loads.  The epilogues accumulate the float sum
// of squares per marked addition while they load it; the verdict is taken here, exactly (see "norms" above: float
// total outside the rounding band of the limit, otherwise the polynomial is re-read and summed in integers).
template <int LOGN, class TM = WaveTeam>
__device__ __forceinline__ void checked_add_verdicts(const Program* __restrict__ prog, const Row row, const Operands& ops,
                                                     uint32_t b, uint32_t bo, int lane, const float* add_ss,
                                                     uint8_t* __restrict__ flags) {
  using G = Geo<LOGN, TM::LL>;
#pragma unroll 1
  for (uint32_t a = 0; a < row.nadds && a < 4; ++a) {
    const AddTerm ad = table_load(&prog->adds[row.add0 + a]);
    if (!(ad.op & (ADD_CHECK | ADD_CHECK2))) continue;
    float part = 0.f;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) part = (sl == (int)a) ? add_ss[sl] : part;
    const double sfl = (double)TM::sum_f32(part), lim = (double)ops.norm_limit;
    bool below;
    if (sfl * (1.0 + 2.0 * (double)kNormSlack) < lim) {
      below = true;
    } else if (sfl * (1.0 - 2.0 * (double)kNormSlack) >= lim) {
      below = false;
    } else {
      const int64_t* __restrict__ src = operand_ptr(ops, ad.op & ADD_OP_MASK, ad.off, b, bo, G::N);
      int32_t v[G::E];
#pragma unroll
      for (int e = 0; e < G::E; ++e) v[e] = (int32_t)src[G::j_p1(lane, e)];
      below = TM::sum_u56(lane_sum_sq_exact<G::E>(v)) < ops.norm_limit;
    }
    if (!below && (lane & 63) == 0) fail_check(flags + bo, ops.pad != 0, (ad.op & ADD_CHECK2) != 0);
  }
}

// One chunk of one plain addition: u[i] +/-= operand coefficient (j_p1(lane, e0 + i)) in 32-bit arithmetic mod q;
// canonical test unless trusted; float sum of squares into add_ss[slot] for checked additions.
template <int LOGN, int CH, class TM = WaveTeam>
__device__ __forceinline__ void add_chunk(uint32_t* u, const AddTerm ad, uint32_t a, const int64_t* __restrict__ src, int lane,
                                          int e0, uint32_t q, uint32_t qhalf, bool trusted, uint32_t& in_bad, uint32_t& in_mx,
                                          float* add_ss) {
  using G = Geo<LOGN, TM::LL>;
  int32_t av[CH];
  if (trusted) {
#pragma unroll
    for (int i = 0; i < CH; ++i) av[i] = (int32_t)ld_stream(src + G::j_p1(lane, e0 + i));
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) av[i] = canon_lo_mx(ld_stream(src + G::j_p1(lane, e0 + i)), qhalf, in_bad, in_mx);
  }
  if (ad.op & (ADD_CHECK | ADD_CHECK2)) {
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const float f = (float)av[i];
      sq = __builtin_fmaf(f, f, sq);
    }
    const uint32_t slot = a < 4 ? a : 3;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) add_ss[sl] += (sl == (int)slot) ? sq : 0.f;
  }
  if (ad.sign >= 0) {
#pragma unroll
    for (int i = 0; i < CH; ++i) u[i] = addq(u[i], zq_from_centered(av[i], q), q);
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) u[i] = subq(u[i], zq_from_centered(av[i], q), q);
  }
}

// plain additions in 32-bit arithmetic mod q, then centre and store / zero test; RZK_EPI_CHUNK coefficients
// per lane at a time.  Checked additions also evaluate the fused norm predicate.
template <int LOGN, class TM = WaveTeam>
__device__ __forceinline__ void row_epilogue(const Program* __restrict__ prog, const Row row, const Operands& ops,
                                             uint32_t b, uint32_t bo, int lane, bool has_terms, int np,
                                             const uint32_t* st_lds, const DevTables& T, uint8_t* __restrict__ flags,
                                             const uint32_t* __restrict__ st_sh = nullptr) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  int nz = 0;
  constexpr int CH = RZK_EPI_CHUNK < E ? RZK_EPI_CHUNK : E;
  const uint32_t q = T.crt.q;
  const bool trusted = ops.trusted != 0;
  float add_ss[4] = {0.f, 0.f, 0.f, 0.f};   // per-lane partial sums of squares of checked additions (slot = add index)
  uint32_t in_bad = 0, in_mx = 0;           // canonical-input test of the additions' coefficients
#pragma unroll
  for (int e0 = 0; e0 < E; e0 += CH) {
    uint32_t u[CH];
    if (has_terms) {
#pragma unroll
      for (int i = 0; i < CH; ++i) u[i] = crt_finish_zq(st_lds[G::j_p1(lane, e0 + i)], np, T.crt);
    } else {
#pragma unroll
      for (int i = 0; i < CH; ++i) u[i] = 0;
    }
    if (st_sh) {   // sum of the row's shift terms, left by the same lanes
#pragma unroll
      for (int i = 0; i < CH; ++i) u[i] = addq(u[i], st_sh[G::j_p1(lane, e0 + i)], q);
    }
#pragma unroll 1
    for (uint32_t a = 0; a < row.nadds; ++a) {
      const AddTerm ad = table_load(&prog->adds[row.add0 + a]);
      add_chunk<LOGN, CH, TM>(u, ad, a, operand_ptr(ops, ad.op & ADD_OP_MASK, ad.off, b, bo, N), lane, e0, q, T.crt.qhalf, trusted,
                          in_bad, in_mx, add_ss);
    }
    if (row.mode == MODE_STORE) {
      int64_t* __restrict__ dst = const_cast<int64_t*>(operand_ptr(ops, row.out_op, row.out_off, b, bo, N));
#pragma unroll
      for (int i = 0; i < CH; ++i) st_stream(dst + G::j_p1(lane, e0 + i), center_from_zq(u[i], T.crt));
    } else {
#pragma unroll
      for (int i = 0; i < CH; ++i) nz |= (u[i] != 0);
    }
  }
  if (row.mode != MODE_STORE) {
    if (__any(nz) && (lane & 63) == 0) flags[bo] = 0;
  }
  if (row.nadds && !trusted && canon_fail(in_bad, in_mx, T.crt.qhalf)) input_fault(ops, flags, bo, lane);
  if (ops.norm_limit) checked_add_verdicts<LOGN, TM>(prog, row, ops, b, bo, lane, add_ss, flags);
}

__device__ __forceinline__ int primes_for(float fbound, const DevTables& T) {
  // |exact result| <= bound: the smallest prime count whose range covers it
  const double bound = (double)fbound * (1.0 + 0x1p-12);   // float sums of up to kMaxTerms rounded products: stay on the safe side
  const int np = bound <= T.cap[1] ? 1 : (bound <= T.cap[2] ? 2 : 3);
  return __builtin_amdgcn_readfirstlane(np);
}

// =============================================================================================
// unit_kernel: the default evaluation of a row program (rzk_dev.h, "wave programs").
//
// One wavefront evaluates the units of one batch entry.  Per auxiliary prime it walks the unit's items: load the
// operand (the first prime pass also proves that every coefficient is canonical and measures the norms that fix the
// number of primes), lift, forward transform, multiply into the rows' accumulators; then inverse transform, fold
// into the Garner state, and after the last prime finish the row (rotation terms, plain additions, store or zero
// test, norm marks).  Register discipline: while an operand is loaded and transformed NOTHING else is live —
//   * the accumulator of row A is parked in LDS (buffer P, N words, key layout: the lane's own 16-byte slots, so
//     no cross-lane synchronisation) and only materialises in registers with the unit's last item, in place of
//     the transform it is computed from;
//   * the accumulator of a pair's row B is born with that last item and parked in P while row A is transformed back;
//   * the Garner state (one or two words per coefficient and row) lives in a per-wave global scratch line
//     ([g][lane][4] order, 16-byte accesses; L2 / Infinity-Cache resident), not in registers or LDS;
//   * rotation terms run after the transforms, accumulating straight into the row's value in registers, with the
//     2N-word image in the (then idle) slab + P.
// LDS per wavefront: transposition slab (N + N/32 words) + P (N words) = 8.1 KiB at N = 1024.
// =============================================================================================
// lines (of N words) of per-wave global scratch the unit / short kernels address
constexpr int kScratchLines = 6;
#ifndef RZK_UNIT_MIN_WAVES
#define RZK_UNIT_MIN_WAVES 1
#endif
#ifndef RZK_STAMPS
#define RZK_STAMPS 0
#endif
#ifndef RZK_FAIR_PRIO
#define RZK_FAIR_PRIO 1
#endif
#if RZK_STAMPS   // section timers of the diagnostic build: wall cycles a wave spends per kind of step
#define RZK_T0() const uint64_t t_sec0 = __builtin_amdgcn_s_memtime()
#define RZK_T1(acc) acc += __builtin_amdgcn_s_memtime() - t_sec0
#else
#define RZK_T0() do { } while (0)
#define RZK_T1(acc) do { } while (0)
#endif
#ifndef RZK_UNIT_OPAQUE
#define RZK_UNIT_OPAQUE 1   // opaque lane ids in unit_kernel: stops hoisting of lane-dependent addresses (91 vs 137 VGPRs at N = 1024)
#endif

// Fair progress among the wavefronts that share a SIMD.  The VALU arbiter serves the highest priority first and,
// among equals, the OLDEST wave: left alone, the four waves of a SIMD finish one after the other (measured at
// N = 1024, one proof per wave: 84 / 105 / 128 / 143 us) and the last one runs its tail alone, with nothing to hide
// its memory latency behind.  Each wave therefore lowers its priority as it advances through its share of the launch
// (quarter by quarter: s_setprio has four levels), so that laggards are served first and all waves stay resident
// until the end.  Speed only: priorities never affect results.
__device__ __forceinline__ void set_priority_level(uint32_t level) {   // 0 = most urgent
  if (level == 0) __builtin_amdgcn_s_setprio(3);
  else if (level == 1) __builtin_amdgcn_s_setprio(2);
  else if (level == 2) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ void set_progress_priority(uint32_t done, uint32_t total) {
#if RZK_FAIR_PRIO
  set_priority_level(__builtin_amdgcn_readfirstlane(total ? (done * 4u) / total : 0u));
#else
  (void)done, (void)total;
#endif
}
// x (transform, phase-3 register order) times `mul` (a resident key entry or a second transform, in registers), into
// row A's accumulator.  init: nothing accumulated yet; to_regs: the unit's last item -> the sum replaces x, else -> P
template <int LOGN, bool to_regs, bool MINUS, class TM = WaveTeam>
__device__ __forceinline__ void mac_park_signed(uint32_t* x, const uint32_t* mul, uint4* P4, int lane, bool init,
                                                const PrimeConsts& pc) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  if (init && !MINUS) {   // first product of a sum: the lazy product IS the sum ([0,2p)), no add and no conditional subtract
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      uint32_t as[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) as[i] = mont_lazy(x[4 * g + i], mul[4 * g + i], pc.p, pc.npinv);
      if (to_regs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[4 * g + i] = as[i];
      } else {
        P4[G::own4(lane, g)] = make_uint4(as[0], as[1], as[2], as[3]);
      }
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < E / 4; ++g) {
    uint4 a = make_uint4(0, 0, 0, 0);
    if (!init) a = P4[G::own4(lane, g)];
    uint32_t as[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      as[i] = MINUS ? mac_sub(as[i], x[4 * g + i], mul[4 * g + i], pc) : mac_add(as[i], x[4 * g + i], mul[4 * g + i], pc);
    if (to_regs) {
#pragma unroll
      for (int i = 0; i < 4; ++i) x[4 * g + i] = as[i];
    } else {
      P4[G::own4(lane, g)] = make_uint4(as[0], as[1], as[2], as[3]);
    }
  }
}
// (the sign is tested once, outside the element loops: a per-element select of mac_add / mac_sub made the compiler
// branch per coefficient; `mul` must be a register array of the caller, never a pointer chosen at run time, or both
// candidates end up in scratch memory)
template <int LOGN, bool to_regs, class TM = WaveTeam>
__device__ __forceinline__ void mac_park(uint32_t* x, const uint32_t* mul, bool minus, uint4* P4, int lane, bool init,
                                         const PrimeConsts& pc) {
  if (minus) mac_park_signed<LOGN, to_regs, true, TM>(x, mul, P4, lane, init, pc);
  else mac_park_signed<LOGN, to_regs, false, TM>(x, mul, P4, lane, init, pc);
}

// Inverse transform of a finished accumulator and Garner step `pi` of `np` against the row's global state lines.
// Returns true when the row's value is complete: acc[e] then holds X mod q in [0,q) for coefficient e*64 + lane.
template <int LOGN, bool OPQ, class TM = WaveTeam>
__device__ __forceinline__ bool inverse_fold_global(int pi, int np, uint32_t* acc, int lane, uint32_t* lds,
                                                    const uint32_t* __restrict__ twi, const PrimeConsts& pc,
                                                    uint32_t* __restrict__ stA, uint32_t* __restrict__ stB,
                                                    const DevTables& T) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  int li = lane;
  RZK_OPAQUE(li);
  uint4* __restrict__ A4 = reinterpret_cast<uint4*>(stA);
  uint4* __restrict__ B4 = reinterpret_cast<uint4*>(stB);
  // the state words this step needs are requested before the transform, which hides their latency
  // (N <= 1024; at N = 2048 a lane holds 32 coefficients and the registers are not there)
  constexpr bool EARLY = E <= 16;
  uint4 sa[E / 4], sb[E / 4];
  if (EARLY && pi >= 1) {
#pragma unroll
    for (int g = 0; g < E / 4; ++g) sa[g] = A4[G::own4(li, g)];
  }
  if (EARLY && pi == 2) {
#pragma unroll
    for (int g = 0; g < E / 4; ++g) sb[g] = B4[G::own4(li, g)];
  }
  wave_inv<LOGN, TM>(acc, li, lds, twi, pc);
  if (!EARLY && pi >= 1) {
#pragma unroll
    for (int g = 0; g < E / 4; ++g) sa[g] = A4[G::own4(li, g)];
  }
  if (!EARLY && pi == 2) {
#pragma unroll
    for (int g = 0; g < E / 4; ++g) sb[g] = B4[G::own4(li, g)];
  }
  if (pi == 0) {
    if (np == 1) {
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = crt1_zq(acc[e], T.pc, T.crt);
      return true;
    }
    if (np == 2) {   // sign-test form (rzk_core.h): the first digit is the canonical residue itself
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = crt2_digit0(acc[e], T.pc);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = crt_fold0(acc[e], np, T.pc, T.crt);
    }
#pragma unroll
    for (int g = 0; g < E / 4; ++g) A4[G::own4(li, g)] = make_uint4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    return false;
  }
  if (pi == 1 && np == 2) {
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      const uint4 dv = sa[g];
      const uint32_t d0[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[4 * g + i] = crt2_zq(acc[4 * g + i], d0[i], T.pc, T.crt);
    }
    return true;
  }
  if (pi == 1) {
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      const uint4 dv = sa[g];
      const uint32_t d0[4] = {dv.x, dv.y, dv.z, dv.w};
      uint32_t va[4], vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t d1 = crt_digit1(acc[4 * g + i], d0[i], np, T.pc, T.crt);
        va[i] = crt_value01_modq(d0[i], d1, T.crt);
        vb[i] = crt_value01_modp2(d0[i], d1, T.pc, T.crt);
      }
      A4[G::own4(li, g)] = make_uint4(va[0], va[1], va[2], va[3]);
      B4[G::own4(li, g)] = make_uint4(vb[0], vb[1], vb[2], vb[3]);
    }
    return false;
  }
#pragma unroll
  for (int g = 0; g < E / 4; ++g) {
    const uint4 av = sa[g], bv = sb[g];
    uint32_t a[4] = {av.x, av.y, av.z, av.w};
    const uint32_t bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      crt_fold2(acc[4 * g + i], T.pc, T.crt, a[i], bb[i]);
      acc[4 * g + i] = crt_finish_zq(a[i], 3, T.crt);
    }
  }
  return true;
}

// u[e] = the row's product sum mod q (coefficient e*64 + lane; zero when the row has no products): adds the sum of
// the row's rotation terms (st_sh, left in the wave's scratch line by the same lanes), the plain additions, then
// store / zero test, norm marks of checked additions, canonical-input test of everything loaded.
template <int LOGN, int CHMAX = 16, class TM = WaveTeam>
__device__ __forceinline__ void finish_row(uint32_t* u, const Program* __restrict__ prog, const Row row,
                                           const Operands& ops, uint32_t b, uint32_t bo, int lane, const DevTables& T,
                                           uint8_t* __restrict__ flags, const uint32_t* __restrict__ st_sh) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  const uint32_t q = T.crt.q, qhalf = T.crt.qhalf;
  const bool trusted = ops.trusted != 0;
  if (st_sh) {
#pragma unroll
    for (int e = 0; e < E; ++e) u[e] = addq(u[e], st_sh[G::j_p1(lane, e)], q);
  }
  // An addition is loaded with up to 16 of a lane's coefficients in flight (a row's wall time is dominated by how
  // often it waits for HBM; 16 sixty-four-bit values are what the register budget of 4 waves per SIMD leaves room for).
  constexpr int CH = E < CHMAX ? E : CHMAX;
  float add_ss[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t in_bad = 0, in_mx = 0;
  int nz = 0;
#pragma unroll
  for (int e0 = 0; e0 < E; e0 += CH) {
#pragma unroll 1
    for (uint32_t a = 0; a < row.nadds; ++a) {
      const AddTerm ad = table_load(&prog->adds[row.add0 + a]);
      add_chunk<LOGN, CH, TM>(u + e0, ad, a, operand_ptr(ops, ad.op & ADD_OP_MASK, ad.off, b, bo, N), lane, e0, q, qhalf, trusted,
                          in_bad, in_mx, add_ss);
    }
    if (row.mode == MODE_STORE) {
      int64_t* __restrict__ dst = const_cast<int64_t*>(operand_ptr(ops, row.out_op, row.out_off, b, bo, N));
#pragma unroll
      for (int i = 0; i < CH; ++i) st_stream(dst + G::j_p1(lane, e0 + i), center_from_zq(u[e0 + i], T.crt));
    } else {
#pragma unroll
      for (int i = 0; i < CH; ++i) nz |= (u[e0 + i] != 0);
    }
  }
  if (row.mode != MODE_STORE) {
    if (__any(nz) && (lane & 63) == 0) flags[bo] = 0;
  }
  if (row.nadds && !trusted && canon_fail(in_bad, in_mx, qhalf)) input_fault(ops, flags, bo, lane);
  if (ops.norm_limit) checked_add_verdicts<LOGN, TM>(prog, row, ops, b, bo, lane, add_ss, flags);
}

// Workgroups: four independent one-wavefront teams (16-wave workgroups whose SIMD mates ranked each other through an LDS
// table for exact fairness measured slower in round 2 — verify rows 90 vs 81 us — and were removed), or ONE
// two-wavefront team (PairTeam, N = 2048).  Teams of two are compiled for 4 waves per SIMD (<= 128 VGPRs: 16
// coefficients per thread, the budget of the N = 1024 kernels).
template <int LOGN, bool HAS_VEC, bool HAS_SHIFT, class TM = WaveTeam>
__global__ void __launch_bounds__(TM::kTeamsPerBlock << TM::LL, ((LOGN <= 10 && HAS_VEC) || TM::LL == 7 ? 4 : RZK_UNIT_MIN_WAVES))   // vector x vector variants: hold the 4 waves per SIMD the LDS allows
unit_kernel(const Program* __restrict__ prog, const WaveProgram* __restrict__ wp, const Operands ops,
            const uint32_t* __restrict__ key_ntt, const double* __restrict__ key_l2, const DevTables* __restrict__ Tp,
            const uint32_t* __restrict__ tw_all, uint32_t* __restrict__ scratch, uint8_t* __restrict__ flags,
            const uint32_t ntasks, const uint32_t units_per_task, const uint32_t tasks_per_entry,
            const uint32_t work_per_task) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  constexpr bool OPQ = RZK_UNIT_OPAQUE || LOGN >= RZK_OPAQUE_LANE_MIN_LOGN;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & (G::LANES - 1);                                         // index inside the team
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> TM::LL);           // team of the workgroup
  constexpr int WPB = TM::kTeamsPerBlock;
  uint32_t* lds = smem + wave * (G::LDS_WORDS + N);             // transposition slab, then P
  uint4* P4 = reinterpret_cast<uint4*>(lds + G::LDS_WORDS);     // G::LDS_WORDS * 4 is a multiple of 16 bytes
  // per-wave global scratch: Garner words [row A | B][word A | B][N], then the sum of row A's rotation terms
  uint32_t* st = scratch + ((size_t)blockIdx.x * WPB + wave) * (size_t)(kScratchLines * N + 16);
  uint32_t* st_sh = st + 4 * N;
#if RZK_STAMPS   // diagnostic build only (tools/wave_timeline.py): when each wavefront ran and where
  const uint64_t stamp0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t cyc0 = __builtin_amdgcn_s_memtime();
  uint64_t t_load = 0, t_fwd = 0, t_mac = 0, t_inv = 0, t_fin = 0, t_rot = 0;
#endif
  const DevTables& T = *Tp;
  const uint32_t qhalf = T.crt.qhalf;
  const bool trusted = ops.trusted != 0;
  const uint32_t nunits = wp->nunits;

  // progress of this wave through its share of the launch, in transforms (work_per_task: the host's estimate)
#ifndef RZK_TOUCH_NEXT
#define RZK_TOUCH_NEXT 0   // experiment: L2 touch of the next item's operand ahead of the current transform
#endif
#ifndef RZK_UNIT_ROT
#define RZK_UNIT_ROT 0   // (rotating the unit order per workgroup measured no gain)
#endif
  const uint32_t rot_sel = RZK_UNIT_ROT == 1 ? (blockIdx.x >> 8) : (RZK_UNIT_ROT == 2 ? blockIdx.x : 0u);
  const uint32_t first_task = blockIdx.x * WPB + wave;
  const uint32_t my_tasks = first_task < ntasks ? (ntasks - first_task + gridDim.x * WPB - 1) / (gridDim.x * WPB) : 0;
  const uint32_t work_total = my_tasks * work_per_task;
  uint32_t work_done = 0;
#define RZK_STEP_PRIORITY()                             \
  do {                                                  \
    set_progress_priority(work_done, work_total);       \
    ++work_done;                                        \
  } while (0)

  for (uint32_t task = first_task; task < ntasks; task += gridDim.x * WPB) {
    const uint32_t b = task / tasks_per_entry;
    const uint32_t u0 = (task - b * tasks_per_entry) * units_per_task;
    const uint32_t u1 = u0 + units_per_task < nunits ? u0 + units_per_task : nunits;
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    if (ops.preset) {   // this team evaluates every row of the entry (host: one task per entry, one flag per entry): it owns the flag
      if (lane == 0) flags[bo] = (uint8_t)ops.preset;
      if (TM::LL != 6) TM::sync();   // the other wavefront of a pair may clear it
    }
    const uint32_t unit_rot = (u1 - u0) > 1 ? rot_sel % (u1 - u0) : 0u;
#pragma unroll 1
    for (uint32_t uk = u0; uk < u1; ++uk) {
      // waves start their task at different units (by workgroup), so that SIMD mates are not all in the same kind of
      // step (load / transform / store) at the same time; the units of an entry are independent of each other
      uint32_t ui = uk + unit_rot;
      if (ui >= u1) ui -= u1 - u0;
      const Unit un = table_load(&wp->units[ui]);
      const Row rowA = table_load(&prog->rows[un.rowA]);
      const bool pair = un.rowB != kNoRow;
      const uint32_t un_items = un.nitems;
      const bool null_unit = un_items == 0;   // no products: additions / rotation terms only
      const bool has_shift = HAS_SHIFT && rowA.nshift > 0;
      if (has_shift) {
        // challenge products first (rotations, image in slab + P); their sum mod q is built in the wave's scratch line
        // (every lane reads and writes only its own coefficients) and waits there for finish_row
        bool fault = false;
        RZK_T0();
#pragma unroll 1
        for (uint32_t t = 0; t < rowA.nshift; ++t) {
          const Term tm = table_load(&prog->terms[rowA.term0 + rowA.nterms + t]);
          const int64_t* __restrict__ pa = operand_ptr(ops, tm.a_op, tm.a_off, b, bo, N);
          int32_t a[E];
          if (trusted) {
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = (int32_t)pa[G::j_p1(lane, e)];
          } else {
            uint32_t abad = 0, amx = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = canon_lo_mx(pa[G::j_p1(lane, e)], qhalf, abad, amx);
            fault = fault || canon_fail(abad, amx, qhalf);
          }
          shift_product<LOGN, false, true, TM>(st_sh, t == 0, tm.sign < 0, a, operand_ptr(ops, tm.b_op, tm.b_off, b, bo, N), lane,
                                           reinterpret_cast<int32_t*>(lds), T, fault, trusted);
        }
        if (fault) input_fault(ops, flags, bo, lane);
        TM::sync();   // the image is dead: slab and P may be overwritten
        RZK_T1(t_rot);
      }
      int np = null_unit ? 1 : kMaxPrimes;
      const uint32_t nit = null_unit ? 1u : un_items;
      float boundA = 0.f, boundB = 0.f;
#pragma unroll 1
      for (int pi = 0; pi < np; ++pi) {
        const PrimeConsts pc = T.pc[pi];
        const uint32_t* __restrict__ twf = tw_all + (size_t)(2 * pi) * kTableLen;
        const bool first = pi == 0;
        bool fault = false;
#pragma unroll 1
        for (uint32_t it = 0; it < nit; ++it) {
          // every array below is local to one trip: nothing is carried in registers from item to item
          const bool last = it + 1 == nit;
          RZK_STEP_PRIORITY();
          int ln = lane;
          RZK_OPAQUE(ln);
          uint32_t acc[E];    // with the last item: row A's sum
#pragma unroll
          for (int c = 0; c < E; ++c) acc[c] = 0;
          if (!null_unit) {
            const Item im = table_load(&wp->items[un.item0 + it]);
            uint32_t x[E];      // the current transform
            float nb = 0.f;
            bool below = true;
            const bool chk = first && (im.flags & (TERM_CHECK | TERM_CHECK2));
            {
              RZK_T0();
#ifdef RZK_EXPERIMENT_REREAD   // diagnostic builds only (wrong results): what the launch would cost if the operand reads
                               // after the first prime's came from L2 — 1: units of two rows only, 2: every unit
              const uint32_t b_ld = (!first && (RZK_EXPERIMENT_REREAD >= 2 || pair)) ? (b & 63u) : b;
#else
              const uint32_t b_ld = b;
#endif
              load_lift<LOGN, TM>(x, operand_ptr(ops, im.b_op, im.b_off, b_ld, bo, N), ln, pc, first, nb, chk, ops.norm_limit, below, qhalf,
                              trusted, fault);
              RZK_T1(t_load);
            }
            if (chk && !below && (lane & 63) == 0) fail_check(flags + bo, ops.pad != 0, (im.flags & TERM_CHECK2) != 0);
            const bool vec = HAS_VEC && im.kind == ITEM_VEC;
            // the resident key entry of row A's product is requested before the transform, which hides its latency
            // (N <= 1024; at N = 2048 the registers are not there and the entry is loaded after the transform)
            constexpr bool EARLY = E <= 16 && !HAS_VEC;   // (and not next to vector x vector items: their second transform needs the registers)
            uint32_t kreg[E];
            const uint4* __restrict__ kpA = reinterpret_cast<const uint4*>(key_ntt + ((size_t)im.keyA * kKeyImages + pi) * N);
            if (EARLY && !vec && im.keyA != kNoKey) {
#pragma unroll
              for (int g = 0; g < E / 4; ++g) {
                const uint4 kv = kpA[G::key4(ln, g)];
                kreg[4 * g] = kv.x, kreg[4 * g + 1] = kv.y, kreg[4 * g + 2] = kv.z, kreg[4 * g + 3] = kv.w;
              }
            }
#if RZK_TOUCH_NEXT
            // one dword per 128-byte line of the NEXT item's operand, requested behind the key entry (the memory counter
            // is in order: the key can be consumed while this is still in flight) and consumed after the product below:
            // the line fetches from HBM then overlap this item's transform
            uint32_t touch = 0;
            if (first && !last) {
              const Item nx = table_load(&wp->items[un.item0 + it + 1]);
              const uint32_t* __restrict__ tp = reinterpret_cast<const uint32_t*>(operand_ptr(ops, nx.b_op, nx.b_off, b, bo, N));
              touch = tp[(size_t)(ln & (N / 16 - 1)) * 32];
            }
#endif
            {
              RZK_T0();
              wave_fwd<LOGN, TM>(x, ln, lds, twf, pc);
              RZK_T1(t_fwd);
            }
            if (!EARLY && !vec && im.keyA != kNoKey) {
#pragma unroll
              for (int g = 0; g < E / 4; ++g) {
                const uint4 kv = kpA[G::key4(ln, g)];
                kreg[4 * g] = kv.x, kreg[4 * g + 1] = kv.y, kreg[4 * g + 2] = kv.z, kreg[4 * g + 3] = kv.w;
              }
            }
            uint32_t xb[E];   // ITEM_VEC: b's transform with N^-1 and the Montgomery factor folded in
            if (vec) {
#pragma unroll
              for (int c = 0; c < E; ++c) xb[c] = csub(mont_lazy(x[c], pc.ninv_r2, pc.p, pc.npinv), pc.p);
              float na = 0.f;
              bool unused_below = true;
              load_lift<LOGN, TM>(x, operand_ptr(ops, im.a_op, im.a_off, b, bo, N), ln, pc, first, na, false, 0, unused_below, qhalf,
                              trusted, fault);
              wave_fwd<LOGN, TM>(x, ln, lds, twf, pc);
              if (first) boundA = bound_fma(na, nb, boundA);
            } else if (first) {
              if (im.keyA != kNoKey) boundA = bound_fma((float)key_l2[im.keyA], nb, boundA);
              if (pair && im.keyB != kNoKey) boundB = bound_fma((float)key_l2[im.keyB], nb, boundB);
            }
            const bool feedsA = vec || im.keyA != kNoKey;
            if (!last) {
              RZK_T0();
              if (vec) mac_park<LOGN, false, TM>(x, xb, im.signA < 0, P4, ln, it == 0, pc);
              else if (feedsA) mac_park<LOGN, false, TM>(x, kreg, im.signA < 0, P4, ln, it == 0, pc);
              RZK_T1(t_mac);
#if RZK_TOUCH_NEXT
              asm volatile("" ::"v"(touch));
#endif
              continue;
            }
            RZK_T0();
            // ---- last item: row A's sum leaves P and materialises in registers ...
            if (fault) input_fault(ops, flags, bo, lane);
            if (first) np = primes_for(boundA > boundB ? boundA : boundB, T);
#pragma unroll
            for (int c = 0; c < E; ++c) acc[c] = x[c];
            if (vec) {
              mac_park<LOGN, true, TM>(acc, xb, im.signA < 0, P4, ln, it == 0, pc);
            } else if (feedsA) {
              mac_park<LOGN, true, TM>(acc, kreg, im.signA < 0, P4, ln, it == 0, pc);
            } else {   // (an item that only feeds row B)
#pragma unroll
              for (int g = 0; g < E / 4; ++g) {
                const uint4 v = P4[G::own4(ln, g)];
                acc[4 * g] = v.x, acc[4 * g + 1] = v.y, acc[4 * g + 2] = v.z, acc[4 * g + 3] = v.w;
              }
            }
            if (pair) {   // ... and row B's only product, from the same transform, takes its place in P
              const uint4* __restrict__ kb = reinterpret_cast<const uint4*>(key_ntt + ((size_t)im.keyB * kKeyImages + pi) * N);
              uint32_t kbr[E];
#pragma unroll
              for (int g = 0; g < E / 4; ++g) {
                const uint4 kv = kb[G::key4(ln, g)];
                kbr[4 * g] = kv.x, kbr[4 * g + 1] = kv.y, kbr[4 * g + 2] = kv.z, kbr[4 * g + 3] = kv.w;
              }
              mac_park<LOGN, false, TM>(x, kbr, im.signB < 0, P4, ln, true, pc);
            }
            RZK_T1(t_mac);
          }
          // ---- transform back, fold, and after the last prime finish the row(s) of the unit
#pragma unroll 1
          for (uint32_t r = 0; r < (pair ? 2u : 1u); ++r) {
            if (r == 1) {
#pragma unroll
              for (int g = 0; g < E / 4; ++g) {
                const uint4 v = P4[G::own4(ln, g)];
                acc[4 * g] = v.x, acc[4 * g + 1] = v.y, acc[4 * g + 2] = v.z, acc[4 * g + 3] = v.w;
              }
            }
            bool done = true;
            RZK_STEP_PRIORITY();
            if (!null_unit) {
              RZK_T0();
              done = inverse_fold_global<LOGN, OPQ, TM>(pi, np, acc, lane, lds, twf + kTableLen, pc, st + (size_t)(2 * r) * N,
                                                    st + (size_t)(2 * r + 1) * N, T);
              RZK_T1(t_inv);
            }
            if (done) {
              RZK_T0();
              finish_row<LOGN, 16, TM>(acc, prog, table_load(&prog->rows[r ? un.rowB : un.rowA]), ops, b, bo, lane, T, flags,
                               (has_shift && r == 0) ? st_sh : nullptr);
              RZK_T1(t_fin);
            }
          }
        }
      }
    }
  }
#undef RZK_STEP_PRIORITY
#if RZK_STAMPS
  if (lane == 0) {
    const uint64_t stamp1 = __builtin_amdgcn_s_memrealtime();
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t* o = st + kScratchLines * N;
    o[0] = (uint32_t)stamp0, o[1] = (uint32_t)(stamp0 >> 32), o[2] = (uint32_t)stamp1, o[3] = (uint32_t)(stamp1 >> 32);
    o[4] = hwid, o[5] = xcc, o[6] = blockIdx.x, o[7] = wave;
    const uint64_t cyc1 = __builtin_amdgcn_s_memtime();
    o[8] = (uint32_t)(cyc1 - cyc0);   // shader-clock cycles of the wave's lifetime
    o[9] = (uint32_t)t_load, o[10] = (uint32_t)t_fwd, o[11] = (uint32_t)t_mac, o[12] = (uint32_t)t_inv, o[13] = (uint32_t)t_fin, o[14] = (uint32_t)t_rot;
  }
#endif
}

// =============================================================================================
// unit_io_kernel ("item outer"): the default evaluation of KEY-PRODUCT programs — every operand is read from HBM ONCE.
//
// Why: the round-3 experiment of DESIGN.md §6 (the same launches with every operand L2-resident: commit 130 -> 100 us,
// verify 76 -> 65 us) showed that unit_kernel's launches are co-bound by HBM traffic: 1.8 x the algorithmic bytes, because
// its prime-outer loop re-reads every operand for the second prime and keeps the Garner state of a row in global lines
// across a whole prime pass (evicted long before it is read back).  Here the loops are swapped:
//   for every item (operand): load it once — canonical test, norm measurement, norm mark — keep the low words in
//     registers, and for primes 0 and 1: lift, forward transform, multiply into that prime's sum of every row it feeds;
//   the sums wait in parking spots between items: row A / prime 0 in LDS (buffer P), the others (row A / prime 1, a
//     pair's row B) in the team's scratch lines in global memory, which are re-used within microseconds and stay in L2;
//   then per row: inverse transform of prime 0, first digit in REGISTERS, inverse transform of prime 1, sign-test
//     reconstruction (crt2_zq), finish_row.  No Garner state ever leaves the registers.
// Two primes are computed for every row (a row that one prime would cover is still exact with two).  Rows that need
// the third prime (full-range operands: Mat::dot on arbitrary vectors, tests) are detected once all operands have been
// measured and take one more pass over the items for prime 2 (operands re-read: the rare path), with the offset-form
// Garner steps in registers.  Everything else — units, pairs, rotation terms first, finish_row, norm marks, input
// faults, progress priorities, teams of one or two wavefronts — is unit_kernel's.
// Parking lines of a team (N words each): 0 = A/p1, 1 = B/p0, 2 = B/p1, 3 = A/p2, 4 = rotation sums, 5 = B/p2.
// =============================================================================================
template <int LOGN, class TM = WaveTeam>
__device__ __forceinline__ void load_measure(int32_t* v, const int64_t* __restrict__ src, int lane, bool measure, float& nrm2,
                                             bool check, uint64_t limit, bool& below, uint32_t qhalf, bool trusted, bool& fault) {
  using G = Geo<LOGN, TM::LL>;
  if (!measure || trusted) {
#pragma unroll
    for (int e = 0; e < G::E; ++e) v[e] = (int32_t)src[G::j_p1(lane, e)];
  } else {
    uint32_t bad = 0, mx = 0;
#pragma unroll
    for (int e = 0; e < G::E; ++e) v[e] = canon_lo_mx(src[G::j_p1(lane, e)], qhalf, bad, mx);
    fault = fault || canon_fail(bad, mx, qhalf);
  }
  if (measure) {
    const float ss = TM::sum_f32(lane_sum_sq_f32<G::E>(v));
    nrm2 = norm2_upper(ss);
    if (check) below = norm_below<G::E, TM>(v, ss, limit);
  }
}
// a parked sum (16-byte slots of the team's own threads) -> registers
template <int LOGN, class TM, class P4T>
__device__ __forceinline__ void unpark(uint32_t* a, P4T P4, int lane) {
  using G = Geo<LOGN, TM::LL>;
#pragma unroll
  for (int g = 0; g < G::E / 4; ++g) {
    const uint4 v = P4[G::own4(lane, g)];
    a[4 * g] = v.x, a[4 * g + 1] = v.y, a[4 * g + 2] = v.z, a[4 * g + 3] = v.w;
  }
}

// Where the sums park between items (LDS budget: 10 KiB per wavefront at 4 waves per SIMD):
//   N = 512   slab 2.1 KiB + A/p0, A/p1, B/p0 (2 KiB each) = 8.1 KiB: only B/p1 and the third-prime sums use global
//             lines.  This is the default kernel of key-product programs at N = 512 (Open cycle 27.1 -> 29.6 M proofs/s).
//   N >= 1024 A/p0 in LDS, everything else in global lines: slower than unit_kernel (Open N = 1024: commit 145 vs 129 us;
//             a variant with a half-size transposition slab and both A sums in LDS: 139-146 us — the two-round
//             transpositions need ~127 VGPRs before any key entry can be requested ahead of a transform, see
//             DESIGN.md §6), so it is reachable only through RZK_UNIT_IO=1 (tests).
#ifndef RZK_IO_B0_LDS
#define RZK_IO_B0_LDS 1
#endif
#ifndef RZK_IO_MIN_WAVES
#define RZK_IO_MIN_WAVES (TM::LL == 7 || LOGN == 10 ? 4 : 1)   // 16 coefficients per thread: 4 waves per SIMD
#endif
template <int LOGN, int LL>
struct IoCfg {
  static constexpr bool P1_FULL = LOGN == 9;                      // A / prime 1 in an LDS buffer
  static constexpr bool B0_LDS = LOGN == 9 && RZK_IO_B0_LDS;      // B / prime 0 in an LDS buffer
  static constexpr int N = 1 << LOGN;
  static constexpr int SLAB = Geo<LOGN, LL>::LDS_WORDS;
  static constexpr int OFF_P1 = SLAB + N;
  static constexpr int OFF_B0 = OFF_P1 + (P1_FULL ? N : 0);
  static constexpr int WORDS = OFF_B0 + (B0_LDS ? N : 0);         // LDS words per team
};
template <int LOGN, bool HAS_SHIFT, class TM = WaveTeam>
__global__ void __launch_bounds__(TM::kTeamsPerBlock << TM::LL, RZK_IO_MIN_WAVES)
unit_io_kernel(const Program* __restrict__ prog, const WaveProgram* __restrict__ wp, const Operands ops,
               const uint32_t* __restrict__ key_ntt, const double* __restrict__ key_l2, const DevTables* __restrict__ Tp,
               const uint32_t* __restrict__ tw_all, uint32_t* __restrict__ scratch, uint8_t* __restrict__ flags,
               const uint32_t ntasks, const uint32_t units_per_task, const uint32_t tasks_per_entry,
               const uint32_t work_per_task) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  constexpr bool OPQ = true;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & (G::LANES - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> TM::LL);
  constexpr int TPB = TM::kTeamsPerBlock;
  // Where LDS allows (N = 512: 6.2 KiB per wavefront) the prime-1 sum of row A parks in a second LDS buffer instead of
  // a global line (IO_P1_LDS)
  using IO = IoCfg<LOGN, TM::LL>;
  constexpr bool P1L = IO::P1_FULL, B0L = IO::B0_LDS;
  uint32_t* lds = smem + wave * IO::WORDS;                                          // transposition slab, then the parking buffers
  uint4* P4 = reinterpret_cast<uint4*>(lds + IO::SLAB);                             // A / prime 0
  uint4* P41 = reinterpret_cast<uint4*>(lds + IO::OFF_P1);                          // A / prime 1   (P1L)
  uint4* PB0 = reinterpret_cast<uint4*>(lds + IO::OFF_B0);                          // B / prime 0   (B0L)
  uint32_t* st = scratch + ((size_t)blockIdx.x * TPB + wave) * (size_t)(kScratchLines * N + 16);
  uint32_t* st_sh = st + 4 * N;
#if RZK_STAMPS
  const uint64_t stamp0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t cyc0 = __builtin_amdgcn_s_memtime();
#endif
  const DevTables& T = *Tp;
  const uint32_t qhalf = T.crt.qhalf;
  const bool trusted = ops.trusted != 0;
  const uint32_t nunits = wp->nunits;
  const uint32_t first_task = blockIdx.x * TPB + wave;
  const uint32_t my_tasks = first_task < ntasks ? (ntasks - first_task + gridDim.x * TPB - 1) / (gridDim.x * TPB) : 0;
  const uint32_t work_total = my_tasks * work_per_task;
  uint32_t work_done = 0;
#define RZK_STEP_PRIORITY()                             \
  do {                                                  \
    set_progress_priority(work_done, work_total);       \
    ++work_done;                                        \
  } while (0)

  for (uint32_t task = first_task; task < ntasks; task += gridDim.x * TPB) {
    const uint32_t b = task / tasks_per_entry;
    const uint32_t u0 = (task - b * tasks_per_entry) * units_per_task;
    const uint32_t u1 = u0 + units_per_task < nunits ? u0 + units_per_task : nunits;
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    if (ops.preset) {   // as in unit_kernel: the team that evaluates the whole entry initialises its flag
      if (lane == 0) flags[bo] = (uint8_t)ops.preset;
      if (TM::LL != 6) TM::sync();
    }
#pragma unroll 1
    for (uint32_t ui = u0; ui < u1; ++ui) {
      const Unit un = table_load(&wp->units[ui]);
      const Row rowA = table_load(&prog->rows[un.rowA]);
      const bool pair = un.rowB != kNoRow;
      const uint32_t nit = un.nitems;
      const bool has_shift = HAS_SHIFT && rowA.nshift > 0;
      if (has_shift) {
        // challenge products first (rotations, image in slab + P); their sum mod q waits in the team's line 4
        bool fault = false;
#pragma unroll 1
        for (uint32_t t = 0; t < rowA.nshift; ++t) {
          const Term tm = table_load(&prog->terms[rowA.term0 + rowA.nterms + t]);
          const int64_t* __restrict__ pa = operand_ptr(ops, tm.a_op, tm.a_off, b, bo, N);
          int32_t a[E];
          if (trusted) {
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = (int32_t)pa[G::j_p1(lane, e)];
          } else {
            uint32_t abad = 0, amx = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = canon_lo_mx(pa[G::j_p1(lane, e)], qhalf, abad, amx);
            fault = fault || canon_fail(abad, amx, qhalf);
          }
          shift_product<LOGN, false, true, TM>(st_sh, t == 0, tm.sign < 0, a, operand_ptr(ops, tm.b_op, tm.b_off, b, bo, N), lane,
                                           reinterpret_cast<int32_t*>(lds), T, fault, trusted);
        }
        if (fault) input_fault(ops, flags, bo, lane);
        TM::sync();   // the image is dead: slab and P may be overwritten
      }
      if (nit == 0) {   // no products: additions / rotation terms only
        uint32_t u[E];
#pragma unroll
        for (int e = 0; e < E; ++e) u[e] = 0;
        RZK_STEP_PRIORITY();
        finish_row<LOGN, 16, TM>(u, prog, rowA, ops, b, bo, lane, T, flags, has_shift ? st_sh : nullptr);
        continue;
      }
      // ---- the items: pass 0 = primes 0 and 1 (operands measured), pass 1 = prime 2, only when the bound asks for it
      float boundA = 0.f, boundB = 0.f;
      bool fault = false, haveA = false;
      int np = 2;
#pragma unroll 1
      for (int pass = 0; pass < (np == 3 ? 2 : 1); ++pass) {
        haveA = false;
#pragma unroll 1
        for (uint32_t it = 0; it < nit; ++it) {
          const bool last = it + 1 == nit;
          int ln = lane;
          RZK_OPAQUE(ln);
          const Item im = table_load(&wp->items[un.item0 + it]);
          const int64_t* __restrict__ src = operand_ptr(ops, im.b_op, im.b_off, b, bo, N);
          const bool chk = pass == 0 && (im.flags & (TERM_CHECK | TERM_CHECK2));
          const bool feedsA = im.keyA != kNoKey;
          const bool feedsB = pair && last && im.keyB != kNoKey;
          const int pi0 = pass == 0 ? 0 : 2, pi1 = pass == 0 ? 2 : 3;
          // The operand's one trip from HBM: canonical test, norm measurement, norm mark (first pass).  With 8
          // coefficients per lane the low words simply stay in registers for the second prime (RETAIN); with 16 or
          // more they are read again — microseconds later, out of L2 — because keeping them through a transform costs
          // the registers that hold the kernel at 4 waves per SIMD.
          constexpr bool RETAIN = E <= 8;
          int32_t vkeep[RETAIN ? E : 1];
          if (RETAIN) {
            float nb = 0.f;
            bool below = true;
            load_measure<LOGN, TM>(vkeep, src, ln, pass == 0, nb, chk, ops.norm_limit, below, qhalf, trusted, fault);
            if (chk && !below && (lane & 63) == 0) fail_check(flags + bo, ops.pad != 0, (im.flags & TERM_CHECK2) != 0);
            if (pass == 0) {
              if (feedsA) boundA = bound_fma((float)key_l2[im.keyA], nb, boundA);
              if (pair && im.keyB != kNoKey) boundB = bound_fma((float)key_l2[im.keyB], nb, boundB);
            }
          }
#pragma unroll 1
          for (int pi = pi0; pi < pi1; ++pi) {
            RZK_STEP_PRIORITY();
            RZK_OPAQUE(ln);   // per transform: lane-dependent addresses must not be hoisted out of this loop (30 VGPRs)
            const PrimeConsts pc = T.pc[pi];
            const uint32_t* __restrict__ twf = tw_all + (size_t)(2 * pi) * kTableLen;
            uint32_t x[E];
            if (RETAIN) {
#pragma unroll
              for (int e = 0; e < E; ++e) x[e] = lift(vkeep[RETAIN ? e : 0], pc);
            } else if (pi == pi0) {
              int32_t v[E];
              float nb = 0.f;
              bool below = true;
              load_measure<LOGN, TM>(v, src, ln, pass == 0, nb, chk, ops.norm_limit, below, qhalf, trusted, fault);
              if (chk && !below && (lane & 63) == 0) fail_check(flags + bo, ops.pad != 0, (im.flags & TERM_CHECK2) != 0);
              if (pass == 0) {
                if (feedsA) boundA = bound_fma((float)key_l2[im.keyA], nb, boundA);
                if (pair && im.keyB != kNoKey) boundB = bound_fma((float)key_l2[im.keyB], nb, boundB);
              }
#pragma unroll
              for (int e = 0; e < E; ++e) x[e] = lift(v[e], pc);
            } else {
              const int32_t* __restrict__ lo32 = reinterpret_cast<const int32_t*>(src);
#pragma unroll
              for (int e = 0; e < E; ++e) x[e] = lift(lo32[2 * G::j_p1(ln, e)], pc);
            }
            // (requesting row A's key entry before the transform, as unit_kernel does, does not pay here:)
            constexpr bool EARLY = false;   // (measured at N = 512: 42.3 us against 40.4 for the verify rows; at N = 1024 the entry
                                            //  would cost 30 VGPRs across the transform)
            const uint4* __restrict__ kpA = reinterpret_cast<const uint4*>(key_ntt + ((size_t)(feedsA ? im.keyA : 0) * kKeyImages + pi) * N);
            uint32_t kreg[E];
            if (EARLY && feedsA) {
              // ... but not ahead of the operand itself: the request is tied to the last lifted coefficient, or the scheduler
              // issues it first and the entry sits in registers next to the 64-bit loads of the operand (+32 VGPRs)
              int lk = ln;
              asm volatile("" : "+v"(lk) : "v"(x[E - 1]));
#pragma unroll
              for (int g = 0; g < E / 4; ++g) {
                const uint4 kv = kpA[G::key4(lk, g)];
                kreg[4 * g] = kv.x, kreg[4 * g + 1] = kv.y, kreg[4 * g + 2] = kv.z, kreg[4 * g + 3] = kv.w;
              }
            }
            wave_fwd<LOGN, TM>(x, ln, lds, twf, pc);
            if (feedsA) {
              if (!EARLY) {
#pragma unroll
                for (int g = 0; g < E / 4; ++g) {
                  const uint4 kv = kpA[G::key4(ln, g)];
                  kreg[4 * g] = kv.x, kreg[4 * g + 1] = kv.y, kreg[4 * g + 2] = kv.z, kreg[4 * g + 3] = kv.w;
                }
              }
              // (parking leaves x untouched: row B's product below is formed from the same transform)
              if (pi == 0) mac_park<LOGN, false, TM>(x, kreg, im.signA < 0, P4, ln, !haveA, pc);
              else if (P1L && pi == 1) mac_park<LOGN, false, TM>(x, kreg, im.signA < 0, P41, ln, !haveA, pc);
              else mac_park<LOGN, false, TM>(x, kreg, im.signA < 0, reinterpret_cast<uint4*>(st + (pi == 1 ? 0 : 3) * N), ln, !haveA, pc);
            }
            if (feedsB) {
              const uint4* __restrict__ kpB = reinterpret_cast<const uint4*>(key_ntt + ((size_t)im.keyB * kKeyImages + pi) * N);
              uint32_t kbr[E];
              int lb = ln;
              asm volatile("" : "+v"(lb));   // row B's entry is requested HERE, not ahead of the transform (16 VGPRs)
#pragma unroll
              for (int g = 0; g < E / 4; ++g) {
                const uint4 kv = kpB[G::key4(lb, g)];
                kbr[4 * g] = kv.x, kbr[4 * g + 1] = kv.y, kbr[4 * g + 2] = kv.z, kbr[4 * g + 3] = kv.w;
              }
              if (B0L && pi == 0) mac_park<LOGN, false, TM>(x, kbr, im.signB < 0, PB0, ln, true, pc);
              else mac_park<LOGN, false, TM>(x, kbr, im.signB < 0, reinterpret_cast<uint4*>(st + (pi == 0 ? 1 : (pi == 1 ? 2 : 5)) * N), ln, true, pc);
            }
          }
          haveA = haveA || feedsA;
        }
        if (pass == 0) {
          if (fault) input_fault(ops, flags, bo, lane);
          np = primes_for(boundA > boundB ? boundA : boundB, T);
          np = np < 2 ? 2 : np;
        }
      }
      // ---- the rows: inverse transforms back to back, reconstruction in registers
#pragma unroll 1
      for (uint32_t r = 0; r < (pair ? 2u : 1u); ++r) {
        int li = lane;
        RZK_OPAQUE(li);
        const bool have = r == 1 || haveA;
        uint32_t u[E];     // the row's value mod q
        if (!have) {
#pragma unroll
          for (int e = 0; e < E; ++e) u[e] = 0;
        } else {
          uint32_t a[E];
          if (r == 0) unpark<LOGN, TM>(a, const_cast<const uint4*>(P4), li);
          else if (B0L) unpark<LOGN, TM>(a, const_cast<const uint4*>(PB0), li);
          else unpark<LOGN, TM>(a, reinterpret_cast<const uint4*>(st + 1 * N), li);
          RZK_STEP_PRIORITY();
          RZK_OPAQUE(li);
          wave_inv<LOGN, TM>(a, li, lds, tw_all + (size_t)(2 * 0 + 1) * kTableLen, T.pc[0]);
          if (np == 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) u[e] = crt2_digit0(a[e], T.pc);
            if (P1L && r == 0) unpark<LOGN, TM>(a, const_cast<const uint4*>(P41), li);
            else unpark<LOGN, TM>(a, reinterpret_cast<const uint4*>(st + (r == 0 ? 0 : 2) * N), li);
            RZK_STEP_PRIORITY();
            RZK_OPAQUE(li);
            wave_inv<LOGN, TM>(a, li, lds, tw_all + (size_t)(2 * 1 + 1) * kTableLen, T.pc[1]);
#pragma unroll
            for (int e = 0; e < E; ++e) u[e] = crt2_zq(a[e], u[e], T.pc, T.crt);
          } else {   // three primes: the offset form, words A and B in registers
            uint32_t wb[E];
#pragma unroll
            for (int e = 0; e < E; ++e) u[e] = crt_fold0(a[e], 3, T.pc, T.crt);
            if (P1L && r == 0) unpark<LOGN, TM>(a, const_cast<const uint4*>(P41), li);
            else unpark<LOGN, TM>(a, reinterpret_cast<const uint4*>(st + (r == 0 ? 0 : 2) * N), li);
            RZK_STEP_PRIORITY();
            RZK_OPAQUE(li);
            wave_inv<LOGN, TM>(a, li, lds, tw_all + (size_t)(2 * 1 + 1) * kTableLen, T.pc[1]);
#pragma unroll
            for (int e = 0; e < E; ++e) {
              wb[e] = 0;
              crt_fold1(a[e], 3, T.pc, T.crt, u[e], wb[e]);
            }
            unpark<LOGN, TM>(a, reinterpret_cast<const uint4*>(st + (r == 0 ? 3 : 5) * N), li);
            RZK_STEP_PRIORITY();
            RZK_OPAQUE(li);
            wave_inv<LOGN, TM>(a, li, lds, tw_all + (size_t)(2 * 2 + 1) * kTableLen, T.pc[2]);
#pragma unroll
            for (int e = 0; e < E; ++e) {
              crt_fold2(a[e], T.pc, T.crt, u[e], wb[e]);
              u[e] = crt_finish_zq(u[e], 3, T.crt);
            }
          }
        }
        finish_row<LOGN, 16, TM>(u, prog, table_load(&prog->rows[r ? un.rowB : un.rowA]), ops, b, bo, lane, T, flags,
                                 (has_shift && r == 0) ? st_sh : nullptr);
      }
    }
  }
#undef RZK_STEP_PRIORITY
#if RZK_STAMPS
  if (lane == 0) {
    const uint64_t stamp1 = __builtin_amdgcn_s_memrealtime();
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t* o = st + kScratchLines * N;
    o[0] = (uint32_t)stamp0, o[1] = (uint32_t)(stamp0 >> 32), o[2] = (uint32_t)stamp1, o[3] = (uint32_t)(stamp1 >> 32);
    o[4] = hwid, o[5] = xcc, o[6] = blockIdx.x, o[7] = wave;
    const uint64_t cyc1 = __builtin_amdgcn_s_memtime();
    o[8] = (uint32_t)(cyc1 - cyc0);
    o[9] = o[10] = o[11] = o[12] = o[13] = o[14] = 0;
  }
#endif
}

// =============================================================================================
// row_kernel: one wavefront per output row, for programs with vector x vector products (x_i (.) g_i sums, products
// with the per-proof scalars g and f: linear.rs:94,124-129, sum.rs:107-115,154-160,301-319, commit.rs:199-209).
// Such a term needs two forward transforms whose results must both be in registers for the multiplication, so the
// unit kernel's register discipline (nothing live while an operand is transformed) does not apply; what pays here is
// the running sum staying in registers across the terms and the Garner word A staying in LDS (measured against
// unit_kernel's parked sums and global state lines: 1.36 vs 1.85 ms per launch for the Sum rows at (4,9,4), V = 8).
// Primes one after the other; the first pass measures the operands (prime count, canonical test, norm marks).
// =============================================================================================
template <int LOGN, bool HAS_SHIFT, class TM = WaveTeam, bool DD = false>
__global__ void __launch_bounds__(TM::kTeamsPerBlock << TM::LL, (LOGN <= 10 || TM::LL == 7 ? 4 : 1))   // N <= 1024 and teams of two: hold the 4 waves per SIMD the LDS allows
row_kernel(const Program* __restrict__ prog, const Operands ops, const uint32_t* __restrict__ key_ntt,
           const double* __restrict__ key_l2, const DevTables* __restrict__ Tp, const uint32_t* __restrict__ tw_all,
           uint32_t* __restrict__ scratch, uint8_t* __restrict__ flags, const uint32_t ntasks) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  constexpr bool OPQ = true;   // opaque lane ids: no hoisted address registers
  constexpr int TPB = TM::kTeamsPerBlock;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & (G::LANES - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> TM::LL);
  // per team: transposition slab, then Garner word A (one per coefficient); together they also hold the 2N-word
  // image of a rotation term, which is finished before the transforms start
  uint32_t* lds = smem + wave * (G::LDS_WORDS + N);
  uint32_t* st_lds = lds + G::LDS_WORDS;
  uint32_t* st = scratch + ((size_t)blockIdx.x * TPB + wave) * (size_t)(kScratchLines * N + 16);
  uint32_t* st_glb = st;            // Garner word B, only touched when a row needs the third prime
  uint32_t* st_sh = st + 4 * N;     // sum of the row's rotation terms mod q
  const DevTables& T = *Tp;
  const uint32_t qhalf = T.crt.qhalf;
  const bool trusted = ops.trusted != 0;
  const uint32_t nrows = prog->nrows;

  // When the task stride is a multiple of the row count a team would meet the same row of the program on every trip —
  // and with it the same SIMD (wave i of a workgroup lands on SIMD i): rows of different cost (Linear's verifier: two
  // relation rows with a rotation term, a key row, a vector x vector row) then load the SIMDs unevenly.  The row index
  // is rotated by the trip count in that case (a permutation inside each batch entry).
  const uint32_t stride = gridDim.x * TPB;
  const bool rotate_rows = RZK_ROW_ROTATE && nrows > 1 && stride % nrows == 0;
  uint32_t trip = 0;
  for (uint32_t task = blockIdx.x * TPB + wave; task < ntasks; task += stride, ++trip) {
    const uint32_t b = task / nrows;
    uint32_t rowi = task - b * nrows;
    if (rotate_rows) {
      rowi += trip % nrows;
      rowi = rowi >= nrows ? rowi - nrows : rowi;
    }
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    const Row row = table_load(&prog->rows[rowi]);
    const bool has_shift = HAS_SHIFT && row.nshift > 0;
    if (has_shift) {
      bool fault = false;
#pragma unroll 1
      for (uint32_t t = 0; t < row.nshift; ++t) {
        const Term tm = table_load(&prog->terms[row.term0 + row.nterms + t]);
        const int64_t* __restrict__ pa = operand_ptr(ops, tm.a_op, tm.a_off, b, bo, N);
        int32_t a[E];
        if (trusted) {
#pragma unroll
          for (int e = 0; e < E; ++e) a[e] = (int32_t)pa[G::j_p1(lane, e)];
        } else {
          uint32_t abad = 0, amx = 0;
#pragma unroll
          for (int e = 0; e < E; ++e) a[e] = canon_lo_mx(pa[G::j_p1(lane, e)], qhalf, abad, amx);
          fault = fault || canon_fail(abad, amx, qhalf);
        }
        shift_product<LOGN, false, true, TM>(st_sh, t == 0, tm.sign < 0, a, operand_ptr(ops, tm.b_op, tm.b_off, b, bo, N), lane,
                                         reinterpret_cast<int32_t*>(lds), T, fault, trusted);
      }
      if (fault) input_fault(ops, flags, bo, lane);
      TM::sync();   // the image is dead: the slab and the state words may be overwritten
    }
    const bool has_terms = row.nterms > 0;
    int np = kMaxPrimes;
    if (has_terms) {
      float bound = 0.f;
#pragma unroll 1
      for (int pi = 0; pi < np; ++pi) {
        const PrimeConsts pc = T.pc[pi];
        const uint32_t* __restrict__ twf = tw_all + (size_t)(2 * pi) * kTableLen;
        const bool first = pi == 0;
        uint32_t acc[E];
#pragma unroll
        for (int c = 0; c < E; ++c) acc[c] = 0;
#pragma unroll 1
        for (uint32_t t = 0; t < row.nterms; ++t)
          term_direct<LOGN, true, OPQ, TM, DD>(acc, table_load(&prog->terms[row.term0 + t]), ops, b, bo, lane, lds, twf, pc, pi, key_ntt,
                                           key_l2, first, bound, flags, qhalf);
        if (first) np = primes_for(bound, T);
        inverse_and_fold<LOGN, OPQ, TM>(pi, np, acc, lane, lds, twf + kTableLen, pc, st_lds, st_glb, T);
      }
    }
    row_epilogue<LOGN, TM>(prog, row, ops, b, bo, lane, has_terms, np, st_lds, T, flags, has_shift ? st_sh : nullptr);
  }
}

// =============================================================================================
// Shift-add row kernel: rows whose products all have a SPARSE multiplier as their `a` operand — the
// challenge d (kappa coefficients +-1, src/challenge_space.rs:12-33) in z = y + r(.)d and in the d-products
// of the verifiers.  No transform at all: the wave keeps the extended image of the other operand in LDS
// (ShiftGeo, rzk_core.h) and adds one rotation per non-zero coefficient of the multiplier; the multiplier's
// coefficients stay in registers and are walked with ballot / readlane (wave-uniform control flow).
// Exact for ANY multiplier (cost ~ its number of non-zeros): sums are kept in 32 bits when the multiplier
// is +-1-valued and |d|_1 |v|_inf < 2^30, in 64 bits (v_mad_i64_i32) below 2^62, and in two 16-bit passes
// beyond that.
// =============================================================================================
#ifndef RZK_SHIFT_MIN_WAVES
#define RZK_SHIFT_MIN_WAVES 1
#endif
template <int LOGN, class TM = WaveTeam>
struct ShiftCfg {   // teams per workgroup: one team's image is 8 * N bytes of LDS, 32 KiB per workgroup at most
  static constexpr int TPB = TM::LL == 6 ? 4 : 1;
  static constexpr int WORDS = ShiftGeo<LOGN, true, TM::LL>::WORDS + (TM::LL == 6 ? 0 : kShiftListWords);   // per team
};

template <int LOGN, bool TRUSTED, class TM = WaveTeam>   // TRUSTED (Operands::trusted) is a template flag here: as a run-time branch around the loads
                                                         // it changed the compiler's load scheduling (79 instead of 116 VGPRs, 86 us instead of 77)
__global__ void __launch_bounds__((ShiftCfg<LOGN, TM>::TPB << TM::LL), (TM::LL == 6 ? RZK_SHIFT_MIN_WAVES : 4))
shift_row_kernel(const Program* __restrict__ prog, const Operands ops, const DevTables* __restrict__ Tp,
                 uint8_t* __restrict__ flags, const uint32_t ntasks) {
  using S = ShiftGeo<LOGN, true, TM::LL>;
  constexpr int E = S::E;
  constexpr int N = S::N;
  constexpr int LANES = S::LANES;
  constexpr int TPB = ShiftCfg<LOGN, TM>::TPB;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & (LANES - 1);                                              // index inside the team
  const uint32_t team = __builtin_amdgcn_readfirstlane(threadIdx.x >> TM::LL);
  int32_t* slab = reinterpret_cast<int32_t*>(smem) + team * ShiftCfg<LOGN, TM>::WORDS;
  const DevTables& T = *Tp;
  const uint32_t q = T.crt.q;
  const uint32_t nrows = prog->nrows;

  // (Tasks of several consecutive rows that keep their common multiplier — the challenge of z = y + r (.) d — in
  // registers from row to row measured no gain at N = 512 / 1024 and a loss at N = 2048: the re-reads hit in L2.)
  for (uint32_t task = blockIdx.x * TPB + team; task < ntasks; task += gridDim.x * TPB) {
    const uint32_t b = task / nrows;
    const uint32_t rowi = task - b * nrows;
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    const Row row = prog->rows[rowi];
    const uint32_t qhalf = T.crt.qhalf;
    constexpr bool trusted = TRUSTED;
    bool fault = false;
    {
      uint32_t res[E];
#pragma unroll
      for (int i = 0; i < E; ++i) res[i] = 0;
#pragma unroll 1
      for (uint32_t t = 0; t < row.nterms; ++t) {
        const Term tm = prog->terms[row.term0 + t];
        int32_t a[E];
        uint32_t abad = 0, amx = 0;
        load_pairs<LOGN, TM::LL>(a, operand_ptr(ops, tm.a_op, tm.a_off, b, bo, N), lane, qhalf, abad, amx, trusted);
        if (!trusted) fault = fault || canon_fail(abad, amx, qhalf);
        shift_product<LOGN, true, false, TM>(res, false, tm.sign < 0, a, operand_ptr(ops, tm.b_op, tm.b_off, b, bo, N), lane,
                                             slab, T, fault, trusted);
      }
      // The sums move to the (now idle) image, each thread's pairs in its own 8-byte slots, so that the additions and
      // the store can run as a rolled loop with few registers and four 16-byte loads in flight per addition.
      TM::sync();
      uint2* own = reinterpret_cast<uint2*>(slab) + lane;
#pragma unroll
      for (int g = 0; g < S::G; ++g) own[g * LANES] = make_uint2(res[2 * g], res[2 * g + 1]);
    }
    constexpr int GC = S::G < 4 ? S::G : 4;   // pairs per trip
    uint32_t in_bad = 0, in_mx = 0;
    int nz = 0;
#pragma unroll 1
    for (int g0 = 0; g0 < S::G; g0 += GC) {
      uint32_t r[2 * GC];
      const uint2* own = reinterpret_cast<const uint2*>(slab) + lane + g0 * LANES;
#pragma unroll
      for (int g = 0; g < GC; ++g) {
        const uint2 v = own[g * LANES];
        r[2 * g] = v.x, r[2 * g + 1] = v.y;
      }
#pragma unroll 1
      for (uint32_t ai = 0; ai < row.nadds; ++ai) {
        const AddTerm ad = prog->adds[row.add0 + ai];
        const longlong2* __restrict__ p =
            reinterpret_cast<const longlong2*>(operand_ptr(ops, ad.op & ADD_OP_MASK, ad.off, b, bo, N)) + g0 * LANES + lane;
        int32_t av[2 * GC];
        if (trusted) {
#pragma unroll
          for (int g = 0; g < GC; ++g) {
            const longlong2 t = ld_stream(p + g * LANES);
            av[2 * g] = (int32_t)t.x, av[2 * g + 1] = (int32_t)t.y;
          }
        } else {
#pragma unroll
          for (int g = 0; g < GC; ++g) canon_pair(ld_stream(p + g * LANES), qhalf, in_bad, in_mx, av[2 * g], av[2 * g + 1]);
        }
        if (ad.sign >= 0) {
#pragma unroll
          for (int i = 0; i < 2 * GC; ++i) r[i] = addq(r[i], zq_from_centered(av[i], q), q);
        } else {
#pragma unroll
          for (int i = 0; i < 2 * GC; ++i) r[i] = subq(r[i], zq_from_centered(av[i], q), q);
        }
      }
      if (row.mode == MODE_STORE) {
        int4* __restrict__ dst =
            reinterpret_cast<int4*>(const_cast<int64_t*>(operand_ptr(ops, row.out_op, row.out_off, b, bo, N))) + g0 * LANES + lane;
#pragma unroll
        for (int g = 0; g < GC; ++g) {
          const int64_t c0 = center_from_zq(r[2 * g], T.crt), c1 = center_from_zq(r[2 * g + 1], T.crt);
          st_stream(dst + g * LANES, make_int4((int32_t)c0, (int32_t)(c0 >> 32), (int32_t)c1, (int32_t)(c1 >> 32)));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2 * GC; ++i) nz |= (r[i] != 0);
      }
    }
    if (fault || canon_fail(in_bad, in_mx, qhalf)) input_fault(ops, flags, bo, lane);
    if (row.mode != MODE_STORE) {
      if (__any(nz) && (lane & 63) == 0) flags[bo] = 0;
    }
    TM::sync();   // the next task's image overwrites the slots read above
  }
}


// ---- row groups ---------------------------------------------------------------------------------------------
// One wavefront evaluates a GROUP of up to kGroupMax rows that are key products over the same operand
// list: each operand is loaded, measured and transformed once per prime and multiplied into one
// accumulator per row.  For [a1;a2].r with (n,k,l) = (4,9,4) that is 23 transforms per prime instead of 56.
// The Garner state of every row of the group lives in a per-wave global scratch line (L2 resident).
#ifndef RZK_GROUP_OPAQUE
#define RZK_GROUP_OPAQUE 1
#endif
#ifndef RZK_GROUP_MIN_WAVES
#define RZK_GROUP_MIN_WAVES 1
#endif
template <int LOGN, int GM>
__global__ void __launch_bounds__(256, RZK_GROUP_MIN_WAVES)
row_group_kernel(const Program* __restrict__ prog, const Operands ops, const uint32_t* __restrict__ key_ntt,
                 const double* __restrict__ key_l2, const DevTables* __restrict__ Tp,
                 const uint32_t* __restrict__ tw_all, uint32_t* __restrict__ scratch, uint8_t* __restrict__ flags,
                 const uint32_t ntasks) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  static_assert(GM >= 1 && GM <= kGroupMax, "group size");
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* lds = smem + wave * G::LDS_WORDS;
  uint32_t* st = scratch + ((size_t)blockIdx.x * 4 + wave) * (size_t)(2 * kGroupMax) * N;   // [g][A|B][N]
  const DevTables& T = *Tp;
  const uint32_t ngroups = prog->ngroups;

  for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
    const uint32_t b = task / ngroups;
    const uint32_t gi = task - b * ngroups;
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    const GroupDesc gd = prog->groups[gi];
    const uint32_t cnt = gd.count;
    const Row row0 = prog->rows[gd.row0];
    const uint32_t nt = row0.nterms;
    int np = kMaxPrimes;
    if (nt > 0) {
      float bound[GM];
#pragma unroll
      for (int g = 0; g < GM; ++g) bound[g] = 0.f;
#pragma unroll 1
      for (int pi = 0; pi < np; ++pi) {
        const PrimeConsts pc = T.pc[pi];
        const uint32_t* __restrict__ twf = tw_all + (size_t)(2 * pi) * kTableLen;
        const bool first = pi == 0;
        uint32_t acc[GM][E];
#pragma unroll
        for (int g = 0; g < GM; ++g)
#pragma unroll
          for (int c = 0; c < E; ++c) acc[g][c] = 0;
#pragma unroll 1
        for (uint32_t t = 0; t < nt; ++t) {
          const Term tm0 = prog->terms[row0.term0 + t];
          uint32_t x[E];
          float nb = 0.f;
          bool below = true;
          const bool chk = first && (tm0.kind & TERM_CHECK);
          int ln = lane;
          if (RZK_GROUP_OPAQUE) asm volatile("" : "+v"(ln));   // no hoisting of lane-dependent addresses (register budget)
          bool fault = false;
          load_lift<LOGN>(x, operand_ptr(ops, tm0.b_op, tm0.b_off, b, bo, N), ln, pc, first, nb, chk, ops.norm_limit, below,
                          T.crt.qhalf, ops.trusted != 0, fault);
          if (chk && !below && lane == 0) flags[bo] = 0;
          if (fault) input_fault(ops, flags, bo, lane);
          wave_fwd<LOGN>(x, ln, lds, twf, pc);
          store_operand_image<LOGN, WaveTeam>(x, ops, tm0.b_op, tm0.b_off, b, pi, ln, nb, first);
#pragma unroll
          for (int g = 0; g < GM; ++g) {
            if ((uint32_t)g < cnt) {
              const Term tg = prog->terms[prog->rows[gd.row0 + g].term0 + t];
              if (first) bound[g] = bound_fma((float)key_l2[tg.a_off], nb, bound[g]);
              const uint4* __restrict__ kp =
                  reinterpret_cast<const uint4*>(key_ntt + ((size_t)tg.a_off * kKeyImages + pi) * N);
              if (tg.sign >= 0) {   // (one wave-uniform branch per term, not a select per coefficient)
#pragma unroll
                for (int q4 = 0; q4 < E / 4; ++q4) {
                  const uint4 kv = kp[q4 * 64 + ln];
                  const uint32_t ks[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
                  for (int i = 0; i < 4; ++i) acc[g][4 * q4 + i] = mac_add(acc[g][4 * q4 + i], x[4 * q4 + i], ks[i], pc);
                }
              } else {
#pragma unroll
                for (int q4 = 0; q4 < E / 4; ++q4) {
                  const uint4 kv = kp[q4 * 64 + ln];
                  const uint32_t ks[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
                  for (int i = 0; i < 4; ++i) acc[g][4 * q4 + i] = mac_sub(acc[g][4 * q4 + i], x[4 * q4 + i], ks[i], pc);
                }
              }
            }
          }
        }
        if (first) {
          float mxb = bound[0];
#pragma unroll
          for (int g = 1; g < GM; ++g) mxb = bound[g] > mxb ? bound[g] : mxb;
          np = primes_for(mxb, T);
        }
        // one inverse-transform instance in a rolled loop; the row's accumulator is picked with selects so
        // that the accumulator array keeps static register indices
#pragma unroll 1
        for (uint32_t g = 0; g < cnt; ++g) {
          uint32_t w[E];
#pragma unroll
          for (int c = 0; c < E; ++c) {
            uint32_t v = acc[0][c];
#pragma unroll
            for (int gg = 1; gg < GM; ++gg) v = g == (uint32_t)gg ? acc[gg][c] : v;
            w[c] = v;
          }
          inverse_and_fold<LOGN, RZK_GROUP_OPAQUE != 0>(pi, np, w, lane, lds, twf + kTableLen, pc, st + (size_t)(2 * g) * N,
                                                        st + (size_t)(2 * g + 1) * N, T);
        }
      }
    }
#pragma unroll 1
    for (uint32_t g = 0; g < cnt; ++g)
      row_epilogue<LOGN>(prog, prog->rows[gd.row0 + g], ops, b, bo, lane, nt > 0, np, st + (size_t)(2 * g) * N, T, flags);
  }
}

// ---- row blocks ---------------------------------------------------------------------------------------------
// One workgroup of kBlockWaves wavefronts evaluates one block (rzk_dev.h: BlockPlan) of one proof.  Per prime:
//   phase 1  wave w transforms operands w, w+8, ... of the block and leaves them in LDS ([c][lane] order:
//            lane-consecutive words, conflict-free); the first prime also measures the operands' norms;
//   barrier; phase 2  wave w evaluates rows w, w+8, ...: multiply-accumulate from the staged transforms and the
//            resident key, inverse transform, fold into the row's Garner state (workgroup scratch in global
//            memory); barrier before the next prime overwrites the staged transforms.
// Every wave runs the same number of barriers: the prime count is the block's maximum, computed by every wave
// from the same norms in LDS (more primes than a row needs is still exact).
template <int LOGN, class TM = WaveTeam>
__global__ void __launch_bounds__(kBlockWaves << TM::LL)
row_block_kernel(const Program* __restrict__ prog, const BlockPlan* __restrict__ plan, const Operands ops,
                 const uint32_t* __restrict__ key_ntt, const double* __restrict__ key_l2,
                 const DevTables* __restrict__ Tp, const uint32_t* __restrict__ tw_all, uint32_t* __restrict__ scratch,
                 uint8_t* __restrict__ flags, const uint32_t ntasks) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & (G::LANES - 1);                                    // index inside the team
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> TM::LL);      // team of the workgroup (kBlockWaves teams)
  if constexpr (TM::LL == 7) TM::init();
  uint32_t* staged = smem;                                                     // [kBlockMaxSlots][N]
  uint32_t* lds = smem + kBlockMaxSlots * N + wave * G::LDS_WORDS;             // this wave's transposition slab
  float* norm1 = reinterpret_cast<float*>(smem + kBlockMaxSlots * N + kBlockWaves * G::LDS_WORDS);   // [slots]
  uint32_t* st = scratch + (size_t)blockIdx.x * (size_t)(2 * kBlockMaxRows) * N;   // [row][A|B][N]
  const DevTables& T = *Tp;
  const uint32_t nblocks = plan->nblocks;

  for (uint32_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
    const uint32_t b = task / nblocks;
    const BlockDesc bd = plan->blk[task - b * nblocks];
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    int np = kMaxPrimes;
#pragma unroll 1
    for (int pi = 0; pi < np; ++pi) {
      const PrimeConsts pc = T.pc[pi];
      const uint32_t* __restrict__ twf = tw_all + (size_t)(2 * pi) * kTableLen;
      const bool first = pi == 0;
      // ---- phase 1: operand transforms into LDS
#pragma unroll 1
      for (uint32_t s = wave; s < bd.nslots; s += kBlockWaves) {
        const uint32_t gs = bd.slot0 + s;
        uint32_t x[E];
        float nb = 0.f;
        bool below = true;
        const bool chk = first && plan->slot_check[gs] && ops.norm_limit;
        bool fault = false;
        int ln = lane;
        asm volatile("" : "+v"(ln));   // opaque lane ids: no lane-dependent addresses kept in registers across the steps
        load_lift<LOGN, TM>(x, operand_ptr(ops, plan->slot_op[gs], plan->slot_off[gs], b, bo, N), ln, pc, first, nb, chk,
                        ops.norm_limit, below, T.crt.qhalf, ops.trusted != 0, fault);
        if (chk && !below && (lane & 63) == 0) flags[bo] = 0;
        if (fault) input_fault(ops, flags, bo, lane);
        if (first && lane == 0) norm1[s] = nb;
        wave_fwd<LOGN, TM>(x, ln, lds, twf, pc);
        store_operand_image<LOGN, TM>(x, ops, plan->slot_op[gs], plan->slot_off[gs], b, pi, ln, nb, first);
        uint32_t* dst = staged + s * N + ln;
#pragma unroll
        for (int c = 0; c < E; ++c) dst[c * G::LANES] = x[c];
      }
      __syncthreads();
      if (first) {
        float mx = 0.f;
#pragma unroll 1
        for (uint32_t r = 0; r < bd.nrows; ++r) {
          const Row row = prog->rows[bd.row0 + r];
          float bound = 0.f;
#pragma unroll 1
          for (uint32_t t = 0; t < row.nterms; ++t)
            bound = bound_fma((float)key_l2[prog->terms[row.term0 + t].a_off], norm1[plan->term_slot[row.term0 + t]], bound);
          mx = bound > mx ? bound : mx;
        }
        np = primes_for(mx, T);
      }
      // ---- phase 2: rows from the staged transforms
#pragma unroll 1
      for (uint32_t r = wave; r < bd.nrows; r += kBlockWaves) {
        const Row row = prog->rows[bd.row0 + r];
        if (row.nterms == 0) continue;
        uint32_t acc[E];
#pragma unroll
        for (int c = 0; c < E; ++c) acc[c] = 0;
#pragma unroll 1
        for (uint32_t t = 0; t < row.nterms; ++t) {
          const Term tm = prog->terms[row.term0 + t];
          int lm = lane;
          asm volatile("" : "+v"(lm));
          const uint32_t* __restrict__ xs = staged + (size_t)plan->term_slot[row.term0 + t] * N + lm;
          const uint4* __restrict__ kp = reinterpret_cast<const uint4*>(key_ntt + ((size_t)tm.a_off * kKeyImages + pi) * N);
          if (tm.sign >= 0) {   // (one wave-uniform branch per term, not a select per coefficient)
#pragma unroll
            for (int g = 0; g < E / 4; ++g) {
              const uint4 kv = kp[G::key4(lm, g)];
              const uint32_t ks[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[4 * g + i] = mac_add(acc[4 * g + i], xs[(4 * g + i) * G::LANES], ks[i], pc);
            }
          } else {
#pragma unroll
            for (int g = 0; g < E / 4; ++g) {
              const uint4 kv = kp[G::key4(lm, g)];
              const uint32_t ks[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[4 * g + i] = mac_sub(acc[4 * g + i], xs[(4 * g + i) * G::LANES], ks[i], pc);
            }
          }
        }
        inverse_and_fold<LOGN, true, TM>(pi, np, acc, lane, lds, twf + kTableLen, pc, st + (size_t)(2 * r) * N,
                                     st + (size_t)(2 * r + 1) * N, T);
      }
      __syncthreads();   // the staged transforms are overwritten by the next prime / next task
    }
#pragma unroll 1
    for (uint32_t r = wave; r < bd.nrows; r += kBlockWaves) {
      const Row row = prog->rows[bd.row0 + r];
      row_epilogue<LOGN, TM>(prog, row, ops, b, bo, lane, row.nterms > 0, np, st + (size_t)(2 * r) * N, T, flags);
    }
  }
}

// acc +/-= stored transform (*) (key entry | second stored transform), 16-byte accesses in the NTT-domain layout
template <int LOGN, bool VEC, bool MINUS>
__device__ __forceinline__ void slot_mac(uint32_t* acc, const uint4* __restrict__ xb, const uint4* __restrict__ other, int lane,
                                         const PrimeConsts& pc) {
  constexpr int E = Geo<LOGN>::E;
#pragma unroll
  for (int g = 0; g < E / 4; ++g) {
    const uint4 xv = xb[g * 64 + lane];
    const uint4 ov = other[g * 64 + lane];
    uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w};
    const uint32_t os[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t w = os[i];
      if (VEC) {   // x_a * x_b * N^-1: two Montgomery steps (the key already carries N^-1 * R)
        xs[i] = mont_lazy(xs[i], os[i], pc.p, pc.npinv);
        w = pc.ninv_r2;
      }
      acc[4 * g + i] = MINUS ? mac_sub(acc[4 * g + i], xs[i], w, pc) : mac_add(acc[4 * g + i], xs[i], w, pc);
    }
  }
}

// ---- shared-operand path ------------------------------------------------------------------------------------
// Forward pass: one wavefront per (proof, slot) transforms the slot's polynomial for the first `np_store`
// primes into ws[((b*nslots + s)*np_store + pi)*N ...] (canonical residues, NTT-domain layout) and records
// its 1-norm / max-norm in norms[(b*nslots + s)*2 ..]; slots of a checked vector also evaluate the fused
// norm predicate.
template <int LOGN>
__global__ void __launch_bounds__(256)
fwd_slots_kernel(const SlotTable* __restrict__ slots, const Operands ops, const DevTables* __restrict__ Tp,
                 const uint32_t* __restrict__ tw_all, uint32_t* __restrict__ ws, double* __restrict__ norms,
                 uint8_t* __restrict__ flags, const uint32_t ntasks, const uint32_t np_store) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* lds = smem + wave * G::LDS_WORDS;
  const DevTables& T = *Tp;
  const uint32_t nslots = slots->nslots;
  for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
    const uint32_t b = task / nslots;
    const uint32_t s = task - b * nslots;
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    const int64_t* __restrict__ src = operand_ptr(ops, slots->op[s], slots->off[s], b, bo, N);
    int32_t v[E];
    if (ops.trusted) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = (int32_t)src[G::j_p1(lane, e)];
    } else {
      uint32_t in_bad = 0, in_mx = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = canon_lo_mx(src[G::j_p1(lane, e)], T.crt.qhalf, in_bad, in_mx);
      if (canon_fail(in_bad, in_mx, T.crt.qhalf)) input_fault(ops, flags, bo, lane);
    }
    const float ss = wave_sum_f32(lane_sum_sq_f32<E>(v));
    if (lane == 0) {
      norms[((size_t)b * nslots + s) * 2 + 0] = (double)norm2_upper(ss) * (1.0 + 1e-6);   // upper bound of the 2-norm (read back as float)
      norms[((size_t)b * nslots + s) * 2 + 1] = 0.0;
    }
    if (slots->check[s] && ops.norm_limit) {
      if (!norm_below<E>(v, ss, ops.norm_limit) && lane == 0) flags[bo] = 0;
    }
#pragma unroll 1
    for (uint32_t pi = 0; pi < np_store; ++pi) {
      const PrimeConsts pc = T.pc[pi];
      uint32_t x[E];
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = lift(v[e], pc);
      wave_fwd<LOGN>(x, lane, lds, tw_all + (size_t)(2 * pi) * kTableLen, pc);
      uint4* __restrict__ dst = reinterpret_cast<uint4*>(ws + (((size_t)b * nslots + s) * np_store + pi) * N);
#pragma unroll
      for (int g = 0; g < E / 4; ++g) {
        uint4 o;
        o.x = csub(csub(x[4 * g + 0], pc.twop), pc.p);
        o.y = csub(csub(x[4 * g + 1], pc.twop), pc.p);
        o.z = csub(csub(x[4 * g + 2], pc.twop), pc.p);
        o.w = csub(csub(x[4 * g + 3], pc.twop), pc.p);
        dst[g * 64 + lane] = o;
      }
    }
  }
}

// Row pass of the shared-operand path.  Work is dealt so that all rows of a proof run on workgroups
// whose ids are congruent mod 8 (one XCD under the observed round-robin placement: the proof's stored
// transforms then come from that XCD's L2; placement affects speed only, never results).
template <int LOGN>
__global__ void __launch_bounds__(256, RZK_ROW_MIN_WAVES)
row_slots_kernel(const Program* __restrict__ prog, const SlotTable* __restrict__ slots, const Operands ops,
                 const uint32_t* __restrict__ key_ntt, const double* __restrict__ key_l2,
                 const DevTables* __restrict__ Tp, const uint32_t* __restrict__ tw_all,
                 const uint32_t* __restrict__ ws, const double* __restrict__ norms, uint32_t* __restrict__ scratch,
                 uint8_t* __restrict__ flags, const uint32_t batch, const uint32_t np_store) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* lds = smem + wave * G::LDS_WORDS;
  uint32_t* st_lds = smem + 4 * G::LDS_WORDS + wave * N;
  uint32_t* st_glb = scratch + ((size_t)blockIdx.x * 4 + wave) * N;
  const DevTables& T = *Tp;
  const uint32_t nrows = prog->nrows;
  const uint32_t nslots = slots->nslots;
  const uint32_t groups = (nrows + 3) / 4;                 // row groups (4 rows, one per wave) per proof
  // item stream of this workgroup's XCD class: proofs xcd, xcd+8, ... ; each proof contributes `groups` items
  const uint32_t xcd = blockIdx.x & 7, lane_blocks = (gridDim.x + 7 - xcd) / 8;   // workgroups in this class
  const uint32_t proofs_here = batch > xcd ? (batch - xcd + 7) / 8 : 0;
  const uint32_t items = proofs_here * groups;
  for (uint32_t item = blockIdx.x >> 3; item < items; item += lane_blocks) {
    const uint32_t b = xcd + 8 * (item / groups);
    const uint32_t rowi = (item % groups) * 4 + wave;
    if (rowi >= nrows) continue;
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    const Row row = prog->rows[rowi];
    const bool has_terms = row.nterms > 0;
    int np = 1;
    if (has_terms) {
      const double* __restrict__ nb = norms + (size_t)b * nslots * 2;
      float bound = 0.f;
#pragma unroll 1
      for (uint32_t t = 0; t < row.nterms; ++t) {
        const Term tm = prog->terms[row.term0 + t];
        const uint32_t sb = slots->term_b[row.term0 + t];
        if ((tm.kind & TERM_KIND_MASK) == TERM_VEC) {
          const uint32_t sa = slots->term_a[row.term0 + t];
          bound = bound_fma((float)nb[2 * sa], (float)nb[2 * sb], bound);   // |a (*) b|_inf <= |a|_2 |b|_2
        } else {
          bound = bound_fma((float)key_l2[tm.a_off], (float)nb[2 * sb], bound);
        }
      }
      np = primes_for(bound, T);
#pragma unroll 1
      for (int pi = 0; pi < np; ++pi) {
        const PrimeConsts pc = T.pc[pi];
        const uint32_t* __restrict__ twf = tw_all + (size_t)(2 * pi) * kTableLen;
        uint32_t acc[E];
#pragma unroll
        for (int c = 0; c < E; ++c) acc[c] = 0;
        if ((uint32_t)pi < np_store) {
          // stored transforms: multiply-accumulate only
#pragma unroll 1
          for (uint32_t t = 0; t < row.nterms; ++t) {
            const Term tm = prog->terms[row.term0 + t];
            const uint4* __restrict__ xb = reinterpret_cast<const uint4*>(
                ws + (((size_t)b * nslots + slots->term_b[row.term0 + t]) * np_store + pi) * N);
            const bool vec = (tm.kind & TERM_KIND_MASK) == TERM_VEC;
            const uint4* __restrict__ other =
                vec ? reinterpret_cast<const uint4*>(
                          ws + (((size_t)b * nslots + slots->term_a[row.term0 + t]) * np_store + pi) * N)
                    : reinterpret_cast<const uint4*>(key_ntt + ((size_t)tm.a_off * kKeyImages + pi) * N);
            // four straight-line variants behind wave-uniform branches (a select per coefficient would evaluate both
            // the add and the subtract form)
            if (vec) {
              if (tm.sign >= 0) slot_mac<LOGN, true, false>(acc, xb, other, lane, pc);
              else slot_mac<LOGN, true, true>(acc, xb, other, lane, pc);
            } else {
              if (tm.sign >= 0) slot_mac<LOGN, false, false>(acc, xb, other, lane, pc);
              else slot_mac<LOGN, false, true>(acc, xb, other, lane, pc);
            }
          }
        } else {
          // more primes needed than were stored: transform in the wave for the missing ones
          float unused = 0.f;
#pragma unroll 1
          for (uint32_t t = 0; t < row.nterms; ++t) {
            Term tm = prog->terms[row.term0 + t];
            tm.kind &= TERM_KIND_MASK;   // norm predicate already evaluated by the forward pass
            term_direct<LOGN, true>(acc, tm, ops, b, bo, lane, lds, twf, pc, pi, key_ntt, key_l2, false, unused, flags,
                                    T.crt.qhalf);
          }
        }
        inverse_and_fold<LOGN>(pi, np, acc, lane, lds, twf + kTableLen, pc, st_lds, st_glb, T);
      }
    }
    row_epilogue<LOGN>(prog, row, ops, b, bo, lane, has_terms, np, st_lds, T, flags);
  }
}

// =============================================================================================
// Key transform: centred key entries -> NTT domain (x N^-1, Montgomery form) for all three primes
// =============================================================================================
template <int LOGN>
__global__ void __launch_bounds__(256)
key_transform_kernel(const int64_t* __restrict__ key, uint32_t entries, uint32_t* __restrict__ key_ntt,
                     const DevTables* __restrict__ Tp, const uint32_t* __restrict__ tw_all) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const DevTables& T = *Tp;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* lds = smem + wave * G::LDS_WORDS;
  const uint32_t ntasks = entries * kKeyImages;
  for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
    const uint32_t entry = task / kKeyImages;
    const int pi = task % kKeyImages;
    const PrimeConsts pc = T.pc[pi];
    const int64_t* __restrict__ src = key + (uint64_t)entry * N;
    uint32_t x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = lift((int32_t)src[G::j_p1(lane, e)], pc);
    wave_fwd<LOGN>(x, lane, lds, tw_all + (size_t)(2 * pi) * kTableLen, pc);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(key_ntt + ((uint64_t)entry * kKeyImages + pi) * N);
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      uint4 v;
      v.x = csub(mont_lazy(x[4 * g + 0], pc.ninv_r2, pc.p, pc.npinv), pc.p);
      v.y = csub(mont_lazy(x[4 * g + 1], pc.ninv_r2, pc.p, pc.npinv), pc.p);
      v.z = csub(mont_lazy(x[4 * g + 2], pc.ninv_r2, pc.p, pc.npinv), pc.p);
      v.w = csub(mont_lazy(x[4 * g + 3], pc.ninv_r2, pc.p, pc.npinv), pc.p);
      dst[g * 64 + lane] = v;
    }
  }
}

// Per-entry multiplier images (Operands::dkey_img): like key_transform_kernel, for polynomials that arrive with the batch
// (the g_i of the Linear / Sum proofs).  One wavefront per polynomial: canonical test, 2-norm, then the three images.
template <int LOGN>
__global__ void __launch_bounds__(256)
dkey_transform_kernel(const int64_t* __restrict__ g, uint64_t count, uint32_t dkey_n, uint32_t* __restrict__ img,
                      double* __restrict__ l2, const DevTables* __restrict__ Tp, const uint32_t* __restrict__ tw_all,
                      uint8_t* __restrict__ flags, uint32_t* __restrict__ bad_word, uint32_t two_bit, uint32_t trusted) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const DevTables& T = *Tp;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* lds = smem + wave * G::LDS_WORDS;
  for (uint64_t poly = (uint64_t)blockIdx.x * 4 + wave; poly < count; poly += (uint64_t)gridDim.x * 4) {
    const int64_t* __restrict__ src = g + poly * N;
    int32_t v[E];
    if (trusted) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = (int32_t)src[G::j_p1(lane, e)];
    } else {
      uint32_t in_bad = 0, in_mx = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = canon_lo_mx(src[G::j_p1(lane, e)], T.crt.qhalf, in_bad, in_mx);
      if (canon_fail(in_bad, in_mx, T.crt.qhalf) && lane == 0) {   // as input_fault: the proof's verdict (all bits) and the sticky word
        const uint64_t entry = poly / dkey_n;
        if (flags) {
          if (two_bit) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(flags + entry);
            __hip_atomic_fetch_and(reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3), ~(0xffu << (8u * (uint32_t)(a & 3u))),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            flags[entry] = 0;
          }
        }
        if (bad_word) *bad_word = 1u;
      }
    }
    const float ss = wave_sum_f32(lane_sum_sq_f32<E>(v));
    if (lane == 0) l2[poly] = (double)norm2_upper(ss) * (1.0 + 1e-6);   // upper bound of the 2-norm (read back as float)
#pragma unroll 1
    for (int pi = 0; pi < kKeyImages; ++pi) {
      const PrimeConsts pc = T.pc[pi];
      uint32_t x[E];
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = lift(v[e], pc);
      wave_fwd<LOGN>(x, lane, lds, tw_all + (size_t)(2 * pi) * kTableLen, pc);
      uint4* __restrict__ dst = reinterpret_cast<uint4*>(img + (poly * kKeyImages + pi) * N);
#pragma unroll
      for (int q4 = 0; q4 < E / 4; ++q4) {
        uint4 o;
        o.x = csub(mont_lazy(x[4 * q4 + 0], pc.ninv_r2, pc.p, pc.npinv), pc.p);
        o.y = csub(mont_lazy(x[4 * q4 + 1], pc.ninv_r2, pc.p, pc.npinv), pc.p);
        o.z = csub(mont_lazy(x[4 * q4 + 2], pc.ninv_r2, pc.p, pc.npinv), pc.p);
        o.w = csub(mont_lazy(x[4 * q4 + 3], pc.ninv_r2, pc.p, pc.npinv), pc.p);
        dst[q4 * 64 + lane] = o;
      }
    }
  }
}

// =============================================================================================
// Stand-alone batched transforms over one auxiliary prime (the "batched NTT" of the headline metric)
// =============================================================================================
template <int LOGN>
__global__ void __launch_bounds__(256)
ntt_fwd_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t count, int pi,
               const DevTables* __restrict__ Tp, const uint32_t* __restrict__ tw_all) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const DevTables& T = *Tp;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* lds = smem + wave * G::LDS_WORDS;
  const PrimeConsts pc = T.pc[pi];
  const uint32_t* __restrict__ tw = tw_all + (size_t)(2 * pi) * kTableLen;
  for (uint64_t poly = (uint64_t)blockIdx.x * 4 + wave; poly < count; poly += (uint64_t)gridDim.x * 4) {
    const uint32_t* __restrict__ src = in + poly * N;
    uint32_t x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = src[G::j_p1(lane, e)];
    wave_fwd<LOGN>(x, lane, lds, tw, pc);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(out + poly * N);
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      uint4 v;
      v.x = csub(csub(x[4 * g + 0], pc.twop), pc.p);
      v.y = csub(csub(x[4 * g + 1], pc.twop), pc.p);
      v.z = csub(csub(x[4 * g + 2], pc.twop), pc.p);
      v.w = csub(csub(x[4 * g + 3], pc.twop), pc.p);
      dst[g * 64 + lane] = v;
    }
  }
}

template <int LOGN>
__global__ void __launch_bounds__(256)
ntt_inv_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t count, int pi,
               const DevTables* __restrict__ Tp, const uint32_t* __restrict__ tw_all) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const DevTables& T = *Tp;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* lds = smem + wave * G::LDS_WORDS;
  const PrimeConsts pc = T.pc[pi];
  const uint32_t* __restrict__ tw = tw_all + (size_t)(2 * pi + 1) * kTableLen;
  for (uint64_t poly = (uint64_t)blockIdx.x * 4 + wave; poly < count; poly += (uint64_t)gridDim.x * 4) {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(in + poly * N);
    uint32_t x[E];
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      const uint4 v = src[g * 64 + lane];
      x[4 * g + 0] = v.x;
      x[4 * g + 1] = v.y;
      x[4 * g + 2] = v.z;
      x[4 * g + 3] = v.w;
    }
    wave_inv<LOGN>(x, lane, lds, tw, pc);
    uint32_t* __restrict__ dst = out + poly * N;
#pragma unroll
    for (int e = 0; e < E; ++e)
      dst[G::j_p1(lane, e)] = csub(mont_lazy(x[e], pc.ninv_r, pc.p, pc.npinv), pc.p);
  }
}

// =============================================================================================
// Element-wise kernels: Mat::add / Mat::sub, norm predicate, equality
// =============================================================================================
__global__ void __launch_bounds__(256)
addsub_kernel(const int64_t* a, const int64_t* b, int64_t* out, uint64_t n2, int sub,
              const DevTables* __restrict__ Tp, uint32_t* __restrict__ bad_word) {
  // two coefficients (16 bytes) per thread and step; out may alias a or b (in-place add/sub)
  const DevTables& T = *Tp;
  const longlong2* a2 = reinterpret_cast<const longlong2*>(a);
  const longlong2* b2 = reinterpret_cast<const longlong2*>(b);
  longlong2* o2 = reinterpret_cast<longlong2*>(out);
  const uint64_t h = T.crt.qhalf;
  bool fault = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const longlong2 x = a2[i], y = b2[i];
    fault = fault || (uint64_t)x.x + h > 2 * h || (uint64_t)x.y + h > 2 * h || (uint64_t)y.x + h > 2 * h ||
            (uint64_t)y.y + h > 2 * h;   // canonical inputs only (see canon_lo)
    longlong2 r;
    r.x = center_rounds<1>(sub ? x.x - y.x : x.x + y.x, T.crt);
    r.y = center_rounds<1>(sub ? x.y - y.y : x.y + y.y, T.crt);
    o2[i] = r;
  }
  if (fault && bad_word) *bad_word = 1u;
}

// One wavefront per proof: all `rows` polynomials must satisfy sum c^2 < limit (= (bound+1)^2),
// i.e. floor(sqrt(sum c^2)) <= bound (src/polynomial.rs:60-73, src/params.rs:105-107).
// The sum is exact: c^2 split into 32-bit halves, accumulated in two 64-bit lane sums.
template <int LOGN>
__global__ void __launch_bounds__(256)
norm_kernel(const int64_t* __restrict__ v, uint32_t rows, uint64_t limit_hi, uint64_t limit_lo,
            uint8_t* __restrict__ ok, uint64_t B, int and_mode, int shift, uint32_t qhalf,
            uint32_t* __restrict__ bad_word) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (uint64_t b = (uint64_t)blockIdx.x * 4 + wave; b < B; b += (uint64_t)gridDim.x * 4) {
    int good = 1;
    for (uint32_t r = 0; r < rows; ++r) {
      const int64_t* __restrict__ p = v + (b * rows + r) * N;
      uint64_t slo = 0, shi = 0;
      int huge = 0;   // a coefficient outside the centred range: not a ZqI64 value, the predicate fails
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int64_t c = p[G::j_p1(lane, e)];
        const uint64_t a = c < 0 ? 0ull - (uint64_t)c : (uint64_t)c;
        huge |= a > (uint64_t)qhalf;
        const uint64_t al = a & 0xffffffffu;
        const uint64_t ll = al * al;
        slo += ll & 0xffffffffu;
        shi += ll >> 32;
      }
      slo = wave_sum_u64(slo);
      shi = wave_sum_u64(shi);
      // total = shi * 2^32 + slo  (shi, slo < 2^50)
      const uint64_t t_lo32 = slo & 0xffffffffu;
      const uint64_t mid = shi + (slo >> 32);
      const uint64_t tot_lo = (mid << 32) | t_lo32;
      const uint64_t tot_hi = mid >> 32;
      const int lt = (tot_hi < limit_hi) || (tot_hi == limit_hi && tot_lo < limit_lo);
      const int any_huge = __any(huge);
      good &= lt && !any_huge;
      if (any_huge && bad_word && lane == 0) *bad_word = 1u;
    }
    if (lane == 0) {
      if (and_mode == 0)
        ok[b] = (uint8_t)good;
      else if (and_mode == 1)
        ok[b] = (uint8_t)(ok[b] & good);
      else
        ok[b] = (uint8_t)(ok[b] | (good << shift));
    }
  }
}

template <int LOGN>
__global__ void __launch_bounds__(256)
eq_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b, uint32_t rows,
          uint8_t* __restrict__ eq, uint64_t B, uint32_t qhalf, uint32_t* __restrict__ bad_word) {
  using G = Geo<LOGN>;
  constexpr int E = G::E;
  constexpr int N = G::N;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (uint64_t p = (uint64_t)blockIdx.x * 4 + wave; p < B; p += (uint64_t)gridDim.x * 4) {
    int ne = 0, bad = 0;
    for (uint32_t r = 0; r < rows; ++r) {
      const int64_t* __restrict__ pa = a + (p * rows + r) * N;
      const int64_t* __restrict__ pb = b + (p * rows + r) * N;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int64_t x = pa[G::j_p1(lane, e)], y = pb[G::j_p1(lane, e)];
        ne |= (x != y);
        // equality of canonical forms (derived PartialEq): anything else is not a ZqI64 value
        bad |= ((uint64_t)x + qhalf > 2ull * qhalf) | ((uint64_t)y + qhalf > 2ull * qhalf);
      }
    }
    const int any_ne = __any(ne), any_bad = __any(bad);
    if (lane == 0) {
      eq[p] = (uint8_t)((any_ne || any_bad) ? 0 : 1);
      if (any_bad && bad_word) *bad_word = 1u;
    }
  }
}

// =============================================================================================
// Small ring degrees (N = 4 .. 256): the reference's own unit / integration tests run at N = 4 and
// N = 16 (src/mat.rs:241, tests/test.rs:8).  One wavefront still owns one row task, but a transform
// makes no sense below one coefficient per lane, so products are the O(N^2) negacyclic convolution
// in 32-bit Montgomery arithmetic mod q, operands staged in LDS.  Same row programs, operand tables,
// epilogue and flags as the big-N kernel; this path exists for drop-in completeness, not for speed.
// =============================================================================================
__global__ void __launch_bounds__(256)
row_kernel_small(const Program* __restrict__ prog, const Operands ops, const uint32_t* __restrict__ key_mont,
                 const DevTables* __restrict__ Tp, uint8_t* __restrict__ flags, const uint32_t ntasks,
                 const uint32_t N, const uint32_t r2q) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* la = smem + wave * 2 * N;   // left operand, plain residues in [0,q)
  uint32_t* lb = la + N;                // right operand, Montgomery form
  const DevTables& T = *Tp;
  const uint32_t q = T.crt.q;
  const uint32_t nrows = prog->nrows;
  constexpr int EMAX = 4;               // N <= 256 -> at most 4 coefficients per lane

  for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
    const uint32_t b = task / nrows;
    const uint32_t rowi = task - b * nrows;
    const uint32_t bo = ops.group > 1 ? b / ops.group : b;
    const Row row = prog->rows[rowi];
    uint64_t pos[EMAX], neg[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) pos[e] = neg[e] = 0;
    const uint32_t qhalf = T.crt.qhalf;
    uint32_t in_bad = 0, in_mx = 0;   // canonical-input test of every coefficient this row loads

    for (uint32_t t = 0; t < row.nterms; ++t) {
      const Term tm = prog->terms[row.term0 + t];
      const int64_t* __restrict__ pb = operand_ptr(ops, tm.b_op, tm.b_off, b, bo, (int)N);
      if ((tm.kind & TERM_KIND_MASK) == TERM_KEY) {
        const uint32_t* __restrict__ km = key_mont + (size_t)tm.a_off * N;
        for (uint32_t i = lane; i < N; i += 64) {
          la[i] = zq_from_centered(canon_lo_mx(pb[i], qhalf, in_bad, in_mx), q);
          lb[i] = km[i];
        }
      } else {
        const int64_t* __restrict__ pa = operand_ptr(ops, tm.a_op, tm.a_off, b, bo, (int)N);
        for (uint32_t i = lane; i < N; i += 64) {
          la[i] = zq_from_centered(canon_lo_mx(pa[i], qhalf, in_bad, in_mx), q);
          lb[i] = montq_u(zq_from_centered(canon_lo_mx(pb[i], qhalf, in_bad, in_mx), q), r2q, T.crt);
        }
      }
      wave_sync();
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        const uint32_t tt = lane + 64 * e;
        if (tt < N) {
          uint64_t p = 0, m = 0;
          for (uint32_t i = 0; i < N; ++i) {
            const uint32_t prod = montq_u(la[i], lb[(tt - i) & (N - 1)], T.crt);
            if (i > tt) m += prod; else p += prod;   // X^N = -1
          }
          if (tm.sign >= 0) { pos[e] += p; neg[e] += m; } else { pos[e] += m; neg[e] += p; }
        }
      }
      wave_sync();
    }

    int nz = 0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      const uint32_t tt = lane + 64 * e;
      if (tt < N) {
        uint32_t u = subq((uint32_t)(pos[e] % q), (uint32_t)(neg[e] % q), q);
        for (uint32_t a = 0; a < row.nadds; ++a) {
          const AddTerm ad = prog->adds[row.add0 + a];
          const uint32_t v = zq_from_centered(
              canon_lo_mx(operand_ptr(ops, ad.op & ADD_OP_MASK, ad.off, b, bo, (int)N)[tt], qhalf, in_bad, in_mx), q);
          u = ad.sign >= 0 ? addq(u, v, q) : subq(u, v, q);
        }
        if (row.mode == MODE_STORE)
          const_cast<int64_t*>(operand_ptr(ops, row.out_op, row.out_off, b, bo, (int)N))[tt] = center_from_zq(u, T.crt);
        else
          nz |= (u != 0);
      }
    }
    if (row.mode != MODE_STORE) {
      if (__any(nz) && lane == 0) flags[bo] = 0;
    }
    if (canon_fail(in_bad, in_mx, qhalf)) input_fault(ops, flags, bo, (int)lane);
  }
}

// key entries -> Montgomery-form residues mod q (one thread per coefficient)
__global__ void __launch_bounds__(256)
key_mont_kernel(const int64_t* __restrict__ key, uint32_t* __restrict__ key_mont, uint64_t ncoef,
                const DevTables* __restrict__ Tp, uint32_t r2q) {
  const DevTables& T = *Tp;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncoef; i += (uint64_t)gridDim.x * blockDim.x)
    key_mont[i] = montq_u(zq_from_centered((int32_t)key[i], T.crt.q), r2q, T.crt);
}

// norm / equality for any N (used below N = 512): one wavefront per proof
__global__ void __launch_bounds__(256)
norm_kernel_small(const int64_t* __restrict__ v, uint32_t rows, uint64_t limit_hi, uint64_t limit_lo,
                  uint8_t* __restrict__ ok, uint64_t B, int and_mode, int shift, uint32_t N, uint32_t qhalf,
                  uint32_t* __restrict__ bad_word) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (uint64_t b = (uint64_t)blockIdx.x * 4 + wave; b < B; b += (uint64_t)gridDim.x * 4) {
    int good = 1;
    for (uint32_t r = 0; r < rows; ++r) {
      const int64_t* __restrict__ p = v + (b * rows + r) * N;
      uint64_t slo = 0, shi = 0;
      int huge = 0;
      for (uint32_t i = lane; i < N; i += 64) {
        const int64_t c = p[i];
        const uint64_t a = c < 0 ? 0ull - (uint64_t)c : (uint64_t)c;
        huge |= a > (uint64_t)qhalf;
        const uint64_t al = a & 0xffffffffu;
        const uint64_t ll = al * al;
        slo += ll & 0xffffffffu;
        shi += ll >> 32;
      }
      slo = wave_sum_u64(slo);
      shi = wave_sum_u64(shi);
      const uint64_t mid = shi + (slo >> 32);
      const uint64_t tot_lo = (mid << 32) | (slo & 0xffffffffu);
      const uint64_t tot_hi = mid >> 32;
      const int lt = (tot_hi < limit_hi) || (tot_hi == limit_hi && tot_lo < limit_lo);
      const int any_huge = __any(huge);
      good &= lt && !any_huge;
      if (any_huge && bad_word && lane == 0) *bad_word = 1u;
    }
    if (lane == 0) {
      if (and_mode == 0)
        ok[b] = (uint8_t)good;
      else if (and_mode == 1)
        ok[b] = (uint8_t)(ok[b] & good);
      else
        ok[b] = (uint8_t)(ok[b] | (good << shift));
    }
  }
}

__global__ void __launch_bounds__(256)
eq_kernel_small(const int64_t* __restrict__ a, const int64_t* __restrict__ b, uint32_t rows,
                uint8_t* __restrict__ eq, uint64_t B, uint32_t N, uint32_t qhalf, uint32_t* __restrict__ bad_word) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (uint64_t p = (uint64_t)blockIdx.x * 4 + wave; p < B; p += (uint64_t)gridDim.x * 4) {
    int ne = 0, bad = 0;
    const uint64_t n = (uint64_t)rows * N;
    for (uint64_t i = lane; i < n; i += 64) {
      const int64_t x = a[p * n + i], y = b[p * n + i];
      ne |= (x != y);
      bad |= ((uint64_t)x + qhalf > 2ull * qhalf) | ((uint64_t)y + qhalf > 2ull * qhalf);
    }
    const int any_ne = __any(ne), any_bad = __any(bad);
    if (lane == 0) {
      eq[p] = (uint8_t)((any_ne || any_bad) ? 0 : 1);
      if (any_bad && bad_word) *bad_word = 1u;
    }
  }
}

// =============================================================================================
// Device-side samplers: the distributions of the reference's host RNG helpers, drawn with a
// counter-based generator (rzk_rng.h).  One thread draws 4 coefficients from one Philox block (two blocks for
// the wide uniform range), so a polynomial is N/4 independent units and any number of polynomials fills the chip.
//   uniform   random_polynomial_within (src/polynomial.rs:14-25): every coefficient uniform in [-bound, bound]
//   gauss     random_polynomial_in_normal_distribution (polynomial.rs:28-44): (i64) N(0, sigma), i.e. truncated
//             toward zero as I::from_f64 does
//   challenge random_polynomial_from_challenge_set (src/challenge_space.rs:12-33): kappa coefficients +-1 at a
//             uniformly random kappa-subset of the N positions (what shuffling kappa marked slots gives)
// =============================================================================================
// One thread = one Philox block = two coefficients = one 16-byte store at a lane-consecutive address (full lines per wave
// instruction); the polynomial index is a shift (N is a power of two).  pair16: `out` is 16-byte aligned.
__device__ __forceinline__ void store_pair(int64_t* __restrict__ out, uint64_t c0, uint64_t ncoef, int64_t v0, int64_t v1, bool pair16) {
  if (pair16 && c0 + 1 < ncoef) {
    st_stream(reinterpret_cast<int4*>(out + c0), make_int4((int32_t)v0, (int32_t)(v0 >> 32), (int32_t)v1, (int32_t)(v1 >> 32)));
  } else {
    out[c0] = v0;
    if (c0 + 1 < ncoef) out[c0 + 1] = v1;
  }
}

__global__ void __launch_bounds__(256)
sample_uniform_kernel(int64_t* __restrict__ out, uint64_t ncoef, uint32_t log_ring, uint64_t seed, uint32_t stream,
                      uint32_t bound) {
  const uint32_t range = 2u * bound + 1u;   // bound <= (2^32 - 2) / 2
  const bool pair16 = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  const uint32_t pair_mask = (1u << (log_ring - 1)) - 1u;
  for (uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x; u * 2 < ncoef; u += (uint64_t)gridDim.x * 256) {
    const uint64_t poly = u >> (log_ring - 1);
    const uint32_t blk = (uint32_t)u & pair_mask;   // block `blk` of a polynomial gives its coefficients 2 blk, 2 blk + 1
    const Philox4 a = sampler_block(seed, stream, poly, blk);
    const int64_t v0 = (int64_t)uniform_below(a.v[0], a.v[1], range) - (int64_t)bound;
    const int64_t v1 = (int64_t)uniform_below(a.v[2], a.v[3], range) - (int64_t)bound;
    store_pair(out, u * 2, ncoef, v0, v1, pair16);
  }
}

// Box-Muller, one pair per Philox block.  F32 (sigma < 2^19: every sigma the parameter sets produce): the radius from a
// 64-bit uniform through exponent + v_log_f32 of the 24-bit mantissa (no cancellation: the tail reaches 9.4 sigma), the
// angle from a 32-bit uniform through sincospif; absolute error of a sample < 0.1 before the truncation toward zero —
// statistical parity as for the generator itself.  Larger sigma (up to the 2^26 the entry point admits) keeps the
// double-precision form, whose samples need more than 24 bits.
template <bool F32>
__global__ void __launch_bounds__(256)
sample_gauss_kernel(int64_t* __restrict__ out, uint64_t ncoef, uint32_t log_ring, uint64_t seed, uint32_t stream,
                    double sigma) {
  const bool pair16 = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  const uint32_t pair_mask = (1u << (log_ring - 1)) - 1u;
  const float sigf = (float)sigma;
  for (uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x; u * 2 < ncoef; u += (uint64_t)gridDim.x * 256) {
    const uint64_t poly = u >> (log_ring - 1);
    const uint32_t blk = (uint32_t)u & pair_mask;
    const Philox4 a = sampler_block(seed, stream, poly, blk);
    int64_t v0, v1;
    if (F32) {
      // u0 = X 2^-64, X = a.v[0]:a.v[1] (X = 0, probability 2^-64, is taken as 1): log2 u0 = log2 m - 1 - lz, m in [1,2)
      uint64_t X = ((uint64_t)a.v[0] << 32) | a.v[1];
      X = X ? X : 1ull;
      const int lz = __builtin_clzll(X);
      const uint32_t top = (uint32_t)((X << lz) >> 40);                  // 24 bits, top bit set
      const float m = (float)top * (1.0f / 8388608.0f);                  // exact: [1, 2)
      const float l2 = __log2f(m) - (float)(lz + 1);                     // <= -2^-24 (m = 2 - 2^-23, lz = 0)
      const float r = sigf * __fsqrt_rn(-1.3862943611198906f * l2);      // sigma sqrt(-2 ln u0)
      float sn, cs;
      sincospif((float)a.v[2] * (2.0f / 4294967296.0f), &sn, &cs);       // angle 2 pi u1
      v0 = (int64_t)(r * cs);                                            // conversion truncates toward zero, like I::from_f64
      v1 = (int64_t)(r * sn);
    } else {
      const double k = 1.0 / 9007199254740992.0;   // 2^-53: 53-bit uniforms, u0 in (0,1]
      const double u0 = ((double)((((uint64_t)a.v[0] << 32) | a.v[1]) >> 11) + 1.0) * k;
      const double u1 = (double)((((uint64_t)a.v[2] << 32) | a.v[3]) >> 11) * k;
      const double r0 = sigma * sqrt(-2.0 * log(u0));
      double s0, c0d;
      sincospi(2.0 * u1, &s0, &c0d);
      v0 = (int64_t)(r0 * c0d);
      v1 = (int64_t)(r0 * s0);
    }
    store_pair(out, u * 2, ncoef, v0, v1, pair16);
  }
}

// one wavefront per polynomial: Floyd's algorithm for a uniform kappa-subset.  Lane t draws step t's candidate (its own
// Philox block half) in parallel; only the collision rule "candidate already marked -> take j" is sequential, walked
// with readlane over an LDS byte map (same picks, same output as a one-lane loop).  All lanes then write the N
// coefficients, two per 16-byte store.
__global__ void __launch_bounds__(256)
sample_challenge_kernel(int64_t* __restrict__ out, uint64_t npoly, uint32_t n_ring, uint64_t seed, uint32_t stream,
                        uint32_t kappa) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int8_t* mark = reinterpret_cast<int8_t*>(smem) + (size_t)wave * n_ring;
  uint32_t* mark_w = reinterpret_cast<uint32_t*>(mark);   // n_ring is a multiple of 4
  const uint32_t kap = kappa < n_ring ? kappa : n_ring;
  const bool pair16 = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  for (uint64_t poly = (uint64_t)blockIdx.x * 4 + wave; poly < npoly; poly += (uint64_t)gridDim.x * 4) {
    for (uint32_t i = lane; i < n_ring / 4; i += 64) mark_w[i] = 0;
    wave_sync();
    for (uint32_t t0 = 0; t0 < kap; t0 += 64) {
      const uint32_t t = t0 + lane;
      const Philox4 r = sampler_block(seed, stream, poly, t >> 1);
      const uint32_t w0 = (t & 1) ? r.v[2] : r.v[0], w1 = (t & 1) ? r.v[3] : r.v[1];
      const uint32_t j = n_ring - kap + t;                               // (lanes beyond kap: unused)
      const uint32_t pick = uniform_below(w0, w1 & ~1u, j + 1);
      const int32_t sign = (w1 & 1u) ? 1 : -1;                           // random_bool(0.5): +1 / -1
      const uint32_t m = kap - t0 < 64u ? kap - t0 : 64u;
#pragma unroll 1
      for (uint32_t e = 0; e < m; ++e) {
        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)pick, (int)e);
        const uint32_t jj = (uint32_t)__builtin_amdgcn_readlane((int)j, (int)e);
        const int32_t sg = __builtin_amdgcn_readlane(sign, (int)e);
        const uint32_t pos = mark[pk] ? jj : pk;
        wave_sync();
        if (lane == 0) mark[pos] = (int8_t)sg;
        wave_sync();
      }
    }
    int64_t* dst = out + poly * n_ring;
    if (pair16) {
      for (uint32_t i = 2 * lane; i < n_ring; i += 128) {
        const int32_t a0 = mark[i], a1 = mark[i + 1];
        st_stream(reinterpret_cast<int4*>(dst + i), make_int4(a0, a0 >> 31, a1, a1 >> 31));
      }
    } else {
      for (uint32_t i = lane; i < n_ring; i += 64) dst[i] = (int64_t)mark[i];
    }
    wave_sync();
  }
}

// =============================================================================================
// Launchers
// =============================================================================================
static inline unsigned grid_for(uint64_t tasks, int num_cus, int waves_per_block = 4, int blocks_per_cu = 8) {
  uint64_t blocks = (tasks + waves_per_block - 1) / waves_per_block;
  const uint64_t cap = (uint64_t)num_cus * blocks_per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

#define RZK_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e_ = hipGetLastError();          \
    if (e_ != hipSuccess) return (int)e_;       \
  } while (0)

size_t row_scratch_words(int logn, int num_cus) { return (size_t)num_cus * 8 * 4 * (((size_t)kScratchLines << logn) + 16); }

// LDS words of one team in unit_kernel / row_kernel: transposition slab + one N-word buffer; a rotation term's image
// (2N words) fits inside that, the non-zero list of a two-wavefront team comes on top
template <int LOGN, class TM, bool HAS_SHIFT>
constexpr size_t team_lds_words() {
  using G = Geo<LOGN, TM::LL>;
  constexpr size_t base = G::LDS_WORDS + G::N;
  constexpr size_t rot = (HAS_SHIFT && TM::LL != 6) ? 2 * (size_t)G::N + kShiftListWords : 0;
  static_assert(!(HAS_SHIFT && TM::LL != 6) || TM::kTeamsPerBlock == 1, "the list lies behind the team's own buffers");
  return (((base > rot ? base : rot) + 3) / 4) * 4;
}

template <int LOGN, bool HAS_VEC, bool HAS_SHIFT, class TM = WaveTeam>
static int launch_units_t(const LaunchCfg& cfg, const Program* d_prog, const WaveProgram* d_wp, const Operands& ops,
                          const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T, const uint32_t* d_tw,
                          uint32_t* d_scratch, uint8_t* d_flags, uint32_t ntasks, uint32_t upt, uint32_t tpe, uint32_t wpt) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int TPB = TM::kTeamsPerBlock;
  // per team: transposition slab + P
  const size_t lds = TPB * team_lds_words<LOGN, TM, HAS_SHIFT>() * sizeof(uint32_t);
  if (lds > 48 * 1024) {   // large dynamic LDS needs an opt-in; per device, so set before every launch (cheap, idempotent)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&unit_kernel<LOGN, HAS_VEC, HAS_SHIFT, TM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  // scratch sizing: at most num_cus * 32 team lines; every team walks its tasks with a grid stride
  const unsigned grid = grid_for(ntasks, cfg.num_cus, TPB, TM::LL == 6 ? 32 / TPB : 8);
  hipLaunchKernelGGL((unit_kernel<LOGN, HAS_VEC, HAS_SHIFT, TM>), dim3(grid), dim3(TPB << TM::LL), lds, (hipStream_t)cfg.stream,
                     d_prog, d_wp, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks, upt, tpe, wpt);
  RZK_LAUNCH_CHECK();
  return 0;
}

template <int LOGN, bool HAS_SHIFT, class TM = WaveTeam>
static int launch_units_io_t(const LaunchCfg& cfg, const Program* d_prog, const WaveProgram* d_wp, const Operands& ops,
                             const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T, const uint32_t* d_tw,
                             uint32_t* d_scratch, uint8_t* d_flags, uint32_t ntasks, uint32_t upt, uint32_t tpe, uint32_t wpt) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int TPB = TM::kTeamsPerBlock;
  const size_t lds = TPB * (size_t)IoCfg<LOGN, TM::LL>::WORDS * sizeof(uint32_t);   // per team: slab + parking buffers
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&unit_io_kernel<LOGN, HAS_SHIFT, TM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = grid_for(ntasks, cfg.num_cus, TPB, TM::LL == 6 ? 32 / TPB : 8);
  hipLaunchKernelGGL((unit_io_kernel<LOGN, HAS_SHIFT, TM>), dim3(grid), dim3(TPB << TM::LL), lds, (hipStream_t)cfg.stream,
                     d_prog, d_wp, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks, upt, tpe, wpt);
  RZK_LAUNCH_CHECK();
  return 0;
}

template <int LOGN, bool HAS_SHIFT, class TM = WaveTeam, bool DD = false>
static int launch_rows_t(const LaunchCfg& cfg, const Program* d_prog, const Operands& ops, const uint32_t* d_key_ntt,
                         const double* d_key_l2, const DevTables* T, const uint32_t* d_tw, uint32_t* d_scratch,
                         uint8_t* d_flags, uint32_t ntasks) {
  using G = Geo<LOGN, TM::LL>;
  constexpr int TPB = TM::kTeamsPerBlock;
  const size_t lds = TPB * team_lds_words<LOGN, TM, HAS_SHIFT>() * sizeof(uint32_t);   // per team: transposition slab + state word A
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&row_kernel<LOGN, HAS_SHIFT, TM, DD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = grid_for(ntasks, cfg.num_cus, TPB, TM::LL == 6 ? 32 / TPB : 8);   // <= 32 team lines per CU (scratch sizing)
  hipLaunchKernelGGL((row_kernel<LOGN, HAS_SHIFT, TM, DD>), dim3(grid), dim3(TPB << TM::LL), lds, (hipStream_t)cfg.stream, d_prog, ops,
                     d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_rows(int logn, const LaunchCfg& cfg, const Program* d_prog, uint32_t nrows, bool has_shift, const Operands& ops,
                const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T, const uint32_t* d_tw,
                uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch, bool has_dd) {
  if (batch == 0 || nrows == 0) return 0;
  if (batch * nrows >= (1ull << 32)) return -2;
  const uint32_t ntasks = (uint32_t)(batch * nrows);
#define RZK_ROWS_ARGS cfg, d_prog, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks
  if (has_dd && !has_shift) {   // rows whose operand transforms may come from the call's operand images (TERM_DD)
    switch (logn) {
      case 9: return launch_rows_t<9, false, WaveTeam, true>(RZK_ROWS_ARGS);
      case 10: return launch_rows_t<10, false, WaveTeam, true>(RZK_ROWS_ARGS);
      case 11: return cfg.pair_poly ? launch_rows_t<11, false, PairTeam, true>(RZK_ROWS_ARGS) : launch_rows_t<11, false, WaveTeam, true>(RZK_ROWS_ARGS);
    }
    return -1;
  }
  switch (logn) {
    case 9: return has_shift ? launch_rows_t<9, true>(RZK_ROWS_ARGS) : launch_rows_t<9, false>(RZK_ROWS_ARGS);
    case 10: return has_shift ? launch_rows_t<10, true>(RZK_ROWS_ARGS) : launch_rows_t<10, false>(RZK_ROWS_ARGS);
    case 11:   // rotation terms at N = 2048: teams of two only (rzk_api.cpp, shift_ok)
      if (has_shift) return cfg.pair_poly ? launch_rows_t<11, true, PairTeam>(RZK_ROWS_ARGS) : -1;
      return cfg.pair_poly ? launch_rows_t<11, false, PairTeam>(RZK_ROWS_ARGS) : launch_rows_t<11, false>(RZK_ROWS_ARGS);
  }
#undef RZK_ROWS_ARGS
  return -1;
}

int launch_units(int logn, const LaunchCfg& cfg, const Program* d_prog, const WaveProgram* d_wp, uint32_t nunits,
                 uint32_t units_per_task, uint32_t work_per_entry, bool has_vec, bool has_shift, const Operands& ops,
                 const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T, const uint32_t* d_tw,
                 uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch) {
  if (batch == 0 || nunits == 0) return 0;
  if (units_per_task == 0) units_per_task = 1;
  const uint32_t tpe = (nunits + units_per_task - 1) / units_per_task;   // tasks per batch entry
  if (batch * tpe >= (1ull << 32)) return -2;   // task index is 32-bit
  const uint32_t ntasks = (uint32_t)(batch * tpe);
  const uint32_t wpt = (work_per_entry + tpe - 1) / tpe;   // transforms per task (estimate, for the progress priorities)
#define RZK_UNIT_ARGS cfg, d_prog, d_wp, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks, units_per_task, tpe, wpt
  if (!has_vec && cfg.unit_io && !(logn == 11 && has_shift)) {   // key-product programs: every operand read once (unit_io_kernel)
    switch (logn) {
      case 9: return has_shift ? launch_units_io_t<9, true>(RZK_UNIT_ARGS) : launch_units_io_t<9, false>(RZK_UNIT_ARGS);
      case 10: return has_shift ? launch_units_io_t<10, true>(RZK_UNIT_ARGS) : launch_units_io_t<10, false>(RZK_UNIT_ARGS);
      case 11:
        if (has_shift) return -1;
        return cfg.pair_poly ? launch_units_io_t<11, false, PairTeam>(RZK_UNIT_ARGS) : launch_units_io_t<11, false>(RZK_UNIT_ARGS);
    }
    return -1;
  }
#define RZK_UNIT_CASE(L)                                                                                            \
  case L:                                                                                                           \
    if (has_shift)                                                                                                  \
      return has_vec ? launch_units_t<L, true, true>(RZK_UNIT_ARGS) : launch_units_t<L, false, true>(RZK_UNIT_ARGS); \
    return has_vec ? launch_units_t<L, true, false>(RZK_UNIT_ARGS) : launch_units_t<L, false, false>(RZK_UNIT_ARGS);
  switch (logn) {
    RZK_UNIT_CASE(9)
    RZK_UNIT_CASE(10)
    case 11:   // rotation terms at N = 2048: teams of two only (rzk_api.cpp, shift_ok)
      if (has_shift) {
        if (!cfg.pair_poly) return -1;
        return has_vec ? launch_units_t<11, true, true, PairTeam>(RZK_UNIT_ARGS) : launch_units_t<11, false, true, PairTeam>(RZK_UNIT_ARGS);
      }
      if (cfg.pair_poly)
        return has_vec ? launch_units_t<11, true, false, PairTeam>(RZK_UNIT_ARGS) : launch_units_t<11, false, false, PairTeam>(RZK_UNIT_ARGS);
      return has_vec ? launch_units_t<11, true, false>(RZK_UNIT_ARGS) : launch_units_t<11, false, false>(RZK_UNIT_ARGS);
  }
#undef RZK_UNIT_ARGS
#undef RZK_UNIT_CASE
  return -1;
}

template <int LOGN, class TM = WaveTeam>
static int launch_shift_t(const LaunchCfg& cfg, const Program* d_prog, const Operands& ops, const DevTables* T,
                          uint8_t* d_flags, uint32_t ntasks) {
  constexpr int TPB = ShiftCfg<LOGN, TM>::TPB;
  // One team's image is 8 N bytes.  The workgroup asks for at least 40 KiB so that a CU holds four workgroups = 4 waves per
  // SIMD: with the 79 VGPRs the kernel needs, five or six would fit, and measured slower (response rows at N = 1024:
  // 86.7 us against 77.5 us at four — the kernel is co-bound by the LDS pipe, more waves only add contention).
  // Teams of two (N = 2048): 18 KiB per 128-thread workgroup, eight workgroups = 4 waves per SIMD.
  size_t lds = (size_t)TPB * ShiftCfg<LOGN, TM>::WORDS * sizeof(uint32_t);
  if (TM::LL == 6 && LOGN >= 10 && lds < 40 * 1024) lds = 40 * 1024;   // (N = 512 keeps its 6 waves per SIMD: 4-KiB images, measured fine in round 2)
#ifdef RZK_SHIFT_PAIR_LDS_KB   // experiment: fewer pairs per CU at N = 2048 (18 KiB = 8 pairs, 22 = 7, 26 = 6)
  if (TM::LL == 7 && lds < (size_t)RZK_SHIFT_PAIR_LDS_KB * 1024) lds = (size_t)RZK_SHIFT_PAIR_LDS_KB * 1024;
#endif
  if (lds > 48 * 1024) {
    hipError_t e = ops.trusted ? hipFuncSetAttribute(reinterpret_cast<const void*>(&shift_row_kernel<LOGN, true, TM>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                               : hipFuncSetAttribute(reinterpret_cast<const void*>(&shift_row_kernel<LOGN, false, TM>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = grid_for(ntasks, cfg.num_cus, TPB, TM::LL == 6 ? 16 : 64);
  if (ops.trusted)
    hipLaunchKernelGGL((shift_row_kernel<LOGN, true, TM>), dim3(grid), dim3(TPB << TM::LL), lds, (hipStream_t)cfg.stream, d_prog, ops,
                       T, d_flags, ntasks);
  else
    hipLaunchKernelGGL((shift_row_kernel<LOGN, false, TM>), dim3(grid), dim3(TPB << TM::LL), lds, (hipStream_t)cfg.stream, d_prog, ops,
                       T, d_flags, ntasks);
  RZK_LAUNCH_CHECK();
  return 0;
}

template <int LOGN>
static int launch_dkey_t(const LaunchCfg& cfg, const int64_t* g, uint64_t count, uint32_t dkey_n, uint32_t* img, double* l2,
                         const DevTables* T, const uint32_t* d_tw, uint8_t* d_flags, uint32_t* d_bad, bool two_bit, bool trusted) {
  using G = Geo<LOGN>;
  hipLaunchKernelGGL(dkey_transform_kernel<LOGN>, dim3(grid_for(count, cfg.num_cus)), dim3(256), 4 * G::LDS_WORDS * sizeof(uint32_t),
                     (hipStream_t)cfg.stream, g, count, dkey_n, img, l2, T, d_tw, d_flags, d_bad, two_bit ? 1u : 0u,
                     trusted ? 1u : 0u);
  RZK_LAUNCH_CHECK();
  return 0;
}
int launch_dkey_transform(int logn, const LaunchCfg& cfg, const int64_t* g, uint64_t count, uint32_t dkey_n, uint32_t* img,
                          double* l2, const DevTables* T, const uint32_t* d_tw, uint8_t* d_flags, uint32_t* d_bad, bool two_bit,
                          bool trusted) {
  if (count == 0) return 0;
  switch (logn) {
    case 9: return launch_dkey_t<9>(cfg, g, count, dkey_n, img, l2, T, d_tw, d_flags, d_bad, two_bit, trusted);
    case 10: return launch_dkey_t<10>(cfg, g, count, dkey_n, img, l2, T, d_tw, d_flags, d_bad, two_bit, trusted);
    case 11: return launch_dkey_t<11>(cfg, g, count, dkey_n, img, l2, T, d_tw, d_flags, d_bad, two_bit, trusted);
  }
  return -1;
}

int launch_shift_rows(int logn, const LaunchCfg& cfg, const Program* d_prog, uint32_t nrows, const Operands& ops,
                      const DevTables* T, uint8_t* d_flags, uint64_t batch) {
  if (batch == 0 || nrows == 0) return 0;
  if (batch * nrows >= (1ull << 32)) return -2;
  const uint32_t ntasks = (uint32_t)(batch * nrows);
  switch (logn) {
    case 9: return launch_shift_t<9>(cfg, d_prog, ops, T, d_flags, ntasks);
    case 10: return launch_shift_t<10>(cfg, d_prog, ops, T, d_flags, ntasks);
    case 11: return cfg.pair_poly ? launch_shift_t<11, PairTeam>(cfg, d_prog, ops, T, d_flags, ntasks) : -1;   // (rzk_api.cpp, shift_ok)
  }
  return -1;
}

size_t group_scratch_words(int logn, int num_cus) {
  return (size_t)num_cus * 8 * 4 * (size_t)(2 * kGroupMax) * ((size_t)1 << logn);
}

template <int LOGN>
static int launch_groups_t(const LaunchCfg& cfg, const Program* d_prog, const Operands& ops, const uint32_t* d_key_ntt,
                           const double* d_key_l2, const DevTables* T, const uint32_t* d_tw, uint32_t* d_scratch,
                           uint8_t* d_flags, uint32_t ntasks) {
  using G = Geo<LOGN>;
  constexpr int GM = LOGN >= 11 ? 2 : RZK_GROUP_GM;   // accumulators per wave (N = 2048 is never grouped by the host)
  hipLaunchKernelGGL((row_group_kernel<LOGN, GM>), dim3(grid_for(ntasks, cfg.num_cus)), dim3(256),
                     4 * G::LDS_WORDS * sizeof(uint32_t), (hipStream_t)cfg.stream, d_prog, ops, d_key_ntt, d_key_l2, T,
                     d_tw, d_scratch, d_flags, ntasks);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_row_groups(int logn, const LaunchCfg& cfg, const Program* d_prog, uint32_t ngroups, const Operands& ops,
                      const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T, const uint32_t* d_tw,
                      uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch) {
  if (batch == 0 || ngroups == 0) return 0;
  if (batch * ngroups >= (1ull << 32)) return -2;
  const uint32_t ntasks = (uint32_t)(batch * ngroups);
  switch (logn) {
    case 9: return launch_groups_t<9>(cfg, d_prog, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
    case 10: return launch_groups_t<10>(cfg, d_prog, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
    case 11: return launch_groups_t<11>(cfg, d_prog, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
  }
  return -1;
}

size_t block_scratch_words(int logn, int num_cus) {
  return (size_t)num_cus * 2 * (size_t)(2 * kBlockMaxRows) * ((size_t)1 << logn);
}

template <int LOGN, class TM = WaveTeam>
static int launch_blocks_t(const LaunchCfg& cfg, const Program* d_prog, const BlockPlan* d_plan, const Operands& ops,
                           const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T, const uint32_t* d_tw,
                           uint32_t* d_scratch, uint8_t* d_flags, uint32_t ntasks) {
  using G = Geo<LOGN, TM::LL>;
  const size_t lds = ((size_t)kBlockMaxSlots * G::N + (size_t)kBlockWaves * G::LDS_WORDS) * sizeof(uint32_t) +
                     kBlockMaxSlots * sizeof(double);
  // > 64 KiB of dynamic LDS needs an explicit opt-in.  The attribute is kept per device and contexts may live on
  // several devices / host threads, so it is set (idempotently, a host-side call of ~1 us) before every launch
  // rather than behind a process-wide "done" flag.
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&row_block_kernel<LOGN, TM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  uint32_t grid = ntasks < (uint32_t)cfg.num_cus * 2 ? ntasks : (uint32_t)cfg.num_cus * 2;   // scratch: num_cus * 2 lines
  hipLaunchKernelGGL((row_block_kernel<LOGN, TM>), dim3(grid), dim3(kBlockWaves << TM::LL), lds, (hipStream_t)cfg.stream, d_prog,
                     d_plan, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_row_blocks(int logn, const LaunchCfg& cfg, const Program* d_prog, const BlockPlan* d_plan, uint32_t nblocks,
                      const Operands& ops, const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T,
                      const uint32_t* d_tw, uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch) {
  if (batch == 0 || nblocks == 0) return 0;
  if (batch * nblocks >= (1ull << 32)) return -2;
  const uint32_t ntasks = (uint32_t)(batch * nblocks);
  switch (logn) {
    case 10: return launch_blocks_t<10>(cfg, d_prog, d_plan, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
    case 11:
      if (cfg.pair_poly)   // eight two-wavefront teams: 16 coefficients per thread, 4 waves per SIMD beside 138 KiB of LDS
        return launch_blocks_t<11, BlockPairTeam>(cfg, d_prog, d_plan, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
      return launch_blocks_t<11>(cfg, d_prog, d_plan, ops, d_key_ntt, d_key_l2, T, d_tw, d_scratch, d_flags, ntasks);
  }
  return -1;
}

template <int LOGN>
static int launch_slots_t(const LaunchCfg& cfg, const Program* d_prog, const SlotTable* d_slots, uint32_t nslots,
                          const Operands& ops, const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* T,
                          const uint32_t* d_tw, uint32_t* d_ws, double* d_norms, uint32_t* d_scratch, uint8_t* d_flags,
                          uint32_t batch, uint32_t np_store) {
  using G = Geo<LOGN>;
  const uint32_t ftasks = batch * nslots;
  hipLaunchKernelGGL(fwd_slots_kernel<LOGN>, dim3(grid_for(ftasks, cfg.num_cus)), dim3(256),
                     4 * G::LDS_WORDS * sizeof(uint32_t), (hipStream_t)cfg.stream, d_slots, ops, T, d_tw, d_ws, d_norms,
                     d_flags, ftasks, np_store);
  RZK_LAUNCH_CHECK();
  unsigned grid = (unsigned)cfg.num_cus * 8;   // multiple of 8: the XCD-class dealing needs whole classes
  grid -= grid % 8;
  hipLaunchKernelGGL(row_slots_kernel<LOGN>, dim3(grid), dim3(256), 4 * (G::LDS_WORDS + G::N) * sizeof(uint32_t),
                     (hipStream_t)cfg.stream, d_prog, d_slots, ops, d_key_ntt, d_key_l2, T, d_tw, d_ws, d_norms,
                     d_scratch, d_flags, batch, np_store);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_row_program_slots(int logn, const LaunchCfg& cfg, const Program* d_prog, const SlotTable* d_slots,
                             uint32_t nslots, const Operands& ops, const uint32_t* d_key_ntt,
                             const double* d_key_l2, const DevTables* T, const uint32_t* d_tw, uint32_t* d_ws,
                             double* d_norms, uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch,
                             uint32_t np_store) {
  if (batch == 0) return 0;
  if (batch * nslots >= (1ull << 32) || batch >= (1ull << 31)) return -2;
  switch (logn) {
    case 9: return launch_slots_t<9>(cfg, d_prog, d_slots, nslots, ops, d_key_ntt, d_key_l2, T, d_tw, d_ws, d_norms, d_scratch, d_flags, (uint32_t)batch, np_store);
    case 10: return launch_slots_t<10>(cfg, d_prog, d_slots, nslots, ops, d_key_ntt, d_key_l2, T, d_tw, d_ws, d_norms, d_scratch, d_flags, (uint32_t)batch, np_store);
    case 11: return launch_slots_t<11>(cfg, d_prog, d_slots, nslots, ops, d_key_ntt, d_key_l2, T, d_tw, d_ws, d_norms, d_scratch, d_flags, (uint32_t)batch, np_store);
  }
  return -1;
}

template <int LOGN>
static int launch_key_t(const LaunchCfg& cfg, const int64_t* d_key, uint32_t entries, uint32_t* d_key_ntt,
                        const DevTables* T, const uint32_t* d_tw) {
  using G = Geo<LOGN>;
  const size_t lds = 4 * G::LDS_WORDS * sizeof(uint32_t);
  const unsigned grid = grid_for((uint64_t)entries * kKeyImages, cfg.num_cus);
  hipLaunchKernelGGL(key_transform_kernel<LOGN>, dim3(grid), dim3(256), lds, (hipStream_t)cfg.stream,
                     d_key, entries, d_key_ntt, T, d_tw);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_key_transform(int logn, const LaunchCfg& cfg, const int64_t* d_key, uint32_t entries,
                         uint32_t* d_key_ntt, const DevTables* T, const uint32_t* d_tw) {
  if (entries == 0) return 0;
  switch (logn) {
    case 9: return launch_key_t<9>(cfg, d_key, entries, d_key_ntt, T, d_tw);
    case 10: return launch_key_t<10>(cfg, d_key, entries, d_key_ntt, T, d_tw);
    case 11: return launch_key_t<11>(cfg, d_key, entries, d_key_ntt, T, d_tw);
  }
  return -1;
}

template <int LOGN>
static int launch_ntt_t(bool inverse, const LaunchCfg& cfg, int prime, const uint32_t* d_in, uint32_t* d_out,
                        uint64_t count, const DevTables* T, const uint32_t* d_tw) {
  using G = Geo<LOGN>;
  const size_t lds = 4 * G::LDS_WORDS * sizeof(uint32_t);
  const unsigned grid = grid_for(count, cfg.num_cus);
  if (inverse)
    hipLaunchKernelGGL(ntt_inv_kernel<LOGN>, dim3(grid), dim3(256), lds, (hipStream_t)cfg.stream, d_in,
                       d_out, count, prime, T, d_tw);
  else
    hipLaunchKernelGGL(ntt_fwd_kernel<LOGN>, dim3(grid), dim3(256), lds, (hipStream_t)cfg.stream, d_in,
                       d_out, count, prime, T, d_tw);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_ntt(int logn, bool inverse, const LaunchCfg& cfg, int prime, const uint32_t* d_in,
               uint32_t* d_out, uint64_t count, const DevTables* T, const uint32_t* d_tw) {
  if (count == 0) return 0;
  switch (logn) {
    case 9: return launch_ntt_t<9>(inverse, cfg, prime, d_in, d_out, count, T, d_tw);
    case 10: return launch_ntt_t<10>(inverse, cfg, prime, d_in, d_out, count, T, d_tw);
    case 11: return launch_ntt_t<11>(inverse, cfg, prime, d_in, d_out, count, T, d_tw);
  }
  return -1;
}

// flags[i] = value: one small launch (hipMemsetAsync's fill kernel takes ~4.5 us for 4 KiB on this stack)
__global__ void __launch_bounds__(256) fill_u8_kernel(uint8_t* __restrict__ p, uint8_t value, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = value;
}
int launch_fill_u8(const LaunchCfg& cfg, uint8_t* p, uint8_t value, uint64_t n) {
  if (n == 0) return 0;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)cfg.stream, p, value, n);
  RZK_LAUNCH_CHECK();
  return 0;
}

static inline uint32_t log2_u32(uint32_t v) {
  uint32_t l = 0;
  while ((1u << l) < v) ++l;
  return l;
}
int launch_sample_uniform(const LaunchCfg& cfg, int64_t* out, uint64_t npoly, uint32_t n_ring, uint64_t seed,
                          uint32_t stream, uint32_t bound) {
  if (npoly == 0) return 0;
  if (n_ring < 2 || (n_ring & (n_ring - 1))) return -1;
  const uint64_t ncoef = npoly * n_ring;
  hipLaunchKernelGGL(sample_uniform_kernel, dim3(grid_for((ncoef + 1) / 2, cfg.num_cus, 256, 16)), dim3(256), 0,
                     (hipStream_t)cfg.stream, out, ncoef, log2_u32(n_ring), seed, stream, bound);
  RZK_LAUNCH_CHECK();
  return 0;
}
int launch_sample_gauss(const LaunchCfg& cfg, int64_t* out, uint64_t npoly, uint32_t n_ring, uint64_t seed,
                        uint32_t stream, double sigma) {
  if (npoly == 0) return 0;
  if (n_ring < 2 || (n_ring & (n_ring - 1))) return -1;
  const uint64_t ncoef = npoly * n_ring;
  const dim3 grid(grid_for((ncoef + 1) / 2, cfg.num_cus, 256, 16));
  if (sigma < 524288.0)   // 9.4 sigma < 2^23: single precision carries every sample with an error far below 1
    hipLaunchKernelGGL(sample_gauss_kernel<true>, grid, dim3(256), 0, (hipStream_t)cfg.stream, out, ncoef, log2_u32(n_ring), seed,
                       stream, sigma);
  else
    hipLaunchKernelGGL(sample_gauss_kernel<false>, grid, dim3(256), 0, (hipStream_t)cfg.stream, out, ncoef, log2_u32(n_ring), seed,
                       stream, sigma);
  RZK_LAUNCH_CHECK();
  return 0;
}
int launch_sample_challenge(const LaunchCfg& cfg, int64_t* out, uint64_t npoly, uint32_t n_ring, uint64_t seed,
                            uint32_t stream, uint32_t kappa) {
  if (npoly == 0) return 0;
  hipLaunchKernelGGL(sample_challenge_kernel, dim3(grid_for(npoly, cfg.num_cus, 4, 16)), dim3(256), 4 * n_ring,
                     (hipStream_t)cfg.stream, out, npoly, n_ring, seed, stream, kappa);
  RZK_LAUNCH_CHECK();
  return 0;
}

// any int64 -> centred representative mod q (utility for data that does not come from a ZqI64: the hot kernels
// assume canonical inputs and read only the low word of every coefficient)
__global__ void __launch_bounds__(256) canonicalize_kernel(const int64_t* __restrict__ in, int64_t* __restrict__ out,
                                                          uint64_t ncoef, int64_t q) {
  const int64_t half = (q - 1) / 2;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < ncoef; i += (uint64_t)gridDim.x * 256) {
    int64_t r = in[i] % q;          // sign of the dividend, |r| < q
    if (r > half) r -= q;
    if (r < -half) r += q;
    out[i] = r;
  }
}
int launch_canonicalize(const LaunchCfg& cfg, const int64_t* in, int64_t* out, uint64_t ncoef, int64_t q) {
  if (ncoef == 0) return 0;
  uint64_t blocks = (ncoef + 255) / 256;
  const uint64_t cap = (uint64_t)cfg.num_cus * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(canonicalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)cfg.stream, in, out, ncoef, q);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_addsub(const LaunchCfg& cfg, bool sub, const int64_t* a, const int64_t* b, int64_t* out,
                  uint64_t ncoef, const DevTables* T, uint32_t* bad_word) {
  if (ncoef == 0) return 0;
  const uint64_t n2 = ncoef / 2;   // ncoef is a multiple of N >= 512
  uint64_t blocks = (n2 + 255) / 256;
  const uint64_t cap = (uint64_t)cfg.num_cus * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(addsub_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)cfg.stream, a, b,
                     out, n2, sub ? 1 : 0, T, bad_word);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_norm(int logn, const LaunchCfg& cfg, const int64_t* v, uint32_t rows, uint64_t limit_hi,
                uint64_t limit_lo, uint8_t* ok, uint64_t B, int and_mode, int shift, uint32_t qhalf,
                uint32_t* bad_word) {
  if (B == 0) return 0;
  const unsigned grid = grid_for(B, cfg.num_cus);
  switch (logn) {
    case 9:
      hipLaunchKernelGGL(norm_kernel<9>, dim3(grid), dim3(256), 0, (hipStream_t)cfg.stream, v, rows,
                         limit_hi, limit_lo, ok, B, and_mode, shift, qhalf, bad_word);
      break;
    case 10:
      hipLaunchKernelGGL(norm_kernel<10>, dim3(grid), dim3(256), 0, (hipStream_t)cfg.stream, v, rows,
                         limit_hi, limit_lo, ok, B, and_mode, shift, qhalf, bad_word);
      break;
    case 11:
      hipLaunchKernelGGL(norm_kernel<11>, dim3(grid), dim3(256), 0, (hipStream_t)cfg.stream, v, rows,
                         limit_hi, limit_lo, ok, B, and_mode, shift, qhalf, bad_word);
      break;
    default: return -1;
  }
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_eq(int logn, const LaunchCfg& cfg, const int64_t* a, const int64_t* b, uint32_t rows,
              uint8_t* eq, uint64_t B, uint32_t qhalf, uint32_t* bad_word) {
  if (B == 0) return 0;
  const unsigned grid = grid_for(B, cfg.num_cus);
  switch (logn) {
    case 9:
      hipLaunchKernelGGL(eq_kernel<9>, dim3(grid), dim3(256), 0, (hipStream_t)cfg.stream, a, b, rows, eq, B, qhalf,
                         bad_word);
      break;
    case 10:
      hipLaunchKernelGGL(eq_kernel<10>, dim3(grid), dim3(256), 0, (hipStream_t)cfg.stream, a, b, rows, eq, B, qhalf,
                         bad_word);
      break;
    case 11:
      hipLaunchKernelGGL(eq_kernel<11>, dim3(grid), dim3(256), 0, (hipStream_t)cfg.stream, a, b, rows, eq, B, qhalf,
                         bad_word);
      break;
    default: return -1;
  }
  RZK_LAUNCH_CHECK();
  return 0;
}

// ---- small ring degrees ------------------------------------------------------------------------------------------
int launch_row_program_small(uint32_t N, const LaunchCfg& cfg, const Program* d_prog, uint32_t nrows,
                             const Operands& ops, const uint32_t* d_key_mont, const DevTables* T, uint32_t r2q,
                             uint8_t* d_flags, uint64_t batch) {
  if (batch == 0 || nrows == 0) return 0;
  if (batch * nrows >= (1ull << 32)) return -2;
  const uint32_t ntasks = (uint32_t)(batch * nrows);
  const unsigned grid = grid_for(ntasks, cfg.num_cus);
  hipLaunchKernelGGL(row_kernel_small, dim3(grid), dim3(256), 4 * 2 * N * sizeof(uint32_t), (hipStream_t)cfg.stream,
                     d_prog, ops, d_key_mont, T, d_flags, ntasks, N, r2q);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_key_mont(const LaunchCfg& cfg, const int64_t* d_key, uint32_t* d_key_mont, uint64_t ncoef,
                    const DevTables* T, uint32_t r2q) {
  if (ncoef == 0) return 0;
  uint64_t blocks = (ncoef + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(key_mont_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)cfg.stream, d_key,
                     d_key_mont, ncoef, T, r2q);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_norm_small(uint32_t N, const LaunchCfg& cfg, const int64_t* v, uint32_t rows, uint64_t limit_hi,
                      uint64_t limit_lo, uint8_t* ok, uint64_t B, int and_mode, int shift, uint32_t qhalf,
                      uint32_t* bad_word) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(norm_kernel_small, dim3(grid_for(B, cfg.num_cus)), dim3(256), 0, (hipStream_t)cfg.stream, v,
                     rows, limit_hi, limit_lo, ok, B, and_mode, shift, N, qhalf, bad_word);
  RZK_LAUNCH_CHECK();
  return 0;
}

int launch_eq_small(uint32_t N, const LaunchCfg& cfg, const int64_t* a, const int64_t* b, uint32_t rows,
                    uint8_t* eq, uint64_t B, uint32_t qhalf, uint32_t* bad_word) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(eq_kernel_small, dim3(grid_for(B, cfg.num_cus)), dim3(256), 0, (hipStream_t)cfg.stream, a, b,
                     rows, eq, B, N, qhalf, bad_word);
  RZK_LAUNCH_CHECK();
  return 0;
}

}  // namespace rzk
