// VALU instruction-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: SURVEY.md §7.3 notes that v_mul_hi_u32 / v_mad_u64_u32 throughput on gfx950 is not
// documented in the guides.  The modular-butterfly design (32-bit Montgomery lanes vs 24-bit
// lanes vs fp64-FMA lanes) depends on it, so measure it before writing the NTT kernels.
//
// Each test: grid = 256 CUs x 8 blocks x 256 threads (8 waves/SIMD), every thread runs ITER
// iterations of UNROLL x 8 independent chains of one instruction.  Reported: wave-instructions
// per ns per SIMD and "cycles per wave-instruction per SIMD" at the nominal 2.4 GHz clock, plus
// the ratio to v_add_u32.
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.cpp -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, \
              __LINE__);                                                           \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int ITER = 2000;

#define CHAINS8(OP)                                                       \
  OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

// ---- single-instruction kernels (inline asm so the instruction is exactly what is named) ----
#define DEF_KERNEL_U32(NAME, ASM)                                                        \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t seed) {        \
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;                                  \
    uint32_t a0 = t + seed, a1 = t * 3 + 1, a2 = t * 5 + 2, a3 = t * 7 + 3,              \
             a4 = t * 11 + 4, a5 = t * 13 + 5, a6 = t * 17 + 6, a7 = t * 19 + 7;         \
    uint32_t b = (t | 1u) * 2654435761u;                                                 \
    for (int i = 0; i < ITER; ++i) {                                                     \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                    \
        asm volatile(ASM : "+v"(a0) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a1) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a2) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a3) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a4) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a5) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a6) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a7) : "v"(b));                                           \
      }                                                                                  \
    }                                                                                    \
    out[t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                      \
  }

DEF_KERNEL_U32(add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL_U32(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL_U32(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL_U32(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL_U32(mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEF_KERNEL_U32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
DEF_KERNEL_U32(fma_f32, "v_fma_f32 %0, %0, %1, %0")
DEF_KERNEL_U32(min_u32, "v_min_u32 %0, %0, %1")
DEF_KERNEL_U32(sub_u32, "v_sub_u32 %0, %0, %1")
DEF_KERNEL_U32(alignbit, "v_alignbit_b32 %0, %0, %1, 8")
DEF_KERNEL_U32(mov_dpp_xor1, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEF_KERNEL_U32(add3_u32, "v_add3_u32 %0, %0, %1, %1")
DEF_KERNEL_U32(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")

// 64-bit destination kernels.
#define DEF_KERNEL_U64(NAME, ASM)                                                        \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t seed) {        \
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;                                  \
    uint64_t a0 = t + seed, a1 = t * 3 + 1, a2 = t * 5 + 2, a3 = t * 7 + 3,              \
             a4 = t * 11 + 4, a5 = t * 13 + 5, a6 = t * 17 + 6, a7 = t * 19 + 7;         \
    uint32_t b = (t | 1u) * 2654435761u;                                                 \
    for (int i = 0; i < ITER; ++i) {                                                     \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                    \
        asm volatile(ASM : "+v"(a0) : "v"(b) : "vcc");                                   \
        asm volatile(ASM : "+v"(a1) : "v"(b) : "vcc");                                   \
        asm volatile(ASM : "+v"(a2) : "v"(b) : "vcc");                                   \
        asm volatile(ASM : "+v"(a3) : "v"(b) : "vcc");                                   \
        asm volatile(ASM : "+v"(a4) : "v"(b) : "vcc");                                   \
        asm volatile(ASM : "+v"(a5) : "v"(b) : "vcc");                                   \
        asm volatile(ASM : "+v"(a6) : "v"(b) : "vcc");                                   \
        asm volatile(ASM : "+v"(a7) : "v"(b) : "vcc");                                   \
      }                                                                                  \
    }                                                                                    \
    uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                  \
    out[t] = (uint32_t)r ^ (uint32_t)(r >> 32);                                          \
  }

// D.u64 = S0.u32 * S1.u32 + S2.u64 ; chain through the 64-bit addend.
DEF_KERNEL_U64(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
DEF_KERNEL_U64(mad_i64_i32, "v_mad_i64_i32 %0, vcc, %1, %1, %0")

// fp64 kernels
#define DEF_KERNEL_F64(NAME, ASM)                                                        \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t seed) {        \
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;                                  \
    double a0 = 1.0 + t * 1e-9 + seed, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, \
           a6 = 1.6, a7 = 1.7;                                                           \
    double b = 1.0 + 1e-12 * (t & 255);                                                  \
    for (int i = 0; i < ITER; ++i) {                                                     \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                    \
        asm volatile(ASM : "+v"(a0) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a1) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a2) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a3) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a4) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a5) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a6) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a7) : "v"(b));                                           \
      }                                                                                  \
    }                                                                                    \
    double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                    \
    out[t] = (uint32_t)__double_as_longlong(r);                                          \
  }

DEF_KERNEL_F64(fma_f64, "v_fma_f64 %0, %0, %1, %0")
DEF_KERNEL_F64(mul_f64, "v_mul_f64 %0, %0, %1")
DEF_KERNEL_F64(add_f64, "v_add_f64 %0, %0, %1")
DEF_KERNEL_F64(rndne_f64, "v_rndne_f64 %0, %0")

// packed single precision: two FP32 lanes per 64-bit register pair (what a floating-point modular arithmetic over
// ~22-bit primes would issue)
#define DEF_KERNEL_PK32(NAME, ASM)                                                       \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t seed) {        \
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;                                  \
    double a0 = 1.0 + t * 1e-9 + seed, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, \
           a6 = 1.6, a7 = 1.7;   /* bit patterns only: each double register pair is two floats to the instruction */ \
    double b = 1.0 + 1e-12 * (t & 255);                                                  \
    for (int i = 0; i < ITER; ++i) {                                                     \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                    \
        asm volatile(ASM : "+v"(a0) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a1) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a2) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a3) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a4) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a5) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a6) : "v"(b));                                           \
        asm volatile(ASM : "+v"(a7) : "v"(b));                                           \
      }                                                                                  \
    }                                                                                    \
    double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                    \
    out[t] = (uint32_t)__double_as_longlong(r);                                          \
  }
DEF_KERNEL_PK32(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %0")
DEF_KERNEL_PK32(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEF_KERNEL_PK32(pk_add_f32, "v_pk_add_f32 %0, %0, %1")

// ---- composite butterflies written in plain C++ (what the compiler makes of them) ----
// Montgomery (R = 2^32) via one 64-bit mad: x*w -> t ; m = lo(t)*pinv ; r = hi(t + m*p) in [0,2p)
__device__ __forceinline__ uint32_t mont_mul_lazy(uint32_t x, uint32_t w, uint32_t p, uint32_t npinv) {
  uint64_t t = (uint64_t)x * w;
  uint32_t m = (uint32_t)t * npinv;  // npinv = -p^{-1} mod 2^32
  uint64_t u = (uint64_t)m * p + t;
  return (uint32_t)(u >> 32);
}
// Shoup: wp = floor(w * 2^32 / p); r = x*w - hi(x*wp)*p  in [0, 2p)
__device__ __forceinline__ uint32_t shoup_mul_lazy(uint32_t x, uint32_t w, uint32_t wp, uint32_t p) {
  uint32_t q = __umulhi(x, wp);
  return x * w - q * p;
}

// Harvey lazy CT butterfly on values in [0,4p): X' = X + W*Y, Y' = X - W*Y + 2p
__global__ void __launch_bounds__(256) k_bfly_mont(uint32_t* out, uint32_t seed) {
  const uint32_t p = 1073692673u;  // < 2^30, = 1 mod 8192
  const uint32_t npinv = seed | 1u;  // not the real constant; timing only
  const uint32_t twop = 2 * p;
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t x[8], y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = t * (2 * j + 3); y[j] = t * (2 * j + 5) + seed; }
  uint32_t w = (t | 1u) * 2654435761u >> 2;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t xx = x[j];
      uint32_t d = xx - twop;
      xx = d < xx ? d : xx;  // min(xx, xx-2p) unsigned
      uint32_t v = mont_mul_lazy(y[j], w, p, npinv);
      x[j] = xx + v;
      y[j] = xx - v + twop;
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= x[j] ^ y[j];
  out[t] = r;
}

__global__ void __launch_bounds__(256) k_bfly_shoup(uint32_t* out, uint32_t seed) {
  const uint32_t p = 1073692673u;
  const uint32_t twop = 2 * p;
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t x[8], y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = t * (2 * j + 3); y[j] = t * (2 * j + 5) + seed; }
  uint32_t w = (t | 1u) * 2654435761u >> 2;
  uint32_t wp = w * 4 + seed;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t xx = x[j];
      uint32_t d = xx - twop;
      xx = d < xx ? d : xx;
      uint32_t v = shoup_mul_lazy(y[j], w, wp, p);
      x[j] = xx + v;
      y[j] = xx - v + twop;
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= x[j] ^ y[j];
  out[t] = r;
}

// fp64 butterfly: p < 2^50, values are integers held in doubles in (-p, p) (balanced, lazy).
// v = y*w mod p : h = y*w ; l = fma(y,w,-h) ; q = rint(y*wop) ; v = fma(-q,p,h) + l
__global__ void __launch_bounds__(256) k_bfly_f64(uint32_t* out, uint32_t seed) {
  const double p = 1125899906826241.0;  // ~2^50
  const double pinv = 1.0 / p;
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  double x[8], y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = (double)(t * (2 * j + 3)); y[j] = (double)(t * (2 * j + 5) + seed); }
  double w = (double)((t | 1u) * 2654435761u) * 1024.0;
  double wop = w * pinv;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double h = y[j] * w;
      double l = __fma_rn(y[j], w, -h);
      double q = __builtin_rint(y[j] * wop);
      double v = __fma_rn(-q, p, h) + l;
      // lazy reduce x into (-p,p): x -= p*rint(x*pinv)
      double xx = x[j];
      xx = __fma_rn(-__builtin_rint(xx * pinv), p, xx);
      x[j] = xx + v;
      y[j] = xx - v;
    }
  }
  double r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r += x[j] + y[j];
  out[t] = (uint32_t)__double_as_longlong(r);
}

// 24-bit Shoup-style butterfly: p < 2^23, values in [0,2p) ; all multiplies are u24 (full-rate?)
__global__ void __launch_bounds__(256) k_bfly_u24(uint32_t* out, uint32_t seed) {
  const uint32_t p = 8380417u;  // 2^23 - 2^13 + 1
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t x[8], y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = (t * (2 * j + 3)) & 0xffffff; y[j] = (t * (2 * j + 5) + seed) & 0xffffff; }
  uint32_t w = ((t | 1u) * 2654435761u) >> 9;
  uint32_t wp = (w * 2 + seed) & 0xffffff;  // floor(w*2^24/p), timing only
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // q = (y*wp) >> 24 : bits 24..47 of the 48-bit product
      uint32_t lo = __umul24(y[j], wp);
      uint32_t hi = __umulhi(y[j] & 0xffffff, wp & 0xffffff);  // compiler may pick mul_hi_u32_u24
      uint32_t q = (hi << 8) | (lo >> 24);
      uint32_t v = __umul24(y[j], w) - __umul24(q, p);  // in [0,2p)
      uint32_t xx = x[j];
      uint32_t d = xx - p;
      xx = d < xx ? d : xx;
      x[j] = (xx + v) & 0xffffff;
      y[j] = (xx - v + 2 * p) & 0xffffff;
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= x[j] ^ y[j];
  out[t] = r;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Test {
  const char* name;
  kern_t k;
  double ops_per_iter;  // wave-instructions (or butterflies) per thread per ITER step
};

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, cus, prop.clockRate);
  // waves per SIMD = blocks per CU (256-thread blocks): default 8, argv[1] overrides (e.g. 4 = the row kernel's occupancy)
  const int per_cu = argc > 1 ? std::atoi(argv[1]) : 8;
  const int blocks = cus * per_cu, threads = 256;
  uint32_t* out;
  CK(hipMalloc(&out, sizeof(uint32_t) * blocks * threads));
  std::vector<Test> tests = {
      {"v_add_u32", k_add_u32, 64},         {"v_sub_u32", k_sub_u32, 64},
      {"v_min_u32", k_min_u32, 64},         {"v_add3_u32", k_add3_u32, 64},
      {"v_lshl_add_u32", k_lshl_add, 64},   {"v_alignbit_b32", k_alignbit, 64},
      {"v_mov_dpp(xor1)", k_mov_dpp_xor1, 64},
      {"v_fma_f32", k_fma_f32, 64},         {"v_mul_lo_u32", k_mul_lo_u32, 64},
      {"v_mul_hi_u32", k_mul_hi_u32, 64},   {"v_mad_u64_u32", k_mad_u64_u32, 64},
      {"v_mad_i64_i32", k_mad_i64_i32, 64},
      {"v_mul_u32_u24", k_mul_u32_u24, 64}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 64},
      {"v_mad_u32_u24", k_mad_u32_u24, 64}, {"v_fma_f64", k_fma_f64, 64},
      {"v_mul_f64", k_mul_f64, 64},         {"v_add_f64", k_add_f64, 64},
      {"v_rndne_f64", k_rndne_f64, 64},
      {"v_pk_fma_f32 (2 lanes/op)", k_pk_fma_f32, 64}, {"v_pk_mul_f32", k_pk_mul_f32, 64}, {"v_pk_add_f32", k_pk_add_f32, 64},
      {"bfly_mont32(mad64)", k_bfly_mont, 8},
      {"bfly_shoup32", k_bfly_shoup, 8},    {"bfly_f64_p50", k_bfly_f64, 8},
      {"bfly_u24", k_bfly_u24, 8},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double add_rate = 0;
  for (auto& t : tests) {
    t.k<<<blocks, threads>>>(out, 1);  // warm
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      t.k<<<blocks, threads>>>(out, 1 + rep);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    double waves = (double)blocks * threads / 64.0;
    double winst = waves * t.ops_per_iter * ITER;         // wave-level ops
    double per_simd_per_ns = winst / (best * 1e6) / (cus * 4.0);
    double cyc = 2.4 / per_simd_per_ns;                   // cycles @2.4GHz per wave-op per SIMD
    double lane_ops = winst * 64 / (best * 1e-3);         // lane-ops per second
    if (add_rate == 0) add_rate = lane_ops;
    printf("%-22s %8.3f ms  %7.2f cyc/wave-op/SIMD(@2.4GHz)  %8.2f Tlane-op/s  rel_add=%.3f\n", t.name,
           best, cyc, lane_ops / 1e12, lane_ops / add_rate);
  }
  CK(hipFree(out));
  return 0;
}
