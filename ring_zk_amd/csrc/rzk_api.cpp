// rzk_api.cpp — the C ABI declared in include/rzk.h: context, resident key, row programs of every
// protocol phase, host-pointer wrappers.  Compiled with hipcc together with rzk_kernels.hip into
// ring_zk_amd/librzk_hip.so.  There is no CPU fallback anywhere in this file: every entry point
// launches HIP kernels on the context's device or returns an error status.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rzk.h"
#include "rzk_core.h"
#include "rzk_dev.h"
#include "rzk_tables.h"

using namespace rzk;

// table_load (rzk_dev.h) reads these records with scalar loads: they must be naturally aligned inside their tables
static_assert(sizeof(Term) == 8 && sizeof(AddTerm) == 4 && sizeof(Row) == 16 && sizeof(Item) == 16 && sizeof(Unit) == 8,
              "program record sizes");
static_assert(offsetof(Program, rows) % 8 == 0 && offsetof(Program, terms) % 8 == 0 && offsetof(Program, adds) % 4 == 0 &&
                  offsetof(WaveProgram, units) % 8 == 0 && offsetof(WaveProgram, items) % 8 == 0,
              "program record alignment");

namespace {

enum KeyClass : uint8_t { KC_ZERO = 0, KC_ONE = 1, KC_GENERAL = 2 };

enum ProgId : int {
  PG_MATVEC = 0,      // variant = which*2 + has_addend
  PG_POLYMUL,
  PG_CMUL,            // variant = rows
  PG_OPEN_COMMIT,
  PG_RESPONSE,        // variant = number of (y,r,z) triples sharing d (1 = open, 2 = linear)
  PG_A1_RELATION,     // a1.z - c1(.)d - t == 0   (open / linear / sum verify)
  PG_LIN_COMMIT2,
  PG_LIN_U,
  PG_LIN_V1,
  PG_LIN_V2,
  PG_SUM_XP,          // variant = V
  PG_SUM_U,           // variant = V
  PG_SUM_W2,          // variant = V
  PG_SUM_V3,          // variant = V
  PG_COMMIT,          // c = [a1;a2].r + [0;x]                       (commit.rs:88-128)
  PG_COMMIT_VERIFY,   // variant bit 1: opening has a scalar f       (commit.rs:173-210)
  PG_A1Z,             // w = a1.z (n rows), norm predicate on z fused (bit 0)   } the A1 relation in two steps for
  PG_REL_ROT,         // w - c1(.)d - t == 0, all rotations                      } n >= 2: grouped rows + rotations
  PG_SUM_D,           // variant = V: D_c = sum_i g_i(.)v_{i,c} - v'_c for the columns c that a2 uses   } sum_i g_i (a2.v_i) - a2.v'
  PG_SUM_V4,          // a2.D - w2(.)d - u == 0                                                           }   = a2.(sum_i g_i v_i - v')
  PG_LIN_V1B,         // Linear verifier, rearranged: relation rows, e = a2.z - c2(.)d, e' = a2.z' - c2'(.)d   } (a2.z)(.)g - a2.z' - (c2(.)g - c2')(.)d - u
  PG_LIN_V2B,         // g(.)e - e' - u == 0                                                                     }   = g(.)e - e' - u
};

struct DevProg {
  Program* d = nullptr;
  uint32_t nrows = 0;
  WaveProgram* d_wp = nullptr;   // units / items of unit_kernel (the default path)
  uint32_t nunits = 0;
  uint32_t work = 0;             // steps (item + inverse trips) of one batch entry per prime pass
  bool has_vec = false;
  // shared-operand path (fwd_slots_kernel + row_slots_kernel), chosen when rows share enough operands
  SlotTable* d_slots = nullptr;
  uint32_t nslots = 0;
  uint32_t np_store = 0;
  uint32_t ngroups = 0;   // > 0: row groups (row_group_kernel)
  bool has_dkey = false;  // products with prepared multiplier images (TERM_DKEY): row_kernel only
  bool has_dd = false;    // ... whose other operand may come from the call's operand images (TERM_DD): row_kernel<.., DD>
  bool shift = false;     // every product has the sparse challenge as multiplier: shift_row_kernel
  bool has_shift = false; // some rows end with challenge products evaluated by rotations inside row_kernel
  bool two_bit = false;   // two-bit verdict flags (CHECK2 marks)
  BlockPlan* d_blocks = nullptr;   // row blocks (row_block_kernel): operands of a block staged once in LDS
  uint32_t nblocks = 0;
  // algorithmic traffic of one batch entry (instrumentation): distinct polynomials read per operand index, rows stored
  uint16_t polys_in[kMaxOperands] = {};
  uint32_t polys_out = 0;
};

struct Arena {   // grow-only device buffer
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct rzk_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  int num_cus = 256;
  int64_t q = 0;
  uint32_t N = 0, logn = 0, n = 0, k = 0, l = 0, kappa = 0;
  uint64_t b = 0;
  uint64_t sigma = 0, commit_bound = 0, verify_bound = 0;
  DevTables hT{};
  DevTables* dT = nullptr;
  uint32_t* d_tw = nullptr;
  uint32_t* d_row_scratch = nullptr;   // per-wave Garner state of the row kernel (third prime only)
  uint32_t* d_group_scratch = nullptr; // per-wave Garner state of the row-group kernel (allocated on first use)
  uint32_t* d_block_scratch = nullptr; // per-workgroup Garner state of the row-block kernel (allocated on first use)
  uint32_t block_min_logn = 11;        // row blocks from this ring degree on (below it row groups do the sharing)
  bool use_groups = true;
  uint32_t units_per_task = 0;         // 0 = automatic (RZK_UPT overrides, tuning)
  bool vec_rows = true;                // programs with vector x vector products: row_kernel (RZK_VEC_ROWS=0: unit_kernel)
  bool unit_io = false;                // key-product programs through unit_io_kernel (every operand read once): where its sums park in
                                       // LDS (N = 512); RZK_UNIT_IO=0 / 1 forces unit_kernel / unit_io_kernel
  bool pair_poly = true;               // N = 2048: two wavefronts per polynomial (RZK_PAIR_POLY=0: one, the round-2 kernels)
  bool trusted = false;                // rzk_ctx_trust_device_outputs: skip the canonical test of loaded coefficients
  bool use_pairs = true;               // unit_kernel: pair rows that share their last operand (RZK_PAIRS=0 turns it off, tuning)
  int group_max = 1;                   // rows per group of row_group_kernel (group_max_for; RZK_GROUP_MAX overrides, tuning)
  bool use_shift = true;               // challenge products as signed rotations (shift_row_kernel) instead of transforms
  int use_dkey = 1;                    // the scalar multipliers g_i of Linear / Sum transformed once per proof and call (TERM_DKEY): 0 never (every row
                                       // transforms them itself), 1 when at least kDkeyMinUses rows of the call multiply by each, 2 always (RZK_DKEY)
  Arena ws_dkey;                       // their images + norms
  const uint32_t* dkey_img = nullptr;  // images of the call in progress (set by prepare_dkey, cleared by the entry point)
  const double* dkey_l2 = nullptr;
  uint32_t dkey_n = 0;
  bool use_oimg = true;                // Sum proof: the transforms of y_i / z_i that the a1 key products compute anyway are kept for the D rows (RZK_OIMG=0: off)
  Arena ws_oimg;
  Operands oimg_state{};               // oimg* fields of the call in progress (oimg == NULL: none); oimg_op is set per launch
  uint32_t oimg_producer_op = 0xffu;   // != 0xff: the next launches store the images of this operand's columns
  bool lin_e = true;                   // Linear verifier: g(.)(a2.z - c2(.)d) - (a2.z' - c2'(.)d) - u == 0 (one product with g; RZK_LIN_E=0: the reference's grouping)
  int sum_d = -1;                      // Sum proof: a2.(sum_i g_i v_i - v') instead of sum_i g_i (a2.v_i) - a2.v' (-1 = by cost, RZK_SUM_D=0|1 forces)
  bool preset_in_kernel = true;        // verdict flags initialised by the unit kernels themselves where one team owns an entry (RZK_PRESET_IN_KERNEL=0: always a fill launch)
  bool small = false;                  // N < 512: schoolbook kernels (rzk_kernels.hip, "small ring degrees")
  uint32_t r2q = 0;                    // 2^64 mod q
  uint32_t* d_key_mont = nullptr;      // small N: key entries as Montgomery-form residues mod q
  // key
  bool key_loaded = false;
  std::vector<uint8_t> key_class;     // (n+l)*k
  std::vector<int32_t> key_entry;     // index into the NTT-domain store, -1 if not GENERAL
  uint32_t n_general = 0;
  uint32_t* d_key_ntt = nullptr;
  double* d_key_l2 = nullptr;
  std::map<std::pair<int, uint32_t>, DevProg> progs;
  Arena ws, stage, ws_slots;
  // canonical-input test (rzk_dev.h, Operands::bad): sticky word set by any kernel that loaded a coefficient
  // outside the centred range on behalf of an entry point without per-proof verdicts; read back at every
  // synchronising call (host-pointer variants, rzk_ctx_synchronize, rzk_ctx_check_inputs)
  uint32_t* d_bad = nullptr;
  uint32_t* h_bad = nullptr;   // pinned
  double slot_share_min = 2.0;   // use the shared-operand path when (operand transforms) / (distinct operands) >= this
  std::string err;
  // profiling of the row kernel with HIP events on the launch stream
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  struct ProfInfo {
    std::string kernel;     // the kernel template instance the launch ran
    uint64_t bytes = 0;     // algorithmic bytes of the launch: 8 N x (distinct polynomials read + rows stored)
  };
  std::vector<ProfInfo> prof_info;   // parallel to the used part of prof_events
  size_t prof_used = 0;
  double prof_us = 0.0;
  uint64_t prof_launches = 0;
};

namespace {

#define HIPCHK(ctx, expr)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                            \
      return RZK_E_HIP;                                                                          \
    }                                                                                            \
  } while (0)

std::string g_create_err;   // last rzk_ctx_create failure (no context exists yet to hold it)

int create_fail(int code, const std::string& msg) {
  g_create_err = msg;
  return code;
}

int fail(rzk_ctx* c, int code, const char* msg) {
  if (c) c->err = msg;
  return code;
}

uint64_t isqrt_u64(uint64_t x) {
  uint64_t r = (uint64_t)std::sqrt((long double)x);
  while ((unsigned __int128)r * r > x) --r;
  while ((unsigned __int128)(r + 1) * (r + 1) <= x) ++r;
  return r;
}

LaunchCfg cfg_of(rzk_ctx* c) {
  (void)hipSetDevice(c->device);   // the calling thread may have another current device
  return LaunchCfg{(void*)c->stream, c->num_cus, c->pair_poly ? 1 : 0, c->unit_io ? 1 : 0};
}

int arena_reserve(rzk_ctx* c, Arena& a, size_t bytes) {
  if (bytes <= a.cap) return RZK_OK;
  // grow-only; happens outside steady state.  The old block may still be in use by queued work.
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (a.p) HIPCHK(c, hipFree(a.p));
  a.p = nullptr;
  a.cap = 0;
  size_t want = bytes + bytes / 4;
  HIPCHK(c, hipMalloc(&a.p, want));
  a.cap = want;
  return RZK_OK;
}

// ---- program builder ------------------------------------------------------------------------------------------
struct PB {
  Program p{};
  bool overflow = false;
  bool two_bit = false;      // the program carries CHECK2 marks: two-bit verdict flags (row_kernel only)
  uint32_t sparse_ops = 0;   // bit i: operand i is a challenge (kappa-sparse, +-1): products with it may use shift-add
  int cur = -1;
  void begin_row(uint8_t out_op, uint32_t out_off, uint8_t mode) {
    if (p.nrows >= (uint32_t)kMaxRows) { overflow = true; return; }
    cur = (int)p.nrows++;
    Row& r = p.rows[cur];
    r.term0 = (uint16_t)p.nterms;
    r.add0 = (uint16_t)p.nadds;
    r.nterms = r.nadds = 0;
    r.nshift = r.pad = 0;
    r.out_op = out_op;
    r.out_off = (uint16_t)out_off;
    r.mode = mode;
    if (out_off > 0xffff) overflow = true;
  }
  // total terms of the program so far (transform terms and shift terms share Program::terms)
  void key_term(int sign, uint32_t entry, uint8_t vop, uint32_t voff) {
    if (cur < 0 || p.nterms >= (uint32_t)kMaxTerms || entry > 0xffff || voff > 0xffff || p.rows[cur].nshift) { overflow = true; return; }
    Term& t = p.terms[p.nterms++];
    t.kind = TERM_KEY;
    t.sign = (int8_t)sign;
    t.a_op = 0;
    t.a_off = (uint16_t)entry;
    t.b_op = vop;
    t.b_off = (uint16_t)voff;
    p.rows[cur].nterms++;
  }
  // sign * (image `idx` of the entry's own multipliers) (.) (bop,boff): Operands::dkey_img, row_kernel only
  void dkey_term(int sign, uint32_t idx, uint8_t bop, uint32_t boff) {
    if (cur < 0 || p.nterms >= (uint32_t)kMaxTerms || idx > 0xffff || boff > 0xffff || p.rows[cur].nshift) { overflow = true; return; }
    Term& t = p.terms[p.nterms++];
    t.kind = TERM_DKEY;
    t.sign = (int8_t)sign;
    t.a_op = 0;
    t.a_off = (uint16_t)idx;
    t.b_op = bop;
    t.b_off = (uint16_t)boff;
    p.rows[cur].nterms++;
  }
  // a product with one of the entry's scalar multipliers (operand gop, index idx): its image when the call prepared
  // them (dk), a vector x vector term otherwise; oi: the other operand's transform may come from the call's operand images
  void scalar_term(bool dk, int sign, uint8_t gop, uint32_t idx, uint8_t bop, uint32_t boff, bool oi = false) {
    if (dk) {
      dkey_term(sign, idx, bop, boff);
      if (oi && !overflow) p.terms[p.nterms - 1].kind = TERM_DD;
    } else {
      vec_term(sign, bop, boff, gop, idx);
    }
  }
  void vec_term(int sign, uint8_t aop, uint32_t aoff, uint8_t bop, uint32_t boff) {
    if (cur < 0 || p.nterms >= (uint32_t)kMaxTerms || aoff > 0xffff || boff > 0xffff || p.rows[cur].nshift) { overflow = true; return; }
    Term& t = p.terms[p.nterms++];
    t.kind = TERM_VEC;
    t.sign = (int8_t)sign;
    t.a_op = aop;
    t.a_off = (uint16_t)aoff;
    t.b_op = bop;
    t.b_off = (uint16_t)boff;
    p.rows[cur].nterms++;
  }
  // sign * (aop,aoff) (.) (bop,boff) with a sparse `a` (the challenge), evaluated as signed rotations inside the
  // row kernel; such terms close a row's term list (stored behind its transform terms)
  void shift_term(int sign, uint8_t aop, uint32_t aoff, uint8_t bop, uint32_t boff) {
    if (cur < 0 || p.nterms >= (uint32_t)kMaxTerms || aoff > 0xffff || boff > 0xffff) { overflow = true; return; }
    Term& t = p.terms[p.nterms++];
    t.kind = TERM_SHIFT;
    t.sign = (int8_t)sign;
    t.a_op = aop;
    t.a_off = (uint16_t)aoff;
    t.b_op = bop;
    t.b_off = (uint16_t)boff;
    p.rows[cur].nshift++;
  }
  // product with the challenge: rotations when enabled, transform product otherwise
  void challenge_term(bool rotate, int sign, uint8_t dop, uint8_t bop, uint32_t boff) {
    if (rotate) shift_term(sign, dop, 0, bop, boff);
    else vec_term(sign, dop, 0, bop, boff);
  }
  void add(int sign, uint8_t op, uint32_t off) {
    if (cur < 0 || p.nadds >= (uint32_t)kMaxAdds || off > 0xffff) { overflow = true; return; }
    AddTerm& a = p.adds[p.nadds++];
    a.op = op;
    a.sign = (int8_t)sign;
    a.off = (uint16_t)off;
    p.rows[cur].nadds++;
  }
};

// Fused norm predicate: mark, for each polynomial (vop, 0..count-1), the first load in program order
// (b operand of a product term, or one of the first four additions of a row).  Returns false when some
// polynomial is never loaded by the program — the caller then keeps the separate norm kernel.
bool mark_checks(PB& pb, uint8_t vop, uint32_t count, bool second = false) {
  const uint8_t tmark = second ? TERM_CHECK2 : TERM_CHECK, amark = second ? ADD_CHECK2 : ADD_CHECK;
  if (second) pb.two_bit = true;
  for (uint32_t j = 0; j < count; ++j) {
    bool done = false;
    for (uint32_t r = 0; r < pb.p.nrows && !done; ++r) {
      const Row& row = pb.p.rows[r];
      for (uint32_t t = 0; t < row.nterms && !done; ++t) {
        Term& tm = pb.p.terms[row.term0 + t];
        if (tm.b_op == vop && tm.b_off == j && !(tm.kind & (TERM_CHECK | TERM_CHECK2))) {
          tm.kind |= tmark;
          done = true;
        }
      }
      for (uint32_t a = 0; a < row.nadds && a < 4 && !done; ++a) {
        AddTerm& ad = pb.p.adds[row.add0 + a];
        if ((ad.op & ADD_OP_MASK) == vop && ad.off == j && !(ad.op & (ADD_CHECK | ADD_CHECK2))) {
          ad.op |= amark;
          done = true;
        }
      }
    }
    if (!done) return false;
  }
  return true;
}

// sign * (row `krow` of [a1;a2]) . v, v = operand (vop, voff .. voff+k-1): skips zero entries, turns
// entries equal to 1 into plain additions (the identity blocks of commit.rs:38-57), products otherwise.
void key_row(rzk_ctx* c, PB& pb, int sign, uint32_t krow, uint8_t vop, uint32_t voff) {
  for (uint32_t j = 0; j < c->k; ++j) {
    const uint32_t idx = krow * c->k + j;
    switch (c->key_class[idx]) {
      case KC_ZERO: break;
      case KC_ONE: pb.add(sign, vop, voff + j); break;
      default: pb.key_term(sign, (uint32_t)c->key_entry[idx], vop, voff + j); break;
    }
  }
}

// Challenge products as signed rotations (shift-add) instead of transforms.  At N = 2048 only with two wavefronts
// per polynomial (16 outputs per thread): with one, a lane holds 32 outputs, the rotation kernels need > 200 VGPRs
// (one or two waves per SIMD) and measured slower than the transform path (round 2).
bool shift_ok(const rzk_ctx* c) { return c->use_shift && !c->small && (c->logn <= 10 || c->pair_poly); }

// bit of a program variant: products with the entry's scalar multipliers (g, g_i) take their prepared images (TERM_DKEY)
constexpr uint32_t kDkeyVar = 0x10000u;
constexpr uint32_t kOimgVar = 0x20000u;   // ... and (PG_SUM_D) the other operand's transform from the call's operand images (TERM_DD)

int build_program(rzk_ctx* c, int id, uint32_t var_in, PB& pb) {
  const uint32_t n = c->n, k = c->k, l = c->l;
  const bool rot = shift_ok(c);   // challenge products inside mixed rows as rotations
  const bool dk = (var_in & kDkeyVar) != 0, oi = (var_in & kOimgVar) != 0;
  const uint32_t var = var_in & ~(kDkeyVar | kOimgVar);
  switch (id) {
    case PG_MATVEC: {   // ops: 0 = v[k], 1 = addend[rows], 2 = out[rows]
      const uint32_t which = var >> 1;
      const bool has_add = var & 1;
      const uint32_t r0 = which == RZK_KEY_A2 ? n : 0;
      const uint32_t rows = which == RZK_KEY_A1 ? n : (which == RZK_KEY_A2 ? l : n + l);
      for (uint32_t i = 0; i < rows; ++i) {
        pb.begin_row(2, i, MODE_STORE);
        key_row(c, pb, +1, r0 + i, 0, 0);
        if (has_add) pb.add(+1, 1, i);
      }
      break;
    }
    case PG_POLYMUL:   // ops: 0 = a, 1 = b, 2 = out
      pb.begin_row(2, 0, MODE_STORE);
      pb.vec_term(+1, 0, 0, 1, 0);
      break;
    case PG_CMUL:      // ops: 0 = m[rows], 1 = p, 2 = out[rows]   (mat.rs:168-178)
      for (uint32_t i = 0; i < var; ++i) {
        pb.begin_row(2, i, MODE_STORE);
        pb.scalar_term(dk, +1, 1, 0, 0, i);
      }
      break;
    case PG_OPEN_COMMIT:   // ops: 0 = x[l], 1 = r[k], 2 = y[k], 3 = c[n+l], 4 = t[n]
      for (uint32_t i = 0; i < n + l; ++i) {   // commit.rs:125: c = [a1;a2].r + [0_n ; x]
        pb.begin_row(3, i, MODE_STORE);
        key_row(c, pb, +1, i, 1, 0);
        if (i >= n) pb.add(+1, 0, i - n);
      }
      for (uint32_t i = 0; i < n; ++i) {       // open.rs:97: t = a1.y
        pb.begin_row(4, i, MODE_STORE);
        key_row(c, pb, +1, i, 2, 0);
      }
      if (var & 1) {   // fused check_commit_constraint(r)  (commit.rs:98-107)
        if (!mark_checks(pb, 1, k)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_COMMIT:   // ops: 0 = x[l], 1 = r[k], 2 = c[n+l]
      for (uint32_t i = 0; i < n + l; ++i) {   // commit.rs:109-125
        pb.begin_row(2, i, MODE_STORE);
        key_row(c, pb, +1, i, 1, 0);
        if (i >= n) pb.add(+1, 0, i - n);
      }
      if (var & 1) {   // fused check_commit_constraint(r)  (commit.rs:98-107)
        if (!mark_checks(pb, 1, k)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_COMMIT_VERIFY:   // ops: 0 = x[l], 1 = r[k], 2 = c[n+l], 3 = f ; flags &= (commit.rs:199-209)
      for (uint32_t i = 0; i < n + l; ++i) {
        pb.begin_row(0, 0, MODE_ZERO);
        key_row(c, pb, +1, i, 1, 0);
        if (var & 2) {   // a.r + z(.)f - c(.)f == 0
          if (i >= n) pb.vec_term(+1, 0, i - n, 3, 0);
          pb.vec_term(-1, 2, i, 3, 0);
        } else {         // a.r + z - c == 0
          if (i >= n) pb.add(+1, 0, i - n);
          pb.add(-1, 2, i);
        }
      }
      if (var & 1) {   // fused check_commit_constraint(r)  (commit.rs:183-185)
        if (!mark_checks(pb, 1, k)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_A1Z:   // ops: 0 = z[k], 1 = w[n]
      for (uint32_t i = 0; i < n; ++i) {
        pb.begin_row(1, i, MODE_STORE);
        key_row(c, pb, +1, i, 0, 0);
      }
      if (var & 1) {   // fused check_verify_constraint(z)
        if (!mark_checks(pb, 0, k)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_REL_ROT:   // ops: 0 = w[n] (= a1.z), 1 = t[n], 2 = c[n+l], 3 = d ; flags &= (w == t + c1(.)d)
      pb.sparse_ops = 1u << 3;
      for (uint32_t i = 0; i < n; ++i) {
        pb.begin_row(0, 0, MODE_ZERO);
        pb.vec_term(-1, 3, 0, 2, i);
        pb.add(+1, 0, i);
        pb.add(-1, 1, i);
      }
      break;
    case PG_RESPONSE:   // ops: 0 = d, then per triple s: 1+3s = y[k], 2+3s = r[k], 3+3s = z[k]
      pb.sparse_ops = 1u << 0;
      for (uint32_t s = 0; s < var; ++s)
        for (uint32_t i = 0; i < k; ++i) {      // open.rs:113-115: z = y + r (.) d
          pb.begin_row((uint8_t)(3 + 3 * s), i, MODE_STORE);
          pb.vec_term(+1, 0, 0, (uint8_t)(2 + 3 * s), i);
          pb.add(+1, (uint8_t)(1 + 3 * s), i);
        }
      break;
    case PG_A1_RELATION:   // ops: 0 = z[k], 1 = t[n], 2 = c[n+l], 3 = d ; flags &= (a1.z == t + c1(.)d)
      // c1 = first l rows of c (Commitment::c1_c2 -> split_rows(n), commit.rs:213-218, mat.rs:203-213);
      // Mat::add requires it to have n rows, so n == l is checked by the caller.
      for (uint32_t i = 0; i < n; ++i) {
        pb.begin_row(0, 0, MODE_ZERO);
        key_row(c, pb, +1, i, 0, 0);
        pb.challenge_term(rot, -1, 3, 2, i);
        pb.add(-1, 1, i);
      }
      if (var & 1) {   // fused check_verify_constraint(z)  (open.rs:167-169)
        if (!mark_checks(pb, 0, k)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_LIN_COMMIT2:
      // ops: 0 = x[l], 1 = gx[l], 2 = r[k], 3 = rp[k], 4 = y[k], 5 = yp[k],
      //      6 = c[n+l], 7 = cp[n+l], 8 = t[n], 9 = tp[n], 10 = a2y[l]
      for (uint32_t i = 0; i < n + l; ++i) {   // linear.rs:97: c = commit(x; r)
        pb.begin_row(6, i, MODE_STORE);
        key_row(c, pb, +1, i, 2, 0);
        if (i >= n) pb.add(+1, 0, i - n);
      }
      for (uint32_t i = 0; i < n + l; ++i) {   // linear.rs:96: cp = commit(g*x; rp)
        pb.begin_row(7, i, MODE_STORE);
        key_row(c, pb, +1, i, 3, 0);
        if (i >= n) pb.add(+1, 1, i - n);
      }
      for (uint32_t i = 0; i < n; ++i) {       // linear.rs:118
        pb.begin_row(8, i, MODE_STORE);
        key_row(c, pb, +1, i, 4, 0);
      }
      for (uint32_t i = 0; i < l; ++i) {       // a2.y, reduced mod q before it meets g (linear.rs:124-127); placed
        pb.begin_row(10, i, MODE_STORE);       // next to t = a1.y so that the two rows can share the transform of y
        key_row(c, pb, +1, n + i, 4, 0);
      }
      for (uint32_t i = 0; i < n; ++i) {       // linear.rs:121
        pb.begin_row(9, i, MODE_STORE);
        key_row(c, pb, +1, i, 5, 0);
      }
      if (var & 1) {   // fused check_commit_constraint: r -> bit 0, rp -> bit 1 of ok (the two commits of linear.rs:96-97)
        if (!mark_checks(pb, 2, k) || !mark_checks(pb, 3, k, true)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_LIN_U:   // ops: 0 = a2y[l], 1 = g, 2 = yp[k], 3 = u[l] : u = a2y(.)g - a2.yp (linear.rs:124-129)
      for (uint32_t i = 0; i < l; ++i) {
        pb.begin_row(3, i, MODE_STORE);
        pb.scalar_term(dk, +1, 1, 0, 0, i);
        key_row(c, pb, -1, n + i, 2, 0);
      }
      break;
    case PG_LIN_V1:
      // ops: 0 = z[k], 1 = zp[k], 2 = t[n], 3 = tp[n], 4 = c[n+l], 5 = cp[n+l], 6 = d, 7 = g,
      //      8 = w1[l] (a2.z), 9 = w2[l] (c2(.)g - c2p)
      for (uint32_t i = 0; i < n; ++i) {       // linear.rs:225-229
        pb.begin_row(0, 0, MODE_ZERO);
        key_row(c, pb, +1, i, 0, 0);
        pb.challenge_term(rot, -1, 6, 4, i);
        pb.add(-1, 2, i);
      }
      for (uint32_t i = 0; i < n; ++i) {       // linear.rs:231-235
        pb.begin_row(0, 0, MODE_ZERO);
        key_row(c, pb, +1, i, 1, 0);
        pb.challenge_term(rot, -1, 6, 5, i);
        pb.add(-1, 3, i);
      }
      for (uint32_t i = 0; i < l; ++i) {       // a2.z (linear.rs:238-241), reduced before (.)g
        pb.begin_row(8, i, MODE_STORE);
        key_row(c, pb, +1, n + i, 0, 0);
      }
      for (uint32_t i = 0; i < l; ++i) {       // c2(.)g - c2p (linear.rs:243-246); c2 = last n rows of c
        pb.begin_row(9, i, MODE_STORE);
        pb.scalar_term(dk, +1, 7, 0, 4, l + i);
        pb.add(-1, 5, l + i);
      }
      if (var & 1) {   // fused check_verify_constraint(z), (zp)  (linear.rs:218-223)
        if (!mark_checks(pb, 0, k) || !mark_checks(pb, 1, k)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_LIN_V1B:
      // ops: 0 = z[k], 1 = zp[k], 2 = t[n], 3 = tp[n], 4 = c[n+l], 5 = cp[n+l], 6 = d, 7 = e[l], 8 = ep[l]
      // linear.rs:237-249 reads (a2.z)(.)g - a2.z' == (c2(.)g - c2')(.)d + u; in a commutative ring that is
      // g(.)(a2.z - c2(.)d) - (a2.z' - c2'(.)d) - u == 0: one product with g instead of two, and none in this program
      for (uint32_t i = 0; i < n; ++i) {       // linear.rs:225-229
        pb.begin_row(0, 0, MODE_ZERO);
        key_row(c, pb, +1, i, 0, 0);
        pb.challenge_term(rot, -1, 6, 4, i);
        pb.add(-1, 2, i);
      }
      for (uint32_t i = 0; i < n; ++i) {       // linear.rs:231-235
        pb.begin_row(0, 0, MODE_ZERO);
        key_row(c, pb, +1, i, 1, 0);
        pb.challenge_term(rot, -1, 6, 5, i);
        pb.add(-1, 3, i);
      }
      for (uint32_t i = 0; i < l; ++i) {       // e = a2.z - c2(.)d ; c2 = last n rows of c
        pb.begin_row(7, i, MODE_STORE);
        key_row(c, pb, +1, n + i, 0, 0);
        pb.challenge_term(rot, -1, 6, 4, l + i);
      }
      for (uint32_t i = 0; i < l; ++i) {       // e' = a2.z' - c2'(.)d
        pb.begin_row(8, i, MODE_STORE);
        key_row(c, pb, +1, n + i, 1, 0);
        pb.challenge_term(rot, -1, 6, 5, l + i);
      }
      if (var & 1) {   // fused check_verify_constraint(z), (zp)  (linear.rs:218-223)
        if (!mark_checks(pb, 0, k) || !mark_checks(pb, 1, k)) return RZK_E_UNSUPPORTED;
      }
      break;
    case PG_LIN_V2B:   // ops: 0 = e[l], 1 = ep[l], 2 = g, 3 = u[l] : g(.)e - e' - u == 0
      for (uint32_t i = 0; i < l; ++i) {
        pb.begin_row(0, 0, MODE_ZERO);
        pb.scalar_term(dk, +1, 2, 0, 0, i);
        pb.add(-1, 1, i);
        pb.add(-1, 3, i);
      }
      break;
    case PG_LIN_V2:
      // ops: 0 = w1[l], 1 = w2[l], 2 = g, 3 = d, 4 = zp[k], 5 = u[l]
      // w1(.)g - a2.zp - w2(.)d - u == 0   (linear.rs:237-249)
      for (uint32_t i = 0; i < l; ++i) {
        pb.begin_row(0, 0, MODE_ZERO);
        pb.scalar_term(dk, +1, 2, 0, 0, i);
        key_row(c, pb, -1, n + i, 4, 0);
        pb.challenge_term(rot, -1, 3, 1, i);
        pb.add(-1, 5, i);
      }
      break;
    case PG_SUM_XP:   // ops: 0 = xs[V*l], 1 = gs[V], 2 = xp[l] : xp = sum_i x_i (.) g_i (sum.rs:107-115)
      for (uint32_t j = 0; j < l; ++j) {
        pb.begin_row(2, j, MODE_STORE);
        for (uint32_t i = 0; i < var; ++i) pb.scalar_term(dk, +1, 1, i, 0, i * l + j);
      }
      break;
    case PG_SUM_U:    // ops: 0 = w[V*l] (a2.y_i), 1 = gs[V], 2 = yp[k], 3 = u[l]   (sum.rs:154-160)
      for (uint32_t j = 0; j < l; ++j) {
        pb.begin_row(3, j, MODE_STORE);
        for (uint32_t i = 0; i < var; ++i) pb.scalar_term(dk, +1, 1, i, 0, i * l + j);
        key_row(c, pb, -1, n + j, 2, 0);
      }
      break;
    case PG_SUM_W2:   // ops: 0 = cs[V*(n+l)], 1 = gs[V], 2 = cp[n+l], 3 = w2[l] : sum_i c2_i(.)g_i - c2p (sum.rs:309-316)
      for (uint32_t j = 0; j < l; ++j) {
        pb.begin_row(3, j, MODE_STORE);
        for (uint32_t i = 0; i < var; ++i) pb.scalar_term(dk, +1, 1, i, 0, i * (n + l) + l + j);
        pb.add(-1, 2, l + j);
      }
      break;
    case PG_SUM_V3:   // ops: 0 = w1[V*l] (a2.z_i), 1 = gs[V], 2 = zp[k], 3 = w2[l], 4 = d, 5 = u[l]   (sum.rs:301-319)
      for (uint32_t j = 0; j < l; ++j) {
        pb.begin_row(0, 0, MODE_ZERO);
        for (uint32_t i = 0; i < var; ++i) pb.scalar_term(dk, +1, 1, i, 0, i * l + j);
        key_row(c, pb, -1, n + j, 2, 0);
        pb.challenge_term(rot, -1, 4, 3, j);
        pb.add(-1, 5, j);
      }
      break;
    case PG_SUM_D: {  // ops: 0 = vs[V*k] (ys or zs), 1 = gs[V], 2 = vp[k] (yp or zp), 3 = D[k]
      // a2 is linear and the ring commutative: sum_i g_i (.) (a2.v_i) - a2.v' = a2.(sum_i g_i (.) v_i - v')
      // (sum.rs:154-160 and 301-308); only the columns a2 has entries in are formed
      for (uint32_t col = 0; col < k; ++col) {
        bool used = false;
        for (uint32_t j = 0; j < l; ++j) used = used || c->key_class[(n + j) * k + col] != KC_ZERO;
        if (!used) continue;
        pb.begin_row(3, col, MODE_STORE);
        for (uint32_t i = 0; i < var; ++i) pb.scalar_term(dk, +1, 1, i, 0, i * k + col, oi);
        pb.add(-1, 2, col);
      }
      break;
    }
    case PG_SUM_V4:   // ops: 0 = D[k], 1 = w2[l], 2 = d, 3 = u[l] : a2.D - w2(.)d - u == 0   (sum.rs:301-319)
      for (uint32_t j = 0; j < l; ++j) {
        pb.begin_row(0, 0, MODE_ZERO);
        key_row(c, pb, +1, n + j, 0, 0);
        pb.challenge_term(rot, -1, 2, 1, j);
        pb.add(-1, 3, j);
      }
      break;
    default: return RZK_E_ARG;
  }
  return RZK_OK;
}

int get_program(rzk_ctx* c, int id, uint32_t var, DevProg& out) {
  const bool needs_key = !(id == PG_POLYMUL || id == PG_CMUL || id == PG_RESPONSE || id == PG_SUM_XP ||
                           id == PG_SUM_W2 || id == PG_REL_ROT);   // (PG_SUM_D reads the key's classification)
  if (needs_key && !c->key_loaded) return fail(c, RZK_E_STATE, "commitment key not loaded");
  auto it = c->progs.find({id, var});
  if (it != c->progs.end()) {
    out = it->second;
    return RZK_OK;
  }
  PB pb;
  int rc = build_program(c, id, var, pb);
  if (rc == RZK_E_UNSUPPORTED) return rc;   // a fused-check variant that cannot cover every polynomial
  if (rc != RZK_OK) return fail(c, rc, "unknown program");
  if (pb.overflow) return fail(c, RZK_E_UNSUPPORTED, "shape exceeds row-program capacity");
  DevProg dp;
  // Row groups: consecutive rows that are key products over the same operand list are evaluated by one
  // wavefront (row_group_kernel).  Used when it at least halves the number of tasks.
  pb.p.ngroups = 0;
  if (shift_ok(c) && pb.sparse_ops && pb.p.nterms > 0) {
    bool all = true;
    for (uint32_t t = 0; t < pb.p.nterms; ++t) {
      const Term& tm = pb.p.terms[t];
      all = all && tm.kind == TERM_VEC && ((pb.sparse_ops >> tm.a_op) & 1u);   // no fused checks either
    }
    dp.shift = all;
  }
  for (uint32_t r = 0; r < pb.p.nrows; ++r) dp.has_shift = dp.has_shift || pb.p.rows[r].nshift > 0;
  for (uint32_t t = 0; t < pb.p.nterms; ++t) dp.has_dd = dp.has_dd || (pb.p.terms[t].kind & TERM_KIND_MASK) == TERM_DD;
  for (uint32_t t = 0; t < pb.p.nterms; ++t)
    dp.has_dkey = dp.has_dkey || (pb.p.terms[t].kind & TERM_KIND_MASK) == TERM_DKEY || (pb.p.terms[t].kind & TERM_KIND_MASK) == TERM_DD;
  dp.two_bit = pb.two_bit;
  // Row blocks: key-only programs whose rows share operands; consecutive rows are packed into blocks of at
  // most kBlockMaxRows rows and kBlockMaxSlots distinct operands.  Used when every operand is needed by at
  // least two terms on average (otherwise nothing is shared and the plain row kernel is as good).
  if (!c->small && c->logn >= 10 && c->logn >= c->block_min_logn && !dp.shift && !dp.has_shift && !dp.two_bit &&
      !dp.has_dkey && pb.p.nterms > 0) {
    bool key_only = true;
    for (uint32_t t = 0; t < pb.p.nterms; ++t) key_only = key_only && (pb.p.terms[t].kind & TERM_KIND_MASK) == TERM_KEY;
    std::vector<BlockPlan> planv(1);
    BlockPlan& bp = planv[0];
    std::memset(&bp, 0, sizeof(bp));
    bool fits = key_only;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> cur;   // (op, off) -> slot of the open block
    auto open_block = [&](uint32_t row) {
      bp.blk[bp.nblocks].row0 = (uint16_t)row;
      bp.blk[bp.nblocks].nrows = 0;
      bp.blk[bp.nblocks].slot0 = (uint16_t)bp.nslots_total;
      bp.blk[bp.nblocks].nslots = 0;
      cur.clear();
    };
    if (fits) open_block(0);
    for (uint32_t r = 0; fits && r < pb.p.nrows; ++r) {
      const Row& row = pb.p.rows[r];
      std::map<std::pair<uint32_t, uint32_t>, uint32_t> add;   // operands this row brings that the block lacks
      for (uint32_t t = 0; t < row.nterms; ++t) {
        const Term& tm = pb.p.terms[row.term0 + t];
        if (!cur.count({tm.b_op, tm.b_off})) add[{tm.b_op, tm.b_off}] = 0;
      }
      BlockDesc* bd = &bp.blk[bp.nblocks];
      if (bd->nrows == kBlockMaxRows || bd->nslots + add.size() > (size_t)kBlockMaxSlots) {
        if (bd->nrows == 0) { fits = false; break; }   // a single row needs more operands than LDS holds
        ++bp.nblocks;
        if (bp.nblocks >= (uint32_t)kMaxRows) { fits = false; break; }
        open_block(r);
        bd = &bp.blk[bp.nblocks];
        add.clear();
        for (uint32_t t = 0; t < row.nterms; ++t) add[{pb.p.terms[row.term0 + t].b_op, pb.p.terms[row.term0 + t].b_off}] = 0;
        if (add.size() > (size_t)kBlockMaxSlots) { fits = false; break; }
      }
      for (auto& kv : add) {
        if (bp.nslots_total >= (uint32_t)kMaxSlots) { fits = false; break; }
        const uint32_t sidx = bd->nslots++;
        cur[kv.first] = sidx;
        bp.slot_op[bp.nslots_total] = (uint16_t)kv.first.first;
        bp.slot_off[bp.nslots_total] = (uint16_t)kv.first.second;
        ++bp.nslots_total;
      }
      for (uint32_t t = 0; fits && t < row.nterms; ++t) {
        const Term& tm = pb.p.terms[row.term0 + t];
        const uint32_t sidx = cur[{tm.b_op, tm.b_off}];
        bp.term_slot[row.term0 + t] = (uint16_t)sidx;
        if (tm.kind & TERM_CHECK) bp.slot_check[bd->slot0 + sidx] = 1;
      }
      bd->nrows++;
    }
    if (fits) {
      ++bp.nblocks;
      if (bp.nslots_total > 0 && (double)pb.p.nterms / bp.nslots_total >= 2.0) {
        HIPCHK(c, hipMalloc((void**)&dp.d_blocks, sizeof(BlockPlan)));
        HIPCHK(c, hipMemcpyAsync(dp.d_blocks, &bp, sizeof(BlockPlan), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dp.nblocks = bp.nblocks;
      }
    }
  }
  if (!c->small && c->use_groups && !dp.shift && !dp.two_bit && !dp.nblocks && !dp.has_dkey) {
    bool key_only = pb.p.nterms > 0;
    for (uint32_t t = 0; t < pb.p.nterms; ++t) key_only = key_only && (pb.p.terms[t].kind & TERM_KIND_MASK) == TERM_KEY;
    if (key_only) {
      const uint32_t gmax = (uint32_t)c->group_max;
      uint32_t ng = 0;
      for (uint32_t r = 0; r < pb.p.nrows;) {
        uint32_t cnt = 1;
        const Row& r0 = pb.p.rows[r];
        while (cnt < gmax && r + cnt < pb.p.nrows) {
          const Row& rr = pb.p.rows[r + cnt];
          bool same = rr.nterms == r0.nterms && r0.nterms > 0;
          for (uint32_t t = 0; same && t < r0.nterms; ++t) {
            const Term& a = pb.p.terms[r0.term0 + t];
            const Term& b2 = pb.p.terms[rr.term0 + t];
            same = a.b_op == b2.b_op && a.b_off == b2.b_off;
          }
          if (!same) break;
          ++cnt;
        }
        pb.p.groups[ng].row0 = (uint16_t)r;
        pb.p.groups[ng].count = (uint16_t)cnt;
        ++ng;
        r += cnt;
      }
      if (ng * 2 <= pb.p.nrows) {
        pb.p.ngroups = ng;
        dp.ngroups = ng;
      }
    }
  }
  HIPCHK(c, hipMalloc((void**)&dp.d, sizeof(Program)));
  HIPCHK(c, hipMemcpyAsync(dp.d, &pb.p, sizeof(Program), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // pb.p is a stack object; one-off per (program, shape)
  dp.nrows = pb.p.nrows;
  if (!c->small) {
    // Wave program of unit_kernel: one unit per row; two consecutive rows become a PAIR when the second has exactly
    // one key product and its operand is the last operand of the first (c0 / c1 of a commitment share r_{k-1};
    // t = a1.y and a2.y share y_{k-1}): that transform is then computed once for both.
    std::vector<WaveProgram> wpv(1);
    WaveProgram& wp = wpv[0];
    std::memset(&wp, 0, sizeof(wp));
    auto key_only = [&](const Row& rr) {
      for (uint32_t t = 0; t < rr.nterms; ++t)
        if ((pb.p.terms[rr.term0 + t].kind & TERM_KIND_MASK) != TERM_KEY) return false;
      return true;
    };
    for (uint32_t r = 0; r < pb.p.nrows;) {
      const Row& ra = pb.p.rows[r];
      Unit& un = wp.units[wp.nunits++];
      un.rowA = (uint16_t)r;
      un.rowB = kNoRow;
      un.item0 = (uint16_t)wp.nitems;
      un.nitems = ra.nterms;
      for (uint32_t t = 0; t < ra.nterms; ++t) {
        const Term& tm = pb.p.terms[ra.term0 + t];
        Item& im = wp.items[wp.nitems++];
        im.kind = (tm.kind & TERM_KIND_MASK) == TERM_VEC ? ITEM_VEC : ITEM_KEY;
        im.flags = tm.kind & (TERM_CHECK | TERM_CHECK2);
        im.b_op = tm.b_op;
        im.b_off = tm.b_off;
        im.a_op = tm.a_op;
        im.a_off = tm.a_off;
        im.keyA = im.kind == ITEM_KEY ? tm.a_off : 0;
        im.keyB = kNoKey;
        im.signA = tm.sign;
        im.signB = 0;
      }
      uint32_t step = 1;
      if (c->use_pairs && r + 1 < pb.p.nrows && ra.nterms >= 1 && ra.nshift == 0 && key_only(ra)) {
        const Row& rb = pb.p.rows[r + 1];
        if (rb.nterms == 1 && rb.nshift == 0 && key_only(rb)) {
          const Term& tb = pb.p.terms[rb.term0];
          const Term& ta = pb.p.terms[ra.term0 + ra.nterms - 1];
          if (tb.b_op == ta.b_op && tb.b_off == ta.b_off) {
            Item& im = wp.items[wp.nitems - 1];
            im.keyB = tb.a_off;
            im.signB = tb.sign;
            im.flags |= tb.kind & (TERM_CHECK | TERM_CHECK2);
            un.rowB = (uint16_t)(r + 1);
            step = 2;
          }
        }
      }
      r += step;
    }
    HIPCHK(c, hipMalloc((void**)&dp.d_wp, sizeof(WaveProgram)));
    HIPCHK(c, hipMemcpyAsync(dp.d_wp, &wp, sizeof(WaveProgram), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dp.nunits = wp.nunits;
    for (uint32_t u = 0; u < wp.nunits; ++u)
      dp.work += (wp.units[u].nitems ? wp.units[u].nitems : 1u) + (wp.units[u].rowB != kNoRow ? 2u : 1u);
  }
  for (uint32_t t = 0; t < pb.p.nterms; ++t) dp.has_vec = dp.has_vec || (pb.p.terms[t].kind & TERM_KIND_MASK) == TERM_VEC;
  // distinct operands of the product terms ("slots"); when rows share them often enough, transform each
  // once per proof (shared-operand path) instead of once per row
  if (!c->small && c->slot_share_min > 0 && pb.p.nterms > 0 && dp.ngroups == 0 && !dp.shift && !dp.has_shift &&
      !dp.two_bit && !dp.nblocks && !dp.has_dkey) {
    std::vector<SlotTable> stv(1);
    SlotTable& st = stv[0];
    std::memset(&st, 0, sizeof(st));
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> index;
    bool fits = true;
    uint32_t transforms = 0;
    auto slot_of = [&](uint8_t op, uint16_t off) -> uint32_t {
      auto it = index.find({op, off});
      if (it != index.end()) return it->second;
      if (st.nslots >= (uint32_t)kMaxSlots) { fits = false; return 0; }
      const uint32_t sidx = st.nslots++;
      st.op[sidx] = op;
      st.off[sidx] = off;
      index[{op, off}] = sidx;
      return sidx;
    };
    for (uint32_t t = 0; t < pb.p.nterms; ++t) {
      const Term& tm = pb.p.terms[t];
      const uint32_t sb = slot_of(tm.b_op, tm.b_off);
      st.term_b[t] = (uint16_t)sb;
      ++transforms;
      if (tm.kind & TERM_CHECK) st.check[sb] = 1;
      if ((tm.kind & TERM_KIND_MASK) == TERM_VEC) {
        st.term_a[t] = (uint16_t)slot_of(tm.a_op, tm.a_off);
        ++transforms;
      }
    }
    if (fits && st.nslots > 0 && (double)transforms / st.nslots >= c->slot_share_min) {
      HIPCHK(c, hipMalloc((void**)&dp.d_slots, sizeof(SlotTable)));
      HIPCHK(c, hipMemcpyAsync(dp.d_slots, &st, sizeof(SlotTable), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      dp.nslots = st.nslots;
      dp.np_store = dp.has_vec ? 3 : 2;
    }
  }
  {
    std::map<std::pair<uint32_t, uint32_t>, int> seen;
    auto touch = [&](uint32_t op, uint32_t off) {
      if (op < (uint32_t)kMaxOperands && !seen.count({op, off})) {
        seen[{op, off}] = 1;
        dp.polys_in[op]++;
      }
    };
    for (uint32_t r = 0; r < pb.p.nrows; ++r) {
      const Row& row = pb.p.rows[r];
      for (uint32_t t = 0; t < (uint32_t)row.nterms + row.nshift; ++t) {
        const Term& tm = pb.p.terms[row.term0 + t];
        touch(tm.b_op, tm.b_off);
        if ((tm.kind & TERM_KIND_MASK) != TERM_KEY) touch(tm.a_op, tm.a_off);
      }
      for (uint32_t a = 0; a < row.nadds; ++a) touch(pb.p.adds[row.add0 + a].op & ADD_OP_MASK, pb.p.adds[row.add0 + a].off);
      if (row.mode == MODE_STORE) dp.polys_out++;
    }
  }
  c->progs[{id, var}] = dp;
  out = dp;
  return RZK_OK;
}

void drop_programs(rzk_ctx* c) {
  for (auto& kv : c->progs) {
    if (kv.second.d) (void)hipFree(kv.second.d);
    if (kv.second.d_slots) (void)hipFree(kv.second.d_slots);
    if (kv.second.d_blocks) (void)hipFree(kv.second.d_blocks);
    if (kv.second.d_wp) (void)hipFree(kv.second.d_wp);
  }
  c->progs.clear();
}

struct OpSpec {
  const int64_t* base;
  uint32_t stride;
  uint32_t outer;   // 0: indexed by the task batch index; otherwise by the proof index b / group
};

// `group` > 1: the batch is B*group (proof, summand) pairs; operands with outer != 0 and the flags are per proof
// sticky: a non-canonical input coefficient fails the CALL (prover-side / Mat-level entry points); verifier-side
// programs pass false — there the offending proof's verdict flag is cleared and the call succeeds.
int check_launch(rzk_ctx* c, int lrc, const char* what);

// preset_value != 0: every verdict flag (nflags of them) starts at that value — written by the launch itself when one
// team evaluates all rows of a batch entry (no 5-us fill launch in front of a 75-us kernel), by a fill launch otherwise.
int run_program(rzk_ctx* c, int id, uint32_t var, const std::vector<OpSpec>& specs, uint8_t* flags,
                uint32_t group, uint64_t batch, uint64_t norm_limit = 0, bool sticky = true, uint8_t preset_value = 0,
                uint64_t nflags = 0) {
  DevProg dp;
  int rc = get_program(c, id, var, dp);
  if (rc != RZK_OK) return rc;
  if (specs.size() > (size_t)kMaxOperands) return fail(c, RZK_E_ARG, "too many operands");
  // units of one batch entry per task: all of them (one team per entry: equal-cost tasks, no tail) once the batch alone
  // fills the chip's wave slots; one unit per task below that
  uint32_t upt = batch >= (uint64_t)c->num_cus * 16 ? dp.nunits : 1;
  if (c->units_per_task) upt = c->units_per_task;   // RZK_UPT (tuning)
  const bool row_path = dp.has_dkey || (dp.has_vec && c->vec_rows);   // row_kernel: vector x vector products, prepared multiplier images
  const bool unit_path = !c->small && !dp.shift && !dp.nblocks && !dp.ngroups && !dp.d_slots && !row_path;
  const bool preset_in_kernel = preset_value && flags && c->preset_in_kernel && unit_path && upt >= dp.nunits && dp.nunits > 0 &&
                                (group ? group : 1) == 1 && nflags == batch;
  if (preset_value && flags && !preset_in_kernel) {
    rc = check_launch(c, launch_fill_u8(cfg_of(c), flags, preset_value, nflags), "flag preset");
    if (rc != RZK_OK) return rc;
  }
  Operands ops{};
  for (size_t i = 0; i < specs.size(); ++i) {
    ops.base[i] = const_cast<int64_t*>(specs[i].base);
    ops.stride[i] = specs[i].stride;
    ops.outer[i] = specs[i].outer ? 1 : 0;
  }
  ops.group = group ? group : 1;
  ops.pad = dp.two_bit ? 1 : 0;
  ops.norm_limit = norm_limit;
  ops.bad = sticky ? c->d_bad : nullptr;
  ops.trusted = c->trusted ? 1u : 0u;
  ops.preset = preset_in_kernel ? preset_value : 0u;
  if (c->oimg_state.oimg) {   // operand images of the call in progress: this launch stores (producer op) and / or reads them
    ops.oimg = c->oimg_state.oimg;
    ops.oimg_l2 = c->oimg_state.oimg_l2;
    ops.oimg_np = c->oimg_state.oimg_np;
    ops.oimg_n = c->oimg_state.oimg_n;
    ops.oimg_group = c->oimg_state.oimg_group;
    ops.oimg_k = c->oimg_state.oimg_k;
    std::memcpy(ops.oimg_col, c->oimg_state.oimg_col, sizeof(ops.oimg_col));
    ops.oimg_op = c->oimg_producer_op;
  } else {
    ops.oimg_op = 0xffu;
  }
  if (dp.has_dkey) {
    if (!c->dkey_img || c->small) return fail(c, RZK_E_STATE, "multiplier images not prepared");
    ops.dkey_img = c->dkey_img;
    ops.dkey_l2 = c->dkey_l2;
    ops.dkey_n = c->dkey_n;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->prof) {
    if (c->prof_used == c->prof_events.size()) {
      hipEvent_t a, b;
      // no system-scope fence at the events: a default event flushes the caches to make results visible to the
      // host, which slows the NEXT kernel (measured 138 -> 162 us for the commit rows)
      HIPCHK(c, hipEventCreateWithFlags(&a, hipEventDisableSystemFence));
      HIPCHK(c, hipEventCreateWithFlags(&b, hipEventDisableSystemFence));
      c->prof_events.push_back({a, b});
    }
    e0 = c->prof_events[c->prof_used].first;
    e1 = c->prof_events[c->prof_used].second;
    if (c->prof_info.size() <= c->prof_used) c->prof_info.resize(c->prof_used + 1);
    rzk_ctx::ProfInfo& pi = c->prof_info[c->prof_used];
    const std::string L = std::to_string(c->logn);
    const bool pairs = c->logn == 11 && c->pair_poly;   // two wavefronts per polynomial
    const std::string tf = c->trusted ? "true" : "false";
    if (c->small) pi.kernel = "row_kernel_small";
    else if (dp.shift) pi.kernel = "shift_row_kernel<" + L + ", " + tf + (pairs ? ", PairTeam>" : ">");
    else if (dp.nblocks) pi.kernel = "row_block_kernel<" + L + (pairs ? ", BlockPairTeam>" : ">");
    else if (dp.ngroups) pi.kernel = "row_group_kernel<" + L + ", " + std::to_string(c->logn >= 11 ? 2 : RZK_GROUP_GM) + ">";
    else if (dp.d_slots) pi.kernel = "fwd_slots_kernel<" + L + "> + row_slots_kernel<" + L + ">";
    else if (row_path)
      pi.kernel = "row_kernel<" + L + ", " + (dp.has_shift ? "true" : "false") + (pairs ? ", PairTeam" : ", WaveTeam") +
                  (dp.has_dd && !dp.has_shift ? ", true>" : ", false>");
    else if (!dp.has_vec && c->unit_io && !(c->logn == 11 && dp.has_shift))
      pi.kernel = "unit_io_kernel<" + L + ", " + (dp.has_shift ? "true" : "false") + (pairs ? ", PairTeam>" : ">");
    else
      pi.kernel = "unit_kernel<" + L + ", " + (dp.has_vec ? "true" : "false") + ", " + (dp.has_shift ? "true" : "false") +
                  (pairs ? ", PairTeam>" : ">");
    pi.bytes = 0;
    for (size_t i = 0; i < specs.size(); ++i)
      pi.bytes += (uint64_t)dp.polys_in[i] * (specs[i].outer ? batch / (group ? group : 1) : batch);
    pi.bytes = (pi.bytes + (uint64_t)dp.polys_out * batch) * 8ull * c->N;
    c->prof_used++;
    HIPCHK(c, hipEventRecord(e0, c->stream));
  }
  int lrc = 0;
  if (c->small) {
    lrc = launch_row_program_small(c->N, cfg_of(c), dp.d, dp.nrows, ops, c->d_key_mont, c->dT, c->r2q, flags, batch);
  } else if (dp.shift) {
    lrc = launch_shift_rows((int)c->logn, cfg_of(c), dp.d, dp.nrows, ops, c->dT, flags, batch);
  } else if (dp.nblocks) {
    if (!c->d_block_scratch)
      HIPCHK(c, hipMalloc((void**)&c->d_block_scratch, block_scratch_words((int)c->logn, c->num_cus) * sizeof(uint32_t)));
    lrc = launch_row_blocks((int)c->logn, cfg_of(c), dp.d, dp.d_blocks, dp.nblocks, ops, c->d_key_ntt, c->d_key_l2, c->dT,
                            c->d_tw, c->d_block_scratch, flags, batch);
  } else if (dp.ngroups) {
    if (!c->d_group_scratch)
      HIPCHK(c, hipMalloc((void**)&c->d_group_scratch, group_scratch_words((int)c->logn, c->num_cus) * sizeof(uint32_t)));
    lrc = launch_row_groups((int)c->logn, cfg_of(c), dp.d, dp.ngroups, ops, c->d_key_ntt, c->d_key_l2, c->dT, c->d_tw,
                            c->d_group_scratch, flags, batch);
  } else if (dp.d_slots) {
    // shared-operand path; the workspace of stored transforms is bounded, so large batches go in chunks
    const size_t per_item = (size_t)dp.nslots * dp.np_store * c->N * sizeof(uint32_t) + (size_t)dp.nslots * 16;
    const size_t cap = (size_t)6 << 30;
    uint64_t chunk = cap / per_item;
    const uint64_t grp = ops.group;
    chunk -= chunk % grp;
    if (chunk < grp) chunk = grp;
    if (chunk > batch) chunk = batch;
    int rc2 = arena_reserve(c, c->ws_slots, (size_t)chunk * per_item + 256);
    if (rc2 != RZK_OK) return rc2;
    uint32_t* d_ws = (uint32_t*)c->ws_slots.p;
    double* d_norms = (double*)((char*)c->ws_slots.p + (((size_t)chunk * dp.nslots * dp.np_store * c->N * 4 + 255) & ~(size_t)255));
    for (uint64_t b0 = 0; b0 < batch && lrc == 0; b0 += chunk) {
      const uint64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
      Operands o2 = ops;
      for (size_t i = 0; i < specs.size(); ++i) {
        if (!o2.base[i]) continue;
        const uint64_t first = o2.outer[i] ? b0 / grp : b0;
        o2.base[i] += first * o2.stride[i] * c->N;
      }
      lrc = launch_row_program_slots((int)c->logn, cfg_of(c), dp.d, dp.d_slots, dp.nslots, o2, c->d_key_ntt,
                                     c->d_key_l2, c->dT, c->d_tw, d_ws, d_norms, c->d_row_scratch,
                                     flags ? flags + b0 / grp : nullptr, nb, dp.np_store);
    }
  } else {
    if (row_path) {
      lrc = launch_rows((int)c->logn, cfg_of(c), dp.d, dp.nrows, dp.has_shift, ops, c->d_key_ntt, c->d_key_l2, c->dT, c->d_tw,
                        c->d_row_scratch, flags, batch, dp.has_dd);
    } else {
      lrc = launch_units((int)c->logn, cfg_of(c), dp.d, dp.d_wp, dp.nunits, upt, 2 * dp.work, dp.has_vec, dp.has_shift, ops,
                         c->d_key_ntt, c->d_key_l2, c->dT, c->d_tw, c->d_row_scratch, flags, batch);
    }
  }
  if (lrc == -2) return fail(c, RZK_E_UNSUPPORTED, "batch * rows must stay below 2^32");
  if (lrc != 0) {
    c->err = std::string("row kernel launch: ") + (lrc > 0 ? hipGetErrorString((hipError_t)lrc) : "bad ring degree");
    return RZK_E_HIP;
  }
  if (c->prof) HIPCHK(c, hipEventRecord(e1, c->stream));
  return RZK_OK;
}

int check_launch(rzk_ctx* c, int lrc, const char* what) {
  if (lrc == 0) return RZK_OK;
  c->err = std::string(what) + ": " + (lrc > 0 ? hipGetErrorString((hipError_t)lrc) : "bad ring degree");
  return RZK_E_HIP;
}

// (bound+1)^2 as hi:lo
void norm_limit(uint64_t bound, uint64_t& hi, uint64_t& lo) {
  unsigned __int128 v = (unsigned __int128)(bound + 1) * (bound + 1);
  hi = (uint64_t)(v >> 64);
  lo = (uint64_t)v;
}

// (bound+1)^2 when the fused predicate applies (NTT kernels only, limit <= 2^48), else 0
uint64_t fused_limit(const rzk_ctx* c, uint64_t bound) {
  if (c->small || bound >= (1ull << 24) - 1) return 0;
  return (bound + 1) * (bound + 1);
}

// Runs the checked variant (var | 1) of a program with the norm predicate fused into it: flags are
// preset to 1 and cleared by failing rows.  Returns RZK_E_UNSUPPORTED when fusion is not possible.
int run_program_checked(rzk_ctx* c, int id, uint32_t var, const std::vector<OpSpec>& specs, uint8_t* flags,
                        uint32_t group, uint64_t batch, uint64_t nflags, uint64_t bound, bool preset = true,
                        bool sticky = true) {
  const uint64_t lim = fused_limit(c, bound);
  if (!lim || !flags) return RZK_E_UNSUPPORTED;
  DevProg dp;
  int rc = get_program(c, id, var | 1, dp);
  if (rc != RZK_OK) return rc;
  // two-bit verdicts are cleared with word atomics: the flag array must be made of whole aligned words
  if (dp.two_bit && ((reinterpret_cast<uintptr_t>(flags) & 3u) || (nflags & 3u))) return RZK_E_UNSUPPORTED;
  const uint8_t all_ok = dp.two_bit ? 3 : 1;
  return run_program(c, id, var | 1, specs, flags, group, batch, lim, sticky, preset ? all_ok : (uint8_t)0, nflags);
}

// The scalar multipliers of a Linear / Sum call (g: one per proof; g_i: V per proof) transformed once into the form of the
// resident key, for the TERM_DKEY terms of the call's row programs (variant bit kDkeyVar).  flags / sticky as in
// run_program: a non-canonical coefficient clears the proof's verdict and / or raises the sticky word.  Leaves
// c->dkey_img NULL when the path is off (RZK_DKEY=0, small rings): the callers then ask for the plain variants.
// uses: rows of this call that multiply by each multiplier.  The forward launch costs three transforms per multiplier and
// a launch of its own; measured: Linear at l = 1 (2 uses) gains nothing, the Sum shapes (9 and 17 uses) 14-15 %.
constexpr uint32_t kDkeyMinUses = 4;
bool dkey_wanted(const rzk_ctx* c, uint32_t uses) {
  return !(c->use_dkey <= 0 || (c->use_dkey == 1 && uses < kDkeyMinUses) || c->small);
}
int prepare_dkey(rzk_ctx* c, const int64_t* g, uint64_t entries, uint32_t per_entry, uint32_t uses, uint8_t* flags, bool sticky) {
  c->dkey_img = nullptr;
  c->dkey_l2 = nullptr;
  c->dkey_n = 0;
  if (!dkey_wanted(c, uses) || entries == 0) return RZK_OK;
  const uint64_t count = entries * per_entry;
  const size_t img_bytes = ((size_t)count * kKeyImages * c->N * sizeof(uint32_t) + 255) & ~(size_t)255;
  int rc = arena_reserve(c, c->ws_dkey, img_bytes + (size_t)count * sizeof(double));
  if (rc != RZK_OK) return rc;
  uint32_t* img = (uint32_t*)c->ws_dkey.p;
  double* l2 = (double*)((char*)c->ws_dkey.p + img_bytes);
  hipEvent_t e1 = nullptr;
  if (c->prof) {
    if (c->prof_used == c->prof_events.size()) {
      hipEvent_t a, b;
      HIPCHK(c, hipEventCreateWithFlags(&a, hipEventDisableSystemFence));
      HIPCHK(c, hipEventCreateWithFlags(&b, hipEventDisableSystemFence));
      c->prof_events.push_back({a, b});
    }
    hipEvent_t e0 = c->prof_events[c->prof_used].first;
    e1 = c->prof_events[c->prof_used].second;
    if (c->prof_info.size() <= c->prof_used) c->prof_info.resize(c->prof_used + 1);
    rzk_ctx::ProfInfo& pi = c->prof_info[c->prof_used];
    pi.kernel = "dkey_transform_kernel<" + std::to_string(c->logn) + ">";
    pi.bytes = count * 8ull * c->N;   // the multipliers it reads; the images are derived data
    c->prof_used++;
    HIPCHK(c, hipEventRecord(e0, c->stream));
  }
  rc = check_launch(c, launch_dkey_transform((int)c->logn, cfg_of(c), g, count, per_entry, img, l2, c->dT, c->d_tw, flags,
                                             sticky ? c->d_bad : nullptr, false, c->trusted), "multiplier images");
  if (rc != RZK_OK) return rc;
  if (e1) HIPCHK(c, hipEventRecord(e1, c->stream));
  c->dkey_img = img;
  c->dkey_l2 = l2;
  c->dkey_n = per_entry;
  return RZK_OK;
}
// variant bit for programs that multiply by the call's scalar multipliers
uint32_t dkv(const rzk_ctx* c) { return c->dkey_img ? kDkeyVar : 0u; }

// Operand images of a Sum call (Operands::oimg): room for the transforms of the columns a2 uses of every (proof, summand)
// vector, all marked "nothing stored yet".  Leaves c->oimg_state.oimg NULL when the path is off or nothing would be saved.
bool sum_uses_d(const rzk_ctx* c, uint32_t V);   // (below, with the Sum entry points)
int prepare_oimg(rzk_ctx* c, uint64_t B, uint32_t V, uint32_t uses) {
  c->oimg_state = Operands{};
  c->oimg_producer_op = 0xffu;
  if (!c->use_oimg || !dkey_wanted(c, uses) || c->k > 32u || !sum_uses_d(c, V)) return RZK_OK;
  const uint32_t n = c->n, k = c->k, l = c->l;
  uint32_t ncols = 0;
  for (uint32_t col = 0; col < 32u; ++col) c->oimg_state.oimg_col[col] = -1;
  for (uint32_t col = 0; col < k; ++col) {
    bool used = false;
    for (uint32_t j = 0; j < l; ++j) used = used || c->key_class[(n + j) * k + col] != KC_ZERO;
    if (used) c->oimg_state.oimg_col[col] = (int8_t)ncols++;
  }
  if (ncols == 0) return RZK_OK;
  const uint64_t slots = B * V * ncols;
  const size_t img_bytes = ((size_t)slots * kKeyImages * c->N * sizeof(uint32_t) + 255) & ~(size_t)255;
  const size_t l2_bytes = ((size_t)slots * sizeof(double) + 255) & ~(size_t)255;
  int rc = arena_reserve(c, c->ws_oimg, img_bytes + l2_bytes + slots);
  if (rc != RZK_OK) return rc;
  char* base = (char*)c->ws_oimg.p;
  HIPCHK(c, hipMemsetAsync(base + img_bytes + l2_bytes, 0, slots, c->stream));
  c->oimg_state.oimg = (uint32_t*)base;
  c->oimg_state.oimg_l2 = (double*)(base + img_bytes);
  c->oimg_state.oimg_np = (uint8_t*)(base + img_bytes + l2_bytes);
  c->oimg_state.oimg_n = ncols;
  c->oimg_state.oimg_group = V;
  c->oimg_state.oimg_k = k;
  return RZK_OK;
}
uint32_t oiv(const rzk_ctx* c) { return (c->oimg_state.oimg && c->dkey_img) ? kOimgVar : 0u; }

bool can_fuse(rzk_ctx* c, int id, uint32_t var, uint64_t bound) {
  DevProg dp;
  return fused_limit(c, bound) != 0 && get_program(c, id, var | 1, dp) == RZK_OK;
}

int run_norm(rzk_ctx* c, const int64_t* v, uint32_t rows, uint64_t bound, uint8_t* ok, uint64_t B, int mode,
             int shift, bool sticky = true) {
  // canonical coefficients are below 2^31 in magnitude, so sum c^2 < 2^73: every bound from 2^37 on holds for
  // all inputs; clamping there keeps (bound+1)^2 inside 128 bits for any u64 bound (sigma grows with b)
  if (bound > (1ull << 40)) bound = 1ull << 40;
  uint64_t hi, lo;
  norm_limit(bound, hi, lo);
  uint32_t* bad = sticky ? c->d_bad : nullptr;
  const uint32_t qh = c->hT.crt.qhalf;
  if (c->small)
    return check_launch(c, launch_norm_small(c->N, cfg_of(c), v, rows, hi, lo, ok, B, mode, shift, qh, bad), "norm kernel");
  return check_launch(c, launch_norm((int)c->logn, cfg_of(c), v, rows, hi, lo, ok, B, mode, shift, qh, bad), "norm kernel");
}

// ---- host-pointer plumbing ---------------------------------------------------------------------------------
struct HostBuf {
  const void* in;   // host source (nullptr for pure outputs)
  void* out;        // host destination (nullptr for pure inputs)
  size_t bytes;
  void* dev;        // filled by stage_in
};

int stage_in(rzk_ctx* c, std::vector<HostBuf>& bufs) {
  size_t total = 0;
  for (auto& b : bufs) total += (b.bytes + 255) & ~size_t(255);
  int rc = arena_reserve(c, c->stage, total);
  if (rc != RZK_OK) return rc;
  size_t off = 0;
  for (auto& b : bufs) {
    b.dev = (char*)c->stage.p + off;
    off += (b.bytes + 255) & ~size_t(255);
    if (b.in && b.bytes) HIPCHK(c, hipMemcpyAsync(b.dev, b.in, b.bytes, hipMemcpyHostToDevice, c->stream));
  }
  return RZK_OK;
}

// Synchronises and reports (then clears) the sticky input-error word.
int take_input_error(rzk_ctx* c) {
  HIPCHK(c, hipMemcpyAsync(c->h_bad, c->d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (*c->h_bad == 0) return RZK_OK;
  *c->h_bad = 0;
  HIPCHK(c, hipMemsetAsync(c->d_bad, 0, sizeof(uint32_t), c->stream));
  return fail(c, RZK_E_ARG,
              "an input coefficient is not the centred representative in [-(q-1)/2, (q-1)/2] (ZqI64 range); "
              "reduce foreign data with rzk_canonicalize_batch first");
}

int stage_out(rzk_ctx* c, std::vector<HostBuf>& bufs) {
  for (auto& b : bufs)
    if (b.out && b.bytes) HIPCHK(c, hipMemcpyAsync(b.out, b.dev, b.bytes, hipMemcpyDeviceToHost, c->stream));
  return take_input_error(c);
}

size_t polys(const rzk_ctx* c, size_t count) { return count * (size_t)c->N * sizeof(int64_t); }

}  // namespace

// =================================================================================================
// context
// =================================================================================================
extern "C" {

int rzk_ctx_create(rzk_ctx** out, int64_t q, uint32_t N, uint32_t n, uint32_t k, uint32_t l, uint32_t kappa,
                   uint64_t b, int device) {
  if (!out) return RZK_E_ARG;
  *out = nullptr;
  const bool pow2 = N >= 4 && (N & (N - 1)) == 0;
  if (!pow2 || N > 2048) return create_fail(RZK_E_UNSUPPORTED, "ring degree must be a power of two in [4, 2048]");
  if (n < 1 || l < 1 || k <= n || n + l > k)   // params.rs:26-31: k > n >= l ; a2' has k-n-l cols
    return create_fail(RZK_E_ARG, "need k > n >= 1, l >= 1, n + l <= k");
  if (q < 3) return create_fail(RZK_E_ARG, "bad modulus");
  int ndev = 0;
  hipError_t de = hipGetDeviceCount(&ndev);
  if (de != hipSuccess) return create_fail(RZK_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(de));
  if (ndev <= 0 || device < 0 || device >= ndev) return create_fail(RZK_E_HIP, "no such HIP device");
  rzk_ctx* c = new rzk_ctx();
  c->device = device;
  c->q = q;
  c->N = N;
  c->logn = 0;
  while ((1u << c->logn) < N) ++c->logn;
  c->small = N < 512;
  c->r2q = (uint32_t)((((unsigned __int128)1) << 64) % (unsigned __int128)q);
  c->n = n;
  c->k = k;
  c->l = l;
  c->kappa = kappa;
  c->b = b;
  // params.rs:94-98, 104, 114 (usize floor square roots)
  c->sigma = b * (11ull * kappa) * isqrt_u64((uint64_t)k * N);
  c->commit_bound = 4 * c->sigma * isqrt_u64(N);
  c->verify_bound = 2 * c->sigma * isqrt_u64(N);
  if (!host::make_crt_consts((uint64_t)q, c->hT.crt)) {
    delete c;
    return create_fail(RZK_E_UNSUPPORTED, "modulus must be odd and below 4 * the smallest auxiliary prime");
  }
  if ((de = hipSetDevice(device)) != hipSuccess) {
    delete c;
    return create_fail(RZK_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(de));
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    c->num_cus = prop.multiProcessorCount;
  if ((de = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    delete c;
    return create_fail(RZK_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(de));
  }
  c->stream = c->own_stream;
  if (const char* e = std::getenv("RZK_SLOT_SHARE_MIN")) c->slot_share_min = std::atof(e);   // tuning knobs
  if (const char* e = std::getenv("RZK_ROW_GROUPS")) c->use_groups = std::atoi(e) != 0;
  c->group_max = group_max_for((int)c->logn);
  if (const char* e = std::getenv("RZK_GROUP_MAX")) {
    const int g = std::atoi(e);
    if (g >= 1 && g <= (c->logn >= 11 ? 2 : RZK_GROUP_GM)) c->group_max = g;   // bounded by the compiled accumulators
  }
  if (const char* e = std::getenv("RZK_SHIFT")) c->use_shift = std::atoi(e) != 0;
  if (const char* e = std::getenv("RZK_SUM_D")) c->sum_d = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("RZK_DKEY")) c->use_dkey = std::atoi(e);
  if (const char* e = std::getenv("RZK_LIN_E")) c->lin_e = std::atoi(e) != 0;
  if (const char* e = std::getenv("RZK_OIMG")) c->use_oimg = std::atoi(e) != 0;
  if (const char* e = std::getenv("RZK_PRESET_IN_KERNEL")) c->preset_in_kernel = std::atoi(e) != 0;
  if (const char* e = std::getenv("RZK_PAIRS")) c->use_pairs = std::atoi(e) != 0;
  if (const char* e = std::getenv("RZK_UPT")) c->units_per_task = (uint32_t)std::atoi(e);
  if (const char* e = std::getenv("RZK_VEC_ROWS")) c->vec_rows = std::atoi(e) != 0;
  if (const char* e = std::getenv("RZK_PAIR_POLY")) c->pair_poly = std::atoi(e) != 0;
  c->unit_io = c->logn == 9;
  if (const char* e = std::getenv("RZK_UNIT_IO")) c->unit_io = std::atoi(e) != 0;
  if (const char* e = std::getenv("RZK_BLOCK_MIN_LOGN")) c->block_min_logn = (uint32_t)std::atoi(e);   // 12 = never

  // twiddle tables: 3 primes x {fwd, inv} x kTableLen
  std::vector<uint32_t> all((size_t)2 * kMaxPrimes * kTableLen);
  for (int i = 0; i < kMaxPrimes; ++i) {
    std::vector<uint32_t> f, v;
    host::make_twiddles(i, f, v);
    std::memcpy(&all[(size_t)(2 * i) * kTableLen], f.data(), sizeof(uint32_t) * kTableLen);
    std::memcpy(&all[(size_t)(2 * i + 1) * kTableLen], v.data(), sizeof(uint32_t) * kTableLen);
    c->hT.pc[i] = host::make_prime_consts(i, N);
  }
  bool okk = hipMalloc((void**)&c->d_tw, all.size() * sizeof(uint32_t)) == hipSuccess &&
             hipMemcpy(c->d_tw, all.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
  c->hT.cap[0] = 0.0;
  for (int np = 1; np <= kMaxPrimes; ++np) c->hT.cap[np] = host::crt_capacity(np);
  okk = okk && (c->small || hipMalloc((void**)&c->d_row_scratch,
                                      row_scratch_words((int)c->logn, c->num_cus) * sizeof(uint32_t)) == hipSuccess);
  okk = okk && hipMalloc((void**)&c->dT, sizeof(DevTables)) == hipSuccess &&
        hipMemcpy(c->dT, &c->hT, sizeof(DevTables), hipMemcpyHostToDevice) == hipSuccess;
  okk = okk && hipMalloc((void**)&c->d_bad, sizeof(uint32_t)) == hipSuccess &&
        hipMemset(c->d_bad, 0, sizeof(uint32_t)) == hipSuccess &&
        hipHostMalloc((void**)&c->h_bad, sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
  if (okk) *c->h_bad = 0;
  if (!okk) {
    rzk_ctx_destroy(c);
    return create_fail(RZK_E_HIP, std::string("table upload: ") + hipGetErrorString(hipGetLastError()));
  }
  *out = c;
  return RZK_OK;
}

void rzk_ctx_destroy(rzk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drop_programs(c);
  for (auto& ev : c->prof_events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (c->d_key_ntt) (void)hipFree(c->d_key_ntt);
  if (c->d_key_l2) (void)hipFree(c->d_key_l2);
  if (c->ws.p) (void)hipFree(c->ws.p);
  if (c->ws_slots.p) (void)hipFree(c->ws_slots.p);
  if (c->ws_dkey.p) (void)hipFree(c->ws_dkey.p);
  if (c->ws_oimg.p) (void)hipFree(c->ws_oimg.p);
  if (c->stage.p) (void)hipFree(c->stage.p);
  if (c->dT) (void)hipFree(c->dT);
  if (c->d_tw) (void)hipFree(c->d_tw);
  if (c->d_row_scratch) (void)hipFree(c->d_row_scratch);
  if (c->d_block_scratch) (void)hipFree(c->d_block_scratch);
  if (c->d_group_scratch) (void)hipFree(c->d_group_scratch);
  if (c->d_key_mont) (void)hipFree(c->d_key_mont);
  if (c->d_bad) (void)hipFree(c->d_bad);
  if (c->h_bad) (void)hipHostFree(c->h_bad);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

uint32_t rzk_abi_version(void) { return RZK_ABI_VERSION; }

int rzk_ctx_trust_device_outputs(rzk_ctx* c, int on) {
  if (!c) return RZK_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->trusted = on != 0;
  return RZK_OK;
}

int rzk_ctx_set_stream(rzk_ctx* c, void* hip_stream) {
  if (!c) return RZK_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = (hipStream_t)hip_stream;   // NULL = HIP's default stream
  return RZK_OK;
}

int rzk_ctx_use_own_stream(rzk_ctx* c) {
  if (!c) return RZK_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = c->own_stream;
  return RZK_OK;
}

int rzk_ctx_synchronize(rzk_ctx* c) {
  if (!c) return RZK_E_ARG;
  (void)hipSetDevice(c->device);
  return take_input_error(c);   // also the place where the asynchronous *_dev calls report non-canonical inputs
}

int rzk_ctx_check_inputs(rzk_ctx* c) { return rzk_ctx_synchronize(c); }

const char* rzk_last_error(const rzk_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }
uint64_t rzk_sigma(const rzk_ctx* c) { return c ? c->sigma : 0; }
uint64_t rzk_commit_bound(const rzk_ctx* c) { return c ? c->commit_bound : 0; }
uint64_t rzk_verify_bound(const rzk_ctx* c) { return c ? c->verify_bound : 0; }

// =================================================================================================
// key
// =================================================================================================
static int key_load_impl(rzk_ctx* c, const int64_t* a_host) {
  const uint32_t N = c->N, rows = c->n + c->l, k = c->k;
  const size_t total = (size_t)rows * k;
  const int64_t half = (c->q - 1) / 2;
  c->key_class.assign(total, KC_GENERAL);
  c->key_entry.assign(total, -1);
  std::vector<int64_t> general;
  std::vector<double> kinf;
  c->n_general = 0;
  for (size_t e = 0; e < total; ++e) {
    const int64_t* p = a_host + e * N;
    bool tail_zero = true;
    long double ssq = 0.0L;
    for (uint32_t j = 0; j < N; ++j) {
      const int64_t v = p[j];
      if (v > half || v < -half) return fail(c, RZK_E_ARG, "key coefficient outside the centred range");
      if (j > 0 && v != 0) tail_zero = false;
      ssq += (long double)v * (long double)v;
    }
    if (tail_zero && p[0] == 0)
      c->key_class[e] = KC_ZERO;
    else if (tail_zero && p[0] == 1)
      c->key_class[e] = KC_ONE;
    else {
      c->key_entry[e] = (int32_t)c->n_general++;
      general.insert(general.end(), p, p + N);
      kinf.push_back((double)(std::sqrt(ssq) * (1.0L + 1e-6L)));   // upper bound of the entry's 2-norm (prime-count bound), still one after rounding to float
    }
  }
  drop_programs(c);   // programs depend on the classification
  if (c->d_key_ntt) HIPCHK(c, hipFree(c->d_key_ntt));
  if (c->d_key_l2) HIPCHK(c, hipFree(c->d_key_l2));
  if (c->d_key_mont) HIPCHK(c, hipFree(c->d_key_mont));
  c->d_key_ntt = nullptr;
  c->d_key_l2 = nullptr;
  c->d_key_mont = nullptr;
  if (c->n_general && c->small) {
    const size_t gbytes = general.size() * sizeof(int64_t);
    int rc = arena_reserve(c, c->stage, gbytes);
    if (rc != RZK_OK) return rc;
    HIPCHK(c, hipMalloc((void**)&c->d_key_mont, general.size() * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpyAsync(c->stage.p, general.data(), gbytes, hipMemcpyHostToDevice, c->stream));
    rc = check_launch(c, launch_key_mont(cfg_of(c), (const int64_t*)c->stage.p, c->d_key_mont, general.size(), c->dT,
                                         c->r2q),
                      "key conversion");
    if (rc != RZK_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  } else if (c->n_general) {
    const size_t gbytes = general.size() * sizeof(int64_t);
    int rc = arena_reserve(c, c->stage, gbytes);
    if (rc != RZK_OK) return rc;
    HIPCHK(c, hipMalloc((void**)&c->d_key_ntt, (size_t)c->n_general * kKeyImages * N * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_key_l2, (size_t)c->n_general * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(c->stage.p, general.data(), gbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_key_l2, kinf.data(), kinf.size() * sizeof(double), hipMemcpyHostToDevice,
                             c->stream));
    rc = check_launch(c, launch_key_transform((int)c->logn, cfg_of(c), (const int64_t*)c->stage.p, c->n_general,
                                              c->d_key_ntt, c->dT, c->d_tw),
                      "key transform");
    if (rc != RZK_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));   // `general` / `kinf` are host temporaries
  }
  c->key_loaded = true;
  return RZK_OK;
}

int rzk_key_load(rzk_ctx* c, const int64_t* a_host) {
  if (!c || !a_host) return RZK_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  return key_load_impl(c, a_host);
}

// CommitmentKey::new (commit.rs:33-60) with the device-side sampler: a1 = [I_n | U], a2 = [0_{l x n} | I_l | U],
// U uniform over the centred range (params.rs:126).  The full key goes back to the caller (the reference's
// CommitmentKey is public data) and is loaded like any other key.
int rzk_key_generate(rzk_ctx* c, uint64_t seed, int64_t* a_host_out) {
  if (!c) return RZK_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t N = c->N, n = c->n, k = c->k, l = c->l;
  const size_t n_rand = (size_t)n * (k - n) + (size_t)l * (k - n - l);
  std::vector<int64_t> rnd(n_rand * N);
  if (n_rand) {
    int rc = arena_reserve(c, c->stage, n_rand * N * sizeof(int64_t));
    if (rc != RZK_OK) return rc;
    rc = check_launch(c, launch_sample_uniform(cfg_of(c), (int64_t*)c->stage.p, n_rand, N, seed, 0x6b6579u /* "key" */,
                                               (uint32_t)((c->q - 1) / 2)),
                      "key sampler");
    if (rc != RZK_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(rnd.data(), c->stage.p, rnd.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  std::vector<int64_t> a((size_t)(n + l) * k * N, 0);
  size_t next = 0;
  auto put_random = [&](size_t row, size_t col) {
    std::memcpy(a.data() + (row * k + col) * N, rnd.data() + next * N, (size_t)N * sizeof(int64_t));
    ++next;
  };
  for (uint32_t i = 0; i < n; ++i) {          // a1 = [I_n | a1']   (commit.rs:38-46)
    a[((size_t)i * k + i) * N] = 1;
    for (uint32_t j = n; j < k; ++j) put_random(i, j);
  }
  for (uint32_t i = 0; i < l; ++i) {          // a2 = [0 | I_l | a2']  (commit.rs:48-57)
    a[((size_t)(n + i) * k + n + i) * N] = 1;
    for (uint32_t j = n + l; j < k; ++j) put_random(n + i, j);
  }
  if (a_host_out) std::memcpy(a_host_out, a.data(), a.size() * sizeof(int64_t));
  return key_load_impl(c, a.data());
}

int rzk_key_load_dev(rzk_ctx* c, const int64_t* a_dev) {
  if (!c || !a_dev) return RZK_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t total = (size_t)(c->n + c->l) * c->k * c->N;
  std::vector<int64_t> h(total);
  HIPCHK(c, hipMemcpyAsync(h.data(), a_dev, total * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return key_load_impl(c, h.data());
}

// =================================================================================================
// ring / Mat primitives
// =================================================================================================
int rzk_polymul_batch_dev(rzk_ctx* c, const int64_t* a, const int64_t* b, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !out) return RZK_E_ARG;
  return run_program(c, PG_POLYMUL, 0, {{a, 1, 0}, {b, 1, 0}, {out, 1, 0}}, nullptr, 1, count);
}

int rzk_matvec_batch_dev(rzk_ctx* c, int which, const int64_t* v, const int64_t* addend, int64_t* out, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !v || !out || which < 0 || which > 2) return RZK_E_ARG;
  const uint32_t rows = which == RZK_KEY_A1 ? c->n : (which == RZK_KEY_A2 ? c->l : c->n + c->l);
  return run_program(c, PG_MATVEC, (uint32_t)which * 2 + (addend ? 1 : 0),
                     {{v, c->k, 0}, {addend, rows, 0}, {out, rows, 0}}, nullptr, 1, B);
}

int rzk_cmul_batch_dev(rzk_ctx* c, const int64_t* m, uint32_t rows, const int64_t* p, int64_t* out, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !m || !p || !out || rows == 0) return RZK_E_ARG;
  if (rows > (uint32_t)kMaxRows) {
    // more rows than one program holds: treat every row as its own batch entry sharing p
    return run_program(c, PG_CMUL, 1, {{m, 1, 0}, {p, 1, 1}, {out, 1, 0}}, nullptr, rows, B * rows);
  }
  return run_program(c, PG_CMUL, rows, {{m, rows, 0}, {p, 1, 0}, {out, rows, 0}}, nullptr, 1, B);
}

int rzk_canonicalize_batch_dev(rzk_ctx* c, const int64_t* in, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;
  if (!c || !in || !out) return RZK_E_ARG;
  return check_launch(c, launch_canonicalize(cfg_of(c), in, out, (uint64_t)count * c->N, c->q), "canonicalize kernel");
}

int rzk_add_batch_dev(rzk_ctx* c, const int64_t* a, const int64_t* b, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !out) return RZK_E_ARG;
  return check_launch(c, launch_addsub(cfg_of(c), false, a, b, out, (uint64_t)count * c->N, c->dT, c->d_bad), "add kernel");
}

int rzk_sub_batch_dev(rzk_ctx* c, const int64_t* a, const int64_t* b, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !out) return RZK_E_ARG;
  return check_launch(c, launch_addsub(cfg_of(c), true, a, b, out, (uint64_t)count * c->N, c->dT, c->d_bad), "sub kernel");
}

int rzk_norm2_le_batch_dev(rzk_ctx* c, const int64_t* v, uint32_t rows, uint64_t bound, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !v || !ok || rows == 0) return RZK_E_ARG;
  return run_norm(c, v, rows, bound, ok, B, 0, 0);
}

int rzk_eq_batch_dev(rzk_ctx* c, const int64_t* a, const int64_t* b, uint32_t rows, uint8_t* eq, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !eq || rows == 0) return RZK_E_ARG;
  const uint32_t qh = c->hT.crt.qhalf;
  if (c->small) return check_launch(c, launch_eq_small(c->N, cfg_of(c), a, b, rows, eq, B, qh, c->d_bad), "eq kernel");
  return check_launch(c, launch_eq((int)c->logn, cfg_of(c), a, b, rows, eq, B, qh, c->d_bad), "eq kernel");
}

int rzk_ntt_forward_batch_dev(rzk_ctx* c, int prime, const uint32_t* in, uint32_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !in || !out || prime < 0 || prime >= kMaxPrimes) return RZK_E_ARG;
  if (c->small) return fail(c, RZK_E_UNSUPPORTED, "batched transforms need N >= 512");
  return check_launch(c, launch_ntt((int)c->logn, false, cfg_of(c), prime, in, out, count, c->dT, c->d_tw), "ntt forward");
}

int rzk_ntt_inverse_batch_dev(rzk_ctx* c, int prime, const uint32_t* in, uint32_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !in || !out || prime < 0 || prime >= kMaxPrimes) return RZK_E_ARG;
  if (c->small) return fail(c, RZK_E_UNSUPPORTED, "batched transforms need N >= 512");
  return check_launch(c, launch_ntt((int)c->logn, true, cfg_of(c), prime, in, out, count, c->dT, c->d_tw), "ntt inverse");
}

uint32_t rzk_ntt_prime(int prime) { return prime >= 0 && prime < kMaxPrimes ? kPrimes[prime] : 0; }
uint32_t rzk_ntt_psi(int prime, uint32_t N) {
  if (prime < 0 || prime >= kMaxPrimes || N == 0 || (N & (N - 1)) || N > (uint32_t)kTableLen) return 0;
  return host::psi_for(prime, N);
}
uint32_t rzk_ntt_layout_index(uint32_t N, uint32_t j) {
  if (j >= N || N < 512) return 0xffffffffu;
  const uint32_t E = N / 64;
  const uint32_t lane = j / E, c = j % E;
  return (c >> 2) * 256 + lane * 4 + (c & 3);
}

// =================================================================================================
// Device-side samplers
// =================================================================================================
int rzk_sample_uniform_dev(rzk_ctx* c, uint64_t seed, uint32_t stream, uint64_t bound, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;
  if (!c || !out || bound == 0 || bound > (uint64_t)(c->q - 1) / 2) return RZK_E_ARG;
  return check_launch(c, launch_sample_uniform(cfg_of(c), out, count, c->N, seed, stream, (uint32_t)bound), "sampler");
}
int rzk_sample_gauss_dev(rzk_ctx* c, uint64_t seed, uint32_t stream, double sigma, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;
  // |x| stays far below (q-1)/2 for every sigma the parameters produce; 2^26 keeps 12 sigma inside the range
  if (!c || !out || !(sigma > 0.0) || sigma > 67108864.0) return RZK_E_ARG;
  return check_launch(c, launch_sample_gauss(cfg_of(c), out, count, c->N, seed, stream, sigma), "sampler");
}
int rzk_sample_challenge_dev(rzk_ctx* c, uint64_t seed, uint32_t stream, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;
  if (!c || !out) return RZK_E_ARG;
  return check_launch(c, launch_sample_challenge(cfg_of(c), out, count, c->N, seed, stream, c->kappa), "sampler");
}

namespace {

// accept &= check_verify_constraint(z) && (a1.z == t + c1 (.) d)      (open.rs:167-173, sum.rs:262-298)
// specs: z[k], t[n], c[n+l], d.  group / batch / nflags as in run_program; preset: initialise accept.
// n == 1: one launch, the d-product is a rotation term inside the relation row.  n >= 2 (and rotations +
// row groups available): a1.z as ONE grouped launch (each z_j transformed once for the n rows, norm predicate
// fused) into `w`, then the relation rows as pure rotations (shift_row_kernel) — 18 instead of 48 transform
// units per (4,9,4) relation.
int run_a1_relation(rzk_ctx* c, const std::vector<OpSpec>& specs, int64_t* w, uint8_t* accept, uint32_t group,
                    uint64_t batch, uint64_t nflags, bool preset) {
  // (at N = 2048 the first step runs as row blocks and the second as transform products)
  const bool blocks = !c->small && c->logn >= 10 && c->logn >= c->block_min_logn;
  const bool split = w && c->n >= 2 && ((shift_ok(c) && c->use_groups && c->group_max > 1) || blocks);
  int rc;
  if (split) {
    const std::vector<OpSpec> a1z = {specs[0], {w, c->n, 0}};
    // verifier side: non-canonical prover data clears the proof's verdict, the call itself succeeds (sticky = false)
    rc = run_program_checked(c, PG_A1Z, 0, a1z, accept, group, batch, nflags, c->verify_bound, preset, false);
    if (rc == RZK_E_UNSUPPORTED) {
      rc = run_norm(c, specs[0].base, group * c->k, c->verify_bound, accept, nflags, preset ? 0 : 1, 0, false);
      if (rc != RZK_OK) return rc;
      rc = run_program(c, PG_A1Z, 0, a1z, accept, group, batch, 0, false);
    }
    if (rc != RZK_OK) return rc;
    return run_program(c, PG_REL_ROT, 0, {{w, c->n, 0}, specs[1], specs[2], specs[3]}, accept, group, batch, 0, false);
  }
  rc = run_program_checked(c, PG_A1_RELATION, 0, specs, accept, group, batch, nflags, c->verify_bound, preset, false);
  if (rc != RZK_E_UNSUPPORTED) return rc;
  rc = run_norm(c, specs[0].base, group * c->k, c->verify_bound, accept, nflags, preset ? 0 : 1, 0, false);
  if (rc != RZK_OK) return rc;
  return run_program(c, PG_A1_RELATION, 0, specs, accept, group, batch, 0, false);
}

}  // namespace

// =================================================================================================
// Commitment scheme
// =================================================================================================
int rzk_commit_batch_dev(rzk_ctx* c, const int64_t* x, const int64_t* r, int64_t* cm, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !x || !r || !cm) return RZK_E_ARG;
  const std::vector<OpSpec> specs = {{x, c->l, 0}, {r, c->k, 0}, {cm, c->n + c->l, 0}};
  int rc = run_program_checked(c, PG_COMMIT, 0, specs, ok, 1, B, B, c->commit_bound);
  if (rc != RZK_E_UNSUPPORTED) return rc;
  // unfused: the exact norm kernel sets ok, then the rows clear it again for proofs with non-canonical inputs
  if (ok) rc = run_norm(c, r, c->k, c->commit_bound, ok, B, 0, 0);
  if (ok && rc != RZK_OK) return rc;
  return run_program(c, PG_COMMIT, 0, specs, ok, 1, B);
}

int rzk_commitment_verify_batch_dev(rzk_ctx* c, const int64_t* cm, const int64_t* x, const int64_t* r,
                                    const int64_t* f, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !cm || !x || !r || !ok) return RZK_E_ARG;
  const uint32_t var = f ? 2u : 0u;
  const std::vector<OpSpec> specs = {{x, c->l, 0}, {r, c->k, 0}, {cm, c->n + c->l, 0}, {f, 1, 0}};
  // commit.rs:183-185: the norm predicate on r rides on the rows that load r
  int rc = run_program_checked(c, PG_COMMIT_VERIFY, var, specs, ok, 1, B, B, c->commit_bound, true, false);
  if (rc != RZK_E_UNSUPPORTED) return rc;
  rc = run_norm(c, r, c->k, c->commit_bound, ok, B, 0, 0, false);
  if (rc != RZK_OK) return rc;
  return run_program(c, PG_COMMIT_VERIFY, var, specs, ok, 1, B, 0, false);
}

// =================================================================================================
// OpenProof
// =================================================================================================
int rzk_open_commit_batch_dev(rzk_ctx* c, const int64_t* x, const int64_t* r, const int64_t* y, int64_t* cm,
                              int64_t* t, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !x || !r || !y || !cm || !t) return RZK_E_ARG;
  const std::vector<OpSpec> specs = {{x, c->l, 0}, {r, c->k, 0}, {y, c->k, 0}, {cm, c->n + c->l, 0}, {t, c->n, 0}};
  // check_commit_constraint(r) (params.rs:102-108) rides on the loads of r the commit rows do anyway
  int rc = run_program_checked(c, PG_OPEN_COMMIT, 0, specs, ok, 1, B, B, c->commit_bound);
  if (rc != RZK_E_UNSUPPORTED) return rc;
  if (ok) rc = run_norm(c, r, c->k, c->commit_bound, ok, B, 0, 0);
  if (ok && rc != RZK_OK) return rc;
  return run_program(c, PG_OPEN_COMMIT, 0, specs, ok, 1, B);
}

int rzk_open_response_batch_dev(rzk_ctx* c, const int64_t* y, const int64_t* r, const int64_t* d, int64_t* z,
                                size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !y || !r || !d || !z) return RZK_E_ARG;
  return run_program(c, PG_RESPONSE, 1, {{d, 1, 0}, {y, c->k, 0}, {r, c->k, 0}, {z, c->k, 0}}, nullptr, 1, B);
}

int rzk_open_verify_batch_dev(rzk_ctx* c, const int64_t* z, const int64_t* t, const int64_t* cm, const int64_t* d,
                              uint8_t* accept, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !z || !t || !cm || !d || !accept) return RZK_E_ARG;
  if (c->n != c->l) return fail(c, RZK_E_ARG, "c1_c2 split needs n == l (reference panics in Mat::add)");
  const std::vector<OpSpec> specs = {{z, c->k, 0}, {t, c->n, 0}, {cm, c->n + c->l, 0}, {d, 1, 0}};
  int rc = arena_reserve(c, c->ws, polys(c, B * c->n));
  if (rc != RZK_OK) return rc;
  // open.rs:167-173: the norm predicate on z rides on the rows that load z
  return run_a1_relation(c, specs, (int64_t*)c->ws.p, accept, 1, B, B, true);
}

// =================================================================================================
// LinearProof
// =================================================================================================
int rzk_linear_commit_batch_dev(rzk_ctx* c, const int64_t* g, const int64_t* x, const int64_t* r, const int64_t* rp,
                                const int64_t* y, const int64_t* yp, int64_t* cm, int64_t* cpm, int64_t* t,
                                int64_t* tp, int64_t* u, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !g || !x || !r || !rp || !y || !yp || !cm || !cpm || !t || !tp || !u) return RZK_E_ARG;
  const uint32_t n = c->n, k = c->k, l = c->l;
  int rc = arena_reserve(c, c->ws, polys(c, 2 * B * l));
  if (rc != RZK_OK) return rc;
  int64_t* gx = (int64_t*)c->ws.p;
  int64_t* a2y = gx + B * l * c->N;
  // the multiplier g of every proof, transformed once for the rows that multiply by it (gx, u)
  rc = prepare_dkey(c, g, B, 1, 2 * l, nullptr, true);      // l rows of gx, l rows of u
  if (rc != RZK_OK) return rc;
  struct DkeyScope {
    rzk_ctx* c;
    ~DkeyScope() { c->dkey_img = nullptr; c->dkey_l2 = nullptr; c->dkey_n = 0; }
  } dkey_scope{c};
  // linear.rs:91-95: gx = x_i * g
  rc = run_program(c, PG_CMUL, l | dkv(c), {{x, l, 0}, {g, 1, 0}, {gx, l, 0}}, nullptr, 1, B);
  if (rc != RZK_OK) return rc;
  const std::vector<OpSpec> c2 = {{x, l, 0}, {gx, l, 0}, {r, k, 0}, {rp, k, 0}, {y, k, 0}, {yp, k, 0}, {cm, n + l, 0},
                                  {cpm, n + l, 0}, {t, n, 0}, {tp, n, 0}, {a2y, l, 0}};
  // the norm predicates on r (bit 0 of ok) and rp (bit 1) ride on the commit rows when the kernel path allows
  bool fused = false;
  if (ok) {
    rc = run_program_checked(c, PG_LIN_COMMIT2, 0, c2, ok, 1, B, B, c->commit_bound);
    if (rc != RZK_OK && rc != RZK_E_UNSUPPORTED) return rc;
    fused = rc == RZK_OK;
  }
  if (!fused) {
    if (ok) {   // bit 0: constraint(r), bit 1: constraint(rp); the rows below clear the byte on non-canonical inputs
      rc = run_norm(c, r, k, c->commit_bound, ok, B, 0, 0);
      if (rc != RZK_OK) return rc;
      rc = run_norm(c, rp, k, c->commit_bound, ok, B, 2, 1);
      if (rc != RZK_OK) return rc;
    }
    rc = run_program(c, PG_LIN_COMMIT2, 0, c2, ok, 1, B);
    if (rc != RZK_OK) return rc;
  }
  return run_program(c, PG_LIN_U, dkv(c), {{a2y, l, 0}, {g, 1, 0}, {yp, k, 0}, {u, l, 0}}, ok, 1, B);
}

int rzk_linear_response_batch_dev(rzk_ctx* c, const int64_t* y, const int64_t* yp, const int64_t* r,
                                  const int64_t* rp, const int64_t* d, int64_t* z, int64_t* zp, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !y || !yp || !r || !rp || !d || !z || !zp) return RZK_E_ARG;
  const uint32_t k = c->k;
  return run_program(c, PG_RESPONSE, 2, {{d, 1, 0}, {y, k, 0}, {r, k, 0}, {z, k, 0}, {yp, k, 0}, {rp, k, 0}, {zp, k, 0}},
                     nullptr, 1, B);
}

int rzk_linear_verify_batch_dev(rzk_ctx* c, const int64_t* z, const int64_t* zp, const int64_t* cm,
                                const int64_t* cpm, const int64_t* g, const int64_t* t, const int64_t* tp,
                                const int64_t* u, const int64_t* d, uint8_t* accept, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !z || !zp || !cm || !cpm || !g || !t || !tp || !u || !d || !accept) return RZK_E_ARG;
  if (c->n != c->l) return fail(c, RZK_E_ARG, "c1_c2 split needs n == l (reference panics in Mat::add)");
  const uint32_t n = c->n, k = c->k, l = c->l;
  int rc = arena_reserve(c, c->ws, polys(c, 2 * B * l));
  if (rc != RZK_OK) return rc;
  int64_t* w1 = (int64_t*)c->ws.p;
  int64_t* w2 = w1 + B * l * c->N;
  const std::vector<OpSpec> v1 = {{z, k, 0}, {zp, k, 0}, {t, n, 0}, {tp, n, 0}, {cm, n + l, 0}, {cpm, n + l, 0},
                                  {d, 1, 0}, {g, 1, 0}, {w1, l, 0}, {w2, l, 0}};
  if (c->lin_e && !c->small) {
    // rearranged (PG_LIN_V1B / V2B): no product with g in the first program, so it runs on the unit kernel (and
    // initialises the verdicts itself where one team owns a proof); w1 / w2 hold e / e'
    const std::vector<OpSpec> v1b = {{z, k, 0}, {zp, k, 0}, {t, n, 0}, {tp, n, 0}, {cm, n + l, 0}, {cpm, n + l, 0},
                                     {d, 1, 0}, {w1, l, 0}, {w2, l, 0}};
    rc = run_program_checked(c, PG_LIN_V1B, 0, v1b, accept, 1, B, B, c->verify_bound, true, false);
    if (rc == RZK_E_UNSUPPORTED) {
      rc = run_norm(c, z, k, c->verify_bound, accept, B, 0, 0, false);
      if (rc != RZK_OK) return rc;
      rc = run_norm(c, zp, k, c->verify_bound, accept, B, 1, 0, false);
      if (rc != RZK_OK) return rc;
      rc = run_program(c, PG_LIN_V1B, 0, v1b, accept, 1, B, 0, false);
    }
    if (rc != RZK_OK) return rc;
    rc = prepare_dkey(c, g, B, 1, l, accept, false);   // (after the verdicts exist; l rows multiply by g)
    if (rc != RZK_OK) return rc;
    struct DkeyScopeE {
      rzk_ctx* c;
      ~DkeyScopeE() { c->dkey_img = nullptr; c->dkey_l2 = nullptr; c->dkey_n = 0; }
    } dkey_scope_e{c};
    return run_program(c, PG_LIN_V2B, dkv(c), {{w1, l, 0}, {w2, l, 0}, {g, 1, 0}, {u, l, 0}}, accept, 1, B, 0, false);
  }
  // The verdicts are initialised first, then the multiplier's images are prepared (a non-canonical g clears its proof's
  // verdict: nothing may preset the flags after that), then the rows.
  rc = check_launch(c, launch_fill_u8(cfg_of(c), accept, 1, B), "flag preset");
  if (rc != RZK_OK) return rc;
  rc = prepare_dkey(c, g, B, 1, 2 * l, accept, false);
  if (rc != RZK_OK) return rc;
  struct DkeyScope {
    rzk_ctx* c;
    ~DkeyScope() { c->dkey_img = nullptr; c->dkey_l2 = nullptr; c->dkey_n = 0; }
  } dkey_scope{c};
  // linear.rs:218-223: norm predicates on z and zp, fused into the rows that load them when possible
  rc = run_program_checked(c, PG_LIN_V1, dkv(c), v1, accept, 1, B, B, c->verify_bound, false, false);
  if (rc == RZK_E_UNSUPPORTED) {
    rc = run_norm(c, z, k, c->verify_bound, accept, B, 1, 0, false);
    if (rc != RZK_OK) return rc;
    rc = run_norm(c, zp, k, c->verify_bound, accept, B, 1, 0, false);
    if (rc != RZK_OK) return rc;
    rc = run_program(c, PG_LIN_V1, dkv(c), v1, accept, 1, B, 0, false);
  }
  if (rc != RZK_OK) return rc;
  return run_program(c, PG_LIN_V2, dkv(c), {{w1, l, 0}, {w2, l, 0}, {g, 1, 0}, {d, 1, 0}, {zp, k, 0}, {u, l, 0}}, accept, 1, B, 0,
                     false);
}

// =================================================================================================
// SumProof
// =================================================================================================
namespace {
// sum_i g_i (.) (a2.v_i) - a2.v'  (sum.rs:154-160, 301-308) evaluated as a2.(sum_i g_i (.) v_i - v'): V matrix-vector
// products (each with its inverse transforms) become one, for (columns of a2) instead of l rows of V vector x vector
// terms.  Chosen by transform count (forward + inverse, per proof): pays when V is large and a2 has few columns beyond
// its identity block — the reference's key shape; a dense a2 with k > l columns keeps the row-wise form.
bool sum_uses_d(const rzk_ctx* c, uint32_t V) {
  if (c->small) return false;
  if (c->sum_d >= 0) return c->sum_d != 0;
  const uint32_t n = c->n, k = c->k, l = c->l;
  uint32_t cols = 0, fwd = 0, key_terms = 0;   // columns a2 uses; of those with general entries; general entries in all
  for (uint32_t col = 0; col < k; ++col) {
    bool used = false, general = false;
    for (uint32_t j = 0; j < l; ++j) {
      const uint8_t kc = c->key_class[(n + j) * k + col];
      used = used || kc != KC_ZERO;
      general = general || kc == KC_GENERAL;
      key_terms += kc == KC_GENERAL;
    }
    cols += used;
    fwd += general;
  }
  const uint64_t matvec2 = 2ull * (fwd + l), matvec3 = 3ull * (fwd + l);           // a2.v at two / three primes
  const uint64_t rowwise = (uint64_t)V * matvec2 + (uint64_t)l * (6ull * V + 3) + 3ull * key_terms;
  const uint64_t by_d = (uint64_t)cols * (6ull * V + 3) + matvec3;
  return by_d < rowwise && (uint64_t)cols * V <= (uint64_t)kMaxTerms && cols <= (uint32_t)kMaxRows;
}
}  // namespace

int rzk_sum_commit_batch_dev(rzk_ctx* c, uint32_t V, const int64_t* gs, const int64_t* xs, const int64_t* rs,
                             const int64_t* rp, const int64_t* ys, const int64_t* yp, int64_t* cs, int64_t* cpm,
                             int64_t* ts, int64_t* tp, int64_t* u, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || V == 0 || !gs || !xs || !rs || !rp || !ys || !yp || !cs || !cpm || !ts || !tp || !u) return RZK_E_ARG;
  const uint32_t n = c->n, k = c->k, l = c->l;
  const size_t wpolys = (size_t)B * V * l > (size_t)B * k ? (size_t)B * V * l : (size_t)B * k;   // w[B*V][l], or D[B][k]
  int rc = arena_reserve(c, c->ws, polys(c, B * l + wpolys));
  if (rc != RZK_OK) return rc;
  int64_t* xp = (int64_t*)c->ws.p;
  int64_t* w = xp + B * l * c->N;
  // sum.rs:107-115: xp = sum_i x_i (.) g_i
  // (XP runs before ok is preset: a non-canonical x_i / g_i is reported through the sticky word, and again by the
  //  later rows that load the same polynomials with the flags in place)
  // the V multipliers g_i of every proof, transformed once for all the rows that multiply by them (XP, D / U)
  rc = prepare_dkey(c, gs, B, V, 2 * l, nullptr, true);   // l rows of XP, l or (columns of a2) rows of U / D
  if (rc != RZK_OK) return rc;
  struct DkeyScope {   // the images belong to this call
    rzk_ctx* c;
    ~DkeyScope() { c->dkey_img = nullptr; c->dkey_l2 = nullptr; c->dkey_n = 0; }
  } dkey_scope{c};
  rc = run_program(c, PG_SUM_XP, V | dkv(c), {{xs, V * l, 0}, {gs, V, 0}, {xp, l, 0}}, nullptr, 1, B);
  if (rc != RZK_OK) return rc;
  // the a1.y_i key products below leave the transforms of y_i's a2-columns for the D rows (operand 2 of the summand launch)
  rc = prepare_oimg(c, B, V, 2 * l);
  if (rc != RZK_OK) return rc;
  struct OimgScope {
    rzk_ctx* c;
    ~OimgScope() { c->oimg_state = Operands{}; c->oimg_producer_op = 0xffu; }
  } oimg_scope{c};
  // sum.rs:116 and 151: cp = commit(xp; rp), tp = a1.yp
  const std::vector<OpSpec> cp_specs = {{xp, l, 0}, {rp, k, 0}, {yp, k, 0}, {cpm, n + l, 0}, {tp, n, 0}};
  const std::vector<OpSpec> cs_specs = {{xs, l, 0}, {rs, k, 0}, {ys, k, 0}, {cs, n + l, 0}, {ts, n, 0}};
  // ok[b] = constraint(rp_b) && all_i constraint(r_{b,i}): fused into the commit rows when possible
  const bool fused = ok && can_fuse(c, PG_OPEN_COMMIT, 0, c->commit_bound);
  if (fused) {
    rc = run_program_checked(c, PG_OPEN_COMMIT, 0, cp_specs, ok, 1, B, B, c->commit_bound, true);
    if (rc != RZK_OK) return rc;
    // sum.rs:117-120 and 145-148: c_i = commit(x_i; r_i), t_i = a1.y_i — the V summands are extra batch entries
    c->oimg_producer_op = 2;
    rc = run_program_checked(c, PG_OPEN_COMMIT, 0, cs_specs, ok, V, B * V, B, c->commit_bound, false);
    c->oimg_producer_op = 0xffu;
    if (rc != RZK_OK) return rc;
  } else {
    if (ok) {
      rc = run_norm(c, rp, k, c->commit_bound, ok, B, 0, 0);
      if (rc != RZK_OK) return rc;
      rc = run_norm(c, rs, V * k, c->commit_bound, ok, B, 1, 0);
      if (rc != RZK_OK) return rc;
    }
    rc = run_program(c, PG_OPEN_COMMIT, 0, cp_specs, ok, 1, B);
    if (rc != RZK_OK) return rc;
    c->oimg_producer_op = 2;
    rc = run_program(c, PG_OPEN_COMMIT, 0, cs_specs, ok, V, B * V);
    c->oimg_producer_op = 0xffu;
    if (rc != RZK_OK) return rc;
  }
  // sum.rs:154-160: u = sum_i (a2.y_i)(.)g_i - a2.yp
  if (sum_uses_d(c, V)) {   // = a2.(sum_i g_i(.)y_i - yp): D (k polynomials per proof, the columns a2 uses) lives where w would
    int64_t* D = w;
    rc = run_program(c, PG_SUM_D, V | dkv(c) | oiv(c), {{ys, V * k, 0}, {gs, V, 0}, {yp, k, 0}, {D, k, 0}}, ok, 1, B);
    if (rc != RZK_OK) return rc;
    return run_program(c, PG_MATVEC, RZK_KEY_A2 * 2, {{D, k, 0}, {nullptr, l, 0}, {u, l, 0}}, ok, 1, B);
  }
  rc = run_program(c, PG_MATVEC, RZK_KEY_A2 * 2, {{ys, k, 0}, {nullptr, l, 0}, {w, l, 0}}, nullptr, 1, B * V);
  if (rc != RZK_OK) return rc;
  return run_program(c, PG_SUM_U, V | dkv(c), {{w, V * l, 0}, {gs, V, 0}, {yp, k, 0}, {u, l, 0}}, ok, 1, B);
}

int rzk_sum_response_batch_dev(rzk_ctx* c, uint32_t V, const int64_t* ys, const int64_t* yp, const int64_t* rs,
                               const int64_t* rp, const int64_t* d, int64_t* zs, int64_t* zp, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || V == 0 || !ys || !yp || !rs || !rp || !d || !zs || !zp) return RZK_E_ARG;
  const uint32_t k = c->k;
  // sum.rs:188-193: z_i = y_i + r_i (.) d — summands as batch entries, d shared by the V entries of a proof
  int rc = run_program(c, PG_RESPONSE, 1, {{d, 1, 1}, {ys, k, 0}, {rs, k, 0}, {zs, k, 0}}, nullptr, V, B * V);
  if (rc != RZK_OK) return rc;
  // sum.rs:195-197
  return run_program(c, PG_RESPONSE, 1, {{d, 1, 0}, {yp, k, 0}, {rp, k, 0}, {zp, k, 0}}, nullptr, 1, B);
}

int rzk_sum_verify_batch_dev(rzk_ctx* c, uint32_t V, const int64_t* zs, const int64_t* zp, const int64_t* cs,
                             const int64_t* cpm, const int64_t* gs, const int64_t* ts, const int64_t* tp,
                             const int64_t* u, const int64_t* d, uint8_t* accept, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || V == 0 || !zs || !zp || !cs || !cpm || !gs || !ts || !tp || !u || !d || !accept) return RZK_E_ARG;
  if (c->n != c->l) return fail(c, RZK_E_ARG, "c1_c2 split needs n == l (reference panics in Mat::add)");
  const uint32_t n = c->n, k = c->k, l = c->l;
  const bool by_d = sum_uses_d(c, V);
  const size_t w1polys = by_d && (size_t)B * k > (size_t)B * V * l ? (size_t)B * k : (size_t)B * V * l;   // w1[B*V][l], or D[B][k]
  int rc = arena_reserve(c, c->ws, polys(c, w1polys + B * l + B * V * n));
  if (rc != RZK_OK) return rc;
  int64_t* w1 = (int64_t*)c->ws.p;
  int64_t* w2 = w1 + w1polys * c->N;
  int64_t* w0 = w2 + B * l * c->N;   // a1.z of the relation checks (n >= 2 only)
  const std::vector<OpSpec> rel_s = {{zs, k, 0}, {ts, n, 0}, {cs, n + l, 0}, {d, 1, 1}};
  const std::vector<OpSpec> rel_p = {{zp, k, 0}, {tp, n, 0}, {cpm, n + l, 0}, {d, 1, 0}};
  // sum.rs:262-271 norm predicates ride on sum.rs:278-291 (every summand; batch entries B*V, flag per proof)
  // and sum.rs:294-298
  // (the a1.z_i key products leave the transforms of z_i's a2-columns for the D rows: operand 0 of their launch)
  rc = prepare_oimg(c, B, V, 2 * l);
  if (rc != RZK_OK) return rc;
  struct OimgScope {
    rzk_ctx* c;
    ~OimgScope() { c->oimg_state = Operands{}; c->oimg_producer_op = 0xffu; }
  } oimg_scope{c};
  c->oimg_producer_op = 0;
  rc = run_a1_relation(c, rel_s, w0, accept, V, B * V, B, true);
  c->oimg_producer_op = 0xffu;
  if (rc != RZK_OK) return rc;
  rc = run_a1_relation(c, rel_p, w0, accept, 1, B, B, false);
  if (rc != RZK_OK) return rc;
  // sum.rs:301-319.  The multipliers' images are prepared AFTER the relation rows above have initialised accept: a
  // non-canonical g_i must clear the proof's verdict, and nothing presets the flags again from here on.
  rc = prepare_dkey(c, gs, B, V, 2 * l, accept, false);
  if (rc != RZK_OK) return rc;
  struct DkeyScope {
    rzk_ctx* c;
    ~DkeyScope() { c->dkey_img = nullptr; c->dkey_l2 = nullptr; c->dkey_n = 0; }
  } dkey_scope{c};
  if (by_d) {   // lhs = a2.(sum_i g_i(.)z_i - zp): D in w1's place, then one relation row per row of a2
    int64_t* D = w1;
    rc = run_program(c, PG_SUM_D, V | dkv(c) | oiv(c), {{zs, V * k, 0}, {gs, V, 0}, {zp, k, 0}, {D, k, 0}}, accept, 1, B, 0, false);
    if (rc != RZK_OK) return rc;
    rc = run_program(c, PG_SUM_W2, V | dkv(c), {{cs, V * (n + l), 0}, {gs, V, 0}, {cpm, n + l, 0}, {w2, l, 0}}, accept, 1, B, 0, false);
    if (rc != RZK_OK) return rc;
    return run_program(c, PG_SUM_V4, 0, {{D, k, 0}, {w2, l, 0}, {d, 1, 0}, {u, l, 0}}, accept, 1, B, 0, false);
  }
  rc = run_program(c, PG_MATVEC, RZK_KEY_A2 * 2, {{zs, k, 0}, {nullptr, l, 0}, {w1, l, 0}}, accept, V, B * V, 0, false);
  if (rc != RZK_OK) return rc;
  rc = run_program(c, PG_SUM_W2, V | dkv(c), {{cs, V * (n + l), 0}, {gs, V, 0}, {cpm, n + l, 0}, {w2, l, 0}}, accept, 1, B, 0, false);
  if (rc != RZK_OK) return rc;
  return run_program(c, PG_SUM_V3, V | dkv(c), {{w1, V * l, 0}, {gs, V, 0}, {zp, k, 0}, {w2, l, 0}, {d, 1, 0}, {u, l, 0}}, accept,
                     1, B, 0, false);
}

// =================================================================================================
// host-pointer variants
// =================================================================================================
#define IN(ptr, bytes) HostBuf{(ptr), nullptr, (bytes), nullptr}
#define OUT(ptr, bytes) HostBuf{nullptr, (ptr), (bytes), nullptr}
#define DEV(i, T) ((T)bufs[i].dev)
// A host-pointer call reports exactly its own input faults: the sticky word is cleared on the stream before the call's
// work (a fault left by earlier *_dev calls whose caller never asked — rzk_ctx_check_inputs — is dropped here, as
// include/rzk.h documents) and consumed again when the call fails half-way, so it never leaks into the next call.
#define HOST_WRAP(call)                                                                    \
  do {                                                                                     \
    HIPCHK(c, hipMemsetAsync(c->d_bad, 0, sizeof(uint32_t), c->stream));                   \
    int rc_ = stage_in(c, bufs);                                                           \
    if (rc_ != RZK_OK) return rc_;                                                         \
    rc_ = (call);                                                                          \
    if (rc_ != RZK_OK) {                                                                   \
      const std::string keep_ = c->err;                                                    \
      (void)take_input_error(c);                                                           \
      c->err = keep_;                                                                      \
      return rc_;                                                                          \
    }                                                                                      \
    return stage_out(c, bufs);                                                             \
  } while (0)

int rzk_polymul_batch(rzk_ctx* c, const int64_t* a, const int64_t* b, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !out) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(a, polys(c, count)), IN(b, polys(c, count)), OUT(out, polys(c, count))};
  HOST_WRAP(rzk_polymul_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, int64_t*), count));
}

int rzk_matvec_batch(rzk_ctx* c, int which, const int64_t* v, const int64_t* addend, int64_t* out, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !v || !out || which < 0 || which > 2) return RZK_E_ARG;
  const uint32_t rows = which == RZK_KEY_A1 ? c->n : (which == RZK_KEY_A2 ? c->l : c->n + c->l);
  std::vector<HostBuf> bufs = {IN(v, polys(c, B * c->k)), IN(addend, addend ? polys(c, B * rows) : 0),
                               OUT(out, polys(c, B * rows))};
  HOST_WRAP(rzk_matvec_batch_dev(c, which, DEV(0, const int64_t*), addend ? DEV(1, const int64_t*) : nullptr,
                                 DEV(2, int64_t*), B));
}

int rzk_cmul_batch(rzk_ctx* c, const int64_t* m, uint32_t rows, const int64_t* p, int64_t* out, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !m || !p || !out || rows == 0) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(m, polys(c, B * rows)), IN(p, polys(c, B)), OUT(out, polys(c, B * rows))};
  HOST_WRAP(rzk_cmul_batch_dev(c, DEV(0, const int64_t*), rows, DEV(1, const int64_t*), DEV(2, int64_t*), B));
}

int rzk_canonicalize_batch(rzk_ctx* c, const int64_t* in, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;
  if (!c || !in || !out) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(in, polys(c, count)), OUT(out, polys(c, count))};
  HOST_WRAP(rzk_canonicalize_batch_dev(c, DEV(0, const int64_t*), DEV(1, int64_t*), count));
}

int rzk_add_batch(rzk_ctx* c, const int64_t* a, const int64_t* b, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !out) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(a, polys(c, count)), IN(b, polys(c, count)), OUT(out, polys(c, count))};
  HOST_WRAP(rzk_add_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, int64_t*), count));
}

int rzk_sub_batch(rzk_ctx* c, const int64_t* a, const int64_t* b, int64_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !out) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(a, polys(c, count)), IN(b, polys(c, count)), OUT(out, polys(c, count))};
  HOST_WRAP(rzk_sub_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, int64_t*), count));
}

int rzk_norm2_le_batch(rzk_ctx* c, const int64_t* v, uint32_t rows, uint64_t bound, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !v || !ok || rows == 0) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(v, polys(c, B * rows)), OUT(ok, B)};
  HOST_WRAP(rzk_norm2_le_batch_dev(c, DEV(0, const int64_t*), rows, bound, DEV(1, uint8_t*), B));
}

int rzk_eq_batch(rzk_ctx* c, const int64_t* a, const int64_t* b, uint32_t rows, uint8_t* eq, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !a || !b || !eq || rows == 0) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(a, polys(c, B * rows)), IN(b, polys(c, B * rows)), OUT(eq, B)};
  HOST_WRAP(rzk_eq_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), rows, DEV(2, uint8_t*), B));
}

int rzk_ntt_forward_batch(rzk_ctx* c, int prime, const uint32_t* in, uint32_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !in || !out) return RZK_E_ARG;
  const size_t bytes = count * c->N * sizeof(uint32_t);
  std::vector<HostBuf> bufs = {IN(in, bytes), OUT(out, bytes)};
  HOST_WRAP(rzk_ntt_forward_batch_dev(c, prime, DEV(0, const uint32_t*), DEV(1, uint32_t*), count));
}

int rzk_ntt_inverse_batch(rzk_ctx* c, int prime, const uint32_t* in, uint32_t* out, size_t count) {
  if (c && count == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !in || !out) return RZK_E_ARG;
  const size_t bytes = count * c->N * sizeof(uint32_t);
  std::vector<HostBuf> bufs = {IN(in, bytes), OUT(out, bytes)};
  HOST_WRAP(rzk_ntt_inverse_batch_dev(c, prime, DEV(0, const uint32_t*), DEV(1, uint32_t*), count));
}

int rzk_commit_batch(rzk_ctx* c, const int64_t* x, const int64_t* r, int64_t* cm, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !x || !r || !cm) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(x, polys(c, B * c->l)), IN(r, polys(c, B * c->k)),
                               OUT(cm, polys(c, B * (c->n + c->l))), OUT(ok, ok ? B : 0)};
  HOST_WRAP(rzk_commit_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, int64_t*),
                                 ok ? DEV(3, uint8_t*) : nullptr, B));
}

int rzk_commitment_verify_batch(rzk_ctx* c, const int64_t* cm, const int64_t* x, const int64_t* r, const int64_t* f,
                                uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !cm || !x || !r || !ok) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(cm, polys(c, B * (c->n + c->l))), IN(x, polys(c, B * c->l)),
                               IN(r, polys(c, B * c->k)), IN(f, f ? polys(c, B) : 0), OUT(ok, B)};
  HOST_WRAP(rzk_commitment_verify_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                            f ? DEV(3, const int64_t*) : nullptr, DEV(4, uint8_t*), B));
}

int rzk_open_commit_batch(rzk_ctx* c, const int64_t* x, const int64_t* r, const int64_t* y, int64_t* cm, int64_t* t,
                          uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !x || !r || !y || !cm || !t) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(x, polys(c, B * c->l)), IN(r, polys(c, B * c->k)), IN(y, polys(c, B * c->k)),
                               OUT(cm, polys(c, B * (c->n + c->l))), OUT(t, polys(c, B * c->n)), OUT(ok, ok ? B : 0)};
  HOST_WRAP(rzk_open_commit_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                      DEV(3, int64_t*), DEV(4, int64_t*), ok ? DEV(5, uint8_t*) : nullptr, B));
}

int rzk_open_response_batch(rzk_ctx* c, const int64_t* y, const int64_t* r, const int64_t* d, int64_t* z, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !y || !r || !d || !z) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(y, polys(c, B * c->k)), IN(r, polys(c, B * c->k)), IN(d, polys(c, B)),
                               OUT(z, polys(c, B * c->k))};
  HOST_WRAP(rzk_open_response_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                        DEV(3, int64_t*), B));
}

int rzk_open_verify_batch(rzk_ctx* c, const int64_t* z, const int64_t* t, const int64_t* cm, const int64_t* d,
                          uint8_t* accept, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !z || !t || !cm || !d || !accept) return RZK_E_ARG;
  std::vector<HostBuf> bufs = {IN(z, polys(c, B * c->k)), IN(t, polys(c, B * c->n)),
                               IN(cm, polys(c, B * (c->n + c->l))), IN(d, polys(c, B)), OUT(accept, B)};
  HOST_WRAP(rzk_open_verify_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                      DEV(3, const int64_t*), DEV(4, uint8_t*), B));
}

int rzk_linear_commit_batch(rzk_ctx* c, const int64_t* g, const int64_t* x, const int64_t* r, const int64_t* rp,
                            const int64_t* y, const int64_t* yp, int64_t* cm, int64_t* cpm, int64_t* t, int64_t* tp,
                            int64_t* u, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !g || !x || !r || !rp || !y || !yp || !cm || !cpm || !t || !tp || !u) return RZK_E_ARG;
  const size_t k = c->k, n = c->n, l = c->l;
  std::vector<HostBuf> bufs = {IN(g, polys(c, B)),          IN(x, polys(c, B * l)),        IN(r, polys(c, B * k)),
                               IN(rp, polys(c, B * k)),     IN(y, polys(c, B * k)),        IN(yp, polys(c, B * k)),
                               OUT(cm, polys(c, B * (n + l))), OUT(cpm, polys(c, B * (n + l))), OUT(t, polys(c, B * n)),
                               OUT(tp, polys(c, B * n)),    OUT(u, polys(c, B * l)),       OUT(ok, ok ? B : 0)};
  HOST_WRAP(rzk_linear_commit_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                        DEV(3, const int64_t*), DEV(4, const int64_t*), DEV(5, const int64_t*),
                                        DEV(6, int64_t*), DEV(7, int64_t*), DEV(8, int64_t*), DEV(9, int64_t*),
                                        DEV(10, int64_t*), ok ? DEV(11, uint8_t*) : nullptr, B));
}

int rzk_linear_response_batch(rzk_ctx* c, const int64_t* y, const int64_t* yp, const int64_t* r, const int64_t* rp,
                              const int64_t* d, int64_t* z, int64_t* zp, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !y || !yp || !r || !rp || !d || !z || !zp) return RZK_E_ARG;
  const size_t kb = polys(c, B * c->k);
  std::vector<HostBuf> bufs = {IN(y, kb), IN(yp, kb), IN(r, kb), IN(rp, kb), IN(d, polys(c, B)), OUT(z, kb), OUT(zp, kb)};
  HOST_WRAP(rzk_linear_response_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                          DEV(3, const int64_t*), DEV(4, const int64_t*), DEV(5, int64_t*),
                                          DEV(6, int64_t*), B));
}

int rzk_linear_verify_batch(rzk_ctx* c, const int64_t* z, const int64_t* zp, const int64_t* cm, const int64_t* cpm,
                            const int64_t* g, const int64_t* t, const int64_t* tp, const int64_t* u, const int64_t* d,
                            uint8_t* accept, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || !z || !zp || !cm || !cpm || !g || !t || !tp || !u || !d || !accept) return RZK_E_ARG;
  const size_t k = c->k, n = c->n, l = c->l;
  std::vector<HostBuf> bufs = {IN(z, polys(c, B * k)),   IN(zp, polys(c, B * k)), IN(cm, polys(c, B * (n + l))),
                               IN(cpm, polys(c, B * (n + l))), IN(g, polys(c, B)), IN(t, polys(c, B * n)),
                               IN(tp, polys(c, B * n)),  IN(u, polys(c, B * l)),  IN(d, polys(c, B)),
                               OUT(accept, B)};
  HOST_WRAP(rzk_linear_verify_batch_dev(c, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                        DEV(3, const int64_t*), DEV(4, const int64_t*), DEV(5, const int64_t*),
                                        DEV(6, const int64_t*), DEV(7, const int64_t*), DEV(8, const int64_t*),
                                        DEV(9, uint8_t*), B));
}

int rzk_sum_commit_batch(rzk_ctx* c, uint32_t V, const int64_t* gs, const int64_t* xs, const int64_t* rs,
                         const int64_t* rp, const int64_t* ys, const int64_t* yp, int64_t* cs, int64_t* cpm,
                         int64_t* ts, int64_t* tp, int64_t* u, uint8_t* ok, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || V == 0 || !gs || !xs || !rs || !rp || !ys || !yp || !cs || !cpm || !ts || !tp || !u) return RZK_E_ARG;
  const size_t k = c->k, n = c->n, l = c->l;
  std::vector<HostBuf> bufs = {IN(gs, polys(c, B * V)),        IN(xs, polys(c, B * V * l)), IN(rs, polys(c, B * V * k)),
                               IN(rp, polys(c, B * k)),        IN(ys, polys(c, B * V * k)), IN(yp, polys(c, B * k)),
                               OUT(cs, polys(c, B * V * (n + l))), OUT(cpm, polys(c, B * (n + l))),
                               OUT(ts, polys(c, B * V * n)),   OUT(tp, polys(c, B * n)),    OUT(u, polys(c, B * l)),
                               OUT(ok, ok ? B : 0)};
  HOST_WRAP(rzk_sum_commit_batch_dev(c, V, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                     DEV(3, const int64_t*), DEV(4, const int64_t*), DEV(5, const int64_t*),
                                     DEV(6, int64_t*), DEV(7, int64_t*), DEV(8, int64_t*), DEV(9, int64_t*),
                                     DEV(10, int64_t*), ok ? DEV(11, uint8_t*) : nullptr, B));
}

int rzk_sum_response_batch(rzk_ctx* c, uint32_t V, const int64_t* ys, const int64_t* yp, const int64_t* rs,
                           const int64_t* rp, const int64_t* d, int64_t* zs, int64_t* zp, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || V == 0 || !ys || !yp || !rs || !rp || !d || !zs || !zp) return RZK_E_ARG;
  const size_t k = c->k;
  std::vector<HostBuf> bufs = {IN(ys, polys(c, B * V * k)), IN(yp, polys(c, B * k)), IN(rs, polys(c, B * V * k)),
                               IN(rp, polys(c, B * k)),     IN(d, polys(c, B)),      OUT(zs, polys(c, B * V * k)),
                               OUT(zp, polys(c, B * k))};
  HOST_WRAP(rzk_sum_response_batch_dev(c, V, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                       DEV(3, const int64_t*), DEV(4, const int64_t*), DEV(5, int64_t*),
                                       DEV(6, int64_t*), B));
}

int rzk_sum_verify_batch(rzk_ctx* c, uint32_t V, const int64_t* zs, const int64_t* zp, const int64_t* cs,
                         const int64_t* cpm, const int64_t* gs, const int64_t* ts, const int64_t* tp, const int64_t* u,
                         const int64_t* d, uint8_t* accept, size_t B) {
  if (c && B == 0) return RZK_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!c || V == 0 || !zs || !zp || !cs || !cpm || !gs || !ts || !tp || !u || !d || !accept) return RZK_E_ARG;
  const size_t k = c->k, n = c->n, l = c->l;
  std::vector<HostBuf> bufs = {IN(zs, polys(c, B * V * k)),       IN(zp, polys(c, B * k)),
                               IN(cs, polys(c, B * V * (n + l))), IN(cpm, polys(c, B * (n + l))),
                               IN(gs, polys(c, B * V)),           IN(ts, polys(c, B * V * n)),
                               IN(tp, polys(c, B * n)),           IN(u, polys(c, B * l)),
                               IN(d, polys(c, B)),                OUT(accept, B)};
  HOST_WRAP(rzk_sum_verify_batch_dev(c, V, DEV(0, const int64_t*), DEV(1, const int64_t*), DEV(2, const int64_t*),
                                     DEV(3, const int64_t*), DEV(4, const int64_t*), DEV(5, const int64_t*),
                                     DEV(6, const int64_t*), DEV(7, const int64_t*), DEV(8, const int64_t*),
                                     DEV(9, uint8_t*), B));
}

// =================================================================================================
// instrumentation
// =================================================================================================
double rzk_bench_ntt_forward_dev(rzk_ctx* c, int prime, const uint32_t* in, uint32_t* out, size_t count, int iters) {
  if (!c || !in || !out || iters <= 0 || prime < 0 || prime >= kMaxPrimes) return (double)RZK_E_ARG;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (double)RZK_E_HIP;
  // warm-up launch, then `iters` back-to-back launches between two events on the launch stream
  int rc = rzk_ntt_forward_batch_dev(c, prime, in, out, count);
  if (rc == RZK_OK && hipEventRecord(e0, c->stream) != hipSuccess) rc = RZK_E_HIP;
  for (int i = 0; i < iters && rc == RZK_OK; ++i) rc = rzk_ntt_forward_batch_dev(c, prime, in, out, count);
  if (rc == RZK_OK && hipEventRecord(e1, c->stream) != hipSuccess) rc = RZK_E_HIP;
  if (rc == RZK_OK && hipEventSynchronize(e1) != hipSuccess) rc = RZK_E_HIP;
  float ms = 0.f;
  if (rc == RZK_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = RZK_E_HIP;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != RZK_OK) return (double)rc;
  return (double)ms * 1000.0 / iters;
}

// Diagnostic: copy of the row-kernel scratch (per-wave lines; builds with -DRZK_STAMPS=1 leave wave time stamps there).
int rzk_debug_read_scratch(rzk_ctx* c, void* dst, size_t bytes, size_t* total) {
  if (!c) return RZK_E_ARG;
  const size_t all = c->d_row_scratch ? row_scratch_words((int)c->logn, c->num_cus) * sizeof(uint32_t) : 0;
  if (total) *total = all;
  if (!dst || !bytes) return RZK_OK;
  if (bytes > all) bytes = all;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(dst, c->d_row_scratch, bytes, hipMemcpyDeviceToHost));
  return RZK_OK;
}

int rzk_prof_enable(rzk_ctx* c, int on) {
  if (!c) return RZK_E_ARG;
  c->prof = on != 0;
  return RZK_OK;
}

int rzk_prof_reset(rzk_ctx* c) {
  if (!c) return RZK_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->prof_used = 0;
  c->prof_us = 0.0;
  c->prof_launches = 0;
  return RZK_OK;
}

uint64_t rzk_prof_count(const rzk_ctx* c) { return c ? (uint64_t)c->prof_used : 0; }

int rzk_prof_read_all(rzk_ctx* c, double* us, size_t cap, size_t* count) {
  if (!c || (!us && cap)) return RZK_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (count) *count = c->prof_used;
  for (size_t i = 0; i < c->prof_used && i < cap; ++i) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
    us[i] = (double)ms * 1000.0;
  }
  return RZK_OK;
}

// One line per recorded launch, in launch order: "<kernel>\t<algorithmic bytes>\n".  Returns the text's length through
// *needed (without the terminating NUL); copies at most cap - 1 characters.
int rzk_prof_read_kernels(rzk_ctx* c, char* buf, size_t cap, size_t* needed) {
  if (!c || (!buf && cap)) return RZK_E_ARG;
  std::string text;
  for (size_t i = 0; i < c->prof_used && i < c->prof_info.size(); ++i)
    text += c->prof_info[i].kernel + "\t" + std::to_string(c->prof_info[i].bytes) + "\n";
  if (needed) *needed = text.size();
  if (cap) {
    const size_t ncopy = text.size() < cap - 1 ? text.size() : cap - 1;
    std::memcpy(buf, text.data(), ncopy);
    buf[ncopy] = 0;
  }
  return RZK_OK;
}

int rzk_prof_read(rzk_ctx* c, double* row_kernel_us, uint64_t* row_kernel_launches) {
  if (!c) return RZK_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->prof_used; ++i) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
    c->prof_us += (double)ms * 1000.0;
    c->prof_launches++;
  }
  c->prof_used = 0;
  if (row_kernel_us) *row_kernel_us = c->prof_us;
  if (row_kernel_launches) *row_kernel_launches = c->prof_launches;
  return RZK_OK;
}

}  // extern "C"
