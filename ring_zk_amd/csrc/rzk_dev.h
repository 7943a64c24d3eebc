// rzk_dev.h — structures shared by the kernels (rzk_kernels.hip) and the C-ABI host code (rzk_api.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "rzk_core.h"

namespace rzk {

// Constants every kernel reads through a pointer to device memory (uniform -> scalar loads).
// The twiddle tables are passed separately as one global pointer: table (2*prime + dir) of kTableLen
// words, dir 0 = psi^{bitrev} * R (forward), dir 1 = psi^{-bitrev} * R (inverse).
struct DevTables {
  PrimeConsts pc[kMaxPrimes];
  CrtConsts crt;
  double cap[kMaxPrimes + 1];           // cap[np] = largest |exact result| np primes can represent
};

// ---- row programs ------------------------------------------------------------------------------------
// Every protocol phase is a small "row program": for each proof b of the batch and each output row,
//
//     out_row = sum_terms sign * (KEY[entry]  (*)  V[b][src])        key entry times a per-proof polynomial
//             + sum_terms sign * (U[b][srcA]  (*)  V[b][srcB])        product of two per-proof polynomials
//             + sum_adds  sign *  T[b][src]                           plain additions
//
// evaluated by ONE wavefront: the products are accumulated in the NTT domain of as many auxiliary
// primes as the exact integer result needs (decided per row from the operands' norms), transformed
// back once, CRT-reconstructed, reduced to the centred representative mod q, then the additions are
// applied.  The result is either stored (MODE_STORE) or tested against zero (MODE_ZERO, the
// `lhs == rhs` of the verifiers, e.g. src/prove/open.rs:173).
// Program tables (rows, terms, additions, units, items, plans) are written by the host before a launch and never by
// a kernel, so the kernels may read them through the constant address space: the compiler then uses scalar loads
// (one s_load per record, no vector-memory latency in the control flow).  The records are 4 / 8 / 16 bytes, naturally
// aligned inside their tables (static_asserts in rzk_api.cpp).
#if defined(__HIPCC__)
template <class Tp>
__device__ __forceinline__ Tp table_load(const Tp* p) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RZK_NO_TABLE_SLOAD)
  typedef const Tp __attribute__((address_space(4))) * cptr_t;
  return *reinterpret_cast<cptr_t>(reinterpret_cast<uintptr_t>(p));
#else
  return *p;
#endif
}
#endif
constexpr int kMaxOperands = 12;
constexpr int kMaxRows = 48;
constexpr int kMaxTerms = 640;
constexpr int kMaxAdds = 128;

// TERM_SHIFT: (a_op,a_off) (*) (b_op,b_off) with a sparse `a` (the challenge d), evaluated as signed negacyclic
// rotations (ShiftGeo, rzk_core.h) instead of transforms.  A row keeps its shift terms behind its transform
// terms: terms[term0 .. term0+nterms) are KEY / VEC, terms[term0+nterms .. +nshift) are SHIFT.
enum : uint8_t { TERM_KEY = 0, TERM_VEC = 1, TERM_SHIFT = 2, TERM_DKEY = 3, TERM_DD = 4, TERM_KIND_MASK = 0x3f };
enum : uint8_t { MODE_STORE = 0, MODE_ZERO = 1 };
// Fused norm predicate (Params::check_*_constraint, src/params.rs:102-118): a term (its b operand) or an
// addition marked with CHECK also tests sum c^2 < Operands::norm_limit for the polynomial it loads and
// clears flags[proof] on failure.  The host marks, for each polynomial of the checked vector, the first
// load in program order, so every polynomial is tested exactly once and no separate norm pass is needed.
// CHECK2 marks a second checked vector whose failure clears only bit 1 of the flag byte (two-bit verdicts:
// Linear commit reports constraint(r) in bit 0 and constraint(r') in bit 1); it is honoured by row_kernel only.
constexpr uint8_t TERM_CHECK = 0x80;   // in Term::kind
constexpr uint8_t TERM_CHECK2 = 0x40;
constexpr uint8_t ADD_CHECK = 0x80;    // in AddTerm::op
constexpr uint8_t ADD_CHECK2 = 0x40;
constexpr uint8_t ADD_OP_MASK = 0x3f;

struct alignas(8) Term {
  uint8_t kind;     // TERM_KEY: KEY[a_off] (*) operand(b_op, b_off);  TERM_VEC: (a_op,a_off) (*) (b_op,b_off);
                    // TERM_DKEY: image a_off of the batch entry's own multipliers (Operands::dkey_img) (*) operand(b_op, b_off)
                    // TERM_DD: the same, and the operand's transform is taken from Operands::oimg when an earlier launch
                    //          of the call left it there (b_off = summand * oimg_k + column); transformed here otherwise
  int8_t sign;      // +1 / -1
  uint8_t a_op, b_op;
  uint16_t a_off, b_off;
};
struct alignas(4) AddTerm {
  uint8_t op;
  int8_t sign;
  uint16_t off;
};
struct alignas(8) Row {
  uint16_t term0, nterms, add0, nadds;
  uint8_t out_op, mode;
  uint16_t out_off;
  uint16_t nshift, pad;
};
// Row groups: consecutive rows made of key products over the SAME operand list (the n rows of a1 over
// columns n..k-1, the l rows of a2, ...).  One wavefront evaluates a whole group: every shared operand is
// transformed once and multiplied into up to kGroupMax accumulators (row_group_kernel).
constexpr int kGroupMax = 4;
struct GroupDesc {
  uint16_t row0, count;
};
struct Program {
  uint32_t nrows, nterms, nadds, ngroups;
  GroupDesc groups[kMaxRows];
  Row rows[kMaxRows];
  Term terms[kMaxTerms];
  AddTerm adds[kMaxAdds];
};

// ---- wave programs (unit_kernel): the default evaluation of a row program -------------------------------------
// A UNIT is one output row — or a PAIR of rows (A, B) — evaluated by one wavefront over an ordered list of ITEMS;
// an item is one forward-transformed operand together with what is multiplied into the rows with it:
//   ITEM_KEY   acc_A += signA * KEY[keyA] (*) X      and / or      acc_B = signB * KEY[keyB] (*) X
//   ITEM_VEC   acc_A += signA * X(a_op,a_off) (*) X(b_op,b_off)                      (single-row units only)
// Row B of a pair has exactly one product and its operand is the unit's LAST item, so the transform of that operand
// is shared (c0 = r0 + K0 r1 + K1 r2 and c1 = r1 + K2 r2 + x of an Open commitment share r2: commit.rs:109-125).
// While operands are loaded and transformed nothing else is live in registers: the accumulator of row A is parked
// in LDS between items (one N-word buffer per wavefront, the layout of the resident key), the one of row B exists
// only from the last item on.  The Garner state between primes lives in a per-wave global scratch line.
// Rotation terms (TERM_SHIFT), plain additions, the store / zero test and the norm marks stay in Program::rows.
enum : uint8_t { ITEM_KEY = 0, ITEM_VEC = 1 };
constexpr uint16_t kNoKey = 0xffff;
constexpr uint16_t kNoRow = 0xffff;
struct alignas(8) Item {
  uint8_t kind;        // ITEM_KEY / ITEM_VEC
  uint8_t flags;       // TERM_CHECK / TERM_CHECK2: the b operand carries a fused norm mark
  uint8_t b_op, a_op;
  uint16_t b_off, a_off;
  uint16_t keyA, keyB;   // key entries (kNoKey = the item does not feed that row)
  int8_t signA, signB;
  uint16_t pad;
};
struct alignas(8) Unit {
  uint16_t rowA, rowB;      // indices into Program::rows; rowB = kNoRow for a single row
  uint16_t item0, nitems;   // nitems items from item0 on
};
struct WaveProgram {
  uint32_t nunits, nitems;
  Unit units[kMaxRows];
  Item items[kMaxTerms];
};

// Shared-operand path: the distinct polynomials ("slots") the product terms of a program read, and for
// each term the slots of its operands.  check[s] != 0: the slot belongs to a norm-checked vector.
constexpr int kMaxSlots = 512;
struct SlotTable {
  uint32_t nslots, pad;
  uint16_t op[kMaxSlots], off[kMaxSlots];
  uint8_t check[kMaxSlots];
  uint16_t term_a[kMaxTerms], term_b[kMaxTerms];
};

// Row blocks (row_block_kernel): consecutive rows of a key-only program whose distinct operands fit in LDS
// together.  One 8-wave workgroup evaluates a block of one proof prime by prime: the waves transform the
// block's operands into LDS (each exactly once), meet at a barrier, then every wave evaluates rows from the
// staged transforms.  For [a1;a2].r at (8,17,8) that is 9 transforms per prime and block instead of 72 + 8.
constexpr int kBlockWaves = 8;     // (9 measured slightly slower in round 1; 10 — no idle wave in the operand phase — +0.7 % in round 2: noise)
constexpr int kBlockMaxSlots = 9;    // staged operand transforms per block: 9 * 4N bytes of LDS (72 KiB at N = 2048)
constexpr int kBlockMaxRows = 16;    // rows per block (two Garner state lines each in the workgroup's scratch)
struct BlockDesc {
  uint16_t row0, nrows, slot0, nslots;
};
struct BlockPlan {
  uint32_t nblocks, nslots_total;
  BlockDesc blk[kMaxRows];
  uint16_t slot_op[kMaxSlots], slot_off[kMaxSlots];   // indexed by BlockDesc::slot0 + s
  uint8_t slot_check[kMaxSlots];
  uint16_t term_slot[kMaxTerms];                      // slot (within its block) of each term's operand
};

// Operand table of one launch.  The batch index b of a task may be a (proof, summand) pair:
// bo = b / group is the proof.  Polynomial (op, off) lives at
// base[op] + ((outer[op] ? bo : b) * stride[op] + off) * N ; verification flags are per proof (flags[bo]).
struct Operands {
  int64_t* base[kMaxOperands];
  uint32_t stride[kMaxOperands];
  uint32_t outer[kMaxOperands];
  uint32_t group;
  uint32_t pad;          // != 0: two-bit verdict flags (TERM_CHECK clears bit 0, TERM_CHECK2 bit 1; row_kernel only)
  uint64_t norm_limit;   // (bound+1)^2 of the fused norm predicate; must be <= 2^48 (0 = unused)
  // Canonical-input test (every coefficient a kernel loads must be the centred representative a ZqI64 holds,
  // src/params.rs:122-127): a violation clears the proof's verdict flag when the launch has flags, and sets this
  // sticky word of the context when it is not NULL (prover-side and Mat-level entry points: the call fails).
  uint32_t* bad;
  // != 0: the caller vouches that every coefficient this launch loads is canonical (written by this library on this
  // device, or validated before): the canonical test is skipped (rzk_ctx_trust_device_outputs).  Norm predicates and
  // the norm measurements that fix the prime count are evaluated as always.
  uint32_t trusted;
  // != 0: the launch itself initialises every verdict flag to this value before its rows can clear it — only set when
  // one team evaluates ALL rows of a batch entry and flags are per entry (unit kernels with units_per_task = all, group 1);
  // otherwise the host presets the flags with a fill launch of its own (rzk_api.cpp, run_program)
  uint32_t preset;
  // Per-entry multiplier images (TERM_DKEY, row_kernel only): the scalar polynomials g_i of the Linear / Sum proofs, which
  // every vector x vector row of a proof multiplies by, transformed ONCE per proof and call (dkey_transform_kernel) into
  // the form of the resident key: [entry][dkey_n][kKeyImages][N] residues x N^-1 x R, NTT-domain layout; dkey_l2 their
  // 2-norms (upper bounds).  Entry = the proof index (b / group).
  const uint32_t* dkey_img;
  const double* dkey_l2;
  uint32_t dkey_n, pad3;
  // Operand images: the key-product kernels that transform the vectors y_i (z_i) of a Sum call anyway — row_group_kernel,
  // row_block_kernel evaluating a1.y_i (a1.z_i) — leave the transforms of the columns a2 uses here (producer: oimg_op =
  // that operand's index, entry = batch entry), and the vector x vector rows of D = sum_i g_i (.) v_i read them back
  // (consumer, TERM_DD: entry = proof * oimg_group + summand) instead of transforming v_{i,c} again.
  // [entry][oimg_n][kKeyImages][N] lazy residues in the NTT-domain layout; oimg_l2 2-norm bounds; oimg_np primes stored.
  uint32_t* oimg;
  double* oimg_l2;
  uint8_t* oimg_np;
  uint32_t oimg_n, oimg_op, oimg_group, oimg_k;
  int8_t oimg_col[32];   // column of the vector -> image index, or -1
};

// ---- launchers (defined in rzk_kernels.hip) --------------------------------------------------------------------
struct LaunchCfg {
  void* stream;   // hipStream_t
  int num_cus;
  int pair_poly;  // N = 2048: two wavefronts per polynomial (PairTeam) in unit_kernel / row_kernel / row_block_kernel
  int unit_io;    // key-product programs through unit_io_kernel (operands read once) instead of unit_kernel
};

// units_per_task: how many consecutive units of one batch entry a wavefront evaluates back to back (all of them =
// one wavefront per proof: equal-cost tasks, no tail)
// row_kernel: one wavefront per (batch entry, row); programs with vector x vector products
int launch_rows(int logn, const LaunchCfg& cfg, const Program* d_prog, uint32_t nrows, bool has_shift, const Operands& ops,
                const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* d_T, const uint32_t* d_tw,
                uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch, bool has_dd = false);
// work_per_entry: transforms one batch entry costs at two primes (the progress priorities only need an estimate)
int launch_units(int logn, const LaunchCfg& cfg, const Program* d_prog, const WaveProgram* d_wp, uint32_t nunits,
                 uint32_t units_per_task, uint32_t work_per_entry, bool has_vec, bool has_shift, const Operands& ops,
                 const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* d_T, const uint32_t* d_tw,
                 uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch);
int launch_row_program_slots(int logn, const LaunchCfg& cfg, const Program* d_prog, const SlotTable* d_slots,
                             uint32_t nslots, const Operands& ops, const uint32_t* d_key_ntt,
                             const double* d_key_l2, const DevTables* d_T, const uint32_t* d_tw, uint32_t* d_ws,
                             double* d_norms, uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch,
                             uint32_t np_store);
int launch_row_groups(int logn, const LaunchCfg& cfg, const Program* d_prog, uint32_t ngroups, const Operands& ops,
                      const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* d_T, const uint32_t* d_tw,
                      uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch);
size_t group_scratch_words(int logn, int num_cus);
int launch_row_blocks(int logn, const LaunchCfg& cfg, const Program* d_prog, const BlockPlan* d_plan, uint32_t nblocks,
                      const Operands& ops, const uint32_t* d_key_ntt, const double* d_key_l2, const DevTables* d_T,
                      const uint32_t* d_tw, uint32_t* d_scratch, uint8_t* d_flags, uint64_t batch);
size_t block_scratch_words(int logn, int num_cus);
// rows whose products all have a sparse multiplier (the challenge) as `a` operand: shift-add kernel, no transforms
// images of `count` scalar multipliers (dkey_n per batch entry) for TERM_DKEY terms; flags / bad as in Operands
int launch_dkey_transform(int logn, const LaunchCfg& cfg, const int64_t* g, uint64_t count, uint32_t dkey_n, uint32_t* img,
                          double* l2, const DevTables* T, const uint32_t* d_tw, uint8_t* d_flags, uint32_t* d_bad, bool two_bit,
                          bool trusted);
int launch_shift_rows(int logn, const LaunchCfg& cfg, const Program* d_prog, uint32_t nrows, const Operands& ops,
                      const DevTables* d_T, uint8_t* d_flags, uint64_t batch);
// rows per group; 1 = no grouping (at N = 2048 the accumulators cost too many registers: measured slower)
#ifndef RZK_GROUP_GM
#define RZK_GROUP_GM 4   // accumulators (= rows per group) of row_group_kernel at N <= 1024; at most kGroupMax
#endif
inline int group_max_for(int logn) { return logn >= 11 ? 1 : RZK_GROUP_GM; }
// words of per-wave global scratch of unit_kernel: (max blocks) * 4 waves * 4N (Garner words A, B of rows A, B)
size_t row_scratch_words(int logn, int num_cus);
int launch_key_transform(int logn, const LaunchCfg& cfg, const int64_t* d_key, uint32_t entries,
                         uint32_t* d_key_ntt, const DevTables* d_T, const uint32_t* d_tw);
int launch_ntt(int logn, bool inverse, const LaunchCfg& cfg, int prime, const uint32_t* d_in,
               uint32_t* d_out, uint64_t count, const DevTables* d_T, const uint32_t* d_tw);
int launch_fill_u8(const LaunchCfg& cfg, uint8_t* p, uint8_t value, uint64_t n);
// device-side samplers (rzk_rng.h): npoly polynomials of n_ring coefficients each
int launch_sample_uniform(const LaunchCfg& cfg, int64_t* out, uint64_t npoly, uint32_t n_ring, uint64_t seed,
                          uint32_t stream, uint32_t bound);
int launch_sample_gauss(const LaunchCfg& cfg, int64_t* out, uint64_t npoly, uint32_t n_ring, uint64_t seed,
                        uint32_t stream, double sigma);
int launch_sample_challenge(const LaunchCfg& cfg, int64_t* out, uint64_t npoly, uint32_t n_ring, uint64_t seed,
                            uint32_t stream, uint32_t kappa);
int launch_canonicalize(const LaunchCfg& cfg, const int64_t* in, int64_t* out, uint64_t ncoef, int64_t q);
int launch_addsub(const LaunchCfg& cfg, bool sub, const int64_t* a, const int64_t* b, int64_t* out,
                  uint64_t ncoef, const DevTables* d_T, uint32_t* bad_word);
// ok[b] = (all `rows` polys of proof b have sum c^2 < limit), limit = (bound+1)^2 given as hi:lo.
// and_mode: 0 = overwrite ok[b], 1 = ok[b] &= result, 2 = ok[b] |= result << shift
// qhalf = (q-1)/2, bad_word: see Operands::bad (a non-canonical coefficient also fails the predicate / equality)
int launch_norm(int logn, const LaunchCfg& cfg, const int64_t* v, uint32_t rows, uint64_t limit_hi,
                uint64_t limit_lo, uint8_t* ok, uint64_t B, int and_mode, int shift, uint32_t qhalf,
                uint32_t* bad_word);
int launch_eq(int logn, const LaunchCfg& cfg, const int64_t* a, const int64_t* b, uint32_t rows,
              uint8_t* eq, uint64_t B, uint32_t qhalf, uint32_t* bad_word);
// small ring degrees (N = 4 .. 256): schoolbook products mod q, same row programs
int launch_row_program_small(uint32_t N, const LaunchCfg& cfg, const Program* d_prog, uint32_t nrows,
                             const Operands& ops, const uint32_t* d_key_mont, const DevTables* d_T, uint32_t r2q,
                             uint8_t* d_flags, uint64_t batch);
int launch_key_mont(const LaunchCfg& cfg, const int64_t* d_key, uint32_t* d_key_mont, uint64_t ncoef,
                    const DevTables* d_T, uint32_t r2q);
int launch_norm_small(uint32_t N, const LaunchCfg& cfg, const int64_t* v, uint32_t rows, uint64_t limit_hi,
                      uint64_t limit_lo, uint8_t* ok, uint64_t B, int and_mode, int shift, uint32_t qhalf,
                      uint32_t* bad_word);
int launch_eq_small(uint32_t N, const LaunchCfg& cfg, const int64_t* a, const int64_t* b, uint32_t rows,
                    uint8_t* eq, uint64_t B, uint32_t qhalf, uint32_t* bad_word);

}  // namespace rzk
