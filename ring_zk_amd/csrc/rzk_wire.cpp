// rzk_wire.cpp — bincode layout of the reference's Mat<I, N> <-> the dense [rows][cols][N] int64 slabs of the
// C ABI (host code, no GPU).
//
// The reference derives serde on Mat (src/mat.rs:11-14: one field `polynomials: Vec<Vec<Polynomial>>`) and on
// every protocol message (src/commit.rs:134, src/prove/open.rs:180-228, ...); its own test serialises with
// bincode's default options (src/mat.rs:424-438): little-endian, fixed-width integers, u64 length prefixes,
// struct fields in declaration order without tags.  A polynomial is its coefficient vector in the crate's
// trimmed representation (no trailing zeros; src/mat.rs:430-434 counts 8 + 3*4 bytes for 1 + 2x + 3x^2 over
// i32).  So a Mat is
//     u64 rows ; rows x { u64 cols ; cols x { u64 len ; len x coefficient } }
// and a message struct is the concatenation of its Mat / Polynomial fields.
// The coefficient width of ZqI64<Q> on the wire is a property of the third-party poly-ring-xnp1 crate that
// no file of the reference pins; both 8-byte (i64, the natural reading) and 4-byte (the width the
// reference's test uses) are supported and the caller says which.
#include <cstring>

#include "../../include/rzk.h"

namespace {

inline void put_u64(uint8_t*& p, uint64_t v) {
  for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(v >> (8 * i));
}
inline uint64_t get_u64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i);
  return v;
}
inline uint32_t trimmed_len(const int64_t* poly, uint32_t N) {
  uint32_t len = N;
  while (len > 0 && poly[len - 1] == 0) --len;
  return len;
}

}  // namespace

extern "C" {

size_t rzk_wire_mat_size(const int64_t* slab, uint32_t rows, uint32_t cols, uint32_t N, uint32_t coef_bytes) {
  if (!slab || (coef_bytes != 4 && coef_bytes != 8)) return 0;
  size_t total = 8 + (size_t)rows * 8;
  for (size_t i = 0; i < (size_t)rows * cols; ++i) total += 8 + (size_t)trimmed_len(slab + i * N, N) * coef_bytes;
  return total;
}

int rzk_wire_mat_encode(const int64_t* slab, uint32_t rows, uint32_t cols, uint32_t N, uint32_t coef_bytes,
                        uint8_t* out, size_t cap, size_t* written) {
  if (!slab || !out || (coef_bytes != 4 && coef_bytes != 8)) return RZK_E_ARG;
  const size_t need = rzk_wire_mat_size(slab, rows, cols, N, coef_bytes);
  if (written) *written = need;
  if (cap < need) return RZK_E_ARG;
  uint8_t* p = out;
  put_u64(p, rows);
  for (uint32_t r = 0; r < rows; ++r) {
    put_u64(p, cols);
    for (uint32_t c = 0; c < cols; ++c) {
      const int64_t* poly = slab + ((size_t)r * cols + c) * N;
      const uint32_t len = trimmed_len(poly, N);
      put_u64(p, len);
      for (uint32_t i = 0; i < len; ++i) {
        if (coef_bytes == 4 && (poly[i] < INT32_MIN || poly[i] > INT32_MAX)) return RZK_E_ARG;
        const uint64_t v = (uint64_t)poly[i];
        for (uint32_t b = 0; b < coef_bytes; ++b) *p++ = (uint8_t)(v >> (8 * b));
      }
    }
  }
  return RZK_OK;
}

int rzk_wire_mat_decode(const uint8_t* in, size_t len, uint32_t N, uint32_t coef_bytes, int64_t q, uint32_t* rows_out,
                        uint32_t* cols_out, int64_t* slab, size_t slab_polys, size_t* consumed) {
  if (!in || (coef_bytes != 4 && coef_bytes != 8) || !rows_out || !cols_out || q < 0) return RZK_E_ARG;
  const int64_t half = q > 0 ? (q - 1) / 2 : 0;   // q > 0: coefficients must be centred residues mod q
  size_t pos = 0;
  auto need = [&](size_t n) { return len - pos >= n; };
  if (!need(8)) return RZK_E_ARG;
  const uint64_t rows = get_u64(in + pos);
  pos += 8;
  if (rows > 0xffffffffull) return RZK_E_ARG;
  uint64_t cols = 0;
  size_t poly_index = 0;
  for (uint64_t r = 0; r < rows; ++r) {
    if (!need(8)) return RZK_E_ARG;
    const uint64_t c = get_u64(in + pos);
    pos += 8;
    if (r == 0) cols = c;
    if (c != cols || c > 0xffffffffull) return RZK_E_ARG;   // ragged rows: not a matrix (Mat::from_vec shapes)
    for (uint64_t j = 0; j < c; ++j) {
      if (!need(8)) return RZK_E_ARG;
      const uint64_t plen = get_u64(in + pos);
      pos += 8;
      if (plen > N || !need(plen * coef_bytes)) return RZK_E_ARG;   // degree >= N cannot be in Z[X]/(X^N+1)
      if (slab && poly_index >= slab_polys) return RZK_E_ARG;
      int64_t* poly = slab ? slab + poly_index * N : nullptr;
      for (uint64_t i = 0; i < plen; ++i) {
        uint64_t v = 0;
        for (uint32_t b = 0; b < coef_bytes; ++b) v |= (uint64_t)in[pos + i * coef_bytes + b] << (8 * b);
        const int64_t cv = coef_bytes == 4 ? (int64_t)(int32_t)(uint32_t)v : (int64_t)v;
        // a ZqI64 holds the centred representative (src/params.rs:122-127): anything else on the wire is not a
        // ring element and must not reach the verifier kernels as its low word
        if (q > 0 && (cv > half || cv < -half)) return RZK_E_ARG;
        if (poly) poly[i] = cv;
      }
      if (poly) std::memset(poly + plen, 0, (size_t)(N - plen) * sizeof(int64_t));   // trimmed -> dense
      pos += plen * coef_bytes;
      ++poly_index;
    }
  }
  *rows_out = (uint32_t)rows;
  *cols_out = (uint32_t)cols;
  if (consumed) *consumed = pos;
  return RZK_OK;
}

}  // extern "C"
