// Host side of the input pipeline: the TFRecord container and the tf.train.Example payload that
// st/tools/tftools/tfRecord.py:20-458 writes (save2tfrecord_*: one bytes feature per field) and :460-824 reads
// (tf.TFRecordReader + tf.parse_single_example + tf.decode_raw).  The reference leaves this to TensorFlow's C++ reader
// on 64 queue threads; here a file is memory-mapped by the host code, indexed once by neurst_tfrecord_scan and every
// field is handed out as an (offset, size) view into the mapping - no copies before the batch is assembled.
//   record  = uint64 length | uint32 masked_crc32c(length) | bytes data[length] | uint32 masked_crc32c(data)   (little endian)
//   masked  = rotr(crc, 15) + 0xa282ead8
//   data    = Example{ 1: Features{ repeated 1: map entry{ 1: key, 2: Feature{ 1: BytesList{ repeated 1: value } } } } }
// Pure host code: no HIP call in this file.
#include <nmmintrin.h>

#include <cstdint>
#include <cstring>

#include "../../include/neurst_hip.h"

void neurst_set_error(const char* fmt, ...);

namespace {

inline uint32_t crc32c_bytes(const uint8_t* p, int64_t n) {
  uint64_t c = 0xffffffffu;
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) { c = _mm_crc32_u8((uint32_t)c, *p++); --n; }
  // one crc32q per 8 bytes: ~3 cycles each, i.e. several GB/s per core - an order of magnitude above what one GPU consumes
  for (; n >= 8; n -= 8, p += 8) { uint64_t v; std::memcpy(&v, p, 8); c = _mm_crc32_u64(c, v); }
  while (n-- > 0) c = _mm_crc32_u8((uint32_t)c, *p++);
  return (uint32_t)c ^ 0xffffffffu;
}

inline uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// protobuf varint; returns false when it runs past `end` or is longer than 10 bytes
inline bool varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
  v = 0;
  for (int shift = 0; shift < 70 && p < end; shift += 7) {
    const uint8_t b = *p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}

// skips one field of wire type `wt`
inline bool skip_field(const uint8_t*& p, const uint8_t* end, int wt) {
  uint64_t v;
  switch (wt) {
    case 0: return varint(p, end, v);
    case 1: if (end - p < 8) return false; p += 8; return true;
    case 2: if (!varint(p, end, v) || (uint64_t)(end - p) < v) return false; p += v; return true;
    case 5: if (end - p < 4) return false; p += 4; return true;
    default: return false;
  }
}

// finds the first length-delimited field `field` in [p, end): sub-message bounds in (sp, send)
inline int find_ld(const uint8_t* p, const uint8_t* end, uint32_t field, const uint8_t*& sp, const uint8_t*& send) {
  while (p < end) {
    uint64_t tag, len;
    if (!varint(p, end, tag)) return -1;
    const int wt = (int)(tag & 7);
    if (wt == 2 && (tag >> 3) == field) {
      if (!varint(p, end, len) || (uint64_t)(end - p) < len) return -1;
      sp = p;
      send = p + len;
      return 1;
    }
    if (!skip_field(p, end, wt)) return -1;
  }
  return 0;
}

}  // namespace

extern "C" uint32_t neurst_crc32c(const void* data, int64_t nbytes) {
  return (data && nbytes > 0) ? crc32c_bytes(static_cast<const uint8_t*>(data), nbytes) : 0u;
}

extern "C" int64_t neurst_tfrecord_scan(const void* buf, int64_t nbytes, int64_t* offsets, int64_t* lengths, int64_t max_records,
                                        int verify_crc) {
  if (!buf || nbytes < 0 || max_records < 0 || (max_records > 0 && (!offsets || !lengths))) {
    neurst_set_error("tfrecord_scan: bad args");
    return NEURST_E_INVALID;
  }
  const uint8_t* base = static_cast<const uint8_t*>(buf);
  int64_t pos = 0, n = 0;
  while (pos < nbytes) {
    if (nbytes - pos < 12) {
      neurst_set_error("tfrecord_scan: truncated header of record %lld at byte %lld", (long long)n, (long long)pos);
      return NEURST_E_INVALID;
    }
    const uint64_t len = rd64(base + pos);
    if (verify_crc && mask_crc(crc32c_bytes(base + pos, 8)) != rd32(base + pos + 8)) {
      neurst_set_error("tfrecord_scan: corrupted length of record %lld at byte %lld", (long long)n, (long long)pos);
      return NEURST_E_INVALID;
    }
    if (len > (uint64_t)(nbytes - pos - 12) || (uint64_t)(nbytes - pos - 12) - len < 4) {
      neurst_set_error("tfrecord_scan: record %lld at byte %lld (%llu bytes) runs past the end of the file", (long long)n,
                       (long long)pos, (unsigned long long)len);
      return NEURST_E_INVALID;
    }
    const uint8_t* data = base + pos + 12;
    if (verify_crc && mask_crc(crc32c_bytes(data, (int64_t)len)) != rd32(data + len)) {
      neurst_set_error("tfrecord_scan: corrupted data of record %lld at byte %lld", (long long)n, (long long)pos);
      return NEURST_E_INVALID;
    }
    if (n < max_records) {
      offsets[n] = pos + 12;
      lengths[n] = (int64_t)len;
    }
    ++n;
    pos += 12 + (int64_t)len + 4;
  }
  return n;
}

extern "C" int neurst_example_bytes_feature(const void* record, int64_t nbytes, const char* key, int64_t* offset, int64_t* size) {
  if (!record || nbytes < 0 || !key || !offset || !size) {
    neurst_set_error("example_bytes_feature: bad args");
    return NEURST_E_INVALID;
  }
  const uint8_t* base = static_cast<const uint8_t*>(record);
  const uint8_t *fp, *fend;
  const size_t klen = std::strlen(key);
  int rc = find_ld(base, base + nbytes, 1, fp, fend);   // Example.features
  if (rc < 0) goto malformed;
  if (rc == 0) goto missing;
  while (fp < fend) {                                   // Features.feature map entries
    uint64_t tag, len;
    if (!varint(fp, fend, tag)) goto malformed;
    if ((tag & 7) != 2 || (tag >> 3) != 1) {
      if (!skip_field(fp, fend, (int)(tag & 7))) goto malformed;
      continue;
    }
    if (!varint(fp, fend, len) || (uint64_t)(fend - fp) < len) goto malformed;
    const uint8_t *ep = fp, *eend = fp + len;
    fp = eend;
    const uint8_t *kp, *kend, *vp, *vend, *bp, *bend, *sp, *send;
    if (find_ld(ep, eend, 1, kp, kend) != 1) continue;
    if ((size_t)(kend - kp) != klen || std::memcmp(kp, key, klen) != 0) continue;
    rc = find_ld(ep, eend, 2, vp, vend);                // entry.value = Feature
    if (rc < 0) goto malformed;
    if (rc == 0) goto missing;
    rc = find_ld(vp, vend, 1, bp, bend);                // Feature.bytes_list
    if (rc < 0) goto malformed;
    if (rc == 0) {
      neurst_set_error("example_bytes_feature: feature '%s' is not a bytes list", key);
      return NEURST_E_UNSUPPORTED;
    }
    rc = find_ld(bp, bend, 1, sp, send);                // BytesList.value[0]
    if (rc < 0) goto malformed;
    if (rc == 0) { sp = send = bp; }                    // an empty list reads as an empty string
    *offset = sp - base;
    *size = send - sp;
    return NEURST_OK;
  }
missing:
  neurst_set_error("example_bytes_feature: no feature '%s' in the example", key);
  return NEURST_E_INVALID;
malformed:
  neurst_set_error("example_bytes_feature: malformed Example message");
  return NEURST_E_INVALID;
}

// ------------------------------------------------------------------------------------------------ dev-loop metrics
// Edit distance between two id sequences.  Replaces `editdistance.eval` in batch_cer / batch_wer
// (st/tools/textTools.py:34-70: plain Levenshtein distance) and, when `sid` is given, also returns the alignment counts
// of st/tools/wer.py:124-151: each cell takes the candidate with the smallest substitutions + insertions + deletions,
// ties in the order substitution/match, insertion, deletion (Python's min() keeps the first of equal candidates).
#include <vector>

extern "C" int64_t neurst_edit_distance(const int32_t* ref, int64_t nref, const int32_t* hyp, int64_t nhyp, int64_t* sid) {
  if (nref < 0 || nhyp < 0 || (nref > 0 && !ref) || (nhyp > 0 && !hyp)) {
    neurst_set_error("edit_distance: bad args");
    return NEURST_E_INVALID;
  }
  struct Cell { int64_t s, i, d; int64_t sum() const { return s + i + d; } };
  std::vector<Cell> prev(nhyp + 1), cur(nhyp + 1);
  for (int64_t d = 0; d <= nhyp; ++d) prev[d] = Cell{0, d, 0};          // errors[0, :, 1] = arange
  for (int64_t r = 0; r < nref; ++r) {
    cur[0] = Cell{0, 0, r + 1};                                          // errors[:, 0, 2] = arange
    for (int64_t d = 0; d < nhyp; ++d) {
      Cell best = prev[d];
      best.s += ref[r] != hyp[d];
      Cell ins = cur[d];
      ins.i += 1;
      if (ins.sum() < best.sum()) best = ins;
      Cell del = prev[d + 1];
      del.d += 1;
      if (del.sum() < best.sum()) best = del;
      cur[d + 1] = best;
    }
    prev.swap(cur);
  }
  const Cell& e = prev[nhyp];
  if (sid) { sid[0] = e.s; sid[1] = e.i; sid[2] = e.d; }
  return e.sum();
}
