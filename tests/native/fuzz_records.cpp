// host-only fuzz driver for records.cpp (ASan + UBSan): random and mutated record files / Example payloads
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <vector>
#include <random>
#include "neurst_hip.h"
void neurst_set_error(const char* fmt, ...) { (void)fmt; }
static uint32_t mask(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }
static void put_varint(std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((v & 0x7f) | 0x80); v >>= 7; } o.push_back((uint8_t)v); }
static std::vector<uint8_t> ld(uint32_t f, const std::vector<uint8_t>& p) { std::vector<uint8_t> o; put_varint(o, (f << 3) | 2); put_varint(o, p.size()); o.insert(o.end(), p.begin(), p.end()); return o; }
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  std::mt19937_64 rng(1);
  // a valid file of 3 records with 2 features each
  std::vector<uint8_t> file;
  for (int r = 0; r < 3; ++r) {
    std::vector<uint8_t> entries;
    for (const char* key : {"feature", "label"}) {
      std::vector<uint8_t> raw(17 + r * 5);
      for (auto& b : raw) b = (uint8_t)rng();
      auto feat = ld(1, ld(1, raw));
      std::vector<uint8_t> k(key, key + strlen(key));
      auto e = ld(1, k);
      auto v = ld(2, feat);
      e.insert(e.end(), v.begin(), v.end());
      auto ent = ld(1, e);
      entries.insert(entries.end(), ent.begin(), ent.end());
    }
    auto ex = ld(1, entries);
    uint64_t n = ex.size();
    uint8_t head[12];
    memcpy(head, &n, 8);
    uint32_t c = mask(neurst_crc32c(head, 8));
    memcpy(head + 8, &c, 4);
    file.insert(file.end(), head, head + 12);
    file.insert(file.end(), ex.begin(), ex.end());
    c = mask(neurst_crc32c(ex.data(), (int64_t)ex.size()));
    uint8_t tail[4];
    memcpy(tail, &c, 4);
    file.insert(file.end(), tail, tail + 4);
  }
  int64_t off[8], len[8];
  int64_t n = neurst_tfrecord_scan(file.data(), (int64_t)file.size(), off, len, 8, 1);
  if (n != 3) { printf("valid file: %lld records\n", (long long)n); return 1; }
  int64_t o, s;
  if (neurst_example_bytes_feature(file.data() + off[1], len[1], "label", &o, &s) != 0 || s != 22) { printf("label lookup failed (%lld)\n", (long long)s); return 1; }
  long ok = 0, bad = 0, found = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> m = file;
    const int nmut = 1 + (int)(rng() % 4);
    for (int k = 0; k < nmut; ++k) {
      const int kind = (int)(rng() % 3);
      if (kind == 0) m[rng() % m.size()] ^= (uint8_t)(1u << (rng() % 8));
      else if (kind == 1) m.resize(rng() % (m.size() + 1));
      else m[rng() % (m.size() ? m.size() : 1)] = (uint8_t)rng();
      if (m.empty()) break;
    }
    // exact-size heap copy so that ASan sees any read past the end
    uint8_t* buf = (uint8_t*)malloc(m.size() ? m.size() : 1);
    memcpy(buf, m.data(), m.size());
    const int verify = (int)(rng() & 1);
    int64_t cnt = neurst_tfrecord_scan(buf, (int64_t)m.size(), off, len, 8, verify);
    if (cnt >= 0) {
      ++ok;
      for (int64_t r = 0; r < cnt && r < 8; ++r)
        for (const char* key : {"feature", "label", "id"})
          if (neurst_example_bytes_feature(buf + off[r], len[r], key, &o, &s) == 0) {
            if (o < 0 || s < 0 || o + s > len[r]) { printf("view outside the record\n"); return 1; }
            volatile uint8_t sink = 0;
            for (int64_t i = 0; i < s; ++i) sink ^= buf[off[r] + o + i];
            ++found;
          }
    } else ++bad;
    free(buf);
  }
  // raw Example payload fuzz (no container)
  for (int it = 0; it < iters; ++it) {
    const size_t sz = rng() % 64;
    uint8_t* buf = (uint8_t*)malloc(sz ? sz : 1);
    for (size_t i = 0; i < sz; ++i) buf[i] = (rng() % 4 == 0) ? (uint8_t)rng() : (uint8_t)(rng() % 16);
    if (neurst_example_bytes_feature(buf, (int64_t)sz, "a", &o, &s) == 0 && (o < 0 || s < 0 || o + s > (int64_t)sz)) { printf("view outside payload\n"); return 1; }
    free(buf);
  }
  int64_t sid[3];
  int32_t a[5] = {1, 2, 3, 4, 5}, b[3] = {1, 3, 5};
  if (neurst_edit_distance(a, 5, b, 3, sid) != 2 || sid[2] != 2) { printf("edit distance\n"); return 1; }
  if (neurst_edit_distance(nullptr, 0, b, 3, sid) != 3) { printf("edit distance empty\n"); return 1; }
  printf("fuzz ok: %ld files accepted, %ld rejected, %ld fields read\n", ok, bad, found);
  return 0;
}
