// fuzz_host_io.cpp -- sanitizer run of the HOST-ONLY parts of libpclhip: the PCD reader / writer (pcd_io.cpp) and the slab
// partition (shard.cpp), compiled by g++ with -fsanitize=address,undefined together with this file
// (tests/test_host_sanitizers.py builds and runs it; no GPU, no HIP runtime: the three symbols those two files take from
// the rest of the library are stubbed below).
//
// What it does, with a fixed seed (argv[1]) for argv[2] rounds:
//   * writes clouds in the three PCD encodings (ascii / binary / binary_compressed, PointXYZ and PointNormal records,
//     organized and not), reads them back and compares the records bit for bit -- the round trip the reference's
//     test/io/test_io.cpp makes;
//   * then MUTATES the file (byte flips, truncation, header tokens replaced by hostile values: huge / negative / zero
//     counts and sizes, missing lines, DATA before FIELDS, compressed sizes that do not match the stream, LZF back
//     references before the start of the output) and runs every reader entry point on it.  The readers may accept or
//     reject a mutant; they may not crash, read or write out of bounds, overflow an integer or leak -- which is what
//     the sanitizers check;
//   * partitions random clouds (duplicates, NaNs, fewer points than slabs) into slabs and checks that the regions tile
//     space: every finite point has exactly one owner, and select_region(margin 0) returns a superset of the owned ones.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../pcl_amd/csrc/pclhip_internal.hpp"
#include "pclhip.h"

// ---- the symbols pcd_io.cpp / shard.cpp take from the rest of the library (no pointer here is device memory, so the
// device-side partition of shard_dev.hip is never reached) ------------------------------------------------------------
namespace pclhip {
bool is_device_pointer(const void*) { return false; }
void set_error(pclhip_ctx*, const std::string&) {}
pclhip_status partition_slabs_device(const void*, size_t, uint64_t, int, float*) { return PCLHIP_ERR_HIP; }
pclhip_status select_region_device(const void*, size_t, uint64_t, const float*, const float*, int32_t*, uint64_t, uint64_t*) {
  return PCLHIP_ERR_HIP;
}
}  // namespace pclhip
extern "C" hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind) { return hipErrorNotSupported; }

namespace {

struct Rng {
  uint64_t s;
  uint64_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return n ? uint32_t(next() % n) : 0u; }
  float unit() { return float(next() >> 40) * (1.0f / 16777216.0f); }
};

std::vector<char> slurp(const std::string& path) {
  std::vector<char> out;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return out;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
  fclose(f);
  return out;
}
void spit(const std::string& path, const std::vector<char>& data) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) abort();
  if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) abort();
  fclose(f);
}

int fails = 0;
#define CHECK(cond, ...)                      \
  do {                                        \
    if (!(cond)) {                            \
      ++fails;                                \
      fprintf(stderr, "CHECK failed: " __VA_ARGS__); \
      fprintf(stderr, " (%s:%d)\n", __FILE__, __LINE__); \
    }                                         \
  } while (0)

void read_everything_untimed(const std::string& path);
int slow_files = 0;
// ... timed: a reader that spends seconds on a 30 KB file has been talked into a huge allocation or loop by its header
void read_everything(const std::string& path) {
  const auto t0 = std::chrono::steady_clock::now();
  read_everything_untimed(path);
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (s > 1.0) {
    const std::string keep = path + ".slow" + std::to_string(slow_files++);
    spit(keep, slurp(path));
    fprintf(stderr, "SLOW: %.1f s for %s\n", s, keep.c_str());
  }
}
// every reader entry point on `path`; buffers are sized from the header the reader itself reports, capped
void read_everything_untimed(const std::string& path) {
  pclhip_pcd_info info;
  memset(&info, 0, sizeof info);
  if (pclhip_pcd_read_header(path.c_str(), &info) != PCLHIP_OK) return;
  const uint64_t cap = info.points < 200000 ? info.points : 200000;  // hostile POINTS values must not size our buffers
  for (int layout = 0; layout < 3; ++layout) {
    const size_t stride = layout == 0 ? 16 : layout == 1 ? 48 : 12, noff = layout == 1 ? 16 : 0;
    std::vector<char> rec(size_t(cap) * stride + 16);
    uint64_t n = 0;
    int dense = 0;
    (void)pclhip_pcd_read(path.c_str(), rec.data(), stride, noff, cap, &n, &dense);
    // a short buffer: must report OVERFLOW (or an error), never write past `capacity` records
    const uint64_t small = cap / 2;
    std::vector<char> rec2(size_t(small) * stride + 16, char(0x5A));
    (void)pclhip_pcd_read(path.c_str(), rec2.data(), stride, noff, small, &n, &dense);
    for (size_t i = size_t(small) * stride; i < rec2.size(); ++i)
      CHECK(rec2[i] == char(0x5A), "pcd_read wrote past the capacity it was given");
  }
  static const char* names[] = {"x", "y", "z", "normal_x", "curvature", "intensity", "rgb", "rgba", "no_such_field", ""};
  std::vector<float> col(size_t(cap) + 4);
  for (const char* f : names)
    for (uint32_t comp = 0; comp < 3; ++comp) {
      uint64_t n = 0;
      (void)pclhip_pcd_read_field(path.c_str(), f, comp, col.data(), cap, &n);
    }
}

const char* hostile_tokens[] = {"0",          "-1",        "4294967295", "4294967296",           "18446744073709551615",
                                "99999999999999999999999", "nan",        "1e30",                 "",
                                "x",          "3 3 3",     "F F F F F F F F F F F F F F F F",     "8",
                                "2147483647", "-2147483648", "0x10",     "1 1 1 1 1 1 1 1 1 1 1", "16777216"};

void mutate(Rng& rng, std::vector<char>& d) {
  if (d.empty()) return;
  switch (rng.below(7)) {
    case 0:  // byte flips anywhere
      for (uint32_t i = 0, k = 1 + rng.below(8); i < k; ++i) d[rng.below(uint32_t(d.size()))] ^= char(1u << rng.below(8));
      break;
    case 1:  // byte flips in the header (first 300 bytes)
      for (uint32_t i = 0, k = 1 + rng.below(4); i < k; ++i)
        d[rng.below(uint32_t(d.size() < 300 ? d.size() : 300))] = char(rng.below(256));
      break;
    case 2:  // truncate
      d.resize(rng.below(uint32_t(d.size())));
      break;
    case 3: {  // replace the value of one header line by a hostile token
      static const char* keys[] = {"FIELDS", "SIZE", "TYPE", "COUNT", "WIDTH", "HEIGHT", "VIEWPOINT", "POINTS", "DATA", "VERSION"};
      const std::string key = keys[rng.below(10)];
      std::string s(d.begin(), d.end());
      const size_t at = s.find(key);
      if (at == std::string::npos) break;
      const size_t eol = s.find('\n', at);
      if (eol == std::string::npos) break;
      const std::string tok = hostile_tokens[rng.below(sizeof hostile_tokens / sizeof *hostile_tokens)];
      s.replace(at + key.size(), eol - at - key.size(), " " + tok);
      d.assign(s.begin(), s.end());
      break;
    }
    case 4: {  // drop one header line
      std::string s(d.begin(), d.end());
      const size_t hdr = s.find("DATA");
      if (hdr == std::string::npos || hdr == 0) break;
      const size_t at = s.rfind('\n', rng.below(uint32_t(hdr)));
      if (at == std::string::npos) break;
      const size_t eol = s.find('\n', at + 1);
      if (eol == std::string::npos) break;
      s.erase(at, eol - at);
      d.assign(s.begin(), s.end());
      break;
    }
    case 5: {  // the two size words of a compressed body (or whatever follows the header)
      std::string s(d.begin(), d.end());
      const size_t at = s.find("DATA");
      if (at == std::string::npos) break;
      const size_t eol = s.find('\n', at);
      if (eol == std::string::npos || eol + 9 > d.size()) break;
      static const uint32_t sizes[] = {0u, 1u, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, 12u, 1u << 20};
      const uint32_t v = sizes[rng.below(7)];
      memcpy(&d[eol + 1 + 4 * rng.below(2)], &v, 4);
      break;
    }
    default:  // garbage appended / body replaced by noise
      for (size_t i = d.size() / 2; i < d.size(); ++i)
        if (rng.below(4) == 0) d[i] = char(rng.below(256));
      break;
  }
}

void pcd_round(Rng& rng, const std::string& dir, int round) {
  const bool normals = rng.below(2) == 1;
  const size_t stride = normals ? 48 : 16, noff = normals ? 16 : 0;
  const uint32_t w = 1 + rng.below(40), h = rng.below(3) == 0 ? 1 + rng.below(6) : 1;
  const uint64_t n = uint64_t(w) * h;
  std::vector<float> rec(size_t(n) * stride / 4, 0.0f);
  for (uint64_t i = 0; i < n; ++i) {
    float* p = &rec[size_t(i) * stride / 4];
    for (int d = 0; d < 3; ++d) p[d] = (rng.unit() - 0.5f) * (rng.below(8) == 0 ? 1e6f : 2.0f);
    if (rng.below(23) == 0) p[rng.below(3)] = NAN;
    p[3] = 1.0f;
    if (normals) {
      for (int d = 0; d < 3; ++d) p[4 + d] = rng.unit() - 0.5f;
      p[8] = rng.unit();
    }
  }
  const int data_type = int(rng.below(3));
  const std::string path = dir + "/fuzz_" + std::to_string(round % 4) + ".pcd";
  float vp[7] = {rng.unit(), rng.unit(), rng.unit(), 1, 0, 0, 0};
  pclhip_status st = h > 1 ? pclhip_pcd_write_organized(path.c_str(), rec.data(), stride, noff, w, h, vp, data_type, 0)
                           : pclhip_pcd_write(path.c_str(), rec.data(), stride, noff, n, data_type, 0);
  CHECK(st == PCLHIP_OK, "pcd_write failed (%d)", int(st));
  if (st != PCLHIP_OK) return;
  // ---- round trip (binary encodings: bit for bit; ascii: the default precision of 8 digits keeps float32 to 1 ulp-ish,
  // so only the structure is checked there) ----
  pclhip_pcd_info info;
  CHECK(pclhip_pcd_read_header(path.c_str(), &info) == PCLHIP_OK, "own file rejected");
  CHECK(info.points == n && info.width == w && info.height == h && info.data_type == data_type, "header does not round-trip");
  std::vector<float> back(rec.size(), -7.0f);
  uint64_t nb = 0;
  int dense = -1;
  CHECK(pclhip_pcd_read(path.c_str(), back.data(), stride, noff, n, &nb, &dense) == PCLHIP_OK && nb == n, "own file unreadable");
  if (data_type != 0)
    for (uint64_t i = 0; i < n; ++i) {
      const float *a = &rec[size_t(i) * stride / 4], *b = &back[size_t(i) * stride / 4];
      CHECK(memcmp(a, b, 12) == 0, "xyz of point %llu does not round-trip", (unsigned long long)i);
      if (normals) CHECK(memcmp(a + 4, b + 4, 12) == 0 && memcmp(a + 8, b + 8, 4) == 0, "normal of point %llu", (unsigned long long)i);
    }
  // ---- mutants ----
  const std::vector<char> good = slurp(path);
  const std::string mpath = dir + "/mutant.pcd";
  for (int m = 0; m < 12; ++m) {
    std::vector<char> d = good;
    for (uint32_t k = 0, kk = 1 + rng.below(3); k < kk; ++k) mutate(rng, d);
    spit(mpath, d);
    read_everything(mpath);
  }
}

void shard_round(Rng& rng) {
  const uint32_t n = rng.below(5) == 0 ? rng.below(6) : 1 + rng.below(3000);
  const int slabs = 1 + int(rng.below(9));
  const size_t stride = rng.below(2) ? 16 : 12;
  std::vector<float> pts(size_t(n) * stride / 4 + 4, 1.0f);
  const int kind = int(rng.below(4));
  for (uint32_t i = 0; i < n; ++i) {
    float* p = &pts[size_t(i) * stride / 4];
    for (int d = 0; d < 3; ++d) {
      float v = rng.unit() * 2.0f - 1.0f;
      if (kind == 1) v = float(int(v * 3.0f));          // lattice: heavy ties
      if (kind == 2 && d > 0) v = 0.25f;                 // a line
      if (kind == 3) v = 0.5f;                           // one site
      p[d] = v;
    }
    if (rng.below(50) == 0) p[rng.below(3)] = rng.below(2) ? NAN : INFINITY;
  }
  std::vector<float> regions(size_t(slabs) * 6);
  const pclhip_status st = pclhip_partition_slabs(pts.data(), stride, n, slabs, regions.data());
  CHECK(st == PCLHIP_OK, "partition_slabs failed (%d)", int(st));
  if (st != PCLHIP_OK) return;
  std::vector<uint32_t> owned(size_t(slabs), 0u);
  for (uint32_t i = 0; i < n; ++i) {
    const float* p = &pts[size_t(i) * stride / 4];
    int owners = 0;
    for (int g = 0; g < slabs; ++g) {
      const float* r = &regions[size_t(g) * 6];
      if (p[0] >= r[0] && p[0] < r[3] && p[1] >= r[1] && p[1] < r[4] && p[2] >= r[2] && p[2] < r[5]) ++owners;
    }
    const bool finite = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
    CHECK(owners == (finite ? 1 : 0), "point %u has %d owners among %d slabs", i, owners, slabs);
    const int o = pclhip_region_owner(regions.data(), slabs, p);
    CHECK((o >= 0) == finite, "region_owner disagrees on point %u", i);
    if (o >= 0) ++owned[size_t(o)];
  }
  for (int g = 0; g < slabs; ++g) {
    std::vector<int32_t> idx(size_t(n) + 1);
    uint64_t cnt = 0;
    const double margin = rng.below(2) ? 0.0 : double(rng.unit()) * 0.3;
    const pclhip_status s2 = pclhip_select_region(pts.data(), stride, n, &regions[size_t(g) * 6], margin, idx.data(), n, &cnt);
    CHECK(s2 == PCLHIP_OK && cnt >= owned[size_t(g)] && cnt <= n, "select_region: %llu selected, %u owned", (unsigned long long)cnt,
          owned[size_t(g)]);
    for (uint64_t k = 1; k < cnt && s2 == PCLHIP_OK; ++k) CHECK(idx[k - 1] < idx[k], "selection not ascending");
    // too small a buffer: the count is still reported, nothing is written
    if (cnt > 1) {
      uint64_t c2 = 0;
      std::vector<int32_t> tiny(1, -5);
      CHECK(pclhip_select_region(pts.data(), stride, n, &regions[size_t(g) * 6], margin, tiny.data(), 1, &c2) == PCLHIP_ERR_OVERFLOW &&
                c2 == cnt && tiny[0] == -5,
            "select_region with a short buffer");
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
  const int rounds = argc > 2 ? atoi(argv[2]) : 200;
  const std::string dir = argc > 3 ? argv[3] : "/tmp";
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 12345};
  for (int r = 0; r < rounds; ++r) {
    pcd_round(rng, dir, r);
    shard_round(rng);
  }
  // the reference's own files, mutated (argv[4..])
  for (int a = 4; a < argc; ++a) {
    const std::vector<char> good = slurp(argv[a]);
    read_everything(argv[a]);
    for (int m = 0; m < 40; ++m) {
      std::vector<char> d = good;
      for (uint32_t k = 0, kk = 1 + rng.below(3); k < kk; ++k) mutate(rng, d);
      spit(dir + "/mutant.pcd", d);
      read_everything(dir + "/mutant.pcd");
    }
  }
  printf("fuzz_host_io: seed %llu, %d rounds, %d failed checks, %d slow reads\n", (unsigned long long)seed, rounds, fails,
         slow_files);
  return (fails || slow_files) ? 1 : 0;
}
