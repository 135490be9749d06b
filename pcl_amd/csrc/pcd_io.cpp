// pcd_io.cpp -- PCD reader / writer for the clouds either side of the ICP path (SURVEY.md section 8(f)
// rank 4): ascii, binary and binary_compressed (LZF) files straight into the strided point records the
// rest of the C ABI consumes (PointXYZ 16 B, PointNormal 48 B, ...; host, pinned or device memory).
//
// Format and semantics follow io/src/pcd_io.cpp of the reference:
//   header      PCDReader::readHeader            :115-392  (FIELDS/SIZE/TYPE/COUNT/WIDTH/HEIGHT/VIEWPOINT/POINTS/DATA)
//   ascii body  PCDReader::readBodyASCII         :456-559  (one point per line, "nan" allowed)
//   binary      PCDReader::readBodyBinary        :561-675  (array of structs; compressed: u32 compressed size,
//                                                            u32 uncompressed size, LZF stream holding the fields as
//                                                            struct of arrays: all x, all y, ...)
//   writer      PCDWriter::generateHeader*, writeASCII / writeBinary / writeBinaryCompressed :848-1500
// LZF is liblzf's format (Marc Lehmann), which the reference vendors as io/src/lzf.cpp: a control byte
// c < 32 starts a literal run of c + 1 bytes; otherwise a back reference of length (c >> 5) + 2 (a
// length field of 7 is extended by the next byte) at distance (((c & 31) << 8) | next byte) + 1.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <string_view>
#include <thread>
#include <utility>
#include <vector>

#include "pclhip_internal.hpp"

using namespace pclhip;

namespace {

struct Field {
  std::string name;
  int size = 4;      // bytes per element
  char type = 'F';   // F float, I signed, U unsigned
  int count = 1;
  size_t offset = 0;  // byte offset inside one point (array-of-structs layout)
};

struct Header {
  std::vector<Field> fields;
  uint64_t points = 0;
  uint32_t width = 0, height = 0;
  int data_type = 0, version = 6;
  size_t point_step = 0;
  float viewpoint[7] = {0, 0, 0, 1, 0, 0, 0};
  uint64_t data_offset = 0;
};

std::vector<std::string> split_ws(const std::string& line) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) ++i;
    size_t j = i;
    while (j < line.size() && line[j] != ' ' && line[j] != '\t' && line[j] != '\r') ++j;
    if (j > i) out.push_back(line.substr(i, j - i));
    i = j;
  }
  return out;
}

bool starts_with(const std::string& s, const char* p) { return s.compare(0, std::strlen(p), p) == 0; }

// PCDReader::readHeader, io/src/pcd_io.cpp:115-392.  `err` receives the reference's message on failure.
bool parse_header(std::string_view bytes, Header& h, std::string& err) {
  size_t pos = 0;
  bool width_read = false, height_read = false, points_read = false, data_seen = false;
  std::vector<int> sizes;
  std::vector<char> types;
  while (pos < bytes.size()) {
    size_t eol = bytes.find('\n', pos);
    if (eol == std::string::npos) eol = bytes.size();
    const std::string line(bytes.substr(pos, eol - pos));
    pos = eol + 1;
    const std::vector<std::string> st = split_ws(line);
    if (st.empty()) continue;
    const std::string& key = st[0];
    if (key[0] == '#') continue;
    if (starts_with(key, "VERSION")) continue;
    if (starts_with(key, "FIELDS") || starts_with(key, "COLUMNS")) {
      h.fields.assign(st.size() - 1, Field());
      sizes.clear();  // a repeated FIELDS line starts over: SIZE / TYPE / COUNT must follow it again
      types.clear();
      size_t off = 0;
      for (size_t i = 0; i + 1 < st.size(); ++i) {  // older files: everything float32 unless SIZE/TYPE say otherwise
        h.fields[i].name = st[i + 1];
        h.fields[i].offset = off;
        off += 4;
      }
      h.point_step = off;
      continue;
    }
    if (starts_with(key, "SIZE")) {
      if (st.size() - 1 != h.fields.size()) {
        err = "The number of elements in <SIZE> differs than the number of elements in <FIELDS>!";
        return false;
      }
      sizes.resize(h.fields.size());
      size_t off = 0;
      for (size_t i = 0; i < h.fields.size(); ++i) {
        sizes[i] = std::atoi(st[i + 1].c_str());
        if (sizes[i] != 1 && sizes[i] != 2 && sizes[i] != 4 && sizes[i] != 8) {
          err = "Invalid SIZE value (must be 1, 2, 4 or 8): " + st[i + 1];
          return false;
        }
        h.fields[i].size = sizes[i];
        h.fields[i].offset = off;
        off += size_t(sizes[i]);
      }
      h.point_step = off;
      continue;
    }
    if (starts_with(key, "TYPE")) {
      if (sizes.empty()) {
        err = "TYPE of FIELDS specified before SIZE in header!";
        return false;
      }
      if (st.size() - 1 != h.fields.size()) {
        err = "The number of elements in <TYPE> differs than the number of elements in <FIELDS>!";
        return false;
      }
      types.resize(h.fields.size());
      for (size_t i = 0; i < h.fields.size(); ++i) {
        types[i] = st[i + 1][0];
        if (types[i] != 'I' && types[i] != 'U' && types[i] != 'F') {
          err = "Invalid TYPE value (must be I, U or F): " + st[i + 1];
          return false;
        }
        if (types[i] == 'F' && sizes[i] != 4 && sizes[i] != 8) {
          err = "TYPE F needs SIZE 4 or 8";
          return false;
        }
        h.fields[i].type = types[i];
      }
      continue;
    }
    if (starts_with(key, "COUNT")) {
      if (sizes.size() != h.fields.size() || types.size() != h.fields.size()) {
        err = "COUNT of FIELDS specified before SIZE or TYPE in header!";
        return false;
      }
      if (st.size() - 1 != h.fields.size()) {
        err = "The number of elements in <COUNT> differs than the number of elements in <FIELDS>!";
        return false;
      }
      size_t off = 0;
      for (size_t i = 0; i < h.fields.size(); ++i) {
        h.fields[i].offset = off;
        h.fields[i].count = std::atoi(st[i + 1].c_str());
        if (h.fields[i].count < 0 || h.fields[i].count > (1 << 20)) {
          err = "Invalid COUNT value: " + st[i + 1];
          return false;
        }
        off += size_t(h.fields[i].count) * size_t(sizes[i]);
      }
      h.point_step = off;
      continue;
    }
    if (starts_with(key, "WIDTH")) {
      if (st.size() < 2) {
        err = "Invalid WIDTH value specified.";
        return false;
      }
      h.width = uint32_t(std::strtoul(st[1].c_str(), nullptr, 10));
      width_read = true;
      continue;
    }
    if (starts_with(key, "HEIGHT")) {
      if (st.size() < 2) {
        err = "Invalid HEIGHT value specified.";
        return false;
      }
      h.height = uint32_t(std::strtoul(st[1].c_str(), nullptr, 10));
      height_read = true;
      continue;
    }
    if (starts_with(key, "VIEWPOINT")) {
      h.version = 7;
      if (st.size() < 8) {
        err = "Not enough number of elements in <VIEWPOINT>! Need 7 values (tx ty tz qw qx qy qz).";
        return false;
      }
      for (int i = 0; i < 7; ++i) h.viewpoint[i] = std::strtof(st[size_t(i) + 1].c_str(), nullptr);
      continue;
    }
    if (starts_with(key, "POINTS")) {
      if (h.point_step == 0) {
        err = "Number of POINTS specified before COUNT in header!";
        return false;
      }
      h.points = st.size() > 1 ? std::strtoull(st[1].c_str(), nullptr, 10) : 0;
      points_read = true;
      continue;
    }
    if (starts_with(key, "DATA")) {
      if (st.size() < 2) continue;
      if (starts_with(st[1], "binary_compressed"))
        h.data_type = 2;
      else if (starts_with(st[1], "binary"))
        h.data_type = 1;
      else if (starts_with(st[1], "ascii"))
        h.data_type = 0;
      else
        continue;  // unknown DATA format: the reference warns and reads on
      h.data_offset = pos;
      data_seen = true;
    }
    break;  // DATA is the last header entry (also: any unknown line ends the header, as in the reference)
  }
  (void)points_read;
  (void)data_seen;
  // untrusted input: every field must lie inside the record, sizes must be consistent, and
  // points * point_step must not overflow
  if (!h.fields.empty() && !sizes.empty() && sizes.size() != h.fields.size()) {
    err = "SIZE does not match the last FIELDS line";
    return false;
  }
  for (const Field& f : h.fields) {
    const size_t cnt = f.count > 0 ? size_t(f.count) : 0;
    if (f.size <= 0 || f.offset + cnt * size_t(f.size) > h.point_step) {
      err = "field '" + f.name + "' does not fit the record described by SIZE/COUNT";
      return false;
    }
  }
  if (h.point_step > (1u << 24) || (h.point_step != 0 && h.points > (uint64_t(1) << 62) / h.point_step)) {
    err = "POINTS x point size overflows";
    return false;
  }
  // fields with COUNT < 1 are dropped (:339-341)
  {
    std::vector<Field> kept;
    for (const Field& f : h.fields)
      if (f.count >= 1) kept.push_back(f);
    h.fields.swap(kept);
  }
  if (!width_read && !height_read) {
    h.width = uint32_t(h.points);
    h.height = 1;
  }
  if (!height_read) {
    h.height = 1;
    if (h.width == 0) h.width = uint32_t(h.points);
  } else if (h.width == 0 && h.points != 0) {
    err = "HEIGHT given but no WIDTH!";
    return false;
  }
  if (uint64_t(h.width) * uint64_t(h.height) != h.points) {
    err = "HEIGHT x WIDTH != number of points";
    return false;
  }
  return true;
}

// liblzf decompression (the reference's pcl::lzfDecompress, io/src/lzf.cpp): returns the number of
// bytes produced, 0 on malformed input or if the output would not fit
size_t lzf_decompress(const unsigned char* ip, size_t in_len, unsigned char* op, size_t out_len) {
  const unsigned char* const in_end = ip + in_len;
  unsigned char* const out_begin = op;
  unsigned char* const out_end = op + out_len;
  while (ip < in_end) {
    unsigned ctrl = *ip++;
    if (ctrl < 32) {  // literal run
      ++ctrl;
      if (op + ctrl > out_end || ip + ctrl > in_end) return 0;
      std::memcpy(op, ip, ctrl);
      op += ctrl;
      ip += ctrl;
    } else {  // back reference
      size_t len = ctrl >> 5;
      if (ip >= in_end) return 0;
      if (len == 7) {
        len += *ip++;
        if (ip >= in_end) return 0;
      }
      const size_t dist = (size_t(ctrl & 0x1f) << 8) + size_t(*ip++) + 1;
      len += 2;
      if (size_t(op - out_begin) < dist || op + len > out_end) return 0;
      const unsigned char* ref = op - dist;
      for (size_t i = 0; i < len; ++i) op[i] = ref[i];  // may overlap: byte by byte
      op += len;
    }
  }
  return size_t(op - out_begin);
}

// liblzf-format compressor: greedy matcher over a hash of 3-byte sequences.  Any stream this produces
// is decoded by the reference's lzfDecompress; it is not byte-identical to the reference's compressor
// (nothing depends on that).  Returns 0 if the result would not fit into out_len.
size_t lzf_compress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  constexpr int HLOG = 16;
  constexpr size_t MAX_LIT = 32, MAX_OFF = 1u << 13, MAX_REF = (1u << 8) + (1u << 3);
  std::vector<uint32_t> htab(size_t(1) << HLOG, 0xFFFFFFFFu);
  size_t ip = 0, op = 0, lit_start = 0;
  auto flush_literals = [&](size_t end) -> bool {
    while (lit_start < end) {
      const size_t run = (end - lit_start) < MAX_LIT ? (end - lit_start) : MAX_LIT;
      if (op + 1 + run > out_len) return false;
      out[op++] = (unsigned char)(run - 1);
      std::memcpy(out + op, in + lit_start, run);
      op += run;
      lit_start += run;
    }
    return true;
  };
  while (ip + 2 < in_len) {
    const uint32_t v = (uint32_t(in[ip]) << 16) | (uint32_t(in[ip + 1]) << 8) | in[ip + 2];
    const uint32_t hs = ((v * 2654435761u) >> (32 - HLOG));
    const uint32_t ref = htab[hs];
    htab[hs] = uint32_t(ip);
    if (ref != 0xFFFFFFFFu && ip - ref <= MAX_OFF && ip > ref && in[ref] == in[ip] && in[ref + 1] == in[ip + 1] &&
        in[ref + 2] == in[ip + 2]) {
      size_t len = 3;
      const size_t max_len = (in_len - ip) < MAX_REF ? (in_len - ip) : MAX_REF;
      while (len < max_len && in[ref + len] == in[ip + len]) ++len;
      if (!flush_literals(ip)) return 0;
      const size_t off = ip - ref - 1, l = len - 2;
      if (op + 3 > out_len) return 0;
      if (l < 7) {
        out[op++] = (unsigned char)((off >> 8) + (l << 5));
      } else {
        out[op++] = (unsigned char)((off >> 8) + (7u << 5));
        out[op++] = (unsigned char)(l - 7);
      }
      out[op++] = (unsigned char)(off & 0xff);
      ip += len;
      lit_start = ip;
    } else {
      ++ip;
    }
  }
  if (!flush_literals(in_len)) return 0;
  return op;
}

// The file mapped read-only: binary bodies are used in place (no copy of the file image), and a header
// query touches only the first pages.
struct MappedFile {
  const char* data = nullptr;
  size_t size = 0;
  int fd = -1;
  ~MappedFile() {
    if (data && size) ::munmap(const_cast<char*>(data), size);
    if (fd >= 0) ::close(fd);
  }
  std::string_view view() const { return std::string_view(data ? data : "", size); }
};

bool read_file(const char* path, MappedFile& mf, std::string& err) {
  mf.fd = ::open(path, O_RDONLY);
  if (mf.fd < 0) {
    err = std::string("cannot open ") + path + ": " + std::strerror(errno);
    return false;
  }
  struct stat st;
  if (::fstat(mf.fd, &st) != 0) {
    err = std::string("cannot stat ") + path + ": " + std::strerror(errno);
    return false;
  }
  mf.size = size_t(st.st_size);
  if (mf.size == 0) return true;
  void* p = ::mmap(nullptr, mf.size, PROT_READ, MAP_PRIVATE, mf.fd, 0);
  if (p == MAP_FAILED) {
    err = std::string("cannot map ") + path + ": " + std::strerror(errno);
    mf.size = 0;
    return false;
  }
  mf.data = static_cast<const char*>(p);
  return true;
}

int find_field(const Header& h, const char* name) {
  for (size_t i = 0; i < h.fields.size(); ++i)
    if (h.fields[i].name == name) return int(i);
  return -1;
}

// ---- small thread pool for the body loops (no OpenMP runtime dependency in the shipped library) ------
template <class F>
void parallel_for(uint64_t n, uint64_t grain, F&& body) {  // body(begin, end)
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  uint64_t nt = (n + grain - 1) / (grain ? grain : 1);
  if (nt > hw) nt = hw;
  if (nt > 8) nt = 8;  // first-touch page faults of a fresh destination buffer stop scaling beyond this
  if (nt <= 1) {
    body(uint64_t(0), n);
    return;
  }
  std::vector<std::thread> th;
  const uint64_t chunk = (n + nt - 1) / nt;
  for (uint64_t t = 0; t < nt; ++t) {
    const uint64_t lo = t * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    if (lo >= hi) break;
    th.emplace_back([&body, lo, hi] { body(lo, hi); });
  }
  for (auto& x : th) x.join();
}

// element `c` of a field stored at q, as double
inline double load_scalar(const unsigned char* q, int size, char type) {
  switch (type) {
    case 'F':
      if (size == 4) { float v; std::memcpy(&v, q, 4); return v; }
      if (size == 8) { double v; std::memcpy(&v, q, 8); return v; }
      break;
    case 'I':
      if (size == 1) { int8_t v; std::memcpy(&v, q, 1); return v; }
      if (size == 2) { int16_t v; std::memcpy(&v, q, 2); return v; }
      if (size == 4) { int32_t v; std::memcpy(&v, q, 4); return v; }
      if (size == 8) { int64_t v; std::memcpy(&v, q, 8); return double(v); }
      break;
    default:
      if (size == 1) { uint8_t v; std::memcpy(&v, q, 1); return v; }
      if (size == 2) { uint16_t v; std::memcpy(&v, q, 2); return v; }
      if (size == 4) { uint32_t v; std::memcpy(&v, q, 4); return v; }
      if (size == 8) { uint64_t v; std::memcpy(&v, q, 8); return double(v); }
  }
  return 0.0;
}

// copyStringValue (io/include/pcl/io/pcd_io.h): "nan" -> quiet NaN and the cloud is not dense
inline void store_ascii_value(unsigned char* q, int size, char type, const char* tok, bool& dense) {
  if (type == 'F') {
    double v;
    if (!std::strcmp(tok, "nan") || !std::strcmp(tok, "-nan") || !std::strcmp(tok, "NaN")) {
      v = std::nan("");
      dense = false;
    } else {
      v = std::strtod(tok, nullptr);
      if (!std::isfinite(v)) dense = false;
    }
    if (size == 4) { const float x = float(v); std::memcpy(q, &x, 4); }
    else if (size == 8) std::memcpy(q, &v, 8);
  } else if (type == 'I') {
    const long long v = std::strtoll(tok, nullptr, 10);
    if (size == 1) { const int8_t x = int8_t(v); std::memcpy(q, &x, 1); }
    else if (size == 2) { const int16_t x = int16_t(v); std::memcpy(q, &x, 2); }
    else if (size == 4) { const int32_t x = int32_t(v); std::memcpy(q, &x, 4); }
    else if (size == 8) { const int64_t x = int64_t(v); std::memcpy(q, &x, 8); }
  } else {
    const unsigned long long v = std::strtoull(tok, nullptr, 10);
    if (size == 1) { const uint8_t x = uint8_t(v); std::memcpy(q, &x, 1); }
    else if (size == 2) { const uint16_t x = uint16_t(v); std::memcpy(q, &x, 2); }
    else if (size == 4) { const uint32_t x = uint32_t(v); std::memcpy(q, &x, 4); }
    else if (size == 8) { const uint64_t x = uint64_t(v); std::memcpy(q, &x, 8); }
  }
}

// The decoded body as per-field views: element c of field f of point i is at
// base[f] + i * stride[f] + c * size.  binary: straight into the file image (array of structs);
// binary_compressed: into the LZF-decoded struct of arrays; ascii: into a parsed array of structs.
struct Body {
  std::vector<const unsigned char*> base;
  std::vector<size_t> stride;
  std::vector<unsigned char> storage;
  bool dense = true;
};

bool decode_body(std::string_view bytes, const Header& h, Body& body, std::string& err) {
  const size_t nf = h.fields.size();
  body.base.assign(nf, nullptr);
  body.stride.assign(nf, 0);
  body.dense = true;
  const uint64_t n = h.points;
  if (n == 0) return true;
  const unsigned char* file = reinterpret_cast<const unsigned char*>(bytes.data());
  if (h.data_type == 0) {  // readBodyASCII :456-559
    size_t elems = 0;
    for (const Field& f : h.fields) elems += size_t(f.count);
    // every ascii point takes at least one byte per element plus a line break
    if (n > bytes.size() || n * (elems > 0 ? elems : 1) > 2 * bytes.size()) {
      err = "file is shorter than the POINTS it announces";
      return false;
    }
    body.storage.assign(size_t(n) * h.point_step, 0);
    // the i-th non-empty line is point i (a malformed line still consumes its point, :489-495)
    std::vector<std::pair<size_t, size_t>> lines;  // [begin, end)
    lines.reserve(size_t(n));
    size_t pos = size_t(h.data_offset);
    while (lines.size() < n && pos < bytes.size()) {
      const void* nl = std::memchr(bytes.data() + pos, '\n', bytes.size() - pos);
      const size_t eol = nl ? size_t(static_cast<const char*>(nl) - bytes.data()) : bytes.size();
      if (eol > pos) lines.emplace_back(pos, eol);
      pos = eol + 1;
    }
    if (lines.size() != n) {
      err = "Number of points read is different than expected";
      return false;
    }
    std::vector<char> dense_flags(64, 1);
    std::atomic<unsigned> slot{0};
    parallel_for(n, 1 << 14, [&](uint64_t lo, uint64_t hi) {
      bool dense = true;
      std::vector<std::pair<const char*, size_t>> tok;
      tok.reserve(elems + 1);
      char buf[128];
      for (uint64_t i = lo; i < hi; ++i) {
        tok.clear();
        const char* p = bytes.data() + lines[size_t(i)].first;
        const char* const e = bytes.data() + lines[size_t(i)].second;
        while (p < e) {
          while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
          const char* q = p;
          while (q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
          if (q > p) tok.emplace_back(p, size_t(q - p));
          p = q;
        }
        if (tok.size() != elems) continue;  // malformed: the point keeps its zeros
        unsigned char* rec = &body.storage[size_t(i) * h.point_step];
        size_t total = 0;
        for (const Field& f : h.fields) {
          if (f.name != "_")
            for (int c = 0; c < f.count; ++c) {
              const auto& t = tok[total + size_t(c)];
              const size_t len = t.second < sizeof(buf) - 1 ? t.second : sizeof(buf) - 1;
              std::memcpy(buf, t.first, len);
              buf[len] = 0;
              store_ascii_value(rec + f.offset + size_t(c) * size_t(f.size), f.size, f.type, buf, dense);
            }
          total += size_t(f.count);
        }
      }
      if (!dense) dense_flags[slot.fetch_add(1) % dense_flags.size()] = 0;
    });
    for (char d : dense_flags)
      if (!d) body.dense = false;
    for (size_t f = 0; f < nf; ++f) {
      body.base[f] = body.storage.data() + h.fields[f].offset;
      body.stride[f] = h.point_step;
    }
    return true;  // is_dense of ascii files comes from the parse (copyStringValue), not from a second scan
  }
  if (h.data_type == 1) {  // binary: array of structs, used in place
    if (h.data_offset + size_t(n) * h.point_step > bytes.size()) {
      err = "file is shorter than WIDTH x HEIGHT x point size";
      return false;
    }
    for (size_t f = 0; f < nf; ++f) {
      body.base[f] = file + h.data_offset + h.fields[f].offset;
      body.stride[f] = h.point_step;
    }
  } else {  // binary_compressed :568-629
    if (h.data_offset + 8 > bytes.size()) {
      err = "truncated binary_compressed header";
      return false;
    }
    uint32_t csize = 0, usize = 0;
    std::memcpy(&csize, file + h.data_offset, 4);
    std::memcpy(&usize, file + h.data_offset + 4, 4);
    if (h.data_offset + 8 + csize > bytes.size()) {
      err = "truncated binary_compressed body";
      return false;
    }
    // An LZF stream grows by at most 88x (a 3-byte back reference yields up to 264 bytes): a size word beyond that is the
    // decompressor's "does not match" failure, known before 4 GB are allocated and cleared for a 500-byte file (the
    // reference resizes first, :576-580; same outcome for every file it accepts)
    if (uint64_t(usize) > uint64_t(csize) * 88u) {
      err = "Size of decompressed lzf data does not match value stored in PCD header";
      return false;
    }
    body.storage.resize(usize);
    const size_t got = usize ? lzf_decompress(file + h.data_offset + 8, csize, body.storage.data(), usize) : 0;
    if (got != usize) {
      err = "Size of decompressed lzf data does not match value stored in PCD header";
      return false;
    }
    // struct of arrays: all elements of field 0, then field 1, ... ("_" padding fields are not stored, :590-600)
    size_t toff = 0;
    for (size_t f = 0; f < nf; ++f) {
      const Field& fd = h.fields[f];
      if (fd.name == "_") continue;
      const size_t fs = size_t(fd.count) * size_t(fd.size);
      body.base[f] = body.storage.data() + toff;
      body.stride[f] = fs;
      toff += fs * size_t(n);
    }
    if (toff > usize) {
      err = "compressed stream smaller than the fields it should hold";
      return false;
    }
  }
  // is_dense: any non-finite value of any floating field (:634-672)
  std::atomic<int> nonfinite{0};
  for (size_t f = 0; f < nf && !nonfinite.load(); ++f) {
    const Field& fd = h.fields[f];
    if (fd.type != 'F' || fd.name == "_" || !body.base[f]) continue;
    parallel_for(n, 1 << 18, [&](uint64_t lo, uint64_t hi) {
      bool bad = false;
      for (uint64_t i = lo; i < hi && !bad; ++i)
        for (int c = 0; c < fd.count; ++c)
          if (!std::isfinite(load_scalar(body.base[f] + size_t(i) * body.stride[f] + size_t(c) * size_t(fd.size), fd.size, fd.type)))
            bad = true;
      if (bad) nonfinite.store(1);
    });
  }
  body.dense = nonfinite.load() == 0;
  return true;
}

void fill_info(const Header& h, pclhip_pcd_info* info) {
  std::memset(info, 0, sizeof *info);
  info->points = h.points;
  info->width = h.width;
  info->height = h.height;
  info->data_type = h.data_type;
  info->version = h.version;
  info->point_step = uint32_t(h.point_step);
  info->num_fields = uint32_t(h.fields.size());
  info->has_xyz = find_field(h, "x") >= 0 && find_field(h, "y") >= 0 && find_field(h, "z") >= 0;
  info->has_normals = find_field(h, "normal_x") >= 0 && find_field(h, "normal_y") >= 0 && find_field(h, "normal_z") >= 0;
  info->has_curvature = find_field(h, "curvature") >= 0;
  info->has_intensity = find_field(h, "intensity") >= 0;
  info->has_rgb = find_field(h, "rgb") >= 0 || find_field(h, "rgba") >= 0;
  std::memcpy(info->viewpoint, h.viewpoint, sizeof h.viewpoint);
  info->data_offset = h.data_offset;
}

pclhip_status fail(const std::string& msg) {
  set_error(nullptr, msg);
  return PCLHIP_ERR_INVALID;
}

}  // namespace

extern "C" {

pclhip_status pclhip_pcd_read_header(const char* path, pclhip_pcd_info* info) {
  if (!path || !info) return PCLHIP_ERR_INVALID;
  MappedFile mapped;
  std::string err;
  if (!read_file(path, mapped, err)) return fail(err);
  const std::string_view bytes = mapped.view();
  Header h;
  if (!parse_header(bytes, h, err)) return fail("[pcl::PCDReader::readHeader] " + err);
  fill_info(h, info);
  return PCLHIP_OK;
}

pclhip_status pclhip_pcd_read(const char* path, void* points, size_t stride, size_t normals_offset, uint64_t capacity,
                              uint64_t* n_out, int* is_dense) {
  if (!path || !n_out) return PCLHIP_ERR_INVALID;
  MappedFile mapped;
  std::string err;
  if (!read_file(path, mapped, err)) return fail(err);
  const std::string_view bytes = mapped.view();
  Header h;
  if (!parse_header(bytes, h, err)) return fail("[pcl::PCDReader::readHeader] " + err);
  *n_out = h.points;
  const int ix = find_field(h, "x"), iy = find_field(h, "y"), iz = find_field(h, "z");
  if (ix < 0 || iy < 0 || iz < 0) return fail("the file has no x/y/z fields");
  if (h.points == 0) {
    if (is_dense) *is_dense = 1;
    return PCLHIP_OK;
  }
  if (!points) return PCLHIP_ERR_INVALID;
  if (capacity < h.points) {
    set_error(nullptr, "output buffer too small for the cloud (see *n_out)");
    return PCLHIP_ERR_OVERFLOW;
  }
  if (stride < 12 || stride % 4 != 0) return fail("stride must be a multiple of 4 and >= 12 bytes");
  const int inx = find_field(h, "normal_x"), iny = find_field(h, "normal_y"), inz = find_field(h, "normal_z");
  const int icv = find_field(h, "curvature");
  const bool want_n = normals_offset != 0 && inx >= 0 && iny >= 0 && inz >= 0;
  if (normals_offset != 0 && (normals_offset % 4 != 0 || normals_offset + 12 > stride))
    return fail("normals_offset must be a multiple of 4 with room for 3 floats inside the record");
  Body body;
  if (!decode_body(bytes, h, body, err)) return fail("[pcl::PCDReader::read] " + err);
  if (is_dense) *is_dense = body.dense ? 1 : 0;
  // assemble the records on the host (a staging buffer when the destination is device memory)
  const bool dev = is_device_pointer(points);
  std::vector<unsigned char> staging;
  unsigned char* dst = static_cast<unsigned char*>(points);
  if (dev) {
    staging.assign(size_t(h.points) * stride, 0);
    dst = staging.data();
  }
  struct Src {
    const unsigned char* base = nullptr;
    size_t stride = 0;
    int size = 4;
    char type = 'F';
  };
  auto src_of = [&](int fi) {
    Src r;
    if (fi >= 0) {
      r.base = body.base[size_t(fi)];
      r.stride = body.stride[size_t(fi)];
      r.size = h.fields[size_t(fi)].size;
      r.type = h.fields[size_t(fi)].type;
    }
    return r;
  };
  const Src sx = src_of(ix), sy = src_of(iy), sz = src_of(iz);
  const Src snx = src_of(want_n ? inx : -1), sny = src_of(want_n ? iny : -1), snz = src_of(want_n ? inz : -1);
  const Src scv = src_of((want_n && icv >= 0 && normals_offset + 20 <= stride) ? icv : -1);
  const bool pad_w = stride >= 16 && !(normals_offset != 0 && normals_offset < 16);  // PointXYZ padding (data[3] = 1)
  const bool pad_n = want_n && normals_offset + 16 <= stride;
  auto get = [](const Src& s, uint64_t i) -> float {
    const unsigned char* q = s.base + size_t(i) * s.stride;
    if (s.type == 'F' && s.size == 4) {  // the common case: a plain copy
      float v;
      std::memcpy(&v, q, 4);
      return v;
    }
    return float(load_scalar(q, s.size, s.type));
  };
  parallel_for(h.points, 1 << 16, [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      float* o = reinterpret_cast<float*>(dst + size_t(i) * stride);
      o[0] = get(sx, i);
      o[1] = get(sy, i);
      o[2] = get(sz, i);
      if (pad_w) o[3] = 1.0f;
      if (want_n) {
        float* nn = reinterpret_cast<float*>(dst + size_t(i) * stride + normals_offset);
        nn[0] = get(snx, i);
        nn[1] = get(sny, i);
        nn[2] = get(snz, i);
        if (pad_n) nn[3] = 0.0f;
        if (scv.base) nn[4] = get(scv, i);
      }
    }
  });
  if (dev) {
    if (hipMemcpy(points, staging.data(), staging.size(), hipMemcpyHostToDevice) != hipSuccess) {
      set_error(nullptr, "hipMemcpy to the device buffer failed");
      return PCLHIP_ERR_HIP;
    }
  }
  return PCLHIP_OK;
}

static pclhip_status pcd_write_impl(const char* path, const void* points, size_t stride, size_t normals_offset,
                                    uint64_t n, uint32_t width, uint32_t height, const float* viewpoint, int data_type,
                                    int precision) {
  if (!path || (n > 0 && !points)) return PCLHIP_ERR_INVALID;
  if (stride < 12 || stride % 4 != 0) return fail("stride must be a multiple of 4 and >= 12 bytes");
  if (data_type < 0 || data_type > 2) return fail("data_type: 0 ascii, 1 binary, 2 binary_compressed");
  const bool with_n = normals_offset != 0;
  if (with_n && (normals_offset % 4 != 0 || normals_offset + 12 > stride)) return fail("bad normals_offset");
  const bool with_curv = with_n && normals_offset + 20 <= stride;
  if (n > 0xFFFFFFFFull) return fail("too many points for a PCD header");
  // gather the records on the host
  std::vector<unsigned char> host;
  const unsigned char* src = static_cast<const unsigned char*>(points);
  if (n > 0 && is_device_pointer(points)) {
    host.resize(size_t(n) * stride);
    if (hipMemcpy(host.data(), points, host.size(), hipMemcpyDeviceToHost) != hipSuccess) {
      set_error(nullptr, "hipMemcpy from the device buffer failed");
      return PCLHIP_ERR_HIP;
    }
    src = host.data();
  }
  const int nf = 3 + (with_n ? 3 : 0) + (with_curv ? 1 : 0);
  const size_t fsize = size_t(nf) * 4;
  // generateHeaderBinary / ASCII (io/src/pcd_io.cpp:848-1043): v0.7 header, unorganized cloud
  std::ostringstream hd;
  hd << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z";
  if (with_n) hd << " normal_x normal_y normal_z";
  if (with_curv) hd << " curvature";
  hd << "\nSIZE";
  for (int i = 0; i < nf; ++i) hd << " 4";
  hd << "\nTYPE";
  for (int i = 0; i < nf; ++i) hd << " F";
  hd << "\nCOUNT";
  for (int i = 0; i < nf; ++i) hd << " 1";
  hd << "\nWIDTH " << width << "\nHEIGHT " << height << "\nVIEWPOINT";
  static const float identity_pose[7] = {0, 0, 0, 1, 0, 0, 0};
  const float* vp = viewpoint ? viewpoint : identity_pose;
  for (int i = 0; i < 7; ++i) hd << ' ' << vp[i];
  hd << "\nPOINTS " << n << "\nDATA "
     << (data_type == 0 ? "ascii" : (data_type == 1 ? "binary" : "binary_compressed")) << "\n";
  const std::string header = hd.str();
  auto value = [&](uint64_t i, int f) -> float {
    const float* o = reinterpret_cast<const float*>(src + size_t(i) * stride);
    if (f < 3) return o[f];
    const float* nn = reinterpret_cast<const float*>(src + size_t(i) * stride + normals_offset);
    return f < 6 ? nn[f - 3] : nn[4];
  };
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return fail(std::string("cannot create ") + path + ": " + std::strerror(errno));
  bool ok = std::fwrite(header.data(), 1, header.size(), f) == header.size();
  if (data_type == 0) {  // writeASCII: values separated by one blank, `precision` significant digits, "nan"
    // (the stream inserter of the reference formats like printf("%.{precision}g") in the classic locale)
    const int prec = precision > 0 ? precision : 8;
    const uint64_t nchunk = n ? (n + (1u << 15) - 1) >> 15 : 0;
    std::vector<std::string> parts;
    parts.resize(static_cast<size_t>(nchunk));
    parallel_for(nchunk, 1, [&](uint64_t clo, uint64_t chi) {
      char buf[64];
      for (uint64_t c = clo; c < chi; ++c) {
        std::string& out = parts[size_t(c)];
        const uint64_t lo = c << 15, hi = (lo + (1u << 15) < n) ? lo + (1u << 15) : n;
        out.reserve(size_t(hi - lo) * size_t(nf) * 12);
        for (uint64_t i = lo; i < hi; ++i) {
          for (int k = 0; k < nf; ++k) {
            const float v = value(i, k);
            if (k) out.push_back(' ');
            if (std::isnan(v)) {
              out.append("nan");
            } else {
              const int len = std::snprintf(buf, sizeof buf, "%.*g", prec, double(v));
              out.append(buf, size_t(len > 0 ? len : 0));
            }
          }
          out.push_back('\n');
        }
      }
    });
    for (const std::string& part : parts) ok = ok && std::fwrite(part.data(), 1, part.size(), f) == part.size();
  } else if (data_type == 1) {  // writeBinary: packed array of structs
    std::vector<float> body(size_t(n) * size_t(nf));
    parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi) {
      for (uint64_t i = lo; i < hi; ++i)
        for (int k = 0; k < nf; ++k) body[size_t(i) * size_t(nf) + size_t(k)] = value(i, k);
    });
    ok = ok && (body.empty() || std::fwrite(body.data(), 4, body.size(), f) == body.size());
  } else {  // writeBinaryCompressed: struct of arrays, LZF, sizes in front
    std::vector<float> soa(size_t(n) * size_t(nf));
    parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi) {
      for (int k = 0; k < nf; ++k)
        for (uint64_t i = lo; i < hi; ++i) soa[size_t(k) * size_t(n) + size_t(i)] = value(i, k);
    });
    const size_t usize = soa.size() * 4;
    if (usize > 0xFFFFFFFFull) {
      std::fclose(f);
      return fail("cloud too large for binary_compressed (32-bit sizes)");
    }
    std::vector<unsigned char> comp(usize + usize / 16 + 64);
    const size_t csize = usize ? lzf_compress(reinterpret_cast<const unsigned char*>(soa.data()), usize, comp.data(), comp.size()) : 0;
    if (usize && csize == 0) {
      std::fclose(f);
      return fail("LZF compression failed");
    }
    const uint32_t c32 = uint32_t(csize), u32 = uint32_t(usize);
    ok = ok && std::fwrite(&c32, 4, 1, f) == 1 && std::fwrite(&u32, 4, 1, f) == 1;
    ok = ok && (csize == 0 || std::fwrite(comp.data(), 1, csize, f) == csize);
  }
  (void)fsize;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) return fail(std::string("write failed on ") + path);
  return PCLHIP_OK;
}

pclhip_status pclhip_pcd_write(const char* path, const void* points, size_t stride, size_t normals_offset, uint64_t n,
                               int data_type, int precision) {
  if (n > 0xFFFFFFFFull) return fail("too many points for a PCD header");
  return pcd_write_impl(path, points, stride, normals_offset, n, uint32_t(n), 1, nullptr, data_type, precision);
}

pclhip_status pclhip_pcd_write_organized(const char* path, const void* points, size_t stride, size_t normals_offset,
                                         uint32_t width, uint32_t height, const float viewpoint[7], int data_type,
                                         int precision) {
  return pcd_write_impl(path, points, stride, normals_offset, uint64_t(width) * uint64_t(height), width, height, viewpoint,
                        data_type, precision);
}

pclhip_status pclhip_pcd_read_field(const char* path, const char* field, uint32_t component, float* out, uint64_t capacity,
                                    uint64_t* n_out) {
  if (!path || !field || !n_out) return PCLHIP_ERR_INVALID;
  MappedFile mapped;
  std::string err;
  if (!read_file(path, mapped, err)) return fail(err);
  const std::string_view bytes = mapped.view();
  Header h;
  if (!parse_header(bytes, h, err)) return fail("[pcl::PCDReader::readHeader] " + err);
  *n_out = h.points;
  const int fi = find_field(h, field);
  if (fi < 0) return fail(std::string("the file has no field named ") + field);
  const Field& fd = h.fields[size_t(fi)];
  if (component >= uint32_t(fd.count)) return fail("component index beyond the field's COUNT");
  if (h.points == 0) return PCLHIP_OK;
  if (!out) return PCLHIP_ERR_INVALID;
  if (capacity < h.points) {
    set_error(nullptr, "output buffer too small for the cloud (see *n_out)");
    return PCLHIP_ERR_OVERFLOW;
  }
  if (is_device_pointer(out)) return fail("pclhip_pcd_read_field writes host memory");
  Body body;
  if (!decode_body(bytes, h, body, err)) return fail("[pcl::PCDReader::read] " + err);
  const unsigned char* base = body.base[size_t(fi)] + size_t(component) * size_t(fd.size);
  const size_t st = body.stride[size_t(fi)];
  const bool raw32 = fd.size == 4 && (fd.type == 'F' || fd.name == "rgb" || fd.name == "rgba");
  parallel_for(h.points, 1 << 16, [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      const unsigned char* q = base + size_t(i) * st;
      if (raw32)
        std::memcpy(&out[i], q, 4);  // float32 as is; packed rgb / rgba keep their 32 bits
      else
        out[i] = float(load_scalar(q, fd.size, fd.type));
    }
  });
  return PCLHIP_OK;
}

}  // extern "C"
