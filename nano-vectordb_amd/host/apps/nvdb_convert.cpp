// nvdb_convert_f16 / nvdb_quantize_i8 in one source (argv[0] or --i8 picks the mode):
//   nvdb_convert_f16 <in_f32.vecbin> <out_f16.vecbin>      fp32 -> IEEE half, round-to-nearest-even
//   nvdb_quantize_i8 <in.vecbin>     <out_i8.vecbin>       per-row scale = max|x|/127, scales after payload
// Output files are byte-identical to the reference tools' (tools/nvdb_convert_f16.cpp,
// apps/nvdb_quantize_i8.cpp) for dim % 8 == 0; see DESIGN.md section 6 for the dim % 8 != 0 tail.
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "nvdb/vector_dataset.h"
#include "nvdb_hip.h"

int main(int argc, char** argv) {
  const std::string self = argv[0];
  bool i8 = self.find("quantize_i8") != std::string::npos;
  int a = 1;
  if (argc > 1 && std::string(argv[1]) == "--i8") { i8 = true; a = 2; }
  if (argc < a + 2) { std::cerr << "Usage: " << (i8 ? "nvdb_quantize_i8 <in.vecbin> <out_i8.vecbin>" : "nvdb_convert_f16 <input_f32.vecbin> <output_f16.vecbin>") << "\n"; return 1; }
  nvdb::VectorDataset ds;
  ds.load(argv[a]);
  const uint64_t n = ds.count();
  const uint32_t d = ds.dim();
  if (!i8 && ds.dtype() != 1) { std::cerr << "Input dtype must be Float32. Got dtype=" << ds.dtype() << "\n"; return 2; }
  if (i8 && ds.dtype() != 1 && ds.dtype() != 2) { std::cerr << "Input must be f32/f16. dtype=" << ds.dtype() << "\n"; return 2; }
  nvdb::VecbinHeader h;
  std::memset(&h, 0, sizeof(h));
  h.magic = nvdb::kMagic; h.version = nvdb::kVersion; h.dtype = i8 ? 3 : 2; h.dim = d; h.count = n;
  std::ofstream out(argv[a + 1], std::ios::binary);
  if (!out) { std::cerr << "Failed to open output: " << argv[a + 1] << "\n"; return 4; }
  out.write(reinterpret_cast<const char*>(&h), sizeof(h));
  const uint64_t slab = 4096;
  std::vector<float> f(slab * d), scales(i8 ? n : 0);
  std::vector<uint16_t> h16(i8 ? 0 : slab * d);
  std::vector<int8_t> q8(i8 ? slab * d : 0);
  for (uint64_t r0 = 0; r0 < n; r0 += slab) {
    const uint64_t take = std::min(slab, n - r0);
    if (ds.dtype() == 1) std::memcpy(f.data(), ds.vector_ptr_f32(r0), take * d * sizeof(float));
    else for (uint64_t i = 0; i < take * d; ++i) { const uint16_t v = ds.vector_ptr_f16(r0)[i]; uint32_t s = (v & 0x8000u) << 16, e = (v >> 10) & 31, m = v & 1023, o;
        if (e == 0) { if (!m) o = s; else { int x = -14; while (!(m & 1024)) { m <<= 1; --x; } o = s | (uint32_t(x + 127) << 23) | ((m & 1023) << 13); } }
        else if (e == 31) o = s | 0x7F800000u | (m << 13); else o = s | ((e + 112) << 23) | (m << 13);
        std::memcpy(&f[i], &o, 4); }
    if (i8) { nvdb_quantize_i8_rows(f.data(), take, d, q8.data(), scales.data() + r0); out.write(reinterpret_cast<const char*>(q8.data()), static_cast<std::streamsize>(take * d)); }
    else { nvdb_f32_to_f16(f.data(), h16.data(), take * d); out.write(reinterpret_cast<const char*>(h16.data()), static_cast<std::streamsize>(take * d * 2)); }
  }
  if (i8) out.write(reinterpret_cast<const char*>(scales.data()), static_cast<std::streamsize>(n * sizeof(float)));
  std::cerr << "Wrote " << (i8 ? "int8" : "FP16") << " vecbin64: " << argv[a + 1] << " count=" << n << " dim=" << d << "\n";
  return out ? 0 : 5;
}
