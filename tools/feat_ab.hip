// GPU-box tool (not part of the library): times k_extract_agz of the library's own translation unit, without phase markers.
// A/B method: an experiment adds an `#ifdef ELF_AB_<name>` switch to the kernel source for its duration (never committed), one
// binary per switch is cross-compiled here and all of them run in one short GPU visit:
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off [-DELF_AB_x] tools/feat_ab.hip -o build/ab_x
#include <hip/hip_runtime.h>
#include "../elf_amd/csrc/elf_amd.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 16384;
  const int fmt = argc > 2 ? atoi(argv[2]) : 0;
  std::vector<uint64_t> z(441);
  FILE* f = fopen("elf_amd/data/zobrist21.bin", "rb");
  if (!f || fread(z.data(), 8, 441, f) != 441) { fprintf(stderr, "zobrist21.bin?\n"); return 1; }
  fclose(f);
  ElfGoEngine* e = nullptr;
  if (elfgo_create(19, rows, 0, z.data(), &e)) return 2;
  std::vector<uint64_t> seeds(rows);
  for (int i = 0; i < rows; ++i) seeds[i] = (uint64_t)i * 0x9E3779B9ull + 1;
  uint64_t* dseeds; uint32_t* dout; int32_t* d4; void* dst;
  hipMalloc((void**)&dseeds, 8 * rows); hipMalloc((void**)&dout, 16 * rows); hipMalloc((void**)&d4, 4 * rows);
  hipMalloc(&dst, (size_t)rows * 6498 * 4);
  hipMemcpy(dseeds, seeds.data(), 8 * rows, hipMemcpyHostToDevice);
  std::vector<int32_t> h4(rows);
  for (int i = 0; i < rows; ++i) h4[i] = (i * 7 + 3) & 7;
  hipMemcpy(d4, h4.data(), 4 * rows, hipMemcpyHostToDevice);
  elfgo_playout(e, nullptr, dseeds, rows, 120, dout, nullptr);   // 120 plies of random play: mid-game positions
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) elfgo_extract_agz_fmt(e, nullptr, d4, rows, dst, 6498, fmt, nullptr);
  hipEventRecord(a, nullptr);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) elfgo_extract_agz_fmt(e, nullptr, d4, rows, dst, 6498, fmt, nullptr);
  hipEventRecord(b, nullptr); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double bytes = (double)rows * (6498.0 * (fmt ? 2 : 4) + 736);
  printf("%s: %d rows fmt %d: %.4f ms  %.2f TB/s\n", argv[0], rows, fmt, ms, bytes / ms / 1e9);
  return 0;
}
