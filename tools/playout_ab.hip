// GPU-box tool (not part of the library): times k_playout of the library's own translation unit, without phase markers.
// A/B method: an experiment adds an `#ifdef ELF_AB_<name>` switch to the kernel source for its duration (never committed), one
// binary per switch is cross-compiled here and all of them run in one short GPU visit:
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off [-DELF_AB_x] tools/playout_ab.hip -o build/ab_x
#include <hip/hip_runtime.h>
#include "../elf_amd/csrc/elf_amd.hip"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  const int boards = argc > 1 ? atoi(argv[1]) : 4096;
  const int n = argc > 2 ? atoi(argv[2]) : 19;
  std::vector<uint64_t> z(441);
  FILE* f = fopen("elf_amd/data/zobrist21.bin", "rb");
  if (!f || fread(z.data(), 8, 441, f) != 441) { fprintf(stderr, "zobrist21.bin?\n"); return 1; }
  fclose(f);
  ElfGoEngine* e = nullptr;
  if (elfgo_create(n, boards, 0, z.data(), &e)) return 2;
  std::vector<uint64_t> seeds(boards);
  uint64_t* dseeds; uint32_t* dout;
  hipMalloc((void**)&dseeds, 8 * boards); hipMalloc((void**)&dout, 16 * boards);
  double best = 0;
  unsigned long long steps = 0;
  for (int rep = 0; rep < 5; ++rep) {
    for (int i = 0; i < boards; ++i) seeds[i] = (uint64_t)(i) * 0x9E3779B9ull + 1;
    hipMemcpy(dseeds, seeds.data(), 8 * boards, hipMemcpyHostToDevice);
    elfgo_reset(e, nullptr, boards, nullptr);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    elfgo_playout(e, nullptr, dseeds, boards, 1 << 20, dout, nullptr);
    hipDeviceSynchronize();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<uint32_t> out(4 * boards);
    hipMemcpy(out.data(), dout, 16 * boards, hipMemcpyDeviceToHost);
    steps = 0;
    for (int i = 0; i < boards; ++i) steps += out[4 * i + 3];
    if (rep && steps / dt > best) best = steps / dt;
  }
  printf("%s: %d boards %dx%d, %llu steps, best %.1f M steps/s\n", argv[0], boards, n, n, steps, best / 1e6);
  return 0;
}
