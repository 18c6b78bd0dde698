// GPU-box tool (not part of the library): cycle attribution of k_playout<19> by phase.  Compiles the library's own
// translation unit with phase markers switched on:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off
//   tools/playout_phases.hip -o /tmp/playout_phases && /tmp/playout_phases [boards]
#include <hip/hip_runtime.h>
#define ELF_PROFILE 1
__device__ unsigned long long g_phase[16];
__device__ unsigned long long g_count[16];
// per-wave accumulation in registers, one atomic per phase per wave at the end: the markers cost two s_memtime per phase
#define ELF_PHASE(bd, k)                                                        \
  do {                                                                          \
    unsigned long long _t = __builtin_amdgcn_s_memtime();                       \
    if ((k) == 7) { for (int _i = 0; _i < 16; ++_i) (bd).ph_acc[_i] = 0; }       \
    else (bd).ph_acc[k] += _t - (bd).ph_t;                                      \
    (bd).ph_t = _t;                                                             \
  } while (0)
#define ELF_PHASE_END(bd)                                                       \
  do {                                                                          \
    if ((threadIdx.x & 63) == 0)                                                \
      for (int _i = 0; _i < 16; ++_i) { atomicAdd(&g_phase[_i], (bd).ph_acc[_i]); atomicAdd(&g_count[_i], 1ull); } \
  } while (0)
#include "../elf_amd/csrc/elf_amd.hip"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int boards = argc > 1 ? atoi(argv[1]) : 4096;
  std::vector<uint64_t> z(441);
  FILE* f = fopen("elf_amd/data/zobrist21.bin", "rb");
  if (!f || fread(z.data(), 8, 441, f) != 441) { fprintf(stderr, "zobrist21.bin?\n"); return 1; }
  fclose(f);
  ElfGoEngine* e = nullptr;
  if (elfgo_create(19, boards, 0, z.data(), &e)) return 2;
  std::vector<uint64_t> seeds(boards);
  uint64_t* dseeds; uint32_t* dout;
  hipMalloc((void**)&dseeds, 8 * boards); hipMalloc((void**)&dout, 16 * boards);
  const char* names[16] = {"legal mask + eyes", "pick", "TryPlay", "superko record + bloom", "liberty give-back after capture", "mover liberties", "history/header/superko check", "-",
                           "classify + enemy lib decrement", "capture removal", "placement + merge relabel", "-", "-", "-", "-", "-"};
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < boards; ++i) seeds[i] = (uint64_t)(i + rep * boards) * 0x9E3779B9ull + 1;
    hipMemcpy(dseeds, seeds.data(), 8 * boards, hipMemcpyHostToDevice);
    elfgo_reset(e, nullptr, boards, nullptr);
    unsigned long long zero[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero)); hipMemcpyToSymbol(HIP_SYMBOL(g_count), zero, sizeof(zero));
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    elfgo_playout(e, nullptr, dseeds, boards, 1 << 20, dout, nullptr);
    hipDeviceSynchronize();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long ph[16], cn[16];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph)); hipMemcpyFromSymbol(cn, HIP_SYMBOL(g_count), sizeof(cn));
    std::vector<uint32_t> out(4 * boards);
    hipMemcpy(out.data(), dout, 16 * boards, hipMemcpyDeviceToHost);
    unsigned long long steps = 0, tot = 0;
    for (int i = 0; i < boards; ++i) steps += out[4 * i + 3];
    for (int k = 0; k < 16; ++k) tot += ph[k];
    printf("rep %d: %d boards, %llu steps, %.3f ms (with markers) = %.1f M steps/s\n", rep, boards, steps, dt * 1e3, steps / dt / 1e6);
    for (int k = 0; k < 16; ++k)
      if (ph[k]) printf("  phase %d %-30s %6.2f %%  %8.1f ticks/step (x%llu)\n", k, names[k], 100.0 * ph[k] / tot, (double)ph[k] / (steps ? steps : 1), cn[k]);
  }
  return 0;
}
