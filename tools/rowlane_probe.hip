// GPU-box tool (not part of the library): the row-per-lane bitboard layout VERDICT r4/r5 asked to prototype for k_playout, against the
// library's own action-order layout, on the part of a board step it would change most: the two dilations and the true-eye test of
// Board::legal_moves<true> (base/board.cc:201-240, 1850-1914).
//   layout A (library):  bit a = x * N + y of a 361-bit string, 64 bits per lane in lanes 0..5 (a second operand rides in lanes 8..13)
//   layout B (probe):    lane x holds row x as a 19-bit mask; a-1 / a+1 are 32-bit shifts, a-N / a+N one DPP wave shift each
// Both kernels run the same positions; the legal-candidate and eye sets must be equal bit for bit (checked on the host), then a timed
// loop of dependent iterations gives ns per (dilate2 + eye test) per wave at 1, 4 and 8 waves per SIMD.  The static VALU counts of the
// two loop bodies come from the disassembly (profiles/r06_rowlane_probe.txt).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off tools/rowlane_probe.hip -o /tmp/rowlane && /tmp/rowlane
#include <hip/hip_runtime.h>
#include "../elf_amd/csrc/elf_amd.hip"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

constexpr int NB = 19;
// ---- layout A: the library's code, verbatim calls
__global__ __launch_bounds__(64) void k_layout_a(const u64* zob, const u64* in /* [boards][4][8]: E, Own, Opp, At words */, u64* out /* [boards][2][8] */, int iters) {
  __shared__ Slot<NB> lds;
  const int lane = threadIdx.x;
  Board<NB> bd;
  bd.init(&lds, zob, nullptr);
  const u64* p = in + (size_t)blockIdx.x * 32;
  u64 E = lane < 8 ? p[lane] : 0, Own = lane < 8 ? p[8 + lane] : 0, Opp = lane < 8 ? p[16 + lane] : 0, At = lane < 8 ? p[24 + lane] : 0;
  u64 okw = 0, eyew = 0;
  for (int it = 0; it < iters; ++it) {
    u64 d1w, d2w;
    bd.dilate2(E | (Own & ~At) | (Opp & At), E | Opp, d1w, d2w);
    okw = E & d1w;
    const u64 allown = okw & ~d2w;
    eyew = 0;
    if (bal_ne64(allown, 0ull)) {
      const u64 pk = bd.sh_m1(Opp) | dpp_u64<0x118>(bd.sh_p1(Opp));
      const u64 d1 = bd.sh_mN(pk), d3 = bd.sh_pN(pk);
      const u64 d2 = dpp_u64<0x108>(d1), d4 = dpp_u64<0x108>(d3);
      const u64 ge1 = d1 | d2 | d3 | d4;
      const u64 ge2 = (d1 & d2) | (d3 & d4) | ((d1 | d2) & (d3 | d4));
      const u64 fake = (bd.mEdge & ge1) | (~bd.mEdge & ge2);
      eyew = allown & ~fake;
    }
    if (it + 1 < iters) At ^= (okw ^ eyew) & Own;   // the next iteration depends on this one (a chain, as in the board step)
  }
  if (lane < 8) { out[(size_t)blockIdx.x * 16 + lane] = lane < Geo<NB>::R ? okw : 0; out[(size_t)blockIdx.x * 16 + 8 + lane] = lane < Geo<NB>::R ? eyew : 0; }
}

// ---- layout B: one row per lane
__device__ __forceinline__ u32 up1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true); }    // wave_shr:1: lane x <- lane x-1
__device__ __forceinline__ u32 down1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true); }  // wave_shl:1: lane x <- lane x+1
__global__ __launch_bounds__(64) void k_layout_b(const u32* in /* [boards][4][32] rows */, u32* out /* [boards][2][32] */, int iters) {
  const int lane = threadIdx.x;
  constexpr u32 M = (1u << NB) - 1;
  const u32 rowmask = lane < NB ? M : 0u;
  const u32 edge = (lane == 0 || lane == NB - 1) ? M : (lane < NB ? (1u | (1u << (NB - 1))) : 0u);
  const u32* p = in + (size_t)blockIdx.x * 128;
  u32 E = lane < 32 ? p[lane] : 0, Own = lane < 32 ? p[32 + lane] : 0, Opp = lane < 32 ? p[64 + lane] : 0, At = lane < 32 ? p[96 + lane] : 0;
  u32 okw = 0, eyew = 0;
  for (int it = 0; it < iters; ++it) {
    const u32 A = E | (Own & ~At) | (Opp & At), B = E | Opp;
    const u32 dA = ((A << 1) | (A >> 1) | up1(A) | down1(A)) & rowmask;
    const u32 dB = ((B << 1) | (B >> 1) | up1(B) | down1(B)) & rowmask;
    okw = E & dA;
    const u32 allown = okw & ~dB;
    eyew = 0;
    if (__ballot(allown != 0) != 0) {
      const u32 U = up1(Opp), D = down1(Opp);
      const u32 d1 = (U << 1) & M, d2 = U >> 1, d3 = (D << 1) & M, d4 = D >> 1;
      const u32 ge1 = d1 | d2 | d3 | d4;
      const u32 ge2 = (d1 & d2) | (d3 & d4) | ((d1 | d2) & (d3 | d4));
      const u32 fake = (edge & ge1) | (~edge & ge2);
      eyew = allown & ~fake;
    }
    if (it + 1 < iters) At ^= (okw ^ eyew) & Own;
  }
  if (lane < 32) { out[(size_t)blockIdx.x * 64 + lane] = okw; out[(size_t)blockIdx.x * 64 + 32 + lane] = eyew; }
}

int main() {
  std::vector<uint64_t> z(441);
  FILE* f = fopen("elf_amd/data/zobrist21.bin", "rb");
  if (!f || fread(z.data(), 8, 441, f) != 441) { fprintf(stderr, "zobrist21.bin?\n"); return 1; }
  fclose(f);
  ElfGoEngine* e = nullptr;
  if (elfgo_create(NB, 64, 0, z.data(), &e)) return 2;
  const int boards = 8192 * 4;
  std::mt19937 rng(7);
  std::vector<uint64_t> ina((size_t)boards * 32, 0);
  std::vector<uint32_t> inb((size_t)boards * 128, 0);
  for (int b = 0; b < boards; ++b) {
    const unsigned dens = 20 + rng() % 70;           // stones per hundred points: sparse to crowded (eyes appear when crowded)
    for (int x = 0; x < NB; ++x)
      for (int y = 0; y < NB; ++y) {
        const unsigned r = rng() % 100;
        int k = r >= dens ? 0 : ((rng() % 100) < 60 ? 1 : 2);   // 0 empty, 1 own, 2 opponent
        const bool at = k != 0 && rng() % 5 == 0;
        const int a = x * NB + y;
        const int w[4] = {k == 0, k == 1, k == 2, at};
        for (int q = 0; q < 4; ++q)
          if (w[q]) { ina[(size_t)b * 32 + q * 8 + (a >> 6)] |= 1ull << (a & 63); inb[(size_t)b * 128 + q * 32 + x] |= 1u << y; }
      }
  }
  u64 *dina, *douta; u32 *dinb, *doutb;
  hipMalloc((void**)&dina, ina.size() * 8); hipMalloc((void**)&douta, (size_t)boards * 16 * 8);
  hipMalloc((void**)&dinb, inb.size() * 4); hipMalloc((void**)&doutb, (size_t)boards * 64 * 4);
  hipMemcpy(dina, ina.data(), ina.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dinb, inb.data(), inb.size() * 4, hipMemcpyHostToDevice);
  // ---- equality of the two formulations on every position (one iteration)
  k_layout_a<<<boards, 64>>>(e->zob, dina, douta, 1);
  k_layout_b<<<boards, 64>>>(dinb, doutb, 1);
  hipDeviceSynchronize();
  std::vector<uint64_t> oa((size_t)boards * 16); std::vector<uint32_t> ob((size_t)boards * 64);
  hipMemcpy(oa.data(), douta, oa.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(ob.data(), doutb, ob.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0, eyes = 0, cands = 0;
  for (int b = 0; b < boards; ++b)
    for (int x = 0; x < NB; ++x)
      for (int y = 0; y < NB; ++y) {
        const int a = x * NB + y;
        for (int q = 0; q < 2; ++q) {
          const int va = (oa[(size_t)b * 16 + q * 8 + (a >> 6)] >> (a & 63)) & 1, vb = (ob[(size_t)b * 64 + q * 32 + x] >> y) & 1;
          bad += va != vb;
          if (q == 0) cands += va; else eyes += va;
        }
      }
  printf("equality: %d positions, %ld legal candidates, %ld true eyes, %ld differing bits\n", boards, cands, eyes, bad);
  // ---- timed: dependent iterations, `waves` waves in flight
  for (int waves : {1024, 4096, 8192, 32768}) {
    const int iters = 4000;
    double t[2];
    for (int v = 0; v < 2; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (v == 0) k_layout_a<<<waves, 64>>>(e->zob, dina, douta, iters); else k_layout_b<<<waves, 64>>>(dinb, doutb, iters);
        hipDeviceSynchronize();
        t[v] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
    }
    printf("%5d waves (%.1f per SIMD): layout A %.1f ns, layout B %.1f ns per dilate2 + eye test and wave; A/B = %.2f\n", waves, waves / 1024.0,
           t[0] / iters * 1e9, t[1] / iters * 1e9, t[0] / t[1]);
  }
  return bad ? 3 : 0;
}
