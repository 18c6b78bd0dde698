/* Plain-C consumer of libelf_amd.so (no HIP headers, no torch): config-2 playouts of `boards` 19x19 games on one MI355X.
 *   gcc -std=c11 -O2 -I include examples/board_playout.c -o board_playout -L elf_amd/lib -lelf_amd -Wl,-rpath,$PWD/elf_amd/lib
 *   ./board_playout [boards] [zobrist21.bin]
 * Prints "<board> <hash hex> <ply> <steps>" for the first boards and the rate; tests/test_gpu_board.py compares the lines with
 * the golden playouts the real reference produced (tests/golden/playout_19.npz). */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "elf_amd.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, elfgo_error_string(rc_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int boards = argc > 1 ? atoi(argv[1]) : 4096;
  const char* zpath = argc > 2 ? argv[2] : "elf_amd/data/zobrist21.bin";
  uint64_t zob[441];
  FILE* f = fopen(zpath, "rb");
  if (!f || fread(zob, sizeof(uint64_t), 441, f) != 441) { fprintf(stderr, "cannot read %s\n", zpath); return 1; }
  fclose(f);
  ElfGoEngine* e = NULL;
  CHECK(elfgo_create(19, boards, 0, zob, &e));
  uint64_t* seeds = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)boards);
  uint32_t* out = (uint32_t*)malloc(sizeof(uint32_t) * 4 * (size_t)boards);
  for (int b = 0; b < boards; ++b) seeds[b] = (uint64_t)b * 0x9E3779B9ull + 1;   /* s_b = 0x9E3779B9*b + 1 */
  void *d_seeds = NULL, *d_out = NULL;
  CHECK(elfgo_malloc(&d_seeds, sizeof(uint64_t) * (size_t)boards));
  CHECK(elfgo_malloc(&d_out, sizeof(uint32_t) * 4 * (size_t)boards));
  CHECK(elfgo_memcpy_h2d(d_seeds, seeds, sizeof(uint64_t) * (size_t)boards));
  double best = 0.0;
  unsigned long long steps = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(elfgo_reset(e, NULL, boards, NULL));
    CHECK(elfgo_sync(e, NULL));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    CHECK(elfgo_playout(e, NULL, (const uint64_t*)d_seeds, boards, 1 << 20, (uint32_t*)d_out, NULL));
    CHECK(elfgo_sync(e, NULL));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    CHECK(elfgo_memcpy_d2h(out, d_out, sizeof(uint32_t) * 4 * (size_t)boards));
    steps = 0;
    for (int b = 0; b < boards; ++b) steps += out[4 * b + 3];
    const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if ((double)steps / dt > best) best = (double)steps / dt;
  }
  for (int b = 0; b < boards && b < 8; ++b)
    printf("%d %08x%08x %u %u\n", b, out[4 * b + 1], out[4 * b], out[4 * b + 2], out[4 * b + 3]);
  printf("boards %d steps %llu rate %.1f M board steps/s\n", boards, steps, best / 1e6);
  CHECK(elfgo_free(d_seeds));
  CHECK(elfgo_free(d_out));
  CHECK(elfgo_destroy(e));
  free(seeds);
  free(out);
  return 0;
}
