// GPU-box tool: dependent-load (pointer chase) latency as a function of the footprint and of the number of concurrent chasing
// waves -- what one level of a tree descent costs when every node lives on its own page of a 100-GB pool.
// hipcc --offload-arch=gfx950 -O3 tools/chase.hip -o /tmp/chase && /tmp/chase
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_chase(const unsigned* next, size_t stride_words, int hops, unsigned* out, int records_per_wave) {
  // one wave per block; every lane reads its own 4 bytes of the record (64 lanes x 4 B = one 256-B piece), lane 0's word is the link
  const int lane = threadIdx.x;
  unsigned cur = (unsigned)blockIdx.x * records_per_wave;
  unsigned acc = 0;
  for (int h = 0; h < hops; ++h) {
    const unsigned v = next[(size_t)cur * stride_words + lane];
    acc += v;
    cur = __builtin_amdgcn_readfirstlane(v);
  }
  if (lane == 0) out[blockIdx.x] = acc + cur;
}
int main() {
  const size_t stride = 12800;                       // bytes between records (a tree node)
  for (double gb : {0.05, 1.0, 8.0, 32.0, 100.0}) {
    for (int waves : {128, 1024}) {
      const size_t nrec = (size_t)(gb * 1e9 / stride);
      const int rpw = (int)(nrec / waves);
      unsigned* d; unsigned* out;
      if (hipMalloc((void**)&d, nrec * stride) != hipSuccess) { printf("%.2f GB: alloc failed\n", gb); continue; }
      hipMalloc((void**)&out, waves * 4);
      // each wave chases a random cycle inside its own slice of records
      std::vector<unsigned> link(nrec);
      srand(1);
      for (int w = 0; w < waves; ++w) {
        std::vector<unsigned> perm(rpw);
        for (int i = 0; i < rpw; ++i) perm[i] = i;
        for (int i = rpw - 1; i > 0; --i) { int j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
        for (int i = 0; i < rpw; ++i) link[(size_t)w * rpw + perm[i]] = (unsigned)((size_t)w * rpw + perm[(i + 1) % rpw]);
      }
      // write the link into word 0 of every record (strided copy)
      hipMemset(d, 0, nrec * stride);
      hipMemcpy2D(d, stride, link.data(), 4, 4, nrec, hipMemcpyHostToDevice);
      const int hops = rpw < 2000 ? rpw : 2000;
      k_chase<<<waves, 64>>>(d, stride / 4, 64, out, rpw);
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      k_chase<<<waves, 64>>>(d, stride / 4, hops, out, rpw);
      hipDeviceSynchronize();
      double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("footprint %6.2f GB, %4d waves, %d hops: %.0f ns per dependent load\n", gb, waves, hops, dt / hops * 1e9);
      hipFree(d); hipFree(out);
    }
  }
  return 0;
}
