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
  const size_t stride = 5888;                        // bytes between records (a small tree-node record, mcts.cuh)
  FILE* js = fopen("gpurun_out/chase.json", "w");    // tools/profile_all.sh copies it to profiles/<tag>_chase.json; bench.py reads it
  if (js) fprintf(js, "{\"stride_bytes\": %zu, \"what\": \"ns per dependent 256-B load (one level of a descent): every wave chases its own random cycle through records spread over the footprint\", \"rows\": [", stride);
  bool first_row = true;
  for (double gb : {0.05, 8.0, 64.0, 160.0}) {
    for (int waves : {256, 2048, 4608}) {
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
      if (js) { fprintf(js, "%s{\"footprint_GB\": %.2f, \"waves\": %d, \"ns_per_hop\": %.1f}", first_row ? "" : ", ", gb, waves, dt / hops * 1e9); first_row = false; }
      hipFree(d); hipFree(out);
    }
  }
  if (js) { fprintf(js, "]}\n"); fclose(js); }
  return 0;
}
