// libelf_amd.so: HIP kernels (gfx950) + the C ABI declared in include/elf_amd.h.
// One wave64 = one board (k_playout and k_extract_agz pack 4 independent waves per workgroup); see go_board.cuh for the device engine.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/elf_amd.h"
#include "go_board.cuh"
#include "engine_host.h"

using namespace elfgo;

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_of(const int32_t* ids, int i) { return ids ? ids[i] : i; }

template <int N>
__global__ __launch_bounds__(WAVE) void k_reset(Pool<N> pool, const int32_t* ids, int n) {
  __shared__ Slot<N> lds;
  int b = slot_of(ids, blockIdx.x);
  Board<N> bd;
  bd.init(&lds, pool.zob, pool.skr(b));
  bd.reset();
  bd.store(&pool.slots[b]);
}

template <int N>
__global__ __launch_bounds__(WAVE) void k_copy(Pool<N> pool, const int32_t* dst, const int32_t* src, int n) {
  using G = Geo<N>;
  int d = dst[blockIdx.x], s = src[blockIdx.x];
  if (d == s) return;
  int lane = threadIdx.x;
  const uint4* sp = reinterpret_cast<const uint4*>(&pool.slots[s]);
  uint4* dp = reinterpret_cast<uint4*>(&pool.slots[d]);
  for (int j = lane; j < (int)(sizeof(Slot<N>) / 16); j += WAVE) dp[j] = sp[j];
  int len = pool.slots[s].h.sk_len;
  const u64* si = pool.skr(s); u64* di = pool.skr(d);
  for (int j = lane; j < len * G::SKW; j += WAVE) di[j] = si[j];
}

template <int N>
__global__ __launch_bounds__(WAVE) void k_forward(Pool<N> pool, const int32_t* ids, const int32_t* moves, int n, uint8_t* ok) {
  __shared__ Slot<N> lds;
  int b = slot_of(ids, blockIdx.x);
  int c = moves[blockIdx.x];
  if (c == M_INVALID || c < 0) {  // go_state.cc:75-77 (reference throws)
    if (threadIdx.x == 0 && ok) ok[blockIdx.x] = 0xFF;
    return;
  }
  Board<N> bd;
  bd.init(&lds, pool.zob, pool.skr(b));
  bd.load(&pool.slots[b]);
  int r = bd.forward(c);
  if (r) bd.store(&pool.slots[b]);
  if (threadIdx.x == 0 && ok) ok[blockIdx.x] = (uint8_t)r;
}

template <int N>
__global__ __launch_bounds__(WAVE) void k_legal_mask(Pool<N> pool, const int32_t* ids, int n, uint8_t* mask) {
  using G = Geo<N>;
  __shared__ Slot<N> lds;
  int b = slot_of(ids, blockIdx.x);
  Board<N> bd;
  bd.init(&lds, pool.zob, pool.skr(b));
  bd.load(&pool.slots[b]);
  u64 legal, cand;
  bd.template legal_moves<false>(legal, cand);
  uint8_t* out = mask + (size_t)blockIdx.x * G::NA;
#pragma unroll
  for (int k = 0; k < G::R; ++k) {
    int a = k * 64 + bd.lane;
    const u64 wk = rl64(legal, k);
    if (a < G::NP) out[a] = (uint8_t)((wk >> bd.lane) & 1);
  }
  if (bd.lane == 0) out[G::NP] = 1;  // pass is always accepted by TryPlay (board.cc:794-800)
}

// BoardFeature::extractAGZ (board_feature.cc:247-290) + Transform (board_feature.h:97-113): wave masks of the 18 planes per
// round of 64 output points, then 1-KiB contiguous stores of the row (go_board.cuh: agz_nhwc_f16 / agz_flat_f32; the staged
// agz_bitplanes / agz_store pair remains for fp16 rows at odd 2-byte offsets).
// AGZ_WAVES rows in flight per workgroup, each wave looping over rows (grid-stride): a 1-wave workgroup per row spends a large
// share of its short life being launched.
#define AGZ_WAVES 4
template <int N>
__global__ __launch_bounds__(WAVE * AGZ_WAVES) void k_extract_agz(Pool<N> pool, const int32_t* ids, const int32_t* d4s, int n,
                                                                   void* dst, int64_t stride, int fmt) {
  using G = Geo<N>;
  __shared__ u64 hist_all[AGZ_WAVES][HIST][2][G::R];
  __shared__ u64 scratch_all[AGZ_WAVES][AGZ_SCRATCH_BYTES / 8];
  const int lane = threadIdx.x & 63, wv = rfl((int)(threadIdx.x >> 6));
  u64 (*hist)[2][G::R] = hist_all[wv];
  u64* tpl = scratch_all[wv];
  const int nw = (int)gridDim.x * AGZ_WAVES;
  for (int r = (int)blockIdx.x * AGZ_WAVES + wv; r < n; r += nw) {
    int b = slot_of(ids, r);
    const Slot<N>* sl = &pool.slots[b];
    const u64* gh = &sl->hist[0][0][0];
    for (int j = lane; j < HIST * 2 * G::R; j += WAVE) (&hist[0][0][0])[j] = gh[j];
    const int cnt = sl->h.hist_cnt, player = sl->h.next_player;
    const int d4 = d4s ? d4s[r] : 0;
    agz_wave_sync();
    char* row = (char*)dst + (size_t)r * stride * (fmt == FEAT_F16_NHWC ? 2 : 4);
    extract_agz_row<N>(hist, tpl, cnt, player, d4, row, fmt, lane);
    agz_wave_sync();
  }
}

template <int N>
__global__ __launch_bounds__(WAVE) void k_evaluate(Pool<N> pool, const int32_t* ids, int n, float komi, float* out) {
  __shared__ Slot<N> lds;
  int b = slot_of(ids, blockIdx.x);
  Board<N> bd;
  bd.init(&lds, pool.zob, pool.skr(b));
  bd.load(&pool.slots[b]);
  float v = bd.evaluate(komi);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

template <int N>
__global__ void k_info(Pool<N> pool, const int32_t* ids, int n, int32_t* out) {
  using G = Geo<N>;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Hdr& h = pool.slots[slot_of(ids, i)].h;
  int32_t* o = out + (size_t)i * ELFGO_INFO_WORDS;
  bool term = (h.last_move[0] == M_PASS && h.last_move[1] == M_PASS) || h.ply >= G::MAXMOVE || h.superko;
  o[0] = h.ply; o[1] = h.next_player; o[2] = h.last_move[0]; o[3] = h.last_move[1];
  o[4] = h.ko_age; o[5] = h.ko_pt; o[6] = h.ko_color; o[7] = h.b_cap; o[8] = h.w_cap;
  o[9] = term; o[10] = h.superko; o[11] = h.hist_cnt < HIST ? h.hist_cnt : HIST; o[12] = h.sk_len;
  o[13] = (int32_t)(u32)h.hash; o[14] = (int32_t)(u32)(h.hash >> 32); o[15] = 0;
}

template <int N>
__global__ __launch_bounds__(WAVE) void k_export(Pool<N> pool, const int32_t* ids, int n, uint8_t* colour, int16_t* libs) {
  using G = Geo<N>;
  const Slot<N>* sl = &pool.slots[slot_of(ids, blockIdx.x)];
  for (int a = threadIdx.x; a < G::NP; a += WAVE) {
    u32 v = sl->pt[Board<N>::a2i(a)];
    colour[(size_t)blockIdx.x * G::NP + a] = v == 0 ? 0 : ((v >> 15) + 1);
    libs[(size_t)blockIdx.x * G::NP + a] = v == 0 ? 0 : (int16_t)sl->libs[v & 0x7FFF];
  }
}

// Whole games inside one launch: position stays in LDS from the first move to the last.
// PLAYOUT_WAVES boards per workgroup, still one wave per board and no synchronisation after the prologue: the waves only
// share a copy of the Zobrist constants in LDS (3.4 KiB at 19x19), so that the capture path reads them with ds_read instead
// of a vector load from global memory (see Board::zob_v).
#define PLAYOUT_WAVES 4
template <int N>
__global__ __launch_bounds__(WAVE * PLAYOUT_WAVES) void k_playout(Pool<N> pool, const int32_t* ids, const u64* seeds, int n,
                                                                   int max_steps, uint32_t* out) {
  using G = Geo<N>;
  __shared__ Slot<N> lds_all[PLAYOUT_WAVES];
  __shared__ u64 zlds[G::P];
  __shared__ u32 mlds[G::NP + 2];   // floor(2^32 / d) for the pick's rng % candidates
  for (int j = threadIdx.x; j < G::P; j += WAVE * PLAYOUT_WAVES) zlds[j] = pool.zob[j];
  for (int j = threadIdx.x; j < G::NP + 2; j += WAVE * PLAYOUT_WAVES) mlds[j] = (u32)pool.zob[G::ZOBW + 4 * G::R + j];
  __syncthreads();
  const int wv = rfl((int)(threadIdx.x >> 6));   // wave-uniform by construction; say so, or the seed and the slot turn into vector values
  const int game = blockIdx.x * PLAYOUT_WAVES + wv;
  if (game >= n) return;
  Slot<N>& lds = lds_all[wv];
  int b = slot_of(ids, game);
  Board<N> bd;
  bd.init(&lds, pool.zob, pool.skr(b));
  bd.load(&pool.slots[b]);
  const GameSK<N> sk{pool.skr(b)};
  const u32 key = playout_key(seeds[game]);
  bd.playout_begin(zlds);
  int steps = 0;
  ELF_PHASE(bd, 7);
  while (steps < max_steps && !bd.terminated()) {
    const u32 x = playout_rng_k(key, (u32)bd.ply);   // depends on the ply only: issued ahead of the legal-move work
    u64 legal, cand;
    bd.template legal_moves<true, true>(legal, cand);
    ELF_PHASE(bd, 0);   // legal mask + true eyes
    // uniformly random candidate: r-th set bit of the lane-distributed candidate bitboard (words in lanes 0..R-1, 0 elsewhere)
    const int cnt = __popcll(cand);
    // inclusive prefix sum over the lanes of row 0 on the DPP network (row_shr with zero fill): lane k < R ends with the
    // number of candidates in words 0..k, lanes R..15 with the total
    int inc = cnt;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
    if (G::R > 4) inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
    const int total = rl(inc, G::R - 1);
    int pick_a = -1;   // action id of the chosen candidate, -1 = pass
    if (total > 0) {
      // rng % total without a runtime division: floor(2^32 / total) from the workgroup's LDS copy of the table behind the Zobrist
      // constants; the estimate is at most one too small
      const u32 q = __umulhi(x, (u32)rfl((int)mlds[total]));
      u32 rr = x - q * (u32)total;
      if (rr >= (u32)total) rr -= (u32)total;
      // the word that holds the r-th candidate: the prefix sums are non-decreasing, so it is the number of words whose
      // inclusive count is <= r; its exclusive count is the rank base
      const int kw = __popc((u32)bal_le((u32)inc, rr) & ((1u << G::R) - 1u));
      const int base = rl(inc - cnt, kw);
      const u64 wk = rl64(cand, kw);
      const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(wk >> 32), __builtin_amdgcn_mbcnt_lo((u32)wk, 0));
      const u64 sel = bal_eq(rank, rr - (u32)base) & wk;
      pick_a = kw * 64 + (int)__builtin_ctzll(sel);
    }
    ELF_PHASE(bd, 1);   // pick the k-th candidate
    // a candidate comes from the legal mask of this very position: forward_legal_action skips TryPlay's re-check
    if (!(pick_a >= 0 ? bd.forward_legal_action(pick_a, sk) : bd.forward(M_PASS, sk))) break;
    ++steps;
  }
  ELF_PHASE_END(bd);
  bd.store(&pool.slots[b]);
  if (bd.lane == 0) {
    u64 h = bd.hash;
    uint32_t* o = out + (size_t)game * 4;
    o[0] = (u32)h; o[1] = (u32)(h >> 32); o[2] = (u32)bd.ply; o[3] = (u32)steps;
  }
}

// ------------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------------
template <int N>
static int create_impl(ElfGoEngine* e, const uint64_t* zob_host) {
  using G = Geo<N>;
  e->slot_bytes = sizeof(Slot<N>);
  HIPCHK(hipMalloc(&e->slots, (size_t)e->capacity * sizeof(Slot<N>)));
  HIPCHK(hipMalloc((void**)&e->sk_rec, (size_t)e->capacity * (G::MAXMOVE + 2) * G::SKW * sizeof(u64)));
  HIPCHK(hipMalloc((void**)&e->zob, (size_t)(G::ZOBW + 4 * G::R + G::NP + 2) * sizeof(u64)));
  // reference Coord order -> internal (transposed) index order, followed by the per-word geometry masks and by the table
  // floor(2^32 / d), d <= N*N, that k_playout's "rng % candidates" uses instead of a runtime division
  std::vector<u64> z(G::ZOBW + 4 * G::R + G::NP + 2, 0);
  for (int d = 1; d <= G::NP + 1; ++d) {
    const u64 m = (1ull << 32) / (u64)d;
    z[G::ZOBW + 4 * G::R + d] = m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m;
  }
  for (int i = 0; i < G::P; ++i) z[i] = zob_host[(i % G::S) * G::S + i / G::S];
  for (int a = 0; a < G::NP; ++a) {
    const int k = a >> 6, y = a % N;
    const u64 bit = 1ull << (a & 63);
    const int x = a / N;
    if (y != 0) z[G::ZOBW + 4 * k + 0] |= bit;
    if (y != N - 1) z[G::ZOBW + 4 * k + 1] |= bit;
    z[G::ZOBW + 4 * k + 2] |= bit;
    if (x == 0 || x == N - 1 || y == 0 || y == N - 1) z[G::ZOBW + 4 * k + 3] |= bit;
  }
  HIPCHK(hipMemcpy(e->zob, z.data(), z.size() * sizeof(u64), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_reset<N>, dim3(e->capacity), dim3(WAVE), 0, 0, pool_of<N>(e), (const int32_t*)nullptr, e->capacity);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

extern "C" {

int elfgo_create(int board_size, int capacity, int device, const uint64_t* zobrist_host, ElfGoEngine** out) {
  if (!out || !zobrist_host || capacity <= 0) return ELFGO_E_BADARG;
  if (board_size != 19 && board_size != 9) return ELFGO_E_BADSIZE;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return ELFGO_E_BADARG;
  DevGuard _dg(device);   // the caller's current device is restored on return
  ElfGoEngine* e = new (std::nothrow) ElfGoEngine();
  if (!e) return ELFGO_E_NOMEM;
  e->n = board_size; e->capacity = capacity; e->device = device;
  int rc = board_size == 19 ? create_impl<19>(e, zobrist_host) : create_impl<9>(e, zobrist_host);
  if (rc) { elfgo_destroy(e); return rc; }
  *out = e;
  return 0;
}

int elfgo_destroy(ElfGoEngine* e) {
  if (!e) return ELFGO_E_BADARG;
  DevGuard _dg(e->device);
  if (e->slots) (void)hipFree(e->slots);
  if (e->sk_rec) (void)hipFree(e->sk_rec);
  if (e->zob) (void)hipFree(e->zob);
  delete e;
  return 0;
}

int elfgo_board_size(const ElfGoEngine* e) { return e ? e->n : ELFGO_E_BADARG; }
int elfgo_capacity(const ElfGoEngine* e) { return e ? e->capacity : ELFGO_E_BADARG; }
size_t elfgo_slot_bytes(const ElfGoEngine* e) { return e ? e->slot_bytes : 0; }

int elfgo_sync(ElfGoEngine* e, void* stream) {
  if (!e) return ELFGO_E_BADARG;
  DevGuard _dg(e->device);
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

#define CHECK_N(e, ids, n)                                             \
  if (!(e) || (n) < 0) return ELFGO_E_BADARG;                          \
  if (!(ids) && (n) > (e)->capacity) return ELFGO_E_BADARG;            \
  if ((n) == 0) return 0;                                              \
  DevGuard _dg((e)->device);

int elfgo_reset(ElfGoEngine* e, const int32_t* ids, int n, void* stream) {
  CHECK_N(e, ids, n);
  DISPATCH(e, hipLaunchKernelGGL(k_reset<N>, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, pool_of<N>(e), ids, n));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_copy(ElfGoEngine* e, const int32_t* dst_ids, const int32_t* src_ids, int n, void* stream) {
  if (!e || n < 0 || !dst_ids || !src_ids) return ELFGO_E_BADARG;
  if (n == 0) return 0;
  DevGuard _dg(e->device);
  DISPATCH(e, hipLaunchKernelGGL(k_copy<N>, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, pool_of<N>(e), dst_ids, src_ids, n));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_forward(ElfGoEngine* e, const int32_t* ids, const int32_t* moves, int n, uint8_t* ok, void* stream) {
  CHECK_N(e, ids, n);
  if (!moves) return ELFGO_E_BADARG;
  DISPATCH(e, hipLaunchKernelGGL(k_forward<N>, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, pool_of<N>(e), ids, moves, n, ok));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_legal_mask(ElfGoEngine* e, const int32_t* ids, int n, uint8_t* mask, void* stream) {
  CHECK_N(e, ids, n);
  if (!mask) return ELFGO_E_BADARG;
  DISPATCH(e, hipLaunchKernelGGL(k_legal_mask<N>, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, pool_of<N>(e), ids, n, mask));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_extract_agz_fmt(ElfGoEngine* e, const int32_t* ids, const int32_t* d4, int n, void* dst, int64_t stride_elems, int fmt,
                          void* stream) {
  CHECK_N(e, ids, n);
  if (!dst || stride_elems < (int64_t)18 * e->n * e->n || (fmt != ELFGO_FEAT_F32_NCHW && fmt != ELFGO_FEAT_F16_NHWC)) return ELFGO_E_BADARG;
  if (((uintptr_t)dst & (fmt == ELFGO_FEAT_F16_NHWC ? 1 : 3)) != 0) return ELFGO_E_BADARG;
  // enough workgroups to fill 256 CUs x 8 waves per SIMD, then rows loop inside the waves
  const int agz_wgs = 4096;
  const int wgs = (n + AGZ_WAVES - 1) / AGZ_WAVES < agz_wgs ? (n + AGZ_WAVES - 1) / AGZ_WAVES : agz_wgs;
  DISPATCH(e, hipLaunchKernelGGL(k_extract_agz<N>, dim3(wgs), dim3(WAVE * AGZ_WAVES), 0, (hipStream_t)stream, pool_of<N>(e), ids, d4, n, dst,
                                 stride_elems, fmt));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_extract_agz(ElfGoEngine* e, const int32_t* ids, const int32_t* d4, int n, float* dst, int64_t stride_floats,
                      void* stream) {
  return elfgo_extract_agz_fmt(e, ids, d4, n, dst, stride_floats, ELFGO_FEAT_F32_NCHW, stream);
}

int elfgo_evaluate(ElfGoEngine* e, const int32_t* ids, int n, float komi, float* out, void* stream) {
  CHECK_N(e, ids, n);
  if (!out) return ELFGO_E_BADARG;
  DISPATCH(e, hipLaunchKernelGGL(k_evaluate<N>, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, pool_of<N>(e), ids, n, komi, out));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_info(ElfGoEngine* e, const int32_t* ids, int n, int32_t* out, void* stream) {
  CHECK_N(e, ids, n);
  if (!out) return ELFGO_E_BADARG;
  DISPATCH(e, hipLaunchKernelGGL(k_info<N>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pool_of<N>(e), ids, n, out));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_export_board(ElfGoEngine* e, const int32_t* ids, int n, uint8_t* colour, int16_t* libs, void* stream) {
  CHECK_N(e, ids, n);
  if (!colour || !libs) return ELFGO_E_BADARG;
  DISPATCH(e, hipLaunchKernelGGL(k_export<N>, dim3(n), dim3(WAVE), 0, (hipStream_t)stream, pool_of<N>(e), ids, n, colour, libs));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_playout(ElfGoEngine* e, const int32_t* ids, const uint64_t* seeds, int n, int max_steps, uint32_t* out,
                  void* stream) {
  CHECK_N(e, ids, n);
  if (!seeds || !out) return ELFGO_E_BADARG;
  DISPATCH(e, hipLaunchKernelGGL(k_playout<N>, dim3((n + PLAYOUT_WAVES - 1) / PLAYOUT_WAVES), dim3(WAVE * PLAYOUT_WAVES), 0, (hipStream_t)stream, pool_of<N>(e), ids,
                                 (const u64*)seeds, n, max_steps, out));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfgo_set_device(int device) { HIPCHK(hipSetDevice(device)); return 0; }
int elfgo_get_device(int* device) { if (!device) return ELFGO_E_BADARG; HIPCHK(hipGetDevice(device)); return 0; }
int elfgo_mem_info(int device, size_t* free_bytes, size_t* total_bytes) {
  if (!free_bytes || !total_bytes) return ELFGO_E_BADARG;
  DevGuard _dg(device);
  HIPCHK(hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}
int elfgo_pointer_kind(const void* p, int* device) {
  if (device) *device = -1;
  if (!p) return 0;
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();   // an address the runtime does not know is pageable host memory, not an error
    return 0;
  }
  if (at.type == hipMemoryTypeDevice) { if (device) *device = at.device; return 2; }
  if (at.type == hipMemoryTypeHost) return 1;
  if (at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeUnified) { if (device) *device = at.device; return 2; }
  return 0;
}
int elfgo_memcpy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, void* stream) {
  if (rows == 0 || width_bytes == 0) return 0;
  if (!dst || !src || dst_pitch < width_bytes || src_pitch < width_bytes) return ELFGO_E_BADARG;
  if (dst_pitch == width_bytes && src_pitch == width_bytes)
    HIPCHK(hipMemcpyAsync(dst, src, width_bytes * rows, hipMemcpyDefault, (hipStream_t)stream));
  else
    HIPCHK(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDefault, (hipStream_t)stream));
  return 0;
}
int elfgo_stream_sync(void* stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return 0; }
int elfgo_malloc(void** dptr, size_t bytes) { HIPCHK(hipMalloc(dptr, bytes)); return 0; }
int elfgo_free(void* dptr) { HIPCHK(hipFree(dptr)); return 0; }
int elfgo_memcpy_h2d(void* dst, const void* src_host, size_t bytes) {
  HIPCHK(hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice));
  return 0;
}
int elfgo_memcpy_d2h(void* dst_host, const void* src, size_t bytes) {
  HIPCHK(hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
const char* elfgo_error_string(int status) {
  if (status == 0) return "ok";
  if (status == ELFGO_E_BADARG) return "elfgo: bad argument";
  if (status == ELFGO_E_BADSIZE) return "elfgo: unsupported board size (19 or 9)";
  if (status == ELFGO_E_NOMEM) return "elfgo: out of host memory";
  if (status == ELFGO_E_NODATA) return "elfgo: no eligible record found (the draw state was left unchanged)";
  if (status <= ELFGO_E_MCTS_BASE && status > ELFGO_E_MCTS_BASE - 32) {
    // ELFGO_E_MCTS_BASE - (OR of ELFMCTS_E_* bits)
    static thread_local char buf[320];
    const int bits = ELFGO_E_MCTS_BASE - status;
    snprintf(buf, sizeof(buf), "elfmcts:%s%s%s%s%s",
             (bits & ELFMCTS_E_POOL) ? " the context's node pool is exhausted (raise nodes_per_game);" : "",
             (bits & ELFMCTS_E_ROOT_HASH) ? " TreeSearch::Root state is not the same as the input state;" : "",
             (bits & ELFMCTS_E_FORWARD) ? " a move could not be applied (illegal move / invalid preload or tree edge);" : "",
             (bits & ELFMCTS_E_RNG) ? " more D4 draws requested than uploaded;" : "",
             (bits & ELFMCTS_E_VERSION) ? " model version of a reply and required version are not consistent;" : "");
    return buf;
  }
  if (status < 0) return "elfgo: unknown status";
  return hipGetErrorString((hipError_t)status);
}
const char* elfgo_version(void) { return "elf_amd 0.3 (gfx950)"; }

}  // extern "C"
