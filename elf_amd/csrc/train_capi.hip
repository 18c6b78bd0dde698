// C ABI of the training-side replay loader (include/elf_amd.h, elftrain_*): record store in HBM + one-launch batch extraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <random>
#include <vector>

#include "engine_host.h"
#include "train.cuh"

struct ElfReplay {
  ElfGoEngine* eng = nullptr;
  ReplayStore st{};
  int P = 0;
  std::vector<int32_t> h_num_moves;     // host mirror for elftrain_draw
  std::vector<int32_t> filled;          // slots that hold a record, in first-put order
  std::vector<uint8_t> is_filled;
  std::mt19937 rng;                     // GoGameBase::_rng of the sampling thread (game_base.h:32-38)
  std::vector<int32_t> h_draw;
  // records put since the last extraction: their checkpoints are written by ONE launch at the head of the next elftrain_extract
  std::vector<int32_t> dirty;
  std::vector<uint8_t> is_dirty;
  int32_t* d_dirty = nullptr;           // [capacity]
  int keep_states = 1;                  // elftrain_set_keep_states
  // page-locked inputs of elftrain_put_async are copied into a small ring of the library's own page-locked buffers and the DMA reads
  // THOSE: the caller's buffer is free when the call returns without a wait for the stream (which may hold extractions queued before)
  static constexpr int kStage = 8;
  char* stage[kStage] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_done[kStage] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes = 0;
  int stage_next = 0;
};

static bool host_is_pinned(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc memory: unknown to the runtime
  return at.type == hipMemoryTypeHost;
}

__global__ void k_replay_put_scalars(ReplayStore st, int slot, int32_t nm, int32_t np, int32_t nv, float w, long long black_ver) {
  st.num_moves[slot] = nm; st.num_pol[slot] = np; st.num_values[slot] = nv; st.winner[slot] = w; st.black_ver[slot] = black_ver;
}

extern "C" {

int elftrain_create(ElfGoEngine* e, int capacity, int max_moves, int with_policies, uint32_t seed, ElfReplay** out) {
  if (!e || !out || capacity <= 0 || max_moves <= 0 || max_moves > 65535) return ELFGO_E_BADARG;
  ElfReplay* r = new (std::nothrow) ElfReplay();
  if (!r) return ELFGO_E_NOMEM;
  r->eng = e;
  r->P = (e->n + 2) * (e->n + 2);
  ReplayStore& st = r->st;
  st.capacity = capacity; st.max_moves = max_moves;
  DevGuard _dg(e->device);
  const size_t cm = (size_t)capacity * max_moves;
#define A(ptr, bytes) do { hipError_t _e = hipMalloc((void**)&(ptr), (bytes)); if (_e != hipSuccess) { elftrain_destroy(r); return (int)_e; } \
                           _e = hipMemset((ptr), 0, (bytes)); if (_e != hipSuccess) { elftrain_destroy(r); return (int)_e; } } while (0)
  A(st.moves, cm * sizeof(u16)); A(st.num_moves, sizeof(int32_t) * capacity); A(st.winner, sizeof(float) * capacity);
  A(st.black_ver, sizeof(int64_t) * capacity); A(st.num_pol, sizeof(int32_t) * capacity);
  A(st.values, cm * sizeof(float)); A(st.num_values, sizeof(int32_t) * capacity);
  if (with_policies) A(st.pol, cm * (size_t)r->P);
  st.nck = max_moves / CK_INTERVAL;
  // the checkpoint slots and the games' superko records (~160 KB per 19x19 record) belong to the checkpointed mode
  // (elftrain_set_keep_states(0)): they are allocated when that mode is first switched on, not for callers that never use it
  A(r->d_dirty, sizeof(int32_t) * capacity);
#undef A
  r->is_dirty.assign(capacity, 0);
  r->h_num_moves.assign(capacity, 0);
  r->is_filled.assign(capacity, 0);
  r->rng.seed(seed);
  *out = r;
  return 0;
}

int elftrain_destroy(ElfReplay* r) {
  if (!r) return ELFGO_E_BADARG;
  DevGuard _dg(r->eng->device);
  void* ptrs[] = {r->st.moves, r->st.num_moves, r->st.winner, r->st.black_ver, r->st.pol, r->st.num_pol, r->st.values, r->st.num_values,
                  r->st.ckpt, r->st.skrec, r->d_dirty};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (int i = 0; i < ElfReplay::kStage; ++i) {
    if (r->stage_done[i]) { (void)hipEventSynchronize(r->stage_done[i]); (void)hipEventDestroy(r->stage_done[i]); }
    if (r->stage[i]) (void)hipHostFree(r->stage[i]);
  }
  delete r;
  return 0;
}

int elftrain_capacity(const ElfReplay* r) { return r ? r->st.capacity : ELFGO_E_BADARG; }
int elftrain_max_moves(const ElfReplay* r) { return r ? r->st.max_moves : ELFGO_E_BADARG; }
int elftrain_num_records(const ElfReplay* r) { return r ? (int)r->filled.size() : ELFGO_E_BADARG; }

// Stream-ordered: the copies are queued on `stream` (host buffers may be reused on return: pageable memory is staged by the
// runtime before the call returns, and page-locked buffers are copied into the library's staging ring first), so a put that reuses the slot of an evicted record cannot overtake an extraction queued on
// the same stream that still reads it.
int elftrain_put_async(ElfReplay* r, int slot, const uint16_t* moves_host, int num_moves, float reward, int64_t black_ver,
                       const uint8_t* policies_host, int num_policies, const float* values_host, int num_values, void* stream) {
  if (!r || slot < 0 || slot >= r->st.capacity || num_moves < 0 || num_moves > r->st.max_moves) return ELFGO_E_BADARG;
  if ((num_moves > 0 && !moves_host) || num_policies < 0 || num_policies > r->st.max_moves || num_values < 0 ||
      num_values > r->st.max_moves) return ELFGO_E_BADARG;
  if ((num_policies > 0 && (!policies_host || !r->st.pol)) || (num_values > 0 && !values_host)) return ELFGO_E_BADARG;
  DevGuard _dg(r->eng->device);
  ReplayStore& st = r->st;
  hipStream_t s = (hipStream_t)stream;
  const size_t base = (size_t)slot * st.max_moves;
  // "host buffers may be reused on return": pageable memory is staged by the runtime before the copy call returns; a page-locked or
  // registered buffer would be read by the DMA engine later, so its bytes go through the library's own staging ring first (one
  // memcpy on the host; an entry is reused after its event, i.e. after kStage later puts -- no wait for the stream's earlier work)
  const size_t nb_m = sizeof(u16) * (size_t)num_moves, nb_p = (size_t)num_policies * r->P, nb_v = sizeof(float) * (size_t)num_values;
  const bool pinned = host_is_pinned(moves_host) || host_is_pinned(policies_host) || host_is_pinned(values_host);
  const void *src_m = moves_host, *src_p = policies_host, *src_v = values_host;
  int se = -1;
  if (pinned) {
    if (!r->stage_bytes) r->stage_bytes = ((size_t)st.max_moves * (sizeof(u16) + (size_t)r->P + sizeof(float)) + 255) & ~(size_t)255;
    se = r->stage_next;
    r->stage_next = (r->stage_next + 1) % ElfReplay::kStage;
    if (!r->stage[se]) {
      HIPCHK(hipHostMalloc((void**)&r->stage[se], r->stage_bytes, hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&r->stage_done[se], hipEventDisableTiming));
    } else {
      HIPCHK(hipEventSynchronize(r->stage_done[se]));      // the copies of the put that used this entry kStage puts ago
    }
    char* b = r->stage[se];
    if (nb_m) { memcpy(b, moves_host, nb_m); src_m = b; }
    b += (size_t)st.max_moves * sizeof(u16);
    if (nb_p) { memcpy(b, policies_host, nb_p); src_p = b; }
    b += (size_t)st.max_moves * r->P;
    if (nb_v) { memcpy(b, values_host, nb_v); src_v = b; }
  }
  if (num_moves) HIPCHK(hipMemcpyAsync(st.moves + base, src_m, nb_m, hipMemcpyHostToDevice, s));
  if (num_policies) HIPCHK(hipMemcpyAsync(st.pol + base * r->P, src_p, nb_p, hipMemcpyHostToDevice, s));
  if (num_values) HIPCHK(hipMemcpyAsync(st.values + base, src_v, nb_v, hipMemcpyHostToDevice, s));
  if (se >= 0) HIPCHK(hipEventRecord(r->stage_done[se], s));
  const float w = reward > 0 ? 1.0f : -1.0f;   // fromRecord, go_state_ext.h:250
  // the record's scalars travel as kernel arguments (copied at launch), not as asynchronous copies from this function's stack
  hipLaunchKernelGGL(k_replay_put_scalars, dim3(1), dim3(1), 0, s, st, slot, (int32_t)num_moves, (int32_t)num_policies, (int32_t)num_values, w,
                     (long long)black_ver);
  HIPCHK(hipGetLastError());
  r->h_num_moves[slot] = num_moves;
  if (!r->is_filled[slot]) { r->is_filled[slot] = 1; r->filled.push_back(slot); }
  if (!r->is_dirty[slot]) { r->is_dirty[slot] = 1; r->dirty.push_back(slot); }   // checkpoints: at the head of the next extraction
  return 0;
}

int elftrain_put(ElfReplay* r, int slot, const uint16_t* moves_host, int num_moves, float reward, int64_t black_ver,
                 const uint8_t* policies_host, int num_policies, const float* values_host, int num_values) {
  const int rc = elftrain_put_async(r, slot, moves_host, num_moves, reward, black_ver, policies_host, num_policies, values_host, num_values, nullptr);
  if (rc) return rc;
  DevGuard _dg(r->eng->device);
  HIPCHK(hipStreamSynchronize(nullptr));
  return 0;
}

int elftrain_set_keep_states(ElfReplay* r, int on) {
  if (!r) return ELFGO_E_BADARG;
  if (!on && !r->st.skrec) {
    // first use of the checkpointed mode: the stores are allocated now; every record put so far is still on the dirty list (the
    // checkpoint launch is skipped while keep_states is on), so the next extraction writes their checkpoints
    DevGuard _dg(r->eng->device);
    ReplayStore& st = r->st;
    const size_t skw = r->eng->n == 19 ? Geo<19>::SKW : Geo<9>::SKW;
    const size_t ck_bytes = (size_t)st.capacity * st.nck * r->eng->slot_bytes;
    const size_t sk_bytes = (size_t)st.capacity * (st.max_moves + 2) * skw * sizeof(u64);
    if (st.nck > 0) {
      hipError_t e = hipMalloc((void**)&st.ckpt, ck_bytes);
      if (e != hipSuccess) { st.ckpt = nullptr; return (int)e; }
      HIPCHK(hipMemset(st.ckpt, 0, ck_bytes));
    }
    hipError_t e = hipMalloc((void**)&st.skrec, sk_bytes);
    if (e != hipSuccess) { st.skrec = nullptr; if (st.ckpt) { (void)hipFree(st.ckpt); st.ckpt = nullptr; } return (int)e; }
    HIPCHK(hipMemset(st.skrec, 0, sk_bytes));
  }
  r->keep_states = on != 0;
  return 0;
}

// GoGameTrain::act :26-40: sample a record, switchRandomMove (move_to = rng() % (size - nfa + 1), records with
// size <= nfa - 1 are rejected and resampled, go_state_ext.h:260-275), generateD4Code (rng() % 8, :277-279).
// The record itself is drawn uniformly from the filled slots with the same stream (the reference's ReaderQueues
// sampler belongs to the replay-buffer control plane).
int elftrain_draw(ElfReplay* r, int n, int num_future_actions, int32_t* rec_dev, int32_t* move_to_dev, int32_t* d4_dev, void* stream) {
  if (!r || n <= 0 || num_future_actions < 1 || !rec_dev || !move_to_dev || !d4_dev) return ELFGO_E_BADARG;
  bool any = false;
  for (int32_t s : r->filled) any = any || r->h_num_moves[s] > num_future_actions - 1;
  if (!any) return ELFGO_E_BADARG;
  DevGuard _dg(r->eng->device);
  r->h_draw.resize((size_t)3 * n);
  for (int i = 0; i < n; ++i) {
    int32_t slot;
    do { slot = r->filled[r->rng() % r->filled.size()]; } while (r->h_num_moves[slot] <= num_future_actions - 1);
    r->h_draw[i] = slot;
    r->h_draw[n + i] = (int32_t)(r->rng() % (uint32_t)(r->h_num_moves[slot] - num_future_actions + 1));
    r->h_draw[2 * n + i] = (int32_t)(r->rng() % 8);
  }
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(rec_dev, r->h_draw.data(), 4 * (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(move_to_dev, r->h_draw.data() + n, 4 * (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d4_dev, r->h_draw.data() + 2 * n, 4 * (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));   // h_draw is reused by the next call
  return 0;
}

int elftrain_extract(ElfReplay* r, const int32_t* rec, const int32_t* move_to, const int32_t* d4, int n, const ElfTrainBatch* b, void* stream) {
  if (!r || !rec || !move_to || !b || n < 0 || (r->keep_states && n > r->eng->capacity)) return ELFGO_E_BADARG;
  if (n == 0) return 0;
  DevGuard _dg(r->eng->device);
  const int nn = r->eng->n;
  if (!b->s || b->s_stride < (int64_t)18 * nn * nn || (b->s_format != ELFGO_FEAT_F32_NCHW && b->s_format != ELFGO_FEAT_F16_NHWC)) return ELFGO_E_BADARG;
  if (b->offline_a && b->num_future_actions < 1) return ELFGO_E_BADARG;
  TrainBatch o;
  o.s = b->s; o.s_stride = b->s_stride; o.fmt = b->s_format;
  o.offline_a = b->offline_a; o.nfa = b->num_future_actions;
  o.winner = b->winner; o.mcts_scores = b->mcts_scores; o.predicted_value = b->predicted_value;
  o.move_idx = b->move_idx; o.num_move = b->num_move; o.aug_code = b->aug_code; o.selfplay_ver = b->selfplay_ver;
  hipStream_t s = (hipStream_t)stream;
  if (!r->dirty.empty() && !r->keep_states) {
    // the records put since the last extraction get their checkpoints and superko records now, ahead of the samples that use them
    // (with keep_states on nothing reads them: the slots stay on the dirty list for a later switch of the mode)
    const int nd = (int)r->dirty.size();
    HIPCHK(hipMemcpyAsync(r->d_dirty, r->dirty.data(), sizeof(int32_t) * nd, hipMemcpyHostToDevice, s));
    DISPATCH(r->eng, hipLaunchKernelGGL((k_replay_checkpoint<N, Pool<N>>), dim3((nd + REPLAY_WAVES_CK - 1) / REPLAY_WAVES_CK), dim3(64 * REPLAY_WAVES_CK), 0, s,
                                        pool_of<N>(r->eng), r->st, r->d_dirty, nd));
    HIPCHK(hipGetLastError());
    for (int32_t sl : r->dirty) r->is_dirty[sl] = 0;
    r->dirty.clear();
  }
  const dim3 grid((n + REPLAY_WAVES - 1) / REPLAY_WAVES), block(64 * REPLAY_WAVES);
  if (r->keep_states) {
    DISPATCH(r->eng, hipLaunchKernelGGL((k_replay_extract<N, Pool<N>, true>), grid, block, 0, s, pool_of<N>(r->eng), r->st, rec, move_to, d4, n, o));
  } else {
    DISPATCH(r->eng, hipLaunchKernelGGL((k_replay_extract<N, Pool<N>, false>), grid, block, 0, s, pool_of<N>(r->eng), r->st, rec, move_to, d4, n, o));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
