// Training-side replay loader on the device: one wave per training sample replays a recorded game up to a
// sampled ply with the board engine of go_board.cuh (position in LDS for the whole replay) and emits every field of the
// reference's "train" batch in one launch.
//
// The reference replays every sample from the empty board (switchBeforeMove: reset + forward x move_to, ~160 forwards per sample).
// The result of that replay is a pure function of (record, move_to), so the store keeps, per record, the state after every
// CK_INTERVAL-th move (written once, by k_replay_checkpoint when the record is put) and the superko records of the whole game;
// a sample then loads the checkpoint below its move_to and forwards at most CK_INTERVAL - 1 moves -- same state, same rows, a
// twentieth of the board steps.  HBM is 288 GB: 45 checkpoints x 3840 B + 724 x 104 B of superko records = 248 KB per 19x19 record.
//
// Replaces, for a batch of n samples, what one reference game thread does per sample:
//   src_cpp/elfgames/go/train/game_train.cc          GoGameTrain::act :23-58
//   src_cpp/elfgames/go/common/go_state_ext.h        GoStateExtOffline::fromRecord :248-258, switchBeforeMove :283-290
//   src_cpp/elfgames/go/common/game_feature.h        extractMoveIdx/NumMove/PredictedValue/AugCode/Winner :73-96,
//                                                    extractStateExtAGZ :103-106, extractMCTSPi :107-127,
//                                                    extractOfflineAction :129-139, extractStateSelfplayVersion :141-145
#pragma once
#include <cstddef>
#include "go_board.cuh"

namespace elfgo {

// Replay store in HBM: `capacity` record slots padded to max_moves plies each (sized for 288 GB: 19x19 records with full
// 441-byte policies are 318 KB, i.e. 100 k games = 32 GB).
struct ReplayStore {
  u16* moves;            // [capacity][max_moves] reference Coords of Record.result.content (sgfstr2coords)
  int32_t* num_moves;    // [capacity] _offline_all_moves.size()
  float* winner;         // [capacity] reward > 0 ? 1 : -1 (go_state_ext.h:250)
  int64_t* black_ver;    // [capacity] Record.request.vers.black_ver
  unsigned char* pol;    // [capacity][max_moves][P] CoordRecord.prob (record.h:180-182), P = (N+2)^2; may be NULL
  int32_t* num_pol;      // [capacity] Record.result.policies.size()
  float* values;         // [capacity][max_moves] Record.result.values
  int32_t* num_values;   // [capacity]
  void* ckpt;            // [capacity][nck] Slot<N>: the state after the first (j + 1) * CK_INTERVAL moves of the record
  u64* skrec;            // [capacity][max_moves + 2][SKW] superko records of the record's whole game (GoState::_board_hashes)
  int capacity, max_moves, nck;
};
constexpr int CK_INTERVAL = 16;   // round 6: 32 -> 16 (a sample forwards 7.5 plies on average instead of 15.5; 173 KB of checkpoints per 19x19 record instead of 88 KB)
#define REPLAY_WAVES_CK 4

// Superko records of a REPLAYED record: every pre-move position of the game is already in the store (k_replay_checkpoint wrote
// them when the record was put), so a replay from a checkpoint stores nothing and consults them only on a Bloom hit --
// same hit rule as go_state.cc:96-111.
template <int N>
struct ReplaySK {
  const u64* rec;
  __device__ __forceinline__ void record(int, u64, u64, u64, int) const {}
  __device__ __forceinline__ bool exact_hit(int sk_len, u64 hash, u64 Bw, u64 Ww, int lane) const {
    return GameSK<N>::scan(rec, sk_len, hash, sk_record_word<N>(hash, Bw, Ww, lane), lane);
  }
};

// One wave per record, once per put (launched by the next elftrain_extract for every record put since the last one): the whole
// game from the empty board (exactly the loop of switchBeforeMove), the superko records into the store, the board slot into a
// checkpoint after every CK_INTERVAL-th move.
template <int N, class PoolT>
__global__ __launch_bounds__(64 * REPLAY_WAVES_CK) void k_replay_checkpoint(PoolT pool, ReplayStore st, const int32_t* slots, int n) {
  using G = Geo<N>;
  __shared__ Slot<N> lds_all[REPLAY_WAVES_CK];
  __shared__ u64 zlds[G::P];
  for (int j = threadIdx.x; j < G::P; j += 64 * REPLAY_WAVES_CK) zlds[j] = pool.zob[j];
  __syncthreads();
  const int wv = rfl((int)(threadIdx.x >> 6));
  const int i = blockIdx.x * REPLAY_WAVES_CK + wv;
  if (i >= n) return;
  const int r = rfl(slots[i]);
  const int nm = rfl(st.num_moves[r]);
  const u16* mv = st.moves + (size_t)r * st.max_moves;
  Board<N> bd;
  bd.init(&lds_all[wv], pool.zob, st.skrec + (size_t)r * (st.max_moves + 2) * G::SKW);
  bd.reset();
  bd.playout_begin(zlds);
  Slot<N>* ck = reinterpret_cast<Slot<N>*>(st.ckpt) + (size_t)r * st.nck;
  for (int t = 0; t < nm; ++t) {
    const int c = rfl((int)mv[t]);
    if (c != M_INVALID) bd.forward(c);           // a refused move is skipped like any other (see k_replay_extract)
    if ((t + 1) % CK_INTERVAL == 0 && (t + 1) / CK_INTERVAL <= st.nck) bd.store(&ck[(t + 1) / CK_INTERVAL - 1]);
  }
}

struct TrainBatch {   // device pointers; any of them except `s` may be NULL
  void* s; int64_t s_stride; int fmt;
  int64_t* offline_a; int nfa;
  float* winner; float* mcts_scores; float* predicted_value;
  int32_t* move_idx; int32_t* num_move; int32_t* aug_code;
  int64_t* selfplay_ver;
};

// BoardFeature::coord2Action (board_feature.h:132-137) with Transform (:97-113); same integer arithmetic for any Coord
template <int N>
__device__ __forceinline__ int coord_to_action(int c, int d4) {
  constexpr int S = N + 2;
  if (c == M_PASS) return N * N;
  int x = c % S - 1, y = c / S - 1;
  const int rot = d4 & 3;
  int ox = x, oy = y;
  if (rot == 1) { ox = y; oy = N - x - 1; }
  else if (rot == 2) { ox = N - x - 1; oy = N - y - 1; }
  else if (rot == 3) { ox = N - y - 1; oy = x; }
  if ((d4 >> 2) == 1) { int t = ox; ox = oy; oy = t; }
  return ox * N + oy;
}

// REPLAY_WAVES samples per workgroup, one wave each and no synchronisation after the prologue: the waves share the LDS copy of
// the Zobrist constants (3.4 KiB at 19x19), which lifts the LDS-bound occupancy from 16 to 20 resident waves per CU.  The kernel
// is a dependent latency chain per wave (a replay of ~160 plies), so what it needs is resident waves: elftrain_extract takes
// any number of samples per launch (several train batches at once: the trainer prefetches).
#define REPLAY_WAVES 4
// KEEP = true: the reference's own procedure (from the empty board, superko records of sample i into board slot i's record area,
// the replayed GoState stored in board slot i of the engine, where elfgo_* can look at it).  KEEP = false (the trainer's mode):
// from the record's checkpoint, nothing but the batch is written.
template <int N, class PoolT, bool KEEP>
__global__ __launch_bounds__(64 * REPLAY_WAVES) void k_replay_extract(PoolT pool, ReplayStore st, const int32_t* rec, const int32_t* move_to,
                                                                       const int32_t* d4s, int n, TrainBatch o) {
  using G = Geo<N>;
  // 19x19: the extraction's scratch reuses the front of the wave's own slot (see below); the 9x9 slot is smaller than the scratch
  constexpr bool ALIAS = offsetof(Slot<N>, bloom) >= (size_t)AGZ_SCRATCH_BYTES && sizeof(Slot<N>) - offsetof(Slot<N>, bloom) >= sizeof(u64) * HIST * 2 * G::R;
  __shared__ Slot<N> lds_all[REPLAY_WAVES];
  __shared__ u64 tpl_all[ALIAS ? 1 : REPLAY_WAVES][ALIAS ? 1 : AGZ_SCRATCH_BYTES / 8];
  __shared__ u64 zlds[G::P];   // Zobrist constants: forward reads them from LDS, not behind its own superko record stores (Board::zob_v)
  for (int j = threadIdx.x; j < G::P; j += 64 * REPLAY_WAVES) zlds[j] = pool.zob[j];
  __syncthreads();
  const int wv = rfl((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int i = blockIdx.x * REPLAY_WAVES + wv;
  if (i >= n) return;
  Slot<N>& lds = lds_all[wv];
  int r = rfl(rec[i]);
  r = r < 0 ? 0 : (r >= st.capacity ? st.capacity - 1 : r);   // an out-of-range record id must not read outside the store
  const int d4 = rfl(d4s ? d4s[i] : 0) & 7;
  const int nm = rfl(st.num_moves[r]);
  int mt = rfl(move_to[i]);
  if (mt > nm) mt = nm;   // the reference asserts move_to < size (go_state_ext.h:284)
  const u16* mv = st.moves + (size_t)r * st.max_moves;
  Board<N> bd;
  if (KEEP) {
    bd.init(&lds, pool.zob, pool.skr(i));
    bd.reset();                                   // _state.reset()
    bd.playout_begin(zlds);
    for (int t = 0; t < mt; ++t) {                // switchBeforeMove: for (i < move_to) _state.forward(moves[i])
      const int c = rfl((int)mv[t]);
      if (c == M_INVALID) continue;               // the reference throws here (go_state.cc:75-77); a refused move is skipped like any other
      bd.forward(c);
    }
    bd.store(&pool.slots[i]);                     // the replayed GoState stays inspectable through elfgo_* (slot i)
  } else {
    const u64* skr = st.skrec + (size_t)r * (st.max_moves + 2) * G::SKW;
    bd.init(&lds, pool.zob, const_cast<u64*>(skr));
    int ck = mt / CK_INTERVAL;                    // checkpoints at or below move_to
    if (ck > st.nck) ck = st.nck;
    if (ck > 0) bd.load(reinterpret_cast<const Slot<N>*>(st.ckpt) + (size_t)r * st.nck + (ck - 1));
    else bd.reset();
    bd.playout_begin(zlds);
    const ReplaySK<N> sk{skr};
    for (int t = ck * CK_INTERVAL; t < mt; ++t) {
      const int c = rfl((int)mv[t]);
      if (c == M_INVALID) continue;
      bd.forward(c, sk);
    }
  }
  // "s": extractStateExtAGZ -> BoardFeature::extractAGZ under the sample's D4 code.  After the store only the history ring of the
  // LDS image is still needed: it moves into the (dead) Bloom words, and the extraction's scratch takes the front of the slot
  // (header + labels + liberties + old ring = 2624 B >= AGZ_SCRATCH_BYTES at 19x19) -- no scratch of its own, so 8 instead of 5
  // waves per SIMD fit the LDS.
  const u64 (*ring)[2][G::R] = lds.hist;
  u64* tpl = tpl_all[ALIAS ? 0 : wv];
  if (ALIAS) {
    Board<N>::wsync();                          // the slot's LDS reads of the store are issued before the ring is overwritten
    u64* ring_src = &lds.hist[0][0][0];
    u64* ring_dst = reinterpret_cast<u64*>(&lds.bloom[0]);
    constexpr int NW = HIST * 2 * G::R, NQ = (NW + 63) / 64;
    u64 ring_v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int j = q * 64 + lane; ring_v[q] = j < NW ? ring_src[j] : 0ull; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int j = q * 64 + lane; if (j < NW) ring_dst[j] = ring_v[q]; }
    Board<N>::wsync();
    ring = reinterpret_cast<const u64 (*)[2][G::R]>(ring_dst);
    tpl = reinterpret_cast<u64*>(&lds);
  }
  char* row = (char*)o.s + (size_t)i * o.s_stride * (o.fmt == FEAT_F16_NHWC ? 2 : 4);
  extract_agz_row<N>(ring, tpl, bd.hist_cnt, bd.next_player, d4, row, o.fmt, lane);
  const int idx = bd.ply - 1;                   // every extractor's move_to = _state.getPly() - 1
  if (lane == 0) {
    if (o.move_idx) o.move_idx[i] = idx;
    if (o.num_move) o.num_move[i] = nm;
    if (o.aug_code) o.aug_code[i] = d4;
    if (o.winner) o.winner[i] = st.winner[r];
    if (o.selfplay_ver) o.selfplay_ver[i] = st.black_ver[r];
    if (o.predicted_value) o.predicted_value[i] = idx < st.num_values[r] ? st.values[(size_t)r * st.max_moves + idx] : 0.0f;
  }
  if (o.offline_a) {                            // extractOfflineAction :129-139
    for (int j = lane; j < o.nfa; j += 64) {
      const int t = idx + j;
      o.offline_a[(size_t)i * o.nfa + j] = t < nm ? (int64_t)coord_to_action<N>(mv[t], d4) : 0;
    }
  }
  if (o.mcts_scores) {                          // extractMCTSPi :107-127
    float* ms = o.mcts_scores + (size_t)i * G::NA;
    const bool have = st.pol != nullptr && idx < rfl(st.num_pol[r]);
    if (have) {
      const unsigned char* pr = st.pol + ((size_t)r * st.max_moves + idx) * G::P;
      static_assert(G::NA <= G::R * 64, "one action per lane per round");
      float v[G::R];
      float sum = 0.0f;                         // integers <= 255 * 362 < 2^24: the fp32 sum is exact in any order
#pragma unroll
      for (int k = 0; k < G::R; ++k) {
        const int a = k * 64 + lane;
        v[k] = 0.0f;
        if (a < G::NA) {
          int coord, a0;
          action_to_coord<N>(a, d4, coord, a0);
          v[k] = (float)pr[coord];
          sum += v[k];
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
#pragma unroll
      for (int k = 0; k < G::R; ++k) {
        const int a = k * 64 + lane;
        if (a < G::NA) ms[a] = __fdiv_rn(v[k], sum);
      }
    } else {
      const int hot = idx < nm ? coord_to_action<N>(mv[idx], d4) : -1;
      for (int a = lane; a < G::NA; a += 64) ms[a] = a == hot ? 1.0f : 0.0f;
    }
  }
}

}  // namespace elfgo
