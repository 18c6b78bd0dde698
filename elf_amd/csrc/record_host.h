// Host-side self-play record assembly shared by record_host.cpp (formatting) and selfplay_host.hip (capture).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

struct SpRecordMeta {          // the MsgRequest half of a Record (record.h:119-149), constant per self-play context
  int board_size;
  int64_t black_ver, white_ver;
  // TSOptions (tree_search_options.h:179-194)
  int num_threads, num_rollouts_per_thread, num_rollouts_per_batch, virtual_loss;
  bool persistent_tree, use_prior, unexplored_q_zero, root_unexplored_q_zero;
  float c_puct, root_epsilon, root_alpha;
  // ClientCtrl (record.h:32-55)
  float black_resign_thres, white_resign_thres, never_resign_prob;
  int num_game_thread_used;
  bool player_swap = false, async = false;
  int pick_method = 0;       // ELFSP_PICK_*
  int client_type = 1;       // ClientCtrl.client_type as the server sent it (CLIENT_SELFPLAY_ONLY = 1, record.h:24-29)
  // TSOptions fields no search reads, echoed as the request carried them
  int max_num_moves = 0;
  int64_t ts_seed = 0;
  bool verbose = false, verbose_time = false;
  std::string log_prefix;
};
struct ElfTsOptions;
void elfrec_meta_set_ts(SpRecordMeta* m, const ElfTsOptions& t);   // the TSOptions half of the meta from a request's mcts_opt

struct SpRecord {              // the MsgResult half (record.h:184-234) + Record's own fields (:236-262)
  std::vector<uint16_t> moves;            // GoState::getAllMoves()
  std::vector<uint8_t> policies;          // [num_policies][(N+2)^2] CoordRecord.prob
  std::vector<float> values;              // predicted values, one per search
  float reward = 0.0f;
  bool never_resign = false;
  int num_move = 0;
  uint64_t timestamp = 0, thread_id = 0;
  int seq = 0;
  std::vector<int64_t> using_models;      // GoStateExt::using_models_ (a std::set: ascending, unique); empty = the request's versions
};

struct ElfSpOptions;
// the MsgRequest a self-play context with these options works under (Client::setRequest, train/distri_client.h:318-331)
SpRecordMeta elfrec_meta_from_options(const ElfSpOptions& o);
// Record::setJsonFields + nlohmann::json::dump() (compact, keys in std::map order)
std::string elfrec_record_json(const SpRecordMeta& m, const SpRecord& r);
// GoStateExt::addMCTSPolicy (go_state_ext.h:158-181): appends one (N+2)^2-byte row to `policies`
void elfrec_append_policy(int board_size, const int32_t* coord, const float* prob, int n, std::vector<uint8_t>* policies);
