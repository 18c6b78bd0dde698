// pybind11 boundary of the device-resident self-play engine: the Python-visible modules the reference builds from
//   src_cpp/elf/Pybind.cc:27-117 + elf/pybind_module.cc           -> _elf
//   src_cpp/elfgames/go/inference/Pybind.cc:18-45                  -> _elfgames_go_inference
//   src_cpp/elfgames/go/train/Pybind.cc:18-63                      -> _elfgames_go
// with the same class, method and field names, over the C ABI of libelf_amd.so (include/elf_amd.h) and nothing else: this
// translation unit is plain C++ (g++), contains no HIP and no torch types.  It is what src_py/elf/utils_elf.py
// (Allocator.spec2batches :59-99, GCWrapper._call :368-414, run :426-437) and src_py/elfgames/go/game.py:265-434 drive:
//
//   co, opt = go.ContextOptions(), go.GameOptions(); ...; GC = go.GameContext(co, opt)
//   smem_opts = GC.ctx().createSharedMemOptions(name, bs); smem = GC.ctx().allocateSharedMem(smem_opts, keys)
//   smem[key].field().sz().vec() / .type_name(); smem[key].set(tensor.data_ptr(), byte_strides)
//   GC.ctx().start(); GC.getClient().setRequest(black_ver, -1, thres, -1)
//   smem = GC.ctx().wait(); idx = smem.getSharedMemOptions().idx(); B = smem.effective_batchsize(); ...; GC.ctx().step()
//
// What stands behind it is not the reference's thread-per-game batcher (elf/base/context.h, sharedmem.h, comm.h: collector
// threads, a dispatcher thread, one std::thread per game) but one lock-step device context (elfsp_*): wait() runs the select +
// leaf-feature kernels of every game, hands the rows out in chunks of at most the group's batchsize, step() takes the replies
// back, and once every chunk of the step has been answered runs expansion + backup (and the move / game boundary host logic).
// The memory contract is the reference's: Python owns every tensor, this side keeps (address, byte strides) per
// key and touches rows [0, effective_batchsize) of the inputs before wait() returns and of the replies inside step().
// Addresses may be pinned host memory (what Allocator._alloc makes, utils_elf.py:40-47), pageable host memory, or HIP device
// memory (device-resident batches: no PCIe copy at all); the kind is detected per address (elfgo_pointer_kind).
//
// The three modules are one extension (_elf) plus two thin ones that re-export its classes under the reference's names
// (pybind_go_modules.cc), so every C++ type lives in exactly one shared object.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/elf_amd.h"

namespace py = pybind11;

namespace {

void chk(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + elfgo_error_string(rc) + " (status " + std::to_string(rc) + ")");
}

// ------------------------------------------------------------------------------------------------
// elf::Size / FuncMapBase / AnyP / SharedMemOptions / SharedMem  (elf/base/common.h:44-112, extractor.h:99-360, sharedmem.h:31-232)
// ------------------------------------------------------------------------------------------------
struct Size {
  std::vector<int> sz;
  const std::vector<int>& vec() const { return sz; }
  size_t nelement() const { size_t n = 1; for (int v : sz) n *= (size_t)v; return n; }
  std::vector<int> continuous_strides(int type_size) const {
    std::vector<int> prod(sz.size(), 1);
    for (int i = (int)sz.size() - 1; i >= 1; --i) prod[i - 1] = prod[i] * sz[i];
    for (int& v : prod) v *= type_size;
    return prod;
  }
  std::string info() const {
    std::stringstream ss;
    ss << "(";
    for (int v : sz) ss << v << ",";
    ss << ")";
    return ss.str();
  }
};

struct FuncMapBase {
  std::string name, type_name;
  size_t type_size = 0;
  int batchsize = 0;
  Size extents;
  int getBatchSize() const { return batchsize; }
  const std::string& getName() const { return name; }
  const Size& getSize() const { return extents; }
  std::string getTypeName() const { return type_name; }
  size_t getSizeOfType() const { return type_size; }
  std::string info() const {
    std::stringstream ss;
    ss << "key: " << name << ", batchsize: " << batchsize << ", Size: " << extents.info() << ", Type name: " << type_name;
    return ss.str();
  }
};

struct AnyP {
  const FuncMapBase* f = nullptr;
  uint64_t p = 0;
  std::vector<int> stride;
  int kind = 0;   // elfgo_pointer_kind of p
  const FuncMapBase& field() const { return *f; }
  // AnyP::setAddress (extractor.h:302-305) + setStride (:349-358: one stride per dimension, each >= the contiguous one)
  void setAddress(uint64_t addr, const std::vector<int>& st) {
    if (st.size() != f->extents.sz.size()) throw std::invalid_argument("AnyP.set: " + f->name + ": one byte stride per dimension expected");
    const std::vector<int> def = f->extents.continuous_strides((int)f->type_size);
    for (size_t i = 0; i < st.size(); ++i)
      if (st[i] < def[i]) throw std::invalid_argument("AnyP.set: " + f->name + ": stride smaller than the contiguous stride");
    // rows are moved with one pitched copy each: everything inside a row must be contiguous
    for (size_t i = 1; i < st.size(); ++i)
      if (st[i] != def[i]) throw std::invalid_argument("AnyP.set: " + f->name + ": only the batch dimension may be strided");
    p = addr;
    stride = st;
    kind = elfgo_pointer_kind(reinterpret_cast<const void*>(addr), nullptr);
  }
  size_t row_bytes() const { return f->extents.sz.empty() ? f->type_size : f->extents.nelement() / (size_t)f->extents.sz[0] * f->type_size; }
  size_t pitch() const { return stride.empty() ? row_bytes() : (size_t)stride[0]; }
  std::string info() const {
    std::stringstream ss;
    ss << std::hex << (void*)p << std::dec << ", Field: " << f->info();
    return ss.str();
  }
};

struct SharedMemOptions {
  std::string label;
  int batchsize = 0, idx = -1, timeout_usec = 0;
  SharedMemOptions(const std::string& l, int bs) : label(l), batchsize(bs) {}
  int getIdx() const { return idx; }
  int getBatchSize() const { return batchsize; }
  const std::string& getLabel() const { return label; }
  void setTimeout(int t) { timeout_usec = t; }
  std::string info() const {
    std::stringstream ss;
    ss << "SMem[" << label << "], idx: " << idx << ", batchsize: " << batchsize;
    if (timeout_usec > 0) ss << ", timeout_usec: " << timeout_usec;
    return ss.str();
  }
};

struct SharedMem {
  SharedMemOptions opts;
  std::map<std::string, AnyP> mem;
  size_t active_batch_size = 0;
  SharedMem(int idx, const SharedMemOptions& o) : opts(o) { opts.idx = idx; }
  AnyP* get(const std::string& key) {
    auto it = mem.find(key);
    return it == mem.end() ? nullptr : &it->second;
  }
  const SharedMemOptions& getSharedMemOptions() const { return opts; }
  size_t getEffectiveBatchSize() const { return active_batch_size; }
  std::string info() const {
    std::stringstream ss;
    ss << opts.info() << std::endl;
    for (const auto& kv : mem) ss << "[" << kv.first << "]: " << kv.second.info() << std::endl;
    return ss.str();
  }
};

enum ReplyStatus { DONE_ONE_JOB = 0, SUCCESS, FAILED, UNKNOWN };   // elf/comm/comm.h:89

// ------------------------------------------------------------------------------------------------
// option structs (same fields, same defaults)
// ------------------------------------------------------------------------------------------------
std::string print_bool(bool b) { return b ? "True" : "False"; }

struct SearchAlgoOptions {     // tree_search_options.h:22-75
  bool use_prior = true;
  float c_puct = 5;
  bool unexplored_q_zero = false;
  bool root_unexplored_q_zero = false;
  std::string info() const {
    std::stringstream ss;
    ss << "[prior=" << use_prior << "]";
    if (use_prior) ss << "[c_puct=" << c_puct << "]";
    ss << "[uqz=" << unexplored_q_zero << "][r_uqz=" << root_unexplored_q_zero << "]";
    return ss.str();
  }
};

struct TSOptions {             // tree_search_options.h:77-229
  int max_num_moves = 0;
  int num_threads = 16;
  int num_rollouts_per_thread = 100;
  int num_rollouts_per_batch = 8;
  bool verbose = false;
  bool verbose_time = false;
  int seed = 0;
  bool persistent_tree = false;
  float root_epsilon = 0.0;
  float root_alpha = 0.0;
  std::string log_prefix = "";
  std::string pick_method = "most_visited";
  SearchAlgoOptions alg_opt;
  int virtual_loss = 0;
  std::string info(bool verbose_ = false) const {
    std::stringstream ss;
    if (verbose_) {
      ss << "Maximal #moves (0 = no constraint): " << max_num_moves << std::endl;
      ss << "Seed: " << seed << std::endl;
      ss << "Log Prefix: " << log_prefix << std::endl;
      ss << "#Threads: " << num_threads << std::endl;
      ss << "#Rollout per thread: " << num_rollouts_per_thread << ", #rollouts per batch: " << num_rollouts_per_batch << std::endl;
      ss << "Verbose: " << print_bool(verbose) << ", Verbose_time: " << print_bool(verbose_time) << std::endl;
      ss << "Persistent tree: " << print_bool(persistent_tree) << std::endl;
      ss << "#Virtual loss: " << virtual_loss << std::endl;
      ss << "Pick method: " << pick_method << std::endl;
      if (root_epsilon > 0) ss << "Root exploration: epsilon: " << root_epsilon << ", alpha: " << root_alpha << std::endl;
      ss << "Algorithm: " << alg_opt.info() << std::endl;
    } else {
      ss << "[#th=" << num_threads << "][rl=" << num_rollouts_per_thread << "][per=" << persistent_tree << "][eps=" << root_epsilon
         << "][alpha=" << root_alpha << "]" << alg_opt.info();
    }
    return ss.str();
  }
};

struct ContextOptions {        // elf/legacy/python_options_utils_cpp.h:19-47
  int num_games = 1;
  int batchsize = 0;
  int T = 1;
  std::string job_id;
  TSOptions mcts_options;
  void print() const {
    std::cout << "JobId: " << job_id << std::endl << "#Game: " << num_games << std::endl << "T: " << T << std::endl
              << mcts_options.info() << std::endl;
  }
};

struct GameOptions {           // elfgames/go/common/go_game_specific.h:16-268
  unsigned int seed = 0;
  int num_future_actions = 3;
  int num_games_per_thread = -1;
  std::string mode;
  bool use_mcts = false;
  bool use_mcts_ai2 = false;
  bool black_use_policy_network_only = false;
  bool white_use_policy_network_only = false;
  int data_aug = -1;
  float start_ratio_pre_moves = 0.5;
  float ratio_pre_moves = 0.0;
  int move_cutoff = -1;
  int policy_distri_cutoff = 20;
  bool policy_distri_training_for_all = false;
  float resign_thres = 0.05;
  float resign_thres_lower_bound = 1e-9;
  float resign_thres_upper_bound = 0.50;
  float resign_prob_never = 0.1;
  float resign_target_fp_rate = 0.05;
  int resign_target_hist_size = 2500;
  int num_reset_ranking = 5000;
  std::string preload_sgf;
  int preload_sgf_move_to = -1;
  bool use_df_feature = false;
  int q_min_size = 10;
  int q_max_size = 1000;
  int num_reader = 50;
  float komi = 7.5;
  int ply_pass_enabled = 0;
  float white_puct = -1.0;
  int white_mcts_rollout_per_batch = -1;
  int white_mcts_rollout_per_thread = -1;
  int eval_num_games = 400;
  float eval_thres = 0.55;
  int client_max_delay_sec = 1200;
  int selfplay_init_num = 5000;
  int selfplay_update_num = 1000;
  bool selfplay_async = false;
  bool following_pass = false;
  bool cheat_eval_new_model_wins_half = false;
  bool cheat_selfplay_random_result = false;
  bool keep_prev_selfplay = false;
  int eval_num_threads = 1;
  int expected_num_clients = -1;
  std::vector<std::string> list_files;
  std::string server_addr;
  std::string server_id;
  int port = 0;
  bool verbose = false;
  bool print_result = false;
  std::string dump_record_prefix;
  std::string time_signature;
  // --- what the reference fixes at compile time (BOARD9x9, base/board.h:29-41) or takes from the process environment; not
  // fields of the reference's struct, defaults reproduce it
  int board_size = 19;
  int gpu = -1;                // HIP device that owns boards and trees; -1 = the calling thread's current device at start()
  int nodes_per_game = 0;      // tree node records per game; 0 = 4 x rollouts per move + 1024
  int keep_records = 0;        // keep the Record JSON of the last N finished games (GameContext.popRecords())
  int log_searches = 0;        // keep the first N search results (Context.searchLog(), tests)

  GameOptions() {
    char buf[64];
    const time_t t = time(nullptr);
    struct tm tmv;
    localtime_r(&t, &tmv);
    strftime(buf, sizeof(buf), "%y%m%d-%H%M%S", &tmv);
    time_signature = buf;
  }
  std::string info() const {
    std::stringstream ss;
    ss << "Seed: " << seed << std::endl;
    ss << "Time signature: " << time_signature << std::endl;
    ss << "#FutureActions: " << num_future_actions << std::endl;
    ss << "mode: " << mode << std::endl;
    ss << "UseMCTS: " << print_bool(use_mcts) << std::endl;
    ss << "MoveCutOff: " << move_cutoff << std::endl;
    ss << "PolicyDistriCutOff: " << policy_distri_cutoff << std::endl;
    ss << "Policy distri training for all moves: " << print_bool(policy_distri_training_for_all) << std::endl;
    ss << "Min Ply from which pass is enabled: " << ply_pass_enabled << std::endl;
    ss << "Resign Threshold: " << resign_thres << std::endl;
    ss << "Komi: " << komi << std::endl;
    if (!preload_sgf.empty()) ss << "Preload SGF:" << preload_sgf << ", move_to" << preload_sgf_move_to << std::endl;
    ss << "Board size: " << board_size << ", device: " << gpu << std::endl;
    return ss.str();
  }
};

struct WinRateStats {          // common/game_utils.h:86-107
  uint64_t black_wins = 0, white_wins = 0;
  float sum_reward = 0.0;
  uint64_t total_games = 0;
  void feed(float reward) {
    if (reward > 0) black_wins++; else white_wins++;
    sum_reward += reward;
    total_games++;
  }
};

struct GameStats {             // common/game_stats.h:19-68 (the counters this engine feeds)
  WinRateStats wr;
  WinRateStats getWinRateStats() const { return wr; }
  std::vector<std::string> getPlayedGames() const { return {}; }   // feedSgf is never called by the reference either (distri_client.h:236)
};

// Sgf::load + iterator (sgf/sgf.cc) -> the Coords GoGameSelfPlay::restart forwards / follows (game_selfplay.cc:202-219,392-405: only
// the entries' moves are used, not their colours).  Entries without a move (uninitialised in the reference) are left out.
std::vector<uint16_t> sgf_main_line(const std::string& path, int n) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("preload_sgf: cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string t = ss.str();
  const int k = elfrec_sgf_parse(n, t.c_str(), nullptr, nullptr, 0, nullptr);
  if (k <= 0) throw std::runtime_error("preload_sgf: " + path + " is not an SGF game");
  std::vector<int32_t> players((size_t)k);
  std::vector<uint16_t> coords((size_t)k), out;
  elfrec_sgf_parse(n, t.c_str(), players.data(), coords.data(), k, nullptr);
  for (int i = 0; i < k; ++i) if (players[(size_t)i] != 0) out.push_back(coords[(size_t)i]);
  return out;
}

// ------------------------------------------------------------------------------------------------
// elf::Context (elf/base/context.h:198-421) over one lock-step device context
// ------------------------------------------------------------------------------------------------
class Context;

// the calling thread's current device is switched to the context's for a scope and restored afterwards
struct DevScope {
  int prev = -1;
  explicit DevScope(int dev) {
    int cur = -1;
    if (elfgo_get_device(&cur) == 0 && cur != dev && elfgo_set_device(dev) == 0) prev = cur;
  }
  ~DevScope() { if (prev >= 0) elfgo_set_device(prev); }
  DevScope(const DevScope&) = delete;
  DevScope& operator=(const DevScope&) = delete;
};

static int pick_code(const std::string& m) {      // TreeSearchT::chooseAction, tree_search.h:506-519
  if (m == "most_visited") return ELFSP_PICK_MOST_VISITED;
  if (m == "strongest_prior") return ELFSP_PICK_STRONGEST_PRIOR;
  if (m == "uniform_random") return ELFSP_PICK_UNIFORM_RANDOM;
  return -1;
}

// GoGameSelfPlay accessors (common/game_selfplay.h:41-56)
struct GameView {
  Context* ctx = nullptr;
  int game = 0;
  std::string showBoard() const;
  std::string getNextPlayer() const;
  std::string getLastMove() const;
  float getScore() const;
  float getLastScore() const;
};

class Context {
 public:
  Context(const ContextOptions& co, const GameOptions& go, bool online) : co_(co), go_(go), online_(online) {
    const int n = go.board_size;
    if (n != 19 && n != 9) throw std::range_error("board_size must be 19 or 9");
    if (co.batchsize <= 0) throw std::range_error("ContextOptions.batchsize must be positive");
    if (co.num_games <= 0) throw std::range_error("ContextOptions.num_games must be positive");
    if (online && co.num_games != 1)
      throw std::range_error("mode online: one game per context (the human_actor prompt is per game)");
    if (go.use_df_feature) throw std::range_error("use_df_feature: the DarkForest feature set is not on this path (AGZ planes only)");
    const TSOptions& ts = co.mcts_options;
    if (pick_code(ts.pick_method) < 0) throw std::range_error("MCTS Pick method unknown! " + ts.pick_method);   // tree_search.h:521-524
    if ((int64_t)ts.num_threads * ts.num_rollouts_per_batch > elfmcts_max_rollouts_per_step() || ts.num_threads < 1 || ts.num_rollouts_per_batch < 1)
      throw std::range_error("num_threads x num_rollouts_per_batch must be in [1, " + std::to_string(elfmcts_max_rollouts_per_step()) +
                             "] (the leaf table of one search step)");
    // GoFeature::registerExtractor (common/game_feature.h:159-206) with batchsize = ContextOptions.batchsize
    const int B = co.batchsize, NA = n * n + 1;
    add_field("s", "float", 4, B, {B, 18, n, n});
    add_field("a", "int64_t", 8, B, {B});
    add_field("rv", "int64_t", 8, B, {B});
    add_field("offline_a", "int64_t", 8, B, {B, go.num_future_actions});
    for (const char* k : {"V", "winner", "predicted_value"}) add_field(k, "float", 4, B, {B});
    for (const char* k : {"pi", "mcts_scores"}) add_field(k, "float", 4, B, {B, NA});
    for (const char* k : {"move_idx", "aug_code", "num_move"}) add_field(k, "int32_t", 4, B, {B});
    for (const char* k : {"black_ver", "white_ver", "selfplay_ver"}) add_field(k, "int64_t", 8, B, {B});
    views_.resize(co.num_games);
    for (int i = 0; i < co.num_games; ++i) { views_[i].ctx = this; views_[i].game = i; }
  }
  ~Context() { destroy(); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;

  // ---- the pybind surface of elf::Context
  SharedMemOptions createSharedMemOptions(const std::string& name, int batchsize) { return SharedMemOptions(name, batchsize); }

  SharedMem& allocateSharedMem(const SharedMemOptions& options, const std::vector<std::string>& keys) {
    if (started_) throw std::runtime_error("allocateSharedMem after start()");
    if (options.batchsize <= 0 || options.batchsize > co_.batchsize)
      throw std::range_error("SharedMem batchsize must be in [1, ContextOptions.batchsize]");   // sharedmem.h:147-157 asserts
    smems_.emplace_back(new SharedMem((int)smems_.size(), options));
    SharedMem& sm = *smems_.back();
    for (const std::string& k : keys) {
      auto it = fields_.find(k);
      if (it == fields_.end()) {
        std::cerr << "Warning! key[" << k << "] is missing!" << std::endl;                        // Extractor::getAnyP :559-566
        continue;
      }
      AnyP a;
      a.f = &it->second;
      sm.mem.emplace(k, a);
    }
    by_label_[options.label].push_back((int)smems_.size() - 1);
    return sm;
  }

  void start() {
    if (started_) return;
    const int n = go_.board_size;
    const TSOptions& ts = co_.mcts_options;
    ElfSpOptions o;
    memset(&o, 0, sizeof(o));
    o.board_size = n;
    o.num_games = co_.num_games;
    const int64_t per_move = (int64_t)ts.num_rollouts_per_thread * ts.num_threads;
    o.nodes_per_game = go_.nodes_per_game > 0 ? (go_.nodes_per_game + 63) / 64 * 64 : (int)((4 * per_move + 1024 + 63) / 64 * 64);
    o.num_rollouts_per_thread = ts.num_rollouts_per_thread;
    o.persistent_tree = ts.persistent_tree;
    o.root_epsilon = ts.root_epsilon;
    o.root_alpha = ts.root_alpha;
    o.seed = go_.seed;
    o.policy_distri_cutoff = go_.policy_distri_cutoff;
    o.move_cutoff = go_.move_cutoff;
    o.resign_thres = 0.0f;             // ClientCtrl defaults until a request arrives (common/record.h:36-38)
    o.never_resign_prob = 0.0f;
    o.keep_records = go_.keep_records;
    if (!go_.dump_record_prefix.empty() && o.keep_records < co_.num_games) o.keep_records = co_.num_games;   // the SGF is written from the record
    o.log_searches = go_.log_searches;
    o.policy_distri_training_for_all = go_.policy_distri_training_for_all;
    o.model_ver = 0;
    o.game_idx_base = 0;
    o.job_hash = (uint64_t)std::hash<std::string>{}(co_.job_id);
    o.mcts.num_rollouts_per_batch = ts.num_rollouts_per_batch;
    o.mcts.virtual_loss = ts.virtual_loss;
    o.mcts.use_prior = ts.alg_opt.use_prior;
    o.mcts.unexplored_q_zero = ts.alg_opt.unexplored_q_zero;
    o.mcts.root_unexplored_q_zero = ts.alg_opt.root_unexplored_q_zero;
    o.mcts.c_puct = ts.alg_opt.c_puct;
    o.mcts.komi = go_.komi;
    o.mcts.ply_pass_enabled = go_.ply_pass_enabled;
    o.mcts.remove_pass_if_dangerous = 1;    // MCTSActorParams defaults (go/mcts/mcts.h:17-37)
    o.mcts.rotation_flip = 1;
    o.mcts.num_threads = ts.num_threads;
    o.mcts.required_version = -1;           // set by the first request
    o.white_puct = go_.white_puct;
    o.white_mcts_rollout_per_batch = go_.white_mcts_rollout_per_batch;
    o.white_mcts_rollout_per_thread = go_.white_mcts_rollout_per_thread;
    o.black_use_policy_network_only = go_.black_use_policy_network_only;
    o.white_use_policy_network_only = go_.white_use_policy_network_only;
    o.pick_method = pick_code(ts.pick_method);
    o.cheat_eval_new_model_wins_half = go_.cheat_eval_new_model_wins_half;
    o.cheat_selfplay_random_result = go_.cheat_selfplay_random_result;
    o.following_pass = online_ && go_.following_pass;
    std::vector<uint64_t> zob = load_zobrist(n);
    int dev = go_.gpu;
    if (dev < 0) chk(elfgo_get_device(&dev), "hipGetDevice");
    device_ = dev;
    chk(elfsp_create(&o, dev, zob.data(), &sp_), "elfsp_create");
    opt_ = o;
    NA_ = n * n + 1;
    row_floats_ = 18 * n * n;
    // device staging: one step of leaf rows, and the replies of one step ("actor_white" staging is made by the first request
    // that names a model for White)
    alloc_staging(0);
    dmalloc(&d_one_, 4096);   // scratch: ids @0, info @64, board colours @256, liberties @1024, a float @3072
    if (!go_.preload_sgf.empty()) {
      const std::vector<uint16_t> mv = sgf_main_line(go_.preload_sgf, n);
      chk(elfsp_preload(sp_, mv.data(), (int)mv.size(), go_.preload_sgf_move_to, stream_), "preload_sgf");
    }
    started_ = true;
  }

  void stop() {
    // Context::stop (context.h:329-370) pumps wait/step(FAILED) until the game threads have joined; here nothing runs between
    // wait() and step(), so stopping is forgetting the open step
    stopped_ = true;
  }

  std::string version() const { return elfgo_version(); }

  // wait(): the next batch of some group, nullptr (None) when timeout_usec > 0 and nothing can be served
  SharedMem* wait(int timeout_usec) {
    if (!started_) throw std::runtime_error("Context.wait() before start()");
    if (stopped_) return nullptr;
    if (current_) throw std::runtime_error("Context.wait(): the previous batch has not been released with step()");
    for (;;) {
      // game_start: once per (re)start of the games by a request (common/dispatcher_callback.h:86-88)
      if (pending_game_start_ > 0) {
        --pending_game_start_;
        SharedMem* sm = pick("game_start");
        if (sm) {
          put_scalar(sm, "black_ver", black_ver_);
          put_scalar(sm, "white_ver", white_ver_);
          return serve(sm, 1, K_GAME_START);
        }
        continue;
      }
      // game_end: once per finished game (GameNotifier::OnGameEnd, train/distri_client.h:228-240)
      if (pending_game_end_ > 0) {
        --pending_game_end_;
        SharedMem* sm = pick("game_end");
        if (sm) return serve(sm, 1, K_GAME_END);
        continue;
      }
      if (rows_pending()) return serve_actor_chunk();
      if (have_request_) {
        // every game waits (a wait request / numThreads = 0): nothing will ever come out of a step
        int64_t pr[6];
        chk(elfsp_progress(sp_, pr), "elfsp_progress");
        if (pr[4] == co_.num_games) {
          if (timeout_usec > 0) return nullptr;
          throw std::runtime_error("Context.wait(): every game is waiting for a request that gives it something to play");
        }
      }
      if (!have_request_)
        throw std::runtime_error(
            "Context.wait(): no request yet -- call getClient().setRequest(black_ver, -1, resign_thres, -1) (or GameContext.setRequest) "
            "after start(); the reference's games also wait for their first MsgRequest (common/record.h:85, game_selfplay.cc:277-279), "
            "which its ZMQ server sends and which is out of scope here");
      if (online_ && !search_running_) {
        if (SharedMem* sm = human_prompt()) return sm;   // nullptr: the reply was consumed without a batch (cannot happen) or search starts
        continue;
      }
      begin_search_step();
    }
  }

  void step(ReplyStatus status) {
    if (!current_) return;   // stop() pumps step(FAILED) without a batch in the reference; nothing to release here
    SharedMem* sm = current_;
    const int kind = current_kind_;
    current_ = nullptr;
    if (status == FAILED) {                 // a failed batch ends the session (act unsuccessful, go/mcts/mcts.h:112-114)
      stopped_ = true;
      return;
    }
    if (kind == K_ACTOR) take_actor_reply(sm);
    else if (kind == K_HUMAN) take_human_reply(sm);
  }

  // ---- services for GameContext / Client
  void setRequest(int64_t black_ver, int64_t white_ver, float thres, int num_threads, bool player_swap = false) {
    // GameContext::setRequest (inference/game_context.h:76-88) / Client::setRequest (train/distri_client.h:318-331).  player_swap
    // (ClientCtrl.player_swap) is not an argument of the reference's method: there it only arrives in the server's MsgRequest
    // (train/ctrl_eval.h); offered here as an optional fifth argument because the server is out of scope.
    if (!started_) throw std::runtime_error("setRequest before start()");
    ElfSpRequest q;
    memset(&q, 0, sizeof(q));
    q.black_ver = black_ver;
    q.white_ver = white_ver;
    q.black_resign_thres = q.white_resign_thres = thres;
    q.never_resign_prob = 0.0f;
    q.num_game_thread_used = num_threads;
    q.player_swap = player_swap ? 1 : 0;
    q.async = go_.selfplay_async ? 1 : 0;
    if (white_ver >= 0 && black_ver >= 0) alloc_staging(1);
    chk(elfsp_set_request2(sp_, &q), "elfsp_set_request");
    if (black_ver >= 0) have_request_ = true;
    // applied at once by the games that are between two searches (the first request always is): its game_start batch comes
    // before anything else
    pending_game_start_ += elfsp_take_game_starts(sp_, &black_ver_, &white_ver_);
  }

  std::map<std::string, int> getParams() const {   // GoFeature::getParams, common/game_feature.h:208-221
    const int n = go_.board_size;
    return {{"num_action", n * n + 1}, {"board_size", n}, {"num_future_actions", go_.num_future_actions}, {"num_planes", 18},
            {"our_stone_plane", 0}, {"opponent_stone_plane", 1}, {"ACTION_SKIP", -100}, {"ACTION_PASS", -99},
            {"ACTION_RESIGN", -98}, {"ACTION_CLEAR", -97}};
  }

  // Limits of this engine that the reference does not have, so that a caller can size a job instead of discovering them by error
  // code (not part of the reference's interface; getParams() stays exactly the reference's dictionary)
  std::map<std::string, int64_t> getLimits() const {
    const TSOptions& ts = co_.mcts_options;
    const int64_t per_move = (int64_t)ts.num_rollouts_per_thread * ts.num_threads;
    const int64_t npg = go_.nodes_per_game > 0 ? (go_.nodes_per_game + 63) / 64 * 64 : (4 * per_move + 1024 + 63) / 64 * 64;
    const int n = go_.board_size;
    // what a game adds to the context's SHARED node pool at npg ids, with the per-game tables of THESE search options (path rows per leaf
    // of a step, one D4 window per search thread: 512 KB + at the 1024-leaf maximum) -- the node pool itself belongs to all games
    const int rpb = ts.num_rollouts_per_batch > 0 ? ts.num_rollouts_per_batch : 1, nth = ts.num_threads > 0 ? ts.num_threads : 1;
    const int64_t steps = (ts.num_rollouts_per_thread + rpb - 1) / rpb;
    const int64_t d4w = steps * rpb * nth;
    const int64_t tree_bytes = (int64_t)elfmcts_tree_bytes_per_game2(n, (int)npg, nth, rpb, (int)std::min<int64_t>(d4w, (int64_t)1 << 30));
    const int64_t node_bytes = npg > 0 ? (tree_bytes + npg - 1) / npg : 0;     // small record + its share of the big pool and the id arrays
    std::map<std::string, int64_t> out = {{"max_rollouts_per_step", elfmcts_max_rollouts_per_step()}, {"nodes_per_game", npg},
                                          {"node_bytes", node_bytes}, {"tree_bytes_per_game_per_ai", tree_bytes},
                                          {"node_pool_shared_by_games", 1}};
    size_t fr = 0, tot = 0;
    int dev = go_.gpu;
    if (dev < 0 && elfgo_get_device(&dev) != 0) dev = 0;
    if (elfgo_mem_info(dev, &fr, &tot) == 0) {
      out["hbm_free_bytes"] = (int64_t)fr;
      out["hbm_total_bytes"] = (int64_t)tot;
      if (tree_bytes > 0) out["max_games_by_free_hbm"] = (int64_t)(fr * 9 / 10) / tree_bytes;
    }
    return out;
  }

  const GameView* getGame(int i) const {
    if (i < 0 || i >= (int)views_.size()) {
      std::cerr << "Invalid game_idx [" << i << "]" << std::endl;   // game_context.h:66-70
      return nullptr;
    }
    return &views_[i];
  }

  GameStats& stats() { return stats_; }
  std::vector<std::string> popRecords() {
    std::vector<std::string> out;
    if (!sp_) return out;
    drain_records();
    out.assign(kept_records_.begin(), kept_records_.end());
    kept_records_.clear();
    return out;
  }
  // finished games' records out of the device context; GameOptions.dump_record_prefix: finish_game writes each as an SGF file
  // (game_selfplay.cc:133-135 -> GoStateExt::dumpSgf)
  void drain_records() {
    while (elfsp_records_pending(sp_) > 0) {
      size_t len = 0;
      elfsp_pop_record(sp_, nullptr, 0, &len);
      std::string buf(len + 1, '\0');
      chk(elfsp_pop_record(sp_, &buf[0], len + 1, &len), "elfsp_pop_record");
      buf.resize(len);
      if (!go_.dump_record_prefix.empty()) {
        char name[1024];
        const int64_t n = elfrec_record_to_sgf(&opt_, buf.c_str(), go_.dump_record_prefix.c_str(), name, sizeof(name), nullptr, 0);
        if (n > 0) {
          std::string text((size_t)n + 1, '\0');
          if (elfrec_record_to_sgf(&opt_, buf.c_str(), go_.dump_record_prefix.c_str(), name, sizeof(name), &text[0], (size_t)n + 1) == n) {
            text.resize((size_t)n);
            std::ofstream oo(name);
            oo << text << std::endl;
          }
        }
      }
      kept_records_.push_back(std::move(buf));
      const size_t cap = (size_t)std::max(opt_.keep_records, 1);
      while (kept_records_.size() > cap) kept_records_.pop_front();
    }
  }
  void setStream(uint64_t s) { stream_ = reinterpret_cast<void*>(s); }
  int device() const { return device_; }
  // search log of the device context (tests): list of (game, move_played, best_action, total_visits, n_edges, coords, visits)
  py::list searchLog() {
    py::list out;
    if (!sp_) return out;
    int64_t st[12];
    chk(elfsp_stats(sp_, st), "elfsp_stats");
    const int n = (int)st[5], NE = elfmcts_edge_stride(elfsp_mcts(sp_));
    if (n == 0) return out;
    std::vector<ElfSpSearch> rec(n);
    std::vector<int32_t> coord((size_t)n * NE), visits((size_t)n * NE);
    std::vector<float> reward((size_t)n * NE);
    chk(elfsp_search_log(sp_, 0, n, rec.data(), coord.data(), visits.data(), nullptr, reward.data()), "elfsp_search_log");
    for (int i = 0; i < n; ++i) {
      const int ne = rec[i].n_edges;
      py::list c, v, r;
      for (int j = 0; j < ne; ++j) { c.append(coord[(size_t)i * NE + j]); v.append(visits[(size_t)i * NE + j]); r.append(reward[(size_t)i * NE + j]); }
      out.append(py::make_tuple(rec[i].game, rec[i].move_played, rec[i].best_action, rec[i].total_visits, ne, c, v, r));
    }
    return out;
  }
  // used by GameView
  ElfSelfPlay* sp() const { return sp_; }
  float komi() const { return go_.komi; }
  int board_size() const { return go_.board_size; }
  void* stream() const { return stream_; }

 private:
  enum { K_ACTOR = 0, K_GAME_START, K_GAME_END, K_HUMAN };

  void add_field(const std::string& name, const std::string& type, size_t tsz, int bs, std::vector<int> ext) {
    FuncMapBase f;
    f.name = name; f.type_name = type; f.type_size = tsz; f.batchsize = bs; f.extents.sz = std::move(ext);
    fields_.emplace(name, std::move(f));
  }

  static std::vector<uint64_t> load_zobrist(int n) {
    // the reference's 441 Zobrist constants (base/hash_num.h:12), shipped as data next to the package
    py::object here = py::module_::import("os").attr("path").attr("dirname")(py::module_::import("elf_amd").attr("__file__"));
    const std::string path = py::str(here).cast<std::string>() + "/data/zobrist21.bin";
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint64_t> z(441);
    f.read(reinterpret_cast<char*>(z.data()), 441 * 8);
    if (!f) throw std::runtime_error("short read: " + path);
    z.resize((size_t)(n + 2) * (n + 2));
    return z;
  }

  void alloc_staging(int a) {
    if (d_s_[a]) return;
    max_rows_[a] = elfsp_max_rows_actor(sp_, a);
    dmalloc(&d_s_[a], (size_t)max_rows_[a] * row_floats_ * 4);
    dmalloc(&d_pi_[a], (size_t)max_rows_[a] * NA_ * 4);
    dmalloc(&d_v_[a], (size_t)max_rows_[a] * 4);
    dmalloc(&d_rv_[a], (size_t)max_rows_[a] * 8);
  }

  void dmalloc(void** p, size_t bytes) {
    DevScope _ds(device_);
    chk(elfgo_malloc(p, bytes), "hipMalloc");
    owned_.push_back(*p);
  }

  void destroy() {
    if (sp_) { elfsp_destroy(sp_); sp_ = nullptr; }
    if (!owned_.empty()) {
      DevScope _ds(device_);
      for (void* p : owned_) elfgo_free(p);
      owned_.clear();
    }
  }

  SharedMem* pick(const std::string& label) {     // round robin over the num_recv buffers of a group (utils_elf.py:82-97)
    auto it = by_label_.find(label);
    if (it == by_label_.end() || it->second.empty()) return nullptr;
    int& rr = rr_[label];
    SharedMem* sm = smems_[it->second[rr % it->second.size()]].get();
    rr = (rr + 1) % (int)it->second.size();
    return sm;
  }

  SharedMem* serve(SharedMem* sm, size_t rows, int kind) {
    sm->active_batch_size = rows;
    current_ = sm;
    current_kind_ = kind;
    return sm;
  }

  AnyP* need(SharedMem* sm, const char* key, const char* type) {
    AnyP* a = sm->get(key);
    if (!a) throw std::runtime_error(std::string("SharedMem[") + sm->opts.label + "] has no key " + key);
    if (a->p == 0) throw std::runtime_error(std::string("SharedMem[") + sm->opts.label + "][" + key + "]: no address set (AnyP.set)");
    if (a->f->type_name != type) throw std::runtime_error(std::string(key) + ": unexpected element type");
    return a;
  }

  void put_scalar(SharedMem* sm, const char* key, int64_t v) {
    AnyP* a = sm->get(key);
    if (!a || a->p == 0) return;
    chk(elfgo_memcpy2d_async(reinterpret_cast<void*>(a->p), 8, &v, 8, 8, 1, stream_), "memcpy");
    chk(elfgo_stream_sync(stream_), "sync");
  }

  // ---- MCTS leaves: one device step served in chunks of at most the group's batchsize; rows of the "actor_black" AI first, then
  // (evaluation games) the rows of the "actor_white" AI in their own group
  void begin_search_step() {
    int rows[2] = {0, 0};
    void* dst[2] = {d_s_[0], d_s_[1]};
    chk(elfsp_begin_step2(sp_, dst, row_floats_, rows, stream_), "elfsp_begin_step");
    pending_game_start_ += elfsp_take_game_starts(sp_, &black_ver_, &white_ver_);   // a request that (re)started games at this boundary
    search_running_ = true;
    for (int a = 0; a < 2; ++a) { step_rows_[a] = rows[a]; rows_left_[a] = rows[a]; rows_done_[a] = 0; have_rv_[a] = false; }
    if (rows[0] + rows[1] == 0) finish_search_step();   // nothing for the net in this step (every leaf terminal / revisited / games idle)
  }

  bool rows_pending() const { return rows_left_[0] > 0 || rows_left_[1] > 0; }

  SharedMem* serve_actor_chunk() {
    const int a = rows_left_[0] > 0 ? 0 : 1;
    const char* group = a == 0 ? "actor_black" : "actor_white";   // MCTSActorParams.actor_name (game_selfplay.cc:166,172)
    SharedMem* sm = pick(group);
    if (!sm) throw std::runtime_error(std::string("no SharedMem allocated for group ") + group);
    const int chunk = std::min(rows_left_[a], sm->opts.batchsize);
    AnyP* s = need(sm, "s", "float");
    const size_t rb = (size_t)row_floats_ * 4;
    chk(elfgo_memcpy2d_async(reinterpret_cast<void*>(s->p), s->pitch(), (const char*)d_s_[a] + (size_t)rows_done_[a] * rb, rb, rb,
                             (size_t)chunk, stream_), "copy of the feature rows");
    chk(elfgo_stream_sync(stream_), "sync");   // rows [0, chunk) are in the caller's tensor before wait() returns
    chunk_rows_ = chunk;
    chunk_actor_ = a;
    return serve(sm, (size_t)chunk, K_ACTOR);
  }

  void take_actor_reply(SharedMem* sm) {
    const int a = chunk_actor_;
    AnyP* pi = need(sm, "pi", "float");
    AnyP* v = need(sm, "V", "float");
    const size_t pb = (size_t)NA_ * 4;
    chk(elfgo_memcpy2d_async((char*)d_pi_[a] + (size_t)rows_done_[a] * pb, pb, reinterpret_cast<const void*>(pi->p), pi->pitch(), pb,
                             (size_t)chunk_rows_, stream_), "copy of pi");
    chk(elfgo_memcpy2d_async((char*)d_v_[a] + (size_t)rows_done_[a] * 4, 4, reinterpret_cast<const void*>(v->p), v->pitch(), 4,
                             (size_t)chunk_rows_, stream_), "copy of V");
    AnyP* rv = sm->get("rv");
    have_rv_[a] = rv && rv->p != 0;
    if (have_rv_[a])
      chk(elfgo_memcpy2d_async((char*)d_rv_[a] + (size_t)rows_done_[a] * 8, 8, reinterpret_cast<const void*>(rv->p), rv->pitch(), 8,
                               (size_t)chunk_rows_, stream_), "copy of rv");
    // "a" (ReplyAction) is a reply key of the actor groups too; the MCTS actor never reads GoReply.c (go/mcts/mcts.h:209-230)
    rows_done_[a] += chunk_rows_;
    rows_left_[a] -= chunk_rows_;
    if (!rows_pending()) finish_search_step();
  }

  void finish_search_step() {
    const int64_t games_before = elfsp_games_finished(sp_);
    const float* pis[2] = {step_rows_[0] > 0 ? (const float*)d_pi_[0] : nullptr, step_rows_[1] > 0 ? (const float*)d_pi_[1] : nullptr};
    const float* vs[2] = {step_rows_[0] > 0 ? (const float*)d_v_[0] : nullptr, step_rows_[1] > 0 ? (const float*)d_v_[1] : nullptr};
    const int64_t* rvs[2] = {have_rv_[0] ? (const int64_t*)d_rv_[0] : nullptr, have_rv_[1] ? (const int64_t*)d_rv_[1] : nullptr};
    const int rc = elfsp_end_step2(sp_, pis, NA_, vs, rvs, stream_);
    if (rc == ELFGO_E_MCTS_BASE - ELFMCTS_E_VERSION || (rc < ELFGO_E_MCTS_BASE && ((ELFGO_E_MCTS_BASE - rc) & ELFMCTS_E_VERSION)))
      throw std::runtime_error("model version of a reply (rv) and required version (black " + std::to_string(black_ver_) + ", white " +
                               std::to_string(white_ver_) + ") are not consistent");      // go/mcts/mcts.h:210-217
    chk(rc, "elfsp_end_step");
    int64_t pr[6];
    chk(elfsp_progress(sp_, pr), "elfsp_progress");     // host counters, no device synchronisation
    if (pr[2] == 0) search_running_ = false;            // no search is open any more: the move boundary has been passed
    const int64_t done = elfsp_games_finished(sp_) - games_before;
    if (done > 0) note_finished((int)done);
  }

  void note_finished(int done) {
    if (!go_.dump_record_prefix.empty()) drain_records();
    std::vector<float> fv((size_t)done + 8);
    const int k = elfsp_take_finished(sp_, fv.data(), (int)fv.size());
    for (int i = 0; i < k; ++i) stats_.wr.feed(fv[i]);
    pending_game_end_ += done;
  }

  // ---- online mode: the human_actor half of GoGameSelfPlay::act (game_selfplay.cc:290-330), one game
  SharedMem* human_prompt() {
    // if (s.terminated()) finish_game(FR_ILLEGAL) :294-297
    int32_t info[ELFGO_INFO_WORDS];
    board_info(0, info);
    if (info[9]) {
      const int32_t g0 = 0;
      const int64_t before = elfsp_games_finished(sp_);
      chk(elfsp_finish(sp_, &g0, 1, ELFSP_FR_ILLEGAL, stream_), "finish_game");
      note_finished((int)(elfsp_games_finished(sp_) - before));
      return nullptr;
    }
    SharedMem* sm = pick("human_actor");
    if (!sm) { search_running_ = true; begin_search_step(); return nullptr; }   // no human group: the AI plays every move
    AnyP* s = need(sm, "s", "float");
    // BoardFeature bf(s): D4 code 0 (no random symmetry for the human's view)
    chk(elfgo_extract_agz(elfsp_engine(sp_), nullptr, nullptr, 1, (float*)d_s_[0], row_floats_, stream_), "extract_agz");
    const size_t rb = (size_t)row_floats_ * 4;
    chk(elfgo_memcpy2d_async(reinterpret_cast<void*>(s->p), s->pitch(), d_s_[0], rb, rb, 1, stream_), "copy of the feature row");
    chk(elfgo_stream_sync(stream_), "sync");
    return serve(sm, 1, K_HUMAN);
  }

  void take_human_reply(SharedMem* sm) {
    AnyP* a = need(sm, "a", "int64_t");
    int64_t act = 0;
    chk(elfgo_memcpy2d_async(&act, 8, reinterpret_cast<const void*>(a->p), 8, 8, 1, stream_), "copy of a");
    chk(elfgo_stream_sync(stream_), "sync");
    const int n = go_.board_size;
    const int32_t g0 = 0;
    const int64_t before = elfsp_games_finished(sp_);
    if (act == -100) {               // SA_SKIP -> M_SKIP: "skip the current move, and ask the ai to move" :303-305
      begin_search_step();
      return;
    }
    if (act == -97) {                // SA_CLEAR :306-311
      int32_t info[ELFGO_INFO_WORDS];
      board_info(0, info);
      if (info[0] > 1) chk(elfsp_finish(sp_, &g0, 1, ELFSP_FR_CLEAR, stream_), "finish_game");   // !justStarted()
    } else if (act == -98) {         // SA_RESIGN :313-316
      chk(elfsp_finish(sp_, &g0, 1, ELFSP_FR_RESIGN, stream_), "finish_game");
    } else {
      int32_t c;
      if (act == -99 || act == n * n || act == -1) c = 0;                      // SA_PASS / action2Coord(pass)
      else if (act < 0 || act > n * n) c = 3;                                 // M_INVALID
      else c = (int32_t)((act % n + 1) * (n + 2) + (act / n + 1));            // action2Coord, D4 code 0: a = x*N + y
      // an invalid move leaves the game untouched; the next wait() prompts again ("please try again" :323-327)
      const int rc = (c == 3) ? ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD : elfsp_play(sp_, &c, stream_);
      if (rc != 0 && rc != ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD) chk(rc, "elfsp_play");
      if (rc != 0) std::cerr << "Invalid move: action " << act << " please try again" << std::endl;
    }
    const int64_t done = elfsp_games_finished(sp_) - before;
    if (done > 0) note_finished((int)done);
  }

 public:
  // host copies of one game board's info record / stones / score, for GameView and the human prompt
  void board_info(int game, int32_t* out) const {
    const int32_t g = game;
    char* d = (char*)d_one_;
    DevScope _ds(device_);
    chk(elfgo_memcpy_h2d(d, &g, 4), "memcpy");
    chk(elfgo_info(elfsp_engine(sp_), (const int32_t*)d, 1, (int32_t*)(d + 64), stream_), "elfgo_info");
    chk(elfgo_stream_sync(stream_), "sync");
    chk(elfgo_memcpy_d2h(out, d + 64, ELFGO_INFO_WORDS * 4), "memcpy");
  }
  void board_stones(int game, std::vector<uint8_t>* colour) const {
    const int32_t g = game;
    const int np = go_.board_size * go_.board_size;
    char* d = (char*)d_one_;
    DevScope _ds(device_);
    chk(elfgo_memcpy_h2d(d, &g, 4), "memcpy");
    chk(elfgo_export_board(elfsp_engine(sp_), (const int32_t*)d, 1, (uint8_t*)(d + 256), (int16_t*)(d + 1024), stream_), "elfgo_export_board");
    chk(elfgo_stream_sync(stream_), "sync");
    colour->resize(np);
    chk(elfgo_memcpy_d2h(colour->data(), d + 256, np), "memcpy");
  }
  float board_score(int game) const {
    const int32_t g = game;
    char* d = (char*)d_one_;
    float v = 0;
    DevScope _ds(device_);
    chk(elfgo_memcpy_h2d(d, &g, 4), "memcpy");
    chk(elfgo_evaluate(elfsp_engine(sp_), (const int32_t*)d, 1, go_.komi, (float*)(d + 3072), stream_), "elfgo_evaluate");
    chk(elfgo_stream_sync(stream_), "sync");
    chk(elfgo_memcpy_d2h(&v, d + 3072, 4), "memcpy");
    return v;
  }
  bool started() const { return started_; }

 private:
  ContextOptions co_;
  GameOptions go_;
  bool online_ = false;
  std::map<std::string, FuncMapBase> fields_;
  std::vector<std::unique_ptr<SharedMem>> smems_;
  std::map<std::string, std::vector<int>> by_label_;
  std::map<std::string, int> rr_;
  std::vector<GameView> views_;
  GameStats stats_;
  ElfSpOptions opt_{};                       // the options the device context was created with
  std::deque<std::string> kept_records_;     // records taken out of the device context, waiting for popRecords()
  ElfSelfPlay* sp_ = nullptr;
  void* stream_ = nullptr;
  int device_ = 0, max_rows_[2] = {0, 0}, NA_ = 0, row_floats_ = 0;
  void *d_s_[2] = {nullptr, nullptr}, *d_pi_[2] = {nullptr, nullptr}, *d_v_[2] = {nullptr, nullptr}, *d_rv_[2] = {nullptr, nullptr};
  void* d_one_ = nullptr;
  std::vector<void*> owned_;
  bool started_ = false, stopped_ = false, have_request_ = false, search_running_ = false, have_rv_[2] = {false, false};
  int64_t black_ver_ = 0, white_ver_ = -1;
  int pending_game_start_ = 0, pending_game_end_ = 0;
  int step_rows_[2] = {0, 0}, rows_left_[2] = {0, 0}, rows_done_[2] = {0, 0}, chunk_rows_ = 0, chunk_actor_ = 0;
  SharedMem* current_ = nullptr;
  int current_kind_ = 0;
};


// ---- GoGameSelfPlay accessors ------------------------------------------------------------------------------------------------
static void need_started(const Context* c) {
  if (!c->started()) throw std::runtime_error("the games exist once Context.start() has been called");
}

// GoState::showBoard -> showBoard2Buf(board, SHOW_LAST_MOVE) (base/board.cc:1414-1479)
std::string GameView::showBoard() const {
  need_started(ctx);
  const int n = ctx->board_size(), S = n + 2;
  std::vector<uint8_t> col;
  ctx->board_stones(game, &col);
  int32_t info[ELFGO_INFO_WORDS];
  ctx->board_info(game, info);
  const int last = info[2];
  std::string prompt;
  for (int x = 0; x < n; ++x) { prompt += (char)('A' + (x >= 8 ? x + 1 : x)); prompt += ' '; }
  prompt.pop_back();
  auto star = [&](int i, int j) {
    if (n == 19) return (i == 3 || i == 9 || i == 15) && (j == 3 || j == 9 || j == 15);
    return (i == 2 || i == 6) && (j == 2 || j == 6);
  };
  std::stringstream ss;
  ss << "   " << prompt << "\n";
  for (int j = n - 1; j >= 0; --j) {
    char b[16];
    snprintf(b, sizeof(b), "%2d ", j + 1);
    ss << b;
    for (int i = 0; i < n; ++i) {
      const int s = col[i * n + j], c = (j + 1) * S + (i + 1);
      if (s == 1) ss << (c == last ? "X)" : "X ");
      else if (s == 2) ss << (c == last ? "O)" : "O ");
      else ss << (star(i, j) ? "+ " : ". ");
    }
    ss << (j + 1);
    if (j == n / 2 + 1) ss << "     WHITE (O) has captured " << info[8] << " stones";
    else if (j == n / 2) ss << "     BLACK (X) has captured " << info[7] << " stones";
    ss << "\n";
  }
  ss << "   " << prompt;
  return ss.str();
}

std::string GameView::getNextPlayer() const {   // player2str, sgf/sgf.h:58-71
  need_started(ctx);
  int32_t info[ELFGO_INFO_WORDS];
  ctx->board_info(game, info);
  return info[1] == 1 ? "B" : info[1] == 2 ? "W" : "U";
}

std::string GameView::getLastMove() const {     // coord2str2(GoStateExt::lastMove()), sgf/sgf.h:73-85, go_state_ext.h:105-110
  need_started(ctx);
  int32_t info[ELFGO_INFO_WORDS];
  ctx->board_info(game, info);
  int c = info[2];
  if (info[0] <= 1) {                           // justStarted(): the last move of the game that just ended
    std::vector<int32_t> lm(elfmcts_num_games(elfsp_mcts(ctx->sp())));
    chk(elfsp_last_moves(ctx->sp(), lm.data()), "elfsp_last_moves");
    c = lm[game] < 0 ? 3 : lm[game];
  }
  if (c == 0) return "PASS";
  if (c == 1) return "RESIGN";
  const int S = ctx->board_size() + 2;
  int x = c % S - 1;
  const int y = c / S - 1;
  if (x >= 8) x++;
  return std::string(1, (char)('A' + x)) + std::to_string(y + 1);
}

float GameView::getScore() const { need_started(ctx); return ctx->board_score(game); }

float GameView::getLastScore() const {
  need_started(ctx);
  std::vector<float> v(elfmcts_num_games(elfsp_mcts(ctx->sp())));
  chk(elfsp_last_score(ctx->sp(), v.data()), "elfsp_last_score");
  return v[game];
}

// ---- GameContext (inference/game_context.h:29-122, train/game_context.h:33-133), Client (train/distri_client.h:262-331) ----------
struct Client {
  Context* ctx;
  void setRequest(int64_t black_ver, int64_t white_ver, float thres, int numThreads, bool player_swap) {
    ctx->setRequest(black_ver, white_ver, thres, numThreads, player_swap);
  }
  GameStats& getGameStats() { return ctx->stats(); }
};

struct Server {};   // never instantiated: the training server is out of scope (getServer() is None, as in the reference's client modes)

class GameContextBase {
 public:
  GameContextBase(const ContextOptions& co, const GameOptions& opt, bool online) : ctx_(new Context(co, opt, online)), client_{ctx_.get()} {}
  Context* ctx() { return ctx_.get(); }
  std::map<std::string, int> getParams() const { return ctx_->getParams(); }
  std::map<std::string, int64_t> getLimits() const { return ctx_->getLimits(); }
  const GameView* getGame(int i) const { return ctx_->getGame(i); }
  std::vector<std::string> popRecords() { return ctx_->popRecords(); }
 protected:
  std::unique_ptr<Context> ctx_;
  Client client_;
};

// _elfgames_go_inference.GameContext: "Only works for online setting." (inference/game_context.h:36-40)
class GameContextInference : public GameContextBase {
 public:
  GameContextInference(const ContextOptions& co, const GameOptions& opt) : GameContextBase(co, check(opt), true) {}
  void setRequest(int64_t black_ver, int64_t white_ver, float thres, int numThreads) { ctx_->setRequest(black_ver, white_ver, thres, numThreads); }
 private:
  static const GameOptions& check(const GameOptions& opt) {
    if (opt.mode != "online") throw std::range_error("options.mode not recognized! " + opt.mode);
    return opt;
  }
};

// _elfgames_go.GameContext: train / offline_train create the Server + GoGameTrain threads (train/game_context.h:49-56) -- the
// training side's replay buffer and ZMQ plumbing are out of scope here (the trainer's input pipeline is elf_amd.ReplayLoader /
// elftrain_*); every other mode is a self-play client
class GameContextTrain : public GameContextBase {
 public:
  GameContextTrain(const ContextOptions& co, const GameOptions& opt) : GameContextBase(co, check(opt), opt.mode == "online") {}
  Client* getClient() { return &client_; }
  Server* getServer() { return nullptr; }
 private:
  static const GameOptions& check(const GameOptions& opt) {
    if (opt.mode == "train" || opt.mode == "offline_train")
      throw std::range_error("options.mode " + opt.mode + ": the training server (replay buffer, ZMQ) is not part of this library; "
                             "use elf_amd.ReplayLoader for the trainer's batches");
    if (opt.mode != "selfplay" && opt.mode != "online") throw std::range_error("options.mode not recognized! " + opt.mode);
    return opt;
  }
};

}  // namespace

#define RW(C, f) .def_readwrite(#f, &C::f)

PYBIND11_MODULE(_elf, m) {
  m.doc() = "elf_amd: the reference's _elf / _elfgames_go* pybind surface over the MI355X-native self-play engine";
  auto ref = py::return_value_policy::reference_internal;

  py::enum_<ReplyStatus>(m, "ReplyStatus")
      .value("SUCCESS", SUCCESS).value("FAILED", FAILED).value("UNKNOWN", UNKNOWN).export_values();

  py::class_<Context>(m, "Context")
      .def("wait", &Context::wait, py::arg("timeout_usec") = 0, ref, py::call_guard<py::gil_scoped_release>())
      .def("step", &Context::step, py::arg("success") = SUCCESS, py::call_guard<py::gil_scoped_release>())
      .def("start", &Context::start)
      .def("stop", &Context::stop)
      .def("version", &Context::version)
      .def("allocateSharedMem", &Context::allocateSharedMem, ref)
      .def("createSharedMemOptions", &Context::createSharedMemOptions)
      // beyond the reference: the HIP stream (hipStream_t as int) device-resident batches are ordered on (default: the NULL
      // stream, which is ordered with PyTorch's default stream), the device, and the search log
      .def("setStream", &Context::setStream)
      .def("device", &Context::device)
      .def("searchLog", &Context::searchLog);

  py::class_<Size>(m, "Size").def("vec", &Size::vec, ref);

  py::class_<SharedMemOptions>(m, "SharedMemOptions")
      .def("idx", &SharedMemOptions::getIdx)
      .def("batchsize", &SharedMemOptions::getBatchSize)
      .def("label", &SharedMemOptions::getLabel, ref)
      .def("setTimeout", &SharedMemOptions::setTimeout);

  py::class_<SharedMem>(m, "SharedMem")
      .def("__getitem__", &SharedMem::get, ref)
      .def("getSharedMemOptions", &SharedMem::getSharedMemOptions, ref)
      .def("effective_batchsize", &SharedMem::getEffectiveBatchSize)
      .def("info", &SharedMem::info);

  py::class_<AnyP>(m, "AnyP")
      .def("info", &AnyP::info)
      .def("field", &AnyP::field, ref)
      .def("set", &AnyP::setAddress);

  py::class_<FuncMapBase>(m, "FuncMapBase")
      .def("batchsize", &FuncMapBase::getBatchSize)
      .def("name", &FuncMapBase::getName, ref)
      .def("sz", &FuncMapBase::getSize, ref)
      .def("type_name", &FuncMapBase::getTypeName)
      .def("type_size", &FuncMapBase::getSizeOfType);

  py::class_<SearchAlgoOptions>(m, "SearchAlgoOptions")
      .def(py::init<>())
      RW(SearchAlgoOptions, use_prior) RW(SearchAlgoOptions, c_puct) RW(SearchAlgoOptions, unexplored_q_zero)
      RW(SearchAlgoOptions, root_unexplored_q_zero)
      .def("info", &SearchAlgoOptions::info);

  py::class_<TSOptions>(m, "TSOptions")
      .def(py::init<>())
      RW(TSOptions, max_num_moves) RW(TSOptions, num_threads) RW(TSOptions, num_rollouts_per_thread) RW(TSOptions, num_rollouts_per_batch)
      RW(TSOptions, verbose) RW(TSOptions, persistent_tree) RW(TSOptions, pick_method) RW(TSOptions, log_prefix) RW(TSOptions, virtual_loss)
      RW(TSOptions, verbose_time) RW(TSOptions, alg_opt) RW(TSOptions, root_epsilon) RW(TSOptions, root_alpha) RW(TSOptions, seed)
      .def("info", &TSOptions::info, py::arg("verbose") = false);

  // the reference registers these with both game modules (inference/Pybind.cc:31-37, train/Pybind.cc:44-52)
  auto go = m.def_submodule("_go", "classes re-exported as _elfgames_go_inference / _elfgames_go");
  py::class_<ContextOptions>(go, "ContextOptions")
      .def(py::init<>())
      RW(ContextOptions, job_id) RW(ContextOptions, batchsize) RW(ContextOptions, num_games) RW(ContextOptions, T)
      RW(ContextOptions, mcts_options)
      .def("print", &ContextOptions::print);

  py::class_<GameOptions>(go, "GameOptions")
      .def(py::init<>())
      RW(GameOptions, seed) RW(GameOptions, mode) RW(GameOptions, data_aug) RW(GameOptions, start_ratio_pre_moves)
      RW(GameOptions, ratio_pre_moves) RW(GameOptions, move_cutoff) RW(GameOptions, num_future_actions) RW(GameOptions, list_files)
      RW(GameOptions, verbose) RW(GameOptions, num_games_per_thread) RW(GameOptions, use_mcts) RW(GameOptions, server_addr)
      RW(GameOptions, server_id) RW(GameOptions, port) RW(GameOptions, policy_distri_cutoff) RW(GameOptions, client_max_delay_sec)
      RW(GameOptions, q_min_size) RW(GameOptions, q_max_size) RW(GameOptions, num_reader) RW(GameOptions, dump_record_prefix)
      RW(GameOptions, use_mcts_ai2) RW(GameOptions, preload_sgf) RW(GameOptions, preload_sgf_move_to) RW(GameOptions, komi)
      RW(GameOptions, print_result) RW(GameOptions, resign_thres) RW(GameOptions, resign_thres_lower_bound)
      RW(GameOptions, resign_thres_upper_bound) RW(GameOptions, resign_prob_never) RW(GameOptions, resign_target_fp_rate)
      RW(GameOptions, num_reset_ranking) RW(GameOptions, ply_pass_enabled) RW(GameOptions, following_pass) RW(GameOptions, use_df_feature)
      RW(GameOptions, policy_distri_training_for_all) RW(GameOptions, black_use_policy_network_only)
      RW(GameOptions, white_use_policy_network_only) RW(GameOptions, cheat_eval_new_model_wins_half)
      RW(GameOptions, cheat_selfplay_random_result) RW(GameOptions, eval_num_games) RW(GameOptions, selfplay_init_num)
      RW(GameOptions, selfplay_update_num) RW(GameOptions, selfplay_async) RW(GameOptions, white_puct)
      RW(GameOptions, white_mcts_rollout_per_batch) RW(GameOptions, white_mcts_rollout_per_thread) RW(GameOptions, eval_thres)
      RW(GameOptions, keep_prev_selfplay) RW(GameOptions, expected_num_clients)
      // beyond the reference's fields (see the struct): what it fixes at compile time / takes from the environment
      RW(GameOptions, board_size) RW(GameOptions, gpu) RW(GameOptions, nodes_per_game) RW(GameOptions, keep_records)
      RW(GameOptions, log_searches)
      .def("info", &GameOptions::info);

  py::class_<WinRateStats>(go, "WinRateStats")
      .def(py::init<>())
      RW(WinRateStats, black_wins) RW(WinRateStats, white_wins) RW(WinRateStats, sum_reward) RW(WinRateStats, total_games);

  py::class_<GameStats>(go, "GameStats")
      .def("getWinRateStats", &GameStats::getWinRateStats)
      .def("getPlayedGames", &GameStats::getPlayedGames);

  py::class_<GameView>(go, "GoGameSelfPlay")
      .def("showBoard", &GameView::showBoard)
      .def("getNextPlayer", &GameView::getNextPlayer)
      .def("getLastMove", &GameView::getLastMove)
      .def("getScore", &GameView::getScore)
      .def("getLastScore", &GameView::getLastScore);

  py::class_<Client>(go, "Client")
      .def("setRequest", &Client::setRequest, py::arg("black_ver"), py::arg("white_ver"), py::arg("thres"), py::arg("numThreads") = -1,
           py::arg("player_swap") = false)
      .def("getGameStats", &Client::getGameStats, ref);

  py::class_<Server>(go, "Server");
  // the main line of an SGF file as reference Coords: what GameOptions.preload_sgf is turned into (exposed for tests)
  go.def("sgf_main_line", [](const std::string& path, int board_size) { return sgf_main_line(path, board_size); });

  py::class_<GameContextInference>(go, "GameContextInference")
      .def(py::init<const ContextOptions&, const GameOptions&>())
      .def("ctx", &GameContextInference::ctx, ref)
      .def("getParams", &GameContextInference::getParams)
      .def("getGame", &GameContextInference::getGame, ref)
      .def("setRequest", &GameContextInference::setRequest, py::arg("black_ver"), py::arg("white_ver"), py::arg("thres"),
           py::arg("numThreads") = -1)
      .def("getLimits", &GameContextInference::getLimits)
      .def("popRecords", &GameContextInference::popRecords);

  py::class_<GameContextTrain>(go, "GameContextTrain")
      .def(py::init<const ContextOptions&, const GameOptions&>())
      .def("ctx", &GameContextTrain::ctx, ref)
      .def("getParams", &GameContextTrain::getParams)
      .def("getGame", &GameContextTrain::getGame, ref)
      .def("getClient", &GameContextTrain::getClient, ref)
      .def("getServer", &GameContextTrain::getServer, ref)
      .def("getLimits", &GameContextTrain::getLimits)
      .def("popRecords", &GameContextTrain::popRecords);

  // _elf._logging / _elf._options (elf/logging/Pybind.cc, elf/options/Pybind.cc: spdlog factories, OptionSpec/OptionMap) are the
  // reference's option/logging plumbing -- control plane, out of scope; the submodules exist so that
  // `import _elf` exposes the same attribute names, and say so when used
  auto lg = m.def_submodule("_logging", "not provided: the reference's spdlog bindings are control plane (out of scope)");
  auto op = m.def_submodule("_options", "not provided: the reference's OptionSpec/OptionMap bindings are control plane (out of scope)");
  (void)lg; (void)op;
}
