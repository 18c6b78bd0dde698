/* elf_amd.h -- C ABI of the MI355X-native ELF OpenGo self-play hot path (libelf_amd.so).
 *
 * Drop-in boundary for the board engine that the reference implements on the CPU in
 *   src_cpp/elfgames/go/base/{board,go_state,board_feature}.cc
 * and calls from src_cpp/elfgames/go/mcts/mcts.h and common/game_selfplay.cc.  Plain C: opaque
 * engine handle, raw pointers, sizes, int status.  No torch / C++ types cross this boundary.
 *
 * Conventions
 *  - status: 0 = ok, >0 = hipError_t from the runtime, <0 = ELFGO_E_* argument errors.
 *    No exceptions cross the ABI (the reference throws std::range_error for M_INVALID,
 *    go_state.cc:75-77; here the per-board `ok` byte is 0xFF for that move).
 *  - a board is addressed by its slot index in the engine-owned HBM pool.
 *  - every bulk pointer (ids, moves, masks, features, ...) is a DEVICE pointer unless the name ends
 *    in _host; `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue
 *    work and return; use elfgo_sync() or your own stream sync before reading results.
 *  - `ids` may be NULL, meaning slots [0, n).
 *  - devices: a handle belongs to the HIP device it was created on; every call switches to that device for its duration and
 *    restores the calling thread's current device on return, so handles on different GPUs can be driven from one process.
 *    `stream` must be a stream of the handle's device (NULL = that device's default stream).
 *  - threading: a handle (engine, search, self-play context, replay store) may be driven by one host thread at a time; different
 *    handles are independent.  Calls on one handle are ordered by the stream they are given; results written by a call on
 *    stream A may be consumed by a call on stream B only after the caller has ordered the streams (event / sync).
 *  - moves are reference Coords: c = (y+1)*(N+2) + (x+1), M_PASS=0, M_RESIGN=1 (base/board.h:183,
 *    base/common.h:43-47).  Actions are NN action ids a = x*N + y, pass = N*N (board.h:189).
 */
#ifndef ELF_AMD_H_
#define ELF_AMD_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ElfGoEngine ElfGoEngine;

#define ELFGO_E_BADARG (-1)
#define ELFGO_E_BADSIZE (-2)
#define ELFGO_E_NOMEM (-3)
#define ELFGO_E_NODATA (-4)   /* a draw found no eligible record (elfrq_draw); the draw state is left as it was before the call */

#define ELFGO_INFO_WORDS 16
/* per-board info record written by elfgo_info (int32 words):
 *  0 ply  1 next_player  2 last_move  3 last_move2  4 ko_age  5 simple_ko  6 simple_ko_color
 *  7 b_cap  8 w_cap  9 terminated  10 superko  11 hist_len  12 sk_len  13 hash_lo  14 hash_hi  15 0 */

/* Engine = HBM pool of `capacity` board slots for one board size (19 or 9) on HIP device `device`.
 * `zobrist` = the (N+2)^2 64-bit constants indexed by reference Coord (base/hash_num.h:12), host ptr. */
int elfgo_create(int board_size, int capacity, int device, const uint64_t* zobrist_host, ElfGoEngine** out);
int elfgo_destroy(ElfGoEngine* e);
int elfgo_board_size(const ElfGoEngine* e);
int elfgo_capacity(const ElfGoEngine* e);
size_t elfgo_slot_bytes(const ElfGoEngine* e);
int elfgo_sync(ElfGoEngine* e, void* stream);

/* GoState::reset (go_state.cc:134-141) / clearBoard (board.cc:79-107) */
int elfgo_reset(ElfGoEngine* e, const int32_t* ids, int n, void* stream);
/* GoState copy constructor (go_state.h:117-124): slot dst[i] <- slot src[i], superko records included */
int elfgo_copy(ElfGoEngine* e, const int32_t* dst_ids, const int32_t* src_ids, int n, void* stream);
/* GoState::forward (go_state.cc:74-94). moves[i] int32 Coord; ok[i] = 1 played, 0 refused, 0xFF M_INVALID */
int elfgo_forward(ElfGoEngine* e, const int32_t* ids, const int32_t* moves, int n, uint8_t* ok, void* stream);
/* legal-move mask: mask[i][a] = GoState::checkMove(action2Coord(a)) (go_state.cc:123-128 via
 * go/mcts/mcts.h:300-312), a in [0, N*N], D4 code 0; row stride = N*N+1 bytes */
int elfgo_legal_mask(ElfGoEngine* e, const int32_t* ids, int n, uint8_t* mask, void* stream);
/* BoardFeature::extractAGZ (board_feature.cc:247-290) under D4 code d4[i] (board_feature.h:88-113):
 * dst[i] = fp32 [18][N][N]; consecutive boards are `stride_floats` apart (>= 18*N*N).
 * This is the write into the batcher's "s" tensor (common/game_feature.h:38-40). d4 NULL = code 0. */
int elfgo_extract_agz(ElfGoEngine* e, const int32_t* ids, const int32_t* d4, int n, float* dst,
                      int64_t stride_floats, void* stream);
/* The same extraction with a selectable row format (emit what an fp16 channels_last net reads, so the
 * fp32->fp16 cast and the NCHW->NHWC permute of the reference's trainer.py/model path disappear):
 *   ELFGO_FEAT_F32_NCHW  fp32 [18][N][N]  (the reference's "s" row, what elfgo_extract_agz writes)
 *   ELFGO_FEAT_F16_NHWC  fp16 [N][N][18]  (= torch channels_last of a [18,N,N] half tensor)
 * stride_elems counts elements of the chosen type between consecutive rows (>= 18*N*N). */
#define ELFGO_FEAT_F32_NCHW 0
#define ELFGO_FEAT_F16_NHWC 1
int elfgo_extract_agz_fmt(ElfGoEngine* e, const int32_t* ids, const int32_t* d4, int n, void* dst, int64_t stride_elems,
                          int fmt, void* stream);
/* GoState::evaluate (go_state.h:194-203): Tromp-Taylor area(black) - area(white) - komi; superko -> +-1 */
int elfgo_evaluate(ElfGoEngine* e, const int32_t* ids, int n, float komi, float* out, void* stream);
/* per-board info records (ELFGO_INFO_WORDS int32 each) */
int elfgo_info(ElfGoEngine* e, const int32_t* ids, int n, int32_t* out, void* stream);
/* stone colour per point (uint8, 0/1/2) and liberties of the group on it (int16, 0 if empty), action order */
int elfgo_export_board(ElfGoEngine* e, const int32_t* ids, int n, uint8_t* colour, int16_t* libs, void* stream);
/* from each slot's current position play uniformly random legal, non-true-eye
 * moves until GoState::terminated(); pass when none.  The move is the (rand % count)-th candidate in x-major order with the
 * counter RNG   key = fmix32(lo32(seed)) ^ fmix32(hi32(seed) + 0x7F4A7C15);   rand = fmix32(key + ply * 0x9E3779B9)
 * (32-bit wrapping arithmetic; fmix32 = murmur3's 32-bit finaliser; oracle/go_oracle.c and oracle/ref_capi.cc state the same
 * function for the CPU checkers).
 * out[i] = {hash_lo, hash_hi, ply, steps}.  Whole games run inside one launch, position in LDS. */
int elfgo_playout(ElfGoEngine* e, const int32_t* ids, const uint64_t* seeds, int n, int max_steps,
                  uint32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-resident MCTS (one tree per game) -- replaces, behind the same seams, the reference's
 *   elf/ai/tree_search/tree_search.h        TreeSearchT / TreeSearchSingleThreadT::batch_rollouts (:200-262)
 *   elf/ai/tree_search/tree_search_node.h   NodeT / SearchTreeT
 *   elfgames/go/mcts/mcts.h                 MCTSActor::evaluate / pi2response (:73-121, :256-332)
 * The net stays outside: elfmcts_select() writes the leaf features of every game into the caller's
 * "s" tensor (the write GoFeature::extractStateAGZ does into the batcher's tensor, common/game_feature.h:38-40)
 * and elfmcts_expand() consumes the caller's "pi"/"V" reply tensors (ReplyPolicy/ReplyValue, :42-71).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ElfMcts ElfMcts;

/* TSOptions + SearchAlgoOptions (elf/ai/tree_search/tree_search_options.h:23-229), MCTSActorParams (go/mcts/mcts.h:17-37) */
typedef struct ElfMctsOptions {
  int32_t num_rollouts_per_batch;   /* TSOptions.num_rollouts_per_batch */
  int32_t virtual_loss;             /* TSOptions.virtual_loss */
  int32_t use_prior;                /* alg_opt.use_prior */
  int32_t unexplored_q_zero;        /* alg_opt.unexplored_q_zero */
  int32_t root_unexplored_q_zero;   /* alg_opt.root_unexplored_q_zero */
  float c_puct;                     /* alg_opt.c_puct */
  float komi;                       /* MCTSActorParams.komi */
  int32_t ply_pass_enabled;         /* MCTSActorParams.ply_pass_enabled */
  int32_t remove_pass_if_dangerous; /* MCTSActorParams.remove_pass_if_dangerous */
  int32_t rotation_flip;            /* MCTSActorParams.rotation_flip */
  int32_t num_threads;              /* TSOptions.num_threads: search threads per game (tree_search.h:345-368), >= 1; one step runs
                                     * their batch_rollouts in sequence on the shared tree, so a move costs
                                     * num_threads x num_rollouts_per_thread rollouts (tree_search.h:472-476).
                                     * num_threads x num_rollouts_per_batch must be <= elfmcts_max_rollouts_per_step() = 1024
                                     * (the leaf table of one step; ELFGO_E_BADARG otherwise) */
  int32_t reserved0;
  int64_t required_version;         /* MCTSActorParams.required_version: < 0 = replies of any model version are accepted */
} ElfMctsOptions;

#define ELFMCTS_E_POOL 1      /* the context's node pool is exhausted (raise nodes_per_game: the pool holds num_games x nodes_per_game ids) */
#define ELFMCTS_E_ROOT_HASH 2 /* TreeSearch::Root state is not the same as the input state (tree_search.h:488-492) */
#define ELFMCTS_E_FORWARD 4   /* a tree edge could not be played */
#define ELFMCTS_E_RNG 8       /* more D4 draws requested than uploaded with elfmcts_set_d4 */
#define ELFMCTS_E_VERSION 16  /* a reply row's model version ("rv") differs from required_version (go/mcts/mcts.h:209-217) */

#define ELFMCTS_ROOT_WORDS 8
/* per-game record of elfmcts_root (int32 words): 0 n_edges 1 num_visits 2 status 3 root id 4 V (float bits)
 * 5 d4 draws consumed this move 6 error bits 7 node ids the game's tree holds */

/* `num_games` trees over boards of engine `e` (game g searches from board slot board_ids[g]).
 * Node memory is ONE pool of num_games x nodes_per_game node ids (nodes_per_game a multiple of 64) shared by the context's games:
 * the reference takes its nodes from the heap (SearchTreeT::addNode, tree_search_node.h:439-443; recursiveFree :445-467), so a game
 * whose kept subtree is large simply holds more of them -- size nodes_per_game for the MEAN tree, not for the worst game.  A step's ids
 * are popped from the pool's free stack with one atomic per game, treeAdvance pushes the freed subtrees back.  d4_window = opt->num_threads x (max D4 draws ONE search thread makes per move):
 * every search thread's MCTSActor owns a generator (TreeSearchT's actor_gen, tree_search.h:339-343; all seeded alike,
 * game_selfplay.cc:45-47,77), so the draws are laid out as one window per thread. */
int elfmcts_create(ElfGoEngine* e, int num_games, int nodes_per_game, int d4_window, const ElfMctsOptions* opt, ElfMcts** out);
int elfmcts_destroy(ElfMcts* m);
/* new options for the following steps; num_threads must be the value the trees were created with (ELFGO_E_BADARG otherwise) */
int elfmcts_set_options(ElfMcts* m, const ElfMctsOptions* opt);
int elfmcts_num_threads(const ElfMcts* m);
/* D4 draws each search thread's actor has consumed this move: host int32 [num_games][num_threads], copied on `stream`
 * (synchronise before reading).  Thread 0's count is also RootInfo word 5. */
int elfmcts_thread_draws(ElfMcts* m, int32_t* out_host, void* stream);
/* row format elfmcts_select / elfsp_begin_step write into s_dst (ELFGO_FEAT_*; default fp32 NCHW). With
 * ELFGO_FEAT_F16_NHWC s_dst points to halfs and the stride argument counts halfs. */
int elfmcts_set_feature_format(ElfMcts* m, int fmt);
int elfmcts_get_feature_format(const ElfMcts* m, int* fmt);
/* largest num_threads x num_rollouts_per_batch a search step can hold (1024: the stride of the per-game leaf tables; the step's table in LDS is sized by the launch) */
int elfmcts_max_rollouts_per_step(void);
/* Which games the per-game launches that follow act on: mask = device bytes [num_games], 0 = the game is left alone (no root
 * check, no noise, no descents, no rows), 1 = it searches, 2 = TreeSearchT::runPolicyOnly (tree_search.h:385-407: the root is
 * evaluated if it has not been, nothing else).  NULL (the default) = every game searches.  Applies to elfmcts_set_root,
 * elfmcts_dirichlet, elfmcts_select (and through it to expand / backup).  The pointer is kept, not copied. */
int elfmcts_set_game_mask(ElfMcts* m, const uint8_t* mask);
/* MCTSActorParams.required_version per game (device int64 [num_games], < 0 = any version); NULL (the default) = the value of
 * ElfMctsOptions.required_version for every game.  The pointer is kept, not copied. */
int elfmcts_set_required_versions(ElfMcts* m, const int64_t* versions);
int elfmcts_num_games(const ElfMcts* m);
int elfmcts_edge_stride(const ElfMcts* m);   /* row length of the per-edge arrays (368 at 19x19, 96 at 9x9) */
/* HBM a game adds to a context at `nodes_per_game` node ids (its share of the pool's records of both classes and id arrays, plus
 * what is per game: stash, leaf / row tables, path rows, D4 windows): what a caller sizes num_games x nodes_per_game against
 * elfgo_mem_info.  A node lives in a 5 888-B record (19x19; 1 920 B at 9x9) until its 17th edge is followed, then in an 11 520-B one;
 * one big record per 16 small ones exists and that class cannot run out first (a big node has >= 17 children that are nodes).
 * The first form assumes 1 search thread, 16 rollouts per batch, a 1024-draw window; the second takes them (path rows are
 * 512 B per leaf of a step: 512 KB per game at the 1024-leaf maximum). */
size_t elfmcts_tree_bytes_per_game(int board_size, int nodes_per_game);
size_t elfmcts_tree_bytes_per_game2(int board_size, int nodes_per_game, int num_threads, int rollouts_per_batch, int d4_window);
size_t elfmcts_node_bytes(const ElfMcts* m);   /* elfmcts_tree_bytes_per_game2 / nodes_per_game, rounded up (6 6xx B at 19x19) */
/* Node ids of the context (synchronises the device).  out8_host: 0 small records in all, 1 of them free (on the pool's stack or in
 * a game's stash), 2 big records in all, 3 free, 4 ids held by all trees now, 5 by the largest tree now, 6 the most ONE tree has
 * held since the last reset, 7 the sum over games of those per-tree maxima (what G fixed per-game pools would have to provide:
 * compare with 4).  reset_peaks != 0 restarts the maxima.  No counterpart in the reference (its nodes live on the heap). */
int elfmcts_pool_info(ElfMcts* m, int64_t* out8_host, int reset_peaks);
/* test / debug service: node ids held per game (host int32 [num_games]), counted by a scan of the pool's owner array -- what RootInfo
 * word 7 tracks incrementally.  Synchronises the device. */
int elfmcts_count_live(ElfMcts* m, int32_t* out_host);
/* SearchTreeT::clear (tree_search_node.h:411-416) for games[0..n) (device int32, NULL = all games) */
int elfmcts_clear(ElfMcts* m, const int32_t* games, int n, void* stream);
/* TreeSearchT::setRootNodeState (tree_search.h:478-493) for every game; board_ids device int32 or NULL (slot g) */
int elfmcts_set_root(ElfMcts* m, const int32_t* board_ids, void* stream);
/* rng() % 8 draws of the search threads' actors (BoardFeature::RandomShuffle, board_feature.h:74-78), host uint8
 * [num_games][num_threads][d4_window / num_threads]: window t = the coming draws of thread t's mt19937 */
int elfmcts_set_d4(ElfMcts* m, const uint8_t* d4_host, void* stream);
/* NodeT::enhanceExploration (tree_search_node.h:132-155): etas device f32 [num_games][edge_stride] in edge
 * iteration order, Z device f32 [num_games] (= 1e-10 + sum of etas, accumulated in fp32 on the host) */
int elfmcts_dirichlet(ElfMcts* m, const float* etas, const float* Z, float epsilon, void* stream);
/* first half of batch_rollouts (:205-233): num_threads x num_rollouts_per_batch descents per game, then the features of
 * every leaf that needs the net into s_dst (row r at s_dst + r*stride_elems elements of the feature format, rows game-major);
 * counts (device int32[4], 8-byte aligned): [0] <- rows, [1] <- OR of error bits, [2..3] += rows (u64 running total). */
int elfmcts_select(ElfMcts* m, const int32_t* board_ids, void* s_dst, int64_t stride_elems, int32_t* counts, void* stream);
/* second half (:235-259): pi2response + setEvaluation for the n_rows leaves of the last select, then backup.
 * rv (device int64 [n_rows], may be NULL) = the "rv" reply column, checked against required_version.
 * n_rows < 0: use the row count the last select left in its `counts` buffer on the device (no host round trip). */
int elfmcts_expand(ElfMcts* m, const float* pi, int64_t pi_stride_floats, const float* value, const int64_t* rv, int n_rows,
                   void* stream);
/* root edges in the reference's iteration order (what MCTSResultT::addActions walks, tree_search_base.h:237-294);
 * info device int32 [num_games][ELFMCTS_ROOT_WORDS]; the per-edge outputs ([num_games][edge_stride]) may be NULL */
int elfmcts_root(ElfMcts* m, int32_t* info, int32_t* coord, int32_t* visits, float* prior, float* reward, int32_t* child, void* stream);
/* Test / debug service: checks the invariants of every live node record (scoring-order prefix sorted and linked to its child
   nodes, never-followed tail without statistics and by descending prior, visit counts add up).  Synchronises the device.
   out5_host: 0 number of violations, 1 code, 2 game, 3 node, 4 position of one of them.  No counterpart in the reference. */
int elfmcts_validate(ElfMcts* m, int32_t* out5_host);
/* statistics: total number of tree nodes descended through by all rollouts so far (mean depth = this / rollouts); synchronous */
int elfmcts_node_visits(ElfMcts* m, int64_t* out_host);
/* SearchTreeT::treeAdvance (tree_search_node.h:420-436), moves device int32 [num_games] (reference Coords) */
int elfmcts_advance(ElfMcts* m, const int32_t* moves, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Self-play driver: G games advanced in lock-step on one GPU -- the host half of
 *   elfgames/go/common/game_selfplay.cc  GoGameSelfPlay::act (:272-430) and
 *   elf/ai/tree_search/mcts.h            MCTSAI_T::act (:59-81)
 * One elfsp_begin_step / elfsp_end_step pair is one batch of the reference's batch interface
 * (GCWrapper._call, src_py/elf/utils_elf.py:368-414): begin fills "s" rows [0, n_rows), the caller runs the
 * net, end consumes "pi" and "V".  Moves, Dirichlet noise, move sampling, resignation and game restarts
 * happen inside end_step when a search completes.
 * ------------------------------------------------------------------------------------------------ */
#define ELFGO_E_MCTS_BASE (-100) /* status = ELFGO_E_MCTS_BASE - (OR of ELFMCTS_E_* bits) */
typedef struct ElfSelfPlay ElfSelfPlay;

typedef struct ElfSpOptions {
  int32_t board_size;               /* 19 or 9 */
  int32_t num_games;                /* ContextOptions.num_games (games per GPU) */
  int32_t nodes_per_game;           /* node records per tree (multiple of 64) */
  int32_t num_rollouts_per_thread;  /* TSOptions.num_rollouts_per_thread; one search thread per game */
  int32_t persistent_tree;          /* TSOptions.persistent_tree */
  float root_epsilon, root_alpha;   /* TSOptions.root_epsilon / root_alpha */
  uint32_t seed;                    /* GameOptions.seed; per-game rule below */
  int32_t policy_distri_cutoff;     /* GameOptions.policy_distri_cutoff */
  int32_t move_cutoff;              /* GameOptions.move_cutoff */
  float resign_thres;               /* ClientCtrl.{black,white}_resign_thres */
  float never_resign_prob;          /* ClientCtrl.never_resign_prob */
  int32_t log_searches;             /* keep the first N search results for elfsp_search_log (tests) */
  int32_t keep_records;             /* > 0: keep the Record JSON of the last N finished games for elfsp_pop_record */
  int32_t policy_distri_training_for_all; /* GameOptions.policy_distri_training_for_all: record the MCTS policy of every move */
  int32_t model_ver;                /* Record.request.vers.black_ver (the self-play model version; white_ver = -1) */
  int32_t game_idx_base;            /* global index of this context's game 0 in the whole job (rank x games + group offset) */
  uint64_t job_hash;                /* std::hash of ContextOptions.job_id, only used by the seed == 0 rule below */
  ElfMctsOptions mcts;
  /* the second AI of evaluation games (a request with white_ver >= 0): init_ai's overrides, game_selfplay.cc:171-181 */
  float white_puct;                         /* GameOptions.white_puct: > 0 replaces c_puct for the "actor_white" AI */
  int32_t white_mcts_rollout_per_batch;     /* GameOptions.white_mcts_rollout_per_batch: > 0 replaces num_rollouts_per_batch */
  int32_t white_mcts_rollout_per_thread;    /* GameOptions.white_mcts_rollout_per_thread: > 0 replaces num_rollouts_per_thread */
  int32_t black_use_policy_network_only;    /* GameOptions.*_use_policy_network_only: that colour's moves are MCTSAI_T::actPolicyOnly */
  int32_t white_use_policy_network_only;    /*   (elf/ai/tree_search/mcts.h:83-90, game_selfplay.cc:357-372) */
  int32_t pick_method;                      /* TSOptions.pick_method, ELFSP_PICK_* (tree_search.h:495-528) */
  int32_t cheat_eval_new_model_wins_half;   /* GameOptions.cheat_* (finish_game, game_selfplay.cc:122-129) */
  int32_t cheat_selfplay_random_result;
  int32_t following_pass;                   /* GameOptions.following_pass with a human opponent (mcts_update_info :104-111) */
  int32_t reserved1;
} ElfSpOptions;
#define ELFSP_PICK_MOST_VISITED 0
#define ELFSP_PICK_STRONGEST_PRIOR 1
#define ELFSP_PICK_UNIFORM_RANDOM 2   /* the reference draws from a process-wide mt19937 seeded with time(NULL) (tree_search_base.h:238);
                                         here one generator per context, seeded seed ^ 0x5EED, or time(NULL) for seed == 0;
                                         elfsp_set_pick_seed re-seeds it */
#define ELFSP_ACTOR_BLACK 0           /* the AI created as "actor_black" with the request's black_ver */
#define ELFSP_ACTOR_WHITE 1           /* the AI created as "actor_white" with the request's white_ver (evaluation games only) */
/* Per-game RNG seeds.  The reference seeds every GoGameBase with GameOptions.seed itself (common/game_base.h:32-38), so with a
 * non-zero seed and a deterministic net all its games are identical; its time-seeded default (seed == 0) is what production
 * runs use.  Here game g (global index i = game_idx_base + g) is seeded
 *   seed != 0:  seed + i            (the rule of this repository's reference harness, oracle/ref_selfplay.cc: distinct,
 *                                    reproducible games; i is unique over groups and ranks, so no two games of a job collide)
 *   seed == 0:  (unix_seconds*1000 + unix_milliseconds + int32(int32(i ^ job_hash) * 2341479)) % 100000000
 *                                   (elf_utils::get_seed(int), elf/utils/utils.h:50-57, as GoGameBase calls it: the game term is a
 *                                    32-bit int product that wraps; job_hash is any 64-bit hash of the job id -- the reference's
 *                                    std::hash<std::string> through the pybind boundary -- so the seeds are of the same form, and
 *                                    time-based like the reference's, never reproducible)
 * and the MCTS actor seed is the first draw of that generator (game_selfplay.cc:45-47). */

/* what GameNotifierBase::OnMCTSResult (common/notifier.h:13) sees after one search */
typedef struct ElfSpSearch {
  int32_t game, move_played, best_action, total_visits, n_edges;
  float root_value, max_score, predicted_value;
} ElfSpSearch;

int elfsp_create(const ElfSpOptions* opt, int device, const uint64_t* zobrist_host, ElfSelfPlay** out);
int elfsp_destroy(ElfSelfPlay* sp);
ElfGoEngine* elfsp_engine(ElfSelfPlay* sp);
ElfMcts* elfsp_mcts(ElfSelfPlay* sp);
/* requests that restarted SOME games with search options (TSOptions) other than the context's while other games were still playing
 * under the old ones: the tree pools belong to the whole context and are rebuilt only when no game is mid-play, so those restarted
 * games kept the context's options (the reference would build their AIs from the request's, game_selfplay.cc:166-180).  0 in every
 * flow where requests reach all games at a boundary (the reference's own server sends a request to all games of a client). */
int64_t elfsp_ts_requests_deferred(const ElfSelfPlay* sp);
/* games that play NOW under the context's search options although their request carried others (see above); their Records carry the
 * options that were used, not the request's */
int elfsp_ts_games_deferred(const ElfSelfPlay* sp);
int elfsp_max_rows(const ElfSelfPlay* sp);   /* num_games * num_threads * num_rollouts_per_batch */
/* the same bound for one of the two AIs (the "actor_white" AI may have its own batch override) */
int elfsp_max_rows_actor(const ElfSelfPlay* sp, int actor);
/* the tree pool of one AI; NULL until a request has created it (actor 1) */
ElfMcts* elfsp_mcts_actor(ElfSelfPlay* sp, int actor);
/* s_dst: device rows in the context's feature format (elfmcts_set_feature_format: fp32 [18][N][N] or fp16 [N][N][18]), rows
 * stride_elems elements apart.  n_rows != NULL: the call waits for the device and stores the number of rows that need the net.
 * n_rows == NULL: nothing is waited for inside a move -- the row count stays on the device, the net evaluates elfsp_max_rows()
 * rows (surplus rows hold stale features and are ignored) and elfsp_end_step expands exactly the counted ones; errors raised by
 * the kernels surface at the next move boundary.  Host work (Dirichlet draws, move choice, records) happens at move boundaries
 * either way. */
int elfsp_begin_step(ElfSelfPlay* sp, void* s_dst, int64_t stride_elems, int* n_rows, void* stream);
/* pi device f32 [rows][N*N+1] (rows pi_stride_floats apart), value device f32 [rows], rv device int64 [rows] or NULL (the reply's
 * model version, checked against the requested one: elfsp_set_request) */
int elfsp_end_step(ElfSelfPlay* sp, const float* pi, int64_t pi_stride_floats, const float* value, const int64_t* rv, void* stream);
/* rows of the last begin_step (synchronises the stream of that call) */
int elfsp_last_rows(ElfSelfPlay* sp, int* n_rows);
/* The same pair for games with two AIs (a request with white_ver >= 0): index ELFSP_ACTOR_BLACK / ELFSP_ACTOR_WHITE = the batch
 * group the rows belong to ("actor_black" / "actor_white", game_selfplay.cc:165-181), each evaluated by its own model.  s_dst[a]
 * may be NULL while AI a has no game searching (always true for actor 1 under a self-play request); n_rows (may be NULL) <- the
 * two row counts.  A game searches with the AI of the colour to move (player_swap exchanges the two); the AIs may need different
 * numbers of steps per move, so searches of different games are not in step.  elfsp_begin_step / elfsp_end_step are the
 * one-AI forms of these calls and refuse (ELFGO_E_BADARG) a step in which the second AI has rows. */
int elfsp_begin_step2(ElfSelfPlay* sp, void* const* s_dst, int64_t stride_elems, int* n_rows, void* stream);
int elfsp_end_step2(ElfSelfPlay* sp, const float* const* pi, int64_t pi_stride_floats, const float* const* value,
                    const int64_t* const* rv, void* stream);
int elfsp_last_rows2(ElfSelfPlay* sp, int* n_rows);
/* Client::setRequest / GameContext::setRequest (train/distri_client.h:318-331, inference/game_context.h:76-88): a MsgRequest
 * (common/record.h:119-149) for the games.  As in the reference a game looks at its mailbox at the top of every fifth act
 * (game_selfplay.cc:273-289) -- or at once while it is waiting -- and then (GoGameSelfPlay::OnReceive :222-270):
 *   black_ver < 0                      the game waits (ModelPair::wait) until a request gives it something to play;
 *   num_game_thread_used = k >= 0      games k, k+1, ... receive the request as a wait request (DispatcherCallback::OnFirstSend);
 *   new versions / new player_swap / the game was waiting:  restart from the empty board without a record, new AIs seeded with
 *                                      the next draws of the game's generator; white_ver >= 0 creates the second AI for White
 *                                      (own tree pool, ElfSpOptions.white_* overrides); player_swap exchanges the two;
 *   same versions or async             only thresholds (and with async: no version check on replies) change.
 * Restarted games stay idle until every game has received the request; then one "game_start" is due
 * (elfsp_take_game_starts) and they play.  Replies must carry the AI's version in "rv" unless async.
 * Requests queue: each is delivered to all games before the next one goes out (elf/base/dispatcher.h:104-152). */
typedef struct ElfTsOptions {         /* TSOptions + SearchAlgoOptions as they travel in a request (tree_search_options.h:23-75,77-213) */
  int32_t max_num_moves, num_threads, num_rollouts_per_thread, num_rollouts_per_batch;
  int32_t verbose, verbose_time, persistent_tree, pick_method /* ELFSP_PICK_*, -1 = a name the search does not know */;
  int64_t seed;
  float root_epsilon, root_alpha;
  int32_t virtual_loss;
  int32_t use_prior, unexplored_q_zero, root_unexplored_q_zero;
  float c_puct;
  char log_prefix[60];
} ElfTsOptions;
typedef struct ElfSpRequest {
  int64_t black_ver, white_ver;       /* ModelPair */
  float black_resign_thres, white_resign_thres, never_resign_prob;   /* ClientCtrl; the resign check uses the mean of the two */
  int32_t num_game_thread_used;       /* ClientCtrl.num_game_thread_used, -1 = all games */
  int32_t player_swap;                /* ClientCtrl.player_swap */
  int32_t async;                      /* ClientCtrl.async */
  int32_t client_type;                /* ClientCtrl.client_type as the server sent it (record.h:24-29); it only travels into the
                                         Records the games dump.  0 = unset: CLIENT_SELFPLAY_ONLY (1) */
} ElfSpRequest;
int elfsp_set_request2(ElfSelfPlay* sp, const ElfSpRequest* request);
/* The same with the TSOptions the request carries (MsgRequest.vers.mcts_opt).  In the reference the SERVER dictates the search
 * options: GoGameSelfPlay::restart builds its AIs from request.vers.mcts_opt (game_selfplay.cc:166-180), self-play requests carry the
 * server's options (train/ctrl_selfplay.h:426), evaluation requests the same with the Dirichlet noise and both *_q_zero flags
 * switched off (EvalSubCtrl, train/ctrl_eval.h:227-237), and ModelPair::operator== compares them, so a request that differs only
 * in mcts_opt restarts the games as a new model does.  Here: the games restart, wait at the barrier, and when every game has
 * received the request the tree pools are rebuilt for the new options (threads, rollouts per thread / per batch, virtual loss,
 * persistent tree, pick method, root epsilon / alpha, c_puct, use_prior, both q_zero flags); ElfSpOptions.white_* overrides still
 * apply on top for the second AI; an async request changes nothing until the next restart, as in the reference.
 * elfsp_max_rows_actor may change with it: size the row destinations for the largest request you send BEFORE the step that
 * delivers it.  mcts_opt == NULL: the context's options at creation (GameContext::setRequest, inference/game_context.h:83).
 * ELFGO_E_BADARG for options the engine cannot run (threads x rollouts per batch above elfmcts_max_rollouts_per_step, an unknown
 * pick method). */
int elfsp_set_request3(ElfSelfPlay* sp, const ElfSpRequest* request, const ElfTsOptions* mcts_opt);
/* the same with both thresholds equal, every game used, no swap */
int elfsp_set_request(ElfSelfPlay* sp, int64_t black_ver, int64_t white_ver, float resign_thres, float never_resign_prob, int async);
/* Seed of the uniform_random pick generator: MCTSResultT::addActions' `static std::mt19937 rng(time(NULL))`
 * (tree_search_base.h:238) draws random_idx = rng() % edges once per search of that method.  A context that is given the value
 * time(NULL) had in a reference process picks the edges that process picked (one game per context: the reference's game threads
 * race for the generator). */
int elfsp_set_pick_seed(ElfSelfPlay* sp, uint32_t seed);
/* host-only progress counters, no device synchronisation: out6 = {searches finished, games finished, searches open, steps,
 * games waiting for a request, games waiting at a request barrier} */
int elfsp_progress(const ElfSelfPlay* sp, int64_t* out6);
/* which AI of `game` is searching now (ELFSP_ACTOR_*), -1 = between two searches; host-only */
int elfsp_game_actor(const ElfSelfPlay* sp, int game);
/* number of times the games were (re)started by a request since the last call (each is one "game_start" batch of the reference,
 * common/dispatcher_callback.h:86-88); *black_ver / *white_ver <- the versions of the current request */
int elfsp_take_game_starts(ElfSelfPlay* sp, int64_t* black_ver, int64_t* white_ver);
/* out[12]: 0 moves played, 1 games finished, 2 rollouts, 3 net rows, 4 steps, 5 searches logged, 6 steps per move, 7 step in
 * move, 8 tree nodes descended through, 9 wall nanoseconds of move-boundary work (move choice, forward + treeAdvance, game
 * ends, Dirichlet / D4 draws up, root of the next search), 10 move boundaries, 11 wall nanoseconds the boundaries first waited
 * for the stream to drain what the host had queued ahead (pipeline depth, not boundary work).
 * Synchronises the device. */
int elfsp_stats(ElfSelfPlay* sp, int64_t* out);
/* Interactive play (the human_actor half of GoGameSelfPlay::act, game_selfplay.cc:290-330, that the GTP console
 * drives): between two searches, forward externally chosen moves (moves_host[g] = reference Coord, < 0 = none) on the game boards;
 * the trees follow.  Refused moves leave their game untouched and make the call return ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD.
 * A move that completes two consecutive passes finishes that game (finish_game(FR_TWO_PASSES), :319-322). */
int elfsp_play(ElfSelfPlay* sp, const int32_t* moves_host, void* stream);
/* GameOptions.preload_sgf / preload_sgf_move_to (GoGameSelfPlay::restart, game_selfplay.cc:202-219): before the first search,
 * make every game follow the move list moves_host[0..n) (reference Coords): the first move_to moves are forwarded at once, then
 * each search's move is replaced by the next listed move (:392-405); the search that finds the list exhausted finishes the game
 * (FR_MAX_STEP).  An illegal listed move returns ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD ("Preload sgf: move not valid!"). */
int elfsp_preload(ElfSelfPlay* sp, const uint16_t* moves_host, int n, int move_to, void* stream);
/* finish_game(reason) + restart (game_selfplay.cc:121-149) for the listed games between two searches: the game is scored
 * (FR_RESIGN: the side to move loses; every other reason: GoState::evaluate(komi), go_state_ext.h:76-103), leaves its record and
 * starts over from the empty board.  reason = FinishReason of common/go_state_ext.h:24-32. */
#define ELFSP_FR_RESIGN 0
#define ELFSP_FR_TWO_PASSES 1
#define ELFSP_FR_MAX_STEP 2
#define ELFSP_FR_CLEAR 3
#define ELFSP_FR_ILLEGAL 4
int elfsp_finish(ElfSelfPlay* sp, const int32_t* games_host, int n, int reason, void* stream);
/* = elfsp_finish(..., ELFSP_FR_CLEAR, ...) (clear_board) */
int elfsp_restart(ElfSelfPlay* sp, const int32_t* games_host, int n, void* stream);
/* final values of the games finished since the last call, oldest first (what GameNotifier::OnGameEnd feeds to
 * GameStats::feedWinRate, train/distri_client.h:228-240); returns the number stored (<= cap) */
int elfsp_take_finished(ElfSelfPlay* sp, float* out_host, int cap);
/* GoGameSelfPlay::getLastScore: final value of the last finished game of every game slot, host f32 [num_games] */
int elfsp_last_score(const ElfSelfPlay* sp, float* out_host);
/* what the last finished search of every game did: out_host[g] = the Coord it forwarded on the game board (after move sampling /
 * preload substitution), 1 (M_RESIGN) if the engine resigned instead of moving, -1 if no search of that game has finished yet */
int elfsp_last_moves(const ElfSelfPlay* sp, int32_t* out_host);
/* Self-play records: with ElfSpOptions.keep_records > 0 every finished game leaves the Record the reference's
 * GameNotifier::OnGameEnd would send (GoStateExt::dumpRecord go_state_ext.h:131-148, Record::setJsonFields record.h:246-254),
 * as the JSON text nlohmann::json::dump() produces.  elfsp_pop_record copies the oldest pending record (NUL-terminated) into buf
 * and removes it; if cap is too small it returns ELFGO_E_BADSIZE, sets *len to the length needed (without NUL) and keeps the record.
 * *len = 0 when nothing is pending. */
int elfsp_records_pending(const ElfSelfPlay* sp);
int elfsp_pop_record(ElfSelfPlay* sp, char* buf, size_t cap, size_t* len);
/* games finished so far (host counter, no device synchronisation) */
int64_t elfsp_games_finished(const ElfSelfPlay* sp);
/* logged searches [first, first+n): records and root edges (host arrays, [n][edge_stride], may be NULL) */
int elfsp_search_log(const ElfSelfPlay* sp, int first, int n, ElfSpSearch* rec, int32_t* coord, int32_t* visits, float* prior,
                     float* reward);

/* ------------------------------------------------------------------------------------------------
 * Training-side replay loader: the trainer's input pipeline on the device.  Replaces, for a whole batch in
 * one launch, what the reference does on one std::thread per sample (2048 threads at batchsize 2048, start_server.sh:11-12):
 *   elfgames/go/train/game_train.cc        GoGameTrain::act :23-58
 *   elfgames/go/common/go_state_ext.h      GoStateExtOffline::fromRecord / switchRandomMove / switchBeforeMove :248-290
 *   elfgames/go/common/game_feature.h      the "train" extractors :73-145 (s, offline_a, winner, mcts_scores, predicted_value,
 *                                          move_idx, num_move, aug_code, selfplay_ver; schema :159-206)
 * Records live in HBM (`capacity` slots padded to `max_moves` plies; with_policies adds the 441-byte quantised MCTS policy
 * per ply, record.h:180-182).  Sample i of a batch replays in board slot i of engine `e` (capacity >= batch), so the
 * replayed GoState can be inspected with elfgo_info / elfgo_legal_mask afterwards.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ElfReplay ElfReplay;

typedef struct ElfTrainBatch {      /* device pointers, rows = samples; every pointer except s may be NULL */
  void* s;                          /* "s" [n][18][N][N] in s_format (ELFGO_FEAT_*), rows s_stride elements apart */
  int64_t s_stride;
  int32_t s_format;
  int32_t num_future_actions;       /* GameOptions.num_future_actions */
  int64_t* offline_a;               /* [n][num_future_actions]  coord2Action of moves[move_idx + j] */
  float* winner;                    /* [n]  +1 / -1 */
  float* mcts_scores;               /* [n][N*N+1]  normalised recorded policy, or one-hot of the played move */
  float* predicted_value;           /* [n]  Record.result.values[move_idx] */
  int32_t* move_idx;                /* [n]  GoState::getPly() - 1 after the replay */
  int32_t* num_move;                /* [n]  number of moves of the record */
  int32_t* aug_code;                /* [n]  D4 code */
  int64_t* selfplay_ver;            /* [n]  Record.request.vers.black_ver */
} ElfTrainBatch;

/* An ElfReplay is not synchronised: call put / draw / extract from one host thread at a time (the reference's reader thread and its
 * batch consumer are one pipeline here).  Device work is ordered by the streams the caller passes. */
int elftrain_create(ElfGoEngine* e, int capacity, int max_moves, int with_policies, uint32_t seed, ElfReplay** out);
int elftrain_destroy(ElfReplay* r);
int elftrain_capacity(const ElfReplay* r);
int elftrain_max_moves(const ElfReplay* r);
int elftrain_num_records(const ElfReplay* r);
/* GoStateExtOffline::fromRecord for record slot `slot`: moves_host = sgfstr2coords(Record.result.content) (see
 * elfrec_sgfstr_to_coords), reward = Record.result.reward, policies_host u8 [num_policies][(N+2)^2], values_host f32.
 * Host pointers; synchronous. */
int elftrain_put(ElfReplay* r, int slot, const uint16_t* moves_host, int num_moves, float reward, int64_t black_ver,
                 const uint8_t* policies_host, int num_policies, const float* values_host, int num_values);
/* the same, ordered on `stream` instead of the default stream: a put that reuses the slot of an evicted record cannot overtake an
 * extraction queued on that stream.  The host arrays may be reused on return. */
int elftrain_put_async(ElfReplay* r, int slot, const uint16_t* moves_host, int num_moves, float reward, int64_t black_ver,
                       const uint8_t* policies_host, int num_policies, const float* values_host, int num_values, void* stream);
/* on != 0 (the default): elftrain_extract does what switchBeforeMove does (go_state_ext.h:283-290: reset, forward x move_to) and
 * leaves the replayed GoState of sample i in board slot i of the engine (elfgo_info / elfgo_legal_mask / ... can look at it;
 * n <= elfgo_capacity).  on == 0 (the trainer's mode): a sample starts from the record's checkpoint below move_to -- the state
 * after every 16th move is written once per put, by the next extraction, together with the game's superko records -- and
 * forwards at most 15 moves; rows are identical, no board slot is written, n is not limited by the engine's capacity. */
int elftrain_set_keep_states(ElfReplay* r, int on);
/* GoGameTrain::act's draws for n samples with the store's std::mt19937 (seeded at create): record, move_to =
 * rng() % (num_moves - num_future_actions + 1), D4 code = rng() % 8; results into device int32 [n] arrays */
int elftrain_draw(ElfReplay* r, int n, int num_future_actions, int32_t* rec, int32_t* move_to, int32_t* d4, void* stream);
/* Sgf::load + its iterator (sgf/sgf.cc:28-240, sgf/sgf.h:21-46,125-245), host only: the entries of an SGF text as (player, Coord) --
 * player 1 = Black, 2 = White; Coord 0 = pass, 3 = M_INVALID (off the board, e.g. "tt" on 19x19: GoState::forward refuses it) -- with
 * the reference's reading of the format: the first node is the header (SZ, KM, HA, RE), setup stones (AB / AW) are not moves,
 * variations are not nested but follow each other, a backslash hides the next character.  Returns the number of entries (the first
 * `cap` are stored), 0 where Sgf::load returns false.  This is what GameOptions.preload_sgf and the ladder suite are read with. */
typedef struct ElfSgfHeader {
  int32_t size, handi, winner /* Stone: 1 Black, 2 White, 3 none */;
  float komi, win_margin;
} ElfSgfHeader;
int elfrec_sgf_parse(int board_size, const char* text, int32_t* players, uint16_t* coords, int cap, ElfSgfHeader* header);
/* GoStateExt::dumpSgf (go_state_ext.cc:26-82): the SGF text the reference's finish_game writes for a finished game when
 * GameOptions.dump_record_prefix is set (file <prefix>_<game>_<seq>_<B|W>.sgf, go_state_ext.h:48-56): RE[] from the final value
 * ("B+R" / "W+R" for +-1, else the margin), PB / PW ("MCTS", "(policy only)" appended), KM, every move with "C[<n>: PredV: <v>]".
 * moves / values / final_value = Record.result.content (as Coords), .values, .reward.  git_hash / git_staged: the two lines of the
 * opening comment, NULL = this library's version and "0".  Returns the length; out == NULL queries it. */
int64_t elfrec_game_sgf(const ElfSpOptions* opt, const uint16_t* moves, int num_moves, const float* values, int num_values,
                        float final_value, const char* filename, const char* git_hash, const char* git_staged, char* out, size_t cap);
/* the same from the Record text of a finished game (elfsp_pop_record): *name_out <- <prefix>_<thread_id>_<seq>_<B|W>.sgf, the file
 * name GoStateExt::dumpSgf() uses (go_state_ext.h:48-56) */
int64_t elfrec_record_to_sgf(const ElfSpOptions* opt, const char* record_json, const char* prefix, char* name_out, size_t name_cap,
                             char* out, size_t cap);
/* ---- the client's wire formats (train/distri_client.h), host only --------------------------------------------------------------
 * What a self-play client sends to the reference's server: Records = {identity, states, records} (common/record.h:401-470) as
 * GuardedRecords keeps and dumps them (distri_client.h:111-170) -- and what it receives: MsgRequestSeq (record.h:152-171).  Texts
 * are byte-identical to nlohmann::json::dump() of the reference's objects, so its server parses them and its tests read them. */
typedef struct ElfThreadState {       /* ThreadState (record.h:354-380) = GoStateExt::getThreadState (go_state_ext.h:149-157) */
  int32_t thread_id, seq, move_idx, reserved;
  int64_t black, white;
} ElfThreadState;
typedef struct ElfClientRecords ElfClientRecords;
int elfrec_client_create(const char* identity, ElfClientRecords** out);
int elfrec_client_destroy(ElfClientRecords* c);
int elfrec_client_feed(ElfClientRecords* c, const char* record_json);              /* GuardedRecords::feed: one finished game */
int elfrec_client_update_state(ElfClientRecords* c, const ElfThreadState* state);  /* Records::updateState (GameNotifier::OnStateUpdate) */
int elfrec_client_size(const ElfClientRecords* c);                                  /* records waiting */
/* GuardedRecords::dumpAndClear: the message for the server.  Returns the text length; out == NULL only queries it (nothing is
 * cleared); ELFGO_E_BADSIZE when cap <= length. */
int64_t elfrec_client_dump_and_clear(ElfClientRecords* c, char* out, size_t cap);
/* the server's reply: MsgRequestSeq::createFromJson.  *mcts_opt <- the TSOptions the server dictates (GoGameSelfPlay::restart builds
 * its AIs from request.vers.mcts_opt, game_selfplay.cc:166-180: hand both to elfsp_set_request3).  ELFGO_E_BADARG
 * for malformed JSON or a missing mandatory field (the reference throws "... cannot not be found!"). */
int elfrec_parse_request_seq(const char* json_text, ElfSpRequest* request, int64_t* seq, ElfTsOptions* mcts_opt);
/* MsgRequestSeq::dumpJsonString: what the reference's server writes for this request */
int64_t elfrec_request_seq_to_json(const ElfSpRequest* request, const ElfTsOptions* mcts_opt, int64_t seq, char* out, size_t cap);
/* GoStateExt::getThreadState of every game of a self-play context (thread_id = the job-wide game index, seq, move_idx = ply - 1,
 * black / white = the versions of the game's current request): what GameNotifier::OnStateUpdate reports at every fifth act */
int elfsp_thread_states(const ElfSelfPlay* sp, ElfThreadState* out, int capacity);
/* The trainer's replay buffer and GoGameTrain::act's draws, host only: elf::shared::ReaderQueuesT<Record>
 * (elf/distributed/shared_reader.h:165-340) -- num_reader (even) queues, each a deque bounded by queue_max_size, filled as
 * TrainCtrl::OnReceive does with InsertWithParity(record, rng, reward > 0) (train/game_ctrl.h:306-311): games Black won go to the odd
 * queues, the others to the even ones -- and the draws of GoGameTrain::act (train/game_train.cc:23-58) over it.  A record is a
 * handle: the slot of an ElfReplay store the caller keeps the record in.  insert_seed seeds TrainCtrl's generator (0 = time(NULL),
 * the reference's rule). */
typedef struct ElfReaderQueues ElfReaderQueues;
int elfrq_create(int num_reader, int queue_min_size, int queue_max_size, uint32_t insert_seed, ElfReaderQueues** out);
int elfrq_destroy(ElfReaderQueues* q);
/* InsertWithParity: queue 2 * (rng() % (num_reader / 2)) + black_win, push_back, the queue's oldest record dropped when it holds
 * more than queue_max_size (*evicted <- its handle, -1 = none: the caller may reuse that slot).  Returns the queue index. */
int elfrq_insert(ElfReaderQueues* q, int32_t slot, int32_t num_moves, int black_win, int32_t* evicted);
int elfrq_sizes(const ElfReaderQueues* q, int32_t* per_queue);
/* one std::mt19937 per GoGameTrain game thread (GoGameBase, common/game_base.h:32-38): seed != 0 -> thread t gets seed + t (the
 * reference gives all of them `seed`: identical streams); seed == 0 -> elf_utils::get_seed(t ^ job_hash), time-based */
int elfrq_set_threads(ElfReaderQueues* q, int num_threads, int64_t seed, uint64_t job_hash);
/* num_acts acts of GoGameTrain::act, the threads taking turns, 64 states each (kNumState): getSamplerWithParity (queue pair
 * rng() % (nq / 2), the odd queue if uniform_real(0, 1) > even_ratio clamped to [0.45, 0.55]: Black's and White's wins are drawn
 * about equally often), Sampler::sample (rng() % queue size), switchRandomMove (a record shorter than num_future_actions is drawn
 * again; move_to = rng() % (num_moves - num_future_actions + 1)), generateD4Code (rng() % 8).  Host int32 arrays of num_acts * 64:
 * feed them to elftrain_extract.  ELFGO_E_BADARG while a queue holds fewer than queue_min_size records (the reference waits). */
int elfrq_draw(ElfReaderQueues* q, int num_acts, int num_future_actions, int32_t* slot, int32_t* move_to, int32_t* d4);
/* one launch: replay record rec[i] up to move_to[i] (switchBeforeMove), then every extractor of the "train" batch under
 * D4 code d4[i] (NULL = 0).  rec/move_to/d4 device int32 [n]; n <= elfgo_capacity(e) */
int elftrain_extract(ElfReplay* r, const int32_t* rec, const int32_t* move_to, const int32_t* d4, int n, const ElfTrainBatch* out,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Record format helpers (host only, no GPU needed): the SGF move string of Record.result.content and the quantised policy.
 * ------------------------------------------------------------------------------------------------ */
/* coords2sgfstr (sgf/sgf.h:87-95): "(;B[xy];W[xy]...)". Returns the length (without NUL); out may be NULL to query it;
 * ELFGO_E_BADSIZE if cap is too small. */
int elfrec_coords_to_sgfstr(int board_size, const uint16_t* coords, int n, char* out, size_t cap);
/* sgfstr2coords + str2coord (sgf/sgf.h:21-46,97-125): returns the number of moves in the string, stores the first `cap` */
int elfrec_sgfstr_to_coords(int board_size, const char* sgf, uint16_t* out, int cap);
/* Record (record.h:236-262) of one finished self-play game as JSON text, from plain host arrays -- the formatting half of
 * elfsp_pop_record on its own (policies u8 [num_policies][(N+2)^2]).  Returns the length (without NUL); out may be NULL. */
int elfrec_record_to_json(const ElfSpOptions* opt, const uint16_t* moves, int num_moves, const uint8_t* policies, int num_policies,
                          const float* values, int num_values, float reward, int never_resign, int seq, uint64_t thread_id,
                          uint64_t timestamp, char* out, size_t cap);
/* the same under an explicit MsgRequest (evaluation games: white_ver, player_swap, the request's thresholds and thread count);
 * request == NULL: the self-play request of the options, as above */
int elfrec_record_to_json2(const ElfSpOptions* opt, const ElfSpRequest* request, const uint16_t* moves, int num_moves,
                           const uint8_t* policies, int num_policies, const float* values, int num_values, float reward, int never_resign,
                           int seq, uint64_t thread_id, uint64_t timestamp, char* out, size_t cap);
/* GoStateExt::addMCTSPolicy (go_state_ext.h:158-181): out[(N+2)^2] = (unsigned char)(prob / max(prob) * 255) at each coord */
int elfrec_quantise_policy(int board_size, const int32_t* coord, const float* prob, int n, uint8_t* out);

/* ------------------------------------------------------------------------------------------------
 * Glue for the PyTorch-ROCm policy/value net (the net itself stays on PyTorch, BASELINE.json north_star): the
 * convolution's epilogue as ONE pass over the fp16 channels_last activation instead of PyTorch's separate
 * bias-add, residual-add and ReLU kernels (src_py/elfgames/go/df_model3.py:62-110 Block.forward:
 * relu(bn(conv(x))) and relu(bn(conv(h)) + x) with eval BatchNorm folded into the conv).
 *   x[r][c] <- act(x[r][c] + bias[c] + (res ? res[r][c] : 0)),  x/res fp16 [rows][channels] contiguous
 *   (channels % 8 == 0, 16-B aligned), bias fp16 [channels] or NULL, relu != 0 applies max(.,0).
 * fp32 arithmetic, one rounding to fp16. */
int elfnet_bias_act_f16(void* x, const void* bias, const void* res, int64_t rows, int channels, int relu, void* stream);
/* the same pass for a bfloat16 activation (round to nearest even) */
int elfnet_bias_act_bf16(void* x, const void* bias, const void* res, int64_t rows, int channels, int relu, void* stream);

/* convenience for callers without a HIP runtime of their own (tests, the pybind11/cgo/ctypes side); these act on the calling
 * thread's current device unless a device is named */
int elfgo_set_device(int device);
int elfgo_get_device(int* device);
/* free / total HBM of a device in bytes (hipMemGetInfo): what a caller sizes num_games x nodes_per_game against --
 * elfmcts_tree_bytes_per_game() per game and AI */
int elfgo_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* what kind of memory a caller-provided address is: 0 = pageable host (or unknown), 1 = page-locked (pinned) host, 2 = device;
 * *device (may be NULL) <- the owning device for kind 2 */
int elfgo_pointer_kind(const void* p, int* device);
/* 2-D copy between any two of {device, pinned host, pageable host} (hipMemcpyDefault), `rows` rows of width_bytes, row pitches
 * in bytes; asynchronous on `stream` when the host side is pinned */
int elfgo_memcpy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, void* stream);
int elfgo_stream_sync(void* stream);
int elfgo_malloc(void** dptr, size_t bytes);
int elfgo_free(void* dptr);
int elfgo_memcpy_h2d(void* dst, const void* src_host, size_t bytes);
int elfgo_memcpy_d2h(void* dst_host, const void* src, size_t bytes);
const char* elfgo_error_string(int status);
const char* elfgo_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ELF_AMD_H_ */
