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
/* GoState::evaluate (go_state.h:194-203): Tromp-Taylor area(black) - area(white) - komi; superko -> +-1 */
int elfgo_evaluate(ElfGoEngine* e, const int32_t* ids, int n, float komi, float* out, void* stream);
/* per-board info records (ELFGO_INFO_WORDS int32 each) */
int elfgo_info(ElfGoEngine* e, const int32_t* ids, int n, int32_t* out, void* stream);
/* stone colour per point (uint8, 0/1/2) and liberties of the group on it (int16, 0 if empty), action order */
int elfgo_export_board(ElfGoEngine* e, const int32_t* ids, int n, uint8_t* colour, int16_t* libs, void* stream);
/* from each slot's current position play uniformly random legal, non-true-eye
 * moves (counter RNG on seeds[i] and ply) until GoState::terminated(); pass when none.
 * out[i] = {hash_lo, hash_hi, ply, steps}.  Whole games run inside one launch, position in LDS. */
int elfgo_playout(ElfGoEngine* e, const int32_t* ids, const uint64_t* seeds, int n, int max_steps,
                  uint32_t* out, void* stream);

/* convenience for callers without a HIP runtime of their own (tests, cgo/ctypes stubs) */
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
