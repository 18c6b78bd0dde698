/* A plain-C consumer of include/elf_amd.h: proves the header is valid C11, that the library links from C, and exercises the
 * host-only entry points (no GPU needed).  Built and run by tests/test_abi.py. */
#include <stdio.h>
#include <string.h>

#include "elf_amd.h"

int main(void) {
  if (strncmp(elfgo_version(), "elf_amd", 7) != 0) return 1;
  if (strstr(elfgo_error_string(ELFGO_E_BADARG), "bad argument") == NULL) return 2;
  /* coords2sgfstr / sgfstr2coords round trip on a 19x19 move list: D4 (x=3,y=3), pass, T19 corner (18,18) */
  const uint16_t mv[3] = {(uint16_t)(4 * 21 + 4), 0, (uint16_t)(19 * 21 + 19)};
  char buf[64];
  int n = elfrec_coords_to_sgfstr(19, mv, 3, buf, sizeof(buf));
  if (n != (int)strlen("(;B[dd];W[];B[ss])") || strcmp(buf, "(;B[dd];W[];B[ss])") != 0) return 3;
  uint16_t back[8];
  if (elfrec_sgfstr_to_coords(19, buf, back, 8) != 3 || memcmp(back, mv, sizeof(mv)) != 0) return 4;
  if (elfrec_coords_to_sgfstr(19, mv, 3, buf, 4) != ELFGO_E_BADSIZE) return 5;
  /* addMCTSPolicy quantisation */
  const int32_t coord[2] = {22, 23};
  const float prob[2] = {0.75f, 0.25f};
  uint8_t q[441];
  if (elfrec_quantise_policy(19, coord, prob, 2, q) != 0 || q[22] != 255 || q[23] != 85 || q[0] != 0) return 6;
  /* a Record from plain arrays */
  ElfSpOptions opt;
  memset(&opt, 0, sizeof(opt));
  opt.board_size = 19; opt.num_games = 1; opt.num_rollouts_per_thread = 1600; opt.persistent_tree = 1;
  opt.mcts.num_rollouts_per_batch = 8; opt.mcts.c_puct = 1.5f; opt.mcts.use_prior = 1; opt.mcts.virtual_loss = 1;
  const float values[3] = {0.5f, -0.25f, 0.125f};
  char json[8192];
  n = elfrec_record_to_json(&opt, mv, 3, q, 1, values, 3, -7.5f, 0, 2, 0, 0, json, sizeof(json));
  if (n <= 0 || strstr(json, "\"content\":\"(;B[dd];W[];B[ss])\"") == NULL || strstr(json, "\"reward\":-7.5") == NULL ||
      strstr(json, "\"values\":[0.5,-0.25,0.125]") == NULL || strstr(json, "\"num_move\":3") == NULL) return 7;
  /* argument errors are status codes */
  if (elfgo_destroy(NULL) != ELFGO_E_BADARG || elftrain_destroy(NULL) != ELFGO_E_BADARG || elfsp_play(NULL, NULL, NULL) != ELFGO_E_BADARG) return 8;
  printf("abi_c_check ok\n");
  return 0;
}
