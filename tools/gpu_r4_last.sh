#!/bin/bash
# Last GPU visit of the round: PMC passes of the final kernels, then the full validation (tests, smoke, search-only lines, default bench).
bash tools/gpu_r4_pmc.sh r04r 2>&1 | grep -E "^==|k_playout|k_replay_extract|k_mcts_select|k_extract" | head -40
bash tools/gpu_r4_final.sh r04z
