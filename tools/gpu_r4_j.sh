#!/bin/bash
# GPU visit: forward with the side-test read in the neighbours' round trip (ELF_FWD_EARLY): A/B on k_playout, then every board / train / MCTS test
TAG=${1:-r04j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
for b in pl_base pl_early pl_base pl_early; do timeout 120 build/$b 4096 19; done
for b in pl_base pl_early; do timeout 120 build/$b 65536 9; timeout 120 build/$b 16384 19; done
} 2>&1 | tee $OUT/playout_ab.txt
timeout 1200 python -m pytest tests/test_gpu_board.py tests/test_gpu_train.py tests/test_gpu_mcts.py -m gpu -q --timeout 300 --tb=short -rf -x > $OUT/pytest.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/pytest.log
