#!/bin/bash
mkdir -p gpurun_out/miopen_db
export TMPDIR=/tmp
echo "== untuned"; MIOPEN_FIND_ENFORCE=1 timeout 300 python tools/net_tune.py /tmp/db_untuned 2>&1 | grep -v "^MIOpen" | tail -4
echo "== tuning"; timeout 1500 python tools/net_tune.py gpurun_out/miopen_db 2>&1 | grep -v "^MIOpen" | tail -6
echo "== tuned db, fresh process, no search"; MIOPEN_FIND_ENFORCE=1 timeout 300 python tools/net_tune.py gpurun_out/miopen_db 2>&1 | grep -v "^MIOpen" | tail -4
du -sh gpurun_out/miopen_db; ls -la gpurun_out/miopen_db
