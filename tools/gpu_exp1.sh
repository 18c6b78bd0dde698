#!/bin/bash
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"
hipcc $F tools/playout_phases.hip -o /tmp/pp_base 2>/dev/null && /tmp/pp_base 4096 | tail -11 | head -1
hipcc $F -DELF_EXPERIMENT_NO_SK_STORE tools/playout_phases.hip -o /tmp/pp_nost 2>/dev/null && /tmp/pp_nost 4096 | tail -11
