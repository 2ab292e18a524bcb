#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout -k 5 120 tools/bin/tap_mix | tee gpurun_out/r3_tap_mix.log
