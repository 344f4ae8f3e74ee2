#!/bin/bash
# scripts/micro/conv_split_bench.sh [extra -D flags]: build and run conv_split_bench.hip on the GPU box, plain and with stamps of band 10
cd "$GRAFT_REPO_ROOT" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=40000 -Wno-pass-failed -I what-matters-for-meta-learning_amd/csrc"
hipcc $F "$@" scripts/micro/conv_split_bench.hip -o /tmp/csb && /tmp/csb
hipcc $F "$@" -DC2S_TS=10 scripts/micro/conv_split_bench.hip -o /tmp/csb_ts && /tmp/csb_ts
