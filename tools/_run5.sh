#!/bin/bash
cd $GRAFT_REPO_ROOT
export DNAGPU_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/libdnagpu_diag.so
for bits in 0 32 64 96; do DNAGPU_DEBUG_SK=$bits timeout -k 5 120 python tools/sk_ablate.py 3e9 31 2>&1 | tail -1 | cut -c1-900; done
