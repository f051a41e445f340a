#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python scripts/mb_ln_bwd_parts.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l_mb_ln_parts.txt
