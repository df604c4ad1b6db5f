#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 120 ./tools/attn_probe.bin > gpurun_out/r02w_attn_probe3.txt 2>&1; tail -24 gpurun_out/r02w_attn_probe3.txt
