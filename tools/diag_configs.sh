#!/bin/bash
# scratch script for one-off GPU experiments (rewritten per experiment; see profiles/README.md for kept results)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
