#!/bin/bash
# round 4, session ah: cost of the 4-byte flag-reset memset node inside a replayed graph
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 200 python tools/memset_node_probe.py 2> gpurun_out/r04ah.err; echo "rc=$?"
