#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/t11.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/t11.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
