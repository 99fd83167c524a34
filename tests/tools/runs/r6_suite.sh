#!/bin/bash
# round 6: the whole GPU suite + smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version" | tail -30 > gpurun_out/r6/r06_suite.txt
tail -6 gpurun_out/r6/r06_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
