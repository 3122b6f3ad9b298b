#!/bin/bash
# One GPU lease of round 4: runs the job script given as $1 (a file under scripts/jobs/) with the common environment.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$ROOT
mkdir -p $ROOT/gpurun_out
cd $ROOT
bash "$@"
