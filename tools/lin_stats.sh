#!/bin/bash
# development: per-kernel times of one linearisation (tools/lin_only.py) under rocprofv3 --stats; run through gpurun
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pm0 -- python3 tools/lin_only.py ${S:-16} > gpurun_out/pm0.log 2>&1
python3 tools/summarize_profile.py gpurun_out/pm0_s.txt --stats gpurun_out/pm0 | head -30
grep "linearize ms" gpurun_out/pm0.log
