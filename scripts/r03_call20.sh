#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 1000 python -m pytest tests -q -m gpu > gpurun_out/t_full20.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/t_full20.log
