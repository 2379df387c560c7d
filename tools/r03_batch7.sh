#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for g in 512 1024; do timeout -k 10 300 python tools/rows_ab.py --grid $g --reps 20 NOSIMT NSUB=4 NSUB=8 NSUB=16 NOSIMT NSUB=4 NSUB=8 NSUB=16 2>&1 | grep -v amdgpu.ids; done
for g in 512 1024; do timeout -k 10 300 python tools/rows_ab.py --workload cfg5 --grid $g --reps 10 NOSIMT NSUB=4 NSUB=8 NSUB=16 NOSIMT NSUB=4 2>&1 | grep -v amdgpu.ids; done
