#!/bin/bash
# two copies of tools/r06/determinism.py side by side on one GPU (the situation of the two-rank tests): bash tools/r06/determinism_pair.sh [ENV=..] -- <args>
cd "$(dirname "$0")/../.."
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
(env "${envs[@]}" timeout 300 python tools/r06/determinism.py --tag "A ${envs[*]}" "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-14}) &
env "${envs[@]}" timeout 300 python tools/r06/determinism.py --tag "B ${envs[*]}" "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-14}
wait
