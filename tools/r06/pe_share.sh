#!/bin/bash
# bash tools/r06/pe_share.sh "<partner kinds>" [victim args]: the pair embedding's repeat check beside each partner load (tools/r06/pe_share.py);
# a kind may carry environment for the partner alone: eps:ABOPT_CORE32=1
cd "$(dirname "$0")/../.."
kinds=$1; shift
for spec in $kinds; do
  k=${spec%%:*}; e=${spec#*:}; [ "$e" = "$spec" ] && e=X=1
  (env $e timeout 200 python tools/r06/pe_share.py partner --kind $k --seconds ${PARTNER_S:-30} 2>&1 | grep -v amdgpu.ids | tail -2) &
  env ${VICTIM_ENV:-X=1} timeout 200 python tools/r06/pe_share.py victim --tag "[$spec]" "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-7}
  wait
done
