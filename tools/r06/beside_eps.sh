#!/bin/bash
# bash tools/r06/beside_eps.sh <seconds of partner> <command ...>: runs the command while abopt_eps_net_forward loops in a second process (starts it, waits until it is running)
cd "$(dirname "$0")/../.."
S=$1; shift
rm -f /tmp/abopt_partner_ready
(ABOPT_CORE32=1 timeout 600 python tools/r06/pe_share.py partner --kind eps --seconds $S --ready-file /tmp/abopt_partner_ready 2>&1 | grep -v amdgpu.ids | tail -1) &
for i in $(seq 1 300); do [ -f /tmp/abopt_partner_ready ] && break; sleep 1; done
"$@"
wait
