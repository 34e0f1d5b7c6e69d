#!/bin/bash
# GPU box, round 6: weight-stationary hop micro-benchmark (both protocols), the parity control, the full GPU suite.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06
mkdir -p $O
(timeout 180 tools/microtests/bin/ws_hop 128 536 > $O/ws_hop.txt 2>&1; echo "rc=$?" >> $O/ws_hop.txt)
(timeout 1800 python tools/parity_control.py --lib lnorder=trafficbots_amd/lib/libtrafficbots_hip_dbg_lnorder.so --json $O/parity_control.json > $O/parity_control.txt 2>&1; echo "rc=$?" >> $O/parity_control.txt)
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests_4.txt 2>&1; echo "rc=$?" >> $O/gputests_4.txt)
tail -n 4 $O/ws_hop.txt $O/gputests_4.txt
