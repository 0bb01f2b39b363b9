#!/bin/bash
# Clocks / power while the 1024-channel kernel runs back to back (is the box power-limited under this kernel?).  Diagnostics.
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|memory)" | head -8
echo "-- under load"
V=0 GPW=0 N=3000 python3 tools/pfb_one.py > /tmp/one.log 2>&1 &
PID=$!
sleep 1.5
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
cat /tmp/one.log | grep -v amdgpu
