#!/bin/bash
# Clocks / power while a kernel runs back to back (is the box power-limited under it?).  clock_watch.sh [pfb|stream|nostore].  Diagnostics.
cd $GRAFT_REPO_ROOT
MODE=${1:-pfb}
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ' '; echo
echo "-- under load ($MODE)"
if [ "$MODE" = stream ]; then
python3 - > /tmp/one.log 2>&1 <<'P' &
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]; sys.path[:0] = [ROOT, os.path.join(ROOT, "wavecap-sdr_amd")]
import torch, wavehip
from wavehip import _lib
m = 1 << 28
src = torch.view_as_complex(torch.randn(m, 2, device="cuda")); dst = torch.empty(2 * m, dtype=torch.complex64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(20): _lib.check(_lib.lib.wh_diag_stream_1r2w(src.data_ptr(), dst.data_ptr(), m, _lib.stream_ptr(torch)), "diag")
ev0.record()
for _ in range(3000): _lib.check(_lib.lib.wh_diag_stream_1r2w(src.data_ptr(), dst.data_ptr(), m, _lib.stream_ptr(torch)), "diag")
ev1.record(); torch.cuda.synchronize()
print("stream_1r2w: %.4f ms per 2^28 samples" % (ev0.elapsed_time(ev1) / 3000))
P
else
V=0 GPW=0 N=3000 python3 tools/pfb_one.py > /tmp/one.log 2>&1 &
fi
PID=$!
sleep 1.6
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power \(W\)" | sed 's/=//g; s/GPU\[0\]//g; s/\t//g' | tr '\n' ' '; echo
  sleep 0.5
done
wait $PID
grep -v amdgpu /tmp/one.log
