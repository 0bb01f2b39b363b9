#!/bin/bash
# FETCH_SIZE of the 1024-channel kernel per setting "V GPW MAP ALT" (one rocprofv3 --pmc pass each).  Diagnostics.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "${@:-5 5 16 0}"; do
  set -- $cfg
  out=gpurun_out/fetch_v$1_g$2_m$3_a$4
  mkdir -p $out
  V=$1 GPW=$2 MAP=$3 ALT=$4 N=6 rocprofv3 --pmc FETCH_SIZE -d $out --output-format csv -- python3 tools/pfb_one.py > $out/log.txt 2>&1
  echo "V=$1 GPW=$2 MAP=$3 ALT=$4: $(python3 tools/pmc_summary.py $out --match pfb1024 | grep -E 'FETCH_SIZE')  $(grep median $out/log.txt)"
done
