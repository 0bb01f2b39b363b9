cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "5 3 0" "5 3 16" "5 5 16" "5 8 16" "5 8 64" "5 16 16"; do
  set -- $cfg
  out=gpurun_out/fetch_v$1_g$2_m$3
  mkdir -p $out
  V=$1 GPW=$2 MAP=$3 N=6 rocprofv3 --pmc FETCH_SIZE -d $out --output-format csv -- python3 tools/pfb_one.py > $out/log.txt 2>&1
  python3 tools/pmc_summary.py $out --match pfb1024 | grep -E "FETCH_SIZE"
  tail -1 $out/log.txt
done
