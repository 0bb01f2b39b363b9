#!/bin/bash
# exp_build_run.sh "<EXTRA hipcc flags>" <python script> [args...]: build a scratch copy of the library with EXTRA
# flags (experiment switches) and run a tools/ script against it (WAVEHIP_PKG_DIR).  The shipped .so is not touched.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
EXTRA_FLAGS="$1"; shift
S="$(mktemp -d /tmp/wavehip_exp.XXXXXX)"
cp -r "$ROOT/wavecap-sdr_amd" "$ROOT/include" "$S/"
rm -rf "$S/wavecap-sdr_amd/build" "$S/wavecap-sdr_amd/wavehip/libwavehip.so"
make -C "$S/wavecap-sdr_amd" EXTRA="$EXTRA_FLAGS" -j"$(nproc)" > "$S/build.log" 2>&1 || { tail -20 "$S/build.log"; exit 1; }
WAVEHIP_PKG_DIR="$S/wavecap-sdr_amd" python3 "$@"
rm -rf "$S"
