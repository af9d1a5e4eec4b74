#!/bin/bash
# Container side of a GPU call: records HEAD for the runner's manifest, sends `tools/gpu_call.sh <tag> <steps...>` to the box through
# gpurun and keeps gpurun's own report as gpurun_out/<tag>_call.log.
#   tools/gpu_send.sh <timeout-seconds> <tag> <step> [args] [-- <step> ...]
set -o pipefail
cd "$(dirname "$0")/.." || exit 9
T=$1; TAG=$2; shift 2
[ -n "$T" ] && [ -n "$TAG" ] && [ $# -gt 0 ] || { echo "usage: gpu_send.sh <timeout-seconds> <tag> <step> [args] [-- <step> ...]"; exit 9; }
bash -n tools/gpu_call.sh || exit 9
git rev-parse --short HEAD > .git_head 2>/dev/null
mkdir -p gpurun_out
printf -v Q '%q ' "$@"
/usr/local/graft/bin/gpurun --timeout "$T" -- "bash tools/gpu_call.sh $TAG $Q" > "gpurun_out/${TAG}_call.log" 2>&1
rc=$?
tail -${GPU_SEND_TAIL:-60} "gpurun_out/${TAG}_call.log"
exit $rc
