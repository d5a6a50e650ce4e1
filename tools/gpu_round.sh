#!/bin/bash
# tools/gpu_round.sh -- one GPU-box session: smoke, the GPU test suite, then whatever A/B commands follow as arguments (each a
# quoted shell command).  A step that is killed at its limit (rc >= 124) ends the session: no GPU step is started after a hang.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
step() { # name limit command...
  local name=$1 lim=$2; shift 2
  echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$lim" bash -c "$*" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc $((SECONDS - t0))s"; tail -n ${TAIL:-6} "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "killed at its limit: stopping"; exit $rc; fi
  return $rc
}
step smoke 300 "python -c 'import __graft_entry__ as g; g.smoke()'" || exit 1
if [ -z "$SKIP_TESTS" ]; then step gpu_tests 1000 "python -m pytest tests -m gpu -x -q" ; fi
i=0
for cmd in "$@"; do i=$((i+1)); step "cmd$i" ${LIMIT:-600} "$cmd"; done
exit 0
