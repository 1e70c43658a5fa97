#!/bin/bash
# one bench line per environment setting, interleaved with the base line (A/B inside one call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/sweep; mkdir -p $O; : > $O/sweep.txt
run() {
  local tag="$1"; shift
  local ms=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events 2>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_median_hipevent'], d['ms_per_step'], d['host_enqueue_ms_per_step'])")
  echo "$tag $ms" | tee -a $O/sweep.txt
}
run base X=1
run bn8192 MPN_BN_BLOCKS=8192
run base X=1
run bn8192 MPN_BN_BLOCKS=8192
run bn32768 MPN_BN_BLOCKS=32768
run bn6144 MPN_BN_BLOCKS=6144
run base X=1
run bn8192 MPN_BN_BLOCKS=8192
