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
run fin0 MPN_BN_FUSED_FINALIZE=0
run base X=1
run fin0 MPN_BN_FUSED_FINALIZE=0
run fin128 MPN_BN_FIN_MAX_TILES=128
run fin32 MPN_BN_FIN_MAX_TILES=32
run base X=1
