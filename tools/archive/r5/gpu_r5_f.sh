#!/bin/bash
# r5 call f: cfg5 (inference chain incl. the list interface on the same workload), NMS latency incl. the 76 725-candidate worst case,
# cfg2 / cfg4 on the stripped build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout 600 python -m pytest tests/test_prn_assign.py tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "prn or infer or batched" 2>&1 | tail -2
timeout 600 python tools/infer_bench.py --iters 8 2>/dev/null | tee $O/cfg5_infer_bench.txt | cut -c1-230
timeout 300 python tools/nms_microbench.py 2>/dev/null | tee $O/nms_microbench.txt
timeout 400 python bench.py --layers 50 --size 480 --batch 16 --dtype f32 --subnet keypoint_subnet --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg2_bench.json 2> $O/cfg2.err
python -c "import json; d=json.loads(open('$O/cfg2_bench.json').read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))"
timeout 400 python bench.py --size 800 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/cfg4_bench.json 2> $O/cfg4.err
python -c "import json; d=json.loads(open('$O/cfg4_bench.json').read().strip().splitlines()[-1]); print('cfg4', d['value'], d['ms_per_step'])"
