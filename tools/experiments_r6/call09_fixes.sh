#!/bin/bash
mkdir -p gpurun_out/r6c09
O=gpurun_out/r6c09
python -m pytest tests/test_round3_gpu.py::test_virtual_concat_equals_the_materialised_concatenation tests/test_round4_gpu.py::test_bf16_pyramids_heads_and_stem_against_the_rounding_oracle tests/test_nms_ref_gpu.py tests/test_round6_gpu.py tests/test_round2_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
tail -12 $O/tests.log
grep "(ii) keypoint head" gpurun_out/parity_report.txt | tail -2
# pyramid weight-gradient slice fitting: experiments build, 0 = round 5's rounding, 1 = fitted (the production default)
make -s -j8 -C multiposenet/pytorch_amd/csrc experiments > $O/make_exp.log 2>&1
for rep in 1 2; do for m in 0 1; do
  MPN_WGRAD_SEG_FIT=$m python tools/bench_experiments.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events > $O/bench_segfit${m}_$rep.json 2> $O/bench_segfit${m}_$rep.err
  grep -o '"ms_per_step": [0-9.]*' $O/bench_segfit${m}_$rep.json || tail -5 $O/bench_segfit${m}_$rep.err
done; done
