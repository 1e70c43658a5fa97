#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3o; mkdir -p $O
MPN_SIDE_STREAM=0 timeout 600 python tools/shape_report.py > $O/shape_report_serial.txt 2>&1; wc -l $O/shape_report_serial.txt
