#!/bin/bash
# round 4, call Z: delivery rate when an instruction's 1 KB is pieces of several rows (conv_igemm's operand tiles: 64 B per row and k-step)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4z; mkdir -p $O
timeout 300 tools/microbench/dma_rate 40 pieces 2>&1 | tee $O/dma_rate_pieces_40k.txt
timeout 300 tools/microbench/dma_rate 16 pieces 2>&1 | tee $O/dma_rate_pieces_16k.txt
