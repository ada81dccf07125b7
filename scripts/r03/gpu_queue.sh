#!/bin/bash
# work queue of the resident kernel: its tests, then queue vs per-stage path over batch sizes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_resident.py -x -q -m gpu --timeout 240 -k "work_queue or larger_than_the_chip or end_like" -s 2>&1 | tail -25 > gpurun_out/queue_tests.log
cat gpurun_out/queue_tests.log
rm -f gpurun_out/queue_sizes.jsonl
timeout 500 python scripts/r03/queue_sizes.py ${QUEUE_SIZES:-32 33 40 64 96 128 256 512} 2>&1 | tail -12
