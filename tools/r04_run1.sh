#!/bin/bash
# round 4, GPU session 1: full gpu suite, bench, MIOpen find-mode sweep of the backbone, glue attribution
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r04a_gpu_tests.txt 2>&1; echo "pytest rc $?"
tail -40 gpurun_out/r04a_gpu_tests.txt
python bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err; echo "bench rc $?"
tail -c 6000 gpurun_out/r04a_bench.json; tail -5 gpurun_out/r04a_bench.err
timeout 300 python tools/diag/backbone_glue.py > gpurun_out/r04a_backbone_glue.txt 2>&1; echo "glue rc $?"; head -60 gpurun_out/r04a_backbone_glue.txt
timeout 1700 python tools/diag/backbone_find_modes.py --modes default,bench,normal,search --db gpurun_out/miopen_db --timeout 1000 > gpurun_out/r04a_find_modes.txt 2>&1
grep -E "^===|backbone_ms" gpurun_out/r04a_find_modes.txt | cut -c1-200
du -sh gpurun_out/miopen_db; find gpurun_out/miopen_db -type f | head -20
