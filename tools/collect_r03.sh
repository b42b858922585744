#!/bin/bash
# round-3 evidence in one call: tools/collect_r03.sh
cd "$GRAFT_REPO_ROOT"
tools/collect_profiles.sh r03 pmc > /dev/null 2>&1
tools/collect_gemm_pmc.sh tools/one_shape_kc.py 15000 4096 128 > /dev/null 2>&1; cp gpurun_out/gemm_pmc.txt gpurun_out/r03/gemm_pmc_sq.txt
tools/prof_stress.sh r03s > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_mkgw; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config mkgw --steps 40 --warmup 10 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/trace.log 2>&1
python3 tools/prof_timeline.py $O/trace $O/timeline.txt; rm -rf $O/trace
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --config mkgw --steps 6 --warmup 2 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --config mkgw --steps 6 --warmup 2 --cpu-seconds 0 --no-kernel-timers --no-extras > $O/pmc_write.log 2>&1
python3 tools/prof_by_shape.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm.txt > /dev/null; rm -rf $O/pmc_fetch $O/pmc_write
python3 bench.py --config mkgw --cpu-seconds 0 2>/dev/null | grep metric > $O/bench.json
python3 bench.py --noise-level step --cpu-seconds 0 2>/dev/null | grep metric > gpurun_out/r03/bench_noise_step.json
ls -la gpurun_out/r03 gpurun_out/r03_mkgw gpurun_out/r03s_uniform gpurun_out/r03s_zipf
