# round 5 profile collection (run on the GPU box through gpurun): kernel-trace statistics and the PMC passes for configs[2] (the headline)
# and configs[4] (one GPU), the marginalization step, the 12-keyframe window, the FETCH_SIZE calibration; summaries go to gpurun_out/r05_*
# and are copied into profiles/ afterwards.  (rocprofv3: the program itself after `--`, counters in runs of their own.)
# New in round 5: a THIRD counter pass per workload — the L2's read requests by size — from which the fetched bytes are exact
# (tools/rocpd_extract.py), and tools/calib_fetch.py (known byte counts) under FETCH_SIZE and under the request-size counters.
set -e
R=$GRAFT_REPO_ROOT
python3 $R/tools/calib_fetch.py > $R/gpurun_out/calib_build.log 2>&1      # build the calibration kernels outside the profiler
cd /tmp && export TMPDIR=/tmp
REQ="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1; echo "$name done"; }
run p_head --kernel-trace --stats -d $R/gpurun_out/p_head -o run -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-config5-leg
run p_c5 --kernel-trace --stats -d $R/gpurun_out/p_c5 -o run -- python3 $R/bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline
run p_marg --kernel-trace --stats -d $R/gpurun_out/p_marg -o run -- python3 $R/tools/prof_marg.py
run p_fetch --pmc FETCH_SIZE -d $R/gpurun_out/p_fetch -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg
run p_write --pmc WRITE_SIZE -d $R/gpurun_out/p_write -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg
run p_req --pmc $REQ -d $R/gpurun_out/p_req -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg
run p_fetch5 --pmc FETCH_SIZE -d $R/gpurun_out/p_fetch5 -o run -- python3 $R/bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
run p_write5 --pmc WRITE_SIZE -d $R/gpurun_out/p_write5 -o run -- python3 $R/bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
run p_req5 --pmc $REQ -d $R/gpurun_out/p_req5 -o run -- python3 $R/bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
run p_calf --pmc FETCH_SIZE -d $R/gpurun_out/p_calf -o run -- python3 $R/tools/calib_fetch.py
run p_calr --pmc $REQ -d $R/gpurun_out/p_calr -o run -- python3 $R/tools/calib_fetch.py
cd $R
python3 tools/rocpd_extract.py stats gpurun_out/p_head/run_results.db gpurun_out/r05_kernel_stats.csv
python3 tools/rocpd_extract.py stats gpurun_out/p_c5/run_results.db gpurun_out/r05_config5_kernel_stats.csv
python3 tools/rocpd_extract.py stats gpurun_out/p_marg/run_results.db gpurun_out/r05_marg_kernel_stats.csv
python3 tools/rocpd_extract.py pmc gpurun_out/p_fetch/run_results.db gpurun_out/p_write/run_results.db gpurun_out/r05_pmc_traffic.json gpurun_out/p_head.log "rocprofv3 --pmc <counters> -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg" gpurun_out/p_req/run_results.db
python3 tools/rocpd_extract.py pmc gpurun_out/p_fetch5/run_results.db gpurun_out/p_write5/run_results.db gpurun_out/r05_config5_pmc_traffic.json gpurun_out/p_c5.log "rocprofv3 --pmc <counters> -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline" gpurun_out/p_req5/run_results.db
python3 tools/rocpd_extract.py calib gpurun_out/p_calf/run_results.db gpurun_out/p_calr/run_results.db gpurun_out/r05_fetch_calibration.json
grep "^{" gpurun_out/p_head.log | tail -1 > gpurun_out/r05_bench_profiled.json
grep "^{" gpurun_out/p_c5.log | tail -1 > gpurun_out/r05_config5_bench_profiled.json
# the reference-shaped 12-keyframe window on both landmark paths (tools/prof_realistic.py)
cd /tmp
run p_real0 --kernel-trace --stats -d $R/gpurun_out/p_real0 -o run -- python3 $R/tools/prof_realistic.py 0
run p_real2 --kernel-trace --stats -d $R/gpurun_out/p_real2 -o run -- python3 $R/tools/prof_realistic.py 2
cd $R
python3 tools/rocpd_extract.py stats gpurun_out/p_real0/run_results.db gpurun_out/r05_realistic_record_kernel_stats.csv
python3 tools/rocpd_extract.py stats gpurun_out/p_real2/run_results.db gpurun_out/r05_realistic_fused_kernel_stats.csv
