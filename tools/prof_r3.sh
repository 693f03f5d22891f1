set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_head -o run -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-config5-leg > $R/gpurun_out/p_head.log 2>&1
echo head done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_c5 -o run -- python3 $R/bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p_c5.log 2>&1
echo c5 done
cd $R
python3 tools/rocpd_extract.py stats gpurun_out/p_head/run_results.db gpurun_out/p_head_stats.csv
python3 tools/rocpd_extract.py stats gpurun_out/p_c5/run_results.db gpurun_out/p_c5_stats.csv
PLBA_PREP_TIMING=1 python3 tools/prof_e2e.py > gpurun_out/prep_timing.log 2>&1 || true
