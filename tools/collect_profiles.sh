set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_head -o run -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-config5-leg > $R/gpurun_out/p_head.log 2>&1
echo head done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_c5 -o run -- python3 $R/bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p_c5.log 2>&1
echo c5 done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_marg -o run -- python3 $R/tools/prof_marg.py > $R/gpurun_out/p_marg.log 2>&1
echo marg done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/p_fetch -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg > $R/gpurun_out/p_fetch.log 2>&1
echo fetch done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/p_write -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5-leg > $R/gpurun_out/p_write.log 2>&1
echo write done
