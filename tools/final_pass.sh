set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fin3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo bench done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 $R/bench.py --steps 200 --warmup 20 > $O/stats.log 2>&1
echo stats done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --mode eager --steps 30 --warmup 5 > $O/pmc_f.log 2>&1
echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --mode eager --steps 30 --warmup 5 > $O/pmc_w.log 2>&1
echo write done
python3 $R/tools/pmc_traffic.py $(ls $O/pmc_f/*counter_collection.csv | head -1) $(ls $O/pmc_w/*counter_collection.csv | head -1) > $O/pmc_traffic.txt
timeout -k 10 200 python3 $R/tools/bench_replay_loop.py > $O/replay_loop.json 2> $O/replay_loop.err
HSCN_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 $R/bench.py > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
echo all done
