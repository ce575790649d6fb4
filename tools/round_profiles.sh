# Profiles committed under profiles/ (run on the GPU box: bash tools/round_profiles.sh): kernel stats for the default bench and for
# one frame in flight, and the HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) with one frame in flight; the same for C3;
# kernel stats of c1gpu and c5.  python tools/collect_profiles.py <tag> copies the summaries.
cd /tmp; export TMPDIR=/tmp; export J2K_TUNING=1   # (the library reads its J2K_* switches only with this set)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > $O/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflight1 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --inflight 1 > $O/stats_inflight1.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 > $O/pmc_$c.log 2>&1
done
tail -1 $O/stats_default.log | cut -c1-200; tail -1 $O/stats_inflight1.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python $R/tools/bench_c3.py 0 0 > $O/stats_c3.log 2>&1
tail -7 $O/stats_c3.log
export J2K_T1_DEC_SPLIT=1        # the same frame through the plane-stepped decoder (lanes kernels), alone: the latency of its chains
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3lanes -- python $R/tools/bench_c3.py 0 0 > $O/stats_c3lanes.log 2>&1
unset J2K_T1_DEC_SPLIT
tail -7 $O/stats_c3lanes.log
python $R/tools/t1_trace.py $O/stats_c3lanes > $O/c3lanes_per_launch.txt 2>&1; cat $O/c3lanes_per_launch.txt | cut -c1-200
export GPU_MAX_HW_QUEUES=32     # c3 / c1gpu want a hardware queue per frame in flight; under rocprofv3 the profiler's preload initialises the runtime before
                                # bench.py can set it, so it is set HERE; unset again before the C5 line
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3bench -- python $R/bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/stats_c3bench.log 2>&1
tail -1 $O/stats_c3bench.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_c3_$c -- python $R/bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > $O/pmc_c3_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c1gpu -- python $R/bench.py --config c1gpu --steps 3 --warmup 1 --no-cpu-baseline > $O/stats_c1gpu.log 2>&1
tail -1 $O/stats_c1gpu.log | cut -c1-200
unset GPU_MAX_HW_QUEUES
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python $R/bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/stats_c5.log 2>&1
tail -1 $O/stats_c5.log | cut -c1-200
# round 5: the closed-loop codec (4K RGB8, MQ coder: pixels -> tile-parts of packets -> pixels) -- the kernels of the decode body
export GPU_MAX_HW_QUEUES=32
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cl -- python $R/bench.py --config cl --steps 2 --warmup 1 --inflight 2 > $O/stats_cl.log 2>&1
tail -1 $O/stats_cl.log | cut -c1-200
# ... and with the reference's HT coder, one frame in flight: every kernel of pixels -> tile-parts -> pixels at its solo duration
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_clht -- python $R/bench.py --config clht --steps 30 --warmup 3 --inflight 1 > $O/stats_clht.log 2>&1
tail -1 $O/stats_clht.log | cut -c1-200
unset GPU_MAX_HW_QUEUES
# (round 5's XCD-group A/B of the inverse level 0 -- J2K_L0_XCD_GROUP=8|16 under --pmc FETCH_SIZE and under --stats -- was collected once with
#  the loop that stood here, profiles/r05_inv_level0_xcd_group_ab.txt; a later PMC pass of it went silent for seven minutes, so it is not repeated)
