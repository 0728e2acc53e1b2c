cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05g}
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/${TAG}_test.log 2>&1; echo "tests rc=$?"; tail -6 $O/${TAG}_test.log
for c in c3 c5 c2; do timeout -k 10 300 python scripts/segment_sort_routes.py $c >> $O/${TAG}_segment_sort_routes.txt 2>/dev/null; done; cat $O/${TAG}_segment_sort_routes.txt
for L in 1 3; do WDGS_METRIC_LANES=$L timeout -k 10 300 python scripts/densify_breakdown.py c3 3 > $O/${TAG}_densify_breakdown_metric_lanes${L}.txt 2>/dev/null; echo "== metric lanes $L"; grep -v "launches" $O/${TAG}_densify_breakdown_metric_lanes${L}.txt; done
rm -rf $O/prof_${TAG}_vpr8
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${TAG}_vpr8 -- python3 bench.py --views-per-rank 8 --lanes 3 --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline --no-profile --min-seconds 0.2 > $O/${TAG}_vpr8_under_rocprof.json 2> $O/${TAG}_vpr8_rocprof.err || { tail -5 $O/${TAG}_vpr8_rocprof.err; exit 1; }
python3 scripts/lane_overlap.py $O/prof_${TAG}_vpr8 4000 > $O/${TAG}_lane_overlap_L3.txt; cat $O/${TAG}_lane_overlap_L3.txt
