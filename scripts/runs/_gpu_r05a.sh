cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05a}
timeout -k 10 600 python -m pytest tests/test_gpu_js_host.py tests/test_gpu_napi.py tests/test_gpu_deferred_sh.py tests/test_gpu_viewer.py -x -q > gpurun_out/${TAG}_new.log 2>&1; echo "new tests rc=$?"; tail -25 gpurun_out/${TAG}_new.log
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_test.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/${TAG}_test.log
timeout -k 10 400 python bench.py --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench_c3.err
timeout -k 10 400 node bindings/napi/bench.js --config c3 > gpurun_out/${TAG}_benchjs_c3.json 2> gpurun_out/${TAG}_benchjs_c3.err; echo "bench.js rc=$?"; tail -3 gpurun_out/${TAG}_benchjs_c3.err
python -c "
import json
d=json.load(open('gpurun_out/${TAG}_bench_c3.json'));print('py ',d['value'],d['ms_per_step'],d['ms_per_step_awaiting_every_step'],d['timed_blocks'])
j=json.load(open('gpurun_out/${TAG}_benchjs_c3.json'));print('js ',j['value'],j['ms_per_step'],j['ms_per_step_awaiting_every_step'],j['timed_blocks'],j['scene_generation_s'])
print(d['kernel_ms_per_view']);print(j['kernel_ms_per_step'])"
