set -o pipefail
mkdir -p gpurun_out/r3x
python -m pytest tests -m gpu -x -q > gpurun_out/r3x/pytest.log 2>&1; rc=$?; tail -n 3 gpurun_out/r3x/pytest.log; [ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r3x/bench.json 2> gpurun_out/r3x/bench.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r3x/bench.json')); print('bf16', d['value'], d['ms_per_step'], d['roofline']['frac'], d['power']); a=d['also']['configs4_fp8_per_gpu']; print('fp8', a['value'], a['ms_per_step'], a['roofline']['frac'], a['power']); print('bucketed', d['also']['configs3_bucketed']['value']); print({k:v for k,v in d['config'].items() if 'logit' in k or 'latent' in k or 'within' in k})"
timeout -k 10 400 python tools/soak.py 200 16 11=1 > gpurun_out/r3x/soak_fp8.log 2>&1; tail -n 2 gpurun_out/r3x/soak_fp8.log
timeout -k 10 400 python tools/soak.py 100 16 > gpurun_out/r3x/soak_bf16.log 2>&1; tail -n 2 gpurun_out/r3x/soak_bf16.log
