# vocoder A/B: tests + timing with and without the gemm_h2w convolutions (BSG_HG_H2W)
set -e
mkdir -p gpurun_out/voc
if [ -n "$VOC_TESTS" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_hifigan.py tests/test_gpu_r4.py -x -q > gpurun_out/voc/tests.log 2>&1 || { tail -40 gpurun_out/voc/tests.log; exit 1; }
tail -3 gpurun_out/voc/tests.log
fi
cat > /tmp/voc_time.py <<'PY'
import os, sys, time, torch, yaml
from collections import OrderedDict
ROOT = os.environ['GRAFT_REPO_ROOT']; sys.path.insert(0, ROOT)
from bisinger_amd import synth
from bisinger_amd.hifigan import HifiGanGenerator
torch.set_grad_enabled(False)
cfg = yaml.safe_load(open(f'{ROOT}/bisinger_amd/configs/hifigan.yaml'))
voc = HifiGanGenerator(cfg)
spec = OrderedDict((k, tuple(v.shape)) for k, v in voc.state_dict().items())
voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 7).items()})
voc = voc.cuda(); voc.remove_weight_norm()
for B in (1, 16):
    mel = torch.randn(B, 80, 1000, device='cuda')
    for _ in range(3): wav = voc(mel)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): wav = voc(mel)
    torch.cuda.synchronize()
    print(f"H2W={os.environ.get('BSG_HG_H2W','1')} B={B}: {(time.perf_counter()-t0)/20*1e3:.3f} ms  finite={bool(torch.isfinite(wav).all())}")
PY
for v in 0 1 0 1; do BSG_HG_H2W=$v timeout -k 10 200 python /tmp/voc_time.py 2>&1 | grep "H2W="; done
