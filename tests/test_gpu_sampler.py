"""GPU parity: DDPM / PLMS loops in libbisinger_hip against the goldens produced by the reference."""
import numpy as np
import pytest
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams
from oracle import diffnet as odn, diffusion as odf
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


@pytest.fixture(scope='module')
def model():
    use_config()
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    m = GaussianDiffusion(_Enc(), 80, DIFF_DECODERS['wavenet'](hparams), timesteps=100, K_step=100,
                          spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    load_formula_weights(m, 0, synth.DIFFNET_GAIN)
    return m.cuda()


def _cond():
    rs = np.random.RandomState(11)
    rs.standard_normal((2, 1, 80, 64))
    return T_(rs.standard_normal((2, 256, 64)).astype(np.float32))


def test_schedule_buffers_bit_exact(model, gold):
    g = gold('schedules')
    for k in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod', 'sqrt_one_minus_alphas_cumprod',
              'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod',
              'posterior_variance', 'posterior_log_variance_clipped', 'posterior_mean_coef1', 'posterior_mean_coef2'):
        assert np.array_equal(getattr(model, k).cpu().numpy(), g[f'lin100_006.{k}']), k
    assert np.array_equal(model.spec_min.cpu().numpy(), g['spec_min'])


def test_p_sample_and_trajectory_golden(model, gold):
    g = gold('sampler')
    cond = _cond().cuda()
    noise = synth.synth_noise(100, 2, 80, 64, seed=1)
    x1 = model.p_sample(T_(noise[0][:, None]).cuda(), torch.full((2,), 99, dtype=torch.long).cuda(), cond,
                        noise=T_(noise[1]).cuda())
    assert maxabs(x1, g['p_sample_t99']) <= 5e-5
    # intermediate checkpoints of the trajectory, then the end point: the 1e-3 bar of BASELINE.json
    x = T_(noise[0][:, None]).cuda().contiguous()
    nz = T_(noise[1:]).cuda()
    model.K_step = 100
    x = model.sample(cond, x, noise=nz[:10], n_steps=10)          # t = 99..90
    assert maxabs(x, g['x_t90']) <= 2e-4
    model.K_step = 90
    x = model.sample(cond, x, noise=nz[10:50], n_steps=40)        # t = 89..50
    assert maxabs(x, g['x_t50']) <= 5e-4
    model.K_step = 50
    x = model.sample(cond, x, noise=nz[50:])                      # t = 49..0
    model.K_step = 100
    assert maxabs(x, g['x_t0']) <= 1e-3 / 2.2                     # normalised units; denorm scales by <= 3


def test_plms_vs_oracle(model):
    """Shipped sampler (pndm_speedup) with batched semantics; oracle = same algorithm on the CPU."""
    sd = cpu_sd(model)
    cond = _cond()
    noise = synth.synth_noise(1, 2, 80, 64, seed=9)
    x0 = T_(noise[0][:, None])
    den = lambda x_, t_: odn.diffnet_forward(sd, x_, t_, cond, 'denoise_fn.')
    sch = odf.make_schedule(100, 'linear', 0.06)
    want = odf.plms_sample(sch, den, x0, 100, 5)
    hparams['pndm_speedup'] = 5
    try:
        got = model.sample(cond.cuda(), x0.cuda().contiguous())
    finally:
        hparams['pndm_speedup'] = 0
    assert maxabs(got, want) <= 5e-4


def test_philox_mode_is_reproducible_and_shard_invariant(model):
    cond = torch.randn(4, 256, 64, generator=torch.Generator().manual_seed(3)).cuda()
    def run(rows, row0):
        x = model.philox_normal((len(rows), 1, 80, 64), 'cuda', 77, 0, row0 * 80 * 64)
        return model.sample(cond[rows].contiguous(), x, seed=77, row0=row0, B_total=4, n_steps=5)
    full = run([0, 1, 2, 3], 0)
    again = run([0, 1, 2, 3], 0)
    assert torch.equal(full, again)
    half = run([2, 3], 2)
    assert torch.equal(full[2:], half)


def test_fused_step_tail_matches_separate_kernels(model, monkeypatch):
    """The DDPM loop's fused tail (skip-proj -> out-proj -> p_sample -> next in-proj) against the separate
    GEMM + sampler launches, with supplied noise and with the Philox stream (which must be the same stream)."""
    B, T = 3, 77          # partial tile, scalar staging path
    cond = torch.randn(B, 256, T, generator=torch.Generator().manual_seed(5)).cuda()
    noise = T_(synth.synth_noise(12, B, 80, T, seed=2)).cuda()
    x0 = noise[0][:, None].contiguous()
    fused = model.sample(cond, x0.clone(), noise=noise[1:], n_steps=12)
    fused_p = model.sample(cond, x0.clone(), seed=9, n_steps=12)
    monkeypatch.setenv('BSG_NO_FUSED_TAIL', '1')
    sep = model.sample(cond, x0.clone(), noise=noise[1:], n_steps=12)
    sep_p = model.sample(cond, x0.clone(), seed=9, n_steps=12)
    assert maxabs(fused, sep) <= 2e-5
    assert maxabs(fused_p, sep_p) <= 2e-5
    assert maxabs(fused_p, fused) > 1e-2


def test_fused_plms_tail_matches_separate_kernels(model, monkeypatch):
    """PLMS loop: from the second iteration on the tail is fused (skip-proj -> out-proj -> eps stored to the history ->
    multistep update -> next in-proj); against the separate GEMM launches + plms_step_kernel.  Partial tile, B > 1."""
    B, T = 3, 77
    cond = torch.randn(B, 256, T, generator=torch.Generator().manual_seed(6)).cuda()
    x0 = T_(synth.synth_noise(1, B, 80, T, seed=4))[0][:, None].contiguous().cuda()
    hparams['pndm_speedup'] = 5
    try:
        fused = model.sample(cond, x0.clone()).clone()
        monkeypatch.setenv('BSG_NO_FUSED_TAIL', '1')
        sep = model.sample(cond, x0.clone()).clone()
    finally:
        hparams['pndm_speedup'] = 0
    assert bool(torch.isfinite(fused).all())
    assert maxabs(fused, sep) <= 2e-5


F23 = {'BSG_WINO': '1', 'BSG_H2': '0'}
F43 = {'BSG_WINO': '2', 'BSG_H2': '0'}
H2 = {'BSG_WINO': '2', 'BSG_H2': '1', 'BSG_H2_Q': '0'}    # 32-row matrix tiles (residual_stack_h2_kernel)
HQ = dict(H2, BSG_H2_Q='1')                                # 16-row matrix tiles (residual_stack_q_kernel, diffnet_h2q.hip: round 5's default)


_STACK_CASES = [
    # the F(4,3) stack launch (the default for launches that fill the chip) against per-layer F(2,3) launches: another rounding
    (16, 1000, F23, F43, 'stack_f43', 1e-5),
    (32, 997, F23, F43, 'stack_f43', 1e-5),                              # two launch groups of whole rows, T % 4 != 0, partial last tile
    (3, 77, F23, dict(F43, BSG_STACK43='2'), 'stack_f43', 1e-5),         # forced: a few tiles, partial tile
    (5, 333, F23, dict(F43, BSG_STACK43='2'), 'stack_f43', 1e-5),
    (2, 31, F23, dict(F43, BSG_STACK43='2'), 'stack_f43', 1e-5),         # T < one tile
    # the split-fp16 stack launch (diffnet_h2.hip, the default for launch groups that are at least half full): the direct K=768 form with
    # every fp32 operand as hi + lo fp16 terms on the 16-bit matrix pipe — a third rounding of the same sums
    (16, 1000, F23, H2, 'stack_h2', 1e-5),                               # 64-frame tiles: 256 workgroups
    (32, 997, F23, H2, 'stack_h2', 1e-5),
    (8, 1000, F23, dict(H2, BSG_H2_PAIR64='0'), 'stack_h2', 1e-5),       # 32-frame tiles (the 64-frame ones would fill half of the CUs)
    (16, 1000, F23, dict(H2, BSG_H2_NCT='1'), 'stack_h2', 1e-5),         # forced 32-frame tiles: two launch groups of 8 rows
    (3, 77, F23, dict(H2, BSG_H2='2', BSG_H2_PART='0'), 'stack_h2', 1e-5),   # forced for a few tiles (32-frame), partial tile
    (5, 333, F23, dict(H2, BSG_H2='2', BSG_H2_PART='0'), 'stack_h2', 1e-5),
    (2, 31, F23, dict(H2, BSG_H2='2', BSG_H2_PART='0'), 'stack_h2', 1e-5),   # T < one tile
    # the PART forms of the split-fp16 launch (four workgroups on four CUs per tile, each a quarter of the channels, 16-row matrix tiles; the
    # default while four workgroups per tile fit the chip): the same sums with GEMM1's k-steps in another order.  Quads of 32-frame tiles:
    (1, 1000, F23, H2, 'stack_h2_quad', 1e-5),                           # a single utterance: 32 tiles, 128 workgroups
    (2, 1000, F23, H2, 'stack_h2_quad', 1e-5),                           # 64 tiles: every CU holds a quarter of a tile
    (3, 77, F23, dict(H2, BSG_H2='2'), 'stack_h2_quad', 1e-5),           # partial tile, rows of 3 tiles
    (5, 333, F23, dict(H2, BSG_H2='2'), 'stack_h2_quad', 1e-5),
    (2, 31, F23, dict(H2, BSG_H2='2'), 'stack_h2_quad', 1e-5),           # T < one tile: no neighbours at all
    # quads of 64-frame tiles (B = 3, 4 at T = 1000)
    (4, 1000, F23, H2, 'stack_h2_quad64', 1e-5),                         # 64 tiles: every CU holds a quarter of a tile
    (3, 77, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0'), 'stack_h2_quad64', 1e-5),   # partial tiles, rows of 2 tiles
    (5, 333, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0'), 'stack_h2_quad64', 1e-5),
    (2, 31, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0'), 'stack_h2_quad64', 1e-5),   # T < one tile
    # odd corners of the part forms: a single frame, fewer frames than a column tile, one frame more than a tile, a long single utterance
    (1, 1, F23, dict(H2, BSG_H2='2'), 'stack_h2_quad', 1e-5),
    (2, 17, F23, dict(H2, BSG_H2='2'), 'stack_h2_quad', 1e-5),
    (3, 65, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0'), 'stack_h2_quad64', 1e-5),
    (7, 129, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0', BSG_H2_QUAD64='0'), 'stack_h2_pair64', 1e-5),
    (1, 2500, F23, H2, 'stack_h2_quad64', 1e-5),                         # 79 tiles of 32 frames do not fit as quads: 40 tiles of 64 do
    # pairs of 64-frame tiles, 8 waves per workgroup (B = 5 .. 8 at T = 1000)
    (8, 1000, F23, H2, 'stack_h2_pair64', 1e-5),                         # 128 tiles: every CU holds a half of a tile
    (3, 77, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0', BSG_H2_QUAD64='0'), 'stack_h2_pair64', 1e-5),
    (5, 333, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0', BSG_H2_QUAD64='0'), 'stack_h2_pair64', 1e-5),
    (2, 31, F23, dict(H2, BSG_H2='2', BSG_H2_QUAD='0', BSG_H2_QUAD64='0'), 'stack_h2_pair64', 1e-5),
    (3, 77, F23, dict(H2, BSG_H2='2', BSG_H2_NCT='2'), 'stack_h2', 1e-5),   # the same with 64-frame tiles
    (2, 31, F23, dict(H2, BSG_H2='2', BSG_H2_NCT='2'), 'stack_h2', 1e-5),
    # the same launch on 16-row matrix tiles (diffnet_h2q.hip): another MFMA shape, the same products in another accumulation order
    (16, 1000, F23, HQ, 'stack_h2q', 1e-5),                                # 64-frame tiles: 256 workgroups
    (32, 997, F23, HQ, 'stack_h2q', 1e-5),                                 # two launch groups, T % 4 != 0, partial last tile
    (8, 1000, F23, dict(HQ, BSG_H2_PAIR64='0'), 'stack_h2q', 1e-5),        # 32-frame tiles
    (16, 1000, F23, dict(HQ, BSG_H2_NCT='1'), 'stack_h2q', 1e-5),          # forced 32-frame tiles: two launch groups of 8 rows
    (3, 77, F23, dict(HQ, BSG_H2='2', BSG_H2_PART='0'), 'stack_h2q', 1e-5),   # forced for a few tiles (32-frame), partial tile
    (5, 333, F23, dict(HQ, BSG_H2='2', BSG_H2_PART='0'), 'stack_h2q', 1e-5),
    (2, 31, F23, dict(HQ, BSG_H2='2', BSG_H2_PART='0'), 'stack_h2q', 1e-5),   # T < one tile
    (1, 1, F23, dict(HQ, BSG_H2='2', BSG_H2_PART='0'), 'stack_h2q', 1e-5),
    (3, 77, F23, dict(HQ, BSG_H2='2', BSG_H2_NCT='2'), 'stack_h2q', 1e-5),    # the same with 64-frame tiles
    (2, 31, F23, dict(HQ, BSG_H2='2', BSG_H2_NCT='2'), 'stack_h2q', 1e-5),
    (3, 65, F23, dict(HQ, BSG_H2='2', BSG_H2_NCT='2'), 'stack_h2q', 1e-5),    # one frame into the second tile
]

# One child process per DISTINCT environment (the switches are read once per process), which runs every shape the list asks of that
# environment — not one child per (case, form): 86 interpreter + HIP start-ups of ~2.8 s were 240 s of the GPU suite.
# A process that runs several shapes on one handle after another is also what a server does.
def _env_key(env):
    return tuple(sorted(env.items()))


_STACK_GROUPS = {}
for _B, _T, _base, _stack, _path, _tol in _STACK_CASES:
    for _env in (_base, _stack):
        _shapes = _STACK_GROUPS.setdefault(_env_key(_env), [])
        if (_B, _T) not in _shapes:
            _shapes.append((_B, _T))
_STACK_RESULTS = {}

_STACK_CHILD = r"""
import sys, json, hashlib, torch, numpy as np
sys.path.insert(0, %r)
from tests.util import load_formula_weights, use_config
from bisinger_amd import synth
from bisinger_amd.hparams import hparams
torch.set_grad_enabled(False)
use_config()
from bisinger_amd.diffnet import DIFF_DECODERS
from bisinger_amd.diffusion import GaussianDiffusion
class E:
    def __len__(self): return 65
    def pad(self): return 0
m = GaussianDiffusion(E(), 80, DIFF_DECODERS['wavenet'](hparams), timesteps=100, K_step=100, spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
load_formula_weights(m, 0, synth.DIFFNET_GAIN); m = m.cuda()
out = {}
for B, T in json.loads(sys.argv[2]):
    g = torch.Generator().manual_seed(3)
    cond = torch.randn(B, 256, T, generator=g).cuda()
    hs = []
    for rep in range(3):
        x = m.philox_normal((B, 1, 80, T), 'cuda', 5, 0, 0)
        x = m.sample(cond, x, seed=5, n_steps=10)
        torch.cuda.synchronize()
        hs.append(hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest())
    path_s = m.denoise_fn.last_path()
    t = torch.arange(B, device='cuda') * 7 %% 100
    eps = m.denoise_fn(x, t, cond)          # per-row timesteps through the same launch form
    np.save('%%s/%%d_%%d.npy' %% (sys.argv[1], B, T), torch.cat([x, eps]).cpu().numpy())
    out['%%d_%%d' %% (B, T)] = {'hashes': hs, 'timeouts': m.denoise_fn.handoff_timeouts(), 'finite': bool(torch.isfinite(x).all()),
                               'path': m.denoise_fn.last_path(), 'path_sampler': path_s}
print(json.dumps(out))
"""


def _stack_child(env, B, T, tmp_path_factory):
    """(record, array) of shape (B, T) under `env`; the first request of an environment runs ALL its shapes in one child."""
    import json
    import os
    import subprocess
    import sys
    key = _env_key(env)
    if (key, B, T) not in _STACK_RESULTS:
        d = tmp_path_factory.mktemp('stack')
        code = _STACK_CHILD % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        shapes = _STACK_GROUPS[key]
        out = subprocess.run([sys.executable, '-c', code, str(d), json.dumps(shapes)], env=dict(os.environ, **env), capture_output=True, text=True,
                             timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        recs = json.loads(out.stdout.strip().splitlines()[-1])
        for b, t in shapes:
            _STACK_RESULTS[(key, b, t)] = (recs[f'{b}_{t}'], np.load(os.path.join(str(d), f'{b}_{t}.npy')))
    return _STACK_RESULTS[(key, B, T)]


@pytest.mark.parametrize('B,T,base,stack,path,tol', _STACK_CASES)
def test_stack_launch_matches_per_layer_launches(B, T, base, stack, path, tol, tmp_path_factory):
    """The on-chip stack launches (all 20 layers in one launch, x on chip, neighbour tiles exchanging 8-frame edges every
    layer) against one launch per layer: bit-identical from run to run, no hand-off give-ups, agreement to rounding after 10
    sampler steps + one evaluation with per-row timesteps.  F(4,3) stack (diffnet_f43.hip, the default at these
    sizes): Winograd F(4,3) transforms instead of F(2,3) and x recovered as image - d: measured 7e-7, bar 1e-5.  One child process per
    environment (the switches are read once per process; every shape of that environment in the same child).  BSG_STACK43=2 / BSG_H2=2 force
    the form for small / ragged shapes.
    Split-fp16 stack (diffnet_h2.hip): products exact to 3 x 2^-24, fp32 accumulation in another order: bar 1e-5."""
    rb, ab = _stack_child(base, B, T, tmp_path_factory)
    rs_, as_ = _stack_child(stack, B, T, tmp_path_factory)
    assert rs_['path'] == path and not rb['path'].startswith('stack')
    assert rs_['timeouts'] == 0 and rs_['finite']
    assert len(set(rb['hashes'])) == 1 and len(set(rs_['hashes'])) == 1
    dev = float(np.abs(ab - as_).max())
    print(f'{path} launch vs per-layer launches B={B} T={T}: max-abs {dev:.2e} after 10 sampler steps + one evaluation')
    assert dev <= tol


@pytest.mark.parametrize('B,T,n', [(2, 96, 12), (10, 900, 4)])
def test_sampler_loop_is_graph_capturable(model, B, T, n):
    """The whole K-step loop is enqueued by one ABI call without host synchronisation (coefficients and Philox keys are
    kernel arguments), so a caller can capture it into a hipGraph and replay it: same bits as the eager call.
    (10, 900): 290 tiles > 256 CUs, i.e. the two concurrent half-batch chains (a second stream forked and joined with
    events inside the capture)."""
    rs = np.random.RandomState(17)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    x0 = T_(rs.standard_normal((B, 1, 80, T)).astype(np.float32)).cuda()
    eager = model.sample(cond, x0.clone(), seed=11, n_steps=n).clone()      # also warms up (workspaces, function attributes, exchange buffers)
    torch.cuda.synchronize()
    eager_path = model.denoise_fn.last_path()
    # round 4: the stack / part launches keep their launch epoch in device memory (every workgroup reads it at entry, the last one through
    # its layers advances it), so the captured loop runs the SAME launches as the eager call and every replay is bit-identical to it
    xg = x0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            model.sample(cond, xg, seed=11, n_steps=n)
        assert model.denoise_fn.last_path() == eager_path
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        xg.copy_(x0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(xg, eager)
    # an eager call between replays advances the same device-side epoch: both stay valid
    again = model.sample(cond, x0.clone(), seed=11, n_steps=n)
    assert torch.equal(again, eager)
    xg.copy_(x0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(xg, eager)
    assert model.denoise_fn.take_handoff_timeouts() == 0


def test_single_layer_graph_replays(model):
    """A graph holding ONE small-batch layer launch (the channel-split kernel: a pair of workgroups per tile with a flag
    hand-off whose value is the launch epoch) replays the same epoch every time; the consumed flag is cleared in-kernel,
    so every replay must wait for its partner again and reproduce the eager result on fresh inputs."""
    net = model.denoise_fn
    B, T = 2, 128
    rs = np.random.RandomState(23)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    t = torch.tensor([5, 60], device='cuda')
    net.prepare(cond)
    xs = [T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda() for _ in range(3)]
    skip = torch.zeros(B, 256, T, device='cuda')
    want = []
    for x in xs:
        skip.zero_()
        want.append((net.residual_layer(7, x, t, skip).clone(), skip.clone()))
    torch.cuda.synchronize()
    xg, sg = xs[0].clone(), torch.zeros_like(skip)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            og = net.residual_layer(7, xg, t, sg)
    torch.cuda.current_stream().wait_stream(side)
    for x, (wo, ws) in zip(xs, want):
        xg.copy_(x)
        sg.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(og, wo) and torch.equal(sg, ws)
    assert net.handoff_timeouts() == 0
