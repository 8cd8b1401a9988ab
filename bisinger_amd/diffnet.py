"""``DiffNet`` — drop-in for the reference's WaveNet denoiser, running on hand-written HIP kernels.

Reference: /root/reference/train_bisinger/usr/diff/net.py:58-130 (ResidualBlock, DiffNet) and the
``DIFF_DECODERS`` registry usr/diffsinger_task.py:24-29.  Same constructor, same hparams keys, same
``state_dict`` names/shapes, same call contract ``denoise_fn(spec[B,1,M,T], t[B] int64, cond[B,H,T])``.
The modules below only *hold* parameters; all arithmetic happens in libbisinger_hip
(csrc/diffnet.hip) through the C ABI of include/bisinger_hip.h.  No CPU/eager fallback exists.
"""
import math
from ctypes import POINTER, byref, c_void_p, cast

import torch
import torch.nn as nn

from . import _lib
from .hparams import hparams


class Mish(nn.Module):
    """Parameter-free placeholder keeping ``mlp.1`` in place (usr/diff/diffusion.py:68-70)."""

    def forward(self, x):  # pragma: no cover - never called, the kernels implement it
        raise _lib.BsgError('Mish is evaluated inside libbisinger_hip')


class SinusoidalPosEmb(nn.Module):
    """net.py:32-44.  Used on the host only to tabulate the step embedding for the library."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def table(self, n):
        half = self.dim // 2
        e = math.log(10000) / (half - 1)
        e = torch.exp(torch.arange(half) * -e)
        e = torch.arange(n)[:, None] * e[None, :]          # int64 * float32 -> float32, as the reference
        return torch.cat((e.sin(), e.cos()), dim=-1)


def _conv1d(*args, **kw):
    layer = nn.Conv1d(*args, **kw)
    nn.init.kaiming_normal_(layer.weight)
    return layer


class ResidualBlock(nn.Module):
    def __init__(self, encoder_hidden, residual_channels, dilation):
        super().__init__()
        self.dilation = dilation
        self.dilated_conv = _conv1d(residual_channels, 2 * residual_channels, 3, padding=dilation, dilation=dilation)
        self.diffusion_projection = nn.Linear(residual_channels, residual_channels)
        self.conditioner_projection = _conv1d(encoder_hidden, 2 * residual_channels, 1)
        self.output_projection = _conv1d(residual_channels, 2 * residual_channels, 1)


class DiffNet(nn.Module, _lib.HandleOwner, _lib.GemmGuarded):
    GUARD_KIND = 'diffnet'

    def __init__(self, in_dims=80):
        super().__init__()
        self.in_dims = in_dims
        self.encoder_hidden = hparams['hidden_size']
        self.n_layers = hparams['residual_layers']
        self.residual_channels = C = hparams['residual_channels']
        self.dilation_cycle_length = hparams['dilation_cycle_length']
        self.max_steps = max(int(hparams.get('timesteps', 1000)), 1000)
        self.input_projection = _conv1d(in_dims, C, 1)
        self.diffusion_embedding = SinusoidalPosEmb(C)
        self.mlp = nn.Sequential(nn.Linear(C, C * 4), Mish(), nn.Linear(C * 4, C))
        self.residual_layers = nn.ModuleList([
            ResidualBlock(self.encoder_hidden, C, 2 ** (i % self.dilation_cycle_length)) for i in range(self.n_layers)])
        self.skip_projection = _conv1d(C, C, 1)
        self.output_projection = _conv1d(C, in_dims, 1)
        nn.init.zeros_(self.output_projection.weight)
        # extension (not in the reference): arithmetic of the fused residual layers, 'fp32' (parity) or 'bf16'
        # (bf16 MFMA operands, fp32 accumulation; BASELINE config 3) — include/bisinger_hip.h bsg_diffnet_set_compute
        self.compute_dtype = str(hparams.get('diff_compute_dtype', 'fp32'))
        self._h = None
        self._h_key = None
        self._bound = None

    # ------------------------------------------------------------------ handle management
    def handle(self):
        """(Re)create the library handle when the parameters moved or changed (load_ckpt, .to())."""
        key = self._key()
        if self._h is not None and key == self._h_key:
            return self._h
        self.release()
        ws = [p.detach() for p in self._weights()]
        for p in ws:
            if not p.is_cuda:
                raise _lib.BsgError('DiffNet parameters must live on the GPU (model.cuda()); there is no CPU path')
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.BsgError('DiffNet parameters must be contiguous float32')
        lib = _lib.load()
        cfg = _lib.DiffnetCfg(self.in_dims, self.residual_channels, self.encoder_hidden, self.n_layers,
                              self.dilation_cycle_length, self.max_steps)
        arr = (c_void_p * len(ws))(*[p.data_ptr() for p in ws])
        table = self.diffusion_embedding.table(self.max_steps).to(ws[0].device).contiguous()
        h = c_void_p()
        with torch.cuda.device(ws[0].device):
            _lib.check(lib.bsg_diffnet_create(byref(h), byref(cfg), cast(arr, POINTER(c_void_p)), len(ws),
                                              _lib.ptr(table), _lib.stream_ptr()), 'bsg_diffnet_create')
        self._h, self._h_key, self._bound = h, key, None
        self._apply_compute()
        self._apply_guard_state()
        if getattr(self, 'split_disabled', False):
            _lib.check(lib.bsg_diffnet_set_split(h, 0), 'bsg_diffnet_set_split')
        return h

    _COMPUTE = {'fp32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1}

    def _apply_compute(self):
        if self.compute_dtype not in self._COMPUTE:
            raise _lib.BsgError(f"diff_compute_dtype={self.compute_dtype!r}: expected 'fp32' or 'bf16'")
        _lib.check(_lib.load().bsg_diffnet_set_compute(self._h, self._COMPUTE[self.compute_dtype]), 'bsg_diffnet_set_compute')

    def set_compute(self, dtype):
        """'fp32' (default, the <=1e-3 parity configuration) or 'bf16' (bf16 MFMA operands, fp32 accumulation)."""
        if str(dtype) != self.compute_dtype:
            self._bound = None          # the bf16 form keeps its own copy of the hoisted conditioner term: bind again
        self.compute_dtype = str(dtype)
        if self._h is not None:
            self._apply_compute()
        elif self.compute_dtype not in self._COMPUTE:
            raise _lib.BsgError(f"diff_compute_dtype={self.compute_dtype!r}: expected 'fp32' or 'bf16'")

    def set_split_fp16(self, enable):
        """False: this handle's residual stack multiplies on the fp32 matrix pipe only (Winograd kernels: channel-split launches at small
        batch, F(4,3) stack launch when it fills the chip, per-layer launches otherwise) — what BSG_H2=0 does for the whole process."""
        _lib.check(_lib.load().bsg_diffnet_set_h2(self.handle(), int(bool(enable))), 'bsg_diffnet_set_h2')
        self._h2_range_off = False

    def set_q_launch(self, enable):
        """False: not the 16-row stack launch (|x| < 3750) but the 32-row one (|x + d| < 60000) — what BSG_H2_Q=0 does for the whole process."""
        _lib.check(_lib.load().bsg_diffnet_set_h2q(self.handle(), int(bool(enable))), 'bsg_diffnet_set_h2q')
        self._h2q_range_off = False

    def release(self):
        if self._h is not None:
            _lib.load().bsg_diffnet_destroy(self._h)
        self._h = self._h_key = self._bound = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def prepare(self, cond):
        """Bind ``cond`` [B,H,T]: hoists every layer's conditioner projection out of the step loop."""
        h = self.handle()
        # a range event of a split-fp16 launch was a property of the condition bound then: with another condition the handle tries the faster
        # launch again — unless this is the repeated pass of an outer guarded call (GaussianDiffusion.forward recomputes cond: a NEW tensor with
        # the same values), or the handle's inputs keep leaving the range (H2_STRIKES_MAX)
        for attr, strikes, setter in (('_h2_range_off', '_h2_strikes', 'bsg_diffnet_set_h2'), ('_h2q_range_off', '_h2q_strikes', 'bsg_diffnet_set_h2q')):
            off = getattr(self, attr, False)
            if off is not False and off is not cond:
                if _lib.in_retry() or getattr(self, strikes, 0) >= self.H2_STRIKES_MAX:
                    setattr(self, attr, cond)
                else:
                    _lib.check(getattr(_lib.load(), setter)(h, 1), setter)
                    setattr(self, attr, False)
        cond = cond.contiguous().float()
        B, H, T = cond.shape
        assert H == self.encoder_hidden
        with torch.cuda.device(cond.device):
            _lib.check(_lib.load().bsg_diffnet_prepare(h, _lib.ptr(cond), B, T, _lib.stream_ptr()), 'bsg_diffnet_prepare')
        self._bound = (cond, cond._version, B, T)      # strong reference: the allocator cannot hand this address to another tensor
        return B, T

    def _ensure_bound(self, cond):
        """Skip the hoisted conditioner work only when ``cond`` IS the tensor bound last (same object, not modified since).
        Keying on the address would be wrong: the caching allocator gives a freed block to the next same-size tensor."""
        b = self._bound
        if (self._h is None or self._key() != self._h_key or b is None or b[0] is not cond or b[1] != cond._version
                or not cond.is_contiguous() or cond.dtype != torch.float32):
            self.prepare(cond)
            if cond.is_contiguous() and cond.dtype == torch.float32:
                self._bound = (cond, cond._version, cond.shape[0], cond.shape[2])

    # ------------------------------------------------------------------ reference call contract
    @torch.no_grad()
    def forward(self, spec, diffusion_step, cond):
        """spec [B,1,M,T], diffusion_step [B] int64, cond [B,H,T] -> [B,1,M,T]   (net.py:107-130)."""
        B, _, M, T = spec.shape
        self._ensure_bound(cond)
        x = spec[:, 0].contiguous().float()
        t = diffusion_step.to(device=x.device, dtype=torch.long).contiguous()
        eps = torch.empty_like(x)

        def run():
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().bsg_diffnet_forward(self._h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(eps), B, T,
                                                           _lib.stream_ptr()), 'bsg_diffnet_forward')
        self.guarded(run, B, T)
        return eps[:, None, :, :]

    # ------------------------------------------------------------------ hand-off health, checked in the SAME call
    def uses_handoffs(self, B, T):
        """True when launches of shape (B, T) may hand data between workgroups (channel-split launches of small batches,
        the on-chip stack launch whose tiles exchange edges every layer).  Otherwise nothing below synchronises."""
        from ctypes import c_int32
        u = c_int32()
        with _lib.on_device(self):
            _lib.check(_lib.load().bsg_diffnet_uses_handoffs(self._h if self._h is not None else self.handle(), B, T, byref(u)), 'bsg_diffnet_uses_handoffs')
        return bool(u.value)

    def take_handoff_timeouts(self):
        """Wait for the current stream; return (and reset) the number of hand-off spins that gave up since the last take
        (plus range events of the split-fp16 launch)."""
        give, rng = self.take_health()
        return give + rng

    def take_health(self):
        """Wait for the current stream of the handle's device; (hand-off spins that gave up, values beyond the fp16 range seen by the
        split-fp16 stack launch) since the last take; resets both."""
        from ctypes import c_int32
        c = (c_int32 * 2)()
        with _lib.on_device(self):
            _lib.check(_lib.load().bsg_diffnet_health_take(self._h, c, _lib.stream_ptr()), 'bsg_diffnet_health_take')
        return c[0], c[1]

    PART_PATHS = ('stack_h2_quad', 'stack_h2_quad64', 'stack_h2_pair64')
    CLEAN_CALLS_TO_REENABLE = 32     # guarded calls without a give-up after which a demoted handle tries hand-off launches again
    H2_STRIKES_MAX = 3               # range events of the split-fp16 stack launch after which the handle stays on the fp32 matrix pipe

    def guarded(self, run, B, T, restore=None):
        """Run ``run()`` (which enqueues evaluations on this handle) so that an invalid result never leaves the call.  Three events are
        told apart, each detected on the device and healed here before anything is returned (``restore()`` first puts back inputs that
        run() modified in place); one stream synchronisation per call, none inside a stream capture (a captured call cannot wait, and
        runs the kernels that need no guard: per-layer launches of the fp32 matrix pipe):
          * an operand of a split-fp16 GEMM (input / conditioner projections) left the fp16 range -> _lib.range_guarded: every GEMM on
            the fp32 matrix pipe, the bound condition projected again, the work repeated.  Read on EVERY path, also for launch shapes
            without hand-offs; nested in an outer guarded call (GaussianDiffusion.forward, which also has to repeat FS2) it is left to it;
          * a value beyond the fp16 range inside the split-fp16 stack launch (|x + d| >= 60000) -> this handle runs the kernels of the
            fp32 matrix pipe until the next prepare() (the event is a property of the bound input), and the work is repeated;
          * a workgroup gave up waiting for its neighbour (not resident: the GPU is shared with another process) -> hand-off launches
            off for this handle (one workgroup per tile), the work repeated; they are tried again after CLEAN_CALLS_TO_REENABLE clean calls."""
        with _lib.on_device(self):
            capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            run()
            return
        if _lib.guard_mode() == 'deferred':
            # no wait: look at what the PREVIOUS call on this handle left (BsgError, after switching the fallback on), run, and enqueue
            # the non-blocking copy of the health words behind this call's work
            _lib.check_deferred(self)
            _lib.range_guarded(run, 'DiffNet', device=self, owners=(self,))
            self._deferred_enqueue()
            return

        def again():
            if self._bound is not None:
                self.prepare(self._bound[0])
            if restore is not None:
                restore()
        _lib.range_guarded(lambda: self._guarded_handoffs(run, B, T, restore), 'DiffNet', on_retry=again, device=self, owners=(self,))

    def _guarded_handoffs(self, run, B, T, restore):
        """run(), then look at the handle's health words and heal — tier by tier, one repeat per tier, whatever the order in which the
        events show up (a repeat on another launch form can raise the OTHER kind of event: ADVICE r05):
          range event (status word 1), by the launch that raised it:
            16-row stack launch (its conv image holds 16 x: |x| >= 3750)  -> bsg_diffnet_set_h2q(h, 0): the 32-row launch (|x + d| < 60000, ~8 % slower)
            any other split-fp16 launch (|x + d| >= 60000)               -> bsg_diffnet_set_h2(h, 0): the fp32 matrix pipe (~2x slower)
            both while this condition is bound (the event is a property of the input); H2_STRIKES_MAX events keep the handle there
          hand-off give-up (status word 0):
            inside a part launch  -> part forms off (one workgroup per tile), CLEAN_CALLS_TO_REENABLE calls
            otherwise             -> every hand-off launch off (per-layer launches on the fp32 matrix pipe)"""
        import warnings
        run()
        if getattr(self, 'split_disabled', False):
            # a demoted handle launches nothing that hands data between workgroups (bsg_diffnet_uses_handoffs is 0 for every shape while
            # the split is off), so the call just made is clean by construction: count it HERE, before the early return below, and give
            # the hand-off launches another try after CLEAN_CALLS_TO_REENABLE of them (a give-up is a property of the moment — another
            # process on the GPU — not of the handle)
            self._clean_calls = getattr(self, '_clean_calls', 0) + 1
            if self._clean_calls >= self.CLEAN_CALLS_TO_REENABLE:
                _lib.check(_lib.load().bsg_diffnet_set_split(self._h, 1), 'bsg_diffnet_set_split')
                self.split_disabled, self._clean_calls = False, 0
                if getattr(self, 'parts_disabled', False):      # both tiers come back together
                    _lib.check(_lib.load().bsg_diffnet_set_parts(self._h, 1), 'bsg_diffnet_set_parts')
                    self.parts_disabled, self._clean_calls_parts = False, 0
            return
        lib = _lib.load()
        for attempt in range(5):      # at most: h2q off, h2 off | parts off, split off — each tier once
            if not self.uses_handoffs(B, T):
                return
            give, rng = self.take_health()
            if not give and not rng:
                if attempt == 0 and getattr(self, 'parts_disabled', False):      # part forms come back after as many clean calls as hand-off launches do
                    self._clean_calls_parts = getattr(self, '_clean_calls_parts', 0) + 1
                    if self._clean_calls_parts >= self.CLEAN_CALLS_TO_REENABLE:
                        _lib.check(lib.bsg_diffnet_set_parts(self._h, 1), 'bsg_diffnet_set_parts')
                        self.parts_disabled, self._clean_calls_parts = False, 0
                return
            path = self.last_path()
            if not give:
                bound = self._bound[0] if self._bound is not None else True
                if path.startswith('stack_h2q') and not getattr(self, '_h2q_range_off', False):
                    warnings.warn(f'bisinger_amd: {rng} waves saw an activation beyond the range of the 16-row split-fp16 stack launch (|x| >= 3750: its '
                                  f'conv image holds 16 x in fp16 planes); this DiffNet handle takes the 32-row launch (|x + d| < 60000, about 8 % slower) '
                                  f'while this condition is bound and the evaluation is repeated')
                    _lib.check(lib.bsg_diffnet_set_h2q(self._h, 0), 'bsg_diffnet_set_h2q')
                    self._h2q_range_off = bound
                    self._h2q_strikes = getattr(self, '_h2q_strikes', 0) + 1
                else:
                    warnings.warn(f'bisinger_amd: {rng} waves saw an activation beyond the fp16 range of the split-fp16 launch {path} (|x + d| >= 60000); '
                                  f'this DiffNet handle runs the kernels of the fp32 matrix pipe while this condition is bound and the '
                                  f'evaluation is repeated')
                    _lib.check(lib.bsg_diffnet_set_h2(self._h, 0), 'bsg_diffnet_set_h2')
                    self._h2_range_off = bound
                    self._h2_strikes = getattr(self, '_h2_strikes', 0) + 1
                # A split-fp16 GEMM of this handle may have counted an operand too.  Behind a launch that tripped that is the rule, not a second
                # fault: the launch's invalid output (inf / NaN) is the operand of the skip projection.  The word is cleared and only the launch
                # goes one tier down; if a GEMM was the CAUSE (a condition or an x beyond |v| < 4062: its product is inf / NaN, never finite
                # garbage — 16 v overflows the fp16 hi term), the repeat trips again, down to the fp32-pipe kernels, whose non-finite output
                # makes the projections count once more — and the GEMM guard around this call then repeats everything, prepare() included,
                # with the handle's GEMMs on the fp32 matrix pipe
                self.gemm_range_take()
            elif path in self.PART_PATHS and not getattr(self, 'parts_disabled', False):
                # first tier: the give-up came out of a PART launch (several workgroups per tile on CUs of one XCD: partners not on this XCD,
                # or not resident).  The one-workgroup-per-tile stack launch does not depend on that placement
                warnings.warn(f'bisinger_amd: {give} inter-workgroup hand-offs gave up inside a part launch ({path}); part forms are off '
                              f'for this DiffNet handle for the next {self.CLEAN_CALLS_TO_REENABLE} calls (one workgroup per tile) and the evaluation is repeated')
                _lib.check(lib.bsg_diffnet_set_parts(self._h, 0), 'bsg_diffnet_set_parts')
                self.parts_disabled, self._clean_calls_parts = True, 0
            else:
                warnings.warn(f'bisinger_amd: {give} inter-workgroup hand-offs gave up (a partner workgroup was not resident); channel-split and '
                              f'stack launches are off for this DiffNet handle for the next {self.CLEAN_CALLS_TO_REENABLE} calls (per-layer launches '
                              f'on the fp32 matrix pipe) and the evaluation is repeated')
                _lib.check(lib.bsg_diffnet_set_split(self._h, 0), 'bsg_diffnet_set_split')
                self.split_disabled, self._clean_calls = True, 0
            if restore is not None:
                restore()
            run()
            if getattr(self, 'split_disabled', False):
                give, rng = self.take_health()
                if give or rng:
                    raise _lib.BsgError(f'{give + rng} inter-workgroup hand-offs gave up with split launches off: the result is invalid')
                return
        raise _lib.BsgError('DiffNet: the evaluation stayed invalid through every fallback tier')

    # ------------------------------------------------------------------ guard_mode = 'deferred'
    def _deferred_enqueue(self):
        rec = getattr(self, '_deferred', None)
        if rec is None:
            rec = self._deferred = _lib._DeferredWord(3)
        _lib._deferred_objs.add(self)
        with _lib.on_device(self):
            _lib.check(_lib.load().bsg_diffnet_status_async(self._h, c_void_p(rec.next_buf().data_ptr()), _lib.stream_ptr()), 'bsg_diffnet_status_async')
            rec.arm()

    def check_deferred(self, block=True):
        """guard_mode 'deferred': raise BsgError if the previous guarded call produced an invalid result — on this handle: a workgroup gave
        up waiting for a partner (hand-off launches are then off for CLEAN_CALLS_TO_REENABLE calls) or a value left the fp16 range of the
        split-fp16 launch (the handle then runs the fp32 matrix pipe while this condition is bound); on its device: an operand of a
        split-fp16 GEMM left the fp16 range.  No-op when nothing is pending."""
        _lib.check_deferred(self, block)

    def _check_deferred_own(self, block=False):
        rec = getattr(self, '_deferred', None)
        if rec is None or not rec.armed() or self._h is None:
            return
        words = rec.take(block)
        if not words:
            return
        # the words are cumulative until a take resets them: the newest completed read says everything
        give, rng, give_split = words[-1]
        if not (give or rng or give_split):
            if getattr(self, 'parts_disabled', False):
                self._clean_calls_parts = getattr(self, '_clean_calls_parts', 0) + len(words)
                if self._clean_calls_parts >= self.CLEAN_CALLS_TO_REENABLE:
                    _lib.check(_lib.load().bsg_diffnet_set_parts(self._h, 1), 'bsg_diffnet_set_parts')
                    self.parts_disabled, self._clean_calls_parts = False, 0
            if getattr(self, 'split_disabled', False):
                self._clean_calls = getattr(self, '_clean_calls', 0) + len(words)
                if self._clean_calls >= self.CLEAN_CALLS_TO_REENABLE:
                    _lib.check(_lib.load().bsg_diffnet_set_split(self._h, 1), 'bsg_diffnet_set_split')
                    self.split_disabled, self._clean_calls = False, 0
                    if getattr(self, 'parts_disabled', False):
                        _lib.check(_lib.load().bsg_diffnet_set_parts(self._h, 1), 'bsg_diffnet_set_parts')
                        self.parts_disabled, self._clean_calls_parts = False, 0
            return
        self.take_health()      # waits for the stream and resets the words;
        rec.drop_pending()      # reads enqueued before this point repeat the same (cumulative) counts: dropped — repeat everything issued since the last clean check
        what = []
        if rng:
            bound = self._bound[0] if self._bound is not None else True
            if self.last_path().startswith('stack_h2q') and not getattr(self, '_h2q_range_off', False):
                _lib.check(_lib.load().bsg_diffnet_set_h2q(self._h, 0), 'bsg_diffnet_set_h2q')
                self._h2q_range_off = bound
                self._h2q_strikes = getattr(self, '_h2q_strikes', 0) + 1
                what.append(f'{rng} waves saw a value beyond the range of the 16-row split-fp16 stack launch (|x| >= 3750; this handle now takes the '
                            f'32-row launch while this condition is bound)')
            else:
                _lib.check(_lib.load().bsg_diffnet_set_h2(self._h, 0), 'bsg_diffnet_set_h2')
                self._h2_range_off = bound
                self._h2_strikes = getattr(self, '_h2_strikes', 0) + 1
                what.append(f'{rng} waves saw a value beyond the fp16 range of the split-fp16 launch (this handle now runs the fp32 matrix pipe '
                            f'while this condition is bound)')
        if give and not give_split and self.last_path() in self.PART_PATHS and not getattr(self, 'parts_disabled', False):
            # first tier (as in the same-call mode): the give-up came out of a part launch; the one-workgroup-per-tile launch stays
            _lib.check(_lib.load().bsg_diffnet_set_parts(self._h, 0), 'bsg_diffnet_set_parts')
            self.parts_disabled, self._clean_calls_parts = True, 0
            what.append(f'{give} inter-workgroup hand-offs gave up inside a part launch (part forms are off for the next '
                        f'{self.CLEAN_CALLS_TO_REENABLE} calls)')
        elif give or give_split:
            _lib.check(_lib.load().bsg_diffnet_set_split(self._h, 0), 'bsg_diffnet_set_split')
            self.split_disabled, self._clean_calls = True, 0
            what.append(f'{give + give_split} inter-workgroup hand-offs gave up (hand-off launches are off for the next '
                        f'{self.CLEAN_CALLS_TO_REENABLE} calls)')
        raise _lib.BsgError('a PREVIOUS call on this DiffNet handle produced an invalid result (guard_mode=deferred): ' + '; '.join(what) +
                            '.  Repeat the work')

    def debug_inject_giveup(self, n_launches):
        """Fault injection (tests): the next ``n_launches`` channel-split launches give up their hand-offs without waiting."""
        _lib.check(_lib.load().bsg_diffnet_debug_inject_giveup(self.handle(), int(n_launches)), 'bsg_diffnet_debug_inject_giveup')

    def debug_inject_xcc(self, n_launches):
        """Fault injection (tests): in the next ``n_launches`` part launches the odd parts of a tile report another XCD than their own."""
        _lib.check(_lib.load().bsg_diffnet_debug_inject_xcc(self.handle(), int(n_launches)), 'bsg_diffnet_debug_inject_xcc')

    def debug_set_epoch(self, epoch):
        """Test hook: the device-side launch epoch of the handle's stack / part launches (wraps to 1 at 2**25, zeroing the hand-off flags)."""
        _lib.check(_lib.load().bsg_diffnet_debug_set_epoch(self.handle(), int(epoch), _lib.stream_ptr()), 'bsg_diffnet_debug_set_epoch')

    def profile(self, enable):
        """Record hipEvent pairs around the residual-layer launches of every evaluation (bench.py roofline)."""
        _lib.check(_lib.load().bsg_diffnet_profile(self.handle(), int(enable)), 'bsg_diffnet_profile')

    def last_path(self):
        """Form of the last residual-layer launch: 'stack', 'layer', 'split2', 'split4', 'wide', 'bf16' or 'none'."""
        return _lib.load().bsg_diffnet_last_path(self._h if self._h is not None else self.handle()).decode()      # (a query of the handle that exists: no look at the weights)

    def handoff_timeouts(self):
        """Hand-off health: spins that gave up and were not yet taken (0 unless a workgroup was not resident). Synchronises."""
        from ctypes import c_int32
        n = c_int32()
        _lib.check(_lib.load().bsg_diffnet_status(self._h, byref(n)), 'bsg_diffnet_status')
        return n.value

    def clock_read(self):
        """-> (shader MHz held over the last profiled stack launch, its in-kernel span in us); (0, 0) when none ran.  Synchronises."""
        from ctypes import c_double
        mhz, span = c_double(), c_double()
        _lib.check(_lib.load().bsg_diffnet_clock_read(self._h, byref(mhz), byref(span)), 'bsg_diffnet_clock_read')
        return mhz.value, span.value

    def profile_read(self):
        """-> (summed device ms of the recorded residual-layer chains, number of layer launches covered)."""
        from ctypes import c_double, c_int64
        ms, n = c_double(), c_int64()
        _lib.check(_lib.load().bsg_diffnet_profile_read(self._h, byref(ms), byref(n)), 'bsg_diffnet_profile_read')
        return ms.value, n.value

    @torch.no_grad()
    def residual_layer(self, layer, x, t, skip):
        """One fused ResidualBlock (net.py:66-78) — unit-test / micro-benchmark hook."""
        B, C, T = x.shape
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().bsg_diffnet_residual_layer(self._h, layer, _lib.ptr(x), _lib.ptr(t), _lib.ptr(out),
                                                              _lib.ptr(skip), B, T, _lib.stream_ptr()),
                       'bsg_diffnet_residual_layer')
        return out


def _fft(hp):
    from .candidate_decoder import FFT
    return FFT(hp['hidden_size'], hp['dec_layers'], hp['dec_ffn_kernel_size'], hp['num_heads'])


# usr/diffsinger_task.py:24-29
DIFF_DECODERS = {
    'wavenet': lambda hp: DiffNet(hp['audio_num_mel_bins']),
    'fft': _fft,
}
