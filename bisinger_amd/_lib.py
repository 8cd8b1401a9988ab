"""ctypes binding of libbisinger_hip.so (include/bisinger_hip.h).

There is NO fallback: if the library is missing or a call fails, the product path raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libbisinger_hip.so')
if os.environ.get('BSG_LIB'):      # development: an alternative build of the same ABI (kernel experiments)
    LIB_PATH = os.environ['BSG_LIB']
ABI_VERSION = 7


class BsgError(RuntimeError):
    pass


class DiffnetCfg(Structure):
    _fields_ = [('in_dims', c_int32), ('residual_channels', c_int32), ('encoder_hidden', c_int32),
                ('residual_layers', c_int32), ('dilation_cycle_length', c_int32), ('max_steps', c_int32)]


class Fs2Cfg(Structure):
    _fields_ = [(n, c_int32) for n in ('hidden_size', 'vocab', 'enc_layers', 'dec_layers', 'num_heads',
                                       'enc_ffn_kernel_size', 'dec_ffn_kernel_size', 'out_dims', 'dur_layers',
                                       'dur_kernel', 'spk_rows', 'esm_heads', 'n_pos', 'n_rel')]


class HifiganCfg(Structure):
    _fields_ = [('n_mel', c_int32), ('upsample_initial_channel', c_int32), ('n_ups', c_int32),
                ('upsample_rates', c_int32 * 8), ('upsample_kernel_sizes', c_int32 * 8), ('n_kernels', c_int32),
                ('resblock_kernel_sizes', c_int32 * 8), ('n_dil', c_int32), ('resblock_dilations', (c_int32 * 4) * 8),
                ('weight_norm', c_int32), ('use_nsf', c_int32), ('sample_rate', c_int32), ('harmonic_num', c_int32),
                ('resblock', c_int32)]


class PitchextCfg(Structure):
    _fields_ = [(n, c_int32) for n in ('hidden_size', 'n_mel', 'conv_layers', 'predictor_layers', 'predictor_kernel', 'use_uv', 'n_pos')]


class Schedule(Structure):
    _fields_ = [('num_timesteps', c_int32),
                ('sqrt_recip_alphas_cumprod', POINTER(c_float)), ('sqrt_recipm1_alphas_cumprod', POINTER(c_float)),
                ('posterior_mean_coef1', POINTER(c_float)), ('posterior_mean_coef2', POINTER(c_float)),
                ('sigma', POINTER(c_float)), ('alphas_cumprod', POINTER(c_float))]


_SIGS = {
    'bsg_abi_version': (c_int32, []),
    'bsg_last_error': (c_char_p, []),
    'bsg_device_arch': (c_char_p, []),
    'bsg_diffnet_create': (c_int32, [POINTER(c_void_p), POINTER(DiffnetCfg), POINTER(c_void_p), c_int32, c_void_p, c_void_p]),
    'bsg_diffnet_destroy': (None, [c_void_p]),
    'bsg_diffnet_prepare': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_diffnet_forward': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_diffnet_residual_layer': (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_diffnet_debug_stamps': (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    'bsg_ddpm_sample': (c_int32, [c_void_p, POINTER(Schedule), c_void_p, c_void_p, c_uint64, c_int32, c_int32, c_int32,
                                  c_int32, c_int32, c_int32, c_void_p]),
    'bsg_plms_sample': (c_int32, [c_void_p, POINTER(Schedule), c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'bsg_ddpm_step': (c_int32, [c_void_p, c_void_p, c_void_p, POINTER(Schedule), c_int32, c_int64, c_uint64, c_uint64, c_void_p]),
    'bsg_plms_step': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, POINTER(Schedule), c_int32, c_int32,
                                c_int64, c_void_p]),
    'bsg_fftden_n_weights': (c_int32, [c_int32]),
    'bsg_fftden_create': (c_int32, [POINTER(c_void_p), c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p), c_int32,
                                    c_void_p, c_void_p, c_void_p]),
    'bsg_fftden_destroy': (None, [c_void_p]),
    'bsg_fftden_prepare': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_fftden_forward': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_philox_normal': (c_int32, [c_void_p, c_int64, c_uint64, c_uint32, c_uint64, c_void_p]),
    'bsg_mel_start': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    'bsg_mel_finish': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    'bsg_diffnet_last_path': (c_char_p, [c_void_p]),
    'bsg_diffnet_clock_read': (c_int32, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    'bsg_diffnet_debug_stack_stamps': (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'bsg_diffnet_status': (c_int32, [c_void_p, POINTER(c_int32)]),
    'bsg_diffnet_set_compute': (c_int32, [c_void_p, c_int32]),
    'bsg_diffnet_status_async': (c_int32, [c_void_p, c_void_p, c_void_p]),     # host_counts: THREE words (ABI v5)
    'bsg_diffnet_handoff_take': (c_int32, [c_void_p, POINTER(c_int32), c_void_p]),
    'bsg_diffnet_health_take': (c_int32, [c_void_p, POINTER(c_int32), c_void_p]),
    'bsg_diffnet_uses_handoffs': (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_int32)]),
    'bsg_diffnet_set_split': (c_int32, [c_void_p, c_int32]),
    'bsg_diffnet_debug_inject_giveup': (c_int32, [c_void_p, c_int32]),
    'bsg_diffnet_debug_set_epoch': (c_int32, [c_void_p, c_uint32, c_void_p]),
    'bsg_diffnet_debug_inject_xcc': (c_int32, [c_void_p, c_int32]),
    'bsg_diffnet_set_parts': (c_int32, [c_void_p, c_int32]),
    'bsg_diffnet_profile': (c_int32, [c_void_p, c_int32]),
    'bsg_diffnet_profile_read': (c_int32, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    'bsg_fs2midi_n_weights': (c_int32, [POINTER(Fs2Cfg)]),
    'bsg_fs2midi_create': (c_int32, [POINTER(c_void_p), POINTER(Fs2Cfg), POINTER(c_void_p), c_int32, c_void_p, c_void_p, c_void_p]),
    'bsg_fs2midi_destroy': (None, [c_void_p]),
    'bsg_fs2midi_encode': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    'bsg_fs2midi_encode_rows': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                          c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'bsg_fs2midi_last_rows': (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    'bsg_length_regulator': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    'bsg_fs2midi_decode': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                     c_void_p, c_void_p]),
    'bsg_hifigan_n_weights': (c_int32, [POINTER(HifiganCfg)]),
    'bsg_hifigan_create': (c_int32, [POINTER(c_void_p), POINTER(HifiganCfg), POINTER(c_void_p), c_int32, c_void_p]),
    'bsg_hifigan_destroy': (None, [c_void_p]),
    'bsg_hifigan_forward': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_hifigan_forward_nsf': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_pitchext_n_weights': (c_int32, [POINTER(PitchextCfg)]),
    'bsg_pitchext_create': (c_int32, [POINTER(c_void_p), POINTER(PitchextCfg), POINTER(c_void_p), c_int32, c_void_p, c_void_p]),
    'bsg_pitchext_destroy': (None, [c_void_p]),
    'bsg_pitchext_forward': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_weight_norm_fold': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'bsg_gemm_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                               c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int32, c_void_p]),
    'bsg_gemm_set_split': (c_int32, [c_int32]),
    'bsg_diffnet_set_h2': (c_int32, [c_void_p, c_int32]),
    'bsg_gemm_range_events': (c_int32, [POINTER(c_int32), c_int32, c_void_p]),
    'bsg_gemm_range_events_async': (c_int32, [c_void_p, c_void_p]),
    'bsg_gemm_presplit_f32': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                        c_int32, c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises BsgError when it is absent: build it with
    `python -m bisinger_amd.build` (or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BsgError(f'{LIB_PATH} not found: the HIP extension is not built. '
                       f'Run `python -m bisinger_amd.build`. There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)           # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.bsg_abi_version() != ABI_VERSION:
        raise BsgError(f'ABI mismatch: library {lib.bsg_abi_version()} vs binding {ABI_VERSION}; rebuild')
    _lib = lib
    return lib


range_retries = 0      # calls repeated on the fp32 matrix pipe because an operand left the fp16 range of the split-fp16 GEMMs
_range_depth = 0       # > 0 inside a range_guarded() call: nested entries leave the check to the outermost one
_range_strikes = 0     # range events seen by this process; from RANGE_STRIKES_MAX on the GEMMs stay on the fp32 matrix pipe
RANGE_STRIKES_MAX = 3


def on_device(device):
    """Context: `device` (a torch.device, a tensor, a module, or None = leave the ambient device) current.  The guard counters live
    per device and are read on the current stream OF that device: a model on cuda:1 while cuda:0 is current must not wait on, read
    or reset cuda:0's state."""
    import contextlib
    import torch
    if device is None:
        return contextlib.nullcontext()
    if isinstance(device, torch.nn.Module):
        device = next((p.device for p in device.parameters()), None)
        if device is None:
            return contextlib.nullcontext()
    elif isinstance(device, torch.Tensor):
        device = device.device
    device = torch.device(device)
    if device.type != 'cuda':
        return contextlib.nullcontext()
    return torch.cuda.device(device)


def gemm_range_take(device=None):
    """Wait for the current stream of `device`; number of split-fp16 GEMM waves that staged an out-of-range operand there since the
    last take (reset)."""
    n = c_int32()
    with on_device(device):
        check(load().bsg_gemm_range_events(ctypes.byref(n), 1, stream_ptr()), 'bsg_gemm_range_events')
    return n.value


def gemm_range_peek(device=None):
    """The same count without resetting it (a nested guard looks at it and leaves the take to the outermost one)."""
    n = c_int32()
    with on_device(device):
        check(load().bsg_gemm_range_events(ctypes.byref(n), 0, stream_ptr()), 'bsg_gemm_range_events')
    return n.value


_in_retry = False      # True while range_guarded runs the REPEATED pass (every GEMM on the fp32 matrix pipe): handles keep their fallbacks
_restore_split_after = False   # deferred mode: a detection moved the GEMMs to the fp32 pipe for the next call; that call puts them back


def guard_mode():
    """hparams['guard_mode']: 'same_call' (default: every guarded entry waits for its stream once and repeats its own work on an event —
    nothing invalid ever leaves a call) or 'deferred' (no wait: the health words are copied to pinned host memory behind the work, and
    the NEXT guarded call — or check_deferred() — raises BsgError if the previous result was invalid, after switching the fallback on.
    For callers that overlap consecutive batches on several streams and can afford to discard one result)."""
    from .hparams import hparams
    return str(hparams.get('guard_mode', 'same_call'))


class _DeferredWord:
    """Pending non-blocking reads of `n` int32 health words: each arm() takes a pinned host buffer (the caller enqueues the copy into
    next_buf() first) and records an event behind it; take() returns the words of every read whose event has PASSED — it never waits unless
    block=True — oldest first.  A check therefore trails the work by as many calls as the GPU is behind the host."""

    def __init__(self, n):
        self.n = n
        self.pending = []      # [(buf, event)]
        self.free = []
        self.cur = None

    def next_buf(self):
        import torch
        self.cur = self.free.pop() if self.free else torch.zeros(self.n, dtype=torch.int32).pin_memory()
        return self.cur

    def armed(self):
        return bool(self.pending)

    def arm(self):
        import torch
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((self.cur, ev))
        self.cur = None

    def drop_pending(self):
        """Forget the reads still in flight (their counts repeat what a blocking take has just reset).  Their pinned buffers are NOT handed
        back to torch's allocator while a copy may still land in them (only the current stream was waited for): they are parked until their
        event has passed."""
        self.parked = [(b, e) for b, e in getattr(self, 'parked', []) + self.pending if not e.query()]
        self.pending = []

    def take(self, block=False):
        out = []
        while self.pending:
            buf, ev = self.pending[0]
            if not ev.query():
                if not block:
                    break
                ev.synchronize()
            self.pending.pop(0)
            out.append([int(v) for v in buf.tolist()])
            self.free.append(buf)
        return out


_deferred_gemm = {}    # device index -> _DeferredWord(1)
import weakref  # noqa: E402
_deferred_objs = weakref.WeakSet()   # handles (DiffNet) with health words of their own: _check_deferred_own() raises for the previous call


def _device_index(device):
    import torch
    with on_device(device):
        return torch.cuda.current_device()


def check_deferred(device=None, block=False):
    """Deferred guard mode: raise BsgError if a guarded call on `device` whose work has completed since the last check left an invalid
    result — a split-fp16 GEMM staged an operand beyond the fp16 range (the GEMMs are then on the fp32 matrix pipe for the next call), or
    one of the registered handles (DiffNet) reports a give-up / a range event.  Never waits for the GPU unless block=True (then every
    pending read is waited for: call it before trusting the last results of a stream of requests).  The counters are per DEVICE: with
    several streams in flight an event cannot be attributed to one of them — treat every result issued since the last clean check as
    suspect."""
    global _range_strikes, range_retries, _restore_split_after
    idx = _device_index(device)
    errs = []
    rec = _deferred_gemm.get(idx)
    if rec is not None and rec.armed():
        n = sum(w[0] for w in rec.take(block))
        if n:
            gemm_range_take(device)      # waits for the stream and resets the counter;
            rec.drop_pending()           # reads enqueued before this point repeat the same (cumulative) count
            with on_device(device):
                check(load().bsg_gemm_set_split(0), 'bsg_gemm_set_split')
            _range_strikes += 1
            range_retries += 1
            _restore_split_after = _range_strikes < RANGE_STRIKES_MAX and os.environ.get('BSG_GEMM_SPLIT', '1') != '0'
            errs.append(f'{n} split-fp16 GEMM waves of a PREVIOUS call staged an operand beyond the fp16 range (|v| >= 4062): its result '
                        f'was invalid (guard_mode=deferred).  Every GEMM is on the fp32 matrix pipe for the next call: repeat the work')
    for obj in list(_deferred_objs):      # one error for everything the previous call left, whichever entry looks first
        if _device_index(obj) == idx:
            try:
                obj._check_deferred_own(block)
            except BsgError as e:
                errs.append(str(e))
    if errs:
        raise BsgError(' | '.join(errs))


def _deferred_enqueue(device):
    idx = _device_index(device)
    rec = _deferred_gemm.get(idx)
    if rec is None:
        rec = _deferred_gemm[idx] = _DeferredWord(1)
    with on_device(device):
        check(load().bsg_gemm_range_events_async(c_void_p(rec.next_buf().data_ptr()), stream_ptr()), 'bsg_gemm_range_events_async')
        rec.arm()


def range_guarded(run, what, on_retry=None, device=None):
    """Every public entry that may enqueue split-fp16 products outside the residual stack (FS2 linears and fused attention, the
    conditioner / input projections, HiFi-GAN's ResBlock pairs, PitchExtractor, the FFT denoiser) goes through here, so that an operand
    beyond the fp16 range of the split (|v| >= 4062 after scaling: counted by the kernels, never clipped) cannot leave the call as a
    silent NaN: the OUTERMOST guarded call waits for its stream once, and on an event moves every GEMM to the fp32 matrix pipe
    (bsg_gemm_set_split(0)), warns and runs `run()` again (`on_retry()` first restores what run() consumed).  The split form comes
    back for the next call — the event was a property of this input — until RANGE_STRIKES_MAX events have been seen in the process.
    Inside a stream capture nothing can wait: the counter is left for the next guarded call (which then repeats its own work).
    `device`: where the guarded work runs (device, tensor or module): the counter of THAT device is read on ITS current stream.
    guard_mode 'deferred' (see guard_mode()): no wait; the previous call's counter is looked at instead, and a BsgError raised for it."""
    global _range_depth, range_retries, _range_strikes, _in_retry, _restore_split_after
    import torch
    with on_device(device):
        capturing = torch.cuda.is_current_stream_capturing()
    if _range_depth > 0 or capturing:
        return run()
    _range_depth += 1
    try:
        if guard_mode() == 'deferred':
            check_deferred(device)
            restore = _restore_split_after
            _restore_split_after = False
            try:
                out = run()
            finally:
                # also when run() raises (a nested handle's check_deferred, say): the switch back to the split-fp16 GEMMs was consumed above and
                # would otherwise be lost — every GEMM of the process on the fp32 matrix pipe for good, silently
                if restore:
                    with on_device(device):
                        check(load().bsg_gemm_set_split(1), 'bsg_gemm_set_split')
            _deferred_enqueue(device)
            return out
        out = run()
        if gemm_range_take(device):
            import warnings
            warnings.warn(f'bisinger_amd: {what}: an operand left the fp16 range of the split-fp16 GEMMs (|v| >= 4062); the call is '
                          f'repeated with every GEMM on the fp32 matrix pipe')
            check(load().bsg_gemm_set_split(0), 'bsg_gemm_set_split')
            range_retries += 1
            _range_strikes += 1
            _in_retry = True
            try:
                if on_retry is not None:
                    on_retry()
                out = run()
            finally:
                _in_retry = False
            gemm_range_take(device)
            if _range_strikes < RANGE_STRIKES_MAX and os.environ.get('BSG_GEMM_SPLIT', '1') != '0':
                check(load().bsg_gemm_set_split(1), 'bsg_gemm_set_split')
    finally:
        _range_depth -= 1
    return out


def declared_symbols():
    return list(_SIGS)


def check(rc, what=''):
    if rc != 0:
        msg = load().bsg_last_error()
        raise BsgError(f'{what} failed (code {rc}): {msg.decode() if msg else ""}')


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA(=HIP) tensor, or NULL for None."""
    if t is None:
        return c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), 'expected a contiguous device tensor'
    return c_void_p(t.data_ptr())
