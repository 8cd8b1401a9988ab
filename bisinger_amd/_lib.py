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
    'bsg_diffnet_set_h2q': (c_int32, [c_void_p, c_int32]),
    'bsg_gemm_range_events': (c_int32, [POINTER(c_int32), c_int32, c_void_p]),
    'bsg_gemm_range_events_async': (c_int32, [c_void_p, c_void_p]),
    'bsg_handle_range_events': (c_int32, [c_int32, c_void_p, POINTER(c_int32), c_int32, c_void_p]),
    'bsg_handle_range_events_async': (c_int32, [c_int32, c_void_p, c_void_p, c_void_p]),
    'bsg_handle_set_gemm_split': (c_int32, [c_int32, c_void_p, c_int32]),
    'bsg_handle_get_gemm_split': (c_int32, [c_int32, c_void_p, POINTER(c_int32)]),
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
    # torch first: libbisinger_hip.so needs libamdhip64, and the process must hold ONE HIP runtime — the one torch ships and initialises.
    # Loaded before torch, the library binds /opt/rocm's copy, torch then loads its own, and every hipMalloc of this library fails with
    # "no ROCm-capable device is detected" (seen with `python __graft_entry__.py smoke`: build() loaded the library before anything had
    # imported torch)
    import torch  # noqa: F401
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


range_retries = 0      # calls repeated on the fp32 matrix pipe because an operand left the fp16 range of the split-fp16 GEMMs (process total)
RANGE_STRIKES_MAX = 3  # range events of ONE handle after which its GEMMs stay on the fp32 matrix pipe
HANDLE_KINDS = {'diffnet': 0, 'fs2midi': 1, 'hifigan': 2, 'pitchext': 3, 'fftden': 4}     # include/bisinger_hip.h BSG_HANDLE_*

import threading  # noqa: E402
import weakref  # noqa: E402

_tls = threading.local()      # per host thread: nesting depth of range_guarded, the handles inside the outermost call, "this is the repeated pass"


def _state():
    st = _tls
    if not hasattr(st, 'depth'):
        st.depth, st.owners, st.in_retry = 0, [], False
    return st


def in_retry():
    """True while range_guarded runs the REPEATED pass of the calling thread (the handles that tripped are on the fp32 matrix pipe):
    handles keep their fallbacks then (DiffNet.prepare)."""
    return _state().in_retry


def on_device(device):
    """Context: `device` (a torch.device, a tensor, a module, or None = leave the ambient device) current.  A handle's guard words live
    on ITS device and are read on the current stream OF that device: a model on cuda:1 while cuda:0 is current must not wait on, read
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
    """PROCESS-WIDE word (the handle-less entries bsg_gemm_f32 / bsg_gemm_presplit_f32 count into it; handles have their own —
    GemmGuarded): wait for the current stream of `device`; number of split-fp16 GEMM waves that staged an out-of-range operand there
    since the last take (reset)."""
    n = c_int32()
    with on_device(device):
        check(load().bsg_gemm_range_events(ctypes.byref(n), 1, stream_ptr()), 'bsg_gemm_range_events')
    return n.value


def gemm_range_peek(device=None):
    """The same count without resetting it."""
    n = c_int32()
    with on_device(device):
        check(load().bsg_gemm_range_events(ctypes.byref(n), 0, stream_ptr()), 'bsg_gemm_range_events')
    return n.value


class HandleOwner:
    """Mixin of every module that owns a library handle: the handle is re-created when a parameter or buffer was moved, replaced or
    modified (load_ckpt, .to(), an in-place edit) — told from (data_ptr, _version) of every state_dict entry.  Walking the module tree for
    that (state_dict()) costs 250-300 us per look, and a pass looks several times: ~1 ms of host time per pass, most of the gaps between
    the front's kernels (tools/_front.sh).  The SLOTS — (dict of the owning module, name) of every parameter and persistent buffer — are
    found once (the module tree of these drop-ins is fixed after construction); a look reads each slot's current tensor: a replaced
    tensor (a buffer after .to(), an assigned Parameter) is seen like a modified one.  50 us."""

    def _slots(self):
        slots = self.__dict__.get('_handle_slots')
        if slots is None:
            slots = []
            for mod in self.modules():
                slots.extend((mod._parameters, n) for n, v in mod._parameters.items() if v is not None)
                slots.extend((mod._buffers, n) for n, v in mod._buffers.items() if v is not None and n not in mod._non_persistent_buffers_set)
            self.__dict__['_handle_slots'] = slots
        return slots

    def _weights(self):
        """The tensors in state_dict() order: the order of the ABI's weight pointer arrays."""
        return list(self.state_dict(keep_vars=True).values())

    def _key(self):
        for attempt in (0, 1):
            k = []
            try:
                for d, n in self._slots():
                    t = d[n]
                    k.append(t.data_ptr())
                    k.append(t._version)
                return tuple(k)
            except (KeyError, AttributeError):      # a parameter was removed or set to None (weight norm folded ...): find the slots again
                self.__dict__.pop('_handle_slots', None)
                if attempt:
                    raise


class GemmGuarded:
    """Mixin of every module that owns a library handle (``self._h``; ``GUARD_KIND`` names its type): the range guard of the handle's
    split-fp16 GEMMs / fused attention / ResBlock convolutions is the HANDLE's — its own device word, its own switch to the fp32 matrix
    pipe, its own strike count (include/bisinger_hip.h bsg_handle_*).  An out-of-range input to one model demotes that model only."""
    GUARD_KIND = None

    def _guard_args(self):
        h = getattr(self, '_h', None)
        return (HANDLE_KINDS[self.GUARD_KIND], h) if h is not None else None

    def gemm_range_take(self):
        """Wait for the current stream of the handle's device; waves of this handle's kernels that staged an operand beyond the fp16
        range (|v| >= 4062) since the last take; resets the word.  0 when the handle does not exist yet."""
        a = self._guard_args()
        if a is None:
            return 0
        n = c_int32()
        with on_device(self):
            check(load().bsg_handle_range_events(a[0], a[1], ctypes.byref(n), 1, stream_ptr()), 'bsg_handle_range_events')
        return n.value

    def gemm_range_peek(self):
        a = self._guard_args()
        if a is None:
            return 0
        n = c_int32()
        with on_device(self):
            check(load().bsg_handle_range_events(a[0], a[1], ctypes.byref(n), 0, stream_ptr()), 'bsg_handle_range_events')
        return n.value

    def set_gemm_split(self, enable):
        """False: this handle's GEMMs on the fp32 matrix pipe (remembered across a re-creation of the handle); True: split-fp16 again."""
        self._gemm_split_off = not enable
        a = self._guard_args()
        if a is not None:
            check(load().bsg_handle_set_gemm_split(a[0], a[1], int(bool(enable))), 'bsg_handle_set_gemm_split')

    def gemm_split_enabled(self):
        """The effective switch: this handle's AND the process-wide one (BSG_GEMM_SPLIT, bsg_gemm_set_split)."""
        a = self._guard_args()
        if a is None:
            return not getattr(self, '_gemm_split_off', False)
        v = c_int32()
        check(load().bsg_handle_get_gemm_split(a[0], a[1], ctypes.byref(v)), 'bsg_handle_get_gemm_split')
        return bool(v.value)

    def _apply_guard_state(self):
        """After (re)creating the handle: a handle that was demoted stays demoted."""
        if getattr(self, '_gemm_split_off', False):
            self.set_gemm_split(False)

    @property
    def gemm_range_strikes(self):
        return getattr(self, '_gemm_strikes', 0)


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


_deferred_objs = weakref.WeakSet()   # handles with reads in flight (guard_mode 'deferred'): their GEMM word, and DiffNet's health words


def _device_index(device):
    import torch
    with on_device(device):
        return torch.cuda.current_device()


def _split_env_on():
    return os.environ.get('BSG_GEMM_SPLIT', '1') != '0'


def check_deferred(device=None, block=False):
    """Deferred guard mode: raise BsgError if a guarded call on `device` whose work has completed since the last check left an invalid
    result — a split-fp16 GEMM of some handle staged an operand beyond the fp16 range (THAT handle's GEMMs are then on the fp32 matrix
    pipe for the next call), or a DiffNet handle reports a give-up / a range event of its stack launch.  Never waits for the GPU unless
    block=True (then every pending read is waited for: call it before trusting the last results of a stream of requests).  With several
    streams in flight an event cannot be attributed to one of them — treat every result issued on the handle since the last clean
    check as suspect."""
    global range_retries
    idx = _device_index(device)
    errs = []
    for obj in list(_deferred_objs):      # one error for everything the previous call left, whichever entry looks first
        if _device_index(obj) != idx:
            continue
        rec = getattr(obj, '_deferred_gemm', None)
        if rec is not None and rec.armed():
            n = sum(w[0] for w in rec.take(block))
            if n:
                obj.gemm_range_take()        # waits for the stream and resets the word;
                rec.drop_pending()           # reads enqueued before this point repeat the same (cumulative) count
                obj.set_gemm_split(False)
                obj._gemm_strikes = obj.gemm_range_strikes + 1
                range_retries += 1
                obj._restore_split_after = obj._gemm_strikes < RANGE_STRIKES_MAX and _split_env_on()
                errs.append(f'{n} split-fp16 GEMM waves of a PREVIOUS call on this {type(obj).__name__} handle staged an operand beyond the '
                            f'fp16 range (|v| >= 4062): its result was invalid (guard_mode=deferred).  The handle\'s GEMMs are on the fp32 '
                            f'matrix pipe for the next call: repeat the work')
        if hasattr(obj, '_check_deferred_own'):
            try:
                obj._check_deferred_own(block)
            except BsgError as e:
                errs.append(str(e))
    if errs:
        raise BsgError(' | '.join(errs))


def _deferred_enqueue(owners):
    for o in owners:
        a = o._guard_args()
        if a is None:
            continue
        rec = getattr(o, '_deferred_gemm', None)
        if rec is None:
            rec = o._deferred_gemm = _DeferredWord(1)
        with on_device(o):
            check(load().bsg_handle_range_events_async(a[0], a[1], c_void_p(rec.next_buf().data_ptr()), stream_ptr()), 'bsg_handle_range_events_async')
            rec.arm()
        _deferred_objs.add(o)


def _take_all(owners, device):
    """ONE wait for everything the outermost call enqueued: the words of all its handles are copied to pinned host memory behind the work,
    the stream is waited for once, and only the words that are non-zero are reset.  -> [(owner, count)] of the handles with events."""
    import torch
    live = [(o, o._guard_args()) for o in owners]
    live = [(o, a) for o, a in live if a is not None]
    if not live:
        return []
    st = _state()
    n = len(live)
    if getattr(st, 'pin', None) is None or st.pin.numel() < n:
        st.pin = torch.zeros(max(8, n), dtype=torch.int32).pin_memory()
    hits = []
    by_dev = {}
    for i, (o, a) in enumerate(live):
        by_dev.setdefault(_device_index(o), []).append((i, o, a))
    for _, group in by_dev.items():
        with on_device(group[0][1]):
            for i, o, a in group:
                check(load().bsg_handle_range_events_async(a[0], a[1], c_void_p(st.pin.data_ptr() + 4 * i), stream_ptr()), 'bsg_handle_range_events_async')
            torch.cuda.current_stream().synchronize()
        for i, o, a in group:
            v = int(st.pin[i])
            if v:
                o.gemm_range_take()      # reset (the stream is idle: no second wait of any length)
                hits.append((o, v))
    return hits


def range_guarded(run, what, on_retry=None, device=None, owners=()):
    """Every public entry that may enqueue split-fp16 products outside the residual stack (FS2 linears and fused attention, the
    conditioner / input projections, HiFi-GAN's ResBlock pairs, PitchExtractor, the FFT denoiser) goes through here with the handle(s)
    it drives (`owners`: GemmGuarded objects), so that an operand beyond the fp16 range of the split (|v| >= 4062 after scaling: counted
    by the kernels into the HANDLE's word, never clipped) cannot leave the call as a silent NaN: the OUTERMOST guarded call of a thread
    waits for its stream once, looks at the words of every handle that ran inside it (nested guarded calls register theirs), and on an
    event moves THOSE handles' GEMMs to the fp32 matrix pipe (bsg_handle_set_gemm_split), warns and runs `run()` again (`on_retry()`
    first restores what run() consumed).  The split form comes back for the next call — the event was a property of this input — until a
    handle has seen RANGE_STRIKES_MAX events.  Other handles of the process are not touched.
    Inside a stream capture nothing can wait: the words are left for the next guarded call (which then repeats its own work).
    `device`: where the guarded work runs (device, tensor or module).
    guard_mode 'deferred' (see guard_mode()): no wait; the previous call's words are looked at instead, and a BsgError raised for them."""
    global range_retries
    import torch
    st = _state()
    with on_device(device):
        capturing = torch.cuda.is_current_stream_capturing()
    if st.depth > 0 or capturing:
        if st.depth > 0:
            for o in owners:
                if all(o is not p for p in st.owners):
                    st.owners.append(o)
        return run()
    st.depth += 1
    st.owners = list(owners)
    try:
        if guard_mode() == 'deferred':
            check_deferred(device)
            # a detection moved a handle to the fp32 pipe for the NEXT call in which it takes part; that call puts it back — also when run()
            # raises (a nested handle's check_deferred, say): the switch back would otherwise be lost and the handle stay on the fp32 pipe
            # for good.  Only the handles of THIS call: another model's call in between must not consume the switch
            try:
                out = run()
            finally:
                for o in st.owners:
                    if getattr(o, '_restore_split_after', False):
                        o._restore_split_after = False
                        o.set_gemm_split(True)
            _deferred_enqueue(st.owners)
            return out
        out = run()
        hits, demoted = _take_all(st.owners, device), []
        while hits:
            import warnings
            names = ', '.join(f'{type(o).__name__} ({n} waves)' for o, n in hits)
            warnings.warn(f'bisinger_amd: {what}: an operand left the fp16 range of the split-fp16 GEMMs (|v| >= 4062) in {names}; the call '
                          f'is repeated with the GEMMs of that handle on the fp32 matrix pipe')
            for o, _ in hits:
                o.set_gemm_split(False)
                o._gemm_strikes = o.gemm_range_strikes + 1
                demoted.append(o)
            range_retries += 1
            st.in_retry = True
            try:
                if on_retry is not None:
                    on_retry()
                out = run()
            finally:
                st.in_retry = False
            # a handle that was clean before can trip in the repeated pass (its input changed with the other handle's arithmetic): it is
            # demoted too and the call repeated once more — every pass takes at least one more handle off the 16-bit pipe, so this ends
            hits = [(o, n) for o, n in _take_all(st.owners, device) if all(o is not p for p in demoted)]
        for o in demoted:
            if o.gemm_range_strikes < RANGE_STRIKES_MAX and _split_env_on():
                o.set_gemm_split(True)
    finally:
        st.depth -= 1
        st.owners = []
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
