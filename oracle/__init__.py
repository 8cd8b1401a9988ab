"""oracle/ — CPU restatement of the reference's mel-generation hot path.  TEST INFRASTRUCTURE.

This package is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
``bisinger_amd`` never imports ``oracle`` (tests/test_abi.py::test_product_package_never_imports_the_oracle enforces that), and the
product path raises when the HIP library is missing instead of falling back to this code.

What it is
    A plain-PyTorch-CPU, functional (state_dict in, tensors out) restatement of
    BiSinger's inference path, written from the reference's algorithm:

    =====================  ==============================================================
    oracle/diffnet.py      DiffNet noise predictor          (TB/usr/diff/net.py:32-130)
    oracle/diffusion.py    beta schedule, p_sample, PLMS    (TB/usr/diff/shallow_diffusion_tts.py:32-285)
    oracle/fs2.py          FastSpeech2-MIDI enc/dec, ESM    (TB/modules/diffsinger_midi/fs2.py:14-197,
                                                             TB/modules/fastspeech/{fs2,tts_modules}.py,
                                                             TB/modules/commons/common_layers.py)
    oracle/hifigan.py      HiFi-GAN generator forward       (TB/modules/hifigan/hifigan.py:30-182)
    =====================  ==============================================================
    (TB/ = /root/reference/train_bisinger/)

How it is pinned
    The reference ships no tests, fixtures or golden vectors and all of its
    arithmetic is third-party ``torch`` (pinned 1.6.0 upstream, 2.10.0 here).  The oracle is
    therefore pinned against **outputs of the reference itself run in the build container**:
    ``tools/make_golden.py`` imports the reference's own modules from /root/reference, loads
    formula weights (``bisinger_amd/synth.py``) into them, runs them, and commits the small
    outputs under ``tests/golden/``.  ``tests/test_oracle_golden.py`` (CPU, no reference
    needed) checks this restatement against every one of those vectors.

    Every function takes ``dtype`` so the same code also gives a float64 trajectory, used
    by the tests to decide whether a deviation is rounding (both fp32 paths scatter around
    the fp64 result) or a bug.
"""
