"""CPU oracle for the TwinGAN G+D training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``twingan_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only
as the checker / the timed CPU baseline -- never as the product path.

PARITY STATUS -- pinned at the level of the reference's own code, unpinned below it:

  * PINNED: ``torch_ref.py`` (networks, losses, gradient penalty, spectral-norm state; 8 configurations) reproduces,
    to 1e-9 in float64, every loss term and every gradient that the REFERENCE'S OWN SOURCE computes for the same
    weights, inputs and random draws.  The reference (jerryli27/TwinGAN, Python-2 / TensorFlow-1.8) is executed
    here by ``oracle/ref_runner.py``: its modules are imported from /root/reference unmodified (lib2to3 + Python-2
    division semantics applied in memory) on top of ``oracle/tf_shim`` -- an eager stand-in for the part of the
    TensorFlow-1.8 API they call.  ``tools/make_golden.py`` freezes those runs as ``tests/golden/twingan_*.npz``;
    ``tests/test_golden.py`` checks the oracle (CPU) and the HIP path (GPU) against them and, where /root/reference
    is mounted, re-derives them.
  * NOT PINNED: the arithmetic of the TensorFlow ops themselves (conv2d SAME padding, moments, avg_pool,
    resize_nearest_neighbor, softmax, Adam ...).  It lives in the un-vendored dependency ``tensorflow==1.8``
    (requirement.txt:1), which is not installed and cannot be (no network); ``np_ops.py``, ``torch_ref.py`` and
    ``tf_shim`` restate those ops from their documented semantics, cross-checked by two independent restatements
    agreeing (float64 NumPy <-> torch) and by the analytic known-answer tests of SURVEY.md Appendix B
    (``tests/test_oracle.py``).  The reference's own tests hold no golden vector for this path.
"""
