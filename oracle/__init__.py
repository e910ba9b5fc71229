"""CPU oracle for the TwinGAN G+D training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``twingan_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only
as the checker / the timed CPU baseline -- never as the product path.

PARITY UNPINNED: the reference (jerryli27/TwinGAN, Python-2 / TensorFlow-1.8) cannot be imported
or run in this environment (no TF, no python2 -- SURVEY.md section 8c) and its tests hold no golden
vectors for this path.  The arithmetic lives in the un-vendored dependency
``tensorflow==1.8`` (requirement.txt:1); this package restates the reference's algorithm
line-by-line from the cited files plus the documented TF-1.8 semantics of the stock ops, and is
pinned only by (a) two independent restatements agreeing (float64 NumPy <-> torch-CPU fp32) and
(b) analytic known-answer tests.
"""
