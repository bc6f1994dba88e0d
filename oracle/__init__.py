"""CPU oracle for the LDR->HDR sky-panorama hot path.  TEST INFRASTRUCTURE ONLY.

This package is a fp32 CPU *restatement* (torch-CPU functional ops + plain
numpy loops) of the arithmetic the reference performs through TensorFlow 2 /
tensorflow_addons.  It is written from the reference source text; every
function cites the reference file:line it follows.

PARITY UNPINNED: TensorFlow / tensorflow_addons are not installed in the build
container (plain ``ModuleNotFoundError``) and the reference ships no tests or
golden vectors, so this oracle could not be checked against outputs of the
reference itself.  What pins it instead (see DESIGN.md "Oracle"):
two independent restatements of every risky op cross-checked in tests/,
closed-form identities, and finite-difference gradient checks.
Exception: oracle/jpeg.py (the JPEG step of the augmentation) restates libjpeg,
not TensorFlow, and IS pinned - bit-exact against libjpeg-turbo itself.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package - and only as the checker.
The product path (the ``*_amd`` package + libhdrsky.so) never imports it.
"""
