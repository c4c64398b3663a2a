"""CPU oracle for the top-down heat-map pose hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy / torch-CPU,
fp32) of the reference algorithms on the hot path named by BASELINE.json's
``north_star``.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
``mindpose_amd`` never imports it and never falls back to it.

Pinning status:

* ``oracle.target``  - PINNED: checked bit-for-bit against golden vectors generated
  by the reference's own numpy implementation (``tests/golden/gen_golden.py`` imports
  ``mindpose/data/transform/topdown_transform.py`` by file path in the build container).
* everything else (decoder, loss, flip aggregation, networks) - PARITY UNPINNED by the
  reference: its arithmetic lives in MindSpore (no pinned version, not vendored, not
  installable here) and the reference's own tests assert shapes only.  These modules
  restate the reference line by line (file:line cited per function), are cross-checked
  against an independent torch-CPU formulation and against the hand known-answers of
  SURVEY.md 8c in ``tests/test_oracle_*.py``.
"""
