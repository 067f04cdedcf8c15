"""CPU oracle for the DiCoW / SE-DiCoW training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / the timed CPU baseline.  The product
path (``ts-asr-whisper_amd/``) never imports this package and fails loudly when
the HIP library is missing.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against fixtures generated in the build
container by importing the real reference (``tests/golden/make_golden.py``,
fixtures ``tests/golden/*.npz``); ``tests/test_oracle_vs_golden.py`` checks
every function here against them.
"""
