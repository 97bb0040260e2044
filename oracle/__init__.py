"""CPU oracle for the ST-GCN / VirtualRadar hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The product path (``skeleton-action-recognition_amd/``) never imports
this package and fails loudly when the HIP extension is missing.

Each function cites the reference file:line (relative to the upstream repo
itskalvik/skeleton-action-recognition) whose semantics it restates.
"""
