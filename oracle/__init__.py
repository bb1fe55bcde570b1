"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the reference hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker.  The product path (``latte_amd``) never imports this package and
fails loudly when its HIP library is missing.
"""
