"""CPU oracle for the mini-batch hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package (and only as the checker / reported baseline).  The product package under
``cugraph-gnn_amd/`` never imports it.  See ``wg_oracle.c`` for the reference file:line each
function restates.
"""
from .oracle import *  # noqa: F401,F403
