"""CPU oracle for the dsygvdx_gpu / zhegvdx_gpu path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``eigensolver_gpu_amd``) never does.
"""
from .pyoracle import *  # noqa: F401,F403
