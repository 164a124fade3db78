"""CPU oracle (TEST INFRASTRUCTURE ONLY — never imported by the product package).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED: see oracle/mpe_oracle.h.
"""
from .binding import *  # noqa: F401,F403
