"""CPU oracle for the hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/cutesv_oracle.c for the restatement and how its parity is pinned.
"""
