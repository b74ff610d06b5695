"""CPU oracle for the `goleft depth` hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/depth_oracle.h for the parity status.
"""
