"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference EFGH hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (efgh_amd/) never does, and has no CPU fallback.
"""
