"""oracle -- CPU restatement of GraphIK's RiemannianSolver hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(as the checker / CPU baseline).  Nothing under graphik_amd/ imports it.
"""
