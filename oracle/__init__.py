"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/nbk_oracle.h).

Nothing under numbotics_amd/ may import this package; tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg are its only users.
"""
