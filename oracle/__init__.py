"""CPU oracle of the stochastic-aggregation path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under stag_amd/ does.
"""
