"""CPU oracle of the hot path -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (visual_odometry_amd) never does.
"""
