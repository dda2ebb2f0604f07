"""TEST INFRASTRUCTURE — CPU restatements of the reference's FLORIS wind-farm step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
