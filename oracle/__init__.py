"""Test infrastructure (CPU oracle of the reference algorithm). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
