"""CPU oracle of the Garment4D hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (garment4d_amd/) never does.  See DESIGN.md "Oracle".
"""
