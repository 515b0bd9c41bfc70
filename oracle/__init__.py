"""CPU oracle for the rasterizer hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/torch_oracle.py for the parity status ("parity unpinned" by the reference; pinned by
in-tree fragments + known-answer tests + fp64 finite differences).
"""
