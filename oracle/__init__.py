"""ORACLE -- test infrastructure only.

CPU restatements of the reference's algorithm for the OccFormer forward hot path
(see occformer_ref.py and bev_pool_ref.c).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import, link or execute anything here; the
product (occformer_amd/) never does, and fails loudly without its HIP library.
"""
