"""Parity oracle (TEST INFRASTRUCTURE ONLY).

CPU restatement of the reference hot path (oracle/cpd_oracle.c) plus, when present, the compiled
reference iou3d_cpu.cpp (oracle/_ref/). Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product package cpd_amd never does.
"""
from .binding import Oracle, build_oracle, load_reference_iou  # noqa: F401
