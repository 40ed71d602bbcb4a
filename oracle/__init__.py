"""Oracle package -- TEST INFRASTRUCTURE ONLY (see wb_oracle.py header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from . import wb_oracle  # noqa: F401
