"""oracle/ — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
multiposenet.pytorch_amd (the product).  See DESIGN.md "Oracle".
"""
