"""Parity oracle for the mizuRoute hot path -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under mizuroute_amd/ imports this package.
"""
