"""CPU oracle - TEST INFRASTRUCTURE ONLY (see oracle/marl_oracle.c).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package."""
