"""CPU oracle (test infrastructure only; see oracle/oracle.py).  Never imported by the product path."""
