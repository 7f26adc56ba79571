"""oracle/ -- CPU restatement of the reference algorithm.  TEST INFRASTRUCTURE ONLY (see oracle.py)."""
