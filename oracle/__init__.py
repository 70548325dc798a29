"""CPU oracle (TEST INFRASTRUCTURE ONLY — see ov_oracle.h).  Pinned by tests/test_known_answer.py (independent mpmath fixtures)."""
