"""CPU oracle (TEST INFRASTRUCTURE ONLY — see ov_oracle.h).  PARITY UNPINNED (no reference golden vectors exist)."""
