"""CPU parity oracle (test infrastructure only; see lqr_oracle.h)."""
