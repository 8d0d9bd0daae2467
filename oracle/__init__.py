"""CPU oracle package -- test infrastructure only (see ppca_oracle.c header)."""
