"""CPU oracle + reference harness: test infrastructure only (see gvd_oracle.py header)."""
