"""CPU oracle package (test infrastructure only -- see cudf_oracle.py header)."""
