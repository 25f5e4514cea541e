"""CPU parity oracle for the PEM-v0 hot path -- test infrastructure, never imported by the product."""
