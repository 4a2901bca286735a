"""Measurement aids that use the oracle as the checker (the oracle may only be used from tests/)."""
